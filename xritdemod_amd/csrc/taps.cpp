// taps.cpp -- host-side filter design, run once per handle.
// Replaces SatHelper::Filters::RRC / Filters::lowPass as called at
// /root/reference/demodulator/src/demodulator.cpp:443-444 and the MMSE
// interpolator table inside SatHelper::ClockRecovery (:449).  The formulae are
// those of the GNU Radio blocks the reference flowgraph names
// (demod_tcp_qt.py:95-96,261-262,266): firdes::root_raised_cosine,
// firdes::low_pass with a Hamming window, mmse_fir_interpolator_cc.
#include "taps.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>

namespace xrit {

static const double kPi = 3.14159265358979323846;

std::vector<float> design_lowpass(double gain, double fs, double cutoff, double tw)
{
    // length rule: Hamming = 53 dB, ntaps = int(53*fs/(22*tw)), made odd
    int n = static_cast<int>(53.0 * fs / (22.0 * tw));
    if (!(n & 1)) ++n;
    const int half = (n - 1) / 2;
    const double w0 = 2.0 * kPi * cutoff / fs;
    std::vector<float> h(static_cast<size_t>(n));
    for (int i = 0; i < n; ++i) {
        const int k = i - half;
        const float win = static_cast<float>(0.54 - 0.46 * std::cos(2.0 * kPi * i / (n - 1)));
        const double ideal = (k == 0) ? w0 / kPi : std::sin(k * w0) / (k * kPi);
        h[static_cast<size_t>(i)] = static_cast<float>(ideal * win);
    }
    // unity (x gain) response at DC, summed the way firdes does it
    double dc = h[static_cast<size_t>(half)];
    for (int k = 1; k <= half; ++k) dc += 2.0 * h[static_cast<size_t>(half + k)];
    const double scale = gain / dc;
    for (auto &v : h) v = static_cast<float>(v * scale);
    return h;
}

std::vector<float> design_rrc(double gain, double fs, double symbol_rate, double alpha, int ntaps)
{
    ntaps |= 1;
    const double spb = fs / symbol_rate;
    std::vector<float> h(static_cast<size_t>(ntaps));
    double sum = 0.0;
    for (int i = 0; i < ntaps; ++i) {
        const double xi = i - ntaps / 2;
        const double a = kPi * xi / spb;
        const double b = 4.0 * alpha * xi / spb;
        double c = b * b - 1.0;
        double num, den;
        if (std::fabs(c) >= 0.000001) {
            num = std::cos((1.0 + alpha) * a);
            num += (i != ntaps / 2) ? std::sin((1.0 - alpha) * a) / (4.0 * alpha * xi / spb)
                                    : (1.0 - alpha) * kPi / (4.0 * alpha);
            den = c * kPi;
        } else {
            if (alpha == 1.0) {
                h[static_cast<size_t>(i)] = -1.0f;
                sum += -1.0;
                continue;
            }
            const double lo = (1.0 - alpha) * a, hi = (1.0 + alpha) * a;
            num = std::sin(hi) * (1.0 + alpha) * kPi
                  - std::cos(lo) * ((1.0 - alpha) * kPi * spb) / (4.0 * alpha * xi)
                  + std::sin(lo) * spb * spb / (4.0 * alpha * xi * xi);
            den = -32.0 * kPi * alpha * alpha * xi / spb;
        }
        h[static_cast<size_t>(i)] = static_cast<float>(4.0 * alpha * num / den);
        sum += h[static_cast<size_t>(i)];
    }
    for (auto &v : h) v = static_cast<float>(v * gain / sum);
    return h;
}

// Least-squares fractional-delay taps for signals band-limited to |f| <= 1/4:
// minimise int_{-B}^{B} |sum_c h_c e^{-j 2 pi f (c-4)} ... |^2 -> normal equations
// with sinc kernels.  Column c of a row multiplies the sample at time 4-c
// (upstream column labels -4..3); the interpolation instant is mu in [0,1].
// Values are rounded through "%.5e" because upstream ships the table as
// 6-significant-digit literals.
static double sinc_pi(double x) { return std::fabs(x) < 1e-12 ? 1.0 : std::sin(kPi * x) / (kPi * x); }

void design_mmse_table(float *table)
{
    const int NT = 8, NS = 128;
    const double twoB = 0.5;
    for (int s = 0; s <= NS; ++s) {
        const double mu = static_cast<double>(s) / NS;
        double G[NT][NT + 1];
        for (int r = 0; r < NT; ++r) {
            for (int c = 0; c < NT; ++c) G[r][c] = twoB * sinc_pi(twoB * (r - c));
            G[r][NT] = twoB * sinc_pi(twoB * (mu + (r - 4)));
        }
        // Gauss-Jordan with partial pivoting on the augmented matrix
        for (int c = 0; c < NT; ++c) {
            int best = c;
            for (int r = c + 1; r < NT; ++r)
                if (std::fabs(G[r][c]) > std::fabs(G[best][c])) best = r;
            if (best != c)
                for (int k = 0; k <= NT; ++k) { double t = G[c][k]; G[c][k] = G[best][k]; G[best][k] = t; }
            const double inv = 1.0 / G[c][c];
            for (int k = c; k <= NT; ++k) G[c][k] *= inv;
            for (int r = 0; r < NT; ++r) {
                if (r == c) continue;
                const double f = G[r][c];
                if (f == 0.0) continue;
                for (int k = c; k <= NT; ++k) G[r][k] -= f * G[c][k];
            }
        }
        for (int c = 0; c < NT; ++c) {
            char txt[40];
            std::snprintf(txt, sizeof txt, "%.5e", G[c][NT]);
            table[s * NT + c] = std::strtof(txt, nullptr);
        }
    }
    for (int c = 0; c < NT; ++c) {
        table[c] = (c == 4) ? 1.0f : 0.0f;
        table[NS * NT + c] = (c == 3) ? 1.0f : 0.0f;
    }
}

}  // namespace xrit
