/*
 * xrit_oracle.c -- CPU restatement of the xritdemod demodulation chain.
 *
 * TEST INFRASTRUCTURE ONLY (see xrit_oracle.h).  PARITY UNPINNED: the reference
 * delegates every per-sample loop to libSatHelper, which is not vendored
 * (/root/reference/Makefile:52-55) and has no tests (/root/reference/Makefile:91-92).
 * Each function below cites the reference call site it serves and names the
 * published algorithm it restates.  All per-sample arithmetic is float32, plain
 * IEEE multiply/add (build with -ffp-contract=off: the reference is built with
 * -O3 only, demodulator/CMakeLists.txt:33-34, i.e. no FMA contraction on x86-64).
 */
#define _GNU_SOURCE
#include "xrit_oracle.h"

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------------- */
/* Knobs (xrit_oracle.h)                                                     */
/* ------------------------------------------------------------------------- */
static xo_knobs g_knobs = {0, XO_MM_FUDGE, 0, 0, 0};

void xo_knobs_default(xo_knobs *k)
{
    k->fir_phase_last = 0;
    k->mm_fudge = XO_MM_FUDGE;
    k->mm_drop_tail = 0;
    k->costas_wrap_pi = 0;
    k->costas_imag_axis = 0;
}

void xo_set_knobs(const xo_knobs *k)
{
    if (!k) { xo_knobs_default(&g_knobs); return; }
    g_knobs = *k;
    if (g_knobs.mm_fudge < 0) g_knobs.mm_fudge = 0;
    if (g_knobs.mm_fudge > 64) g_knobs.mm_fudge = 64;
}

void xo_get_knobs(xo_knobs *k) { *k = g_knobs; }

/* ------------------------------------------------------------------------- */
/* Tap designers                                                             */
/* ------------------------------------------------------------------------- */

/* Filters::lowPass(1, Fs, Fs_circ/2, 100e3, HAMMING, 6.76), demodulator.cpp:444.
 * GNU Radio firdes::low_pass: length = int(A*Fs/(22*tw)) forced odd with
 * A = 53 dB for Hamming (beta is ignored for Hamming); windowed sinc; taps
 * scaled so that the DC gain equals `gain`. */
int xo_lowpass_ntaps(double fs, double transition_width)
{
    int ntaps = (int)(53.0 * fs / (22.0 * transition_width));
    if ((ntaps & 1) == 0) ntaps++;
    return ntaps;
}

int xo_lowpass_taps(double gain, double fs, double cutoff, double transition_width,
                    float *taps, int cap)
{
    int ntaps = xo_lowpass_ntaps(fs, transition_width);
    if (ntaps > cap) return -ntaps;
    int M = (ntaps - 1) / 2;
    double fwT0 = 2.0 * M_PI * cutoff / fs;
    for (int n = -M; n <= M; n++) {
        /* Hamming window, stored as float like the upstream window vector */
        float w = (float)(0.54 - 0.46 * cos((2.0 * M_PI * (n + M)) / (ntaps - 1)));
        if (n == 0)
            taps[n + M] = (float)(fwT0 / M_PI * w);
        else
            taps[n + M] = (float)(sin(n * fwT0) / (n * M_PI) * w);
    }
    double fmax = taps[M];
    for (int n = 1; n <= M; n++) fmax += 2.0 * taps[n + M];
    gain /= fmax;
    for (int i = 0; i < ntaps; i++) taps[i] = (float)(taps[i] * gain);
    return ntaps;
}

/* Filters::RRC(1, Fs_circ, symbolRate, alpha, 63), demodulator.cpp:443.
 * GNU Radio firdes::root_raised_cosine. */
int xo_rrc_taps(double gain, double fs, double symbol_rate, double alpha, int ntaps,
                float *taps, int cap)
{
    ntaps |= 1;
    if (ntaps > cap) return -ntaps;
    double spb = fs / symbol_rate;
    double scale = 0;
    for (int i = 0; i < ntaps; i++) {
        double x1, x2, x3, num, den;
        double xindx = i - ntaps / 2;
        x1 = M_PI * xindx / spb;
        x2 = 4 * alpha * xindx / spb;
        x3 = x2 * x2 - 1;
        if (fabs(x3) >= 0.000001) {
            if (i != ntaps / 2)
                num = cos((1 + alpha) * x1) + sin((1 - alpha) * x1) / (4 * alpha * xindx / spb);
            else
                num = cos((1 + alpha) * x1) + (1 - alpha) * M_PI / (4 * alpha);
            den = x3 * M_PI;
        } else {
            if (alpha == 1) {
                taps[i] = -1;
                scale += taps[i];
                continue;
            }
            x3 = (1 - alpha) * x1;
            x2 = (1 + alpha) * x1;
            num = (sin(x2) * (1 + alpha) * M_PI
                   - cos(x3) * ((1 - alpha) * M_PI * spb) / (4 * alpha * xindx)
                   + sin(x3) * spb * spb / (4 * alpha * xindx * xindx));
            den = -32 * M_PI * alpha * alpha * xindx / spb;
        }
        taps[i] = (float)(4 * alpha * num / den);
        scale += taps[i];
    }
    for (int i = 0; i < ntaps; i++) taps[i] = (float)(taps[i] * gain / scale);
    return ntaps;
}

/* MMSE interpolator table used by ClockRecovery (demodulator.cpp:449).
 * GNU Radio ships a generated table (interpolator_taps.h, 8 taps x 128 steps)
 * that minimises the mean squared error of a fractional delay for signals
 * band-limited to |f| <= B = 0.25 cycles/sample.  The generated file is not
 * available here; this solves the same least-squares problem in closed form:
 *     R h = p,  R[a][b] = 2B sinc(2B (a-b)),  p[a] = 2B sinc(2B (mu + j_a))
 * where column a carries the label j_a = a-4 ("-4 ... 3" in the upstream header)
 * and multiplies the sample at time -j_a; the target instant is mu in [0,1].
 * Row 0 is the unit tap on column 4, row 128 the unit tap on column 3.
 * Spot check against rows recalled from the upstream header (unverifiable
 * here): row 1 = {-1.54700e-04, 8.53777e-04, -2.76968e-03, 7.89295e-03,
 * 9.98534e-01, -5.41054e-03, 1.24642e-03, -1.98993e-04} and row 64 =
 * {-6.77751e-03, 3.94578e-02, -1.42658e-01, 6.09836e-01, ...mirror} agree with
 * this solution in every printed digit (tests/test_oracle_kat.py).
 */
static double xo_sinc(double x)
{
    if (fabs(x) < 1e-12) return 1.0;
    return sin(M_PI * x) / (M_PI * x);
}

static void xo_solve8(double A[8][8], double b[8], double x[8])
{
    int n = 8;
    for (int c = 0; c < n; c++) {
        int piv = c;
        for (int r = c + 1; r < n; r++)
            if (fabs(A[r][c]) > fabs(A[piv][c])) piv = r;
        if (piv != c) {
            for (int k = 0; k < n; k++) { double t = A[c][k]; A[c][k] = A[piv][k]; A[piv][k] = t; }
            double t = b[c]; b[c] = b[piv]; b[piv] = t;
        }
        for (int r = c + 1; r < n; r++) {
            double f = A[r][c] / A[c][c];
            for (int k = c; k < n; k++) A[r][k] -= f * A[c][k];
            b[r] -= f * b[c];
        }
    }
    for (int r = n - 1; r >= 0; r--) {
        double s = b[r];
        for (int k = r + 1; k < n; k++) s -= A[r][k] * x[k];
        x[r] = s / A[r][r];
    }
}

void xo_mmse_table(float *table)
{
    const double B = 0.25;
    for (int s = 0; s <= XO_MM_NSTEPS; s++) {
        double mu = (double)s / XO_MM_NSTEPS;
        double A[8][8], p[8], h[8];
        for (int a = 0; a < 8; a++) {
            for (int b = 0; b < 8; b++) A[a][b] = 2 * B * xo_sinc(2 * B * (a - b));
            p[a] = 2 * B * xo_sinc(2 * B * (mu + (a - 4)));
        }
        xo_solve8(A, p, h);
        for (int a = 0; a < 8; a++) {
            /* upstream stores the table as "%.5e" literals: round the same way */
            char txt[32];
            snprintf(txt, sizeof txt, "%.5e", h[a]);
            table[s * 8 + a] = strtof(txt, NULL);
        }
    }
    /* exact end rows, as in the upstream table */
    for (int a = 0; a < 8; a++) {
        table[a] = (a == 4) ? 1.0f : 0.0f;
        table[XO_MM_NSTEPS * 8 + a] = (a == 3) ? 1.0f : 0.0f;
    }
}

/* ------------------------------------------------------------------------- */
/* FirFilter                                                                 */
/* ------------------------------------------------------------------------- */

struct xo_fir {
    unsigned D;
    int      phase;  /* 0, or D-1 (knob fir_phase_last): output m reads x[m D + phase - k] */
    int      T;
    float   *rtaps;  /* reversed taps: rtaps[i] = h[T-1-i] */
    xo_cf   *buf;    /* [T-1 history | new samples] */
    size_t   cap;
};

xo_fir *xo_fir_create(unsigned decimation, const float *taps, int ntaps)
{
    xo_fir *f = (xo_fir *)calloc(1, sizeof(*f));
    f->D = decimation ? decimation : 1;
    f->phase = g_knobs.fir_phase_last ? (int)f->D - 1 : 0;
    f->T = ntaps;
    f->rtaps = (float *)malloc(sizeof(float) * ntaps);
    for (int i = 0; i < ntaps; i++) f->rtaps[i] = taps[ntaps - 1 - i];
    f->cap = 0;
    f->buf = NULL;
    return f;
}

void xo_fir_destroy(xo_fir *f)
{
    if (!f) return;
    free(f->rtaps);
    free(f->buf);
    free(f);
}

/* FirFilter::Work(in, out, nOut), demodulator.cpp:138,148.  GNU Radio
 * fir_filter_ccf semantics: T-1 samples of zero-initialised history persist
 * across calls; output m of a call is sum_k h[k] x[m*D - k] with x[0] the first
 * new sample.  The summation order of the upstream dot product (VOLK SIMD) is
 * unspecified; this restatement accumulates in float32 in four interleaved
 * partial sums (lane = tap index mod 4 over the time-ordered window), then
 * (s0+s1)+(s2+s3) -- the shape of a 4-wide SIMD dot product. */
void xo_fir_work(xo_fir *f, const xo_cf *in, xo_cf *out, int n_out)
{
    int T = f->T;
    size_t n_in = (size_t)n_out * f->D;
    size_t need = (size_t)(T - 1) + n_in;
    if (need > f->cap) {
        xo_cf *nb = (xo_cf *)calloc(need + 64, sizeof(xo_cf));
        if (f->buf) memcpy(nb, f->buf, sizeof(xo_cf) * (size_t)(T - 1));
        free(f->buf);
        f->buf = nb;
        f->cap = need + 64;
    }
    memcpy(f->buf + (T - 1), in, sizeof(xo_cf) * n_in);
    const float *rt = f->rtaps;
    for (int m = 0; m < n_out; m++) {
        const xo_cf *w = f->buf + (size_t)m * f->D + f->phase; /* w[i] = x[m*D + phase - (T-1) + i] */
        float sr[4] = {0, 0, 0, 0}, si[4] = {0, 0, 0, 0};
        int i = 0;
        for (; i + 4 <= T; i += 4) {
            for (int j = 0; j < 4; j++) {
                sr[j] += rt[i + j] * w[i + j].re;
                si[j] += rt[i + j] * w[i + j].im;
            }
        }
        for (int j = 0; i < T; i++, j++) {
            sr[j] += rt[i] * w[i].re;
            si[j] += rt[i] * w[i].im;
        }
        out[m].re = (sr[0] + sr[1]) + (sr[2] + sr[3]);
        out[m].im = (si[0] + si[1]) + (si[2] + si[3]);
    }
    /* keep the last T-1 consumed samples as history */
    if (n_in > 0) memmove(f->buf, f->buf + n_in, sizeof(xo_cf) * (size_t)(T - 1));
}

/* ------------------------------------------------------------------------- */
/* AGC                                                                       */
/* ------------------------------------------------------------------------- */

void xo_agc_init(xo_agc *a, float rate, float reference, float gain, float max_gain)
{
    a->rate = rate;
    a->reference = reference;
    a->gain = gain;
    a->max_gain = max_gain;
}

/* AGC::Work, demodulator.cpp:143 with AGC(0.01, 0.5, 1, 4000) (:447).
 * GNU Radio analog::agc_cc::scale + set_max_gain. */
void xo_agc_work(xo_agc *a, const xo_cf *in, xo_cf *out, int n)
{
    float g = a->gain;
    for (int i = 0; i < n; i++) {
        float yr = in[i].re * g;
        float yi = in[i].im * g;
        out[i].re = yr;
        out[i].im = yi;
        g += a->rate * (a->reference - sqrtf(yr * yr + yi * yi));
        if (a->max_gain > 0.0f && g > a->max_gain) g = a->max_gain;
    }
    a->gain = g;
}

/* ------------------------------------------------------------------------- */
/* CostasLoop                                                                */
/* ------------------------------------------------------------------------- */

/* CostasLoop(pllAlpha, LOOP_ORDER=2), demodulator.cpp:448; loop bandwidth
 * defaults to CLOCK_ALPHA=0.0037 (demodulator.cpp:220).  GNU Radio
 * blocks::control_loop(bw, +1, -1): damping sqrt(2)/2. */
void xo_costas_init(xo_costas *c, float loop_bw)
{
    float damping = sqrtf(2.0f) / 2.0f;
    float denom = (1.0f + 2.0f * damping * loop_bw + loop_bw * loop_bw);
    c->alpha = (4 * damping * loop_bw) / denom;
    c->beta = (4 * loop_bw * loop_bw) / denom;
    c->phase = 0;
    c->freq = 0;
    c->max_freq = 1.0f;
    c->min_freq = -1.0f;
    c->wrap_pi = g_knobs.costas_wrap_pi;
    c->imag_axis = g_knobs.costas_imag_axis;
}

/* sin and cos of the loop phase.  The upstream block calls the C library's sincosf (gr::sincosf -> ::sincosf); on this
 * image that is glibc 2.35's __sincosf_fma (sysdeps/ieee754/flt-32/s_sincosf.c with the x86 two-lane polynomial of
 * sysdeps/x86/fpu/sincosf_poly.h, selected by ifunc on every CPU with FMA): range reduction and two polynomials in
 * double precision with fused multiply-adds, rounded to float once.  It is restated here operation for operation so that
 * (a) the oracle's bits do not depend on which variant the host's ifunc picks, and (b) the device can evaluate the very
 * same double-precision operations (csrc/exact_sincos.h).  tests/test_oracle_kat.py::test_sincosf_restatement checks
 * it against the C library on a sample; oracle/check_sincosf.c does so for EVERY float with |x| < 120 (2 246 049 792
 * arguments, 0 differences on this image).  |x| >= 120 (never a loop phase: those are wrapped to +-2 pi) defers to libm. */
static const double XO_SC_HPI_INV = 0x1.45F306DC9C883p+23;   /* 2/pi * 2^24 */
static const double XO_SC_HPI = 0x1.921FB54442D18p0;         /* pi/2 */
static const double XO_SC_C[5] = {0x1p0, -0x1.ffffffd0c621cp-2, 0x1.55553e1068f19p-5, -0x1.6c087e89a359dp-10,
                                  0x1.99343027bf8c3p-16};
static const double XO_SC_S[3] = {-0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13};

static inline void xo_sincosf_poly(double x, double x2, int negc, int n, float *sinp, float *cosp)
{
    /* quadrants 2 and 3 (n & 2): the cosine polynomial negated, the sine's sign folded into x by the caller */
    const double c0 = negc ? -XO_SC_C[0] : XO_SC_C[0], c1 = negc ? -XO_SC_C[1] : XO_SC_C[1];
    const double c2 = negc ? -XO_SC_C[2] : XO_SC_C[2], c3 = negc ? -XO_SC_C[3] : XO_SC_C[3];
    const double c4 = negc ? -XO_SC_C[4] : XO_SC_C[4];
    const double x3 = x2 * x, x4 = x2 * x2;
    const double s1 = __builtin_fma(x2, XO_SC_S[2], XO_SC_S[1]);
    const double cc2 = __builtin_fma(x2, c4, c3);
    const double cc1 = __builtin_fma(x2, c1, c0);
    const double x5 = x3 * x2, x6 = x4 * x2;
    const double s = __builtin_fma(x3, XO_SC_S[0], x);
    const double c = __builtin_fma(x4, c2, cc1);
    const float sv = (float)__builtin_fma(x5, s1, s);
    const float cv = (float)__builtin_fma(x6, cc2, c);
    if (n & 1) { *sinp = cv; *cosp = sv; }
    else { *sinp = sv; *cosp = cv; }
}

void xo_sincosf(float y, float *sinp, float *cosp)
{
    uint32_t u;
    memcpy(&u, &y, 4);
    const uint32_t top = (u >> 20) & 0x7ff;       /* abstop12 */
    const double x = (double)y;
    if (top < 0x3f4) {                            /* |y| < pi/4 */
        if (top < 0x398) {                        /* |y| < 2^-12 */
            *sinp = y;
            *cosp = 1.0f;
            return;
        }
        xo_sincosf_poly(x, x * x, 0, 0, sinp, cosp);
        return;
    }
    if (top >= 0x42f) {                           /* |y| >= 120: not a loop phase */
        sincosf(y, sinp, cosp);
        return;
    }
    const double r = x * XO_SC_HPI_INV;
    const int n = ((int32_t)r + 0x800000) >> 24;
    const double xr = __builtin_fma(-(double)n, XO_SC_HPI, x);
    const double sg = ((n + 1) & 2) ? -1.0 : 1.0; /* sign[n & 3] = {1, -1, -1, 1} */
    xo_sincosf_poly(xr * sg, xr * xr, (n & 2) != 0, n, sinp, cosp);
}

static inline float xo_clip(float x, float clip)
{
    /* branchless_clip: 0.5*(|x+clip| - |x-clip|) */
    float x1 = fabsf(x + clip);
    float x2 = fabsf(x - clip);
    x1 -= x2;
    return 0.5f * x1;
}

/* CostasLoop::Work, demodulator.cpp:152.  GNU Radio digital::costas_loop_cc,
 * order 2, no SNR weighting: out = in * exp(-j phase); error = re*im clipped to
 * +-1; freq += beta*err; phase += freq + alpha*err; wrap to +-2pi; limit freq. */
void xo_costas_work(xo_costas *c, const xo_cf *in, xo_cf *out, int n)
{
    const float twopi = (float)(2.0 * M_PI);
    float phase = c->phase, freq = c->freq;
    for (int i = 0; i < n; i++) {
        float s, co;
        xo_sincosf(-phase, &s, &co);
        float yr = in[i].re * co - in[i].im * s;
        float yi = in[i].re * s + in[i].im * co;
        out[i].re = yr;
        out[i].im = yi;
        float err = yr * yi;
        if (c->imag_axis) err = -err;                  /* knob: stable lock a quarter turn away */
        err = xo_clip(err, 1.0f);
        freq = freq + c->beta * err;
        phase = phase + freq + c->alpha * err;
        if (c->wrap_pi) {                              /* knob: (-pi, pi] */
            while (phase > 0.5f * twopi) phase -= twopi;
            while (phase <= -0.5f * twopi) phase += twopi;
        } else {
            while (phase > twopi) phase -= twopi;
            while (phase < -twopi) phase += twopi;
        }
        if (freq > c->max_freq) freq = c->max_freq;
        else if (freq < c->min_freq) freq = c->min_freq;
    }
    c->phase = phase;
    c->freq = freq;
}

/* ------------------------------------------------------------------------- */
/* ClockRecovery (Mueller & Mueller, complex)                                */
/* ------------------------------------------------------------------------- */

struct xo_mm {
    float mu, omega, omega_mid, omega_lim, gain_omega, gain_mu;
    xo_cf p_2t, p_1t, p_0t, c_2t, c_1t, c_0t;
    float table[(XO_MM_NSTEPS + 1) * XO_MM_NTAPS];
    xo_cf *buf;   /* [carry | new] */
    size_t cap;
    int    carry;
    int    fudge, drop_tail;   /* knobs mm_fudge, mm_drop_tail */
};

xo_mm *xo_mm_create(float omega, float gain_omega, float mu, float gain_mu, float omega_rel_limit)
{
    xo_mm *m = (xo_mm *)calloc(1, sizeof(*m));
    m->mu = mu;
    m->omega = omega;
    m->omega_mid = omega;
    m->omega_lim = omega * omega_rel_limit;
    m->gain_omega = gain_omega;
    m->gain_mu = gain_mu;
    m->fudge = g_knobs.mm_fudge;
    m->drop_tail = g_knobs.mm_drop_tail;
    xo_mmse_table(m->table);
    return m;
}

void xo_mm_destroy(xo_mm *m)
{
    if (!m) return;
    free(m->buf);
    free(m);
}

void xo_mm_get_state(const xo_mm *m, xo_mm_state *s)
{
    s->mu = m->mu; s->omega = m->omega; s->omega_mid = m->omega_mid; s->omega_lim = m->omega_lim;
    s->gain_omega = m->gain_omega; s->gain_mu = m->gain_mu;
    s->p_2t = m->p_2t; s->p_1t = m->p_1t; s->p_0t = m->p_0t;
    s->c_2t = m->c_2t; s->c_1t = m->c_1t; s->c_0t = m->c_0t;
    s->carry = m->carry;
}

/* Test infrastructure for the multi-rank tests (tests/dist_twin.py): the state ClockRecovery carries from one Work call to
 * the next -- the reference holds ONE such object for the whole stream (demodulator.cpp:449) -- taken out of one object and
 * put into another, so that a CPU twin of a stream cut across ranks can hand it on as the product does across GPUs. */
void xo_mm_export(const xo_mm *m, xo_mm_state *s, xo_cf *carry /* [s->carry] after the call; at least 2048 */)
{
    xo_mm_get_state(m, s);
    if (m->carry > 0) memcpy(carry, m->buf, sizeof(xo_cf) * (size_t)m->carry);
}

void xo_mm_import(xo_mm *m, const xo_mm_state *s, const xo_cf *carry)
{
    m->mu = s->mu; m->omega = s->omega;
    m->p_2t = s->p_2t; m->p_1t = s->p_1t; m->p_0t = s->p_0t;
    m->c_2t = s->c_2t; m->c_1t = s->c_1t; m->c_0t = s->c_0t;
    if ((size_t)s->carry > m->cap) {
        free(m->buf);
        m->buf = (xo_cf *)malloc(sizeof(xo_cf) * ((size_t)s->carry + 64));
        m->cap = (size_t)s->carry + 64;
    }
    if (s->carry > 0) memcpy(m->buf, carry, sizeof(xo_cf) * (size_t)s->carry);
    m->carry = s->carry;
}

/* ClockRecovery::Work(in, out, n) -> symbols, demodulator.cpp:156.  GNU Radio
 * digital::clock_recovery_mm_cc::general_work.  A symbol is produced while the
 * read index ii satisfies ii < available - NTAPS - FUDGE (the upstream `ni`);
 * samples not yet consumed are carried to the next call, which makes the
 * symbol sequence independent of how the stream is chunked (GNU Radio's
 * scheduler gives the same guarantee through consume_each(ii)). */
int xo_mm_work_trace(xo_mm *m, const xo_cf *in, int n, xo_cf *out, int *arm, float *mu_trace)
{
    size_t total = (size_t)m->carry + (size_t)n;
    if (total > m->cap) {
        xo_cf *nb = (xo_cf *)malloc(sizeof(xo_cf) * (total + 64));
        if (m->carry) memcpy(nb, m->buf, sizeof(xo_cf) * (size_t)m->carry);
        free(m->buf);
        m->buf = nb;
        m->cap = total + 64;
    }
    memcpy(m->buf + m->carry, in, sizeof(xo_cf) * (size_t)n);
    const xo_cf *x = m->buf;
    long ni = (long)total - XO_MM_NTAPS - m->fudge;
    long ii = 0;
    int oo = 0;
    float mu = m->mu, omega = m->omega;
    /* a symbol consumes at least one sample in any sane configuration: the bound only stops a NaN state (an AGC
     * driven outside its stable range upstream) from emitting symbols for ever into the caller's n + 64 buffer */
    while (ii < ni && oo < n + XO_MM_FUDGE + 8) {
        m->p_2t = m->p_1t;
        m->p_1t = m->p_0t;
        /* mmse_fir_interpolator_cc::interpolate */
        int imu = (int)rint(mu * XO_MM_NSTEPS);
        const float *row = m->table + imu * XO_MM_NTAPS;
        float ar = 0, ai = 0;
        for (int k = 0; k < XO_MM_NTAPS; k++) {
            ar += row[XO_MM_NTAPS - 1 - k] * x[ii + k].re;
            ai += row[XO_MM_NTAPS - 1 - k] * x[ii + k].im;
        }
        m->p_0t.re = ar;
        m->p_0t.im = ai;

        m->c_2t = m->c_1t;
        m->c_1t = m->c_0t;
        /* slicer_0deg */
        m->c_0t.re = m->p_0t.re > 0 ? 1.0f : 0.0f;
        m->c_0t.im = m->p_0t.im > 0 ? 1.0f : 0.0f;

        /* x = (c0 - c2) * conj(p1); y = (p0 - p2) * conj(c1); mm = Re(y - x) */
        float dcr = m->c_0t.re - m->c_2t.re, dci = m->c_0t.im - m->c_2t.im;
        float xr = dcr * m->p_1t.re + dci * m->p_1t.im;
        float dpr = m->p_0t.re - m->p_2t.re, dpi = m->p_0t.im - m->p_2t.im;
        float yr = dpr * m->c_1t.re + dpi * m->c_1t.im;
        float mm_val = yr - xr;
        if (arm) arm[oo] = imu;
        if (mu_trace) mu_trace[oo] = mu;
        out[oo++] = m->p_0t;

        mm_val = xo_clip(mm_val, 1.0f);
        omega = omega + m->gain_omega * mm_val;
        omega = m->omega_mid + xo_clip(omega - m->omega_mid, m->omega_lim);
        mu = mu + omega + m->gain_mu * mm_val;
        float fl = floorf(mu);
        ii += (long)fl;
        mu -= fl;
    }
    m->mu = mu;
    m->omega = omega;
    if (ii > (long)total) ii = (long)total;
    m->carry = m->drop_tail ? 0 : (int)((long)total - ii);
    memmove(m->buf, m->buf + ii, sizeof(xo_cf) * (size_t)m->carry);
    return oo;
}

int xo_mm_work(xo_mm *m, const xo_cf *in, int n, xo_cf *out)
{
    return xo_mm_work_trace(m, in, n, out, NULL, NULL);
}

/* ------------------------------------------------------------------------- */
/* The chain                                                                 */
/* ------------------------------------------------------------------------- */

static void xo_config_common(xo_config *c, float sample_rate, uint32_t decimation)
{
    c->sample_rate = sample_rate;
    c->decimation = decimation;
    c->rrc_taps = 63;                                   /* RRC_TAPS */
    c->agc_rate = 0.01f;                                /* AGC_RATE */
    c->agc_reference = 0.5f;                            /* AGC_REFERENCE */
    c->agc_gain = 1.f;                                  /* AGC_GAIN */
    c->agc_max_gain = 4000;                             /* AGC_MAX_GAIN */
    c->pll_alpha = 0.0037f;                             /* (float)CLOCK_ALPHA, demodulator.cpp:220 */
    c->clock_mu = 0.5f;                                 /* CLOCK_MU */
    c->clock_alpha = 0.0037f;                           /* CLOCK_ALPHA */
    c->clock_gain_omega = (0.0037f * 0.0037f) / 4.0f;   /* CLOCK_GAIN_OMEGA */
    c->clock_omega_limit = 0.005f;                      /* CLOCK_OMEGA_LIMIT */
}

void xo_config_lrit(xo_config *c, float sample_rate, uint32_t decimation)
{
    xo_config_common(c, sample_rate, decimation);
    c->symbol_rate = 293883;
    c->rrc_alpha = 0.5f;
}

void xo_config_hrit(xo_config *c, float sample_rate, uint32_t decimation)
{
    xo_config_common(c, sample_rate, decimation);
    c->symbol_rate = 927000;
    c->rrc_alpha = 0.3f;
}

struct xo_demod {
    xo_config cfg;
    float     sps;
    int       dec_ntaps;
    float    *dec_taps;
    float     rrc_taps[256];
    xo_fir   *decimator;
    xo_fir   *rrc;
    xo_agc    agc;
    xo_costas costas;
    xo_mm    *mm;
    xo_cf    *stage[6];  /* 0 converted input .. 5 clock recovery */
    int       stage_n[6];
    size_t    cap;
    int       imag_axis; /* knob costas_imag_axis: SymbolManager keeps Im(symbol) */
    xo_rtl    rtl;       /* the RTL frontend's conversion state (XO_SAMPLE_U8IQ) */
};

/* main(), demodulator.cpp:436-450 */
xo_demod *xo_demod_create(const xo_config *cfg)
{
    xo_demod *d = (xo_demod *)calloc(1, sizeof(*d));
    d->cfg = *cfg;
    float circuit_rate = cfg->sample_rate / ((float)cfg->decimation);  /* :436 */
    float sps = circuit_rate / ((float)cfg->symbol_rate);              /* :437 */
    d->sps = sps;
    xo_rrc_taps(1, circuit_rate, cfg->symbol_rate, cfg->rrc_alpha, cfg->rrc_taps, d->rrc_taps, 256);
    d->dec_ntaps = xo_lowpass_ntaps(cfg->sample_rate, 100e3);
    d->dec_taps = (float *)malloc(sizeof(float) * (size_t)d->dec_ntaps);
    xo_lowpass_taps(1, cfg->sample_rate, circuit_rate / 2, 100e3, d->dec_taps, d->dec_ntaps);
    d->decimator = xo_fir_create(cfg->decimation, d->dec_taps, d->dec_ntaps);
    xo_agc_init(&d->agc, cfg->agc_rate, cfg->agc_reference, cfg->agc_gain, cfg->agc_max_gain);
    xo_costas_init(&d->costas, cfg->pll_alpha);
    d->mm = xo_mm_create(sps, cfg->clock_gain_omega, cfg->clock_mu, cfg->clock_alpha,
                         cfg->clock_omega_limit);
    d->rrc = xo_fir_create(1, d->rrc_taps, cfg->rrc_taps | 1);
    d->imag_axis = g_knobs.costas_imag_axis;
    xo_rtl_init(&d->rtl, cfg->sample_rate);
    return d;
}

void xo_demod_destroy(xo_demod *d)
{
    if (!d) return;
    xo_fir_destroy(d->decimator);
    xo_fir_destroy(d->rrc);
    xo_mm_destroy(d->mm);
    free(d->dec_taps);
    for (int i = 0; i < 6; i++) free(d->stage[i]);
    free(d);
}

/* RtlFrontend (RtlFrontend.cpp:26-28 table, :57 alpha, :102-116 conversion) */
void xo_rtl_init(xo_rtl *r, float sample_rate)
{
    for (int i = 0; i < 256; i++) r->lut[i] = (i - 128) * (1.f / 127.f);
    r->alpha = 1.f - exp(-1.0 / (sample_rate * 0.05f));
    r->iavg = 0;
    r->qavg = 0;
}

void xo_rtl_work(xo_rtl *r, const uint8_t *data, unsigned int length, float *iq)
{
    for (unsigned int i = 0; i < length; i++) {
        iq[i] = r->lut[data[i]];
        if (i % 1) {                                   /* sic: never true */
            r->qavg += r->alpha * (iq[i] - r->qavg);
            iq[i] -= r->qavg;
        } else {
            r->iavg += r->alpha * (iq[i] - r->iavg);
            iq[i] -= r->iavg;
        }
    }
}

/* onSamplesAvailable, demodulator.cpp:54-74 (+ the FIFO -> complex
 * deinterleave of :124-128, which is the identity on an interleaved buffer) */
void xo_convert_samples(const void *in, int sample_type, xo_cf *out, size_t n)
{
    if (sample_type == XO_SAMPLE_FLOATIQ) {
        memcpy(out, in, n * sizeof(xo_cf));
    } else if (sample_type == XO_SAMPLE_S16IQ) {
        const int16_t *d = (const int16_t *)in;
        for (size_t i = 0; i < n; i++) {
            out[i].re = d[2 * i] / 32768.f;
            out[i].im = d[2 * i + 1] / 32768.f;
        }
    } else if (sample_type == XO_SAMPLE_S8IQ) {
        const int8_t *d = (const int8_t *)in;
        for (size_t i = 0; i < n; i++) {
            out[i].re = d[2 * i] / 128.f;
            out[i].im = d[2 * i + 1] / 128.f;
        }
    }
}

/* processSamples(), demodulator.cpp:100-168 */
int xo_demod_process(xo_demod *d, const void *samples, int n, int sample_type,
                     float *soft_out, int cap_out)
{
    if (n <= 0) return 0;
    if ((size_t)n > d->cap) {
        for (int i = 0; i < 6; i++) {
            free(d->stage[i]);
            d->stage[i] = (xo_cf *)malloc(sizeof(xo_cf) * ((size_t)n + 64));
        }
        d->cap = (size_t)n;
    }
    int length = n;
    if (sample_type == XO_SAMPLE_U8IQ)      /* the frontend converts, the callback receives FLOATIQ */
        xo_rtl_work(&d->rtl, (const uint8_t *)samples, 2u * (unsigned int)n, (float *)d->stage[0]);
    else
        xo_convert_samples(samples, sample_type, d->stage[0], (size_t)n);
    const xo_cf *cur = d->stage[0];
    d->stage_n[0] = n;
    if (d->cfg.decimation > 1) {                       /* :136-140 */
        length /= (int)d->cfg.decimation;              /* remainder samples are dropped */
        xo_fir_work(d->decimator, cur, d->stage[1], length);
        cur = d->stage[1];
    } else {
        memcpy(d->stage[1], cur, sizeof(xo_cf) * (size_t)length);
    }
    d->stage_n[1] = length;
    xo_agc_work(&d->agc, cur, d->stage[2], length);    /* :143 */
    d->stage_n[2] = length;
    xo_fir_work(d->rrc, d->stage[2], d->stage[3], length);      /* :148 */
    d->stage_n[3] = length;
    xo_costas_work(&d->costas, d->stage[3], d->stage[4], length); /* :152 */
    d->stage_n[4] = length;
    /* the M&M output never exceeds the input length for sps >= 1 */
    int symbols = xo_mm_work(d->mm, d->stage[4], length, d->stage[5]); /* :156 */
    d->stage_n[5] = symbols;
    if (symbols > cap_out) return -symbols;
    for (int i = 0; i < symbols; i++)                                  /* SymbolManager.cpp:104 */
        soft_out[i] = d->imag_axis ? d->stage[5][i].im : d->stage[5][i].re;
    return symbols;
}

const xo_cf *xo_demod_stage(const xo_demod *d, int stage, int *n)
{
    if (stage < 0 || stage > 4) return NULL;
    if (n) *n = d->stage_n[stage + 1];
    return d->stage[stage + 1];
}

int xo_demod_decimator_ntaps(const xo_demod *d) { return d->dec_ntaps; }
const float *xo_demod_decimator_taps(const xo_demod *d) { return d->dec_taps; }
const float *xo_demod_rrc_taps(const xo_demod *d) { return d->rrc_taps; }
float xo_demod_sps(const xo_demod *d) { return d->sps; }
/* the chain's own loop objects (tests: a Costas loop started at a phase of pi, the clock recovery's carried state) */
xo_costas *xo_demod_costas(xo_demod *d) { return &d->costas; }
xo_mm *xo_demod_mm(xo_demod *d) { return d->mm; }

/* SymbolManager::process, SymbolManager.cpp:43-46 */
void xo_quantize_i8(const float *in, int8_t *out, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        float f = in[i] * 127;
        f = f > 127 ? 127 : f;
        f = f < -128 ? -128 : f;
        out[i] = (int8_t)(char)(f);
    }
}

/* ---- decoder front end ("next" row): frame synchronisation ---------------------------
 * What decoder/src/newdecoder.cpp does with SatHelper::Correlator before Viterbi (:145-151 addWord of the two
 * rate-1/2 encoded 64-bit sync words, :218-245 correlate / getHighestCorrelationPosition / WordNumber /
 * getHighestCorrelation, MINCORRELATIONBITS 46 at decoder/src/parameters.h:31).  The class lives in libSatHelper
 * (absent, like the DSP blocks: parity unpinned); restated from its published behaviour: a word's bit k (MSB
 * first) is stored as 0xFF / 0x00; a soft byte agrees with it when
 *      (byte >= 127 && wordbyte == 0x00) || (byte < 127 && wordbyte == 0xFF)        -- bytes taken as UNSIGNED,
 * i.e. int8 0..126 count as a one, 127 and every negative value as a zero; for every start position i in
 * [0, length - 64) the agreeing bits are counted per word; per word the FIRST position with the highest count is
 * kept (strict >), then the FIRST word with the highest count. */
void xo_sync_correlate(const int8_t *data, uint32_t length, const uint64_t *words, int nwords,
                       uint32_t *word_out, uint32_t *pos_out, uint32_t *corr_out)
{
    uint32_t best_c[8] = {0}, best_p[8] = {0};
    if (nwords > 8) nwords = 8;
    const int max_search = (int)length - 64;
    for (int i = 0; i < max_search; i++) {
        for (int n = 0; n < nwords; n++) {
            uint32_t c = 0;
            for (int k = 0; k < 64; k++) {
                const uint8_t b = (uint8_t)data[i + k];
                const uint8_t w = ((words[n] >> (63 - k)) & 1) ? 0xFF : 0x00;
                c += (uint32_t)(((b >= 127) & (w == 0x00)) | ((b < 127) & (w == 0xFF)));
            }
            if (c > best_c[n]) { best_c[n] = c; best_p[n] = (uint32_t)i; }
        }
    }
    uint32_t corr = 0, word = 0;
    for (int n = 0; n < nwords; n++)
        if (best_c[n] > corr) { word = (uint32_t)n; corr = best_c[n]; }
    *word_out = word;
    *pos_out = best_p[word];
    *corr_out = corr;
}

/* Frame alignment and phase fix as the decoder does them between the correlator and Viterbi
 * (decoder/src/newdecoder.cpp:239-270): below the acceptance the chunk is skipped (:239-242); otherwise the
 * frame starts at the correlation position (the chunk is shifted down and `pos` more bytes are read, :245-258)
 * and, for the 180-degree word, PacketFixer::fixPacket(..., DEG_180, false) inverts every byte (:232,:265-267). */
void xo_sync_fix_frames(const int8_t *data, size_t n, const uint32_t *word, const uint32_t *pos, const uint32_t *corr,
                        uint32_t frame, uint32_t min_corr, int8_t *frames, uint8_t *valid)
{
    const size_t nf = n / frame;
    for (size_t f = 0; f < nf; f++) {
        const size_t src = f * frame + pos[f];
        int8_t *dst = frames + f * frame;
        if (corr[f] < min_corr || src + frame > n) {
            valid[f] = 0;
            for (uint32_t i = 0; i < frame; i++) dst[i] = 0;
            continue;
        }
        valid[f] = 1;
        for (uint32_t i = 0; i < frame; i++) {
            uint8_t b = (uint8_t)data[src + i];
            if (word[f] != 0) b ^= 0xFF;
            dst[i] = (int8_t)b;
        }
    }
}
