// loop_core.h -- per-sample / per-symbol arithmetic of the three feedback loops,
// written once and used by every kernel lane.  The statement order follows the
// algorithms SatHelper's AGC / CostasLoop / ClockRecovery implement (call sites
// /root/reference/demodulator/src/demodulator.cpp:143,152,156; parameters
// Parameters.h:27-37).  Build with -ffp-contract=off: these recurrences use
// separate multiplies and adds, like the reference's x86-64 -O3 build, so that a
// lane's trajectory tracks the CPU one as closely as float32 allows.
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define XR_HD __host__ __device__ __forceinline__
#else
#define XR_HD inline
#endif

namespace xrit {

struct cf32 { float x, y; };

#define XR_TWOPI_F 6.28318530717958647692f
#define XR_PI_D    3.14159265358979323846

// 0.5*(|x+c| - |x-c|): the "branchless clip" of the upstream loops.  Its float
// roundings are part of the recurrence (it quantises small x to ulp(c)), so it
// must not be replaced by fmin/fmax.
XR_HD float bclip(float x, float c)
{
    float a = fabsf(x + c);
    float b = fabsf(x - c);
    a -= b;
    return 0.5f * a;
}

// ------------------------------------------------------------------ AGC ----
// g' = min(a*g + b, c) maps; composition is associative for a >= 0.
struct AgcMap { float a, b, c; };

XR_HD AgcMap agc_identity() { return AgcMap{1.0f, 0.0f, INFINITY}; }

// first `lo`, then `hi`
XR_HD AgcMap agc_compose(const AgcMap &lo, const AgcMap &hi)
{
    AgcMap r;
    r.a = hi.a * lo.a;
    r.b = hi.a * lo.b + hi.b;
    r.c = fminf(hi.a * lo.c + hi.b, hi.c);
    return r;
}

XR_HD float agc_apply(const AgcMap &m, float g) { return fminf(m.a * g + m.b, m.c); }

// |x| on the device: v_sqrt_f32 (1 ulp) instead of the ~16-instruction correctly rounded sqrtf.  The matched
// filter's AGC fill takes two square roots per sample and was VALU bound on them (1256 VALU per wave, a quarter of
// it sqrtf refinement); the AGC loop is contractive, so a last-bit difference in |x| does not accumulate (stage
// parity against the oracle unchanged at ~1e-6).
XR_HD float agc_sqrt(float v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sqrtf(v);
#else
    return sqrtf(v);
#endif
}

// map of one sample: g += rate*(ref - |x| g); clamp to max (max<=0: no clamp)
XR_HD AgcMap agc_sample_map(float xr, float xi, float rate, float ref, float maxg)
{
    float mag = agc_sqrt(xr * xr + xi * xi);
    AgcMap m;
    m.a = 1.0f - rate * mag;
    m.b = rate * ref;
    m.c = maxg > 0.0f ? maxg : INFINITY;
    return m;
}

// exact serial step, same statement order as the CPU chain
XR_HD void agc_step(float xr, float xi, float &g, float rate, float ref, float maxg, float &yr, float &yi)
{
    yr = xr * g;
    yi = xi * g;
    g += rate * (ref - agc_sqrt(yr * yr + yi * yi));
    if (maxg > 0.0f && g > maxg) g = maxg;
}

// --------------------------------------------------------------- Costas ----
struct CostasGains { float alpha, beta; };

XR_HD CostasGains costas_gains(float loop_bw)
{
    float damping = sqrtf(2.0f) / 2.0f;
    float denom = (1.0f + 2.0f * damping * loop_bw + loop_bw * loop_bw);
    CostasGains g;
    g.alpha = (4 * damping * loop_bw) / denom;
    g.beta = (4 * loop_bw * loop_bw) / denom;
    return g;
}

// sin and cos of a loop phase (|x| stays within a few multiples of pi).
XR_HD void loop_sincos(float x, float &s, float &c)
{
#if defined(__HIP_DEVICE_COMPILE__) && defined(XRIT_ACCURATE_SINCOS)
    // (make EXTRA=-DXRIT_ACCURATE_SINCOS, scripts/r4_floor_vs_frontend.py: the math library's sincosf, ~1 ulp, ~25 instructions)
    ::sincosf(x, &s, &c);
#elif defined(__HIP_DEVICE_COMPILE__)
    // v_sin_f32 / v_cos_f32 take turns (|t| <= 256; here |x| <= 2 pi + 1.5).  Quarter rate, but three
    // instructions against ~25 for a Cody-Waite reduction with two polynomials -- the Costas passes are VALU
    // bound and this call was a third of their per-sample work (measured: -16 % per pass).  Absolute error
    // ~2e-7, the size of the float32 rounding of the phase itself; parity with the oracle's libm sincosf is
    // unchanged (Costas stage 1.2e-6 rms on the smoke burst before and after).
    const float t = x * 0.15915494309189533577f;
    s = __builtin_amdgcn_sinf(t);
    c = __builtin_amdgcn_cosf(t);
#else
    ::sincosf(x, &s, &c);
#endif
}

// tangent of (phase, freq) w.r.t. the chain's start (phase0, freq0)
struct CostasTan { float pp, pf, fp, ff; };

template <bool TANGENT>
XR_HD void costas_step(float zr, float zi, float &phase, float &freq, const CostasGains &g,
                       float &yr, float &yi, CostasTan &t)
{
    float s, c;
    loop_sincos(-phase, s, c);
    yr = zr * c - zi * s;
    yi = zr * s + zi * c;
    float err = yr * yi;
    float ed = 0.0f;
    if (TANGENT) ed = (fabsf(err) < 1.0f) ? (yi * yi - yr * yr) : 0.0f;
    err = bclip(err, 1.0f);
    freq = freq + g.beta * err;
    phase = phase + freq + g.alpha * err;
    if (TANGENT) {
        float be = g.beta * ed, ae = g.alpha * ed;
        t.fp = t.fp + be * t.pp;
        t.ff = t.ff + be * t.pf;
        float npp = t.pp + t.fp + ae * t.pp;
        float npf = t.pf + t.ff + ae * t.pf;
        t.pp = npp;
        t.pf = npf;
    }
    // phase_wrap(): the upstream while-loops run at most once per sample here, because a step moves the phase
    // by less than |freq| + alpha <= 1.02 rad and costas_prewrap() has put the start value in range
    phase = phase > XR_TWOPI_F ? phase - XR_TWOPI_F : phase;
    phase = phase < -XR_TWOPI_F ? phase + XR_TWOPI_F : phase;
    // frequency_limit()
    const bool clamped = (freq > 1.0f) | (freq < -1.0f);
    freq = fminf(fmaxf(freq, -1.0f), 1.0f);
    if (TANGENT) {
        t.fp = clamped ? 0.0f : t.fp;
        t.ff = clamped ? 0.0f : t.ff;
    }
}

// bring a start phase (a guess may sit a few pi away) into the range the per-sample wrap assumes
XR_HD float costas_prewrap(float phase)
{
    while (phase > XR_TWOPI_F + 1.5f) phase -= XR_TWOPI_F;
    while (phase < -XR_TWOPI_F - 1.5f) phase += XR_TWOPI_F;
    return phase;
}

// ---------------------------------------------------- Mueller & Mueller ----
#define XR_MM_NTAPS  8
#define XR_MM_NSTEPS 128
#define XR_MM_FUDGE  16

struct ClockPar { float omega_mid, omega_lim, gain_omega, gain_mu; };

// state between symbols.  ii is the absolute read index into the call's
// [carry | new] sample buffer.
struct ClockState {
    int64_t ii;
    float mu, omega;
    cf32 p0, p1;   // last two interpolated outputs (p_0T, p_1T after the step)
    cf32 c0, c1;   // their 0/1 slicer decisions
};

// The 8-tap interpolation, acc += tap * sample with separately rounded multiply and add (the build has
// -ffp-contract=off, like the CPU chain).  On the device the (re, im) pair is one 2-vector, i.e. v_pk_mul_f32 +
// v_pk_add_f32 per tap; the loop kernels are compiled without SLP vectorisation, which otherwise pairs up
// unrelated scalars of the M&M update and pays for it in register moves.
#if defined(__HIP_DEVICE_COMPILE__)
#define XR_MM_INTERPOLATE(ROW, W, AR, AI)                                        \
    do {                                                                         \
        typedef float __attribute__((ext_vector_type(2))) xr_f2_;                \
        xr_f2_ a_;                                                               \
        _Pragma("unroll") for (int k_ = 0; k_ < XR_MM_NTAPS; ++k_) {             \
            const float tp_ = (float)(ROW)[XR_MM_NTAPS - 1 - k_];                \
            const cf32 v_ = (W)[k_];                                             \
            const xr_f2_ vv_ = {v_.x, v_.y};                                     \
            a_ = k_ == 0 ? vv_ * tp_ : a_ + vv_ * tp_;   /* 0 + p == p */        \
        }                                                                        \
        AR = a_.x;                                                               \
        AI = a_.y;                                                               \
    } while (0)
#else
#define XR_MM_INTERPOLATE(ROW, W, AR, AI)                                        \
    do {                                                                         \
        for (int k_ = 0; k_ < XR_MM_NTAPS; ++k_) {                               \
            const float tp_ = (float)(ROW)[XR_MM_NTAPS - 1 - k_];                \
            const cf32 v_ = (W)[k_];                                             \
            AR += tp_ * v_.x;                                                    \
            AI += tp_ * v_.y;                                                    \
        }                                                                        \
    } while (0)
#endif

// The step in two halves, so that a kernel which gets a symbol's history from somewhere else (clock_relay.h:
// the two previous symbols sit in neighbouring lanes) runs the very same float operations.
// First half: the interpolated sample at (window, mu).
// (the interpolator arm given: clock_relay.h keeps it next to the read index)
template <typename TableT>
XR_HD cf32 clock_interp_arm(const cf32 *w, const TableT *table, int imu)
{
    const TableT *row = table + imu * XR_MM_NTAPS;
    float ar = 0.0f, ai = 0.0f;
    XR_MM_INTERPOLATE(row, w, ar, ai);
    return cf32{ar, ai};
}
template <typename TableT>
XR_HD cf32 clock_interp(const cf32 *w, const TableT *table, float mu_now, int *arm_out = nullptr)
{
    int imu = (int)rintf(mu_now * (float)XR_MM_NSTEPS);
    if (arm_out) *arm_out = imu;
    return clock_interp_arm(w, table, imu);
}

// Second half, again in two: the Mueller & Mueller timing error of the symbol p0 given the history in s ...
XR_HD float clock_timing_error(const cf32 p0, const ClockState &s)
{
    cf32 p2 = s.p1, p1 = s.p0;
    cf32 c2 = s.c1, c1 = s.c0;
    cf32 c0{p0.x > 0.0f ? 1.0f : 0.0f, p0.y > 0.0f ? 1.0f : 0.0f};
    float dcr = c0.x - c2.x, dci = c0.y - c2.y;
    float xr = dcr * p1.x + dci * p1.y;
    float dpr = p0.x - p2.x, dpi = p0.y - p2.y;
    float yr = dpr * c1.x + dpi * c1.y;
    float mm = yr - xr;
    return bclip(mm, 1.0f);
}

// ... and the loop filters: omega and mu move, s.ii advances by floor(mu), the history shifts.
XR_HD void clock_advance(float mm, const cf32 p0, ClockState &s, const ClockPar &par)
{
    float omega = s.omega + par.gain_omega * mm;
    omega = par.omega_mid + bclip(omega - par.omega_mid, par.omega_lim);
    float mu = s.mu + omega + par.gain_mu * mm;
    float fl = floorf(mu);
    s.ii += (int)fl;            // |mu + omega| is a few samples: the 32-bit conversion is exact and one instruction
    s.mu = mu - fl;
    s.omega = omega;
    s.p1 = s.p0; s.p0 = p0;
    s.c1 = s.c0; s.c0 = cf32{p0.x > 0.0f ? 1.0f : 0.0f, p0.y > 0.0f ? 1.0f : 0.0f};
}

XR_HD void clock_update(const cf32 p0, ClockState &s, const ClockPar &par)
{
    const float mm = clock_timing_error(p0, s);
    clock_advance(mm, p0, s, par);
}

// One symbol.  w points at the 8-sample window x[ii .. ii+7] (global memory or an
// LDS copy of it), table at the 129x8 MMSE taps.
template <typename TableT>
XR_HD cf32 clock_step_w(const cf32 *w, const TableT *table, ClockState &s, const ClockPar &par, int *arm_out = nullptr)
{
    const cf32 p0 = clock_interp(w, table, s.mu, arm_out);
    clock_update(p0, s, par);
    return p0;
}

// Same step with the read index kept as a 32-bit offset into a staged window (row points at the
// window's first sample): no 64-bit arithmetic on the per-symbol path.
template <typename TableT>
XR_HD cf32 clock_step_rel(const cf32 *row, int &off, const TableT *table, ClockState &s, const ClockPar &par)
{
    cf32 p2 = s.p1, p1 = s.p0;
    cf32 c2 = s.c1, c1 = s.c0;
    int imu = (int)rintf(s.mu * (float)XR_MM_NSTEPS);
    const TableT *trow = table + imu * XR_MM_NTAPS;
    const cf32 *w = row + off;
    float ar = 0.0f, ai = 0.0f;
    XR_MM_INTERPOLATE(trow, w, ar, ai);
    cf32 p0{ar, ai};
    cf32 c0{p0.x > 0.0f ? 1.0f : 0.0f, p0.y > 0.0f ? 1.0f : 0.0f};
    float dcr = c0.x - c2.x, dci = c0.y - c2.y;
    float xr = dcr * p1.x + dci * p1.y;
    float dpr = p0.x - p2.x, dpi = p0.y - p2.y;
    float yr = dpr * c1.x + dpi * c1.y;
    float mm = yr - xr;
    mm = bclip(mm, 1.0f);
    float omega = s.omega + par.gain_omega * mm;
    omega = par.omega_mid + bclip(omega - par.omega_mid, par.omega_lim);
    float mu = s.mu + omega + par.gain_mu * mm;
    float fl = floorf(mu);
    off += (int)fl;
    s.mu = mu - fl;
    s.omega = omega;
    s.p1 = p1; s.p0 = p0;
    s.c1 = c1; s.c0 = c0;
    return p0;
}

// x points at the sample buffer base
template <typename TableT>
XR_HD cf32 clock_step(const cf32 *x, const TableT *table, ClockState &s, const ClockPar &par, int *arm_out = nullptr)
{
    return clock_step_w(x + s.ii, table, s, par, arm_out);
}

// t += dt on the (ii, mu) pair
XR_HD void clock_shift(ClockState &s, float dt)
{
    float m = s.mu + dt;
    float fl = floorf(m);
    s.ii += (int64_t)fl;
    s.mu = m - fl;
}

XR_HD float clock_tdiff(const ClockState &a, const ClockState &b)
{
    return (float)(a.ii - b.ii) + (a.mu - b.mu);
}

}  // namespace xrit
