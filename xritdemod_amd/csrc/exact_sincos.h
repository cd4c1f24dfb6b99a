// exact_sincos.h -- the Costas loop's sincosf, bit for bit the CPU chain's.
// The reference's loop (SatHelper::CostasLoop::Work, /root/reference/demodulator/src/demodulator.cpp:152) takes the sine and
// cosine of its phase from the C library; on x86-64 with FMA that is glibc's __sincosf_fma: the argument is reduced and two
// polynomials are evaluated in DOUBLE precision with fused multiply-adds, the results rounded to float once.  IEEE double
// arithmetic is the same on the device (v_mul_f64 / v_fma_f64 / v_cvt_*), so the very same operations in the same order give
// the same bits: xo_sincosf of the test tier's CPU restatement is the operation-for-operation restatement this follows (checked against
// the C library for every float below 120 there); tests/test_gpu_exact.py checks the device against it.
// Valid for |y| < 120 -- a loop phase is wrapped to +-2 pi.
#pragma once

#include <hip/hip_runtime.h>

namespace xrit {

__device__ __forceinline__ void exact_sincosf_poly(double x, double x2, bool negc, int n, float &sn, float &cs)
{
    // In quadrants 2 and 3 the C library takes a table whose cosine coefficients are all negated (the sine's sign is in x).
    // Negating every coefficient negates every fused multiply-add's exact result, and rounding is symmetric: the polynomial is
    // evaluated once with the positive coefficients and its rounded value negated -- the same bits.
    const double c0 = 0x1p0, c1 = -0x1.ffffffd0c621cp-2, c2 = 0x1.55553e1068f19p-5, c3 = -0x1.6c087e89a359dp-10,
                 c4 = 0x1.99343027bf8c3p-16;
    const double s1c = -0x1.555545995a603p-3, s2c = 0x1.1107605230bc4p-7, s3c = -0x1.994eb3774cf24p-13;
    const double x3 = x2 * x, x4 = x2 * x2;
    const double s1 = __builtin_fma(x2, s3c, s2c);
    const double cc2 = __builtin_fma(x2, c4, c3);
    const double cc1 = __builtin_fma(x2, c1, c0);
    const double x5 = x3 * x2, x6 = x4 * x2;
    const double s = __builtin_fma(x3, s1c, x);
    const double c = __builtin_fma(x4, c2, cc1);
    const float sv = (float)__builtin_fma(x5, s1, s);
    float cv = (float)__builtin_fma(x6, cc2, c);
    cv = negc ? -cv : cv;
    sn = (n & 1) ? cv : sv;
    cs = (n & 1) ? sv : cv;
}

__device__ __forceinline__ void exact_sincosf(float y, float &sn, float &cs)
{
    // One straight line for every |y| < 120 (the lanes of a wave sit on both sides of pi/4 now and then: no divergence).  The C
    // library's first branch (|y| < pi/4: the polynomials on y itself) is the general one with n = 0 -- there
    // n = ((int)(y 2^24 2/pi) + 2^23) >> 24 is 0, the reduction fma(-0, pi/2, y) is y exactly, the sign 1, the table the first --;
    // its second (|y| < 2^-12: sin = y, cos = 1) is a select behind it.
    const unsigned top = (__float_as_uint(y) >> 20) & 0x7ffu;
    const double x = (double)y;
    const double r = x * 0x1.45F306DC9C883p+23;       // 2/pi * 2^24
    const int n = ((int)r + 0x800000) >> 24;          // (int): towards zero, like cvttsd2si
    const double xr = __builtin_fma(-(double)n, 0x1.921FB54442D18p0, x);
    const double sg = ((n + 1) & 2) ? -1.0 : 1.0;     // sign[n & 3] = {1, -1, -1, 1}
    float s1, c1;
    exact_sincosf_poly(xr * sg, xr * xr, (n & 2) != 0, n, s1, c1);
    const bool tiny = top < 0x398u;                   // |y| < 2^-12
    sn = tiny ? y : s1;
    cs = tiny ? 1.0f : c1;
}

}  // namespace xrit
