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
    // the cosine polynomial negated in quadrants 2 and 3; the sine's sign is in x
    const double c0 = negc ? -0x1p0 : 0x1p0;
    const double c1 = negc ? 0x1.ffffffd0c621cp-2 : -0x1.ffffffd0c621cp-2;
    const double c2 = negc ? -0x1.55553e1068f19p-5 : 0x1.55553e1068f19p-5;
    const double c3 = negc ? 0x1.6c087e89a359dp-10 : -0x1.6c087e89a359dp-10;
    const double c4 = negc ? -0x1.99343027bf8c3p-16 : 0x1.99343027bf8c3p-16;
    const double s1c = -0x1.555545995a603p-3, s2c = 0x1.1107605230bc4p-7, s3c = -0x1.994eb3774cf24p-13;
    const double x3 = x2 * x, x4 = x2 * x2;
    const double s1 = __builtin_fma(x2, s3c, s2c);
    const double cc2 = __builtin_fma(x2, c4, c3);
    const double cc1 = __builtin_fma(x2, c1, c0);
    const double x5 = x3 * x2, x6 = x4 * x2;
    const double s = __builtin_fma(x3, s1c, x);
    const double c = __builtin_fma(x4, c2, cc1);
    const float sv = (float)__builtin_fma(x5, s1, s);
    const float cv = (float)__builtin_fma(x6, cc2, c);
    sn = (n & 1) ? cv : sv;
    cs = (n & 1) ? sv : cv;
}

__device__ __forceinline__ void exact_sincosf(float y, float &sn, float &cs)
{
    const unsigned top = (__float_as_uint(y) >> 20) & 0x7ffu;
    const double x = (double)y;
    if (top < 0x3f4u) {                               // |y| < pi/4
        if (top < 0x398u) { sn = y; cs = 1.0f; return; }      // |y| < 2^-12
        exact_sincosf_poly(x, x * x, false, 0, sn, cs);
        return;
    }
    const double r = x * 0x1.45F306DC9C883p+23;       // 2/pi * 2^24
    const int n = ((int)r + 0x800000) >> 24;          // (int): towards zero, like cvttsd2si
    const double xr = __builtin_fma(-(double)n, 0x1.921FB54442D18p0, x);
    const double sg = ((n + 1) & 2) ? -1.0 : 1.0;     // sign[n & 3] = {1, -1, -1, 1}
    exact_sincosf_poly(xr * sg, xr * xr, (n & 2) != 0, n, sn, cs);
}

}  // namespace xrit
