// misc.hip -- small kernels around the chain: the int8 soft-symbol quantiser of
// SymbolManager::process (/root/reference/demodulator/src/SymbolManager.cpp:43-46),
// the ingest conversion of demodulator.cpp:54-74 for the decimation==1 case, and
// the synthetic burst generator that stands in for the cf32 capture file of
// CFileFrontend.cpp:34-56 (formulae: xritdemod_amd/synth.py).
#include "kernels.h"

namespace xrit {

__global__ void quantize_i8_kernel(const float *__restrict__ in, int8_t *__restrict__ out, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float f = in[i] * 127;
    f = f > 127 ? 127 : f;
    f = f < -128 ? -128 : f;
    out[i] = (int8_t)(int)f;   // C cast: truncation toward zero
}

int launch_quantize_i8(const float *in, int8_t *out, size_t n, hipStream_t s)
{
    if (n == 0) return XRIT_OK;
    hipLaunchKernelGGL(quantize_i8_kernel, dim3(div_up(n, 256)), dim3(256), 0, s, in, out, n);
    XR_HIP(hipGetLastError());
    return XRIT_OK;
}

template <int TYPE>
__global__ void convert_kernel(const void *__restrict__ in, float2 *__restrict__ out, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (TYPE == XRIT_SAMPLE_S16IQ) {
        short2 v = reinterpret_cast<const short2 *>(in)[i];
        out[i] = make_float2(v.x / 32768.f, v.y / 32768.f);
    } else if (TYPE == XRIT_SAMPLE_S8IQ) {
        char2 v = reinterpret_cast<const char2 *>(in)[i];
        out[i] = make_float2(v.x / 128.f, v.y / 128.f);
    } else {
        out[i] = reinterpret_cast<const float2 *>(in)[i];
    }
}

int launch_convert(const void *in, int type, float2 *out, size_t n, hipStream_t s)
{
    if (n == 0) return XRIT_OK;
    dim3 g(div_up(n, 256)), b(256);
    if (type == XRIT_SAMPLE_S16IQ) hipLaunchKernelGGL(convert_kernel<XRIT_SAMPLE_S16IQ>, g, b, 0, s, in, out, n);
    else if (type == XRIT_SAMPLE_S8IQ) hipLaunchKernelGGL(convert_kernel<XRIT_SAMPLE_S8IQ>, g, b, 0, s, in, out, n);
    else hipLaunchKernelGGL(convert_kernel<XRIT_SAMPLE_FLOATIQ>, g, b, 0, s, in, out, n);
    XR_HIP(hipGetLastError());
    return XRIT_OK;
}

// ------------------------------------------------------------------ synth
__device__ __forceinline__ uint64_t synth_hash(uint64_t seed, uint64_t counter)
{
    uint64_t z = seed + counter * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ __forceinline__ double synth_rrc(double t, double a)
{
    const double pi = XR_PI_D;
    if (fabs(t) < 1e-9) return 1.0 - a + 4.0 * a / pi;
    if (fabs(fabs(4.0 * a * t) - 1.0) < 1e-7)
        return (a / sqrt(2.0)) * ((1.0 + 2.0 / pi) * sin(pi / (4.0 * a)) + (1.0 - 2.0 / pi) * cos(pi / (4.0 * a)));
    double q = 4.0 * a * t;
    return (sin(pi * t * (1.0 - a)) + q * cos(pi * t * (1.0 + a))) / (pi * t * (1.0 - q * q));
}

__global__ void __launch_bounds__(256) synth_kernel(xrit_synth_params p, uint64_t start, size_t n, float2 *__restrict__ out,
                                                    double rate, double sigma, double dphi)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t idx = (int64_t)(start + i);
    const double u = (double)idx * rate - p.timing_offset;
    const double k0 = floor(u);
    const int64_t k0i = (int64_t)k0;
    double acc = 0.0;
    for (int j = -15; j <= 16; ++j) {
        int64_t k = k0i + j;
        uint64_t h = synth_hash(p.seed, (uint64_t)k);
        double bit = (h >> 63) ? 1.0 : -1.0;
        acc += bit * synth_rrc(u - (double)k, p.alpha);
    }
    double ph = dphi * (double)idx + p.phase0;
    double sn, cs;
    sincos(ph, &sn, &cs);
    double re = p.amplitude * acc * cs, im = p.amplitude * acc * sn;
    if (sigma > 0.0) {
        uint64_t h = synth_hash(p.seed + 1, (uint64_t)idx);
        double u1 = ((double)(h >> 40) + 0.5) / 16777216.0;
        double u2 = ((double)(h & 0xFFFFFFull) + 0.5) / 16777216.0;
        double r = sigma * sqrt(-log(u1));
        double s2, c2;
        sincos(2.0 * XR_PI_D * u2, &s2, &c2);
        re += r * c2;
        im += r * s2;
    }
    out[i] = make_float2((float)re, (float)im);
}

int launch_synth(const xrit_synth_params &p, uint64_t start, size_t n, float2 *out, hipStream_t s)
{
    if (n == 0) return XRIT_OK;
    double rate = p.symbol_rate * (1.0 + p.clock_ppm * 1e-6) / p.fs_in;
    double sigma = 0.0;
    if (p.esn0_db > -900.0)
        sigma = sqrt(p.amplitude * p.amplitude * (p.fs_in / p.symbol_rate) / pow(10.0, p.esn0_db / 10.0));
    double dphi = 2.0 * XR_PI_D * p.carrier_hz / p.fs_in;
    hipLaunchKernelGGL(synth_kernel, dim3(div_up(n, 256)), dim3(256), 0, s, p, start, n, out, rate, sigma, dphi);
    XR_HIP(hipGetLastError());
    return XRIT_OK;
}

}  // namespace xrit
