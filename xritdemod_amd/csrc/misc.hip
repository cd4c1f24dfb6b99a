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

// ------------------------------------------------------------ measurement
// What a hand-written read-only sweep reaches on this device (bench.py reports it next to the vendor peak, as
// SURVEY.md section 8(d) asks): every workgroup owns contiguous 64 KiB tiles, 16-byte non-temporal loads, eight in
// flight per thread -- the best of the patterns scripts/ubench/read_bw.hip compares (7.0-7.1 TB/s on MI355X).
__global__ void __launch_bounds__(256) read_bw_kernel(const float4 *__restrict__ in, size_t words, float *sink)
{
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f *p = reinterpret_cast<const v4f *>(in);
    const size_t tile = 4096;                                  // 16-byte words per tile
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t t0 = (size_t)blockIdx.x * tile; t0 + tile <= words; t0 += (size_t)gridDim.x * tile) {
        for (size_t i = threadIdx.x; i < tile; i += 256 * 8) {
            v4f v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(p + t0 + i + (size_t)u * 256);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = 1.f;
}

int launch_read_bw(const void *buf, size_t bytes, int reps, hipStream_t s, double *gbs)
{
    *gbs = 0;
    const size_t words = bytes / 16 / 4096 * 4096;
    if (words == 0 || reps < 1) { set_error("read bandwidth probe: buffer too small"); return XRIT_E_INVALID; }
    float *sink = nullptr;
    XR_HIP(hipMalloc((void **)&sink, 64));
    hipEvent_t a, b;
    XR_HIP(hipEventCreate(&a));
    XR_HIP(hipEventCreate(&b));
    const unsigned blocks = (unsigned)(words / 4096 < 8192 ? words / 4096 : 8192);
    hipLaunchKernelGGL(read_bw_kernel, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const float4 *>(buf), words, sink);
    XR_HIP(hipEventRecord(a, s));
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL(read_bw_kernel, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const float4 *>(buf), words, sink);
    XR_HIP(hipEventRecord(b, s));
    XR_HIP(hipEventSynchronize(b));
    float ms = 0;
    XR_HIP(hipEventElapsedTime(&ms, a, b));
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    (void)hipFree(sink);
    if (ms > 0) *gbs = (double)words * 16.0 * reps / (ms * 1e-3) / 1e9;
    return XRIT_OK;
}

}  // namespace xrit
