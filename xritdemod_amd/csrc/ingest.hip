// ingest.hip -- RTL-SDR ingest: unsigned 8-bit IQ -> float IQ with the frontend's DC tracker.
// Replaces RtlFrontend::internalCallback (/root/reference/demodulator/src/RtlFrontend.cpp:102-116; table built at
// :26-28, alpha set at :57): every byte goes through lut[b] = (b - 128) * (1.f / 127.f), then
//        avg += alpha * (v - avg);  v -= avg;
// with ONE running average for the whole interleaved I/Q byte stream -- the reference's branch is `if (i % 1)`,
// which is never taken, so the "q" average is dead code and `iavg` sees I and Q alike.  That is what is restated.
// The recurrence is the affine map avg -> (1 - alpha) avg + alpha v per byte: a prefix scan of (a, b) pairs gives
// every thread its start value, the thread then replays the literal statements over its own bytes (scan.h).
#include "kernels.h"
#include "scan.h"

namespace xrit {

struct RtlScanF {
    typedef double2 T;                // (a, b): avg -> a * avg + b.  Double: 1 - alpha is 1 - 8e-6, which float32
                                      // holds to 0.4 % of alpha -- composed in float the average drifts from the serial one
    const unsigned char *in;
    float *out;                       // 2n floats (interleaved I, Q)
    const float *avg_in;
    float *avg_out;
    float alpha;
    long long nbytes;
    __device__ static float lut(unsigned char b) { return (float)((int)b - 128) * (1.f / 127.f); }
    __device__ T identity() const { return make_double2(1.0, 0.0); }
    __device__ T combine(const T &lo, const T &hi) const { return make_double2(hi.x * lo.x, hi.x * lo.y + hi.y); }
    __device__ T reduce_run(long long i0, int cnt) const
    {
        T m = identity();
        for (int k = 0; k < cnt; ++k) {
            const float v = lut(in[i0 + k]);
            // avg' = avg + alpha * (v - avg)
            m = combine(m, make_double2(1.0 - (double)alpha, (double)alpha * (double)v));
        }
        return m;
    }
    __device__ void apply_run(long long i0, int cnt, const T &pre) const
    {
        float avg = (float)(pre.x * (double)avg_in[0] + pre.y);
        for (int k = 0; k < cnt; ++k) {
            float v = lut(in[i0 + k]);
            avg += alpha * (v - avg);
            v -= avg;
            out[i0 + k] = v;
        }
        if (i0 + cnt == nbytes) avg_out[0] = avg;
    }
};

int RtlIngestStage::init(float sample_rate)
{
    // RtlFrontend::SetSampleRate, RtlFrontend.cpp:57 (double expression, stored as float)
    alpha = (float)(1.f - exp(-1.0 / (sample_rate * 0.05f)));
    XR_TRY(state.reserve(2 * sizeof(float)));
    XR_HIP(hipMemset(state.p, 0, 2 * sizeof(float)));
    cur = 0;
    return XRIT_OK;
}

int RtlIngestStage::reset(hipStream_t s)
{
    XR_HIP(hipMemsetAsync(state.p, 0, 2 * sizeof(float), s));
    cur = 0;
    return XRIT_OK;
}

void RtlIngestStage::release()
{
    state.release();
    aggs.release();
}

int RtlIngestStage::run(const void *in_u8, float2 *out, size_t n_complex, hipStream_t s, Profiler *prof)
{
    if (n_complex == 0) return XRIT_OK;
    const long long nb_ = (long long)n_complex * 2;
    const int nb = scan_blocks(nb_);
    XR_TRY(aggs.reserve((size_t)(nb + scan_blocks(nb) + 4) * sizeof(double2)));
    RtlScanF f{reinterpret_cast<const unsigned char *>(in_u8), reinterpret_cast<float *>(out), state.as<float>() + cur,
               state.as<float>() + (cur ^ 1), alpha, nb_};
    ProfScope ps(prof, "rtl_ingest", s);
    hipLaunchKernelGGL(scan_reduce_kernel<RtlScanF>, dim3(nb), dim3(SCAN_BLOCK), 0, s, f, nb_, aggs.as<double2>());
    scan_aggs_launch(f, aggs.as<double2>(), nb, s);
    hipLaunchKernelGGL(scan_apply_kernel<RtlScanF>, dim3(nb), dim3(SCAN_BLOCK), 0, s, f, nb_, aggs.as<double2>());
    XR_HIP(hipGetLastError());
    cur ^= 1;
    return XRIT_OK;
}

}  // namespace xrit
