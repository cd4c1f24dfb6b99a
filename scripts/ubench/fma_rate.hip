// Micro-benchmark (dev aid): issue rate of v_fma_f32 vs v_pk_fma_f32 on gfx950, to know what the FIR
// kernels' 48 packed FMAs per 16 samples can reach at best.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, float a, float b, int iters)
{
    float2 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = make_float2(threadIdx.x * 0.001f + i, i * 0.5f);
    float2 x = make_float2(a, b);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) {   // packed: acc.xy = t * x.xy + acc.xy
                acc[i].x = fmaf(a, x.x, acc[i].x);
                acc[i].y = fmaf(a, x.y, acc[i].y);
            } else {           // scalar, prevented from packing by using different multipliers
                acc[i].x = fmaf(a, acc[i].x, b);
                acc[i].y = fmaf(b, acc[i].y, a);
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main()
{
    float *d;
    const int blocks = 256 * 8, iters = 20000;
    hipMalloc(&d, blocks * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.9999f, iters);
            else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.9999f, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double flop = (double)blocks * 256 * iters * 16 * 2;
            if (rep) printf("mode %d (%s): %.3f ms, %.1f TFLOP/s\n", mode, mode == 0 ? "packable" : "scalar", ms, flop / ms / 1e9);
        }
    }
    return 0;
}
