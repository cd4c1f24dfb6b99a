#include <hip/hip_runtime.h>
#include <cstdio>
typedef float __attribute__((ext_vector_type(2))) f2;
// MODE 0: v_pk_fma all VGPR; 1: src0 = SGPR pair, op_sel_hi [0,1,1] (low half broadcast); 2: v_fma_f32 x2 with SGPR; 3: v_pk_fma with src0 SGPR pair plain
template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, float a, float b, int iters)
{
    f2 acc0 = {threadIdx.x * 0.001f, 1.f}, acc1 = acc0 + 1.f, acc2 = acc0 + 2.f, acc3 = acc0 + 3.f, acc4 = acc0 + 4.f, acc5 = acc0 + 5.f;
    f2 x = {threadIdx.x * 0.5f, 2.f};
    f2 t = {a, b};
    for (int it = 0; it < iters; ++it) {
#define ROUND(A) \
        if (MODE == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(A) : "v"(t), "v"(x)); \
        else if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(A) : "s"(t), "v"(x)); \
        else if (MODE == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(A) : "s"(t), "v"(x)); \
        else { asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(A.x) : "s"(a), "v"(x.x)); asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(A.y) : "s"(a), "v"(x.y)); }
#pragma unroll
        for (int u = 0; u < 8; ++u) { ROUND(acc0) ROUND(acc1) ROUND(acc2) ROUND(acc3) ROUND(acc4) ROUND(acc5) }
    }
    f2 s = acc0 + acc1 + acc2 + acc3 + acc4 + acc5;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}
int main()
{
    float *d;
    const int iters = 4000;
    hipMalloc(&d, 256 * 16 * 256 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps : {1, 2, 5}) {
        const int blocks = 256 * wps;   // 4 waves per block: wps waves per SIMD
        for (int mode = 0; mode < 4; ++mode) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) k<0><<<blocks, 256>>>(d, 1.0001f, 0.9999f, iters);
                if (mode == 1) k<1><<<blocks, 256>>>(d, 1.0001f, 0.9999f, iters);
                if (mode == 2) k<2><<<blocks, 256>>>(d, 1.0001f, 0.9999f, iters);
                if (mode == 3) k<3><<<blocks, 256>>>(d, 1.0001f, 0.9999f, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            double pairs = (double)blocks * 4 * iters * 48;   // wave-level "2-FMA" ops
            // cycles per wave-op per SIMD at 2.4 GHz
            double cyc = ms * 1e-3 * 2.4e9 / (pairs / (256 * 4));
            printf("waves/SIMD %d mode %d: %.3f ms, %.1f TFLOP/s, %.2f cyc (2.4 GHz) per packed pair\n", wps, mode, ms, pairs * 64 * 4 / ms / 1e9, cyc);
        }
    }
    return 0;
}
