// graph_floor.hip -- what a chain of tiny dependent kernels costs per kernel: stream launches vs one hipGraph.
//   hipcc --offload-arch=gfx950 -O3 -o graph_floor graph_floor.hip && ./graph_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void tiny(int *p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1; }

int main()
{
    int *d;
    CK(hipMalloc(&d, 1 << 20));
    CK(hipMemset(d, 0, 1 << 20));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int NK = 64, REPS = 50;
    for (int blocks : {1, 64, 512}) {
        // plain launches
        for (int w = 0; w < 3; ++w) { for (int k = 0; k < NK; ++k) tiny<<<blocks, 256, 0, s>>>(d, blocks * 256); CK(hipStreamSynchronize(s)); }
        auto t0 = std::chrono::steady_clock::now();
        CK(hipEventRecord(a, s));
        for (int r = 0; r < REPS; ++r)
            for (int k = 0; k < NK; ++k) tiny<<<blocks, 256, 0, s>>>(d, blocks * 256);
        CK(hipEventRecord(b, s));
        CK(hipEventSynchronize(b));
        auto t1 = std::chrono::steady_clock::now();
        float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
        printf("blocks %4d  stream launches: %.2f us per kernel on the GPU, %.2f us wall\n", blocks, ms * 1e3 / (REPS * NK),
               std::chrono::duration<double, std::micro>(t1 - t0).count() / (REPS * NK));
        // graph of NK kernels
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int k = 0; k < NK; ++k) tiny<<<blocks, 256, 0, s>>>(d, blocks * 256);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int w = 0; w < 3; ++w) { CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s)); }
        t0 = std::chrono::steady_clock::now();
        CK(hipEventRecord(a, s));
        for (int r = 0; r < REPS; ++r) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(b, s));
        CK(hipEventSynchronize(b));
        t1 = std::chrono::steady_clock::now();
        CK(hipEventElapsedTime(&ms, a, b));
        printf("blocks %4d  hipGraph (%d nodes): %.2f us per kernel on the GPU, %.2f us wall\n", blocks, NK, ms * 1e3 / (REPS * NK),
               std::chrono::duration<double, std::micro>(t1 - t0).count() / (REPS * NK));
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
