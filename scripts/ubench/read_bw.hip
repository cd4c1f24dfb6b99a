// read_bw.hip -- what read-only bandwidth a hand-written sweep reaches on this part, by access pattern.
//   hipcc --offload-arch=gfx950 -O3 -o read_bw read_bw.hip && ./read_bw
// Patterns: which 16-byte word a (block, thread, iteration) reads.
//   tile    block b owns one contiguous tile of T bytes (the FIR kernels' shape), UNR loads in flight per thread
//   stride  grid-stride over the whole buffer with a resident grid
//   xcd     like tile, but the tiles of one XCD (blockIdx % 8) are contiguous in memory
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float __attribute__((ext_vector_type(4))) f4;

template <int UNR, bool NT>
__global__ __launch_bounds__(256) void tile_kernel(const f4 *__restrict__ in, float *out, size_t words_per_block, int xcd_major, int nblk, int shift = 0, float2 *st = nullptr, int coalesced = 0)
{
    size_t b = blockIdx.x;
    if (xcd_major) {
        const size_t per = (size_t)nblk / 8;
        b = (b % 8) * per + b / 8;
    }
    const f4 *p = in + b * words_per_block + shift;
    f4 acc = {0, 0, 0, 0};
    for (size_t i = threadIdx.x; i < words_per_block; i += 256 * UNR) {
        f4 v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const f4 *q = p + i + (size_t)u * 256;
            v[u] = NT ? __builtin_nontemporal_load(q) : *q;
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) acc += v[u];
    }
    if (st) {   // the FIR's output pattern: 3 x 8 bytes per lane, lanes 24 bytes apart, a fifth of the bytes read
        float2 *q = st + (b * 256 + threadIdx.x) * 3 * (words_per_block / 1920);
        if (coalesced) {
            q = st + b * 256 * 3 * (words_per_block / 1920) + threadIdx.x;
            for (size_t j = 0; j < 3 * (words_per_block / 1920); ++j) q[j * 256] = make_float2(acc.x, acc.y + j);
        } else
        for (size_t j = 0; j < 3 * (words_per_block / 1920); ++j) q[j] = make_float2(acc.x, acc.y + j);
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = 1.f;
}

template <int UNR, bool NT>
__global__ __launch_bounds__(256) void stride_kernel(const f4 *__restrict__ in, float *out, size_t words)
{
    const size_t step = (size_t)gridDim.x * 256 * UNR;
    f4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 * UNR + threadIdx.x; i < words; i += step) {
        f4 v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const f4 *q = in + i + (size_t)u * 256;
            v[u] = NT ? __builtin_nontemporal_load(q) : *q;
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) acc += v[u];
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = 1.f;
}

template <typename F> static double time_ms(F f, int reps)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main(int argc, char **argv)
{
    const size_t bytes = (argc > 1 ? (size_t)atoll(argv[1]) : (size_t)2048) << 20;
    const size_t words = bytes / 16;
    f4 *in; float *out;
    CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, 64));
    CK(hipMemset(in, 0, bytes));
    printf("buffer %zu MiB\n", bytes >> 20);
    for (size_t tile_kb : {32, 64, 128, 256, 1024}) {
        const size_t wpb = tile_kb * 1024 / 16;
        const int nblk = (int)(words / wpb);
        for (int xm = 0; xm < 2; ++xm) {
#define RUN(UNR, NT) { double ms = time_ms([&] { tile_kernel<UNR, NT><<<nblk, 256>>>(in, out, wpb, xm, nblk); }, 10); \
            printf("tile %5zu KiB %s unr %d %s: %6.0f GB/s\n", tile_kb, xm ? "xcd-major" : "linear   ", UNR, NT ? "nt" : "  ", bytes / ms / 1e6); }
            RUN(1, false) RUN(2, false) RUN(4, false) RUN(8, false) RUN(4, true)
#undef RUN
        }
    }
    {   // the decimator's shape: 30 KiB tiles (3840 samples), window start 48 bytes past a line, strided stores
        const size_t wpb = 1920;
        const int nblk = (int)(words / wpb) - 1;
        float2 *st; CK(hipMalloc(&st, (size_t)nblk * 256 * 3 * 8 + 4096));
        for (int shift : {3}) for (int store = 0; store < 2; ++store) for (int ldskb : {0, 32, 40, 53, 80}) {
#define RUN(UNR, NT) { double ms = time_ms([&] { tile_kernel<UNR, NT><<<nblk, 256, ldskb * 1024>>>(in, out, wpb, 0, nblk, shift, store ? st : nullptr, store == 2); }, 10); \
            printf("fir-shape lds %2d KiB store %d unr %d %s: %6.0f GB/s read\n", ldskb, store, UNR, NT ? "nt" : "  ", (double)nblk * wpb * 16 / ms / 1e6); }
            RUN(8, false)
#undef RUN
        }
    }
    for (int wpc : {4, 8, 16}) {
        const int grid = 256 * wpc;
#define RUN(UNR, NT) { double ms = time_ms([&] { stride_kernel<UNR, NT><<<grid, 256>>>(in, out, words); }, 10); \
        printf("stride grid %5d unr %d %s: %6.0f GB/s\n", grid, UNR, NT ? "nt" : "  ", bytes / ms / 1e6); }
        RUN(1, false) RUN(2, false) RUN(4, false) RUN(8, false) RUN(4, true)
#undef RUN
    }
    return 0;
}
