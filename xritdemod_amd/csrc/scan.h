// scan.h -- three-kernel prefix scan over a non-commutative monoid (affine maps).
// Used by the AGC gain recurrence and by the chain-boundary hand-off solves of
// the Costas and clock-recovery loops.
//
// A functor F supplies
//     typedef ... T;                                  the monoid element (POD)
//     __device__ T    identity() const;
//     __device__ T    combine(const T& lo, const T& hi) const;   lo happens first
//     __device__ T    reduce_run(long long i0, int cnt) const;   compose elements [i0, i0+cnt)
//     __device__ void apply_run(long long i0, int cnt, const T& prefix_excl) const;
// Each thread owns SCAN_IPT consecutive elements; a block owns SCAN_BLOCK*SCAN_IPT.
#pragma once

#include <hip/hip_runtime.h>

namespace xrit {

constexpr int SCAN_BLOCK = 256;
constexpr int SCAN_IPT = 4;
constexpr int SCAN_TILE = SCAN_BLOCK * SCAN_IPT;

// Inclusive Hillis-Steele scan across the block; buf has SCAN_BLOCK entries.
template <typename F>
__device__ __forceinline__ typename F::T block_scan_inclusive(const F &f, typename F::T v, typename F::T *buf)
{
    const int t = threadIdx.x;
    buf[t] = v;
    __syncthreads();
    for (int off = 1; off < SCAN_BLOCK; off <<= 1) {
        typename F::T lo = f.identity();
        bool has = t >= off;
        if (has) lo = buf[t - off];
        __syncthreads();
        if (has) {
            v = f.combine(lo, v);
            buf[t] = v;
        }
        __syncthreads();
    }
    return v;
}

template <typename F>
__global__ void __launch_bounds__(SCAN_BLOCK) scan_reduce_kernel(F f, long long n, typename F::T *aggs)
{
    __shared__ typename F::T buf[SCAN_BLOCK];
    long long i0 = (long long)blockIdx.x * SCAN_TILE + (long long)threadIdx.x * SCAN_IPT;
    int cnt = 0;
    if (i0 < n) cnt = (int)((n - i0) < SCAN_IPT ? (n - i0) : SCAN_IPT);
    typename F::T v = cnt > 0 ? f.reduce_run(i0, cnt) : f.identity();
    v = block_scan_inclusive(f, v, buf);
    if (threadIdx.x == SCAN_BLOCK - 1) aggs[blockIdx.x] = v;
}

// single block: aggs[b] <- exclusive prefix of aggs[0..nb); total -> aggs[nb]
template <typename F>
__global__ void __launch_bounds__(SCAN_BLOCK) scan_aggs_kernel(F f, typename F::T *aggs, int nb)
{
    __shared__ typename F::T buf[SCAN_BLOCK];
    const int t = threadIdx.x;
    const int run = (nb + SCAN_BLOCK - 1) / SCAN_BLOCK;   // dependent loads: keep nb small (scan_aggs_launch)
    const int b0 = t * run;
    const int b1 = min(nb, b0 + run);
    typename F::T v = f.identity();
    for (int b = b0; b < b1; ++b) v = f.combine(v, aggs[b]);
    typename F::T inc = block_scan_inclusive(f, v, buf);
    // exclusive prefix of this thread's run = inclusive of the previous thread
    typename F::T pre = f.identity();
    if (t > 0) pre = buf[t - 1];
    for (int b = b0; b < b1; ++b) {
        typename F::T e = aggs[b];
        aggs[b] = pre;
        pre = f.combine(pre, e);
    }
    if (t == SCAN_BLOCK - 1) aggs[nb] = inc;
}

template <typename F>
__global__ void __launch_bounds__(SCAN_BLOCK) scan_apply_kernel(F f, long long n, const typename F::T *aggs)
{
    __shared__ typename F::T buf[SCAN_BLOCK];
    long long i0 = (long long)blockIdx.x * SCAN_TILE + (long long)threadIdx.x * SCAN_IPT;
    int cnt = 0;
    if (i0 < n) cnt = (int)((n - i0) < SCAN_IPT ? (n - i0) : SCAN_IPT);
    typename F::T v = cnt > 0 ? f.reduce_run(i0, cnt) : f.identity();
    block_scan_inclusive(f, v, buf);
    typename F::T pre = aggs[blockIdx.x];
    if (threadIdx.x > 0) pre = f.combine(pre, buf[threadIdx.x - 1]);
    if (cnt > 0) f.apply_run(i0, cnt, pre);
}

// The same, taking the RAW block aggregates of scan_reduce_kernel: every block composes the aggregates in front of
// it by itself (a few hundred at most: one or two per thread, one block scan), which saves the single-block launch
// that would scan them -- these scans are a chain of launch latencies, not work.
template <typename F>
__global__ void __launch_bounds__(SCAN_BLOCK) scan_apply_lookback_kernel(F f, long long n, const typename F::T *aggs)
{
    __shared__ typename F::T buf[SCAN_BLOCK];
    const int t = threadIdx.x;
    const int nbefore = blockIdx.x;
    const int run = (nbefore + SCAN_BLOCK - 1) / SCAN_BLOCK;
    typename F::T v = f.identity();
    for (int b = t * run; b < min(nbefore, (t + 1) * run); ++b) v = f.combine(v, aggs[b]);
    block_scan_inclusive(f, v, buf);
    const typename F::T block_pre = buf[SCAN_BLOCK - 1];
    __syncthreads();
    long long i0 = (long long)blockIdx.x * SCAN_TILE + (long long)threadIdx.x * SCAN_IPT;
    int cnt = 0;
    if (i0 < n) cnt = (int)((n - i0) < SCAN_IPT ? (n - i0) : SCAN_IPT);
    v = cnt > 0 ? f.reduce_run(i0, cnt) : f.identity();
    block_scan_inclusive(f, v, buf);
    typename F::T pre = block_pre;
    if (threadIdx.x > 0) pre = f.combine(pre, buf[threadIdx.x - 1]);
    if (cnt > 0) f.apply_run(i0, cnt, pre);
}

static inline int scan_blocks(long long n) { return (int)((n + SCAN_TILE - 1) / SCAN_TILE); }

// the aggregates of a large scan are scanned with the same three kernels, one level up
template <typename F>
struct AggScanF {
    typedef typename F::T T;
    F f;
    T *a;
    __device__ T identity() const { return f.identity(); }
    __device__ T combine(const T &lo, const T &hi) const { return f.combine(lo, hi); }
    __device__ T reduce_run(long long i0, int cnt) const
    {
        T m = f.identity();
        for (int k = 0; k < cnt; ++k) m = f.combine(m, a[i0 + k]);
        return m;
    }
    __device__ void apply_run(long long i0, int cnt, const T &pre) const
    {
        T p = pre;
        for (int k = 0; k < cnt; ++k) {
            T e = a[i0 + k];
            a[i0 + k] = p;
            p = f.combine(p, e);
        }
    }
};

// aggs[0..nb) -> exclusive prefixes in place; aggs must have room for nb + scan_blocks(nb) + 2 elements
template <typename F>
static inline void scan_aggs_launch(const F &f, typename F::T *aggs, int nb, hipStream_t s)
{
    if (nb <= 4 * SCAN_BLOCK) {
        hipLaunchKernelGGL(scan_aggs_kernel<F>, dim3(1), dim3(SCAN_BLOCK), 0, s, f, aggs, nb);
        return;
    }
    const int nb2 = scan_blocks(nb);
    typename F::T *lvl2 = aggs + nb + 1;
    AggScanF<F> af{f, aggs};
    hipLaunchKernelGGL(scan_reduce_kernel<AggScanF<F>>, dim3(nb2), dim3(SCAN_BLOCK), 0, s, af, (long long)nb, lvl2);
    hipLaunchKernelGGL(scan_apply_lookback_kernel<AggScanF<F>>, dim3(nb2), dim3(SCAN_BLOCK), 0, s, af, (long long)nb, lvl2);
}

}  // namespace xrit
