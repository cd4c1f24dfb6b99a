// newton.h -- the chain-boundary hand-off solve shared by the Costas and clock loops.
//
// Multiple shooting: chain k maps its start state S[k] to an end state E[k] with
// Jacobian J[k] (2x2).  With r_k = E[k] - S[k+1] (after removing the loop's
// exact symmetry: multiples of pi in phase / whole symbols in time, carried as an
// integer "aux") one Newton step solves
//        delta[k+1] = r_k + Jc_k delta[k],   delta[0] = 0
// which is a prefix scan of 2x2 affine maps.  Jc_k = J_k, or 0 where the
// un-gated solution delta_lin[k] is outside the trust region (the linearisation
// of a chain whose start is that far off is not believed; such a boundary gets
// the plain continuity hand-off).  Three launches per pass:
//   A  per-block reduce of the un-gated maps                         -> agg0
//   B  un-gated scan (delta_lin), gated maps built on the fly, reduce -> agg1
//   C  gated scan, per-boundary update through the loop's policy P
// Block prefixes come from an in-kernel look-back over the (<= NEWTON_MAX_BLOCKS)
// block aggregates, so there is no separate aggregate-scan launch.
//
// Policy P (device-callable members):
//   P::Elem P::fetch(k)                   everything boundary k needs from memory, loaded unconditionally: the
//                                         kernels fetch all their boundaries first, so the loads of a thread
//                                         are in flight together (these kernels are a chain of memory
//                                         latencies, not bandwidth)
//   bool   P::active(el)                  chain k produced something (else: pass state through)
//   void   P::residual(el, float& r1, float& r2, int& aux)
//   float4 P::jac(el)                     (a11, a12, a21, a22)
//   bool   P::outside_trust(d1, d2)
//   bool   P::distrust(r1, r2)            this boundary's own residual is outside the range in which the chain's
//                                         linearisation means anything (Costas: closer to the unstable
//                                         equilibrium a quarter turn away than to the lock point): plain hand-off
//   void   P::update(k, el, jd1, jd2, nd1, nd2, aux_prefix, aux_k, r1, NewtonStat&)   apply to S[k+1]
//   unsigned* P::cnt                      this pass's counter slot
//   void   P::decide(int* ctl)            stop test from the pass's counters; run by one thread of the block
//                                         of kernel C that finishes last
//   unsigned* P::cnt                      [0..5] statistics (see NewtonStat), [7] blocks of kernel C done
//   control block ctl: [0] non-zero = the hand-off has closed, the solve is a no-op; [5] non-zero = residuals
//   are still large enough for the trust gate to matter (else kernel B returns at once and C scans un-gated)
#pragma once

#include <hip/hip_runtime.h>

namespace xrit {

constexpr int NEWTON_BLOCK = 256;
constexpr int NEWTON_IPT = 4;
constexpr int NEWTON_TILE = NEWTON_BLOCK * NEWTON_IPT;
constexpr int NEWTON_MAX_BLOCKS = 4096;

struct AffMap { float a11, a12, a21, a22, b1, b2; int aux; };

// per-pass statistics, reduced per block before touching global memory
struct NewtonStat {
    unsigned changed, open_, large;
    float max_r;
    unsigned long long sum_sq;    // sum of min(r1^2, 1) in units of 2^-40: integer adds commute, so the
                                  // statistic (and the stop decision taken from it) is run-to-run deterministic
};
__device__ __forceinline__ unsigned long long newton_fix(float sq) { return (unsigned long long)(fminf(sq, 1.0f) * 1099511627776.0f); }
__device__ __forceinline__ float newton_unfix(unsigned long long v) { return (float)((double)v * (1.0 / 1099511627776.0)); }
// counter slot layout (8 words): [0] changed, [1] not frozen, [2] max |r1| bits, [3] large, [4..5] sum r1^2 (u64, 2^-40)

__device__ __forceinline__ AffMap aff_identity() { return AffMap{1.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0}; }

// lo first, then hi
__device__ __forceinline__ AffMap aff_combine(const AffMap &lo, const AffMap &hi)
{
    AffMap r;
    r.a11 = hi.a11 * lo.a11 + hi.a12 * lo.a21;
    r.a12 = hi.a11 * lo.a12 + hi.a12 * lo.a22;
    r.a21 = hi.a21 * lo.a11 + hi.a22 * lo.a21;
    r.a22 = hi.a21 * lo.a12 + hi.a22 * lo.a22;
    r.b1 = hi.a11 * lo.b1 + hi.a12 * lo.b2 + hi.b1;
    r.b2 = hi.a21 * lo.b1 + hi.a22 * lo.b2 + hi.b2;
    r.aux = lo.aux + hi.aux;
    return r;
}

// inclusive scan over the block of NB threads; returns this thread's inclusive value, buf[] holds all of them
template <int NB = NEWTON_BLOCK>
__device__ __forceinline__ AffMap aff_block_scan(AffMap v, AffMap *buf)
{
    const int t = threadIdx.x;
    buf[t] = v;
    __syncthreads();
    for (int off = 1; off < NB; off <<= 1) {
        AffMap lo = aff_identity();
        const bool has = t >= off;
        if (has) lo = buf[t - off];
        __syncthreads();
        if (has) {
            v = aff_combine(lo, v);
            buf[t] = v;
        }
        __syncthreads();
    }
    return v;
}

// prefix of all blocks before blockIdx.x, computed cooperatively from the aggregate array
__device__ __forceinline__ AffMap aff_lookback(const AffMap *aggs, AffMap *buf)
{
    const int nb = blockIdx.x;
    const int t = threadIdx.x;
    const int run = (nb + NEWTON_BLOCK - 1) / NEWTON_BLOCK;
    const int b0 = t * run, b1 = min(nb, b0 + run);
    AffMap v = aff_identity();
    for (int b = b0; b < b1; ++b) v = aff_combine(v, aggs[b]);
    aff_block_scan(v, buf);
    AffMap total = buf[NEWTON_BLOCK - 1];
    __syncthreads();
    return total;
}

template <typename P>
__device__ __forceinline__ AffMap newton_element(const P &p, const typename P::Elem &el, bool cut)
{
    AffMap m = aff_identity();
    if (!p.active(el)) {
        m.a11 = m.a22 = 0.f;    // nothing to hand over beyond the end of the data
        return m;
    }
    float r1, r2;
    int aux;
    p.residual(el, r1, r2, aux);
    if (cut || p.distrust(r1, r2)) { m.a11 = m.a12 = m.a21 = m.a22 = 0.f; }
    else {
        float4 j = p.jac(el);
        m.a11 = j.x; m.a12 = j.y; m.a21 = j.z; m.a22 = j.w;
    }
    m.b1 = r1; m.b2 = r2; m.aux = aux;
    return m;
}

// the thread's NEWTON_IPT boundaries, all loads issued before anything is used (indices clamped, not predicated)
template <typename P>
__device__ __forceinline__ void newton_fetch(const P &p, long long i0, long long n, typename P::Elem (&el)[NEWTON_IPT])
{
#pragma unroll
    for (int q = 0; q < NEWTON_IPT; ++q) el[q] = p.fetch(i0 + q < n ? i0 + q : n - 1);
}

template <typename P>
__global__ void __launch_bounds__(NEWTON_BLOCK) newton_reduce_kernel(P p, long long n, AffMap *agg0, const int *ctl)
{
    if (ctl[0]) return;
    __shared__ AffMap buf[NEWTON_BLOCK];
    const long long i0 = (long long)blockIdx.x * NEWTON_TILE + (long long)threadIdx.x * NEWTON_IPT;
    typename P::Elem el[NEWTON_IPT];
    newton_fetch(p, i0, n, el);
    AffMap v = aff_identity();
#pragma unroll
    for (int q = 0; q < NEWTON_IPT; ++q)
        if (i0 + q < n) v = aff_combine(v, newton_element(p, el[q], false));
    v = aff_block_scan(v, buf);
    if (threadIdx.x == NEWTON_BLOCK - 1) agg0[blockIdx.x] = v;
}

template <typename P>
__global__ void __launch_bounds__(NEWTON_BLOCK) newton_gate_kernel(P p, long long n, const AffMap *agg0, AffMap *agg1,
                                                                   float2 *dlin, const int *ctl)
{
    if (ctl[0] || !ctl[5]) return;
    __shared__ AffMap buf[NEWTON_BLOCK];
    const long long i0 = (long long)blockIdx.x * NEWTON_TILE + (long long)threadIdx.x * NEWTON_IPT;
    typename P::Elem el[NEWTON_IPT];
    newton_fetch(p, i0, n, el);
    AffMap pre = aff_lookback(agg0, buf);
    AffMap e[NEWTON_IPT];
    AffMap v = aff_identity();
#pragma unroll
    for (int q = 0; q < NEWTON_IPT; ++q) {
        e[q] = (i0 + q < n) ? newton_element(p, el[q], false) : aff_identity();
        v = aff_combine(v, e[q]);
    }
    aff_block_scan(v, buf);
    if (threadIdx.x > 0) pre = aff_combine(pre, buf[threadIdx.x - 1]);
    __syncthreads();
    // delta_lin at the start of this thread's run is the offset of the prefix map (delta[0] = 0)
    float d1 = pre.b1, d2 = pre.b2;
    AffMap g = aff_identity();
    for (int q = 0; q < NEWTON_IPT; ++q) {
        if (i0 + q >= n) break;
        dlin[i0 + q] = make_float2(d1, d2);
        AffMap m = e[q];
        if (p.outside_trust(d1, d2)) { m.a11 = m.a12 = m.a21 = m.a22 = 0.f; }
        g = aff_combine(g, m);
        float n1 = e[q].a11 * d1 + e[q].a12 * d2 + e[q].b1;
        float n2 = e[q].a21 * d1 + e[q].a22 * d2 + e[q].b2;
        d1 = n1; d2 = n2;
    }
    g = aff_block_scan(g, buf);
    if (threadIdx.x == NEWTON_BLOCK - 1) agg1[blockIdx.x] = g;
}

template <typename P>
__global__ void __launch_bounds__(NEWTON_BLOCK) newton_apply_kernel(P p, long long n, const AffMap *agg0,
                                                                    const AffMap *agg1, const float2 *dlin, int *ctl,
                                                                    NewtonStat *slots)
{
    if (ctl[0]) return;
    __shared__ AffMap buf[NEWTON_BLOCK];
    const long long i0 = (long long)blockIdx.x * NEWTON_TILE + (long long)threadIdx.x * NEWTON_IPT;
    const bool gated = ctl[5] != 0;
    typename P::Elem el[NEWTON_IPT];
    newton_fetch(p, i0, n, el);
    float2 dl[NEWTON_IPT];
#pragma unroll
    for (int q = 0; q < NEWTON_IPT; ++q) dl[q] = gated ? dlin[i0 + q < n ? i0 + q : n - 1] : make_float2(0.f, 0.f);
    AffMap pre = aff_lookback(gated ? agg1 : agg0, buf);
    AffMap e[NEWTON_IPT];
    AffMap v = aff_identity();
#pragma unroll
    for (int q = 0; q < NEWTON_IPT; ++q) {
        if (i0 + q < n) e[q] = newton_element(p, el[q], gated && p.outside_trust(dl[q].x, dl[q].y));
        else e[q] = aff_identity();
        v = aff_combine(v, e[q]);
    }
    aff_block_scan(v, buf);
    if (threadIdx.x > 0) pre = aff_combine(pre, buf[threadIdx.x - 1]);
    float d1 = pre.b1, d2 = pre.b2;
    int aux = pre.aux;
    NewtonStat st{0u, 0u, 0u, 0.f, 0ull};
#pragma unroll
    for (int q = 0; q < NEWTON_IPT; ++q) {
        const long long k = i0 + q;
        if (k >= n) break;
        float j1 = e[q].a11 * d1 + e[q].a12 * d2;
        float j2 = e[q].a21 * d1 + e[q].a22 * d2;
        float n1 = e[q].b1 + j1, n2 = e[q].b2 + j2;
        p.update(k, el[q], j1, j2, n1, n2, aux, e[q].aux, e[q].b1, st);
        d1 = n1; d2 = n2;
        aux += e[q].aux;
    }
    // one set of atomics per block
    for (int off = 32; off > 0; off >>= 1) {
        st.changed += __shfl_down(st.changed, off, 64);
        st.open_ += __shfl_down(st.open_, off, 64);
        st.large += __shfl_down(st.large, off, 64);
        st.max_r = fmaxf(st.max_r, __shfl_down(st.max_r, off, 64));
        st.sum_sq += (unsigned long long)__shfl_down((long long)st.sum_sq, off, 64);
    }
    __shared__ NewtonStat wst[NEWTON_BLOCK / 64];
    __shared__ int is_last;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) wst[threadIdx.x >> 6] = st;
    __syncthreads();
    if (threadIdx.x == 0) {
        NewtonStat t = wst[0];
        for (int w = 1; w < NEWTON_BLOCK / 64; ++w) {
            t.changed += wst[w].changed; t.open_ += wst[w].open_; t.large += wst[w].large;
            t.max_r = fmaxf(t.max_r, wst[w].max_r); t.sum_sq += wst[w].sum_sq;
        }
        // The block's statistics go to its own slot; the block that finishes last adds the slots up and takes the
        // stop decision (nobody reads ctl any more in this launch) -- no separate decision launch, and one atomic
        // per block instead of six on one cache line (release: this block's slot; acquire: everybody's).
        slots[blockIdx.x] = t;
        is_last = __hip_atomic_fetch_add(&p.cnt[7], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u;
    }
    __syncthreads();
    if (!is_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // every wave of this block reads the others' slots
    NewtonStat t{0u, 0u, 0u, 0.f, 0ull};
    for (int b = threadIdx.x; b < (int)gridDim.x; b += NEWTON_BLOCK) {
        const NewtonStat o = slots[b];
        t.changed += o.changed; t.open_ += o.open_; t.large += o.large;
        t.max_r = fmaxf(t.max_r, o.max_r); t.sum_sq += o.sum_sq;
    }
    for (int off = 32; off > 0; off >>= 1) {
        t.changed += __shfl_down(t.changed, off, 64);
        t.open_ += __shfl_down(t.open_, off, 64);
        t.large += __shfl_down(t.large, off, 64);
        t.max_r = fmaxf(t.max_r, __shfl_down(t.max_r, off, 64));
        t.sum_sq += (unsigned long long)__shfl_down((long long)t.sum_sq, off, 64);
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) wst[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        t = wst[0];
        for (int w = 1; w < NEWTON_BLOCK / 64; ++w) {
            t.changed += wst[w].changed; t.open_ += wst[w].open_; t.large += wst[w].large;
            t.max_r = fmaxf(t.max_r, wst[w].max_r); t.sum_sq += wst[w].sum_sq;
        }
        p.cnt[0] = t.changed;
        p.cnt[1] = t.open_;
        p.cnt[2] = t.open_ ? __float_as_uint(t.max_r) : 0u;
        p.cnt[3] = t.large;
        *reinterpret_cast<unsigned long long *>(&p.cnt[4]) = t.open_ ? t.sum_sq : 0ull;
        __threadfence();
        p.decide(ctl);
    }
}

// counters as the deciding thread must read them (other blocks' atomics)
__device__ __forceinline__ unsigned newton_cnt_load(const unsigned *c)
{
    return __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- small calls: the whole solve in ONE workgroup, one launch.  The reference hands the chain 32 Ki - 512 Ki
// samples per call (demodulator.cpp:113): a few thousand chain boundaries, for which the three launches above are
// three times a kernel's fixed cost with one or two blocks each.  Same phases, same arithmetic, same order of
// composition inside a thread's run; the block prefixes that the look-back provides there are simply absent.
constexpr int NEWTON_SMALL_BLOCK = 1024;
constexpr int NEWTON_SMALL_MAX = NEWTON_SMALL_BLOCK * NEWTON_IPT;

template <typename P>
__global__ void __launch_bounds__(NEWTON_SMALL_BLOCK) newton_small_kernel(P p, long long n, int *ctl)
{
    if (ctl[0]) return;
    __shared__ AffMap buf[NEWTON_SMALL_BLOCK];
    __shared__ NewtonStat wst[NEWTON_SMALL_BLOCK / 64];
    const long long i0 = (long long)threadIdx.x * NEWTON_IPT;
    const bool gated = ctl[5] != 0;
    typename P::Elem el[NEWTON_IPT];
    newton_fetch(p, i0, n, el);
    AffMap e[NEWTON_IPT];
    AffMap v = aff_identity();
#pragma unroll
    for (int q = 0; q < NEWTON_IPT; ++q) {
        e[q] = (i0 + q < n) ? newton_element(p, el[q], false) : aff_identity();
        v = aff_combine(v, e[q]);
    }
    aff_block_scan<NEWTON_SMALL_BLOCK>(v, buf);
    AffMap pre = threadIdx.x > 0 ? buf[threadIdx.x - 1] : aff_identity();
    __syncthreads();
    if (gated) {
        // delta_lin at the start of this thread's run is the offset of the un-gated prefix (delta[0] = 0)
        float d1 = pre.b1, d2 = pre.b2;
        AffMap g = aff_identity();
#pragma unroll
        for (int q = 0; q < NEWTON_IPT; ++q) {
            if (i0 + q < n) {
                const bool cut = p.outside_trust(d1, d2);
                const float n1 = e[q].a11 * d1 + e[q].a12 * d2 + e[q].b1;
                const float n2 = e[q].a21 * d1 + e[q].a22 * d2 + e[q].b2;
                d1 = n1; d2 = n2;
                if (cut) { e[q].a11 = e[q].a12 = e[q].a21 = e[q].a22 = 0.f; }
                g = aff_combine(g, e[q]);
            }
        }
        aff_block_scan<NEWTON_SMALL_BLOCK>(g, buf);
        pre = threadIdx.x > 0 ? buf[threadIdx.x - 1] : aff_identity();
        __syncthreads();
    }
    float d1 = pre.b1, d2 = pre.b2;
    int aux = pre.aux;
    NewtonStat st{0u, 0u, 0u, 0.f, 0ull};
#pragma unroll
    for (int q = 0; q < NEWTON_IPT; ++q) {
        const long long k = i0 + q;
        if (k >= n) break;
        float j1 = e[q].a11 * d1 + e[q].a12 * d2;
        float j2 = e[q].a21 * d1 + e[q].a22 * d2;
        float n1 = e[q].b1 + j1, n2 = e[q].b2 + j2;
        p.update(k, el[q], j1, j2, n1, n2, aux, e[q].aux, e[q].b1, st);
        d1 = n1; d2 = n2;
        aux += e[q].aux;
    }
    for (int off = 32; off > 0; off >>= 1) {
        st.changed += __shfl_down(st.changed, off, 64);
        st.open_ += __shfl_down(st.open_, off, 64);
        st.large += __shfl_down(st.large, off, 64);
        st.max_r = fmaxf(st.max_r, __shfl_down(st.max_r, off, 64));
        st.sum_sq += (unsigned long long)__shfl_down((long long)st.sum_sq, off, 64);
    }
    if ((threadIdx.x & 63) == 0) wst[threadIdx.x >> 6] = st;
    __syncthreads();
    if (threadIdx.x == 0) {
        NewtonStat t = wst[0];
        for (int w = 1; w < NEWTON_SMALL_BLOCK / 64; ++w) {
            t.changed += wst[w].changed; t.open_ += wst[w].open_; t.large += wst[w].large;
            t.max_r = fmaxf(t.max_r, wst[w].max_r); t.sum_sq += wst[w].sum_sq;
        }
        p.cnt[0] = t.changed;
        p.cnt[1] = t.open_;
        p.cnt[2] = t.open_ ? __float_as_uint(t.max_r) : 0u;
        p.cnt[3] = t.large;
        *reinterpret_cast<unsigned long long *>(&p.cnt[4]) = t.open_ ? t.sum_sq : 0ull;
        __threadfence();
        p.decide(ctl);
    }
}

// ---- wave-aligned solve: the steady-state path ---------------------------------------------------------------
// The three launches above cost a burst ~60 us per pass in launch latency alone (kernel time 36 us, of which 13 are
// spent re-reading what the pass kernel just had in registers).  Here the pass kernel itself leaves, per wave of 64
// chains, the composition of its 64 boundary maps (newton_wave_aggregate) -- a few thousand aggregates per burst --
// and ONE launch applies the step: every workgroup (16 waves = 1024 boundaries) scans the aggregates in front of
// its own waves in LDS (two per thread at C2, ten Hillis-Steele steps), each wave recomputes its 64 maps, scans
// them with shuffles and starts from its prefix.  (Measured and dropped: the scan done by the workgroup of the
// pass kernel that finishes last -- one wave walking 1770 aggregates is a chain of ~56 memory latencies, +70 us
// per pass.)  The trust gate needs the un-gated solution first and a second global scan; this path has none: a
// wave that finds a boundary outside the trust region raises ctl[NEWTON_CTL_TAKEOVER] and the host continues with
// the gated three-launch solve (acquisition, cold starts: the calls that need a host round trip per batch anyway).
// Boundary k sits between chains k and k + 1; wave w covers boundaries [64 w, 64 w + 64).
constexpr int NEWTON_CTL_TAKEOVER = 8;   // control word: the wave-aligned path met a boundary outside the trust region
constexpr int NEWTON_WAVES_BLOCK = 1024;
constexpr int NEWTON_WAVES_PER_BLOCK = NEWTON_WAVES_BLOCK / 64;

__device__ __forceinline__ AffMap aff_shfl_down(const AffMap &v, int off)
{
    AffMap r;
    r.a11 = __shfl_down(v.a11, off, 64); r.a12 = __shfl_down(v.a12, off, 64);
    r.a21 = __shfl_down(v.a21, off, 64); r.a22 = __shfl_down(v.a22, off, 64);
    r.b1 = __shfl_down(v.b1, off, 64); r.b2 = __shfl_down(v.b2, off, 64);
    r.aux = __shfl_down(v.aux, off, 64);
    return r;
}
__device__ __forceinline__ AffMap aff_shfl_up(const AffMap &v, int off)
{
    AffMap r;
    r.a11 = __shfl_up(v.a11, off, 64); r.a12 = __shfl_up(v.a12, off, 64);
    r.a21 = __shfl_up(v.a21, off, 64); r.a22 = __shfl_up(v.a22, off, 64);
    r.b1 = __shfl_up(v.b1, off, 64); r.b2 = __shfl_up(v.b2, off, 64);
    r.aux = __shfl_up(v.aux, off, 64);
    return r;
}
__device__ __forceinline__ AffMap aff_shfl(const AffMap &v, int src)
{
    AffMap r;
    r.a11 = __shfl(v.a11, src, 64); r.a12 = __shfl(v.a12, src, 64);
    r.a21 = __shfl(v.a21, src, 64); r.a22 = __shfl(v.a22, src, 64);
    r.b1 = __shfl(v.b1, src, 64); r.b2 = __shfl(v.b2, src, 64);
    r.aux = __shfl(v.aux, src, 64);
    return r;
}

// inclusive scan over the wave (lane order = boundary order)
__device__ __forceinline__ AffMap aff_wave_scan(AffMap v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const AffMap lo = aff_shfl_up(v, off);
        if (lane >= off) v = aff_combine(lo, v);
    }
    return v;
}

// Epilogue of a pass kernel, called by the wave that holds chains [64 w, 64 w + 64) with `e` = this lane's boundary
// map (identity beyond the last boundary): the ordered composition of the 64 maps goes to aggs[w].
__device__ __forceinline__ void newton_wave_aggregate(AffMap e, int w, AffMap *aggs)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const AffMap hi = aff_shfl_down(e, off);
        e = aff_combine(e, hi);          // lanes whose partner is out of range hold garbage nobody reads
    }
    if ((threadIdx.x & 63) == 0) aggs[w] = e;
}

// Calls of more than NEWTON_WAVES_TWO_LEVEL wave aggregates (bursts at the circuit rate: 16 k of them) take the composition
// in front of every workgroup from ONE scan (this kernel) instead of every workgroup scanning all aggregates in front of it
// -- that is quadratic in the call's length: 0.12 - 0.33 ms per solve at C1 / C3 (profiles/r4_c3_kernel_stats.csv).
constexpr int NEWTON_WAVES_TWO_LEVEL = 4096;
template <int UNUSED = 0>       // (a template: the header is included by two translation units)
__global__ void __launch_bounds__(NEWTON_WAVES_BLOCK) newton_wave_prefix_kernel(const AffMap *__restrict__ aggs, int nw,
                                                                               AffMap *__restrict__ wgpre, const int *ctl)
{
    __shared__ AffMap buf[NEWTON_WAVES_BLOCK];
    __shared__ int skip;
    if (threadIdx.x == 0) skip = ctl[0] || ctl[NEWTON_CTL_TAKEOVER];
    __syncthreads();
    if (skip) return;
    const int run = (nw + NEWTON_WAVES_BLOCK - 1) / NEWTON_WAVES_BLOCK;
    {
        const int b0 = (int)threadIdx.x * run, b1 = min(nw, b0 + run);
        AffMap v = aff_identity();
        for (int b = b0; b < b1; ++b) v = aff_combine(v, aggs[b]);
        aff_block_scan<NEWTON_WAVES_BLOCK>(v, buf);              // buf[t] = aggregates [0, (t + 1) run)
    }
    const int nwg = (nw + NEWTON_WAVES_PER_BLOCK - 1) / NEWTON_WAVES_PER_BLOCK;
    for (int g = (int)threadIdx.x; g < nwg; g += NEWTON_WAVES_BLOCK) {
        const int w0 = g * NEWTON_WAVES_PER_BLOCK, t0 = w0 / run;
        AffMap pre = t0 > 0 ? buf[t0 - 1] : aff_identity();
        for (int b = t0 * run; b < w0; ++b) pre = aff_combine(pre, aggs[b]);
        wgpre[g] = pre;                                         // aggregates [0, first wave of workgroup g)
    }
}

template <typename P>
__global__ void __launch_bounds__(NEWTON_WAVES_BLOCK) newton_apply_waves_kernel(P p, long long n, const AffMap *aggs,
                                                                               int *ctl, NewtonStat *slots,
                                                                               const AffMap *wgpre)
{
    __shared__ AffMap buf[NEWTON_WAVES_BLOCK];
    __shared__ NewtonStat wst[NEWTON_WAVES_PER_BLOCK];
    __shared__ int is_last, skip;
    // One decision per workgroup: another workgroup of this launch may raise the take-over flag at any time, and waves
    // of one workgroup that disagree about it would leave the block scan below with missing inputs.  (A workgroup that
    // saw the flag too late still takes part in the count of finished workgroups; the host goes over to the gated
    // solve either way.)
    if (threadIdx.x == 0) skip = ctl[0] || ctl[NEWTON_CTL_TAKEOVER];
    __syncthreads();
    if (skip) return;
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int w0 = blockIdx.x * NEWTON_WAVES_PER_BLOCK;         // first wave of this workgroup
    const int w = w0 + wib;
    const long long k = (long long)w * 64 + lane;
    // this lane's boundary first: its loads overlap the scan below
    const typename P::Elem el = p.fetch(k < n ? k : n - 1);
    // composition of the aggregates [0, w) for every wave of the workgroup: thread t composes a run of consecutive
    // aggregates, the workgroup scans the runs, a wave finishes from the run boundary in front of it
    const int nw = (int)((n + 63) / 64);
    const int need = min(nw, w0 + NEWTON_WAVES_PER_BLOCK);      // aggregates [0, need) are looked at
    const int run = (need + NEWTON_WAVES_BLOCK - 1) / NEWTON_WAVES_BLOCK;
    if (wgpre == nullptr) {
        const int b0 = (int)threadIdx.x * run, b1 = min(need, b0 + run);
        AffMap v = aff_identity();
        for (int b = b0; b < b1; ++b) v = aff_combine(v, aggs[b]);
        aff_block_scan<NEWTON_WAVES_BLOCK>(v, buf);              // buf[t] = aggregates [0, (t + 1) run)
    }
    NewtonStat st{0u, 0u, 0u, 0.f, 0ull};
    if ((long long)w * 64 < n) {
        AffMap pre;
        if (wgpre != nullptr) {
            // (long calls: what is in front of the workgroup comes from newton_wave_prefix_kernel)
            pre = wgpre[blockIdx.x];
            for (int b = w0; b < w; ++b) pre = aff_combine(pre, aggs[b]);
        } else {
            const int t0 = w / run;                             // run that holds aggregate w
            pre = t0 > 0 ? buf[t0 - 1] : aff_identity();
            for (int b = t0 * run; b < w; ++b) pre = aff_combine(pre, aggs[b]);
        }
        const AffMap e = k < n ? newton_element(p, el, false) : aff_identity();
        const AffMap inc = aff_wave_scan(e);
        AffMap ex = aff_shfl_up(inc, 1);
        if (lane == 0) ex = aff_identity();
        // delta[0] = 0: delta at this lane's boundary = (everything in front of it)(0)
        const float d1 = ex.a11 * pre.b1 + ex.a12 * pre.b2 + ex.b1, d2 = ex.a21 * pre.b1 + ex.a22 * pre.b2 + ex.b2;
        const int aux = pre.aux + ex.aux;
        const bool out = k < n && p.active(el) && p.outside_trust(d1, d2);
        if (__any(out)) {
            // the linearisation is not believed this far out: the gated solve takes over (host side)
            if (lane == 0) atomicExch(&ctl[NEWTON_CTL_TAKEOVER], 1);
        } else if (k < n) {
            const float j1 = e.a11 * d1 + e.a12 * d2, j2 = e.a21 * d1 + e.a22 * d2;
            p.update(k, el, j1, j2, e.b1 + j1, e.b2 + j2, aux, e.aux, e.b1, st);
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        st.changed += __shfl_down(st.changed, off, 64);
        st.open_ += __shfl_down(st.open_, off, 64);
        st.large += __shfl_down(st.large, off, 64);
        st.max_r = fmaxf(st.max_r, __shfl_down(st.max_r, off, 64));
        st.sum_sq += (unsigned long long)__shfl_down((long long)st.sum_sq, off, 64);
    }
    if (lane == 0) wst[wib] = st;
    __syncthreads();
    if (threadIdx.x == 0) {
        NewtonStat t = wst[0];
        for (int q = 1; q < NEWTON_WAVES_PER_BLOCK; ++q) {
            t.changed += wst[q].changed; t.open_ += wst[q].open_; t.large += wst[q].large;
            t.max_r = fmaxf(t.max_r, wst[q].max_r); t.sum_sq += wst[q].sum_sq;
        }
        slots[blockIdx.x] = t;
        is_last = __hip_atomic_fetch_add(&p.cnt[7], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u;
    }
    __syncthreads();
    if (!is_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    NewtonStat t{0u, 0u, 0u, 0.f, 0ull};
    for (int b = threadIdx.x; b < (int)gridDim.x; b += NEWTON_WAVES_BLOCK) {
        const NewtonStat o = slots[b];
        t.changed += o.changed; t.open_ += o.open_; t.large += o.large;
        t.max_r = fmaxf(t.max_r, o.max_r); t.sum_sq += o.sum_sq;
    }
    for (int off = 32; off > 0; off >>= 1) {
        t.changed += __shfl_down(t.changed, off, 64);
        t.open_ += __shfl_down(t.open_, off, 64);
        t.large += __shfl_down(t.large, off, 64);
        t.max_r = fmaxf(t.max_r, __shfl_down(t.max_r, off, 64));
        t.sum_sq += (unsigned long long)__shfl_down((long long)t.sum_sq, off, 64);
    }
    __syncthreads();
    if (lane == 0) wst[wib] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        t = wst[0];
        for (int q = 1; q < NEWTON_WAVES_PER_BLOCK; ++q) {
            t.changed += wst[q].changed; t.open_ += wst[q].open_; t.large += wst[q].large;
            t.max_r = fmaxf(t.max_r, wst[q].max_r); t.sum_sq += wst[q].sum_sq;
        }
        p.cnt[0] = t.changed;
        p.cnt[1] = t.open_;
        p.cnt[2] = t.open_ ? __float_as_uint(t.max_r) : 0u;
        p.cnt[3] = t.large;
        *reinterpret_cast<unsigned long long *>(&p.cnt[4]) = t.open_ ? t.sum_sq : 0ull;
        __threadfence();
        if (!__hip_atomic_load(&ctl[NEWTON_CTL_TAKEOVER], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) p.decide(ctl);
    }
}

static inline int newton_waves(long long n) { return (int)((n + 63) / 64); }
// storage of the wave-aligned solve for nw wave aggregates: the aggregates, the statistics slots, the workgroup prefixes
static inline size_t newton_waves_bytes(size_t nw)
{
    return (nw + 2) * sizeof(AffMap) + (nw / 16 + 4) * sizeof(NewtonStat) + (nw / 16 + 4) * sizeof(AffMap) + 64;
}
// storage: newton_waves(n) AffMaps (aggregates) and newton_waves(n) / 16 + 1 NewtonStat slots
template <typename P>
static inline void newton_apply_waves(const P &p, long long n, const AffMap *aggs, int *ctl, NewtonStat *slots,
                                      hipStream_t s)
{
    if (n <= 0) return;
    const int nw = newton_waves(n);
    const int nwg = (nw + NEWTON_WAVES_PER_BLOCK - 1) / NEWTON_WAVES_PER_BLOCK;
    const AffMap *wgpre = nullptr;
    if (nw > NEWTON_WAVES_TWO_LEVEL) {
        // (behind the statistics slots: newton_waves_bytes() holds room for it)
        AffMap *wp = reinterpret_cast<AffMap *>(slots + nwg + 2);
        hipLaunchKernelGGL(newton_wave_prefix_kernel<0>, dim3(1), dim3(NEWTON_WAVES_BLOCK), 0, s, aggs, nw, wp, (const int *)ctl);
        wgpre = wp;
    }
    hipLaunchKernelGGL(newton_apply_waves_kernel<P>, dim3(nwg), dim3(NEWTON_WAVES_BLOCK), 0, s, p, n, aggs, ctl, slots, wgpre);
}

static inline int newton_blocks(long long n) { return (int)((n + NEWTON_TILE - 1) / NEWTON_TILE); }

// agg storage: 3 * (blocks + 1) AffMaps (two sets of block aggregates, one statistics slot per block); dlin: n + 1 float2
template <typename P>
static inline int newton_solve(const P &p, long long n, AffMap *aggs, float2 *dlin, int *ctl, hipStream_t s)
{
    if (n <= 0) return 0;
    if (n <= NEWTON_SMALL_MAX) {
        hipLaunchKernelGGL(newton_small_kernel<P>, dim3(1), dim3(NEWTON_SMALL_BLOCK), 0, s, p, n, ctl);
        return 0;
    }
    const int nb = newton_blocks(n);
    if (nb > NEWTON_MAX_BLOCKS) return -1;
    AffMap *agg0 = aggs, *agg1 = aggs + nb + 1;
    hipLaunchKernelGGL(newton_reduce_kernel<P>, dim3(nb), dim3(NEWTON_BLOCK), 0, s, p, n, agg0, ctl);
    hipLaunchKernelGGL(newton_gate_kernel<P>, dim3(nb), dim3(NEWTON_BLOCK), 0, s, p, n, agg0, agg1, dlin, ctl);
    static_assert(sizeof(NewtonStat) <= sizeof(AffMap), "a statistics slot fits an AffMap");
    NewtonStat *slots = reinterpret_cast<NewtonStat *>(aggs + 2 * (nb + 1));
    hipLaunchKernelGGL(newton_apply_kernel<P>, dim3(nb), dim3(NEWTON_BLOCK), 0, s, p, n, agg0, agg1, dlin, ctl, slots);
    return 0;
}

}  // namespace xrit
