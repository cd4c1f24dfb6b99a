// exact_walk.h -- cfg.front_exact = 2: a serial float32 recurrence put back together from exactly walked ranges.
// Shared by the exact Costas loop (costas_exact.hip) and the exact AGC (agc.hip).  Both loops are contractive recurrences
// whose float32 trajectory, started a few ulps beside the true one, COINCIDES with it after a while and stays on it; both
// are walked 64 samples per step on one wave (the policy's block(): a Picard iteration whose scans run the serial loop's own
// additions -- on the float lattice as integer prefix sums, certified by the literal step, or as a systolic DPP scan).
//
//  * A call is cut into W ranges of Lw samples; walker w (one wave) starts H samples in front of its range from an
//    approximate state (the policy's start(): the chains' hand-off states, the gain maps' prefixes), walks those quietly,
//    then writes its range and leaves, behind every block, the state it reached (bs).  Walker 0 -- and every walker that
//    would start in front of the call -- starts from the carried state, exactly.
//  * Joints (xw_fix_kernel, one round per launch).  Walker w's output belongs to the start state used[w]; it is right iff that
//    is walker w - 1's end state je[w - 1], bit for bit: by induction from walker 0 every output then is the serial loop's.
//    Where it is not, the wave of joint w walks range w again from je[w - 1], rewriting output and records, until the state
//    it reaches behind a block IS the record there (from that block on what stands is the continuation of an exact state);
//    if it reaches the end of the range without meeting, je[w] changes and joint w + 1 is looked at again in the next round.
//    Rounds repeat until none finds anything to do.
#pragma once

#include <hip/hip_runtime.h>

namespace xrit {
namespace xw {

__device__ __forceinline__ float shr1(float v, float first)
{
    // lane i <- lane i - 1, lane 0 <- first
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(first), __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_of(float v, int src)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}
__device__ __forceinline__ bool all(bool p) { return __builtin_amdgcn_ballot_w64(p) == ~0ull; }
__device__ __forceinline__ bool any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }

// floats as lattice points: monotone in x, consecutive floats are consecutive integers (both signs)
__device__ __forceinline__ int ord(float x)
{
    const int i = __float_as_int(x);
    return i >= 0 ? i : (int)(0x80000000u - (unsigned)i);
}
__device__ __forceinline__ float inv(int o)
{
    return __int_as_float(o >= 0 ? o : (int)(0x80000000u - (unsigned)o));
}
// sum over the 64 lanes, the same value in every lane (a guess's ingredient: the order of the additions does not matter)
__device__ __forceinline__ float wave_sum(float v)
{
#define XW_ADD(CTRL, RM) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, RM, 0xf, false))
    XW_ADD(0x111, 0xf); XW_ADD(0x112, 0xf); XW_ADD(0x114, 0xf); XW_ADD(0x118, 0xf); XW_ADD(0x142, 0xa); XW_ADD(0x143, 0xc);
#undef XW_ADD
    return lane_of(v, 63);
}
// inclusive prefix sum over the 64 lanes (rows of 16, then row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2..3)
__device__ __forceinline__ int prefix(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
    return v;
}

constexpr int MAX_ROUNDS = 72;       // lanes 0..j are exact after j Picard rounds whatever the guess: 64 always suffice
constexpr int NCNT = 8;              // counters: [0] joints that did not fit (this round), [1] blocks walked, [2] Picard rounds,
                                     // [3] blocks at the round limit, [4] lattice segments, [5] scans that fell back

struct Args {
    const float2 *x;        // the loop's input
    float2 *y;              // out: the exact output (policies with GUESS: in, the approximate one)
    float2 *js, *je, *used; // per walker: state at the start of its range / at its end / the start state its output belongs to
    float2 *bs;             // per block: the state behind the block
    unsigned *cnt;
    long long n;
    int Lw, H, W;
    int mode;               // the policy's scan switches
    int prio;               // the walkers' waves at a raised issue priority
};
template <typename P> struct KArgs { Args a; typename P::Par p; };

__device__ __forceinline__ bool same(float2 a, float2 b)
{
    return __float_as_uint(a.x) == __float_as_uint(b.x) && __float_as_uint(a.y) == __float_as_uint(b.y);
}

// blocks [from, to) (sample indices, multiples of 64 except the call's end) from state st; outputs and records are written
// from `write_from` on.  MEET: stop at the first block boundary where the state equals the record (true).
template <typename P, bool MEET>
__device__ __forceinline__ bool walk(const KArgs<P> &K, long long from, long long to, long long write_from, float2 &st,
                                     float2 *js_slot, long long js_at)
{
    const Args &A = K.a;
    const int lane = threadIdx.x & 63;
    unsigned blocks = 0, rounds = 0, bad = 0;
    unsigned lat[2] = {0, 0};          // lattice segments, scans that fell back
    bool met = false;
    float aux = 0.f;                   // the policy's scratch from block to block (never part of the state)
    const long long last = A.n - 1;
    auto idx_of = [&](long long blk) { const long long i = blk + lane; return i < last ? i : last; };
    float2 xn = A.x[idx_of(from)], yn = make_float2(0.f, 0.f);
    if (P::GUESS) yn = A.y[idx_of(from)];
    for (long long blk = from; blk < to; blk += 64) {
        const float2 x = xn, ya = yn;
        if (blk + 64 < to) {
            xn = A.x[idx_of(blk + 64)];
            if (P::GUESS) yn = A.y[idx_of(blk + 64)];
        }
        const int cnt = (int)min((long long)64, to - blk);
        float2 out;
        const int r = P::block(K.p, x, ya, cnt, st, out, A.mode, lat, aux);
        ++blocks;
        rounds += (unsigned)r;
        bad += r >= MAX_ROUNDS ? 1u : 0u;
        if (blk >= write_from && lane < cnt) A.y[blk + lane] = out;
        if (js_slot != nullptr && blk + 64 == js_at && lane == 0) *js_slot = st;
        if (blk >= write_from) {
            float2 *rec = A.bs + (blk >> 6);
            if (MEET) {
                if (same(*rec, st)) { met = true; break; }
            }
            if (lane == 0) *rec = st;
        }
    }
    if (lane == 0) {
        atomicAdd(A.cnt + 1, blocks);
        atomicAdd(A.cnt + 2, rounds);
        if (bad) atomicAdd(A.cnt + 3, bad);
        if (lat[0]) atomicAdd(A.cnt + 4, lat[0]);
        if (lat[1]) atomicAdd(A.cnt + 5, lat[1]);
    }
    return met;
}

template <typename P>
__global__ void __launch_bounds__(64) main_kernel(KArgs<P> K)
{
    // (a walker is a chain of dependent instructions on one wave: latency, not work -- it goes in front of whatever shares its SIMD)
    if (K.a.prio) __builtin_amdgcn_s_setprio(3);
    const Args &A = K.a;
    const int w = blockIdx.x;
    const int lane = threadIdx.x;
    const long long a = (long long)w * A.Lw;
    const long long b = min(A.n, a + A.Lw);
    const long long s = a > A.H ? a - A.H : 0;
    float2 st = P::start(K.p, s);
    if (s == a && lane == 0) A.js[w] = st;
    (void)walk<P, false>(K, s, b, a, st, s < a ? A.js + w : nullptr, a);
    if (lane == 0) {
        A.je[w] = st;
        if (b == A.n) P::carry_out(K.p, st);
    }
}

// the hand-over of the joints' start states: used[w] = js[w] (a kernel of its own: every walker has finished)
template <typename P>
__global__ void used_kernel(Args A)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w < A.W) A.used[w] = A.js[w];
    if (w == 0) A.cnt[0] = 0;
}

// one round over the joints.  what: 0 = walk the joints that do not fit, 1 = count them only
template <typename P>
__global__ void __launch_bounds__(64) fix_kernel(KArgs<P> K, int what)
{
    if (K.a.prio) __builtin_amdgcn_s_setprio(3);
    const Args &A = K.a;
    const int w = blockIdx.x + 1;
    const int lane = threadIdx.x;
    const float2 prev = A.je[w - 1];
    if (same(prev, A.used[w])) return;
    if (lane == 0) atomicAdd(A.cnt + 0, 1u);
    if (what == 1) return;
    const long long a = (long long)w * A.Lw;
    const long long b = min(A.n, a + A.Lw);
    float2 st = prev;
    const bool met = walk<P, true>(K, a, b, a, st, nullptr, 0);
    if (lane == 0) {
        A.used[w] = prev;
        if (!met) {
            A.je[w] = st;
            if (b == A.n) P::carry_out(K.p, st);
        }
    }
}

template <typename P>
__global__ void zero_kernel(unsigned *cnt, int n) { if ((int)threadIdx.x < n) cnt[threadIdx.x] = 0; }

}  // namespace xw
}  // namespace xrit
