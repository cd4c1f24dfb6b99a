// clock_overlap.h -- the clock recovery of a burst as OVERLAPPING exactly walked blocks (round 5).
//
// BASELINE.json's north star asks for the feedback loops "tiled into overlapping blocks so per-wavefront serial state is
// carried in registers while blocks run concurrently".  For the Mueller & Mueller loop (ClockRecovery::Work,
// /root/reference/demodulator/src/demodulator.cpp:156,449) a plain cold-started overlap does not reach the parity asked
// for -- the float32 recurrence lives on a lattice and does not forget a one-unit difference for ~1e5 symbols (clock_relay.h,
// DESIGN.md) -- but an EXACTLY walked overlap does: what the relay's passes buy is exactly walked history in front of
// every symbol (a symbol of the default three-pass relay has 50 k .. 74 k symbols of it), and a walker can just as well walk
// that history itself.  So:
//   * the burst's de-rotated samples are cut at fixed SAMPLE positions into G ranges; walker s (one wave, the 64-symbols-
//     per-step literal walk of clock_relay.h) starts H symbols in front of its range from the timing guess -- in the samples
//     of the range before, for the first ranges in the last samples of the burst before, which are kept in front of the new
//     ones --, walks them quietly, then stages the symbols it reads inside its range;
//   * no walker waits for another: ONE launch, no passes, no hand-over of end states, and -- what round 4's review asked
//     for -- nothing that ties the clock recovery of burst b + 1 to that of burst b: it is enqueued behind its Costas loop
//     while the walkers of burst b are still at work (demod.cpp), its latency is hidden, and its segments can be long
//     (work = 1 + H / L walks of the stream instead of three);
//   * the joints are settled afterwards (clock_overlap_scan_kernel): walker s stages a few symbols in front of its range,
//     the scan picks the one that sits where walker s - 1's end state says ITS next symbol would be (they agree to 1e-4
//     sample; half a symbol of margin), counts, lays out the output; a joint that does not fit within a quarter symbol --
//     a timing guess that counted a symbol more or less than the loop -- makes the call fall back to the relay of
//     clock_relay.h, walked to closure.  The first range of a burst is joined to the carried state of the burst before in
//     the same way, so a stream's symbol sequence has no seam.
// The symbols are those of float32 M&M trajectories that were started 40 k .. 80 k symbols earlier: as close to the serial
// trajectory as the three-pass relay's (measured: profiles/r5_overlap_parity.json).  cfg.clock_exact = 1 still IS the serial
// trajectory, word for word; this is the default configuration's plan for calls of a million symbols or more.
#pragma once

#include "clock_relay.h"

namespace xrit {

constexpr int OV_HEAD = 8;           // symbols at the head of a range whose positions are kept for the joint
constexpr int OV_EXHAUSTED = 2;      // the input ran out inside this range
constexpr int OV_EMPTY = 4;          // the input ran out before this walker's start
constexpr int OV_STUCK = 8;          // a watchdog ended the walk

struct OverlapSeg {
    ClockState end;                  // state in front of the symbol behind the last staged one (ii: index into the call's buffer)
    int count;                       // symbols staged
    int flags;
    int head_ii[OV_HEAD];            // read position (ii, mu) of the first staged symbols
    float head_mu[OV_HEAD];
    int pad_[2];
};
static_assert(sizeof(OverlapSeg) == 128, "OverlapSeg is a 128-byte record");

struct OverlapArgs {
    const float2 *x;                 // [history | new samples]
    const float *table;
    long long N, ni;
    const ClockState *start;         // [G] where every walker starts (clock_overlap_guess_kernel)
    int G;
    int store0;                      // walker 0 stages the symbols it reads at or beyond this sample
    int first_bound, Ls;             // range s >= 1 begins at sample first_bound + (s - 1) Ls
    int early;                       // ... and its walker stages from this many samples in front of it
    int stride;                      // staging slots per walker
    float *stage;                    // [G * stride]
    OverlapSeg *segs;                // [G]
    ClockPar par;
    int q_om, q_mu;                  // lattice steps of omega and of mu + omega, units of 2^-24 sample (clock_relay.h)
    unsigned *stat;                  // [0] steps, [1] guess rounds, [2] watchdog mark, [3] walkers that walked
    unsigned long long *moments;     // [2] sum |s|, sum s^2 over the staged symbols, units of 2^-20
    unsigned *simd_claim;            // [RELAY_CLAIM_WORDS] (clock_relay.h: which SIMDs of a CU hold a walker)
};

__device__ __forceinline__ int overlap_bound(const OverlapArgs &a, int s)
{
    return s <= 0 ? a.store0 : (s >= a.G ? 0x7fffffff : a.first_bound + (s - 1) * a.Ls);
}

// Two waves per workgroup like clock_relay_kernel<.., RING = true>: the prefetcher keeps the sample ring filled, the walker
// walks.  `span`: samples a block of 64 symbols can cover.
// RX: samples in the ring.  (Measured, round 5, C2 streamed: rings of 1024 samples -- XRIT_OV_SMALL_RING=1 -- cost the walk more
// than the LDS they free gives the front end beside it, 1.92 against 1.76 ms per burst; four walkers per workgroup, which puts a
// burst's walkers on a quarter of the CUs, 1.94 against 1.90.)
template <int RX>
__global__ void __launch_bounds__(128) clock_overlap_kernel(OverlapArgs a, int span)
{
    __shared__ float table[(XR_MM_NSTEPS + 1) * XR_MM_NTAPS];
    __shared__ cf32 xr[RX + RELAY_XMIR];
    __shared__ int sh_xhi, sh_pos_ii, sh_done, sh_simd[2], sh_swap, sh_claim;
    clock_table_to_lds(table, a.table);
    const int s = blockIdx.x, lane = threadIdx.x & 63;
    int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));

    ClockState T = a.start[s];
    T.ii = __builtin_amdgcn_readfirstlane((int)T.ii);
    T.mu = relay_lane(T.mu, 0); T.omega = relay_lane(T.omega, 0);
    T.p0 = cf32{relay_lane(T.p0.x, 0), relay_lane(T.p0.y, 0)}; T.p1 = cf32{relay_lane(T.p1.x, 0), relay_lane(T.p1.y, 0)};
    const int ni_w = (int)(a.ni < 0x7fffffffLL ? a.ni : 0x7fffffffLL);
    // staging begins with the symbols read at or beyond lo; those read at or beyond hi are the next walker's
    const int lo = s == 0 ? a.store0 : overlap_bound(a, s) - a.early;
    const int hi = overlap_bound(a, s + 1);
    const int x_lo = (int)(T.ii > 4 ? T.ii - 4 : 0);
    if (threadIdx.x == 0) { sh_xhi = x_lo; sh_pos_ii = (int)T.ii; sh_done = 0; }
    __syncthreads();
    if ((int)T.ii >= ni_w || (int)T.ii < 0) {
        // nothing to walk: the input ends in front of this walker (both waves leave)
        if (threadIdx.x == 0) {
            OverlapSeg e{};
            e.end = T;
            e.flags = OV_EMPTY;
            a.segs[s] = e;
        }
        return;
    }
    // which of the two waves walks (clock_relay.h: no SIMD holds two walkers)
    unsigned *claim = nullptr;
    if (a.simd_claim) {
        const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));       // HW_REG_HW_ID
        const unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));     // HW_REG_XCC_ID
        if (lane == 0) sh_simd[threadIdx.x >> 6] = (int)((hw >> 4) & 3u);
        __syncthreads();
        claim = a.simd_claim + ((((xcc & 7u) << 8) | ((hw >> 8) & 255u)) & (RELAY_CLAIM_WORDS - 1));
        if (threadIdx.x == 0) {
            const unsigned b0 = 1u << sh_simd[0], b1 = 1u << sh_simd[1];
            int swap = 0, mine = -1;
            if (!(atomicOr(claim, b0) & b0)) mine = sh_simd[0];
            else if (b1 != b0 && !(atomicOr(claim, b1) & b1)) { mine = sh_simd[1]; swap = 1; }
            sh_swap = swap;
            sh_claim = mine;
        }
        __syncthreads();
        role ^= __builtin_amdgcn_readfirstlane(sh_swap);
    }
    const cf32 *xs = reinterpret_cast<const cf32 *>(a.x);

    if (role == 1) {
        // ---- the prefetcher (clock_relay.h): [position, position + RX - XCH) of the samples in the ring
        const long long nlast = a.N > 0 ? a.N - 1 : 0;
        // (a walker whose range ends at hi never asks for samples beyond hi + a block's span)
        const long long xend = hi < 0x7fffffff ? (long long)hi + span + 16 : nlast + span + 16;
        int x_hi = x_lo;
        unsigned rounds = 0;
        while (!relay_ld(&sh_done)) {
            if (++rounds > (1u << 24)) { if (lane == 0) a.stat[2] = 0xc0000000u | (unsigned)s; break; }   // watchdog
            const int pii = relay_ld(&sh_pos_ii);
            const bool fx = x_hi + RELAY_XCH - RX <= pii && (long long)x_hi <= xend && (long long)x_hi <= nlast + span + 16;
            if (!fx) { __builtin_amdgcn_s_sleep(4); continue; }
            cf32 vx[RELAY_XCH / 64];
#pragma unroll
            for (int q = 0; q < RELAY_XCH / 64; ++q) {
                const long long i = (long long)x_hi + lane + 64 * q;
                vx[q] = xs[i < nlast ? i : nlast];
            }
#pragma unroll
            for (int q = 0; q < RELAY_XCH / 64; ++q) {
                const int slot = (x_hi + lane + 64 * q) & (RX - 1);
                xr[slot] = vx[q];
                if (slot < RELAY_XMIR) xr[RX + slot] = vx[q];
            }
            x_hi += RELAY_XCH;
            if (lane == 0) relay_st(&sh_xhi, x_hi);
        }
        return;
    }

    // ---- the walker
    const float gkw = a.par.gain_omega * (16777216.0f / (float)a.q_om), gkm = a.par.gain_mu * (16777216.0f / (float)a.q_mu);
    const int sh_om = 31 - __builtin_clz((unsigned)a.q_om), sh_mu = 31 - __builtin_clz((unsigned)a.q_mu);
    const int stride = __builtin_amdgcn_readfirstlane(a.stride);
    float *stage = a.stage + (size_t)s * (size_t)stride;
    OverlapSeg *seg = a.segs + s;
    int n_st = 0;                       // symbols staged
    unsigned steps = 0, rounds_total = 0;
    bool exhausted = false, stuck = false;
    int x_hi = x_lo;
    float m1 = 0.f, m2 = 0.f;
#ifdef XRIT_RELAY_TIMING
    unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
    unsigned n_refresh = 0, n_spin = 0;
#endif
    for (;;) {
        RELAY_TICK(5);
        const int ii0 = (int)T.ii;
        if ((unsigned)ii0 >= (unsigned)ni_w) { exhausted = true; break; }
        if (ii0 >= hi) break;                               // the next symbol is the next walker's
        ++steps;
        const int need_x = ii0 + span + 8;
        if (x_hi < need_x) {
            int spins = 0;
#ifdef XRIT_RELAY_TIMING
            ++n_refresh;
#endif
#pragma nounroll
            while (x_hi < need_x) {
                x_hi = relay_ld(&sh_xhi);
                if (x_hi >= need_x) break;
#ifdef XRIT_RELAY_TIMING
                ++n_spin;
#endif
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1 << 22)) { if (lane == 0) a.stat[2] = 0x80000000u | (unsigned)s; stuck = true; break; }
            }
            if (stuck) break;
        }
        RELAY_TICK(0);
        // the walker's state on the lattice; first guess: symbol n + lane at the walker's own rate
        const int mu0u = (int)(T.mu * 16777216.0f), W0 = (int)(T.omega * 16777216.0f);
        const int wint = W0 >> 24, wfrac = W0 & 0xffffff;
        const int fr0 = mu0u + lane * wfrac, bii = ii0 + lane * wint;
        int cii = bii + (fr0 >> 24), carm;
        float cmu = (float)(fr0 & 0xffffff) * (1.0f / 16777216.0f), com = T.omega;
        if (lane == 0) cmu = T.mu;
        carm = (int)rintf(cmu * (float)XR_MM_NSTEPS);
        cf32 p0{0.f, 0.f};
        float mm = 0.f;
        ClockState hs{};
        bool stale = false, inrange = true;
        RELAY_TICK(1);
        for (int round = 0; round < RELAY_ROUNDS; ++round) {
            ++rounds_total;
            inrange = cii >= ii0 && cii + XR_MM_NTAPS <= need_x;
            cf32 w[XR_MM_NTAPS];
            const cf32 *wp = xr + (cii & (RX - 1));
#pragma unroll
            for (int q = 0; q < XR_MM_NTAPS; ++q) w[q] = wp[q];
            p0 = clock_interp_arm(w, table, carm);
            hs.p0 = cf32{relay_shr1_from(p0.x, T.p0.x), relay_shr1_from(p0.y, T.p0.y)};
            hs.p1 = cf32{relay_shr1_from(hs.p0.x, T.p1.x), relay_shr1_from(hs.p0.y, T.p1.y)};
            // (the slicer decisions of the two symbols in front ARE the signs of those symbols -- clock_advance sets them so,
            // the start states of clock_overlap_guess_kernel and a flipped state too --: taken from the shifted symbols
            // instead of being shifted and carried themselves: four lane reads less per step)
            hs.c0 = cf32{hs.p0.x > 0.f ? 1.f : 0.f, hs.p0.y > 0.f ? 1.f : 0.f};
            hs.c1 = cf32{hs.p1.x > 0.f ? 1.f : 0.f, hs.p1.y > 0.f ? 1.f : 0.f};
            mm = clock_timing_error(p0, hs);
            const int dW = (int)rintf(mm * gkw) << sh_om;
            const int dM = (int)rintf(mm * gkm) << sh_mu;
            const int C = relay_scan(dW, lane);
            const int E = C + dM;
            const int D = relay_scan(E, lane) - E;
            const int fr = fr0 + D;
            int nii = bii + (fr >> 24);
            float nmu = (float)(fr & 0xffffff) * (1.0f / 16777216.0f);
            float nom = (float)(W0 + C - dW) * (1.0f / 16777216.0f);
            if (lane == 0) { nii = ii0; nmu = T.mu; nom = T.omega; }
            const int narm = (int)rintf(nmu * (float)XR_MM_NSTEPS);
            stale = nii != cii || narm != carm;
            cii = nii; carm = narm; cmu = nmu; com = nom;
            if (!__any(stale)) break;
        }
        RELAY_TICK(2);
        // the literal step from every lane's state, compared with the neighbour's state bit for bit
        ClockState st = hs;
        st.ii = cii; st.mu = cmu; st.omega = com;
        clock_advance(mm, p0, st, a.par);
        const int nxt_ii = relay_dpp<0x130>(cii);
        const float nxt_mu = relay_shl1(cmu), nxt_om = relay_shl1(com);
        const bool exists = (unsigned)cii < (unsigned)ni_w;
        const bool good = !stale && inrange;
        const bool ok = good && exists && lane < 63 && (int)st.ii == nxt_ii && st.mu == nxt_mu && st.omega == nxt_om;
        const unsigned long long okm = __ballot(ok), exm = __ballot(exists), gdm = __ballot(good), him = __ballot(cii >= hi),
                                 lom = __ballot(cii >= lo);
        const int m = ~okm ? __builtin_ctzll(~okm) : 64;
        const int e = ~exm ? __builtin_ctzll(~exm) : 64;
        const int g = ~gdm ? __builtin_ctzll(~gdm) : 64;
        const int h = him ? __builtin_ctzll(him) : 64;         // first lane whose symbol is the next walker's
        int nv = m + 1 < 64 ? m + 1 : 64;
        nv = nv < g ? nv : g;
        nv = nv < h ? nv : h;
        if (e < nv) { nv = e; exhausted = true; }
        RELAY_TICK(3);
        if (nv > 0) {
            // stage the verified symbols read inside the range (the lanes' positions rise with the lane: a suffix of [0, nv))
            const int l0 = lom ? __builtin_ctzll(lom) : 64;
            if (lane >= l0 && lane < nv) {
                const int o = n_st + lane - l0;
                if (o < stride) stage[o] = p0.x;
                if (o < OV_HEAD) { seg->head_ii[o] = cii; seg->head_mu[o] = cmu; }
                m1 += fabsf(p0.x);
                m2 += p0.x * p0.x;
            }
            if (l0 < nv) n_st += nv - l0;
            const int src = nv - 1;
            ClockState nt;
            nt.ii = __builtin_amdgcn_readlane((int)st.ii, src);
            nt.mu = relay_lane(st.mu, src);
            nt.omega = relay_lane(st.omega, src);
            nt.p0 = cf32{relay_lane(st.p0.x, src), relay_lane(st.p0.y, src)};
            nt.p1 = cf32{relay_lane(st.p1.x, src), relay_lane(st.p1.y, src)};
            T = nt;
            if (lane == 0) relay_st(&sh_pos_ii, (int)T.ii);
        }
        RELAY_TICK(4);
        if (exhausted || nv == 0) { exhausted = true; break; }
    }
#ifdef XRIT_RELAY_TIMING
    if (lane == 0) {
        for (int q = 0; q < 6; ++q) atomicAdd(&relay_dbg[q], tacc[q]);
        atomicAdd(&relay_dbg[6], (unsigned long long)steps);
        atomicAdd(&relay_dbg[7], (unsigned long long)n_refresh);
        atomicAdd(&relay_dbg[8], (unsigned long long)n_spin);
    }
#endif
    for (int off = 32; off > 0; off >>= 1) { m1 += __shfl_xor(m1, off, 64); m2 += __shfl_xor(m2, off, 64); }
    if (lane == 0) {
        relay_st(&sh_done, 1);
        if (claim && sh_claim >= 0) atomicAnd(claim, ~(1u << sh_claim));
        atomicAdd(&a.stat[0], steps);
        atomicAdd(&a.stat[1], rounds_total);
        atomicAdd(&a.stat[3], 1u);
        atomicAdd(&a.moments[0], (unsigned long long)((double)m1 * 1048576.0));
        atomicAdd(&a.moments[1], (unsigned long long)((double)m2 * 1048576.0));
        T.c0 = cf32{T.p0.x > 0.f ? 1.f : 0.f, T.p0.y > 0.f ? 1.f : 0.f};
        T.c1 = cf32{T.p1.x > 0.f ? 1.f : 0.f, T.p1.y > 0.f ? 1.f : 0.f};
        seg->end = T;
        seg->count = n_st < stride ? n_st : stride;
        seg->flags = (stuck ? OV_STUCK : 0) | (exhausted && !stuck ? OV_EXHAUSTED : 0) | (n_st > stride ? OV_STUCK : 0);
    }
}

// ---- where the walkers start ------------------------------------------------------------------------------------
// Walker s starts `hist` samples in front of its range, at a symbol instant of the timing curve (clock.hip: the unwrapped
// Oerder & Meyr line, symbol count against sample position, one point per block of BL samples).  `cnt` is this burst's curve,
// in the coordinates of its first NEW sample (buffer index padN); a start that falls into the history in front of it reads
// the curve of the burst before (`cnt_prev`, whose first new sample sits n_prev samples in front of this burst's).  Walker 0
// starts from the carried state instead (w0_carried: at buffer index w0_ii) when there is no such history.
__global__ void clock_overlap_guess_kernel(const double *__restrict__ cnt, int nb, const double *__restrict__ cnt_prev, int nb_prev,
                                           long long n_prev, int BL, double sps, float omega0, const float2 *__restrict__ x,
                                           const float *__restrict__ table, long long ni, int padN, int hist, int G,
                                           int store0, int first_bound, int Ls, const ClockState *__restrict__ carried,
                                           int w0_carried, int w0_ii, int valid_lo, ClockState *__restrict__ S,
                                           unsigned *__restrict__ stat, unsigned long long *__restrict__ moments)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x == 0) {
        if (threadIdx.x < 8) stat[threadIdx.x] = 0u;
        if (threadIdx.x < 2) moments[threadIdx.x] = 0ull;
    }
    if (s >= G) return;
    if (s == 0 && w0_carried) {
        ClockState c = carried[0];
        c.ii = w0_ii;
        S[0] = c;
        return;
    }
    const int bound = s == 0 ? store0 : first_bound + (s - 1) * Ls;
    double t = (double)bound - (double)hist;              // buffer coordinates
    if (t < (double)valid_lo + 3.0 + 2.0 * sps) t = (double)valid_lo + 3.0 + 2.0 * sps;
    // the curve that covers t, and t in its coordinates (u = samples behind that curve's first sample)
    const bool prev = t < (double)padN && cnt_prev != nullptr && nb_prev > 0;
    const double *c = prev ? cnt_prev : cnt;
    const int nbc = prev ? nb_prev : nb;
    const double org = prev ? (double)padN - (double)n_prev : (double)padN;
    double u = t - org;
    // the symbol instant nearest to u: invert the piecewise-linear count curve (clock_guess_kernel)
    const double target = rint(clk_count_at(c, nbc, sps, u, 0.0, BL));
    for (int it = 0; it < 4; ++it) u += (target - clk_count_at(c, nbc, sps, u, 0.0, BL)) * sps;
    t = u + org - 3.0;            // the M&M read position t = ii + mu sits 3 samples before the interpolation instant
    if (t < (double)valid_lo) t = (double)valid_lo;
    ClockState st;
    st.ii = (long long)floor(t);
    st.mu = (float)(t - floor(t));
    st.omega = omega0;
    st.p0 = cf32{0.f, 0.f}; st.p1 = cf32{0.f, 0.f};
    st.c0 = cf32{0.f, 0.f}; st.c1 = cf32{0.f, 0.f};
    if (st.ii >= ni) { st.ii = 0x7fffffff; S[s] = st; return; }      // the input ends in front of this walker
    for (int back = 2; back >= 1; --back) {
        const double tb = t - back * (double)omega0;
        if (tb < (double)valid_lo) continue;
        const long long ib = (long long)floor(tb);
        if (ib >= ni) continue;
        const float mub = (float)(tb - floor(tb));
        const int imu = (int)rintf(mub * (float)XR_MM_NSTEPS);
        const float *row = table + imu * XR_MM_NTAPS;
        float ar = 0.f, ai = 0.f;
        for (int q = 0; q < XR_MM_NTAPS; ++q) {
            const float2 v = x[ib + q];
            ar += row[XR_MM_NTAPS - 1 - q] * v.x;
            ai += row[XR_MM_NTAPS - 1 - q] * v.y;
        }
        st.p1 = st.p0; st.c1 = st.c0;
        st.p0 = cf32{ar, ai};
        st.c0 = cf32{ar > 0.f ? 1.f : 0.f, ai > 0.f ? 1.f : 0.f};
    }
    S[s] = st;
}

// ---- the joints ---------------------------------------------------------------------------------------------------
// One workgroup.  For every walker: which of its first staged symbols is the one behind the last symbol of the walker in
// front (the carried state, for walker 0) -- the staged symbol nearest to where that walker's end state reads next --, how
// many symbols it contributes and where they go; then the call's result as clock_relay_finalize_kernel leaves it (symbol
// count, carried state, unread tail, the look at the signal-to-noise ratio).
// ctl[10] = 1 (passes), [11] = 0, [14] = 2 Es/N0 (float bits), [15] = stuck, [17] = joints that do not fit, [18] = largest
// distance between a joint's two trajectories (float bits, samples), [19] = walkers
__global__ void __launch_bounds__(256) clock_overlap_scan_kernel(const OverlapSeg *__restrict__ segs, int G, int stride,
                                                                  const ClockState *__restrict__ carried_in, int w0_ii,
                                                                  int w0_carried, float omega_mid, int *__restrict__ j0,
                                                                  unsigned long long *__restrict__ offs,
                                                                  ClockState *__restrict__ carried_out, ClockResult *__restrict__ res,
                                                                  const float2 *__restrict__ x, float2 *__restrict__ tail_out,
                                                                  long long N, int *__restrict__ ctl,
                                                                  const unsigned long long *__restrict__ moments,
                                                                  unsigned long long cap)
{
    __shared__ unsigned long long part[256];
    __shared__ int s_term, s_bad, s_stuck;
    __shared__ unsigned s_far;
    __shared__ long long s_ii;
    if (threadIdx.x == 0) { s_term = 0x7fffffff; s_bad = 0; s_stuck = 0; s_far = 0u; }
    __syncthreads();
    // joints
    for (int s = threadIdx.x; s < G; s += blockDim.x) {
        const OverlapSeg e = segs[s];
        int pick = 0;
        if (e.flags & OV_STUCK) s_stuck = 1;
        if (e.flags & (OV_EXHAUSTED | OV_EMPTY)) atomicMin(&s_term, s);
        if (!(e.flags & OV_EMPTY) && !(s == 0 && w0_carried)) {
            // where the trajectory in front reads its next symbol
            double want;
            if (s == 0) want = (double)w0_ii + (double)carried_in[0].mu;
            else { const ClockState p = segs[s - 1].end; want = (double)p.ii + (double)p.mu; }
            double best = 1e30;
            const int nh = e.count < OV_HEAD ? e.count : OV_HEAD;
            for (int q = 0; q < nh; ++q) {
                const double d = fabs((double)e.head_ii[q] + (double)e.head_mu[q] - want);
                if (d < best) { best = d; pick = q; }
            }
            // (a walker in front that ended by exhaustion leaves nothing for this one: it is empty or beyond the end)
            const bool front_ended = s > 0 && (segs[s - 1].flags & (OV_EXHAUSTED | OV_EMPTY)) != 0;
            if (!front_ended) {
                if (nh == 0 || !(best < 0.25 * (double)omega_mid)) atomicAdd(&s_bad, 1);
                else atomicMax(&s_far, __float_as_uint((float)best));
            }
        }
        j0[s] = pick;
    }
    __syncthreads();
    const int term = s_term;            // the walker in whose range the input ends (none: the call fails)
    // output offsets: exclusive prefix sum of (count - j0) over the walkers up to the terminal one
    unsigned long long run = 0;
    {
        // (G <= a few thousand: every thread sums a contiguous piece, thread 0 scans the 256 partial sums)
        const int per = (G + (int)blockDim.x - 1) / (int)blockDim.x;
        const int b = threadIdx.x * per, e = b + per < G ? b + per : G;
        unsigned long long sum = 0;
        for (int s = b; s < e; ++s) sum += (s <= term && segs[s].count > j0[s]) ? (unsigned long long)(segs[s].count - j0[s]) : 0ull;
        part[threadIdx.x] = sum;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long acc = 0;
            for (int q = 0; q < (int)blockDim.x; ++q) { const unsigned long long v = part[q]; part[q] = acc; acc += v; }
            run = acc;
        }
        __syncthreads();
        unsigned long long acc = part[threadIdx.x];
        for (int s = b; s < e; ++s) {
            offs[s] = acc;
            acc += (s <= term && segs[s].count > j0[s]) ? (unsigned long long)(segs[s].count - j0[s]) : 0ull;
            if (s > term) offs[s] = ~0ull;          // nothing of this walker goes out
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        ClockState st;
        ctl[10] = 1;
        ctl[11] = 0;
        ctl[12] = 0;
        ctl[15] = s_stuck;
        ctl[16] = 0;
        ctl[17] = s_bad;
        ctl[18] = (int)s_far;
        ctl[19] = G;
        const bool ok = term < G && !s_stuck && s_bad == 0 && run <= cap;
        if (!ok) {
            res->ok = 0;
            res->n_symbols = term < G ? run : 0ull;
            res->terminal_chain = -1;
            st = carried_in[0];
            st.ii = w0_ii;
        } else {
            res->ok = 1;
            res->terminal_chain = term;
            res->n_symbols = run;
            st = segs[term].end;
        }
        {
            const double nsym = (double)run;
            const double a1 = nsym > 0 ? (double)moments[0] / 1048576.0 / nsym : 0.0, a2 = nsym > 0 ? (double)moments[1] / 1048576.0 / nsym : 0.0;
            const double var = a2 - a1 * a1;
            ctl[14] = __float_as_int(nsym > 0 && var > 0 ? (float)(a1 * a1 / var) : 1e30f);
        }
        long long ii = st.ii;
        if (ii > N) ii = N;
        if (ii < 0) ii = 0;
        res->ii_final = ii;
        s_ii = ii;
        st.ii = 0;
        carried_out[0] = st;
    }
    __syncthreads();
    const long long ii = s_ii;
    long long carry = N - ii;
    if (carry > 1024) {
        carry = 1024;
        if (threadIdx.x == 0) res->ok = 0;
    }
    for (long long i = threadIdx.x; i < carry; i += blockDim.x) tail_out[i] = x[ii + i];
}

// staged symbols to their place in the output: walker blockIdx.y, 1024 symbols per workgroup
__global__ void __launch_bounds__(256) clock_overlap_copy_kernel(const float *__restrict__ stage, const OverlapSeg *__restrict__ segs,
                                                                  const int *__restrict__ j0, const unsigned long long *__restrict__ offs,
                                                                  int stride, float *__restrict__ soft, unsigned long long cap,
                                                                  const ClockResult *__restrict__ res)
{
    if (!res->ok) return;
    const int s = blockIdx.y;
    const unsigned long long off = offs[s];
    if (off == ~0ull) return;
    const int first = j0[s], len = segs[s].count - first;
    const float *src = stage + (size_t)s * (size_t)stride + first;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = (int)blockIdx.x * 1024 + q * 256 + (int)threadIdx.x;
        if (i < len && off + (unsigned long long)i < cap) soft[off + (unsigned long long)i] = src[i];
    }
}

}  // namespace xrit
