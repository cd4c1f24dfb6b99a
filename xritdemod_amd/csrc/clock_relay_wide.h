// clock_relay_wide.h -- the relay's walker as a team of W waves: 62 W symbols per step.
//
// clock_relay.h walks a segment of the clock recovery (ClockRecovery::Work, /root/reference/demodulator/src/
// demodulator.cpp:156,449) exactly, 64 symbols per step, with ONE wave: guess where the next 64 symbols sit, form their
// timing errors side by side, two integer prefix sums give every lane its state on the float32 lattice, the literal
// float32 step verifies.  That wave issues one instruction every 5..7 cycles and a pass over all segments takes
// (segment length / 64) steps whatever the chip could do beside it; what the passes buy in parity is their horizon --
// passes x segment length -- so the lever is symbols per step.  Here W waves (one per SIMD; 8: two) take a step
// together:
//  * wave w owns symbols 62 w .. 62 w + 61 of the block (lanes 2..63).  Lanes 0 and 1 hold the two symbols in FRONT of
//    them -- the history the timing error needs -- which the wave interpolates itself from the same (read index, arm)
//    as their owner (wave w - 1's lanes 62, 63; for wave 0: the walker's carried p0, p1).  No interpolated value ever
//    crosses waves.
//  * the prefix sums cross waves as four integers per wave (totals of both sums and the two local prefixes its
//    successor needs for ITS history lanes) through an LDS mailbox; every wave then knows every position it needs.
//  * every lane runs the literal float32 step (clock_advance) from its state and compares the result with the integer
//    model's state after its own symbol; a wave publishes how many of its symbols stand verified, whether all of
//    them do, and the literal state after the last one.  All waves read all verdicts and come to the same conclusion:
//    another round (some (index, arm) moved), or commit the verified prefix of the block and go on from the literal
//    state behind it.
//  * a further wave streams samples and first guesses into LDS rings with loads that write LDS themselves
//    (global_load_lds_dwordx4), six refill units in flight, and tells the walkers how far the rings are filled.
// Waves wait for one another on sequence-numbered mailbox words (a wave's DS operations execute in order; the
// prefetching wave never takes part, so s_barrier is not used); every wait has a watchdog that marks the segment
// RELAY_STUCK instead of hanging the device.
#pragma once

#include "clock_relay.h"

namespace xrit {

constexpr int RW_OWN = 62;            // symbols a wave owns per step
constexpr int RW_XUNIT = 512;         // samples per refill unit: four loads of 1 KB (+ one of 64 bytes for the mirror)
constexpr int RW_GUNIT = 1280;        // first guesses per refill unit: five loads of 1 KB
constexpr int RW_UNIT_LOADS = 5;      // ... so that every unit in flight is five counts of vmcnt
constexpr int RW_DEPTH = 6;           // refill units in flight at most
constexpr int RW_SLOT = 8;            // mailbox words per wave
constexpr int RW_REC_PAD = 2048;      // words behind the last segment's records a refill unit may touch

template <int W> struct RelayWide {
    static constexpr int RX = W <= 2 ? 2048 : (W <= 4 ? 4096 : 8192);     // sample ring
    static constexpr int GR = W <= 4 ? 2048 : 4096;                       // ring of first guesses (symbols)
    // samples a block may span at most: the ring holds the block, a unit being refilled, the 128-sample rounding of
    // the ring's start and some slack
    static constexpr int MAX_SPAN = RX - RW_XUNIT - 128 - 80;
    static constexpr size_t lds_bytes()
    {
        return sizeof(float) * (XR_MM_NSTEPS + 1) * XR_MM_NTAPS + sizeof(cf32) * (RX + RELAY_XMIR) + sizeof(unsigned) * GR +
               sizeof(int) * (3 * RW_SLOT * W + 16);
    }
};

// wait for the oldest of `inflight` refill units (LDS-direct loads complete in the order issued; s_waitcnt takes an
// immediate)
__device__ __forceinline__ void rw_wait_oldest(int inflight)
{
    switch (inflight) {
    case 1: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(25)" ::: "memory"); break;
    }
}

__device__ __forceinline__ void rw_st4(int *p, int a, int b, int c, int d)
{
    typedef int rw_v4i __attribute__((ext_vector_type(4)));
    const rw_v4i v = {a, b, c, d};
    asm volatile("ds_write_b128 %0, %1" : : "v"(relay_lds_addr(p)), "v"(v) : "memory");
}
// every lane its own word of the mailbox
__device__ __forceinline__ int rw_ld_lane(const int *p)
{
    int v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(relay_lds_addr(p)) : "memory");
    return v;
}
__device__ __forceinline__ int rw_word(int v, int i) { return __builtin_amdgcn_readlane(v, i); }

#ifdef XRIT_RELAY_TIMING
// (instrumented build: cycles per phase of a step summed over wave 1's of all teams -- relay_dbg[0..6] --, steps in [7];
// the prefetchers: cycles waiting for their oldest unit [8], units landed [9], loop turns [10], cycles alive [11])
#define RW_TICK(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); wacc[i] += t_ - wlast; wlast = t_; } while (0)
#else
#define RW_TICK(i) do { } while (0)
#endif

constexpr int RW_SPIN_LIMIT = 1 << 20;      // polls of a mailbox (~100 cycles each) before a wave gives up

// Segment s of the call, walked by W waves (threads 0 .. 64 W - 1) fed by one more (the last 64 threads).
template <bool SYM, int W>
__global__ void __launch_bounds__(64 * (W + 1)) clock_relay_wide_kernel(RelayArgs a, int pass, int span)
{
    using CFG = RelayWide<W>;
    constexpr int RX = CFG::RX, GR = CFG::GR;
    if (a.ctl && !a.ctl[0]) return;                              // the tiled hand-off has not closed: nothing to refine yet
    if (pass > 0 && a.changed[RELAY_STAT * (pass - 1)] == 0) return;      // closed in an earlier pass
    __shared__ __attribute__((aligned(16))) float table[(XR_MM_NSTEPS + 1) * XR_MM_NTAPS];
    __shared__ __attribute__((aligned(16))) cf32 xr[RX + RELAY_XMIR];
    __shared__ __attribute__((aligned(16))) unsigned gr[GR];
    __shared__ __attribute__((aligned(16))) int mailA[RW_SLOT * W];          // scan totals, one generation
    __shared__ __attribute__((aligned(16))) int mailC[2 * RW_SLOT * W];      // verdicts, two generations
    __shared__ int sh_xhi, sh_pos_ii, sh_done, sh_ghi, sh_pos_n;
    clock_table_to_lds(table, a.table);
    const int s = blockIdx.x, lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // 0 .. W - 1: walkers; W: the prefetcher
    const RelaySeg *ein = a.ends[(pass + 1) & 1];
    RelaySeg *eout = a.ends[pass & 1];
    const int Lseg = a.cps * a.NS;
    const long long obase = (long long)s * Lseg;

    ClockState T{};
    bool dead = false;
    if (s == 0) T = a.first[0];
    else if (pass == 0) {
        const int k = s * a.cps;
        if (k < a.K) T = a.S[k];
        else dead = true;
    } else {
        const RelaySeg e = ein[s - 1];
        T = e.s;
        dead = (e.flags & (RELAY_EXHAUSTED | RELAY_DEAD | RELAY_STUCK)) != 0;
    }
    // (the walker's state is wave-uniform: kept in scalar registers, branches on it are scalar branches)
    T.ii = __builtin_amdgcn_readfirstlane((int)T.ii);
    T.mu = relay_lane(T.mu, 0); T.omega = relay_lane(T.omega, 0);
    T.p0 = cf32{relay_lane(T.p0.x, 0), relay_lane(T.p0.y, 0)}; T.p1 = cf32{relay_lane(T.p1.x, 0), relay_lane(T.p1.y, 0)};
    dead = __builtin_amdgcn_readfirstlane((int)dead) != 0;
    const RelaySeg prev = a.start[s];
    const cf32 *xs = reinterpret_cast<const cf32 *>(a.x);
    // ring positions are sample indices + shx, so that an even position is a 16-byte aligned address (the samples are
    // 8-byte aligned only when a call is run again from another carried state, ClockStage::redo_flipped: the sample in
    // front of the buffer's first then exists, kernels.h)
    const int shx = (int)((reinterpret_cast<unsigned long long>(xs) >> 3) & 1ull);
    const int x_lo = ((int)(T.ii > 4 ? T.ii - 4 : 0) + shx) & ~127;          // position the ring starts at
    for (int i = threadIdx.x; i < RW_SLOT * W; i += blockDim.x) { mailA[i] = -1; mailC[i] = -1; mailC[RW_SLOT * W + i] = -1; }
    if (threadIdx.x == 0) { sh_xhi = x_lo; sh_pos_ii = (int)T.ii + shx; sh_done = 0; sh_ghi = 0; sh_pos_n = 0; }
    __syncthreads();      // (the only barrier: all waves pass it before any can leave)
    if (dead) {
        if (pass > 0 && (prev.flags & RELAY_DEAD)) { if (threadIdx.x == 0) eout[s] = ein[s]; return; }
        if (threadIdx.x == 0) {
            RelaySeg e{};
            e.flags = RELAY_DEAD;
            eout[s] = e;
            a.start[s] = e;
            atomicAdd(&a.changed[RELAY_STAT * pass], 1u);
        }
        return;
    }
    if (pass > 0 && (prev.flags & RELAY_WALKED) && relay_same_state(prev.s, T)) { if (threadIdx.x == 0) eout[s] = ein[s]; return; }
    // the record of the walk before (this call's: the flags are cleared when a call's relay starts)
    const bool use_rec = a.rec != nullptr &&
                         __builtin_amdgcn_readfirstlane((int)((prev.flags & RELAY_WALKED) != 0 && prev.n_done > 0)) != 0;
    const int n_rec = use_rec ? __builtin_amdgcn_readfirstlane(prev.n_done) : 0;
    const int ref = __builtin_amdgcn_readfirstlane((int)(s == 0 ? a.first[0].ii : a.S[min(s * a.cps, a.K - 1)].ii)) -
                    RELAY_REF_MARGIN;
    unsigned *recs = a.rec ? a.rec + obase : nullptr;
    const int ni_w = (int)(a.ni < 0x7fffffffLL ? a.ni : 0x7fffffffLL);     // (read indices are 32-bit here)

    if (wv == W) {
        // ---- the prefetcher.  Positions [walker, x_landed) of the samples sit in the ring; a slot is overwritten only when
        // the walker's published position is past it.  Same for the first guesses, by symbol number.
        const int plast = (int)(a.N > 1 ? a.N - 1 : 1) + shx;
        const int pmax = (plast + 32) & ~1;           // (the input buffer has 64 samples of room behind its end)
        const float4 *x4 = reinterpret_cast<const float4 *>(xs - shx);       // position p (even) at x4[p / 2]
        int x_iss = x_lo, x_land = x_lo, g_iss = 0, g_land = 0, inflight = 0;
        unsigned types = 0, rounds = 0;
#ifdef XRIT_RELAY_TIMING
        unsigned long long pf_wait = 0, pf_units = 0;
        const unsigned long long pf_t0 = __builtin_amdgcn_s_memtime();
#endif
        while (!relay_ld(&sh_done)) {
            if (++rounds > (1u << 24)) { if (lane == 0) a.changed[RELAY_STAT * pass + 3] = 0xc0000000u | (unsigned)s; break; }   // watchdog
            const int ppos = relay_ld(&sh_pos_ii);
            const bool fx = x_iss + RW_XUNIT - RX <= ppos && x_iss <= plast + span + 16;
            bool fg = false;
            int pn = 0;
            if (use_rec) { pn = relay_ld(&sh_pos_n); fg = g_iss < n_rec && g_iss + RW_GUNIT - GR <= pn; }
            if (inflight < RW_DEPTH && (fx || fg)) {
                // (first guesses first when they are the scarcer of the two: less than three blocks ahead)
                const bool do_g = fg && (!fx || g_iss - pn < 3 * RW_OWN * W);
                if (do_g) {
#pragma unroll
                    for (int q = 0; q < RW_UNIT_LOADS; ++q)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(recs + g_iss + 256 * q + 4 * lane),
                                                         (__attribute__((address_space(3))) void *)(gr + ((g_iss + 256 * q) & (GR - 1))), 16, 0, 0);
                    g_iss += RW_GUNIT;
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        int p = x_iss + 128 * q + 2 * lane;
                        p = p < pmax ? p : pmax;
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(x4 + (p >> 1)),
                                                         (__attribute__((address_space(3))) void *)(xr + ((x_iss + 128 * q) & (RX - 1))), 16, 0, 0);
                    }
                    // the ring's first samples again behind its end (a window never wraps): those of the revolution this unit
                    // belongs to -- the real thing for the unit at the ring's start, the same bytes once more for the others
                    {
                        int p = (x_iss & ~(RX - 1)) + 2 * (lane & 3);
                        p = p < pmax ? p : pmax;
                        if (lane < 4)
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(x4 + (p >> 1)),
                                                             (__attribute__((address_space(3))) void *)(xr + RX), 16, 0, 0);
                    }
                    x_iss += RW_XUNIT;
                }
                types |= (do_g ? 1u : 0u) << inflight;
                ++inflight;
                continue;
            }
            if (inflight > 0) {
#ifdef XRIT_RELAY_TIMING
                const unsigned long long tw_ = __builtin_amdgcn_s_memtime();
                rw_wait_oldest(inflight);
                pf_wait += __builtin_amdgcn_s_memtime() - tw_;
                ++pf_units;
#else
                rw_wait_oldest(inflight);
#endif
                if (types & 1u) { g_land += RW_GUNIT; if (lane == 0) relay_st(&sh_ghi, g_land); }
                else { x_land += RW_XUNIT; if (lane == 0) relay_st(&sh_xhi, x_land); }
                types >>= 1;
                --inflight;
                continue;
            }
            __builtin_amdgcn_s_sleep(4);
        }
        // (nothing may still be on its way into this workgroup's LDS when the wave ends)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef XRIT_RELAY_TIMING
        if (lane == 0 && pass < 8) {
            atomicAdd(&relay_dbg[8], pf_wait);
            atomicAdd(&relay_dbg[9], pf_units);
            atomicAdd(&relay_dbg[10], (unsigned long long)rounds);
            atomicAdd(&relay_dbg[11], __builtin_amdgcn_s_memtime() - pf_t0);
        }
#endif
        return;
    }

    // ---- the walkers
    if (threadIdx.x == 0) {
        atomicAdd(&a.changed[RELAY_STAT * pass], 1u);
        // how far this start is from the one the segment was last walked from (samples): what the automatic closure
        // looks at (ClockStage::finish).  Non-negative floats order like their bits; watchdog marks stay on top.
        if (pass > 0 && (prev.flags & RELAY_WALKED)) {
            const float mv = fabsf(clock_tdiff(prev.s, T));
            atomicMax(&a.changed[RELAY_STAT * pass + 3], __float_as_uint(mv));
            const float m1 = fminf(mv, 1.0f);
            atomicAdd(reinterpret_cast<unsigned long long *>(&a.changed[RELAY_STAT * pass + 4]),
                      (unsigned long long)(m1 * m1 * 1099511627776.0f));
            atomicAdd(&a.changed[RELAY_STAT * pass + 6], 1u);
        }
    }
    const ClockState T0 = T;
    const float gkw = a.par.gain_omega * (16777216.0f / (float)a.q_om), gkm = a.par.gain_mu * (16777216.0f / (float)a.q_mu);
    const int sh_om = 31 - __builtin_clz((unsigned)a.q_om), sh_mu = 31 - __builtin_clz((unsigned)a.q_mu);
    const long long room = (long long)a.cap - obase;
    const int n_out = room <= 0 ? 0 : (room < (long long)Lseg ? (int)room : Lseg);
    float *softs = a.soft ? a.soft + obase : nullptr;
    float2 *syms = (SYM && a.sym) ? a.sym + obase : nullptr;
    const bool owned = lane >= 2;
    const int jw = RW_OWN * wv;                   // the block's symbol this wave's lane 2 holds
    const int jl = lane - 2;                      // this lane's symbol, relative to that
    int n = 0;
    unsigned steps = 0, rounds_total = 0;
    unsigned q = 0;                               // mailbox sequence number: one per round
    bool exhausted = false, stuck = false;
    int x_hi = x_lo, g_hi = 0;                    // what the rings are known to hold
    float m1 = 0.f, m2 = 0.f;                     // per lane: sum |s|, sum s^2 of the symbols it committed (first pass only)
#ifdef XRIT_RELAY_TIMING
    unsigned long long wacc[7] = {0, 0, 0, 0, 0, 0, 0}, wlast = __builtin_amdgcn_s_memtime();
#endif
    while (n < Lseg) {
        RW_TICK(6);
        ++steps;
        const int ii0 = (int)T.ii;
        if ((unsigned)ii0 >= (unsigned)ni_w) { exhausted = true; break; }
        const int need_x = ii0 + shx + span + 8;                // (a position)
        const int blk = n + RW_OWN * W < Lseg ? n + RW_OWN * W : Lseg;
        const int need_g = blk < n_rec ? blk : n_rec;
        if (x_hi < need_x || (use_rec && g_hi < need_g)) {
            int spins = 0;
#pragma nounroll
            while (x_hi < need_x) {
                x_hi = relay_ld(&sh_xhi);
                if (x_hi >= need_x) break;
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1 << 22)) { stuck = true; break; }
            }
#pragma nounroll
            while (use_rec && !stuck && g_hi < need_g) {
                g_hi = relay_ld(&sh_ghi);
                if (g_hi >= need_g) break;
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1 << 22)) { stuck = true; break; }
            }
            if (stuck) break;
        }
        RW_TICK(0);
        // the walker's state on the lattice; this wave's lane 2 at the walker's rate (64-bit on the scalar side: 500
        // symbols at the rate's fraction overflow 32 bits), the lanes relative to it
        const int mu0u = (int)(T.mu * 16777216.0f), W0 = (int)(T.omega * 16777216.0f);
        const int wint = W0 >> 24, wfrac = W0 & 0xffffff;
        const long long fb = (long long)mu0u + (long long)jw * (long long)wfrac;
        const int bii_w = ii0 + jw * wint + (int)(fb >> 24), fr_w = (int)(fb & 0xffffff);
        const int fr0 = fr_w + jl * wfrac, bii = bii_w + jl * wint;
        const bool hist_t = wv == 0 && lane < 2;           // wave 0's history lanes: the walker's own p1, p0
        const bool first = wv == 0 && lane == 2;           // the block's first symbol starts from the walker's state itself
        int cii = bii + (fr0 >> 24), carm;
        float cmu = (float)(fr0 & 0xffffff) * (1.0f / 16777216.0f), com = T.omega;
        if (first) { cii = ii0; cmu = T.mu; }
        carm = (int)rintf(cmu * (float)XR_MM_NSTEPS);
        if (use_rec && !first && !hist_t && n + jw + jl < n_rec) {
            const unsigned g = gr[(n + jw + jl) & (GR - 1)];
            if (g != RELAY_NOGUESS) {
                cii = ref + (int)(g >> 8);
                carm = min((int)(g & 0xffu), XR_MM_NSTEPS);
            }
        }
        cf32 p0{0.f, 0.f};
        float mm = 0.f;
        ClockState st{};
        int cnt = 0, src = -1, nv = 0;
        bool any_exh = false;
        RW_TICK(1);
        for (int round = 0; round < RELAY_ROUNDS; ++round) {
            ++rounds_total;
            ++q;
            const bool inrange = hist_t || (cii >= ii0 && cii + shx + XR_MM_NTAPS <= need_x);
            {
                cf32 w[XR_MM_NTAPS];
                const cf32 *wp = xr + ((cii + shx) & (RX - 1));
#pragma unroll
                for (int k = 0; k < XR_MM_NTAPS; ++k) w[k] = wp[k];
                p0 = clock_interp_arm(w, table, carm);
            }
            if (hist_t) p0 = lane == 1 ? T.p0 : T.p1;
            // the two symbols in front sit in the two lanes below
            ClockState hs{};
            hs.p0 = cf32{relay_shr1(p0.x), relay_shr1(p0.y)};
            hs.p1 = cf32{relay_shr1(hs.p0.x), relay_shr1(hs.p0.y)};
            hs.c0 = cf32{hs.p0.x > 0.f ? 1.f : 0.f, hs.p0.y > 0.f ? 1.f : 0.f};
            hs.c1 = cf32{hs.p1.x > 0.f ? 1.f : 0.f, hs.p1.y > 0.f ? 1.f : 0.f};
            mm = clock_timing_error(p0, hs);
            // omega and mu on the lattice: additions of rounded increments, i.e. two prefix sums -- inside the wave here
            const int dW = owned ? (int)rintf(mm * gkw) << sh_om : 0;
            const int dM = owned ? (int)rintf(mm * gkm) << sh_mu : 0;
            const int lc = relay_scan(dW, lane);
            const int le = lc + dM;
            const int ldx = relay_scan(le, lane) - le;
            // ... and across the waves: totals, and the two local prefixes the successor's history lanes stand behind
            if (wv < W - 1) {
                const int l62 = relay_dpp<0x138>(ldx);          // lane 63 <- lane 62
                if (lane == 63) {
                    rw_st4(&mailA[RW_SLOT * wv], lc, ldx + le, l62, ldx);
                    relay_st(&mailA[RW_SLOT * wv + 4], (int)q);
                }
            }
            int Cp = 0, Dp = 0, Dh0 = 0, Dh1 = 0;
            RW_TICK(2);
            if (wv > 0) {
                int v, spins = 0;
#pragma nounroll
                for (;;) {
                    v = rw_ld_lane(&mailA[lane < RW_SLOT * W ? lane : 0]);
                    const bool mine = lane < RW_SLOT * wv && (lane & (RW_SLOT - 1)) == 4;       // sequence words of the waves in front
                    if (!__any(mine && v != (int)q)) break;
                    if (++spins > RW_SPIN_LIMIT) { stuck = true; break; }
                }
                if (stuck) break;
#pragma unroll
                for (int u = 0; u < W - 1; ++u) {
                    if (u < wv) {
                        if (u == wv - 1) {
                            Dh0 = Dp + (RW_OWN - 2) * Cp + rw_word(v, RW_SLOT * u + 2);
                            Dh1 = Dp + (RW_OWN - 1) * Cp + rw_word(v, RW_SLOT * u + 3);
                        }
                        Dp += RW_OWN * Cp + rw_word(v, RW_SLOT * u + 1);
                        Cp += rw_word(v, RW_SLOT * u);
                    }
                }
            }
            RW_TICK(3);
            // every lane's state from the sums
            const int D = owned ? Dp + jl * Cp + ldx : (lane == 0 ? Dh0 : Dh1);
            const int fr = fr0 + D;
            int nii = bii + (fr >> 24);
            float nmu = (float)(fr & 0xffffff) * (1.0f / 16777216.0f);
            float nom = (float)(W0 + Cp + lc - dW) * (1.0f / 16777216.0f);
            if (first) { nii = ii0; nmu = T.mu; nom = T.omega; }
            int narm = (int)rintf(nmu * (float)XR_MM_NSTEPS);
            if (hist_t) { nii = cii; narm = carm; }
            const bool stale = nii != cii || narm != carm;
            cii = nii; carm = narm; cmu = nmu; com = nom;
            const bool wstale = __any(stale);
            // the literal step from every lane's state, compared bit for bit with the model's state behind its symbol
            st = hs;
            st.ii = cii; st.mu = cmu; st.omega = com;
            clock_advance(mm, p0, st, a.par);
            const int frn = fr + wfrac + Cp + le;
            const int nxt_ii = bii + wint + (frn >> 24);
            const float nxt_mu = (float)(frn & 0xffffff) * (1.0f / 16777216.0f);
            const float nxt_om = (float)(W0 + Cp + lc) * (1.0f / 16777216.0f);
            const bool exists = (unsigned)cii < (unsigned)ni_w;
            const bool good = !stale && inrange;                      // this lane's interpolation belongs to its state
            const bool ok = good && exists && (int)st.ii == nxt_ii && st.mu == nxt_mu && st.omega == nxt_om;
            // (history lanes count as fine when they are: a stale one is its owner's lane 62 / 63 too, which cuts the chain there)
            const unsigned long long okm = __ballot(ok || !owned), exm = __ballot(exists || !owned), gdm = __ballot(good);
            const int m = ~okm ? __builtin_ctzll(~okm) : 64;         // lanes 2 .. m start from verified states
            const int e = ~exm ? __builtin_ctzll(~exm) : 64;         // first lane whose symbol does not exist
            const int g = ~gdm ? __builtin_ctzll(~gdm) : 64;         // first lane whose own step is not to be trusted
            int nvw = m + 1 < 64 ? m + 1 : 64;
            nvw = nvw < g ? nvw : g;
            bool exh = false;
            if (e < nvw) { nvw = e; exh = true; }
            int c = nvw - 2;
            c = c > 0 ? c : 0;
            const int lim = Lseg - n - jw;
            if (c > lim) { c = lim > 0 ? lim : 0; exh = false; }
            const bool full = c == RW_OWN && m == 64;
            // the verdict: the lane behind whose symbol the wave's verified stretch ends leaves its literal state
            {
                const int wl = c > 0 ? c + 1 : 2;
                const int packed = (int)(((q & 0xffffu) << 16) | (wstale ? 0x200u : 0u) | (exh ? 0x100u : 0u) | (full ? 0x80u : 0u) | (unsigned)c);
                if (lane == wl) {
                    int *slot = &mailC[(q & 1u) * (RW_SLOT * W) + RW_SLOT * wv];
                    rw_st4(slot, (int)st.ii, __float_as_int(st.mu), __float_as_int(st.omega), __float_as_int(st.p0.x));
                    rw_st4(slot + 4, __float_as_int(st.p0.y), __float_as_int(st.p1.x), __float_as_int(st.p1.y), packed);
                }
            }
            RW_TICK(4);
            int v, spins = 0;
#pragma nounroll
            for (;;) {
                v = rw_ld_lane(&mailC[(q & 1u) * (RW_SLOT * W) + (lane < RW_SLOT * W ? lane : 0)]);
                const bool seqw = lane < RW_SLOT * W && (lane & (RW_SLOT - 1)) == 7;
                if (!__any(seqw && ((unsigned)v >> 16) != (q & 0xffffu))) break;
                if (++spins > RW_SPIN_LIMIT) { stuck = true; break; }
            }
            if (stuck) break;
            RW_TICK(5);
            bool any_stale = false;
            nv = 0; src = -1; any_exh = false;
            {
                bool chain = true;
#pragma unroll
                for (int u = 0; u < W; ++u) {
                    const int pk = rw_word(v, RW_SLOT * u + 7);
                    any_stale |= (pk & 0x200) != 0;
                    if (chain) {
                        const int cu = pk & 0x7f;
                        if (cu > 0) { nv += cu; src = u; }
                        any_exh |= (pk & 0x100) != 0;
                        chain = (pk & 0x80) != 0;
                    }
                }
            }
            cnt = c;
            if (any_stale && round + 1 < RELAY_ROUNDS) continue;
            // the literal state behind the last verified symbol: where the next block starts
            if (src >= 0) {
                T.ii = rw_word(v, RW_SLOT * src);
                T.mu = __int_as_float(rw_word(v, RW_SLOT * src + 1));
                T.omega = __int_as_float(rw_word(v, RW_SLOT * src + 2));
                T.p0 = cf32{__int_as_float(rw_word(v, RW_SLOT * src + 3)), __int_as_float(rw_word(v, RW_SLOT * src + 4))};
                T.p1 = cf32{__int_as_float(rw_word(v, RW_SLOT * src + 5)), __int_as_float(rw_word(v, RW_SLOT * src + 6))};
            }
            break;
        }
        if (stuck) break;
        // commit the verified prefix of the block
        {
            const int jb = jw + jl;            // this lane's symbol within the block
            if (owned && jb < nv) {
                if (pass == 0) { m1 += fabsf(p0.x); m2 += p0.x * p0.x; }
                const int o = n + jb;
                if (o < n_out) {
                    if (softs) softs[o] = p0.x;
                    if (SYM && syms) syms[o] = make_float2(p0.x, p0.y);
                }
                if (recs) {
                    const unsigned rel = (unsigned)(cii - ref);
                    recs[o] = rel < (1u << 24) ? (rel << 8) | (unsigned)carm : RELAY_NOGUESS;
                }
            }
        }
        (void)cnt;
        n += nv;
        if (threadIdx.x == 0) { relay_st(&sh_pos_ii, (int)T.ii + shx); relay_st(&sh_pos_n, n); }
        if (any_exh || nv == 0) { exhausted = true; break; }      // (nv == 0 without exhaustion cannot happen: the first lane is good)
    }
#ifdef XRIT_RELAY_TIMING
    if (lane == 0 && wv == (W > 1 ? 1 : 0) && pass < 8) {
        for (int i = 0; i < 7; ++i) atomicAdd(&relay_dbg[i], wacc[i]);
        atomicAdd(&relay_dbg[7], (unsigned long long)steps);
    }
#endif
    if (pass == 0 && a.moments) {
        // (lane sums in step order, lanes added in a fixed tree, waves and segments by integer atomics: the same value run after run)
        for (int off = 32; off > 0; off >>= 1) { m1 += __shfl_xor(m1, off, 64); m2 += __shfl_xor(m2, off, 64); }
        if (lane == 0) {
            atomicAdd(&a.moments[0], (unsigned long long)((double)m1 * 1048576.0));
            atomicAdd(&a.moments[1], (unsigned long long)((double)m2 * 1048576.0));
        }
    }
    if (threadIdx.x == 0) {
        relay_st(&sh_done, 1);
        atomicAdd(&a.changed[RELAY_STAT * pass + 1], steps);
        atomicAdd(&a.changed[RELAY_STAT * pass + 2], rounds_total);
        if (stuck) a.changed[RELAY_STAT * pass + 3] = 0x80000000u | (unsigned)s;
        RelaySeg st0{};
        st0.s = T0;
        st0.s.c0 = cf32{T0.p0.x > 0.f ? 1.f : 0.f, T0.p0.y > 0.f ? 1.f : 0.f};
        st0.s.c1 = cf32{T0.p1.x > 0.f ? 1.f : 0.f, T0.p1.y > 0.f ? 1.f : 0.f};
        st0.n_done = n;                 // symbols the record holds
        st0.flags = RELAY_WALKED;
        a.start[s] = st0;
        RelaySeg e{};
        e.s = T;
        e.s.c0 = cf32{T.p0.x > 0.f ? 1.f : 0.f, T.p0.y > 0.f ? 1.f : 0.f};
        e.s.c1 = cf32{T.p1.x > 0.f ? 1.f : 0.f, T.p1.y > 0.f ? 1.f : 0.f};
        e.n_done = n;
        e.flags = stuck ? RELAY_STUCK : (exhausted ? RELAY_EXHAUSTED : 0);
        eout[s] = e;
    }
}

}  // namespace xrit
