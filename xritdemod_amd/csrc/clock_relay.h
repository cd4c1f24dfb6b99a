// clock_relay.h -- exact closure of the clock recovery's time tiling (cfg.clock_exact).
//
// Why the hand-off passes of clock.hip stall at ~1e-4 sample: in float32 the M&M recurrence
// (ClockRecovery::Work, /root/reference/demodulator/src/demodulator.cpp:156,449) is an integer recurrence on a
// lattice -- omega and mu move by whole units of 2^-21 sample -- that sees its own state only through the
// interpolator arm rint(mu * 128) and the read index.  Two trajectories whose arms agree stay an exact translation
// of one another for ever (same symbols, same timing errors), and one arm that differs kicks them ~60 units apart;
// brought together they take ~8e4 symbols (1 % of the pairs: 4e5) to meet bit for bit.  A Newton step that moves
// 1e5 chain starts by 1e-4 sample flips arms in most chains, so the passes hover; nothing short of walking the
// recurrence closes it.
//
// So it is walked -- 64 symbols per step, and in many places at once:
//  * Relay.  The call is cut into G segments of `cps` chains.  Every segment is walked EXACTLY, one wave per
//    segment, from a start state: the carried state for segment 0, the tiled evaluation's hand-off state in the
//    first relay pass, and from then on the end state its predecessor reached in the pass before.  A pass in which
//    no segment's start changed has reproduced the serial trajectory: segment 0 starts from the exact state, so
//    by induction every segment does.  Trajectories forget their start (bit for bit, see above), so the passes
//    needed are ~5e5 symbols / segment length, not G; a segment whose start did not change is not walked again.
//  * The walk, 64 symbols per step.  Given the interpolator arm and read index of every symbol of a block, the
//    timing errors mm_i are independent of one another (symbol i needs the samples and the interpolated values of
//    symbols i - 1, i - 2), and (omega, mu) follow from them by ADDITIONS on the lattice: two integer prefix sums
//    over the wave.  So lane i guesses where symbol n + i sits (first from the nominal rate), interpolates,
//    forms mm_i; the sums give every lane its state; a lane whose arm or index came out different interpolates
//    again (two or three rounds settle all 64 -- a wrong arm moves mm by 1e-3, mu by 4e-6).  Then every lane runs
//    the LITERAL float32 step from its state and compares the result with its neighbour's state, bit for bit:
//    the verified prefix is committed.  The integer model is only a guess generator -- ties, the omega clip, a
//    binade boundary make it differ from the float arithmetic, the comparison then ends the block early and the
//    next block starts from the literal result.  Lane 0 always starts from the walker's own state.
//  * The samples are streamed into an LDS ring by a second wave of the workgroup that runs ahead of the walker
//    (the walker itself reads LDS only and issues stores nobody waits for).
//  * A segment that is walked again meets nearly the same timing errors as the walk before: every walk leaves, per symbol,
//    how far it ADVANCED behind it (difference from the nominal period, 2^-24 sample), the prefetching wave streams the
//    record of the walk before into a second LDS ring, and the lanes take their first guess from it -- the advances summed
//    over the lanes in front, anchored at the walker's own state at lane 0, plus the rate at which this walk closed in on
//    the recorded one over the block before.  One round settles all but a few per cent of the blocks, even behind a walk
//    that started 4e-2 sample away (round 4; rounds 3-4 kept (index, arm), which is only right within a fraction of an arm).
#pragma once

#include "kernels.h"

namespace xrit {

constexpr int RELAY_STAT = 8;        // counters per relay pass (RelayArgs::changed)
constexpr int RELAY_WALKED = 1;      // start[]: the segment has been walked exactly from start[].s
constexpr int RELAY_EXHAUSTED = 2;   // ends[]: the input ran out inside this segment (n_done symbols exist)
constexpr int RELAY_DEAD = 4;        // the input ran out before this segment
constexpr int RELAY_STUCK = 8;       // ends[]: a watchdog ended the walk (a ring or a mailbox never filled): the call fails
constexpr int RELAY_APPROX = 16;     // start[]: the segment has been walked from start[].s by an approximate pass (apx > 0): not exactly
constexpr int RELAY_REC = 32;        // start[]: ... which left the record of its (index, arm) guesses
constexpr int RELAY_CLAIM_WORDS = 2048;   // one word per CU, indexed by (XCC_ID, SE_ID, SH_ID, CU_ID)

struct RelaySeg {
    ClockState s;
    int n_done;
    int flags;
    int pad[2];
};
static_assert(sizeof(RelaySeg) == 64, "RelaySeg is one 64-byte record");

struct RelayArgs {
    const float2 *x;
    const float *table;
    long long N, ni;
    const ClockState *first;      // the call's carried state: start of segment 0
    const ClockState *S;          // chain starts of the tiled evaluation (relay pass 0)
    int K, cps, NS, G;
    RelaySeg *start;              // [G] what each segment was last walked from
    RelaySeg *ends[2];            // [G] end states, ping-pong between passes
    float *soft;
    float2 *sym;
    unsigned long long cap;
    ClockPar par;
    int q_om, q_mu;               // lattice steps of omega and of mu + omega, in units of 2^-24 sample
    unsigned *changed;            // [RELAY_STAT * pass] segments whose start changed in that pass, [+1] steps, [+2] guess rounds,
                                  // [+3] largest move of a start against the walk before (float bits) or a watchdog mark,
                                  // [+4, +5] sum of the squared moves (64 bits, 2^-40 sample^2), [+6] moves summed
    const int *ctl;               // clock control block: ctl[0] != 0 once the tiled hand-off has closed (null: do not ask)
    unsigned long long *moments;  // [2] sum |s| and sum s^2 over the soft symbols of the first relay pass, units of 2^-20 (the
                                  // default configuration's look at the signal-to-noise ratio: ClockStage::finish)
    unsigned *rec;                // [G * cps * NS] the advance behind every symbol as last walked (one-wave walker: difference from the
                                  // nominal period, 2^-24 sample; walker teams: (read index - segment reference) << 8 | arm)
                                  // (null: no records -- every block starts from the nominal rate)
    unsigned *simd_claim;         // [RELAY_CLAIM_WORDS] per CU: the SIMDs that hold a walker (null: roles by wave number)
    int sym_skip;                 // this pass's walks from a GUESS (every segment but the first of a plan without hand-off passes, in its
                                  // first pass) store no symbols and do not count as walked: the next pass walks them again whatever its starts
    int rec_use, rec_write;       // this pass takes its first guesses from the record of the walk before / leaves its own
};

__device__ __forceinline__ bool relay_same_state(const ClockState &a, const ClockState &b)
{
    return a.ii == b.ii && a.mu == b.mu && a.omega == b.omega && a.p0.x == b.p0.x && a.p0.y == b.p0.y &&
           a.p1.x == b.p1.x && a.p1.y == b.p1.y && a.c0.x == b.c0.x && a.c0.y == b.c0.y && a.c1.x == b.c1.x &&
           a.c1.y == b.c1.y;
}

// ---- wave helpers (DPP: no LDS round trip) ---------------------------------------------------------------------
template <int CTRL> __device__ __forceinline__ int relay_dpp(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);      // lanes without a source lane get 0
}
__device__ __forceinline__ float relay_shr1(float v) { return __int_as_float(relay_dpp<0x138>(__float_as_int(v))); }   // lane i <- i - 1
__device__ __forceinline__ float relay_shl1(float v) { return __int_as_float(relay_dpp<0x130>(__float_as_int(v))); }   // lane i <- i + 1
__device__ __forceinline__ float relay_lane(float v, int src)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}
// lane i <- i - 1, lane 0 <- first (the DPP move leaves lanes without a source lane what they held)
__device__ __forceinline__ float relay_shr1_from(float v, float first)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(first), __float_as_int(v), 0x138, 0xf, 0xf, false));
}
// inclusive prefix sum over the 64 lanes: inside every row of 16, then row 0's total into row 1 and row 2's into
// row 3 (row_bcast:15, rows 1 and 3), then the total of rows 0..1 into rows 2..3 (row_bcast:31)
__device__ __forceinline__ int relay_scan(int v, int)
{
    v += relay_dpp<0x111>(v);       // row_shr:1 .. 8
    v += relay_dpp<0x112>(v);
    v += relay_dpp<0x114>(v);
    v += relay_dpp<0x118>(v);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
    return v;
}

// The hand-shake words between the two waves live in LDS and order LDS traffic only.  A wave's DS operations are
// executed in the order it issued them and the LDS serves one CU, so all that is needed is that the COMPILER keeps
// the order.  Release / acquire atomics (also with the "local" fence scope) and volatile accesses both make this
// compiler wait for the wave's outstanding global stores (s_waitcnt vmcnt(0)) at every publish -- measured: 2.5 us
// per step -- hence the two instructions by hand, each a compiler barrier.
__device__ __forceinline__ unsigned relay_lds_addr(const void *p)
{
    return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) char *)p;
}
__device__ __forceinline__ int relay_ld(const int *p)
{
    int v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(relay_lds_addr(p)) : "memory");
    // every lane read the same word: said so, else every branch on it is an exec-mask region and the walker's state
    // lives in vector registers
    return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ void relay_st(int *p, int v)
{
    asm volatile("ds_write_b32 %0, %1" : : "v"(relay_lds_addr(p)), "v"(v) : "memory");
}

#ifdef XRIT_RELAY_TIMING
// (instrumented build, make EXTRA=-DXRIT_RELAY_TIMING: shader-clock cycles the walkers spend per phase of a step, summed
// over all walkers and passes of a call; printed with XRIT_TRACE)
__device__ unsigned long long relay_dbg[16];
// per walker of the first four passes of a call: cycles from its first step to its last, cycles waiting for the rings, steps,
// guess rounds, HW_ID, XCC_ID (where it ran: which SIMD of which CU)
constexpr int RELAY_WDBG_SEGS = 1024, RELAY_WDBG_PASSES = 4, RELAY_WDBG_WORDS = 6;
__device__ unsigned relay_wdbg[RELAY_WDBG_PASSES * RELAY_WDBG_SEGS * RELAY_WDBG_WORDS];
#define RELAY_TICK(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tacc[i] += t_ - tlast; tlast = t_; } while (0)
#else
#define RELAY_TICK(i) do { } while (0)
#endif

constexpr int RELAY_RX = 2048;       // sample ring (RING): about five steps of lead for the prefetching wave.  (Twice the size is as fast for the
                                     // walk, but three workgroups then hold 135 KB of a CU's LDS and the front end of the next burst, started in
                                     // front of the relay -- xrit_demod_prefetch_device --, cannot move in next to them: 2.65 against 2.33 ms per
                                     // burst in the three-pass configuration; half the size again costs the walk 6 %.)
constexpr int RELAY_XCH = 512;       // samples per refill (8 per lane)
constexpr int RELAY_ROUNDS = 4;      // guess rounds per step at most
constexpr int RELAY_XMIR = 8;        // the ring's first samples again behind its end: a window never wraps
constexpr int RELAY_GR = 1024;       // ring of first guesses (symbols), refilled RELAY_GCH at a time
constexpr int RELAY_GCH = 256;
constexpr unsigned RELAY_NOGUESS = 0xffffffffu;
constexpr unsigned RELAY_NOPOS = 0x80000000u;     // record of the one-wave walker: no position
constexpr int RELAY_REF_MARGIN = 1 << 20;     // a segment's reference index sits this far in front of its nominal start

// RING: two waves per workgroup, samples through the LDS ring (span = samples a block of 64 symbols can cover
// <= RELAY_RX - RELAY_XCH - 72); else one wave that reads its windows from global memory (any symbol rate).
// apx (round 4, cfg.clock_exact = -3: the passes in front of the last): an APPROXIMATE walk.  What those passes are for is
// the end state of every segment -- the start of the segment behind it in the next pass -- and the loop forgets: an end state
// is as good as the loop's memory of the errors made on the way.  So the walk takes its guess rounds and stops there:
//   apx = 1   ONE round: every symbol interpolated where the walker's own rate puts it (off by up to an arm, 8e-3 sample, at
//             the end of a block), the timing errors from that, the states from their sums; nothing is stored;
//   apx = 2   two rounds at most (the second with the arms the first one's sums gave: all but a few per cent of the symbols
//             are then where the exact walk has them), and the record of (index, arm) is left for the next pass' first guesses.
// No literal verification (a block is 64 symbols unless the input ends), no symbols, the segment is not marked as walked:
// an exact pass (apx = 0) walks it whatever its start was.
template <bool SYM, bool RING>
__global__ void __launch_bounds__(RING ? 128 : 64) clock_relay_kernel(RelayArgs a, int pass, int span, int apx)
{
    if (a.ctl && !a.ctl[0]) return;                              // the tiled hand-off has not closed: nothing to refine yet
                                                                 // (null: it never will -- pass budget used up -- go anyway)
    if (pass > 0 && a.changed[RELAY_STAT * (pass - 1)] == 0) return;      // closed in an earlier pass
    __shared__ float table[(XR_MM_NSTEPS + 1) * XR_MM_NTAPS];
    __shared__ cf32 xr[RING ? RELAY_RX + RELAY_XMIR : 1];
    __shared__ unsigned gr[RING ? RELAY_GR : 1];
    __shared__ int sh_xhi, sh_pos_ii, sh_done, sh_ghi, sh_pos_n, sh_simd[2], sh_swap, sh_claim;
    clock_table_to_lds(table, a.table);
    const int s = blockIdx.x, lane = threadIdx.x & 63;
    // (wave-uniform by construction; said so, so that the two roles are two scalar branches and not two exec masks)
    int role = RING ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
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
    T.c0 = cf32{relay_lane(T.c0.x, 0), relay_lane(T.c0.y, 0)}; T.c1 = cf32{relay_lane(T.c1.x, 0), relay_lane(T.c1.y, 0)};
    dead = __builtin_amdgcn_readfirstlane((int)dead) != 0;
    const RelaySeg prev = a.start[s];
    const int x_lo = (int)(T.ii > 4 ? T.ii - 4 : 0);
    if (threadIdx.x == 0) { sh_xhi = x_lo; sh_pos_ii = (int)T.ii; sh_done = 0; sh_ghi = 0; sh_pos_n = 0; }
    __syncthreads();      // (the only barrier: both waves pass it before either can leave)
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
    // Which of the two waves walks.  A walker is one wave whose every instruction waits for the one before; of two waves
    // on a SIMD the older one issues first, and a walker that the hardware has put on the SIMD of an older walker (the
    // third workgroup of a CU, in one CU out of ten) takes a third longer than the others and sets the pace of the pass.
    // So the workgroup looks where its two waves sit (HW_ID) and which SIMDs of the CU hold a walker already: if wave 0's
    // is taken and wave 1's is free, the waves swap roles.  (Both waves are still here: every return above is taken by
    // the whole workgroup.)
    unsigned *claim = nullptr;
    if (RING && a.simd_claim) {
        const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));       // HW_REG_HW_ID: SIMD_ID 5:4; CU_ID, SH_ID, SE_ID in 15:8
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
    // the record of the walk before (this call's: the flags are cleared when a call's relay starts): prev.n_done symbols,
    // read indices relative to a reference that does not depend on the pass
    const bool use_rec = RING && a.rec != nullptr && a.rec_use &&
                         __builtin_amdgcn_readfirstlane((int)((prev.flags & RELAY_REC) != 0 && prev.n_done > 0)) != 0;
    const int n_rec = __builtin_amdgcn_readfirstlane(prev.n_done);
    // (the nominal period the records are relative to: does not depend on the pass)
    const int wnom = (int)(a.par.omega_mid * 16777216.0f), wint_n = wnom >> 24, wfrac_n = wnom & 0xffffff;
    unsigned *recs = a.rec ? a.rec + obase : nullptr;

    if (RING && role == 1) {
        // ---- the prefetcher: keeps [position, position + RX - XCH) of the samples in the ring; a slot is overwritten
        // only when the walker's published position is past it.  Same for the first guesses, by symbol number.
        const long long nlast = a.N > 0 ? a.N - 1 : 0;
        int x_hi = x_lo, g_hi = 0;
        unsigned rounds = 0;
        while (!relay_ld(&sh_done)) {
            if (++rounds > (1u << 24)) { if (lane == 0) a.changed[RELAY_STAT * pass + 3] = 0xc0000000u | (unsigned)s; break; }   // watchdog
            const int pii = relay_ld(&sh_pos_ii);
            // (... and not beyond what a walker standing on the last sample can ask for: its block's span)
            const bool fx = x_hi + RELAY_XCH - RELAY_RX <= pii && (long long)x_hi <= nlast + span + 16;
            bool fg = false;
            if (use_rec) fg = g_hi < Lseg && g_hi + RELAY_GCH - RELAY_GR <= relay_ld(&sh_pos_n);
            if (!fx && !fg) { __builtin_amdgcn_s_sleep(4); continue; }
            if (fx) {
                cf32 vx[RELAY_XCH / 64];
#pragma unroll
                for (int q = 0; q < RELAY_XCH / 64; ++q) {
                    const long long i = (long long)x_hi + lane + 64 * q;
                    vx[q] = xs[i < nlast ? i : nlast];
                }
#pragma unroll
                for (int q = 0; q < RELAY_XCH / 64; ++q) {
                    const int slot = (x_hi + lane + 64 * q) & (RELAY_RX - 1);
                    xr[slot] = vx[q];
                    if (slot < RELAY_XMIR) xr[RELAY_RX + slot] = vx[q];
                }
                x_hi += RELAY_XCH;
                if (lane == 0) relay_st(&sh_xhi, x_hi);
            }
            if (fg) {
                unsigned vg[RELAY_GCH / 64];
#pragma unroll
                for (int q = 0; q < RELAY_GCH / 64; ++q) {
                    const int m = g_hi + lane + 64 * q;
                    vg[q] = m < n_rec ? recs[m] : RELAY_NOPOS;
                }
#pragma unroll
                for (int q = 0; q < RELAY_GCH / 64; ++q) gr[(g_hi + lane + 64 * q) & (RELAY_GR - 1)] = vg[q];
                g_hi += RELAY_GCH;
                if (lane == 0) relay_st(&sh_ghi, g_hi);
            }
        }
        return;
    }

    // ---- the walker
    if (lane == 0) {
        atomicAdd(&a.changed[RELAY_STAT * pass], 1u);
        // how far this start is from the one the segment was last walked from (samples): what the automatic closure
        // looks at (ClockStage::finish).  Non-negative floats order like their bits; watchdog marks stay on top.
        if (pass > 0 && (prev.flags & (RELAY_WALKED | RELAY_APPROX))) {
            const float mv = fabsf(clock_tdiff(prev.s, T));
            atomicMax(&a.changed[RELAY_STAT * pass + 3], __float_as_uint(mv));
            // (sum of squares in units of 2^-40 sample^2, moves beyond a sample count as one)
            const float m1 = fminf(mv, 1.0f);
            atomicAdd(reinterpret_cast<unsigned long long *>(&a.changed[RELAY_STAT * pass + 4]),
                      (unsigned long long)(m1 * m1 * 1099511627776.0f));
            atomicAdd(&a.changed[RELAY_STAT * pass + 6], 1u);
        }
    }
    const bool quiet = a.sym_skip != 0 && s > 0;
    // (kernel arguments used inside the walk are taken once: behind the hand-written LDS traffic the compiler reloads them)
    const bool rec_write = __builtin_amdgcn_readfirstlane(a.rec_write) != 0;
    const bool g_ahead = (__builtin_amdgcn_readfirstlane(a.rec_use) & 2) != 0;
    const ClockState T0 = T;
    // (the integer model is a guess generator, not the arithmetic: the gains folded into one factor each, the lattice
    // steps -- powers of two, float32 spacings -- as shifts)
    const float gkw = a.par.gain_omega * (16777216.0f / (float)a.q_om), gkm = a.par.gain_mu * (16777216.0f / (float)a.q_mu);
    const int sh_om = 31 - __builtin_clz((unsigned)a.q_om), sh_mu = 31 - __builtin_clz((unsigned)a.q_mu);
    // symbols of this segment the output holds
    const long long room = (long long)a.cap - obase;
    const int n_out = room <= 0 ? 0 : (room < (long long)Lseg ? (int)room : Lseg);
    float *softs = a.soft ? a.soft + obase : nullptr;
    float2 *syms = (SYM && a.sym) ? a.sym + obase : nullptr;
    const int ni_w = (int)(a.ni < 0x7fffffffLL ? a.ni : 0x7fffffffLL);     // (read indices are 32-bit here: x_lo, need_x)
    int n = 0;
    unsigned steps = 0, rounds_total = 0;
    bool exhausted = false, stuck = false;
    int x_hi = x_lo, g_hi = 0;           // what the rings are known to hold (asked again only when that is not enough)
    unsigned g_next = 0u;                // the record of the next block, read ahead
    bool g_have = false;
    float rec_slope = 0.0f;              // how much further than the recorded walk this one advanced per symbol over the block before (2^-24 sample)
    float m1 = 0.f, m2 = 0.f;            // per lane: sum |s|, sum s^2 of the symbols it committed (first pass only)
#ifdef XRIT_RELAY_TIMING
    unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
    const unsigned long long tbegin = tlast;
#endif
    while (n < Lseg) {
        RELAY_TICK(5);
        ++steps;          // (every step commits at least one symbol or ends the walk: no more than Lseg steps)
        // (a walker that stands beyond the input has nothing to wait for: the prefetcher stops at the end of the input)
        const int ii0 = (int)T.ii;
        if ((unsigned)ii0 >= (unsigned)ni_w) { exhausted = true; break; }
        const int need_x = ii0 + span + 8;
        const int need_g = n + 64 < Lseg ? n + 64 : Lseg;
        if (RING && (x_hi < need_x || (use_rec && g_hi < need_g))) {
            // wait until the rings hold this step's samples and guesses (they do, unless memory is slower than ten steps)
            int spins = 0;
#pragma nounroll
            while (x_hi < need_x) {
                x_hi = relay_ld(&sh_xhi);
                if (x_hi >= need_x) break;
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1 << 22)) { if (lane == 0) a.changed[RELAY_STAT * pass + 3] = 0x80000000u | (unsigned)s; stuck = true; break; }
            }
#pragma nounroll
            while (use_rec && !stuck && g_hi < need_g) {
                g_hi = relay_ld(&sh_ghi);
                if (g_hi >= need_g) break;
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1 << 22)) { if (lane == 0) a.changed[RELAY_STAT * pass + 3] = 0x90000000u | (unsigned)s; stuck = true; break; }
            }
            if (stuck) break;
        }
        RELAY_TICK(0);
        // the walker's state on the lattice; the first guess puts symbol n + lane where the walk before had it, or,
        // without one, at the nominal rate
        const int mu0u = (int)(T.mu * 16777216.0f), W0 = (int)(T.omega * 16777216.0f);
        const int wint = W0 >> 24, wfrac = W0 & 0xffffff;
        const int fr0 = mu0u + lane * wfrac, bii = ii0 + lane * wint;     // symbol n + lane at the walker's rate
        int cii, carm;
        float cmu, com = T.omega;
        cii = bii + (fr0 >> 24);
        cmu = (float)(fr0 & 0xffffff) * (1.0f / 16777216.0f);
        if (lane == 0) cmu = T.mu;
        carm = (int)rintf(cmu * (float)XR_MM_NSTEPS);
        int rinc = 0;                  // recorded advances summed up to and including this lane (blocks that guess from the record)
        bool rec_block = false;
        if (use_rec) {
            // The record holds, per symbol, how far the walk before ADVANCED behind it (omega + gain_mu mm, as the difference
            // from the nominal period, units of 2^-24 sample: 32 bits).  Summed over the lanes in front (one more prefix sum)
            // and anchored at this walk's own state at lane 0, that is where the symbols sat relative to one another -- this
            // walk's trajectory runs beside that one at a distance that changes slowly (they are merging: a damped
            // oscillation, smooth over 64 symbols), so the rate at which the distance changed over the block before is added
            // as a slope.  (Round 4, late.  With (index, arm) records a walk whose start had moved by half an arm
            // re-interpolated every block of its segment, 2.0 guess rounds per step where the others took 1.1 -- and a pass
            // waits for its slowest walker; and the record of a walk from the timing guess, five arms away, was of no use at
            // all: 2.30 rounds per step in the second pass, now 1.3.)
            // (taken at the end of the step before where the ring already held it: the LDS latency is behind the loop head)
            const unsigned g = g_have ? g_next : gr[(n + lane) & (RELAY_GR - 1)];
            rec_block = !__any(g == RELAY_NOPOS);
            if (rec_block) {
                rinc = relay_scan((int)g, lane);
                const int rex = rinc - (int)g;
                const int frg = mu0u + lane * wfrac_n + rex + (int)rintf(rec_slope * (float)lane);
                if (lane != 0) {
                    cii = ii0 + lane * wint_n + (frg >> 24);
                    carm = ((frg & 0xffffff) + (1 << 16)) >> 17;
                }
            } else {
                rec_slope = 0.0f;
            }
        }
        RELAY_TICK(1);
        cf32 p0{0.f, 0.f};
        float mm = 0.f;
        ClockState hs{};               // the history symbol n + lane sees: (p0, p1) of the two symbols in front of it
        bool stale = false, inrange = true;
        const int max_rounds = apx == 1 ? 1 : (apx == 2 ? 2 : RELAY_ROUNDS);
        for (int round = 0; round < max_rounds; ++round) {
            ++rounds_total;
            // (re)interpolate where the read index or the arm moved
            inrange = cii >= ii0 && cii + XR_MM_NTAPS <= need_x;
            cf32 w[XR_MM_NTAPS];
            if (RING) {
                const cf32 *wp = xr + (cii & (RELAY_RX - 1));
#pragma unroll
                for (int q = 0; q < XR_MM_NTAPS; ++q) w[q] = wp[q];
            } else {
                long long wi = cii;
                wi = wi < 0 ? 0 : (wi >= a.ni ? a.ni - 1 : wi);
#pragma unroll
                for (int q = 0; q < XR_MM_NTAPS; ++q) w[q] = xs[wi + q];
                inrange = true;
            }
            p0 = clock_interp_arm(w, table, carm);
            // the two symbols in front: the neighbours' interpolated values, the walker's own history in front of lane 0
            const cf32 sl{p0.x > 0.f ? 1.f : 0.f, p0.y > 0.f ? 1.f : 0.f};
            hs.p0 = cf32{relay_shr1_from(p0.x, T.p0.x), relay_shr1_from(p0.y, T.p0.y)};
            hs.p1 = cf32{relay_shr1_from(hs.p0.x, T.p1.x), relay_shr1_from(hs.p0.y, T.p1.y)};
            hs.c0 = cf32{relay_shr1_from(sl.x, T.c0.x), relay_shr1_from(sl.y, T.c0.y)};
            hs.c1 = cf32{relay_shr1_from(hs.c0.x, T.c1.x), relay_shr1_from(hs.c0.y, T.c1.y)};
            mm = clock_timing_error(p0, hs);
            // omega and mu on the lattice: additions of rounded increments, i.e. two prefix sums
            const int dW = (int)rintf(mm * gkw) << sh_om;
            const int dM = (int)rintf(mm * gkm) << sh_mu;
            const int C = relay_scan(dW, lane);                  // omega after symbol n + lane, minus W0
            const int E = C + dM;
            const int D = relay_scan(E, lane) - E;               // position in front of symbol n + lane, minus the nominal one
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
        const bool good = (!stale || apx) && inrange;             // this lane's interpolation belongs to its state (apx: near enough)
        const bool ok = good && exists && lane < 63 && (apx || ((int)st.ii == nxt_ii && st.mu == nxt_mu && st.omega == nxt_om));
        const unsigned long long okm = __ballot(ok), exm = __ballot(exists), gdm = __ballot(good);
        const int m = ~okm ? __builtin_ctzll(~okm) : 64;         // lanes 0 .. m start from verified states
        const int e = ~exm ? __builtin_ctzll(~exm) : 64;         // first lane whose symbol does not exist
        const int g = ~gdm ? __builtin_ctzll(~gdm) : 64;         // first lane whose own step is not to be trusted
        int nv = m + 1 < 64 ? m + 1 : 64;
        const int lim = Lseg - n;
        nv = nv < lim ? nv : lim;
        nv = nv < g ? nv : g;
        if (e < nv) { nv = e; exhausted = true; }
        RELAY_TICK(3);
        if (pass == 0 && lane < nv) { m1 += fabsf(p0.x); m2 += p0.x * p0.x; }
        if (lane < nv && apx != 1) {
            const int o = n + lane;
            if (o < n_out && !apx && !quiet) {
                if (softs) softs[o] = p0.x;
                if (SYM && syms) syms[o] = make_float2(p0.x, p0.y);
            }
            if (RING && recs && rec_write) {
                // (this symbol's advance: the lane's own literal step, already verified)
                const int dev = (((int)st.ii - cii - wint_n) << 24) + ((int)(st.mu * 16777216.0f) - (int)(cmu * 16777216.0f)) - wfrac_n;
                recs[o] = (unsigned)dev != RELAY_NOPOS ? (unsigned)dev : 0u;
            }
        }
        if (nv > 0) {
            const int src = nv - 1;
            ClockState nt;
            nt.ii = __builtin_amdgcn_readlane((int)st.ii, src);
            nt.mu = relay_lane(st.mu, src);
            nt.omega = relay_lane(st.omega, src);
            nt.p0 = cf32{relay_lane(st.p0.x, src), relay_lane(st.p0.y, src)};
            nt.p1 = cf32{relay_lane(st.p1.x, src), relay_lane(st.p1.y, src)};
            nt.c0 = cf32{relay_lane(st.c0.x, src), relay_lane(st.c0.y, src)};
            nt.c1 = cf32{relay_lane(st.c1.x, src), relay_lane(st.c1.y, src)};
            if (rec_block) {
                // (what this walk advanced over the block beyond what the record said, per symbol: the next block's slope)
                const int act = (((int)nt.ii - ii0 - nv * wint_n) << 24) + ((int)(nt.mu * 16777216.0f) - mu0u) - nv * wfrac_n;
                const float over = (float)(act - __builtin_amdgcn_readlane(rinc, src));
                rec_slope = nv == 64 ? over * (1.0f / 64.0f) : over / (float)nv;
            }
            T = nt;
            n += nv;
            g_have = use_rec && g_ahead && g_hi >= (n + 64 < Lseg ? n + 64 : Lseg);
            if (g_have) g_next = gr[(n + lane) & (RELAY_GR - 1)];
            if (RING && lane == 0) { relay_st(&sh_pos_ii, (int)T.ii); relay_st(&sh_pos_n, n); }
        }
        RELAY_TICK(4);
        if (exhausted || nv == 0) { exhausted = true; break; }      // (nv == 0 without exhaustion cannot happen: lane 0 is good)
    }
#ifdef XRIT_RELAY_TIMING
    if (lane == 0) {
        const int o = pass >= 8 ? 8 : 0;
        for (int q = 0; q < 6; ++q) atomicAdd(&relay_dbg[o + q], tacc[q]);
        atomicAdd(&relay_dbg[o + 6], (unsigned long long)steps);
        if (pass < RELAY_WDBG_PASSES && s < RELAY_WDBG_SEGS) {
            unsigned *w = relay_wdbg + ((size_t)pass * RELAY_WDBG_SEGS + s) * RELAY_WDBG_WORDS;
            w[0] = (unsigned)(__builtin_amdgcn_s_memtime() - tbegin);
            w[1] = (unsigned)tacc[0];
            w[2] = steps;
            w[3] = rounds_total;
            w[4] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));      // HW_REG_HW_ID
            w[5] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));     // HW_REG_XCC_ID
        }
    }
#endif
    if (pass == 0 && a.moments) {
        // (lane sums in step order, lanes added in a fixed tree, segments by integer atomics: the same value run after run)
        for (int off = 32; off > 0; off >>= 1) { m1 += __shfl_xor(m1, off, 64); m2 += __shfl_xor(m2, off, 64); }
        if (lane == 0) {
            atomicAdd(&a.moments[0], (unsigned long long)((double)m1 * 1048576.0));
            atomicAdd(&a.moments[1], (unsigned long long)((double)m2 * 1048576.0));
        }
    }
    if (lane == 0) {
        if (RING) relay_st(&sh_done, 1);
        if (RING && claim && sh_claim >= 0) atomicAnd(claim, ~(1u << sh_claim));      // (the next pass finds the CU's word clear)
        atomicAdd(&a.changed[RELAY_STAT * pass + 1], steps);
        atomicAdd(&a.changed[RELAY_STAT * pass + 2], rounds_total);
        RelaySeg st0{};
        st0.s = T0;
        st0.n_done = n;                 // symbols the record holds
        st0.flags = (apx || quiet) ? (RELAY_APPROX | ((apx == 2 || !apx) && rec_write ? RELAY_REC : 0)) : (RELAY_WALKED | (rec_write ? RELAY_REC : 0));
        a.start[s] = st0;
        RelaySeg e{};
        e.s = T;
        e.n_done = n;
        // (a walk a watchdog ended is not the end of the input: its own flag, and the call fails -- clock_relay_finalize_kernel)
        e.flags = stuck ? RELAY_STUCK : (exhausted ? RELAY_EXHAUSTED : 0);
        eout[s] = e;
    }
}

// first relay pass of a call: nothing has been walked
__global__ void __launch_bounds__(256) clock_relay_init_kernel(RelaySeg *start, int G, unsigned *changed, int npass, int *ctl,
                                                              unsigned long long *moments, unsigned *simd_claim)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (simd_claim && i < RELAY_CLAIM_WORDS) simd_claim[i] = 0u;
    if (i < G) start[i].flags = 0;
    if (i < RELAY_STAT * npass) changed[i] = 0u;
    if (i == 0) { ctl[10] = 0; ctl[11] = 0; ctl[12] = 0; ctl[14] = 0; ctl[15] = 0; moments[0] = 0ull; moments[1] = 0ull; }
}

}  // namespace xrit
