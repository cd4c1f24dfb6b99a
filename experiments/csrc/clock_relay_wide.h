// clock_relay_wide.h -- the relay's walker as a team of W waves: 62 W symbols per step.
//
// clock_relay.h walks a segment of the clock recovery (ClockRecovery::Work, /root/reference/demodulator/src/
// demodulator.cpp:156,449) exactly, 64 symbols per step, with ONE wave: guess where the next 64 symbols sit, form their
// timing errors side by side, two integer prefix sums give every lane its state on the float32 lattice, the literal
// float32 step verifies.  That wave issues one instruction every 5..7 cycles and a pass over all segments takes
// (segment length / 64) steps whatever the chip could do beside it.  Here W waves of one workgroup, one per SIMD, take a
// step together:
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
//  * the waves feed themselves: at the top of a step each wave loads its share of the samples (and first guesses) the team
//    will read two steps later into registers and drops them into the LDS rings at the top of the next step -- a whole
//    step hides the memory latency, no wave is set aside for it, and so the team can meet at s_barrier (two per round:
//    after the sums, after the verdicts) instead of polling mailbox words.  The symbols a step commits are stored
//    behind the next step's loads, so that the wait in front of a drop never includes a young store.
#pragma once

#include "../../xritdemod_amd/csrc/clock_relay.h"

namespace xrit {

constexpr int RW_OWN = 62;            // symbols a wave owns per step
constexpr int RW_SLOT = 8;            // mailbox words per wave
constexpr int RW_REC_PAD = 2048;      // words behind the last segment's records a refill may touch
constexpr int RW_XLOADS = 8;          // samples a lane loads per refill unit (a wave: 512 samples: more than its share of a step)

template <int W> struct RelayWide {
    static constexpr int RX = W <= 2 ? 4096 : 8192;                       // sample ring
    static constexpr int GR = 1024;                                       // ring of first guesses (symbols)
    static constexpr int XU = 64 * RW_XLOADS * W;                         // samples per refill unit
    // samples a block may span at most: the ring holds three blocks (the one being walked, the next one, and the one whose
    // samples are dropped in at the end of this step -- readable from the step after next) and a unit of slack
    static constexpr int MAX_SPAN = (RX - XU - 64) / 3 < XU - 32 ? (RX - XU - 64) / 3 : XU - 32;     // (... and a unit refills more than a step uses up)
};

// the team meets: LDS traffic of every wave is done (the mailbox, the rings), nothing else is waited for -- global
// loads and stores stay in flight across it (__syncthreads() would drain them)
__device__ __forceinline__ void rw_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
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
// (instrumented build: cycles per phase of a step summed over wave 1's of all teams -- relay_dbg[0..6] --, steps in [7])
#define RW_TICK(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); wacc[i] += t_ - wlast; wlast = t_; } while (0)
#else
#define RW_TICK(i) do { } while (0)
#endif

// Segment s of the call, walked by W waves.
template <bool SYM, int W>
__global__ void __launch_bounds__(64 * W) clock_relay_wide_kernel(RelayArgs a, int pass, int span)
{
    using CFG = RelayWide<W>;
    constexpr int RX = CFG::RX, GR = CFG::GR, XU = CFG::XU, GU = 64 * W;
    if (a.ctl && !a.ctl[0]) return;                              // the tiled hand-off has not closed: nothing to refine yet
    if (pass > 0 && a.changed[RELAY_STAT * (pass - 1)] == 0) return;      // closed in an earlier pass
    __shared__ __attribute__((aligned(16))) float table[(XR_MM_NSTEPS + 1) * XR_MM_NTAPS];
    __shared__ __attribute__((aligned(16))) cf32 xr[RX + RELAY_XMIR];
    __shared__ __attribute__((aligned(16))) unsigned gr[GR];
    __shared__ __attribute__((aligned(16))) int mailA[RW_SLOT * W];          // scan totals
    __shared__ __attribute__((aligned(16))) int mailC[RW_SLOT * W];          // verdicts
    clock_table_to_lds(table, a.table);
    const int s = blockIdx.x, lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
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
    __syncthreads();      // (the table is in place)
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
    const cf32 *xs = reinterpret_cast<const cf32 *>(a.x);
    const int ni_w = (int)(a.ni < 0x7fffffffLL ? a.ni : 0x7fffffffLL);     // (read indices are 32-bit here)
    const int nlast = (int)(a.N > 0 ? (a.N - 1 < 0x7fffffffLL ? a.N - 1 : 0x7fffffffLL) : 0);

    if (threadIdx.x == 0) {
        atomicAdd(&a.changed[RELAY_STAT * pass], 1u);
        // how far this start is from the one the segment was last walked from (samples): what the automatic closure
        // looks at (ClockStage::finish).  Non-negative floats order like their bits.
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
    const bool hist_t = wv == 0 && lane < 2;      // wave 0's history lanes: the walker's own p1, p0
    const bool first = wv == 0 && lane == 2;      // the block's first symbol starts from the walker's state itself

    // ---- the rings.  Samples [x_lo, x_wr) and first guesses [0, g_wr) have been dropped in; what was dropped in before the
    // team's last meeting can be read.  A unit (XU samples / GU guesses: more than a step uses up) is loaded at the top of
    // step k, dropped in at the top of step k + 1 -- a whole step for the memory to answer -- and read from step k + 2 on.
    const int x_lo = (int)(T.ii > 4 ? T.ii - 4 : 0) & ~(XU - 1);       // (units start on multiples of XU: the ring's slot 0 too)
    int x_wr = x_lo, g_wr = 0;
    cf32 xst[RW_XLOADS];
    unsigned gst = RELAY_NOGUESS;
    bool x_pend = false, g_pend = false;          // a unit is loaded and not yet dropped in
    auto load_x = [&]() {
#pragma unroll
        for (int q = 0; q < RW_XLOADS; ++q) {
            const unsigned i = (unsigned)(x_wr + 64 * RW_XLOADS * wv + lane + 64 * q);
            xst[q] = xs[i < (unsigned)nlast ? i : (unsigned)nlast];
        }
    };
    auto drop_x = [&]() {
        const int slot0 = (x_wr & (RX - 1)) + 64 * RW_XLOADS * wv + lane;
#pragma unroll
        for (int q = 0; q < RW_XLOADS; ++q) xr[slot0 + 64 * q] = xst[q];
        // (the ring's first samples again behind its end, so that a window never wraps)
        if ((x_wr & (RX - 1)) == 0 && wv == 0 && lane < RELAY_XMIR) xr[RX + lane] = xst[0];
    };
    auto load_g = [&]() {
        const int m = g_wr + 64 * wv + lane;
        gst = m < n_rec ? recs[(unsigned)m] : RELAY_NOGUESS;
    };
    auto drop_g = [&]() { gr[(g_wr + 64 * wv + lane) & (GR - 1)] = gst; };
    // prologue: what the first two steps read, loaded and dropped in at once
    {
#pragma nounroll
        while (x_wr < (int)T.ii + 2 * span + 16) { load_x(); drop_x(); x_wr += XU; }
        if (use_rec) {
#pragma nounroll
            while (g_wr < 2 * RW_OWN * W && g_wr < n_rec) { load_g(); drop_g(); g_wr += GU; }
        }
    }
    rw_barrier();
    int x_vis = x_wr, g_vis = g_wr;               // what every wave may read

    // the symbols a step commits leave at the top of the NEXT step, between the drop and the new loads: the wait in front of
    // a drop is then for loads and stores of a step ago and nothing younger
    float pc_x = 0.f, pc_y = 0.f;
    unsigned pc_rec = 0, pc_o = 0;
    bool pc_soft = false, pc_any = false;
    auto flush_commit = [&]() {
        if (pc_any) {
            if (pc_soft) {
                if (softs) softs[pc_o] = pc_x;
                if (SYM && syms) syms[pc_o] = make_float2(pc_x, pc_y);
            }
            if (recs) recs[pc_o] = pc_rec;
        }
        pc_any = false;
    };

    int n = 0;
    unsigned steps = 0, rounds_total = 0;
    bool exhausted = false, stuck = false;
    float m1 = 0.f, m2 = 0.f;                     // per lane: sum |s|, sum s^2 of the symbols it committed (first pass only)
#ifdef XRIT_RELAY_TIMING
    unsigned long long wacc[7] = {0, 0, 0, 0, 0, 0, 0}, wlast = __builtin_amdgcn_s_memtime();
#endif
    while (n < Lseg) {
        RW_TICK(6);
        ++steps;
        const int ii0 = (int)T.ii;
        if ((unsigned)ii0 >= (unsigned)ni_w) { exhausted = true; break; }
        // what the step before loaded goes into the rings (every wave is past its reads of that step; what is overwritten lies
        // behind the walker) ...
        x_vis = x_wr; g_vis = g_wr;
        if (x_pend) { drop_x(); x_wr += XU; }
        if (g_pend) { drop_g(); g_wr += GU; }
        // ... the symbols of the step before leave (nothing is in flight at this point: whatever the compiler makes these
        // stores wait for costs nothing) ...
        flush_commit();
        // ... and the next unit is asked for, as far as the ring has room (a slot is free once the walker stands behind it)
        // (the guesses first: this compiler waits for everything in flight before it reuses their register)
        g_pend = use_rec && g_wr < n + 3 * RW_OWN * W && g_wr < n_rec && g_wr + GU - GR <= n;
        if (g_pend) load_g();
        x_pend = x_wr < ii0 + 3 * span + 16 && x_wr + XU - RX <= ii0 && x_wr <= nlast + span + 16;
        if (x_pend) load_x();
        const int need_x = ii0 + span + 8;
        const int blk = n + RW_OWN * W < Lseg ? n + RW_OWN * W : Lseg;
        const int need_g = blk < n_rec ? blk : n_rec;
        // (the refill rule keeps what can be read a block ahead of the walker -- a unit is more than a step uses up, and one is
        // asked for whenever less than three blocks are in place; if that ever fails the walk gives up rather than read
        // samples that are not there)
        if (x_vis < need_x || (use_rec && g_vis < need_g)) { stuck = true; break; }
        RW_TICK(0);
        // the walker's state on the lattice; this wave's lane 2 at the walker's rate (64-bit on the scalar side), the lanes
        // relative to it
        const int mu0u = (int)(T.mu * 16777216.0f), W0 = (int)(T.omega * 16777216.0f);
        const int wint = W0 >> 24, wfrac = W0 & 0xffffff;
        const long long fb = (long long)mu0u + (long long)jw * (long long)wfrac;
        const int bii_w = ii0 + jw * wint + (int)(fb >> 24), fr_w = (int)(fb & 0xffffff);
        const int fr0 = fr_w + jl * wfrac, bii = bii_w + jl * wint;
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
        int nv = 0;
        bool any_exh = false;
        RW_TICK(1);
        for (int round = 0; round < RELAY_ROUNDS; ++round) {
            ++rounds_total;
            const bool inrange = hist_t || (cii >= ii0 && cii + XR_MM_NTAPS <= need_x);
            {
                cf32 w[XR_MM_NTAPS];
                const cf32 *wp = xr + (cii & (RX - 1));
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
                if (lane == 63) rw_st4(&mailA[RW_SLOT * wv], lc, ldx + le, l62, ldx);
            }
            RW_TICK(2);
            rw_barrier();
            int Cp = 0, Dp = 0, Dh0 = 0, Dh1 = 0;
            if (wv > 0) {
                const int v = rw_ld_lane(&mailA[lane < RW_SLOT * W ? lane : 0]);
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
                const int packed = (int)((wstale ? 0x200u : 0u) | (exh ? 0x100u : 0u) | (full ? 0x80u : 0u) | (unsigned)c);
                if (lane == wl) {
                    int *slot = &mailC[RW_SLOT * wv];
                    rw_st4(slot, (int)st.ii, __float_as_int(st.mu), __float_as_int(st.omega), __float_as_int(st.p0.x));
                    rw_st4(slot + 4, __float_as_int(st.p0.y), __float_as_int(st.p1.x), __float_as_int(st.p1.y), packed);
                }
            }
            RW_TICK(4);
            rw_barrier();
            const int v = rw_ld_lane(&mailC[lane < RW_SLOT * W ? lane : 0]);
            bool any_stale = false;
            int src = -1;
            nv = 0; any_exh = false;
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
            RW_TICK(5);
            // (everybody has read the verdicts and the sums before anybody writes the next round's: the next meeting is
            // behind the next round's sums, and those are written only by waves that have passed this point -- a wave that
            // is still reading mailA of this round has not met the others at the verdicts yet)
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
        // the verified prefix of the block: its symbols leave at the top of the next step
        {
            const int jb = jw + jl;            // this lane's symbol within the block
            pc_any = owned && jb < nv;
            if (pc_any) {
                if (pass == 0) { m1 += fabsf(p0.x); m2 += p0.x * p0.x; }
                const unsigned rel = (unsigned)(cii - ref);
                pc_rec = rel < (1u << 24) ? (rel << 8) | (unsigned)carm : RELAY_NOGUESS;
                pc_o = (unsigned)(n + jb);
                pc_soft = n + jb < n_out;
                pc_x = p0.x; pc_y = p0.y;
            }
        }
        n += nv;
        if (any_exh || nv == 0) { exhausted = true; break; }      // (nv == 0 without exhaustion cannot happen: the first lane is good)
    }
    flush_commit();
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
