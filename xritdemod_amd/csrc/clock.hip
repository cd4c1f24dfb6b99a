// clock.hip -- Mueller & Mueller symbol-timing recovery over symbol-count chains.
// Replaces SatHelper::ClockRecovery::Work (/root/reference/demodulator/src/
// demodulator.cpp:156; object built at :449 with Parameters.h:30-33).  The
// recurrence carries (read index ii, mu, omega) plus two symbols of history and
// advances ii by floor(mu) each symbol, so the output rate is data dependent.
//
// Tiling: chain k produces symbols [k*NS, (k+1)*NS) -- a fixed symbol count, so
// every chain knows where its output goes and the map "start state -> end
// state" is smooth apart from the 1/128 interpolator-arm quantisation.  Start
// positions are guessed from an Oerder & Meyr timing estimate (|x|^2 line at the
// symbol rate, unwrapped over the call), then corrected by Newton steps on the
// multiple-shooting system; the Jacobian of each chain comes from two extra
// lanes that run the chain from (t+h_t, omega) and (t, omega+h_w).  A residual of
// m whole symbols at a boundary is carried as a symbol slip that shifts all later
// chains, not "corrected".  The arm quantisation makes the recurrence chaotic at
// the 1e-5 level in mu, so the passes stop when the residuals
// stop falling (device-side test in ClockPolicy::decide) rather than at bitwise closure.
#include "kernels.h"

#include <cstdlib>
#include <type_traits>
#include "scan.h"
#include "newton.h"

namespace xrit {

constexpr int CLK_OM_BLOCK = 256;     // samples per timing-estimate block
constexpr float CLK_H_T = 0.0625f;    // finite-difference steps
constexpr float CLK_H_W = 1e-3f;

struct ClockResult {
    unsigned long long n_symbols;
    long long ii_final;
    int terminal_chain;
    int ok;
};

// --------------------------------------------------------- timing estimate
__global__ void __launch_bounds__(256) clock_om_kernel(const float2 *__restrict__ x, double2 *__restrict__ X,
                                                       long long N, int nb, double inv_sps)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long b = (long long)blockIdx.x * 4 + wave;
    if (b >= nb) return;
    float sr = 0.f, si = 0.f;
    for (int i = lane; i < CLK_OM_BLOCK; i += 64) {
        long long j = b * CLK_OM_BLOCK + i;
        if (j < N) {
            float2 v = x[j];
            float p = v.x * v.x + v.y * v.y;
            double ph = (double)j * inv_sps;
            ph -= floor(ph);
            float sn, cs;
            sincosf(-6.28318530717958647692f * (float)ph, &sn, &cs);
            sr += p * cs;
            si += p * sn;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        sr += __shfl_down(sr, off, 64);
        si += __shfl_down(si, off, 64);
    }
    if (lane == 0) X[b] = make_double2((double)sr, (double)si);
}

__device__ __forceinline__ double clk_wrap(double x) { return x - 2.0 * XR_PI_D * rint(x / (2.0 * XR_PI_D)); }

struct ClkUnwrapF {
    typedef double T;
    const double2 *X;
    double *cnt;      // out: symbol count (continuous) at block centres
    int nb;
    double sps;
    double off;       // block b covers buffer samples [off + b*BL, off + (b+1)*BL)
    int BL;
    double rot;       // the phasor of X counts samples from buffer sample `off`, not 0: X = X_0 e^{+j rot}, rot = 2 pi off / sps
    __device__ T identity() const { return 0.0; }
    __device__ T combine(const T &lo, const T &hi) const { return lo + hi; }
    __device__ double ang(long long b) const
    {
        double sr = 0, si = 0;
        for (long long q = b - 2; q <= b + 2; ++q)
            if (q >= 0 && q < nb) { sr += X[q].x; si += X[q].y; }
        return atan2(si, sr);
    }
    // (a thread's run takes every angle once: the differences chain through `prev`)
    __device__ T reduce_run(long long i0, int n) const
    {
        double s = 0, prev = i0 == 0 ? rot : ang(i0 - 1);
        for (int k = 0; k < n; ++k) {
            const double cur = ang(i0 + k);
            s += clk_wrap(cur - prev);
            prev = cur;
        }
        return s;
    }
    __device__ void apply_run(long long i0, int n, const T &pre) const
    {
        double s = pre, prev = i0 == 0 ? rot : ang(i0 - 1);
        for (int k = 0; k < n; ++k) {
            const double cur = ang(i0 + k);
            s += clk_wrap(cur - prev);
            prev = cur;
            double cb = off + ((double)(i0 + k) + 0.5) * BL;
            cnt[i0 + k] = (cb + s / (2.0 * XR_PI_D) * sps) / sps;
        }
    }
};

__device__ __forceinline__ double clk_count_at(const double *cnt, int nb, double sps, double t, double off, int BL)
{
    double fb = (t - off) / BL - 0.5;
    int b = (int)floor(fb);
    b = max(0, min(nb - 2, b));
    if (nb < 2) return cnt[0] + (t - off - 0.5 * BL) / sps;
    double c0 = cnt[b], c1 = cnt[b + 1];
    return c0 + (c1 - c0) * (fb - b);
}

// start state of every chain from the unwrapped symbol-count curve
__global__ void clock_guess_kernel(const double *__restrict__ cnt, int nb, double sps, ClockState *__restrict__ S,
                                   const ClockState *__restrict__ carried, int K, int NS, float omega0,
                                   const float2 *__restrict__ x, const float *__restrict__ table, long long ni,
                                   double off, int BL, int *__restrict__ dirty, int *__restrict__ ctl, int ctl_words,
                                   int *__restrict__ terminal, int *__restrict__ written)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x == 0) {
        // the call's control block and per-pass counters start from zero, "no chain has run out of input yet"
        for (int i = threadIdx.x; i < ctl_words; i += blockDim.x) ctl[i] = 0;
        if (threadIdx.x == 0) *terminal = 0x7fffffff;
        __syncthreads();
    }
    if (k >= K) return;
    dirty[k] = 1;                       // every chain runs in the first pass
    written[k] = 0;                     // ... and none has left its symbols yet
    ClockState s0 = carried[0];
    if (k == 0) { S[0] = s0; ctl[5] = 1; return; }     // first solve: gated
    double t0 = (double)s0.ii + (double)s0.mu;
    // the M&M read position t = ii+mu sits 3 samples before the interpolation instant
    double ca = clk_count_at(cnt, nb, sps, t0, off, BL) + 3.0 / sps;
    double target = rint(ca) + (double)k * NS;
    // invert the piecewise-linear count curve around the nominal position
    double t = (target - cnt[0]) * sps + off + 0.5 * BL;
    for (int it = 0; it < 4; ++it) {
        double c = clk_count_at(cnt, nb, sps, t, off, BL);
        t += (target - c) * sps;
    }
    t -= 3.0;
    if (t < 0) t = 0;
    ClockState s;
    s.ii = (long long)floor(t);
    s.mu = (float)(t - floor(t));
    s.omega = omega0;
    s.p0 = cf32{0.f, 0.f}; s.p1 = cf32{0.f, 0.f};
    s.c0 = cf32{0.f, 0.f}; s.c1 = cf32{0.f, 0.f};
    // history: the two symbols before the chain, interpolated one and two nominal periods earlier
    for (int back = 2; back >= 1; --back) {
        double tb = t - back * (double)omega0;
        if (tb < 0) continue;
        long long ib = (long long)floor(tb);
        if (ib >= ni) continue;
        float mub = (float)(tb - floor(tb));
        int imu = (int)rintf(mub * (float)XR_MM_NSTEPS);
        const float *row = table + imu * XR_MM_NTAPS;
        float ar = 0.f, ai = 0.f;
        for (int q = 0; q < XR_MM_NTAPS; ++q) {
            float2 v = x[ib + q];
            ar += row[XR_MM_NTAPS - 1 - q] * v.x;
            ai += row[XR_MM_NTAPS - 1 - q] * v.y;
        }
        s.p1 = s.p0; s.c1 = s.c0;
        s.p0 = cf32{ar, ai};
        s.c0 = cf32{ar > 0.f ? 1.f : 0.f, ai > 0.f ? 1.f : 0.f};
    }
    S[k] = s;
}

// ------------------------------------------------------------ hand-off solve
// Policy for newton.h.  State components: (t = ii + mu, omega).  A residual of m
// whole symbol periods is carried as a slip count (aux) that shifts all later chains.
struct ClockPolicy {
    ClockState *S;
    const ClockState *E;
    const float4 *J;
    int *dirty;
    const int *nrun;      // symbols chain k produced when it last ran
    unsigned *cnt;        // [0] changed, [1] not frozen, [2] max |r_t| bits, [3] large, [4] sum r_t^2 (float)
    float trust_t, trust_w, tol_t, tol_w;
    int min_passes;
    const float4 *jmean;  // not null: every chain takes the stream's mean Jacobian (ClockStage::begin)
    float floor_sq;       // mean squared residual (samples^2) at which a hand-off counts as being at the recurrence's floor
    int relay_after;      // > 0: the exact closure follows (cfg.clock_exact >= 1) -- the walkers need start states near the
                          // trajectory, not a hand-off at its floor: closed after this many passes once no residual is large

    struct Elem { ClockState e, s; float4 j; int nrun; };
    __device__ float4 jac_of(long long k) const { return jmean ? jmean[0] : J[k]; }
    __device__ Elem fetch(long long k) const { return Elem{E[k], S[k + 1], jac_of(k), nrun[k]}; }
    __device__ bool active(const Elem &el) const { return el.nrun > 0; }
    __device__ void residual(const Elem &el, float &r1, float &r2, int &aux) const
    {
        float rt = clock_tdiff(el.e, el.s);
        float m = rintf(rt / el.e.omega);
        r1 = rt - m * el.e.omega;
        r2 = el.e.omega - el.s.omega;
        aux = (int)m;
    }
    __device__ float4 jac(const Elem &el) const
    {
        float4 j = el.j;
        if (!(fabsf(j.x) < 4.f) || !(fabsf(j.y) < 16384.f) || !(fabsf(j.z) < 1.f) || !(fabsf(j.w) < 4.f))
            j = make_float4(0.f, 0.f, 0.f, 0.f);
        return j;
    }
    __device__ bool outside_trust(float d1, float d2) const
    {
        return !(fabsf(d1) <= trust_t) || !(fabsf(d2) <= trust_w);
    }
    // same idea as the Costas policy: a timing residual of a good part of a sample is an acquisition or slip
    // transient, not something the finite-difference Jacobian describes
    __device__ bool distrust(float r1, float) const { return !(fabsf(r1) <= 0.5f); }
    __device__ void update(long long k, const Elem &el, float j1, float j2, float n1, float n2, int slip, int slip_k,
                           float r1, NewtonStat &st) const
    {
        const ClockState ek = el.e, old = el.s;
        const bool hist_same = ek.p0.x == old.p0.x && ek.p0.y == old.p0.y && ek.p1.x == old.p1.x &&
                               ek.p1.y == old.p1.y && ek.c0.x == old.c0.x && ek.c0.y == old.c0.y &&
                               ek.c1.x == old.c1.x && ek.c1.y == old.c1.y;
        if (!active(el)) {
            // chain k produced nothing: its successor starts where it stands
            const bool same = old.ii == ek.ii && old.mu == ek.mu && old.omega == ek.omega && hist_same;
            if (!same) { S[k + 1] = ek; dirty[k + 1] = 1; st.changed += 1; }
            return;
        }
        const bool frozen = fabsf(n1) <= tol_t && fabsf(n2) <= tol_w && slip == 0 && slip_k == 0 && hist_same;
        if (frozen) return;
        ClockState nw = ek;
        clock_shift(nw, (float)slip * ek.omega + j1);
        nw.omega = ek.omega + j2;
        if (nw.ii < 0) { nw.ii = 0; nw.mu = 0.f; }
        st.open_ += 1;
        st.max_r = fmaxf(st.max_r, fabsf(r1));
        if (fabsf(r1) > 0.02f || slip_k != 0) st.large += 1;
        st.sum_sq += newton_fix(r1 * r1);
        const bool same = old.ii == nw.ii && old.mu == nw.mu && old.omega == nw.omega && hist_same;
        if (!same) { S[k + 1] = nw; dirty[k + 1] = 1; st.changed += 1; }
    }
    // After every solve: ctl[0] done, ctl[1] passes run, ctl[2] open boundaries, ctl[3] max residual (bits),
    // ctl[4] previous summed squared residual (bits).  The recurrence is chaotic at the 1e-5 level
    // (interpolator-arm quantisation), so boundaries keep moving by that much for ever; what must close are the
    // LARGE residuals (acquisition at the head of a cold-started call, symbol slips: decision flips kick mu by up
    // to ~2e-3, acquisition and slips leave residuals >> 0.02 samples).  After that the passes go on only while
    // the summed squared residual still falls by > 45 % per pass.
    __device__ void decide(int *ctl) const
    {
        const unsigned changed = newton_cnt_load(cnt + 0), open_ = newton_cnt_load(cnt + 1);
        const unsigned mr = newton_cnt_load(cnt + 2), large = newton_cnt_load(cnt + 3);
        const unsigned long long sq = (unsigned long long)newton_cnt_load(cnt + 4) |
                                      ((unsigned long long)newton_cnt_load(cnt + 5) << 32);
        ctl[1] += 1;
        ctl[2] = (int)open_;
        ctl[3] = (int)mr;
        const float q = newton_unfix(sq);
        const float q_prev = ctl[1] == 1 ? INFINITY : __int_as_float(ctl[4]);
        ctl[4] = __float_as_int(q);
        ctl[5] = (large != 0 || __uint_as_float(mr) > 0.02f) ? 1 : 0;   // trust gate only while residuals are large
        ctl[9] = (int)large;       // boundaries with a residual beyond 0.02 sample or an open slip: what a caller can act on
        const int open_prev = ctl[1] == 1 ? 0x7fffffff : ctl[6];
        ctl[6] = (int)open_;
        if (changed == 0) { ctl[0] = 1; ctl[2] = 0; return; }
        if (relay_after > 0 && ctl[1] >= relay_after && large == 0 && !(__uint_as_float(mr) > 0.02f)) { ctl[0] = 1; return; }
        // "stalled" is the chaos floor only if the boundaries have also stopped freezing: at C2 112 653 of 113 266
        // stay open from pass to pass (they move by 1e-5 for ever), whereas a call of a few dozen chains closes
        // EXACTLY given the passes (20 -> 19 -> ... -> 0 open, then 3e-7 from the serial loop) and its summed
        // residual does not fall monotonically on the way -- the stall test alone stopped such calls at 2e-3
        // sample (fuzz at Es/N0 3..8 dB, HRIT: 5e-4..1e-3 rms, a hard decision flipped here and there).
        // ... while the residuals are still above what the floor looks like at 12 dB (rms 1e-4 sample): a small call
        // on a clean signal stops where a big one does (6 passes, 2e-4 from the serial loop) instead of running
        // 15..25 passes of ~100 us each down to 1e-6.
        // A call that is still acquiring timing (a residual beyond 0.02 sample two or more passes in) is also one in
        // which the loop is far from its fixed point and sensitive: 4e-5 sample left at the hand-offs of such a call
        // showed as 5e-3 in its symbols (fuzz: cold HRIT start, 3147 symbols, 14.7 dB).  It runs on while boundaries
        // freeze, whatever the level.
        int memo = ctl[7];
        if (ctl[1] >= 3 && large != 0) memo |= 0x100;
        const bool acquiring = (memo & 0x100) != 0;
        const bool above_floor = open_ != 0u && (acquiring || q > floor_sq * (float)open_);
        const bool freezing = above_floor && open_prev != 0x7fffffff && (long long)open_prev - (long long)open_ >= 1 &&
                              200ll * ((long long)open_prev - (long long)open_) >= (long long)open_prev;
        // With a few thousand boundaries or fewer the summed residual is a noisy statistic (a handful of boundaries
        // carry it): one pass without a 45 % fall is not yet the floor there, two in a row are.
        const bool flat = q > 0.55f * q_prev;
        const int flat_runs = flat ? (memo & 0xff) + 1 : 0;
        ctl[7] = (memo & 0x100) | (flat_runs > 0xff ? 0xff : flat_runs);
        const bool stalled = !freezing && (open_ >= 4096u ? flat : flat_runs >= 2);
        if (ctl[1] >= min_passes && large == 0 && stalled) ctl[0] = 1;
    }
};

// --------------------------------------------------------------------- pass
// Sample access.  A lane advances through its chain at its own, data dependent
// pace, reading an 8-sample window per symbol; done straight from global memory
// every wave instruction touches 64 different cache lines.  Instead the block
// keeps, for every chain, a ring of R = WP samples in LDS (row stride WS = R + 1
// float2 so that the per-lane ds_read_b64 stay spread over the banks).  The ring of
// a chain that starts at read index ii0 begins at origin = ii0 - CLK_M and moves
// on a FIXED schedule: during sub-step j (SS symbols) it holds samples
// origin + cum_j + [0, R), cum_j = floor(j * SS * omega_mid).  Within a chain the
// read index stays within a sample or two of that schedule (omega is clipped to
// +-omega_lim around omega_mid), so the samples a sub-step adds -- cum_{j+1} - cum_j
// per row -- are known without looking at the state: they are requested before
// the current symbols are computed and stored afterwards (sample a lives in slot
// (a - origin) mod R), and every sample is fetched once.  A wave in which some lane
// has left its ring (acquisition, wildly wrong start, end of input) computes that
// sub-step from global memory.
constexpr int CLK_M = 2;      // samples kept below the start index
constexpr int CLK_SLACK = 3;  // head room above the schedule: R >= CLK_M + CLK_SLACK + A + 8
// The first CLK_MIR slots of a ring are kept twice, once more behind slot R - 1: the 8-sample window of a symbol
// then never wraps, its reads are one address and seven immediate offsets instead of eight masked indices
// (24 of the ~95 vector instructions of a symbol went into those indices).  Row stride WS = R + CLK_MIR float2,
// odd, so the per-lane ds_read_b64 stay spread over the banks.
#ifndef XR_NOSTORE
#define XR_NOSTORE 0
#endif
#ifndef XR_LOAD_AUX
#define XR_LOAD_AUX 0
#endif
#ifndef XR_CLK_MIR
#define XR_CLK_MIR (XR_MM_NTAPS - 1)
#endif
constexpr int CLK_MIR = XR_CLK_MIR;      // 0: no mirror, masked window indices (row stride R + 1)
constexpr int CLK_ROW_EXTRA = CLK_MIR > 0 ? CLK_MIR : 1;

struct ClockTile {
    float *table;          // 129 x 8, one copy per workgroup
    int *wb;               // ring origin per chain (-1: row unused); the buffer index fits 32 bits
    float2 *tile;          // 64 x WS
    float2 *dump;          // this thread's dump slot: stores that have nothing to store go there (a select instead
                           // of a branch around the store)
};

// LDS of a workgroup: the table, then NG groups (one per 64 chains) of ring origins and rings, then the dump slots.
// NG > 1 only with one wave per group (NV == 1): the groups then share nothing but the table.
__device__ __forceinline__ ClockTile clock_tile_carve(char *smem, int grp, int ngroups, int WS)
{
    ClockTile t;
    t.table = reinterpret_cast<float *>(smem);
    t.wb = reinterpret_cast<int *>(smem + 4160) + 64 * grp;
    float2 *tiles = reinterpret_cast<float2 *>(smem + 4160 + 256 * ngroups);
    t.tile = tiles + (size_t)grp * 64 * WS;
    t.dump = tiles + (size_t)ngroups * 64 * WS + threadIdx.x;
    return t;
}

static inline size_t clock_tile_bytes(int WS, int ngroups, int threads)
{
    return 4160 + 256 * (size_t)ngroups + ((size_t)ngroups * 64 * WS + threads) * sizeof(float2);
}

// The call's samples as a raw buffer whose base is the lowest ring origin of the workgroup: element offsets are
// 32-bit byte offsets from there (consecutive chains: a few hundred KiB), the position of a fill along the
// schedule goes into the scalar offset, and a read beyond the end of the input returns zero instead of needing a
// clamp -- one buffer_load per element, no address arithmetic in vector registers.
struct ClockSrc {
    __amdgpu_buffer_rsrc_t rsrc;
    int omin;
};
__device__ __forceinline__ ClockSrc clock_src(const float2 *x, long long N, int omin)
{
    ClockSrc c;
    omin = __builtin_amdgcn_readfirstlane(omin);     // wave-uniform by construction: keep the descriptor in SGPRs
    c.omin = omin;
    const long long left = (N - (long long)omin) * 8;
    const unsigned bytes = left <= 0 ? 0u : (left > 0xffffffffLL ? 0xffffffffu : (unsigned)left);
    c.rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(x + omin), 0, bytes, 0x00020000);
    return c;
}

// All threads of the block (NV waves).  A fill moves nc consecutive samples of every row, starting at
// origin + first, into ring slots (first + col) mod R (and their mirror); only columns < ncol are stored (a
// sub-step adds nc or nc - 1 samples, the surplus column is the first sample of the next fill: same cache line, no
// traffic).  The 64 x nc elements are dealt to the lanes in row-major order, IT wave instructions per wave; every
// lane keeps its elements' source offset and LDS position in registers (prepare), so a fill is a straight run of
// unconditional loads that are all in flight together -- issued before the symbols they overlap are computed,
// stored afterwards.  The loads must not sit under a lane predicate: the registers would become phis and the
// compiler would wait for the data on the spot.
template <int NV, int IT> struct ClockFill {
    unsigned gb[IT];    // byte offset from the source base of the element at first = 0 (unused rows, surplus: 0)
    int lc[IT];         // (row * WS) << 8 | col, -1: nothing to store
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    v2u v[IT];
    __device__ __forceinline__ void prepare(const int *origin, int nc, int WS, int omin)
    {
        const int lane = threadIdx.x & 63, wave = NV > 1 ? threadIdx.x >> 6 : 0;
        const int magic = 65536 / nc + 1;                 // floor(e / nc) = (e * magic) >> 16 for e < 64 * 64
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int e = (it * NV + wave) * 64 + lane;
            const int row = (e * magic) >> 16, col = e - row * nc;
            const bool ok = row < 64 && origin[min(row, 63)] >= 0;
            gb[it] = ok ? (unsigned)(origin[min(row, 63)] - omin + col) * 8u : 0u;
            lc[it] = ok ? ((row * WS) << 8) | col : -1;
        }
    }
    __device__ __forceinline__ void issue(int first, const ClockSrc &src)
    {
#pragma unroll
        for (int it = 0; it < IT; ++it) v[it] = __builtin_amdgcn_raw_buffer_load_b64(src.rsrc, gb[it], first * 8, XR_LOAD_AUX);
    }
    template <int R> __device__ __forceinline__ void commit(float2 *tile, float2 *dump, int first, int ncol) const
    {
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const bool ok = (unsigned)(lc[it] & 255) < (unsigned)ncol && lc[it] >= 0;
            const int slot = (first + lc[it]) & (R - 1);
            float2 *p = tile + (lc[it] >> 8) + slot;
            const float2 val = make_float2(__uint_as_float(v[it].x), __uint_as_float(v[it].y));
            *(ok ? p : dump) = val;
            if (CLK_MIR > 0) *((ok && slot < CLK_MIR) ? p + R : dump) = val;
        }
    }
};

// ring position of the schedule after jj sub-steps (16.16 fixed point step)
__device__ __forceinline__ int clock_cum(int jj, int STEP) { return (int)(((long long)jj * STEP) >> 16); }

// One symbol from the ring: the window x[ii .. ii+7] starts at slot off mod R (off = ii - origin) and runs on
// without wrapping, into the mirror slots if need be.
template <int R>
__device__ __forceinline__ cf32 clock_step_ring(const cf32 *row, int &off, const float *table, ClockState &s,
                                                const ClockPar &par)
{
    cf32 wl[XR_MM_NTAPS];
    const cf32 *w = row + (off & (R - 1));
    if (CLK_MIR == 0) {
#pragma unroll
        for (int k = 0; k < XR_MM_NTAPS; ++k) wl[k] = row[(off + k) & (R - 1)];
        w = wl;
    }
    ClockState t = s;
    t.ii = 0;
    cf32 p = clock_step_w(w, table, t, par);
    off += (int)t.ii;
    t.ii = s.ii;
    s = t;
    return p;
}

// interpolator table -> LDS.  All loads are issued before the first store: as a plain copy loop the compiler
// emits load / wait / store per element, 17 serial memory latencies at the head of every 64-thread block.
__device__ __forceinline__ void clock_table_to_lds(float *dst, const float *__restrict__ src)
{
    constexpr int NEL = (XR_MM_NSTEPS + 1) * XR_MM_NTAPS;
    constexpr int NIT = (NEL + 63) / 64;
    const int nthr = blockDim.x;
    float tv[NIT];
#pragma unroll
    for (int q = 0; q < NIT; ++q) tv[q] = src[min((int)threadIdx.x + q * nthr, NEL - 1)];
#pragma unroll
    for (int q = 0; q < NIT; ++q) {
        const int i = (int)threadIdx.x + q * nthr;
        if (i < NEL) dst[i] = tv[q];
    }
}

}  // namespace xrit
#include "clock_relay.h"
namespace xrit {
struct ClockResult;
__device__ __forceinline__ double clk_count_at(const double *cnt, int nb, double sps, double t, double off, int BL);
}
#include "clock_overlap.h"
#ifdef XRIT_EXPERIMENTS
#include "../../experiments/csrc/clock_relay_wide.h"      // walker teams: built, verified, no faster; not in the shipped library
#else
namespace xrit { constexpr int RW_REC_PAD = 0; }
#endif
namespace xrit {

// SS symbols of one lane.  Fast path: every running lane of the wave stays inside its ring for the whole
// sub-step and cannot reach the end of the input -> no per-symbol guards, LDS reads only.  Otherwise the wave
// computes the sub-step from global memory with the guards.  orow (output pass): where the symbols go.
// the output tile holds complex symbols, or only their real parts when nobody asks for the complex ones
__device__ __forceinline__ void clock_put(cf32 &d, const cf32 &p) { d = p; }
__device__ __forceinline__ void clock_put(float &d, const cf32 &p) { d = p.x; }

template <int WP, bool OUT, typename OutT>
__device__ __forceinline__ void clock_substep(const ClockTile &t, const float2 *__restrict__ x, int WS, int lane,
                                              int origin, int cum, int lim, int SS, int A, long long ni,
                                              const ClockPar &par, ClockState &s, int &off, bool &alive,
                                              int &produced, OutT *orow)
{
    const int rel = off - cum;
    const bool safe = !alive || (lim == SS && rel >= 0 && rel + A + XR_MM_NTAPS <= WP &&
                                 (long long)origin + off + A < ni);
    if (__all(safe)) {
        if (alive) {
            const cf32 *rowp = reinterpret_cast<const cf32 *>(t.tile + lane * WS);
            if (SS == 4) {
                // the usual sub-step, unrolled: the symbol-to-symbol hand-over of (p0, p1, c0, c1) becomes
                // register renaming instead of a dozen moves per symbol
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    cf32 p = clock_step_ring<WP>(rowp, off, t.table, s, par);
                    if (OUT) clock_put(orow[i], p);
                }
            } else {
                for (int i = 0; i < SS; ++i) {
                    cf32 p = clock_step_ring<WP>(rowp, off, t.table, s, par);
                    if (OUT) {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (q == i) clock_put(orow[q], p);
                    }
                }
            }
            s.ii = (long long)origin + off;
            produced += SS;
        }
    } else {
        for (int i = 0; i < lim; ++i) {
            if (alive && (s.ii >= ni || s.ii < 0)) alive = false;
            if (alive) {
                cf32 p = clock_step_w(reinterpret_cast<const cf32 *>(x) + s.ii, t.table, s, par);
                if (OUT) {
                    // (constant indices: orow[i] with a run-time i put the caller's four symbols in scratch memory --
                    // 48 bytes per lane that the fast path then went through as well)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (q == i) clock_put(orow[q], p);
                }
                ++produced;
            }
        }
        off = (int)(s.ii - origin);
    }
}

// The sub-step loop with the ring fills two sub-steps ahead of the compute: the samples sub-step j+2 adds are
// requested (into registers) before sub-step j is computed and stored after sub-step j+1.  compute(j, cum_j)
// does the work of sub-step j.  Called by all threads of the block, t.wb (ring origins) already visible.
// a tile is shared by the NV waves of its group: with one wave per group LDS program order is all it takes
template <int NV> __device__ __forceinline__ void clock_tile_barrier()
{
    if (NV > 1) lds_barrier();
    else __builtin_amdgcn_wave_barrier();
}

template <int NV, int WP, int IT, typename Compute>
__device__ __forceinline__ void clock_pipeline(const ClockTile &t, const ClockSrc &src, int WS, int nsub, int STEP,
                                               Compute &&compute)
{
    {
        // the first window: all R columns of every row
        ClockFill<NV, (WP + NV - 1) / NV> w;
        w.prepare(t.wb, WP, WS, src.omin);
        w.issue(0, src);
        w.template commit<WP>(t.tile, t.dump, 0, WP);
    }
    const int nc = (STEP >> 16) + 1;
    if constexpr (IT * NV > 32) {
        // wide rings (sps > ~18): one fill in flight -- two would not fit the register file
        ClockFill<NV, IT> f;
        f.prepare(t.wb, nc, WS, src.omin);
        clock_tile_barrier<NV>();
        int c0 = 0;
        for (int j = 0; j < nsub; ++j) {
            const int c1 = clock_cum(j + 1, STEP);
            if (j + 1 < nsub) f.issue(c0 + WP, src);
            compute(j, c0);
            clock_tile_barrier<NV>();
            if (j + 1 < nsub) f.template commit<WP>(t.tile, t.dump, c0 + WP, c1 - c0);
            clock_tile_barrier<NV>();
            c0 = c1;
        }
        return;
    }
    ClockFill<NV, IT> f0, f1;
    f0.prepare(t.wb, nc, WS, src.omin);
#pragma unroll
    for (int it = 0; it < IT; ++it) { f1.gb[it] = f0.gb[it]; f1.lc[it] = f0.lc[it]; }
    clock_tile_barrier<NV>();
    int c0 = 0, c1 = clock_cum(1, STEP);
    if (nsub > 1) f1.issue(c0 + WP, src);
    for (int j = 0; j < nsub; j += 2) {
        const int c2 = clock_cum(j + 2, STEP), c3 = clock_cum(j + 3, STEP);
        if (j + 2 < nsub) f0.issue(c1 + WP, src);
        compute(j, c0);
        clock_tile_barrier<NV>();
        if (j + 1 < nsub) f1.template commit<WP>(t.tile, t.dump, c0 + WP, c1 - c0);
        clock_tile_barrier<NV>();
        if (j + 1 >= nsub) break;
        if (j + 3 < nsub) f1.issue(c2 + WP, src);
        compute(j + 1, c1);
        clock_tile_barrier<NV>();
        if (j + 2 < nsub) f0.template commit<WP>(t.tile, t.dump, c1 + WP, c2 - c1);
        clock_tile_barrier<NV>();
        c0 = c2;
        c1 = c3;
    }
}

// lowest ring origin of the workgroup's rows (rows in use only); t.wb must be visible
__device__ __forceinline__ int clock_origin_min(const int *wb)
{
    int o = wb[threadIdx.x & 63];
    o = o < 0 ? 0x7fffffff : o;
    for (int off = 32; off > 0; off >>= 1) o = min(o, __shfl_xor(o, off, 64));
    return o == 0x7fffffff ? 0 : o;
}

// NV == 3 (192 threads, one group): wave 0 = base trajectories of 64 chains, wave 1 = start shifted by h_t,
// wave 2 = omega shifted by h_w; the base lane forms the finite-difference Jacobian.  NV == 1: base trajectories
// only -- the Jacobian of an earlier pass is kept, or the stream's mean Jacobian is used (quasi-Newton, see
// ClockStage::begin) -- and the workgroup holds blockDim.x / 64 groups of 64 chains that share the table.
// what a pass that also leaves the symbols needs (OUTP): from the pass the stop test usually fires after, the passes
// write every symbol they compute -- the pass after which the hand-off closes then IS the output pass, and
// clock_output_kernel (a whole extra sweep of the stream) returns at once.  The first such pass runs every chain.
struct ClockPassOut {
    float *soft;
    float2 *sym;
    unsigned long long cap;
    int *valid;          // ctl word: the symbols in soft / sym belong to the end states in E
    int *written;        // per chain: its symbols have been written in this call (a chain that has not runs, dirty or not)
    float4 *stage;       // soft symbols in the order the waves produce them: [wave of 64 chains][sub-step][lane] -> a wave
                         // instruction writes 1 KiB of whole lines (clock_unstage_kernel puts them where they belong)
};
constexpr int CLK_CTL_SYMBOLS = 13;

template <int NV, int WP, int NCM, bool OUTP = false>
__global__ void __launch_bounds__(NV > 1 ? 64 * NV : 512) clock_pass_kernel(const float2 *__restrict__ x, const float *__restrict__ table_g,
                                                             const ClockState *__restrict__ S, ClockState *__restrict__ E,
                                                             float4 *__restrict__ J, int *__restrict__ dirty,
                                                             int *__restrict__ nrun, long long N, long long ni, int K,
                                                             int NS, ClockPar par, int SS, int W, int WS, int A,
                                                             int STEP, const int *__restrict__ ctl, ClockPolicy pol,
                                                             AffMap *__restrict__ aggs, ClockPassOut po = ClockPassOut{})
{
    // the hand-off already closed (later passes of the batch are no-ops), or the gated solve has taken over
    if (ctl[0] || (aggs != nullptr && ctl[NEWTON_CTL_TAKEOVER])) return;
    if (OUTP && blockIdx.x == 0 && threadIdx.x == 0) *po.valid = 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ float2 endv[2][NV > 1 ? 64 : 1];
    __shared__ long long ref_ii[NV > 1 ? 64 : 1];
    __shared__ int any_run;
    const int ngroups = NV > 1 ? 1 : (int)(blockDim.x >> 6);
    const int variant = NV > 1 ? threadIdx.x >> 6 : 0, grp = NV > 1 ? 0 : threadIdx.x >> 6, lane = threadIdx.x & 63;
    const ClockTile t = clock_tile_carve(smem, grp, ngroups, WS);
    const int wv = blockIdx.x * ngroups + grp;      // which 64 chains
    const int k = wv * 64 + lane;
    bool run = k < K;
    if (run) run = dirty[k] != 0 || (OUTP && po.written[k] == 0);
    if (threadIdx.x == 0) any_run = 0;
    __syncthreads();
    if (run && variant == 0) any_run = 1;
    clock_table_to_lds(t.table, table_g);
    __syncthreads();
    if (!any_run && aggs == nullptr) return;
    ClockState s{};
    int produced = 0;
    float4 jk = make_float4(0.f, 0.f, 0.f, 0.f);
    // (a group none of whose chains runs has nothing to walk; with NV > 1 the three waves of the group decide together)
    const bool walk = NV > 1 ? any_run != 0 : __any(run);
    if (walk) {
        bool alive = run;
        if (run) {
            s = S[k];
            if (NV > 1 && variant == 1) clock_shift(s, CLK_H_T);
            if (NV > 1 && variant == 2) s.omega += CLK_H_W;
        }
        if (variant == 0) t.wb[lane] = alive ? max((int)s.ii - CLK_M, 0) : -1;
        if (NV > 1) __syncthreads();
        else __builtin_amdgcn_wave_barrier();
        const int origin = t.wb[lane];
        int off = (int)(s.ii - origin);
        const int nsub = (NS + SS - 1) / SS;
        const ClockSrc src = clock_src(x, N, clock_origin_min(t.wb));
        const unsigned long long obase = (unsigned long long)k * NS;
        clock_pipeline<NV, WP, (NCM + NV - 1) / NV>(t, src, WS, nsub, STEP, [&](int j, int cum) {
            if (!OUTP) {
                clock_substep<WP, false>(t, x, WS, lane, origin, cum, min(SS, NS - j * SS), SS, A, ni, par, s, off, alive,
                                         produced, (float *)nullptr);
                return;
            }
            // (as in clock_output_kernel: a lane's symbols of one sub-step leave as one 16-byte store)
            cf32 ps[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) ps[i] = cf32{0.f, 0.f};
            const int before = produced;
            clock_substep<WP, true>(t, x, WS, lane, origin, cum, min(SS, NS - j * SS), SS, A, ni, par, s, off, alive,
                                    produced, ps);
            const int nv = produced - before;
            const unsigned long long o = obase + (unsigned long long)j * SS;
            // (SS == 4 here, see ClockStage::enqueue_passes.)  Written straight to k * NS + i, a lane's 16 bytes of a
            // sub-step were a partial line of their own: 64 lines per wave instruction, each touched again 4 us
            // later -- 226 MB written and 167 MB more fetched for 50 MB of symbols, the pass 70 us longer.
            if (po.soft && run) po.stage[((size_t)wv * nsub + j) * 64 + lane] = make_float4(ps[0].x, ps[1].x, ps[2].x, ps[3].x);      // (a chain that does not run keeps what an earlier pass staged)
            if (nv == 4 && (NS & 3) == 0 && o + 3 < po.cap) {
                if (po.sym) {
                    *reinterpret_cast<float4 *>(po.sym + o) = make_float4(ps[0].x, ps[0].y, ps[1].x, ps[1].y);
                    *reinterpret_cast<float4 *>(po.sym + o + 2) = make_float4(ps[2].x, ps[2].y, ps[3].x, ps[3].y);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (i < nv && o + i < po.cap) {
                        if (po.sym) po.sym[o + i] = make_float2(ps[i].x, ps[i].y);
                    }
                }
            }
        });
        if (NV > 1) {
            // hand the perturbed end states to the base lane as (t - t_ref, omega) with a common reference
            if (variant == 0) ref_ii[lane] = s.ii;
            __syncthreads();
            if (variant > 0) endv[variant - 1][lane] = make_float2((float)(s.ii - ref_ii[lane]) + s.mu, s.omega);
            __syncthreads();
        }
        if (variant == 0 && run) {
            if (NV > 1) {
                float2 et = endv[0][lane], ew = endv[1][lane];
                float tb = s.mu;
                jk.x = (et.x - tb) / CLK_H_T;        // dt/dt0
                jk.y = (ew.x - tb) / CLK_H_W;        // dt/dw0
                jk.z = (et.y - s.omega) / CLK_H_T;   // dw/dt0
                jk.w = (ew.y - s.omega) / CLK_H_W;   // dw/dw0
                J[k] = jk;
            }
            E[k] = s;
            nrun[k] = produced;
            dirty[k] = 0;
            if (OUTP) po.written[k] = 1;
        }
    }
    if (aggs == nullptr || variant != 0 || wv * 64 >= K) return;
    // wave-aligned hand-off solve (newton.h): this wave's 64 boundary maps, composed.  A chain that ran has its end
    // state in registers.
    AffMap e = aff_identity();
    if (k < K - 1) {
        ClockPolicy::Elem el;
        el.s = S[k + 1];
        if (run) {
            el.e = s;
            el.nrun = produced;
            el.j = NV > 1 ? jk : pol.jac_of(k);
        } else {
            el.e = E[k];
            el.nrun = nrun[k];
            el.j = pol.jac_of(k);
        }
        e = newton_element(pol, el, false);
    }
    newton_wave_aggregate(e, wv, aggs);
}

// output pass: base trajectories only; symbol i of chain k goes to k*NS + i.  A lane's symbols of one sub-step
// leave as one 16-byte store (soft) -- consecutive sub-steps fill the rest of the 128-byte line, which the L2 holds
// until then (the lines in flight, 64 chains x 128 B per wave, fit it many times over).
template <int WP, int NCM, bool SYM>
__global__ void __launch_bounds__(512) clock_output_kernel(const float2 *__restrict__ x, const float *__restrict__ table_g,
                                                          const ClockState *__restrict__ S, ClockState *__restrict__ E,
                                                          int *__restrict__ counts, float *__restrict__ soft,
                                                          float2 *__restrict__ sym, unsigned long long cap, long long N,
                                                          long long ni, int K, int NS, ClockPar par,
                                                          int *__restrict__ terminal, int SS, int W, int WS, int A,
                                                          int STEP, const int *__restrict__ ctl)
{
    if (ctl[CLK_CTL_SYMBOLS]) return;        // the last pass wrote the symbols (ClockPassOut): E, nrun hold the rest
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ngroups = (int)(blockDim.x >> 6), grp = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const ClockTile t = clock_tile_carve(smem, grp, ngroups, WS);
    clock_table_to_lds(t.table, table_g);
    __syncthreads();
    const int k = (blockIdx.x * ngroups + grp) * 64 + lane;
    const bool mine = k < K;
    if (!__any(mine)) return;
    ClockState s{};
    if (mine) s = S[k];
    int produced = 0;
    bool alive = mine;
    t.wb[lane] = alive ? max((int)s.ii - CLK_M, 0) : -1;
    __builtin_amdgcn_wave_barrier();
    const int origin = t.wb[lane];
    int off = (int)(s.ii - origin);
    const int nsub = (NS + SS - 1) / SS;
    const ClockSrc src = clock_src(x, N, clock_origin_min(t.wb));
    const unsigned long long obase = (unsigned long long)k * NS;
    clock_pipeline<1, WP, NCM>(t, src, WS, nsub, STEP, [&](int j, int cum) {
        cf32 ps[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) ps[i] = cf32{0.f, 0.f};
        const int before = produced;
        // (sub-steps have at most 4 symbols, see ClockStage::begin)
        clock_substep<WP, true>(t, x, WS, lane, origin, cum, min(SS, NS - j * SS), SS, A, ni, par, s, off, alive,
                                produced, ps);
        const int nv = produced - before;          // symbols of this sub-step that exist
        const unsigned long long o = obase + (unsigned long long)j * SS;
        if (nv == 4 && (NS & 3) == 0 && SS == 4 && o + 3 < cap) {
            if (soft && !XR_NOSTORE) *reinterpret_cast<float4 *>(soft + o) = make_float4(ps[0].x, ps[1].x, ps[2].x, ps[3].x);
            if (SYM && sym) {
                *reinterpret_cast<float4 *>(sym + o) = make_float4(ps[0].x, ps[0].y, ps[1].x, ps[1].y);
                *reinterpret_cast<float4 *>(sym + o + 2) = make_float4(ps[2].x, ps[2].y, ps[3].x, ps[3].y);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (i < nv && o + i < cap) {
                    if (soft) soft[o + i] = ps[i].x;
                    if (SYM && sym) sym[o + i] = make_float2(ps[i].x, ps[i].y);
                }
            }
        }
    });
    if (!mine) return;
    E[k] = s;
    counts[k] = produced;
    if (produced < NS) atomicMin(terminal, k);   // ran out of input: the first such chain ends the call
}

// (chains longer than this -- cfg.clock_chain_syms -- do not stage their symbols: the tile of 64 x (NS + 1) floats must fit LDS)
constexpr int CLK_STAGE_MAX_NS = 512;
// a hand-off pass left the soft symbols in wave order (ClockPassOut::stage): a workgroup takes the 64 chains of one
// wave -- 64 * NS symbols that are ONE contiguous stretch of the output -- through LDS; both sides whole lines
__global__ void __launch_bounds__(256) clock_unstage_kernel(const float4 *__restrict__ stage, float *__restrict__ soft,
                                                            const int *__restrict__ nrun, const int *__restrict__ ctl,
                                                            unsigned long long cap, int K, int NS)
{
    if (!ctl[CLK_CTL_SYMBOLS]) return;        // the output pass wrote the symbols itself
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *tile = reinterpret_cast<float *>(smem);          // [64][NS + 1]
    int *cnt = reinterpret_cast<int *>(tile + 64 * (NS + 1));
    const int wv = blockIdx.x, nsub = NS / 4;
    if (threadIdx.x < 64) cnt[threadIdx.x] = wv * 64 + (int)threadIdx.x < K ? nrun[wv * 64 + threadIdx.x] : 0;
    for (int e = threadIdx.x; e < nsub * 64; e += 256) {
        const float4 v = stage[(size_t)wv * nsub * 64 + e];
        const int j = e >> 6, c = e & 63;
        float *row = tile + c * (NS + 1) + 4 * j;
        row[0] = v.x; row[1] = v.y; row[2] = v.z; row[3] = v.w;
    }
    __syncthreads();
    const unsigned long long base = (unsigned long long)wv * 64 * NS;
    for (int e = threadIdx.x; e < 64 * NS; e += 256) {
        const int c = e / NS, i = e - c * NS;
        if (i < cnt[c] && base + e < cap) soft[base + e] = tile[c * (NS + 1) + i];
    }
}

// a hand-off pass left the symbols (ClockPassOut): the first chain that ran out of input, from that pass's counts
__global__ void __launch_bounds__(256) clock_terminal_kernel(const int *__restrict__ nrun, const int *__restrict__ ctl,
                                                             int *__restrict__ terminal, int K, int NS)
{
    if (!ctl[CLK_CTL_SYMBOLS]) return;        // the output pass has found it
    const int k = blockIdx.x * 256 + threadIdx.x;
    const bool short_ = k < K && nrun[k] < NS;
    const unsigned long long m = __ballot(short_);
    if (m && (threadIdx.x & 63) == 0) atomicMin(terminal, k + __builtin_ctzll(m));
}

// result of the call + the state and the unread tail carried to the next call
__global__ void __launch_bounds__(1024) clock_finalize_kernel(const ClockState *__restrict__ E,
                                                              const int *__restrict__ counts_out,
                                                              const int *__restrict__ nrun,
                                                              const int *__restrict__ terminal,
                                                              const int *__restrict__ ctl,
                                                              const ClockState *__restrict__ carried_in,
                                                              ClockState *__restrict__ carried_out,
                                                              ClockResult *__restrict__ res,
                                                              const float2 *__restrict__ x,
                                                              float2 *__restrict__ tail_out, long long N, int K, int NS)
{
    __shared__ long long s_ii;
    // symbols per chain: what the output pass counted, or -- when a hand-off pass left the symbols -- what that pass did
    const int *counts = ctl[CLK_CTL_SYMBOLS] ? nrun : counts_out;
    if (threadIdx.x == 0) {
        int k = *terminal;
        ClockState s;
        if (k < 0 || k >= K) {
            // no chain reached the end of the input: the chain budget was too small
            res->ok = 0;
            res->n_symbols = 0;
            res->terminal_chain = -1;
            s = carried_in[0];
        } else {
            res->ok = 1;
            res->terminal_chain = k;
            res->n_symbols = (unsigned long long)k * NS + (unsigned long long)counts[k];
            s = E[k];
        }
        long long ii = s.ii;
        if (ii > N) ii = N;
        if (ii < 0) ii = 0;
        res->ii_final = ii;
        s_ii = ii;
        s.ii = 0;           // the carried tail starts at the read index
        carried_out[0] = s;
    }
    __syncthreads();
    // the unread tail goes to its own buffer: the call's input stays intact until the call is committed
    const long long ii = s_ii;
    const long long carry = N - ii;
    if (threadIdx.x < carry) tail_out[threadIdx.x] = x[ii + threadIdx.x];
}

__global__ void __launch_bounds__(1024) clock_tail_kernel(const float2 *__restrict__ tail, float2 *__restrict__ x, int carry)
{
    if ((int)threadIdx.x < carry) x[threadIdx.x] = tail[threadIdx.x];
}

// ------------------------------------------------------------------- serial
// The recurrence as the CPU runs it: ONE trajectory from the carried state to the end of the call, no chains, no
// hand-offs (cfg.clock_serial).  A single wave: every lane computes the same symbol from an LDS window that the
// wave refills together (coalesced); lane 0 stores.  ~0.3 us per symbol -- a diagnostic that separates what the
// time-tiled evaluation adds from what any float32 M&M fed by this chain's own Costas output differs from the CPU
// chain by (the recurrence lives on a lattice of 2^-21 sample in mu and omega and does not forget a one-ulp
// difference), not a production mode.
constexpr int CLK_SER_W = 2048;      // samples per LDS window

__global__ void __launch_bounds__(64) clock_serial_kernel(const float2 *__restrict__ x, const float *__restrict__ table_g,
                                                          const ClockState *__restrict__ carried_in,
                                                          ClockState *__restrict__ carried_out,
                                                          ClockResult *__restrict__ res, float *__restrict__ soft,
                                                          float2 *__restrict__ sym, unsigned long long cap, long long N,
                                                          long long ni, ClockPar par, float2 *__restrict__ tail_out)
{
    __shared__ float table[(XR_MM_NSTEPS + 1) * XR_MM_NTAPS];
    __shared__ cf32 win[CLK_SER_W];
    clock_table_to_lds(table, table_g);
    const int lane = threadIdx.x;
    ClockState s = carried_in[0];
    long long base = -(long long)CLK_SER_W;      // window = samples [base, base + CLK_SER_W)
    unsigned long long oo = 0;
    __syncthreads();
    while (s.ii < ni && s.ii >= 0 && oo < cap) {
        if (s.ii < base || s.ii + XR_MM_NTAPS > base + CLK_SER_W) {
            __syncthreads();
            base = s.ii;
            for (int i = lane; i < CLK_SER_W; i += 64) {
                const long long j = base + i;
                const float2 v = x[j < N ? j : N - 1];
                win[i] = cf32{v.x, v.y};
            }
            __syncthreads();
        }
        ClockState t = s;
        t.ii = 0;
        const cf32 p = clock_step_w(win + (int)(s.ii - base), table, t, par);
        s.ii += t.ii;
        t.ii = s.ii;
        s = t;
        if (lane == 0) {
            if (soft) soft[oo] = p.x;
            if (sym) sym[oo] = make_float2(p.x, p.y);
        }
        ++oo;
    }
    long long ii = s.ii;
    if (ii > N) ii = N;
    if (ii < 0) ii = 0;
    if (lane == 0) {
        res->ok = (s.ii >= ni || s.ii < 0) ? 1 : 0;     // 0: the output buffer filled before the input ran out
        res->terminal_chain = 0;
        res->n_symbols = oo;
        res->ii_final = ii;
        ClockState c = s;
        c.ii = 0;
        carried_out[0] = c;
    }
    const long long carry = N - ii;
    for (long long i = lane; i < carry && i < 1024; i += 64) tail_out[i] = x[ii + i];
}


__global__ void clk_fill_int_kernel(int *p, int v, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// start of a call: the pass counters cleared and "no chain has run out of input yet", in one launch
__global__ void __launch_bounds__(256) clock_reset_kernel(unsigned *counters, int words, int *terminal)
{
    for (int i = threadIdx.x; i < words; i += 256) counters[i] = 0u;
    if (threadIdx.x == 0 && terminal) *terminal = 0x7fffffff;
}

// Mean chain Jacobian of a locked call (chains of the middle three quarters, entries the policy would accept), in a
// fixed summation order.  The chain Jacobians of a locked stream scatter by 2 % around their mean, and the hand-off
// converges with the mean just as it does with each chain's own (tests/experiments/clock_emulator.py), so the
// finite-difference pass -- three trajectories per chain -- is run once per stream, not once per call.
__global__ void __launch_bounds__(256) clock_jmean_kernel(const float4 *__restrict__ J, int K, float4 *__restrict__ out)
{
    __shared__ double acc[4][256];
    __shared__ int cnt[256];
    const int k0 = K / 8, k1 = K - K / 8;
    double a = 0, b = 0, c = 0, d = 0;
    int n = 0;
    for (int k = k0 + (int)threadIdx.x; k < k1; k += 256) {
        const float4 j = J[k];
        if (fabsf(j.x) < 4.f && fabsf(j.y) < 16384.f && fabsf(j.z) < 1.f && fabsf(j.w) < 4.f) {
            a += j.x; b += j.y; c += j.z; d += j.w;
            ++n;
        }
    }
    acc[0][threadIdx.x] = a; acc[1][threadIdx.x] = b; acc[2][threadIdx.x] = c; acc[3][threadIdx.x] = d;
    cnt[threadIdx.x] = n;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            for (int q = 0; q < 4; ++q) acc[q][threadIdx.x] += acc[q][threadIdx.x + off];
            cnt[threadIdx.x] += cnt[threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double inv = cnt[0] > 0 ? 1.0 / cnt[0] : 0.0;
        out[0] = make_float4((float)(acc[0][0] * inv), (float)(acc[1][0] * inv), (float)(acc[2][0] * inv), (float)(acc[3][0] * inv));
        reinterpret_cast<int *>(out + 1)[0] = cnt[0];
    }
}

static int relay_span(const ClockPar &par, int symbols);

int ClockStage::init(float omega, float gain_omega, float mu, float gain_mu, float omega_rel_limit, int chain_syms,
                     int max_passes_)
{
    sps = omega;
    par.omega_mid = omega;
    par.omega_lim = omega * omega_rel_limit;
    par.gain_omega = gain_omega;
    par.gain_mu = gain_mu;
    mu0 = mu;
    // (fewer than 32 symbols per chain are not taken: 16-symbol chains were seen to mis-resolve a symbol slip on
    // noisy input -- one symbol more or less in the output -- and gain nothing where they were meant to)
    NS = chain_syms > 0 ? (chain_syms < 32 ? 32 : chain_syms) : 64;
    auto_ns = chain_syms <= 0;
    max_passes = max_passes_ > 0 ? max_passes_ : 192;      // see CostasStage::init
    min_passes = max_passes < 4 ? max_passes : 4;
    std::vector<float> tb((XR_MM_NSTEPS + 1) * XR_MM_NTAPS);
    design_mmse_table(tb.data());
    XR_TRY(table.reserve(tb.size() * sizeof(float)));
    XR_HIP(hipMemcpy(table.p, tb.data(), tb.size() * sizeof(float), hipMemcpyHostToDevice));
    XR_TRY(st.reserve(2 * sizeof(ClockState)));
    ClockState s0{};
    s0.ii = 0; s0.mu = mu; s0.omega = omega;
    ClockState both[2] = {s0, s0};
    XR_HIP(hipMemcpy(st.p, both, sizeof both, hipMemcpyHostToDevice));
    XR_TRY(tail.reserve(2 * 1024 * sizeof(float2)));
    XR_TRY(jmean.reserve(64));
    jmean_valid = false;
    XR_TRY(counters.reserve((size_t)(max_passes + 12) * 8 * sizeof(unsigned)));
    XR_HIP(hipHostMalloc((void **)&h_res, 128));
    {
        int dev = 0, v = 0;
        XR_HIP(hipGetDevice(&dev));
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cu_count = v;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, dev) == hipSuccess && v >= 65536)
            lds_per_cu = v;
    }
    cur = 0;
    carry = 0;
    ov_enabled = getenv("XRIT_NO_OVERLAP") == nullptr;
    if (const char *e = getenv("XRIT_OV_HIST")) { const int v = atoi(e); if (v >= 1024) ov_hist = v; }
#ifdef XRIT_EXPERIMENTS
    // (the plan's other parameters: A/B runs of DESIGN.md sections 5-6, not in the shipped library)
    if (const char *e = getenv("XRIT_OV_MIN")) { const long long v = atoll(e); if (v > 0) ov_min = v; }
    if (const char *e = getenv("XRIT_OV_SMALL_RING")) ov_small_ring = atoi(e) != 0;
    if (const char *e = getenv("XRIT_OV_MINL")) { const int v = atoi(e); if (v >= 1024) ov_min_range = v; }
    if (const char *e = getenv("XRIT_OV_MINW")) { const int v = atoi(e); if (v >= 1) ov_min_walkers = v; }
    if (const char *e = getenv("XRIT_OV_LRATIO")) { const double v = atof(e); if (v >= 0.125 && v <= 16.0) ov_lratio = v; }
#endif
    {
        // history a warm walker 0 needs in front of the new samples: its warm-up, the symbols it stages in front of the carried
        // position, the two symbols of history its start state interpolates
        const double wmax = (double)par.omega_mid + (double)par.omega_lim + 0.004;
        ov_pad_need = (int)ceil((double)ov_hist * wmax) + XR_MM_NTAPS + XR_MM_FUDGE + 6 * (int)ceil(wmax) + 64;
        ov_pad_need = (ov_pad_need + 15) & ~15;
        // (only where the LDS-staged walker applies; else the pad stays what the carried tail needs)
        const bool ring_ok = relay_span(par, 64) + 8 <= RELAY_RX - RELAY_XCH - 72;
        xpad = 1024;
        if (ov_enabled && ring_ok && (size_t)ov_pad_need > xpad) xpad = (size_t)ov_pad_need;
        XR_TRY(ov_claim.reserve(RELAY_CLAIM_WORDS * sizeof(unsigned)));
        XR_HIP(hipMemset(ov_claim.p, 0, RELAY_CLAIM_WORDS * sizeof(unsigned)));
        for (auto &j : ov)
            if (hipEventCreateWithFlags(&j.ev_walk, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&j.ev_guess, hipEventDisableTiming) != hipSuccess) { set_error("hipEventCreate failed"); return XRIT_E_HIP; }
    }
    // diagnostics, read once, here (a host that calls setenv races with getenv on a launch path): the fallback paths of the
    // shipped configuration
    force_gated = getenv("XRIT_GATED_SOLVE") != nullptr;
    relay_global = getenv("XRIT_RELAY_GLOBAL") != nullptr;
    trace_env = getenv("XRIT_TRACE") != nullptr;
#ifdef XRIT_EXPERIMENTS
    // A/B switches of the measurements in DESIGN.md (make EXTRA=-DXRIT_EXPERIMENTS: not in the shipped library)
    if (const char *e = getenv("XRIT_NO_HANDOFF")) relay_no_handoff = atoi(e) != 0 ? 1 : 0;
    if (const char *e = getenv("XRIT_RELAY_WAVES")) relay_waves = atoi(e);
    if (const char *e = getenv("XRIT_RELAY_TEAMS")) relay_teams_per_cu = atoi(e) > 0 ? atoi(e) : 1;
    relay_no_rec = getenv("XRIT_RELAY_NO_REC") != nullptr;
    relay_no_claim = getenv("XRIT_RELAY_NO_CLAIM") != nullptr;
    if (const char *e = getenv("XRIT_RELAY_REC01")) relay_rec01 = atoi(e) != 0;
    if (const char *e = getenv("XRIT_RELAY_APX")) { int a0 = -1, a1 = -1; if (sscanf(e, "%d,%d", &a0, &a1) >= 1) { relay_apx_cfg[0] = a0; relay_apx_cfg[1] = a1; } }
    if (const char *e = getenv("XRIT_AUTO_PASSES")) auto_passes = atoi(e);
    if (const char *e = getenv("XRIT_AUTO_LONG_SEG")) auto_long_seg = atoi(e);
    if (const char *e = getenv("XRIT_RELAY_PER_CU")) { relay_per_cu = atoi(e) > 0 ? atoi(e) : 3; relay_per_cu_set = true; }
    no_meanj = getenv("XRIT_NO_MEANJ") != nullptr;
    pass_writes = getenv("XRIT_NO_PASS_OUTPUT") == nullptr;
    ng_max = getenv("XRIT_CLOCK_NG") ? atoi(getenv("XRIT_CLOCK_NG")) : 8;
#endif
    return XRIT_OK;
}

__global__ void clock_state_reset_kernel(ClockState *st, float mu, float omega)
{
    ClockState s{};
    s.ii = 0; s.mu = mu; s.omega = omega;
    st[0] = s;
    st[1] = s;
}

// as after construction; what the stream taught (the mean chain Jacobian, the pass batch) is kept
int ClockStage::reset(hipStream_t s)
{
    hipLaunchKernelGGL(clock_state_reset_kernel, dim3(1), dim3(1), 0, s, st.as<ClockState>(), mu0, par.omega_mid);
    cur = 0;
    carry = 0;
    x_pending = -1;         // (samples a producer has put in place for a call that will not come)
    for (auto &j : ov) j.state = 0;     // (the caller has waited for whatever walked ahead: xrit_demod_reset synchronises its streams)
    ov_job = -1;
    hist_xb = -1; hist_len = 0; hist_job = -1;
    if (ov_claim.p) (void)hipMemsetAsync(ov_claim.p, 0, RELAY_CLAIM_WORDS * sizeof(unsigned), s);
    om_ext = false;         // (... and its timing statistic and count curve)
    om_scanned = false;
    in_flight = false;
    // what the stage has learnt from the stream it is leaving: as on a new handle (a reset handle gives a new handle's words)
    passes = 0;
    last_passes = -1;
    batch = 7;
    relay_batch = 96;
    jmean_valid = false;
    last_symbols = 0;
    prev_carry = 0;
    prev_n = 0;
    redo_ok = false;        // (nothing of an earlier call is left to run again, nor a flipped loop's state to start it from)
    alt_valid = false;
    return XRIT_OK;
}

void ClockStage::release()
{
#ifdef XRIT_RELAY_TIMING
    if (getenv("XRIT_WALKER_PHASES")) {
        // (everything the overlap walkers of this process spent since the last look: no copy inside the pipeline, which a
        // hipMemcpy on the null stream would serialise)
        unsigned long long hd[16] = {0}, z[16] = {0};
        if (hipDeviceSynchronize() == hipSuccess && hipMemcpyFromSymbol(hd, HIP_SYMBOL(relay_dbg), sizeof hd) == hipSuccess && hd[6]) {
            (void)hipMemcpyToSymbol(HIP_SYMBOL(relay_dbg), z, sizeof z);
            const double st = (double)hd[6];
            fprintf(stderr, "[xrit] overlap walkers over the handle's life, cycles per step: ring wait %.0f, setup %.0f, guess rounds %.0f, literal step + verdict %.0f, "
                            "stage + commit %.0f, loop %.0f; sum %.0f (%llu steps; the ring's fill level read %.3f times per step, %.3f sleeps per step)\n", hd[0] / st, hd[1] / st, hd[2] / st, hd[3] / st, hd[4] / st,
                    hd[5] / st, (hd[0] + hd[1] + hd[2] + hd[3] + hd[4] + hd[5]) / st, hd[6], hd[7] / st, hd[8] / st);
        }
    }
#endif
    for (auto &b : xbuf) b.release();
    for (auto &j : ov) {
        j.om.release(); j.om_work.release(); j.segs.release(); j.S.release(); j.stage.release(); j.aux.release();
        if (j.ev_walk) (void)hipEventDestroy(j.ev_walk);
        if (j.ev_guess) (void)hipEventDestroy(j.ev_guess);
        j.ev_walk = nullptr; j.ev_guess = nullptr;
    }
    ov_claim.release();
    table.release(); st.release(); S.release(); E.release(); J.release(); om.release();
    work.release(); counters.release(); sym.release(); dlin.release(); flags.release(); tail.release(); wsolve.release(); jmean.release();
    relay.release(); relay_rec.release(); alt.release(); stage.release(); om_work.release();
    if (h_res) (void)hipHostFree(h_res);
    h_res = nullptr;
}

// The producer of this call's samples may deliver the timing-line statistic itself: nb blocks of BL samples,
// block b covering buffer samples [offset + b*BL, ...).  Returns where to write it (double2 per block).
double2 *ClockStage::om_slot(int nb, int BL)
{
    if (ov_job >= 0 && ov[ov_job].state == 1) {
        // (an overlap job keeps its own statistic and curve: the walkers of the job behind it read them for their first ranges)
        OvJob &j = ov[ov_job];
        if (nb < 1 || j.om.reserve((size_t)nb * (sizeof(double2) + sizeof(double))) != XRIT_OK) return nullptr;
        j.om_ext = true; j.om_scanned = false; j.nb = nb; j.BL = BL;
        return j.om.as<double2>();
    }
    if (nb < 1 || om.reserve((size_t)nb * (sizeof(double2) + sizeof(double))) != XRIT_OK) return nullptr;
    om_ext = true;
    om_scanned = false;
    om_nb = nb;
    om_BL = BL;
    return om.as<double2>();
}

// The symbol-count curve of the producer's statistic (om_slot), unwrapped on the producer's stream behind the kernel that
// leaves the statistic.  Counted from the first NEW sample: the curve does not depend on what the previous call carries
// over (its integer values -- the symbol instants -- are the same whichever sample the phasor counts from, because the line
// turns once per symbol), so the next burst's curve is ready while this burst's relay still runs; begin() places it
// `carry` samples into the buffer.
int ClockStage::om_scan(hipStream_t s)
{
    if (ov_job >= 0 && ov[ov_job].state == 1) {
        OvJob &j = ov[ov_job];
        if (!j.om_ext || j.nb < 1) return XRIT_OK;
        // (an overlap job's curve is needed by its walkers' start states only: unwrapped on THEIR stream, ov_launch -- the
        // producer's stream, which sets a burst's pace, is spared two launches)
        if (!ov_scan_now) return XRIT_OK;
        const int nb = j.nb, nbB = scan_blocks(nb);
        XR_TRY(j.om_work.reserve((size_t)(nbB + 2) * sizeof(double)));
        double2 *X = j.om.as<double2>();
        double *cnt = reinterpret_cast<double *>(j.om.as<char>() + (size_t)nb * sizeof(double2));
        ClkUnwrapF uf{X, cnt, nb, (double)sps, 0.0, j.BL, 0.0};
        hipLaunchKernelGGL(scan_reduce_kernel<ClkUnwrapF>, dim3(nbB), dim3(SCAN_BLOCK), 0, s, uf, (long long)nb, j.om_work.as<double>());
        hipLaunchKernelGGL(scan_apply_lookback_kernel<ClkUnwrapF>, dim3(nbB), dim3(SCAN_BLOCK), 0, s, uf, (long long)nb, j.om_work.as<double>());
        XR_HIP(hipGetLastError());
        j.om_scanned = true;
        return XRIT_OK;
    }
    if (!om_ext || om_nb < 1) return XRIT_OK;
    const int nb = om_nb, nbB = scan_blocks(nb);
    XR_TRY(om_work.reserve((size_t)(nbB + 2) * sizeof(double)));
    double2 *X = om.as<double2>();
    double *cnt = reinterpret_cast<double *>(om.as<char>() + (size_t)nb * sizeof(double2));
    ClkUnwrapF uf{X, cnt, nb, (double)sps, 0.0, om_BL, 0.0};
    hipLaunchKernelGGL(scan_reduce_kernel<ClkUnwrapF>, dim3(nbB), dim3(SCAN_BLOCK), 0, s, uf, (long long)nb, om_work.as<double>());
    hipLaunchKernelGGL(scan_apply_lookback_kernel<ClkUnwrapF>, dim3(nbB), dim3(SCAN_BLOCK), 0, s, uf, (long long)nb, om_work.as<double>());
    XR_HIP(hipGetLastError());
    om_scanned = true;
    return XRIT_OK;
}

// where the producer writes the n new samples of this call: behind the `carry` samples left unread by the
// previous call (those are copied in from the tail buffer when the call begins)
int ClockStage::input_slot(size_t n, float2 **slot, hipStream_t s)
{
    // A buffer nobody reads: not the one of the call in flight, not one whose walkers are at work (overlap jobs), and not the
    // one that holds the stream's last samples -- the history the next overlap call's first walkers warm up over.
    bool busy[NXB] = {};
    if (in_flight) busy[xb] = true;
    for (int q = 0; q < NXB; ++q) if (ov[q].state != 0) busy[q] = true;
    int b = -1;
    // (the call that follows the stream's last samples may not overwrite them -- unless nothing else is free)
    for (int q = 0; q < NXB && b < 0; ++q) { const int c = (xb + 1 + q) % NXB; if (!busy[c] && c != hist_xb) b = c; }
    for (int q = 0; q < NXB && b < 0; ++q) { const int c = (xb + q) % NXB; if (!busy[c]) b = c; }
    if (b < 0) { set_error("clock recovery: no free sample buffer (%d calls are already in flight)", NXB); return XRIT_E_INVALID; }
    // (a job whose first walkers warm up over what this buffer holds must have read it: its pad copy and start states)
    for (int q = 0; q < NXB; ++q)
        if (ov[q].state >= 2 && ov[q].hist_src == b) XR_HIP(hipStreamWaitEvent(s, ov[q].ev_guess, 0));
    if (b == hist_xb) { hist_xb = -1; hist_len = 0; hist_job = -1; }
    XR_TRY(xbuf[b].reserve((xpad + n + 64 + 16) * sizeof(float2)));
    x_pending = b;
    *slot = xbuf[b].as<float2>() + xpad;
    ov_job = -1;
    if (ov_eligible(n)) {
        OvJob &j = ov[b];
        j.state = 1;
        j.n = n;
        j.om_ext = false; j.om_scanned = false; j.nb = 0;
        j.serial = ++ov_serial;
        ov_job = b;
    }
    return XRIT_OK;
}

// control block layout in `counters`: [0..24) ctl words, [24..32) ClockResult, per-pass counter slots from 32
constexpr int CLK_CTL_WORDS = 32;
constexpr int CLK_RES_WORD = 24;
static inline int *clock_ctl(const DevBuf &b) { return b.as<int>(); }
static inline ClockResult *clock_res(const DevBuf &b) { return reinterpret_cast<ClockResult *>(b.as<unsigned>() + CLK_RES_WORD); }
static inline unsigned *clock_cnt(const DevBuf &b, int pass) { return b.as<unsigned>() + CLK_CTL_WORDS + (size_t)pass * 8; }

// dynamic LDS beyond the default limit has to be asked for, once per kernel
template <typename KernelT> static void clock_allow_lds(KernelT kernel, size_t bytes)
{
    if (bytes > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

// ---- exact closure: host side ---------------------------------------------------------------------------------
// After a batch of relay passes: how many ran, whether they closed, and -- from the end states of the last pass
// that did something -- the call's result as clock_finalize_kernel leaves it (symbol count, carried state, tail).
__global__ void __launch_bounds__(256) clock_relay_finalize_kernel(const RelaySeg *__restrict__ e0,
                                                                    const RelaySeg *__restrict__ e1,
                                                                    const unsigned *__restrict__ changed, int enq, int G,
                                                                    int Lseg, const ClockState *__restrict__ carried_in,
                                                                    ClockState *__restrict__ carried_out,
                                                                    ClockResult *__restrict__ res,
                                                                    const float2 *__restrict__ x,
                                                                    float2 *__restrict__ tail_out, long long N, int *ctl, int force,
                                                                    const unsigned long long *__restrict__ moments)
{
    if (!ctl[0] && !force) return;      // the tiled hand-off did not close in its batch: ClockStage::finish starts over
    __shared__ int s_term, s_buf, s_stuck;
    __shared__ long long s_ii;
    if (threadIdx.x == 0) {
        int ran = enq, closed = 0;
        for (int p = 0; p < enq; ++p)
            if (changed[RELAY_STAT * p] == 0u) { ran = p + 1; closed = 1; break; }
        ctl[10] = ran;
        ctl[11] = closed;
        {
            // how far the starts still moved in the last pass: mean square (samples^2, float bits)
            const unsigned *c = changed + RELAY_STAT * (ran - 1);
            const unsigned long long sq = (unsigned long long)c[4] | ((unsigned long long)c[5] << 32);
            ctl[12] = __float_as_int(c[6] ? (float)((double)sq / 1099511627776.0 / (double)c[6]) : 0.0f);
            ctl[16] = (int)(c[3] >= 0x80000000u ? 0x7f800000u : c[3]);     // ... and the largest move (a watchdog mark: infinity)
        }
        s_buf = (ran - 1) & 1;
        s_term = 0x7fffffff;
        s_stuck = 0;
    }
    __syncthreads();
    const RelaySeg *e = s_buf ? e1 : e0;
    for (int i = threadIdx.x; i < G; i += (int)blockDim.x) {
        if (e[i].flags & RELAY_EXHAUSTED) atomicMin(&s_term, i);
        // (a walk that a watchdog ended is not the end of the input: the call fails, ClockStage::finish)
        if (e[i].flags & RELAY_STUCK) s_stuck = 1;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        ClockState s;
        const int k = s_stuck ? G : s_term;
        ctl[15] = s_stuck;
        if (k >= G) {
            res->ok = 0;
            res->n_symbols = 0;
            res->terminal_chain = -1;
            s = carried_in[0];
        } else {
            res->ok = 1;
            res->terminal_chain = k;
            res->n_symbols = (unsigned long long)k * (unsigned long long)Lseg + (unsigned long long)e[k].n_done;
            s = e[k].s;
        }
        {
            // BPSK soft symbols s = +-A + noise: (mean |s|)^2 / (mean s^2 - (mean |s|)^2) ~ A^2 / sigma^2 = 2 Es/N0 (float bits)
            const double nsym = res->ok ? (double)res->n_symbols : 0.0;
            const double a1 = nsym > 0 ? (double)moments[0] / 1048576.0 / nsym : 0.0, a2 = nsym > 0 ? (double)moments[1] / 1048576.0 / nsym : 0.0;
            const double var = a2 - a1 * a1;
            ctl[14] = __float_as_int(nsym > 0 && var > 0 ? (float)(a1 * a1 / var) : 1e30f);
        }
        long long ii = s.ii;
        if (ii > N) ii = N;
        if (ii < 0) ii = 0;
        res->ii_final = ii;
        s_ii = ii;
        s.ii = 0;
        carried_out[0] = s;
    }
    __syncthreads();
    const long long ii = s_ii;
    long long carry = N - ii;
    // (the hand-over slot holds 1024 samples, like clock_finalize_kernel's: a walk that did not reach the end of the input --
    // no segment ran out of samples, or a walker gave up -- leaves more than that unread and the call fails, ClockStage::finish)
    if (carry > 1024) {
        carry = 1024;
        if (threadIdx.x == 0) res->ok = 0;
    }
    for (long long i = threadIdx.x; i < carry; i += blockDim.x) tail_out[i] = x[ii + i];
}

// segments of the exact closure: 3 per CU (what its LDS holds of the staged walk) unless a window is given; the relay
// buffer holds three segment records per segment and four counters per pass
// samples a block of `symbols` symbols can cover at the fastest admissible clock
static int relay_span(const ClockPar &par, int symbols)
{
    return (int)ceil((double)symbols * ((double)par.omega_mid + (double)par.omega_lim + 0.004)) + 24;
}

int ClockStage::relay_plan()
{
    Job &j = job;
    // the walker: a team of 8, 4 or 2 waves (clock_relay_wide.h) where a block of 62 symbols per wave fits the team's sample
    // ring, else one wave (64 symbols per step; any symbol rate: clock_relay.h)
    j.relay_w = 0;
#ifdef XRIT_EXPERIMENTS
    if (!relay_global && relay_waves >= 2) {
        if (relay_waves >= 4 && relay_span(par, RW_OWN * 4) + 8 <= RelayWide<4>::MAX_SPAN) j.relay_w = 4;
        else if (relay_span(par, RW_OWN * 2) + 8 <= RelayWide<2>::MAX_SPAN) j.relay_w = 2;
    }
#endif
    // (without hand-off passes: two walkers per CU -- 24.8 k symbols per segment at C2, three passes; measured in the streamed
    // bench against one per CU with two passes and three per CU with four: 2.12 / 2.35 / 2.27 ms per burst)
    // ... and where the call is long enough for MORE walkers whose segments still hold auto_long_seg symbols -- the two-pass plan:
    // bursts at the circuit rate, 63 M (LRIT) / 100 M (HRIT) symbols per 2^28 samples -- up to four per CU, one per SIMD: the
    // passes' latency is a segment's length, and a two-pass plan over 62 k / 97 k-symbol segments has the same exact history
    // as over 123 k / 195 k (C3: 6.5 instead of 9.6 ms per burst, C1: 5.7 instead of 6.2, profiles/r4_relay_shortcuts.txt))
    int auto_per_cu = 2;
    if (j.no_handoff && exact == 0)
        for (int p = 4; p > 2; --p)
            if ((long long)j.K * NS / ((long long)p * cu_count) >= auto_long_seg) { auto_per_cu = p; break; }
    const int per_cu = j.relay_w > 0 ? relay_teams_per_cu : (j.no_handoff && !relay_per_cu_set ? auto_per_cu : relay_per_cu);
    int cps = relay_window > 0 ? relay_window : (j.K + per_cu * cu_count - 1) / (per_cu * cu_count);
    // (a call much shorter than the ~1e5 symbols two trajectories need to meet is walked front to back whatever the
    // cut: segments of at least 2048 symbols then cost the fewest passes -- a pass is a launch)
    // With a budget of relay passes (cfg.clock_exact = n > 1) what the passes buy is their horizon, n x the segment length
    // -- a segment is exact once everything within the merge length in front of it is: segments no shorter than the big
    // bursts' (16 k symbols), whatever the size of the call, so that n means the same parity everywhere.
    // (without hand-off passes -- the default configuration -- a segment must be long enough for the loop to forget the timing
    // guess it starts from: never shorter than auto_long_seg / 2, 24.6 k symbols, which three passes go with.  A call of up
    // to three such segments is ONE segment instead -- one exact walk from the carried state takes no longer than three
    // passes over a third of it, and it IS the serial trajectory.)
    int min_syms = j.relay_budget > 0 ? 16384 : 2048;
    if (j.no_handoff && exact == 0 && !relay_per_cu_set) {
        const long long all = (long long)j.K * NS;
        min_syms = all <= one_walk_limit() ? (int)all : auto_long_seg / 2;
    }
    if (relay_window <= 0 && cps * NS < min_syms) cps = (min_syms + NS - 1) / NS;
    if (cps < 1) cps = 1;
    j.cps = cps;
    j.G = (j.K + cps - 1) / cps;
    // (counters for G + 1 passes whatever the budget: the default configuration raises its own, ClockStage::finish)
    XR_TRY(relay.reserve((size_t)j.G * 3 * sizeof(RelaySeg) + ((size_t)j.G + 1 + 8) * RELAY_STAT * sizeof(unsigned) + 2 * sizeof(unsigned long long) +
                         RELAY_CLAIM_WORDS * sizeof(unsigned)));
    if (!relay_no_rec) XR_TRY(relay_rec.reserve(((size_t)j.G * cps * NS + RW_REC_PAD) * sizeof(unsigned)));
    relay_segments = j.G;
    relay_seg_chains = cps;
    return XRIT_OK;
}

int ClockStage::relay_limit() const
{
    // a pass moves the exact front by at least one segment: G + 1 passes always close
    const int hard = job.G + 1;
    return job.relay_budget > 0 ? (job.relay_budget < hard ? job.relay_budget : hard) : hard;
}

// `restart`: first batch of a call (or the tiled evaluation was redone): nothing has been walked.  Every batch ends
// with the finalize kernel and the copy of the control block.
int ClockStage::enqueue_relay(int count, bool restart, hipStream_t s, Profiler *prof)
{
    Job &j = job;
    const int limit = relay_limit();
    RelaySeg *segs = relay.as<RelaySeg>();
    unsigned *changed = reinterpret_cast<unsigned *>(segs + 3 * (size_t)j.G);
    RelayArgs a{};
    a.x = xbase(); a.table = table.as<float>(); a.N = j.N; a.ni = j.ni;
    a.first = st.as<ClockState>() + cur; a.S = S.as<ClockState>();
    a.K = j.K; a.cps = j.cps; a.NS = NS; a.G = j.G;
    a.start = segs; a.ends[0] = segs + j.G; a.ends[1] = segs + 2 * (size_t)j.G;
    // steps of the float32 lattice omega and mu + omega live on, in units of 2^-24 sample (the walkers' integer model)
    {
        const float om = par.omega_mid, su = par.omega_mid + 0.5f;
        const double u = 1.0 / 16777216.0;
        a.q_om = (int)((double)(nextafterf(om, INFINITY) - om) / u);
        a.q_mu = (int)((double)(nextafterf(su, INFINITY) - su) / u);
        if (a.q_om < 1) a.q_om = 1;
        if (a.q_mu < 1) a.q_mu = 1;
    }
    a.soft = j.soft; a.sym = j.sym; a.cap = (unsigned long long)j.cap; a.par = par;
    a.changed = changed; a.ctl = j.relay_force ? nullptr : clock_ctl(counters);
    a.rec = relay_no_rec ? nullptr : relay_rec.as<unsigned>();
    a.moments = reinterpret_cast<unsigned long long *>(changed + ((size_t)j.G + 1 + 8) * RELAY_STAT);
    a.simd_claim = relay_no_claim ? nullptr : reinterpret_cast<unsigned *>(a.moments + 2);

    if (restart) {
        j.relay_enq = 0;
        const int words = RELAY_STAT * (j.G + 3) > RELAY_CLAIM_WORDS ? RELAY_STAT * (j.G + 3) : RELAY_CLAIM_WORDS;
        hipLaunchKernelGGL(clock_relay_init_kernel, dim3(div_up((size_t)words, 256)), dim3(256), 0, s, segs, j.G, changed,
                           j.G + 3, clock_ctl(counters), a.moments, a.simd_claim);
    }
    // samples a block of 64 symbols can cover; the LDS-staged walk takes what fits its refill chunk
    const int span = relay_span(par, 64);
    const bool lds_walk = span + 8 <= RELAY_RX - RELAY_XCH - 72 && !relay_global;
    {
        ProfScope ps(prof, "clock_relay", s);
        for (int q = 0; q < count && j.relay_enq < limit; ++q, ++j.relay_enq) {
            const int apx = j.relay_enq < 2 ? j.relay_apx[j.relay_enq] : 0;      // (the LDS-staged one-wave walker only)
            a.sym_skip = j.no_handoff && j.relay_enq == 0 && limit >= 2 && !apx;
            // (every pass leaves its record and guesses from the one before: since the records hold advances and a walk anchors
            // them at its own state, with the slope of the block before, even the record of a walk from the timing guess -- 4e-2
            // sample away -- saves the next pass a guess round per step: 1.28 instead of 2.30)
            a.rec_write = relay_rec01 || !(j.no_handoff && j.relay_enq == 0);
            a.rec_use = (relay_rec01 || !(j.no_handoff && j.relay_enq == 1)) ? 3 : 0;      // (bit 1: the next block's record is read ahead)
#ifdef XRIT_EXPERIMENTS
#define XR_RELAY_WIDE(WV)                                                                                             \
    do {                                                                                                              \
        const int wspan = relay_span(par, RW_OWN * WV);                                                               \
        if (j.sym) hipLaunchKernelGGL((clock_relay_wide_kernel<true, WV>), dim3(j.G), dim3(64 * WV), 0, s, a, j.relay_enq, wspan);       \
        else hipLaunchKernelGGL((clock_relay_wide_kernel<false, WV>), dim3(j.G), dim3(64 * WV), 0, s, a, j.relay_enq, wspan);            \
    } while (0)
            if (j.relay_w == 4) XR_RELAY_WIDE(4);
            else if (j.relay_w == 2) XR_RELAY_WIDE(2);
            else
#undef XR_RELAY_WIDE
#endif
            if (lds_walk && j.sym) hipLaunchKernelGGL((clock_relay_kernel<true, true>), dim3(j.G), dim3(128), 0, s, a, j.relay_enq, span, apx);
            else if (lds_walk) hipLaunchKernelGGL((clock_relay_kernel<false, true>), dim3(j.G), dim3(128), 0, s, a, j.relay_enq, span, apx);
            else if (j.sym) hipLaunchKernelGGL((clock_relay_kernel<true, false>), dim3(j.G), dim3(64), 0, s, a, j.relay_enq, span, apx);
            else hipLaunchKernelGGL((clock_relay_kernel<false, false>), dim3(j.G), dim3(64), 0, s, a, j.relay_enq, span, apx);
        }
        // (ONE wave: a block of 1024 needs sixteen free wave slots on one CU and, with the next burst's front end filling
        // the chip behind the relay, waited ~90 us for them; four waves still waited 75 us behind the 2.4 ms matched filter
        // of a burst at the circuit rate, profiles/r4_c1_kernel_stats.csv)
        hipLaunchKernelGGL(clock_relay_finalize_kernel, dim3(1), dim3(64), 0, s, a.ends[0], a.ends[1], changed,
                           j.relay_enq, j.G, j.cps * NS, st.as<ClockState>() + cur, st.as<ClockState>() + (cur ^ 1),
                           clock_res(counters), a.x, tail.as<float2>() + 1024 * (cur ^ 1), j.N, clock_ctl(counters), j.relay_force ? 1 : 0,
                           a.moments);
    }
    XR_HIP(hipGetLastError());
    XR_HIP(hipMemcpyAsync(h_res, counters.p, CLK_CTL_WORDS * sizeof(unsigned), hipMemcpyDeviceToHost, s));
    return XRIT_OK;
}

int ClockStage::enqueue_passes(int count, hipStream_t s, Profiler *prof)
{
    const Job &j = job;
    ClockPolicy pol{S.as<ClockState>(), E.as<ClockState>(), J.as<float4>(), j.dirty, j.nrun, nullptr,
                    0.75f, 0.01f, tol_t, tol_w, min_passes, j.mean_j ? jmean.as<float4>() : nullptr,
                    // (3e-4 sample rms at the 2.7 .. 4.25 samples per symbol the rule was tuned at; the floor is a
                    // fraction of a SYMBOL: at 21 or 68 samples per symbol it sits that much higher in samples, and
                    // calls there went on freezing a few boundaries per pass for 185 passes)
                    9e-8f * (sps > 4.2534f ? (sps / 4.2534f) * (sps / 4.2534f) : 1.0f),
                    // (measured at C2: the walks meet the serial trajectory after as many relay passes from the starts two
                    // hand-off passes leave -- rms residual 8e-4 sample -- as from those of five, 1.2e-4)
                    // (segments of a few hundred symbols -- cfg.clock_exact_window -- do not get that far in a pass: they start from
                    // a hand-off at its floor as before)
                    j.relay && (long long)j.cps * NS >= 2048 ? 2 : 0};
    const unsigned nw = div_up((size_t)j.K, 64);              // waves of 64 chains
    const float2 *x = xbase();
    // wave-aligned solve (newton.h): the pass leaves its waves' aggregates, one more launch applies them;
    // `gated`: the three-launch solve with the trust gate (after a take-over, or when asked for)
    const bool wave = !job.gated;
    AffMap *aggs = wave ? wsolve.as<AffMap>() : nullptr;
    NewtonStat *wslots = reinterpret_cast<NewtonStat *>(wsolve.as<AffMap>() + nw + 1);
    for (int q = 0; q < count && job.enqueued < max_passes; ++q, ++job.enqueued) {
        const int p = job.enqueued;
        pol.cnt = clock_cnt(counters, p);
        const bool jac = p < jac_passes && !j.mean_j;
        {
            ProfScope ps(prof, jac ? "clock_pass_jac" : "clock_pass", s);
#define XR_CLK_PASS(NV, WPV, NCM, OUTV)                                                                               \
    do {                                                                                                              \
        const size_t lds = NV > 1 ? j.tile_bytes3 : j.tile_bytes;                                                     \
        clock_allow_lds(clock_pass_kernel<NV, WPV, NCM, OUTV>, lds);                                                  \
        hipLaunchKernelGGL((clock_pass_kernel<NV, WPV, NCM, OUTV>), dim3(NV > 1 ? nw : div_up(nw, j.NG)),             \
                           dim3(NV > 1 ? 64 * NV : 64 * j.NG), lds, s, x, table.as<float>(), S.as<ClockState>(),      \
                           E.as<ClockState>(), J.as<float4>(), j.dirty, j.nrun, j.N, j.ni, j.K, NS, par, j.SS, j.W,   \
                           j.WS, j.A, j.STEP, clock_ctl(counters), pol, aggs, po);                                    \
    } while (0)
#define XR_CLK_PASS_NV(NV, OUTV)                                                                                      \
    do {                                                                                                              \
        if (!j.wide && narrow) XR_CLK_PASS(NV, 32, 20, OUTV);                                                         \
        else if (!j.wide) XR_CLK_PASS(NV, 32, 32, OUTV);                                                              \
        else XR_CLK_PASS(NV, 64, 64, OUTV);                                                                           \
    } while (0)
            const bool narrow = (j.STEP >> 16) + 1 <= 20;      // columns a sub-step adds to a ring
            // from the pass the stop test is expected to fire after (what the previous call needed, never before the
            // fourth: ClockPolicy::decide stops no earlier) the passes leave the symbols themselves
            const bool writes = !jac && !j.relay && p >= j.write_from && j.SS == 4 && (NS & 3) == 0 && NS <= CLK_STAGE_MAX_NS;
            const ClockPassOut po{j.soft, j.sym, (unsigned long long)j.cap, clock_ctl(counters) + CLK_CTL_SYMBOLS, j.written,
                                  stage.as<float4>()};
            if (jac) XR_CLK_PASS_NV(3, false);
            else if (writes) XR_CLK_PASS_NV(1, true);
            else XR_CLK_PASS_NV(1, false);
#undef XR_CLK_PASS_NV
#undef XR_CLK_PASS
        }
        {
            ProfScope ps(prof, "clock_solve", s);
            if (wave) {
                newton_apply_waves(pol, (long long)j.K - 1, aggs, clock_ctl(counters), wslots, s);
            } else if (newton_solve(pol, (long long)j.K - 1, work.as<AffMap>(), dlin.as<float2>(), clock_ctl(counters), s) != 0) {
                set_error("clock hand-off: %d chains exceed the solver's block budget", j.K);
                return XRIT_E_INVALID;
            }
        }
    }
    XR_HIP(hipGetLastError());
    return XRIT_OK;
}

int ClockStage::enqueue_output(hipStream_t s, Profiler *prof, bool again)
{
    const Job &j = job;
    const float2 *x = xbase();
    const ClockState *st_in = st.as<ClockState>() + cur;
    ClockState *st_out = st.as<ClockState>() + (cur ^ 1);
    float2 *tail_out = tail.as<float2>() + 1024 * (cur ^ 1);
    const unsigned nw = div_up((size_t)j.K, 64);
    {
        ProfScope ps(prof, "clock_output", s);
        // (the first output pass of a call finds the marker set by clock_reset_kernel)
        if (again) hipLaunchKernelGGL(clk_fill_int_kernel, dim3(1), dim3(1), 0, s, j.terminal, 0x7fffffff, 1);
#define XR_CLK_OUT_S(WPV, NCM, SYMV)                                                                                  \
    do {                                                                                                              \
        clock_allow_lds(clock_output_kernel<WPV, NCM, SYMV>, j.tile_bytes);                                           \
        hipLaunchKernelGGL((clock_output_kernel<WPV, NCM, SYMV>), dim3(div_up(nw, j.NG)), dim3(64 * j.NG),            \
                           j.tile_bytes, s, x, table.as<float>(), S.as<ClockState>(), E.as<ClockState>(), j.counts,   \
                           j.soft, j.sym, (unsigned long long)j.cap, j.N, j.ni, j.K, NS, par, j.terminal, j.SS, j.W,  \
                           j.WS, j.A, j.STEP, clock_ctl(counters));                                                   \
    } while (0)
#define XR_CLK_OUT(WPV, NCM)                              \
    do {                                                  \
        if (j.sym) XR_CLK_OUT_S(WPV, NCM, true);          \
        else XR_CLK_OUT_S(WPV, NCM, false);               \
    } while (0)
        const bool narrow = (j.STEP >> 16) + 1 <= 20;
        if (!j.wide && narrow) XR_CLK_OUT(32, 20);
        else if (!j.wide) XR_CLK_OUT(32, 32);
        else XR_CLK_OUT(64, 64);
#undef XR_CLK_OUT_S
#undef XR_CLK_OUT
        if (j.soft && j.SS == 4 && (NS & 3) == 0 && pass_writes && NS <= CLK_STAGE_MAX_NS) {
            const size_t lds = (size_t)64 * (NS + 1) * sizeof(float) + 64 * sizeof(int);
            clock_allow_lds(clock_unstage_kernel, lds);
            hipLaunchKernelGGL(clock_unstage_kernel, dim3(nw), dim3(256), lds, s, stage.as<float4>(), j.soft, j.nrun,
                               clock_ctl(counters), (unsigned long long)j.cap, j.K, NS);
        }
        hipLaunchKernelGGL(clock_terminal_kernel, dim3(div_up((size_t)j.K, 256)), dim3(256), 0, s, j.nrun, clock_ctl(counters),
                           j.terminal, j.K, NS);
        hipLaunchKernelGGL(clock_finalize_kernel, dim3(1), dim3(1024), 0, s, E.as<ClockState>(), j.counts, j.nrun, j.terminal,
                           clock_ctl(counters), st_in, st_out, clock_res(counters), x, tail_out, j.N, j.K, NS);
    }
    XR_HIP(hipGetLastError());
    XR_HIP(hipMemcpyAsync(h_res, counters.p, CLK_CTL_WORDS * sizeof(unsigned), hipMemcpyDeviceToHost, s));
    return XRIT_OK;
}

// Everything of one call is put on the stream without waiting: the carried tail, the timing guess, a batch of
// ---- overlapping exactly walked blocks: host side (clock_overlap.h) -------------------------------------------------
bool ClockStage::ov_eligible(size_t n) const
{
    if (!ov_enabled || !ov_allow || exact != 0 || relay_quick || serial || relay_global || xbase_fixed) return false;
    if (relay_no_handoff == 0 || auto_passes <= 0) return false;          // (A/B switches that ask for round 3's / round 2's plans)
    if (relay_window > 0) return false;                                    // (cfg.clock_exact_window: the caller asks for the relay's segments)
    if ((size_t)ov_pad_need > xpad) return false;                          // (no LDS-staged walker at this symbol rate)
    if (n >= ((size_t)1 << 31) - xpad - 4096) return false;
    return (double)n / (double)par.omega_mid >= (double)ov_min;
}

// the job's ranges.  `ahead`: the call in front has not finished -- its carry is not known; such a job needs the history.
int ClockStage::ov_plan(OvJob &j, bool ahead)
{
    const double wmax = (double)par.omega_mid + (double)par.omega_lim + 0.004, wmin = (double)par.omega_mid - (double)par.omega_lim;
    const int cw = (int)ceil(wmax);
    j.hist = (int)ceil((double)ov_hist * (double)par.omega_mid);
    j.early = (int)ceil(1.5 * wmax) + 2;
    // history: the samples the call in front left at the end of its buffer, and the timing curve that covers them
    const bool have_hist = hist_xb >= 0 && hist_len >= (size_t)ov_pad_need && hist_job >= 0 && ov[hist_job].om_scanned;      // (scanned: the job in front has been launched)
    if (ahead && !have_hist) { set_error("clock recovery: an overlap job cannot start ahead without the history of the call in front"); return XRIT_E_INVALID; }
    j.ahead = ahead;
    if (have_hist) {
        j.padN = ov_pad_need;
        j.w0_carried = false;
        j.w0_ii = ahead ? 0 : j.padN - (int)carry;          // (ahead: the scan kernel is told at the call, ov_finalize)
        // the carried state reads its next symbol 19 .. 24 + a period samples in front of the new ones (the walk in front stops at
        // the first read index within 8 + 16 samples of its end): staging starts a period in front of that
        j.store0 = j.padN - (XR_MM_NTAPS + XR_MM_FUDGE) - cw - 2;
    } else {
        if (carry > 1024) { set_error("clock recovery: carry of %zu samples exceeds the hand-over buffer", carry); return XRIT_E_INVALID; }
        j.padN = (int)carry;
        j.w0_carried = true;
        j.w0_ii = 0;
        j.store0 = 0;
    }
    j.N = (long long)j.padN + (long long)j.n;
    j.ni = j.N - XR_MM_NTAPS - XR_MM_FUDGE;
    // walkers: ranges about as long as the history in front of them (every symbol is walked twice), at least 240 of them where
    // the call is long enough for ranges of 8192 symbols (few walkers: their latency is the call's), at most four per CU
    const double nsym = (double)j.n / (double)par.omega_mid;
    double gt = nsym / ((double)ov_hist * ov_lratio);
    if (gt < (double)ov_min_walkers) gt = (double)ov_min_walkers;
    if (nsym / gt < (double)ov_min_range) gt = nsym / (double)ov_min_range;
    // (at most 2.5 per CU: bursts at the circuit rate -- 63 M / 100 M symbols -- would take four per CU by the rule above; with 640
    // walkers of 98 k / 156 k symbols every symbol is walked 1.5 / 1.3 times instead of 1.8 / 1.5: C1 4.8 instead of 5.1 ms per
    // burst, C3 5.4 instead of 5.55 -- their latency is hidden like any other's)
    if (gt > 2.5 * cu_count) gt = 2.5 * cu_count;
    if (gt < 1.0) gt = 1.0;
    long long Ls = (long long)ceil((double)j.n / gt);
    Ls = (Ls + 63) & ~63LL;
    if (Ls < 4096) Ls = 4096;
    j.Ls = (int)Ls;
    // range 0 ends where walker 1 has its whole history behind it
    const long long w0_start = j.w0_carried ? 0 : (long long)j.store0 - j.hist;
    long long fb = w0_start + j.hist + Ls;
    if (fb < (long long)j.padN + Ls) fb = (long long)j.padN + Ls;
    j.first_bound = (int)fb;
    // (the last range is not shorter than a quarter of the others)
    const long long lim = j.N - Ls / 4;
    j.G = fb <= lim ? 2 + (int)((lim - fb) / Ls) : 1;
    const long long last_start = j.G >= 2 ? fb + (long long)(j.G - 2) * Ls : (long long)j.store0;
    long long longest = fb - j.store0;
    if (j.G >= 2 && Ls + j.early > longest) longest = Ls + j.early;
    if (j.N - last_start + j.early > longest) longest = j.N - last_start + j.early;
    if (j.G == 1) longest = j.N - j.store0;
    long long stride = (long long)ceil((double)longest / wmin) + 64;
    stride = (stride + 63) & ~63LL;
    if (stride * j.G >= (1LL << 31)) { set_error("clock recovery: call too long for the overlap plan"); return XRIT_E_INVALID; }
    j.stride = (int)stride;
    XR_TRY(j.segs.reserve((size_t)j.G * sizeof(OverlapSeg)));
    XR_TRY(j.S.reserve((size_t)j.G * sizeof(ClockState)));
    XR_TRY(j.stage.reserve((size_t)j.G * (size_t)j.stride * sizeof(float) + 256));
    // aux: [0, 8) counters, [8, 12) moments (two 64-bit words), then j0[G] and offs[G] (64-bit, 8-byte aligned)
    XR_TRY(j.aux.reserve(64 + (size_t)j.G * sizeof(int) + 8 + (size_t)j.G * sizeof(unsigned long long)));
    return XRIT_OK;
}

bool ClockStage::ov_can_launch_ahead(int job) const
{
    if (job < 0 || job >= NXB || ov[job].state != 1 || !ov[job].om_ext) return false;
    // the job in front must be an overlap job whose samples and timing curve are in place (its walkers may still be at work)
    for (int q = 0; q < NXB; ++q)
        if (q != job && ov[q].state >= 1 && ov[q].serial + 1 == ov[job].serial)
            return ov[q].om_scanned && (size_t)ov[q].padN + ov[q].n >= (size_t)ov_pad_need && ov[q].state >= 2;
    return false;
}

static inline unsigned *ov_stat(ClockStage::OvJob &j) { return j.aux.as<unsigned>(); }
static inline unsigned long long *ov_moments(ClockStage::OvJob &j) { return reinterpret_cast<unsigned long long *>(j.aux.as<unsigned>() + 8); }
static inline unsigned long long *ov_offs(ClockStage::OvJob &j) { return reinterpret_cast<unsigned long long *>(j.aux.as<char>() + 64); }
static inline int *ov_j0(ClockStage::OvJob &j) { return reinterpret_cast<int *>(j.aux.as<char>() + 64 + (size_t)j.G * sizeof(unsigned long long)); }

int ClockStage::ov_launch(int job, hipStream_t sw, bool ahead, Profiler *prof)
{
    OvJob &j = ov[job];
    // the history: the job in front when launched ahead (it has not finished: hist_* still describe the call before it)
    int h_xb = hist_xb, h_job = hist_job;
    size_t h_len = hist_len;
    size_t h_n = 0;
    if (ahead) {
        int front = -1;
        for (int q = 0; q < NXB; ++q) if (q != job && ov[q].state >= 1 && ov[q].serial + 1 == j.serial) front = q;
        if (front < 0) { set_error("clock recovery: no job in front of a job launched ahead"); return XRIT_E_INVALID; }
        h_xb = front; h_job = front; h_len = (size_t)ov[front].padN + ov[front].n;
        XR_HIP(hipStreamWaitEvent(sw, ov[front].ev_guess, 0));      // (the curve of the job in front is unwrapped on ITS walkers' stream)
        // (the plan looks at hist_*: described for the duration of the plan)
        const int k_xb = hist_xb, k_job = hist_job; const size_t k_len = hist_len;
        hist_xb = h_xb; hist_job = h_job; hist_len = h_len;
        const int rc = ov_plan(j, true);
        hist_xb = k_xb; hist_job = k_job; hist_len = k_len;
        XR_TRY(rc);
    } else {
        XR_TRY(ov_plan(j, false));
    }
    float2 *data = xbuf[job].as<float2>() + xpad;
    float2 *base = data - j.padN;
    if (!j.w0_carried) {
        // the last padN samples of the stream in front of the new ones
        if (!(h_job >= 0 && h_xb == h_job) || h_len < (size_t)j.padN) { set_error("clock recovery: history without its job"); return XRIT_E_INVALID; }
        h_n = ov[h_job].n;
        const float2 *hend = xbuf[h_xb].as<float2>() + xpad + h_n;
        XR_HIP(hipMemcpyAsync(base, hend - j.padN, (size_t)j.padN * sizeof(float2), hipMemcpyDeviceToDevice, sw));
        j.hist_src = h_xb;
    } else if (j.padN > 0) {
        hipLaunchKernelGGL(clock_tail_kernel, dim3(1), dim3(1024), 0, sw, tail.as<float2>() + 1024 * cur, base, j.padN);
    }
    // the timing curve: the producer's (Costas final pass + om_scan), else computed here
    const int BL = j.om_ext ? j.BL : CLK_OM_BLOCK;
    if (!j.om_ext) {
        j.nb = (int)((j.n + CLK_OM_BLOCK - 1) / CLK_OM_BLOCK);
        j.BL = CLK_OM_BLOCK;
        XR_TRY(j.om.reserve((size_t)j.nb * (sizeof(double2) + sizeof(double))));
        hipLaunchKernelGGL(clock_om_kernel, dim3(div_up((size_t)j.nb, 4)), dim3(256), 0, sw, data, j.om.as<double2>(), (long long)j.n, j.nb,
                           1.0 / (double)sps);
        j.om_ext = true;
    }
    if (!j.om_scanned) {
        const int k_job = ov_job;
        const int k_state = j.state;
        ov_job = job; j.state = 1;
        ov_scan_now = true;
        const int rc = om_scan(sw);
        ov_scan_now = false;
        ov_job = k_job; j.state = k_state;
        XR_TRY(rc);
    }
    const double *cnt = reinterpret_cast<const double *>(j.om.as<char>() + (size_t)j.nb * sizeof(double2));
    const double *cnt_prev = nullptr;
    int nb_prev = 0;
    if (!j.w0_carried && h_job >= 0) {
        OvJob &p = ov[h_job];
        cnt_prev = reinterpret_cast<const double *>(p.om.as<char>() + (size_t)p.nb * sizeof(double2));
        nb_prev = p.nb;
        if (p.BL != BL) { set_error("clock recovery: timing curves of consecutive calls differ in their block length"); return XRIT_E_INVALID; }
    }
    OverlapArgs a{};
    a.x = base; a.table = table.as<float>(); a.N = j.N; a.ni = j.ni;
    a.start = j.S.as<ClockState>(); a.G = j.G; a.store0 = j.store0; a.first_bound = j.first_bound; a.Ls = j.Ls; a.early = j.early;
    a.stride = j.stride; a.stage = j.stage.as<float>(); a.segs = j.segs.as<OverlapSeg>(); a.par = par;
    {
        const float om = par.omega_mid, su = par.omega_mid + 0.5f;
        const double u = 1.0 / 16777216.0;
        a.q_om = (int)((double)(nextafterf(om, INFINITY) - om) / u);
        a.q_mu = (int)((double)(nextafterf(su, INFINITY) - su) / u);
        if (a.q_om < 1) a.q_om = 1;
        if (a.q_mu < 1) a.q_mu = 1;
    }
    a.stat = ov_stat(j); a.moments = ov_moments(j); a.simd_claim = relay_no_claim ? nullptr : ov_claim.as<unsigned>();
    const int span = relay_span(par, 64);
    j.hist_src = j.w0_carried ? -1 : j.hist_src;
    {
        ProfScope ps(prof, "clock_guess", sw);
        hipLaunchKernelGGL(clock_overlap_guess_kernel, dim3(div_up((size_t)j.G, 256)), dim3(256), 0, sw, cnt, j.nb, cnt_prev, nb_prev,
                           (long long)h_n, BL, (double)sps, par.omega_mid, base, table.as<float>(), j.ni, j.padN, j.hist, j.G, j.store0,
                           j.first_bound, j.Ls, st.as<ClockState>() + cur, j.w0_carried ? 1 : 0, j.w0_ii, 0, j.S.as<ClockState>(),
                           a.stat, a.moments);
    }
    XR_HIP(hipEventRecord(j.ev_guess, sw));       // (from here on the history this job was started from may be overwritten)
    {
        ProfScope ps(prof, "clock_overlap", sw);
        if (ov_small_ring && span + 8 <= 1024 - RELAY_XCH - 72) hipLaunchKernelGGL(clock_overlap_kernel<1024>, dim3(j.G), dim3(128), 0, sw, a, span);
        else hipLaunchKernelGGL(clock_overlap_kernel<RELAY_RX>, dim3(j.G), dim3(128), 0, sw, a, span);
    }
    XR_HIP(hipGetLastError());
    XR_HIP(hipEventRecord(j.ev_walk, sw));
    j.state = 2;
    return XRIT_OK;
}

int ClockStage::ov_restart(int job, hipStream_t s)
{
    (void)s;
    if (job < 0 || job >= NXB) return XRIT_OK;
    OvJob &j = ov[job];
    if (j.state == 2) XR_HIP(hipEventSynchronize(j.ev_walk));
    if (j.state >= 1) { j.state = 1; j.om_scanned = false; }
    return XRIT_OK;
}

// a call that began as an overlap job and whose result is not taken (low Es/N0: the default walks such calls to closure; a joint
// that did not fit: a timing guess off by a symbol): the relay of clock_relay.h, to closure, on the same samples
int ClockStage::ov_fallback(size_t *n_out, hipStream_t s, Profiler *prof, bool to_closure)
{
    const int job = ov_cur;
    OvJob &j = ov[job];
    ov_cur = -1;
    j.state = 0;
    const int k_exact = exact;
    // (no signal -- the walkers of a loop that is not locked do not meet at the joints --: the relay's own default, which does not
    // pursue a closure on such an input either; ClockStage::finish)
    exact = to_closure ? 1 : 0;
    xb = job;
    xbase_fixed = xbuf[job].as<float2>() + xpad - carry;        // (the call's samples lie where they were produced; the tail goes in front)
    const int rc = run(j.n, ov_soft, nullptr, ov_cap, n_out, s, prof);
    xbase_fixed = nullptr;
    exact = k_exact;
    if (rc == XRIT_OK) { hist_xb = job; hist_len = (size_t)j.padN + j.n; hist_job = job; }
    ov_fell_back = true;
    return rc;
}

// the joints, the output and the call's result, on the call's stream behind the walkers
int ClockStage::ov_finalize(int job, float *soft_out, size_t cap, hipStream_t s, Profiler *prof)
{
    OvJob &j = ov[job];
    XR_HIP(hipStreamWaitEvent(s, j.ev_walk, 0));
    if (j.ahead) j.w0_ii = j.padN - (int)carry;        // (known now: the call in front has finished)
    {
        ProfScope ps(prof, "clock_joints", s);
        hipLaunchKernelGGL(clock_reset_kernel, dim3(1), dim3(256), 0, s, counters.as<unsigned>(), CLK_CTL_WORDS, (int *)nullptr);
        hipLaunchKernelGGL(clock_overlap_scan_kernel, dim3(1), dim3(256), 0, s, j.segs.as<OverlapSeg>(), j.G, j.stride,
                           st.as<ClockState>() + cur, j.w0_ii, j.w0_carried ? 1 : 0, par.omega_mid, ov_j0(j), ov_offs(j),
                           st.as<ClockState>() + (cur ^ 1), clock_res(counters), xbuf[job].as<float2>() + xpad - j.padN,
                           tail.as<float2>() + 1024 * (cur ^ 1), j.N, clock_ctl(counters), ov_moments(j), (unsigned long long)cap);
        if (soft_out)
            hipLaunchKernelGGL(clock_overlap_copy_kernel, dim3(div_up((size_t)j.stride, 1024), j.G), dim3(256), 0, s, j.stage.as<float>(),
                               j.segs.as<OverlapSeg>(), ov_j0(j), ov_offs(j), j.stride, soft_out, (unsigned long long)cap, clock_res(counters));
    }
    XR_HIP(hipGetLastError());
    XR_HIP(hipMemcpyAsync(h_res, counters.p, CLK_CTL_WORDS * sizeof(unsigned), hipMemcpyDeviceToHost, s));
    j.state = 3;
    return XRIT_OK;
}

// hand-off passes (no-ops once the device-side test has declared the hand-off closed), the output pass and the
// copy of the control block.  finish() runs after the caller has synchronised the stream.
int ClockStage::begin(size_t n, float *soft_out, float2 *sym_out, size_t cap, hipStream_t s, Profiler *prof)
{
    passes = 0;
    unconverged = 0;
    large_open = 0;
    max_residual = 0;
    prev_carry = carry;
    prev_n = n;
    redo_ok = false;
    ov_fell_back = false;
    // this call's overlap job, if its samples were produced into one (input_slot): the oldest job that waits for its call
    ov_cur = -1;
    if (!xbase_fixed) {
        unsigned long long first = ~0ull;
        for (int q = 0; q < NXB; ++q)
            if ((ov[q].state == 1 || ov[q].state == 2) && ov[q].serial < first) { first = ov[q].serial; ov_cur = q; }
        if (ov_cur >= 0 && (ov[ov_cur].n != n || sym_out != nullptr)) {
            set_error("clock recovery: the call does not match the overlap job its samples were produced for");
            return XRIT_E_INVALID;
        }
    }
    if (ov_cur >= 0) {
        job = Job{};
        job.n = n; job.soft = soft_out; job.cap = cap;
        xb = ov_cur;
        x_pending = -1;
        ov_job = -1;
        in_flight = true;
        om_ext = false; om_scanned = false;
        ov_soft = soft_out; ov_cap = cap;
        carry_before_fallback = carry;
        if (ov[ov_cur].state == 1) XR_TRY(ov_launch(ov_cur, s, false, prof));
        return ov_finalize(ov_cur, soft_out, cap, s, prof);
    }
    job = Job{};
    Job &j = job;
    j.n = n; j.soft = soft_out; j.sym = sym_out; j.cap = cap;
    j.N = (long long)(carry + n);
    j.ni = j.N - XR_MM_NTAPS - XR_MM_FUDGE;
    if (j.N >= (1LL << 31)) { set_error("clock recovery: more than 2^31 samples in one call"); return XRIT_E_INVALID; }
    if (!xbase_fixed) {
        if (x_pending >= 0) xb = x_pending;
        x_pending = -1;
        XR_TRY(xbuf[xb].reserve((size_t)(xpad + n + 64 + 16) * sizeof(float2)));
    }
    in_flight = true;
    float2 *x = xbase();
    if (carry)
        hipLaunchKernelGGL(clock_tail_kernel, dim3(1), dim3(1024), 0, s, tail.as<float2>() + 1024 * cur, x, (int)carry);
    const bool ext = om_ext;          // statistic supplied by the producer of the samples (Costas final pass)
    const bool scanned = ext && om_scanned;       // ... and its count curve too (om_scan)
    om_ext = false;
    om_scanned = false;
    if (j.ni <= 0) {
        // not enough samples for a single symbol: everything is carried to the next call
        j.short_input = true;
        if (j.N > 1024) { set_error("clock recovery: unread tail exceeds the hand-over buffer"); return XRIT_E_INVALID; }
        if (j.N > 0)
            XR_HIP(hipMemcpyAsync(tail.as<float2>() + 1024 * (cur ^ 1), x, (size_t)j.N * sizeof(float2),
                                  hipMemcpyDeviceToDevice, s));
        XR_HIP(hipMemcpyAsync(st.as<ClockState>() + (cur ^ 1), st.as<ClockState>() + cur, sizeof(ClockState),
                              hipMemcpyDeviceToDevice, s));
        return XRIT_OK;
    }
    if (serial) {
        j.K = 1;
        hipLaunchKernelGGL(clock_reset_kernel, dim3(1), dim3(256), 0, s, counters.as<unsigned>(), CLK_CTL_WORDS, (int *)nullptr);
        {
            ProfScope ps(prof, "clock_serial", s);
            hipLaunchKernelGGL(clock_serial_kernel, dim3(1), dim3(64), 0, s, x, table.as<float>(),
                               st.as<ClockState>() + cur, st.as<ClockState>() + (cur ^ 1), clock_res(counters), soft_out,
                               sym_out, (unsigned long long)cap, j.N, j.ni, par, tail.as<float2>() + 1024 * (cur ^ 1));
        }
        XR_HIP(hipGetLastError());
        XR_HIP(hipMemcpyAsync(h_res, counters.p, CLK_CTL_WORDS * sizeof(unsigned), hipMemcpyDeviceToHost, s));
        return XRIT_OK;
    }
    // chain budget: the slowest admissible symbol clock plus slack
    const double min_omega = (double)par.omega_mid - (double)par.omega_lim;
    // sample rings (see ClockTile).  Over SS symbols the read index advances by A at most; a ring of R samples
    // must hold CLK_M below the schedule, CLK_SLACK above it, the advance and the 8 interpolator taps.  SS
    // divides the output tile (16 symbols).
    const double max_adv = (double)par.omega_mid + (double)par.omega_lim + 0.004;
    static const int tries[6][2] = {{32, 4}, {32, 2}, {32, 1}, {64, 4}, {64, 2}, {64, 1}};
    int SS = 1, A = 0, R = 64;
    for (int q = 0; q < 6; ++q) {
        R = tries[q][0];
        SS = tries[q][1];
        A = (int)ceil(SS * max_adv) + 1;
        if (CLK_M + CLK_SLACK + A + XR_MM_NTAPS <= R) break;    // else: very large sps, sub-steps run from global memory
    }
    j.wide = R > 32;
    j.SS = SS; j.W = R; j.A = A;
    j.STEP = (int)floor((double)SS * (double)par.omega_mid * 65536.0);
    j.WS = R + CLK_ROW_EXTRA;
    j.tile_bytes3 = clock_tile_bytes(j.WS, 1, 192);
    // one-wave groups share the table: as many groups per workgroup as gives the CU the most waves (the rings
    // fill LDS; two waves per SIMD is what the registers allow)
    int waves_cu = 0;
    j.NG = 1;
    for (int ng = 1; ng <= 8 && ng <= ng_max; ++ng) {
        const long long need = (long long)clock_tile_bytes(j.WS, ng, 64 * ng);
        if (need > lds_per_cu) break;
        long long wv = (long long)ng * (lds_per_cu / need);
        if (wv > 8) wv = 8;
        if (wv >= waves_cu) { waves_cu = (int)wv; j.NG = ng; }
    }
    if (waves_cu < 1) { set_error("clock recovery: a ring of %d samples per chain does not fit LDS", R); return XRIT_E_INVALID; }
    j.tile_bytes = clock_tile_bytes(j.WS, j.NG, 64 * j.NG);
    if (auto_ns) {
        // One wave = 64 chains, and a wave's time is its chain length times a serial per-symbol latency: a pass
        // costs (generations of resident waves) x NS.  So the chain length is chosen from what the chip holds --
        // LDS decides: CUs x floor(LDS / ring tile) waves -- such that the call's waves fill a whole number g of
        // generations: NS = symbols / (g x resident chains), with the smallest g that keeps NS <= 256 (longer
        // chains lose more in the passes than their fewer hand-offs gain).  C2: 1792 resident waves, g = 1,
        // NS = 112 (the former 64 gave 1.66 generations = 2 x 64 symbol times per pass, and 190 k hand-offs
        // instead of 113 k).
        const long long resident = (long long)cu_count * waves_cu;   // waves
        const double symbols = (double)j.N / min_omega;
        // (calls that do not fill the chip keep 64: 16-symbol chains would make their passes four times shorter,
        // but the fuzz runs found symbol slips and count mismatches with them at low Es/N0; 32-symbol chains are
        // clean and 15-25 % quicker per small call, but at Es/N0 < 3.6 dB 4 % of the cases had a hard decision
        // flipped instead of 2.5 %)
        NS = 64;
        for (int g = 1; g <= 64; ++g) {
            const double chains = (double)(g * resident * 64 - 4);          // K = symbols / NS + 3 must fit
            int ns = (int)ceil(symbols / chains);
            ns = (ns + 15) & ~15;                                            // whole output tiles, 64-byte rows
            if (ns <= 256) { NS = ns < 64 ? 64 : ns; break; }
        }
    }
    const int K = (int)((double)j.N / (min_omega * NS)) + 3;
    j.K = K;
    const int BL = ext ? om_BL : CLK_OM_BLOCK;
    // (the producer's statistic: block b covers buffer samples [carry + b BL, ...), its phasor counted from the first new sample:
    // the count curve is computed in those coordinates -- same symbol instants, see om_scan -- and placed `carry` samples in)
    const double om_off = ext ? (double)carry : 0.0;
    const int nb = ext ? om_nb : (int)((j.N + CLK_OM_BLOCK - 1) / CLK_OM_BLOCK);
    XR_TRY(S.reserve((size_t)K * sizeof(ClockState)));
    XR_TRY(E.reserve((size_t)K * sizeof(ClockState)));
    XR_TRY(J.reserve((size_t)K * sizeof(float4)));
    XR_TRY(flags.reserve((size_t)(4 * K + 4) * sizeof(int)));
    if (pass_writes) XR_TRY(stage.reserve(((size_t)div_up((size_t)K, 64) * 64 * NS + 64) * sizeof(float)));
    XR_TRY(om.reserve((size_t)nb * (sizeof(double2) + sizeof(double))));
    const int nbK = scan_blocks(K), nbB = scan_blocks(nb);
    const int nbmax = nbK > nbB ? nbK : nbB;
    XR_TRY(dlin.reserve((size_t)(K + 1) * sizeof(float2)));
    const int nbN = newton_blocks(K) > nbmax ? newton_blocks(K) : nbmax;
    XR_TRY(work.reserve((size_t)(3 * nbN + 8) * sizeof(AffMap)));
    {
        const size_t nw = div_up((size_t)K, 64);
        XR_TRY(wsolve.reserve(newton_waves_bytes(nw)));
        j.gated = force_gated;
        j.mean_j = jmean_valid && jmean_ns == NS && K >= 256 && !force_gated && !no_meanj;
    }
    // cfg.clock_exact: 1 -- relayed to closure; n > 1 -- n relay passes; 0 -- auto_passes of them, and on to closure from
    // finish() when the walks have not settled by then (calls of fewer than auto_min symbols: hand-off passes, relayed from
    // finish() when they stall high); < 0 -- hand-off passes only
    j.relay = exact >= 1 || (exact == 0 && auto_passes > 0 && (long long)K * NS >= auto_min);
    j.relay_budget = exact > 1 ? exact : (exact == 1 ? 0 : auto_passes);
    // Round 4: the default configuration has NO hand-off passes.  The relay's first pass walks every segment from the timing
    // guess itself: the loop forgets a start that is 2.6e-2 sample off within ~12 time constants (37 k symbols), so with one
    // segment per CU (49.6 k symbols at C2) the end states of that pass are as close to the serial trajectory as those of a
    // walk from a closed hand-off, and the second pass starts from them.  Measured at C2 (steady-state bursts,
    // profiles/r4_handoff_vs_guess.json): guess + 2 passes over 256 segments 5.3e-5 rms from the serial trajectory, where two
    // hand-off passes + 3 relay passes over 766 segments had 6.4e-5 -- for two sweeps of the stream instead of five.
    // (... where the call fills the chip with such segments, or is one segment: in between -- 74 k to 6 M symbols: C5's bursts,
    // calls of 2^19 .. 2^22 samples -- the walkers are few and it is their latency that counts: two hand-off passes, which
    // cost such a call 0.1 ms, start them close enough for three passes over segments of 16 k symbols)
    const long long all_syms = (long long)K * NS;
    j.no_handoff = j.relay && (relay_no_handoff >= 0 ? relay_no_handoff != 0
                                                     : exact == 0 && (all_syms <= one_walk_limit() || all_syms >= auto_guess_min));
    if (j.relay) XR_TRY(relay_plan());
    if (j.no_handoff && exact == 0) {
        // passes for the default's parity by segment length (measured, same file: 16.5 k symbols per segment: 3 passes 9.6e-5,
        // 4 passes 6.2e-5; 24.8 k: ...; 49.6 k: 2 passes 5.3e-5, 3 passes 2.9e-5)
        const long long L = (long long)j.cps * NS;
        j.relay_budget = L >= auto_long_seg ? 2 : (L >= auto_long_seg / 2 ? 3 : 4);      // (shorter segments: cfg.clock_exact_window)
        // cfg.clock_exact = -3, the quick relay: the passes in front of the last are there for their end states and are walked
        // approximately (clock_relay_kernel's apx) -- the first ones in one guess round, the one before the last in two, leaving
        // the record of its guesses; no literal verification, no symbols.  Measured at C2 (steady-state bursts, streamed):
        // 1.92 instead of 2.07 ms per burst, soft symbols 1.15e-4 instead of 5.6e-5 rms from the serial trajectory (which is
        // itself 1.0e-4 from the oracle); only the first pass approximate: 1.95 ms, 7.5e-5.  (Few segments: a call of up to
        // `budget` segments closes exactly within its budget if every pass is exact, and stays that way.)
        // (a plan of two passes -- segments of 49 k symbols and more: C1, C3 -- has one pass in front of its last, and walking that
        // one in two rounds saves nothing: measured 10.2 against 9.4 ms per C3 burst)
        if (relay_quick && j.relay_budget >= 3 && j.G > j.relay_budget && j.relay_w == 0) {
            for (int p = 0; p < 2 && p < j.relay_budget - 1; ++p) j.relay_apx[p] = p == j.relay_budget - 2 ? 2 : 1;
        }
    }
    for (int p = 0; p < 2; ++p) if (relay_apx_cfg[p] >= 0 && j.relay && j.G > 1) j.relay_apx[p] = relay_apx_cfg[p];

    // What the relay passes buy is exact history: after p passes a symbol has between (p - 1) and p segments of exactly
    // walked trajectory in front of it, and the default's three passes are sized for the segments of the big LRIT bursts
    // (16.5 k symbols: 33 k .. 50 k symbols of history).  Where a call's segments are three times that long -- bursts at the
    // circuit rate, 2^28 samples without a decimator: 82 k (LRIT) or 130 k (HRIT) symbols per segment -- two passes leave more
    // history than that (measured at 49 k symbols per segment: 4.4e-5 rms from the serial trajectory against the three
    // passes' 6.4e-5, profiles/r3_late_experiments.txt), and the third pass, a third of the relay's time, is not run; nor is the watch on
    // the segment starts, which compares the starts of the last two passes (here the hand-off's own): the horizon of two
    // such passes already is that of the seven the watch would ask for.  The look at Es/N0 stays (finish()).
    j.relay_long = j.relay && !j.no_handoff && exact == 0 && auto_passes >= 3 && (long long)j.cps * NS >= auto_long_seg;
    if (j.relay_long) j.relay_budget = 2;
    // (measured at C2: a pass that writes costs ~40 us more than one that does not -- 16-byte stores, 64 lines per wave
    // instruction --, a call whose last pass did not write pays the output pass, 175 us.  Writing from one pass earlier
    // than the previous call's last (two writing passes per call) was slower: 2.21 against 2.13 ms per step.)
    j.write_from = pass_writes ? (last_passes - 1 > 3 ? last_passes - 1 : 3) : 0x7fffffff;
    j.dirty = flags.as<int>();
    j.counts = flags.as<int>() + K;
    j.nrun = flags.as<int>() + 2 * K;
    j.terminal = flags.as<int>() + 3 * K;
    j.written = flags.as<int>() + 3 * K + 4;
    double2 *X = om.as<double2>();
    double *cnt = reinterpret_cast<double *>(om.as<char>() + (size_t)nb * sizeof(double2));
    const ClockState *st_in = st.as<ClockState>() + cur;
    if (K <= 1) hipLaunchKernelGGL(clock_reset_kernel, dim3(1), dim3(256), 0, s, counters.as<unsigned>(), (max_passes + 5) * 8, j.terminal);
    if (K > 1) {
        {
            ProfScope ps(prof, "clock_guess", s);
            if (!ext)
                hipLaunchKernelGGL(clock_om_kernel, dim3(div_up((size_t)nb, 4)), dim3(256), 0, s, x, X, j.N, nb,
                                   1.0 / (double)sps);
            if (!scanned) {
                ClkUnwrapF uf{X, cnt, nb, (double)sps, 0.0, BL, 0.0};
                hipLaunchKernelGGL(scan_reduce_kernel<ClkUnwrapF>, dim3(nbB), dim3(SCAN_BLOCK), 0, s, uf, (long long)nb,
                                   work.as<double>());
                hipLaunchKernelGGL(scan_apply_lookback_kernel<ClkUnwrapF>, dim3(nbB), dim3(SCAN_BLOCK), 0, s, uf, (long long)nb,
                                   work.as<double>());
            }
            hipLaunchKernelGGL(clock_guess_kernel, dim3(div_up((size_t)K, 256)), dim3(256), 0, s, cnt, nb, (double)sps,
                               S.as<ClockState>(), st_in, K, NS, par.omega_mid, x, table.as<float>(), j.ni, om_off, BL, j.dirty,
                               clock_ctl(counters), (max_passes + 5) * 8, j.terminal, j.written);
        }
        // (no hand-off passes: the relay's first pass walks from the timing guess itself, see above)
        if (j.no_handoff) j.relay_force = true;
        else XR_TRY(enqueue_passes(batch < max_passes ? batch : max_passes, s, prof));
    } else {
        XR_HIP(hipMemcpyAsync(S.p, st_in, sizeof(ClockState), hipMemcpyDeviceToDevice, s));
    }
    // (exact closure: the first relay pass walks every segment and writes every symbol; there is no output pass)
    if (j.relay && K > 1) {
        // (in front of the relay, not of the hand-off passes: started with those it competes with kernels that fill the
        // chip and the burst takes 2.9 instead of 2.4 ms)
        // (the stream's other work starts when the relay kernels do -- an event recorded here -- but is enqueued BEHIND them: the
        // host takes ~0.2 ms to enqueue a front end and a Costas loop, which the relay kernels need not wait for)
        if (before_relay) XR_TRY(before_relay(0));
        int rc_relay = enqueue_relay(relay_batch, true, s, prof);
        if (rc_relay == XRIT_OK && before_relay) rc_relay = before_relay(1);
        return rc_relay;
    }
    return enqueue_output(s, prof);
}

bool ClockStage::closed() const
{
    if (job.short_input || job.K <= 1 || job.no_handoff) return true;
    return reinterpret_cast<const int *>(h_res)[0] != 0;
}

// After a stream synchronise: continue the passes if the first batch did not close (cold start), commit the
// carried state, report the symbol count.
int ClockStage::finish(size_t *n_out, hipStream_t s, Profiler *prof)
{
    *n_out = 0;
    in_flight = false;      // (the caller has synchronised: whatever finish() still enqueues it waits for itself)
    if (ov_cur >= 0) {
        OvJob &oj = ov[ov_cur];
        const int *hc = reinterpret_cast<const int *>(h_res);
        ClockResult r;
        memcpy(&r, reinterpret_cast<const unsigned *>(h_res) + CLK_RES_WORD, sizeof r);
        float snr2, far_;
        memcpy(&snr2, &hc[14], sizeof snr2);
        memcpy(&far_, &hc[18], sizeof far_);
        snr_estimate = snr2;
        ov_walkers = oj.G;
        ov_joint_max = far_;
        relay_passes = 1;
        relay_closed = false;
        relay_auto = false;
        relay_segments = oj.G;
        relay_seg_chains = 0;
        passes = 0;
        unconverged = 0; max_residual = 0; large_open = 0;     // (the hand-off's figures: this plan has no hand-off)
        if (trace_env) {
            unsigned hs[8];
            XR_HIP(hipMemcpy(hs, oj.aux.p, sizeof hs, hipMemcpyDeviceToHost));
            fprintf(stderr, "[xrit] overlap: %d walkers over ranges of %d samples behind %d samples of history (%s), %u steps, %.2f guess rounds per step; "
                            "joints: %d do not fit, largest distance %.3e sample; 2 Es/N0 = %.1f; %llu symbols%s\n",
                    oj.G, oj.Ls, oj.hist, oj.w0_carried ? "walker 0 from the carried state" : "walker 0 warms up in the burst before", hs[0],
                    hs[0] ? (double)hs[1] / hs[0] : 0.0, hc[17], (double)far_, (double)snr2, (unsigned long long)r.n_symbols, r.ok ? "" : " -- NOT taken");
#ifdef XRIT_RELAY_TIMING
            // (what the walkers that have finished since the last look spent per step, whichever burst they belong to)
            unsigned long long hd[16] = {0}, z[16] = {0};
            XR_HIP(hipMemcpyFromSymbol(hd, HIP_SYMBOL(relay_dbg), sizeof hd));
            XR_HIP(hipMemcpyToSymbol(HIP_SYMBOL(relay_dbg), z, sizeof z));
            const double st = hd[6] ? (double)hd[6] : 1.0;
            fprintf(stderr, "[xrit] overlap walkers, cycles per step: ring wait %.0f, setup %.0f, guess rounds %.0f, literal step + verdict %.0f, "
                            "stage + commit %.0f, loop %.0f; sum %.0f (%llu steps)\n", hd[0] / st, hd[1] / st, hd[2] / st, hd[3] / st, hd[4] / st,
                    hd[5] / st, (hd[0] + hd[1] + hd[2] + hd[3] + hd[4] + hd[5]) / st, hd[6]);
#endif
        }
        if (hc[15]) { set_error("clock recovery: a walker gave up waiting for its sample ring (watchdog)"); oj.state = 0; ov_cur = -1; return XRIT_E_HIP; }
        // The default configuration's two looks at such a call (ClockStage::finish below has them for the relay): Es/N0 below 7 dB
        // -- walked to closure, the serial trajectory whatever the noise --, and a joint whose two trajectories do not meet within a
        // quarter symbol -- the timing guess and the loop disagree about a symbol count.
        const bool signal = snr2 >= auto_snr_floor;
        if ((signal && !(snr2 >= auto_snr)) || hc[17] > 0) return ov_fallback(n_out, s, prof, signal);
        oj.state = 0;
        const int jb = ov_cur;
        ov_cur = -1;
        if (!r.ok) {
            if ((size_t)r.n_symbols > ov_cap) { set_error("clock recovery produced %zu symbols, capacity %zu", (size_t)r.n_symbols, ov_cap); return XRIT_E_CAPACITY; }
            set_error("clock recovery: the walkers did not reach the end of the input");
            return XRIT_E_INVALID;
        }
        cur ^= 1;
        carry = (size_t)(oj.N - r.ii_final);
        if (carry > 1024) { set_error("clock recovery: carry of %zu samples exceeds the hand-over buffer", carry); return XRIT_E_INVALID; }
        last_symbols = (size_t)r.n_symbols;
        *n_out = last_symbols;
        hist_xb = jb; hist_len = (size_t)oj.padN + oj.n; hist_job = jb;
        redo_ok = true;
        return XRIT_OK;
    }
    if (job.short_input) {
        carry = (size_t)job.N;
        last_symbols = 0;
        cur ^= 1;
        return XRIT_OK;
    }
    const int *hctl = reinterpret_cast<const int *>(h_res);
    const bool in_batch = closed();
    if (!in_batch) {
        // (past 48 passes a hand-off that is still open is an acquisition that closes a chain or two per pass: unless
        // the exact closure is switched off the relay takes over -- it walks from whatever start states there are)
        const int give_up = exact != -1 && max_passes > 48 ? 48 : max_passes;
        while (hctl[0] == 0 && job.enqueued < give_up) {
            // a boundary outside the trust region (acquisition, a slip): from here on the gated three-launch solve
            if (hctl[NEWTON_CTL_TAKEOVER]) job.gated = true;
            XR_TRY(enqueue_passes(4, s, prof));
            XR_HIP(hipMemcpyAsync(h_res, counters.p, CLK_CTL_WORDS * sizeof(unsigned), hipMemcpyDeviceToHost, s));
            XR_HIP(hipStreamSynchronize(s));
        }
        // (a hand-off that used up its pass budget without closing -- ctl[0] still 0 -- is relayed all the same: the
        // walkers need start states, not a closed hand-off)
        job.relay_force = hctl[0] == 0;
        if (job.relay && job.K > 1) XR_TRY(enqueue_relay(relay_batch, true, s, prof));
        else XR_TRY(enqueue_output(s, prof, true));
        XR_HIP(hipStreamSynchronize(s));
    }
    relay_passes = 0;
    relay_closed = false;
    relay_auto = false;
    if ((exact == 0 || exact <= -2) && job.K > 1 && !job.relay) {
        // (cfg.clock_exact = 0: a call too short for the relay to be planned at its start; -2: the fast configuration)
        // The hand-off passes normally stall at the recurrence's own floor, an rms residual of ~1e-4 sample (Es/N0 12 dB).
        // At low Es/N0 they stall at 5e-4 .. 1e-3 instead -- every wrong decision kicks mu by 2e-3 -- and which
        // near-zero symbols then fall on the other side differs from the serial loop (a third
        // of the 2..6 dB fuzz cases flip 1..5 decisions).  Such a call is closed exactly: the relay reproduces the
        // serial trajectory whatever the noise.
        float q;
        memcpy(&q, &hctl[4], sizeof q);
        const int open_ = hctl[6];
        // (... and a hand-off that never closed at all -- the pass budget ran out on an acquisition -- has every reason
        // to be walked: the relay needs start states, not a closed hand-off)
        if (hctl[0] == 0 || (open_ > 0 && q > auto_rms * auto_rms * (float)open_)) {
            job.relay = relay_auto = true;
            job.relay_force = hctl[0] == 0;
            job.relay_budget = 0;
            XR_TRY(relay_plan());
            XR_TRY(enqueue_relay(relay_batch, true, s, prof));
            XR_HIP(hipStreamSynchronize(s));
        }
    }
    if (job.relay && job.K > 1) {
        if (exact == 0 && !relay_auto && hctl[11] == 0) {
            // The default: after its three passes the segment starts of a clean signal move by a few 1e-4 sample rms from
            // pass to pass.  Where they still move by more -- low Es/N0: every decision that differs kicks mu by 2e-3 --,
            // or the hand-off passes never closed, the call is walked to closure: the serial trajectory whatever the noise.
            // Not to closure at once: four more passes at a time until the starts have settled too (Es/N0 6 dB: after six
            // passes, 4.4 ms per 2^28-sample burst; at 3 dB they never do before the relay closes, 40 passes).
            float shift_sq;
            memcpy(&shift_sq, &hctl[12], sizeof shift_sq);
            float snr2;            // 2 Es/N0 as the first pass's soft symbols show it (12 dB: 32, 7 dB: 10)
            memcpy(&snr2, &hctl[14], sizeof snr2);
            snr_estimate = snr2;
            // Below auto_snr the call is walked to closure: the serial trajectory whatever the noise (in a 2..6 dB fuzz
            // slice of 60 calls up to 350 k symbols the passes-until-settled rule alone left one hard decision different
            // from the serial trajectory's -- a call of 14 segments whose starts happened to move by 3e-4 -- and the
            // hand-off passes of rounds 2-3 ten).
            // (... unless there is no signal to speak of: |noise| alone shows 2 / (pi - 2) = 1.75, BPSK at Es/N0 0 dB 2.4; a
            // loop that is not locked has no trajectory to close on, the relay would run its G + 1 passes for nothing)
            const bool signal = snr2 >= auto_snr_floor;
            if (signal && ((job.relay_force && !job.no_handoff) || !(snr2 >= auto_snr))) {
                relay_auto = true;
                job.relay_budget = 0;
            }
            if (!signal || relay_auto || job.relay_long || job.no_handoff) shift_sq = 0.0f;
            // Without hand-off passes nothing has settled symbol slips between the timing guess and the loop: a start that moved
            // by the better part of a symbol between the first pass (the guess) and a later one (the end state of the segment
            // in front) means the guess counted a symbol more or less than the loop there, and every segment behind it is one
            // symbol off until the exact front reaches it -- the call is walked to closure.
            if (job.no_handoff && signal && !relay_auto) {
                float mv;
                memcpy(&mv, &hctl[16], sizeof mv);
                if (!(mv < 0.25f * sps)) {
                    relay_auto = true;
                    job.relay_budget = 0;
                }
            }
            while (!relay_auto && !(shift_sq <= auto_shift * auto_shift) && hctl[11] == 0 && job.relay_enq < job.G + 1) {
                relay_auto = true;
                job.relay_budget = job.relay_enq + 4;
                XR_TRY(enqueue_relay(4, false, s, prof));
                XR_HIP(hipStreamSynchronize(s));
                memcpy(&shift_sq, &hctl[12], sizeof shift_sq);
            }
        }
        // the relay goes on until a pass changes nothing (or the pass budget of a partial closure is used up)
        while (hctl[11] == 0 && job.relay_enq < relay_limit()) {
            // (a closure nobody asked for -- cfg.clock_exact = 0 / -2 -- is not pursued on an input without a signal: an
            // unlocked loop has no trajectory to close on and would take all G + 1 passes)
            float snr2;
            memcpy(&snr2, &hctl[14], sizeof snr2);
            if (relay_auto && exact != 1 && !(snr2 >= auto_snr_floor)) break;
            XR_TRY(enqueue_relay(relay_batch < 32 ? 32 : relay_batch, false, s, prof));
            XR_HIP(hipStreamSynchronize(s));
        }
        relay_passes = hctl[10];
        relay_closed = hctl[11] != 0;
        if (trace_env && exact == 0) fprintf(stderr, "[xrit] relay: the first pass's soft symbols show 2 Es/N0 = %.1f (to closure below %.1f)\n", (double)snr_estimate, (double)auto_snr);
        if (trace_env) {
            std::vector<unsigned> hc((size_t)relay_passes * RELAY_STAT);
            const unsigned *changed = reinterpret_cast<const unsigned *>(relay.as<RelaySeg>() + 3 * (size_t)job.G);
            XR_HIP(hipMemcpy(hc.data(), changed, hc.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
            for (int p = 0; p < relay_passes; ++p) {
                const unsigned *c = &hc[(size_t)RELAY_STAT * p];
                float mv;
                memcpy(&mv, &c[3], sizeof mv);
                const unsigned long long sq = (unsigned long long)c[4] | ((unsigned long long)c[5] << 32);
                fprintf(stderr, "[xrit] relay pass %d: segments walked %u of %d, steps %u, rounds %u (%.2f per step); starts moved by %.3e sample rms, %.3e at most%s\n", p,
                        c[0], job.G, c[1], c[2], c[1] ? (double)c[2] / c[1] : 0.0,
                        c[6] ? sqrt((double)sq / 1099511627776.0 / (double)c[6]) : 0.0,
                        c[3] >= 0x80000000u ? 0.0 : (double)mv, c[3] >= 0x80000000u ? " (watchdog mark)" : "");
            }
        }
#ifdef XRIT_RELAY_TIMING
        if (trace_env) {
            unsigned long long hd[16] = {0}, z[16] = {0};
            XR_HIP(hipMemcpyFromSymbol(hd, HIP_SYMBOL(relay_dbg), sizeof hd));
            XR_HIP(hipMemcpyToSymbol(HIP_SYMBOL(relay_dbg), z, sizeof z));
            std::vector<unsigned> wd((size_t)RELAY_WDBG_PASSES * RELAY_WDBG_SEGS * RELAY_WDBG_WORDS);
            XR_HIP(hipMemcpyFromSymbol(wd.data(), HIP_SYMBOL(relay_wdbg), wd.size() * sizeof(unsigned)));
            for (int p = 0; p < relay_passes && p < RELAY_WDBG_PASSES && job.relay_w == 0; ++p)
                for (int g = 0; g < job.G && g < RELAY_WDBG_SEGS; ++g) {
                    const unsigned *w = &wd[((size_t)p * RELAY_WDBG_SEGS + g) * RELAY_WDBG_WORDS];
                    fprintf(stderr, "[xrit] walker %d %d %u %u %u %u %u %u\n", p, g, w[0], w[1], w[2], w[3], w[4], w[5]);
                }
            if (job.relay_w > 0) {
                const double st = hd[7] ? (double)hd[7] : 1.0, un = hd[9] ? (double)hd[9] : 1.0;
                (void)un;
                fprintf(stderr, "[xrit] relay team (wave 1) cycles per step: rings %.0f, setup %.0f, interpolate..scan %.0f, meeting + sums %.0f, "
                                "positions..verdict %.0f, meeting + verdicts %.0f, commit + loop %.0f (%llu steps)\n",
                        hd[0] / st, hd[1] / st, hd[2] / st, hd[3] / st, hd[4] / st, hd[5] / st, hd[6] / st, hd[7]);
            }
            for (int o = 0; o < 16 && job.relay_w == 0; o += 8) {
                const double st = hd[o + 6] ? (double)hd[o + 6] : 1.0;
                fprintf(stderr, "[xrit] relay walker cycles per step, passes %s: wait %.0f, setup %.0f, rounds %.0f, verify %.0f, commit %.0f, loop head %.0f (%llu steps)\n",
                        o ? ">= 8" : "< 8", hd[o] / st, hd[o + 1] / st, hd[o + 2] / st, hd[o + 3] / st, hd[o + 4] / st, hd[o + 5] / st, hd[o + 6]);
            }
        }
#endif
        const int want = relay_passes + relay_passes / 4 + 8;
        relay_batch = want < 32 ? 32 : (want > 4096 ? 4096 : want);
    }
    passes = job.K > 1 ? hctl[1] : 0;
    if (job.K > 1) {
        // one spare pass beyond what this call needed.  Unlike the Costas loop's, it is never dropped: the stall
        // rule ends the passes one earlier or later from burst to burst (measured: one C2 burst in ten needs six
        // instead of five), and a call that runs out of enqueued passes costs a host round trip, two more passes
        // and a second output pass -- 0.5 ms against the 19 us the four idle launches take.
        last_passes = passes;
        const int want = passes + 1;
        // (small calls run on while boundaries still freeze: 10..20 passes; a call that hands over to the relay after two
        // passes does not need five launches enqueued -- the idle ones are 10 us each)
        const int least = job.relay && passes <= 2 ? 3 : 5;
        batch = want < least ? least : (want > 32 ? 32 : want);
    }
    if (job.K > 1) {
        const bool clean = in_batch && hctl[NEWTON_CTL_TAKEOVER] == 0 && hctl[9] == 0 && passes <= 12;
        if (job.mean_j && !clean) jmean_valid = false;           // the stream has changed: measure again
        else if (!job.mean_j && !job.gated && clean && job.K >= 1024 && jac_passes > 0 && !job.no_handoff) {
            hipLaunchKernelGGL(clock_jmean_kernel, dim3(1), dim3(256), 0, s, J.as<float4>(), job.K, jmean.as<float4>());
            jmean_valid = true;
            jmean_ns = NS;
        }
    }
    unconverged = job.K > 1 ? (unsigned)hctl[2] : 0;
    large_open = job.K > 1 ? (unsigned)hctl[9] : 0;
    memcpy(&max_residual, &hctl[3], sizeof(float));
    if (trace_env && job.K > 1) {
        std::vector<unsigned> hc((size_t)passes * 8);
        XR_HIP(hipMemcpy(hc.data(), clock_cnt(counters, 0), hc.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
        for (int p = 0; p < passes; ++p) {
            float mr;
            unsigned long long qf;
            memcpy(&mr, &hc[(size_t)p * 8 + 2], 4);
            memcpy(&qf, &hc[(size_t)p * 8 + 4], 8);
            const float q = (float)((double)qf / 1099511627776.0);
            fprintf(stderr, "[xrit] clock pass %d: K=%d changed=%u open=%u max_r=%.3e large=%u rms_r=%.3e\n", p, job.K,
                    hc[(size_t)p * 8], hc[(size_t)p * 8 + 1], mr, hc[(size_t)p * 8 + 3],
                    hc[(size_t)p * 8 + 1] ? sqrtf(q / hc[(size_t)p * 8 + 1]) : 0.f);
        }
    }
    ClockResult r;
    memcpy(&r, reinterpret_cast<const unsigned *>(h_res) + CLK_RES_WORD, sizeof r);
    cur ^= 1;
    if (!r.ok && serial) {
        set_error("clock recovery produced more than the %zu symbols the output holds", job.cap);
        return XRIT_E_CAPACITY;
    }
    if (!r.ok && job.relay && hctl[15]) {
        set_error("clock recovery: a relay walker gave up waiting for its sample ring or its team (watchdog)");
        return XRIT_E_HIP;
    }
    if (!r.ok) {
        set_error("clock recovery: chain budget exhausted before the end of the input");
        return XRIT_E_INVALID;
    }
    carry = (size_t)(job.N - r.ii_final);
    if (carry > 1024) {
        set_error("clock recovery: carry of %zu samples exceeds the hand-over buffer", carry);
        return XRIT_E_INVALID;
    }
    last_symbols = (size_t)r.n_symbols;
    *n_out = last_symbols;
    if (last_symbols > job.cap) {
        set_error("clock recovery produced %zu symbols, capacity %zu", last_symbols, job.cap);
        return XRIT_E_CAPACITY;
    }
    redo_ok = true;
    // (what this call left in its buffer is history for an overlap call behind it only together with a timing curve: none here)
    hist_xb = xbase_fixed ? hist_xb : xb; hist_len = xbase_fixed ? hist_len : prev_carry + prev_n; hist_job = xbase_fixed ? hist_job : -1;
    return XRIT_OK;
}

__global__ void __launch_bounds__(256) clock_negate_kernel(float2 *__restrict__ x, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { float2 v = x[i]; x[i] = make_float2(-v.x, -v.y); }
}

// the carried state under the other sign of the stream: the two symbols of history change sign, their slicer
// decisions (1 where a component is positive) follow; timing (mu, omega) is what it was
__global__ void clock_flip_state_kernel(ClockState *st)
{
    ClockState s = st[0];
    s.p0 = cf32{-s.p0.x, -s.p0.y};
    s.p1 = cf32{-s.p1.x, -s.p1.y};
    s.c0 = cf32{s.p0.x > 0.f ? 1.f : 0.f, s.p0.y > 0.f ? 1.f : 0.f};
    s.c1 = cf32{s.p1.x > 0.f ? 1.f : 0.f, s.p1.y > 0.f ? 1.f : 0.f};
    st[0] = s;
}

int ClockStage::run(size_t n, float *soft_out, float2 *sym_out, size_t cap, size_t *n_out, hipStream_t s,
                    Profiler *prof)
{
    XR_TRY(begin(n, soft_out, sym_out, cap, s, prof));
    XR_HIP(hipStreamSynchronize(s));
    return finish(n_out, s, prof);
}

static constexpr size_t CLK_ALT_SLOT = sizeof(ClockState) + 1024 * sizeof(float2);

int ClockStage::redo_flipped(float *soft_out, float2 *sym_out, size_t cap, size_t *n_out, hipStream_t s, Profiler *prof)
{
    *n_out = 0;
    if (!redo_ok) { set_error("clock recovery: no finished call to run again"); return XRIT_E_INVALID; }
    // finish() moved on to the other state / tail slot: back to the one the call started from (untouched since)
    cur ^= 1;
    float2 *data = xdata();     // where the call's input lies
    const size_t n = prev_n;
    if (alt_valid) {
        // the flipped loop's own state (make_alt) in place of this one's
        XR_HIP(hipMemcpyAsync(st.as<ClockState>() + cur, alt.p, sizeof(ClockState), hipMemcpyDeviceToDevice, s));
        XR_HIP(hipMemcpyAsync(tail.as<float2>() + 1024 * cur, alt.as<char>() + sizeof(ClockState), 1024 * sizeof(float2),
                              hipMemcpyDeviceToDevice, s));
        carry = alt_carry;
        alt_valid = false;
    } else {
        carry = prev_carry;
        hipLaunchKernelGGL(clock_flip_state_kernel, dim3(1), dim3(1), 0, s, st.as<ClockState>() + cur);
        if (carry) hipLaunchKernelGGL(clock_negate_kernel, dim3(div_up(carry, 256)), dim3(256), 0, s, tail.as<float2>() + 1024 * cur, carry);
    }
    if (n) hipLaunchKernelGGL(clock_negate_kernel, dim3(div_up(n, 256)), dim3(256), 0, s, data, n);
    // (the samples kept as the NEXT call's history are negated now: walkers of an overlap job must not warm up over them)
    hist_xb = -1; hist_len = 0; hist_job = -1;
    XR_HIP(hipGetLastError());
    xbase_fixed = data - carry;             // (carry <= 1024 <= xpad: never in front of the buffer)
    const int rc = run(n, soft_out, sym_out, cap, n_out, s, prof);
    xbase_fixed = nullptr;
    return rc;
}

__global__ void __launch_bounds__(1024) clock_carry_pack_kernel(unsigned char *__restrict__ rec, const ClockState *__restrict__ st,
                                                                const float2 *__restrict__ tail, int carry)
{
    if (threadIdx.x == 0) {
        unsigned *h = reinterpret_cast<unsigned *>(rec);
        h[0] = 1u; h[1] = (unsigned)carry; h[2] = 0u; h[3] = 0u;
        *reinterpret_cast<ClockState *>(rec + 16) = *st;
    }
    float2 *t = reinterpret_cast<float2 *>(rec + ClockStage::CARRY_HEAD);
    t[threadIdx.x] = (int)threadIdx.x < carry ? tail[threadIdx.x] : make_float2(0.f, 0.f);
}

__global__ void __launch_bounds__(1024) clock_carry_unpack_kernel(const unsigned char *__restrict__ rec, ClockState *__restrict__ st,
                                                                  float2 *__restrict__ tail, int carry)
{
    if (threadIdx.x == 0) *st = *reinterpret_cast<const ClockState *>(rec + 16);
    const float2 *t = reinterpret_cast<const float2 *>(rec + ClockStage::CARRY_HEAD);
    if ((int)threadIdx.x < carry) tail[threadIdx.x] = t[threadIdx.x];
}

static_assert(16 + sizeof(ClockState) <= ClockStage::CARRY_HEAD, "the carried record's head holds a ClockState");

int ClockStage::export_carry(void *d_rec, int which, hipStream_t s)
{
    if (in_flight) { set_error("clock recovery: a call is in flight"); return XRIT_E_INVALID; }
    if (which != 0 && !redo_ok) { set_error("clock recovery: no finished call whose start state is still held"); return XRIT_E_INVALID; }
    const int slot = which == 0 ? cur : cur ^ 1;
    const size_t c = which == 0 ? carry : prev_carry;
    hipLaunchKernelGGL(clock_carry_pack_kernel, dim3(1), dim3(1024), 0, s, static_cast<unsigned char *>(d_rec), st.as<ClockState>() + slot,
                       tail.as<float2>() + 1024 * slot, (int)c);
    XR_HIP(hipGetLastError());
    return XRIT_OK;
}

bool ClockStage::last_walk_exact() const
{
    if (!redo_ok || ov_cur >= 0 || job.short_input) return false;
    return serial || (job.relay && (relay_closed || job.G == 1));
}

int ClockStage::redo_from(const void *d_rec, size_t carry_rec, float *soft_out, float2 *sym_out, size_t cap, size_t *n_out, hipStream_t s, Profiler *prof)
{
    *n_out = 0;
    if (!redo_ok) { set_error("clock recovery: no finished call to run again"); return XRIT_E_INVALID; }
    if (carry_rec > 1024) { set_error("clock recovery: a carried record with %zu unread samples", carry_rec); return XRIT_E_INVALID; }
    // finish() moved on to the other state / tail slot: back to the one the call started from, which takes the record
    cur ^= 1;
    float2 *data = xdata();     // where the call's input lies
    const size_t n = prev_n;
    hipLaunchKernelGGL(clock_carry_unpack_kernel, dim3(1), dim3(1024), 0, s, static_cast<const unsigned char *>(d_rec), st.as<ClockState>() + cur,
                       tail.as<float2>() + 1024 * cur, (int)carry_rec);
    XR_HIP(hipGetLastError());
    carry = carry_rec;
    alt_valid = false;
    hist_xb = -1; hist_len = 0; hist_job = -1;      // (what lies in front of the input now is another handle's tail: no history for walkers)
    xbase_fixed = data - carry;             // (carry <= 1024 <= xpad: never in front of the buffer)
    const int rc = run(n, soft_out, sym_out, cap, n_out, s, prof);
    xbase_fixed = nullptr;
    return rc;
}

int ClockStage::make_alt(hipStream_t s, Profiler *prof)
{
    alt_valid = false;
    if (!redo_ok) { set_error("clock recovery: no finished call to run again"); return XRIT_E_INVALID; }
    XR_TRY(alt.reserve(2 * CLK_ALT_SLOT));
    // this sign's outcome aside ...
    char *keep = alt.as<char>() + CLK_ALT_SLOT;
    const size_t carry_keep = carry;
    XR_HIP(hipMemcpyAsync(keep, st.as<ClockState>() + cur, sizeof(ClockState), hipMemcpyDeviceToDevice, s));
    XR_HIP(hipMemcpyAsync(keep + sizeof(ClockState), tail.as<float2>() + 1024 * cur, 1024 * sizeof(float2), hipMemcpyDeviceToDevice, s));
    // ... the call again on its negated input (symbols dropped) ...
    size_t k = 0;
    XR_TRY(redo_flipped(nullptr, nullptr, (size_t)1 << 40, &k, s, prof));
    XR_HIP(hipMemcpyAsync(alt.p, st.as<ClockState>() + cur, sizeof(ClockState), hipMemcpyDeviceToDevice, s));
    XR_HIP(hipMemcpyAsync(alt.as<char>() + sizeof(ClockState), tail.as<float2>() + 1024 * cur, 1024 * sizeof(float2), hipMemcpyDeviceToDevice, s));
    alt_carry = carry;
    // ... and this sign's state back
    XR_HIP(hipMemcpyAsync(st.as<ClockState>() + cur, keep, sizeof(ClockState), hipMemcpyDeviceToDevice, s));
    XR_HIP(hipMemcpyAsync(tail.as<float2>() + 1024 * cur, keep + sizeof(ClockState), 1024 * sizeof(float2), hipMemcpyDeviceToDevice, s));
    carry = carry_keep;
    alt_valid = true;
    redo_ok = false;
    return XRIT_OK;
}

}  // namespace xrit
