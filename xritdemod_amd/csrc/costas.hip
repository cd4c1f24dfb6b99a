// costas.hip -- 2nd-order BPSK Costas loop over time-tiled chains.
// Replaces SatHelper::CostasLoop::Work (/root/reference/demodulator/src/
// demodulator.cpp:152; object built at :448 with loop bandwidth CLOCK_ALPHA,
// :220).  The loop is a serial recurrence in (phase, freq).  Here the call's
// samples are cut into chains of L samples, one lane per chain with its state in
// registers; chains run concurrently from guessed start states and the guesses
// are corrected by a Newton step on the multiple-shooting system
//        S[k+1] = G_k(S[k]),  k = 0..K-2,   S[0] = state carried from the last call
// using the tangent dG_k/dS each lane propagates next to its trajectory.  The
// linearised system is a prefix scan of 2x2 affine maps.  Passes repeat until no
// start state moves by more than (tol_phase, tol_freq); chain 0 always starts
// from the true state, so the converged prefix grows by at least one chain per
// pass whatever the guesses were.  The loop is pi-periodic in phase (the
// detector is Re*Im): a residual of m*pi at a boundary is carried as a parity
// that shifts every later start by pi instead of being "corrected".
#include "kernels.h"

#include <cstdlib>
#include "scan.h"
#include "newton.h"

namespace xrit {

constexpr int COSTAS_HEAD = 64;   // chains covered by the sequential coarse model
constexpr int COSTAS_CTL_WORDS = 16;   // control words in front of the per-pass counter slots
constexpr int COSTAS_CTL_MODEL = 9;    // control word: this call's guesses went through the model step

// ---------------------------------------------------------------- statistics
// sub[r] = sum z^2 over the run of COSTAS_SUB samples [8 r, 8 r + 8) (squaring removes the BPSK modulation).  In the
// chain the matched filter's epilogue leaves it (fir.hip); a stand-alone stage computes it here.
constexpr int COSTAS_SUB = 8;

__global__ void __launch_bounds__(256) costas_sub_kernel(const float2 *__restrict__ z, float2 *__restrict__ sub,
                                                         long long n, long long runs)
{
    // eight lanes per run, one sample each (a wave reads 512 consecutive bytes), three row shifts add the run up
    const long long j = (long long)blockIdx.x * 256 + threadIdx.x;
    float sr = 0.f, si = 0.f;
    if (j < n) {
        const float2 v = z[j];
        sr = v.x * v.x - v.y * v.y;
        si = 2.0f * v.x * v.y;
    }
    sr += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sr), 0x111, 0xf, 0xf, true));      // row_shr:1
    si += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(si), 0x111, 0xf, 0xf, true));
    sr += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sr), 0x112, 0xf, 0xf, true));      // row_shr:2
    si += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(si), 0x112, 0xf, 0xf, true));
    sr += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sr), 0x114, 0xf, 0xf, true));      // row_shr:4
    si += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(si), 0x114, 0xf, 0xf, true));
    if ((threadIdx.x & 7) == 7 && (j >> 3) < runs) sub[j >> 3] = make_float2(sr, si);
}

// stat[k] = sum z^2 over chain k, from the runs.  RPC32 (chains of 256 samples = 32 runs): a lane takes two runs per load
// (16 bytes; a wave reads 1 KB in one piece), a row of 16 lanes holds a chain, four row shifts add it up in a fixed order;
// four loads in flight per lane.  Otherwise a thread adds one chain's runs in order.  Deterministic either way.
template <bool RPC32>
__global__ void __launch_bounds__(256) costas_stat_kernel(const float2 *__restrict__ sub, float2 *__restrict__ stat,
                                                          long long runs, int rpc, int K)
{
    if (RPC32) {
        // wave w of the launch: chains [16 w, 16 w + 16), load j covers chains 16 w + 4 j .. + 3
        const long long w = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
        const int lane = threadIdx.x & 63;
        float4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long long q = (w * 4 + j) * 64 + lane;                    // pair of runs 2 q, 2 q + 1
            v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (2 * q + 1 < runs) v[j] = reinterpret_cast<const float4 *>(sub)[q];
            else if (2 * q < runs) { const float2 a = sub[2 * q]; v[j].x = a.x; v[j].y = a.y; }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float sr = v[j].x + v[j].z, si = v[j].y + v[j].w;
#define XR_ROW_ADD(CTRL)                                                                                          \
            sr += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sr), CTRL, 0xf, 0xf, true));       \
            si += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(si), CTRL, 0xf, 0xf, true));
            XR_ROW_ADD(0x111) XR_ROW_ADD(0x112) XR_ROW_ADD(0x114) XR_ROW_ADD(0x118)      // row_shr:1, 2, 4, 8
#undef XR_ROW_ADD
            const long long k = w * 16 + j * 4 + (lane >> 4);
            if ((lane & 15) == 15 && k < K) stat[k] = make_float2(sr, si);
        }
        return;
    }
    const long long k = (long long)blockIdx.x * 256 + threadIdx.x;
    if (k >= K) return;
    const long long r0 = k * rpc;
    const int cnt = (int)min((long long)rpc, runs - r0);
    float sr = 0.f, si = 0.f;
    for (int i = 0; i < cnt; ++i) {
        const float2 v = sub[r0 + i];
        sr += v.x; si += v.y;
    }
    stat[k] = make_float2(sr, si);
}

// -------------------------------------------------------------------- guess
__device__ __forceinline__ double wrap_pi_d(double x)
{
    return x - 2.0 * XR_PI_D * rint(x / (2.0 * XR_PI_D));
}

// prefix sum of wrapped differences of 2*theta -> unwrapped 2*theta per chain
struct UnwrapF {
    typedef double T;
    const float2 *stat;
    double *th2;        // out: unwrapped 2*theta at chain centres
    __device__ T identity() const { return 0.0; }
    __device__ T combine(const T &lo, const T &hi) const { return lo + hi; }
    __device__ double ang(long long k) const { const float2 a = stat[k]; return (double)atan2f(a.y, a.x); }
    // (a thread's run takes every angle once -- single precision: 1e-7 rad on a guess that is 2e-2 off --, the differences chain
    // through `prev`; two double-precision atan2 per element in each of the scan's two kernels cost 37 us of latency per burst)
    __device__ T reduce_run(long long i0, int cnt) const
    {
        double s = 0, prev = i0 > 0 ? ang(i0 - 1) : 0.0;
        for (int k = 0; k < cnt; ++k) {
            const double cur = ang(i0 + k);
            s += (i0 + k == 0) ? cur : wrap_pi_d(cur - prev);
            prev = cur;
        }
        return s;
    }
    __device__ void apply_run(long long i0, int cnt, const T &pre) const
    {
        double s = pre, prev = i0 > 0 ? ang(i0 - 1) : 0.0;
        for (int k = 0; k < cnt; ++k) {
            const double cur = ang(i0 + k);
            s += (i0 + k == 0) ? cur : wrap_pi_d(cur - prev);
            prev = cur;
            th2[i0 + k] = s;
        }
    }
};

__global__ void costas_guess_kernel(const double *__restrict__ th2, float2 *__restrict__ S, int K, int L,
                                    int *__restrict__ dirty, int *__restrict__ ctl, int ctl_words,
                                    const float2 *__restrict__ state)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x == 0) {
        // the call's control block and per-pass counters start from zero (no launch of its own for that)
        for (int i = threadIdx.x; i < ctl_words; i += blockDim.x) ctl[i] = 0;
        __syncthreads();
    }
    if (k >= K) return;
    dirty[k] = 1;                       // every chain runs in the first pass
    if (k == 0) {                       // chain 0 starts from the carried state; first solve: gated
        S[0] = state[0];
        ctl[5] = 1;
        return;
    }
    // boundary k lies between chain centres k-1 and k
    double thb = 0.25 * (th2[k - 1] + th2[k]);
    int a = max(0, k - 2), b = min(K - 1, k + 1);
    double f = (b > a) ? 0.5 * (th2[b] - th2[a]) / ((double)(b - a) * L) : 0.0;
    S[k] = make_float2((float)wrap_pi_d(thb), (float)f);
}

// Chain-level model of the loop over the first chains of the call, where the
// true trajectory may still be acquiring: averaged detector = Im(e^{-2j phi} c_k)/2.
__global__ void costas_head_kernel(const float2 *__restrict__ stat, float2 *__restrict__ S,
                                   const float2 *__restrict__ state, int K, int L, CostasGains g, int head)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    float2 s0 = state[0];
    S[0] = s0;
    double phi = s0.x, fr = s0.y;
    int H = min(head, K - 1);
    for (int k = 0; k < H; ++k) {
        double pm = phi + fr * L * 0.5;
        double sn, cs;
        sincos(-2.0 * pm, &sn, &cs);
        float2 c = stat[k];
        double eb = 0.5 * (cs * c.y + sn * c.x);
        phi = phi + L * fr + (g.alpha + g.beta * L * 0.5) * eb;
        fr = fr + g.beta * eb;
        S[k + 1] = make_float2((float)wrap_pi_d(phi), (float)fr);
    }
}

// ------------------------------------------------------------ hand-off solve
// Policy for newton.h: the loop is pi-periodic in phase, so a residual of m*pi is
// carried as a parity (aux) that shifts every later start by pi.
struct CostasPolicy {
    float2 *S;
    const float2 *E;
    const float4 *J;
    int *dirty;
    unsigned *cnt;           // [0] changed, [1] not frozen, [2] max |r_phase| bits
    float trust_p, trust_f, tol_p, tol_f;
    float accept, gate;      // stop test: accept when max residual <= accept; trust gate needed while > gate
    const int *expansive;    // ctl[7]: some chain of this call saw the loop expansive
    int model = 0;           // this solve follows the pass over the sub-block model (costas_model_pass_kernel), not a pass over the samples
    float model_accept = 0.f;   // first pass over the samples behind a model step: accept on prediction when no residual exceeds this
    // In lock an error in a chain's start decays along the chain and the hand-off tolerance (1e-5 rad) is what is
    // left of it in the output.  While the loop pulls in it is expansive for stretches (d phase / d start up to 2
    // per chain over several chains in a row, more through a cycle slip): what one call leaves within tolerance
    // the next one multiplies -- fuzz: cold start at -516 Hz cut in three short calls, first call 4e-6 off the
    // serial loop, second 6e-4, third with two hard decisions flipped.  Such calls hand off three times tighter
    // (first call 3e-7, second 4e-5: the floor the same amplification puts under the float32 differences of the
    // two implementations); a locked stream never sees it.
    __device__ float scale() const { return *expansive ? 0.3f : 1.0f; }

    struct Elem { float2 e, s; float4 j; };
    __device__ Elem fetch(long long k) const { return Elem{E[k], S[k + 1], J[k]}; }
    __device__ bool active(const Elem &) const { return true; }
    __device__ void residual(const Elem &el, float &r1, float &r2, int &aux) const
    {
        float rp = el.e.x - el.s.x;
        float m = rintf(rp * (float)(1.0 / XR_PI_D));
        r1 = rp - m * (float)XR_PI_D;
        r2 = el.e.y - el.s.y;
        aux = ((int)m) & 1;
    }
    __device__ float4 jac(const Elem &el) const { return el.j; }
    __device__ bool outside_trust(float d1, float d2) const
    {
        return !(fabsf(d1) <= trust_p) || !(fabsf(d2) <= trust_f);
    }
    // The loop's lock points are pi apart with an unstable equilibrium half way.  A residual beyond ~pi/8 means the
    // next chain was started nearer to that than the tangent is good for (there it is expansive, and a Newton
    // step through it lands on either side: boundaries then swap sides for ever -- seen on cold-started short
    // calls, 1.5 rad residuals after 32 passes).  Such a boundary gets the plain hand-off; the correction that
    // reaches it from upstream is dropped, and the region closes chain by chain.
    __device__ bool distrust(float r1, float) const { return !(fabsf(r1) <= 0.4f); }
    __device__ void update(long long k, const Elem &el, float j1, float j2, float n1, float n2, int aux_prefix,
                           int aux_k, float r1, NewtonStat &st) const
    {
        const int par = aux_prefix & 1;
        const float amp = 1.0f / scale();
        const bool frozen = fabsf(n1) * amp <= tol_p && fabsf(n2) * amp <= tol_f && par == 0 && (aux_k & 1) == 0;
        if (frozen) return;
        const float2 ek = el.e;
        float2 nw = make_float2(ek.x + (par ? (float)XR_PI_D : 0.f) + j1, ek.y + j2);
        const float2 old = el.s;
        st.open_ += 1;
        // what the stop test looks at: the hand-off residual r1 AND the whole Newton update n1 = r1 + (what reaches
        // this boundary from upstream).  n1 estimates how far the start the last pass ran from was off; where the
        // loop is not contractive (pull-in, near a cycle slip) small residuals add up along the chains, and a stop
        // test on r1 alone closed calls whose output was 6e-4 away from the serial loop's (fuzz: cold start at
        // -516 Hz in three short calls, two hard-decision flips behind it).
        st.max_r = fmaxf(st.max_r, amp * fmaxf(fabsf(r1), fabsf(n1)));
        st.sum_sq += newton_fix(r1 * r1);
        if (nw.x != old.x || nw.y != old.y) {
            S[k + 1] = nw;
            dirty[k + 1] = 1;
            st.changed += 1;
        }
    }
    // After every solve: ctl[0] done, ctl[1] passes run, ctl[2] boundaries still open, ctl[3] max residual (bits),
    // ctl[4] max residual of the previous pass (bits), ctl[6] accepted on prediction: verify after the final pass.
    // The residuals of the pass just run measure the starts it ran from; the Newton update this solve applied
    // leaves ~C r^2 with C = r / r_prev^2 seen between the last two passes (C ~ 0.15 at C2: 6e-2 -> 7e-4 -> would
    // be 1e-7).  Below ~1e-5 the residuals stop falling anyway: that is float32 rounding along a 256-sample chain,
    // which no start state removes.  So when the predicted residual is well inside the acceptance, the next pass
    // would only confirm it: the final pass runs from the updated starts right away, and costas_verify_kernel
    // checks the residuals it leaves (against twice the acceptance: they ARE the rounding floor); if one is
    // outside, the call goes on iterating.
    __device__ void decide(int *ctl) const
    {
        const unsigned changed = newton_cnt_load(cnt + 0), open_ = newton_cnt_load(cnt + 1), mr = newton_cnt_load(cnt + 2);
        if (model) {
            // the step on the model refines the guesses; it is no pass of the loop and decides nothing
            ctl[COSTAS_CTL_MODEL] = 1;
            ctl[5] = __uint_as_float(mr) > gate ? 1 : 0;
            return;
        }
        ctl[1] += 1;
        ctl[2] = (int)open_;
        const float max_r = __uint_as_float(mr);
        ctl[3] = __float_as_int(max_r * scale());       // reported in radians
        const float r_prev = ctl[1] >= 2 ? __int_as_float(ctl[4]) : 0.0f;
        ctl[4] = (int)mr;
        ctl[6] = 0;
        // nothing moved, or what is still open sits within a factor two of the tolerance: accept
        // (max_r comes in units of the scaled tolerance: update() multiplies by 1 / scale())
        if (changed == 0 || max_r <= accept) { ctl[0] = 1; ctl[2] = 0; }
        else if (r_prev > 0.0f && max_r < 0.25f * r_prev && 4.0f * max_r * (max_r / r_prev) * (max_r / r_prev) <= accept) {
            ctl[0] = 1;
            ctl[6] = 1;
        }
        // The first pass behind a model step has no earlier pass to take C from.  Its residuals measure the model's error
        // (4e-4 rad rms, 2..3e-3 at worst over a burst's 2e5 chains at C2); the update this solve applied leaves C r^2
        // with C ~ 0.15: 1e-6.  Up to model_accept the final pass runs right away -- and is verified like any
        // hand-off accepted on prediction.
        // Only on a tracking loop: the call before closed in the minimum number of passes (model_accept is 0 otherwise) and no
        // chain of this pass saw the loop expansive.
        else if (ctl[COSTAS_CTL_MODEL] && ctl[1] == 1 && !*expansive && max_r <= model_accept) {
            ctl[0] = 1;
            ctl[6] = 1;
        }
        ctl[5] = max_r > gate ? 1 : 0;     // residuals this small cannot leave the trust region: skip the gate scan
    }
};

// --------------------------------------------------------------------- pass
// One wave = 64 chains, one lane per chain.  Samples move through LDS tiles of
// COSTAS_CT samples per chain so that HBM sees coalesced 16-byte accesses
// (4 lanes cover one chain's 64-byte row, 16 chains per wave instruction) while
// each lane walks its own chain: row stride COSTAS_CT+1 float2 = 18 dwords keeps the
// per-lane ds_read_b64 conflict free.  FINAL: write the de-rotated samples (through
// the same kind of tile), no tangent.
#ifndef XR_AMP_THR
#define XR_AMP_THR 1.02f
#endif
constexpr int COSTAS_CT = 8;

template <bool FINAL>
__global__ void __launch_bounds__(64) costas_pass_kernel(const float2 *__restrict__ z, float2 *__restrict__ y,
                                                         const float2 *__restrict__ S, float2 *__restrict__ E,
                                                         float4 *__restrict__ J, int *__restrict__ dirty,
                                                         float2 *__restrict__ state_out, long long n, int L, int K,
                                                         CostasGains g, double2 *__restrict__ om, long long om_off,
                                                         double inv_sps, float rot_c, float rot_s,
                                                         int *__restrict__ ctl, CostasPolicy pol,
                                                         AffMap *__restrict__ aggs, int warm)
{
    // (warm, cfg.front_exact: the FINAL pass starts every chain `warm` chains early, from the start state there, and walks those
    // samples quietly -- what a hand-off leaves a chain's start beside its predecessor's end, a few 1e-6 rad, the loop forgets at
    // a half per ~600 samples --; every other pass: 0)
    // the hand-off already closed (later passes of the batch are no-ops), or the gated solve has taken over
    if (!FINAL && (ctl[0] || (aggs != nullptr && ctl[NEWTON_CTL_TAKEOVER]))) return;
    // Two tiles of COSTAS_CT samples per chain at a time (round 4): the pair's samples sit in two LDS areas, a lane walks its
    // chain through both, and FINAL leaves the de-rotated samples IN PLACE -- the pair then holds 16 outputs per chain, one whole
    // 128-byte line, written by eight lanes per row.  No output tile of its own: 9 KB of LDS per wave instead of 17.5, so that
    // all of a burst's 3277 waves are resident at once (13 per CU; nine were, in two generations), and the NEXT pair's samples
    // are in flight in registers while this one is walked (two tiles ahead instead of one).
    __shared__ float2 tin[2][64][COSTAS_CT + 1];
    const int lane = threadIdx.x;
    const int kbase = blockIdx.x * 64;
    const int k = kbase + lane;
    bool mine = k < K;
    if (!FINAL && mine) mine = dirty[k] != 0;
    const bool any_mine = __any(mine);
    if (!any_mine && (FINAL || aggs == nullptr)) return;
    float2 e_own = make_float2(0.f, 0.f);
    float4 j_own = make_float4(0.f, 0.f, 0.f, 0.f);
    if (any_mine) {
    const long long base = (long long)k * L;
    int cnt = 0;
    float phase = 0.f, freq = 0.f;
    if (mine) {
        cnt = (int)min((long long)L, n - base);
        float2 s = S[FINAL ? (k > warm ? k - warm : 0) : k];
        phase = costas_prewrap(s.x);
        freq = s.y;
    }
    CostasTan t{1.f, 0.f, 0.f, 1.f};
    float amp = 1.f;       // largest |d phase(t) / d phase(0)| seen at the tile ends of this chain
    // FINAL with om != nullptr: the timing-line statistic of the clock-recovery guess, sum |y|^2 e^{-j 2 pi m / sps}
    // over this chain (m = index in the clock-recovery input buffer), so that stage needs no sweep of its own.
    // The phasor advances by a fixed rotation per sample, restarted per chain from a double-precision phase.
    float om_c = 1.f, om_s = 0.f, om_r = 0.f, om_i = 0.f;
    if (FINAL && om != nullptr && mine) {
        double ph = (double)(om_off + base) * inv_sps;
        ph -= floor(ph);
        float sn, cs;
        loop_sincos(-6.28318530717958647692f * (float)ph, sn, cs);
        om_c = cs;
        om_s = sn;
    }
    const int nt = L / COSTAS_CT;
    constexpr int LPR = COSTAS_CT / 2;            // lanes per chain row (16 bytes each)
    constexpr int RPI = 64 / LPR;                 // rows per wave instruction
    constexpr int NIT = 64 / RPI;
    const int lrow = lane / LPR, lcol = (lane % LPR) * 2;
    float4 pre[2][NIT];
    // unconditional, clamped loads (chains past K / samples past n re-read the last pair; their results are
    // never used; the input buffer has 8 samples of slack): no branches between the loads, so a pair's requests are all in flight together
    const long long last_pair = (n - 1) & ~1LL;     // an odd n reads one sample of the buffer's slack
    auto fetch = [&](int tile, int h) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = min(it * RPI + lrow, K - 1 - kbase);
            long long j = (long long)(kbase + c) * L + (long long)tile * COSTAS_CT + lcol;
            j = j < last_pair ? j : last_pair;
            j = j < 0 ? 0 : j;                  // (warm-up tiles in front of the call's first sample: never walked)
            pre[h][it] = *reinterpret_cast<const float4 *>(z + j);
        }
    };
    auto stash = [&](int r, int h) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = it * RPI + lrow;
            tin[h][c][lcol] = make_float2(pre[r][it].x, pre[r][it].y);
            tin[h][c][lcol + 1] = make_float2(pre[r][it].z, pre[r][it].w);
        }
    };
    // one tile of this lane's chain from area h
    auto walk = [&](int tile, int h) {
        const int i0 = tile * COSTAS_CT;
        if (FINAL && tile < 0) {
            // warm-up (wave-uniform): no output, no statistic; a chain that would start in front of the call's first sample
            // (the first `warm` chains) waits for it with the call's start state
            const bool act = mine && (long long)k * L + i0 >= 0;
#pragma unroll
            for (int i = 0; i < COSTAS_CT; ++i) {
                float yr, yi;
                const float2 v = tin[h][lane][i];
                float ph = phase, fr = freq;
                costas_step<false>(v.x, v.y, ph, fr, g, yr, yi, t);
                if (act) { phase = ph; freq = fr; }
            }
            return;
        }
        if (__all(!mine || i0 + COSTAS_CT <= cnt)) {
            // every chain of the wave has the whole tile: no per-sample guard (idle lanes compute on zeros)
#pragma unroll
            for (int i = 0; i < COSTAS_CT; ++i) {
                float yr, yi;
                float2 v = tin[h][lane][i];
                costas_step<!FINAL>(v.x, v.y, phase, freq, g, yr, yi, t);
                if (FINAL) {
                    tin[h][lane][i] = make_float2(yr, yi);
                    const float p = yr * yr + yi * yi;
                    om_r = fmaf(p, om_c, om_r);
                    om_i = fmaf(p, om_s, om_i);
                    const float nc = om_c * rot_c - om_s * rot_s;
                    om_s = om_c * rot_s + om_s * rot_c;
                    om_c = nc;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < COSTAS_CT; ++i) {
                float yr = 0.f, yi = 0.f;
                if (i0 + i < cnt) {
                    float2 v = tin[h][lane][i];
                    costas_step<!FINAL>(v.x, v.y, phase, freq, g, yr, yi, t);
                    if (FINAL) {
                        const float p = yr * yr + yi * yi;
                        om_r = fmaf(p, om_c, om_r);
                        om_i = fmaf(p, om_s, om_i);
                        const float nc = om_c * rot_c - om_s * rot_s;
                        om_s = om_c * rot_s + om_s * rot_c;
                        om_c = nc;
                    }
                }
                if (FINAL) tin[h][lane][i] = make_float2(yr, yi);
            }
        }
        if (!FINAL) amp = fmaxf(amp, fabsf(t.pp));
    };
    if (!FINAL) {
        // (the passes with the tangent are bound by their arithmetic: one tile at a time, the next one in flight, measured
        // 0.113 against 0.123 ms per pass for the pairs)
        fetch(0, 0);
        stash(0, 0);
        __syncthreads();
        for (int tile = 0; tile < nt; ++tile) {
            const int cur = tile & 1;
            if (tile + 1 < nt) fetch(tile + 1, 0);
            walk(tile, cur);
            if (tile + 1 < nt) stash(0, cur ^ 1);
            __syncthreads();
        }
    } else {
    const int tb = -warm * nt;                 // (warm > 0 only with an even number of tiles per chain: no pair straddles tile 0)
    fetch(tb, 0);
    if (tb + 1 < nt) fetch(tb + 1, 1);
    stash(0, 0);
    if (tb + 1 < nt) stash(1, 1);
    __syncthreads();
    for (int tile = tb; tile < nt; tile += 2) {
        const bool two = tile + 1 < nt;
        if (tile + 2 < nt) fetch(tile + 2, 0);
        if (tile + 3 < nt) fetch(tile + 3, 1);
        walk(tile, 0);
        if (two) walk(tile + 1, 1);
        if (FINAL && tile >= 0) {
            // 16 samples per chain = one full 128-byte line, 8 lanes per row (the pair's first tile in area 0, its second in area 1)
            __syncthreads();
            const int ncol = two ? 2 * COSTAS_CT : COSTAS_CT;
            const int ob = tile * COSTAS_CT;
            const int orow = lane >> 3, ocol = (lane & 7) * 2;
            const int oh = ocol >= COSTAS_CT ? 1 : 0, oc = ocol - oh * COSTAS_CT;
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int c = it * 8 + orow;
                const long long j = (long long)(kbase + c) * L + (long long)ob + ocol;
                if (kbase + c < K && ocol < ncol) {
                    float2 a = tin[oh][c][oc], b = tin[oh][c][oc + 1];
                    if (j + 1 < n) *reinterpret_cast<float4 *>(y + j) = make_float4(a.x, a.y, b.x, b.y);
                    else if (j < n) y[j] = a;
                }
            }
        }
        __syncthreads();
        if (tile + 2 < nt) stash(0, 0);
        if (tile + 3 < nt) stash(1, 1);
        __syncthreads();
    }
    }
    if (mine) {
    if (FINAL) {
        if (k == K - 1) state_out[0] = make_float2(phase, freq);
        if (om != nullptr) om[k] = make_double2((double)om_r, (double)om_i);
        E[k] = make_float2(phase, freq);        // costas_verify_kernel checks the hand-off that was actually used
    } else {
        e_own = make_float2(phase, freq);
        j_own = make_float4(t.pp, t.pf, t.fp, t.ff);
        E[k] = e_own;
        J[k] = j_own;
        // a chain in which the loop is expansive for a while (pull-in, a cycle slip: never in lock, where amp is
        // exactly 1) puts the whole call on the tight tolerances, see CostasPolicy::scale
        if (amp > XR_AMP_THR && ctl[7] == 0) ctl[7] = 1;
        dirty[k] = 0;
    }
    }
    }   // any_mine
    if (FINAL || aggs == nullptr) return;
    // wave-aligned hand-off solve (newton.h): compose this wave's 64 boundary maps; the last workgroup scans
    AffMap e = aff_identity();
    if (k < K - 1) {
        CostasPolicy::Elem el;
        el.s = S[k + 1];
        if (mine) { el.e = e_own; el.j = j_own; }
        else { el.e = E[k]; el.j = J[k]; }
        e = newton_element(pol, el, false);
    }
    newton_wave_aggregate(e, (int)blockIdx.x, aggs);
}


// ------------------------------------------------------------------ model step
// The guesses above are block averages (1/2 arg sum z^2 over 256 samples): 1.4e-2..2.2e-2 rad rms away from the loop's
// phase, which answers the noise with its own second-order response -- the first Newton step over the samples then
// lands at 3e-5 rad and a second pass has to follow.  The loop's response is cheap to model: over a run of R = 8
// samples the phase moves little, so the run's summed detector is 1/2 Im(sub[r] e^{-2j phi_mid}) and
//      f += beta e,   phi += R f + (alpha + beta (R + 1) / 2) e
// (the R per-sample updates with e spread evenly).  That recurrence, run serially over the whole stream, stays within
// 4e-4 rad rms of the loop (scripts/costas_model.py; the error halves with R), on an eighth of a sweep.  Here it is
// run the way the loop itself is: one lane per chain from the guessed start, with its tangent, and ONE Newton step
// (the same solve) corrects all starts -- they come out at the model's floor, 35 x closer than the block averages, the
// first pass over the samples lands far inside the hand-off tolerance, and the final pass follows it directly
// (CostasPolicy::decide, verified by costas_verify_kernel as every hand-off accepted on prediction is).
__global__ void __launch_bounds__(64) costas_model_pass_kernel(const float2 *__restrict__ sub, const float2 *__restrict__ S,
                                                               float2 *__restrict__ E, float4 *__restrict__ J, int rpc, int K,
                                                               long long runs, CostasGains g, CostasPolicy pol,
                                                               AffMap *__restrict__ aggs)
{
    // the wave's 64 chains x rpc runs are one contiguous piece of sub[]: loaded 16 bytes per lane (1 KB per wave
    // instruction) into rows of rpc + 1 pairs, so that HBM sees whole lines while each lane walks its own row
    extern __shared__ float2 rows[];
    const int lane = threadIdx.x;
    const int kbase = blockIdx.x * 64;
    const int k = kbase + lane;
    const int stride = rpc + 1;
    {
        const long long r0 = (long long)kbase * rpc;               // first run of the wave (even: rpc is)
        const int pairs = 32 * rpc;                                 // 64 chains x rpc / 2
        const float4 *q = reinterpret_cast<const float4 *>(sub + r0);
        const long long last = (runs - 1 - r0) >> 1;                // last pair that lies inside sub[] (buffer: 2 runs of slack)
        for (int i = lane; i < pairs; i += 64) {
            const float4 v = q[i <= last ? i : (last > 0 ? last : 0)];
            const int r = 2 * i, row = r / rpc, col = r - row * rpc;
            rows[row * stride + col] = make_float2(v.x, v.y);
            rows[row * stride + col + 1] = make_float2(v.z, v.w);
        }
    }
    __syncthreads();
    constexpr float R = (float)COSTAS_SUB, H = 0.5f * (COSTAS_SUB - 1);
    const float ka = g.alpha + g.beta * (0.5f * (COSTAS_SUB + 1));
    AffMap e = aff_identity();
    if (k < K - 1) {        // whole chains only: the last one hands nothing over
        const float2 s = S[k];
        float p = s.x, f = s.y;
        CostasTan t{1.f, 0.f, 0.f, 1.f};
        const float2 *q = rows + lane * stride;
        for (int b = 0; b < rpc; ++b) {
            const float2 c = q[b];
            const float pm = p + H * f;
            float sn, cs;
            loop_sincos(-2.0f * pm, sn, cs);
            const float wr = c.x * cs - c.y * sn, wi = c.x * sn + c.y * cs;
            const float err = 0.5f * wi, ed = -wr;          // d err / d phi_mid
            const float dp = ed * (t.pp + H * t.fp), df = ed * (t.pf + H * t.ff);
            const float npp = t.pp + R * t.fp + ka * dp, npf = t.pf + R * t.ff + ka * df;
            t.fp += g.beta * dp;
            t.ff += g.beta * df;
            t.pp = npp;
            t.pf = npf;
            p = p + R * f + ka * err;
            f = f + g.beta * err;
        }
        CostasPolicy::Elem el;
        el.e = make_float2(p, f);
        el.j = make_float4(t.pp, t.pf, t.fp, t.ff);
        el.s = S[k + 1];
        E[k] = el.e;
        J[k] = el.j;
        e = newton_element(pol, el, false);
    }
    newton_wave_aggregate(e, (int)blockIdx.x, aggs);
}

// After the final pass of a hand-off that was accepted on prediction (ctl[6]): the residuals its starts leave
// must be inside the acceptance, else the call is not closed (ctl[0] = 0) and the host goes on iterating.
__global__ void __launch_bounds__(256) costas_verify_kernel(CostasPolicy p, long long n, int *ctl)
{
    if (!ctl[6]) return;
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    bool bad = false;
    if (k < n) {
        float r1, r2;
        int aux;
        p.residual(p.fetch(k), r1, r2, aux);
        bad = !(fabsf(r1) <= 2.0f * p.accept * p.scale()) || (aux & 1);
    }
    const unsigned long long m = __ballot(bad);
    if ((threadIdx.x & 63) == 0 && m) {
        atomicExch(&ctl[0], 0);
        atomicAdd(&ctl[2], (int)__popcll(m));
    }
}

int CostasStage::init(float loop_bw, int chain_len, int max_passes_)
{
    if (const char *e = getenv("XRIT_CX_HIST")) { const int v = atoi(e); if (v >= 0) ex_hist = v; }
    {
        int cus = 0, dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
            ex_walkers = 8 * cus;
        if (const char *e = getenv("XRIT_CX_WALKERS")) { const int v = atoi(e); if (v > 0) ex_walkers = v; }
    }
    if (const char *e = getenv("XRIT_CX_MODE")) ex_mode = atoi(e) & 7;        // (bit 2: first guesses without the phase-offset correction)
    if (const char *e = getenv("XRIT_CX_PRIO")) ex_prio = atoi(e) != 0;
    gains = costas_gains(loop_bw);
    L = chain_len > 0 ? chain_len : 256;
    L = (L + COSTAS_CT - 1) / COSTAS_CT * COSTAS_CT;   // whole LDS tiles per chain
    // a cold start far outside the lock-in range pulls in over many chains with cycle slips, and such a region closes
    // at a chain or two per pass: 32 passes left hard-decision errors on captures with > 1 kHz offset (fuzz), 192 do not
    max_passes = max_passes_ > 0 ? max_passes_ : 192;
    XR_TRY(state.reserve(2 * sizeof(float2)));
    XR_HIP(hipMemset(state.p, 0, 2 * sizeof(float2)));
    XR_TRY(counters.reserve((size_t)(max_passes + 6) * 8 * sizeof(unsigned)));
    XR_HIP(hipHostMalloc((void **)&h_counters, COSTAS_CTL_WORDS * sizeof(unsigned)));
    force_gated = getenv("XRIT_GATED_SOLVE") != nullptr;
#ifdef XRIT_EXPERIMENTS
    keep_spare = getenv("XRIT_KEEP_SPARE") != nullptr;
    // (scripts/r4_floor_vs_frontend.py: the hand-off's stop rule tightened or loosened by a factor)
    if (const char *e = getenv("XRIT_COSTAS_MODEL")) { model_step = atoi(e) != 0; }      // A/B: 0 = block-average guesses only
    if (const char *e = getenv("XRIT_COSTAS_MODEL_ACCEPT")) { model_accept = (float)atof(e); }
    if (const char *e = getenv("XRIT_COSTAS_FINAL_WARM")) { const int v = atoi(e); if (v >= 0 && v <= 16) final_warm = v; }
    if (const char *e = getenv("XRIT_COSTAS_TOL")) { const float k = (float)atof(e); if (k > 0) { tol_phase *= k; tol_freq *= k; } }
#endif
    trace_env = getenv("XRIT_TRACE") != nullptr;
    no_serial_walk = getenv("XRIT_NO_SERIAL_WALK") != nullptr;
    cur = 0;
    return XRIT_OK;
}

int CostasStage::reset(hipStream_t s)
{
    XR_HIP(hipMemsetAsync(state.p, 0, 2 * sizeof(float2), s));
    cur = 0;
    passes = 0;
    unconverged = 0;
    stable = 0;
    last_passes = -1;
    batch = 4;              // (as on a new handle)
    return XRIT_OK;
}

void CostasStage::release()
{
    xj.release(); xbs.release(); xcnt.release();
    if (h_xcnt) { (void)hipHostFree(h_xcnt); h_xcnt = nullptr; }
    state.release(); S.release(); E.release(); J.release(); stat.release(); sub.release(); dlin.release();
    work.release(); flags.release(); counters.release(); wsolve.release(); rescue.release();
    if (h_counters) (void)hipHostFree(h_counters);
    h_counters = nullptr;
}

int CostasStage::get_state(float *phase, float *freq, hipStream_t s)
{
    float2 h;
    XR_HIP(hipMemcpyAsync(&h, state.as<float2>() + cur, sizeof h, hipMemcpyDeviceToHost, s));
    XR_HIP(hipStreamSynchronize(s));
    *phase = h.x;
    *freq = h.y;
    return XRIT_OK;
}

__global__ void costas_flip_phase_kernel(float2 *st)
{
    float ph = st[0].x;
    ph = ph > 0.f ? ph - 3.14159265358979323846f : ph + 3.14159265358979323846f;
    st[0].x = ph;
}

int CostasStage::flip_phase(hipStream_t s)
{
    hipLaunchKernelGGL(costas_flip_phase_kernel, dim3(1), dim3(1), 0, s, state.as<float2>() + cur);
    XR_HIP(hipGetLastError());
    return XRIT_OK;
}

// control block: counters[0..16) = ctl words, per-pass counter slots after it

static inline int *costas_ctl(const DevBuf &b) { return b.as<int>(); }
// ---- serial rescue -----------------------------------------------------------------------------------------------
// While the loop pulls in through cycle slips the hand-off closes a chain or two per pass: a cold start at the
// edge of the lock-in range and at low Es/N0 can slip for a thousand chains (fuzz: LRIT, +417 Hz, 6.5 dB, 960
// chains -- 136 boundaries still open after 192 passes, 137 hard decisions off).  Such a region is cheaper to walk
// than to iterate: ONE wave runs the recurrence through it sample by sample (every lane the same arithmetic on
// v_readlane'd samples, ~0.1 us per sample) and leaves the exact start state of every chain on the way; the
// passes then only have to confirm the closure and carry a parity change to the chains behind the region.
__global__ void costas_open_range_kernel(const float2 *__restrict__ E, const float2 *__restrict__ S, int K,
                                         float tol_p, float tol_f, int *__restrict__ range)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;       // boundary between chains k and k + 1
    if (k >= K - 1) return;
    const float2 e = E[k], s = S[k + 1];
    const float rp = e.x - s.x;
    const float m = rintf(rp * (float)(1.0 / XR_PI_D));
    const float r1 = rp - m * (float)XR_PI_D, r2 = e.y - s.y;
    const bool open = !(fabsf(r1) <= tol_p) || !(fabsf(r2) <= tol_f) || (((int)m) & 1);
    if (open) {
        atomicMin(&range[0], k);
        atomicMax(&range[1], k);
    }
}

__global__ void __launch_bounds__(64) costas_serial_states_kernel(const float2 *__restrict__ z, float2 *__restrict__ S,
                                                                  int *__restrict__ dirty, long long n, int L, int K,
                                                                  int k_first, int k_last, CostasGains g)
{
    const int lane = threadIdx.x;
    const float2 st = S[k_first];
    float phase = costas_prewrap(st.x), freq = st.y;
    CostasTan tan_unused{1.f, 0.f, 0.f, 1.f};
    for (int k = k_first; k <= k_last; ++k) {
        const long long base = (long long)k * L;
        const int len = (int)min((long long)L, n - base);
        for (int i0 = 0; i0 < len; i0 += 64) {
            const long long j = base + i0 + lane;
            const float2 v = j < n ? z[j] : make_float2(0.f, 0.f);
            const int cnt = min(64, len - i0);
            for (int q = 0; q < cnt; ++q) {                                     // wave-uniform
                const float zr = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.x), q));
                const float zi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.y), q));
                float yr, yi;
                costas_step<false>(zr, zi, phase, freq, g, yr, yi, tan_unused);
            }
        }
        if (k + 1 < K && lane == 0) {
            S[k + 1] = make_float2(phase, freq);
            dirty[k + 1] = 1;
        }
    }
}

static inline unsigned *costas_cnt(const DevBuf &b, int pass) { return b.as<unsigned>() + COSTAS_CTL_WORDS + (size_t)pass * 8; }

int CostasStage::enqueue_passes(int count, hipStream_t s, Profiler *prof)
{
    const long long nel = job.K - 1;
    CostasPolicy pol{S.as<float2>(), E.as<float2>(), J.as<float4>(), flags.as<int>(), nullptr, trust, trust / 256.0f,
                     tol_phase, tol_freq, 2.0f * tol_phase, 0.02f * trust, costas_ctl(counters) + 7, 0, job.model_accept};
    const unsigned gridK = div_up((size_t)job.K, 64);
    // wave-aligned solve (newton.h) unless this call has gone over to the gated three-launch one
    const bool wave = !job.gated;
    const int nw = (int)gridK;
    AffMap *aggs = wave ? wsolve.as<AffMap>() : nullptr;
    NewtonStat *wslots = reinterpret_cast<NewtonStat *>(wsolve.as<AffMap>() + nw + 1);
    for (int q = 0; q < count && job.enqueued < max_passes; ++q, ++job.enqueued) {
        pol.cnt = costas_cnt(counters, job.enqueued);
        {
            ProfScope ps(prof, "costas_pass", s);
            hipLaunchKernelGGL(costas_pass_kernel<false>, dim3(gridK), dim3(64), 0, s, job.in, job.out, S.as<float2>(),
                               E.as<float2>(), J.as<float4>(), flags.as<int>(), (float2 *)nullptr, (long long)job.n, L,
                               job.K, gains, (double2 *)nullptr, 0LL, 0.0, 1.f, 0.f, costas_ctl(counters), pol, aggs, 0);
        }
        {
            ProfScope ps(prof, "costas_solve", s);
            if (wave) {
                newton_apply_waves(pol, nel, aggs, costas_ctl(counters), wslots, s);
            } else if (newton_solve(pol, nel, work.as<AffMap>(), dlin.as<float2>(), costas_ctl(counters), s) != 0) {
                set_error("Costas hand-off: %d chains exceed the solver's block budget", job.K);
                return XRIT_E_INVALID;
            }
        }
    }
    return XRIT_OK;
}

int CostasStage::enqueue_final(hipStream_t s, Profiler *prof)
{
    {
    ProfScope ps(prof, "costas_final", s);
    const double dth = -2.0 * XR_PI_D * job.inv_sps;
    float2 *st_out = state.as<float2>() + (cur ^ 1);
    hipLaunchKernelGGL(costas_pass_kernel<true>, dim3(div_up((size_t)job.K, 64)), dim3(64), 0, s, job.in, job.out,
                       S.as<float2>(), E.as<float2>(), J.as<float4>(), flags.as<int>(), st_out, (long long)job.n, L,
                       job.K, gains, job.om, job.om_off, job.inv_sps, (float)cos(dth), (float)sin(dth),
                       costas_ctl(counters), CostasPolicy{}, (AffMap *)nullptr, (L / COSTAS_CT) % 2 == 0 ? final_warm : 0);
    if (job.K > 1) {
        CostasPolicy pol{S.as<float2>(), E.as<float2>(), J.as<float4>(), flags.as<int>(), nullptr, trust, trust / 256.0f,
                         tol_phase, tol_freq, 2.0f * tol_phase, 0.02f * trust, costas_ctl(counters) + 7, 0, job.model_accept};
        hipLaunchKernelGGL(costas_verify_kernel, dim3(div_up((size_t)job.K - 1, 256)), dim3(256), 0, s, pol,
                           (long long)job.K - 1, costas_ctl(counters));
    }
    XR_HIP(hipMemcpyAsync(h_counters, counters.p, COSTAS_CTL_WORDS * sizeof(unsigned), hipMemcpyDeviceToHost, s));
    XR_HIP(hipGetLastError());
    }
    if (job.exact) XR_TRY(enqueue_exact(s, prof));      // (behind the bracket of the final pass: the profiler's scopes do not nest)
    return XRIT_OK;
}

// Everything of one call is put on the stream without waiting: guess, a batch of hand-off passes (each one a
// no-op once the device-side test has declared the hand-off closed), the final pass and the copy of the control
// block.  finish() is called after the caller has synchronised the stream.
int CostasStage::begin(const float2 *in, float2 *out, size_t n, hipStream_t s, Profiler *prof,
                       const float2 *sub_ext, double2 *om, long long om_off, double inv_sps, bool use_exact)
{
    const bool locked = passes > 0 && passes <= 3 && unconverged == 0;      // how the previous call went
    passes = 0;
    unconverged = 0;
    max_residual = 0;
    job = Job{};
    walked = false;
    ex_args_valid = false;
    job.in = in; job.out = out; job.n = n; job.om = om; job.om_off = om_off; job.inv_sps = inv_sps;
    job.exact = exact || use_exact;
    job.model_accept = locked ? model_accept : 0.f;
    if (n == 0) return XRIT_OK;
    const int K = (int)((n + (size_t)L - 1) / (size_t)L);
    job.K = K;
    const float2 *st_in = state.as<float2>() + cur;
    XR_TRY(S.reserve((size_t)K * sizeof(float2)));
    XR_TRY(E.reserve((size_t)K * sizeof(float2)));
    XR_TRY(J.reserve((size_t)K * sizeof(float4)));
    XR_TRY(stat.reserve((size_t)(K + 1) * sizeof(float2)));
    XR_TRY(flags.reserve((size_t)K * sizeof(int)));
    const int nbK = scan_blocks(K);
    const size_t agg_bytes = (((size_t)(3 * newton_blocks(K) + 6) * sizeof(AffMap)) + 15) & ~(size_t)15;
    XR_TRY(dlin.reserve((size_t)(K + 1) * sizeof(float2)));
    XR_TRY(work.reserve(agg_bytes + (size_t)K * sizeof(double)));
    {
        const size_t nw = div_up((size_t)K, 64);
        XR_TRY(wsolve.reserve(newton_waves_bytes(nw)));
        job.gated = force_gated;
    }
    double *th2 = reinterpret_cast<double *>(work.as<char>() + agg_bytes);
    const int ctl_words = (max_passes + 4) * 8;
    if (K <= 1) XR_HIP(hipMemsetAsync(counters.p, 0, (size_t)ctl_words * sizeof(unsigned), s));
    if (K > 1) {
        const int rpc = L / COSTAS_SUB;
        const long long runs = (long long)((n + COSTAS_SUB - 1) / COSTAS_SUB);
        const float2 *sb = sub_ext;       // the runs' statistic left by the producer (matched filter's epilogue) or computed here
        {
            ProfScope ps(prof, "costas_guess", s);
            if (!sb) {
                XR_TRY(sub.reserve((size_t)(runs + 2) * sizeof(float2)));
                hipLaunchKernelGGL(costas_sub_kernel, dim3(div_up((size_t)runs * COSTAS_SUB, 256)), dim3(256), 0, s, in,
                                   sub.as<float2>(), (long long)n, runs);
                sb = sub.as<float2>();
            }
            if (rpc == 32)
                hipLaunchKernelGGL(costas_stat_kernel<true>, dim3(div_up((size_t)K, 64)), dim3(256), 0, s, sb, stat.as<float2>(), runs, rpc, K);
            else
                hipLaunchKernelGGL(costas_stat_kernel<false>, dim3(div_up((size_t)K, 256)), dim3(256), 0, s, sb, stat.as<float2>(), runs, rpc, K);
            const float2 *st = stat.as<float2>();
            UnwrapF uf{st, th2};
            hipLaunchKernelGGL(scan_reduce_kernel<UnwrapF>, dim3(nbK), dim3(SCAN_BLOCK), 0, s, uf, (long long)K,
                               work.as<double>());
            hipLaunchKernelGGL(scan_apply_lookback_kernel<UnwrapF>, dim3(nbK), dim3(SCAN_BLOCK), 0, s, uf, (long long)K,
                               work.as<double>());
            hipLaunchKernelGGL(costas_guess_kernel, dim3(div_up((size_t)K, 256)), dim3(256), 0, s, th2, S.as<float2>(), K, L,
                               flags.as<int>(), costas_ctl(counters), ctl_words, st_in);
            // the sequential head model (64 dependent steps, ~20 us) is for calls that start unlocked; a call that
            // follows one which closed in the minimum number of passes starts on a tracking loop: plain guesses do
            if (!locked)
                hipLaunchKernelGGL(costas_head_kernel, dim3(1), dim3(1), 0, s, st, S.as<float2>(), st_in, K, L, gains,
                                   COSTAS_HEAD);
        }
        // (64 rows of rpc + 1 pairs: within the 48 KB of dynamic LDS a launch gets without asking)
        if (model_step && !job.gated && K > 2 && (rpc & 1) == 0 && rpc <= 92) {
            // one Newton step on the sub-block model of the loop (costas_model_pass_kernel)
            ProfScope ps(prof, "costas_model", s);
            CostasPolicy pol{S.as<float2>(), E.as<float2>(), J.as<float4>(), flags.as<int>(), costas_cnt(counters, max_passes),
                             trust, trust / 256.0f, tol_phase, tol_freq, 2.0f * tol_phase, 0.02f * trust,
                             costas_ctl(counters) + 7, 1, job.model_accept};
            const int nw = (int)div_up((size_t)K, 64);
            AffMap *aggs = wsolve.as<AffMap>();
            NewtonStat *wslots = reinterpret_cast<NewtonStat *>(wsolve.as<AffMap>() + nw + 1);
            hipLaunchKernelGGL(costas_model_pass_kernel, dim3(nw), dim3(64), (size_t)64 * (rpc + 1) * sizeof(float2), s, sb,
                               S.as<float2>(), E.as<float2>(), J.as<float4>(), rpc, K, runs, gains, pol, aggs);
            newton_apply_waves(pol, (long long)K - 1, aggs, costas_ctl(counters), wslots, s);
        }
        XR_TRY(enqueue_passes(batch < max_passes ? batch : max_passes, s, prof));
    } else {
        XR_HIP(hipMemcpyAsync(S.p, st_in, sizeof(float2), hipMemcpyDeviceToDevice, s));
    }
    return enqueue_final(s, prof);
}

// after a stream synchronise: true when the hand-off closed within the passes enqueued so far
bool CostasStage::closed() const
{
    return job.n == 0 || job.K <= 1 || h_counters[0] != 0;
}

// Continues a call whose first batch did not close (cold start, unlocked input): more passes, looked at from the
// host every two, then the final pass again.  *redone tells the caller that the output was rewritten.
int CostasStage::finish(hipStream_t s, Profiler *prof, bool *redone)
{
    if (redone) *redone = false;
    if (job.n == 0) return XRIT_OK;
    const bool in_batch = closed();
    if (!in_batch) {
        if (redone) *redone = true;
        while (h_counters[0] == 0 && job.enqueued < max_passes) {
            if (h_counters[NEWTON_CTL_TAKEOVER]) job.gated = true;     // a boundary outside the trust region
            if (!job.rescued && job.enqueued >= rescue_after) XR_TRY(serial_rescue(s, prof));
            XR_TRY(enqueue_passes(2, s, prof));
            XR_HIP(hipMemcpyAsync(h_counters, counters.p, COSTAS_CTL_WORDS * sizeof(unsigned), hipMemcpyDeviceToHost, s));
            XR_HIP(hipStreamSynchronize(s));
        }
        XR_TRY(enqueue_final(s, prof));
        XR_HIP(hipStreamSynchronize(s));
    }
    passes = job.K > 1 ? (int)h_counters[1] : 0;
    if (job.K > 1) {
        // passes to enqueue next time before the host looks: what this call needed plus one spare -- four launches
        // that do nothing when the prediction holds.  After two calls in a row that closed inside their batch with
        // the same count the spare is dropped; the first call that then needs more (it continues from the host
        // and rewrites its output, see above) brings it back.
        stable = (in_batch && passes == last_passes) ? stable + 1 : 0;
        last_passes = passes;
        const int want = passes + ((stable >= 2 && !keep_spare) ? 0 : 1);
        batch = want < 1 ? 1 : (want > 6 ? 6 : want);
    }
    unconverged = job.K > 1 && h_counters[0] == 0 ? h_counters[2] : 0;
    uint32_t bits = h_counters[3];
    memcpy(&max_residual, &bits, sizeof(float));
    if (trace_env && job.K > 1) {
        if (h_counters[COSTAS_CTL_MODEL]) {
            unsigned hm[8];
            XR_HIP(hipMemcpy(hm, costas_cnt(counters, max_passes), sizeof hm, hipMemcpyDeviceToHost));
            float mr;
            unsigned long long qf;
            memcpy(&mr, &hm[2], 4);
            memcpy(&qf, &hm[4], 8);
            fprintf(stderr, "[xrit] costas model step: K=%d changed=%u open=%u max_r=%.3e rms_r=%.3e\n", job.K, hm[0], hm[1], mr,
                    hm[1] ? sqrtf((float)((double)qf / 1099511627776.0) / hm[1]) : 0.f);
        }
        // (a slot per pass enqueued: those that returned at once -- hand-off closed, or taken over by the gated solve -- are empty)
        std::vector<unsigned> hc((size_t)job.enqueued * 8);
        XR_HIP(hipMemcpy(hc.data(), costas_cnt(counters, 0), hc.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
        for (int p = 0; p < job.enqueued; ++p) {
            float mr;
            unsigned long long qf;
            memcpy(&mr, &hc[(size_t)p * 8 + 2], 4);
            memcpy(&qf, &hc[(size_t)p * 8 + 4], 8);
            const float q = (float)((double)qf / 1099511627776.0);
            if (hc[(size_t)p * 8 + 7] == 0 && hc[(size_t)p * 8 + 1] == 0) continue;
            fprintf(stderr, "[xrit] costas pass slot %d: K=%d changed=%u open=%u max_r=%.3e rms_r=%.3e\n", p, job.K,
                    hc[(size_t)p * 8], hc[(size_t)p * 8 + 1], mr,
                    hc[(size_t)p * 8 + 1] ? sqrtf(q / hc[(size_t)p * 8 + 1]) : 0.f);
        }
        fprintf(stderr, "[xrit] costas: %d passes, closed %u, on prediction %u, left open %u, max residual %.3e\n", passes, h_counters[0],
                h_counters[6], h_counters[2], max_residual);
    }
    if (job.exact) {
        XR_TRY(finish_exact(s, prof, redone));
        ex_blocks += h_xcnt ? h_xcnt[1] : 0;
        ex_picard += h_xcnt ? h_xcnt[2] : 0;
        ex_segs += h_xcnt ? h_xcnt[4] : 0;
        ex_fallbacks += h_xcnt ? h_xcnt[5] : 0;
        if (trace_env && h_xcnt)
            fprintf(stderr, "[xrit] costas exact: %d walkers, %u blocks, %.2f rounds per block, %u at the round limit, %u joints open after the batch, %d more rounds; lattice: %u segments, %u scans fell back\n",
                    ex_W, h_xcnt[1], h_xcnt[1] ? (double)h_xcnt[2] / h_xcnt[1] : 0.0, h_xcnt[3], ex_open, ex_rounds, h_xcnt[4], h_xcnt[5]);
    }
    cur ^= 1;     // the carried state now is the one the final pass left
    return XRIT_OK;
}

// The region between the first and the last boundary still open, walked serially (see costas_serial_states_kernel).
// Called from finish() with the stream idle; regions beyond rescue_max_samples are left to the passes.
int CostasStage::serial_rescue(hipStream_t s, Profiler *prof)
{
    job.rescued = true;
    if (job.K <= 2 || no_serial_walk) return XRIT_OK;
    XR_TRY(rescue.reserve(2 * sizeof(int)));
    const int init[2] = {0x7fffffff, -1};
    XR_HIP(hipMemcpyAsync(rescue.p, init, sizeof init, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(costas_open_range_kernel, dim3(div_up((size_t)job.K, 256)), dim3(256), 0, s, E.as<float2>(),
                       S.as<float2>(), job.K, tol_phase, tol_freq, rescue.as<int>());
    int range[2];
    XR_HIP(hipMemcpyAsync(range, rescue.p, sizeof range, hipMemcpyDeviceToHost, s));
    XR_HIP(hipStreamSynchronize(s));
    if (range[1] < 0) return XRIT_OK;                       // nothing open after all
    // chain range[0] ends at the first open boundary: it started from a closed one, so its start is good
    const int k_first = range[0], k_last = range[1];
    if ((long long)(k_last - k_first + 1) * L > rescue_max_samples) return XRIT_OK;
    {
        ProfScope ps(prof, "costas_serial", s);
        hipLaunchKernelGGL(costas_serial_states_kernel, dim3(1), dim3(64), 0, s, job.in, S.as<float2>(), flags.as<int>(),
                           (long long)job.n, L, job.K, k_first, k_last, gains);
    }
    XR_HIP(hipGetLastError());
    rescues += 1;
    walked = true;
    return XRIT_OK;
}

int CostasStage::run(const float2 *in, float2 *out, size_t n, hipStream_t s, Profiler *prof, const float2 *sub_ext,
                     double2 *om, long long om_off, double inv_sps)
{
    XR_TRY(begin(in, out, n, s, prof, sub_ext, om, om_off, inv_sps));
    XR_HIP(hipStreamSynchronize(s));
    return finish(s, prof, nullptr);
}

}  // namespace xrit
