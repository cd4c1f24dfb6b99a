// costas_exact.hip -- cfg.front_exact = 2: the Costas loop walked exactly, 64 samples per step on one wave.
// Replaces SatHelper::CostasLoop::Work (/root/reference/demodulator/src/demodulator.cpp:152) bit for bit as the CPU chain
// evaluates it (the test tier's CPU restatement: xo_costas_work with xo_sincosf).  costas.hip evaluates the loop as chains
// corrected by a Newton hand-off: within ~1e-6 rms of the serial loop, not ON it -- and the float32 Mueller & Mueller
// recurrence behind it turns any difference into 5e-5 .. 1.1e-4 rms of its own.  This file puts the
// output on the serial trajectory; the ranges, walkers and joints are exact_walk.h's, this is the policy:
//
//  * One step (cx_block).  Given the state (phase, freq) in front of a block of 64 samples, lane n holds a guess of the phase
//    in front of sample n, de-rotates its sample with it (exact_sincos.h: the C library's sincosf, operation for operation in
//    double precision) and forms its detector output e_n.  The loop filters are then two float additions per sample that do
//    not involve the samples any more -- freq += beta e_n; phase = (phase + freq) + alpha e_n -- and are run over the 64
//    samples by a scan that performs exactly the serial loop's additions (below).  If the PHASES that come out are bit for bit
//    the guesses that went in, all 64 e_n were the serial loop's and so is everything the scan made of them; if not, the new
//    phases are the next guess (a Picard iteration: lanes 0..j are exact after j rounds whatever the guess; the loop's gain
//    over 64 samples is ~0.08).  The first guess costs no trigonometry: e_n from the APPROXIMATE output costas.hip has left in
//    the output buffer, corrected for the slowly moving angle between the two (measured on the block before).  1.4 rounds per
//    block on the test signals.
//  * The scans.  cx_scan_lattice: inside a binade a float is an integer multiple of its ulp and adding a small increment moves
//    it by the increment rounded to that lattice, so the two recurrences are integer prefix sums (DPP) -- CERTIFIED lane by lane
//    by the literal step, restarted behind the first lane whose step does not reproduce (a tie, a binade change, a wrap).
//    cx_scan_fast / cx_scan_general: systolic, every lane adds its predecessor's values 63 times (v_add_f32_dpp wave_shr:1).
//  * Start states: costas.hip's chain start states S (approximate: a trajectory started a few ulps beside the true one coincides
//    with it after ~10 k samples, median; 99 %: 30 k), the carried state for the first range.  costas.hip still runs in this
//    mode: it leaves S, the first guess, and the clock recovery's timing statistic.
#include "kernels.h"

#include <cstdlib>
#include "exact_sincos.h"
#include "exact_walk.h"

namespace xrit {

namespace cx {

// the loop filters of ONE sample, literally (xo_costas_work: the phase takes the unclamped frequency)
__device__ __forceinline__ void cx_filters(float &phase, float &freq, float ae, float be)
{
    const float nf = freq + be;
    float np = (phase + nf) + ae;
    np = np > XR_TWOPI_F ? np - XR_TWOPI_F : np;
    np = np < -XR_TWOPI_F ? np + XR_TWOPI_F : np;
    phase = np;
    freq = fminf(fmaxf(nf, -1.0f), 1.0f);
}

// Systolic scan.  In: this lane's (alpha e_n, beta e_n) and the block's start state (uniform).  Out: the state in front of
// this lane's sample.  Every addition is the serial loop's.
__device__ __forceinline__ void cx_scan_general(float ae, float be, float ph0, float fr0, float &pv, float &fv)
{
    const int lane = threadIdx.x & 63;
    const float ash = xw::shr1(ae, 0.f), bsh = xw::shr1(be, 0.f);
    pv = ph0;
    fv = fr0;
    for (int it = 0; it < 63; ++it) {
        float ps = xw::shr1(pv, ph0), fs = xw::shr1(fv, fr0);
        cx_filters(ps, fs, ash, bsh);
        if (lane > 0) { pv = ps; fv = fs; }
    }
}

// The same without the phase wrap and the frequency limit, three instructions per round; the caller checks that neither
// would have acted (it does so on these very values: up to the first sample at which one acts they are the serial loop's).
__device__ __forceinline__ void cx_scan_fast(float ae, float be, float ph0, float fr0, float &pv, float &fv)
{
    const float ash = xw::shr1(ae, 0.f), bsh = xw::shr1(be, 0.f);
    float t = ph0;
    pv = ph0;
    fv = fr0;
    // lane 0 has no source lane: with bound_ctrl off a DPP instruction does not write it, so fv[0] stays fr0 and t[0] stays
    // ph0 (pv[0] = t[0] + 0).  A DPP source written by the VALU needs two wait states in front of the read: pv (written by the
    // last instruction of a round) is read two instructions later -- the s_nop makes it three.
    asm volatile("s_nop 1\n\t"
                 ".rept 63\n\t"
                 "v_add_f32_dpp %0, %0, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0\n\t"
                 "v_add_f32_dpp %1, %2, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32 %2, %1, %4\n\t"
                 ".endr\n\t"
                 "s_nop 1"
                 : "+v"(fv), "+v"(t), "+v"(pv)
                 : "v"(bsh), "v"(ash));
}


// ---- the scan on the float32 LATTICE, log depth ------------------------------------------------------------------------------
// Inside one binade a float is an integer multiple of its ulp, and adding a small increment to it moves it by the increment
// rounded to that lattice -- the same whole number of ulps whatever the running value is (ties and binade changes aside).  So
// the two recurrences become integer prefix sums: lane n turns its increment into lattice units against the block's BASE
// state (ord(fl(base + inc)) - ord(base): one float addition, the rounding is the hardware's), two DPP prefix sums give every
// lane its candidate state.  The candidates are then CERTIFIED, not trusted: every lane runs the literal step (cx_filters: the
// serial loop's own additions, wrap and limiter) from its candidate state and compares the result with its neighbour's
// candidate bit for bit.  Up to the first lane whose step does not reproduce, the states are the serial loop's; that lane's
// literal result is the exact state behind it and becomes the base of the next segment (a tie, a phase that crosses a power
// of two, a wrap at +-2 pi: one more segment each; a phase that hovers around zero changes binade all the time and falls
// back to the systolic scan).  ~60 instructions per segment against 250 for the systolic scan.
constexpr int CX_MAX_SEG = 4;


// true: (pv, fv) hold the exact state in front of every lane's sample.  *segs: segments used (statistics).
__device__ __forceinline__ bool cx_scan_lattice(float ae, float be, float ph0, float fr0, float &pv, float &fv, int *segs)
{
    const int lane = threadIdx.x & 63;
    float bph = ph0, bfr = fr0;           // the exact state in front of lane s
    int s = 0;
    pv = ph0;
    fv = fr0;
    for (int seg = 0; seg < CX_MAX_SEG; ++seg) {
        const bool in = lane >= s;
        const int of0 = xw::ord(bfr), op0 = xw::ord(bph);
        const int dF = in ? xw::ord(bfr + be) - of0 : 0;
        const int SF = xw::prefix(dF);
        const float fnext = xw::inv(of0 + SF), fown = xw::inv(of0 + SF - dF);
        const int d1 = in ? xw::ord(bph + fnext) - op0 : 0;
        const int d2 = in ? xw::ord(bph + ae) - op0 : 0;
        const int SP = xw::prefix(d1 + d2);
        const float pnext = xw::inv(op0 + SP), pown = xw::inv(op0 + SP - d1 - d2);
        float lp = pown, lf = fown;
        cx_filters(lp, lf, ae, be);
        const bool ok = __float_as_uint(lp) == __float_as_uint(pnext) && __float_as_uint(lf) == __float_as_uint(fnext);
        const unsigned long long bad = __builtin_amdgcn_ballot_w64(in && !ok);
        const int k = bad ? (int)__builtin_ctzll(bad) : 64;          // the first lane whose step the lattice did not reproduce
        if (in && lane <= k) { pv = pown; fv = fown; }
        if (segs) *segs = seg + 1;
        if (k >= 63) return true;         // (k == 63: its own state is certified; the state behind it is the caller's literal step)
        bph = xw::lane_of(lp, k);
        bfr = xw::lane_of(lf, k);
        s = k + 1;
    }
    return false;
}


// mode bit 0: the three-instruction systolic round where it is valid; bit 1: the lattice scan first
__device__ __forceinline__ void cx_scan(float ae, float be, float ph0, float fr0, float &pv, float &fv, int mode, unsigned *stat)
{
    if (mode & 2) {
        int segs = 0;
        const bool ok = cx_scan_lattice(ae, be, ph0, fr0, pv, fv, &segs);
        if (stat) { stat[0] += (unsigned)segs; stat[1] += ok ? 0u : 1u; }
        if (ok) return;
    }
    if (mode & 1) {
        cx_scan_fast(ae, be, ph0, fr0, pv, fv);
        // would the wrap or the limiter have acted anywhere?  (on the state in front of every sample and on the one behind
        // the last: lane 63's own step)
        const float nf = fv + be;
        const float np = (pv + nf) + ae;
        const bool out = !(fabsf(pv) <= XR_TWOPI_F) || !(fabsf(fv) <= 1.0f) || !(fabsf(np) <= XR_TWOPI_F) || !(fabsf(nf) <= 1.0f);
        if (!xw::any(out)) return;
    }
    cx_scan_general(ae, be, ph0, fr0, pv, fv);
}

struct CxGains { float alpha, beta; };

// One block of up to 64 samples from (ph, fr): lane n's sample x, the first guess ya (the approximate output), cnt valid
// lanes.  Leaves the exact outputs in (yr, yi) and the state behind the block's last sample in (ph, fr).
// Returns the number of Picard rounds (statistics).
// dphi (in / out): how far the exact phase was from the approximate one over the block before -- the approximate output is a
// rotation of the exact one by that angle, which moves slowly (it is what the chains' hand-offs left, forgotten over ~600
// samples): corrected for it, the first guess of the detector outputs is the serial loop's in most blocks and one Picard round
// confirms it (1.2 rounds per block instead of 2.1).
__device__ __forceinline__ int cx_block(float2 x, float2 ya, int cnt, float &ph, float &fr, CxGains g, float &yr, float &yi, int mode, unsigned *stat,
                                        float &dphi)
{
    const int lane = threadIdx.x & 63;
    const bool act = lane < cnt;
    // y_exact = y_approx e^{-j dphi}:  Re Im moves by dphi (Im^2 - Re^2)
    float e = bclip((mode & 4) ? ya.x * ya.y : ya.x * ya.y + dphi * (ya.y * ya.y - ya.x * ya.x), 1.0f);
    e = (act && e == e) ? e : 0.0f;
    float ae = g.alpha * e, be = g.beta * e;
    float pv, fv;
    cx_scan(ae, be, ph, fr, pv, fv, mode, stat);
    int rounds = 0;
    for (;;) {
        ++rounds;
        float sn, cs;
        exact_sincosf(-pv, sn, cs);
        yr = x.x * cs - x.y * sn;
        yi = x.x * sn + x.y * cs;
        e = bclip(yr * yi, 1.0f);
        e = act ? e : 0.0f;
        ae = g.alpha * e;
        be = g.beta * e;
        float p2, f2;
        cx_scan(ae, be, ph, fr, p2, f2, mode, stat);
        // (the detector outputs depend on the PHASES only: phases that come out as they went in were all the serial loop's -- lane
        // 0's is the block's start, and each one right makes the next one right --, and so is everything this scan made of
        // them, the frequencies included, whatever the frequencies of the round before were)
        const bool same = __float_as_uint(p2) == __float_as_uint(pv);
        pv = p2;
        fv = f2;
        if (xw::all(same) || rounds >= xw::MAX_ROUNDS) break;
    }
    {
        // the angle between the exact output and the approximate one, over this block: the next block's correction
        const float num = xw::wave_sum(act ? yr * ya.y - yi * ya.x : 0.0f), den = xw::wave_sum(act ? ya.x * ya.x + ya.y * ya.y : 0.0f);
        dphi = den > 0.0f ? num / den : 0.0f;
        dphi = fabsf(dphi) < 1e-3f ? dphi : 0.0f;        // (a guess's ingredient: anything wild is dropped)
    }
    // the state behind sample cnt - 1: that lane's own step
    float np = pv, nf = fv;
    cx_filters(np, nf, ae, be);
    ph = xw::lane_of(np, cnt - 1);
    fr = xw::lane_of(nf, cnt - 1);
    return rounds;
}


// the walker framework's policy (exact_walk.h): state = (phase, freq)
struct CostasWalk {
    static constexpr bool GUESS = true;        // the approximate output is the first guess of every block
    struct Par {
        const float2 *S;        // approximate chain start states (costas.hip), chains of L samples
        const float2 *st_in;    // the state carried into the call (exact)
        float2 *st_out;         // the state carried out of it
        int L;
        CxGains g;
    };
    __device__ static __forceinline__ float2 start(const Par &p, long long s)
    {
        const float2 st = s == 0 ? p.st_in[0] : p.S[s / p.L];
        return make_float2(costas_prewrap(st.x), st.y);
    }
    __device__ static __forceinline__ void carry_out(const Par &p, float2 st) { p.st_out[0] = st; }
    __device__ static __forceinline__ int block(const Par &p, float2 x, float2 ya, int cnt, float2 &st, float2 &out, int mode, unsigned *lat, float &aux)
    {
        float yr, yi;
        const int r = cx_block(x, ya, cnt, st.x, st.y, p.g, yr, yi, mode, lat, aux);
        out = make_float2(yr, yi);
        return r;
    }
};

__global__ void loop_sincosf_kernel(const float *__restrict__ x, float *__restrict__ sn, float *__restrict__ cs, long long n)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s, c;
    exact_sincosf(x[i], s, c);
    sn[i] = s;
    cs[i] = c;
}

}  // namespace cx
using namespace cx;

int launch_loop_sincosf(const float *d_x, float *d_sin, float *d_cos, size_t n, hipStream_t s)
{
    if (n == 0) return XRIT_OK;
    hipLaunchKernelGGL(loop_sincosf_kernel, dim3(div_up(n, 256)), dim3(256), 0, s, d_x, d_sin, d_cos, (long long)n);
    XR_HIP(hipGetLastError());
    return XRIT_OK;
}

// The plan of a call.  A walker is one wave; the chip holds one per SIMD without their sharing an issue port (ex_walkers: 4 per
// CU), so a call is cut into that many ranges.  A range's walker must have MET the serial trajectory by the end of its range
// (its end state is what the next joint is settled from): `span` = 32 Ki samples walked at least (the merge time's 99 %).
//  * ranges of `span` samples and more (large calls, where the walkers' total work is what counts): NO warm-up.  Every walker
//    starts at its own range from the approximate state, and the first round over the joints lets walker w - 1 go on into
//    range w until it meets walker w's record (~10 k samples, median).  A warm-up is paid by every walker, going on only by
//    the part that is missing: work n + 12 k per joint instead of n + 32 k per walker.
//  * shorter ranges (calls below span x walkers: the chip is not full, latency is what counts): a warm-up of span - range
//    samples, so that 99 % of the joints fit at once.
// ex_hist >= 0 (XRIT_CX_HIST, xrit_costas_set_exact): that warm-up whatever the call's size.
int CostasStage::exact_plan(size_t n, int *Lw, int *W, int *H) const
{
    const size_t unit = (size_t)L * 64 / (size_t)(L % 64 == 0 ? 64 : 1);      // lcm(L, 64) for the chain lengths in use
    const size_t span = (32768 + unit - 1) / unit * unit;
    size_t lw = (n + (size_t)ex_walkers - 1) / (size_t)ex_walkers;
    // (twice the walkers where their ranges still reach `span` without a warm-up -- 2^28 samples at the circuit rate: the walkers
    // are throughput there, and four waves per SIMD issue better than two: 4.6 against 6.0 ms per C1 burst; at C2's 54 M
    // samples the shorter ranges would have to be warmed up: 8.0 against 6.5 ms)
    if (lw >= 2 * span) lw = (lw + 1) / 2;
    if (lw < 2048) lw = 2048;
    lw = (lw + unit - 1) / unit * unit;
    size_t h = lw >= span ? 0 : span - lw;
    if (ex_hist >= 0) h = ((size_t)ex_hist + unit - 1) / unit * unit;
    *Lw = (int)lw;
    *W = (int)((n + lw - 1) / lw);
    *H = (int)h;
    return XRIT_OK;
}

static xw::KArgs<CostasWalk> cx_args(CostasStage &c, int Lw, int W, int H)
{
    xw::KArgs<CostasWalk> K{};
    xw::Args &A = K.a;
    A.x = c.job.in; A.y = c.job.out;
    A.js = c.xj.as<float2>(); A.je = A.js + W; A.used = A.je + W;
    A.bs = c.xbs.as<float2>(); A.cnt = c.xcnt.as<unsigned>();
    A.n = (long long)c.job.n; A.Lw = Lw; A.H = H; A.W = W;
    A.mode = c.ex_mode;
    A.prio = c.ex_prio;
    K.p.S = c.S.as<float2>();
    K.p.st_in = c.state.as<float2>() + c.cur;
    K.p.st_out = c.state.as<float2>() + (c.cur ^ 1);
    K.p.L = c.L;
    K.p.g = CxGains{c.gains.alpha, c.gains.beta};
    return K;
}

int CostasStage::enqueue_exact(hipStream_t s, Profiler *prof)
{
    if (job.n == 0) return XRIT_OK;
    int Lw = 0, W = 0, H = 0;
    XR_TRY(exact_plan(job.n, &Lw, &W, &H));
    XR_TRY(xj.reserve((size_t)3 * W * sizeof(float2)));
    XR_TRY(xbs.reserve(((job.n >> 6) + 2) * sizeof(float2)));
    XR_TRY(xcnt.reserve(xw::NCNT * sizeof(unsigned)));
    if (!h_xcnt) XR_HIP(hipHostMalloc((void **)&h_xcnt, 64));
    const xw::KArgs<CostasWalk> K = cx_args(*this, Lw, W, H);
    ex_W = W;
    ex_rounds = 0;
    {
        ProfScope ps(prof, "costas_exact", s);
        hipLaunchKernelGGL(xw::zero_kernel<CostasWalk>, dim3(1), dim3(64), 0, s, K.a.cnt, xw::NCNT);
        hipLaunchKernelGGL(xw::main_kernel<CostasWalk>, dim3(W), dim3(64), 0, s, K);
        hipLaunchKernelGGL(xw::used_kernel<CostasWalk>, dim3(div_up((size_t)W, 256)), dim3(256), 0, s, K.a);
    }
    if (W > 1) {
        ProfScope ps(prof, "costas_exact_fix", s);
        // (rounds enqueued with the call: the first one does the settling, the next catch the joints whose predecessor's end state
        // that changed; the host looks at what is left)
        // (four: a round that finds nothing to do is a launch of waves that look at two words and leave, ~10 us; a joint the host has
        // to close costs the burst its speculatively started clock-recovery walkers -- with two rounds one C2 burst in five)
        for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(xw::fix_kernel<CostasWalk>, dim3(W - 1), dim3(64), 0, s, K, 0);
        hipLaunchKernelGGL(xw::zero_kernel<CostasWalk>, dim3(1), dim3(64), 0, s, K.a.cnt, 1);
        hipLaunchKernelGGL(xw::fix_kernel<CostasWalk>, dim3(W - 1), dim3(64), 0, s, K, 1);
    }
    XR_HIP(hipMemcpyAsync(h_xcnt, xcnt.p, xw::NCNT * sizeof(unsigned), hipMemcpyDeviceToHost, s));
    XR_HIP(hipGetLastError());
    ex_args_valid = true;
    return XRIT_OK;
}

// after the host has seen the end of what enqueue_exact() queued: joints still open are closed round by round
int CostasStage::finish_exact(hipStream_t s, Profiler *prof, bool *redone)
{
    if (job.n == 0 || !ex_args_valid) return XRIT_OK;
    ex_open = h_xcnt[0];
    ex_nonconverged = h_xcnt[3];
    if (ex_W <= 1) return XRIT_OK;
    int Lw = 0, W = 0, H = 0;
    XR_TRY(exact_plan(job.n, &Lw, &W, &H));
    const xw::KArgs<CostasWalk> K = cx_args(*this, Lw, W, H);
    while (h_xcnt[0] != 0) {
        if (ex_rounds > W + 4) { set_error("Costas loop (exact): %u joints still open after %d rounds", h_xcnt[0], ex_rounds); return XRIT_E_NOT_CONVERGED; }
        if (redone) *redone = true;
        ++ex_rounds;
        ProfScope ps(prof, "costas_exact_fix", s);
        hipLaunchKernelGGL(xw::fix_kernel<CostasWalk>, dim3(W - 1), dim3(64), 0, s, K, 0);
        hipLaunchKernelGGL(xw::zero_kernel<CostasWalk>, dim3(1), dim3(64), 0, s, K.a.cnt, 1);
        hipLaunchKernelGGL(xw::fix_kernel<CostasWalk>, dim3(W - 1), dim3(64), 0, s, K, 1);
        XR_HIP(hipMemcpyAsync(h_xcnt, xcnt.p, xw::NCNT * sizeof(unsigned), hipMemcpyDeviceToHost, s));
        XR_HIP(hipStreamSynchronize(s));
    }
    return XRIT_OK;
}

}  // namespace xrit
