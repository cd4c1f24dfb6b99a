// costas_exact.hip -- cfg.front_exact = 2: the Costas loop walked exactly, 64 samples per step on one wave.
// Replaces SatHelper::CostasLoop::Work (/root/reference/demodulator/src/demodulator.cpp:152) bit for bit as the CPU chain
// evaluates it (the test tier's CPU restatement: xo_costas_work with xo_sincosf).  costas.hip evaluates the loop as chains corrected by a
// Newton hand-off: within ~1e-6 rms of the serial loop, not ON it -- and the float32 Mueller & Mueller recurrence behind it
// turns any difference into 5e-5 .. 1.1e-4 rms of its own.  This file puts the output on the serial
// trajectory:
//
//  * One step.  Given the state (phase, freq) in front of a block of 64 samples, lane n holds a guess of the phase in front of
//    sample n, de-rotates its sample with it (exact_sincos.h: the C library's sincosf, operation for operation in double
//    precision) and forms its detector output e_n.  The loop filters are then two float additions per sample that do not
//    involve the samples any more -- freq += beta e_n; phase = (phase + freq) + alpha e_n -- and are run as a SYSTOLIC scan:
//    every lane adds its predecessor's values (v_add_f32_dpp ... wave_shr:1), 63 times, so that lane n ends up with the state
//    in front of sample n computed by exactly the serial loop's additions in the serial loop's order.  If the phases that
//    come out are bit for bit the guesses that went in, all 64 e_n were the serial loop's and so is everything else; if not,
//    the new phases are the next guess (a Picard iteration: lanes 0..j are exact after j rounds whatever the guess, and the
//    loop's gain over 64 samples is ~0.08, so a guess that is 1e-6 off is exact after one or two rounds).  The first guess
//    costs no trigonometry: e_n is taken from the APPROXIMATE output costas.hip has left in the output buffer.
//    Measured on the CPU (prototype of this scheme on the oracle's signal): 1.76 rounds per block on average.
//  * Walkers.  A call is cut into ranges; the walker of a range starts `hist` samples in front of it from the approximate
//    chain start state there (costas.hip's S) and walks those samples quietly: the loop is contractive, and a float32
//    trajectory started a few ulps beside the true one COINCIDES with it after ~10 k samples (median; 99 %: 30 k; 250 trials
//    per mode on the CPU) and stays on it.  Walker 0 -- and every walker that would start in front of the call -- starts
//    from the carried state, exactly.
//  * Joints.  Walker w's state at the start of its range must be walker w - 1's state at the end of its own, bit for bit;
//    by induction from walker 0 every output then is the serial loop's.  Where it is not (~1 % of the joints at the default
//    history), walker w - 1 goes on into range w from its end state, rewriting the output, until its state meets the record
//    walker w left at every block boundary (costas_exact_fix_kernel; rounds until nothing changes: two are enqueued with
//    the call, the host looks at the count afterwards and goes on in the rare case something is left).
#include "kernels.h"

#include <cstdlib>
#include "exact_sincos.h"

namespace xrit {

namespace {

constexpr int CX_MAX_ROUNDS = 72;       // lanes 0..j are exact after j Picard rounds: 64 always suffice

__device__ __forceinline__ float cx_shr1(float v, float first)
{
    // lane i <- lane i - 1, lane 0 <- first
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(first), __float_as_int(v), 0x138, 0xf, 0xf, false));
}

__device__ __forceinline__ float cx_lane(float v, int src)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

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
    const float ash = cx_shr1(ae, 0.f), bsh = cx_shr1(be, 0.f);
    pv = ph0;
    fv = fr0;
    for (int it = 0; it < 63; ++it) {
        float ps = cx_shr1(pv, ph0), fs = cx_shr1(fv, fr0);
        cx_filters(ps, fs, ash, bsh);
        if (lane > 0) { pv = ps; fv = fs; }
    }
}

// The same without the phase wrap and the frequency limit, three instructions per round; the caller checks that neither
// would have acted (it does so on these very values: up to the first sample at which one acts they are the serial loop's).
__device__ __forceinline__ void cx_scan_fast(float ae, float be, float ph0, float fr0, float &pv, float &fv)
{
    const float ash = cx_shr1(ae, 0.f), bsh = cx_shr1(be, 0.f);
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

__device__ __forceinline__ bool cx_all(bool p) { return __builtin_amdgcn_ballot_w64(p) == ~0ull; }
__device__ __forceinline__ bool cx_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }

__device__ __forceinline__ void cx_scan(float ae, float be, float ph0, float fr0, float &pv, float &fv, bool fast_ok)
{
    if (fast_ok) {
        cx_scan_fast(ae, be, ph0, fr0, pv, fv);
        // would the wrap or the limiter have acted anywhere?  (on the state in front of every sample and on the one behind
        // the last: lane 63's own step)
        float np = pv, nf = fv;
        {
            const float f1 = fv + be;
            np = (pv + f1) + ae;
            nf = f1;
        }
        const bool out = !(fabsf(pv) <= XR_TWOPI_F) || !(fabsf(fv) <= 1.0f) || !(fabsf(np) <= XR_TWOPI_F) || !(fabsf(nf) <= 1.0f);
        if (!cx_any(out)) return;
    }
    cx_scan_general(ae, be, ph0, fr0, pv, fv);
}

struct CxGains { float alpha, beta; };

// One block of up to 64 samples from (ph, fr): lane n's sample x, the first guess ya (the approximate output), cnt valid
// lanes.  Leaves the exact outputs in (yr, yi) and the state behind the block's last sample in (ph, fr).
// Returns the number of Picard rounds (statistics).
__device__ __forceinline__ int cx_block(float2 x, float2 ya, int cnt, float &ph, float &fr, CxGains g, float &yr, float &yi, bool fast_ok)
{
    const int lane = threadIdx.x & 63;
    const bool act = lane < cnt;
    float e = bclip(ya.x * ya.y, 1.0f);
    e = (act && e == e) ? e : 0.0f;
    float ae = g.alpha * e, be = g.beta * e;
    float pv, fv;
    cx_scan(ae, be, ph, fr, pv, fv, fast_ok);
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
        cx_scan(ae, be, ph, fr, p2, f2, fast_ok);
        const bool same = __float_as_uint(p2) == __float_as_uint(pv) && __float_as_uint(f2) == __float_as_uint(fv);
        pv = p2;
        fv = f2;
        if (cx_all(same) || rounds >= CX_MAX_ROUNDS) break;
    }
    // the state behind sample cnt - 1: that lane's own step
    float np = pv, nf = fv;
    cx_filters(np, nf, ae, be);
    ph = cx_lane(np, cnt - 1);
    fr = cx_lane(nf, cnt - 1);
    return rounds;
}

struct CxArgs {
    const float2 *x;        // the loop's input (matched filter output)
    float2 *y;              // in: the approximate output; out: the exact one
    const float2 *S;        // approximate chain start states (costas.hip), chains of L samples
    const float2 *st_in;    // the state carried into the call (exact)
    float2 *st_out;         // the state carried out of it
    float2 *js, *je, *used; // per walker: state at the start of its range / at its end / the start state its output belongs to
    float2 *bs;             // per block of its range: the state behind the block
    unsigned *cnt;          // [0] joints that did not fit (this round), [1] blocks walked, [2] Picard rounds, [3] non-converged blocks
    long long n;
    int L, Lw, H, W;
    CxGains g;
    int fast_ok;
};

// walker w over blocks [from, to) (sample indices, multiples of 64 except the call's end) starting from (ph, fr); outputs
// are written from `write_from` on.  MEET: stop at the first block boundary where the state equals the record in bs
// (returns true there); otherwise the records are (re)written.
template <bool MEET>
__device__ __forceinline__ bool cx_walk(const CxArgs &A, long long from, long long to, long long write_from, float &ph, float &fr,
                                        float2 *js_slot, long long js_at)
{
    const int lane = threadIdx.x & 63;
    unsigned blocks = 0, rounds = 0, bad = 0;
    bool met = false;
    const long long last = A.n - 1;
    auto idx_of = [&](long long blk) { const long long i = blk + lane; return i < last ? i : last; };
    float2 xn = A.x[idx_of(from)], yn = A.y[idx_of(from)];
    for (long long blk = from; blk < to; blk += 64) {
        const float2 x = xn, ya = yn;
        if (blk + 64 < to) { xn = A.x[idx_of(blk + 64)]; yn = A.y[idx_of(blk + 64)]; }
        const int cnt = (int)min((long long)64, to - blk);
        float yr, yi;
        const int r = cx_block(x, ya, cnt, ph, fr, A.g, yr, yi, A.fast_ok != 0);
        ++blocks;
        rounds += (unsigned)r;
        bad += r >= CX_MAX_ROUNDS ? 1u : 0u;
        if (blk >= write_from && lane < cnt) A.y[blk + lane] = make_float2(yr, yi);
        if (js_slot != nullptr && blk + 64 == js_at && lane == 0) *js_slot = make_float2(ph, fr);
        if (blk >= write_from) {
            float2 *rec = A.bs + (blk >> 6);
            if (MEET) {
                const float2 old = *rec;
                if (__float_as_uint(old.x) == __float_as_uint(ph) && __float_as_uint(old.y) == __float_as_uint(fr)) { met = true; break; }
            }
            if (lane == 0) *rec = make_float2(ph, fr);
        }
    }
    if (lane == 0) {
        atomicAdd(A.cnt + 1, blocks);
        atomicAdd(A.cnt + 2, rounds);
        if (bad) atomicAdd(A.cnt + 3, bad);
    }
    return met;
}

__global__ void __launch_bounds__(64) costas_exact_kernel(CxArgs A)
{
    const int w = blockIdx.x;
    const int lane = threadIdx.x;
    const long long a = (long long)w * A.Lw;
    const long long b = min(A.n, a + A.Lw);
    const long long s = a > A.H ? a - A.H : 0;
    float2 st = s == 0 ? A.st_in[0] : A.S[s / A.L];
    float ph = costas_prewrap(st.x), fr = st.y;
    if (s == a && lane == 0) A.js[w] = make_float2(ph, fr);
    (void)cx_walk<false>(A, s, b, a, ph, fr, s < a ? A.js + w : nullptr, a);
    if (lane == 0) {
        A.je[w] = make_float2(ph, fr);
        if (b == A.n) A.st_out[0] = make_float2(ph, fr);
    }
}

// the hand-over of the joints' start states: used[w] = js[w] (a kernel of its own: every walker has finished)
__global__ void costas_exact_used_kernel(CxArgs A)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w < A.W) A.used[w] = A.js[w];
    if (w == 0) A.cnt[0] = 0;
}

// one round over the joints.  count_only: just count the joints that do not fit.
__global__ void __launch_bounds__(64) costas_exact_fix_kernel(CxArgs A, int count_only)
{
    const int w = blockIdx.x + 1;
    const int lane = threadIdx.x;
    const float2 prev = A.je[w - 1], mine = A.used[w];
    if (__float_as_uint(prev.x) == __float_as_uint(mine.x) && __float_as_uint(prev.y) == __float_as_uint(mine.y)) return;
    if (lane == 0) atomicAdd(A.cnt + 0, 1u);
    if (count_only) return;
    const long long a = (long long)w * A.Lw;
    const long long b = min(A.n, a + A.Lw);
    float ph = prev.x, fr = prev.y;
    const bool met = cx_walk<true>(A, a, b, a, ph, fr, nullptr, 0);
    if (lane == 0) {
        A.used[w] = prev;
        if (!met) {
            A.je[w] = make_float2(ph, fr);
            if (b == A.n) A.st_out[0] = make_float2(ph, fr);
        }
    }
}

__global__ void costas_exact_zero_kernel(unsigned *cnt) { if (threadIdx.x < 4) cnt[threadIdx.x] = 0; }

__global__ void loop_sincosf_kernel(const float *__restrict__ x, float *__restrict__ sn, float *__restrict__ cs, long long n)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s, c;
    exact_sincosf(x[i], s, c);
    sn[i] = s;
    cs[i] = c;
}

}  // namespace

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
    if (lw < 2048) lw = 2048;
    lw = (lw + unit - 1) / unit * unit;
    size_t h = lw >= span ? 0 : span - lw;
    if (ex_hist >= 0) h = ((size_t)ex_hist + unit - 1) / unit * unit;
    *Lw = (int)lw;
    *W = (int)((n + lw - 1) / lw);
    *H = (int)h;
    return XRIT_OK;
}

int CostasStage::enqueue_exact(hipStream_t s, Profiler *prof)
{
    if (job.n == 0) return XRIT_OK;
    int Lw = 0, W = 0, H = 0;
    XR_TRY(exact_plan(job.n, &Lw, &W, &H));
    XR_TRY(xj.reserve((size_t)3 * W * sizeof(float2)));
    XR_TRY(xbs.reserve(((job.n >> 6) + 2) * sizeof(float2)));
    XR_TRY(xcnt.reserve(8 * sizeof(unsigned)));
    if (!h_xcnt) XR_HIP(hipHostMalloc((void **)&h_xcnt, 64));
    CxArgs A{};
    A.x = job.in; A.y = job.out; A.S = S.as<float2>();
    A.st_in = state.as<float2>() + cur; A.st_out = state.as<float2>() + (cur ^ 1);
    A.js = xj.as<float2>(); A.je = A.js + W; A.used = A.je + W;
    A.bs = xbs.as<float2>(); A.cnt = xcnt.as<unsigned>();
    A.n = (long long)job.n; A.L = L; A.Lw = Lw; A.H = H; A.W = W;
    A.g = CxGains{gains.alpha, gains.beta};
    A.fast_ok = ex_fast ? 1 : 0;
    ex_W = W;
    ex_rounds = 0;
    {
        ProfScope ps(prof, "costas_exact", s);
        hipLaunchKernelGGL(costas_exact_zero_kernel, dim3(1), dim3(64), 0, s, A.cnt);
        hipLaunchKernelGGL(costas_exact_kernel, dim3(W), dim3(64), 0, s, A);
        hipLaunchKernelGGL(costas_exact_used_kernel, dim3(div_up((size_t)W, 256)), dim3(256), 0, s, A);
    }
    if (W > 1) {
        ProfScope ps(prof, "costas_exact_fix", s);
        // (rounds enqueued with the call: without a warm-up the first one does the settling, the second catches the joints
        // whose predecessor's end state that changed; the host looks at what is left)
        for (int r = 0; r < (H == 0 ? 3 : 2); ++r) hipLaunchKernelGGL(costas_exact_fix_kernel, dim3(W - 1), dim3(64), 0, s, A, 0);
        hipLaunchKernelGGL(costas_exact_zero_kernel, dim3(1), dim3(1), 0, s, A.cnt);
        hipLaunchKernelGGL(costas_exact_fix_kernel, dim3(W - 1), dim3(64), 0, s, A, 1);
    }
    XR_HIP(hipMemcpyAsync(h_xcnt, xcnt.p, 4 * sizeof(unsigned), hipMemcpyDeviceToHost, s));
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
    CxArgs A{};
    A.x = job.in; A.y = job.out; A.S = S.as<float2>();
    A.st_in = state.as<float2>() + cur; A.st_out = state.as<float2>() + (cur ^ 1);
    A.js = xj.as<float2>(); A.je = A.js + W; A.used = A.je + W;
    A.bs = xbs.as<float2>(); A.cnt = xcnt.as<unsigned>();
    A.n = (long long)job.n; A.L = L; A.Lw = Lw; A.H = H; A.W = W;
    A.g = CxGains{gains.alpha, gains.beta};
    A.fast_ok = ex_fast ? 1 : 0;
    while (h_xcnt[0] != 0) {
        if (ex_rounds > W + 4) { set_error("Costas loop (exact): %u joints still open after %d rounds", h_xcnt[0], ex_rounds); return XRIT_E_NOT_CONVERGED; }
        if (redone) *redone = true;
        ++ex_rounds;
        ProfScope ps(prof, "costas_exact_fix", s);
        hipLaunchKernelGGL(costas_exact_fix_kernel, dim3(W - 1), dim3(64), 0, s, A, 0);
        hipLaunchKernelGGL(costas_exact_zero_kernel, dim3(1), dim3(1), 0, s, A.cnt);
        hipLaunchKernelGGL(costas_exact_fix_kernel, dim3(W - 1), dim3(64), 0, s, A, 1);
        XR_HIP(hipMemcpyAsync(h_xcnt, xcnt.p, 4 * sizeof(unsigned), hipMemcpyDeviceToHost, s));
        XR_HIP(hipStreamSynchronize(s));
    }
    return XRIT_OK;
}

}  // namespace xrit
