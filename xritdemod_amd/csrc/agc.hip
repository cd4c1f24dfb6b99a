// agc.hip -- automatic gain control as an exact parallel scan.
// Replaces SatHelper::AGC::Work (/root/reference/demodulator/src/demodulator.cpp:143,
// object built at :447 with Parameters.h:34-37): y = x*g; g += rate*(ref - |y|);
// g = min(g, max).  For g > 0 one step is the map g -> min(a g + b, c) with
// a = 1 - rate*|x| >= 0, and such maps compose associatively, so block-start
// gains come from a prefix scan; inside its own 4-sample run every lane replays
// the recurrence literally.  A sample with rate*|x| > 1 (a < 0) breaks
// monotonicity: it raises a guard flag and a single-lane kernel redoes the call
// serially (never seen with normalised SDR samples; kept for exactness).
#include "kernels.h"

#include <cstdlib>
#include "scan.h"
#include "agc_wave.h"
#include "exact_walk.h"

namespace xrit {

struct AgcScanF {
    typedef AgcMap T;
    const float2 *x;
    float2 *y;
    const float *state_in;   // [0] gain
    float *state_out;        // [0] gain after the call, [1] guard flag
    float rate, ref, maxg;
    long long n;
    int vec;                 // both buffers 16-byte aligned: 16-byte accesses for whole runs

    __device__ T identity() const { return agc_identity(); }
    __device__ T combine(const T &lo, const T &hi) const { return agc_compose(lo, hi); }
    __device__ T reduce_run(long long i0, int cnt) const
    {
        T m = agc_identity();
        bool bad = false;
        float2 v[SCAN_IPT];
        if (cnt == SCAN_IPT && vec) {
            const float4 *xp = reinterpret_cast<const float4 *>(x + i0);
            float4 a = xp[0], b = xp[1];
            v[0] = make_float2(a.x, a.y); v[1] = make_float2(a.z, a.w);
            v[2] = make_float2(b.x, b.y); v[3] = make_float2(b.z, b.w);
        } else {
            for (int k = 0; k < cnt; ++k) v[k] = x[i0 + k];
        }
        for (int k = 0; k < cnt; ++k) {
            T e = agc_sample_map(v[k].x, v[k].y, rate, ref, maxg);
            bad |= !(e.a >= 0.0f);
            m = agc_compose(m, e);
        }
        if (bad) state_out[1] = 1.0f;
        return m;
    }
    __device__ void apply_run(long long i0, int cnt, const T &pre) const
    {
        float g = agc_apply(pre, state_in[0]);
        if (cnt == SCAN_IPT && vec) {
            // whole run: 16-byte loads and stores (i0 is a multiple of 4 samples)
            const float4 *xp = reinterpret_cast<const float4 *>(x + i0);
            float4 a = xp[0], b = xp[1], oa, ob;
            agc_step(a.x, a.y, g, rate, ref, maxg, oa.x, oa.y);
            agc_step(a.z, a.w, g, rate, ref, maxg, oa.z, oa.w);
            agc_step(b.x, b.y, g, rate, ref, maxg, ob.x, ob.y);
            agc_step(b.z, b.w, g, rate, ref, maxg, ob.z, ob.w);
            float4 *yp = reinterpret_cast<float4 *>(y + i0);
            yp[0] = oa;
            yp[1] = ob;
        } else {
            for (int k = 0; k < cnt; ++k) {
                float2 v = x[i0 + k];
                float yr, yi;
                agc_step(v.x, v.y, g, rate, ref, maxg, yr, yi);
                y[i0 + k] = make_float2(yr, yi);
            }
        }
        if (i0 + cnt == n) state_out[0] = g;
    }
};

__global__ void agc_serial_kernel(const float2 *x, float2 *y, const float *state_in, float *state_out,
                                  float rate, float ref, float maxg, long long n, int force)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    if (!force && state_out[1] == 0.0f) return;
    float g = state_in[0];
    for (long long i = 0; i < n; ++i) {
        float2 v = x[i];
        float yr, yi;
        agc_step(v.x, v.y, g, rate, ref, maxg, yr, yi);
        y[i] = make_float2(yr, yi);
    }
    state_out[0] = g;
    state_out[1] = 2.0f;  // serial path taken
}

__global__ void agc_begin_kernel(float *state_out)
{
    state_out[1] = 0.0f;
}

int AgcStage::init(float rate_, float reference, float gain0_, float max_gain)
{
    rate = rate_;
    ref = reference;
    maxg = max_gain;
    gain0 = gain0_;
    {
        int cus = 0, dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
            ex_walkers = 8 * cus;
        if (const char *e = getenv("XRIT_CX_WALKERS")) { const int v = atoi(e); if (v > 0) ex_walkers = v; }
        if (const char *e = getenv("XRIT_CX_MODE")) ex_mode = atoi(e) & 3;
    }
    XR_TRY(state.reserve(4 * sizeof(float)));
    float h[4] = {gain0_, 0.0f, gain0_, 0.0f};   // two (gain, flag) slots, ping-pong
    XR_HIP(hipMemcpy(state.p, h, sizeof h, hipMemcpyHostToDevice));
    if (!h_flag) XR_HIP(hipHostMalloc((void **)&h_flag, 64));
    *h_flag = 0.0f;
    cur = 0;
    return XRIT_OK;
}

__global__ void agc_reset_kernel(float *st, float g)
{
    st[0] = g; st[1] = 0.f; st[2] = g; st[3] = 0.f;
}

int AgcStage::reset(hipStream_t s)
{
    hipLaunchKernelGGL(agc_reset_kernel, dim3(1), dim3(1), 0, s, state.as<float>(), gain0);
    cur = 0;
    return XRIT_OK;
}

void AgcStage::release()
{
    state.release();
    aggs.release();
    joints.release();
    if (h_flag) { (void)hipHostFree(h_flag); h_flag = nullptr; }
}

int AgcStage::request_flag_at(const float *flag, hipStream_t s)
{
    XR_HIP(hipMemcpyAsync(h_flag, flag, sizeof(float), hipMemcpyDeviceToHost, s));
    return XRIT_OK;
}

int AgcStage::request_flag(hipStream_t s)
{
    XR_HIP(hipMemcpyAsync(h_flag, state.as<float>() + 2 * cur + 1, sizeof(float), hipMemcpyDeviceToHost, s));
    return XRIT_OK;
}

int AgcStage::run(const float2 *in, float2 *out, size_t n, hipStream_t s, Profiler *prof, bool use_exact)
{
    if (exact || use_exact) return run_exact(in, out, n, s, prof);
    if (n == 0) return XRIT_OK;
    float *sin_ = state.as<float>() + 2 * cur;
    float *sout = state.as<float>() + 2 * (cur ^ 1);
    int nb = scan_blocks((long long)n);
    XR_TRY(aggs.reserve((size_t)(nb + scan_blocks(nb) + 4) * sizeof(AgcMap)));
    const int vec = (((uintptr_t)in | (uintptr_t)out) & 15) == 0;
    AgcScanF f{in, out, sin_, sout, rate, ref, maxg, (long long)n, vec};
    {
        ProfScope ps(prof, "agc_reduce", s);
        hipLaunchKernelGGL(agc_begin_kernel, dim3(1), dim3(1), 0, s, sout);
        hipLaunchKernelGGL(scan_reduce_kernel<AgcScanF>, dim3(nb), dim3(SCAN_BLOCK), 0, s, f, (long long)n,
                           aggs.as<AgcMap>());
    }
    {
        ProfScope ps(prof, "agc_scan", s);
        scan_aggs_launch(f, aggs.as<AgcMap>(), nb, s);
    }
    {
        ProfScope ps(prof, "agc_apply", s);
        hipLaunchKernelGGL(scan_apply_kernel<AgcScanF>, dim3(nb), dim3(SCAN_BLOCK), 0, s, f, (long long)n,
                           aggs.as<AgcMap>());
        hipLaunchKernelGGL(agc_serial_kernel, dim3(1), dim3(1), 0, s, in, out, sin_, sout, rate, ref, maxg,
                           (long long)n, 0);
    }
    XR_HIP(hipGetLastError());
    cur ^= 1;
    return XRIT_OK;
}


// ---- cfg.front_exact = 2: the AGC walked exactly -----------------------------------------------------------------------------
// The scan above evaluates the recurrence as composed maps: exact in real arithmetic, some ulps from the float32 recurrence
// the CPU chain runs (AGC::Work, demodulator.cpp:143; the test tier's CPU restatement: xo_agc_work).  Bit for bit that
// recurrence is only what it is when it is walked -- but a walk from a start gain that is a few ulps off BECOMES the true
// trajectory: the loop is contractive and its state a single float, so the two coincide bit for bit after ~2 k samples
// (median; 99 %: 5.7 k, tests/experiments/agc_merge_time.py) and stay together.  The walk is exact_walk.h's: ranges, one wave
// each, started AGC_EX_WARM samples early from the gain the map scan gives there (range 0 and every range that would start
// in front of the call: from the carried gain, exactly), joints settled bit for bit afterwards.  One step = 64 samples:
//   * first guess of the gain in front of every sample: the lanes' gain maps composed across the wave (agc_wave_exclusive);
//   * Picard rounds: every lane forms its increment rate * (ref - |x g|) from its guess -- the correctly rounded square root
//     here, the shipped default takes v_sqrt_f32 --, the additions g += d (and the clamp) are then run for all 64 samples as
//     an integer prefix sum on the float lattice, certified lane by lane by the literal step (agx_scan, the scheme of
//     costas_exact.hip with one state variable); gains that come out as they went in are the serial recurrence's.
// Joints still open after the rounds enqueued with the call (never seen) raise the guard flag and the serial kernel redoes
// the call.  (Round 6's first version walked chains of 4096 samples one LANE each: 2.1 ms per C2 burst of pure latency --
// 16 k dependent steps of ~300 cycles; this one 0.3.)
constexpr int AGC_EX_WARM = 12288;
constexpr int AGC_EX_ROUNDS = 3;

__device__ __forceinline__ void agc_step_exact(float xr, float xi, float &g, float rate, float ref, float maxg, float &yr, float &yi)
{
    yr = xr * g;
    yi = xi * g;
    g += rate * (ref - ::sqrtf(yr * yr + yi * yi));      // (sqrtf: the correctly rounded expansion; __fsqrt_rn is v_sqrt_f32, 1 ulp)
    if (maxg > 0.0f && g > maxg) g = maxg;
}

// the additions of 64 samples: d = this lane's increment, g0 the gain in front of lane 0 (uniform).  Out: the gain in front
// of this lane's sample, by the serial recurrence's own additions.
__device__ __forceinline__ float agx_scan(float d, float g0, float maxg, int mode, unsigned *lat)
{
    const int lane = threadIdx.x & 63;
    float gv = g0;
    if (mode & 2) {
        float gb = g0;
        int s = 0, seg = 0;
        bool done = false;
        for (; seg < 4 && !done; ++seg) {
            const bool in = lane >= s;
            const int og = xw::ord(gb);
            const int dG = in ? xw::ord(gb + d) - og : 0;
            const int S = xw::prefix(dG);
            const float gnext = xw::inv(og + S), gown = xw::inv(og + S - dG);
            float lg = gown + d;
            if (maxg > 0.0f && lg > maxg) lg = maxg;
            const unsigned long long bad = __builtin_amdgcn_ballot_w64(in && __float_as_uint(lg) != __float_as_uint(gnext));
            const int k = bad ? (int)__builtin_ctzll(bad) : 64;
            if (in && lane <= k) gv = gown;
            if (k >= 63) { done = true; break; }
            gb = xw::lane_of(lg, k);
            s = k + 1;
        }
        if (lat) { lat[0] += (unsigned)(seg + (done ? 1 : 0)); lat[1] += done ? 0u : 1u; }
        if (done) return gv;
    }
    // systolic: every lane adds its predecessor's value, 63 times
    const float dsh = xw::shr1(d, 0.f);
    gv = g0;
    for (int it = 0; it < 63; ++it) {
        float ng = xw::shr1(gv, g0) + dsh;
        if (maxg > 0.0f && ng > maxg) ng = maxg;
        if (lane > 0) gv = ng;
    }
    return gv;
}

struct AgcWalk {
    static constexpr bool GUESS = false;
    struct Par {
        const AgcMap *pre;          // exclusive prefix maps of the scan's blocks (SCAN_TILE samples each)
        const float *state_in;      // [0] the gain carried into the call
        float *state_out;           // [0] the gain carried out of it, [1] guard flag
        float rate, ref, maxg;
    };
    __device__ static __forceinline__ float2 start(const Par &p, long long s)
    {
        const float g0 = p.state_in[0];
        return make_float2(s == 0 ? g0 : agc_apply(p.pre[s / SCAN_TILE], g0), 0.f);
    }
    __device__ static __forceinline__ void carry_out(const Par &p, float2 st) { p.state_out[0] = st.x; }
    __device__ static __forceinline__ int block(const Par &p, float2 x, float2, int cnt, float2 &st, float2 &out, int mode, unsigned *lat, float &)
    {
        const int lane = threadIdx.x & 63;
        const bool act = lane < cnt;
        const float g0 = st.x;
        AgcMap m = agc_identity();
        if (act) m = agc_sample_map(x.x, x.y, p.rate, p.ref, p.maxg);
        float gown = agc_apply(agc_wave_exclusive(m), g0);
        float yr, yi, d;
        int rounds = 0;
        for (;;) {
            ++rounds;
            yr = x.x * gown;
            yi = x.y * gown;
            d = p.rate * (p.ref - ::sqrtf(yr * yr + yi * yi));
            d = act ? d : 0.0f;
            const float g2 = agx_scan(d, g0, p.maxg, mode, lat);
            const bool same = __float_as_uint(g2) == __float_as_uint(gown);
            gown = g2;
            if (xw::all(same) || rounds >= xw::MAX_ROUNDS) break;
        }
        out = make_float2(yr, yi);
        float gn = gown + d;
        if (p.maxg > 0.0f && gn > p.maxg) gn = p.maxg;
        st.x = xw::lane_of(gn, cnt - 1);
        return rounds;
    }
};

__global__ void agc_exact_flag_kernel(const unsigned *cnt, float *state_out)
{
    if (cnt[0] != 0) state_out[1] = 1.0f;        // joints still open after the rounds: the serial kernel follows
}

__global__ void agc_serial_exact_kernel(const float2 *x, float2 *y, const float *state_in, float *state_out,
                                        float rate, float ref, float maxg, long long n)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    if (state_out[1] == 0.0f) return;
    float g = state_in[0];
    for (long long i = 0; i < n; ++i) {
        const float2 v = x[i];
        float yr, yi;
        agc_step_exact(v.x, v.y, g, rate, ref, maxg, yr, yi);
        y[i] = make_float2(yr, yi);
    }
    state_out[0] = g;
    state_out[1] = 2.0f;  // serial path taken
}

int AgcStage::run_exact(const float2 *in, float2 *out, size_t n, hipStream_t s, Profiler *prof)
{
    if (n == 0) return XRIT_OK;
    float *sin_ = state.as<float>() + 2 * cur;
    float *sout = state.as<float>() + 2 * (cur ^ 1);
    const int nb = scan_blocks((long long)n);
    // ranges: two walkers per SIMD on a large call, multiples of the scan's blocks (the start gains are the blocks' prefixes)
    size_t lw = (n + (size_t)ex_walkers - 1) / (size_t)ex_walkers;
    if (lw >= 8 * (size_t)AGC_EX_WARM) lw = (lw + 1) / 2;      // (twice the walkers where the warm-up stays a fifth of a range: 2.8 against 3.6 ms per C1 burst)
    if (lw < 4096) lw = 4096;
    lw = (lw + SCAN_TILE - 1) / SCAN_TILE * SCAN_TILE;
    const int W = (int)((n + lw - 1) / lw);
    XR_TRY(aggs.reserve((size_t)(nb + scan_blocks(nb) + 4) * sizeof(AgcMap)));
    XR_TRY(joints.reserve((size_t)3 * W * sizeof(float2) + ((n >> 6) + 2) * sizeof(float2) + xw::NCNT * sizeof(unsigned)));
    AgcScanF f{in, out, sin_, sout, rate, ref, maxg, (long long)n, 0};
    {
        ProfScope ps(prof, "agc_reduce", s);
        hipLaunchKernelGGL(agc_begin_kernel, dim3(1), dim3(1), 0, s, sout);
        hipLaunchKernelGGL(scan_reduce_kernel<AgcScanF>, dim3(nb), dim3(SCAN_BLOCK), 0, s, f, (long long)n, aggs.as<AgcMap>());
    }
    {
        ProfScope ps(prof, "agc_scan", s);
        scan_aggs_launch(f, aggs.as<AgcMap>(), nb, s);
    }
    {
        ProfScope ps(prof, "agc_exact", s);
        xw::KArgs<AgcWalk> K{};
        xw::Args &A = K.a;
        A.x = in; A.y = out;
        A.js = joints.as<float2>(); A.je = A.js + W; A.used = A.je + W;
        A.bs = A.used + W;
        A.cnt = reinterpret_cast<unsigned *>(A.bs + (n >> 6) + 2);
        ex_cnt = A.cnt;
        A.n = (long long)n; A.Lw = (int)lw; A.H = AGC_EX_WARM; A.W = W;
        A.mode = ex_mode; A.prio = 1;
        K.p = AgcWalk::Par{aggs.as<AgcMap>(), sin_, sout, rate, ref, maxg};
        hipLaunchKernelGGL(xw::zero_kernel<AgcWalk>, dim3(1), dim3(64), 0, s, A.cnt, xw::NCNT);
        hipLaunchKernelGGL(xw::main_kernel<AgcWalk>, dim3(W), dim3(64), 0, s, K);
        hipLaunchKernelGGL(xw::used_kernel<AgcWalk>, dim3(div_up((size_t)W, 256)), dim3(256), 0, s, A);
        if (W > 1) {
            for (int r = 0; r < AGC_EX_ROUNDS; ++r) hipLaunchKernelGGL(xw::fix_kernel<AgcWalk>, dim3(W - 1), dim3(64), 0, s, K, 0);
            hipLaunchKernelGGL(xw::zero_kernel<AgcWalk>, dim3(1), dim3(64), 0, s, A.cnt, 1);
            hipLaunchKernelGGL(xw::fix_kernel<AgcWalk>, dim3(W - 1), dim3(64), 0, s, K, 1);
            hipLaunchKernelGGL(agc_exact_flag_kernel, dim3(1), dim3(1), 0, s, A.cnt, sout);
        }
        hipLaunchKernelGGL(agc_serial_exact_kernel, dim3(1), dim3(1), 0, s, in, out, sin_, sout, rate, ref, maxg, (long long)n);
    }
    XR_HIP(hipGetLastError());
    cur ^= 1;
    return XRIT_OK;
}

// ---- fused path: the producer of `in` (the decimator) has left one composed map per run of 64 * PL samples --
// the outputs of one of its waves (fir.hip, AgcEpilogue).  Their exclusive prefixes give every run its start
// gain; here too a run is one wave (PL samples per lane), so the prefix inside a run is a shuffle scan, and the
// recurrence is replayed literally per lane.
struct AgcRunF {
    typedef AgcMap T;
    __device__ T identity() const { return agc_identity(); }
    __device__ T combine(const T &lo, const T &hi) const { return agc_compose(lo, hi); }
};

template <int PL>
__global__ void __launch_bounds__(256) agc_apply_runs_kernel(const float2 *__restrict__ x, float2 *__restrict__ y,
                                                             const AgcMap *__restrict__ pre_run,
                                                             const float *__restrict__ state_in,
                                                             float *__restrict__ state_out, float rate, float ref,
                                                             float maxg, long long n)
{
    const int lane = threadIdx.x & 63;
    const long long run = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long i0 = run * (64 * PL) + (long long)lane * PL;
    int cnt = 0;
    if (i0 < n) cnt = (int)((n - i0) < PL ? (n - i0) : PL);
    float2 v[PL];
#pragma unroll
    for (int k = 0; k < PL; ++k) v[k] = k < cnt ? x[i0 + k] : make_float2(0.f, 0.f);
    AgcMap m = agc_identity();
#pragma unroll
    for (int k = 0; k < PL; ++k)
        if (k < cnt) m = agc_compose(m, agc_sample_map(v[k].x, v[k].y, rate, ref, maxg));
    const AgcMap ex = agc_wave_exclusive(m);      // the lanes before this one, in order
    if (cnt == 0) return;
    float g = agc_apply(agc_compose(pre_run[run], ex), state_in[0]);
#pragma unroll
    for (int k = 0; k < PL; ++k) {
        if (k < cnt) {
            float yr, yi;
            agc_step(v[k].x, v[k].y, g, rate, ref, maxg, yr, yi);
            y[i0 + k] = make_float2(yr, yi);
        }
    }
    if (i0 + cnt == n) state_out[0] = g;
}

// Without a decimator in front nobody hands the run maps over: one read-only sweep composes them (a wave per
// run of 64 * PL samples), the matched filter then applies the gains in its window fill as in the fused path.
// (a wave takes AGC_RUNS_PER_WAVE consecutive runs, all their loads in flight before the first map is composed: one run per
// wave was a launch of 1.4 M waves of ~110 instructions each at the circuit rate -- 0.66 ms for 2.1 GB, 3.2 TB/s)
constexpr int AGC_RUNS_PER_WAVE = 4;
template <int PL>
__global__ void __launch_bounds__(256) agc_run_maps_kernel(const float2 *__restrict__ x, AgcMap *__restrict__ maps,
                                                           float *__restrict__ state_out, float rate, float ref,
                                                           float maxg, long long n)
{
    const int lane = threadIdx.x & 63;
    const long long run0 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * AGC_RUNS_PER_WAVE;
    float2 v[AGC_RUNS_PER_WAVE][PL];
#pragma unroll
    for (int r = 0; r < AGC_RUNS_PER_WAVE; ++r) {
        const long long i0 = (run0 + r) * (64 * PL) + (long long)lane * PL;
#pragma unroll
        for (int k = 0; k < PL; ++k) v[r][k] = x[min(i0 + k, n - 1)];
    }
    bool bad = false;
#pragma unroll
    for (int r = 0; r < AGC_RUNS_PER_WAVE; ++r) {
        const long long run = run0 + r;
        if (run * (64 * PL) >= n) break;                       // wave-uniform
        const long long i0 = run * (64 * PL) + (long long)lane * PL;
        int cnt = 0;
        if (i0 < n) cnt = (int)((n - i0) < PL ? (n - i0) : PL);
        AgcMap m = agc_identity();
#pragma unroll
        for (int k = 0; k < PL; ++k) {
            if (k < cnt) {
                const AgcMap e = agc_sample_map(v[r][k].x, v[r][k].y, rate, ref, maxg);
                bad |= !(e.a >= 0.0f);
                m = agc_compose(m, e);
            }
        }
        m = agc_wave_total(m);
        if (lane == 0) maps[run] = m;
    }
    if (bad) state_out[1] = 1.0f;
}

int AgcStage::fused_reduce(const float2 *in, size_t n, int per_lane, hipStream_t s, Profiler *prof)
{
    AgcEpilogue epi{};
    XR_TRY(fused_begin(n, per_lane, s, &epi));
    if (n == 0) return XRIT_OK;
    if (per_lane != 3) { set_error("AGC: unsupported run shape %d", per_lane); return XRIT_E_INVALID; }
    const size_t rl = (size_t)64 * per_lane;
    const unsigned nr = div_up(n, rl);
    ProfScope ps(prof, "agc_reduce", s);
    hipLaunchKernelGGL(agc_run_maps_kernel<3>, dim3(div_up((size_t)nr, 4 * AGC_RUNS_PER_WAVE)), dim3(256), 0, s, in, epi.maps, epi.state_out,
                       rate, ref, maxg, (long long)n);
    XR_HIP(hipGetLastError());
    return XRIT_OK;
}

int AgcStage::fused_begin(size_t n, int per_lane, hipStream_t s, AgcEpilogue *epi)
{
    const size_t rl = (size_t)64 * per_lane;
    const int nr = (int)((n + rl - 1) / rl);
    XR_TRY(aggs.reserve((size_t)(nr + scan_blocks(nr) + 8) * sizeof(AgcMap)));
    float *sout = state.as<float>() + 2 * (cur ^ 1);
    hipLaunchKernelGGL(agc_begin_kernel, dim3(1), dim3(1), 0, s, sout);
    epi->maps = aggs.as<AgcMap>();
    epi->state_out = sout;
    epi->rate = rate;
    epi->ref = ref;
    epi->maxg = maxg;
    return XRIT_OK;
}

int AgcStage::fused_finish(const float2 *in, float2 *out, size_t n, int per_lane, hipStream_t s, Profiler *prof)
{
    if (n == 0) return XRIT_OK;
    float *sin_ = state.as<float>() + 2 * cur;
    float *sout = state.as<float>() + 2 * (cur ^ 1);
    const size_t rl = (size_t)64 * per_lane;
    const int nr = (int)((n + rl - 1) / rl);
    {
        ProfScope ps(prof, "agc_scan", s);
        scan_aggs_launch(AgcRunF{}, aggs.as<AgcMap>(), nr, s);
    }
    {
        ProfScope ps(prof, "agc_apply", s);
        const dim3 grid(div_up((size_t)nr, 4));
#define XR_AGC_APPLY(PL)                                                                                        \
    hipLaunchKernelGGL(agc_apply_runs_kernel<PL>, grid, dim3(256), 0, s, in, out, aggs.as<AgcMap>(), sin_, sout, \
                       rate, ref, maxg, (long long)n)
        switch (per_lane) {
        case 1: XR_AGC_APPLY(1); break;
        case 2: XR_AGC_APPLY(2); break;
        case 3: XR_AGC_APPLY(3); break;
        case 4: XR_AGC_APPLY(4); break;
        case 5: XR_AGC_APPLY(5); break;
        default: set_error("AGC: unsupported run shape %d", per_lane); return XRIT_E_INVALID;
        }
#undef XR_AGC_APPLY
        hipLaunchKernelGGL(agc_serial_kernel, dim3(1), dim3(1), 0, s, in, out, sin_, sout, rate, ref, maxg,
                           (long long)n, 0);
    }
    XR_HIP(hipGetLastError());
    cur ^= 1;
    return XRIT_OK;
}

int AgcStage::fused_scan(const float2 *in, float2 *fallback, size_t n, int per_lane, hipStream_t s, Profiler *prof,
                         AgcFill *fill)
{
    float *sin_ = state.as<float>() + 2 * cur;
    float *sout = state.as<float>() + 2 * (cur ^ 1);
    const size_t rl = (size_t)64 * per_lane;
    const int nr = (int)((n + rl - 1) / rl);
    {
        ProfScope ps(prof, "agc_scan", s);
        if (nr > 0) scan_aggs_launch(AgcRunF{}, aggs.as<AgcMap>(), nr, s);
        hipLaunchKernelGGL(agc_serial_kernel, dim3(1), dim3(1), 0, s, in, fallback, sin_, sout, rate, ref, maxg,
                           (long long)n, 0);
    }
    XR_HIP(hipGetLastError());
    fill->x = in;
    fill->pre_run = aggs.as<AgcMap>();
    fill->state_in = sin_;
    fill->state_out = sout;
    fill->rate = rate;
    fill->ref = ref;
    fill->maxg = maxg;
    fill->per_lane = per_lane;
    cur ^= 1;
    return XRIT_OK;
}

int AgcStage::exact_counters(unsigned *out8, hipStream_t s)
{
    for (int i = 0; i < 8; ++i) out8[i] = 0;
    if (!ex_cnt) return XRIT_OK;
    XR_HIP(hipMemcpyAsync(out8, ex_cnt, 8 * sizeof(unsigned), hipMemcpyDeviceToHost, s));
    XR_HIP(hipStreamSynchronize(s));
    return XRIT_OK;
}

int AgcStage::gain(float *g, hipStream_t s)
{
    XR_HIP(hipMemcpyAsync(g, state.as<float>() + 2 * cur, sizeof(float), hipMemcpyDeviceToHost, s));
    XR_HIP(hipStreamSynchronize(s));
    return XRIT_OK;
}

int AgcStage::fallback_flag(float *flag, hipStream_t s)
{
    XR_HIP(hipMemcpyAsync(flag, state.as<float>() + 2 * cur + 1, sizeof(float), hipMemcpyDeviceToHost, s));
    XR_HIP(hipStreamSynchronize(s));
    return XRIT_OK;
}

}  // namespace xrit
