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
#include "scan.h"

namespace xrit {

constexpr int COSTAS_HEAD = 64;   // chains covered by the sequential coarse model

// ---------------------------------------------------------------- statistics
// stat[k] = sum z^2 over chain k (squaring removes the BPSK modulation)
__global__ void __launch_bounds__(256) costas_stat_kernel(const float2 *__restrict__ z, float2 *__restrict__ stat,
                                                          long long n, int L, int K)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long k = (long long)blockIdx.x * 4 + wave;
    if (k >= K) return;
    const long long base = k * L;
    float sr = 0.f, si = 0.f;
    for (int i = lane; i < L; i += 64) {
        long long j = base + i;
        if (j < n) {
            float2 v = z[j];
            sr += v.x * v.x - v.y * v.y;
            si += 2.0f * v.x * v.y;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        sr += __shfl_down(sr, off, 64);
        si += __shfl_down(si, off, 64);
    }
    if (lane == 0) stat[k] = make_float2(sr, si);
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
    double *th2;   // out: unwrapped 2*theta at chain centres
    __device__ T identity() const { return 0.0; }
    __device__ T combine(const T &lo, const T &hi) const { return lo + hi; }
    __device__ double diff(long long k) const
    {
        float2 a = stat[k];
        double cur = atan2((double)a.y, (double)a.x);
        if (k == 0) return cur;
        float2 b = stat[k - 1];
        double prev = atan2((double)b.y, (double)b.x);
        return wrap_pi_d(cur - prev);
    }
    __device__ T reduce_run(long long i0, int cnt) const
    {
        double s = 0;
        for (int k = 0; k < cnt; ++k) s += diff(i0 + k);
        return s;
    }
    __device__ void apply_run(long long i0, int cnt, const T &pre) const
    {
        double s = pre;
        for (int k = 0; k < cnt; ++k) {
            s += diff(i0 + k);
            th2[i0 + k] = s;
        }
    }
};

__global__ void costas_guess_kernel(const double *__restrict__ th2, float2 *__restrict__ S, int K, int L)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K || k == 0) return;
    // boundary k lies between chain centres k-1 and k
    double thb = 0.25 * (th2[k - 1] + th2[k]);
    int a = max(0, k - 2), b = min(K - 1, k + 1);
    double f = (b > a) ? 0.5 * (th2[b] - th2[a]) / ((double)(b - a) * L) : 0.0;
    S[k] = make_float2((float)wrap_pi_d(thb), (float)f);
}

// Chain-level model of the loop over the first chains of the call, where the
// true trajectory may still be acquiring: averaged detector = Im(e^{-2j phi} c_k)/2.
__global__ void costas_head_kernel(const float2 *__restrict__ stat, float2 *__restrict__ S,
                                   const float2 *__restrict__ state, int K, int L, CostasGains g)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    float2 s0 = state[0];
    S[0] = s0;
    double phi = s0.x, fr = s0.y;
    int H = min(COSTAS_HEAD, K - 1);
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

// --------------------------------------------------------------------- pass
// One lane = one chain.  FINAL: write the de-rotated samples, no tangent.
template <bool FINAL>
__global__ void __launch_bounds__(64) costas_pass_kernel(const float2 *__restrict__ z, float2 *__restrict__ y,
                                                         const float2 *__restrict__ S, float2 *__restrict__ E,
                                                         float4 *__restrict__ J, int *__restrict__ dirty,
                                                         float2 *__restrict__ state_out, long long n, int L, int K,
                                                         CostasGains g)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    if (!FINAL) {
        if (!dirty[k]) return;
        dirty[k] = 0;
    }
    const long long base = (long long)k * L;
    const int cnt = (int)min((long long)L, n - base);
    float2 s = S[k];
    float phase = s.x, freq = s.y;
    CostasTan t{1.f, 0.f, 0.f, 1.f};
    const float2 *zp = z + base;
    float2 *yp = y + base;
    int i = 0;
    for (; i + 2 <= cnt; i += 2) {
        float4 v = *reinterpret_cast<const float4 *>(zp + i);
        float y0r, y0i, y1r, y1i;
        costas_step<!FINAL>(v.x, v.y, phase, freq, g, y0r, y0i, t);
        costas_step<!FINAL>(v.z, v.w, phase, freq, g, y1r, y1i, t);
        if (FINAL) *reinterpret_cast<float4 *>(yp + i) = make_float4(y0r, y0i, y1r, y1i);
    }
    if (i < cnt) {
        float2 v = zp[i];
        float yr, yi;
        costas_step<!FINAL>(v.x, v.y, phase, freq, g, yr, yi, t);
        if (FINAL) yp[i] = make_float2(yr, yi);
    }
    if (FINAL) {
        if (k == K - 1) state_out[0] = make_float2(phase, freq);
    } else {
        E[k] = make_float2(phase, freq);
        J[k] = make_float4(t.pp, t.pf, t.fp, t.ff);
    }
}

// ------------------------------------------------------------ hand-off solve
// element k (0..K-2) describes boundary k+1: delta[k+1] = r_k + Jc_k delta[k]
struct CostasMap { float a11, a12, a21, a22, b1, b2; int par; };

struct CostasNewtonF {
    typedef CostasMap T;
    const float2 *S_ro;      // current start states
    float2 *S;               // same buffer, written in the final phase
    const float2 *E;
    const float4 *J;
    float2 *dlin;            // delta of the un-gated solve, per boundary index
    int *dirty;
    unsigned *counters;      // [0] changed, [1] not frozen, [2] max |r_phase| bits
    float trust_p, trust_f, tol_p, tol_f;
    int phase;               // 0: write dlin, 1: gated solve + update
    int K;

    __device__ T identity() const { return T{1.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0}; }
    __device__ T combine(const T &lo, const T &hi) const
    {
        T r;
        r.a11 = hi.a11 * lo.a11 + hi.a12 * lo.a21;
        r.a12 = hi.a11 * lo.a12 + hi.a12 * lo.a22;
        r.a21 = hi.a21 * lo.a11 + hi.a22 * lo.a21;
        r.a22 = hi.a21 * lo.a12 + hi.a22 * lo.a22;
        r.b1 = hi.a11 * lo.b1 + hi.a12 * lo.b2 + hi.b1;
        r.b2 = hi.a21 * lo.b1 + hi.a22 * lo.b2 + hi.b2;
        r.par = lo.par ^ hi.par;
        return r;
    }
    __device__ bool cut(long long k) const
    {
        if (phase == 0) return false;
        float2 d = dlin[k];
        return !(fabsf(d.x) <= trust_p) || !(fabsf(d.y) <= trust_f);
    }
    __device__ T element(long long k) const
    {
        float2 e = E[k], s = S_ro[k + 1];
        float rp = e.x - s.x, rf = e.y - s.y;
        float m = rintf(rp * (float)(1.0 / XR_PI_D));
        rp -= m * (float)XR_PI_D;
        float4 j = J[k];
        T t;
        if (cut(k)) { t.a11 = t.a12 = t.a21 = t.a22 = 0.f; }
        else { t.a11 = j.x; t.a12 = j.y; t.a21 = j.z; t.a22 = j.w; }
        t.b1 = rp; t.b2 = rf;
        t.par = ((int)m) & 1;
        return t;
    }
    __device__ T reduce_run(long long i0, int cnt) const
    {
        T m = identity();
        for (int k = 0; k < cnt; ++k) m = combine(m, element(i0 + k));
        return m;
    }
    __device__ void apply_run(long long i0, int cnt, const T &pre) const
    {
        // delta[0] = 0 -> delta at boundary i0 is the offset part of the prefix
        float dp = pre.b1, df = pre.b2;
        int par = pre.par;
        for (int q = 0; q < cnt; ++q) {
            long long k = i0 + q;
            T e = element(k);
            float jp = e.a11 * dp + e.a12 * df;
            float jf = e.a21 * dp + e.a22 * df;
            float ndp = e.b1 + jp, ndf = e.b2 + jf;
            if (phase == 0) {
                dlin[k + 1] = make_float2(ndp, ndf);
                if (k == 0) dlin[0] = make_float2(0.f, 0.f);
            } else {
                bool frozen = fabsf(ndp) <= tol_p && fabsf(ndf) <= tol_f && par == 0 && e.par == 0;
                if (!frozen) {
                    float2 ek = E[k];
                    float2 nw = make_float2(ek.x + (par ? (float)XR_PI_D : 0.f) + jp, ek.y + jf);
                    float2 old = S_ro[k + 1];
                    atomicAdd(&counters[1], 1u);
                    atomicMax(&counters[2], __float_as_uint(fabsf(e.b1)));
                    if (nw.x != old.x || nw.y != old.y) {
                        S[k + 1] = nw;
                        dirty[k + 1] = 1;
                        atomicAdd(&counters[0], 1u);
                    }
                }
            }
            dp = ndp; df = ndf;
            par ^= e.par;
        }
    }
};

__global__ void fill_int_kernel(int *p, int v, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

int CostasStage::init(float loop_bw, int chain_len, int max_passes_)
{
    gains = costas_gains(loop_bw);
    L = chain_len > 0 ? chain_len : 256;
    if (L & 1) ++L;
    max_passes = max_passes_ > 0 ? max_passes_ : 32;
    XR_TRY(state.reserve(2 * sizeof(float2)));
    XR_HIP(hipMemset(state.p, 0, 2 * sizeof(float2)));
    XR_TRY(counters.reserve(8 * sizeof(unsigned)));
    XR_HIP(hipHostMalloc((void **)&h_counters, 8 * sizeof(unsigned)));
    cur = 0;
    return XRIT_OK;
}

void CostasStage::release()
{
    state.release(); S.release(); E.release(); J.release(); stat.release(); dlin.release();
    work.release(); flags.release(); counters.release();
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

int CostasStage::run(const float2 *in, float2 *out, size_t n, hipStream_t s, Profiler *prof)
{
    passes = 0;
    unconverged = 0;
    max_residual = 0;
    if (n == 0) return XRIT_OK;
    const int K = (int)((n + (size_t)L - 1) / (size_t)L);
    const float2 *st_in = state.as<float2>() + cur;
    float2 *st_out = state.as<float2>() + (cur ^ 1);
    XR_TRY(S.reserve((size_t)K * sizeof(float2)));
    XR_TRY(E.reserve((size_t)K * sizeof(float2)));
    XR_TRY(J.reserve((size_t)K * sizeof(float4)));
    XR_TRY(stat.reserve((size_t)K * sizeof(float2)));
    XR_TRY(dlin.reserve((size_t)(K + 1) * sizeof(float2)));
    XR_TRY(flags.reserve((size_t)K * sizeof(int)));
    const int nbK = scan_blocks(K);
    const size_t agg_bytes = (((size_t)(nbK + 2) * sizeof(CostasMap)) + 15) & ~(size_t)15;
    XR_TRY(work.reserve(agg_bytes + (size_t)K * sizeof(double)));
    double *th2 = reinterpret_cast<double *>(work.as<char>() + agg_bytes);
    const unsigned gridK = div_up((size_t)K, 64);

    if (K > 1) {
        {
            ProfScope ps(prof, "costas_guess", s);
            hipLaunchKernelGGL(costas_stat_kernel, dim3(div_up((size_t)K, 4)), dim3(256), 0, s, in, stat.as<float2>(),
                               (long long)n, L, K);
            UnwrapF uf{stat.as<float2>(), th2};
            hipLaunchKernelGGL(scan_reduce_kernel<UnwrapF>, dim3(nbK), dim3(SCAN_BLOCK), 0, s, uf, (long long)K,
                               work.as<double>());
            hipLaunchKernelGGL(scan_aggs_kernel<UnwrapF>, dim3(1), dim3(SCAN_BLOCK), 0, s, uf, work.as<double>(), nbK);
            hipLaunchKernelGGL(scan_apply_kernel<UnwrapF>, dim3(nbK), dim3(SCAN_BLOCK), 0, s, uf, (long long)K,
                               work.as<double>());
            hipLaunchKernelGGL(costas_guess_kernel, dim3(div_up((size_t)K, 256)), dim3(256), 0, s, th2, S.as<float2>(), K, L);
            hipLaunchKernelGGL(costas_head_kernel, dim3(1), dim3(1), 0, s, stat.as<float2>(), S.as<float2>(), st_in, K,
                               L, gains);
            hipLaunchKernelGGL(fill_int_kernel, dim3(div_up((size_t)K, 256)), dim3(256), 0, s, flags.as<int>(), 1, K);
        }
        CostasNewtonF nf{S.as<float2>(), S.as<float2>(), E.as<float2>(), J.as<float4>(), dlin.as<float2>(),
                         flags.as<int>(), counters.as<unsigned>(), trust, trust / 256.0f, tol_phase, tol_freq, 0, K};
        const long long nel = K - 1;
        const int nbE = scan_blocks(nel);
        for (int p = 0; p < max_passes; ++p) {
            {
                ProfScope ps(prof, "costas_pass", s);
                hipLaunchKernelGGL(costas_pass_kernel<false>, dim3(gridK), dim3(64), 0, s, in, out, S.as<float2>(),
                                   E.as<float2>(), J.as<float4>(), flags.as<int>(), st_out, (long long)n, L, K, gains);
            }
            {
                ProfScope ps(prof, "costas_solve", s);
                XR_HIP(hipMemsetAsync(counters.p, 0, 8 * sizeof(unsigned), s));
                nf.phase = 0;
                hipLaunchKernelGGL(scan_reduce_kernel<CostasNewtonF>, dim3(nbE), dim3(SCAN_BLOCK), 0, s, nf, nel,
                                   work.as<CostasMap>());
                hipLaunchKernelGGL(scan_aggs_kernel<CostasNewtonF>, dim3(1), dim3(SCAN_BLOCK), 0, s, nf,
                                   work.as<CostasMap>(), nbE);
                hipLaunchKernelGGL(scan_apply_kernel<CostasNewtonF>, dim3(nbE), dim3(SCAN_BLOCK), 0, s, nf, nel,
                                   work.as<CostasMap>());
                nf.phase = 1;
                hipLaunchKernelGGL(scan_reduce_kernel<CostasNewtonF>, dim3(nbE), dim3(SCAN_BLOCK), 0, s, nf, nel,
                                   work.as<CostasMap>());
                hipLaunchKernelGGL(scan_aggs_kernel<CostasNewtonF>, dim3(1), dim3(SCAN_BLOCK), 0, s, nf,
                                   work.as<CostasMap>(), nbE);
                hipLaunchKernelGGL(scan_apply_kernel<CostasNewtonF>, dim3(nbE), dim3(SCAN_BLOCK), 0, s, nf, nel,
                                   work.as<CostasMap>());
            }
            XR_HIP(hipMemcpyAsync(h_counters, counters.p, 4 * sizeof(unsigned), hipMemcpyDeviceToHost, s));
            XR_HIP(hipStreamSynchronize(s));
            ++passes;
            unconverged = h_counters[1];
            uint32_t bits = h_counters[2];
            memcpy(&max_residual, &bits, sizeof(float));
            if (h_counters[0] == 0) { unconverged = 0; break; }
        }
    } else {
        XR_HIP(hipMemcpyAsync(S.p, st_in, sizeof(float2), hipMemcpyDeviceToDevice, s));
    }
    {
        ProfScope ps(prof, "costas_final", s);
        hipLaunchKernelGGL(costas_pass_kernel<true>, dim3(gridK), dim3(64), 0, s, in, out, S.as<float2>(),
                           E.as<float2>(), J.as<float4>(), flags.as<int>(), st_out, (long long)n, L, K, gains);
    }
    XR_HIP(hipGetLastError());
    cur ^= 1;
    return XRIT_OK;
}

}  // namespace xrit
