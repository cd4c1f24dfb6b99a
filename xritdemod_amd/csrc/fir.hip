// fir.hip -- decimating FIR, complex samples x real taps.
// Replaces SatHelper::FirFilter::Work as called at
// /root/reference/demodulator/src/demodulator.cpp:138 (decimating low-pass) and
// :148 (63-tap RRC, decimation 1):  y[m] = sum_k h[k] x[m*D - k], T-1 samples of
// history kept across calls.  The ingest conversion of demodulator.cpp:54-74
// (s16 -> /32768.f, s8 -> /128.f) is fused into the tile load.
//
// Layout: one workgroup stages a contiguous window of the input in LDS (coalesced
// 8-byte loads); every lane then produces RC consecutive outputs, walking its
// window once and feeding each sample to RC accumulators, so an LDS read is
// amortised over RC complex MACs.  The taps a lane needs at step i are the same
// for all lanes (wave-uniform), so they come through the scalar cache as SGPR
// operands.  RC*D odd gives a conflict-free ds_read_b64 lane stride; for even
// strides the LDS image is skewed by one sample per D (PAD).
#include "kernels.h"
#include "agc_wave.h"

namespace xrit {

template <int TYPE> struct SampleLoad;
template <> struct SampleLoad<XRIT_SAMPLE_FLOATIQ> {
    static __device__ __forceinline__ float2 at(const void *p, size_t j) { return reinterpret_cast<const float2 *>(p)[j]; }
};
template <> struct SampleLoad<XRIT_SAMPLE_S16IQ> {
    static __device__ __forceinline__ float2 at(const void *p, size_t j)
    {
        short2 v = reinterpret_cast<const short2 *>(p)[j];
        return make_float2(v.x / 32768.f, v.y / 32768.f);
    }
};
template <> struct SampleLoad<XRIT_SAMPLE_S8IQ> {
    static __device__ __forceinline__ float2 at(const void *p, size_t j)
    {
        char2 v = reinterpret_cast<const char2 *>(p)[j];
        return make_float2(v.x / 128.f, v.y / 128.f);
    }
};

// g: RC rows of Wpad taps, g[c][i] = h[T-1 + c*D - i] (0 outside), i.e. the tap
// that sample i of a lane's window contributes to the lane's c-th output.
// stat != nullptr (RRC stage of the chain): also leaves sum z^2 per run of statL outputs -- the
// statistic the Costas guess needs -- so that no separate sweep over the filtered stream is required.
// statL divides the outputs of a block, every run belongs to one block, and the partial sums are combined
// in a fixed order (deterministic).
template <int RC, bool PAD, int TYPE>
__global__ void __launch_bounds__(256)
fir_decim_kernel(const void *__restrict__ in, const float2 *__restrict__ hist, float2 *__restrict__ out,
                 const float *__restrict__ g, int T, int D, int Wpad, long long n_out, long long n_in,
                 int tile_len, float2 *__restrict__ stat, int statL, AgcEpilogue agc)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float2 *tile = reinterpret_cast<float2 *>(smem_raw);
    const int nthr = blockDim.x;
    const int tid = threadIdx.x;
    const long long OB = (long long)nthr * RC;
    const long long out_base = (long long)blockIdx.x * OB;
    const long long tile_start = out_base * D - (T - 1);

    if (tile_start >= 0 && tile_start + tile_len <= n_in) {
        // interior block: no history, no end of input -> eight loads in flight per lane before the first store
        // (the guarded loop below waits for every single load: ~16 serial memory latencies per block)
        int idx = tid;
        if (TYPE == XRIT_SAMPLE_FLOATIQ && !PAD && (tile_start & 1) == 0 &&
            (reinterpret_cast<size_t>(in) & 15) == 0) {
            // cf32 input, even start: two samples per 16-byte load / LDS store
            const float4 *in4 = reinterpret_cast<const float4 *>(reinterpret_cast<const float2 *>(in) + tile_start);
            float4 *tile4 = reinterpret_cast<float4 *>(tile);
            const int pairs = tile_len >> 1;
            int p = tid;
#define XR_TILE_BATCH4(U)                                                                   \
    for (; p + (U - 1) * nthr < pairs; p += U * nthr) {                                     \
        float4 v[U];                                                                        \
        _Pragma("unroll") for (int u = 0; u < U; ++u) v[u] = in4[p + u * nthr];             \
        _Pragma("unroll") for (int u = 0; u < U; ++u) tile4[p + u * nthr] = v[u];           \
    }
            XR_TILE_BATCH4(8)
            XR_TILE_BATCH4(4)
            XR_TILE_BATCH4(2)
            XR_TILE_BATCH4(1)
#undef XR_TILE_BATCH4
            idx = 2 * pairs + tid;          // an odd last sample is left to the 8-byte loop below
        }
#define XR_TILE_BATCH(U)                                                                                      \
    for (; idx + (U - 1) * nthr < tile_len; idx += U * nthr) {                                               \
        float2 v[U];                                                                                          \
        _Pragma("unroll") for (int u = 0; u < U; ++u)                                                         \
            v[u] = SampleLoad<TYPE>::at(in, (size_t)(tile_start + idx + u * nthr));                           \
        _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                       \
            const int q = idx + u * nthr;                                                                     \
            tile[PAD ? q + q / D : q] = v[u];                                                                 \
        }                                                                                                     \
    }
        XR_TILE_BATCH(8)
        XR_TILE_BATCH(4)
        XR_TILE_BATCH(2)
        XR_TILE_BATCH(1)
#undef XR_TILE_BATCH
    } else {
        for (int idx = tid; idx < tile_len; idx += nthr) {
            long long j = tile_start + idx;
            float2 v = make_float2(0.f, 0.f);
            if (j < 0) {
                long long hj = (T - 1) + j;
                if (hj >= 0) v = hist[hj];
            } else if (j < n_in) {
                v = SampleLoad<TYPE>::at(in, (size_t)j);
            }
            int pos = PAD ? idx + idx / D : idx;
            tile[pos] = v;
        }
    }
    __syncthreads();

    float2 acc[RC];
#pragma unroll
    for (int c = 0; c < RC; ++c) acc[c] = make_float2(0.f, 0.f);

    const int lane_base = PAD ? tid * RC * (D + 1) : tid * RC * D;
    const float2 *w = tile + lane_base;
    // PAD: window index i sits at i + i/D; walk it D samples at a time
    if (PAD) {
        int i = 0, off = 0;
        while (i < Wpad) {
            int run = min(D, Wpad - i);
            for (int k = 0; k < run; ++k) {
                float2 x = w[off + k];
#pragma unroll
                for (int c = 0; c < RC; ++c) {
                    float tp = g[c * Wpad + i + k];
                    acc[c].x = fmaf(tp, x.x, acc[c].x);
                    acc[c].y = fmaf(tp, x.y, acc[c].y);
                }
            }
            i += run;
            off += D + 1;
        }
    } else {
        for (int i = 0; i < Wpad; i += 4) {
            float2 x0 = w[i], x1 = w[i + 1], x2 = w[i + 2], x3 = w[i + 3];
#pragma unroll
            for (int c = 0; c < RC; ++c) {
                const float *gc = g + c * Wpad + i;
                float t0 = gc[0], t1 = gc[1], t2 = gc[2], t3 = gc[3];
                acc[c].x = fmaf(t0, x0.x, acc[c].x);
                acc[c].y = fmaf(t0, x0.y, acc[c].y);
                acc[c].x = fmaf(t1, x1.x, acc[c].x);
                acc[c].y = fmaf(t1, x1.y, acc[c].y);
                acc[c].x = fmaf(t2, x2.x, acc[c].x);
                acc[c].y = fmaf(t2, x2.y, acc[c].y);
                acc[c].x = fmaf(t3, x3.x, acc[c].x);
                acc[c].y = fmaf(t3, x3.y, acc[c].y);
            }
        }
    }
    const long long m0 = out_base + (long long)tid * RC;
#pragma unroll
    for (int c = 0; c < RC; ++c)
        if (m0 + c < n_out) out[m0 + c] = acc[c];
    if (!PAD && agc.maps != nullptr) {
        // AGC reduce sweep, fused: the composed gain map g -> min(a g + b, c) of every run of 64 * RC outputs,
        // i.e. of the outputs of one wave.  Composition is not commutative: a lane composes its own outputs in
        // order, agc_wave_total composes the lanes in order.
        AgcMap v = agc_identity();
        bool bad = false;
#pragma unroll
        for (int c = 0; c < RC; ++c) {
            if (m0 + c < n_out) {
                AgcMap e = agc_sample_map(acc[c].x, acc[c].y, agc.rate, agc.ref, agc.maxg);
                bad |= !(e.a >= 0.0f);
                v = agc_compose(v, e);
            }
        }
        if (bad) agc.state_out[1] = 1.0f;
        v = agc_wave_total(v);
        const int lane = tid & 63;
        if (lane == 0 && m0 < n_out) agc.maps[m0 / (64 * RC)] = v;
    }
    if (stat != nullptr && statL == 64 * RC) {
        // one run per wave: sum in registers, DPP row sums, four row totals -- fixed order, no LDS
        float sr = 0.f, si = 0.f;
#pragma unroll
        for (int c = 0; c < RC; ++c) {
            if (m0 + c < n_out) {
                const float zr = acc[c].x, zi = acc[c].y;
                sr += zr * zr - zi * zi;
                si += 2.0f * zr * zi;
            }
        }
        sr += agc_dpp<0x111>(0.f, sr); si += agc_dpp<0x111>(0.f, si);
        sr += agc_dpp<0x112>(0.f, sr); si += agc_dpp<0x112>(0.f, si);
        sr += agc_dpp<0x114>(0.f, sr); si += agc_dpp<0x114>(0.f, si);
        sr += agc_dpp<0x118>(0.f, sr); si += agc_dpp<0x118>(0.f, si);
        float tr = 0.f, ti = 0.f;
#pragma unroll
        for (int r = 15; r < 64; r += 16) {
            tr += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sr), r));
            ti += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(si), r));
        }
        if ((tid & 63) == 0 && m0 < n_out) stat[m0 / (64 * RC)] = make_float2(tr, ti);
    } else if (stat != nullptr) {
        // per-thread partial sums of z^2, split where the thread's outputs cross into the next run
        __syncthreads();                      // the window tile is dead: reuse it
        float2 *pa = tile, *pb = tile + nthr;
        const int o0 = tid * RC;              // first output of this thread inside the block
        const int first_run = o0 / statL;
        float2 a = make_float2(0.f, 0.f), b = make_float2(0.f, 0.f);
#pragma unroll
        for (int c = 0; c < RC; ++c) {
            if (m0 + c < n_out) {
                float zr = acc[c].x, zi = acc[c].y;
                float vr = zr * zr - zi * zi, vi = 2.0f * zr * zi;
                if ((o0 + c) / statL == first_run) { a.x += vr; a.y += vi; }
                else { b.x += vr; b.y += vi; }
            }
        }
        pa[tid] = a;
        pb[tid] = b;
        __syncthreads();
        // one wave per run (round robin), one partial per lane, fixed-order shuffle tree: deterministic
        const int runs = (nthr * RC) / statL;
        const int wave = tid >> 6, lane = tid & 63, nwaves = nthr >> 6;
        for (int j = wave; j < runs; j += nwaves) {
            const int lo = j * statL, hi = lo + statL;              // outputs [lo, hi) of the block
            const int t0 = lo / RC, t1 = min(nthr - 1, (hi - 1) / RC);
            float sr = 0.f, si = 0.f;
            for (int t = t0 + lane; t <= t1; t += 64) {
                const int fr = (t * RC) / statL;
                if (fr == j) { sr += pa[t].x; si += pa[t].y; }
                else if (fr + 1 == j) { sr += pb[t].x; si += pb[t].y; }
            }
            for (int off = 32; off > 0; off >>= 1) {
                sr += __shfl_down(sr, off, 64);
                si += __shfl_down(si, off, 64);
            }
            const long long run = out_base / statL + j;
            if (lane == 0 && run * statL < n_out) stat[run] = make_float2(sr, si);
        }
    }
}

// new history = last T-1 samples of (hist | in[0..n_in)), converted to float
template <int TYPE>
__global__ void fir_hist_kernel(const void *__restrict__ in, const float2 *__restrict__ hist_old,
                                float2 *__restrict__ hist_new, int T, long long n_in)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T - 1) return;
    long long j = n_in - (T - 1) + i;  // index into in, negative -> old history
    float2 v;
    if (j >= 0) v = SampleLoad<TYPE>::at(in, (size_t)j);
    else {
        long long hj = (T - 1) + j;
        v = hj >= 0 ? hist_old[hj] : make_float2(0.f, 0.f);
    }
    hist_new[i] = v;
}

int FirStage::init(const float *taps, int ntaps, int decim)
{
    T = ntaps;
    D = decim < 1 ? 1 : decim;
    // measured at C2: five outputs per lane halve the decimator's occupancy (52 KiB window) and lose 45 %
    if (D == 1) RC = 5;
    else RC = 3;
    pad = ((RC * D) % 2) == 0;
    W = T + (RC - 1) * D;
    Wpad = (W + 3) & ~3;
    std::vector<float> rows((size_t)RC * Wpad, 0.0f);
    for (int c = 0; c < RC; ++c)
        for (int i = 0; i < W; ++i) {
            int k = T - 1 + c * D - i;
            if (k >= 0 && k < T) rows[(size_t)c * Wpad + i] = taps[k];
        }
    XR_TRY(g.reserve(rows.size() * sizeof(float)));
    XR_HIP(hipMemcpy(g.p, rows.data(), rows.size() * sizeof(float), hipMemcpyHostToDevice));
    // block size: keep the LDS window under 64 KiB
    threads = 256;
    for (;;) {
        long long ob = (long long)threads * RC;
        long long tl = (ob - 1) * D + T + (Wpad - W) + 8;
        long long padded = pad ? tl + tl / D + 8 : tl;
        if (padded * 8 <= 64 * 1024 || threads == 64) {
            tile_len = (int)tl;
            lds_bytes = (size_t)padded * 8;
            break;
        }
        threads /= 2;
    }
    if (lds_bytes > 160 * 1024) {
        set_error("FIR window of %d taps x decimation %d does not fit LDS", T, D);
        return XRIT_E_INVALID;
    }
    XR_TRY(hist[0].reserve((size_t)(T > 1 ? T - 1 : 1) * sizeof(float2)));
    XR_TRY(hist[1].reserve((size_t)(T > 1 ? T - 1 : 1) * sizeof(float2)));
    XR_HIP(hipMemset(hist[0].p, 0, hist[0].bytes));
    XR_HIP(hipMemset(hist[1].p, 0, hist[1].bytes));
    cur = 0;
    return XRIT_OK;
}

bool FirStage::agc_supported() const
{
    return !pad && threads % 64 == 0 && RC <= AGC_RUN_MAX_PER_LANE;      // one run per wave: 64 * RC outputs
}

bool FirStage::stat_supported(int statL) const
{
    // runs must not straddle blocks, and the two partial arrays must fit in the window tile
    if (statL == 64 * RC && !pad && threads % 64 == 0) return true;      // one run per wave
    return statL > 0 && !pad && (threads * RC) % statL == 0 && statL >= RC && (size_t)2 * threads * sizeof(float2) <= lds_bytes;
}

void FirStage::release()
{
    g.release();
    hist[0].release();
    hist[1].release();
}

template <int RC, bool PAD>
static int fir_launch_t(const FirStage &f, const void *in, int type, float2 *out, size_t n_out, size_t n_in,
                        hipStream_t s, float2 *stat, int statL, const AgcEpilogue &agc)
{
    unsigned blocks = div_up(n_out, (size_t)f.threads * RC);
    const float2 *h = f.hist[f.cur].as<float2>();
    const float *g = f.g.as<float>();
#define XR_FIR_GO(TY)                                                                                          \
    hipLaunchKernelGGL((fir_decim_kernel<RC, PAD, TY>), dim3(blocks), dim3(f.threads), f.lds_bytes, s, in, h,  \
                       out, g, f.T, f.D, f.Wpad, (long long)n_out, (long long)n_in, f.tile_len, stat, statL, agc)
    if (type == XRIT_SAMPLE_FLOATIQ) XR_FIR_GO(XRIT_SAMPLE_FLOATIQ);
    else if (type == XRIT_SAMPLE_S16IQ) XR_FIR_GO(XRIT_SAMPLE_S16IQ);
    else XR_FIR_GO(XRIT_SAMPLE_S8IQ);
#undef XR_FIR_GO
    XR_HIP(hipGetLastError());
    return XRIT_OK;
}

int FirStage::run(const void *in, int type, float2 *out, size_t n_out, hipStream_t s, Profiler *prof, float2 *stat,
                  int statL, const AgcEpilogue *agc_in)
{
    if (stat && !stat_supported(statL)) stat = nullptr;
    AgcEpilogue agc{nullptr, nullptr, 0.f, 0.f, 0.f};
    if (agc_in) {
        if (!agc_supported()) {
            set_error("FIR: this block shape cannot produce the AGC epilogue");
            return XRIT_E_INVALID;
        }
        agc = *agc_in;
    }
    size_t n_in = n_out * (size_t)D;
    if (n_out > 0) {
        ProfScope ps(prof, D > 1 ? "fir_decim" : "fir_rrc", s);
        if (RC == 5 && !pad) XR_TRY((fir_launch_t<5, false>(*this, in, type, out, n_out, n_in, s, stat, statL, agc)));
        else if (RC == 3 && !pad) XR_TRY((fir_launch_t<3, false>(*this, in, type, out, n_out, n_in, s, stat, statL, agc)));
        else XR_TRY((fir_launch_t<3, true>(*this, in, type, out, n_out, n_in, s, stat, statL, agc)));
    }
    if (T > 1 && n_in > 0) {
        ProfScope ps(prof, "fir_hist", s);
        int nb = div_up((size_t)T - 1, 256);
        float2 *hn = hist[cur ^ 1].as<float2>();
        const float2 *ho = hist[cur].as<float2>();
        if (type == XRIT_SAMPLE_FLOATIQ)
            hipLaunchKernelGGL(fir_hist_kernel<XRIT_SAMPLE_FLOATIQ>, dim3(nb), dim3(256), 0, s, in, ho, hn, T, (long long)n_in);
        else if (type == XRIT_SAMPLE_S16IQ)
            hipLaunchKernelGGL(fir_hist_kernel<XRIT_SAMPLE_S16IQ>, dim3(nb), dim3(256), 0, s, in, ho, hn, T, (long long)n_in);
        else
            hipLaunchKernelGGL(fir_hist_kernel<XRIT_SAMPLE_S8IQ>, dim3(nb), dim3(256), 0, s, in, ho, hn, T, (long long)n_in);
        XR_HIP(hipGetLastError());
        cur ^= 1;
    }
    return XRIT_OK;
}

}  // namespace xrit
