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

#include <cstdlib>
#include <new>
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
// AGC applied inside the window fill (APL > 0: the AGC's runs are 64 * APL samples).  The runs that overlap
// the block's window are dealt to the waves; a wave does for its run what agc_apply_runs_kernel does -- APL
// samples per lane, the gain at the lane's first sample from the run's prefix map and a DPP scan over the
// lanes, the recurrence replayed literally -- and drops the result into the LDS tile.
template <int APL>
__device__ __forceinline__ void fir_fill_through_agc(float2 *tile, const AgcFill &af, const float2 *__restrict__ hist,
                                                     int T, long long tile_start, int tile_len, long long n_in)
{
    constexpr int RL = 64 * APL;
    constexpr int MAXIT = 3;              // a window of <= 7 runs + two partial ones, four waves
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int wave = tid >> 6, lane = tid & 63, nwaves = nthr >> 6;
    const long long t_end = tile_start + tile_len < n_in ? tile_start + tile_len : n_in;
    const long long first_run = (tile_start > 0 ? tile_start : 0) / RL;
    float2 v[MAXIT][APL];
    int cnt[MAXIT];
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const long long rr = first_run + wave + (long long)it * nwaves;
        const long long i0 = rr * RL + (long long)lane * APL;
        cnt[it] = 0;
        if (rr * RL < t_end && i0 < n_in) cnt[it] = (int)((n_in - i0) < APL ? (n_in - i0) : APL);
#pragma unroll
        for (int k = 0; k < APL; ++k) v[it][k] = k < cnt[it] ? af.x[i0 + k] : make_float2(0.f, 0.f);
    }
    const float g0 = af.state_in[0];
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const long long rr = first_run + wave + (long long)it * nwaves;
        if (rr * RL >= t_end) break;                       // wave-uniform
        AgcMap m = agc_identity();
#pragma unroll
        for (int k = 0; k < APL; ++k)
            if (k < cnt[it]) m = agc_compose(m, agc_sample_map(v[it][k].x, v[it][k].y, af.rate, af.ref, af.maxg));
        const AgcMap ex = agc_wave_exclusive(m);
        float gg = agc_apply(agc_compose(af.pre_run[rr], ex), g0);
        const long long i0 = rr * RL + (long long)lane * APL;
#pragma unroll
        for (int k = 0; k < APL; ++k) {
            if (k < cnt[it]) {
                float yr, yi;
                agc_step(v[it][k].x, v[it][k].y, gg, af.rate, af.ref, af.maxg, yr, yi);
                const long long idx = i0 + k - tile_start;
                if (idx >= 0 && idx < tile_len) tile[idx] = make_float2(yr, yi);
                if (i0 + k == n_in - 1) af.state_out[0] = gg;         // the AGC's gain after the call
            }
        }
    }
    // in front of the stream: the carried history (already AGC output); behind its end: zeros
    for (int idx = tid; idx < tile_len; idx += nthr) {
        const long long j = tile_start + idx;
        if (j < 0) {
            const long long hj = (T - 1) + j;
            tile[idx] = hj >= 0 ? hist[hj] : make_float2(0.f, 0.f);
        } else if (j >= n_in) {
            tile[idx] = make_float2(0.f, 0.f);
        }
    }
}

// new history = last T-1 samples of (hist | in[0..n_in)), converted to float: done by the last workgroup of the
// filter kernel itself (one launch less per call than a kernel of its own)
template <int TYPE>
__device__ __forceinline__ void fir_leave_history(const void *__restrict__ in, const float2 *__restrict__ hist_old,
                                                  float2 *__restrict__ hist_new, int T, long long n_in)
{
    for (int i = threadIdx.x; i < T - 1; i += blockDim.x) {
        const long long j = n_in - (T - 1) + i;  // index into in, negative -> old history
        float2 v;
        if (j >= 0) v = SampleLoad<TYPE>::at(in, (size_t)j);
        else {
            const long long hj = (T - 1) + j;
            v = hj >= 0 ? hist_old[hj] : make_float2(0.f, 0.f);
        }
        hist_new[i] = v;
    }
}

// TS > 0 (decimation 1, TS taps -- the chain's 63-tap matched filter): the window walk is straight-line code with
// all taps in scalar registers.  The generic loop fetches RC rows of taps per eight samples through the scalar
// cache and waits for them and for its LDS reads three times per iteration: 36 % of the vector rate.
// MF (matrix-pipe experiment below): the window tile is skewed by 20 samples per 80, so that the 16 groups a wave's
// ds_read_b64 touches (80 samples = 160 dwords apart: banks 0 / 32 only, 8-way conflicts) sit 200 dwords apart
// (8 mod 64: two lanes per bank, the minimum for 128 dwords)
template <bool MF> __device__ __forceinline__ int fir_mf_pos(int q) { return MF ? q + 20 * (q / 80) : q; }

#ifndef XRIT_FE_PRIO
#define XRIT_FE_PRIO 1
#endif
template <int RC, bool PAD, int TYPE, int APL = 0, int TS = 0, int DS = 1, bool MF = false, bool EX = false>
__global__ void __launch_bounds__(256)
fir_decim_kernel(const void *__restrict__ in, const float2 *__restrict__ hist, float2 *__restrict__ out,
                 const float *__restrict__ g, int T, int D, int Wpad, long long n_out, long long n_in,
                 int tile_len, float2 *__restrict__ stat, int statL, AgcEpilogue agc, AgcFill af,
                 float2 *__restrict__ hist_new, const float *__restrict__ mfb = nullptr)
{
    // The filters' waves go in front of the clock recovery's walkers at instruction issue.  Round 3 measured this as a loss
    // (a burst waited for its relay's three dependent passes: an issue slot a walker lost was lost for good); since round 5
    // the walkers' latency is hidden behind two more bursts (demod.cpp) and what counts is how long the front end's stream is
    // busy per burst: 1.73-1.76 against 1.80-1.85 ms per C2 burst (same box, interleaved; the Costas passes at the same
    // priority as well: 1.79-1.82, so they stay where they are).
    // (asked for per launch -- FirStage::prio, carried in the upper half of statL --: the chain raises it for bursts whose clock
    // recovery walks overlapping blocks; beside the RELAY, whose passes a burst still waits for, it is the loss round 3 measured:
    // cfg.clock_exact = -3 2.60 against 2.02 ms per C2 burst)
    if (statL >> 16) __builtin_amdgcn_s_setprio(XRIT_FE_PRIO);
    statL &= 0xffff;
    // (AGC in the window fill: `in` is the serially produced AGC output if the guard has tripped, else see below)
    if (hist_new != nullptr && blockIdx.x == gridDim.x - 1 && (APL == 0 || af.state_out[1] != 0.0f))
        fir_leave_history<TYPE>(in, hist, hist_new, T, n_in);
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float2 *tile = reinterpret_cast<float2 *>(smem_raw);
    const int nthr = blockDim.x;
    const int tid = threadIdx.x;
    const long long OB = (long long)nthr * RC;
    const long long out_base = (long long)blockIdx.x * OB;
    const long long tile_start = out_base * D - (T - 1);

    if (APL > 0 && af.state_out[1] == 0.0f) {
        fir_fill_through_agc<(APL > 0 ? APL : 1)>(tile, af, hist, T, tile_start, tile_len, n_in);
    } else if (tile_start >= 0 && tile_start + tile_len <= n_in) {
        // interior block: no history, no end of input -> eight loads in flight per lane before the first store
        // (the guarded loop below waits for every single load: ~16 serial memory latencies per block)
        int idx = tid;
        if (TYPE == XRIT_SAMPLE_FLOATIQ && !PAD && !MF && (tile_start & 1) == 0 &&
            (reinterpret_cast<size_t>(in) & 15) == 0) {
            // cf32 input, even start: two samples per 16-byte load / LDS store
            const float4 *in4 = reinterpret_cast<const float4 *>(reinterpret_cast<const float2 *>(in) + tile_start);
            float4 *tile4 = reinterpret_cast<float4 *>(tile);
            const int pairs = tile_len >> 1;
            int p = tid;
#ifndef XRIT_NO_LDS_DIRECT
            // the tile is filled by loads that write LDS themselves (global_load_lds_dwordx4, gfx950): a wave instruction
            // drops 64 x 16 bytes at M0 + lane * 16 -- no round trip through registers, no ds_write, nothing for the wave
            // to wait for until the barrier.  Measured at C2 (make EXTRA=-DXRIT_NO_LDS_DIRECT for the register path below):
            // the decimator beside the relay 1.02 instead of 1.10 ms, the burst 2.34 instead of 2.38 ms; outputs bit-identical.
            for (; p - (tid & 63) < pairs; p += nthr) {
                if (p < pairs)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(in4 + p),
                                                     (__attribute__((address_space(3))) void *)(tile4 + (p - (tid & 63))), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
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
            tile[PAD ? q + q / D : fir_mf_pos<MF>(q)] = v[u];                                                 \
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
            int pos = PAD ? idx + idx / D : fir_mf_pos<MF>(idx);
            tile[pos] = v;
        }
    }
    __syncthreads();
    if (APL > 0 && hist_new != nullptr && blockIdx.x == gridDim.x - 1 && af.state_out[1] == 0.0f) {
        // the new history is AGC OUTPUT, which exists nowhere but in the window tiles: the last workgroup's tile
        // ends with it (D == 1: the tile starts T - 1 samples in front of the workgroup's first output)
        for (int i = tid; i < T - 1; i += nthr) hist_new[i] = tile[(int)(n_in - (T - 1) + i - tile_start)];
    }

    float2 acc[RC];
#pragma unroll
    for (int c = 0; c < RC; ++c) acc[c] = make_float2(0.f, 0.f);

    const int lane_base = PAD ? tid * RC * (D + 1) : tid * RC * D;
    const float2 *w = tile + lane_base;
    // PAD: window index i sits at i + i/D; walk it D samples at a time
    if (MF) {
#ifdef XRIT_EXPERIMENTS      // (make EXTRA=-DXRIT_EXPERIMENTS: not in the shipped library)
#include "../../experiments/csrc/fir_mfma_dec.inc"
#endif
    } else if (TS > 0) {
        // row 0 of g is the reversed filter, g[i] = h[TS-1-i] (zero behind it); output c takes sample i with tap
        // g[i - c].  Taps sit in scalar registers as aligned pairs and v_pk_fma_f32 broadcasts either half of a
        // pair through op_sel -- written out, because left to itself the compiler builds a (tap, tap) pair per use
        // and spills scalar registers over it (150 v_readlane + 190 s_nop next to the 315 multiply-adds).
        typedef float v2f __attribute__((ext_vector_type(2)));
        constexpr int NP = (TS + 1) / 2;
        const unsigned long long *gp = reinterpret_cast<const unsigned long long *>(g);
        unsigned long long tp[NP > 0 ? NP : 1];
#pragma unroll
        for (int k = 0; k < NP; ++k) tp[k] = gp[k];
        v2f a[RC];
#pragma unroll
        for (int c = 0; c < RC; ++c) a[c] = (v2f){0.f, 0.f};
#pragma unroll
        for (int i = 0; i < TS + (RC - 1) * DS; ++i) {
            const float2 xs = w[i];
            const v2f x = {xs.x, xs.y};
#pragma unroll
            for (int c = 0; c < RC; ++c) {
                const int j = i - c * DS;
                if (j >= 0 && j < TS) {
                    if (j & 1)
                        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(a[c]) : "s"(tp[j >> 1]), "v"(x));
                    else
                        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(a[c]) : "s"(tp[j >> 1]), "v"(x));
                }
            }
        }
#pragma unroll
        for (int c = 0; c < RC; ++c) acc[c] = make_float2(a[c].x, a[c].y);
    } else if (PAD && EX) {
        // cfg.front_exact = 2 on a skewed window (even strides: C5's 963 taps at d = 32), D a multiple of 4 (the launcher
        // checks): the window is walked a run of D samples at a time (the skew sits between runs), four samples per step --
        // window index mod 4 is the position in the run mod 4, so the four partial sums keep fixed registers, and (c D) mod 4 = 0:
        // the labels are the CPU chain's as they stand
        typedef float v2f __attribute__((ext_vector_type(2)));
        v2f a[RC][4];
#pragma unroll
        for (int c = 0; c < RC; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) a[c][j] = (v2f){0.f, 0.f};
        int i = 0, off = 0;
        while (i < Wpad) {
            const int run = min(D, Wpad - i);
            for (int k = 0; k < run; k += 4) {
                const float2 q0 = w[off + k], q1 = w[off + k + 1], q2 = w[off + k + 2], q3 = w[off + k + 3];
                const v2f x0 = {q0.x, q0.y}, x1 = {q1.x, q1.y}, x2 = {q2.x, q2.y}, x3 = {q3.x, q3.y};
#pragma unroll
                for (int c = 0; c < RC; ++c) {
                    const float *gc = g + c * Wpad + i + k;
                    const float t0 = gc[0], t1 = gc[1], t2 = gc[2], t3 = gc[3];
                    const v2f p0 = x0 * t0, p1 = x1 * t1, p2 = x2 * t2, p3 = x3 * t3;
                    a[c][0] = a[c][0] + p0;
                    a[c][1] = a[c][1] + p1;
                    a[c][2] = a[c][2] + p2;
                    a[c][3] = a[c][3] + p3;
                }
            }
            i += run;
            off += D + 1;
        }
#pragma unroll
        for (int c = 0; c < RC; ++c) {
            const v2f r = (a[c][0] + a[c][1]) + (a[c][2] + a[c][3]);
            acc[c] = make_float2(r.x, r.y);
        }
    } else if (PAD) {
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
    } else if (EX) {
        // cfg.front_exact = 2, windows that need no skew: the CPU chain's summation order with the window walk SHARED by the
        // lane's RC outputs.  Output c takes window sample i with tap index t = i - c D; the CPU chain adds the products into
        // four partial sums by t mod 4, each in order of t.  Here the sums are kept by i mod 4 -- the same four sets of terms in
        // the same order, labelled (c D) mod 4 further on -- and the labels are put right in the final (s0 + s1) + (s2 + s3).
        // Window samples outside an output's taps meet a zero tap: +0 added to a sum that started at +0 changes nothing (finite
        // input).  Separate v_pk_mul_f32 / v_pk_add_f32 (-ffp-contract=off), an LDS read feeds RC of each.
        typedef float v2f __attribute__((ext_vector_type(2)));
        v2f a[RC][4];
#pragma unroll
        for (int c = 0; c < RC; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) a[c][j] = (v2f){0.f, 0.f};
        for (int i = 0; i < Wpad; i += 4) {
            const float2 q0 = w[i], q1 = w[i + 1], q2 = w[i + 2], q3 = w[i + 3];
            const v2f x0 = {q0.x, q0.y}, x1 = {q1.x, q1.y}, x2 = {q2.x, q2.y}, x3 = {q3.x, q3.y};
#pragma unroll
            for (int c = 0; c < RC; ++c) {
                const float *gc = g + c * Wpad + i;
                const float t0 = gc[0], t1 = gc[1], t2 = gc[2], t3 = gc[3];
                const v2f p0 = x0 * t0, p1 = x1 * t1, p2 = x2 * t2, p3 = x3 * t3;
                a[c][0] = a[c][0] + p0;
                a[c][1] = a[c][1] + p1;
                a[c][2] = a[c][2] + p2;
                a[c][3] = a[c][3] + p3;
            }
        }
#pragma unroll
        for (int c = 0; c < RC; ++c) {
            const int sh = (c * D) & 3;           // wave-uniform
            v2f r;
            if (sh == 0) r = (a[c][0] + a[c][1]) + (a[c][2] + a[c][3]);
            else if (sh == 1) r = (a[c][1] + a[c][2]) + (a[c][3] + a[c][0]);
            else if (sh == 2) r = (a[c][2] + a[c][3]) + (a[c][0] + a[c][1]);
            else r = (a[c][3] + a[c][0]) + (a[c][1] + a[c][2]);
            acc[c] = make_float2(r.x, r.y);
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
    // The outputs leave through LDS: a lane holds RC consecutive outputs, so stores straight from registers are
    // 8-byte pieces RC * 8 bytes apart -- 64 separate write requests per wave instruction, every 128-byte line
    // written in 16 pieces; re-ordered, a wave instruction writes 512 consecutive bytes.
    {
        __syncthreads();                          // the window tile (or the wave maps in it) is dead
        float2 *ot = tile;
#pragma unroll
        for (int c = 0; c < RC; ++c) ot[tid * RC + c] = acc[c];
        __syncthreads();
        const long long left = n_out - out_base;
        const int cnt = left < OB ? (int)left : (int)OB;
        for (int k = tid; k < cnt; k += nthr) out[out_base + k] = ot[k];
    }
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
    if (stat != nullptr) {
        // sum z^2 per run of statL outputs, from the outputs staged in LDS above (statL divides the block's outputs).
        const float2 *ot = tile;
        if (statL == 8) {
            // Round 4: runs of 8 -- the statistic of the Costas loop's sub-block model (costas_refine_kernel), from which the
            // chain-level one is summed.  Thread t takes outputs t, t + nthr, ...: eight neighbouring lanes hold a run, three
            // row shifts add it up (fixed order: deterministic), the run's last lane stores it.
            for (int i = tid; i < (int)OB; i += nthr) {
                const long long o = out_base + i;
                float sr = 0.f, si = 0.f;
                if (o < n_out) {
                    const float2 z = ot[i];
                    sr = z.x * z.x - z.y * z.y;
                    si = 2.0f * z.x * z.y;
                }
                sr += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sr), 0x111, 0xf, 0xf, true));      // row_shr:1
                si += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(si), 0x111, 0xf, 0xf, true));
                sr += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sr), 0x112, 0xf, 0xf, true));      // row_shr:2
                si += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(si), 0x112, 0xf, 0xf, true));
                sr += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sr), 0x114, 0xf, 0xf, true));      // row_shr:4
                si += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(si), 0x114, 0xf, 0xf, true));
                if ((tid & 7) == 7 && o - 7 < n_out) stat[o >> 3] = make_float2(sr, si);
            }
        } else {
            // one wave per run (round robin), lane l adds outputs l, l + 64, ... of the run, then a fixed-order shuffle
            // tree over the lanes -- deterministic
            const int runs = (int)(OB / statL);
            const int wave = tid >> 6, lane = tid & 63, nwaves = nthr >> 6;
            for (int j = wave; j < runs; j += nwaves) {
                const long long first = out_base + (long long)j * statL;      // wave-uniform
                if (first >= n_out) break;
                float sr = 0.f, si = 0.f;
                for (int i = lane; i < statL; i += 64) {
                    if (first + i < n_out) {
                        const float2 z = ot[j * statL + i];
                        sr += z.x * z.x - z.y * z.y;
                        si += 2.0f * z.x * z.y;
                    }
                }
                for (int off = 32; off > 0; off >>= 1) {
                    sr += __shfl_down(sr, off, 64);
                    si += __shfl_down(si, off, 64);
                }
                if (lane == 0) stat[first / statL] = make_float2(sr, si);
            }
        }
    }
}

// ---- polyphase decimator for large decimations (DP = 16, 32 or 64 phases; BASELINE config 5: 963 taps, d = 32)
// y[m] = sum_k h[k] x[m*DP - k] with k = q*DP + p:  y[m] = sum_p sum_q h_p[q] x[(m - q)*DP - p].  The kernel
// above gives a lane consecutive OUTPUTS, which for DP = 32 means windows 256 B apart (every lane on the same
// LDS bank unless the tile is skewed) and no reuse between a lane's outputs.  Here a lane owns one PHASE p of a
// group of PR consecutive outputs: its NQ = ceil(T/DP) taps h_p[q] sit in registers for the whole kernel, the
// samples it needs, x[(m - q)*DP - p], are DP apart in time -- so the lanes of a group read CONSECUTIVE LDS
// addresses (no skew, no conflicts) -- and one LDS read feeds up to PR multiply-adds (the phase's own little FIR
// slides over the outputs).  The DP partial sums of an output are then added across the lanes with DPP row
// rotations.  64 / DP groups per wave, PR outputs per group.
#ifndef XRIT_POLY_THREADS
#define XRIT_POLY_THREADS 256
#endif
template <int DP, int PR, int NQ, int TYPE>
__global__ void __launch_bounds__(XRIT_POLY_THREADS)
fir_poly_kernel(const void *__restrict__ in, const float2 *__restrict__ hist, float2 *__restrict__ out,
                const float *__restrict__ hq /* [DP][NQ] */, int T, long long n_out, long long n_in, int tile_len,
                float2 *__restrict__ hist_new)
{
    // (no raised issue priority here, unlike fir_decim_kernel: C5's bursts have few walkers and a short front end, the walkers'
    // latency is not hidden there -- 1.27 against 1.07 ms per burst with it)
    if (hist_new != nullptr && blockIdx.x == 0) fir_leave_history<TYPE>(in, hist, hist_new, T, n_in);
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float2 *tile = reinterpret_cast<float2 *>(smem_raw);
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int OB = (nthr / DP) * PR;                  // outputs per block
    // Neighbouring tiles share (NQ-1)/(NQ-1+OB/...) of their window -- a third at d = 32.  Workgroups go to the
    // 8 XCDs round robin and every XCD has its own L2, so tile t = blockIdx would have that third fetched from
    // HBM twice; instead XCD x walks the contiguous tile range [x*per, (x+1)*per) and finds the overlap in its L2.
    const unsigned per = gridDim.x >> 3;              // the grid is a multiple of 8
    const long long tb = (long long)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
    const long long out_base = tb * OB;
    if (out_base >= n_out) return;
    // one sample more than the oldest any lane touches: an even start keeps the 16-byte loads aligned
    const long long tile_start = out_base * DP - (long long)(NQ * DP);

    if (TYPE == XRIT_SAMPLE_FLOATIQ && tile_start >= 0 && tile_start + tile_len <= n_in && (tile_start & 1) == 0 &&
        (reinterpret_cast<size_t>(in) & 15) == 0) {
        const float4 *in4 = reinterpret_cast<const float4 *>(reinterpret_cast<const float2 *>(in) + tile_start);
        float4 *tile4 = reinterpret_cast<float4 *>(tile);
        const int pairs = tile_len >> 1;
        // the whole window in one batch of loads per thread (256 threads): one memory latency per block
        constexpr int UF = (((NQ + (XRIT_POLY_THREADS / DP) * PR - 1) * DP + 2) / 2 + XRIT_POLY_THREADS - 1) / XRIT_POLY_THREADS;
#ifndef XRIT_NO_LDS_DIRECT
        // (loads that write LDS themselves, as in fir_decim_kernel)
        for (int p = tid; p - (tid & 63) < pairs; p += nthr)
            if (p < pairs)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(in4 + p),
                                                 (__attribute__((address_space(3))) void *)(tile4 + (p - (tid & 63))), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int p0 = pairs + tid; p0 < pairs; p0 += UF * nthr) {
#else
        for (int p0 = tid; p0 < pairs; p0 += UF * nthr) {
#endif
            float4 v[UF];
#pragma unroll
            for (int u = 0; u < UF; ++u) v[u] = in4[min(p0 + u * nthr, pairs - 1)];
#pragma unroll
            for (int u = 0; u < UF; ++u)
                if (p0 + u * nthr < pairs) tile4[p0 + u * nthr] = v[u];
        }
        if ((tile_len & 1) && tid == 0) tile[tile_len - 1] = SampleLoad<TYPE>::at(in, (size_t)(tile_start + tile_len - 1));
    } else {
        for (int idx = tid; idx < tile_len; idx += nthr) {
            const long long j = tile_start + idx;
            float2 v = make_float2(0.f, 0.f);
            if (j < 0) {
                const long long hj = (T - 1) + j;
                if (hj >= 0) v = hist[hj];
            } else if (j < n_in) {
                v = SampleLoad<TYPE>::at(in, (size_t)j);
            }
            tile[idx] = v;
        }
    }
    const int lane = tid & 63;
    const int p = lane % DP;                          // phase of this lane
    const int grp = tid / DP;                         // output group inside the block
    float h[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) h[q] = hq[p * NQ + q];
    lds_barrier();

    // output c of the group (m = out_base + grp*PR + c), tap q: sample index (m - q)*DP - p
    //   = tile_start + NQ*DP + (grp*PR + c - q)*DP - p, i.e. with u = c - q in [-(NQ-1), PR-1]:
    const float2 *w = tile + NQ * DP + grp * PR * DP - p;
    // (re, im) pairs as 2-vectors so that a multiply-add is one v_pk_fma_f32 (tap broadcast through op_sel)
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f accv[PR];
#pragma unroll
    for (int c = 0; c < PR; ++c) accv[c] = (v2f){0.f, 0.f};
#pragma unroll
    for (int u = -(NQ - 1); u < PR; ++u) {
        const float2 xs = w[u * DP];
        const v2f x = {xs.x, xs.y};
#pragma unroll
        for (int c = 0; c < PR; ++c) {
            const int q = c - u;
            if (q >= 0 && q < NQ) accv[c] = __builtin_elementwise_fma((v2f){h[q], h[q]}, x, accv[c]);
        }
    }
    float2 acc[PR];
#pragma unroll
    for (int c = 0; c < PR; ++c) acc[c] = make_float2(accv[c].x, accv[c].y);
    if (PR == 16) {
        // Add the DP phases of every output -- as a butterfly (round 5, late): every lane holds 16 partial sums and only ONE
        // lane has to end up with each output's total, so a step adds the partner lane's sums of the outputs THIS lane keeps
        // (lane ^ 8: outputs 0..7 or 8..15 by the lane's bit 3; then lane ^ 4, ^ 2, ^ 1), and lane c of a row ends with the
        // row's sum of output c: 8 + 4 + 2 + 1 adds (+ as many selects) per component instead of 4 x 16, and the lane that
        // stores an output holds it -- no chain of 15 selects in front of the store.  A tenth of the kernel's instructions.
        // The same pairs are added as by the rotations below (lane l with l + 8, the pair sums with those 4 away, ...: sums
        // of the same two words commute), so the outputs are the same words.
        const bool b3 = (lane & 8) != 0, b2 = (lane & 4) != 0, b1 = (lane & 2) != 0, b0 = (lane & 1) != 0;
        float2 m8[8], m4[4], m2[2];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float2 lo = acc[c], hi = acc[c + 8];
            const float xr = lo.x + agc_dpp<0x128>(0.f, lo.x), xi = lo.y + agc_dpp<0x128>(0.f, lo.y);       // row_ror:8 = lane ^ 8
            const float yr = hi.x + agc_dpp<0x128>(0.f, hi.x), yi = hi.y + agc_dpp<0x128>(0.f, hi.y);
            m8[c] = b3 ? make_float2(yr, yi) : make_float2(xr, xi);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            // lanes with bit 2 clear keep outputs c (partner 4 lanes up: row_ror:12), the others c + 4 (4 lanes down: row_ror:4)
            const float2 lo = m8[c], hi = m8[c + 4];
            const float xr = lo.x + agc_dpp<0x12C>(0.f, lo.x), xi = lo.y + agc_dpp<0x12C>(0.f, lo.y);
            const float yr = hi.x + agc_dpp<0x124>(0.f, hi.x), yi = hi.y + agc_dpp<0x124>(0.f, hi.y);
            m4[c] = b2 ? make_float2(yr, yi) : make_float2(xr, xi);
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float2 lo = m4[c], hi = m4[c + 2];
            const float xr = lo.x + agc_dpp<0x4E>(0.f, lo.x), xi = lo.y + agc_dpp<0x4E>(0.f, lo.y);         // quad_perm [2,3,0,1] = lane ^ 2
            const float yr = hi.x + agc_dpp<0x4E>(0.f, hi.x), yi = hi.y + agc_dpp<0x4E>(0.f, hi.y);
            m2[c] = b1 ? make_float2(yr, yi) : make_float2(xr, xi);
        }
        float sr, si;
        {
            const float2 lo = m2[0], hi = m2[1];
            const float xr = lo.x + agc_dpp<0xB1>(0.f, lo.x), xi = lo.y + agc_dpp<0xB1>(0.f, lo.y);         // quad_perm [1,0,3,2] = lane ^ 1
            const float yr = hi.x + agc_dpp<0xB1>(0.f, hi.x), yi = hi.y + agc_dpp<0xB1>(0.f, hi.y);
            sr = b0 ? yr : xr;
            si = b0 ? yi : xi;
        }
        // lane c of every row now holds the row's sum of output c; the rows of a group: rows 1 and 3 add the row below them,
        // then (64 phases) rows 2 and 3 the sum two rows below -- the order of the row broadcasts below
        if (DP >= 32) {
            const float tr = __shfl_up(sr, 16, 64), ti = __shfl_up(si, 16, 64);
            if (lane & 16) { sr += tr; si += ti; }
        }
        if (DP >= 64) {
            const float tr = __shfl_up(sr, 32, 64), ti = __shfl_up(si, 32, 64);
            if (lane & 32) { sr += tr; si += ti; }
        }
        // the LAST row of 16 lanes of every group holds the group's sums: its lane c stores output c
        const int pl = p - (DP - 16);
        if (pl >= 0) {
            const long long m = out_base + (long long)grp * PR + pl;
            if (m < n_out) out[m] = make_float2(sr, si);
        }
        return;
    }
    // add the DP phases of every output: rotations inside the rows of 16 lanes, then across rows
#pragma unroll
    for (int c = 0; c < PR; ++c) {
        float sr = acc[c].x, si = acc[c].y;
        sr += agc_dpp<0x128>(0.f, sr); si += agc_dpp<0x128>(0.f, si);      // row_ror:8
        sr += agc_dpp<0x124>(0.f, sr); si += agc_dpp<0x124>(0.f, si);      // row_ror:4
        sr += agc_dpp<0x122>(0.f, sr); si += agc_dpp<0x122>(0.f, si);      // row_ror:2
        sr += agc_dpp<0x121>(0.f, sr); si += agc_dpp<0x121>(0.f, si);      // row_ror:1
        if (DP >= 32) {
            // rows 1 and 3 take the sum of the row below them (lane 15 of it, broadcast): no LDS round trip
            sr += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sr), 0x142, 0xA, 0xF, false));
            si += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(si), 0x142, 0xA, 0xF, false));
        }
        if (DP >= 64) {
            // rows 2 and 3 take lane 31 (rows 0 + 1): row 3 then holds all four
            sr += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sr), 0x143, 0xC, 0xF, false));
            si += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(si), 0x143, 0xC, 0xF, false));
        }
        acc[c] = make_float2(sr, si);
    }
    // the LAST row of 16 lanes of every group holds the group's sums: its lane c stores output c
    const int pl = p - (DP - 16);
    if (pl >= 0 && pl < PR) {
        float2 v = acc[0];
#pragma unroll
        for (int c = 1; c < PR; ++c) v = (pl == c) ? acc[c] : v;
        const long long m = out_base + (long long)grp * PR + pl;
        if (m < n_out) out[m] = v;
    }
}


// ---- cfg.front_exact = 2: the filter summed the way the CPU chain sums it -------------------------------------------------
// FirFilter::Work, demodulator.cpp:138,148, as the test tier's CPU restatement (xo_fir_work) sums it: output m = sum over the time-ordered window
// w[t] = x[m D - (T - 1) + t] of rt[t] w[t] (rt: the taps reversed) in FOUR interleaved float32 partial sums -- sum j takes
// t = j mod 4, in order of t, every product rounded before it is added (no FMA) --, then (s0 + s1) + (s2 + s3).  Float
// addition does not associate, so this kernel does exactly that: a lane takes one output at a time and walks its window
// once with four accumulator pairs (v_pk_mul_f32 + v_pk_add_f32 per tap; the build has -ffp-contract=off).  Consecutive
// lanes take consecutive outputs (windows D samples apart in the LDS tile: 2 D dwords, conflict-free for odd D; even D: the
// tile is skewed by one sample per D); the stores are coalesced as they are.  Twice the vector work of the FMA kernels and
// no reuse of a window read between outputs: the price of the parity mode, not the shipped default.
template <int TYPE, bool PAD>
__global__ void __launch_bounds__(256)
fir_exact_kernel(const void *__restrict__ in, const float2 *__restrict__ hist, float2 *__restrict__ out,
                 const float *__restrict__ rt, int T, int D, long long n_out, long long n_in, int tile_len, int opt,
                 float2 *__restrict__ stat, float2 *__restrict__ hist_new)
{
    if (hist_new != nullptr && blockIdx.x == gridDim.x - 1) fir_leave_history<TYPE>(in, hist, hist_new, T, n_in);
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float2 *tile = reinterpret_cast<float2 *>(smem_raw);
    const int nthr = blockDim.x, tid = threadIdx.x;
    const long long OB = (long long)nthr * opt;
    const long long out_base = (long long)blockIdx.x * OB;
    const long long tile_start = out_base * D - (T - 1);
    for (int idx = tid; idx < tile_len; idx += nthr) {
        const long long j = tile_start + idx;
        float2 v = make_float2(0.f, 0.f);
        if (j < 0) {
            const long long hj = (T - 1) + j;
            if (hj >= 0) v = hist[hj];
        } else if (j < n_in) {
            v = SampleLoad<TYPE>::at(in, (size_t)j);
        }
        tile[PAD ? idx + idx / D : idx] = v;
    }
    __syncthreads();
    typedef float v2f __attribute__((ext_vector_type(2)));
    const int T4 = T & ~3;
    for (int k = 0; k < opt; ++k) {
        const int m = tid + k * nthr;                         // block-relative output
        const long long o = out_base + m;
        // window index m D + t sits at m D + t (+ (m D + t) / D = m + t / D when skewed)
        const float2 *w = tile + (PAD ? m * (D + 1) : m * D);
        v2f s0 = {0.f, 0.f}, s1 = {0.f, 0.f}, s2 = {0.f, 0.f}, s3 = {0.f, 0.f};
        int pos = 0, run = 0;                                 // run = t mod D (skewed tiles)
#define XR_EX_TAP(S, TT)                                                              \
        {                                                                             \
            const float2 xs = w[pos];                                                 \
            const v2f x = {xs.x, xs.y};                                               \
            const float tp = rt[TT];                                                  \
            const v2f pr = x * tp;                                                    \
            S = S + pr;                                                               \
            ++pos;                                                                    \
            if (PAD && ++run == D) { run = 0; ++pos; }                                \
        }
        for (int t = 0; t < T4; t += 4) {
            XR_EX_TAP(s0, t) XR_EX_TAP(s1, t + 1) XR_EX_TAP(s2, t + 2) XR_EX_TAP(s3, t + 3)
        }
        if (T4 < T) XR_EX_TAP(s0, T4)
        if (T4 + 1 < T) XR_EX_TAP(s1, T4 + 1)
        if (T4 + 2 < T) XR_EX_TAP(s2, T4 + 2)
#undef XR_EX_TAP
        const v2f r = (s0 + s1) + (s2 + s3);
        if (o < n_out) out[o] = make_float2(r.x, r.y);
        if (stat != nullptr) {
            // sum z^2 per run of 8 outputs (the Costas guess's statistic, as fir_decim_kernel leaves it): eight neighbouring
            // lanes hold a run
            float sr = 0.f, si = 0.f;
            if (o < n_out) { sr = r.x * r.x - r.y * r.y; si = 2.0f * r.x * r.y; }
            sr += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sr), 0x111, 0xf, 0xf, true));      // row_shr:1
            si += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(si), 0x111, 0xf, 0xf, true));
            sr += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sr), 0x112, 0xf, 0xf, true));      // row_shr:2
            si += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(si), 0x112, 0xf, 0xf, true));
            sr += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sr), 0x114, 0xf, 0xf, true));      // row_shr:4
            si += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(si), 0x114, 0xf, 0xf, true));
            if ((tid & 7) == 7 && o - 7 < n_out) stat[o >> 3] = make_float2(sr, si);
        }
    }
}

int FirStage::init(const float *taps, int ntaps, int decim)
{
    taps_host.assign(taps, taps + ntaps);
    // the exact-order twin (the chain chooses per call; a stage that is exact itself has none)
    if (twin) { twin->release(); delete twin; twin = nullptr; }
    last_twin = false;
    if (!exact) {
        twin = new (std::nothrow) FirStage();
        if (!twin) return XRIT_E_NOMEM;
        twin->exact = true;
        XR_TRY(twin->init(taps, ntaps, decim));
    }

    // diagnostic switches are read once, here: never on the launch path (a host that calls setenv races with getenv)
    no_static_dec = getenv("XRIT_NO_STATIC_DEC") != nullptr;
    no_static_mf = getenv("XRIT_NO_STATIC_MF") != nullptr;
#ifdef XRIT_EXPERIMENTS
    mfma_dec = getenv("XRIT_MFMA_DEC") != nullptr;
#endif
    T = ntaps;
    D = decim < 1 ? 1 : decim;
    // measured at C2: five outputs per lane halve the decimator's occupancy (52 KiB window) and lose 45 %
    if (D == 1) RC = 5;
    else RC = 3;
    // (cfg.front_exact = 2 at large decimations: one output per lane -- a window of 64 KiB then holds 128 outputs, four workgroups
    // per CU; with three per lane it held 192 on ONE wave per workgroup and two waves per CU: 4.0 ms per C5 burst)
    if (exact && D >= 16) RC = 1;
    // large decimations whose phases map onto lanes take the polyphase kernel
    poly = (D == 16 || D == 32 || D == 64) && (T + D - 1) / D <= POLY_NQ;
    if (exact) {
        // (cfg.front_exact = 2: fir_exact_kernel; the reversed taps, a tile of at most 64 KiB)
        poly = false;
        std::vector<float> r((size_t)T);
        for (int i = 0; i < T; ++i) r[(size_t)i] = taps[T - 1 - i];
        XR_TRY(rt.reserve(r.size() * sizeof(float)));
        XR_HIP(hipMemcpy(rt.p, r.data(), r.size() * sizeof(float), hipMemcpyHostToDevice));
        ex_pad = (D % 2) == 0;
        ex_threads = 256;
        ex_opt = D == 1 ? 4 : 2;
        for (;;) {
            const long long ob = (long long)ex_threads * ex_opt;
            const long long tl = (ob - 1) * D + T;
            const long long padded = ex_pad ? tl + tl / D + 2 : tl;
            if (padded * 8 <= 64 * 1024 || (ex_threads == 64 && ex_opt == 1)) {
                ex_tile_len = (int)tl;
                ex_lds = (size_t)padded * 8;
                break;
            }
            if (ex_opt > 1) --ex_opt; else ex_threads /= 2;
        }
        if (ex_lds > 160 * 1024) { set_error("FIR window of %d taps x decimation %d does not fit LDS", T, D); return XRIT_E_INVALID; }
    }
    if (poly) {
        threads = XRIT_POLY_THREADS;
        RC = 1;
        pad = false;
        const int OB = (threads / D) * POLY_PR;
        tile_len = (POLY_NQ + OB - 1) * D + 2;
        lds_bytes = (size_t)tile_len * sizeof(float2);
        std::vector<float> hq((size_t)D * POLY_NQ, 0.0f);
        for (int ph = 0; ph < D; ++ph)
            for (int q = 0; q < POLY_NQ; ++q)
                if (q * D + ph < T) hq[(size_t)ph * POLY_NQ + q] = taps[q * D + ph];
        XR_TRY(g.reserve(hq.size() * sizeof(float)));
        XR_HIP(hipMemcpy(g.p, hq.data(), hq.size() * sizeof(float), hipMemcpyHostToDevice));
        XR_TRY(hist[0].reserve((size_t)(T > 1 ? T - 1 : 1) * sizeof(float2)));
        XR_TRY(hist[1].reserve((size_t)(T > 1 ? T - 1 : 1) * sizeof(float2)));
        XR_HIP(hipMemset(hist[0].p, 0, hist[0].bytes));
        XR_HIP(hipMemset(hist[1].p, 0, hist[1].bytes));
        cur = 0;
        return XRIT_OK;
    }
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
    if (mfma_dec && T == 151 && D == 5) {
        // the Toeplitz operand of the matrix-pipe experiment: step q, lane l -> H[4 q + (l >> 4)][l & 15] = g[s - 5 j]
        const int KS = (15 * 5 + 151 + 3) / 4;
        std::vector<float> hb((size_t)KS * 64, 0.0f);
        for (int q = 0; q < KS; ++q)
            for (int l = 0; l < 64; ++l) {
                const int sp = 4 * q + (l >> 4) - 5 * (l & 15);
                if (sp >= 0 && sp < T) hb[(size_t)q * 64 + l] = rows[(size_t)sp];
            }
        XR_TRY(mfb.reserve(hb.size() * sizeof(float)));
        XR_HIP(hipMemcpy(mfb.p, hb.data(), hb.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    // block size: keep the LDS window under 64 KiB
    threads = 256;
    // (XRIT_DEC_THREADS, read here, when the stage is created: the decimator's workgroup size for A/B runs -- smaller
    // workgroups hold smaller windows, and more of them fit next to the relay's walkers)
#ifdef XRIT_EXPERIMENTS
    if (D > 1 && getenv("XRIT_DEC_THREADS")) {
        const int t = atoi(getenv("XRIT_DEC_THREADS"));
        if (t == 64 || t == 128 || t == 192 || t == 256) threads = t;
    }
#endif
    for (;;) {
        long long ob = (long long)threads * RC;
        long long tl = (ob - 1) * D + T + (Wpad - W) + 8;
        long long padded = pad ? tl + tl / D + 8 : tl;
        if (padded * 8 <= 64 * 1024 || threads == 64) {
            tile_len = (int)tl;
            lds_bytes = (size_t)padded * 8;
            break;
        }
        threads = threads == 192 ? 128 : threads / 2;
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

int FirStage::reset(hipStream_t s)
{
    if (twin) XR_TRY(twin->reset(s));
    last_twin = false;
    XR_HIP(hipMemsetAsync(hist[0].p, 0, hist[0].bytes, s));
    XR_HIP(hipMemsetAsync(hist[1].p, 0, hist[1].bytes, s));
    cur = 0;
    return XRIT_OK;
}

bool FirStage::agc_supported() const
{
    return !exact && !poly && !pad && threads % 64 == 0 && RC <= AGC_RUN_MAX_PER_LANE;      // one run per wave: 64 * RC outputs
}

bool FirStage::stat_supported(int statL) const
{
    // runs must not straddle blocks (the sums are taken from the block's outputs staged in LDS)
    if (exact) return statL == 8;
    if (poly) return false;
    return statL > 0 && !pad && threads % 64 == 0 && (threads * RC) % statL == 0;
}

int FirStage::set_exact(bool on)
{
    if (on == exact) return XRIT_OK;
    if (taps_host.empty()) { set_error("FIR: set_exact before init"); return XRIT_E_INVALID; }
    exact = on;
    const std::vector<float> t = taps_host;
    return init(t.data(), (int)t.size(), D);
}

void FirStage::release()
{
    if (twin) { twin->release(); delete twin; twin = nullptr; }
    rt.release();
    g.release();
    mfb.release();
    hist[0].release();
    hist[1].release();
}

template <int RC, bool PAD>
static int fir_launch_t(const FirStage &f, const void *in, int type, float2 *out, size_t n_out, size_t n_in,
                        hipStream_t s, float2 *stat, int statL, const AgcEpilogue &agc)
{
    const AgcFill af{};
    unsigned blocks = div_up(n_out, (size_t)f.threads * RC);
    const float2 *h = f.hist[f.cur].as<float2>();
    const float *g = f.g.as<float>();
#define XR_FIR_GO(TY)                                                                                          \
    hipLaunchKernelGGL((fir_decim_kernel<RC, PAD, TY>), dim3(blocks), dim3(f.threads), f.lds_bytes, s, in, h,  \
                       out, g, f.T, f.D, f.Wpad, (long long)n_out, (long long)n_in, f.tile_len, stat, statL | (f.prio << 16), agc, af, \
                       f.T > 1 ? f.hist[f.cur ^ 1].as<float2>() : (float2 *)nullptr)
#ifdef XRIT_EXPERIMENTS
    if (RC == 3 && !PAD && f.T == 151 && f.D == 5 && type == XRIT_SAMPLE_FLOATIQ && f.mfma_dec && f.threads == 256)
        hipLaunchKernelGGL((fir_decim_kernel<RC, PAD, XRIT_SAMPLE_FLOATIQ, 0, (RC == 3 && !PAD) ? 151 : 0, 5, (RC == 3 && !PAD)>), dim3(blocks),
                           dim3(f.threads), f.lds_bytes + (f.tile_len / 80 + 1) * 20 * 8, s, in, h, out, g, f.T, f.D, f.Wpad, (long long)n_out, (long long)n_in,
                           f.tile_len, stat, statL | (f.prio << 16), agc, af, f.T > 1 ? f.hist[f.cur ^ 1].as<float2>() : (float2 *)nullptr,
                           f.mfb.as<float>());
    else
#endif
    if (RC == 3 && !PAD && f.T == 151 && f.D == 5 && type == XRIT_SAMPLE_FLOATIQ && !f.no_static_dec)
        hipLaunchKernelGGL((fir_decim_kernel<RC, PAD, XRIT_SAMPLE_FLOATIQ, 0, (RC == 3 && !PAD) ? 151 : 0, 5>), dim3(blocks),
                           dim3(f.threads), f.lds_bytes, s, in, h, out, g, f.T, f.D, f.Wpad, (long long)n_out, (long long)n_in,
                           f.tile_len, stat, statL | (f.prio << 16), agc, af, f.T > 1 ? f.hist[f.cur ^ 1].as<float2>() : (float2 *)nullptr);
    else if (RC == 5 && !PAD && f.T == 63 && f.D == 1 && type == XRIT_SAMPLE_FLOATIQ && !f.no_static_mf)
        hipLaunchKernelGGL((fir_decim_kernel<RC, PAD, XRIT_SAMPLE_FLOATIQ, 0, (RC == 5 && !PAD) ? 63 : 0>), dim3(blocks),
                           dim3(f.threads), f.lds_bytes, s, in, h, out, g, f.T, f.D, f.Wpad, (long long)n_out, (long long)n_in,
                           f.tile_len, stat, statL | (f.prio << 16), agc, af, f.T > 1 ? f.hist[f.cur ^ 1].as<float2>() : (float2 *)nullptr);
    else if (type == XRIT_SAMPLE_FLOATIQ) XR_FIR_GO(XRIT_SAMPLE_FLOATIQ);
    else if (type == XRIT_SAMPLE_S16IQ) XR_FIR_GO(XRIT_SAMPLE_S16IQ);
    else XR_FIR_GO(XRIT_SAMPLE_S8IQ);
#undef XR_FIR_GO
    XR_HIP(hipGetLastError());
    return XRIT_OK;
}

bool FirStage::agc_fill_supported(int per_lane) const
{
    // the window must be covered by three rounds of the block's waves, one run each
    const long long runs = (tile_len + (long long)64 * per_lane - 1) / (64 * per_lane) + 1;
    return !exact && D == 1 && !pad && threads % 64 == 0 && per_lane == 3 && RC == 5 && runs <= 3 * (threads / 64);
}

// matched filter with the AGC applied in its window fill; `in` is the fallback stream (AgcFill)
static int fir_launch_agc_fill(FirStage &f, const float2 *in, float2 *out, size_t n, hipStream_t s, Profiler *prof,
                               float2 *stat, int statL, const AgcFill &af)
{
    const AgcEpilogue none{nullptr, nullptr, 0.f, 0.f, 0.f};
    const unsigned blocks = div_up(n, (size_t)f.threads * 5);
    {
        ProfScope ps(prof, "fir_rrc", s);
        if (f.T == 63 && !f.no_static_mf)
            hipLaunchKernelGGL((fir_decim_kernel<5, false, XRIT_SAMPLE_FLOATIQ, 3, 63>), dim3(blocks), dim3(f.threads), f.lds_bytes, s,
                               in, f.hist[f.cur].as<float2>(), out, f.g.as<float>(), f.T, f.D, f.Wpad, (long long)n,
                               (long long)n, f.tile_len, stat, statL | (f.prio << 16), none, af, f.hist[f.cur ^ 1].as<float2>());
        else
        hipLaunchKernelGGL((fir_decim_kernel<5, false, XRIT_SAMPLE_FLOATIQ, 3>), dim3(blocks), dim3(f.threads), f.lds_bytes, s,
                           in, f.hist[f.cur].as<float2>(), out, f.g.as<float>(), f.T, f.D, f.Wpad, (long long)n,
                           (long long)n, f.tile_len, stat, statL | (f.prio << 16), none, af, f.hist[f.cur ^ 1].as<float2>());
    }
    XR_HIP(hipGetLastError());
    f.cur ^= 1;
    return XRIT_OK;
}

int FirStage::run(const void *in, int type, float2 *out, size_t n_out, hipStream_t s, Profiler *prof, float2 *stat,
                  int statL, const AgcEpilogue *agc_in, const AgcFill *fill, bool use_exact)
{
    if (twin && T > 1 && use_exact != last_twin) {
        // the call changes sides: the T - 1 samples of history go along
        FirStage &from = last_twin ? *twin : *this, &to = last_twin ? *this : *twin;
        XR_HIP(hipMemcpyAsync(to.hist[to.cur].p, from.hist[from.cur].p, (size_t)(T - 1) * sizeof(float2), hipMemcpyDeviceToDevice, s));
    }
    if (twin) last_twin = use_exact;
    if (twin && use_exact) {
        twin->prio = prio;
        return twin->run(in, type, out, n_out, s, prof, stat, statL, agc_in, fill, false);
    }
    if (fill) {
        if (!agc_fill_supported(fill->per_lane) || type != XRIT_SAMPLE_FLOATIQ || T < 2) {
            set_error("FIR: this filter cannot apply the AGC in its window fill");
            return XRIT_E_INVALID;
        }
        if (stat && !stat_supported(statL)) stat = nullptr;
        if (n_out == 0) return XRIT_OK;
        return fir_launch_agc_fill(*this, reinterpret_cast<const float2 *>(in), out, n_out, s, prof, stat, statL, *fill);
    }
    if (stat && !stat_supported(statL)) stat = nullptr;
    if (exact) {
        if (agc_in) { set_error("FIR: the exact-order kernel has no AGC epilogue"); return XRIT_E_INVALID; }
        const size_t n_in = n_out * (size_t)D;
        if (n_out > 0) {
            ProfScope ps(prof, D > 1 ? "fir_decim" : "fir_rrc", s);
            const unsigned blocks = div_up(n_out, (size_t)ex_threads * ex_opt);
            const float2 *h = hist[cur].as<float2>();
            float2 *hn = T > 1 ? hist[cur ^ 1].as<float2>() : (float2 *)nullptr;
#define XR_EX_GO(TY, PD)                                                                                               \
    hipLaunchKernelGGL((fir_exact_kernel<TY, PD>), dim3(blocks), dim3(ex_threads), ex_lds, s, in, h, out, rt.as<float>(), T, D, \
                       (long long)n_out, (long long)n_in, ex_tile_len, ex_opt, stat, hn)
#define XR_EX_TY(PD)                                                          \
    do {                                                                      \
        if (type == XRIT_SAMPLE_FLOATIQ) XR_EX_GO(XRIT_SAMPLE_FLOATIQ, PD);   \
        else if (type == XRIT_SAMPLE_S16IQ) XR_EX_GO(XRIT_SAMPLE_S16IQ, PD);  \
        else XR_EX_GO(XRIT_SAMPLE_S8IQ, PD);                                  \
    } while (0)
            if (!pad && threads % 64 == 0) {
                // (windows that need no skew: the window walk shared by a lane's RC outputs, fir_decim_kernel<..., EX>)
                const AgcEpilogue none{nullptr, nullptr, 0.f, 0.f, 0.f};
                const AgcFill af{};
                const unsigned blk = div_up(n_out, (size_t)threads * RC);
#define XR_EXS_GO(RCV, TY)                                                                                                     \
    hipLaunchKernelGGL((fir_decim_kernel<RCV, false, TY, 0, 0, 1, false, true>), dim3(blk), dim3(threads), lds_bytes, s, in, h, out, \
                       g.as<float>(), T, D, Wpad, (long long)n_out, (long long)n_in, tile_len, stat, statL | (prio << 16), none, af, hn)
#define XR_EXS_TY(RCV)                                                          \
    do {                                                                        \
        if (type == XRIT_SAMPLE_FLOATIQ) XR_EXS_GO(RCV, XRIT_SAMPLE_FLOATIQ);   \
        else if (type == XRIT_SAMPLE_S16IQ) XR_EXS_GO(RCV, XRIT_SAMPLE_S16IQ);  \
        else XR_EXS_GO(RCV, XRIT_SAMPLE_S8IQ);                                  \
    } while (0)
                if (RC == 5) XR_EXS_TY(5);
                else XR_EXS_TY(3);
#undef XR_EXS_TY
#undef XR_EXS_GO
            }
            else if (pad && D % 4 == 0 && threads % 64 == 0 && (RC == 3 || RC == 1)) {
                // (skewed windows with D a multiple of 4 -- 16, 32, 64: the shared walk too)
                const AgcEpilogue none{nullptr, nullptr, 0.f, 0.f, 0.f};
                const AgcFill af{};
                const unsigned blk = div_up(n_out, (size_t)threads * RC);
#define XR_EXP_GO(RCV, TY)                                                                                                     \
    hipLaunchKernelGGL((fir_decim_kernel<RCV, true, TY, 0, 0, 1, false, true>), dim3(blk), dim3(threads), lds_bytes, s, in, h, out, \
                       g.as<float>(), T, D, Wpad, (long long)n_out, (long long)n_in, tile_len, (float2 *)nullptr, 0, none, af, hn)
#define XR_EXP_TY(RCV)                                                          \
    do {                                                                        \
        if (type == XRIT_SAMPLE_FLOATIQ) XR_EXP_GO(RCV, XRIT_SAMPLE_FLOATIQ);   \
        else if (type == XRIT_SAMPLE_S16IQ) XR_EXP_GO(RCV, XRIT_SAMPLE_S16IQ);  \
        else XR_EXP_GO(RCV, XRIT_SAMPLE_S8IQ);                                  \
    } while (0)
                if (RC == 1) XR_EXP_TY(1);
                else XR_EXP_TY(3);
#undef XR_EXP_TY
#undef XR_EXP_GO
            }
            else if (ex_pad) XR_EX_TY(true);
            else XR_EX_TY(false);
#undef XR_EX_TY
#undef XR_EX_GO
            XR_HIP(hipGetLastError());
        }
        if (T > 1 && n_in > 0) cur ^= 1;
        return XRIT_OK;
    }
    AgcEpilogue agc{nullptr, nullptr, 0.f, 0.f, 0.f};
    if (agc_in) {
        if (!agc_supported()) {
            set_error("FIR: this block shape cannot produce the AGC epilogue");
            return XRIT_E_INVALID;
        }
        agc = *agc_in;
    }
    size_t n_in = n_out * (size_t)D;
    if (n_out > 0 && poly) {
        ProfScope ps(prof, "fir_decim", s);
        const int OB = (threads / D) * POLY_PR;
        const unsigned blocks = (div_up(n_out, (size_t)OB) + 7u) & ~7u;          // XCD-contiguous tile ranges
        const float2 *h = hist[cur].as<float2>();
#define XR_POLY_GO(DPV, TY)                                                                                      \
    hipLaunchKernelGGL((fir_poly_kernel<DPV, POLY_PR, POLY_NQ, TY>), dim3(blocks), dim3(threads), lds_bytes, s, in, h,  \
                       out, g.as<float>(), T, (long long)n_out, (long long)n_in, tile_len,                               \
                       T > 1 ? hist[cur ^ 1].as<float2>() : (float2 *)nullptr)
#define XR_POLY_TY(DPV)                                                         \
    do {                                                                        \
        if (type == XRIT_SAMPLE_FLOATIQ) XR_POLY_GO(DPV, XRIT_SAMPLE_FLOATIQ);  \
        else if (type == XRIT_SAMPLE_S16IQ) XR_POLY_GO(DPV, XRIT_SAMPLE_S16IQ); \
        else XR_POLY_GO(DPV, XRIT_SAMPLE_S8IQ);                                 \
    } while (0)
        if (D == 16) XR_POLY_TY(16);
        else if (D == 32) XR_POLY_TY(32);
        else XR_POLY_TY(64);
#undef XR_POLY_TY
#undef XR_POLY_GO
        XR_HIP(hipGetLastError());
    } else if (n_out > 0) {
        ProfScope ps(prof, D > 1 ? "fir_decim" : "fir_rrc", s);
        if (RC == 5 && !pad) XR_TRY((fir_launch_t<5, false>(*this, in, type, out, n_out, n_in, s, stat, statL, agc)));
        else if (RC == 3 && !pad) XR_TRY((fir_launch_t<3, false>(*this, in, type, out, n_out, n_in, s, stat, statL, agc)));
        else XR_TRY((fir_launch_t<3, true>(*this, in, type, out, n_out, n_in, s, stat, statL, agc)));
    }
    if (T > 1 && n_in > 0) cur ^= 1;      // the filter kernel's last workgroup has left the new history
    return XRIT_OK;
}

}  // namespace xrit
