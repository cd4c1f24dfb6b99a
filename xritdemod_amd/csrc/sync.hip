// sync.hip -- frame synchronisation front end of the decoder ("next" row).
// Replaces SatHelper::Correlator as /root/reference/decoder/src/newdecoder.cpp uses it before Viterbi:
// :145-151 addWord() of the two rate-1/2 encoded 64-bit sync words (as sent / inverted = 180 degrees), :218-245
// correlate(codedData, CODEDFRAMESIZE) then getCorrelationWordNumber / getHighestCorrelationPosition /
// getHighestCorrelation (accepted from MINCORRELATIONBITS = 46, decoder/src/parameters.h:31).
//
// A soft byte agrees with a word bit when  (byte >= 127 && bit == 0) || (byte < 127 && bit == 1),  bytes taken as
// unsigned: int8 0..126 is a one, 127 and every negative value a zero.  So a window of soft bytes is first turned
// into hard bits (16 bytes per thread and load, four at a time with integer SWAR), and the agreement of the 64
// bytes at offset i with a word is 64 - popcount(window_i ^ word): two 32-bit funnel shifts and two popcounts per
// offset and word, instead of 64 byte compares.  One workgroup per frame window; the per-word maximum keeps the FIRST offset that reaches it and the
// overall result the FIRST word, like the reference's strict '>' scans.
#include "kernels.h"

namespace xrit {

constexpr int SYNC_MAX_WORDS = 4;
constexpr int SYNC_THREADS = 256;

struct SyncWords { unsigned long long w[SYNC_MAX_WORDS]; int n; };

// key orders (correlation desc, position asc): larger key = better.  correlation <= 64, position < 2^20.
__device__ __forceinline__ unsigned sync_key(unsigned corr, unsigned pos) { return (corr << 20) | (0xFFFFFu - pos); }

// hard bits of four soft bytes (lowest address first -> most significant bit of the nibble)
__device__ __forceinline__ unsigned sync_nibble(unsigned x)
{
    // unsigned byte >= 127  <=>  bit 7 set, or the low seven bits are all ones
    const unsigned ge = (x | ((x & 0x7f7f7f7fu) + 0x01010101u)) & 0x80808080u;
    const unsigned one = (~ge & 0x80808080u) >> 7;        // 1 at bit 0 / 8 / 16 / 24
    return ((one * 0x08040201u) >> 24) & 0xFu;            // byte 0 -> bit 3 ... byte 3 -> bit 0, no carries
}

__global__ void __launch_bounds__(SYNC_THREADS) sync_correlate_kernel(const int8_t *__restrict__ data, unsigned frame,
                                                                      SyncWords words, xrit_sync_hit *__restrict__ hits)
{
    extern __shared__ unsigned bits[];                    // hard bits of the window, 32 per word, MSB = first byte
    __shared__ unsigned best[SYNC_MAX_WORDS][SYNC_THREADS / 64];
    const int8_t *win = data + (size_t)blockIdx.x * frame;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned n32 = (frame + 31) / 32 + 2;           // two words of zeros behind the window
    unsigned short *bits16 = reinterpret_cast<unsigned short *>(bits);
    const bool vec = ((reinterpret_cast<size_t>(win) | frame) & 15) == 0;
    // 16 soft bytes per thread and step -> 16 hard bits, stored so that a 32-bit LDS word reads MSB-first
    for (unsigned c = tid; c < (n32 * 32 + 15) / 16; c += SYNC_THREADS) {
        unsigned u16 = 0;
        const unsigned i0 = c * 16;
        if (vec && i0 + 16 <= frame) {
            const uint4 v = *reinterpret_cast<const uint4 *>(win + i0);
            u16 = (sync_nibble(v.x) << 12) | (sync_nibble(v.y) << 8) | (sync_nibble(v.z) << 4) | sync_nibble(v.w);
        } else {
            for (unsigned k = 0; k < 16; ++k) {
                const unsigned i = i0 + k;
                const unsigned one = (i < frame && (unsigned char)win[i] < 127) ? 1u : 0u;
                u16 |= one << (15 - k);
            }
        }
        bits16[c ^ 1u] = (unsigned short)u16;             // even chunk = high half of the little-endian word
    }
    __syncthreads();
    unsigned k[SYNC_MAX_WORDS];
    unsigned whi[SYNC_MAX_WORDS], wlo[SYNC_MAX_WORDS];
#pragma unroll
    for (int n = 0; n < SYNC_MAX_WORDS; ++n) {
        k[n] = 0;
        whi[n] = (unsigned)(words.w[n] >> 32);
        wlo[n] = (unsigned)(words.w[n] & 0xFFFFFFFFull);
    }
    const int max_search = (int)frame - 64;               // positions 0 .. frame-65, like the reference
    for (int i = tid; i < max_search; i += SYNC_THREADS) {
        const unsigned j = (unsigned)i >> 5, r = (unsigned)i & 31;
        const unsigned a = bits[j], b = bits[j + 1], c = bits[j + 2];
        const unsigned hi = __funnelshift_l(b, a, r), lo = __funnelshift_l(c, b, r);     // r = 0: a, b
#pragma unroll
        for (int n = 0; n < SYNC_MAX_WORDS; ++n) {
            if (n < words.n) {
                const unsigned cnt = 64u - (unsigned)__popc(hi ^ whi[n]) - (unsigned)__popc(lo ^ wlo[n]);
                k[n] = max(k[n], sync_key(cnt, (unsigned)i));
            }
        }
    }
#pragma unroll
    for (int n = 0; n < SYNC_MAX_WORDS; ++n) {
        unsigned v = k[n];
        for (int off = 32; off > 0; off >>= 1) v = max(v, (unsigned)__shfl_down((int)v, off, 64));
        if (lane == 0) best[n][wave] = v;
    }
    __syncthreads();
    if (tid == 0) {
        unsigned bc = 0, bp = 0, bw = 0;
        for (int n = 0; n < words.n; ++n) {
            unsigned v = best[n][0];
            for (int w = 1; w < SYNC_THREADS / 64; ++w) v = max(v, best[n][w]);
            // the reference starts every word at correlation 0 / position 0 and replaces on '>': a window with
            // no agreeing bit at all reports position 0
            const unsigned c = v >> 20, pos = c ? 0xFFFFFu - (v & 0xFFFFFu) : 0u;
            if (c > bc) { bc = c; bp = pos; bw = (unsigned)n; }
        }
        xrit_sync_hit h;
        h.word = bw;
        h.position = bp;
        h.correlation = bc;
        h.reserved = 0;
        hits[blockIdx.x] = h;
    }
}

// Frame alignment and phase fix (newdecoder.cpp:239-270): window f's frame starts at the correlation position --
// the reference shifts the chunk down by `pos` and reads `pos` more bytes -- and every byte is inverted when the
// 180-degree word won (PacketFixer::fixPacket(codedData, CODEDFRAMESIZE, DEG_180, false)).  Below the acceptance
// (:239-242 `continue`) or past the end of the buffer: no frame (zeros, valid = 0).  Four output bytes per lane
// from two aligned dwords and a funnel shift; bytes where the dword path would read or write outside.
__global__ void __launch_bounds__(256) sync_fix_kernel(const int8_t *__restrict__ data, size_t n, const xrit_sync_hit *__restrict__ hits,
                                                       unsigned frame, unsigned min_corr, int8_t *__restrict__ out,
                                                       unsigned char *__restrict__ valid)
{
    const unsigned f = blockIdx.y;
    const xrit_sync_hit h = hits[f];
    const size_t src0 = (size_t)f * frame + h.position;
    const bool ok = h.correlation >= min_corr && src0 + frame <= n;
    if (blockIdx.x == 0 && threadIdx.x == 0) valid[f] = ok ? 1 : 0;
    const unsigned inv = (h.word != 0) ? 0xFFFFFFFFu : 0u;
    int8_t *dst = out + (size_t)f * frame;
    const bool words_ok = ((reinterpret_cast<size_t>(dst) | reinterpret_cast<size_t>(data)) & 3) == 0;
    for (unsigned i = (blockIdx.x * 256u + threadIdx.x) * 4u; i < frame; i += gridDim.x * 1024u) {
        const size_t src = src0 + i;
        const size_t base = src & ~(size_t)3;
        if (ok && words_ok && i + 4 <= frame && base + 8 <= n) {
            const unsigned lo = *reinterpret_cast<const unsigned *>(data + base);
            const unsigned hi = *reinterpret_cast<const unsigned *>(data + base + 4);
            const unsigned v = __funnelshift_r(lo, hi, (unsigned)(src & 3) * 8u);
            *reinterpret_cast<unsigned *>(dst + i) = v ^ inv;
        } else {
            for (unsigned k = 0; k < 4 && i + k < frame; ++k)
                dst[i + k] = ok ? (int8_t)(data[src + k] ^ (int8_t)inv) : (int8_t)0;
        }
    }
}

int launch_sync_fix(const int8_t *data, size_t n, const xrit_sync_hit *hits, unsigned frame, unsigned min_corr,
                    int8_t *frames, unsigned char *valid, hipStream_t s)
{
    if (frame < 65 || frame > (1u << 20)) { set_error("sync: 65..2^20 bytes per frame"); return XRIT_E_INVALID; }
    const size_t nf = n / frame;
    if (nf == 0) return XRIT_OK;
    if (nf > 65535) { set_error("sync: at most 65535 frames per call"); return XRIT_E_INVALID; }
    const unsigned bx = (frame / 4 + 255) / 256 > 16 ? 16 : (frame / 4 + 255) / 256;
    hipLaunchKernelGGL(sync_fix_kernel, dim3(bx ? bx : 1, (unsigned)nf), dim3(256), 0, s, data, n, hits, frame, min_corr, frames,
                       valid);
    XR_HIP(hipGetLastError());
    return XRIT_OK;
}

int launch_sync_correlate(const int8_t *data, size_t n, const unsigned long long *words, int nwords, unsigned frame,
                          xrit_sync_hit *hits, hipStream_t s)
{
    if (nwords < 1 || nwords > SYNC_MAX_WORDS || frame < 65 || frame > (1u << 20)) {
        set_error("sync: 1..%d words and 65..2^20 bytes per frame", SYNC_MAX_WORDS);
        return XRIT_E_INVALID;
    }
    const size_t nf = n / frame;
    if (nf == 0) return XRIT_OK;
    SyncWords sw{};
    sw.n = nwords;
    for (int i = 0; i < nwords; ++i) sw.w[i] = words[i];
    const size_t lds = (((size_t)frame + 31) / 32 + 4) * sizeof(unsigned);
    hipLaunchKernelGGL(sync_correlate_kernel, dim3((unsigned)nf), dim3(SYNC_THREADS), lds, s, data, frame, sw, hits);
    XR_HIP(hipGetLastError());
    return XRIT_OK;
}

}  // namespace xrit
