// group.cpp -- one capture across the GPUs of a node, behind the C ABI.
//
// The reference demodulates one stream on one CPU thread (demodulator.cpp:170-175).  Here a burst is cut in `world`
// time slices, one per GPU.  What crosses GPUs (point to point over xGMI, RCCL ncclSend / ncclRecv, plus ONE
// all-gather of two integers per rank):
//   1. the halo: the last H input samples of rank g-1 go to rank g, which demodulates them first, from a cold
//      start, so that its filter histories are primed and its loops locked when its slice begins;
//   2. the last TAIL soft symbols of rank g-1, against which rank g settles the Costas pi ambiguity (sign of the
//      correlation) and the symbol that straddles the slice boundary (lag of the correlation peak);
//   3. (relative polarity, symbol count) of every rank: prefix product / prefix sum = absolute polarity and
//      output offset.
// Independent capture segments (BASELINE config 4) need none of this: every rank uses xrit_group_chain() as a
// plain chain handle.
//
// Two transports: RCCL (one process per GPU with a shared ncclUniqueId, or one process driving several GPUs
// through ncclCommInitAll) and an in-process fabric (ranks = threads of one process that exchange through
// device-to-device copies): the latter is what a single-GPU box can test.
#include "kernels.h"

#include <rccl/rccl.h>

#include <condition_variable>
#include <mutex>
#include <new>

using namespace xrit;

namespace {

constexpr int GROUP_TAIL = 256;     // soft symbols compared across a boundary
constexpr int GROUP_KEEP = 8;       // halo symbols kept in front of the slice output, for the straddling symbol

struct Transport {
    virtual ~Transport() {}
    // device buffers; the send and the receive of one exchange step are posted together
    virtual int exchange(const void *send, size_t send_bytes, int to, void *recv, size_t recv_bytes, int from,
                         hipStream_t s) = 0;
    virtual int allgather2(const long long mine[2], long long *all /* [2 * world] */, hipStream_t s) = 0;
    virtual int allreduce_max(double *v, hipStream_t s) = 0;
};

#define XR_NCCL(expr)                                                                   \
    do {                                                                                \
        ncclResult_t _r = (expr);                                                       \
        if (_r != ncclSuccess) {                                                        \
            set_error("%s failed: %s", #expr, ncclGetErrorString(_r));                  \
            return XRIT_E_HIP;                                                          \
        }                                                                               \
    } while (0)

struct RcclTransport : Transport {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    bool owns = true;
    DevBuf scratch;
    ~RcclTransport() override
    {
        if (comm && owns) (void)ncclCommDestroy(comm);
        scratch.release();
    }
    int exchange(const void *send, size_t send_bytes, int to, void *recv, size_t recv_bytes, int from,
                 hipStream_t s) override
    {
        XR_NCCL(ncclGroupStart());
        if (to >= 0 && send_bytes) XR_NCCL(ncclSend(send, send_bytes, ncclChar, to, comm, s));
        if (from >= 0 && recv_bytes) XR_NCCL(ncclRecv(recv, recv_bytes, ncclChar, from, comm, s));
        XR_NCCL(ncclGroupEnd());
        return XRIT_OK;
    }
    int allgather2(const long long mine[2], long long *all, hipStream_t s) override
    {
        XR_TRY(scratch.reserve((size_t)(2 * world + 2) * sizeof(long long)));
        long long *d = scratch.as<long long>();
        XR_HIP(hipMemcpyAsync(d, mine, 2 * sizeof(long long), hipMemcpyHostToDevice, s));
        XR_NCCL(ncclAllGather(d, d + 2, 2, ncclInt64, comm, s));
        XR_HIP(hipMemcpyAsync(all, d + 2, (size_t)2 * world * sizeof(long long), hipMemcpyDeviceToHost, s));
        XR_HIP(hipStreamSynchronize(s));
        return XRIT_OK;
    }
    int allreduce_max(double *v, hipStream_t s) override
    {
        XR_TRY(scratch.reserve((size_t)(2 * world + 2) * sizeof(long long)));
        double *d = scratch.as<double>();
        XR_HIP(hipMemcpyAsync(d, v, sizeof(double), hipMemcpyHostToDevice, s));
        XR_NCCL(ncclAllReduce(d, d, 1, ncclDouble, ncclMax, comm, s));
        XR_HIP(hipMemcpyAsync(v, d, sizeof(double), hipMemcpyDeviceToHost, s));
        XR_HIP(hipStreamSynchronize(s));
        return XRIT_OK;
    }
};

}  // namespace

// ranks = threads of one process; a mailbox per directed neighbour pair
struct xrit_local_fabric {
    int world = 1;
    std::mutex m;
    std::condition_variable cv;
    struct Slot { const void *src = nullptr; size_t bytes = 0; int src_dev = 0; bool full = false, taken = false; };
    std::vector<Slot> box;              // box[to]: what rank to-1 offers to rank `to`
    std::vector<long long> gathered;    // 2 * world
    int arrived = 0, generation = 0;
    std::vector<double> red;
};

namespace {

struct LocalTransport : Transport {
    xrit_local_fabric *f = nullptr;
    int rank = 0, device = 0;
    int exchange(const void *send, size_t send_bytes, int to, void *recv, size_t recv_bytes, int from,
                 hipStream_t s) override
    {
        // whatever produced `send` must be finished before the neighbour copies from it
        XR_HIP(hipStreamSynchronize(s));
        std::unique_lock<std::mutex> lk(f->m);
        if (to >= 0 && send_bytes) {
            auto &b = f->box[to];
            f->cv.wait(lk, [&] { return !b.full; });
            b.src = send; b.bytes = send_bytes; b.src_dev = device; b.full = true; b.taken = false;
            f->cv.notify_all();
        }
        if (from >= 0 && recv_bytes) {
            auto &b = f->box[rank];
            f->cv.wait(lk, [&] { return b.full && !b.taken; });
            if (b.bytes != recv_bytes) { set_error("fabric: %zu bytes offered, %zu expected", b.bytes, recv_bytes); return XRIT_E_INVALID; }
            const void *src = b.src;
            const int src_dev = b.src_dev;
            lk.unlock();
            hipError_t e = src_dev == device ? hipMemcpy(recv, src, recv_bytes, hipMemcpyDeviceToDevice)
                                             : hipMemcpyPeer(recv, device, src, src_dev, recv_bytes);
            lk.lock();
            b.taken = true;
            f->cv.notify_all();
            if (e != hipSuccess) { set_error("fabric copy failed: %s", hipGetErrorString(e)); return XRIT_E_HIP; }
        }
        if (to >= 0 && send_bytes) {
            // the buffer stays ours until the neighbour has copied it
            auto &b = f->box[to];
            f->cv.wait(lk, [&] { return b.taken; });
            b.full = false;
            f->cv.notify_all();
        }
        return XRIT_OK;
    }
    int rendezvous(std::unique_lock<std::mutex> &lk)
    {
        const int gen = f->generation;
        if (++f->arrived == f->world) {
            f->arrived = 0;
            ++f->generation;
            f->cv.notify_all();
        } else {
            f->cv.wait(lk, [&] { return f->generation != gen; });
        }
        return XRIT_OK;
    }
    int allgather2(const long long mine[2], long long *all, hipStream_t) override
    {
        std::unique_lock<std::mutex> lk(f->m);
        f->gathered[2 * rank] = mine[0];
        f->gathered[2 * rank + 1] = mine[1];
        rendezvous(lk);
        for (int i = 0; i < 2 * f->world; ++i) all[i] = f->gathered[i];
        rendezvous(lk);          // nobody overwrites before everybody has read
        return XRIT_OK;
    }
    int allreduce_max(double *v, hipStream_t) override
    {
        std::unique_lock<std::mutex> lk(f->m);
        f->red[rank] = *v;
        rendezvous(lk);
        double m = f->red[0];
        for (int i = 1; i < f->world; ++i) m = f->red[i] > m ? f->red[i] : m;
        *v = m;
        rendezvous(lk);
        return XRIT_OK;
    }
};

}  // namespace

struct xrit_group {
    xrit_demod *chain = nullptr;
    Transport *tr = nullptr;
    int rank = 0, world = 1, device = 0;
    size_t halo = 0;            // input samples taken over from the previous rank
    DevBuf halo_in, halo_syms, soft_int, tail_out, tail_in, host_in, host_out;
    std::vector<float> h_halo_syms, h_head, h_prev_tail;
};

namespace {

size_t group_halo_samples(const xrit_demod_config &cfg, float sps, int lpf_taps, int warm_symbols)
{
    // FIR histories + loop warm-up (the clock recovery is the slow one: ~13 time constants of 1800 symbols for
    // 1e-4-level agreement with the uninterrupted stream) + interpolator look-ahead, in whole decimation periods
    const size_t circuit = (size_t)((cfg.rrc_taps | 1) - 1) + (size_t)(warm_symbols * (double)sps) + 8 + 24;
    size_t h = (cfg.decimation > 1 ? (size_t)(lpf_taps - 1) : 0) + (size_t)cfg.decimation * circuit;
    return h - h % cfg.decimation;
}

__global__ void group_emit_kernel(const float *__restrict__ in, float *__restrict__ out, size_t n, float pol)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = pol * in[i];
}

// dist.py split_align: (relative polarity, lag) from the last TAIL symbols of the previous rank against
// [kept halo symbols | first KEEP slice symbols]
void group_align(const std::vector<float> &prev_tail, const std::vector<float> &halo_syms, const std::vector<float> &head,
                 int *pol, int *lag)
{
    *pol = 1;
    *lag = 0;
    if (prev_tail.empty()) return;
    std::vector<float> seq(halo_syms);
    seq.insert(seq.end(), head.begin(), head.end());
    const long nh = (long)halo_syms.size(), m = (long)prev_tail.size();
    double best = 0.0;
    for (long lg = -GROUP_KEEP + 1; lg < GROUP_KEEP; ++lg) {
        const long end = nh + lg;
        if (end - m < 0 || end > (long)seq.size()) continue;
        double c = 0;
        for (long i = 0; i < m; ++i) c += (double)prev_tail[i] * (double)seq[end - m + i];
        if (fabs(c) > best) { best = fabs(c); *lag = (int)lg; *pol = c >= 0 ? 1 : -1; }
    }
}

int group_finish_create(xrit_group *g, const xrit_demod_config *cfg)
{
    xrit_demod_config c = *cfg;
    XR_TRY(xrit_demod_create(&c, &g->chain));
    g->device = c.device;
    g->halo = g->world > 1 ? group_halo_samples(c, xrit_demod_sps(g->chain), xrit_demod_decimator_ntaps(g->chain), 24576) : 0;
    return XRIT_OK;
}

}  // namespace

extern "C" {

int xrit_group_unique_id(void *id128)
{
    if (!id128) { set_error("null argument"); return XRIT_E_INVALID; }
    static_assert(sizeof(ncclUniqueId) <= XRIT_GROUP_ID_BYTES, "ncclUniqueId fits the ABI's id buffer");
    ncclUniqueId id;
    XR_NCCL(ncclGetUniqueId(&id));
    memset(id128, 0, XRIT_GROUP_ID_BYTES);
    memcpy(id128, &id, sizeof id);
    return XRIT_OK;
}

int xrit_group_create(const xrit_demod_config *cfg, int rank, int world, const void *id128, xrit_group **out)
{
    if (!cfg || !out || !id128 || world < 1 || rank < 0 || rank >= world) { set_error("bad argument"); return XRIT_E_INVALID; }
    *out = nullptr;
    xrit_group *g = new (std::nothrow) xrit_group();
    if (!g) return XRIT_E_NOMEM;
    g->rank = rank;
    g->world = world;
    int rc = group_finish_create(g, cfg);
    if (rc == XRIT_OK) {
        RcclTransport *t = new (std::nothrow) RcclTransport();
        if (!t) rc = XRIT_E_NOMEM;
        else {
            g->tr = t;
            t->rank = rank;
            t->world = world;
            ncclUniqueId id;
            memcpy(&id, id128, sizeof id);
            if (hipSetDevice(g->device) != hipSuccess) { set_error("hipSetDevice failed"); rc = XRIT_E_HIP; }
            else {
                ncclResult_t r = ncclCommInitRank(&t->comm, world, id, rank);
                if (r != ncclSuccess) { set_error("ncclCommInitRank failed: %s", ncclGetErrorString(r)); rc = XRIT_E_HIP; }
            }
        }
    }
    if (rc != XRIT_OK) { xrit_group_destroy(g); return rc; }
    *out = g;
    return XRIT_OK;
}

int xrit_group_create_all(const xrit_demod_config *cfg, const int *devices, int world, xrit_group **out)
{
    if (!cfg || !out || !devices || world < 1) { set_error("bad argument"); return XRIT_E_INVALID; }
    for (int i = 0; i < world; ++i) out[i] = nullptr;
    std::vector<ncclComm_t> comms((size_t)world);
    XR_NCCL(ncclCommInitAll(comms.data(), world, devices));
    int rc = XRIT_OK;
    for (int i = 0; i < world && rc == XRIT_OK; ++i) {
        xrit_group *g = new (std::nothrow) xrit_group();
        if (!g) { rc = XRIT_E_NOMEM; break; }
        out[i] = g;
        g->rank = i;
        g->world = world;
        xrit_demod_config c = *cfg;
        c.device = devices[i];
        rc = group_finish_create(g, &c);
        RcclTransport *t = new (std::nothrow) RcclTransport();
        if (!t) { rc = XRIT_E_NOMEM; break; }
        g->tr = t;
        t->rank = i;
        t->world = world;
        t->comm = comms[(size_t)i];
        comms[(size_t)i] = nullptr;
    }
    if (rc != XRIT_OK) {
        for (int i = 0; i < world; ++i) { if (out[i]) xrit_group_destroy(out[i]); out[i] = nullptr; }
        for (auto c : comms) if (c) (void)ncclCommDestroy(c);
    }
    return rc;
}

int xrit_local_fabric_create(int world, xrit_local_fabric **out)
{
    if (!out || world < 1) { set_error("bad argument"); return XRIT_E_INVALID; }
    xrit_local_fabric *f = new (std::nothrow) xrit_local_fabric();
    if (!f) return XRIT_E_NOMEM;
    f->world = world;
    f->box.resize((size_t)world);
    f->gathered.resize((size_t)2 * world);
    f->red.resize((size_t)world);
    *out = f;
    return XRIT_OK;
}

void xrit_local_fabric_destroy(xrit_local_fabric *f) { delete f; }

int xrit_group_create_local(const xrit_demod_config *cfg, int rank, xrit_local_fabric *fabric, xrit_group **out)
{
    if (!cfg || !out || !fabric || rank < 0 || rank >= fabric->world) { set_error("bad argument"); return XRIT_E_INVALID; }
    *out = nullptr;
    xrit_group *g = new (std::nothrow) xrit_group();
    if (!g) return XRIT_E_NOMEM;
    g->rank = rank;
    g->world = fabric->world;
    int rc = group_finish_create(g, cfg);
    if (rc == XRIT_OK) {
        LocalTransport *t = new (std::nothrow) LocalTransport();
        if (!t) rc = XRIT_E_NOMEM;
        else { t->f = fabric; t->rank = rank; t->device = g->device; g->tr = t; }
    }
    if (rc != XRIT_OK) { xrit_group_destroy(g); return rc; }
    *out = g;
    return XRIT_OK;
}

void xrit_group_destroy(xrit_group *g)
{
    if (!g) return;
    (void)hipSetDevice(g->device);
    delete g->tr;
    if (g->chain) xrit_demod_destroy(g->chain);
    g->halo_in.release(); g->halo_syms.release(); g->soft_int.release(); g->tail_out.release(); g->tail_in.release(); g->host_in.release(); g->host_out.release();
    delete g;
}

xrit_demod *xrit_group_chain(xrit_group *g) { return g ? g->chain : nullptr; }
int xrit_group_rank(const xrit_group *g) { return g ? g->rank : -1; }
int xrit_group_world(const xrit_group *g) { return g ? g->world : 0; }
size_t xrit_group_halo_samples(const xrit_group *g) { return g ? g->halo : 0; }

int xrit_group_allreduce_max(xrit_group *g, double *value, void *stream)
{
    if (!g || !value) { set_error("null argument"); return XRIT_E_INVALID; }
    XR_HIP(hipSetDevice(g->device));
    if (g->world == 1) return XRIT_OK;
    return g->tr->allreduce_max(value, stream ? (hipStream_t)stream : (hipStream_t)xrit_demod_stream(g->chain));
}

int xrit_group_process_slice_device(xrit_group *g, const void *d_samples, size_t n, int type, float *d_soft, size_t cap,
                                    size_t *n_out, uint64_t *offset_out, int *polarity_out, void *stream)
{
    if (!g || !n_out || (n && !d_samples) || !d_soft) { set_error("null argument"); return XRIT_E_INVALID; }
    if (type < 0 || type > 3) { set_error("unknown sample type %d", type); return XRIT_E_INVALID; }
    *n_out = 0;
    if (offset_out) *offset_out = 0;
    if (polarity_out) *polarity_out = 1;
    XR_HIP(hipSetDevice(g->device));
    hipStream_t s = stream ? (hipStream_t)stream : (hipStream_t)xrit_demod_stream(g->chain);
    const size_t esz = type == XRIT_SAMPLE_FLOATIQ ? 8 : (type == XRIT_SAMPLE_S16IQ ? 4 : 2);
    const int rank = g->rank, world = g->world;
    const size_t H = g->halo;
    if (world > 1 && n < H) { set_error("slice of %zu samples is shorter than the halo of %zu", n, H); return XRIT_E_INVALID; }
    // every slice starts from a cold chain: the stream position this rank stopped at is not where this slice begins
    // (a world of one rank is the plain chain: consecutive calls are consecutive pieces of one stream)
    if (world > 1) XR_TRY(xrit_demod_reset(g->chain, s));

    // 1. halo: my last H samples to rank + 1, the last H of rank - 1 to me
    const bool has_next = rank + 1 < world, has_prev = rank > 0;
    if (has_prev) XR_TRY(g->halo_in.reserve(H * esz + 16));
    if (world > 1)
        XR_TRY(g->tr->exchange(has_next ? (const char *)d_samples + (n - H) * esz : nullptr, has_next ? H * esz : 0,
                               has_next ? rank + 1 : -1, has_prev ? g->halo_in.p : nullptr, has_prev ? H * esz : 0,
                               has_prev ? rank - 1 : -1, s));
    // 1b. the halo through the chain (cold start); its last symbols are looked at on the host
    const size_t keep = GROUP_TAIL + GROUP_KEEP;
    g->h_halo_syms.clear();
    if (has_prev) {
        const size_t hcap = H + 64;
        XR_TRY(g->halo_syms.reserve(hcap * sizeof(float)));
        size_t hk = 0;
        XR_TRY(xrit_demod_process_device(g->chain, g->halo_in.p, H, type, g->halo_syms.as<float>(), hcap, &hk, s));
        const size_t take = hk < keep ? hk : keep;
        g->h_halo_syms.resize(take);
        if (take)
            XR_HIP(hipMemcpyAsync(g->h_halo_syms.data(), g->halo_syms.as<float>() + (hk - take), take * sizeof(float),
                                  hipMemcpyDeviceToHost, s));
    }
    // 2. the slice
    const size_t icap = cap + 64;
    XR_TRY(g->soft_int.reserve(icap * sizeof(float)));
    size_t k = 0;
    XR_TRY(xrit_demod_process_device(g->chain, d_samples, n, type, g->soft_int.as<float>(), cap, &k, s));
    // 2b. boundary symbols: my last TAIL to rank + 1 (zero padded in front), those of rank - 1 to me
    XR_TRY(g->tail_out.reserve(GROUP_TAIL * sizeof(float)));
    XR_TRY(g->tail_in.reserve(GROUP_TAIL * sizeof(float)));
    g->h_prev_tail.clear();
    if (world > 1) {
        const size_t m = k < (size_t)GROUP_TAIL ? k : (size_t)GROUP_TAIL;
        XR_HIP(hipMemsetAsync(g->tail_out.p, 0, GROUP_TAIL * sizeof(float), s));
        if (m)
            XR_HIP(hipMemcpyAsync(g->tail_out.as<float>() + (GROUP_TAIL - m), g->soft_int.as<float>() + (k - m),
                                  m * sizeof(float), hipMemcpyDeviceToDevice, s));
        XR_TRY(g->tr->exchange(has_next ? g->tail_out.p : nullptr, has_next ? GROUP_TAIL * sizeof(float) : 0,
                               has_next ? rank + 1 : -1, has_prev ? g->tail_in.p : nullptr,
                               has_prev ? GROUP_TAIL * sizeof(float) : 0, has_prev ? rank - 1 : -1, s));
        if (has_prev) {
            g->h_prev_tail.resize(GROUP_TAIL);
            XR_HIP(hipMemcpyAsync(g->h_prev_tail.data(), g->tail_in.p, GROUP_TAIL * sizeof(float), hipMemcpyDeviceToHost, s));
        }
    }
    const size_t nhead = k < (size_t)GROUP_KEEP ? k : (size_t)GROUP_KEEP;
    g->h_head.resize(nhead);
    if (nhead) XR_HIP(hipMemcpyAsync(g->h_head.data(), g->soft_int.p, nhead * sizeof(float), hipMemcpyDeviceToHost, s));
    XR_HIP(hipStreamSynchronize(s));
    int pol_rel = 1, lag = 0;
    group_align(g->h_prev_tail, g->h_halo_syms, g->h_head, &pol_rel, &lag);
    // lag > 0: rank - 1 already emitted my first `lag` symbols; lag < 0: the -lag symbols before my slice were
    // only emitted here, over the halo
    size_t count = k;
    if (lag > 0) count = k > (size_t)lag ? k - (size_t)lag : 0;
    if (lag < 0) count = k + (size_t)(-lag);
    if (count > cap) { set_error("group: %zu symbols, capacity %zu", count, cap); return XRIT_E_CAPACITY; }
    // 3. absolute polarity and offset
    long long mine[2] = {pol_rel, (long long)count};
    std::vector<long long> all((size_t)2 * world, 0);
    if (world > 1) XR_TRY(g->tr->allgather2(mine, all.data(), s));
    else { all[0] = mine[0]; all[1] = mine[1]; }
    int pol = 1;
    unsigned long long offset = 0;
    for (int r = 1; r <= rank; ++r) pol *= (int)all[(size_t)2 * r];
    for (int r = 0; r < rank; ++r) offset += (unsigned long long)all[(size_t)2 * r + 1];
    // the aligned symbols, in the stream's polarity
    size_t pre = 0;
    if (lag < 0) {
        pre = (size_t)(-lag);
        const size_t nh = g->h_halo_syms.size();
        std::vector<float> tmp(pre);
        for (size_t i = 0; i < pre; ++i) tmp[i] = (float)pol * g->h_halo_syms[nh - pre + i];
        XR_HIP(hipMemcpyAsync(d_soft, tmp.data(), pre * sizeof(float), hipMemcpyHostToDevice, s));
        XR_HIP(hipStreamSynchronize(s));      // tmp goes out of scope
    }
    const size_t skip = lag > 0 ? (size_t)lag : 0;
    const size_t body = k > skip ? k - skip : 0;
    if (body)
        hipLaunchKernelGGL(group_emit_kernel, dim3(div_up(body, 256)), dim3(256), 0, s, g->soft_int.as<float>() + skip,
                           d_soft + pre, body, (float)pol);
    XR_HIP(hipGetLastError());
    *n_out = count;
    if (offset_out) *offset_out = offset;
    if (polarity_out) *polarity_out = pol;
    return XRIT_OK;
}

int xrit_group_process_slice_host(xrit_group *g, const void *samples, size_t n, int type, float *soft_out, size_t cap,
                                  size_t *n_out, uint64_t *offset_out, int *polarity_out)
{
    if (!g || !n_out || (n && !samples) || !soft_out) { set_error("null argument"); return XRIT_E_INVALID; }
    XR_HIP(hipSetDevice(g->device));
    hipStream_t s = (hipStream_t)xrit_demod_stream(g->chain);
    const size_t esz = type == XRIT_SAMPLE_FLOATIQ ? 8 : (type == XRIT_SAMPLE_S16IQ ? 4 : 2);
    XR_TRY(g->host_in.reserve(n * esz + 16));
    XR_TRY(g->host_out.reserve((cap + 1) * sizeof(float)));
    if (n) XR_HIP(hipMemcpyAsync(g->host_in.p, samples, n * esz, hipMemcpyHostToDevice, s));
    XR_TRY(xrit_group_process_slice_device(g, g->host_in.p, n, type, g->host_out.as<float>(), cap, n_out, offset_out,
                                           polarity_out, s));
    if (*n_out) XR_HIP(hipMemcpyAsync(soft_out, g->host_out.p, *n_out * sizeof(float), hipMemcpyDeviceToHost, s));
    XR_HIP(hipStreamSynchronize(s));
    return XRIT_OK;
}

}  // extern "C"
