// group.cpp -- one capture across the GPUs of a node, behind the C ABI.
//
// The reference demodulates one stream on one CPU thread (demodulator.cpp:170-175).  Here a burst is cut in `world`
// time slices, one per GPU.  What crosses GPUs (point to point over xGMI, RCCL ncclSend / ncclRecv, plus ONE
// all-gather of two integers per rank):
//   1. the halo: the last H input samples of rank g-1 go to rank g, which demodulates them first, from a cold
//      start, so that its filter histories are primed and its loops locked when its slice begins -- with the bit-exact
//      front end, ON the stream's own float32 trajectories (a rank that fell into the other Costas lock starts again
//      from a phase of pi, step 3b);
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

#include <rccl/rccl.h>      // types and prototypes only: the library is looked up at run time (RcclApi below)

#include <dlfcn.h>

#include <condition_variable>
#include <mutex>
#include <new>
#include <string>

using namespace xrit;

namespace {

constexpr int GROUP_TAIL = 256;     // soft symbols compared across a boundary
constexpr int GROUP_KEEP = 8;       // halo symbols kept in front of the slice output, for the straddling symbol

struct Transport {
    virtual ~Transport() {}
    // device buffers; the send and the receive of one exchange step are posted together
    virtual int exchange(const void *send, size_t send_bytes, int to, void *recv, size_t recv_bytes, int from,
                         hipStream_t s) = 0;
    virtual int allgather2(const long long mine[2], long long *all /* [2 * world] */, hipStream_t s) = 0;   // (a, b) of every rank
    virtual int allreduce_max(double *v, hipStream_t s) = 0;
    // a rank that cannot go on serving its peers inside a collective call: they must not wait for it for ever
    virtual void abort() {}
    // ranks the transport's own communicator reports (ncclCommCount); 0: no RCCL communicator behind this transport
    virtual int comm_ranks() { return 0; }
};

// ONE copy of RCCL per process.  A host that has brought its own (PyTorch ships librccl.so.1 inside its wheel and loads it
// for torch.distributed) must not end up with /opt/rocm's beside it -- two copies do not share communicators, device
// state or the bootstrap network --, so libxritdemod_amd.so does not link RCCL: the first group call takes the copy that is
// already in the process (dlopen RTLD_NOLOAD by soname) and loads /opt/rocm's only when there is none.
struct RcclApi {
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok = false;
    std::string why;
    RcclApi()
    {
        void *h = nullptr;
        for (const char *name : {"librccl.so.1", "librccl.so"})
            if (!h) h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);        // the copy the process already has
        std::string last;                // (dlerror() hands its message out ONCE and clears it: taken right after the failing call)
        for (const char *name : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"})
            if (!h) {
                h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                if (!h) { const char *e = dlerror(); if (e) last = e; }
            }
        if (!h) { why = "RCCL not found: " + (last.empty() ? std::string("dlopen failed") : last); return; }
        bool all = true;
#define XR_SYM(F) do { F = reinterpret_cast<decltype(F)>(dlsym(h, "nccl" #F)); if (!F) { all = false; why = "RCCL lacks nccl" #F; } } while (0)
        XR_SYM(GetUniqueId); XR_SYM(CommInitRank); XR_SYM(CommInitAll); XR_SYM(CommDestroy); XR_SYM(CommAbort); XR_SYM(CommCount);
        XR_SYM(GroupStart); XR_SYM(GroupEnd); XR_SYM(Send); XR_SYM(Recv); XR_SYM(AllGather); XR_SYM(AllReduce);
        XR_SYM(GetErrorString);
#undef XR_SYM
        ok = all;
    }
};
static RcclApi &rccl()
{
    static RcclApi api;
    return api;
}
#define XR_RCCL_READY()                                                                 \
    do {                                                                                \
        if (!rccl().ok) { set_error("%s", rccl().why.c_str()); return XRIT_E_HIP; }     \
    } while (0)

#define XR_NCCL(expr)                                                                   \
    do {                                                                                \
        XR_RCCL_READY();                                                                \
        ncclResult_t _r = (expr);                                                       \
        if (_r != ncclSuccess) {                                                        \
            set_error("%s failed: %s", #expr, rccl().GetErrorString(_r));                  \
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
        if (comm && owns && rccl().ok) (void)rccl().CommDestroy(comm);
        scratch.release();
    }
    void abort() override
    {
        if (comm && rccl().ok) (void)rccl().CommAbort(comm);      // pending and future operations of every rank on it return an error
        comm = nullptr;
    }
    int comm_ranks() override
    {
        int n = 0;
        if (!comm || !rccl().ok || rccl().CommCount(comm, &n) != ncclSuccess) return 0;
        return n;
    }
    int exchange(const void *send, size_t send_bytes, int to, void *recv, size_t recv_bytes, int from,
                 hipStream_t s) override
    {
        XR_NCCL(rccl().GroupStart());
        if (to >= 0 && send_bytes) XR_NCCL(rccl().Send(send, send_bytes, ncclChar, to, comm, s));
        if (from >= 0 && recv_bytes) XR_NCCL(rccl().Recv(recv, recv_bytes, ncclChar, from, comm, s));
        XR_NCCL(rccl().GroupEnd());
        return XRIT_OK;
    }
    int allgather2(const long long mine[2], long long *all, hipStream_t s) override
    {
        XR_TRY(scratch.reserve((size_t)(2 * world + 2) * sizeof(long long)));
        long long *d = scratch.as<long long>();
        XR_HIP(hipMemcpyAsync(d, mine, 2 * sizeof(long long), hipMemcpyHostToDevice, s));
        XR_NCCL(rccl().AllGather(d, d + 2, 2, ncclInt64, comm, s));
        XR_HIP(hipMemcpyAsync(all, d + 2, (size_t)2 * world * sizeof(long long), hipMemcpyDeviceToHost, s));
        XR_HIP(hipStreamSynchronize(s));
        return XRIT_OK;
    }
    int allreduce_max(double *v, hipStream_t s) override
    {
        XR_TRY(scratch.reserve((size_t)(2 * world + 2) * sizeof(long long)));
        double *d = scratch.as<double>();
        XR_HIP(hipMemcpyAsync(d, v, sizeof(double), hipMemcpyHostToDevice, s));
        XR_NCCL(rccl().AllReduce(d, d, 1, ncclDouble, ncclMax, comm, s));
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
    bool broken = false;                // a rank gave up inside a collective call: nobody waits any more
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
            f->cv.wait(lk, [&] { return !b.full || f->broken; });
            if (f->broken) { set_error("fabric: a rank gave up inside a collective call"); return XRIT_E_INVALID; }
            b.src = send; b.bytes = send_bytes; b.src_dev = device; b.full = true; b.taken = false;
            f->cv.notify_all();
        }
        if (from >= 0 && recv_bytes) {
            auto &b = f->box[rank];
            f->cv.wait(lk, [&] { return (b.full && !b.taken) || f->broken; });
            if (f->broken) { set_error("fabric: a rank gave up inside a collective call"); return XRIT_E_INVALID; }
            if (b.bytes != recv_bytes) { set_error("fabric: %zu bytes offered, %zu expected", b.bytes, recv_bytes); return XRIT_E_INVALID; }
            const void *src = b.src;
            const int src_dev = b.src_dev;
            lk.unlock();
            // on the receiver's stream, and finished before the slot is handed back: a caller-supplied non-blocking
            // stream is not ordered against the null stream
            hipError_t e = src_dev == device ? hipMemcpyAsync(recv, src, recv_bytes, hipMemcpyDeviceToDevice, s)
                                             : hipMemcpyPeerAsync(recv, device, src, src_dev, recv_bytes, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            lk.lock();
            b.taken = true;
            f->cv.notify_all();
            if (e != hipSuccess) { set_error("fabric copy failed: %s", hipGetErrorString(e)); return XRIT_E_HIP; }
        }
        if (to >= 0 && send_bytes) {
            // the buffer stays ours until the neighbour has copied it
            auto &b = f->box[to];
            f->cv.wait(lk, [&] { return b.taken || f->broken; });
            if (f->broken) { set_error("fabric: a rank gave up inside a collective call"); return XRIT_E_INVALID; }
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
            f->cv.wait(lk, [&] { return f->generation != gen || f->broken; });
        }
        if (f->broken) { set_error("fabric: a rank gave up inside a collective call"); return XRIT_E_INVALID; }
        return XRIT_OK;
    }
    int allgather2(const long long mine[2], long long *all, hipStream_t) override
    {
        std::unique_lock<std::mutex> lk(f->m);
        f->gathered[2 * rank] = mine[0];
        f->gathered[2 * rank + 1] = mine[1];
        XR_TRY(rendezvous(lk));
        for (int i = 0; i < 2 * f->world; ++i) all[i] = f->gathered[i];
        return rendezvous(lk);          // nobody overwrites before everybody has read
    }
    int allreduce_max(double *v, hipStream_t) override
    {
        std::unique_lock<std::mutex> lk(f->m);
        f->red[rank] = *v;
        XR_TRY(rendezvous(lk));
        double m = f->red[0];
        for (int i = 1; i < f->world; ++i) m = f->red[i] > m ? f->red[i] : m;
        *v = m;
        return rendezvous(lk);
    }
    void abort() override
    {
        std::unique_lock<std::mutex> lk(f->m);
        f->broken = true;
        f->cv.notify_all();
    }
};

}  // namespace

struct xrit_group {
    xrit_demod *chain = nullptr;
    Transport *tr = nullptr;
    int rank = 0, world = 1, device = 0;
    size_t halo = 0;            // input samples taken over from the previous rank
    DevBuf halo_in, halo_syms, soft_int, tail_out, tail_in, host_in, host_out, zeros, pre_dev;
    std::vector<float> h_halo_syms, h_head, h_prev_tail, h_pre;
    unsigned decimation = 1;
    // streaming: consecutive calls are consecutive bursts of ONE capture.  The last rank keeps the end of its slice
    // (halo samples, boundary symbols in the stream's polarity) and hands it to rank 0 at the start of the next call --
    // the exchanges become a ring -- so that rank 0 warms up over a halo like every other rank instead of starting cold
    // in the middle of the stream.
    unsigned long long calls = 0;       // slice calls of this capture that succeeded
    DevBuf keep_halo, keep_tail, keep_carry;
    DevBuf carry_out, carry_in, carry_mine;     // the clock recovery's carried state: mine at the end of the slice, the previous rank's, mine at its start
    std::vector<unsigned char> h_carry_in, h_carry_mine;
    unsigned long long relocks = 0;      // slices started a second time, from the other Costas lock
    unsigned long long handovers = 0;    // slices whose clock recovery ran again from the previous rank's loop state
    unsigned long long joined = 0;       // slices that had met the previous rank's loop state bit for bit inside their halo
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
    g->decimation = c.decimation;
    // warm-up in front of a slice: what the walkers of overlapping blocks inside a GPU take (clock_overlap.h), 49 152 symbols --
    // with 24 576 (rounds 2 - 5) the clock recovery was still 2e-4 from the stream's over a slice's first 10 - 20 k symbols,
    // with 49 152 it has met it (bit for bit behind the bit-exact front end: profiles/r6_group_windows.txt)
    int warm = 49152;
    if (const char *e = getenv("XRIT_GROUP_WARM")) { const int v = atoi(e); if (v >= 1024) warm = v; }    // (measurement switch, scripts/r6_group_windows.py)
    g->halo = g->world > 1 ? group_halo_samples(c, xrit_demod_sps(g->chain), xrit_demod_decimator_ntaps(g->chain), warm) : 0;
    return XRIT_OK;
}

}  // namespace

extern "C" {

int xrit_group_unique_id(void *id128)
{
    if (!id128) { set_error("null argument"); return XRIT_E_INVALID; }
    static_assert(sizeof(ncclUniqueId) <= XRIT_GROUP_ID_BYTES, "ncclUniqueId fits the ABI's id buffer");
    ncclUniqueId id;
    XR_NCCL(rccl().GetUniqueId(&id));
    memset(id128, 0, XRIT_GROUP_ID_BYTES);
    memcpy(id128, &id, sizeof id);
    return XRIT_OK;
}

int xrit_group_create(const xrit_demod_config *cfg, int rank, int world, const void *id128, xrit_group **out)
{
    if (!cfg || !out || !id128 || world < 1 || rank < 0 || rank >= world) { set_error("bad argument"); return XRIT_E_INVALID; }
    *out = nullptr;
    XR_RCCL_READY();
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
                ncclResult_t r = rccl().CommInitRank(&t->comm, world, id, rank);
                if (r != ncclSuccess) { set_error("ncclCommInitRank failed: %s", rccl().GetErrorString(r)); rc = XRIT_E_HIP; }
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
    XR_NCCL(rccl().CommInitAll(comms.data(), world, devices));
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
        for (auto c : comms) if (c && rccl().ok) (void)rccl().CommDestroy(c);
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
    g->halo_in.release(); g->halo_syms.release(); g->soft_int.release(); g->tail_out.release(); g->tail_in.release(); g->host_in.release(); g->host_out.release(); g->zeros.release(); g->pre_dev.release(); g->keep_halo.release(); g->keep_tail.release(); g->keep_carry.release(); g->carry_out.release(); g->carry_in.release(); g->carry_mine.release();
    delete g;
}

xrit_demod *xrit_group_chain(xrit_group *g) { return g ? g->chain : nullptr; }
int xrit_group_rank(const xrit_group *g) { return g ? g->rank : -1; }
int xrit_group_world(const xrit_group *g) { return g ? g->world : 0; }
size_t xrit_group_halo_samples(const xrit_group *g) { return g ? g->halo : 0; }
void xrit_group_counters(const xrit_group *g, uint64_t *relocks, uint64_t *handovers, uint64_t *joined)
{
    if (relocks) *relocks = g ? g->relocks : 0;
    if (handovers) *handovers = g ? g->handovers : 0;
    if (joined) *joined = g ? g->joined : 0;
}
int xrit_group_rccl_ranks(xrit_group *g) { return g && g->tr ? g->tr->comm_ranks() : 0; }

int xrit_group_restart(xrit_group *g)
{
    if (!g) { set_error("null argument"); return XRIT_E_INVALID; }
    g->calls = 0;        // the next slice call begins a new capture (collective by convention: every rank calls it)
    return XRIT_OK;
}

int xrit_group_allreduce_max(xrit_group *g, double *value, void *stream)
{
    if (!g || !value) { set_error("null argument"); return XRIT_E_INVALID; }
    XR_HIP(hipSetDevice(g->device));
    if (g->world == 1) return XRIT_OK;
    return g->tr->allreduce_max(value, stream ? (hipStream_t)stream : (hipStream_t)xrit_demod_stream(g->chain));
}

// The call is a collective: every rank must reach every exchange and both all-gathers whatever happens to it in
// between, or its neighbours wait for ever in ncclRecv / ncclAllGather (or in the fabric's condition variable).  So
// the local work never returns early: a failure is remembered, the rank goes on exchanging (zeros where it has
// nothing), the status rides in the gathered words, and every rank returns an error after the gather that showed one.
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
    const unsigned D = g->decimation;
    const bool ring = world > 1 && g->calls > 0;                      // (every rank counts the same calls)
    const bool has_next = rank + 1 < world || ring, has_prev = rank > 0 || ring;
    const int to = rank + 1 < world ? rank + 1 : 0, from = rank > 0 ? rank - 1 : world - 1;
    const bool wrap_send = ring && rank + 1 == world;                 // the last rank sends what it kept of the call before
    int rc = XRIT_OK;              // this rank's own status
    std::string why;
    auto fail = [&](int code) { if (rc == XRIT_OK) { rc = code; why = get_error(); } };
#define GR_STEP(expr) do { if (rc == XRIT_OK) { int r_ = (expr); if (r_ != XRIT_OK) fail(r_); } } while (0)
#define GR_HIP(expr) do { if (rc == XRIT_OK) { hipError_t e_ = (expr); if (e_ != hipSuccess) { set_error("%s failed: %s", #expr, hipGetErrorString(e_)); fail(XRIT_E_HIP); } } } while (0)
    if (world > 1 && n < H) { set_error("slice of %zu samples is shorter than the halo of %zu", n, H); fail(XRIT_E_INVALID); }
    // (demodulator.cpp:137 drops n mod decimation samples per call: between two ranks that would shift the phase of
    // the decimator against the uninterrupted stream)
    if (world > 1 && D > 1 && n % D) {
        set_error("slice of %zu samples is not a whole number of decimation periods (%u): the next slice's decimator phase would shift", n, D);
        fail(XRIT_E_INVALID);
    }
    // every slice starts from a cold chain: the stream position this rank stopped at is not where this slice begins
    // (a world of one rank is the plain chain: consecutive calls are consecutive pieces of one stream)
    if (world > 1) GR_STEP(xrit_demod_reset(g->chain, s));

    // 1. halo: my last H samples to rank + 1, the last H of rank - 1 to me (a rank that failed sends zeros)
    if (world > 1) {
        if (has_prev && g->halo_in.reserve(H * esz + 16) != XRIT_OK) fail(XRIT_E_NOMEM);
        const void *snd = nullptr;
        if (has_next) {
            if (wrap_send) snd = g->keep_halo.p;
            else if (rc == XRIT_OK) snd = (const char *)d_samples + (n - H) * esz;
            else if (g->zeros.reserve(H * esz + 16) == XRIT_OK && hipMemsetAsync(g->zeros.p, 0, H * esz, s) == hipSuccess) snd = g->zeros.p;
        }
        // (no buffer at all to send from or to receive into: the peers cannot be served -- the communicator is torn down)
        if ((has_next && !snd) || (has_prev && !g->halo_in.p)) { g->tr->abort(); set_error("group: out of device memory inside a collective call (%s)", why.c_str()); return XRIT_E_NOMEM; }
        int xr = g->tr->exchange(snd, has_next ? H * esz : 0, has_next ? to : -1, has_prev ? g->halo_in.p : nullptr,
                                 has_prev ? H * esz : 0, has_prev ? from : -1, s);
        if (xr != XRIT_OK) fail(xr);
    }
    // 1b. the halo through the chain (cold start); its last symbols are looked at on the host
    const size_t keep = GROUP_TAIL + GROUP_KEEP;
    g->h_halo_syms.clear();
    auto run_halo = [&]() {
        const size_t hcap = H + 64;
        GR_STEP(g->halo_syms.reserve(hcap * sizeof(float)));
        size_t hk = 0;
        GR_STEP(xrit_demod_process_device(g->chain, g->halo_in.p, H, type, g->halo_syms.as<float>(), hcap, &hk, s));
        GR_STEP(xrit_demod_prepare_flipped(g->chain, s));       // what the clock recovery would carry on the other sign of the stream
        const size_t take = hk < keep ? hk : keep;
        if (rc == XRIT_OK) g->h_halo_syms.resize(take);
        if (take) GR_HIP(hipMemcpyAsync(g->h_halo_syms.data(), g->halo_syms.as<float>() + (hk - take), take * sizeof(float), hipMemcpyDeviceToHost, s));
    };
    if (has_prev && rc == XRIT_OK) run_halo();
    // 2. the slice
    const size_t icap = cap + 64;
    GR_STEP(g->soft_int.reserve(icap * sizeof(float)));
    size_t k = 0;
    GR_STEP(xrit_demod_process_device(g->chain, d_samples, n, type, g->soft_int.as<float>(), cap, &k, s));
    // 2b. boundary symbols: my last TAIL to rank + 1 (zero padded in front), those of rank - 1 to me
    g->h_prev_tail.clear();
    if (world > 1) {
        if (g->tail_out.reserve(GROUP_TAIL * sizeof(float)) != XRIT_OK || g->tail_in.reserve(GROUP_TAIL * sizeof(float)) != XRIT_OK) {
            g->tr->abort(); set_error("group: out of device memory inside a collective call"); return XRIT_E_NOMEM;
        }
        const size_t m = rc != XRIT_OK ? 0 : (k < (size_t)GROUP_TAIL ? k : (size_t)GROUP_TAIL);
        (void)hipMemsetAsync(g->tail_out.p, 0, GROUP_TAIL * sizeof(float), s);
        if (m) GR_HIP(hipMemcpyAsync(g->tail_out.as<float>() + (GROUP_TAIL - m), g->soft_int.as<float>() + (k - m), m * sizeof(float), hipMemcpyDeviceToDevice, s));
        // (boundary symbols travel in the polarity their rank computed them in, except the last rank's kept ones, which
        // are in the stream's: rank 0 then reads its ABSOLUTE polarity off them)
        int xr = g->tr->exchange(has_next ? (wrap_send ? g->keep_tail.p : g->tail_out.p) : nullptr, has_next ? GROUP_TAIL * sizeof(float) : 0,
                                 has_next ? to : -1, has_prev ? g->tail_in.p : nullptr,
                                 has_prev ? GROUP_TAIL * sizeof(float) : 0, has_prev ? from : -1, s);
        if (xr != XRIT_OK) fail(xr);
        if (has_prev && rc == XRIT_OK) {
            g->h_prev_tail.resize(GROUP_TAIL);
            GR_HIP(hipMemcpyAsync(g->h_prev_tail.data(), g->tail_in.p, GROUP_TAIL * sizeof(float), hipMemcpyDeviceToHost, s));
        }
    }
    auto fetch_head = [&]() {
        const size_t nhead = k < (size_t)GROUP_KEEP ? k : (size_t)GROUP_KEEP;
        g->h_head.resize(nhead);
        if (nhead) GR_HIP(hipMemcpyAsync(g->h_head.data(), g->soft_int.p, nhead * sizeof(float), hipMemcpyDeviceToHost, s));
        GR_HIP(hipStreamSynchronize(s));
    };
    fetch_head();
    int pol_rel = 1, lag = 0;
    if (rc == XRIT_OK) group_align(g->h_prev_tail, g->h_halo_syms, g->h_head, &pol_rel, &lag);

    // 3. every rank's polarity against its predecessor (and status) -> absolute polarity
    std::vector<long long> all((size_t)2 * world, 0);
    auto gather = [&](long long a, long long b) -> int {
        long long mine[2] = {a, b};
        if (world > 1) return g->tr->allgather2(mine, all.data(), s);
        all[0] = mine[0]; all[1] = mine[1];
        return XRIT_OK;
    };
    auto peers_ok = [&]() -> bool {     // the gathered second words are status codes
        for (int r = 0; r < world; ++r)
            if (all[(size_t)2 * r + 1] != XRIT_OK) {
                if (rc != XRIT_OK) set_error("%s", why.c_str());
                else set_error("group: rank %d failed with code %lld; this rank's slice is dropped with it", r, all[(size_t)2 * r + 1]);
                return false;
            }
        return true;
    };
    {
        int gr = gather(pol_rel, rc);
        if (gr != XRIT_OK) { g->tr->abort(); return gr; }
        if (!peers_ok()) { g->calls = 0; return rc != XRIT_OK ? rc : XRIT_E_INVALID; }
    }
    int pol = 1;
    for (int r = 0; r <= rank; ++r) pol *= (int)all[(size_t)2 * r];      // (rank 0: +1 on a capture's first call)
    // 3b. a rank that locked pi away from the stream.
    // With the bit-exact front end (cfg.front_exact: every call of parity mode, calls below a million symbols by default) the
    // rank starts once more, from a Costas phase of pi instead of 0: the loop's equations do not see a half turn, so it pulls
    // in along the same path into the OTHER lock, the stream's; there its float32 trajectory meets the stream's own bit for
    // bit within the halo, as a rank that fell on the right side at once does (merge lengths: DESIGN.md section 4b), and the
    // clock recovery behind it starts on the stream's own words.  (profiles/r6_group_windows.txt: flipped ranks, whose Costas
    // output is the stream's to rounding only, sat at the clock recovery's floor -- 0.6 .. 1.8e-4 over 141 k symbols -- next
    // to 0.8 .. 3e-5 for unflipped ones.)
    bool relocked = false;
    if (pol < 0 && has_prev && rc == XRIT_OK && xrit_demod_front_exact_for(g->chain, n) == 1 && xrit_demod_front_exact_for(g->chain, H) == 1) {
        GR_STEP(xrit_demod_reset(g->chain, s));
        GR_STEP(xrit_demod_flip_costas_phase(g->chain, s));
        run_halo();
        k = 0;
        GR_STEP(xrit_demod_process_device(g->chain, d_samples, n, type, g->soft_int.as<float>(), cap, &k, s));
        fetch_head();
        int p2 = 1, lag2 = 0;
        if (rc == XRIT_OK) group_align(g->h_prev_tail, g->h_halo_syms, g->h_head, &p2, &lag2);
        // (p2 is this run against rank - 1's boundary symbols, which are in rank - 1's FIRST polarity: the stream's lock is
        // the one that changed sign against them)
        if (rc == XRIT_OK && p2 == -pol_rel) { relocked = true; lag = lag2; }
        g->relocks += 1;
    }
    // Otherwise (the fast front end, or a second start that fell on the same side): the Mueller & Mueller detector slices to
    // {0, 1}, so the loop on -y is another loop than minus the loop on y (3e-3 rms in the symbols).  Its clock recovery runs
    // once more, on the sign-flipped Costas output, from the state it had at the start of the slice: the symbols then come
    // out in the stream's polarity, to the same floor as any other rank's.
    if (pol < 0 && !relocked) {
        GR_STEP(xrit_demod_redo_clock_flipped(g->chain, g->soft_int.as<float>(), cap, &k, s));
        for (auto &v : g->h_halo_syms) v = -v;
        fetch_head();
        int p2 = 1;
        if (rc == XRIT_OK) group_align(g->h_prev_tail, g->h_halo_syms, g->h_head, &p2, &lag);
    }
    // 3c. ONE loop state across the slices.  The reference's clock recovery is one object that carries its state across every
    // chunk (demodulator.cpp:446-450, :156): here the state at the end of a slice -- mu, omega, the last symbols and decisions,
    // the samples not consumed yet, one record of 8 KB -- travels from every rank to the rank behind it (the last rank keeps its
    // own for rank 0's next call).  A rank compares it with the state its own warm-up over the halo reached at the slice's
    // start: bit for bit the same (the float32 loops do meet: `joined`) and its symbols continue the stream's as they are;
    // otherwise, where the slice's recovery was ONE exact walk behind the bit-exact front end, it is walked again from the
    // handed-over state (`handovers`) -- the symbols are then the single chain's, word for word where that chain's are the CPU
    // chain's -- and the record it hands on is the corrected one, so such ranks receive first and send afterwards.  Slices long
    // enough for overlapping walkers do not walk again: a rank boundary is to them what the 256 joints inside every GPU are
    // (DESIGN.md section 10); they send and receive at once.
    bool seamless = false;
    if (world > 1) {
        const size_t RB = xrit_demod_clock_carry_bytes();
        if (g->carry_out.reserve(RB) != XRIT_OK || g->carry_in.reserve(RB) != XRIT_OK || g->carry_mine.reserve(RB) != XRIT_OK ||
            g->keep_carry.reserve(RB) != XRIT_OK) {
            g->tr->abort(); set_error("group: out of device memory inside a collective call"); return XRIT_E_NOMEM;
        }
        auto pack_end = [&](void *dst) {        // what the next slice starts from (a rank that failed: no record)
            if (rc != XRIT_OK || xrit_demod_export_clock_carry(g->chain, 0, dst, s) != XRIT_OK) (void)hipMemsetAsync(dst, 0, RB, s);
        };
        const bool onward = rank + 1 < world;
        const bool exact_here = rc == XRIT_OK && has_prev && xrit_demod_front_exact_for(g->chain, n) == 1 &&
                                xrit_demod_front_exact_for(g->chain, H) == 1 && xrit_demod_last_clock_exact(g->chain) == 1;
        if (wrap_send) {
            int xr = g->tr->exchange(g->keep_carry.p, RB, to, nullptr, 0, -1, s);
            if (xr != XRIT_OK) fail(xr);
        }
        bool sent = false;
        if (has_prev) {
            int xr;
            if (!exact_here && onward) {
                pack_end(g->carry_out.p);
                xr = g->tr->exchange(g->carry_out.p, RB, to, g->carry_in.p, RB, from, s);
                sent = true;
            } else {
                xr = g->tr->exchange(nullptr, 0, -1, g->carry_in.p, RB, from, s);
            }
            if (xr != XRIT_OK) fail(xr);
        }
        if (has_prev && rc == XRIT_OK && xrit_demod_export_clock_carry(g->chain, 1, g->carry_mine.p, s) == XRIT_OK) {
            g->h_carry_in.resize(RB); g->h_carry_mine.resize(RB);
            GR_HIP(hipMemcpyAsync(g->h_carry_in.data(), g->carry_in.p, RB, hipMemcpyDeviceToHost, s));
            GR_HIP(hipMemcpyAsync(g->h_carry_mine.data(), g->carry_mine.p, RB, hipMemcpyDeviceToHost, s));
            GR_HIP(hipStreamSynchronize(s));
            unsigned head[2] = {0, 0};
            if (rc == XRIT_OK) memcpy(head, g->h_carry_in.data(), sizeof head);
            if (rc == XRIT_OK && head[0] == 1u && head[1] <= 1024u) {
                if (memcmp(g->h_carry_in.data(), g->h_carry_mine.data(), RB) == 0) {
                    seamless = true;
                    g->joined += 1;
                } else if (exact_here) {
                    GR_STEP(xrit_demod_redo_clock_from(g->chain, g->carry_in.p, g->soft_int.as<float>(), cap, &k, s));
                    fetch_head();
                    if (rc == XRIT_OK) { seamless = true; g->handovers += 1; }
                }
            }
        }
        if (onward && !sent) {
            pack_end(g->carry_out.p);
            int xr = g->tr->exchange(g->carry_out.p, RB, to, nullptr, 0, -1, s);
            if (xr != XRIT_OK) fail(xr);
        }
        if (!onward) pack_end(g->keep_carry.p);      // the last rank: for rank 0, should the capture go on
    }
    // (a slice that continues from the very state the rank in front ended in has no straddling symbol to settle)
    if (seamless) lag = 0;
    // lag > 0: rank - 1 already emitted my first `lag` symbols; lag < 0: the -lag symbols before my slice were
    // only emitted here, over the halo
    size_t count = k;
    if (lag > 0) count = k > (size_t)lag ? k - (size_t)lag : 0;
    if (lag < 0) count = k + (size_t)(-lag);
    if (rc == XRIT_OK && count > cap) { set_error("group: %zu symbols, capacity %zu", count, cap); fail(XRIT_E_CAPACITY); }
    // 4. counts (and status) -> output offset
    {
        int gr = gather((long long)count, rc);
        if (gr != XRIT_OK) { g->tr->abort(); return gr; }
        if (!peers_ok()) { g->calls = 0; return rc != XRIT_OK ? rc : XRIT_E_INVALID; }
    }
    unsigned long long offset = 0;
    for (int r = 0; r < rank; ++r) offset += (unsigned long long)all[(size_t)2 * r];
    // From here on nothing is exchanged any more, but `calls` is the collective switch between a capture's first call and
    // its ring: a rank that failed now (device memory, a copy) and simply returned would count one call fewer than its
    // peers and wait for ever in the next call's halo exchange.  It tears the transport down instead: its peers' next
    // exchange fails and every rank begins a new capture.
    auto tail_work = [&]() -> int {
        // the aligned symbols, in the stream's polarity (a rank that ran again already has them so)
        const float emit = 1.0f;
        size_t pre = 0;
        if (lag < 0) {
            pre = (size_t)(-lag);
            const size_t nh = g->h_halo_syms.size();
            XR_TRY(g->pre_dev.reserve(64 * sizeof(float)));
            g->h_pre.resize(pre);
            // (halo symbols are in this rank's own polarity, already negated above where it ran again)
            for (size_t i = 0; i < pre; ++i) g->h_pre[i] = (pol < 0 ? 1.0f : (float)pol) * g->h_halo_syms[nh - pre + i];
            XR_HIP(hipMemcpyAsync(d_soft, g->h_pre.data(), pre * sizeof(float), hipMemcpyHostToDevice, s));
        }
        const size_t skip = lag > 0 ? (size_t)lag : 0;
        const size_t body = k > skip ? k - skip : 0;
        if (body)
            hipLaunchKernelGGL(group_emit_kernel, dim3(div_up(body, 256)), dim3(256), 0, s, g->soft_int.as<float>() + skip,
                               d_soft + pre, body, emit);
        XR_HIP(hipGetLastError());
        if (pre) XR_HIP(hipStreamSynchronize(s));       // h_pre is read by the copy until then
        // the last rank keeps what rank 0 needs to go on from here in the next call
        if (world > 1 && rank + 1 == world) {
            XR_TRY(g->keep_halo.reserve(H * esz + 16));
            XR_TRY(g->keep_tail.reserve(GROUP_TAIL * sizeof(float)));
            XR_HIP(hipMemcpyAsync(g->keep_halo.p, (const char *)d_samples + (n - H) * esz, H * esz, hipMemcpyDeviceToDevice, s));
            const size_t m = count < (size_t)GROUP_TAIL ? count : (size_t)GROUP_TAIL;
            XR_HIP(hipMemsetAsync(g->keep_tail.p, 0, GROUP_TAIL * sizeof(float), s));
            if (m) XR_HIP(hipMemcpyAsync(g->keep_tail.as<float>() + (GROUP_TAIL - m), d_soft + (count - m), m * sizeof(float), hipMemcpyDeviceToDevice, s));
        }
        return XRIT_OK;
    };
    {
        const int tr_ = tail_work();
        if (tr_ != XRIT_OK) {
            g->calls = 0;
            if (world > 1) g->tr->abort();
            return tr_;
        }
    }
    g->calls += 1;
    *n_out = count;
    if (offset_out) *offset_out = offset;
    if (polarity_out) *polarity_out = pol;
#undef GR_STEP
#undef GR_HIP
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
