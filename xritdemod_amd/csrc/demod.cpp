// demod.cpp -- the chain object and the C ABI (include/xritdemod_amd.h).
// Stage order, buffer hand-over and parameter defaults follow
// /root/reference/demodulator/src/demodulator.cpp:100-168 (processSamples) and
// :436-450 (construction); constants from Parameters.h:16-37.
#include "kernels.h"

#include <chrono>
#include <cmath>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

namespace xrit {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
const char *get_error() { return g_err; }

static int select_device(int device)
{
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        set_error("no usable HIP device (%s); this library has no CPU path",
                  e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
        return XRIT_E_NO_DEVICE;
    }
    if (device < 0 || device >= count) {
        set_error("device %d out of range (0..%d)", device, count - 1);
        return XRIT_E_INVALID;
    }
    XR_HIP(hipSetDevice(device));
    return XRIT_OK;
}

}  // namespace xrit

using namespace xrit;

struct xrit_demod {
    xrit_demod_config cfg{};
    int device = 0;
    hipStream_t stream = nullptr;
    float sps = 0, circuit_rate = 0;
    int dec_ntaps = 0;
    FirStage dec, rrc;
    AgcStage agc;
    CostasStage costas;
    ClockStage clock;
    // Two sets of front-end buffers: the front end of the NEXT burst (xrit_demod_prefetch_device, on stream2) fills one
    // while the feedback loops of the current burst read the other.
    DevBuf bufA[2], bufB[2], bufC[2], bufR[2], stat[2], in_dev, soft_dev, q_in, q_out;
    int next_set = 0;
    hipStream_t stream2 = nullptr;
    hipStream_t stream_c_pool = nullptr;    // the set's Costas stream (used by cfg.front_exact = 2 only)
    bool pooled = false;                    // the streams go back to the process's pool when the handle is destroyed
    hipStream_t stream_c = nullptr;     // cfg.front_exact = 2: the Costas loops (their exact walkers are latency, not work) beside the next front end
    // bursts whose Costas loop is a handful of small kernels (few circuit-rate samples: large decimations) run that loop on the
    // walker stream their own walkers will take, beside the next burst's decimator instead of behind it (ov_service)
    size_t costas_own_stream_below = 16u << 20;                     // circuit-rate samples per burst (XRIT_OV_CSTREAM_BELOW)
#ifndef XRIT_WALK_STREAMS
#define XRIT_WALK_STREAMS 2
#endif
    hipStream_t stream3[XRIT_WALK_STREAMS] = {};                    // the clock recovery's walkers of bursts started ahead (round 5):
                                                                    // two streams, so that two bursts' walkers run side by side
    hipEvent_t ev_done = nullptr;                                   // the current call's clock recovery has left its result
    hipEvent_t ev_ready = nullptr, ev_fe[2] = {nullptr, nullptr};   // input ready on the caller's stream / front end of a set done
    hipEvent_t ev_relay = nullptr;                                  // the relay kernels of the current call come next
    hipEvent_t ev_costas = nullptr;                                 // the Costas loop started ahead on stream2 has run its batch
    bool costas_idle = false;   // the current call's Costas loop was finished before its clock recovery began: the stage is free
    int last_fe_set = -1;       // set of the front end that ran last (its event orders the next one behind it)
    struct Prefetched {
        const void *samples = nullptr; size_t n = 0; int type = 0; int set = 0;
        size_t length = 0; const float2 *rrc = nullptr; bool stat_ready = false; const float *agc_flag = nullptr;
        bool exact = false;             // its front end was the bit-exact one (front_exact_call)
        bool costas_begun = false;      // the Costas loop of this input has been started too (on stream2, behind its front end)
        float2 *slot = nullptr;         // ... writing here (the clock recovery's next input buffer)
        // round 5, bursts whose clock recovery walks overlapping blocks (clock_overlap.h): everything up to the walkers runs ahead
        bool costas_finished = false;   // the host has looked at the loop's stop test (and continued it where it had not closed)
        int ov_job = -1;                // the clock stage's job of this input
        bool walk_launched = false;     // its walkers have been enqueued (stream3)
        int costas_stream = 0;          // 1: its Costas loop runs on the walker stream its own walkers will take (c_stream)
        hipStream_t c_stream = nullptr; // the stream its Costas loop was enqueued on
        bool agc_fallback = false;      // what the AGC's guard and the Costas loop reported for this input
        int c_passes = 0; unsigned c_unconverged = 0; float c_max_residual = 0; bool c_walked = false;
        bool launched = true;   // false: registered only -- a handle whose clock recovery is relayed (cfg.clock_exact >= 1)
                                // starts the front end of the next burst in front of the relay kernels of the current one,
                                // which leave most of the chip idle, instead of under the loops that fill it
    } pf[XRIT_AHEAD + 1];                    // inputs registered ahead, oldest first: the one of the next process call and the two behind it
    int pf_count = 0;
    RtlIngestStage rtl;
    bool poisoned = false;      // a call failed after some stage had advanced its carried state
    // what the Costas loop of the CURRENT call reported when it was finished: taken there, because with a registered next input the
    // stage begins the next burst's loop (and resets its counters) before this call returns
    struct CostasSeen { int passes = 0; unsigned unconverged = 0; float max_residual = 0; bool walked = false; } costas_seen;
    bool no_defer = false;      // XRIT_NO_DEFER: registered front ends start at once also with the exact closure on (A/B runs)
    bool keep_stages = false;   // every stage's output is copied (diagnostics, tests): no fusion across stages
    bool keep_symbols = false;  // only the complex symbols of the clock recovery are kept (constellation tap)
    bool agc_fallback_seen = false;   // this call: the AGC's guard sent a slice down the serial path
    DevBuf stage_buf[5];
    size_t stage_n[5] = {0, 0, 0, 0, 0};
    Profiler prof;
    xrit_demod_stats stats{};
    std::vector<std::string> prof_names;
};

struct xrit_fir { int device; hipStream_t stream; FirStage st; DevBuf in, out; };
struct xrit_agc { int device; hipStream_t stream; AgcStage st; DevBuf in, out; };
struct xrit_costas { int device; hipStream_t stream; CostasStage st; DevBuf in, out; };
struct xrit_clock { int device; hipStream_t stream; ClockStage st; DevBuf in, out; };
struct xrit_rtl { int device; hipStream_t stream; RtlIngestStage st; DevBuf in, out; };

template <typename H> static int stage_open(H *h, int device)
{
    XR_TRY(select_device(device));
    h->device = device;
    XR_HIP(hipStreamCreate(&h->stream));
    return XRIT_OK;
}
template <typename H> static void stage_close(H *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) { (void)hipStreamSynchronize(h->stream); }
    h->st.release();
    h->in.release();
    h->out.release();
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}


extern "C" {

const char *xrit_last_error(void) { return get_error(); }
const char *xrit_version(void) { return "xritdemod_amd 0.1 (gfx950)"; }

int xrit_device_count(void)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) return 0;
    return count;
}

int xrit_lowpass_taps(double gain, double sample_rate, double cutoff, double transition_width, float *taps, int cap)
{
    std::vector<float> h = design_lowpass(gain, sample_rate, cutoff, transition_width);
    if ((int)h.size() > cap || !taps) return -(int)h.size();
    memcpy(taps, h.data(), h.size() * sizeof(float));
    return (int)h.size();
}

int xrit_rrc_taps(double gain, double sample_rate, double symbol_rate, double alpha, int ntaps, float *taps, int cap)
{
    std::vector<float> h = design_rrc(gain, sample_rate, symbol_rate, alpha, ntaps);
    if ((int)h.size() > cap || !taps) return -(int)h.size();
    memcpy(taps, h.data(), h.size() * sizeof(float));
    return (int)h.size();
}

void xrit_mmse_table(float *table) { design_mmse_table(table); }

static void config_common(xrit_demod_config *c, float sample_rate, uint32_t decimation)
{
    memset(c, 0, sizeof *c);
    c->sample_rate = sample_rate;
    c->decimation = decimation ? decimation : 1;
    c->rrc_taps = 63;                                  // RRC_TAPS
    c->agc_rate = 0.01f;                               // AGC_RATE
    c->agc_reference = 0.5f;                           // AGC_REFERENCE
    c->agc_gain = 1.f;                                 // AGC_GAIN
    c->agc_max_gain = 4000;                            // AGC_MAX_GAIN
    c->pll_alpha = 0.0037f;                            // (float)CLOCK_ALPHA, demodulator.cpp:220
    c->clock_mu = 0.5f;                                // CLOCK_MU
    c->clock_alpha = 0.0037f;                          // CLOCK_ALPHA
    c->clock_gain_omega = (0.0037f * 0.0037f) / 4.0f;  // CLOCK_GAIN_OMEGA
    c->clock_omega_limit = 0.005f;                     // CLOCK_OMEGA_LIMIT
}

void xrit_demod_config_lrit(xrit_demod_config *c, float sample_rate, uint32_t decimation)
{
    config_common(c, sample_rate, decimation);
    c->symbol_rate = 293883;   // LRIT_SYMBOL_RATE
    c->rrc_alpha = 0.5f;       // LRIT_RRC_ALPHA
}

void xrit_demod_config_hrit(xrit_demod_config *c, float sample_rate, uint32_t decimation)
{
    config_common(c, sample_rate, decimation);
    c->symbol_rate = 927000;   // HRIT_SYMBOL_RATE
    c->rrc_alpha = 0.3f;       // HRIT_RRC_ALPHA
}

// A stream with a hardware queue of its own.  HIP deals ordinary streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4) by
// its own count of what the process has created so far, and two streams of a handle that land in one queue serialise: the
// walkers of two bursts behind one another, or a Costas loop behind a burst's walkers -- measured late in round 5: the second and
// the fourth handle of a process ran their bursts in 2.5 ms instead of 1.8, and so did a handle created after others had been
// destroyed.  A stream created with a CU mask gets a queue of its own; the mask here is every CU.  (Not a guarantee either: with
// eight handles alive in one process -- 16 such queues beside HIP's pools -- the fourth and later ones were slow again, hardware
// queue slots being what they are; the first three handles of a process run at the same speed, which HIP's own dealing did not
// give the second one.  XRIT_SHARED_QUEUES=1: plain hipStreamCreate.)
static hipError_t create_own_queue_stream(hipStream_t *s)
{
    if (getenv("XRIT_SHARED_QUEUES")) return hipStreamCreate(s);      // (A/B: HIP's own dealing)
    uint32_t mask[32];
    for (auto &m : mask) m = 0xffffffffu;
    int cus = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    if (cus > 32 * 32) cus = 32 * 32;            // (the mask array holds 32 words)
    const uint32_t words = (uint32_t)((cus + 31) / 32);
    if (cus % 32) mask[words - 1] = (1u << (cus % 32)) - 1u;
    const hipError_t e = hipExtStreamCreateWithCUMask(s, words, mask);
    if (e == hipSuccess) return e;
    (void)hipGetLastError();
    return hipStreamCreate(s);
}

static bool create_walker_streams(xrit_demod *d)
{
    for (auto &w : d->stream3) if (create_own_queue_stream(&w) != hipSuccess) return false;
    return true;
}

// The streams of a handle are kept when the handle is destroyed and given to the next handle created on that device (round 6).
// How fast a handle runs depends on which hardware queues its streams sit on (above), and what HIP deals a NEW stream depends
// on everything the process has created before: a handle created after others had been destroyed ran its bursts in 2.5 ms
// instead of 1.8 (round 5, profiles/r5_handles_in_one_process.txt).  With the pool such a handle runs on the very streams --
// the very queues -- of the one before it: create / destroy sequences are as fast as the first handle, whatever their length.
// (Handles that are alive TOGETHER still take a set each: more queues than the hardware has slots for are time-sliced.)
namespace {
struct StreamSet {
    int device = -1;
    hipStream_t s = nullptr, s2 = nullptr, s3[XRIT_WALK_STREAMS] = {}, sc = nullptr;
};
std::mutex g_pool_mu;
std::vector<StreamSet> g_pool;

bool stream_set_acquire(int device, StreamSet *out)
{
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        for (size_t i = 0; i < g_pool.size(); ++i)
            if (g_pool[i].device == device) { *out = g_pool[i]; g_pool.erase(g_pool.begin() + (long)i); return true; }
    }
    StreamSet n;
    n.device = device;
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    // (the second stream at the lowest priority: a hardware queue of its own -- streams of one priority share a few
    // queues round-robin, and a front end that lands in the queue of the caller's stream does not overlap anything --
    // and the loops of the current burst go first where the two compete)
    bool ok = hipStreamCreate(&n.s) == hipSuccess && hipStreamCreateWithPriority(&n.s2, hipStreamDefault, prio_least) == hipSuccess;
    for (auto &w : n.s3) ok = ok && create_own_queue_stream(&w) == hipSuccess;
    ok = ok && create_own_queue_stream(&n.sc) == hipSuccess;
    if (!ok) {
        if (n.s) (void)hipStreamDestroy(n.s);
        if (n.s2) (void)hipStreamDestroy(n.s2);
        for (auto w : n.s3) if (w) (void)hipStreamDestroy(w);
        if (n.sc) (void)hipStreamDestroy(n.sc);
        return false;
    }
    *out = n;
    return true;
}

void stream_set_release(const StreamSet &st)
{
    if (st.s) (void)hipStreamSynchronize(st.s);
    if (st.s2) (void)hipStreamSynchronize(st.s2);
    for (auto w : st.s3) if (w) (void)hipStreamSynchronize(w);
    if (st.sc) (void)hipStreamSynchronize(st.sc);
    std::lock_guard<std::mutex> lk(g_pool_mu);
    g_pool.push_back(st);
}
}  // namespace

int xrit_demod_create(const xrit_demod_config *cfg, xrit_demod **out)
{
    if (!cfg || !out) { set_error("null argument"); return XRIT_E_INVALID; }
    *out = nullptr;
    if (cfg->decimation < 1 || cfg->symbol_rate == 0 || !(cfg->sample_rate > 0)) {
        set_error("invalid configuration");
        return XRIT_E_INVALID;
    }
    if (cfg->costas_chain_len != 0 && (cfg->costas_chain_len < 16 || cfg->costas_chain_len > 320)) {
        set_error("costas_chain_len = %d: 0 (= 256) or 16..320 samples; beyond that the hand-off solve does not converge reliably",
                  cfg->costas_chain_len);
        return XRIT_E_INVALID;
    }
    if (cfg->rrc_taps < 3) {
        set_error("rrc_taps = %d: the matched filter needs at least 3 taps (RRC_TAPS is 63, Parameters.h:28)", cfg->rrc_taps);
        return XRIT_E_INVALID;
    }
    // the AGC's gain recurrence is evaluated as a scan of maps g -> min(a g + b, c), valid for positive gains
    if (!(cfg->agc_rate > 0) || !(cfg->agc_reference > 0) || !(cfg->agc_gain > 0) || !(cfg->agc_max_gain >= 0)) {
        set_error("AGC rate, reference and initial gain must be positive (Parameters.h:34-37: 0.01, 0.5, 1, 4000)");
        return XRIT_E_INVALID;
    }
    if (cfg->front_exact < -1 || cfg->front_exact > 2) {
        set_error("front_exact = %d: -1 (never), 0 (below the big-burst size), 1 or 2 (always)", cfg->front_exact);
        return XRIT_E_INVALID;
    }
    if (!(cfg->sample_rate / (float)cfg->decimation / (float)cfg->symbol_rate >= 1.0f)) {
        set_error("fewer than one sample per symbol after decimation");
        return XRIT_E_INVALID;
    }
    XR_TRY(select_device(cfg->device));
    xrit_demod *d = new (std::nothrow) xrit_demod();
    if (!d) return XRIT_E_NOMEM;
    d->cfg = *cfg;
    d->device = cfg->device;
    // demodulator.cpp:436-437 (float arithmetic as written there)
    d->circuit_rate = cfg->sample_rate / ((float)cfg->decimation);
    d->sps = d->circuit_rate / ((float)cfg->symbol_rate);
    int rc = XRIT_OK;
    do {
        {
            StreamSet st;
            if (!stream_set_acquire(d->device, &st)) { set_error("hipStreamCreate failed"); rc = XRIT_E_HIP; break; }
            d->stream = st.s; d->stream2 = st.s2; d->stream_c_pool = st.sc;
            for (int i = 0; i < XRIT_WALK_STREAMS; ++i) d->stream3[i] = st.s3[i];
            d->pooled = true;
        }
        if (hipEventCreateWithFlags(&d->ev_done, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&d->ev_ready, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&d->ev_fe[0], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&d->ev_fe[1], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&d->ev_relay, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&d->ev_costas, hipEventDisableTiming) != hipSuccess) { set_error("hipEventCreate failed"); rc = XRIT_E_HIP; break; }
        std::vector<float> rrc = design_rrc(1, d->circuit_rate, cfg->symbol_rate, cfg->rrc_alpha, cfg->rrc_taps);
        std::vector<float> lp = design_lowpass(1, cfg->sample_rate, d->circuit_rate / 2, 100e3);
        d->dec_ntaps = (int)lp.size();
        // (cfg.front_exact = 2: both filters summed in the CPU chain's order, the AGC and the Costas loop walked literally --
        // the front end bit for bit the CPU chain's through the Costas loop; fir.hip, agc.hip, costas_exact.hip)
        d->dec.exact = d->rrc.exact = d->agc.exact = d->costas.exact = cfg->front_exact == 2;
        if (cfg->front_exact == 2 && !getenv("XRIT_NO_COSTAS_STREAM")) d->stream_c = d->stream_c_pool;
        // (wherever such a call takes the bit-exact front end -- the default and parity mode --, a call of up to 200 k symbols, i.e.
        // every chunk size of the reference, is ONE exact walk of the clock recovery: the CPU chain's words.  With the fast
        // front end on calls of every size, cfg.front_exact = -1 / 1, the limit stays where one walk costs no more than the
        // relay, 73.7 k symbols: there is no word to keep.)
        if (cfg->front_exact == 2 || cfg->front_exact == 0) d->clock.one_walk_max = 200000;
        if ((rc = d->dec.init(lp.data(), (int)lp.size(), (int)cfg->decimation)) != XRIT_OK) break;
        if ((rc = d->rtl.init(cfg->sample_rate)) != XRIT_OK) break;
        if ((rc = d->agc.init(cfg->agc_rate, cfg->agc_reference, cfg->agc_gain, cfg->agc_max_gain)) != XRIT_OK) break;
        if ((rc = d->rrc.init(rrc.data(), (int)rrc.size(), 1)) != XRIT_OK) break;
        if ((rc = d->costas.init(cfg->pll_alpha, cfg->costas_chain_len, cfg->max_passes)) != XRIT_OK) break;
        // (cfg.front_exact, the opt-in parity mode: the final pass starts every chain four chains early)
        if (cfg->front_exact == 1) d->costas.final_warm = 4;
        if ((rc = d->clock.init(d->sps, cfg->clock_gain_omega, cfg->clock_mu, cfg->clock_alpha, cfg->clock_omega_limit,
                                cfg->clock_chain_syms, cfg->max_passes > 0 ? cfg->max_passes : 0)) != XRIT_OK) break;
        d->clock.serial = cfg->clock_serial != 0;
        d->clock.exact = cfg->clock_exact == -3 ? 0 : cfg->clock_exact;        // (-3: the default's plan, its first relay passes approximate)
        d->clock.relay_quick = cfg->clock_exact == -3;
#ifdef XRIT_EXPERIMENTS
        d->no_defer = getenv("XRIT_NO_DEFER") != nullptr;
#endif
#ifdef XRIT_EXPERIMENTS
        if (const char *e = getenv("XRIT_OV_CSTREAM_BELOW")) d->costas_own_stream_below = (size_t)atoll(e);
#endif
        d->clock.relay_window = cfg->clock_exact_window > 0 ? cfg->clock_exact_window : 0;
        if (cfg->clock_min_passes > 0)
            d->clock.min_passes = cfg->clock_min_passes < d->clock.max_passes ? cfg->clock_min_passes : d->clock.max_passes;
    } while (0);
    if (rc != XRIT_OK) { xrit_demod_destroy(d); return rc; }
    *out = d;
    return XRIT_OK;
}

void xrit_demod_destroy(xrit_demod *d)
{
    if (!d) return;
    (void)hipSetDevice(d->device);
    if (d->stream2) (void)hipStreamSynchronize(d->stream2);     // a front end that ran ahead may still be at work
    if (d->stream_c) (void)hipStreamSynchronize(d->stream_c);
    for (auto w : d->stream3) if (w) (void)hipStreamSynchronize(w);     // ... or walkers
    if (d->stream) (void)hipStreamSynchronize(d->stream);
    d->dec.release(); d->rrc.release(); d->agc.release(); d->costas.release(); d->clock.release();
    if (d->stream_c_pool) (void)hipStreamSynchronize(d->stream_c_pool);
    if (d->ev_done) (void)hipEventDestroy(d->ev_done);
    if (d->ev_ready) (void)hipEventDestroy(d->ev_ready);
    if (d->ev_relay) (void)hipEventDestroy(d->ev_relay);
    if (d->ev_costas) (void)hipEventDestroy(d->ev_costas);
    for (int i = 0; i < 2; ++i) if (d->ev_fe[i]) (void)hipEventDestroy(d->ev_fe[i]);
    for (int i = 0; i < 2; ++i) { d->bufA[i].release(); d->bufB[i].release(); d->bufC[i].release(); d->bufR[i].release(); d->stat[i].release(); }
    d->rtl.release();
    d->in_dev.release(); d->soft_dev.release();
    d->q_in.release(); d->q_out.release();
    for (auto &b : d->stage_buf) b.release();
    if (d->pooled) {
        StreamSet st;
        st.device = d->device; st.s = d->stream; st.s2 = d->stream2; st.sc = d->stream_c_pool;
        for (int i = 0; i < XRIT_WALK_STREAMS; ++i) st.s3[i] = d->stream3[i];
        stream_set_release(st);
    }
    delete d;
}

int xrit_demod_reset(xrit_demod *d, void *stream)
{
    if (!d) { set_error("null argument"); return XRIT_E_INVALID; }
    XR_HIP(hipSetDevice(d->device));
    hipStream_t s = stream ? (hipStream_t)stream : d->stream;
    XR_HIP(hipStreamSynchronize(d->stream2));       // a front end that ran ahead belongs to the stream being left
    if (d->stream_c) XR_HIP(hipStreamSynchronize(d->stream_c));
    for (auto w : d->stream3) XR_HIP(hipStreamSynchronize(w));      // ... and so do walkers
    if (d->costas.job.n && d->pf_count > 0) {
        // (a Costas loop begun ahead and never looked at: the stage's bookkeeping is brought to an end before its state is reset)
        for (int i = 0; i < d->pf_count; ++i)
            if (d->pf[i].costas_begun && !d->pf[i].costas_finished) { bool redone = false; (void)d->costas.finish(d->pf[i].c_stream ? d->pf[i].c_stream : d->stream2, nullptr, &redone); }
    }
    d->pf_count = 0;
    d->last_fe_set = -1;
    XR_TRY(d->dec.reset(s));
    XR_TRY(d->rrc.reset(s));
    XR_TRY(d->agc.reset(s));
    XR_TRY(d->rtl.reset(s));
    XR_TRY(d->costas.reset(s));
    XR_TRY(d->clock.reset(s));
    d->poisoned = false;
    return XRIT_OK;
}

int xrit_build_experiments(void)
{
#ifdef XRIT_EXPERIMENTS
    return 1;
#else
    return 0;
#endif
}

void *xrit_demod_stream(xrit_demod *d) { return d ? (void *)d->stream : nullptr; }
float xrit_demod_sps(const xrit_demod *d) { return d ? d->sps : 0.f; }
int xrit_demod_decimator_ntaps(const xrit_demod *d) { return d ? d->dec_ntaps : 0; }

static int keep_stage(xrit_demod *d, int idx, const void *src, size_t n, hipStream_t s)
{
    if (!d->keep_stages) return XRIT_OK;
    XR_TRY(d->stage_buf[idx].reserve((n + 1) * sizeof(float2)));
    if (n) XR_HIP(hipMemcpyAsync(d->stage_buf[idx].p, src, n * sizeof(float2), hipMemcpyDeviceToDevice, s));
    d->stage_n[idx] = n;
    return XRIT_OK;
}

// ---- one time slice through the chain, in two halves ------------------------------------------------------
// front_end(): ingest conversion, decimator, AGC, RRC (demodulator.cpp:54-74,136-149) into buffer set `set`;
// loops(): Costas and clock recovery (:152-157) on what front_end() left.  Split so that the front end of the
// next slice can run on a second stream while the feedback loops of this one iterate.
struct SliceIO {
    size_t length = 0;          // circuit-rate samples
    const float2 *rrc = nullptr;
    bool stat_ready = false;
    int set = 0;                // which set of front-end buffers
    const float *agc_flag = nullptr;    // the AGC guard flag of THIS front end (the stage's slot moves on with the next)
    bool exact = false;         // this call's front end is the bit-exact one (cfg.front_exact = 2, or 0 on a call below the big-burst size)
};

// Which front end a call of n input samples takes.  cfg.front_exact = 2: the bit-exact one, always; 0 (default, round 6): the
// bit-exact one for calls below the big-burst size -- the reference's own chunk sizes (32 Ki .. 512 Ki samples) and everything
// up to a million symbols, where a call is launch latency and not throughput: the soft symbols of a call of up to 74 k symbols
// are then the CPU chain's word for word -- and the fast one for bursts of a million symbols and more, the ones `value` is
// quoted on; 1: the fast one with the Costas loop's final pass warmed up; -1: the fast one, always (round 5's default).
static bool front_exact_call(const xrit_demod *d, size_t n)
{
    if (d->cfg.front_exact == 2) return true;
    if (d->cfg.front_exact != 0) return false;
    const unsigned D = d->cfg.decimation;
    const double length = (double)(D > 1 ? n / D : n);
    return length / (double)d->sps < (double)d->clock.ov_min;
}

int xrit_demod_front_exact_for(const xrit_demod *d, size_t n)
{
    if (!d) return XRIT_E_INVALID;
    return front_exact_call(d, n) ? 1 : 0;
}

static int front_end(xrit_demod *d, const void *in, size_t n, int type, int set, hipStream_t s, Profiler *prof, SliceIO *io)
{
    io->set = set;
    const bool ex = front_exact_call(d, n);
    io->exact = ex;
    const unsigned D = d->cfg.decimation;
    if (type == XRIT_SAMPLE_U8IQ) {
        // RtlFrontend::internalCallback (RtlFrontend.cpp:102-116) hands FLOATIQ to onSamplesAvailable
        XR_TRY(d->bufR[set].reserve((n + 8) * sizeof(float2)));
        XR_TRY(d->rtl.run(in, d->bufR[set].as<float2>(), n, s, prof));
        in = d->bufR[set].p;
        type = XRIT_SAMPLE_FLOATIQ;
    }
    size_t length = n;
    if (D > 1) length = n / D;   // demodulator.cpp:137 -- the remainder of the chunk is dropped
    io->length = length;
    XR_TRY(d->bufA[set].reserve((length + 8) * sizeof(float2)));
    XR_TRY(d->bufB[set].reserve((length + 8) * sizeof(float2)));
    float2 *A = d->bufA[set].as<float2>(), *B = d->bufB[set].as<float2>();
    const float2 *cur = nullptr;
    // with a decimator in front, its epilogue leaves the AGC's composed gain maps: the AGC sweeps the stream
    // twice (scan of the maps aside) instead of three times
    const bool agc_fused = !ex && D > 1 && length > 0 && d->dec.agc_supported();
    if (D > 1) {
        AgcEpilogue epi{};
        if (agc_fused) XR_TRY(d->agc.fused_begin(length, d->dec.RC, s, &epi));
        XR_TRY(d->dec.run(in, type, A, length, s, prof, nullptr, 0, agc_fused ? &epi : nullptr, nullptr, ex));          // :138
        cur = A;
    } else if (type != XRIT_SAMPLE_FLOATIQ) {
        ProfScope ps(prof, "convert", s);
        XR_TRY(launch_convert(in, type, A, length, s));            // :57-70
        cur = A;
    } else {
        cur = reinterpret_cast<const float2 *>(in);
    }
    XR_TRY(keep_stage(d, 0, cur, length, s));
    // ... and when nobody asks for the AGC output itself, the matched filter applies the gains while it fills its
    // window: the AGC then costs the stream no sweep of its own at all
    const bool agc_in_rrc = agc_fused && !d->keep_stages && d->rrc.agc_fill_supported(d->dec.RC);
    // no decimator (C1, C3), or one whose kernel has no such epilogue (the polyphase one, C5): the run maps come
    // from one read-only sweep, the rest is the same
    const bool agc_in_rrc_d1 = !ex && !agc_fused && length > 0 && !d->keep_stages && d->rrc.agc_fill_supported(3);
    AgcFill fill{};
    float2 *Cfb = nullptr;      // where the serial fallback would put the AGC output (guard tripped)
    if (agc_in_rrc || agc_in_rrc_d1) {
        XR_TRY(d->bufC[set].reserve((length + 8) * sizeof(float2)));
        Cfb = d->bufC[set].as<float2>();
        if (agc_in_rrc_d1) XR_TRY(d->agc.fused_reduce(cur, length, 3, s, prof));
        XR_TRY(d->agc.fused_scan(cur, Cfb, length, agc_in_rrc ? d->dec.RC : 3, s, prof, &fill));    // :143
    }
    else if (agc_fused) XR_TRY(d->agc.fused_finish(cur, B, length, d->dec.RC, s, prof));
    else XR_TRY(d->agc.run(cur, B, length, s, prof, ex));
    XR_TRY(keep_stage(d, 1, B, length, s));
    // the RRC epilogue leaves the per-chain statistic of the Costas guess, the Costas final pass the
    // timing-line statistic of the clock-recovery guess: neither stage sweeps its input once more for it
    // (sum z^2 per run of 8 outputs: the chain sums and the loop's sub-block model both come from it, costas.hip)
    constexpr int SUB = 8;
    XR_TRY(d->stat[set].reserve(((length + SUB - 1) / SUB + 2) * sizeof(float2)));
    io->stat_ready = length > 0 && d->rrc.stat_supported(SUB) && d->costas.L % SUB == 0;
    // (fused: A holds the decimator output, which the fill reads; the filter output goes to B's place instead)
    const bool fill_on = agc_in_rrc || agc_in_rrc_d1;
    float2 *rrc_out = fill_on ? B : A;
    XR_TRY(d->rrc.run(fill_on ? Cfb : B, XRIT_SAMPLE_FLOATIQ, rrc_out, length, s, prof, io->stat_ready ? d->stat[set].as<float2>() : nullptr, SUB,
                      nullptr, fill_on ? &fill : nullptr, ex)); // :148
    XR_TRY(keep_stage(d, 2, rrc_out, length, s));
    io->rrc = rrc_out;
    io->agc_flag = d->agc.state.as<float>() + 2 * d->agc.cur + 1;
    return XRIT_OK;
}

static bool ov_call(xrit_demod *d, size_t n);

// the front end of a registered input on stream2 (xrit_demod_prefetch_device); `after`: not before this event
static int launch_prefetched(xrit_demod *d, xrit_demod::Prefetched &f, hipEvent_t after)
{
    // stream2 may read the input once whatever the caller queued on its stream at registration is done, and may touch
    // the stages once the previous front end (on either stream) is done
    XR_HIP(hipStreamWaitEvent(d->stream2, d->ev_ready, 0));
    if (after) XR_HIP(hipStreamWaitEvent(d->stream2, after, 0));
    if (d->last_fe_set >= 0) XR_HIP(hipStreamWaitEvent(d->stream2, d->ev_fe[d->last_fe_set], 0));
    const int set = d->next_set;
    d->next_set ^= 1;
    SliceIO io;
    Profiler *prof = d->prof.enabled ? &d->prof : nullptr;
    // (the filters' waves in front of the walkers at issue -- where the walkers' latency is hidden: bursts that walk overlapping blocks)
    d->dec.prio = d->rrc.prio = ov_call(d, f.n) ? 1 : 0;
    int rc = front_end(d, f.samples, f.n, f.type, set, d->stream2, prof, &io);
    d->dec.prio = d->rrc.prio = 0;
    if (rc != XRIT_OK) { d->poisoned = true; return rc; }
    XR_HIP(hipEventRecord(d->ev_fe[set], d->stream2));
    d->last_fe_set = set;
    f.set = set;
    f.length = io.length; f.rrc = io.rrc; f.stat_ready = io.stat_ready; f.agc_flag = io.agc_flag; f.exact = io.exact;
    f.launched = true;
    return XRIT_OK;
}

// the Costas loop of a slice (demodulator.cpp:152): guess, a batch of passes with a device-side stop test and the final pass,
// enqueued on s; it writes into the clock recovery's (next) input buffer
static int costas_enqueue(xrit_demod *d, const SliceIO &io, hipStream_t s, Profiler *prof, float2 **slot_out)
{
    const size_t length = io.length;
    const int L = d->costas.L;
    float2 *slot = nullptr;
    XR_TRY(d->clock.input_slot(length, &slot, s));
    double2 *om = nullptr;
    if (length) om = d->clock.om_slot((int)((length + (size_t)L - 1) / (size_t)L), L);
    const double inv_sps = 1.0 / (double)d->sps;
    const float2 *stat = io.stat_ready ? d->stat[io.set].as<float2>() : nullptr;
    XR_TRY(d->costas.begin(io.rrc, slot, length, s, prof, stat, om, 0, inv_sps, io.exact));
    if (length) XR_TRY(d->clock.om_scan(s));     // the timing guess's count curve, behind the final pass that leaves its statistic
    if (length) XR_TRY(d->agc.request_flag_at(io.agc_flag, s));   // the AGC's guard flag rides along: no wait of its own
    *slot_out = slot;
    return XRIT_OK;
}

// (a registered front end of the next burst goes in front of this call's relay kernels -- and behind it, where this call's
// own Costas loop is done with the stage, the next burst's Costas loop)
static void set_relay_hook(xrit_demod *d, hipStream_t s)
{
    d->clock.before_relay = [d, s](int phase) -> int {
        // phase 0: in front of the relay kernels (an event marks the spot); phase 1: once they are enqueued
        if (phase == 0) {
            if (d->pf_count > 0 && !d->pf[0].launched) XR_HIP(hipEventRecord(d->ev_relay, s));
            return XRIT_OK;
        }
        if (d->pf_count > 0 && !d->pf[0].launched) {
            if (d->clock.trace_env) fprintf(stderr, "[xrit] the registered front end starts with the relay kernels%s\n", d->costas_idle ? ", its Costas loop behind it" : "");
            XR_TRY(launch_prefetched(d, d->pf[0], d->ev_relay));
            if (d->costas_idle) {
                xrit_demod::Prefetched &f = d->pf[0];
                SliceIO io;
                io.length = f.length; io.rrc = f.rrc; io.stat_ready = f.stat_ready; io.set = f.set; io.agc_flag = f.agc_flag; io.exact = f.exact;
                int rc = costas_enqueue(d, io, d->stream2, d->prof.enabled ? &d->prof : nullptr, &f.slot);
                if (rc != XRIT_OK) { d->poisoned = true; return rc; }
                XR_HIP(hipEventRecord(d->ev_costas, d->stream2));
                f.costas_begun = true;
            }
        }
        return XRIT_OK;
    };
}

// clock recovery (:156, SymbolManager.cpp:104) of a slice whose Costas loop has been enqueued on s (costas_done: has been
// finished ahead of this call, on stream2)
static void costas_note(xrit_demod *d)
{
    d->costas_seen.passes = d->costas.passes;
    d->costas_seen.unconverged = d->costas.unconverged;
    d->costas_seen.max_residual = d->costas.max_residual;
    d->costas_seen.walked = d->costas.job.rescued && d->costas.walked;
}

static int loops(xrit_demod *d, const SliceIO &io, float *d_soft, size_t cap, size_t *nsym, hipStream_t s, Profiler *prof,
                 bool costas_done = false, float2 *slot_done = nullptr)
{
    const size_t length = io.length;
    float2 *slot = slot_done;
    // Costas (:152) and clock recovery are enqueued back to back -- guesses, a batch of hand-off passes each with a
    // device-side stop test, final / output passes -- and the host waits once.  Only when a batch did not close (cold
    // start, unlocked input) does it continue pass by pass.
    if (!costas_done) XR_TRY(costas_enqueue(d, io, s, prof, &slot));
    if (!costas_done && d->pf_count > 0 && !d->pf[0].launched) {
        // A stream with its next input registered: this call's Costas loop is finished FIRST (one more wait of the host), so
        // that the stage is free for the next burst's loop behind its front end, under this call's relay -- from the next
        // call on that is how every Costas loop of the stream runs and the extra wait is gone.
        XR_HIP(hipEventRecord(d->ev_costas, s));
        XR_HIP(hipEventSynchronize(d->ev_costas));
        if (length && d->agc.requested_flag() == 2.0f) d->agc_fallback_seen = true;
        bool redone = false;
        XR_TRY(d->costas.finish(s, prof, &redone));
        costas_note(d);
        if (redone) d->clock.om_scanned = false;     // (the final pass ran again: the statistic is new, begin() unwraps it itself)
        costas_done = true;
    }
    float2 *sym = nullptr;
    if (d->keep_stages || d->keep_symbols) {
        XR_TRY(d->stage_buf[4].reserve((cap + 1) * sizeof(float2)));
        sym = d->stage_buf[4].as<float2>();
    }
    d->costas_idle = costas_done;
    set_relay_hook(d, s);
    const int rc_begin = d->clock.begin(length, d_soft, sym, cap, s, prof);
    d->clock.before_relay = nullptr;
    d->costas_idle = false;
    XR_TRY(rc_begin);
    XR_HIP(hipStreamSynchronize(s));
    if (!costas_done) {
        if (length && d->agc.requested_flag() == 2.0f) d->agc_fallback_seen = true;
        bool redone = false;
        XR_TRY(d->costas.finish(s, prof, &redone));
        costas_note(d);
        if (redone) {
            // the Costas output was rewritten after more passes: the clock recovery starts over on it
            const int L = d->costas.L;
            if (length) (void)d->clock.om_slot((int)((length + (size_t)L - 1) / (size_t)L), L);
            XR_TRY(d->clock.begin(length, d_soft, sym, cap, s, prof));
            XR_HIP(hipStreamSynchronize(s));
        }
    }
    XR_TRY(keep_stage(d, 3, slot, length, s));
    int rc = d->clock.finish(nsym, s, prof);
    if (d->keep_stages || d->keep_symbols) XR_HIP(hipStreamSynchronize(s));
    return rc;
}

// ---- round 5: bursts whose clock recovery walks overlapping blocks (clock_overlap.h) -----------------------------------------
// Nothing in such a burst's clock recovery waits for the burst in front of it, so EVERYTHING up to its walkers runs ahead of
// its process call: front end and Costas loop on stream2 (in the order of the inputs: the stages carry their state from one
// to the next), the walkers on stream3 behind the Costas loop.  With two inputs registered behind the current call
// (xrit_demod_prefetch_device) the device holds, at any time, the walkers of bursts b and b + 1 and the front end / Costas loop
// of burst b + 2; a process call enqueues what the newest registration allows, waits for its own walkers and lays out its
// symbols (ClockStage::ov_finalize) -- the relay's latency, which bounded a burst in round 4, is hidden.
static bool ov_call(xrit_demod *d, size_t n)
{
    const unsigned D = d->cfg.decimation;
    d->clock.ov_allow = !(d->keep_stages || d->keep_symbols);
    return d->clock.ov_eligible(D > 1 ? n / D : n);
}

// whatever can be enqueued for the registered inputs, oldest first, without waiting for the device.  Returns an error code;
// *progress: something was enqueued or finished.
static int ov_service(xrit_demod *d, bool *progress, int limit = 1 << 30)
{
    Profiler *prof = d->prof.enabled ? &d->prof : nullptr;
    if (progress) *progress = false;
    bool costas_busy = false;       // a Costas loop begun and not yet finished: the stage takes the next input behind it
    for (int i = 0; i < d->pf_count && i < limit; ++i) {
        xrit_demod::Prefetched &f = d->pf[i];
        if (!ov_call(d, f.n)) break;                    // (inputs that take the other path are started by their own calls)
        // (two sets of front-end buffers: the input two in front used the set this one takes, and its Costas loop reads that set
        // until the host has seen it close -- a loop that did not close inside its batch goes on from the host)
        if (!f.launched && (i < 2 || d->pf[i - 2].costas_finished)) {
            XR_TRY(launch_prefetched(d, f, nullptr));
            if (progress) *progress = true;
        }
        if (!f.launched) break;
        if (f.costas_begun && !f.costas_finished) {
            if (hipEventQuery(d->ev_costas) != hipSuccess) { costas_busy = true; continue; }
            if (f.length && d->agc.requested_flag() == 2.0f) f.agc_fallback = true;
            bool redone = false;
            int rc = d->costas.finish(f.c_stream ? f.c_stream : d->stream2, prof, &redone);
            if (rc != XRIT_OK) { d->poisoned = true; return rc; }
            f.c_passes = d->costas.passes; f.c_unconverged = d->costas.unconverged; f.c_max_residual = d->costas.max_residual;
            f.c_walked = d->costas.job.rescued && d->costas.walked;
            f.costas_finished = true;
            if (redone && f.ov_job >= 0) {
                // the loop went on from the host and rewrote its output (and the timing statistic): walkers that were started
                // behind the first batch have read the old one
                rc = d->clock.ov_restart(f.ov_job, d->stream2);
                if (rc != XRIT_OK) { d->poisoned = true; return rc; }
                f.walk_launched = false;
            }
            if (progress) *progress = true;
        }
        if (!f.costas_begun && !costas_busy) {
            SliceIO io;
            io.length = f.length; io.rrc = f.rrc; io.stat_ready = f.stat_ready; io.set = f.set; io.agc_flag = f.agc_flag; io.exact = f.exact;
            // (on a stream of its own behind this input's front end: the front end of the input behind it runs beside this loop)
            // (on the front ends' stream, behind this input's: measured with every stream on a hardware queue of its own
            // -- GPU_MAX_HW_QUEUES=8 -- a Costas stream beside the front ends' costs 10 %, 2.05 against 1.85 ms per C2 burst: the
            // loop's passes and the next input's decimator fill the chip each and only slow one another)
            // (... unless the loop is a handful of small kernels -- C5: 2 M samples at the circuit rate, 0.3 ms of launches --, which
            // then overlap the next input's decimator: 1.1 -> 0.9 ms per C5 burst)
            // (round 5, late: not a stream of its own but the walker stream this burst's OWN walkers will take -- the one the
            // walkers of the burst two in front are on: the loop then runs behind them instead of beside them, and its walkers
            // behind it.  That is what HIP's dealing of streams onto hardware queues had arranged by accident on the first handle
            // of a process, and measured better than a queue of its own: C5 1.08 against 1.11-1.14 ms per burst)
            // (round 6, cfg.front_exact = 2: the exact AGC chains and the exact Costas walkers are a wave per SIMD or less for
            // milliseconds -- latency, not work --, so the Costas loop of burst b runs on a stream of its own beside the front end
            // of burst b + 1 instead of in front of it)
            f.costas_stream = d->stream_c ? 2 : (f.length < d->costas_own_stream_below ? 1 : 0);
            hipStream_t sc = f.costas_stream == 2 ? d->stream_c : f.costas_stream ? d->stream3[(d->clock.ov_serial + 1) % XRIT_WALK_STREAMS] : d->stream2;
            f.c_stream = sc;
            if (f.costas_stream) XR_HIP(hipStreamWaitEvent(sc, d->ev_fe[f.set], 0));
            int rc = costas_enqueue(d, io, sc, prof, &f.slot);
            if (rc != XRIT_OK) { d->poisoned = true; return rc; }
            XR_HIP(hipEventRecord(d->ev_costas, sc));
            f.costas_begun = true;
            f.ov_job = d->clock.ov_job;
            costas_busy = true;
            if (progress) *progress = true;
        }
        if (f.costas_begun && f.ov_job >= 0 && !f.walk_launched && d->clock.ov_can_launch_ahead(f.ov_job)) {
            // (behind the Costas loop's batch and the timing curve: speculative until the host has seen the loop's stop test)
            // (two walker streams, alternating: the walkers of bursts b and b + 1 side by side, those of b + 2 behind b's)
            hipStream_t sw = d->stream3[d->clock.ov[f.ov_job].serial % XRIT_WALK_STREAMS];
            XR_HIP(hipStreamWaitEvent(sw, d->ev_costas, 0));
            int rc = d->clock.ov_launch(f.ov_job, sw, true, prof);
            if (rc != XRIT_OK) { d->poisoned = true; return rc; }
            f.walk_launched = true;
            if (progress) *progress = true;
        }
        if (!f.costas_begun || !f.costas_finished) costas_busy = true;      // (the inputs behind wait for this one's loop)
    }
    return XRIT_OK;
}

static int process_overlap(xrit_demod *d, const void *d_samples, size_t n, int type, float *d_soft, size_t cap, size_t *n_out,
                           hipStream_t s, Profiler *prof)
{
    // the call's own input goes through the same queue: registered now if it was not registered ahead
    if (d->pf_count > 0) {
        if (d->pf[0].samples != d_samples || d->pf[0].n != n || d->pf[0].type != type) {
            set_error("process calls must take the prefetched inputs in the order they were prefetched");
            d->poisoned = true;
            return XRIT_E_INVALID;
        }
    } else {
        XR_HIP(hipEventRecord(d->ev_ready, s));
        xrit_demod::Prefetched &f = d->pf[0];
        f = xrit_demod::Prefetched{};
        f.samples = d_samples; f.n = n; f.type = type; f.launched = false;
        d->pf_count = 1;
    }
    auto fail = [&](int rc) { d->poisoned = true; *n_out = 0; return rc; };
    int rc = ov_service(d, nullptr);
    if (rc != XRIT_OK) return fail(rc);
    // this call's Costas loop must have been looked at before its clock recovery is laid out
    // (only THIS input here: what the host enqueues for the inputs behind it -- a Costas loop, a front end: 0.2 ms of launches --
    // goes behind this call's walkers, which are on the critical path when nothing ran ahead)
    for (int spins = 0; !d->pf[0].costas_finished; ++spins) {
        if (d->pf[0].costas_begun) { if (hipEventSynchronize(d->ev_costas) != hipSuccess) { set_error("hipEventSynchronize failed"); return fail(XRIT_E_HIP); } }
        if ((rc = ov_service(d, nullptr, 1)) != XRIT_OK) return fail(rc);
        if (spins > 8) { set_error("the Costas loop of the call could not be started"); return fail(XRIT_E_INVALID); }
    }
    xrit_demod::Prefetched &f0 = d->pf[0];
    d->agc_fallback_seen = f0.agc_fallback;
    d->costas_seen.passes = f0.c_passes; d->costas_seen.unconverged = f0.c_unconverged;
    d->costas_seen.max_residual = f0.c_max_residual; d->costas_seen.walked = f0.c_walked;
    // joints, output and result behind this call's walkers (started here if they did not run ahead), on the call's stream.
    // (the host has SEEN the end of everything stream2 did for this input -- the Costas loop's event, a continued loop's
    // synchronise --, so s needs no event of stream2's; the finalize kernels wait for the walkers' event)
    rc = d->clock.begin(f0.length, d_soft, nullptr, cap, s, prof);
    if (rc != XRIT_OK) return fail(rc);
    if (hipEventRecord(d->ev_done, s) != hipSuccess) { set_error("hipEventRecord failed"); return fail(XRIT_E_HIP); }
    if ((rc = ov_service(d, nullptr)) != XRIT_OK) return fail(rc);
    // while the walkers finish: keep the inputs behind this one moving (the Costas loop of the next one may close meanwhile,
    // which frees the stage for the one after it)
    for (;;) {
        const hipError_t q = hipEventQuery(d->ev_done);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) { set_error("hipEventQuery failed: %s", hipGetErrorString(q)); return fail(XRIT_E_HIP); }
        bool progress = false;
        if ((rc = ov_service(d, &progress)) != XRIT_OK) return fail(rc);
        if (!progress) std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
    size_t nsym = 0;
    rc = d->clock.finish(&nsym, s, prof);
    const size_t length = f0.length;
    for (int i = 1; i < d->pf_count; ++i) d->pf[i - 1] = d->pf[i];
    --d->pf_count;
    if (rc != XRIT_OK) return fail(rc);
    if (prof) d->prof.collect();
    d->stage_n[4] = nsym;
    d->stats.samples_in = n;
    d->stats.circuit_samples = length;
    d->stats.symbols_out = nsym;
    d->stats.costas_passes = d->costas_seen.passes;
    d->stats.clock_passes = d->clock.passes;
    d->stats.costas_unconverged = d->costas_seen.unconverged;
    d->stats.clock_unconverged = d->clock.unconverged;
    d->stats.costas_max_residual = d->costas_seen.max_residual;
    d->stats.clock_max_residual = d->clock.max_residual;
    d->stats.agc_serial_fallback = d->agc_fallback_seen;
    d->stats.clock_open_large = d->clock.large_open;
    d->stats.costas_serial_walk = d->costas_seen.walked ? 1 : 0;
    d->stats.clock_relay_passes = d->clock.relay_passes;
    d->stats.clock_relay_closed = d->clock.relay_closed ? 1 : 0;
    d->stats.clock_relay_segments = d->clock.relay_segments;
    *n_out = nsym;
    if (d->cfg.strict && d->costas_seen.unconverged) {
        set_error("Costas hand-off did not close: %u boundaries above tolerance", d->costas_seen.unconverged);
        return XRIT_E_NOT_CONVERGED;
    }
    return XRIT_OK;
}

int xrit_demod_process_device(xrit_demod *d, const void *d_samples, size_t n, int type, float *d_soft, size_t cap,
                              size_t *n_out, void *stream)
{
    if (!d || !n_out || (n && !d_samples)) { set_error("null argument"); return XRIT_E_INVALID; }
    if (type < 0 || type > 3) { set_error("unknown sample type %d", type); return XRIT_E_INVALID; }
    *n_out = 0;
    XR_HIP(hipSetDevice(d->device));
    hipStream_t s = stream ? (hipStream_t)stream : d->stream;
    Profiler *prof = d->prof.enabled ? &d->prof : nullptr;
    const unsigned D = d->cfg.decimation;
    if (d->poisoned) {
        set_error("this handle's carried state is inconsistent after an earlier failed call: destroy it and create a new one");
        return XRIT_E_INVALID;
    }
    // the symbols this call can produce at most (slowest admissible symbol clock, plus the unread tail of the last
    // call): checked before any stage advances its state
    {
        const size_t length = D > 1 ? n / D : n;
        const double min_omega = (double)d->sps * (1.0 - (double)d->cfg.clock_omega_limit);
        const size_t worst = (size_t)(((double)length + (double)d->clock.carry) / min_omega) + 2;
        if (worst > cap && length > 0) {
            set_error("output capacity %zu is below the %zu symbols this call may produce (n / (decimation * sps * (1 - omega limit)) + 2)", cap, worst);
            *n_out = worst;        // the capacity that suffices; nothing has been consumed, the handle is unchanged: a front
            return XRIT_E_CAPACITY;    // end that ran ahead stays queued for the retry with the same (pointer, n, type)
        }
    }
    // (round 5: bursts of a million symbols and more in the default configuration)
    if (ov_call(d, n) && !(d->pf_count > 0 && d->pf[0].costas_begun && d->pf[0].ov_job < 0))
        return process_overlap(d, d_samples, n, type, d_soft, cap, n_out, s, prof);
    size_t total_sym = 0, total_len = 0;
    d->agc_fallback_seen = false;
    SliceIO io;
    int rc = XRIT_OK;
    bool costas_ahead = false;
    if (d->pf_count > 0) {
        // the front end of this call ran ahead (xrit_demod_prefetch_device): the loops wait for it, nothing else
        if (d->pf[0].samples != d_samples || d->pf[0].n != n || d->pf[0].type != type) {
            set_error("process calls must take the prefetched inputs in the order they were prefetched");
            d->poisoned = true;
            return XRIT_E_INVALID;
        }
        // (registered only: the call before had no relay phase to put it in front of)
        if (!d->pf[0].launched) XR_TRY(launch_prefetched(d, d->pf[0], nullptr));
        const xrit_demod::Prefetched f = d->pf[0];
        for (int i = 1; i < d->pf_count; ++i) d->pf[i - 1] = d->pf[i];
        --d->pf_count;
        io.length = f.length; io.rrc = f.rrc; io.stat_ready = f.stat_ready; io.set = f.set; io.agc_flag = f.agc_flag; io.exact = f.exact;
        if (f.costas_begun) {
            // its Costas loop ran ahead as well (under the relay of the call before): the host looks at its stop test now
            // (continuing the passes on stream2 in the rare case the batch did not close), the clock recovery follows on s.
            // The host has WAITED for everything stream2 was given for this burst -- front end, Costas loop, count curve; a
            // continued loop ends with a stream synchronise of its own -- so s needs no event to wait for (every runtime
            // call here sits between the end of one burst's relay and the start of the next one's, with the device idle:
            // measured, steady state: 8 us between two calls, 6 us from entry to ClockStage::begin, 30 us to enqueue
            // tail + guess + relay kernels, of which the device spends 20 in the first two).
            // (from here on the registered input has been consumed: a failure poisons the handle below, no early return)
            if (hipEventSynchronize(d->ev_costas) != hipSuccess) { set_error("hipEventSynchronize failed"); rc = XRIT_E_HIP; }
            if (io.length && d->agc.requested_flag() == 2.0f) d->agc_fallback_seen = true;
            bool redone = false;
            if (rc == XRIT_OK) rc = d->costas.finish(d->stream2, prof, &redone);
            costas_note(d);
            if (redone) d->clock.om_scanned = false;
            if (rc == XRIT_OK) rc = loops(d, io, d_soft, cap, &total_sym, s, prof, true, f.slot);
            costas_ahead = true;
        } else {
            if (hipStreamWaitEvent(s, d->ev_fe[f.set], 0) != hipSuccess) { set_error("hipStreamWaitEvent failed"); rc = XRIT_E_HIP; }
        }
    } else {
        const int set = d->next_set;
        d->next_set ^= 1;
        // (a front end that ran ahead on the other stream earlier must be done before this one touches the stages)
        if (d->last_fe_set >= 0) XR_HIP(hipStreamWaitEvent(s, d->ev_fe[d->last_fe_set], 0));
        rc = front_end(d, d_samples, n, type, set, s, prof, &io);
        if (rc == XRIT_OK) { XR_HIP(hipEventRecord(d->ev_fe[set], s)); d->last_fe_set = set; }
    }
    if (rc == XRIT_OK && !costas_ahead) rc = loops(d, io, d_soft, cap, &total_sym, s, prof);
    if (rc != XRIT_OK) {
        // some stage has flipped its ping-pong state, a later one has not: the handle cannot go on
        d->poisoned = true;
        *n_out = 0;
        return rc;
    }
    total_len = io.length;
    const int worst_cp = d->costas_seen.passes, worst_kp = d->clock.passes;
    const unsigned unc_c = d->costas_seen.unconverged, unc_k = d->clock.unconverged, large_k = d->clock.large_open;
    const float res_c = d->costas_seen.max_residual, res_k = d->clock.max_residual;
    if (prof) d->prof.collect();
    d->stage_n[4] = total_sym;
    d->stats.samples_in = n;
    d->stats.circuit_samples = total_len;
    d->stats.symbols_out = total_sym;
    d->stats.costas_passes = worst_cp;
    d->stats.clock_passes = worst_kp;
    d->stats.costas_unconverged = unc_c;
    d->stats.clock_unconverged = unc_k;
    d->stats.costas_max_residual = res_c;
    d->stats.clock_max_residual = res_k;
    d->stats.agc_serial_fallback = d->agc_fallback_seen;
    d->stats.clock_open_large = large_k;
    d->stats.costas_serial_walk = d->costas_seen.walked ? 1 : 0;
    d->stats.clock_relay_passes = d->clock.relay_passes;
    d->stats.clock_relay_closed = d->clock.relay_closed ? 1 : 0;
    d->stats.clock_relay_segments = d->clock.job.relay ? d->clock.relay_segments : 0;
    *n_out = total_sym;
    if (d->cfg.strict && unc_c) {
        set_error("Costas hand-off did not close: %u boundaries above tolerance", unc_c);
        return XRIT_E_NOT_CONVERGED;
    }
    if (d->cfg.strict && large_k) {
        set_error("clock hand-off ended with %u boundaries beyond 0.02 sample or with an open symbol slip", large_k);
        return XRIT_E_NOT_CONVERGED;
    }
    return XRIT_OK;
}

int xrit_demod_prepare_flipped(xrit_demod *d, void *stream)
{
    if (!d) { set_error("null argument"); return XRIT_E_INVALID; }
    if (d->poisoned) { set_error("this handle's carried state is inconsistent after an earlier failed call"); return XRIT_E_INVALID; }
    // (a registered input's front end / Costas loop may have started from the unflipped state, and left the clock stage its
    // timing statistic: the flip belongs between two plain calls)
    if (d->pf_count > 0) { set_error("a prefetched input waits for its process call: the flipped re-run belongs between plain calls"); return XRIT_E_INVALID; }
    XR_HIP(hipSetDevice(d->device));
    hipStream_t s = stream ? (hipStream_t)stream : d->stream;
    int rc = d->clock.make_alt(s, d->prof.enabled ? &d->prof : nullptr);
    if (rc != XRIT_OK) d->poisoned = true;
    return rc;
}

size_t xrit_demod_clock_carry_bytes(void) { return ClockStage::CARRY_BYTES; }

int xrit_demod_export_clock_carry(xrit_demod *d, int which, void *d_rec, void *stream)
{
    if (!d || !d_rec || which < 0 || which > 1) { set_error("null or invalid argument"); return XRIT_E_INVALID; }
    if (d->poisoned) { set_error("this handle's carried state is inconsistent after an earlier failed call"); return XRIT_E_INVALID; }
    if (d->pf_count > 0) { set_error("a prefetched input waits for its process call: the carried state is not between two calls"); return XRIT_E_INVALID; }
    XR_HIP(hipSetDevice(d->device));
    return d->clock.export_carry(d_rec, which, stream ? (hipStream_t)stream : d->stream);
}

int xrit_demod_last_clock_exact(const xrit_demod *d)
{
    if (!d) return XRIT_E_INVALID;
    return d->clock.last_walk_exact() ? 1 : 0;
}

int xrit_demod_redo_clock_from(xrit_demod *d, const void *d_rec, float *d_soft, size_t cap, size_t *n_out, void *stream)
{
    if (!d || !d_rec || !n_out || !d_soft) { set_error("null argument"); return XRIT_E_INVALID; }
    *n_out = 0;
    if (d->poisoned) { set_error("this handle's carried state is inconsistent after an earlier failed call"); return XRIT_E_INVALID; }
    if (d->pf_count > 0) { set_error("a prefetched input waits for its process call: the re-run belongs between plain calls"); return XRIT_E_INVALID; }
    XR_HIP(hipSetDevice(d->device));
    hipStream_t s = stream ? (hipStream_t)stream : d->stream;
    unsigned head[4] = {0, 0, 0, 0};
    XR_HIP(hipMemcpyAsync(head, d_rec, sizeof head, hipMemcpyDeviceToHost, s));
    XR_HIP(hipStreamSynchronize(s));
    if (head[0] != 1u || head[1] > 1024u) { set_error("not a carried clock state (xrit_demod_export_clock_carry)"); return XRIT_E_INVALID; }
    Profiler *prof = d->prof.enabled ? &d->prof : nullptr;
    float2 *sym = (d->keep_stages || d->keep_symbols) ? d->stage_buf[4].as<float2>() : nullptr;
    size_t k = 0;
    const int rc = d->clock.redo_from(d_rec, head[1], d_soft, sym, cap, &k, s, prof);
    if (rc != XRIT_OK) { d->poisoned = true; return rc; }
    d->stage_n[4] = k;
    d->stats.symbols_out = k;
    d->stats.clock_passes = d->clock.passes;
    d->stats.clock_relay_passes = d->clock.relay_passes;
    d->stats.clock_relay_closed = d->clock.relay_closed ? 1 : 0;
    *n_out = k;
    return XRIT_OK;
}

int xrit_demod_flip_costas_phase(xrit_demod *d, void *stream)
{
    if (!d) { set_error("null argument"); return XRIT_E_INVALID; }
    if (d->poisoned) { set_error("this handle's carried state is inconsistent after an earlier failed call"); return XRIT_E_INVALID; }
    if (d->pf_count > 0) { set_error("a prefetched input waits for its process call: its Costas loop may already have started"); return XRIT_E_INVALID; }
    XR_HIP(hipSetDevice(d->device));
    return d->costas.flip_phase(stream ? (hipStream_t)stream : d->stream);
}

int xrit_demod_redo_clock_flipped(xrit_demod *d, float *d_soft, size_t cap, size_t *n_out, void *stream)
{
    if (!d || !n_out || !d_soft) { set_error("null argument"); return XRIT_E_INVALID; }
    *n_out = 0;
    if (d->poisoned) { set_error("this handle's carried state is inconsistent after an earlier failed call"); return XRIT_E_INVALID; }
    // (a registered input's front end / Costas loop may have started from the unflipped state, and left the clock stage its
    // timing statistic: the flip belongs between two plain calls)
    if (d->pf_count > 0) { set_error("a prefetched input waits for its process call: the flipped re-run belongs between plain calls"); return XRIT_E_INVALID; }
    XR_HIP(hipSetDevice(d->device));
    hipStream_t s = stream ? (hipStream_t)stream : d->stream;
    Profiler *prof = d->prof.enabled ? &d->prof : nullptr;
    float2 *sym = (d->keep_stages || d->keep_symbols) ? d->stage_buf[4].as<float2>() : nullptr;
    size_t k = 0;
    // The negated Costas output is the other lock's output to rounding only, which alone moves a float32 M&M by its
    // 5e-5 .. 1e-4.  The run is repeated in the handle's own configuration (cfg.clock_exact: by
    // default two hand-off passes and three relay passes, csrc/clock_relay.h).
    int rc = d->clock.redo_flipped(d_soft, sym, cap, &k, s, prof);
    if (rc == XRIT_OK) rc = d->costas.flip_phase(s);
    if (rc != XRIT_OK) { d->poisoned = true; return rc; }
    d->stage_n[4] = k;
    d->stats.symbols_out = k;
    d->stats.clock_passes = d->clock.passes;
    d->stats.clock_relay_passes = d->clock.relay_passes;
    d->stats.clock_relay_closed = d->clock.relay_closed ? 1 : 0;
    *n_out = k;
    return XRIT_OK;
}

int xrit_demod_prefetch_depth(xrit_demod *d, size_t n)
{
    if (!d) return 0;
    if (d->keep_stages || d->keep_symbols || (d->prof.enabled && !d->prof.light)) return 0;
    return ov_call(d, n) ? XRIT_AHEAD : 1;
}

int xrit_demod_prefetch_device(xrit_demod *d, const void *d_samples, size_t n, int type, void *stream)
{
    if (!d || (n && !d_samples)) { set_error("null argument"); return XRIT_E_INVALID; }
    if (type < 0 || type > 3) { set_error("unknown sample type %d", type); return XRIT_E_INVALID; }
    if (d->poisoned) { set_error("this handle's carried state is inconsistent after an earlier failed call"); return XRIT_E_INVALID; }
    // Two sets of front-end buffers: the front end of the input two behind a call would overwrite what that call's loops still
    // read.  Bursts whose clock recovery walks overlapping blocks hold their front ends back until that is safe (ov_service) and
    // may wait three deep -- the call in progress and two behind it --; everything else two deep, as before round 5.
    {
        bool all_ov = ov_call(d, n);
        for (int i = 0; i < d->pf_count && all_ov; ++i) all_ov = ov_call(d, d->pf[i].n);
        const int room = all_ov ? XRIT_AHEAD + 1 : 2;
        if (d->pf_count >= room) { set_error("%d prefetched inputs are already waiting for their process calls", d->pf_count); return XRIT_E_INVALID; }
    }
    // stage copies and per-kernel event brackets belong to one call at a time: no running ahead then
    if (d->keep_stages || d->keep_symbols || (d->prof.enabled && !d->prof.light)) return XRIT_OK;
    XR_HIP(hipSetDevice(d->device));
    hipStream_t s = stream ? (hipStream_t)stream : d->stream;
    // stream2 may read the input once whatever the caller queued on its stream is done, and may touch the stages once
    // the previous front end (on either stream) is done
    XR_HIP(hipEventRecord(d->ev_ready, s));
    xrit_demod::Prefetched &f = d->pf[d->pf_count];
    f = xrit_demod::Prefetched{};
    f.samples = d_samples; f.n = n; f.type = type; f.launched = false;
    // Unless the relay is off (cfg.clock_exact < 0) the next process call ends in relay kernels that occupy three
    // waves per CU: the front end waits for those (ClockStage::before_relay) instead of competing with the Costas and
    // hand-off passes.  (Registered inputs start in order: the process call that takes one starts it if it still waits,
    // and starts the one behind it in front of its own relay kernels.)
    // (bursts whose clock recovery walks overlapping blocks: the next process call enqueues everything that can run ahead)
    const bool defer = (d->clock.relay_by_default() && !d->no_defer) || ov_call(d, n);
    if (!defer) {
        // (front ends run in the order of their inputs: one that is still waiting goes first)
        if (d->pf_count > 0 && !d->pf[0].launched) XR_TRY(launch_prefetched(d, d->pf[0], nullptr));
        XR_TRY(launch_prefetched(d, f, nullptr));
    }
    ++d->pf_count;
    return XRIT_OK;
}

int xrit_demod_process(xrit_demod *d, const void *samples, size_t n, int type, float *soft_out, size_t cap,
                       size_t *n_out)
{
    if (!d || !n_out || (n && !samples) || (cap && !soft_out)) { set_error("null argument"); return XRIT_E_INVALID; }
    if (type < 0 || type > 3) { set_error("unknown sample type %d", type); return XRIT_E_INVALID; }
    XR_HIP(hipSetDevice(d->device));
    const size_t esz = type == XRIT_SAMPLE_FLOATIQ ? 8 : (type == XRIT_SAMPLE_S16IQ ? 4 : 2);
    XR_TRY(d->in_dev.reserve(n * esz + 16));
    XR_TRY(d->soft_dev.reserve((cap + 1) * sizeof(float)));
    if (n) XR_HIP(hipMemcpyAsync(d->in_dev.p, samples, n * esz, hipMemcpyHostToDevice, d->stream));
    int rc = xrit_demod_process_device(d, d->in_dev.p, n, type, d->soft_dev.as<float>(), cap, n_out, d->stream);
    if (rc != XRIT_OK) return rc;
    if (*n_out)
        XR_HIP(hipMemcpyAsync(soft_out, d->soft_dev.p, *n_out * sizeof(float), hipMemcpyDeviceToHost, d->stream));
    XR_HIP(hipStreamSynchronize(d->stream));
    return XRIT_OK;
}

int xrit_demod_read_stage(xrit_demod *d, int stage, float *out, size_t cap, size_t *n)
{
    if (!d || stage < 0 || stage > 4 || !n) { set_error("bad argument"); return XRIT_E_INVALID; }
    if (!d->keep_stages && !(d->keep_symbols && stage == 4)) {
        set_error("stage copies are off: call xrit_demod_keep_stages(d, 1) first (2 keeps stage 4 only)");
        return XRIT_E_INVALID;
    }
    *n = d->stage_n[stage];
    if (!out) return XRIT_OK;
    if (*n > cap) { set_error("stage %d holds %zu elements, capacity %zu", stage, *n, cap); return XRIT_E_CAPACITY; }
    XR_HIP(hipSetDevice(d->device));
    XR_HIP(hipStreamSynchronize(d->stream));
    if (*n) XR_HIP(hipMemcpy(out, d->stage_buf[stage].p, *n * sizeof(float2), hipMemcpyDeviceToHost));
    return XRIT_OK;
}

int xrit_demod_keep_stages(xrit_demod *d, int enable)
{
    if (!d) return XRIT_E_INVALID;
    // (what runs ahead of a registered input's call is chosen by these flags: a call whose Costas loop and walkers have already
    // been started as an overlap job must not be taken by the stage-keeping path afterwards)
    if (d->pf_count > 0) { set_error("prefetched inputs wait for their process calls: stage copies are switched between plain calls"); return XRIT_E_INVALID; }
    d->keep_stages = enable == 1;
    d->keep_symbols = enable == 2;
    return XRIT_OK;
}

int xrit_demod_get_stats(const xrit_demod *d, xrit_demod_stats *s)
{
    if (!d || !s) return XRIT_E_INVALID;
    *s = d->stats;
    return XRIT_OK;
}

int xrit_demod_profile(xrit_demod *d, int enable)
{
    if (!d) return XRIT_E_INVALID;
    d->prof.enabled = enable != 0;
    d->prof.light = enable == 2;
    d->prof.reset();
    return XRIT_OK;
}

int xrit_demod_profile_read(xrit_demod *d, const char **names, float *total_ms, int *launches, int cap)
{
    if (!d) return XRIT_E_INVALID;
    d->prof_names = d->prof.order;
    int n = 0;
    for (auto &nm : d->prof_names) {
        if (n >= cap) break;
        auto &v = d->prof.acc[nm];
        if (names) names[n] = nm.c_str();
        if (total_ms) total_ms[n] = (float)v.first;
        if (launches) launches[n] = v.second;
        ++n;
    }
    return n;
}

int xrit_demod_profile_samples(xrit_demod *d, const char *name, float *ms, int cap)
{
    if (!d || !name) return XRIT_E_INVALID;
    auto it = d->prof.samples.find(name);
    if (it == d->prof.samples.end()) return 0;
    int n = 0;
    for (float v : it->second) {
        if (n >= cap) break;
        if (ms) ms[n] = v;
        ++n;
    }
    return n;
}

int xrit_quantize_i8_device(const float *d_soft, int8_t *d_out, size_t n, int device, void *stream)
{
    XR_TRY(select_device(device));
    return launch_quantize_i8(d_soft, d_out, n, (hipStream_t)stream);
}

int xrit_sync_correlate_device(const int8_t *d_symbols, size_t n, const uint64_t *words, int nwords, uint32_t frame,
                               xrit_sync_hit *d_hits, int device, void *stream)
{
    if (!words || (n >= frame && (!d_symbols || !d_hits))) { set_error("null argument"); return XRIT_E_INVALID; }
    XR_TRY(select_device(device));
    return launch_sync_correlate(d_symbols, n, reinterpret_cast<const unsigned long long *>(words), nwords, frame, d_hits,
                                 (hipStream_t)stream);
}

int xrit_sync_correlate(const int8_t *symbols, size_t n, const uint64_t *words, int nwords, uint32_t frame,
                        xrit_sync_hit *hits, int device)
{
    if (!words || frame == 0 || (n >= frame && (!symbols || !hits))) { set_error("null argument"); return XRIT_E_INVALID; }
    XR_TRY(select_device(device));
    const size_t nf = n / frame;
    if (nf == 0) return XRIT_OK;
    DevBuf din, dh;
    int rc = din.reserve(nf * frame);
    if (rc == XRIT_OK) rc = dh.reserve(nf * sizeof(xrit_sync_hit));
    if (rc == XRIT_OK && hipMemcpy(din.p, symbols, nf * frame, hipMemcpyHostToDevice) != hipSuccess) { set_error("hipMemcpy failed"); rc = XRIT_E_HIP; }
    if (rc == XRIT_OK)
        rc = launch_sync_correlate(din.as<int8_t>(), nf * frame, reinterpret_cast<const unsigned long long *>(words), nwords, frame,
                                   dh.as<xrit_sync_hit>(), nullptr);
    if (rc == XRIT_OK && (hipDeviceSynchronize() != hipSuccess ||
                          hipMemcpy(hits, dh.p, nf * sizeof(xrit_sync_hit), hipMemcpyDeviceToHost) != hipSuccess)) {
        set_error("sync: device error");
        rc = XRIT_E_HIP;
    }
    din.release();
    dh.release();
    return rc;
}

int xrit_sync_fix_frames_device(const int8_t *d_symbols, size_t n, const xrit_sync_hit *d_hits, uint32_t frame,
                                uint32_t min_correlation, int8_t *d_frames, uint8_t *d_valid, int device, void *stream)
{
    if (frame == 0 || (n >= frame && (!d_symbols || !d_hits || !d_frames || !d_valid))) { set_error("null argument"); return XRIT_E_INVALID; }
    XR_TRY(select_device(device));
    return launch_sync_fix(d_symbols, n, d_hits, frame, min_correlation, d_frames, d_valid, (hipStream_t)stream);
}

int xrit_sync_fix_frames(const int8_t *symbols, size_t n, const xrit_sync_hit *hits, uint32_t frame,
                         uint32_t min_correlation, int8_t *frames, uint8_t *valid, int device)
{
    if (frame == 0 || (n >= frame && (!symbols || !hits || !frames || !valid))) { set_error("null argument"); return XRIT_E_INVALID; }
    XR_TRY(select_device(device));
    const size_t nf = n / frame;
    if (nf == 0) return XRIT_OK;
    DevBuf din, dh, dout, dv;
    int rc = din.reserve(n + 8);
    if (rc == XRIT_OK) rc = dh.reserve(nf * sizeof(xrit_sync_hit));
    if (rc == XRIT_OK) rc = dout.reserve(nf * frame);
    if (rc == XRIT_OK) rc = dv.reserve(nf);
    if (rc == XRIT_OK && (hipMemcpy(din.p, symbols, n, hipMemcpyHostToDevice) != hipSuccess ||
                          hipMemcpy(dh.p, hits, nf * sizeof(xrit_sync_hit), hipMemcpyHostToDevice) != hipSuccess)) {
        set_error("hipMemcpy failed");
        rc = XRIT_E_HIP;
    }
    if (rc == XRIT_OK)
        rc = launch_sync_fix(din.as<int8_t>(), n, dh.as<xrit_sync_hit>(), frame, min_correlation, dout.as<int8_t>(),
                             dv.as<unsigned char>(), nullptr);
    if (rc == XRIT_OK && (hipDeviceSynchronize() != hipSuccess ||
                          hipMemcpy(frames, dout.p, nf * frame, hipMemcpyDeviceToHost) != hipSuccess ||
                          hipMemcpy(valid, dv.p, nf, hipMemcpyDeviceToHost) != hipSuccess)) {
        set_error("sync: device error");
        rc = XRIT_E_HIP;
    }
    din.release(); dh.release(); dout.release(); dv.release();
    return rc;
}

int xrit_quantize_i8(xrit_demod *d, const float *soft, int8_t *out, size_t n)
{
    if (!d || (n && (!soft || !out))) { set_error("null argument"); return XRIT_E_INVALID; }
    XR_HIP(hipSetDevice(d->device));
    XR_TRY(d->q_in.reserve(n * sizeof(float) + 16));
    XR_TRY(d->q_out.reserve(n + 16));
    if (!n) return XRIT_OK;
    XR_HIP(hipMemcpyAsync(d->q_in.p, soft, n * sizeof(float), hipMemcpyHostToDevice, d->stream));
    XR_TRY(launch_quantize_i8(d->q_in.as<float>(), d->q_out.as<int8_t>(), n, d->stream));
    XR_HIP(hipMemcpyAsync(out, d->q_out.p, n, hipMemcpyDeviceToHost, d->stream));
    XR_HIP(hipStreamSynchronize(d->stream));
    return XRIT_OK;
}

// ---------------------------------------------------------------- stage objects
int xrit_fir_create(unsigned decimation, const float *taps, int ntaps, int device, xrit_fir **out)
{
    if (!taps || ntaps < 1 || !out) { set_error("bad argument"); return XRIT_E_INVALID; }
    xrit_fir *f = new (std::nothrow) xrit_fir();
    if (!f) return XRIT_E_NOMEM;
    f->stream = nullptr;
    int rc = stage_open(f, device);
    if (rc == XRIT_OK) rc = f->st.init(taps, ntaps, (int)decimation);
    if (rc != XRIT_OK) { stage_close(f); return rc; }
    *out = f;
    return XRIT_OK;
}

int xrit_fir_work(xrit_fir *f, const float *in, float *out, size_t n_out)
{
    if (!f || (n_out && (!in || !out))) { set_error("null argument"); return XRIT_E_INVALID; }
    XR_HIP(hipSetDevice(f->device));
    size_t n_in = n_out * (size_t)f->st.D;
    XR_TRY(f->in.reserve((n_in + 8) * sizeof(float2)));
    XR_TRY(f->out.reserve((n_out + 8) * sizeof(float2)));
    if (n_in) XR_HIP(hipMemcpyAsync(f->in.p, in, n_in * sizeof(float2), hipMemcpyHostToDevice, f->stream));
    XR_TRY(f->st.run(f->in.p, XRIT_SAMPLE_FLOATIQ, f->out.as<float2>(), n_out, f->stream, nullptr));
    if (n_out) XR_HIP(hipMemcpyAsync(out, f->out.p, n_out * sizeof(float2), hipMemcpyDeviceToHost, f->stream));
    XR_HIP(hipStreamSynchronize(f->stream));
    return XRIT_OK;
}

int xrit_fir_set_exact(xrit_fir *f, int exact)
{
    if (!f) { set_error("null argument"); return XRIT_E_INVALID; }
    XR_HIP(hipSetDevice(f->device));
    XR_HIP(hipStreamSynchronize(f->stream));
    return f->st.set_exact(exact != 0);
}

void xrit_fir_destroy(xrit_fir *f) { stage_close(f); }

int xrit_agc_create(float rate, float reference, float gain, float max_gain, int device, xrit_agc **out)
{
    if (!out) return XRIT_E_INVALID;
    xrit_agc *a = new (std::nothrow) xrit_agc();
    if (!a) return XRIT_E_NOMEM;
    a->stream = nullptr;
    int rc = stage_open(a, device);
    if (rc == XRIT_OK) rc = a->st.init(rate, reference, gain, max_gain);
    if (rc != XRIT_OK) { stage_close(a); return rc; }
    *out = a;
    return XRIT_OK;
}

int xrit_agc_work(xrit_agc *a, const float *in, float *out, size_t n)
{
    if (!a || (n && (!in || !out))) { set_error("null argument"); return XRIT_E_INVALID; }
    XR_HIP(hipSetDevice(a->device));
    XR_TRY(a->in.reserve((n + 8) * sizeof(float2)));
    XR_TRY(a->out.reserve((n + 8) * sizeof(float2)));
    if (n) XR_HIP(hipMemcpyAsync(a->in.p, in, n * sizeof(float2), hipMemcpyHostToDevice, a->stream));
    XR_TRY(a->st.run(a->in.as<float2>(), a->out.as<float2>(), n, a->stream, nullptr));
    if (n) XR_HIP(hipMemcpyAsync(out, a->out.p, n * sizeof(float2), hipMemcpyDeviceToHost, a->stream));
    XR_HIP(hipStreamSynchronize(a->stream));
    return XRIT_OK;
}

int xrit_agc_set_exact(xrit_agc *a, int exact)
{
    if (!a) { set_error("null argument"); return XRIT_E_INVALID; }
    a->st.exact = exact != 0;
    return XRIT_OK;
}

int xrit_agc_exact_stats(xrit_agc *a, uint32_t *counters8)
{
    if (!a || !counters8) { set_error("null argument"); return XRIT_E_INVALID; }
    XR_HIP(hipSetDevice(a->device));
    return a->st.exact_counters(counters8, a->stream);
}

float xrit_agc_gain(xrit_agc *a)
{
    float g = NAN;
    if (!a) return g;
    (void)hipSetDevice(a->device);
    (void)a->st.gain(&g, a->stream);
    return g;
}

void xrit_agc_destroy(xrit_agc *a) { stage_close(a); }

int xrit_costas_create(float loop_bw, int order, int device, xrit_costas **out)
{
    if (!out) return XRIT_E_INVALID;
    if (order != 2) { set_error("only the 2nd-order (BPSK) loop is implemented, as the reference uses (LOOP_ORDER 2)"); return XRIT_E_INVALID; }
    xrit_costas *c = new (std::nothrow) xrit_costas();
    if (!c) return XRIT_E_NOMEM;
    c->stream = nullptr;
    int rc = stage_open(c, device);
    if (rc == XRIT_OK) rc = c->st.init(loop_bw, 0, 0);
    if (rc != XRIT_OK) { stage_close(c); return rc; }
    *out = c;
    return XRIT_OK;
}

int xrit_costas_work(xrit_costas *c, const float *in, float *out, size_t n)
{
    if (!c || (n && (!in || !out))) { set_error("null argument"); return XRIT_E_INVALID; }
    XR_HIP(hipSetDevice(c->device));
    XR_TRY(c->in.reserve((n + 8) * sizeof(float2)));
    XR_TRY(c->out.reserve((n + 8) * sizeof(float2)));
    if (n) XR_HIP(hipMemcpyAsync(c->in.p, in, n * sizeof(float2), hipMemcpyHostToDevice, c->stream));
    XR_TRY(c->st.run(c->in.as<float2>(), c->out.as<float2>(), n, c->stream, nullptr));
    if (n) XR_HIP(hipMemcpyAsync(out, c->out.p, n * sizeof(float2), hipMemcpyDeviceToHost, c->stream));
    XR_HIP(hipStreamSynchronize(c->stream));
    return XRIT_OK;
}

int xrit_loop_sincosf(const float *x, float *sin_out, float *cos_out, size_t n, int device)
{
    if (n && (!x || !sin_out || !cos_out)) { set_error("null argument"); return XRIT_E_INVALID; }
    XR_TRY(select_device(device));
    if (n == 0) return XRIT_OK;
    DevBuf dx, ds, dc;
    int rc = dx.reserve(n * sizeof(float));
    if (rc == XRIT_OK) rc = ds.reserve(n * sizeof(float));
    if (rc == XRIT_OK) rc = dc.reserve(n * sizeof(float));
    if (rc == XRIT_OK && hipMemcpy(dx.p, x, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { set_error("hipMemcpy failed"); rc = XRIT_E_HIP; }
    if (rc == XRIT_OK) rc = launch_loop_sincosf(dx.as<float>(), ds.as<float>(), dc.as<float>(), n, nullptr);
    if (rc == XRIT_OK && (hipDeviceSynchronize() != hipSuccess ||
                          hipMemcpy(sin_out, ds.p, n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess ||
                          hipMemcpy(cos_out, dc.p, n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)) {
        set_error("loop_sincosf: device error");
        rc = XRIT_E_HIP;
    }
    dx.release(); ds.release(); dc.release();
    return rc;
}

int xrit_costas_set_exact(xrit_costas *c, int exact, int history)
{
    if (!c) { set_error("null argument"); return XRIT_E_INVALID; }
    if (history < 0) { set_error("history = %d", history); return XRIT_E_INVALID; }
    c->st.exact = exact != 0;
    c->st.ex_hist = history > 0 ? history : -1;
    return XRIT_OK;
}

int xrit_costas_exact_stats(xrit_costas *c, uint64_t *blocks, uint64_t *rounds, uint32_t *joints_open, uint32_t *fix_rounds,
                            uint64_t *lattice_segments, uint64_t *lattice_fallbacks)
{
    if (!c) { set_error("null argument"); return XRIT_E_INVALID; }
    if (blocks) *blocks = c->st.ex_blocks;
    if (rounds) *rounds = c->st.ex_picard;
    if (joints_open) *joints_open = c->st.ex_open;
    if (fix_rounds) *fix_rounds = (uint32_t)c->st.ex_rounds;
    if (lattice_segments) *lattice_segments = c->st.ex_segs;
    if (lattice_fallbacks) *lattice_fallbacks = c->st.ex_fallbacks;
    return XRIT_OK;
}

int xrit_costas_state(xrit_costas *c, float *phase, float *freq)
{
    if (!c || !phase || !freq) return XRIT_E_INVALID;
    XR_HIP(hipSetDevice(c->device));
    return c->st.get_state(phase, freq, c->stream);
}

void xrit_costas_destroy(xrit_costas *c) { stage_close(c); }

int xrit_clock_create(float omega, float gain_omega, float mu, float gain_mu, float omega_rel_limit, int device,
                      xrit_clock **out)
{
    if (!out || !(omega >= 1.0f)) { set_error("bad argument"); return XRIT_E_INVALID; }
    xrit_clock *c = new (std::nothrow) xrit_clock();
    if (!c) return XRIT_E_NOMEM;
    c->stream = nullptr;
    int rc = stage_open(c, device);
    if (rc == XRIT_OK) rc = c->st.init(omega, gain_omega, mu, gain_mu, omega_rel_limit, 0, 0);
    if (rc != XRIT_OK) { stage_close(c); return rc; }
    c->st.ov_allow = false;         // (the stage object hands out complex symbols: ClockRecovery::Work's output)
    *out = c;
    return XRIT_OK;
}

int xrit_clock_work(xrit_clock *c, const float *in, size_t n, float *out, size_t cap, size_t *n_out)
{
    if (!c || !n_out || (n && !in)) { set_error("null argument"); return XRIT_E_INVALID; }
    XR_HIP(hipSetDevice(c->device));
    float2 *slot = nullptr;
    XR_TRY(c->st.input_slot(n, &slot, c->stream));
    if (n) XR_HIP(hipMemcpyAsync(slot, in, n * sizeof(float2), hipMemcpyHostToDevice, c->stream));
    XR_TRY(c->out.reserve((cap + 8) * sizeof(float2)));
    XR_TRY(c->st.run(n, nullptr, c->out.as<float2>(), cap, n_out, c->stream, nullptr));
    if (*n_out && out)
        XR_HIP(hipMemcpyAsync(out, c->out.p, *n_out * sizeof(float2), hipMemcpyDeviceToHost, c->stream));
    XR_HIP(hipStreamSynchronize(c->stream));
    return XRIT_OK;
}

int xrit_clock_set_serial(xrit_clock *c, int serial)
{
    if (!c) return XRIT_E_INVALID;
    c->st.serial = serial != 0;
    return XRIT_OK;
}

int xrit_clock_set_exact(xrit_clock *c, int exact, int window)
{
    if (!c) return XRIT_E_INVALID;
    if (exact < -3) { set_error("clock_exact = %d: -3 .. n (see xrit_demod_config.clock_exact)", exact); return XRIT_E_INVALID; }
    c->st.exact = exact == -3 ? 0 : exact;          // (-3: the default's plan, its first relay passes approximate -- as in xrit_demod_create)
    c->st.relay_quick = exact == -3;
    c->st.relay_window = window > 0 ? window : 0;
    return XRIT_OK;
}

void xrit_clock_destroy(xrit_clock *c) { stage_close(c); }

int xrit_rtl_create(float sample_rate, int device, xrit_rtl **out)
{
    if (!out || !(sample_rate > 0)) { set_error("bad argument"); return XRIT_E_INVALID; }
    xrit_rtl *r = new (std::nothrow) xrit_rtl();
    if (!r) return XRIT_E_NOMEM;
    r->stream = nullptr;
    int rc = stage_open(r, device);
    if (rc == XRIT_OK) rc = r->st.init(sample_rate);
    if (rc != XRIT_OK) { stage_close(r); return rc; }
    *out = r;
    return XRIT_OK;
}

int xrit_rtl_work(xrit_rtl *r, const uint8_t *data, size_t n_complex, float *out_iq)
{
    if (!r || (n_complex && (!data || !out_iq))) { set_error("null argument"); return XRIT_E_INVALID; }
    XR_HIP(hipSetDevice(r->device));
    XR_TRY(r->in.reserve(2 * n_complex + 16));
    XR_TRY(r->out.reserve((n_complex + 8) * sizeof(float2)));
    if (n_complex) XR_HIP(hipMemcpyAsync(r->in.p, data, 2 * n_complex, hipMemcpyHostToDevice, r->stream));
    XR_TRY(r->st.run(r->in.p, r->out.as<float2>(), n_complex, r->stream, nullptr));
    if (n_complex) XR_HIP(hipMemcpyAsync(out_iq, r->out.p, n_complex * sizeof(float2), hipMemcpyDeviceToHost, r->stream));
    XR_HIP(hipStreamSynchronize(r->stream));
    return XRIT_OK;
}

void xrit_rtl_destroy(xrit_rtl *r) { stage_close(r); }

// ---------------------------------------------------------------- synth
void xrit_synth_defaults(xrit_synth_params *p)
{
    if (!p) return;
    p->fs_in = 1.25e6;
    p->symbol_rate = 293883.0;
    p->alpha = 0.5;
    p->amplitude = 0.1;
    p->carrier_hz = 500.0;
    p->phase0 = 0.7;
    p->timing_offset = 0.3;
    p->clock_ppm = 20.0;
    p->esn0_db = 12.0;
    p->seed = 0x58524954ull;
}

int xrit_synth_generate_device(const xrit_synth_params *p, uint64_t start, size_t n, float *d_out, int device,
                               void *stream)
{
    if (!p || (n && !d_out)) { set_error("null argument"); return XRIT_E_INVALID; }
    XR_TRY(select_device(device));
    return launch_synth(*p, start, n, reinterpret_cast<float2 *>(d_out), (hipStream_t)stream);
}

int xrit_device_read_bandwidth(const void *d_buf, size_t bytes, int reps, int device, void *stream, double *gb_per_s)
{
    if (!d_buf || !gb_per_s) { set_error("null argument"); return XRIT_E_INVALID; }
    XR_TRY(select_device(device));
    return launch_read_bw(d_buf, bytes, reps, (hipStream_t)stream, gb_per_s);
}

}  // extern "C"
