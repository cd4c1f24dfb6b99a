// kernels.h -- stage objects of the chain (device state + launch plans).
#pragma once

#include <functional>

#include "common.h"
#include "loop_core.h"
#include "taps.h"

namespace xrit {

// ---- FirFilter (demodulator.cpp:446,450; Work at :138,:148) ---------------
struct FirStage {
    int T = 0, D = 1, RC = 3, W = 0, Wpad = 0, threads = 256, tile_len = 0, cur = 0;
    bool pad = false;
    bool mfma_dec = false;      // XRIT_MFMA_DEC=1: the C2 decimator on the matrix pipe (experiment, fir.hip)
    bool no_static_dec = false, no_static_mf = false;   // XRIT_NO_STATIC_DEC / XRIT_NO_STATIC_MF, read once in init (A/B runs)
    int prio = 0;        // 1: this launch's waves at a raised issue priority (fir_decim_kernel; set by the chain per front end)
    bool poly = false;   // polyphase kernel (lanes = phases) for decimation 16 / 32 / 64
    // cfg.front_exact = 2: summed in the CPU chain's order, no FMA (fir_exact_kernel); set before init or through set_exact()
    bool exact = false, ex_pad = false;
    int ex_threads = 256, ex_opt = 2, ex_tile_len = 0;
    size_t ex_lds = 0;
    DevBuf rt;           // the taps reversed (exact kernel)
    std::vector<float> taps_host;
    int set_exact(bool on);
#ifndef XRIT_POLY_PR
#define XRIT_POLY_PR 16
#endif
    static constexpr int POLY_PR = XRIT_POLY_PR, POLY_NQ = 32;     // outputs per lane group, taps per phase (T <= 32 * D)
    size_t lds_bytes = 0;
    DevBuf g;        // RC x Wpad window taps (polyphase: D x POLY_NQ phase taps)
    DevBuf mfb;      // matrix-pipe experiment: the Toeplitz tap operand, one float per (step, lane)
    DevBuf hist[2];  // T-1 samples of history (ping-pong)
    int init(const float *taps, int ntaps, int decim);
    int reset(hipStream_t s);     // zero history, as after construction (no reallocation)
    void release();
    // consumes n_out*D samples of `in` (sample_type as in FrontendDevice.h:11-13)
    // stat (optional): sum z^2 per run of statL outputs, written when stat_supported(statL)
    // agc (optional): the AGC's composed gain map per run of 64 * RC outputs, see AgcStage::fused_begin
    // fill (optional, D == 1): the input is the AGC's INPUT stream; the window fill applies the AGC on the fly
    // (AgcStage::fused_scan fills the descriptor) -- the AGC output is never written
    // use_exact (the chain's per-call choice, round 6): this call through the exact-order twin (a second stage object with
    // `exact` set, created beside every stage that is not exact itself; the T - 1 samples of history follow the call)
    int run(const void *in, int sample_type, float2 *out, size_t n_out, hipStream_t s, Profiler *prof,
            float2 *stat = nullptr, int statL = 0, const struct AgcEpilogue *agc = nullptr,
            const struct AgcFill *fill = nullptr, bool use_exact = false);
    FirStage *twin = nullptr;   // (owned: released in release())
    bool last_twin = false;     // the last call went through the twin: the history lives there
    FirStage() = default;
    FirStage(const FirStage &) = delete;
    FirStage &operator=(const FirStage &) = delete;
    bool agc_fill_supported(int per_lane) const;
    bool stat_supported(int statL) const;
    bool agc_supported() const;
};

// What the decimator's epilogue needs to leave the AGC's first sweep behind (AgcStage::fused_begin fills it).
constexpr int AGC_RUN_MAX_PER_LANE = 5;
// What the matched filter's window fill needs to apply the AGC itself: the stream in front of the AGC, the
// exclusive prefix of the run maps, the call's start gain.  If the guard flag is up the gains are not the
// scan's: the kernel then reads `in`, which agc_serial_kernel has filled (never seen with normalised samples).
struct AgcFill {
    const float2 *x;
    const struct AgcMap *pre_run;
    const float *state_in;      // [0] gain at the start of the call
    float *state_out;           // [0] gain after the call (written by the history kernel), [1] guard flag
    float rate, ref, maxg;
    int per_lane;               // a run = 64 * per_lane samples
};
struct AgcEpilogue {
    struct AgcMap *maps;     // one per run of 64 * RC outputs (the outputs of one wave of the FIR kernel)
    float *state_out;        // [1] guard flag
    float rate, ref, maxg;
};

// ---- RTL-SDR ingest (RtlFrontend.cpp:26-28,57,102-116): u8 IQ -> float IQ with the frontend's DC tracker ----
struct RtlIngestStage {
    float alpha = 0;
    DevBuf state;    // running average, ping-pong across calls
    DevBuf aggs;
    int cur = 0;
    int init(float sample_rate);
    int reset(hipStream_t s);
    void release();
    int run(const void *in_u8, float2 *out, size_t n_complex, hipStream_t s, Profiler *prof);
};

// ---- AGC (demodulator.cpp:447; Work at :143) ------------------------------
struct AgcStage {
    float rate = 0, ref = 0, maxg = 0;
    DevBuf state;    // two (gain, guard flag) slots, ping-pong across calls
    DevBuf aggs;     // per-block composed maps
    int cur = 0;
    int init(float rate, float reference, float gain, float max_gain);
    int reset(hipStream_t s);
    float gain0 = 1.0f;
    void release();
    int run(const float2 *in, float2 *out, size_t n, hipStream_t s, Profiler *prof, bool use_exact = false);
    // cfg.front_exact = 2: chains walked literally from the scan's gains, warmed up until they ARE the serial recurrence (agc.hip)
    bool exact = false;
    DevBuf joints;   // exact_walk.h's joints, block records and counters
    int ex_walkers = 2048, ex_mode = 3;     // ranges per large call (two walkers per SIMD); scan switches (XRIT_CX_MODE)
    int run_exact(const float2 *in, float2 *out, size_t n, hipStream_t s, Profiler *prof);
    unsigned *ex_cnt = nullptr;             // the last exact call's counters (exact_walk.h: xw::NCNT words, device)
    int exact_counters(unsigned *out8, hipStream_t s);
    // the same in two halves around the kernel that produces `in` (the decimator): fused_begin() before it
    // (hands out the epilogue descriptor), fused_finish() after it -- the stream is swept twice, not three times
    int fused_begin(size_t n, int per_lane, hipStream_t s, AgcEpilogue *epi);      // per_lane: the producer's RC
    int fused_finish(const float2 *in, float2 *out, size_t n, int per_lane, hipStream_t s, Profiler *prof);
    // no producer with an epilogue in front: one read-only sweep composes the run maps (then fused_scan())
    int fused_reduce(const float2 *in, size_t n, int per_lane, hipStream_t s, Profiler *prof);
    // instead of fused_finish(): scan only; the consumer (the matched filter) applies the gains in its window fill.
    // `fallback` receives the serial result if the guard trips.
    int fused_scan(const float2 *in, float2 *fallback, size_t n, int per_lane, hipStream_t s, Profiler *prof, AgcFill *fill);
    int gain(float *g, hipStream_t s);
    int fallback_flag(float *flag, hipStream_t s);
    // the same without a wait of its own: the copy goes into a pinned word behind whatever is queued on s, and
    // the caller reads it after its next synchronise
    int request_flag(hipStream_t s);
    int request_flag_at(const float *flag, hipStream_t s);      // a given call's flag slot (the stage may have moved on)
    float requested_flag() const { return h_flag ? *h_flag : 0.0f; }
    float *h_flag = nullptr;
};

// ---- CostasLoop (demodulator.cpp:448; Work at :152) -----------------------
struct CostasStage {
    CostasGains gains{};
    int L = 256;            // samples per chain
    int max_passes = 32;
    float trust = 1.0f, tol_phase = 1e-5f, tol_freq = 3e-8f;
    DevBuf state;           // float2 (phase, freq) carried across calls
    DevBuf S, E, J, stat, sub, dlin, work, flags, counters, wsolve, rescue;
    // round 4: one Newton step on a model of the loop over runs of 8 samples refines the guesses before the first pass
    // over the samples (costas_model_pass_kernel); that pass is accepted on prediction up to model_accept (rad)
    bool model_step = true;
    float model_accept = 5e-3f;
    bool force_gated = false;         // always the three-launch solve with the trust gate (XRIT_GATED_SOLVE=1)
    int final_warm = 0;             // chains of warm-up in front of every chain of the final pass (cfg.front_exact: 4; costas.hip)
    bool keep_spare = false, trace_env = false, no_serial_walk = false;   // XRIT_KEEP_SPARE, XRIT_TRACE, XRIT_NO_SERIAL_WALK (read in init)
    // a hand-off still open after rescue_after passes is walked serially between its first and last open boundary
    // (costas_serial_states_kernel), if that is at most rescue_max_samples samples
    int rescue_after = 32;
    long long rescue_max_samples = 4 << 20;
    unsigned rescues = 0;             // calls of this handle that took the serial walk
    bool walked = false;              // ... the last call did
    int serial_rescue(hipStream_t s, Profiler *prof);
    unsigned *h_counters = nullptr;   // pinned
    int cur = 0;
    int passes = 0;
    unsigned unconverged = 0;
    float max_residual = 0;
    int init(float loop_bw, int chain_len, int max_passes);
    int reset(hipStream_t s);
    void release();
    // sub_ext (optional): sum z^2 per run of 8 samples already left by the producer (FIR epilogue); om (optional): receives
    // the clock recovery's timing-line statistic per chain (index offset om_off in that stage's input buffer)
    int run(const float2 *in, float2 *out, size_t n, hipStream_t s, Profiler *prof, const float2 *sub_ext = nullptr,
            double2 *om = nullptr, long long om_off = 0, double inv_sps = 0.0);
    // the same in two halves: begin() only enqueues (guess, a batch of passes with a device-side stop test, final
    // pass); finish() runs after the caller synchronised the stream and continues the passes if they did not close
    int begin(const float2 *in, float2 *out, size_t n, hipStream_t s, Profiler *prof, const float2 *sub_ext,
              double2 *om, long long om_off, double inv_sps, bool use_exact = false);
    bool closed() const;
    int finish(hipStream_t s, Profiler *prof, bool *redone);
    int enqueue_passes(int count, hipStream_t s, Profiler *prof);
    int enqueue_final(hipStream_t s, Profiler *prof);
    struct Job {
        const float2 *in = nullptr; float2 *out = nullptr; size_t n = 0; int K = 0; int enqueued = 0;
        double2 *om = nullptr; long long om_off = 0; double inv_sps = 0;
        bool gated = false;     // this call has gone over to the gated solve
        bool rescued = false;   // the serial walk has been tried
        bool exact = false;     // this call's output is put on the serial trajectory (costas_exact.hip)
        float model_accept = 0; // CostasPolicy::model_accept of this call: the stage's on a tracking loop, else 0
    } job;
    int batch = 4;          // passes enqueued before the host looks: what the previous call needed + a spare one
    int stable = 0, last_passes = -1;   // calls in a row that closed inside their batch with the same count
                            // (2 on a locked signal); every surplus pass is ~4 no-op launches of ~5 us
    int get_state(float *phase, float *freq, hipStream_t s);
    // cfg.front_exact = 2 (costas_exact.hip): behind the final pass the output is put ON the serial float32 trajectory by
    // exactly walked, overlapping ranges; finish() closes what joints are still open
    bool exact = false;
    int ex_mode = 3;                // bit 0: the three-instruction systolic round where neither wrap nor limiter acts; bit 1: the
                                    // lattice scan in front of it (XRIT_CX_MODE: A/B runs)
    int ex_hist = -1;               // >= 0: samples of warm-up in front of every range whatever the call (XRIT_CX_HIST); -1: by plan
    int ex_prio = 1;                // the walkers' waves at a raised issue priority (XRIT_CX_PRIO=0: off)
    int ex_walkers = 2048;          // ranges per large call: two walkers per SIMD (set from the device's CU count in init; one wave
                                    // issues a dependent instruction every ~7 cycles, two share a SIMD's port almost for free)
    DevBuf xj, xbs, xcnt;           // joints (start / end / used), block records, counters
    unsigned *h_xcnt = nullptr;     // pinned: [0] joints open, [1] blocks, [2] Picard rounds, [3] blocks that hit the round limit
    int ex_W = 0, ex_rounds = 0;
    bool ex_args_valid = false;
    unsigned ex_open = 0, ex_nonconverged = 0;
    unsigned long long ex_blocks = 0, ex_picard = 0, ex_segs = 0, ex_fallbacks = 0;    // totals over the handle's calls (statistics)
    int exact_plan(size_t n, int *Lw, int *W, int *H) const;
    int enqueue_exact(hipStream_t s, Profiler *prof);
    int finish_exact(hipStream_t s, Profiler *prof, bool *redone);
    int flip_phase(hipStream_t s);      // the carried phase moves by pi: the other of the loop's two locks
};

// the exact Costas loop's sincosf (exact_sincos.h) on an array: what tests/test_gpu_exact.py holds against the oracle's
int launch_loop_sincosf(const float *d_x, float *d_sin, float *d_cos, size_t n, hipStream_t s);

// ---- ClockRecovery (demodulator.cpp:449; Work at :156) --------------------
struct ClockStage {
    ClockPar par{};
    float sps = 0, mu0 = 0.5f;
    int NS = 64;            // symbols per chain (auto_ns: chosen per call so that its waves fill whole generations)
    int cu_count = 256;     // what the chip holds at once decides the chain length, see ClockStage::begin
    long long lds_per_cu = 160 * 1024;
    bool auto_ns = true;
    bool serial = false;    // one trajectory, no chains (cfg.clock_serial): the floor measurement, ~0.3 us per symbol
    int max_passes = 48, min_passes = 4;
    int jac_passes = 1;     // passes that recompute the chain Jacobians (then quasi-Newton; measured: no gain from more)
    DevBuf table;           // 129 x 8 MMSE taps
#ifndef XRIT_AHEAD
#define XRIT_AHEAD 2            // inputs that may wait behind the call in progress (bursts that walk overlapping blocks)
#endif
    static constexpr int NXB = XRIT_AHEAD + 1;
    DevBuf xbuf[NXB];       // [pad | carry | new] input samples of a call.  Three of them (round 5): the producer of the samples of
                            // the call after next (the Costas loop of burst b + 2: xrit_demod_prefetch_device) fills one while the
                            // walkers of bursts b and b + 1 read the other two (round 4: two, one burst ahead)
    int xb = 0;             // the buffer the call in flight (the last call) reads
    int x_pending = -1;     // the buffer input_slot() handed out for the next call
    bool in_flight = false; // between begin() and finish()
    // The new samples (the Costas loop's output rows of 128 bytes) start at a fixed place, xpad samples into the buffer, on a
    // 128-byte boundary -- wherever the producer writes them it does not need to know how many samples the call before left
    // unread --; those `carry` samples (at most 1024) are copied in right in front of them when the call begins.  Round 5: the
    // pad also holds the HISTORY the overlapping walkers of clock_overlap.h warm up over (the last samples of the burst before).
    size_t xpad = 1024;
    float2 *xbase_fixed = nullptr;
    float2 *xdata() const { return xbuf[xb].as<float2>() + xpad; }
    float2 *xbase() const { return xbase_fixed ? xbase_fixed : xdata() - carry; }
    DevBuf st;              // carried ClockState + carry count
    DevBuf S, E, J, om, work, counters, sym, dlin, flags, wsolve, jmean;
    bool jmean_valid = false;         // jmean holds the mean chain Jacobian of an earlier, locked call with ...
    int jmean_ns = 0;                 // ... this chain length
    bool force_gated = false;         // always the three-launch solve with the trust gate (XRIT_GATED_SOLVE=1: A/B runs)
    DevBuf tail;                      // 2 x 1024 samples left unread by a call (ping-pong with the state)
    void *h_res = nullptr;            // pinned copy of the control block + result
    float tol_t = 2e-6f, tol_w = 2e-7f;
    int cur = 0;
    size_t carry = 0;       // samples held over from the previous call
    int passes = 0;
    unsigned unconverged = 0;   // boundaries the last solve still moved (~all of them on any healthy call: the floor)
    unsigned large_open = 0;    // boundaries left with a residual beyond 0.02 sample or an open symbol slip
    float max_residual = 0;
    size_t last_symbols = 0;
    int init(float omega, float gain_omega, float mu, float gain_mu, float omega_rel_limit, int chain_syms,
             int max_passes);
    int reset(hipStream_t s);
    void release();
    // where the producer must write the n new samples of this call
    int input_slot(size_t n, float2 **slot, hipStream_t s);
    // (the statistic's phasor counts samples from the first NEW sample of the call: begin() turns it by the carried ones)
    double2 *om_slot(int nb, int BL);
    bool om_ext = false;
    int om_nb = 0, om_BL = 256;
    // (round 4: ... and the producer's stream also unwraps it into the symbol-count curve, om_scan(): two launches less between
    // the end of one burst's relay and the start of the next one's)
    bool om_scanned = false;
    DevBuf om_work;
    int om_scan(hipStream_t s);
    // soft (real parts) and/or complex symbols; either may be null
    int run(size_t n, float *soft_out, float2 *sym_out, size_t cap, size_t *n_out, hipStream_t s, Profiler *prof);
    // The last call once more on its sign-flipped input (one capture across GPUs: this rank's Costas loop turned out
    // to sit pi away from the stream's, csrc/group.hip): the carried state goes back to what it was before the
    // call, its history and unread tail change sign, the call's input is negated in place, the recovery runs again.
    int redo_flipped(float *soft_out, float2 *sym_out, size_t cap, size_t *n_out, hipStream_t s, Profiler *prof);
    size_t prev_carry = 0, prev_n = 0;      // of the last call
    bool redo_ok = false;                   // ... which ended normally (its input is still in xbuf)
    // What the loop carries from one call to the next as ONE device record (one capture across GPUs, csrc/group.hip: the rank
    // in front hands it on, so that the stream has one loop state across all slices as the reference has across all chunks,
    // demodulator.cpp:446-450): [valid, carry, -, -], the ClockState, then the unread tail (1024 samples, zero padded).
    static constexpr size_t CARRY_HEAD = 64, CARRY_BYTES = 64 + 1024 * sizeof(float2);
    // which = 0: what the NEXT call starts from; 1: what the last call started from (untouched since, as redo_flipped relies on)
    int export_carry(void *d_rec, int which, hipStream_t s);
    // The last call once more from the record of another handle (`carry_rec` samples of unread tail in it, read by the caller):
    // the same input, the given loop state in front of it.
    int redo_from(const void *d_rec, size_t carry_rec, float *soft_out, float2 *sym_out, size_t cap, size_t *n_out, hipStream_t s, Profiler *prof);
    // the last call's symbols are those of ONE float32 walk from its start state (cfg.clock_serial, the exact closure, or a call
    // short enough for a single exact walk): what a hand-over of the exact start state turns into the stream's own words
    bool last_walk_exact() const;
    // What the recovery would carry had it run on the sign-flipped stream so far (make_alt(): the last call -- the
    // halo of a time slice, from a cold start -- is run again on its negated input and the outcome kept aside); a
    // later redo_flipped() starts from it instead of from the other sign's state with its history negated, whose
    // timing sits 1e-3 sample off the flipped loop's and takes 1e4..1e5 symbols to come home on the lattice.
    DevBuf alt;                             // ClockState + 1024 samples of unread tail, twice (the second: scratch)
    size_t alt_carry = 0;
    bool alt_valid = false;
    int make_alt(hipStream_t s, Profiler *prof);
    // the same in two halves (see CostasStage): begin() only enqueues, finish() runs after a stream synchronise
    int begin(size_t n, float *soft_out, float2 *sym_out, size_t cap, hipStream_t s, Profiler *prof);
    bool closed() const;
    int finish(size_t *n_out, hipStream_t s, Profiler *prof);
    int enqueue_passes(int count, hipStream_t s, Profiler *prof);
    int enqueue_output(hipStream_t s, Profiler *prof, bool again = false);
    struct Job {
        size_t n = 0, cap = 0; float *soft = nullptr; float2 *sym = nullptr;
        long long N = 0, ni = 0; int K = 0, enqueued = 0, SS = 0, W = 0, WS = 0, A = 0, STEP = 0; bool wide = false, short_input = false;
        size_t tile_bytes = 0, tile_bytes3 = 0;   // one-wave groups (NG per workgroup) / the three-wave Jacobian pass
        int NG = 1;
        bool mean_j = false;    // the passes use the stream's mean Jacobian: no finite-difference pass
        bool gated = false;     // this call has gone over to the gated solve
        bool rescued = false;   // the serial walk has been tried
        float model_accept = 0; // CostasPolicy::model_accept of this call: the stage's on a tracking loop, else 0
        int *dirty = nullptr, *counts = nullptr, *nrun = nullptr, *terminal = nullptr, *written = nullptr;
        int G = 0, cps = 0, relay_enq = 0;      // exact closure: segments, chains per segment, passes enqueued
        int relay_w = 0;                        // waves per walker team (clock_relay_wide.h); 0: the one-wave walker
        bool relay = false;                     // ... is on for this call
        bool relay_force = false;               // ... although the tiled hand-off never closed (pass budget used up)
        bool no_handoff = false;                // ... straight from the timing guess: no hand-off passes at all
        int relay_budget = 0;                   // relay passes at most (0: until closed)
        int relay_apx[2] = {0, 0};              // how the relay's first two passes walk (clock_relay_kernel's apx): 0 exactly
        bool relay_long = false;                // the default configuration on segments of >= auto_long_seg symbols: two passes, no watch on the starts
        int write_from = 0x7fffffff;            // hand-off passes from this one on leave the symbols (ClockPassOut)
    } job;
    bool pass_writes = true;    // the last hand-off pass is the output pass (XRIT_NO_PASS_OUTPUT=1, read at init: a separate output pass)
    int batch = 7;          // passes enqueued before the host looks: what the previous call needed + a spare one
    int last_passes = -1;
                            // (5-6 in steady state)
    // ---- exact closure (clock_relay.h, cfg.clock_exact): segments of the call walked exactly, relayed until the
    // serial trajectory is reproduced bit for bit
    int exact = 0;              // 0 (default), by call size (ClockStage::begin): one exact walk / two hand-off passes + auto_passes relay
                                // passes / no hand-off passes, relayed from the timing guess (bursts that fill the chip); four more passes at
                                // a time while the segment starts still move by more than auto_shift (low Es/N0), to closure when the
                                // hand-off never closed (cfg.clock_exact = -3 arrives here as 0 with relay_quick set);
                                // 1: always until closed; n > 1: n relay passes, nothing else; -2: hand-off passes only (the
                                // fast configuration: five of them, 2.6e-4 rms from the serial trajectory), relayed to closure
                                // when they stall above auto_rms or never close (round 2's default); -1: never relayed
    float auto_rms = 3e-4f;     // (calls too short for the relay to be planned -- fewer than auto_min symbols -- and
                                // cfg.clock_exact = 0: closed exactly when the hand-off stalls above this rms residual, samples)
    // the default's three relay passes: measured at C2 after the third, the starts move by 3.8e-4 sample rms at Es/N0
    // 12 dB, 1.1e-3 at 6 dB, 4.3e-3 at 3 dB (XRIT_AUTO_PASSES=n: another count; 0: the hand-off passes' result as in
    // round 2 unless they stall)
    int auto_passes = 3;
    long long auto_guess_min = 6000000;   // symbols from which the default configuration relays straight from the timing guess
    int auto_long_seg = 49152;  // symbols per segment from which two relay passes are the default's budget (ClockStage::begin)
    // A call of up to this many symbols is ONE exact walk from the carried state (0: three of the default's shortest segments,
    // 73.7 k symbols -- no slower than three passes over a third of it).  Handles whose calls of this size take the bit-exact front end (cfg.front_exact 0 / 2) set 200 k: every
    // chunk the reference hands its blocks (up to 512 Ki samples: 123 k symbols LRIT, 194 k HRIT) then comes out as the CPU
    // chain's words, at 56 symbols per microsecond on the one walking wave.
    int one_walk_max = 0;
    long long one_walk_limit() const { return one_walk_max > 0 ? (long long)one_walk_max : 3LL * (auto_long_seg / 2); }
    float auto_shift = 6e-4f;
    float auto_snr = 10.0f;     // ... and to closure outright when the first pass's soft symbols show 2 Es/N0 below this (7 dB)
    float auto_snr_floor = 2.0f;  // ... but not below this: no signal (noise alone shows 1.75)
    float snr_estimate = 0.f;   // (of the last call that looked)
    long long auto_min = 1;     // (round 4: a short call is one exact walk of a few dozen steps -- cheaper than hand-off passes that run until they close)
    bool relay_by_default() const { return exact >= 1 || (exact == 0 && auto_passes > 0); }
    bool relay_auto = false;    // ... the last call was
    int relay_window = 0;       // chains per segment (0: chosen per call, ~4 segments per CU)
    std::function<int(int)> before_relay;   // called by begin() in front of the relay kernels of a call that plans them (the chain
                                         // starts the front end of the next burst there: xrit_demod_prefetch_device)
    DevBuf relay;               // segment records + per-pass counters
    DevBuf relay_rec;           // per symbol: read index and interpolator arm of the last exact walk (the next walk's first guess)
    bool relay_no_rec = false;  // XRIT_RELAY_NO_REC: walkers always guess from the nominal rate (A/B runs)
    bool relay_no_claim = false;  // XRIT_RELAY_NO_CLAIM: the walker is always wave 0 of its workgroup, wherever the hardware put it (A/B runs)
    DevBuf stage;               // soft symbols of a writing hand-off pass, in wave order (ClockPassOut::stage)
    int relay_batch = 96;       // relay passes enqueued before the host looks
    int relay_passes = 0;       // relay passes the last call ran (the closing, change-free one included)
    bool relay_closed = false;  // ... and whether they reproduced the serial trajectory
    int relay_segments = 0, relay_seg_chains = 0;
    bool trace_env = false;     // XRIT_TRACE=1 (read at init): per-pass statistics on stderr
    bool no_meanj = false;      // XRIT_NO_MEANJ=1: finite-difference Jacobians in every call
    int ng_max = 8;             // XRIT_CLOCK_NG: one-wave groups per clock workgroup at most
    bool relay_global = false;  // walk from global memory even where the LDS-staged kernel applies (A/B runs)
    int relay_per_cu = 3;       // XRIT_RELAY_PER_CU: segments (walkers) per CU the relay plans (A/B runs)
    bool relay_per_cu_set = false;
    int relay_no_handoff = -1;       // the relay's first pass starts from the timing guess, no hand-off passes: -1: in the default
                                     // configuration (cfg.clock_exact = 0); XRIT_NO_HANDOFF=0 / 1: never / with every relayed call (A/B runs)
    bool relay_rec01 = true;           // a plan from the timing guess records its first pass and guesses from that in its second (XRIT_RELAY_REC01=0: not; A/B runs)
    bool relay_quick = false;          // cfg.clock_exact = -3: the default configuration with its first relay passes walked approximately
    int relay_apx_cfg[2] = {-1, -1};   // XRIT_RELAY_APX=a,b: the walk of the relay's first two passes (-1: the configuration's choice; A/B runs)
    int relay_waves = 1;        // waves per walker team at most (clock_relay_wide.h; < 2: the one-wave walker of clock_relay.h)
    int relay_teams_per_cu = 1; // segments (teams) per CU the relay plans with the wide walker
    int enqueue_relay(int count, bool restart, hipStream_t s, Profiler *prof);
    int relay_limit() const;
    int relay_plan();

    // ---- overlapping exactly walked blocks (clock_overlap.h, round 5): the default configuration's plan for calls of ov_min
    // symbols or more.  A job is one such call; up to three exist at a time (xrit_demod_prefetch_device: the walkers of bursts
    // b and b + 1 at work, burst b + 2's samples being written), each with its own sample buffer (xbuf[job]) and timing curve.
    struct OvJob {
        int state = 0;                  // 0: free; 1: slot handed out (samples being produced); 2: walkers enqueued; 3: being finalized
        size_t n = 0;                   // new samples
        long long N = 0, ni = 0;        // samples in [history | new], and how many of them a symbol may start at
        int padN = 0;                   // history samples in front of the new ones
        int G = 0, Ls = 0, first_bound = 0, store0 = 0, early = 0, stride = 0, hist = 0;
        bool w0_carried = false;        // walker 0 starts from the carried state (no history to warm up over)
        int w0_ii = 0;                  // buffer index the carried state's read index 0 corresponds to (padN - carry)
        bool ahead = false;             // enqueued before the call in front of it had finished (its carry is not known to the plan)
        int nb = 0, BL = 256;           // the timing statistic: nb blocks of BL samples, counted from the first new sample
        bool om_ext = false, om_scanned = false;
        DevBuf om, om_work, segs, S, stage, aux;
        hipEvent_t ev_walk = nullptr;   // the walkers have finished
        hipEvent_t ev_guess = nullptr;  // the history has been copied in and the start states computed
        int hist_src = -1;              // the buffer whose last samples this job's first walkers warm up over (-1: none)
        unsigned long long serial = 0;  // order of the calls (the job in front: serial - 1)
    } ov[NXB];
    bool ov_allow = true;       // the caller takes float soft symbols only (the chain sets it per call; stage objects: off)
    bool ov_enabled = true;     // XRIT_NO_OVERLAP=1 (read at init): the relay of clock_relay.h for every call, as in round 4
    long long ov_min = 1000000; // symbols from which the default configuration walks overlapping blocks
    int ov_hist = 49152;        // symbols of exactly walked history in front of every range (XRIT_OV_HIST)
    int ov_min_range = 8192;    // symbols per range at least (XRIT_OV_MINL)
    int ov_min_walkers = 240;   // walkers at least, where ranges of 8192 symbols allow (XRIT_OV_MINW): few walkers, long latency
    bool ov_small_ring = false; // XRIT_OV_SMALL_RING=1: sample rings of 1024 instead of 2048 samples where a block's span allows
    double ov_lratio = 1.0;     // range length aimed at, in histories (XRIT_OV_LRATIO): every symbol is walked 1 + 1 / ov_lratio times
    int ov_job = -1;            // the job of the call being set up / in flight (-1: the call is not such a call)
    unsigned long long ov_serial = 0;
    // what the last finished call left in its buffer: the history the next call's first walkers warm up over
    int hist_xb = -1;           // buffer
    size_t hist_len = 0;        // samples in it that end where the stream stands ([pad | new] of that call, contiguous)
    int hist_job = -1;          // ... and, when that call was an overlap job, the job whose timing curve covers them
    DevBuf ov_claim;            // which SIMDs hold a walker (clock_relay.h), shared by the jobs in flight
    int ov_pad_need = 0;        // history samples a warm walker 0 needs in front of the new samples
    bool ov_eligible(size_t n) const;
    int ov_plan(OvJob &j, bool ahead);
    // the walkers of the call whose samples have been produced into the slot input_slot() handed out (and whose timing curve
    // has been scanned): pad copy, start states, walkers, on `sw` -- ahead of the call itself (ahead: the call in front has not
    // finished) or from begin()
    int ov_launch(int job, hipStream_t sw, bool ahead, Profiler *prof);
    bool ov_can_launch_ahead(int job) const;
    int ov_finalize(int job, float *soft_out, size_t cap, hipStream_t s, Profiler *prof);
    // a job whose walkers were started ahead on samples that have since been rewritten (the Costas loop went on from the host):
    // waits for them (on `s`'s behalf: the host synchronises the event) and takes the job back to "samples produced"
    int ov_restart(int job, hipStream_t s);
    int ov_fallback(size_t *n_out, hipStream_t s, Profiler *prof, bool to_closure);
    bool ov_scan_now = false;   // om_scan() is being called from ov_launch (an overlap job's curve is unwrapped on its walkers' stream)
    int ov_cur = -1;            // the overlap job of the call between begin() and finish()
    float *ov_soft = nullptr; size_t ov_cap = 0; size_t carry_before_fallback = 0;
    bool ov_fell_back = false;  // the last call's overlap result was not taken (low Es/N0, a joint that did not fit): relayed to closure
    int ov_walkers = 0;         // walkers of the last overlap call
    float ov_joint_max = 0.f;   // ... and the largest distance between two trajectories at a joint (samples)
};

// ---- helpers ---------------------------------------------------------------
int launch_sync_correlate(const int8_t *data, size_t n, const unsigned long long *words, int nwords, unsigned frame,
                          xrit_sync_hit *hits, hipStream_t s);
int launch_sync_fix(const int8_t *data, size_t n, const xrit_sync_hit *hits, unsigned frame, unsigned min_corr,
                    int8_t *frames, unsigned char *valid, hipStream_t s);
int launch_quantize_i8(const float *in, int8_t *out, size_t n, hipStream_t s);
int launch_convert(const void *in, int type, float2 *out, size_t n, hipStream_t s);
int launch_synth(const xrit_synth_params &p, uint64_t start, size_t n, float2 *out, hipStream_t s);
int launch_read_bw(const void *buf, size_t bytes, int reps, hipStream_t s, double *gbs);

}  // namespace xrit
