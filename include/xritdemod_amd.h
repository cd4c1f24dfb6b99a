/*
 * xritdemod_amd.h -- C ABI of the MI355X-native xRIT BPSK demodulation chain.
 *
 * The reference (opensatelliteproject/xritdemod) has no FFI for this path; the
 * seam is the libSatHelper C++ class API as used by
 * demodulator/src/demodulator.cpp.  Every entry point below names the
 * reference interface it replaces (paths relative to /root/reference).
 *
 * Conventions
 *   - plain C: opaque handles, pointers and sizes; no exceptions cross the ABI.
 *   - return value: 0 (XRIT_OK) or a negative XRIT_E_* code; xrit_last_error()
 *     gives the text of the last failure on the calling thread.
 *   - complex samples are interleaved (re, im) float32 = std::complex<float>.
 *   - "host" entry points take host pointers and do the H2D/D2H copies;
 *     "_device" entry points take device pointers (HBM-resident data) and a
 *     hipStream_t passed as void* (NULL = the handle's own stream).
 *   - a handle is single-consumer (the reference calls Work() from one thread,
 *     demodulator.cpp:170-175,475); distinct handles may be used concurrently.
 *   - the compute path is HIP only: if no HIP device is usable, create fails
 *     with XRIT_E_NO_DEVICE.  There is no CPU fallback.
 */
#ifndef XRITDEMOD_AMD_H_
#define XRITDEMOD_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XRIT_OK            0
#define XRIT_E_INVALID    -1   /* bad argument */
#define XRIT_E_NO_DEVICE  -2   /* no usable HIP device */
#define XRIT_E_HIP        -3   /* HIP runtime error (text in xrit_last_error) */
#define XRIT_E_NOMEM      -4
#define XRIT_E_CAPACITY   -5   /* output buffer too small */
#define XRIT_E_NOT_CONVERGED -6 /* strict mode only: hand-off passes exhausted */

/* FrontendDevice.h:11-13 */
#define XRIT_SAMPLE_FLOATIQ 0
#define XRIT_SAMPLE_S16IQ   1
#define XRIT_SAMPLE_S8IQ    2
/* not a FrontendDevice.h type: the RTL-SDR frontend converts its unsigned bytes itself and hands floats to
 * onSamplesAvailable (RtlFrontend.cpp:102-116: lut[b] = (b - 128) / 127.f, then a running-average DC tracker with
 * alpha = 1 - exp(-1 / (0.05 sampleRate)), :57).  With this type the chain takes the raw bytes and does that
 * conversion on the device, statement for statement (including the reference's `i % 1`, which makes one average
 * serve I and Q alike). */
#define XRIT_SAMPLE_U8IQ    3

const char *xrit_last_error(void);
const char *xrit_version(void);
/* number of visible HIP devices (0 when none / runtime unusable) */
int xrit_device_count(void);
/* 1 when the library was built with the measurement switches of DESIGN.md section 7 (make EXTRA=-DXRIT_EXPERIMENTS: the
 * matrix-pipe decimator, walker teams, A/B environment switches); the shipped build has none of them: 0 */
int xrit_build_experiments(void);

/* ------------------------------------------------------------------------
 * Tap designers -- SatHelper::Filters (demodulator.cpp:443-444), host only.
 * ------------------------------------------------------------------------ */
/* Filters::lowPass(gain, sampleRate, cutFreq, transitionWidth, HAMMING, beta)
 * -> number of taps written, or the negative required count if cap is short */
int xrit_lowpass_taps(double gain, double sample_rate, double cutoff, double transition_width,
                      float *taps, int cap);
/* Filters::RRC(gain, sampleRate, symbolRate, alpha, nTaps) -> taps written */
int xrit_rrc_taps(double gain, double sample_rate, double symbol_rate, double alpha, int ntaps,
                  float *taps, int cap);
/* the 129 x 8 MMSE interpolator table used by ClockRecovery */
void xrit_mmse_table(float *table /* [129*8] */);

/* ------------------------------------------------------------------------
 * The chain -- processSamples(), demodulator.cpp:100-168, with the objects
 * main() builds at demodulator.cpp:436-450.
 * ------------------------------------------------------------------------ */
typedef struct xrit_demod xrit_demod;

typedef struct xrit_demod_config {
    /* reference parameters */
    float    sample_rate;       /* device->GetSampleRate(), demodulator.cpp:436 */
    uint32_t decimation;        /* baseDecimation, cfg "decimation" (:300-301) */
    uint32_t symbol_rate;       /* cfg "symbolRate"; LRIT 293883 / HRIT 927000 */
    float    rrc_alpha;         /* cfg "rrcAlpha"; 0.5 / 0.3 */
    int32_t  rrc_taps;          /* RRC_TAPS = 63, Parameters.h:28 */
    float    agc_rate, agc_reference, agc_gain, agc_max_gain;   /* Parameters.h:34-37 */
    float    pll_alpha;         /* Costas loop bandwidth = CLOCK_ALPHA (demodulator.cpp:220) */
    float    clock_mu, clock_alpha, clock_gain_omega, clock_omega_limit; /* Parameters.h:30-33 */
    /* placement */
    int32_t  device;            /* HIP device ordinal */
    /* time-slice tiling of the feedback loops (0 = library default) */
    int32_t  costas_chain_len;  /* samples per Costas chain: 0 = 256, else 16..320 (rejected beyond: a chain that is
                                 * long against the loop's pull-in time ends in a state that is no longer a smooth
                                 * function of its start, and the hand-off solve then needs tens of passes or does
                                 * not close -- fuzz, HRIT at 512: 186 passes and flipped decisions) */
    int32_t  clock_chain_syms;  /* symbols per clock-recovery chain; 0 = chosen per call (64..256) so that the call's
                                 * waves fill whole generations of what the chip holds.  Every chain boundary is a
                                 * place where the recovered clock may differ from the serial loop's by the loop's
                                 * own chaos level, so short chains cost parity; values below
                                 * 32 are raised to 32 (16-symbol chains measured 6.5e-4 rms against 2.2e-4 at
                                 * the default and were seen to mis-resolve a symbol slip at Es/N0 < 4 dB) */
    int32_t  max_passes;        /* hand-off passes before giving up (per loop); 0 = 192: a locked signal needs 2 + 5,
                                 * a cold start within the loops' lock-in range ~15, a pull-in with cycle slips
                                 * a pass or two per chain of the slipping stretch */
    int32_t  strict;            /* 1: XRIT_E_NOT_CONVERGED when the Costas hand-off stays above its tolerance or the
                                 * clock hand-off ends with large residuals / open slips (stats.clock_open_large) */
    int32_t  clock_min_passes;  /* clock hand-off passes always run (0 = default); max_passes caps both loops */
    int32_t  slices;            /* ignored (round 1 could cut a call into time slices on two streams; measured slower
                                   on MI355X and removed) */
    int32_t  clock_serial;      /* 1: the clock recovery runs as ONE serial trajectory on the device (a single wave,
                                 * ~0.3 us per symbol) instead of time-tiled chains: no hand-offs, so what is left
                                 * against the CPU chain is what any float32 M&M fed by this chain's Costas output
                                 * shows (the recurrence lives on a 2^-21-sample lattice and keeps a one-ulp
                                 * difference for ~1e5 symbols).  A diagnostic, not a production mode. */
    int32_t  clock_exact;       /* the relay of the clock recovery (csrc/clock_relay.h): the call is cut into segments that
                                 * are walked with the literal recurrence, 64 symbols per step, from the end states the
                                 * segments in front of them reached a pass earlier.  Every pass extends what a segment
                                 * knows of its past by one segment (no shorter than 16 384 symbols unless the relay
                                 * runs to closure); a pass that changes nothing has reproduced the serial trajectory
                                 * (clock_serial = 1) bit for bit.
                                 *   0 (default), by call size (round 4): up to 74 k symbols (200 k where the call takes the bit-exact front end: cfg.front_exact 0 / 2) ONE exact walk from the carried
                                 *      state (the serial trajectory itself); from 6 M symbols -- every BASELINE burst -- no hand-off
                                 *      passes at all: two walkers per CU, the first relay pass from the timing guess, three passes at
                                 *      24.8 k symbols per segment, two from 49 k (soft symbols 5.6e-5 rms from the serial
                                 *      trajectory, within 5 % of its own distance from the CPU chain; 2.1 ms per streamed 256 Mi-sample
                                 *      burst); in between two hand-off passes, then three relay passes.  Walked to closure on its own
                                 *      when the soft symbols of the first relay pass show Es/N0 below 7 dB (hard decisions
                                 *      would otherwise differ from the serial loop's), when a segment start moves by a quarter of
                                 *      a symbol between passes (the guess miscounted) or the hand-off passes never closed,
                                 *      and four passes at a time while the segment starts still move by more than 6e-4
                                 *      sample rms from pass to pass (stats.clock_relay_passes / clock_relay_closed);
                                 *   1: every call to closure (~9 ms per 256 Mi-sample burst; the serial wave: 3.9 s);
                                 *   n > 1: two hand-off passes and n relay passes, nothing else;
                                 *   -2: hand-off passes only, five of them on a clean signal -- the fast configuration
                                 *      (2.1 ms per burst, soft symbols 2.2e-4 .. 2.6e-4 rms from the serial trajectory:
                                 *      DESIGN.md section 6); a call whose passes stall above 3e-4 sample rms (low Es/N0)
                                 *      or never close is still relayed to closure (rounds 2-3's default);
                                 *   -1: hand-off passes only, never relayed;
                                 *   -3: the quick relay -- the default's plan with the passes in front of its last walked
                                 *      approximately (one guess round, then two; no literal verification, nothing stored: they are
                                 *      there for their end states): 1.92 ms per streamed burst, 1.15e-4 rms from the serial
                                 *      trajectory.  Faster AND closer than -2. */
    int32_t  clock_exact_window;/* chains per relay segment; 0 = chosen per call (~4 segments per CU) */
    int32_t  front_exact;       /* which front end a call takes (round 6).  0 (default): the bit-exact one (see 2) on calls below the
                                 * big-burst size -- the reference's chunk sizes and everything up to a million symbols, where a call
                                 * is launch latency, not throughput: a call of up to 200 k symbols (every chunk the reference hands its blocks, 512 Ki
                                 * samples = 123 k symbols LRIT / 194 k HRIT included) is ONE exact walk of the clock recovery and yields the CPU chain's soft
                                 * symbols word for word -- and the fast one on bursts of a million symbols and more (what `value`
                                 * is quoted on); -1: the fast one on calls of every size (round 5's default).
                                 * The fast front end:  What the soft symbols' distance from the CPU chain is made
                                 * of is the front end's distance at the Costas loop's output (1.1e-6 rms as shipped; ANY distance
                                 * costs a float32 M&M 5.5e-5 on LRIT), and most of that is what the
                                 * hand-offs between the loop's 256-sample chains leave: a chain's start a few 1e-6 rad beside its
                                 * predecessor's end (a chain walked again from a start an ulp away rounds its 256 steps afresh: more passes do not
                                 * remove it), which
                                 * the loop forgets at a half per ~600 samples.
                                 *   1: the Costas loop's final pass starts every chain FOUR chains early and walks those samples
                                 *      quietly -- the stage's distance from the serial loop 1.16e-6 -> 6.1e-7, the soft symbols on
                                 *      steady-state 256 Mi-sample LRIT bursts 9.8e-5 -> 8.4e-5 rms (five bursts; HRIT 1.33e-4 ->
                                 *      1.20e-4: still a miss), for 12 % of such a burst's time (3-5 % at decimation 32; 20-30 % without a decimator, where the
                                 *      circuit-rate stream is the whole burst: profiles/r5_costas_variants_parity.json, r5_bench_c*.json).
                                 *   2 (round 6): the front end BIT FOR BIT the CPU chain's through the Costas loop -- both filters
                                 *      summed in the CPU chain's order without FMA (fir.hip: fir_exact_kernel), the AGC walked
                                 *      literally in warmed-up chains (agc.hip), the Costas loop walked exactly 64 samples per
                                 *      step with the C library's sincosf evaluated in double precision (costas_exact.hip).
                                 *      What is left of the soft symbols' distance is the clock recovery's own (its distance
                                 *      from the serial trajectory, 5-6e-5 LRIT) -- and on calls of up to 200 k symbols nothing: in
                                 *      this mode, as in the default, such a call's clock recovery is ONE exact walk from the carried state
                                 *      (with the fast front end on calls of every size, -1 / 1: up to 74 k), so every chunk the reference hands its blocks --
                                 *      32 Ki .. 512 Ki samples, at most 123 k symbols LRIT / 194 k HRIT, demodulator.cpp:108-119 --
                                 *      comes out as the CPU chain's soft symbols word for word (a 512 Ki-sample call: 5.0 ms
                                 *      instead of 3.2). */
    int32_t  reserved[2];
} xrit_demod_config;

/* setLRITMode / setHRITMode + Parameters.h defaults (demodulator.cpp:177-197) */
void xrit_demod_config_lrit(xrit_demod_config *cfg, float sample_rate, uint32_t decimation);
void xrit_demod_config_hrit(xrit_demod_config *cfg, float sample_rate, uint32_t decimation);

int  xrit_demod_create(const xrit_demod_config *cfg, xrit_demod **out);
void xrit_demod_destroy(xrit_demod *d);

/* One processSamples() pass: onSamplesAvailable conversion (:54-74), decimator
 * (:136-140, n/decimation outputs, remainder dropped), AGC (:143), RRC (:148),
 * Costas (:152), clock recovery (:156); soft_out receives Re(symbol)
 * (SymbolManager.cpp:104).  State persists across calls like the SatHelper
 * objects.  cap must be >= n/(decimation*sps*0.99)+64: a smaller one is refused before anything runs
 * (XRIT_E_CAPACITY, n_out = a capacity that suffices, the handle unchanged).  A call that fails later, after a
 * stage has advanced its carried state, leaves the handle unusable (every further call returns XRIT_E_INVALID).
 * Host buffers. */
int xrit_demod_process(xrit_demod *d, const void *samples, size_t n_complex, int sample_type,
                       float *soft_out, size_t cap, size_t *n_out);
/* Same with device-resident input and output (no PCIe in the call). */
int xrit_demod_process_device(xrit_demod *d, const void *d_samples, size_t n_complex, int sample_type,
                              float *d_soft_out, size_t cap, size_t *n_out, void *stream);
/* The clock recovery of the LAST process call once more, on the sign-flipped Costas output (xrit_group_*: this
 * rank's Costas loop locked pi away from the stream the capture's first GPU follows; the Mueller & Mueller detector
 * slices to {0, 1}, so the loop on -y is not minus the loop on y and the recovery has to run on the right sign).
 * The clock recovery's carried state goes back to what it was before that call (history and unread tail change sign),
 * the Costas loop's carried phase moves by pi; d_soft receives the call's symbols again.
 * Refused (XRIT_E_INVALID, nothing changed) while an input registered with xrit_demod_prefetch_device waits for its
 * process call: its front end and Costas loop may already have started from the unflipped state. */
int xrit_demod_redo_clock_flipped(xrit_demod *d, float *d_soft_out, size_t cap, size_t *n_out, void *stream);
/* Called after the process call that warmed the chain up over a time slice's halo (from a cold start): that call's
 * clock recovery is run once more on the negated Costas output and what it would carry is kept aside, so that a later
 * xrit_demod_redo_clock_flipped starts from the flipped loop's own state (without it, it starts from this sign's state
 * with the symbol history negated -- right to 1e-3 sample in timing, which takes 1e4..1e5 symbols to settle).
 * Refused like xrit_demod_redo_clock_flipped while a prefetched input waits. */
int xrit_demod_prepare_flipped(xrit_demod *d, void *stream);
/* Move the carried Costas phase by pi (a freshly reset chain: 0 -> pi).  The loop's equations do not see a half turn
 * (the BPSK detector I*Q does not), so a chain started so pulls in along the same path into the other of its two locks:
 * how a rank of xrit_group_* that found itself pi away from the stream starts again with the bit-exact front end, to
 * meet the stream's own float32 trajectory instead of its negative-to-rounding.  (The reference starts at 0 and keeps
 * whichever lock it falls into, demodulator.cpp:152.)  Refused while a prefetched input waits. */
int xrit_demod_flip_costas_phase(xrit_demod *d, void *stream);
/* The clock recovery's carried state as one device record of xrit_demod_clock_carry_bytes() bytes (the Mueller & Mueller
 * loop's mu, omega, last symbols and decisions, and the <= 1024 de-rotated samples it has not consumed yet):
 * which = 0: what the NEXT process call starts from, which = 1: what the LAST one started from.  The reference's loop
 * objects carry ONE state across all chunks (demodulator.cpp:446-450, 136-157); across the GPUs of xrit_group_* this record
 * is that state, handed from the rank in front to the rank behind. */
size_t xrit_demod_clock_carry_bytes(void);
int xrit_demod_export_clock_carry(xrit_demod *d, int which, void *d_record, void *stream);
/* The clock recovery of the LAST process call once more from another handle's record (which = 0 of the handle that
 * demodulated the samples in front): same input, that loop state in front of it; d_soft receives the call's symbols again.
 * Refused (XRIT_E_INVALID, nothing changed) while a prefetched input waits, or for a record that is not one. */
int xrit_demod_redo_clock_from(xrit_demod *d, const void *d_record, float *d_soft_out, size_t cap, size_t *n_out, void *stream);
/* 1 if the last process call's symbols are those of ONE float32 walk from the state it started from (cfg.clock_serial,
 * the exact closure cfg.clock_exact = 1, or a call short enough for a single exact walk -- up to 200 k symbols in the
 * default configuration and with cfg.front_exact = 2, 73 k with the fast front end on calls of every size), 0 if they are relayed / overlapping walks (close to it, not it), < 0 on a null handle. */
int xrit_demod_last_clock_exact(const xrit_demod *d);
/* 1 if a process call of n_complex input samples on this handle takes the bit-exact front end (cfg.front_exact and the
 * call's length decide, see xrit_demod_config), 0 if the fast one, < 0 on a null handle. */
int xrit_demod_front_exact_for(const xrit_demod *d, size_t n_complex);
/* Streaming at full rate: register the input of a LATER process call now, so that the library can run it ahead of
 * that call on streams of its own while the calls in between are at work:
 *     prefetch(b); prefetch(b+1);  prefetch(b+2); process_device(b);  prefetch(b+3); process_device(b+1);  ...
 * Inputs are taken by the process calls in the order they were registered (each must pass exactly the registered
 * pointer, count and type) and must stay valid until the call that takes them has returned.  The reference's input FIFO
 * plays the same role (demodulator.cpp:38,54-74: the frontend thread fills it while the DSP thread works).  A no-op while
 * stage copies or full per-kernel profiling are on.  What runs ahead, and how many inputs may wait:
 *   * calls of a million symbols or more in the default configuration (round 5: the clock recovery walks overlapping
 *     blocks, csrc/clock_overlap.h, and waits for no other burst): the front end, the Costas loop AND the clock
 *     recovery's walkers of the next TWO bursts -- three inputs may wait (the call in progress and two behind it).  A
 *     process call enqueues what the newest registration allows, waits for its own walkers and lays out its symbols;
 *   * every other call (smaller calls, clock_exact != 0): the front end and the Costas loop of the NEXT burst, started in
 *     front of the current call's relay kernels (round 4) or at once (clock_exact < 0) -- two inputs may wait.
 * The symbols are those of plain consecutive calls, word for word, either way.  XRIT_E_INVALID when as many inputs as
 * may wait are waiting already. */
int xrit_demod_prefetch_device(xrit_demod *d, const void *d_samples, size_t n_complex, int sample_type, void *stream);
/* How many inputs of n_complex samples may wait BEHIND the call in progress on this handle: 2 (calls that walk overlapping
 * blocks), 1 (every other call), 0 (stage copies or full profiling are on: nothing runs ahead). */
int xrit_demod_prefetch_depth(xrit_demod *d, size_t n_complex);
/* Back to the state right after xrit_demod_create (filter histories, gain, loop states, unread tail), without
 * giving up the device buffers: the start of another stream.  Also revives a handle a failed call left unusable. */
int xrit_demod_reset(xrit_demod *d, void *stream);
/* the hipStream_t the handle runs on when a call passes stream = NULL */
void *xrit_demod_stream(xrit_demod *d);
/* symbols per sample as the reference computes it (demodulator.cpp:437) */
float xrit_demod_sps(const xrit_demod *d);
int   xrit_demod_decimator_ntaps(const xrit_demod *d);

/* Diagnostics: enable=1 makes every later process call keep a copy of each
 * stage's output (costs D2D copies and un-fuses the stages; off by default).
 * enable=2 keeps only stage 4, the complex symbols that DiagManager::addSamples
 * taps (demodulator.cpp:161-163), at no other cost. */
int xrit_demod_keep_stages(xrit_demod *d, int enable);
/* Copies a stage's output of the LAST process call to host (tests/diagnostics):
 * 0 decimator, 1 agc, 2 rrc, 3 costas, 4 clock recovery (complex symbols).
 * Returns the element count through n; out may be NULL to query the count. */
int xrit_demod_read_stage(xrit_demod *d, int stage, float *out_interleaved, size_t cap, size_t *n);

typedef struct xrit_demod_stats {
    uint64_t samples_in;          /* last call */
    uint64_t circuit_samples;
    uint64_t symbols_out;
    int32_t  costas_passes;       /* hand-off passes run, last call */
    int32_t  clock_passes;
    uint32_t costas_unconverged;  /* chain boundaries left above tolerance */
    uint32_t clock_unconverged;   /* boundaries the last clock hand-off solve still moved: close to all of them on any
                                   * healthy call (they keep moving at the recurrence's own 1e-4 floor), NOT a failure
                                   * count -- that is clock_open_large */
    float    costas_max_residual; /* rad */
    float    clock_max_residual;  /* samples */
    int32_t  agc_serial_fallback; /* 1 if the affine scan guard tripped (|x|*rate > 1) */
    uint32_t clock_open_large;    /* boundaries left with a timing residual beyond 0.02 sample or an unresolved symbol
                                   * slip when the passes ended: 0 on a healthy call; non-zero means acquisition did
                                   * not finish within max_passes (symbol count or decisions may be off) */
    int32_t  costas_serial_walk;  /* 1 if the carrier hand-off was still open after 32 passes (pull-in through cycle
                                   * slips closes a chain or two per pass) and the open region was walked by one
                                   * serial wave instead: exact, ~0.1 us per sample of the region, once per acquisition */
    int32_t  clock_relay_passes;  /* clock_exact: relay passes of the last call (the closing one included) */
    int32_t  clock_relay_closed;  /* ... 1 if they ended with a pass that changed nothing: the symbols are the serial
                                   * trajectory's, bit for bit */
    int32_t  clock_relay_segments;/* ... segments the call was cut into */
} xrit_demod_stats;
int xrit_demod_get_stats(const xrit_demod *d, xrit_demod_stats *s);

/* Per-kernel timing with HIP events on the stream the kernels are launched on.
 * enable=1 brackets every launch of the following process calls; enable=2 only the decimating FIR (the
 * launch that reads the input) -- an event record is a queue barrier that stops neighbouring kernels from
 * overlapping, about 0.27 ms per 256 Mi-sample burst when all ~70 launches are bracketed; enable=0: off. */
int xrit_demod_profile(xrit_demod *d, int enable);
/* name/ms arrays are filled with up to cap entries (accumulated since enable);
 * launches[] = number of launches per kernel.  Returns entries written. */
int xrit_demod_profile_read(xrit_demod *d, const char **names, float *total_ms, int *launches, int cap);
/* every bracket of one kernel name since enable, in launch order (min / median of a kernel, not only its mean);
 * returns the number written */
int xrit_demod_profile_samples(xrit_demod *d, const char *name, float *ms, int cap);

/* SymbolManager::process quantiser (SymbolManager.cpp:43-46): f=s*127, clamp
 * [-128,127], C cast (truncation).  Device kernel on device pointers. */
int xrit_quantize_i8_device(const float *d_soft, int8_t *d_out, size_t n, int device, void *stream);
int xrit_quantize_i8(xrit_demod *d, const float *soft, int8_t *out, size_t n);

/* ------------------------------------------------------------------------
 * Decoder front end ("next" row): frame synchronisation.
 * Replaces SatHelper::Correlator as decoder/src/newdecoder.cpp uses it before
 * Viterbi: addWord() of the encoded 64-bit sync words (:145-151; LRIT
 * 0xfca2b63db00d9794 / 0x035d49c24ff2686b, HRIT 0xfc4ef4fd0cc2df89 /
 * 0x25010b02f33d2076, :21-24), correlate(codedData, CODEDFRAMESIZE = 16384)
 * and the three getters (:218-245).  For every consecutive window of `frame`
 * int8 soft symbols: the word with the most agreeing hard bits (first word on a
 * tie), the first position where it reaches that count, and the count
 * (MINCORRELATIONBITS = 46 is the decoder's acceptance, parameters.h:31).
 * word 0 = as sent, word 1 = inverted (PhaseShift::DEG_180, :232). */
typedef struct xrit_sync_hit {
    uint32_t word;
    uint32_t position;
    uint32_t correlation;
    uint32_t reserved;
} xrit_sync_hit;
/* device pointers; hits has n_symbols / frame entries */
int xrit_sync_correlate_device(const int8_t *d_symbols, size_t n_symbols, const uint64_t *words, int nwords,
                               uint32_t frame, xrit_sync_hit *d_hits, int device, void *stream);
/* host buffers (one H2D / D2H round trip) */
int xrit_sync_correlate(const int8_t *symbols, size_t n_symbols, const uint64_t *words, int nwords, uint32_t frame,
                        xrit_sync_hit *hits, int device);

/* Frame alignment and phase fix, the two steps between the correlator and Viterbi
 * (newdecoder.cpp:239-270): for window f the frame that starts at the correlation
 * position, symbols[f*frame + position .. + frame) -- the reference shifts its chunk
 * down and reads `position` more bytes -- with every byte XOR 0xFF when word != 0
 * (PacketFixer::fixPacket(..., DEG_180, false), :232,:265-267; LRIT only: HRIT is
 * differentially coded and skips it, so pass hits with word forced to 0 there).
 * A window whose correlation is below min_correlation (MINCORRELATIONBITS, :239-242)
 * or whose frame would run past n_symbols gives no frame: zeros, valid[f] = 0.
 * frames: (n_symbols / frame) * frame bytes; valid: n_symbols / frame bytes. */
int xrit_sync_fix_frames_device(const int8_t *d_symbols, size_t n_symbols, const xrit_sync_hit *d_hits, uint32_t frame,
                                uint32_t min_correlation, int8_t *d_frames, uint8_t *d_valid, int device, void *stream);
int xrit_sync_fix_frames(const int8_t *symbols, size_t n_symbols, const xrit_sync_hit *hits, uint32_t frame,
                         uint32_t min_correlation, int8_t *frames, uint8_t *valid, int device);

/* ------------------------------------------------------------------------
 * Stage objects -- the SatHelper classes one by one, for stage-level parity
 * and for callers that keep the reference's five-Work() structure.
 * in/out are HOST pointers unless the _device variant is used.
 * ------------------------------------------------------------------------ */
typedef struct xrit_fir     xrit_fir;      /* SatHelper::FirFilter      (demodulator.cpp:446,450) */
typedef struct xrit_agc     xrit_agc;      /* SatHelper::AGC            (:447) */
typedef struct xrit_costas  xrit_costas;   /* SatHelper::CostasLoop     (:448) */
typedef struct xrit_clock   xrit_clock;    /* SatHelper::ClockRecovery  (:449) */
typedef struct xrit_rtl     xrit_rtl;      /* RtlFrontend's byte -> float conversion (RtlFrontend.cpp:102-116) */

/* sample_rate: what RtlFrontend::SetSampleRate received (alpha of the DC tracker, RtlFrontend.cpp:57) */
int  xrit_rtl_create(float sample_rate, int device, xrit_rtl **out);
/* n_complex IQ pairs = 2 n_complex bytes in, 2 n_complex floats out (what the frontend passes to its callback) */
int  xrit_rtl_work(xrit_rtl *r, const uint8_t *data, size_t n_complex, float *out_iq);
void xrit_rtl_destroy(xrit_rtl *r);

int  xrit_fir_create(unsigned decimation, const float *taps, int ntaps, int device, xrit_fir **out);
/* FirFilter::Work(in, out, nOut): consumes nOut*decimation samples */
int  xrit_fir_work(xrit_fir *f, const float *in, float *out, size_t n_out);
/* exact = 1: the dot product summed in the CPU chain's order -- four interleaved float32 partial sums over the time-ordered
 * window, products rounded before they are added (xrit_demod_config.front_exact = 2) */
int  xrit_fir_set_exact(xrit_fir *f, int exact);
void xrit_fir_destroy(xrit_fir *f);

int  xrit_agc_create(float rate, float reference, float gain, float max_gain, int device, xrit_agc **out);
int  xrit_agc_work(xrit_agc *a, const float *in, float *out, size_t n);
/* exact = 1: the float32 recurrence walked literally in warmed-up chains whose joints are checked bit for bit (front_exact = 2) */
int  xrit_agc_set_exact(xrit_agc *a, int exact);
/* the last exact call's counters: [0] joints open after the rounds, [1] 64-sample blocks walked, [2] Picard rounds, [3] blocks at the
 * round limit, [4] lattice segments, [5] scans that fell back to the systolic one, [6..7] unused */
int  xrit_agc_exact_stats(xrit_agc *a, uint32_t *counters8);
float xrit_agc_gain(xrit_agc *a);
void xrit_agc_destroy(xrit_agc *a);

int  xrit_costas_create(float loop_bw, int order, int device, xrit_costas **out);
int  xrit_costas_work(xrit_costas *c, const float *in, float *out, size_t n);
int  xrit_costas_state(xrit_costas *c, float *phase, float *freq);
/* exact = 1: behind the chains' hand-off the output is put on the serial float32 trajectory by exactly walked overlapping
 * ranges (front_exact = 2; csrc/costas_exact.hip).  history: samples of warm-up in front of every range (0 = by plan:
 * none on large calls, where a walker goes on into the next range until the two meet; up to 32768 on small ones) */
int  xrit_costas_set_exact(xrit_costas *c, int exact, int history);
/* totals over the handle's calls: 64-sample blocks walked and Picard rounds spent on them; of the last call: joints that were
 * still open after the rounds enqueued with the call, and the rounds the host added; totals again: segments of the lattice
 * scans and scans that fell back to the systolic one (any pointer may be null) */
int  xrit_costas_exact_stats(xrit_costas *c, uint64_t *blocks, uint64_t *rounds, uint32_t *joints_open, uint32_t *fix_rounds,
                             uint64_t *lattice_segments, uint64_t *lattice_fallbacks);
void xrit_costas_destroy(xrit_costas *c);

/* The sincosf of the exact Costas loop on an array of (host) floats, |x| < 120: the C library's sincosf as the reference's loop
 * calls it (glibc's __sincosf_fma: double-precision reduction and polynomials, fused multiply-adds, one rounding), evaluated
 * operation for operation on the device (csrc/exact_sincos.h; tests hold it against the CPU chain's bit for bit). */
int  xrit_loop_sincosf(const float *x, float *sin_out, float *cos_out, size_t n, int device);

int  xrit_clock_create(float omega, float gain_omega, float mu, float gain_mu, float omega_rel_limit,
                       int device, xrit_clock **out);
/* ClockRecovery::Work(in, out, n) -> symbols through n_out */
int  xrit_clock_work(xrit_clock *c, const float *in, size_t n, float *out, size_t cap, size_t *n_out);
/* serial = 1: one trajectory on a single wave (see xrit_demod_config.clock_serial); 0: time-tiled chains (default) */
int  xrit_clock_set_serial(xrit_clock *c, int serial);
/* exact closure of the tiled evaluation (see xrit_demod_config.clock_exact / clock_exact_window) */
int  xrit_clock_set_exact(xrit_clock *c, int exact, int window);
void xrit_clock_destroy(xrit_clock *c);

/* ------------------------------------------------------------------------
 * Synthetic burst generator -- stands in for the cf32 capture
 * the reference reads through CFileFrontend (CFileFrontend.cpp:34-56).
 * Writes n cf32 samples [start, start+n) into a device buffer.
 * ------------------------------------------------------------------------ */
typedef struct xrit_synth_params {
    double   fs_in, symbol_rate, alpha, amplitude, carrier_hz, phase0, timing_offset, clock_ppm;
    double   esn0_db;     /* < -900: no noise */
    uint64_t seed;
} xrit_synth_params;
void xrit_synth_defaults(xrit_synth_params *p);
int  xrit_synth_generate_device(const xrit_synth_params *p, uint64_t start, size_t n,
                                float *d_out_interleaved, int device, void *stream);

/* ------------------------------------------------------------------------
 * One capture across the GPUs of a node.  The reference runs the
 * chain on one CPU thread (demodulator.cpp:170-175); a burst is cut here in
 * `world` time slices, one per GPU, and three small things cross GPUs -- RCCL
 * ncclSend / ncclRecv over xGMI and one ncclAllGather of two integers per rank:
 * the halo (the last xrit_group_halo_samples() input samples of rank g-1, which
 * rank g demodulates first from a cold start), the last 256 soft symbols of rank
 * g-1 (Costas pi ambiguity, straddling symbol), and (relative polarity, symbol
 * count) of every rank (-> absolute polarity, output offset).  Independent
 * capture segments need none of it: every rank uses xrit_group_chain() as a
 * plain chain handle.
 * ------------------------------------------------------------------------ */
typedef struct xrit_group xrit_group;
#define XRIT_GROUP_ID_BYTES 128
/* One process per GPU: rank 0 makes the id (ncclGetUniqueId), the launcher hands
 * it to every rank (environment, file, MPI, torch.distributed ...), every rank
 * calls xrit_group_create with cfg->device = its GPU (collective call). */
int xrit_group_unique_id(void *id /* [XRIT_GROUP_ID_BYTES] */);
int xrit_group_create(const xrit_demod_config *cfg, int rank, int world, const void *id, xrit_group **out);
/* One process driving `world` GPUs (ncclCommInitAll): out[0..world) are the ranks'
 * handles, to be used from one thread each. */
int xrit_group_create_all(const xrit_demod_config *cfg, const int *devices, int world, xrit_group **out);
/* The same exchange without RCCL: ranks are threads of one process that hand
 * their buffers over through device-to-device copies (hipMemcpyPeer across
 * GPUs).  What a single-GPU box can run: several ranks on one device. */
typedef struct xrit_local_fabric xrit_local_fabric;
int  xrit_local_fabric_create(int world, xrit_local_fabric **out);
void xrit_local_fabric_destroy(xrit_local_fabric *f);
int  xrit_group_create_local(const xrit_demod_config *cfg, int rank, xrit_local_fabric *fabric, xrit_group **out);
void xrit_group_destroy(xrit_group *g);
xrit_demod *xrit_group_chain(xrit_group *g);
int    xrit_group_rank(const xrit_group *g);
int    xrit_group_world(const xrit_group *g);
size_t xrit_group_halo_samples(const xrit_group *g);
/* What this rank did at its slice boundaries so far (see xrit_group_process_slice_device): slices started a second time
 * from the other Costas lock; slices whose clock recovery ran again from the loop state of the rank in front; slices
 * that had met that state bit for bit inside their halo (nothing to run again).  Any pointer may be null. */
void xrit_group_counters(const xrit_group *g, uint64_t *relocks, uint64_t *handovers, uint64_t *joined);
/* Ranks of the RCCL communicator the group exchanges over, as RCCL itself counts them (ncclCommCount); 0 when the
 * ranks are threads of one process (xrit_group_create_local: no communicator). */
int    xrit_group_rccl_ranks(xrit_group *g);
/* Collective: every rank passes ITS slice (n samples, device resident, slices in
 * rank order make up the burst; n >= the halo and a whole number of decimation
 * periods) and receives its symbols in the stream's
 * polarity with their offset in the burst's symbol sequence: rank r's symbols are
 * out[offset .. offset + n_out) of what one chain would emit for the whole burst
 * (a rank whose Costas loop locked pi away from the stream's starts once more
 * from a phase of pi where its front end is the bit-exact one -- it then meets the
 * stream's own trajectory inside the halo, like a rank that fell on the right side
 * at once: symbols that are the single chain's word for word wherever that chain's
 * are the CPU chain's --, and with the fast front end runs its clock recovery once
 * more on the sign-flipped stream, so both locks end at the same floor).  Consecutive calls are consecutive bursts of ONE
 * capture: the last rank keeps the end of its slice and hands it to rank 0 at the
 * start of the next call (the exchanges become a ring), so rank 0 warms up over a
 * halo like every other rank; xrit_group_restart() begins a new capture (as does
 * a failed call).  Every slice must be a whole number of decimation periods.
 * A failure on one rank (capacity, a stage that did not converge in strict mode,
 * HIP) is carried to every rank in the all-gather: every rank returns an error
 * from the same call and nobody is left waiting in an exchange; a rank that
 * cannot even serve its peers (out of device memory) aborts the communicator. */
int xrit_group_process_slice_device(xrit_group *g, const void *d_samples, size_t n_complex, int sample_type,
                                    float *d_soft_out, size_t cap, size_t *n_out, uint64_t *offset_out,
                                    int *polarity_out, void *stream);
/* the same with host buffers (one H2D / D2H round trip per call, like xrit_demod_process) */
int xrit_group_process_slice_host(xrit_group *g, const void *samples, size_t n_complex, int sample_type, float *soft_out,
                                  size_t cap, size_t *n_out, uint64_t *offset_out, int *polarity_out);
/* max over the ranks (timing: the slowest rank's seconds) */
int xrit_group_restart(xrit_group *g);   /* the next slice call is a capture's first (every rank calls it) */
int xrit_group_allreduce_max(xrit_group *g, double *value, void *stream);

/* ------------------------------------------------------------------------
 * Measurement helper ("also measure a device read microbenchmark
 * on the box and report both"): GB/s of a hand-written read-only sweep over a
 * device buffer, `reps` sweeps timed with HIP events on `stream`.  Not part of
 * the reference's interface.
 * ------------------------------------------------------------------------ */
int xrit_device_read_bandwidth(const void *d_buf, size_t bytes, int reps, int device, void *stream, double *gb_per_s);

#ifdef __cplusplus
}
#endif
#endif /* XRITDEMOD_AMD_H_ */
