/*
 * xrit_oracle.h -- CPU restatement of the xritdemod BPSK demodulation chain.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it,
 * and only as the checker / the timed CPU baseline.
 *
 * PARITY UNPINNED.  The arithmetic of this path lives in the third-party library
 * libSatHelper (github.com/opensatelliteproject/libsathelper, fetched un-pinned by
 * /root/reference/Makefile:52-55, absent from /root/reference and from this
 * image), and the reference ships no tests or golden vectors (Makefile:91-92).
 * This file restates the published algorithms those classes implement (the GNU
 * Radio 3.7 blocks the in-repo flowgraph demod_tcp_qt.py:95-96,261-266,275-276
 * wires with the same parameter lists) and anchors on the reference's own call
 * sites: demodulator/src/demodulator.cpp:54-74 (ingest), :100-168 (stage
 * order), :436-450 (construction), Parameters.h:16-37 (constants),
 * SymbolManager.cpp:43-46,104 (soft-symbol selection and int8 quantiser).
 */
#ifndef XRIT_ORACLE_H_
#define XRIT_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float re, im; } xo_cf;

/* Sample types, FrontendDevice.h:11-13 */
#define XO_SAMPLE_FLOATIQ 0
#define XO_SAMPLE_S16IQ   1
#define XO_SAMPLE_S8IQ    2
#define XO_SAMPLE_U8IQ    3   /* raw RTL-SDR bytes: converted as RtlFrontend.cpp:102-116 does before its callback */

#define XO_MM_NTAPS  8
#define XO_MM_NSTEPS 128
#define XO_MM_FUDGE  16

/* ---- knobs for the semantics that cannot be checked against upstream ----
 * libSatHelper is absent (PARITY UNPINNED above), so a handful of details of its blocks are design choices of
 * this restatement, not facts.  Each has a knob; the defaults are the GNU Radio 3.7 behaviour the in-repo
 * flowgraph wires (demodulator/demod_tcp_qt.py).  Objects read the knobs when they are created.
 * tests/test_oracle_kat.py runs the implementation-independent known-answer tests over the whole knob matrix and
 * records how far each knob moves the soft symbols. */
typedef struct {
    int fir_phase_last;   /* FirFilter with decimation D.  0: output m = sum h[k] x[m D - k] (taken at the FIRST sample
                           * of each group of D, GNU Radio fir_filter_ccf).  1: sum h[k] x[m D + D-1 - k] (at the
                           * last sample of the group: a loop that filters after collecting D samples). */
    int mm_fudge;         /* ClockRecovery: samples held back beyond the 8 interpolator taps (GNU Radio: 16); 0..64 */
    int mm_drop_tail;     /* 0: samples a call did not read are carried to the next call (chunk invariant, what
                           * GNU Radio's scheduler does through consume_each).  1: every call starts reading at its
                           * own first sample, the unread tail of the previous call is lost (a Work() that keeps only
                           * mu / omega / the last symbols between calls). */
    int costas_wrap_pi;   /* 0: phase wrapped by while-loops at +-2 pi (GNU Radio control_loop::phase_wrap).
                           * 1: wrapped into (-pi, pi] every sample. */
    int costas_imag_axis; /* SymbolManager.cpp:104: "old was imaginary due bug in libSatHelper costas loop".
                           * 0: the loop locks with the data on the real axis (error = re * im) and the chain keeps
                           * Re(symbol).  1: the old behaviour as far as that comment tells: data on the imaginary
                           * axis (error = -re * im), the chain keeps Im(symbol). */
} xo_knobs;
void xo_knobs_default(xo_knobs *k);
void xo_set_knobs(const xo_knobs *k);   /* NULL: defaults */
void xo_get_knobs(xo_knobs *k);

/* ---- tap designers (run once at start-up, demodulator.cpp:443-444) ---- */
/* Filters::lowPass(gain, Fs, cutoff, transitionWidth, HAMMING, beta): returns
 * the tap count (GNU Radio firdes length rule); writes at most cap taps. */
int  xo_lowpass_ntaps(double fs, double transition_width);
int  xo_lowpass_taps(double gain, double fs, double cutoff, double transition_width,
                     float *taps, int cap);
/* Filters::RRC(gain, Fs, symbolRate, alpha, ntaps): ntaps is forced odd; returns it. */
int  xo_rrc_taps(double gain, double fs, double symbol_rate, double alpha, int ntaps,
                 float *taps, int cap);
/* 8-tap, 128-step MMSE fractional interpolator table (129 rows x 8), one-sided
 * design bandwidth 0.25 cycles/sample.  Row layout follows GNU Radio's
 * interpolator_taps.h: column c multiplies the sample 3-(7-c) ... see .c file. */
void xo_mmse_table(float *table /* [129*8] */);

/* ---- FirFilter(decimation, taps): y[m] = sum_k h[k] x[m*D - k] ---- */
typedef struct xo_fir xo_fir;
xo_fir *xo_fir_create(unsigned decimation, const float *taps, int ntaps);
void    xo_fir_destroy(xo_fir *f);
/* Work(in, out, nOut): consumes nOut*D input samples (demodulator.cpp:137-138) */
void    xo_fir_work(xo_fir *f, const xo_cf *in, xo_cf *out, int n_out);

/* ---- AGC(rate, reference, gain, maxGain) ---- */
typedef struct { float rate, reference, gain, max_gain; } xo_agc;
void xo_agc_init(xo_agc *a, float rate, float reference, float gain, float max_gain);
void xo_agc_work(xo_agc *a, const xo_cf *in, xo_cf *out, int n);

/* ---- CostasLoop(loopBw, order=2) ---- */
typedef struct { float phase, freq, alpha, beta, max_freq, min_freq; int wrap_pi, imag_axis; } xo_costas;
void xo_costas_init(xo_costas *c, float loop_bw);
void xo_costas_work(xo_costas *c, const xo_cf *in, xo_cf *out, int n);
/* the loop's sincosf: glibc 2.35's __sincosf_fma restated operation for operation (xrit_oracle.c) */
void xo_sincosf(float y, float *sinp, float *cosp);

/* ---- ClockRecovery(omega, gainOmega, mu, gainMu, omegaRelativeLimit) ---- */
typedef struct xo_mm xo_mm;
typedef struct {
    float mu, omega, omega_mid, omega_lim, gain_omega, gain_mu;
    xo_cf p_2t, p_1t, p_0t, c_2t, c_1t, c_0t;
    int   carry; /* unread samples held for the next call */
} xo_mm_state;
xo_mm *xo_mm_create(float omega, float gain_omega, float mu, float gain_mu, float omega_rel_limit);
void   xo_mm_destroy(xo_mm *m);
/* returns symbols written; out must hold at least n/ (omega*(1-lim)) + 2 */
int    xo_mm_work(xo_mm *m, const xo_cf *in, int n, xo_cf *out);
void   xo_mm_get_state(const xo_mm *m, xo_mm_state *s);
/* the carried state out of one object and into another (tests/dist_twin.py: one loop state across the ranks of a cut stream) */
void   xo_mm_export(const xo_mm *m, xo_mm_state *s, xo_cf *carry /* [>= 2048] */);
void   xo_mm_import(xo_mm *m, const xo_mm_state *s, const xo_cf *carry);
/* per-symbol trace for diagnostics (tests only): arm index of the last call */
int    xo_mm_work_trace(xo_mm *m, const xo_cf *in, int n, xo_cf *out, int *arm, float *mu_trace);

/* ---- the chain, demodulator.cpp:100-168 ---- */
typedef struct {
    float    sample_rate;      /* device->GetSampleRate() */
    uint32_t decimation;       /* baseDecimation, cfg "decimation" */
    uint32_t symbol_rate;      /* LRIT 293883 / HRIT 927000, Parameters.h:18,23 */
    float    rrc_alpha;        /* Parameters.h:19,24 */
    int      rrc_taps;         /* RRC_TAPS 63 */
    float    agc_rate, agc_reference, agc_gain, agc_max_gain; /* Parameters.h:34-37 */
    float    pll_alpha;        /* = CLOCK_ALPHA, demodulator.cpp:220 */
    float    clock_mu, clock_alpha, clock_gain_omega, clock_omega_limit; /* Parameters.h:30-33 */
} xo_config;

void xo_config_lrit(xo_config *c, float sample_rate, uint32_t decimation);
void xo_config_hrit(xo_config *c, float sample_rate, uint32_t decimation);

typedef struct xo_demod xo_demod;
xo_demod *xo_demod_create(const xo_config *cfg);
void      xo_demod_destroy(xo_demod *d);
/* One processSamples() pass over n complex samples of the given type.
 * soft_out receives the real parts of the recovered symbols
 * (SymbolManager.cpp:104); returns the symbol count.  cap_out is checked. */
int       xo_demod_process(xo_demod *d, const void *samples, int n, int sample_type,
                           float *soft_out, int cap_out);
/* Stage taps for tests / the staged-parity checks.  stage: 0 decimator out,
 * 1 agc out, 2 rrc out, 3 costas out, 4 clock-recovery out (complex).
 * Valid until the next process() call. */
const xo_cf *xo_demod_stage(const xo_demod *d, int stage, int *n);
int       xo_demod_decimator_ntaps(const xo_demod *d);
const float *xo_demod_decimator_taps(const xo_demod *d);
const float *xo_demod_rrc_taps(const xo_demod *d);
float     xo_demod_sps(const xo_demod *d);
xo_costas *xo_demod_costas(xo_demod *d);   /* the chain's own loop objects (tests) */
xo_mm     *xo_demod_mm(xo_demod *d);

/* SymbolManager::process quantiser, SymbolManager.cpp:43-46 */
void xo_quantize_i8(const float *in, int8_t *out, size_t n);

/* SatHelper::Correlator as the decoder uses it (decoder/src/newdecoder.cpp:145-151,218-245) */
void xo_sync_correlate(const int8_t *data, uint32_t length, const uint64_t *words, int nwords,
                       uint32_t *word_out, uint32_t *pos_out, uint32_t *corr_out);
/* frame alignment + phase fix between correlator and Viterbi (decoder/src/newdecoder.cpp:239-270) */
void xo_sync_fix_frames(const int8_t *data, size_t n, const uint32_t *word, const uint32_t *pos, const uint32_t *corr,
                        uint32_t frame, uint32_t min_corr, int8_t *frames, uint8_t *valid);
/* RtlFrontend::internalCallback, RtlFrontend.cpp:102-116 (table :26-28, alpha :57): bytes -> floats with the
 * frontend's running-average DC tracker; `length` bytes (2 per IQ pair), `length` floats out.  Literal, including
 * the never-taken `if (i % 1)` that leaves ONE average for I and Q. */
typedef struct { float lut[256]; float alpha, iavg, qavg; } xo_rtl;
void xo_rtl_init(xo_rtl *r, float sample_rate);
void xo_rtl_work(xo_rtl *r, const uint8_t *data, unsigned int length, float *iq);
/* ingest conversion, demodulator.cpp:54-74 */
void xo_convert_samples(const void *in, int sample_type, xo_cf *out, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* XRIT_ORACLE_H_ */
