/* asdr_oracle.h -- CPU ORACLE for the AudioSDR update() demodulation chain.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may include, link or call anything in oracle/.  The
 * product (audiosdr_amd/csrc, libasdr_hip.so) never does, and has no CPU fallback.
 *
 * PARITY UNPINNED.  The reference (DerekRowell/AudioSDR @ v1.10) ships no tests, no golden
 * vectors and no fixtures for this path, and it cannot be compiled in this image without
 * writing stand-ins for headers/libraries that are absent (Teensy core AudioStream.h /
 * Arduino.h, and the CMSIS-DSP biquad which is shipped only as Cortex-M4 .a archives), which
 * the build rules forbid.  This file is therefore a careful, line-cited RESTATEMENT of
 * /root/reference/SRC/AudioSDRlib/AudioSDR.{h,cpp}; what pins it is listed in DESIGN.md
 * ("What pins the oracle"): table data checked literal-for-literal against the reference
 * header, documented-behaviour known answers, and independent float64 cross-checks.
 *
 * One asdr_oracle_t == one reference `AudioSDR` instance (the reference's function-static
 * buffers, AudioSDR.cpp:41-44 and :690-694, are per-instance here: SURVEY.md 8a-Q1).
 *
 * Short cites: ".cpp" = SRC/AudioSDRlib/AudioSDR.cpp, ".h" = SRC/AudioSDRlib/AudioSDR.h.
 */
#ifndef ASDR_ORACLE_H_
#define ASDR_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AO_BLOCK 128 /* .h:73 n_block; AUDIO_BLOCK_SAMPLES of the Teensy core */

/* demodulation modes, .h:44-50 */
enum { AO_LSB = 0, AO_USB = 1, AO_CW_LSB = 2, AO_CW_USB = 3, AO_AM = 4, AO_SAM = 5, AO_WSPR = 6 };
/* audio filters, .h:56-66 */
enum { AO_AUDIO_AM = 0, AO_AUDIO_CW = 1, AO_AUDIO_WSPR = 2, AO_AUDIO_2100 = 3, AO_AUDIO_2300 = 4,
       AO_AUDIO_2500 = 5, AO_AUDIO_2700 = 6, AO_AUDIO_2900 = 7, AO_AUDIO_3100 = 8, AO_AUDIO_3300 = 9,
       AO_AUDIO_BYPASS = 10 };
/* AGC presets, .h:68-71 */
enum { AO_AGC_OFF = 0, AO_AGC_FAST = 1, AO_AGC_MEDIUM = 2, AO_AGC_SLOW = 3 };

/* stage taps recorded by ao_update() when taps are enabled (test/debug aid) */
enum {
  AO_TAP_SCALED_I = 0, AO_TAP_SCALED_Q, /* after input scaling      .cpp:67-70  */
  AO_TAP_NB_I, AO_TAP_NB_Q,             /* after the noise blanker  .cpp:73     */
  AO_TAP_IF_I, AO_TAP_IF_Q,             /* after the IF band-pass   .cpp:77-78  */
  AO_TAP_MIX_I, AO_TAP_MIX_Q,           /* after the SSB/AM shifter + (AM) image filter / PLL rotation */
  AO_TAP_DEMOD,                         /* _audioOut after demodulation .cpp:115-143 */
  AO_TAP_AUDIO_FILT,                    /* after audioFilter        .cpp:149    */
  AO_TAP_AGC,                           /* after agcProcessor       .cpp:152    */
  AO_TAP_ALS,                           /* after ALSfilter          .cpp:155    */
  AO_N_TAPS
};

typedef struct asdr_oracle asdr_oracle_t;

/* lifetime: ao_create() == `static AudioSDR sdr;` (zeroed storage, in-class initialisers, init()) */
asdr_oracle_t *ao_create(void);
void ao_destroy(asdr_oracle_t *o);
void ao_enable_taps(asdr_oracle_t *o, int on);
const float *ao_tap(const asdr_oracle_t *o, int tap); /* 128 floats */

/* the hot path: one 128-sample block, I/Q in, mono audio out (.cpp:39-168) */
void ao_update(asdr_oracle_t *o, const int16_t *blockI, const int16_t *blockQ, int16_t *out);

/* ---- control surface, same names as the reference class (.h:88-156) ---- */
void ao_init(asdr_oracle_t *o);
void ao_setMute(asdr_oracle_t *o, int muted);
int ao_getMute(const asdr_oracle_t *o);
void ao_setInputGain(asdr_oracle_t *o, float g);
void ao_setIQgainBalance(asdr_oracle_t *o, float balance);
void ao_setOutputGain(asdr_oracle_t *o, float g);
float ao_setDemodMode(asdr_oracle_t *o, int mode);
int16_t ao_getDemodMode(const asdr_oracle_t *o);
float ao_getTuningOffset(const asdr_oracle_t *o);
float ao_getBPFlower(const asdr_oracle_t *o);
float ao_getBPFupper(const asdr_oracle_t *o);

void ao_enableAudioFilter(asdr_oracle_t *o);
void ao_disableAudioFilter(asdr_oracle_t *o);
void ao_setAudioFilter(asdr_oracle_t *o, int filter);
int ao_getAudioFilter(const asdr_oracle_t *o);

void ao_enableALSfilter(asdr_oracle_t *o);
void ao_disableALSfilter(asdr_oracle_t *o);
void ao_setALSfilterNotch(asdr_oracle_t *o);
void ao_setALSfilterPeak(asdr_oracle_t *o);
void ao_setALSfilterAdaptive(asdr_oracle_t *o);
void ao_setALSfilterStatic(asdr_oracle_t *o);
void ao_setALSfilterParams(asdr_oracle_t *o, unsigned int m, float lambda, float delay);
int ao_ALSfilterIsEnabled(const asdr_oracle_t *o);
int ao_ALSfilterIsNotch(const asdr_oracle_t *o);
int ao_ALSfilterIsPeak(const asdr_oracle_t *o);
int ao_ALSfilterIsAdaptive(const asdr_oracle_t *o);

void ao_enableAGC(asdr_oracle_t *o);
void ao_disableAGC(asdr_oracle_t *o);
int ao_AGCisEnabled(const asdr_oracle_t *o);
int ao_AGCisActive(const asdr_oracle_t *o);
void ao_setAGCthreshold(asdr_oracle_t *o, float v);
void ao_setAGCslope(asdr_oracle_t *o, float v);
void ao_setAGCmode(asdr_oracle_t *o, int mode);
void ao_setAGCkneeWidth(asdr_oracle_t *o, float v);
void ao_setAGCattackTime(asdr_oracle_t *o, float ms);
void ao_setAGCreleaseTime(asdr_oracle_t *o, float ms);
void ao_setAGChangTime(asdr_oracle_t *o, float ms);
void ao_setAGCstaticGain(asdr_oracle_t *o, float g);
float ao_getAGCthreshold(const asdr_oracle_t *o);
float ao_getAGCslope(const asdr_oracle_t *o);
float ao_getAGCkneeWidth(const asdr_oracle_t *o);
float ao_getAGCattack(const asdr_oracle_t *o);
float ao_getAGCrelease(const asdr_oracle_t *o);
float ao_getAAGalphaAttack(const asdr_oracle_t *o); /* sic, .h:137 */
float ao_getAGCbetaAttack(const asdr_oracle_t *o);
float ao_getAGCalphaRelease(const asdr_oracle_t *o);
float ao_getAGCbetaRelease(const asdr_oracle_t *o);
float ao_getAGClookup(const asdr_oracle_t *o, int i);
float ao_getAGCstaticGain(const asdr_oracle_t *o);
float ao_getAMcarrierLevel(const asdr_oracle_t *o);
uint32_t ao_getAGChangCount(const asdr_oracle_t *o); /* not in the reference API; test aid */

void ao_enableNoiseBlanker(asdr_oracle_t *o);
void ao_disableNoiseBlanker(asdr_oracle_t *o);
void ao_setNoiseBlankerThreshold(asdr_oracle_t *o, float ratio);
void ao_setNoiseBlankerThresholdDb(asdr_oracle_t *o, float db);
int ao_NoiseBlankerisEnabled(const asdr_oracle_t *o);
int ao_NoiseBlankerDetection(const asdr_oracle_t *o);

float ao_getSAMfrequency(const asdr_oracle_t *o);
int ao_getSAMphaseLockStatus(const asdr_oracle_t *o);

/* ---- stage-level functions (known-answer tests, and kernel unit parity) ---- */
float ao_sin_f32(float phase);                 /* .h:358-370 */
float ao_cos_f32(float phase);                 /* .h:375-377 */
uint16_t ao_sin_index(float phase);            /* the uint16 intPhase of .h:364 */
float ao_sin_from_index(uint16_t int_phase);   /* .h:365-369 */
float ao_approx_atan2_f32(float y, float x);   /* .h:384-408 */
float ao_fast_sqrt_f32(float x, int n_iter);   /* .h:434-446 */
float ao_log2_approx_f32(float x);             /* .h:483-491 */
/* CMSIS-DSP V1.4.5 arm_biquad_cascade_df1_f32 restated (arm_math.h:1360-1378) */
void ao_biquad_cascade_df1(const float *coefs, float *state, int n_stages,
                           const float *src, float *dst, int n);
float ao_freq_shifter(float *I, float *Q, float freq_shift, float phase0); /* .h:508-526 */
double ao_scale_sample(int16_t s, float gain);                             /* .cpp:68 before the store */
int32_t ao_f64_to_i32(double v);   /* (int)double with the ARM target's saturating semantics (.cpp:160) */
float ao_agc_static_compressor(const float *table130, uint16_t input);     /* .cpp:483-494 */
const float *ao_hilbert_taps(void);   /* 64 floats */
const float *ao_sine_table(void);     /* 257 floats */
const float *ao_biquad_table(int pool_index); /* 20 floats, index as in asdr_tables.h */

/* ---- exhaustive checks of exact-arithmetic shortcuts used by the HIP kernels (see asdr_kernels.hip) */
uint64_t ao_check_sin_index_division(uint32_t bits_lo, uint32_t bits_hi);
uint64_t ao_check_sin_index_one_multiply(uint32_t bits_lo, uint32_t bits_hi);
int ao_check_scale_division(void);
int ao_check_scale_unit_gain(void);
int ao_check_sin_interp_f32(void);
uint64_t ao_check_pll_phase_update(uint32_t seed, uint64_t n);   /* asdr_kernels.hip: the PLL phase update as one fmaf */

/* ---- CPU-baseline helper for bench.py: run `n_channels` independent default-constructed
 * instances configured by `config` (0 = C2 USB chain, see oracle source) over `n_blocks`
 * blocks of caller-provided I/Q (layout [channel][block][128]); returns seconds of wall time
 * spent inside update() only.  `n_threads` host threads over disjoint channel ranges. */
double ao_bench_run(int config, int n_channels, int n_blocks, const int16_t *I, const int16_t *Q,
                    int16_t *out, int n_threads);

#ifdef __cplusplus
}
#endif
/* Unknown demodulation-mode values (anything but 0..6; reachable through setDemodMode, AudioSDR.cpp:188): the reference
 * leaves _audioOut untouched, so the audio filter / AGC / ALS / output stage re-process the PREVIOUS block's already
 * processed audio (AudioSDR.cpp:84,122,149-161).  That is the oracle's default -- and the HIP product's (it keeps every block's
 * post-ALS row in HBM).  on = 1 models the product's OPT-OUT, asdr_set_exact_unknown_mode(b, 0): the post-processing of a silent block. */
void ao_set_unknown_mode_silence(asdr_oracle_t *o, int on);
/* SAM PLL phase wrap (AudioSDR.cpp:735-736: two unbounded loops).  Default 0 = the reference's loops, run to their end; where they
 * could never end (phase_est -+ twoPI == phase_est) the oracle leaves them and sets the flag ao_pll_stalled() reports (the
 * reference would hang there).  on = 1 models the HIP product's defined difference: at most 64 turns per sample, then the estimate
 * restarts at 0.  The GPU parity tests switch it on (tests/conftest.py); it never acts on a finite loop-filter step. */
void ao_set_pll_wrap_bound(asdr_oracle_t *o, int on);
int ao_pll_stalled(const asdr_oracle_t *o);
void ao_test_set_pll_phase(asdr_oracle_t *o, float phase_est);   /* test hooks: the PLL's phase estimate */
float ao_test_get_pll_phase(const asdr_oracle_t *o);

/* same 12 derived constants as the product's asdr_get_chain_constants (include/asdr.h) */
void ao_get_chain_constants(const asdr_oracle_t *o, float out[12]);
/* bench.py host calibration: seconds for `iters` iterations of a register-only float loop on each of n_threads threads */
double ao_spin_calibrate(int n_threads, long iters);

#endif /* ASDR_ORACLE_H_ */
