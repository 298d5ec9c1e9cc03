/* asdr_front_oracle.h -- CPU restatement of the three AudioStream blocks AROUND the AudioSDR hot path
 * (SURVEY.md 8(f) rows 2-4): AudioSDRpreProcessor, AudioIQgenerator, AudioGrabberComplex256.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import,
 * call, link or execute anything under oracle/; nothing under audiosdr_amd/ or include/ does.
 *
 * PARITY UNPINNED as a whole, as for asdr_oracle.h: the reference is Teensy/Arduino C++ and cannot be built or run in this
 * image (no Teensy core, CMSIS-DSP only as ARM binaries there), and it ships no golden vectors.  Each function
 * below cites the reference file:line it restates.  The two CMSIS-DSP functions the pre-processor's detector calls --
 * `arm_cfft_f32(&arm_cfft_sR_f32_len128, buf, 0, 1)` and `arm_cmplx_mag_squared_f32` (AudioSDRpreProcessor.cpp:93-94) -- have no
 * source in /root/reference, only Cortex-M4 objects inside libarm_cortexM4lf_math.a.  Until round 3 the detector here used this
 * project's own radix-2 FFT (same transform, 1e-7 rounding differences: tolerance-only parity for the power spectrum).  From round
 * 4 ao_fft128 restates the published CMSIS algorithm (radix-8-by-2 decimation in frequency + digit reversal) operation for operation
 * and is PINNED AGAINST THE REFERENCE'S BINARY: tests/thumb_emu.py links and executes the objects, tests/test_cmsis_object.py holds
 * ao_fft128 equal to them bit for bit.  Everything else (skew correction incl. the -1 branch's quirk, the summation ORDER of the line
 * powers, thresholds, counters, swap; the IQ generator; the grabber) is restated operation for operation from the reference's source.
 */
#ifndef ASDR_FRONT_ORACLE_H_
#define ASDR_FRONT_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- AudioSDRpreProcessor (AudioSDRpreProcessor.h:49-84, .cpp:46-169) ---- */
typedef struct ao_pre ao_pre_t;
ao_pre_t *ao_pre_create(void);
void ao_pre_destroy(ao_pre_t *p);
void ao_pre_update(ao_pre_t *p, int16_t *blockI, int16_t *blockQ);        /* in place, like the reference (.cpp:46-138) */
void ao_pre_startAutoI2SerrorDetection(ao_pre_t *p);                       /* .cpp:141-147 */
void ao_pre_stopAutoI2SerrorDetection(ao_pre_t *p);                        /* .cpp:150-153 */
int ao_pre_getAutoI2SerrorDetectionStatus(const ao_pre_t *p);              /* .cpp:157 */
void ao_pre_setI2SerrorCompensation(ao_pre_t *p, int correction);          /* .cpp:160-163 */
int16_t ao_pre_getI2SerrorCompensation(const ao_pre_t *p);                 /* .cpp:166 */
void ao_pre_swapIQ(ao_pre_t *p, int swap);                                 /* .cpp:169 */
/* internals for the tests: counters and the detector's last measurements */
typedef struct {
  int16_t correction, saved_sample, failure_count, success_count;
  int32_t auto_detect, swap;
  int32_t max_line, strong;          /* last detector pass: strongest line, and whether it cleared the floor */
  float max_power, avg_power, ratio; /* last detector pass */
} ao_pre_state_t;
void ao_pre_get_state(const ao_pre_t *p, ao_pre_state_t *s);
const float *ao_pre_power_spectrum(const ao_pre_t *p);                     /* 128 line powers of the last detector pass */

/* The reference's 128-point complex float32 FFT -- CMSIS-DSP arm_cfft_f32, len 128, forward, bit-reversed to natural order -- in place
 * on interleaved re/im (asdr_front_oracle.c spells the algorithm out): radix-2 DIF pass with twiddleCoef_128 over the quarters
 * (q, q + 32, q + 64, q + 96), two 64-point radix-8 transforms (stage 1: column 0 twiddle-free, columns 1..7 with twiddles; stage 2:
 * eight twiddle-free butterflies on consecutive points), digit reversal (line 16 c + 2 b + a <- position 64 a + 8 b + c).  Every
 * product and sum separately rounded.  Butterflies of one stage are independent, so any schedule gives the same bits. */
void ao_fft128(float *buf /* [256] */);

/* ---- AudioIQgenerator (AudioIQgenerator.h:48-106, .cpp:33-87) ---- */
typedef struct ao_iqgen ao_iqgen_t;
ao_iqgen_t *ao_iqgen_create(void);
void ao_iqgen_destroy(ao_iqgen_t *g);
void ao_iqgen_update(ao_iqgen_t *g, const int16_t *in, int16_t *outI, int16_t *outQ);   /* .cpp:33-87 */
void ao_iqgen_setGainBalance(ao_iqgen_t *g, float balance);                               /* .h:55-59 */
const float *ao_iqgen_hilbert_taps(void);

/* ---- AudioGrabberComplex256 (AudioGrabberComplex256.h:44-63, .cpp:39-90) ---- */
typedef struct ao_grab ao_grab_t;
ao_grab_t *ao_grab_create(void);
void ao_grab_destroy(ao_grab_t *g);
void ao_grab_update(ao_grab_t *g, const int16_t *blockI, const int16_t *blockQ);          /* .cpp:50-72 */
int ao_grab_newDataAvailable(const ao_grab_t *g);                                          /* .cpp:75-77 */
void ao_grab_grab(ao_grab_t *g, int16_t *destination /* [512] */);                         /* .cpp:80-90 */

/* Panadapter spectrum of one grabber buffer (NOT part of the reference library -- its example sketches run an FFT on the
 * grabbed samples in application code; SURVEY.md 8(f) row 4 asks for a device-side version): x[n] = (re + j*im) / 32768
 * (exact in float32), this project's 256-point FFT (ao_fft256: same radix-2 DIT arithmetic as ao_fft128 with
 * asdr_fft256_tw), power[k] = Re^2 + Im^2, k = 0..255 in natural order (k >= 128 are the negative frequencies). */
void ao_fft256(float *buf /* [512] */);
void ao_grab_power_spectrum(const int16_t *buffer /* [512] interleaved re, im */, float *power /* [256] */);

/* exhaustive: (float)((double)s / 32767.0) == (float)(Markstein reciprocal form) for all int16 s (count of mismatches) */
int ao_front_check_div32767(void);

#ifdef __cplusplus
}
#endif
#endif /* ASDR_FRONT_ORACLE_H_ */
