/* asdr_front_oracle.h -- CPU restatement of the three AudioStream blocks AROUND the AudioSDR hot path
 * (SURVEY.md 8(f) rows 2-4): AudioSDRpreProcessor, AudioIQgenerator, AudioGrabberComplex256.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import,
 * call, link or execute anything under oracle/; nothing under audiosdr_amd/ or include/ does.
 *
 * PARITY UNPINNED, as for asdr_oracle.h: the reference is Teensy/Arduino C++ and cannot be built or run in this
 * image (no Teensy core, CMSIS-DSP only as ARM binaries there), and it ships no golden vectors.  Each function
 * below cites the reference file:line it restates.  One part cannot even be restated operation-for-operation:
 * the pre-processor's detector calls CMSIS-DSP `arm_cfft_f32(&arm_cfft_sR_f32_len128, buf, 0, 1)` (CMSIS-DSP as
 * bundled with the Teensy core, arm_math.h of CMSIS 4.5; a radix-8-by-2 decimation-in-frequency transform followed
 * by bit reversal), whose SOURCE is absent from /root/reference.  The detector here therefore uses this project's
 * own 128-point float32 FFT (ao_fft128, a radix-2 decimation-in-time transform whose arithmetic is defined below
 * and implemented identically on the GPU): same mathematical transform, rounding differs at the 1e-7 level, so the
 * power spectrum agrees with the reference's to float32 tolerance and the vote counters agree except for blocks
 * whose power ratios sit within that tolerance of a threshold.  Everything else (skew correction incl. the -1
 * branch's quirk, the summation ORDER of the line powers, thresholds, counters, swap; the IQ generator; the
 * grabber) is restated operation for operation.
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

/* this project's 128-point complex float32 FFT, in place on interleaved re/im:
 *   X[bitrev7(n)] = x[n];  for s = 1..7 (m = 2^s, h = m/2): for every k = 0, m, 2m, ... and j = 0..h-1:
 *     w = asdr_fft128_tw[j * (128/m)];  u = X[k+j];  v = X[k+j+h];
 *     t.re = w.re*v.re - w.im*v.im;  t.im = w.re*v.im + w.im*v.re;     (each product and the sum rounded: no FMA)
 *     X[k+j] = u + t;  X[k+j+h] = u - t;
 * Butterflies of one stage are independent, so any schedule gives the same bits. */
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
