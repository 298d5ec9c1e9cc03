/* asdr_front.h -- C ABI of the three AudioStream blocks around the AudioSDR hot path, batched on one MI355X
 * (SURVEY.md 8(f) rows 2-4).  Same library (libasdr_hip.so), same conventions as asdr.h: N independent
 * instances ("channels") per batch object, `ch` = channel index or ASDR_ALL (-1) for setters, int16 audio blocks
 * of 128 samples laid out [channel][block][128] with explicit row strides in blocks, device pointers 16-byte
 * aligned, errors through asdr_last_error().  A batch created with ASDR_NO_DEVICE carries the control plane only;
 * its update calls fail ("needs a HIP device") -- there is no CPU fallback.
 *
 *   asdr_pre_*    replaces class AudioSDRpreProcessor   (SRC/AudioSDRlib/AudioSDRpreProcessor.h:49-84, .cpp:46-169)
 *   asdr_iqgen_*  replaces class AudioIQgenerator       (SRC/AudioSDRlib/AudioIQgenerator.h:48-106,  .cpp:33-87)
 *   asdr_grab_*   replaces class AudioGrabberComplex256 (SRC/AudioSDRlib/AudioGrabberComplex256.h:44-63, .cpp:39-90)
 */
#ifndef ASDR_FRONT_H_
#define ASDR_FRONT_H_

#include <stdint.h>

#include "asdr.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ======================= AudioSDRpreProcessor ======================= */
typedef struct asdr_pre_batch asdr_pre_t;

/* per-channel state as the reference's private members (AudioSDRpreProcessor.h:72-83) plus the detector's last
 * measurements (diagnostics; the reference computes them as locals, .cpp:96-107) */
typedef struct {
  int16_t correction;      /* I2Scorrection  */
  int16_t saved_sample;    /* savedSample    */
  int16_t failure_count;   /* failureCount   */
  int16_t success_count;   /* successCount   */
  int32_t auto_detect;     /* autoDetectFlag */
  int32_t swap;            /* IQswap         */
  int32_t max_line;        /* strongest spectral line of the last detector pass */
  int32_t strong;          /* 1 if it cleared spectralAvgMultiplier x average   */
  float max_power, avg_power, ratio;
} asdr_pre_state_t;

asdr_pre_t *asdr_pre_create(int n_channels, int device);          /* AudioSDRpreProcessor(), .h:51 */
void asdr_pre_destroy(asdr_pre_t *p);
int asdr_pre_n_channels(const asdr_pre_t *p);

/* AudioSDRpreProcessor::update(), .cpp:46-138.  The reference rewrites its two blocks in place
 * (receiveWritable/transmit): pass out pointers equal to the in pointers for that, or distinct buffers.
 *  _update        : host pointers, in place, synchronous.
 *  _update_device : device pointers, asynchronous on `stream`; I/Q rows in_stride_blocks*128 samples apart,
 *                   output rows out_stride_blocks*128.  NULL dI or dQ = the missing-input guard (.cpp:50-52):
 *                   nothing happens, return 0. */
int asdr_pre_update(asdr_pre_t *p, int16_t *I, int16_t *Q, int n_blocks);
int asdr_pre_update_device(asdr_pre_t *p, const int16_t *dI, const int16_t *dQ, int16_t *dIout, int16_t *dQout,
                           int n_blocks, long in_stride_blocks, long out_stride_blocks, void *stream);
int asdr_pre_synchronize(asdr_pre_t *p);

void asdr_pre_startAutoI2SerrorDetection(asdr_pre_t *p, int ch);            /* .cpp:141-147 */
void asdr_pre_stopAutoI2SerrorDetection(asdr_pre_t *p, int ch);             /* .cpp:150-153 */
int asdr_pre_getAutoI2SerrorDetectionStatus(asdr_pre_t *p, int ch);         /* .cpp:157 */
void asdr_pre_setI2SerrorCompensation(asdr_pre_t *p, int ch, int correction); /* .cpp:160-163 */
int16_t asdr_pre_getI2SerrorCompensation(asdr_pre_t *p, int ch);            /* .cpp:166 */
void asdr_pre_swapIQ(asdr_pre_t *p, int ch, int swap);                      /* .cpp:169 */
/* bulk read of every channel's state (synchronises) */
int asdr_pre_read_state(asdr_pre_t *p, asdr_pre_state_t *dst /* [n_channels] */);
float asdr_pre_last_kernel_ms(asdr_pre_t *p);

/* ======================= AudioIQgenerator ======================= */
typedef struct asdr_iqgen_batch asdr_iqgen_t;
asdr_iqgen_t *asdr_iqgen_create(int n_channels, int device);      /* AudioIQgenerator(), .h:51 */
void asdr_iqgen_destroy(asdr_iqgen_t *g);
int asdr_iqgen_n_channels(const asdr_iqgen_t *g);
/* AudioIQgenerator::update(), .cpp:33-87: one real int16 block in, an I block (the input delayed 128 samples)
 * and a Q block (its length-257 Hilbert transform) out.  Every channel has its own 3-block delay line (the
 * reference's are function-static, i.e. shared by all instances of the class -- one instance per batch channel
 * is the meaning kept here). */
int asdr_iqgen_update(asdr_iqgen_t *g, const int16_t *in, int16_t *I, int16_t *Q, int n_blocks);
int asdr_iqgen_update_device(asdr_iqgen_t *g, const int16_t *dIn, int16_t *dI, int16_t *dQ, int n_blocks,
                             long in_stride_blocks, long out_stride_blocks, void *stream);
int asdr_iqgen_synchronize(asdr_iqgen_t *g);
void asdr_iqgen_setGainBalance(asdr_iqgen_t *g, int ch, float balance);     /* .h:55-59 */
float asdr_iqgen_last_kernel_ms(asdr_iqgen_t *g);

/* ======================= AudioGrabberComplex256 ======================= */
typedef struct asdr_grab_batch asdr_grab_t;
asdr_grab_t *asdr_grab_create(int n_channels, int device);        /* AudioGrabberComplex256(), .h:46 */
void asdr_grab_destroy(asdr_grab_t *g);
int asdr_grab_n_channels(const asdr_grab_t *g);
/* AudioGrabberComplex256::update(), .cpp:50-72: every pair of blocks becomes 256 interleaved complex int16
 * samples (re, im, re, im, ...) in the channel's output buffer.  Read-only on I/Q. */
int asdr_grab_update(asdr_grab_t *g, const int16_t *I, const int16_t *Q, int n_blocks);
int asdr_grab_update_device(asdr_grab_t *g, const int16_t *dI, const int16_t *dQ, int n_blocks, long in_stride_blocks,
                            void *stream);
int asdr_grab_newDataAvailable(asdr_grab_t *g, int ch);                      /* .cpp:75-77 */
/* grab(), .cpp:80-90: copies the 512 int16 of channel `ch` to destination if a complete buffer exists (returns 1),
 * else leaves destination untouched (returns 0); clears the channel's new-data flag either way.  < 0 on error. */
int asdr_grab_grab(asdr_grab_t *g, int ch, int16_t *destination /* [512] */);
/* bulk form: every channel's buffer as [n_channels][512] in one copy; clears every new-data flag; returns 1/0 as above */
int asdr_grab_grab_all(asdr_grab_t *g, int16_t *destination /* [n_channels][512] */);
/* the output buffers in HBM, [n_channels][512] int16 (valid after the first complete pair), for device-side consumers */
const int16_t *asdr_grab_device_ptr(asdr_grab_t *g);
int asdr_grab_synchronize(asdr_grab_t *g);
/* Panadapter: power spectrum of every channel's output buffer on the device, |FFT256((re + j*im) / 32768)|^2 in natural bin
 * order (bins 128..255 = negative frequencies), float32 [n_channels][256].  Not a function of the reference library (its
 * example sketches transform the grabbed samples in application code): the arithmetic is this
 * project's radix-2 float32 FFT, stated in oracle/asdr_front_oracle.h.  Returns 1 if buffers were valid (spectra written),
 * 0 if no complete buffer exists yet (destination untouched), < 0 on error.
 *  _power_spectrum        : host destination, synchronous.
 *  _power_spectrum_device : device destination (16-byte aligned), asynchronous on `stream`. */
int asdr_grab_power_spectrum(asdr_grab_t *g, float *destination /* [n_channels][256] */);
int asdr_grab_power_spectrum_device(asdr_grab_t *g, float *dDestination, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ASDR_FRONT_H_ */
