/* asdr_front_device.h -- launch arguments shared by asdr_front.hip (kernels) and asdr_front_host.cpp (C ABI) for the
 * blocks around the hot path: AudioSDRpreProcessor, AudioIQgenerator, AudioGrabberComplex256 (include/asdr_front.h). */
#ifndef ASDR_FRONT_DEVICE_H_
#define ASDR_FRONT_DEVICE_H_

#include <stdint.h>

#include "../../include/asdr_front.h"

#define ASDR_N 128

typedef struct {
  asdr_pre_state_t *state;     /* [n_channels] */
  const int16_t *in_i, *in_q;  /* [n_channels][in_stride][128] */
  int16_t *out_i, *out_q;      /* [n_channels][out_stride][128]; may alias the inputs */
  int32_t n_channels, n_blocks, in_stride, out_stride;
} PreArgs;

typedef struct {
  int16_t *hist;               /* [n_channels][2][128]: the two older blocks of the 3-block delay line (AudioIQgenerator.cpp:52-60), RAW, as a ring */
  uint32_t phase;              /* slot of the OLDER one (the same for every channel: blocks processed so far, mod 2); block b of a call replaces slot (phase + b) & 1 */
  const float *gains;          /* [n_channels][2]: gainI, gainQ */
  const int16_t *in;           /* [n_channels][in_stride][128] */
  int16_t *out_i, *out_q;      /* [n_channels][out_stride][128] */
  int32_t n_channels, n_blocks, in_stride, out_stride;
} IqgenArgs;

typedef struct {
  int16_t *buffer;             /* [n_channels][512]: _buffer    (AudioGrabberComplex256.h:55) */
  int16_t *out_buffer;         /* [n_channels][512]: _outBuffer (AudioGrabberComplex256.h:56) */
  const int16_t *in_i, *in_q;  /* [n_channels][in_stride][128] */
  int32_t n_channels, n_blocks, in_stride;
  int32_t parity;              /* _buffStart / 256 before this call (the same for every channel of a batch) */
} GrabArgs;

#ifdef __cplusplus
extern "C" {
#endif
int asdr_front_upload_tables(void);
int asdr_launch_pre(const PreArgs *a, void *stream);
int asdr_launch_iqgen(const IqgenArgs *a, void *stream);
int asdr_launch_grab(const GrabArgs *a, void *stream);
int asdr_launch_grab_spectrum(const int16_t *out_buffer, float *power, int n_channels, void *stream);
#ifdef __cplusplus
}
#endif

#endif /* ASDR_FRONT_DEVICE_H_ */
