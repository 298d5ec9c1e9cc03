// STAND-IN for CMSIS-DSP V1.4.5's arm_math.h (tools/ref_shim/README.md): the types and the two biquad functions AudioSDR uses
// (arm_math.h:1257-1262, 1360-1378 of the reference's ARM_MATH UPDATE copy).  The process function is the operation order recovered from
// the reference's own Cortex-M4 object (tests/test_cmsis_object.py executes that object: vmul / vadd only, acc = b0 x; += b1 x1; += b2 x2;
// += a1 y1; += a2 y2, state {x1, x2, y1, y2} per stage, stage-major, later stages in place on pDst).
#pragma once
#include <stdint.h>
#include <string.h>
typedef float float32_t;
typedef double float64_t;
typedef int16_t q15_t;
typedef int32_t q31_t;
typedef struct {
  uint32_t numStages;
  float32_t *pState;
  float32_t *pCoeffs;
} arm_biquad_casd_df1_inst_f32;
static inline void arm_biquad_cascade_df1_init_f32(arm_biquad_casd_df1_inst_f32 *S, uint8_t numStages, float32_t *pCoeffs, float32_t *pState) {
  S->numStages = numStages; S->pCoeffs = pCoeffs; S->pState = pState;
  memset(pState, 0, (4u * (uint32_t)numStages) * sizeof(float32_t));
}
static inline void arm_biquad_cascade_df1_f32(const arm_biquad_casd_df1_inst_f32 *S, float32_t *pSrc, float32_t *pDst, uint32_t blockSize) {
  float32_t *pIn = pSrc, *pState = S->pState, *pCoeffs = S->pCoeffs;
  for (uint32_t stage = 0; stage < S->numStages; stage++) {
    const float32_t b0 = pCoeffs[0], b1 = pCoeffs[1], b2 = pCoeffs[2], a1 = pCoeffs[3], a2 = pCoeffs[4];
    volatile float32_t acc;   // (volatile: every operation separately rounded whatever the host compiler would like to contract)
    float32_t Xn1 = pState[0], Xn2 = pState[1], Yn1 = pState[2], Yn2 = pState[3];
    float32_t *pOut = pDst;
    for (uint32_t n = 0; n < blockSize; n++) {
      const float32_t Xn = pIn[n];
      acc = b0 * Xn; acc = acc + b1 * Xn1; acc = acc + b2 * Xn2; acc = acc + a1 * Yn1; acc = acc + a2 * Yn2;
      pOut[n] = acc;
      Xn2 = Xn1; Xn1 = Xn; Yn2 = Yn1; Yn1 = acc;
    }
    pState[0] = Xn1; pState[1] = Xn2; pState[2] = Yn1; pState[3] = Yn2;
    pState += 4; pCoeffs += 5;
    pIn = pDst;
  }
}
// The pre-processor's image detector calls CMSIS's 128-point complex FFT and squared magnitudes (AudioSDRpreProcessor.cpp:91-92): neither is in
// the reference tree in source form, so the stand-in build has NONE -- the front check below runs the paths that do not reach them (skew
// correction, I/Q swap) and aborts loudly if they are ever called.  (The FFT itself is pinned where it can be: tests/test_cmsis_object.py runs
// the reference's own Cortex-M4 objects.)
#include <stdio.h>
#include <stdlib.h>
typedef struct { uint16_t fftLen; const float32_t *pTwiddle; const uint16_t *pBitRevTable; uint16_t bitRevLength; } arm_cfft_instance_f32;
static const arm_cfft_instance_f32 arm_cfft_sR_f32_len128 = {128, NULL, NULL, 0};
static inline void arm_cfft_f32(const arm_cfft_instance_f32 *, float32_t *, uint8_t, uint8_t) { fprintf(stderr, "stand-in build: arm_cfft_f32 is not available\n"); abort(); }
static inline void arm_cmplx_mag_squared_f32(float32_t *, float32_t *, uint32_t) { fprintf(stderr, "stand-in build: arm_cmplx_mag_squared_f32 is not available\n"); abort(); }
