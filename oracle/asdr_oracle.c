/* asdr_oracle.c -- CPU ORACLE (test infrastructure; see asdr_oracle.h header comment).
 *
 * PARITY UNPINNED: restated from /root/reference/SRC/AudioSDRlib/AudioSDR.{h,cpp}; the
 * reference has no tests/golden vectors and is unbuildable in this image.
 *
 * Build: gcc -std=c11 -O2 -ffp-contract=off (no FMA contraction: every float operation below
 * is a separately rounded IEEE-754 binary32 operation, every double operation binary64, exactly
 * as the C++ usual-arithmetic-conversion rules give for the reference's expressions with
 * FLT_EVAL_METHOD == 0).  An unsuffixed literal in a comment, e.g. 32767.0, marks a place where
 * the reference computes in double ("double islands").
 *
 * Cites: ".cpp:N" = SRC/AudioSDRlib/AudioSDR.cpp line N; ".h:N" = SRC/AudioSDRlib/AudioSDR.h.
 */
#define _GNU_SOURCE
#include "asdr_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../audiosdr_amd/csrc/asdr_tables.h"

#define AO_PLL_WRAP_MAX 64 /* == ASDR_PLL_WRAP_MAX of the product */

#if defined(FLT_EVAL_METHOD) && FLT_EVAL_METHOD != 0
#error "oracle requires FLT_EVAL_METHOD == 0 (SSE2 float/double arithmetic)"
#endif

/* Arduino.h's PI (a double literal); every PI in the reference is this value. */
#define AO_PI 3.1415926535897932384626433832795
/* AUDIO_SAMPLE_RATE_EXACT of the Teensy 4.x core (a float literal).  SURVEY.md 0.1. */
#define AO_FS 44100.0f

#define N AO_BLOCK

/* ------------------------------------------------------------------------------------------ */
/* per-instance data: parameters + carried state (.h:161-326, .cpp:41-44, .cpp:690-694)       */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
  const float *coefs; /* 4 sections x {b0,b1,b2,a1,a2}   (arm_math.h:1257-1262: pCoeffs)  */
  float state[16];    /* 4 sections x {x1,x2,y1,y2}      (pState)                         */
} ao_biquad4_t;

struct asdr_oracle {
  /* --- general (.h:164-182) */
  float if_center, bw_ssb, bw_cw, bw_wspr, bw_am;
  float in_gain, in_gain_i, in_gain_q, gain_balance;
  float output_gain, out_gain, current_out_gain;
  float freq_shift;
  uint16_t mode;
  int muted;
  float audio[N], I[N], Q[N]; /* _audioOut, _Idata, _Qdata are members: persist across calls */
  /* --- IIR filters (.h:184-195) */
  ao_biquad4_t if_i, if_q, img_i, img_q, audio_bq;
  int16_t current_filter;
  int audio_filter_enabled;
  /* --- SSB path statics (.cpp:41-44) */
  float buf_i[4 * N], buf_q[4 * N];
  float phase_ssb, phase_am;
  /* --- ALS (.h:198-205) */
  int16_t als_m, als_delay;
  float als_lambda;
  float als_in[2 * N], als_w[N];
  int als_enabled, als_notch, als_adaptive;
  /* --- AGC (.h:208-232) */
  float agc_carrier, agc_alpha_att, agc_alpha_rel, agc_attack_ms, agc_beta_att, agc_beta_rel;
  float agc_gain;
  float agc_table[130]; /* [129] in the reference; entry 129 aliases _agc_hangTime (.h:219-220) */
  float agc_knee, agc_static_gain, agc_slope, agc_release_ms, agc_threshold;
  float agc_abs, agc_old_abs;
  uint32_t agc_hang_count, agc_hang_counter;
  int agc_active, agc_enabled;
  /* --- noise blanker (.h:235-246) */
  float nb_i[3 * N], nb_q[3 * N], nb_mask[3 * N];
  float nb_alpha, nb_beta, nb_threshold, nb_mag, nb_avg;
  int16_t nb_pre, nb_post;
  int nb_enabled, nb_detected;
  /* --- SAM PLL (.h:249-284 members; .cpp:690-694 function statics) */
  float two_pi_f, half_pi_f, alpha_freq, beta_freq, f_conv, lock_lo, lock_hi;
  float pll_b0, pll_b1, pll_a1;
  float pll_d0, pll_d1, pll_phase_est, pll_freq;
  int pll_locked;
  float pll_y_re, pll_y_im, pll_prev_filt, pll_err, pll_filt;
  /* --- test aid */
  int unknown_mode_silence; /* model the PRODUCT's opt-out for unknown mode values (default 0 = the reference) */
  int pll_wrap_bound;       /* model the PRODUCT's bounded PLL phase wrap (default 0 = the reference's unbounded loops) */
  int pll_stalled;          /* default mode: a wrap loop of .cpp:735-736 could never have ended (the reference would hang) */
  int taps_on;
  float taps[AO_N_TAPS][N];
};

/* ------------------------------------------------------------------------------------------ */
/* scalar helpers                                                                              */
/* ------------------------------------------------------------------------------------------ */

/* .h:364: intPhase = (long)(Phase * 65535.0 / twoPI) after one conditional wrap each way
 * (.h:362-363); twoPI is `const float32_t twoPI = 2.0 * PI` (.h:359), so the division is by the
 * float value widened to double. */
uint16_t ao_sin_index(float phase) {
  const float two_pi = (float)(2.0 * AO_PI);
  if (phase >= two_pi) phase -= two_pi;       /* float - float */
  if (phase < 0.0) phase += two_pi;           /* compare in double == compare in float */
  long ip = (long)((double)phase * 65535.0 / (double)two_pi);
  return (uint16_t)ip;
}

/* .h:365-369: table index = high byte, remainder = low byte, linear interpolation
 * `val1 + (((val2 - val1) * (float)delta) / 256.0)`: float product, double divide, double add,
 * rounded to float by the return. */
float ao_sin_from_index(uint16_t ip) {
  uint16_t index = ip >> 8, delta = ip & 0xFF;
  float v1 = asdr_sine_table[index], v2 = asdr_sine_table[index + 1];
  return (float)((double)v1 + ((double)((v2 - v1) * (float)delta) / 256.0));
}

float ao_sin_f32(float phase) { return ao_sin_from_index(ao_sin_index(phase)); }

/* .h:375-377: cos(x) = sin(x + PI/2.0); the sum is formed in double and rounded to the float
 * parameter of sin_f32. */
float ao_cos_f32(float phase) { return ao_sin_f32((float)((double)phase + AO_PI / 2.0)); }

/* .h:384-388 */
static float approx_atan(float z) {
  const float n1 = 0.97239411f, n2 = -0.19194795f;
  return (n1 + n2 * z * z) * z; /* ((n2*z)*z + n1) * z, all float */
}

/* .h:390-408.  Note the mixed constants: `+ PI` / `- PI` add the DOUBLE PI (.h:396-397), while
 * halfPI is the local `const float32_t halfPI = 0.5 * PI` (.h:391), so those adds are float. */
float ao_approx_atan2_f32(float y, float x) {
  const float half_pi = (float)(0.5 * AO_PI);
  if (x != 0.0) {
    if (fabsf(x) > fabsf(y)) {
      float z = y / x;
      if (x > 0.0) return approx_atan(z);
      else if (y >= 0.0) return (float)((double)approx_atan(z) + AO_PI);
      else return (float)((double)approx_atan(z) - AO_PI);
    } else {
      float z = x / y;
      if (y > 0.0) return -approx_atan(z) + half_pi;
      else return -approx_atan(z) - half_pi;
    }
  } else {
    if (y > 0.0) return half_pi;
    else if (y < 0.0) return -half_pi;
  }
  return 0.0f;
}

/* .h:434-446 (the Serial.print at :444 is not behaviour).  uint32 arithmetic wraps, so x = 0
 * gives bits 0x9FC00000 before the Newton step, as on the target. */
float ao_fast_sqrt_f32(float x, int n_iter) {
  union { float f; uint32_t i; } v;
  v.f = x;
  v.i -= 1u << 23;
  v.i >>= 1;
  v.i += 1u << 29;
  float out = v.f;
  for (int i = 0; i < n_iter; i++) out = (float)(0.5 * (double)(out + x / out));
  return out;
}

/* .h:483-491 */
float ao_log2_approx_f32(float input) {
  int exponent;
  float mantissa = frexpf(fabsf(input), &exponent);
  return (((1.23149591368684f * mantissa - 4.11852516267426f) * mantissa + 6.02197014179219f) * mantissa -
          3.13396450166353f) + exponent;
}

/* CMSIS-DSP V1.4.5 arm_biquad_cascade_df1_f32 (prototype arm_math.h:1360-1378; implementation
 * only shipped as Cortex-M4 objects -> restated from the published CMSIS algorithm):
 * stage-major; per sample acc = b0*x; acc += b1*x1; acc += b2*x2; acc += a1*y1; acc += a2*y2
 * with separately rounded multiplies and adds; state {x1,x2,y1,y2} per stage; stages after the
 * first run in place on dst. */
void ao_biquad_cascade_df1(const float *coefs, float *state, int n_stages, const float *src, float *dst, int n) {
  const float *in = src;
  for (int s = 0; s < n_stages; s++) {
    const float b0 = coefs[5 * s], b1 = coefs[5 * s + 1], b2 = coefs[5 * s + 2];
    const float a1 = coefs[5 * s + 3], a2 = coefs[5 * s + 4];
    float x1 = state[4 * s], x2 = state[4 * s + 1], y1 = state[4 * s + 2], y2 = state[4 * s + 3];
    for (int i = 0; i < n; i++) {
      float x = in[i];
      float acc = b0 * x;
      acc += b1 * x1;
      acc += b2 * x2;
      acc += a1 * y1;
      acc += a2 * y2;
      x2 = x1; x1 = x; y2 = y1; y1 = acc;
      dst[i] = acc;
    }
    state[4 * s] = x1; state[4 * s + 1] = x2; state[4 * s + 2] = y1; state[4 * s + 3] = y2;
    in = dst;
  }
}

static void biquad4_init(ao_biquad4_t *f, int pool_index) { /* arm_biquad_cascade_df1_init_f32 */
  f->coefs = asdr_bq_pool[pool_index];
  memset(f->state, 0, sizeof f->state);
}
static void biquad4_run(ao_biquad4_t *f, const float *src, float *dst) {
  ao_biquad_cascade_df1(f->coefs, f->state, 4, src, dst, N);
}

/* .h:508-526: complex multiply by e^{j phase}; phase accumulates sequentially in float. */
float ao_freq_shifter(float *I, float *Q, float freq_shift, float phase0) {
  const float two_pi = (float)(2.0 * AO_PI);
  float phase_inc = freq_shift * (two_pi / AO_FS);
  float phase = phase0;
  for (int i = 0; i < N; i++) {
    float c = ao_cos_f32(phase), s = ao_sin_f32(phase);
    float ti = I[i], tq = Q[i];
    I[i] = ti * c - tq * s;
    Q[i] = tq * c + ti * s;
    phase += phase_inc;
    if (phase > two_pi) phase -= two_pi;
    else if (phase < 0.0) phase += two_pi;
  }
  return phase;
}

/* `(int)` of a double outside the int range is undefined in C.  The reference's target (ARM Cortex-M7,
 * vcvt.s32.f64) SATURATES and converts NaN to 0 -- and so does the GPU's v_cvt_i32_f64 -- whereas x86's cvttsd2si
 * returns INT_MIN.  The oracle pins the ARM definition (only reachable when a diverging ALS filter or extreme
 * gains push the audio beyond +-65536 full scales). */
int32_t ao_f64_to_i32(double v) {
  if (v != v) return 0;
  if (v >= 2147483647.0) return INT32_MAX;
  if (v <= -2147483648.0) return INT32_MIN;
  return (int32_t)v;
}

/* .cpp:68: ((float)s / 32767.0) * gain, in double, before the store rounds it to float */
double ao_scale_sample(int16_t s, float gain) { return ((double)(float)s / 32767.0) * (double)gain; }

const float *ao_hilbert_taps(void) { return asdr_hilbert_taps; }
const float *ao_sine_table(void) { return asdr_sine_table; }
const float *ao_biquad_table(int i) { return asdr_bq_pool[i]; }

/* ------------------------------------------------------------------------------------------ */
/* stages                                                                                      */
/* ------------------------------------------------------------------------------------------ */

/* .cpp:676-682 */
static void nb_reset(asdr_oracle_t *o) {
  for (int i = 0; i < 3 * N; i++) { o->nb_i[i] = 0.0f; o->nb_q[i] = 0.0f; o->nb_mask[i] = 1.0f; }
}

/* .cpp:606-650.  3-block sliding buffers; detection over i = 78..255 (restarting 50 samples
 * early, so those update the running average twice); trailing-edge ramp; output = mask x oldest
 * block (2 blocks of latency).  Both branches at .cpp:638/:641 test the same condition, so only
 * the first (transition_dn) ever executes. */
static void nb_process(asdr_oracle_t *o, float *I, float *Q) {
  static const float trans_dn[7] = {0.933, 0.750, 0.500, 0.250, 0.067, 0.000, 0.000};
  o->nb_detected = 0;
  for (int i = 0; i < N; i++) {
    o->nb_i[i] = o->nb_i[N + i]; o->nb_i[N + i] = o->nb_i[2 * N + i]; o->nb_i[2 * N + i] = I[i];
    o->nb_q[i] = o->nb_q[N + i]; o->nb_q[N + i] = o->nb_q[2 * N + i]; o->nb_q[2 * N + i] = Q[i];
    o->nb_mask[i] = o->nb_mask[N + i]; o->nb_mask[N + i] = o->nb_mask[2 * N + i]; o->nb_mask[2 * N + i] = 1.0f;
  }
  for (int i = N - 50; i < 2 * N; i++) {
    o->nb_mag = ao_fast_sqrt_f32(o->nb_i[i] * o->nb_i[i] + o->nb_q[i] * o->nb_q[i], 1);
    if (o->nb_mag > o->nb_avg * o->nb_threshold) {
      for (int j = -o->nb_pre; j < o->nb_post + 1; j++) o->nb_mask[i + j] = 0.0f;
      o->nb_detected = 1;
    }
    o->nb_avg = o->nb_alpha * o->nb_avg + o->nb_beta * o->nb_mag;
  }
  for (int i = N; i < 2 * N; i++) {
    if (o->nb_mask[i] == 1.0 && o->nb_mask[i - 1] == 0.0)
      for (int j = 0; j < 7; j++) o->nb_mask[i - 7 + j] = trans_dn[j];
    /* .cpp:641-643 repeats the same test in an else-if: dead code */
  }
  for (int i = 0; i < N; i++) { I[i] = o->nb_mask[i] * o->nb_i[i]; Q[i] = o->nb_mask[i] * o->nb_q[i]; }
}

/* .cpp:84-119: shift to baseband, slide the 4-block history, folded 257-tap Hilbert on Q
 * (k ascending, float accumulate from 0.0), I delayed by 128, sideband combine. */
static void ssb_demod(asdr_oracle_t *o) {
  o->phase_ssb = ao_freq_shifter(o->I, o->Q, -o->freq_shift, o->phase_ssb);
  for (int i = 0; i < N; i++) {
    o->buf_i[i] = o->buf_i[N + i]; o->buf_i[N + i] = o->buf_i[2 * N + i];
    o->buf_i[2 * N + i] = o->buf_i[3 * N + i]; o->buf_i[3 * N + i] = o->I[i];
    o->buf_q[i] = o->buf_q[N + i]; o->buf_q[N + i] = o->buf_q[2 * N + i];
    o->buf_q[2 * N + i] = o->buf_q[3 * N + i]; o->buf_q[3 * N + i] = o->Q[i];
  }
  const int flen = 257, fdelay = (257 - 1) / 2; /* .h:755-756 */
  for (int i = 0; i < N; i++) {
    o->Q[i] = 0.0f;
    for (int k = 0; k < flen / 4; k++) {
      int i1 = (3 * N + i) - (2 * k + 1);
      int i2 = (3 * N + i) - flen + 2 * (k + 1);
      o->Q[i] += asdr_hilbert_taps[k] * (o->buf_q[i1] - o->buf_q[i2]);
    }
    o->I[i] = o->buf_i[3 * N + i - fdelay];
  }
  for (int i = 0; i < N; i++) {
    if (o->mode == AO_USB || o->mode == AO_CW_USB || o->mode == AO_WSPR) o->audio[i] = o->I[i] - o->Q[i];
    else if (o->mode == AO_LSB || o->mode == AO_CW_LSB) o->audio[i] = o->I[i] + o->Q[i];
  }
}

/* .cpp:688-749: quadrature PLL.  The per-sample rotation of (I,Q) only happens while the lock
 * detector is true AT THAT SAMPLE; the caller then takes audio = Q (.cpp:126-128). */
static void sam_demod(asdr_oracle_t *o) {
  const float two_pi = (float)(2.0 * AO_PI); /* .cpp:689 local const float32_t twoPI */
  for (int i = 0; i < N; i++) {
    float x_re = o->I[i], x_im = o->Q[i];
    float d_re = x_re * o->pll_y_re + x_im * o->pll_y_im;
    float d_im = x_im * o->pll_y_re - x_re * o->pll_y_im;
    o->pll_err = ao_approx_atan2_f32(d_im, d_re);
    o->pll_d1 = o->pll_d0;
    o->pll_d0 = o->pll_err - o->pll_a1 * o->pll_d1;
    o->pll_filt = o->pll_b0 * o->pll_d0 + o->pll_b1 * o->pll_d1;
    /* .cpp:732: `phase_est += (a + b)/2.0` : float sum, double halve, double add, float store */
    o->pll_phase_est = (float)((double)o->pll_phase_est + (double)(o->pll_filt + o->pll_prev_filt) / 2.0);
    o->pll_prev_filt = o->pll_filt;
    if (!o->pll_wrap_bound) {
      /* DEFAULT = the reference, .cpp:735-736: two unbounded loops (compare vs double PI).  They never end once
       * phase_est -+ twoPI == phase_est (an infinite or huge estimate): a Teensy instance would hang there.  The oracle cannot
       * hang the test process, so it notes the fact (ao_pll_stalled) and leaves the loop; everything else is the reference's. */
      while (o->pll_phase_est >= AO_PI) {
        float nxt = o->pll_phase_est - two_pi;
        if (nxt == o->pll_phase_est) { o->pll_stalled = 1; break; }
        o->pll_phase_est = nxt;
      }
      while (o->pll_phase_est < -AO_PI) {
        float nxt = o->pll_phase_est + two_pi;
        if (nxt == o->pll_phase_est) { o->pll_stalled = 1; break; }
        o->pll_phase_est = nxt;
      }
    } else {
      /* ao_set_pll_wrap_bound(o, 1): the PRODUCT's defined difference -- at most ASDR_PLL_WRAP_MAX turns per sample,
       * then phase_est = 0 (a wave must not hang the GPU).  A physical loop-filter step is below pi, i.e. at most one turn, so
       * the bound never acts on a finite signal; the GPU tests switch it on so that the two stay comparable on any input. */
      int turns = 0;
      while (o->pll_phase_est >= AO_PI && turns < AO_PLL_WRAP_MAX) { o->pll_phase_est -= two_pi; turns++; }
      while (o->pll_phase_est < -AO_PI && turns < AO_PLL_WRAP_MAX) { o->pll_phase_est += two_pi; turns++; }
      if (turns >= AO_PLL_WRAP_MAX) o->pll_phase_est = 0.0f;
    }
    o->pll_y_re = ao_cos_f32(o->pll_phase_est);
    o->pll_y_im = ao_sin_f32(o->pll_phase_est);
    o->pll_freq = o->alpha_freq * o->pll_freq + o->beta_freq * (o->pll_filt * o->f_conv);
    o->pll_locked = (o->pll_freq > o->lock_lo) && (o->pll_freq < o->lock_hi);
    if (o->pll_locked) {
      float ti = o->I[i], tq = o->Q[i];
      o->I[i] = ti * o->pll_y_re + tq * o->pll_y_im;
      o->Q[i] = -ti * o->pll_y_im + tq * o->pll_y_re;
    }
  }
}

/* .cpp:132-143: envelope detector (also the SAM fall-back when the PLL is unlocked at the end
 * of the block).  The carrier tracker runs in double (`.995*x + 0.005*abs(y)`), stored float. */
static void am_envelope(asdr_oracle_t *o) {
  o->phase_am = ao_freq_shifter(o->I, o->Q, -o->if_center, o->phase_am);
  biquad4_run(&o->img_i, o->I, o->I);
  biquad4_run(&o->img_q, o->Q, o->Q);
  for (int i = 0; i < N; i++) {
    o->audio[i] = sqrtf(o->I[i] * o->I[i] + o->Q[i] * o->Q[i]); /* sqrt of a float, correctly rounded */
    o->agc_carrier = (float)(.995 * (double)o->agc_carrier + 0.005 * (double)fabsf(o->audio[i]));
  }
}

/* .cpp:483-494 */
float ao_agc_static_compressor(const float *table, uint16_t input) {
  uint16_t indx = input >> 8;
  if (indx > 127) indx = 127;
  uint16_t frac = input & 0xFF;
  float delta = (float)((double)(float)frac / 256.0);
  return table[indx] + (table[indx + 1] - table[indx]) * delta;
}

/* .cpp:404-436 */
static void agc_process(asdr_oracle_t *o, float *buf) {
  for (int i = 0; i < N; i++) {
    if (o->mode == AO_AM) o->agc_abs = (float)(2.0 * (double)o->agc_carrier);
    else o->agc_abs = (float)fabs((double)buf[i]);
    if (o->agc_abs > 1.0) o->agc_abs = 1.0f;
    if (o->agc_abs > o->agc_old_abs) { /* attack */
      o->agc_abs = o->agc_alpha_att * o->agc_old_abs + o->agc_beta_att * o->agc_abs;
      o->agc_old_abs = o->agc_abs;
      o->agc_hang_counter = o->agc_hang_count;
      o->agc_gain = ao_agc_static_compressor(o->agc_table, (uint16_t)(int)((double)o->agc_abs * 32767.0));
    } else {
      if (o->agc_hang_counter > 0) o->agc_hang_counter--; /* hang */
      else {                                              /* release */
        o->agc_abs = o->agc_alpha_rel * o->agc_old_abs + o->agc_beta_rel * o->agc_abs;
        o->agc_old_abs = o->agc_abs;
        o->agc_gain = ao_agc_static_compressor(o->agc_table, (uint16_t)(int)((double)o->agc_abs * 32767.0));
      }
    }
    o->agc_active = ((double)o->agc_gain < 0.99);
    float out = o->agc_gain * o->agc_static_gain * buf[i];
    out = (out > 1.0) ? 1.0f : out;
    out = (out < -1.0) ? -1.0f : out;
    buf[i] = out;
  }
}

/* .cpp:459-480.  130 entries are written (the reference's 130th lands on _agc_hangTime). */
static void agc_build_table(asdr_oracle_t *o) {
  float lin_lo = expf((float)(2.3025 * ((double)o->agc_threshold - (double)o->agc_knee / 2.0) / 20.0));
  float lin_hi = expf((float)(2.3025 * ((double)o->agc_threshold + (double)o->agc_knee / 2.0) / 20.0));
  for (int i = 0; i < 129 + 1; i++) {
    float input = (float)((double)(float)i / 128.0);
    float in_db = (float)(6.026 * (double)ao_log2_approx_f32(input));
    float out_db;
    if (input < lin_lo) o->agc_table[i] = 1.0f;
    else if (input > lin_hi) {
      out_db = o->agc_threshold + (in_db - o->agc_threshold) * o->agc_slope;
      o->agc_table[i] = expf((float)(2.3025 * (double)(out_db - in_db) / 20.0));
    } else {
      double t = (double)(in_db - o->agc_threshold) + (double)o->agc_knee / 2.0;
      out_db = (float)((double)in_db + (((double)o->agc_slope - 1.0) * t * t) / (2.0 * (double)o->agc_knee));
      o->agc_table[i] = expf((float)(2.3025 * (double)(out_db - in_db) / 20.0));
    }
  }
}

/* .cpp:439-457 */
static void agc_init(asdr_oracle_t *o) {
  o->agc_threshold = -60.0f; o->agc_slope = 0.1f; o->agc_knee = 2.0f;
  o->agc_attack_ms = 5.0f; o->agc_release_ms = 500.0f;
  o->agc_table[129] = 100.0f; /* _agc_hangTime = 100.0 */
  o->agc_hang_count = (uint32_t)((double)AO_FS * ((double)o->agc_table[129] / 1000.0));
  o->agc_alpha_att = (float)exp(log(0.1) / ((double)(AO_FS * o->agc_attack_ms) / 1000.0));
  o->agc_beta_att = (float)(1.0 - (double)o->agc_alpha_att);
  o->agc_alpha_rel = (float)exp(log(0.1) / ((double)(AO_FS * o->agc_release_ms) / 1000.0));
  o->agc_beta_rel = (float)(1.0 - (double)o->agc_alpha_rel);
  o->agc_enabled = 1;
  agc_build_table(o);
}

/* .cpp:324-352.  Reads outside the 256-sample history (M + delay > 128 or delay < 0) are
 * undefined in the reference; here (and in the product) they read 0.0 -- DESIGN.md. */
static float als_hist(const asdr_oracle_t *o, int idx) { return (idx >= 0 && idx < 2 * N) ? o->als_in[idx] : 0.0f; }
static void als_process(asdr_oracle_t *o, float *buf) {
  uint16_t count = 0;
  for (int i = 0; i < N; i++) { o->als_in[i] = o->als_in[N + i]; o->als_in[N + i] = buf[i]; }
  for (int i = N; i < 2 * N; i++) {
    float y = 0.0f;
    for (int j = 0; j < o->als_m; j++) y += o->als_w[j] * als_hist(o, (i - o->als_delay) - j);
    float e = o->als_in[i] - y;
    if (o->als_adaptive) {
      if (count == 0)
        for (int j = 0; j < o->als_m; j++) {
          float g = e * als_hist(o, i - o->als_delay - j);
          o->als_w[j] += o->als_lambda * g;
        }
      count = (uint16_t)((count + 1) % 4);
    }
    buf[i - N] = o->als_notch ? e : y;
  }
}

/* ------------------------------------------------------------------------------------------ */
/* update()  .cpp:39-168                                                                       */
/* ------------------------------------------------------------------------------------------ */
#define TAP(id, src) do { if (o->taps_on) memcpy(o->taps[id], (src), sizeof(float) * N); } while (0)

void ao_update(asdr_oracle_t *o, const int16_t *bi, const int16_t *bq, int16_t *out) {
  for (int i = 0; i < N; i++) { /* .cpp:67-70 */
    o->I[i] = (float)ao_scale_sample(bi[i], o->in_gain_i);
    o->Q[i] = (float)ao_scale_sample(bq[i], o->in_gain_q);
  }
  TAP(AO_TAP_SCALED_I, o->I); TAP(AO_TAP_SCALED_Q, o->Q);
  if (o->nb_enabled) nb_process(o, o->I, o->Q); /* .cpp:73 */
  TAP(AO_TAP_NB_I, o->I); TAP(AO_TAP_NB_Q, o->Q);
  biquad4_run(&o->if_i, o->I, o->I); /* .cpp:77-78 */
  biquad4_run(&o->if_q, o->Q, o->Q);
  TAP(AO_TAP_IF_I, o->I); TAP(AO_TAP_IF_Q, o->Q);

  if (o->mode == AO_USB || o->mode == AO_LSB || o->mode == AO_CW_USB || o->mode == AO_CW_LSB || o->mode == AO_WSPR) {
    ssb_demod(o);
  } else if (o->mode == AO_AM || o->mode == AO_SAM) {
    if (o->mode == AO_SAM) {
      sam_demod(o);
      for (int i = 0; i < N; i++) o->audio[i] = o->Q[i];
    }
    if (o->mode == AO_AM || (o->mode == AO_SAM && !o->pll_locked)) am_envelope(o);
  } else if (o->unknown_mode_silence) { /* the product with asdr_set_exact_unknown_mode(b, 0): silence instead of stale audio */
    memset(o->audio, 0, sizeof o->audio);
  } /* any other mode value: _audioOut keeps last block's (already post-processed) samples (.cpp:84,122,149-161) */
  TAP(AO_TAP_MIX_I, o->I); TAP(AO_TAP_MIX_Q, o->Q);
  TAP(AO_TAP_DEMOD, o->audio);

  if (o->audio_filter_enabled) { /* .cpp:149, .cpp:280-286 */
    float tmp[N];
    memcpy(tmp, o->audio, sizeof tmp);
    biquad4_run(&o->audio_bq, tmp, o->audio);
  }
  TAP(AO_TAP_AUDIO_FILT, o->audio);
  if (o->agc_enabled) agc_process(o, o->audio); /* .cpp:152 */
  TAP(AO_TAP_AGC, o->audio);
  if (o->als_enabled) als_process(o, o->audio); /* .cpp:155 */
  TAP(AO_TAP_ALS, o->audio);

  /* .cpp:158-161: float product, double x 32767.0, truncate to int, wrap into int16 */
  if (o->muted) for (int i = 0; i < N; i++) out[i] = 0;
  else for (int i = 0; i < N; i++) out[i] = (int16_t)ao_f64_to_i32((double)(o->output_gain * o->audio[i]) * 32767.0);
}

/* ------------------------------------------------------------------------------------------ */
/* construction / control surface                                                              */
/* ------------------------------------------------------------------------------------------ */
asdr_oracle_t *ao_create(void) {
  asdr_oracle_t *o = (asdr_oracle_t *)calloc(1, sizeof *o); /* static storage: zero first */
  if (!o) return NULL;
  /* in-class initialisers, .h:164-284 */
  o->if_center = 6890.0f; o->bw_ssb = 3000.0f; o->bw_cw = 1000.0f; o->bw_wspr = 1000.0f; o->bw_am = 8500.0f;
  o->in_gain = 1.0f; o->in_gain_i = 1.0f; o->in_gain_q = 1.0f; o->gain_balance = 1.0f;
  o->output_gain = 0.5f; o->out_gain = 1.0f; o->current_out_gain = 1.0f;
  o->mode = 0; o->muted = 1;
  o->als_m = 55; o->als_delay = 3; o->als_lambda = 0.5f;
  o->als_enabled = 0; o->als_notch = 1; o->als_adaptive = 1;
  o->agc_table[129] = 100.0f; o->agc_static_gain = 10.0f; o->agc_active = 1; o->agc_enabled = 1;
  o->nb_alpha = 0.995f; o->nb_beta = (float)(1.0 - (double)o->nb_alpha);
  o->nb_threshold = 1.2f; o->nb_mag = 0.0f; o->nb_avg = 10.0f; o->nb_pre = 10; o->nb_post = 10;
  o->nb_enabled = 1; o->nb_detected = 0;
  o->two_pi_f = (float)(2.0 * AO_PI); o->half_pi_f = (float)(0.5 * AO_PI);
  o->alpha_freq = 0.995f; o->beta_freq = (float)(1.0 - (double)o->alpha_freq);
  o->f_conv = AO_FS / o->two_pi_f;
  o->lock_lo = (float)((double)o->if_center - 1000.0); o->lock_hi = (float)((double)o->if_center + 1000.0);
  { /* .h:260-284 PLL loop filter */
    float wn = 0.07f, zeta = 0.707f, Ka = 1000.f;
    float tau1 = Ka / (wn * wn);
    float tau2 = 2 * zeta / wn;
    o->pll_b0 = (float)((double)(2 * Ka / tau1) * (1.0 + 2.0 * (double)tau2));
    o->pll_b1 = (float)((double)(2 * Ka / tau1) * (1.0 - 2.0 * (double)tau2));
    o->pll_a1 = -1.0f;
  }
  o->pll_phase_est = 0.0f; o->pll_freq = 0.0f;
  ao_init(o);
  return o;
}

void ao_destroy(asdr_oracle_t *o) { free(o); }
void ao_enable_taps(asdr_oracle_t *o, int on) { o->taps_on = on; }
void ao_set_unknown_mode_silence(asdr_oracle_t *o, int on) { o->unknown_mode_silence = on; }
void ao_set_pll_wrap_bound(asdr_oracle_t *o, int on) { o->pll_wrap_bound = on; }
int ao_pll_stalled(const asdr_oracle_t *o) { return o->pll_stalled; }
void ao_test_set_pll_phase(asdr_oracle_t *o, float phase_est) { o->pll_phase_est = phase_est; }
float ao_test_get_pll_phase(const asdr_oracle_t *o) { return o->pll_phase_est; }
const float *ao_tap(const asdr_oracle_t *o, int tap) { return o->taps[tap]; }

void ao_init(asdr_oracle_t *o) { /* .cpp:174-185 */
  biquad4_init(&o->audio_bq, ASDR_TBL_AUDIO_BASE + AO_AUDIO_2700);
  biquad4_init(&o->if_i, ASDR_TBL_IF_SSB);
  biquad4_init(&o->if_q, ASDR_TBL_IF_SSB);
  biquad4_init(&o->img_i, ASDR_TBL_AM_IMAGE);
  biquad4_init(&o->img_q, ASDR_TBL_AM_IMAGE);
  agc_init(o);
  nb_reset(o);
  ao_setDemodMode(o, AO_LSB);
  o->muted = 0;
}

float ao_setDemodMode(asdr_oracle_t *o, int new_mode) { /* .cpp:187-222 */
  o->mode = (uint16_t)new_mode;
  int tbl = -1;
  if (o->mode == AO_USB) { o->freq_shift = (float)((double)o->if_center - (double)o->bw_ssb / 2.0); tbl = ASDR_TBL_IF_SSB; }
  else if (o->mode == AO_LSB) { o->freq_shift = (float)((double)o->if_center + (double)o->bw_ssb / 2.0); tbl = ASDR_TBL_IF_SSB; }
  else if (o->mode == AO_WSPR) { o->freq_shift = (float)((double)o->if_center - (double)o->bw_ssb / 2.0); tbl = ASDR_TBL_IF_WSPR; }
  else if (o->mode == AO_CW_USB) { o->freq_shift = (float)((double)o->if_center - (double)o->bw_cw / 2.0); tbl = ASDR_TBL_IF_CW; }
  else if (o->mode == AO_CW_LSB) { o->freq_shift = (float)((double)o->if_center + (double)o->bw_cw / 2.0); tbl = ASDR_TBL_IF_CW; }
  else if (o->mode == AO_AM || o->mode == AO_SAM) { o->freq_shift = o->if_center; tbl = ASDR_TBL_IF_AM; }
  if (tbl >= 0) { biquad4_init(&o->if_i, tbl); biquad4_init(&o->if_q, tbl); }
  return o->freq_shift;
}
float ao_getTuningOffset(const asdr_oracle_t *o) { return o->freq_shift; }
int16_t ao_getDemodMode(const asdr_oracle_t *o) { return (int16_t)o->mode; }

void ao_setInputGain(asdr_oracle_t *o, float g) { /* .cpp:232-238 */
  if (g > 10.0) g = 10.0f;
  if (g < 0.0) g = 0.0f;
  o->in_gain = g;
  o->in_gain_i = o->in_gain * o->gain_balance;
  o->in_gain_q = o->in_gain;
}
void ao_setIQgainBalance(asdr_oracle_t *o, float balance) { /* .cpp:240-244: local shadows the member */
  float gb = sqrtf(balance);
  o->in_gain_i = o->in_gain * gb;
  o->in_gain_q = o->in_gain / gb;
}
void ao_setOutputGain(asdr_oracle_t *o, float g) { o->output_gain = g; }
void ao_setMute(asdr_oracle_t *o, int muted) { /* .cpp:249-253 */
  o->muted = muted ? 1 : 0;
  o->current_out_gain = o->muted ? 0.0f : o->out_gain;
}
int ao_getMute(const asdr_oracle_t *o) { return o->muted; }

float ao_getBPFlower(const asdr_oracle_t *o) { /* .cpp:259-265 */
  if (o->mode == AO_USB || o->mode == AO_LSB) return (float)((double)o->if_center - (double)o->bw_ssb / 2.0);
  else if (o->mode == AO_CW_USB || o->mode == AO_CW_LSB) return (float)((double)o->if_center - (double)o->bw_cw / 2.0);
  else if (o->mode == AO_AM || o->mode == AO_SAM) return (float)((double)o->if_center - (double)o->bw_am / 2.0);
  else if (o->mode == AO_WSPR) return (float)((double)o->if_center - (double)o->bw_wspr / 2.0);
  return 0.0f;
}
float ao_getBPFupper(const asdr_oracle_t *o) { /* .cpp:267-273, including the `+-` at :271 */
  if (o->mode == AO_USB || o->mode == AO_LSB) return (float)((double)o->if_center + (double)o->bw_ssb / 2.0);
  else if (o->mode == AO_CW_USB || o->mode == AO_CW_LSB) return (float)((double)o->if_center + (double)o->bw_cw / 2.0);
  else if (o->mode == AO_AM || o->mode == AO_SAM) return (float)((double)o->if_center + (double)o->bw_am / 2.0);
  else if (o->mode == AO_WSPR) return (float)((double)o->if_center + -((double)o->bw_wspr / 2.0));
  return 0.0f;
}

void ao_enableAudioFilter(asdr_oracle_t *o) { o->audio_filter_enabled = 1; }
void ao_disableAudioFilter(asdr_oracle_t *o) { o->audio_filter_enabled = 0; }
int ao_getAudioFilter(const asdr_oracle_t *o) { return o->current_filter; }
void ao_setAudioFilter(asdr_oracle_t *o, int filter) { /* .cpp:298-311 */
  if (filter == AO_AUDIO_BYPASS) o->audio_filter_enabled = 0;
  else if (filter >= AO_AUDIO_AM && filter <= AO_AUDIO_3300) biquad4_init(&o->audio_bq, ASDR_TBL_AUDIO_BASE + filter);
  o->current_filter = (int16_t)filter;
}

void ao_disableALSfilter(asdr_oracle_t *o) { o->als_enabled = 0; }
void ao_setALSfilterNotch(asdr_oracle_t *o) { o->als_notch = 1; }
void ao_setALSfilterPeak(asdr_oracle_t *o) { o->als_notch = 0; }
void ao_setALSfilterAdaptive(asdr_oracle_t *o) { o->als_adaptive = 1; }
void ao_setALSfilterStatic(asdr_oracle_t *o) { o->als_adaptive = 0; }
int ao_ALSfilterIsEnabled(const asdr_oracle_t *o) { return o->als_enabled; }
int ao_ALSfilterIsNotch(const asdr_oracle_t *o) { return o->als_notch; }
int ao_ALSfilterIsPeak(const asdr_oracle_t *o) { return !o->als_notch; }
int ao_ALSfilterIsAdaptive(const asdr_oracle_t *o) { return o->als_adaptive; }
void ao_enableALSfilter(asdr_oracle_t *o) { /* .cpp:384-391 */
  o->als_enabled = 1;
  for (int i = 0; i < N; i++) { o->als_w[i] = 0.0f; o->als_in[i] = 0.0f; o->als_in[i + N] = 0.0f; }
}
void ao_setALSfilterParams(asdr_oracle_t *o, unsigned int m, float lambda, float delay) { /* .cpp:393-398 */
  o->als_m = (int16_t)m;
  if (o->als_m >= N) o->als_m = N;
  o->als_lambda = lambda;
  o->als_delay = (int16_t)delay;
}

float ao_getAMcarrierLevel(const asdr_oracle_t *o) { return o->agc_carrier; }
void ao_enableAGC(asdr_oracle_t *o) { o->agc_enabled = 1; }
void ao_disableAGC(asdr_oracle_t *o) { o->agc_enabled = 0; }
void ao_setAGCstaticGain(asdr_oracle_t *o, float g) { o->agc_static_gain = g; }
int ao_AGCisEnabled(const asdr_oracle_t *o) { return o->agc_enabled; }
int ao_AGCisActive(const asdr_oracle_t *o) { return o->agc_active; }
void ao_setAGCthreshold(asdr_oracle_t *o, float v) { o->agc_threshold = v; agc_build_table(o); }
void ao_setAGCslope(asdr_oracle_t *o, float v) { o->agc_slope = v; agc_build_table(o); }
void ao_setAGCkneeWidth(asdr_oracle_t *o, float v) { o->agc_knee = v; agc_build_table(o); }
void ao_setAGCattackTime(asdr_oracle_t *o, float ms) { /* .cpp:551-555 */
  o->agc_attack_ms = ms;
  o->agc_alpha_att = (float)exp(log(0.1) / ((double)(AO_FS * o->agc_attack_ms) / 1000.0));
  o->agc_beta_att = (float)(1.0 - (double)o->agc_alpha_att);
}
void ao_setAGCreleaseTime(asdr_oracle_t *o, float ms) { /* .cpp:557-561 */
  o->agc_release_ms = ms;
  o->agc_alpha_rel = (float)exp(log(0.1) / ((double)(AO_FS * o->agc_release_ms) / 1000.0));
  o->agc_beta_rel = (float)(1.0 - (double)o->agc_alpha_rel);
}
void ao_setAGChangTime(asdr_oracle_t *o, float ms) { /* .cpp:563-566: float product, then / 1000.0 */
  o->agc_table[129] = ms; /* _agc_hangTime shares storage with the table's 130th entry */
  o->agc_hang_count = (uint32_t)((double)(ms * AO_FS) / 1000.0);
}
void ao_setAGCmode(asdr_oracle_t *o, int mode) { /* .cpp:524-544 */
  mode = (int16_t)mode;
  if (mode == AO_AGC_OFF) ao_disableAGC(o);
  else if (mode == AO_AGC_FAST) { ao_setAGCattackTime(o, 2.0f); ao_setAGCreleaseTime(o, 100.0f); ao_setAGChangTime(o, 100.0f); ao_enableAGC(o); }
  else if (mode == AO_AGC_MEDIUM) { ao_setAGCattackTime(o, 5.0f); ao_setAGCreleaseTime(o, 250.0f); ao_setAGChangTime(o, 500.0f); ao_enableAGC(o); }
  else if (mode == AO_AGC_SLOW) { ao_setAGCattackTime(o, 10.0f); ao_setAGCreleaseTime(o, 500.0f); ao_setAGChangTime(o, 2000.0f); ao_enableAGC(o); }
}
float ao_getAGCthreshold(const asdr_oracle_t *o) { return o->agc_threshold; }
float ao_getAGCslope(const asdr_oracle_t *o) { return o->agc_slope; }
float ao_getAGCkneeWidth(const asdr_oracle_t *o) { return o->agc_knee; }
float ao_getAGCattack(const asdr_oracle_t *o) { return o->agc_attack_ms; }
float ao_getAGCrelease(const asdr_oracle_t *o) { return o->agc_release_ms; }
float ao_getAAGalphaAttack(const asdr_oracle_t *o) { return o->agc_alpha_att; }
float ao_getAGCbetaAttack(const asdr_oracle_t *o) { return o->agc_beta_att; }
float ao_getAGCalphaRelease(const asdr_oracle_t *o) { return o->agc_alpha_rel; }
float ao_getAGCbetaRelease(const asdr_oracle_t *o) { return o->agc_beta_rel; }
float ao_getAGClookup(const asdr_oracle_t *o, int i) { return (i >= 0 && i < 130) ? o->agc_table[i] : 0.0f; }
float ao_getAGCstaticGain(const asdr_oracle_t *o) { return o->agc_static_gain; }
uint32_t ao_getAGChangCount(const asdr_oracle_t *o) { return o->agc_hang_count; }
void ao_get_chain_constants(const asdr_oracle_t *o, float out[12]) {
  const float v[12] = {o->pll_b0, o->pll_b1, o->pll_a1, o->alpha_freq, o->beta_freq, o->f_conv, o->lock_lo, o->lock_hi,
                       o->two_pi_f, o->half_pi_f, o->two_pi_f / AO_FS, o->nb_beta};
  memcpy(out, v, sizeof v);
}

void ao_enableNoiseBlanker(asdr_oracle_t *o) { o->nb_enabled = 1; nb_reset(o); }
void ao_disableNoiseBlanker(asdr_oracle_t *o) { o->nb_enabled = 0; }
int ao_NoiseBlankerisEnabled(const asdr_oracle_t *o) { return o->nb_enabled; }
int ao_NoiseBlankerDetection(const asdr_oracle_t *o) { return o->nb_detected; }
void ao_setNoiseBlankerThreshold(asdr_oracle_t *o, float r) { o->nb_threshold = r; nb_reset(o); }
void ao_setNoiseBlankerThresholdDb(asdr_oracle_t *o, float db) { /* .cpp:671-674 */
  o->nb_threshold = powf(10.0f, (float)((double)db / 20.0));
  nb_reset(o);
}

float ao_getSAMfrequency(const asdr_oracle_t *o) { return o->pll_freq; }
int ao_getSAMphaseLockStatus(const asdr_oracle_t *o) { return o->pll_locked; }

/* ------------------------------------------------------------------------------------------ */
/* exhaustive checks of the exact-arithmetic shortcuts the HIP kernels use (test helpers)       */
/* ------------------------------------------------------------------------------------------ */
static double div_by_const_fma(double x, double c, double r) { /* Markstein: RN(x/c) from r = RN(1/c) */
  double q0 = x * r;
  double rem = fma(-q0, c, x);
  return fma(rem, r, q0);
}

/* For every float32 bit pattern in [bits_lo, bits_hi): does the reciprocal/fma quotient of
 * (double)phase*65535.0 by (double)(float)(2*pi) equal the true IEEE quotient (AudioSDR.h:364)?  Returns the
 * number of mismatching QUOTIENTS (stronger than equal truncations). */
uint64_t ao_check_sin_index_division(uint32_t bits_lo, uint32_t bits_hi) {
  const double c = (double)(float)(2.0 * AO_PI), r = 1.0 / c;
  uint64_t bad = 0;
  for (uint32_t b = bits_lo; b < bits_hi; b++) {
    union { uint32_t u; float f; } v; v.u = b;
    double x = (double)v.f * 65535.0;
    if (div_by_const_fma(x, c, r) != x / c) bad++;
  }
  return bad;
}

/* The kernels' ONE-multiply table phase (asdr_kernels.hip sin_index): (long)((double)phase * S) with S = RN(RN(65535 / c) * (1 + 2^-49))
 * against the reference's (long)((double)phase * 65535.0 / c) (AudioSDR.h:364), for every float32 bit pattern in [bits_lo, bits_hi).
 * Returns the number of phases whose truncations differ. */
uint64_t ao_check_sin_index_one_multiply(uint32_t bits_lo, uint32_t bits_hi) {
  const double c = (double)(float)(2.0 * AO_PI), S = (65535.0 / c) * (1.0 + 0x1p-49);
  uint64_t bad = 0;
  for (uint32_t b = bits_lo; b < bits_hi; b++) {
    union { uint32_t u; float f; } v; v.u = b;
    const double x = (double)v.f;
    if ((long)(x * S) != (long)(x * 65535.0 / c)) bad++;
  }
  return bad;
}

/* all int16: s/32767.0 (AudioSDR.cpp:68) */
int ao_check_scale_division(void) {
  int bad = 0;
  if (1.0 / 32767.0 != 0x1.0002000400080p-15) bad++;
  for (int s = -32768; s < 32768; s++) {
    if (div_by_const_fma((double)s, 32767.0, 1.0 / 32767.0) != (double)s / 32767.0) bad++;
    /* the kernels' two-operation form (asdr_kernels.hip div_i16_by_32767): fma(x, r, x * r * 2^-60) */
    if (fma((double)s, 0x1.0002000400080p-15, (double)s * 0x1.0002000400080p-75) != (double)s / 32767.0) bad++;
  }
  return bad;
}

/* all int16 at input gain 1.0 (the reference's default, AudioSDR.h:172-175): (float)(((float)s / 32767.0) * 1.0) -- the binary64 quotient
 * rounded to binary32 -- against the kernels' binary32 form fmaf(x, rh, x * rl) with rh = 0x1.0002p-15, rl = 0x1.0002p-45 (1/32767 is
 * the bit pattern 2^-15 repeated every 15 bits: rh holds two copies, rl the next two; asdr_kernels.hip scale8, unit-gain waves). */
int ao_check_scale_unit_gain(void) {
  int bad = 0;
  for (int s = -32768; s < 32768; s++) {
    const float want = (float)(((double)(float)s / 32767.0) * (double)1.0f);
    const float x = (float)s;
    const float got = fmaf(x, 0x1.0002p-15f, x * 0x1.0002p-45f);
    if (memcmp(&want, &got, 4) != 0) bad++;
  }
  return bad;
}

/* The kernels' PLL phase update fmaf(filt + prev, 0.5f, phase) against the reference's (float)((double)phase + (double)(filt +
 * prev) / 2.0) (AudioSDR.cpp:732): halving is exact in binary64, and rounding the binary64 sum of two binary32-precision values
 * to binary32 is innocuous double rounding (53 >= 2*24 + 2), so the single-rounding fma must agree for EVERY pair.  Checked on n
 * pseudo-random pairs of float bit patterns (all exponents, denormals and zeros included; NaN / inf skipped) plus near-tie pairs. */
uint64_t ao_check_pll_phase_update(uint32_t seed, uint64_t n) {
  uint64_t bad = 0, s = seed * 2654435761u + 12345u;
  for (uint64_t i = 0; i < n; i++) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    union { uint32_t u; float f; } a, b;
    a.u = (uint32_t)(s >> 32); b.u = (uint32_t)s;
    if ((i & 3) == 1) b.u = (a.u & 0x7F800000u) - ((uint32_t)(s >> 7) % 40u << 23) + (b.u & 0x807FFFFFu);   /* exponents 0..39 apart */
    if ((i & 3) == 2) { b.u = (b.u & 0x80000000u) | ((a.u & 0x7F800000u) - (24u << 23)) | ((i & 4) ? 0u : 1u); }   /* half-ulp ties */
    if (a.f != a.f || b.f != b.f || a.f - a.f != 0.0f || b.f - b.f != 0.0f) continue;
    const float t = b.f;                                   /* filt + prev_filt as the reference forms it (float) */
    const float want = (float)((double)a.f + (double)t / 2.0);
    const float got = fmaf(t, 0.5f, a.f);
    if (memcmp(&want, &got, 4) != 0) bad++;
  }
  return bad;
}

/* all 65,536 table phases: float32 interpolation == the reference's mixed float/double form (AudioSDR.h:369) */
int ao_check_sin_interp_f32(void) {
  int bad = 0;
  for (int ip = 0; ip < 65536; ip++) {
    uint16_t index = (uint16_t)(ip >> 8), delta = (uint16_t)(ip & 0xFF);
    float v1 = asdr_sine_table[index], v2 = asdr_sine_table[index + 1];
    float f32 = v1 + ((v2 - v1) * (float)delta) * (1.0f / 256.0f);
    float ref = ao_sin_from_index((uint16_t)ip);
    if (memcmp(&f32, &ref, 4) != 0) bad++;
  }
  return bad;
}

/* ------------------------------------------------------------------------------------------ */
/* CPU-baseline runner for bench.py                                                            */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
  int ch0, ch1, n_blocks;
  asdr_oracle_t **inst;
  const int16_t *I, *Q;
  int16_t *out;
} bench_job_t;

static void bench_configure(asdr_oracle_t *o, int config) {
  switch (config) {
    case 0: /* C2 of SURVEY.md 8d: USB, audio filter on (bw2700 from init), NB + AGC defaults */
      ao_setDemodMode(o, AO_USB); ao_enableAudioFilter(o); break;
    case 1: /* C1: AM defaults */
      ao_setDemodMode(o, AO_AM); break;
    case 2: /* C3: SAM, NB threshold 10 dB, audio AM filter on */
      ao_setDemodMode(o, AO_SAM); ao_setNoiseBlankerThresholdDb(o, 10.0f);
      ao_enableAudioFilter(o); ao_setAudioFilter(o, AO_AUDIO_AM); break;
    default: break;
  }
}

static void *bench_worker(void *arg) {
  bench_job_t *j = (bench_job_t *)arg;
  for (int c = j->ch0; c < j->ch1; c++) {
    asdr_oracle_t *o = j->inst[c];
    size_t base = (size_t)c * j->n_blocks * N;
    for (int b = 0; b < j->n_blocks; b++)
      ao_update(o, j->I + base + (size_t)b * N, j->Q + base + (size_t)b * N, j->out + base + (size_t)b * N);
  }
  return NULL;
}

/* Seconds spent in update() only: the instances are created and configured (AGC table build, ...) BEFORE the timed region
 * and destroyed after it. */
double ao_bench_run(int config, int n_channels, int n_blocks, const int16_t *I, const int16_t *Q, int16_t *out,
                    int n_threads) {
  if (n_threads < 1) n_threads = 1;
  if (n_threads > 256) n_threads = 256;
  pthread_t th[256];
  bench_job_t jobs[256];
  struct timespec t0, t1;
  asdr_oracle_t **inst = (asdr_oracle_t **)malloc((size_t)n_channels * sizeof *inst);
  for (int c = 0; c < n_channels; c++) { inst[c] = ao_create(); bench_configure(inst[c], config); }
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int t = 0; t < n_threads; t++) {
    jobs[t] = (bench_job_t){(int)((long)n_channels * t / n_threads), (int)((long)n_channels * (t + 1) / n_threads),
                            n_blocks, inst, I, Q, out};
    pthread_create(&th[t], NULL, bench_worker, &jobs[t]);
  }
  for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
  clock_gettime(CLOCK_MONOTONIC, &t1);
  for (int c = 0; c < n_channels; c++) ao_destroy(inst[c]);
  free(inst);
  return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* Host calibration for bench.py: a register-only float loop (no memory traffic, nothing of the oracle) on n_threads threads;
 * returns seconds.  (n_threads x t(1 thread)) / t(n_threads) = how many threads' worth of arithmetic the box really gives this
 * process -- sandboxes often expose more logical CPUs than they schedule. */
static void *spin_worker(void *arg) {
  long iters = *(long *)arg;
  volatile float sink;
  float a = 1.0f, b = 0.5f, c = 0.25f, d = 0.125f;
  for (long i = 0; i < iters; i++) { a = a * 0.999999f + 1e-7f; b = b * 0.999998f + 2e-7f; c = c * 0.999997f + 3e-7f; d = d * 0.999996f + 4e-7f; }
  sink = a + b + c + d; (void)sink;
  return NULL;
}
double ao_spin_calibrate(int n_threads, long iters) {
  if (n_threads < 1) n_threads = 1;
  if (n_threads > 256) n_threads = 256;
  pthread_t th[256];
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int t = 0; t < n_threads; t++) pthread_create(&th[t], NULL, spin_worker, &iters);
  for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
  clock_gettime(CLOCK_MONOTONIC, &t1);
  return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
