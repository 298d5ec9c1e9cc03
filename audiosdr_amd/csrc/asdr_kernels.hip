// asdr_kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4, wave64) for the batched
// AudioSDR::update() demodulation chain (reference: SRC/AudioSDRlib/AudioSDR.cpp:39-168).
//
// Execution shape
//   * one workgroup == one wavefront (64 lanes) == 8 channels ("slots" c8 = lane>>3, s8 = lane&7);
//     no inter-wave communication, no MFMA, no atomics, no collectives.
//   * pointwise / FIR stages: lane (c8, s8) owns 16 samples of its channel's 128-sample block
//     (contiguous 16 for load/scale/mix/output; 16 same-parity outputs for the Hilbert FIR).
//   * IIR biquad cascades: a 4-lane systolic pipeline per cascade (lane = stage, one sample of skew),
//     stage-to-stage hand-off by DPP row_shr:1, so 8 channels x {I,Q} x 4 stages fill the wave.
//   * strictly sequential scalar recurrences (noise-blanker average, mixer phase, AGC envelope, PLL,
//     AM carrier tracker): lane s8==0 of each channel, fed from / draining to LDS.
//   * every channel's block, filter scratch and FIR history are staged in LDS (2576 B per channel);
//     carried state lives in HBM in per-channel rows (asdr_device.h) and is loaded/stored with
//     coalesced 16-B-per-lane accesses.
//
// Numerics: built with -ffp-contract=off.  Every float operation is a separately rounded binary32
// operation in the reference's order; the reference's "double islands"  are computed
// in binary64 here too, so the int16 output is intended to be bit-identical to the CPU restatement.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "asdr_device.h"
#include "asdr_tables.h"
#include "../../include/asdr.h"

// ---- per-channel LDS layout (floats) --------------------------------------------------------------
#define CH_STRIDE 516   // 2064 B: 16-B slot stride 129 == 1 (mod 16) keeps the 8 channels' rows on different LDS slots
#define W0 0     // working row A: I, later the mono audio
#define W1 128   // working row B: Q; dead once the mixer has read it, then part of the Hilbert history below
#define HB 128   // Hilbert history, de-interleaved: X0 = odd samples [HB, HB+192), X1 = even samples (shifted by one) [HB+192, HB+384)
#define HX1 (HB + 192)
#define PH 384   // mixer phase sequence [PH, PH+128): inside X1, written/read before the history is assembled
#define SCR 512  // 4 per-channel scalar words (flags broadcast between a channel's lanes)
// noise-blanker overlay (before the rows above are live)
#define NB_MSK 0     // mask[0..265]
#define NB_C 272     // padded running detection count C[0..222] (int): C[3..23] = 0, C[24+t] = count after detection
                     // index t (t = i-78, 0..177), C[202..222] = final count.  Before the sequential pass the slots
                     // C[24+t] hold the envelope mag[t] (float), which the pass consumes chunk by chunk.
#define NB_MAG (NB_C + 24)
// AGC overlay
#define AGC_GV W1    // per sample: the envelope value whose compressor gain applies (-1 = gain carried in)
#define AGC_TAB 256  // this channel's gain table (row of 132 floats)
// ALS overlay
#define ALS_X 128         // [0..255] previous + current block
#define ALS_W 384         // [0..127] taps
#define ALS_OUT W0        // in place: the input was copied to ALS_X first

#define PI_D 3.1415926535897932384626433832795 /* Arduino.h PI (double) */

// Build-time ablation mask for profiling builds (DESIGN.md "ablation"); the shipped library uses 0.
#ifndef ASDR_ABLATE
#define ASDR_ABLATE 0
#endif
#define ABL_NB 1
#define ABL_IF 2
#define ABL_SAM 4
#define ABL_MIX 8
#define ABL_HIL 16
#define ABL_ENV 32
#define ABL_AF 64
#define ABL_AGC 128
#define ABL_ALS 256
#define ABL_ON(x) (!(ASDR_ABLATE & (x)))

// register budget: 2 -> <=256 VGPRs (8 waves/CU), 3 -> <=168 (LDS then allows 9 waves/CU)
#ifndef ASDR_WAVES_PER_EU
#define ASDR_WAVES_PER_EU 2
#endif

__constant__ float c_bq_pool[ASDR_N_BQ_TABLES][ASDR_BQ_COEFS];
__constant__ float c_hilbert[ASDR_HILBERT_TAPS];
__constant__ float c_sine[ASDR_SINE_TABLE_LEN];

extern "C" int asdr_kernels_upload_tables(void) {
  if (hipMemcpyToSymbol(HIP_SYMBOL(c_bq_pool), asdr_bq_pool, sizeof(asdr_bq_pool)) != hipSuccess) return -1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(c_hilbert), asdr_hilbert_taps, sizeof(asdr_hilbert_taps)) != hipSuccess) return -1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(c_sine), asdr_sine_table, sizeof(asdr_sine_table)) != hipSuccess) return -1;
  return 0;
}

// ---- scalar helpers (AudioSDR.h:358-446) ------------------------------------------------------------
// Correctly rounded binary64 quotient x / c for a constant c with r = RN(1/c) (Markstein): q0 = RN(x*r),
// rem = x - c*q0 exactly (fma), q = RN(q0 + rem*r) == RN(x/c).  Replaces the ~30-instruction IEEE f64 division
// sequence by mul + 2 fma.  Verified exhaustively against true division by the CPU test-suite (all float32
// phases in [0, 2*pi) for c = (double)(float)(2*pi); all int16 for c = 32767.0) and on the GPU by the taps.
__device__ __forceinline__ double div_by_const(double x, double c, double r) {
  const double q0 = x * r;
  const double rem = __builtin_fma(-q0, c, x);
  return __builtin_fma(rem, r, q0);
}
// sin_f32's phase -> uint16 table phase (AudioSDR.h:362-364): index logic, evaluated in binary64.
__device__ __forceinline__ uint32_t sin_index(float phase, float two_pi, double inv_two_pi) {
  if (phase >= two_pi) phase -= two_pi;
  if (phase < 0.0f) phase += two_pi;
  const double q = div_by_const((double)phase * 65535.0, (double)two_pi, inv_two_pi);
  return (uint32_t)(int)q & 0xFFFFu;
}
// AudioSDR.h:365-369: val1 + (((val2 - val1) * (float)delta) / 256.0).  The double divide-by-256 is exact and the
// double sum of two floats this close in exponent is exact, so the single final rounding equals the float32 sum
// (checked for all 65,536 table phases by the CPU test-suite).
__device__ __forceinline__ float sin_lut(const float *sine, uint32_t ip) {
  const uint32_t idx = ip >> 8, d = ip & 0xFFu;
  const float v1 = sine[idx], v2 = sine[idx + 1];
  return v1 + ((v2 - v1) * (float)d) * (1.0f / 256.0f);
}
__device__ __forceinline__ float sin_f32(const float *sine, float phase, float two_pi, double inv_two_pi) {
  return sin_lut(sine, sin_index(phase, two_pi, inv_two_pi));
}
// AudioSDR.h:375-377
__device__ __forceinline__ float cos_f32(const float *sine, float phase, float two_pi, double inv_two_pi) {
  return sin_f32(sine, (float)((double)phase + PI_D / 2.0), two_pi, inv_two_pi);
}
// AudioSDR.h:384-408
__device__ __forceinline__ float approx_atan(float z) {
  const float n1 = 0.97239411f, n2 = -0.19194795f;
  return (n1 + n2 * z * z) * z;
}
__device__ __forceinline__ float approx_atan2(float y, float x, float half_pi) {
  if (x != 0.0f) {
    if (fabsf(x) > fabsf(y)) {
      float z = y / x;
      if (x > 0.0f) return approx_atan(z);
      else if (y >= 0.0f) return (float)((double)approx_atan(z) + PI_D);
      else return (float)((double)approx_atan(z) - PI_D);
    } else {
      float z = x / y;
      if (y > 0.0f) return -approx_atan(z) + half_pi;
      else return -approx_atan(z) - half_pi;
    }
  } else {
    if (y > 0.0f) return half_pi;
    else if (y < 0.0f) return -half_pi;
  }
  return 0.0f;
}
// AudioSDR.h:434-446, n_iter = 1
__device__ __forceinline__ float fast_sqrt1(float x) {
  uint32_t i = __float_as_uint(x);
  i -= 1u << 23;
  i >>= 1;
  i += 1u << 29;
  float out = __uint_as_float(i);
  return 0.5f * (out + x / out);
}
// AudioSDR.cpp:483-494
__device__ __forceinline__ float agc_compress(const float *tab, float abs_val) {
  uint32_t input = (uint32_t)(int)((double)abs_val * 32767.0) & 0xFFFFu;
  uint32_t indx = input >> 8;
  if (indx > 127u) indx = 127u;
  float delta = (float)(input & 0xFFu) * (1.0f / 256.0f); /* float(frac)/256.0 is exact */
  float t0 = tab[indx], t1 = tab[indx + 1];
  return t0 + (t1 - t0) * delta;
}

__device__ __forceinline__ float dpp_row_shr1(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xF, 0xF, false));
}

// ---- 4-stage DF1 biquad cascade as a lane pipeline -------------------------------------------------
// Restates CMSIS-DSP arm_biquad_cascade_df1_f32 (arm_math.h:1360-1378; call sites AudioSDR.cpp:77-78,
// 136-137, 285): per stage, per sample  acc = b0*x; acc += b1*x1; acc += b2*x2; acc += a1*y1; acc += a2*y2
// (separately rounded, this association).  The reference runs stage-major over the block; each stage is causal,
// so running the four stages on four adjacent lanes with one 4-sample CHUNK of skew produces identical values.
// At chunk-step c lane `st` handles samples 4(c-st)..4(c-st)+3: stage 0 reads them from the LDS row (one
// ds_read_b128, prefetched a step ahead), stage k>0 takes the previous lane's four outputs of the previous step
// through DPP row_shr:1, stage 3 writes back in place (ds_write_b128).  The y-independent part
// p = (b0*x + b1*x1) + b2*x2 of all four samples is off the critical path; the recurrence is 3 dependent ops/sample.
__device__ __forceinline__ void biquad_pipe(float *row, bool on, int st, const float *cf, float *sv) {
  const float b0 = cf[0], b1 = cf[1], b2 = cf[2], a1 = cf[3], a2 = cf[4];
  float x1 = sv[0], x2 = sv[1], y1 = sv[2], y2 = sv[3];
  float yo0 = 0.0f, yo1 = 0.0f, yo2 = 0.0f, yo3 = 0.0f;
  const float4 *row4 = reinterpret_cast<const float4 *>(row);
  float4 xn = row4[0];
#pragma unroll 1
  for (int c = 0; c < ASDR_N / 4 + 3; ++c) {
    const int cn = c - st;
    const float4 xl = xn;
    xn = row4[(c + 1 < ASDR_N / 4) ? c + 1 : ASDR_N / 4 - 1];   // prefetch the next chunk for stage 0
    const float d0 = dpp_row_shr1(yo0), d1 = dpp_row_shr1(yo1), d2 = dpp_row_shr1(yo2), d3 = dpp_row_shr1(yo3);
    const bool s0 = (st == 0);
    const float xa = s0 ? xl.x : d0, xb = s0 ? xl.y : d1, xc = s0 ? xl.z : d2, xd = s0 ? xl.w : d3;
    // y-independent partial sums, in the reference's association
    float pa = b0 * xa; pa += b1 * x1; pa += b2 * x2;
    float pb = b0 * xb; pb += b1 * xa; pb += b2 * x1;
    float pc = b0 * xc; pc += b1 * xb; pc += b2 * xa;
    float pd = b0 * xd; pd += b1 * xc; pd += b2 * xb;
    // recurrence
    float ya = pa + a1 * y1; ya += a2 * y2;
    float yb = pb + a1 * ya; yb += a2 * y1;
    float yc = pc + a1 * yb; yc += a2 * ya;
    float yd = pd + a1 * yc; yd += a2 * yb;
    const bool act = (cn >= 0) && (cn < ASDR_N / 4);
    if (act) { x1 = xd; x2 = xc; y1 = yd; y2 = yc; }
    yo0 = ya; yo1 = yb; yo2 = yc; yo3 = yd;
    if (on && act && st == 3) reinterpret_cast<float4 *>(row)[cn] = make_float4(ya, yb, yc, yd);
  }
  sv[0] = x1; sv[1] = x2; sv[2] = y1; sv[3] = y2;
}

__device__ __forceinline__ void load16(const float *p, float *v) {
  const float4 *q = reinterpret_cast<const float4 *>(p);
#pragma unroll
  for (int r = 0; r < 4; ++r) { float4 t = q[r]; v[4 * r] = t.x; v[4 * r + 1] = t.y; v[4 * r + 2] = t.z; v[4 * r + 3] = t.w; }
}
__device__ __forceinline__ void store16(float *p, const float *v) {
  float4 *q = reinterpret_cast<float4 *>(p);
#pragma unroll
  for (int r = 0; r < 4; ++r) q[r] = make_float4(v[4 * r], v[4 * r + 1], v[4 * r + 2], v[4 * r + 3]);
}

// keeps the instruction scheduler from interleaving all iterations of a fully unrolled loop (register pressure)
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#define WAVE_SYNC() __syncthreads() /* workgroup == one wave: lowers to a wave barrier + LDS/VMEM waits */

// =====================================================================================================
extern "C" __global__ __launch_bounds__(64, ASDR_WAVES_PER_EU) void asdr_update_kernel(UpdateArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[8 * CH_STRIDE + 264];
  const int lane = threadIdx.x, c8 = lane >> 3, s8_ = lane & 7;
  float *sine = lds + 8 * CH_STRIDE;
  for (int i = lane; i < ASDR_SINE_TABLE_LEN; i += 64) sine[i] = c_sine[i];

  const int ch_ = a.sched[blockIdx.x * 8 + c8];
  const bool valid = ch_ < a.n_channels;
  const int loff_ = c8 * CH_STRIDE;
  const ChanParams *Pp_ = a.params + ch_;
#define P (*Pp)
  const ChainConsts K = a.k;

  const uint32_t mode = Pp_->mode;
  const uint32_t pflags = Pp_->flags;
  const bool is_ssb = (mode == ASDR_USBmode) || (mode == ASDR_LSBmode) || (mode == ASDR_CW_USBmode) ||
                      (mode == ASDR_CW_LSBmode) || (mode == ASDR_WSPRmode);
  const bool is_am = (mode == ASDR_AMmode), is_sam = (mode == ASDR_SAMmode);
  const bool sub_q = (mode == ASDR_USBmode) || (mode == ASDR_CW_USBmode) || (mode == ASDR_WSPRmode);
  const bool nb_en = pflags & ASDR_F_NB_EN, af_en = pflags & ASDR_F_AF_EN, agc_en = pflags & ASDR_F_AGC_EN;
  const bool als_en = pflags & ASDR_F_ALS_EN, muted = pflags & ASDR_F_MUTED;
  const float two_pi = K.two_pi_f;
  WAVE_SYNC();

#pragma unroll 1
  for (int blk = 0; blk < a.n_blocks; ++blk) {
    // Per-iteration opaque copies of the lane coordinates: stops LICM from hoisting every per-lane address
    // of the (long) block body out of this loop, which would pin >100 VGPRs.
    int s8 = s8_; asm volatile("" : "+v"(s8));
    int loff = loff_; asm volatile("" : "+v"(loff));
    int ch = ch_; asm volatile("" : "+v"(ch));
    const ChanParams *Pp = a.params + ch;
    ChanSmall *S = a.small + ch;
    float *L = lds + loff;
    int *Li = reinterpret_cast<int *>(L);
    const int k0 = 16 * s8;
    const bool lead = (s8 == 0);
    const bool tap_on = (a.taps != nullptr) && valid && (blk == a.n_blocks - 1);
    float *tap_base = tap_on ? a.taps + (size_t)ch * ASDR_N + k0 : nullptr;
    const size_t tap_stride = (size_t)a.n_channels * ASDR_N;
#define TAP_REGS(id, v) do { if (tap_on) store16(tap_base + (size_t)(id) * tap_stride, v); } while (0)
#define TAP_ROW(id, rowoff) do { if (tap_on) { float tv_[16]; load16(L + (rowoff) + k0, tv_); store16(tap_base + (size_t)(id) * tap_stride, tv_); } } while (0)

    // ---- issue every load whose address is known now (per-channel scalars, raw input, blanker ring) before any use
    uint32_t status = S->status;
    const uint32_t ns = S->nb_slot % 3u, hs = S->hil_slot & 1u;   // oldest NB ring slot (of 3), Hilbert ring parity
    float carrier_now = 0.0f;   // set by the envelope path when it runs in this block
    bool carrier_fresh = false;
    const size_t io = ((size_t)ch * a.n_blocks + blk) * ASDR_N + k0;
    union { int4 v[2]; int16_t s[16]; } ri, rq;
    ri.v[0] = ri.v[1] = rq.v[0] = rq.v[1] = make_int4(0, 0, 0, 0);
    if (valid) {
      const int4 *pi = reinterpret_cast<const int4 *>(a.in_i + io);
      const int4 *pq = reinterpret_cast<const int4 *>(a.in_q + io);
      ri.v[0] = pi[0]; ri.v[1] = pi[1]; rq.v[0] = pq[0]; rq.v[1] = pq[1];
    }
    const bool nb_run = ABL_ON(ABL_NB) && nb_en;
    const bool nb_wave = ABL_ON(ABL_NB) && __any(nb_en);   // wave-uniform: at least one of the 8 channels has the blanker on
    float *hist = a.nb_hist + (size_t)ch * 768;
    const uint32_t ns_mid = (ns + 1u) % 3u, ns_new = (ns + 2u) % 3u;
    float *mrow = a.nb_mask + (size_t)ch * ASDR_NB_MASK_ROW;
    // ---- input scale, AudioSDR.cpp:67-70: ((float)s / 32767.0) * gain in binary64, stored float --------
    float xi[16], xq[16];
    {
      const double gi = (double)P.in_gain_i, gq = (double)P.in_gain_q;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        xi[j] = (float)(div_by_const((double)ri.s[j], 32767.0, 1.0 / 32767.0) * gi);
        xq[j] = (float)(div_by_const((double)rq.s[j], 32767.0, 1.0 / 32767.0) * gq);
        if ((j & 1) == 1) SCHED_FENCE();
      }
    }
    TAP_REGS(ASDR_TAP_SCALED_I, xi); TAP_REGS(ASDR_TAP_SCALED_Q, xq);
    float oi[16], oq[16], mi[16], mq[16];
    if (nb_run) {
      const float *old_i = hist + ns * 256 + k0, *mid_i = hist + ns_mid * 256 + k0;
      load16(mid_i, mi); load16(mid_i + 128, mq);
      if (s8 >= 4) { load16(old_i, oi); load16(old_i + 128, oq); }   // only samples 78..127 of the oldest block are re-scanned
    }
    if (nb_wave) {
      // Third ring slot: the newest block is parked in HBM, so no registers are held across the blanker's phases.
      // Channels whose blanker is OFF use their (otherwise dead: enabling the blanker always resets it,
      // AudioSDR.cpp:653-656) ring slot as the same parking space and read it back with mask 1.0.
      float *new_i = hist + ns_new * 256 + k0;
      store16(new_i, xi); store16(new_i + 128, xq);
    }

    // ---- impulse noise blanker, AudioSDR.cpp:606-650 ------------------------------------------------------
    // Buffer coordinates as in the reference: [0,128) oldest, [128,256) middle, [256,384) newest.  The ring
    // in HBM holds oldest+middle; the newest block is only stored.  Output = mask x oldest (2 blocks late).
    if (nb_wave) {
      if (nb_en) {
        // envelope for detection indices i = 78..255 -> t = i - 78 (fast_sqrt_f32(I^2+Q^2, 1), :628)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          L[NB_MAG + 50 + k0 + j] = fast_sqrt1(mi[j] * mi[j] + mq[j] * mq[j]);
          const int k = k0 + j;
          if (k >= 78) L[NB_MAG + k - 78] = fast_sqrt1(oi[j] * oi[j] + oq[j] * oq[j]);
          if ((j & 3) == 3) SCHED_FENCE();
        }
        // mask: carried part = previous mask[128..265] (row of 144, 138 used); then the rest of the (new) newest
        // block is 1.0 (:621-623) -- written AFTER the row so that it wins on entries 138..143
        float4 mk4[5];
#pragma unroll
        for (int r = 0; r < 5; ++r) { const int q = s8 + 8 * r; mk4[r] = (q < 36) ? reinterpret_cast<const float4 *>(mrow)[q] : make_float4(1.f, 1.f, 1.f, 1.f); }
#pragma unroll
        for (int r = 0; r < 5; ++r) { const int q = s8 + 8 * r; if (q < 36) *reinterpret_cast<float4 *>(L + NB_MSK + 4 * q) = mk4[r]; }
#pragma unroll
        for (int j = 0; j < 16; ++j) L[NB_MSK + ASDR_NB_MASK_USED + k0 + j] = 1.0f;
      }
      WAVE_SYNC();
      if (nb_en && lead) {   // sequential: threshold test against the running average (:627-635), 8 samples per trip
        float avg = S->nb_avg;
        const float thr = P.nb_threshold;
        int cnt = 0;
        for (int z = 3; z < 24; ++z) Li[NB_C + z] = 0;
#pragma unroll 1
        for (int t = 0; t < 176; t += 8) {
          float m[8]; int cv[8];
          const float4 ma = *reinterpret_cast<const float4 *>(L + NB_MAG + t), mb = *reinterpret_cast<const float4 *>(L + NB_MAG + t + 4);
          m[0] = ma.x; m[1] = ma.y; m[2] = ma.z; m[3] = ma.w; m[4] = mb.x; m[5] = mb.y; m[6] = mb.z; m[7] = mb.w;
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            cnt += (m[u] > avg * thr) ? 1 : 0;
            cv[u] = cnt;
            avg = K.nb_alpha * avg + K.nb_beta * m[u];
          }
          *reinterpret_cast<int4 *>(Li + NB_MAG + t) = make_int4(cv[0], cv[1], cv[2], cv[3]);
          *reinterpret_cast<int4 *>(Li + NB_MAG + t + 4) = make_int4(cv[4], cv[5], cv[6], cv[7]);
        }
        for (int t = 176; t < 178; ++t) {
          const float m = L[NB_MAG + t];
          cnt += (m > avg * thr) ? 1 : 0;
          Li[NB_MAG + t] = cnt;
          avg = K.nb_alpha * avg + K.nb_beta * m;
        }
        for (int z = 202; z < 223; ++z) Li[NB_C + z] = cnt;
        S->nb_avg = avg;
        S->nb_slot = ns_mid;
        status = (status & ~ASDR_S_NB_DETECTED) | (cnt > 0 ? ASDR_S_NB_DETECTED : 0u);
      }
      WAVE_SYNC();
      if (nb_en) {   // zero mask[i-10 .. i+10] around every detection (:630); all writes are 0.0, so order-free:
                     // mask[m] is hit iff a detection index lies in [m-10, m+10] iff C[m-44] - C[m-65] > 0
        const int m0 = 68 + s8 * 25;
#pragma unroll
        for (int g5 = 0; g5 < 25; g5 += 5) {
          int hi[5], lo[5];
#pragma unroll
          for (int r = 0; r < 5; ++r) { hi[r] = Li[NB_C + m0 - 44 + g5 + r]; lo[r] = Li[NB_C + m0 - 65 + g5 + r]; }
#pragma unroll
          for (int r = 0; r < 5; ++r) if (m0 + g5 + r <= 265 && hi[r] - lo[r] > 0) L[NB_MSK + m0 + g5 + r] = 0.0f;
          SCHED_FENCE();
        }
      }
      WAVE_SYNC();
      float ev[17];
      if (nb_en) {   // trailing-edge ramp (:637-644; only the first branch is reachable).  An edge at i
                     // writes mask[i-7..i-1] only, which later iterations never read: read all, then write.
#pragma unroll
        for (int j = 0; j < 17; ++j) ev[j] = L[NB_MSK + 127 + k0 + j];
      }
      WAVE_SYNC();
      if (nb_en) {
        const float trans_dn[7] = {(float)0.933, (float)0.750, (float)0.500, (float)0.250, (float)0.067, 0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          if (ev[j + 1] == 1.0f && ev[j] == 0.0f) {
            const int i = 128 + k0 + j;
#pragma unroll
            for (int q = 0; q < 7; ++q) L[NB_MSK + i - 7 + q] = trans_dn[q];
          }
        }
      }
      WAVE_SYNC();
      { const float *old_i = hist + (nb_en ? ns : ns_new) * 256 + k0; load16(old_i, oi); load16(old_i + 128, oq); }
      {              // output = mask x oldest block (:646-649); carry mask[128..265(..271)] to the next call
        float mk[16];
        load16(L + NB_MSK + k0, mk);
#pragma unroll
        for (int j = 0; j < 16; ++j) { const float mv = nb_en ? mk[j] : 1.0f; xi[j] = mv * oi[j]; xq[j] = mv * oq[j]; }
      }
      if (nb_en) {
#pragma unroll
        for (int r = 0; r < 5; ++r) { const int q = s8 + 8 * r; if (q < 36) reinterpret_cast<float4 *>(mrow)[q] = *reinterpret_cast<const float4 *>(L + NB_MSK + 128 + 4 * q); }
      }
      WAVE_SYNC();
    }
    TAP_REGS(ASDR_TAP_NB_I, xi); TAP_REGS(ASDR_TAP_NB_Q, xq);

    // ---- IF band-pass, AudioSDR.cpp:77-78: 2 x 4-stage cascade, 64 lanes = 8 ch x {I,Q} x 4 stages -------
    store16(L + W0 + k0, xi); store16(L + W1 + k0, xq);
    WAVE_SYNC();
    if (ABL_ON(ABL_IF)) {
      const int iq = s8 >> 2, st = s8 & 3;
      float sv[4];
      const float4 s4 = *reinterpret_cast<const float4 *>(&S->if_state[iq][4 * st]);
      sv[0] = s4.x; sv[1] = s4.y; sv[2] = s4.z; sv[3] = s4.w;
      biquad_pipe(L + (iq ? W1 : W0), true, st, &c_bq_pool[P.if_table][5 * st], sv);
      *reinterpret_cast<float4 *>(&S->if_state[iq][4 * st]) = make_float4(sv[0], sv[1], sv[2], sv[3]);
    }
    WAVE_SYNC();
    TAP_ROW(ASDR_TAP_IF_I, W0); TAP_ROW(ASDR_TAP_IF_Q, W1);

    // ---- SAM: quadrature PLL, AudioSDR.cpp:688-749 (sequential per channel) ---------------------------------
    bool pll_locked = false;
    if (ABL_ON(ABL_SAM) && __any(is_sam)) {
      if (is_sam && lead) {
        float y_re = S->pll_y_re, y_im = S->pll_y_im, prev_filt = S->pll_prev_filt;
        float d0 = S->pll_d0, d1 = S->pll_d1, phase_est = S->pll_phase_est, pfreq = S->pll_freq;
        bool locked = false;
#pragma unroll 1
        for (int i = 0; i < ASDR_N; ++i) {
          const float x_re = L[W0 + i], x_im = L[W1 + i];
          const float d_re = x_re * y_re + x_im * y_im;
          const float d_im = x_im * y_re - x_re * y_im;
          const float err = approx_atan2(d_im, d_re, K.half_pi_f);
          d1 = d0;
          d0 = err - K.pll_a1 * d1;
          const float filt = K.pll_b0 * d0 + K.pll_b1 * d1;
          phase_est = (float)((double)phase_est + (double)(filt + prev_filt) / 2.0);
          prev_filt = filt;
          while ((double)phase_est >= PI_D) phase_est -= two_pi;
          while ((double)phase_est < -PI_D) phase_est += two_pi;
          y_re = cos_f32(sine, phase_est, two_pi, K.inv_two_pi_d);
          y_im = sin_f32(sine, phase_est, two_pi, K.inv_two_pi_d);
          pfreq = K.pll_alpha_freq * pfreq + K.pll_beta_freq * (filt * K.pll_f_conv);
          locked = (pfreq > K.pll_lock_lo) && (pfreq < K.pll_lock_hi);
          if (locked) {
            L[W0 + i] = x_re * y_re + x_im * y_im;
            L[W1 + i] = -x_re * y_im + x_im * y_re;
          }
        }
        S->pll_y_re = y_re; S->pll_y_im = y_im; S->pll_prev_filt = prev_filt;
        S->pll_d0 = d0; S->pll_d1 = d1; S->pll_phase_est = phase_est; S->pll_freq = pfreq;
        status = (status & ~ASDR_S_PLL_LOCKED) | (locked ? ASDR_S_PLL_LOCKED : 0u);
        Li[SCR + 0] = locked ? 1 : 0;
      }
      WAVE_SYNC();
      if (is_sam) pll_locked = Li[SCR + 0] != 0;
    }
    // envelope detector runs for AM, and for SAM when the PLL is unlocked at the end of the block (:132)
    const bool do_env = is_am || (is_sam && !pll_locked);
    const bool do_mix = is_ssb || do_env;

    // ---- mixer phase sequence, AudioSDR.h:508-526 (phase accumulates sequentially in float) -----------------
    if (do_mix && lead) {
      float phase = is_ssb ? S->phase_ssb : S->phase_am;
      const float fs = is_ssb ? -P.freq_shift : -K.if_center;
      const float inc = fs * K.phase_inc_unit;
#pragma unroll 1
      for (int i = 0; i < ASDR_N; i += 4) {
        float pv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          pv[u] = phase;
          const float t = phase + inc, t_dn = t - two_pi, t_up = t + two_pi;
          phase = (t > two_pi) ? t_dn : ((t < 0.0f) ? t_up : t);   // if (> twoPI) -= twoPI; else if (< 0) += twoPI
        }
        *reinterpret_cast<float4 *>(L + PH + i) = make_float4(pv[0], pv[1], pv[2], pv[3]);
      }
      if (is_ssb) S->phase_ssb = phase; else S->phase_am = phase;
    }
    WAVE_SYNC();
    float *hi_ring = a.hil_i + (size_t)ch * 256 + k0;   // 2-slot ring of mixed I blocks: slot hs = this block, hs^1 = previous
    float *hq = a.hil_q + (size_t)ch * 256;
    float i_del[16];
    float mi_[16], mq_[16];   // mixed (shifted) I, Q of this lane's 16 samples
    if (ABL_ON(ABL_MIX) && do_mix) {
      float ph[16], vi[16], vq[16];
      load16(L + PH + k0, ph); load16(L + W0 + k0, vi); load16(L + W1 + k0, vq);
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float c = cos_f32(sine, ph[j], two_pi, K.inv_two_pi_d), s = sin_f32(sine, ph[j], two_pi, K.inv_two_pi_d);
        mi_[j] = vi[j] * c - vq[j] * s;
        mq_[j] = vq[j] * c + vi[j] * s;
        if ((j & 1) == 1) SCHED_FENCE();
      }
    }
    WAVE_SYNC();   // all reads of the phase row done before the history overlay is written

    // ---- SSB/CW/WSPR: 257-tap folded Hilbert on Q, I delayed 128, AudioSDR.cpp:89-118 ----------------------
    if (is_ssb) {
      float q_old[16], q_mid[16];
      load16(hq + hs * 128 + k0, q_old);          // two blocks back
      load16(hq + (hs ^ 1u) * 128 + k0, q_mid);   // previous block
      store16(hi_ring + hs * 128, mi_);
      store16(hq + hs * 128 + k0, mq_);           // newest replaces oldest
      // history sample m = B + k0 + j (B = 0, 128, 256): odd m -> X0[(m-1)/2], even m -> X1[(m-2)/2]
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int m = k0 + j;
        if (j & 1) {
          L[HB + (m - 1) / 2] = q_old[j]; L[HB + (128 + m - 1) / 2] = q_mid[j]; L[HB + (256 + m - 1) / 2] = mq_[j];
        } else {
          if (m >= 2) L[HX1 + (m - 2) / 2] = q_old[j];
          L[HX1 + (128 + m - 2) / 2] = q_mid[j]; L[HX1 + (256 + m - 2) / 2] = mq_[j];
        }
      }
      if (lead) S->hil_slot = hs ^ 1u;
    }
    WAVE_SYNC();
    if (ABL_ON(ABL_HIL) && is_ssb) {
      // lane (par, g): outputs i = 2*(16g + j) + par, j = 0..15:
      //   Q[i] = sum_k h[k] * (X[127 + 16g + j - k] - X[16g + j + k]),  k ascending, accumulate from 0.0
      const int par = s8 & 1, g = s8 >> 1;
      const float *X = L + (par ? HX1 : HB);
      float acc[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[j] = 0.0f;
#pragma unroll 1
      for (int kc = 0; kc < 4; ++kc) {
        float dw[32], uw[32];
        const float *dp = X + 112 + 16 * (g - kc), *up = X + 16 * (g + kc);
        load16(dp, dw); load16(dp + 16, dw + 16);
        load16(up, uw); load16(up + 16, uw + 16);
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
          const float h = c_hilbert[16 * kc + kk];
#pragma unroll
          for (int j = 0; j < 16; ++j) acc[j] += h * (dw[15 + j - kk] - uw[j + kk]);
          SCHED_FENCE();
        }
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) L[W0 + 2 * (16 * g + j) + par] = acc[j];
    }
    WAVE_SYNC();
    if (is_ssb) {
      float qh[16], au[16];
      load16(hi_ring + (hs ^ 1u) * 128, i_del);   // previous block's mixed I == bufferI[3n+i-128] (:111)
      load16(L + W0 + k0, qh);
#pragma unroll
      for (int j = 0; j < 16; ++j) au[j] = sub_q ? (i_del[j] - qh[j]) : (i_del[j] + qh[j]);
      store16(L + W0 + k0, au);
      TAP_REGS(ASDR_TAP_MIX_I, i_del); TAP_REGS(ASDR_TAP_MIX_Q, qh);
    }

    // ---- AM envelope path, AudioSDR.cpp:132-143 -------------------------------------------------------------
    if (ABL_ON(ABL_ENV) && __any(do_env)) {
      if (do_env) { store16(L + W0 + k0, mi_); store16(L + W1 + k0, mq_); }
      WAVE_SYNC();
      {
        const int iq = s8 >> 2, st = s8 & 3;
        float sv[4];
        const float4 s4 = *reinterpret_cast<const float4 *>(&S->img_state[iq][4 * st]);
        sv[0] = s4.x; sv[1] = s4.y; sv[2] = s4.z; sv[3] = s4.w;
        biquad_pipe(L + (iq ? W1 : W0), do_env, st, &c_bq_pool[ASDR_TBL_AM_IMAGE][5 * st], sv);
        if (do_env) *reinterpret_cast<float4 *>(&S->img_state[iq][4 * st]) = make_float4(sv[0], sv[1], sv[2], sv[3]);
      }
      WAVE_SYNC();
      if (do_env) {
        float vi[16], vq[16], au[16];
        load16(L + W0 + k0, vi); load16(L + W1 + k0, vq);
        TAP_REGS(ASDR_TAP_MIX_I, vi); TAP_REGS(ASDR_TAP_MIX_Q, vq);
#pragma unroll
        for (int j = 0; j < 16; ++j) { au[j] = sqrtf(vi[j] * vi[j] + vq[j] * vq[j]); if ((j & 3) == 3) SCHED_FENCE(); }
        store16(L + W0 + k0, au);
      }
      WAVE_SYNC();
      if (do_env && lead) {   // carrier level tracker in binary64, stored float each sample (:141)
        float lvl = S->am_carrier;
#pragma unroll 4
        for (int i = 0; i < ASDR_N; ++i) lvl = (float)(.995 * (double)lvl + 0.005 * (double)fabsf(L[W0 + i]));
        S->am_carrier = lvl; carrier_now = lvl; carrier_fresh = true;
      }
    }
    if (is_sam && pll_locked) {   // audio = rotated Q (:126-128)
      float vi[16], vq[16];
      load16(L + W0 + k0, vi); load16(L + W1 + k0, vq);
      TAP_REGS(ASDR_TAP_MIX_I, vi); TAP_REGS(ASDR_TAP_MIX_Q, vq);
      store16(L + W0 + k0, vq);
    }
    if (!is_ssb && !is_am && !is_sam) {   // unknown mode: the reference re-processes stale audio; we emit silence
      float z[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) z[j] = 0.0f;
      store16(L + W0 + k0, z);
    }
    WAVE_SYNC();
    TAP_ROW(ASDR_TAP_DEMOD, W0);

    // AGC inputs are requested here, one phase early (latency hides behind the audio-filter pipeline)
    float agc_gain_in = 0.f, agc_old0 = 0.f, agc_carrier0 = 0.f;
    uint32_t agc_hc0 = 0u;
    if (ABL_ON(ABL_AGC) && agc_en) {
      agc_gain_in = S->agc_gain; agc_old0 = S->agc_old_abs; agc_hc0 = S->agc_hang_counter; agc_carrier0 = S->am_carrier;
    }
    // ---- audio IIR filter, AudioSDR.cpp:149, 280-286: lanes s8 = 0..3 are the four stages ---------------------
    if (ABL_ON(ABL_AF) && __any(af_en)) {
      const int st = s8 & 3;
      const bool on = af_en && (s8 < 4);
      float sv[4];
      const float4 s4 = *reinterpret_cast<const float4 *>(&S->af_state[4 * st]);
      sv[0] = s4.x; sv[1] = s4.y; sv[2] = s4.z; sv[3] = s4.w;
      biquad_pipe(L + W0, on, st, &c_bq_pool[P.audio_table][5 * st], sv);
      if (on) *reinterpret_cast<float4 *>(&S->af_state[4 * st]) = make_float4(sv[0], sv[1], sv[2], sv[3]);
      WAVE_SYNC();
    }
    TAP_ROW(ASDR_TAP_AUDIO_FILT, W0);

    // ---- AGC, AudioSDR.cpp:404-436 --------------------------------------------------------------------------
    // Split into (a) the sequential envelope/hang recurrence, which records for every sample which update
    // governs the gain, and (b) a parallel pass that evaluates the static compressor and applies the gain.
    if (ABL_ON(ABL_AGC) && __any(agc_en)) {
      const float *tab = L + AGC_TAB;
      const float gain_in = agc_gain_in;
      if (agc_en) {   // stage the channel's gain table (row of 132 floats) in LDS, all loads in flight at once
        const float *gtab = a.agc_tab + (size_t)P.agc_table * ASDR_AGC_TAB_ROW;
        float4 agc_t4[5];
#pragma unroll
        for (int r = 0; r < 5; ++r) { const int q = s8 + 8 * r; agc_t4[r] = (q < 33) ? reinterpret_cast<const float4 *>(gtab)[q] : make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
        for (int r = 0; r < 5; ++r) { const int q = s8 + 8 * r; if (q < 33) *reinterpret_cast<float4 *>(L + AGC_TAB + 4 * q) = agc_t4[r]; }
      }
      WAVE_SYNC();
      if (agc_en && lead) {
        float old_abs = agc_old0;
        uint32_t hc = agc_hc0;
        const float am_level = (float)(2.0 * (double)(carrier_fresh ? carrier_now : agc_carrier0));
        const float al_a = P.agc_alpha_att, be_a = P.agc_beta_att, al_r = P.agc_alpha_rel, be_r = P.agc_beta_rel;
        const uint32_t hang = P.agc_hang_count;
        float gv = -1.0f;   // envelope value governing the current gain; -1 = no update yet in this block
#pragma unroll 1
        for (int i = 0; i < ASDR_N; i += 8) {
          float x[8], gvv[8];
          const float4 xa = *reinterpret_cast<const float4 *>(L + W0 + i), xb = *reinterpret_cast<const float4 *>(L + W0 + i + 4);
          x[0] = xa.x; x[1] = xa.y; x[2] = xa.z; x[3] = xa.w; x[4] = xb.x; x[5] = xb.y; x[6] = xb.z; x[7] = xb.w;
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            float av = is_am ? am_level : fabsf(x[u]);
            av = (av > 1.0f) ? 1.0f : av;                  // if (absVal > 1.0) absVal = 1.0 (NaN-preserving like the reference)
            const bool att = av > old_abs;                 // attack
            const bool idle = (hc == 0u);                  // not hanging: release when not attacking
            const float v_att = al_a * old_abs + be_a * av;
            const float v_rel = al_r * old_abs + be_r * av;
            const float v_new = att ? v_att : v_rel;
            const bool upd = att || idle;
            old_abs = upd ? v_new : old_abs;
            gv = upd ? v_new : gv;
            hc = att ? hang : (idle ? 0u : hc - 1u);
            gvv[u] = gv;
          }
          *reinterpret_cast<float4 *>(L + AGC_GV + i) = make_float4(gvv[0], gvv[1], gvv[2], gvv[3]);
          *reinterpret_cast<float4 *>(L + AGC_GV + i + 4) = make_float4(gvv[4], gvv[5], gvv[6], gvv[7]);
        }
        S->agc_old_abs = old_abs;
        S->agc_hang_counter = hc;
        const float g_end = (gv < 0.0f) ? gain_in : agc_compress(tab, gv);
        S->agc_gain = g_end;
        status = (status & ~ASDR_S_AGC_ACTIVE) | (((double)g_end < 0.99) ? ASDR_S_AGC_ACTIVE : 0u);
      }
      WAVE_SYNC();
      if (agc_en) {
        float au[16], gvr[16];
        load16(L + W0 + k0, au); load16(L + AGC_GV + k0, gvr);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const float gain = (gvr[j] < 0.0f) ? gain_in : agc_compress(tab, gvr[j]);
          float o = gain * P.agc_static_gain * au[j];
          o = (o > 1.0f) ? 1.0f : o;
          o = (o < -1.0f) ? -1.0f : o;
          au[j] = o;
          if ((j & 3) == 3) SCHED_FENCE();
        }
        store16(L + W0 + k0, au);
      }
      WAVE_SYNC();
    }
    TAP_ROW(ASDR_TAP_AGC, W0);

    // ---- ALS adaptive notch / peak filter, AudioSDR.cpp:324-352 ------------------------------------------------
    int arow = W0;
    if (ABL_ON(ABL_ALS) && __any(als_en)) {
      const int M = P.als_m, D = P.als_delay;
      const bool adaptive = P.flags & ASDR_F_ALS_ADAPTIVE, notch = P.flags & ASDR_F_ALS_NOTCH;
      float *gx = a.als_x + (size_t)ch * ASDR_N + k0, *gw = a.als_w + (size_t)ch * ASDR_N + k0;
      if (als_en) {
        float t[16];
        load16(gx, t); store16(L + ALS_X + k0, t);            // previous block
        load16(L + W0 + k0, t); store16(L + ALS_X + 128 + k0, t); store16(gx, t);
        load16(gw, t); store16(L + ALS_W + k0, t);
      }
      WAVE_SYNC();
#define ALS_HIST(idx) (((idx) >= 0 && (idx) < 256) ? L[ALS_X + (idx)] : 0.0f)
      if (__any(als_en && !adaptive)) {
        if (als_en && !adaptive) {
          for (int j = 0; j < 16; ++j) {
            const int i = 128 + k0 + j;
            float y = 0.0f;
            for (int q = 0; q < M; ++q) y += L[ALS_W + q] * ALS_HIST(i - D - q);
            const float e = L[ALS_X + i] - y;
            L[ALS_OUT + k0 + j] = notch ? e : y;
          }
        }
      }
      if (__any(als_en && adaptive)) {
        // taps change only after samples n = 0, 4, 8, ...; samples sharing one tap set run on lanes s8 = 0..3
#pragma unroll 1
        for (int ep = -1; ep < 32; ++ep) {
          const int base = (ep < 0) ? 0 : 4 * ep + 1;
          const int cntn = (ep < 0) ? 1 : ((ep == 31) ? 3 : 4);
          const int n = base + s8;
          const bool mine = als_en && adaptive && (s8 < cntn);
          if (mine) {
            const int i = 128 + n;
            float y = 0.0f;
            for (int q = 0; q < M; ++q) y += L[ALS_W + q] * ALS_HIST(i - D - q);
            const float e = L[ALS_X + i] - y;
            L[ALS_OUT + n] = notch ? e : y;
            if ((n & 3) == 0) L[SCR + 1] = e;
          }
          WAVE_SYNC();
          const int nu = (ep < 0) ? 0 : 4 * ep + 4;   // the updating sample of this epoch
          if (als_en && adaptive && nu < ASDR_N) {
            const float e = L[SCR + 1];
            const int iu = 128 + nu;
            for (int q = s8; q < M; q += 8) {
              const float gq = e * ALS_HIST(iu - D - q);
              L[ALS_W + q] += P.als_lambda * gq;
            }
          }
          WAVE_SYNC();
        }
      }
      WAVE_SYNC();
      if (als_en) {
        float t[16];
        load16(L + ALS_W + k0, t); store16(gw, t);
        arow = ALS_OUT;   // == W0
      }
    }
    {
      float au[16];
      load16(L + arow + k0, au);
      TAP_REGS(ASDR_TAP_ALS, au);
      // ---- output, AudioSDR.cpp:158-161: float product, x 32767.0 in binary64, truncate, wrap to int16 ------
      union { int4 v[2]; int16_t s[16]; } ro;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int v = (int)((double)(P.output_gain * au[j]) * 32767.0);
        ro.s[j] = muted ? (int16_t)0 : (int16_t)v;
        if ((j & 3) == 3) SCHED_FENCE();
      }
      if (valid) {
        int4 *po = reinterpret_cast<int4 *>(a.out + io);
        po[0] = ro.v[0]; po[1] = ro.v[1];
      }
    }
    if (lead) S->status = status;
    WAVE_SYNC();
  }
}

// ---- state (re-)initialisation kernel: applies ChanParams.reset bits, one thread per (channel, word) ------
extern "C" __global__ void asdr_reset_kernel(UpdateArgs a, const uint32_t *reset_bits, int n_rows) {
  const int ch = blockIdx.x;
  if (ch >= n_rows) return;
  const uint32_t r = reset_bits[ch];
  if (!r) return;
  const int t = threadIdx.x;   // 128 threads
  ChanSmall *S = a.small + ch;
  const bool all = r & ASDR_R_ALL;
  if (all) {
    uint32_t *w = reinterpret_cast<uint32_t *>(S);
    for (int i = t; i < (int)(sizeof(ChanSmall) / 4); i += 128) w[i] = 0u;
    a.hil_i[(size_t)ch * 256 + t] = 0.0f; a.hil_i[(size_t)ch * 256 + 128 + t] = 0.0f;
    a.hil_q[(size_t)ch * 256 + t] = 0.0f; a.hil_q[(size_t)ch * 256 + 128 + t] = 0.0f;
  }
  __syncthreads();
  if (all && t == 0) { S->nb_avg = 10.0f; S->status = ASDR_S_AGC_ACTIVE; }   // AudioSDR.h:242, :230
  if (all || (r & ASDR_R_IF)) { if (t < 32) (&S->if_state[0][0])[t] = 0.0f; }
  if (all || (r & ASDR_R_IMG)) { if (t < 32) (&S->img_state[0][0])[t] = 0.0f; }
  if (all || (r & ASDR_R_AF)) { if (t < 16) S->af_state[t] = 0.0f; }
  if (all || (r & ASDR_R_NB)) {
    for (int i = t; i < 768; i += 128) a.nb_hist[(size_t)ch * 768 + i] = 0.0f;
    for (int i = t; i < ASDR_NB_MASK_ROW; i += 128) a.nb_mask[(size_t)ch * ASDR_NB_MASK_ROW + i] = 1.0f;
  }
  if (all || (r & ASDR_R_ALS)) {
    a.als_x[(size_t)ch * 128 + t] = 0.0f;
    a.als_w[(size_t)ch * 128 + t] = 0.0f;
  }
}

extern "C" int asdr_launch_update(const UpdateArgs *a, hipStream_t stream) {
  const int n_waves = a->n_sched / 8;
  if (n_waves <= 0) return 0;
  hipLaunchKernelGGL(asdr_update_kernel, dim3(n_waves), dim3(64), 0, stream, *a);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

extern "C" int asdr_launch_reset(const UpdateArgs *a, const uint32_t *d_reset_bits, int n_rows, hipStream_t stream) {
  hipLaunchKernelGGL(asdr_reset_kernel, dim3(n_rows), dim3(128), 0, stream, *a, d_reset_bits, n_rows);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
