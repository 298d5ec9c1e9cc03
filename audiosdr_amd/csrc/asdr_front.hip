// asdr_front.hip -- gfx950 kernels of the AudioStream blocks around the AudioSDR hot path (include/asdr_front.h):
//   asdr_pre_kernel    AudioSDRpreProcessor::update()    (AudioSDRpreProcessor.cpp:46-138)
//   asdr_iqgen_kernel  AudioIQgenerator::update()        (AudioIQgenerator.cpp:33-87)
//   asdr_grab_kernel   AudioGrabberComplex256::update()  (AudioGrabberComplex256.cpp:50-72)
// All three are HBM-streaming int16 work around a little float32 arithmetic; arithmetic order follows the reference
// (built with -ffp-contract=off), the binary64 islands are evaluated in binary64.  One workgroup == one wave64, so
// LDS hand-offs need only a wave barrier.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "asdr_front_device.h"
#include "asdr_front_tables.h"
#include "asdr_fir.h"

#define WAVE_SYNC() __syncthreads() /* workgroup == one wave: a wave barrier + LDS/VMEM waits */
#ifndef ASDR_PRE_ABLATE
#define ASDR_PRE_ABLATE 0   /* profiling builds: detector phases compiled out (1 radix-8-by-2, 2 stage 1, 4 stage 2, 8 powers, 16 sum, 32 maximum scan, 64 unit scale + store) */
#endif
#define PRE_ON(bit) (!(ASDR_PRE_ABLATE & (bit)))
#ifndef ASDR_IQ_ABLATE
#define ASDR_IQ_ABLATE 0    /* profiling builds: IQ generator phases compiled out (1 FIR, 2 binary64 output conversion, 4 scale + staging of the carried blocks) */
#endif
#define IQ_ON(bit) (!(ASDR_IQ_ABLATE & (bit)))

__constant__ float c_iq_taps[64];
__constant__ float c_cftw[128][2];   // CMSIS twiddleCoef_128 (cos, +sin): the detector's FFT is the reference's arm_cfft_f32
__constant__ float c_tw256[128][2];

extern "C" int asdr_front_upload_tables(void) {
  if (hipMemcpyToSymbol(HIP_SYMBOL(c_iq_taps), asdr_iqgen_hilbert_taps, sizeof(asdr_iqgen_hilbert_taps)) != hipSuccess) return -1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(c_cftw), asdr_cfft128_tw, sizeof(asdr_cfft128_tw)) != hipSuccess) return -1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(c_tw256), asdr_fft256_tw, sizeof(asdr_fft256_tw)) != hipSuccess) return -1;
  return 0;
}

// (float)((double)s / 32767.0): the correctly rounded binary64 quotient as ONE multiply and ONE fma -- 1/32767 = r + r 2^-60 + ..
// with r = RN(1/32767) = 0x1.0002000400080p-15, and fma(x, r, x * r 2^-60) equals true division for every int16 (exhaustive:
// oracle ao_check_scale_division; asdr_kernels.hip div_i16_by_32767) -- then one rounding to float32: the reference's
// `float(x)/32767.0` (AudioSDRpreProcessor.cpp:89-90, AudioIQgenerator.cpp:56).
__device__ __forceinline__ float unit_scale(int s) {
  const double x = (double)s;
  return (float)__builtin_fma(x, 0x1.0002000400080p-15, x * 0x1.0002000400080p-75);
}

// The same value in three binary32 operations (the update kernel's unit-gain scale, asdr_kernels.hip scale8): equal to the form above for
// every int16 (oracle ao_check_scale_unit_gain, CPU-exhaustive).  The IQ generator scales its carried blocks again in every launch.
__device__ __forceinline__ float unit_scale32(int s) {
  const float x = (float)s;
  return __builtin_fmaf(x, 0x1.0002p-15f, x * 0x1.0002p-45f);
}

union Raw8 { int4 v; int16_t s[8]; };
__device__ __forceinline__ int pack16(int lo, int hi) { return (lo & 0xFFFF) | (hi << 16); }

// =====================================================================================================
// AudioSDRpreProcessor: 16 lanes per channel (8 samples each; a DPP row is exactly one channel), 4 channels per wave.
// Blocks of one call are processed in order because the detector may change the correction between blocks.
// =====================================================================================================
// One radix-8 butterfly of CMSIS-DSP arm_radix8_butterfly_f32 on the points xin[0..7] = X[i1], X[i1 + n2], .., X[i1 + 7 n2], outputs in
// o[0..7] for the same places (registers: the caller loads and stores) (oracle/asdr_front_oracle.c
// cf_radix8, which is held bit for bit against the reference's Cortex-M4 object: tests/test_cmsis_object.py): TW = false is the
// twiddle-free form (the first column of a stage / the last stage), TW = true multiplies outputs 2..8 by (co[k], si[k]) = tw[(k - 1) tws].
// Every product and sum separately rounded, in the object's association.
#define CF_C81 0.70710678118f
template <bool TW>
__device__ __forceinline__ void cf_radix8(const float2 *xin, float2 *o, const float2 *tw, int tws) {
  const float2 x1 = xin[0], x2 = xin[1], x3 = xin[2], x4 = xin[3], x5 = xin[4], x6 = xin[5], x7 = xin[6], x8 = xin[7];
  float r1, r2, r3, r4, r5, r6, r7, r8, t1, t2, s1, s2, s3, s4, s5, s6, s7, s8;
  r1 = x1.x + x5.x; r5 = x1.x - x5.x;
  r2 = x2.x + x6.x; r6 = x2.x - x6.x;
  r3 = x3.x + x7.x; r7 = x3.x - x7.x;
  r4 = x4.x + x8.x; r8 = x4.x - x8.x;
  t1 = r1 - r3; r1 = r1 + r3; r3 = r2 - r4; r2 = r2 + r4;
  if (!TW) {
    float2 o1, o2, o3, o4, o5, o6, o7, o8;
    o1.x = r1 + r2; o5.x = r1 - r2;
    r1 = x1.y + x5.y; s5 = x1.y - x5.y;
    r2 = x2.y + x6.y; s6 = x2.y - x6.y;
    s3 = x3.y + x7.y; s7 = x3.y - x7.y;
    r4 = x4.y + x8.y; s8 = x4.y - x8.y;
    t2 = r1 - s3; r1 = r1 + s3; s3 = r2 - r4; r2 = r2 + r4;
    o1.y = r1 + r2; o5.y = r1 - r2;
    o3.x = t1 + s3; o7.x = t1 - s3; o3.y = t2 - r3; o7.y = t2 + r3;
    r1 = (r6 - r8) * CF_C81; r6 = (r6 + r8) * CF_C81; r2 = (s6 - s8) * CF_C81; s6 = (s6 + s8) * CF_C81;
    t1 = r5 - r1; r5 = r5 + r1; r8 = r7 - r6; r7 = r7 + r6;
    t2 = s5 - r2; s5 = s5 + r2; s8 = s7 - s6; s7 = s7 + s6;
    o2.x = r5 + s7; o8.x = r5 - s7; o6.x = t1 + s8; o4.x = t1 - s8;
    o2.y = s5 - r7; o8.y = s5 + r7; o6.y = t2 - r8; o4.y = t2 + r8;
    o[0] = o1; o[1] = o2; o[2] = o3; o[3] = o4; o[4] = o5; o[5] = o6; o[6] = o7; o[7] = o8;
  } else {
    float2 o1, o2, o3, o4, o5, o6, o7, o8;
    float p1, p2, p3, p4;
    o1.x = r1 + r2; r2 = r1 - r2;
    s1 = x1.y + x5.y; s5 = x1.y - x5.y;
    s2 = x2.y + x6.y; s6 = x2.y - x6.y;
    s3 = x3.y + x7.y; s7 = x3.y - x7.y;
    s4 = x4.y + x8.y; s8 = x4.y - x8.y;
    t2 = s1 - s3; s1 = s1 + s3; s3 = s2 - s4; s2 = s2 + s4;
    r1 = t1 + s3; t1 = t1 - s3;
    o1.y = s1 + s2; s2 = s1 - s2;
    s1 = t2 - r3; t2 = t2 + r3;
    const float2 w2 = tw[tws], w3 = tw[2 * tws], w4 = tw[3 * tws], w5 = tw[4 * tws], w6 = tw[5 * tws], w7 = tw[6 * tws], w8 = tw[7 * tws];
    p1 = w5.x * r2; p2 = w5.y * s2; p3 = w5.x * s2; p4 = w5.y * r2; o5.x = p1 + p2; o5.y = p3 - p4;
    p1 = w3.x * r1; p2 = w3.y * s1; p3 = w3.x * s1; p4 = w3.y * r1; o3.x = p1 + p2; o3.y = p3 - p4;
    p1 = w7.x * t1; p2 = w7.y * t2; p3 = w7.x * t2; p4 = w7.y * t1; o7.x = p1 + p2; o7.y = p3 - p4;
    r1 = (r6 - r8) * CF_C81; r6 = (r6 + r8) * CF_C81; s1 = (s6 - s8) * CF_C81; s6 = (s6 + s8) * CF_C81;
    t1 = r5 - r1; r5 = r5 + r1; r8 = r7 - r6; r7 = r7 + r6;
    t2 = s5 - s1; s5 = s5 + s1; s8 = s7 - s6; s7 = s7 + s6;
    r1 = r5 + s7; r5 = r5 - s7; r6 = t1 + s8; t1 = t1 - s8;
    s1 = s5 - r7; s5 = s5 + r7; s6 = t2 - r8; t2 = t2 + r8;
    p1 = w2.x * r1; p2 = w2.y * s1; p3 = w2.x * s1; p4 = w2.y * r1; o2.x = p1 + p2; o2.y = p3 - p4;
    p1 = w8.x * r5; p2 = w8.y * s5; p3 = w8.x * s5; p4 = w8.y * r5; o8.x = p1 + p2; o8.y = p3 - p4;
    p1 = w6.x * r6; p2 = w6.y * s6; p3 = w6.x * s6; p4 = w6.y * r6; o6.x = p1 + p2; o6.y = p3 - p4;
    p1 = w4.x * t1; p2 = w4.y * t2; p3 = w4.x * t2; p4 = w4.y * t1; o4.x = p1 + p2; o4.y = p3 - p4;
    o[0] = o1; o[1] = o2; o[2] = o3; o[3] = o4; o[4] = o5; o[5] = o6; o[6] = o7; o[7] = o8;
  }
}

__device__ __forceinline__ int dpp_row_shr1_i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false); }

__global__ __launch_bounds__(64) void asdr_pre_kernel(PreArgs a) {
  // FFT work area, one row per channel of the wave; point `pos` lives at XI(pos) = pos + pos / 8 (one pad per eight points): a lane's eight
  // consecutive points (natural-order store, last radix-8 stage) are then 9 slots from its neighbour's and the 16 lanes of a channel hit
  // 16 different bank pairs; unpadded, that stride is 16 floats -> 8-way conflicts (the columns of the first radix-8 stage: 2-way -> none)
  __shared__ float2 X[4][144];
#define XI(pos) ((pos) + ((pos) >> 3))
  __shared__ float P[4][132];       // line powers 0..127 (+ the reference's buffer[128] at [128])
  __shared__ float2 TW[128];        // CMSIS twiddleCoef_128: (cos, +sin)(2 pi k / 128)
  const int lane = threadIdx.x, r = lane >> 4, l = lane & 15;
  TW[lane] = make_float2(c_cftw[lane][0], c_cftw[lane][1]);
  TW[lane + 64] = make_float2(c_cftw[lane + 64][0], c_cftw[lane + 64][1]);
  // PERSISTENT waves (round 5, like the IQ generator below): the grid is what the chip holds at once and a wave walks over its channel
  // quads g = blockIdx.x, + gridDim.x, ... with the next work item's state and input block requested before the detector of the current
  // one.  (One workgroup per quad ran the resident waves in step -- everybody waits for HBM, everybody transforms, everybody stores: 46 % of
  // a wave's life parked at a wait.)
  const int n_quads = (a.n_channels + 3) >> 2;
  asdr_pre_state_t n_st;
  Raw8 n_ri, n_rq;
  auto ask_state = [&](int g) { const int c = g * 4 + r; n_st = a.state[(c < a.n_channels) ? c : 0]; };   // every lane of the row keeps a copy; all of them update it identically
  auto ask_block = [&](int g, int blk) {
    const int c = g * 4 + r;
    const size_t io = ((size_t)((c < a.n_channels) ? c : 0) * a.in_stride + blk) * ASDR_N + 8 * l;
    n_ri.v = *reinterpret_cast<const int4 *>(a.in_i + io);
    n_rq.v = *reinterpret_cast<const int4 *>(a.in_q + io);
  };
  int g = (int)blockIdx.x;
  if (g < n_quads) { ask_state(g); ask_block(g, 0); }
  WAVE_SYNC();
#pragma unroll 1
  for (; g < n_quads; g += (int)gridDim.x) {
  const int ch = g * 4 + r;
  const bool valid = ch < a.n_channels;
  asdr_pre_state_t st = n_st;
  if (!valid) { st.correction = 0; st.saved_sample = 0; st.failure_count = 0; st.success_count = 0; st.auto_detect = 0; st.swap = 0;
                st.max_line = 0; st.strong = 0; st.max_power = 0.f; st.avg_power = 0.f; st.ratio = 0.f; }
  int corr = st.correction, saved = st.saved_sample, fail = st.failure_count, succ = st.success_count;
  int autodet = valid ? st.auto_detect : 0;
  const int swap = st.swap;

#pragma unroll 1
  for (int blk = 0; blk < a.n_blocks; ++blk) {
    Raw8 ri = n_ri, rq = n_rq;
    if (!valid) ri.v = rq.v = make_int4(0, 0, 0, 0);
    // the next work item's rows: asked for now, used after this block's detector
    if (blk + 1 < a.n_blocks) ask_block(g, blk + 1);
    else if (g + (int)gridDim.x < n_quads) { ask_state(g + (int)gridDim.x); ask_block(g + (int)gridDim.x, 0); }
    int xi[8], xq[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { xi[j] = ri.s[j]; xq[j] = rq.s[j]; }

    // ---- single-sample I/Q skew compensation, .cpp:62-72 -----------------------------------------------
    {
      const int prev_i = dpp_row_shr1_i(xi[7]), prev_q = dpp_row_shr1_i(xq[7]);   // left neighbour's last sample
      const int tail_i = __shfl(xi[7], 15, 16), tail_q = __shfl(xq[7], 15, 16);   // the row's sample 127
      if (corr == 1) {            // I delayed by one sample; sample 127 is carried to the next block
#pragma unroll
        for (int j = 7; j > 0; --j) xi[j] = xi[j - 1];
        xi[0] = (l == 0) ? saved : prev_i;
        saved = tail_i;
      } else if (corr == -1) {    // Q delayed by one sample -- but the carried Q sample is written to I[0] (.cpp:69)
        const int q0 = xq[0];
#pragma unroll
        for (int j = 7; j > 0; --j) xq[j] = xq[j - 1];
        xq[0] = (l == 0) ? q0 : prev_q;
        if (l == 0) xi[0] = saved;
        saved = tail_q;
      }
    }

    // ---- skew detector, .cpp:82-122: image ratio of the strongest line of a 128-point FFT ---------------
    if (__any(autodet)) {
      if (PRE_ON(64) && autodet) {
        // the block as complex float32 in natural order (.cpp:88-91)
#pragma unroll
        for (int j = 0; j < 8; ++j) X[r][9 * l + j] = make_float2(unit_scale32(xi[j]), unit_scale32(xq[j]));   // XI(8 l + j)
      }
      WAVE_SYNC();
      // arm_cfft_f32(&arm_cfft_sR_f32_len128, buffer, 0, 1) (.cpp:93), operation for operation (oracle ao_fft128 == the reference's
      // Cortex-M4 objects, bit for bit).  (1) arm_cfft_radix8by2_f32: the quarters q, q + 32, q + 64, q + 96 -- 32 values of q, two per lane.
      if (PRE_ON(1) && autodet) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int q = l + 16 * h;
          const int q1 = XI(q), q3 = XI(q + 32), q2 = XI(q + 64), q4 = XI(q + 96);
          const float2 t1 = X[r][q1], t3 = X[r][q3], t2o = X[r][q2], t4o = X[r][q4], w = TW[q];
          const float2 sum1 = make_float2(t1.x + t2o.x, t1.y + t2o.y), t2 = make_float2(t1.x - t2o.x, t1.y - t2o.y);
          const float2 sum3 = make_float2(t3.x + t4o.x, t3.y + t4o.y), t4 = make_float2(t4o.x - t3.x, t4o.y - t3.y);
          float m0 = t2.x * w.x, m1 = t2.y * w.y, m2 = t2.y * w.x, m3 = t2.x * w.y;
          const float2 o2 = make_float2(m0 + m1, m2 - m3);
          m0 = t4.x * w.y; m1 = t4.y * w.x; m2 = t4.y * w.y; m3 = t4.x * w.x;
          const float2 o4 = make_float2(m0 - m1, m2 + m3);
          X[r][q1] = sum1; X[r][q3] = sum3; X[r][q2] = o2; X[r][q4] = o4;
        }
      }
      WAVE_SYNC();
      // (2) arm_radix8_butterfly_f32(half, 64, twiddleCoef_128, 2) on both halves, stage 1 (n2 = 8): 16 butterflies, one per lane --
      // column j = l & 7 of half l >> 3; column 0 is the twiddle-free form, column j uses tw[(k - 1) 2 j]
      if (PRE_ON(2) && autodet) {
        float2 *Xh = X[r] + 72 * (l >> 3);          // XI(64 half + j + 8 k) = 72 half + j + 9 k
        const int j = l & 7;
        float2 x[8], o[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = Xh[j + 9 * k];
        if (j == 0) cf_radix8<false>(x, o, TW, 0);
        else cf_radix8<true>(x, o, TW, 2 * j);
#pragma unroll
        for (int k = 0; k < 8; ++k) Xh[j + 9 * k] = o[k];
      }
      WAVE_SYNC();
      // (3) stage 2 (n2 = 1): eight twiddle-free butterflies per half on consecutive points, one per lane: lane l owns positions
      // 8 l .. 8 l + 7.  (4) arm_bitreversal_32 is a permutation -- natural-order line k = 16 c + 2 b + a sits at position 64 a + 8 b + c --
      // and the powers (arm_cmplx_mag_squared_f32, .cpp:94: re re + im im, also held against its object) are formed straight from the
      // butterfly's outputs in registers: position 8 l + j is line 16 j + 2 (l & 7) + (l >> 3), so the 16 lanes of a channel write 16
      // consecutive words of P for every j.
      if (autodet) {
        float2 x[8], o[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = X[r][9 * l + k];
        if (PRE_ON(4)) cf_radix8<false>(x, o, TW, 0);
        else {
#pragma unroll
          for (int k = 0; k < 8; ++k) o[k] = x[k];
        }
        if (PRE_ON(8)) {
          const int k0 = 2 * (l & 7) + (l >> 3);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float p = o[j].x * o[j].x, q = o[j].y * o[j].y;
            P[r][16 * j + k0] = p + q;
          }
          if (l == 0) P[r][128] = o[4].x;   // line 64's real part (position 4): what the reference's buffer[128] holds after the in-place magnitude pass
        }
      }
      WAVE_SYNC();
      if (autodet) {
        // .cpp:96-105.  The float SUM over lines 5..122 is sequential in the reference and stays so (every lane of the row redundantly:
        // same LDS addresses -> broadcast reads, four lines per read; no hand-off).  The first-maximum scan
        // (`if (p > max) { line = i; max = p; }`, i ascending, from max = 0.0, line = 0) is order-free once ties go to the LOWER
        // line: each lane scans its own eight lines, then four exchange steps over the row's 16 lanes keep, of two candidates, the
        // higher-indexed one only if it is STRICTLY greater (a NaN power never wins, as in the reference: `NaN > max` is false).
        float avg = 0.0f;
        if (PRE_ON(16)) {
          const float4 *P4 = reinterpret_cast<const float4 *>(P[r]);
          float4 v = P4[1];                                   // lines 4..7
          avg += v.y; avg += v.z; avg += v.w;                 // 5, 6, 7
#pragma unroll 4
          for (int q = 2; q < 30; ++q) { v = P4[q]; avg += v.x; avg += v.y; avg += v.z; avg += v.w; }   // 8..119
          v = P4[30];                                         // lines 120..123
          avg += v.x; avg += v.y; avg += v.z;                 // 120, 121, 122
        }
        float mx = 0.0f;
        int line = 0;
        if (PRE_ON(32)) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int i = 8 * l + j;
          const float p = P[r][i];
          if (i >= 5 && i < 123 && p > mx) { line = i; mx = p; }
        }
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) {
          const float om = __shfl_xor(mx, d, 16);
          const int ol = __shfl_xor(line, d, 16);
          const bool other_is_higher = (l & d) == 0;          // the partner holds higher line numbers
          // keep the lower-indexed candidate unless the higher-indexed one is strictly greater
          const bool take = other_is_higher ? (om > mx) : !(mx > om);
          mx = take ? om : mx; line = take ? ol : line;
        }
        }
        avg /= 118.0f;
        const float ratio = mx / P[r][128 - line];                       // .cpp:107
        int strong = 0;
        if ((double)mx > 10.0 * (double)avg) {                            // .cpp:109
          strong = 1;
          if ((double)ratio < 10.0) fail = (int16_t)(fail + 1); else fail = 0;   // .cpp:110-111
          if (fail > 10) {                                                 // .cpp:112-117
            corr = (int16_t)(corr + 1);
            if (corr > 1) corr = -1;
            fail = 0; succ = 0;
          }
          succ = (int16_t)(succ + 1);                                      // .cpp:118
        }
        if (succ > 1000) autodet = 0;                                      // .cpp:120-122
        st.max_line = line; st.strong = strong; st.max_power = mx; st.avg_power = avg; st.ratio = ratio;
      }
      WAVE_SYNC();
    }

    // ---- I/Q swap (.cpp:127-133) and store -----------------------------------------------------------------
    if (valid) {
      Raw8 oi, oq;
#pragma unroll
      for (int j = 0; j < 8; ++j) { oi.s[j] = (int16_t)(swap ? xq[j] : xi[j]); oq.s[j] = (int16_t)(swap ? xi[j] : xq[j]); }
      const size_t oo = ((size_t)ch * a.out_stride + blk) * ASDR_N + 8 * l;
      *reinterpret_cast<int4 *>(a.out_i + oo) = oi.v;
      *reinterpret_cast<int4 *>(a.out_q + oo) = oq.v;
    }
  }
  if (valid && l == 0) {
    st.correction = (int16_t)corr; st.saved_sample = (int16_t)saved; st.failure_count = (int16_t)fail; st.success_count = (int16_t)succ;
    st.auto_detect = autodet;
    a.state[ch] = st;
  }
  }
}

// workgroups of 64 threads the current device holds at once (persistent waves): CUs x what the occupancy calculator says for the kernel
template <typename K>
static int resident_workgroups(K kernel, int *cache) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  if (dev < 0 || dev >= 64) dev = 0;
  if (cache[dev] == 0) {
    int cus = 0, per_cu = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 64, 0) != hipSuccess || per_cu <= 0) per_cu = 8;
    cache[dev] = cus * per_cu;
  }
  return cache[dev];
}

extern "C" int asdr_launch_pre(const PreArgs *a, void *stream) {
  static int resident[64] = {0};
  const int res = resident_workgroups(asdr_pre_kernel, resident);
  if (res <= 0) return -1;
  const int n_quads = (a->n_channels + 3) / 4;
  hipLaunchKernelGGL(asdr_pre_kernel, dim3(n_quads < res ? n_quads : res), dim3(64), 0, (hipStream_t)stream, *a);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// =====================================================================================================
// AudioIQgenerator: 8 channels per wave, 8 lanes per channel, 16 output samples per lane.  The 3-block delay
// line (384 floats) of each channel lives in LDS for the whole call and slides by one block per iteration.
//   Q[i] = sum_k c[k] * (w[255 + i - 2k] - w[i + 2k + 1]),  k ascending, float32 (.cpp:65-73);  I[i] = w[128 + i] (.cpp:75)
// The folded FIR runs on register windows: for the 8 same-parity outputs of a lane and 16 consecutive taps, both
// operand streams are 23-entry stride-2 windows of the delay line.
// =====================================================================================================
#define IQ_STRIDE 388   /* 97 sixteen-byte slots == 1 (mod 16): the 8 channel rows start on different LDS slots */

// PERSISTENT waves (round 5): the grid is what the chip holds at once (asdr_launch_iqgen) and every wave walks over its channel groups
// g = blockIdx.x, + gridDim.x, ... with the NEXT work item's raw rows (the carried ring slots, the input block, the gains) requested before the
// FIR of the current one.  With one workgroup per channel group all resident waves ran the same phase at the same time -- everybody loads,
// everybody runs the FIR, everybody stores -- in three rounds (8,192 groups on 3,072 slots): load + store time and FIR time added up
// (19.6 + 29.5 us of 49 measured with either compiled out, profiles/README.md).  What is left is arithmetic: the counters show the vector
// units 98 % busy (three waves per SIMD, each with a vector instruction active in 33 % of its cycles).
__global__ __launch_bounds__(64) void asdr_iqgen_kernel(IqgenArgs a) {
  __shared__ float W[8 * IQ_STRIDE];
  const int lane = threadIdx.x, c8 = lane >> 3, s8 = lane & 7;
  float *w = W + c8 * IQ_STRIDE;
  const int i0 = 16 * s8;
  // Global rows are touched in whole 16-byte pieces with the 8 lanes of a channel on 8 ADJACENT pieces (one wave instruction reads or
  // writes 128 contiguous bytes of each of its 8 rows), as in the update kernel: lanes on every second or eighth piece make every
  // instruction touch every line of the rows (round 4: 0.0557 -> see profiles/README.md).  int16 rows: lane s8 owns samples
  // kA + 64 h + j (h = 0, 1; j < 8).
  const int kA = 8 * s8;
  const int n_groups = (a.n_channels + 7) >> 3;
  const uint32_t slot_old = a.phase & 1u;
  // The two carried blocks come from a RAW int16 ring in HBM (slot a.phase = the older one) and are scaled again here: the scale is a pure
  // function of the sample, so the floats are the reference's bit for bit at a quarter of the carried traffic (round 5: 2,048 B of float
  // history read + written per channel and launch -> 512 B read + 256 B written; what the update kernel does for its blanker ring).
  Raw8 n_old[2], n_prev[2], n_new[2];   // the next work item's rows, in flight
  float n_gi = 0.0f, n_gq = 0.0f;
  auto ask_group = [&](int g) {          // ring slots + gains of group g (lanes of a padding channel read channel 0 and discard)
    const int ch = g * 8 + c8, chc = (ch < a.n_channels) ? ch : 0;
    const int16_t *hrow = a.hist + (size_t)chc * 256;
    const int4 *po = reinterpret_cast<const int4 *>(hrow + slot_old * 128 + kA), *pp = reinterpret_cast<const int4 *>(hrow + (slot_old ^ 1u) * 128 + kA);
    n_old[0].v = po[0]; n_old[1].v = po[8]; n_prev[0].v = pp[0]; n_prev[1].v = pp[8];
    n_gi = a.gains[2 * chc]; n_gq = a.gains[2 * chc + 1];
  };
  auto ask_block = [&](int g, int blk) { // input block blk of group g
    const int ch = g * 8 + c8, chc = (ch < a.n_channels) ? ch : 0;
    const int4 *p = reinterpret_cast<const int4 *>(a.in + ((size_t)chc * a.in_stride + blk) * ASDR_N + kA);
    n_new[0].v = p[0]; n_new[1].v = p[8];   // int4 #8 = 64 samples on
  };
  int g = (int)blockIdx.x;
  if (g < n_groups) { ask_group(g); ask_block(g, 0); }
#pragma unroll 1
  for (; g < n_groups; g += (int)gridDim.x) {
    const int ch = g * 8 + c8;
    const bool valid = ch < a.n_channels;
    const int chc = valid ? ch : 0;
    int16_t *const hrow = a.hist + (size_t)chc * 256;
    const float gain_i = n_gi, gain_q = n_gq;
    // Delay line in LDS: sample x[m] of the reference's 384-sample window (two carried blocks + the newest) at w[m - 1], m = 1..383
    // -- natural order shifted by one float (x[0] is never read), the layout the shared packed FIR wants (asdr_fir.h).
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int m = 64 * h + kA + j;                        // x[m] of the oldest block at w[m - 1]; x[0] is never read: the padding word
        if (IQ_ON(4)) {
        w[(m == 0) ? 383 : m - 1] = unit_scale32(n_old[h].s[j]);
        w[127 + m] = unit_scale32(n_prev[h].s[j]);            // x[128 + m]
        }
      }
    }
#pragma unroll 1
    for (int blk = 0; blk < a.n_blocks; ++blk) {
      {   // newest block -> x[256..383] = w[255..382], scaled (.cpp:56)
        const Raw8 r0 = n_new[0], r1 = n_new[1];
#pragma unroll
        for (int j = 0; j < 8; ++j) { w[255 + kA + j] = unit_scale32(r0.s[j]); w[255 + 64 + kA + j] = unit_scale32(r1.s[j]); }
        if (valid && blk >= a.n_blocks - 2) {   // one of the two blocks the next call starts from: raw, into the slot of the block it replaces
          int4 *ph = reinterpret_cast<int4 *>(hrow + ((a.phase + (uint32_t)blk) & 1u) * 128 + kA);
          ph[0] = r0.v; ph[8] = r1.v;
        }
      }
      // the next work item's rows: asked for now, used after this block's FIR
      if (blk + 1 < a.n_blocks) ask_block(g, blk + 1);
      else if (g + (int)gridDim.x < n_groups) { ask_group(g + (int)gridDim.x); ask_block(g + (int)gridDim.x, 0); }
      WAVE_SYNC();
      // this lane's 16 outputs i0 + 2e, i0 + 2e + 1 (e = 0..7) as 8 packed pairs: AudioIQgenerator.cpp:60-76, taps c_iq_taps
      v2f acc2[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) acc2[e] = (v2f){0.0f, 0.0f};
      if (IQ_ON(1)) hilbert_fir_rows<0, 8>(w, i0 >> 1, acc2, c_iq_taps);
      // The FIR owns 16 contiguous outputs per lane, the global rows 8-sample pieces: the Q row is handed over through LDS -- in the
      // words of the oldest block (x[1..128] at w[0..127]), which no FIR read needs any more once every lane is through.  (Word 127 =
      // x[128] is the first delayed I sample: the I pieces are read before the hand-over.)
      float ivr[16];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int j = 0; j < 8; ++j) ivr[8 * h + j] = w[127 + 64 * h + kA + j];   // x[128 + k]: the input delayed by one block
      }
      WAVE_SYNC();
#pragma unroll
      for (int e = 0; e < 8; e += 2) *reinterpret_cast<float4 *>(w + i0 + 2 * e) = make_float4(acc2[e][0], acc2[e][1], acc2[e + 1][0], acc2[e + 1][1]);
      WAVE_SYNC();
      {   // .cpp:78-82: (int16_t)(float * 32767.0 * gain) in binary64; v_cvt_i32_f64 saturates like the ARM target
        Raw8 oi[2], oq[2];   // (int16_t) keeps the low half, like the reference's cast
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          int vi[8], vq[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float iv = ivr[8 * h + j];
            const float qv = w[64 * h + kA + j];
            if (IQ_ON(2)) {
            vi[j] = (int)(((double)iv * 32767.0) * (double)gain_i);
            vq[j] = (int)(((double)qv * 32767.0) * (double)gain_q);
            } else { vi[j] = (int)(iv * 32767.0f); vq[j] = (int)(qv * 32767.0f); }
          }
          oi[h].v = make_int4(pack16(vi[0], vi[1]), pack16(vi[2], vi[3]), pack16(vi[4], vi[5]), pack16(vi[6], vi[7]));
          oq[h].v = make_int4(pack16(vq[0], vq[1]), pack16(vq[2], vq[3]), pack16(vq[4], vq[5]), pack16(vq[6], vq[7]));
        }
        if (valid) {
          const size_t oo = ((size_t)ch * a.out_stride + blk) * ASDR_N + kA;
          int4 *po = reinterpret_cast<int4 *>(a.out_i + oo), *pq = reinterpret_cast<int4 *>(a.out_q + oo);
          po[0] = oi[0].v; po[8] = oi[1].v; pq[0] = oq[0].v; pq[8] = oq[1].v;
        }
      }
      WAVE_SYNC();   // (every read of this block's rows is done: the slide, or the next group's staging, may overwrite them)
      if (blk + 1 == a.n_blocks) break;   // (the carried blocks are in the ring already)
      // slide the delay line by one block (.cpp:54-55, 57-58): w[0..127] = w[128..255], then w[128..255] = w[256..383] (word 383 is padding);
      // source and destination of each phase are disjoint, 16 floats per lane
#pragma unroll
      for (int ph = 0; ph < 2; ++ph) {
        const float4 *src = reinterpret_cast<const float4 *>(w + 128 * (ph + 1) + 16 * s8);
        float4 *dst = reinterpret_cast<float4 *>(w + 128 * ph + 16 * s8);
        const float4 t0 = src[0], t1 = src[1], t2 = src[2], t3 = src[3];
        dst[0] = t0; dst[1] = t1; dst[2] = t2; dst[3] = t3;
        WAVE_SYNC();
      }
    }
  }
}

extern "C" int asdr_launch_iqgen(const IqgenArgs *a, void *stream) {
  // as many workgroups as the device holds at once (persistent waves), at most one per channel group
  static int resident[64] = {0};
  const int res = resident_workgroups(asdr_iqgen_kernel, resident);
  if (res <= 0) return -1;
  const int n_groups = (a->n_channels + 7) / 8;
  const int grid = n_groups < res ? n_groups : res;
  hipLaunchKernelGGL(asdr_iqgen_kernel, dim3(grid), dim3(64), 0, (hipStream_t)stream, *a);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// =====================================================================================================
// AudioGrabberComplex256: one wave per channel, two complex samples per lane and half buffer.  Only the blocks
// that can still be visible after the call are touched: the last pair completed by this call (-> _outBuffer and
// _buffer) and a trailing unpaired block (-> first half of _buffer).
// =====================================================================================================
__device__ __forceinline__ int2 interleave2(int i2, int q2) {   // (i0,i1),(q0,q1) -> (i0,q0),(i1,q1)
  return make_int2((i2 & 0xFFFF) | (q2 << 16), ((unsigned)i2 >> 16) | (q2 & 0xFFFF0000));
}

__global__ __launch_bounds__(64) void asdr_grab_kernel(GrabArgs a) {
  const int ch = blockIdx.x, lane = threadIdx.x;
  const int total = a.parity + a.n_blocks, n_pairs = total >> 1;
  int2 *buf = reinterpret_cast<int2 *>(a.buffer + (size_t)ch * 512);       // 128 int2: [0..63] first half, [64..127] second
  int2 *outb = reinterpret_cast<int2 *>(a.out_buffer + (size_t)ch * 512);
  const size_t row = (size_t)ch * a.in_stride * ASDR_N;
  auto load_block = [&](int blk) {
    const int i2 = *reinterpret_cast<const int *>(a.in_i + row + (size_t)blk * ASDR_N + 2 * lane);
    const int q2 = *reinterpret_cast<const int *>(a.in_q + row + (size_t)blk * ASDR_N + 2 * lane);
    return interleave2(i2, q2);
  };
  if (n_pairs >= 1) {
    const int e = 2 * n_pairs - 1 - a.parity;                  // block (of this call) that completes the last pair
    const int2 second = load_block(e);
    const int2 first = (e >= 1) ? load_block(e - 1) : buf[lane];   // its first half may predate this call
    outb[lane] = first; outb[64 + lane] = second;               // .cpp:63-65
    buf[lane] = first; buf[64 + lane] = second;
  }
  if (total & 1) buf[lane] = load_block(a.n_blocks - 1);        // trailing unpaired block, .cpp:59
}

extern "C" int asdr_launch_grab(const GrabArgs *a, void *stream) {
  hipLaunchKernelGGL(asdr_grab_kernel, dim3(a->n_channels), dim3(64), 0, (hipStream_t)stream, *a);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// =====================================================================================================
// Panadapter spectrum of the grabber buffers: 4 channels per wave, 16 lanes per channel, 16 points per lane; 256-point
// radix-2 DIT FFT in LDS with the arithmetic of ao_fft256 (oracle/asdr_front_oracle.c), then Re^2 + Im^2.
// =====================================================================================================
__global__ __launch_bounds__(64) void asdr_grab_spectrum_kernel(const int16_t *out_buffer, float *power, int n_channels) {
  __shared__ float2 X[4][256];
  __shared__ float2 TW[128];
  const int lane = threadIdx.x, r = lane >> 4, l = lane & 15;
  const int ch = blockIdx.x * 4 + r;
  const bool valid = ch < n_channels;
  TW[lane] = make_float2(c_tw256[lane][0], c_tw256[lane][1]);
  TW[64 + lane] = make_float2(c_tw256[64 + lane][0], c_tw256[64 + lane][1]);
  // point n = 16*l + j (j = 0..15) -> X[bitrev8(n)]: bitrev8 = bitrev4(j) << 4 | bitrev4(l)
  const int brl = ((l & 1) << 3) | ((l & 2) << 1) | ((l & 4) >> 1) | ((l & 8) >> 3);
  if (valid) {
    const int4 *src = reinterpret_cast<const int4 *>(out_buffer + (size_t)ch * 512 + 32 * l);   // 16 complex int16 = 64 B
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      Raw8 w; w.v = src[q];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int j = 4 * q + t;
        const int brj = ((j & 1) << 3) | ((j & 2) << 1) | ((j & 4) >> 1) | ((j & 8) >> 3);
        X[r][(brj << 4) | brl] = make_float2((float)w.s[2 * t] * (1.0f / 32768.0f), (float)w.s[2 * t + 1] * (1.0f / 32768.0f));
      }
    }
  }
  WAVE_SYNC();
#pragma unroll
  for (int s = 1; s <= 8; ++s) {
    const int h = 1 << (s - 1);
    if (valid) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int b = l + 16 * q;
        const int j = b & (h - 1), k = (b >> (s - 1)) << s;
        const float2 w = TW[j << (8 - s)];
        const float2 u = X[r][k + j], v = X[r][k + j + h];
        const float p0 = w.x * v.x, p1 = w.y * v.y, p2 = w.x * v.y, p3 = w.y * v.x;
        const float tr = p0 - p1, ti = p2 + p3;
        X[r][k + j] = make_float2(u.x + tr, u.y + ti);
        X[r][k + j + h] = make_float2(u.x - tr, u.y - ti);
      }
    }
    WAVE_SYNC();
  }
  if (valid) {
    float *dst = power + (size_t)ch * 256 + 16 * l;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float pw[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) { const float2 x = X[r][16 * l + 4 * q + t]; const float a = x.x * x.x, b = x.y * x.y; pw[t] = a + b; }
      reinterpret_cast<float4 *>(dst)[q] = make_float4(pw[0], pw[1], pw[2], pw[3]);
    }
  }
}

extern "C" int asdr_launch_grab_spectrum(const int16_t *out_buffer, float *power, int n_channels, void *stream) {
  hipLaunchKernelGGL(asdr_grab_spectrum_kernel, dim3((n_channels + 3) / 4), dim3(64), 0, (hipStream_t)stream, out_buffer, power, n_channels);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
