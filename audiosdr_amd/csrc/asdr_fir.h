// asdr_fir.h -- the folded 257-tap Hilbert FIR shared by the update kernels (AudioSDR.cpp:99-110) and the IQ generator
// (AudioIQgenerator.cpp:60-76): same structure, different taps.  Device code only.
#ifndef ASDR_FIR_H
#define ASDR_FIR_H
#include <hip/hip_runtime.h>

typedef float v2f __attribute__((ext_vector_type(2)));   // one aligned VGPR pair: operand of v_pk_mul_f32 / v_pk_add_f32
#ifndef ASDR_FIR_TAPS_IN_VGPRS
#define ASDR_FIR_TAPS_IN_VGPRS 0
#endif
#ifndef SCHED_FENCE
#define SCHED_FENCE() do { } while (0)
#endif

// Output pairs E0 .. E0 + NE - 1 of this lane's eight (outputs i = 16 s8 + 2e, + 1):
//   Q[i] = sum_k h[k] * (x[255 + i - 2k] - x[i + 2k + 1]),  k = 0..63 ascending, accumulate from 0.0, every operation separately rounded
// `hist` holds three blocks of history in natural order shifted by one float -- x[m] at hist[m - 1], m = 1..383 (x[0] is never
// used) -- so that every operand pair PX[p] = (x[2p+1], x[2p+2]) is an 8-byte-aligned LDS pair.  The operands of the output pair e
// are PX[127 + p0 + e - k] and PX[p0 + e + k] (p0 = 8 s8): per chunk of 8 taps two contiguous (NE + 7)-pair register windows
// (ds_read_b128).  v_pk_mul_f32 / v_pk_add_f32 round each half exactly like the scalar ops, so the result is bit-identical to
// the scalar loop.  `taps`: 64 floats in constant memory (uniform address: scalar loads).
template <int E0, int NE, bool TAPS_V = (ASDR_FIR_TAPS_IN_VGPRS != 0)>
__device__ __forceinline__ void hilbert_fir_rows(const float *hist, int p0, v2f *acc2, const float *taps) {
  constexpr int NW = NE + 7;
  const v2f *PX = reinterpret_cast<const v2f *>(hist) + E0;
#pragma unroll 1
  for (int kc = 0; kc < 8; ++kc) {
    v2f dw[NW], uw[NW];   // dw[t] = PX[120 + p0 - 8kc + t], uw[t] = PX[p0 + 8kc + t]  (t = 0..NW-1, pairs counted from E0)
    const float4 *dp = reinterpret_cast<const float4 *>(PX + 120 + p0 - 8 * kc);
    const float4 *up = reinterpret_cast<const float4 *>(PX + p0 + 8 * kc);
#pragma unroll
    for (int q = 0; q < NW / 2; ++q) {
      const float4 d4 = dp[q], u4 = up[q];
      dw[2 * q] = (v2f){d4.x, d4.y}; dw[2 * q + 1] = (v2f){d4.z, d4.w};
      uw[2 * q] = (v2f){u4.x, u4.y}; uw[2 * q + 1] = (v2f){u4.z, u4.w};
    }
    if (NW & 1) { dw[NW - 1] = PX[120 + p0 - 8 * kc + NW - 1]; uw[NW - 1] = PX[p0 + 8 * kc + NW - 1]; }
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      float hk = taps[8 * kc + kk];
      // Round 6.  The tap arrives through a scalar load; as an SGPR operand of the eight v_pk_mul_f32 below it makes a third of the loop's vector
      // instructions read the scalar file, and a stream that dense in scalar operands does not share a SIMD: tools/ubench/issue_rate.hip -- this
      // loop runs 4.95 cycles per instruction alone and 8.2 for EACH of two or three resident waves (0.20 / 0.24 / 0.36 instructions per cycle and
      // SIMD), while the same operations on VGPR operands keep ~4.6 cycles per instruction up to three waves (0.22 / 0.44 / 0.66).  One move per tap.
      if constexpr (TAPS_V) asm("" : "+v"(hk));
      const v2f hk2 = (v2f){hk, hk};
      v2f d[NE];   // the pair-chains of a tap are independent: issue them interleaved (no dependent back-to-back pk ops)
#pragma unroll
      for (int e = 0; e < NE; ++e) d[e] = dw[7 + e - kk] - uw[e + kk];
#pragma unroll
      for (int e = 0; e < NE; ++e) d[e] = hk2 * d[e];
#pragma unroll
      for (int e = 0; e < NE; ++e) acc2[e] += d[e];
      SCHED_FENCE();
    }
  }
}
#endif
