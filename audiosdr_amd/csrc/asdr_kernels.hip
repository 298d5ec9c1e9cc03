// asdr_kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4, wave64) for the batched
// AudioSDR::update() demodulation chain (reference: SRC/AudioSDRlib/AudioSDR.cpp:39-168).
//
// Execution shape
//   * one workgroup == one wavefront (64 lanes) == 8 channels ("slots" c8 = lane>>3, s8 = lane&7);
//     no inter-wave communication, no MFMA, no atomics, no collectives.
//   * pointwise / FIR stages: lane (c8, s8) owns 16 samples of its channel's 128-sample block
//     (contiguous 16, processed LDS-resident in two 8-sample pieces; 16 same-parity outputs for the Hilbert FIR).
//   * IIR biquad cascades: a 4-lane systolic pipeline per cascade (lane = stage, one 4-sample chunk of skew),
//     stage-to-stage hand-off by DPP row_shr:1, so 8 channels x {I,Q} x 4 stages fill the wave.
//   * strictly sequential scalar recurrences (noise-blanker average, mixer phase, AGC envelope, PLL,
//     AM carrier tracker): lane s8==0 of each channel, fed from / draining to LDS in 4/8-sample chunks.
//   * every channel's block, filter scratch and FIR history are staged in LDS (1552 B per channel; 2064 B in the
//     ALS instantiation); carried state lives in HBM in per-channel rows (asdr_device.h) and is loaded/stored with
//     coalesced 16-B-per-lane accesses.
//
// Numerics: built with -ffp-contract=off.  Every float operation is a separately rounded binary32
// operation in the reference's order; the reference's "double islands"  are computed
// in binary64 here too, so the int16 output is bit-identical to the CPU restatement.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>
#include <type_traits>

#include "asdr_device.h"
#include "asdr_tables.h"
#include "../../include/asdr.h"

// ---- per-channel LDS layout (floats) --------------------------------------------------------------
// Two instantiations: STRIDE 388 (no channel of the batch uses the ALS filter; 11 waves/CU fit) and 516 (ALS).
// 388 = 97 sixteen-byte slots == 1 (mod 16): the 8 channels' rows start on different LDS slots.
#define W0 0       // working row A: I, later the mono audio
#define W1 128     // working row B: Q
#define PH 256     // mixer phase sequence [256,384)
#define XP 0       // Hilbert history x[1..383] in natural order at [0,383), overlays W0/W1/PH once the mixer has consumed them
#ifndef ASDR_STRIDE
#define ASDR_STRIDE 388
#endif
// Rows of the instantiations whose channels run a SHORT ALS filter on the compact overlay (als_small kinds, SAM + ALS post role): 392 floats,
// i.e. 8 banks from channel to channel -- the filter's 5-bank operand windows of neighbouring channels then never overlap (at 388, 4 banks
// apart, they do: 2-way).  One process per build: C4 -2.3 %, all-USB + ALS -2.2 %; the plain kinds lose 1.5 % at 392 and keep 388.
#ifndef ASDR_AGC_QUIET_PATH
#define ASDR_AGC_QUIET_PATH 1   /* 0: every block through the chunk loop (measurements) */
#endif
#ifndef ASDR_MW_OWN_STORES
#define ASDR_MW_OWN_STORES 1   /* the four-wave form: a chain's results go back to HBM from the channel's OWN wave (its lead lanes: stores beside its other state), not from the duty wave's 32 lanes */
#endif
#ifndef ASDR_AGC_LEAN
#define ASDR_AGC_LEAN 1   /* the four-wave form's AGC duty: the envelope-only chain when no hang counter can run out inside the block (see the duty) */
#endif
#ifndef ASDR_MW_AGC_PIPELINED
#define ASDR_MW_AGC_PIPELINED 0   /* ... and that chain BESIDE the audio duty, a chunk behind the cascades, while the bank attacks (see the audio duty).  Built, bit-exact, measured
                                     (round 6): the workgroup's timeline gets 3.5 k cycles shorter in a fresh bank, the launch does not get faster (driver window +0...2.8 %,
                                     steady state +2.3 %: the early decision, the progress words and the extra waves' instructions cost what the parking saved) -- off */
#endif
#ifndef ASDR_ONEBLK
#define ASDR_ONEBLK 1   /* one-block launches of the plain / short-ALS uniform kernels take their loop-free twins (asdr_launch_update); 0: the looped kernels (measurements) */
#endif
#ifndef ASDR_MW
#define ASDR_MW 1         /* large one-block direct launches of the plain kernel take the four-wave workgroup form (asdr_update_kernel_mw); 0: one wave per workgroup */
#endif
#ifndef ASDR_MW_SHARE
#define ASDR_MW_SHARE 7   /* what the four waves of an MW workgroup share: 1 audio cascades (2 waves x 16 channels), 2 blanker / phase chains, 4 AGC chain (1 wave x 32 channels) */
#endif
#ifndef ASDR_MIX_KEEP_Q
#define ASDR_MIX_KEEP_Q 1   /* uniform SSB waves keep the mixed Q in registers from the mixer to the Hilbert stage (no LDS round trip of the mixed rows); 0: through W0 / W1 */
#endif
#ifndef ASDR_UNIT_SCALE
#define ASDR_UNIT_SCALE 1   /* waves whose input gains are all 1.0 scale their samples with two binary32 operations (scale8); 0: always the binary64 form */
#endif
#ifndef ASDR_C16
#define ASDR_C16 0        /* large direct one-block launches of SSB-class groups take the 16-waves-per-CU kernel (asdr_update_kernel_c16); environment ASDR_C16 overrides */
#endif
#ifndef ASDR_C16_MIN_WAVES
#define ASDR_C16_MIN_WAVES 64
#endif
#ifndef ASDR_MW_MIN_WAVES
#define ASDR_MW_MIN_WAVES 64
#endif
#ifndef ASDR_MW_PRIO
#define ASDR_MW_PRIO 2   /* experiments: s_setprio <n> inside the duty sections of the four-wave form (three sibling waves wait for the duty wave) */
#endif
#ifndef ASDR_CHAIN_PRIO
#define ASDR_CHAIN_PRIO 0   /* experiments: s_setprio <n> around the dependent-chain phases of EVERY form (IF / audio pipelines, blanker / AGC chains) */
#endif
#define CHAIN_PRIO_ON() do { if (ASDR_CHAIN_PRIO) __builtin_amdgcn_s_setprio(ASDR_CHAIN_PRIO); } while (0)
#define CHAIN_PRIO_OFF() do { if (ASDR_CHAIN_PRIO) __builtin_amdgcn_s_setprio(0); } while (0)
#ifndef ASDR_ONEBLK_ALS
#define ASDR_ONEBLK_ALS 1   /* the loop-free form for the ALS instantiations too (short-filter uniform kernel, SAM post role with the filter); 0: measurements */
#endif
#ifndef ASDR_ONEBLK_ROLES
#define ASDR_ONEBLK_ROLES 1   /* the SAM pre / post and the two-launch ALS pre roles (always one block per launch) compiled without the block loop */
#endif
#ifndef ASDR_ALS_WINDOW
#define ASDR_ALS_WINDOW 1   /* the default-length ALS filter reads its operands once per FOUR tap sets (sliding windows in registers); 0: per tap set */
#endif
#ifndef ASDR_ALS_STRIDE
#define ASDR_ALS_STRIDE 392
#endif
#define ASDR_COMPACT_ROWS(stride) ((stride) < 500)   /* 388 / 392 against the long filters' 516 */
#define SCR0 387   // SAM lock flag (last, unused word of the AGC table row; never live together)
// noise-blanker overlay (dead before the rows above are written)
#define NB_B 2     // [2,180): beta*mag[t] of detection indices 78..255 (t = 0..177) from the envelope pass, overwritten in place by
                   // avg[t] (the running average BEFORE sample t) by the sequential pass; dead once the detection flags are in
                   // registers.  (Offset 2: the lanes' pieces t = 50 + 8 s8 + 64h and t = 8 s8 - 14 fall on 16-byte slots; words 0, 1 are padding.)
#define NB_MSKB 0  // general path only: 272 BYTE codes of mask[0..265(..271)] = 68 words, overlays the dead B row
#define NB_CB 68   // general path only: 56 words = 224 bytes of padded running detection counts: CB[3..23] = 0, CB[24+t] = count
                   // after detection index t (0..177), CB[202..222] = final count (also inside the dead B row: the PH row survives)
// AGC overlay
#define AGC_GV 128   // per sample: the envelope value whose compressor gain applies (-1 = gain carried in)
#define AGC_TAB 256  // this channel's gain table (row of 132 floats)
// ALS overlay: from word 128 (the layout constants are in the body's ALS section: they depend on the row length)
#define ALS_OUT W0   // in place: the input is copied to the overlay first

#define PI_D 3.1415926535897932384626433832795 /* Arduino.h PI (double) */

typedef float v4f __attribute__((ext_vector_type(4)));   // one aligned VGPR quad (operand of the exchange rings' inline-assembly accesses)   // one aligned VGPR pair: operand of v_pk_mul_f32 / v_pk_add_f32

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

// register budget of the main instantiation: 3 -> <=168 VGPRs (12 waves/CU with 13.4 KB LDS per wave)
#ifndef ASDR_WAVES_PER_EU
#define ASDR_WAVES_PER_EU 3
#endif

__constant__ float c_bq_pool[ASDR_N_BQ_TABLES][ASDR_BQ_COEFS];
__constant__ float c_hilbert[ASDR_HILBERT_TAPS];
__constant__ float c_sine[ASDR_SINE_TABLE_LEN];
__constant__ float c_mask_val[8];       // blanker mask: code -> value

extern "C" int asdr_kernels_upload_tables(void) {
  if (hipMemcpyToSymbol(HIP_SYMBOL(c_bq_pool), asdr_bq_pool, sizeof(asdr_bq_pool)) != hipSuccess) return -1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(c_hilbert), asdr_hilbert_taps, sizeof(asdr_hilbert_taps)) != hipSuccess) return -1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(c_sine), asdr_sine_table, sizeof(asdr_sine_table)) != hipSuccess) return -1;
  {   // blanker mask codes (AudioSDR.cpp:608, 623, 630: the mask only ever holds these seven values)
    const float val[8] = {0.0f, 1.0f, (float)0.933, (float)0.750, (float)0.500, (float)0.250, (float)0.067, 1.0f};
    if (hipMemcpyToSymbol(HIP_SYMBOL(c_mask_val), val, sizeof val) != hipSuccess) return -1;
  }
  return 0;
}

// Scheduling fences inside fully unrolled loops (they stop the scheduler from interleaving all iterations) were needed while
// the kernel spilled; at the current register budget the unfenced schedule is 4 % faster (profiles/README.md, round 2).
#ifdef ASDR_FENCE
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define SCHED_FENCE() do { } while (0)
#endif
#include "asdr_fir.h"
// ---- scalar helpers (AudioSDR.h:358-446) ------------------------------------------------------------
// Correctly rounded binary64 quotient x / c for a constant c with r = RN(1/c) (Markstein): q0 = RN(x*r),
// rem = x - c*q0 exactly (fma), q = RN(q0 + rem*r) == RN(x/c).  Replaces the ~30-instruction IEEE f64 division
// sequence by mul + 2 fma.  Verified exhaustively against true division by the CPU test-suite (all float32
// phases in [0, 2*pi] for c = (double)(float)(2*pi); all int16 for c = 32767.0) and on the GPU by the taps.
__device__ __forceinline__ double div_by_const(double x, double c, double r) {
  const double q0 = x * r;
  const double rem = __builtin_fma(-q0, c, x);
  return __builtin_fma(rem, r, q0);
}
// sin_f32's phase -> uint16 table phase (AudioSDR.h:362-364): intPhase = (long)(Phase * 65535.0 / twoPI) in binary64, after one
// conditional wrap each way.  The quotient as ONE multiply: q' = (double)phase * S with S = RN(RN(65535 / c) (1 + 2^-49)), c = (double)twoPI.
//   * The exact quotient q = P 2^a 65535 / (C 2^b) (P, C the 24-bit significands) is an integer n or at least 2^-40 q away from
//     one: |P 65535 2^a - n C 2^b| is a non-zero multiple of 2^min(a, b), which is >= q C 2^b 2^-40 in either case.
//   * q' lies in (q, q (1 + 2^-48)) for q > 0: the bias 2^-49 exceeds the three roundings (3 x 2^-53), so q' > q, and it stays far
//     inside the 2^-40 gap -- trunc(q') == trunc(q) whether or not q is an integer.  The reference's own RN(x / c) is within 2^-53
//     of q, so its truncation is trunc(q) as well.
// (Round 2 formed the correctly rounded quotient with a multiply and two fmas: two binary64 operations more per lookup, all of
// them on the PLL's dependent chain.)  Checked for every float32 phase in [0, 2 pi] by the CPU suite (oracle
// ao_check_sin_index_one_multiply).  -DASDR_SIN_INDEX_DIV: the round-2 form (measurements).
// BELOW_TWO_PI: the caller guarantees phase < twoPI (the PLL: |phase_est| < pi after its wrap, so phase_est and
// phase_est + pi/2 are below 4.72), which makes the first test dead -- three instructions less on the PLL's dependent chain.
// NONNEG: the caller guarantees phase >= 0 (the mixer: its phase stays in [0, twoPI], AudioSDR.h:514-517, and so does phase + pi/2),
// which makes the second test dead.
struct SinIndexK { double inv_two_pi, scale; };
template <bool BELOW_TWO_PI = false, bool NONNEG = false>
__device__ __forceinline__ uint32_t sin_index(float phase, float two_pi, SinIndexK k) {
  if (!BELOW_TWO_PI && phase >= two_pi) phase -= two_pi;
  if (!NONNEG && phase < 0.0f) phase += two_pi;
#ifdef ASDR_SIN_INDEX_DIV
  const double q = div_by_const((double)phase * 65535.0, (double)two_pi, k.inv_two_pi);
#else
  const double q = (double)phase * k.scale;
#endif
  return (uint32_t)(int)q & 0xFFFFu;
}
// AudioSDR.h:365-369: val1 + (((val2 - val1) * (float)delta) / 256.0).  The double divide-by-256 is exact and the
// double sum of two floats this close in exponent is exact, so the single final rounding equals the float32 sum
// (checked for all 65,536 table phases by the CPU test-suite).
// `sine` = the wave's LDS copy of the table, or nullptr = read the __constant__ table through L1 (saves 1 KB of LDS
// per wave; used when no channel runs the PLL, whose per-sample dependent lookups want the LDS latency).
__device__ __forceinline__ float sin_lut(const float *sine, uint32_t ip) {
  const uint32_t idx = ip >> 8, d = ip & 0xFFu;
  const float v1 = sine ? sine[idx] : c_sine[idx], v2 = sine ? sine[idx + 1] : c_sine[idx + 1];
  return v1 + ((v2 - v1) * (float)d) * (1.0f / 256.0f);
}
template <bool BELOW_TWO_PI = false>
__device__ __forceinline__ float sin_f32(const float *sine, float phase, float two_pi, SinIndexK inv_two_pi) {
  return sin_lut(sine, sin_index<BELOW_TWO_PI>(phase, two_pi, inv_two_pi));
}
// AudioSDR.h:375-377
template <bool BELOW_TWO_PI = false>
__device__ __forceinline__ float cos_f32(const float *sine, float phase, float two_pi, SinIndexK inv_two_pi, double half_pi_d) {
  return sin_f32<BELOW_TWO_PI>(sine, (float)((double)phase + half_pi_d), two_pi, inv_two_pi);
}
// cos_f32 / sin_f32 of NP phases (the mixer, AudioSDR.h:519-520): all 2 NP table phases first, then the 2 NP gathers TOGETHER, then the
// interpolations.  Written lookup by lookup the compiler waits for every gather right behind its issue -- one exposed L1 round
// trip per lookup.  Same operations per value as sin_f32 / cos_f32 above.
typedef const __attribute__((address_space(3))) float *lds_cfloat_ptr;
typedef __attribute__((address_space(3))) int *lds_int_ptr;
typedef int v2i __attribute__((ext_vector_type(2)));
// LDS_TAB: `sine` is the workgroup's LDS copy of the table (named as an LDS pointer: through the generic `sine ? sine : c_sine` the compiler emits FLAT
// loads, which wait for the outstanding global loads as well -- see sincos_pll).
template <int NP, bool LDS_TAB = false>
__device__ __forceinline__ void sincos_batch(const float *sine, const float *ph, float *cc, float *sn, float two_pi, SinIndexK inv_two_pi, double half_pi_d) {
  uint32_t ipc[NP], ips[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    ipc[j] = sin_index<false, true>((float)((double)ph[j] + half_pi_d), two_pi, inv_two_pi);   // cos_f32: sin_f32(phase + PI/2.0)
    ips[j] = sin_index<false, true>(ph[j], two_pi, inv_two_pi);   // (mixer phases are in [0, twoPI]: never negative)
  }
  float c1[NP], c2[NP], s1[NP], s2[NP];
  if constexpr (LDS_TAB) {
    lds_cfloat_ptr tab = (lds_cfloat_ptr)sine;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      lds_cfloat_ptr tc = tab + (ipc[j] >> 8), ts = tab + (ips[j] >> 8);
      c1[j] = tc[0]; c2[j] = tc[1]; s1[j] = ts[0]; s2[j] = ts[1];
    }
  } else {
  const float *tab = sine ? sine : c_sine;
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    const float *tc = tab + (ipc[j] >> 8), *ts = tab + (ips[j] >> 8);
    c1[j] = tc[0]; c2[j] = tc[1]; s1[j] = ts[0]; s2[j] = ts[1];
  }
  }
  __builtin_amdgcn_sched_barrier(0);   // (every request is out before the first value is waited for)
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    cc[j] = c1[j] + ((c2[j] - c1[j]) * (float)(ipc[j] & 0xFFu)) * (1.0f / 256.0f);   // sin_lut
    sn[j] = s1[j] + ((s2[j] - s1[j]) * (float)(ips[j] & 0xFFu)) * (1.0f / 256.0f);
  }
}
// The PLL's y = (cos_f32, sin_f32)(phase_est) (AudioSDR.cpp:737-738) from the wave's LDS copy of the table, |phase| < pi.  `sine` is
// named as an LDS pointer: through the generic `sine ? sine : c_sine` form of sin_lut the compiler emits FLAT loads, which count
// against the vector-memory counter as well -- every lookup then waited for the PLL kernel's outstanding global sample loads too.
// Both table phases first, both reads together, then the interpolations (same operations per value as sin_f32 / cos_f32).
__device__ __forceinline__ void sincos_pll(const float *sine, float phase, float &c, float &s, float two_pi, SinIndexK inv_two_pi, double half_pi_d) {
  const uint32_t ipc = sin_index<true>((float)((double)phase + half_pi_d), two_pi, inv_two_pi);
  const uint32_t ips = sin_index<true>(phase, two_pi, inv_two_pi);
  lds_cfloat_ptr tab = (lds_cfloat_ptr)sine;
  const float c1 = tab[ipc >> 8], c2 = tab[(ipc >> 8) + 1], s1 = tab[ips >> 8], s2 = tab[(ips >> 8) + 1];
  c = c1 + ((c2 - c1) * (float)(ipc & 0xFFu)) * (1.0f / 256.0f);
  s = s1 + ((s2 - s1) * (float)(ips & 0xFFu)) * (1.0f / 256.0f);
}
// AudioSDR.h:384-408
__device__ __forceinline__ float approx_atan(float z) {
  const float n1 = 0.97239411f, n2 = -0.19194795f;
  return (n1 + n2 * z * z) * z;
}
// The reference's branch tree (quadrant by quadrant, one division per branch) evaluated branch-free with ONE division:
// every lane performs exactly the operations of the branch it would have taken (the operands of the division are
// selected first; -a - half_pi == -a + (-half_pi) and a - PI == a + (-PI) exactly), the other results are discarded.
template <bool TWO_SUMS>
__device__ __forceinline__ float approx_atan2(float y, float x, float half_pi) {
  const bool xnz = (x != 0.0f);
  const bool big = fabsf(x) > fabsf(y);
  const float num = big ? y : x, den = big ? x : y;
  const float z = num / den;
  const float a = approx_atan(z);
  // a + PI for y >= 0, a - PI otherwise (a - PI == a + (-PI) exactly).  The select between the two binary64 constants parks
  // their three distinct words in VGPRs for the whole kernel: the SAM instantiation affords them (this is its longest dependent
  // chain: two instructions fewer per sample), the ALS instantiations (TWO_SUMS) form both sums and select the result.
  float a_pi;
  if (TWO_SUMS) {
    const double ad = (double)a;
    const float a_pp = (float)(ad + PI_D);
    float a_pm = (float)(ad - PI_D);
    asm volatile("" : "+v"(a_pm));   // (keeps the compiler from folding the two sums back into that select)
    a_pi = (y >= 0.0f) ? a_pp : a_pm;
  } else {
    const double pis = (y >= 0.0f) ? PI_D : -PI_D;
    a_pi = (float)((double)a + pis);
  }
  const float r_big = (x > 0.0f) ? a : a_pi;                               // |x| > |y|
#ifdef ASDR_ATAN2_R4
  const float r_small = -a + ((y > 0.0f) ? half_pi : -half_pi);            // |x| <= |y|, x != 0
  const float r_x0 = (y > 0.0f) ? half_pi : ((y < 0.0f) ? -half_pi : 0.0f);   // x == 0
  return xnz ? (big ? r_big : r_small) : r_x0;
#else
  // |x| <= |y|, x != 0: y is not zero there, so its sign bit says which half_pi (one v_bfi instead of compare + select).  And the x == 0 branch
  // needs no arithmetic of its own (round 5): with y != 0 it is the value above -- z = 0 / y = +-0, a = +-0, -a + (+-half_pi) = +-half_pi --
  // and with y == 0 or NaN (x == 0 is num == 0 there; `!(|den| > 0)` is the reference's "neither y > 0 nor y < 0") it is 0.  Three vector
  // instructions less per PLL sample; equal bit for bit, NaNs included, on 4 x 10^8 random and special operand pairs (CPU, both forms in C).
  (void)xnz;
  const float r_small = -a + __builtin_copysignf(half_pi, y);
  const float r = big ? r_big : r_small;
  return (num == 0.0f && !(fabsf(den) > 0.0f)) ? 0.0f : r;
#endif
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
// The same with the IEEE division x / out (an 11-instruction sequence) replaced by v_rcp_f32 + one Newton step on the reciprocal + one
// residual correction of the quotient (6 instructions): bit-identical RESULTS for every x >= 7.6e-32 and for x = 0, checked over all 2^31
// non-negative finite floats on the GPU (tools/ubench/sqrt_div_check.hip; tests/test_gpu_parity.py runs it).  Used where x = I^2 + Q^2 of
// samples scaled with gain exactly 1.0 (x = 0 or x >= 9.3e-10: the unit-gain waves of scale8).
__device__ __forceinline__ float fast_sqrt1_short(float x) {
  uint32_t i = __float_as_uint(x);
  i -= 1u << 23;
  i >>= 1;
  i += 1u << 29;
  const float out = __uint_as_float(i);
  const float r0 = __builtin_amdgcn_rcpf(out);
  const float r = __builtin_fmaf(__builtin_fmaf(-out, r0, 1.0f), r0, r0);
  const float q0 = x * r;
  const float e = __builtin_fmaf(-out, q0, x);
  const float q1 = __builtin_fmaf(e, r, q0);
  return 0.5f * (out + q1);
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
  // bound_ctrl = true: the lane without a source (lane 0 of a 16-lane row; always a stage-0 lane, which ignores the value)
  // reads 0 and no `old` operand has to be materialised, so the move folds into its consumer (v_cndmask_b32_dpp)
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xF, 0xF, true));
}

// ---- 4-stage DF1 biquad cascade as a lane pipeline -------------------------------------------------
// Restates CMSIS-DSP arm_biquad_cascade_df1_f32 (arm_math.h:1360-1378; call sites AudioSDR.cpp:77-78,
// 136-137, 285): per stage, per sample  acc = b0*x; acc += b1*x1; acc += b2*x2; acc += a1*y1; acc += a2*y2
// (separately rounded, this association).  The reference runs stage-major over the block; each stage is causal,
// so running the four stages on four adjacent lanes with one CHUNK (4 or 8 samples) of skew produces identical values.
// At chunk-step c lane `st` handles samples C(c-st)..C(c-st)+C-1: stage 0 reads them from the LDS row
// (ds_read_b128, prefetched a step ahead), stage k>0 takes the previous lane's four outputs of the previous step
// through DPP row_shr:1, stage 3 writes back in place (ds_write_b128).  The y-independent part
// p = (b0*x + b1*x1) + b2*x2 of all four samples is off the critical path; the recurrence is 3 dependent ops/sample.
#ifndef ASDR_PIPE_PK_MASK
#define ASDR_PIPE_PK_MASK 8   /* which kernel kinds run the pipelines' y-independent products packed (see PIPE_PK in the body): the block pipeline's role waves (alone on their
                                  SIMDs, where a packed instruction costs an issue slot like any other: C5 share -1 %); everywhere else measured equal or slower */
#endif
#ifndef ASDR_PIPE_PREFETCH_FENCE
#define ASDR_PIPE_PREFETCH_FENCE 0   /* measured: no gain (profiles/README.md, round 5) */
#endif
#ifndef ASDR_PIPE_CHUNK
#define ASDR_PIPE_CHUNK 8   /* samples per lane per pipeline step: 4 (35 steps) or 8 (19 steps, less per-step overhead) */
#endif
#ifndef ASDR_PIPE_BANK_SELECT
#define ASDR_PIPE_BANK_SELECT 0   /* the stage-0 select of the pipelines as a bank-masked DPP move (see the step); 0: round 5's v_cndmask_b32_dpp on VCC */
#endif
// SIGNAL (round 6, the four-wave form's audio duty): after every step that wrote a chunk back, the lane `sig` publishes the number of
// chunks of the row that are final in *prog (release at workgroup scope) -- the AGC duty wave follows one chunk behind.
template <bool PK = false, bool SIGNAL = false>
__device__ __forceinline__ void biquad_pipe(float *row, bool on, int st, const float *cf, float *sv, int *prog = nullptr, bool sig = false) {
  constexpr int C = ASDR_PIPE_CHUNK, NSTEP = ASDR_N / C + 3;
  const float b0 = cf[0], b1 = cf[1], b2 = cf[2], a1 = cf[3], a2 = cf[4];
  float x1 = sv[0], x2 = sv[1], y1 = sv[2], y2 = sv[3];
  float yo[C], xn[C];
#pragma unroll
  for (int j = 0; j < C; ++j) yo[j] = 0.0f;
#pragma unroll
  for (int q = 0; q < C / 4; ++q) { const float4 t = reinterpret_cast<const float4 *>(row)[q]; xn[4 * q] = t.x; xn[4 * q + 1] = t.y; xn[4 * q + 2] = t.z; xn[4 * q + 3] = t.w; }
  const bool s0 = (st == 0);
  const unsigned long long s0m = __ballot(s0);   // EXEC is full here
  uint32_t prog_lds = 0u;
  if constexpr (SIGNAL) prog_lds = (uint32_t)(uintptr_t)(lds_int_ptr)prog;
  auto step = [&](auto edge_tag, const int c) {
    constexpr bool EDGE = decltype(edge_tag)::value;   // a step of the systolic fill / drain: some stages have no chunk
    const int cn = c - st;
    float x[C], p[C], y[C];
#if ASDR_PIPE_BANK_SELECT
    // Round 6.  Every caller runs stage `st` on the lanes (lane & 15) >> 2 == st (PipeMap below): the four cascades of a row of 16 lanes have
    // their stage-0 lanes in BANK 0 (lanes 0..3 of the row), stage s hands its chunk to the lane FOUR places on.  The DPP move `row_shr:4` writes
    // banks 1..3 only (bank_mask 0xE) on top of the LDS chunk, which stays where it is on the stage-0 lanes -- one v_mov_b32_dpp per sample, no
    // mask register.  (The v_cndmask_b32_dpp form below reads VCC: a run of eight of them costs a lone wave 3 x and a busy SIMD 7 x a
    // plain instruction each -- profiles/r05_op_cost.txt -- and under three resident waves that run was a third of a pipeline step.)
#pragma unroll
    for (int j = 0; j < C; ++j) x[j] = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(xn[j]), __float_as_int(yo[j]), 0x114, 0xF, 0xE, false));   // row_shr:4, banks 1..3
#elif defined(ASDR_PIPE_PLAIN_SELECT) || ASDR_PIPE_CHUNK != 8
#pragma unroll
    for (int j = 0; j < C; ++j) { const float d = dpp_row_shr1(yo[j]); x[j] = s0 ? xn[j] : d; }
#else
    // x[j] = stage 0 ? the LDS chunk : the previous lane's output, as ONE instruction per sample: v_cndmask_b32 with the DPP
    // shift on its first source (VOP2 form: the condition must be VCC).  The compiler keeps the stage-0 mask in an SGPR pair and
    // emits v_mov_b32_dpp + v_cndmask_b32_e64 (16 instead of 8 per step), hence the assembly.
    asm("s_mov_b64 vcc, %[m]\n\ts_nop 1\n\t"
        "v_cndmask_b32_dpp %[x0], %[y0], %[n0], vcc row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_cndmask_b32_dpp %[x1], %[y1], %[n1], vcc row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_cndmask_b32_dpp %[x2], %[y2], %[n2], vcc row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_cndmask_b32_dpp %[x3], %[y3], %[n3], vcc row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_cndmask_b32_dpp %[x4], %[y4], %[n4], vcc row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_cndmask_b32_dpp %[x5], %[y5], %[n5], vcc row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_cndmask_b32_dpp %[x6], %[y6], %[n6], vcc row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_cndmask_b32_dpp %[x7], %[y7], %[n7], vcc row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
        : [x0] "=&v"(x[0]), [x1] "=&v"(x[1]), [x2] "=&v"(x[2]), [x3] "=&v"(x[3]), [x4] "=&v"(x[4]), [x5] "=&v"(x[5]), [x6] "=&v"(x[6]), [x7] "=&v"(x[7])
        : [y0] "v"(yo[0]), [y1] "v"(yo[1]), [y2] "v"(yo[2]), [y3] "v"(yo[3]), [y4] "v"(yo[4]), [y5] "v"(yo[5]), [y6] "v"(yo[6]), [y7] "v"(yo[7]),
          [n0] "v"(xn[0]), [n1] "v"(xn[1]), [n2] "v"(xn[2]), [n3] "v"(xn[3]), [n4] "v"(xn[4]), [n5] "v"(xn[5]), [n6] "v"(xn[6]), [n7] "v"(xn[7]),
          [m] "s"(s0m)
        : "vcc");
#endif
    {   // prefetch the next chunk for stage 0
      const int nc = (c + 1 < ASDR_N / C) ? c + 1 : ASDR_N / C - 1;
#pragma unroll
      for (int q = 0; q < C / 4; ++q) { const float4 t = reinterpret_cast<const float4 *>(row)[nc * (C / 4) + q]; xn[4 * q] = t.x; xn[4 * q + 1] = t.y; xn[4 * q + 2] = t.z; xn[4 * q + 3] = t.w; }
    }
    const uint32_t prog_lds_ = prog_lds;   // (named outside the `if constexpr`: a generic lambda does not capture from inside it)
    if constexpr (SIGNAL) {
      // The chunks the PREVIOUS steps wrote back are final: c - 3 of them.  Published here, at the top of the step, so that the wait in front of the
      // next step's select has a step's arithmetic between it and this write (at the end of the step it sat out the write: +60 cycles per step).  No
      // release fence: the LDS executes a wave's operations in the order issued, and the "memory" clobber keeps the compiler from moving the row
      // stores of the previous step below this one.
      if (c >= 4 && sig) asm volatile("ds_write_b32 %0, %1" :: "v"(prog_lds_), "v"(c - 3) : "memory");
    }
#if ASDR_PIPE_PREFETCH_FENCE
    // Round 5: the two LDS reads stay HERE, at the top of the step.  Left alone the scheduler sinks them to the step's last dozen
    // instructions (their eight result registers are not needed before the next step), and the `s_waitcnt lgkmcnt(0)` at the top of the
    // next step then sits out most of an LDS round trip -- once per step, 38 steps per block.
    __builtin_amdgcn_sched_barrier(0);
#endif
#if ASDR_PIPE_CHUNK == 8
    if constexpr (PK) {
    // Packed form (same separately rounded operations): samples j and j + 4 share a register pair, so that x[j-1] and x[j-2] of
    // both are the pairs one and two places back -- every product of the y-independent part is an aligned v_pk_mul_f32.
    {
      const v2f B0 = (v2f){b0, b0}, B1 = (v2f){b1, b1}, B2 = (v2f){b2, b2};
      v2f P[6];   // P[k] = (x[k-2], x[k+2]): k = 0, 1 hold the carried x2, x1 in their low halves
      P[0] = (v2f){x2, x[2]}; P[1] = (v2f){x1, x[3]};
#pragma unroll
      for (int k = 0; k < 4; ++k) P[2 + k] = (v2f){x[k], x[k + 4]};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        v2f t = B0 * P[2 + k]; t += B1 * P[1 + k]; t += B2 * P[k];
        p[k] = t.x; p[k + 4] = t.y;
      }
    }
    } else
#endif
    {
    // y-independent partial sums p = (b0*x + b1*x[-1]) + b2*x[-2], in the reference's association
#pragma unroll
    for (int j = 0; j < C; ++j) {
      const float xm1 = (j >= 1) ? x[j - 1] : x1, xm2 = (j >= 2) ? x[j - 2] : ((j == 1) ? x1 : x2);
      float t = b0 * x[j]; t += b1 * xm1; t += b2 * xm2;
      p[j] = t;
    }
    }
    // recurrence: y = (p + a1*y[-1]) + a2*y[-2]
#pragma unroll
    for (int j = 0; j < C; ++j) {
      const float ym1 = (j >= 1) ? y[j - 1] : y1, ym2 = (j >= 2) ? y[j - 2] : ((j == 1) ? y1 : y2);
      float t = p[j] + a1 * ym1; t += a2 * ym2;
      y[j] = t;
    }
    const bool act = !EDGE || ((cn >= 0) && (cn < ASDR_N / C));
    if (act) { x1 = x[C - 1]; x2 = x[C - 2]; y1 = y[C - 1]; y2 = y[C - 2]; }
#pragma unroll
    for (int j = 0; j < C; ++j) yo[j] = y[j];
    if (on && act && st == 3) {
#pragma unroll
      for (int q = 0; q < C / 4; ++q) reinterpret_cast<float4 *>(row)[cn * (C / 4) + q] = make_float4(y[4 * q], y[4 * q + 1], y[4 * q + 2], y[4 * q + 3]);
    }
  };
#if defined(ASDR_PIPE_ONE_LOOP) || ASDR_PIPE_CHUNK != 8
#pragma unroll 1
  for (int c = 0; c < NSTEP; ++c) step(std::true_type{}, c);
#else
  // Steps 3 .. 15 have a chunk for every stage: no activity test, no predicated state update (round 2 ran all 19 steps through the
  // general form: 9 instructions of 102 per step).  Two steps per trip, so that the carried state is a register renaming.
#pragma unroll 1
  for (int c = 0; c < 3; ++c) step(std::true_type{}, c);
#pragma unroll 1
  for (int c = 3; c < 15; c += 2) { step(std::false_type{}, c); step(std::false_type{}, c + 1); }
  step(std::false_type{}, 15);
#pragma unroll 1
  for (int c = 16; c < NSTEP; ++c) step(std::true_type{}, c);
#endif
  if constexpr (SIGNAL) { if (sig) asm volatile("ds_write_b32 %0, %1" :: "v"(prog_lds), "v"(ASDR_N / C) : "memory"); }   // ... and the last one
  sv[0] = x1; sv[1] = x2; sv[2] = y1; sv[3] = y2;
}

__device__ __forceinline__ void load16(const float *p, float *v) {
  const float4 *q = reinterpret_cast<const float4 *>(p);
#pragma unroll
  for (int r = 0; r < 4; ++r) { float4 t = q[r]; v[4 * r] = t.x; v[4 * r + 1] = t.y; v[4 * r + 2] = t.z; v[4 * r + 3] = t.w; }
}
__device__ __forceinline__ void load8(const float *p, float *v) {
  const float4 *q = reinterpret_cast<const float4 *>(p);
  const float4 a = q[0], b = q[1];
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void load4(const float *p, float *v) {
  const float4 a = *reinterpret_cast<const float4 *>(p);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
}
__device__ __forceinline__ void store4(float *p, const float *v) { *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]); }
// the same as a non-temporal (streaming) store: `global_store_dwordx4 ... nt`
__device__ __forceinline__ void store4_nt(float *p, const float *v) {
  typedef float f4v __attribute__((ext_vector_type(4)));
  const f4v tv = {v[0], v[1], v[2], v[3]};
  __builtin_nontemporal_store(tv, reinterpret_cast<f4v *>(p));
}
#ifndef ASDR_NT_LOADS
#define ASDR_NT_LOADS 2   /* non-temporal loads for rows a launch reads once and nobody reads again: 1 = input rows, 2 = + the blanker ring's oldest block (-1.5 % on C2), 3 = + the Hilbert ring's oldest block (no further gain) */
#endif
__device__ __forceinline__ int4 load_int4_nt(const int4 *p) {
  typedef int i4v __attribute__((ext_vector_type(4)));
  const i4v t = __builtin_nontemporal_load(reinterpret_cast<const i4v *>(p));
  return make_int4(t[0], t[1], t[2], t[3]);
}
__device__ __forceinline__ void load4_nt(const float *p, float *v) {
  typedef float f4v __attribute__((ext_vector_type(4)));
  const f4v t = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(p));
  v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
}
__device__ __forceinline__ void store_int4_nt(int4 *p, int4 v) {
  typedef int i4v __attribute__((ext_vector_type(4)));
  const i4v tv = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(tv, reinterpret_cast<i4v *>(p));
}
// the even entries of eight (the 16-waves-per-CU form keeps the mixer's phases of even samples only)
__device__ __forceinline__ void store4e(float *p, const float *v) { *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[2], v[4], v[6]); }
__device__ __forceinline__ void store8(float *p, const float *v) {
  float4 *q = reinterpret_cast<float4 *>(p);
  q[0] = make_float4(v[0], v[1], v[2], v[3]); q[1] = make_float4(v[4], v[5], v[6], v[7]);
}

// RN(x / 32767.0) in binary64 for an int16 x with ONE multiply and ONE fma.  1/(2^15 - 1) is the bit pattern 2^-15 repeated
// every 15 bits, so its binary64 rounding r = 0x1.0002000400080p-15 leaves the tail 1/32767 - r = r * 2^-60 * (1 + 2^-60 + ..);
// x*r + RN(x * r*2^-60) differs from x/32767 by < 2^-118 |x|, and for none of the 65,536 inputs does that move the rounding
// (checked exhaustively against true division by the CPU test-suite: oracle ao_check_scale_division).
__device__ __forceinline__ double div_i16_by_32767(double x) {
  return __builtin_fma(x, 0x1.0002000400080p-15, x * 0x1.0002000400080p-75);
}
// AudioSDR.cpp:68-69 for 8 raw samples: ((float)s / 32767.0) * gain in binary64, rounded to float by the store
// UNIT (wave-uniform: every gain of the wave's rows is exactly 1.0f, the reference's default, AudioSDR.h:172-175): the binary64 quotient
// rounded to binary32 as TWO binary32 operations -- fmaf(x, rh, x * rl), rh = 0x1.0002p-15f and rl = 0x1.0002p-45f being the first four
// copies of the 15-bit pattern of 1/32767 -- instead of five binary64-rate ones (convert, multiply, fma, multiply by the gain,
// convert): equal for all 65,536 inputs (oracle ao_check_scale_unit_gain); the product with 1.0 is exact.  Round 5.
__device__ __forceinline__ void scale8(const int16_t *s, double g, float *out, bool unit = false) {
  if (unit) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float x = (float)s[j]; out[j] = __builtin_fmaf(x, 0x1.0002p-15f, x * 0x1.0002p-45f); }
    return;
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
#ifdef ASDR_OLD_SCALE
    out[j] = (float)(div_by_const((double)s[j], 32767.0, 1.0 / 32767.0) * g);
#else
    out[j] = (float)(div_i16_by_32767((double)s[j]) * g);
#endif
    if ((j & 1) == 1) SCHED_FENCE();
  }
}
// The blanker mask only ever holds {0, 1, .933, .750, .500, .250, .067} (AudioSDR.cpp:608, 623, 630): one byte each in HBM
// (codes 0..6 in that order).  Neither direction touches memory (round 2: gathers through L1 in both directions were half of
// what the blanker's general path costs over the quiet one):
// Code -> value: every lane holds c_mask_val[lane & 7] (`tabv`, one load per wave on the general path); a lookup is a
//   ds_bpermute from lane `code` -- the LDS crossbar, no LDS memory.  (Select chains on the seven literals made the compiler keep
//   ~30 literals in VGPRs for the whole kernel.)
// Value -> code: bits 21..25 of the float are distinct for the seven values (0.0: 0, .067: 12, .25: 20, .5: 24, .75: 26,
//   .933: 27, 1.0: 28); three times that, less 36 (mod 64: 0.0 lands on 28), indexes 3-bit fields of one 64-bit constant.
//   The mask rows are only ever written with these seven values (decoded codes, the 1.0 fill, the 0.0 zeroing, the ramp
//   read from the same table), so no other bit pattern can arrive here.
#define ASDR_MASK_MAGIC 0x14c4005000006ull   /* fields: [0] = 6, [24] = 5, [36] = 4, [42] = 3, [45] = 2, [48] = 1, [28] = 0 */
// `group4` = 4 x the first lane of the caller's group of 8 lanes: a lane reads from its OWN group (the lanes of one channel are
// active together; ds_bpermute returns 0 for an inactive source lane).  Call it with the whole group active.
__device__ __forceinline__ float mask_decode_byte(float tabv, uint32_t group4, uint32_t word, int k) {   // byte k of a code word
  const uint32_t sh = (k == 0) ? (word << 2) : (word >> (8 * k - 2));
  const uint32_t addr = (sh & 0x1Cu) | group4;   // (code * 4: codes < 8)
  return __int_as_float(__builtin_amdgcn_ds_bpermute((int)addr, __float_as_int(tabv)));
}
__device__ __forceinline__ uint32_t mask_encode(float v) {
  const uint32_t idx = (__float_as_uint(v) >> 21) & 0x1Fu;
  return (uint32_t)(ASDR_MASK_MAGIC >> ((idx * 3u - 36u) & 63u)) & 7u;
}
__device__ __forceinline__ uint32_t mask_encode4(float4 mv) {   // four entries -> one code word
  const uint32_t c0 = mask_encode(mv.x), c1 = mask_encode(mv.y), c2 = mask_encode(mv.z), c3 = mask_encode(mv.w);
  return c0 | (c1 << 8) | (c2 << 16) | (c3 << 24);
}

// ALS FIR (AudioSDR.cpp:331-335): y = sum_{q < M} w[q] * x[top - q], q ascending, every product and sum separately rounded.
// The taps sit in LDS de-interleaved: even taps at w[0..WH), odd taps at w[WH..2 WH) (ALS_TAP; WH = 64, or 32 in the compact
// layout of the 388-float rows, see the ALS section of the body), so that the two halves of a
// channel's eight lanes can each fetch four of "their" taps with one ds_read_b128 (als_dot_split below).
#define ALS_TAP(q, wh) ((((q) & 1) ? (wh) : 0) + ((q) >> 1))
// CHECKED = false requires every index top - q (q < M) to lie inside the 256-sample history; CHECKED = true reads 0.0
// outside it (DESIGN.md "defined differences").  One lane does the whole sum (static taps, or parameters outside the safe range).
template <bool CHECKED, int WH>
__device__ __forceinline__ float als_dot(const float *w, const float *x, int top, int M) {
  float y = 0.0f;
  int q = 0;
#pragma unroll 2
  for (; q + 4 <= M; q += 4) {
    const float2 we = *reinterpret_cast<const float2 *>(w + (q >> 1)), wo = *reinterpret_cast<const float2 *>(w + WH + (q >> 1));
    const int t = top - q;
    float x0, x1, x2, x3;
    if (CHECKED) {
      x0 = (t >= 0 && t < 256) ? x[t] : 0.0f; x1 = (t - 1 >= 0 && t - 1 < 256) ? x[t - 1] : 0.0f;
      x2 = (t - 2 >= 0 && t - 2 < 256) ? x[t - 2] : 0.0f; x3 = (t - 3 >= 0 && t - 3 < 256) ? x[t - 3] : 0.0f;
    } else {
      x0 = x[t]; x1 = x[t - 1]; x2 = x[t - 2]; x3 = x[t - 3];
    }
    float p;
    p = we.x * x0; y += p; p = wo.x * x1; y += p; p = we.y * x2; y += p; p = wo.y * x3; y += p;
  }
  for (; q < M; ++q) {
    const int t = top - q;
    const float xv = CHECKED ? ((t >= 0 && t < 256) ? x[t] : 0.0f) : x[t];
    const float p = w[ALS_TAP(q, WH)] * xv;
    y += p;
  }
  return y;
}

// lane + 4 of the same row of 16 (the other half of this channel's eight lanes); folds into the consuming v_add_f32 as a DPP operand
__device__ __forceinline__ float dpp_row_shl4(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x104, 0xF, 0xF, true));
}

// The same sum for the adaptive filter's four samples per tap set, on all eight lanes of the channel: lanes 0-3 hold the four
// samples and lanes 4-7 the same four again; the low half multiplies the even taps, the high half the odd taps, and the low half
// adds both products in tap order (its own, then the partner's through a DPP operand).  Same products, same additions, same
// order -- half the multiplies and loads per lane.  The result is valid on lanes 0-3 only.  Taps past M contribute a +0.0
// (the running sum starts at +0.0 and so is never -0.0: adding +0.0 leaves it unchanged); their operands are read but not used.
// Requires the unchecked index range of als_dot<false>.  `h` = lane half (0 / 1), `xh` = x + top - h, `wh` = w + WH h.
struct AlsGroup { float4 w; float x0, x1, x2, x3; };   // one half's operands of a group of eight taps
__device__ __forceinline__ void als_group_load(AlsGroup &G, const float *wh, const float *xh, int g) {
  G.w = *reinterpret_cast<const float4 *>(wh + 4 * g);
  G.x0 = xh[-8 * g]; G.x1 = xh[-8 * g - 2]; G.x2 = xh[-8 * g - 4]; G.x3 = xh[-8 * g - 6];
}
__device__ __forceinline__ float als_group_sum(float y, const AlsGroup &G) {
  const float p0 = G.w.x * G.x0, p1 = G.w.y * G.x1, p2 = G.w.z * G.x2, p3 = G.w.w * G.x3;   // (as two v_pk_mul_f32: no gain, profiles/README.md)
  y += p0; y += dpp_row_shl4(p0); y += p1; y += dpp_row_shl4(p1);
  y += p2; y += dpp_row_shl4(p2); y += p3; y += dpp_row_shl4(p3);
  return y;
}
__device__ __forceinline__ float als_group_sum_tail(float y, const AlsGroup &G, int rem) {   // this half's taps exist for 2t < rem
  float p0 = G.w.x * G.x0, p1 = G.w.y * G.x1, p2 = G.w.z * G.x2, p3 = G.w.w * G.x3;
  p0 = (0 < rem) ? p0 : 0.0f; p1 = (2 < rem) ? p1 : 0.0f; p2 = (4 < rem) ? p2 : 0.0f; p3 = (6 < rem) ? p3 : 0.0f;
  y += p0; y += dpp_row_shl4(p0); y += p1; y += dpp_row_shl4(p1);
  y += p2; y += dpp_row_shl4(p2); y += p3; y += dpp_row_shl4(p3);
  return y;
}
// M per lane (channels of different filter lengths in one wave): a divergent loop, one group at a time
__device__ __forceinline__ float als_dot_split(const float *wh, const float *xh, int M, int h) {
  float y = 0.0f;
  const int n_full = M >> 3;
  AlsGroup A;
  for (int g = 0; g < n_full; ++g) { als_group_load(A, wh, xh, g); y = als_group_sum(y, A); }
  if (M & 7) { als_group_load(A, wh, xh, n_full); y = als_group_sum_tail(y, A, (M & 7) - h); }
  return y;
}
// M the same in the whole wave (a scalar): software-pipelined over two operand sets -- group g + 1's operands are requested
// before group g's sums, so the LDS round trip is off the chain of dependent additions.  The request past the last group reads
// inside the row and is dropped.
__device__ __forceinline__ float als_dot_split_uniform(const float *wh, const float *xh, int M, int h) {
  float y = 0.0f;
  const int n_full = M >> 3, rem = (M & 7) - h;
  AlsGroup A, B;
  als_group_load(A, wh, xh, 0);
  int g = 0;
#pragma unroll 1
  for (; g + 2 <= n_full; g += 2) {
    als_group_load(B, wh, xh, g + 1);
    __builtin_amdgcn_sched_barrier(0);   // (the scheduler would sink the requests below the sums to reuse their registers)
    y = als_group_sum(y, A);
    als_group_load(A, wh, xh, g + 2);
    __builtin_amdgcn_sched_barrier(0);
    y = als_group_sum(y, B);
  }
  if (g < n_full) {
    als_group_load(B, wh, xh, g + 1);
    __builtin_amdgcn_sched_barrier(0);
    y = als_group_sum(y, A);
    if (M & 7) y = als_group_sum_tail(y, B, rem);
  } else if (M & 7) {
    y = als_group_sum_tail(y, A, rem);
  }
  return y;
}

// ... and for a filter length known at compile time (the reference's default, ALS_M_DEFAULT = 55 taps, AudioSDR.h:198): fully
// unrolled -- immediate LDS offsets instead of pointer arithmetic, no loop control, the last group's existence tests folded.
#define ALS_M_DEFAULT 55
template <int MC>
__device__ __forceinline__ float als_dot_split_const(const float *wh, const float *xh, int h) {
  constexpr int NF = MC >> 3, R = MC & 7, NG = NF + (R ? 1 : 0);
  float y = 0.0f;
  AlsGroup G[2];
  als_group_load(G[0], wh, xh, 0);
#pragma unroll
  for (int g = 0; g < NF; ++g) {
    if (g + 1 < NG) als_group_load(G[(g + 1) & 1], wh, xh, g + 1);
    __builtin_amdgcn_sched_barrier(0);
    y = als_group_sum(y, G[g & 1]);
  }
  if (R) y = als_group_sum_tail(y, G[NF & 1], R - h);
  return y;
}
// The tap update w[q] += lambda * (e * x[iu - D - q]) (AudioSDR.cpp:341-343) of lane s8's taps q = s8 + 8k, for MC taps in all:
// every operand requested before the first product, every result written after the last.
template <int MC>
__device__ __forceinline__ void als_tap_update_const(const float *xq, float *wq, float e, float lam, int s8) {
  constexpr int NF = MC >> 3, R = MC & 7, NK = NF + (R ? 1 : 0);
  float xv[NK], wv[NK];
#pragma unroll
  for (int k = 0; k < NK; ++k) { xv[k] = xq[-8 * k]; wv[k] = wq[4 * k]; }
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const float gq = e * xv[k]; const float dq = lam * gq; const float wn = wv[k] + dq;
    wv[k] = (k < NF || s8 < R) ? wn : wv[k];
  }
#pragma unroll
  for (int k = 0; k < NK; ++k) wq[4 * k] = wv[k];
}

// keeps the instruction scheduler from interleaving all iterations of a fully unrolled loop (register pressure)
// A register array that is written under a condition and read later under the same condition must still be DEFINED on every
// path of the block-loop iteration: otherwise its value "from the previous iteration" is formally live around the whole loop
// body and pins its registers everywhere (tools/isa_liveness.py found 32 such VGPRs).
#define DEFINE_ALL_PATHS(arr, n) do { _Pragma("unroll") for (int z_ = 0; z_ < (n); ++z_) (arr)[z_] = 0.0f; } while (0)
// (Also kept in the uniform-key instantiations, where the conditions are scalar branches: without it they spill 216 VGPRs.)
// Per-channel rows addressed as (uniform base) + (32-bit byte offset): the offset stays in ONE VGPR and the access uses
// the scalar-base addressing mode, instead of a 64-bit pointer in two VGPRs per array.  Valid because every state array is
// smaller than 4 GiB (1,048,577 rows of at most 1.5 KiB).
template <typename T>
__device__ __forceinline__ T *row_ptr(T *base, uint32_t byte_off) {
  return reinterpret_cast<T *>(reinterpret_cast<char *>(base) + byte_off);
}
template <typename T>
__device__ __forceinline__ const T *row_ptr(const T *base, uint32_t byte_off) {
  return reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + byte_off);
}
// Hand-off between the lanes of ONE wave through LDS / HBM: outstanding memory operations drained and no code motion across it.
// (A wave's LDS operations execute in order, so no s_barrier is needed; the multi-wave SAM instantiation uses a real
// __syncthreads() only around its shared PLL phase.)
#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); \
                         __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); } while (0)

// =====================================================================================================
// Pointwise stages work LDS-resident in two 8-sample pieces per lane (`#pragma unroll 1` loops), so no 16-wide
// register array is carried from one stage to the next.
// UNIFORM: the host guarantees that the wave's 8 slots are real channels with the same schedule key (mode, enable flags, tables):
// mode and flags become scalars, every `if (nb_en)` / `if (is_ssb)` a scalar branch instead of an EXEC-mask region.
// WAVES: waves per workgroup.  1 everywhere except the SAM instantiation (4): there the per-sample PLL recurrence -- by far the
// longest dependent chain of the whole path, one lane per channel -- is run for all 32 channels of the workgroup by ONE wave
// (32 lanes busy instead of 8 in each of 4 waves) while the sibling waves wait at a workgroup barrier.
// SAM quadrature PLL, AudioSDR.cpp:688-749: one channel's 128 steps on ONE lane (strictly sequential).  ld(i, xr, xi) / st(i, xr, xi)
// read / write the four IF samples i .. i + 3 of the channel (its LDS rows in the fused kernels, its exchange tile in the
// stand-alone PLL kernel); Sc = the channel's state row.  Returns the lock flag after the block.
// CH = samples per loop trip (4 from LDS; 16 in the PLL kernel, whose accessor requests the next trip's samples from HBM first).
// PF: the accessor reads HBM (the stand-alone PLL kernel): the NEXT trip's samples are requested before this trip's arithmetic, so
// that the chain never waits for memory.
#ifndef ASDR_PLL_CONSTS_IN_VGPRS
#define ASDR_PLL_CONSTS_IN_VGPRS 1   /* the stand-alone PLL kernel keeps the loop's constants in VGPRs (a quarter of its vector instructions read the scalar file otherwise: tools/ubench/issue_rate.hip) */
#endif
template <bool TWO_SUMS, int CH, bool PF = false, typename LD, typename ST>
__device__ __forceinline__ bool pll_loop(ChanSmall *Sc, const ChainConsts &K_, const float *sine, float two_pi, LD ld, ST st) {
    // PF = the stand-alone PLL kernel (51 VGPRs, four homogeneous waves per SIMD: exactly the situation of the issue-rate micro-benchmark): its
    // constants as VGPR operands.  The fused kernels (no register to spare) keep them scalar.
    ChainConsts K = K_;
    if constexpr (PF && ASDR_PLL_CONSTS_IN_VGPRS) {
      asm("" : "+v"(K.pll_a1), "+v"(K.pll_b0), "+v"(K.pll_b1), "+v"(K.half_pi_f), "+v"(two_pi), "+v"(K.pll_alpha_freq), "+v"(K.pll_beta_freq), "+v"(K.pll_f_conv),
               "+v"(K.pll_lock_lo), "+v"(K.pll_lock_hi));
      asm("" : "+v"(K.sin_index_scale_d), "+v"(K.half_pi_d));
    }
    float y_re = Sc->pll_y_re, y_im = Sc->pll_y_im, prev_filt = Sc->pll_prev_filt;
    float d0 = Sc->pll_d0, d1 = Sc->pll_d1, phase_est = Sc->pll_phase_est, pfreq = Sc->pll_freq;
    bool locked = false;
    float nr[CH], ni[CH];
    if (PF) ld(0, nr, ni);
#pragma unroll 1
    for (int i = 0; i < ASDR_N; i += CH) {
      float xr[CH], xi[CH];
      if (PF) {
#pragma unroll
        for (int u = 0; u < CH; ++u) { xr[u] = nr[u]; xi[u] = ni[u]; }
        ld((i + CH < ASDR_N) ? i + CH : i, nr, ni);   // (the last trip re-reads its own samples: no branch)
      } else {
        ld(i, xr, xi);
      }
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        const float x_re = xr[u], x_im = xi[u];
        const float d_re = x_re * y_re + x_im * y_im;
        const float d_im = x_im * y_re - x_re * y_im;
        const float err = approx_atan2<TWO_SUMS>(d_im, d_re, K.half_pi_f);
        d1 = d0;
        d0 = err - K.pll_a1 * d1;
        const float filt = K.pll_b0 * d0 + K.pll_b1 * d1;
        // (float)((double)phase_est + (double)(filt + prev_filt) / 2.0) (:732) as ONE binary32 fma: the halving is exact in
        // binary64 and rounding the binary64 sum of two binary32-precision values to binary32 is innocuous double rounding
        // (53 >= 2 * 24 + 2), so the single rounding of the exact value agrees for every operand pair (oracle
        // ao_check_pll_phase_update).  Four binary64-rate operations less on the loop's dependent chain.
        phase_est = __builtin_fmaf(filt + prev_filt, 0.5f, phase_est);
        prev_filt = filt;
        // The reference's two unbounded wrap loops (:735-736) never end once |phase_est| is so large that
        // phase_est -+ twoPI == phase_est (infinity; or a huge step): there they stall one Teensy instance, here they would
        // hang the wave and with it the batch.  Defined difference: at most ASDR_PLL_WRAP_MAX turns per sample, then the
        // estimate restarts at 0.  A physical loop-filter step is below pi, i.e. one turn; the oracle mirrors the bound.
        // Comparisons against the double PI are comparisons against the float just above it: no float lies between them
        // (pe >= PI_D <=> pe >= pi_up; pe < -PI_D <=> pe <= -pi_up).  The first turn is a select; further turns (never for a
        // finite loop-filter step) run in a loop that a wave enters only if one of its lanes still needs it.
        { const float pi_up = 3.14159274101257324f;   // RN_float(PI) > PI_D
          // (one turn towards zero: phase_est - two_pi for phase_est >= pi_up, phase_est + two_pi == phase_est - (-two_pi) for phase_est <= -pi_up)
          const float turned = phase_est - __builtin_copysignf(two_pi, phase_est);
          phase_est = (fabsf(phase_est) >= pi_up) ? turned : phase_est;
          if (__any(fabsf(phase_est) >= pi_up)) {
            int turns = 1;
            while (phase_est >= pi_up && turns < ASDR_PLL_WRAP_MAX) { phase_est -= two_pi; ++turns; }
            while (phase_est <= -pi_up && turns < ASDR_PLL_WRAP_MAX) { phase_est += two_pi; ++turns; }
            if (turns >= ASDR_PLL_WRAP_MAX) phase_est = 0.0f;
          } }
#ifndef ASDR_PLL_FLAT_LOOKUPS
        sincos_pll(sine, phase_est, y_re, y_im, two_pi, SinIndexK{K.inv_two_pi_d, K.sin_index_scale_d}, K.half_pi_d);   // |phase_est| < pi here (wrap above)
#else
        y_re = cos_f32<true>(sine, phase_est, two_pi, SinIndexK{K.inv_two_pi_d, K.sin_index_scale_d}, K.half_pi_d);   // |phase_est| < pi here (wrap above)
        y_im = sin_f32<true>(sine, phase_est, two_pi, SinIndexK{K.inv_two_pi_d, K.sin_index_scale_d});
#endif
        pfreq = K.pll_alpha_freq * pfreq + K.pll_beta_freq * (filt * K.pll_f_conv);
        locked = (pfreq > K.pll_lock_lo) && (pfreq < K.pll_lock_hi);
        const float o_re = x_re * y_re + x_im * y_im, o_im = -x_re * y_im + x_im * y_re;
        xr[u] = locked ? o_re : x_re;      // rotated sample while locked (:720-723), else the sample stays
        xi[u] = locked ? o_im : x_im;
      }
      st(i, xr, xi);
    }
    Sc->pll_y_re = y_re; Sc->pll_y_im = y_im; Sc->pll_prev_filt = prev_filt;
    Sc->pll_d0 = d0; Sc->pll_d1 = d1; Sc->pll_phase_est = phase_est; Sc->pll_freq = pfreq;
    return locked;
}

// Folded Hilbert FIR of the update kernels (asdr_fir.h): Lrow = the channel's LDS row (history at XP), taps c_hilbert.
template <int E0, int NE, bool TAPS_V = (ASDR_FIR_TAPS_IN_VGPRS != 0)>
__device__ __forceinline__ void hilbert_fir(const float *Lrow, int p0, v2f *acc2) {
  hilbert_fir_rows<E0, NE, TAPS_V>(Lrow + XP, p0, acc2, c_hilbert);
}
// Second wave of a role-2 workgroup of the streaming pipeline: nothing but the other half of the FIR, in step with the first
// wave's three barriers per block (history staged | all reads done | both halves in W1).
// E0, NE: this wave's output pairs (round 6: asdr_stream_kernel_h3 runs THREE helper waves beside the role wave, two pairs each).
template <int STRIDE, int E0 = 4, int NE = 4>
__device__ __forceinline__ void asdr_stream_fir_helper(const UpdateArgs &a, float *lds) {
  const int lane = threadIdx.x & 63, c8 = lane >> 3, s8 = lane & 7, k0 = 16 * s8;
  float *L = lds + c8 * STRIDE;
  // an AM group has no FIR (and its first wave passes no barrier): this wave is done
  const int wave_g = (int)blockIdx.x % a.stream_waves;
  if (((a.direct_ch0 >= 0) ? a.direct_mode : a.sched[wave_g * 8].mode) == ASDR_AMmode) return;
#pragma unroll 1
  for (int blk = 0; blk < a.n_blocks; ++blk) {
    v2f acc2[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) acc2[e] = (v2f){0.0f, 0.0f};
    __syncthreads();
    if (ABL_ON(ABL_HIL)) hilbert_fir<E0, NE>(L, k0 >> 1, acc2);
    __syncthreads();
#pragma unroll
    for (int e = 0; e < NE; e += 2) *reinterpret_cast<float4 *>(L + W1 + k0 + 2 * (E0 + e)) = make_float4(acc2[e][0], acc2[e][1], acc2[e + 1][0], acc2[e + 1][1]);
    __syncthreads();
  }
}

// ROLE: 0 = the whole chain (every launch but the streaming pipeline's).  1 / 2 / 3 = one third of it, for asdr_stream_kernel:
// 1 = input scale + blanker + IF filter, 2 = mixer + Hilbert + sideband, 3 = audio filter + AGC + output (SSB-class modes,
// no ALS).  A role skips the other thirds by seeing their enables as off; the rows that cross a boundary (IF output I/Q; the
// demodulated audio) travel through exchange rings in HBM, ordered by per-wave progress counters (release / acquire, agent scope).
// Hand-off between workgroups without agent-scope fences (which cost 1.7 us per acquire and 1.7-6.5 us per release on this part:
// MI355X_MICROARCH.md, inter-workgroup visibility): EVERY store of the handed-off rows is an `sc1` (write-through) store, the
// storing wave drains them (`s_waitcnt vmcnt(0)`) before ONE lane stores the `sc1` counter; the consumer polls the counter with
// `sc1` loads and then reads EVERY handed-off byte with `sc1` loads to registers (they bypass the CU's vector L1, which another
// CU's stores never refresh).  One wave per workgroup, one workgroup per CU (the host caps the pipeline at 3 x waves <= 255).
// Spin-wait until *p >= target (all lanes read the same word); false = gave up (error flag set, see ASDR_STREAM_SPIN_LIMIT).
// Returns the counter value it saw (>= target; the caller keeps it: a producer that is several blocks ahead is polled once).
// ASDR_STREAM_FAIL = this wave gave up after `limit` polls (it sets the launch's error word), or another wave had (the word is
// looked at every 64 polls): the caller leaves its block loop -- the host has a snapshot of the state and re-runs the call.
__device__ __forceinline__ uint32_t stream_wait(const uint32_t *p, uint32_t target, uint32_t *err, uint32_t limit) {
  uint32_t spins = 0, v;
  while ((v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < target) {
    __builtin_amdgcn_s_sleep(2);
    ++spins;
    if (spins > limit) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); v = ASDR_STREAM_FAIL; break; }
    if ((spins & 63u) == 0u && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { v = ASDR_STREAM_FAIL; break; }
  }
  asm volatile("" ::: "memory");   // no load of the handed-off rows may be moved in front of the poll
  return v;
}
__device__ __forceinline__ void stream_signal(uint32_t *p, uint32_t value, int lane) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's stores (all of a channel group's rows are its own) have left
  if (lane == 0) __hip_atomic_store(p, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// four 16-byte pieces (p + 32 m floats, m = 0..3) of an exchange row, `sc1`; the loads are waited for inside
// 16 bytes another workgroup wrote during this launch, as two relaxed agent-scope loads (`sc1`, like xch_load4x4) that the COMPILER tracks: the value may be
// asked for long before it is used (role 2's prefetch of the next block's rows), which inline assembly cannot promise (a register it defines counts as written at once).
__device__ __forceinline__ v4f ld_sc1_v4(const float *p) {
  const unsigned long long lo = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long hi = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  v4f r;
  r[0] = __uint_as_float((uint32_t)lo); r[1] = __uint_as_float((uint32_t)(lo >> 32)); r[2] = __uint_as_float((uint32_t)hi); r[3] = __uint_as_float((uint32_t)(hi >> 32));
  return r;
}
__device__ __forceinline__ void xch_store4x4(float *p, v4f v0, v4f v1, v4f v2, v4f v3) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx4 %0, %2, off offset:128 sc1\n\t"
               "global_store_dwordx4 %0, %3, off offset:256 sc1\n\tglobal_store_dwordx4 %0, %4, off offset:384 sc1\n\t"
               "s_nop 1"   // (a store wider than 8 bytes reads its data a wait state late: nothing may overwrite v3 right behind it)
               :: "v"(p), "v"(v0), "v"(v1), "v"(v2), "v"(v3) : "memory");
}
__device__ __forceinline__ void xch_load4x4(const float *p, v4f &v0, v4f &v1, v4f &v2, v4f &v3) {
  asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %4, off offset:128 sc1\n\t"
               "global_load_dwordx4 %2, %4, off offset:256 sc1\n\tglobal_load_dwordx4 %3, %4, off offset:384 sc1\n\t"
               "s_waitcnt vmcnt(0)"
               : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(p) : "memory");
}

// The ALS filter of one block (AudioSDR.cpp:329-351) on a channel's LDS rows: L[XB + idx] = sample idx of the reference's 256-sample
// buffer (idx >= 128 - history kept), taps de-interleaved at L[AW ..] (ALS_TAP, WH per half), error broadcast word L[SCR], output sample
// n at L[OUT + n].  OUT may overlay the history at buffer index i - 65 (the stand-alone ALS kernel): a sample's sums read no index
// below i - 64, the four samples of a tap set are summed before any of them is written, and the tap update behind them reads from
// i + 3 - 64 upwards.  Called by all lanes of the wave (wave-uniform branches inside).
template <bool COMPACT, int XB, int AW, int WH, int SCR, int OUT>
__device__ __forceinline__ void als_compute(float *L, bool als_en, bool adaptive, bool notch, int M, int D, float lam, int s8, int k0) {
    constexpr int AH = 2 * WH;
    // every history index i - D - q (i = 128..255, q < M) is inside the kept window iff D >= 0 and D + M <= AH + 1: true for
    // the reference's defaults (M 55, D 3) and checked per wave.  (The host sends a channel to a compact instantiation only
    // if it is: the checked forms are compiled for the 516-float rows alone.)
    const bool als_safe = COMPACT || __all(!als_en || (D >= 0 && D + M <= AH + 1));
    const int M_u = __builtin_amdgcn_readfirstlane(M);
    const bool als_m_uniform = __all(als_en && adaptive && M == M_u);   // the usual case: one filter length in the wave
    const bool als_m_default = als_m_uniform && als_safe && M_u == ALS_M_DEFAULT;   // ... and the reference's default one
    if (__any(als_en && !adaptive)) {   // static taps: every lane computes its 16 outputs, all results written after all reads
      float yo[16];                     // (the output row may overlay the history: OUT != a row of its own in the stand-alone kernel)
#pragma unroll
      for (int j = 0; j < 16; ++j) yo[j] = 0.0f;
      if (als_en && !adaptive) {
        for (int j = 0; j < 16; ++j) {
          const int i = 128 + k0 + j;
          float y;
          if constexpr (COMPACT) y = als_dot<false, WH>(L + AW, L + XB, i - D, M);
          else y = als_safe ? als_dot<false, WH>(L + AW, L + XB, i - D, M) : als_dot<true, WH>(L + AW, L + XB, i - D, M);
          const float e = L[XB + i] - y;
          yo[j] = notch ? e : y;
        }
      }
      WAVE_SYNC();
      if (als_en && !adaptive) {
#pragma unroll
        for (int j = 0; j < 16; ++j) L[OUT + k0 + j] = yo[j];
      }
    }
#ifndef ASDR_ALS_TAPS_IN_LDS
    if (als_m_default) {
      // The reference's default length (55 taps, adaptive) in the whole wave: THE TAPS LIVE IN REGISTERS for the block, each held
      // once.  A channel's eight lanes are two quads: the low one works on the even taps, the high one on the odd taps, and lane
      // t of a quad owns tap t of every group of four (tap q = 8 g + 2 t + h: seven registers).  The products of a sum take their
      // tap from the owner through a DPP quad broadcast folded into the multiply (v_mul_f32_dpp quad_perm:[t,t,t,t]) -- no LDS
      // read, no extra instruction -- and the tap update w[q] += lambda * (e * x[iu - D - q]) (AudioSDR.cpp:341-343) is done by the
      // owner alone: 7 x 3 operations per lane and tap set instead of 28 x 3.  Same products, same sums, same order.  Per tap
      // set: 18 two-sample LDS reads and ~135 VALU instructions (round 2: 7 b128 + 56 accesses and ~105 + 28 wait states; the
      // filter was bound by LDS bandwidth -- asdr_als_kernel 0.237 ms for 131,072 channels with the LDS pipe ~65 % busy).
      const int h = s8 >> 2, tq = s8 & 3;
      float wr[7];
#pragma unroll
      for (int g = 0; g < 7; ++g) wr[g] = L[AW + WH * h + 4 * g + tq];
      const bool own_valid6 = !(tq == 3 && h != 0);   // tap 55 does not exist
      // One tap set (four samples, one per lane of a quad; the first tap set is sample 0 alone).  J >= 0: the tap set is number J of a
      // group of four whose operands were read TOGETHER in front of the group: a lane's 28 sum operands are x[i - D - h - 2 m],
      // m = 0..27, and i advances by 4 from tap set to tap set, so 26 of the 28 are the previous tap set's, two places on -- E[k] =
      // x[i0 - D - h - 2 k], k = -6..27, serves the four tap sets (tap set J uses E[m - 2 J]); likewise the operands of the lane's own
      // taps for the update, F[k] = x[iu0 - D - h - 2 tq - 4 k], k = -3..12 (tap set J, tap group g: F[2 g - J]).  34 + 16 + 4 floats per
      // lane and group instead of 4 x 36: the filter's LDS reads drop from 19 two-sample reads per tap set to 7.
      auto epoch = [&](auto first_tag, auto j_tag, int ep, const float *E, const float *F, const float *XI) {
        constexpr bool FIRST = decltype(first_tag)::value;
        constexpr int J = decltype(j_tag)::value;
        const int base = FIRST ? 0 : 4 * ep + 1;
        const int cntn = FIRST ? 1 : ((ep == 31) ? 3 : 4);
        const int n = base + tq, i = 128 + n;
        const bool mine = (s8 < cntn);
        const int nu = FIRST ? 0 : 4 * ep + 4;   // the updating sample of this tap set
        float xs[7][4], xo[7], xi;
        if constexpr (J < 0) {
          const float *xh = L + XB + (i - D - h);
#pragma unroll
          for (int g = 0; g < 7; ++g) { xs[g][0] = xh[-8 * g]; xs[g][1] = xh[-8 * g - 2]; xs[g][2] = xh[-8 * g - 4]; xs[g][3] = xh[-8 * g - 6]; }
          // the operands of this lane's own taps for the update below: x[iu - D - q], q = 8 g + 2 tq + h (requested with the sums' operands)
          const float *xu = L + XB + (128 + nu - D - h - 2 * tq);
#pragma unroll
          for (int g = 0; g < 7; ++g) xo[g] = xu[-8 * g];
          xi = L[XB + i];   // the sample itself (e = x[i] - y), requested with the operands
        } else {
#pragma unroll
          for (int g = 0; g < 7; ++g) {
#pragma unroll
            for (int t = 0; t < 4; ++t) xs[g][t] = E[6 + 4 * g + t - 2 * J];
            xo[g] = F[3 + 2 * g - J];
          }
          xi = XI[J];
        }
        float y = 0.0f;
#pragma unroll
        for (int g = 0; g < 7; ++g) {
          float p[4];
          p[0] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(wr[g]), 0x00, 0xF, 0xF, true)) * xs[g][0];   // quad_perm:[0,0,0,0]
          p[1] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(wr[g]), 0x55, 0xF, 0xF, true)) * xs[g][1];   // [1,1,1,1]
          p[2] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(wr[g]), 0xAA, 0xF, 0xF, true)) * xs[g][2];   // [2,2,2,2]
          p[3] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(wr[g]), 0xFF, 0xF, 0xF, true)) * xs[g][3];   // [3,3,3,3]
          if (g == 6) p[3] = (h == 0) ? p[3] : 0.0f;   // tap 55 does not exist (the running sum starts at +0.0: adding +0.0 leaves it)
#pragma unroll
          for (int t = 0; t < 4; ++t) { y += p[t]; y += dpp_row_shl4(p[t]); }
#ifdef ASDR_ALS_GROUP_BARRIER   /* (round 3, first form: kept the scheduler from forming all 28 products up front -- and kept 20 of the 28 DPP operands from folding) */
          __builtin_amdgcn_sched_barrier(0);
#endif
        }
#ifndef ASDR_ALS_SUMS_SINKABLE
        // The sum is formed by ALL lanes, here: left to itself the compiler sinks the 56 additions into the `mine` branch below (only
        // there is y used), where the partner quad is inactive -- so the 28 `row_shl:4` operands cannot fold into the additions and
        // become 28 v_mov_b32_dpp per tap set in front of the branch.  With the sums here and no scheduling barrier between the tap
        // groups every one of them is a v_add_f32_dpp, the products rotate through four registers (mul | add | add_dpp per tap pair)
        // and a tap set is 144 instructions instead of 151 (21 of them one-cycle s_nop for the DPP read-after-write hazard): C4 -3.8 %.
        asm volatile("" : "+v"(y));
#endif
        // The error of the updating sample (n = nu: lane tq = 3 of the low quad, tq = 0 in the first tap set) goes to the channel's
        // eight lanes through two DPP moves -- its quad, then the partner quad (row_shr:4 into banks 1 and 3 only) -- instead of an
        // LDS word between two wave syncs: the tap set's dependent path error -> tap update -> next products stays in registers.
        // (Nothing else crosses lanes through LDS inside the loop: the history is read-only and the output samples are read after it.)
        float e_own = 0.0f;
        if (mine) {
          e_own = xi - y;
          L[OUT + n] = notch ? e_own : y;
        }
        if (nu < ASDR_N) {
          const int eq = __builtin_amdgcn_update_dpp(0, __float_as_int(e_own), FIRST ? 0x00 : 0xFF, 0xF, 0xF, true);   // quad_perm [0,0,0,0] / [3,3,3,3]
          const float e = __int_as_float(__builtin_amdgcn_update_dpp(eq, eq, 0x114, 0xF, 0xA, false));                // row_shr:4, banks 1 and 3
#pragma unroll
          for (int g = 0; g < 7; ++g) {
            const float gq = e * xo[g]; const float dq = lam * gq; const float wn = wr[g] + dq;
            wr[g] = (g == 6 && !own_valid6) ? wr[g] : wn;
          }
        }
      };
      epoch(std::true_type{}, std::integral_constant<int, -1>{}, -1, nullptr, nullptr, nullptr);
#if ASDR_ALS_WINDOW
      // OUT may overlay the history (the stand-alone kernel: output n at buffer index 128 + n - 65): a group's reads, made in front of its
      // first tap set, reach back to index i0 - D - h - 54 >= i0 - 64 - ... -- never below what the PREVIOUS groups' outputs (n <= base0 - 1,
      // index <= i0 - 66) have overwritten -- and forward to samples of the current block, which nothing writes.
#pragma unroll 1
      for (int grp = 0; grp < 8; ++grp) {
        const int ep0 = 4 * grp, i0 = 128 + 4 * ep0 + 1 + tq;
        const float *xh0 = L + XB + (i0 - D - h);
        const float *xu0 = L + XB + (128 + 4 * ep0 + 4 - D - h - 2 * tq);
        float E[34], F[16], XI[4];
#pragma unroll
        for (int k = 0; k < 34; ++k) E[k] = xh0[-2 * (k - 6)];
#pragma unroll
        for (int k = 0; k < 16; ++k) F[k] = xu0[-4 * (k - 3)];
#pragma unroll
        for (int j = 0; j < 4; ++j) XI[j] = L[XB + i0 + 4 * j];
        epoch(std::false_type{}, std::integral_constant<int, 0>{}, ep0, E, F, XI);
        epoch(std::false_type{}, std::integral_constant<int, 1>{}, ep0 + 1, E, F, XI);
        epoch(std::false_type{}, std::integral_constant<int, 2>{}, ep0 + 2, E, F, XI);
        epoch(std::false_type{}, std::integral_constant<int, 3>{}, ep0 + 3, E, F, XI);
      }
#else
#pragma unroll 1
      for (int ep = 0; ep < 32; ++ep) epoch(std::false_type{}, std::integral_constant<int, -1>{}, ep, nullptr, nullptr, nullptr);
#endif
      // the taps back to their LDS rows
#pragma unroll
      for (int g = 0; g < 7; ++g) L[AW + WH * h + 4 * g + tq] = wr[g];
      WAVE_SYNC();
    } else
#endif
    if (__any(als_en && adaptive)) {
      // taps change only after samples n = 0, 4, 8, ...; samples sharing one tap set run on lanes s8 = 0..3
#pragma unroll 1
      for (int ep = -1; ep < 32; ++ep) {
        const int base = (ep < 0) ? 0 : 4 * ep + 1;
        const int cntn = (ep < 0) ? 1 : ((ep == 31) ? 3 : 4);
        const int n = base + (s8 & 3);
        const bool mine = als_en && adaptive && (s8 < cntn);
        if (als_safe) {   // all eight lanes of the channel share the sum (als_dot_split); lanes 0-3 keep the results
          const int i = 128 + n, h = s8 >> 2;
          float y = 0.0f;
          if (als_m_default) y = als_dot_split_const<ALS_M_DEFAULT>(L + AW + WH * h, L + XB + (i - D - h), h);
          else if (als_m_uniform) y = als_dot_split_uniform(L + AW + WH * h, L + XB + (i - D - h), M_u, h);   // scalar loop count
          else if (als_en && adaptive) y = als_dot_split(L + AW + WH * h, L + XB + (i - D - h), M, h);
          if (mine) {
            const float e = L[XB + i] - y;
            L[OUT + n] = notch ? e : y;
            if ((n & 3) == 0) L[SCR] = e;
          }
        } else if (mine) {
          if constexpr (!COMPACT) {
            const int i = 128 + n;
            const float y = als_dot<true, WH>(L + AW, L + XB, i - D, M);
            const float e = L[XB + i] - y;
            L[OUT + n] = notch ? e : y;
            if ((n & 3) == 0) L[SCR] = e;
          }
        }
        WAVE_SYNC();
        const int nu = (ep < 0) ? 0 : 4 * ep + 4;   // the updating sample of this epoch
        if (als_en && adaptive && nu < ASDR_N) {
          const float e = L[SCR];
          const int iu = 128 + nu;
          if (als_m_default) {
            als_tap_update_const<ALS_M_DEFAULT>(L + XB + iu - D - s8, L + AW + ALS_TAP(s8, WH), e, lam, s8);
          } else if (als_m_uniform && als_safe) {
            // lane s8 owns taps s8 + 8k (k < nk).  Eight of them per step, all operands requested together and all results
            // written together: one LDS round trip per step instead of one per tap.  Steps run to a multiple of 8 taps per
            // lane: the surplus ones (still inside the tap rows: M <= 64 -> k <= 7, M <= 128 -> k <= 15) are written back unchanged.
            const float *xq = L + XB + iu - D - s8;
            float *wq = L + AW + ALS_TAP(s8, WH);
            const int nk = (M_u - s8 + 7) >> 3, nk_max = (M_u + 7) >> 3;
#pragma unroll 1
            for (int k0 = 0; k0 < nk_max; k0 += 8) {
              float xv[8], wv[8];
#pragma unroll
              for (int j = 0; j < 8; ++j) { xv[j] = xq[-8 * (k0 + j)]; wv[j] = wq[4 * (k0 + j)]; }
#pragma unroll
              for (int j = 0; j < 8; ++j) { const float gq = e * xv[j]; const float dq = lam * gq; const float wn = wv[j] + dq; wv[j] = (k0 + j < nk) ? wn : wv[j]; }
#pragma unroll
              for (int j = 0; j < 8; ++j) wq[4 * (k0 + j)] = wv[j];
            }
          } else if (als_safe) {
#pragma unroll 2
            for (int q = s8; q < M; q += 8) { const float gq = e * L[XB + iu - D - q]; const float dq = lam * gq; L[AW + ALS_TAP(q, WH)] += dq; }
          } else {
            if constexpr (!COMPACT) {
              for (int q = s8; q < M; q += 8) { const float gq = e * (((iu - D - q) >= 0 && (iu - D - q) < 256) ? L[XB + (iu - D - q)] : 0.0f); const float dq = lam * gq; L[AW + ALS_TAP(q, WH)] += dq; }
            }
          }
        }
        WAVE_SYNC();
      }
    }
}
// ONEBLK (round 5): the launch processes ONE block per channel (a.n_blocks == 1: every large batch -- the host issues a multi-block
// call of a large schedule block by block -- and always the SAM / two-launch ALS roles).  The block loop below is then no loop at all:
// nothing is loop-invariant, so nothing is hoisted in front of the body and kept in registers across all of it (lane masks, cache-entry
// addresses, flag words: the plain kernel's 66 SGPR spills and 8 of its VGPRs were exactly that).
template <int STRIDE, bool HAS_ALS, bool HAS_SAM, bool UNIFORM, int WAVES, int ROLE = 0, bool ONEBLK_ = false, int FIR_HELPERS = 1>
__device__ __forceinline__ void asdr_update_body(const UpdateArgs &a, float *lds_wg) {
  // ROLE 4 / 5: the SAM sub-range as three launches -- 4 = everything in front of the PLL (scale, blanker, IF filter), then the
  // stand-alone PLL kernel (asdr_sam_pll_kernel: one LANE per channel, 64 channels per wave -- the PLL is a 128-step dependent chain
  // per channel, and as one phase of a fused kernel it kept a whole workgroup waiting for ~58 k cycles per block), 5 = everything
  // behind it.  The IF rows cross in xch_sam; the lock flag in the status word.
  // ROLE 6: the whole chain up to and including the AGC for channels whose (short) ALS filter runs as a launch of its own
  // (asdr_als_kernel): the post-AGC row goes to the ALS input ring instead of through the filter and the output stage.
  constexpr bool C16 = (STRIDE == 320);   // the 16-waves-per-CU form (see below, where the block loop starts)
  // ROLE 7: role 6 with the block loop kept (ALS role streams: a chunk of blocks per launch).  ROLE 8 / 9: roles 4 / 5 with the block loop
  // kept (SAM role streams in chunks): block k of the launch works on tile set (a.sam_set + k) % a.sam_sets.
  // ROLE 10: role 9 whose post-AGC row goes to the ALS stage instead of through the output stage (the three-stage ALS role streams).
  constexpr bool PRE_ROLE = (ROLE == 4 || ROLE == 8), POST_ROLE = (ROLE == 5 || ROLE == 9 || ROLE == 10), LOOPED_ROLE = (ROLE == 7 || ROLE == 8 || ROLE == 9 || ROLE == 10);
  constexpr bool DO1 = (ROLE == 0 || ROLE == 1 || PRE_ROLE || ROLE == 6 || ROLE == 7), DO2 = (ROLE == 0 || ROLE == 2 || POST_ROLE || ROLE == 6 || ROLE == 7),
                 DO3 = (ROLE == 0 || ROLE == 3 || POST_ROLE || ROLE == 6 || ROLE == 7);
  constexpr bool TO_ALS = (ROLE == 6 || ROLE == 7 || ROLE == 10);
#ifndef ASDR_ALS_FULL_OPT
#define ASDR_ALS_FULL_OPT 2
#endif
  // The ALS instantiations gave up two of the plain kernel's orderings for registers (IF rows consumed before the ring prefetches; the
  // merged average + phase loop).  Their LOOP-FREE compact-row forms have the registers for both (round 5: asdr_update_kernel_als_small_one
  // 160 VGPRs, no spills; C4 share -1.3 %, on the lanes -2.5 %); the looped forms keep the old orderings.  Level 0 / 1: measurements.
  constexpr bool ALS_LOOPFREE = (ASDR_ONEBLK_ALS != 0) && (ONEBLK_ || (ROLE >= 4 && !LOOPED_ROLE && ASDR_ONEBLK_ROLES));
  constexpr bool ALS_FULL_OPT = (ASDR_ALS_FULL_OPT >= 1) && ASDR_COMPACT_ROWS(STRIDE) && ALS_LOOPFREE,
                 ALS_FULL_OPT2 = (ASDR_ALS_FULL_OPT >= 2) && ASDR_COMPACT_ROWS(STRIDE) && ALS_LOOPFREE;
  constexpr bool STREAM = (ROLE >= 1 && ROLE <= 3);
  // biquad pipelines with packed products (biquad_pipe<true>), by instantiation: bit 0 = the ALS kinds, 1 = the plain kinds, 2 = the SAM roles, 3 = the block pipeline
  constexpr bool PIPE_PK = ((ASDR_PIPE_PK_MASK & 1) && HAS_ALS) || ((ASDR_PIPE_PK_MASK & 2) && !HAS_ALS && !HAS_SAM && ROLE == 0) ||
                           ((ASDR_PIPE_PK_MASK & 4) && (HAS_SAM || ROLE > 3)) || ((ASDR_PIPE_PK_MASK & 8) && STREAM);
  if (ROLE == 0 && a.run_if != nullptr && *a.run_if == 0u) return;   // the pipeline's fallback launch: nothing to do unless the pipeline gave up
  // MW (round 5): the plain chain in workgroups of FOUR waves = 32 channels that stop wasting lanes on the phases that use a fraction of
  // a wave: the audio cascade (4 lanes per channel: 32 of 64 busy in a wave of its own) runs for 16 channels per wave on TWO of the
  // four waves, the blanker's running average + mixer phase recurrences and the AGC's envelope recurrence (one lane per channel:
  // 8 of 64) run for all 32 channels on ONE wave each, between two workgroup barriers -- what the multi-wave SAM instantiation does
  // for the PLL.  The duties rotate with the workgroup index (rel: 0 = blanker / phase chains, 1 and 2 = audio cascades, 3 = AGC
  // chain), so that every SIMD gets every duty.  Direct launches of one settings group only (mode, flags and tables are launch-uniform
  // scalars: every barrier below is reached by all four waves or by none); padding waves of the last workgroup run the dummy channel.
  constexpr bool MW = (WAVES > 1) && !HAS_SAM && (ROLE == 0);
  constexpr int MW_SHARE = MW ? ASDR_MW_SHARE : 0;   // bit 0: audio cascades, bit 1: blanker / phase chains, bit 2: AGC chain (measurements: any subset)
  const int wave = (WAVES > 1) ? (int)(threadIdx.x >> 6) : 0;
#ifndef ASDR_MW_ROT
#define ASDR_MW_ROT 0
#endif
  const uint32_t mw_b = blockIdx.x;
  const uint32_t mw_rot = (ASDR_MW_ROT == 0) ? mw_b : (ASDR_MW_ROT == 1) ? (mw_b >> 3) : (ASDR_MW_ROT == 2) ? ((mw_b >> 3) + (mw_b >> 8)) :
                          (ASDR_MW_ROT == 3) ? ((mw_b * 0x9E3779B1u) >> 30) : (ASDR_MW_ROT == 4) ? ((mw_b >> 3) % 3u) : (mw_b >> 5);
  const int mw_rel = MW ? (int)(((uint32_t)wave - mw_rot) & (uint32_t)(WAVES - 1)) : 0;
  float *const mwx = lds_wg + WAVES * 8 * STRIDE;   // MW: [8 * WAVES][MWX] floats of hand-off scratch behind the rows
#ifndef ASDR_MW_STRAGGLER_PRIO
#define ASDR_MW_STRAGGLER_PRIO 0   /* experiment (round 6): at three points between the chain duty and the audio duty a wave that finds itself the LAST of its workgroup raises its priority */
#endif
  // (progress words of the four waves behind the hand-off scratch and the sine table)
  int *const mw_prog = reinterpret_cast<int *>(lds_wg + WAVES * 8 * STRIDE + 8 * WAVES * 16 + 260);
  int *const mw_flags = mw_prog + 4;   // [0]: the AGC duty took the lean chain (this block); [1]: chunks whose |x| and beta |x| are in the rows, [2], [3]: chunks the audio duty waves have finished (the AGC duty beside the audio duty)
  constexpr bool PIPE_AGC = MW && WAVES == 4 && (MW_SHARE & 7) == 7 && (ASDR_MW_AGC_PIPELINED != 0) && (ASDR_AGC_LEAN != 0) && (ASDR_MW_OWN_STORES != 0) && ABL_ON(ABL_AGC) && ABL_ON(ABL_AF) && ABL_ON(ABL_NB);
  if constexpr (PIPE_AGC) { if (threadIdx.x < 3) mw_flags[1 + threadIdx.x] = 0; }   // (five barriers in front of the first reader)
  if constexpr (MW && ASDR_MW_STRAGGLER_PRIO != 0) { if ((threadIdx.x & 63) == 0) mw_prog[wave] = 0; }
  auto mw_straggler = [&](int point) {
    if constexpr (MW && ASDR_MW_STRAGGLER_PRIO != 0) {
      if ((threadIdx.x & 63) == 0) mw_prog[wave] = point;
      const int4 pr = *reinterpret_cast<const int4 *>(mw_prog);
      const int behind = (pr.x < point) + (pr.y < point) + (pr.z < point) + (pr.w < point);   // siblings that have not reached this point yet
      const bool last = __builtin_amdgcn_readfirstlane(behind) == 0;
      if (last) __builtin_amdgcn_s_setprio(ASDR_MW_STRAGGLER_PRIO); else __builtin_amdgcn_s_setprio(0);
    }
  };
  constexpr int MWX = 16;   // per channel: 0 blanker average in / out, 1 mixer phase in / "a phase sequence was computed" out, 2 increment, 3 flags, 4 phase after the block,
                            // 5 AGC quiet flag, 6 AM level, 7 gain after the block, 8 envelope in / out, 9 hang counter in / out, 10 gain in, 11..14 attack / release alpha, beta, 15 hang count
  // schedule slot / channel index of the workgroup's channel q (0 .. 8 * WAVES - 1), for the lanes that work on other waves' channels
  auto mw_channel = [&](int q) -> int { const int sidx = (int)blockIdx.x * (8 * WAVES) + q; return (sidx < a.n_sched) ? a.direct_ch0 + sidx : a.n_channels; };
  // this wave's index in the launched schedule sub-range (the streaming pipeline launches its three roles one after the other)
  const int wave_g = STREAM ? (int)blockIdx.x % a.stream_waves : (int)blockIdx.x * WAVES + wave;
  float *const lds = lds_wg + wave * 8 * STRIDE;            // this wave's 8 channel rows
  const int lane = threadIdx.x & 63, c8 = lane >> 3, s8_ = lane & 7;
#ifndef ASDR_MW_SINE_LDS
#define ASDR_MW_SINE_LDS 0   /* 1: filled at the top of every workgroup (measured: divergent mixer phases -1.3 %, steady state +0.3 %: the fill and its barrier);
                                  2: filled between the blanker chain's two barriers, by the three waves that park there, and only in a workgroup one of whose waves
                                  carries different mixer phases (each wave says so in a word of its own in front of the first barrier; the wave-uniform lookups keep
                                  the constant table): divergent phases -0.4 %, steady state +0.3 %.  Off */
#endif
  // Round 6: the four-wave form keeps a copy of the sine table in LDS as well (1 KB per workgroup of 32 channels: 3 workgroups per CU still fit) --
  // waves whose channels carry DIFFERENT mixer phases (receivers tuned at different times: 2 x 16 lookups per lane) read it through the LDS
  // instead of gathering from the constant table through L1; one barrier at the top of the kernel.
  constexpr bool MW_SINE = MW && (ASDR_MW_SINE_LDS != 0);
  constexpr bool MW_SINE_LAZY = MW_SINE && (ASDR_MW_SINE_LDS == 2) && (MW_SHARE & 2) != 0 && WAVES == 4;   // (the table exists where a wave needs it: the per-channel lookups; the wave-uniform ones read the constant table)
  constexpr bool SINE_LDS = HAS_SAM || (MW_SINE && !MW_SINE_LAZY);   // the wave-uniform lookups (two per lane) and the oscillator cache's writer
  constexpr bool SINE_LDS_PC = HAS_SAM || MW_SINE;                   // the per-channel lookups (16 per lane)
  float *const sine_tab = HAS_SAM ? lds_wg + WAVES * 8 * STRIDE : (MW_SINE ? lds_wg + WAVES * 8 * STRIDE + 8 * WAVES * 16 : nullptr);
  float *const sine = SINE_LDS ? sine_tab : nullptr;
  int *const mw_sflag = reinterpret_cast<int *>(lds_wg + WAVES * 8 * STRIDE + 8 * WAVES * 16 + 260) + 8;   // MW_SINE_LAZY: per wave "my channels carry different mixer phases"
  if (HAS_SAM) for (int i = lane; i < ASDR_SINE_TABLE_LEN; i += 64) sine_tab[i] = c_sine[i];
  if (MW_SINE && !MW_SINE_LAZY) { for (int i = (int)threadIdx.x; i < ASDR_SINE_TABLE_LEN; i += 64 * WAVES) sine_tab[i] = c_sine[i]; __syncthreads(); }

  int4 slot = make_int4(a.n_channels, 0, 0, 0);
  const bool mw_pad = MW && (wave_g * 8 >= a.n_sched);   // a wave behind the sub-range's last one: works on the dummy channel, stores nothing outside it
  if (UNIFORM && a.direct_ch0 >= 0) slot = make_int4(mw_pad ? a.n_channels : a.direct_ch0 + wave_g * 8 + c8, (int)a.direct_mode, (int)a.direct_flags, (int)((a.direct_lo & 0xFFu) | (wave_g == 0 ? a.lo_writer_bit : 0u)));
  else if (WAVES == 1 || wave_g * 8 < a.n_sched) slot = *reinterpret_cast<const int4 *>(a.sched + wave_g * 8 + c8);   // {channel, mode, flags, -}
  else { const ChanParams *pd = a.params + a.n_channels; slot.y = (int)pd->mode; slot.z = (int)pd->flags; }   // padding wave of a multi-wave workgroup: dummy channel
  const int ch_ = slot.x;
  const bool valid = (UNIFORM && !mw_pad) || (ch_ < a.n_channels);
  const int loff_ = c8 * STRIDE;
#define P (*Pp)
  const ChainConsts K = a.k;

  const uint32_t mode = UNIFORM ? (uint32_t)__builtin_amdgcn_readfirstlane(slot.y) : (uint32_t)slot.y;
  const uint32_t pflags = UNIFORM ? (uint32_t)__builtin_amdgcn_readfirstlane(slot.z) : (uint32_t)slot.z;
  const uint32_t lo_slot = UNIFORM ? (uint32_t)__builtin_amdgcn_readfirstlane(slot.w) : 0u;   // wave-uniform: an SGPR from here on
  const bool is_ssb = DO2 && ((mode == ASDR_USBmode) || (mode == ASDR_LSBmode) || (mode == ASDR_CW_USBmode) ||
                              (mode == ASDR_CW_LSBmode) || (mode == ASDR_WSPRmode));
  // the host launches SAM channels with the SAM (or ALS) instantiation only: the plain one carries no PLL code
#ifndef ASDR_MW_NO_AM
#define ASDR_MW_NO_AM 0   /* experiment: the four-wave form compiled without the AM paths (code size) */
#endif
  const bool is_am = !C16 && !(MW && ASDR_MW_NO_AM) && (mode == ASDR_AMmode), is_sam = (HAS_SAM || ROLE >= 4) && (mode == ASDR_SAMmode);
  const bool sub_q = (mode == ASDR_USBmode) || (mode == ASDR_CW_USBmode) || (mode == ASDR_WSPRmode);
  const bool nb_en = DO1 && (pflags & ASDR_F_NB_EN), af_en = DO3 && (pflags & ASDR_F_AF_EN), agc_en = DO3 && (pflags & ASDR_F_AGC_EN);
  if constexpr (MW_SINE_LAZY) {
    if (!(ABL_ON(ABL_NB) && __any(nb_en))) {   // no blanker in this launch (launch-uniform in the four-wave form), hence no chain barriers: the table is filled here
      for (int i = (int)threadIdx.x; i < ASDR_SINE_TABLE_LEN; i += 64 * WAVES) sine_tab[i] = c_sine[i];
      __syncthreads();
    }
  }
  const bool als_en = HAS_ALS && (pflags & ASDR_F_ALS_EN);
  const bool muted = pflags & ASDR_F_MUTED;
  const float two_pi = K.two_pi_f;
  WAVE_SYNC();
  // streaming pipeline: the neighbours' progress as last seen, and this role's own progress not yet published (roles 2 and 3
  // publish block b only after their loads of block b + 1 have returned: that wait drains the stores of block b for free)
  uint32_t seen_in = 0u, seen_free = 0u, seen_lo = 0u, sig_pending = 0u;
  // ... and the state role 2 carries from block to block in registers instead of reading back what it has just stored (the
  // stores still happen: HBM holds the state after the call): mixer phase, frequency shift, Hilbert ring parity
  float carry_phase = 0.0f, carry_fsh = 0.0f;
  uint32_t carry_hs = 0u;
#ifndef ASDR_STREAM_PREFETCH_IN
#define ASDR_STREAM_PREFETCH_IN 1   /* the block pipeline's role 1 requests the NEXT block's input rows at the top of a block (its one HBM round trip per block otherwise sits in front of the scale) */
#endif
  int4 pf_in[4] = {make_int4(0, 0, 0, 0), make_int4(0, 0, 0, 0), make_int4(0, 0, 0, 0), make_int4(0, 0, 0, 0)};   // ROLE 1: the next block's input pieces (I, I + 64, Q, Q + 64)
#ifndef ASDR_STREAM_R2_ONE_TRIP
#define ASDR_STREAM_R2_ONE_TRIP 1   /* role 2 asks for everything another workgroup wrote for this block -- oscillator key, its 2 x 128 pairs, the IF rows -- in ONE round trip */
#endif
#ifndef ASDR_STREAM_R2_PREFETCH
#define ASDR_STREAM_R2_PREFETCH 1   /* role 2 of the three-helper form asks for block k + 1's oscillator pairs and IF rows in front of block k's FIR, when both producers are known to be that far */
#endif
  constexpr bool R2_ONE_TRIP = (ROLE == 2) && (FIR_HELPERS == 3) && (ASDR_STREAM_R2_ONE_TRIP != 0);   // (68 registers in flight: the three-helper form has them -- one workgroup per compute unit --, the two-wave form spills with it)
  constexpr bool R2_PREFETCH = R2_ONE_TRIP && (ASDR_STREAM_R2_PREFETCH != 0);
  v4f r2n_key = (v4f){0.f, 0.f, 0.f, 0.f}, r2n_c4[4], r2n_s4[4], r2n_vi[4], r2n_vq[4];
  bool r2_pref = false;
  uint32_t r2_look_in = 0u, r2_look_lo = 0u;
#ifndef ASDR_STREAM_R1_CARRY
#define ASDR_STREAM_R1_CARRY 0   /* role 1 keeping its IF cascade's state and coefficients (and the input gains) in registers from block to block: measured +1.5 % (C5 share 4.025 vs 3.96 ms), off */
#endif
  float4 r1_if_s4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float r1_if_cf[5] = {0.f, 0.f, 0.f, 0.f, 0.f}, r1_gain_i = 0.f, r1_gain_q = 0.f;
  uint32_t *const my_prog = STREAM ? a.stream_prog + (ROLE - 1) * a.stream_waves + wave_g : nullptr;

  // (The ALS instantiations' loop-free builds failed the ALS parity tests until the guard of the tap store-back was taken from an opaque copy
  // of the flag word: a compiler issue, see there.)
  constexpr bool ONEBLK = (!HAS_ALS || ASDR_ONEBLK_ALS) && (ONEBLK_ || (ROLE >= 4 && !LOOPED_ROLE && ASDR_ONEBLK_ROLES));
  constexpr bool UNIT_OK = (ASDR_UNIT_SCALE != 0) && ((ONEBLK && ROLE == 0) || ROLE == 1);   // (round 6: the block pipeline's role 1 too -- its scale was 1.3 k cycles of binary64 per block on a wave that is alone on its SIMD)
  // C16 (round 5, asdr_update_kernel_c16): the same chain on 320-float rows and <= 128 VGPRs -- 10,240 B of LDS per wave, FOUR waves per
  // SIMD = 16 per CU.  Direct one-block launches of ONE SSB-class settings group without stage taps (the launcher's choice: mode and
  // flags are launch-uniform scalars, the AM / SAM / unknown-mode paths are compiled out).  What fits 320 floats and 128 registers:
  //   * the mixer's phase row holds the EVEN samples' phases only (64 floats at PH): an odd one is one recurrence step from its neighbour
  //   * the Hilbert FIR runs as TWO passes of 64 outputs over a 318-float window of the history, shifted down by 64 in between
  //   * the AGC's gain table is read through L1 from its HBM row (no LDS copy)
  //   * nothing is prefetched across a phase that does not need it (Hilbert ring, oscillator pairs, IF / audio / AGC state), and the blanker
  //     keeps the oldest block RAW across its sequential pass (scaled again for the output)
  static_assert(!C16 || (ONEBLK && UNIFORM && !HAS_ALS && !HAS_SAM && WAVES == 1 && ROLE == 0), "the 16-waves-per-CU form is the plain uniform one-block kernel");   // the unit-gain scale (scale8): the loop-free instantiations have the registers for its second code path
#pragma unroll 1
  for (int blk = 0; blk < (ONEBLK ? 1 : a.n_blocks); ++blk) {
    // Per-iteration opaque copies of the lane coordinates: stops LICM from hoisting every per-lane address
    // of the (long) block body out of this loop, which would pin >100 VGPRs.
    int s8 = s8_; asm volatile("" : "+v"(s8));
    int loff = loff_; asm volatile("" : "+v"(loff));
    int ch = ch_; asm volatile("" : "+v"(ch));
    int lane_i = lane; asm volatile("" : "+v"(lane_i));   // for the rarely taken paths: their addresses must not be hoisted (and spilled)
#ifdef ASDR_TIMELINE
    // (pipeline roles of channel group 0: slots 2 (3 + role) + block parity, entry 16 = the time before the role's waits)
    if (STREAM && a.taps != nullptr && wave_g == 0 && lane == 0) (reinterpret_cast<unsigned long long *>(a.taps) + 64 * (3 + ROLE) + 32 * (blk & 1))[16] = clock64();
#endif
    if (STREAM) {   // streaming pipeline: wait for this block's input rows and for a free slot in the ring this role writes
      uint32_t *prog = a.stream_prog, *err = a.stream_err;
      const uint32_t b1 = (uint32_t)blk + 1u, freed = (blk >= ASDR_STREAM_DEPTH) ? (uint32_t)(blk - ASDR_STREAM_DEPTH + 1) : 0u;
      const bool need_in = (ROLE != 1) && seen_in < b1, need_free = (ROLE != 3) && seen_free < freed;
      if ((need_in || need_free) && sig_pending) { stream_signal(my_prog, sig_pending, lane); sig_pending = 0u; }   // publish before blocking
      if (need_in) { seen_in = stream_wait(prog + (ROLE - 2) * a.stream_waves + wave_g, b1, err, a.stream_spin_limit); if (seen_in == ASDR_STREAM_FAIL) break; }   // the previous role has stored block blk
      if (need_free) { seen_free = stream_wait(prog + ROLE * a.stream_waves + wave_g, freed, err, a.stream_spin_limit); if (seen_free == ASDR_STREAM_FAIL) break; }   // the next role has left slot blk % DEPTH
    }
    const ChanParams *Pp = row_ptr(a.params, (uint32_t)ch * (uint32_t)sizeof(ChanParams));
    ChanSmall *S = row_ptr(a.small, (uint32_t)ch * (uint32_t)sizeof(ChanSmall));
    float *L = lds + loff;
    int *Li = reinterpret_cast<int *>(L);
    // Lanes of the biquad pipelines (biquad_pipe): with the bank-masked select, a row of 16 lanes = 2 channels runs its four cascades (channel
    // 2r / 2r + 1, I / Q) with stage s on lanes 4s .. 4s + 3 of the row: lane (lane & 3) of every group of four = cascade k, stage = (lane & 15) >> 2.
    // pl_c = the wave's channel (0..7) whose cascade this lane works on, pl_iq = its row (I / Q); the audio cascade (one per channel) runs on
    // cascades k = 0, 1 of the row, k = 2, 3 repeat them without writing.  The channel's state / parameter rows come through its index.
    constexpr bool PLB = (ASDR_PIPE_BANK_SELECT != 0);
#define pl_st (PLB ? ((lane_i & 15) >> 2) : (s8 & 3))
#define pl_iq (PLB ? (lane_i & 1) : (s8 >> 2))
#define pl_c (PLB ? (2 * (lane_i >> 4) + ((lane_i >> 1) & 1)) : (lane_i >> 3))
#define pa_c (PLB ? (2 * (lane_i >> 4) + (lane_i & 1)) : (lane_i >> 3))   /* audio cascade: the channel, and whether this lane's cascade is a real one */
#define pa_real (PLB ? ((lane_i & 3) < 2) : ((lane_i & 7) < 4))
    // (everything below is formed where it is used, from the lane index and the channel index: as variables of the whole block they cost the
    // looped instantiations eight registers and spills)
#if ASDR_PIPE_BANK_SELECT
#define PL_CH(c) (__builtin_amdgcn_ds_bpermute((c) << 5, ch))   /* lane 8 c holds channel c's index */
#define Sp row_ptr(a.small, (uint32_t)PL_CH(pl_c) * (uint32_t)sizeof(ChanSmall))
#define Spa row_ptr(a.small, (uint32_t)PL_CH(pa_c) * (uint32_t)sizeof(ChanSmall))
#define Ppl row_ptr(a.params, (uint32_t)PL_CH(pl_c) * (uint32_t)sizeof(ChanParams))
#define Ppa row_ptr(a.params, (uint32_t)PL_CH(pa_c) * (uint32_t)sizeof(ChanParams))
#define Lp (lds + pl_c * STRIDE)
#define Lpa (lds + pa_c * STRIDE)
#else   /* round 5's lanes: a channel's cascades on its own eight lanes */
#define PL_CH(c) ch
#define Sp S
#define Spa S
#define Ppl Pp
#define Ppa Pp
#define Lp L
#define Lpa L
#endif
    const bool af_en_pa = (PLB && !UNIFORM) ? (__builtin_amdgcn_ds_bpermute(pa_c << 5, af_en ? 1 : 0) != 0) : af_en;   // the audio filter's enable of THAT channel (uniform waves: one value)
    // Sample ownership.  Global rows are always touched in whole 16-byte pieces with the 8 lanes of a channel on 8 ADJACENT
    // pieces, so that one wave-instruction reads or writes 128 contiguous bytes of each of its 8 rows (lanes on every second
    // or fourth piece make every instruction touch every line of the rows: 2.9 instead of 7 TB/s, tools/ubench/mem_pattern.hip):
    //   int16 rows (input, output, blanker ring): lane s8 owns samples kA + 64h + j   (h = 0, 1;  j < 8)
    //   float rows (Hilbert rings, ALS rows):     lane s8 owns samples kF + 32m + j   (m = 0..3;  j < 4)
    // Pointwise stages work on the kA map; the FIR stages own 16 contiguous outputs (k0) and exchange through LDS.
    const int k0 = 16 * s8, kA = 8 * s8, kF = 4 * s8;
    const bool lead = (s8 == 0);
#ifdef ASDR_TIMELINE
    // profiling build (tools/timeline.py): lane 0 of a few waves timestamps the phase boundaries into the taps buffer
    // (the four-wave form: all four waves of workgroups 0, 683, 1365, 2047 -> slots 4 i + wave; entries 17..28 = arrival at / departure from the
    // six workgroup barriers, see TL(17) ff.)
    const int tl_wg = ((int)blockIdx.x == 0) ? 0 : (((int)blockIdx.x == 683) ? 1 : (((int)blockIdx.x == 1365) ? 2 : (((int)blockIdx.x == 2047) ? 3 : -1)));
    const int tl_slot = STREAM ? ((wave_g == 0) ? 2 * (3 + ROLE) + (blk & 1) : -1)
                               : (MW ? ((tl_wg < 0) ? -1 : 4 * tl_wg + wave)
                                     : ((wave_g == 0) ? 0 : ((wave_g == 2731) ? 1 : ((wave_g == 5461) ? 2 : ((wave_g == 8191) ? 3 : -1)))));
    unsigned long long *tl = reinterpret_cast<unsigned long long *>(a.taps) + 32 * (tl_slot < 0 ? 0 : tl_slot);
#define TL(i) do { if (a.taps != nullptr && tl_slot >= 0 && lane == 0) tl[i] = clock64(); } while (0)
    const bool tap_on = false;
    TL(0);
    if (MW && a.taps != nullptr && tl_slot >= 0 && lane == 0) { uint32_t hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); tl[29] = hw; }   // SIMD / wave slot / CU of the wave
#else
#define TL(i) do { } while (0)
    const bool tap_on = (a.taps != nullptr) && valid && (blk == a.n_blocks - 1);
#endif
    float *tap_base = tap_on ? a.taps + (size_t)ch * ASDR_N + kA : nullptr;
    const size_t tap_stride = (size_t)a.n_channels * ASDR_N;
#define TAP8(id, h, v) do { if (tap_on) store8(tap_base + (size_t)(id) * tap_stride + 64 * (h), v); } while (0)
#define TAP4(id, m, v) do { if (tap_on) store4(tap_base - kA + kF + (size_t)(id) * tap_stride + 32 * (m), v); } while (0)
#define TAP_ROW(id, rowoff) do { if (tap_on) { for (int h_ = 0; h_ < 2; ++h_) { float tv_[8]; load8(L + (rowoff) + kA + 64 * h_, tv_); \
                                  store8(tap_base + (size_t)(id) * tap_stride + 64 * h_, tv_); } } } while (0)

    uint32_t status = S->status;
    const uint32_t status_in = status;   // (a role's own bits as it found them: nobody else writes those)
    // Small operands first (memory waits count loads in order: what is requested first can be waited for alone): the carried
    // mixer phase of the modes whose mixer does not depend on this block's data, the frequency shift, the oscillator cache's key.
    const bool mix_early = is_ssb || (DO2 && is_am);   // (the pipeline's roles 1 and 3 run no mixer)
    float mphase = 0.0f, minc = 0.0f, mphase_end = 0.0f;   // mphase_end: valid on the lead lanes once the sequence has been computed
    float fsh_raw = 0.0f;
    if (mix_early) {
      if (ROLE == 2 && blk > 0) { mphase = carry_phase; fsh_raw = carry_fsh; }
      else { mphase = is_ssb ? S->phase_ssb : S->phase_am; fsh_raw = P.freq_shift; }
    }
    const float nb_thr = P.nb_threshold;   // used after the blanker's sequential pass: requested here, not there
    // Oscillator pairs of this block: the launch-to-launch cache entry, or -- pipeline role 2 -- the entry the pipeline's
    // oscillator role has left for block blk (EVERY block of the call can hit; read with `sc1` loads, see stream_wait)
    // this wave's entry of the local-oscillator cache: one per settings group (the host numbers the groups; waves of mixed slots
    // and groups beyond ASDR_LO_ENTRIES have none and compute their own pairs)
    const uint32_t lo_e = (lo_slot & 0xFFu) ? (lo_slot & 0xFFu) - 1u : 0u;
    const bool lo_has = (ROLE == 2) || ((lo_slot & 0xFFu) != 0u);
    const LoEntry *lo_rd = (ROLE == 2) ? a.lo_ring + (blk % ASDR_LO_RING) : a.lo_cache + (a.lo_parity & 1u) * ASDR_LO_ENTRIES + lo_e;
    uint32_t lo_kp, lo_ki;
    float lo_end;
    v4f r2_c4[4], r2_s4[4], r2_vi[4], r2_vq[4];
    if (ROLE == 2) {
      v4f key;
      if (R2_PREFETCH && r2_pref) {   // asked for during the previous block (both producers had published this one by then)
        key = r2n_key;
#pragma unroll
        for (int m = 0; m < 4; ++m) { r2_c4[m] = r2n_c4[m]; r2_s4[m] = r2n_s4[m]; r2_vi[m] = r2n_vi[m]; r2_vq[m] = r2n_vq[m]; }
        r2_pref = false;
      } else {
      if (seen_lo < (uint32_t)blk + 1u) { seen_lo = stream_wait(a.stream_prog + 3 * a.stream_waves, (uint32_t)blk + 1u, a.stream_err, a.stream_spin_limit); if (seen_lo == ASDR_STREAM_FAIL) break; }
      if (R2_ONE_TRIP) {
        // Round 6: the key, the pairs and the rows used to be four `sc1` round trips one behind the other (each waited for where it was issued: ~1 k cycles
        // apiece for a wave that is alone on its SIMD); both producers have published this block (the waits above and at the top of the block).
        const float *lp = lo_rd->c + 4 * s8_, *xp = a.xch_a + ((size_t)ch * ASDR_STREAM_DEPTH + (size_t)(blk % ASDR_STREAM_DEPTH)) * (2 * ASDR_N) + 4 * s8_;
        asm volatile("global_load_dwordx4 %0, %17, off sc1\n\t"
                     "global_load_dwordx4 %1, %18, off sc1\n\tglobal_load_dwordx4 %2, %18, off offset:128 sc1\n\tglobal_load_dwordx4 %3, %18, off offset:256 sc1\n\tglobal_load_dwordx4 %4, %18, off offset:384 sc1\n\t"
                     "global_load_dwordx4 %5, %18, off offset:512 sc1\n\tglobal_load_dwordx4 %6, %18, off offset:640 sc1\n\tglobal_load_dwordx4 %7, %18, off offset:768 sc1\n\tglobal_load_dwordx4 %8, %18, off offset:896 sc1\n\t"
                     "global_load_dwordx4 %9, %19, off sc1\n\tglobal_load_dwordx4 %10, %19, off offset:128 sc1\n\tglobal_load_dwordx4 %11, %19, off offset:256 sc1\n\tglobal_load_dwordx4 %12, %19, off offset:384 sc1\n\t"
                     "global_load_dwordx4 %13, %19, off offset:512 sc1\n\tglobal_load_dwordx4 %14, %19, off offset:640 sc1\n\tglobal_load_dwordx4 %15, %19, off offset:768 sc1\n\tglobal_load_dwordx4 %16, %19, off offset:896 sc1\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(key), "=&v"(r2_c4[0]), "=&v"(r2_c4[1]), "=&v"(r2_c4[2]), "=&v"(r2_c4[3]), "=&v"(r2_s4[0]), "=&v"(r2_s4[1]), "=&v"(r2_s4[2]), "=&v"(r2_s4[3]),
                       "=&v"(r2_vi[0]), "=&v"(r2_vi[1]), "=&v"(r2_vi[2]), "=&v"(r2_vi[3]), "=&v"(r2_vq[0]), "=&v"(r2_vq[1]), "=&v"(r2_vq[2]), "=&v"(r2_vq[3])
                     : "v"(lo_rd), "v"(lp), "v"(xp) : "memory");
      } else
      asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(key) : "v"(lo_rd) : "memory");
      }
      lo_kp = __float_as_uint(key[0]); lo_ki = __float_as_uint(key[1]); lo_end = key[2];
    } else {
      lo_kp = lo_rd->key_phase; lo_ki = lo_rd->key_inc; lo_end = lo_rd->phase_end;
    }
    // Ring positions.  The blanker ring advances once per processed block for EVERY channel (a batch-wide block counter):
    // enabling the blanker or changing its threshold resets all three slots (AudioSDR.cpp:653-682), so a channel whose
    // blanker was off meanwhile never sees a stale position -- and no load has to wait for a per-channel slot word.
    const uint32_t ns = (a.nb_phase + (uint32_t)blk) % 3u;                          // oldest NB ring slot (of 3)
    const uint32_t hs = (ROLE == 2 && blk > 0) ? carry_hs : (S->hil_slot & 1u);     // Hilbert ring parity
    const uint32_t ns_mid = (ns + 1u) % 3u, ns_new = (ns + 2u) % 3u;
    float carrier_now = 0.0f;   // set by the envelope path when it runs in this block
    bool carrier_fresh = false;
    const size_t io = ((size_t)ch * a.in_stride + blk) * ASDR_N + kA;       // this lane's input samples: two pieces of 8
    const size_t io_out = ((size_t)ch * a.out_stride + blk) * ASDR_N + kA;  // ... and its output samples (capture rows may be longer)
    const bool nb_wave = ABL_ON(ABL_NB) && __any(nb_en);   // wave-uniform: some channel of this wave has the blanker on
    int16_t *hist = row_ptr(a.nb_hist, (uint32_t)ch * 1536u);  // 3 slots x {I,Q} x 128 raw int16 samples
    uint32_t *mrow = reinterpret_cast<uint32_t *>(row_ptr(a.nb_mask, (uint32_t)ch * ASDR_NB_MASK_ROW + 4u * (uint32_t)s8));   // this lane's mask codes: dwords s8, s8 + 8, .. (entries 32r + 4 s8 ..)

    // ---- every load of the blanker is issued here, together with the input (no dependent address) ----------
    union Raw8 { int4 v; int16_t s[8]; };
    Raw8 ri[2], rq[2], roi[2], roq[2], rmi[2], rmq[2];
    ri[0].v = ri[1].v = rq[0].v = rq[1].v = make_int4(0, 0, 0, 0);
    roi[0].v = roi[1].v = roq[0].v = roq[1].v = rmi[0].v = rmi[1].v = rmq[0].v = rmq[1].v = make_int4(0, 0, 0, 0);
    uint32_t mkc[5] = {0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u};   // defined on every path
    float gain_i, gain_q;
    if (ROLE == 1 && ASDR_STREAM_R1_CARRY && blk > 0) { gain_i = r1_gain_i; gain_q = r1_gain_q; }
    else { gain_i = P.in_gain_i; gain_q = P.in_gain_q; if (ROLE == 1) { r1_gain_i = gain_i; r1_gain_q = gain_q; } }
    float g_oi = gain_i, g_oq = gain_q, g_mi = 0.0f, g_mq = 0.0f, nb_avg0 = 0.0f;
    uint32_t agc_hc_early = 0u, agc_hang_early = 0u;
    bool agc_piped = false;   // (the four-wave form: the AGC duty runs the lean chain beside the audio duty: see the audio filter)
    if (nb_wave && nb_en) {
      // the gains first: they are converted right away, and a wait for the oldest loads leaves all the others in flight
      g_oi = S->nb_gain[ns][0]; g_oq = S->nb_gain[ns][1]; g_mi = S->nb_gain[ns_mid][0]; g_mq = S->nb_gain[ns_mid][1];
      nb_avg0 = S->nb_avg;
      if constexpr (PIPE_AGC) { agc_hc_early = S->agc_hang_counter; agc_hang_early = P.agc_hang_count; }   // (for the decision "AGC chain beside the audio duty", published with the blanker chain's inputs)
   // (for the decision "AGC chain beside the audio duty", published with the blanker chain's inputs)
      const int4 *old4 = reinterpret_cast<const int4 *>(hist + ns * 256 + kA), *mid4 = reinterpret_cast<const int4 *>(hist + ns_mid * 256 + kA);
#if ASDR_NT_LOADS >= 2
      rmi[0].v = mid4[0]; rmi[1].v = mid4[8]; rmq[0].v = mid4[16]; rmq[1].v = mid4[24];   // (the middle block comes back once more: temporal)
      roi[0].v = load_int4_nt(old4); roi[1].v = load_int4_nt(old4 + 8); roq[0].v = load_int4_nt(old4 + 16); roq[1].v = load_int4_nt(old4 + 24);
#else
      rmi[0].v = mid4[0]; rmi[1].v = mid4[8]; rmq[0].v = mid4[16]; rmq[1].v = mid4[24];   // Q row starts 128 samples = 16 int4 later
      roi[0].v = old4[0]; roi[1].v = old4[8]; roq[0].v = old4[16]; roq[1].v = old4[24];
#endif
#pragma unroll
      for (int r = 0; r < 5; ++r) mkc[r] = mrow[8 * r];   // codes of the carried mask[128..265]
      if constexpr (C16) {   // all five collapse into one word here (all ones or not); see the quiet test
        const uint32_t all1 = (mkc[0] & mkc[1] & mkc[2] & mkc[3] & mkc[4]) == 0x01010101u && (mkc[0] | mkc[1] | mkc[2] | mkc[3] | mkc[4]) == 0x01010101u ? 0x01010101u : 0u;
#pragma unroll
        for (int r = 0; r < 5; ++r) mkc[r] = all1;
      }
    }
    // The input rows come LAST: with the blanker on, this block's samples only go into the ring (the chain works on the block that
    // arrived two calls ago), so nothing below waits for them until that store -- and memory waits count loads in order.
    if (DO1 && valid) {
      const int4 *pi = reinterpret_cast<const int4 *>(a.in_i + io);
      const int4 *pq = reinterpret_cast<const int4 *>(a.in_q + io);
      if constexpr (ROLE == 1 && ASDR_STREAM_PREFETCH_IN != 0) {
        // the pipeline's role 1: this block's rows were requested a block ago; the next block's are requested now (a role wave is alone on its
        // SIMD: nothing hides the round trip for it)
        if (blk == 0) { ri[0].v = load_int4_nt(pi); ri[1].v = load_int4_nt(pi + 8); rq[0].v = load_int4_nt(pq); rq[1].v = load_int4_nt(pq + 8); }
        else { ri[0].v = pf_in[0]; ri[1].v = pf_in[1]; rq[0].v = pf_in[2]; rq[1].v = pf_in[3]; }
        // (the next block's rows are requested in front of the IF pipeline, behind the late signal: its wait must not sit out these loads)
      } else {
#if ASDR_NT_LOADS >= 1
      ri[0].v = load_int4_nt(pi); ri[1].v = load_int4_nt(pi + 8); rq[0].v = load_int4_nt(pq); rq[1].v = load_int4_nt(pq + 8);
#else
      ri[0].v = pi[0]; ri[1].v = pi[8]; rq[0].v = pq[0]; rq[1].v = pq[8];   // int4 #8 = 64 samples on
#endif
      }
    }
    // mixer increment (AudioSDR.h:508-512) and the local-oscillator cache test (asdr_device.h LoEntry): hit = every channel of the
    // wave starts this block with exactly the cached phase and increment -> no recurrence, no table lookups, the pairs are read
    // from the entry.  (The operands were requested before the input rows: the test does not wait for those.)
    if (mix_early) minc = (is_ssb ? -fsh_raw : -K.if_center) * K.phase_inc_unit;
    bool ph_ready = false;   // this channel's phase sequence is in its PH row
    bool lo_hit = false;
    if ((blk == 0 || ROLE == 2) && lo_has) lo_hit = __all(mix_early && __float_as_uint(mphase) == lo_kp && __float_as_uint(minc) == lo_ki);
    if (lo_hit && lead) { if (is_ssb) S->phase_ssb = lo_end; else S->phase_am = lo_end; }

    // ---- input scale, AudioSDR.cpp:67-70: ((float)s / 32767.0) * gain in binary64, stored float --------
    // With a blanker in the wave, the blanker's delay line is kept as RAW int16 samples plus the gains that were in
    // force when each block arrived: the scaled float is an exact function of (sample, gain), so re-scaling on
    // read reproduces the reference's stored floats bit for bit at a quarter of the HBM traffic.
    if (nb_wave) {
      // (the newest block goes to its ring slot after the envelopes, below)
    } else if (DO1) {
      const bool unit_in = UNIT_OK && __all(gain_i == 1.0f && gain_q == 1.0f);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float xi[8], xq[8];
        scale8(ri[h].s, (double)gain_i, xi, unit_in); scale8(rq[h].s, (double)gain_q, xq, unit_in);
        TAP8(ASDR_TAP_SCALED_I, h, xi); TAP8(ASDR_TAP_SCALED_Q, h, xq);
        store8(L + W0 + kA + 64 * h, xi); store8(L + W1 + kA + 64 * h, xq);
        SCHED_FENCE();
      }
    }

    float4 if_s4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float if_cf[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    bool if_pre = false;
    auto load_if_rows_early = [&]() {
      if_s4 = *reinterpret_cast<const float4 *>(&Sp->if_state[pl_iq][4 * pl_st]);
      const float *cf = &c_bq_pool[Ppl->if_table][5 * pl_st];
#pragma unroll
      for (int z = 0; z < 5; ++z) if_cf[z] = cf[z];
      if_pre = true;
    };
    TL(1);
    // ---- impulse noise blanker, AudioSDR.cpp:606-650 ------------------------------------------------------
    // Buffer coordinates as in the reference: [0,128) oldest, [128,256) middle, [256,384) newest.  The 3-slot ring
    // in HBM holds them; output = mask x oldest (2 blocks late).
    if (nb_wave) {
      // This lane's 16 samples of the OLDEST block, scaled once: they are the blanker's output (before masking) and, for
      // samples 78..127, part of the re-scanned envelope (:627).  Blanker-off channels pass their own input.
      float vio[16], vqo[16], mgm[16], mgt[8];
      if (!nb_en) {
#pragma unroll
        for (int h = 0; h < 2; ++h) { roi[h].v = ri[h].v; roq[h].v = rq[h].v; }
      }
      // envelope for detection indices i = 78..255 -> t = i - 78 (fast_sqrt_f32(I^2+Q^2, 1), :628): t = 50 + k for this lane's
      // samples k of the middle block, t = k - 78 for its samples k >= 78 of the oldest block (the second piece, 64 + 8 s8 + j,
      // of lanes s8 >= 1).  beta * envelope goes to the B row (index NB_B + t) for the sequential pass; the envelopes
      // themselves stay in registers for the threshold test.
      const bool own_tail = (s8 >= 1);
      if (C16 && nb_en) {   // the 16-waves-per-CU form sends the newest block to its ring slot FIRST: 16 registers less across the envelope pass
        int4 *ni = reinterpret_cast<int4 *>(hist + ns_new * 256 + kA);
        store_int4_nt(ni, ri[0].v); store_int4_nt(ni + 8, ri[1].v); store_int4_nt(ni + 16, rq[0].v); store_int4_nt(ni + 24, rq[1].v);
        if (lead) { S->nb_gain[ns_new][0] = gain_i; S->nb_gain[ns_new][1] = gain_q; }
      }
      // (every gain the wave's ring slots arrived with is 1.0: the two-operation binary32 scale, see scale8)
      const bool unit_ring = UNIT_OK && __all(g_oi == 1.0f && g_oq == 1.0f && (!nb_en || (g_mi == 1.0f && g_mq == 1.0f)));
      // (the choice is made per scale call and per GROUP OF FOUR envelopes: a test per envelope keeps the divisions of a piece from interleaving,
      // the whole pass in two copies costs the loop-free kernels 2 - 26 spilled registers)
      auto env4 = [&](const float *pw4, float *mg4) {
        if (unit_ring) {
#pragma unroll
          for (int j = 0; j < 4; ++j) mg4[j] = fast_sqrt1_short(pw4[j]);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) mg4[j] = fast_sqrt1(pw4[j]);
        }
      };
      if constexpr (C16) {
        // The 16-waves-per-CU form keeps the oldest block RAW across the sequential pass (16 registers instead of 32: it is scaled again
        // for the output below) and takes the pieces in an order that lets every scaled piece die right behind its envelopes: middle
        // block's first pieces | oldest block's second pieces (the re-scanned envelopes) | middle block's second pieces.
        auto mid_piece = [&](int h) {
          float vim[8], vqm[8], bm8[8], pw[8];
          scale8(rmi[h].s, (double)g_mi, vim, unit_ring); scale8(rmq[h].s, (double)g_mq, vqm, unit_ring);
#pragma unroll
          for (int j = 0; j < 8; ++j) pw[j] = vim[j] * vim[j] + vqm[j] * vqm[j];
          env4(pw, mgm + 8 * h); env4(pw + 4, mgm + 8 * h + 4);
#pragma unroll
          for (int j = 0; j < 8; ++j) bm8[j] = K.nb_beta * mgm[8 * h + j];
          if (nb_en) store8(L + NB_B + 50 + kA + 64 * h, bm8);
        };
        mid_piece(0);
        __builtin_amdgcn_sched_barrier(0);
        if (own_tail) {
          float tio[8], tqo[8], bm8[8], pw[8];
          scale8(roi[1].s, (double)g_oi, tio, unit_ring); scale8(roq[1].s, (double)g_oq, tqo, unit_ring);
#pragma unroll
          for (int j = 0; j < 8; ++j) pw[j] = tio[j] * tio[j] + tqo[j] * tqo[j];
          env4(pw, mgt); env4(pw + 4, mgt + 4);
#pragma unroll
          for (int j = 0; j < 8; ++j) bm8[j] = K.nb_beta * mgt[j];
          if (nb_en) { if (s8 >= 2) store8(L + kA - 12, bm8); else *reinterpret_cast<float4 *>(L) = make_float4(bm8[4], bm8[5], bm8[6], bm8[7]); }
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) mgt[j] = 0.0f;
        }
        __builtin_amdgcn_sched_barrier(0);
        mid_piece(1);
      } else {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float vim[8], vqm[8], bm8[8], pw[8];
        scale8(roi[h].s, (double)g_oi, vio + 8 * h, unit_ring); scale8(roq[h].s, (double)g_oq, vqo + 8 * h, unit_ring);
        scale8(rmi[h].s, (double)g_mi, vim, unit_ring); scale8(rmq[h].s, (double)g_mq, vqm, unit_ring);
#pragma unroll
        for (int j = 0; j < 8; ++j) pw[j] = vim[j] * vim[j] + vqm[j] * vqm[j];
        env4(pw, mgm + 8 * h); env4(pw + 4, mgm + 8 * h + 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) bm8[j] = K.nb_beta * mgm[8 * h + j];
        if (nb_en) store8(L + NB_B + 50 + kA + 64 * h, bm8);
      }
      if (own_tail) {
        float bm8[8], pw[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) pw[j] = vio[8 + j] * vio[8 + j] + vqo[8 + j] * vqo[8 + j];
        env4(pw, mgt); env4(pw + 4, mgt + 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) bm8[j] = K.nb_beta * mgt[j];
        // index NB_B + k - 78 = 8 s8 + j - 12: lane 1 holds k = 72..79, of which 76, 77 land on the two padding words below NB_B
        if (nb_en) { if (s8 >= 2) store8(L + kA - 12, bm8); else *reinterpret_cast<float4 *>(L) = make_float4(bm8[4], bm8[5], bm8[6], bm8[7]); }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) mgt[j] = 0.0f;
      }
      }   // !C16
      if (nb_en && !C16) {   // newest block -> third ring slot, with its gains (blanker-off channels pass their own input: see above)
        int4 *ni = reinterpret_cast<int4 *>(hist + ns_new * 256 + kA);
#ifndef ASDR_TEMPORAL_RINGS
        store_int4_nt(ni, ri[0].v); store_int4_nt(ni + 8, ri[1].v); store_int4_nt(ni + 16, rq[0].v); store_int4_nt(ni + 24, rq[1].v);
#else
        ni[0] = ri[0].v; ni[8] = ri[1].v; ni[16] = rq[0].v; ni[24] = rq[1].v;
#endif
        if (lead) { S->nb_gain[ns_new][0] = gain_i; S->nb_gain[ns_new][1] = gain_q; }
      }
      if (tap_on) {
#pragma unroll
        for (int h = 0; h < 2; ++h) { float xi[8], xq[8]; scale8(ri[h].s, (double)gain_i, xi); scale8(rq[h].s, (double)gain_q, xq);
                                      TAP8(ASDR_TAP_SCALED_I, h, xi); TAP8(ASDR_TAP_SCALED_Q, h, xq); }
      }
      WAVE_SYNC();
      TL(2);
#ifndef ASDR_MW_IF_PRE
#define ASDR_MW_IF_PRE 1
#endif
      if (DO1 && ABL_ON(ABL_IF) && (WAVES == 1 || (MW && ASDR_MW_IF_PRE)) && !C16) {   // the IF pipeline's state and coefficient row are requested here: they arrive during the sequential pass
                                            // (not in the multi-wave SAM instantiation: no registers to spare)
        if_s4 = *reinterpret_cast<const float4 *>(&Sp->if_state[pl_iq][4 * pl_st]);
        const float *cf = &c_bq_pool[Ppl->if_table][5 * pl_st];
#pragma unroll
        for (int z = 0; z < 5; ++z) if_cf[z] = cf[z];
        if_pre = true;
      }
      // sequential, one lane per channel: the running average (:633-634), which does not depend on the detections:
      // avg = alpha*avg + beta*mag; avg[t] (the average BEFORE sample t) replaces beta*mag[t] in place.  The mixer's phase
      // recurrence (AudioSDR.h:513-518) is an independent dependent chain and rides in the same loop.
      // The two recurrences of one channel on ONE lane (`Lc` = the channel's rows, avg0 / phase0 / inc = its carried average, mixer phase
      // and increment, Sc = its state row, chain_phase = the phase sequence is wanted (wave-uniform: a lane that does not need it computes
      // and drops it), en = the blanker is on).  Returns the phase after the block.
#ifndef ASDR_CHAIN_CONSTS_IN_VGPRS
#define ASDR_CHAIN_CONSTS_IN_VGPRS 0
#endif
#ifndef ASDR_MW_OWN_STORES
#define ASDR_MW_OWN_STORES 1   /* the four-wave form: a chain's results go back to HBM from the channel's OWN wave (its lead lanes: stores beside its other state), not from the duty wave's 32 lanes */
#endif
      float nb_avg_end = 0.0f;   // (MW with ASDR_MW_OWN_STORES: the duty wave's lane leaves the average here instead of storing it)
      auto nb_chain = [&](float *Lc, ChanSmall *Sc, float avg0, float phase0, float inc, bool chain_phase, bool want_phase, bool en, bool ssb) -> float {
        float avg = avg0, phase = phase0;
        // The chain's constants as VGPR operands (round 6): as scalar operands they are in every other instruction of the loops below, and a stream
        // that dense in scalar-file reads does not share its SIMD -- each resident wave then issues once per 8 cycles instead of 4.5
        // (tools/ubench/issue_rate.hip: chain2 against chain4x_vgpr_constants).  Not in the 16-waves-per-CU form (no register to spare).
        float nb_alpha_v = K.nb_alpha, two_pi_v = two_pi;
        if constexpr (!C16 && ONEBLK && ASDR_CHAIN_CONSTS_IN_VGPRS) asm("" : "+v"(nb_alpha_v), "+v"(two_pi_v));   // (nor in the looped forms: they spill with it)
        // `if (t > twoPI) t -= twoPI; else if (t < 0) t += twoPI;` (.h:514-517) with one test per sample: the phase stays in
        // [0, twoPI], so for inc >= 0 only the first branch can fire and for inc < 0 only the second.  The test is written as
        // (t with its sign flipped for inc < 0) > (twoPI or 0), and t - twoPI == t + (-twoPI) exactly.
        const bool up = !(inc < 0.0f);
        const float wrapv = up ? -two_pi : two_pi, lim = up ? two_pi : 0.0f;
        const uint32_t flip = up ? 0u : 0x80000000u;
#pragma unroll
        for (int u = 0; u < 2; ++u) { const float bmu = Lc[NB_B + u]; Lc[NB_B + u] = avg; const float aa = nb_alpha_v * avg; avg = aa + bmu; }
        float bm[8];
        load8(Lc + NB_B + 2, bm);
        int c_tail = 0;   // first chunk of the average-only loop below
        if constexpr (C16) {
          // The 16-waves-per-CU form of both loops: the averages replace the chunk IN its registers (no second array), the phases of the
          // even samples only are kept: 20 registers instead of 32.  Same operations per sample.
          float bn[8];
          const int n_both = chain_phase ? 16 : 0;
#pragma unroll 1
          for (int c = 0; c < 22; c += 2) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
              float *cur = half ? bn : bm, *nxt = half ? bm : bn;
              load8(Lc + NB_B + 2 + 8 * ((c + half < 21) ? c + half + 1 : 21), nxt);
              if (c < n_both) {   // (wave-uniform)
                float pe[4];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                  const float a0 = avg; const float aa = nb_alpha_v * avg; avg = aa + cur[u]; cur[u] = a0;
                  if ((u & 1) == 0) pe[u >> 1] = phase;
                  const float t = phase + inc, tw = t + wrapv;
                  phase = (__uint_as_float(__float_as_uint(t) ^ flip) > lim) ? tw : t;
                }
                store4(Lc + PH + 4 * (c + half), pe);
              } else {
#pragma unroll
                for (int u = 0; u < 8; ++u) { const float a0 = avg; const float aa = nb_alpha_v * avg; avg = aa + cur[u]; cur[u] = a0; }
              }
              store8(Lc + NB_B + 2 + 8 * (c + half), cur);
            }
          }
          c_tail = n_both;
        } else {
        if (chain_phase) {
        c_tail = 16;
        // Both chains in ONE basic block (the scheduler interleaves them).  Every mode's shift is downwards (inc < 0): then the
        // wrap `t < 0 ? t + twoPI : t` is a sign-mask select (v_ashrrev + v_bfi, no compare -> VCC -> select hazard): the phase
        // chain is add, shift, select.  (t = -0.0 cannot occur: the phase is never -0.0 and x + y = -0.0 needs both -0.0.)
        if (!(HAS_ALS && !ALS_FULL_OPT2) && __all(inc < 0.0f)) {   // (the ALS instantiations take the compact loop below: this one costs them spills)
#pragma unroll 1
          for (int c = 0; c < 16; c += 2) {
            float av[8], bn[8], pv[8];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
              float *cur = half ? bn : bm, *nxt = half ? bm : bn;   // ping-pong: no register copies
              load8(Lc + NB_B + 2 + 8 * (c + half + 1), nxt);      // next chunk, a step ahead
#pragma unroll
              for (int u = 0; u < 8; ++u) {
                av[u] = avg; const float aa = nb_alpha_v * avg; avg = aa + cur[u];
                pv[u] = phase;
                const float t = phase + inc, tw = t + two_pi_v;
                const uint32_t tb = __float_as_uint(t), m = (uint32_t)((int32_t)tb >> 31);
#ifdef ASDR_PHASE_SELECT_C
                phase = __uint_as_float((__float_as_uint(tw) & m) | (tb & ~m));
#else
                // (m & tw) | (~m & t) as ONE v_bfi_b32: from the C expression the compiler builds v_max_i32(0, t) + v_and_or_b32
                asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(phase) : "v"(m), "v"(tw), "v"(t));
#endif
              }
              if constexpr (C16) store4e(Lc + PH + 4 * (c + half), pv); else store8(Lc + PH + 8 * (c + half), pv);
              store8(Lc + NB_B + 2 + 8 * (c + half), av);
            }
          }
        } else {
#pragma unroll 1
          for (int c = 0; c < 16; ++c) {
            float av[8], bn[8], pv[8];
            load8(Lc + NB_B + 2 + 8 * (c + 1), bn);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              av[u] = avg; const float aa = nb_alpha_v * avg; avg = aa + bm[u];
              pv[u] = phase;
              const float t = phase + inc, tw = t + wrapv;
              phase = (__uint_as_float(__float_as_uint(t) ^ flip) > lim) ? tw : t;
            }
            if constexpr (C16) store4e(Lc + PH + 4 * c, pv); else store8(Lc + PH + 8 * c, pv);
            store8(Lc + NB_B + 2 + 8 * c, av);
#pragma unroll
            for (int u = 0; u < 8; ++u) bm[u] = bn[u];
          }
        }
        }   // chain_phase
#pragma unroll 1
        for (int c = c_tail; c < 22; c += 2) {   // c_tail is even; two chunks per trip, ping-pong like above
          float av[8], bn[8];
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            float *cur = half ? bn : bm, *nxt = half ? bm : bn;
            load8(Lc + NB_B + 2 + 8 * ((c + half < 21) ? c + half + 1 : 21), nxt);
#pragma unroll
            for (int u = 0; u < 8; ++u) { av[u] = avg; const float aa = nb_alpha_v * avg; avg = aa + cur[u]; }
            store8(Lc + NB_B + 2 + 8 * (c + half), av);
          }
        }
        }   // !C16
        constexpr bool OWN_STORES = MW && ((MW_SHARE & 2) != 0) && (ASDR_MW_OWN_STORES != 0);
        if (OWN_STORES) nb_avg_end = avg;
        if (!OWN_STORES && en) Sc->nb_avg = avg;
        if (!OWN_STORES && want_phase && c_tail != 0) {
          // (the word's offset is formed HERE: hoisted in front of the block loop as a 64-bit select it costs the multi-wave SAM instantiation a spill)
          uint32_t off = (uint32_t)offsetof(ChanSmall, phase_am);
          asm volatile("" : "+v"(off));
          if (ssb) off = (uint32_t)offsetof(ChanSmall, phase_ssb);
          *reinterpret_cast<float *>(reinterpret_cast<char *>(Sc) + off) = phase;
        }
        return phase;
      };
      const bool chain_phase_own = !lo_hit && (DO2 || !UNIFORM);
      if constexpr ((MW_SHARE & 2) != 0) {
        // every channel's lead lane publishes its chain inputs; the duty wave (rel 0) runs the chains of all 8 * WAVES channels, one per lane
        if (lead) *reinterpret_cast<float4 *>(mwx + MWX * (wave * 8 + c8)) = make_float4(nb_avg0, mphase, minc, __int_as_float((chain_phase_own ? 1 : 0) | ((mix_early && !lo_hit) ? 2 : 0)));
        if constexpr (PIPE_AGC) {
          // bit 0: this channel's AGC is off or in the lean regime (no hang counter can run out inside the block); bit 1: it attacked in the PREVIOUS block
          // (an attack at sample u leaves the counter at hang - (127 - u)).  See the audio filter: the AGC chain beside the audio duty.
          if (lead) {
            const bool capable = af_en && nb_en && (!agc_en || (!is_am && agc_hc_early >= 128u && agc_hang_early >= 128u));
            const bool recent = agc_en && (agc_hang_early - agc_hc_early <= 127u);
            mwx[MWX * (wave * 8 + c8) + 7] = __int_as_float((capable ? 1 : 0) | (recent ? 2 : 0));
          }
        }
        if constexpr (MW_SINE_LAZY) {
          const uint32_t p1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(mphase)), i1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(minc));
          const bool differ = mix_early && !__all(__float_as_uint(mphase) == p1 && __float_as_uint(minc) == i1);
          if (lane_i == 0) mw_sflag[wave] = differ ? 1 : 0;
        }
        TL(17);
        __syncthreads();
        TL(18);
        if constexpr (MW_SINE_LAZY) {
          const int4 sf = *reinterpret_cast<const int4 *>(mw_sflag);
          if (__builtin_amdgcn_readfirstlane(sf.x | sf.y | sf.z | sf.w) != 0 && mw_rel != 0) {   // the three waves that would park now fill the table (visible behind the second barrier)
#pragma unroll 1
            for (int i = (mw_rel - 1) * 64 + lane_i; i < ASDR_SINE_TABLE_LEN; i += 192) { sine_tab[i] = c_sine[i]; asm volatile("" ::: "memory"); }   // (one entry in flight: this is the kernel's register peak)
          }
        }
        if (ASDR_MW_PRIO && mw_rel == 0) __builtin_amdgcn_s_setprio(ASDR_MW_PRIO);
        if (mw_rel == 0 && lane_i < 8 * WAVES) {
          const int q = lane_i;
          const float4 in4 = *reinterpret_cast<const float4 *>(mwx + MWX * q);
          const int fl = __float_as_int(in4.w);
          const bool any_phase = __any((fl & 1) != 0);
          const float pe = nb_chain(lds_wg + q * STRIDE, row_ptr(a.small, (uint32_t)mw_channel(q) * (uint32_t)sizeof(ChanSmall)), in4.x, in4.y, in4.z, any_phase, (fl & 2) != 0, nb_en, is_ssb);
          mwx[MWX * q + 4] = pe;
          if (ASDR_MW_OWN_STORES) { mwx[MWX * q] = nb_avg_end; mwx[MWX * q + 1] = __int_as_float(any_phase ? 1 : 0); }
        }
#ifndef ASDR_MW_PRIO_REL0_TAIL
#define ASDR_MW_PRIO_REL0_TAIL 0   /* experiments: the priority the blanker-chain duty wave keeps up to the audio barrier (the timeline's straggler there) */
#endif
        if (ASDR_MW_PRIO && mw_rel == 0) __builtin_amdgcn_s_setprio(ASDR_MW_PRIO_REL0_TAIL);
        TL(19);
        __syncthreads();
        TL(20);
        if constexpr (PIPE_AGC) {   // (every wave forms the decision from the 32 channels' words; nothing waits for this read before the audio barrier)
          const int fl = __float_as_int(mwx[MWX * (lane_i & (8 * WAVES - 1)) + 7]);
          agc_piped = __all((fl & 1) != 0) && __any((fl & 2) != 0);
        }
        if (lead) {
          mphase_end = mwx[MWX * (wave * 8 + c8) + 4];
          if (ASDR_MW_OWN_STORES) {   // the chain's results back to the channel's state row, from its own wave
            if (nb_en) S->nb_avg = mwx[MWX * (wave * 8 + c8)];
            if (mix_early && !lo_hit && __float_as_int(mwx[MWX * (wave * 8 + c8) + 1]) != 0) { if (is_ssb) S->phase_ssb = mphase_end; else S->phase_am = mphase_end; }
          }
        }
      } else {
        CHAIN_PRIO_ON();
        if (lead) mphase_end = nb_chain(L, S, nb_avg0, mphase, minc, chain_phase_own, mix_early && !lo_hit, nb_en, is_ssb);
        CHAIN_PRIO_OFF();
      }
      ph_ready = mix_early && !lo_hit;
      WAVE_SYNC();
      // parallel: threshold test mag[t] > avg[t]*threshold (:628) on the lanes that hold the envelopes
      // detection flags, shifted in one sample at a time (x + x + flag: no `1 << k` literals, which the compiler would park in nine
      // VGPRs for the whole kernel): bit 15 - (8h + j) <-> this lane's sample kA + 64h + j of the middle block / bit 7 - j <-> 64 + kA + j of the oldest
      uint32_t fm = 0u, ft = 0u;
      if (nb_en) {
        const float thr = nb_thr;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          float av[8];
          load8(L + NB_B + 50 + kA + 64 * h, av);
#pragma unroll
          for (int j = 0; j < 8; ++j) fm = fm + fm + ((mgm[8 * h + j] > av[j] * thr) ? 1u : 0u);
        }
        if (own_tail) {
          float av[8];
          if (s8 >= 2) load8(L + kA - 12, av);
          else { const float4 t4 = *reinterpret_cast<const float4 *>(L); av[0] = av[1] = av[2] = av[3] = 0.0f; av[4] = t4.x; av[5] = t4.y; av[6] = t4.z; av[7] = t4.w; }
#pragma unroll
          for (int j = 0; j < 8; ++j) ft = ft + ft + (((64 + kA + j >= 78) && (mgt[j] > av[j] * thr)) ? 1u : 0u);
        }
      }
      TL(3);
      if (C16 && DO1 && ABL_ON(ABL_IF)) { load_if_rows_early(); }   // (the envelopes are dead: the IF pipeline's state and coefficients are asked for here)
      // Quiet fast path (wave-uniform): if no channel of the wave has a detection in this block and every carried mask
      // entry is 1.0, the mask stays all ones -- no zeroing, no trailing ramp, the carried row is unchanged and the output is
      // the oldest block times 1.0 (x * 1.0f == x for every float) -- so counts / decode / zero / ramp / encode / multiply are skipped.
      const unsigned long long det_bal = __ballot((fm | ft) != 0u);
      const bool ch_det = ((uint32_t)(det_bal >> (8 * c8)) & 0xFFu) != 0u;   // some lane of this channel saw a detection
      bool nb_quiet = !ch_det;
      if (nb_en && lead) status = (status & ~ASDR_S_NB_DETECTED) | (ch_det ? ASDR_S_NB_DETECTED : 0u);
#pragma unroll
      for (int r = 0; r < 5; ++r) nb_quiet = nb_quiet && (mkc[r] == 0x01010101u);
      if constexpr (C16) {   // (the 16-waves-per-CU form: the five code words are not kept across the block -- the general path, rare, asks for them again)
        if (__any(!nb_quiet) && nb_en) {
#pragma unroll
          for (int r = 0; r < 5; ++r) mkc[r] = mrow[8 * r];
        }
      }
#ifdef ASDR_NB_ALWAYS_SLOW
      const bool nb_slow = true;
#else
      const bool nb_slow = __any(!nb_quiet);
#endif
      if constexpr (C16) {   // the oldest block, scaled for the output (see the envelope pass)
        asm volatile("" : "+v"(roi[0].v.x), "+v"(roi[1].v.x), "+v"(roq[0].v.x), "+v"(roq[1].v.x));   // (a NEW value chain: nothing of the first scale is kept alive)
#pragma unroll
        for (int h = 0; h < 2; ++h) { scale8(roi[h].s, (double)g_oi, vio + 8 * h, unit_ring); scale8(roq[h].s, (double)g_oq, vqo + 8 * h, unit_ring); }
      }
      if (nb_slow) {
        // General path, on the mask's BYTE CODES (0..6 <-> {0, 1, .933, .75, .5, .25, .067}: the only values the mask ever holds,
        // AudioSDR.cpp:608, 623, 630): the carried codes go to LDS as they came from HBM, zeroing and the trailing ramp write code
        // bytes, the carry-out is copied back as it is -- no float mask row, no encode / decode round trip (round 2 kept the mask
        // as floats in LDS: 85 decode + ~110 encode instructions per lane and block).  Only this lane's 16 output samples are
        // decoded, for the multiply.  Both arrays live in the dead B row, so the phase sequence in the PH row survives.
        const float mask_tab = c_mask_val[lane_i & 7];   // code -> value, one entry per lane (mask_decode_byte)
        const uint32_t group4 = (uint32_t)(lane_i & 56) << 2;
        uint8_t *const mb = reinterpret_cast<uint8_t *>(Li + NB_MSKB);   // mb[m] = code of mask[m], m = 0..271
        WAVE_SYNC();               // every lane has read its averages: the B row is dead
        if (nb_en) {
          // running detection counts as bytes: CB[0..23] = 0, CB[24 + t] = detections up to and including index t (t = 0..177),
          // CB[202..223] = the block's total.  Order of t: oldest-block samples 78..127 (second pieces of lanes 1..7), then the
          // middle block's first pieces (samples 8 s8 + j of lanes 0..7), then its second pieces (64 + 8 s8 + j).
          auto scan8 = [&](uint32_t v) {   // inclusive scan over the channel's 8 lanes (DPP row_shr inside the 16-lane row, masked by s8)
            { const uint32_t t1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true); v += (s8 >= 1) ? t1 : 0u; }
            { const uint32_t t2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true); v += (s8 >= 2) ? t2 : 0u; }
            { const uint32_t t4 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true); v += (s8 >= 4) ? t4 : 0u; }
            return v;
          };
          const uint32_t cnt_t = (uint32_t)__builtin_popcount(ft), cnt_a = (uint32_t)__builtin_popcount(fm >> 8), cnt_b = (uint32_t)__builtin_popcount(fm & 0xFFu);
          const uint32_t inc_t = scan8(cnt_t), inc_a = scan8(cnt_a), inc_b = scan8(cnt_b);
          const int last = (lane_i & ~7) | 7;
          const uint32_t tot_t = (uint32_t)__builtin_amdgcn_ds_bpermute(last << 2, (int)inc_t), tot_a = (uint32_t)__builtin_amdgcn_ds_bpermute(last << 2, (int)inc_a),
                         tot_m = tot_a + (uint32_t)__builtin_amdgcn_ds_bpermute(last << 2, (int)inc_b);   // the totals sit on the channel's last lane
          uint8_t *cbw = reinterpret_cast<uint8_t *>(Li + NB_CB);
          uint32_t run = inc_t - cnt_t;
          if (own_tail) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { run += (ft >> (7 - j)) & 1u; if (64 + kA + j >= 78) cbw[24 + 64 + kA + j - 78] = (uint8_t)run; }
          }
          run = tot_t + inc_a - cnt_a;
#pragma unroll
          for (int j = 0; j < 8; ++j) { run += (fm >> (15 - j)) & 1u; cbw[24 + 50 + kA + j] = (uint8_t)run; }
          run = tot_t + tot_a + inc_b - cnt_b;
#pragma unroll
          for (int j = 0; j < 8; ++j) { run += (fm >> (7 - j)) & 1u; cbw[24 + 50 + 64 + kA + j] = (uint8_t)run; }
          if (lead) {
            const uint32_t fin = ((tot_t + tot_m) & 0xFFu) * 0x01010101u;
#pragma unroll
            for (int z = 0; z < 6; ++z) Li[NB_CB + z] = 0;                                               // bytes 0..23
            cbw[202] = (uint8_t)fin; cbw[203] = (uint8_t)fin;
#pragma unroll
            for (int z = 51; z < 56; ++z) Li[NB_CB + z] = (int)fin;                                      // bytes 204..223
          }
          // mask codes: carried part = the previous call's mask[128..265], as it came from HBM (lane s8 holds entries 32r + 4 s8 .. + 3
          // of its five code words; entries >= 138 of the row are padding with code 1 = 1.0), then the rest of the (new) newest
          // block is 1.0 (:621-623): code words 40..67 (entries 160..271)
#pragma unroll
          for (int r = 0; r < 5; ++r) Li[NB_MSKB + 8 * r + s8] = (int)mkc[r];
#pragma unroll
          for (int k = 0; k < 4; ++k) { const int w = 40 + s8 + 8 * k; if (w < 68) Li[NB_MSKB + w] = 0x01010101; }
        }
        WAVE_SYNC();
        if (nb_en) {   // zero mask[i-10 .. i+10] around every detection (:630); all writes are code 0, so order-free:
                       // mask[m] is hit iff a detection index lies in [m-10, m+10] iff CB[m-44] - CB[m-65] > 0.
          // Four entries per operation (bytes of a word): lane s8 owns the code words 17 + 7 s8 .. + 6 (entries 68 + 28 s8 ..; words
          // from 67 on lie behind the last entry a detection can reach: for them the two counts are the block's total and their
          // difference 0).  The counts at m - 44 start on a word of CB, the counts at m - 65 three bytes into one (v_alignbyte).
          const int wz = 17 + 7 * s8;
          const uint32_t *cbw32 = reinterpret_cast<const uint32_t *>(Li + NB_CB);
          uint32_t hi[7], lo[8], old[7];
#pragma unroll
          for (int z = 0; z < 7; ++z) { hi[z] = cbw32[wz + z - 11]; old[z] = (uint32_t)Li[NB_MSKB + wz + z]; }
#pragma unroll
          for (int z = 0; z < 8; ++z) lo[z] = cbw32[wz + z - 17];
#pragma unroll
          for (int z = 0; z < 7; ++z) {
            const uint32_t lo3 = __builtin_amdgcn_alignbyte(lo[z + 1], lo[z], 3);   // counts at m - 65 for the word's four entries
            const uint32_t d = hi[z] - lo3;                                           // bytewise: the counts only grow and differ by <= 21
            const uint32_t nz = (d + 0x7F7F7F7Fu) & 0x80808080u;                      // bit 7 of every non-zero byte
            const uint32_t ff = nz | (nz - (nz >> 7));                                // ... spread over the byte
            if (wz + z <= 66) Li[NB_MSKB + wz + z] = (int)(old[z] & ~ff);             // code 0 where a detection reaches
          }
        }
        WAVE_SYNC();
        if (nb_en) {   // trailing-edge ramp (:637-644; only the first branch is reachable).  An edge at i
                       // writes mask[i-7..i-1] only, which later iterations never read: read all, then write.
                       // codes of mask[124 + k0 .. 143 + k0] (five aligned words); entry 124 + k0 + p is byte p of them
          uint32_t evw[5];
#pragma unroll
          for (int z = 0; z < 5; ++z) evw[z] = (uint32_t)Li[NB_MSKB + 31 + 4 * s8 + z];
          WAVE_SYNC();   // (every lane's reads are issued before any lane's writes: one wave, LDS in order)
          // edge at i = 124 + k0 + p  <=>  code[p] == 1 and code[p - 1] == 0, p = 4..19: bit 7 of byte p of the words below (exact
          // bytewise tests: a byte is zero iff neither its low seven bits nor its top bit are set)
          uint32_t zb[5], ob[5];
#pragma unroll
          for (int z = 0; z < 5; ++z) {
            const uint32_t w = evw[z], x = w ^ 0x01010101u;
            zb[z] = ~(((w & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | w) & 0x80808080u;
            ob[z] = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;
          }
          uint32_t ed[4];
#pragma unroll
          for (int z = 1; z < 5; ++z) ed[z - 1] = ob[z] & __builtin_amdgcn_alignbit(zb[z], zb[z - 1], 24);   // zero flags moved up one byte
          if ((ed[0] | ed[1] | ed[2] | ed[3]) != 0u) {   // (rare: at most a few edges per channel and block)
            // {.933, .750, .500, .250, .067, 0, 0} (:608) = codes 2, 3, 4, 5, 6, 0, 0 out of two registers materialised HERE (as seven
            // literals the compiler parks them in VGPRs for the whole kernel)
            uint32_t lo4 = 0x05040302u, hi4 = 0x00000006u;
            asm volatile("" : "+v"(lo4), "+v"(hi4));
#pragma unroll
            for (int z = 0; z < 4; ++z) {
              uint32_t e = ed[z];
              while (e != 0u) {
                const int p = 4 * (z + 1) + (__builtin_ctz(e) >> 3);
                e &= e - 1u;
                uint8_t *w = mb + 124 + k0 + p - 7;
                w[0] = (uint8_t)lo4; w[1] = (uint8_t)(lo4 >> 8); w[2] = (uint8_t)(lo4 >> 16); w[3] = (uint8_t)(lo4 >> 24);
                w[4] = (uint8_t)hi4; w[5] = (uint8_t)(hi4 >> 8); w[6] = (uint8_t)(hi4 >> 16);
              }
            }
          }
        }
        WAVE_SYNC();
        {   // this lane's 16 output samples' codes (entries kA .. kA + 7 and 64 + kA ..), and mask[128..265(..271)] carried to the next call
          uint32_t own[4] = {0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u};
          if (nb_en) {
            const int2 a2 = *reinterpret_cast<const int2 *>(Li + NB_MSKB + 2 * s8), b2 = *reinterpret_cast<const int2 *>(Li + NB_MSKB + 16 + 2 * s8);
            own[0] = (uint32_t)a2.x; own[1] = (uint32_t)a2.y; own[2] = (uint32_t)b2.x; own[3] = (uint32_t)b2.y;
#pragma unroll
            for (int r = 0; r < 5; ++r) {
              const int e = 32 * r + 4 * s8;   // entries >= 138 (+ 2) are padding: store code 1
              mrow[8 * r] = (e < ASDR_NB_MASK_USED + 2) ? (uint32_t)Li[NB_MSKB + 32 + 8 * r + s8] : 0x01010101u;
            }
          }
          // output = mask x oldest block (:646-649); blanker-off channels of the wave pass their samples (code 1 = 1.0)
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const float mv = mask_decode_byte(mask_tab, group4, own[j >> 2], j & 3);   // (all 8 lanes of the group)
            vio[j] = mv * vio[j]; vqo[j] = mv * vqo[j];
          }
        }
      }   // nb_slow
      WAVE_SYNC();   // the B row / mask row is dead: the IF filter rows may overwrite it
      store8(L + W0 + kA, vio); store8(L + W0 + kA + 64, vio + 8); store8(L + W1 + kA, vqo); store8(L + W1 + kA + 64, vqo + 8);
    }
    WAVE_SYNC();
    TL(4);
    mw_straggler(1);
    if (DO1) { TAP_ROW(ASDR_TAP_NB_I, W0); TAP_ROW(ASDR_TAP_NB_Q, W1); }

    // The IF pipeline's state and coefficients must be IN registers before the ring prefetches below are issued: memory waits
    // count loads in order, and a wait for these values placed after the (conditional) prefetches would be a wait for everything
    // -- the first pipeline step would sit out the prefetches' HBM latency (it did: vmcnt(0) inside the pipeline loop).
    auto load_if_rows = [&]() {
      if (ROLE == 1 && ASDR_STREAM_R1_CARRY && blk > 0) {   // (a call's settings are fixed; the state is what the previous block's pipeline left -- it is stored every block as well)
        if_s4 = r1_if_s4;
#pragma unroll
        for (int z = 0; z < 5; ++z) if_cf[z] = r1_if_cf[z];
        return;
      }
      if_s4 = *reinterpret_cast<const float4 *>(&Sp->if_state[pl_iq][4 * pl_st]);
      const float *cf = &c_bq_pool[Ppl->if_table][5 * pl_st];
#pragma unroll
      for (int z = 0; z < 5; ++z) if_cf[z] = cf[z];
    };
    if (DO1 && ABL_ON(ABL_IF) && !(HAS_ALS && !ALS_FULL_OPT)) {   // (the ALS instantiations keep the old order: they spill with this one)
      if (!if_pre) load_if_rows();
      asm volatile("" : "+v"(if_s4.x), "+v"(if_s4.y), "+v"(if_s4.z), "+v"(if_s4.w), "+v"(if_cf[0]), "+v"(if_cf[1]), "+v"(if_cf[2]), "+v"(if_cf[3]), "+v"(if_cf[4]));
    }
    float *hi_ring = row_ptr(a.hil_i, (uint32_t)ch * 1024u + 4u * (uint32_t)kF);   // float rows: pieces kF + 32m
    float *hq_ring = row_ptr(a.hil_q, (uint32_t)ch * 1024u + 4u * (uint32_t)kF);
    float hq_o[16], hq_m[16], idl[16];   // this lane's pieces (samples kF + 32m + j at [4m + j]) of the Hilbert rings
    DEFINE_ALL_PATHS(hq_o, 16); DEFINE_ALL_PATHS(hq_m, 16); DEFINE_ALL_PATHS(idl, 16);
    // Hilbert ring (two previous blocks of mixed Q) is requested before the pipeline: 32 registers that the
    // pipeline and the mixer do not need, instead of two exposed HBM round trips after the mixer.  (Not in the SAM-only
    // instantiation: its only SSB channels are the padding slots, and it has no registers to spare at 3 waves/SIMD.)
    constexpr bool RING_PREFETCH = !HAS_SAM && !(HAS_ALS && !UNIFORM && ASDR_COMPACT_ROWS(STRIDE));   // (nor in the compact ALS rows' general form -- a few remainder waves: it spills with them)
    if (RING_PREFETCH && is_ssb) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
#if ASDR_NT_LOADS >= 3
        load4_nt(hq_ring + hs * 128 + 32 * m, hq_o + 4 * m);          // the oldest block: overwritten below, never read again
#else
        load4(hq_ring + hs * 128 + 32 * m, hq_o + 4 * m);
#endif
        load4(hq_ring + (hs ^ 1u) * 128 + 32 * m, hq_m + 4 * m);
      }
    }
    // the cached sin/cos pairs of this lane's pieces likewise (local-oscillator cache hit: they come from L2)
    float lo_c[16], lo_s[16];
    DEFINE_ALL_PATHS(lo_c, 16); DEFINE_ALL_PATHS(lo_s, 16);
    // The 16-waves-per-CU form: the cached pairs are the same for all 8 channels of the wave -- ONE 16-byte piece per lane (lane l: words
    // 4 l .. 4 l + 3 of the entry's 128 cos | 128 sin), requested here and parked, behind the IF pipeline, in the phase areas of channels
    // 0..3 (cos in 0 and 1, sin in 2 and 3: the layout the wave-uniform mixer stages its own pairs in; a cache hit leaves them unused).
    float lo4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (C16 && lo_hit) load4(lo_rd->c + 4 * lane_i, lo4);
    if (!C16 && RING_PREFETCH && lo_hit) {
      if (ROLE == 2) {   // written by another workgroup during this launch: `sc1` loads
        v4f c4[4], s4[4];
        if (R2_ONE_TRIP) {
#pragma unroll
          for (int m = 0; m < 4; ++m) { c4[m] = r2_c4[m]; s4[m] = r2_s4[m]; }
        } else { xch_load4x4(lo_rd->c + kF, c4[0], c4[1], c4[2], c4[3]); xch_load4x4(lo_rd->s + kF, s4[0], s4[1], s4[2], s4[3]); }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
#pragma unroll
          for (int j = 0; j < 4; ++j) { lo_c[4 * m + j] = c4[m][j]; lo_s[4 * m + j] = s4[m][j]; }
        }
      } else {
#pragma unroll
        for (int m = 0; m < 4; ++m) { load4(lo_rd->c + kF + 32 * m, lo_c + 4 * m); load4(lo_rd->s + kF + 32 * m, lo_s + 4 * m); }
      }
    }
    // ---- IF band-pass, AudioSDR.cpp:77-78: 2 x 4-stage cascade, 64 lanes = 8 ch x {I,Q} x 4 stages -------
    if (ROLE == 1 && sig_pending) { stream_signal(my_prog, sig_pending, lane); sig_pending = 0u; }   // the previous block's rows have left by now (they were stored a prologue and a scale ago)
    if constexpr (ROLE == 1 && ASDR_STREAM_PREFETCH_IN != 0) {
      if (DO1 && valid && blk + 1 < a.n_blocks) {   // the next block's input rows: they arrive during the pipeline
        const int4 *pi = reinterpret_cast<const int4 *>(a.in_i + io) + ASDR_N / 8, *pq = reinterpret_cast<const int4 *>(a.in_q + io) + ASDR_N / 8;
        pf_in[0] = load_int4_nt(pi); pf_in[1] = load_int4_nt(pi + 8); pf_in[2] = load_int4_nt(pq); pf_in[3] = load_int4_nt(pq + 8);
      }
    }
    if (DO1 && ABL_ON(ABL_IF)) {
      const int iq = pl_iq, st = pl_st;
      float sv[4];
      if (HAS_ALS && !ALS_FULL_OPT && !if_pre) load_if_rows();
      sv[0] = if_s4.x; sv[1] = if_s4.y; sv[2] = if_s4.z; sv[3] = if_s4.w;
      CHAIN_PRIO_ON();
      biquad_pipe<PIPE_PK>(Lp + (iq ? W1 : W0), true, st, if_cf, sv);
      CHAIN_PRIO_OFF();
      *reinterpret_cast<float4 *>(&Sp->if_state[iq][4 * st]) = make_float4(sv[0], sv[1], sv[2], sv[3]);
      if (ROLE == 1 && ASDR_STREAM_R1_CARRY) {
        r1_if_s4 = make_float4(sv[0], sv[1], sv[2], sv[3]);
#pragma unroll
        for (int z = 0; z < 5; ++z) r1_if_cf[z] = if_cf[z];
      }
    }
    if (C16 && lo_hit) store4(lds + (lane_i >> 4) * STRIDE + PH + ((4 * lane_i) & 63), lo4);   // word i of [cos | sin]: row i >> 6, place i & 63
    WAVE_SYNC();
    TL(5);
    mw_straggler(2);
    if (DO1) { TAP_ROW(ASDR_TAP_IF_I, W0); TAP_ROW(ASDR_TAP_IF_Q, W1); }
    // A role's own status bits go back with atomics: the three roles of a channel group update one word
    auto store_status_bits = [&](uint32_t mask) {
      if (lead && ((status ^ status_in) & mask) != 0u) { atomicAnd(&S->status, ~mask); atomicOr(&S->status, status & mask); }   // (only when a bit of its own changed)
    };
    if (ROLE == 1 || ROLE == 2) {   // streaming pipeline, boundary A: the IF output rows I (W0), Q (W1) cross through the exchange ring
      float *xa = a.xch_a + ((size_t)ch * ASDR_STREAM_DEPTH + (size_t)(blk % ASDR_STREAM_DEPTH)) * (2 * ASDR_N) + kF;
      if (ROLE == 1) {
        v4f vi[4], vq[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) { vi[m] = *reinterpret_cast<const v4f *>(L + W0 + kF + 32 * m); vq[m] = *reinterpret_cast<const v4f *>(L + W1 + kF + 32 * m); }
        xch_store4x4(xa, vi[0], vi[1], vi[2], vi[3]); xch_store4x4(xa + ASDR_N, vq[0], vq[1], vq[2], vq[3]);
        store_status_bits(ASDR_S_NB_DETECTED);
#ifndef ASDR_STREAM_LATE_SIGNAL
#define ASDR_STREAM_LATE_SIGNAL 1   /* role 1 publishes block k in front of block k + 1's IF pipeline (the stores' round trip used to sit at the end of every block: ~1.7 k cycles) */
#endif
        if (ASDR_STREAM_LATE_SIGNAL) sig_pending = (uint32_t)blk + 1u;   // (published below, in front of the next block's IF pipeline; in front of any wait; behind the last block)
        else stream_signal(a.stream_prog + wave_g, (uint32_t)blk + 1u, lane);
        continue;
      } else {
        v4f vi[4], vq[4];
        if (R2_ONE_TRIP) {
#pragma unroll
          for (int m = 0; m < 4; ++m) { vi[m] = r2_vi[m]; vq[m] = r2_vq[m]; }
          if (sig_pending) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (this wave's stores of the previous block have left: they were issued in front of the loads above, which have returned)
        } else { xch_load4x4(xa, vi[0], vi[1], vi[2], vi[3]); xch_load4x4(xa + ASDR_N, vq[0], vq[1], vq[2], vq[3]); }
        if (sig_pending) { if (lane == 0) __hip_atomic_store(my_prog, sig_pending, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); sig_pending = 0u; }   // (vmcnt(0) inside the loads)
#pragma unroll
        for (int m = 0; m < 4; ++m) { *reinterpret_cast<v4f *>(L + W0 + kF + 32 * m) = vi[m]; *reinterpret_cast<v4f *>(L + W1 + kF + 32 * m) = vq[m]; }
        WAVE_SYNC();
        if constexpr (R2_PREFETCH) {   // how far the two producers are: read in front of the Hilbert FIR (the prefetch of the next block's rows)
          r2_look_in = __hip_atomic_load(a.stream_prog + wave_g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          r2_look_lo = __hip_atomic_load(a.stream_prog + 3 * a.stream_waves, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }

    // SAM as three launches: this wave's tile of the exchange buffer, [sample][I, Q][slot in tile] (8 KB, written / read whole)
    // The tile moves as 512 sixteen-byte pieces, lane l on pieces l, l + 64, ..: piece p = (sample p / 4, I or Q = (p / 2) % 2, slots
    // 4 (p % 2) .. + 3) -- one wave instruction moves 1 KB contiguous; the four values of a piece come from four channels' LDS rows.
    // (the looped roles: tile set (a.sam_set + blk) % a.sam_sets; the others: a.sam_set_stride == 0, one set)
    const uint32_t sam_set_k = (LOOPED_ROLE && a.sam_sets > 1u) ? (a.sam_set + (uint32_t)blk) % a.sam_sets : 0u;
    float *const sam_tile = (ROLE >= 4) ? a.xch_sam + (size_t)sam_set_k * a.sam_set_stride + (size_t)wave_g * (2 * ASDR_N * 8) : nullptr;
    float *const sam_lds = lds + (4 * (lane_i & 1)) * STRIDE + (((lane_i >> 1) & 1) ? W1 : W0) + (lane_i >> 2);
    if (PRE_ROLE) {   // the IF output leaves for the PLL kernel; the rest of the chain is the post kernel's
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float *src = sam_lds + 16 * k;
        *reinterpret_cast<float4 *>(sam_tile + 4 * (lane_i + 64 * k)) = make_float4(src[0], src[STRIDE], src[2 * STRIDE], src[3 * STRIDE]);
      }
      store_status_bits(ASDR_S_NB_DETECTED);   // (this role's bit only, atomically: the post role of the PREVIOUS block may be running beside it)
      continue;
    }
    // ---- SAM: quadrature PLL, AudioSDR.cpp:688-749 (sequential per channel) ---------------------------------
    bool pll_locked = false;
    // One channel's 128 PLL steps on ONE lane: Lc = the channel's LDS rows, Sc = its state row.  Returns the lock flag.
    auto pll_run = [&](float *Lc, ChanSmall *Sc) -> bool {
      return pll_loop<HAS_ALS, 4>(Sc, K, sine, two_pi,
        [&](int i, float *xr, float *xi) {
          const float4 r4 = *reinterpret_cast<const float4 *>(Lc + W0 + i), i4 = *reinterpret_cast<const float4 *>(Lc + W1 + i);
          xr[0] = r4.x; xr[1] = r4.y; xr[2] = r4.z; xr[3] = r4.w; xi[0] = i4.x; xi[1] = i4.y; xi[2] = i4.z; xi[3] = i4.w; },
        [&](int i, const float *xr, const float *xi) {
          *reinterpret_cast<float4 *>(Lc + W0 + i) = make_float4(xr[0], xr[1], xr[2], xr[3]);
          *reinterpret_cast<float4 *>(Lc + W1 + i) = make_float4(xi[0], xi[1], xi[2], xi[3]); });
    };
    if constexpr (HAS_SAM && WAVES > 1) {
      // Every wave of the workgroup has its IF output in LDS; wave 0 runs the PLL of all 8 * WAVES channels, one per lane.
      __syncthreads();
      if (ABL_ON(ABL_SAM) && wave == 0 && lane_i < 8 * WAVES) {
        const int sj = (int)blockIdx.x * WAVES * 8 + lane_i;       // schedule slot of workgroup channel lane_i
        int chj = a.n_channels; uint32_t modej = a.params[a.n_channels].mode;
        if (sj < a.n_sched) { const int4 sl = *reinterpret_cast<const int4 *>(a.sched + sj); chj = sl.x; modej = (uint32_t)sl.y; }
        if (modej == ASDR_SAMmode) {
          float *Lc = lds_wg + lane_i * STRIDE;
          const bool lk = pll_run(Lc, row_ptr(a.small, (uint32_t)chj * (uint32_t)sizeof(ChanSmall)));
          reinterpret_cast<int *>(Lc)[SCR0] = lk ? 1 : 0;
        }
      }
      __syncthreads();
      if (is_sam) {
        pll_locked = Li[SCR0] != 0;
        if (lead) status = (status & ~ASDR_S_PLL_LOCKED) | (pll_locked ? ASDR_S_PLL_LOCKED : 0u);
      }
    } else if (HAS_SAM && ABL_ON(ABL_SAM) && __any(is_sam)) {
      if (is_sam && lead) {
        const bool lk = pll_run(L, S);
        status = (status & ~ASDR_S_PLL_LOCKED) | (lk ? ASDR_S_PLL_LOCKED : 0u);
        Li[SCR0] = lk ? 1 : 0;
      }
      WAVE_SYNC();
      if (is_sam) pll_locked = Li[SCR0] != 0;
    }
    if (POST_ROLE) {   // the rows as the PLL kernel left them (rotated where it was locked), and its lock flag
      float4 pc[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) pc[k] = *reinterpret_cast<const float4 *>(sam_tile + 4 * (lane_i + 64 * k));
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float *dst = sam_lds + 16 * k;
        dst[0] = pc[k].x; dst[STRIDE] = pc[k].y; dst[2 * STRIDE] = pc[k].z; dst[3 * STRIDE] = pc[k].w;
      }
      if (is_sam) pll_locked = a.sam_lock[(size_t)sam_set_k * a.sam_lock_stride + wave_g * 8 + c8] != 0u;   // (as the PLL kernel left it for THIS block: see UpdateArgs.sam_lock)
      WAVE_SYNC();
    }
    // envelope detector runs for AM, and for SAM when the PLL is unlocked at the end of the block (:132)
    const bool do_env = DO2 && (is_am || (is_sam && !pll_locked));
    const bool do_mix = is_ssb || do_env;

    // ---- mixer phase sequence, AudioSDR.h:508-526 (phase accumulates sequentially in float) -----------------
    // Normally the sequence is already in the PH row (it rode in the blanker's sequential loop).  It is computed here for SAM
    // channels that fall back to the envelope detector (their mixer runs only if the PLL ends the block unlocked, :132), when
    // no channel of the wave runs the blanker, and after the blanker's general path (whose scratch overlays the PH row).
    if (do_mix && !mix_early) { mphase = S->phase_am; minc = -K.if_center * K.phase_inc_unit; }
    const bool ph_seq = do_mix && !ph_ready && !lo_hit;
    if (__any(ph_seq)) {
      if (ph_seq && lead) {
        float phase = mphase;
        const bool up = !(minc < 0.0f);   // one wrap test per sample, see the blanker's sequential loop
        const float wrapv = up ? -two_pi : two_pi, lim = up ? two_pi : 0.0f;
        const uint32_t flip = up ? 0u : 0x80000000u;
#pragma unroll 1
        for (int i = 0; i < ASDR_N; i += 8) {
          float pv[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            pv[u] = phase;
            const float t = phase + minc, tw = t + wrapv;
            phase = (__uint_as_float(__float_as_uint(t) ^ flip) > lim) ? tw : t;
          }
          if constexpr (C16) store4e(L + PH + (i >> 1), pv); else store8(L + PH + i, pv);
        }
        if (is_ssb) S->phase_ssb = phase; else S->phase_am = phase;
        mphase_end = phase;
      }
      WAVE_SYNC();
    }
    if (ROLE == 2) {   // next block's phase = this block's end phase: from the cache entry, or from the channel's lead lane
      const float endp = __int_as_float(__builtin_amdgcn_ds_bpermute((lane_i & ~7) << 2, __float_as_int(mphase_end)));
      carry_phase = lo_hit ? lo_end : endp;
      carry_fsh = fsh_raw;
    }
    // Wave-uniform mixer: all 8 lanes of a channel hold the channel's carried phase and increment, so the wave can test whether
    // EVERY channel mixes with the same pair -- true for receivers that were configured together, whose phases then stay
    // identical for ever.  Then the 128 sin/cos pairs are evaluated once per wave (2 samples per lane, from channel 0's phase
    // row, staged in the PH rows of channels 1 and 2) instead of 16 per lane.
    const uint32_t ph_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(mphase));
    const uint32_t inc_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(minc));
    const bool mix_uni = __all(do_mix && __float_as_uint(mphase) == ph_first && __float_as_uint(minc) == inc_first);
    TL(6);
    // One step of the phase recurrence (AudioSDR.h:513-518), exactly as the sequential loops form it: the 16-waves-per-CU form keeps the
    // even samples' phases and takes an odd one from its even neighbour.
    auto phase_step = [&](float p, float inc) -> float {
      const bool up = !(inc < 0.0f);
      const float wrapv = up ? -two_pi : two_pi, lim = up ? two_pi : 0.0f;
      const uint32_t flip = up ? 0u : 0x80000000u;
      const float t = p + inc, tw = t + wrapv;
      return (__uint_as_float(__float_as_uint(t) ^ flip) > lim) ? tw : t;
    };
    if (ABL_ON(ABL_MIX) && mix_uni && !lo_hit) {   // sin/cos of the wave's 128 phases, two per lane
      float ph2[2];
      if constexpr (C16) { ph2[0] = lds[PH + lane_i]; ph2[1] = phase_step(ph2[0], minc); WAVE_SYNC(); }   // (every lane has read channel 0's phases: the pairs below overwrite them)
      else { const float2 p2 = *reinterpret_cast<const float2 *>(lds + PH + 2 * lane); ph2[0] = p2.x; ph2[1] = p2.y; }
      float c2[2], s2[2];
      sincos_batch<2, SINE_LDS>(sine, ph2, c2, s2, two_pi, SinIndexK{K.inv_two_pi_d, K.sin_index_scale_d}, K.half_pi_d);
      if constexpr (C16) {   // 128 cos in the 64-float phase areas of channels 0 and 1, 128 sin in those of channels 2 and 3
        *reinterpret_cast<float2 *>(lds + (lane_i >> 5) * STRIDE + PH + ((2 * lane_i) & 63)) = make_float2(c2[0], c2[1]);
        *reinterpret_cast<float2 *>(lds + (2 + (lane_i >> 5)) * STRIDE + PH + ((2 * lane_i) & 63)) = make_float2(s2[0], s2[1]);
      } else {
      *reinterpret_cast<float2 *>(lds + STRIDE + PH + 2 * lane) = make_float2(c2[0], c2[1]);
      *reinterpret_cast<float2 *>(lds + 2 * STRIDE + PH + 2 * lane) = make_float2(s2[0], s2[1]);
      }
      WAVE_SYNC();
    }
    // complex multiply by e^{j phase}, in place on this lane's own samples; the mixed I also goes to its 2-slot ring
    // in HBM (slot hs = this block, hs^1 = previous block = the reference's exact 128-sample delay, :111)
    // Uniform SSB waves (round 5): the mixed Q stays in registers for the Hilbert stage (same lane, same pieces) and neither mixed row
    // goes back to LDS -- the delayed I comes from the HBM ring, nothing reads W0 / W1 again before the history overlays them.
    constexpr bool QN_DIRECT_K = UNIFORM && ABL_ON(ABL_MIX) && (ASDR_MIX_KEEP_Q != 0);
    const bool qn_direct = QN_DIRECT_K && is_ssb;
    float qn[16];   // own pieces (samples kF + 32m + j at [4m + j]) of the mixed Q
    DEFINE_ALL_PATHS(qn, 16);
    if constexpr (C16) {
      // The 16-waves-per-CU form: TWO copies of the multiply, chosen once per wave -- pairs staged in the phase areas (cache hit or
      // wave-uniform phases) | per-channel lookups.  The second is the register-hungry one (binary64 table phases, gathers) and runs
      // WITHOUT the Hilbert ring's 32 prefetched registers: they are given up in front of it and asked for again behind it.
      if (ABL_ON(ABL_MIX) && do_mix) {
        // (written out twice by a macro: as a generic lambda taking the staged / per-channel choice, the mixed-Q array it fills went to scratch)
#define C16_MIX_LOOP(STAGED)                                                                                                                  \
          _Pragma("unroll")                                                                                                                   \
          for (int m = 0; m < 4; ++m) {                                                                                                       \
            float cc[4], sn[4], vi[4], vq[4], mi[4], mq[4];                                                                                   \
            load4(L + W0 + kF + 32 * m, vi); load4(L + W1 + kF + 32 * m, vq);                                                                 \
            if (STAGED) {                                                                                                                     \
              load4(lds + (m >> 1) * STRIDE + PH + 32 * (m & 1) + kF, cc); load4(lds + (2 + (m >> 1)) * STRIDE + PH + 32 * (m & 1) + kF, sn); \
            } else {                                                                                                                          \
              float ph[4];                                                                                                                    \
              const float2 e2 = *reinterpret_cast<const float2 *>(L + PH + ((kF + 32 * m) >> 1));                                             \
              ph[0] = e2.x; ph[1] = phase_step(e2.x, minc); ph[2] = e2.y; ph[3] = phase_step(e2.y, minc);                                     \
              sincos_batch<4, SINE_LDS>(sine, ph, cc, sn, two_pi, SinIndexK{K.inv_two_pi_d, K.sin_index_scale_d}, K.half_pi_d);                         \
            }                                                                                                                                 \
            _Pragma("unroll")                                                                                                                 \
            for (int j = 0; j < 4; ++j) {                                                                                                     \
              mi[j] = vi[j] * cc[j] - vq[j] * sn[j];                                                                                          \
              mq[j] = vq[j] * cc[j] + vi[j] * sn[j];                                                                                          \
            }                                                                                                                                 \
            if (qn_direct) {                                                                                                                  \
              _Pragma("unroll")                                                                                                               \
              for (int j = 0; j < 4; ++j) qn[4 * m + j] = mq[j];                                                                              \
            } else { store4(L + W0 + kF + 32 * m, mi); store4(L + W1 + kF + 32 * m, mq); }                                                    \
            store4_nt(hi_ring + hs * 128 + 32 * m, mi);                                                                                       \
          }
        if (lo_hit || mix_uni) { C16_MIX_LOOP(true) }
        else {
          DEFINE_ALL_PATHS(hq_o, 16); DEFINE_ALL_PATHS(hq_m, 16);   // (given up: nothing of the prefetch is live across the lookups)
          C16_MIX_LOOP(false)
#pragma unroll
          for (int m = 0; m < 4; ++m) { load4(hq_ring + hs * 128 + 32 * m, hq_o + 4 * m); load4(hq_ring + (hs ^ 1u) * 128 + 32 * m, hq_m + 4 * m); }
        }
      }
    } else
    if (ABL_ON(ABL_MIX) && do_mix) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        float cc[4], sn[4], vi[4], vq[4], mi[4], mq[4];
        load4(L + W0 + kF + 32 * m, vi); load4(L + W1 + kF + 32 * m, vq);
        if (lo_hit) {
          if (RING_PREFETCH) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { cc[j] = lo_c[4 * m + j]; sn[j] = lo_s[4 * m + j]; }
          } else { load4(lo_rd->c + kF + 32 * m, cc); load4(lo_rd->s + kF + 32 * m, sn); }
        } else if (mix_uni) {
          load4(lds + STRIDE + PH + kF + 32 * m, cc); load4(lds + 2 * STRIDE + PH + kF + 32 * m, sn);
        } else {
          // per-channel phases (the wave's channels carry different mixer phases): 4 x (cos, sin) lookups per piece, gathers batched
          float ph[4];
          load4(L + PH + kF + 32 * m, ph);
#ifndef ASDR_MIX_SERIAL_LOOKUPS
          sincos_batch<4, SINE_LDS_PC>(SINE_LDS_PC ? sine_tab : nullptr, ph, cc, sn, two_pi, SinIndexK{K.inv_two_pi_d, K.sin_index_scale_d}, K.half_pi_d);
#else
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            cc[j] = cos_f32(sine, ph[j], two_pi, SinIndexK{K.inv_two_pi_d, K.sin_index_scale_d}, K.half_pi_d); sn[j] = sin_f32(sine, ph[j], two_pi, SinIndexK{K.inv_two_pi_d, K.sin_index_scale_d});
            if ((j & 1) == 1) SCHED_FENCE();
          }
#endif
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          mi[j] = vi[j] * cc[j] - vq[j] * sn[j];
          mq[j] = vq[j] * cc[j] + vi[j] * sn[j];
        }
        if (qn_direct) {
#pragma unroll
          for (int j = 0; j < 4; ++j) qn[4 * m + j] = mq[j];
        } else { store4(L + W0 + kF + 32 * m, mi); store4(L + W1 + kF + 32 * m, mq); }
#ifndef ASDR_TEMPORAL_RINGS
        if (is_ssb) store4_nt(hi_ring + hs * 128 + 32 * m, mi);
#else
        if (is_ssb) store4(hi_ring + hs * 128 + 32 * m, mi);
#endif
      }
    }

    // Requested here, consumed after the Hilbert FIR (their latency hides behind it): the audio filter's state and
    // coefficient row, the AGC scalars and the index of the AGC table.
    float4 af_s4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float af_cf[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    float agc_gain_in = 0.f, agc_old0 = 0.f, agc_carrier0 = 0.f;
    uint32_t agc_hc0 = 0u;
    int agc_tab_idx = 0;
    // MW: the two audio-duty waves (rel 1, 2) run the cascades of 16 channels each, lane = (channel of the half) x 4 + stage
    const bool mw_audio_duty = (MW_SHARE & 1) && (mw_rel == 1 || mw_rel == 2);
    const int mw_casc = PLB ? (4 * (lane_i >> 4) + (lane_i & 3)) : (lane_i >> 2);   // which of the duty wave's 16 cascades this lane works on (stage pl_st)
    ChanSmall *mw_af_S = S;
    auto load_af_agc_state = [&]() {
    ChanSmall *const Saf = PLB ? Spa : S; const ChanParams *const Paf = PLB ? Ppa : Pp;   // (formed by all lanes, see the AM image filter)
    if (ABL_ON(ABL_AF) && (((MW_SHARE & 1) != 0) ? af_en : af_en_pa)) {
      if constexpr ((MW_SHARE & 1) != 0) {
        if (mw_audio_duty) {
          const int chq = mw_channel(16 * (mw_rel - 1) + mw_casc);
          mw_af_S = row_ptr(a.small, (uint32_t)chq * (uint32_t)sizeof(ChanSmall));
          af_s4 = *reinterpret_cast<const float4 *>(&mw_af_S->af_state[4 * pl_st]);
          const float *cf = &c_bq_pool[row_ptr(a.params, (uint32_t)chq * (uint32_t)sizeof(ChanParams))->audio_table][5 * pl_st];
#pragma unroll
          for (int z = 0; z < 5; ++z) af_cf[z] = cf[z];
        }
      } else {
      af_s4 = *reinterpret_cast<const float4 *>(&Saf->af_state[4 * pl_st]);
      const float *cf = &c_bq_pool[Paf->audio_table][5 * pl_st];
#pragma unroll
      for (int z = 0; z < 5; ++z) af_cf[z] = cf[z];
      }
    }
    if (ABL_ON(ABL_AGC) && agc_en) {
      agc_gain_in = S->agc_gain; agc_old0 = S->agc_old_abs; agc_hc0 = S->agc_hang_counter; agc_carrier0 = S->am_carrier;
      agc_tab_idx = P.agc_table;
    }
    };
    if constexpr (!C16) load_af_agc_state();   // (the 16-waves-per-CU form asks behind the FIR: its registers are the FIR's)
    TL(7);
    // ---- SSB/CW/WSPR: 257-tap folded Hilbert on Q, I delayed 128, AudioSDR.cpp:89-118 ----------------------
    if (__any(is_ssb)) {
      if (is_ssb && !qn_direct) {
#pragma unroll
        for (int m = 0; m < 4; ++m) load4(L + W1 + kF + 32 * m, qn + 4 * m);   // mixed Q, written above by other lanes too (same wave: LDS is in order)
      }
      WAVE_SYNC();   // every lane has consumed W0/W1/PH: the history may now overlay them
      if constexpr (C16) {
        // The 16-waves-per-CU form: 320-float rows cannot hold the 383-sample history at once.  Output i reads x[i + 1 .. i + 255]: outputs
        // 0..63 need x[1..318], outputs 64..127 need x[65..382] -- the same 318-float window 64 samples on.  Pass 1 on x[1..318] (word m - 1
        // as ever), the rows move down by 64 words, the block's last 64 samples fill the top, pass 2 runs the SAME code on the same words.
        // Every lane computes 8 outputs per pass (4 pairs: 11-pair register windows instead of 15).
        // (the ring's two blocks were requested in front of the IF pipeline, like the other forms: 32 registers the pipeline has to spare)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int k = kF + 32 * m + j;
            if (k >= 1) L[XP + k - 1] = hq_o[4 * m + j];
            L[XP + 127 + k] = hq_m[4 * m + j];
            if (k < 63) L[XP + 255 + k] = qn[4 * m + j];   // x[256 + k], k < 63: what pass 1 reads of this block
          }
          store4_nt(hq_ring + hs * 128 + 32 * m, qn + 4 * m);   // newest replaces oldest in the HBM ring
        }
        if (lead) S->hil_slot = hs ^ 1u;
        WAVE_SYNC();
        TL(8);
        v2f fa[4], fb[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { fa[e] = (v2f){0.0f, 0.0f}; fb[e] = (v2f){0.0f, 0.0f}; }
        if (ABL_ON(ABL_HIL)) hilbert_fir<0, 4, false>(L, 4 * s8, fa);   // outputs 8 s8 + 2e, + 1  (taps as scalar operands: this form has no register for them)
        WAVE_SYNC();                                               // pass 1 has read everything it needs
        {
          float t[32];
#pragma unroll
          for (int q = 0; q < 8; ++q) load4(L + XP + 64 + 32 * s8 + 4 * q, t + 4 * q);
          WAVE_SYNC();                                             // (every lane's reads before any lane's writes)
#pragma unroll
          for (int q = 0; q < 8; ++q) store4(L + XP + 32 * s8 + 4 * q, t + 4 * q);   // word j <- word j + 64: x[65 + j] at j
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
#pragma unroll
          for (int j = 0; j < 4; ++j) { const int k = kF + 32 * m + j; if (k >= 63 && k < 127) L[XP + 191 + k] = qn[4 * m + j]; }   // x[256 + k] at word 256 + k - 65
        }
        WAVE_SYNC();
        if (ABL_ON(ABL_HIL)) hilbert_fir<0, 4, false>(L, 4 * s8, fb);   // the same window 64 samples on: outputs 64 + 8 s8 + 2e, + 1
        WAVE_SYNC();                                               // all history reads done: W1 may overwrite it
        TL(9);
        load_af_agc_state();
#pragma unroll
        for (int m = 0; m < 4; ++m) load4(hi_ring + (hs ^ 1u) * 128 + 32 * m, idl + 4 * m);   // delayed I = previous block's mixed I (:111)
        *reinterpret_cast<float4 *>(L + W1 + 8 * s8) = make_float4(fa[0][0], fa[0][1], fa[1][0], fa[1][1]);
        *reinterpret_cast<float4 *>(L + W1 + 8 * s8 + 4) = make_float4(fa[2][0], fa[2][1], fa[3][0], fa[3][1]);
        *reinterpret_cast<float4 *>(L + W1 + 64 + 8 * s8) = make_float4(fb[0][0], fb[0][1], fb[1][0], fb[1][1]);
        *reinterpret_cast<float4 *>(L + W1 + 64 + 8 * s8 + 4) = make_float4(fb[2][0], fb[2][1], fb[3][0], fb[3][1]);
        WAVE_SYNC();
      } else {
      if (is_ssb) {
        // history sample x[m'] with m' = B + k (B = 0: two blocks back = ring slot hs, 128: previous = slot hs^1,
        // 256: this block) is stored at L[XP + m' - 1] (x[0] is never used): natural order shifted by one float, so
        // that every operand pair (x[odd], x[odd+1]) of the FIR is an 8-byte-aligned LDS pair.
        if (!RING_PREFETCH) {
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            load4(hq_ring + hs * 128 + 32 * m, hq_o + 4 * m);
            load4(hq_ring + (hs ^ 1u) * 128 + 32 * m, hq_m + 4 * m);
          }
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int k = kF + 32 * m + j;
            if (k >= 1) L[XP + k - 1] = hq_o[4 * m + j];
            L[XP + 127 + k] = hq_m[4 * m + j];
          }
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
#pragma unroll
          for (int j = 0; j < 4; ++j) L[XP + 255 + kF + 32 * m + j] = qn[4 * m + j];
#ifndef ASDR_TEMPORAL_RINGS
          store4_nt(hq_ring + hs * 128 + 32 * m, qn + 4 * m);
#else
          store4(hq_ring + hs * 128 + 32 * m, qn + 4 * m);   // newest replaces oldest (loaded before the IF pipeline)
#endif
        }
        if (lead) S->hil_slot = hs ^ 1u;
        carry_hs = hs ^ 1u;
      }
      WAVE_SYNC();
      TL(8);
      mw_straggler(3);
      if constexpr (R2_PREFETCH) {
        // The NEXT block's oscillator entry and IF rows, if both producers are known to have published it (the counters as last polled: a producer that is
        // ahead is polled once for several blocks): the round trip of ~1.9 k cycles runs under the FIR.  Ring slots (blk + 1) % depth are this wave's to read
        // until it publishes block blk + 1 itself.
        const uint32_t b2 = (uint32_t)blk + 2u;
        if (blk + 1 < a.n_blocks && (seen_in < b2 || seen_lo < b2)) {   // one look at the two counters, asked for behind this block's rows (no waiting: a producer that is not there yet is waited for at the top of the next block)
          const uint32_t ui = (uint32_t)__builtin_amdgcn_readfirstlane((int)r2_look_in), ul = (uint32_t)__builtin_amdgcn_readfirstlane((int)r2_look_lo);
          asm volatile("" ::: "memory");   // (no load of the published rows moves in front of the look)
          if (ui != ASDR_STREAM_FAIL && ui > seen_in) seen_in = ui;
          if (ul != ASDR_STREAM_FAIL && ul > seen_lo) seen_lo = ul;
        }
        if (blk + 1 < a.n_blocks && seen_in != ASDR_STREAM_FAIL && seen_lo != ASDR_STREAM_FAIL && seen_in >= b2 && seen_lo >= b2) {
          const LoEntry *ln = a.lo_ring + ((blk + 1) % ASDR_LO_RING);
          const float *lp = ln->c + kF, *xp = a.xch_a + ((size_t)ch * ASDR_STREAM_DEPTH + (size_t)((blk + 1) % ASDR_STREAM_DEPTH)) * (2 * ASDR_N) + kF;
          r2n_key = ld_sc1_v4(reinterpret_cast<const float *>(ln));
#pragma unroll
          for (int m = 0; m < 4; ++m) { r2n_c4[m] = ld_sc1_v4(lp + 32 * m); r2n_s4[m] = ld_sc1_v4(lp + ASDR_N + 32 * m); r2n_vi[m] = ld_sc1_v4(xp + 32 * m); r2n_vq[m] = ld_sc1_v4(xp + ASDR_N + 32 * m); }
          r2_pref = true;
        }
      }
      constexpr bool IDL_EARLY = (ROLE == 2);   // the pipeline's role 2 has the registers to request the delayed I before the FIR
      if (IDL_EARLY && is_ssb) {
#pragma unroll
        for (int m = 0; m < 4; ++m) load4(hi_ring + (hs ^ 1u) * 128 + 32 * m, idl + 4 * m);
      }
      // In the pipeline's role 2 the FIR is shared with the workgroup's other waves (asdr_stream_fir_helper): this wave computes the
      // output pairs 0..3 of every lane and ONE helper 4..7 (asdr_stream_kernel), or 0, 1 and THREE helpers two pairs each
      // (asdr_stream_kernel_h3), between two workgroup barriers.
      constexpr int FIR_NE = (ROLE == 2) ? 8 / (FIR_HELPERS + 1) : 8;   // (one helper: pairs 0..3 here; three helpers: pairs 0, 1)
      v2f acc2[FIR_NE];
#pragma unroll
      for (int e = 0; e < FIR_NE; ++e) acc2[e] = (v2f){0.0f, 0.0f};
      if (ROLE == 2) __syncthreads();   // the history is staged: the helper may read it
      if (ABL_ON(ABL_HIL) && is_ssb) hilbert_fir<0, FIR_NE>(L, k0 >> 1, acc2);
      if (ROLE == 2) __syncthreads(); else WAVE_SYNC();   // all history reads done: rows W0/W1 may overwrite the start of the history
      TL(9);
      if (!IDL_EARLY && is_ssb) {   // delayed I = previous block's mixed I (:111); requested here: the FIR has no registers to spare for it
#pragma unroll
        for (int m = 0; m < 4; ++m) {
#if ASDR_NT_LOADS >= 4
          load4_nt(hi_ring + (hs ^ 1u) * 128 + 32 * m, idl + 4 * m);
#else
          load4(hi_ring + (hs ^ 1u) * 128 + 32 * m, idl + 4 * m);
#endif
        }
      }
      if (is_ssb) {   // the FIR owns 16 contiguous outputs per lane; the combine works on the float-row pieces: hand over through W1
#pragma unroll
        for (int e = 0; e < FIR_NE; e += 2) *reinterpret_cast<float4 *>(L + W1 + k0 + 2 * e) = make_float4(acc2[e][0], acc2[e][1], acc2[e + 1][0], acc2[e + 1][1]);
      }
      if (ROLE == 2) __syncthreads(); else WAVE_SYNC();   // (the helper's half is in W1 as well)
      }   // !C16
      if (is_ssb) {   // sideband combine (:115-118) with the delayed I
        // x - y == x + (-y) exactly: the sideband is a sign bit, not a select between two forms (a wave-uniform
        // `sub_q ? a - b : a + b` per sample compiles to a scalar branch cascade per sample in the uniform-key instantiations)
        const uint32_t sgn = sub_q ? 0x80000000u : 0u;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          float qh[4], au[4];
          load4(L + W1 + kF + 32 * m, qh);
#pragma unroll
          for (int j = 0; j < 4; ++j) au[j] = idl[4 * m + j] + __uint_as_float(__float_as_uint(qh[j]) ^ sgn);
          store4(L + W0 + kF + 32 * m, au);
          TAP4(ASDR_TAP_MIX_I, m, idl + 4 * m); TAP4(ASDR_TAP_MIX_Q, m, qh);
        }
      }
    }

    // ---- AM envelope path, AudioSDR.cpp:132-143 (rows W0/W1 hold the mixed I/Q already) ----------------------
    if (ABL_ON(ABL_ENV) && __any(do_env)) {
      WAVE_SYNC();
      {
        const int iq = pl_iq, st = pl_st;
        const bool do_env_p = PLB ? (__builtin_amdgcn_ds_bpermute(pl_c << 5, do_env ? 1 : 0) != 0) : do_env;   // (of the channel this lane's cascade belongs to)
        float sv[4];
        ChanSmall *const Simg = Sp;   // (formed by ALL lanes: the channel index comes through ds_bpermute, which reads 0 from a lane that is switched off)
        const float4 s4 = *reinterpret_cast<const float4 *>(&Simg->img_state[iq][4 * st]);
        sv[0] = s4.x; sv[1] = s4.y; sv[2] = s4.z; sv[3] = s4.w;
        biquad_pipe<PIPE_PK>(Lp + (iq ? W1 : W0), do_env_p, st, &c_bq_pool[ASDR_TBL_AM_IMAGE][5 * st], sv);
        if (do_env_p) *reinterpret_cast<float4 *>(&Simg->img_state[iq][4 * st]) = make_float4(sv[0], sv[1], sv[2], sv[3]);
      }
      WAVE_SYNC();
      if (do_env) {
#pragma unroll 1
        for (int h = 0; h < 2; ++h) {
          float vi[8], vq[8], au[8];
          load8(L + W0 + kA + 64 * h, vi); load8(L + W1 + kA + 64 * h, vq);
          TAP8(ASDR_TAP_MIX_I, h, vi); TAP8(ASDR_TAP_MIX_Q, h, vq);
#pragma unroll
          for (int j = 0; j < 8; ++j) { au[j] = sqrtf(vi[j] * vi[j] + vq[j] * vq[j]); if ((j & 3) == 3) SCHED_FENCE(); }
          store8(L + W0 + kA + 64 * h, au);
        }
      }
      WAVE_SYNC();
      if (do_env && lead) {   // carrier level tracker in binary64, stored float each sample (:141)
        float lvl = S->am_carrier;
#pragma unroll 1
        for (int i = 0; i < ASDR_N; i += 4) {
          const float4 x4 = *reinterpret_cast<const float4 *>(L + W0 + i);
          lvl = (float)(.995 * (double)lvl + 0.005 * (double)fabsf(x4.x));
          lvl = (float)(.995 * (double)lvl + 0.005 * (double)fabsf(x4.y));
          lvl = (float)(.995 * (double)lvl + 0.005 * (double)fabsf(x4.z));
          lvl = (float)(.995 * (double)lvl + 0.005 * (double)fabsf(x4.w));
        }
        S->am_carrier = lvl; carrier_now = lvl; carrier_fresh = true;
        if (ROLE == 2) {   // the AGC of this block runs in role 3 (:407-409 take twice THIS block's level): crosses beside the audio row
          float *xc = a.xch_c + (size_t)ch * ASDR_STREAM_DEPTH + (size_t)(blk % ASDR_STREAM_DEPTH);
          asm volatile("global_store_dword %0, %1, off sc1\n\ts_nop 0" :: "v"(xc), "v"(lvl) : "memory");
        }
      }
    }
    if (is_sam && pll_locked) {   // audio = rotated Q (:126-128)
#pragma unroll 1
      for (int h = 0; h < 2; ++h) {
        float vi[8], vq[8];
        load8(L + W0 + kA + 64 * h, vi); load8(L + W1 + kA + 64 * h, vq);
        TAP8(ASDR_TAP_MIX_I, h, vi); TAP8(ASDR_TAP_MIX_Q, h, vq);
        store8(L + W0 + kA + 64 * h, vq);
      }
    }
    if ((ROLE == 0 || ROLE == 6 || ROLE == 7) && !is_ssb && !is_am && mode != ASDR_SAMmode) {
      // Unknown mode value: neither demodulator branch runs (AudioSDR.cpp:84, 122), _audioOut still holds what the PREVIOUS block
      // left in it -- its audio after the audio filter, AGC and ALS -- and those stages now process it again (:149-161).  The row
      // comes back from HBM (every block stores it, below).  Without the row (asdr_set_exact_unknown_mode(b, 0)): silence.
      float z[16];
      DEFINE_ALL_PATHS(z, 16);
      if (a.audio_prev != nullptr) {
        const float *ap = row_ptr(a.audio_prev, (uint32_t)ch * 512u + 4u * (uint32_t)kF);
#pragma unroll
        for (int m = 0; m < 4; ++m) load4(ap + 32 * m, z + 4 * m);
      }
#pragma unroll
      for (int m = 0; m < 4; ++m) store4(L + W0 + kF + 32 * m, z + 4 * m);
    }
    WAVE_SYNC();
    TL(10);
    TAP_ROW(ASDR_TAP_DEMOD, W0);
    if (ROLE == 2 || ROLE == 3) {   // streaming pipeline, boundary B: the demodulated audio row (W0)
      float *xb = a.xch_b + ((size_t)ch * ASDR_STREAM_DEPTH + (size_t)(blk % ASDR_STREAM_DEPTH)) * ASDR_N + kF;
      if (ROLE == 2) {
        v4f v[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) v[m] = *reinterpret_cast<const v4f *>(L + W0 + kF + 32 * m);
        xch_store4x4(xb, v[0], v[1], v[2], v[3]);
        sig_pending = (uint32_t)blk + 1u;
        continue;
      } else {
        v4f v[4];
        if (is_am) {   // ... and the carrier level role 2's envelope path left for this block (one wait for both)
          const float *xc = a.xch_c + (size_t)ch * ASDR_STREAM_DEPTH + (size_t)(blk % ASDR_STREAM_DEPTH);
          asm volatile("global_load_dword %4, %6, off sc1\n\t"
                       "global_load_dwordx4 %0, %5, off sc1\n\tglobal_load_dwordx4 %1, %5, off offset:128 sc1\n\t"
                       "global_load_dwordx4 %2, %5, off offset:256 sc1\n\tglobal_load_dwordx4 %3, %5, off offset:384 sc1\n\t"
                       "s_waitcnt vmcnt(0)"
                       : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(carrier_now) : "v"(xb), "v"(xc) : "memory");
          carrier_fresh = true;
        } else
        xch_load4x4(xb, v[0], v[1], v[2], v[3]);
        if (sig_pending) { if (lane == 0) __hip_atomic_store(my_prog, sig_pending, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); sig_pending = 0u; }   // (vmcnt(0) inside the loads)
#pragma unroll
        for (int m = 0; m < 4; ++m) *reinterpret_cast<v4f *>(L + W0 + kF + 32 * m) = v[m];
        WAVE_SYNC();
      }
    }

    // the AGC gain table (row of 132 floats) is requested before the audio-filter pipeline and staged in LDS after it
    float4 agc_t4[5];
#pragma unroll
    for (int r = 0; r < 5; ++r) agc_t4[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    float agc_al_a = 0.f, agc_be_a = 0.f, agc_al_r = 0.f, agc_be_r = 0.f, agc_sg = 0.f;   // likewise the channel's AGC parameters
    uint32_t agc_hang = 0u;
    if (ABL_ON(ABL_AGC) && agc_en) {
      agc_al_a = P.agc_alpha_att; agc_be_a = P.agc_beta_att; agc_al_r = P.agc_alpha_rel; agc_be_r = P.agc_beta_rel;
      agc_sg = P.agc_static_gain; agc_hang = P.agc_hang_count;
      const float *gtab = a.agc_tab + (size_t)agc_tab_idx * ASDR_AGC_TAB_ROW;
      if constexpr (!C16) {   // (the 16-waves-per-CU form reads the table through L1 where it needs it: no LDS copy)
#pragma unroll
      for (int r = 0; r < 5; ++r) { const int q = s8 + 8 * r; if (q < 33) agc_t4[r] = reinterpret_cast<const float4 *>(gtab)[q]; }
      }
    }
    // ---- audio IIR filter, AudioSDR.cpp:149, 280-286: lanes s8 = 0..3 are the four stages ---------------------
    if constexpr ((MW_SHARE & 1) != 0) {
      if (ABL_ON(ABL_AF) && af_en) {   // (launch-uniform: all four waves or none)
        // Round 6: the AGC duty BESIDE the audio duty.  While the blocks of a bank attack (a fresh bank's first ~150 blocks, any bank after a level step)
        // the AGC's envelope chain (6 k cycles for which three waves parked) follows the audio cascades on the two waves that would otherwise sit out
        // the audio duty at the barrier: the blanker-duty wave (rel 0) takes the filtered samples from the rows as the cascades' last stage leaves
        // them (progress words of the two audio waves in LDS), forms |x| and beta |x| for all 32 channels, four samples per lane; the AGC-duty wave
        // (rel 3) follows IT with the lean chain (below): envelope after every sample, 32 channels on 32 lanes.  Taken when every channel of the
        // workgroup is in the lean regime and some channel attacked in the PREVIOUS block (decided behind the blanker chain's barrier, from words
        // published in front of it): a steady bank, whose blocks are quiet, does not pay for chains nobody needs.
        if constexpr (PIPE_AGC) {
          if (agc_piped && lead && agc_en) {
            float *mq = mwx + MWX * (wave * 8 + c8);
            mq[8] = agc_old0; mq[11] = agc_al_a; mq[12] = agc_be_a;
          }
        }
        TL(21);
        if constexpr (MW && (ASDR_MW_STRAGGLER_PRIO != 0 || ASDR_MW_PRIO_REL0_TAIL != 0)) __builtin_amdgcn_s_setprio(0);
        __syncthreads();               // every wave's demodulated audio is in its W0 rows
        TL(22);
        if (mw_audio_duty) {           // 16 cascades x 4 stages: the whole wave works
#ifndef ASDR_MW_PRIO_AUDIO
#define ASDR_MW_PRIO_AUDIO ASDR_MW_PRIO   /* the audio-cascade duty's priority (experiments: 0 = only the short chain duties are raised) */
#endif
          if (ASDR_MW_PRIO_AUDIO) __builtin_amdgcn_s_setprio(ASDR_MW_PRIO_AUDIO);
          const int st = pl_st;
          float sv[4];
          sv[0] = af_s4.x; sv[1] = af_s4.y; sv[2] = af_s4.z; sv[3] = af_s4.w;
          if (PIPE_AGC && agc_piped) biquad_pipe<PIPE_PK, PIPE_AGC>(lds_wg + (16 * (mw_rel - 1) + mw_casc) * STRIDE + W0, true, st, af_cf, sv, mw_flags + 1 + mw_rel, lane_i == 0);
          else biquad_pipe<PIPE_PK>(lds_wg + (16 * (mw_rel - 1) + mw_casc) * STRIDE + W0, true, st, af_cf, sv);   // (a steady bank: the cascades as they were)
          *reinterpret_cast<float4 *>(&mw_af_S->af_state[4 * st]) = make_float4(sv[0], sv[1], sv[2], sv[3]);
          if (ASDR_MW_PRIO_AUDIO) __builtin_amdgcn_s_setprio(0);
        } else if constexpr (PIPE_AGC) {
          if (agc_piped) {
            // Progress words are read with explicit ds_read + s_waitcnt (a `volatile` access is completed on the spot by the compiler, and through a generic
            // pointer it takes the flat path): the poll for a chunk is in flight during the previous chunk's arithmetic, and a chunk's operands are
            // requested a chunk ahead whenever the producer is known to be that far (two dependent LDS round trips per chunk -- poll, then operands --
            // are longer than the chunk's arithmetic under load).  The compiler does not count these reads in its own s_waitcnt operands: it then
            // waits for MORE than it needs, never for less.
            if (ASDR_MW_PRIO) __builtin_amdgcn_s_setprio(ASDR_MW_PRIO);
            if (mw_rel == 0) {
              // |x| clamped to 1.0 (:410-411; NaN-preserving like the owners' pass) -> [256 + ..), beta |x| -> [AGC_GV + ..): lane = (channel, half chunk)
              const int q = lane_i >> 1;
              float *Lc = lds_wg + q * STRIDE + 4 * (lane_i & 1);
              const float be_a = mwx[MWX * q + 12];
              const uint32_t src_lds = (uint32_t)(uintptr_t)(lds_int_ptr)(mw_flags + 2), dst_lds = (uint32_t)(uintptr_t)(lds_int_ptr)(mw_flags + 1);
              v2i pr;
              auto poll_issue = [&]() { asm volatile("ds_read_b64 %0, %1" : "=v"(pr) : "v"(src_lds) : "memory"); };
              auto poll_value = [&]() -> int {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pr) :: "memory");   // (and no row read moves above the poll it depends on)
                return __builtin_amdgcn_readfirstlane(pr.x < pr.y ? pr.x : pr.y);
              };
              poll_issue();
              int known = poll_value();
              while (known < 1) { __builtin_amdgcn_s_sleep(2); poll_issue(); known = poll_value(); }
              float xr[4];
              load4(Lc + W0, xr);
#pragma unroll 1
              for (int c = 0; c < ASDR_N / 8; ++c) {
                const int nc = (c + 1 < ASDR_N / 8) ? c + 1 : c;
                const bool pre = known > nc;   // (wave-uniform)
                float xn[4];
                if (pre) load4(Lc + W0 + 8 * nc, xn);
                poll_issue();   // (consumed behind the chunk)
                float ax[4], pb[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const float t = fabsf(xr[u]); ax[u] = (t > 1.0f) ? 1.0f : t; pb[u] = be_a * ax[u]; }
                store4(Lc + 256 + 8 * c, ax); store4(Lc + AGC_GV + 8 * c, pb);
                if (lane_i == 0) asm volatile("ds_write_b32 %0, %1" :: "v"(dst_lds), "v"(c + 1) : "memory");   // (the LDS executes a wave's operations in order)
                known = poll_value();
                if (!pre) {
                  while (known <= nc) { __builtin_amdgcn_s_sleep(1); poll_issue(); known = poll_value(); }
                  load4(Lc + W0 + 8 * nc, xn);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) xr[u] = xn[u];
              }
            } else if (lane_i < 8 * WAVES) {   // rel 3
              const int q = lane_i;
              float *Lc = lds_wg + q * STRIDE;
              float old_abs = mwx[MWX * q + 8];
              const float al_a = mwx[MWX * q + 11];
              const uint32_t src_lds = (uint32_t)(uintptr_t)(lds_int_ptr)(mw_flags + 1);
              int pr;
              auto poll_issue = [&]() { asm volatile("ds_read_b32 %0, %1" : "=v"(pr) : "v"(src_lds) : "memory"); };
              auto poll_value = [&]() -> int {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pr) :: "memory");
                return __builtin_amdgcn_readfirstlane(pr);
              };
              poll_issue();
              int known = poll_value();
              while (known < 1) { __builtin_amdgcn_s_sleep(2); poll_issue(); known = poll_value(); }
              float ax[8], pb[8];
              load8(Lc + 256, ax); load8(Lc + AGC_GV, pb);
#pragma unroll 1
              for (int c = 0; c < ASDR_N / 8; ++c) {
                const int nc = (c + 1 < ASDR_N / 8) ? c + 1 : c;
                const bool pre = known > nc;
                float axn[8], pbn[8];
                if (pre) { load8(Lc + 256 + 8 * nc, axn); load8(Lc + AGC_GV + 8 * nc, pbn); }
                poll_issue();
                float ov[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                  const float pa = al_a * old_abs;
                  const float v_new = pa + pb[u];
                  old_abs = (ax[u] > old_abs) ? v_new : old_abs;
                  ov[u] = old_abs;
                }
                store8(Lc + AGC_GV + 8 * c, ov);
                known = poll_value();
                if (!pre) {
                  while (known <= nc) { __builtin_amdgcn_s_sleep(1); poll_issue(); known = poll_value(); }
                  load8(Lc + 256 + 8 * nc, axn); load8(Lc + AGC_GV + 8 * nc, pbn);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) { ax[u] = axn[u]; pb[u] = pbn[u]; }
              }
              mwx[MWX * q + 8] = old_abs;
            }
            if (ASDR_MW_PRIO) __builtin_amdgcn_s_setprio(0);
          }
        }
        TL(23);
        __syncthreads();
        TL(24);
      }
    } else
    if (ABL_ON(ABL_AF) && __any(af_en)) {
      const int st = pl_st;
      const bool on = af_en_pa && pa_real;   // (the enable of the channel this lane's cascade belongs to)
      float sv[4];
      sv[0] = af_s4.x; sv[1] = af_s4.y; sv[2] = af_s4.z; sv[3] = af_s4.w;
      ChanSmall *const Saf = Spa;   // (by all lanes)
      CHAIN_PRIO_ON();
      biquad_pipe<PIPE_PK>(Lpa + W0, on, st, af_cf, sv);
      CHAIN_PRIO_OFF();
      if (on) *reinterpret_cast<float4 *>(&Saf->af_state[4 * st]) = make_float4(sv[0], sv[1], sv[2], sv[3]);
      WAVE_SYNC();
    }
    TL(11);
    TAP_ROW(ASDR_TAP_AUDIO_FILT, W0);

    // ---- AGC, AudioSDR.cpp:404-436 --------------------------------------------------------------------------
    // Split into (a) the sequential envelope/hang recurrence, which records for every sample the envelope value
    // that governs its gain, and (b) a parallel pass that evaluates the static compressor and applies the gain.
    if (ABL_ON(ABL_AGC) && __any(agc_en)) {
      const float *tab = C16 ? a.agc_tab + (size_t)agc_tab_idx * ASDR_AGC_TAB_ROW : L + AGC_TAB;
      const float gain_in = agc_gain_in;
      // A QUIET block: in no channel of the wave does a sample exceed the envelope (the block's largest |x|, formed by all lanes:
      // piece maxima, then three DPP steps over the channel's eight lanes) and no hang counter can run out inside the block -- then
      // the envelope, the governing value and the gain do not move in any of the 128 samples (:412-428: neither branch fires): the
      // counters drop by 128, every sample gets the gain carried in, and neither the table nor the per-sample rows are needed.
      // With a steady signal that is most blocks once the envelope has crept up to the peaks (profiles/README.md).
      float av16[16];
      DEFINE_ALL_PATHS(av16, 16);
      float blockmax = 0.0f;
      if (agc_en) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          float au[8];
          load8(L + W0 + kA + 64 * h, au);
          // |x| clamped to 1.0 (:410-411) for every sample, in parallel; the sequential pass below replaces it in place by the
          // governing envelope value.  (v_max ignores a NaN operand, as `NaN > old` is false: a NaN sample never attacks.)
#pragma unroll
          for (int j = 0; j < 8; ++j) { const float ax = fabsf(au[j]); av16[8 * h + j] = (ax > 1.0f) ? 1.0f : ax; }   // NaN-preserving like the reference
          float mx = av16[8 * h];
#pragma unroll
          for (int u = 1; u < 8; ++u) mx = fmaxf(mx, av16[8 * h + u]);
          blockmax = (h == 0) ? mx : fmaxf(blockmax, mx);
        }
        blockmax = fmaxf(blockmax, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(blockmax), 0xB1, 0xF, 0xF, true)));    // quad_perm [1,0,3,2]
        blockmax = fmaxf(blockmax, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(blockmax), 0x4E, 0xF, 0xF, true)));    // quad_perm [2,3,0,1]
        blockmax = fmaxf(blockmax, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(blockmax), 0x141, 0xF, 0xF, true)));   // row_half_mirror: the partner quad
      }
      const float am_level = (float)(2.0 * (double)(carrier_fresh ? carrier_now : agc_carrier0));
      const float am_clamped = (am_level > 1.0f) ? 1.0f : am_level;
      // (AM: twice the carrier level stands in for |x| of every sample, :407-409 -- the lead lane holds it)
      const bool agc_quiet = ASDR_AGC_QUIET_PATH && __all(!(agc_en && lead) || (agc_hc0 >= 128u && agc_hang >= 8u && !((is_am ? am_clamped : blockmax) > agc_old0)));
      // (the four-wave form, round 6: the AGC duty wave ran the lean chain beside the audio duty -- the envelope rows are there already; behind barrier 4)
      if (agc_en && !agc_quiet) {   // stage the channel's gain table and the |x| rows in LDS
        if constexpr (!C16) {
#pragma unroll
        for (int r = 0; r < 5; ++r) { const int q = s8 + 8 * r; if (q < 33) *reinterpret_cast<float4 *>(L + AGC_TAB + 4 * q) = agc_t4[r]; }
        }
        if (!agc_piped) { store8(L + AGC_GV + kA, av16); store8(L + AGC_GV + kA + 64, av16 + 8); }
      }
      if (!agc_quiet) WAVE_SYNC();
      TL(12);
      if (agc_quiet) {
        if (agc_en && lead) {
          S->agc_hang_counter = agc_hc0 - 128u;      // envelope and gain stay as they are in HBM
          status = (status & ~ASDR_S_AGC_ACTIVE) | (((double)gain_in < 0.99) ? ASDR_S_AGC_ACTIVE : 0u);
        }
      }
      {
      // One channel's envelope / hang recurrence on ONE lane: Lc = the channel's rows (|x| row at AGC_GV, replaced in place by the governing
      // envelope values; gain table at AGC_TAB), Sc = its state row.  Returns the gain after the block.
      float agc_old_end = 0.0f; uint32_t agc_hc_end = 0u;
      auto agc_chain = [&](auto own_tag, float *Lc, ChanSmall *Sc, float old0, uint32_t hc0, float g_in, float al_a, float be_a, float al_r, float be_r, uint32_t hang,
                           bool am, float am_lvl) -> float {
        float old_abs = old0;
        uint32_t hc = hc0;
        float gv = -1.0f;   // envelope value governing the current gain; -1 = no update yet in this block
        // Per sample (:412-428): attack if |x| > envelope, else release unless the hang counter runs.  A lone wave issues one
        // instruction per ~4.5 cycles whatever the dependency depth (tools/ubench/dep_chain.hip), so the loop is written for the
        // fewest instructions: only the taken branch's alpha / beta are selected, then one mul, mul, add.
        // AM: twice the carrier level stands in for |x| of every sample (:407-409): the lead lane overwrites its channel's row
        // with that constant once, so that the recurrence below needs no per-sample select
        if (am) {
#pragma unroll 1
          for (int i = 0; i < ASDR_N; i += 4) *reinterpret_cast<float4 *>(Lc + AGC_GV + i) = make_float4(am_lvl, am_lvl, am_lvl, am_lvl);
        }
        float x_[8];
        load8(Lc + AGC_GV, x_);
#pragma unroll 1
        for (int i2 = 0; i2 < ASDR_N; i2 += 16) {   // two chunks per trip: the prefetched chunk ping-pongs between x and xn (no copies)
          float xn_[8];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int i = i2 + 8 * half;
          float *x = half ? xn_ : x_, *xn = half ? x_ : xn_;
          float gvv[8];
          load8(Lc + AGC_GV + ((i + 8 < ASDR_N) ? i + 8 : i), xn);   // next chunk, a step ahead
          // Hanging chunk: no sample of the chunk attacks (none exceeds the envelope, which therefore does not move) and the hang
          // counter cannot run out inside it -> envelope, gain and governing value stay, the counter drops by 8.  With a steady
          // signal most chunks between two envelope peaks are like this.  (v_max ignores a NaN operand, as `NaN > old` is false.)
          float mx = x[0];
#pragma unroll
          for (int u = 1; u < 8; ++u) mx = fmaxf(mx, x[u]);
          // While the hang counter cannot run out inside the chunk (counter >= 8 now, and an attack re-arms it with >= 8), nothing
          // releases: only attacks change the envelope.  The counter is then kept as a base value: an attack at sample u sets it to
          // hang + (u + 1), so that base - 8 is the counter after the chunk.  All tests are wave-uniform (EXEC = the lead lanes).
          const bool no_release = __all(hc >= 8u && hang >= 8u);
          if (no_release && __all(!(mx > old_abs))) {
            hc -= 8u;
#pragma unroll
            for (int u = 0; u < 8; ++u) gvv[u] = gv;
          } else if (no_release) {
            uint32_t hcb = hc;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const float av = x[u];
              const bool att = av > old_abs;
#ifndef ASDR_AGC_ATTACK_BRANCHFREE
#define ASDR_AGC_ATTACK_BRANCHFREE 1
#endif
              // The four-wave form's duty wave runs 32 channels' chains at once: in an attacking chunk some channel attacks at almost every sample, so the
              // per-sample branch (right for the eight channels of a wave of its own: round 4) only adds its test -- straight-line selects there (round 6).
              if ((MW && ASDR_AGC_ATTACK_BRANCHFREE) || __any(att)) {   // (one wave per workgroup: rare -- a sample above the envelope in some channel of the wave)
                if (!(MW && ASDR_AGC_ATTACK_BRANCHFREE)) asm volatile("");   // keeps this a branch (the compiler would otherwise flatten it into selects for every sample)
                const float pa = al_a * old_abs, pb = be_a * av;   // (:418)
                const float v_new = pa + pb;
                old_abs = att ? v_new : old_abs;
                gv = att ? v_new : gv;
                hcb = att ? hang + (uint32_t)(u + 1) : hcb;
              }
              gvv[u] = gv;
            }
            hc = hcb - 8u;
          } else if (__all(hc == 0u && hang == 0u)) {
            // No hang time at all (setAGChangTime(0)) and no counter running: the counter stays 0 whatever happens (an attack re-arms it with
            // 0), so EVERY sample updates the envelope -- attack or release coefficients by the one comparison -- and governs its own gain.
            // Half the general form's instructions per sample (round 5; `robustness.agc_general_form` in the bench line).
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const float av = x[u];
              const bool att = av > old_abs;
              const float al = att ? al_a : al_r, be = att ? be_a : be_r;
              const float pa = al * old_abs, pb = be * av;
              old_abs = pa + pb;
              gvv[u] = old_abs;
            }
            gv = old_abs;
          } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const float av = x[u];
              const bool att = av > old_abs;                 // attack
              const bool idle = (hc == 0u);                  // not hanging: release when not attacking
              const float al = att ? al_a : al_r, be = att ? be_a : be_r;   // only the branch that is taken is evaluated (:418 / :424)
              const float pa = al * old_abs, pb = be * av;
              const float v_new = pa + pb;
              const bool upd = att || idle;
              old_abs = upd ? v_new : old_abs;
              gv = upd ? v_new : gv;
              hc = att ? hang : __builtin_elementwise_sub_sat(hc, 1u);   // 0 stays 0
              gvv[u] = gv;
            }
          }
          store8(Lc + AGC_GV + i, gvv);
        }
        }
        const float g_end = (gv < 0.0f) ? g_in : agc_compress(C16 ? tab : Lc + AGC_TAB, gv);
        if constexpr (decltype(own_tag)::value) { agc_old_end = old_abs; agc_hc_end = hc; }   // (the four-wave form: the channel's own wave stores them)
        else { Sc->agc_old_abs = old_abs; Sc->agc_hang_counter = hc; Sc->agc_gain = g_end; }
        return g_end;
      };
      // What the LEAN chain (below: the four-wave form's duty, or the wave's own lead lanes) leaves to be done, by all eight lanes of a channel.
      auto agc_lean_finish = [&](float old_end_lead) {
        // The lean chain left the envelope after every sample in the row.  Sample u attacked iff |x|[u] > the envelope after sample u - 1 (carried in for
        // u = 0); before the block's first attack the governing value does not exist (-1: the gain carried in applies); the counter after the block is
        // hang - (127 - last attack), or the carried one less 128; the gain after the block is the compressor at the last envelope, if anything attacked.
        int first = 128, last = -1;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          float ov[8];
          load8(L + AGC_GV + kA + 64 * h, ov);
          const int p0 = kA + 64 * h;
          float before = (p0 == 0) ? agc_old0 : L[AGC_GV + ((p0 == 0) ? 0 : p0 - 1)];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const bool att = av16[8 * h + j] > before;
            first = (att && p0 + j < first) ? p0 + j : first;
            last = att ? p0 + j : last;
            before = ov[j];
          }
        }
        // over the channel's eight lanes: minimum of `first`, maximum of `last` (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror: as the block maximum above)
        { const int t = __builtin_amdgcn_update_dpp(0, first, 0xB1, 0xF, 0xF, true); first = t < first ? t : first; }
        { const int t = __builtin_amdgcn_update_dpp(0, first, 0x4E, 0xF, 0xF, true); first = t < first ? t : first; }
        { const int t = __builtin_amdgcn_update_dpp(0, first, 0x141, 0xF, 0xF, true); first = t < first ? t : first; }
        { const int t = __builtin_amdgcn_update_dpp(0, last, 0xB1, 0xF, 0xF, true); last = t > last ? t : last; }
        { const int t = __builtin_amdgcn_update_dpp(0, last, 0x4E, 0xF, 0xF, true); last = t > last ? t : last; }
        { const int t = __builtin_amdgcn_update_dpp(0, last, 0x141, 0xF, 0xF, true); last = t > last ? t : last; }
        WAVE_SYNC();   // (every lane has read the envelopes it compares with before anybody overwrites a row entry)
#pragma unroll
        for (int h = 0; h < 2; ++h) {   // the rows as the general chain leaves them: -1 in front of the first attack
          const int p0 = kA + 64 * h;
          if (p0 < first) {
            float ov[8];
            load8(L + AGC_GV + p0, ov);
#pragma unroll
            for (int j = 0; j < 8; ++j) ov[j] = (p0 + j < first) ? -1.0f : ov[j];
            store8(L + AGC_GV + p0, ov);
          }
        }
        if (lead) {
          const float old_end = old_end_lead;
          const float g_end = (last < 0) ? gain_in : agc_compress(tab, old_end);
          status = (status & ~ASDR_S_AGC_ACTIVE) | (((double)g_end < 0.99) ? ASDR_S_AGC_ACTIVE : 0u);
          S->agc_old_abs = old_end;
          S->agc_hang_counter = (last < 0) ? agc_hc0 - 128u : agc_hang - (uint32_t)(127 - last);
          S->agc_gain = g_end;
        }
      };
      if constexpr ((MW_SHARE & 4) != 0) {
        // MW: every wave says whether its block is quiet; the AGC-duty wave (rel 3) runs the chains of the other waves' channels, one per lane
        // (the |x| rows and the gain tables are in the channels' LDS rows; the scalars come from the channels' state / parameter rows)
        constexpr bool LEAN_OK = (ASDR_AGC_LEAN != 0) && (ASDR_MW_OWN_STORES != 0);
        if (!agc_piped) {
        if (lead) {
          float *mq = mwx + MWX * (wave * 8 + c8);
          mq[5] = __int_as_float(agc_quiet ? 1 : 0); mq[6] = am_clamped;
          if (ASDR_MW_OWN_STORES && !agc_quiet) {   // the chain's scalars: this wave has loaded them for the quiet test (the duty wave asked for them again: 8 scattered loads per lane in front of its chain)
            *reinterpret_cast<float4 *>(mq + 8) = make_float4(agc_old0, __uint_as_float(agc_hc0), gain_in, agc_al_a);
            *reinterpret_cast<float4 *>(mq + 12) = make_float4(agc_be_a, agc_al_r, agc_be_r, __uint_as_float(agc_hang));
          }
        }
        TL(25);
        __syncthreads();
        TL(26);
        if (ASDR_MW_PRIO && mw_rel == 3) __builtin_amdgcn_s_setprio(ASDR_MW_PRIO);
        // Round 6, the LEAN chain.  When no hang counter of the duty wave's channels can run out inside the block (counter >= 128 at its start and a
        // hang time of >= 128 samples: an attack re-arms it with that) nothing releases, and the only recurrence is the envelope itself:
        //     attack = |x| > envelope;  envelope = attack ? alpha envelope + beta |x| : envelope        (AudioSDR.cpp:412-420)
        // -- four instructions on the chain per sample instead of eight.  The duty wave leaves the envelope AFTER every sample in the row; which
        // sample attacked (|x|[u] > envelope after sample u - 1), from where on the governing value exists at all, the hang counter and the gain after
        // the block are functions of that row and are formed in parallel by the channels' own waves behind the barrier (below).
        if (mw_rel == 3 && lane_i < 8 * WAVES) {
          const int q = lane_i;
          const bool active = (__float_as_int(mwx[MWX * q + 5]) == 0);
          float4 i0 = make_float4(0.f, 0.f, 0.f, 0.f), i1 = i0;
          if (ASDR_MW_OWN_STORES && active) { i0 = *reinterpret_cast<const float4 *>(mwx + MWX * q + 8); i1 = *reinterpret_cast<const float4 *>(mwx + MWX * q + 12); }
          const bool lean = LEAN_OK && !is_am && __all(!active || (__float_as_uint(i0.y) >= 128u && __float_as_uint(i1.w) >= 128u));
          if (q == 0) mw_flags[0] = lean ? 1 : 0;
          if (lean) {
            if (active) {
              float *Lc = lds_wg + q * STRIDE;
              float old_abs = i0.x;
              const float al_a = i0.w, be_a = i1.x;
              float x_[8];
              load8(Lc + AGC_GV, x_);
#pragma unroll 1
              for (int i2 = 0; i2 < ASDR_N; i2 += 16) {
                float xn_[8];
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                  const int i = i2 + 8 * half;
                  float *x = half ? xn_ : x_, *xn = half ? x_ : xn_;
                  load8(Lc + AGC_GV + ((i + 8 < ASDR_N) ? i + 8 : i), xn);
                  float pb[8], ov[8];
#pragma unroll
                  for (int u = 0; u < 8; ++u) pb[u] = be_a * x[u];
#pragma unroll
                  for (int u = 0; u < 8; ++u) {
                    const float pa = al_a * old_abs;
                    const float v_new = pa + pb[u];
                    old_abs = (x[u] > old_abs) ? v_new : old_abs;
                    ov[u] = old_abs;
                  }
                  store8(Lc + AGC_GV + i, ov);
                }
              }
              mwx[MWX * q + 8] = old_abs;
            }
          } else
          if (active) {
            if (ASDR_MW_OWN_STORES) {
              mwx[MWX * q + 7] = agc_chain(std::true_type{}, lds_wg + q * STRIDE, nullptr, i0.x, __float_as_uint(i0.y), i0.z, i0.w, i1.x, i1.y, i1.z, __float_as_uint(i1.w), is_am, mwx[MWX * q + 6]);
              mwx[MWX * q + 8] = agc_old_end; mwx[MWX * q + 9] = __uint_as_float(agc_hc_end);
            } else {
            const int chq = mw_channel(q);
            ChanSmall *Sq = row_ptr(a.small, (uint32_t)chq * (uint32_t)sizeof(ChanSmall));
            const ChanParams *Pq = row_ptr(a.params, (uint32_t)chq * (uint32_t)sizeof(ChanParams));
            mwx[MWX * q + 7] = agc_chain(std::false_type{}, lds_wg + q * STRIDE, Sq, Sq->agc_old_abs, Sq->agc_hang_counter, Sq->agc_gain, Pq->agc_alpha_att, Pq->agc_beta_att,
                                       Pq->agc_alpha_rel, Pq->agc_beta_rel, Pq->agc_hang_count, is_am, mwx[MWX * q + 6]);
            }
          }
        }
        if (ASDR_MW_PRIO && mw_rel == 3) __builtin_amdgcn_s_setprio(0);
        TL(27);
        __syncthreads();
        TL(28);
        } else { TL(25); TL(26); TL(27); TL(28); }
        const bool lean_done = LEAN_OK && (agc_piped || __builtin_amdgcn_readfirstlane(mw_flags[0]) != 0);
        if (lean_done && !agc_quiet && agc_en) {
          agc_lean_finish(lead ? mwx[MWX * (wave * 8 + c8) + 8] : 0.0f);
        } else
        if (!agc_quiet && lead) {
          const float *mq = mwx + MWX * (wave * 8 + c8);
          const float g_end = mq[7]; status = (status & ~ASDR_S_AGC_ACTIVE) | (((double)g_end < 0.99) ? ASDR_S_AGC_ACTIVE : 0u);
          if (ASDR_MW_OWN_STORES && agc_en) { S->agc_old_abs = mq[8]; S->agc_hang_counter = __float_as_uint(mq[9]); S->agc_gain = g_end; }
        }
      } else {
#ifndef ASDR_AGC_LEAN1
#define ASDR_AGC_LEAN1 1   /* the lean chain for the forms whose waves run their own channels' chains too (one lane per channel) */
#endif
        // The lean chain (see the four-wave form's AGC duty above) on the wave's own lead lanes: no channel of the wave can run out of its hang counter
        // inside the block (and none is AM: twice the carrier level stands in for |x| there) -- the envelope recurrence alone, half the general form's
        // instructions per sample; what the block leaves behind is formed in parallel by agc_lean_finish.
        constexpr bool LEAN1_OK = (ASDR_AGC_LEAN1 != 0) && !C16 && !(ROLE == 0 && WAVES == 1 && !ONEBLK_ && !HAS_ALS && !HAS_SAM && UNIFORM);   // (the looped plain kernel spills with it)
        const bool lean1 = LEAN1_OK && !agc_quiet && __all(!(agc_en && lead) || (!is_am && agc_hc0 >= 128u && agc_hang >= 128u));
        if (!agc_quiet) CHAIN_PRIO_ON();
        if (lean1) {
          float old_abs = agc_old0;
          if (agc_en && lead) {
            const float al_a = agc_al_a, be_a = agc_be_a;
            float x_[8];
            load8(L + AGC_GV, x_);
#pragma unroll 1
            for (int i2 = 0; i2 < ASDR_N; i2 += 16) {
              float xn_[8];
#pragma unroll
              for (int half = 0; half < 2; ++half) {
                const int i = i2 + 8 * half;
                float *x = half ? xn_ : x_, *xn = half ? x_ : xn_;
                load8(L + AGC_GV + ((i + 8 < ASDR_N) ? i + 8 : i), xn);
                float pb[8], ov[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) pb[u] = be_a * x[u];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                  const float pa = al_a * old_abs;
                  const float v_new = pa + pb[u];
                  old_abs = (x[u] > old_abs) ? v_new : old_abs;
                  ov[u] = old_abs;
                }
                store8(L + AGC_GV + i, ov);
              }
            }
          }
          WAVE_SYNC();
          if (agc_en) agc_lean_finish(old_abs);
        } else
        if (!agc_quiet && agc_en && lead) {
          const float g_end = agc_chain(std::false_type{}, L, S, agc_old0, agc_hc0, gain_in, agc_al_a, agc_be_a, agc_al_r, agc_be_r, agc_hang, is_am, am_clamped);
          status = (status & ~ASDR_S_AGC_ACTIVE) | (((double)g_end < 0.99) ? ASDR_S_AGC_ACTIVE : 0u);
        }
        if (!agc_quiet) CHAIN_PRIO_OFF();
      }
      }
      if (!agc_quiet) WAVE_SYNC();
      TL(13);
      if (agc_quiet) {
        if (agc_en) {   // every sample at the gain carried in (:430-433)
          const float sg = agc_sg;
#pragma unroll 1
          for (int h = 0; h < 2; ++h) {
            float au[8];
            load8(L + W0 + kA + 64 * h, au);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              float o = gain_in * sg * au[j];
              o = (o > 1.0f) ? 1.0f : o;
              o = (o < -1.0f) ? -1.0f : o;
              au[j] = o;
            }
            store8(L + W0 + kA + 64 * h, au);
          }
        }
      } else if (agc_en) {
        const float sg = agc_sg;
#pragma unroll 1
        for (int h = 0; h < 2; ++h) {
          float au[8], gvr[8];
          load8(L + W0 + kA + 64 * h, au); load8(L + AGC_GV + kA + 64 * h, gvr);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float gain = (gvr[j] < 0.0f) ? gain_in : agc_compress(tab, gvr[j]);
            float o = gain * sg * au[j];
            o = (o > 1.0f) ? 1.0f : o;
            o = (o < -1.0f) ? -1.0f : o;
            au[j] = o;
            if ((j & 3) == 3) SCHED_FENCE();
          }
          store8(L + W0 + kA + 64 * h, au);
        }
      }
      WAVE_SYNC();
    }
    TL(14);
    TAP_ROW(ASDR_TAP_AGC, W0);

    // ---- ALS adaptive notch / peak filter, AudioSDR.cpp:324-352 (ALS instantiations only) -------------------------
    // LDS overlay from word 128 of the channel's row: [kept tail of the previous block | current block | even taps | odd taps | e].
    //   516-float rows: 128 + 128 + 64 + 64 (any M <= 128, any delay)                      -> 9 waves per CU
    //   388-float rows: 64 + 128 + 32 + 32 (M <= 64 and delay + M <= 65: the host's choice) -> 12 waves per CU: the filter is
    //   a chain of dependent operations and its throughput follows the resident waves (profiles/README.md, occupancy experiment)
    if constexpr (HAS_ALS) {
      constexpr bool COMPACT = ASDR_COMPACT_ROWS(STRIDE);
      constexpr int AH = COMPACT ? 64 : 128;       // history kept from the previous block
      constexpr int XB = 128 - (128 - AH);         // L[XB + idx] = sample idx of the reference's 256-sample buffer (idx >= 128 - AH)
      constexpr int AW = XB + 256;                 // taps
      constexpr int WH = AH / 2;                   // taps per half (ALS_TAP)
      constexpr int SCR = AW + 2 * WH + 1;         // error broadcast word
      constexpr int M0 = COMPACT ? 2 : 0;          // first 32-float piece of the previous block / number of tap pieces kept
      constexpr int NW = COMPACT ? 2 : 4;
      if (ABL_ON(ABL_ALS) && __any(als_en)) {
        const int M = P.als_m, D = P.als_delay;
        const float lam = P.als_lambda;   // (requested once, not in every tap-update epoch)
        const bool adaptive = pflags & ASDR_F_ALS_ADAPTIVE, notch = pflags & ASDR_F_ALS_NOTCH;
        const uint32_t as = (a.als_phase + (uint32_t)blk) & 1u;   // ring slot of this block's input; the other one holds the previous block's
        const uint32_t xoff = (uint32_t)ch * 1024u + 4u * (uint32_t)kF;   // float rows: pieces kF + 32m
        float *gx = row_ptr(a.als_x, xoff + as * 512u), *gw = row_ptr(a.als_w, (uint32_t)ch * 512u + 4u * (uint32_t)kF);
        const float *gx_prev = row_ptr(a.als_x, xoff + (as ^ 1u) * 512u);
        if (als_en) {
          float tx[16], tw[16], tn[16];   // all row loads in flight together
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            if (m >= M0) load4(gx_prev + 32 * m, tx + 4 * m);
            if (m < NW) load4(gw + 32 * m, tw + 4 * m);
            load4(L + W0 + kF + 32 * m, tn + 4 * m);
          }
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            if (m >= M0) store4(L + XB + kF + 32 * m, tx + 4 * m);            // previous block
            store4(L + XB + 128 + kF + 32 * m, tn + 4 * m); store4(gx + 32 * m, tn + 4 * m);
            if (m < NW) {
              float *we = L + AW + ((kF + 32 * m) >> 1);   // taps de-interleaved (ALS_TAP)
              *reinterpret_cast<float2 *>(we) = make_float2(tw[4 * m], tw[4 * m + 2]);
              *reinterpret_cast<float2 *>(we + WH) = make_float2(tw[4 * m + 1], tw[4 * m + 3]);
            }
          }
        }
        WAVE_SYNC();
        als_compute<COMPACT, XB, AW, WH, SCR, ALS_OUT>(L, als_en, adaptive, notch, M, D, lam, s8, k0);
        WAVE_SYNC();
        // The guard is formed from an OPAQUE copy of the flag word.  From `als_en` itself, in the loop-free uniform instantiations, hipcc
        // (ROCm 7.2) emits it with the wrong polarity -- `s_bitcmp0_b32 flags, 3; s_cselect_b64 m, -1, 0` (m = NOT als_en, shared with the
        // staging branch above) and here `s_andn2_b64 vcc, exec, m; s_cbranch_vccnz <behind the stores>` -- so the taps never went back to
        // their HBM rows and every block started from the taps of the first one: the ALS tap ~1 % off with every earlier tap exact (the
        // round's "parity failure not understood"; found in the ISA, profiles/README.md).
        uint32_t flags_again = pflags;
        if constexpr (UNIFORM) asm volatile("" : "+s"(flags_again)); else asm volatile("" : "+v"(flags_again));
        if (flags_again & ASDR_F_ALS_EN) {
#pragma unroll
          for (int m = 0; m < NW; ++m) {
            const float *we = L + AW + ((kF + 32 * m) >> 1);
            const float2 e2 = *reinterpret_cast<const float2 *>(we), o2 = *reinterpret_cast<const float2 *>(we + WH);
            const float t[4] = {e2.x, o2.x, e2.y, o2.y};
            store4(gw + 32 * m, t);
          }
        }
      }
    }
    if (TO_ALS && valid) {   // this block's ALS input (AudioSDR.cpp:326-329: _als_in[n_block + i] = buff[i]) -> its ring slot; the filter is the next launch
      float *gx = row_ptr(a.als_x, (uint32_t)ch * 1024u + ((a.als_phase + (uint32_t)blk) & 1u) * 512u + 4u * (uint32_t)kF);
      float *gs = (a.als_stage != nullptr) ? row_ptr(a.als_stage, (uint32_t)ch * (uint32_t)(ASDR_ALS_STAGE_SLOTS * 512) + ((a.als_stage_cur + (uint32_t)blk) % ASDR_ALS_STAGE_SLOTS) * 512u + 4u * (uint32_t)kF) : nullptr;   // ALS role streams: the row the filter launch reads
#pragma unroll
      for (int m = 0; m < 4; ++m) { float t[4]; load4(L + W0 + kF + 32 * m, t); store4(gx + 32 * m, t); if (gs != nullptr) store4(gs + 32 * m, t); }
    }
    if (DO3 && !TO_ALS && valid && a.audio_prev != nullptr) {   // _audioOut as this block leaves it: what an unknown mode value would re-process
      float *ap = row_ptr(a.audio_prev, (uint32_t)ch * 512u + 4u * (uint32_t)kF);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        float t[4]; load4(L + W0 + kF + 32 * m, t);
#ifndef ASDR_PREV_TEMPORAL
        store4_nt(ap + 32 * m, t);   // written every block, read (if ever) a launch later: streaming store, keeps L2 for the rows that come back
#else
        store4(ap + 32 * m, t);
#endif
      }
    }
    {
      // ---- output, AudioSDR.cpp:158-161: float product, x 32767.0 in binary64, truncate, wrap to int16 ------
      const float og = P.output_gain;
      union { int4 v; int16_t s[8]; } ro[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float au[8];
        load8(L + W0 + kA + 64 * h, au);
        TAP8(ASDR_TAP_ALS, h, au);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int v = (int)((double)(og * au[j]) * 32767.0);
          ro[h].s[j] = muted ? (int16_t)0 : (int16_t)v;
        }
        SCHED_FENCE();
      }
      if (DO3 && !TO_ALS && valid) {
        int4 *po = reinterpret_cast<int4 *>(a.out + io_out);
#ifndef ASDR_TEMPORAL_OUT
        store_int4_nt(po, ro[0].v); store_int4_nt(po + 8, ro[1].v);
#else
        po[0] = ro[0].v; po[8] = ro[1].v;
#endif
      }
    }
    if (POST_ROLE) store_status_bits(ASDR_S_AGC_ACTIVE);   // (the pre role and the PLL of the NEXT block may be running beside this one)
    else if (!STREAM) { if (lead) S->status = status; }
    else { store_status_bits(ASDR_S_AGC_ACTIVE); sig_pending = (uint32_t)blk + 1u; }
    WAVE_SYNC();
    // The first wave of every settings group leaves the NEXT block's local-oscillator pairs in the group's entry of the other half (LoEntry).
    if (a.lo_write && ROLE != 2 && (lo_slot & a.lo_writer_bit) && (lo_slot & 0xFFu) != 0u && blk == a.n_blocks - 1 && mix_uni && __all(mix_early)) {
      uint32_t lo_wi = ((a.lo_parity & 1u) ^ 1u) * ASDR_LO_ENTRIES + lo_e;
      asm volatile("" : "+v"(lo_wi));   // the entry's address is formed HERE (hoisted, it sits in a VGPR pair for the whole kernel)
      LoEntry *lo_wr = a.lo_cache + lo_wi;
      const float start = lo_hit ? lo_end : __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(mphase_end)));
      const float inc = __uint_as_float(inc_first);
      if (lane == 0) {
        float phase = start;
        const bool up = !(inc < 0.0f);
        const float wrapv = up ? -two_pi : two_pi, lim = up ? two_pi : 0.0f;
        const uint32_t flip = up ? 0u : 0x80000000u;
#pragma unroll 1
        for (int i = 0; i < ASDR_N; i += 8) {
          float pv[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            pv[u] = phase;
            const float t = phase + inc, tw = t + wrapv;
            phase = (__uint_as_float(__float_as_uint(t) ^ flip) > lim) ? tw : t;
          }
          store8(lds + (C16 ? 0 : PH) + i, pv);   // (the block is done: in the 16-waves-per-CU form channel 0's first 128 words serve)
        }
        lo_wr->key_phase = __float_as_uint(start); lo_wr->key_inc = __float_as_uint(inc); lo_wr->phase_end = phase;
      }
      WAVE_SYNC();
      const float2 p2 = *reinterpret_cast<const float2 *>(lds + (C16 ? 0 : PH) + 2 * lane_i);
      const float ph2[2] = {p2.x, p2.y};
      float c2[2], s2[2];
      sincos_batch<2, SINE_LDS>(sine, ph2, c2, s2, two_pi, SinIndexK{K.inv_two_pi_d, K.sin_index_scale_d}, K.half_pi_d);
      *reinterpret_cast<float2 *>(lo_wr->c + 2 * lane_i) = make_float2(c2[0], c2[1]);
      *reinterpret_cast<float2 *>(lo_wr->s + 2 * lane_i) = make_float2(s2[0], s2[1]);
      WAVE_SYNC();
    }
    TL(15);
  }
  if (STREAM && sig_pending) stream_signal(my_prog, sig_pending, lane);
#undef P
#undef PL_CH
#undef Sp
#undef Spa
#undef Ppl
#undef Ppa
#undef Lp
#undef Lpa
#undef pl_st
#undef pl_iq
#undef pl_c
#undef pa_c
#undef pa_real
}

// Three instantiations by what a channel needs (the host launches each sub-range of its sorted schedule with its own):
//   plain  no ALS filter, not SAM: 388 floats of LDS per channel, sine table through L1, no PLL code (12,416 B per wave ->
//          12 waves/CU at <= 168 VGPRs)
//   sam    SAM mode: the PLL's per-sample dependent sin/cos lookups read a per-wave LDS copy of the table
//   als    ALS filter enabled (any mode): 516 floats per channel (256-sample ALS history + 128 taps in LDS)
// each in a form for waves whose 8 slots share one schedule key (uniform: scalar mode / flag tests) and a general form for the
// remainders of the key groups and the padding.
#define ASDR_KERNEL(name, lds_floats, bounds, stride, als, sam, uni, waves)                                     \
  extern "C" __global__ __launch_bounds__(64 * waves, bounds) void name(UpdateArgs a) {                         \
    __shared__ __attribute__((aligned(16))) float lds[lds_floats];                                              \
    asdr_update_body<stride, als, sam, uni, waves>(a, lds);                                                     \
  }
ASDR_KERNEL(asdr_update_kernel, 8 * ASDR_STRIDE, ASDR_WAVES_PER_EU, ASDR_STRIDE, false, false, true, 1)
ASDR_KERNEL(asdr_update_kernel_mixed, 8 * ASDR_STRIDE, ASDR_WAVES_PER_EU, ASDR_STRIDE, false, false, false, 1)
// ... and the loop-free twin of the uniform-key kernel that carries the big batches (ONEBLK): launched when a.n_blocks == 1
extern "C" __global__ __launch_bounds__(64, ASDR_WAVES_PER_EU) void asdr_update_kernel_one(UpdateArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[8 * ASDR_STRIDE];
  asdr_update_body<ASDR_STRIDE, false, false, true, 1, 0, true>(a, lds);
}
// ... the 16-waves-per-CU form (C16, asdr_update_body): 320-float rows = 10,240 B of LDS per wave, <= 128 VGPRs; direct one-block launches
// of an SSB-class settings group without stage taps
#define ASDR_C16_STRIDE 320
extern "C" __global__ __launch_bounds__(64, 4) void asdr_update_kernel_c16(UpdateArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[8 * ASDR_C16_STRIDE];
  asdr_update_body<ASDR_C16_STRIDE, false, false, true, 1, 0, true>(a, lds);
}
// ... and its four-wave workgroup form (MW, asdr_update_body): 32 channels per workgroup, 49,664 B of rows + 1 KB of hand-off scratch -> 3 workgroups
// = 12 waves per CU as before; direct launches (one settings group of consecutive channels) of one block
#define ASDR_MW_WAVES 4
extern "C" __global__ __launch_bounds__(64 * ASDR_MW_WAVES, ASDR_WAVES_PER_EU) void asdr_update_kernel_mw(UpdateArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[ASDR_MW_WAVES * 8 * ASDR_STRIDE + ASDR_MW_WAVES * 8 * 16 + 260 + 12];   // rows | hand-off scratch | sine table | progress words | flags
  asdr_update_body<ASDR_STRIDE, false, false, true, ASDR_MW_WAVES, 0, true>(a, lds);
}
// SAM: 4 waves = 32 channels per workgroup (50,704 B of LDS -> 3 workgroups = 12 waves per CU), general form only
ASDR_KERNEL(asdr_update_kernel_sam, ASDR_SAM_WAVES * 8 * ASDR_STRIDE + 260, ASDR_WAVES_PER_EU, ASDR_STRIDE, false, true, false, ASDR_SAM_WAVES)
#ifndef ASDR_ALS_WAVES_PER_EU
#define ASDR_ALS_WAVES_PER_EU 3   /* LDS allows 9 waves/CU: needs 3 on one SIMD */
#endif
#ifndef ASDR_ALS_LDS_PAD
#define ASDR_ALS_LDS_PAD 0        /* occupancy experiments only: extra floats of LDS per wave (tools/ablate.py) */
#endif
ASDR_KERNEL(asdr_update_kernel_als, 8 * 516 + 260 + ASDR_ALS_LDS_PAD, ASDR_ALS_WAVES_PER_EU, 516, true, true, true, 1)
ASDR_KERNEL(asdr_update_kernel_als_mixed, 8 * 516 + 260 + ASDR_ALS_LDS_PAD, ASDR_ALS_WAVES_PER_EU, 516, true, true, false, 1)
// ALS with a short filter (taps <= 64, delay + taps <= 65: the reference's defaults are 55 and 3) on a channel that is not in SAM
// mode: the filter's rows fit the plain instantiation's 388 floats per channel -> 12 waves per CU
ASDR_KERNEL(asdr_update_kernel_als_small, 8 * ASDR_ALS_STRIDE, ASDR_WAVES_PER_EU, ASDR_ALS_STRIDE, true, false, true, 1)
ASDR_KERNEL(asdr_update_kernel_als_small_mixed, 8 * ASDR_ALS_STRIDE, ASDR_WAVES_PER_EU, ASDR_ALS_STRIDE, true, false, false, 1)
// ... and the loop-free twin of the uniform short-filter kernel (one-block launches: every large batch)
extern "C" __global__ __launch_bounds__(64, ASDR_WAVES_PER_EU) void asdr_update_kernel_als_small_one(UpdateArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[8 * ASDR_ALS_STRIDE];
  asdr_update_body<ASDR_ALS_STRIDE, true, false, true, 1, 0, true>(a, lds);
}

// SAM sub-range as three launches (asdr_launch_update): pre | PLL | post
extern "C" __global__ __launch_bounds__(64, ASDR_WAVES_PER_EU) void asdr_sam_pre_kernel(UpdateArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[8 * ASDR_STRIDE];
  asdr_update_body<ASDR_STRIDE, false, false, false, 1, 4>(a, lds);
}
extern "C" __global__ __launch_bounds__(64, ASDR_WAVES_PER_EU) void asdr_sam_post_kernel(UpdateArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[8 * ASDR_STRIDE];
  asdr_update_body<ASDR_STRIDE, false, false, false, 1, 5>(a, lds);
}
// ... and their forms for waves of one schedule key (scalar mode / enable tests)
// The pre role needs rows W0 and W1 only (input scale, blanker overlays, IF filter, tile transposition: all inside the first 256 words;
// the phase row, the Hilbert history and the AGC overlay belong to the post role): 260-float rows (260 = 65 sixteen-byte slots == 1
// mod 16, like 388: the 8 channels' rows start on different slots) = 8,320 B per wave -> 19 waves per CU by LDS, and at 126 VGPRs
// FOUR waves per SIMD: the first instantiation of the chain at 16 waves per CU (round 4).
#ifndef ASDR_PRE_STRIDE
#define ASDR_PRE_STRIDE 260
#endif
extern "C" __global__ __launch_bounds__(64, 4) void asdr_sam_pre_kernel_uniform(UpdateArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[8 * ASDR_PRE_STRIDE];
  asdr_update_body<ASDR_PRE_STRIDE, false, false, true, 1, 4>(a, lds);
}
#ifndef ASDR_POST_BOUNDS
#define ASDR_POST_BOUNDS ASDR_WAVES_PER_EU
#endif
extern "C" __global__ __launch_bounds__(64, ASDR_POST_BOUNDS) void asdr_sam_post_kernel_uniform(UpdateArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[8 * ASDR_STRIDE];
  asdr_update_body<ASDR_STRIDE, false, false, true, 1, 5>(a, lds);
}
// ... and with the block loop kept (ROLE 8 / 9): the chunked SAM role streams of small banks (asdr_launch_sam_role with n_blocks > 1; two waves
// per SIMD asked for: at most a few hundred waves, and the looped forms want the registers)
extern "C" __global__ __launch_bounds__(64, 2) void asdr_sam_pre_loop_kernel_uniform(UpdateArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[8 * ASDR_STRIDE];
  asdr_update_body<ASDR_STRIDE, false, false, true, 1, 8>(a, lds);
}
extern "C" __global__ __launch_bounds__(64, 2) void asdr_sam_post_loop_kernel_uniform(UpdateArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[8 * ASDR_STRIDE];
  asdr_update_body<ASDR_STRIDE, false, false, true, 1, 9>(a, lds);
}
// ... and the back half of the chain as a role of its own (ROLE 10: mixer .. AGC from the tile the looped pre role left, the post-AGC row into
// the ALS stage): the three-stage ALS role streams
extern "C" __global__ __launch_bounds__(64, 2) void asdr_als_back_loop_kernel(UpdateArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[8 * ASDR_STRIDE];
  asdr_update_body<ASDR_STRIDE, false, false, true, 1, 10>(a, lds);
}
// ... and for SAM channels with a short ALS filter (ASDR_KERNEL_SAM_ALS): the filter is the post role's last stage, on the compact rows
extern "C" __global__ __launch_bounds__(64, ASDR_WAVES_PER_EU) void asdr_sam_post_als_kernel(UpdateArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[8 * ASDR_ALS_STRIDE];
  asdr_update_body<ASDR_ALS_STRIDE, true, false, false, 1, 5>(a, lds);
}
extern "C" __global__ __launch_bounds__(64, ASDR_WAVES_PER_EU) void asdr_sam_post_als_kernel_uniform(UpdateArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[8 * ASDR_ALS_STRIDE];
  asdr_update_body<ASDR_ALS_STRIDE, true, false, true, 1, 5>(a, lds);
}
#ifndef ASDR_PLL_LANES
#define ASDR_PLL_LANES 64   /* channels per wave of the PLL kernel (experiments: 32 = twice the waves, each half empty) */
#endif
// Channels with a short ALS filter (kind ASDR_KERNEL_ALS_SMALL), when there are enough of them: the chain up to the AGC as the PLAIN
// instantiation (ROLE 6: every optimisation of the plain kernel, which the fused ALS instantiations had to give up for registers),
// then the filter as a launch of its own on small LDS rows (asdr_als_kernel below).
extern "C" __global__ __launch_bounds__(64, ASDR_WAVES_PER_EU) void asdr_als_pre_kernel(UpdateArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[8 * ASDR_STRIDE];
  asdr_update_body<ASDR_STRIDE, false, false, true, 1, 6>(a, lds);
}
// ... and with the block loop kept (ROLE 7): the chain launches of the ALS role streams take a chunk of blocks each
// (two waves per SIMD asked for: these launches are small banks' -- at most 512 waves on 1,024 SIMDs -- and the looped form wants the registers)
extern "C" __global__ __launch_bounds__(64, 2) void asdr_als_pre_loop_kernel(UpdateArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[8 * ASDR_STRIDE];
  asdr_update_body<ASDR_STRIDE, false, false, true, 1, 7>(a, lds);
}
// The ALS filter + the output stage of 8 channels per wave, from the ALS input ring (AudioSDR.cpp:324-352, 158-161).  The filter is 33
// epochs of dependent work per block -- a 55-long chain of additions, then the tap update -- and its throughput follows the resident
// waves (profiles/README.md: 9 -> 12 waves per CU gave -22 %): alone, it needs 268 floats of LDS per channel instead of the chain's 388
// -- [out row / kept history (the output sample n overlays buffer index 128 + n - 65, see als_compute) | current block | even taps |
// odd taps | e] -- and a fraction of the registers: 18 waves per CU by LDS.
#define ALS_K_STRIDE 268
#define ALS_K_XB (-63)     /* L[ALS_K_XB + idx] = sample idx of the reference's 256-sample buffer, idx >= 64 (word 1 ..) */
#define ALS_K_AW 196       /* taps: 32 even | 32 odd */
#define ALS_K_SCR 261
#ifndef ASDR_ALS_K_WAVES_PER_EU
#define ASDR_ALS_K_WAVES_PER_EU 3
#endif
extern "C" __global__ __launch_bounds__(64, ASDR_ALS_K_WAVES_PER_EU) void asdr_als_kernel(UpdateArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[8 * ALS_K_STRIDE];
  const int lane = threadIdx.x, c8 = lane >> 3, s8_ = lane & 7;
  const int wave_g = (int)blockIdx.x;
  int4 slot;
  if (a.direct_ch0 >= 0) slot = make_int4(a.direct_ch0 + wave_g * 8 + c8, (int)a.direct_mode, (int)a.direct_flags, 0);
  else slot = *reinterpret_cast<const int4 *>(a.sched + wave_g * 8 + c8);   // {channel, mode, flags, -}
  const int ch_ = slot.x;
  const uint32_t pflags = (uint32_t)slot.z;
  const bool als_en = pflags & ASDR_F_ALS_EN, adaptive = pflags & ASDR_F_ALS_ADAPTIVE, notch = pflags & ASDR_F_ALS_NOTCH, muted = pflags & ASDR_F_MUTED;
#pragma unroll 1
  for (int blk = 0; blk < a.n_blocks; ++blk) {   // (the host issues one launch per block: the ring holds two)
    // per-iteration opaque copies of the lane coordinates: stops LICM from hoisting every per-lane LDS address out of this loop
    int s8 = s8_; asm volatile("" : "+v"(s8));
    int loff = c8 * ALS_K_STRIDE; asm volatile("" : "+v"(loff));
    float *L = lds + loff;
    const int k0 = 16 * s8, kA = 8 * s8, kF = 4 * s8;
    int ch = ch_; asm volatile("" : "+v"(ch));
    const ChanParams *Pp = row_ptr(a.params, (uint32_t)ch * (uint32_t)sizeof(ChanParams));
    const int M = Pp->als_m, D = Pp->als_delay;
    const float lam = Pp->als_lambda;
    const uint32_t as = (a.als_phase + (uint32_t)blk) & 1u;
    const float *gx_cur = a.als_x + (size_t)ch * (2 * ASDR_N) + as * ASDR_N + kF, *gx_prev = a.als_x + (size_t)ch * (2 * ASDR_N) + (as ^ 1u) * ASDR_N + kF;
    if (a.als_stage != nullptr) {   // ALS role streams: this block's and the previous block's rows wait in the stage (ASDR_ALS_STAGE_SLOTS = 32 slots per channel; the ring's slots belong to the chain launches running ahead)
      const uint32_t sc = (a.als_stage_cur + (uint32_t)blk) % ASDR_ALS_STAGE_SLOTS, sp = (a.als_stage_prev + (uint32_t)blk) % ASDR_ALS_STAGE_SLOTS;
      gx_cur = a.als_stage + (size_t)ch * (ASDR_ALS_STAGE_SLOTS * ASDR_N) + sc * ASDR_N + kF; gx_prev = a.als_stage + (size_t)ch * (ASDR_ALS_STAGE_SLOTS * ASDR_N) + sp * ASDR_N + kF;
    }
    float *gw = a.als_w + (size_t)ch * ASDR_N + kF;
    {
      float tx[8], tw[8], tn[16];   // all row loads in flight together
#pragma unroll
      for (int m = 0; m < 2; ++m) { load4(gx_prev + 32 * (m + 2), tx + 4 * m); load4(gw + 32 * m, tw + 4 * m); }
#pragma unroll
      for (int m = 0; m < 4; ++m) load4(gx_cur + 32 * m, tn + 4 * m);
#pragma unroll
      for (int m = 0; m < 2; ++m) {
#pragma unroll
        for (int j = 0; j < 4; ++j) L[ALS_K_XB + 64 + kF + 32 * m + j] = tx[4 * m + j];   // previous block's samples 64..127 (the rows start one word off a 16-byte boundary)
        float *we = L + ALS_K_AW + ((kF + 32 * m) >> 1);   // taps de-interleaved (ALS_TAP)
        *reinterpret_cast<float2 *>(we) = make_float2(tw[4 * m], tw[4 * m + 2]);
        *reinterpret_cast<float2 *>(we + 32) = make_float2(tw[4 * m + 1], tw[4 * m + 3]);
      }
#pragma unroll
      for (int m = 0; m < 4; ++m) {
#pragma unroll
        for (int j = 0; j < 4; ++j) L[ALS_K_XB + 128 + kF + 32 * m + j] = tn[4 * m + j];
      }
    }
    WAVE_SYNC();
    als_compute<true, ALS_K_XB, ALS_K_AW, 32, ALS_K_SCR, 0>(L, als_en, adaptive, notch, M, D, lam, s8, k0);
    WAVE_SYNC();
    // (the rows' addresses are formed again from the channel index here: held across the filter they cost registers it needs)
    int ch2 = ch; asm volatile("" : "+v"(ch2));
    {   // taps back in natural order
      float *gw = a.als_w + (size_t)ch2 * ASDR_N + kF;
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const float *we = L + ALS_K_AW + ((kF + 32 * m) >> 1);
        const float2 e2 = *reinterpret_cast<const float2 *>(we), o2 = *reinterpret_cast<const float2 *>(we + 32);
        const float t[4] = {e2.x, o2.x, e2.y, o2.y};
        store4(gw + 32 * m, t);
      }
    }
    if (a.audio_prev != nullptr) {   // _audioOut as this block leaves it (asdr_device.h audio_prev)
      float *ap = row_ptr(a.audio_prev, (uint32_t)ch2 * 512u + 4u * (uint32_t)kF);
#pragma unroll
      for (int m = 0; m < 4; ++m) { float t[4]; load4(L + kF + 32 * m, t); store4_nt(ap + 32 * m, t); }
    }
    {   // output, AudioSDR.cpp:158-161: float product, x 32767.0 in binary64, truncate, wrap to int16
      const float og = row_ptr(a.params, (uint32_t)ch2 * (uint32_t)sizeof(ChanParams))->output_gain;
      union { int4 v; int16_t s[8]; } ro[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float au[8];
        load8(L + kA + 64 * h, au);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int v = (int)((double)(og * au[j]) * 32767.0);
          ro[h].s[j] = muted ? (int16_t)0 : (int16_t)v;
        }
      }
      int4 *po = reinterpret_cast<int4 *>(a.out + ((size_t)ch2 * a.out_stride + blk) * ASDR_N + kA);
      store_int4_nt(po, ro[0].v); store_int4_nt(po + 8, ro[1].v);
    }
    WAVE_SYNC();
  }
}
// One LANE per channel: lane l of workgroup w runs the PLL of schedule slot 64 w + l on the rows the pre kernel left in its tile.
extern "C" __global__ __launch_bounds__(64) void asdr_sam_pll_kernel(UpdateArgs a) {
  __shared__ float sine[ASDR_SINE_TABLE_LEN];
  const int lane = threadIdx.x;
  for (int i = lane; i < ASDR_SINE_TABLE_LEN; i += 64) sine[i] = c_sine[i];
  __syncthreads();
  if (lane >= ASDR_PLL_LANES) return;
  const int s = (int)blockIdx.x * ASDR_PLL_LANES + lane;
  if (s >= a.n_sched) return;
  const int4 sl = *reinterpret_cast<const int4 *>(a.sched + s);   // {channel, mode, flags, -}
  if (sl.x >= a.n_channels || (uint32_t)sl.y != ASDR_SAMmode) return;
  ChanSmall *Sc = a.small + sl.x;
  bool lk = false;
#pragma unroll 1
  for (int blk = 0; blk < a.n_blocks; ++blk) {   // (one block per launch but in the chunked role streams: tile set (a.sam_set + blk) % a.sam_sets)
  const uint32_t set_k = (a.sam_sets > 1u) ? (a.sam_set + (uint32_t)blk) % a.sam_sets : 0u;
  float *const xt = a.xch_sam + (size_t)set_k * a.sam_set_stride + (size_t)(s >> 3) * (2 * ASDR_N * 8) + (s & 7);
  // (The kernel is bound by its instruction count -- ~120 per sample and lane, 61 k wave instructions per SIMD for C3, as many as
  // the whole C2 chain -- not by these 32-byte accesses: 16-sample trips with the next trip's values requested ahead, and without
  // the stores altogether, changed its 0.18 ms by less than 0.01 / 0.04 ms.)
#ifndef ASDR_PLL_NO_PREFETCH
  lk = pll_loop<false, 4, true>(Sc, a.k, sine, a.k.two_pi_f,
#else
  lk = pll_loop<false, 4>(Sc, a.k, sine, a.k.two_pi_f,
#endif
    [&](int i, float *xr, float *xi) {
#pragma unroll
      for (int u = 0; u < 4; ++u) { const float *e = xt + (size_t)(i + u) * 16; xr[u] = e[0]; xi[u] = e[8]; } },
    [&](int i, const float *xr, const float *xi) {
#pragma unroll
      for (int u = 0; u < 4; ++u) { float *e = xt + (size_t)(i + u) * 16; e[0] = xr[u]; e[8] = xi[u]; } });
  a.sam_lock[(size_t)set_k * a.sam_lock_stride + s] = lk ? 1u : 0u;
  }
  if (((Sc->status & ASDR_S_PLL_LOCKED) != 0u) != lk) {   // (one bit of a word other roles update too; only this kernel writes this one)
    if (lk) atomicOr(&Sc->status, ASDR_S_PLL_LOCKED); else atomicAnd(&Sc->status, ~ASDR_S_PLL_LOCKED);
  }
}

// Streaming pipeline: workgroups [0, W) run role 1, [W, 2W) role 2, [2W, 3W) role 3 of the W channel groups (W = a.stream_waves; the oscillator role is workgroup 0's second wave),
// workgroup 3W the oscillator role; all 3W + 1 workgroups must be resident together (the host launches it only for small batches).  Uniform-key SSB waves only.
// The pipeline's oscillator role (one wave): for every block of the call, the 128 phases that start where the previous block
// ended -- the recurrence of AudioSDR.h:513-518 from the first channel's carried phase -- and their cos / sin pairs, left in
// lo_ring[b % ASDR_LO_RING] under the key (start phase, increment).  A role-2 wave whose channels carry exactly that key reads
// the pairs instead of running the recurrence and the table lookups itself; any other wave computes its own, as ever.
__device__ __forceinline__ void asdr_stream_lo_role(const UpdateArgs &a, float *lds) {
  const int lane = threadIdx.x & 63;
  const ChainConsts K = a.k;
  const float two_pi = K.two_pi_f;
  const int ch0 = (a.direct_ch0 >= 0) ? a.direct_ch0 : a.sched[0].ch;
  const bool am0 = ((a.direct_ch0 >= 0) ? a.direct_mode : a.sched[0].mode) == ASDR_AMmode;
  float phase = am0 ? a.small[ch0].phase_am : a.small[ch0].phase_ssb;
  const float inc = (am0 ? -K.if_center : -a.params[ch0].freq_shift) * K.phase_inc_unit;   // as the role-2 waves form it (AudioSDR.h:508-512, .cpp:86, 134)
  uint32_t *const prog = a.stream_prog, *const err = a.stream_err, *const my = a.stream_prog + 3 * a.stream_waves;
  const bool up = !(inc < 0.0f);
  const float wrapv = up ? -two_pi : two_pi, lim = up ? two_pi : 0.0f;
  const uint32_t flip = up ? 0u : 0x80000000u;
  bool gave_up = false;
#pragma unroll 1
  for (int blk = 0; blk < a.n_blocks; ++blk) {
    if (blk >= ASDR_LO_RING) {   // every role-2 wave has left the entry this block overwrites
      const uint32_t need = (uint32_t)(blk - ASDR_LO_RING + 1);
      uint32_t spins = 0;
      while (true) {
        uint32_t mn = 0xFFFFFFFFu;
        for (int w = lane; w < a.stream_waves; w += 64) { const uint32_t v = __hip_atomic_load(prog + a.stream_waves + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); mn = v < mn ? v : mn; }
        if (__all(mn >= need)) break;
        __builtin_amdgcn_s_sleep(8);
        if (++spins > a.stream_spin_limit) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); gave_up = true; break; }
        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { gave_up = true; break; }
      }
      asm volatile("" ::: "memory");
      if (gave_up) break;
    }
    LoEntry *e = a.lo_ring + (blk % ASDR_LO_RING);
    const float start = phase;
    if (lane == 0) {
#pragma unroll 1
      for (int i = 0; i < ASDR_N; i += 8) {
        float pv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          pv[u] = phase;
          const float t = phase + inc, tw = t + wrapv;
          phase = (__uint_as_float(__float_as_uint(t) ^ flip) > lim) ? tw : t;
        }
        store8(lds + i, pv);
      }
    }
    phase = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(phase)));   // lane 0's end phase, to all
    WAVE_SYNC();
    const float2 p2 = *reinterpret_cast<const float2 *>(lds + 2 * lane);
    v2f c2, s2;
    {
      const float ph2[2] = {p2.x, p2.y};
      float cb[2], sb[2];
      sincos_batch<2>(nullptr, ph2, cb, sb, two_pi, SinIndexK{K.inv_two_pi_d, K.sin_index_scale_d}, K.half_pi_d);
      c2[0] = cb[0]; c2[1] = cb[1]; s2[0] = sb[0]; s2[1] = sb[1];
    }
    asm volatile("global_store_dwordx2 %0, %1, off sc1\n\tglobal_store_dwordx2 %2, %3, off sc1\n\ts_nop 1"
                 :: "v"(e->c + 2 * lane), "v"(c2), "v"(e->s + 2 * lane), "v"(s2) : "memory");
    if (lane == 0) {
      v4f key;
      key[0] = start; key[1] = inc; key[2] = phase; key[3] = 0.0f;
      asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(e), "v"(key) : "memory");
    }
    stream_signal(my, (uint32_t)blk + 1u, lane);
    WAVE_SYNC();
  }
}

extern "C" __global__ __launch_bounds__(128, ASDR_WAVES_PER_EU) void asdr_stream_kernel(UpdateArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[8 * ASDR_STRIDE];
  __shared__ __attribute__((aligned(16))) float lo_scratch[ASDR_N];   // the oscillator role's phase row (workgroup 0's second wave)
  const int role = (int)blockIdx.x / a.stream_waves;   // workgroup-uniform
  // Two waves per workgroup.  The second one shares the Hilbert FIR in a role-2 workgroup, is the launch's oscillator role in
  // workgroup 0 (so that the pipeline is exactly 3 W workgroups: 512 channel groups = 4,096 receivers fill an MI355X's 1,536
  // resident workgroups of this kernel) and is gone at once everywhere else.
  if (threadIdx.x >= 64) {
    if (role == 1) asdr_stream_fir_helper<ASDR_STRIDE>(a, lds);
    else if (blockIdx.x == 0) asdr_stream_lo_role(a, lo_scratch);
    return;
  }
  if (role == 0) asdr_update_body<ASDR_STRIDE, false, false, true, 1, 1>(a, lds);
  else if (role == 1) asdr_update_body<ASDR_STRIDE, false, false, true, 1, 2>(a, lds);
  else asdr_update_body<ASDR_STRIDE, false, false, true, 1, 3>(a, lds);
}

// Round 6: the same pipeline with THREE helper waves in a role-2 workgroup (256 threads per workgroup): the Hilbert FIR in quarters.  The roles
// are balanced at ~15 k cycles per block (profiles/r03_timeline_stream_roles.txt) and role 2 carries 5.8 k of half a FIR; the chip is three
// quarters idle at the bank sizes the pipeline serves best, so the extra waves are free.  Taken by the host while every workgroup has a
// compute unit to itself (3 w <= compute units); larger banks keep the two-wave form, whose workgroups pack six to a compute unit.
extern "C" __global__ __launch_bounds__(256, 1) void asdr_stream_kernel_h3(UpdateArgs a) {   // (no register cap: one workgroup per compute unit is all this form asks for)
  __shared__ __attribute__((aligned(16))) float lds[8 * ASDR_STRIDE];
  __shared__ __attribute__((aligned(16))) float lo_scratch[ASDR_N];
  const int role = (int)blockIdx.x / a.stream_waves;   // workgroup-uniform
  const int hw = (int)threadIdx.x >> 6;
  if (hw != 0) {
    if (role == 1) {
      if (hw == 1) asdr_stream_fir_helper<ASDR_STRIDE, 2, 2>(a, lds);
      else if (hw == 2) asdr_stream_fir_helper<ASDR_STRIDE, 4, 2>(a, lds);
      else asdr_stream_fir_helper<ASDR_STRIDE, 6, 2>(a, lds);
    } else if (blockIdx.x == 0 && hw == 1) asdr_stream_lo_role(a, lo_scratch);
    return;
  }
  if (role == 0) asdr_update_body<ASDR_STRIDE, false, false, true, 1, 1>(a, lds);
  else if (role == 1) asdr_update_body<ASDR_STRIDE, false, false, true, 1, 2, false, 3>(a, lds);
  else asdr_update_body<ASDR_STRIDE, false, false, true, 1, 3>(a, lds);
}

// ---- launch census (round 6): every kernel launch of this file goes through ASDR_LAUNCH, which counts it under the kernel's name -- so that a
// measurement can say WHICH instantiation it timed (bench.py's `roofline.kernel` was a hard-coded string that had gone stale).  The table is the
// list of this file's kernels: tests/test_build_properties.py checks it against the __global__ definitions and the launch sites.
static const char *const k_kernel_names[] = {
  "asdr_als_back_loop_kernel",
  "asdr_als_kernel",
  "asdr_als_pre_kernel",
  "asdr_als_pre_loop_kernel",
  "asdr_als_stage_seed_kernel",
  "asdr_reset_kernel",
  "asdr_sam_pll_kernel",
  "asdr_sam_post_als_kernel",
  "asdr_sam_post_als_kernel_uniform",
  "asdr_sam_post_kernel",
  "asdr_sam_post_kernel_uniform",
  "asdr_sam_post_loop_kernel_uniform",
  "asdr_sam_pre_kernel",
  "asdr_sam_pre_kernel_uniform",
  "asdr_sam_pre_loop_kernel_uniform",
  "asdr_spin_kernel",
  "asdr_stream_ack_kernel",
  "asdr_stream_kernel",
  "asdr_stream_kernel_h3",
  "asdr_stream_restore_kernel",
  "asdr_stream_snapshot_kernel",
  "asdr_update_kernel",
  "asdr_update_kernel_als",
  "asdr_update_kernel_als_mixed",
  "asdr_update_kernel_als_small",
  "asdr_update_kernel_als_small_mixed",
  "asdr_update_kernel_als_small_one",
  "asdr_update_kernel_c16",
  "asdr_update_kernel_mixed",
  "asdr_update_kernel_mw",
  "asdr_update_kernel_one",
  "asdr_update_kernel_sam"
};
#define ASDR_N_KERNEL_NAMES ((int)(sizeof k_kernel_names / sizeof k_kernel_names[0]))
static unsigned long long g_kernel_launches[ASDR_N_KERNEL_NAMES];
static inline int kernel_name_index(const char *name) {
  for (int i = 0; i < ASDR_N_KERNEL_NAMES; ++i)
    if (strcmp(k_kernel_names[i], name) == 0) return i;
  return -1;
}
static inline void count_launch(int i) { if (i >= 0) __atomic_fetch_add(&g_kernel_launches[i], 1ull, __ATOMIC_RELAXED); }
// (the name is looked up once per launch site: a function-local static)
#define ASDR_LAUNCH(kernel, ...) do { static const int site_index_ = kernel_name_index(#kernel); count_launch(site_index_); hipLaunchKernelGGL(kernel, __VA_ARGS__); } while (0)
extern "C" int asdr_kernels_count(void) { return ASDR_N_KERNEL_NAMES; }
extern "C" const char *asdr_kernels_name(int i) { return (i >= 0 && i < ASDR_N_KERNEL_NAMES) ? k_kernel_names[i] : nullptr; }
extern "C" unsigned long long asdr_kernels_launches(int i) { return (i >= 0 && i < ASDR_N_KERNEL_NAMES) ? __atomic_load_n(&g_kernel_launches[i], __ATOMIC_RELAXED) : 0ull; }
extern "C" void asdr_kernels_launches_reset(void) { for (int i = 0; i < ASDR_N_KERNEL_NAMES; ++i) __atomic_store_n(&g_kernel_launches[i], 0ull, __ATOMIC_RELAXED); }

extern "C" int asdr_launch_stream(const UpdateArgs *a, hipStream_t stream, int fir_helpers) {
  if (a->stream_waves <= 0) return 0;
  if (fir_helpers == 3) ASDR_LAUNCH(asdr_stream_kernel_h3, dim3(3 * a->stream_waves), dim3(256), 0, stream, *a);
  else ASDR_LAUNCH(asdr_stream_kernel, dim3(3 * a->stream_waves), dim3(128), 0, stream, *a);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
// How many workgroups of the pipeline kernel the device can hold at once (occupancy x compute units): the pipeline's roles wait
// for each other, so all 3 W + 1 of them must be resident together.  -1 = the runtime could not tell.
extern "C" int asdr_stream_capacity(int device, int *compute_units, int fir_helpers) {
  int per_cu = 0;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return -1;
  if ((fir_helpers == 3 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, asdr_stream_kernel_h3, 256, 0)
                        : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, asdr_stream_kernel, 128, 0)) != hipSuccess) return -1;
  if (compute_units) *compute_units = prop.multiProcessorCount;
  return per_cu * prop.multiProcessorCount;
}

// ---- the pipeline as a transaction ---------------------------------------------------------------------------------------
// A pipeline call advances its channels' state block by block across three roles; if a role's bounded wait ever fires (the
// roles were not co-resident: a shared GPU, masked CUs) the state is left part-way.  So the host brackets the launch:
//   asdr_stream_snapshot_kernel   copies the rows the call will advance (ASDR_SNAP_BYTES per channel) aside
//   asdr_stream_kernel            the pipeline; on a timeout it sets stream_err[0] and every wave leaves
//   asdr_stream_restore_kernel    stream_err[0] != 0: copies the rows back
//   asdr_update_kernel (run_if)   stream_err[0] != 0: the same call on the in-kernel block loop (no inter-workgroup waits)
//   asdr_stream_ack_kernel        stream_err[0] != 0: counts the recovery in stream_err[1], clears stream_err[0]
// all on the caller's stream, so that whatever the caller enqueues next sees exact results either way.
// One workgroup per group of 8 schedule slots; snap = [slot][ASDR_SNAP_BYTES / 16] uint4.
template <bool RESTORE>
__device__ __forceinline__ void stream_snapshot_body(const UpdateArgs &a, uint4 *snap) {
  if (RESTORE && *a.stream_err == 0u) return;
  const int t = threadIdx.x;   // 256 threads
  for (int j = 0; j < 8; ++j) {
    const int slot = (int)blockIdx.x * 8 + j;
    if (slot >= a.n_sched) return;
    const int ch = (a.direct_ch0 >= 0) ? a.direct_ch0 + slot : a.sched[slot].ch;
    if (ch >= a.n_channels) continue;
    uint4 *dst = snap + (size_t)slot * (ASDR_SNAP_BYTES / 16);
    // piece p of the channel's snapshot row: which array, which 16-byte piece of the channel's row in it
    for (int p = t; p < ASDR_SNAP_BYTES / 16; p += 256) {
      uint4 *row; int q = p;
      if (q < 28) row = reinterpret_cast<uint4 *>(a.small + ch);
      else if ((q -= 28) < 96) row = reinterpret_cast<uint4 *>(a.nb_hist + (size_t)ch * 768);
      else if ((q -= 96) < ASDR_NB_MASK_ROW / 16) row = reinterpret_cast<uint4 *>(a.nb_mask + (size_t)ch * ASDR_NB_MASK_ROW);
      else if ((q -= ASDR_NB_MASK_ROW / 16) < 64) row = reinterpret_cast<uint4 *>(a.hil_q + (size_t)ch * 256);
      else if ((q -= 64) < 64) row = reinterpret_cast<uint4 *>(a.hil_i + (size_t)ch * 256);
      else { q -= 64; row = a.audio_prev ? reinterpret_cast<uint4 *>(a.audio_prev + (size_t)ch * 128) : nullptr; }
      if (row == nullptr) continue;
      if (RESTORE) row[q] = dst[p]; else dst[p] = row[q];
    }
  }
}
extern "C" __global__ __launch_bounds__(256) void asdr_stream_snapshot_kernel(UpdateArgs a, uint4 *snap) { stream_snapshot_body<false>(a, snap); }
extern "C" __global__ __launch_bounds__(256) void asdr_stream_restore_kernel(UpdateArgs a, uint4 *snap) { stream_snapshot_body<true>(a, snap); }
extern "C" __global__ void asdr_stream_ack_kernel(uint32_t *err) {
  if (threadIdx.x == 0 && blockIdx.x == 0 && err[0] != 0u) { err[1] += 1u; err[0] = 0u; }
}
extern "C" int asdr_launch_stream_snapshot(const UpdateArgs *a, void *snap, int restore, hipStream_t stream) {
  const int groups = a->n_sched / 8;
  if (groups <= 0) return 0;
  if (restore) ASDR_LAUNCH(asdr_stream_restore_kernel, dim3(groups), dim3(256), 0, stream, *a, reinterpret_cast<uint4 *>(snap));
  else ASDR_LAUNCH(asdr_stream_snapshot_kernel, dim3(groups), dim3(256), 0, stream, *a, reinterpret_cast<uint4 *>(snap));
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
extern "C" int asdr_launch_stream_ack(uint32_t *err, hipStream_t stream) {
  ASDR_LAUNCH(asdr_stream_ack_kernel, dim3(1), dim3(64), 0, stream, err);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ---- state (re-)initialisation kernel: applies ChanParams.reset bits, one thread per (channel, word) ------
extern "C" __global__ void asdr_reset_kernel(UpdateArgs a, const uint32_t *reset_bits, int first_row, int n_rows) {
  if ((int)blockIdx.x >= n_rows) return;
  const int ch = first_row + blockIdx.x;
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
    if (a.audio_prev != nullptr) a.audio_prev[(size_t)ch * 128 + t] = 0.0f;   // _audioOut: static storage, zero before the first block
  }
  __syncthreads();
  if (all && t == 0) { S->nb_avg = 10.0f; S->status = ASDR_S_AGC_ACTIVE; }   // AudioSDR.h:242, :230
  if (all || (r & ASDR_R_IF)) { if (t < 32) (&S->if_state[0][0])[t] = 0.0f; }
  if (all || (r & ASDR_R_IMG)) { if (t < 32) (&S->img_state[0][0])[t] = 0.0f; }
  if (all || (r & ASDR_R_AF)) { if (t < 16) S->af_state[t] = 0.0f; }
  if (all || (r & ASDR_R_NB)) {
    for (int i = t; i < 768; i += 128) a.nb_hist[(size_t)ch * 768 + i] = 0;
    for (int i = t; i < ASDR_NB_MASK_ROW; i += 128) a.nb_mask[(size_t)ch * ASDR_NB_MASK_ROW + i] = 1;   // code 1 == 1.0
    if (t < 6) (&S->nb_gain[0][0])[t] = 0.0f;
  }
  if (all || (r & ASDR_R_ALS)) {
    a.als_x[(size_t)ch * 256 + t] = 0.0f; a.als_x[(size_t)ch * 256 + 128 + t] = 0.0f;
    a.als_w[(size_t)ch * 128 + t] = 0.0f;
  }
}

// `variant`: ASDR_KERNEL_PLAIN / _SAM / _ALS / _ALS_SMALL / _SAM_ALS; `uniform`: every wave of the sub-range holds 8 real channels with one schedule key
extern "C" int asdr_launch_update(const UpdateArgs *a, int variant, int uniform, hipStream_t stream) {
  const int n_waves = a->n_sched / 8;
  if (n_waves <= 0) return 0;
  if (variant == ASDR_KERNEL_ALS || (variant == ASDR_KERNEL_SAM_ALS && a->xch_sam == nullptr)) { if (uniform) ASDR_LAUNCH(asdr_update_kernel_als, dim3(n_waves), dim3(64), 0, stream, *a); else ASDR_LAUNCH(asdr_update_kernel_als_mixed, dim3(n_waves), dim3(64), 0, stream, *a); }
  else if (variant == ASDR_KERNEL_ALS_SMALL && uniform == 2) {   // chain up to the AGC | the filter + output, two launches (one block per call: the host loops)
    ASDR_LAUNCH(asdr_als_pre_kernel, dim3(n_waves), dim3(64), 0, stream, *a);
    ASDR_LAUNCH(asdr_als_kernel, dim3(n_waves), dim3(64), 0, stream, *a);
  }
  else if (variant == ASDR_KERNEL_ALS_SMALL) { if (uniform && ASDR_ONEBLK && ASDR_ONEBLK_ALS && a->n_blocks == 1 && a->run_if == nullptr) ASDR_LAUNCH(asdr_update_kernel_als_small_one, dim3(n_waves), dim3(64), 0, stream, *a); else if (uniform) ASDR_LAUNCH(asdr_update_kernel_als_small, dim3(n_waves), dim3(64), 0, stream, *a); else ASDR_LAUNCH(asdr_update_kernel_als_small_mixed, dim3(n_waves), dim3(64), 0, stream, *a); }
  else if ((variant == ASDR_KERNEL_SAM || variant == ASDR_KERNEL_SAM_ALS) && a->xch_sam != nullptr) {   // pre | PLL | post (one block per call: the host loops)
    if (uniform) ASDR_LAUNCH(asdr_sam_pre_kernel_uniform, dim3(n_waves), dim3(64), 0, stream, *a);
    else ASDR_LAUNCH(asdr_sam_pre_kernel, dim3(n_waves), dim3(64), 0, stream, *a);
    ASDR_LAUNCH(asdr_sam_pll_kernel, dim3((a->n_sched + ASDR_PLL_LANES - 1) / ASDR_PLL_LANES), dim3(64), 0, stream, *a);
    if (variant == ASDR_KERNEL_SAM_ALS) {
      if (uniform) ASDR_LAUNCH(asdr_sam_post_als_kernel_uniform, dim3(n_waves), dim3(64), 0, stream, *a);
      else ASDR_LAUNCH(asdr_sam_post_als_kernel, dim3(n_waves), dim3(64), 0, stream, *a);
    } else {
      if (uniform) ASDR_LAUNCH(asdr_sam_post_kernel_uniform, dim3(n_waves), dim3(64), 0, stream, *a);
      else ASDR_LAUNCH(asdr_sam_post_kernel, dim3(n_waves), dim3(64), 0, stream, *a);
    }
  }
  else if (variant == ASDR_KERNEL_SAM) ASDR_LAUNCH(asdr_update_kernel_sam, dim3((n_waves + ASDR_SAM_WAVES - 1) / ASDR_SAM_WAVES), dim3(64 * ASDR_SAM_WAVES), 0, stream, *a);
  else {
    // the four-wave workgroup form: large direct one-block launches (ASDR_MW=0 / ASDR_MW_MIN_WAVES=<n> in the environment: measurements, tests)
    static int mw_on = -1, mw_min = ASDR_MW_MIN_WAVES, c16_on = ASDR_C16, c16_min = ASDR_C16_MIN_WAVES, c16_pad = 0;
    if (mw_on < 0) {
      const char *e = getenv("ASDR_MW"); mw_on = e ? atoi(e) : ASDR_MW; const char *m = getenv("ASDR_MW_MIN_WAVES"); if (m) mw_min = atoi(m);
      const char *c = getenv("ASDR_C16"); if (c) c16_on = atoi(c); const char *cm = getenv("ASDR_C16_MIN_WAVES"); if (cm) c16_min = atoi(cm);
      const char *cp = getenv("ASDR_C16_PAD"); if (cp) c16_pad = atoi(cp);
    }
    const uint32_t dm = a->direct_mode;
    const bool ssb_class = dm == ASDR_USBmode || dm == ASDR_LSBmode || dm == ASDR_CW_USBmode || dm == ASDR_CW_LSBmode || dm == ASDR_WSPRmode;
    if (uniform && c16_on && a->n_blocks == 1 && a->run_if == nullptr && a->direct_ch0 >= 0 && a->taps == nullptr && ssb_class && n_waves >= c16_min)
      ASDR_LAUNCH(asdr_update_kernel_c16, dim3(n_waves), dim3(64), (size_t)c16_pad, stream, *a);   // (c16_pad: dynamic LDS that takes the occupancy back, measurements)
    else
    if (uniform && mw_on && a->n_blocks == 1 && a->run_if == nullptr && a->direct_ch0 >= 0 && n_waves >= mw_min)
      ASDR_LAUNCH(asdr_update_kernel_mw, dim3((n_waves + ASDR_MW_WAVES - 1) / ASDR_MW_WAVES), dim3(64 * ASDR_MW_WAVES), 0, stream, *a);
    else
    if (uniform && ASDR_ONEBLK && a->n_blocks == 1 && a->run_if == nullptr) ASDR_LAUNCH(asdr_update_kernel_one, dim3(n_waves), dim3(64), 0, stream, *a);
    else if (uniform) ASDR_LAUNCH(asdr_update_kernel, dim3(n_waves), dim3(64), 0, stream, *a);
    else ASDR_LAUNCH(asdr_update_kernel_mixed, dim3(n_waves), dim3(64), 0, stream, *a);
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ALS role streams: the two launches of the chain | filter form on their own (role 0 = the chain up to the AGC, 1 = the filter + output),
// and the kernel that seeds the stage's "previous block" slot from the als_x ring in front of a call's first block.
extern "C" __global__ void asdr_als_stage_seed_kernel(UpdateArgs a, int ch0, int n) {
  const int c = (int)blockIdx.x, t = (int)threadIdx.x;   // one workgroup of 128 threads per channel ch0 + c
  if (c >= n) return;
  const size_t ch = (size_t)(ch0 + c);
  a.als_stage[ch * (ASDR_ALS_STAGE_SLOTS * ASDR_N) + a.als_stage_prev * ASDR_N + t] = a.als_x[ch * (2 * ASDR_N) + ((a.als_phase & 1u) ^ 1u) * ASDR_N + t];
}
extern "C" int asdr_launch_als_role(const UpdateArgs *a, int role, hipStream_t stream) {
  const int n_waves = a->n_sched / 8;
  if (n_waves <= 0) return 0;
  if (role == 2) ASDR_LAUNCH(asdr_sam_pre_loop_kernel_uniform, dim3(n_waves), dim3(64), 0, stream, *a);      // three stages: the front half (scale, blanker, IF) -> tiles
  else if (role == 3) ASDR_LAUNCH(asdr_als_back_loop_kernel, dim3(n_waves), dim3(64), 0, stream, *a);       // ... the back half (mixer .. AGC) -> stage
  else if (role == 0 && a->n_blocks > 1) ASDR_LAUNCH(asdr_als_pre_loop_kernel, dim3(n_waves), dim3(64), 0, stream, *a);
  else if (role == 0) ASDR_LAUNCH(asdr_als_pre_kernel, dim3(n_waves), dim3(64), 0, stream, *a);
  else ASDR_LAUNCH(asdr_als_kernel, dim3(n_waves), dim3(64), 0, stream, *a);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
extern "C" int asdr_launch_als_stage_seed(const UpdateArgs *a, int ch0, int n, hipStream_t stream) {
  if (n <= 0) return 0;
  ASDR_LAUNCH(asdr_als_stage_seed_kernel, dim3(n), dim3(ASDR_N), 0, stream, *a, ch0, n);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// One role of the three-launch SAM form on its own stream (role 0 = pre, 1 = PLL, 2 = post): the host overlaps the roles of
// consecutive blocks of a multi-block call (asdr_host.cpp, SAM role streams).
extern "C" int asdr_launch_sam_role(const UpdateArgs *a, int variant, int uniform, int role, hipStream_t stream) {
  const int n_waves = a->n_sched / 8;
  if (n_waves <= 0) return 0;
  // The chunked form is named by the TILE SETS, not by the block count: a call's last chunk may hold ONE block (n_blocks % 8 == 1), and the
  // one-block pre / post kernels know nothing of tile sets (they work on set 0 while the PLL kernel reads set sam_set: round 5's 17- and
  // 25-block calls ended with a wrong last block).  The looped kernels run a one-block chunk like any other.
  if (a->n_blocks > 1 || a->sam_sets > 1u) {   // uniform SAM sub-ranges only (asdr_host.cpp)
    if (!uniform || variant != ASDR_KERNEL_SAM) return -1;
    if (role == 0) ASDR_LAUNCH(asdr_sam_pre_loop_kernel_uniform, dim3(n_waves), dim3(64), 0, stream, *a);
    else if (role == 1) ASDR_LAUNCH(asdr_sam_pll_kernel, dim3((a->n_sched + ASDR_PLL_LANES - 1) / ASDR_PLL_LANES), dim3(64), 0, stream, *a);
    else ASDR_LAUNCH(asdr_sam_post_loop_kernel_uniform, dim3(n_waves), dim3(64), 0, stream, *a);
    return hipGetLastError() == hipSuccess ? 0 : -1;
  }
  if (role == 0) {
    if (uniform) ASDR_LAUNCH(asdr_sam_pre_kernel_uniform, dim3(n_waves), dim3(64), 0, stream, *a);
    else ASDR_LAUNCH(asdr_sam_pre_kernel, dim3(n_waves), dim3(64), 0, stream, *a);
  } else if (role == 1) {
    ASDR_LAUNCH(asdr_sam_pll_kernel, dim3((a->n_sched + ASDR_PLL_LANES - 1) / ASDR_PLL_LANES), dim3(64), 0, stream, *a);
  } else if (variant == ASDR_KERNEL_SAM_ALS) {
    if (uniform) ASDR_LAUNCH(asdr_sam_post_als_kernel_uniform, dim3(n_waves), dim3(64), 0, stream, *a);
    else ASDR_LAUNCH(asdr_sam_post_als_kernel, dim3(n_waves), dim3(64), 0, stream, *a);
  } else {
    if (uniform) ASDR_LAUNCH(asdr_sam_post_kernel_uniform, dim3(n_waves), dim3(64), 0, stream, *a);
    else ASDR_LAUNCH(asdr_sam_post_kernel, dim3(n_waves), dim3(64), 0, stream, *a);
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// A kernel that does nothing for `ticks` ticks of the 100 MHz real-time counter (one wave): the host's probe that two streams of the pool
// really run concurrently (asdr_host.cpp lanes_overlap_probe).  Bounded: the caller passes a few thousand ticks.
extern "C" __global__ void asdr_spin_kernel(unsigned long long ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
extern "C" int asdr_launch_spin(unsigned long long ticks, hipStream_t stream) {
  ASDR_LAUNCH(asdr_spin_kernel, dim3(1), dim3(64), 0, stream, ticks);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

extern "C" int asdr_launch_reset(const UpdateArgs *a, const uint32_t *d_reset_bits, int first_row, int n_rows, hipStream_t stream) {
  ASDR_LAUNCH(asdr_reset_kernel, dim3(n_rows), dim3(128), 0, stream, *a, d_reset_bits, first_row, n_rows);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
