// Micro-benchmark (round 6): what ONE wave gets through per cycle when it shares its SIMD with W - 1 others running the same code, W = 1..4,
// for the instruction mixes of asdr_update_kernel_mw's phases -- the "per-instruction latency at 3 waves per SIMD" of the latency model
// (tools/latency_model.py, bench.py `roofline_latency`).  Every CU of the chip runs 4 W one-wave workgroups (dynamic LDS sized so that exactly
// 4 W fit: W per SIMD); each wave times its own loop with s_memtime (shader clock) and the host takes the median over all waves.
// Mixes (instruction counts per trip from the compiled loop; the .s is the reference):
//   chain2     one dependent chain mul, add                           (blanker average: 2 dependent instructions per sample)
//   chain4x    four independent chains mul, add                       (what in-wave parallelism buys)
//   biquad     one step of biquad_pipe as shipped: 8 v_cndmask_b32_dpp on VCC, 40 products / sums off the chain, 32 on it, 2 ds_read_b128, 2 ds_write_b128
//   fir        one trip of the packed folded Hilbert FIR (asdr_fir.h hilbert_fir_rows<0, 8>): 192 v_pk_* + 30 ds_read_b128 per trip
//   pointwise  the envelope pass' shape: convert, scale (2 fma-class), squares, sum, short division, select -- independent per sample
//   agc        the AGC's attack chunk: compare, mul, mul, add, three selects per sample, dependent
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -I audiosdr_amd/csrc tools/ubench/issue_rate.hip -o tools/ubench/issue_rate && tools/ubench/issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#include "asdr_fir.h"

#define TRIPS 256
__constant__ float c_taps[64];

template <int MIX>
__global__ __launch_bounds__(64) void k(float *out, unsigned long long *cyc, float a, float b, float c) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x;
  for (int i = lane; i < 2400; i += 64) lds[i] = 1e-3f * (float)(i & 63);
  __syncthreads();
  float x = lane * 1e-3f, y = x + 1.0f, z = y + 1.0f, w = z + 1.0f;
  float acc = 0.0f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (MIX == 0) {
#pragma unroll 1
    for (int t = 0; t < TRIPS; ++t) {
#pragma unroll
      for (int u = 0; u < 16; ++u) { x = x * a; x = x + b; }
    }
  } else if (MIX == 1) {
#pragma unroll 1
    for (int t = 0; t < TRIPS; ++t) {
#pragma unroll
      for (int u = 0; u < 4; ++u) { x = x * a; y = y * a; z = z * a; w = w * a; x = x + b; y = y + b; z = z + b; w = w + b; }
    }
  } else if (MIX >= 9 && MIX <= 16) {
    // what makes a dense stream of plain FP32 operations share a SIMD badly?  9: the constants in VGPRs; 10: an s_nop behind every four; 11: eight chains;
    // 12: the second operand of every addition a VGPR of another chain (the biquad's shape); 13: a four times longer loop body
    float va = a, vb = b;
    asm volatile("" : "+v"(va), "+v"(vb));
    float x4 = w + 1.0f, x5 = w + 2.0f, x6 = w + 3.0f, x7 = w + 4.0f;
#pragma unroll 1
    for (int t = 0; t < TRIPS / (MIX == 13 ? 4 : 1); ++t) {
#pragma unroll
      for (int u = 0; u < (MIX == 13 ? 16 : 4); ++u) {
        if (MIX == 9) { x = x * va; y = y * va; z = z * va; w = w * va; x = x + vb; y = y + vb; z = z + vb; w = w + vb; }
        else if (MIX == 10) { x = x * a; y = y * a; z = z * a; w = w * a; asm volatile("s_nop 0"); x = x + b; y = y + b; z = z + b; w = w + b; asm volatile("s_nop 0"); }
        else if (MIX == 11) { if (u & 1) { x = x * a; y = y * a; z = z * a; w = w * a; x4 = x4 * a; x5 = x5 * a; x6 = x6 * a; x7 = x7 * a; } else { x = x + b; y = y + b; z = z + b; w = w + b; x4 = x4 + b; x5 = x5 + b; x6 = x6 + b; x7 = x7 + b; } }
        else if (MIX == 12) { x = x * a; y = y * a; z = z * a; w = w * a; x = x + w; y = y + x; z = z + y; w = w + z; }
        else if (MIX == 14) { x = x * 0.999f; y = y * 0.999f; z = z * 0.999f; w = w * 0.999f; x = x + 1e-3f; y = y + 1e-3f; z = z + 1e-3f; w = w + 1e-3f; }   // 32-bit literals in the instruction
        else if (MIX == 15) { x = x * 0.5f; y = y * 0.5f; z = z * 0.5f; w = w * 0.5f; x = x + 1.0f; y = y + 1.0f; z = z + 1.0f; w = w + 1.0f; }               // inline constants
        else if (MIX == 16) { x = x * a; y = y * va; z = z * va; w = w * va; x = x + vb; y = y + vb; z = z + vb; w = w + vb; }                             // one instruction in eight reads an SGPR
        else { x = x * a; y = y * a; z = z * a; w = w * a; x = x + b; y = y + b; z = z + b; w = w + b; }
      }
    }
    acc = x4 + x5 + x6 + x7;
  } else if (MIX == 2 || MIX == 6 || MIX == 7) {
    // one pipeline step per trip (the shipped form: assembly block for the select, products off the chain, the recurrence, LDS chunk in / out)
    float *row = lds + (lane >> 3) * 260 + ((lane >> 2) & 1) * 128;   // the update kernel's rows: channel rows 4 banks apart, I / Q rows 128 floats apart (stride 260 here, 388 there: the same banks)
    float yo[8], xn[8];
    float x1 = x, x2 = y, y1 = z, y2 = w;
    const float b0 = a, b1 = b, b2 = c, a1 = 0.5f * a, a2 = -0.25f * a;
    const unsigned long long s0m = __ballot((lane & 3) == 0);
#pragma unroll
    for (int j = 0; j < 8; ++j) { yo[j] = 0.0f; xn[j] = row[j]; }
#pragma unroll 1
    for (int t = 0; t < TRIPS; ++t) {
      float xv[8], p[8], yv[8];
      if (MIX == 6) {
#pragma unroll
        for (int j = 0; j < 8; ++j) xv[j] = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(xn[j]), __float_as_int(yo[j]), 0x114, 0xF, 0xE, false));
      } else if (MIX == 7) {
#pragma unroll
        for (int j = 0; j < 8; ++j) xv[j] = xn[j] + yo[j] * 0.0f;
      } else
      asm("s_mov_b64 vcc, %[m]\n\ts_nop 1\n\t"
          "v_cndmask_b32_dpp %[x0], %[y0], %[n0], vcc row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
          "v_cndmask_b32_dpp %[x1], %[y1], %[n1], vcc row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
          "v_cndmask_b32_dpp %[x2], %[y2], %[n2], vcc row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
          "v_cndmask_b32_dpp %[x3], %[y3], %[n3], vcc row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
          "v_cndmask_b32_dpp %[x4], %[y4], %[n4], vcc row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
          "v_cndmask_b32_dpp %[x5], %[y5], %[n5], vcc row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
          "v_cndmask_b32_dpp %[x6], %[y6], %[n6], vcc row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
          "v_cndmask_b32_dpp %[x7], %[y7], %[n7], vcc row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
          : [x0] "=&v"(xv[0]), [x1] "=&v"(xv[1]), [x2] "=&v"(xv[2]), [x3] "=&v"(xv[3]), [x4] "=&v"(xv[4]), [x5] "=&v"(xv[5]), [x6] "=&v"(xv[6]), [x7] "=&v"(xv[7])
          : [y0] "v"(yo[0]), [y1] "v"(yo[1]), [y2] "v"(yo[2]), [y3] "v"(yo[3]), [y4] "v"(yo[4]), [y5] "v"(yo[5]), [y6] "v"(yo[6]), [y7] "v"(yo[7]),
            [n0] "v"(xn[0]), [n1] "v"(xn[1]), [n2] "v"(xn[2]), [n3] "v"(xn[3]), [n4] "v"(xn[4]), [n5] "v"(xn[5]), [n6] "v"(xn[6]), [n7] "v"(xn[7]),
            [m] "s"(s0m)
          : "vcc");
      { const int nc = (t + 1) & 15;
        const float4 q0 = reinterpret_cast<const float4 *>(row)[2 * nc], q1 = reinterpret_cast<const float4 *>(row)[2 * nc + 1];
        xn[0] = q0.x; xn[1] = q0.y; xn[2] = q0.z; xn[3] = q0.w; xn[4] = q1.x; xn[5] = q1.y; xn[6] = q1.z; xn[7] = q1.w; }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xm1 = (j >= 1) ? xv[j - 1] : x1, xm2 = (j >= 2) ? xv[j - 2] : ((j == 1) ? x1 : x2);
        float s = b0 * xv[j]; s += b1 * xm1; s += b2 * xm2;
        p[j] = s;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float ym1 = (j >= 1) ? yv[j - 1] : y1, ym2 = (j >= 2) ? yv[j - 2] : ((j == 1) ? y1 : y2);
        float s = p[j] + a1 * ym1; s += a2 * ym2;
        yv[j] = s;
      }
      x1 = xv[7]; x2 = xv[6]; y1 = yv[7]; y2 = yv[6];
#pragma unroll
      for (int j = 0; j < 8; ++j) yo[j] = yv[j];
      if ((lane & 3) == 3) {
        reinterpret_cast<float4 *>(row)[2 * (t & 15)] = make_float4(yv[0], yv[1], yv[2], yv[3]);
        reinterpret_cast<float4 *>(row)[2 * (t & 15) + 1] = make_float4(yv[4], yv[5], yv[6], yv[7]);
      }
    }
    acc = x1 + x2 + y1 + y2;
  } else if (MIX == 3) {
    float *L = lds + (lane >> 3) * 388;
    v2f a2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) a2[e] = (v2f){0.0f, 0.0f};
#pragma unroll 1
    for (int t = 0; t < TRIPS / 8; ++t) hilbert_fir_rows<0, 8>(L, 8 * (lane & 7), a2, c_taps);   // 8 trips inside
#pragma unroll
    for (int e = 0; e < 8; ++e) acc += a2[e][0] + a2[e][1];
  } else if (MIX == 4) {
    const int16_t *raw = reinterpret_cast<const int16_t *>(lds) + 16 * lane;
#pragma unroll 1
    for (int t = 0; t < TRIPS; ++t) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xi = (float)raw[(j + t) & 15], xq = (float)raw[(j + t + 5) & 15];
        const float si = __builtin_fmaf(xi, 0x1.0002p-15f, xi * 0x1.0002p-45f), sq = __builtin_fmaf(xq, 0x1.0002p-15f, xq * 0x1.0002p-45f);
        const float pw = si * si + sq * sq;
        uint32_t i = __float_as_uint(pw); i -= 1u << 23; i >>= 1; i += 1u << 29;
        const float o = __uint_as_float(i);
        const float r0 = __builtin_amdgcn_rcpf(o);
        const float r = __builtin_fmaf(__builtin_fmaf(-o, r0, 1.0f), r0, r0);
        const float q0 = pw * r;
        const float e = __builtin_fmaf(-o, q0, pw);
        const float q1 = __builtin_fmaf(e, r, q0);
        acc += 0.5f * (o + q1) * b;
      }
    }
  } else if (MIX == 8) {
    // the AGC's attack chunk without a lane mask: for non-negative operands av > old <=> bits(old) - bits(av) < 0; the select is a v_bfi_b32
    float old_abs = x, gv = -1.0f; uint32_t hcb = 100u;
    const float *row = lds + (lane & 31) * 68;   // (rows 4 banks apart, like the kernel's 388-float rows)
#pragma unroll 1
    for (int t = 0; t < TRIPS; ++t) {
      float av8[8], pb8[8];
      const float4 q0 = reinterpret_cast<const float4 *>(row)[2 * (t & 7)], q1 = reinterpret_cast<const float4 *>(row)[2 * (t & 7) + 1];
      av8[0] = q0.x; av8[1] = q0.y; av8[2] = q0.z; av8[3] = q0.w; av8[4] = q1.x; av8[5] = q1.y; av8[6] = q1.z; av8[7] = q1.w;
#pragma unroll
      for (int u = 0; u < 8; ++u) pb8[u] = b * av8[u];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const uint32_t m = (uint32_t)((int32_t)(__float_as_uint(old_abs) - __float_as_uint(av8[u])) >> 31);
        const float pa = a * old_abs;
        const float v_new = pa + pb8[u];
        uint32_t o, g, h;
        asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(o) : "v"(m), "v"(__float_as_uint(v_new)), "v"(__float_as_uint(old_abs)));
        asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(g) : "v"(m), "v"(__float_as_uint(v_new)), "v"(__float_as_uint(gv)));
        asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(h) : "v"(m), "v"(4410u + (uint32_t)(u + 1)), "v"(hcb));
        old_abs = __uint_as_float(o); gv = __uint_as_float(g); hcb = h;
      }
      acc += gv;
    }
    acc += old_abs + (float)hcb;
  } else {
    float old_abs = x, gv = -1.0f; uint32_t hcb = 100u;
    const float *row = lds + (lane & 31) * 68;   // (rows 4 banks apart, like the kernel's 388-float rows)
#pragma unroll 1
    for (int t = 0; t < TRIPS; ++t) {
      float av8[8];
      const float4 q0 = reinterpret_cast<const float4 *>(row)[2 * (t & 7)], q1 = reinterpret_cast<const float4 *>(row)[2 * (t & 7) + 1];
      av8[0] = q0.x; av8[1] = q0.y; av8[2] = q0.z; av8[3] = q0.w; av8[4] = q1.x; av8[5] = q1.y; av8[6] = q1.z; av8[7] = q1.w;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float av = av8[u];
        const bool att = av > old_abs;
        const float pa = a * old_abs, pb = b * av;
        const float v_new = pa + pb;
        old_abs = att ? v_new : old_abs;
        gv = att ? v_new : gv;
        hcb = att ? 4410u + (uint32_t)(u + 1) : hcb;
      }
      acc += gv;
    }
    acc += old_abs + (float)hcb;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 64 + lane] = x + y + z + w + acc;
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MIX>
static double run(int W, float *d_out, unsigned long long *d_cyc, int n_cu) {
  const int wgs = n_cu * 4 * W;
  // exactly 4 W one-wave workgroups per CU: LDS per workgroup just above 160 KB / (4 W + 1)
  size_t lds = (size_t)(160 * 1024) / (4 * W) - 512;
  if (lds > 64 * 1024) lds = 64 * 1024;                   // (a workgroup may not ask for more than 64 KB: W = 1 then relies on the grid size alone -- one workgroup per SIMD slot is what the dispatcher does with 4 per CU)
  if (lds < 2400 * 4) lds = 2400 * 4;
  hipFuncSetAttribute(reinterpret_cast<const void *>(k<MIX>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  std::vector<unsigned long long> h(wgs);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(k<MIX>, dim3(wgs), dim3(64), lds, 0, d_out, d_cyc, 0.999f, 1e-3f, 0.37f);
    hipDeviceSynchronize();
  }
  hipMemcpy(h.data(), d_cyc, wgs * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  return (double)h[wgs / 2];
}

int main() {
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int n_cu = prop.multiProcessorCount;
  float taps[64]; for (int i = 0; i < 64; ++i) taps[i] = 0.01f * (float)(i + 1);
  hipMemcpyToSymbol(HIP_SYMBOL(c_taps), taps, sizeof taps);
  float *d_out; unsigned long long *d_cyc;
  hipMalloc(&d_out, (size_t)n_cu * 16 * 64 * sizeof(float)); hipMalloc(&d_cyc, (size_t)n_cu * 16 * sizeof(unsigned long long));
  // instructions per trip of each mix (VALU + LDS + the loop's scalar instructions), read off the compiled loops (issue_rate.s)
  const char *names[] = {"chain2", "chain4x", "biquad", "fir", "pointwise", "agc", "biquad_bank_masked_mov_dpp", "biquad_no_select", "agc_mask_free",
                         "chain4x_vgpr_constants", "chain4x_s_nop_every_4", "chain8x", "chain4x_vgpr_second_operand", "chain4x_long_body",
                         "chain4x_literal_constants", "chain4x_inline_constants", "chain4x_one_sgpr_read_in_eight"};
  double per_trip[] = {35, 35, 103, 226, 218, 70, 110, 109, 75, 35, 43, 35, 35, 131 / 4.0, 35, 35, 35};   // (llvm-objdump / -S of this file: the loops' instruction counts, waits and scalar loop control included)
  printf("{\"device\": \"%s\", \"cus\": %d, \"trips\": %d, \"mixes\": {\n", prop.name, n_cu, TRIPS);
  for (int m = 0; m < 17; ++m) {
    double cyc[5] = {0, 0, 0, 0, 0};
    for (int W = 1; W <= 4; ++W) {
      double c = 0;
      switch (m) {
        case 0: c = run<0>(W, d_out, d_cyc, n_cu); break; case 1: c = run<1>(W, d_out, d_cyc, n_cu); break;
        case 2: c = run<2>(W, d_out, d_cyc, n_cu); break; case 3: c = run<3>(W, d_out, d_cyc, n_cu); break;
        case 4: c = run<4>(W, d_out, d_cyc, n_cu); break; case 5: c = run<5>(W, d_out, d_cyc, n_cu); break;
        case 6: c = run<6>(W, d_out, d_cyc, n_cu); break; case 7: c = run<7>(W, d_out, d_cyc, n_cu); break; case 8: c = run<8>(W, d_out, d_cyc, n_cu); break;
        case 9: c = run<9>(W, d_out, d_cyc, n_cu); break; case 10: c = run<10>(W, d_out, d_cyc, n_cu); break; case 11: c = run<11>(W, d_out, d_cyc, n_cu); break;
        case 12: c = run<12>(W, d_out, d_cyc, n_cu); break; case 13: c = run<13>(W, d_out, d_cyc, n_cu); break;
        case 14: c = run<14>(W, d_out, d_cyc, n_cu); break; case 15: c = run<15>(W, d_out, d_cyc, n_cu); break; default: c = run<16>(W, d_out, d_cyc, n_cu); break;
      }
      cyc[W] = c / TRIPS;
    }
    printf("  \"%s\": {\"instructions_per_trip\": %.0f, \"cycles_per_trip_by_waves_per_simd\": [%.1f, %.1f, %.1f, %.1f], "
           "\"cycles_per_instruction_of_a_wave\": [%.2f, %.2f, %.2f, %.2f], \"simd_instructions_per_cycle\": [%.3f, %.3f, %.3f, %.3f]}%s\n",
           names[m], per_trip[m], cyc[1], cyc[2], cyc[3], cyc[4],
           cyc[1] / per_trip[m], cyc[2] / per_trip[m], cyc[3] / per_trip[m], cyc[4] / per_trip[m],
           1 * per_trip[m] / cyc[1], 2 * per_trip[m] / cyc[2], 3 * per_trip[m] / cyc[3], 4 * per_trip[m] / cyc[4], m < 16 ? "," : "");
  }
  printf("}}\n");
  return 0;
}
