// Micro-benchmark: issue cost of single vector instructions on one MI355X SIMD, in shader cycles (s_memtime), for a lone wave and for 8 waves per
// SIMD -- the price list behind tools/issue_model.py.  Each kernel runs ITER trips of 16 copies of ONE instruction on independent registers.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/op_cost.hip -o tools/ubench/op_cost && tools/ubench/op_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>
#define ITER 2048
#define R16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
// body: an asm string with %0 = destination / accumulator register of copy k, %1 = a second vector operand, %2 = a third
#define KERNEL(name, ASMSTR, CONS)                                                                                   \
  __global__ __launch_bounds__(256) void name(float *out, unsigned long long *cyc, float a, float b) {              \
    float r[8]; double d[8];                                                                                         \
    for (int k = 0; k < 8; ++k) { r[k] = threadIdx.x + k + a; d[k] = r[k]; }                                        \
    float x = b, y = a * 3.0f; double dx = b, dy = a;                                                                \
    (void)d; (void)dx; (void)dy; (void)x; (void)y;                                                                   \
    CONS(0)                                                                                                          \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                      \
    _Pragma("unroll 1") for (int i = 0; i < ITER; ++i) {                                                             \
      R16(ASMSTR)                                                                                                    \
    }                                                                                                                \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                      \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                      \
    float s = 0; for (int k = 0; k < 8; ++k) s += r[k] + (float)d[k];                                               \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + x + y + (float)dx;                                              \
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                                                 \
  }
#define NOPRE(k)
#define PRE_VCC_VALU(k) { asm volatile("v_cmp_gt_f32 vcc, %0, %1" :: "v"(r[0]), "v"(x) : "vcc"); }
#define PRE_VCC_SALU(k) { asm volatile("s_mov_b64 vcc, exec" ::: "vcc"); }
#define PRE_S20_VALU(k) { asm volatile("v_cmp_gt_f32_e64 s[20:21], %0, %1" :: "v"(r[0]), "v"(x) : "s20", "s21"); }
#define PRE_S20_SALU(k) { asm volatile("s_mov_b64 s[20:21], exec" ::: "s20", "s21"); }
#define F32_2(op) asm volatile(op " %0, %1, %0" : "+v"(r[k_]) : "v"(x));
#define A_ADD(k) { constexpr int k_ = k; asm volatile("v_add_f32 %0, %1, %0" : "+v"(r[k_]) : "v"(x)); }
#define A_MUL(k) { constexpr int k_ = k; asm volatile("v_mul_f32 %0, %1, %0" : "+v"(r[k_]) : "v"(x)); }
#define A_FMA(k) { constexpr int k_ = k; asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r[k_]) : "v"(x), "v"(y)); }
#define A_MAX(k) { constexpr int k_ = k; asm volatile("v_max_f32 %0, %1, %0" : "+v"(r[k_]) : "v"(x)); }
#define A_AND(k) { constexpr int k_ = k; asm volatile("v_and_b32 %0, %1, %0" : "+v"(r[k_]) : "v"(x)); }
#define A_LSHL(k) { constexpr int k_ = k; asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(r[k_])); }
#define A_BFI(k) { constexpr int k_ = k; asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(r[k_]) : "v"(x), "v"(y)); }
#define A_MOV(k) { constexpr int k_ = k; asm volatile("v_mov_b32 %0, %1" : "+v"(r[k_]) : "v"(x)); }
#define A_CND(k) { constexpr int k_ = k; asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[k_]) : "v"(x) : "vcc"); }
#define A_CND64(k) { constexpr int k_ = k; asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(r[k_]) : "v"(x) : "s20", "s21"); }
#define A_CMP(k) { constexpr int k_ = k; asm volatile("v_cmp_gt_f32 vcc, %0, %1" :: "v"(r[k_]), "v"(x) : "vcc"); }
#define A_CMP64(k) { constexpr int k_ = k; asm volatile("v_cmp_gt_f32_e64 s[20:21], %0, %1" :: "v"(r[k_]), "v"(x) : "s20", "s21"); }
#define A_CMPCND(k) { constexpr int k_ = k; asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[k_]) : "v"(x) : "vcc"); }
#define A_RCP(k) { constexpr int k_ = k; asm volatile("v_rcp_f32 %0, %0" : "+v"(r[k_])); }
#define A_SQRT(k) { constexpr int k_ = k; asm volatile("v_sqrt_f32 %0, %0" : "+v"(r[k_])); }
#define A_CVTFI(k) { constexpr int k_ = k; asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(r[k_])); }
#define A_CVTIF(k) { constexpr int k_ = k; asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(r[k_])); }
#define A_CVTUB(k) { constexpr int k_ = k; asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(r[k_])); }
#define A_CVTD(k) { constexpr int k_ = k; asm volatile("v_cvt_f64_f32 %0, %1" : "+v"(d[k_]) : "v"(r[k_])); }
#define A_CVTFD(k) { constexpr int k_ = k; asm volatile("v_cvt_f32_f64 %0, %1" : "+v"(r[k_]) : "v"(d[k_])); }
#define A_CVTID(k) { constexpr int k_ = k; asm volatile("v_cvt_i32_f64 %0, %1" : "+v"(r[k_]) : "v"(d[k_])); }
#define A_ADDD(k) { constexpr int k_ = k; asm volatile("v_add_f64 %0, %1, %0" : "+v"(d[k_]) : "v"(dx)); }
#define A_MULD(k) { constexpr int k_ = k; asm volatile("v_mul_f64 %0, %1, %0" : "+v"(d[k_]) : "v"(dx)); }
#define A_FMAD(k) { constexpr int k_ = k; asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[k_]) : "v"(dx), "v"(dy)); }
#define A_DSCALE(k) { constexpr int k_ = k; asm volatile("v_div_scale_f32 %0, vcc, %1, %1, %0" : "+v"(r[k_]) : "v"(x) : "vcc"); }
#define A_DFMAS(k) { constexpr int k_ = k; asm volatile("v_div_fmas_f32 %0, %0, %1, %2" : "+v"(r[k_]) : "v"(x), "v"(y) : "vcc"); }
#define A_DFIX(k) { constexpr int k_ = k; asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(r[k_]) : "v"(x), "v"(y)); }
#define A_DPPMOV(k) { constexpr int k_ = k; asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r[k_]) : "v"(x)); }
#define A_DPPADD(k) { constexpr int k_ = k; asm volatile("v_add_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r[k_]) : "v"(x)); }
#define A_DPPCND(k) { constexpr int k_ = k; asm volatile("v_cndmask_b32_dpp %0, %0, %1, vcc row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r[k_]) : "v"(x) : "vcc"); }
#define A_PKADD(k) { constexpr int k_ = k; asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(d[k_]) : "v"(dx)); }
#define A_PKMUL(k) { constexpr int k_ = k; asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(d[k_]) : "v"(dx)); }
#define A_PKFMA(k) { constexpr int k_ = k; asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(d[k_]) : "v"(dx), "v"(dy)); }
#define A_RFL(k) { constexpr int k_ = k; asm volatile("v_readfirstlane_b32 s20, %0" :: "v"(r[k_]) : "s20"); }
#define A_SAND(k) { asm volatile("s_and_b64 s[20:21], s[20:21], exec" ::: "s20", "s21", "scc"); }
#define A_SNOP(k) { asm volatile("s_nop 0"); }
#define A_SUBSAT(k) { constexpr int k_ = k; asm volatile("v_sub_u32 %0, %0, %1 clamp" : "+v"(r[k_]) : "v"(x)); }
#define A_SDWA(k) { constexpr int k_ = k; asm volatile("v_lshlrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(r[k_]) : "v"(x)); }
KERNEL(k_add, A_ADD, NOPRE) KERNEL(k_mul, A_MUL, NOPRE) KERNEL(k_fma, A_FMA, NOPRE) KERNEL(k_max, A_MAX, NOPRE) KERNEL(k_and, A_AND, NOPRE) KERNEL(k_lshl, A_LSHL, NOPRE)
KERNEL(k_bfi, A_BFI, NOPRE) KERNEL(k_mov, A_MOV, NOPRE) KERNEL(k_cnd, A_CND, NOPRE) KERNEL(k_cnd64, A_CND64, NOPRE) KERNEL(k_cmp, A_CMP, NOPRE) KERNEL(k_cmp64, A_CMP64, NOPRE)
KERNEL(k_cmpcnd, A_CMPCND, NOPRE) KERNEL(k_rcp, A_RCP, NOPRE) KERNEL(k_sqrt, A_SQRT, NOPRE) KERNEL(k_cvtfi, A_CVTFI, NOPRE) KERNEL(k_cvtif, A_CVTIF, NOPRE) KERNEL(k_cvtub, A_CVTUB, NOPRE)
KERNEL(k_cvtd, A_CVTD, NOPRE) KERNEL(k_cvtfd, A_CVTFD, NOPRE) KERNEL(k_cvtid, A_CVTID, NOPRE) KERNEL(k_addd, A_ADDD, NOPRE) KERNEL(k_muld, A_MULD, NOPRE) KERNEL(k_fmad, A_FMAD, NOPRE)
KERNEL(k_dscale, A_DSCALE, NOPRE) KERNEL(k_dfmas, A_DFMAS, NOPRE) KERNEL(k_dfix, A_DFIX, NOPRE) KERNEL(k_dppmov, A_DPPMOV, NOPRE) KERNEL(k_dppadd, A_DPPADD, NOPRE) KERNEL(k_dppcnd, A_DPPCND, NOPRE)
KERNEL(k_pkadd, A_PKADD, NOPRE) KERNEL(k_pkmul, A_PKMUL, NOPRE) KERNEL(k_pkfma, A_PKFMA, NOPRE) KERNEL(k_rfl, A_RFL, NOPRE) KERNEL(k_sand, A_SAND, NOPRE) KERNEL(k_snop, A_SNOP, NOPRE)
KERNEL(k_subsat, A_SUBSAT, NOPRE) KERNEL(k_sdwa, A_SDWA, NOPRE)
KERNEL(k_cnd_vv, A_CND, PRE_VCC_VALU) KERNEL(k_cnd_vs, A_CND, PRE_VCC_SALU) KERNEL(k_cnd64_v, A_CND64, PRE_S20_VALU) KERNEL(k_cnd64_s, A_CND64, PRE_S20_SALU)
KERNEL(k_dppcnd_v, A_DPPCND, PRE_VCC_VALU)
#define A_CNDTRIP(k) { constexpr int k_ = k; if (k_ == 0) asm volatile("s_and_b64 vcc, vcc, exec" ::: "vcc", "scc"); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[k_]) : "v"(x) : "vcc"); }
#define A_CNDTRIPV(k) { constexpr int k_ = k; if (k_ == 0) asm volatile("v_cmp_gt_f32 vcc, %0, %1" :: "v"(r[7]), "v"(x) : "vcc"); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[k_]) : "v"(x) : "vcc"); }
KERNEL(k_cnd_trip_s, A_CNDTRIP, NOPRE) KERNEL(k_cnd_trip_v, A_CNDTRIPV, NOPRE)
#define A_CMP2CND(k) { constexpr int k_ = k; asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc\n\tv_cndmask_b32 %2, %2, %1, vcc" : "+v"(r[k_]) : "v"(x), "v"(y) : "vcc"); }
#define A_CMPADDCND(k) { constexpr int k_ = k; asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_add_f32 %2, %1, %2\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[k_]), "+v"(y) : "v"(x) : "vcc"); }
#define A_CND64VCC(k) { constexpr int k_ = k; asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(r[k_]) : "v"(x) : "vcc"); }
#define A_CMP64_2CND(k) { constexpr int k_ = k; asm volatile("v_cmp_gt_f32_e64 s[20:21], %0, %1\n\tv_cndmask_b32_e64 %0, %0, %1, s[20:21]\n\tv_cndmask_b32_e64 %2, %2, %1, s[20:21]" : "+v"(r[k_]) : "v"(x), "v"(y) : "s20", "s21"); }
#define A_CMP64_CND(k) { constexpr int k_ = k; asm volatile("v_cmp_gt_f32_e64 s[20:21], %0, %1\n\tv_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(r[k_]) : "v"(x) : "s20", "s21"); }
#define A_DPPCND_MIX(k) { constexpr int k_ = k; asm volatile("v_cndmask_b32_dpp %0, %0, %4, vcc row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_fma_f32 %1, %4, %4, %1\n\tv_fma_f32 %2, %4, %4, %2\n\tv_fma_f32 %3, %4, %4, %3" : "+v"(r[k_]), "+v"(r[(k_ + 1) & 7]), "+v"(r[(k_ + 2) & 7]), "+v"(r[(k_ + 3) & 7]) : "v"(x) : "vcc"); }
#define A_FMA3(k) { constexpr int k_ = k; asm volatile("v_fma_f32 %0, %3, %3, %0\n\tv_fma_f32 %1, %3, %3, %1\n\tv_fma_f32 %2, %3, %3, %2" : "+v"(r[(k_ + 1) & 7]), "+v"(r[(k_ + 2) & 7]), "+v"(r[(k_ + 3) & 7]) : "v"(x)); }
#define A_MOVDPP_CND64_MIX(k) { constexpr int k_ = k; asm volatile("v_mov_b32_dpp %5, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_cndmask_b32_e64 %0, %5, %4, s[20:21]\n\tv_fma_f32 %1, %4, %4, %1\n\tv_fma_f32 %2, %4, %4, %2\n\tv_fma_f32 %3, %4, %4, %3" : "+v"(r[k_]), "+v"(r[(k_ + 1) & 7]), "+v"(r[(k_ + 2) & 7]), "+v"(r[(k_ + 3) & 7]) : "v"(x), "v"(y) : "s20", "s21"); }
#define A_ANDDPP_OR_MIX(k) { constexpr int k_ = k; asm volatile("v_and_b32_dpp %5, %0, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_or_b32 %0, %5, %4\n\tv_fma_f32 %1, %4, %4, %1\n\tv_fma_f32 %2, %4, %4, %2\n\tv_fma_f32 %3, %4, %4, %3" : "+v"(r[k_]), "+v"(r[(k_ + 1) & 7]), "+v"(r[(k_ + 2) & 7]), "+v"(r[(k_ + 3) & 7]) : "v"(x), "v"(y)); }
KERNEL(k_dppcnd_mix, A_DPPCND_MIX, PRE_VCC_VALU) KERNEL(k_fma3, A_FMA3, NOPRE) KERNEL(k_movdpp_cnd64_mix, A_MOVDPP_CND64_MIX, PRE_S20_VALU) KERNEL(k_anddpp_or_mix, A_ANDDPP_OR_MIX, NOPRE)
KERNEL(k_cmp2cnd, A_CMP2CND, NOPRE) KERNEL(k_cmpaddcnd, A_CMPADDCND, NOPRE) KERNEL(k_cnd64vcc, A_CND64VCC, PRE_VCC_VALU) KERNEL(k_cmp64_2cnd, A_CMP64_2CND, NOPRE) KERNEL(k_cmp64_cnd, A_CMP64_CND, NOPRE)
typedef void (*kern_t)(float *, unsigned long long *, float, float);
struct Op { const char *name; kern_t k; int per_copy; };
int main() {
  float *d; unsigned long long *c; hipMalloc(&d, 256 * 8 * 256 * sizeof(float)); hipMalloc(&c, 256 * 8 * sizeof(unsigned long long));
  std::vector<Op> ops = {{"v_add_f32", k_add, 1}, {"v_mul_f32", k_mul, 1}, {"v_fma_f32", k_fma, 1}, {"v_max_f32", k_max, 1}, {"v_and_b32", k_and, 1}, {"v_lshlrev_b32", k_lshl, 1},
    {"v_bfi_b32", k_bfi, 1}, {"v_mov_b32", k_mov, 1}, {"v_cndmask_b32 (vcc)", k_cnd, 1}, {"v_cndmask_b32_e64 (sgpr pair)", k_cnd64, 1}, {"v_cmp_gt_f32 -> vcc", k_cmp, 1},
    {"v_cmp_gt_f32_e64 -> sgpr pair", k_cmp64, 1}, {"v_cmp + v_cndmask (dependent pair)", k_cmpcnd, 2}, {"v_rcp_f32", k_rcp, 1}, {"v_sqrt_f32", k_sqrt, 1},
    {"v_cvt_f32_i32", k_cvtfi, 1}, {"v_cvt_i32_f32", k_cvtif, 1}, {"v_cvt_f32_ubyte0", k_cvtub, 1}, {"v_cvt_f64_f32", k_cvtd, 1}, {"v_cvt_f32_f64", k_cvtfd, 1},
    {"v_cvt_i32_f64", k_cvtid, 1}, {"v_add_f64", k_addd, 1}, {"v_mul_f64", k_muld, 1}, {"v_fma_f64", k_fmad, 1}, {"v_div_scale_f32", k_dscale, 1}, {"v_div_fmas_f32", k_dfmas, 1},
    {"v_div_fixup_f32", k_dfix, 1}, {"v_mov_b32_dpp row_shr:1", k_dppmov, 1}, {"v_add_f32_dpp row_shr:1", k_dppadd, 1}, {"v_cndmask_b32_dpp row_shr:1", k_dppcnd, 1},
    {"v_pk_add_f32", k_pkadd, 1}, {"v_pk_mul_f32", k_pkmul, 1}, {"v_pk_fma_f32", k_pkfma, 1}, {"v_readfirstlane_b32", k_rfl, 1}, {"s_and_b64", k_sand, 1}, {"s_nop 0", k_snop, 1},
    {"v_sub_u32 clamp", k_subsat, 1}, {"v_lshlrev_b32_sdwa", k_sdwa, 1},
    {"v_cndmask vcc, vcc written ONCE by VALU", k_cnd_vv, 1}, {"v_cndmask vcc, vcc written ONCE by SALU", k_cnd_vs, 1},
    {"v_cndmask_e64 s[], written ONCE by VALU", k_cnd64_v, 1}, {"v_cndmask_e64 s[], written ONCE by SALU", k_cnd64_s, 1},
    {"v_cndmask_dpp vcc, vcc once by VALU", k_dppcnd_v, 1}, {"8 x v_cndmask vcc per SALU write of vcc", k_cnd_trip_s, 1}, {"8 x v_cndmask vcc per VALU write of vcc", k_cnd_trip_v, 1},
    {"GROUP: v_cndmask_dpp vcc + 3 v_fma (cycles per group)", k_dppcnd_mix, 1}, {"GROUP: 3 v_fma (cycles per group)", k_fma3, 1}, {"GROUP: v_mov_dpp + v_cndmask_e64 s[] + 3 v_fma", k_movdpp_cnd64_mix, 1}, {"GROUP: v_and_dpp + v_or + 3 v_fma", k_anddpp_or_mix, 1},
    {"v_cmp vcc; v_cndmask vcc; v_cndmask vcc", k_cmp2cnd, 3}, {"v_cmp vcc; v_add; v_cndmask vcc", k_cmpaddcnd, 3}, {"v_cndmask_b32_e64 ..., vcc (VOP3 form)", k_cnd64vcc, 1},
    {"v_cmp_e64 s[]; 2 x v_cndmask_e64 s[]", k_cmp64_2cnd, 3}, {"v_cmp_e64 s[]; v_cndmask_e64 s[]", k_cmp64_cnd, 2}};
  {   // calibration: s_memtime ticks against wall time for the 8-waves-per-SIMD v_add_f32 run
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_add, dim3(256 * 8), dim3(256), 0, 0, d, c, 1.0000001f, 1e-9f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_add, dim3(256 * 8), dim3(256), 0, 0, d, c, 1.0000001f, 1e-9f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(256 * 8);
    hipMemcpy(h.data(), c, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double sum = 0; for (auto v : h) sum += (double)v;
    const double ticks = sum / h.size();
    printf("calibration: the 8-waves-per-SIMD v_add_f32 kernel: %.1f us wall (whole launch), %.0f s_memtime ticks per wave loop -> >= %.3f ticks per ns; 262,144 instructions per SIMD in that time = %.2f ns each\n",
           ms * 1e3, ticks, ticks / (ms * 1e6), ms * 1e6 / 262144.0);
  }
  printf("%-36s %14s %14s\n", "instruction", "1 wave / SIMD", "8 waves / SIMD");
  for (auto &o : ops) {
    double res[2];
    for (int mode = 0; mode < 2; ++mode) {
      // mode 0: one wave per SIMD (256 workgroups of 256 threads = 4 waves per CU); mode 1: 8 waves per SIMD
      const int blocks = mode == 0 ? 256 : 256 * 8;
      hipLaunchKernelGGL(o.k, dim3(blocks), dim3(256), 0, 0, d, c, 1.0000001f, 1e-9f);
      hipLaunchKernelGGL(o.k, dim3(blocks), dim3(256), 0, 0, d, c, 1.0000001f, 1e-9f);
      hipDeviceSynchronize();
      std::vector<unsigned long long> h(blocks);
      hipMemcpy(h.data(), c, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
      double sum = 0; for (auto v : h) sum += (double)v;
      const double per_wave_cycles = sum / blocks;   // s_memtime ticks of one wave's loop
      const double waves_per_simd = mode == 0 ? 1.0 : 8.0;
      res[mode] = per_wave_cycles / ((double)ITER * 16 * o.per_copy) / waves_per_simd;   // SIMD cycles per instruction
    }
    printf("%-36s %14.2f %14.2f\n", o.name, res[0], res[1]);
  }
  return 0;
}
