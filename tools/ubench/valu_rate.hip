// Micro-benchmark: issue rate of scalar vs packed FP32 VALU ops (and f64) on one MI355X, full occupancy.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define ITER 4096
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, float a, float b) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  v2f p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, p4 = {x1, x0}, p5 = {x3, x2}, p6 = {x5, x4}, p7 = {x7, x6};
  double d0 = x0, d1 = x1, d2 = x2, d3 = x3, d4 = x4, d5 = x5, d6 = x6, d7 = x7;
  v2f a2 = {a, a}, b2 = {b, b};
#pragma unroll 1
  for (int i = 0; i < ITER; ++i) {
    if (MODE == 0) {   // 16 scalar f32 ops
      x0 = x0 * a; x1 = x1 * a; x2 = x2 * a; x3 = x3 * a; x4 = x4 * a; x5 = x5 * a; x6 = x6 * a; x7 = x7 * a;
      x0 = x0 + b; x1 = x1 + b; x2 = x2 + b; x3 = x3 + b; x4 = x4 + b; x5 = x5 + b; x6 = x6 + b; x7 = x7 + b;
    } else if (MODE == 1) {   // 16 packed f32 ops (32 flops/lane)
      p0 = p0 * a2; p1 = p1 * a2; p2 = p2 * a2; p3 = p3 * a2; p4 = p4 * a2; p5 = p5 * a2; p6 = p6 * a2; p7 = p7 * a2;
      p0 = p0 + b2; p1 = p1 + b2; p2 = p2 + b2; p3 = p3 + b2; p4 = p4 + b2; p5 = p5 + b2; p6 = p6 + b2; p7 = p7 + b2;
    } else {   // 16 f64 ops
      d0 = d0 * a; d1 = d1 * a; d2 = d2 * a; d3 = d3 * a; d4 = d4 * a; d5 = d5 * a; d6 = d6 * a; d7 = d7 * a;
      d0 = d0 + b; d1 = d1 + b; d2 = d2 + b; d3 = d3 + b; d4 = d4 + b; d5 = d5 + b; d6 = d6 + b; d7 = d7 + b;
    }
  }
  float r = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p5.y + p6.x + p7.y +
            (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int MODE>
double run(float *d) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256 * 8;   // 8 workgroups of 4 waves per CU
  k<MODE><<<blocks, 256>>>(d, 1.0000001f, 1e-9f);
  hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(d, 1.0000001f, 1e-9f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double insts = (double)blocks * 4 /*waves*/ * ITER * 16;
  return insts / (ms * 1e-3) / (256.0 * 4) / 1e9;   // wave-instructions per ns per SIMD
}
int main() {
  float *d; hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
  const char *names[] = {"v_mul/v_add_f32", "v_pk_mul/v_pk_add_f32", "v_mul/v_add_f64"};
  double r[3] = {run<0>(d), run<1>(d), run<2>(d)};
  for (int i = 0; i < 3; i++) printf("%-24s %.3f wave-instr/ns/SIMD  -> %.2f cycles per instruction at 2.4 GHz\n", names[i], r[i], 2.4 / r[i]);
  return 0;
}
