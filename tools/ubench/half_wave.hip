// Micro-benchmark: does a wave64 vector instruction whose upper (or lower) 32 lanes are switched off in EXEC cost one pass of the SIMD-32
// instead of two?  (If it did, a one-lane-per-channel recurrence such as the SAM PLL could run 32 channels per wave on twice the waves
// at the same vector-unit cost.)  f32 and f64, 8 waves per SIMD, independent instructions.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench/half_wave.hip -o tools/ubench/half_wave && tools/ubench/half_wave
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 4096
template <int F64, int HALF>
__global__ __launch_bounds__(256) void k(float *out, float a, float b) {
  const int lane = threadIdx.x & 63;
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  double d0 = x0, d1 = x1, d2 = x2, d3 = x3, d4 = x4, d5 = x5, d6 = x6, d7 = x7;
  const bool on = HALF == 0 ? true : (HALF == 1 ? lane < 32 : (HALF == 2 ? lane >= 32 : (lane & 1) == 0));
  if (on) {
#pragma unroll 1
    for (int i = 0; i < ITER; ++i) {
      if (!F64) {
        x0 = x0 * a; x1 = x1 * a; x2 = x2 * a; x3 = x3 * a; x4 = x4 * a; x5 = x5 * a; x6 = x6 * a; x7 = x7 * a;
        x0 = x0 + b; x1 = x1 + b; x2 = x2 + b; x3 = x3 + b; x4 = x4 + b; x5 = x5 + b; x6 = x6 + b; x7 = x7 + b;
      } else {
        d0 = d0 * a; d1 = d1 * a; d2 = d2 * a; d3 = d3 * a; d4 = d4 * a; d5 = d5 * a; d6 = d6 * a; d7 = d7 * a;
        d0 = d0 + b; d1 = d1 + b; d2 = d2 + b; d3 = d3 + b; d4 = d4 + b; d5 = d5 + b; d6 = d6 + b; d7 = d7 + b;
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
}
template <int F64, int HALF>
double run(float *d) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256 * 8;
  k<F64, HALF><<<blocks, 256>>>(d, 1.0000001f, 1e-9f);
  hipEventRecord(e0);
  k<F64, HALF><<<blocks, 256>>>(d, 1.0000001f, 1e-9f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double insts = (double)blocks * 4 * ITER * 16;
  return insts / (ms * 1e-3) / (256.0 * 4) / 1e9;
}
int main() {
  float *d; hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
  const char *h[] = {"all 64 lanes", "lanes 0..31", "lanes 32..63", "even lanes"};
  double r[2][4] = {{run<0, 0>(d), run<0, 1>(d), run<0, 2>(d), run<0, 3>(d)}, {run<1, 0>(d), run<1, 1>(d), run<1, 2>(d), run<1, 3>(d)}};
  for (int f = 0; f < 2; f++) for (int i = 0; i < 4; i++)
    printf("%s %-14s %.3f wave-instr/ns/SIMD -> %.2f cycles per instruction at 2.4 GHz\n", f ? "f64" : "f32", h[i], r[f][i], 2.4 / r[f][i]);
  return 0;
}
