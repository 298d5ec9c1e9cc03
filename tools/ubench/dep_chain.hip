// Micro-benchmark: cycles per instruction of ONE wave on a CU for the dependent-chain shapes of asdr_update_kernel's
// sequential phases (s_memtime around an unrolled loop; gfx950).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench/dep_chain.hip -o tools/ubench/dep_chain && tools/ubench/dep_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 4096
template <int MODE>
__global__ __launch_bounds__(64) void k(float *out, unsigned long long *cyc, float a, float b, float c) {
  float x = threadIdx.x * 1e-3f, y = x + 1.0f, z = y + 1.0f, w = z + 1.0f;
  unsigned hc = threadIdx.x;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 16
  for (int i = 0; i < N; ++i) {
    if (MODE == 0) { x = x * a; x = x + b; }                                    // mul, add (blanker average)
    if (MODE == 1) { x = x * a; x = x + b; y = y * a; y = y + b; }              // two independent chains
    if (MODE == 2) { x = x * a; x = x + b; y = y * a; y = y + b; z = z * a; z = z + b; w = w * a; w = w + b; }   // four
    if (MODE == 3) { const float t = x + a, tw = t + b; x = (t > c) ? tw : t; }   // add, add, cmp, cndmask (mixer phase)
    if (MODE == 4) { const float t = x + a, tw = t + b; x = (t > c) ? tw : t; y = y * a; y = y + b; }   // phase + average
    if (MODE == 5) {   // AGC step shape
      const bool att = y > x; const bool idle = (hc == 0u);
      const float al = att ? a : b, be = att ? b : a;
      const float v = al * x + be * y;
      x = (att || idle) ? v : x;
      hc = att ? 100u : (idle ? 0u : hc - 1u);
      y = y + c;
    }
    if (MODE == 6) { x = x * a + y * b; }                                        // mul, mul, add (biquad-like: 2 deep)
    if (MODE == 7) { x = __builtin_amdgcn_update_dpp(0.0f, x, 0x111, 0xF, 0xF, true) + a; }   // dpp + add
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = x + y + z + w + (float)hc;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  float *d; unsigned long long *c; hipMalloc(&d, 256); hipMalloc(&c, 8);
  const char *names[] = {"1 chain mul+add", "2 chains mul+add", "4 chains mul+add", "add,add,cmp,cndmask", "phase step + average step",
                         "AGC step", "mul,mul,add", "dpp-mov + add"};
  const int per_iter[] = {2, 4, 8, 4, 6, 11, 3, 2};
  for (int m = 0; m < 8; ++m) {
    for (int rep = 0; rep < 2; ++rep) {
      switch (m) {
        case 0: k<0><<<1, 64>>>(d, c, 0.999f, 1e-3f, 6.0f); break; case 1: k<1><<<1, 64>>>(d, c, 0.999f, 1e-3f, 6.0f); break;
        case 2: k<2><<<1, 64>>>(d, c, 0.999f, 1e-3f, 6.0f); break; case 3: k<3><<<1, 64>>>(d, c, 0.7f, -6.28f, 6.28f); break;
        case 4: k<4><<<1, 64>>>(d, c, 0.7f, -6.28f, 6.28f); break; case 5: k<5><<<1, 64>>>(d, c, 0.99f, 0.01f, 1e-4f); break;
        case 6: k<6><<<1, 64>>>(d, c, 0.5f, 0.4f, 0.f); break; default: k<7><<<1, 64>>>(d, c, 0.5f, 0.4f, 0.f); break;
      }
      hipDeviceSynchronize();
    }
    unsigned long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    printf("%-28s %7.1f cycles per iteration (~%d VALU) = %.1f cycles per instruction\n", names[m], (double)h / N, per_iter[m], (double)h / N / per_iter[m]);
  }
  return 0;
}
