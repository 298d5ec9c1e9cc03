// Exhaustive check (all 2^32 float bit patterns) of a 5-instruction replacement for the IEEE division inside the
// reference's fast_sqrt_f32(x, 1) (AudioSDR.h:434-446): out = magic(x); result = 0.5f*(out + x/out).
// Prints how many inputs give a different RESULT bit pattern, and the range of |x| where that happens.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench/sqrt_div_check.hip -o tools/ubench/sqrt_div_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
__device__ __forceinline__ float magic(float x) { uint32_t i = __float_as_uint(x); i -= 1u << 23; i >>= 1; i += 1u << 29; return __uint_as_float(i); }
__device__ __forceinline__ float ref_sqrt(float x) { const float out = magic(x); return 0.5f * (out + x / out); }
__device__ __forceinline__ float fast_div(float n, float d) {
  const float r0 = __builtin_amdgcn_rcpf(d);
  const float r = __builtin_fmaf(__builtin_fmaf(-d, r0, 1.0f), r0, r0);   // one Newton step on the reciprocal
  const float q0 = n * r;
  const float e = __builtin_fmaf(-d, q0, n);
  const float q1 = __builtin_fmaf(e, r, q0);
#ifdef THIRD
  const float e2 = __builtin_fmaf(-d, q1, n);
  return __builtin_fmaf(e2, r, q1);
#else
  return q1;
#endif
}
__device__ __forceinline__ float new_sqrt(float x) { const float out = magic(x); return 0.5f * (out + fast_div(x, out)); }
__global__ void k(unsigned long long *bad, uint32_t *lo, uint32_t *hi, uint32_t *example) {
  const uint32_t base = (blockIdx.x * blockDim.x + threadIdx.x);
  unsigned long long nb = 0; uint32_t mn = 0xFFFFFFFFu, mx = 0u;
  for (uint32_t rep = 0; rep < 1024; ++rep) {
    const uint32_t bits = base + rep * (1u << 22);
    if (bits >= 0x7F800000u) continue;   // x >= 0 and finite only (x = I*I + Q*Q)
    const float x = __uint_as_float(bits);
    const float a = ref_sqrt(x), b = new_sqrt(x);
    const bool same = (__float_as_uint(a) == __float_as_uint(b)) || (a != a && b != b);
    if (!same) { nb++; const uint32_t m = bits & 0x7FFFFFFFu; mn = m < mn ? m : mn; mx = m > mx ? m : mx; if (nb == 1) atomicExch(example, bits); }
  }
  if (nb) { atomicAdd(bad, nb); atomicMin(lo, mn); atomicMax(hi, mx); }
}
int main() {
  unsigned long long *bad; uint32_t *lo, *hi, *ex;
  hipMalloc(&bad, 8); hipMalloc(&lo, 4); hipMalloc(&hi, 4); hipMalloc(&ex, 4);
  unsigned long long z = 0; uint32_t l = 0xFFFFFFFFu, h = 0, e = 0;
  hipMemcpy(bad, &z, 8, hipMemcpyHostToDevice); hipMemcpy(lo, &l, 4, hipMemcpyHostToDevice); hipMemcpy(hi, &h, 4, hipMemcpyHostToDevice); hipMemcpy(ex, &e, 4, hipMemcpyHostToDevice);
  k<<<(1u << 22) / 256, 256>>>(bad, lo, hi, ex);
  hipDeviceSynchronize();
  hipMemcpy(&z, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&l, lo, 4, hipMemcpyDeviceToHost); hipMemcpy(&h, hi, 4, hipMemcpyDeviceToHost); hipMemcpy(&e, ex, 4, hipMemcpyDeviceToHost);
  float fl, fh, fe; memcpy(&fl, &l, 4); memcpy(&fh, &h, 4); memcpy(&fe, &e, 4);
  printf("mismatching inputs: %llu of 4294967296; |x| range of mismatches: [%g (0x%08x), %g (0x%08x)]; first example 0x%08x = %g\n", z, fl, l, fh, h, e, fe);
  return 0;
}
