#include <hip/hip_runtime.h>
__global__ void k(int *out) {
  int v = threadIdx.x;
  int r = __builtin_amdgcn_update_dpp(0, v, 0x13C, 0xF, 0xF, false);   // wave_ror:1
  int s = __builtin_amdgcn_update_dpp(0, v, 0x138, 0xF, 0xF, false);   // wave_shr:1
  out[threadIdx.x] = r * 1000 + s;
}
int main() { int *d; hipMalloc(&d, 256); k<<<1,64>>>(d); int h[64]; hipMemcpy(h, d, 256, hipMemcpyDeviceToHost); for (int i = 0; i < 64; i += 9) printf("%d:%d ", i, h[i]); printf("\n"); return 0; }
