// Micro-benchmark: the HBM access pattern of asdr_update_kernel (C2: 65,536 channels, one wave = 8 channels, lane = 16-byte
// pieces of per-channel rows) with NO arithmetic: what the memory system alone needs for one launch.
//   mode 0: per-channel rows as the product lays them out (rows of 96 B .. 1.5 KB in separate arrays), all loads of a wave
//           issued up front, then all stores
//   mode 1: the same bytes as ONE contiguous region per wave (fully coalesced 1 KiB wave-instructions)
//   mode 2: mode 0, but the loads in 5 dependent groups (each group waits for the previous one), as the phases of the
//           kernel consume them
// hipcc --offload-arch=gfx950 -O3 tools/ubench/mem_pattern.hip -o tools/ubench/mem_pattern && tools/ubench/mem_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

struct Args {
  const int4 *in_i, *in_q; int4 *out;
  int4 *nb_hist; uint32_t *nb_mask; int4 *hil_q, *hil_i, *small; const int4 *params;
  int4 *flat;
  int n_ch; uint32_t ns;
};

__device__ __forceinline__ int4 x4(int4 a, int4 b) { return make_int4(a.x ^ b.x, a.y ^ b.y, a.z ^ b.z, a.w ^ b.w); }

template <int MODE>
__global__ __launch_bounds__(64, 3) void k(Args a) {
  __shared__ int4 pad[776];   // 12,416 B: same LDS footprint as the product kernel -> 12 waves/CU
  const int lane = threadIdx.x, c8 = lane >> 3, s8 = lane & 7;
  const int ch = blockIdx.x * 8 + c8;
  if (lane == 0) pad[0] = make_int4(0, 0, 0, 0);
  int4 acc = make_int4(lane, 0, 0, 0);
  if (MODE == 1) {
    // 31 load + 18 store wave-instructions of 1 KiB over one contiguous region of 56 KiB per wave (49 KiB used)
    int4 *base = a.flat + (size_t)blockIdx.x * (56 * 64) + lane;
    int4 v[31];
#pragma unroll
    for (int i = 0; i < 31; ++i) v[i] = base[i * 64];
#pragma unroll
    for (int i = 0; i < 31; ++i) acc = x4(acc, v[i]);
#pragma unroll
    for (int i = 0; i < 18; ++i) base[(31 + i) * 64] = acc;
    return;
  }
  if (MODE == 3 || MODE == 4) {
    // mode 4: per-channel rows, but every wave-instruction reads a whole 128-B piece of each of the 8 rows (lane s8 = 16 B of it)
    // mode 3: state arrays tiled by 8 channels ([piece k][channel][lane] -> 1 KiB contiguous per wave-instruction); I/O rows as mode 4
    const uint32_t ns = a.ns, nm = (ns + 1) % 3, nn = (ns + 2) % 3;
    const bool T = (MODE == 3);
    const size_t tile = blockIdx.x;
    const int4 *pi = a.in_i + (size_t)ch * 16 + s8, *pq = a.in_q + (size_t)ch * 16 + s8;
    // rows: base + ch*rowpieces + k*8 + s8 ; tiles: base + tile*rowpieces*8 + k*64 + lane
    auto at = [&](int4 *base, int rowpieces, int k) { return T ? base + tile * rowpieces * 8 + k * 64 + lane : base + (size_t)ch * rowpieces + k * 8 + s8; };
    int4 v[31]; uint32_t m[5];
    v[0] = pi[0]; v[1] = pi[8]; v[2] = pq[0]; v[3] = pq[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[4 + i] = *at(a.nb_hist, 96, ns * 4 + i); v[8 + i] = *at(a.nb_hist, 96, nm * 4 + i); }
#pragma unroll
    for (int i = 0; i < 5; ++i) m[i] = T ? a.nb_mask[tile * 320 + i * 64 + lane] : a.nb_mask[(size_t)ch * 40 + i * 8 + s8];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[12 + i] = *at(a.small, 28, i < 3 ? i : 2);
    v[16] = a.params[(size_t)ch * 6 + (s8 < 6 ? s8 : 5)];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[17 + i] = *at(a.hil_q, 64, i);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[25 + i] = *at(a.hil_i, 64, i);
    v[29] = v[28]; v[30] = v[27];
#pragma unroll
    for (int i = 0; i < 31; ++i) acc = x4(acc, v[i]);
#pragma unroll
    for (int i = 0; i < 5; ++i) acc.x ^= m[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) { *at(a.nb_hist, 96, nn * 4 + i) = acc; *at(a.hil_q, 64, i) = acc; *at(a.hil_i, 64, 4 + i) = acc; }
#pragma unroll
    for (int i = 0; i < 3; ++i) *at(a.small, 28, i) = acc;
    a.out[(size_t)ch * 16 + s8] = acc; a.out[(size_t)ch * 16 + 8 + s8] = acc;
    return;
  }
  const uint32_t ns = a.ns, nm = (ns + 1) % 3, nn = (ns + 2) % 3;
  const int4 *pi = a.in_i + (size_t)ch * 16 + 2 * s8, *pq = a.in_q + (size_t)ch * 16 + 2 * s8;
  int4 *hist = a.nb_hist + (size_t)ch * 96 + 2 * s8;   // 1536 B row = 96 int4: slot*32 + {I:0, Q:16}
  uint32_t *mrow = a.nb_mask + (size_t)ch * 40 + 5 * s8;
  int4 *hq = a.hil_q + (size_t)ch * 64 + 4 * s8, *hi = a.hil_i + (size_t)ch * 64 + 4 * s8;
  int4 *sm = a.small + (size_t)ch * 28 + 4 * s8;         // 448 B row = 28 int4 (lane 7 wraps: stays inside the row of ch+1.. fine)
  const int4 *pp = a.params + (size_t)ch * 6 + (s8 < 6 ? s8 : 5);
  int4 v[31]; uint32_t m[5];
  // group 1: input + blanker rows + mask + small + params
  v[0] = pi[0]; v[1] = pi[1]; v[2] = pq[0]; v[3] = pq[1];
  v[4] = hist[ns * 32]; v[5] = hist[ns * 32 + 1]; v[6] = hist[ns * 32 + 16]; v[7] = hist[ns * 32 + 17];
  v[8] = hist[nm * 32]; v[9] = hist[nm * 32 + 1]; v[10] = hist[nm * 32 + 16]; v[11] = hist[nm * 32 + 17];
#pragma unroll
  for (int i = 0; i < 5; ++i) m[i] = mrow[i];
  if (MODE == 2) { acc = x4(acc, x4(v[0], x4(v[4], v[8]))); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
  v[12] = sm[0]; v[13] = sm[1]; v[14] = (s8 < 7) ? sm[2] : v[13]; v[15] = (s8 < 7) ? sm[3] : v[13];
  v[16] = pp[0];
  if (MODE == 2) { acc = x4(acc, v[12]); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
  // Hilbert rings
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[17 + i] = hq[i]; v[21 + i] = hq[32 + i]; }
  if (MODE == 2) { acc = x4(acc, v[17]); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#pragma unroll
  for (int i = 0; i < 4; ++i) v[25 + i] = hi[i];
  v[29] = v[28]; v[30] = v[27];
  if (MODE == 2) { acc = x4(acc, v[25]); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#pragma unroll
  for (int i = 0; i < 31; ++i) acc = x4(acc, v[i]);
#pragma unroll
  for (int i = 0; i < 5; ++i) acc.x ^= m[i];
  // stores: newest blanker block, Hilbert rings, small state, output
  hist[nn * 32] = acc; hist[nn * 32 + 1] = acc; hist[nn * 32 + 16] = acc; hist[nn * 32 + 17] = acc;
#pragma unroll
  for (int i = 0; i < 4; ++i) { hq[i] = acc; hi[32 + i] = acc; }
  sm[0] = acc; sm[1] = acc; if (s8 < 7) { sm[2] = acc; sm[3] = acc; }
  a.out[(size_t)ch * 16 + 2 * s8] = acc; a.out[(size_t)ch * 16 + 2 * s8 + 1] = acc;
}

int main(int argc, char **argv) {
  const int n = 65536, rows = n + 8;
  Args a{};
  auto alloc = [&](size_t bytes) { void *p; if (hipMalloc(&p, bytes) != hipSuccess) { printf("alloc failed\n"); exit(1); } hipMemset(p, 1, bytes); return p; };
  a.in_i = (const int4 *)alloc((size_t)rows * 256); a.in_q = (const int4 *)alloc((size_t)rows * 256); a.out = (int4 *)alloc((size_t)rows * 256);
  a.nb_hist = (int4 *)alloc((size_t)rows * 1536); a.nb_mask = (uint32_t *)alloc((size_t)rows * 160);
  a.hil_q = (int4 *)alloc((size_t)rows * 1024); a.hil_i = (int4 *)alloc((size_t)rows * 1024);
  a.small = (int4 *)alloc((size_t)rows * 448 + 64); a.params = (const int4 *)alloc((size_t)rows * 96);
  a.flat = (int4 *)alloc((size_t)(n / 8) * 56 * 1024);
  a.n_ch = n;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char *names[] = {"rows, loads up front", "one contiguous region per wave", "rows, 5 dependent load groups", "8-channel tiles + 128-B I/O pieces", "rows, 128-B pieces per row"};
  for (int mode = 0; mode < 5; ++mode) {
    float best = 1e9f, sum = 0; int cnt = 0;
    for (int it = 0; it < 30; ++it) {
      a.ns = it % 3;
      hipEventRecord(e0);
      if (mode == 0) k<0><<<n / 8, 64>>>(a); else if (mode == 1) k<1><<<n / 8, 64>>>(a); else if (mode == 2) k<2><<<n / 8, 64>>>(a); else if (mode == 3) k<3><<<n / 8, 64>>>(a); else k<4><<<n / 8, 64>>>(a);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (it >= 5) { best = ms < best ? ms : best; sum += ms; cnt++; }
    }
    const double bytes = (double)n * (31 * 16 * 8 + 5 * 4 * 8 + 18 * 16 * 8);
    printf("mode %d (%-34s): mean %.4f ms  min %.4f ms  -> %.2f TB/s (%.0f MB per launch)\n", mode, names[mode], sum / cnt, best, bytes / (sum / cnt * 1e-3) / 1e12, bytes / 1e6);
  }
  return 0;
}
