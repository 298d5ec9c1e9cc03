// Micro-benchmark: WHERE do the waves of a launch shaped like asdr_update_kernel_mw land?  2,048 workgroups of four waves with 50,688 B of LDS
// (three workgroups per CU); every wave records HW_ID (SIMD, wave slot, CU, SE), XCC_ID and its start / end time, and spins ~20 us so that the
// first round of workgroups is resident together.  The host prints: which SIMDs the four waves of a workgroup sit on; which workgroup indices
// share a CU in the first round (and what that makes of a duty rotation `(wave - f(blockIdx)) & 3` for a few f); the XCD of workgroup b.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/placement.hip -o tools/ubench/placement && tools/ubench/placement
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <map>
#include <algorithm>
struct Rec { uint32_t hw_id, xcc_id; unsigned long long t0, t1; };
__global__ __launch_bounds__(256, 3) void k(Rec *out, unsigned long long spin_ticks) {
  __shared__ float lds[50688 / 4];
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  uint32_t hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  lds[threadIdx.x] = (float)hw;
  while (__builtin_amdgcn_s_memrealtime() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(4);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { Rec r; r.hw_id = hw; r.xcc_id = xcc + (uint32_t)(lds[threadIdx.x] < 0.0f); r.t0 = t0; r.t1 = __builtin_amdgcn_s_memrealtime(); out[blockIdx.x * 4 + (threadIdx.x >> 6)] = r; }
}
int main() {
  const int wgs = 2048;
  Rec *d; hipMalloc(&d, wgs * 4 * sizeof(Rec));
  std::vector<Rec> h(wgs * 4);
  for (int rep = 0; rep < 2; rep++) { k<<<wgs, 256>>>(d, 2000ull); hipDeviceSynchronize(); }   // 20 us at 100 MHz
  hipMemcpy(h.data(), d, h.size() * sizeof(Rec), hipMemcpyDeviceToHost);
  auto simd = [](uint32_t x) { return (x >> 4) & 3; };
  auto slot = [](uint32_t x) { return x & 15; };
  auto cu = [](uint32_t x) { return (x >> 8) & 15; };
  auto sh = [](uint32_t x) { return (x >> 12) & 1; };
  auto se = [](uint32_t x) { return (x >> 13) & 7; };
  // 1. SIMDs of the four waves of a workgroup
  std::map<std::string, int> pat;
  for (int b = 0; b < wgs; b++) { char s[16]; snprintf(s, sizeof s, "%u%u%u%u", simd(h[4 * b].hw_id), simd(h[4 * b + 1].hw_id), simd(h[4 * b + 2].hw_id), simd(h[4 * b + 3].hw_id)); pat[s]++; }
  printf("SIMD of waves 0..3 of a workgroup: "); for (auto &p : pat) printf("%s x %d  ", p.first.c_str(), p.second); printf("\n");
  // 2. XCD by workgroup index
  printf("XCC_ID (low 4 bits) of workgroups 0..23: "); for (int b = 0; b < 24; b++) printf("%u ", h[4 * b].xcc_id & 15); printf("\n");
  // 3. first round: workgroups sharing a CU
  unsigned long long tmin = ~0ull; for (auto &r : h) tmin = std::min(tmin, r.t0);
  std::map<uint64_t, std::vector<int>> bycu;
  for (int b = 0; b < wgs; b++) {
    if (h[4 * b].t0 - tmin > 500) continue;   // started within 5 us of the first: the first round
    const uint32_t x = h[4 * b].hw_id;
    bycu[((uint64_t)(h[4 * b].xcc_id & 15) << 16) | (se(x) << 8) | (sh(x) << 4) | cu(x)].push_back(b);
  }
  printf("first round: %zu CUs hold workgroups; examples (xcc.se.sh.cu: workgroups [wave-0 SIMD/slot]):\n", bycu.size());
  int shown = 0; std::map<int, int> nper;
  for (auto &c : bycu) {
    nper[(int)c.second.size()]++;
    if (shown++ < 12) { printf("  %llx: ", (unsigned long long)c.first); for (int b : c.second) printf("%d[%u/%u] ", b, simd(h[4 * b].hw_id), slot(h[4 * b].hw_id)); printf("\n"); }
  }
  printf("workgroups per CU in the first round: "); for (auto &p : nper) printf("%d x %d  ", p.first, p.second); printf("\n");
  // 4. what a rotation makes of it: for every CU and SIMD, the duties of the resident waves (the wave on SIMD s of workgroup b has wave index w:
  //    duty = (w - f(b)) & 3); count CUs x SIMDs whose resident waves all have DIFFERENT duties / all the SAME duty
  const char *fn[] = {"b", "b >> 3", "(b >> 3) + (b >> 8)", "b * 0x9E3779B1 >> 30", "(b >> 3) % 3"};
  for (int f = 0; f < 5; f++) {
    int same = 0, diff = 0, other = 0;
    for (auto &c : bycu) {
      for (uint32_t s = 0; s < 4; s++) {
        std::vector<int> duties;
        for (int b : c.second) for (int w = 0; w < 4; w++) if (simd(h[4 * b + w].hw_id) == s) {
          const uint32_t rot = f == 0 ? (uint32_t)b : f == 1 ? (uint32_t)b >> 3 : f == 2 ? ((uint32_t)b >> 3) + ((uint32_t)b >> 8) : f == 3 ? ((uint32_t)b * 0x9E3779B1u) >> 30 : ((uint32_t)b >> 3) % 3u;
          duties.push_back((int)((w - rot) & 3));
        }
        if (duties.size() < 2) continue;
        std::sort(duties.begin(), duties.end());
        const bool all_same = duties.front() == duties.back();
        const bool all_diff = std::adjacent_find(duties.begin(), duties.end()) == duties.end();
        if (all_same) same++; else if (all_diff) diff++; else other++;
      }
    }
    printf("rotation f(b) = %-22s: SIMDs whose resident waves have all the same duty %d, all different %d, mixed %d\n", fn[f], same, diff, other);
  }
  // 5. later rounds: the workgroup that takes over a freed slot -- index distance to the one that left (dynamic assignment)
  printf("launch span %.1f us\n", (double)(std::max_element(h.begin(), h.end(), [](const Rec &a, const Rec &b) { return a.t1 < b.t1; })->t1 - tmin) / 100.0);
  return 0;
}
