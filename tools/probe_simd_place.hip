// Developer aid: on which SIMD of its CU does wave w of a 256-thread workgroup land? (k_pass leaves ONE wave per workgroup -
// wave 0 - running its 9 us plane-fit / row chain after the list walk; if every workgroup's wave 0 sat on the same SIMD, a CU's
// seven chains would share one SIMD's issue slots.)  Grid and LDS as k_pass at BASELINE config 2.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/probe_simd tools/probe_simd_place.hip && /tmp/probe_simd
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(7, 7))) k_where(unsigned *out, unsigned *xcc_out, int spin) {
  __shared__ volatile char pad[17296];
  pad[threadIdx.x] = (char)threadIdx.x;
  unsigned hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = hw, xcc_out[blockIdx.x * 4 + (threadIdx.x >> 6)] = xcc;
  long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < spin) {}  // keep the workgroups resident together, as a real pass does
  if (pad[threadIdx.x ^ 1] == 77) out[0] = 0;
}
int main() {
  const int nwg = 1563;
  unsigned *d, *dx;
  hipMalloc(&d, sizeof(unsigned) * nwg * 4);
  hipMalloc(&dx, sizeof(unsigned) * nwg * 4);
  std::vector<unsigned> h(nwg * 4), hx(nwg * 4);
  hipLaunchKernelGGL(k_where, dim3(nwg), dim3(256), 0, 0, d, dx, 20000);
  hipMemcpy(h.data(), d, sizeof(unsigned) * nwg * 4, hipMemcpyDeviceToHost);
  hipMemcpy(hx.data(), dx, sizeof(unsigned) * nwg * 4, hipMemcpyDeviceToHost);
  int distinct4 = 0, consecutive = 0;
  for (int b = 0; b < nwg; b++) {
    unsigned m = 0;
    for (int w = 0; w < 4; w++) m |= 1u << ((h[b * 4 + w] >> 4) & 3);
    distinct4 += m == 15;
    bool cons = true;
    for (int w = 1; w < 4; w++) cons = cons && (((h[b * 4 + w] >> 4) & 3) == ((((h[b * 4] >> 4) & 3) + w) & 3));
    consecutive += cons;
  }
  printf("workgroups whose four waves sit on four different SIMDs: %d of %d (on consecutive SIMDs from wave 0's: %d)\n", distinct4, nwg, consecutive);
  printf("first workgroups: (xcc, se/sh/cu bits 8-15, simd of waves 0..3)\n");
  for (int b = 0; b < 24; b++)
    printf("  wg %4d: xcc %u cu %02x  simd %u %u %u %u\n", b, hx[b * 4] & 15, (h[b * 4] >> 8) & 0xFF, (h[b * 4] >> 4) & 3, (h[b * 4 + 1] >> 4) & 3,
           (h[b * 4 + 2] >> 4) & 3, (h[b * 4 + 3] >> 4) & 3);
  // gfx9 HW_ID: wave_id [3:0], simd_id [5:4], pipe [7:6], cu_id [11:8], sh_id [12], se_id [15:13] (gfx950: se bits may be wider)
  long hist[4][4] = {};
  std::map<unsigned, std::vector<int>> per_cu;  // (everything above the simd field) -> wave-0 count per SIMD
  for (int b = 0; b < nwg; b++)
    for (int w = 0; w < 4; w++) {
      const unsigned hw = h[b * 4 + w], simd = (hw >> 4) & 3;
      hist[w][simd]++;
      if (w == 0) {
        auto &v = per_cu[((hx[b * 4] & 15) << 8) | ((hw >> 8) & 0xFF)];
        v.resize(4);
        v[simd]++;
      }
    }
  for (int w = 0; w < 4; w++) printf("wave %d of a workgroup: SIMD 0/1/2/3 = %ld / %ld / %ld / %ld\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
  int worst = 0, cus = 0;
  long sum_max = 0;
  for (auto &kv : per_cu) {
    int m = 0;
    for (int s = 0; s < 4; s++) m = kv.second[s] > m ? kv.second[s] : m;
    worst = m > worst ? m : worst, sum_max += m, cus++;
  }
  printf("%d CUs seen; wave 0s on the busiest SIMD of a CU: mean %.2f, worst %d (7 workgroups per CU: 1.75 would be even)\n", cus,
         (double)sum_max / cus, worst);
  int shown = 0;
  for (auto &kv : per_cu)
    if (shown++ < 6) printf("  cu %05x: wave 0 per SIMD %d %d %d %d\n", kv.first, kv.second[0], kv.second[1], kv.second[2], kv.second[3]);
  return 0;
}
