// Developer aid: does a wave64 VALU instruction cost less when only 16 (or 32) of its lanes are active?
// One wave, a chain of dependent operations (f32 mul+add, f64 mul+add, f32 division, f64 division), timed with
// s_memtime for 64 / 32 / 16 active lanes (the low lanes) and for 16 lanes spread one per quad-of-four.
// The question behind it: would splitting one wave's 64-query plane-fit chain over four waves of 16 queries shorten it?
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -o /tmp/probe_exec tools/probe_exec_mask.hip && /tmp/probe_exec
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND>
__global__ void k_chain(long long *out, const float *in, int active, int spread, int n) {
  const int lane = threadIdx.x;
  const bool on = spread ? (lane % (64 / active) == 0) : lane < active;
  float a = in[lane], b = in[64 + lane];
  double da = a, db = b;
  long long t0 = 0, t1 = 0;
  __builtin_amdgcn_s_waitcnt(0);
  t0 = __builtin_readcyclecounter();
  if (on) {
    for (int i = 0; i < n; i++) {
      if (KIND == 0) a = a * b + 0.5f, a = a * b - 0.25f, a = a * b + 0.5f, a = a * b - 0.25f;
      if (KIND == 1) da = da * db + 0.5, da = da * db - 0.25, da = da * db + 0.5, da = da * db - 0.25;
      if (KIND == 2) a = 1.0f / (a + 1.5f), a = 1.0f / (a + 1.5f), a = 1.0f / (a + 1.5f), a = 1.0f / (a + 1.5f);
      if (KIND == 3) da = 1.0 / (da + 1.5), da = 1.0 / (da + 1.5), da = 1.0 / (da + 1.5), da = 1.0 / (da + 1.5);
    }
  }
  t1 = __builtin_readcyclecounter();
  if (on) out[64 + lane] = (long long)(a + (float)da);
  if (lane == 0) out[0] = t1 - t0;
}
int main() {
  long long *d_out, h[1];
  float *d_in, hin[128];
  for (int i = 0; i < 128; i++) hin[i] = 0.9f + 0.001f * i;
  hipMalloc(&d_out, 8 * 256);
  hipMalloc(&d_in, sizeof(hin));
  hipMemcpy(d_in, hin, sizeof(hin), hipMemcpyHostToDevice);
  const char *names[4] = {"f32 mul+add", "f64 mul+add", "f32 division", "f64 division"};
  const int n = 2000;
  for (int kind = 0; kind < 4; kind++)
    for (int spread = 0; spread < 2; spread++)
      for (int active : {64, 32, 16, 4, 1}) {
        if (spread && active == 64) continue;
        long long best = 1LL << 60;
        for (int rep = 0; rep < 5; rep++) {
          if (kind == 0) hipLaunchKernelGGL(k_chain<0>, dim3(1), dim3(64), 0, 0, d_out, d_in, active, spread, n);
          if (kind == 1) hipLaunchKernelGGL(k_chain<1>, dim3(1), dim3(64), 0, 0, d_out, d_in, active, spread, n);
          if (kind == 2) hipLaunchKernelGGL(k_chain<2>, dim3(1), dim3(64), 0, 0, d_out, d_in, active, spread, n);
          if (kind == 3) hipLaunchKernelGGL(k_chain<3>, dim3(1), dim3(64), 0, 0, d_out, d_in, active, spread, n);
          hipMemcpy(h, d_out, 8, hipMemcpyDeviceToHost);
          if (h[0] < best) best = h[0];
        }
        printf("%-13s %2d lanes %-8s: %7.2f ticks of s_memtime per dependent operation\n", names[kind], active,
               spread ? "(spread)" : "(low)", (double)best / (4.0 * n));
      }
  return 0;
}
