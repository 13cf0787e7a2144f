// Developer aid: host cost of enqueueing three small dependent kernels - three hipLaunchKernelGGL calls against one
// hipGraphLaunch of a captured three-node graph (what a pass of the gated update loop enqueues per iteration).
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/probe_graph tools/probe_graph.hip && /tmp/probe_graph
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_a(int *p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0]++; }
__global__ void k_b(int *p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[1]++; }
__global__ void k_c(int *p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[2]++; }
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  int *d = nullptr;
  hipMalloc(&d, 64);
  hipMemset(d, 0, 64);
  hipStream_t s;
  hipStreamCreate(&s);
  const int R = 2000;
  for (int w = 0; w < 2; w++) {
    double t0 = now_us();
    for (int i = 0; i < R; i++) {
      hipLaunchKernelGGL(k_a, dim3(1563), dim3(256), 0, s, d);
      hipLaunchKernelGGL(k_b, dim3(391), dim3(256), 0, s, d);
      hipLaunchKernelGGL(k_c, dim3(73), dim3(256), 0, s, d);
    }
    double t1 = now_us();
    hipStreamSynchronize(s);
    double t2 = now_us();
    if (w) printf("3 launches: %.2f us of host time per unit, %.2f us per unit until done\n", (t1 - t0) / R, (t2 - t0) / R);
  }
  hipGraph_t g;
  hipGraphExec_t ge;
  hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
  hipLaunchKernelGGL(k_a, dim3(1563), dim3(256), 0, s, d);
  hipLaunchKernelGGL(k_b, dim3(391), dim3(256), 0, s, d);
  hipLaunchKernelGGL(k_c, dim3(73), dim3(256), 0, s, d);
  hipStreamEndCapture(s, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  for (int w = 0; w < 2; w++) {
    double t0 = now_us();
    for (int i = 0; i < R; i++) hipGraphLaunch(ge, s);
    double t1 = now_us();
    hipStreamSynchronize(s);
    double t2 = now_us();
    if (w) printf("graph launch: %.2f us of host time per unit, %.2f us per unit until done\n", (t1 - t0) / R, (t2 - t0) / R);
  }
  return 0;
}
