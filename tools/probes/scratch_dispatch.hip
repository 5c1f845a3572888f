// Does a kernel that uses scratch (private segment) pay a dispatch penalty?  Chains of dependent launches, with and
// without a few bytes of scratch, alone and alternating with a non-scratch kernel, timed with events.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/scratch_dispatch.hip -o tools/probes/scratch_dispatch
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k_plain(float* p, int n) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = p[i] * 1.0001f + 1.f;
}
__global__ __launch_bounds__(256) void k_scratch(float* p, int n, int j) {
  volatile float loc[8];  // forced to the private segment by the dynamic index
  int i = blockIdx.x * 256 + threadIdx.x;
  for (int u = 0; u < 8; ++u) loc[u] = (float)(u + i);
  float v = loc[j & 7] + loc[(j + 3) & 7];
  if (i < n) p[i] = p[i] * 1.0001f + v * 0.f + 1.f;
}
int main() {
  const int n = 256 * 256;
  float* p;
  hipMalloc(&p, n * 4);
  hipMemset(p, 0, n * 4);
  hipStream_t st, st2;
  hipStreamCreate(&st);
  hipStreamCreate(&st2);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto run = [&](const char* name, int mode, bool graph) {
    const int N = 200;
    auto body = [&](hipStream_t s) {
      for (int i = 0; i < N; ++i) {
        if (mode == 0 || (mode == 2 && (i & 1))) hipLaunchKernelGGL(k_plain, dim3(256), dim3(256), 0, s, p, n);
        else hipLaunchKernelGGL(k_scratch, dim3(256), dim3(256), 0, s, p, n, i);
      }
    };
    float ms = 0;
    if (graph) {
      hipGraph_t g; hipGraphExec_t ge;
      hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
      body(st);
      hipStreamEndCapture(st, &g);
      hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
      hipGraphLaunch(ge, st); hipStreamSynchronize(st);
      hipEventRecord(e0, st); hipGraphLaunch(ge, st); hipEventRecord(e1, st); hipStreamSynchronize(st);
      hipEventElapsedTime(&ms, e0, e1);
    } else {
      body(st); hipStreamSynchronize(st);
      hipEventRecord(e0, st); body(st); hipEventRecord(e1, st); hipStreamSynchronize(st);
      hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%-44s %s  %.2f us per launch\n", name, graph ? "graph" : "eager", ms * 1e3 / N);
  };
  for (int g = 0; g < 2; ++g) {
    run("plain kernel chain", 0, g);
    run("scratch kernel chain", 1, g);
    run("alternating scratch / plain", 2, g);
  }
  // two streams: a scratch kernel on one stream while plain kernels run on another (graph with a fork)
  {
    hipGraph_t g; hipGraphExec_t ge; hipEvent_t f, j;
    hipEventCreate(&f); hipEventCreate(&j);
    for (int mode = 0; mode < 2; ++mode) {
      hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
      hipEventRecord(f, st); hipStreamWaitEvent(st2, f, 0);
      for (int i = 0; i < 100; ++i) {
        hipLaunchKernelGGL(k_plain, dim3(256), dim3(256), 0, st, p, n);
        if (mode) hipLaunchKernelGGL(k_scratch, dim3(256), dim3(256), 0, st2, p + n / 2, n / 2, i);
        else hipLaunchKernelGGL(k_plain, dim3(256), dim3(256), 0, st2, p + n / 2, n / 2);
      }
      hipEventRecord(j, st2); hipStreamWaitEvent(st, j, 0);
      hipStreamEndCapture(st, &g);
      hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
      hipGraphLaunch(ge, st); hipStreamSynchronize(st);
      float ms;
      hipEventRecord(e0, st); hipGraphLaunch(ge, st); hipEventRecord(e1, st); hipStreamSynchronize(st);
      hipEventElapsedTime(&ms, e0, e1);
      printf("two streams x 100, second stream %-8s graph  %.2f us per pair\n", mode ? "scratch" : "plain", ms * 1e3 / 100);
    }
  }
  return 0;
}
