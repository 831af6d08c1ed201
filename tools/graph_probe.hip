// graph_probe.hip — host cost of N small launches vs one hipGraphLaunch of the same N kernel nodes
//   hipcc --offload-arch=gfx950 -O2 tools/graph_probe.hip -o /tmp/graph_probe && /tmp/graph_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <vector>

struct Desc {
  int n;
  float scale;
};

__global__ void k_small(float* p, const Desc* d, int which) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < d->n) p[i] = p[i] * d->scale + which;
}

#define CK(x)                                                              \
  do {                                                                     \
    hipError_t e = (x);                                                    \
    if (e != hipSuccess) {                                                 \
      printf("%s failed: %s\n", #x, hipGetErrorString(e));                 \
      return 1;                                                            \
    }                                                                      \
  } while (0)

int main() {
  const int N = 13, reps = 2000, kBurst = 8;
  float* p;
  Desc* d;
  CK(hipMalloc(&p, 1 << 20));
  CK(hipHostMalloc(&d, sizeof(Desc)));
  d->n = 1 << 16;
  d->scale = 1.f;
  hipStream_t s, s2;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t ev;
  CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  using clk = std::chrono::steady_clock;
  auto us = [](clk::time_point a, clk::time_point b) {
    return std::chrono::duration<double, std::micro>(b - a).count();
  };
  // plain launches
  for (int w = 0; w < 2; w++) {
    double tot = 0;
    for (int r = 0; r < reps; r += kBurst) {  // enqueue time only: bursts into an idle stream
      auto t0 = clk::now();
      for (int q = 0; q < kBurst; q++)
        for (int k = 0; k < N; k++) hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s, p, d, k);
      tot += us(t0, clk::now());
      CK(hipStreamSynchronize(s));
    }
    if (w) printf("%d plain launches: %.2f us host per sequence (%.2f us per launch)\n", N, tot / reps, tot / reps / N);
  }
  // event record + stream wait
  {
    auto t0 = clk::now();
    for (int r = 0; r < reps; r++) {
      CK(hipEventRecord(ev, s));
      CK(hipStreamWaitEvent(s2, ev, 0));
    }
    printf("hipEventRecord + hipStreamWaitEvent: %.2f us\n", us(t0, clk::now()) / reps);
    CK(hipStreamSynchronize(s));
    CK(hipStreamSynchronize(s2));
  }
  // graph by capture
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int k = 0; k < N; k++) hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s, p, d, k);
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int w = 0; w < 2; w++) {
    double tot = 0;
    for (int r = 0; r < reps; r += kBurst) {
      auto t0 = clk::now();
      for (int q = 0; q < kBurst; q++) CK(hipGraphLaunch(ge, s));
      tot += us(t0, clk::now());
      CK(hipStreamSynchronize(s));
    }
    if (w) printf("hipGraphLaunch of %d kernel nodes: %.2f us host per launch\n", N, tot / reps);
  }
  // explicit graph, every node's params rewritten before each launch
  {
    hipGraph_t g2;
    CK(hipGraphCreate(&g2, 0));
    std::vector<hipGraphNode_t> nodes(N);
    std::vector<int> which(N);
    const Desc* dc = d;
    for (int k = 0; k < N; k++) {
      which[k] = k;
      void* args[3] = {&p, &dc, &which[k]};
      hipKernelNodeParams kp = {};
      kp.func = (void*)k_small;
      kp.gridDim = dim3(1);
      kp.blockDim = dim3(64);
      kp.kernelParams = args;
      CK(hipGraphAddKernelNode(&nodes[k], g2, k ? &nodes[k - 1] : nullptr, k ? 1 : 0, &kp));
    }
    hipGraphExec_t ge2;
    CK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
    for (int w = 0; w < 2; w++) {
      double tot = 0;
      for (int r = 0; r < reps; r += kBurst) {
        auto t0 = clk::now();
        for (int q = 0; q < kBurst; q++) {
          for (int k = 0; k < N; k++) {
            int wv = k + r + q;
            void* args[3] = {&p, &dc, &wv};
            hipKernelNodeParams kp = {};
            kp.func = (void*)k_small;
            kp.gridDim = dim3(1 + (q & 1));
            kp.blockDim = dim3(64);
            kp.kernelParams = args;
            CK(hipGraphExecKernelNodeSetParams(ge2, nodes[k], &kp));
          }
          CK(hipGraphLaunch(ge2, s));
        }
        tot += us(t0, clk::now());
        CK(hipStreamSynchronize(s));
      }
      if (w) printf("%d x hipGraphExecKernelNodeSetParams + hipGraphLaunch: %.2f us host\n", N, tot / reps);
    }
  }
  // GPU-side duration of both forms
  {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    float ms;
    CK(hipEventRecord(a, s));
    for (int r = 0; r < 200; r++)
      for (int k = 0; k < N; k++) hipLaunchKernelGGL(k_small, dim3(256), dim3(256), 0, s, p, d, k);
    CK(hipEventRecord(b, s));
    CK(hipEventSynchronize(b));
    CK(hipEventElapsedTime(&ms, a, b));
    printf("GPU time, plain: %.2f us per sequence\n", ms * 1e3 / 200);
    CK(hipEventRecord(a, s));
    for (int r = 0; r < 200; r++) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(b, s));
    CK(hipEventSynchronize(b));
    CK(hipEventElapsedTime(&ms, a, b));
    printf("GPU time, graph: %.2f us per sequence\n", ms * 1e3 / 200);
  }
  return 0;
}
