// launch_probe.hip — host cost of one kernel launch by the ways HIP offers, on an MI355X box:
//   hipLaunchKernelGGL (what launch_k uses), hipModuleLaunchKernel on a function handle looked up once
//   (hipGetFuncBySymbol) with a kernelParams array or a packed argument buffer
//   (HIP_LAUNCH_PARAM_BUFFER_POINTER), round-robin over 4 streams, and two host threads launching at
//   once into their own streams.
//   hipcc --offload-arch=gfx950 -O2 -pthread tools/launch_probe.hip -o tools/_bin/launch_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

struct Geom {
  int w, h, tw, th, nb, pad[7];
};

__global__ void k_many(float* p, const float* a, const float* b, const unsigned* c, unsigned n, unsigned m, Geom g,
                       unsigned* d, unsigned* e, const unsigned* f, unsigned* h, float s, int which) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (int)n && g.w > 0) p[i] = p[i] * s + which;
}

#define CK(x)                                                      \
  do {                                                             \
    hipError_t e_ = (x);                                           \
    if (e_ != hipSuccess) {                                        \
      printf("%s failed: %s\n", #x, hipGetErrorString(e_));        \
      return 1;                                                    \
    }                                                              \
  } while (0)

using clk = std::chrono::steady_clock;
static double us(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); }

int main() {
  const int reps = 4000, kBurst = 16;
  float* p;
  CK(hipMalloc(&p, 1 << 20));
  unsigned* u;
  CK(hipMalloc(&u, 1 << 20));
  hipStream_t st[4];
  for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  Geom g = {640, 480, 32, 16, 1201, {}};
  unsigned n = 4096, m = 7;
  float sc = 1.f;
  const float* cp = p;
  const unsigned* cu = u;
  hipFunction_t fn;
  CK(hipGetFuncBySymbol(&fn, (const void*)k_many));
  auto plain = [&](hipStream_t s, int which) {
    hipLaunchKernelGGL(k_many, dim3(16), dim3(256), 0, s, p, cp, cp, cu, n, m, g, u, u, cu, u, sc, which);
  };
  auto module_params = [&](hipStream_t s, int which) {
    void* args[13] = {&p, &cp, &cp, &cu, &n, &m, &g, &u, &u, &cu, &u, &sc, &which};
    (void)hipModuleLaunchKernel(fn, 16, 1, 1, 256, 1, 1, 0, s, args, nullptr);
  };
  struct __attribute__((packed, aligned(8))) Packed {
    float* p; const float* a; const float* b; const unsigned* c; unsigned n, m; Geom g;
    unsigned* d; unsigned* e; const unsigned* f; unsigned* h; float s; int which;
  };
  auto module_buffer = [&](hipStream_t s, int which) {
    Packed a = {p, cp, cp, cu, n, m, g, u, u, cu, u, sc, which};
    size_t sz = sizeof(a);
    void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
    (void)hipModuleLaunchKernel(fn, 16, 1, 1, 256, 1, 1, 0, s, nullptr, cfg);
  };
  auto run = [&](const char* name, auto&& launch, int nstreams) {
    double best = 1e30;
    for (int w = 0; w < 3; w++) {
      double tot = 0;
      for (int r = 0; r < reps; r += kBurst) {
        auto t0 = clk::now();
        for (int q = 0; q < kBurst; q++) launch(st[q % nstreams], q);
        tot += us(t0, clk::now());
        for (int k = 0; k < nstreams; k++) (void)hipStreamSynchronize(st[k]);
      }
      best = std::min(best, tot / reps);
    }
    printf("%-44s %d stream(s): %.2f us per launch\n", name, nstreams, best);
  };
  run("hipLaunchKernelGGL", plain, 1);
  run("hipModuleLaunchKernel, kernelParams", module_params, 1);
  run("hipModuleLaunchKernel, packed buffer", module_buffer, 1);
  run("hipLaunchKernelGGL", plain, 4);
  run("hipModuleLaunchKernel, kernelParams", module_params, 4);
  CK(hipGetLastError());
  // two host threads, each into its own stream
  for (int form = 0; form < 2; form++) {
    double per[2] = {0, 0};
    auto worker = [&](int t) {
      double tot = 0;
      for (int r = 0; r < reps; r += kBurst) {
        auto t0 = clk::now();
        for (int q = 0; q < kBurst; q++)
          if (form) module_params(st[t], q);
          else plain(st[t], q);
        tot += us(t0, clk::now());
        (void)hipStreamSynchronize(st[t]);
      }
      per[t] = tot / reps;
    };
    std::thread a(worker, 0), b(worker, 1);
    a.join();
    b.join();
    printf("two threads at once, %s: %.2f / %.2f us per launch\n", form ? "hipModuleLaunchKernel" : "hipLaunchKernelGGL", per[0], per[1]);
  }
  // event record + wait, for scale
  {
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    auto t0 = clk::now();
    for (int r = 0; r < reps; r++) {
      CK(hipEventRecord(ev, st[0]));
      CK(hipStreamWaitEvent(st[1], ev, 0));
    }
    printf("hipEventRecord + hipStreamWaitEvent: %.2f us\n", us(t0, clk::now()) / reps);
    CK(hipDeviceSynchronize());
  }
  return 0;
}
