// H2D of pinned host memory: the copy engines (hipMemcpyAsync) against a kernel that reads the pinned buffer
// itself.  hipcc --offload-arch=gfx950 -O3 tools/h2d_probe.hip -o tools/_bin/h2d_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_pull(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

int main() {
  const size_t cap = 8u << 20;
  uint8_t* hp; void* dp;
  CK(hipHostMalloc((void**)&hp, cap, hipHostMallocDefault));
  CK(hipMalloc(&dp, cap));
  memset(hp, 1, cap);
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t sizes[] = {64u << 10, 700u << 10, 2730u << 10, 5300u << 10};
  for (size_t bytes : sizes) {
    for (int grid : {0, 32, 64, 128, 256, 512}) {
      float best = 1e9f, sum = 0; double host_best = 1e9;
      for (int it = 0; it < 12; it++) {
        CK(hipStreamSynchronize(s));
        const auto t0 = std::chrono::steady_clock::now();
        CK(hipEventRecord(e0, s));
        if (grid == 0) CK(hipMemcpyAsync(dp, hp, bytes, hipMemcpyHostToDevice, s));
        else hipLaunchKernelGGL(k_pull, dim3(grid), dim3(256), 0, s, (const uint4*)hp, (uint4*)dp, bytes / 16);
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        const double hw = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (it >= 2) { best = ms < best ? ms : best; sum += ms; host_best = hw < host_best ? hw : host_best; }
      }
      printf("%7zu KB %-12s grid %3d: device %.1f us best, %.1f avg (%.1f GB/s); host enqueue->sync %.1f us best\n", bytes >> 10,
             grid ? "k_pull" : "hipMemcpy", grid, best * 1e3, sum / 10 * 1e3, bytes / (best * 1e-3) / 1e9, host_best);
    }
  }
  return 0;
}
