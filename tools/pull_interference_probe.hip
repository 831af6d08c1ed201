// Does a kernel that reads pinned HOST memory slow other kernels down while it runs?  (rocprofv3 of plain calls from
// pageable memory: k_tile_hist 6 -> 40 us, k_tile_apply 11 -> 40 us beside k_stage_pull_packed.)
//   victim A: streams 2.67 MB of 16-byte records out of HBM and counts them into a 1200-word table (like k_tile_hist)
//   victim B: 4096 threads, each a dependent chain of 64 loads over 64 MB (latency)
//   puller  : grid g, reads the pinned buffer round after round until told to stop
// hipcc --offload-arch=gfx950 -O3 tools/pull_interference_probe.hip -o tools/_bin/pull_interference_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_victim_hist(const uint4* __restrict__ ev, size_t n, unsigned* __restrict__ tab) {
  __shared__ unsigned h[1200];
  for (int i = threadIdx.x; i < 1200; i += 256) h[i] = 0;
  __syncthreads();
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint4 v = ev[i];
    atomicAdd(&h[(v.x ^ v.y ^ (unsigned)i) % 1200u], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 1200; i += 256) if (h[i]) atomicAdd(&tab[i], h[i]);
}

__global__ __launch_bounds__(64) void k_victim_chase(const unsigned* __restrict__ next, unsigned* out) {
  unsigned p = (blockIdx.x * 64 + threadIdx.x) * 4099u;
  for (int k = 0; k < 64; k++) p = next[p & ((16u << 20) - 1)];
  if (p == 0xffffffffu) *out = p;
}

template <int W>
__global__ __launch_bounds__(256) void k_puller(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, size_t bytes,
                                                const volatile unsigned* stop, int store) {
  const size_t n = bytes / W;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (int round = 0; round < 100000; round++) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
      if (W == 16) {
        const uint4 v = ((const uint4*)src)[i];
        if (store || v.x == 0x12345u) ((uint4*)dst)[i] = v;
      } else {
        const uint2 v = ((const uint2*)src)[i];
        if (store || v.x == 0x12345u) ((uint4*)dst)[i] = make_uint4(v.x, v.y, v.x, v.y);
      }
    }
    if (*stop) return;
  }
}

int main() {
  const size_t evb = 2730u << 10, pullb = 1365u << 10;
  uint8_t* hp; uint8_t *dp, *ev; unsigned *tab, *next, *out; unsigned* stop_h;
  CK(hipHostMalloc((void**)&hp, 8u << 20, hipHostMallocDefault));
  CK(hipHostMalloc((void**)&stop_h, 64, hipHostMallocDefault));
  CK(hipMalloc((void**)&dp, 16u << 20));
  CK(hipMalloc((void**)&ev, evb));
  CK(hipMalloc((void**)&tab, 1200 * 4));
  CK(hipMalloc((void**)&next, 64u << 20));
  CK(hipMalloc((void**)&out, 4));
  memset(hp, 1, 8u << 20);
  CK(hipMemset(ev, 3, evb));
  CK(hipMemset(tab, 0, 4800));
  {
    std::vector<unsigned> nx(16u << 20);
    unsigned x = 12345;
    for (auto& v : nx) { x = x * 1664525u + 1013904223u; v = x >> 8; }
    CK(hipMemcpy(next, nx.data(), 64u << 20, hipMemcpyHostToDevice));
  }
  hipStream_t sv, sp;
  CK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sp, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto victim = [&](int which, float* best, float* avg) -> int {
    *best = 1e9f; float sum = 0;
    for (int it = 0; it < 22; it++) {
      CK(hipEventRecord(e0, sv));
      if (which == 0) hipLaunchKernelGGL(k_victim_hist, dim3(512), dim3(256), 0, sv, (const uint4*)ev, evb / 16, tab);
      else hipLaunchKernelGGL(k_victim_chase, dim3(64), dim3(64), 0, sv, (const unsigned*)next, out);
      CK(hipEventRecord(e1, sv));
      CK(hipStreamSynchronize(sv));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (it >= 2) { *best = ms < *best ? ms : *best; sum += ms; }
    }
    *avg = sum / 20;
    return 0;
  };
  float b, a;
  for (int which = 0; which < 2; which++) {
    if (victim(which, &b, &a)) return 1;
    printf("victim %s alone: %.1f us best, %.1f avg\n", which ? "chase" : "hist ", b * 1e3, a * 1e3);
  }
  struct Cfg { int grid, w, store; };
  const Cfg cfgs[] = {{64, 8, 1}, {64, 16, 1}, {64, 8, 0}, {32, 8, 1}, {16, 8, 1}, {8, 8, 1}, {128, 8, 1}, {256, 8, 1}};
  for (const Cfg& c : cfgs) {
    *stop_h = 0;
    // the puller's own rate, alone
    hipEvent_t p0, p1; CK(hipEventCreate(&p0)); CK(hipEventCreate(&p1));
    if (c.w == 8) hipLaunchKernelGGL(k_puller<8>, dim3(c.grid), dim3(256), 0, sp, hp, dp, pullb, stop_h, c.store);
    else hipLaunchKernelGGL(k_puller<16>, dim3(c.grid), dim3(256), 0, sp, hp, dp, 2 * pullb, stop_h, c.store);
    std::this_thread::sleep_for(std::chrono::milliseconds(2));
    float r[2][2];
    for (int which = 0; which < 2; which++)
      if (victim(which, &r[which][0], &r[which][1])) return 1;
    *stop_h = 1;
    CK(hipStreamSynchronize(sp));
    printf("beside puller grid %3d, %2d-byte loads, %s: hist %.1f us best %.1f avg; chase %.1f best %.1f avg\n", c.grid, c.w,
           c.store ? "stores   " : "no stores", r[0][0] * 1e3, r[0][1] * 1e3, r[1][0] * 1e3, r[1][1] * 1e3);
  }
  // the puller's rate per grid, alone (one round)
  for (int grid : {8, 16, 32, 64, 128}) {
    *stop_h = 1;  // (one round)
    float best = 1e9f;
    for (int it = 0; it < 8; it++) {
      CK(hipEventRecord(e0, sp));
      hipLaunchKernelGGL(k_puller<8>, dim3(grid), dim3(256), 0, sp, hp, dp, pullb, stop_h, 1);
      CK(hipEventRecord(e1, sp));
      CK(hipStreamSynchronize(sp));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (it >= 2 && ms < best) best = ms;
    }
    printf("puller alone, grid %3d, 8-byte loads, 1.33 MB: %.1f us (%.1f GB/s)\n", grid, best * 1e3, pullb / (best * 1e-3) / 1e9);
  }
  return 0;
}
