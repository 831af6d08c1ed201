// hist probe: where do k_tile_hist's 37 us at C5's batch size (6.7 M events, 107 MB) go?  The same
// geometry (410 blocks x 1024 threads, scatter blocks of 4096 events, 1841 buckets) with the parts
// taken away one at a time, and the plain streaming read of the same bytes in other launch shapes.
//   hipcc --offload-arch=gfx950 -O3 -o tools/hist_probe tools/hist_probe.hip && tools/hist_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kBins = 2048, kT = 1024;
struct Geom {
  int W, H, tiles_x, nt_cam, nbins;
};
__device__ __forceinline__ uint32_t bin_of(const Geom& g, uint32_t xy, bool right) {
  const uint32_t x = xy & 0xffffu, y = xy >> 16;
  if (x >= (uint32_t)g.W || y >= (uint32_t)g.H) return (uint32_t)g.nbins - 1u;
  return (right ? (uint32_t)g.nt_cam : 0u) + (y >> 5) * (uint32_t)g.tiles_x + (x >> 5);
}
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// MODE bits: 1 = LDS atomics, 2 = P rows written, 4 = time range reduction
template <int MODE>
__global__ __launch_bounds__(kT) void k_hist(const uint4* __restrict__ ev, const uint4* __restrict__ evR, uint32_t nL, uint32_t n, Geom g, uint32_t te,
                                             uint32_t nblk, uint32_t group, uint32_t* __restrict__ Pm,
                                             uint32_t* __restrict__ Tm, uint32_t* __restrict__ meta) {
  constexpr int UE = 4;
  __shared__ uint32_t h[kBins], run[kBins];
  const int nb = g.nbins;
  for (int i = threadIdx.x; i < kBins; i += kT) {
    h[i] = 0;
    run[i] = 0;
  }
  uint32_t tmin = 0xffffffffu, tmax = 0, tor = 0;
  uint4 ne[UE];
  auto request = [&](uint32_t k) {
    const uint32_t b = blockIdx.x * group + k;
    const uint32_t lo = b * te, hi = (k < group && b < nblk) ? min(lo + te, n) : lo;
#pragma unroll
    for (int j = 0; j < UE; j++) {
      const uint32_t i = lo + threadIdx.x + j * kT;
      ne[j] = make_uint4(0, 0, 0, 0);
      if (j * kT < (int)te && i < hi) ne[j] = i >= nL ? evR[i - nL] : ev[i];
    }
  };
  request(0);
  __syncthreads();
  for (uint32_t k = 0; k < group; k++) {
    const uint32_t b = blockIdx.x * group + k;
    const bool live = b < nblk;
    const uint32_t lo = b * te, hi = live ? min(lo + te, n) : lo;
    uint32_t bins[UE];
#pragma unroll
    for (int j = 0; j < UE; j++) {
      const uint32_t i = lo + threadIdx.x + j * kT;
      bins[j] = 0xffffffffu;
      if (j * kT < (int)te && i < hi) {
        bins[j] = bin_of(g, ne[j].x, i >= nL);
        if ((MODE & 4) && bins[j] != (uint32_t)nb - 1u) {
          tmin = min(tmin, ne[j].y);
          tmax = max(tmax, ne[j].y);
          tor |= ne[j].z;
        }
        if (!(MODE & 4)) tor ^= ne[j].y ^ ne[j].z ^ ne[j].w;
      }
    }
    if (k + 1 < group) request(k + 1);
    if (MODE & 1) {
#pragma unroll
      for (int j = 0; j < UE; j++)
        if (bins[j] != 0xffffffffu) atomicAdd(&h[bins[j]], 1u);
    } else {
#pragma unroll
      for (int j = 0; j < UE; j++) tor += bins[j];
    }
    if (MODE & 3) lds_barrier();
    if ((MODE & 2) && live)
      for (int i = threadIdx.x; i < nb; i += kT) {
        const uint32_t r = run[i];
        Pm[(size_t)b * nb + i] = r;
        run[i] = r + h[i];
        h[i] = 0;
      }
    if (MODE & 3) lds_barrier();
  }
  if (MODE & 2)
    for (int i = threadIdx.x; i < nb; i += kT) Tm[(size_t)blockIdx.x * nb + i] = run[i];
  if (tmin == 12345u || tmax == 0xfffffff0u || tor == 0x13579bdfu) meta[0] = tmin + tmax + tor;  // keeps the loads alive
}

// the plain stream: every thread takes U records per step, grid-stride
template <int U, int T>
__global__ __launch_bounds__(T) void k_read(const uint4* __restrict__ ev, uint32_t n, uint32_t* __restrict__ meta) {
  uint32_t acc = 0;
  for (uint32_t base = blockIdx.x * (uint32_t)(T * U); base < n; base += gridDim.x * (uint32_t)(T * U)) {
    uint4 e[U];
#pragma unroll
    for (int j = 0; j < U; j++) {
      const uint32_t i = base + threadIdx.x + j * T;
      e[j] = i < n ? ev[i] : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < U; j++) acc ^= e[j].x ^ e[j].y ^ e[j].z ^ e[j].w;
  }
  if (acc == 0x13579bdfu) meta[0] = acc;
}

__global__ __launch_bounds__(256) void k_dirty(uint4* __restrict__ p, uint32_t n, uint32_t v) {
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) p[i] = make_uint4(v, i, v, i);
}

#define CK(x)                                                        \
  do {                                                               \
    hipError_t e_ = (x);                                             \
    if (e_ != hipSuccess) {                                          \
      printf("%s: %s\n", #x, hipGetErrorString(e_));                 \
      return 1;                                                      \
    }                                                                \
  } while (0)

int main(int argc, char** argv) {
  const uint32_t n = argc > 1 ? (uint32_t)atol(argv[1]) : 6700000u, nL = argc > 2 ? (uint32_t)atol(argv[2]) : n / 2;
  const int dirty_mb = argc > 4 ? atoi(argv[4]) : 0;  // written by another kernel before every timed launch
  const int skew = argc > 3 ? atoi(argv[3]) : 0;  // percent of the events that fall into 1/16 of the sensor
  const int NB = 4, iters = 24;
  Geom g{1280, 720, 40, 920, 1841};
  std::vector<uint4> h(n);
  uint32_t s = 12345;
  auto rnd = [&]() { return s = s * 1664525u + 1013904223u, s >> 8; };
  for (uint32_t i = 0; i < n; i++) h[i] = make_uint4(((int)(rnd() % 100) < skew ? (rnd() % 320) | ((rnd() % 180) << 16) : (rnd() % 1280) | ((rnd() % 720) << 16)), 1700000000u, rnd() % 1000000000u, rnd() & 1);
  uint4* ev[NB];
  for (int b = 0; b < NB; b++) {
    CK(hipMalloc(&ev[b], (size_t)n * 16));
    CK(hipMemcpy(ev[b], h.data(), (size_t)n * 16, hipMemcpyHostToDevice));
  }
  const uint32_t te = 4096, nblk = (n + te - 1) / te;
  uint32_t *P, *T, *meta;
  CK(hipMalloc(&P, (size_t)nblk * g.nbins * 4));
  CK(hipMalloc(&T, (size_t)nblk * g.nbins * 4));
  CK(hipMalloc(&meta, 64));
  uint4* scratch = nullptr;
  if (dirty_mb) CK(hipMalloc(&scratch, (size_t)dirty_mb << 20));
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  auto timeit = [&](const char* name, auto launch) {
    for (int i = 0; i < 3; i++) launch(ev[i % NB]);
    hipDeviceSynchronize();
    float best = 1e9f, sum = 0;
    for (int i = 0; i < iters; i++) {
      if (dirty_mb) hipLaunchKernelGGL(k_dirty, dim3(2048), dim3(256), 0, 0, scratch, (uint32_t)(((size_t)dirty_mb << 20) / 16), (uint32_t)i);
      hipEventRecord(a);
      launch(ev[i % NB]);
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms;
      hipEventElapsedTime(&ms, a, b);
      best = ms < best ? ms : best;
      sum += ms;
    }
    const double mb = n * 16.0 / 1e6;
    printf("%-44s avg %7.2f us  min %7.2f us   %5.2f TB/s of the records (min)\n", name, sum / iters * 1e3, best * 1e3, mb / (best * 1e3));
  };
  printf("n %u nL %u skew %d%% dirty %d MB\n", n, nL, skew, dirty_mb);
  for (uint32_t group : {4u}) {
    const uint32_t nseg = (nblk + group - 1) / group;
    char nm[96];
    snprintf(nm, sizeof nm, "hist full (atomics+P+range) group %u, %u blocks", group, nseg);
    timeit(nm, [&](uint4* e) { hipLaunchKernelGGL(k_hist<7>, dim3(nseg), dim3(kT), 0, 0, e, e + nL, nL, n, g, te, nblk, group, P, T, meta); });
    snprintf(nm, sizeof nm, "hist atomics+P group %u", group);
    timeit(nm, [&](uint4* e) { hipLaunchKernelGGL(k_hist<3>, dim3(nseg), dim3(kT), 0, 0, e, e + nL, nL, n, g, te, nblk, group, P, T, meta); });
    snprintf(nm, sizeof nm, "hist atomics only group %u", group);
    timeit(nm, [&](uint4* e) { hipLaunchKernelGGL(k_hist<1>, dim3(nseg), dim3(kT), 0, 0, e, e + nL, nL, n, g, te, nblk, group, P, T, meta); });
    snprintf(nm, sizeof nm, "hist P rows only group %u", group);
    timeit(nm, [&](uint4* e) { hipLaunchKernelGGL(k_hist<2>, dim3(nseg), dim3(kT), 0, 0, e, e + nL, nL, n, g, te, nblk, group, P, T, meta); });
    snprintf(nm, sizeof nm, "hist loads only (no LDS, no barriers) group %u", group);
    timeit(nm, [&](uint4* e) { hipLaunchKernelGGL(k_hist<0>, dim3(nseg), dim3(kT), 0, 0, e, e + nL, nL, n, g, te, nblk, group, P, T, meta); });
  }
  timeit("read 256 thr x 4 rec, 2048 blocks", [&](uint4* e) { hipLaunchKernelGGL((k_read<4, 256>), dim3(2048), dim3(256), 0, 0, e, n, meta); });
  timeit("read 256 thr x 4 rec, 4096 blocks", [&](uint4* e) { hipLaunchKernelGGL((k_read<4, 256>), dim3(4096), dim3(256), 0, 0, e, n, meta); });
  timeit("read 256 thr x 8 rec, 2048 blocks", [&](uint4* e) { hipLaunchKernelGGL((k_read<8, 256>), dim3(2048), dim3(256), 0, 0, e, n, meta); });
  timeit("read 256 thr x 4 rec, one pass (6544 blocks)", [&](uint4* e) { hipLaunchKernelGGL((k_read<4, 256>), dim3((n + 1023) / 1024), dim3(256), 0, 0, e, n, meta); });
  timeit("read 1024 thr x 4 rec, 512 blocks", [&](uint4* e) { hipLaunchKernelGGL((k_read<4, 1024>), dim3(512), dim3(1024), 0, 0, e, n, meta); });
  timeit("read 1024 thr x 4 rec, one pass (1636 blocks)", [&](uint4* e) { hipLaunchKernelGGL((k_read<4, 1024>), dim3((n + 4095) / 4096), dim3(1024), 0, 0, e, n, meta); });
  timeit("read 512 thr x 4 rec, 1024 blocks", [&](uint4* e) { hipLaunchKernelGGL((k_read<4, 512>), dim3(1024), dim3(512), 0, 0, e, n, meta); });
  timeit("hipMemsetAsync of P (11.8 MB written)", [&](uint4*) { hipMemsetAsync(P, 0, (size_t)nblk * g.nbins * 4, 0); });
  return 0;
}
