// clock probe: single-wave dependent chains, shader clock (s_memtime) vs wall clock
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void chain_fma(float* out, int n, long long* cyc) {
  float x = out[0];
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; i++) { x = fmaf(x, 1.0000001f, 0.5f); x = fmaf(x, 0.9999999f, -0.5f); x = fmaf(x, 1.0000001f, 0.5f); x = fmaf(x, 0.9999999f, -0.5f); }
  long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = x; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void chain_readlane(int* out, int n, long long* cyc) {
  int x = out[threadIdx.x];
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; i++) { int s = __builtin_amdgcn_readlane(x, 5); s = s * 3 + 1; x = x + s; }
  long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = x; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  float* d; long long* c; int* di;
  hipMalloc(&d, 4096); hipMalloc(&c, 8); hipMalloc(&di, 4096); hipMemset(d, 0, 4096); hipMemset(di, 0, 4096);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 3; rep++) {
    int n = 100000;
    hipEventRecord(a); hipLaunchKernelGGL(chain_fma, dim3(1), dim3(64), 0, 0, d, n, c); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); long long cy; hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
    printf("fma chain: %d x4 dependent fma: %.3f ms, memtime ticks %lld -> %.2f ns/fma, ticks/fma %.2f\n", n, ms, cy, ms * 1e6 / (4.0 * n), cy / (4.0 * n));
    hipEventRecord(a); hipLaunchKernelGGL(chain_readlane, dim3(1), dim3(64), 0, 0, di, n, c); hipEventRecord(b); hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b); hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
    printf("readlane->salu->valu chain: %.3f ms -> %.2f ns/iter\n", ms, ms * 1e6 / n);
  }
  return 0;
}
