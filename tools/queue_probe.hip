// Which HIP streams of one process end up on the same hardware queue?  A kernel that spins for ~200 us goes to
// stream a, an empty kernel to stream b; if b's kernel only finishes when a's has, the two streams share a queue
// (every dispatch of a queue carries the barrier bit).   hipcc --offload-arch=gfx950 -O2 tools/queue_probe.hip -o /tmp/queue_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void k_spin(unsigned long long ticks) {  // wall_clock64: 100 MHz
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
__global__ void k_nop() {}
int main(int argc, char** argv) {
  const int sets = argc > 1 ? atoi(argv[1]) : 2;  // how many "handles" (sets of five streams) to create
  int lo = 0, hi = 0;
  hipDeviceGetStreamPriorityRange(&lo, &hi);
  printf("priority range: least %d greatest %d\n", lo, hi);
  struct S { hipStream_t s; const char* name; int set; };
  std::vector<S> st;
  const char* names[7] = {"main(hi)", "prefetch(lo)", "spec(hi)", "stereo(lo)", "stereo2(lo)", "stager(norm)", "xchg(norm)"};
  const int prio[7] = {1, -1, 1, -1, -1, 0, 0};
  for (int h = 0; h < sets; h++)
    for (int i = 0; i < 7; i++) {
      hipStream_t s;
      if (prio[i] == 0) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
      else hipStreamCreateWithPriority(&s, hipStreamNonBlocking, prio[i] > 0 ? hi : lo);
      hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, s);  // (first use: the queue is created now)
      hipStreamSynchronize(s);
      st.push_back({s, names[i], h});
    }
  const int n = (int)st.size();
  for (int a = 0; a < n; a++) {
    printf("set %d %-13s shares a queue with:", st[a].set, st[a].name);
    for (int b = 0; b < n; b++) {
      if (a == b) continue;
      hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st[a].s, 20000ull);  // 200 us
      const auto t0 = std::chrono::steady_clock::now();
      hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, st[b].s);
      hipStreamSynchronize(st[b].s);
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      hipStreamSynchronize(st[a].s);
      if (us > 120.0) printf("  [set %d %s]", st[b].set, st[b].name);
    }
    printf("\n");
  }
  return 0;
}
