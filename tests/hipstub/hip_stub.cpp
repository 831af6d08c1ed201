// A HIP runtime for ThreadSanitizer runs (see hip/hip_runtime.h here): every operation completes on the calling
// thread before the call returns; "device" and pinned memory are calloc'ed host memory; streams and events are
// atomics that carry the ordering HIP guarantees (an operation on a stream happens after the stream's earlier
// operations and after every event the stream waited for; hipEventSynchronize / hipStreamSynchronize / a
// successful hipEventQuery happen after what the event or stream had recorded).
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <map>
#include <set>

struct ihipStream_t {
  std::atomic<uint64_t> seq{0};
  std::mutex mu;  // a stream runs one operation at a time
};
struct ihipEvent_t {
  std::atomic<uint64_t> seq{0};
};
namespace {
ihipStream_t g_null_stream;
ihipStream_t* S(hipStream_t s) { return s ? s : &g_null_stream; }
std::mutex g_mu;
std::map<const void*, size_t> g_pinned;  // hipHostMalloc / hipHostRegister'ed ranges: base -> bytes
std::atomic<uint64_t> g_allocs{0};
// HIPSTUB_FAIL_EVERY=n: every n-th hipEventRecord fails (once each) — a launch that fails in the middle of a call,
// on whichever thread issues it
std::atomic<long> g_records{0};
std::atomic<int> g_armed{1};
long fail_every() {
  static const long n = getenv("HIPSTUB_FAIL_EVERY") ? atol(getenv("HIPSTUB_FAIL_EVERY")) : 0;
  return n;
}
}  // namespace

extern "C" void hipstub_arm_faults(int on) { g_armed.store(on, std::memory_order_relaxed); }

void hipstub_stream_begin(hipStream_t s) {
  S(s)->mu.lock();
  S(s)->seq.fetch_add(1, std::memory_order_acq_rel);
}
void hipstub_stream_end(hipStream_t s) {
  S(s)->seq.fetch_add(1, std::memory_order_acq_rel);
  S(s)->mu.unlock();
}

hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "hipstub"; }
hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest) { *least = 0; *greatest = -1; return hipSuccess; }
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 256; return hipSuccess; }
hipError_t hipMemGetInfo(size_t* f, size_t* t) { *f = (size_t)64 << 30; *t = (size_t)288 << 30; return hipSuccess; }
hipError_t hipMalloc(void** p, size_t bytes) {
  *p = calloc(bytes ? bytes : 1, 1);
  g_allocs.fetch_add(1, std::memory_order_relaxed);
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned) {
  *p = calloc(bytes ? bytes : 1, 1);
  if (!*p) return hipErrorOutOfMemory;
  std::lock_guard<std::mutex> lk(g_mu);
  g_pinned[*p] = bytes ? bytes : 1;
  return hipSuccess;
}
hipError_t hipHostFree(void* p) {
  {
    std::lock_guard<std::mutex> lk(g_mu);
    g_pinned.erase(p);
  }
  free(p);
  return hipSuccess;
}
hipError_t hipHostGetDevicePointer(void** dev, void* host, unsigned) { *dev = host; return hipSuccess; }
hipError_t hipHostRegister(void* p, size_t bytes, unsigned) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_pinned[p] = bytes ? bytes : 1;
  return hipSuccess;
}
hipError_t hipHostUnregister(void* p) {
  std::lock_guard<std::mutex> lk(g_mu);
  return g_pinned.erase(p) ? hipSuccess : hipErrorInvalidValue;
}
hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void* p) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_pinned.upper_bound(p);  // (the range that contains p, like the real runtime)
  if (it == g_pinned.begin()) return hipErrorInvalidValue;  // (what the real runtime says of malloc'ed memory)
  --it;
  if ((const char*)p >= (const char*)it->first + it->second) return hipErrorInvalidValue;
  a->type = hipMemoryTypeHost;
  a->device = 0;
  a->devicePointer = a->hostPointer = (void*)p;
  return hipSuccess;
}
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) {
  hipstub_stream_begin(nullptr);
  if (n) memmove(d, s, n);
  hipstub_stream_end(nullptr);
  return hipSuccess;
}
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t st) {
  hipstub_stream_begin(st);
  if (n) memmove(d, s, n);
  hipstub_stream_end(st);
  return hipSuccess;
}
hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t st) {
  hipstub_stream_begin(st);
  for (size_t y = 0; y < h; y++) memmove((char*)d + y * dp, (const char*)s + y * sp, w);
  hipstub_stream_end(st);
  return hipSuccess;
}
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st) {
  hipstub_stream_begin(st);
  if (n) memset(d, v, n);
  hipstub_stream_end(st);
  return hipSuccess;
}
hipError_t hipStreamCreate(hipStream_t* s) { *s = new ihipStream_t; return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { return hipStreamCreate(s); }
hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { return hipStreamCreate(s); }
hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t s) { (void)S(s)->seq.load(std::memory_order_acquire); return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t s) { (void)S(s)->seq.load(std::memory_order_acquire); return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned) {
  (void)e->seq.load(std::memory_order_acquire);
  hipstub_stream_op(s);
  return hipSuccess;
}
hipError_t hipEventCreate(hipEvent_t* e) { *e = new ihipEvent_t; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
  if (fail_every() > 0 && g_armed.load(std::memory_order_relaxed) && (g_records.fetch_add(1, std::memory_order_relaxed) + 1) % fail_every() == 0) return hipErrorInvalidValue;
  hipstub_stream_op(s);
  e->seq.fetch_add(1, std::memory_order_acq_rel);
  return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t e) { (void)e->seq.load(std::memory_order_acquire); return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t e) { (void)e->seq.load(std::memory_order_acquire); return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.001f; return hipSuccess; }
hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipSuccess; }
hipError_t hipGraphExecKernelNodeSetParams(hipGraphExec_t, hipGraphNode_t, const hipKernelNodeParams*) { return hipSuccess; }
