// Stand-in for <hip/hip_runtime.h> — TEST INFRASTRUCTURE of this repository's own host code, nothing of the
// reference's.  tests/test_host_tsan.py builds the library's host side (fe_api / fe_track / fe_stages / fe_image /
// fe_evstage / fe_host .cpp) with g++ -fsanitize=thread against this header, hip_stub.cpp (a HIP runtime that runs
// everything at once on the calling thread: "device" memory is host memory) and fake_device.cpp (kernel launchers
// that do nothing, or just enough for the host logic to have work), so that ThreadSanitizer sees every thread the
// library starts — the launch thread, the staging helpers, the RANSAC pool — on a box without a GPU.
// Streams and events carry the happens-before edges HIP gives them as atomics, so that what HIP orders is ordered
// for the sanitizer as well, and what only luck orders is reported.
#pragma once
#include <stddef.h>
#include <stdint.h>

#define __host__
#define __device__
#define __global__
#ifndef __forceinline__
#define __forceinline__ inline __attribute__((always_inline))
#endif

typedef enum hipError_t { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotReady = 600 } hipError_t;
typedef struct ihipStream_t* hipStream_t;
typedef struct ihipEvent_t* hipEvent_t;
typedef struct ihipGraph* hipGraph_t;
typedef struct hipGraphExec* hipGraphExec_t;
typedef struct hipGraphNode* hipGraphNode_t;

struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct double2 { double x, y; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
struct int2 { int x, y; };
static inline float2 make_float2(float x, float y) { float2 r = {x, y}; return r; }
static inline float4 make_float4(float x, float y, float z, float w) { float4 r = {x, y, z, w}; return r; }
static inline double2 make_double2(double x, double y) { double2 r = {x, y}; return r; }
static inline uint2 make_uint2(unsigned x, unsigned y) { uint2 r = {x, y}; return r; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { uint4 r = {x, y, z, w}; return r; }
struct dim3 {
  unsigned x, y, z;
  dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};

enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum { hipHostMallocDefault = 0, hipHostRegisterDefault = 0, hipEventDisableTiming = 2, hipStreamNonBlocking = 1 };
enum hipMemoryType { hipMemoryTypeUnregistered = 0, hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2 };
struct hipPointerAttribute_t { hipMemoryType type; int device; void* devicePointer; void* hostPointer; };
struct hipKernelNodeParams { void* func; dim3 gridDim, blockDim; unsigned sharedMemBytes; void** kernelParams; void** extra; };

extern "C++" {
hipError_t hipGetDeviceCount(int* n);
hipError_t hipSetDevice(int d);
hipError_t hipGetDevice(int* d);
hipError_t hipGetLastError(void);
const char* hipGetErrorString(hipError_t e);
hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest);
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 63 };
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t a, int dev);
hipError_t hipMemGetInfo(size_t* free_b, size_t* total_b);
hipError_t hipMalloc(void** p, size_t bytes);
template <typename T> static inline hipError_t hipMalloc(T** p, size_t bytes) { return hipMalloc((void**)p, bytes); }
hipError_t hipFree(void* p);
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned flags);
template <typename T> static inline hipError_t hipHostMalloc(T** p, size_t bytes, unsigned flags) { return hipHostMalloc((void**)p, bytes, flags); }
hipError_t hipHostFree(void* p);
hipError_t hipHostGetDevicePointer(void** dev, void* host, unsigned flags);
template <typename T> static inline hipError_t hipHostGetDevicePointer(T** dev, void* host, unsigned flags) { return hipHostGetDevicePointer((void**)dev, host, flags); }
hipError_t hipHostRegister(void* p, size_t bytes, unsigned flags);
hipError_t hipHostUnregister(void* p);
hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void* p);
hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind k, hipStream_t s);
hipError_t hipMemcpy2DAsync(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, hipMemcpyKind k, hipStream_t s);
hipError_t hipMemsetAsync(void* dst, int v, size_t bytes, hipStream_t s);
hipError_t hipStreamCreate(hipStream_t* s);
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags);
hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned flags, int prio);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamQuery(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventQuery(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
hipError_t hipGraphLaunch(hipGraphExec_t g, hipStream_t s);
hipError_t hipGraphExecKernelNodeSetParams(hipGraphExec_t g, hipGraphNode_t n, const hipKernelNodeParams* p);
// the stub's own: a stream-ordered operation begins / ends on stream s (the fake kernels bracket their work with
// these): a stream runs its operations one after the other, whichever threads enqueued them
void hipstub_stream_begin(hipStream_t s);
void hipstub_stream_end(hipStream_t s);
static inline void hipstub_stream_op(hipStream_t s) { hipstub_stream_begin(s); hipstub_stream_end(s); }
}
