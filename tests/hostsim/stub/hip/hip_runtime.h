// TEST INFRASTRUCTURE ONLY (tests/hostsim): a host-memory stand-in for the few HIP runtime calls libgpk's host
// orchestration makes, so that gpk_core / gpk_api / gpk_pref / gpk_class / gpk_grow / gpk_comm can be compiled with
// g++ -fsanitize=address,undefined and run WITHOUT a GPU (host-side sanitizer build).  "Device"
// memory is malloc'ed host memory, every operation is synchronous, streams and events carry no ordering.  The product
// (gptest_amd/libgpk.so) is never built from this header and never loads anything from tests/hostsim.
#pragma once
#include <chrono>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>

typedef enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 } hipError_t;
typedef enum { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 } hipMemcpyKind;
struct hostsim_stream { int id; };
struct hostsim_event { double t_ms; };
typedef hostsim_stream* hipStream_t;
typedef hostsim_event* hipEvent_t;
struct hipDeviceProp_t { char gcnArchName[256]; int multiProcessorCount; };
struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };
#define hipStreamNonBlocking 1
#define hipEventDisableTiming 2
#define __global__
#define __launch_bounds__(...)
#define __restrict__

extern thread_local dim3 blockIdx, threadIdx, blockDim, gridDim;
// serial execution of a barrier-free kernel: every (block, thread) in turn
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...)                               \
  do {                                                                                            \
    const dim3 g_ = (grid), b_ = (block);                                                         \
    gridDim = g_; blockDim = b_;                                                                  \
    for (unsigned bz = 0; bz < g_.z; ++bz) for (unsigned by = 0; by < g_.y; ++by) for (unsigned bx = 0; bx < g_.x; ++bx) \
      for (unsigned tz = 0; tz < b_.z; ++tz) for (unsigned ty = 0; ty < b_.y; ++ty) for (unsigned tx = 0; tx < b_.x; ++tx) { \
        blockIdx = dim3(bx, by, bz); threadIdx = dim3(tx, ty, tz);                                \
        kernel(__VA_ARGS__);                                                                      \
      }                                                                                           \
  } while (0)

static inline double hostsim_now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "success" : "hostsim error"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
  std::strcpy(p->gcnArchName, "gfx950:hostsim");
  p->multiProcessorCount = 256;
  return hipSuccess;
}
static inline hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = -1; return hipSuccess; }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = new hostsim_stream{0}; return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = new hostsim_stream{0}; return hipSuccess; }
static inline hipError_t hipExtStreamCreateWithCUMask(hipStream_t* s, uint32_t, const uint32_t*) { *s = new hostsim_stream{1}; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t e, unsigned) { return e ? hipSuccess : hipErrorInvalidValue; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new hostsim_event{0.0}; return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t_ms = hostsim_now_ms(); return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t_ms - a->t_ms); return hipSuccess; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
static inline hipError_t hipHostMalloc(void** p, size_t n) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { std::memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { std::memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemcpy2DAsync(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height,
                                          hipMemcpyKind, hipStream_t) {
  for (size_t r = 0; r < height; ++r) std::memmove((char*)d + r * dpitch, (const char*)s + r * spitch, width);
  return hipSuccess;
}
