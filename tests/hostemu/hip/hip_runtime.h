// TEST HARNESS ONLY (tests/hostemu): a stand-in for <hip/hip_runtime.h> with which g++ compiles the SOURCE of some of this
// repository's own HIP kernels for the host, so that kernels that have not yet run on a GPU can be held against the oracle on the
// CPU: one "thread" after the other, blockIdx / threadIdx as globals.  It exists to find logic errors (indices, flags, signs,
// operation order) early; it proves nothing about the device build, is never loaded by the package, and is no CPU fallback of
// the library (libsphx.so has none).  Only what sphx_api.hip and sa_io.hip use is here.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <climits>

#define __HIPCC__ 1          // the repository's headers keep their device helpers behind it
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static
#define __restrict__ __restrict

struct alignas(8)  float2 { float x, y; };
struct             float3 { float x, y, z; };
struct alignas(16) float4 { float x, y, z, w; };
struct             int3   { int x, y, z; };
struct alignas(16) int4   { int x, y, z, w; };
struct             uint3  { unsigned x, y, z; };
struct alignas(16) uint4  { unsigned x, y, z, w; };
struct alignas(8)  uint2  { unsigned x, y; };
struct alignas(8)  ushort4 { unsigned short x, y, z, w; };
struct alignas(8)  short4  { short x, y, z, w; };
struct alignas(16) double2 { double x, y; };
struct dim3 { unsigned x = 1, y = 1, z = 1; dim3() {} dim3(unsigned a, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float3 make_float3(float x, float y, float z) { return float3{x, y, z}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline int3 make_int3(int x, int y, int z) { return int3{x, y, z}; }
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
static inline uint3 make_uint3(unsigned x, unsigned y, unsigned z) { return uint3{x, y, z}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
static inline ushort4 make_ushort4(unsigned short x, unsigned short y, unsigned short z, unsigned short w) { return ushort4{x, y, z, w}; }

// the "thread" that is running
extern thread_local dim3 blockIdx, threadIdx, blockDim, gridDim;

// serial execution: an atomic is the plain operation, a shuffle hands back the lane's own value, a barrier is nothing.  Block-wide
// reductions (the CFL maxima of the forces kernel) are therefore NOT emulated and must not be compared.
template<typename T> static inline T atomicAdd(T *p, T v) { const T o = *p; *p = o + v; return o; }
template<typename T> static inline T atomicMax(T *p, T v) { const T o = *p; if (v > o) *p = v; return o; }
template<typename T> static inline T atomicMin(T *p, T v) { const T o = *p; if (v < o) *p = v; return o; }
template<typename T> static inline T atomicOr(T *p, T v) { const T o = *p; *p = o | v; return o; }
template<typename T> static inline T atomicExch(T *p, T v) { const T o = *p; *p = v; return o; }
template<typename T> static inline T __shfl_down(T v, unsigned) { return v; }
template<typename T> static inline T __shfl_xor(T v, unsigned) { return v; }
template<typename T> static inline T __shfl(T v, int) { return v; }
static inline void __syncthreads() {}
static inline void __threadfence() {}
#define __powf(a, b) powf((a), (b))
#define __expf(a) expf(a)
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
static inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
static inline float min(float a, float b) { return fminf(a, b); }
static inline float max(float a, float b) { return fmaxf(a, b); }
static inline float __fdividef(float a, float b) { return a/b; }
static inline float rsqrtf(float a) { return 1.0f/sqrtf(a); }
// (sphx_internal.h's LDS-DMA helpers are not used by the files built here; they only have to parse)
static inline unsigned __builtin_amdgcn_readfirstlane(unsigned v) { return v; }
static inline void __builtin_amdgcn_global_load_lds(const void *src, void *dst, int bytes, int, int) { memcpy(dst, src, (size_t)bytes); }
static inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
static inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
static inline unsigned __float_as_uint(float f) { unsigned i; memcpy(&i, &f, 4); return i; }
static inline float __uint_as_float(unsigned i) { float f; memcpy(&f, &i, 4); return f; }

// ---- the runtime API the host side of the two files calls: memory is host memory, streams and events are nothing -------------
typedef int hipError_t;
typedef void *hipStream_t;
typedef void *hipEvent_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorNotReady = 600 };
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum hipStreamCaptureStatus { hipStreamCaptureStatusNone, hipStreamCaptureStatusActive };
enum { hipHostMallocDefault = 0, hipEventDisableTiming = 2, hipStreamNonBlocking = 1 };
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount };
static inline const char *hipGetErrorString(hipError_t) { return "host emulation"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipMalloc(void **p, size_t n) { *p = calloc(1, n ? n : 1); return *p ? hipSuccess : hipErrorInvalidValue; }
template<typename T> static inline hipError_t hipMalloc(T **p, size_t n) { return hipMalloc((void**)p, n); }
static inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
template<typename T> static inline hipError_t hipHostMalloc(T **p, size_t n, unsigned f = 0) { return hipMalloc((void**)p, n); }
static inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipMemset(void *p, int v, size_t n) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t = nullptr) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int *c) { *c = 1; return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t, int) { *v = 256; return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t *s) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned = 0) { return hipSuccess; }
static inline hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus *s) { *s = hipStreamCaptureStatusNone; return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t = nullptr) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }

// a launch: every block, every thread, in order
#define SPHX_LAUNCH(kernel, grid, block, stream, ...) do { \
	const unsigned g_ = (unsigned)(grid), b_ = (unsigned)(block); (void)(stream); \
	gridDim = dim3(g_); blockDim = dim3(b_); \
	for (unsigned bi_ = 0; bi_ < g_; ++bi_) for (unsigned ti_ = 0; ti_ < b_; ++ti_) { \
		blockIdx = dim3(bi_); threadIdx = dim3(ti_); kernel(__VA_ARGS__); } } while (0)

// the same for sources whose launches are rewritten by tests/hostemu_lib.py instead of going through a macro of their own
// (sa_bounds.hip: verified on the GPU and left as it is; kernel<<<grid, block, 0, stream>>>(args) -> SPHX_EMU_LAUNCH((kernel), grid, block, args))
#define SPHX_EMU_LAUNCH(kernel, grid, block, ...) do { \
	const unsigned g_ = (unsigned)(grid), b_ = (unsigned)(block); \
	gridDim = dim3(g_); blockDim = dim3(b_); \
	for (unsigned bi_ = 0; bi_ < g_; ++bi_) for (unsigned ti_ = 0; ti_ < b_; ++ti_) { \
		blockIdx = dim3(bi_); threadIdx = dim3(ti_); kernel(__VA_ARGS__); } } while (0)
