// TEST HARNESS ONLY (tests/hostemu): a stand-in for <hip/hip_runtime.h> with which g++ compiles the SOURCE of some of this
// repository's own HIP kernels for the host, so that their logic can be held against the oracle on the CPU, where there is no GPU:
// element-wise kernels one "thread" after the other (SPHX_LAUNCH), wave-cooperative kernels (SPHX_LAUNCH_WAVES: ballots, shuffles,
// readlane, LDS rows shared by the lanes of a wave) as 64 fibres per wave that meet at every wave operation, so that a lane sees
// the values the other lanes hold at that instruction, as on the device.  It exists to find logic errors (indices, flags, signs,
// operation order, lanes that miss a wave operation) early; it proves nothing about the device build, is never loaded by the
// package, and is no CPU fallback of the library (libsphx.so has none).  Only what sphx_api.hip, sa_io.hip and sa_bounds.hip use is here.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <climits>
#include <functional>
#include <vector>
#include <ucontext.h>

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
struct dim3 { unsigned x = 1, y = 1, z = 1; dim3() {} dim3(unsigned a, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
	explicit operator unsigned() const { return x; } };      // one-dimensional launches only
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
static inline void __threadfence() {}

// ---- waves of fibres ---------------------------------------------------------------------------------------------------------------
// A block of a SPHX_LAUNCH_WAVES launch is a team of fibres (ucontext), one per thread, run round-robin by a scheduler on the
// calling OS thread.  A wave operation is a rendezvous of the 64 fibres of a wave: each deposits its operand, waits (yields) until
// all 64 have, and then reads what it needs from the deposits; deposits are double-buffered by the parity of the rendezvous, so one
// rendezvous per operation is enough.  A fibre that returns while its wave still waits for it, or a wave operation under a
// lane-dependent condition, shows up as a round of the scheduler in which nothing moves: reported and aborted.  __syncthreads is
// the same over the whole block.  Outside such a launch (serial SPHX_LAUNCH) a thread is a wave of its own lane.
namespace emu {
struct Fibre { ucontext_t ctx; bool done; };
struct Wave { int arrived; unsigned gen; unsigned long long slot[2][64]; };
struct Team {
	std::vector<Fibre> fibre;
	std::vector<Wave> wave;
	ucontext_t scheduler;
	unsigned current;
	unsigned long moved;
	int barArrived; unsigned barGen;
	const std::function<void()> *body;
};
inline Team *team = nullptr;
inline int marks[2048]; inline unsigned long meets[2048], meetsAtMark[2048];      // SPHX_EMU_MARK(n): where a fibre was last seen, printed when a block gets stuck
inline std::vector<char> stacks;
static const size_t STACK = 512u*1024u;
inline void yield() { Team *t = team; swapcontext(&t->fibre[t->current].ctx, &t->scheduler); }
inline void fibre_main()
{
	Team *t = team;
	(*t->body)();
	t->fibre[t->current].done = true;
	t->moved++;
	swapcontext(&t->fibre[t->current].ctx, &t->scheduler);
}
// deposit v, meet the other lanes of the wave, hand back the deposits of this operation
inline const unsigned long long *meet(unsigned long long v)
{
	Team *t = team;
	const unsigned tid = t->current, lane = tid & 63u;
	meets[tid]++;
	Wave &w = t->wave[tid >> 6];
	const unsigned g = w.gen, par = g & 1u;
	w.slot[par][lane] = v;
	if (++w.arrived == 64) { w.arrived = 0; w.gen++; t->moved++; }
	else while (w.gen == g) yield();
	return w.slot[par];
}
inline void meet_block()
{
	Team *t = team;
	const unsigned g = t->barGen;
	if (++t->barArrived == (int)t->fibre.size()) { t->barArrived = 0; t->barGen++; t->moved++; }
	else while (t->barGen == g) yield();
}
}
template<typename T> static inline unsigned long long emu_bits(T v) { static_assert(sizeof(T) <= 8, "operand of a wave operation"); unsigned long long b = 0; memcpy(&b, &v, sizeof(T)); return b; }
template<typename T> static inline T emu_value(unsigned long long b) { T v; memcpy(&v, &b, sizeof(T)); return v; }
static inline unsigned emu_lane();
template<typename T> static inline T __shfl(T v, int src) { if (!emu::team) return v; const unsigned long long *d = emu::meet(emu_bits(v)); return emu_value<T>(d[(unsigned)src & 63u]); }
template<typename T> static inline T __shfl_xor(T v, int m) { if (!emu::team) return v; const unsigned l = emu_lane(); const unsigned long long *d = emu::meet(emu_bits(v)); return emu_value<T>(d[(l ^ (unsigned)m) & 63u]); }
template<typename T> static inline T __shfl_down(T v, unsigned k) { if (!emu::team) return v; const unsigned l = emu_lane(); const unsigned long long *d = emu::meet(emu_bits(v)); return l + k < 64u ? emu_value<T>(d[l + k]) : v; }
static inline unsigned long long __builtin_amdgcn_ballot_w64(bool b)
{
	if (!emu::team) return b ? (1ull << emu_lane()) : 0ull;
	const unsigned long long *d = emu::meet(b ? 1ull : 0ull);
	unsigned long long m = 0;
	for (unsigned l = 0; l < 64u; ++l) m |= (d[l] & 1ull) << l;
	return m;
}
static inline int __builtin_amdgcn_readlane(int v, int l) { if (!emu::team) return v; const unsigned long long *d = emu::meet(emu_bits(v)); return emu_value<int>(d[(unsigned)l & 63u]); }
// every lane's value of v at once (wave_list.h's ordered_sums reads all lanes of a term: one meeting instead of sixty-four)
static inline void emu_wave_gather(float v, float *out) { const unsigned long long *d = emu::meet(emu_bits(v)); for (unsigned l = 0; l < 64u; ++l) out[l] = emu_value<float>(d[l]); }
#define SPHX_WAVE_GATHER(v, out) emu_wave_gather((v), (out))
static inline void __builtin_amdgcn_wave_barrier() { if (emu::team) (void)emu::meet(0ull); }
static inline void __syncthreads() { if (emu::team) emu::meet_block(); }
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
static inline unsigned __builtin_amdgcn_readfirstlane(unsigned v) { if (!emu::team) return v; const unsigned long long *d = emu::meet(emu_bits(v)); return emu_value<unsigned>(d[0]); }
static inline void __builtin_amdgcn_global_load_lds(const void *src, void *dst, int bytes, int, int) { memcpy(dst, src, (size_t)bytes); }
static inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
static inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
static inline unsigned __float_as_uint(float f) { unsigned i; memcpy(&i, &f, 4); return i; }
static inline float __uint_as_float(unsigned i) { float f; memcpy(&f, &i, 4); return f; }

// ---- what neibs_build.hip uses on top of that (round 6: the list build with its distance tests on the matrix cores) ------------
#define ext_vector_type(n) vector_size(4*(n))      // four-byte elements only (float x 16, uint32 x 2 / x 4); index with [], not .x
#define __builtin_assume(x) ((void)0)
template<typename T> static inline T atomicCAS(T *p, T cmp, T v) { const T o = *p; if (o == cmp) *p = v; return o; }
typedef unsigned emu_u32x2 __attribute__((vector_size(8)));
typedef unsigned emu_u32x4 __attribute__((vector_size(16)));
typedef float emu_f32x16 __attribute__((vector_size(64)));
// v_permlane32_swap_b32 vdst, vsrc: lanes 32..63 of vdst trade places with lanes 0..31 of vsrc; returns {vdst, vsrc}
static inline emu_u32x2 __builtin_amdgcn_permlane32_swap(unsigned vdst, unsigned vsrc, bool, bool)
{
	const unsigned l = emu_lane();
	const unsigned long long *d = emu::meet(((unsigned long long)vsrc << 32) | vdst);
	emu_u32x2 r;
	r[0] = l < 32u ? vdst : (unsigned)(d[l - 32u] >> 32);          // upper half of vdst <- lower half of vsrc
	r[1] = l < 32u ? (unsigned)(d[l + 32u] & 0xFFFFFFFFu) : vsrc;  // lower half of vsrc <- upper half of vdst
	return r;
}
// v_mfma_f32_32x32x2_f32: D = A (32 x 2) B (2 x 32) + C.  A[i][k] is lane i + 32 k's `a`, B[k][j] lane j + 32 k's `b`; lane l holds
// D[8 (r / 4) + 4 (l / 32) + r % 4][l % 32] in element r (the layout of the CDNA3/4 ISA guides).  The rounding of the unit is not
// modelled (one fused multiply-add per k here): the kernel's band logic must not depend on it, and does not
static inline emu_f32x16 __builtin_amdgcn_mfma_f32_32x32x2f32(float a, float b, emu_f32x16 c, int, int, int)
{
	const unsigned l = emu_lane();
	const unsigned long long *d = emu::meet(((unsigned long long)emu_bits(b) << 32) | emu_bits(a));
	unsigned long long all[64];
	for (unsigned k = 0; k < 64u; ++k) all[k] = d[k];
	const unsigned j = l & 31u;
	for (unsigned r = 0; r < 16u; ++r) {
		const unsigned i = 8u*(r/4u) + 4u*(l/32u) + (r % 4u);
		float acc = c[r];
		for (unsigned k = 0; k < 2u; ++k) {
			const float A = emu_value<float>(all[i + 32u*k] & 0xFFFFFFFFull), B = emu_value<float>(all[j + 32u*k] >> 32);
			acc = fmaf(A, B, acc);
		}
		c[r] = acc;
	}
	return c;
}
static inline unsigned __builtin_amdgcn_alignbit(unsigned hi, unsigned lo, unsigned s)
{
	return (unsigned)(((((unsigned long long)hi) << 32) | lo) >> (s & 31u));
}
struct __amdgpu_buffer_rsrc_t { const char *base; unsigned bytes; };
static inline __amdgpu_buffer_rsrc_t __builtin_amdgcn_make_buffer_rsrc(void *p, int, int bytes, int) { return __amdgpu_buffer_rsrc_t{(const char*)p, (unsigned)bytes}; }
static inline emu_u32x4 __builtin_amdgcn_raw_buffer_load_b128(__amdgpu_buffer_rsrc_t r, int voff, int soff, int)
{
	emu_u32x4 v = {0u, 0u, 0u, 0u};
	const unsigned o = (unsigned)voff + (unsigned)soff;
	if (r.bytes == 0u || o + 16u <= r.bytes) memcpy(&v, r.base + o, 16);      // rows past the array read as zeros
	return v;
}

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

static inline unsigned emu_lane() { return threadIdx.x & 63u; }

// a launch of a wave-cooperative kernel: block after block, each as a team of fibres
namespace emu {
inline void run_grid(unsigned grid, unsigned block, const std::function<void()> &body)
{
	if (block % 64u) { fprintf(stderr, "hostemu: a wave launch needs whole waves (block of %u)\n", block); abort(); }
	if (stacks.size() < (size_t)block*STACK) stacks.resize((size_t)block*STACK);
	Team t;
	t.body = &body;
	gridDim = dim3(grid); blockDim = dim3(block);
	static const bool trace = getenv("SPHX_EMU_TRACE") != nullptr;
	for (unsigned bi = 0; bi < grid; ++bi) {
		if (trace) fprintf(stderr, "hostemu: block %u of %u\n", bi, grid);
		blockIdx = dim3(bi);
		t.fibre.assign(block, Fibre());
		t.wave.assign(block/64u, Wave());
		t.moved = 0; t.barArrived = 0; t.barGen = 0;
		for (unsigned i = 0; i < block && i < 2048u; ++i) { meets[i] = 0; marks[i] = 0; }
		team = &t;
		for (unsigned i = 0; i < block; ++i) {
			Fibre &f = t.fibre[i];
			f.done = false;
			getcontext(&f.ctx);
			f.ctx.uc_stack.ss_sp = stacks.data() + (size_t)i*STACK;
			f.ctx.uc_stack.ss_size = STACK;
			f.ctx.uc_link = nullptr;
			makecontext(&f.ctx, fibre_main, 0);
		}
		for (;;) {
			bool alive = false;
			const unsigned long before = t.moved;
			for (unsigned i = 0; i < block; ++i) {
				if (t.fibre[i].done) continue;
				alive = true;
				t.current = i;
				threadIdx = dim3(i);
				swapcontext(&t.scheduler, &t.fibre[i].ctx);
			}
			if (!alive) break;
			if (t.moved == before) {
				fprintf(stderr, "hostemu: block %u is stuck: a wave operation (or __syncthreads) was not reached by every lane\n", bi);
				for (unsigned i = 0; i < block; ++i)
					fprintf(stderr, "%s%d%s:%lu/%lu%s", (i & 63u) ? " " : "  wave marks: ", marks[i], t.fibre[i].done ? "d" : "", meetsAtMark[i], meets[i], (i & 63u) == 63u ? "\n" : "");
				abort();
			}
		}
		team = nullptr;
	}
}
}
// a state the emulation cannot carry on from (fibres have no execution mask: ballots under lane-dependent conditions): say so and stop
#define SPHX_EMU_REFUSE(cond, what) do { if (cond) { fprintf(stderr, "hostemu: %s\n", what); abort(); } } while (0)
#define SPHX_EMU_MARK(n) (emu::marks[threadIdx.x] = (n), emu::meetsAtMark[threadIdx.x] = emu::meets[threadIdx.x])
#define SPHX_LAUNCH_WAVES(kernel, grid, block, stream, ...) do { (void)(stream); \
	emu::run_grid((unsigned)(grid), (unsigned)(block), [&]() { kernel(__VA_ARGS__); }); } while (0)

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
