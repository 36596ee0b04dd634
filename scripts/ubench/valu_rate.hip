// Micro-benchmark: issue rate of the vector instructions the pair loop is made of, per SIMD, in shader cycles
// (s_memtime), at 1 and 2 waves per SIMD.  Settles whether a wave64 fp32 VALU instruction costs 2 or 4 cycles on gfx950
// and what v_pk_*_f32, v_sqrt/v_rcp and ds_read_b128 cost next to it.  Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
#define ITERS 2048
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template<int MODE>
__global__ void __launch_bounds__(1024) bench(float *out, unsigned long long *cyc, float x, float y)
{
	__shared__ float4 lds[2048];
	float a[16]; v2f b[16];
	for (int i = 0; i < 16; ++i) { a[i] = x + i; b[i] = v2f{x + i, y + i}; }
	for (int i = threadIdx.x; i < 2048; i += blockDim.x) lds[i] = make_float4(i, x, y, 1.0f);
	__syncthreads();
	v2f xy = {x, y};
	unsigned addr = ((threadIdx.x*1103515245u + 12345u) >> 8) % 2048u * 16u;
	float4 l0 = {0,0,0,0};
	const unsigned long long t0 = __builtin_readcyclecounter();
	for (int it = 0; it < ITERS; ++it) {
		if (MODE == 0) {
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
			REP16(X)
#undef X
		} else if (MODE == 1) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(b[i]) : "v"(xy), "v"(xy));
			REP16(X)
#undef X
		} else if (MODE == 2) {
#define X(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
			REP16(X)
#undef X
		} else if (MODE == 3) {   // the pair loop's mix: 12 plain, 12 packed, 2 transcendental... here 8 fma + 6 pk + 2 sqrt
#define X(i) if (i < 8) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y)); \
	else if (i < 14) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(b[i]) : "v"(xy), "v"(xy)); \
	else asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
			REP16(X)
#undef X
		} else if (MODE == 4) {   // dependent chain of plain fma (latency)
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(x), "v"(y));
			REP16(X)
#undef X
		} else if (MODE == 5) {   // random 16-byte LDS reads, 16 in flight
#define X(i) { float4 t; asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"(addr + (unsigned)(i*2064u) % 32768u)); l0.x += t.x; }
			REP16(X)
#undef X
			asm volatile("s_waitcnt lgkmcnt(0)");
		} else if (MODE == 6) {   // v_pk_mul_f32
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(b[i]) : "v"(xy));
			REP16(X)
#undef X
		} else if (MODE == 7) {   // v_add_f32 (VOP2)
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
			REP16(X)
#undef X
		}
	}
	const unsigned long long t1 = __builtin_readcyclecounter();
	float s = l0.x;
	for (int i = 0; i < 16; ++i) s += a[i] + b[i].x + b[i].y;
	out[blockIdx.x*blockDim.x + threadIdx.x] = s;
	if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template<int MODE> void run(const char *name, int threads)
{
	const int blocks = 256;
	float *out; unsigned long long *cyc;
	hipMalloc(&out, sizeof(float)*blocks*1024); hipMalloc(&cyc, sizeof(unsigned long long)*blocks);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	bench<MODE><<<blocks, threads>>>(out, cyc, 1.0001f, 0.9999f);
	hipEventRecord(e0);
	bench<MODE><<<blocks, threads>>>(out, cyc, 1.0001f, 0.9999f);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	std::vector<unsigned long long> h(blocks);
	hipMemcpy(h.data(), cyc, sizeof(unsigned long long)*blocks, hipMemcpyDeviceToHost);
	double mean = 0; for (auto v : h) mean += (double)v; mean /= blocks;
	const double instrPerWave = (double)ITERS*16.0, wavesPerSimd = threads/256.0;
	// s_memtime / readcyclecounter ticks at a fixed 100 MHz on gfx9: convert with the wall time instead
	printf("%-28s %4d thr/WG  %8.3f ms  ticks %10.0f  => %6.2f ns per instr per wave, %6.3f ns per instr per SIMD\n",
		name, threads, ms, mean, ms*1e6/instrPerWave, ms*1e6/(instrPerWave*wavesPerSimd));
	hipFree(out); hipFree(cyc);
}

int main()
{
	int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
	printf("clock rate attribute: %d kHz; at 2.4 GHz one cycle = 0.4167 ns\n", clk);
	for (int threads : {256, 512, 768, 1024}) {
		run<0>("v_fma_f32 x16 indep", threads);
		run<7>("v_add_f32 x16 indep", threads);
		run<1>("v_pk_fma_f32 x16 indep", threads);
		run<6>("v_pk_mul_f32 x16 indep", threads);
		run<2>("v_sqrt_f32 x16 indep", threads);
		run<3>("mix 8 fma + 6 pk + 2 rcp", threads);
		run<4>("v_fma_f32 dependent chain", threads);
		run<5>("ds_read_b128 random x16", threads);
	}
	return 0;
}
