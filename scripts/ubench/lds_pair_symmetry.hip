// Micro-benchmark behind the go / no-go on pair symmetry in the tiled forces kernel: what the LDS pipe of a CU
// does with the traffic of one pair slot as the kernel issues it now -- three random 16-byte row reads (position, velocity,
// EOS row of the neighbour) -- and with the traffic a symmetric evaluation adds: the reaction on the neighbour has to be added
// to an accumulator that another lane owns, four ds_add_f32 (ax, ay, az, drho) to a random one of the tile's <= 640 home rows.
// One workgroup of 512 threads per CU, as the kernel runs; with and without the vector arithmetic of a pair next to it.
// Build: hipcc --offload-arch=gfx950 -O3 lds_pair_symmetry.hip -o lds_pair_symmetry
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
#define ITERS 4096
#define ROWS 2876          // window records (TILE_WCAP)
#define HOME 640           // home particles of a tile (TILE_PMAX)

// READS: 3 ds_read_b128 per slot;  ADDS: 4 ds_add_f32 per slot;  VALU: plain + packed vector instructions per slot
template<int READS, int ADDS, int PLAIN, int PACKED>
__global__ void __launch_bounds__(512) bench(float *out, float x, float y)
{
	__shared__ float4 sPos[ROWS], sVel[ROWS], sAux[ROWS];
	__shared__ float4 sAcc[HOME];
	for (int i = threadIdx.x; i < ROWS; i += blockDim.x) { sPos[i] = make_float4(i, x, y, 1.0f); sVel[i] = sPos[i]; sAux[i] = sPos[i]; }
	for (int i = threadIdx.x; i < HOME; i += blockDim.x) sAcc[i] = make_float4(0, 0, 0, 0);
	__syncthreads();
	float a[12]; v2f b[6];
	for (int i = 0; i < 12; ++i) a[i] = x + i;
	for (int i = 0; i < 6; ++i) b[i] = v2f{x + i, y + i};
	const v2f xy = {x, y};
	// eight random rows / home rows per lane in registers, moved on by a fixed stride every eight slots: two vector
	// instructions of address arithmetic per slot and array, as the kernel has (entry -> address of the third array)
	unsigned seed = threadIdx.x*2654435761u + blockIdx.x*40503u + 12345u;
	unsigned rk[8], hk[8];
	for (int k = 0; k < 8; ++k) {
		seed = seed*1664525u + 1013904223u; rk[k] = ((seed >> 8) % ROWS)*16u;
		seed = seed*1664525u + 1013904223u; hk[k] = ((seed >> 8) % HOME)*16u;
	}
	float4 acc = {0, 0, 0, 0};
	const unsigned velOff = (unsigned)((size_t)sVel - (size_t)sPos), auxOff = (unsigned)((size_t)sAux - (size_t)sPos);
	const unsigned posBase = (unsigned)(size_t)sPos, accBase = (unsigned)(size_t)sAcc;
	for (int it = 0; it < ITERS; it += 8) {
#pragma unroll
		for (int k = 0; k < 8; ++k) {
			unsigned row = rk[k] + 592u; row = row >= ROWS*16u ? row - ROWS*16u : row; rk[k] = row;
			unsigned home = hk[k] + 176u; home = home >= HOME*16u ? home - HOME*16u : home; hk[k] = home;
			float4 p = {0, 0, 0, 0}, v = p, e = p;
			if (READS) {
				const unsigned ad = posBase + row;
				asm volatile("ds_read_b128 %0, %1" : "=v"(p) : "v"(ad));
				asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(ad + velOff));
				asm volatile("ds_read_b128 %0, %1" : "=v"(e) : "v"(ad + auxOff));
			}
#pragma unroll
			for (int q = 0; q < PLAIN; ++q) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[q % 12]) : "v"(x), "v"(y));
#pragma unroll
			for (int q = 0; q < PACKED; ++q) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(b[q % 6]) : "v"(xy), "v"(xy));
			if (READS) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); acc.x += p.x + v.y + e.z; }
			if (ADDS) {
				const unsigned ad = accBase + home;
				asm volatile("ds_add_f32 %0, %1" :: "v"(ad), "v"(a[0]) : "memory");
				asm volatile("ds_add_f32 %0, %1 offset:4" :: "v"(ad), "v"(a[1]) : "memory");
				asm volatile("ds_add_f32 %0, %1 offset:8" :: "v"(ad), "v"(a[2]) : "memory");
				asm volatile("ds_add_f32 %0, %1 offset:12" :: "v"(ad), "v"(a[3]) : "memory");
			}
		}
	}
	asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
	__syncthreads();
	float s = acc.x + sAcc[threadIdx.x % HOME].x;
	for (int i = 0; i < 12; ++i) s += a[i];
	for (int i = 0; i < 6; ++i) s += b[i].x + b[i].y;
	out[blockIdx.x*blockDim.x + threadIdx.x] = s;
}

template<int READS, int ADDS, int PLAIN, int PACKED> double run(const char *name)
{
	const int blocks = 256, threads = 512;
	float *out; hipMalloc(&out, sizeof(float)*blocks*threads);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	bench<READS, ADDS, PLAIN, PACKED><<<blocks, threads>>>(out, 1.0001f, 0.9999f);
	hipEventRecord(e0);
	bench<READS, ADDS, PLAIN, PACKED><<<blocks, threads>>>(out, 1.0001f, 0.9999f);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	// one workgroup per CU, 8 waves: the CU sees 8 * ITERS wave-slots
	const double ns = ms*1e6/(8.0*ITERS);
	printf("%-64s %8.3f ms   %6.2f ns per pair slot of a wave, per CU\n", name, ms, ns);
	hipFree(out);
	return ns;
}

int main()
{
	printf("one 512-thread workgroup per CU, %d pair slots per lane; rows at random in a window of %d records, accumulators at random among %d home rows\n", ITERS, ROWS, HOME);
	const double r = run<1, 0, 0, 0>("3 x ds_read_b128 (the loop now, LDS side alone)");
	const double ra = run<1, 1, 0, 0>("3 x ds_read_b128 + 4 x ds_add_f32 (a symmetric pair, LDS side alone)");
	const double a = run<0, 1, 0, 0>("4 x ds_add_f32 alone");
	const double v = run<0, 0, 24, 6>("24 plain + 6 packed vector instructions alone (36 issue slots: one pair slot)");
	const double rv = run<1, 0, 24, 6>("3 reads + the arithmetic of one pair slot (the loop now)");
	const double rav = run<1, 1, 24, 6>("3 reads + 4 adds + the arithmetic of one pair slot");
	const double rav2 = run<1, 1, 30, 8>("3 reads + 4 adds + 1.25 x the arithmetic (reaction terms of the other side)");
	printf("\nper STORED pair with a fraction f = 0.63 of the pairs inside one tile (each evaluated once for both ends):\n");
	printf("  now                      : %6.2f ns\n", rv);
	printf("  symmetric, same arithmetic: %6.2f ns  (0.685 evaluations per stored pair: 0.37 one-sided at %0.2f + 0.315 two-sided at %0.2f)\n",
		0.37*rv + 0.315*rav, rv, rav);
	printf("  symmetric, 1.25 x arithm. : %6.2f ns\n", 0.37*rv + 0.315*rav2);
	(void)r; (void)ra; (void)a; (void)v;
	return 0;
}
