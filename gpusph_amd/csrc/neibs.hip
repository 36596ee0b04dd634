// neibs.hip -- neighbour engine for gfx950: cell hash, cell-binning sort, reorder + cell
// start/end, cell-linked neighbour list.  Replaces CUDANeibsEngine (GPUSPH
// src/cuda/buildneibs.cu:68-496, kernels src/cuda/buildneibs_kernel.cu:659-1185); results are
// bit-identical to the reference algorithm restated in oracle/sph_oracle.c.
//
// Numerics (DESIGN.md "Numerics"): compiled with -ffp-contract=off; `a - b*c` shapes are
// explicit fmaf, squared lengths are fmaf(z,z,fmaf(y,y,x*x)), division is IEEE.
#include "sphx_internal.h"
#include <cstring>

#define BLOCK_HASH    256
#define BLOCK_SORT    256
#define BLOCK_REORDER 256
#define BLOCK_NEIBS   256
#define SCAN_ITEMS    1024   // elements per scan block (256 threads x 4)

// ------------------------------------------------------------------------------------------
// calcHash: src/cuda/buildneibs_kernel.cu:659-776 (clampGridPos :225-298)
// 52 B/particle streaming: R pos16 info8 hash4 (+4 devmap), W pos16 hash4 partIndex4
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int clamp_axis(int gp, int &off, int gs, bool periodic, bool &toofar)
{
	int ng = gp + off;
	if (periodic) {
		if (ng < 0) ng += gs;
		if (ng >= gs) ng -= gs;
	} else {
		ng = min(max(0, ng), gs - 1);
		if (abs(off) > 1 && ng == gp)
			toofar = true;
		off = ng - gp;
	}
	return ng;
}

__global__ void __launch_bounds__(BLOCK_HASH)
calc_hash_kernel(DevParams p, float4 *__restrict__ posArray, uint32_t *__restrict__ particleHash,
	uint32_t *__restrict__ particleIndex, const particleinfo *__restrict__ particleInfo,
	const uint32_t *__restrict__ compactDeviceMap, uint32_t numParticles)
{
	const uint32_t index = blockIdx.x*BLOCK_HASH + threadIdx.x;
	if (index >= numParticles) return;

	const particleinfo info = particleInfo[index];
	uint32_t gridHash = particleHash[index] & CELLTYPE_BITMASK;

	if (IS_FLUID(info) || IS_MOVING(info) || (IS_SURFACE(info) && !IS_FLUID(info))) {
		float4 pos = posArray[index];
		const int3 gridPos = grid_pos_from_hash(p, gridHash);

		// floor(pos/cellSize + (pos<0 ? 0.5 : 0.49999997)), see the long comment at :696-725
		int ox = (int)floorf(pos.x/p.cs[0] + (pos.x < 0 ? 0.5f : 0.49999997f));
		int oy = (int)floorf(pos.y/p.cs[1] + (pos.y < 0 ? 0.5f : 0.49999997f));
		int oz = (int)floorf(pos.z/p.cs[2] + (pos.z < 0 ? 0.5f : 0.49999997f));

		bool toofar = false;
		const int nx = clamp_axis(gridPos.x, ox, p.gs[0], p.periodic & SPHX_PERIODIC_X, toofar);
		const int ny = clamp_axis(gridPos.y, oy, p.gs[1], p.periodic & SPHX_PERIODIC_Y, toofar);
		const int nz = clamp_axis(gridPos.z, oz, p.gs[2], p.periodic & SPHX_PERIODIC_Z, toofar);
		gridHash = grid_hash(p, nx, ny, nz);

		pos.x = fmaf(-(float)ox, p.cs[0], pos.x);
		pos.y = fmaf(-(float)oy, p.cs[1], pos.y);
		pos.z = fmaf(-(float)oz, p.cs[2], pos.z);

		if (toofar)
			pos.w = __uint_as_float(0x7fc00000u); // disable_particle: mass = NaN

		if (!is_active_w(pos.w))
			gridHash = CELL_HASH_MAX;

		posArray[index] = pos;
	}

	if (compactDeviceMap && gridHash != CELL_HASH_MAX)
		gridHash |= compactDeviceMap[gridHash];

	particleHash[index] = gridHash;
	particleIndex[index] = index;
}

// fixHash: src/cuda/buildneibs_kernel.cu:786-814
__global__ void __launch_bounds__(BLOCK_HASH)
fix_hash_kernel(uint32_t *__restrict__ particleHash, uint32_t *__restrict__ particleIndex,
	const uint32_t *__restrict__ compactDeviceMap, uint32_t numParticles)
{
	const uint32_t index = blockIdx.x*BLOCK_HASH + threadIdx.x;
	if (index >= numParticles) return;
	if (particleHash) {
		const uint32_t h = particleHash[index];
		if (compactDeviceMap)
			particleHash[index] = h | compactDeviceMap[h & CELLTYPE_BITMASK];
	}
	particleIndex[index] = index;
}

// ------------------------------------------------------------------------------------------
// sort: the reference calls thrust::sort_by_key with the comparator ptype_hash_compare
// (src/cuda/buildneibs.cu:358-412): total order (hash incl. cell-type bits, PART_TYPE, id).
// Because the order is total, any correct sort yields the same permutation.  Here: a
// cell-binning sort -- histogram of the hash with wave-aggregated atomics, exclusive scan of
// the bins, scatter, then an in-bin rank by (type, id).  ~60 B/particle instead of the
// ~200 B/particle of an 8-pass 64-bit LSD radix sort, and no comparison sort at all.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t hash_to_bin(uint32_t h, uint32_t cells, uint32_t lastBin)
{
	return (h == CELL_HASH_MAX) ? lastBin : (h >> 30)*cells + (h & CELLTYPE_BITMASK);
}

// The in-bin rank below compares every particle of a bin with every other one: fine for cells (~18 particles), quadratic
// for the one bin all INACTIVE particles share (hash CELL_HASH_MAX): tens of thousands of them after an outflow or after
// disableFreeSurfParts would cost 1e9-1e10 comparisons on a single bin.  When that bin holds more than SORT_BIG_BIN
// particles its members instead keep their relative order (slot = number of inactive particles before them, from a scan
// of flags) and are not ranked.  The reference orders them by (type, id) like everything else; nothing reads that order:
// the caller drops the inactive tail (newNumParticles, src/cuda/buildneibs_kernel.cu:905-909, GPUWorker.cc:1471-1515).
// All of it is decided on the device from the bin's count (the kernels return at once for a small bin): no host sync.
#define SORT_BIG_BIN 4096u

__global__ void __launch_bounds__(BLOCK_SORT)
inactive_flags_kernel(const uint32_t *__restrict__ hash, uint32_t *__restrict__ flags, uint32_t n, const uint32_t *__restrict__ guard)
{
	if (*guard <= SORT_BIG_BIN) return;
	const uint32_t i = blockIdx.x*BLOCK_SORT + threadIdx.x;
	if (i < n) flags[i] = hash[i] == CELL_HASH_MAX ? 1u : 0u;
}

__global__ void __launch_bounds__(BLOCK_SORT)
inactive_slots_kernel(const uint32_t *__restrict__ hash, const uint32_t *__restrict__ before, uint32_t *__restrict__ slot,
	uint32_t n, const uint32_t *__restrict__ guard)
{
	if (*guard <= SORT_BIG_BIN) return;
	const uint32_t i = blockIdx.x*BLOCK_SORT + threadIdx.x;
	if (i < n && hash[i] == CELL_HASH_MAX) slot[i] = before[i];
}

__global__ void __launch_bounds__(BLOCK_SORT)
sort_count_kernel(const uint32_t *__restrict__ hash, uint32_t *__restrict__ binCount,
	uint32_t *__restrict__ slot, uint32_t cells, uint32_t lastBin, uint32_t n)
{
	const uint32_t i = blockIdx.x*BLOCK_SORT + threadIdx.x;
	const uint32_t lane = threadIdx.x & 63u;
	const bool valid = i < n;
	const uint32_t bin = valid ? hash_to_bin(hash[i], cells, lastBin) : 0xFFFFFFFFu;

	// particles arrive almost sorted from the previous rebuild: consecutive lanes mostly share
	// a bin.  One atomic per run of equal bins instead of one per particle.
	const uint32_t prev = __shfl_up(bin, 1);
	const bool head = (lane == 0) || (bin != prev);
	const unsigned long long heads = __ballot(head);
	const unsigned long long upto = heads & ((lane == 63u) ? ~0ull : ((2ull << lane) - 1ull));
	const int head_lane = 63 - __clzll((long long)upto);
	const unsigned long long above = (head_lane == 63) ? 0ull : (heads & ~((2ull << head_lane) - 1ull));
	const int next_head = above ? (__ffsll((long long)above) - 1) : 64;
	uint32_t base = 0;
	if (valid && (int)lane == head_lane)
		base = atomicAdd(&binCount[bin], (uint32_t)(next_head - head_lane));
	base = __shfl(base, head_lane);
	if (valid)
		slot[i] = base + (lane - (uint32_t)head_lane);
}

// exclusive scan of uint32 in three launches (reduce / scan partials / downsweep)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, uint32_t lane)
{
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		const uint32_t t = __shfl_up(v, d);
		if ((int)lane >= d) v += t;
	}
	return v;
}

__device__ __forceinline__ uint32_t block_excl_scan_256(uint32_t v, uint32_t &blockTotal)
{
	__shared__ uint32_t waveSums[4];
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	const uint32_t incl = wave_incl_scan(v, lane);
	if (lane == 63u) waveSums[wave] = incl;
	__syncthreads();
	uint32_t offset = 0;
	for (uint32_t w = 0; w < wave; ++w) offset += waveSums[w];
	blockTotal = waveSums[0] + waveSums[1] + waveSums[2] + waveSums[3];
	__syncthreads();
	return offset + incl - v;
}

__global__ void __launch_bounds__(256)
scan_reduce_kernel(const uint32_t *__restrict__ in, uint32_t *__restrict__ partials, uint32_t n,
	const uint32_t *__restrict__ guard = nullptr)
{
	if (guard && *guard <= SORT_BIG_BIN) return;
	const uint32_t base = blockIdx.x*SCAN_ITEMS + threadIdx.x*4;
	uint32_t s = 0;
	if (base + 3 < n) {
		const uint4 v = *reinterpret_cast<const uint4*>(in + base);
		s = v.x + v.y + v.z + v.w;
	} else {
		for (uint32_t k = 0; k < 4; ++k) if (base + k < n) s += in[base + k];
	}
	uint32_t total;
	(void)block_excl_scan_256(s, total);
	if (threadIdx.x == 0) partials[blockIdx.x] = total;
}

__global__ void __launch_bounds__(256)
scan_partials_kernel(uint32_t *__restrict__ partials, uint32_t numPartials, const uint32_t *__restrict__ guard = nullptr)
{
	if (guard && *guard <= SORT_BIG_BIN) return;
	uint32_t carry = 0;
	for (uint32_t base = 0; base < numPartials; base += 256) {
		const uint32_t i = base + threadIdx.x;
		const uint32_t v = i < numPartials ? partials[i] : 0;
		uint32_t total;
		const uint32_t ex = block_excl_scan_256(v, total);
		if (i < numPartials) partials[i] = carry + ex;
		carry += total;
	}
}

__global__ void __launch_bounds__(256)
scan_final_kernel(const uint32_t *__restrict__ in, uint32_t *__restrict__ out,
	const uint32_t *__restrict__ partials, uint32_t n, const uint32_t *__restrict__ guard = nullptr)
{
	if (guard && *guard <= SORT_BIG_BIN) return;
	const uint32_t base = blockIdx.x*SCAN_ITEMS + threadIdx.x*4;
	uint32_t v[4] = {0, 0, 0, 0};
	if (base + 3 < n) {
		const uint4 t = *reinterpret_cast<const uint4*>(in + base);
		v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
	} else {
		for (uint32_t k = 0; k < 4; ++k) if (base + k < n) v[k] = in[base + k];
	}
	const uint32_t s = v[0] + v[1] + v[2] + v[3];
	uint32_t total;
	uint32_t ex = block_excl_scan_256(s, total) + partials[blockIdx.x];
	uint32_t o[4];
	o[0] = ex; o[1] = o[0] + v[0]; o[2] = o[1] + v[1]; o[3] = o[2] + v[2];
	if (base + 3 < n) {
		*reinterpret_cast<uint4*>(out + base) = make_uint4(o[0], o[1], o[2], o[3]);
	} else {
		for (uint32_t k = 0; k < 4; ++k) if (base + k < n) out[base + k] = o[k];
	}
}

__global__ void __launch_bounds__(BLOCK_SORT)
sort_scatter_kernel(const uint32_t *__restrict__ hash, const uint2 *__restrict__ info,
	const uint32_t *__restrict__ partIndex, const uint32_t *__restrict__ slot,
	const uint32_t *__restrict__ binStart,
	uint32_t *__restrict__ tmpHash, uint2 *__restrict__ tmpInfo, uint32_t *__restrict__ tmpIndex,
	uint32_t cells, uint32_t lastBin, uint32_t n)
{
	const uint32_t i = blockIdx.x*BLOCK_SORT + threadIdx.x;
	if (i >= n) return;
	const uint32_t h = hash[i];
	const uint32_t p = binStart[hash_to_bin(h, cells, lastBin)] + slot[i];
	tmpHash[p] = h;
	tmpInfo[p] = info[i];
	tmpIndex[p] = partIndex[i];
}

// (PART_TYPE, id) as one 64-bit key; particleinfo viewed as uint2: .x = type|flags + fluid/object<<16,
// .y = id (z | w<<16)
__device__ __forceinline__ unsigned long long type_id_key(uint2 info)
{
	return ((unsigned long long)(info.x & 7u) << 32) | (unsigned long long)info.y;
}

__global__ void __launch_bounds__(BLOCK_SORT)
sort_rank_kernel(const uint32_t *__restrict__ tmpHash, const uint2 *__restrict__ tmpInfo,
	const uint32_t *__restrict__ tmpIndex,
	const uint32_t *__restrict__ binStart, const uint32_t *__restrict__ binCount,
	uint32_t *__restrict__ hash, uint2 *__restrict__ info, uint32_t *__restrict__ partIndex,
	uint32_t cells, uint32_t lastBin, uint32_t n)
{
	const uint32_t p = blockIdx.x*BLOCK_SORT + threadIdx.x;
	if (p >= n) return;
	const uint32_t h = tmpHash[p];
	const uint32_t bin = hash_to_bin(h, cells, lastBin);
	const uint32_t s = binStart[bin];
	const uint32_t e = s + binCount[bin];
	const uint2 myInfo = tmpInfo[p];
	const unsigned long long key = type_id_key(myInfo);
	uint32_t rank = 0;
	if (bin == lastBin && e - s > SORT_BIG_BIN) {
		rank = p - s;          // a big inactive bin arrives in its final (stable) order, see SORT_BIG_BIN
	} else {
		for (uint32_t q = s; q < e; ++q) {
			const unsigned long long k = type_id_key(tmpInfo[q]);
			rank += (k < key) || (k == key && q < p);
		}
	}
	const uint32_t dst = s + rank;
	hash[dst] = h;
	info[dst] = myInfo;
	partIndex[dst] = tmpIndex[p];
}

// ------------------------------------------------------------------------------------------
// reorderDataAndFindCellStart: src/cuda/buildneibs_kernel.cu:836-992.  The neighbouring hash
// comes from the L1-resident previous element instead of a shared-memory stage.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(BLOCK_REORDER)
reorder_kernel(uint32_t *__restrict__ cellStart, uint32_t *__restrict__ cellEnd,
	uint32_t *__restrict__ segmentStart,
	float4 *__restrict__ sortedPos, float4 *__restrict__ sortedVel,
	const float4 *__restrict__ unsortedPos, const float4 *__restrict__ unsortedVel,
	const uint32_t *__restrict__ particleHash, const uint32_t *__restrict__ particleIndex,
	uint32_t numParticles, uint32_t *__restrict__ newNumParticles)
{
	const uint32_t index = blockIdx.x*BLOCK_REORDER + threadIdx.x;
	if (index >= numParticles) return;

	const uint32_t cellHash = particleHash[index];
	const uint32_t prevHash = index > 0 ? particleHash[index - 1] : 0u;

	if (index == 0 || cellHash != prevHash) {
		if (cellHash != CELL_HASH_MAX)
			cellStart[cellHash & CELLTYPE_BITMASK] = index;
		else
			*newNumParticles = index;
		if (index > 0)
			cellEnd[prevHash & CELLTYPE_BITMASK] = index;
	}

	if (cellHash == CELL_HASH_MAX)
		return;

	if (index == numParticles - 1) {
		cellEnd[cellHash & CELLTYPE_BITMASK] = index + 1;
		*newNumParticles = numParticles;
	}

	if (segmentStart) {
		const uint32_t curr_type = cellHash >> 30;
		const uint32_t prev_type = prevHash >> 30;
		if (index == 0 || curr_type != prev_type)
			segmentStart[curr_type] = index;
	}

	const uint32_t sortedIndex = particleIndex[index];
	sortedPos[index] = unsortedPos[sortedIndex];
	sortedVel[index] = unsortedVel[sortedIndex];
}

// cell ranges of a sorted sub-range (imported halo), same scan as above without the gather
__global__ void __launch_bounds__(BLOCK_REORDER)
find_cell_start_kernel(uint32_t *__restrict__ cellStart, uint32_t *__restrict__ cellEnd,
	const uint32_t *__restrict__ particleHash, uint32_t from, uint32_t to)
{
	const uint32_t index = from + blockIdx.x*BLOCK_REORDER + threadIdx.x;
	if (index >= to) return;
	const uint32_t cellHash = particleHash[index];
	if (cellHash == CELL_HASH_MAX) return;
	if (index == from || cellHash != particleHash[index - 1]) {
		cellStart[cellHash & CELLTYPE_BITMASK] = index;
		if (index > from) {
			const uint32_t prev = particleHash[index - 1];
			if (prev != CELL_HASH_MAX) cellEnd[prev & CELLTYPE_BITMASK] = index;
		}
	}
	if (index == to - 1)
		cellEnd[cellHash & CELLTYPE_BITMASK] = index + 1;
}

// first non-fluid particle of every cell (cells are sorted fluid-first, ptype_hash_compare): lets the
// list build know a candidate's type from its index alone in the fluid segment of a cell
__global__ void __launch_bounds__(256)
cell_fluid_end_kernel(uint32_t *__restrict__ cellFluidEnd, const uint32_t *__restrict__ particleHash,
	const particleinfo *__restrict__ infoArray, uint32_t numParticles)
{
	const uint32_t i = blockIdx.x*256 + threadIdx.x;
	if (i >= numParticles) return;
	const uint32_t h = particleHash[i];
	if (h == CELL_HASH_MAX) return;
	const bool fluid = IS_FLUID(infoArray[i]);
	const bool firstOfCell = (i == 0) || (particleHash[i - 1] != h);
	const bool lastOfCell = (i == numParticles - 1) || (particleHash[i + 1] != h);
	if (!fluid && (firstOfCell || IS_FLUID(infoArray[i - 1])))
		cellFluidEnd[h & CELLTYPE_BITMASK] = i;              // first non-fluid particle of the cell
	else if (fluid && lastOfCell)
		cellFluidEnd[h & CELLTYPE_BITMASK] = i + 1;          // all fluid: the segment ends with the cell
}

// ------------------------------------------------------------------------------------------
// forces tiles (consumed by forces_tile_kernel, forces.hip).  A tile is a block of k x 2 x 2 cells
// (k consecutive cells along COORD1 in each of 4 adjacent grid rows) holding <= TILE_PMAX
// particles; its neighbour window is the (k+2) x 4 x 4 block around it (16 contiguous particle
// ranges) and must fit TILE_WCAP records.  One thread per 2x2 bundle of rows walks the cells along
// COORD1 and closes a tile greedily.
// ------------------------------------------------------------------------------------------
// pre-pass of the tile builder, one thread per (row bundle, column): records in the 16 window rows of that column
// and whether any of those cells holds fluid (cells are sorted fluid-first, so the first particle tells).  The
// greedy walk below is serial per bundle (a few thousand threads); summing 16 cells per step inside it made it
// latency bound (2.5 ms per build at 32 M particles).
__global__ void __launch_bounds__(256)
tile_columns_kernel(DevParams p, const uint32_t *__restrict__ cellStart, const uint32_t *__restrict__ cellEnd,
	const particleinfo *__restrict__ info, uint32_t *__restrict__ cols)
{
	const int gs1 = p.gs1;
	const int gs2 = (p.c2 == 0) ? p.gs[0] : (p.c2 == 1) ? p.gs[1] : p.gs[2];
	const int gs3 = (p.c3 == 0) ? p.gs[0] : (p.c3 == 1) ? p.gs[1] : p.gs[2];
	const int nG2 = (gs2 + 1)/2, nG3 = (gs3 + 1)/2;
	const uint32_t t = blockIdx.x*256 + threadIdx.x;
	if (t >= (uint32_t)(nG2*nG3*gs1)) return;
	const int c = (int)(t % (uint32_t)gs1), sr = (int)(t / (uint32_t)gs1);
	const int g2 = 2*(sr % nG2), g3 = 2*(sr / nG2);
	const bool per2 = p.periodic & (1u << p.c2), per3 = p.periodic & (1u << p.c3);
	uint32_t sum = 0, fluid = 0;
	for (int d3 = -1; d3 <= 2; ++d3) for (int d2 = -1; d2 <= 2; ++d2) {
		int c2v = g2 + d2, c3v = g3 + d3;
		if (c2v < 0) { if (per2) c2v = gs2 - 1; else continue; } else if (c2v >= gs2) { if (per2) c2v = 0; else continue; }
		if (c3v < 0) { if (per3) c3v = gs3 - 1; else continue; } else if (c3v >= gs3) { if (per3) c3v = 0; else continue; }
		const uint32_t h = (uint32_t)(c + c2v*gs1 + c3v*p.gs12);
		const uint32_t cs = cellStart[h];
		if (cs == CELL_EMPTY) continue;
		sum += cellEnd[h] - cs;
		if (IS_FLUID(info[cs])) fluid = 0x80000000u;
	}
	cols[t] = sum | fluid;
}

__global__ void __launch_bounds__(128)
build_tiles_kernel(DevParams p, const uint32_t *__restrict__ cellStart, const uint32_t *__restrict__ cellEnd,
	const uint32_t *__restrict__ cols, uint32_t rangeEnd,
	uint32_t *__restrict__ tiles, uint32_t *__restrict__ ctl, uint32_t capacity)
{
	// the SPS window also holds tau (and, with several fluids, the EOS rows as well)
	const uint32_t wcap = (p.turbmodel == SPHX_SPS) ? (p.numfluids > 1 ? TILE_WCAP_SPS : TILE_WCAP_SPS1) : TILE_WCAP;
	const int gs1 = p.gs1;
	const int gs2 = (p.c2 == 0) ? p.gs[0] : (p.c2 == 1) ? p.gs[1] : p.gs[2];
	const int gs3 = (p.c3 == 0) ? p.gs[0] : (p.c3 == 1) ? p.gs[1] : p.gs[2];
	const int nG2 = (gs2 + 1)/2, nG3 = (gs3 + 1)/2;
	// thread -> row bundle in blocks of 8 x 4 bundles (COORD2 x COORD3): the threads of a wave emit their tiles in step, so
	// consecutive tile numbers are neighbouring bundles at the same column range, and the forces kernel hands 32 consecutive
	// tiles to one XCD: with 2-D blocks the windows of those 32 tiles overlap along COORD2 AND COORD3 and re-use each other's
	// rows in that XCD's L2 (a 1-D run of 32 bundles shares rows along COORD2 only)
	const int t = (int)(blockIdx.x*128 + threadIdx.x);
	const int nB2 = (nG2 + 7)/8, nB3 = (nG3 + 3)/4;
	if (t >= nB2*nB3*32) return;
	const int blk = t >> 5, within = t & 31;
	const int b2 = (blk % nB2)*8 + (within & 7), b3 = (blk / nB2)*4 + (within >> 3);
	if (b2 >= nG2 || b3 >= nG3) return;
	const int sr = b2 + nG2*b3;
	const int g2 = 2*b2, g3 = 2*b3;
	const bool per1 = p.periodic & (1u << p.c1), per2 = p.periodic & (1u << p.c2), per3 = p.periodic & (1u << p.c3);

	auto cell_cnt = [&](int c1v, int c2v, int c3v, bool wrap, uint32_t &start) -> uint32_t {
		if (c1v < 0) { if (per1 && wrap) c1v = gs1 - 1; else return 0u; } else if (c1v >= gs1) { if (per1 && wrap) c1v = 0; else return 0u; }
		if (c2v < 0) { if (per2 && wrap) c2v = gs2 - 1; else return 0u; } else if (c2v >= gs2) { if (per2 && wrap) c2v = 0; else return 0u; }
		if (c3v < 0) { if (per3 && wrap) c3v = gs3 - 1; else return 0u; } else if (c3v >= gs3) { if (per3 && wrap) c3v = 0; else return 0u; }
		const uint32_t h = (uint32_t)(c1v + c2v*gs1 + c3v*p.gs12);
		const uint32_t cs = cellStart[h];
		if (cs == CELL_EMPTY) return 0u;
		start = cs;
		return cellEnd[h] - cs;
	};
	// window column c (tile_columns_kernel); columns -1 and gs1 wrap with periodicity along COORD1
	const uint32_t *myCols = cols + (size_t)sr*gs1;
	auto column = [&](int c, uint32_t &hasFluid) -> uint32_t {
		hasFluid = 0;
		if (c < 0) { if (per1) c = gs1 - 1; else return 0u; } else if (c >= gs1) { if (per1) c = 0; else return 0u; }
		const uint32_t v = myCols[c];
		hasFluid = v >> 31;
		return v & 0x7FFFFFFFu;
	};
	uint32_t hc[TILE_HROWS] = {0, 0, 0, 0}, first[TILE_HROWS] = {0, 0, 0, 0};
	auto emit = [&](int ca, int ncells, uint32_t wc, uint32_t fl) {
		const uint32_t idx = atomicAdd(&ctl[0], 1u);
		if (idx >= capacity) { ctl[1] = 1u; return; }
		uint32_t *d = tiles + (size_t)TILE_DESC*idx;
		d[0] = (uint32_t)g2; d[1] = (uint32_t)g3; d[2] = (uint32_t)ca; d[3] = (uint32_t)ncells;
		for (int r = 0; r < TILE_HROWS; ++r) { d[4 + r] = first[r]; d[8 + r] = hc[r]; }
		d[12] = wc; d[13] = fl; d[14] = 0u; d[15] = 0u;
	};

	uint32_t hsum = 0, wc = 0, wfl = 0;
	int ca = 0;
	uint32_t f_prev, f_cur, f_next;
	uint32_t cs_prev = column(-1, f_prev), cs_cur = column(0, f_cur);
	for (int c = 0; c < gs1; ++c) {
		const uint32_t cs_next = column(c + 1, f_next);
		uint32_t n4[TILE_HROWS], st4[TILE_HROWS];
		uint32_t ncol = 0;
		bool contiguous = true;
		for (int r = 0; r < TILE_HROWS; ++r) {
			st4[r] = 0;
			n4[r] = cell_cnt(c, g2 + (r & 1), g3 + (r >> 1), false, st4[r]);
			if (n4[r] && st4[r] >= rangeEnd) n4[r] = 0;     // halo cells hold no particle with a neighbour list
			ncol += n4[r];
			if (n4[r] && hc[r] && st4[r] != first[r] + hc[r]) contiguous = false;
		}
		const bool fits = hsum && ncol && (hsum + ncol <= TILE_PMAX) && (wc + cs_next <= wcap) &&
			(c - ca + 1 <= TILE_MAXCELLS) && contiguous;
		if (fits) {
			for (int r = 0; r < TILE_HROWS; ++r) { if (n4[r] && !hc[r]) first[r] = st4[r]; hc[r] += n4[r]; }
			hsum += ncol; wc += cs_next; wfl |= f_next;
		} else {
			if (hsum) emit(ca, c - ca, wc, wfl);
			hsum = 0;
			for (int r = 0; r < TILE_HROWS; ++r) hc[r] = 0;
			if (ncol) {
				ca = c; hsum = ncol;
				for (int r = 0; r < TILE_HROWS; ++r) { hc[r] = n4[r]; first[r] = st4[r]; }
				wc = cs_prev + cs_cur + cs_next; wfl = f_prev | f_cur | f_next;
				if (ncol > TILE_PMAX || wc > wcap) ctl[1] = 1u;   // would not fit: generic kernel
			}
		}
		cs_prev = cs_cur; cs_cur = cs_next; f_prev = f_cur; f_cur = f_next;
	}
	if (hsum) emit(ca, gs1 - ca, wc, wfl);
}


// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
extern "C" int sphx_calc_hash(sphx_ctx *ctx, void *pos, uint32_t *hash, uint32_t *partIndex,
	const void *info, const uint32_t *compactDeviceMap, uint32_t n, void *stream)
{
	SPHX_REQUIRE(ctx && ctx->have_params, "sphx_calc_hash: constants not set");
	SPHX_REQUIRE(pos && hash && partIndex && info, "sphx_calc_hash: missing buffer (POS, HASH, PARTINDEX, INFO)");
	if (!n) return SPHX_OK;
	calc_hash_kernel<<<div_up_u(n, BLOCK_HASH), BLOCK_HASH, 0, (hipStream_t)stream>>>(ctx->dev,
		(float4*)pos, hash, partIndex, (const particleinfo*)info, compactDeviceMap, n);
	SPHX_LAUNCH_CHECK("calc_hash_kernel");
	return SPHX_OK;
}

extern "C" int sphx_fix_hash(sphx_ctx *ctx, uint32_t *hash, uint32_t *partIndex,
	const void *info, const uint32_t *compactDeviceMap, uint32_t n, void *stream)
{
	(void)info;
	SPHX_REQUIRE(ctx && ctx->have_params, "sphx_fix_hash: constants not set");
	SPHX_REQUIRE(partIndex, "sphx_fix_hash: missing PARTINDEX buffer");
	if (!n) return SPHX_OK;
	fix_hash_kernel<<<div_up_u(n, BLOCK_HASH), BLOCK_HASH, 0, (hipStream_t)stream>>>(hash, partIndex, compactDeviceMap, n);
	SPHX_LAUNCH_CHECK("fix_hash_kernel");
	return SPHX_OK;
}

extern "C" int sphx_sort(sphx_ctx *ctx, uint32_t *hash, void *info, uint32_t *partIndex,
	uint32_t n, void *stream_)
{
	SPHX_REQUIRE(ctx && ctx->have_params, "sphx_sort: constants not set");
	SPHX_REQUIRE(hash && info && partIndex, "sphx_sort: missing buffer (HASH, INFO, PARTINDEX)");
	if (!n) return SPHX_OK;
	hipStream_t stream = (hipStream_t)stream_;
	int rc = sphx_ensure_scratch(ctx, n);
	if (rc != SPHX_OK) return rc;
	const uint32_t cells = ctx->params.gridSize[0]*ctx->params.gridSize[1]*ctx->params.gridSize[2];
	const uint32_t bins = 4*cells + 1;
	const uint32_t lastBin = bins - 1;
	const uint32_t nb = div_up_u(n, BLOCK_SORT);
	const uint32_t scanBlocks = div_up_u(bins, SCAN_ITEMS);

	SPHX_HIP(hipMemsetAsync(ctx->bin_count, 0, sizeof(uint32_t)*(size_t)bins, stream));
	sort_count_kernel<<<nb, BLOCK_SORT, 0, stream>>>(hash, ctx->bin_count, ctx->slot, cells, lastBin, n);
	SPHX_LAUNCH_CHECK("sort_count_kernel");
	scan_reduce_kernel<<<scanBlocks, 256, 0, stream>>>(ctx->bin_count, ctx->scan_partials, bins);
	scan_partials_kernel<<<1, 256, 0, stream>>>(ctx->scan_partials, scanBlocks);
	scan_final_kernel<<<scanBlocks, 256, 0, stream>>>(ctx->bin_count, ctx->bin_start, ctx->scan_partials, bins);
	SPHX_LAUNCH_CHECK("scan kernels");
	{	// a big bin of inactive particles: stable slots from a scan of flags (tmp_hash / tmp_index are free until the scatter)
		const uint32_t *guard = ctx->bin_count + lastBin;
		const uint32_t nScan = div_up_u(n, SCAN_ITEMS);
		inactive_flags_kernel<<<nb, BLOCK_SORT, 0, stream>>>(hash, ctx->tmp_hash, n, guard);
		scan_reduce_kernel<<<nScan, 256, 0, stream>>>(ctx->tmp_hash, ctx->scan_partials, n, guard);
		scan_partials_kernel<<<1, 256, 0, stream>>>(ctx->scan_partials, nScan, guard);
		scan_final_kernel<<<nScan, 256, 0, stream>>>(ctx->tmp_hash, ctx->tmp_index, ctx->scan_partials, n, guard);
		inactive_slots_kernel<<<nb, BLOCK_SORT, 0, stream>>>(hash, ctx->tmp_index, ctx->slot, n, guard);
		SPHX_LAUNCH_CHECK("inactive slot kernels");
	}
	sort_scatter_kernel<<<nb, BLOCK_SORT, 0, stream>>>(hash, (const uint2*)info, partIndex, ctx->slot,
		ctx->bin_start, ctx->tmp_hash, ctx->tmp_info, ctx->tmp_index, cells, lastBin, n);
	SPHX_LAUNCH_CHECK("sort_scatter_kernel");
	sort_rank_kernel<<<nb, BLOCK_SORT, 0, stream>>>(ctx->tmp_hash, ctx->tmp_info, ctx->tmp_index,
		ctx->bin_start, ctx->bin_count, hash, (uint2*)info, partIndex, cells, lastBin, n);
	SPHX_LAUNCH_CHECK("sort_rank_kernel");
	return SPHX_OK;
}

extern "C" int sphx_reorder(sphx_ctx *ctx, uint32_t *segmentStart,
	uint32_t *cellStart, uint32_t *cellEnd,
	void *sortedPos, void *sortedVel, const void *unsortedPos, const void *unsortedVel,
	const void *sortedInfo, const uint32_t *sortedHash, const uint32_t *partIndex,
	uint32_t n, uint32_t *newNumParticles, void *stream_)
{
	(void)sortedInfo;
	SPHX_REQUIRE(ctx && ctx->have_params, "sphx_reorder: constants not set");
	SPHX_REQUIRE(cellStart && cellEnd && sortedPos && sortedVel && unsortedPos && unsortedVel &&
		sortedHash && partIndex && newNumParticles, "sphx_reorder: missing buffer");
	SPHX_REQUIRE(sortedPos != unsortedPos && sortedVel != unsortedVel, "sphx_reorder: sorted and unsorted buffers alias");
	hipStream_t stream = (hipStream_t)stream_;
	if (segmentStart)
		SPHX_HIP(hipMemsetAsync(segmentStart, 0xFF, 4*sizeof(uint32_t), stream)); // EMPTY_SEGMENT
	if (!n) return SPHX_OK;
	reorder_kernel<<<div_up_u(n, BLOCK_REORDER), BLOCK_REORDER, 0, stream>>>(cellStart, cellEnd, segmentStart,
		(float4*)sortedPos, (float4*)sortedVel, (const float4*)unsortedPos, (const float4*)unsortedVel,
		sortedHash, partIndex, n, newNumParticles);
	SPHX_LAUNCH_CHECK("reorder_kernel");
	return SPHX_OK;
}

// Reorder of the optional per-particle arrays the reference gathers in the same kernel when they are present in the buffer
// lists (src/cuda/buildneibs.cu:263-311: VOLUME, INTERNAL_ENERGY, BOUNDELEMENTS, GRADGAMMA, VERTICES, TKE, EPSILON, TURBVISC,
// EFFPRES, EULERVEL): sorted[i] = unsorted[partIndex[i]] for rows of 4, 8 or 16 bytes
template<typename T>
__global__ void __launch_bounds__(BLOCK_REORDER)
gather_rows_kernel(T * __restrict__ sorted, const T * __restrict__ unsorted, const uint32_t * __restrict__ partIndex, uint32_t n)
{
	const uint32_t i = blockIdx.x*BLOCK_REORDER + threadIdx.x;
	if (i < n) sorted[i] = unsorted[partIndex[i]];
}

extern "C" int sphx_gather_rows(sphx_ctx *ctx, void *sorted, const void *unsorted, uint32_t rowBytes,
	const uint32_t *partIndex, uint32_t n, void *stream_)
{
	SPHX_REQUIRE(ctx != nullptr, "sphx_gather_rows: NULL ctx");
	SPHX_REQUIRE(sorted && unsorted && partIndex, "sphx_gather_rows: missing buffer");
	SPHX_REQUIRE(sorted != unsorted, "sphx_gather_rows: sorted and unsorted buffers alias");
	SPHX_REQUIRE(rowBytes == 4 || rowBytes == 8 || rowBytes == 16, "sphx_gather_rows: rows of 4, 8 or 16 bytes");
	if (!n) return SPHX_OK;
	hipStream_t stream = (hipStream_t)stream_;
	const dim3 grid(div_up_u(n, BLOCK_REORDER));
	if (rowBytes == 4) gather_rows_kernel<uint32_t><<<grid, BLOCK_REORDER, 0, stream>>>((uint32_t*)sorted, (const uint32_t*)unsorted, partIndex, n);
	else if (rowBytes == 8) gather_rows_kernel<uint2><<<grid, BLOCK_REORDER, 0, stream>>>((uint2*)sorted, (const uint2*)unsorted, partIndex, n);
	else gather_rows_kernel<uint4><<<grid, BLOCK_REORDER, 0, stream>>>((uint4*)sorted, (const uint4*)unsorted, partIndex, n);
	SPHX_LAUNCH_CHECK("gather_rows_kernel");
	return SPHX_OK;
}

extern "C" int sphx_find_cell_start(sphx_ctx *ctx, uint32_t *cellStart, uint32_t *cellEnd, const uint32_t *sortedHash,
	uint32_t from, uint32_t to, void *stream)
{
	SPHX_REQUIRE(ctx && ctx->have_params, "sphx_find_cell_start: constants not set");
	SPHX_REQUIRE(cellStart && cellEnd && sortedHash, "sphx_find_cell_start: missing buffer");
	SPHX_REQUIRE(from <= to, "sphx_find_cell_start: invalid range");
	if (from == to) return SPHX_OK;
	find_cell_start_kernel<<<div_up_u(to - from, BLOCK_REORDER), BLOCK_REORDER, 0, (hipStream_t)stream>>>(cellStart, cellEnd,
		sortedHash, from, to);
	SPHX_LAUNCH_CHECK("find_cell_start_kernel");
	return SPHX_OK;
}

extern "C" int sphx_build_neibs(sphx_ctx *ctx, uint16_t *neibsList,
	const void *pos, const void *info, const uint32_t *hash,
	const uint32_t *cellStart, const uint32_t *cellEnd,
	uint32_t numParticles, uint32_t particleRangeEnd, uint32_t gridCells,
	float sqinfluenceradius, float boundNlSqInflRad, void *stream)
{
	SPHX_REQUIRE(ctx && ctx->have_params, "sphx_build_neibs: constants not set");
	SPHX_REQUIRE(ctx->params.boundarytype != SPHX_SA_BOUNDARY, "sphx_build_neibs: SA_BOUNDARY lists need sphx_build_neibs_sa");
	return sphx_build_neibs_sa(ctx, neibsList, nullptr, nullptr, nullptr, pos, info, nullptr, nullptr, hash, cellStart, cellEnd,
		numParticles, particleRangeEnd, gridCells, sqinfluenceradius, boundNlSqInflRad, stream);
}

// the partial counter sets of a list build into the counters (and emptied for the next build); accumulating, like the
// reference's atomics: resetinfo starts a new count
static __global__ void __launch_bounds__(NEIBS_SPREAD)
neibs_counters_fold_kernel(NeibsCounters *counters)
{
	__shared__ int sMaxFB[NEIBS_SPREAD], sMaxV[NEIBS_SPREAD];
	__shared__ unsigned long long sSum[NEIBS_SPREAD];
	NeibsSpread *part = reinterpret_cast<NeibsSpread*>(counters + 1) + threadIdx.x;
	sMaxFB[threadIdx.x] = part->maxFluidBoundaryNeibs; sMaxV[threadIdx.x] = part->maxVertexNeibs; sSum[threadIdx.x] = part->numInteractions;
	part->maxFluidBoundaryNeibs = 0; part->maxVertexNeibs = 0; part->numInteractions = 0ull;
	__syncthreads();
	for (int d = NEIBS_SPREAD/2; d > 0; d >>= 1) {
		if ((int)threadIdx.x < d) {
			sMaxFB[threadIdx.x] = max(sMaxFB[threadIdx.x], sMaxFB[threadIdx.x + d]);
			sMaxV[threadIdx.x] = max(sMaxV[threadIdx.x], sMaxV[threadIdx.x + d]);
			sSum[threadIdx.x] += sSum[threadIdx.x + d];
		}
		__syncthreads();
	}
	if (threadIdx.x == 0) {
		counters->maxFluidBoundaryNeibs = max(counters->maxFluidBoundaryNeibs, sMaxFB[0]);
		counters->maxVertexNeibs = max(counters->maxVertexNeibs, sMaxV[0]);
		counters->numInteractions += (int)(unsigned int)sSum[0];      // wraps like the reference's 32-bit counter
		counters->numInteractions64 += sSum[0];
	}
}

// Lists of particles by a property, in no particular order: every wave looks at LIST_CHUNKS x 64 consecutive particles, counts its
// takers, reserves their places with ONE atomic and writes them.  (One atomic per 64 particles, the first form of these sweeps, made
// the counter the bottleneck: next to a wall every wave has a taker, tens of thousands of additions to one word are served one
// after the other -- 0.18-0.23 ms per list at 8.6 M particles where the sweep itself is 0.03, profiles/r06_sa_bc_rows.txt.)
#define LIST_CHUNKS 16
template<class PRED>
__device__ __forceinline__ void wave_append_takers(uint32_t n, uint32_t *__restrict__ out, PRED pred)
{
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t first = (blockIdx.x*(blockDim.x >> 6) + (threadIdx.x >> 6))*(64u*LIST_CHUNKS);
	if (first >= n) return;
	unsigned long long masks[LIST_CHUNKS];
	uint32_t total = 0;
#pragma unroll
	for (int c = 0; c < LIST_CHUNKS; ++c) {
		const uint32_t i = first + (uint32_t)c*64u + lane;
		masks[c] = __builtin_amdgcn_ballot_w64(i < n && pred(i));
		total += (uint32_t)__builtin_popcountll(masks[c]);
	}
	if (!total) return;
	uint32_t base = 0;
	if (lane == 0) base = atomicAdd(out, total);
	base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
#pragma unroll
	for (int c = 0; c < LIST_CHUNKS; ++c) {
		const unsigned long long m = masks[c];
		if ((m >> lane) & 1ull) out[1u + base + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull))] = first + (uint32_t)c*64u + lane;
		base += (uint32_t)__builtin_popcountll(m);
	}
}
static uint32_t list_sweep_grid(uint32_t n) { return div_up_u(n, 256u*LIST_CHUNKS); }

// SA_BOUNDARY: the active particles of type `ptype` (fluid; vertex for moving bodies) whose boundary section is not empty
static __global__ void __launch_bounds__(256)
sa_wall_list_kernel(const neibdata *__restrict__ list, const particleinfo *__restrict__ info, const float4 *__restrict__ pos,
	uint32_t n, uint32_t stride, uint32_t neibboundpos, uint32_t *__restrict__ wall, uint32_t ptype)
{
	wave_append_takers(n, wall, [&](uint32_t i) {
		return PART_TYPE(info[i]) == ptype && is_active_w(pos[i].w) && list[(size_t)neibboundpos*stride + i] != NEIBS_END;
	});
}

// ... and every particle of type `ptype` below n (the rows of the boundary-condition passes)
static __global__ void __launch_bounds__(256)
sa_type_list_kernel(const particleinfo *__restrict__ info, uint32_t n, uint32_t *__restrict__ rows, uint32_t ptype)
{
	wave_append_takers(n, rows, [&](uint32_t i) { return PART_TYPE(info[i]) == ptype; });
}

extern "C" int sphx_build_neibs_sa(sphx_ctx *ctx, uint16_t *neibsList, void *vertPos0, void *vertPos1, void *vertPos2,
	const void *pos, const void *info, const void *vertices, const void *boundElements, const uint32_t *hash,
	const uint32_t *cellStart, const uint32_t *cellEnd,
	uint32_t numParticles, uint32_t particleRangeEnd, uint32_t gridCells,
	float sqinfluenceradius, float boundNlSqInflRad, void *stream)
{
	SPHX_REQUIRE(ctx && ctx->have_params, "sphx_build_neibs: constants not set");
	const bool sa = ctx->params.boundarytype == SPHX_SA_BOUNDARY;
	SPHX_REQUIRE(!sa || (vertPos0 && vertPos1 && vertPos2 && vertices && boundElements),
		"sphx_build_neibs_sa: SA_BOUNDARY needs BUFFER_VERTICES, BUFFER_BOUNDELEMENTS and BUFFER_VERTPOS");
	SPHX_REQUIRE(neibsList && pos && info && hash && cellStart && cellEnd, "sphx_build_neibs: missing buffer");
	SPHX_REQUIRE(gridCells == ctx->params.gridSize[0]*ctx->params.gridSize[1]*ctx->params.gridSize[2],
		"sphx_build_neibs: gridCells does not match the grid set by set_constants");
	SPHX_REQUIRE(particleRangeEnd <= ctx->params.neiblist_stride, "sphx_build_neibs: range exceeds the neighbour list stride");
	if (!particleRangeEnd) return SPHX_OK;
	hipStream_t st = (hipStream_t)stream;
	int rc = sphx_ensure_scratch(ctx, numParticles);
	if (rc != SPHX_OK) return rc;
	SPHX_REQUIRE(gridCells <= ctx->cells_reserved, "sphx_build_neibs: grid larger than reserved");
	// per-cell end of the fluid segment (particles of a cell are sorted fluid-first)
	SPHX_HIP(hipMemcpyAsync(ctx->cell_fluid_end, cellStart, sizeof(uint32_t)*(size_t)gridCells, hipMemcpyDeviceToDevice, st));
	cell_fluid_end_kernel<<<div_up_u(numParticles, 256), 256, 0, st>>>(ctx->cell_fluid_end, hash, (const particleinfo*)info, numParticles);
	SPHX_LAUNCH_CHECK("cell_fluid_end_kernel");
	// tiling of the sorted particles for the forces engine (forces.hip "Tiled path")
	ctx->tiles_built = false;
	ctx->tiles_overflow = -1;
	const bool tile_cols_fit = (size_t)((ctx->dev.gs[ctx->dev.c2] + 1)/2)*(size_t)((ctx->dev.gs[ctx->dev.c3] + 1)/2)*(size_t)ctx->dev.gs1
		<= (size_t)ctx->cells_reserved/2 + 1024;   // tile_cols allocation (degenerate 1-D grids: generic kernels)
	// tiles serve sphx_forces_basicstep's pair loop only (SPH_F1, inviscid or Newtonian, DYN / LJ / MK boundaries) and sphx_calc_visc
	// ... and, with SA_BOUNDARY, the particle <- particle sums of the SA forces, density summation and density diffusion
	// (one fluid; with k-epsilon the density summation and the diffusion, which do not involve the model: sphx_sa_tiles_run, forces.hip)
	// (a run with open boundaries rebuilds the list in every step; the fluid <- fluid sums of its passes are the solid-wall ones and go
	// through the tiles since round 6: the tile lists of a rebuild cost a tenth of what its list walkers did per step)
	const bool tiled_options = ctx->params.sph_formulation == SPHX_SPH_F1 && ctx->params.rheologytype <= SPHX_NEWTONIAN &&
		(!sa || (ctx->dev.numfluids == 1 && (ctx->dev.turbmodel == SPHX_LAMINAR_FLOW || ctx->dev.turbmodel == SPHX_KEPSILON)));
	if (tiled_options && ctx->tiles && !ctx->disable_tiles && tile_cols_fit) {
		rc = sphx_ensure_tile_lists(ctx);      // first tiled build: the tile lists are allocated now (or never: generic kernels)
		if (rc != SPHX_OK) return rc;
	}
	bool tiling_on_side = false;
	// whatever exit this function takes once the tiling has been forked to the side stream, the caller's stream waits for it:
	// a retry or the next build would otherwise race with kernels still writing the tiles (and such a tiling is not used)
	struct SideJoin {
		sphx_ctx *ctx; hipStream_t st; bool armed;
		~SideJoin() {
			if (!armed) return;
			(void)hipEventRecord(ctx->side_join, ctx->side_stream);
			(void)hipStreamWaitEvent(st, ctx->side_join, 0);
			(void)hipGetLastError();
			ctx->tiles_built = false;
		}
	} sideJoin = { ctx, st, false };
	if (tiled_options && ctx->tiles && !ctx->disable_tiles && tile_cols_fit && ctx->tile_list) {
		// The tiling reads the cell tables only and is needed by the tile lists, not by the list build: it runs beside
		// build_neibs_kernel on the context's side stream (build_tiles_kernel is a serial walk per row bundle, ~400 waves for
		// ~1 ms at 32 M particles, on a GPU that holds 8000).  Fork behind everything the caller has queued, join before the
		// tile lists.  Without the side stream (creation failed) the same launches go to the caller's stream
		static const bool noSide = getenv("SPHX_TILING_INLINE") != nullptr;      // A/B switch: the tiling on the caller's stream
		if (!ctx->side_stream && !noSide) {
			// SPHX_SIDE_PRIORITY=1 (experiment of round 6): the side stream at the highest priority, so that the dispatcher prefers
			// its workgroups whenever room comes free on a CU
			static const bool sidePrio = getenv("SPHX_SIDE_PRIORITY") != nullptr;
			int prLeast = 0, prGreatest = 0;
			if (sidePrio) (void)hipDeviceGetStreamPriorityRange(&prLeast, &prGreatest);
			if ((sidePrio ? hipStreamCreateWithPriority(&ctx->side_stream, hipStreamNonBlocking, prGreatest)
			              : hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking)) != hipSuccess) { (void)hipGetLastError(); ctx->side_stream = nullptr; }
			else if (hipEventCreateWithFlags(&ctx->side_fork, hipEventDisableTiming) != hipSuccess ||
			         hipEventCreateWithFlags(&ctx->side_join, hipEventDisableTiming) != hipSuccess) {
				(void)hipGetLastError(); (void)hipStreamDestroy(ctx->side_stream); ctx->side_stream = nullptr;
			}
		}
		hipStream_t ts = st;
		if (ctx->side_stream) {
			SPHX_HIP(hipEventRecord(ctx->side_fork, st));
			SPHX_HIP(hipStreamWaitEvent(ctx->side_stream, ctx->side_fork, 0));
			ts = ctx->side_stream; tiling_on_side = true; sideJoin.armed = true;
		}
		SPHX_HIP(hipMemcpyAsync(ctx->cell_end_copy, cellEnd, sizeof(uint32_t)*(size_t)gridCells, hipMemcpyDeviceToDevice, ts));
		SPHX_HIP(hipMemsetAsync(ctx->tile_ctl, 0, 2*sizeof(uint32_t), ts));
		SPHX_HIP(hipMemsetAsync(ctx->tile_ctl + 12, 0, 2*sizeof(uint32_t), ts));      // cursors of the list stream and of the lane tables
		const DevParams &dp = ctx->dev;
		const uint32_t gs2 = (uint32_t)dp.gs[dp.c2], gs3 = (uint32_t)dp.gs[dp.c3];
		const uint32_t bundles = ((gs2 + 1)/2)*((gs3 + 1)/2);
		tile_columns_kernel<<<div_up_u(bundles*(uint32_t)dp.gs1, 256), 256, 0, ts>>>(ctx->dev, cellStart, ctx->cell_end_copy,
			(const particleinfo*)info, ctx->tile_cols);
		SPHX_LAUNCH_CHECK("tile_columns_kernel");
		const uint32_t tileThreads = ((gs2 + 1)/2 + 7)/8*(((gs3 + 1)/2 + 3)/4)*32u;   // 8 x 4 blocks of bundles, see the kernel
		build_tiles_kernel<<<div_up_u(tileThreads, 128), 128, 0, ts>>>(ctx->dev, cellStart, ctx->cell_end_copy,
			ctx->tile_cols, particleRangeEnd, ctx->tiles, ctx->tile_ctl, ctx->tile_capacity);
		SPHX_LAUNCH_CHECK("build_tiles_kernel");
		if (tiling_on_side) SPHX_HIP(hipEventRecord(ctx->side_join, ctx->side_stream));
		ctx->tiles_built = true;
		ctx->tiles_cellstart = cellStart;
		ctx->tiles_neibslist = neibsList;
	}
	// A tiled build in parts.  tile_lists_kernel is a chain of dependent round trips per tile at ten waves per CU (forces.hip): it
	// needs little of the vector units and leaves most of the CU idle; build_neibs_kernel is bound by vector issue.  So the list is
	// built in `list_parts` launches over consecutive particle ranges, and the tile lists of the tiles whose home particles are
	// all listed go to the side stream behind each of them: they run beside the list build of the next part, and only the last
	// part's tile lists are left over at the end.  Lists, counters and tile lists are what one launch each gives (the tiles get
	// their room in the list stream in another order, which nothing reads).
	int parts = 1;
	if (ctx->tiles_built && tiling_on_side && ctx->list_parts > 1 && particleRangeEnd >= 65536u*(uint32_t)ctx->list_parts) {
		if (!ctx->list_part_events) {
			int made = 0;
			for (; made < SPHX_LIST_PARTS_MAX; ++made)
				if (hipEventCreateWithFlags(&ctx->list_part[made], hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); break; }
			if (made == SPHX_LIST_PARTS_MAX) ctx->list_part_events = true;
			else for (int k = 0; k < made; ++k) (void)hipEventDestroy(ctx->list_part[k]);
		}
		if (ctx->list_part_events) parts = ctx->list_parts;
	}
	if (parts > 1) {
		const uint32_t per = div_up_u(div_up_u(particleRangeEnd, (uint32_t)parts), 1024u)*1024u;      // whole workgroups of the list build
		for (int k = 0; k < parts; ++k) {
			const uint32_t from = (uint32_t)k*per, to = (k == parts - 1) ? particleRangeEnd : min(particleRangeEnd, from + per);
			if (from >= particleRangeEnd) break;
			rc = sphx_neibs_list_launch_part(ctx, neibsList, pos, info, hash, cellStart, cellEnd, vertices, boundElements, vertPos0, vertPos1, vertPos2,
				numParticles, from, to, sqinfluenceradius, boundNlSqInflRad, st);      // neibs_build.hip
			if (rc != SPHX_OK) return rc;
			SPHX_HIP(hipEventRecord(ctx->list_part[k], st));
			SPHX_HIP(hipStreamWaitEvent(ctx->side_stream, ctx->list_part[k], 0));
			rc = sphx_tile_lists_launch(ctx, neibsList, info, hash, cellStart, sa, ctx->side_stream, from, (k == parts - 1) ? 0xFFFFFFFFu : to);
			if (rc != SPHX_OK) return rc;
		}
		SPHX_HIP(hipEventRecord(ctx->side_join, ctx->side_stream));
	} else {
		rc = sphx_neibs_list_launch(ctx, neibsList, pos, info, hash, cellStart, cellEnd, vertices, boundElements, vertPos0, vertPos1, vertPos2,
			numParticles, particleRangeEnd, sqinfluenceradius, boundNlSqInflRad, st);      // neibs_build.hip
		if (rc != SPHX_OK) return rc;
	}
	neibs_counters_fold_kernel<<<1, NEIBS_SPREAD, 0, st>>>(ctx->counters_dev);
	SPHX_LAUNCH_CHECK("neibs_counters_fold_kernel");
	if (tiling_on_side) {      // the tiling (and, of a build in parts, the tile lists) is there for everything queued from here on
		SPHX_HIP(hipStreamWaitEvent(st, ctx->side_join, 0));
		sideJoin.armed = false;
	}
	if (sa) {   // the fluid particles with boundary elements in reach
		if (!ctx->sa_wall && hipMalloc((void**)&ctx->sa_wall, sizeof(uint32_t)*((size_t)ctx->reserved_particles + 1)) != hipSuccess) {
			(void)hipGetLastError();
			ctx->sa_wall = nullptr;      // the engines fall back to one thread per particle
		}
		ctx->sa_wall_neibslist = nullptr;
		++ctx->sa_wall_gen;      // what was kept of |grad gamma_as| per list entry belongs to the previous list
		if (!ctx->sa_wall_gen) ctx->sa_wall_gen = 1u;
		if (ctx->sa_wall && !ctx->sa_wall_cache) {      // room for a quarter of the particles next to a wall; the rest recompute
			const size_t cap = (size_t)ctx->reserved_particles/4u + 1024u;
			if (hipMalloc((void**)&ctx->sa_wall_cache, sizeof(float)*SA_WALL_CACHE_ENTRIES*cap) == hipSuccess &&
			    hipMalloc((void**)&ctx->sa_wall_tag, sizeof(float4)*cap) == hipSuccess) {
				ctx->sa_wall_capacity = (uint32_t)cap;
				SPHX_HIP(hipMemsetAsync(ctx->sa_wall_tag, 0, sizeof(float4)*cap, st));
			} else {
				(void)hipGetLastError();
				if (ctx->sa_wall_cache) (void)hipFree(ctx->sa_wall_cache);
				ctx->sa_wall_cache = nullptr; ctx->sa_wall_tag = nullptr; ctx->sa_wall_capacity = 0;
			}
		}
		if (ctx->sa_wall) {
			SPHX_HIP(hipMemsetAsync(ctx->sa_wall, 0, sizeof(uint32_t), st));
			sa_wall_list_kernel<<<list_sweep_grid(particleRangeEnd), 256, 0, st>>>(neibsList, (const particleinfo*)info, (const float4*)pos,
				particleRangeEnd, ctx->dev.stride, ctx->dev.neibboundpos, ctx->sa_wall, (uint32_t)PT_FLUID);
			SPHX_LAUNCH_CHECK("sa_wall_list_kernel");
			if (ctx->params.simflags & SPHX_ENABLE_MOVING_BODIES) {      // the vertex rows: their gamma is integrated by the density summation
				if (!ctx->sa_wall_vert && hipMalloc((void**)&ctx->sa_wall_vert, sizeof(uint32_t)*((size_t)ctx->reserved_particles + 1)) != hipSuccess) {
					(void)hipGetLastError();
					ctx->sa_wall_vert = nullptr;      // one thread per vertex row then
				}
				if (ctx->sa_wall_vert) {
					SPHX_HIP(hipMemsetAsync(ctx->sa_wall_vert, 0, sizeof(uint32_t), st));
					sa_wall_list_kernel<<<list_sweep_grid(particleRangeEnd), 256, 0, st>>>(neibsList, (const particleinfo*)info, (const float4*)pos,
						particleRangeEnd, ctx->dev.stride, ctx->dev.neibboundpos, ctx->sa_wall_vert, (uint32_t)PT_VERTEX);
					SPHX_LAUNCH_CHECK("sa_wall_list_kernel<vertices>");
				}
			}
			// the rows of the two boundary-condition passes: every boundary element, every vertex particle
			ctx->sa_rows_range = 0;
			for (int k = 0; k < 2; ++k) {
				uint32_t *&rows = k ? ctx->sa_rows_vert : ctx->sa_rows_bound;
				if (!rows && hipMalloc((void**)&rows, sizeof(uint32_t)*((size_t)ctx->reserved_particles + 1)) != hipSuccess) {
					(void)hipGetLastError();
					rows = nullptr;      // one thread per particle then
				}
				if (rows) {
					SPHX_HIP(hipMemsetAsync(rows, 0, sizeof(uint32_t), st));
					sa_type_list_kernel<<<list_sweep_grid(particleRangeEnd), 256, 0, st>>>((const particleinfo*)info, particleRangeEnd, rows,
						k ? (uint32_t)PT_VERTEX : (uint32_t)PT_BOUNDARY);
					SPHX_LAUNCH_CHECK("sa_type_list_kernel");
				}
			}
			if (ctx->sa_rows_bound && ctx->sa_rows_vert) ctx->sa_rows_range = particleRangeEnd;
			ctx->sa_wall_neibslist = neibsList;
		}
	}
	if (ctx->tiles_built) {   // the lists of the tiled particles in the form the tiled forces kernel walks (forces.hip)
		if (parts == 1) rc = sphx_tile_lists_launch(ctx, neibsList, info, hash, cellStart, sa, st);
		if (rc != SPHX_OK) return rc;
		// the tiling's overflow flag travels to the host behind the build, without a synchronisation: the forces passes that
		// find it arrived (sphx_tiles_overflow_poll) launch exactly one kernel, the others keep the guarded stand-by
		if (ctx->tiles_built && ctx->ovf_host) {
			SPHX_HIP(hipMemcpyAsync(ctx->ovf_host, ctx->tile_ctl, 2*sizeof(uint32_t), hipMemcpyDeviceToHost, st));
			SPHX_HIP(hipEventRecord(ctx->ovf_event, st));
			ctx->ovf_pending = true;
		}
	}
	return SPHX_OK;
}

extern "C" int sphx_neibs_resetinfo(sphx_ctx *ctx, void *stream)
{
	SPHX_REQUIRE(ctx != nullptr, "sphx_neibs_resetinfo: NULL ctx");
	NeibsCounters z;
	std::memset(&z, 0, sizeof(z));
	z.hasTooManyNeibs = -1;
	// small H2D of a stack object: use the synchronous-with-respect-to-host staging of hipMemcpyAsync
	// from pageable memory (the runtime copies the source before returning)
	SPHX_HIP(hipMemcpyAsync(ctx->counters_dev, &z, sizeof(z), hipMemcpyHostToDevice, (hipStream_t)stream));
	return SPHX_OK;
}

extern "C" int sphx_neibs_interactions64(sphx_ctx *ctx, uint64_t *out, void *stream)
{
	SPHX_REQUIRE(ctx && out, "sphx_neibs_interactions64: NULL argument");
	NeibsCounters c;
	SPHX_HIP(hipMemcpyAsync(&c, ctx->counters_dev, sizeof(c), hipMemcpyDeviceToHost, (hipStream_t)stream));
	SPHX_HIP(hipStreamSynchronize((hipStream_t)stream));
	*out = (uint64_t)c.numInteractions64;
	return SPHX_OK;
}

extern "C" int sphx_neibs_getinfo(sphx_ctx *ctx, sphx_neibs_info *out, void *stream)
{
	SPHX_REQUIRE(ctx && out, "sphx_neibs_getinfo: NULL argument");
	NeibsCounters c;
	SPHX_HIP(hipMemcpyAsync(&c, ctx->counters_dev, sizeof(c), hipMemcpyDeviceToHost, (hipStream_t)stream));
	SPHX_HIP(hipStreamSynchronize((hipStream_t)stream));
	out->numInteractions = c.numInteractions;
	out->maxFluidBoundaryNeibs = c.maxFluidBoundaryNeibs;
	out->maxVertexNeibs = c.maxVertexNeibs;
	out->hasTooManyNeibs = c.hasTooManyNeibs;
	for (int i = 0; i < 3; ++i) out->hasMaxNeibs[i] = c.hasMaxNeibs[i];
	return SPHX_OK;
}
