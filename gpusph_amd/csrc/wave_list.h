// wave_list.h -- one 64-lane wave over ONE particle's neighbour list: 64 entries of a list section decoded at once, and the
// reductions the open-boundary passes (sa_io.hip) need on what the lanes make of them.
//
// Why a wave per particle: the elements of the open faces (a few thousand vertices and segments, and the fluid particles next to
// them) are few, and each costs a hundred neighbours with two powf per neighbour; one thread per particle leaves a handful of
// waves walking their lists entry by entry while the chip idles.  Here lane l takes the l-th entry of the section, the per-entry
// work runs 64 wide, and what is left per element is the reduction.
//
// Order of the sums: the passes that use this are held bit for bit to the one-thread-per-particle statement of the same sums (the
// CPU oracle), because the decisions that hang on them -- whether an open vertex releases a particle in this step -- are
// thresholds, and a run that released one a step late would differ in particle count and ids from then on.  So the lanes'
// terms are added in list order (ordered_sums: a v_readlane and an add per term and live lane, nothing against the powf they
// follow); integer reductions (counts, maxima, the arg-min of find-outgoing) use butterflies.
#pragma once
#include "sphx_internal.h"

#define WAVE_SECTION_FLUID    0      // slots 0, 1, ... up
#define WAVE_SECTION_BOUNDARY 1      // slots neibboundpos, neibboundpos - 1, ... down
#define WAVE_SECTION_VERTEX   2      // slots neibboundpos + 1, ... up (SA_BOUNDARY)

__device__ __forceinline__ unsigned long long wave_ballot(bool b) { return __builtin_amdgcn_ballot_w64(b); }
__device__ __forceinline__ float wave_lane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ uint32_t wave_lane_u(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ uint32_t wave_lanes_below(unsigned long long m, uint32_t lane) { return (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull)); }
__device__ __forceinline__ uint32_t wave_sum_u(uint32_t v)
{
#pragma unroll
	for (int d = 32; d > 0; d >>= 1) v += (uint32_t)__shfl_xor((int)v, d);
	return v;
}
__device__ __forceinline__ uint32_t wave_max_u(uint32_t v)
{
#pragma unroll
	for (int d = 32; d > 0; d >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)v, d); v = o > v ? o : v; }
	return v;
}
__device__ __forceinline__ float wave_min_f(float v)
{
#pragma unroll
	for (int d = 32; d > 0; d >>= 1) v = fminf(v, __shfl_xor(v, d));
	return v;
}

// acc[k] += term[k] of every lane of `mask`, lane after lane: the sum a single thread walking the list would form
template<int N>
__device__ __forceinline__ void ordered_sums(float (&acc)[N], const float (&term)[N], unsigned long long mask)
{
#ifdef SPHX_WAVE_GATHER      // tests/hostemu: the lanes' terms in one meeting of the fibres instead of one per readlane (same additions)
	float all[N][64];
	for (int k = 0; k < N; ++k) SPHX_WAVE_GATHER(term[k], all[k]);
	for (; mask; mask &= mask - 1ull)
		for (int k = 0; k < N; ++k) acc[k] += all[k][__builtin_ctzll(mask)];
	return;
#endif
	while (mask) {
		const int l = __builtin_ctzll(mask);
		mask &= mask - 1ull;
#pragma unroll
		for (int k = 0; k < N; ++k) acc[k] += wave_lane_f(term[k], l);
	}
}

// What lane l holds of entry s0 + l of a section: the neighbour's index and the wave's own particle as seen from the neighbour's
// cell (own position minus the cell offset: subtracting the neighbour's cell-local position gives the relative position, the
// same two roundings as neib_iter.h's walker).
struct WaveEntry { bool live; uint32_t j; float ox, oy, oz; };

// cellCarry: the cell code in force at the end of the previous chunk; more: the section goes on behind this chunk
template<int SECTION>
__device__ __forceinline__ WaveEntry wave_entries(const DevParams &p, const neibdata *__restrict__ list, const uint32_t *__restrict__ cellStart,
	uint32_t index, const float4 &own, const int3 &gridPos, int s0, uint32_t lane, int &cellCarry, bool &more)
{
	const int s = s0 + (int)lane;
	int slot;
	bool inside;
	if (SECTION == WAVE_SECTION_BOUNDARY) { slot = (int)p.neibboundpos - s; inside = slot >= 0; }
	else { slot = (SECTION == WAVE_SECTION_FLUID ? 0 : (int)p.neibboundpos + 1) + s; inside = slot < (int)p.neiblistsize; }
	const uint32_t d = inside ? (uint32_t)list[(size_t)slot*p.stride + index] : (uint32_t)NEIBS_END;
	const unsigned long long ended = wave_ballot(d == NEIBS_END);
	const int firstEnd = ended ? __builtin_ctzll(ended) : 64;
	WaveEntry e;
	e.live = (int)lane < firstEnd;
	more = firstEnd == 64;
	// the cell of an entry is the one the nearest entry at or before it names
	const unsigned long long named = wave_ballot(e.live && d >= CELLNUM_ENCODED);
	const unsigned long long upto = named & (~0ull >> (63u - lane));
	const int from = upto ? 63 - __builtin_clzll(upto) : 0;
	const int theirs = __shfl((int)(d >> CELLNUM_SHIFT), from) - 1;
	const int c = upto ? theirs : cellCarry;
	cellCarry = __shfl(c, 63);
	const int cz = c/9, cy = (c - cz*9)/3, cx = c - cz*9 - cy*3;
	e.j = index; e.ox = own.x; e.oy = own.y; e.oz = own.z;
	if (e.live) {
		e.j = cellStart[grid_hash_periodic(p, gridPos.x + cx - 1, gridPos.y + cy - 1, gridPos.z + cz - 1)] + (d & NEIBINDEX_MASK);
		e.ox = fmaf(-(float)(cx - 1), p.cs[0], own.x);
		e.oy = fmaf(-(float)(cy - 1), p.cs[1], own.y);
		e.oz = fmaf(-(float)(cz - 1), p.cs[2], own.z);
	}
	return e;
}

// Use:   int carry = 0; bool more = true;
//        for (int s0 = 0; more; s0 += 64) { const WaveEntry e = wave_entries<SECTION>(..., s0, lane, carry, more); ... }
// The loop condition is wave-uniform; the body runs with all 64 lanes (dead lanes carry e.live == false).
