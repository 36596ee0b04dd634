// neibs_build.hip -- the cell-linked neighbour list of the neighbour engine for gfx950 (buildNeibsListDevice + neibsInCell,
// src/cuda/buildneibs_kernel.cu:536-644,1019-1185).  Split from neibs.hip in round 6; results bit-identical to the reference
// algorithm restated in oracle/sph_oracle.c.  Compiled with -ffp-contract=off like neibs.hip.
#include "sphx_internal.h"
#include <cstring>
#include <cstdlib>

#define BLOCK_NEIBS   256
// (tests/hostemu compiles this file for the host and runs the kernel's waves as fibres: its stand-in header defines the macro)
#ifndef SPHX_EMU_MARK
#define SPHX_EMU_MARK(n) ((void)0)
#define SPHX_EMU_REFUSE(cond, what) ((void)0)
#endif
#ifndef SPHX_LAUNCH_WAVES
#define SPHX_LAUNCH_WAVES(kernel, grid, block, stream, ...) kernel<<<(grid), (block), 0, (stream)>>>(__VA_ARGS__)
#endif

// ------------------------------------------------------------------------------------------
// buildNeibsList: src/cuda/buildneibs_kernel.cu:1019-1185, neibsInCell :536-644
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t neib_list_offset(const DevParams &p, uint32_t neib_num, uint32_t neib_type)
{
	return (neib_type == PT_FLUID) ? neib_num :
		(neib_type == PT_BOUNDARY) ? p.neibboundpos - neib_num :
		neib_num + p.neibboundpos + 1;
}

__device__ __forceinline__ bool too_many_neibs(const DevParams &p, uint32_t nf, uint32_t nb, uint32_t nv, uint32_t neib_type)
{
	switch (neib_type) {
	case PT_FLUID:    return !(nf < p.neibboundpos);
	case PT_BOUNDARY: return !(nf + nb < p.neibboundpos);
	case PT_VERTEX:   return !(nv < p.neiblistsize - p.neibboundpos - 1);
	default: return true;
	}
}

__device__ __forceinline__ bool neib_cell_axis(int &g, int off, int gs, bool periodic)
{
	g += off;
	if (g < 0) { if (periodic) g = gs - 1; else return false; }
	else if (g >= gs) { if (periodic) g = 0; else return false; }
	return true;
}

#define NEIB_MLP 4   // candidate positions fetched per batch in the fluid segment
#define NEIB_FRING 32   // rows of the fluid section a wave keeps in LDS before writing them out as full lines
#define NEIB_BRING 16   // ... of the boundary section

// Per-wave LDS staging of list entries.  The list is slot-major
// ([slot*stride + particle], 2 B) and the lanes of a wave reach a given slot at different times, so storing
// entries as they are found wrote every 128-B line of the list ~12 times (measured: 53 GB of HBM writes for a
// 4.3 GB list at 32 M particles).  Every lane parks its entries in its own column of a ring of FR (fluid) + BR
// (boundary) rows: the ring holds the window [flushed, stored) of the lane's section.  Rows leave the ring in
// wave-wide steps: when some lane's window is about to fill up, the oldest row r of that lane is written for
// EVERY lane whose next row to flush is r and that has it -- in a wave of neighbouring particles, whose lists grow
// at similar rates, that is most of a 128-B line.  Lanes that lag behind keep their entries and write the
// row later (with the lanes that lag like them).  Nothing is ever stored around the ring, so the hot loop has no
// "direct store" state to carry.
template<int FR, int BR>
struct NeibRing {
	neibdata (*fring)[64];
	neibdata (*bring)[64];
	neibdata *column;        // &list[particle]
	size_t stride;
	uint32_t nbp, lane;
	uint32_t sf, sb;         // entries stored so far (== neibs_num unless the list overflowed)
	uint32_t ff, fb;         // ... of which the first ff / fb have been written to the list

	__device__ __forceinline__ void init(neibdata (*rows)[64], neibdata *col, size_t str, uint32_t neibboundpos, uint32_t ln)
	{
		fring = rows; bring = rows + FR; column = col; stride = str; nbp = neibboundpos; lane = ln;
		sf = sb = ff = fb = 0;
	}
	// wave-uniform control flow (call with the lanes of the wave converged): make room for `need` more fluid entries in
	// every lane; once a lane is short of room, rows are written until it has `need + slack` free
	__device__ __forceinline__ void room_f(uint32_t need, uint32_t slack)
	{
		if (!__builtin_amdgcn_ballot_w64(sf + need - ff > (uint32_t)FR)) return;
		for (;;) {
			const unsigned long long m = __builtin_amdgcn_ballot_w64(sf + need + slack - ff > (uint32_t)FR && sf > ff);
			if (!m) break;
			const uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)ff, (int)__builtin_ctzll(m));
			if (ff == r && sf > r) { column[(size_t)r*stride] = fring[r % FR][lane]; ++ff; }
		}
	}
	__device__ __forceinline__ void room_b(uint32_t need, uint32_t slack)
	{
		if (!__builtin_amdgcn_ballot_w64(sb + need - fb > (uint32_t)BR)) return;
		for (;;) {
			const unsigned long long m = __builtin_amdgcn_ballot_w64(sb + need + slack - fb > (uint32_t)BR && sb > fb);
			if (!m) break;
			const uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)fb, (int)__builtin_ctzll(m));
			if (fb == r && sb > r) { column[(size_t)(nbp - r)*stride] = bring[r % BR][lane]; ++fb; }
		}
	}
	// stores from divergent code (the non-fluid tail of a cell, the terminators): a lane whose window is full writes its
	// own oldest entry first
	__device__ __forceinline__ void store_f(uint32_t slot, uint32_t val)
	{
		if (slot - ff >= (uint32_t)FR) { column[(size_t)ff*stride] = fring[ff % FR][lane]; ++ff; }
		fring[slot % FR][lane] = (neibdata)val;
		sf = slot + 1u;
	}
	__device__ __forceinline__ void store_b(uint32_t k, uint32_t val)   // k-th boundary entry, slot neibboundpos - k
	{
		if (k - fb >= (uint32_t)BR) { column[(size_t)(nbp - fb)*stride] = bring[fb % BR][lane]; ++fb; }
		bring[k % BR][lane] = (neibdata)val;
		sb = k + 1u;
	}
	// end of the walk: whatever is still parked (terminators included); wave-uniform control flow
	__device__ __forceinline__ void finish()
	{
		for (;;) {
			const unsigned long long m = __builtin_amdgcn_ballot_w64(ff < sf);
			if (!m) break;
			const uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)ff, (int)__builtin_ctzll(m));
			if (ff == r && sf > r) { column[(size_t)r*stride] = fring[r % FR][lane]; ++ff; }
		}
		for (;;) {
			const unsigned long long m = __builtin_amdgcn_ballot_w64(fb < sb);
			if (!m) break;
			const uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)fb, (int)__builtin_ctzll(m));
			if (fb == r && sb > r) { column[(size_t)(nbp - r)*stride] = bring[r % BR][lane]; ++fb; }
		}
	}
};

// buildNeibsListDevice + neibsInCell (src/cuda/buildneibs_kernel.cu:536-644,1019-1185).  The candidate
// scan is a gather through L1/L2 and is bound by the number of gather instructions, so it avoids the
// ones the reference cannot: the particleinfo of a candidate is only read in the non-fluid tail of a
// cell (in the fluid segment the type is known from the index), and DYN/LJ boundary particles, which
// never list boundary neighbours, do not visit the non-fluid tail at all.  Candidate order, tests and
// encodings are the reference's, so the list is bit-identical.
// Stores go through NeibRing (above).
typedef uint32_t neib_u32x4 __attribute__((ext_vector_type(4)));
// candidate row j0 + u of the position array as a buffer load: descriptor in SGPRs, byte offset j0*16 in one VGPR,
// u*16 in the instruction -- no per-candidate address arithmetic; rows past the array read as zeros
__device__ __forceinline__ float4 load_pos_row(__amdgpu_buffer_rsrc_t rsrc, uint32_t voff, int u)
{
	const neib_u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(voff + 16u*(uint32_t)u), 0, 0);
	return make_float4(__uint_as_float(r[0]), __uint_as_float(r[1]), __uint_as_float(r[2]), __uint_as_float(r[3]));
}

// SA_BOUNDARY members of buildneibs_params (src/cuda/buildneibs_params.h:66-115)
struct SaNeibArgs {
	const uint4 *vertices;          // vertexinfo of the segments
	const float4 *boundElements;    // normal + area of the segments
	float2 *vertPos[3];             // [out] in-plane offsets of a segment's three vertices
	float boundNlSqInflRad;         // search radius for boundary neighbours
};


// ------------------------------------------------------------------------------------------
// The distance tests of the list build on the matrix cores (round 6).
//
// The walk below ("general walk") tests ~475 candidates per particle at ~17 vector instructions each, one lane per home
// particle: 7.7 G vector instructions per build at 32 M particles, 13 ms.  The test itself -- |x_i - x_j|^2 < R^2 for every
// home particle of a wave against every particle of the cells around them -- is a dense contraction: with the wave's 64 home
// particles as columns and 32 candidates as rows, |c|^2 - 2 c.h + (|h|^2 - R^2) is a 32 x 64 product of inner dimension 4, two
// v_mfma_f32_32x32x2_f32 per half of the wave, and what is left to the vector unit per pair is ONE instruction that shifts the
// result's sign into a bit mask (v_alignbit) and a third of one that tracks the smallest |result| (v_min3_u32).  The 64 home
// particles of a wave are consecutive in the sorted order, i.e. they sit in a few adjacent cells of one grid row (cells along
// COORD1 are contiguous in memory), so the candidates of ALL of them are nine contiguous index ranges, one per neighbouring
// row: cells [cmin - 1, cmax + 1] of that row.  A "group" is the lanes of a wave that share a grid row.
//
// What keeps the list bit-identical to the reference's: the matrix product does not reproduce the reference's roundings
// (x_i - shift first, then the difference, then three fused multiply-adds), so its verdict is only trusted away from the
// surface of the sphere: a result within `band` of zero (band = a bound of everything the two evaluations can differ by, from
// the magnitudes of the operands, see nm_band) sends that home particle's 32 candidates of the tile through the reference's
// own arithmetic.  That is ~1 tile in 50.  Order, encodings, overflow rules and the ring are the general walk's: per cell in
// cell-code order, fluid segment then tail, the candidates of a cell in index order = the set bits of the mask in ascending
// order.  Everything unusual stays with the general walk, lane by lane: COORD1 = z (the cells of a code triple are then not
// one row), particles in the first / last cell of a periodic COORD1, rows whose candidate range exceeds NM_TMAX tiles even for
// a single column of home cells, SA_BOUNDARY (vertex section, VERTPOS) -- a lane the prepass does not take is simply still
// `walking` afterwards.
//
// MEASURED (round 6, MI355X, DamBreak3D 31.8 M particles, profiles/r06_neibs_mfma.txt): lists bit-identical to the oracle's up to
// 32 M particles (tests/test_gpu_parity.py with SPHX_NEIBS_MFMA=1, tests/test_neibs_mfma_hostemu.py), and SLOWER than the general
// walk: 18.8 - 19.8 ms per launch against 12.96.  Where it goes (variants of the same launch): masks alone 7.8 ms, the emission's
// scaffolding with empty masks (27 x cell tables + mask words per lane, one memory round trip each at three waves per SIMD -- the
// masks' LDS and the ring allow no more) 10.4 ms, the entries themselves ~4 ms.  And the premise was wrong by a factor: fp32 on the
// matrix cores has the SAME rate as on the vector unit (256 flop/clk/CU either way; only the lower precisions are faster), so the
// product buys issue slots, not throughput -- 45 tiles x 6 v_mfma x 64 cycles are 3.5 ms of every SIMD's matrix pipe per launch
// before anything else.  It is therefore OFF unless SPHX_NEIBS_MFMA=1 (kept: it is a second, independent implementation of
// the list build that the parity tests hold against the first).
// ------------------------------------------------------------------------------------------
#define NM_TMAX 6                    // tiles of 32 candidates per row whose masks a lane keeps (192 candidates)
#define NM_WORDS (NM_TMAX + 2)       // mask words per lane and row in LDS: the tiles + zeros for the 96-bit reads of the emission
#define NM_MAXSPAN 40                // cells of one group along COORD1 (+ 2 neighbours: one lane each in the range look-up)
typedef float nm_f16 __attribute__((ext_vector_type(16)));
typedef uint32_t nm_u2 __attribute__((ext_vector_type(2)));

// both halves of the wave see lanes 0..31 (x) / lanes 32..63 (y) of v: lane l gets v[l & 31] and v[32 + (l & 31)]
__device__ __forceinline__ void nm_both_halves(uint32_t v, uint32_t &lo, uint32_t &hi)
{
	const nm_u2 r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
	lo = r[0]; hi = r[1];
}
__device__ __forceinline__ void nm_both_halves(float v, float &lo, float &hi)
{
	uint32_t a, b;
	nm_both_halves(__float_as_uint(v), a, b);
	lo = __uint_as_float(a); hi = __uint_as_float(b);
}
__device__ __forceinline__ unsigned long long nm_low_bits(uint32_t n)      // n <= 64 ones
{
	return n >= 64u ? ~0ull : ((1ull << n) - 1ull);
}

// MC1: the axis of COORD1 (0 = x, 1 = y).  sMask: [3 rows of a slab][64 lanes][NM_WORDS], this wave's
template<int MC1, int FR, int BR>
__device__ __forceinline__ void neibs_mfma_prepass(const DevParams &p, const float4 *__restrict__ posArray,
	const particleinfo *__restrict__ infoArray, const uint32_t *__restrict__ particleHash,
	const uint32_t *__restrict__ cellStart, const uint32_t *__restrict__ cellEnd, const uint32_t *__restrict__ cellFluidEnd,
	float sqinfluenceradius, uint32_t lane, uint32_t index, const float4 &pos, const int3 &gridPos, const particleinfo &info,
	bool fluidOnly, bool noBB, bool &walking, uint32_t &nf, uint32_t &nb, uint32_t &nv, NeibRing<FR, BR> &ring, neibdata *column,
	uint32_t (*sMask)[64][NM_WORDS])
{
	constexpr int OA = 1 - MC1;      // the other horizontal axis; z is never COORD1 here
	const int gs1 = p.gs[MC1], gsOA = p.gs[OA], gsZ = p.gs[2];
	const float cs1 = p.cs[MC1], csOA = p.cs[OA], csZ = p.cs[2];
	const bool per1 = (p.periodic & (1u << MC1)) != 0u, perOA = (p.periodic & (1u << OA)) != 0u, perZ = (p.periodic & 4u) != 0u;
	const int c1 = MC1 == 0 ? gridPos.x : gridPos.y, oa = MC1 == 0 ? gridPos.y : gridPos.x, gz = gridPos.z;
	const float pos1 = MC1 == 0 ? pos.x : pos.y, posOA = MC1 == 0 ? pos.y : pos.x;
	const int rowKey = oa*p.hs[OA] + gz*p.hs[2];      // hash of the row's column 0 (hs[COORD1] = 1)
	const bool boundary = IS_BOUNDARY(info);
	const bool lowHalf = lane < 32u;
	// candidate (row of the matrix) -> place in its tile of 32: the product leaves candidate rows 8a + 4h + b (a < 4, h < 2, b < 4)
	// in register 4a + b of half h of the wave; numbered this way the sixteen bits a lane collects are sixteen CONSECUTIVE candidates
	const uint32_t li = lane & 31u;
	const uint32_t candPlace = 16u*((li >> 2) & 1u) + 4u*(li >> 3) + (li & 3u);

	bool todo = walking && !(per1 && (c1 == 0 || c1 == gs1 - 1));
	for (;;) {
		SPHX_EMU_MARK(1);
		const unsigned long long remaining = __builtin_amdgcn_ballot_w64(todo);
		if (!remaining) break;
		const int lead = __builtin_ctzll(remaining);
		const int key = __builtin_amdgcn_readlane(rowKey, lead);
		const unsigned long long gm0 = __builtin_amdgcn_ballot_w64(todo && rowKey == key);
		const int cmin = __builtin_amdgcn_readlane(c1, __builtin_ctzll(gm0));
		int cmax = __builtin_amdgcn_readlane(c1, 63 - __builtin_clzll(gm0));      // lanes are in hash order: c1 does not decrease
		if (cmax - cmin > NM_MAXSPAN) cmax = cmin + NM_MAXSPAN;
		const int oaG = __builtin_amdgcn_readlane(oa, lead), zG = __builtin_amdgcn_readlane(gz, lead);

		// the nine rows: hash of their column 0 (or invalid), and the index range of their cells [cmin - 1, cmax + 1]
		int rowBase[9];
		uint32_t rowLo[9], rowHi[9];
		bool fits = false;
		for (;;) {
			SPHX_EMU_MARK(2);
			const int colLo = max(cmin - 1, 0), colHi = min(cmax + 1, gs1 - 1);
			uint32_t longest = 0;
#pragma unroll
			for (int r9 = 0; r9 < 9; ++r9) {
				int oar = oaG + (r9 % 3 - 1), zr = zG + (r9 / 3 - 1);
				bool rv = true;
				if (oar < 0) { if (perOA) oar = gsOA - 1; else rv = false; } else if (oar >= gsOA) { if (perOA) oar = 0; else rv = false; }
				if (zr < 0) { if (perZ) zr = gsZ - 1; else rv = false; } else if (zr >= gsZ) { if (perZ) zr = 0; else rv = false; }
				rowBase[r9] = rv ? oar*p.hs[OA] + zr*p.hs[2] : -1;
				rowLo[r9] = 0u; rowHi[r9] = 0u;
				if (!rv) continue;
				const bool mine = (int)lane <= colHi - colLo;
				const uint32_t h = (uint32_t)(rowBase[r9] + colLo + (mine ? (int)lane : 0));
				const uint32_t cs = cellStart[h];
				const bool nonEmpty = mine && cs != CELL_EMPTY;
				const uint32_t ce = nonEmpty ? cellEnd[h] : 0u;
				const unsigned long long m = __builtin_amdgcn_ballot_w64(nonEmpty);
				if (m) {
					rowLo[r9] = (uint32_t)__builtin_amdgcn_readlane((int)cs, __builtin_ctzll(m));
					rowHi[r9] = (uint32_t)__builtin_amdgcn_readlane((int)ce, 63 - __builtin_clzll(m));
					longest = max(longest, rowHi[r9] - rowLo[r9]);
				}
			}
			if (longest <= 32u*NM_TMAX) { fits = true; break; }
			if (cmax == cmin) break;
			cmax = cmin + (cmax - cmin)/2;
		}
		if (!fits) {      // one column of home cells whose neighbourhood does not fit: the general walk takes these lanes
			todo = todo && !(rowKey == key && c1 == cmin);
			continue;
		}
		const bool inG = todo && rowKey == key && c1 <= cmax;
		todo = todo && !inG;
		walking = walking && !inG;

		// the group's frame along COORD1: column cref; the band inside which the product's verdict is not trusted
		const int cref = (cmin + cmax) >> 1;
		const float h1 = fmaf((float)(c1 - cref), cs1, pos1);
		float band;
		{
			// |operands| <= m1 along COORD1 (half the span + the neighbour column + half a cell), 1.5 cells along the other two (the
			// home particle shifted into the neighbouring row's frame); M bounds |c|^2 and |h|^2.  The product's value differs from
			// the exact |x_i - x_j|^2 of the same inputs by at most the roundings of the frame conversion (2^-23 of a coordinate,
			// times 2 r), of |c|^2, |h|^2 (3 x 2^-24 M each) and of four multiply-accumulate steps on magnitudes <= 4 M
			// (2^-23 each if the unit rounds product and sum separately; the constant term enters as a fifth, exact, product): < 2^-19 M + 2^-18 R^2; the reference's own value differs
			// from the exact one by < 2^-21 R^2.  The band is twice that sum.
			const float m1 = ((float)(cmax - cmin)*0.5f + 2.0f)*cs1;
			const float M = fmaf(m1, m1, fmaf(1.5f*csOA, 1.5f*csOA, (1.5f*csZ)*(1.5f*csZ)));
			band = fmaf(M, 1.0f/262144.0f, sqinfluenceradius*(1.0f/131072.0f));
		}
		const uint32_t ambBits = __float_as_uint(2.0f*band);      // 0 <= result + band < 2 band: inside the band
		float h1s[2], hoas[2], hzs[2];
		nm_both_halves(h1, h1s[0], h1s[1]);
		nm_both_halves(posOA, hoas[0], hoas[1]);
		nm_both_halves(pos.z, hzs[0], hzs[1]);

#pragma unroll 1
		for (int sl = 0; sl < 3; ++sl) {      // slab: dz = sl - 1
			const float shZ = (float)(sl - 1)*csZ;
			// this slab's three rows (static indices only: a run-time index would send the tables to scratch memory)
			int curBase[3]; uint32_t curLo[3], curHi[3];
#pragma unroll
			for (int rs = 0; rs < 3; ++rs) {
				curBase[rs] = sl == 0 ? rowBase[rs] : sl == 1 ? rowBase[3 + rs] : rowBase[6 + rs];
				curLo[rs] = sl == 0 ? rowLo[rs] : sl == 1 ? rowLo[3 + rs] : rowLo[6 + rs];
				curHi[rs] = sl == 0 ? rowHi[rs] : sl == 1 ? rowHi[3 + rs] : rowHi[6 + rs];
			}
			// ---- the masks of the slab's three rows.  The candidates of a row (<= NM_TMAX tiles: a position row and a hash per lane
			// and tile) are requested together, so that a row costs one memory round trip (round 6, first version: one exposed trip
			// per TILE, ~40 per group at three waves per SIMD: the prepass lost to the walk it replaces, 18.8 against 13.0 ms at 32 M)
			auto request_row = [&](uint32_t lo, uint32_t hi, float4 (&cp)[NM_TMAX], uint32_t (&ch)[NM_TMAX]) {
#pragma unroll
				for (int t = 0; t < NM_TMAX; ++t) {
					const uint32_t j = lo + 32u*(uint32_t)t + candPlace;
					if (lo + 32u*(uint32_t)t < hi) {      // (wave-uniform)
						const uint32_t jj = j < hi ? j : lo;
						cp[t] = posArray[jj]; ch[t] = particleHash[jj];
					}
				}
			};
			auto compute_row = [&](int rs, const float4 (&cpv)[NM_TMAX], const uint32_t (&chv)[NM_TMAX]) {
				uint32_t *myWords = sMask[rs][lane];
#pragma unroll
				for (int w = 0; w < NM_WORDS; ++w) myWords[w] = 0u;
				const uint32_t lo = curLo[rs], hi = curHi[rs];
				const int rbase = curBase[rs];
				if (hi == lo) return;
				const float shOA = (float)(rs - 1)*csOA;
				// the home particle in this row's frame (the reference shifts the home particle, not the candidate)
				const float hOAr = posOA - shOA, hZr = pos.z - shZ;
				const float nOwn = fmaf(hZr, hZr, fmaf(hOAr, hOAr, h1*h1));
				// |h|^2 - R^2 + band, the term that depends on the home particle alone, rides as a fifth inner dimension (candidate
				// side 1): the accumulator starts from the literal zero and needs no sixteen registers of copies per half
				float cinit[2];
				nm_both_halves((nOwn - sqinfluenceradius) + band, cinit[0], cinit[1]);
				float b1[2], b2[2], b3[2];
#pragma unroll
				for (int s = 0; s < 2; ++s) {
					b1[s] = lowHalf ? h1s[s] : hoas[s] - shOA;
					b2[s] = lowHalf ? hzs[s] - shZ : 1.0f;
					b3[s] = lowHalf ? cinit[s] : 0.0f;
				}
				const float a3 = lowHalf ? 1.0f : 0.0f;
#pragma unroll
				for (int t = 0; t < NM_TMAX; ++t) {
					const uint32_t tileBase = lo + 32u*(uint32_t)t;
					if (tileBase >= hi) break;
					SPHX_EMU_MARK(100 + 10*rs + t);
					const uint32_t j = tileBase + candPlace;
					const bool inRow = j < hi;
					const float4 cp = cpv[t];
					const int col = (int)(chv[t] & CELLTYPE_BITMASK) - rbase;
					const float c1f = fmaf((float)(col - cref), cs1, MC1 == 0 ? cp.x : cp.y);
					const float cOA = MC1 == 0 ? cp.y : cp.x;
					const float n = fmaf(cp.z, cp.z, fmaf(cOA, cOA, c1f*c1f));
					const bool ok = inRow && is_active_w(cp.w) && n < __builtin_inff();
					const float a1 = ok ? -2.0f*(lowHalf ? c1f : cOA) : 0.0f;
					const float a2 = lowHalf ? (ok ? -2.0f*cp.z : 0.0f) : (ok ? n : __builtin_inff());
					uint32_t q[2], amb[2];
#pragma unroll
					for (int s = 0; s < 2; ++s) {
						nm_f16 acc;
#pragma unroll
						for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
						acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a3, b3[s], acc, 0, 0, 0);
						acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1[s], acc, 0, 0, 0);
						acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b2[s], acc, 0, 0, 0);
						uint32_t bits = 0u, smallest = 0xFFFFFFFFu;
#pragma unroll
						for (int r = 15; r >= 0; --r) {
							const uint32_t v = __float_as_uint(acc[r]);
							bits = __builtin_amdgcn_alignbit(bits, v, 31);      // (bits << 1) | sign: negative = inside the sphere
							smallest = min(smallest, v);                       // negative values are the largest unsigned ones
						}
						q[s] = bits; amb[s] = smallest < ambBits ? 1u : 0u;
					}
					// lane p < 32 holds home p: bits of candidates 0..15 in q[0], of 16..31 in lane p + 32's q[0]; lane p + 32 holds
					// home p + 32: candidates 0..15 in lane p's q[1], 16..31 in its own q[1] -- one swap of the halves sorts that out
					const nm_u2 sw = __builtin_amdgcn_permlane32_swap(q[0], q[1], false, false);
					uint32_t mask = sw[0] | (sw[1] << 16);
					const nm_u2 sa = __builtin_amdgcn_permlane32_swap(amb[0], amb[1], false, false);
					const bool recheck = inG && (sa[0] | sa[1]) != 0u;
					if (__builtin_amdgcn_ballot_w64(recheck)) {
						// the reference's arithmetic for the 32 candidates of this tile (neibsInCell: the home particle shifted by the
						// cell offset, then the difference, then the squared length) for the home particles that ask for it
						uint32_t exact = 0u;
						const float pOA = posOA - shOA, pZ = pos.z - shZ;      // = fmaf(-(float)offset, cell size, pos) of the reference
#pragma unroll 1
						for (uint32_t u = 0; u < 32u; ++u) {
							const uint32_t ju = tileBase + u;
							if (ju >= hi) break;
							const float4 cu = posArray[ju];
							const int d1 = (int)(particleHash[ju] & CELLTYPE_BITMASK) - rbase - c1;
							const float p1 = fmaf(-(float)d1, cs1, pos1);
							const float r1 = p1 - (MC1 == 0 ? cu.x : cu.y), rOA = pOA - (MC1 == 0 ? cu.y : cu.x), rz = pZ - cu.z;
							const float rx = MC1 == 0 ? r1 : rOA, ry = MC1 == 0 ? rOA : r1;
							const float r2 = fmaf(rz, rz, fmaf(ry, ry, rx*rx));
							if (r2 < sqinfluenceradius && is_active_w(cu.w)) exact |= 1u << u;
						}
						mask = recheck ? exact : mask;
					}
					myWords[t] = mask;
				}
			};
#pragma unroll
			for (int rs = 0; rs < 3; ++rs) {
				float4 cpv[NM_TMAX]; uint32_t chv[NM_TMAX];
#pragma unroll
				for (int t = 0; t < NM_TMAX; ++t) { cpv[t] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); chv[t] = 0u; }
				request_row(curLo[rs], curHi[rs], cpv, chv);
				compute_row(rs, cpv, chv);
			}
			// ---- emission: the nine cells of the slab in cell-code order
			struct CellMeta { uint32_t start, fluidEnd, end; };
			auto cell_meta = [&](int cc) {
				const int dx = cc % 3 - 1, dy = cc/3 - 1;
				const int d1 = MC1 == 0 ? dx : dy, rs = (MC1 == 0 ? dy : dx) + 1;
				const int col = c1 + d1, base = rs == 0 ? curBase[0] : rs == 1 ? curBase[1] : curBase[2];
				const bool valid = inG && base >= 0 && col >= 0 && col < gs1;
				const uint32_t h = valid ? (uint32_t)(base + col) : 0u;
				CellMeta m;
				m.start = cellStart[h]; m.fluidEnd = cellFluidEnd[h]; m.end = cellEnd[h];
				if (!valid) m.start = CELL_EMPTY;
				return m;
			};
			CellMeta nextMeta = cell_meta(0);
#pragma unroll 1
			for (int cc = 0; cc < 9; ++cc) {
				SPHX_EMU_MARK(200 + cc);
				const CellMeta cur = nextMeta;
				nextMeta = cell_meta(min(cc + 1, 8));
				const int dx = cc % 3 - 1, dy = cc/3 - 1;
				const int rs = (MC1 == 0 ? dy : dx) + 1;
				const uint32_t cell = (uint32_t)(cc + 9*sl);
				const uint32_t code = (cell + 1u) << CELLNUM_SHIFT;
				const bool has = cur.start != CELL_EMPTY;
				const uint32_t bucketEnd = fluidOnly ? cur.fluidEnd : cur.end;
				const uint32_t len = has ? bucketEnd - cur.start : 0u, flen = has ? min(cur.fluidEnd - cur.start, len) : 0u;
				const uint32_t off = has ? cur.start - (rs == 0 ? curLo[0] : rs == 1 ? curLo[1] : curLo[2]) : 0u;
				const uint32_t selfrel = (cell == 13u) ? index - cur.start : 0xFFFFFFFFu;
				uint32_t encv = code, neib_type = PT_FLUID;
				const uint32_t *myWords = sMask[rs][lane];
				for (uint32_t k0 = 0; __builtin_amdgcn_ballot_w64(k0 < len); k0 += 64u) {
					unsigned long long bits = 0ull;
					if (k0 < len) {
						const uint32_t bo = off + k0, wi = bo >> 5, sh = bo & 31u;
						const uint32_t w0 = myWords[wi], w1 = myWords[wi + 1u], w2 = myWords[wi + 2u];
						const uint32_t lo32 = __builtin_amdgcn_alignbit(w1, w0, sh), hi32 = __builtin_amdgcn_alignbit(w2, w1, sh);
						bits = ((unsigned long long)hi32 << 32) | lo32;
						bits &= nm_low_bits(len - k0);
						if (selfrel - k0 < 64u) bits &= ~(1ull << (selfrel - k0));
					}
					const unsigned long long fmask = nm_low_bits(flen > k0 ? flen - k0 : 0u);
					unsigned long long fm = bits & fmask, tm = bits & ~fmask;
					// fluid segment: the type is known from the index
					while (__builtin_amdgcn_ballot_w64(fm != 0ull)) {
						ring.room_f(1, 4);
						if (fm) {
							const uint32_t b = (uint32_t)__builtin_ctzll(fm);
							fm &= fm - 1ull;
							nf += 1u;
							if (!too_many_neibs(p, nf, nb, nv, PT_FLUID)) {
								ring.fring[ring.sf % (uint32_t)FR][lane] = (neibdata)(k0 + b + encv);
								ring.sf += 1u;
								encv = 0u;
							}
						}
					}
					// tail of the cell (boundary particles; test points are nobody's neighbours): the reference's steps for the
					// candidates in reach, in index order
					if (__builtin_amdgcn_ballot_w64(tm != 0ull)) {
						ring.room_b(BR/2, 0);
						while (tm) {
							const uint32_t b = (uint32_t)__builtin_ctzll(tm);
							tm &= tm - 1ull;
							const uint32_t neib_index = cur.start + k0 + b;
							const particleinfo neib_info = infoArray[neib_index];
							if (IS_TESTPOINT(neib_info)) continue;
							if (neib_type != PART_TYPE(neib_info)) encv = code;
							neib_type = PART_TYPE(neib_info);
							if (noBB && boundary && IS_BOUNDARY(neib_info)) continue;
							const uint32_t num = (neib_type == PT_FLUID) ? nf : (neib_type == PT_BOUNDARY) ? nb : nv;
							if (neib_type == PT_FLUID) nf++; else if (neib_type == PT_BOUNDARY) nb++; else nv++;
							if (!too_many_neibs(p, nf, nb, nv, neib_type)) {
								const uint32_t val = (k0 + b) + encv;
								if (neib_type == PT_FLUID) ring.store_f(num, val);
								else if (neib_type == PT_BOUNDARY) ring.store_b(num, val);
								else column[(size_t)neib_list_offset(p, num, neib_type)*p.stride] = (neibdata)val;
								encv = 0u;
							}
						}
					}
				}
			}
		}
	}
}

// BUF: the position array is smaller than 4 GB and is read through a buffer descriptor
// SA: semi-analytical boundaries (vertex section, wider boundary radius, VERTPOS of the segments)
// MC1: -1, or the axis of COORD1 (0 = x, 1 = y) for the prepass on the matrix cores above (never with SA)
template<bool BUF, bool SA, int MC1 = -1>
__global__ void __launch_bounds__(BLOCK_NEIBS, (MC1 >= 0 ? 3 : 1))      // (the prepass's LDS allows three workgroups per CU: registers for as many)
build_neibs_kernel(DevParams p, SaNeibArgs sa, neibdata *__restrict__ neibsList,
	const float4 *__restrict__ posArray, const particleinfo *__restrict__ infoArray,
	const uint32_t *__restrict__ particleHash,
	const uint32_t *__restrict__ cellStart, const uint32_t *__restrict__ cellEnd,
	const uint32_t *__restrict__ cellFluidEnd,
	uint32_t particleRangeEnd, uint32_t posRows, float sqinfluenceradius, NeibsCounters *__restrict__ counters,
	uint32_t *__restrict__ neibCounts /* [out] entries of the fluid section | of the second section << 16, for the tile lists */,
	uint32_t firstParticle /* the launch builds the lists of [firstParticle, particleRangeEnd): a build in parts, sphx_build_neibs_sa */)
	// (the partial counter sets lie behind *counters: NeibsSpread)
{
	__shared__ neibdata sRing[BLOCK_NEIBS/64][NEIB_FRING + NEIB_BRING][64];
	__shared__ uint32_t sMask[MC1 >= 0 ? (BLOCK_NEIBS/64)*3*64*NM_WORDS : 1];      // [wave][row of a slab][lane][word]
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t index = firstParticle + blockIdx.x*BLOCK_NEIBS + threadIdx.x;
	const bool inRange = index < particleRangeEnd;
	uint32_t nf = 0, nb = 0, nv = 0; // neibs_num[PT_FLUID, PT_BOUNDARY, PT_VERTEX]
	neibdata *const column = neibsList + (inRange ? index : 0u);
	NeibRing<NEIB_FRING, NEIB_BRING> ring;
	ring.init(sRing[threadIdx.x >> 6], column, p.stride, p.neibboundpos, lane);

	particleinfo info = make_ushort4(0, 0, 0, 0);
	float4 pos = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
	bool walking = false;
	if (inRange) {
		info = infoArray[index];
		bool build_nl = IS_FLUID(info) || IS_TESTPOINT(info) || IS_FLOATING(info) || HAS_COMPUTE_FORCE(info);
		if (SA) build_nl = build_nl || PART_TYPE(info) == PT_VERTEX || IS_BOUNDARY(info);
		if (p.boundarytype == SPHX_DYN_BOUNDARY) build_nl = true;
		if (build_nl) {
			pos = posArray[index];
			walking = is_active_w(pos.w);
		}
	}
	const int3 gridPos = walking ? grid_pos_from_hash(p, particleHash[index] & CELLTYPE_BITMASK) : make_int3(0, 0, 0);
	const bool boundary = IS_BOUNDARY(info);
	// boundary particles never list boundary neighbours with LJ boundaries, nor with DYN boundaries unless the formulation is
	// SPH_GRENIER, whose sigma sums over them (:588-601); MK_BOUNDARY (uploaded as LJ + mk_mask) has no such rule
	const bool noBB = (p.boundarytype == SPHX_LJ_BOUNDARY && !p.mk_mask) ||
		(p.boundarytype == SPHX_DYN_BOUNDARY && p.formulation != SPHX_SPH_GRENIER);
	const bool fluidOnly = boundary && noBB;

	// sa_boundary_niC_vars (:147-190): in-plane frame of a segment, the ids of its vertices
	uint4 ownVerts = make_uint4(0, 0, 0, 0);
	float3 coord1 = make_float3(0.0f, 0.0f, 0.0f), coord2 = coord1;
	if (SA && walking && boundary) {
		ownVerts = sa.vertices[index];
		const float4 be = sa.boundElements[index];
		const int j = (fabsf(be.z) < fabsf(be.y) && fabsf(be.z) < fabsf(be.x)) ? 2 : (fabsf(be.y) < fabsf(be.x) ? 1 : 0);
		const float cx = -((j == 1)*be.z) + (j == 2)*be.y;
		const float cy = (j == 0)*be.z - ((j == 2)*be.x);
		const float cz = -((j == 0)*be.y) + (j == 1)*be.x;
		const float inv = 1.0f/sqrtf(fmaf(0.0f, 0.0f, fmaf(cz, cz, fmaf(cy, cy, cx*cx))));
		coord1 = make_float3(cx*inv, cy*inv, cz*inv);
		coord2 = make_float3(fmaf(be.y, coord1.z, -(be.z*coord1.y)), fmaf(be.z, coord1.x, -(be.x*coord1.z)),
			fmaf(be.x, coord1.y, -(be.y*coord1.x)));
	}

	const __amdgpu_buffer_rsrc_t posRsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(posArray), 0,
		BUF ? (int)(posRows*16u) : 0, 0x00020000);
	if constexpr (MC1 >= 0 && !SA)
		neibs_mfma_prepass<MC1, NEIB_FRING, NEIB_BRING>(p, posArray, infoArray, particleHash, cellStart, cellEnd, cellFluidEnd,
			sqinfluenceradius, lane, index, pos, gridPos, info, fluidOnly, noBB, walking, nf, nb, nv, ring, column,
			reinterpret_cast<uint32_t (*)[64][NM_WORDS]>(sMask + (threadIdx.x >> 6)*(3*64*NM_WORDS)));
	SPHX_EMU_MARK(300);
	const unsigned long long wmask = __builtin_amdgcn_ballot_w64(walking);
	SPHX_EMU_REFUSE(wmask != 0ull, "build_neibs_kernel: lanes are left to the general walk, whose ballots sit under lane-dependent conditions");
	// first particle, end of the fluid segment and end of neighbour cell `c` of every lane.  The three loads do not depend
	// on each other and are issued one cell ahead of their use, so that a cell costs one memory round trip (its first batch
	// of positions) instead of two
	struct CellMeta { uint32_t start, fluidEnd, end; };
	auto cell_meta = [&](int c) {
		const int x = c % 3 - 1, y = (c/3) % 3 - 1, z = c/9 - 1;
		int gx = gridPos.x, gy = gridPos.y, gz = gridPos.z;
		bool valid = walking;
		valid = valid && neib_cell_axis(gx, x, p.gs[0], p.periodic & SPHX_PERIODIC_X);
		valid = valid && neib_cell_axis(gy, y, p.gs[1], p.periodic & SPHX_PERIODIC_Y);
		valid = valid && neib_cell_axis(gz, z, p.gs[2], p.periodic & SPHX_PERIODIC_Z);
		const uint32_t cellHash = valid ? grid_hash(p, gx, gy, gz) : 0u;
		CellMeta m;
		m.start = cellStart[cellHash];
		m.fluidEnd = cellFluidEnd[cellHash];
		m.end = cellEnd[cellHash];
		if (!valid) m.start = CELL_EMPTY;
		return m;
	};
	CellMeta nextMeta = { CELL_EMPTY, 0u, 0u };
	if (wmask) nextMeta = cell_meta(0);
	if (wmask)
	for (int c = 0; c < 27; ++c) {
		const CellMeta cur = nextMeta;
		nextMeta = cell_meta(min(c + 1, 26));
		const int x = c % 3 - 1, y = (c/3) % 3 - 1, z = c/9 - 1;
		const uint32_t bucketStart = cur.start;
		if (bucketStart != CELL_EMPTY) {
			const uint32_t fluidEnd = cur.fluidEnd;
			const uint32_t bucketEnd = fluidOnly ? fluidEnd : cur.end;
			const uint32_t cell = (uint32_t)c;

			const float px = fmaf(-(float)x, p.cs[0], pos.x);
			const float py = fmaf(-(float)y, p.cs[1], pos.y);
			const float pz = fmaf(-(float)z, p.cs[2], pos.z);

			const uint32_t code = (cell + 1u) << CELLNUM_SHIFT;
			uint32_t encv = code;             // cell code still owed to the first entry stored for this (cell, type) run
			uint32_t neib_type = PT_FLUID;
			// --- fluid segment: type known, NEIB_MLP position gathers in flight ---
			const uint32_t selfrel = (cell == 13u) ? index - bucketStart : 0xFFFFFFFFu;
			for (uint32_t j0 = bucketStart; __builtin_amdgcn_ballot_w64(j0 < fluidEnd); j0 += NEIB_MLP) {
				// the loop is kept wave-uniform (a lane past its segment masks its candidates) so that the ring can be
				// flushed by the whole wave from inside it
				const bool in = j0 < fluidEnd;
				float4 cp[NEIB_MLP];
				if (BUF) {
					// (j0 < 2^28 with a position array below 4 GB: the row offsets j0*16 + 16 u cannot wrap, so they may ride in the
					// instruction's immediate instead of costing a shift and an add per candidate)
					__builtin_assume(j0 < (1u << 28) - 4u);
#pragma unroll
					for (int u = 0; u < NEIB_MLP; ++u) cp[u] = load_pos_row(posRsrc, j0*16u, u);
				} else {
#pragma unroll
					for (int u = 0; u < NEIB_MLP; ++u) cp[u] = posArray[in ? min(j0 + (uint32_t)u, fluidEnd - 1u) : 0u];
				}
				// The scan is instruction-issue bound (PMC: VALU + SALU of this loop), so it is a straight line: every lane
				// writes the candidate at its current slot of the ring and only an accepted one advances the slot (the next
				// candidate overwrites a rejected one); an inactive candidate turns its distance into NaN (0*w) instead of
				// a separate test.  A full list is the one case left to the general code below (one wave-uniform test).
				ring.room_f(NEIB_MLP, 4);
				float r2[NEIB_MLP];
#pragma unroll
				for (int u = 0; u < NEIB_MLP; ++u) {
					const float rx = px - cp[u].x, ry = py - cp[u].y, rz = pz - cp[u].z;
					r2[u] = fmaf(rz, rz, fmaf(ry, ry, rx*rx));
				}
				const uint32_t jrel = j0 - bucketStart;
				const uint32_t rem = in ? fluidEnd - j0 : 0u;
				if (!__builtin_amdgcn_ballot_w64(nf + (uint32_t)NEIB_MLP >= p.neibboundpos)) {
#pragma unroll
					for (int u = 0; u < NEIB_MLP; ++u) {
						const float r2a = fmaf(0.0f, cp[u].w, r2[u]);
						// (selfrel is no index of any other cell, so the self test needs no "is this the home cell" around it: as a
						// select between two per-lane flags that guard cost five vector instructions per candidate)
						const bool acc = (r2a < sqinfluenceradius) && ((uint32_t)u < rem) && (jrel + (uint32_t)u != selfrel);
						ring.fring[nf % (uint32_t)NEIB_FRING][lane] = (neibdata)(jrel + (uint32_t)u + encv);
						encv = acc ? 0u : encv;
						nf += acc ? 1u : 0u;
					}
					ring.sf = nf;
					continue;
				}
#pragma unroll
				for (int u = 0; u < NEIB_MLP; ++u) {
					// (the candidate's index j0 + u is named nowhere: its row offset is an immediate of the load above)
					const bool acc = (r2[u] < sqinfluenceradius) && ((uint32_t)u < rem) && (jrel + (uint32_t)u != selfrel) &&
						is_active_w(cp[u].w);
					nf += acc ? 1u : 0u;
					const bool ok = acc && !too_many_neibs(p, nf, nb, nv, PT_FLUID);
					ring.fring[ring.sf % (uint32_t)NEIB_FRING][lane] = (neibdata)((jrel + (uint32_t)u) + encv);
					ring.sf += ok ? 1u : 0u;
					encv = ok ? 0u : encv;
				}
			}
			ring.room_b(NEIB_BRING/2, 0);
			// --- non-fluid tail (boundary / vertex / testpoint candidates): the reference's loop as is ---
			for (uint32_t neib_index = max(fluidEnd, bucketStart); neib_index < bucketEnd; ++neib_index) {
				if (neib_index == index) continue;
				const particleinfo neib_info = infoArray[neib_index];
				if (IS_TESTPOINT(neib_info)) continue;
				if (neib_type != PART_TYPE(neib_info))
					encv = code;
				neib_type = PART_TYPE(neib_info);
				if (noBB && boundary && IS_BOUNDARY(neib_info))
					continue;
				const float4 neib_pos = posArray[neib_index];
				if (!is_active_w(neib_pos.w)) continue;
				const float rx = px - neib_pos.x, ry = py - neib_pos.y, rz = pz - neib_pos.z;
				const float r2 = fmaf(rz, rz, fmaf(ry, ry, rx*rx));
				// isCloseEnough (:389-408): boundary neighbours a little beyond the radius are kept with SA boundaries
				const bool close_enough = (r2 < sqinfluenceradius) || (SA && r2 < sa.boundNlSqInflRad && IS_BOUNDARY(neib_info));
				if (SA && boundary) {      // process_niC_segment (:433-463)
					const uint32_t nid = info_id(neib_info);
					const int k = (nid == ownVerts.x) ? 0 : (nid == ownVerts.y) ? 1 : (nid == ownVerts.z) ? 2 : -1;
					if (k >= 0)
						sa.vertPos[k][index] = make_float2(fmaf(rz, coord1.z, fmaf(ry, coord1.y, rx*coord1.x)),
							fmaf(rz, coord2.z, fmaf(ry, coord2.y, rx*coord2.x)));
				}
				if (close_enough) {
					const uint32_t num = (neib_type == PT_FLUID) ? nf : (neib_type == PT_BOUNDARY) ? nb : nv;
					if (neib_type == PT_FLUID) nf++; else if (neib_type == PT_BOUNDARY) nb++; else nv++;
					if (!too_many_neibs(p, nf, nb, nv, neib_type)) {
						const uint32_t val = (neib_index - bucketStart) + encv;
						if (neib_type == PT_FLUID) ring.store_f(num, val);
						else if (neib_type == PT_BOUNDARY) ring.store_b(num, val);
						else column[(size_t)neib_list_offset(p, num, neib_type)*p.stride] = (neibdata)val;
						encv = 0u;
					}
				}
			}
		}
	}

	// terminators (every particle below particleRangeEnd gets them, walking or not), then what is left in the rings
	if (inRange) {
		bool overflow = too_many_neibs(p, nf, nb, nv, PT_FLUID);
		if (overflow) column[(size_t)p.neibboundpos*p.stride] = NEIBS_END;
		else ring.store_f(nf, NEIBS_END);
		overflow |= too_many_neibs(p, nf, nb, nv, PT_BOUNDARY);
		if (!overflow) ring.store_b(nb, NEIBS_END);
		if (SA) {
			overflow |= too_many_neibs(p, nf, nb, nv, PT_VERTEX);
			const uint32_t marker_pos = overflow ? p.neiblistsize - 1u : p.neibboundpos + 1u + nv;
			column[(size_t)marker_pos*p.stride] = NEIBS_END;
		}
		if (overflow) {
			const int pid = (int)info_id(info);
			if (atomicCAS(&counters->hasTooManyNeibs, -1, pid) == -1) {
				counters->hasMaxNeibs[0] = nf; counters->hasMaxNeibs[1] = nb; counters->hasMaxNeibs[2] = nv;
			}
		}
	}
	SPHX_EMU_MARK(400);
	ring.finish();
	// the section lengths of this list, for the builder of the tile lists (forces.hip): it sizes the rows of a chunk of 64
	// particles from them before it reads a single entry (second section: boundary particles, or the vertices with SA_BOUNDARY)
	if (inRange) neibCounts[index] = min(nf, 0xFFFFu) | (min(SA ? nv : nb, 0xFFFFu) << 16);

	// neibcount: per-block max / total, one atomic pair per wave
	uint32_t total = nf + nb + nv;
	uint32_t mx = nf + nb, mv = nv;
#pragma unroll
	for (int d = 32; d > 0; d >>= 1) {
		total += __shfl_down(total, d);
		mx = max(mx, (uint32_t)__shfl_down(mx, d));
		if (SA) mv = max(mv, (uint32_t)__shfl_down(mv, d));
	}
	if ((threadIdx.x & 63u) == 0 && total) {
		NeibsSpread *part = reinterpret_cast<NeibsSpread*>(counters + 1) + (blockIdx.x & (NEIBS_SPREAD - 1u));
		atomicMax(&part->maxFluidBoundaryNeibs, (int)mx);
		if (SA) atomicMax(&part->maxVertexNeibs, (int)mv);
		atomicAdd(&part->numInteractions, (unsigned long long)total);
	}
}


// the launch behind sphx_build_neibs_sa (neibs.hip): the list of [firstParticle, particleRangeEnd), the counters (added to what
// earlier parts of the same build left), the section lengths
int sphx_neibs_list_launch_part(sphx_ctx *ctx, uint16_t *neibsList, const void *pos, const void *info, const uint32_t *hash,
	const uint32_t *cellStart, const uint32_t *cellEnd, const void *vertices, const void *boundElements,
	void *vertPos0, void *vertPos1, void *vertPos2, uint32_t numParticles, uint32_t firstParticle, uint32_t particleRangeEnd,
	float sqinfluenceradius, float boundNlSqInflRad, hipStream_t st)
{
	if (firstParticle >= particleRangeEnd) return SPHX_OK;
	const bool sa = ctx->params.boundarytype == SPHX_SA_BOUNDARY;
	const bool posBuf = (size_t)numParticles*16u < ((size_t)1 << 32);
	SaNeibArgs saArgs;
	saArgs.vertices = (const uint4*)vertices; saArgs.boundElements = (const float4*)boundElements;
	saArgs.vertPos[0] = (float2*)vertPos0; saArgs.vertPos[1] = (float2*)vertPos1; saArgs.vertPos[2] = (float2*)vertPos2;
	saArgs.boundNlSqInflRad = boundNlSqInflRad;
	// the prepass on the matrix cores (COORD1 = x or y, plain boundaries, positions behind a buffer descriptor): an experiment that
	// is bit-identical and measured slower than the general walk (see the account above neibs_mfma_prepass): SPHX_NEIBS_MFMA=1 runs it
	// (read when the context was created, like SPHX_DISABLE_TILES)
	const int mc1 = (!sa && posBuf && ctx->neibs_mfma && ctx->dev.c1 <= 1) ? ctx->dev.c1 : -1;
#define SPHX_NB_LAUNCH(K) SPHX_LAUNCH_WAVES(K, div_up_u(particleRangeEnd - firstParticle, BLOCK_NEIBS), BLOCK_NEIBS, st, ctx->dev, \
		saArgs, neibsList, (const float4*)pos, (const particleinfo*)info, hash, cellStart, cellEnd, ctx->cell_fluid_end, \
		particleRangeEnd, numParticles, sqinfluenceradius, ctx->counters_dev, ctx->neib_counts, firstParticle)
	if (sa) { if (posBuf) SPHX_NB_LAUNCH((build_neibs_kernel<true, true>)); else SPHX_NB_LAUNCH((build_neibs_kernel<false, true>)); }
	else if (mc1 == 0) SPHX_NB_LAUNCH((build_neibs_kernel<true, false, 0>));
	else if (mc1 == 1) SPHX_NB_LAUNCH((build_neibs_kernel<true, false, 1>));
	else if (posBuf) SPHX_NB_LAUNCH((build_neibs_kernel<true, false>));
	else SPHX_NB_LAUNCH((build_neibs_kernel<false, false>));
#undef SPHX_NB_LAUNCH
	SPHX_LAUNCH_CHECK("build_neibs_kernel");
	return SPHX_OK;
}

int sphx_neibs_list_launch(sphx_ctx *ctx, uint16_t *neibsList, const void *pos, const void *info, const uint32_t *hash,
	const uint32_t *cellStart, const uint32_t *cellEnd, const void *vertices, const void *boundElements,
	void *vertPos0, void *vertPos1, void *vertPos2, uint32_t numParticles, uint32_t particleRangeEnd,
	float sqinfluenceradius, float boundNlSqInflRad, hipStream_t st)
{
	return sphx_neibs_list_launch_part(ctx, neibsList, pos, info, hash, cellStart, cellEnd, vertices, boundElements, vertPos0, vertPos1, vertPos2,
		numParticles, 0u, particleRangeEnd, sqinfluenceradius, boundNlSqInflRad, st);
}
