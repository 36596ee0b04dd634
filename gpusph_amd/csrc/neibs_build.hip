// neibs_build.hip -- the cell-linked neighbour list of the neighbour engine for gfx950 (buildNeibsListDevice + neibsInCell,
// src/cuda/buildneibs_kernel.cu:536-644,1019-1185).  Split from neibs.hip in round 6; results bit-identical to the reference
// algorithm restated in oracle/sph_oracle.c.  Compiled with -ffp-contract=off like neibs.hip.
#include "sphx_internal.h"
#include <cstring>

#define BLOCK_NEIBS   256

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
	return make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w));
}

// SA_BOUNDARY members of buildneibs_params (src/cuda/buildneibs_params.h:66-115)
struct SaNeibArgs {
	const uint4 *vertices;          // vertexinfo of the segments
	const float4 *boundElements;    // normal + area of the segments
	float2 *vertPos[3];             // [out] in-plane offsets of a segment's three vertices
	float boundNlSqInflRad;         // search radius for boundary neighbours
};

// BUF: the position array is smaller than 4 GB and is read through a buffer descriptor
// SA: semi-analytical boundaries (vertex section, wider boundary radius, VERTPOS of the segments)
template<bool BUF, bool SA>
__global__ void __launch_bounds__(BLOCK_NEIBS)
build_neibs_kernel(DevParams p, SaNeibArgs sa, neibdata *__restrict__ neibsList,
	const float4 *__restrict__ posArray, const particleinfo *__restrict__ infoArray,
	const uint32_t *__restrict__ particleHash,
	const uint32_t *__restrict__ cellStart, const uint32_t *__restrict__ cellEnd,
	const uint32_t *__restrict__ cellFluidEnd,
	uint32_t particleRangeEnd, uint32_t posRows, float sqinfluenceradius, NeibsCounters *__restrict__ counters,
	uint32_t *__restrict__ neibCounts /* [out] entries of the fluid section | of the second section << 16, for the tile lists */)
	// (the partial counter sets lie behind *counters: NeibsSpread)
{
	__shared__ neibdata sRing[BLOCK_NEIBS/64][NEIB_FRING + NEIB_BRING][64];
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t index = blockIdx.x*BLOCK_NEIBS + threadIdx.x;
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
	const unsigned long long wmask = __builtin_amdgcn_ballot_w64(walking);
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


// the launch behind sphx_build_neibs_sa (neibs.hip): the list of [0, particleRangeEnd), the counters, the section lengths
int sphx_neibs_list_launch(sphx_ctx *ctx, uint16_t *neibsList, const void *pos, const void *info, const uint32_t *hash,
	const uint32_t *cellStart, const uint32_t *cellEnd, const void *vertices, const void *boundElements,
	void *vertPos0, void *vertPos1, void *vertPos2, uint32_t numParticles, uint32_t particleRangeEnd,
	float sqinfluenceradius, float boundNlSqInflRad, hipStream_t st)
{
	const bool sa = ctx->params.boundarytype == SPHX_SA_BOUNDARY;
	const bool posBuf = (size_t)numParticles*16u < ((size_t)1 << 32);
	SaNeibArgs saArgs;
	saArgs.vertices = (const uint4*)vertices; saArgs.boundElements = (const float4*)boundElements;
	saArgs.vertPos[0] = (float2*)vertPos0; saArgs.vertPos[1] = (float2*)vertPos1; saArgs.vertPos[2] = (float2*)vertPos2;
	saArgs.boundNlSqInflRad = boundNlSqInflRad;
	(sa ? (posBuf ? build_neibs_kernel<true, true> : build_neibs_kernel<false, true>)
	    : (posBuf ? build_neibs_kernel<true, false> : build_neibs_kernel<false, false>))<<<div_up_u(particleRangeEnd, BLOCK_NEIBS), BLOCK_NEIBS, 0, st>>>(ctx->dev,
		saArgs, neibsList, (const float4*)pos, (const particleinfo*)info, hash, cellStart, cellEnd, ctx->cell_fluid_end,
		particleRangeEnd, numParticles, sqinfluenceradius, ctx->counters_dev, ctx->neib_counts);
	SPHX_LAUNCH_CHECK("build_neibs_kernel");
	return SPHX_OK;
}
