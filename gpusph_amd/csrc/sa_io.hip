// sa_io.hip -- open boundaries of the semi-analytical wall model (SA_BOUNDARY + ENABLE_INLET_OUTLET): the passes over
// the ELEMENTS of the open faces -- their vertices, their segments and the fluid particles about to leave through them -- for gfx950.
//
// What these passes do, and which entry points of CUDABoundaryConditionsEngine / the forces engine they stand in for:
//   sphx_sa_identify_corner_vertices    saIdentifyCornerVertices   src/cuda/boundary_conditions.cu:667   (_kernel.cu:2319-2362)
//   sphx_sa_init_io_mass_vertex_count   initIOmass_vertexCount     src/cuda/boundary_conditions.cu:578   (_kernel.cu:1999-2064)
//   sphx_sa_init_io_mass                initIOmass                 src/cuda/boundary_conditions.cu:610   (_kernel.cu:2078-2172)
//   sphx_sa_find_outgoing_segment       findOutgoingSegment        src/cuda/boundary_conditions.cu:238   (_kernel.cu:1647-1750)
//   sphx_sa_disable_outgoing_parts      disableOutgoingParts       src/cuda/boundary_conditions.cu:76    (_kernel.cu:2374-2398)
//   sphx_sa_segment_bc_io               saSegmentBoundaryConditions with open boundaries                  (_kernel.cu:1427-1520)
//   sphx_sa_vertex_bc_io                saVertexBoundaryConditions with open boundaries                   (_kernel.cu:2197-2252)
//   sphx_sa_io_water_depth              ENABLE_WATER_DEPTH in the vertex forces pass    src/cuda/forces_kernel.def:192-205,3285-3303
//   sphx_flux_computation               FLUX_COMPUTATION           src/cuda/post_process.cu:485-570
// The passes of such a run over ALL fluid particles (density summation, forces, density diffusion) are the list walkers of
// sa_bounds.hip with their open-boundary terms.
//
// Design.  A run has a few thousand open elements among millions of particles, each with ~100 neighbours and two powf per
// neighbour.  Every pass here therefore runs in two launches: (1) one coalesced sweep over particleinfo that appends the rows the
// pass is about to an index (wave-aggregated: one atomic per wave), (2) a grid of waves that take those rows one WAVE per row:
// lane l decodes and evaluates entry l of the list section in question (wave_list.h), the per-row result is a reduction over the
// lanes and a short scalar epilogue.  The solid-wall rows of the two boundary-condition passes go to the kernels of
// sa_bounds.hip, which skip the open rows.  The physics is written from the equations (characteristic boundary condition of a
// weakly compressible flow; area coordinates of a point in a triangle); results are held to the CPU oracle bit for bit where a
// threshold decides what happens next (see wave_list.h on the order of sums).
#include "sphx_internal.h"
#include "neib_iter.h"
#include "sa_args.h"
#include "wave_list.h"
#include "sa_wall_gamma.h"      // V3 and the corners of a boundary element from BUFFER_VERTPOS (wall_tri_setup)

// tests/hostemu runs this file's SOURCE on the host (a test harness: the library has no CPU path; hipcc sees the definitions
// below): row kernels as 64 lock-stepped fibres per wave, the element-wise ones thread after thread
#ifndef SPHX_LAUNCH
#define SPHX_LAUNCH(kernel, grid, block, stream, ...) kernel<<<(grid), (block), 0, (stream)>>>(__VA_ARGS__)
#define SPHX_LAUNCH_WAVES(kernel, grid, block, stream, ...) kernel<<<(grid), (block), 0, (stream)>>>(__VA_ARGS__)
#endif

#define OPEN_MAX_RING 30          // other vertices of the open segments around a vertex that are remembered (_kernel.cu:1977)
#define ROW_THREADS 256           // four rows per workgroup
#ifndef ROW_GRID
#define ROW_GRID 1024             // waves stride over the index: 4096 waves cover any open face in a few rounds
#endif

// ---- (1) the index of a pass ---------------------------------------------------------------------------------------------------
enum RowKind {
	ROWS_OPEN_VERTICES,            // every vertex of an open face
	ROWS_OPEN_INNER_VERTICES,      // ... that is not a corner (shared with a wall or another face)
	ROWS_OPEN_SEGMENTS,
	ROWS_DEPTH_GAUGES,             // active vertices of pressure-driven faces
	ROWS_LEAVING_CANDIDATES        // active, unmarked fluid particles with boundary elements in reach
};

struct RowSweep {
	const particleinfo *info;
	const float4 *pos;             // masses (activity): DEPTH_GAUGES, LEAVING_CANDIDATES
	const uint4 *marks;            // BUFFER_VERTICES of the fluid rows: LEAVING_CANDIDATES
	const neibdata *list;          // LEAVING_CANDIDATES
	uint32_t first, end, stride, neibboundpos;
	uint32_t *rows;                // [0] = how many, [1..] = which
};

template<int KIND>
__global__ void __launch_bounds__(256)
open_rows_kernel(RowSweep a)
{
	const uint32_t i = a.first + blockIdx.x*256 + threadIdx.x;
	bool take = false;
	if (i < a.end) {
		const particleinfo f = a.info[i];
		if (KIND == ROWS_OPEN_VERTICES) take = PART_TYPE(f) == PT_VERTEX && SA_IS_OPEN(f);
		if (KIND == ROWS_OPEN_INNER_VERTICES) take = PART_TYPE(f) == PT_VERTEX && SA_IS_OPEN(f) && !SA_IS_CORNER(f);
		if (KIND == ROWS_OPEN_SEGMENTS) take = PART_TYPE(f) == PT_BOUNDARY && SA_IS_OPEN(f);
		if (KIND == ROWS_DEPTH_GAUGES) take = PART_TYPE(f) == PT_VERTEX && SA_IS_OPEN(f) && !SA_IS_VELOCITY_DRIVEN(f) && is_active_w(a.pos[i].w);
		if (KIND == ROWS_LEAVING_CANDIDATES) {
			take = PART_TYPE(f) == PT_FLUID && is_active_w(a.pos[i].w) && a.list[(size_t)a.neibboundpos*a.stride + i] != NEIBS_END;
			if (take) { const uint4 m = a.marks[i]; take = (m.x | m.y) == 0u; }      // a marked one stays as it is (:1679-1686)
		}
	}
	const unsigned long long m = wave_ballot(take);
	if (!m) return;
	const uint32_t lane = threadIdx.x & 63u;
	const int leader = __builtin_ctzll(m);
	uint32_t base = 0;
	if ((int)lane == leader) base = atomicAdd(a.rows, (uint32_t)__builtin_popcountll(m));
	base = (uint32_t)__shfl((int)base, leader);
	if (take) a.rows[1u + base + wave_lanes_below(m, lane)] = i;
}

// the rows of a launch, wave after wave
#define FOR_MY_ROWS(rows, index) \
	const uint32_t lane = threadIdx.x & 63u; \
	const uint32_t waves_ = gridDim.x*(ROW_THREADS/64), count_ = (rows)[0]; \
	for (uint32_t w_ = blockIdx.x*(ROW_THREADS/64) + (threadIdx.x >> 6), index; \
	     w_ < count_ && ((index = __builtin_amdgcn_readfirstlane((rows)[1u + w_])), true); w_ += waves_)

// ---- the list of a row, as the row kernels see it ---------------------------------------------------------------------------------
struct RowLists {
	const float4 *pos;
	const uint32_t *hash, *cellStart;
	const neibdata *list;
};

__device__ __forceinline__ bool tri_has(const uint4 &t, uint32_t id) { return t.x == id || t.y == id || t.z == id; }

// ---- physics: the equation of state and the characteristics of the open-boundary condition ------------------------------------------
// Tait: P = B((rho/rho0)^gamma - 1), c = c0 (rho/rho0)^((gamma-1)/2); densities are relative (rho/rho0 - 1).  The Riemann
// invariants of the 1-D problem along the face normal are u_n +- psi(rho) with psi = 2 c /(gamma - 1).  (phys_core.cu:106-127)
struct Tait {
	float B, gamma, c0, rho0, halfGm1;
	__device__ __forceinline__ Tait(const DevParams &p, uint32_t fl) : B(p.bcoeff[fl]), gamma(p.gammacoeff[fl]), c0(p.sscoeff[fl]), rho0(p.rho0[fl]),
		halfGm1(p.sspowercoeff[fl]) {}
	__device__ __forceinline__ float pressure(float rel) const { return B*(powf(rel + 1.0f, gamma) - 1.0f); }
	__device__ __forceinline__ float density_of(float pres) const { return powf(pres/B + 1.0f, 1.0f/gamma) - 1.0f; }
	__device__ __forceinline__ float celerity(float rel) const { return c0*powf(rel + 1.0f, halfGm1); }
	__device__ __forceinline__ float psi(float rel) const { return 2.0f/(gamma - 1.0f)*c0*powf(rel + 1.0f, 0.5f*gamma - 0.5f); }
	__device__ __forceinline__ float psi_inverse(float r) const
	{ return (float)((double)powf((float)(((double)gamma - 1.)*(double)r/(2.*(double)c0)), (float)(2./((double)gamma - 1.))) - 1.0); }
	__device__ __forceinline__ float absolute(float rel) const { return (rel + 1.0f)*rho0; }
};

// The state just inside the face (density `inside`, velocity u with normal part uIn) meets what the problem imposes outside.
//   velocity-driven face: the normal velocity uOut is imposed, the density follows.  If the outside runs away from the interior
//     (uOut <= uIn) an expansion fan connects the states and psi + u_n is conserved across it; otherwise a shock does, its strength
//     from the jump conditions, unless the shock's characteristic is slower than the interior's (then nothing has arrived yet).
//   pressure-driven face: the density `outside` is imposed, the normal velocity follows from the same two waves, tried in the
//     order the density jump suggests and each kept only if its own characteristic says it is the wave that forms; the tangential
//     velocity of the interior is kept on outflow and dropped on inflow.
// (calculateIOboundaryCondition, _kernel.cu:111-200)
__device__ __forceinline__ float4 open_face_state(const Tait &eos, bool velocityDriven, float4 imposed, float inside, float outside,
	V3 u, float uIn, float uOut, V3 n)
{
	const float psiIn = eos.psi(inside);
	if (velocityDriven) {
		float r;
		if (uOut <= uIn)
			r = psiIn + (uOut - uIn);
		else {
			const float behindShock = eos.density_of(eos.pressure(inside) + eos.absolute(inside)*uIn*(uIn - uOut));
			r = eos.psi(behindShock);
			if (uOut + eos.celerity(behindShock) <= uIn + eos.celerity(inside)) r = psiIn;
		}
		imposed.w = eos.psi_inverse(r);
		return imposed;
	}
	const float cOut = eos.celerity(outside);
	const float fastestIn = uIn + eos.celerity(inside);
	const float fan = uIn + (eos.psi(outside) - psiIn);
	float shock = (eos.pressure(inside) - eos.pressure(outside))/(eos.absolute(inside)*fmaxf(uIn, 1e-5f*eos.c0)) + uIn;
	if (fabsf(shock) > eos.c0*0.1f) shock = uIn;
	float un;
	if (outside <= inside) {      // the fan first
		un = fan;
		if (un + cOut > fastestIn) { un = shock; if (un + cOut <= fastestIn) un = uIn; }
	} else {                      // the shock first
		un = shock;
		if (un + cOut <= fastestIn) { un = fan; if (un + cOut > fastestIn) un = uIn; }
	}
	if (outside < 0.0f) un = fminf(un, 0.0f);
	float4 s = make_float4(0.0f, 0.0f, 0.0f, outside);
	if (un < 0.0f) {
		const float along = u.x*n.x + u.y*n.y + u.z*n.z;
		s.x = u.x - along*n.x; s.y = u.y - along*n.y; s.z = u.z - along*n.z;
	}
	s.x += n.x*un; s.y += n.y*un; s.z += n.z*un;
	return s;
}

// ---- geometry: how a point's mass is shared among the three vertices of a segment ----------------------------------------------
// q[k]: the point seen from vertex k.  The shares are the area coordinates of the point's projection onto the segment's plane:
// share k = the (signed) area of the triangle the projection forms with the edge opposite vertex k, over the segment's area.
// A projection outside the segment makes one or two of them negative; it is then moved to the nearest point of the outline: with
// two negative the remaining vertex takes all, with one negative the projection slides along the line towards that vertex until
// it meets the edge.  Areas in double where the reference's 0.5*dot() promotes them.  (getMassRepartitionFactor, :213-283)
__device__ __forceinline__ float half_cross_along(V3 a, V3 b, V3 n) { return (float)(0.5*(double)dot(cross(a, b), n)); }
__device__ __forceinline__ void vertex_shares(const V3 q[3], V3 n, float share[3])
{
	// edge[k] is opposite vertex k and runs from vertex k+1 to vertex k+2; flat[k]: q[k] without its normal part
	const V3 edge[3] = { q[2] - q[1], q[0] - q[2], -(q[0] - q[1]) };
	V3 flat[3];
#pragma unroll
	for (int k = 0; k < 3; ++k) flat[k] = q[k] - n*dot(q[k], n);
	const float whole = half_cross_along(q[0] - q[1], q[0] - q[2], n);
	float area[3];
#pragma unroll
	for (int k = 0; k < 3; ++k) area[k] = half_cross_along(flat[(k + 2)%3], edge[k], n);
	const bool out0 = area[0] < 0.0f, out1 = area[1] < 0.0f, out2 = area[2] < 0.0f;
	if ((int)out0 + (int)out1 + (int)out2 >= 2) {
		const int keeper = out0 ? (out2 ? 1 : 2) : 0;
#pragma unroll
		for (int k = 0; k < 3; ++k) area[k] = (k == keeper) ? whole : 0.0f;
	} else if (out0 | out1 | out2) {
		const int k = out0 ? 0 : (out1 ? 1 : 2), k1 = (k + 1)%3, k2 = (k + 2)%3;
		const float back = (float)((double)area[k]/(0.5*(double)dot(cross(flat[k], edge[k]), n)));
		flat[k1] = flat[k1] - flat[k]*back;
		flat[k] = flat[k]*(float)(1.0 - (double)back);
		area[k] = 0.0f;
		area[k1] = half_cross_along(flat[k], edge[k1], n);
		area[k2] = half_cross_along(flat[k1], edge[k2], n);
	}
#pragma unroll
	for (int k = 0; k < 3; ++k) share[k] = area[k]/whole;
}

// the vertices of segment j seen from its centre (scale: +1 towards them, -1 away), from BUFFER_VERTPOS
struct SegmentCorners { const float2 *c0, *c1, *c2; };
__device__ __forceinline__ void corners_of(const SegmentCorners &vp, uint32_t j, V3 n, float scale, V3 corner[3])
{
	V3 u, v;
	wall_plane_basis(n, u, v);
	const float2 c[3] = { vp.c0[j], vp.c1[j], vp.c2[j] };
	const float inv = 1.0f/scale;
#pragma unroll
	for (int k = 0; k < 3; ++k) corner[k] = (-(u*c[k].x + v*c[k].y))*inv;
}

// ---- corner vertices: an open vertex that also belongs to a segment of something else ----------------------------------------------
__global__ void __launch_bounds__(ROW_THREADS)
open_corner_rows_kernel(DevParams p, RowLists a, const uint32_t *rows, const uint4 *triangles, const particleinfo *info, particleinfo *infoOut)
{
	FOR_MY_ROWS(rows, index) {
		particleinfo mine = info[index];
		const uint32_t face = OBJECT_NUM(mine), id = info_id(mine);
		const float4 own = a.pos[index];
		const int3 cell = grid_pos_from_hash(p, a.hash[index] & CELLTYPE_BITMASK);
		unsigned long long foreign = 0ull;
		int carry = 0; bool more = true;
		for (int s0 = 0; more; s0 += 64) {
			const WaveEntry e = wave_entries<WAVE_SECTION_BOUNDARY>(p, a.list, a.cellStart, index, own, cell, s0, lane, carry, more);
			bool hit = false;
			if (e.live) {
				const particleinfo f = info[e.j];
				hit = !(OBJECT_NUM(f) == face && SA_IS_OPEN(f)) && tri_has(triangles[e.j], id);
			}
			foreign |= wave_ballot(hit);
		}
		if (foreign && lane == 0u) { mine.x |= SA_FG_CORNER; infoOut[index] = mine; }
	}
}

// ---- the ring of an open vertex: the other vertices of the open segments it belongs to, in list order, each as often as it occurs ----
// (one LDS row of OPEN_MAX_RING ids per wave; returns how many were kept)
__device__ __forceinline__ uint32_t vertex_ring(const DevParams &p, const RowLists &a, const uint4 *triangles, const particleinfo *info,
	uint32_t index, const float4 &own, const int3 &cell, uint32_t id, uint32_t lane, uint32_t *ring)
{
	uint32_t kept = 0;
	int carry = 0; bool more = true;
	for (int s0 = 0; more; s0 += 64) {
		const WaveEntry e = wave_entries<WAVE_SECTION_BOUNDARY>(p, a.list, a.cellStart, index, own, cell, s0, lane, carry, more);
		uint32_t other[3], n = 0;
		if (e.live && SA_IS_OPEN(info[e.j])) {
			const uint4 t = triangles[e.j];
			if (tri_has(t, id)) {
				if (t.x != id) other[n++] = t.x;
				if (t.y != id) other[n++] = t.y;
				if (t.z != id) other[n++] = t.z;
			}
		}
		// exclusive prefix of n (0..3) over the lanes
		const unsigned long long ge1 = wave_ballot(n >= 1u), ge2 = wave_ballot(n >= 2u), ge3 = wave_ballot(n >= 3u);
		const uint32_t at = kept + wave_lanes_below(ge1, lane) + wave_lanes_below(ge2, lane) + wave_lanes_below(ge3, lane);
		for (uint32_t k = 0; k < n; ++k)
			if (at + k < OPEN_MAX_RING) ring[at + k] = other[k];
		kept += (uint32_t)(__builtin_popcountll(ge1) + __builtin_popcountll(ge2) + __builtin_popcountll(ge3));
	}
	__builtin_amdgcn_wave_barrier();
	return kept < OPEN_MAX_RING ? kept : OPEN_MAX_RING;
}

struct OpenMassArgs {
	const uint4 *triangles;
	const particleinfo *info;
	float4 *forces;            // .w: the count (written by the first pass, read by the second)
	float4 *newPos;
	float deltap;
};

// first pass: how many non-corner ring vertices an open vertex has among its vertex neighbours (with multiplicity)
// second pass: every open vertex is levelled towards half a fluid particle's mass; odd ids take what they lack from their even ring
// neighbours in equal parts, even ids give what their odd ring neighbours lack (by those neighbours' counts)
template<bool SECOND>
__global__ void __launch_bounds__(ROW_THREADS)
open_mass_rows_kernel(DevParams p, RowLists a, const uint32_t *rows, OpenMassArgs o)
{
	__shared__ uint32_t sRing[ROW_THREADS/64][OPEN_MAX_RING + 2];
	uint32_t *ring = sRing[threadIdx.x >> 6];
	FOR_MY_ROWS(rows, index) {
		const particleinfo mine = o.info[index];
		const uint32_t id = info_id(mine);
		const float4 own = SECOND ? a.pos[index] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
		const int3 cell = grid_pos_from_hash(p, a.hash[index] & CELLTYPE_BITMASK);
		const uint32_t ringSize = vertex_ring(p, a, o.triangles, o.info, index, own, cell, id, lane, ring);
		const bool taker = (id & 1u) != 0u;
		const float target = 0.5f*o.deltap*o.deltap*o.deltap*p.rho0[FLUID_NUM(mine)];
		const float lacking = target - own.w;
		const float myCount = SECOND ? o.forces[index].w : 0.0f;
		uint32_t total = 0;
		float change = 0.0f;
		int carry = 0; bool more = true;
		for (int s0 = 0; more; s0 += 64) {
			const WaveEntry e = wave_entries<WAVE_SECTION_VERTEX>(p, a.list, a.cellStart, index, own, cell, s0, lane, carry, more);
			uint32_t times = 0;
			float part = 0.0f;
			if (e.live) {
				const particleinfo f = o.info[e.j];
				const uint32_t nid = info_id(f);
				for (uint32_t k = 0; k < ringSize; ++k) times += (ring[k] == nid) ? 1u : 0u;
				if (SA_IS_CORNER(f)) times = 0;
				if (SECOND) {
					if (((nid & 1u) != 0u) == taker) times = 0;
					if (taker) { part = lacking/myCount; if (!(lacking > 0.0f)) times = 0; }
					else {
						const float theirs = target - a.pos[e.j].w;
						part = -(theirs/o.forces[e.j].w);
						if (!(theirs > 0.0f)) times = 0;
					}
				}
			}
			if (!SECOND) total += times;
			else {
				unsigned long long any = wave_ballot(times != 0u);
				while (any) {
					const int l = __builtin_ctzll(any);
					any &= any - 1ull;
					const uint32_t c = wave_lane_u(times, l);
					const float t = wave_lane_f(part, l);
					for (uint32_t k = 0; k < c; ++k) change += t;
				}
			}
		}
		__builtin_amdgcn_wave_barrier();      // the ring row is rewritten by the next row
		if (!SECOND) { total = wave_sum_u(total); if (lane == 0u) o.forces[index].w = (float)total; }
		else if (lane == 0u) o.newPos[index].w = own.w + change;
	}
}

// ---- fluid particles about to leave: the nearest open segment they are behind and moving out of ------------------------------------
struct LeavingArgs {
	const float4 *vel, *boundElement;
	SegmentCorners corners;
	const particleinfo *info;
	uint4 *marks;              // BUFFER_VERTICES: a leaving particle's row takes the three vertex ids of its segment
	float4 *shares;            // BUFFER_GRADGAMMA: ... and its row here the three shares and the particle's mass
	float reach;
	// the rows are the context's list of fluid particles with boundary elements in reach (built with the neighbour list) instead of a
	// sweep of this pass: what the sweep would have filtered -- the range, activity, a mark from an earlier pass -- is checked here
	int filter;
	uint32_t end;
};

__global__ void __launch_bounds__(ROW_THREADS)
open_leaving_rows_kernel(DevParams p, RowLists a, const uint32_t *rows, LeavingArgs o)
{
	FOR_MY_ROWS(rows, index) {
		if (o.filter && index >= o.end) continue;
		const float4 own = a.pos[index];
		if (o.filter) {
			if (!is_active_w(own.w)) continue;
			const uint4 m = o.marks[index];
			if ((m.x | m.y) != 0u) continue;      // a marked one stays as it is (:1679-1686)
		}
		const float4 u = o.vel[index];
		const int3 cell = grid_pos_from_hash(p, a.hash[index] & CELLTYPE_BITMASK);
		float best = o.reach*o.reach;
		uint32_t bestSeg = 0xFFFFFFFFu;
		V3 bestN = v3(0.0f, 0.0f, 0.0f), bestRel = bestN;
		int carry = 0; bool more = true;
		for (int s0 = 0; more; s0 += 64) {
			const WaveEntry e = wave_entries<WAVE_SECTION_BOUNDARY>(p, a.list, a.cellStart, index, own, cell, s0, lane, carry, more);
			float d2 = __int_as_float(0x7f800000);
			V3 n = v3(0.0f, 0.0f, 0.0f), rel = n;
			if (e.live && SA_IS_OPEN(o.info[e.j])) {
				const float4 np = a.pos[e.j], be = o.boundElement[e.j], nu = o.vel[e.j];
				rel = v3(e.ox - np.x, e.oy - np.y, e.oz - np.z);
				n = v3(be.x, be.y, be.z);
				const V3 du = v3(u.x - nu.x, u.y - nu.y, u.z - nu.z);
				// behind the element and moving out relative to it
				if (dot(n, rel) <= 0.0f && dot(n, du) < 0.0f) d2 = dot(rel, rel);
			}
			const float least = wave_min_f(d2);
			if (least < best) {      // strictly closer than anything before; among equals the earliest entry
				const int l = __builtin_ctzll(wave_ballot(d2 == least));
				best = least;
				bestSeg = wave_lane_u(e.j, l);
				bestN = v3(wave_lane_f(n.x, l), wave_lane_f(n.y, l), wave_lane_f(n.z, l));
				bestRel = v3(wave_lane_f(rel.x, l), wave_lane_f(rel.y, l), wave_lane_f(rel.z, l));
			}
		}
		if (bestSeg == 0xFFFFFFFFu) continue;
		V3 corner[3], q[3];
		corners_of(o.corners, bestSeg, bestN, 1.0f, corner);
		for (int k = 0; k < 3; ++k) q[k] = bestRel - corner[k];
		float share[3];
		vertex_shares(q, bestN, share);
		if (lane == 0u) {
			o.marks[index] = o.marks[bestSeg];
			o.shares[index] = make_float4(share[0], share[1], share[2], own.w);
		}
	}
}

__global__ void __launch_bounds__(256)
open_remove_marked_kernel(float4 *pos, uint4 *marks, const particleinfo *info, uint32_t numParticles)
{
	const uint32_t i = blockIdx.x*256 + threadIdx.x;
	if (i >= numParticles) return;
	if (PART_TYPE(info[i]) != PT_FLUID) return;
	float4 row = pos[i];
	if (!is_active_w(row.w)) return;
	const uint4 m = marks[i];
	if ((m.x | m.y) == 0u) return;
	row.w = __uint_as_float(0x7fc00000u);      // disable_particle: the mass becomes NaN
	pos[i] = row;
	marks[i] = make_uint4(0u, 0u, 0u, 0u);
}

// ---- the boundary-condition passes on the open faces ---------------------------------------------------------------------------------
struct OpenFaceArgs {
	float4 *vel, *gGam, *eulerVel;          // in place
	const float4 *boundElement;
	const uint4 *triangles;
	const particleinfo *info;
	int step;
	// vertex pass
	float4 *newPos, *forces, *boundElementW;
	uint4 *trianglesW;
	particleinfo *infoW;
	uint32_t *hashW, *nextIDs, *newNumParticles;
	SegmentCorners corners;
	uint32_t totParticles, numOpenVertices;
	float deltap, dt;
};

// what a segment or a vertex takes from a fluid neighbour: its Shepard weight W V, its pressure, its velocity
struct FluidSample { float weight, pressure; float4 vel; };
__device__ __forceinline__ FluidSample sample_fluid(const DevParams &p, const float4 *vel, const particleinfo *info, uint32_t j, float dist, float mass)
{
	FluidSample s;
	const uint32_t fl = FLUID_NUM(info[j]);
	s.vel = vel[j];
	s.weight = kernel_W<SPHX_WENDLAND>(p, dist)*mass/((s.vel.w + 1.0f)*p.rho0[fl]);
	s.pressure = p.bcoeff[fl]*(powf(s.vel.w + 1.0f, p.gammacoeff[fl]) - 1.0f);
	return s;
}

// Shepard means over the fluid in reach of an open element: { sum w (u + u_E), sum w max(P, 0), sum w } (io_fluid_contrib, :852-909)
enum { SUM_UX, SUM_UY, SUM_UZ, SUM_P, SUM_W, SUM_RETURNED, FLUID_SUMS };

// One open SEGMENT per wave.  Its Eulerian state is the Shepard mean of the fluid in front of it run through the characteristic
// condition; gamma is re-derived from its three vertices when it has to be.  (impose_io_bc, :1362-1413)
__global__ void __launch_bounds__(ROW_THREADS)
open_segment_rows_kernel(DevParams p, RowLists a, const uint32_t *rows, OpenFaceArgs o)
{
	FOR_MY_ROWS(rows, index) {
		const particleinfo mine = o.info[index];
		const float4 own = a.pos[index], be = o.boundElement[index];
		const uint4 tri = o.triangles[index];
		const int3 cell = grid_pos_from_hash(p, a.hash[index] & CELLTYPE_BITMASK);
		const V3 n = v3(be.x, be.y, be.z);
		const bool bodies = (p.simflags & SPHX_ENABLE_MOVING_BODIES) != 0;
		const bool velocityDriven = SA_IS_VELOCITY_DRIVEN(mine);
		const bool carried = bodies && (mine.x & FG_MOVING_BOUNDARY);
		float4 imposed = o.eulerVel[index];
		if (velocityDriven) imposed.w = 0.0f;
		float gammaStored = o.gGam[index].w;
		const bool renewGamma = bodies || !is_active_w(gammaStored) || o.step == 0;
		// the three vertices: mean gamma (and mean velocity of a segment that rides on a body)
		float mean[7] = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f };
		{
			int carry = 0; bool more = true;
			for (int s0 = 0; more; s0 += 64) {
				const WaveEntry e = wave_entries<WAVE_SECTION_VERTEX>(p, a.list, a.cellStart, index, own, cell, s0, lane, carry, more);
				bool corner = false;
				float t[7] = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f };
				if (e.live && is_active_w(a.pos[e.j].w) && tri_has(tri, info_id(o.info[e.j]))) {
					corner = true;
					const float4 g = o.gGam[e.j], v = o.vel[e.j];
					t[0] = g.x; t[1] = g.y; t[2] = g.z; t[3] = g.w; t[4] = v.x; t[5] = v.y; t[6] = v.z;
				}
				ordered_sums(mean, t, wave_ballot(corner));
			}
		}
		float gamma = gammaStored;
		if (renewGamma) {
			const float third = 1.0f/3;
			const float4 g = make_float4(mean[0]*third, mean[1]*third, mean[2]*third, mean[3]*third);
			if (lane == 0u) o.gGam[index] = g;
			gamma = fmaxf(g.w, 1e-5f);
		}
		float4 result = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
		if (carried) { result.x = mean[4]/3; result.y = mean[5]/3; result.z = mean[6]/3; }
		// the fluid in front of the segment
		float sum[FLUID_SUMS] = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f };
		{
			int carry = 0; bool more = true;
			for (int s0 = 0; more; s0 += 64) {
				const WaveEntry e = wave_entries<WAVE_SECTION_FLUID>(p, a.list, a.cellStart, index, own, cell, s0, lane, carry, more);
				bool seen = false;
				float t[FLUID_SUMS] = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f };
				if (e.live) {
					const float4 np = a.pos[e.j];
					const float rx = e.ox - np.x, ry = e.oy - np.y, rz = e.oz - np.z;
					if (is_active_w(np.w)) {
						const float dist = sqrtf(rx*rx + ry*ry + rz*rz);
						const FluidSample f = sample_fluid(p, o.vel, o.info, e.j, dist, np.w);
						if (dist < p.influenceradius && (n.x*rx + n.y*ry + n.z*rz) < 0.0f) {
							seen = true;
							const float4 ne = o.eulerVel[e.j];
							t[SUM_UX] = f.weight*(f.vel.x + ne.x); t[SUM_UY] = f.weight*(f.vel.y + ne.y); t[SUM_UZ] = f.weight*(f.vel.z + ne.z);
							t[SUM_P] = f.weight*fmaxf(0.0f, f.pressure);
							t[SUM_W] = f.weight;
						}
					}
				}
				ordered_sums(sum, t, wave_ballot(seen));
			}
		}
		const uint32_t fl = FLUID_NUM(mine);
		const Tait eos(p, fl);
		V3 u;
		float inside;
		if (sum[SUM_W] > 0.1f*gamma) {
			u = v3(sum[SUM_UX]/sum[SUM_W], sum[SUM_UY]/sum[SUM_W], sum[SUM_UZ]/sum[SUM_W]);
			inside = eos.density_of(sum[SUM_P]/sum[SUM_W]);
		} else if (velocityDriven) {      // no fluid to speak of: the imposed velocity, at rest density
			u = v3(imposed.x, imposed.y, imposed.z);
			inside = 0.0f;
		} else {                          // ... or the imposed density, at rest
			u = v3(0.0f, 0.0f, 0.0f);
			inside = imposed.w;
		}
		const float uIn = u.x*n.x + u.y*n.y + u.z*n.z;
		const float uOut = imposed.x*n.x + imposed.y*n.y + imposed.z*n.z;
		const float4 state = open_face_state(eos, velocityDriven, imposed, inside, imposed.w, u, uIn, uOut, n);
		result.w = state.w;
		if (lane == 0u) { o.eulerVel[index] = state; o.vel[index] = result; }
	}
}

// One open non-corner VERTEX per wave: its Eulerian state as for a segment (without the half-space test), then its mass: it
// gains what flows in through its share of the adjacent open segments, takes over the particles that left through them, and
// releases a fluid particle when it has grown to half of one and the flow is inward.  (impose_vertex_io_bc :1168-1252,
// io_boundary_contrib :937-988, generate_new_particles :1101-1159, createNewFluidParticle :73-104)
__global__ void __launch_bounds__(ROW_THREADS)
open_vertex_rows_kernel(DevParams p, RowLists a, const uint32_t *rows, OpenFaceArgs o)
{
	FOR_MY_ROWS(rows, index) {
		const particleinfo mine = o.infoW[index];
		const float4 own = a.pos[index], be = o.boundElementW[index];
		const float gamma = o.gGam[index].w;
		const uint32_t fl = FLUID_NUM(mine), id = info_id(mine);
		const bool velocityDriven = SA_IS_VELOCITY_DRIVEN(mine);
		const V3 n = v3(be.x, be.y, be.z);
		const float fullMass = o.deltap*o.deltap*o.deltap*p.rho0[fl];
		const int3 cell = grid_pos_from_hash(p, o.hashW[index] & CELLTYPE_BITMASK);
		float sum[FLUID_SUMS] = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f };
		{
			int carry = 0; bool more = true;
			for (int s0 = 0; more; s0 += 64) {
				const WaveEntry e = wave_entries<WAVE_SECTION_FLUID>(p, a.list, a.cellStart, index, own, cell, s0, lane, carry, more);
				bool seen = false;
				float t[FLUID_SUMS] = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f };
				if (e.live) {
					const float4 np = a.pos[e.j];
					const float rx = e.ox - np.x, ry = e.oy - np.y, rz = e.oz - np.z;
					if (is_active_w(np.w)) {
						const float dist = sqrtf(rx*rx + ry*ry + rz*rz);
						const FluidSample f = sample_fluid(p, o.vel, o.infoW, e.j, dist, np.w);
						if (dist < p.influenceradius) {
							seen = true;
							const float4 ne = o.eulerVel[e.j];
							t[SUM_UX] = f.weight*(f.vel.x + ne.x); t[SUM_UY] = f.weight*(f.vel.y + ne.y); t[SUM_UZ] = f.weight*(f.vel.z + ne.z);
							t[SUM_P] = f.weight*fmaxf(0.0f, f.pressure);
							t[SUM_W] = f.weight;
							if (o.step == 2) {      // a particle marked as leaving: its mass, by this vertex's share
								const uint4 m = o.trianglesW[e.j];
								if ((m.x | m.y) != 0u) {
									const float4 s = o.gGam[e.j];
									const float share = m.x == id ? s.x : m.y == id ? s.y : m.z == id ? s.z : 0.0f;
									// adding +0 to a sum that is +0 or positive changes nothing: a lane without a share adds nothing
									if (share > 0) t[SUM_RETURNED] = share*s.w;
								}
							}
						}
					}
				}
				ordered_sums(sum, t, wave_ballot(seen));
			}
		}
		// the mass flux through the adjacent open segments, each by this vertex's share of a point at the segment's centre
		float influx[1] = { 0.0f };
		{
			int carry = 0; bool more = true;
			for (int s0 = 0; more; s0 += 64) {
				const WaveEntry e = wave_entries<WAVE_SECTION_BOUNDARY>(p, a.list, a.cellStart, index, own, cell, s0, lane, carry, more);
				bool mineToo = false;
				float t[1] = { 0.0f };
				if (e.live) {
					const uint4 tri = o.trianglesW[e.j];
					const particleinfo f = o.infoW[e.j];
					if (tri_has(tri, id) && SA_IS_OPEN(f)) {
						mineToo = true;
						const float4 nb = o.boundElementW[e.j];
						const V3 nn = v3(nb.x, nb.y, nb.z);
						V3 q[3];
						corners_of(o.corners, e.j, nn, -1.0f, q);
						float share[3];
						vertex_shares(q, nn, share);
						const float w = tri.x == id ? share[0] : tri.y == id ? share[1] : tri.z == id ? share[2] : 0.0f;
						const float4 ne = o.eulerVel[e.j];
						t[0] = ((o.vel[e.j].w + 1.0f)*p.rho0[FLUID_NUM(f)])*nb.w*w*(ne.x*nn.x + ne.y*nn.y + ne.z*nn.z);
					}
				}
				ordered_sums(influx, t, wave_ballot(mineToo));
			}
		}
		const float massIn = influx[0];
		const float shepard = fmaxf(sum[SUM_W], 0.1f*gamma);
		const Tait eos(p, fl);
		float4 state = o.eulerVel[index];
		if (shepard > 0.1f*gamma) {
			const V3 u = v3(sum[SUM_UX]/shepard, sum[SUM_UY]/shepard, sum[SUM_UZ]/shepard);
			const float inside = eos.density_of(sum[SUM_P]/shepard);
			const float uIn = u.x*n.x + u.y*n.y + u.z*n.z;
			const float uOut = state.x*n.x + state.y*n.y + state.z*n.z;
			state = open_face_state(eos, velocityDriven, state, inside, state.w, u, uIn, uOut, n);
		} else if (velocityDriven)
			state.w = 0.0f;
		else
			state.x = state.y = state.z = 0.0f;
		float4 row = own;
		const float un = n.x*state.x + n.y*state.y + n.z*state.z;
		if (o.step != 0) {
			row.w += o.dt*massIn;
			if (shepard < 0.1f*gamma && massIn < 0.0f) row.w = 0.0f;
			row.w = fmaxf(-2.0f*fullMass, fminf(2.0f*fullMass, row.w));
			if (massIn < 0.0f || un < 1e-5f*p.sscoeff[fl]) {
				const float bound = fullMass*be.w;      // be.w of a vertex is NaN: fminf / fmaxf hand back the other operand
				row.w = fmaxf(-bound, fminf(bound, row.w));
			}
		}
		float returned = sum[SUM_RETURNED];
		const bool release = o.step == 2 && row.w > fullMass*0.5f && massIn > 0 && un > 1e-5f && (velocityDriven || state.w > 1e-5f);
		if (lane == 0u) {
			o.eulerVel[index] = state;
			o.vel[index].w = state.w;
			if (release) {
				const uint32_t born = atomicAdd(o.newNumParticles, 1u);
				if (born < o.totParticles) {
					const uint32_t bornId = o.nextIDs[index];
					o.nextIDs[index] = bornId + o.numOpenVertices;
					particleinfo f;
					f.x = PT_FLUID; f.y = (unsigned short)(fl << 12); f.z = (unsigned short)(bornId & 0xFFFFu); f.w = (unsigned short)(bornId >> 16);
					float4 at = row;
					at.w = fullMass;
					returned -= at.w;
					o.newPos[born] = at;
					o.infoW[born] = f;
					o.hashW[born] = o.hashW[index] & CELLTYPE_BITMASK;
					o.vel[born] = state;
					o.gGam[born] = o.gGam[index];
					o.eulerVel[born] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
					o.forces[born] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
					o.trianglesW[born] = make_uint4(0u, 0u, 0u, 0u);
					o.nextIDs[born] = 0xFFFFFFFFu;
					const float none = __uint_as_float(0xffc00000u);
					o.boundElementW[born] = make_float4(none, none, none, none);
				}
			}
			row.w += returned;
			o.newPos[index] = row;
		}
	}
}

// ---- the water level at the pressure-driven faces: the highest fluid neighbour of their vertices, as a fixed-point height -------------
__global__ void __launch_bounds__(ROW_THREADS)
open_depth_rows_kernel(DevParams p, RowLists a, const uint32_t *rows, const particleinfo *info, uint32_t *level)
{
	FOR_MY_ROWS(rows, index) {
		const float4 own = a.pos[index];
		const int3 cell = grid_pos_from_hash(p, a.hash[index] & CELLTYPE_BITMASK);
		uint32_t top = 0u;
		int carry = 0; bool more = true;
		for (int s0 = 0; more; s0 += 64) {
			const WaveEntry e = wave_entries<WAVE_SECTION_FLUID>(p, a.list, a.cellStart, index, own, cell, s0, lane, carry, more);
			if (e.live) {
				const float4 np = a.pos[e.j];
				const float rx = e.ox - np.x, ry = e.oy - np.y, rz = e.oz - np.z;
				const float dist = sqrtf(rx*rx + ry*ry + rz*rz);
				if (is_active_w(np.w) && !(dist >= p.influenceradius) && !(rz < 0.0f)) {
					float z = own.z - rz + cell.z*p.cs[2] + 0.5f*p.cs[2];
					z *= ((float)UINT_MAX)/(p.gs[2]*p.cs[2]);
					const uint32_t h = (uint32_t)z;
					top = h > top ? h : top;
				}
			}
		}
		top = wave_max_u(top);
		if (top && lane == 0u) atomicMax(level + OBJECT_NUM(info[index]), top);
	}
}

// ---- FLUX_COMPUTATION: per open face the volume flux sum A_s (u_E . n_s) over its segments (the sums start from zero here; the
// reference adds onto a freshly allocated array) ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
open_flux_kernel(const particleinfo *info, const float4 *eulerVel, const float4 *boundElement, float *flux, uint32_t faces, uint32_t numParticles)
{
	const uint32_t i = blockIdx.x*256 + threadIdx.x;
	if (i >= numParticles) return;
	const particleinfo f = info[i];
	if (!(SA_IS_OPEN(f) && PART_TYPE(f) == PT_BOUNDARY)) return;
	const uint32_t face = OBJECT_NUM(f);
	if (face >= faces) return;
	const float4 be = boundElement[i], e = eulerVel[i];
	atomicAdd(flux + face, be.w*(e.x*be.x + e.y*be.y + e.z*be.z));
}

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
static int open_check(sphx_ctx *ctx, const char *who)
{
	if (!ctx || !ctx->have_params) return sphx_set_error(SPHX_ERR_INVALID, "sphx_sa_io: constants not set");
	if (ctx->params.boundarytype != SPHX_SA_BOUNDARY) return sphx_set_error(SPHX_ERR_INVALID, who);
	return SPHX_OK;
}
static int open_bc_check(sphx_ctx *ctx, const char *who)
{
	int rc = open_check(ctx, who);
	if (rc != SPHX_OK) return rc;
	if (ctx->params.kerneltype != SPHX_WENDLAND)
		return sphx_set_error(SPHX_ERR_UNSUPPORTED, "sphx_sa_io: SA_BOUNDARY is built for the Wendland kernel");
	if (ctx->dev.turbmodel == SPHX_KEPSILON)
		return sphx_set_error(SPHX_ERR_UNSUPPORTED, "sphx_sa_io: open boundaries with k-epsilon are not built");
	return SPHX_OK;
}

// the index of a pass, in the context's scratch (one pass at a time per context, in stream order)
template<int KIND>
static int open_rows(sphx_ctx *ctx, RowSweep sw, hipStream_t st, uint32_t **rows)
{
	const uint32_t need = ctx->params.neiblist_stride + 1u;
	if (ctx->open_rows_cap < need) {
		if (ctx->open_rows) (void)hipFree(ctx->open_rows);
		ctx->open_rows = nullptr; ctx->open_rows_cap = 0;
		SPHX_HIP(hipMalloc((void**)&ctx->open_rows, sizeof(uint32_t)*(size_t)need));
		ctx->open_rows_cap = need;
	}
	sw.rows = ctx->open_rows;
	sw.stride = ctx->dev.stride; sw.neibboundpos = ctx->dev.neibboundpos;
	SPHX_HIP(hipMemsetAsync(ctx->open_rows, 0, sizeof(uint32_t), st));
	SPHX_LAUNCH_WAVES(open_rows_kernel<KIND>, div_up_u(sw.end - sw.first, 256), 256, st, sw);
	SPHX_LAUNCH_CHECK("open_rows_kernel");
	*rows = ctx->open_rows;
	return SPHX_OK;
}

extern "C" int sphx_sa_identify_corner_vertices(sphx_ctx *ctx, const void *pos, void *info, const uint32_t *hash, const void *vertices,
	const uint32_t *cellStart, const uint16_t *neibsList, uint32_t numParticles, uint32_t particleRangeEnd, void *stream)
{
	(void)numParticles;
	int rc = open_check(ctx, "saIdentifyCornerVertices called without SA_BOUNDARY");
	if (rc != SPHX_OK) return rc;
	SPHX_REQUIRE(pos && info && hash && vertices && cellStart && neibsList, "sphx_sa_identify_corner_vertices: missing buffer");
	SPHX_REQUIRE(particleRangeEnd <= ctx->params.neiblist_stride, "sphx_sa_identify_corner_vertices: range exceeds the neighbour list stride");
	if (!particleRangeEnd) return SPHX_OK;
	hipStream_t st = (hipStream_t)stream;
	RowSweep sw = {};
	sw.info = (const particleinfo*)info; sw.end = particleRangeEnd;
	uint32_t *rows;
	rc = open_rows<ROWS_OPEN_VERTICES>(ctx, sw, st, &rows);
	if (rc != SPHX_OK) return rc;
	const RowLists l = { (const float4*)pos, hash, cellStart, neibsList };
	// a row's flag is read by nobody in this launch: the test looks at segments, the flag sits on vertices
	SPHX_LAUNCH_WAVES(open_corner_rows_kernel, ROW_GRID, ROW_THREADS, st, ctx->dev, l, rows, (const uint4*)vertices, (const particleinfo*)info, (particleinfo*)info);
	SPHX_LAUNCH_CHECK("open_corner_rows_kernel");
	return SPHX_OK;
}

static int open_mass_pass(sphx_ctx *ctx, bool second, const void *pos, const void *forces, const void *vertices, const uint32_t *hash,
	const void *info, const uint32_t *cellStart, const uint16_t *neibsList, void *newPos, uint32_t particleRangeEnd, float deltap, hipStream_t st)
{
	SPHX_REQUIRE(particleRangeEnd <= ctx->params.neiblist_stride, "sphx_sa_init_io_mass: range exceeds the neighbour list stride");
	RowSweep sw = {};
	sw.info = (const particleinfo*)info; sw.end = particleRangeEnd;
	uint32_t *rows;
	int rc = open_rows<ROWS_OPEN_INNER_VERTICES>(ctx, sw, st, &rows);
	if (rc != SPHX_OK) return rc;
	const RowLists l = { (const float4*)pos, hash, cellStart, neibsList };
	OpenMassArgs o = { (const uint4*)vertices, (const particleinfo*)info, (float4*)const_cast<void*>(forces), (float4*)newPos, deltap };
	if (second) SPHX_LAUNCH_WAVES(open_mass_rows_kernel<true>, ROW_GRID, ROW_THREADS, st, ctx->dev, l, rows, o);
	else SPHX_LAUNCH_WAVES(open_mass_rows_kernel<false>, ROW_GRID, ROW_THREADS, st, ctx->dev, l, rows, o);
	SPHX_LAUNCH_CHECK("open_mass_rows_kernel");
	return SPHX_OK;
}

extern "C" int sphx_sa_init_io_mass_vertex_count(sphx_ctx *ctx, const void *vertices, const uint32_t *hash, const void *info,
	const uint32_t *cellStart, const uint16_t *neibsList, void *forces, const void *pos,
	uint32_t numParticles, uint32_t particleRangeEnd, void *stream)
{
	(void)numParticles;
	int rc = open_check(ctx, "initIOmass_vertexCount called without SA_BOUNDARY");
	if (rc != SPHX_OK) return rc;
	SPHX_REQUIRE(vertices && hash && info && cellStart && neibsList && forces && pos, "sphx_sa_init_io_mass_vertex_count: missing buffer");
	if (!particleRangeEnd) return SPHX_OK;
	return open_mass_pass(ctx, false, pos, forces, vertices, hash, info, cellStart, neibsList, nullptr, particleRangeEnd, 0.0f, (hipStream_t)stream);
}

extern "C" int sphx_sa_init_io_mass(sphx_ctx *ctx, const void *oldPos, const void *forces, const void *vertices, const uint32_t *hash,
	const void *info, const uint32_t *cellStart, const uint16_t *neibsList, void *newPos,
	uint32_t numParticles, uint32_t particleRangeEnd, float deltap, void *stream)
{
	(void)numParticles;
	int rc = open_check(ctx, "initIOmass called without SA_BOUNDARY");
	if (rc != SPHX_OK) return rc;
	SPHX_REQUIRE(oldPos && forces && vertices && hash && info && cellStart && neibsList && newPos && oldPos != newPos,
		"sphx_sa_init_io_mass: missing buffer (newPos must not be oldPos)");
	if (!particleRangeEnd) return SPHX_OK;
	// every row is carried over; the open vertices then get their new masses
	SPHX_HIP(hipMemcpyAsync(newPos, oldPos, sizeof(float4)*(size_t)particleRangeEnd, hipMemcpyDeviceToDevice, (hipStream_t)stream));
	return open_mass_pass(ctx, true, oldPos, forces, vertices, hash, info, cellStart, neibsList, newPos, particleRangeEnd, deltap, (hipStream_t)stream);
}

extern "C" int sphx_sa_find_outgoing_segment(sphx_ctx *ctx, const void *pos, const void *vel, void *vertices, void *gGam,
	const void *vertPos0, const void *vertPos1, const void *vertPos2, const void *boundElements, const void *info,
	const uint32_t *hash, const uint32_t *cellStart, const uint16_t *neibsList,
	uint32_t numParticles, uint32_t particleRangeEnd, float influenceradius, void *stream)
{
	(void)numParticles;
	int rc = open_check(ctx, "findOutgoingSegment called without SA_BOUNDARY");
	if (rc != SPHX_OK) return rc;
	SPHX_REQUIRE(pos && vel && vertices && gGam && vertPos0 && vertPos1 && vertPos2 && boundElements && info && hash && cellStart && neibsList,
		"sphx_sa_find_outgoing_segment: missing buffer");
	SPHX_REQUIRE(particleRangeEnd <= ctx->params.neiblist_stride, "sphx_sa_find_outgoing_segment: range exceeds the neighbour list stride");
	if (!particleRangeEnd) return SPHX_OK;
	hipStream_t st = (hipStream_t)stream;
	RowSweep sw = {};
	sw.info = (const particleinfo*)info; sw.pos = (const float4*)pos; sw.marks = (const uint4*)vertices; sw.list = neibsList; sw.end = particleRangeEnd;
	uint32_t *rows;
	// the candidates: the fluid particles with boundary elements in reach -- the list the neighbour-list build left for this list, if
	// there is one (the sweep that makes it anew costs 0.23 ms at 8.6 M particles), else a sweep of this pass
	const bool listed = ctx->sa_wall && ctx->sa_wall_neibslist == (const void*)neibsList && particleRangeEnd <= ctx->sa_rows_range;
	if (listed) rows = ctx->sa_wall;
	else {
		rc = open_rows<ROWS_LEAVING_CANDIDATES>(ctx, sw, st, &rows);
		if (rc != SPHX_OK) return rc;
	}
	const RowLists l = { (const float4*)pos, hash, cellStart, neibsList };
	LeavingArgs o = { (const float4*)vel, (const float4*)boundElements, { (const float2*)vertPos0, (const float2*)vertPos1, (const float2*)vertPos2 },
		(const particleinfo*)info, (uint4*)vertices, (float4*)gGam, influenceradius, listed ? 1 : 0, particleRangeEnd };
	SPHX_LAUNCH_WAVES(open_leaving_rows_kernel, ROW_GRID, ROW_THREADS, st, ctx->dev, l, rows, o);
	SPHX_LAUNCH_CHECK("open_leaving_rows_kernel");
	return SPHX_OK;
}

extern "C" int sphx_sa_disable_outgoing_parts(sphx_ctx *ctx, void *pos, void *vertices, const void *info, uint32_t numParticles, void *stream)
{
	int rc = open_check(ctx, "disableOutgoingParts called without SA_BOUNDARY");
	if (rc != SPHX_OK) return rc;
	SPHX_REQUIRE(pos && vertices && info, "sphx_sa_disable_outgoing_parts: missing buffer");
	if (!numParticles) return SPHX_OK;
	SPHX_LAUNCH(open_remove_marked_kernel, div_up_u(numParticles, 256), 256, (hipStream_t)stream, (float4*)pos, (uint4*)vertices,
		(const particleinfo*)info, numParticles);
	SPHX_LAUNCH_CHECK("open_remove_marked_kernel");
	return SPHX_OK;
}

extern "C" int sphx_sa_segment_bc_io(sphx_ctx *ctx, void *vel, void *gGam, void *eulerVel, const void *pos, const void *vertices,
	const void *boundElements, const void *info, const uint32_t *hash, const uint32_t *cellStart, const uint16_t *neibsList,
	uint32_t numParticles, uint32_t particleRangeEnd, int step, void *stream)
{
	(void)numParticles;
	int rc = open_bc_check(ctx, "saSegmentBoundaryConditions called without SA_BOUNDARY");
	if (rc != SPHX_OK) return rc;
	SPHX_REQUIRE(vel && gGam && eulerVel && pos && vertices && boundElements && info && hash && cellStart && neibsList,
		"sphx_sa_segment_bc_io: missing buffer");
	SPHX_REQUIRE(particleRangeEnd <= ctx->params.neiblist_stride, "sphx_sa_segment_bc_io: range exceeds the neighbour list stride");
	if (!particleRangeEnd) return SPHX_OK;
	hipStream_t st = (hipStream_t)stream;
	const int s = step == -1 ? 0 : step;
	// the walls: sa_bounds.hip's pass, which leaves the open segments alone and clears the walls' Eulerian velocity.  A segment
	// reads vertex and fluid rows only, so the two launches do not see each other's writes
	SaArgs w = {};
	w.vel = (float4*)vel; w.gGam = (float4*)gGam; w.eulerVel = (float4*)eulerVel; w.pos = (const float4*)pos; w.vertices = (const uint4*)vertices;
	w.boundElement = (float4*)const_cast<void*>(boundElements); w.info = (const particleinfo*)info; w.hash = hash; w.cellStart = cellStart;
	w.neibsList = neibsList; w.numParticles = particleRangeEnd; w.step = s; w.openFaces = 1;
	rc = sphx_sa_solid_rows_launch(ctx, w, false, st);
	if (rc != SPHX_OK) return rc;
	RowSweep sw = {};
	sw.info = (const particleinfo*)info; sw.end = particleRangeEnd;
	uint32_t *rows;
	rc = open_rows<ROWS_OPEN_SEGMENTS>(ctx, sw, st, &rows);
	if (rc != SPHX_OK) return rc;
	const RowLists l = { (const float4*)pos, hash, cellStart, neibsList };
	OpenFaceArgs o = {};
	o.vel = (float4*)vel; o.gGam = (float4*)gGam; o.eulerVel = (float4*)eulerVel; o.boundElement = (const float4*)boundElements;
	o.triangles = (const uint4*)vertices; o.info = (const particleinfo*)info; o.step = s;
	SPHX_LAUNCH_WAVES(open_segment_rows_kernel, ROW_GRID, ROW_THREADS, st, ctx->dev, l, rows, o);
	SPHX_LAUNCH_CHECK("open_segment_rows_kernel");
	return SPHX_OK;
}

extern "C" int sphx_sa_vertex_bc_io(sphx_ctx *ctx, void *vel, const void *pos, void *newPos, void *gGam, void *eulerVel, void *forces,
	void *vertices, void *boundElements, const void *vertPos0, const void *vertPos1, const void *vertPos2, void *info, uint32_t *hash,
	uint32_t *nextIDs, uint32_t *newNumParticles, const uint32_t *cellStart, const uint16_t *neibsList,
	uint32_t numParticles, uint32_t particleRangeEnd, uint32_t totParticles, float deltap, float dt, int step,
	uint32_t numOpenVertices, void *stream)
{
	(void)numParticles;
	int rc = open_bc_check(ctx, "saVertexBoundaryConditions called without SA_BOUNDARY");
	if (rc != SPHX_OK) return rc;
	SPHX_REQUIRE(vel && pos && newPos && gGam && eulerVel && forces && vertices && boundElements && vertPos0 && vertPos1 &&
		vertPos2 && info && hash && nextIDs && newNumParticles && cellStart && neibsList, "sphx_sa_vertex_bc_io: missing buffer");
	SPHX_REQUIRE(particleRangeEnd <= ctx->params.neiblist_stride, "sphx_sa_vertex_bc_io: range exceeds the neighbour list stride");
	if (!particleRangeEnd) return SPHX_OK;
	hipStream_t st = (hipStream_t)stream;
	const int s = step == -1 ? 0 : step;
	// the walls' vertices (and the corners of the open faces) first, with sa_bounds.hip's pass.  A vertex reads fluid and segment
	// rows and writes its own (pos == newPos, the reference's call, is fine: xyz unchanged, the masses read are fluid rows'); the
	// rows of released particles lie behind particleRangeEnd
	SaArgs w = {};
	w.vel = (float4*)vel; w.gGam = (float4*)gGam; w.pos = (const float4*)pos; w.info = (const particleinfo*)info; w.hash = hash;
	w.cellStart = cellStart; w.neibsList = neibsList; w.numParticles = particleRangeEnd; w.step = s; w.openFaces = 1;
	rc = sphx_sa_solid_rows_launch(ctx, w, true, st);
	if (rc != SPHX_OK) return rc;
	RowSweep sw = {};
	sw.info = (const particleinfo*)info; sw.end = particleRangeEnd;
	uint32_t *rows;
	rc = open_rows<ROWS_OPEN_INNER_VERTICES>(ctx, sw, st, &rows);
	if (rc != SPHX_OK) return rc;
	const RowLists l = { (const float4*)pos, hash, cellStart, neibsList };
	OpenFaceArgs o = {};
	o.vel = (float4*)vel; o.gGam = (float4*)gGam; o.eulerVel = (float4*)eulerVel; o.step = s;
	o.newPos = (float4*)newPos; o.forces = (float4*)forces; o.boundElementW = (float4*)boundElements; o.trianglesW = (uint4*)vertices;
	o.infoW = (particleinfo*)info; o.hashW = hash; o.nextIDs = nextIDs; o.newNumParticles = newNumParticles;
	o.corners.c0 = (const float2*)vertPos0; o.corners.c1 = (const float2*)vertPos1; o.corners.c2 = (const float2*)vertPos2;
	o.totParticles = totParticles; o.numOpenVertices = numOpenVertices; o.deltap = deltap; o.dt = dt;
	SPHX_LAUNCH_WAVES(open_vertex_rows_kernel, ROW_GRID, ROW_THREADS, st, ctx->dev, l, rows, o);
	SPHX_LAUNCH_CHECK("open_vertex_rows_kernel");
	return SPHX_OK;
}

extern "C" int sphx_sa_io_water_depth(sphx_ctx *ctx, uint32_t *IOwaterdepth, const void *pos, const void *info, const uint32_t *hash,
	const uint32_t *cellStart, const uint16_t *neibsList, uint32_t numParticles, uint32_t fromParticle, uint32_t toParticle, void *stream)
{
	int rc = open_bc_check(ctx, "the water depth is measured with SA_BOUNDARY only");
	if (rc != SPHX_OK) return rc;
	SPHX_REQUIRE(IOwaterdepth && pos && info && hash && cellStart && neibsList, "sphx_sa_io_water_depth: missing buffer");
	SPHX_REQUIRE(fromParticle <= toParticle && toParticle <= numParticles, "sphx_sa_io_water_depth: invalid particle range");
	SPHX_REQUIRE(toParticle <= ctx->params.neiblist_stride, "sphx_sa_io_water_depth: range exceeds the neighbour list stride");
	if (fromParticle == toParticle) return SPHX_OK;
	hipStream_t st = (hipStream_t)stream;
	RowSweep sw = {};
	sw.info = (const particleinfo*)info; sw.pos = (const float4*)pos; sw.first = fromParticle; sw.end = toParticle;
	uint32_t *rows;
	rc = open_rows<ROWS_DEPTH_GAUGES>(ctx, sw, st, &rows);
	if (rc != SPHX_OK) return rc;
	const RowLists l = { (const float4*)pos, hash, cellStart, neibsList };
	SPHX_LAUNCH_WAVES(open_depth_rows_kernel, ROW_GRID, ROW_THREADS, st, ctx->dev, l, rows, (const particleinfo*)info, IOwaterdepth);
	SPHX_LAUNCH_CHECK("open_depth_rows_kernel");
	return SPHX_OK;
}

extern "C" int sphx_flux_computation(sphx_ctx *ctx, float *IOflux, const void *info, const void *eulerVel, const void *boundElements,
	uint32_t numParticles, uint32_t particleRangeEnd, uint32_t numOpenBoundaries, void *stream)
{
	(void)numParticles;
	int rc = open_bc_check(ctx, "the flux through open boundaries is computed with SA_BOUNDARY only");
	if (rc != SPHX_OK) return rc;
	SPHX_REQUIRE(IOflux && info && eulerVel && boundElements, "sphx_flux_computation: missing buffer");
	if (!numOpenBoundaries) return SPHX_OK;
	SPHX_HIP(hipMemsetAsync(IOflux, 0, numOpenBoundaries*sizeof(float), (hipStream_t)stream));
	if (!particleRangeEnd) return SPHX_OK;
	SPHX_LAUNCH(open_flux_kernel, div_up_u(particleRangeEnd, 256), 256, (hipStream_t)stream, (const particleinfo*)info,
		(const float4*)eulerVel, (const float4*)boundElements, IOflux, numOpenBoundaries, particleRangeEnd);
	SPHX_LAUNCH_CHECK("open_flux_kernel");
	return SPHX_OK;
}
