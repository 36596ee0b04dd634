// sa_io.hip -- open boundaries of the semi-analytical wall model (SA_BOUNDARY + ENABLE_INLET_OUTLET): the FIRST
// kernels of that half of the row, for gfx950.  Replaces, of CUDABoundaryConditionsEngine,
//   saIdentifyCornerVertices   src/cuda/boundary_conditions.cu:667   saIdentifyCornerVerticesDevice   _kernel.cu:2319-2362
//   initIOmass_vertexCount     src/cuda/boundary_conditions.cu:578   initIOmass_vertexCountDevice     _kernel.cu:1999-2064
//   initIOmass                 src/cuda/boundary_conditions.cu:610   initIOmassDevice                 _kernel.cu:2078-2172
//   findOutgoingSegment        src/cuda/boundary_conditions.cu:238   findOutgoingSegmentDevice        _kernel.cu:1647-1750
//   disableOutgoingParts       src/cuda/boundary_conditions.cu:76    disableOutgoingPartsDevice       _kernel.cu:2374-2398
// The boundary-condition passes with open boundaries, the density summation and the forces with the Eulerian velocity, and the
// command sequence are NOT built (sphx_sa_segment_bc & co. still refuse ENABLE_INLET_OUTLET); the CPU oracle restates all of
// them already (oracle/sph_oracle.c "Open boundaries", tests/test_sa_io_oracle.py), so these five are what a run needs besides.
// One thread per particle over the reference's u16 list, the reference's operation order (no FMA contraction; the areas of
// getMassRepartitionFactor in double where the reference's 0.5*dot(...) promotes them): bit-identical to the oracle.
#include "sphx_internal.h"
#include "neib_iter.h"

// Every launch of this file goes through one macro, so that tests/hostemu can run the kernels' SOURCE on the host, thread after
// thread, against the oracle before a GPU is at hand (a test harness: the library has no CPU path and the macro below is what
// hipcc sees).
#ifndef SPHX_LAUNCH
#define SPHX_LAUNCH(kernel, grid, block, stream, ...) kernel<<<(grid), (block), 0, (stream)>>>(__VA_ARGS__)
#endif

// particleinfo flags of open boundaries (src/particleinfo.h:153-156, 222-241)
#define FG_INLET             (PART_FLAG_START << 2)
#define FG_OUTLET            (PART_FLAG_START << 3)
#define FG_VELOCITY_DRIVEN   (PART_FLAG_START << 4)
#define FG_CORNER            (PART_FLAG_START << 5)
#define IS_IO_BOUNDARY(f)    ((f).x & (FG_INLET | FG_OUTLET))
#define IS_VEL_IO(f)         ((f).x & FG_VELOCITY_DRIVEN)
#define IS_CORNER(f)         ((f).x & FG_CORNER)
#define SA_IO_MAXNEIBVERTS 30       // boundary_conditions_kernel.cu:1977

struct SaIoArgs {
	const float4 *pos;             // the walker prefetches a position row per list entry
	const uint32_t *hash, *cellStart;
	const neibdata *neibsList;
	const uint4 *vertices;
	const particleinfo *info;
	uint32_t numParticles;
};

__device__ __forceinline__ bool io_has_vertex(const uint4 &v, uint32_t id) { return v.x == id || v.y == id || v.z == id; }

__global__ void __launch_bounds__(128)
sa_identify_corner_vertices_kernel(DevParams p, SaIoArgs a, particleinfo *infoOut)
{
	const uint32_t index = blockIdx.x*128 + threadIdx.x;
	if (index >= a.numParticles) return;
	particleinfo info = a.info[index];
	if (!(PART_TYPE(info) == PT_VERTEX && IS_IO_BOUNDARY(info))) return;
	const uint32_t obj = OBJECT_NUM(info), my_id = info_id(info);
	const float4 pos = a.pos[index];
	const int3 gridPos = grid_pos_from_hash(p, a.hash[index] & CELLTYPE_BITMASK);
	bool corner = false;
	for_each_neib<PT_BOUNDARY>(p, a, index, pos, gridPos, [&](uint32_t j, const float4 &, float, float, float) {
		const particleinfo ninfo = a.info[j];
		// a segment that is not of this open boundary and holds this vertex
		if (!(obj == OBJECT_NUM(ninfo) && IS_IO_BOUNDARY(ninfo)) && io_has_vertex(a.vertices[j], my_id)) corner = true;
	});
	if (corner) { info.x |= FG_CORNER; infoOut[index] = info; }
}

// the ids of the other vertices of the open-boundary segments vertex `index` belongs to, in list order (both kernels below)
__device__ __forceinline__ uint32_t io_adjacent_vertex_ids(const DevParams &p, const SaIoArgs &a, uint32_t index, const float4 &pos,
	const int3 &gridPos, uint32_t my_id, uint32_t *ids)
{
	uint32_t count = 0;
	for_each_neib<PT_BOUNDARY>(p, a, index, pos, gridPos, [&](uint32_t j, const float4 &, float, float, float) {
		if (!IS_IO_BOUNDARY(a.info[j])) return;
		const uint4 nv = a.vertices[j];
		if (!io_has_vertex(nv, my_id)) return;
		if (my_id != nv.x && count < SA_IO_MAXNEIBVERTS) ids[count++] = nv.x;
		if (my_id != nv.y && count < SA_IO_MAXNEIBVERTS) ids[count++] = nv.y;
		if (my_id != nv.z && count < SA_IO_MAXNEIBVERTS) ids[count++] = nv.z;
	});
	return count;
}

__global__ void __launch_bounds__(128)
sa_init_io_mass_vertex_count_kernel(DevParams p, SaIoArgs a, float4 *forces)
{
	const uint32_t index = blockIdx.x*128 + threadIdx.x;
	if (index >= a.numParticles) return;
	const particleinfo info = a.info[index];
	if (!(PART_TYPE(info) == PT_VERTEX && IS_IO_BOUNDARY(info) && !IS_CORNER(info))) return;
	const float4 pos = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
	const int3 gridPos = grid_pos_from_hash(p, a.hash[index] & CELLTYPE_BITMASK);
	uint32_t ids[SA_IO_MAXNEIBVERTS];
	const uint32_t nids = io_adjacent_vertex_ids(p, a, index, pos, gridPos, info_id(info), ids);
	uint32_t vertexCount = 0;
	for_each_neib<PT_VERTEX>(p, a, index, pos, gridPos, [&](uint32_t j, const float4 &, float, float, float) {
		const particleinfo ninfo = a.info[j];
		const uint32_t nid = info_id(ninfo);
		for (uint32_t k = 0; k < nids; ++k)
			if (nid == ids[k] && !IS_CORNER(ninfo)) vertexCount += 1;
	});
	forces[index].w = (float)vertexCount;
}

__global__ void __launch_bounds__(128)
sa_init_io_mass_kernel(DevParams p, SaIoArgs a, const float4 *forces, float4 *newPos, float deltap)
{
	const uint32_t index = blockIdx.x*128 + threadIdx.x;
	if (index >= a.numParticles) return;
	const particleinfo info = a.info[index];
	const float4 pos = a.pos[index];
	newPos[index] = pos;
	if (!(PART_TYPE(info) == PT_VERTEX && IS_IO_BOUNDARY(info) && !IS_CORNER(info))) return;
	const int3 gridPos = grid_pos_from_hash(p, a.hash[index] & CELLTYPE_BITMASK);
	const bool getMass = (info_id(info) % 2u) != 0u;      // odd ids take, even ids give
	float massChange = 0.0f;
	const float refMass = 0.5f*deltap*deltap*deltap*p.rho0[FLUID_NUM(info)];      // half a fluid particle
	const float massDiff = refMass - pos.w;
	const float vertexCount = forces[index].w;
	uint32_t ids[SA_IO_MAXNEIBVERTS];
	const uint32_t nids = io_adjacent_vertex_ids(p, a, index, pos, gridPos, info_id(info), ids);
	for_each_neib<PT_VERTEX>(p, a, index, pos, gridPos, [&](uint32_t j, const float4 &npos, float, float, float) {
		const particleinfo ninfo = a.info[j];
		const uint32_t nid = info_id(ninfo);
		for (uint32_t k = 0; k < nids; ++k) {
			if (nid != ids[k]) continue;
			const bool neibGetMass = (nid % 2u) != 0u;
			if (getMass != neibGetMass && !IS_CORNER(ninfo)) {
				if (getMass) {
					if (massDiff > 0.0f) massChange += massDiff/vertexCount;
				} else {
					const float neibMassDiff = refMass - npos.w;
					if (neibMassDiff > 0.0f) massChange -= neibMassDiff/forces[j].w;
				}
			}
		}
	});
	newPos[index].w = pos.w + massChange;
}

// ---- getMassRepartitionFactor (_kernel.cu:213-283) and calcVertexRelPos (src/cuda/gamma.cuh) for findOutgoingSegment ----
struct IoV3 { float x, y, z; };
__device__ __forceinline__ IoV3 iov(float x, float y, float z) { IoV3 r = { x, y, z }; return r; }
__device__ __forceinline__ IoV3 io_sub(IoV3 a, IoV3 b) { return iov(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ IoV3 io_scale(IoV3 a, float s) { return iov(a.x*s, a.y*s, a.z*s); }
__device__ __forceinline__ float io_dot(IoV3 a, IoV3 b) { return a.x*b.x + a.y*b.y + a.z*b.z; }
__device__ __forceinline__ IoV3 io_cross(IoV3 a, IoV3 b) { return iov(a.y*b.z - a.z*b.y, a.z*b.x - a.x*b.z, a.x*b.y - a.y*b.x); }

__device__ __forceinline__ void io_vertex_rel_pos(IoV3 q[3], IoV3 ns, float2 v0, float2 v1, float2 v2, float slength)
{
	unsigned j = 0;
	if (fabsf(ns.x) > fabsf(ns.y)) j = 1;
	if ((1 - j)*fabsf(ns.x) + j*fabsf(ns.y) > fabsf(ns.z)) j = 2;
	IoV3 c1 = iov(-((j == 1)*ns.z) + (j == 2)*ns.y, (j == 0)*ns.z - ((j == 2)*ns.x), -((j == 0)*ns.y) + (j == 1)*ns.x);
	c1 = io_scale(c1, 1.0f/sqrtf(io_dot(c1, c1)));
	const IoV3 c2 = io_cross(ns, c1);
	const float2 vp[3] = { v0, v1, v2 };
	const float inv = 1.0f/slength;
#pragma unroll
	for (int k = 0; k < 3; ++k) {
		const IoV3 s = iov(c1.x*vp[k].x + c2.x*vp[k].y, c1.y*vp[k].x + c2.y*vp[k].y, c1.z*vp[k].x + c2.z*vp[k].y);
		q[k] = io_scale(iov(-s.x, -s.y, -s.z), inv);
	}
}

__device__ __forceinline__ void io_mass_repartition(const IoV3 q[3], IoV3 n, float beta[3])
{
	const IoV3 v01 = io_sub(q[0], q[1]), v02 = io_sub(q[0], q[2]);
	IoV3 p0 = io_sub(q[0], io_scale(n, io_dot(q[0], n)));
	IoV3 p1 = io_sub(q[1], io_scale(n, io_dot(q[1], n)));
	IoV3 p2 = io_sub(q[2], io_scale(n, io_dot(q[2], n)));
	const float refSurface = (float)(0.5*(double)io_dot(io_cross(v01, v02), n));
	const IoV3 v21 = io_sub(q[2], q[1]);
	float s0 = (float)(0.5*(double)io_dot(io_cross(p2, v21), n));
	float s1 = (float)(0.5*(double)io_dot(io_cross(p0, v02), n));
	float s2 = (float)(-0.5*(double)io_dot(io_cross(p1, v01), n));
	if (s0 < 0.0f && s2 < 0.0f) { s0 = 0.0f; s1 = refSurface; s2 = 0.0f; }
	else if (s0 < 0.0f && s1 < 0.0f) { s0 = 0.0f; s1 = 0.0f; s2 = refSurface; }
	else if (s1 < 0.0f && s2 < 0.0f) { s0 = refSurface; s1 = 0.0f; s2 = 0.0f; }
	else if (s0 < 0.0f) {
		const float coef = (float)((double)s0/(0.5*(double)io_dot(io_cross(p0, v21), n)));
		p1 = io_sub(p1, io_scale(p0, coef));
		p0 = io_scale(p0, (float)(1.0 - (double)coef));
		s0 = 0.0f;
		s1 = (float)(0.5*(double)io_dot(io_cross(p0, v02), n));
		s2 = (float)(-0.5*(double)io_dot(io_cross(p1, v01), n));
	} else if (s1 < 0.0f) {
		const float coef = (float)((double)s1/(0.5*(double)io_dot(io_cross(p1, v02), n)));
		p2 = io_sub(p2, io_scale(p1, coef));
		p1 = io_scale(p1, (float)(1.0 - (double)coef));
		s0 = (float)(0.5*(double)io_dot(io_cross(p2, v21), n));
		s1 = 0.0f;
		s2 = (float)(-0.5*(double)io_dot(io_cross(p1, v01), n));
	} else if (s2 < 0.0f) {
		const float coef = (float)(-(double)s2/(0.5*(double)io_dot(io_cross(p2, v01), n)));
		p0 = io_sub(p0, io_scale(p2, coef));
		p2 = io_scale(p2, (float)(1.0 - (double)coef));
		s0 = (float)(0.5*(double)io_dot(io_cross(p2, v21), n));
		s1 = (float)(0.5*(double)io_dot(io_cross(p0, v02), n));
		s2 = 0.0f;
	}
	beta[0] = s0/refSurface; beta[1] = s1/refSurface; beta[2] = s2/refSurface;
}

struct SaIoOutArgs {
	const float4 *vel, *boundElement;
	const float2 *vertPos0, *vertPos1, *vertPos2;
	uint4 *vertices;
	float4 *gGam;
	float influenceradius;
};

__global__ void __launch_bounds__(128)
sa_find_outgoing_segment_kernel(DevParams p, SaIoArgs a, SaIoOutArgs o)
{
	const uint32_t index = blockIdx.x*128 + threadIdx.x;
	if (index >= a.numParticles) return;
	const particleinfo info = a.info[index];
	if (PART_TYPE(info) != PT_FLUID) return;
	const float4 pos = a.pos[index];
	if (!is_active_w(pos.w)) return;
	const uint4 mine = o.vertices[index];
	if (mine.x | mine.y) return;           // already marked ("this shouldn't happen", :1679-1686)
	const int3 gridPos = grid_pos_from_hash(p, a.hash[index] & CELLTYPE_BITMASK);
	const float4 vel = o.vel[index];
	float r2_min = o.influenceradius*o.influenceradius;
	uint32_t index_min = 0xFFFFFFFFu;
	IoV3 normal_min = iov(0.0f, 0.0f, 0.0f), relPos_min = normal_min;
	for_each_neib<PT_BOUNDARY>(p, a, index, pos, gridPos, [&](uint32_t j, const float4 &, float rx, float ry, float rz) {
		if (!IS_IO_BOUNDARY(a.info[j])) return;
		const float4 nrm = o.boundElement[j];
		const float4 nvel = o.vel[j];
		const IoV3 relPos = iov(rx, ry, rz), normal = iov(nrm.x, nrm.y, nrm.z);
		const IoV3 relVel = iov(vel.x - nvel.x, vel.y - nvel.y, vel.z - nvel.z);
		const float r2 = io_dot(relPos, relPos);
		// closer than the others, behind the element, moving out relative to it
		if (r2 < r2_min && io_dot(normal, relPos) <= 0.0f && io_dot(normal, relVel) < 0.0f) {
			r2_min = r2; index_min = j; normal_min = normal; relPos_min = relPos;
		}
	});
	if (index_min == 0xFFFFFFFFu) return;
	IoV3 vx[3];
	io_vertex_rel_pos(vx, normal_min, o.vertPos0[index_min], o.vertPos1[index_min], o.vertPos2[index_min], 1.0f);
	for (int k = 0; k < 3; ++k) vx[k] = io_sub(relPos_min, vx[k]);
	float beta[3];
	io_mass_repartition(vx, normal_min, beta);
	o.vertices[index] = o.vertices[index_min];
	o.gGam[index] = make_float4(beta[0], beta[1], beta[2], pos.w);      // the shares and the mass travel where grad gamma was
}

__global__ void __launch_bounds__(256)
sa_disable_outgoing_parts_kernel(float4 *pos, uint4 *vertices, const particleinfo *info, uint32_t numParticles)
{
	const uint32_t index = blockIdx.x*256 + threadIdx.x;
	if (index >= numParticles) return;
	if (PART_TYPE(info[index]) != PT_FLUID) return;
	float4 ps = pos[index];
	if (!is_active_w(ps.w)) return;
	const uint4 v = vertices[index];
	if ((v.x | v.y) != 0u) {
		ps.w = __uint_as_float(0x7fc00000u);      // disable_particle
		pos[index] = ps;
		vertices[index] = make_uint4(0u, 0u, 0u, 0u);
	}
}

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
static int sa_io_check(sphx_ctx *ctx, const char *who)
{
	if (!ctx || !ctx->have_params) return sphx_set_error(SPHX_ERR_INVALID, "sphx_sa_io: constants not set");
	if (ctx->params.boundarytype != SPHX_SA_BOUNDARY) return sphx_set_error(SPHX_ERR_INVALID, who);
	return SPHX_OK;
}

extern "C" int sphx_sa_identify_corner_vertices(sphx_ctx *ctx, const void *pos, void *info, const uint32_t *hash, const void *vertices,
	const uint32_t *cellStart, const uint16_t *neibsList, uint32_t numParticles, uint32_t particleRangeEnd, void *stream)
{
	(void)numParticles;
	int rc = sa_io_check(ctx, "saIdentifyCornerVertices called without SA_BOUNDARY");
	if (rc != SPHX_OK) return rc;
	SPHX_REQUIRE(pos && info && hash && vertices && cellStart && neibsList, "sphx_sa_identify_corner_vertices: missing buffer");
	if (!particleRangeEnd) return SPHX_OK;
	SaIoArgs a = { (const float4*)pos, hash, cellStart, neibsList, (const uint4*)vertices, (const particleinfo*)info, particleRangeEnd };
	SPHX_LAUNCH(sa_identify_corner_vertices_kernel, div_up_u(particleRangeEnd, 128), 128, (hipStream_t)stream, ctx->dev, a, (particleinfo*)info);
	SPHX_LAUNCH_CHECK("sa_identify_corner_vertices_kernel");
	return SPHX_OK;
}

extern "C" int sphx_sa_init_io_mass_vertex_count(sphx_ctx *ctx, const void *vertices, const uint32_t *hash, const void *info,
	const uint32_t *cellStart, const uint16_t *neibsList, void *forces, const void *pos,
	uint32_t numParticles, uint32_t particleRangeEnd, void *stream)
{
	(void)numParticles;
	int rc = sa_io_check(ctx, "initIOmass_vertexCount called without SA_BOUNDARY");
	if (rc != SPHX_OK) return rc;
	SPHX_REQUIRE(vertices && hash && info && cellStart && neibsList && forces && pos, "sphx_sa_init_io_mass_vertex_count: missing buffer");
	if (!particleRangeEnd) return SPHX_OK;
	SaIoArgs a = { (const float4*)pos, hash, cellStart, neibsList, (const uint4*)vertices, (const particleinfo*)info, particleRangeEnd };
	SPHX_LAUNCH(sa_init_io_mass_vertex_count_kernel, div_up_u(particleRangeEnd, 128), 128, (hipStream_t)stream, ctx->dev, a, (float4*)forces);
	SPHX_LAUNCH_CHECK("sa_init_io_mass_vertex_count_kernel");
	return SPHX_OK;
}

extern "C" int sphx_sa_init_io_mass(sphx_ctx *ctx, const void *oldPos, const void *forces, const void *vertices, const uint32_t *hash,
	const void *info, const uint32_t *cellStart, const uint16_t *neibsList, void *newPos,
	uint32_t numParticles, uint32_t particleRangeEnd, float deltap, void *stream)
{
	(void)numParticles;
	int rc = sa_io_check(ctx, "initIOmass called without SA_BOUNDARY");
	if (rc != SPHX_OK) return rc;
	SPHX_REQUIRE(oldPos && forces && vertices && hash && info && cellStart && neibsList && newPos && oldPos != newPos,
		"sphx_sa_init_io_mass: missing buffer (newPos must not be oldPos)");
	if (!particleRangeEnd) return SPHX_OK;
	SaIoArgs a = { (const float4*)oldPos, hash, cellStart, neibsList, (const uint4*)vertices, (const particleinfo*)info, particleRangeEnd };
	SPHX_LAUNCH(sa_init_io_mass_kernel, div_up_u(particleRangeEnd, 128), 128, (hipStream_t)stream, ctx->dev, a, (const float4*)forces,
		(float4*)newPos, deltap);
	SPHX_LAUNCH_CHECK("sa_init_io_mass_kernel");
	return SPHX_OK;
}

extern "C" int sphx_sa_find_outgoing_segment(sphx_ctx *ctx, const void *pos, const void *vel, void *vertices, void *gGam,
	const void *vertPos0, const void *vertPos1, const void *vertPos2, const void *boundElements, const void *info,
	const uint32_t *hash, const uint32_t *cellStart, const uint16_t *neibsList,
	uint32_t numParticles, uint32_t particleRangeEnd, float influenceradius, void *stream)
{
	(void)numParticles;
	int rc = sa_io_check(ctx, "findOutgoingSegment called without SA_BOUNDARY");
	if (rc != SPHX_OK) return rc;
	SPHX_REQUIRE(pos && vel && vertices && gGam && vertPos0 && vertPos1 && vertPos2 && boundElements && info && hash && cellStart && neibsList,
		"sphx_sa_find_outgoing_segment: missing buffer");
	if (!particleRangeEnd) return SPHX_OK;
	SaIoArgs a = { (const float4*)pos, hash, cellStart, neibsList, (const uint4*)vertices, (const particleinfo*)info, particleRangeEnd };
	SaIoOutArgs o = { (const float4*)vel, (const float4*)boundElements, (const float2*)vertPos0, (const float2*)vertPos1,
		(const float2*)vertPos2, (uint4*)vertices, (float4*)gGam, influenceradius };
	SPHX_LAUNCH(sa_find_outgoing_segment_kernel, div_up_u(particleRangeEnd, 128), 128, (hipStream_t)stream, ctx->dev, a, o);
	SPHX_LAUNCH_CHECK("sa_find_outgoing_segment_kernel");
	return SPHX_OK;
}

extern "C" int sphx_sa_disable_outgoing_parts(sphx_ctx *ctx, void *pos, void *vertices, const void *info, uint32_t numParticles, void *stream)
{
	int rc = sa_io_check(ctx, "disableOutgoingParts called without SA_BOUNDARY");
	if (rc != SPHX_OK) return rc;
	SPHX_REQUIRE(pos && vertices && info, "sphx_sa_disable_outgoing_parts: missing buffer");
	if (!numParticles) return SPHX_OK;
	SPHX_LAUNCH(sa_disable_outgoing_parts_kernel, div_up_u(numParticles, 256), 256, (hipStream_t)stream, (float4*)pos, (uint4*)vertices,
		(const particleinfo*)info, numParticles);
	SPHX_LAUNCH_CHECK("sa_disable_outgoing_parts_kernel");
	return SPHX_OK;
}

// ==========================================================================================
// The boundary-condition passes with open boundaries (saSegmentBoundaryConditionsDevice / saVertexBoundaryConditionsDevice with
// has_io, _kernel.cu:1427-1520, 2197-2252; laminar, not repacking, Wendland).  WRITTEN AT THE END OF ROUND 4 AND NOT YET RUN ON
// A GPU: ports of the oracle's orc_sa_segment_bc_io / orc_sa_vertex_bc_io (which are held by known answers and by a whole
// open-channel run on the CPU, tests/test_sa_io_oracle.py); their GPU parity test is in tests/test_gpu_sa_io.py behind
// SPHX_TEST_SA_IO_BC=1 until it has passed once.  Nothing in the engines calls them yet.
// ==========================================================================================
__device__ __forceinline__ float io_eos_rho(const DevParams &p, float pres, uint32_t fl)      // RHO, phys_core.cu:106-112
{ return powf(pres/p.bcoeff[fl] + 1.0f, 1.0f/p.gammacoeff[fl]) - 1.0f; }
__device__ __forceinline__ float io_R(const DevParams &p, float rho_tilde, uint32_t fl)       // Riemann celerity, :114-120
{ return 2.0f/(p.gammacoeff[fl] - 1.0f)*p.sscoeff[fl]*powf(rho_tilde + 1.0f, 0.5f*p.gammacoeff[fl] - 0.5f); }
__device__ __forceinline__ float io_RHOR(const DevParams &p, float r, uint32_t fl)            // its inverse, :122-127 (double constants)
{ return (float)((double)powf((float)(((double)p.gammacoeff[fl] - 1.)*(double)r/(2.*(double)p.sscoeff[fl])), (float)(2./((double)p.gammacoeff[fl] - 1.))) - 1.0); }

// calculateIOboundaryCondition, _kernel.cu:111-200
__device__ __forceinline__ void io_boundary_condition(const DevParams &p, float4 &eulerVel, bool velocity_driven, uint32_t a,
	float rhoInt, float rhoExt, float ux, float uy, float uz, float unInt, float unExt, float nx, float ny, float nz)
{
	const float rInt = io_R(p, rhoInt, a);
	if (velocity_driven) {
		float riemannR = 0.0f;
		if (unExt <= unInt)
			riemannR = rInt + (unExt - unInt);
		else {
			const float riemannRho = io_eos_rho(p, sa_P(p, rhoInt, a) + ((rhoInt + 1.0f)*p.rho0[a])*unInt*(unInt - unExt), a);
			riemannR = io_R(p, riemannRho, a);
			const float lambda = unExt + sa_sound_speed(p, riemannRho, a);
			const float lambdaInt = unInt + sa_sound_speed(p, rhoInt, a);
			if (lambda <= lambdaInt) riemannR = rInt;
		}
		eulerVel.w = io_RHOR(p, riemannR, a);
	} else {
		float flux = 0.0f;
		const float cExt = sa_sound_speed(p, rhoExt, a), cInt = sa_sound_speed(p, rhoInt, a);
		const float lambdaInt = unInt + cInt;
		const float rExt = io_R(p, rhoExt, a);
		const float shock = (sa_P(p, rhoInt, a) - sa_P(p, rhoExt, a))/(((rhoInt + 1.0f)*p.rho0[a])*fmaxf(unInt, 1e-5f*p.sscoeff[a])) + unInt;
		if (rhoExt <= rhoInt) {
			flux = unInt + (rExt - rInt);
			float lambda = flux + cExt;
			if (lambda > lambdaInt) {
				flux = shock;
				if (fabsf(flux) > p.sscoeff[a]*0.1f) flux = unInt;
				lambda = flux + cExt;
				if (lambda <= lambdaInt) flux = unInt;
			}
		} else {
			flux = shock;
			if (fabsf(flux) > p.sscoeff[a]*0.1f) flux = unInt;
			float lambda = flux + cExt;
			if (lambda <= lambdaInt) {
				flux = unInt + (rExt - rInt);
				lambda = flux + cExt;
				if (lambda > lambdaInt) flux = unInt;
			}
		}
		eulerVel.x = eulerVel.y = eulerVel.z = 0.0f;
		if (rhoExt < 0.0f) flux = fminf(flux, 0.0f);
		if (flux < 0.0f) {
			const float un = ux*nx + uy*ny + uz*nz;
			eulerVel.x = ux - un*nx; eulerVel.y = uy - un*ny; eulerVel.z = uz - un*nz;
		}
		eulerVel.x += nx*flux; eulerVel.y += ny*flux; eulerVel.z += nz*flux;
		eulerVel.w = rhoExt;
	}
}

struct SaIoBcArgs {
	float4 *vel, *gGam, *eulerVel;          // in place (boundary / vertex rows; the clones' rows in the last vertex pass)
	const float4 *pos;                      // the walker's position rows
	const float4 *boundElementRO;           // segment pass: read only
	const uint4 *verticesRO;
	const particleinfo *infoRO;
	const uint32_t *hash, *cellStart;
	const neibdata *neibsList;
	uint32_t numParticles;
	int step;
	// vertex pass
	float4 *newPos, *forces, *boundElement;
	uint4 *vertices;
	particleinfo *info;
	uint32_t *hashW, *nextIDs, *newNumParticles;
	const float2 *vertPos0, *vertPos1, *vertPos2;
	uint32_t totParticles, numOpenVertices;
	float deltap, dt;
	// what the walker reads
	const uint32_t *hashRO;
};

struct IoNdata { float r, w, press; float4 vel; };
__device__ __forceinline__ IoNdata io_fluid_ndata(const DevParams &p, const float4 *vel, const particleinfo *info, uint32_t j,
	float rx, float ry, float rz, float mass)
{
	IoNdata n;
	const uint32_t nfl = FLUID_NUM(info[j]);
	n.vel = vel[j];
	n.r = sqrtf(rx*rx + ry*ry + rz*rz);
	n.w = kernel_W<SPHX_WENDLAND>(p, n.r)*mass/((n.vel.w + 1.0f)*p.rho0[nfl]);
	n.press = p.bcoeff[nfl]*(powf(n.vel.w + 1.0f, p.gammacoeff[nfl]) - 1.0f);
	return n;
}

struct IoWalk {       // what for_each_neib needs
	const float4 *pos; const uint32_t *cellStart; const neibdata *neibsList;
};

__global__ void __launch_bounds__(128)
sa_segment_bc_io_kernel(DevParams p, SaIoBcArgs a)
{
	const uint32_t index = blockIdx.x*128 + threadIdx.x;
	if (index >= a.numParticles) return;
	const particleinfo info = a.infoRO[index];
	if (!IS_BOUNDARY(info)) return;
	const IoWalk wk = { a.pos, a.cellStart, a.neibsList };
	const float4 pos = a.pos[index];
	const float4 normal = a.boundElementRO[index];
	const uint4 verts = a.verticesRO[index];
	const int3 gridPos = grid_pos_from_hash(p, a.hashRO[index] & CELLTYPE_BITMASK);
	const bool has_moving = (p.simflags & SPHX_ENABLE_MOVING_BODIES) != 0;
	const bool io = IS_IO_BOUNDARY(info) != 0, vdriven = (info.x & FG_VELOCITY_DRIVEN) != 0;
	float sumpWall = 0.0f, shepard_div = 0.0f, sump = 0.0f, svx = 0.0f, svy = 0.0f, svz = 0.0f;
	float4 gGam = make_float4(0.0f, 0.0f, 0.0f, a.gGam[index].w);
	float4 vel = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
	const bool calcGam = has_moving || !is_active_w(gGam.w) || a.step == 0;
	if (calcGam) gGam.w = 0.0f;
	const bool moving = has_moving && (info.x & FG_MOVING_BOUNDARY);
	float4 eulerVel = make_float4(0.0f, 0.0f, 0.0f, 0.0f);          // eulervel_pout, IO constructor (:488-505)
	if (io) { eulerVel = a.eulerVel[index]; if (vdriven) eulerVel.w = 0.0f; }
	for_each_neib<PT_VERTEX>(p, wk, index, pos, gridPos, [&](uint32_t j, const float4 &npos, float, float, float) {
		if (!is_active_w(npos.w)) return;
		if (!io_has_vertex(verts, info_id(a.infoRO[j]))) return;
		if (moving) { const float4 nv = a.vel[j]; vel.x += nv.x; vel.y += nv.y; vel.z += nv.z; }
		if (calcGam) { const float4 g = a.gGam[j]; gGam.x += g.x; gGam.y += g.y; gGam.z += g.z; gGam.w += g.w; }
	});
	if (calcGam) {
		const float inv = 1.0f/3;
		gGam.x *= inv; gGam.y *= inv; gGam.z *= inv; gGam.w *= inv;
		a.gGam[index] = gGam;
		gGam.w = fmaxf(gGam.w, 1e-5f);
	}
	vel.x /= 3; vel.y /= 3; vel.z /= 3;
	const uint32_t fl = FLUID_NUM(info);
	for_each_neib<PT_FLUID>(p, wk, index, pos, gridPos, [&](uint32_t j, const float4 &npos, float rx, float ry, float rz) {
		if (!is_active_w(npos.w)) return;
		const IoNdata n = io_fluid_ndata(p, a.vel, a.infoRO, j, rx, ry, rz, npos.w);
		if (!(n.r < p.influenceradius && (normal.x*rx + normal.y*ry + normal.z*rz) < 0.0f)) return;
		const float gdot = p.gravity[0]*rx + p.gravity[1]*ry + p.gravity[2]*rz;
		sumpWall += fmaxf(n.press + ((n.vel.w + 1.0f)*p.rho0[fl])*gdot, 0.0f)*n.w;
		if (io) {         // io_fluid_contrib, segments (:852-866)
			const float4 ne = a.eulerVel[j];
			svx += n.w*(n.vel.x + ne.x); svy += n.w*(n.vel.y + ne.y); svz += n.w*(n.vel.z + ne.z);
			sump += n.w*fmaxf(0.0f, n.press);
		}
		shepard_div += n.w;
	});
	if (io) {             // impose_io_bc (:1362-1413)
		if (shepard_div > 0.1f*gGam.w) {
			svx /= shepard_div; svy /= shepard_div; svz /= shepard_div;
			sump /= shepard_div;
			vel.w = io_eos_rho(p, sump, fl);
			if (!vdriven) a.eulerVel[index] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
		} else {
			sump = 0.0f;
			if (vdriven) { svx = eulerVel.x; svy = eulerVel.y; svz = eulerVel.z; vel.w = 0.0f; }
			else { svx = svy = svz = 0.0f; vel.w = a.eulerVel[index].w; }
		}
		const float unInt = svx*normal.x + svy*normal.y + svz*normal.z;
		const float unExt = eulerVel.x*normal.x + eulerVel.y*normal.y + eulerVel.z*normal.z;
		float4 ev = eulerVel;
		io_boundary_condition(p, ev, vdriven, fl, vel.w, eulerVel.w, svx, svy, svz, unInt, unExt, normal.x, normal.y, normal.z);
		a.eulerVel[index] = ev;
		vel.w = ev.w;
	} else {              // impose_solid_bc<true>, impose_solid_eulerVel
		shepard_div = fmaxf(shepard_div, 0.1f*gGam.w);
		vel.w = io_eos_rho(p, sumpWall/shepard_div, fl);
		a.eulerVel[index] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
	}
	a.vel[index] = vel;
}

__global__ void __launch_bounds__(128)
sa_vertex_bc_io_kernel(DevParams p, SaIoBcArgs a)
{
	const uint32_t index = blockIdx.x*128 + threadIdx.x;
	if (index >= a.numParticles) return;
	const particleinfo info = a.info[index];
	if (PART_TYPE(info) != PT_VERTEX) return;
	const IoWalk wk = { a.pos, a.cellStart, a.neibsList };
	const float4 pos = a.pos[index];
	const float gam = a.gGam[index].w;
	const uint32_t fl = FLUID_NUM(info), my_id = info_id(info);
	const bool io = IS_IO_BOUNDARY(info) != 0, corner = IS_CORNER(info) != 0, vdriven = (info.x & FG_VELOCITY_DRIVEN) != 0;
	const float4 normal = a.boundElement[index];
	const float refMass = a.deltap*a.deltap*a.deltap*p.rho0[fl];
	const int3 gridPos = grid_pos_from_hash(p, a.hashW[index] & CELLTYPE_BITMASK);
	float sumpWall = 0.0f, shepard_div = 0.0f, sump = 0.0f, sumMdot = 0.0f, massFluid = 0.0f, svx = 0.0f, svy = 0.0f, svz = 0.0f;
	for_each_neib<PT_FLUID>(p, wk, index, pos, gridPos, [&](uint32_t j, const float4 &npos, float rx, float ry, float rz) {
		if (!is_active_w(npos.w)) return;
		const IoNdata n = io_fluid_ndata(p, a.vel, a.info, j, rx, ry, rz, npos.w);
		if (!(n.r < p.influenceradius)) return;
		const float gdot = p.gravity[0]*rx + p.gravity[1]*ry + p.gravity[2]*rz;
		sumpWall += fmaxf(n.press + ((n.vel.w + 1.0f)*p.rho0[fl])*gdot, 0.0f)*n.w;
		shepard_div += n.w;
		if (!io) return;      // io_fluid_contrib, vertices (:868-909)
		if (!corner) {
			const float4 ne = a.eulerVel[j];
			svx += n.w*(n.vel.x + ne.x); svy += n.w*(n.vel.y + ne.y); svz += n.w*(n.vel.z + ne.z);
			sump += n.w*fmaxf(0.0f, n.press);
		}
		if (a.step == 2) {    // a particle marked by findOutgoingSegment: its mass, by this vertex's share
			const uint4 nv = a.vertices[j];
			if ((nv.x | nv.y) != 0u) {
				const float4 w = a.gGam[j];
				const float weight = nv.x == my_id ? w.x : nv.y == my_id ? w.y : nv.z == my_id ? w.z : 0.0f;
				if (weight > 0) massFluid += weight*w.w;
			}
		}
	});
	if (io && !corner) {      // vertex_boundary_loop / io_boundary_contrib (:937-988): the mass flux through the adjacent open segments
		for_each_neib<PT_BOUNDARY>(p, wk, index, pos, gridPos, [&](uint32_t j, const float4 &, float, float, float) {
			const uint4 nv = a.vertices[j];
			if (!io_has_vertex(nv, my_id)) return;
			const particleinfo ninfo = a.info[j];
			if (!IS_IO_BOUNDARY(ninfo)) return;
			const float4 nn = a.boundElement[j];
			IoV3 vx[3];
			io_vertex_rel_pos(vx, iov(nn.x, nn.y, nn.z), a.vertPos0[j], a.vertPos1[j], a.vertPos2[j], -1.0f);
			float beta[3];
			io_mass_repartition(vx, iov(nn.x, nn.y, nn.z), beta);
			const float weight = nv.x == my_id ? beta[0] : nv.y == my_id ? beta[1] : nv.z == my_id ? beta[2] : 0.0f;
			const float4 ne = a.eulerVel[j];
			sumMdot += ((a.vel[j].w + 1.0f)*p.rho0[FLUID_NUM(ninfo)])*nn.w*weight*(ne.x*nn.x + ne.y*nn.y + ne.z*nn.z);
		});
	}
	shepard_div = fmaxf(shepard_div, 0.1f*gam);
	a.vel[index].w = io_eos_rho(p, sumpWall/shepard_div, fl);
	if (!io || corner) return;
	// impose_vertex_io_bc (:1168-1252)
	float4 eulerVel = a.eulerVel[index];
	if (shepard_div > 0.1f*gam) {
		svx /= shepard_div; svy /= shepard_div; svz /= shepard_div;
		sump /= shepard_div;
		const float unInt = svx*normal.x + svy*normal.y + svz*normal.z;
		const float unExt = eulerVel.x*normal.x + eulerVel.y*normal.y + eulerVel.z*normal.z;
		const float rhoInt = io_eos_rho(p, sump, fl);
		io_boundary_condition(p, eulerVel, vdriven, fl, rhoInt, eulerVel.w, svx, svy, svz, unInt, unExt, normal.x, normal.y, normal.z);
	} else if (vdriven)
		eulerVel.w = 0.0f;
	else
		eulerVel.x = eulerVel.y = eulerVel.z = 0.0f;
	a.eulerVel[index] = eulerVel;
	a.vel[index].w = eulerVel.w;
	float4 np = pos;
	const float un = normal.x*eulerVel.x + normal.y*eulerVel.y + normal.z*eulerVel.z;
	if (a.step != 0) {
		np.w += a.dt*sumMdot;
		if (shepard_div < 0.1f*gam && sumMdot < 0.0f) np.w = 0.0f;
		np.w = fmaxf(-2.0f*refMass, fminf(2.0f*refMass, np.w));
		if (sumMdot < 0.0f || un < 1e-5f*p.sscoeff[fl]) {
			const float weightedMass = refMass*normal.w;      // normal.w of a vertex is NaN: fminf / fmaxf return the other operand
			np.w = fmaxf(-weightedMass, fminf(weightedMass, np.w));
		}
	}
	// generate_new_particles (:1101-1159), createNewFluidParticle (:73-104)
	if (a.step == 2 && np.w > refMass*0.5f && sumMdot > 0 && un > 1e-5f && (vdriven || eulerVel.w > 1e-5f)) {
		const uint32_t clone = atomicAdd(a.newNumParticles, 1u);
		if (clone < a.totParticles) {
			const uint32_t new_id = a.nextIDs[index];
			a.nextIDs[index] = new_id + a.numOpenVertices;
			particleinfo ci;
			ci.x = PT_FLUID; ci.y = (unsigned short)(fl << 12); ci.z = (unsigned short)(new_id & 0xFFFFu); ci.w = (unsigned short)(new_id >> 16);
			float4 cp = np;
			cp.w = refMass;
			massFluid -= cp.w;
			a.newPos[clone] = cp;
			a.info[clone] = ci;
			a.hashW[clone] = a.hashW[index] & CELLTYPE_BITMASK;
			a.vel[clone] = eulerVel;
			a.gGam[clone] = a.gGam[index];
			a.eulerVel[clone] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
			a.forces[clone] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
			a.vertices[clone] = make_uint4(0u, 0u, 0u, 0u);
			a.nextIDs[clone] = 0xFFFFFFFFu;
			const float nanv = __uint_as_float(0xffc00000u);      // -NAN
			a.boundElement[clone] = make_float4(nanv, nanv, nanv, nanv);
		}
	}
	np.w += massFluid;
	a.newPos[index] = np;
}

static int sa_io_bc_check(sphx_ctx *ctx, const char *who)
{
	int rc = sa_io_check(ctx, who);
	if (rc != SPHX_OK) return rc;
	if (ctx->params.kerneltype != SPHX_WENDLAND)
		return sphx_set_error(SPHX_ERR_UNSUPPORTED, "sphx_sa_io: SA_BOUNDARY is built for the Wendland kernel");
	if (ctx->dev.turbmodel == SPHX_KEPSILON)
		return sphx_set_error(SPHX_ERR_UNSUPPORTED, "sphx_sa_io: open boundaries with k-epsilon are not built");
	return SPHX_OK;
}

extern "C" int sphx_sa_segment_bc_io(sphx_ctx *ctx, void *vel, void *gGam, void *eulerVel, const void *pos, const void *vertices,
	const void *boundElements, const void *info, const uint32_t *hash, const uint32_t *cellStart, const uint16_t *neibsList,
	uint32_t numParticles, uint32_t particleRangeEnd, int step, void *stream)
{
	(void)numParticles;
	int rc = sa_io_bc_check(ctx, "saSegmentBoundaryConditions called without SA_BOUNDARY");
	if (rc != SPHX_OK) return rc;
	SPHX_REQUIRE(vel && gGam && eulerVel && pos && vertices && boundElements && info && hash && cellStart && neibsList,
		"sphx_sa_segment_bc_io: missing buffer");
	if (!particleRangeEnd) return SPHX_OK;
	SaIoBcArgs a = {};
	a.vel = (float4*)vel; a.gGam = (float4*)gGam; a.eulerVel = (float4*)eulerVel; a.pos = (const float4*)pos;
	a.boundElementRO = (const float4*)boundElements; a.verticesRO = (const uint4*)vertices; a.infoRO = (const particleinfo*)info;
	a.hashRO = hash; a.cellStart = cellStart; a.neibsList = neibsList; a.numParticles = particleRangeEnd; a.step = step == -1 ? 0 : step;
	SPHX_LAUNCH(sa_segment_bc_io_kernel, div_up_u(particleRangeEnd, 128), 128, (hipStream_t)stream, ctx->dev, a);
	SPHX_LAUNCH_CHECK("sa_segment_bc_io_kernel");
	return SPHX_OK;
}

extern "C" int sphx_sa_vertex_bc_io(sphx_ctx *ctx, void *vel, const void *pos, void *newPos, void *gGam, void *eulerVel, void *forces,
	void *vertices, void *boundElements, const void *vertPos0, const void *vertPos1, const void *vertPos2, void *info, uint32_t *hash,
	uint32_t *nextIDs, uint32_t *newNumParticles, const uint32_t *cellStart, const uint16_t *neibsList,
	uint32_t numParticles, uint32_t particleRangeEnd, uint32_t totParticles, float deltap, float dt, int step,
	uint32_t numOpenVertices, void *stream)
{
	(void)numParticles;
	int rc = sa_io_bc_check(ctx, "saVertexBoundaryConditions called without SA_BOUNDARY");
	if (rc != SPHX_OK) return rc;
	SPHX_REQUIRE(vel && pos && newPos && gGam && eulerVel && forces && vertices && boundElements && vertPos0 && vertPos1 &&
		vertPos2 && info && hash && nextIDs && newNumParticles && cellStart && neibsList, "sphx_sa_vertex_bc_io: missing buffer");
	if (!particleRangeEnd) return SPHX_OK;
	// pos == newPos (the reference's call: the read and the write list hold the same array) is fine: a thread writes its own row,
	// xyz unchanged, and reads the masses of fluid rows only, which the pass does not write
	SaIoBcArgs a = {};
	a.vel = (float4*)vel; a.gGam = (float4*)gGam; a.eulerVel = (float4*)eulerVel; a.pos = (const float4*)pos; a.newPos = (float4*)newPos;
	a.forces = (float4*)forces; a.vertices = (uint4*)vertices; a.boundElement = (float4*)boundElements; a.info = (particleinfo*)info;
	a.hashW = hash; a.nextIDs = nextIDs; a.newNumParticles = newNumParticles; a.cellStart = cellStart; a.neibsList = neibsList;
	a.vertPos0 = (const float2*)vertPos0; a.vertPos1 = (const float2*)vertPos1; a.vertPos2 = (const float2*)vertPos2;
	a.numParticles = particleRangeEnd; a.totParticles = totParticles; a.numOpenVertices = numOpenVertices;
	a.deltap = deltap; a.dt = dt; a.step = step == -1 ? 0 : step;
	SPHX_LAUNCH(sa_vertex_bc_io_kernel, div_up_u(particleRangeEnd, 128), 128, (hipStream_t)stream, ctx->dev, a);
	SPHX_LAUNCH_CHECK("sa_vertex_bc_io_kernel");
	return SPHX_OK;
}

// ==========================================================================================
// Density summation and forces with open boundaries (density_sum_kernel.cu:119-140,206-250,374-420,606-655;
// forces_kernel.def:1485-1497,2494-2507,2703-2708), one thread per particle over the list -- the list-walker kernels of
// sa_bounds.hip (sa_density_sum_kernel, sa_forces_kernel without k-epsilon) with the open boundaries' terms, as the oracle's
// orc_sa_density_sum_io / orc_forces_sa_io have them.  WRITTEN AT THE END OF ROUND 4, NOT YET RUN ON A GPU (see above); the
// tiled fast paths do not know open boundaries, so these are the whole pass for such a run.
// ==========================================================================================
#include "sa_wall_gamma.h"
#include "sa_args.h"

struct SaIoDensitySumArgs {
	float4 *newVel, *newGGam, *forces;
	const float4 *oldPos, *pos, *oldVel, *oldEulerVel, *oldGGam, *boundElement;
	const float2 *vertPos[3];
	const particleinfo *info;
	const uint32_t *hash, *cellStart;
	const neibdata *neibsList;
	uint32_t numParticles;
	float dt;
};

__global__ void __launch_bounds__(128)
sa_density_sum_io_kernel(DevParams p, SaIoDensitySumArgs a)
{
	const uint32_t index = blockIdx.x*128 + threadIdx.x;
	if (index >= a.numParticles) return;
	const particleinfo info = a.info[index];
	if (PART_TYPE(info) != PT_FLUID) {
		if (PART_TYPE(info) == PT_VERTEX || PART_TYPE(info) == PT_BOUNDARY) a.newGGam[index] = a.oldGGam[index];
		return;
	}
	const float4 posN = a.oldPos[index], posNp1 = a.pos[index];
	const int3 gridPos = grid_pos_from_hash(p, a.hash[index] & CELLTYPE_BITMASK);
	const float dx = posNp1.x - posN.x, dy = posNp1.y - posN.y, dz = posNp1.z - posN.z;
	const IoWalk w = { a.oldPos, a.cellStart, a.neibsList };      // r_ab at step n: the walker reads the OLD positions
	float sumPmwN = 0.0f, sumPmwNp1 = 0.0f, sumVmwDelta = 0.0f;
	auto volumic = [&](uint32_t j, const float4 &nN, float pcx, float pcy, float pcz) {
		if (!is_active_w(nN.w)) return;
		const particleinfo ninfo = a.info[j];
		const float4 nNp1 = a.pos[j];
		const float rx = pcx - nN.x, ry = pcy - nN.y, rz = pcz - nN.z;
		const float qx = (pcx - nNp1.x) + dx, qy = (pcy - nNp1.y) + dy, qz = (pcz - nNp1.z) + dz;
		if (!IS_IO_BOUNDARY(ninfo)) {
			const float rN = sqrtf(rx*rx + ry*ry + rz*rz);
			sumPmwN -= nN.w*kernel_W<SPHX_WENDLAND>(p, rN);
		}
		const float rNp1 = sqrtf(qx*qx + qy*qy + qz*qz);
		if (rNp1 < p.influenceradius) sumPmwNp1 += nN.w*kernel_W<SPHX_WENDLAND>(p, rNp1);
		if (IS_IO_BOUNDARY(ninfo)) {      // densitySumOpenBoundaryContribution: the neighbour displaced by its Eulerian velocity
			const float4 e = a.oldEulerVel[j], v = a.oldVel[j];
			const float ex = rx + a.dt*(e.x - v.x), ey = ry + a.dt*(e.y - v.y), ez = rz + a.dt*(e.z - v.z);
			const float newDist = sqrtf(ex*ex + ey*ey + ez*ez);
			if (newDist < p.influenceradius) sumVmwDelta -= nN.w*kernel_W<SPHX_WENDLAND>(p, newDist);
		}
	};
	for_each_neib<PT_FLUID, true>(p, w, index, posN, gridPos, volumic);
	for_each_neib<PT_VERTEX, true>(p, w, index, posN, gridPos, volumic);
	const float fw = sumPmwNp1 + sumPmwN + sumVmwDelta;
	a.forces[index].w = fw;
	float gGamDotR = 0.0f, sumSgamDelta = 0.0f, sumSgamN = 0.0f;
	V3 gGam = v3(0.0f, 0.0f, 0.0f);
	for_each_neib<PT_BOUNDARY, true>(p, w, index, posN, gridPos, [&](uint32_t j, const float4 &nN, float pcx, float pcy, float pcz) {
		if (!is_active_w(nN.w)) return;
		const float4 nNp1 = a.pos[j];
		const float inv = 1.0f/p.slength;
		const V3 qN = v3((pcx - nN.x)*inv, (pcy - nN.y)*inv, (pcz - nN.z)*inv);
		const V3 qNp1 = v3(((pcx - nNp1.x) + dx)*inv, ((pcy - nNp1.y) + dy)*inv, ((pcz - nNp1.z) + dz)*inv);
		const float4 be = a.boundElement[j];
		const V3 ns = v3(be.x, be.y, be.z);
		WallTri tri;
		wall_tri_setup(tri, ns, a.vertPos[0][j], a.vertPos[1][j], a.vertPos[2][j], p.slength);
		const V3 gN = ns*(wall_grad_gamma(tri, qN)/p.slength);
		const V3 gNp1 = ns*(wall_grad_gamma(tri, qNp1)/p.slength);
		gGamDotR += 0.5f*dot(gN + gNp1, qNp1 - qN);
		gGam = gGam + gNp1;
		if (IS_IO_BOUNDARY(a.info[j])) {      // io_gamma_contrib (:374-395)
			const float4 e = a.oldEulerVel[j], v = a.oldVel[j];
			const V3 deltaR = v3(a.dt*(e.x - v.x), a.dt*(e.y - v.y), a.dt*(e.z - v.z));
			const V3 qDelta = qN + deltaR/p.slength;
			const V3 gDelta = ns*(wall_grad_gamma(tri, qDelta)/p.slength);
			sumSgamDelta += dot(deltaR, gDelta);
			sumSgamN += dot(deltaR, gN);
		}
	});
	gGamDotR *= p.slength;
	const float4 gGamN = a.oldGGam[index];
	float4 g = make_float4(gGam.x, gGam.y, gGam.z, gGamN.w + gGamDotR);
	float imposedGam = gGamN.w + (sumSgamDelta + sumSgamN)/2.0f;      // compute_imposed_gamma (:404-417)
	if (imposedGam > 1.0f) imposedGam = 1.0f;
	else if (imposedGam < 0.1f) imposedGam = 0.1f;
	const uint32_t fl = FLUID_NUM(info);
	const float rho = (imposedGam*((a.oldVel[index].w + 1.0f)*p.rho0[fl]) + fw)/g.w;
	if (g.w > 1.0f || sqrtf(g.x*g.x + g.y*g.y + g.z*g.z)*p.slength < 1e-10f) g.w = 1.0f;
	else if (g.w < 0.1f) g.w = 0.1f;
	a.newVel[index].w = rho/p.rho0[fl] - 1.0f;
	a.newGGam[index] = g;
}

struct SaIoForcesArgs {
	float4 *forces;
	float *cfl, *cflGamma, *cflGammaBlocks;
	const float4 *pos, *vel, *eulerVel, *gGam, *boundElement;
	const float2 *vertPos[3];
	const particleinfo *info;
	const uint32_t *hash, *cellStart;
	const neibdata *neibsList;
	uint32_t fromParticle, toParticle, cflOffset;
	float deltap;
};

__global__ void __launch_bounds__(SPHX_BLOCK_FORCES)
sa_forces_io_kernel(DevParams p, SaIoForcesArgs a)
{
	__shared__ float sMax[SPHX_BLOCK_FORCES/64], sMaxG[SPHX_BLOCK_FORCES/64];
	const uint32_t index = blockIdx.x*SPHX_BLOCK_FORCES + threadIdx.x + a.fromParticle;
	float cflTerm = 0.0f, gammaCfl = 0.0f;
	if (index < a.toParticle) {
		const particleinfo info = a.info[index];
		const float4 pos = a.pos[index];
		if (is_active_w(pos.w)) {
			float4 force = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
			const float4 vel = a.vel[index];
			const uint32_t fl = FLUID_NUM(info);
			if (PART_TYPE(info) == PT_FLUID) {
				const IoWalk wk = { a.pos, a.cellStart, a.neibsList };
				const int3 gridPos = grid_pos_from_hash(p, a.hash[index] & CELLTYPE_BITMASK);
				const float p_rho = (vel.w + 1.0f)*p.rho0[fl];
				const float p_precalc = sa_P(p, vel.w, fl)/(p_rho*p_rho);
				const float4 p_euler = a.eulerVel[index];
				const bool density_sum = (p.simflags & SPHX_ENABLE_DENSITY_SUM) != 0;
				const bool newtonian = p.rheology == SPHX_NEWTONIAN;
				// fluid <- fluid (VERT false) and fluid <- vertex (VERT true: the viscous term sees relVel + relEulerVel, :2494-2507)
				auto particle_pair = [&](bool VERT, uint32_t j, const float4 &npos, float rx, float ry, float rz) {
					if (!is_active_w(npos.w)) return;
					const float r = sqrtf(fmaf(rz, rz, fmaf(ry, ry, rx*rx)));
					if (r >= p.influenceradius) return;
					const float4 nvel = a.vel[j];
					const float vx = vel.x - nvel.x, vy = vel.y - nvel.y, vz = vel.z - nvel.z;
					const float vel_dot_pos = sa_dot3(vx, vy, vz, rx, ry, rz);
					const float qm2 = r/p.slength - 2.0f;
					const float f = qm2*qm2*qm2*p.fcoeff;
					const uint32_t nfl = FLUID_NUM(a.info[j]);
					const float n_rho = (nvel.w + 1.0f)*p.rho0[nfl];
					const float n_precalc = sa_P(p, nvel.w, nfl)/(n_rho*n_rho);
					const float nmass = npos.w;
					if (!density_sum) force.w += nmass*vel_dot_pos*f;
					const float s = (p_precalc + n_precalc)*nmass*f;
					float dx = 0.0f, dy = 0.0f, dz = 0.0f;
					dx -= s*rx; dy -= s*ry; dz -= s*rz;
					if (newtonian) {
						float wx = vx, wy = vy, wz = vz;
						if (VERT) { const float4 ne = a.eulerVel[j]; wx = vx + (p_euler.x - ne.x); wy = vy + (p_euler.y - ne.y); wz = vz + (p_euler.z - ne.z); }
						const float vf = sa_visc_avg(p, p.visccoeff[fl], p.visccoeff[nfl], p_rho, n_rho, nmass)*f;
						dx += vf*wx; dy += vf*wy; dz += vf*wz;
					}
					force.x += dx; force.y += dy; force.z += dz;
				};
				for_each_neib<PT_FLUID>(p, wk, index, pos, gridPos, [&](uint32_t j, const float4 &np_, float rx, float ry, float rz) { particle_pair(false, j, np_, rx, ry, rz); });
				for_each_neib<PT_VERTEX>(p, wk, index, pos, gridPos, [&](uint32_t j, const float4 &np_, float rx, float ry, float rz) { particle_pair(true, j, np_, rx, ry, rz); });
				for_each_neib<PT_BOUNDARY>(p, wk, index, pos, gridPos, [&](uint32_t j, const float4 &npos, float rx, float ry, float rz) {
					if (!is_active_w(npos.w)) return;
					const float r = sqrtf(fmaf(rz, rz, fmaf(ry, ry, rx*rx)));
					if (r >= p.influenceradius + a.deltap) return;
					const particleinfo ninfo = a.info[j];
					const float4 nvel = a.vel[j];
					const float vx = vel.x - nvel.x, vy = vel.y - nvel.y, vz = vel.z - nvel.z;
					const float4 ne = a.eulerVel[j];
					const float wx = vx + (p_euler.x - ne.x), wy = vy + (p_euler.y - ne.y), wz = vz + (p_euler.z - ne.z);
					const uint32_t nfl = FLUID_NUM(ninfo);
					const float n_rho = (nvel.w + 1.0f)*p.rho0[nfl];
					const float n_precalc = sa_P(p, nvel.w, nfl)/(n_rho*n_rho);
					const float4 be = a.boundElement[j];
					const V3 ns = v3(be.x, be.y, be.z);
					const float inv_h = 1.0f/p.slength;
					WallTri tri;
					wall_tri_setup(tri, ns, a.vertPos[0][j], a.vertPos[1][j], a.vertPos[2][j], p.slength);
					const float ggamAS = wall_grad_gamma(tri, v3(rx*inv_h, ry*inv_h, rz*inv_h))/p.slength;
					const float vn = sa_dot3(vx, vy, vz, be.x, be.y, be.z);
					if (a.cflGamma) {
						const float va = sa_dot3(vel.x, vel.y, vel.z, be.x, be.y, be.z);
						const float vs = sa_dot3(vel.x - vx, vel.y - vy, vel.z - vz, be.x, be.y, be.z);
						gammaCfl = fmaxf(gammaCfl, ggamAS*fmaxf(fabsf(vn), fmaxf(fabsf(va), fabsf(vs))));
						// compute_gamma_cfl_open_boundary (:1485-1497): n.(v_a + relEulerVel), n.(v_s - relEulerVel)
						const float ex = wx - vx, ey = wy - vy, ez = wz - vz;
						const float a1 = sa_dot3(vel.x + ex, vel.y + ey, vel.z + ez, be.x, be.y, be.z);
						const float a2 = sa_dot3(-vx + vel.x - ex, -vy + vel.y - ey, -vz + vel.z - ez, be.x, be.y, be.z);
						gammaCfl = fmaxf(gammaCfl, ggamAS*fmaxf(fabsf(a1), fabsf(a2)));
					}
					if (!density_sum) { float DrDt = 0.0f; DrDt -= p_rho*vn*ggamAS; force.w += DrDt; }
					const float ps = (p_precalc + n_precalc)*n_rho*ggamAS;
					float dx = 0.0f, dy = 0.0f, dz = 0.0f;
					dx += ps*be.x; dy += ps*be.y; dz += ps*be.z;
					if (newtonian) {      // compute_laminar_visc_contrib, boundary term (:2680-2718) with open boundaries
						const float r_as = fmaxf(fabsf(sa_dot3(rx, ry, rz, be.x, be.y, be.z)), a.deltap);
						const float wn = IS_IO_BOUNDARY(ninfo) ? 0.0f : sa_dot3(wx, wy, wz, be.x, be.y, be.z);
						const float tx = wx - wn*be.x, ty = wy - wn*be.y, tz = wz - wn*be.z;
						const float our_mu = (p.compvisc == SPHX_KINEMATIC) ? p.visccoeff[fl]*p_rho : p.visccoeff[fl];
						const float neib_mu = (p.compvisc == SPHX_KINEMATIC) ? p.visccoeff[nfl]*n_rho : p.visccoeff[nfl];
						const float avg = (p.avgop == SPHX_ARITHMETIC) ? (our_mu + neib_mu)*0.5f :
							(p.avgop == SPHX_HARMONIC) ? 2*our_mu*neib_mu/(our_mu + neib_mu) : sqrtf(our_mu*neib_mu);
						const float c = ggamAS*2*avg/r_as;
						const float inv_rho = 1.0f/p_rho;
						dx -= (c*tx)*inv_rho; dy -= (c*ty)*inv_rho; dz -= (c*tz)*inv_rho;
					}
					force.x += dx; force.y += dy; force.z += dz;
				});
				const float gam = a.gGam[index].w;
				force.x /= gam; force.y /= gam; force.z /= gam; force.w /= gam;
				force.w /= p.rho0[fl];
				force.x += p.gravity[0]; force.y += p.gravity[1]; force.z += p.gravity[2];
				if (p.simflags & SPHX_ENABLE_DTADAPT) {
					const float sspeed = sa_sound_speed(p, vel.w, fl);
					const float acc = sqrtf(fmaf(force.z, force.z, fmaf(force.y, force.y, force.x*force.x)));
					cflTerm = fmaxf(acc, sspeed*sspeed/p.slength);
				}
			}
			a.forces[index] = force;
		}
		if (a.cflGamma) a.cflGamma[index] = gammaCfl;
	}
#pragma unroll
	for (int d = 32; d > 0; d >>= 1) {
		cflTerm = fmaxf(cflTerm, __shfl_down(cflTerm, d));
		gammaCfl = fmaxf(gammaCfl, __shfl_down(gammaCfl, d));
	}
	if ((threadIdx.x & 63u) == 0u) { sMax[threadIdx.x >> 6] = cflTerm; sMaxG[threadIdx.x >> 6] = gammaCfl; }
	__syncthreads();
	if (threadIdx.x == 0 && a.cfl && (p.simflags & SPHX_ENABLE_DTADAPT)) {
		float m = sMax[0], mg = sMaxG[0];
		for (int w = 1; w < SPHX_BLOCK_FORCES/64; ++w) { m = fmaxf(m, sMax[w]); mg = fmaxf(mg, sMaxG[w]); }
		a.cfl[a.cflOffset + blockIdx.x] = m;
		if (a.cflGammaBlocks) a.cflGammaBlocks[a.cflOffset + blockIdx.x] = mg;
	}
}

extern "C" int sphx_sa_density_sum_io(sphx_ctx *ctx, void *newVel, void *newGGam, void *forces, const void *oldPos, const void *newPos,
	const void *oldVel, const void *oldEulerVel, const void *oldGGam, const void *boundElements,
	const void *vertPos0, const void *vertPos1, const void *vertPos2, const void *info,
	const uint32_t *hash, const uint32_t *cellStart, const uint16_t *neibsList,
	uint32_t numParticles, uint32_t particleRangeEnd, float dt, void *stream)
{
	(void)numParticles;
	int rc = sa_io_bc_check(ctx, "density_sum called without SA_BOUNDARY");
	if (rc != SPHX_OK) return rc;
	SPHX_REQUIRE(newVel && newGGam && forces && oldPos && newPos && oldVel && oldEulerVel && oldGGam && boundElements && vertPos0 && vertPos1 &&
		vertPos2 && info && hash && cellStart && neibsList, "sphx_sa_density_sum_io: missing buffer");
	if (!particleRangeEnd) return SPHX_OK;
	SaIoDensitySumArgs a = { (float4*)newVel, (float4*)newGGam, (float4*)forces, (const float4*)oldPos, (const float4*)newPos,
		(const float4*)oldVel, (const float4*)oldEulerVel, (const float4*)oldGGam, (const float4*)boundElements,
		{ (const float2*)vertPos0, (const float2*)vertPos1, (const float2*)vertPos2 }, (const particleinfo*)info, hash, cellStart,
		neibsList, particleRangeEnd, dt };
	SPHX_LAUNCH(sa_density_sum_io_kernel, div_up_u(particleRangeEnd, 128), 128, (hipStream_t)stream, ctx->dev, a);
	SPHX_LAUNCH_CHECK("sa_density_sum_io_kernel");
	return SPHX_OK;
}

extern "C" int sphx_forces_basicstep_sa_io(sphx_ctx *ctx, void *forces, float *cfl, float *cflGamma, const void *pos, const void *vel,
	const void *eulerVel, const void *info, const uint32_t *hash, const uint32_t *cellStart, const uint16_t *neibsList,
	const void *gGam, const void *boundElements, const void *vertPos0, const void *vertPos1, const void *vertPos2,
	uint32_t numParticles, uint32_t fromParticle, uint32_t toParticle, float deltap, uint32_t cflOffset, uint32_t *h_numBlocks, void *stream)
{
	int rc = sa_io_bc_check(ctx, "forces called without SA_BOUNDARY");
	if (rc != SPHX_OK) return rc;
	SPHX_REQUIRE(forces && pos && vel && eulerVel && info && hash && cellStart && neibsList && gGam && boundElements && vertPos0 && vertPos1 && vertPos2,
		"sphx_forces_basicstep_sa_io: missing buffer");
	SPHX_REQUIRE(fromParticle <= toParticle && toParticle <= numParticles, "sphx_forces_basicstep_sa_io: invalid particle range");
	const uint32_t numBlocks = round_up_u(div_up_u(toParticle - fromParticle, SPHX_BLOCK_FORCES), 4u);
	if (h_numBlocks) *h_numBlocks = numBlocks;
	if (!numBlocks) return SPHX_OK;
	const bool dtadapt = (ctx->dev.simflags & SPHX_ENABLE_DTADAPT) != 0;
	if (dtadapt) SPHX_REQUIRE(cfl != nullptr, "sphx_forces_basicstep_sa_io: ENABLE_DTADAPT needs the CFL buffer");
	const bool gcfl = cflGamma && dtadapt && !(ctx->dev.simflags & SPHX_ENABLE_GAMMA_QUADRATURE);
	SaIoForcesArgs a = {};
	a.forces = (float4*)forces; a.cfl = dtadapt ? cfl : nullptr;
	a.cflGamma = gcfl ? cflGamma : nullptr; a.cflGammaBlocks = gcfl ? cflGamma + round_up_u(numParticles, 4u) : nullptr;
	a.pos = (const float4*)pos; a.vel = (const float4*)vel; a.eulerVel = (const float4*)eulerVel; a.gGam = (const float4*)gGam;
	a.boundElement = (const float4*)boundElements;
	a.vertPos[0] = (const float2*)vertPos0; a.vertPos[1] = (const float2*)vertPos1; a.vertPos[2] = (const float2*)vertPos2;
	a.info = (const particleinfo*)info; a.hash = hash; a.cellStart = cellStart; a.neibsList = neibsList;
	a.fromParticle = fromParticle; a.toParticle = toParticle; a.cflOffset = cflOffset; a.deltap = deltap;
	SPHX_LAUNCH(sa_forces_io_kernel, numBlocks, SPHX_BLOCK_FORCES, (hipStream_t)stream, ctx->dev, a);
	SPHX_LAUNCH_CHECK("sa_forces_io_kernel");
	return SPHX_OK;
}

// ==========================================================================================
// The Brezzi diffusion with open boundaries and the water depth at the pressure-driven ones: computeDensityDiffusionDevice
// with ENABLE_INLET_OUTLET (forces_kernel.def:4536-4582; the boundary term :1836-1852) and what forcesDevice<PT_VERTEX, PT_FLUID>
// leaves behind with ENABLE_WATER_DEPTH (:192-205, 1375-1389, 3285-3303).  One thread per particle over the list; the checkers
// are orc_sa_density_diffusion_io / orc_sa_io_water_depth.  WRITTEN AT THE END OF ROUND 4, NOT YET RUN ON A GPU (see above).
// ==========================================================================================
struct SaIoDiffusionArgs {
	float4 *forces;
	const float4 *pos, *vel, *gGam, *boundElement;
	const float2 *vertPos[3];
	const particleinfo *info;
	const uint32_t *hash, *cellStart;
	const neibdata *neibsList;
	uint32_t numParticles;
	float dt, deltap;
};

__global__ void __launch_bounds__(128)
sa_density_diffusion_io_kernel(DevParams p, SaIoDiffusionArgs a)
{
	const uint32_t index = blockIdx.x*128 + threadIdx.x;
	if (index >= a.numParticles) return;
	const particleinfo info = a.info[index];
	if (PART_TYPE(info) != PT_FLUID) return;
	const float4 pos = a.pos[index];
	if (!is_active_w(pos.w)) return;
	const float4 vel = a.vel[index];
	const uint32_t fl = FLUID_NUM(info);
	const float rho = (vel.w + 1.0f)*p.rho0[fl];
	const float pres = sa_P(p, vel.w, fl);
	const int3 gridPos = grid_pos_from_hash(p, a.hash[index] & CELLTYPE_BITMASK);
	float DrDt = 0.0f;
	for_each_neib<PT_FLUID>(p, a, index, pos, gridPos, [&](uint32_t j, const float4 &npos, float rx, float ry, float rz) {
		const float r = sqrtf(rx*rx + ry*ry + rz*rz);
		if (!is_active_w(npos.w)) return;
		if (r >= p.influenceradius) return;
		const float4 nvel = a.vel[j];
		const uint32_t nfl = FLUID_NUM(a.info[j]);
		const float neib_rho = (nvel.w + 1.0f)*p.rho0[nfl];
		const float qm2 = r/p.slength - 2.0f;
		const float f = qm2*qm2*qm2*p.fcoeff;
		const float gdotr = p.gravity[0]*rx + p.gravity[1]*ry + p.gravity[2]*rz;
		float n = 0.0f;
		n += p.densityDiffCoeff*((2.0f/(rho + neib_rho))*(pres - sa_P(p, nvel.w, nfl)) - gdotr)*npos.w/neib_rho*f*a.dt*2.0f*rho;
		DrDt += n;
	});
	// the segments of PRESSURE-driven open boundaries: V_b grad W -> |grad gamma_as| / r_as, no diffusion coefficient, in double
	// as the reference's literals make it (:1848)
	for_each_neib<PT_BOUNDARY>(p, a, index, pos, gridPos, [&](uint32_t j, const float4 &npos, float rx, float ry, float rz) {
		const float r = sqrtf(rx*rx + ry*ry + rz*rz);
		if (!is_active_w(npos.w)) return;
		if (r >= p.influenceradius + a.deltap) return;
		const particleinfo ninfo = a.info[j];
		if (!(IS_IO_BOUNDARY(ninfo) && !IS_VEL_IO(ninfo))) return;      // nout.DrDt stays 0: DrDt += 0 changes nothing
		const float4 be = a.boundElement[j];
		const V3 ns = v3(be.x, be.y, be.z);
		const float r_as = fmaxf(fabsf(sa_dot3(rx, ry, rz, be.x, be.y, be.z)), a.deltap);
		const float inv_h = 1.0f/p.slength;
		WallTri tri;
		wall_tri_setup(tri, ns, a.vertPos[0][j], a.vertPos[1][j], a.vertPos[2][j], p.slength);
		const float ggamAS = wall_grad_gamma(tri, v3(rx*inv_h, ry*inv_h, rz*inv_h))/p.slength;
		const float nrt = a.vel[j].w;
		const uint32_t nfl = FLUID_NUM(ninfo);
		const float neib_rho = (nrt + 1.0f)*p.rho0[nfl];
		const float gdotr = p.gravity[0]*rx + p.gravity[1]*ry + p.gravity[2]*rz;
		const double t = ((2.0/(rho + neib_rho))*(pres - sa_P(p, nrt, nfl)) - gdotr)*ggamAS/r_as*a.dt*2.0f*rho;
		float n = 0.0f;
		n = (float)(n - t);
		DrDt += n;
	});
	DrDt /= a.gGam[index].w;
	a.forces[index].w = DrDt/p.rho0[fl];
}

struct SaIoDepthArgs {
	uint32_t *IOwaterdepth;
	const float4 *pos;
	const particleinfo *info;
	const uint32_t *hash, *cellStart;
	const neibdata *neibsList;
	uint32_t fromParticle, toParticle;
};

__global__ void __launch_bounds__(128)
sa_io_water_depth_kernel(DevParams p, SaIoDepthArgs a)
{
	const uint32_t index = blockIdx.x*128 + threadIdx.x + a.fromParticle;
	if (index >= a.toParticle) return;
	const particleinfo info = a.info[index];
	if (PART_TYPE(info) != PT_VERTEX) return;
	const float4 pos = a.pos[index];
	if (!is_active_w(pos.w)) return;
	if (!(IS_IO_BOUNDARY(info) && !IS_VEL_IO(info))) return;      // skip_neiblist (:1375-1389)
	const int3 gridPos = grid_pos_from_hash(p, a.hash[index] & CELLTYPE_BITMASK);
	uint32_t best = 0u;
	for_each_neib<PT_FLUID>(p, a, index, pos, gridPos, [&](uint32_t j, const float4 &npos, float rx, float ry, float rz) {
		(void)j;
		const float r = sqrtf(rx*rx + ry*ry + rz*rz);
		if (!is_active_w(npos.w)) return;
		if (r >= p.influenceradius) return;
		if (rz < 0.0f) return;
		float nZpos = pos.z - rz + gridPos.z*p.cs[2] + 0.5f*p.cs[2];
		nZpos *= ((float)UINT_MAX)/(p.gs[2]*p.cs[2]);
		const uint32_t u = (uint32_t)nZpos;
		best = u > best ? u : best;
	});
	if (best) atomicMax(a.IOwaterdepth + OBJECT_NUM(info), best);      // a maximum: one atomic per vertex gives the same number
}

extern "C" int sphx_sa_compute_density_diffusion_io(sphx_ctx *ctx, void *forces, const void *pos, const void *vel, const void *gGam,
	const void *boundElements, const void *vertPos0, const void *vertPos1, const void *vertPos2, const void *info,
	const uint32_t *hash, const uint32_t *cellStart, const uint16_t *neibsList,
	uint32_t numParticles, uint32_t particleRangeEnd, float deltap, float dt, void *stream)
{
	(void)numParticles;
	int rc = sa_io_bc_check(ctx, "compute_density_diffusion called without SA_BOUNDARY");
	if (rc != SPHX_OK) return rc;
	if (ctx->params.densitydiffusiontype != SPHX_BREZZI || !(ctx->params.simflags & SPHX_ENABLE_DENSITY_SUM) ||
		ctx->params.sph_formulation == SPHX_SPH_HA)
		return sphx_set_error(SPHX_ERR_UNSUPPORTED, "sphx_sa_compute_density_diffusion_io: built for Brezzi diffusion with density summation");
	SPHX_REQUIRE(forces && pos && vel && gGam && boundElements && vertPos0 && vertPos1 && vertPos2 && info && hash && cellStart && neibsList,
		"sphx_sa_compute_density_diffusion_io: missing buffer");
	if (!particleRangeEnd) return SPHX_OK;
	SaIoDiffusionArgs a = { (float4*)forces, (const float4*)pos, (const float4*)vel, (const float4*)gGam, (const float4*)boundElements,
		{ (const float2*)vertPos0, (const float2*)vertPos1, (const float2*)vertPos2 }, (const particleinfo*)info, hash, cellStart,
		neibsList, particleRangeEnd, dt, deltap };
	SPHX_LAUNCH(sa_density_diffusion_io_kernel, div_up_u(particleRangeEnd, 128), 128, (hipStream_t)stream, ctx->dev, a);
	SPHX_LAUNCH_CHECK("sa_density_diffusion_io_kernel");
	return SPHX_OK;
}

extern "C" int sphx_sa_io_water_depth(sphx_ctx *ctx, uint32_t *IOwaterdepth, const void *pos, const void *info, const uint32_t *hash,
	const uint32_t *cellStart, const uint16_t *neibsList, uint32_t numParticles, uint32_t fromParticle, uint32_t toParticle, void *stream)
{
	int rc = sa_io_bc_check(ctx, "the water depth is measured with SA_BOUNDARY only");
	if (rc != SPHX_OK) return rc;
	SPHX_REQUIRE(IOwaterdepth && pos && info && hash && cellStart && neibsList, "sphx_sa_io_water_depth: missing buffer");
	SPHX_REQUIRE(fromParticle <= toParticle && toParticle <= numParticles, "sphx_sa_io_water_depth: invalid particle range");
	if (fromParticle == toParticle) return SPHX_OK;
	SaIoDepthArgs a = { IOwaterdepth, (const float4*)pos, (const particleinfo*)info, hash, cellStart, neibsList, fromParticle, toParticle };
	SPHX_LAUNCH(sa_io_water_depth_kernel, div_up_u(toParticle - fromParticle, 128), 128, (hipStream_t)stream, ctx->dev, a);
	SPHX_LAUNCH_CHECK("sa_io_water_depth_kernel");
	return SPHX_OK;
}

// ==========================================================================================
// FLUX_COMPUTATION of the post-processing engine (src/cuda/post_process.cu:485-570, fluxComputationDevice
// src/cuda/post_process_kernel.cu:822-840): per open boundary the volume flux sum A_s (u_E . n_s) over its segments.  The reference
// adds onto a freshly allocated, uncleared device array; the sums start from zero here.  WRITTEN AT THE END OF ROUND 4, NOT YET RUN
// ON A GPU (see above); the checker is orc_flux_computation.
// ==========================================================================================
__global__ void __launch_bounds__(256)
sa_io_flux_kernel(const particleinfo *pinfo, const float4 *eulerVel, const float4 *boundElement, float *IOflux, uint32_t numOpenBoundaries,
	uint32_t numParticles)
{
	const uint32_t index = blockIdx.x*256 + threadIdx.x;
	if (index >= numParticles) return;
	const particleinfo info = pinfo[index];
	if (!(IS_IO_BOUNDARY(info) && PART_TYPE(info) == PT_BOUNDARY)) return;
	const uint32_t ob = OBJECT_NUM(info);
	if (ob >= numOpenBoundaries) return;      // (the reference would write out of bounds)
	const float4 normal = boundElement[index], e = eulerVel[index];
	atomicAdd(IOflux + ob, normal.w*(e.x*normal.x + e.y*normal.y + e.z*normal.z));
}

extern "C" int sphx_flux_computation(sphx_ctx *ctx, float *IOflux, const void *info, const void *eulerVel, const void *boundElements,
	uint32_t numParticles, uint32_t particleRangeEnd, uint32_t numOpenBoundaries, void *stream)
{
	(void)numParticles;
	int rc = sa_io_bc_check(ctx, "the flux through open boundaries is computed with SA_BOUNDARY only");
	if (rc != SPHX_OK) return rc;
	SPHX_REQUIRE(IOflux && info && eulerVel && boundElements, "sphx_flux_computation: missing buffer");
	if (!numOpenBoundaries) return SPHX_OK;
	SPHX_HIP(hipMemsetAsync(IOflux, 0, numOpenBoundaries*sizeof(float), (hipStream_t)stream));
	if (!particleRangeEnd) return SPHX_OK;
	SPHX_LAUNCH(sa_io_flux_kernel, div_up_u(particleRangeEnd, 256), 256, (hipStream_t)stream, (const particleinfo*)info,
		(const float4*)eulerVel, (const float4*)boundElements, IOflux, numOpenBoundaries, particleRangeEnd);
	SPHX_LAUNCH_CHECK("sa_io_flux_kernel");
	return SPHX_OK;
}
