// sa_bounds.hip -- boundary-conditions engine of the semi-analytical wall model (SA_BOUNDARY) for gfx950.
// Replaces, for solid walls (no open boundaries, no k-epsilon), CUDABoundaryConditionsEngine's
//   computeVertexNormal           src/cuda/boundary_conditions.cu:417-452   computeVertexNormalDevice        _kernel.cu:1766-1831
//   saSegmentBoundaryConditions   src/cuda/boundary_conditions.cu:108-235   saSegmentBoundaryConditions[Repack]Device :1425-1640
//   saVertexBoundaryConditions    src/cuda/boundary_conditions.cu:280-410   saVertexBoundaryConditions[Repack]Device  :2195-2310
// These run twice per step over the wall particles only (a few per cent of the particles) and walk the reference's u16
// list; like the density filters they are written for fidelity: no FMA contraction, IEEE division and sqrt, the reference's
// operation order -- bit-identical to the CPU oracle except for powf in the equation of state.
#include "sphx_internal.h"
#include "neib_iter.h"

struct SaArgs {
	float4 *vel;                 // in place: boundary rows (segment kernel) / vertex rows (vertex kernel) are written
	float4 *gGam;                // in place: boundary rows written when gamma is (re)computed
	float4 *boundElement;        // vertex-normal kernel: vertex rows written
	const float4 *pos;
	const uint4 *vertices;
	const particleinfo *info;
	const uint32_t *hash, *cellStart;
	const neibdata *neibsList;
	uint32_t numParticles;
	int step, repack;
};

__device__ __forceinline__ bool has_vertex(const uint4 &v, uint32_t id) { return v.x == id || v.y == id || v.z == id; }

// RHO (src/cuda/phys_core.cu:106-112): relative density from pressure
__device__ __forceinline__ float eos_rho(const DevParams &p, float pres, uint32_t fl)
{
	return powf(pres/p.bcoeff[fl] + 1.0f, 1.0f/p.gammacoeff[fl]) - 1.0f;
}

// common_ndata (:633-657): what a segment or a vertex needs of a fluid neighbour
struct SaNdata { float r, w, press; float4 vel; };
template<int KERNEL>
__device__ __forceinline__ SaNdata sa_fluid_ndata(const DevParams &p, const SaArgs &a, uint32_t j, float rx, float ry, float rz, float mass)
{
	SaNdata n;
	const uint32_t nfl = FLUID_NUM(a.info[j]);
	n.vel = a.vel[j];
	n.r = sqrtf(rx*rx + ry*ry + rz*rz);
	n.w = kernel_W<KERNEL>(p, n.r)*mass/((n.vel.w + 1.0f)*p.rho0[nfl]);
	n.press = p.bcoeff[nfl]*(powf(n.vel.w + 1.0f, p.gammacoeff[nfl]) - 1.0f);
	return n;
}

__global__ void __launch_bounds__(128)
sa_vertex_normal_kernel(DevParams p, SaArgs a)
{
	const uint32_t index = blockIdx.x*128 + threadIdx.x;
	if (index >= a.numParticles) return;
	const particleinfo info = a.info[index];
	if (PART_TYPE(info) != PT_VERTEX) return;
	const uint32_t our_id = info_id(info);
	const float4 pos = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
	float ax = 0.0f, ay = 0.0f, az = 0.0f;
	const int3 gridPos = grid_pos_from_hash(p, a.hash[index] & CELLTYPE_BITMASK);
	for_each_neib<PT_BOUNDARY>(p, a, index, pos, gridPos, [&](uint32_t j, const float4 &, float, float, float) {
		if (!has_vertex(a.vertices[j], our_id)) return;
		const float4 be = a.boundElement[j];
		ax += be.x*be.w; ay += be.y*be.w; az += be.z*be.w;
	});
	const float inv = 1.0f/sqrtf(ax*ax + ay*ay + az*az);
	a.boundElement[index] = make_float4(ax*inv, ay*inv, az*inv, NAN);
}

template<int KERNEL>
__global__ void __launch_bounds__(128)
sa_segment_bc_kernel(DevParams p, SaArgs a)
{
	const uint32_t index = blockIdx.x*128 + threadIdx.x;
	if (index >= a.numParticles) return;
	const particleinfo info = a.info[index];
	if (!IS_BOUNDARY(info)) return;
	const float4 pos = a.pos[index];
	const float4 normal = a.boundElement[index];
	const uint4 verts = a.vertices[index];
	const int3 gridPos = grid_pos_from_hash(p, a.hash[index] & CELLTYPE_BITMASK);
	const bool has_moving = (p.simflags & SPHX_ENABLE_MOVING_BODIES) != 0;

	// common_pout, common_segment_pout (:390-432)
	float sumpWall = 0.0f, shepard_div = 0.0f;
	float4 gGam = make_float4(0.0f, 0.0f, 0.0f, a.gGam[index].w);
	float4 vel = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
	const bool calcGam = has_moving || !is_active_w(gGam.w) || a.step == 0;       // !isfinite
	if (calcGam) gGam.w = 0.0f;
	const bool moving = has_moving && !a.repack && (info.x & FG_MOVING_BOUNDARY);

	for_each_neib<PT_VERTEX>(p, a, index, pos, gridPos, [&](uint32_t j, const float4 &npos, float, float, float) {
		if (!is_active_w(npos.w)) return;
		if (!has_vertex(verts, info_id(a.info[j]))) return;
		if (moving) {                     // moving_vertex_contrib (:781-793)
			const float4 nv = a.vel[j];
			vel.x += nv.x; vel.y += nv.y; vel.z += nv.z;
		}
		if (calcGam) {
			const float4 g = a.gGam[j];
			gGam.x += g.x; gGam.y += g.y; gGam.z += g.z; gGam.w += g.w;
		}
	});
	if (calcGam) {
		const float inv = 1.0f/3;          // float4 /= float (src/vector_math.h:1093-1097)
		gGam.x *= inv; gGam.y *= inv; gGam.z *= inv; gGam.w *= inv;
		a.gGam[index] = gGam;
		gGam.w = fmaxf(gGam.w, 1e-5f);
	}
	if (!a.repack) { vel.x /= 3; vel.y /= 3; vel.z /= 3; }

	const uint32_t fl = FLUID_NUM(info);
	for_each_neib<PT_FLUID>(p, a, index, pos, gridPos, [&](uint32_t j, const float4 &npos, float rx, float ry, float rz) {
		if (!is_active_w(npos.w)) return;
		const SaNdata n = sa_fluid_ndata<KERNEL>(p, a, j, rx, ry, rz, npos.w);
		if (!(n.r < p.influenceradius && (normal.x*rx + normal.y*ry + normal.z*rz) < 0.0f)) return;
		const float gdot = p.gravity[0]*rx + p.gravity[1]*ry + p.gravity[2]*rz;
		sumpWall += fmaxf(n.press + ((n.vel.w + 1.0f)*p.rho0[fl])*gdot, 0.0f)*n.w;
		shepard_div += n.w;
	});
	// impose_solid_bc (:1295-1306)
	shepard_div = fmaxf(shepard_div, 0.1f*gGam.w);
	vel.w = eos_rho(p, sumpWall/shepard_div, fl);
	a.vel[index] = vel;
}

template<int KERNEL>
__global__ void __launch_bounds__(128)
sa_vertex_bc_kernel(DevParams p, SaArgs a)
{
	const uint32_t index = blockIdx.x*128 + threadIdx.x;
	if (index >= a.numParticles) return;
	const particleinfo info = a.info[index];
	if (PART_TYPE(info) != PT_VERTEX) return;
	const float4 pos = a.pos[index];
	const float gam = a.gGam[index].w;
	const uint32_t fl = FLUID_NUM(info);
	const int3 gridPos = grid_pos_from_hash(p, a.hash[index] & CELLTYPE_BITMASK);
	float sumpWall = 0.0f, shepard_div = 0.0f;
	for_each_neib<PT_FLUID>(p, a, index, pos, gridPos, [&](uint32_t j, const float4 &npos, float rx, float ry, float rz) {
		if (!is_active_w(npos.w)) return;
		const SaNdata n = sa_fluid_ndata<KERNEL>(p, a, j, rx, ry, rz, npos.w);
		if (n.r < p.influenceradius) {
			const float gdot = p.gravity[0]*rx + p.gravity[1]*ry + p.gravity[2]*rz;
			sumpWall += fmaxf(n.press + ((n.vel.w + 1.0f)*p.rho0[fl])*gdot, 0.0f)*n.w;
			shepard_div += n.w;
		}
	});
	shepard_div = fmaxf(shepard_div, 0.1f*gam);
	a.vel[index].w = eos_rho(p, sumpWall/shepard_div, fl);
}

static int sa_check(sphx_ctx *ctx, const char *who)
{
	if (!ctx || !ctx->have_params) return sphx_set_error(SPHX_ERR_INVALID, "sphx_sa: constants not set");
	if (ctx->params.boundarytype != SPHX_SA_BOUNDARY)
		return sphx_set_error(SPHX_ERR_INVALID, who);      // the reference throws "... called without SA_BOUNDARY"
	if (ctx->params.kerneltype != SPHX_WENDLAND)
		return sphx_set_error(SPHX_ERR_UNSUPPORTED, "sphx_sa: SA_BOUNDARY is built for the Wendland kernel (as the reference, src/cuda/gamma.cuh:241-250)");
	if (ctx->params.simflags & (SPHX_ENABLE_INLET_OUTLET | SPHX_ENABLE_DENSITY_SUM))
		return sphx_set_error(SPHX_ERR_UNSUPPORTED, "sphx_sa: open boundaries and density summation are not built");
	return SPHX_OK;
}

extern "C" int sphx_sa_compute_vertex_normal(sphx_ctx *ctx, void *boundElements, const void *vertices, const void *info,
	const uint32_t *hash, const uint32_t *cellStart, const uint16_t *neibsList,
	uint32_t numParticles, uint32_t particleRangeEnd, void *stream)
{
	(void)numParticles;
	int rc = sa_check(ctx, "computeVertexNormal called without SA_BOUNDARY");
	if (rc != SPHX_OK) return rc;
	SPHX_REQUIRE(boundElements && vertices && info && hash && cellStart && neibsList, "sphx_sa_compute_vertex_normal: missing buffer");
	if (!particleRangeEnd) return SPHX_OK;
	SaArgs a = {};
	a.boundElement = (float4*)boundElements; a.vertices = (const uint4*)vertices; a.info = (const particleinfo*)info;
	a.hash = hash; a.cellStart = cellStart; a.neibsList = neibsList; a.numParticles = particleRangeEnd;
	a.pos = (const float4*)boundElements;     // the walker prefetches a position row per entry; the kernel does not use it
	sa_vertex_normal_kernel<<<div_up_u(particleRangeEnd, 128), 128, 0, (hipStream_t)stream>>>(ctx->dev, a);
	SPHX_LAUNCH_CHECK("sa_vertex_normal_kernel");
	return SPHX_OK;
}

extern "C" int sphx_sa_segment_bc(sphx_ctx *ctx, void *vel, void *gGam, const void *pos, const void *vertices,
	const void *boundElements, const void *info, const uint32_t *hash, const uint32_t *cellStart, const uint16_t *neibsList,
	uint32_t numParticles, uint32_t particleRangeEnd, float deltap, float slength, float influenceradius,
	int step, int run_mode, void *stream)
{
	(void)numParticles; (void)deltap;
	int rc = sa_check(ctx, "saSegmentBoundaryConditions called without SA_BOUNDARY");
	if (rc != SPHX_OK) return rc;
	SPHX_REQUIRE(vel && gGam && pos && vertices && boundElements && info && hash && cellStart && neibsList,
		"sphx_sa_segment_bc: missing buffer");
	SPHX_REQUIRE(step >= -1 && step <= 2, "sphx_sa_segment_bc: unsupported step");
	SPHX_REQUIRE(slength == ctx->params.slength && influenceradius == ctx->params.influenceradius,
		"sphx_sa_segment_bc: slength / influenceradius differ from the uploaded constants");
	if (!particleRangeEnd) return SPHX_OK;
	SaArgs a = {};
	a.vel = (float4*)vel; a.gGam = (float4*)gGam; a.pos = (const float4*)pos; a.vertices = (const uint4*)vertices;
	a.boundElement = (float4*)const_cast<void*>(boundElements); a.info = (const particleinfo*)info;
	a.hash = hash; a.cellStart = cellStart; a.neibsList = neibsList; a.numParticles = particleRangeEnd;
	a.step = (step == -1) ? 0 : step;         // "step -1 is the same as step 0", boundary_conditions.cu:177-180
	a.repack = (run_mode == SPHX_REPACK);
	sa_segment_bc_kernel<SPHX_WENDLAND><<<div_up_u(particleRangeEnd, 128), 128, 0, (hipStream_t)stream>>>(ctx->dev, a);
	SPHX_LAUNCH_CHECK("sa_segment_bc_kernel");
	return SPHX_OK;
}

extern "C" int sphx_sa_vertex_bc(sphx_ctx *ctx, void *vel, const void *gGam, const void *pos, const void *info,
	const uint32_t *hash, const uint32_t *cellStart, const uint16_t *neibsList,
	uint32_t numParticles, uint32_t particleRangeEnd, float deltap, float slength, float influenceradius,
	int step, int run_mode, void *stream)
{
	(void)numParticles; (void)deltap; (void)step; (void)run_mode;
	int rc = sa_check(ctx, "saVertexBoundaryConditions called without SA_BOUNDARY");
	if (rc != SPHX_OK) return rc;
	SPHX_REQUIRE(vel && gGam && pos && info && hash && cellStart && neibsList, "sphx_sa_vertex_bc: missing buffer");
	SPHX_REQUIRE(slength == ctx->params.slength && influenceradius == ctx->params.influenceradius,
		"sphx_sa_vertex_bc: slength / influenceradius differ from the uploaded constants");
	if (!particleRangeEnd) return SPHX_OK;
	SaArgs a = {};
	a.vel = (float4*)vel; a.gGam = (float4*)const_cast<void*>(gGam); a.pos = (const float4*)pos; a.info = (const particleinfo*)info;
	a.hash = hash; a.cellStart = cellStart; a.neibsList = neibsList; a.numParticles = particleRangeEnd;
	sa_vertex_bc_kernel<SPHX_WENDLAND><<<div_up_u(particleRangeEnd, 128), 128, 0, (hipStream_t)stream>>>(ctx->dev, a);
	SPHX_LAUNCH_CHECK("sa_vertex_bc_kernel");
	return SPHX_OK;
}
