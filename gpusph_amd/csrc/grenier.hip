// grenier.hip -- SPH_GRENIER (Grenier et al. multi-fluid volume formulation) for gfx950: the options of the reference's
// Bubble / LockExchange / RTInstability / OilJet problems (formulation<SPH_GRENIER>, viscosity<DYNAMICVISC>,
// boundary<DYN_BOUNDARY>, multi-fluid).
//   sphx_compute_density            CUDAForcesEngine::compute_density (src/cuda/forces.cu:208-246), densityGrenierDevice
//                                   (src/cuda/forces_kernel.cu:284-398): sigma and the smoothed density, vel.w in place
//   sphx_forces_basicstep_grenier   basicstep with the Grenier specialisations of src/cuda/forces_kernel.def
//                                   (precalc_pressure :445-455, mass_continuity_div_vel_term :2018-2028,
//                                   apply_pseudo_surface_tension :2226-2238, compute_pressure_contrib :2383-2392,
//                                   compute_laminar_visc_contrib :2628-2646, forces_fixup :3181-3190) + finalize
//   sphx_euler_basicstep_grenier    euler.hip (volume integration, euler_kernel.def:210-216,281-289)
// One thread per particle walking its own u16 list (neib_iter.h), like the other fidelity engines: the pair terms are in
// the reference's order; the quantities that depend on the neighbour alone (P/sigma, 1/sigma, rho, mu) come from a
// per-particle row written once per launch, so the pair loop has no powf.  Results are checked against oracle/sph_oracle.c
// at fp32 tolerance (tests/test_gpu_grenier.py).
#include "neib_iter.h"

struct GrenierDensityArgs {
	float *sigma;
	float4 *vel;
	const float4 *pos, *vol;
	const particleinfo *info;
	const uint32_t *hash, *cellStart;
	const neibdata *neibsList;
	const NeibsCounters *counters;    // maxFluidBoundaryNeibs of the last list build (cuneibs::d_maxFluidBoundaryNeibs)
	uint32_t numParticles;
};

__global__ void __launch_bounds__(128)
grenier_density_kernel(DevParams p, GrenierDensityArgs a)
{
	const uint32_t index = blockIdx.x*128 + threadIdx.x;
	if (index >= a.numParticles) return;
	const particleinfo info = a.info[index];
	const bool dyn = p.boundarytype == SPHX_DYN_BOUNDARY;
	if (!dyn && !IS_FLUID(info)) return;
	const float4 pos = a.pos[index];
	if (!is_active_w(pos.w)) return;
	const uint32_t fnum = FLUID_NUM(info);
	const uint32_t ptype = PART_TYPE(info);
	const float vol = a.vol[index].w;
	float4 vel = a.vel[index];
	float corr = kernel_W<SPHX_WENDLAND>(p, 0.0f);    // self contribution
	float sigma = corr;
	float mass_corr = pos.w*corr;
	const int3 gridPos = grid_pos_from_hash(p, a.hash[index] & CELLTYPE_BITMASK);
	bool has_fluid_neibs = false;
	auto pair = [&](uint32_t j, const float4 &npos, float rx, float ry, float rz) {
		const float r = sqrtf(rx*rx + ry*ry + rz*rz);
		if (!is_active_w(npos.w) || r >= p.influenceradius) return;
		const particleinfo ninfo = a.info[j];
		const float w = kernel_W<SPHX_WENDLAND>(p, r);
		sigma += w;
		if (IS_FLUID(ninfo)) has_fluid_neibs = true;
		// smoothed mass: particles of the same type (with DYN_BOUNDARY) and the same fluid
		if ((!dyn || PART_TYPE(ninfo) == ptype) && FLUID_NUM(ninfo) == fnum) {
			mass_corr += npos.w*w;
			corr += w;
		}
	};
	for_each_neib<PT_FLUID>(p, a, index, pos, gridPos, pair);
	if (dyn) for_each_neib<PT_BOUNDARY>(p, a, index, pos, gridPos, pair);
	if (dyn && !IS_FLUID(info) && !has_fluid_neibs) {
		// 'typical' specific volume for boundary particles out of reach of the fluid (:381-389)
		const float R = p.influenceradius;
		sigma = (float)(3*a.counters->maxFluidBoundaryNeibs)/(4*3.14159265358979323846f*R*R*R);
	}
	vel.w = mass_corr/(corr*vol);
	vel.w = vel.w/p.rho0[fnum] - 1.0f;    // numerical_density
	a.vel[index] = vel;
	a.sigma[index] = sigma;
}

extern "C" int sphx_compute_density(sphx_ctx *ctx, float *sigma, void *vel, const void *pos, const void *info,
	const uint32_t *hash, const void *vol, const uint32_t *cellStart, const uint16_t *neibsList,
	uint32_t numParticles, float slength, float influenceradius, void *stream)
{
	SPHX_REQUIRE(ctx && ctx->have_params, "sphx_compute_density: constants not set");
	if (ctx->params.sph_formulation != SPHX_SPH_GRENIER)
		return SPHX_OK;      // CUDADensityHelper does nothing for the other formulations (src/cuda/forces.cu:192-206)
	SPHX_REQUIRE(sigma && vel && pos && info && hash && vol && cellStart && neibsList, "sphx_compute_density: missing buffer");
	SPHX_REQUIRE(slength == ctx->params.slength && influenceradius == ctx->params.influenceradius,
		"sphx_compute_density: slength / influenceradius differ from the uploaded constants");
	if (!numParticles) return SPHX_OK;
	GrenierDensityArgs a = {};
	a.sigma = sigma; a.vel = (float4*)vel; a.pos = (const float4*)pos; a.vol = (const float4*)vol;
	a.info = (const particleinfo*)info; a.hash = hash; a.cellStart = cellStart; a.neibsList = neibsList;
	a.counters = ctx->counters_dev; a.numParticles = numParticles;
	grenier_density_kernel<<<div_up_u(numParticles, 128), 128, 0, (hipStream_t)stream>>>(ctx->dev, a);
	SPHX_LAUNCH_CHECK("grenier_density_kernel");
	return SPHX_OK;
}

// per-particle row of the forces pass: P/sigma, 1/sigma, rho, mu (get_dynamic_visc :279-291)
static __global__ void __launch_bounds__(256)
grenier_row_kernel(DevParams p, const float4 *__restrict__ vel, const particleinfo *__restrict__ info,
	const float *__restrict__ sigma, float4 *__restrict__ row, uint32_t n)
{
	const uint32_t i = blockIdx.x*256 + threadIdx.x;
	if (i >= n) return;
	const uint32_t fl = FLUID_NUM(info[i]);
	const float ratio = vel[i].w + 1.0f;
	const float P = p.bcoeff[fl]*(powf(ratio, p.gammacoeff[fl]) - 1.0f);
	const float rho = ratio*p.rho0[fl];
	const float s = sigma[i];
	const float mu = (p.compvisc == SPHX_KINEMATIC) ? p.visccoeff[fl]*rho : p.visccoeff[fl];
	row[i] = make_float4(P/s, 1/s, rho, mu);
}

struct GrenierForcesArgs {
	float4 *forces;
	float *cfl;
	const float4 *pos, *vel, *row;
	const float *sigma;
	const particleinfo *info;
	const uint32_t *hash, *cellStart;
	const neibdata *neibsList;
	uint32_t fromParticle, toParticle, cflOffset;
};

__global__ void __launch_bounds__(SPHX_BLOCK_FORCES)
grenier_forces_kernel(DevParams p, GrenierForcesArgs a)
{
	__shared__ float sMax[SPHX_BLOCK_FORCES/64];
	const uint32_t index = blockIdx.x*SPHX_BLOCK_FORCES + threadIdx.x + a.fromParticle;
	float cflTerm = 0.0f;
	if (index < a.toParticle) {
		const particleinfo info = a.info[index];
		const float4 pos = a.pos[index];
		const uint32_t ptype = PART_TYPE(info);
		if (is_active_w(pos.w) && (ptype == PT_FLUID || ptype == PT_BOUNDARY)) {
			float4 force = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
			const float4 vel = a.vel[index];
			const uint32_t fl = FLUID_NUM(info);
			const int3 gridPos = grid_pos_from_hash(p, a.hash[index] & CELLTYPE_BITMASK);
			const float4 self = a.row[index];      // P/sigma, 1/sigma, rho, mu
			const bool newtonian = p.rheology == SPHX_NEWTONIAN;
			const bool fluid = ptype == PT_FLUID;
			// momentum: every pair of a fluid particle; boundary particles integrate their volume, and those of a body with
			// force feedback also collect the momentum terms (compute_pp_interaction :3688-3705)
			const bool momentum = fluid || HAS_COMPUTE_FORCE(info);
			auto pair = [&](uint32_t j, const float4 &npos, float rx, float ry, float rz, bool nfluid) {
				if (!is_active_w(npos.w)) return;
				const float r = sqrtf(fmaf(rz, rz, fmaf(ry, ry, rx*rx)));
				if (r >= p.influenceradius) return;
				const float4 nvel = a.vel[j];
				const float vx = vel.x - nvel.x, vy = vel.y - nvel.y, vz = vel.z - nvel.z;
				const float vel_dot_pos = fmaf(vz, rz, fmaf(vy, ry, vx*rx));
				const float qm2 = r/p.slength - 2.0f;
				const float f = qm2*qm2*qm2*p.fcoeff;
				float DrDt = 0.0f;
				DrDt -= vel_dot_pos*f;           // D(log J)/Dt without the 1/sigma in front
				force.w += DrDt;
				if (!momentum) return;
				const float4 nrow = a.row[j];
				float pGradTerm = self.x + nrow.x;
				if (fluid && nfluid && FLUID_NUM(a.info[j]) != fl)     // interface between two fluids
					pGradTerm += p.epsinterface*(fabsf(self.x) + fabsf(nrow.x));
				const float s = pGradTerm*f;
				float dx = 0.0f, dy = 0.0f, dz = 0.0f;
				dx -= s*rx; dy -= s*ry; dz -= s*rz;
				if (newtonian) {
					const float a_mu = self.w, b_mu = nrow.w;
					const float avg_mu = (p.avgop == SPHX_ARITHMETIC) ? (a_mu + b_mu)*0.5f :
						(p.avgop == SPHX_HARMONIC) ? 2*a_mu*b_mu/(a_mu + b_mu) : sqrtf(a_mu*b_mu);
					const float c = avg_mu*(self.y + nrow.y)*f;
					dx += c*vx; dy += c*vy; dz += c*vz;
				}
				force.x += dx; force.y += dy; force.z += dz;
			};
			// launch order of the reference: fluid <- fluid, fluid <- boundary; boundary <- fluid
			for_each_neib<PT_FLUID>(p, a, index, pos, gridPos,
				[&](uint32_t j, const float4 &npos, float rx, float ry, float rz) { pair(j, npos, rx, ry, rz, true); });
			if (fluid)
				for_each_neib<PT_BOUNDARY>(p, a, index, pos, gridPos,
					[&](uint32_t j, const float4 &npos, float rx, float ry, float rz) { pair(j, npos, rx, ry, rz, false); });
			// forces_fixup :3181-3190
			force.x /= self.z; force.y /= self.z; force.z /= self.z;
			force.w /= a.sigma[index];
			if (fluid) {
				force.x += p.gravity[0]; force.y += p.gravity[1]; force.z += p.gravity[2];
				// GeometryForce / PlaneForce (src/cuda/forces_kernel.cu:140-203), as in forces.hip finalize_particle
				if ((p.simflags & SPHX_ENABLE_PLANES) && p.numplanes) {
					for (uint32_t k = 0; k < p.numplanes; ++k) {
						const float ddx = (gridPos.x - p.plane_gridpos[k][0])*p.cs[0] + (pos.x - p.plane_pos[k][0]);
						const float ddy = (gridPos.y - p.plane_gridpos[k][1])*p.cs[1] + (pos.y - p.plane_pos[k][1]);
						const float ddz = (gridPos.z - p.plane_gridpos[k][2])*p.cs[2] + (pos.z - p.plane_pos[k][2]);
						const float r = fabsf(ddx*p.plane_normal[k][0] + ddy*p.plane_normal[k][1] + ddz*p.plane_normal[k][2]);
						if (r < p.r0) {
							const float DvDt = p.dcoeff*(powf(p.r0/r, p.p1coeff) - powf(p.r0/r, p.p2coeff))/(r*r);
							const float qx = p.plane_normal[k][0]*r, qy = p.plane_normal[k][1]*r, qz = p.plane_normal[k][2]*r;
							force.x += DvDt*qx; force.y += DvDt*qy; force.z += DvDt*qz;
							if (newtonian) {
								const float dd = (vel.x*qx + vel.y*qy + vel.z*qz)/r, inv = 1.0f/r;
								const float coeff = -self.w*p.partsurf/(pos.w*r);
								force.x += coeff*(vel.x - (dd*qx)*inv); force.y += coeff*(vel.y - (dd*qy)*inv);
								force.z += coeff*(vel.z - (dd*qz)*inv);
							}
						}
					}
				}
				if (p.simflags & SPHX_ENABLE_DTADAPT) {
					const float sspeed = p.sscoeff[fl]*powf(vel.w + 1.0f, p.sspowercoeff[fl]);
					const float acc = sqrtf(fmaf(force.z, force.z, fmaf(force.y, force.y, force.x*force.x)));
					cflTerm = fmaxf(acc, sspeed*sspeed/p.slength);
				}
			}
			a.forces[index] = force;
		}
	}
#pragma unroll
	for (int d = 32; d > 0; d >>= 1) cflTerm = fmaxf(cflTerm, __shfl_down(cflTerm, d));
	if ((threadIdx.x & 63u) == 0u) sMax[threadIdx.x >> 6] = cflTerm;
	__syncthreads();
	if (threadIdx.x == 0 && a.cfl && (p.simflags & SPHX_ENABLE_DTADAPT)) {
		float m = sMax[0];
		for (int w = 1; w < SPHX_BLOCK_FORCES/64; ++w) m = fmaxf(m, sMax[w]);
		a.cfl[a.cflOffset + blockIdx.x] = m;
	}
}

int sphx_grenier_check(const sphx_ctx *ctx, const char *what)
{
	SPHX_REQUIRE(ctx && ctx->have_params, "sphx (SPH_GRENIER): constants not set");
	const sphx_params &q = ctx->params;
	if (q.sph_formulation != SPHX_SPH_GRENIER) return sphx_set_error(SPHX_ERR_INVALID, what);
	return SPHX_OK;
}

extern "C" int sphx_forces_basicstep_grenier(sphx_ctx *ctx, void *forces, float *cfl,
	const void *pos, const void *vel, const void *info, const uint32_t *hash,
	const uint32_t *cellStart, const uint16_t *neibsList, const float *sigma,
	uint32_t numParticles, uint32_t fromParticle, uint32_t toParticle,
	float deltap, float slength, float dtadaptfactor, float influenceradius,
	uint32_t cflOffset, int run_mode, int step, float dt, uint32_t *h_numBlocks, void *stream)
{
	(void)deltap; (void)dtadaptfactor; (void)step; (void)dt;
	int rc = sphx_grenier_check(ctx, "sphx_forces_basicstep_grenier called without SPH_GRENIER");
	if (rc != SPHX_OK) return rc;
	if (run_mode != SPHX_SIMULATE)
		return sphx_set_error(SPHX_ERR_UNSUPPORTED, "sphx_forces_basicstep_grenier: repacking runs use sphx_forces_basicstep (no Grenier terms there)");
	SPHX_REQUIRE(forces && pos && vel && info && hash && cellStart && neibsList && sigma, "sphx_forces_basicstep_grenier: missing buffer");
	SPHX_REQUIRE(fromParticle <= toParticle && toParticle <= numParticles, "sphx_forces_basicstep_grenier: empty or inverted range");
	SPHX_REQUIRE(numParticles <= ctx->reserved_particles, "sphx_forces_basicstep_grenier: more particles than sphx_reserve() allowed for");
	SPHX_REQUIRE(slength == ctx->params.slength && influenceradius == ctx->params.influenceradius,
		"sphx_forces_basicstep_grenier: slength / influenceradius differ from the uploaded constants");
	hipStream_t st = (hipStream_t)stream;
	const uint32_t count = toParticle - fromParticle;
	const uint32_t blocks = div_up_u(count, SPHX_BLOCK_FORCES);
	const uint32_t numBlocks = round_up_u(blocks, 4u);
	if (h_numBlocks) *h_numBlocks = numBlocks;
	if (!count) return SPHX_OK;
	const bool dtadapt = (ctx->params.simflags & SPHX_ENABLE_DTADAPT) != 0;
	if (dtadapt) {
		SPHX_REQUIRE(cfl != nullptr, "sphx_forces_basicstep_grenier: ENABLE_DTADAPT needs the CFL buffer");
		if (numBlocks > blocks) SPHX_HIP(hipMemsetAsync(cfl + cflOffset + blocks, 0, sizeof(float)*(numBlocks - blocks), st));
	}
	// the rows are read for the neighbours as well: every particle, not just the range
	ctx->eos_tag_vel = nullptr;      // the scratch rows change hands
	grenier_row_kernel<<<div_up_u(numParticles, 256), 256, 0, st>>>(ctx->dev, (const float4*)vel, (const particleinfo*)info, sigma,
		ctx->eos_aux, numParticles);
	SPHX_LAUNCH_CHECK("grenier_row_kernel");
	GrenierForcesArgs a = {};
	a.forces = (float4*)forces; a.cfl = cfl; a.pos = (const float4*)pos; a.vel = (const float4*)vel; a.row = ctx->eos_aux;
	a.sigma = sigma; a.info = (const particleinfo*)info; a.hash = hash; a.cellStart = cellStart; a.neibsList = neibsList;
	a.fromParticle = fromParticle; a.toParticle = toParticle; a.cflOffset = cflOffset;
	grenier_forces_kernel<<<blocks, SPHX_BLOCK_FORCES, 0, st>>>(ctx->dev, a);
	SPHX_LAUNCH_CHECK("grenier_forces_kernel");
	return SPHX_OK;
}

// ProblemCore::init_volume (src/ProblemCore.cc:1586-1606) runs on the host in the reference; here on the uploaded arrays
__global__ void __launch_bounds__(256)
grenier_init_volume_kernel(DevParams p, float4 *vol, const float4 *pos, const float4 *vel, const particleinfo *info, uint32_t n)
{
	const uint32_t i = blockIdx.x*256 + threadIdx.x;
	if (i >= n) return;
	const float v = pos[i].w/((vel[i].w + 1.0f)*p.rho0[FLUID_NUM(info[i])]);
	vol[i] = make_float4(v, 0.0f, 0.0f, v);
}

extern "C" int sphx_init_volume(sphx_ctx *ctx, void *vol, const void *pos, const void *vel, const void *info,
	uint32_t numParticles, void *stream)
{
	SPHX_REQUIRE(ctx && ctx->have_params, "sphx_init_volume: constants not set");
	SPHX_REQUIRE(vol && pos && vel && info, "sphx_init_volume: missing buffer");
	if (!numParticles) return SPHX_OK;
	grenier_init_volume_kernel<<<div_up_u(numParticles, 256), 256, 0, (hipStream_t)stream>>>(ctx->dev, (float4*)vol,
		(const float4*)pos, (const float4*)vel, (const particleinfo*)info, numParticles);
	SPHX_LAUNCH_CHECK("grenier_init_volume_kernel");
	return SPHX_OK;
}
