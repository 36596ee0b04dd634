// energy.hip -- ENABLE_INTERNAL_ENERGY (AccuracyTest.cu's flag) for gfx950.
//   sphx_forces_internal_energy   BUFFER_INTERNAL_ENERGY_UPD of a forces pass: add_internal_energy (src/cuda/forces_kernel.def:3308-3320),
//                                 DEDt_a = -1/2 sum_b DvDt_ab . v_ab over every pair whose momentum term the three forcesDevice
//                                 launches compute (fluid <- fluid, fluid <- boundary, boundary <- fluid; with this flag the
//                                 boundary particles of DYN_BOUNDARY compute theirs too, :3661)
//   sphx_euler_internal_energy    integrate_energy / write_energy of eulerDevice (src/cuda/euler_kernel.def:184-199,296-309)
// The optimised forces kernels accumulate the acceleration only; the energy rate needs every pair's own term, so it is a
// separate walk over the lists (one thread per particle, neib_iter.h) that re-evaluates the pair momentum terms in the
// reference's order with exact powf / division.  Checked against oracle/sph_oracle.c at fp32 tolerance.
// Built for SPH_F1 / SPH_F2, artificial or laminar Newtonian viscosity, LJ / MK / DYN boundaries; not with SPS, SA_BOUNDARY,
// SPH_GRENIER or the generalized Newtonian rheologies.
#include "neib_iter.h"

struct EnergyArgs {
	float *dedt;
	const float4 *pos, *vel, *row;      // row: P, rho, c (fidelity_row_kernel, rheology.hip)
	const particleinfo *info;
	const uint32_t *hash, *cellStart;
	const neibdata *neibsList;
	uint32_t fromParticle, toParticle;
};

__device__ __forceinline__ float en_F(const DevParams &p, float r)
{
	const float R = r/p.slength;
	switch (p.kerneltype) {
	case SPHX_CUBICSPLINE: return ((R < 1.0f) ? (-4.0f + 3.0f*R)/p.slength : -(-2.0f + R)*(-2.0f + R)/r)*p.fcoeff;
	case SPHX_QUADRATIC: return (-2.0f + R)/r*p.fcoeff;
	case SPHX_GAUSSIAN: return -expf(-R*R)*p.fcoeff;
	default: { const float qm2 = R - 2.0f; return qm2*qm2*qm2*p.fcoeff; }
	}
}

__global__ void __launch_bounds__(128)
internal_energy_kernel(DevParams p, EnergyArgs a)
{
	const uint32_t index = blockIdx.x*128 + threadIdx.x + a.fromParticle;
	if (index >= a.toParticle) return;
	const particleinfo info = a.info[index];
	const float4 pos = a.pos[index];
	const uint32_t ptype = PART_TYPE(info);
	float dedt = 0.0f;
	if (is_active_w(pos.w) && (ptype == PT_FLUID || ptype == PT_BOUNDARY)) {
		const float4 vel = a.vel[index];
		const uint32_t fl = FLUID_NUM(info);
		const int3 gridPos = grid_pos_from_hash(p, a.hash[index] & CELLTYPE_BITMASK);
		const bool fluid = ptype == PT_FLUID;
		const bool f2 = p.formulation == SPHX_SPH_F2;
		const bool repulsive = p.boundarytype == SPHX_LJ_BOUNDARY;      // LJ_BOUNDARY, or MK_BOUNDARY (uploaded as LJ + mk_mask)
		const bool dyn = p.boundarytype == SPHX_DYN_BOUNDARY;
		const float4 self = a.row[index];
		const float p_rho = self.y;
		const float p_P = self.x;
		const float p_precalc = f2 ? p_P : p_P/(p_rho*p_rho);
		const float p_sspeed = self.z;
		// the momentum terms of one pair, in the order of compute_all_pp_interaction (pressure, turbulent, laminar)
		auto pair_pp = [&](uint32_t j, const float4 &npos, float rx, float ry, float rz) {
			if (!is_active_w(npos.w)) return;
			const float r = sqrtf(fmaf(rz, rz, fmaf(ry, ry, rx*rx)));
			if (r >= p.influenceradius) return;
			const float4 nvel = a.vel[j];
			const float vx = vel.x - nvel.x, vy = vel.y - nvel.y, vz = vel.z - nvel.z;
			const uint32_t nfl = FLUID_NUM(a.info[j]);
			const float4 nrow = a.row[j];
			const float n_rho = nrow.y;
			const float n_P = nrow.x;
			const float n_precalc = f2 ? n_P : n_P/(n_rho*n_rho);
			const float f = en_F(p, r);
			const float nmass = npos.w;
			const float pGradTerm = f2 ? (p_precalc + n_precalc)/(p_rho*n_rho) : p_precalc + n_precalc;
			const float s = pGradTerm*nmass*f;
			float dx = 0.0f, dy = 0.0f, dz = 0.0f;
			dx -= s*rx; dy -= s*ry; dz -= s*rz;
			if (p.turbmodel == SPHX_ARTIFICIAL) {
				const float vel_dot_pos = sa_dot3(vx, vy, vz, rx, ry, rz);
				if (vel_dot_pos < 0.0f) {
					const float visc = vel_dot_pos*p.slength*p.artvisccoeff*(p_sspeed + nrow.z)/
						((r*r + p.epsartvisc)*(p_rho + n_rho));
					dx += visc*rx*nmass*f; dy += visc*ry*nmass*f; dz += visc*rz*nmass*f;
				}
			}
			if (p.rheology == SPHX_NEWTONIAN) {
				const float vf = sa_visc_avg(p, p.visccoeff[fl], p.visccoeff[nfl], p_rho, n_rho, nmass)*f;
				dx += vf*vx; dy += vf*vy; dz += vf*vz;
			}
			dedt -= sa_dot3(dx, dy, dz, vx, vy, vz)/2;
		};
		// repulsion of a boundary particle (compute_repulsive_force :3001-3016)
		auto pair_lj = [&](uint32_t j, const float4 &npos, float rx, float ry, float rz) {
			if (!is_active_w(npos.w)) return;
			const float r = sqrtf(fmaf(rz, rz, fmaf(ry, ry, rx*rx)));
			if (r >= p.influenceradius) return;
			const float4 nvel = a.vel[j];
			const float vx = vel.x - nvel.x, vy = vel.y - nvel.y, vz = vel.z - nvel.z;
			float ljf = 0.0f;
			if (!p.mk_mask) {
				if (r <= p.r0) ljf = p.dcoeff*(powf(p.r0/r, p.p1coeff) - powf(p.r0/r, p.p2coeff))/(r*r);
			} else if (r <= 2*p.slength) {
				const float qq = r/p.slength;
				const float w = 1.8f*powf(1.0f - 0.5f*qq, 4.0f)*(2.0f*qq + 1.0f);
				const float dist = fmaxf(p.epsartvisc, r - p.MK_d);
				ljf = p.MK_K*w*2*pos.w/(p.MK_beta*dist*r*(pos.w + pos.w));
			}
			dedt -= sa_dot3(ljf*rx, ljf*ry, ljf*rz, vx, vy, vz)/2;
		};
		if (fluid) {
			for_each_neib<PT_FLUID>(p, a, index, pos, gridPos, pair_pp);
			if (dyn) for_each_neib<PT_BOUNDARY>(p, a, index, pos, gridPos, pair_pp);
			else if (repulsive) for_each_neib<PT_BOUNDARY>(p, a, index, pos, gridPos, pair_lj);
		} else if (dyn) {
			for_each_neib<PT_FLUID>(p, a, index, pos, gridPos, pair_pp);      // every wall particle, with this flag (:3661)
		} else if (repulsive && HAS_COMPUTE_FORCE(info)) {
			for_each_neib<PT_FLUID>(p, a, index, pos, gridPos, pair_lj);      // bodies with force feedback (:3620-3645)
		}
	}
	a.dedt[index] = dedt;
}

static int energy_check(const sphx_ctx *ctx, const char *who)
{
	SPHX_REQUIRE(ctx && ctx->have_params, "sphx (internal energy): constants not set");
	const sphx_params &q = ctx->params;
	if (!(q.simflags & SPHX_ENABLE_INTERNAL_ENERGY)) return sphx_set_error(SPHX_ERR_INVALID, who);
	if (q.turbmodel == SPHX_SPS || q.boundarytype == SPHX_SA_BOUNDARY || q.sph_formulation == SPHX_SPH_GRENIER || q.rheologytype > SPHX_NEWTONIAN)
		return sphx_set_error(SPHX_ERR_UNSUPPORTED, "sphx: ENABLE_INTERNAL_ENERGY is built for SPH_F1/F2 with artificial or laminar Newtonian viscosity and LJ / MK / DYN boundaries");
	return SPHX_OK;
}

extern "C" int sphx_forces_internal_energy(sphx_ctx *ctx, float *DEDt,
	const void *pos, const void *vel, const void *info, const uint32_t *hash,
	const uint32_t *cellStart, const uint16_t *neibsList,
	uint32_t numParticles, uint32_t fromParticle, uint32_t toParticle, void *stream)
{
	int rc = energy_check(ctx, "sphx_forces_internal_energy called without ENABLE_INTERNAL_ENERGY");
	if (rc != SPHX_OK) return rc;
	SPHX_REQUIRE(DEDt && pos && vel && info && hash && cellStart && neibsList, "sphx_forces_internal_energy: missing buffer");
	SPHX_REQUIRE(fromParticle <= toParticle && toParticle <= numParticles, "sphx_forces_internal_energy: empty or inverted range");
	if (fromParticle == toParticle) return SPHX_OK;
	{ const int rc0 = sphx_ensure_scratch(ctx, numParticles); if (rc0 != SPHX_OK) return rc0; }
	sphx_fidelity_rows_launch(ctx, vel, info, numParticles, (hipStream_t)stream);
	EnergyArgs a = {};
	a.row = ctx->eos_aux;
	a.dedt = DEDt; a.pos = (const float4*)pos; a.vel = (const float4*)vel; a.info = (const particleinfo*)info;
	a.hash = hash; a.cellStart = cellStart; a.neibsList = neibsList; a.fromParticle = fromParticle; a.toParticle = toParticle;
	internal_energy_kernel<<<div_up_u(toParticle - fromParticle, 128), 128, 0, (hipStream_t)stream>>>(ctx->dev, a);
	SPHX_LAUNCH_CHECK("internal_energy_kernel");
	return SPHX_OK;
}

__global__ void __launch_bounds__(256)
euler_energy_kernel(DevParams p, float *newEnergy, const float *oldEnergy, const float *dedt, const float4 *oldPos,
	const particleinfo *info, const float *d_dt, float dt, float dt_scale, uint32_t n)
{
	const uint32_t i = blockIdx.x*256 + threadIdx.x;
	if (i >= n) return;
	const float h = d_dt ? d_dt[0]*dt_scale : dt;
	float e = oldEnergy[i];
	const uint32_t ptype = PART_TYPE(info[i]);
	if (is_active_w(oldPos[i].w) && (ptype == PT_FLUID || ((ptype == PT_BOUNDARY || ptype == PT_VERTEX) && p.boundarytype == SPHX_DYN_BOUNDARY)))
		e = fmaf(h, dedt[i], e);
	newEnergy[i] = e;
}

extern "C" int sphx_euler_internal_energy(sphx_ctx *ctx, float *newEnergy, const float *oldEnergy, const float *DEDt,
	const void *oldPos, const void *info, uint32_t numParticles, uint32_t particleRangeEnd,
	float dt, const float *d_dt, float dt_scale, void *stream)
{
	(void)numParticles;
	SPHX_REQUIRE(ctx && ctx->have_params, "sphx_euler_internal_energy: constants not set");
	if (!(ctx->params.simflags & SPHX_ENABLE_INTERNAL_ENERGY))
		return sphx_set_error(SPHX_ERR_INVALID, "sphx_euler_internal_energy called without ENABLE_INTERNAL_ENERGY");
	SPHX_REQUIRE(newEnergy && oldEnergy && DEDt && oldPos && info, "sphx_euler_internal_energy: missing buffer");
	if (!particleRangeEnd) return SPHX_OK;
	euler_energy_kernel<<<div_up_u(particleRangeEnd, 256), 256, 0, (hipStream_t)stream>>>(ctx->dev, newEnergy, oldEnergy, DEDt,
		(const float4*)oldPos, (const particleinfo*)info, d_dt, dt, dt_scale, particleRangeEnd);
	SPHX_LAUNCH_CHECK("euler_energy_kernel");
	return SPHX_OK;
}
