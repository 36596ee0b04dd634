// rheology.hip -- generalized Newtonian rheologies (RheologyType BINGHAM .. ZHU, src/visc_spec.h:44-56) for gfx950.
//   sphx_calc_effvisc              CUDAViscEngine::calc_visc for NEEDS_EFFECTIVE_VISC rheologies (src/cuda/visc.cu:86-170),
//                                  effectiveViscDevice (src/cuda/visc_kernel.cu:655-713): shear rate norm of every particle from
//                                  its neighbours, effective viscosity = shear term + (regularised) yield term, clamped;
//                                  BUFFER_EFFVISC holds mu_eff (compvisc DYNAMIC) or mu_eff/rho (KINEMATIC); the largest
//                                  kinematic value comes back for the viscous dt limit
//   sphx_forces_basicstep_effvisc  basicstep of the forces engine with effective_visc_forces_params (BUFFER_EFFVISC read per
//                                  particle and per neighbour, get_laminar_visc_coeff src/cuda/forces_kernel.def:250-259)
// One thread per particle walking its own u16 list (neib_iter.h), like the other fidelity engines: the reference's pair terms
// in the reference's order, exact powf / expf / division.  Checked against oracle/sph_oracle.c at fp32 tolerance
// (tests/test_gpu_rheology.py).  Built for SPH_F1, DYN_BOUNDARY, LAMINAR_FLOW + MORRIS, density diffusion NONE / COLAGROSSI /
// FERRARI, every kernel function; no bodies with force feedback.
#include "neib_iter.h"

// F<kerneltype>(r, h) = (1/r) dW/dr: src/cuda/sph_core.cu:140-215 (same forms as kernel_F of forces.hip, IEEE division)
template<int KERNEL>
__device__ __forceinline__ float gn_kernel_F(const DevParams &p, float r)
{
	const float R = r/p.slength;
	if (KERNEL == SPHX_WENDLAND) {
		const float qm2 = R - 2.0f;
		return qm2*qm2*qm2*p.fcoeff;
	}
	if (KERNEL == SPHX_CUBICSPLINE) {
		float val;
		if (R < 1.0f) val = (-4.0f + 3.0f*R)/p.slength;   // unused for R >= 2
		else val = -(-2.0f + R)*(-2.0f + R)/r;
		return val*p.fcoeff;
	}
	if (KERNEL == SPHX_QUADRATIC)
		return (-2.0f + R)/r*p.fcoeff;
	return -expf(-R*R)*p.fcoeff;
}

__device__ __forceinline__ float gn_F(const DevParams &p, float r)
{
	switch (p.kerneltype) {
	case SPHX_CUBICSPLINE: return gn_kernel_F<SPHX_CUBICSPLINE>(p, r);
	case SPHX_QUADRATIC: return gn_kernel_F<SPHX_QUADRATIC>(p, r);
	case SPHX_GAUSSIAN: return gn_kernel_F<SPHX_GAUSSIAN>(p, r);
	default: return gn_kernel_F<SPHX_WENDLAND>(p, r);
	}
}

// horner_one_minus_exp_minus_over<8> (src/cuda/visc_kernel.cu:420-451): (1 - exp(-x))/x below x = 1
__device__ __forceinline__ float horner_one_minus_exp_minus_over8(float x)
{
	float inner = fmaf(x, -1.0f/(8 + 1.0f), 1.0f);
#pragma unroll
	for (int order = 7; order >= 2; --order) inner = fmaf(x*inner, -1.0f/(order + 1.0f), 1.0f);
	return fmaf(x*inner, -0.5f, 1.0f);
}

// viscShearTerm + viscYieldTerm + clamp_visc (src/cuda/visc_kernel.cu:454-569)
__device__ __forceinline__ float effective_visc_value(const DevParams &p, float S, uint32_t fluid)
{
	const int rh = p.rheology;
	float effvisc = 0.0f;
	if (p.visccoeff[fluid] != 0.0f) {
		if (rh >= SPHX_DEKEE_TURCOTTE) effvisc += p.visccoeff[fluid]*expf(-p.visc_nonlinear_param[fluid]*S);
		else if (rh >= SPHX_POWER_LAW) effvisc += p.visccoeff[fluid]*powf(S, p.visc_nonlinear_param[fluid] - 1);
		else effvisc += p.visccoeff[fluid];
	}
	if (p.yield_strength[fluid] != 0.0f) {
		const bool reg = rh == SPHX_PAPANASTASIOU || rh == SPHX_ALEXANDROU || rh == SPHX_ZHU;
		const bool yielding = rh > SPHX_NEWTONIAN && rh != SPHX_POWER_LAW && rh != SPHX_GRANULAR;
		if (reg) {
			const float m = p.visc_regularization_param[fluid];
			const float mx = m*S;
			float r;
			if (mx < 1) r = m*horner_one_minus_exp_minus_over8(mx);
			else r = (1 - expf(-mx))/S;
			effvisc += p.yield_strength[fluid]*r;
		} else if (yielding)
			effvisc += p.yield_strength[fluid]/S;
	}
	return fminf(effvisc, p.limiting_kinvisc*p.rho0[fluid]);
}

struct EffViscArgs {
	float *effvisc;
	uint32_t *maxKinvisc;      // float bits, atomicMax (non-negative values order like their bits)
	const float4 *pos, *vel;
	const particleinfo *info;
	const uint32_t *hash, *cellStart;
	const neibdata *neibsList;
	uint32_t numParticles;
};

__global__ void __launch_bounds__(128)
effective_visc_kernel(DevParams p, EffViscArgs a)
{
	const uint32_t index = blockIdx.x*128 + threadIdx.x;
	float kinvisc = 0.0f;
	if (index < a.numParticles) {
		const float4 pos = a.pos[index];
		if (is_active_w(pos.w)) {
			const float4 vel = a.vel[index];
			const uint32_t fluid = FLUID_NUM(a.info[index]);
			const int3 gridPos = grid_pos_from_hash(p, a.hash[index] & CELLTYPE_BITMASK);
			float dvx[3] = {0, 0, 0}, dvy[3] = {0, 0, 0}, dvz[3] = {0, 0, 0};
			// shearRate<MIXED_TENSOR> (:307-367): every neighbour, fluid then boundary (for_every_neib, non-SA)
			auto pair = [&](uint32_t j, const float4 &npos, float rx, float ry, float rz) {
				const float r = sqrtf(fmaf(rz, rz, fmaf(ry, ry, rx*rx)));
				if (!is_active_w(npos.w) || r >= p.influenceradius) return;
				const float4 nvel = a.vel[j];
				const float n_rho = (nvel.w + 1.0f)*p.rho0[FLUID_NUM(a.info[j])];
				const float vx = vel.x - nvel.x, vy = vel.y - nvel.y, vz = vel.z - nvel.z;
				const float weight = gn_F(p, r)*npos.w/n_rho;
				const float mx = rx*weight, my = ry*weight, mz = rz*weight;
				dvx[0] -= vx*mx; dvx[1] -= vx*my; dvx[2] -= vx*mz;
				dvy[0] -= vy*mx; dvy[1] -= vy*my; dvy[2] -= vy*mz;
				dvz[0] -= vz*mx; dvz[1] -= vz*my; dvz[2] -= vz*mz;
			};
			for_each_neib<PT_FLUID>(p, a, index, pos, gridPos, pair);
			for_each_neib<PT_BOUNDARY>(p, a, index, pos, gridPos, pair);
			const float txx = dvx[0], txy = dvx[1] + dvy[0], txz = dvx[2] + dvz[0];
			const float tyy = dvy[1], tyz = dvy[2] + dvz[1], tzz = dvz[2];
			float diag_terms = txx*txx + tyy*tyy + tzz*tzz;
			diag_terms *= 2.0f;
			const float off_terms = txy*txy + txz*txz + tyz*tyz;
			const float S = sqrtf(diag_terms + off_terms);
			const float effvisc = effective_visc_value(p, S, fluid);
			kinvisc = effvisc/((vel.w + 1.0f)*p.rho0[fluid]);
			a.effvisc[index] = (p.compvisc == SPHX_KINEMATIC) ? kinvisc : effvisc;
		}
	}
	// reduce_kinvisc: the reference pre-reduces per block into the CFL array and finishes with cflmax; one atomic per wave here
#pragma unroll
	for (int d = 32; d > 0; d >>= 1) kinvisc = fmaxf(kinvisc, __shfl_down(kinvisc, d));
	if ((threadIdx.x & 63u) == 0u && kinvisc > 0.0f) atomicMax(a.maxKinvisc, __float_as_uint(kinvisc));
}

static int gn_check(const sphx_ctx *ctx, const char *who)
{
	SPHX_REQUIRE(ctx && ctx->have_params, "sphx (generalized Newtonian): constants not set");
	if (ctx->params.rheologytype <= SPHX_NEWTONIAN || ctx->params.rheologytype == SPHX_GRANULAR)
		return sphx_set_error(SPHX_ERR_INVALID, who);
	return SPHX_OK;
}

extern "C" int sphx_calc_effvisc(sphx_ctx *ctx, float *effvisc, float *h_max_kinvisc,
	const void *pos, const void *vel, const void *info, const uint32_t *hash,
	const uint32_t *cellStart, const uint16_t *neibsList,
	uint32_t numParticles, uint32_t particleRangeEnd, float deltap, float slength, float influenceradius, void *stream)
{
	(void)numParticles; (void)deltap;
	int rc = gn_check(ctx, "sphx_calc_effvisc: the rheology needs no effective viscosity (NEEDS_EFFECTIVE_VISC)");
	if (rc != SPHX_OK) return rc;
	SPHX_REQUIRE(effvisc && pos && vel && info && hash && cellStart && neibsList, "sphx_calc_effvisc: missing buffer");
	SPHX_REQUIRE(slength == ctx->params.slength && influenceradius == ctx->params.influenceradius,
		"sphx_calc_effvisc: slength / influenceradius differ from the uploaded constants");
	hipStream_t st = (hipStream_t)stream;
	uint32_t *d_max = (uint32_t*)(ctx->dt_scratch + 1);      // scratch word next to the one of the synchronous dt reduction
	SPHX_HIP(hipMemsetAsync(d_max, 0, sizeof(uint32_t), st));
	if (particleRangeEnd) {
		EffViscArgs a = {};
		a.effvisc = effvisc; a.maxKinvisc = d_max; a.pos = (const float4*)pos; a.vel = (const float4*)vel;
		a.info = (const particleinfo*)info; a.hash = hash; a.cellStart = cellStart; a.neibsList = neibsList;
		a.numParticles = particleRangeEnd;
		effective_visc_kernel<<<div_up_u(particleRangeEnd, 128), 128, 0, st>>>(ctx->dev, a);
		SPHX_LAUNCH_CHECK("effective_visc_kernel");
	}
	if (h_max_kinvisc) {      // what calc_visc returns with ENABLE_DTADAPT (NaN otherwise, src/cuda/visc.cu:163-169)
		float m = 0.0f;
		SPHX_HIP(hipMemcpyAsync(&m, d_max, sizeof(float), hipMemcpyDeviceToHost, st));
		SPHX_HIP(hipStreamSynchronize(st));
		*h_max_kinvisc = (ctx->params.simflags & SPHX_ENABLE_DTADAPT) ? m : __builtin_nanf("");
	}
	return SPHX_OK;
}

// what the pair loop needs from a particle besides position and velocity, evaluated once per particle instead of once per pair
// (same expressions, same values): pressure, density, sound speed, viscosity (BUFFER_EFFVISC or the fluid's coefficient)
__global__ void __launch_bounds__(256)
fidelity_row_kernel(DevParams p, const float4 *__restrict__ vel, const particleinfo *__restrict__ info,
	const float *__restrict__ effvisc, float4 *__restrict__ row, uint32_t n)
{
	const uint32_t i = blockIdx.x*256 + threadIdx.x;
	if (i >= n) return;
	const uint32_t fl = FLUID_NUM(info[i]);
	const float rt = vel[i].w;
	row[i] = make_float4(sa_P(p, rt, fl), (rt + 1.0f)*p.rho0[fl], sa_sound_speed(p, rt, fl), effvisc ? effvisc[i] : p.visccoeff[fl]);
}

void sphx_fidelity_rows_launch(sphx_ctx *ctx, const void *vel, const void *info, uint32_t numParticles, hipStream_t st)
{
	ctx->eos_tag_vel = nullptr;      // the scratch rows change hands
	fidelity_row_kernel<<<div_up_u(numParticles, 256), 256, 0, st>>>(ctx->dev, (const float4*)vel, (const particleinfo*)info, nullptr, ctx->eos_aux, numParticles);
}

struct GnForcesArgs {
	float4 *forces;
	float *cfl;
	const float4 *pos, *vel, *row;
	const float *effvisc;
	const particleinfo *info;
	const uint32_t *hash, *cellStart;
	const neibdata *neibsList;
	uint32_t fromParticle, toParticle, cflOffset;
};

// forcesDevice<FLUID,FLUID>, <FLUID,BOUNDARY>, <BOUNDARY,FLUID> + finalizeforcesDevice for SPH_F1 / DYN_BOUNDARY with the
// per-particle viscosity (the terms of oracle/sph_oracle.c forces_pass, same order)
__global__ void __launch_bounds__(SPHX_BLOCK_FORCES)
gn_forces_kernel(DevParams p, GnForcesArgs a)
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
			const bool fluid = ptype == PT_FLUID;
			const float4 self = a.row[index];      // P, rho, c, viscosity
			const float p_rho = self.y;
			const float p_P = self.x;
			// SPH_HA (Hu & Adams): P instead of P/rho^2, the volumes V = m/rho in the pressure term, the particle's own mass in the
			// continuity equation (forces_kernel.def:458-468,2030-2046,2269-2286,2436-2448)
			const bool ha = p.formulation == SPHX_SPH_HA;
			const float p_precalc = ha ? p_P : p_P/(p_rho*p_rho);
			const float p_sspeed = self.z;
			const bool viscous = p.rheology != SPHX_INVISCID;
			const float p_visc = self.w;      // per particle (generalized Newtonian) or per fluid
			const float p_volume = pos.w/p_rho;
			const bool momentum = fluid || HAS_COMPUTE_FORCE(info);
			auto pair = [&](uint32_t j, const float4 &npos, float rx, float ry, float rz, bool nfluid) {
				if (!is_active_w(npos.w)) return;
				const float r = sqrtf(fmaf(rz, rz, fmaf(ry, ry, rx*rx)));
				if (r >= p.influenceradius) return;
				const float4 nvel = a.vel[j];
				const float vx = vel.x - nvel.x, vy = vel.y - nvel.y, vz = vel.z - nvel.z;
				const float vel_dot_pos = sa_dot3(vx, vy, vz, rx, ry, rz);
				const float f = gn_F(p, r);
				const uint32_t nfl = FLUID_NUM(a.info[j]);
				const float4 nrow = a.row[j];
				const float n_rho = nrow.y;
				const float nmass = npos.w;
				// compute_density_derivative: divergence of velocity + density diffusion (fluid neighbours only)
				float DrDt = nmass*vel_dot_pos*f;
				if (ha) DrDt = pos.w*vel_dot_pos*f;
				const float n_volume = nmass/n_rho;
				if (nfluid && p.densitydiff == SPHX_COLAGROSSI && nfl == fl) {
					const float gdotr = sa_dot3(p.gravity[0], p.gravity[1], p.gravity[2], rx, ry, rz);
					if (!(fabsf(p_P - nrow.x) < fabsf(gdotr*p_rho))) {      // same fluid: the neighbour's own pressure
						if (ha) DrDt -= p.densityDiffCoeff*p.sscoeff[fl]*(p_volume/n_volume - 1)*f*pos.w;      // :1954-1996
						else DrDt -= p.densityDiffCoeff*p.sscoeff[fl]*(n_rho/p_rho - 1)*f*nmass;
					}
				}
				if (nfluid && p.densitydiff == SPHX_FERRARI && ha) {      // :1639-1677, same fluid only; the reference's 1./V makes it double
					if (nfl == fl) {
						const float sqC0 = p.sscoeff[fl]*p.sscoeff[fl];
						const float grav_corr = -sa_dot3(p.gravity[0], p.gravity[1], p.gravity[2], rx, ry, rz)*p.rho0[fl]/sqC0;
						float fx = 0.0f, fy = 0.0f, fz = 0.0f;
						if (r > 1e-4f*p.slength) {
							const float sc = (float)((double)fmaxf(p_sspeed, nrow.z)*
								((double)pos.w*(1./(double)p_volume - (double)(1.0f/(1.0f*n_volume))) + (double)grav_corr)/(double)p_rho/(double)r);
							fx = sc*rx; fy = sc*ry; fz = sc*rz;
						}
						DrDt += p.densityDiffCoeff*nmass*sa_dot3(fx, fy, fz, rx, ry, rz)*f;
					}
				} else
				if (nfluid && p.densitydiff == SPHX_FERRARI) {
					const float sqC0 = p.sscoeff[fl]*p.sscoeff[fl];
					const float grav_corr = -sa_dot3(p.gravity[0], p.gravity[1], p.gravity[2], rx, ry, rz)*p.rho0[fl]/sqC0;
					float fx = 0.0f, fy = 0.0f, fz = 0.0f;
					if (r > 1e-4f*p.slength) {
						const float sc = fmaxf(p_sspeed, nrow.z)*(p_rho - n_rho + grav_corr)/p_rho/r;
						fx = sc*rx; fy = sc*ry; fz = sc*rz;
					}
					DrDt += p.densityDiffCoeff*nmass*sa_dot3(fx, fy, fz, rx, ry, rz)*f;
				}
				force.w += DrDt;
				if (!momentum) return;
				const float n_P = nrow.x;
				const float n_precalc = ha ? n_P : n_P/(n_rho*n_rho);
				float s = (p_precalc + n_precalc)*nmass*f;
				if (ha) s = (p_precalc*p_volume*p_volume + n_precalc*n_volume*n_volume)/pos.w*f;
				float dx = 0.0f, dy = 0.0f, dz = 0.0f;
				dx -= s*rx; dy -= s*ry; dz -= s*rz;
				if (viscous && p.viscmodel == SPHX_ESPANOL_REVENGA) {
					// Espanol & Revenga (:2650-2677): shear and bulk viscosity, along the relative velocity and the relative position
					const float pv = (p.compvisc == SPHX_KINEMATIC) ? p.visccoeff[fl]*p_rho : p.visccoeff[fl];
					const float nv = (p.compvisc == SPHX_KINEMATIC) ? p.visccoeff[nfl]*n_rho : p.visccoeff[nfl];
					const float pb = p.visc2coeff[fl], nb = p.visc2coeff[nfl];
					float avs, avb;
					if (p.avgop == SPHX_ARITHMETIC) { avs = (pv + nv)*0.5f; avb = (pb + nb)*0.5f; }
					else if (p.avgop == SPHX_HARMONIC) { avs = 2*pv*nv/(pv + nv); avb = 2*pb*nb/(pb + nb); }
					else { avs = sqrtf(pv*nv); avb = sqrtf(pb*nb); }
					const float visc_thirds = avs/3;
					const float coeff = nmass/(p_rho*n_rho)*f;
					const float pos_den = sa_dot3(rx, ry, rz, rx, ry, rz) + p.epsartvisc;
					const float cv = 5*visc_thirds - avb, cr = 5*(visc_thirds + avb)*vel_dot_pos/pos_den;
					dx += coeff*(cv*vx + cr*rx); dy += coeff*(cv*vy + cr*ry); dz += coeff*(cv*vz + cr*rz);
				} else if (viscous) {
					const float vf = sa_visc_avg(p, p_visc, nrow.w, p_rho, n_rho, nmass)*f;
					if (p.viscmodel == SPHX_MONAGHAN) {      // along the relative position, approaching pairs only (:2531-2562)
						const float den = sa_dot3(rx, ry, rz, rx, ry, rz) + p.epsartvisc;
						const float c = vel_dot_pos < 0 ? p.monaghan_visc_coeff*vel_dot_pos/den : 0.0f;
						dx += vf*(c*rx); dy += vf*(c*ry); dz += vf*(c*rz);
					} else {
						dx += vf*vx; dy += vf*vy; dz += vf*vz;
					}
				}
				force.x += dx; force.y += dy; force.z += dz;
			};
			for_each_neib<PT_FLUID>(p, a, index, pos, gridPos,
				[&](uint32_t j, const float4 &npos, float rx, float ry, float rz) { pair(j, npos, rx, ry, rz, true); });
			if (fluid)
				for_each_neib<PT_BOUNDARY>(p, a, index, pos, gridPos,
					[&](uint32_t j, const float4 &npos, float rx, float ry, float rz) { pair(j, npos, rx, ry, rz, false); });
			force.w /= p.rho0[fl];       // forces_fixup :3212-3218
			if (fluid) {
				force.x += p.gravity[0]; force.y += p.gravity[1]; force.z += p.gravity[2];
				if ((p.simflags & SPHX_ENABLE_PLANES) && p.numplanes) {
					// GeometryForce / PlaneForce (src/cuda/forces_kernel.cu:140-203) with mu = get_laminar_dyn_visc of the particle
					const float dynvisc = !viscous ? 0.0f : (p.compvisc == SPHX_KINEMATIC) ? p_visc*p_rho : p_visc;
					for (uint32_t k = 0; k < p.numplanes; ++k) {
						const float ddx = (gridPos.x - p.plane_gridpos[k][0])*p.cs[0] + (pos.x - p.plane_pos[k][0]);
						const float ddy = (gridPos.y - p.plane_gridpos[k][1])*p.cs[1] + (pos.y - p.plane_pos[k][1]);
						const float ddz = (gridPos.z - p.plane_gridpos[k][2])*p.cs[2] + (pos.z - p.plane_pos[k][2]);
						const float r = fabsf(ddx*p.plane_normal[k][0] + ddy*p.plane_normal[k][1] + ddz*p.plane_normal[k][2]);
						if (r < p.r0) {
							const float DvDt = p.dcoeff*(powf(p.r0/r, p.p1coeff) - powf(p.r0/r, p.p2coeff))/(r*r);
							const float qx = p.plane_normal[k][0]*r, qy = p.plane_normal[k][1]*r, qz = p.plane_normal[k][2]*r;
							force.x += DvDt*qx; force.y += DvDt*qy; force.z += DvDt*qz;
							const float dd = (vel.x*qx + vel.y*qy + vel.z*qz)/r, inv = 1.0f/r;
							const float coeff = -dynvisc*p.partsurf/(pos.w*r);
							force.x += coeff*(vel.x - (dd*qx)*inv); force.y += coeff*(vel.y - (dd*qy)*inv);
							force.z += coeff*(vel.z - (dd*qz)*inv);
						}
					}
				}
				if (p.simflags & SPHX_ENABLE_DTADAPT) {
					const float acc = sqrtf(fmaf(force.z, force.z, fmaf(force.y, force.y, force.x*force.x)));
					cflTerm = fmaxf(acc, p_sspeed*p_sspeed/p.slength);
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

// the kernel above for a SIMULATE pass: with BUFFER_EFFVISC (generalized Newtonian rheologies, SPH_F1 or SPH_HA) or without
// (effvisc = NULL: the SPH_HA formulation with a Newtonian or no viscosity, routed here by sphx_forces_basicstep)
int sphx_fidelity_forces_launch(sphx_ctx *ctx, void *forces, float *cfl,
	const void *pos, const void *vel, const void *info, const uint32_t *hash,
	const uint32_t *cellStart, const uint16_t *neibsList, const float *effvisc,
	uint32_t numParticles, uint32_t fromParticle, uint32_t toParticle,
	float slength, float influenceradius, uint32_t cflOffset, uint32_t *h_numBlocks, void *stream);

extern "C" int sphx_forces_basicstep_effvisc(sphx_ctx *ctx, void *forces, float *cfl,
	const void *pos, const void *vel, const void *info, const uint32_t *hash,
	const uint32_t *cellStart, const uint16_t *neibsList, const float *effvisc,
	uint32_t numParticles, uint32_t fromParticle, uint32_t toParticle,
	float deltap, float slength, float dtadaptfactor, float influenceradius,
	uint32_t cflOffset, int run_mode, int step, float dt, uint32_t *h_numBlocks, void *stream)
{
	(void)deltap; (void)dtadaptfactor; (void)step; (void)dt;
	int rc = gn_check(ctx, "sphx_forces_basicstep_effvisc called for a rheology without effective viscosity");
	if (rc != SPHX_OK) return rc;
	if (run_mode != SPHX_SIMULATE)
		return sphx_set_error(SPHX_ERR_INVALID, "sphx_forces_basicstep_effvisc: repacking runs use sphx_forces_basicstep (no viscous term there)");
	SPHX_REQUIRE(effvisc != nullptr, "sphx_forces_basicstep_effvisc: missing buffer");
	return sphx_fidelity_forces_launch(ctx, forces, cfl, pos, vel, info, hash, cellStart, neibsList, effvisc, numParticles, fromParticle,
		toParticle, slength, influenceradius, cflOffset, h_numBlocks, stream);
}

int sphx_fidelity_forces_launch(sphx_ctx *ctx, void *forces, float *cfl,
	const void *pos, const void *vel, const void *info, const uint32_t *hash,
	const uint32_t *cellStart, const uint16_t *neibsList, const float *effvisc,
	uint32_t numParticles, uint32_t fromParticle, uint32_t toParticle,
	float slength, float influenceradius, uint32_t cflOffset, uint32_t *h_numBlocks, void *stream)
{
	const sphx_params &q = ctx->params;
	if ((q.sph_formulation != SPHX_SPH_F1 && q.sph_formulation != SPHX_SPH_HA) || q.boundarytype != SPHX_DYN_BOUNDARY || q.turbmodel != SPHX_LAMINAR_FLOW)
		return sphx_set_error(SPHX_ERR_UNSUPPORTED, "sphx: generalized Newtonian rheologies, SPH_HA and the MONAGHAN / ESPANOL_REVENGA viscous models are built for DYN_BOUNDARY and LAMINAR_FLOW");
	if (q.simflags & (SPHX_ENABLE_XSPH | SPHX_ENABLE_MOVING_BODIES))
		return sphx_set_error(SPHX_ERR_UNSUPPORTED, "sphx: generalized Newtonian rheologies and SPH_HA are built without XSPH and without moving bodies");
	SPHX_REQUIRE(forces && pos && vel && info && hash && cellStart && neibsList, "sphx_forces_basicstep (SPH_HA / effective viscosity): missing buffer");
	SPHX_REQUIRE(fromParticle <= toParticle && toParticle <= numParticles, "sphx_forces_basicstep_effvisc: empty or inverted range");
	SPHX_REQUIRE(slength == q.slength && influenceradius == q.influenceradius,
		"sphx_forces_basicstep_effvisc: slength / influenceradius differ from the uploaded constants");
	hipStream_t st = (hipStream_t)stream;
	const uint32_t count = toParticle - fromParticle;
	const uint32_t blocks = div_up_u(count, SPHX_BLOCK_FORCES);
	const uint32_t numBlocks = round_up_u(blocks, 4u);
	if (h_numBlocks) *h_numBlocks = numBlocks;
	if (!count) return SPHX_OK;
	if (q.simflags & SPHX_ENABLE_DTADAPT) {
		SPHX_REQUIRE(cfl != nullptr, "sphx_forces_basicstep_effvisc: ENABLE_DTADAPT needs the CFL buffer");
		if (numBlocks > blocks) SPHX_HIP(hipMemsetAsync(cfl + cflOffset + blocks, 0, sizeof(float)*(numBlocks - blocks), st));
	}
	{ const int rc0 = sphx_ensure_scratch(ctx, numParticles); if (rc0 != SPHX_OK) return rc0; }
	// the rows are read for the neighbours as well: every particle, not just the range
	ctx->eos_tag_vel = nullptr;      // the scratch rows change hands
	fidelity_row_kernel<<<div_up_u(numParticles, 256), 256, 0, st>>>(ctx->dev, (const float4*)vel, (const particleinfo*)info, effvisc, ctx->eos_aux, numParticles);
	SPHX_LAUNCH_CHECK("fidelity_row_kernel");
	GnForcesArgs a = {};
	a.forces = (float4*)forces; a.cfl = cfl; a.pos = (const float4*)pos; a.vel = (const float4*)vel; a.effvisc = effvisc; a.row = ctx->eos_aux;
	a.info = (const particleinfo*)info; a.hash = hash; a.cellStart = cellStart; a.neibsList = neibsList;
	a.fromParticle = fromParticle; a.toParticle = toParticle; a.cflOffset = cflOffset;
	gn_forces_kernel<<<blocks, SPHX_BLOCK_FORCES, 0, st>>>(ctx->dev, a);
	SPHX_LAUNCH_CHECK("gn_forces_kernel");
	return SPHX_OK;
}
