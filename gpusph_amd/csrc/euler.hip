// euler.hip -- predictor/corrector Euler update for gfx950.
// Replaces CUDAPredCorrEngine::basicstep (GPUSPH src/cuda/euler.cu:329-366) and eulerDevice
// (src/cuda/euler_kernel.def:396-538).  Pure streaming: 60 B read + 32 B written per particle.
// The adaptive dt can be read from a device scalar (written by dt_final_kernel) so the
// corrector never waits for a host round trip.
// Numerics: -ffp-contract=off, the a + b*c updates are explicit fmaf (DESIGN.md "Numerics");
// bit-identical to oracle/sph_oracle.c.
#include "sphx_internal.h"

#define BLOCK_EULER 256

struct EulerArgs {
	float4 *newPos, *newVel;
	const float4 *oldPos, *oldVel, *forces;
	const float4 *xsph;      // ENABLE_XSPH: mean neighbourhood velocity from the forces pass, else NULL
	float4 *newVol;          // SPH_GRENIER: BUFFER_VOLUME (x initial volume, y log of current/initial, w current), else NULL
	const float4 *oldVol;
	const particleinfo *info;
	const uint32_t *hash;
	const RbParams *rb;
	const float *d_dt;
	float dt, dt_scale;
	uint32_t numParticles;
	float4 *eosRows;         // the forces engine's EOS rows of the new densities (sphx_eos_rows_follow_euler), else NULL
};

// REPACK = eulerDevice with euler_repack_params (src/cuda/euler_params.h:203, euler.cu:346-353): boundaries are not
// integrated, bodies do not move, the density is not evolved
template<int STEP, bool REPACK>
__global__ void __launch_bounds__(BLOCK_EULER)
euler_kernel(DevParams p, EulerArgs a)
{
	const uint32_t index = blockIdx.x*BLOCK_EULER + threadIdx.x;
	if (index >= a.numParticles) return;

	const float dt = a.d_dt ? a.d_dt[0]*a.dt_scale : a.dt;

	const particleinfo info = a.info[index];
	const uint32_t ptype = PART_TYPE(info);
	const float4 force = a.forces[index];
	float4 pos = a.oldPos[index];
	float4 vel = a.oldVel[index];
	// SPH_GRENIER: the continuity equation integrates the log of the volume ratio instead of the density
	// (continuity_integration euler_kernel.def:210-216, write_volume :281-289)
	const bool grenier = !REPACK && a.oldVol != nullptr;
	float4 vol = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
	if (grenier) vol = a.oldVol[index];

	const bool integrateBoundary = !REPACK && (p.boundarytype == SPHX_DYN_BOUNDARY || p.boundarytype == SPHX_SA_BOUNDARY);
	if (is_active_w(pos.w) && !(ptype == PT_BOUNDARY && !integrateBoundary && !IS_MOVING(info))) {
		// standard_corrected_velocity (euler_kernel.def:147-169)
		float vcx = vel.x, vcy = vel.y, vcz = vel.z;
		if (STEP == 2) {
			const float hdt = dt/2;
			vcx = fmaf(force.x, hdt, vcx);
			vcy = fmaf(force.y, hdt, vcy);
			vcz = fmaf(force.z, hdt, vcz);
		}
		if (!REPACK && a.xsph) {   // XSPH correction of the advecting velocity (compute_corrected_velocity :171-180)
			const float4 xs = a.xsph[index];
			vcx = fmaf(p.epsxsph, xs.x, vcx);
			vcy = fmaf(p.epsxsph, xs.y, vcy);
			vcz = fmaf(p.epsxsph, xs.z, vcz);
		}
		if (ptype == PT_FLUID) {
			pos.x = fmaf(vcx, dt, pos.x);
			pos.y = fmaf(vcy, dt, pos.y);
			pos.z = fmaf(vcz, dt, pos.z);
			if (grenier) vol.y = fmaf(dt, force.w, vol.y);
			else if (!REPACK) vel.w = fmaf(dt, force.w, vel.w);   // continuity_integration :203-209
			vel.x = fmaf(dt, force.x, vel.x);
			vel.y = fmaf(dt, force.y, vel.y);
			vel.z = fmaf(dt, force.z, vel.z);
		} else if (ptype == PT_BOUNDARY || ptype == PT_VERTEX) {
			if (!REPACK && IS_MOVING(info)) { // rigid motion, euler_kernel.def:470-497, applyrot euler_kernel.cu:67-74
				const uint32_t obj = OBJECT_NUM(info);
				const int3 gp = grid_pos_from_hash(p, a.hash[index] & CELLTYPE_BITMASK);
				const float rx = (gp.x - a.rb->cgGridPosE[obj][0])*p.cs[0] + (pos.x - a.rb->cgPosE[obj][0]);
				const float ry = (gp.y - a.rb->cgGridPosE[obj][1])*p.cs[1] + (pos.y - a.rb->cgPosE[obj][1]);
				const float rz = (gp.z - a.rb->cgGridPosE[obj][2])*p.cs[2] + (pos.z - a.rb->cgPosE[obj][2]);
				const float *rot = a.rb->steprot[obj];
				pos.x += (rot[0] - 1.0f)*rx + rot[1]*ry + rot[2]*rz;
				pos.y += rot[3]*rx + (rot[4] - 1.0f)*ry + rot[5]*rz;
				pos.z += rot[6]*rx + rot[7]*ry + (rot[8] - 1.0f)*rz;
				pos.x += a.rb->trans[obj][0];
				pos.y += a.rb->trans[obj][1];
				pos.z += a.rb->trans[obj][2];
				const float *w = a.rb->angularvel[obj];
				vel.x = a.rb->linearvel[obj][0] + (w[1]*rz - w[2]*ry);
				vel.y = a.rb->linearvel[obj][1] + (w[2]*rx - w[0]*rz);
				vel.z = a.rb->linearvel[obj][2] + (w[0]*ry - w[1]*rx);
			}
			if (p.boundarytype == SPHX_DYN_BOUNDARY) {
				if (grenier) vol.y = fmaf(dt, force.w, vol.y);
				else vel.w = fmaf(dt, force.w, vel.w);
			}
		}
	}
	a.newPos[index] = pos;
	a.newVel[index] = vel;
	if (!REPACK && a.eosRows) {
		// {P/rho^2, c, P, rho} of the density just written, as eos_kernel (forces.hip) makes them from this very row: the forces
		// pass that follows reads them without a pass of its own over the velocity buffer
		const uint32_t fl = (p.numfluids > 1) ? FLUID_NUM(info) : 0u;
		const float ratio = vel.w + 1.0f;
		const float P = p.bcoeff[fl]*(powf(ratio, p.gammacoeff[fl]) - 1.0f);
		float c = p.sscoeff[fl]*powf(ratio, p.sspowercoeff[fl]);
		if (p.numfluids > 1) c = __uint_as_float((__float_as_uint(c) & ~3u) | fl);
		const float rho = ratio*p.rho0[fl];
		a.eosRows[index] = make_float4(P/(rho*rho), c, P, rho);
	}
	if (grenier) {
		vol.w = expf(vol.y)*vol.x;
		a.newVol[index] = vol;
	}
}

static int euler_launch(sphx_ctx *ctx, void *newPos, void *newVel, void *newVol,
	const void *oldPos, const void *oldVel, const void *oldVol, const void *info, const uint32_t *hash,
	const void *forces, const void *xsph,
	uint32_t numParticles, uint32_t particleRangeEnd,
	float dt, const float *d_dt, float dt_scale, int step, float t,
	float slength, float influenceradius, int run_mode, void *stream)
{
	(void)t; (void)slength; (void)influenceradius; (void)numParticles;
	SPHX_REQUIRE(ctx && ctx->have_params, "sphx_euler_basicstep: constants not set");
	SPHX_REQUIRE(newPos && newVel && oldPos && oldVel && info && hash && forces, "sphx_euler_basicstep: missing buffer");
	// SA_BOUNDARY with moving bodies: this step moves boundary elements, so what was kept of |grad gamma_as| per list entry
	// (ctx->sa_wall_cache, sa_wall.hip) belongs to a state of the elements that is gone: a new generation of the rows
	if (ctx->params.boundarytype == SPHX_SA_BOUNDARY && (ctx->params.simflags & SPHX_ENABLE_MOVING_BODIES)) {
		++ctx->sa_wall_gen;
		if (!ctx->sa_wall_gen) ctx->sa_wall_gen = 1u;
	}
	// SA_BOUNDARY: the fluid integrates like everywhere else, walls are copied; gamma follows in sphx_sa_integrate_gamma or,
	// with density summation (FORCES.w is zero then), density and gamma in sphx_sa_density_sum
	if ((ctx->dev.simflags & SPHX_ENABLE_XSPH) && run_mode == SPHX_SIMULATE)
		SPHX_REQUIRE(xsph != nullptr, "sphx_euler_basicstep: ENABLE_XSPH needs the XSPH buffer");
	if (run_mode != SPHX_SIMULATE && run_mode != SPHX_REPACK)
		return sphx_set_error(SPHX_ERR_INVALID, "sphx_euler_basicstep: invalid run mode");
	if (step != 1 && step != 2)
		return sphx_set_error(SPHX_ERR_INVALID, "unsupported predcorr timestep"); // src/cuda/euler.cu:361
	if (!particleRangeEnd) return SPHX_OK;
	{ const int rcf = sphx_rb_flush(ctx, (hipStream_t)stream); if (rcf != SPHX_OK) return rcf; }
	EulerArgs a;
	a.newPos = (float4*)newPos; a.newVel = (float4*)newVel;
	a.oldPos = (const float4*)oldPos; a.oldVel = (const float4*)oldVel; a.forces = (const float4*)forces;
	a.info = (const particleinfo*)info; a.hash = hash; a.rb = ctx->rb_dev;
	a.xsph = (ctx->dev.simflags & SPHX_ENABLE_XSPH) ? (const float4*)xsph : nullptr;
	a.newVol = (float4*)newVol; a.oldVol = (const float4*)oldVol;
	a.d_dt = d_dt; a.dt = dt; a.dt_scale = dt_scale; a.numParticles = particleRangeEnd;
	const dim3 grid(div_up_u(particleRangeEnd, BLOCK_EULER));
	const bool repack = run_mode == SPHX_REPACK;
	// the EOS rows ride along when the caller asked for it and the rows are the forces engine's to use (sphx_forces_basicstep's
	// option sets; the other engines keep other things in that scratch)
	a.eosRows = nullptr;
	if (ctx->eos_follow && !repack && !newVol && ctx->eos_aux && particleRangeEnd <= ctx->reserved_particles &&
	    (ctx->dev.boundarytype == SPHX_DYN_BOUNDARY || ctx->dev.boundarytype == SPHX_LJ_BOUNDARY))
		a.eosRows = ctx->eos_aux;
	ctx->eos_tag_vel = a.eosRows ? newVel : nullptr;
	ctx->eos_tag_n = a.eosRows ? particleRangeEnd : 0u;
	ctx->eos_armed = false;
	if (step == 1) {
		if (repack) euler_kernel<1, true><<<grid, BLOCK_EULER, 0, (hipStream_t)stream>>>(ctx->dev, a);
		else euler_kernel<1, false><<<grid, BLOCK_EULER, 0, (hipStream_t)stream>>>(ctx->dev, a);
	} else {
		if (repack) euler_kernel<2, true><<<grid, BLOCK_EULER, 0, (hipStream_t)stream>>>(ctx->dev, a);
		else euler_kernel<2, false><<<grid, BLOCK_EULER, 0, (hipStream_t)stream>>>(ctx->dev, a);
	}
	SPHX_LAUNCH_CHECK("euler_kernel");
	return SPHX_OK;
}

extern "C" int sphx_euler_basicstep(sphx_ctx *ctx, void *newPos, void *newVel,
	const void *oldPos, const void *oldVel, const void *info, const uint32_t *hash,
	const void *forces, const void *xsph,
	uint32_t numParticles, uint32_t particleRangeEnd,
	float dt, const float *d_dt, float dt_scale, int step, float t,
	float slength, float influenceradius, int run_mode, void *stream)
{
	if (ctx && ctx->have_params && ctx->params.sph_formulation == SPHX_SPH_GRENIER && run_mode == SPHX_SIMULATE)
		return sphx_set_error(SPHX_ERR_INVALID, "sphx_euler_basicstep: SPH_GRENIER integrates BUFFER_VOLUME, use sphx_euler_basicstep_grenier");
	return euler_launch(ctx, newPos, newVel, nullptr, oldPos, oldVel, nullptr, info, hash, forces, xsph, numParticles, particleRangeEnd,
		dt, d_dt, dt_scale, step, t, slength, influenceradius, run_mode, stream);
}

// The EOS rows of the forces engine follow the Euler step (on != 0) or are made by a pass of the forces engine (0, the default).
// With it on, a caller that knows that the velocity buffer it is about to hand to sphx_forces_basicstep is exactly what the
// last sphx_euler_basicstep of this context wrote, or what the last sphx_forces_basicstep read (a second stripe of one pass) --
// nothing has changed a density since: no filter, no boundary-condition pass, no sort, no import of halo rows, no upload --
// says so with sphx_eos_rows_current right before that call, and the forces pass skips its EOS pre-pass (0.19 ms of 10.9 per
// pass at 32 M particles).  The statement holds for one call.  A buffer or a row count other than the one the rows were made
// for is ignored (the pre-pass runs).
extern "C" int sphx_eos_rows_follow_euler(sphx_ctx *ctx, int on)
{
	SPHX_REQUIRE(ctx != nullptr, "sphx_eos_rows_follow_euler: NULL ctx");
	ctx->eos_follow = on != 0;
	ctx->eos_tag_vel = nullptr; ctx->eos_tag_n = 0; ctx->eos_armed = false;
	return SPHX_OK;
}

extern "C" int sphx_eos_rows_current(sphx_ctx *ctx, const void *vel, uint32_t numParticles)
{
	SPHX_REQUIRE(ctx != nullptr, "sphx_eos_rows_current: NULL ctx");
	ctx->eos_armed = ctx->eos_follow && vel != nullptr && vel == ctx->eos_tag_vel && numParticles == ctx->eos_tag_n;
	return SPHX_OK;
}

// SPH_GRENIER: the same step with BUFFER_VOLUME read (old) and written (new) (euler_params.h:153-156 Vol_params)
extern "C" int sphx_euler_basicstep_grenier(sphx_ctx *ctx, void *newPos, void *newVel, void *newVol,
	const void *oldPos, const void *oldVel, const void *oldVol, const void *info, const uint32_t *hash,
	const void *forces, const void *xsph,
	uint32_t numParticles, uint32_t particleRangeEnd,
	float dt, const float *d_dt, float dt_scale, int step, float t,
	float slength, float influenceradius, int run_mode, void *stream)
{
	SPHX_REQUIRE(ctx && ctx->have_params, "sphx_euler_basicstep_grenier: constants not set");
	if (ctx->params.sph_formulation != SPHX_SPH_GRENIER)
		return sphx_set_error(SPHX_ERR_INVALID, "sphx_euler_basicstep_grenier called without SPH_GRENIER");
	if (run_mode == SPHX_SIMULATE)
		SPHX_REQUIRE(newVol && oldVol && newVol != oldVol, "sphx_euler_basicstep_grenier: BUFFER_VOLUME is double buffered");
	return euler_launch(ctx, newPos, newVel, newVol, oldPos, oldVel, oldVol, info, hash, forces, xsph, numParticles, particleRangeEnd,
		dt, d_dt, dt_scale, step, t, slength, influenceradius, run_mode, stream);
}

// update_normals (src/cuda/euler_kernel.def:237-254), the part of eulerDevice that exists with SA_BOUNDARY and ENABLE_MOVING_BODIES:
// BUFFER_BOUNDELEMENTS of the new state = that of step n with the normals of the moving segments and vertices turned by the
// body's rotation over the step (applyrot, euler_kernel.cu:67-74); everything else copied.  A launch of its own behind
// sphx_euler_basicstep: the Euler entry point keeps the signature the other boundary models use.
__global__ void __launch_bounds__(BLOCK_EULER)
sa_update_normals_kernel(float4 *__restrict__ newBoundElem, const float4 *__restrict__ oldBoundElem, const particleinfo *__restrict__ info,
	const RbParams *__restrict__ rb, uint32_t n)
{
	const uint32_t index = blockIdx.x*BLOCK_EULER + threadIdx.x;
	if (index >= n) return;
	const particleinfo pi = info[index];
	float4 normal = oldBoundElem[index];
	if (IS_MOVING(pi) && (PART_TYPE(pi) == PT_BOUNDARY || PART_TYPE(pi) == PT_VERTEX)) {
		const float *rot = rb->steprot[OBJECT_NUM(pi)];
		const float rx = normal.x, ry = normal.y, rz = normal.z;
		normal.x += (rot[0] - 1.0f)*rx + rot[1]*ry + rot[2]*rz;
		normal.y += rot[3]*rx + (rot[4] - 1.0f)*ry + rot[5]*rz;
		normal.z += rot[6]*rx + rot[7]*ry + (rot[8] - 1.0f)*rz;
	}
	newBoundElem[index] = normal;
}

extern "C" int sphx_sa_update_normals(sphx_ctx *ctx, void *newBoundElements, const void *oldBoundElements, const void *info,
	uint32_t numParticles, uint32_t particleRangeEnd, void *stream)
{
	(void)numParticles;
	SPHX_REQUIRE(ctx && ctx->have_params, "sphx_sa_update_normals: constants not set");
	if (ctx->params.boundarytype != SPHX_SA_BOUNDARY || !(ctx->params.simflags & SPHX_ENABLE_MOVING_BODIES))
		return sphx_set_error(SPHX_ERR_INVALID, "sphx_sa_update_normals: for SA_BOUNDARY with ENABLE_MOVING_BODIES");
	SPHX_REQUIRE(newBoundElements && oldBoundElements && info && newBoundElements != oldBoundElements,
		"sphx_sa_update_normals: BUFFER_BOUNDELEMENTS is double buffered with moving bodies");
	if (!particleRangeEnd) return SPHX_OK;
	{ const int rcf = sphx_rb_flush(ctx, (hipStream_t)stream); if (rcf != SPHX_OK) return rcf; }
	++ctx->sa_wall_gen;      // the elements turn: rows of |grad gamma_as| kept before this call are of another state (sa_wall.hip)
	if (!ctx->sa_wall_gen) ctx->sa_wall_gen = 1u;
	sa_update_normals_kernel<<<div_up_u(particleRangeEnd, BLOCK_EULER), BLOCK_EULER, 0, (hipStream_t)stream>>>((float4*)newBoundElements,
		(const float4*)oldBoundElements, (const particleinfo*)info, ctx->rb_dev, particleRangeEnd);
	SPHX_LAUNCH_CHECK("sa_update_normals_kernel");
	return SPHX_OK;
}

// disableFreeSurfPartsDevice (src/cuda/euler_kernel.cu:158-180)
__global__ void __launch_bounds__(BLOCK_EULER)
disable_free_surf_kernel(float4 *pos, const particleinfo *info, uint32_t n)
{
	const uint32_t index = blockIdx.x*BLOCK_EULER + threadIdx.x;
	if (index >= n) return;
	const particleinfo pi = info[index];
	if (IS_SURFACE(pi) && PART_TYPE(pi) != PT_FLUID) {
		float4 p = pos[index];
		if (is_active_w(p.w)) {
			p.w = __builtin_nanf("");
			pos[index] = p;
		}
	}
}

extern "C" int sphx_disable_free_surf_parts(sphx_ctx *ctx, void *pos, const void *info,
	uint32_t numParticles, uint32_t particleRangeEnd, void *stream)
{
	(void)numParticles;
	SPHX_REQUIRE(ctx && pos && info, "sphx_disable_free_surf_parts: missing buffer");
	if (!particleRangeEnd) return SPHX_OK;
	disable_free_surf_kernel<<<div_up_u(particleRangeEnd, BLOCK_EULER), BLOCK_EULER, 0, (hipStream_t)stream>>>(
		(float4*)pos, (const particleinfo*)info, particleRangeEnd);
	SPHX_LAUNCH_CHECK("disable_free_surf_kernel");
	return SPHX_OK;
}

// TIME_STEP_EPILOGUE of a run whose dt lives on the device: t += dt as GPUSPH::runSimulation adds them (double += float,
// src/GPUSPH.cc:650-657), without a trip to the host
static __global__ void time_advance_kernel(double *t, const float *dt) { *t += (double)*dt; }

extern "C" int sphx_time_advance(sphx_ctx *ctx, double *d_t, const float *d_dt, void *stream)
{
	SPHX_REQUIRE(ctx && d_t && d_dt, "sphx_time_advance: missing buffer");
	time_advance_kernel<<<1, 1, 0, (hipStream_t)stream>>>(d_t, d_dt);
	SPHX_LAUNCH_CHECK("time_advance_kernel");
	return SPHX_OK;
}
