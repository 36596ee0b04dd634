// filters.hip -- density filter engines for gfx950: Shepard and MLS corrections of rho~.
// Replaces CUDAFilterEngine<SHEPARD_FILTER|MLS_FILTER> (GPUSPH src/cuda/forces.cu:1008-1147) and the kernels
// shepardDevice / MlsDevice (src/cuda/forces_kernel.cu:418-505, 508-721).
//
// Filters run every N-th iteration (WaveTank: Shepard every 20), so they are not on the roofline path; they are
// written for fidelity: compiled without FMA contraction, IEEE division and sqrt, the reference's operation
// order, so that the result is bit-identical to the CPU oracle for the polynomial kernels.
#include "sphx_internal.h"
#include "neib_iter.h"
#include <cfloat>

struct FilterArgs {
	float4 *newVel;
	const float4 *pos, *vel;
	const particleinfo *info;
	const uint32_t *hash, *cellStart;
	const neibdata *neibsList;
	uint32_t numParticles;
};

template<int KERNEL>
__global__ void __launch_bounds__(128)
shepard_kernel(DevParams p, FilterArgs a)
{
	const uint32_t index = blockIdx.x*128 + threadIdx.x;
	if (index >= a.numParticles) return;
	const particleinfo info = a.info[index];
	const float4 pos = a.pos[index];
	if (!is_active_w(pos.w)) return;          // inactive: nothing is written (as the reference)
	float4 vel = a.vel[index];
	if (PART_TYPE(info) != PT_FLUID) { a.newVel[index] = vel; return; }
	const uint32_t fl = FLUID_NUM(info);
	float temp1 = pos.w*kernel_W<KERNEL>(p, 0.0f);
	float temp2 = temp1/((vel.w + 1.0f)*p.rho0[fl]);
	const int3 gridPos = grid_pos_from_hash(p, a.hash[index] & CELLTYPE_BITMASK);
	auto pair = [&](uint32_t j, const float4 &npos, float rx, float ry, float rz) {
		if (!is_active_w(npos.w)) return;
		const float r = sqrtf(rx*rx + ry*ry + rz*rz);
		const float neib_rho = (a.vel[j].w + 1.0f)*p.rho0[FLUID_NUM(a.info[j])];
		if (r < p.influenceradius) {
			const float w = kernel_W<KERNEL>(p, r)*npos.w;
			temp1 += w;
			temp2 += w/neib_rho;
		}
	};
	for_each_neib<PT_FLUID>(p, a, index, pos, gridPos, pair);
	if (p.boundarytype == SPHX_DYN_BOUNDARY)
		for_each_neib<PT_BOUNDARY>(p, a, index, pos, gridPos, pair);
	vel.w = (temp1/temp2)/p.rho0[fl] - 1.0f;
	a.newVel[index] = vel;
}

// symmetric 4x4 tensor helpers: src/cuda/tensor.cu:65-100 (det), :240-282 (dot, ddot, adjugate_row1)
struct SymTensor4 { float xx, xy, xz, xw, yy, yz, yw, zz, zw, ww; };

__device__ __forceinline__ float st4_det(const SymTensor4 &T)
{
	float ret = 0, M = 0;
	M += T.xx*(T.yy*T.zz - T.yz*T.yz);
	M -= T.xy*(T.xy*T.zz - T.xz*T.yz);
	M += T.xz*(T.xy*T.yz - T.xz*T.yy);
	ret += M*T.ww;
	M = 0;
	M += T.xx*(T.yy*T.zw - T.yz*T.yw);
	M -= T.xy*(T.xy*T.zw - T.xz*T.yw);
	M += T.xw*(T.xy*T.yz - T.xz*T.yy);
	ret -= M*T.zw;
	M = 0;
	M += T.xx*(T.yz*T.zw - T.zz*T.yw);
	M -= T.xz*(T.xy*T.zw - T.xz*T.yw);
	M += T.xw*(T.xy*T.zz - T.xz*T.yz);
	ret += M*T.yw;
	M = 0;
	M += T.xy*(T.yz*T.zw - T.zz*T.yw);
	M -= T.xz*(T.yy*T.zw - T.yz*T.yw);
	M += T.xw*(T.yy*T.zz - T.yz*T.yz);
	ret -= M*T.xw;
	return ret;
}
__device__ __forceinline__ float4 st4_adjugate_row1(const SymTensor4 &T)
{
	return make_float4(
		T.yy*T.zz*T.ww + T.yz*T.zw*T.yw + T.yw*T.yz*T.zw - T.yy*T.zw*T.zw - T.yz*T.yz*T.ww - T.yw*T.zz*T.yw,
		T.xy*T.zw*T.zw + T.yz*T.xz*T.ww + T.yw*T.zz*T.xw - T.xy*T.zz*T.ww - T.yz*T.zw*T.xw - T.yw*T.xz*T.zw,
		T.xy*T.yz*T.ww + T.yy*T.zw*T.xw + T.yw*T.xz*T.yw - T.xy*T.zw*T.yw - T.yy*T.xz*T.ww - T.yw*T.yz*T.xw,
		T.xy*T.zz*T.yw + T.yy*T.xz*T.zw + T.yz*T.yz*T.xw - T.xy*T.yz*T.zw - T.yy*T.zz*T.xw - T.yz*T.xz*T.yw);
}
__device__ __forceinline__ float4 st4_dot(const SymTensor4 &T, const float4 &v)
{
	return make_float4(
		T.xx*v.x + T.xy*v.y + T.xz*v.z + T.xw*v.w,
		T.xy*v.x + T.yy*v.y + T.yz*v.z + T.yw*v.w,
		T.xz*v.x + T.yz*v.y + T.zz*v.z + T.zw*v.w,
		T.xw*v.x + T.yw*v.y + T.zw*v.z + T.ww*v.w);
}
__device__ __forceinline__ float st4_ddot(const SymTensor4 &T, const float4 &v)
{
	return T.xx*v.x*v.x + T.yy*v.y*v.y + T.zz*v.z*v.z + T.ww*v.w*v.w +
		2*((T.xy*v.y + T.xw*v.w)*v.x + (T.yz*v.z + T.yw*v.w)*v.y + (T.xz*v.x + T.zw*v.w)*v.z);
}
__device__ __forceinline__ float f4_dot(const float4 &a, const float4 &b) { return a.x*b.x + a.y*b.y + a.z*b.z + a.w*b.w; }
__device__ __forceinline__ float4 f4_scale(const float4 &a, float s) { return make_float4(a.x*s, a.y*s, a.z*s, a.w*s); }
__device__ __forceinline__ float f4_hypot(const float4 &v)   // src/vector_math.h:1231-1240
{
	const float pm = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
	if (!pm) return 0;
	const float4 w = f4_scale(v, 1.0f/pm);
	return pm*sqrtf(f4_dot(w, w));
}

template<int KERNEL>
__global__ void __launch_bounds__(128)
mls_kernel(DevParams p, FilterArgs a)
{
	const uint32_t index = blockIdx.x*128 + threadIdx.x;
	if (index >= a.numParticles) return;
	const particleinfo info = a.info[index];
	const float4 pos = a.pos[index];
	if (!is_active_w(pos.w)) return;
	float4 vel = a.vel[index];
	const uint32_t fl = FLUID_NUM(info);
	const bool dyn = p.boundarytype == SPHX_DYN_BOUNDARY;
	const float inv_h = 1.0f/p.slength;
	SymTensor4 mls = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
	mls.xx = kernel_W<KERNEL>(p, 0.0f)*pos.w/((vel.w + 1.0f)*p.rho0[fl]);
	const int3 gridPos = grid_pos_from_hash(p, a.hash[index] & CELLTYPE_BITMASK);
	auto first = [&](uint32_t j, const float4 &npos, float rx, float ry, float rz) {
		if (!is_active_w(npos.w)) return;
		const float r = sqrtf(rx*rx + ry*ry + rz*rz);
		const float neib_rho = (a.vel[j].w + 1.0f)*p.rho0[FLUID_NUM(a.info[j])];
		if (r < p.influenceradius) {
			const float w = kernel_W<KERNEL>(p, r)*npos.w/neib_rho;   // Wij*Vj
			const float sx = rx*inv_h, sy = ry*inv_h, sz = rz*inv_h;     // relPos/slength
			mls.xx += w;
			mls.xy += sx*w; mls.xz += sy*w; mls.xw += sz*w;
			mls.yy += sx*sx*w; mls.yz += sx*sy*w; mls.yw += sx*sz*w;
			mls.zz += sy*sy*w; mls.zw += sy*sz*w; mls.ww += sz*sz*w;
		}
	};
	for_each_neib<PT_FLUID>(p, a, index, pos, gridPos, first);
	if (dyn) for_each_neib<PT_BOUNDARY>(p, a, index, pos, gridPos, first);

	// M B = E, E = (1,0,0,0): adjugate start + conjugate-residual refinement (forces_kernel.cu:602-656)
	const float4 E = make_float4(1, 0, 0, 0);
	const float D = st4_det(mls);
	float4 B;
	if (fabsf(D) < FLT_EPSILON) {
		SymTensor4 me = mls;
		const float eps = fabsf(D) + FLT_EPSILON;
		me.xx += eps; me.yy += eps; me.zz += eps; me.ww += eps;
		const float De = st4_det(me);
		B = f4_scale(st4_adjugate_row1(me), 1.0f/De);
	} else {
		B = f4_scale(st4_adjugate_row1(mls), 1.0f/D);
	}
	for (unsigned steps = 0; steps < 32; ++steps) {
		const float lenB = f4_hypot(B);
		const float4 MdotB = st4_dot(mls, B);
		const float4 residual = make_float4(E.x - MdotB.x, E.y - MdotB.y, E.z - MdotB.z, E.w - MdotB.w);
		const float num = st4_ddot(mls, residual);
		const float4 Mp = st4_dot(mls, residual);
		const float den = f4_dot(Mp, Mp);
		const float4 corr = f4_scale(residual, num/den);
		const float lencorr = f4_hypot(corr);
		if (f4_hypot(residual) < lenB*FLT_EPSILON) break;
		if (lencorr < 2*lenB*FLT_EPSILON) break;
		B.x += corr.x; B.y += corr.y; B.z += corr.z; B.w += corr.w;
	}
	B.y /= p.slength; B.z /= p.slength; B.w /= p.slength;

	float rho = B.x*kernel_W<KERNEL>(p, 0.0f)*pos.w;
	auto second = [&](uint32_t j, const float4 &npos, float rx, float ry, float rz) {
		if (!is_active_w(npos.w)) return;
		const float r = sqrtf(rx*rx + ry*ry + rz*rz);
		if (r < p.influenceradius && (dyn || PART_TYPE(a.info[j]) == PT_FLUID)) {
			const float w = kernel_W<KERNEL>(p, r)*npos.w;   // rho_j*Wij*Vj = mj*Wij
			rho += (B.x + B.y*rx + B.z*ry + B.w*rz)*w;
		}
	};
	for_each_neib<PT_FLUID>(p, a, index, pos, gridPos, second);
	if (dyn) for_each_neib<PT_BOUNDARY>(p, a, index, pos, gridPos, second);
	vel.w = rho/p.rho0[fl] - 1.0f;
	a.newVel[index] = vel;
}

// ==========================================================================================
// post-processing engines (run before writes): src/cuda/post_process_kernel.cu:58-392, host src/cuda/post_process.cu
// ==========================================================================================
// F<kerneltype>(r, h) with the reference's operations (src/cuda/sph_core.cu:146-191); the forces kernel has its own
// fast variant
template<int KERNEL>
__device__ __forceinline__ float kernel_F_exact(const DevParams &p, float r)
{
	const float R = r/p.slength;
	if (KERNEL == SPHX_CUBICSPLINE) {
		const float val = (R < 1.0f) ? (-4.0f + 3.0f*R)/p.slength : -(-2.0f + R)*(-2.0f + R)/r;
		return val*p.fcoeff;
	} else if (KERNEL == SPHX_QUADRATIC) {
		return ((-2.0f + R)/r)*p.fcoeff;
	} else if (KERNEL == SPHX_WENDLAND) {
		const float qm2 = r/p.slength - 2.0f;
		return qm2*qm2*qm2*p.fcoeff;
	} else {
		return -expf(-R*R)*p.fcoeff;
	}
}

struct PostArgs {
	float *vorticity;          // VORTICITY: 3 floats per particle
	float4 *velInOut;          // TESTPOINTS: updated in place
	particleinfo *infoInOut;   // SURFACE_DETECTION: updated in place
	float4 *normals;           // SURFACE_DETECTION, optional
	float cosconeanglefluid, cosconeanglenonfluid;
};

// calcVortDevice (:58-135)
template<int KERNEL>
__global__ void __launch_bounds__(128)
vorticity_kernel(DevParams p, FilterArgs a, PostArgs o)
{
	const uint32_t index = blockIdx.x*128 + threadIdx.x;
	if (index >= a.numParticles) return;
	const particleinfo info = a.info[index];
	const float4 pos = a.pos[index];
	float *out = o.vorticity + 3*(size_t)index;
	if (PART_TYPE(info) != PT_FLUID || !is_active_w(pos.w)) { out[0] = out[1] = out[2] = NAN; return; }
	const float4 vel = a.vel[index];
	float vx = 0.0f, vy = 0.0f, vz = 0.0f;
	const int3 gridPos = grid_pos_from_hash(p, a.hash[index] & CELLTYPE_BITMASK);
	for_each_neib<PT_FLUID>(p, a, index, pos, gridPos, [&](uint32_t j, const float4 &npos, float rx, float ry, float rz) {
		if (!is_active_w(npos.w)) return;
		const float r = sqrtf(rx*rx + ry*ry + rz*rz);
		const float4 nvel = a.vel[j];
		const float ux = vel.x - nvel.x, uy = vel.y - nvel.y, uz = vel.z - nvel.z;
		if (r < p.influenceradius) {
			const float f = kernel_F_exact<KERNEL>(p, r)*npos.w/((nvel.w + 1.0f)*p.rho0[FLUID_NUM(a.info[j])]);
			vx += f*(uy*rz - uz*ry);
			vy += f*(uz*rx - ux*rz);
			vz += f*(ux*ry - uy*rx);
		}
	});
	out[0] = vx; out[1] = vy; out[2] = vz;
}

// XSPH mean velocity of the forces pass (ENABLE_XSPH; compute_mean_vel forces_kernel.def:2986-2994, write_xsph :3366-3368):
// xsph_i = 2 * sum over fluid neighbours j of -m_j W(r_ij) (v_i - v_j) / (rho_i + rho_j), fluid particles only.  The reference
// accumulates it inside its fluid-fluid forces launch; here it is a pass of its own with this file's exact arithmetic (the
// option is rare: none of the reference's shipped problems enables it), which keeps the ~10 extra instructions per pair
// out of the hot kernels.  Same operations as the oracle (fmaf where nvcc contracts): bit-equal.
template<int KERNEL>
__global__ void __launch_bounds__(128)
xsph_kernel(DevParams p, FilterArgs a, float4 *__restrict__ xsph, uint32_t fromParticle)
{
	const uint32_t index = fromParticle + blockIdx.x*128 + threadIdx.x;
	if (index >= a.numParticles) return;
	const particleinfo info = a.info[index];
	const float4 pos = a.pos[index];
	if (PART_TYPE(info) != PT_FLUID || !is_active_w(pos.w)) return;
	const float4 vel = a.vel[index];
	const float rho = (vel.w + 1.0f)*p.rho0[FLUID_NUM(info)];
	float mx = 0.0f, my = 0.0f, mz = 0.0f;
	const int3 gridPos = grid_pos_from_hash(p, a.hash[index] & CELLTYPE_BITMASK);
	for_each_neib<PT_FLUID>(p, a, index, pos, gridPos, [&](uint32_t j, const float4 &npos, float rx, float ry, float rz) {
		if (!is_active_w(npos.w)) return;
		const float r = sqrtf(fmaf(rz, rz, fmaf(ry, ry, rx*rx)));
		if (r >= p.influenceradius) return;
		const float4 nvel = a.vel[j];
		const float n_rho = (nvel.w + 1.0f)*p.rho0[FLUID_NUM(a.info[j])];
		const float t = npos.w*kernel_W<KERNEL>(p, r);
		const float inv = 1.0f/(rho + n_rho);
		mx = fmaf(-(t*(vel.x - nvel.x)), inv, mx);
		my = fmaf(-(t*(vel.y - nvel.y)), inv, my);
		mz = fmaf(-(t*(vel.z - nvel.z)), inv, mz);
	});
	xsph[index] = make_float4(2.0f*mx, 2.0f*my, 2.0f*mz, 0.0f);
}

int sphx_xsph_launch(sphx_ctx *ctx, void *xsph, const void *pos, const void *vel, const void *info, const uint32_t *hash,
	const uint32_t *cellStart, const uint16_t *neibsList, uint32_t fromParticle, uint32_t toParticle, hipStream_t st)
{
	if (toParticle <= fromParticle) return SPHX_OK;
	FilterArgs a;
	a.newVel = nullptr; a.pos = (const float4*)pos; a.vel = (const float4*)vel;
	a.info = (const particleinfo*)info; a.hash = hash; a.cellStart = cellStart; a.neibsList = neibsList;
	a.numParticles = toParticle;
	const dim3 grid(div_up_u(toParticle - fromParticle, 128));
	switch (ctx->dev.kerneltype) {
	case SPHX_CUBICSPLINE: xsph_kernel<SPHX_CUBICSPLINE><<<grid, 128, 0, st>>>(ctx->dev, a, (float4*)xsph, fromParticle); break;
	case SPHX_QUADRATIC:   xsph_kernel<SPHX_QUADRATIC><<<grid, 128, 0, st>>>(ctx->dev, a, (float4*)xsph, fromParticle); break;
	case SPHX_WENDLAND:    xsph_kernel<SPHX_WENDLAND><<<grid, 128, 0, st>>>(ctx->dev, a, (float4*)xsph, fromParticle); break;
	case SPHX_GAUSSIAN:    xsph_kernel<SPHX_GAUSSIAN><<<grid, 128, 0, st>>>(ctx->dev, a, (float4*)xsph, fromParticle); break;
	default: return sphx_set_error(SPHX_ERR_INVALID, "sphx_forces_basicstep: invalid kernel type");
	}
	SPHX_LAUNCH_CHECK("xsph_kernel");
	return SPHX_OK;
}

// calcTestpointsVelocityDevice (:138-236), non-SA, no k-epsilon buffers
template<int KERNEL>
__global__ void __launch_bounds__(128)
testpoints_kernel(DevParams p, FilterArgs a, PostArgs o)
{
	const uint32_t index = blockIdx.x*128 + threadIdx.x;
	if (index >= a.numParticles) return;
	const particleinfo info = a.info[index];
	if (!IS_TESTPOINT(info)) return;
	const float4 pos = a.pos[index];
	float4 avg = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
	float alpha = 0.0f;
	const int3 gridPos = grid_pos_from_hash(p, a.hash[index] & CELLTYPE_BITMASK);
	for_each_neib<PT_FLUID>(p, a, index, pos, gridPos, [&](uint32_t j, const float4 &npos, float rx, float ry, float rz) {
		const float r = sqrtf(rx*rx + ry*ry + rz*rz);
		if (r < p.influenceradius) {
			const float4 nvel = o.velInOut[j];
			const uint32_t nfl = FLUID_NUM(a.info[j]);
			const float w = kernel_W<KERNEL>(p, r)*npos.w/((nvel.w + 1.0f)*p.rho0[nfl]);
			avg.x += w*nvel.x; avg.y += w*nvel.y; avg.z += w*nvel.z;
			avg.w += w*(p.bcoeff[nfl]*(powf(nvel.w + 1.0f, p.gammacoeff[nfl]) - 1.0f));
			alpha += w;
		}
	});
	if (alpha > 1e-5f) {
		const float inv = 1.0f/alpha;
		avg.x *= inv; avg.y *= inv; avg.z *= inv; avg.w *= inv;
	} else {
		avg = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
	}
	o.velInOut[index] = avg;
}

// calcSurfaceparticleDevice (:239-392), non-SA
template<int KERNEL>
__global__ void __launch_bounds__(128)
surface_kernel(DevParams p, FilterArgs a, PostArgs o)
{
	const uint32_t index = blockIdx.x*128 + threadIdx.x;
	if (index >= a.numParticles) return;
	particleinfo info = o.infoInOut[index];
	const float4 pos = a.pos[index];
	if (PART_TYPE(info) != PT_FLUID || !is_active_w(pos.w)) {
		if (o.normals) o.normals[index] = make_float4(NAN, NAN, NAN, NAN);
		return;
	}
	const int3 gridPos = grid_pos_from_hash(p, a.hash[index] & CELLTYPE_BITMASK);
	info.x &= (unsigned short)~FG_SURFACE;
	float4 normal = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
	normal.w = kernel_W<KERNEL>(p, 0.0f)*pos.w/((a.vel[index].w + 1.0f)*p.rho0[FLUID_NUM(info)]);
	auto first = [&](uint32_t j, const float4 &npos, float rx, float ry, float rz) {
		if (!is_active_w(npos.w)) return;
		const float r = sqrtf(rx*rx + ry*ry + rz*rz);
		const float neib_vol = npos.w/((a.vel[j].w + 1.0f)*p.rho0[FLUID_NUM(o.infoInOut[j])]);
		if (r < p.influenceradius) {
			const float f = kernel_F_exact<KERNEL>(p, r)*neib_vol;
			normal.x -= f*rx; normal.y -= f*ry; normal.z -= f*rz;
			normal.w += kernel_W<KERNEL>(p, r)*neib_vol;
		}
	};
	for_each_neib<PT_FLUID>(p, a, index, pos, gridPos, first);
	for_each_neib<PT_BOUNDARY>(p, a, index, pos, gridPos, first);
	if (p.simflags & SPHX_ENABLE_PLANES)
		for (uint32_t k = 0; k < p.numplanes; ++k) {
			const float dx = (gridPos.x - p.plane_gridpos[k][0])*p.cs[0] + (pos.x - p.plane_pos[k][0]);
			const float dy = (gridPos.y - p.plane_gridpos[k][1])*p.cs[1] + (pos.y - p.plane_pos[k][1]);
			const float dz = (gridPos.z - p.plane_gridpos[k][2])*p.cs[2] + (pos.z - p.plane_pos[k][2]);
			const float r = fabsf(dx*p.plane_normal[k][0] + dy*p.plane_normal[k][1] + dz*p.plane_normal[k][2]);
			if (r < p.influenceradius) {
				const float len = sqrtf(normal.x*normal.x + normal.y*normal.y + normal.z*normal.z);
				normal.x += p.plane_normal[k][0]*len; normal.y += p.plane_normal[k][1]*len; normal.z += p.plane_normal[k][2]*len;
			}
		}
	const float normal_length = sqrtf(normal.x*normal.x + normal.y*normal.y + normal.z*normal.z);
	int nc = 0;
	auto second = [&](uint32_t j, const float4 &npos, float rx, float ry, float rz) {
		if (!is_active_w(npos.w)) return;
		const float r = sqrtf(rx*rx + ry*ry + rz*rz);
		if (r < p.influenceradius) {
			const float criteria = -(normal.x*rx + normal.y*ry + normal.z*rz);
			const float cosconeangle = (PART_TYPE(o.infoInOut[j]) == PT_FLUID) ? o.cosconeanglefluid : o.cosconeanglenonfluid;
			if (criteria > r*normal_length*cosconeangle) nc++;
		}
	};
	for_each_neib<PT_FLUID>(p, a, index, pos, gridPos, second);
	for_each_neib<PT_BOUNDARY>(p, a, index, pos, gridPos, second);
	if (!nc) info.x |= FG_SURFACE;
	o.infoInOut[index] = info;
	if (o.normals) {
		normal.x /= normal_length; normal.y /= normal_length; normal.z /= normal_length;
		o.normals[index] = normal;
	}
}

// calcInterfaceparticleDevice (:388-560), non-SA: FG_SURFACE and FG_INTERFACE of a multi-fluid run.  Two SPH normals per
// particle (all neighbours / same-fluid and non-fluid neighbours); the gradient sums take the particle's own volume after
// the loop, planes do not enter (unlike the surface kernel)
template<int KERNEL>
__global__ void __launch_bounds__(128)
interface_kernel(DevParams p, FilterArgs a, PostArgs o)
{
	const uint32_t index = blockIdx.x*128 + threadIdx.x;
	if (index >= a.numParticles) return;
	particleinfo info = o.infoInOut[index];
	const float4 pos = a.pos[index];
	if (PART_TYPE(info) != PT_FLUID || !is_active_w(pos.w)) {
		if (o.normals) o.normals[index] = make_float4(NAN, NAN, NAN, NAN);
		return;
	}
	const int3 gridPos = grid_pos_from_hash(p, a.hash[index] & CELLTYPE_BITMASK);
	info.x &= (unsigned short)~(FG_SURFACE | FG_INTERFACE);
	const uint32_t fnum = FLUID_NUM(info);
	const float p_volume = pos.w/((a.vel[index].w + 1.0f)*p.rho0[fnum]);
	float4 nfs = make_float4(0.0f, 0.0f, 0.0f, 0.0f), nif = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
	nfs.w = kernel_W<KERNEL>(p, 0.0f)*p_volume;
	nif.w = kernel_W<KERNEL>(p, 0.0f)*p_volume;
	auto first = [&](uint32_t j, const float4 &npos, float rx, float ry, float rz) {
		if (!is_active_w(npos.w)) return;
		const particleinfo n_info = o.infoInOut[j];
		const float r = sqrtf(rx*rx + ry*ry + rz*rz);
		const float n_volume = npos.w/((a.vel[j].w + 1.0f)*p.rho0[FLUID_NUM(n_info)]);
		if (r < p.influenceradius) {
			const float f = kernel_F_exact<KERNEL>(p, r);
			nfs.x -= f*rx; nfs.y -= f*ry; nfs.z -= f*rz;
			nfs.w += kernel_W<KERNEL>(p, r)*n_volume;
		}
		if (r < p.influenceradius && (fnum == FLUID_NUM(n_info) || PART_TYPE(n_info) != PT_FLUID)) {
			const float f = kernel_F_exact<KERNEL>(p, r);
			nif.x -= f*rx; nif.y -= f*ry; nif.z -= f*rz;
			nif.w += kernel_W<KERNEL>(p, r)*n_volume;
		}
	};
	for_each_neib<PT_FLUID>(p, a, index, pos, gridPos, first);
	for_each_neib<PT_BOUNDARY>(p, a, index, pos, gridPos, first);
	nfs.x *= p_volume; nfs.y *= p_volume; nfs.z *= p_volume;
	nif.x *= p_volume; nif.y *= p_volume; nif.z *= p_volume;
	const float lfs = sqrtf(nfs.x*nfs.x + nfs.y*nfs.y + nfs.z*nfs.z);
	const float lif = sqrtf(nif.x*nif.x + nif.y*nif.y + nif.z*nif.z);
	int nc_fs = 0, nc_if = 0;
	auto second = [&](uint32_t j, const float4 &npos, float rx, float ry, float rz) {
		if (!is_active_w(npos.w)) return;
		const float r = sqrtf(rx*rx + ry*ry + rz*rz);
		const particleinfo n_info = o.infoInOut[j];
		const float cosconeangle = (PART_TYPE(n_info) == PT_FLUID) ? o.cosconeanglefluid : o.cosconeanglenonfluid;
		if (r < p.influenceradius) {
			const float criteria = -(nfs.x*rx + nfs.y*ry + nfs.z*rz);
			if (criteria > r*lfs*cosconeangle) nc_fs++;
		}
		if (r < p.influenceradius && (fnum == FLUID_NUM(n_info) || PART_TYPE(n_info) != PT_FLUID)) {
			const float criteria = -(nif.x*rx + nif.y*ry + nif.z*rz);
			if (criteria > r*lif*cosconeangle) nc_if++;
		}
	};
	for_each_neib<PT_FLUID>(p, a, index, pos, gridPos, second);
	for_each_neib<PT_BOUNDARY>(p, a, index, pos, gridPos, second);
	if (!nc_fs) info.x |= FG_SURFACE;
	if (!nc_if && nc_fs) info.x |= FG_INTERFACE;
	o.infoInOut[index] = info;
	if (o.normals) {
		nfs.x /= lfs; nfs.y /= lfs; nfs.z /= lfs;
		nif.x /= lif; nif.y /= lif; nif.z /= lif;
		o.normals[index] = (!nc_if && nc_fs) ? nif : nfs;
	}
}

template<int KERNEL>
static void launch_post(int type, dim3 grid, hipStream_t st, const DevParams &p, const FilterArgs &a, const PostArgs &o)
{
	if (type == SPHX_VORTICITY) vorticity_kernel<KERNEL><<<grid, 128, 0, st>>>(p, a, o);
	else if (type == SPHX_TESTPOINTS) testpoints_kernel<KERNEL><<<grid, 128, 0, st>>>(p, a, o);
	else if (type == SPHX_INTERFACE_DETECTION) interface_kernel<KERNEL><<<grid, 128, 0, st>>>(p, a, o);
	else surface_kernel<KERNEL><<<grid, 128, 0, st>>>(p, a, o);
}

extern "C" int sphx_postprocess(sphx_ctx *ctx, int type,
	void *vorticity, void *velInOut, void *infoInOut, void *normals,
	const void *pos, const void *vel, const void *info, const uint32_t *hash,
	const uint32_t *cellStart, const uint16_t *neibsList,
	uint32_t numParticles, uint32_t particleRangeEnd, float cosconeanglefluid, float cosconeanglenonfluid, void *stream)
{
	(void)numParticles;
	SPHX_REQUIRE(ctx && ctx->have_params, "sphx_postprocess: constants not set");
	SPHX_REQUIRE(type == SPHX_VORTICITY || type == SPHX_TESTPOINTS || type == SPHX_SURFACE_DETECTION || type == SPHX_INTERFACE_DETECTION,
		"sphx_postprocess: non-existing postprocess filter invoked");
	SPHX_REQUIRE(pos && hash && cellStart && neibsList, "sphx_postprocess: missing buffer (POS, HASH, CELLSTART, NEIBSLIST)");
	if (ctx->dev.boundarytype != SPHX_DYN_BOUNDARY && ctx->dev.boundarytype != SPHX_LJ_BOUNDARY)
		return sphx_set_error(SPHX_ERR_UNSUPPORTED, "sphx_postprocess: only LJ/DYN boundaries are built");
	if (!particleRangeEnd) return SPHX_OK;
	FilterArgs a;
	a.newVel = nullptr; a.pos = (const float4*)pos; a.vel = (const float4*)vel;
	a.info = (const particleinfo*)info; a.hash = hash; a.cellStart = cellStart; a.neibsList = neibsList;
	a.numParticles = particleRangeEnd;
	PostArgs o;
	o.vorticity = (float*)vorticity; o.velInOut = (float4*)velInOut; o.infoInOut = (particleinfo*)infoInOut;
	o.normals = (float4*)normals; o.cosconeanglefluid = cosconeanglefluid; o.cosconeanglenonfluid = cosconeanglenonfluid;
	if (type == SPHX_VORTICITY) SPHX_REQUIRE(vorticity && vel && info, "sphx_postprocess(VORTICITY): needs VORTICITY, VEL, INFO");
	if (type == SPHX_TESTPOINTS) SPHX_REQUIRE(velInOut && info, "sphx_postprocess(TESTPOINTS): needs VEL (updated in place), INFO");
	if (type == SPHX_SURFACE_DETECTION || type == SPHX_INTERFACE_DETECTION) {
		SPHX_REQUIRE(infoInOut && vel, "sphx_postprocess(SURFACE/INTERFACE_DETECTION): needs INFO (updated in place), VEL");
		a.info = (const particleinfo*)infoInOut;
	}
	const dim3 grid(div_up_u(particleRangeEnd, 128));
	hipStream_t st = (hipStream_t)stream;
	switch (ctx->dev.kerneltype) {
	case SPHX_CUBICSPLINE: launch_post<SPHX_CUBICSPLINE>(type, grid, st, ctx->dev, a, o); break;
	case SPHX_QUADRATIC:   launch_post<SPHX_QUADRATIC>(type, grid, st, ctx->dev, a, o); break;
	case SPHX_WENDLAND:    launch_post<SPHX_WENDLAND>(type, grid, st, ctx->dev, a, o); break;
	case SPHX_GAUSSIAN:    launch_post<SPHX_GAUSSIAN>(type, grid, st, ctx->dev, a, o); break;
	default: return sphx_set_error(SPHX_ERR_INVALID, "sphx_postprocess: invalid kernel type");
	}
	SPHX_LAUNCH_CHECK("postprocess kernel");
	return SPHX_OK;
}

// ==========================================================================================
// Repacking run mode: run_repack (src/cuda/forces.cu:828-896) = repackDevice<fluid,fluid> +
// repackDevice<fluid,boundary> (forces_kernel.def:4155-4262, compute_repacking_contrib :3024-3055) +
// finalizeRepackDevice (:4263-4349) as ONE kernel: the accumulator that the reference carries through the FORCES
// buffer between its launches stays in a register (same additions, same order).  A preparation run of at most
// repack_maxiter iterations, not a roofline path: written for fidelity like the filters above.
// ==========================================================================================
struct RepackArgs {
	float4 *forces, *rbforces, *rbtorques;
	float *cfl;
	const RbParams *rb;
	uint32_t fromParticle, toParticle, cflOffset;
};

template<int KERNEL>
__global__ void __launch_bounds__(SPHX_BLOCK_FORCES)
repack_kernel(DevParams p, FilterArgs a, RepackArgs o)
{
	const uint32_t index = blockIdx.x*SPHX_BLOCK_FORCES + threadIdx.x + o.fromParticle;
	float cfl_term = 0.0f;
	do {
		if (index >= o.toParticle) break;
		const particleinfo info = a.info[index];
		const float4 pos = a.pos[index];
		if (!is_active_w(pos.w)) break;
		const uint32_t fl = FLUID_NUM(info);
		// the caller clobbers FORCES before basicstep (src/GPUWorker.cc:1949)
		float4 force = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
		const int3 gridPos = grid_pos_from_hash(p, a.hash[index] & CELLTYPE_BITMASK);
		if (PART_TYPE(info) == PT_FLUID) {
			auto pair = [&](uint32_t j, const float4 &npos, float rx, float ry, float rz) {
				if (!is_active_w(npos.w)) return;
				const float r = sqrtf(fmaf(rz, rz, fmaf(ry, ry, rx*rx)));
				if (r >= p.influenceradius) return;
				const float n_rho = (a.vel[j].w + 1.0f)*p.rho0[FLUID_NUM(a.info[j])];
				const float f = kernel_F_exact<KERNEL>(p, r);
				const float s = p.repack_a*p.sscoeff[fl]*p.sscoeff[fl]*npos.w/n_rho*f;
				force.x -= s*rx; force.y -= s*ry; force.z -= s*rz;
			};
			for_each_neib<PT_FLUID>(p, a, index, pos, gridPos, pair);
			for_each_neib<PT_BOUNDARY>(p, a, index, pos, gridPos, pair);
		}
		// finalizeRepackDevice
		force.w /= p.rho0[fl];   // repack_fixup :3238-3244
		if (PART_TYPE(info) == PT_FLUID) {
			const float4 vel = a.vel[index];
			const float damp = p.repack_alpha*p.sscoeff[fl]/p.deltap;
			force.x += damp*vel.x; force.y += damp*vel.y; force.z += damp*vel.z;
			// GeometryForce with dynvisc = d_visccoeff*rho (:4314); d_visccoeff is NaN for inviscid problems in the
			// reference (GPUSPH.cc:1488-1493), taken as 0 = free slip here
			const float dynvisc = (p.rheology == SPHX_NEWTONIAN) ? p.visccoeff[fl]*((vel.w + 1.0f)*p.rho0[fl]) : 0.0f;
			if ((p.simflags & SPHX_ENABLE_PLANES) && p.numplanes) {
				for (uint32_t k = 0; k < p.numplanes; ++k) {
					const float dx = (gridPos.x - p.plane_gridpos[k][0])*p.cs[0] + (pos.x - p.plane_pos[k][0]);
					const float dy = (gridPos.y - p.plane_gridpos[k][1])*p.cs[1] + (pos.y - p.plane_pos[k][1]);
					const float dz = (gridPos.z - p.plane_gridpos[k][2])*p.cs[2] + (pos.z - p.plane_pos[k][2]);
					const float r = fabsf(dx*p.plane_normal[k][0] + dy*p.plane_normal[k][1] + dz*p.plane_normal[k][2]);
					if (r < p.r0) {
						const float DvDt = p.dcoeff*(powf(p.r0/r, p.p1coeff) - powf(p.r0/r, p.p2coeff))/(r*r);
						const float qx = p.plane_normal[k][0]*r, qy = p.plane_normal[k][1]*r, qz = p.plane_normal[k][2]*r;
						force.x += DvDt*qx; force.y += DvDt*qy; force.z += DvDt*qz;
						if (dynvisc != 0.0f) {
							const float d = (vel.x*qx + vel.y*qy + vel.z*qz)/r, inv = 1.0f/r;
							const float coeff = -dynvisc*p.partsurf/(pos.w*r);
							force.x += coeff*(vel.x - (d*qx)*inv); force.y += coeff*(vel.y - (d*qy)*inv);
							force.z += coeff*(vel.z - (d*qz)*inv);
						}
					}
				}
			}
			const float sspeed = p.sscoeff[fl]*powf(vel.w + 1.0f, p.sspowercoeff[fl]);
			const float amag = sqrtf(fmaf(force.z, force.z, fmaf(force.y, force.y, force.x*force.x)));
			cfl_term = fmaxf(amag, sspeed*sspeed/p.slength);
		}
		if (HAS_COMPUTE_FORCE(info) && PART_TYPE(info) != PT_VERTEX && o.rbforces) {
			const uint32_t rbindex = (uint32_t)((int)info_id(info) + o.rb->rbstart[OBJECT_NUM(info)]);
			o.rbforces[rbindex] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
			o.rbtorques[rbindex] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
		}
		o.forces[index] = force;
	} while (0);

	if (o.cfl) {   // maxBlockReduce (src/cuda/device_core.cu:40-59)
		__shared__ float wave_max[SPHX_BLOCK_FORCES/64];
#pragma unroll
		for (int d = 32; d > 0; d >>= 1)
			cfl_term = fmaxf(cfl_term, __shfl_down(cfl_term, d));
		if ((threadIdx.x & 63u) == 0) wave_max[threadIdx.x >> 6] = cfl_term;
		__syncthreads();
		if (threadIdx.x == 0) {
			float m = wave_max[0];
			for (int w = 1; w < SPHX_BLOCK_FORCES/64; ++w) m = fmaxf(m, wave_max[w]);
			o.cfl[o.cflOffset + blockIdx.x] = m;
		}
	}
}

// called by sphx_forces_basicstep (forces.hip) when run_mode == SPHX_REPACK, after its argument checks
int sphx_repack_launch(sphx_ctx *ctx, void *forces, float *cfl, void *rbforces, void *rbtorques,
	const void *pos, const void *vel, const void *info, const uint32_t *hash,
	const uint32_t *cellStart, const uint16_t *neibsList,
	uint32_t fromParticle, uint32_t toParticle, uint32_t cflOffset, uint32_t numBlocks, float deltap, hipStream_t st)
{
	if (!(ctx->dev.simflags & SPHX_ENABLE_REPACKING))   // src/main.cc:357-358
		return sphx_set_error(SPHX_ERR_INVALID, "sphx_forces_basicstep: REPACK run mode needs ENABLE_REPACKING in simflags");
	// the particle spacing of the velocity damping term is an argument of basicstep in the reference (run_repack,
	// src/cuda/forces.cu:828-896), not an uploaded constant
	DevParams dev = ctx->dev;
	if (deltap > 0.0f) dev.deltap = deltap;
	SPHX_REQUIRE(dev.deltap > 0.0f, "sphx_forces_basicstep: REPACK run mode needs the particle spacing deltap");
	FilterArgs a;
	a.newVel = nullptr; a.pos = (const float4*)pos; a.vel = (const float4*)vel;
	a.info = (const particleinfo*)info; a.hash = hash; a.cellStart = cellStart; a.neibsList = neibsList;
	a.numParticles = toParticle;
	RepackArgs o;
	o.forces = (float4*)forces; o.rbforces = (float4*)rbforces; o.rbtorques = (float4*)rbtorques;
	o.cfl = (ctx->dev.simflags & SPHX_ENABLE_DTADAPT) ? cfl : nullptr;
	o.rb = ctx->rb_dev; o.fromParticle = fromParticle; o.toParticle = toParticle; o.cflOffset = cflOffset;
	const dim3 grid(numBlocks);
	switch (ctx->dev.kerneltype) {
	case SPHX_CUBICSPLINE: repack_kernel<SPHX_CUBICSPLINE><<<grid, SPHX_BLOCK_FORCES, 0, st>>>(dev, a, o); break;
	case SPHX_QUADRATIC:   repack_kernel<SPHX_QUADRATIC><<<grid, SPHX_BLOCK_FORCES, 0, st>>>(dev, a, o); break;
	case SPHX_WENDLAND:    repack_kernel<SPHX_WENDLAND><<<grid, SPHX_BLOCK_FORCES, 0, st>>>(dev, a, o); break;
	case SPHX_GAUSSIAN:    repack_kernel<SPHX_GAUSSIAN><<<grid, SPHX_BLOCK_FORCES, 0, st>>>(dev, a, o); break;
	default: return sphx_set_error(SPHX_ERR_INVALID, "sphx_forces_basicstep: invalid kernel type");
	}
	SPHX_LAUNCH_CHECK("repack_kernel");
	return SPHX_OK;
}

template<int KERNEL>
static void launch_filter(int filtertype, dim3 grid, hipStream_t st, const DevParams &p, const FilterArgs &a)
{
	if (filtertype == SPHX_SHEPARD_FILTER) shepard_kernel<KERNEL><<<grid, 128, 0, st>>>(p, a);
	else mls_kernel<KERNEL><<<grid, 128, 0, st>>>(p, a);
}

extern "C" int sphx_filter_process(sphx_ctx *ctx, int filtertype, void *newVel,
	const void *pos, const void *oldVel, const void *info, const uint32_t *hash,
	const uint32_t *cellStart, const uint16_t *neibsList,
	uint32_t numParticles, uint32_t particleRangeEnd, float slength, float influenceradius, void *stream)
{
	(void)numParticles;
	SPHX_REQUIRE(ctx && ctx->have_params, "sphx_filter_process: constants not set");
	SPHX_REQUIRE(filtertype == SPHX_SHEPARD_FILTER || filtertype == SPHX_MLS_FILTER, "sphx_filter_process: non-existing filter invoked");
	SPHX_REQUIRE(newVel && pos && oldVel && info && hash && cellStart && neibsList, "sphx_filter_process: missing buffer");
	SPHX_REQUIRE(newVel != oldVel, "sphx_filter_process: the filter reads neighbours' old densities, it cannot run in place");
	SPHX_REQUIRE(slength == ctx->params.slength && influenceradius == ctx->params.influenceradius,
		"sphx_filter_process: slength/influenceradius differ from set_constants");
	if (ctx->dev.boundarytype != SPHX_DYN_BOUNDARY && ctx->dev.boundarytype != SPHX_LJ_BOUNDARY)
		return sphx_set_error(SPHX_ERR_UNSUPPORTED, "sphx_filter_process: only LJ/DYN boundaries are built");
	if (!particleRangeEnd) return SPHX_OK;
	FilterArgs a;
	a.newVel = (float4*)newVel; a.pos = (const float4*)pos; a.vel = (const float4*)oldVel;
	a.info = (const particleinfo*)info; a.hash = hash; a.cellStart = cellStart; a.neibsList = neibsList;
	a.numParticles = particleRangeEnd;
	const dim3 grid(div_up_u(particleRangeEnd, 128));
	hipStream_t st = (hipStream_t)stream;
	switch (ctx->dev.kerneltype) {
	case SPHX_CUBICSPLINE: launch_filter<SPHX_CUBICSPLINE>(filtertype, grid, st, ctx->dev, a); break;
	case SPHX_QUADRATIC:   launch_filter<SPHX_QUADRATIC>(filtertype, grid, st, ctx->dev, a); break;
	case SPHX_WENDLAND:    launch_filter<SPHX_WENDLAND>(filtertype, grid, st, ctx->dev, a); break;
	case SPHX_GAUSSIAN:    launch_filter<SPHX_GAUSSIAN>(filtertype, grid, st, ctx->dev, a); break;
	default: return sphx_set_error(SPHX_ERR_INVALID, "sphx_filter_process: invalid kernel type");
	}
	SPHX_LAUNCH_CHECK("shepard_kernel/mls_kernel");
	return SPHX_OK;
}
