// neib_iter.h -- the neighbour-list walker and the kernel function shared by the engines that are written for fidelity
// (density filters, post-processing, SA boundary conditions): one thread per particle, the reference's u16 list.
#pragma once
#include "sphx_internal.h"

// W<kerneltype>(r, h): src/cuda/sph_core.cu:66-137
template<int KERNEL>
__device__ __forceinline__ float kernel_W(const DevParams &p, float r)
{
	const float R = r/p.slength;
	float val;
	if (KERNEL == SPHX_CUBICSPLINE) {
		if (R < 1) val = 1.0f - 1.5f*R*R + 0.75f*R*R*R;
		else val = 0.25f*(2.0f - R)*(2.0f - R)*(2.0f - R);
	} else if (KERNEL == SPHX_QUADRATIC) {
		val = 0.25f*R*R - R + 1.0f;
	} else if (KERNEL == SPHX_WENDLAND) {
		val = 1.0f - 0.5f*R;
		val *= val;
		val *= val;
		val *= 1.0f + 2.0f*R;
	} else {
		val = expf(-R*R);
		val -= p.wsub_gaussian;
	}
	return val*p.wcoeff;
}

// neiblist_iterator (src/cuda/neibs_iteration.cuh:83-360) over one section of a particle's list:
// f(neib_index, relPos.x, relPos.y, relPos.z) for every stored neighbour, in list order
// A: any struct with members pos (float4*), cellStart, neibsList
// CORR: hand the callback the cell-corrected own position (pos_corr of the reference's iterator) instead of relPos
template<int NPTYPE, bool CORR = false, class A, class F>
__device__ __forceinline__ void for_each_neib(const DevParams &p, const A &a, uint32_t index,
	const float4 &pos, const int3 &gridPos, F &&f)
{
	// FB entries per batch: the list entries of the next batch, the cell bases and the neighbour positions of a batch are
	// independent loads in flight together (the same restructuring took the SPS stress kernel from 8.6 to 2.9 ms at 8 M
	// particles); f is still called once per stored neighbour, in list order, with the same arguments: results unchanged
	constexpr int FB = 4;
	const size_t stride = p.stride;
	// sections: fluid from slot 0 up, boundary from neibboundpos down, vertex (SA_BOUNDARY) from neibboundpos + 1 up
	constexpr bool UP = (NPTYPE != PT_BOUNDARY);
	int slot = (NPTYPE == PT_FLUID) ? 0 : (NPTYPE == PT_BOUNDARY) ? (int)p.neibboundpos : (int)p.neibboundpos + 1;
	uint32_t nd[FB], ndn[FB];
	auto load = [&](int sl0, uint32_t *out) {
#pragma unroll
		for (int k = 0; k < FB; ++k) {
			const int sl = UP ? min(sl0 + k, (int)p.neiblistsize - 1) : max(sl0 - k, 0);
			out[k] = a.neibsList[(size_t)sl*stride + index];
		}
	};
	load(slot, nd);
	int cell = 0;
	uint32_t cell_base = 0;
	bool done = false;
	while (!done) {
		slot = UP ? slot + FB : slot - FB;
		load(slot, ndn);
		bool valid[FB], enc[FB];
		int c[FB];
		uint32_t cb[FB];
		bool alive = true;
#pragma unroll
		for (int k = 0; k < FB; ++k) {
			const uint32_t d = nd[k];
			alive = alive && (d != NEIBS_END);
			valid[k] = alive;
			enc[k] = alive && (d >= CELLNUM_ENCODED);
			c[k] = enc[k] ? (int)(d >> CELLNUM_SHIFT) - 1 : (k ? c[k > 0 ? k - 1 : 0] : cell);
			cb[k] = 0;
			if (enc[k]) {
				const int cz = c[k]/9, cy = (c[k] - cz*9)/3, cx = c[k] - cz*9 - cy*3;
				cb[k] = a.cellStart[grid_hash_periodic(p, gridPos.x + cx - 1, gridPos.y + cy - 1, gridPos.z + cz - 1)];
			}
		}
		done = !alive;
#pragma unroll
		for (int k = 0; k < FB; ++k)
			if (!enc[k]) cb[k] = k ? cb[k > 0 ? k - 1 : 0] : cell_base;
		cell = c[FB - 1];
		cell_base = cb[FB - 1];
		float4 npos[FB];
		uint32_t jj[FB];
#pragma unroll
		for (int k = 0; k < FB; ++k) {
			jj[k] = valid[k] ? cb[k] + (nd[k] & NEIBINDEX_MASK) : index;
			npos[k] = a.pos[jj[k]];
		}
#pragma unroll
		for (int k = 0; k < FB; ++k) {
			if (!valid[k]) continue;
			const int cz = c[k]/9, cy = (c[k] - cz*9)/3, cx = c[k] - cz*9 - cy*3;
			const float pcx = fmaf(-(float)(cx - 1), p.cs[0], pos.x);
			const float pcy = fmaf(-(float)(cy - 1), p.cs[1], pos.y);
			const float pcz = fmaf(-(float)(cz - 1), p.cs[2], pos.z);
			if (CORR) f(jj[k], npos[k], pcx, pcy, pcz);
			else f(jj[k], npos[k], pcx - npos[k].x, pcy - npos[k].y, pcz - npos[k].z);
		}
#pragma unroll
		for (int k = 0; k < FB; ++k) nd[k] = ndn[k];
	}
}


// ---- small pieces of physics shared by the fidelity engines (sa_bounds.hip, rheology.hip): exact powf / division forms ----
__device__ __forceinline__ float sa_dot3(float ax, float ay, float az, float bx, float by, float bz)
{ return fmaf(az, bz, fmaf(ay, by, ax*bx)); }
__device__ __forceinline__ float sa_P(const DevParams &p, float rho_tilde, uint32_t fl)
{ return p.bcoeff[fl]*(powf(rho_tilde + 1.0f, p.gammacoeff[fl]) - 1.0f); }
__device__ __forceinline__ float sa_sound_speed(const DevParams &p, float rho_tilde, uint32_t fl)
{ return p.sscoeff[fl]*powf(rho_tilde + 1.0f, p.sspowercoeff[fl]); }

// visc_avg<ViscSpec> (src/cuda/visc_avg.cu:40-190), neighbour mass included
__device__ __forceinline__ float sa_visc_avg_rho(int avgop, float rho, float neib_rho, float neib_mass)
{
	if (avgop == SPHX_ARITHMETIC) return neib_mass*(rho + neib_rho)/(rho*neib_rho);
	if (avgop == SPHX_HARMONIC) return 4*neib_mass/(rho + neib_rho);
	return 2*neib_mass*(1.0f/sqrtf(rho*neib_rho));
}
__device__ __forceinline__ float sa_visc_avg_dyn(int avgop, bool is_const, float visc, float neib_visc, float rho, float neib_rho, float neib_mass)
{
	if (is_const) return 2*neib_mass*visc/(rho*neib_rho);
	if (avgop == SPHX_ARITHMETIC) return neib_mass*(visc + neib_visc)/(rho*neib_rho);
	if (avgop == SPHX_HARMONIC) return 4*neib_mass*(visc*neib_visc)/(visc + neib_visc)/(rho*neib_rho);
	return 2*neib_mass*sqrtf(visc*neib_visc)/(rho*neib_rho);
}
__device__ __forceinline__ float sa_visc_avg(const DevParams &p, float visc, float neib_visc, float rho, float neib_rho, float neib_mass)
{
	if (p.compvisc == SPHX_DYNAMIC) return sa_visc_avg_dyn(p.avgop, p.is_const_visc, visc, neib_visc, rho, neib_rho, neib_mass);
	if (p.is_const_visc) return visc*sa_visc_avg_rho(p.avgop, rho, neib_rho, neib_mass);
	// non-constant kinematic: the dynamic variant, whose constness is re-derived as IS_SINGLEFLUID && NEWTONIAN
	// (src/cuda/visc_avg.cu:170-190, src/visc_spec.h:268-272)
	return sa_visc_avg_dyn(p.avgop, !(p.simflags & SPHX_ENABLE_MULTIFLUID) && p.rheology == SPHX_NEWTONIAN && p.turbmodel != SPHX_KEPSILON,
		visc*rho, neib_visc*neib_rho, rho, neib_rho, neib_mass);
}

