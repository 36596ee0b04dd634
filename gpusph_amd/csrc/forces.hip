// forces.hip -- forces engine for gfx950: fused pair summation (fluid<-fluid, fluid<-boundary,
// boundary<-fluid) + finalize + CFL block maxima in ONE launch, CFL reduction -> dt on the
// device, rigid-body force totals, SPS stress tensor.
// Replaces CUDAForcesEngine / CUDAViscEngine (GPUSPH src/cuda/forces.cu:252-1006,
// src/cuda/forces_kernel.def:3914-4150, src/cuda/visc_kernel.cu:759-811).
//
// Why one launch: the reference launches forcesDevice 3x + finalizeforcesDevice, each doing a
// read-modify-write of forces[] (16 B x 2 x 4 per particle) and re-reading pos/vel/info/hash.
// The neighbour list is typed (fluid slots grow up from 0, boundary slots down from
// neibboundpos), so a single thread can walk both sections in the reference's order and keep the
// accumulator in registers: forces[] is written once, own data is read once.
//
// Numerics: accumulation order per particle is the reference's (list order, fluid section then
// boundary section), so results differ from oracle/sph_oracle.c only through the fast
// transcendental path (v_log_f32/v_exp_f32 for the Tait EOS instead of powf, v_rcp_f32,
// v_sqrt_f32), the same class of deviation the reference's own __powf has.
#include "sphx_internal.h"

struct ForcesArgs {
	float4 *forces;
	float  *cfl;
	float4 *rbforces;
	float4 *rbtorques;
	const float4 *pos;
	const float4 *vel;
	const particleinfo *info;
	const uint32_t *hash;
	const uint32_t *cellStart;
	const neibdata *neibsList;
	const float2 *tau0, *tau1, *tau2;
	const float4 *aux;   // per-particle EOS pre-pass {P/rho^2, c, P, rho}
	float2 *otau0, *otau1, *otau2; float *oturbvisc;   // stress mode of the tiled kernel (SPHX_TURB_STRESS): outputs
	const float4 *tauPack;   // SPS + tiled kernel: [0,n) = {xx,xy,xz,yy}, [n,2n) = {yz,zz,P/rho^2,c or P} (tau_pack_kernel); tauPackN = n
	uint32_t tauPackN;
	// tiled kernel: the tile lists of this neighbour list (tile_lists_kernel)
	const uint2 *tileList;       // the stream of list batches: [batch][64 lanes] x 4 uint16 window offsets
	const uint32_t *tileRuns;    // [tile][TILE_RUNTAB]: which wave walks which batches of which chunk (sphx_internal.h)
	const uint32_t *tileRows;    // [tile][TILE_ROWDESC]: the window rows
	const uint32_t *tileLaneRec; // per lane of every chunk: byte offset of the particle's own row in its tile's window | flags << 16
	const uint32_t *tileLaneIndex; // ... the particle (0xFFFFFFFF: idle lane)
	float4 *xsph;        // ENABLE_XSPH: mean velocity correction of fluid particles, else NULL
	const float4 *saGam; // SA_BOUNDARY modes of the tiled kernel (SPHX_TURB_SA*): gamma in .w (diffusion mode), the step's dt
	float saDt;
	const RbParams *rb;
	uint32_t fromParticle, toParticle, cflOffset;
	int wholeRange;       // tiled kernel: [fromParticle, toParticle) holds every tiled particle (no per-tile / per-particle range test)
	uint32_t numBlocks;   // generic kernel: blocks of SPHX_BLOCK_FORCES particles to cover
	int compute_object_forces;
	uint32_t *pin;              // always NULL (see pin_batch)
#ifdef SPHX_TILE_DEBUG_BUILD
	unsigned long long *prof;   // per wave of every workgroup: cycles per phase of the tile loop (SPHX_TILE_DEBUG=16), else NULL
#endif
};

__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// F<kerneltype>(r, h): src/cuda/sph_core.cu:146-191
template<int KERNEL>
__device__ __forceinline__ float kernel_F(const DevParams &p, float r, float inv_h)
{
	if (KERNEL == SPHX_WENDLAND) {
		const float qm2 = fmaf(r, inv_h, -2.0f);
		return qm2*qm2*qm2*p.fcoeff;
	} else if (KERNEL == SPHX_CUBICSPLINE) {
		const float R = r*inv_h;
		const float val = (R < 1.0f) ? (-4.0f + 3.0f*R)*inv_h : -(-2.0f + R)*(-2.0f + R)*fast_rcp(r);
		return val*p.fcoeff;
	} else if (KERNEL == SPHX_QUADRATIC) {
		const float R = r*inv_h;
		return (-2.0f + R)*fast_rcp(r)*p.fcoeff;
	} else {
		const float R = r*inv_h;
		return -fast_exp2(-R*R*1.44269504088896340736f)*p.fcoeff;
	}
}

// ------------------------------------------------------------------------------------------
// per-particle EOS pre-pass: aux[i] = {P/rho^2, c, P, rho}.
// The reference recomputes P (__powf), soundSpeed (__powf) and P/rho^2 (division) of the
// NEIGHBOUR inside every pair (forces_kernel.def:690-702,1126-1128; phys_core.cu:99-135).  They are
// pure functions of the neighbour's own rho~, so evaluating them once per particle and gathering
// 16 B gives the same numbers with ~20 fewer issue slots per pair -- and, being per particle, can
// afford the accurate powf and IEEE division (closer to the oracle than __powf would be).
// ------------------------------------------------------------------------------------------
static __global__ void __launch_bounds__(256)
eos_kernel(DevParams p, const float4 *__restrict__ vel, const particleinfo *__restrict__ info,
	float4 *__restrict__ aux, uint32_t n)
{
	const uint32_t i = blockIdx.x*256 + threadIdx.x;
	if (i >= n) return;
	const uint32_t fl = (p.numfluids > 1) ? FLUID_NUM(info[i]) : 0u;
	const float ratio = vel[i].w + 1.0f;
	const float P = p.bcoeff[fl]*(powf(ratio, p.gammacoeff[fl]) - 1.0f);
	float c = p.sscoeff[fl]*powf(ratio, p.sspowercoeff[fl]);
	// multi-fluid runs: the fluid number replaces the two lowest mantissa bits of the sound speed (2^-22 relative; c only
	// enters the artificial viscosity, Ferrari and CFL terms), so that the LDS window of the tiled kernel, which holds
	// pos / vel / this row but not the particle info, can tell same-fluid pairs and per-fluid viscosities.  Both forces
	// kernels read the same tagged value; single-fluid runs are untouched.
	if (p.numfluids > 1) c = __uint_as_float((__float_as_uint(c) & ~3u) | fl);
	const float rho = ratio*p.rho0[fl];
	aux[i] = make_float4(P/(rho*rho), c, P, rho);
}

// SPS + tiled kernel: the three float2 arrays of BUFFER_TAU repacked as two float4 rows per particle, so that the window
// rows can be staged with the same 16-byte LDS DMA as pos / vel / the EOS row.  The two free slots of the second row carry what
// a pair needs of the neighbour's EOS row besides the density (which is one add and one multiply away from the velocity row's
// rho~): P/rho^2 and the one of {sound speed, pressure} the run's density diffusion reads (Ferrari / none: c; Colagrossi: P).
// With one fluid the window of an SPS run then needs no EOS rows at all: 64 instead of 80 bytes per record, tiles of six
// instead of four cells (TILE_WCAP_SPS1), 82 % instead of 55 % of the lanes of a tile occupied.
static __global__ void __launch_bounds__(256)
tau_pack_kernel(const float2 *__restrict__ t0, const float2 *__restrict__ t1, const float2 *__restrict__ t2,
	const float4 *__restrict__ aux, int wantPressure, float4 *__restrict__ out, uint32_t n)
{
	const uint32_t i = blockIdx.x*256 + threadIdx.x;
	if (i >= n) return;
	const float2 a = t0[i], b = t1[i], c = t2[i];
	const float4 e = aux[i];
	out[i] = make_float4(a.x, a.y, b.x, b.y);
	out[(size_t)n + i] = make_float4(c.x, c.y, e.x, wantPressure ? e.z : e.y);
}

struct Self {
	float4 pos, vel;
	int3 gridPos;
	float p_precalc, sspeed, P, rho, inv_rho;
	uint32_t fl;
	float tau[6];
	// NEWTONIAN rheology: own viscosity terms and the averaging selectors, per lane (see laminar_factor)
	float visc_c, visc_mu, visc_kin, visc_onemk, visc_wA, visc_wH, visc_wG;
	uint32_t visc_constmask, visc_ownmask;
	uint32_t f2mask;   // all ones for SPH_F2 (generic kernel only), per lane like the viscosity selectors
	float sa_dt2rho;   // SA density diffusion mode of the tiled kernel: dt * 2 * rho_i
};

// one pair (i <- j).  Terms, in the reference's order (compute_all_pp_interaction,
// forces_kernel.def:3565-3610): continuity + density diffusion -> force.w; pressure + viscous ->
// force.xyz.  (pcx,pcy,pcz) = own position shifted into the neighbour's cell frame.
// density-diffusion template codes (the parameter keeps its historical name COLAGROSSI): 1 keeps `true` = Colagrossi
#define DIFF_NONE 0
#define DIFF_COLAGROSSI 1
#define DIFF_FERRARI 2

// TURB template codes: the turbulence model in the low bits, SPHX_TURB_NEWT set for the NEWTONIAN rheology
#define SPHX_TURB_NEWT 8
#define SPHX_TURB_MF 16     // tiled kernel only: more than one fluid, the neighbour's fluid number rides in the EOS row (eos_kernel)
#define SPHX_TURB_STRESS 32 // tiled kernel only: not a forces pass but the SPS stress tensor (SPSstressMatrixDevice) over the same tiles
// SA_BOUNDARY over the same tiles (sa_bounds.hip finishes each of them with the boundary-element terms, by the list walker):
// the particle <- particle sums of the forces (fluid and vertex neighbours; the vertex section takes the place of the boundary
// section in the tile lists), of the density summation and of the Brezzi density diffusion
#define SPHX_TURB_SA 64
#define SPHX_TURB_SA_DSUM 128
#define SPHX_TURB_SA_DIFF 256
#define SPHX_TURB_SA_ANY (SPHX_TURB_SA | SPHX_TURB_SA_DSUM | SPHX_TURB_SA_DIFF)
#define TURB_MODEL(T) ((T) & 7)

// select chain on the (at most four) kernel-argument values instead of a lane-indexed load from the argument block
__device__ __forceinline__ float visc_of(const DevParams &p, uint32_t fl)
{
	const float lo = (fl & 1u) ? p.visccoeff[1] : p.visccoeff[0], hi = (fl & 1u) ? p.visccoeff[3] : p.visccoeff[2];
	return (fl & 2u) ? hi : lo;
}

// visc_avg (src/cuda/visc_avg.cu:40-190) without the neighbour mass: the per-pair factor of the laminar Morris term,
// nu-or-mu averaged over the pair / densities.  Every flavour of the reference is one instance of
//   mu_i = c_i rho_i^k , mu_j = c'_j rho_j^k  (k = 1 kinematic, 0 dynamic; c'_j = c_i when the viscosity is declared
//   constant, the neighbour's fluid coefficient otherwise),
//   arithmetic (mu_i + mu_j)/(rho_i rho_j), harmonic 4 mu_i mu_j/((mu_i + mu_j) rho_i rho_j), geometric 2 sqrt(mu_i mu_j)/(rho_i rho_j)
// (the constant-viscosity shortcuts of visc_avg.cu are these with c'_j = c_i, up to rounding).  It is evaluated
// branch-free with per-lane selector values prepared once per particle (init_visc) instead of wave-uniform branches on
// the kernel arguments: with the branches, the generic kernel took the constant-viscosity path from the second list
// batch of the boundary section on (the uniform condition did not survive that loop), which the two-fluid
// non-constant-viscosity parity test exposed; the select form is also what keeps the tiled and generic kernels bit-equal.
__device__ __forceinline__ void init_visc(const DevParams &p, Self &s)
{
	float c = visc_of(p, s.fl);
	uint32_t cm = p.is_const_visc ? 0xFFFFFFFFu : 0u;
	float kin = (p.compvisc == SPHX_KINEMATIC) ? 1.0f : 0.0f;
	float wA = (p.avgop == SPHX_ARITHMETIC) ? 1.0f : 0.0f, wH = (p.avgop == SPHX_HARMONIC) ? 4.0f : 0.0f,
		wG = (p.avgop == SPHX_GEOMETRIC) ? 2.0f : 0.0f;
	// a single-fluid framework forced to non-constant KINEMATIC viscosity: the reference's with_computational_visc<DYNAMIC>
	// re-derives is_const_visc = true (src/visc_spec.h:268-272,298-300, src/cuda/visc_avg.cu:180-190) and evaluates the
	// constant dynamic formula 2 mu_i/(rho_i rho_j): mu_j := mu_i with the arithmetic weights (pinned by ref_viscavg.npz)
	const bool own = !p.is_const_visc && p.compvisc == SPHX_KINEMATIC && !(p.simflags & SPHX_ENABLE_MULTIFLUID) && p.rheology == SPHX_NEWTONIAN;
	uint32_t om = own ? 0xFFFFFFFFu : 0u;
	if (own) { wA = 1.0f; wH = 0.0f; wG = 0.0f; }
	asm volatile("" : "+v"(c), "+v"(cm), "+v"(kin), "+v"(wA), "+v"(wH), "+v"(wG), "+v"(om));   // keep them per-lane values
	s.visc_c = c; s.visc_constmask = cm; s.visc_ownmask = om; s.visc_kin = kin; s.visc_onemk = 1.0f - kin;
	s.visc_wA = wA; s.visc_wH = wH; s.visc_wG = wG;
	s.visc_mu = c*fmaf(kin, s.rho, s.visc_onemk);      // c rho or c, exactly
}

__device__ __forceinline__ float laminar_factor(const DevParams &p, const Self &s, uint32_t nfl, float n_rho)
{
	const uint32_t cb = __float_as_uint(s.visc_c), nb = __float_as_uint(visc_of(p, nfl));
	const float nc = __uint_as_float((cb & s.visc_constmask) | (nb & ~s.visc_constmask));
	const float nmu0 = nc*fmaf(s.visc_kin, n_rho, s.visc_onemk);
	const float nmu = __uint_as_float((__float_as_uint(s.visc_mu) & s.visc_ownmask) | (__float_as_uint(nmu0) & ~s.visc_ownmask));
	const float S = s.visc_mu + nmu, P = s.visc_mu*nmu;
	const float num = fmaf(s.visc_wA, S, fmaf(s.visc_wH*P, fast_rcp(fmaxf(S, 1.0e-30f)), s.visc_wG*fast_sqrt(P)));
	return num*fast_rcp(s.rho*n_rho);
}

template<int KERNEL, int TURB, int COLAGROSSI, bool MOMENTUM, bool DIFFUSE>
__device__ __forceinline__ void pair_interact(const DevParams &p, const Self &s, float inv_h,
	float pcx, float pcy, float pcz, const float4 &npos, const float4 &nvel, const float4 &naux,
	bool same_fluid, bool valid, const float *ntau, float4 &force, bool rt_momentum = true, bool rt_diffuse = true,
	uint32_t nfl = 0, bool f2cap = false, float range = -1.0f)
{
	// range: the influence radius as the caller holds it (the tiled kernel reads it from the shift-table row so that the
	// row is read whole); < 0 = the kernel argument
	// Branch-free on purpose: a rejected pair (list terminator passed, r >= influence radius) gets the weight
	// m_j F_ij = 0 and every term below becomes +-0, which leaves the accumulators untouched -- the same result
	// as the reference's `continue`, without ~6 exec-mask branch sequences per pair and without paying for lane
	// divergence inside a wave.  (Inactive particles sit outside every cell after the sort, so they are never
	// listed and need no test here.)
	// Multiply-adds are written as explicit fmaf where nvcc's default -fmad=true contracts them in the
	// reference as well: the pair loop is VALU-bound and these are ~10 % of its instructions.
	const float rx = pcx - npos.x, ry = pcy - npos.y, rz = pcz - npos.z;
	const float nmass = npos.w;
	const float r2 = fmaf(rz, rz, fmaf(ry, ry, rx*rx));
	const float r = fast_sqrt(r2);
	const bool on = valid && (r < (range < 0.0f ? p.influenceradius : range));

	const float vx = s.vel.x - nvel.x, vy = s.vel.y - nvel.y, vz = s.vel.z - nvel.z;
	const float vel_dot_pos = fmaf(vz, rz, fmaf(vy, ry, vx*rx));
	const float f = kernel_F<KERNEL>(p, r, inv_h);
	const float n_precalc = naux.x, n_sspeed = naux.y, n_P = naux.z, n_rho = naux.w;
	const float mf = on ? nmass*f : 0.0f;

	// mass_continuity_div_vel_term (forces_kernel.def:2140-2151)
	float dsel = 0.0f;
	if (COLAGROSSI == DIFF_FERRARI && DIFFUSE) { // compute_density_diffusion, Ferrari (forces_kernel.def:1607-1635)
		const float gdotr = fmaf(p.gravity[2], rz, fmaf(p.gravity[1], ry, p.gravity[0]*rx));
		const float c0 = p.sscoeff[s.fl];
		const float grav_corr = -gdotr*p.rho0[s.fl]*fast_rcp(c0*c0);
		const float sc = fmaxf(s.sspeed, n_sspeed)*(s.rho - n_rho + grav_corr)*s.inv_rho*fast_rcp(r);
		const float fterm = p.densityDiffCoeff*mf*(sc*r2);
		dsel = (rt_diffuse && r > 1e-4f*p.slength) ? -fterm : 0.0f;
	}
	if (COLAGROSSI == DIFF_COLAGROSSI && DIFFUSE) { // compute_density_diffusion (forces_kernel.def:1916-1952)
		const float gdotr = fmaf(p.gravity[2], rz, fmaf(p.gravity[1], ry, p.gravity[0]*rx));
		const bool diff = same_fluid && rt_diffuse && !(fabsf(s.P - n_P) < fabsf(gdotr*s.rho));
		const float dterm = p.densityDiffCoeff*p.sscoeff[s.fl]*fmaf(n_rho, s.inv_rho, -1.0f)*mf;
		dsel = diff ? dterm : 0.0f;
	}
	float drdt = fmaf(mf, vel_dot_pos, -dsel);
	if (f2cap) {   // SPH_F2: density ratio applied after the diffusion term (mass_continuity_density_ratio :2154-2165)
		const float ratio = s.rho*fast_rcp(n_rho);
		drdt *= __uint_as_float((__float_as_uint(ratio) & s.f2mask) | (0x3f800000u & ~s.f2mask));
	}
	force.w += drdt;

	if (MOMENTUM) {
		// compute_pressure_contrib (forces_kernel.def:2451-2466): -(P_i/rho_i^2 + P_j/rho_j^2) m_j F r_ij
		float pgrad = s.p_precalc + n_precalc;
		if (f2cap) {   // SPH_F2: (P_i + P_j)/(rho_i rho_j) (pressure_gradient_term :2253-2266)
			const float pg2 = (s.P + n_P)*fast_rcp(s.rho*n_rho);
			pgrad = __uint_as_float((__float_as_uint(pg2) & s.f2mask) | (__float_as_uint(pgrad) & ~s.f2mask));
		}
		float kk = -pgrad*mf;
		if (TURB_MODEL(TURB) == SPHX_ARTIFICIAL) {
			// artvisc (src/cuda/visc_kernel.cu:74-85, forces_kernel.def:2748-2764): only for approaching pairs
			// |rho~_j| >= 0 never wins the minimum: it only keeps the fourth component of the neighbour's velocity row live,
			// so that the tiled kernel reads the row as one 16-byte LDS access (v_min3_f32 with an abs modifier: no extra
			// instruction; a 12-byte read costs twice the LDS cycles)
			const float vdpn = fminf(fminf(vel_dot_pos, 0.0f), fabsf(nvel.w));
			const float visc = vdpn*(p.slength*p.artvisccoeff)*(s.sspeed + n_sspeed)*
				fast_rcp((r2 + p.epsartvisc)*(s.rho + n_rho));
			kk = fmaf(visc, mf, kk);
		}
		kk = rt_momentum ? kk : 0.0f;
		if (TURB_MODEL(TURB) == SPHX_SPS) { // forces_kernel.def:2777-2798
			const float mg = rt_momentum ? mf : 0.0f;
			const float xx = s.tau[0] + ntau[0], xy = s.tau[1] + ntau[1], xz = s.tau[2] + ntau[2];
			const float yy = s.tau[3] + ntau[3], yz = s.tau[4] + ntau[4], zz = s.tau[5] + ntau[5];
			force.x = fmaf(mg, fmaf(xz, rz, fmaf(xy, ry, xx*rx)), force.x);
			force.y = fmaf(mg, fmaf(yz, rz, fmaf(yy, ry, xy*rx)), force.y);
			force.z = fmaf(mg, fmaf(zz, rz, fmaf(yz, ry, xz*rx)), force.z);
		}
		force.x = fmaf(kk, rx, force.x); force.y = fmaf(kk, ry, force.y); force.z = fmaf(kk, rz, force.z);
		if (TURB & SPHX_TURB_NEWT) {
			// compute_laminar_visc_contrib, MORRIS (forces_kernel.def:2606-2625): visc_avg * F * (v_i - v_j), after the
			// turbulent term (compute_viscous_contrib :2881-2886)
			const float lv = rt_momentum ? laminar_factor(p, s, nfl, n_rho)*mf : 0.0f;
			force.x = fmaf(lv, vx, force.x); force.y = fmaf(lv, vy, force.y); force.z = fmaf(lv, vz, force.z);
		}
	}
}

// Lennard-Jones repulsion of a boundary particle (compute_repulsive_force forces_kernel.def:3001-3016, LJForce
// src/cuda/forces_kernel.cu:94-103): the whole fluid<-boundary (and, for bodies with force feedback,
// boundary<-fluid) interaction of LJ_BOUNDARY.  Few pairs, generic kernel only: accurate powf and IEEE division.
__device__ __forceinline__ void lj_interact(const DevParams &p, float pcx, float pcy, float pcz,
	const float4 &npos, bool valid, float4 &force)
{
	const float rx = pcx - npos.x, ry = pcy - npos.y, rz = pcz - npos.z;
	const float r = sqrtf(fmaf(rz, rz, fmaf(ry, ry, rx*rx)));
	const bool in = valid && is_active_w(npos.w) && r < p.influenceradius;
	float ljf = 0.0f;
	if (in && r <= p.r0)
		ljf = p.dcoeff*(powf(p.r0/r, p.p1coeff) - powf(p.r0/r, p.p2coeff))/(r*r);
	// Monaghan-Kajtar law (MKForce src/cuda/forces_kernel.cu:105-133, both masses = the central particle's, so they
	// cancel): K w(q)/(beta max(eps, r - d) r), w = 1.8 (1 - q/2)^4 (2q + 1); chosen per lane by a mask
	const float qq = r/p.slength, om = 1.0f - 0.5f*qq;
	const float w = 1.8f*((om*om)*(om*om))*(2.0f*qq + 1.0f);
	const float mkf = (in && r <= 2.0f*p.slength) ? p.MK_K*w/(p.MK_beta*fmaxf(p.epsartvisc, r - p.MK_d)*r) : 0.0f;
	uint32_t mk = p.mk_mask;
	asm volatile("" : "+v"(mk));
	ljf = __uint_as_float((__float_as_uint(mkf) & mk) | (__float_as_uint(ljf) & ~mk));
	force.x += ljf*rx; force.y += ljf*ry; force.z += ljf*rz;
}

// finalizeforcesDevice (forces_kernel.def:4032-4150) for one particle; returns its CFL term
__device__ __forceinline__ float finalize_particle(const DevParams &p, const ForcesArgs &a, uint32_t index,
	const particleinfo &info, const Self &s, float4 force)
{
	float cfl_term = 0.0f;
	const uint32_t ptype = PART_TYPE(info);
	force.w /= p.rho0[s.fl]; // forces_fixup :3212-3218
	if (ptype == PT_FLUID) {
		force.x += p.gravity[0]; force.y += p.gravity[1]; force.z += p.gravity[2];
		// GeometryForce / PlaneForce (src/cuda/forces_kernel.cu:140-203): Lennard-Jones repulsion along the plane
		// normal; the friction term vanishes for the inviscid rheology built here (viscous_plane_coefficient :3103-3107)
		// PlaneForce (:140-185) of a plane given by its unit normal and the grid + local position of a point on it
		auto plane_force = [&](const float nrm[3], const int pgp[3], const float ppos[3]) {
			const float dx = (s.gridPos.x - pgp[0])*p.cs[0] + (s.pos.x - ppos[0]);
			const float dy = (s.gridPos.y - pgp[1])*p.cs[1] + (s.pos.y - ppos[1]);
			const float dz = (s.gridPos.z - pgp[2])*p.cs[2] + (s.pos.z - ppos[2]);
			const float r = fabsf(dx*nrm[0] + dy*nrm[1] + dz*nrm[2]);
			if (r < p.r0) {
				const float DvDt = p.dcoeff*(powf(p.r0/r, p.p1coeff) - powf(p.r0/r, p.p2coeff))/(r*r);
				const float qx = nrm[0]*r, qy = nrm[1]*r, qz = nrm[2]*r;
				force.x += DvDt*qx; force.y += DvDt*qy; force.z += DvDt*qz;
				if (p.rheology == SPHX_NEWTONIAN) {
					// wall friction of PlaneForce (:153-185): -mu partsurf/(m r) v_t, mu = get_laminar_dyn_visc
					const float dynvisc = (p.compvisc == SPHX_KINEMATIC) ? p.visccoeff[s.fl]*s.rho : p.visccoeff[s.fl];
					const float d = (s.vel.x*qx + s.vel.y*qy + s.vel.z*qz)/r, inv = 1.0f/r;
					const float coeff = -dynvisc*p.partsurf/(s.pos.w*r);
					force.x += coeff*(s.vel.x - (d*qx)*inv); force.y += coeff*(s.vel.y - (d*qy)*inv);
					force.z += coeff*(s.vel.z - (d*qz)*inv);
				}
			}
		};
		// DemLJForce (src/cuda/forces_kernel.cu:205-226), the LJ_BOUNDARY case of the finalize kernel (forces_kernel.def:4093-4102):
		// a particle less than demzmin above the terrain is repelled by the terrain's tangent plane below it
		if ((p.simflags & SPHX_ENABLE_DEM) && p.dem && p.boundarytype == SPHX_LJ_BOUNDARY && !p.mk_mask) {
			float nrm[3], ppos[3]; int pgp[3];
			if (dem_plane(p, s.gridPos, s.pos.x, s.pos.y, s.pos.z, nrm, pgp, ppos)) plane_force(nrm, pgp, ppos);
		}
		if ((p.simflags & SPHX_ENABLE_PLANES) && p.numplanes) {
			for (uint32_t k = 0; k < p.numplanes; ++k) plane_force(p.plane_normal[k], p.plane_gridpos[k], p.plane_pos[k]);
		}
		// dyndt_forces_shared_data::store (:3436-3457)
		const float amag = sqrtf(fmaf(force.z, force.z, fmaf(force.y, force.y, force.x*force.x)));
		cfl_term = fmaxf(amag, s.sspeed*s.sspeed/p.slength);
	}
	if (HAS_COMPUTE_FORCE(info) && ptype != PT_VERTEX && a.rbforces) { // :4121-4142
		force.x *= s.pos.w; force.y *= s.pos.w; force.z *= s.pos.w;
		const uint32_t obj = OBJECT_NUM(info);
		const uint32_t rbindex = (uint32_t)((int)info_id(info) + a.rb->rbstart[obj]);
		a.rbforces[rbindex] = force;
		const float armx = (s.gridPos.x - a.rb->cgGridPos[obj][0])*p.cs[0] + (s.pos.x - a.rb->cgPos[obj][0]);
		const float army = (s.gridPos.y - a.rb->cgGridPos[obj][1])*p.cs[1] + (s.pos.y - a.rb->cgPos[obj][1]);
		const float armz = (s.gridPos.z - a.rb->cgGridPos[obj][2])*p.cs[2] + (s.pos.z - a.rb->cgPos[obj][2]);
		a.rbtorques[rbindex] = make_float4(army*force.z - armz*force.y,
			armz*force.x - armx*force.z, armx*force.y - army*force.x, 0.0f);
	}
	a.forces[index] = force;
	return cfl_term;
}

template<int TURB>
__device__ __forceinline__ void load_self(const DevParams &p, const ForcesArgs &a, uint32_t index,
	const particleinfo &info, const float4 &pos, bool multifluid, Self &s)
{
	s.pos = pos;
	s.vel = a.vel[index];
	s.gridPos = grid_pos_from_hash(p, a.hash[index] & CELLTYPE_BITMASK);
	s.fl = multifluid ? FLUID_NUM(info) : 0u;
	const float4 ax = a.aux[index];
	s.p_precalc = ax.x; s.sspeed = ax.y; s.P = ax.z; s.rho = ax.w;
	s.inv_rho = fast_rcp(ax.w);
	if (TURB & SPHX_TURB_NEWT) init_visc(p, s);
	uint32_t f2 = (p.formulation == SPHX_SPH_F2) ? 0xFFFFFFFFu : 0u;
	asm volatile("" : "+v"(f2));
	s.f2mask = f2;
	if (TURB_MODEL(TURB) == SPHX_SPS) {
		const float2 t0 = a.tau0[index], t1 = a.tau1[index], t2 = a.tau2[index];
		s.tau[0] = t0.x; s.tau[1] = t0.y; s.tau[2] = t1.x; s.tau[3] = t1.y; s.tau[4] = t2.x; s.tau[5] = t2.y;
	}
}

// run-time section: sec 0 = fluid neighbours (slots 0 upward), sec 1 = boundary neighbours
// (slots neibboundpos downward); `batch` counts TILE_NB-entry batches from the section start
__device__ __forceinline__ void load_list_rt(const DevParams &p, const neibdata *__restrict__ list,
	uint32_t index, int sec, int batch, uint32_t nd[TILE_NB])
{
#pragma unroll
	for (int k = 0; k < TILE_NB; ++k) {
		const int e = batch*TILE_NB + k;
		const int sl = sec ? max((int)p.neibboundpos - e, 0) : min(e, (int)p.neiblistsize - 1);
		nd[k] = list[(size_t)sl*p.stride + index];
	}
}

template<int NPTYPE, int N>
__device__ __forceinline__ void load_list_batch(const DevParams &p, const neibdata *__restrict__ list,
	uint32_t index, int slot, uint32_t nd[N])
{
#pragma unroll
	for (int k = 0; k < N; ++k) {
		// entries past the terminator are never used; the clamp only keeps the address in bounds
		const int sl = (NPTYPE == PT_FLUID) ? min(slot + k, (int)p.neiblistsize - 1) : max(slot - k, 0);
		nd[k] = list[(size_t)sl*p.stride + index];
	}
}

// ==========================================================================================
// Generic path: neighbour rows gathered through L2.  Used for multi-fluid and SPS runs, for
// neighbour lists not built by this context, and whenever the tiling below overflowed.
// ==========================================================================================
#define NB 4   // neighbours resolved per batch: list entries, cell bases and particle rows of a
               // batch are fetched with independent loads in flight (3 dependent round trips per NB
               // neighbours instead of per neighbour); the next batch's list entries are prefetched

// walk one typed section of the neighbour list (neiblist_iterator_simple,
// src/cuda/neibs_iteration.cuh:165-205; getNeibIndex src/cuda/cellgrid.cuh:200-228)
// LJW: the section is a Lennard-Jones repulsion walk (positions only): a template flag, so that the pair loop holds one
// interaction body
template<int KERNEL, int TURB, int COLAGROSSI, bool MULTIFLUID, int NPTYPE, bool MOMENTUM, bool DIFFUSE, bool LJW = false>
__device__ __forceinline__ void walk_section(const DevParams &p, const ForcesArgs &a, uint32_t index,
	const Self &s, float inv_h, float4 &force)
{
	int slot = (NPTYPE == PT_FLUID) ? 0 : (int)p.neibboundpos;
	uint32_t nd[NB], ndn[NB];
	load_list_batch<NPTYPE, NB>(p, a.neibsList, index, slot, nd);

	int cell = 0;
	uint32_t cell_base = 0;
	bool done = false;

	while (!done) {
		slot = (NPTYPE == PT_FLUID) ? slot + NB : slot - NB;
		load_list_batch<NPTYPE, NB>(p, a.neibsList, index, slot, ndn);

		// stage 1: decode entries, fetch the base index of every cell that changes in this batch
		bool valid[NB], enc[NB];
		int c[NB];
		uint32_t cb[NB];
		bool alive = true;
#pragma unroll
		for (int k = 0; k < NB; ++k) {
			const uint32_t d = nd[k];
			alive = alive && (d != NEIBS_END);
			valid[k] = alive;
			enc[k] = alive && (d >= CELLNUM_ENCODED);
			c[k] = enc[k] ? (int)(d >> CELLNUM_SHIFT) - 1 : (k ? c[k > 0 ? k - 1 : 0] : cell);
			cb[k] = 0;
			if (enc[k]) {
				const int cz = c[k]/9, cy = (c[k] - cz*9)/3, cx = c[k] - cz*9 - cy*3;
				cb[k] = a.cellStart[grid_hash_periodic(p, s.gridPos.x + cx - 1, s.gridPos.y + cy - 1, s.gridPos.z + cz - 1)];
			}
		}
		done = !alive;
#pragma unroll
		for (int k = 0; k < NB; ++k)
			if (!enc[k]) cb[k] = k ? cb[k > 0 ? k - 1 : 0] : cell_base;
		cell = c[NB - 1];
		cell_base = cb[NB - 1];

		// stage 2: gather the neighbour rows of the whole batch
		float4 npos[NB], nvel[NB], naux[NB];
		bool same[NB];
		uint32_t nfl[NB];
		float ntau[NB][6];
#pragma unroll
		for (int k = 0; k < NB; ++k) {
			const uint32_t j = valid[k] ? cb[k] + (nd[k] & NEIBINDEX_MASK) : index;
			npos[k] = a.pos[j];
			nvel[k] = a.vel[j];
			naux[k] = a.aux[j];
			same[k] = true; nfl[k] = 0u;
			if (MULTIFLUID) { nfl[k] = FLUID_NUM(a.info[j]); same[k] = nfl[k] == s.fl; }
			if (TURB_MODEL(TURB) == SPHX_SPS && MOMENTUM) {
				const float2 t0 = a.tau0[j], t1 = a.tau1[j], t2 = a.tau2[j];
				ntau[k][0] = t0.x; ntau[k][1] = t0.y; ntau[k][2] = t1.x; ntau[k][3] = t1.y; ntau[k][4] = t2.x; ntau[k][5] = t2.y;
			}
		}

		// stage 3: pair interactions, in list order
#pragma unroll
		for (int k = 0; k < NB; ++k) {
			const int cz = c[k]/9, cy = (c[k] - cz*9)/3, cx = c[k] - cz*9 - cy*3;
			const float pcx = fmaf(-(float)(cx - 1), p.cs[0], s.pos.x);
			const float pcy = fmaf(-(float)(cy - 1), p.cs[1], s.pos.y);
			const float pcz = fmaf(-(float)(cz - 1), p.cs[2], s.pos.z);
			if (LJW)
				lj_interact(p, pcx, pcy, pcz, npos[k], valid[k], force);
			else
				pair_interact<KERNEL, TURB, COLAGROSSI, MOMENTUM, DIFFUSE>(p, s, inv_h, pcx, pcy, pcz,
					npos[k], nvel[k], naux[k], same[k], valid[k], ntau[k], force, true, true, nfl[k], true);
		}
#pragma unroll
		for (int k = 0; k < NB; ++k) nd[k] = ndn[k];
	}
}

template<int KERNEL, int TURB, int COLAGROSSI, bool MULTIFLUID>
__global__ void __launch_bounds__(SPHX_BLOCK_FORCES)
forces_kernel(DevParams p, ForcesArgs a, const uint32_t *__restrict__ runIfNonZero)
{
	// when the tiled kernel handles this launch the generic one is skipped on the device side
	if (runIfNonZero && *runIfNonZero == 0) return;
	// block-stride loop: as the stand-by of the tiled kernel this one is launched with a small grid (a quarter
	// of a million workgroups that return at once cost 50 us), and still covers every block if it has to run
	for (uint32_t blk = blockIdx.x; blk < a.numBlocks; blk += gridDim.x) {
	const uint32_t index = blk*SPHX_BLOCK_FORCES + threadIdx.x + a.fromParticle;
	float cfl_term = 0.0f;

	do {
		if (index >= a.toParticle) break;
		const particleinfo info = a.info[index];
		const uint32_t ptype = PART_TYPE(info);
		const float4 pos = a.pos[index];
		if (!is_active_w(pos.w)) break;

		Self s;
		load_self<TURB>(p, a, index, info, pos, MULTIFLUID, s);
		const float inv_h = fast_rcp(p.slength);
		const bool dyn = p.boundarytype == SPHX_DYN_BOUNDARY, lj = p.boundarytype == SPHX_LJ_BOUNDARY;

		// the caller clobbers FORCES to 0 before basicstep (src/GPUWorker.cc:1949): the accumulator
		// starts from that value without re-reading it
		float4 force = make_float4(0.0f, 0.0f, 0.0f, 0.0f);

		if (ptype == PT_FLUID) {
			// fluid <- fluid : compute_all_pp_interaction (forces_kernel.def:3565-3610)
			walk_section<KERNEL, TURB, COLAGROSSI, MULTIFLUID, PT_FLUID, true, true>(p, a, index, s, inv_h, force);
			// fluid <- boundary : same interaction for DYN_BOUNDARY (forces_kernel.def:3717-3726),
			// no density diffusion from boundary neighbours (:1596-1606)
			// ... Lennard-Jones repulsion for LJ_BOUNDARY (:3688-3705)
			if (dyn)
				walk_section<KERNEL, TURB, COLAGROSSI, MULTIFLUID, PT_BOUNDARY, true, false>(p, a, index, s, inv_h, force);
			else if (lj)
				walk_section<KERNEL, TURB, COLAGROSSI, MULTIFLUID, PT_BOUNDARY, true, false, true>(p, a, index, s, inv_h, force);
		} else if (ptype == PT_BOUNDARY && lj) {
			// boundary <- fluid with LJ_BOUNDARY (:3620-3645): only particles of bodies with force feedback, and only
			// when the caller asked for object forces (run_forces launches this pass only then, src/cuda/forces.cu:775-782)
			if (HAS_COMPUTE_FORCE(info) && a.compute_object_forces)
				walk_section<KERNEL, TURB, COLAGROSSI, MULTIFLUID, PT_FLUID, true, false, true>(p, a, index, s, inv_h, force);
		} else if (ptype == PT_BOUNDARY && dyn) {
			// boundary <- fluid (forces_kernel.def:3650-3679): DYN always evolves density; momentum
			// only for particles of bodies with force feedback
			if (HAS_COMPUTE_FORCE(info))
				walk_section<KERNEL, TURB, COLAGROSSI, MULTIFLUID, PT_FLUID, true, true>(p, a, index, s, inv_h, force);
			else
				walk_section<KERNEL, TURB, COLAGROSSI, MULTIFLUID, PT_FLUID, false, true>(p, a, index, s, inv_h, force);
		}
		cfl_term = finalize_particle(p, a, index, info, s, force);
	} while (0);

	// maxBlockReduce (src/cuda/device_core.cu:40-59) as wave shuffles + one LDS word per wave
	if (a.cfl) {
		__shared__ float wave_max[SPHX_BLOCK_FORCES/64];
#pragma unroll
		for (int d = 32; d > 0; d >>= 1)
			cfl_term = fmaxf(cfl_term, __shfl_down(cfl_term, d));
		if ((threadIdx.x & 63u) == 0) wave_max[threadIdx.x >> 6] = cfl_term;
		__syncthreads();
		if (threadIdx.x == 0) {
			float m = wave_max[0];
			for (int w = 1; w < SPHX_BLOCK_FORCES/64; ++w) m = fmaxf(m, wave_max[w]);
			a.cfl[a.cflOffset + blk] = m;
		}
		__syncthreads();   // wave_max is reused by the next block of this workgroup
	}
	}
}

// the last workgroup of a launch re-arms the tile tickets for the next one (no memset launch in between)
__device__ __forceinline__ void tile_group_done(uint32_t *tileCtl)
{
	__threadfence();
	if (atomicAdd(tileCtl + 2, 1u) == gridDim.x - 1u) {
#pragma unroll
		for (int k = 0; k < 8; ++k) tileCtl[4 + k] = 0u;
		tileCtl[2] = 0u;
	}
}

// ==========================================================================================
// Tiled path (the fast one): LDS staging of the neighbour window per workgroup.
//
// A tile is a block of k x 2 x 2 cells (k consecutive cells along COORD1 in four adjacent grid rows) holding
// <= TILE_THREADS particles; its neighbour window is the (k+2) x 4 x 4 block around it, i.e. 16 contiguous
// particle ranges, because cells along COORD1 are contiguous in the sorted particle arrays.  The workgroup
// copies those ranges (pos, vel, EOS aux: 48 B/particle) into LDS once by DMA, then every thread walks its
// particle's neighbour list reading the neighbour rows from LDS with ds_read_b128 instead of gathering them
// through L1/L2 (where a 64-lane gather touches 12-20 different cache lines per instruction and the L2->L1
// line fills, ~1/4 used, were the bottleneck).  Tiles are produced by build_tiles_kernel (neibs.hip) at
// neighbour-list build time.  Persistent grid: one workgroup per CU (the window is the LDS), tiles handed out
// by per-XCD ticket counters.  DESIGN.md 5.2 has the full account.
// ==========================================================================================
// list entries of one section, TILE_AHEAD batches deep: buffer q[j] holds batch j (mod TILE_AHEAD)
struct ListWindow { uint2 q[TILE_AHEAD]; };   // a batch = 4 uint16 entries in 8 bytes

// workgroup barrier that orders LDS traffic only.  __syncthreads() is a release/acquire fence over ALL address
// spaces: with global stores or loads in flight it drains vmcnt, i.e. it exposes an HBM round trip (the forces
// stores of the previous tile, the list batches just requested) at every barrier of the tile loop.
__device__ __forceinline__ void lds_barrier()
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
	__builtin_amdgcn_s_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__device__ __forceinline__ bool wave_any(bool x) { return __builtin_amdgcn_ballot_w64(x) != 0ull; }

// Tile lists.  The reference-format neighbour list (u16 entries: index within the neighbour cell, cell code on the
// first entry of each cell run) costs the pair loop a running cell code, two dependent LDS table look-ups (code ->
// first window slot of that cell as seen from the home cell, then the row itself), the shift of the own position into
// the neighbour's cell frame and a per-lane "still alive" flag: ~20 of ~70 vector instructions per pair plus a serial
// LDS round trip.  All of that depends only on the list and the tiling, i.e. it changes once per neighbour-list build,
// not once per forces pass (20 passes per build), so tile_lists_kernel (below) does it at build time and leaves, per
// stored neighbour, one uint16: the byte offset of the neighbour's row in the tile's window (slot * 16; the window arrays
// are parallel, 16 B per row, <= 4095 rows).  The cell shift is gone from the pair altogether because the window holds
// positions in one frame per tile (tile_shift, sphx_internal.h): the wave that stages a window row converts it.
// (Round 2 kept cell-local positions and a 4-bit code advance in the entry: 7 more vector instructions and one more
// 16-byte LDS read per pair.)
// Rows [0, nF) hold the fluid section, rows [R-nB, R) the boundary section (first entry in row R-1, like the
// reference's list runs down from neibboundpos); nF and nB are PER WAVE (multiples of TILE_LIST_BATCH): lanes with
// shorter lists are padded with offset 0, a dummy row (mass 0, far away), so the pair loop has a scalar trip
// count, no terminator test and no validity flag.  In memory the four rows of a batch are interleaved per particle
// ([batch][particle][4] uint16), so a lane fetches a whole batch with ONE 8-byte buffer load -- a quarter of the vector
// memory instructions of a row-per-load layout, whose 2-byte loads kept the CU's address unit busy for half of the
// pair loop's duration.  The pair loop is wave-uniform, so the batch number is a scalar: SGPR descriptor (slab base) +
// the per-lane byte offset index*8, which is fixed for the whole tile -- no vector address arithmetic at all.
struct StressAcc { float x, y, z, w, u; };   // stress mode: the accumulators that do not fit the float4 of the forces

// one pair of SPSstressMatrixDevice (src/cuda/visc_kernel.cu:759-811) in the tiled kernel: dv_ab -= (v_a - nv_a) r_b F m/rho_n,
// the nine sums in the order and arithmetic of sps_kernel, so both give the same bits (single fluid: rho_n from the
// neighbour's relative density); acc = dv[0..3], acc2 = dv[4..8]
template<int KERNEL>
__device__ __forceinline__ void stress_interact(const DevParams &p, const Self &s, float inv_h, float qx, float qy, float qz,
	const float4 &npos, const float4 &nvel, bool valid, float4 &acc, StressAcc &acc2)
{
	const float rx = qx - npos.x, ry = qy - npos.y, rz = qz - npos.z;
	const float r = fast_sqrt(fmaf(rz, rz, fmaf(ry, ry, rx*rx)));
	const bool on = valid && is_active_w(npos.w) && r < p.influenceradius;
	const float n_rho = (nvel.w + 1.0f)*p.rho0[0];
	const float wgt = kernel_F<KERNEL>(p, r, inv_h)*npos.w*fast_rcp(n_rho);
	const float weight = on ? wgt : 0.0f;
	const float mx = rx*weight, my = ry*weight, mz = rz*weight;
	const float vx = s.vel.x - nvel.x, vy = s.vel.y - nvel.y, vz = s.vel.z - nvel.z;
	acc.x -= vx*mx; acc.y -= vx*my; acc.z -= vx*mz;
	acc.w -= vy*mx; acc2.x -= vy*my; acc2.y -= vy*mz;
	acc2.z -= vz*mx; acc2.w -= vz*my; acc2.u -= vz*mz;
}

// SA_BOUNDARY, density summation (densitySumVolumicDevice, src/cuda/density_sum_kernel.cu:206-250): -m W(r^n) for every stored
// neighbour, + m W(r^n+1) for those still in range, as two sums (acc.x, acc.y).  The window holds the positions at step n and, in
// place of the velocities, the displacement n -> n+1 of every particle: r^n+1 = r^n + (d_i - d_j).  Pad entries have mass 0.
__device__ __forceinline__ void sa_dsum_interact(const DevParams &p, const Self &s, float inv_h, float qx, float qy, float qz,
	const float4 &npos, const float4 &ndisp, bool valid, float4 &acc)
{
	const float rx = qx - npos.x, ry = qy - npos.y, rz = qz - npos.z;
	const float m = valid ? npos.w : 0.0f;
	auto W = [&](float r) {
		const float R = r*inv_h;
		float val = fmaf(-0.5f, R, 1.0f);
		val *= val; val *= val;
		return val*fmaf(2.0f, R, 1.0f)*p.wcoeff;
	};
	const float rN = fast_sqrt(fmaf(rz, rz, fmaf(ry, ry, rx*rx)));
	acc.x = fmaf(-m, W(rN), acc.x);
	const float ux = rx + (s.vel.x - ndisp.x), uy = ry + (s.vel.y - ndisp.y), uz = rz + (s.vel.z - ndisp.z);
	const float rNp1 = fast_sqrt(fmaf(uz, uz, fmaf(uy, uy, ux*ux)));
	acc.y = fmaf((rNp1 < p.influenceradius) ? m : 0.0f, W(rNp1), acc.y);
}

// SA_BOUNDARY, Brezzi density diffusion (computeDensityDiffusionDevice, src/cuda/forces_kernel.def:1766-1783,4515-4560) of one
// fluid neighbour; the EOS rows give the neighbour's pressure and density.  acc.w accumulates, dt2rho = dt * 2 * rho_i
__device__ __forceinline__ void sa_diff_interact(const DevParams &p, const Self &s, float inv_h, float qx, float qy, float qz,
	const float4 &npos, const float4 &naux, bool valid, float dt2rho, float4 &acc)
{
	const float rx = qx - npos.x, ry = qy - npos.y, rz = qz - npos.z;
	const float r = fast_sqrt(fmaf(rz, rz, fmaf(ry, ry, rx*rx)));
	const bool on = valid && r < p.influenceradius;
	const float f = kernel_F<SPHX_WENDLAND>(p, r, inv_h);
	const float gdotr = fmaf(p.gravity[2], rz, fmaf(p.gravity[1], ry, p.gravity[0]*rx));
	const float n_rho = naux.w;
	const float t = p.densityDiffCoeff*fmaf(2.0f*fast_rcp(s.rho + n_rho), s.P - naux.z, -gdotr)*npos.w*fast_rcp(n_rho)*f*dt2rho;
	acc.w += on ? t : 0.0f;
}

#define TILE_HB 2   // pairs per pipeline stage ("half batch")
// window capacity of a tiled-kernel instantiation, and the LDS placement of the SPS rows behind the EOS rows
#define TILE_WC(T) ((TURB_MODEL(T) == SPHX_SPS) ? (((T) & SPHX_TURB_MF) ? TILE_WCAP_SPS : TILE_WCAP_SPS1) : TILE_WCAP)
// SPS, one fluid: no EOS rows in the window, see tau_pack_kernel
#define TILE_SPS_COMPACT(T) (TURB_MODEL(T) == SPHX_SPS && !((T) & SPHX_TURB_MF))
struct Gathered {
	float4 npos[TILE_HB], nvel[TILE_HB], naux[TILE_HB];
	float ntau[TILE_HB][6];   // SPS only
};

__device__ __forceinline__ const float4 &lds_row(const float4 *base, uint32_t byteOffset)
{
	return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + byteOffset);
}

// stage 1 of the pair pipeline: issue the LDS reads of TILE_HB neighbours.  A tile-list entry IS the byte offset of the
// neighbour's row in the window arrays (positions in the tile's frame, see tile_shift): no decode, nothing here depends
// on a previous LDS read, so the round trips overlap with the arithmetic of the previous pairs.
template<int TURB>
__device__ __forceinline__ void gather_half(uint32_t packed, const float4 *sPos, const float4 *sVel, const float4 *sAux, float rho0, Gathered &g)
{
	static_assert(TILE_HB == 2, "a half batch is one 32-bit word of the tile list");
	constexpr uint32_t WS = TILE_WC(TURB) + 1u;
#pragma unroll
	for (int k = 0; k < TILE_HB; ++k) {
		const uint32_t L = k ? packed >> 16 : packed & 0xFFFFu;
		g.npos[k] = lds_row(sPos, L);
		if (!(TURB & SPHX_TURB_SA_DIFF)) g.nvel[k] = lds_row(sVel, L);
		if (TILE_SPS_COMPACT(TURB)) {      // two stress rows, the EOS values in the second one's spare slots (tau_pack_kernel)
			const float4 ta = lds_row(sAux, L), tb = lds_row(sAux + WS, L);
			g.ntau[k][0] = ta.x; g.ntau[k][1] = ta.y; g.ntau[k][2] = ta.z; g.ntau[k][3] = ta.w; g.ntau[k][4] = tb.x; g.ntau[k][5] = tb.y;
			// {P/rho^2, c, P, rho} as eos_kernel makes them; only one of c, P is there, the one the pair reads
			g.naux[k] = make_float4(tb.z, tb.w, tb.w, (g.nvel[k].w + 1.0f)*rho0);
			continue;
		}
		if (!(TURB & (SPHX_TURB_STRESS | SPHX_TURB_SA_DSUM))) g.naux[k] = lds_row(sAux, L);
		if (TURB_MODEL(TURB) == SPHX_SPS) {   // the SPS rows lie behind the EOS rows: sAux[WS + slot], sAux[2 WS + slot]
			const float4 ta = lds_row(sAux + WS, L), tb = lds_row(sAux + 2*WS, L);
			g.ntau[k][0] = ta.x; g.ntau[k][1] = ta.y; g.ntau[k][2] = ta.z; g.ntau[k][3] = ta.w; g.ntau[k][4] = tb.x; g.ntau[k][5] = tb.y;
		}
	}
}

// Two pairs at once on packed fp32 (v_pk_mul/add/fma_f32: two results per instruction at the issue cost of one).
// Same terms in the same order as pair_interact for each of the two pairs.  What is packed: everything computed from
// computed values (r^2, v.r, F, g.r, the viscous and diffusive coefficients).  What is not: the first consumer of every value
// read from LDS (the two rows sit in unrelated registers; pairing them up would cost the moves the packing saves),
// sqrt / rcp / min / compares / selects (no packed forms), and the accumulation into force (list order is kept).
// Instantiated for the Wendland kernel with artificial viscosity, one fluid, no or Colagrossi diffusion.
// Two things differ from pair_interact by rounding only: the positions are in the tile's frame (r_ij = x_i - x_j of two
// shifted positions instead of a shift applied to x_i alone), and the window stores m_j * fcoeff (the constant factor of F
// is folded into the mass when the window is staged), so F here is the bare (q - 2)^3.
// The momentum switch of a lane (boundary particles without force feedback accumulate no acceleration) is applied once per
// particle after the walk, not per pair.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f pk_splat(float a) { return v2f{a, a}; }

// A lane that does not take the section at all (another particle type, an idle lane) computes along and its accumulator is
// put back by the caller (walk_section_lds), so the pair has no validity flag; outside the kernel's support (q >= 2, which
// includes every pad entry: the dummy row is a thousand cells away) F = (q - 2)^3 >= 0 while it is negative inside, so the
// range test is min(m F, 0).
template<int COLAGROSSI>
__device__ __forceinline__ void pair_interact_pk(const DevParams &p, const Self &s, const float3 &q, float inv_h, const Gathered &g,
	bool rt_diffuse, float4 &force)
{
	static_assert(TILE_HB == 2, "two pairs per packed operation");
	const float4 &n0 = g.npos[0], &n1 = g.npos[1];
	const v2f rx = {q.x - n0.x, q.x - n1.x}, ry = {q.y - n0.y, q.y - n1.y}, rz = {q.z - n0.z, q.z - n1.z};
	const v2f r2 = pk_fma(rz, rz, pk_fma(ry, ry, rx*rx));
	const v2f r = {fast_sqrt(r2.x), fast_sqrt(r2.y)};
	const float4 &w0 = g.nvel[0], &w1 = g.nvel[1];
	const v2f vx = {s.vel.x - w0.x, s.vel.x - w1.x}, vy = {s.vel.y - w0.y, s.vel.y - w1.y}, vz = {s.vel.z - w0.z, s.vel.z - w1.z};
	const v2f vel_dot_pos = pk_fma(vz, rz, pk_fma(vy, ry, vx*rx));
	const v2f qm2 = pk_fma(r, pk_splat(inv_h), pk_splat(-2.0f));
	const v2f f = qm2*qm2*qm2;                       // fcoeff rides in the window's mass
	const float4 &a0 = g.naux[0], &a1 = g.naux[1];   // {P/rho^2, c, P, rho}
	const v2f mf = {fminf(n0.w*f.x, 0.0f), fminf(n1.w*f.y, 0.0f)};

	v2f dsel = {0.0f, 0.0f};
	if (COLAGROSSI == DIFF_COLAGROSSI) {
		const v2f gdotr = pk_fma(pk_splat(p.gravity[2]), rz, pk_fma(pk_splat(p.gravity[1]), ry, pk_splat(p.gravity[0])*rx));
		const v2f gr = gdotr*pk_splat(s.rho);
		const bool d0 = rt_diffuse && !(fabsf(s.P - a0.z) < fabsf(gr.x)), d1 = rt_diffuse && !(fabsf(s.P - a1.z) < fabsf(gr.y));
		const v2f ratio = {fmaf(a0.w, s.inv_rho, -1.0f), fmaf(a1.w, s.inv_rho, -1.0f)};
		const v2f dterm = pk_splat(p.densityDiffCoeff*p.sscoeff[s.fl])*ratio*mf;
		dsel = v2f{d0 ? dterm.x : 0.0f, d1 ? dterm.y : 0.0f};
	}
	const v2f drdt = pk_fma(mf, vel_dot_pos, -dsel);
	force.w += drdt.x;
	force.w += drdt.y;

	// compute_pressure_contrib + artvisc, as in pair_interact
	const v2f pgrad = {s.p_precalc + a0.x, s.p_precalc + a1.x};
	v2f kk = -pgrad*mf;
	const v2f vdpn = {fminf(fminf(vel_dot_pos.x, 0.0f), fabsf(w0.w)), fminf(fminf(vel_dot_pos.y, 0.0f), fabsf(w1.w))};
	const v2f ssum = {s.sspeed + a0.y, s.sspeed + a1.y};
	const v2f rsum = {s.rho + a0.w, s.rho + a1.w};
	const v2f den = (r2 + pk_splat(p.epsartvisc))*rsum;
	const v2f iden = {fast_rcp(den.x), fast_rcp(den.y)};
	const v2f visc = vdpn*pk_splat(p.slength*p.artvisccoeff)*ssum*iden;
	kk = pk_fma(visc, mf, kk);
	force.x = fmaf(kk.x, rx.x, force.x); force.y = fmaf(kk.x, ry.x, force.y); force.z = fmaf(kk.x, rz.x, force.z);
	force.x = fmaf(kk.y, rx.y, force.x); force.y = fmaf(kk.y, ry.y, force.y); force.z = fmaf(kk.y, rz.y, force.z);
}

// the instantiations whose pair loop is pair_interact_pk: their window holds m * fcoeff
template<int KERNEL, int TURB, int COLAGROSSI, bool LJ>
struct TilePk { static constexpr bool value = KERNEL == SPHX_WENDLAND && TURB == SPHX_ARTIFICIAL && COLAGROSSI != DIFF_FERRARI && !LJ; };

#ifndef SPHX_ASMRING_ALL
#define SPHX_ASMRING_ALL 0
#endif
#ifndef SPHX_RING_WARMUP
#define SPHX_RING_WARMUP 1
#endif
// the instantiations whose list ring is hand-managed (load_list_b<true>): those that keep every value in registers.  The SPS
// ones spill (their pair holds twelve more values per neighbour) and stay on compiler-managed loads
template<int TURB>
struct TileAsmRing { static constexpr bool value = SPHX_ASMRING_ALL || TURB_MODEL(TURB) != SPHX_SPS; };

// stage 2: the pair interactions of a gathered half, in list order; q = own position in the tile's frame
// LJ = the run uses LJ_BOUNDARY: pairs of the boundary section (ljsec, wave-uniform) and all pairs of boundary
// particles with force feedback (ljlane) are Lennard-Jones repulsions instead of SPH interactions
template<int KERNEL, int TURB, int COLAGROSSI, bool LJ>
__device__ __forceinline__ void compute_half(const DevParams &p, const Gathered &g, const Self &s, const float3 &q, float inv_h,
	bool take, bool momentum, bool diffuse, bool ljsec, bool ljlane, float4 &force, StressAcc &fx)
{
	if (TURB & SPHX_TURB_STRESS) {   // velocity-gradient sums of the SPS stress tensor: force = dv[0..3], fx = dv[4..8]
#pragma unroll
		for (int k = 0; k < TILE_HB; ++k)
			stress_interact<KERNEL>(p, s, inv_h, q.x, q.y, q.z, g.npos[k], g.nvel[k], take, force, fx);
		return;
	}
	if (TURB & SPHX_TURB_SA_DSUM) {
#pragma unroll
		for (int k = 0; k < TILE_HB; ++k)
			sa_dsum_interact(p, s, inv_h, q.x, q.y, q.z, g.npos[k], g.nvel[k], take, force);
		return;
	}
	if (TURB & SPHX_TURB_SA_DIFF) {
#pragma unroll
		for (int k = 0; k < TILE_HB; ++k)
			sa_diff_interact(p, s, inv_h, q.x, q.y, q.z, g.npos[k], g.naux[k], take, s.sa_dt2rho, force);
		return;
	}
	if (LJ && ljsec) {
#pragma unroll
		for (int k = 0; k < TILE_HB; ++k)
			lj_interact(p, q.x, q.y, q.z, g.npos[k], take, force);
		return;
	}
	if (TilePk<KERNEL, TURB, COLAGROSSI, LJ>::value) {
		pair_interact_pk<COLAGROSSI>(p, s, q, inv_h, g, diffuse, force);
		return;
	}
	const bool anyLj = LJ && wave_any(ljlane);
#pragma unroll
	for (int k = 0; k < TILE_HB; ++k) {
		uint32_t nfl = 0u;
		if (TURB & SPHX_TURB_MF) nfl = __float_as_uint(g.naux[k].y) & 3u;     // fluid number tag of the EOS row
		pair_interact<KERNEL, TURB, COLAGROSSI, true, true>(p, s, inv_h, q.x, q.y, q.z,
			g.npos[k], g.nvel[k], g.naux[k], nfl == s.fl, take && !(LJ && ljlane), g.ntau[k], force, momentum, diffuse, nfl, false);
		if (anyLj)
			lj_interact(p, q.x, q.y, q.z, g.npos[k], take && ljlane, force);
	}
}

// Walk the runs of one wave of a tile against the LDS window.
//
// The neighbour lists of a tile's home particles are dealt out to the eight waves of the workgroup BATCH BY BATCH, not
// particle by particle (tile_lists_kernel): the home particles are sorted by list length and cut into chunks of 64 (one per
// lane), the list batches of all chunks are laid end to end -- chunk after chunk, fluid section then second section, every
// section padded to the chunk's longest list -- and every wave takes an eighth of that sequence.  A RUN is the part of one
// chunk's batches that one wave walks; a wave has one to three of them per tile.  At the start of a run the lanes load their
// particle of that chunk (own row from the window), at its end they park the accumulator in LDS (sPart, one slot per run);
// when all waves are through, the finalize stage adds up the slots of each chunk in list order.  What this buys: all eight
// waves (two per SIMD: the fp32 issue rate needs both) stay busy to the end of every tile whatever its number of particles
// and however uneven their lists -- with whole chunks per wave a tile of 363 particles occupied six waves, a tile took as long
// as its longest chunk, and a wave alone on its SIMD issued at half rate.
//  * WAVE-UNIFORM control with scalar trip counts: pad entries are pairs of weight 0 (slot 0 of the window is a dummy record a
//    kilometre away), pair_interact is branch-free, a lane that does not take a section computes along and gets its
//    accumulator back at the end of the section;
//  * the wave's batches are ONE stream in memory (512 B per batch: 64 lanes x 4 window offsets), fetched TILE_AHEAD batches
//    ahead into a ring of register buffers that runs through the section and run boundaries (rotated by unrolling, not by
//    copying: a copy of an in-flight load would wait for it); its first batches are requested one tile ahead;
//  * the LDS reads of the next TILE_HB pairs are issued before the current TILE_HB pairs are computed;
//  * section, momentum and diffusion switches are run-time values so that the pair code exists once in the kernel.
struct ListStream { const uint2 *list; uint32_t *pin; };

typedef uint32_t list_u32x2 __attribute__((ext_vector_type(2)));
// A buffer load the compiler sees.  It then also decides where to wait for it, and in the four-buffer ring of walk_runs it
// decides badly: it rotates the buffers with register copies at the loop latch and has to drain the whole ring
// (s_waitcnt vmcnt(0)) in front of the copies -- once every four batches the walk waits out the latency of its youngest load,
// which was issued one batch before (the state of rounds 1-3: the ISA shows it, the source did not).  Kept for the
// instantiations that spill (AccRing below explains why) and for the requests made one tile ahead.
__device__ __forceinline__ void load_list_b(const ListStream &ls, uint32_t firstBatch /* of this wave, absolute */,
	int batch /* relative */, int lastBatch, uint32_t lane8, uint2 &nd)
{
	// batches past the end are never computed (the walk stops there); the clamp only keeps the prefetch inside the tile's
	// stream.  No branch around the load: one load per step, whatever happens.
	const int b = min(__builtin_amdgcn_readfirstlane(batch), lastBatch);
	const char *base = reinterpret_cast<const char*>(ls.list + ((size_t)__builtin_amdgcn_readfirstlane(firstBatch) + (uint32_t)b)*64u);
	const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, 0xFFFFFFFF, 0x00020000);
	const list_u32x2 d = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)lane8, 0, 0);
	nd = make_uint2(d.x, d.y);
}

// The hand-managed ring: the list batches in flight live in ACCUMULATION registers, which the compiler does not use in a kernel
// that spills nothing (gfx950 gives a wave 512 registers, 256 of them a0..a255; with two waves per SIMD the 256 a wave may
// have are split as the kernel declares).  A value in flight in a register the compiler manages can be copied or spilled by
// it at any moment -- it has no idea the load has not landed (tried: tied asm operands; the compiler peeled the loop and moved
// the buffers between registers) --, a register it never touches cannot.
//   a[2J : 2J+1], J < 4      buffer J of the running tile's ring
//   a[8+2k : 9+2k], k < 4    batch k of the NEXT tile's stream, requested one tile ahead
// All statements are asm volatile (kept in order among themselves) and name the registers in the clobber list, which is also
// what makes the compiler count them in the kernel's register budget.  vmcnt counts loads in issue order and the ring issues
// exactly one load per batch, so "at most three younger loads outstanding" (acc_wait) is "this buffer has landed".
// scripts/check_ring_isa.py verifies on the ISA that nothing but these statements touches an accumulation register.
#define SPHX_ACC_CLOBBERS "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15"
template<int REG>
__device__ __forceinline__ void acc_load(const ListStream &ls, uint32_t firstBatch, int batch, int lastBatch, uint32_t lane8)
{
	const int b = min(__builtin_amdgcn_readfirstlane(batch), lastBatch);
	const char *base = reinterpret_cast<const char*>(ls.list + ((size_t)__builtin_amdgcn_readfirstlane(firstBatch) + (uint32_t)b)*64u);
	static_assert(REG >= 0 && REG < 16 && (REG & 1) == 0, "a register pair");
#define SPHX_ACC_LOAD(R0, R1) asm volatile("global_load_dwordx2 a[" #R0 ":" #R1 "], %0, %1 ; ACCRING" :: "v"(lane8), "s"(base) : SPHX_ACC_CLOBBERS)
	if (REG == 0) SPHX_ACC_LOAD(0, 1); else if (REG == 2) SPHX_ACC_LOAD(2, 3); else if (REG == 4) SPHX_ACC_LOAD(4, 5);
	else if (REG == 6) SPHX_ACC_LOAD(6, 7); else if (REG == 8) SPHX_ACC_LOAD(8, 9); else if (REG == 10) SPHX_ACC_LOAD(10, 11);
	else if (REG == 12) SPHX_ACC_LOAD(12, 13); else SPHX_ACC_LOAD(14, 15);
#undef SPHX_ACC_LOAD
}
// the ring's own loads inside the pair loop: a running scalar base (advanced once per ring cycle by the caller) and an immediate
// offset per buffer -- no clamp, no index arithmetic per batch: past the end of a wave's share the loads read the next share (or the
// allocation's slack behind the last one, sphx_ensure_tile_lists), rows that are gathered and never computed
template<int REG, int OFF>
__device__ __forceinline__ void acc_load_at(const char *base, uint32_t lane8)
{
	static_assert(REG >= 0 && REG < 8 && (REG & 1) == 0 && OFF >= 0 && OFF < 4096, "a register pair of the ring, a 12-bit offset");
#define SPHX_ACC_LOAD_AT(R0, R1) asm volatile("global_load_dwordx2 a[" #R0 ":" #R1 "], %0, %1 offset:%2 ; ACCRING" :: "v"(lane8), "s"(base), "n"(OFF) : SPHX_ACC_CLOBBERS)
	if (REG == 0) SPHX_ACC_LOAD_AT(0, 1); else if (REG == 2) SPHX_ACC_LOAD_AT(2, 3); else if (REG == 4) SPHX_ACC_LOAD_AT(4, 5); else SPHX_ACC_LOAD_AT(6, 7);
#undef SPHX_ACC_LOAD_AT
}
// buffer REG/2 of the ring -> two ordinary registers (the batch has landed: acc_wait, or it was requested a tile ahead)
template<int REG>
__device__ __forceinline__ uint2 acc_read()
{
	uint2 r;
#define SPHX_ACC_READ(R0, R1) asm volatile("v_accvgpr_read_b32 %0, a" #R0 " ; ACCRING\n\tv_accvgpr_read_b32 %1, a" #R1 " ; ACCRING" : "=v"(r.x), "=v"(r.y) :: SPHX_ACC_CLOBBERS)
	if (REG == 0) SPHX_ACC_READ(0, 1); else if (REG == 2) SPHX_ACC_READ(2, 3); else if (REG == 4) SPHX_ACC_READ(4, 5); else if (REG == 6) SPHX_ACC_READ(6, 7);
	else if (REG == 8) SPHX_ACC_READ(8, 9); else if (REG == 10) SPHX_ACC_READ(10, 11); else if (REG == 12) SPHX_ACC_READ(12, 13); else SPHX_ACC_READ(14, 15);
#undef SPHX_ACC_READ
	return r;
}
__device__ __forceinline__ void acc_wait()
{
	static_assert(TILE_AHEAD == 4, "vmcnt(TILE_AHEAD - 1)");
	asm volatile("s_waitcnt vmcnt(3) ; ACCRING" ::: SPHX_ACC_CLOBBERS);
}
// The batches requested one tile ahead (a8..a15) spend the stretch between two pair phases -- the finalize stage, the window
// conversion: long code with temporaries of its own, where the compiler does reach for low accumulation registers -- in ordinary
// registers: acc_fetch_ahead once everything has landed (the caller waited vmcnt(0)), acc_start_ring right before the walk.
// What is left for the compiler to respect are the gaps between the ring's own statements inside the pair loop.
__device__ __forceinline__ void acc_fetch_ahead(ListWindow &lw)
{
	lw.q[0] = acc_read<8>(); lw.q[1] = acc_read<10>(); lw.q[2] = acc_read<12>(); lw.q[3] = acc_read<14>();
}
__device__ __forceinline__ void acc_start_ring(const ListWindow &lw)
{
	asm volatile("v_accvgpr_write_b32 a0, %0 ; ACCRING\n\tv_accvgpr_write_b32 a1, %1 ; ACCRING\n\tv_accvgpr_write_b32 a2, %2 ; ACCRING\n\t"
		"v_accvgpr_write_b32 a3, %3 ; ACCRING\n\tv_accvgpr_write_b32 a4, %4 ; ACCRING\n\tv_accvgpr_write_b32 a5, %5 ; ACCRING\n\t"
		"v_accvgpr_write_b32 a6, %6 ; ACCRING\n\tv_accvgpr_write_b32 a7, %7 ; ACCRING"
		:: "v"(lw.q[0].x), "v"(lw.q[0].y), "v"(lw.q[1].x), "v"(lw.q[1].y), "v"(lw.q[2].x), "v"(lw.q[2].y), "v"(lw.q[3].x), "v"(lw.q[3].y)
		: SPHX_ACC_CLOBBERS);
}

// (compiler-managed ring) LLVM sinks a load whose first use is three ring steps (and several side exits) away down to that use, which
// undoes the prefetch distance.  A use in a never-taken side block (pin is always NULL, but a kernel argument
// the compiler cannot see through) keeps the loads where they are issued; the main path pays one scalar branch.
__device__ __forceinline__ void pin_batch(const ListStream &ls, const uint2 &nd)
{
	if (__builtin_expect(ls.pin != nullptr, 0))
		ls.pin[threadIdx.x] = nd.x ^ nd.y;
}

// flags of a lane record (tile_lists_kernel): what the pair loop needs of the particle info
#define LANE_TYPE_MASK 7u
#define LANE_COMPUTE_FORCE 8u
#define LANE_FLUID_SHIFT 4
#define LANE_VALID 64u

// the part of a tile's run table a wave works from (all scalar)
struct WaveJob { uint32_t firstRun, nRuns, firstBatch, nBatches, laneBase, chunks; };
__device__ __forceinline__ WaveJob wave_job(uint32_t rt /* this lane's word of the run table */, uint32_t dw /* ... of the descriptor */, uint32_t wave)
{
	WaveJob j;
	const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)rt, (int)wave);
	j.firstRun = w & 31u; j.nRuns = (w >> 5) & 31u;
	j.firstBatch = (uint32_t)__builtin_amdgcn_readlane((int)dw, 14) + ((w >> 10) & 4095u);
	j.nBatches = w >> 22;
	j.laneBase = (uint32_t)__builtin_amdgcn_readlane((int)dw, 15);
	j.chunks = (uint32_t)__builtin_amdgcn_readlane((int)rt, 8) & 255u;
	return j;
}

template<int KERNEL, int TURB, int COLAGROSSI, bool LJ, bool HIW>
__device__ __forceinline__ void walk_runs(const DevParams &p, const ForcesArgs &a, const ListStream &ls, const WaveJob &wj, uint32_t rt,
	uint32_t lane, float inv_h, const float4 *sPos, const float4 *sVel, const float4 *sAux, float4 *sPart, const uint32_t *sLaneRec,
	float *sVal, ListWindow &lw /* batches 0..TILE_AHEAD-1 of the wave's stream, already requested */)
{
	static_assert(TILE_AHEAD == 4 && TILE_NB == 2*TILE_HB, "the ring below is unrolled by hand: 4 buffers of 2 halves");
	constexpr uint32_t WC = TILE_WC(TURB), WS = WC + 1;
	constexpr bool SPSW = TURB_MODEL(TURB) == SPHX_SPS, SPSC = TILE_SPS_COMPACT(TURB), STRESS = (TURB & SPHX_TURB_STRESS) != 0;
	constexpr bool PREMUL = TilePk<KERNEL, TURB, COLAGROSSI, LJ>::value;
	constexpr bool SA = (TURB & SPHX_TURB_SA_ANY) != 0, SA_DSUM = (TURB & SPHX_TURB_SA_DSUM) != 0, SA_DIFF = (TURB & SPHX_TURB_SA_DIFF) != 0;
	constexpr bool NOAUX = STRESS || SA_DSUM || SPSC, NOVEL = SA_DIFF;
	constexpr uint32_t TAU0 = SPSC ? 0u : WS;
	constexpr bool ASMR = TileAsmRing<TURB>::value;
	const uint32_t lane8 = lane*(uint32_t)(TILE_NB*sizeof(uint16_t));
	const int last = (int)wj.nBatches - 1;
	const bool dyn = p.boundarytype == SPHX_DYN_BOUNDARY;
	int next = TILE_AHEAD;                 // next batch of the stream to fetch (scalar)
	uint32_t run = wj.firstRun;            // scalar
	const uint32_t runsEnd = wj.firstRun + wj.nRuns;

	// state of the run / section being walked
	Self s;
	float3 q = make_float3(0.0f, 0.0f, 0.0f);
	float4 force = make_float4(0.0f, 0.0f, 0.0f, 0.0f), keep = force;
	StressAcc fx = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
	bool take = false, take0 = false, take1 = false, momentum = false, ljlane = false, diffuse = true;
	int sec = 0, leftSeg = 0;
	uint32_t secondLeft = 0;               // batches of the run's second section still to come (scalar)
	uint32_t runWord = 0;                  // the run's table word (scalar)
	s.pos = make_float4(0.0f, 0.0f, 0.0f, 0.0f); s.gridPos = make_int3(0, 0, 0); s.fl = 0u;

	auto start_run = [&]() {
		const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)rt, (int)(TILE_RT_RUN + run));
		runWord = w;
		const uint32_t nF = (w >> 4) & 255u;
		secondLeft = (w >> 12) & 255u;
		// the lanes' particles of this run's chunk.  From LDS (staged with the window): a global load here would sit between
		// the loads of the list ring, whose in-order completion count (s_waitcnt vmcnt(N)) the compiler can then no longer tell
		const uint32_t rec = sLaneRec[(w & 15u)*64u + lane];
		const uint32_t oslot = rec & 0xFFFFu, fl = rec >> 16;
		// own rows from the window (already in the tile's frame); idle lanes read the dummy row
		const float4 opos = lds_row(sPos, oslot), ovel = NOVEL ? make_float4(0.0f, 0.0f, 0.0f, 0.0f) : lds_row(sVel, oslot);
		q = make_float3(opos.x, opos.y, opos.z);
		s.vel = ovel;
		s.fl = (TURB & SPHX_TURB_MF) ? ((fl >> LANE_FLUID_SHIFT) & 3u) : 0u;
		if (STRESS) {
			s.rho = (ovel.w + 1.0f)*p.rho0[0];
		} else if (SPSC) {      // no EOS rows in the window: P/rho^2 and the one of {c, P} the pair reads ride in the stress rows
			const float4 tb = lds_row(sAux + TAU0 + WS, oslot);
			s.p_precalc = tb.z; s.sspeed = tb.w; s.P = tb.w; s.rho = (ovel.w + 1.0f)*p.rho0[0];
			s.inv_rho = fast_rcp(s.rho);
		} else if (!NOAUX) {
			const float4 oaux = lds_row(sAux, oslot);
			s.p_precalc = oaux.x; s.sspeed = oaux.y; s.P = oaux.z; s.rho = oaux.w;
			s.inv_rho = fast_rcp(oaux.w);
		}
		if (SA_DIFF) s.sa_dt2rho = a.saDt*2.0f*s.rho;
		if (TURB & SPHX_TURB_NEWT) init_visc(p, s);
		if (SPSW) {   // own stress tensor
			const float4 ta = lds_row(sAux + TAU0, oslot), tb = lds_row(sAux + TAU0 + WS, oslot);
			s.tau[0] = ta.x; s.tau[1] = ta.y; s.tau[2] = ta.z; s.tau[3] = ta.w; s.tau[4] = tb.x; s.tau[5] = tb.y;
		}
		const uint32_t ptype = fl & LANE_TYPE_MASK;
		const bool active = (fl & LANE_VALID) != 0u, hasCF = (fl & LANE_COMPUTE_FORCE) != 0u;
		const bool isFluid = ptype == PT_FLUID, isBound = ptype == PT_BOUNDARY, isDynBound = isBound && dyn;
		// fluid: fluid section then boundary section (DYN: SPH pairs, LJ: repulsion); DYN boundary: fluid section
		// only, with the momentum part only for bodies with force feedback (forces_kernel.def:3650-3679);
		// LJ boundary: fluid section only for bodies with force feedback, when object forces are asked for (:3620-3645)
		momentum = isFluid || hasCF;
		ljlane = LJ && isBound;
		// SA_BOUNDARY modes: fluid particles only, fluid section then vertex section (none in the diffusion)
		take0 = active && (SA ? isFluid : (STRESS || isFluid || isDynBound || (ljlane && hasCF && a.compute_object_forces)));
		take1 = active && (SA ? (isFluid && !SA_DIFF) : (STRESS || (isFluid && (dyn || LJ))));
		force = make_float4(0.0f, 0.0f, 0.0f, 0.0f); keep = force;
		fx.x = 0.0f; fx.y = 0.0f; fx.z = 0.0f; fx.w = 0.0f; fx.u = 0.0f;
		if (nF) { sec = 0; leftSeg = (int)nF; take = take0; diffuse = true; }
		else { sec = 1; leftSeg = (int)secondLeft; secondLeft = 0u; take = take1; diffuse = false; }
	};
	// end of a section: false when the wave has walked its last run
	auto end_segment = [&]() -> bool {
		// pair_interact_pk has no per-lane validity flag: a lane that does not take the section gets its accumulator back
		force.x = take ? force.x : keep.x; force.y = take ? force.y : keep.y; force.z = take ? force.z : keep.z; force.w = take ? force.w : keep.w;
		if (sec == 0 && secondLeft) {
			sec = 1; leftSeg = (int)secondLeft; secondLeft = 0u; take = take1; diffuse = false; keep = force;
			return true;
		}
		if (PREMUL && !momentum) { force.x = 0.0f; force.y = 0.0f; force.z = 0.0f; }   // ... and no momentum switch
		float4 *dst = sPart + (size_t)run*64u + lane;
		dst[0] = force;
		if (STRESS) {
			dst[TILE_RUNS_MAX*64] = make_float4(fx.x, fx.y, fx.z, fx.w);
			dst[2*TILE_RUNS_MAX*64] = make_float4(fx.u, 0.0f, 0.0f, 0.0f);
		}
		// the chunk's last run leaves what the finalize stage needs of the particle itself: its sound speed (CFL term), or in
		// stress mode its relative density
		if (runWord & TILE_RUN_LAST) sVal[(runWord & 15u)*64u + lane] = STRESS ? s.vel.w : s.sspeed;
		if (++run == runsEnd) return false;
		start_run();
		return true;
	};

	start_run();
	Gathered A, B;
	// one batch per step; the first half of the NEXT batch is gathered before the second half of this one is computed,
	// also after the last batch (a clamped, in-bounds prefetch whose rows are never computed).
	// The two waves of a SIMD (w and w + 4) take turns at being the one the issue arbiter prefers, batch by batch (priorities
	// 3,3,0,0 against 2,1,2,1): left alone the older wave always wins and the younger one is starved while both have work
	constexpr bool hiw = HIW;      // which of the two waves of its SIMD this one is: a copy of the walk per value, no branch per batch
#ifndef TILE_PRIO_VARIANT
#define TILE_PRIO_VARIANT 0
#endif
#if TILE_PRIO_VARIANT == 0
#define SPHX_RING_PRIO(J) \
	if (hiw) { if ((J) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(2); } \
	else { if ((J) < 2) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0); }
#elif TILE_PRIO_VARIANT == 1
#define SPHX_RING_PRIO(J)
#elif TILE_PRIO_VARIANT == 2
#define SPHX_RING_PRIO(J) \
	if (hiw) { if ((J) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(2); } \
	else { if ((J) < 3) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0); }
#elif TILE_PRIO_VARIANT == 3
#define SPHX_RING_PRIO(J) \
	if (hiw) { __builtin_amdgcn_s_setprio(1); } \
	else { if ((J) < 2) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0); }
#else
#define SPHX_RING_PRIO(J) \
	if (hiw) { if ((J) & 1) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0); } \
	else { if ((J) & 1) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(3); }
#endif
	if (ASMR) {
		// the ring lives in a0..a7 (AccRing above); `cur` is the batch being walked, in ordinary registers
		acc_start_ring(lw);
		const char *ringBase = reinterpret_cast<const char*>(ls.list + ((size_t)__builtin_amdgcn_readfirstlane(wj.firstBatch) + TILE_AHEAD)*64u);
		uint2 cur = lw.q[0];
		gather_half<TURB>(cur.x, sPos, sVel, sAux, p.rho0[0], A);
#define SPHX_RING_STEP(J, JN, WAIT) \
		SPHX_RING_PRIO(J) \
		gather_half<TURB>(cur.y, sPos, sVel, sAux, p.rho0[0], B); \
		compute_half<KERNEL, TURB, COLAGROSSI, LJ>(p, A, s, q, inv_h, take, momentum, diffuse, sec == 1, ljlane, force, fx); \
		acc_load_at<2*(J), (J)*512>(ringBase, lane8);      /* buffer J is free: its batch is in `cur` */ \
		if ((J) == TILE_AHEAD - 1) ringBase += TILE_AHEAD*512; \
		/* the batch to gather next: requested a tile ahead if it is one of the first TILE_AHEAD (then fewer than TILE_AHEAD loads \
		 * are in flight and the wait is a no-op), else TILE_AHEAD - 1 loads ago */ \
		/* ... and the first TILE_AHEAD - 1 steps of a tile take buffers that acc_start_ring wrote from ordinary registers: with   \
		 * RING_WARMUP they do not wait at all -- vmcnt(3) there waited for the requests made for the NEXT tile moments before     \
		 * (four list batches, two lane-index rows: all older than the ring's first load), a memory latency per tile that nothing  \
		 * needed: -0.9 % per launch at 32 M, -0.7 % at 8 M (profiles/r06_tile_chain_experiments.txt) */ \
		if (WAIT) acc_wait(); \
		cur = acc_read<2*(JN)>(); \
		gather_half<TURB>(cur.x, sPos, sVel, sAux, p.rho0[0], A); \
		compute_half<KERNEL, TURB, COLAGROSSI, LJ>(p, B, s, q, inv_h, take, momentum, diffuse, sec == 1, ljlane, force, fx); \
		if (--leftSeg == 0) { if (!end_segment()) { __builtin_amdgcn_s_setprio(0); return; } }
		// (not for the LJ_BOUNDARY instantiations: with a second copy of their pair code the compiler parks values in a0 inside the
		// walk -- scripts/check_ring_isa.py finds it --, and they are not what a step of theirs waits for)
		if (SPHX_RING_WARMUP && !LJ) {      // the tile's first ring cycle, peeled: nothing to wait for in its first TILE_AHEAD - 1 steps
			SPHX_RING_STEP(0, 1, false)
			SPHX_RING_STEP(1, 2, false)
			SPHX_RING_STEP(2, 3, false)
			SPHX_RING_STEP(3, 0, true)
		}
		for (;;) {
			SPHX_RING_STEP(0, 1, true)
			SPHX_RING_STEP(1, 2, true)
			SPHX_RING_STEP(2, 3, true)
			SPHX_RING_STEP(3, 0, true)
		}
#undef SPHX_RING_STEP
	} else {
		gather_half<TURB>(lw.q[0].x, sPos, sVel, sAux, p.rho0[0], A);
#define SPHX_RING_STEP(J, JN) \
		SPHX_RING_PRIO(J) \
		gather_half<TURB>(lw.q[J].y, sPos, sVel, sAux, p.rho0[0], B); \
		compute_half<KERNEL, TURB, COLAGROSSI, LJ>(p, A, s, q, inv_h, take, momentum, diffuse, sec == 1, ljlane, force, fx); \
		load_list_b(ls, wj.firstBatch, next, last, lane8, lw.q[J]); \
		pin_batch(ls, lw.q[J]); \
		++next; \
		gather_half<TURB>(lw.q[JN].x, sPos, sVel, sAux, p.rho0[0], A); \
		compute_half<KERNEL, TURB, COLAGROSSI, LJ>(p, B, s, q, inv_h, take, momentum, diffuse, sec == 1, ljlane, force, fx); \
		if (--leftSeg == 0) { if (!end_segment()) { __builtin_amdgcn_s_setprio(0); return; } }
		for (;;) {
			SPHX_RING_STEP(0, 1)
			SPHX_RING_STEP(1, 2)
			SPHX_RING_STEP(2, 3)
			SPHX_RING_STEP(3, 0)
		}
	}
#undef SPHX_RING_PRIO
#undef SPHX_RING_STEP
}

// tail of SPSstressMatrixDevice (src/cuda/visc_kernel.cu:780-811): shear rate -> nu_SPS, tau
__device__ __forceinline__ void stress_finalize(const DevParams &p, const ForcesArgs &a, uint32_t index, float rho,
	const float4 &acc, const StressAcc &acc2)
{
	float txx = acc.x, txy = acc.y + acc.w, txz = acc.z + acc2.z;
	float tyy = acc2.x, tyz = acc2.y + acc2.w, tzz = acc2.u;
	const float SijSij_bytwo = 2.0f*(txx*txx + tyy*tyy + tzz*tzz) + (txy*txy + txz*txz + tyz*tyz);
	const float S = sqrtf(SijSij_bytwo);
	const float nu_SPS = p.smagfactor*S;
	const float divu_SPS = 0.6666666666f*nu_SPS*(txx + tyy + tzz);
	const float Blinetal_SPS = p.kspsfactor*SijSij_bytwo;
	if (a.oturbvisc) a.oturbvisc[index] = nu_SPS;
	if (a.otau0) {
		txx = (nu_SPS*(txx + txx) - divu_SPS - Blinetal_SPS)/rho;
		txy *= nu_SPS/rho;
		txz *= nu_SPS/rho;
		tyy = (nu_SPS*(tyy + tyy) - divu_SPS - Blinetal_SPS)/rho;
		tyz *= nu_SPS/rho;
		tzz = (nu_SPS*(tzz + tzz) - divu_SPS - Blinetal_SPS)/rho;
		a.otau0[index] = make_float2(txx, txy);
		a.otau1[index] = make_float2(txz, tyy);
		a.otau2[index] = make_float2(tyz, tzz);
	}
}

// the two window rows a wave stages (wave w: rows w and w + 8), from tile_rows.  The table is requested one tile
// ahead and only turned into scalars (the LDS-DMA destination and the trip counts derive from them) when its tile
// starts: a readlane at request time would wait for the load there and then
// (a workgroup of TILE_WAVES waves: wave w stages the TILE_RPW rows w, w + TILE_WAVES, ...)
static_assert(TILE_WAVES*TILE_RPW == TILE_WROWS && (TILE_WAVES % 2) == 0 && TILE_ROWDESC == 32 && TILE_DESC == 16, "the window rows are dealt out evenly, two rows per table word; tables of 32 and 16 words");
static_assert((TILE_WAVES == 8 || TILE_WAVES == 4) && TILE_CHUNKS <= 2*TILE_WAVES && TILE_CHUNKS <= 15 && TILE_RUNS_MAX <= 24 && TILE_RT_RUN + TILE_RUNS_MAX <= TILE_RUNTAB && TILE_RUNTAB <= 64,
	"run table: a wave finalizes the chunks w and w + TILE_WAVES; chunk numbers are 4 bits, run numbers 5; one table word per lane");
// (every 64-lane load costs the CU's address unit the same ~16 cycles however little it fetches, and that unit is what bounds the
// staging of a tile: a table is fetched with ONE load, a word per lane, and read out with v_readlane)
struct RowJobs { uint32_t start[TILE_RPW], total[TILE_RPW], base[TILE_RPW]; bool contig[TILE_RPW]; };
__device__ __forceinline__ uint32_t request_row_jobs(const uint32_t *__restrict__ tileRows, uint32_t tile, uint32_t lane)
{
	return tileRows[(size_t)TILE_ROWDESC*tile + (lane & (uint32_t)(TILE_ROWDESC - 1))];
}
__device__ __forceinline__ void resolve_row_jobs(uint32_t rr /* this lane's word of the tile's row table */, uint32_t wave, RowJobs &j)
{
	const uint32_t sh = 16u*(wave & 1u);
#pragma unroll
	for (int k = 0; k < TILE_RPW; ++k) {
		const uint32_t row = wave + (uint32_t)(TILE_WAVES*k);      // same parity as the wave: TILE_WAVES is even
		j.start[k] = (uint32_t)__builtin_amdgcn_readlane((int)rr, (int)row);
		const uint32_t tb = ((uint32_t)__builtin_amdgcn_readlane((int)rr, (int)(16u + (row >> 1))) >> sh) & 0xFFFFu;
		const uint32_t bb = ((uint32_t)__builtin_amdgcn_readlane((int)rr, (int)(24u + (row >> 1))) >> sh) & 0xFFFFu;
		j.total[k] = tb; j.base[k] = bb & 0x7FFFu; j.contig[k] = !(bb & 0x8000u);
	}
}

#define TILE_HCH (TILE_RPW > 2 ? 3 : 5)   // 64-record chunks of a window row whose cell hashes are fetched along with the DMA (longer rows: later)

// does the tile hold particles of [fromParticle, toParticle)?  (multi-GPU stripes launch the kernel on a range)
__device__ __forceinline__ bool tile_in_range(uint32_t dw /* this lane's word of the descriptor */, uint32_t fromParticle, uint32_t toParticle)
{
	uint32_t firstMin = 0xFFFFFFFFu, lastMax = 0u;
#pragma unroll
	for (int r = 0; r < TILE_HROWS; ++r) {
		const uint32_t first = (uint32_t)__builtin_amdgcn_readlane((int)dw, 4 + r), cnt = (uint32_t)__builtin_amdgcn_readlane((int)dw, 8 + r);
		if (cnt) { firstMin = min(firstMin, first); lastMax = max(lastMax, first + cnt); }
	}
	return !(firstMin >= toParticle || lastMax <= fromParticle);
}

template<int KERNEL, int TURB, int COLAGROSSI, bool LJ>
__global__ void __launch_bounds__(TILE_THREADS, 2)
forces_tile_kernel(DevParams p, ForcesArgs a, const uint32_t *__restrict__ tiles,
	uint32_t *tileCtl /* [0]=count, [1]=overflow, [2]=finished workgroups, [4..11]=per-XCD tile tickets */,
	const uint32_t *__restrict__ cellEnd)
{
	constexpr uint32_t WC = TILE_WC(TURB);
	constexpr bool SPSW = TURB_MODEL(TURB) == SPHX_SPS;
	constexpr bool SPSC = TILE_SPS_COMPACT(TURB);      // ... whose window holds no EOS rows (one fluid)
	constexpr bool STRESS = (TURB & SPHX_TURB_STRESS) != 0;   // stress mode: no EOS rows, every active particle walks both sections
	constexpr bool PREMUL = TilePk<KERNEL, TURB, COLAGROSSI, LJ>::value;   // the window holds m * fcoeff (pair_interact_pk)
	// SA_BOUNDARY modes: sums over the fluid and vertex neighbours of the fluid particles, finished by sa_bounds.hip
	constexpr bool SA = (TURB & SPHX_TURB_SA_ANY) != 0, SA_DSUM = (TURB & SPHX_TURB_SA_DSUM) != 0, SA_DIFF = (TURB & SPHX_TURB_SA_DIFF) != 0;
	constexpr bool NOAUX = STRESS || SA_DSUM || SPSC;   // no EOS rows in the window
	constexpr bool NOVEL = SA_DIFF;             // no velocity rows
	// positions through registers on their way into the window (stage_window_row): in the plain forces pass only.  The rows held in
	// flight cost registers; every other instantiation pays for them in its pair loop.  Measured per option set with and without
	// (scripts/measure_options.sh, M updates/s): SPS at 8 M 1122 -> 1251 (forces 1.85 -> 1.61 ms, stress 1.18 -> 1.06), two fluids
	// 2268 -> 2389, laminar + Ferrari (StillWater 4 M) 2022 -> 2129, WaveTank 904 -> 972, SA walls with density summation 485 -> 508
	constexpr bool POS_REG = !SPSW && !STRESS && !SA && !(TURB & (SPHX_TURB_MF | SPHX_TURB_NEWT));
	constexpr uint32_t WS = WC + 1;   // window arrays: the dummy row the pad entries of the tile lists point to (slot 0) + WC records
	constexpr int PARTV = STRESS ? 3 : 1;      // float4 rows per lane of a run's partial sums
	__shared__ __attribute__((aligned(16))) float4 sPos[WS];
	__shared__ __attribute__((aligned(16))) float4 sVel[NOVEL ? 1 : WS];
	// SPS: EOS rows, then tau {xx,xy,xz,yy}, then {yz,zz,-,-}; with one fluid only the two stress rows (SPSC)
	__shared__ __attribute__((aligned(16))) float4 sAux[SPSC ? 2*WS : SPSW ? 3*WS : (NOAUX ? 1 : WS)];
	__shared__ __attribute__((aligned(16))) float4 sPart[PARTV*TILE_RUNS_MAX*64];   // partial sums of the tile's runs (walk_runs)
	// lane records of the tile's chunks (own window row | flags << 16): of the tile being walked and of the one being finalized
	__shared__ uint32_t sLaneRec[2][TILE_CHUNKS*64];
	__shared__ float sVal[TILE_CHUNKS*64];                         // per lane of every chunk: see walk_runs
	// the own value the finalize stage needs comes from LDS (sVal) unless the window does not hold it: SPS with one fluid keeps
	// P instead of c with Colagrossi diffusion; SA diffusion needs gamma, the other SA modes nothing
	constexpr bool VAL_LDS = !SPSC && !SA;
	constexpr uint32_t TAU0 = SPSC ? 0u : WS;     // where the stress rows start in sAux
	__shared__ uint32_t sTileQ[2];                                 // [0] first tile, [1] next tile of this workgroup

	if (tileCtl[1]) return;                 // tiling overflowed: the generic kernel handles this launch
	const uint32_t numTiles = tileCtl[0];
	const uint32_t tid = threadIdx.x;
	const uint32_t lane = tid & 63u, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	// XCD-aware tile assignment: workgroup b runs on XCD b % 8 (MI355X_MICROARCH.md), each XCD has its own
	// 4 MB L2.  Consecutive tiles are neighbouring row bundles that share window rows, so in every round of
	// gridDim tiles XCD x takes the 32 CONSECUTIVE tiles [x*32, x*32+32) instead of every 8th one: the
	// shared rows then hit in that XCD's L2.  Rounds still interleave all XCDs, which keeps them balanced
	// (giving each XCD one contiguous eighth of the list measured 40 % slower: unequal work per eighth).
	// Placement only affects speed, never results.
	// Tiles are handed out dynamically, one ticket counter per XCD: ticket n of XCD x is tile
	// (n / perRound)*gridDim + x*perRound + n % perRound, i.e. the same placement as above, but a workgroup that
	// drew cheap tiles simply draws more (static round-robin left the slowest workgroup ~5 % behind the mean).
	// Tickets are drawn two tiles ahead by thread 0 and published through LDS, so the atomic's latency is hidden.
	const bool xcdAware = (gridDim.x & 7u) == 0u;
	const uint32_t xcd = xcdAware ? (blockIdx.x & 7u) : 0u;
	const uint32_t perRound = xcdAware ? (gridDim.x >> 3) : 1u;
	uint32_t src = xcd;   // XCD whose tickets this workgroup currently draws (thread 0 only): its own, until they run out
	auto tile_of = [&](uint32_t from, uint32_t n) -> uint32_t {
		return xcdAware ? (n / perRound)*gridDim.x + from*perRound + n % perRound : n;
	};
	const uint32_t tileEnd = numTiles;
	// a ticket past the end of `src`'s share: steal from the other XCDs' shares (synchronous atomics, but only in
	// the last few tiles of a launch); returns a tile >= tileEnd once every share is exhausted
	auto resolve = [&](uint32_t n) -> uint32_t {
		uint32_t t = tile_of(src, n);
		for (int tries = 1; t >= tileEnd && xcdAware && tries < 8; ++tries) {
			src = (src + 1u) & 7u;
			t = tile_of(src, atomicAdd(tileCtl + 4 + src, 1u));
		}
		return t;
	};
	if (tid == 0) {
		const uint32_t n0 = atomicAdd(tileCtl + 4 + src, 1u);
		sTileQ[0] = resolve(n0);
		const uint32_t n1 = atomicAdd(tileCtl + 4 + src, 1u);
		sTileQ[1] = resolve(n1);
	}
	__syncthreads();
	uint32_t tile = __builtin_amdgcn_readfirstlane(sTileQ[0]);
	if (tile >= tileEnd) {
		if (tid == 0) tile_group_done(tileCtl);
		return;
	}

	if (tid == 32) {   // the dummy row (slot 0): no mass, a kilometre away, finite everywhere -> every term of the pair is +-0
		sPos[0] = make_float4(1.0e3f, 1.0e3f, 1.0e3f, 0.0f);
		sVel[0] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
		sAux[0] = make_float4(0.0f, 1.0f, 0.0f, 1.0f);
		if (SPSW) { sAux[TAU0] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); sAux[TAU0 + WS] = make_float4(0.0f, 0.0f, 0.0f, 1.0f); }
	}
	const float inv_h = fast_rcp(p.slength);
	ListStream stream; stream.list = a.tileList; stream.pin = a.pin;

	// software pipeline over tiles: descriptor, window rows, run table, the lanes' first run and their first list batches of
	// the NEXT tile are fetched while the current one computes, so a tile costs one memory round trip (the window DMA); the
	// particles of the PREVIOUS tile are finalized while that DMA is in flight
	// the tile's descriptor, row table and run table: one load each, a word per lane
	uint32_t dwc = tiles[(size_t)TILE_DESC*tile + (lane & (uint32_t)(TILE_DESC - 1))], dwn = 0u;
	uint32_t rrc = request_row_jobs(a.tileRows, tile, lane), rrn = 0u;
	uint32_t rtc = a.tileRuns[(size_t)TILE_RUNTAB*tile + min(lane, (uint32_t)(TILE_RUNTAB - 1))], rtn = 0u;
	__builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): first descriptor, first run table
	const bool wholeRange = a.wholeRange != 0;

	// what a wave asks for one tile ahead, once that tile's descriptor and run table are there: the first batches of its list
	// stream and the particles it will finalize (chunks `wave` and `wave + 8`)
	struct Ahead { ListWindow lw; uint32_t idx[2]; };
	auto request_ahead = [&](uint32_t dw, uint32_t rt, Ahead &o) {
		const WaveJob j = wave_job(rt, dw, wave);
		if (j.nRuns) {
			const uint32_t lane8 = lane*(uint32_t)(TILE_NB*sizeof(uint16_t));
			if (TileAsmRing<TURB>::value) {
				acc_load<8>(stream, j.firstBatch, 0, (int)j.nBatches - 1, lane8); acc_load<10>(stream, j.firstBatch, 1, (int)j.nBatches - 1, lane8);
				acc_load<12>(stream, j.firstBatch, 2, (int)j.nBatches - 1, lane8); acc_load<14>(stream, j.firstBatch, 3, (int)j.nBatches - 1, lane8);
			} else {
#pragma unroll
				for (int k = 0; k < TILE_AHEAD; ++k) load_list_b(stream, j.firstBatch, k, (int)j.nBatches - 1, lane8, o.lw.q[k]);
			}
		}
#pragma unroll
		for (int k = 0; k < 2; ++k) {
			const uint32_t c = wave + (uint32_t)(TILE_WAVES*k);
			o.idx[k] = (c < j.chunks) ? a.tileLaneIndex[j.laneBase + c*64u + lane] : 0xFFFFFFFFu;
		}
	};
	Ahead ac, an;
	ac.idx[0] = ac.idx[1] = 0xFFFFFFFFu;
#pragma unroll
	for (int k = 0; k < TILE_AHEAD; ++k) ac.lw.q[k] = make_uint2(0u, 0u);
	an = ac;
	if (wholeRange || tile_in_range(dwc, a.fromParticle, a.toParticle)) request_ahead(dwc, rtc, ac);

	// the particles of the tile whose pair phase has just ended, waiting to be finalized: particle, where its partial sums
	// lie, and (where LDS does not have it, see VAL_LDS) one value of its own
	uint32_t fIdx[2] = {0xFFFFFFFFu, 0xFFFFFFFFu};
	float fVal[2] = {0.0f, 0.0f};
	uint32_t par = 0u;                     // scalar: which half of sLaneRec the tile being walked uses
	uint32_t fTab[2] = {0u, 0u};           // scalar: first run | runs << 8 of the chunks this wave finalizes
	bool havePrev = false;
	float cflRun = 0.0f;   // largest CFL term of this lane's particles over all tiles of the workgroup
#ifdef SPHX_TILE_DEBUG_BUILD
	// phase timers (shader cycles, per wave): 0 total, 1 top barrier, 2 descriptor + DMA issue, 3 finalize of the previous tile,
	// 4 landing of the DMA, 5 conversion, 6 window barrier, 7 requests, 8 pair phase, 9 tiles
	const bool prof = a.prof != nullptr;
	unsigned long long tBegin = prof ? __builtin_amdgcn_s_memtime() : 0ull, tp = tBegin, pacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define SPHX_PROF(K) do { if (prof) { const unsigned long long tq = __builtin_amdgcn_s_memtime(); pacc[K] += tq - tp; tp = tq; } } while (0)
#else
#define SPHX_PROF(K) do { } while (0)
#endif

	auto finalize_prev = [&]() {
#pragma unroll
		for (int k = 0; k < 2; ++k) {
			const uint32_t rf = fTab[k] & 255u, rn = (fTab[k] >> 8) & 255u;     // scalar
			if (k && !__builtin_amdgcn_ballot_w64(fIdx[k] != 0xFFFFFFFFu)) continue;
			// the chunk's sums: its runs in list order
			float4 force = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
			StressAcc fx = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
			for (uint32_t j = 0; j < rn; ++j) {
				const float4 *src = sPart + (size_t)(rf + j)*64u + lane;
				const float4 f0 = src[0];
				force.x += f0.x; force.y += f0.y; force.z += f0.z; force.w += f0.w;
				if (STRESS) {
					const float4 f1 = src[TILE_RUNS_MAX*64], f2 = src[2*TILE_RUNS_MAX*64];
					fx.x += f1.x; fx.y += f1.y; fx.z += f1.z; fx.w += f1.w; fx.u += f2.x;
				}
			}
			const uint32_t index = fIdx[k];
			const bool mine = index != 0xFFFFFFFFu && (wholeRange || (index >= a.fromParticle && index < a.toParticle));
			// the lane's flags: the record of the finalized tile is in the other half of sLaneRec
			const uint32_t c = wave + (uint32_t)(TILE_WAVES*k);
			const uint32_t fl = sLaneRec[par ^ 1u][min(c, (uint32_t)(TILE_CHUNKS - 1))*64u + lane] >> 16;
			float val = fVal[k];
			if (VAL_LDS) {
				val = sVal[min(c, (uint32_t)(TILE_CHUNKS - 1))*64u + lane];
				// a chunk none of whose lists has an entry has no run to leave the value: from memory (the forces need it of fluid
				// particles only, the stress tensor of every particle)
				if (rn == 0u && __builtin_amdgcn_ballot_w64(mine && (STRESS || (fl & LANE_TYPE_MASK) == PT_FLUID)))
					val = mine ? (STRESS ? reinterpret_cast<const float*>(a.vel)[4u*(size_t)index + 3u] : reinterpret_cast<const float*>(a.aux)[4u*(size_t)index + 1u]) : 0.0f;
			}
			if (!mine) continue;
			particleinfo info;      // what the finalize stage reads of it: type, force-feedback flag, fluid (rigid-body rows: the real one)
			info.x = (unsigned short)((fl & LANE_TYPE_MASK) | ((fl & LANE_COMPUTE_FORCE) ? FG_COMPUTE_FORCE : 0u));
			info.y = (unsigned short)(((fl >> LANE_FLUID_SHIFT) & 3u) << 12); info.z = 0; info.w = 0;
			if (STRESS) stress_finalize(p, a, index, (val + 1.0f)*p.rho0[0], force, fx);
			else if (SA) {
				if (PART_TYPE(info) == PT_FLUID) {
					if (SA_DSUM) a.forces[index].w = force.y + force.x + 0.0f;     // sumPmwNp1 + sumPmwN
					else if (SA_DIFF) a.forces[index].w = (force.w/val)/p.rho0[0];
					else {
						if (p.simflags & SPHX_ENABLE_DENSITY_SUM) force.w = 0.0f;   // no continuity equation then
						a.forces[index] = force;      // unfinished: sa_forces_kernel adds the boundary elements, divides by gamma, ...
					}
				}
			} else {
				Self s;
				s.pos = make_float4(0.0f, 0.0f, 0.0f, 0.0f); s.gridPos = make_int3(0, 0, 0);
				s.vel = make_float4(0.0f, 0.0f, 0.0f, 0.0f); s.rho = 0.0f;
				s.fl = (TURB & SPHX_TURB_MF) ? FLUID_NUM(info) : 0u;
				s.sspeed = val;
				// planes, terrain and rigid-body rows need the cell-local position, the mass, the cell (and plane friction the
				// velocity and the density): few runs, few particles
				const bool geom = (PART_TYPE(info) == PT_FLUID && (p.simflags & (SPHX_ENABLE_PLANES | SPHX_ENABLE_DEM))) ||
					(HAS_COMPUTE_FORCE(info) && a.rbforces);
				if (geom) {
					s.pos = a.pos[index]; s.gridPos = grid_pos_from_hash(p, a.hash[index] & CELLTYPE_BITMASK);
					s.vel = a.vel[index]; s.rho = a.aux[index].w;
					info = a.info[index];
				}
				cflRun = fmaxf(cflRun, finalize_particle(p, a, index, info, s, force));
			}
		}
	};

	const int gs1 = p.gs1;
	for (;;) {
		uint32_t drawn = 0;
		if (tid == 0) drawn = atomicAdd(tileCtl + 4 + src, 1u);   // the tile after next; consumed after the window barrier
		const int g2 = __builtin_amdgcn_readlane((int)dwc, 0), g3 = __builtin_amdgcn_readlane((int)dwc, 1);
		const int ca = __builtin_amdgcn_readlane((int)dwc, 2), ncells = __builtin_amdgcn_readlane((int)dwc, 3);
		const bool inRange = wholeRange || tile_in_range(dwc, a.fromParticle, a.toParticle);   // tiles outside [fromParticle, toParticle) (multi-GPU stripes) are skipped whole
		const bool pairs = ((uint32_t)__builtin_amdgcn_readlane((int)dwc, 13) & 1u) || STRESS;   // no fluid anywhere in the window: nothing interacts

		SPHX_PROF(8);
		lds_barrier();   // the previous tile's pair phase is over: its partial sums are complete, the window is free
		SPHX_PROF(1);
		// everything requested so far landed long ago (the requests were made before the pair phase); waiting for it HERE
		// keeps the finalize stage below from queueing behind the DMA that is issued next
		__builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
		for (int k = 0; k < 2; ++k) asm volatile("" : "+v"(fVal[k]), "+v"(fIdx[k]));
		// ... which includes what the previous tile's ring fetched past its end and the batches requested for this tile
		if (TileAsmRing<TURB>::value) acc_fetch_ahead(ac.lw);
		const uint32_t nextTile = __builtin_amdgcn_readfirstlane(sTileQ[1]);
		const bool haveNext = nextTile < tileEnd;
		if (haveNext) dwn = tiles[(size_t)TILE_DESC*nextTile + (lane & (uint32_t)(TILE_DESC - 1))];
		RowJobs rjc;
		resolve_row_jobs(rrc, wave, rjc);

		// 1. the window: wave w stages rows w and w + 8 (a row is ~3 chunks of 64 records per array) by LDS DMA and fetches
		//    the cell hash of every record along with it; when both have landed it moves ITS rows into the tile's frame
		//    (tile_shift of the record's cell: row from the row number, column from the hash) -- no other wave has to wait for
		//    that, the one barrier below publishes the finished window
		uint32_t hsh[TILE_RPW][TILE_HCH];
		float4 posr[TILE_RPW][TILE_HCH];
		auto stage_window_row = [&](int k) {
			const uint32_t total = rjc.total[k], base = rjc.base[k], rs = rjc.start[k];
#pragma unroll
			for (int c = 0; c < TILE_HCH; ++c) hsh[k][c] = 0u;
			if (!(inRange && pairs) || !total || base + total > WC || !rjc.contig[k]) return;
			// positions go through registers: they are moved into the tile's frame on their way into the window (a DMA'd row had to
			// be read back from LDS for that); everything else by DMA.  Rows longer than TILE_HCH chunks: the rest by DMA + a pass
			if (POS_REG) {
#pragma unroll
				for (int c = 0; c < TILE_HCH; ++c)
					if ((uint32_t)c*64u + lane < total) posr[k][c] = a.pos[rs + (uint32_t)c*64u + lane];
				if (total > (uint32_t)TILE_HCH*64u)
					stage_row_wave(a.pos + rs + TILE_HCH*64, sPos + 1 + base + TILE_HCH*64, total - (uint32_t)TILE_HCH*64u, lane);
			} else
				stage_row_wave(a.pos + rs, sPos + 1 + base, total, lane);
			if (!NOVEL) stage_row_wave(a.vel + rs, sVel + 1 + base, total, lane);
			if (!NOAUX) stage_row_wave(a.aux + rs, sAux + 1 + base, total, lane);
			if (SPSW) {
				stage_row_wave(a.tauPack + rs, sAux + TAU0 + 1 + base, total, lane);
				stage_row_wave(a.tauPack + a.tauPackN + rs, sAux + TAU0 + WS + 1 + base, total, lane);
			}
#pragma unroll
			for (int c = 0; c < TILE_HCH; ++c)
				if ((uint32_t)c*64u + lane < total) hsh[k][c] = a.hash[rs + (uint32_t)c*64u + lane];
		};
#pragma unroll
		for (int k = 0; k < TILE_RPW; ++k) stage_window_row(k);
		const WaveJob wj = wave_job(rtc, dwc, wave);
		if (inRange) {      // ... and the lane records of the chunks (wave w: chunks w and w + 8), 4 bytes per lane
#pragma unroll
			for (int k = 0; k < 2; ++k) {
				const uint32_t c = wave + (uint32_t)(TILE_WAVES*k);
				if (c < wj.chunks)
					__builtin_amdgcn_global_load_lds((gptr_t)(a.tileLaneRec + wj.laneBase + c*64u + lane), (lptr_t)(sLaneRec[par] + c*64u), 4, 0, 0);
			}
		}
		if (haveNext) {   // behind the DMA in the memory pipeline, consumed when the next tile starts
			rrn = request_row_jobs(a.tileRows, nextTile, lane);
			rtn = a.tileRuns[(size_t)TILE_RUNTAB*nextTile + min(lane, (uint32_t)(TILE_RUNTAB - 1))];
		}
		SPHX_PROF(2);
		// 2. finalize the previous tile while the DMA drains: partial sums from LDS, the particle's own data from LDS and
		//    registers, results to memory.  (What bounds the staging is the rate at which the CU's address unit takes the DMA
		//    instructions, ~16 cycles each; they are all queued first.  Measured: with the finalize stage between the two rows
		//    of a wave the unit runs dry while all eight waves compute, 3.90 -> 4.02 ms per launch)
		if (havePrev) finalize_prev();
		SPHX_PROF(3);
		__builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's LDS-DMA and hashes have landed
		SPHX_PROF(4);
		if (inRange && pairs) {
#pragma unroll
			for (int k = 0; k < TILE_RPW; ++k) {
				const uint32_t total = rjc.total[k], base = rjc.base[k], rs = rjc.start[k];
				const int r = (int)wave + TILE_WAVES*k;
				if (!total || base + total > WC) continue;     // cannot overflow for tiles of build_tiles_kernel
				if (rjc.contig[k]) {
					const int h0 = window_row_hash0(p, g2, g3, r) + ca - 1;      // cell hash of window column 0 of this row
					// shift of a record of window cell (r, col): the row's part is wave-uniform, the column's part is col * cell size
					// along COORD1 (home particles read their own row from the window, so nothing else has to reproduce this sum)
					const float3 sh0 = tile_shift(p, ncells, r, 0);
					const float stx = (p.c1 == 0) ? p.csc1 : 0.0f, sty = (p.c1 == 1) ? p.csc1 : 0.0f, stz = (p.c1 == 2) ? p.csc1 : 0.0f;
					const bool wrap1 = (p.periodic & (1u << p.c1)) != 0u;
					auto shifted = [&](float4 P, uint32_t h) {
						int col = (int)(h & CELLTYPE_BITMASK) - h0;
						if (wrap1) {      // the window wraps around a periodic COORD1
							if (col > ncells + 1) col -= gs1;
							if (col < 0) col += gs1;
						}
						const float fc = (float)col;
						P.x += fmaf(fc, stx, sh0.x); P.y += fmaf(fc, sty, sh0.y); P.z += fmaf(fc, stz, sh0.z);
						if (PREMUL) P.w *= p.fcoeff;
						return P;
					};
#pragma unroll
					for (int c = 0; c < TILE_HCH; ++c)
						if ((uint32_t)c*64u + lane < total) {
							const uint32_t w = 1u + base + (uint32_t)c*64u + lane;
							sPos[w] = shifted(POS_REG ? posr[k][c] : sPos[w], hsh[k][c]);
						}
					for (uint32_t q = (uint32_t)TILE_HCH*64u + lane; q < total; q += 64u) sPos[1u + base + q] = shifted(sPos[1u + base + q], a.hash[rs + q]);
				} else {   // a row that is not one range in memory (cell-type segments of a device map not split on COORD3,
					       // a periodic row lying wholly inside the window): cell by cell, in window order
					uint32_t off = 0;
					for (int col = 0; col < ncells + 2; ++col) {
						const uint32_t h = window_cell_hash(p, g2, g3, ca, ncells, r, col);
						if (h == 0xFFFFFFFFu) continue;
						const uint32_t st = a.cellStart[h];
						if (st == CELL_EMPTY) continue;
						const uint32_t cnt = cellEnd[h] - st, cb = 1u + base + off;
						const float3 sh0 = tile_shift(p, ncells, r, 0);
						const float fc = (float)col;
						const float3 sh = make_float3(fmaf(fc, (p.c1 == 0) ? p.csc1 : 0.0f, sh0.x), fmaf(fc, (p.c1 == 1) ? p.csc1 : 0.0f, sh0.y),
							fmaf(fc, (p.c1 == 2) ? p.csc1 : 0.0f, sh0.z));
						for (uint32_t q = lane; q < cnt; q += 64u) {
							float4 P = a.pos[st + q];
							P.x += sh.x; P.y += sh.y; P.z += sh.z;
							if (PREMUL) P.w *= p.fcoeff;
							sPos[cb + q] = P;
							if (!NOVEL) sVel[cb + q] = a.vel[st + q];
							if (!NOAUX) sAux[cb + q] = a.aux[st + q];
							if (SPSW) { sAux[TAU0 + cb + q] = a.tauPack[st + q]; sAux[TAU0 + WS + cb + q] = a.tauPack[a.tauPackN + st + q]; }
						}
						off += cnt;
					}
				}
			}
		}
		SPHX_PROF(5);
		__syncthreads();                      // everybody's rows are in place and converted; the previous tile's sums are consumed
		SPHX_PROF(6);
		if (tid == 0) sTileQ[1] = resolve(drawn);   // read after the first barrier of the next iteration
		// 3. requests: the next tile's first batches and lanes; the own data of THIS tile's particles for its finalize stage
		const bool nextInRange = haveNext && (wholeRange || tile_in_range(dwn, a.fromParticle, a.toParticle));
		if (nextInRange) request_ahead(dwn, rtn, an);
#pragma unroll
		for (int k = 0; k < 2; ++k) {
			const uint32_t c = wave + (uint32_t)(TILE_WAVES*k);
			fIdx[k] = inRange ? ac.idx[k] : 0xFFFFFFFFu;
			// a tile whose window was not staged (no fluid in reach) has only wall particles at home: no runs were walked
			fTab[k] = (inRange && pairs && c < wj.chunks) ? (uint32_t)__builtin_amdgcn_readlane((int)rtc, (int)(TILE_RT_CHUNK + c)) : 0u;
			if (!VAL_LDS && SA_DIFF && (k == 0 || __builtin_amdgcn_ballot_w64(fIdx[k] != 0xFFFFFFFFu))) {
				const uint32_t li = (fIdx[k] != 0xFFFFFFFFu) ? fIdx[k] : 0u;
				fVal[k] = reinterpret_cast<const float*>(a.saGam)[4u*(size_t)li + 3u];
			}
			if (!VAL_LDS && SPSC && (k == 0 || __builtin_amdgcn_ballot_w64(fIdx[k] != 0xFFFFFFFFu))) {
				const uint32_t li = (fIdx[k] != 0xFFFFFFFFu) ? fIdx[k] : 0u;
				fVal[k] = reinterpret_cast<const float*>(a.aux)[4u*(size_t)li + 1u];
			}
		}

		SPHX_PROF(7);
		// 4. the pair phase: this wave's share of the tile's list batches
		if (inRange && pairs && wj.nRuns) {
			if ((TILE_WAVES > 4) ? (__builtin_amdgcn_readfirstlane(threadIdx.x >> 8) != 0) : (((blockIdx.x/256u) & 1u) != 0u))
				walk_runs<KERNEL, TURB, COLAGROSSI, LJ, true>(p, a, stream, wj, rtc, lane, inv_h, sPos, sVel, sAux, sPart, sLaneRec[par], sVal, ac.lw);
			else
				walk_runs<KERNEL, TURB, COLAGROSSI, LJ, false>(p, a, stream, wj, rtc, lane, inv_h, sPos, sVel, sAux, sPart, sLaneRec[par], sVal, ac.lw);
		}
		havePrev = inRange;
#ifdef SPHX_TILE_DEBUG_BUILD
		if (prof) pacc[9] += 1;
#endif
		if (!haveNext) break;
		tile = nextTile;
		dwc = dwn; rrc = rrn; rtc = rtn;
		ac = an;
		par ^= 1u;
	}
	lds_barrier();
	__builtin_amdgcn_s_waitcnt(0x0F70);
	par ^= 1u;      // finalize_prev reads the other half
	if (havePrev) finalize_prev();
	// CFL: the array is only ever max-reduced (fmaxDevice / dtreduce), so the maxima need not sit in the reference's
	// one-entry-per-128-particles places: every wave maxes its running value into one entry of the caller's range, once per
	// launch.  Non-negative floats order like their bit patterns; the caller zeroed CFL.
	if (!STRESS && !SA && a.cfl) {
#pragma unroll
		for (int dd = 32; dd > 0; dd >>= 1)
			cflRun = fmaxf(cflRun, __shfl_down(cflRun, dd));
		if (lane == 0 && a.numBlocks)
			atomicMax(reinterpret_cast<unsigned int*>(a.cfl + a.cflOffset + (blockIdx.x*(TILE_THREADS/64) + wave) % a.numBlocks), __float_as_uint(cflRun));
	}
	if (tid == 0) tile_group_done(tileCtl);
#ifdef SPHX_TILE_DEBUG_BUILD
	if (prof && lane == 0) {
		unsigned long long *o = a.prof + 10*((size_t)blockIdx.x*(TILE_THREADS/64) + wave);
		pacc[0] = __builtin_amdgcn_s_memtime() - tBegin;
#pragma unroll
		for (int k = 0; k < 10; ++k) o[k] = pacc[k];
	}
#endif
#undef SPHX_PROF
}

// ------------------------------------------------------------------------------------------
// CFL reduction -> dt.  fmaxDevice (src/cuda/forces_kernel.cu:734-793) + dtreduce
// (src/cuda/forces.cu:556-606), finished on the device so that the adaptive dt never has to
// visit the host inside a step.
// ------------------------------------------------------------------------------------------
#define BLOCK_FMAX 256

__device__ __forceinline__ float block_max_256(float v)
{
	__shared__ float wm[4];
#pragma unroll
	for (int d = 32; d > 0; d >>= 1) v = fmaxf(v, __shfl_down(v, d));
	if ((threadIdx.x & 63u) == 0) wm[threadIdx.x >> 6] = v;
	__syncthreads();
	v = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
	__syncthreads();
	return v;
}

static __global__ void __launch_bounds__(BLOCK_FMAX)
fmax_kernel(float *__restrict__ output, const float4 *__restrict__ input, uint32_t numquarts)
{
	float m = 0.0f;
	for (uint32_t i = blockIdx.x*BLOCK_FMAX + threadIdx.x; i < numquarts; i += BLOCK_FMAX*gridDim.x) {
		const float4 v = input[i];
		m = fmaxf(m, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
	}
	m = block_max_256(m);
	if (threadIdx.x == 0) output[blockIdx.x] = m;
}

static __global__ void __launch_bounds__(BLOCK_FMAX)
dt_final_kernel(float *__restrict__ d_dt, const float *__restrict__ partial, uint32_t numPartials,
	float slength, float dtadaptfactor, float sspeed_cfl, float max_kinematic, int viscous, int combine_min)
{
	float m = 0.0f;
	for (uint32_t i = threadIdx.x; i < numPartials; i += BLOCK_FMAX) m = fmaxf(m, partial[i]);
	m = block_max_256(m);
	if (threadIdx.x == 0) {
		float dt = dtadaptfactor*fminf(sqrtf(slength/m), slength/sspeed_cfl);
		if (viscous) {
			float dt_visc = slength*slength/max_kinematic;
			dt_visc = (float)((double)dt_visc*0.125);
			if (dt_visc < dt) dt = dt_visc;
		}
		d_dt[0] = combine_min ? fminf(d_dt[0], dt) : dt;
	}
}

// ------------------------------------------------------------------------------------------
// rigid-body totals: the reference does two thrust::inclusive_scan_by_key and reads the last
// element of each segment (src/cuda/forces.cu:966-1004); only those totals are consumed
// (src/GPUWorker.cc REDUCE_BODIES_FORCES), so they are produced directly: one block per body.
// ------------------------------------------------------------------------------------------
static __global__ void __launch_bounds__(256)
rb_reduce_kernel(const float4 *__restrict__ rbforces, const float4 *__restrict__ rbtorques,
	const uint32_t *__restrict__ rbnum, const uint32_t *__restrict__ lastindex,
	float *__restrict__ totals /* 6 per body */, uint32_t numParts)
{
	const uint32_t body = blockIdx.x;
	const uint32_t last = lastindex[body];
	const uint32_t key = rbnum[last];
	float fx = 0, fy = 0, fz = 0, tx = 0, ty = 0, tz = 0;
	for (uint32_t i = threadIdx.x; i < numParts && i <= last; i += 256) {
		if (rbnum[i] == key) {
			const float4 f = rbforces[i], t = rbtorques[i];
			fx += f.x; fy += f.y; fz += f.z; tx += t.x; ty += t.y; tz += t.z;
		}
	}
	__shared__ float red[6][4];
	float v[6] = { fx, fy, fz, tx, ty, tz };
#pragma unroll
	for (int c = 0; c < 6; ++c) {
#pragma unroll
		for (int d = 32; d > 0; d >>= 1) v[c] += __shfl_down(v[c], d);
		if ((threadIdx.x & 63u) == 0) red[c][threadIdx.x >> 6] = v[c];
	}
	__syncthreads();
	if (threadIdx.x < 6)
		totals[6*body + threadIdx.x] = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
}

// ------------------------------------------------------------------------------------------
// SPS stress tensor: SPSstressMatrixDevice (src/cuda/visc_kernel.cu:759-811), shearRate :307-367
// ------------------------------------------------------------------------------------------
struct SpsArgs {
	float2 *tau0, *tau1, *tau2;
	float *turbvisc;
	const float4 *pos, *vel;
	const particleinfo *info;
	const uint32_t *hash, *cellStart;
	const neibdata *neibsList;
	uint32_t numParticles;
};

// one typed section of the list, NB entries per batch like walk_section: the list entries of the next batch, the cell bases
// and the neighbour rows of a batch are independent loads in flight together (the one-neighbour-at-a-time version spent
// 8.6 ms per call at 8 M particles on dependent L2 round trips); branch-free, rejected pairs get weight 0
template<int KERNEL, int NPTYPE, bool MULTIFLUID>
__device__ __forceinline__ void sps_section(const DevParams &p, const SpsArgs &a, uint32_t index,
	const float4 &pos, const float4 &vel, const int3 &gridPos, float inv_h, float dv[9])
{
	int slot = (NPTYPE == PT_FLUID) ? 0 : (int)p.neibboundpos;
	uint32_t nd[NB], ndn[NB];
	load_list_batch<NPTYPE, NB>(p, a.neibsList, index, slot, nd);
	int cell = 0;
	uint32_t cell_base = 0;
	bool done = false;
	while (!done) {
		slot = (NPTYPE == PT_FLUID) ? slot + NB : slot - NB;
		load_list_batch<NPTYPE, NB>(p, a.neibsList, index, slot, ndn);
		bool valid[NB], enc[NB];
		int c[NB];
		uint32_t cb[NB];
		bool alive = true;
#pragma unroll
		for (int k = 0; k < NB; ++k) {
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
		for (int k = 0; k < NB; ++k)
			if (!enc[k]) cb[k] = k ? cb[k > 0 ? k - 1 : 0] : cell_base;
		cell = c[NB - 1];
		cell_base = cb[NB - 1];
		float4 npos[NB], nvel[NB];
		uint32_t nfl[NB];
#pragma unroll
		for (int k = 0; k < NB; ++k) {
			const uint32_t j = valid[k] ? cb[k] + (nd[k] & NEIBINDEX_MASK) : index;
			npos[k] = a.pos[j];
			nvel[k] = a.vel[j];
			nfl[k] = 0u;
			if (MULTIFLUID) nfl[k] = FLUID_NUM(a.info[j]);
		}
#pragma unroll
		for (int k = 0; k < NB; ++k) {
			const int cz = c[k]/9, cy = (c[k] - cz*9)/3, cx = c[k] - cz*9 - cy*3;
			const float rx = fmaf(-(float)(cx - 1), p.cs[0], pos.x) - npos[k].x;
			const float ry = fmaf(-(float)(cy - 1), p.cs[1], pos.y) - npos[k].y;
			const float rz = fmaf(-(float)(cz - 1), p.cs[2], pos.z) - npos[k].z;
			const float r = fast_sqrt(fmaf(rz, rz, fmaf(ry, ry, rx*rx)));
			const bool on = valid[k] && is_active_w(npos[k].w) && r < p.influenceradius;
			const float n_rho = (nvel[k].w + 1.0f)*p.rho0[nfl[k]];
			const float wgt = kernel_F<KERNEL>(p, r, inv_h)*npos[k].w*fast_rcp(n_rho);
			const float weight = on ? wgt : 0.0f;
			const float mx = rx*weight, my = ry*weight, mz = rz*weight;
			const float vx = vel.x - nvel[k].x, vy = vel.y - nvel[k].y, vz = vel.z - nvel[k].z;
			dv[0] -= vx*mx; dv[1] -= vx*my; dv[2] -= vx*mz;
			dv[3] -= vy*mx; dv[4] -= vy*my; dv[5] -= vy*mz;
			dv[6] -= vz*mx; dv[7] -= vz*my; dv[8] -= vz*mz;
		}
#pragma unroll
		for (int k = 0; k < NB; ++k) nd[k] = ndn[k];
	}
}

template<int KERNEL, bool MULTIFLUID>
__global__ void __launch_bounds__(128)
sps_kernel(DevParams p, SpsArgs a, const uint32_t *runIfNonZero)
{
	if (runIfNonZero && !*runIfNonZero) return;   // the tiled stress pass covered this launch
	// block-stride loop: as the stand-by of the tiled pass this kernel is launched with a capped grid (see forces_kernel)
	for (uint32_t blk = blockIdx.x; blk*128u < a.numParticles; blk += gridDim.x) {
	const uint32_t index = blk*128 + threadIdx.x;
	if (index >= a.numParticles) continue;
	const float4 pos = a.pos[index];
	if (!is_active_w(pos.w)) continue;
	const float4 vel = a.vel[index];
	const particleinfo info = a.info[index];
	const int3 gridPos = grid_pos_from_hash(p, a.hash[index] & CELLTYPE_BITMASK);
	const float inv_h = fast_rcp(p.slength);
	float dv[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
	sps_section<KERNEL, PT_FLUID, MULTIFLUID>(p, a, index, pos, vel, gridPos, inv_h, dv);
	sps_section<KERNEL, PT_BOUNDARY, MULTIFLUID>(p, a, index, pos, vel, gridPos, inv_h, dv);

	float txx = dv[0], txy = dv[1] + dv[3], txz = dv[2] + dv[6];
	float tyy = dv[4], tyz = dv[5] + dv[7], tzz = dv[8];
	const float SijSij_bytwo = 2.0f*(txx*txx + tyy*tyy + tzz*tzz) + (txy*txy + txz*txz + tyz*tyz);
	const float S = sqrtf(SijSij_bytwo);
	const float nu_SPS = p.smagfactor*S;
	const float divu_SPS = 0.6666666666f*nu_SPS*(txx + tyy + tzz);
	const float Blinetal_SPS = p.kspsfactor*SijSij_bytwo;
	if (a.turbvisc) a.turbvisc[index] = nu_SPS;
	if (a.tau0) {
		const float rho = (vel.w + 1.0f)*p.rho0[FLUID_NUM(info)];
		txx = (nu_SPS*(txx + txx) - divu_SPS - Blinetal_SPS)/rho;
		txy *= nu_SPS/rho;
		txz *= nu_SPS/rho;
		tyy = (nu_SPS*(tyy + tyy) - divu_SPS - Blinetal_SPS)/rho;
		tyz *= nu_SPS/rho;
		tzz = (nu_SPS*(tzz + tzz) - divu_SPS - Blinetal_SPS)/rho;
		a.tau0[index] = make_float2(txx, txy);
		a.tau1[index] = make_float2(txz, tyy);
		a.tau2[index] = make_float2(tyz, tzz);
	}
	}
}

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
// This file is compiled five times (see the Makefile): once per SPH kernel type with SPHX_FORCES_PART = that type, for the
// template instantiations of the type (each is minutes of compile time: ~30 tiled + ~30 generic kernels), and once without
// SPHX_FORCES_PART for the C ABI, which reaches the instantiations through the sphx_part_* functions below.
struct SpsArgs;
#define SPHX_PART_DECL(K) \
	int sphx_part_forces_k##K(const sphx_ctx *ctx, dim3 grid, hipStream_t stream, const ForcesArgs &a, bool use_tiles); \
	void sphx_part_stress_k##K(const sphx_ctx *ctx, hipStream_t stream, const ForcesArgs &fa); \
	void sphx_part_sps_k##K(const sphx_ctx *ctx, dim3 grid, hipStream_t stream, const SpsArgs &a, const uint32_t *guard);
SPHX_PART_DECL(1) SPHX_PART_DECL(2) SPHX_PART_DECL(3) SPHX_PART_DECL(4)
#undef SPHX_PART_DECL
// SA_BOUNDARY modes of the tiled kernel (Wendland; its own translation unit: -DSPHX_FORCES_PART=3 -DSPHX_FORCES_SA)
void sphx_part_sa_tile(const sphx_ctx *ctx, hipStream_t stream, const ForcesArgs &fa, int mode, bool newtonian);
static_assert(SPHX_CUBICSPLINE == 1 && SPHX_QUADRATIC == 2 && SPHX_WENDLAND == 3 && SPHX_GAUSSIAN == 4, "part numbering = kernel type");

#ifndef SPHX_FORCES_PART
extern "C" uint32_t sphx_forces_fmax_elements(uint32_t n) { return round_up_u(div_up_u(n, SPHX_BLOCK_FORCES), 4u); }
extern "C" uint32_t sphx_forces_fmax_temp_elements(uint32_t nels)
{
	const uint32_t numquarts = nels/4;
	uint32_t numBlocks = div_up_u(numquarts, BLOCK_FMAX);
	if (numBlocks > 1) {
		numBlocks = round_up_u(numBlocks, 4u);
		if (numBlocks > BLOCK_FMAX*4) numBlocks = BLOCK_FMAX*4;
	}
	return numBlocks;
}
extern "C" uint32_t sphx_forces_round_particles(uint32_t n) { return (n/SPHX_BLOCK_FORCES)*SPHX_BLOCK_FORCES; }
#endif

template<int KERNEL, int TURB, int COLA>
static void launch_forces_mf(bool multifluid, dim3 grid, hipStream_t stream, const DevParams &p, const ForcesArgs &a,
	const uint32_t *guard)
{
	if (guard) grid.x = grid.x < 2048u ? grid.x : 2048u;   // stand-by launch (see forces_kernel)
	if (multifluid)
		forces_kernel<KERNEL, TURB, COLA, true><<<grid, SPHX_BLOCK_FORCES, 0, stream>>>(p, a, guard);
	else
		forces_kernel<KERNEL, TURB, COLA, false><<<grid, SPHX_BLOCK_FORCES, 0, stream>>>(p, a, guard);
}

// sphx_forces_timing: HIP events on the launch stream around the dominant kernel of a forces pass
struct ForcesTimer {
	const sphx_ctx *ctx; hipStream_t stream; hipEvent_t stop;
	ForcesTimer(const sphx_ctx *c, hipStream_t s, bool on) : ctx(c), stream(s), stop(nullptr) {
		if (!on || !c->time_forces || !c->forces_events) return;
		hipEvent_t e0, e1;
		if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return;
		(void)hipEventRecord(e0, s);
		c->forces_events->push_back(std::make_pair(e0, e1));
		stop = e1;
	}
	~ForcesTimer() { if (stop) (void)hipEventRecord(stop, stream); }
};

template<int KERNEL, int TURB, int COLA>
static void launch_tile(const sphx_ctx *ctx, hipStream_t stream, const ForcesArgs &a)
{
	ForcesTimer t(ctx, stream, true);
	if (ctx->dev.boundarytype == SPHX_LJ_BOUNDARY)
		forces_tile_kernel<KERNEL, TURB, COLA, true><<<ctx->tile_grid, TILE_THREADS, 0, stream>>>(ctx->dev, a,
			ctx->tiles, ctx->tile_ctl, ctx->cell_end_copy);
	else
		forces_tile_kernel<KERNEL, TURB, COLA, false><<<ctx->tile_grid, TILE_THREADS, 0, stream>>>(ctx->dev, a,
			ctx->tiles, ctx->tile_ctl, ctx->cell_end_copy);
}

// launches the tiled kernel when the tiling of this neighbour list is available, plus the generic
// kernel guarded by the device-side overflow flag (it returns at once when the tiles were used)
template<int KERNEL>
static int launch_forces_k(const sphx_ctx *ctx, dim3 grid, hipStream_t stream, const ForcesArgs &a, bool use_tiles)
{
	const DevParams &p = ctx->dev;
	const bool mf = p.numfluids > 1;
	const int diff = (p.densitydiff == SPHX_COLAGROSSI) ? DIFF_COLAGROSSI : (p.densitydiff == SPHX_FERRARI) ? DIFF_FERRARI : DIFF_NONE;
	const bool newt = p.rheology == SPHX_NEWTONIAN;
	const uint32_t *guard = use_tiles ? ctx->tile_ctl + 1 : nullptr;
	const bool standby = !use_tiles || ctx->tiles_overflow != 0;   // the host saw the tiling succeed (sphx_neibs_getinfo): no stand-by launch
	ForcesTimer t(ctx, stream, !use_tiles);   // without tiles the generic kernel is the dominant one
	// tiled kernel, then the generic one, guarded by the overflow flag
	// more than one fluid or Ferrari diffusion: tiled only for the Wendland kernel (every such problem of the reference uses
	// it), to bound the number of kernel instantiations; use_tiles is false for the other kernels then
#define SPHX_LAUNCH_TILE(T) do { if (use_tiles) { \
		if (mf) { if (KERNEL == SPHX_WENDLAND) { \
			if (diff == DIFF_COLAGROSSI) launch_tile<SPHX_WENDLAND, (T) | SPHX_TURB_MF, DIFF_COLAGROSSI>(ctx, stream, a); \
			else if (diff == DIFF_FERRARI) launch_tile<SPHX_WENDLAND, (T) | SPHX_TURB_MF, DIFF_FERRARI>(ctx, stream, a); \
			else launch_tile<SPHX_WENDLAND, (T) | SPHX_TURB_MF, DIFF_NONE>(ctx, stream, a); } } \
		else if (diff == DIFF_COLAGROSSI) launch_tile<KERNEL, T, DIFF_COLAGROSSI>(ctx, stream, a); \
		else if (diff == DIFF_FERRARI) { if (KERNEL == SPHX_WENDLAND) launch_tile<SPHX_WENDLAND, T, DIFF_FERRARI>(ctx, stream, a); } \
		else launch_tile<KERNEL, T, DIFF_NONE>(ctx, stream, a); } } while (0)
#define SPHX_LAUNCH_GENERIC(T, G) do { if (!standby) break; \
		if (diff == DIFF_COLAGROSSI) launch_forces_mf<KERNEL, T, DIFF_COLAGROSSI>(mf, grid, stream, p, a, G); \
		else if (diff == DIFF_FERRARI) launch_forces_mf<KERNEL, T, DIFF_FERRARI>(mf, grid, stream, p, a, G); \
		else launch_forces_mf<KERNEL, T, DIFF_NONE>(mf, grid, stream, p, a, G); } while (0)
	switch (p.turbmodel) {
	case SPHX_ARTIFICIAL:
		SPHX_LAUNCH_TILE(SPHX_ARTIFICIAL);
		SPHX_LAUNCH_GENERIC(SPHX_ARTIFICIAL, guard);
		break;
	case SPHX_SPS:
		if (newt) {
			SPHX_LAUNCH_TILE(SPHX_SPS | SPHX_TURB_NEWT);
			SPHX_LAUNCH_GENERIC(SPHX_SPS | SPHX_TURB_NEWT, guard);
		} else {
			SPHX_LAUNCH_TILE(SPHX_SPS);
			SPHX_LAUNCH_GENERIC(SPHX_SPS, guard);
		}
		break;
	case SPHX_LAMINAR_FLOW:
		if (newt) {
			SPHX_LAUNCH_TILE(SPHX_LAMINAR_FLOW | SPHX_TURB_NEWT);
			SPHX_LAUNCH_GENERIC(SPHX_LAMINAR_FLOW | SPHX_TURB_NEWT, guard);
		} else {
			SPHX_LAUNCH_TILE(SPHX_LAMINAR_FLOW);
			SPHX_LAUNCH_GENERIC(SPHX_LAMINAR_FLOW, guard);
		}
		break;
	default:
		return sphx_set_error(SPHX_ERR_UNSUPPORTED, "sphx_forces_basicstep: turbulence model not built");
	}
#undef SPHX_LAUNCH_TILE
#undef SPHX_LAUNCH_GENERIC
	return SPHX_OK;
}

#if defined(SPHX_FORCES_PART) && defined(SPHX_FORCES_SA)
void sphx_part_sa_tile(const sphx_ctx *ctx, hipStream_t stream, const ForcesArgs &fa, int mode, bool newtonian)
{
#define SPHX_SA_TILE(T) forces_tile_kernel<SPHX_WENDLAND, (T), DIFF_NONE, false><<<ctx->tile_grid, TILE_THREADS, 0, stream>>>(ctx->dev, fa, \
		ctx->tiles, ctx->tile_ctl, ctx->cell_end_copy)
	if (mode == SPHX_SA_TILE_DSUM) SPHX_SA_TILE(SPHX_LAMINAR_FLOW | SPHX_TURB_SA_DSUM);
	else if (mode == SPHX_SA_TILE_DIFF) SPHX_SA_TILE(SPHX_LAMINAR_FLOW | SPHX_TURB_SA_DIFF);
	else if (newtonian) SPHX_SA_TILE(SPHX_LAMINAR_FLOW | SPHX_TURB_NEWT | SPHX_TURB_SA);
	else SPHX_SA_TILE(SPHX_LAMINAR_FLOW | SPHX_TURB_SA);
#undef SPHX_SA_TILE
}
#elif defined(SPHX_FORCES_PART) && defined(SPHX_PROBE_PLAIN)
// one instantiation alone, for reading its ISA (scripts/probe_plain_isa.sh): the plain forces pass of the bench workload
void sphx_probe_plain(const sphx_ctx *ctx, hipStream_t stream, const ForcesArgs &a)
{
	forces_tile_kernel<SPHX_WENDLAND, SPHX_ARTIFICIAL, DIFF_COLAGROSSI, false><<<ctx->tile_grid, TILE_THREADS, 0, stream>>>(ctx->dev, a,
		ctx->tiles, ctx->tile_ctl, ctx->cell_end_copy);
}
#elif defined(SPHX_FORCES_PART)
#define SPHX_PASTE2(a, b) a##b
#define SPHX_PASTE(a, b) SPHX_PASTE2(a, b)
int SPHX_PASTE(sphx_part_forces_k, SPHX_FORCES_PART)(const sphx_ctx *ctx, dim3 grid, hipStream_t stream, const ForcesArgs &a, bool use_tiles)
{
	return launch_forces_k<SPHX_FORCES_PART>(ctx, grid, stream, a, use_tiles);
}
void SPHX_PASTE(sphx_part_stress_k, SPHX_FORCES_PART)(const sphx_ctx *ctx, hipStream_t stream, const ForcesArgs &fa)
{
	forces_tile_kernel<SPHX_FORCES_PART, SPHX_ARTIFICIAL | SPHX_TURB_STRESS, DIFF_NONE, false>
		<<<ctx->tile_grid, TILE_THREADS, 0, stream>>>(ctx->dev, fa, ctx->tiles, ctx->tile_ctl, ctx->cell_end_copy);
}
void SPHX_PASTE(sphx_part_sps_k, SPHX_FORCES_PART)(const sphx_ctx *ctx, dim3 grid, hipStream_t stream, const SpsArgs &a, const uint32_t *guard)
{
	if (ctx->dev.numfluids > 1) sps_kernel<SPHX_FORCES_PART, true><<<grid, 128, 0, stream>>>(ctx->dev, a, guard);
	else sps_kernel<SPHX_FORCES_PART, false><<<grid, 128, 0, stream>>>(ctx->dev, a, guard);
}
#else   // the C ABI, to the end of the file

// ------------------------------------------------------------------------------------------
// Tile lists: the reference-format neighbour lists of the tiled particles, translated once per neighbour-list build
// into the form the tiled pair loop walks (see walk_runs above).  One workgroup per tile, one thread per home particle:
//  1. the window layout of the tile (tile_rows: first record, length and first slot of each of the 16 rows, and whether the
//     row is one range in memory): window row r holds the cells of grid row (g2-1+(r&3), g3-1+(r>>2)) from column ca-1 on, rows
//     follow each other without gaps, so the slot of a record is (records of the rows before) + (records of the cells before
//     it in its row) + its index in its cell;
//  2. the home particles sorted by the length of their fluid section, longest first (a stable counting sort on the lengths the
//     list builder left in neib_counts), and cut into chunks of 64: the lanes of a wave then walk lists of nearly equal
//     length (in home order a chunk is padded to its longest list: 14 % of the pair slots were padding);
//  3. the schedule: all batches of the tile (chunk after chunk, fluid section then second section, each padded to the chunk's
//     longest) split evenly over the eight waves of the forces kernel -> runs (a wave's stretch of one chunk), the run table;
//  4. space for the tile's list stream and lane tables from two global cursors (tile_ctl[12], [13]);
//  5. the translation: every lane rewrites its particle's entries as window offsets (slot * 16: the byte offset of the
//     neighbour's row in the window arrays, used as the LDS address as it is), four to a batch, padded with offset 0 (the
//     dummy record), straight into the stream position its wave will read it from (512-byte coalesced stores).
// ------------------------------------------------------------------------------------------
#define TL_THREADS TILE_PMAX
#define TL_BINS 132      // list lengths 0..128 (+ padding)

// which home particle of the tile thread t stands for: the four home rows one after the other (build_tiles_kernel)
struct TileHome { uint32_t index; int hrow; };
__device__ __forceinline__ TileHome tile_home(uint32_t dw /* this lane's word of the descriptor */, uint32_t t)
{
	TileHome h;
	const uint32_t c0 = (uint32_t)__builtin_amdgcn_readlane((int)dw, 8), c1n = (uint32_t)__builtin_amdgcn_readlane((int)dw, 9),
		c2n = (uint32_t)__builtin_amdgcn_readlane((int)dw, 10);
	int hrow = 0; uint32_t hoff = t;
	if (hoff >= c0) { hoff -= c0; hrow = 1; if (hoff >= c1n) { hoff -= c1n; hrow = 2; if (hoff >= c2n) { hoff -= c2n; hrow = 3; } } }
	const uint32_t f0 = (uint32_t)__builtin_amdgcn_readlane((int)dw, 4), f1 = (uint32_t)__builtin_amdgcn_readlane((int)dw, 5),
		f2 = (uint32_t)__builtin_amdgcn_readlane((int)dw, 6), f3 = (uint32_t)__builtin_amdgcn_readlane((int)dw, 7);
	const uint32_t hfirst = (hrow == 0) ? f0 : (hrow == 1) ? f1 : (hrow == 2) ? f2 : f3;
	h.hrow = hrow;
	h.index = hfirst + hoff;
	return h;
}

// The share of a tile's batches that each wave of the forces kernel walks, in 1/256: `lo` for each of the waves 0..3, 64 - lo
// for each of the waves 4..7 (the second wave of every SIMD).  Even shares (lo = 32) for every kernel but one: in the plain
// forces pass (artificial or no viscosity, one fluid, no SA walls) the second wave of a SIMD reaches the end of its pair phase
// first when the shares are equal (phase timers, profiles/r04_tile_phases_32M.txt: waves 0..7 wait 0.06, 0.18, 0.24, 0.37, 0.81,
// 0.77, 0.69, 1.16 M cycles of 9.26 M at the barrier on top of the next tile), and the tile ends when the last wave does.
// Measured at 32 M particles, two boxes (scripts/ab_forces.sh; ms per launch): lo = 32: 3.877;  31: 3.874;  30: 3.822;  29: 3.837;
// weights in proportion to the measured waits (29 30 30 31 34 33 33 36): 3.860, twice that correction: 3.919 -- the waits are not
// pair work that can be handed over one for one (the two waves of a SIMD share its issue slots).  The other instantiations
// LOSE with lo = 30 (scripts/measure_all.sh: SPS forces at 8 M 1.75 -> 1.90 ms per launch, two fluids 1.24 -> 1.35, the laminar +
// Ferrari pass of the StillWater mirror 4.5 % per step): their waves are balanced differently, so they keep even shares
#define TILE_SHARE_LO_PLAIN 30u
__device__ __forceinline__ uint32_t tile_share_start(uint32_t w, uint32_t lo)     // sum of the weights of the waves before w; 256 for w = 8
{
	return w <= 4u ? w*lo : 4u*lo + (w - 4u)*(64u - lo);
}

// Measured at 32 M particles (scripts/ab_builder.sh, round 4; linearisation xzy).  First a warning: with the sixteen list loads
// of the translation loop each under a condition of its own, the compiler made every load wait for all the earlier ones
// (s_waitcnt vmcnt(0) in front of each) in SOME builds, depending on edits elsewhere in the kernel -- the same source ran 7.4 or
// 11.3 ms per launch, and every A/B of this kernel made before that was found compared the two states of the compiler as much as
// the two variants.  The loads are unconditional now (the address is valid for every thread) and issue back to back (checked in
// the ISA): 6.3 ms.  With that: 8 / 16 / 24 / 32 loads in flight per lane 6.58 / 6.27 / 6.81 / 6.71 ms; two workgroups per CU
// 7.24; the tile's stream assembled in LDS and written as one linear copy instead of eight-byte stores from all over the tile
// 6.85 against 6.65 (yzx: 7.49 against 7.25): neither the latency of the loads nor the scattered stores bound it.  Without the
// translation (sections 1 to 4 alone) 2.29 ms, with the fluid section only 5.89: the 3.6 ms of the fluid section are its
// ~15 vector instructions and one LDS look-up per entry at ten waves per CU.  (Tried before the loads were fixed, and to be read
// with the warning above: a second, wave-major copy of the list written by build_neibs_kernel's ring -- this kernel 7.36 -> 6.37 ms,
// build_neibs_kernel 12.9 -> 14.1 ms for the extra stores.)
#ifndef TL_MINWAVES
#define TL_MINWAVES 1
#define TL_LOADS 16
#define TL_GRIDMUL 8
#endif
__global__ void __launch_bounds__(TL_THREADS, TL_MINWAVES)
tile_lists_kernel(DevParams p, const neibdata *__restrict__ list, const uint32_t *__restrict__ neibCounts,
	const particleinfo *__restrict__ info, const uint32_t *__restrict__ hash,
	const uint32_t *__restrict__ cellStart, const uint32_t *__restrict__ cellEnd,
	uint32_t *__restrict__ tiles, uint32_t *tileCtl, uint32_t *__restrict__ tileRows, uint32_t *__restrict__ tileRuns,
	uint2 *__restrict__ tileList, uint32_t listCapBatches, uint32_t *__restrict__ laneRec, uint32_t *__restrict__ laneIndex,
	uint32_t laneCap, int saVertex /* SA_BOUNDARY lists: the second section is the VERTEX section */,
	uint32_t shareLo /* tile_share_start */,
	uint32_t homeFrom, uint32_t homeTo /* only the tiles whose last home particle lies in [homeFrom, homeTo): a build in parts */)
{
	__shared__ uint32_t sCellRel[TILE_WROWS*TILE_KW], sCellBase[TILE_WROWS*TILE_KW], sCellStart[TILE_WROWS*TILE_KW];
	__shared__ uint32_t sRowTotal[TILE_WROWS], sRowStart[TILE_WROWS], sRowContig[TILE_WROWS], sRowBase[TILE_WROWS];
	__shared__ int sCodeOff[32];                                   // cell code-1 -> window-table offset
	__shared__ uint16_t sCB[TILE_HROWS*TILE_MAXCELLS*27 + 8];      // [home cell][code] -> slot of the cell's first record
	__shared__ uint16_t sHist[TILE_CHUNKS][TL_BINS];               // per wave: particles with a fluid section of that length; then: ... in the waves before
	__shared__ uint16_t sBinStart[TL_BINS];
	__shared__ uint32_t sBinWave[4];
	__shared__ uint16_t sLaneOf[TILE_PMAX], sLenF[TILE_PMAX], sLenB[TILE_PMAX];   // lane of the tile by home-order number; list lengths by lane
	__shared__ uint32_t sChunkF[TILE_CHUNKS + 1], sChunkB[TILE_CHUNKS + 1], sChunkStart[TILE_CHUNKS + 1];   // batches per section; first batch
	__shared__ uint32_t sSorted[32];
	__shared__ uint32_t sRunTab[TILE_RUNTAB];
	__shared__ uint32_t sBase[4];                                  // list base, lane base, overflow
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	// tiling overflowed: the generic kernel handles this list.  (One read per workgroup: another workgroup of this very kernel may
	// raise the flag at any time, and threads that disagree about it would not meet at the barriers below)
	if (tid == 0) sBase[3] = tileCtl[1];
	__syncthreads();
	if (sBase[3]) return;
	const uint32_t numTiles = tileCtl[0];
	// neighbour-cell offset (ox,oy,oz) -> index of that cell in the window table, relative to the
	// particle's own (row, column): o1 + KW*(o2+1) + 4*KW*(o3+1) + 1 with (o1,o2,o3) the offsets along COORD1..3
	if (tid < 27) {
		const int c = (int)tid;                    // d_cell_to_offset order: c = (x+1) + 3(y+1) + 9(z+1)
		const int cz = c/9, cy = (c - cz*9)/3;
		const int ox = c - cz*9 - cy*3 - 1, oy = cy - 1, oz = cz - 1;
		const int mx = (p.c1 == 0) ? 1 : (p.c2 == 0) ? TILE_KW : 4*TILE_KW;
		const int my = (p.c1 == 1) ? 1 : (p.c2 == 1) ? TILE_KW : 4*TILE_KW;
		const int mz = (p.c1 == 2) ? 1 : (p.c2 == 2) ? TILE_KW : 4*TILE_KW;
		sCodeOff[c] = ox*mx + oy*my + oz*mz + 1 + TILE_KW + 4*TILE_KW;
	}
	const int wr = (int)(tid/TILE_KW), wcol = (int)(tid - (tid/TILE_KW)*TILE_KW);   // my window cell (tid < 256)
	const size_t stride = (size_t)p.stride;
	// Everything a tile's setup reads from memory -- its descriptor, the window's cells (section 1), the list lengths / hash / info of the
	// thread's home particle (sections 2 and 5) -- depends on the tile's number alone, and the setup is a chain of barriers with nothing to
	// hide a round trip behind (~8 us per tile, half the kernel).  So it is asked for one tile ahead (the descriptor two ahead): while a
	// tile's lists are translated the next tile's rows arrive.  (Round 6: 4.61 -> 4.28 ms at 32 M with the loads at the top of the same
	// tile's setup, profiles/r06_tile_lists_prefetch_ab.txt.)
	struct TilePre { uint32_t index, cnt, hash; particleinfo info; uint32_t wStart, wEnd; };
	auto desc_of = [&](uint32_t t) -> uint32_t { return t < numTiles ? tiles[(size_t)TILE_DESC*t + (lane & (uint32_t)(TILE_DESC - 1))] : 0u; };
	auto ask = [&](uint32_t dd, bool real, TilePre &q) {
		q.index = 0u; q.cnt = 0u; q.hash = 0u; q.info = particleinfo{0, 0, 0, 0}; q.wStart = CELL_EMPTY; q.wEnd = 0u;
		if (!real) return;
		const uint32_t Pn = (uint32_t)(__builtin_amdgcn_readlane((int)dd, 8) + __builtin_amdgcn_readlane((int)dd, 9) +
			__builtin_amdgcn_readlane((int)dd, 10) + __builtin_amdgcn_readlane((int)dd, 11));
		q.index = tile_home(dd, tid < Pn ? tid : 0u).index;
		q.cnt = neibCounts[q.index]; q.hash = hash[q.index]; q.info = info[q.index];
		if (tid < TILE_WROWS*TILE_KW) {
			const uint32_t h = window_cell_hash(p, __builtin_amdgcn_readlane((int)dd, 0), __builtin_amdgcn_readlane((int)dd, 1),
				__builtin_amdgcn_readlane((int)dd, 2), __builtin_amdgcn_readlane((int)dd, 3), wr, wcol);
			if (h != 0xFFFFFFFFu) { q.wStart = cellStart[h]; q.wEnd = cellEnd[h]; }
		}
	};
	uint32_t dAhead = desc_of(blockIdx.x);
	TilePre ahead;
	ask(dAhead, blockIdx.x < numTiles, ahead);
	uint32_t dAhead2 = desc_of(blockIdx.x + gridDim.x);
	for (uint32_t tile = blockIdx.x; tile < numTiles; tile += gridDim.x) {
		__syncthreads();   // the previous tile's tables are no longer read
		const uint32_t d = dAhead;      // the descriptor, a word per lane
		const TilePre pre = ahead;
		dAhead = dAhead2;
		ask(dAhead, tile + gridDim.x < numTiles, ahead);
		dAhead2 = desc_of(tile + 2u*gridDim.x);
		const int dg2 = __builtin_amdgcn_readlane((int)d, 0), dg3 = __builtin_amdgcn_readlane((int)d, 1);
		const int ca = __builtin_amdgcn_readlane((int)d, 2), dnc = __builtin_amdgcn_readlane((int)d, 3);
		const uint32_t P = (uint32_t)(__builtin_amdgcn_readlane((int)d, 8) + __builtin_amdgcn_readlane((int)d, 9) +
			__builtin_amdgcn_readlane((int)d, 10) + __builtin_amdgcn_readlane((int)d, 11));        // home particles (<= TILE_PMAX)
		{	// a build in parts: the tile belongs to the part that holds its last home particle (its lists are all there then); every
			// thread of the workgroup sees the same descriptor, so all of them skip or none
			uint32_t homeEnd = 0u;
#pragma unroll
			for (int r = 0; r < TILE_HROWS; ++r) {
				const uint32_t first = (uint32_t)__builtin_amdgcn_readlane((int)d, 4 + r), cnt = (uint32_t)__builtin_amdgcn_readlane((int)d, 8 + r);
				if (cnt) homeEnd = max(homeEnd, first + cnt);
			}
			const uint32_t last = homeEnd ? homeEnd - 1u : 0u;
			if (last < homeFrom || last >= homeTo) continue;
		}
		const uint32_t C = (P + 63u)/64u;
		const uint32_t preCnt = pre.cnt, preHash = pre.hash;
		const particleinfo preInfo = pre.info;
		for (uint32_t e = tid; e < TILE_CHUNKS*TL_BINS; e += TL_THREADS) (&sHist[0][0])[e] = 0;
		if (tid < TILE_WROWS*TILE_KW) {
			uint32_t wStart = 0, wCnt = 0;      // window_cell, asked for a tile ahead
			if (pre.wStart != CELL_EMPTY) { wStart = pre.wStart; wCnt = pre.wEnd - pre.wStart; }
			uint32_t incl = wCnt;
			uint32_t lo = wCnt ? wStart : 0xFFFFFFFFu;
			uint32_t hi = wCnt ? wStart + wCnt : 0u;
#pragma unroll
			for (int dd = 1; dd < TILE_KW; dd <<= 1) {
				const uint32_t t = __shfl_up(incl, dd, TILE_KW);
				if (wcol >= dd) incl += t;
			}
#pragma unroll
			for (int dd = TILE_KW/2; dd > 0; dd >>= 1) {
				lo = min(lo, (uint32_t)__shfl_xor(lo, dd, TILE_KW));
				hi = max(hi, (uint32_t)__shfl_xor(hi, dd, TILE_KW));
			}
			sCellRel[tid] = incl - wCnt;
			sCellStart[tid] = wStart;
			// one DMA per row needs the cells to lie in memory in window order: each non-empty cell starts where
			// the previous ones end.  (Extent == count alone is not enough: a periodic row that is wholly inside
			// the window has the wrapped column first in the window but last in memory.)
			uint32_t inOrder = (wCnt == 0u || wStart - lo == incl - wCnt) ? 1u : 0u;
#pragma unroll
			for (int dd = TILE_KW/2; dd > 0; dd >>= 1)
				inOrder &= (uint32_t)__shfl_xor(inOrder, dd, TILE_KW);
			if (wcol == TILE_KW - 1) {
				sRowTotal[wr] = incl;
				sRowStart[wr] = incl ? lo : 0u;
				sRowContig[wr] = (incl == 0u || (hi - lo == incl && inOrder)) ? 1u : 0u;
			}
		}
		__syncthreads();
		if (tid < TILE_WROWS*TILE_KW) {
			uint32_t base = 0;
			for (int r = 0; r < wr; ++r) base += sRowTotal[r];
			sCellBase[tid] = sCellRel[tid] + base;
			if (wcol == 0) sRowBase[wr] = base;
		}
		__syncthreads();
		if (tid < TILE_WROWS) tileRows[(size_t)TILE_ROWDESC*tile + tid] = sRowStart[tid];
		else if (tid < TILE_WROWS + 8u) {
			const uint32_t k = tid - TILE_WROWS;
			tileRows[(size_t)TILE_ROWDESC*tile + 16u + k] = (sRowTotal[2*k] & 0xFFFFu) | (sRowTotal[2*k + 1] << 16);
		} else if (tid < TILE_WROWS + 16u) {
			const uint32_t k = tid - TILE_WROWS - 8u;
			const uint32_t b0 = (sRowBase[2*k] & 0x7FFFu) | (sRowContig[2*k] ? 0u : 0x8000u);
			const uint32_t b1 = (sRowBase[2*k + 1] & 0x7FFFu) | (sRowContig[2*k + 1] ? 0u : 0x8000u);
			tileRows[(size_t)TILE_ROWDESC*tile + 24u + k] = b0 | (b1 << 16);
		}
		for (uint32_t e = tid; e < TILE_HROWS*TILE_MAXCELLS*27; e += TL_THREADS) {
			const uint32_t m = e/27u, c1 = e - m*27u;
			const uint32_t hr = m/TILE_MAXCELLS, col = m - hr*TILE_MAXCELLS;
			sCB[e + 1] = (uint16_t)sCellBase[sCodeOff[c1] + (int)(((hr & 1u) + 4u*(hr >> 1))*TILE_KW + col)];
		}

		// ---- 2. sort the home particles by the length of their fluid section, longest first; equal lengths keep their home order
		bool overflow = false;
		uint32_t F = 0, B = 0;
		const bool inHome = tid < P;
		if (inHome) {
			const uint32_t cnt = preCnt;
			F = cnt & 0xFFFFu; B = cnt >> 16;
			// A list that overflowed while it was built (build_neibs_kernel: nf >= neibboundpos, nf + nb >= neibboundpos, with
			// SA_BOUNDARY nv >= neiblistsize - neibboundpos - 1) keeps COUNTING its entries but stops writing them: its counts
			// name slots of the other section, or slots that were never written.  Such a tiling goes to the generic kernel
			const bool unwritten = F >= p.neibboundpos || (saVertex ? B >= p.neiblistsize - p.neibboundpos - 1u : F + B >= p.neibboundpos);
			if (F > 128u || B > 128u || unwritten) { overflow = true; F = 0; B = 0; }
		}
		uint32_t rankInWave = 0;
		{
			unsigned long long rem = __builtin_amdgcn_ballot_w64(inHome);
			while (rem) {
				const int lead = __builtin_ctzll(rem);
				const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)F, lead);
				const unsigned long long m = __builtin_amdgcn_ballot_w64(inHome && F == v);
				if (inHome && F == v) rankInWave = (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull));
				if ((int)lane == lead) sHist[wave][v] = (uint16_t)__builtin_popcountll(m);
				rem &= ~m;
			}
		}
		__syncthreads();
		uint32_t binTot = 0, binSuffix = 0;      // threads 0..128: particles of this length; ... of this length and the longer ones of my wave's 64 lengths
		if (tid < 192u) {      // per length: particles in the waves before each wave, and in all
			if (tid < 129u)
				for (uint32_t w = 0; w < TILE_CHUNKS; ++w) { const uint32_t c = sHist[w][tid]; sHist[w][tid] = (uint16_t)binTot; binTot += c; }
			// longest first: a length starts behind all longer ones -- a suffix scan over the 129 lengths in three waves (a loop
			// over the longer lengths per thread was 128 dependent LDS reads for length 0: ~3 us of the ~8 a tile takes before
			// its lists are translated)
			binSuffix = binTot;
#pragma unroll
			for (int dd = 1; dd < 64; dd <<= 1) {
				const uint32_t t2 = (uint32_t)__shfl_down((int)binSuffix, dd);
				if (lane + (uint32_t)dd < 64u) binSuffix += t2;
			}
			if (lane == 0) sBinWave[wave] = binSuffix;
		}
		__syncthreads();
		if (tid < 129u) {
			uint32_t behind = 0;
			for (uint32_t w2 = wave + 1u; w2 < 3u; ++w2) behind += sBinWave[w2];
			sBinStart[tid] = (uint16_t)(binSuffix - binTot + behind);
		}
		__syncthreads();
		if (inHome) {
			const uint32_t pos = (uint32_t)sBinStart[F] + sHist[wave][F] + rankInWave;
			sLaneOf[tid] = (uint16_t)pos; sLenF[pos] = (uint16_t)F; sLenB[pos] = (uint16_t)B;
		}
		__syncthreads();
		// for a moment thread L is LANE L of the tile (lane L & 63 of chunk L >> 6 = this thread's wave)
		const uint32_t L = tid;
		const uint32_t myF = L < P ? sLenF[L] : 0u, myB = L < P ? sLenB[L] : 0u;
		{	// the chunk's sections are padded to its longest list, in whole batches
			uint32_t mxF = myF, mxB = myB;
#pragma unroll
			for (int dd = 32; dd > 0; dd >>= 1) {
				mxF = max(mxF, (uint32_t)__shfl_xor(mxF, dd)); mxB = max(mxB, (uint32_t)__shfl_xor(mxB, dd));
			}
			if (lane == 0) { sChunkF[wave] = (mxF + TILE_NB - 1u)/TILE_NB; sChunkB[wave] = (mxB + TILE_NB - 1u)/TILE_NB; }
		}
		__syncthreads();
		// ---- 3. + 4. the schedule, by the first wave.  Lane c < 10 stands for chunk c, lane 16 + w for wave w.  The batches of
		// the tile are numbered 0..T-1 (chunk after chunk); wave w walks [w share, (w+1) share).  A run starts wherever a chunk
		// or a wave's share starts: the run boundaries are the set S of those numbers, a run's number is its rank in S.
		if (wave == 0) {
			uint32_t nF = 0, nB = 0;
			if (lane < C) { nF = sChunkF[lane]; nB = sChunkB[lane]; }
			uint32_t incl = nF + nB;            // inclusive scan over the chunk lanes -> first batch of every chunk
#pragma unroll
			for (int dd = 1; dd < 16; dd <<= 1) {
				const uint32_t t = __shfl_up(incl, dd, 16);
				if ((lane & 15u) >= (uint32_t)dd) incl += t;
			}
			const uint32_t T = (uint32_t)__builtin_amdgcn_readlane((int)incl, 15);
			// wave w walks the batches [T c(w), T c(w+1)) / 256, c = the running sum of the waves' weights (tile_share_start)
			const uint32_t share = (T*max(shareLo, 64u - shareLo) + 255u) >> 8;
			const uint32_t cStart = incl - (nF + nB), cEnd = incl;
			const bool isChunk = lane < C && nF + nB > 0u;
			const bool isWave = lane >= 16u && lane < 16u + TILE_WAVES;
			const uint32_t wv = lane - 16u;
			// even shares: T/8 batches each, the T % 8 left over go one each to the first waves, i.e. to different SIMDs
			const uint32_t sbase = T/TILE_WAVES, srem = T - sbase*TILE_WAVES;
			const bool even = shareLo == 32u;
			const uint32_t ws = !isWave ? 0u : even ? wv*sbase + min(wv, srem) : (T*tile_share_start(wv, shareLo) + 128u) >> 8;
			const uint32_t we = !isWave ? 0u : even ? ws + sbase + (wv < srem ? 1u : 0u) : (T*tile_share_start(wv + 1u, shareLo) + 128u) >> 8;
			// my boundary point (chunk lanes: the chunk's first batch; wave lanes: the share's first batch) and the range I ask about
			const uint32_t lo = isChunk ? cStart : ws, hi = isChunk ? cEnd : we;
			// a wave's start that is also a chunk's start is one boundary, the chunk's
			bool dup = false;
			uint32_t cntLt = 0, cntIn = 0, cstar = 0;
			// chunk lanes first: which wave starts coincide with a chunk start
			for (int q = 0; q < (int)TILE_CHUNKS; ++q) {
				const bool qv = (__builtin_amdgcn_ballot_w64(isChunk) >> q) & 1ull;
				if (!qv) continue;
				const uint32_t x = (uint32_t)__builtin_amdgcn_readlane((int)cStart, q);
				if (isWave && x == ws) dup = true;
				if (x <= lo) cstar = (uint32_t)q;                    // the chunk my boundary point lies in (the last one starting at or before it)
				cntLt += (x < lo) ? 1u : 0u; cntIn += (x >= lo && x < hi) ? 1u : 0u;
			}
			const bool inS = isChunk || (isWave && ws < we && !dup);
			for (int q = 16; q < 16 + (int)TILE_WAVES; ++q) {
				const bool qv = (__builtin_amdgcn_ballot_w64(inS) >> q) & 1ull;
				if (!qv) continue;
				const uint32_t x = (uint32_t)__builtin_amdgcn_readlane((int)ws, q);
				cntLt += (x < lo) ? 1u : 0u; cntIn += (x >= lo && x < hi) ? 1u : 0u;
			}
			const uint32_t R = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(inS));   // runs of the tile
			if (inS) sSorted[cntLt] = lo;         // cntLt = my rank in S
			if (lane < TILE_RUNTAB) sRunTab[lane] = 0u;
			// (one wave: its LDS operations execute in order; the fences only keep the compiler from moving them)
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
			const uint32_t chunkSel = isChunk ? lane : cstar;
			const uint32_t chStart = __shfl(cStart, (int)chunkSel), chF = __shfl(nF, (int)chunkSel), chEnd = __shfl(cEnd, (int)chunkSel);
			if (inS) {         // my run: from my boundary point to the next one
				const uint32_t r = cntLt;
				const uint32_t end = (r + 1u < R) ? sSorted[r + 1u] : T;
				const uint32_t a0 = lo - chStart, len = end - lo;
				const uint32_t runF = (a0 < chF) ? min(chF - a0, len) : 0u;
				sRunTab[TILE_RT_RUN + min(r, (uint32_t)(TILE_RUNS_MAX - 1))] = chunkSel | (runF << 4) | ((len - runF) << 12) | (end == chEnd ? TILE_RUN_LAST : 0u);
			}
			// wave (lane - 16): first run (the one starting at its first batch: that batch is a boundary), the runs that start
			// inside its share, first batch, batches
			if (isWave) sRunTab[lane - 16u] = (ws < we) ? (cntLt | (cntIn << 5) | (ws << 10) | ((we - ws) << 22)) : 0u;
			if (lane < TILE_CHUNKS) {
				sRunTab[TILE_RT_CHUNK + lane] = isChunk ? (cntLt | (cntIn << 8)) : 0u;
				sChunkStart[lane] = cStart;
			}
			if (lane == 0) {
				sRunTab[8] = C | (P << 8) | (R << 24);
				// ---- 4. room in the list stream and in the lane tables
				const uint32_t lb = atomicAdd(tileCtl + 12, T), nb2 = atomicAdd(tileCtl + 13, C*64u);
				const bool ovf = (uint64_t)lb + T > listCapBatches || (uint64_t)nb2 + C*64u > laneCap || R > TILE_RUNS_MAX || T >= 4096u || share >= 1024u;
				sBase[0] = lb; sBase[1] = nb2; sBase[2] = ovf ? 1u : 0u;
			}
		}
		__syncthreads();
		if (sBase[2]) overflow = true;
		const uint32_t listBase = sBase[0], laneBase = sBase[1];
		if (!sBase[2]) {
			if (tid == 0) { tiles[(size_t)TILE_DESC*tile + 14] = listBase; tiles[(size_t)TILE_DESC*tile + 15] = laneBase; }
			if (tid < TILE_RUNTAB) tileRuns[(size_t)TILE_RUNTAB*tile + tid] = sRunTab[tid];
		}
		// ---- 5. the lane tables and the translated lists.  Back in home order: thread t stands for home particle t (consecutive
		// threads read consecutive columns of the neighbour list: whole lines; in lane order a wave gathered two-byte entries
		// from all over the tile) and, from P on, for the idle lanes of the last chunk.  The eight-byte stores go to the lane's
		// place in its chunk's batches (scattered, but a quarter as many as the loads)
		const bool isHome = tid < P, isLane = tid < C*64u;
		const uint32_t myLane = isHome ? (uint32_t)sLaneOf[tid] : tid;
		const uint32_t ch = min(myLane >> 6, (uint32_t)(TILE_CHUNKS - 1)), ln = myLane & 63u;
		const TileHome h = tile_home(d, isHome ? tid : 0u);
		const uint32_t index = h.index;
		const int3 gp = grid_pos_from_hash(p, preHash & CELLTYPE_BITMASK);
		const int myG1 = (p.c1 == 0) ? gp.x : (p.c1 == 1) ? gp.y : gp.z;
		const int myCol = min(max(myG1 - ca, 0), TILE_MAXCELLS - 1);
		const uint16_t *myCB = sCB + (h.hrow*TILE_MAXCELLS + myCol)*27;
		if (!sBase[2]) {
			if (isLane) {
				uint32_t rec = 0u, idx = 0xFFFFFFFFu;
				if (isHome) {   // the particle's own row in the window (the forces kernel reads its position, velocity and EOS row there)
					const int wc = (5 + (h.hrow & 1) + 4*(h.hrow >> 1))*TILE_KW + myCol + 1;
					const uint32_t slot = 1u + sCellBase[wc] + (index - sCellStart[wc]);
					if (slot > 4095u) overflow = true;
					const particleinfo pi = preInfo;
					const uint32_t flags = PART_TYPE(pi) | (HAS_COMPUTE_FORCE(pi) ? LANE_COMPUTE_FORCE : 0u) | ((FLUID_NUM(pi) & 3u) << LANE_FLUID_SHIFT) | LANE_VALID;
					rec = ((slot << 4) & 0xFFFFu) | (flags << 16);
					idx = index;
				}
				laneRec[laneBase + myLane] = rec; laneIndex[laneBase + myLane] = idx;
			}
			const uint32_t chunkStart = sChunkStart[ch], chF = sChunkF[ch], chB = sChunkB[ch];
#pragma unroll 1
			for (int sec = 0; sec < 2; ++sec) {
				// section 0: slots 0 upward, section 1: slots neibboundpos downward (SA_BOUNDARY: the vertex section, slots
				// neibboundpos + 1 upward; the boundary elements are not particles of a window).  Every lane of a chunk writes all
				// the chunk's batches of the section, padded with the dummy row's offset
				const uint32_t nbat = isLane ? (sec ? chB : chF) : 0u;
				const uint32_t cnt = isHome ? (sec ? B : F) : 0u;
				uint2 *out = tileList + ((size_t)listBase + chunkStart + (sec ? chF : 0u))*64u + ln;
				uint32_t nbatWave = nbat;      // the wave walks to the longest of its lanes' chunks
#pragma unroll
				for (int dd = 32; dd > 0; dd >>= 1) nbatWave = max(nbatWave, (uint32_t)__shfl_xor(nbatWave, dd));
				uint32_t code = 0;
				// entry sl of the section: slot sl, or neibboundpos - sl, or neibboundpos + 1 + sl -- a base and a signed step, both
				// the same for every lane; the lane's part of the address is its particle's index alone, ONE 32-bit offset for all
				// the loads of the loop (with an address pair per load the compiler recycled the pairs as load destinations and
				// waited for the loads in flight before each re-use)
				const neibdata *const secList = list + (size_t)(!sec ? 0u : saVertex ? p.neibboundpos + 1u : p.neibboundpos)*stride;
				const long long secStep = (sec && !saVertex) ? -(long long)stride : (long long)stride;
				const uint32_t secCap = !sec ? p.neiblistsize : saVertex ? p.neiblistsize - p.neibboundpos - 1u : p.neibboundpos + 1u;
				constexpr int LOADS = TL_LOADS;   // entries per lane in flight: the walk is latency bound
#pragma unroll 1
				for (uint32_t b0 = 0; b0 < nbatWave; b0 += LOADS/TILE_NB) {
					uint32_t e[LOADS];
#pragma unroll
					for (int k = 0; k < LOADS; ++k) {
						const uint32_t sl = min(b0*TILE_NB + (uint32_t)k, secCap - 1u);   // the same slot for every lane; past a lane's list the entry is not used
						// (unconditional: the address is valid for every thread, and a load under a branch of its own made the compiler
						// wait for all the earlier ones before it in some builds -- 7 to 13 ms per launch depending on unrelated edits)
						e[k] = (uint32_t)(secList + (long long)sl*secStep)[index];
					}
					uint32_t val[LOADS];
#pragma unroll
					for (int k = 0; k < LOADS; ++k) {
						const uint32_t dd = e[k];
						const bool live = b0*TILE_NB + (uint32_t)k < cnt;
						// (the list is build_neibs_kernel's own, written moments ago, and its section lengths were checked against the
						// list's geometry above: the entries are not validated one by one -- that was a third of this loop's
						// instructions -- only that the section opened with a cell code, below)
						code = (live && dd >= CELLNUM_ENCODED) ? (dd >> CELLNUM_SHIFT) : code;
						const uint32_t slot = 1u + (uint32_t)myCB[min(code, 27u)] + (dd & NEIBINDEX_MASK);   // slot 0 = dummy
						if (live && slot > 4095u) overflow = true;
						val[k] = live ? ((slot << 4) & 0xFFFFu) : 0u;
					}
#pragma unroll
					for (int k = 0; k < LOADS/TILE_NB; ++k)
						if (b0 + (uint32_t)k < nbat)
							out[(size_t)(b0 + (uint32_t)k)*64u] = make_uint2(val[4*k] | (val[4*k + 1] << 16), val[4*k + 2] | (val[4*k + 3] << 16));
				}
				if (cnt && code == 0u) overflow = true;      // a section that never named a cell: not a list of the builder
			}
		}
		if (overflow) tileCtl[1] = 1u;      // generic kernel
	}
}

int sphx_tile_lists_launch(sphx_ctx *ctx, const uint16_t *neibsList, const void *info, const uint32_t *hash, const uint32_t *cellStart, bool sa, hipStream_t st,
	uint32_t homeFrom, uint32_t homeTo)
{
	if (!ctx->tile_list || !ctx->tile_runs || !ctx->tile_rows || !ctx->tile_lane_rec || !ctx->tile_lane_index || !ctx->neib_counts) {
		ctx->tiles_built = false;
		return SPHX_OK;
	}
	const uint32_t grid = ctx->tile_grid*TL_GRIDMUL < ctx->tile_capacity ? ctx->tile_grid*TL_GRIDMUL : ctx->tile_capacity;
	// the plain forces pass is the only walker of these lists that gains from uneven shares (tile_share_start)
	const bool plain = !sa && ctx->dev.turbmodel == SPHX_ARTIFICIAL && ctx->dev.numfluids == 1 && ctx->dev.boundarytype == SPHX_DYN_BOUNDARY;
	tile_lists_kernel<<<grid, TL_THREADS, 0, st>>>(ctx->dev, neibsList, ctx->neib_counts, (const particleinfo*)info, hash, cellStart, ctx->cell_end_copy,
		ctx->tiles, ctx->tile_ctl, ctx->tile_rows, ctx->tile_runs, ctx->tile_list, ctx->tile_list_batches, ctx->tile_lane_rec, ctx->tile_lane_index,
		ctx->tile_lane_cap, sa ? 1 : 0, plain ? TILE_SHARE_LO_PLAIN : 32u, homeFrom, homeTo);
	SPHX_LAUNCH_CHECK("tile_lists_kernel");
	return SPHX_OK;
}

extern "C" int sphx_forces_basicstep(sphx_ctx *ctx,
	void *forces, float *cfl, void *rbforces, void *rbtorques,
	const void *pos, const void *vel, const void *info, const uint32_t *hash,
	const uint32_t *cellStart, const uint16_t *neibsList,
	const void *tau0, const void *tau1, const void *tau2, void *xsph,
	uint32_t numParticles, uint32_t fromParticle, uint32_t toParticle,
	float deltap, float slength, float dtadaptfactor, float influenceradius,
	uint32_t cflOffset, int run_mode, int step, float dt, int compute_object_forces,
	uint32_t *h_numBlocks, void *stream)
{
	(void)dtadaptfactor; (void)step; (void)dt;
	SPHX_REQUIRE(ctx && ctx->have_params, "sphx_forces_basicstep: constants not set");
	SPHX_REQUIRE(forces && pos && vel && info && hash && cellStart && neibsList, "sphx_forces_basicstep: missing buffer");
	SPHX_REQUIRE(fromParticle <= toParticle && toParticle <= numParticles, "sphx_forces_basicstep: invalid particle range");
	SPHX_REQUIRE(slength == ctx->params.slength && influenceradius == ctx->params.influenceradius,
		"sphx_forces_basicstep: slength/influenceradius differ from set_constants");
	if (run_mode != SPHX_SIMULATE && run_mode != SPHX_REPACK)
		return sphx_set_error(SPHX_ERR_INVALID, "sphx_forces_basicstep: invalid run mode");
	if (ctx->dev.boundarytype != SPHX_DYN_BOUNDARY && ctx->dev.boundarytype != SPHX_LJ_BOUNDARY)
		return sphx_set_error(SPHX_ERR_UNSUPPORTED, "sphx_forces_basicstep: only DYN_BOUNDARY and LJ_BOUNDARY pair interactions are built");
	if (ctx->params.rheologytype > SPHX_NEWTONIAN && run_mode == SPHX_SIMULATE)
		return sphx_set_error(SPHX_ERR_INVALID, "sphx_forces_basicstep: generalized Newtonian rheologies read BUFFER_EFFVISC, use sphx_forces_basicstep_effvisc");
	if ((ctx->params.sph_formulation == SPHX_SPH_HA || (ctx->params.rheologytype == SPHX_NEWTONIAN && ctx->params.viscmodel != SPHX_MORRIS)) &&
		run_mode == SPHX_SIMULATE) {      // Hu & Adams, MONAGHAN / ESPANOL_REVENGA viscous models: rheology.hip
		if (compute_object_forces || rbforces)
			return sphx_set_error(SPHX_ERR_UNSUPPORTED, "sphx_forces_basicstep: SPH_HA / MONAGHAN / ESPANOL_REVENGA with bodies that feel the fluid are not built");
		return sphx_fidelity_forces_launch(ctx, forces, cfl, pos, vel, info, hash, cellStart, neibsList, nullptr, numParticles, fromParticle,
			toParticle, slength, influenceradius, cflOffset, h_numBlocks, stream);
	}
	if (ctx->params.sph_formulation == SPHX_SPH_GRENIER && run_mode == SPHX_SIMULATE)
		return sphx_set_error(SPHX_ERR_INVALID, "sphx_forces_basicstep: SPH_GRENIER reads BUFFER_SIGMA, use sphx_forces_basicstep_grenier");
	if (ctx->dev.turbmodel == SPHX_SPS && run_mode == SPHX_SIMULATE)
		SPHX_REQUIRE(tau0 && tau1 && tau2, "sphx_forces_basicstep: SPS needs the three TAU arrays");
	if ((ctx->dev.simflags & SPHX_ENABLE_DTADAPT))
		SPHX_REQUIRE(cfl != nullptr, "sphx_forces_basicstep: ENABLE_DTADAPT needs the CFL buffer");
	SPHX_REQUIRE((rbforces == nullptr) == (rbtorques == nullptr), "sphx_forces_basicstep: RB_FORCES and RB_TORQUES must come together");
	if ((ctx->dev.simflags & SPHX_ENABLE_XSPH) && run_mode == SPHX_SIMULATE)
		SPHX_REQUIRE(xsph != nullptr, "sphx_forces_basicstep: ENABLE_XSPH needs the XSPH buffer");

	const uint32_t nrange = toParticle - fromParticle;
	const uint32_t numBlocks = round_up_u(div_up_u(nrange, SPHX_BLOCK_FORCES), 4u);
	if (h_numBlocks) *h_numBlocks = numBlocks;
	if (!numBlocks) return SPHX_OK;
	{ const int rcf = sphx_rb_flush(ctx, (hipStream_t)stream); if (rcf != SPHX_OK) return rcf; }
	if (run_mode == SPHX_REPACK)   // run_repack, src/cuda/forces.cu:828-896 (filters.hip)
		return sphx_repack_launch(ctx, forces, cfl, rbforces, rbtorques, pos, vel, info, hash, cellStart, neibsList,
			fromParticle, toParticle, cflOffset, numBlocks, deltap, (hipStream_t)stream);

	{	// EOS pre-pass over ALL particles: neighbours may lie outside [fromParticle,toParticle)
		int rc0 = sphx_ensure_scratch(ctx, numParticles);
		if (rc0 != SPHX_OK) return rc0;
		// ... unless the Euler step that wrote these densities left the rows behind and the caller vouches for the buffer
		// (sphx_eos_rows_current, euler.hip)
		const bool current = ctx->eos_armed && ctx->eos_tag_vel == vel && ctx->eos_tag_n == numParticles;
		ctx->eos_armed = false;
		if (!current) {
			// the rows are this buffer's from here on (a second stripe of the same pass may be vouched for as well)
			ctx->eos_tag_vel = ctx->eos_follow ? vel : nullptr;
			ctx->eos_tag_n = numParticles;
			eos_kernel<<<div_up_u(numParticles, 256), 256, 0, (hipStream_t)stream>>>(ctx->dev, (const float4*)vel,
				(const particleinfo*)info, ctx->eos_aux, numParticles);
			SPHX_LAUNCH_CHECK("eos_kernel");
		}
	}
	ForcesArgs a;
	a.forces = (float4*)forces; a.cfl = (ctx->dev.simflags & SPHX_ENABLE_DTADAPT) ? cfl : nullptr;
	a.rbforces = (float4*)rbforces; a.rbtorques = (float4*)rbtorques;
	a.pos = (const float4*)pos; a.vel = (const float4*)vel; a.info = (const particleinfo*)info;
	a.hash = hash; a.cellStart = cellStart; a.neibsList = neibsList;
	a.tau0 = (const float2*)tau0; a.tau1 = (const float2*)tau1; a.tau2 = (const float2*)tau2;
	a.rb = ctx->rb_dev;
	a.aux = ctx->eos_aux;
	a.tileList = ctx->tile_list; a.tileRuns = ctx->tile_runs; a.tileRows = ctx->tile_rows;
	a.tileLaneRec = ctx->tile_lane_rec; a.tileLaneIndex = ctx->tile_lane_index;
	a.xsph = (float4*)xsph;
	a.fromParticle = fromParticle; a.toParticle = toParticle; a.cflOffset = cflOffset;
	a.wholeRange = (fromParticle == 0u && toParticle == numParticles) ? 1 : 0;
	a.numBlocks = numBlocks;
	a.compute_object_forces = compute_object_forces;
	a.pin = nullptr;
#ifdef SPHX_TILE_DEBUG_BUILD
	a.prof = nullptr;
	if (ctx->tile_debug & 16) {
		if (!ctx->tile_prof && hipMalloc((void**)&ctx->tile_prof, 10*(TILE_THREADS/64)*sizeof(unsigned long long)*ctx->tile_grid) != hipSuccess)
			return sphx_set_error(SPHX_ERR_RUNTIME, "sphx_forces_basicstep: cannot allocate the tile profile buffer");
		a.prof = ctx->tile_prof;
	}
#endif

	sphx_tiles_overflow_poll(ctx);
	// the tiling belongs to the neighbour list built last by this context from these very buffers
	const bool use_tiles = ctx->tiles_built && ctx->tiles_overflow != 1 && ctx->tiles_cellstart == cellStart && ctx->tiles_neibslist == neibsList &&
		((ctx->dev.numfluids == 1 && ctx->dev.densitydiff != SPHX_FERRARI) || ctx->dev.kerneltype == SPHX_WENDLAND) &&
		ctx->dev.formulation == SPHX_SPH_F1 && !ctx->disable_tiles && ctx->tile_list != nullptr;
	a.tauPack = nullptr; a.tauPackN = 0;
	a.otau0 = a.otau1 = a.otau2 = nullptr; a.oturbvisc = nullptr;
	if (use_tiles && ctx->dev.turbmodel == SPHX_SPS) {   // window rows of the stress tensor (see tau_pack_kernel)
		SPHX_REQUIRE(ctx->tau_pack != nullptr && numParticles <= ctx->reserved_particles, "sphx_forces_basicstep: SPS scratch not reserved");
		tau_pack_kernel<<<div_up_u(numParticles, 256), 256, 0, (hipStream_t)stream>>>((const float2*)tau0, (const float2*)tau1,
			(const float2*)tau2, ctx->eos_aux, ctx->dev.densitydiff == SPHX_COLAGROSSI ? 1 : 0, ctx->tau_pack, numParticles);
		SPHX_LAUNCH_CHECK("tau_pack_kernel");
		a.tauPack = ctx->tau_pack; a.tauPackN = numParticles;
	}
	int rc;
	switch (ctx->dev.kerneltype) {
	case SPHX_CUBICSPLINE: rc = sphx_part_forces_k1(ctx, dim3(numBlocks), (hipStream_t)stream, a, use_tiles); break;
	case SPHX_QUADRATIC:   rc = sphx_part_forces_k2(ctx, dim3(numBlocks), (hipStream_t)stream, a, use_tiles); break;
	case SPHX_WENDLAND:    rc = sphx_part_forces_k3(ctx, dim3(numBlocks), (hipStream_t)stream, a, use_tiles); break;
	case SPHX_GAUSSIAN:    rc = sphx_part_forces_k4(ctx, dim3(numBlocks), (hipStream_t)stream, a, use_tiles); break;
	default: return sphx_set_error(SPHX_ERR_INVALID, "sphx_forces_basicstep: invalid kernel type");
	}
	if (rc != SPHX_OK) return rc;
	SPHX_LAUNCH_CHECK("forces_kernel");
	if (ctx->dev.simflags & SPHX_ENABLE_XSPH)   // mean neighbourhood velocity of the fluid particles (filters.hip)
		return sphx_xsph_launch(ctx, xsph, pos, vel, info, hash, cellStart, neibsList, fromParticle, toParticle, (hipStream_t)stream);
	return SPHX_OK;
}

// displacement of every particle over the step (cell-local positions of the same cell: the hash only changes at the next sort)
static __global__ void __launch_bounds__(256)
sa_displacement_kernel(const float4 *__restrict__ oldPos, const float4 *__restrict__ newPos, float4 *__restrict__ disp, uint32_t n)
{
	const uint32_t i = blockIdx.x*256 + threadIdx.x;
	if (i >= n) return;
	const float4 a = oldPos[i], b = newPos[i];
	disp[i] = make_float4(b.x - a.x, b.y - a.y, b.z - a.z, 0.0f);
}

// The particle <- particle sums of an SA_BOUNDARY engine over the tiles of the current neighbour list (see SPHX_TURB_SA).
// *used = the tiled kernel was launched: the caller's list-walker kernel then only finishes the particles (boundary elements,
// division by gamma, ...) unless the device-side overflow flag *guard says the tiles could not be used after all.
int sphx_sa_tiles_run(sphx_ctx *ctx, int mode, void *forces, const void *pos, const void *vel, const void *newPos,
	const void *info, const uint32_t *hash, const uint32_t *cellStart, const uint16_t *neibsList, const void *gGam,
	uint32_t numParticles, uint32_t fromParticle, uint32_t toParticle, float dt, hipStream_t stream,
	bool *used, const uint32_t **guard)
{
	*used = false; *guard = nullptr;
	sphx_tiles_overflow_poll(ctx);
	const DevParams &d = ctx->dev;
	const bool ok = ctx->tiles_built && ctx->tiles_overflow != 1 && ctx->tiles_cellstart == cellStart && ctx->tiles_neibslist == neibsList &&
		!ctx->disable_tiles && ctx->tile_list != nullptr && d.boundarytype == SPHX_SA_BOUNDARY && d.kerneltype == SPHX_WENDLAND &&
		d.numfluids == 1 && (d.turbmodel == SPHX_LAMINAR_FLOW || (d.turbmodel == SPHX_KEPSILON && mode != SPHX_SA_TILE_FORCES)) &&
		d.formulation == SPHX_SPH_F1 && d.rheology <= SPHX_NEWTONIAN &&
		numParticles <= ctx->reserved_particles;
	if (!ok || fromParticle >= toParticle) return SPHX_OK;
	ForcesArgs a = {};
	a.forces = (float4*)forces;
	a.pos = (const float4*)pos; a.vel = (const float4*)vel; a.info = (const particleinfo*)info;
	a.hash = hash; a.cellStart = cellStart; a.neibsList = neibsList;
	a.aux = ctx->eos_aux;
	ctx->eos_tag_vel = nullptr;      // the scratch rows change hands
	if (mode == SPHX_SA_TILE_DSUM) {      // the window's second array holds the displacements
		sa_displacement_kernel<<<div_up_u(numParticles, 256), 256, 0, stream>>>((const float4*)pos, (const float4*)newPos, ctx->eos_aux, numParticles);
		SPHX_LAUNCH_CHECK("sa_displacement_kernel");
		a.vel = ctx->eos_aux;
	} else {
		eos_kernel<<<div_up_u(numParticles, 256), 256, 0, stream>>>(ctx->dev, (const float4*)vel, (const particleinfo*)info, ctx->eos_aux, numParticles);
		SPHX_LAUNCH_CHECK("eos_kernel");
	}
	a.tileList = ctx->tile_list; a.tileRuns = ctx->tile_runs; a.tileRows = ctx->tile_rows;
	a.tileLaneRec = ctx->tile_lane_rec; a.tileLaneIndex = ctx->tile_lane_index;
	a.saGam = (const float4*)gGam; a.saDt = dt;
	a.rb = ctx->rb_dev;
	a.fromParticle = fromParticle; a.toParticle = toParticle;
	a.wholeRange = (fromParticle == 0u && toParticle == numParticles) ? 1 : 0;
	sphx_part_sa_tile(ctx, stream, a, mode, d.rheology == SPHX_NEWTONIAN);
	SPHX_LAUNCH_CHECK("forces_tile_kernel (SA_BOUNDARY)");
	*used = true;
	*guard = ctx->tiles_overflow == 0 ? nullptr : ctx->tile_ctl + 1;     // NULL: the host has seen the tiling succeed
	return SPHX_OK;
}

extern "C" int sphx_forces_reserve_cus(sphx_ctx *ctx, uint32_t cus)
{
	SPHX_REQUIRE(ctx != nullptr, "sphx_forces_reserve_cus: NULL ctx");
	int total = 0;
	SPHX_HIP(hipDeviceGetAttribute(&total, hipDeviceAttributeMultiprocessorCount, ctx->device));
	const uint32_t all = (uint32_t)(total > 0 ? total : 256);
	const uint32_t k = (cus + 7u)/8u*8u;
	SPHX_REQUIRE(k < all, "sphx_forces_reserve_cus: more CUs reserved than the device has");
	ctx->tile_grid = (all - k)*TILE_WGS_PER_CU;
	return SPHX_OK;
}

static int dtreduce_launch(sphx_ctx *ctx, float slength, float dtadaptfactor, float sspeed_cfl,
	float max_kinematic, const float *cfl, float *cflTemp, uint32_t numBlocks,
	float *d_dt, int combine_min, hipStream_t stream)
{
	SPHX_REQUIRE(ctx && ctx->have_params, "sphx_forces_dtreduce: constants not set");
	SPHX_REQUIRE(cfl && cflTemp && d_dt, "sphx_forces_dtreduce: missing buffer (CFL, CFL_TEMP)");
	SPHX_REQUIRE((numBlocks & 3u) == 0 && numBlocks > 0, "sphx_forces_dtreduce: number of elements to reduce is not a multiple of 4");
	const uint32_t nPartials = sphx_forces_fmax_temp_elements(numBlocks);
	fmax_kernel<<<nPartials, BLOCK_FMAX, 0, stream>>>(cflTemp, (const float4*)cfl, numBlocks/4);
	const int viscous = (ctx->dev.rheology != SPHX_INVISCID || ctx->dev.turbmodel > SPHX_ARTIFICIAL) ? 1 : 0;
	dt_final_kernel<<<1, BLOCK_FMAX, 0, stream>>>(d_dt, cflTemp, nPartials, slength, dtadaptfactor, sspeed_cfl,
		max_kinematic, viscous, combine_min);
	SPHX_LAUNCH_CHECK("fmax_kernel/dt_final_kernel");
	return SPHX_OK;
}

extern "C" int sphx_forces_dtreduce_device(sphx_ctx *ctx, float slength, float dtadaptfactor, float sspeed_cfl,
	float max_kinematic, const float *cfl, float *cflTemp, uint32_t numBlocks,
	float *d_dt, int combine_min, void *stream)
{
	return dtreduce_launch(ctx, slength, dtadaptfactor, sspeed_cfl, max_kinematic, cfl, cflTemp, numBlocks,
		d_dt, combine_min, (hipStream_t)stream);
}

extern "C" int sphx_forces_dtreduce(sphx_ctx *ctx, float slength, float dtadaptfactor, float sspeed_cfl,
	float max_kinematic, const float *cfl, float *cflTemp, uint32_t numBlocks,
	float *h_dt, void *stream)
{
	SPHX_REQUIRE(h_dt != nullptr, "sphx_forces_dtreduce: h_dt is NULL");
	int rc = dtreduce_launch(ctx, slength, dtadaptfactor, sspeed_cfl, max_kinematic, cfl, cflTemp, numBlocks,
		ctx ? ctx->dt_scratch : nullptr, 0, (hipStream_t)stream);
	if (rc != SPHX_OK) return rc;
	SPHX_HIP(hipMemcpyAsync(h_dt, ctx->dt_scratch, sizeof(float), hipMemcpyDeviceToHost, (hipStream_t)stream));
	SPHX_HIP(hipStreamSynchronize((hipStream_t)stream));
	return SPHX_OK;
}

extern "C" int sphx_reduce_rb_forces(sphx_ctx *ctx, void *rbforces, void *rbtorques, const uint32_t *rbnum,
	const uint32_t *h_lastindex, float *h_totalforce3, float *h_totaltorque3,
	uint32_t numforcesbodies, uint32_t numForcesBodiesParticles, void *stream_)
{
	SPHX_REQUIRE(ctx && rbforces && rbtorques && rbnum && h_lastindex && h_totalforce3 && h_totaltorque3,
		"sphx_reduce_rb_forces: NULL argument");
	SPHX_REQUIRE(numforcesbodies <= SPHX_MAX_BODIES, "sphx_reduce_rb_forces: too many bodies");
	if (!numforcesbodies) return SPHX_OK;
	hipStream_t stream = (hipStream_t)stream_;
	uint32_t *d_last = nullptr; float *d_tot = nullptr;
	SPHX_HIP(hipMalloc((void**)&d_last, sizeof(uint32_t)*numforcesbodies));
	SPHX_HIP(hipMalloc((void**)&d_tot, sizeof(float)*6*numforcesbodies));
	SPHX_HIP(hipMemcpyAsync(d_last, h_lastindex, sizeof(uint32_t)*numforcesbodies, hipMemcpyHostToDevice, stream));
	rb_reduce_kernel<<<numforcesbodies, 256, 0, stream>>>((const float4*)rbforces, (const float4*)rbtorques,
		rbnum, d_last, d_tot, numForcesBodiesParticles);
	float tot[6*SPHX_MAX_BODIES];
	SPHX_HIP(hipMemcpyAsync(tot, d_tot, sizeof(float)*6*numforcesbodies, hipMemcpyDeviceToHost, stream));
	SPHX_HIP(hipStreamSynchronize(stream));
	(void)hipFree(d_last); (void)hipFree(d_tot);
	for (uint32_t b = 0; b < numforcesbodies; ++b)
		for (int c = 0; c < 3; ++c) {
			h_totalforce3[3*b + c] = tot[6*b + c];
			h_totaltorque3[3*b + c] = tot[6*b + 3 + c];
		}
	return SPHX_OK;
}

extern "C" int sphx_calc_visc(sphx_ctx *ctx, void *tau0, void *tau1, void *tau2, float *spsturbvisc,
	const void *pos, const void *vel, const void *info, const uint32_t *hash,
	const uint32_t *cellStart, const uint16_t *neibsList,
	uint32_t numParticles, uint32_t particleRangeEnd,
	float deltap, float slength, float influenceradius, void *stream)
{
	(void)deltap; (void)numParticles;
	SPHX_REQUIRE(ctx && ctx->have_params, "sphx_calc_visc: constants not set");
	if (ctx->dev.turbmodel != SPHX_SPS)
		return sphx_set_error(SPHX_ERR_UNSUPPORTED, "sphx_calc_visc: only the SPS branch is built");
	SPHX_REQUIRE(pos && vel && info && hash && cellStart && neibsList, "sphx_calc_visc: missing buffer");
	SPHX_REQUIRE((tau0 && tau1 && tau2) || (!tau0 && !tau1 && !tau2), "sphx_calc_visc: the three TAU arrays must come together");
	SPHX_REQUIRE(slength == ctx->params.slength && influenceradius == ctx->params.influenceradius,
		"sphx_calc_visc: slength/influenceradius differ from set_constants");
	if (!particleRangeEnd) return SPHX_OK;
	SpsArgs a;
	a.tau0 = (float2*)tau0; a.tau1 = (float2*)tau1; a.tau2 = (float2*)tau2; a.turbvisc = spsturbvisc;
	a.pos = (const float4*)pos; a.vel = (const float4*)vel; a.info = (const particleinfo*)info;
	a.hash = hash; a.cellStart = cellStart; a.neibsList = neibsList; a.numParticles = particleRangeEnd;
	dim3 grid(div_up_u(particleRangeEnd, 128));
	sphx_tiles_overflow_poll(ctx);
	// single fluid with the tiling of this neighbour list at hand: the stress mode of the tiled kernel (neighbour rows from
	// the LDS window instead of gathers), then the gather kernel as a stand-by guarded by the tiling's overflow flag
	const bool use_tiles = ctx->tiles_built && ctx->tiles_overflow != 1 && ctx->tiles_cellstart == cellStart && ctx->tiles_neibslist == neibsList &&
		ctx->dev.numfluids == 1 && !ctx->disable_tiles && ctx->tile_list != nullptr;
	const uint32_t *guard = nullptr;
	if (use_tiles) {
		ForcesArgs fa = ForcesArgs();
		fa.pos = a.pos; fa.vel = a.vel; fa.info = a.info; fa.hash = hash; fa.cellStart = cellStart; fa.neibsList = neibsList;
		fa.otau0 = a.tau0; fa.otau1 = a.tau1; fa.otau2 = a.tau2; fa.oturbvisc = spsturbvisc;
		fa.fromParticle = 0; fa.toParticle = particleRangeEnd;
		fa.tileList = ctx->tile_list; fa.tileRuns = ctx->tile_runs; fa.tileRows = ctx->tile_rows;
		fa.tileLaneRec = ctx->tile_lane_rec; fa.tileLaneIndex = ctx->tile_lane_index;
		switch (ctx->dev.kerneltype) {
		case SPHX_CUBICSPLINE: sphx_part_stress_k1(ctx, (hipStream_t)stream, fa); break;
		case SPHX_QUADRATIC:   sphx_part_stress_k2(ctx, (hipStream_t)stream, fa); break;
		case SPHX_WENDLAND:    sphx_part_stress_k3(ctx, (hipStream_t)stream, fa); break;
		default:               sphx_part_stress_k4(ctx, (hipStream_t)stream, fa); break;
		}
		SPHX_LAUNCH_CHECK("forces_tile_kernel (SPS stress)");
		guard = ctx->tile_ctl + 1;
		grid.x = grid.x < 2048u ? grid.x : 2048u;   // stand-by launch: every block returns at once unless the tiling overflowed
		if (ctx->tiles_overflow == 0) return SPHX_OK;   // the host saw the tiling succeed: no stand-by
	}
	switch (ctx->dev.kerneltype) {
	case SPHX_CUBICSPLINE: sphx_part_sps_k1(ctx, grid, (hipStream_t)stream, a, guard); break;
	case SPHX_QUADRATIC:   sphx_part_sps_k2(ctx, grid, (hipStream_t)stream, a, guard); break;
	case SPHX_WENDLAND:    sphx_part_sps_k3(ctx, grid, (hipStream_t)stream, a, guard); break;
	default:               sphx_part_sps_k4(ctx, grid, (hipStream_t)stream, a, guard); break;
	}
	SPHX_LAUNCH_CHECK("sps_kernel");
	return SPHX_OK;
}

#ifdef SPHX_TILE_DEBUG_BUILD
// timing experiments only (library built with -DSPHX_TILE_DEBUG_BUILD, SPHX_TILE_DEBUG=16): cycles per phase of every wave of
// the last tiled forces launch
extern "C" int sphx_dbg_tile_profile(sphx_ctx *ctx, unsigned long long *host, uint32_t maxGroups)
{
	if (!ctx || !ctx->tile_prof) return -1;
	const uint32_t n = maxGroups < ctx->tile_grid ? maxGroups : ctx->tile_grid;
	if (hipMemcpy(host, ctx->tile_prof, 10*(TILE_THREADS/64)*sizeof(unsigned long long)*n, hipMemcpyDeviceToHost) != hipSuccess) return -1;
	return (int)n;
}
#endif

// timing experiments only, not part of include/sphx.h: the tile descriptors of the last neighbour-list build
extern "C" int sphx_dbg_tiles(sphx_ctx *ctx, uint32_t *host, uint32_t maxTiles)
{
	if (!ctx || !ctx->tiles || !ctx->tiles_built) return -1;
	uint32_t ctl[4];
	if (hipMemcpy(ctl, ctx->tile_ctl, sizeof(ctl), hipMemcpyDeviceToHost) != hipSuccess) return -1;
	const uint32_t n = ctl[0] < maxTiles ? ctl[0] : maxTiles;
	if (hipMemcpy(host, ctx->tiles, (size_t)TILE_DESC*sizeof(uint32_t)*n, hipMemcpyDeviceToHost) != hipSuccess) return -1;
	return (int)n;
}

// diagnostics only, not part of include/sphx.h: raw copies of the tiling's tables (0 descriptors, 1 run tables, 2 lane records,
// 3 lane particles, 4 the list stream, 5 tile_ctl, 6 window rows), `bytes` from the start; synchronises
extern "C" int sphx_dbg_tile_table(sphx_ctx *ctx, int which, void *host, size_t bytes)
{
	if (!ctx || !host) return -1;
	const void *src = which == 0 ? (const void*)ctx->tiles : which == 1 ? (const void*)ctx->tile_runs : which == 2 ? (const void*)ctx->tile_lane_rec :
		which == 3 ? (const void*)ctx->tile_lane_index : which == 4 ? (const void*)ctx->tile_list : which == 5 ? (const void*)ctx->tile_ctl :
		which == 6 ? (const void*)ctx->tile_rows : nullptr;
	if (!src) return -1;
	return hipMemcpy(host, src, bytes, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}

// tests only, not part of include/sphx.h: 1 when the last neighbour-list build left a usable tiling (built, lists allocated, no
// overflow), i.e. the tiled kernels are the ones that run; synchronises
extern "C" int sphx_dbg_tiles_usable(sphx_ctx *ctx)
{
	if (!ctx || !ctx->tiles || !ctx->tiles_built || !ctx->tile_list || ctx->disable_tiles) return 0;
	uint32_t ctl[2];
	if (hipMemcpy(ctl, ctx->tile_ctl, sizeof(ctl), hipMemcpyDeviceToHost) != hipSuccess) return -1;
	return (ctl[0] > 0u && ctl[1] == 0u) ? 1 : 0;
}

// Optional profiling hook (bench.py): when enabled, every forces pass records a pair of HIP events on its launch
// stream around its dominant kernel (forces_tile_kernel, or forces_kernel when the tiling is not used);
// sphx_forces_timing_read synchronises the recorded events, returns their summed elapsed time and count, and
// releases them.
extern "C" int sphx_forces_timing(sphx_ctx *ctx, int enable)
{
	SPHX_REQUIRE(ctx != nullptr, "sphx_forces_timing: NULL ctx");
	if (!ctx->forces_events) ctx->forces_events = new std::vector<std::pair<hipEvent_t, hipEvent_t> >();
	ctx->time_forces = enable != 0;
	return SPHX_OK;
}

extern "C" int sphx_forces_timing_read(sphx_ctx *ctx, double *total_ms, uint32_t *launches)
{
	SPHX_REQUIRE(ctx && total_ms && launches, "sphx_forces_timing_read: NULL argument");
	double sum = 0.0; uint32_t n = 0;
	if (ctx->forces_events) {
		for (auto &e : *ctx->forces_events) {
			float ms = 0.0f;
			if (hipEventSynchronize(e.second) == hipSuccess && hipEventElapsedTime(&ms, e.first, e.second) == hipSuccess) { sum += ms; ++n; }
			(void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second);
		}
		ctx->forces_events->clear();
	}
	*total_ms = sum; *launches = n;
	return SPHX_OK;
}
#endif // !SPHX_FORCES_PART
