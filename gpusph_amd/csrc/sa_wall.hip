// sa_wall.hip -- SA_BOUNDARY: the boundary-element terms of the forces, of the density summation and of the gamma quadrature
// for the fluid particles next to a wall, one element per lane.  Compiled with floating-point contraction and with its own
// arctangent: these sums are formed in another order than the one-thread-per-particle kernels of sa_bounds.hip (which mirror the
// CPU oracle operation by operation) anyway, and agree with them to rounding (tests/test_gpu_sa.py runs every SA test both ways).
#include "sphx_internal.h"
#include "neib_iter.h"
#include "sa_args.h"
#include "wave_list.h"

// atan2 for the edge integrals: quotient by reciprocal + one correction step, arctangent on [0, 1] as t + t s q(s), s = t^2, q
// fitted here (minimax of the error of atan(t)/t, degree 9: 0.8 ulp on [0, 1]; the whole function stays within 1.8 ulp of the
// correctly rounded arctangent over random arguments, scripts/fit_atan.py) -- about 25 instructions, against ~50 of the library's
__device__ __forceinline__ float wall_fast_atan2(float y, float x)
{
	const float ax = fabsf(x), ay = fabsf(y);
	const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
	const float r = __builtin_amdgcn_rcpf(mx);
	float t = mn*r;
	t = fmaf(fmaf(-mx, t, mn), r, t);
	t = (mx == 0.0f || !(mx < 3.0e38f)) ? ((mn == mx && mx != 0.0f) ? 1.0f : 0.0f) : t;     // atan2(0, 0) = 0; infinities
	const float s = t*t;
	float q = -1.8272230183369424e-3f;
	q = fmaf(q, s, 1.1064477995577955e-2f);
	q = fmaf(q, s, -3.145575541231577e-2f);
	q = fmaf(q, s, 5.823458879035968e-2f);
	q = fmaf(q, s, -8.419460484655855e-2f);
	q = fmaf(q, s, 1.0957576381515889e-1f);
	q = fmaf(q, s, -1.4265245372329702e-1f);
	q = fmaf(q, s, 1.9998638615746267e-1f);
	q = fmaf(q, s, -3.3333301866324133e-1f);
	float a = fmaf(t*s, q, t);
	a = (ay > ax) ? 1.57079632679489661923f - a : a;
	a = (x < 0.0f) ? 3.14159265358979323846f - a : a;
	return copysignf(a, y);
}
#define SPHX_WG_ATAN2 wall_fast_atan2
#include "sa_wall_gamma.h"

// ---- boundary-element terms with one element per lane ---------------------------------------------------------------------
// A boundary element costs hundreds of instructions (wall_grad_gamma, wall_gamma) and only the few per cent of the fluid
// particles next to a wall have any: with one thread per particle those few waves run long and alone, each through its
// 25..80 elements one after the other.  The kernels below give every such particle (sa_wall_list_kernel, neibs.hip) a whole
// wave: lane l evaluates the l-th entry of the boundary section, the lanes' terms are summed by a butterfly.  The sums differ
// from the list-order sums of the one-thread kernels by rounding only; these stay as the fallback and the CPU oracle's mirror.
struct WallEntry { bool alive; uint32_t j; float pcx, pcy, pcz; };

// entries s0 .. s0+63 of the boundary section of particle `index` (it runs down from neibboundpos): decoded in parallel, the
// cell code of an entry is that of the nearest encoded entry at or before it (cellCarry: from the chunks before)
__device__ __forceinline__ WallEntry wall_chunk(const DevParams &p, const neibdata *__restrict__ list, const uint32_t *__restrict__ cellStart,
	uint32_t index, const float4 &pos, const int3 &gridPos, int s0, uint32_t lane, int &cellCarry, bool &more)
{
	WallEntry e;
	const int slot = (int)p.neibboundpos - (s0 + (int)lane);
	const uint32_t d = slot >= 0 ? (uint32_t)list[(size_t)slot*p.stride + index] : NEIBS_END;
	const unsigned long long endmask = __builtin_amdgcn_ballot_w64(d == NEIBS_END);
	const int firstEnd = endmask ? __builtin_ctzll(endmask) : 64;
	e.alive = (int)lane < firstEnd;
	const bool enc = e.alive && d >= CELLNUM_ENCODED;
	const unsigned long long encmask = __builtin_amdgcn_ballot_w64(enc);
	const unsigned long long le = encmask & (~0ull >> (63u - lane));      // encoded entries at or before this lane
	const int src = le ? 63 - __builtin_clzll(le) : 0;
	const int codeSrc = __shfl((int)(d >> CELLNUM_SHIFT), src) - 1;
	const int c = le ? codeSrc : cellCarry;
	cellCarry = __shfl(c, 63);
	more = firstEnd == 64;
	const int cz = c/9, cy = (c - cz*9)/3, cx = c - cz*9 - cy*3;
	e.j = index; e.pcx = pos.x; e.pcy = pos.y; e.pcz = pos.z;
	if (e.alive) {
		e.j = cellStart[grid_hash_periodic(p, gridPos.x + cx - 1, gridPos.y + cy - 1, gridPos.z + cz - 1)] + (d & NEIBINDEX_MASK);
		e.pcx = fmaf(-(float)(cx - 1), p.cs[0], pos.x);
		e.pcy = fmaf(-(float)(cy - 1), p.cs[1], pos.y);
		e.pcz = fmaf(-(float)(cz - 1), p.cs[2], pos.z);
	}
	return e;
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
	for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
	return v;
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
	for (int d = 32; d > 0; d >>= 1) v = fmaxf(v, __shfl_xor(v, d));
	return v;
}

#define SA_WALL_THREADS 256
// the wave's particles: wall[1 + w], w = wave number, +waves in the grid, ...
#define SA_WALL_LOOP(wall) \
	const uint32_t lane = threadIdx.x & 63u; \
	const uint32_t nWaves = gridDim.x*(SA_WALL_THREADS/64), count = (wall)[0]; \
	for (uint32_t w = blockIdx.x*(SA_WALL_THREADS/64) + (threadIdx.x >> 6); w < count; w += nWaves)

// |grad gamma_as| handed from one pass to the next (SaWallCache).  The density summation (or the gamma quadrature) of a step
// evaluates it for every element in reach of a particle at the particle's NEW position; the forces pass that follows runs at
// those very positions with the same list, so it would evaluate the same numbers again -- a third of an SA step.  The writers
// leave them per (wall particle w, entry of the boundary section) together with a tag: the bits of the position they were
// evaluated at and the generation of the neighbour list.  The reader takes a row only if both match what it sees itself, so
// anything that moved the particle or rebuilt the list in between simply makes it evaluate afresh.  (The walls themselves do
// not move: the engines this serves refuse moving SA bodies.)  Rows of particles with more than SA_WALL_CACHE_ENTRIES entries,
// and of wall particles beyond the capacity, are not kept.
__device__ __forceinline__ bool wall_cache_row(const SaWallCache &wc, uint32_t w) { return wc.values != nullptr && w < wc.capacity; }
__device__ __forceinline__ void wall_cache_put(const SaWallCache &wc, uint32_t w, uint32_t entry, float ggamAS)
{
	if (entry < SA_WALL_CACHE_ENTRIES) wc.values[(size_t)w*SA_WALL_CACHE_ENTRIES + entry] = ggamAS;
}
__device__ __forceinline__ void wall_cache_seal(const SaWallCache &wc, uint32_t w, const float4 &pos, bool complete)
{
	wc.tag[w] = make_float4(pos.x, pos.y, pos.z, __uint_as_float(complete ? wc.gen : 0u));
}
__device__ __forceinline__ bool wall_cache_valid(const SaWallCache &wc, uint32_t w, const float4 &pos)
{
	if (!wall_cache_row(wc, w) || !wc.gen) return false;
	const float4 t = wc.tag[w];
	return __float_as_uint(t.x) == __float_as_uint(pos.x) && __float_as_uint(t.y) == __float_as_uint(pos.y) &&
		__float_as_uint(t.z) == __float_as_uint(pos.z) && __float_as_uint(t.w) == wc.gen;
}

// sum_s grad gamma_as of a wall particle's elements at step n (SaWallCache::gsum): left by the forces pass of a run with open
// boundaries when it evaluates the elements afresh (the first pass of a step, right after the rebuild such a run does in every
// step), taken by the two density summations of that step -- both integrate gamma from the positions of step n, so the step-n
// half of their sum 1/2 (grad gamma_as(n) + grad gamma_as(n+1)) . dq is 1/2 (this sum) . dq for walls at rest, and the elements
// are evaluated once per pass instead of twice.  sa_density_sum_wall_kernel<false> takes that sum from BUFFER_GRADGAMMA; with open
// boundaries the rows of the particles an open vertex released in the previous step hold the VERTEX's gradient instead (the
// reference copies it, vertex rows of sa_io.hip), so the stored row is not trusted there.  Tag: position bits and list generation,
// as for the rows of |grad gamma_as|; no match (another list, the forces pass skipped or served from kept rows): two evaluations.
__device__ __forceinline__ void wall_gsum_put(const SaWallCache &wc, uint32_t w, const float4 &pos, float gx, float gy, float gz)
{
	wc.gsum[2u*w] = make_float4(gx, gy, gz, __uint_as_float(wc.gen));
	wc.gsum[2u*w + 1u] = make_float4(pos.x, pos.y, pos.z, 0.0f);
}
__device__ __forceinline__ bool wall_gsum_get(const SaWallCache &wc, uint32_t w, const float4 &pos, float4 &sum)
{
	if (wc.gsum == nullptr || w >= wc.capacity || !wc.gen) return false;
	sum = wc.gsum[2u*w];
	const float4 t = wc.gsum[2u*w + 1u];
	return __float_as_uint(sum.w) == wc.gen && __float_as_uint(t.x) == __float_as_uint(pos.x) &&
		__float_as_uint(t.y) == __float_as_uint(pos.y) && __float_as_uint(t.z) == __float_as_uint(pos.z);
}

// the fluid <- boundary-element part of sa_forces_kernel<false> (same terms, see there), added to the sums the tiled kernel left
__global__ void __launch_bounds__(SA_WALL_THREADS)
sa_forces_wall_kernel(DevParams p, SaForcesArgs a, const uint32_t *__restrict__ wall)
{
	if (a.tileGuard && *a.tileGuard) return;      // no tiles after all: sa_forces_kernel does everything
	SA_WALL_LOOP(wall) {
		const uint32_t index = __builtin_amdgcn_readfirstlane(wall[1u + w]);
		if (index < a.fromParticle || index >= a.toParticle) continue;
		const float4 pos = a.pos[index], vel = a.vel[index];
		const uint32_t fl = FLUID_NUM(a.info[index]);
		const int3 gridPos = grid_pos_from_hash(p, a.hash[index] & CELLTYPE_BITMASK);
		const float p_rho = (vel.w + 1.0f)*p.rho0[fl];
		const float p_precalc = sa_P(p, vel.w, fl)/(p_rho*p_rho);
		const bool density_sum = (p.simflags & SPHX_ENABLE_DENSITY_SUM) != 0;
		const bool newtonian = p.rheology == SPHX_NEWTONIAN;
		// a.open: a run with open boundaries (sa_forces_kernel<false, true>, see there): the viscous term of a boundary element sees the
		// relative velocity plus the relative Eulerian velocity (no normal part taken out for the segment of an open face), the gamma CFL
		// term gains compute_gamma_cfl_open_boundary, and the fluid <- vertex pairs, whose sums the tiled kernel formed with the relative
		// velocity alone, get the Eulerian part of their viscous term here (below)
		const bool open = a.open != 0;
		const float4 p_euler = open ? a.eulerVel[index] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
		float fx = 0.0f, fy = 0.0f, fz = 0.0f, fw = 0.0f, gammaCfl = 0.0f;
		// |grad gamma_as| of this particle's elements as the previous pass left them, if they were evaluated at this very position
		const bool kept = __builtin_amdgcn_readfirstlane((int)wall_cache_valid(a.wc, w, pos)) != 0;
		// an open run, elements evaluated here: their sum for the density summations of this step (wall_gsum_put)
		const bool leaveSum = open && !kept && a.wc.gsum != nullptr && w < a.wc.capacity;
		float sgx = 0.0f, sgy = 0.0f, sgz = 0.0f;
		int cellCarry = 0;
		bool more = true;
		for (int s0 = 0; more; s0 += 64) {
			const WallEntry e = wall_chunk(p, a.neibsList, a.cellStart, index, pos, gridPos, s0, lane, cellCarry, more);
			const uint32_t j = e.j;
			const float4 npos = a.pos[j];
			const float rx = e.pcx - npos.x, ry = e.pcy - npos.y, rz = e.pcz - npos.z;
			const float r = sqrtf(fmaf(rz, rz, fmaf(ry, ry, rx*rx)));
			// (the density summation has no cut at this distance: an element behind it whose corners reach into the support all the
			// same -- a wall meshed coarser than the particle spacing -- is in ITS sum, so it is in the sum left for it)
			const bool inReach = r < p.influenceradius + a.deltap;
			if (!e.alive || !is_active_w(npos.w) || (!inReach && !leaveSum)) continue;
			const float4 be = a.boundElement[j];
			float ggamAS;
			if (kept)
				ggamAS = a.wc.values[(size_t)w*SA_WALL_CACHE_ENTRIES + (uint32_t)s0 + lane];
			else {
				const float inv_h = 1.0f/p.slength;
				WallTri tri;
				wall_tri_setup(tri, v3(be.x, be.y, be.z), a.vertPos[0][j], a.vertPos[1][j], a.vertPos[2][j], p.slength);
				ggamAS = wall_grad_gamma_flat(tri, v3(rx*inv_h, ry*inv_h, rz*inv_h))/p.slength;
			}
			if (leaveSum) { sgx = fmaf(ggamAS, be.x, sgx); sgy = fmaf(ggamAS, be.y, sgy); sgz = fmaf(ggamAS, be.z, sgz); }
			if (!inReach) continue;
			const float4 nvel = a.vel[j];
			const float vx = vel.x - nvel.x, vy = vel.y - nvel.y, vz = vel.z - nvel.z;
			const uint32_t nfl = FLUID_NUM(a.info[j]);
			const float n_rho = (nvel.w + 1.0f)*p.rho0[nfl];
			const float n_precalc = sa_P(p, nvel.w, nfl)/(n_rho*n_rho);
			const float vn = sa_dot3(vx, vy, vz, be.x, be.y, be.z);
			float4 ne = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
			if (open) ne = a.eulerVel[j];
			if (a.cflGamma) {
				const float va = sa_dot3(vel.x, vel.y, vel.z, be.x, be.y, be.z);
				const float vs = sa_dot3(vel.x - vx, vel.y - vy, vel.z - vz, be.x, be.y, be.z);
				gammaCfl = fmaxf(gammaCfl, ggamAS*fmaxf(fabsf(vn), fmaxf(fabsf(va), fabsf(vs))));
				if (open) {      // n.(v_a + relEulerVel), n.(v_s - relEulerVel)
					const float ex = (vx + (p_euler.x - ne.x)) - vx, ey = (vy + (p_euler.y - ne.y)) - vy, ez = (vz + (p_euler.z - ne.z)) - vz;
					const float a1 = sa_dot3(vel.x + ex, vel.y + ey, vel.z + ez, be.x, be.y, be.z);
					const float a2 = sa_dot3(-vx + vel.x - ex, -vy + vel.y - ey, -vz + vel.z - ez, be.x, be.y, be.z);
					gammaCfl = fmaxf(gammaCfl, ggamAS*fmaxf(fabsf(a1), fabsf(a2)));
				}
			}
			if (!density_sum) fw -= p_rho*vn*ggamAS;
			const float ps = (p_precalc + n_precalc)*n_rho*ggamAS;
			float dx = ps*be.x, dy = ps*be.y, dz = ps*be.z;
			if (newtonian) {
				const float r_as = fmaxf(fabsf(sa_dot3(rx, ry, rz, be.x, be.y, be.z)), a.deltap);
				float tx = vx - vn*be.x, ty = vy - vn*be.y, tz = vz - vn*be.z;
				if (open) {
					const float wx = vx + (p_euler.x - ne.x), wy = vy + (p_euler.y - ne.y), wz = vz + (p_euler.z - ne.z);
					const float wn = SA_IS_OPEN(a.info[j]) ? 0.0f : sa_dot3(wx, wy, wz, be.x, be.y, be.z);
					tx = wx - wn*be.x; ty = wy - wn*be.y; tz = wz - wn*be.z;
				}
				const float our_mu = (p.compvisc == SPHX_KINEMATIC) ? p.visccoeff[fl]*p_rho : p.visccoeff[fl];
				const float neib_mu = (p.compvisc == SPHX_KINEMATIC) ? p.visccoeff[nfl]*n_rho : p.visccoeff[nfl];
				const float avg = (p.avgop == SPHX_ARITHMETIC) ? (our_mu + neib_mu)*0.5f :
					(p.avgop == SPHX_HARMONIC) ? 2*our_mu*neib_mu/(our_mu + neib_mu) : sqrtf(our_mu*neib_mu);
				const float c = ggamAS*2*avg/r_as;
				const float inv_rho = 1.0f/p_rho;
				dx -= (c*tx)*inv_rho; dy -= (c*ty)*inv_rho; dz -= (c*tz)*inv_rho;
			}
			fx += dx; fy += dy; fz += dz;
		}
		if (open && newtonian) {      // fluid <- vertex: the Eulerian part of the laminar term, vf (u_a^E - u_b^E) (get_viscous_relVel)
			int carry = 0;
			bool moreV = true;
			for (int s0 = 0; moreV; s0 += 64) {
				const WaveEntry e = wave_entries<WAVE_SECTION_VERTEX>(p, a.neibsList, a.cellStart, index, pos, gridPos, s0, lane, carry, moreV);
				const uint32_t j = e.j;
				const float4 ne = a.eulerVel[j];
				const float ex = p_euler.x - ne.x, ey = p_euler.y - ne.y, ez = p_euler.z - ne.z;
				const bool has = e.live && (ex != 0.0f || ey != 0.0f || ez != 0.0f);
				if (!__builtin_amdgcn_ballot_w64(has)) continue;      // vertices of walls: no Eulerian velocity on either side
				const float4 npos = a.pos[j];
				const float rx = e.ox - npos.x, ry = e.oy - npos.y, rz = e.oz - npos.z;
				const float r = sqrtf(fmaf(rz, rz, fmaf(ry, ry, rx*rx)));
				if (!has || !is_active_w(npos.w) || r >= p.influenceradius) continue;
				const float4 nvel = a.vel[j];
				const uint32_t nfl = FLUID_NUM(a.info[j]);
				const float n_rho = (nvel.w + 1.0f)*p.rho0[nfl];
				const float qm2 = r/p.slength - 2.0f;
				const float f = qm2*qm2*qm2*p.fcoeff;
				const float vf = sa_visc_avg(p, p.visccoeff[fl], p.visccoeff[nfl], p_rho, n_rho, npos.w)*f;
				fx += vf*ex; fy += vf*ey; fz += vf*ez;
			}
		}
		fx = wave_sum(fx); fy = wave_sum(fy); fz = wave_sum(fz); fw = wave_sum(fw); gammaCfl = wave_max(gammaCfl);
		if (leaveSum) {
			sgx = wave_sum(sgx); sgy = wave_sum(sgy); sgz = wave_sum(sgz);
			if (lane == 0) wall_gsum_put(a.wc, w, pos, sgx, sgy, sgz);
		}
		if (lane == 0) {
			float4 f = a.forces[index];
			f.x += fx; f.y += fy; f.z += fz; f.w += fw;
			a.forces[index] = f;
			if (a.cflGamma) a.cflGamma[index] = gammaCfl;
		}
	}
}

// the boundary-element sums of sa_density_sum_kernel: {sum grad gamma(n+1), sum 1/2 (grad gamma(n) + grad gamma(n+1)) . (q(n+1) - q(n))}
// into the particle's row of newGGam, where sa_density_sum_kernel picks them up.
// ONE evaluation of |grad gamma_as| per element, not two: the walls this entry point is built for do not move (it refuses
// ENABLE_MOVING_BODIES), so q(n+1) - q(n) is the particle's own displacement for every element and the step-n half of the
// second sum is 1/2 (sum_s grad gamma_as(n)) . dq -- and that sum is the particle's grad gamma of step n, which BUFFER_GRADGAMMA
// holds (oldGGam.xyz: written by this same engine at the end of the previous step, or by saInitGamma).  The one-thread kernel
// keeps the reference's two evaluations per element; the two agree to rounding.
// What that rests on: the stored grad gamma(n) and the sum over THIS list are sums over the same elements.  The lists differ after
// a rebuild between the two steps, and a halo copy's grad gamma was summed by the rank that owns the particle -- but only in
// elements that contribute nothing: an element wholly outside the kernel's support has |grad gamma_as| = 0 exactly
// (sa_wall_gamma.h dismisses it before any arithmetic) and every element inside the support is in any valid list (the boundary
// section is built out to boundNlSqInflRad >= the support).  So the assumption is the validity of the neighbour list itself,
// which every sum of this engine already needs.  Held by tests/test_gpu_sa.py (this kernel against the one-thread kernel over
// 12 sloshing steps across a rebuild) and by the two-rank run test_two_ranks_on_one_gpu_equal_single_domain[sa-walls-tiled] of tests/test_gpu_parity.py.
// OPEN: a run with open boundaries (sa_density_sum_kernel<true>, density_sum_kernel.cu:119-140,374-417).  Three things on top:
//  * the step-n sum is the one the forces pass of this step left (wall_gsum_get), not BUFFER_GRADGAMMA's; no such row: the elements
//    are evaluated at step n as well, as the one-thread kernel does;
//  * a segment of an open face adds the flux of gamma through it -- the virtual displacement dt (u_E - u) against grad gamma_as at
//    step n and at the displaced point -- to the gamma that multiplies the old density: the half sum goes to the particle's row
//    of newVel.w, where sa_density_sum_kernel<true> picks it up before it writes the new density there;
//  * a vertex of an open face is not a reservoir of mass at rest: the tiled kernel has summed -m W(r_n) for it as for any
//    neighbour; that term is taken back here and the kernel at the distance after the virtual displacement put in its place
//    (FORCES.w, which holds the tiled sums).
template<bool OPEN>
__global__ void __launch_bounds__(SA_WALL_THREADS)
sa_density_sum_wall_kernel(DevParams p, SaDensitySumArgs a, const uint32_t *__restrict__ wall)
{
	if (a.tileGuard && *a.tileGuard) return;
	SA_WALL_LOOP(wall) {
		const uint32_t index = __builtin_amdgcn_readfirstlane(wall[1u + w]);
		if (index >= a.numParticles) continue;
		const float4 posN = a.oldPos[index], posNp1 = a.pos[index];
		const int3 gridPos = grid_pos_from_hash(p, a.hash[index] & CELLTYPE_BITMASK);
		const float dx = posNp1.x - posN.x, dy = posNp1.y - posN.y, dz = posNp1.z - posN.z;
		float4 gGamN = a.oldGGam[index];
		bool haveN = true;
		if (OPEN) haveN = __builtin_amdgcn_readfirstlane((int)wall_gsum_get(a.wc, w, posN, gGamN)) != 0;
		const float inv = 1.0f/p.slength;
		float gx = 0.0f, gy = 0.0f, gz = 0.0f, dotNp1 = 0.0f;
		float nx = 0.0f, ny = 0.0f, nz = 0.0f, flux = 0.0f;
		bool anyOpen = false, seesOpen = false;
		const bool keep = wall_cache_row(a.wc, w);
		bool complete = true;
		int cellCarry = 0;
		bool more = true;
		for (int s0 = 0; more; s0 += 64) {
			const WallEntry e = wall_chunk(p, a.neibsList, a.cellStart, index, posN, gridPos, s0, lane, cellCarry, more);
			if (__builtin_amdgcn_ballot_w64(e.alive && (uint32_t)s0 + lane >= SA_WALL_CACHE_ENTRIES)) complete = false;
			const uint32_t j = e.j;
			const float4 nN = a.oldPos[j];
			if (!e.alive || !is_active_w(nN.w)) continue;
			const float4 nNp1 = a.pos[j];
			const V3 qN = v3((e.pcx - nN.x)*inv, (e.pcy - nN.y)*inv, (e.pcz - nN.z)*inv);
			const V3 qNp1 = v3(((e.pcx - nNp1.x) + dx)*inv, ((e.pcy - nNp1.y) + dy)*inv, ((e.pcz - nNp1.z) + dz)*inv);
			const float4 be = a.boundElement[j];
			const V3 ns = v3(be.x, be.y, be.z);
			WallTri tri;
			wall_tri_setup(tri, ns, a.vertPos[0][j], a.vertPos[1][j], a.vertPos[2][j], p.slength);
			const float ggamAS = wall_grad_gamma_flat(tri, qNp1)/p.slength;
			if (keep) wall_cache_put(a.wc, w, (uint32_t)s0 + lane, ggamAS);
			const V3 gNp1 = ns*ggamAS;
			dotNp1 += dot(gNp1, qNp1 - qN);
			gx += gNp1.x; gy += gNp1.y; gz += gNp1.z;
			if (OPEN) {
				const bool openSeg = SA_IS_OPEN(a.info[j]);
				if (!haveN || openSeg) {
					const float ggamN = wall_grad_gamma_flat(tri, qN)/p.slength;
					if (!haveN) { nx = fmaf(ggamN, ns.x, nx); ny = fmaf(ggamN, ns.y, ny); nz = fmaf(ggamN, ns.z, nz); }
					if (openSeg) {
						seesOpen = true;
						const float4 ev = a.oldEulerVel[j], v = a.oldVel[j];
						const V3 drift = v3(a.dt*(ev.x - v.x), a.dt*(ev.y - v.y), a.dt*(ev.z - v.z));
						const float ggamMoved = wall_grad_gamma_flat(tri, qN + drift*inv)/p.slength;
						flux += 0.5f*dot(drift, ns)*(ggamMoved + ggamN);
						anyOpen = true;
					}
				}
			}
		}
		float corr = 0.0f;
		if (OPEN) {      // the open vertices among the neighbours
			int carry = 0;
			bool moreV = true;
			for (int s0 = 0; moreV; s0 += 64) {
				const WaveEntry e = wave_entries<WAVE_SECTION_VERTEX>(p, a.neibsList, a.cellStart, index, posN, gridPos, s0, lane, carry, moreV);
				const uint32_t j = e.j;
				const bool openV = e.live && SA_IS_OPEN(a.info[j]);
				if (!__builtin_amdgcn_ballot_w64(openV)) continue;
				const float4 nN = a.oldPos[j];
				if (!openV || !is_active_w(nN.w)) continue;
				anyOpen = true;
				const float rx = e.ox - nN.x, ry = e.oy - nN.y, rz = e.oz - nN.z;
				const float rN = sqrtf(rx*rx + ry*ry + rz*rz);
				corr += nN.w*kernel_W<SPHX_WENDLAND>(p, rN);
				const float4 ev = a.oldEulerVel[j], v = a.oldVel[j];
				const float ex = rx + a.dt*(ev.x - v.x), ey = ry + a.dt*(ev.y - v.y), ez = rz + a.dt*(ev.z - v.z);
				const float moved = sqrtf(ex*ex + ey*ey + ez*ez);
				if (moved < p.influenceradius) corr -= nN.w*kernel_W<SPHX_WENDLAND>(p, moved);
			}
			anyOpen = __builtin_amdgcn_ballot_w64(anyOpen) != 0;
		}
		gx = wave_sum(gx); gy = wave_sum(gy); gz = wave_sum(gz); dotNp1 = wave_sum(dotNp1);
		if (OPEN && !haveN) { gGamN.x = wave_sum(nx); gGamN.y = wave_sum(ny); gGamN.z = wave_sum(nz); }
		if (OPEN && anyOpen) { flux = wave_sum(flux); corr = wave_sum(corr); }
		if (lane == 0) {
			const float dotN = (gGamN.x*dx + gGamN.y*dy + gGamN.z*dz)*inv;
			a.newGGam[index] = make_float4(gx, gy, gz, 0.5f*(dotN + dotNp1));
			if (keep) wall_cache_seal(a.wc, w, posNp1, complete);
			if (OPEN) {
				a.newVel[index].w = anyOpen ? flux : 0.0f;
				if (anyOpen) a.forces[index].w += corr;
			}
		}
		if (OPEN && a.openList && __builtin_amdgcn_ballot_w64(seesOpen)) {      // for the Brezzi diffusion that follows (sphx_sa_wall_density_diffusion_open)
			if (lane == 0) a.openList[1u + atomicAdd(&a.openList[0], 1u)] = index;
		}
	}
}

// ... with ENABLE_MOVING_BODIES (sa_density_sum_kernel<., MOVING>, density_sum_kernel.cu:422-484): an element is where it was AND where
// it is -- position and normal of both states.  The fluid particles next to a wall and, as a list of their own (ctx->sa_wall_vert;
// nothing is kept of them), the VERTEX rows, whose gamma these terms integrate.  |grad gamma_as| of the NEW state is kept for the
// forces pass that follows at that state, as for walls at rest: a row is tagged with the particle's position and the generation of
// the rows, and every call that moves elements (the Euler step of such a run, sphx_sa_update_normals) starts a new generation.
// (Round 6: the one-thread kernel was 61 % of a step of the SAPaddleBox mirror at 4.3 M particles,
// profiles/r06_sa_moving_kernel_stats.txt)
// OPEN: ... together with open boundaries (sa_density_sum_kernel<true, true>, the option set of CompleteSaExample.cu).  The FLUID rows get what
// sa_density_sum_wall_kernel<true> adds -- the flux of gamma through the open segments (against the element as it is at step n, as the
// one-thread kernel has it), the virtual displacement of the open vertices, the list for the Brezzi diffusion -- and take NOTHING from the
// stored grad gamma of step n: the row of a particle an open vertex released holds the vertex's gradient, and the sum the forces pass leaves
// for runs with walls at rest belongs to a generation of rows that ends when the elements move; every element is evaluated at both states.
// The vertex rows are integrated as without open boundaries.
template<bool OPEN>
__global__ void __launch_bounds__(SA_WALL_THREADS)
sa_density_sum_wall_moving_kernel(DevParams p, SaDensitySumArgs a, const uint32_t *__restrict__ wall)
{
	if (a.tileGuard && *a.tileGuard) return;
	SA_WALL_LOOP(wall) {
		const uint32_t index = __builtin_amdgcn_readfirstlane(wall[1u + w]);
		if (index >= a.numParticles) continue;
		const bool openFluid = OPEN && PART_TYPE(a.info[index]) == PT_FLUID;      // (wave-uniform)
		const float4 posN = a.oldPos[index], posNp1 = a.pos[index];
		const int3 gridPos = grid_pos_from_hash(p, a.hash[index] & CELLTYPE_BITMASK);
		const float dx = posNp1.x - posN.x, dy = posNp1.y - posN.y, dz = posNp1.z - posN.z;
		const float inv = 1.0f/p.slength;
		float gx = 0.0f, gy = 0.0f, gz = 0.0f, dotSum = 0.0f;
		float mx = 0.0f, my = 0.0f, mz = 0.0f;      // grad gamma_as(n) of the elements that moved
		float flux = 0.0f;
		bool seesOpen = false;
		// a particle that did not move (a vertex of a wall at rest) none of whose elements moved: the sums are those of the previous
		// step, to the bit -- the same elements seen from the same place by this same kernel (after a rebuild or the initialisation:
		// the same numbers by another route) -- and are taken from there.  Most vertex rows of a run are of this kind
		if (dx == 0.0f && dy == 0.0f && dz == 0.0f && !openFluid) {
			bool moved = false;
			int carry0 = 0;
			bool more0 = true;
			for (int s0 = 0; more0; s0 += 64) {
				const WallEntry e = wall_chunk(p, a.neibsList, a.cellStart, index, posN, gridPos, s0, lane, carry0, more0);
				const uint32_t j = e.j;
				const float4 nN = a.oldPos[j], nNp1 = a.pos[j], be = a.boundElement[j], ben = a.boundElementNew[j];
				const bool same = __float_as_uint(be.x) == __float_as_uint(ben.x) && __float_as_uint(be.y) == __float_as_uint(ben.y) &&
					__float_as_uint(be.z) == __float_as_uint(ben.z) && __float_as_uint(nN.x) == __float_as_uint(nNp1.x) &&
					__float_as_uint(nN.y) == __float_as_uint(nNp1.y) && __float_as_uint(nN.z) == __float_as_uint(nNp1.z);
				if (__builtin_amdgcn_ballot_w64(e.alive && is_active_w(nN.w) && !same)) moved = true;
			}
			if (!moved) {
				if (lane == 0) {
					const float4 gGamN = a.oldGGam[index];
					a.newGGam[index] = make_float4(gGamN.x, gGamN.y, gGamN.z, 0.0f);
				}
				continue;      // (nothing kept: the forces pass does not read rows of particles at rest -- fluid particles move)
			}
		}
		const bool keep = wall_cache_row(a.wc, w);
		bool complete = true;
		int cellCarry = 0;
		bool more = true;
		for (int s0 = 0; more; s0 += 64) {
			const WallEntry e = wall_chunk(p, a.neibsList, a.cellStart, index, posN, gridPos, s0, lane, cellCarry, more);
			if (__builtin_amdgcn_ballot_w64(e.alive && (uint32_t)s0 + lane >= SA_WALL_CACHE_ENTRIES)) complete = false;
			const uint32_t j = e.j;
			const float4 nN = a.oldPos[j];
			if (!e.alive || !is_active_w(nN.w)) continue;
			const float4 nNp1 = a.pos[j];
			const V3 qN = v3((e.pcx - nN.x)*inv, (e.pcy - nN.y)*inv, (e.pcz - nN.z)*inv);
			const V3 qNp1 = v3(((e.pcx - nNp1.x) + dx)*inv, ((e.pcy - nNp1.y) + dy)*inv, ((e.pcz - nNp1.z) + dz)*inv);
			const float4 be = a.boundElement[j], ben = a.boundElementNew[j];
			const V3 ns = v3(be.x, be.y, be.z), nsNew = v3(ben.x, ben.y, ben.z);
			// Most elements of such a run do not move (the same bits in both states).  For those q(n+1) - q(n) is the particle's own
			// displacement, the same for all of them, so their step-n half of the second sum is 1/2 (sum grad gamma_as(n)) . dq with
			// the sum taken from the particle's stored grad gamma of step n minus the moving elements' part of it (what
			// sa_density_sum_wall_kernel does for walls at rest, see there for what it rests on): ONE evaluation per element at
			// rest, two -- the corners set up once per normal -- only for the elements that moved
			const bool sameEl = __float_as_uint(be.x) == __float_as_uint(ben.x) && __float_as_uint(be.y) == __float_as_uint(ben.y) &&
				__float_as_uint(be.z) == __float_as_uint(ben.z) && __float_as_uint(nN.x) == __float_as_uint(nNp1.x) &&
				__float_as_uint(nN.y) == __float_as_uint(nNp1.y) && __float_as_uint(nN.z) == __float_as_uint(nNp1.z);
			WallTri tri;
			V3 gN = v3(0.0f, 0.0f, 0.0f);
			const bool both = !sameEl || openFluid;      // evaluated at step n as well
			if (both) {
				wall_tri_setup(tri, ns, a.vertPos[0][j], a.vertPos[1][j], a.vertPos[2][j], p.slength);
				const float ggamN = wall_grad_gamma_flat(tri, qN)/p.slength;
				gN = ns*ggamN;
				if (OPEN && openFluid && SA_IS_OPEN(a.info[j])) {      // the flux of gamma through an open segment, wholly at step n
					seesOpen = true;
					const float4 ev = a.oldEulerVel[j], v = a.oldVel[j];
					const V3 drift = v3(a.dt*(ev.x - v.x), a.dt*(ev.y - v.y), a.dt*(ev.z - v.z));
					const float ggamMoved = wall_grad_gamma_flat(tri, qN + drift*inv)/p.slength;
					flux += 0.5f*dot(drift, ns)*(ggamMoved + ggamN);
				}
			}
			wall_tri_setup(tri, nsNew, a.vertPos[0][j], a.vertPos[1][j], a.vertPos[2][j], p.slength);
			const float ggamNp1 = wall_grad_gamma_flat(tri, qNp1)/p.slength;
			if (keep) wall_cache_put(a.wc, w, (uint32_t)s0 + lane, ggamNp1);      // for the forces pass at this state (fluid rows)
			const V3 gNp1 = nsNew*ggamNp1;
			// an element at rest: its step-(n+1) half here, its step-n half from the stored sum below; one evaluated at both states: both halves
			dotSum += both ? 0.5f*dot(gN + gNp1, qNp1 - qN) : 0.5f*dot(gNp1, qNp1 - qN);
			mx += gN.x; my += gN.y; mz += gN.z;
			gx += gNp1.x; gy += gNp1.y; gz += gNp1.z;
		}
		float corr = 0.0f;
		bool anyOpenV = false;
		if (OPEN && openFluid) {      // the open vertices among the neighbours (see sa_density_sum_wall_kernel<true>)
			int carry = 0;
			bool moreV = true;
			for (int s0 = 0; moreV; s0 += 64) {
				const WaveEntry e = wave_entries<WAVE_SECTION_VERTEX>(p, a.neibsList, a.cellStart, index, posN, gridPos, s0, lane, carry, moreV);
				const uint32_t j = e.j;
				const bool openV = e.live && SA_IS_OPEN(a.info[j]);
				if (!__builtin_amdgcn_ballot_w64(openV)) continue;
				const float4 nN = a.oldPos[j];
				if (!openV || !is_active_w(nN.w)) continue;
				anyOpenV = true;
				const float rx = e.ox - nN.x, ry = e.oy - nN.y, rz = e.oz - nN.z;
				const float rN = sqrtf(rx*rx + ry*ry + rz*rz);
				corr += nN.w*kernel_W<SPHX_WENDLAND>(p, rN);
				const float4 ev = a.oldEulerVel[j], v = a.oldVel[j];
				const float ex = rx + a.dt*(ev.x - v.x), ey = ry + a.dt*(ev.y - v.y), ez = rz + a.dt*(ev.z - v.z);
				const float moved = sqrtf(ex*ex + ey*ey + ez*ez);
				if (moved < p.influenceradius) corr -= nN.w*kernel_W<SPHX_WENDLAND>(p, moved);
			}
			anyOpenV = __builtin_amdgcn_ballot_w64(anyOpenV) != 0;
			if (anyOpenV) corr = wave_sum(corr);
			flux = wave_sum(flux);
		}
		mx = wave_sum(mx); my = wave_sum(my); mz = wave_sum(mz);
		gx = wave_sum(gx); gy = wave_sum(gy); gz = wave_sum(gz); dotSum = wave_sum(dotSum);
		if (lane == 0) {
			const float4 gGamN = a.oldGGam[index];
			// the elements at rest at step n (with open boundaries a fluid row has none of that kind: all were evaluated)
			const float dotRest = openFluid ? 0.0f : ((gGamN.x - mx)*dx + (gGamN.y - my)*dy + (gGamN.z - mz)*dz)*inv;
			a.newGGam[index] = make_float4(gx, gy, gz, dotSum + 0.5f*dotRest);
			if (keep) wall_cache_seal(a.wc, w, posNp1, complete);
			if (OPEN && openFluid) {
				a.newVel[index].w = flux;
				if (anyOpenV) a.forces[index].w += corr;
			}
		}
		if (OPEN && openFluid && a.openList && __builtin_amdgcn_ballot_w64(seesOpen)) {
			if (lane == 0) a.openList[1u + atomicAdd(&a.openList[0], 1u)] = index;
		}
	}
}

// sa_integrate_gamma_kernel for the particles with boundary elements in reach
__global__ void __launch_bounds__(SA_WALL_THREADS)
sa_integrate_gamma_wall_kernel(DevParams p, SaIntGammaArgs a, const uint32_t *__restrict__ wall)
{
	SA_WALL_LOOP(wall) {
		const uint32_t index = __builtin_amdgcn_readfirstlane(wall[1u + w]);
		if (index >= a.numParticles) continue;
		const float4 pos = a.pos[index], og = a.oldGGam[index];
		const int3 gridPos = grid_pos_from_hash(p, a.hash[index] & CELLTYPE_BITMASK);
		const V3 oldg = v3(og.x, og.y, og.z);
		float gx = 0.0f, gy = 0.0f, gz = 0.0f, gam = 0.0f;
		const bool keep = wall_cache_row(a.wc, w);
		bool complete = true;
		int cellCarry = 0;
		bool more = true;
		for (int s0 = 0; more; s0 += 64) {
			const WallEntry e = wall_chunk(p, a.neibsList, a.cellStart, index, pos, gridPos, s0, lane, cellCarry, more);
			if (__builtin_amdgcn_ballot_w64(e.alive && (uint32_t)s0 + lane >= SA_WALL_CACHE_ENTRIES)) complete = false;
			if (!e.alive) continue;
			const uint32_t j = e.j;
			const float4 npos = a.pos[j];
			const float4 be = a.boundElement[j];
			const V3 normal = v3(be.x, be.y, be.z);
			const V3 q = v3(e.pcx - npos.x, e.pcy - npos.y, e.pcz - npos.z)/p.slength;
			WallTri tri;
			wall_tri_setup(tri, normal, a.vertPos[0][j], a.vertPos[1][j], a.vertPos[2][j], p.slength);
			const float ggamAS = wall_grad_gamma_flat(tri, q)/p.slength;
			if (keep) wall_cache_put(a.wc, w, (uint32_t)s0 + lane, ggamAS);
			gx += ggamAS*be.x; gy += ggamAS*be.y; gz += ggamAS*be.z;
			gam += wall_gamma_flat<false>(tri, q, oldg, p.slength, a.epsilon);
		}
		gx = wave_sum(gx); gy = wave_sum(gy); gz = wave_sum(gz); gam = wave_sum(gam);
		if (lane == 0) {
			a.newGGam[index] = make_float4(gx, gy, gz, 1.0f - gam);
			if (keep) wall_cache_seal(a.wc, w, pos, complete);
		}
	}
}

// Brezzi diffusion in a run with open boundaries (sa_density_diffusion_kernel<true>): what the segments of the pressure-driven open
// faces exchange with the fluid particles next to them, added to the rows the tiled kernel finished (fluid <- fluid, / gamma / rho0)
__global__ void __launch_bounds__(SA_WALL_THREADS)
sa_density_diffusion_open_wall_kernel(DevParams p, SaDiffusionArgs a, const uint32_t *__restrict__ wall)
{
	if (a.tileGuard && *a.tileGuard) return;      // no tiles after all: the stand-by launch of the walker is the whole pass
	SA_WALL_LOOP(wall) {
		const uint32_t index = __builtin_amdgcn_readfirstlane(wall[1u + w]);
		if (index >= a.numParticles) continue;
		const float4 pos = a.pos[index], vel = a.vel[index];
		const uint32_t fl = FLUID_NUM(a.info[index]);
		const float rho = (vel.w + 1.0f)*p.rho0[fl];
		const float pres = sa_P(p, vel.w, fl);
		const int3 gridPos = grid_pos_from_hash(p, a.hash[index] & CELLTYPE_BITMASK);
		float sum = 0.0f;
		bool any = false;
		int cellCarry = 0;
		bool more = true;
		for (int s0 = 0; more; s0 += 64) {
			const WallEntry e = wall_chunk(p, a.neibsList, a.cellStart, index, pos, gridPos, s0, lane, cellCarry, more);
			const uint32_t j = e.j;
			const particleinfo ninfo = a.info[j];
			const bool open = e.alive && SA_IS_OPEN(ninfo) && !SA_IS_VELOCITY_DRIVEN(ninfo);
			if (!__builtin_amdgcn_ballot_w64(open)) continue;      // most particles next to a wall see no open face
			const float4 npos = a.pos[j];
			const float rx = e.pcx - npos.x, ry = e.pcy - npos.y, rz = e.pcz - npos.z;
			const float r = sqrtf(rx*rx + ry*ry + rz*rz);
			if (!open || !is_active_w(npos.w) || r >= p.influenceradius + a.deltap) continue;
			any = true;
			const float4 be = a.boundElement[j];
			const float r_as = fmaxf(fabsf(sa_dot3(rx, ry, rz, be.x, be.y, be.z)), a.deltap);
			const float inv_h = 1.0f/p.slength;
			WallTri tri;
			wall_tri_setup(tri, v3(be.x, be.y, be.z), a.vertPos[0][j], a.vertPos[1][j], a.vertPos[2][j], p.slength);
			const float ggamAS = wall_grad_gamma_flat(tri, v3(rx*inv_h, ry*inv_h, rz*inv_h))/p.slength;
			const float nrt = a.vel[j].w;
			const uint32_t nfl = FLUID_NUM(ninfo);
			const float neib_rho = (nrt + 1.0f)*p.rho0[nfl];
			const float gdotr = p.gravity[0]*rx + p.gravity[1]*ry + p.gravity[2]*rz;
			const double t = ((2.0/(rho + neib_rho))*(pres - sa_P(p, nrt, nfl)) - gdotr)*ggamAS/r_as*a.dt*2.0f*rho;
			sum -= (float)t;
		}
		if (!__builtin_amdgcn_ballot_w64(any)) continue;
		sum = wave_sum(sum);
		if (lane == 0) a.forces[index].w += (sum/a.gGam[index].w)/p.rho0[fl];
	}
}

// a grid that fills the device with waves; each takes every (number of waves)-th wall particle
static uint32_t sa_wall_grid(const sphx_ctx *ctx) { return ctx->tile_grid*8u; }

int sphx_sa_wall_forces(sphx_ctx *ctx, const SaForcesArgs &a, hipStream_t st)
{
	sa_forces_wall_kernel<<<sa_wall_grid(ctx), SA_WALL_THREADS, 0, st>>>(ctx->dev, a, ctx->sa_wall);
	SPHX_LAUNCH_CHECK("sa_forces_wall_kernel");
	return SPHX_OK;
}
int sphx_sa_wall_density_sum(sphx_ctx *ctx, const SaDensitySumArgs &a, hipStream_t st)
{
	if (a.oldEulerVel) sa_density_sum_wall_kernel<true><<<sa_wall_grid(ctx), SA_WALL_THREADS, 0, st>>>(ctx->dev, a, ctx->sa_wall);
	else sa_density_sum_wall_kernel<false><<<sa_wall_grid(ctx), SA_WALL_THREADS, 0, st>>>(ctx->dev, a, ctx->sa_wall);
	SPHX_LAUNCH_CHECK("sa_density_sum_wall_kernel");
	return SPHX_OK;
}
int sphx_sa_wall_density_sum_moving(sphx_ctx *ctx, const SaDensitySumArgs &a, hipStream_t st)
{
	if (a.oldEulerVel) sa_density_sum_wall_moving_kernel<true><<<sa_wall_grid(ctx), SA_WALL_THREADS, 0, st>>>(ctx->dev, a, ctx->sa_wall);
	else sa_density_sum_wall_moving_kernel<false><<<sa_wall_grid(ctx), SA_WALL_THREADS, 0, st>>>(ctx->dev, a, ctx->sa_wall);
	SPHX_LAUNCH_CHECK("sa_density_sum_wall_moving_kernel");
	if (a.wallDone & 2) {      // the vertex rows (row numbers of another list: nothing kept; no open terms: they are the fluid rows')
		SaDensitySumArgs av = a;
		av.wc.values = nullptr; av.wc.tag = nullptr; av.wc.capacity = 0; av.wc.gen = 0;
		av.openList = nullptr;
		sa_density_sum_wall_moving_kernel<false><<<sa_wall_grid(ctx), SA_WALL_THREADS, 0, st>>>(ctx->dev, av, ctx->sa_wall_vert);
		SPHX_LAUNCH_CHECK("sa_density_sum_wall_moving_kernel<vertices>");
	}
	return SPHX_OK;
}
int sphx_sa_wall_density_diffusion_open(sphx_ctx *ctx, const SaDiffusionArgs &a, hipStream_t st)
{
	// the particles with an open segment in reach, if the density summation of this step left their list for this neighbour list
	// (a fifth of the wall particles of the open channel: 0.80 -> 0.32 ms per pass at 8.6 M particles); else every wall particle
	const bool shortList = ctx->sa_wall_open && ctx->sa_wall_open_neibslist == a.neibsList && ctx->sa_wall_open_gen == ctx->sa_wall_gen;
	sa_density_diffusion_open_wall_kernel<<<sa_wall_grid(ctx), SA_WALL_THREADS, 0, st>>>(ctx->dev, a, shortList ? ctx->sa_wall_open : ctx->sa_wall);
	SPHX_LAUNCH_CHECK("sa_density_diffusion_open_wall_kernel");
	return SPHX_OK;
}
int sphx_sa_wall_integrate_gamma(sphx_ctx *ctx, const SaIntGammaArgs &a, hipStream_t st)
{
	sa_integrate_gamma_wall_kernel<<<sa_wall_grid(ctx), SA_WALL_THREADS, 0, st>>>(ctx->dev, a, ctx->sa_wall);
	SPHX_LAUNCH_CHECK("sa_integrate_gamma_wall_kernel");
	return SPHX_OK;
}
