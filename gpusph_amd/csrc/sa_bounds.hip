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
#include "sa_args.h"


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
	// :1813-1814: a vertex of an open boundary averages over that boundary's segments, any other vertex over the solid ones
	// (FG_INLET | FG_OUTLET, src/particleinfo.h:153-154; without such flags anywhere: every adjacent segment)
	const uint32_t io_flags = (PART_FLAG_START << 2) | (PART_FLAG_START << 3);
	const bool our_io = (info.x & io_flags) != 0;
	for_each_neib<PT_BOUNDARY>(p, a, index, pos, gridPos, [&](uint32_t j, const float4 &, float, float, float) {
		if (((a.info[j].x & io_flags) != 0) != our_io) return;
		if (!has_vertex(a.vertices[j], our_id)) return;
		const float4 be = a.boundElement[j];
		ax += be.x*be.w; ay += be.y*be.w; az += be.z*be.w;
	});
	const float inv = 1.0f/sqrtf(ax*ax + ay*ay + az*az);
	a.boundElement[index] = make_float4(ax*inv, ay*inv, az*inv, NAN);
}

template<int KERNEL, bool KEPS>
__global__ void __launch_bounds__(128)
sa_segment_bc_kernel(DevParams p, SaArgs a)
{
	uint32_t index = blockIdx.x*128 + threadIdx.x;
	if (a.rows) {      // thread t: the t-th boundary element
		if (index >= a.rows[0]) return;
		index = a.rows[1u + index];
	}
	if (index >= a.numParticles) return;
	const particleinfo info = a.info[index];
	if (!IS_BOUNDARY(info)) return;
	if (a.openFaces && SA_IS_OPEN(info)) return;       // the segments of the open faces are sa_io.hip's (one wave each)
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
	float sumtke = 0.0f, sumeps = 0.0f;                                   // common_keps_pout (:509-520)
	float4 eulerVel = make_float4(0.0f, 0.0f, 0.0f, 0.0f);                // eulervel_pout (:470-484)

	for_each_neib<PT_VERTEX>(p, a, index, pos, gridPos, [&](uint32_t j, const float4 &npos, float, float, float) {
		if (!is_active_w(npos.w)) return;
		if (!has_vertex(verts, info_id(a.info[j]))) return;
		if (KEPS) {                       // keps_vertex_contrib (:748-758)
			const float4 e = a.eulerVel[j];
			eulerVel.x += e.x; eulerVel.y += e.y; eulerVel.z += e.z; eulerVel.w += e.w;
		}
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
		if (KEPS) {      // keps_fluid_contrib (:816-826): dk/dn = 0, de/dn = 4 c_mu^(3/4) k^(3/2)/(kappa r) (de_dn_solid :806-813)
			const float norm_dist = fmaxf(fabsf(normal.x*rx + normal.y*ry + normal.z*rz), a.deltap);
			const float nk = a.tke[j], ne = a.eps[j];
			sumtke += n.w*nk;
			sumeps += n.w*(ne + 1.603090412f*powf(nk, 1.5f)/norm_dist);
		}
		shepard_div += n.w;
	});
	// impose_solid_bc (:1295-1306)
	shepard_div = fmaxf(shepard_div, 0.1f*gGam.w);
	vel.w = eos_rho(p, sumpWall/shepard_div, fl);
	a.vel[index] = vel;
	if (!KEPS && a.openFaces) a.eulerVel[index] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);      // impose_solid_eulerVel of a run with open boundaries
	if (KEPS) {          // impose_solid_keps_bc (:1262-1277); the normal is the float4 boundary element: its .w rides along
		a.tke[index] = sumtke/shepard_div;
		a.eps[index] = fmaxf(sumeps/shepard_div, 1e-5f);
		const float inv = 1.0f/3;
		eulerVel.x *= inv; eulerVel.y *= inv; eulerVel.z *= inv; eulerVel.w *= inv;
		const float d = eulerVel.x*normal.x + eulerVel.y*normal.y + eulerVel.z*normal.z;
		eulerVel.x -= d*normal.x; eulerVel.y -= d*normal.y; eulerVel.z -= d*normal.z; eulerVel.w -= d*normal.w;
		a.eulerVel[index] = eulerVel;
	}
}

template<int KERNEL, bool KEPS>
__global__ void __launch_bounds__(128)
sa_vertex_bc_kernel(DevParams p, SaArgs a)
{
	uint32_t index = blockIdx.x*128 + threadIdx.x;
	if (a.rows) {      // thread t: the t-th vertex particle
		if (index >= a.rows[0]) return;
		index = a.rows[1u + index];
	}
	if (index >= a.numParticles) return;
	const particleinfo info = a.info[index];
	if (PART_TYPE(info) != PT_VERTEX) return;
	if (a.openFaces && SA_IS_OPEN(info) && !SA_IS_CORNER(info)) return;      // sa_io.hip's; corner vertices are walls to this pass
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
	// vertex_boundary_loop (:1002-1021) with KEPSILON: k and epsilon of a vertex are the means over its adjacent segments
	// (keps_boundary_contrib :918-927, impose_vertex_keps_bc :1052-1072); its Eulerian velocity is made tangential to the wall
	float sumtke = 0.0f, sumeps = 0.0f; int numseg = 0;
	if (KEPS) {
		const uint32_t our_id = info_id(info);
		for_each_neib<PT_BOUNDARY>(p, a, index, pos, gridPos, [&](uint32_t j, const float4 &, float, float, float) {
			if (!has_vertex(a.vertices[j], our_id)) return;
			sumtke += a.tke[j]; sumeps += a.eps[j]; numseg += 1;
		});
	}
	shepard_div = fmaxf(shepard_div, 0.1f*gam);
	a.vel[index].w = eos_rho(p, sumpWall/shepard_div, fl);
	if (KEPS) {
		a.tke[index] = fmaxf(sumtke/numseg, 1e-6f);
		a.eps[index] = fmaxf(sumeps/numseg, 1e-6f);
		const float4 nrm = a.boundElement[index];
		float4 e = a.eulerVel[index];
		const float d = e.x*nrm.x + e.y*nrm.y + e.z*nrm.z;
		e.x -= d*nrm.x; e.y -= d*nrm.y; e.z -= d*nrm.z;
		a.eulerVel[index] = e;
	}
}

#include "sa_wall_gamma.h"

struct SaGammaArgs {
	float4 *newGGam;
	const float4 *pos, *boundElement;
	const float2 *vertPos[3];
	const particleinfo *info;
	const uint32_t *hash, *cellStart;
	const neibdata *neibsList;
	uint32_t numParticles;
	float deltap, epsilon;
};

template<int CPTYPE>
__global__ void __launch_bounds__(128)
sa_init_gamma_kernel(DevParams p, SaGammaArgs a)
{
	const uint32_t index = blockIdx.x*128 + threadIdx.x;
	if (index >= a.numParticles) return;
	const particleinfo info = a.info[index];
	if (PART_TYPE(info) != (uint32_t)CPTYPE) return;
	const float4 pos = a.pos[index];
	const int3 gridPos = grid_pos_from_hash(p, a.hash[index] & CELLTYPE_BITMASK);
	float gam = 1.0f;
	V3 gGam = v3(0.0f, 0.0f, 0.0f);
	// grad gamma first (gamma needs its direction), then gamma: two walks over the boundary elements in reach
#pragma unroll 1
	for (int pass = 0; pass < 2; ++pass) {
		for_each_neib<PT_BOUNDARY>(p, a, index, pos, gridPos, [&](uint32_t j, const float4 &, float rx, float ry, float rz) {
			const V3 relPos = v3(rx, ry, rz);                       // InitGammaVars :1833-1868
			if (length(relPos) > p.influenceradius + a.deltap*0.5f) return;
			const float4 be = a.boundElement[j];
			const V3 normal = v3(be.x, be.y, be.z);
			const V3 q = relPos/p.slength;
			WallTri tri;
			wall_tri_setup(tri, normal, a.vertPos[0][j], a.vertPos[1][j], a.vertPos[2][j], p.slength);
			if (pass == 0) gGam = gGam + normal*(wall_grad_gamma(tri, q)/p.slength);
			else gam -= wall_gamma<CPTYPE == PT_VERTEX>(tri, q, gGam, p.slength, a.epsilon);
		});
	}
	a.newGGam[index] = make_float4(gGam.x, gGam.y, gGam.z, gam);
}

// ---- forces with SA_BOUNDARY (solid walls, Newtonian laminar viscosity or inviscid, continuity equation) ----------------
// forcesDevice<PT_FLUID, PT_FLUID | PT_VERTEX | PT_BOUNDARY> + finalizeforcesDevice (src/cuda/forces_kernel.def:3914-4150)
// in one launch per fluid particle: fluid and vertex neighbours interact as particles, a boundary element contributes
// through |grad gamma_as| (continuity :2079-2090, pressure :2414-2427, wall shear :2680-2718); the sums are divided by gamma
// (forces_fixup :3192-3210).  Written like the boundary-conditions kernels above: the reference's operation order, IEEE
// division and sqrt, no contraction other than the fmaf the CPU oracle spells out.  This kernel is the CPU oracle's mirror and the
// fallback; on a tiled neighbour list the particle <- particle sums come from forces_tile_kernel (SPHX_TURB_SA, forces.hip), the
// boundary-element terms from sa_forces_wall_kernel (sa_wall.hip), and this kernel only finishes the particle (a.tiled, a.wallDone).

// sa_dot3, sa_P, sa_sound_speed, sa_visc_avg: neib_iter.h (shared with the other fidelity engines)

// KEPS: the k-epsilon model on top (solid walls).  The reference runs forcesDevice once per neighbour type and every launch
// starts from a fresh keps_particle_output and stores it (forces_particle_output :1003-1013, write_keps :3331-3339), so the DKDE /
// TAU rows the finalize kernel reads hold the sums of the LAST launch over the particle: fluid <- boundary for a fluid particle,
// a cleared row for a vertex (forcesDevice<PT_VERTEX, PT_FLUID>, whose viscous term never reaches the force, :3750-3783).  This
// kernel therefore accumulates the k-epsilon sums over the boundary elements only; the CPU oracle restates the launches one by one.
// OPEN: a run with open boundaries (laminar): the viscous term of a fluid <- vertex pair and of a boundary element sees the
// relative velocity plus the relative Eulerian velocity (get_viscous_relVel, forces_kernel.def:2494-2507; no normal part is taken
// out for the segment of an open face, :2703-2708), and the gamma CFL term gains compute_gamma_cfl_open_boundary (:1485-1497)
template<bool KEPS, bool OPEN = false>
__global__ void __launch_bounds__(SPHX_BLOCK_FORCES)
sa_forces_kernel(DevParams p, SaForcesArgs a)
{
	__shared__ float sMax[SPHX_BLOCK_FORCES/64], sMaxG[SPHX_BLOCK_FORCES/64], sMaxK[SPHX_BLOCK_FORCES/64];
	const uint32_t index = blockIdx.x*SPHX_BLOCK_FORCES + threadIdx.x + a.fromParticle;
	float cflTerm = 0.0f, gammaCfl = 0.0f, kepsCfl = 0.0f;
	if (index < a.toParticle) {
		const particleinfo info = a.info[index];
		const float4 pos = a.pos[index];
		const bool fluid = PART_TYPE(info) == PT_FLUID;
		if (is_active_w(pos.w)) {
			float4 force = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
			const float4 vel = a.vel[index];
			const uint32_t fl = FLUID_NUM(info);
			float diff_k_out = 0.0f, diff_e_out = 0.0f, ce2yap_out = 1.92f;
			if (fluid) {
				const int3 gridPos = grid_pos_from_hash(p, a.hash[index] & CELLTYPE_BITMASK);
				const float p_rho = (vel.w + 1.0f)*p.rho0[fl];
				// keps_particle_data :633-655, eulerVel_particle_data :553-561; pressure_for_precalc :389-401: P + 2/3 k/rho
				const float p_k = KEPS ? a.tke[index] : 0.0f, p_e = KEPS ? a.eps[index] : 0.0f, p_turb = KEPS ? a.turbvisc[index] : 0.0f;
				const float4 p_euler = (KEPS || OPEN) ? a.eulerVel[index] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
				const float p_precalc = KEPS ? (sa_P(p, vel.w, fl) + 2.0f*p_k/p_rho/3.0f)/(p_rho*p_rho) : sa_P(p, vel.w, fl)/(p_rho*p_rho);
				float diff_k = 0.0f, diff_e = 0.0f, ce2yap = 1.92f, txx = 0.0f, txy = 0.0f, txz = 0.0f, tyy = 0.0f, tyz = 0.0f, tzz = 0.0f;
				const bool density_sum = (p.simflags & SPHX_ENABLE_DENSITY_SUM) != 0;
				const bool newtonian = p.rheology == SPHX_NEWTONIAN;
				// fluid <- fluid and fluid <- vertex: compute_all_pp_interaction with the general specialisations
				auto particle_pair = [&](bool vertexSection, uint32_t j, const float4 &npos, float rx, float ry, float rz) {
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
					const float n_precalc = KEPS ? (sa_P(p, nvel.w, nfl) + 2.0f*a.tke[j]/n_rho/3.0f)/(n_rho*n_rho) : sa_P(p, nvel.w, nfl)/(n_rho*n_rho);
					const float nmass = npos.w;
					if (!density_sum) force.w += nmass*vel_dot_pos*f;
					const float s = (p_precalc + n_precalc)*nmass*f;
					float dx = 0.0f, dy = 0.0f, dz = 0.0f;
					dx -= s*rx; dy -= s*ry; dz -= s*rz;
					if (KEPS && newtonian) {
						// get_visc_coeff :262-270: laminar coefficient + eddy viscosity (fluid particles only: turbViscForViscTerm :645-655),
						// MORRIS along relVel + relEulerVel (get_viscous_relVel :2494-2507)
						const float4 ne = a.eulerVel[j];
						const float wx = vx + (p_euler.x - ne.x), wy = vy + (p_euler.y - ne.y), wz = vz + (p_euler.z - ne.z);
						const float n_tvv = PART_TYPE(a.info[j]) == PT_FLUID ? a.turbvisc[j] : 0.0f;
						const float vf = sa_visc_avg(p, p.visccoeff[fl] + p_turb, p.visccoeff[nfl] + n_tvv, p_rho, n_rho, nmass)*f;
						dx += vf*wx; dy += vf*wy; dz += vf*wz;
					} else
					if (newtonian) {
						float wx = vx, wy = vy, wz = vz;
						if (OPEN && vertexSection) {
							const float4 ne = a.eulerVel[j];
							wx = vx + (p_euler.x - ne.x); wy = vy + (p_euler.y - ne.y); wz = vz + (p_euler.z - ne.z);
						}
						const float vf = sa_visc_avg(p, p.visccoeff[fl], p.visccoeff[nfl], p_rho, n_rho, nmass)*f;
						dx += vf*wx; dy += vf*wy; dz += vf*wz;
					}
					force.x += dx; force.y += dy; force.z += dz;
				};
				const bool finish = a.tiled && !(a.tileGuard && *a.tileGuard);
				if (finish)
					force = a.forces[index];
				else {
					for_each_neib<PT_FLUID>(p, a, index, pos, gridPos, [&](uint32_t j, const float4 &np_, float rx, float ry, float rz) { particle_pair(false, j, np_, rx, ry, rz); });
					for_each_neib<PT_VERTEX>(p, a, index, pos, gridPos, [&](uint32_t j, const float4 &np_, float rx, float ry, float rz) { particle_pair(true, j, np_, rx, ry, rz); });
				}
				if (finish && a.wallDone) {      // sa_forces_wall_kernel: the sums are in `force`, the gamma CFL term in its place
					if (a.cflGamma && a.neibsList[(size_t)p.neibboundpos*p.stride + index] != NEIBS_END) gammaCfl = a.cflGamma[index];
				} else
				// fluid <- boundary element
				for_each_neib<PT_BOUNDARY>(p, a, index, pos, gridPos, [&](uint32_t j, const float4 &npos, float rx, float ry, float rz) {
					if (!is_active_w(npos.w)) return;
					const float r = sqrtf(fmaf(rz, rz, fmaf(ry, ry, rx*rx)));
					if (r >= p.influenceradius + a.deltap) return;
					const float4 nvel = a.vel[j];
					const float vx = vel.x - nvel.x, vy = vel.y - nvel.y, vz = vel.z - nvel.z;
					const uint32_t nfl = FLUID_NUM(a.info[j]);
					const float n_rho = (nvel.w + 1.0f)*p.rho0[nfl];
					const float n_precalc = sa_P(p, nvel.w, nfl)/(n_rho*n_rho);
					const float4 be = a.boundElement[j];
					const V3 ns = v3(be.x, be.y, be.z);
					const float inv_h = 1.0f/p.slength;
					WallTri tri;
					wall_tri_setup(tri, ns, a.vertPos[0][j], a.vertPos[1][j], a.vertPos[2][j], p.slength);
					const float ggamAS = wall_grad_gamma(tri, v3(rx*inv_h, ry*inv_h, rz*inv_h))/p.slength;
					const float vn = sa_dot3(vx, vy, vz, be.x, be.y, be.z);
					if (a.cflGamma) {      // compute_gamma_cfl_solid_wall (:1458-1474): n.(v_a - v_s), n.v_a, n.v_s
						const float va = sa_dot3(vel.x, vel.y, vel.z, be.x, be.y, be.z);
						const float vs = sa_dot3(vel.x - vx, vel.y - vy, vel.z - vz, be.x, be.y, be.z);
						gammaCfl = fmaxf(gammaCfl, ggamAS*fmaxf(fabsf(vn), fmaxf(fabsf(va), fabsf(vs))));
						if (OPEN) {      // n.(v_a + relEulerVel), n.(v_s - relEulerVel)
							const float4 ne = a.eulerVel[j];
							const float ex = (vx + (p_euler.x - ne.x)) - vx, ey = (vy + (p_euler.y - ne.y)) - vy, ez = (vz + (p_euler.z - ne.z)) - vz;
							const float a1 = sa_dot3(vel.x + ex, vel.y + ey, vel.z + ez, be.x, be.y, be.z);
							const float a2 = sa_dot3(-vx + vel.x - ex, -vy + vel.y - ey, -vz + vel.z - ez, be.x, be.y, be.z);
							gammaCfl = fmaxf(gammaCfl, ggamAS*fmaxf(fabsf(a1), fabsf(a2)));
						}
					}
					if (!density_sum) {
						float DrDt = 0.0f;
						DrDt -= p_rho*vn*ggamAS;
						force.w += DrDt;
					}
					const float n_precalc_k = KEPS ? (sa_P(p, nvel.w, nfl) + 2.0f*a.tke[j]/n_rho/3.0f)/(n_rho*n_rho) : n_precalc;
					const float ps = (p_precalc + n_precalc_k)*n_rho*ggamAS;
					float dx = 0.0f, dy = 0.0f, dz = 0.0f;
					dx += ps*be.x; dy += ps*be.y; dz += ps*be.z;
					if (KEPS) {
						const float r_as = fmaxf(fabsf(sa_dot3(rx, ry, rz, be.x, be.y, be.z)), a.deltap);
						const float our_visc = (p.compvisc == SPHX_KINEMATIC) ? p.visccoeff[fl] : p.visccoeff[fl]/p_rho;
						const float neib_visc = (p.compvisc == SPHX_KINEMATIC) ? p.visccoeff[nfl] : p.visccoeff[nfl]/n_rho;
						// compute_turb_visc_contrib, boundary term :2824-2878: wall shear stress from the law of the wall
						if (!(p_k < a.epsilon)) {
							const float ux = vx + p_euler.x, uy = vy + p_euler.y, uz = vz + p_euler.z;
							const float un = sa_dot3(ux, uy, uz, be.x, be.y, be.z);
							const float tx = ux - un*be.x, ty = uy - un*be.y, tz = uz - un*be.z;
							const float abs_u_t = sqrtf(sa_dot3(tx, ty, tz, tx, ty, tz));
							float u_star = 0.0f;
							const float uk = 0.547722558f*sqrtf(p_k);
							float y_plus = r_as/our_visc*uk;
							if (y_plus < 2.43902439f)
								u_star = abs_u_t/y_plus;
							else {
								float utau = 0.118599857f*neib_visc/r_as;
								for (int i = 0; i < 10; i++) {
									y_plus = fmaxf(r_as*utau/neib_visc, 2.43902439f);
									utau = (0.41f*abs_u_t + utau)/(logf(y_plus) + 3.132f);
								}
								u_star = abs_u_t/(logf(y_plus)/0.41f + 5.2f);
							}
							const float sc = 2.0f*ggamAS*u_star*u_star, inv = 1.0f/fmaxf(abs_u_t, 1e-6f);
							dx -= (sc*tx)*inv; dy -= (sc*ty)*inv; dz -= (sc*tz)*inv;
						}
						// compute_keps_term, boundary term :2949-2980
						const float lyap = 0.400772603f*powf(p_k, 1.5f)/(p_e*r_as);
						if (lyap > 1.0f)
							ce2yap = fminf(ce2yap, fmaxf(1.92f - 0.83f*(lyap - 1.0f)*lyap*lyap, 0.0f));
						diff_e += 0.276923077f*p_k*p_k/r_as*ggamAS;
						const float4 ne = a.eulerVel[j];
						const float wx = vx + (p_euler.x - ne.x), wy = vy + (p_euler.y - ne.y), wz = vz + (p_euler.z - ne.z);
						const float mx = (ggamAS*be.x)*n_rho, my = (ggamAS*be.y)*n_rho, mz = (ggamAS*be.z)*n_rho;
						txx += wx*mx; txy += wx*my + wy*mx; txz += wx*mz + wz*mx;      // add_strain_rate :924-937
						tyy += wy*my; tyz += wy*mz + wz*my; tzz += wz*mz;
					} else
					if (newtonian) {
						const float r_as = fmaxf(fabsf(sa_dot3(rx, ry, rz, be.x, be.y, be.z)), a.deltap);
						float tx, ty, tz;
						if (OPEN) {
							const float4 ne = a.eulerVel[j];
							const float wx = vx + (p_euler.x - ne.x), wy = vy + (p_euler.y - ne.y), wz = vz + (p_euler.z - ne.z);
							const float wn = SA_IS_OPEN(a.info[j]) ? 0.0f : sa_dot3(wx, wy, wz, be.x, be.y, be.z);
							tx = wx - wn*be.x; ty = wy - wn*be.y; tz = wz - wn*be.z;
						} else { tx = vx - vn*be.x; ty = vy - vn*be.y; tz = vz - vn*be.z; }
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
				// forces_fixup, gravity, CFL term
				const float gam = a.gGam[index].w;
				force.x /= gam; force.y /= gam; force.z /= gam; force.w /= gam;
				force.w /= p.rho0[fl];
				if (KEPS) {
					// viscous_fixup with KEPSILON + SA_BOUNDARY :3123-3170
					const float rhoGam = p_rho*gam;
					diff_k /= rhoGam; diff_e /= rhoGam;
					float SijSij_bytwo = 2.0f*(txx*txx + tyy*tyy + tzz*tzz) + txy*txy + txz*txz + tyz*tyz;
					const float S = sqrtf(SijSij_bytwo)/rhoGam;
					SijSij_bytwo /= rhoGam*rhoGam;
					const float Pturb = fminf(p_turb*SijSij_bytwo, 0.3f*p_k*S);
					diff_k += Pturb;
					diff_e += p_e*1.44f*Pturb/p_k;
					kepsCfl = p_turb;             // dyndt_keps_shared_data :3481-3501
					diff_k_out = diff_k; diff_e_out = diff_e; ce2yap_out = ce2yap;
				}
				force.x += p.gravity[0]; force.y += p.gravity[1]; force.z += p.gravity[2];
				if (p.simflags & SPHX_ENABLE_DTADAPT) {
					const float sspeed = sa_sound_speed(p, vel.w, fl);
					const float acc = sqrtf(fmaf(force.z, force.z, fmaf(force.y, force.y, force.x*force.x)));
					cflTerm = fmaxf(acc, sspeed*sspeed/p.slength);
				}
			}
			a.forces[index] = force;
			if (KEPS && (fluid || PART_TYPE(info) == PT_VERTEX)) {      // write_keps :3331-3339 (a fresh output for the vertices)
				float *d = a.dkde + 3*(size_t)index;
				if (fluid) { d[0] = diff_k_out; d[1] = diff_e_out; d[2] = ce2yap_out; }
				else { d[0] = 0.0f; d[1] = 0.0f; d[2] = 1.92f; }
			}
		}
		if (a.cflGamma) a.cflGamma[index] = gammaCfl;
	}
	// one CFL entry per block of SPHX_BLOCK_FORCES particles (maxBlockReduce)
#pragma unroll
	for (int d = 32; d > 0; d >>= 1) {
		cflTerm = fmaxf(cflTerm, __shfl_down(cflTerm, d));
		gammaCfl = fmaxf(gammaCfl, __shfl_down(gammaCfl, d));
		if (KEPS) kepsCfl = fmaxf(kepsCfl, __shfl_down(kepsCfl, d));
	}
	if ((threadIdx.x & 63u) == 0u) { sMax[threadIdx.x >> 6] = cflTerm; sMaxG[threadIdx.x >> 6] = gammaCfl; sMaxK[threadIdx.x >> 6] = kepsCfl; }
	__syncthreads();
	if (threadIdx.x == 0 && a.cfl && (p.simflags & SPHX_ENABLE_DTADAPT)) {
		float m = sMax[0], mg = sMaxG[0], mk = sMaxK[0];
		for (int w = 1; w < SPHX_BLOCK_FORCES/64; ++w) { m = fmaxf(m, sMax[w]); mg = fmaxf(mg, sMaxG[w]); mk = fmaxf(mk, sMaxK[w]); }
		a.cfl[a.cflOffset + blockIdx.x] = m;
		if (a.cflGammaBlocks) a.cflGammaBlocks[a.cflOffset + blockIdx.x] = mg;
		if (KEPS && a.cflKeps) a.cflKeps[a.cflOffset + blockIdx.x] = mk;
	}
}

// repackDevice x3 + finalizeRepackDevice with SA_BOUNDARY (src/cuda/forces_kernel.def:3024-3086,4155-4349; run_repack
// src/cuda/forces.cu:828-896): the mixing force of the repacking run mode on the fluid particles -- a c0^2 V_b F r over the fluid
// neighbours, a c0 V_b F r over the vertex neighbours (one c0: the reference's expression), + a c0^2 |grad gamma_as| n_s over
// the boundary elements; sums divided by gamma; velocity damping and the CFL term as without SA
__global__ void __launch_bounds__(SPHX_BLOCK_FORCES)
sa_repack_kernel(DevParams p, SaForcesArgs a)
{
	__shared__ float sMax[SPHX_BLOCK_FORCES/64];
	const uint32_t index = blockIdx.x*SPHX_BLOCK_FORCES + threadIdx.x + a.fromParticle;
	float cflTerm = 0.0f;
	if (index < a.toParticle) {
		const particleinfo info = a.info[index];
		const float4 pos = a.pos[index];
		if (is_active_w(pos.w)) {
			float4 force = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
			if (PART_TYPE(info) == PT_FLUID) {
				const float4 vel = a.vel[index];
				const uint32_t fl = FLUID_NUM(info);
				const int3 gridPos = grid_pos_from_hash(p, a.hash[index] & CELLTYPE_BITMASK);
				const float c0 = p.sscoeff[fl];
				auto volumic = [&](uint32_t j, const float4 &npos, float rx, float ry, float rz, bool vertex) {
					if (!is_active_w(npos.w)) return;
					const float r = sqrtf(fmaf(rz, rz, fmaf(ry, ry, rx*rx)));
					if (r >= p.influenceradius) return;
					const float n_rho = (a.vel[j].w + 1.0f)*p.rho0[FLUID_NUM(a.info[j])];
					const float qm2 = r/p.slength - 2.0f;
					const float f = qm2*qm2*qm2*p.fcoeff;
					const float s = vertex ? p.repack_a*c0*npos.w/n_rho*f : p.repack_a*c0*c0*npos.w/n_rho*f;
					force.x -= s*rx; force.y -= s*ry; force.z -= s*rz;
				};
				for_each_neib<PT_FLUID>(p, a, index, pos, gridPos,
					[&](uint32_t j, const float4 &npos, float rx, float ry, float rz) { volumic(j, npos, rx, ry, rz, false); });
				for_each_neib<PT_VERTEX>(p, a, index, pos, gridPos,
					[&](uint32_t j, const float4 &npos, float rx, float ry, float rz) { volumic(j, npos, rx, ry, rz, true); });
				for_each_neib<PT_BOUNDARY>(p, a, index, pos, gridPos, [&](uint32_t j, const float4 &npos, float rx, float ry, float rz) {
					if (!is_active_w(npos.w)) return;
					const float r = sqrtf(fmaf(rz, rz, fmaf(ry, ry, rx*rx)));
					if (r >= p.influenceradius + a.deltap) return;
					const float4 be = a.boundElement[j];
					const V3 ns = v3(be.x, be.y, be.z);
					const float inv_h = 1.0f/p.slength;
					WallTri tri;
					wall_tri_setup(tri, ns, a.vertPos[0][j], a.vertPos[1][j], a.vertPos[2][j], p.slength);
					const float ggamAS = wall_grad_gamma(tri, v3(rx*inv_h, ry*inv_h, rz*inv_h))/p.slength;
					const float c = p.repack_a*c0*c0*ggamAS;
					force.x += c*be.x; force.y += c*be.y; force.z += c*be.z;
				});
				const float gam = a.gGam[index].w;      // repack_fixup :3220-3236
				force.x /= gam; force.y /= gam; force.z /= gam; force.w /= gam;
				force.w /= p.rho0[fl];
				const float damp = p.repack_alpha*c0/a.deltap;
				force.x += damp*vel.x; force.y += damp*vel.y; force.z += damp*vel.z;
				if (p.simflags & SPHX_ENABLE_DTADAPT) {
					const float sspeed = sa_sound_speed(p, vel.w, fl);
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

// integrateGammaDevice, quadrature flavour (src/cuda/density_sum_kernel.cu:690-765) for fluid particles at their new positions;
// the rows of the other particle types are copied (copyTypeDataDevice, src/cuda/euler.cu:253-262)

__global__ void __launch_bounds__(128)
sa_integrate_gamma_kernel(DevParams p, SaIntGammaArgs a)
{
	const uint32_t index = blockIdx.x*128 + threadIdx.x;
	if (index >= a.numParticles) return;
	const particleinfo info = a.info[index];
	const float4 og = a.oldGGam[index];
	const bool vertexRow = a.vertexRows && PART_TYPE(info) == PT_VERTEX;      // moving bodies: Gamma<PT_VERTEX> against the new elements
	if (PART_TYPE(info) != PT_FLUID && !vertexRow) {
		if (PART_TYPE(info) == PT_VERTEX || PART_TYPE(info) == PT_BOUNDARY) a.newGGam[index] = og;
		return;
	}
	if (a.wallDone) {      // the particles near a wall are sa_integrate_gamma_wall_kernel's
		if (a.neibsList[(size_t)p.neibboundpos*p.stride + index] == NEIBS_END || !is_active_w(a.pos[index].w))
			a.newGGam[index] = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
		return;
	}
	const float4 pos = a.pos[index];
	const int3 gridPos = grid_pos_from_hash(p, a.hash[index] & CELLTYPE_BITMASK);
	float4 g = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
	const V3 oldg = v3(og.x, og.y, og.z);
	for_each_neib<PT_BOUNDARY>(p, a, index, pos, gridPos, [&](uint32_t j, const float4 &, float rx, float ry, float rz) {
		const float4 be = a.boundElement[j];
		const V3 normal = v3(be.x, be.y, be.z);
		const V3 q = v3(rx, ry, rz)/p.slength;
		WallTri tri;
		wall_tri_setup(tri, normal, a.vertPos[0][j], a.vertPos[1][j], a.vertPos[2][j], p.slength);
		const float ggamAS = wall_grad_gamma(tri, q)/p.slength;
		g.x += ggamAS*be.x; g.y += ggamAS*be.y; g.z += ggamAS*be.z;
		g.w -= vertexRow ? wall_gamma<true>(tri, q, oldg, p.slength, a.epsilon) : wall_gamma<false>(tri, q, oldg, p.slength, a.epsilon);
	});
	a.newGGam[index] = g;
}

// ---- density summation with dynamic gamma (src/cuda/density_sum_kernel.cu:206-250,419-478,523-655) and Brezzi diffusion ----
// OPEN: a run with open boundaries.  A vertex of an open face is not a reservoir of mass at rest: its term at step n is dropped
// and replaced by the kernel at the distance it would have after moving with its Eulerian velocity instead of its own for dt
// (densitySumOpenBoundaryContribution, density_sum_kernel.cu:119-140); the segments of an open face add to the gamma that
// multiplies the old density the flux of gamma through them (io_gamma_contrib / compute_imposed_gamma, :374-417)
// MOVING: ENABLE_MOVING_BODIES.  The elements are where they were AND where they are: the normal of step n comes from
// boundElement, that of the new state from boundElementNew (the Euler step turned it, sphx_sa_update_normals), the corners are
// set up once per normal; gamma of the VERTEX rows is integrated by the same boundary terms instead of copied
// (integrateGammaDevice<PT_VERTEX>, density_sum_impl src/cuda/euler.cu:112-160); boundary rows are left to the segment condition
template<bool OPEN, bool MOVING = false>
__global__ void __launch_bounds__(128)
sa_density_sum_kernel(DevParams p, SaDensitySumArgs a)
{
	const uint32_t index = blockIdx.x*128 + threadIdx.x;
	if (index >= a.numParticles) return;
	const particleinfo info = a.info[index];
	const bool vertexRow = MOVING && PART_TYPE(info) == PT_VERTEX;
	if (PART_TYPE(info) != PT_FLUID && !vertexRow) {
		// MOVING: the reference leaves the BOUNDARY rows of the write buffer as they are (whatever an older step left there) and
		// relies on the segment condition, which re-derives gamma of every segment in each step of such a run
		// (sa_segment_bc_kernel, has_moving); here they are copied, as sa_integrate_gamma_kernel does, so that the rows between the
		// two commands are a function of the inputs (tests/test_sa_moving.py holds both kernels to it)
		if (PART_TYPE(info) == PT_BOUNDARY || (!MOVING && PART_TYPE(info) == PT_VERTEX)) a.newGGam[index] = a.oldGGam[index];
		return;
	}
	const float4 posN = a.oldPos[index], posNp1 = a.pos[index];
	const int3 gridPos = grid_pos_from_hash(p, a.hash[index] & CELLTYPE_BITMASK);
	const float dx = posNp1.x - posN.x, dy = posNp1.y - posN.y, dz = posNp1.z - posN.z;
	// the walker hands over r_ab at step n (own old position against a.pos = new neighbour rows is NOT what is wanted), so
	// it is pointed at the OLD positions and the new neighbour row is fetched here
	struct { const float4 *pos; const uint32_t *cellStart; const neibdata *neibsList; } w = { a.oldPos, a.cellStart, a.neibsList };
	float sumPmwN = 0.0f, sumPmwNp1 = 0.0f, sumOpen = 0.0f;
	auto volumic = [&](uint32_t j, const float4 &nN, float pcx, float pcy, float pcz) {
		if (!is_active_w(nN.w)) return;
		const float4 nNp1 = a.pos[j];
		// r_ab at n = pos_corr - neighbour old; at n+1 = (pos_corr - neighbour new) + own displacement
		const float rx = pcx - nN.x, ry = pcy - nN.y, rz = pcz - nN.z;
		const float qx = (pcx - nNp1.x) + dx, qy = (pcy - nNp1.y) + dy, qz = (pcz - nNp1.z) + dz;
		const bool openNeib = OPEN && SA_IS_OPEN(a.info[j]);
		if (!openNeib) {
			const float rN = sqrtf(rx*rx + ry*ry + rz*rz);
			sumPmwN -= nN.w*kernel_W<SPHX_WENDLAND>(p, rN);
		}
		const float rNp1 = sqrtf(qx*qx + qy*qy + qz*qz);
		if (rNp1 < p.influenceradius) sumPmwNp1 += nN.w*kernel_W<SPHX_WENDLAND>(p, rNp1);
		if (openNeib) {
			const float4 e = a.oldEulerVel[j], v = a.oldVel[j];
			const float ex = rx + a.dt*(e.x - v.x), ey = ry + a.dt*(e.y - v.y), ez = rz + a.dt*(e.z - v.z);
			const float moved = sqrtf(ex*ex + ey*ey + ez*ez);
			if (moved < p.influenceradius) sumOpen -= nN.w*kernel_W<SPHX_WENDLAND>(p, moved);
		}
	};
	float fw = 0.0f;
	if (vertexRow) { }      // a vertex has no density to sum
	else if (a.tiled && !(a.tileGuard && *a.tileGuard))
		fw = a.forces[index].w;
	else {
		for_each_neib<PT_FLUID, true>(p, w, index, posN, gridPos, volumic);
		for_each_neib<PT_VERTEX, true>(p, w, index, posN, gridPos, volumic);
		fw = sumPmwNp1 + sumPmwN + sumOpen;
		a.forces[index].w = fw;
	}
	float gGamDotR = 0.0f, gamFluxMoved = 0.0f, gamFluxN = 0.0f;
	V3 gGam = v3(0.0f, 0.0f, 0.0f);
	// sa_density_sum_wall_kernel; with MOVING sa_density_sum_wall_moving_kernel: wallDone & 1 the fluid rows, & 2 the vertex rows
	if ((vertexRow ? (a.wallDone & 2) : (a.wallDone & 1)) && a.tiled && !(a.tileGuard && *a.tileGuard)) {
		if (a.neibsList[(size_t)p.neibboundpos*p.stride + index] != NEIBS_END && is_active_w(posN.w)) {
			const float4 t = a.newGGam[index];
			gGam = v3(t.x, t.y, t.z); gGamDotR = t.w;
			// sa_density_sum_wall_kernel<true>: half the sum of the two fluxes of gamma through the open segments, left where the new
			// density is about to be written
			if (OPEN && !vertexRow) gamFluxMoved = 2.0f*a.newVel[index].w;
		}
	} else
	for_each_neib<PT_BOUNDARY, true>(p, w, index, posN, gridPos, [&](uint32_t j, const float4 &nN, float pcx, float pcy, float pcz) {
		if (!is_active_w(nN.w)) return;
		const float4 nNp1 = a.pos[j];
		const float inv = 1.0f/p.slength;
		const V3 qN = v3((pcx - nN.x)*inv, (pcy - nN.y)*inv, (pcz - nN.z)*inv);
		const V3 qNp1 = v3(((pcx - nNp1.x) + dx)*inv, ((pcy - nNp1.y) + dy)*inv, ((pcz - nNp1.z) + dz)*inv);
		const float4 be = a.boundElement[j];
		const V3 ns = v3(be.x, be.y, be.z);
		WallTri tri;       // one set-up, two positions
		wall_tri_setup(tri, ns, a.vertPos[0][j], a.vertPos[1][j], a.vertPos[2][j], p.slength);
		const V3 gN = ns*(wall_grad_gamma(tri, qN)/p.slength);
		// the flux of gamma through an open segment (io_gamma_contrib, :374-397): the virtual displacement dt (u_E - u) at step n,
		// against the element as it is at step n.  (With ENABLE_MOVING_BODIES as well -- CompleteSaExample.cu's option set -- the
		// reference hands this term the corners set up for the NEW normal together with the OLD normal, and says so: "TODO check
		// if we need the old or the new normal here, in case of moving open boundaries (for fixed open boundaries, it makes no
		// difference)", :470-476.  Open faces that move are in none of its problems; for the fixed ones the two set-ups are the
		// same numbers, and here the term is evaluated before the corners are set up again, i.e. wholly at step n.)
		if (OPEN && SA_IS_OPEN(a.info[j])) {
			const float4 e = a.oldEulerVel[j], v = a.oldVel[j];
			const V3 drift = v3(a.dt*(e.x - v.x), a.dt*(e.y - v.y), a.dt*(e.z - v.z));
			const V3 gMoved = ns*(wall_grad_gamma(tri, qN + drift/p.slength)/p.slength);
			gamFluxMoved += dot(drift, gMoved);
			gamFluxN += dot(drift, gN);
		}
		V3 nsNew = ns;
		if (MOVING) {      // ... unless the element turned: its corners in the frame of the new normal
			const float4 ben = a.boundElementNew[j];
			nsNew = v3(ben.x, ben.y, ben.z);
			wall_tri_setup(tri, nsNew, a.vertPos[0][j], a.vertPos[1][j], a.vertPos[2][j], p.slength);
		}
		const V3 gNp1 = nsNew*(wall_grad_gamma(tri, qNp1)/p.slength);
		gGamDotR += 0.5f*dot(gN + gNp1, qNp1 - qN);
		gGam = gGam + gNp1;
	});
	gGamDotR *= p.slength;
	const float4 gGamN = a.oldGGam[index];
	float4 g = make_float4(gGam.x, gGam.y, gGam.z, gGamN.w + gGamDotR);
	if (vertexRow) { a.newGGam[index] = g; return; }      // integrateGammaDeviceFunc with dynamic gamma (:669-685): no clamp
	const uint32_t fl = FLUID_NUM(info);
	float gamOld = gGamN.w;
	if (OPEN) {
		gamOld = gGamN.w + (gamFluxMoved + gamFluxN)/2.0f;
		gamOld = gamOld > 1.0f ? 1.0f : (gamOld < 0.1f ? 0.1f : gamOld);
	}
	const float rho = (gamOld*((a.oldVel[index].w + 1.0f)*p.rho0[fl]) + fw)/g.w;
	if (g.w > 1.0f || sqrtf(g.x*g.x + g.y*g.y + g.z*g.z)*p.slength < 1e-10f) g.w = 1.0f;
	else if (g.w < 0.1f) g.w = 0.1f;
	a.newVel[index].w = rho/p.rho0[fl] - 1.0f;
	a.newGGam[index] = g;
}


// computeDensityDiffusionDevice<.., BREZZI, SA_BOUNDARY, PT_FLUID> (forces_kernel.def:1766-1783, 4515-4560)
// OPEN: with ENABLE_INLET_OUTLET the segments of PRESSURE-driven open faces exchange density with the fluid as a fluid neighbour
// would, with |grad gamma_as| / r_as in the place of V_b F, no diffusion coefficient, evaluated in double as the literals of the
// reference's expression make it (:1836-1852, 4536-4582)
template<bool OPEN>
__global__ void __launch_bounds__(128)
sa_density_diffusion_kernel(DevParams p, SaDiffusionArgs a)
{
	if (a.tileGuard && !*a.tileGuard) return;
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
	if (OPEN)
	for_each_neib<PT_BOUNDARY>(p, a, index, pos, gridPos, [&](uint32_t j, const float4 &npos, float rx, float ry, float rz) {
		const float r = sqrtf(rx*rx + ry*ry + rz*rz);
		if (!is_active_w(npos.w)) return;
		if (r >= p.influenceradius + a.deltap) return;
		const particleinfo ninfo = a.info[j];
		if (!SA_IS_OPEN(ninfo) || SA_IS_VELOCITY_DRIVEN(ninfo)) return;
		const float4 be = a.boundElement[j];
		const float r_as = fmaxf(fabsf(sa_dot3(rx, ry, rz, be.x, be.y, be.z)), a.deltap);
		const float inv_h = 1.0f/p.slength;
		WallTri tri;
		wall_tri_setup(tri, v3(be.x, be.y, be.z), a.vertPos[0][j], a.vertPos[1][j], a.vertPos[2][j], p.slength);
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

// updateDensityDevice (src/cuda/euler_kernel.cu:116-136)
__global__ void __launch_bounds__(256)
sa_update_density_kernel(float4 *vel, const float4 *forces, const particleinfo *info, uint32_t numParticles, float dt)
{
	const uint32_t index = blockIdx.x*256 + threadIdx.x;
	if (index >= numParticles) return;
	if (PART_TYPE(info[index]) != PT_FLUID) return;
	const float rho = vel[index].w;
	const float delta = forces[index].w*dt;
	vel[index].w = rho + delta;
}

// gamma part of dtreduce with dynamic gamma (src/cuda/forces.cu:576-585), on the device scalar the step reads its dt from
__global__ void __launch_bounds__(256)
sa_gamma_dt_kernel(float *d_dt, const float *cflGammaBlocks, uint32_t numBlocks)
{
	__shared__ float s[256];
	float m = 0.0f;
	for (uint32_t i = threadIdx.x; i < numBlocks; i += 256) m = fmaxf(m, cflGammaBlocks[i]);
	s[threadIdx.x] = m;
	__syncthreads();
	for (int d = 128; d > 0; d >>= 1) { if ((int)threadIdx.x < d) s[threadIdx.x] = fmaxf(s[threadIdx.x], s[threadIdx.x + d]); __syncthreads(); }
	if (threadIdx.x == 0) {
		const float dt = d_dt[0];
		const float maxcfl = fmaxf(s[0], 1e-5f/dt);
		const float dt_gam = 0.001f/maxcfl;
		if (dt_gam < dt) d_dt[0] = dt_gam;
	}
}

static int sa_check(sphx_ctx *ctx, const char *who)
{
	if (!ctx || !ctx->have_params) return sphx_set_error(SPHX_ERR_INVALID, "sphx_sa: constants not set");
	if (ctx->params.boundarytype != SPHX_SA_BOUNDARY)
		return sphx_set_error(SPHX_ERR_INVALID, who);      // the reference throws "... called without SA_BOUNDARY"
	if (ctx->params.kerneltype != SPHX_WENDLAND)
		return sphx_set_error(SPHX_ERR_UNSUPPORTED, "sphx_sa: SA_BOUNDARY is built for the Wendland kernel (as the reference, src/cuda/gamma.cuh:241-250)");
	// Open boundaries (ENABLE_INLET_OUTLET): the entry points such a run calls besides those of sa_io.hip (vertex normals, initial
	// gamma) do what it needs; the passes of sa_io.hip ran against the oracle on an MI355X in round 5 (tests/test_gpu_sa_io.py)
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

static int sa_segment_bc_impl(sphx_ctx *ctx, void *vel, void *gGam, float *tke, float *eps, void *eulerVel,
	const void *pos, const void *vertices,
	const void *boundElements, const void *info, const uint32_t *hash, const uint32_t *cellStart, const uint16_t *neibsList,
	uint32_t numParticles, uint32_t particleRangeEnd, float deltap, float slength, float influenceradius,
	int step, int run_mode, void *stream)
{
	(void)numParticles;
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
	if (ctx->sa_wall_neibslist == neibsList && particleRangeEnd <= ctx->sa_rows_range) a.rows = ctx->sa_rows_bound;
	if (tke) {
		a.tke = tke; a.eps = eps; a.eulerVel = (float4*)eulerVel; a.deltap = deltap;
		sa_segment_bc_kernel<SPHX_WENDLAND, true><<<div_up_u(particleRangeEnd, 128), 128, 0, (hipStream_t)stream>>>(ctx->dev, a);
	} else
		sa_segment_bc_kernel<SPHX_WENDLAND, false><<<div_up_u(particleRangeEnd, 128), 128, 0, (hipStream_t)stream>>>(ctx->dev, a);
	SPHX_LAUNCH_CHECK("sa_segment_bc_kernel");
	return SPHX_OK;
}

extern "C" int sphx_sa_segment_bc(sphx_ctx *ctx, void *vel, void *gGam, const void *pos, const void *vertices,
	const void *boundElements, const void *info, const uint32_t *hash, const uint32_t *cellStart, const uint16_t *neibsList,
	uint32_t numParticles, uint32_t particleRangeEnd, float deltap, float slength, float influenceradius,
	int step, int run_mode, void *stream)
{
	if (ctx && ctx->params.turbmodel == SPHX_KEPSILON && run_mode == SPHX_SIMULATE)
		return sphx_set_error(SPHX_ERR_INVALID, "sphx_sa_segment_bc: KEPSILON evolves k, epsilon and the Eulerian velocity of the walls: call sphx_sa_segment_bc_keps");
	return sa_segment_bc_impl(ctx, vel, gGam, nullptr, nullptr, nullptr, pos, vertices, boundElements, info, hash, cellStart, neibsList,
		numParticles, particleRangeEnd, deltap, slength, influenceradius, step, run_mode, stream);
}

extern "C" int sphx_sa_segment_bc_keps(sphx_ctx *ctx, void *vel, void *gGam, float *tke, float *eps, void *eulerVel,
	const void *pos, const void *vertices,
	const void *boundElements, const void *info, const uint32_t *hash, const uint32_t *cellStart, const uint16_t *neibsList,
	uint32_t numParticles, uint32_t particleRangeEnd, float deltap, float slength, float influenceradius,
	int step, int run_mode, void *stream)
{
	SPHX_REQUIRE(ctx && ctx->params.turbmodel == SPHX_KEPSILON, "sphx_sa_segment_bc_keps: the uploaded option set is not KEPSILON");
	SPHX_REQUIRE(tke && eps && eulerVel, "sphx_sa_segment_bc_keps: missing buffer");
	SPHX_REQUIRE(run_mode == SPHX_SIMULATE, "sphx_sa_segment_bc_keps: the repacking run mode has no k-epsilon (call sphx_sa_segment_bc)");
	return sa_segment_bc_impl(ctx, vel, gGam, tke, eps, eulerVel, pos, vertices, boundElements, info, hash, cellStart, neibsList,
		numParticles, particleRangeEnd, deltap, slength, influenceradius, step, run_mode, stream);
}

static int sa_vertex_bc_impl(sphx_ctx *ctx, void *vel, const void *gGam, float *tke, float *eps, void *eulerVel,
	const void *vertices, const void *boundElements, const void *pos, const void *info,
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
	if (ctx->sa_wall_neibslist == neibsList && particleRangeEnd <= ctx->sa_rows_range) a.rows = ctx->sa_rows_vert;
	if (tke) {
		a.tke = tke; a.eps = eps; a.eulerVel = (float4*)eulerVel;
		a.vertices = (const uint4*)vertices; a.boundElement = (float4*)const_cast<void*>(boundElements);
		sa_vertex_bc_kernel<SPHX_WENDLAND, true><<<div_up_u(particleRangeEnd, 128), 128, 0, (hipStream_t)stream>>>(ctx->dev, a);
	} else
		sa_vertex_bc_kernel<SPHX_WENDLAND, false><<<div_up_u(particleRangeEnd, 128), 128, 0, (hipStream_t)stream>>>(ctx->dev, a);
	SPHX_LAUNCH_CHECK("sa_vertex_bc_kernel");
	return SPHX_OK;
}

extern "C" int sphx_sa_vertex_bc(sphx_ctx *ctx, void *vel, const void *gGam, const void *pos, const void *info,
	const uint32_t *hash, const uint32_t *cellStart, const uint16_t *neibsList,
	uint32_t numParticles, uint32_t particleRangeEnd, float deltap, float slength, float influenceradius,
	int step, int run_mode, void *stream)
{
	if (ctx && ctx->params.turbmodel == SPHX_KEPSILON && run_mode == SPHX_SIMULATE)
		return sphx_set_error(SPHX_ERR_INVALID, "sphx_sa_vertex_bc: KEPSILON evolves k and epsilon of the vertices: call sphx_sa_vertex_bc_keps");
	return sa_vertex_bc_impl(ctx, vel, gGam, nullptr, nullptr, nullptr, nullptr, nullptr, pos, info, hash, cellStart, neibsList,
		numParticles, particleRangeEnd, deltap, slength, influenceradius, step, run_mode, stream);
}

extern "C" int sphx_sa_vertex_bc_keps(sphx_ctx *ctx, void *vel, const void *gGam, float *tke, float *eps, void *eulerVel,
	const void *vertices, const void *boundElements, const void *pos, const void *info,
	const uint32_t *hash, const uint32_t *cellStart, const uint16_t *neibsList,
	uint32_t numParticles, uint32_t particleRangeEnd, float deltap, float slength, float influenceradius,
	int step, int run_mode, void *stream)
{
	SPHX_REQUIRE(ctx && ctx->params.turbmodel == SPHX_KEPSILON, "sphx_sa_vertex_bc_keps: the uploaded option set is not KEPSILON");
	SPHX_REQUIRE(tke && eps && eulerVel && vertices && boundElements, "sphx_sa_vertex_bc_keps: missing buffer");
	SPHX_REQUIRE(run_mode == SPHX_SIMULATE, "sphx_sa_vertex_bc_keps: the repacking run mode has no k-epsilon (call sphx_sa_vertex_bc)");
	return sa_vertex_bc_impl(ctx, vel, gGam, tke, eps, eulerVel, vertices, boundElements, pos, info, hash, cellStart, neibsList,
		numParticles, particleRangeEnd, deltap, slength, influenceradius, step, run_mode, stream);
}

extern "C" int sphx_sa_init_gamma(sphx_ctx *ctx, void *newGGam, const void *oldGGam, const void *pos, const void *boundElements,
	const void *vertPos0, const void *vertPos1, const void *vertPos2, const void *info,
	const uint32_t *hash, const uint32_t *cellStart, const uint16_t *neibsList,
	float slength, float influenceradius, float deltap, float epsilon,
	uint32_t numParticles, uint32_t particleRangeEnd, void *stream)
{
	(void)numParticles; (void)oldGGam;      // the reference's kernel does not read the old values either
	int rc = sa_check(ctx, "saInitGamma called without SA_BOUNDARY");
	if (rc != SPHX_OK) return rc;
	SPHX_REQUIRE(newGGam && pos && boundElements && vertPos0 && vertPos1 && vertPos2 && info && hash && cellStart && neibsList,
		"sphx_sa_init_gamma: missing buffer");
	SPHX_REQUIRE(slength == ctx->params.slength && influenceradius == ctx->params.influenceradius,
		"sphx_sa_init_gamma: slength / influenceradius differ from the uploaded constants");
	if (!particleRangeEnd) return SPHX_OK;
	SaGammaArgs a = {};
	a.newGGam = (float4*)newGGam; a.pos = (const float4*)pos; a.boundElement = (const float4*)boundElements;
	a.vertPos[0] = (const float2*)vertPos0; a.vertPos[1] = (const float2*)vertPos1; a.vertPos[2] = (const float2*)vertPos2;
	a.info = (const particleinfo*)info; a.hash = hash; a.cellStart = cellStart; a.neibsList = neibsList;
	a.numParticles = particleRangeEnd; a.deltap = deltap; a.epsilon = epsilon;
	hipStream_t st = (hipStream_t)stream;
	// fluid particles, then vertex particles (saInitGamma, src/cuda/boundary_conditions.cu:497-535)
	sa_init_gamma_kernel<PT_FLUID><<<div_up_u(particleRangeEnd, 128), 128, 0, st>>>(ctx->dev, a);
	SPHX_LAUNCH_CHECK("sa_init_gamma_kernel<PT_FLUID>");
	sa_init_gamma_kernel<PT_VERTEX><<<div_up_u(particleRangeEnd, 128), 128, 0, st>>>(ctx->dev, a);
	SPHX_LAUNCH_CHECK("sa_init_gamma_kernel<PT_VERTEX>");
	return SPHX_OK;
}

static int sa_forces_check(sphx_ctx *ctx, const char *who)
{
	int rc = sa_check(ctx, who);
	if (rc != SPHX_OK) return rc;
	const sphx_params &q = ctx->params;
	// two forms are built: the continuity equation with gamma by quadrature (StillWaterRepackSA) and density summation with
	// dynamic gamma and, optionally, Brezzi diffusion (StillWaterSA and most SA problems of the reference)
	const bool dsum = (q.simflags & SPHX_ENABLE_DENSITY_SUM) != 0, quad = (q.simflags & SPHX_ENABLE_GAMMA_QUADRATURE) != 0;
	if (dsum == quad)
		return sphx_set_error(SPHX_ERR_UNSUPPORTED, "sphx_sa: built are ENABLE_DENSITY_SUM with dynamic gamma, and the continuity equation with ENABLE_GAMMA_QUADRATURE");
	// ENABLE_MOVING_BODIES: bodies with prescribed motion (segments and vertices flagged FG_MOVING_BOUNDARY); the pair terms read the
	// elements' velocities from BUFFER_VEL as they do for walls at rest.  Bodies that FEEL the fluid (FG_COMPUTE_FORCE rows of
	// BUFFER_RB_FORCES with SA_BOUNDARY): the pressure force on their elements is sphx_sa_body_pressure_forces, below.
	if (q.simflags & (SPHX_ENABLE_XSPH | SPHX_ENABLE_PLANES))
		return sphx_set_error(SPHX_ERR_UNSUPPORTED, "sphx_sa: SA forces are built without XSPH and planes");
	if (q.sph_formulation != SPHX_SPH_F1)
		return sphx_set_error(SPHX_ERR_UNSUPPORTED, "sphx_sa: SA forces are built for SPH_F1");
	if (!(q.densitydiffusiontype == SPHX_DENSITY_DIFFUSION_NONE || (dsum && q.densitydiffusiontype == SPHX_BREZZI)))
		return sphx_set_error(SPHX_ERR_UNSUPPORTED, "sphx_sa: density diffusion with SA_BOUNDARY: Brezzi with density summation only");
	if (q.turbmodel != SPHX_LAMINAR_FLOW && q.turbmodel != SPHX_KEPSILON && q.rheologytype != SPHX_INVISCID)
		return sphx_set_error(SPHX_ERR_UNSUPPORTED, "sphx_sa: SA forces are built for laminar flow and the k-epsilon model");
	if (q.turbmodel == SPHX_KEPSILON && (q.rheologytype != SPHX_NEWTONIAN || q.viscmodel != SPHX_MORRIS))
		return sphx_set_error(SPHX_ERR_UNSUPPORTED, "sphx_sa: KEPSILON is built for a Newtonian fluid with the MORRIS viscous model");
	return SPHX_OK;
}

struct SaKepsBuffers { float *cflKeps, *dkde; const float *tke, *eps, *turbvisc; const void *eulerVel; float epsilon; };

static int sa_forces_impl(sphx_ctx *ctx, void *forces, float *cfl, float *cflGamma, const SaKepsBuffers *ke,
	const void *pos, const void *vel, const void *info, const uint32_t *hash, const uint32_t *cellStart, const uint16_t *neibsList,
	const void *gGam, const void *boundElements, const void *vertPos0, const void *vertPos1, const void *vertPos2,
	uint32_t numParticles, uint32_t fromParticle, uint32_t toParticle,
	float deltap, float slength, float dtadaptfactor, float influenceradius, uint32_t cflOffset,
	int run_mode, int step, float dt, uint32_t *h_numBlocks, void *stream)
{
	(void)dtadaptfactor; (void)step; (void)dt;
	int rc = sa_forces_check(ctx, "forces basicstep (SA) called without SA_BOUNDARY");
	if (rc != SPHX_OK) return rc;
	if (run_mode != SPHX_SIMULATE && run_mode != SPHX_REPACK)
		return sphx_set_error(SPHX_ERR_INVALID, "sphx_forces_basicstep_sa: invalid run mode");
	if (run_mode == SPHX_REPACK && !(ctx->params.simflags & SPHX_ENABLE_GAMMA_QUADRATURE))
		return sphx_set_error(SPHX_ERR_UNSUPPORTED, "sphx_forces_basicstep_sa: repacking with dynamic gamma (its CFL condition) is not built; ENABLE_GAMMA_QUADRATURE is");
	SPHX_REQUIRE(forces && pos && vel && info && hash && cellStart && neibsList && gGam && boundElements && vertPos0 && vertPos1 && vertPos2,
		"sphx_forces_basicstep_sa: missing buffer");
	SPHX_REQUIRE(!(ctx->params.simflags & SPHX_ENABLE_DTADAPT) || cfl, "sphx_forces_basicstep_sa: ENABLE_DTADAPT needs the CFL buffer");
	SPHX_REQUIRE(fromParticle <= toParticle, "sphx_forces_basicstep_sa: empty range");
	SPHX_REQUIRE(slength == ctx->params.slength && influenceradius == ctx->params.influenceradius,
		"sphx_forces_basicstep_sa: slength / influenceradius differ from the uploaded constants");
	// numBlocks of the reference's basicstep: rounded up to a multiple of 4 for the reduction (src/cuda/forces.cu:741-743)
	const uint32_t blocks = div_up_u(toParticle - fromParticle, SPHX_BLOCK_FORCES);
	const uint32_t numBlocks = (blocks + 3u)/4u*4u;
	if (h_numBlocks) *h_numBlocks = numBlocks;
	if (!blocks) return SPHX_OK;
	hipStream_t st = (hipStream_t)stream;
	if (cfl && numBlocks > blocks) SPHX_HIP(hipMemsetAsync(cfl + cflOffset + blocks, 0, sizeof(float)*(numBlocks - blocks), st));
	SaForcesArgs a = {};
	a.forces = (float4*)forces; a.cfl = cfl; a.pos = (const float4*)pos; a.vel = (const float4*)vel; a.gGam = (const float4*)gGam;
	a.boundElement = (const float4*)boundElements;
	a.vertPos[0] = (const float2*)vertPos0; a.vertPos[1] = (const float2*)vertPos1; a.vertPos[2] = (const float2*)vertPos2;
	a.info = (const particleinfo*)info; a.hash = hash; a.cellStart = cellStart; a.neibsList = neibsList;
	a.fromParticle = fromParticle; a.toParticle = toParticle; a.cflOffset = cflOffset; a.deltap = deltap;
	if (run_mode == SPHX_REPACK) {
		sa_repack_kernel<<<blocks, SPHX_BLOCK_FORCES, 0, st>>>(ctx->dev, a);
		SPHX_LAUNCH_CHECK("sa_repack_kernel");
		return SPHX_OK;
	}
	// the CFL condition of the gamma transport (dynamic gamma + adaptive dt): BUFFER_CFL_GAMMA in the reference's layout
	const bool gcfl = !(ctx->params.simflags & SPHX_ENABLE_GAMMA_QUADRATURE) && (ctx->params.simflags & SPHX_ENABLE_DTADAPT);
	SPHX_REQUIRE(!gcfl || cflGamma, "sphx_forces_basicstep_sa: dynamic gamma with ENABLE_DTADAPT needs BUFFER_CFL_GAMMA");
	if (gcfl) {
		a.cflGamma = cflGamma; a.cflGammaBlocks = cflGamma + (numParticles + 3u)/4u*4u;
		if (numBlocks > blocks) SPHX_HIP(hipMemsetAsync(a.cflGammaBlocks + cflOffset + blocks, 0, sizeof(float)*(numBlocks - blocks), st));
	}
	if (ke) {
		a.tke = ke->tke; a.eps = ke->eps; a.turbvisc = ke->turbvisc; a.eulerVel = (const float4*)ke->eulerVel;
		a.dkde = ke->dkde; a.cflKeps = ke->cflKeps; a.epsilon = ke->epsilon;
		if (ke->cflKeps && numBlocks > blocks) SPHX_HIP(hipMemsetAsync(ke->cflKeps + cflOffset + blocks, 0, sizeof(float)*(numBlocks - blocks), st));
		sa_forces_kernel<true><<<blocks, SPHX_BLOCK_FORCES, 0, st>>>(ctx->dev, a);
	} else {
		bool used = false;
		rc = sphx_sa_tiles_run(ctx, SPHX_SA_TILE_FORCES, forces, pos, vel, nullptr, info, hash, cellStart, neibsList, gGam,
			numParticles, fromParticle, toParticle, 0.0f, st, &used, &a.tileGuard);
		if (rc != SPHX_OK) return rc;
		a.tiled = used ? 1 : 0;
		if (used && ctx->sa_wall && ctx->sa_wall_neibslist == neibsList) {
			a.wallDone = 1;
			// |grad gamma_as| kept from the density summation of this state (a row is tagged with the particle's position and the
			// generation of the rows; what moves elements -- a rebuild, the Euler step and the normals of a run with moving bodies --
			// starts a new generation)
			a.wc.values = ctx->sa_wall_cache; a.wc.tag = ctx->sa_wall_tag; a.wc.capacity = ctx->sa_wall_capacity; a.wc.gen = ctx->sa_wall_gen;
			rc = sphx_sa_wall_forces(ctx, a, st);
			if (rc != SPHX_OK) return rc;
		}
		sa_forces_kernel<false><<<blocks, SPHX_BLOCK_FORCES, 0, st>>>(ctx->dev, a);
	}
	SPHX_LAUNCH_CHECK("sa_forces_kernel");
	return SPHX_OK;
}

extern "C" int sphx_forces_basicstep_sa(sphx_ctx *ctx, void *forces, float *cfl, float *cflGamma,
	const void *pos, const void *vel, const void *info, const uint32_t *hash, const uint32_t *cellStart, const uint16_t *neibsList,
	const void *gGam, const void *boundElements, const void *vertPos0, const void *vertPos1, const void *vertPos2,
	uint32_t numParticles, uint32_t fromParticle, uint32_t toParticle,
	float deltap, float slength, float dtadaptfactor, float influenceradius, uint32_t cflOffset,
	int run_mode, int step, float dt, uint32_t *h_numBlocks, void *stream)
{
	if (ctx && ctx->params.turbmodel == SPHX_KEPSILON && run_mode == SPHX_SIMULATE)
		return sphx_set_error(SPHX_ERR_INVALID, "sphx_forces_basicstep_sa: the KEPSILON forces read k, epsilon, the eddy viscosity and the Eulerian velocity: call sphx_forces_basicstep_sa_keps");
	return sa_forces_impl(ctx, forces, cfl, cflGamma, nullptr, pos, vel, info, hash, cellStart, neibsList, gGam, boundElements,
		vertPos0, vertPos1, vertPos2, numParticles, fromParticle, toParticle, deltap, slength, dtadaptfactor, influenceradius,
		cflOffset, run_mode, step, dt, h_numBlocks, stream);
}

// ---- the force of the fluid on the boundary elements of bodies that feel it (FG_COMPUTE_FORCE: floating bodies, force-feedback
// bodies, fixed ones whose load is measured).  With SA_BOUNDARY the pair loops leave nothing on a segment; its force is the
// pressure the boundary conditions gave it, acting on its area against its normal: F = -P(rho~) A n
// (compute_boundary_pressure_force, src/cuda/forces_kernel.def:3258-3266), written to the body's rows of BUFFER_RB_FORCES /
// BUFFER_RB_TORQUES (torque about the centre of gravity the forces engine holds) and to the segment's own row of
// BUFFER_FORCES with w = 0, as finalizeforcesDevice does for these rows (:4115-4145; vertices write nothing).  The totals are
// sphx_reduce_rb_forces' as for every other boundary model.
struct SaBodyForceArgs {
	float4 *forces, *rbforces, *rbtorques;
	const float4 *pos, *vel, *boundElement;
	const particleinfo *info; const uint32_t *hash;
	const RbParams *rb;
	uint32_t from, to;
};

__global__ void __launch_bounds__(256)
sa_body_pressure_force_kernel(DevParams p, SaBodyForceArgs a)
{
	const uint32_t index = a.from + blockIdx.x*256 + threadIdx.x;
	if (index >= a.to) return;
	const particleinfo info = a.info[index];
	if (PART_TYPE(info) != PT_BOUNDARY || !HAS_COMPUTE_FORCE(info)) return;
	const float4 pos = a.pos[index];
	if (!is_active_w(pos.w)) return;
	const float4 be = a.boundElement[index];
	const float scale = -sa_P(p, a.vel[index].w, FLUID_NUM(info))*be.w;
	const float4 force = make_float4(scale*be.x, scale*be.y, scale*be.z, 0.0f);
	const uint32_t obj = OBJECT_NUM(info);
	const uint32_t rbindex = (uint32_t)((int)info_id(info) + a.rb->rbstart[obj]);
	a.rbforces[rbindex] = force;
	const int3 gp = grid_pos_from_hash(p, a.hash[index] & CELLTYPE_BITMASK);
	const float armx = (gp.x - a.rb->cgGridPos[obj][0])*p.cs[0] + (pos.x - a.rb->cgPos[obj][0]);
	const float army = (gp.y - a.rb->cgGridPos[obj][1])*p.cs[1] + (pos.y - a.rb->cgPos[obj][1]);
	const float armz = (gp.z - a.rb->cgGridPos[obj][2])*p.cs[2] + (pos.z - a.rb->cgPos[obj][2]);
	a.rbtorques[rbindex] = make_float4(army*force.z - armz*force.y, armz*force.x - armx*force.z, armx*force.y - army*force.x, 0.0f);
	a.forces[index] = force;
}

extern "C" int sphx_sa_body_pressure_forces(sphx_ctx *ctx, void *forces, void *rbforces, void *rbtorques,
	const void *pos, const void *vel, const void *info, const uint32_t *hash, const void *boundElements,
	uint32_t fromParticle, uint32_t toParticle, void *stream)
{
	int rc = sa_check(ctx, "sphx_sa_body_pressure_forces called without SA_BOUNDARY");
	if (rc != SPHX_OK) return rc;
	SPHX_REQUIRE(forces && rbforces && rbtorques && pos && vel && info && hash && boundElements, "sphx_sa_body_pressure_forces: missing buffer");
	SPHX_REQUIRE(fromParticle <= toParticle, "sphx_sa_body_pressure_forces: empty range");
	if (fromParticle == toParticle) return SPHX_OK;
	{ const int rcf = sphx_rb_flush(ctx, (hipStream_t)stream); if (rcf != SPHX_OK) return rcf; }
	SaBodyForceArgs a = {};
	a.forces = (float4*)forces; a.rbforces = (float4*)rbforces; a.rbtorques = (float4*)rbtorques;
	a.pos = (const float4*)pos; a.vel = (const float4*)vel; a.boundElement = (const float4*)boundElements;
	a.info = (const particleinfo*)info; a.hash = hash; a.rb = ctx->rb_dev; a.from = fromParticle; a.to = toParticle;
	sa_body_pressure_force_kernel<<<div_up_u(toParticle - fromParticle, 256), 256, 0, (hipStream_t)stream>>>(ctx->dev, a);
	SPHX_LAUNCH_CHECK("sa_body_pressure_force_kernel");
	return SPHX_OK;
}

extern "C" int sphx_forces_basicstep_sa_keps(sphx_ctx *ctx, void *forces, float *cfl, float *cflGamma, float *cflKeps, float *dkde,
	const void *pos, const void *vel, const void *info, const uint32_t *hash, const uint32_t *cellStart, const uint16_t *neibsList,
	const void *gGam, const void *boundElements, const void *vertPos0, const void *vertPos1, const void *vertPos2,
	const float *tke, const float *eps, const float *turbvisc, const void *eulerVel,
	uint32_t numParticles, uint32_t fromParticle, uint32_t toParticle,
	float deltap, float slength, float dtadaptfactor, float influenceradius, float epsilon, uint32_t cflOffset,
	int run_mode, int step, float dt, uint32_t *h_numBlocks, void *stream)
{
	SPHX_REQUIRE(ctx && ctx->params.turbmodel == SPHX_KEPSILON, "sphx_forces_basicstep_sa_keps: the uploaded option set is not KEPSILON");
	SPHX_REQUIRE(run_mode == SPHX_SIMULATE, "sphx_forces_basicstep_sa_keps: the repacking run mode has no k-epsilon (call sphx_forces_basicstep_sa)");
	SPHX_REQUIRE(dkde && tke && eps && turbvisc && eulerVel, "sphx_forces_basicstep_sa_keps: missing buffer");
	SPHX_REQUIRE(!(ctx->params.simflags & SPHX_ENABLE_DTADAPT) || cflKeps, "sphx_forces_basicstep_sa_keps: ENABLE_DTADAPT needs BUFFER_CFL_KEPS");
	const SaKepsBuffers ke = { cflKeps, dkde, tke, eps, turbvisc, eulerVel, epsilon };
	return sa_forces_impl(ctx, forces, cfl, cflGamma, &ke, pos, vel, info, hash, cellStart, neibsList, gGam, boundElements,
		vertPos0, vertPos1, vertPos2, numParticles, fromParticle, toParticle, deltap, slength, dtadaptfactor, influenceradius,
		cflOffset, run_mode, step, dt, h_numBlocks, stream);
}

// ---- Euler step of the k-epsilon model (src/cuda/euler_kernel.def:219-231,262-274,325-337): semi-implicit k and epsilon of the
// fluid particles, Eulerian velocity of the wall particles (+= dt force), eddy viscosity 0.9 k^2/epsilon of every particle (the
// reference's constant).  A separate launch over the rows the Euler kernel integrates; same values as the fused kernel.
struct KepsEulerArgs {
	float *newTke, *newEps, *newTurbVisc; float4 *newEulerVel;
	const float *oldTke, *oldEps, *dkde; const float4 *oldEulerVel, *forces, *oldPos;
	const particleinfo *info;
	const float *d_dt; float dt, dt_scale;
	uint32_t numParticles;
};

__global__ void __launch_bounds__(256)
keps_euler_kernel(KepsEulerArgs a)
{
	const uint32_t index = blockIdx.x*256 + threadIdx.x;
	if (index >= a.numParticles) return;
	if (!is_active_w(a.oldPos[index].w)) return;
	const float dt = a.d_dt ? a.d_dt[0]*a.dt_scale : a.dt;
	const particleinfo info = a.info[index];
	float k = a.oldTke[index], e = a.oldEps[index];
	float4 ev = a.oldEulerVel[index];
	if (PART_TYPE(info) == PT_FLUID) {
		const float *d = a.dkde + 3*(size_t)index;
		const float oldK = k;
		k = (oldK + dt*d[0])/(1.0f + dt*e/oldK);
		e = (e + dt*d[1])/(1.0f + dt*e/oldK*d[2]);
	} else if (PART_TYPE(info) == PT_BOUNDARY || PART_TYPE(info) == PT_VERTEX) {
		const float4 f = a.forces[index];
		ev.x += dt*f.x; ev.y += dt*f.y; ev.z += dt*f.z; ev.w += dt*f.w;
	}
	a.newTke[index] = k; a.newEps[index] = e; a.newTurbVisc[index] = 0.9f*k*k/e;
	a.newEulerVel[index] = ev;
}

extern "C" int sphx_euler_keps(sphx_ctx *ctx, float *newTke, float *newEps, float *newTurbVisc, void *newEulerVel,
	const float *oldTke, const float *oldEps, const void *oldEulerVel, const float *dkde, const void *forces,
	const void *oldPos, const void *info, uint32_t numParticles, uint32_t particleRangeEnd,
	float dt, const float *d_dt, float dt_scale, void *stream)
{
	(void)numParticles;
	SPHX_REQUIRE(ctx && ctx->params.turbmodel == SPHX_KEPSILON, "sphx_euler_keps: the uploaded option set is not KEPSILON");
	SPHX_REQUIRE(newTke && newEps && newTurbVisc && newEulerVel && oldTke && oldEps && oldEulerVel && dkde && forces && oldPos && info,
		"sphx_euler_keps: missing buffer");
	if (!particleRangeEnd) return SPHX_OK;
	KepsEulerArgs a = { newTke, newEps, newTurbVisc, (float4*)newEulerVel, oldTke, oldEps, dkde, (const float4*)oldEulerVel,
		(const float4*)forces, (const float4*)oldPos, (const particleinfo*)info, d_dt, dt, dt_scale, particleRangeEnd };
	keps_euler_kernel<<<div_up_u(particleRangeEnd, 256), 256, 0, (hipStream_t)stream>>>(a);
	SPHX_LAUNCH_CHECK("keps_euler_kernel");
	return SPHX_OK;
}

// viscous part of dtreduce with KEPSILON (src/cuda/forces.cu:585-598): dt = min(dt, 0.125 h^2/(max_kinematic + max eddy viscosity))
__global__ void __launch_bounds__(256)
keps_dt_kernel(float *d_dt, const float *cflKeps, uint32_t numBlocks, float slength, float max_kinematic)
{
	__shared__ float sm[4];
	float m = 0.0f;
	for (uint32_t i = threadIdx.x; i < numBlocks; i += 256) m = fmaxf(m, cflKeps[i]);
#pragma unroll
	for (int d = 32; d > 0; d >>= 1) m = fmaxf(m, __shfl_down(m, d));
	if ((threadIdx.x & 63u) == 0u) sm[threadIdx.x >> 6] = m;
	__syncthreads();
	if (threadIdx.x == 0) {
		m = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
		float dt_visc = slength*slength/(max_kinematic + m);
		dt_visc *= 0.125f;
		if (dt_visc < d_dt[0]) d_dt[0] = dt_visc;
	}
}

extern "C" int sphx_forces_dtreduce_keps_device(sphx_ctx *ctx, const float *cflKeps, uint32_t numBlocks, float slength,
	float max_kinematic, float *d_dt, void *stream)
{
	SPHX_REQUIRE(ctx && ctx->params.turbmodel == SPHX_KEPSILON, "sphx_forces_dtreduce_keps_device: the uploaded option set is not KEPSILON");
	SPHX_REQUIRE(cflKeps && d_dt, "sphx_forces_dtreduce_keps_device: missing buffer");
	if (!numBlocks) return SPHX_OK;
	keps_dt_kernel<<<1, 256, 0, (hipStream_t)stream>>>(d_dt, cflKeps, numBlocks, slength, max_kinematic);
	SPHX_LAUNCH_CHECK("keps_dt_kernel");
	return SPHX_OK;
}

extern "C" int sphx_forces_dtreduce_keps(sphx_ctx *ctx, const float *cflKeps, uint32_t numBlocks, float slength,
	float max_kinematic, float *h_dt_inout, void *stream)
{
	SPHX_REQUIRE(h_dt_inout != nullptr, "sphx_forces_dtreduce_keps: NULL dt");
	float *d = nullptr;
	SPHX_HIP(hipMalloc((void**)&d, sizeof(float)));
	SPHX_HIP(hipMemcpyAsync(d, h_dt_inout, sizeof(float), hipMemcpyHostToDevice, (hipStream_t)stream));
	int rc = sphx_forces_dtreduce_keps_device(ctx, cflKeps, numBlocks, slength, max_kinematic, d, stream);
	if (rc == SPHX_OK) {
		SPHX_HIP(hipMemcpyAsync(h_dt_inout, d, sizeof(float), hipMemcpyDeviceToHost, (hipStream_t)stream));
		SPHX_HIP(hipStreamSynchronize((hipStream_t)stream));
	}
	(void)hipFree(d);
	return rc;
}

extern "C" int sphx_sa_integrate_gamma(sphx_ctx *ctx, void *newGGam, const void *oldGGam, const void *newPos,
	const void *boundElements, const void *vertPos0, const void *vertPos1, const void *vertPos2, const void *info,
	const uint32_t *hash, const uint32_t *cellStart, const uint16_t *neibsList,
	uint32_t numParticles, uint32_t particleRangeEnd, float dt, int step, float t,
	float epsilon, float slength, float influenceradius, int run_mode, void *stream)
{
	(void)numParticles; (void)dt; (void)step; (void)t;
	int rc = sa_check(ctx, "integrate_gamma called without SA_BOUNDARY");
	if (rc != SPHX_OK) return rc;
	if (!(ctx->params.simflags & SPHX_ENABLE_GAMMA_QUADRATURE))
		return sphx_set_error(SPHX_ERR_UNSUPPORTED, "sphx_sa_integrate_gamma: dynamic gamma (transport equation) is not built; ENABLE_GAMMA_QUADRATURE is");
	// ENABLE_MOVING_BODIES: boundElements is BUFFER_BOUNDELEMENTS of the NEW state (quadrature_gamma_neib_data reads
	// params.newBoundElement) and gamma of the vertex rows is integrated as well (integrate_gamma_impl, src/cuda/euler.cu:250-253).
	// Not in the REPACK run mode: its branch (:222-239) integrates the fluid rows only and copies the vertex and boundary rows
	// whatever the flags, against the elements of the state that is read (nothing moves while repacking)
	const bool moving = (ctx->params.simflags & SPHX_ENABLE_MOVING_BODIES) != 0 && run_mode != SPHX_REPACK;
	SPHX_REQUIRE(newGGam && oldGGam && newPos && boundElements && vertPos0 && vertPos1 && vertPos2 && info && hash && cellStart && neibsList,
		"sphx_sa_integrate_gamma: missing buffer");
	SPHX_REQUIRE(newGGam != oldGGam, "sphx_sa_integrate_gamma: in-place use is not supported");
	SPHX_REQUIRE(slength == ctx->params.slength && influenceradius == ctx->params.influenceradius,
		"sphx_sa_integrate_gamma: slength / influenceradius differ from the uploaded constants");
	if (!particleRangeEnd) return SPHX_OK;
	SaIntGammaArgs a = {};
	a.newGGam = (float4*)newGGam; a.oldGGam = (const float4*)oldGGam; a.pos = (const float4*)newPos;
	a.boundElement = (const float4*)boundElements;
	a.vertPos[0] = (const float2*)vertPos0; a.vertPos[1] = (const float2*)vertPos1; a.vertPos[2] = (const float2*)vertPos2;
	a.info = (const particleinfo*)info; a.hash = hash; a.cellStart = cellStart; a.neibsList = neibsList;
	a.numParticles = particleRangeEnd; a.epsilon = epsilon;
	a.vertexRows = moving ? 1 : 0;
	if (ctx->sa_wall && ctx->sa_wall_neibslist == neibsList && !ctx->disable_tiles && !moving) {
		a.wallDone = 1;
		a.wc.values = ctx->sa_wall_cache; a.wc.tag = ctx->sa_wall_tag; a.wc.capacity = ctx->sa_wall_capacity; a.wc.gen = ctx->sa_wall_gen;
		rc = sphx_sa_wall_integrate_gamma(ctx, a, (hipStream_t)stream);
		if (rc != SPHX_OK) return rc;
	}
	sa_integrate_gamma_kernel<<<div_up_u(particleRangeEnd, 128), 128, 0, (hipStream_t)stream>>>(ctx->dev, a);
	SPHX_LAUNCH_CHECK("sa_integrate_gamma_kernel");
	return SPHX_OK;
}

extern "C" int sphx_forces_dtreduce_gamma_device(sphx_ctx *ctx, const float *cflGamma, uint32_t numParticles, uint32_t numBlocks,
	float *d_dt, void *stream)
{
	int rc = sa_check(ctx, "dtreduce (gamma) called without SA_BOUNDARY");
	if (rc != SPHX_OK) return rc;
	SPHX_REQUIRE(cflGamma && d_dt, "sphx_forces_dtreduce_gamma_device: missing buffer");
	if (!numBlocks) return SPHX_OK;
	sa_gamma_dt_kernel<<<1, 256, 0, (hipStream_t)stream>>>(d_dt, cflGamma + (numParticles + 3u)/4u*4u, numBlocks);
	SPHX_LAUNCH_CHECK("sa_gamma_dt_kernel");
	return SPHX_OK;
}

extern "C" int sphx_forces_dtreduce_gamma(sphx_ctx *ctx, const float *cflGamma, uint32_t numParticles, uint32_t numBlocks,
	float *h_dt_inout, void *stream)
{
	SPHX_REQUIRE(h_dt_inout != nullptr, "sphx_forces_dtreduce_gamma: NULL dt");
	float *d = nullptr;
	SPHX_HIP(hipMalloc((void**)&d, sizeof(float)));
	SPHX_HIP(hipMemcpyAsync(d, h_dt_inout, sizeof(float), hipMemcpyHostToDevice, (hipStream_t)stream));
	int rc = sphx_forces_dtreduce_gamma_device(ctx, cflGamma, numParticles, numBlocks, d, stream);
	if (rc == SPHX_OK) {
		SPHX_HIP(hipMemcpyAsync(h_dt_inout, d, sizeof(float), hipMemcpyDeviceToHost, (hipStream_t)stream));
		SPHX_HIP(hipStreamSynchronize((hipStream_t)stream));
	}
	(void)hipFree(d);
	return rc;
}

extern "C" int sphx_sa_density_sum(sphx_ctx *ctx, void *newVel, void *newGGam, void *forces,
	const void *oldPos, const void *newPos, const void *oldVel, const void *oldGGam, const void *boundElements,
	const void *vertPos0, const void *vertPos1, const void *vertPos2, const void *info,
	const uint32_t *hash, const uint32_t *cellStart, const uint16_t *neibsList,
	uint32_t numParticles, uint32_t particleRangeEnd, float dt, int step, float t, float epsilon,
	float deltap, float slength, float influenceradius, void *stream)
{
	(void)dt; (void)step; (void)t; (void)epsilon; (void)deltap;
	int rc = sa_check(ctx, "density_sum called without SA_BOUNDARY");
	if (rc != SPHX_OK) return rc;
	if (!(ctx->params.simflags & SPHX_ENABLE_DENSITY_SUM) || (ctx->params.simflags & SPHX_ENABLE_GAMMA_QUADRATURE))
		return sphx_set_error(SPHX_ERR_INVALID, "sphx_sa_density_sum: needs ENABLE_DENSITY_SUM with dynamic gamma");
	if (ctx->params.simflags & SPHX_ENABLE_MOVING_BODIES)
		return sphx_set_error(SPHX_ERR_INVALID, "sphx_sa_density_sum: with ENABLE_MOVING_BODIES the boundary elements are double buffered: call sphx_sa_density_sum_moving");
	if (ctx->params.sph_formulation != SPHX_SPH_F1 && ctx->params.sph_formulation != SPHX_SPH_F2)
		return sphx_set_error(SPHX_ERR_UNSUPPORTED, "sphx_sa_density_sum: SPH_HA is not built");
	SPHX_REQUIRE(newVel && newGGam && forces && oldPos && newPos && oldVel && oldGGam && boundElements && vertPos0 && vertPos1 && vertPos2 &&
		info && hash && cellStart && neibsList, "sphx_sa_density_sum: missing buffer");
	SPHX_REQUIRE(newGGam != oldGGam, "sphx_sa_density_sum: gamma is double buffered");
	SPHX_REQUIRE(slength == ctx->params.slength && influenceradius == ctx->params.influenceradius,
		"sphx_sa_density_sum: slength / influenceradius differ from the uploaded constants");
	if (!particleRangeEnd) return SPHX_OK;
	SaDensitySumArgs a = {};
	a.newVel = (float4*)newVel; a.newGGam = (float4*)newGGam; a.forces = (float4*)forces;
	a.oldPos = (const float4*)oldPos; a.pos = (const float4*)newPos; a.oldVel = (const float4*)oldVel; a.oldGGam = (const float4*)oldGGam;
	a.boundElement = (const float4*)boundElements;
	a.vertPos[0] = (const float2*)vertPos0; a.vertPos[1] = (const float2*)vertPos1; a.vertPos[2] = (const float2*)vertPos2;
	a.info = (const particleinfo*)info; a.hash = hash; a.cellStart = cellStart; a.neibsList = neibsList; a.numParticles = particleRangeEnd;
	{
		bool used = false;
		rc = sphx_sa_tiles_run(ctx, SPHX_SA_TILE_DSUM, forces, oldPos, nullptr, newPos, info, hash, cellStart, neibsList, nullptr,
			numParticles, 0u, particleRangeEnd, 0.0f, (hipStream_t)stream, &used, &a.tileGuard);
		if (rc != SPHX_OK) return rc;
		a.tiled = used ? 1 : 0;
		if (used && ctx->sa_wall && ctx->sa_wall_neibslist == neibsList) {
			a.wallDone = 1;
			a.wc.values = ctx->sa_wall_cache; a.wc.tag = ctx->sa_wall_tag; a.wc.capacity = ctx->sa_wall_capacity; a.wc.gen = ctx->sa_wall_gen;
			rc = sphx_sa_wall_density_sum(ctx, a, (hipStream_t)stream);
			if (rc != SPHX_OK) return rc;
		}
	}
	sa_density_sum_kernel<false><<<div_up_u(particleRangeEnd, 128), 128, 0, (hipStream_t)stream>>>(ctx->dev, a);
	SPHX_LAUNCH_CHECK("sa_density_sum_kernel");
	return SPHX_OK;
}

// density_sum with ENABLE_MOVING_BODIES (density_sum_impl<SA_BOUNDARY>, src/cuda/euler.cu:112-160): the old and the new
// BUFFER_BOUNDELEMENTS (the read and the write list of the reference's call hold one each), fluid rows as sphx_sa_density_sum,
// gamma of the vertex rows integrated, boundary rows untouched.  The list walker is the whole pass (the tiled window sums and
// the wave-per-wall-particle kernels assume elements at rest).
extern "C" int sphx_sa_density_sum_moving(sphx_ctx *ctx, void *newVel, void *newGGam, void *forces,
	const void *oldPos, const void *newPos, const void *oldVel, const void *oldGGam,
	const void *oldBoundElements, const void *newBoundElements,
	const void *vertPos0, const void *vertPos1, const void *vertPos2, const void *info,
	const uint32_t *hash, const uint32_t *cellStart, const uint16_t *neibsList,
	uint32_t numParticles, uint32_t particleRangeEnd, void *stream)
{
	(void)numParticles;
	int rc = sa_check(ctx, "density_sum called without SA_BOUNDARY");
	if (rc != SPHX_OK) return rc;
	if (!(ctx->params.simflags & SPHX_ENABLE_DENSITY_SUM) || (ctx->params.simflags & SPHX_ENABLE_GAMMA_QUADRATURE))
		return sphx_set_error(SPHX_ERR_INVALID, "sphx_sa_density_sum_moving: needs ENABLE_DENSITY_SUM with dynamic gamma");
	if (!(ctx->params.simflags & SPHX_ENABLE_MOVING_BODIES))
		return sphx_set_error(SPHX_ERR_INVALID, "sphx_sa_density_sum_moving: the uploaded option set has no ENABLE_MOVING_BODIES (call sphx_sa_density_sum)");
	if (ctx->params.simflags & SPHX_ENABLE_INLET_OUTLET)
		return sphx_set_error(SPHX_ERR_INVALID, "sphx_sa_density_sum_moving: with ENABLE_INLET_OUTLET as well the pass reads the Eulerian velocities: call sphx_sa_density_sum_io_moving");
	SPHX_REQUIRE(newVel && newGGam && forces && oldPos && newPos && oldVel && oldGGam && oldBoundElements && newBoundElements &&
		vertPos0 && vertPos1 && vertPos2 && info && hash && cellStart && neibsList, "sphx_sa_density_sum_moving: missing buffer");
	SPHX_REQUIRE(newGGam != oldGGam && oldBoundElements != newBoundElements, "sphx_sa_density_sum_moving: gamma and the boundary elements are double buffered");
	if (!particleRangeEnd) return SPHX_OK;
	SaDensitySumArgs a = {};
	a.newVel = (float4*)newVel; a.newGGam = (float4*)newGGam; a.forces = (float4*)forces;
	a.oldPos = (const float4*)oldPos; a.pos = (const float4*)newPos; a.oldVel = (const float4*)oldVel; a.oldGGam = (const float4*)oldGGam;
	a.boundElement = (const float4*)oldBoundElements; a.boundElementNew = (const float4*)newBoundElements;
	a.vertPos[0] = (const float2*)vertPos0; a.vertPos[1] = (const float2*)vertPos1; a.vertPos[2] = (const float2*)vertPos2;
	a.info = (const particleinfo*)info; a.hash = hash; a.cellStart = cellStart; a.neibsList = neibsList; a.numParticles = particleRangeEnd;
	{
		// the particle <- particle sums (fluid and vertex neighbours, the moving vertices with their own displacement like any other
		// particle) through the tiled window as for walls at rest (round 6)
		bool used = false;
		rc = sphx_sa_tiles_run(ctx, SPHX_SA_TILE_DSUM, forces, oldPos, nullptr, newPos, info, hash, cellStart, neibsList, nullptr,
			numParticles, 0u, particleRangeEnd, 0.0f, (hipStream_t)stream, &used, &a.tileGuard);
		if (rc != SPHX_OK) return rc;
		a.tiled = used ? 1 : 0;
		// ... and the boundary-element terms of the fluid particles next to a wall with one element per lane, both states of
		// every element (sa_density_sum_wall_moving_kernel); the vertex rows and a run without tiles stay with the walker below
		if (used && ctx->sa_wall && ctx->sa_wall_neibslist == neibsList) {
			a.wallDone = ctx->sa_wall_vert ? 3 : 1;
			a.wc.values = ctx->sa_wall_cache; a.wc.tag = ctx->sa_wall_tag; a.wc.capacity = ctx->sa_wall_capacity; a.wc.gen = ctx->sa_wall_gen;
			rc = sphx_sa_wall_density_sum_moving(ctx, a, (hipStream_t)stream);
			if (rc != SPHX_OK) return rc;
		}
	}
	sa_density_sum_kernel<false, true><<<div_up_u(particleRangeEnd, 128), 128, 0, (hipStream_t)stream>>>(ctx->dev, a);
	SPHX_LAUNCH_CHECK("sa_density_sum_kernel<moving>");
	return SPHX_OK;
}

extern "C" int sphx_sa_compute_density_diffusion(sphx_ctx *ctx, void *forces, const void *pos, const void *vel, const void *gGam,
	const void *info, const uint32_t *hash, const uint32_t *cellStart, const uint16_t *neibsList,
	uint32_t numParticles, uint32_t particleRangeEnd, float deltap, float slength, float influenceradius, float dt, void *stream)
{
	(void)deltap;
	int rc = sa_check(ctx, "compute_density_diffusion called without SA_BOUNDARY");
	if (rc != SPHX_OK) return rc;
	if (ctx->params.densitydiffusiontype != SPHX_BREZZI || !(ctx->params.simflags & SPHX_ENABLE_DENSITY_SUM) ||
		ctx->params.sph_formulation == SPHX_SPH_HA)
		return sphx_set_error(SPHX_ERR_UNSUPPORTED, "sphx_sa_compute_density_diffusion: built for Brezzi diffusion with density summation");
	SPHX_REQUIRE(forces && pos && vel && gGam && info && hash && cellStart && neibsList, "sphx_sa_compute_density_diffusion: missing buffer");
	SPHX_REQUIRE(slength == ctx->params.slength && influenceradius == ctx->params.influenceradius,
		"sphx_sa_compute_density_diffusion: slength / influenceradius differ from the uploaded constants");
	if (!particleRangeEnd) return SPHX_OK;
	SaDiffusionArgs a = {};
	a.forces = (float4*)forces; a.pos = (const float4*)pos; a.vel = (const float4*)vel; a.gGam = (const float4*)gGam;
	a.info = (const particleinfo*)info; a.hash = hash; a.cellStart = cellStart; a.neibsList = neibsList;
	a.numParticles = particleRangeEnd; a.dt = dt;
	{
		bool used = false;
		rc = sphx_sa_tiles_run(ctx, SPHX_SA_TILE_DIFF, forces, pos, vel, nullptr, info, hash, cellStart, neibsList, gGam,
			numParticles, 0u, particleRangeEnd, dt, (hipStream_t)stream, &used, &a.tileGuard);
		if (rc != SPHX_OK) return rc;
		if (used && !a.tileGuard) return SPHX_OK;      // the host has seen the tiling succeed: no stand-by launch
	}
	sa_density_diffusion_kernel<false><<<div_up_u(particleRangeEnd, 128), 128, 0, (hipStream_t)stream>>>(ctx->dev, a);
	SPHX_LAUNCH_CHECK("sa_density_diffusion_kernel");
	return SPHX_OK;
}

extern "C" int sphx_apply_density_diffusion(sphx_ctx *ctx, void *vel, const void *forces, const void *info,
	uint32_t numParticles, uint32_t particleRangeEnd, float dt, void *stream)
{
	(void)numParticles;
	SPHX_REQUIRE(ctx && ctx->have_params, "sphx_apply_density_diffusion: constants not set");
	SPHX_REQUIRE(vel && forces && info, "sphx_apply_density_diffusion: missing buffer");
	if (!particleRangeEnd) return SPHX_OK;
	sa_update_density_kernel<<<div_up_u(particleRangeEnd, 256), 256, 0, (hipStream_t)stream>>>((float4*)vel, (const float4*)forces,
		(const particleinfo*)info, particleRangeEnd, dt);
	SPHX_LAUNCH_CHECK("sa_update_density_kernel");
	return SPHX_OK;
}

// ==========================================================================================
// A run with open boundaries (SA_BOUNDARY + ENABLE_INLET_OUTLET, laminar): the passes over the FLUID particles.  They are the list
// walkers above with their OPEN terms (the tiled window and the wave-per-wall-particle kernels do not know those terms, so the
// walkers are the whole pass here); the passes over the elements of the open faces themselves are sa_io.hip's.
//   density_sum       src/cuda/density_sum_kernel.cu:119-140,206-250,374-420,606-655
//   forces            src/cuda/forces_kernel.def:1485-1497,2494-2507,2703-2708
//   density diffusion src/cuda/forces_kernel.def:1836-1852,4536-4582
// ==========================================================================================
static int sa_open_check(sphx_ctx *ctx, const char *who)
{
	int rc = sa_check(ctx, who);
	if (rc != SPHX_OK) return rc;
	if (ctx->dev.turbmodel == SPHX_KEPSILON)
		return sphx_set_error(SPHX_ERR_UNSUPPORTED, "sphx_sa_io: open boundaries with k-epsilon are not built");
	return SPHX_OK;
}

int sphx_sa_solid_rows_launch(sphx_ctx *ctx, const SaArgs &a_, bool vertexPass, hipStream_t st)
{
	SaArgs a = a_;
	if (ctx->sa_wall_neibslist == a.neibsList && a.numParticles <= ctx->sa_rows_range) a.rows = vertexPass ? ctx->sa_rows_vert : ctx->sa_rows_bound;
	if (vertexPass) sa_vertex_bc_kernel<SPHX_WENDLAND, false><<<div_up_u(a.numParticles, 128), 128, 0, st>>>(ctx->dev, a);
	else sa_segment_bc_kernel<SPHX_WENDLAND, false><<<div_up_u(a.numParticles, 128), 128, 0, st>>>(ctx->dev, a);
	SPHX_LAUNCH_CHECK("sa_segment_bc_kernel / sa_vertex_bc_kernel (solid rows of a run with open boundaries)");
	return SPHX_OK;
}

extern "C" int sphx_sa_density_sum_io(sphx_ctx *ctx, void *newVel, void *newGGam, void *forces, const void *oldPos, const void *newPos,
	const void *oldVel, const void *oldEulerVel, const void *oldGGam, const void *boundElements,
	const void *vertPos0, const void *vertPos1, const void *vertPos2, const void *info,
	const uint32_t *hash, const uint32_t *cellStart, const uint16_t *neibsList,
	uint32_t numParticles, uint32_t particleRangeEnd, float dt, void *stream)
{
	int rc = sa_open_check(ctx, "density_sum called without SA_BOUNDARY");
	if (rc != SPHX_OK) return rc;
	SPHX_REQUIRE(newVel && newGGam && forces && oldPos && newPos && oldVel && oldEulerVel && oldGGam && boundElements && vertPos0 && vertPos1 &&
		vertPos2 && info && hash && cellStart && neibsList, "sphx_sa_density_sum_io: missing buffer");
	if (!particleRangeEnd) return SPHX_OK;
	SaDensitySumArgs a = {};
	a.newVel = (float4*)newVel; a.newGGam = (float4*)newGGam; a.forces = (float4*)forces;
	a.oldPos = (const float4*)oldPos; a.pos = (const float4*)newPos; a.oldVel = (const float4*)oldVel; a.oldGGam = (const float4*)oldGGam;
	a.boundElement = (const float4*)boundElements;
	a.vertPos[0] = (const float2*)vertPos0; a.vertPos[1] = (const float2*)vertPos1; a.vertPos[2] = (const float2*)vertPos2;
	a.info = (const particleinfo*)info; a.hash = hash; a.cellStart = cellStart; a.neibsList = neibsList; a.numParticles = particleRangeEnd;
	a.oldEulerVel = (const float4*)oldEulerVel; a.dt = dt;
	if (newVel != oldVel) {
		// round 6: the particle <- particle sums through the tiled window as for solid walls (an open vertex summed like any vertex),
		// the boundary elements with one element per lane; what the open faces change -- the flux of gamma through their segments,
		// the virtual displacement of their vertices -- rides with the latter (sa_density_sum_wall_kernel<true>, sa_wall.hip), which
		// hands the flux over in newVel.w (hence two buffers).  The one-thread kernel was 49 % of a step of the SAChannelIO mirror at
		// 8.6 M particles (profiles/r06_sa_io_kernel_stats.txt)
		bool used = false;
		rc = sphx_sa_tiles_run(ctx, SPHX_SA_TILE_DSUM, forces, oldPos, nullptr, newPos, info, hash, cellStart, neibsList, nullptr,
			numParticles, 0u, particleRangeEnd, 0.0f, (hipStream_t)stream, &used, &a.tileGuard);
		if (rc != SPHX_OK) return rc;
		if (used && ctx->sa_wall && ctx->sa_wall_neibslist == neibsList) {
			a.tiled = 1; a.wallDone = 1;
			a.wc.values = ctx->sa_wall_cache; a.wc.tag = ctx->sa_wall_tag; a.wc.capacity = ctx->sa_wall_capacity; a.wc.gen = ctx->sa_wall_gen;
			a.wc.gsum = ctx->sa_wall_gsum;
			ctx->sa_wall_open_neibslist = nullptr;
			if (!ctx->sa_wall_open && hipMalloc((void**)&ctx->sa_wall_open, sizeof(uint32_t)*((size_t)ctx->reserved_particles + 1)) != hipSuccess) {
				(void)hipGetLastError();
				ctx->sa_wall_open = nullptr;      // the diffusion goes through every wall particle then
			}
			if (ctx->sa_wall_open) {
				SPHX_HIP(hipMemsetAsync(ctx->sa_wall_open, 0, sizeof(uint32_t), (hipStream_t)stream));
				a.openList = ctx->sa_wall_open;
			}
			rc = sphx_sa_wall_density_sum(ctx, a, (hipStream_t)stream);
			if (rc != SPHX_OK) return rc;
			if (ctx->sa_wall_open && particleRangeEnd == numParticles) {      // (a pass over a part of the particles leaves a part of the list)
				ctx->sa_wall_open_neibslist = neibsList; ctx->sa_wall_open_gen = ctx->sa_wall_gen;
			}
		} else
			a.tileGuard = nullptr;      // (no list of wall particles: the walker is the whole pass, whatever the tiles left in FORCES.w)
	}
	sa_density_sum_kernel<true><<<div_up_u(particleRangeEnd, 128), 128, 0, (hipStream_t)stream>>>(ctx->dev, a);
	SPHX_LAUNCH_CHECK("sa_density_sum_kernel<open>");
	return SPHX_OK;
}

// ENABLE_INLET_OUTLET | ENABLE_DENSITY_SUM | ENABLE_MOVING_BODIES, the option set of CompleteSaExample.cu (:46): the open faces'
// terms of sphx_sa_density_sum_io in the boundary loop of sphx_sa_density_sum_moving (io_gamma_contrib inside
// computeDensitySumBoundaryTerms, src/cuda/density_sum_kernel.cu:422-484) -- elements where they were and where they are, gamma
// of the VERTEX rows integrated, the virtual displacement of the open vertices and segments in the sums of the fluid
extern "C" int sphx_sa_density_sum_io_moving(sphx_ctx *ctx, void *newVel, void *newGGam, void *forces, const void *oldPos, const void *newPos,
	const void *oldVel, const void *oldEulerVel, const void *oldGGam, const void *oldBoundElements, const void *newBoundElements,
	const void *vertPos0, const void *vertPos1, const void *vertPos2, const void *info,
	const uint32_t *hash, const uint32_t *cellStart, const uint16_t *neibsList,
	uint32_t numParticles, uint32_t particleRangeEnd, float dt, void *stream)
{
	int rc = sa_open_check(ctx, "density_sum called without SA_BOUNDARY");
	if (rc != SPHX_OK) return rc;
	if (!(ctx->params.simflags & SPHX_ENABLE_MOVING_BODIES) || !(ctx->params.simflags & SPHX_ENABLE_INLET_OUTLET))
		return sphx_set_error(SPHX_ERR_INVALID, "sphx_sa_density_sum_io_moving: the uploaded option set has not both ENABLE_INLET_OUTLET and ENABLE_MOVING_BODIES");
	SPHX_REQUIRE(newVel && newGGam && forces && oldPos && newPos && oldVel && oldEulerVel && oldGGam && oldBoundElements && newBoundElements &&
		vertPos0 && vertPos1 && vertPos2 && info && hash && cellStart && neibsList, "sphx_sa_density_sum_io_moving: missing buffer");
	SPHX_REQUIRE(newGGam != oldGGam && oldBoundElements != newBoundElements, "sphx_sa_density_sum_io_moving: gamma and the boundary elements are double buffered");
	if (!particleRangeEnd) return SPHX_OK;
	SaDensitySumArgs a = {};
	a.newVel = (float4*)newVel; a.newGGam = (float4*)newGGam; a.forces = (float4*)forces;
	a.oldPos = (const float4*)oldPos; a.pos = (const float4*)newPos; a.oldVel = (const float4*)oldVel; a.oldGGam = (const float4*)oldGGam;
	a.boundElement = (const float4*)oldBoundElements; a.boundElementNew = (const float4*)newBoundElements;
	a.vertPos[0] = (const float2*)vertPos0; a.vertPos[1] = (const float2*)vertPos1; a.vertPos[2] = (const float2*)vertPos2;
	a.info = (const particleinfo*)info; a.hash = hash; a.cellStart = cellStart; a.neibsList = neibsList; a.numParticles = particleRangeEnd;
	a.oldEulerVel = (const float4*)oldEulerVel; a.dt = dt;
	if (newVel != oldVel) {
		// round 6: as sphx_sa_density_sum_io and sphx_sa_density_sum_moving -- the particle <- particle sums through the tiled window, the
		// boundary-element terms of the fluid rows (with the open faces' terms) and of the vertex rows with one element per lane
		// (sa_density_sum_wall_moving_kernel<true>); the flux of gamma rides in newVel.w, hence two buffers
		bool used = false;
		rc = sphx_sa_tiles_run(ctx, SPHX_SA_TILE_DSUM, forces, oldPos, nullptr, newPos, info, hash, cellStart, neibsList, nullptr,
			numParticles, 0u, particleRangeEnd, 0.0f, (hipStream_t)stream, &used, &a.tileGuard);
		if (rc != SPHX_OK) return rc;
		if (used && ctx->sa_wall && ctx->sa_wall_neibslist == neibsList) {
			a.tiled = 1; a.wallDone = ctx->sa_wall_vert ? 3 : 1;
			a.wc.values = ctx->sa_wall_cache; a.wc.tag = ctx->sa_wall_tag; a.wc.capacity = ctx->sa_wall_capacity; a.wc.gen = ctx->sa_wall_gen;
			ctx->sa_wall_open_neibslist = nullptr;
			if (!ctx->sa_wall_open && hipMalloc((void**)&ctx->sa_wall_open, sizeof(uint32_t)*((size_t)ctx->reserved_particles + 1)) != hipSuccess) {
				(void)hipGetLastError();
				ctx->sa_wall_open = nullptr;
			}
			if (ctx->sa_wall_open) {
				SPHX_HIP(hipMemsetAsync(ctx->sa_wall_open, 0, sizeof(uint32_t), (hipStream_t)stream));
				a.openList = ctx->sa_wall_open;
			}
			rc = sphx_sa_wall_density_sum_moving(ctx, a, (hipStream_t)stream);
			if (rc != SPHX_OK) return rc;
			if (ctx->sa_wall_open && particleRangeEnd == numParticles) {
				ctx->sa_wall_open_neibslist = neibsList; ctx->sa_wall_open_gen = ctx->sa_wall_gen;
			}
		} else
			a.tileGuard = nullptr;      // (no list of wall particles: the walker is the whole pass, whatever the tiles left in FORCES.w)
	}
	sa_density_sum_kernel<true, true><<<div_up_u(particleRangeEnd, 128), 128, 0, (hipStream_t)stream>>>(ctx->dev, a);
	SPHX_LAUNCH_CHECK("sa_density_sum_kernel<open, moving>");
	return SPHX_OK;
}

extern "C" int sphx_forces_basicstep_sa_io(sphx_ctx *ctx, void *forces, float *cfl, float *cflGamma, const void *pos, const void *vel,
	const void *eulerVel, const void *info, const uint32_t *hash, const uint32_t *cellStart, const uint16_t *neibsList,
	const void *gGam, const void *boundElements, const void *vertPos0, const void *vertPos1, const void *vertPos2,
	uint32_t numParticles, uint32_t fromParticle, uint32_t toParticle, float deltap, uint32_t cflOffset, uint32_t *h_numBlocks, void *stream)
{
	int rc = sa_open_check(ctx, "forces called without SA_BOUNDARY");
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
	SaForcesArgs a = {};
	a.forces = (float4*)forces; a.cfl = dtadapt ? cfl : nullptr;
	a.cflGamma = gcfl ? cflGamma : nullptr; a.cflGammaBlocks = gcfl ? cflGamma + round_up_u(numParticles, 4u) : nullptr;
	a.pos = (const float4*)pos; a.vel = (const float4*)vel; a.eulerVel = (const float4*)eulerVel; a.gGam = (const float4*)gGam;
	a.boundElement = (const float4*)boundElements;
	a.vertPos[0] = (const float2*)vertPos0; a.vertPos[1] = (const float2*)vertPos1; a.vertPos[2] = (const float2*)vertPos2;
	a.info = (const particleinfo*)info; a.hash = hash; a.cellStart = cellStart; a.neibsList = neibsList;
	a.fromParticle = fromParticle; a.toParticle = toParticle; a.cflOffset = cflOffset; a.deltap = deltap;
	{
		// round 6: the particle <- particle sums through the tiled window as for solid walls, the boundary elements with one element
		// per lane; what an open boundary adds -- the Eulerian velocities in the viscous terms of vertices and elements, the second
		// gamma CFL term -- rides with the latter (sa_forces_wall_kernel, a.open); this kernel then only finishes the rows
		bool used = false;
		rc = sphx_sa_tiles_run(ctx, SPHX_SA_TILE_FORCES, forces, pos, vel, nullptr, info, hash, cellStart, neibsList, gGam,
			numParticles, fromParticle, toParticle, 0.0f, (hipStream_t)stream, &used, &a.tileGuard);
		if (rc != SPHX_OK) return rc;
		if (used && ctx->sa_wall && ctx->sa_wall_neibslist == neibsList) {
			a.tiled = 1; a.wallDone = 1; a.open = 1;
			a.wc.values = ctx->sa_wall_cache; a.wc.tag = ctx->sa_wall_tag; a.wc.capacity = ctx->sa_wall_capacity; a.wc.gen = ctx->sa_wall_gen;
			// (the sum of grad gamma_as at step n for the density summations of this step -- not with moving bodies, whose elements are
			// elsewhere when those run)
			if (!(ctx->params.simflags & SPHX_ENABLE_MOVING_BODIES)) {
				if (!ctx->sa_wall_gsum && ctx->sa_wall_capacity) {      // two rows per wall particle that has rows of |grad gamma_as|
					if (hipMalloc((void**)&ctx->sa_wall_gsum, sizeof(float4)*2u*ctx->sa_wall_capacity) == hipSuccess)
						SPHX_HIP(hipMemsetAsync(ctx->sa_wall_gsum, 0, sizeof(float4)*2u*ctx->sa_wall_capacity, (hipStream_t)stream));
					else {
						(void)hipGetLastError();
						ctx->sa_wall_gsum = nullptr;      // the density summations evaluate the elements at step n themselves
					}
				}
				a.wc.gsum = ctx->sa_wall_gsum;
			}
			rc = sphx_sa_wall_forces(ctx, a, (hipStream_t)stream);
			if (rc != SPHX_OK) return rc;
		} else
			a.tileGuard = nullptr;      // (no list of wall particles: the walker is the whole pass)
	}
	sa_forces_kernel<false, true><<<numBlocks, SPHX_BLOCK_FORCES, 0, (hipStream_t)stream>>>(ctx->dev, a);
	SPHX_LAUNCH_CHECK("sa_forces_kernel<open>");
	return SPHX_OK;
}

extern "C" int sphx_sa_compute_density_diffusion_io(sphx_ctx *ctx, void *forces, const void *pos, const void *vel, const void *gGam,
	const void *boundElements, const void *vertPos0, const void *vertPos1, const void *vertPos2, const void *info,
	const uint32_t *hash, const uint32_t *cellStart, const uint16_t *neibsList,
	uint32_t numParticles, uint32_t particleRangeEnd, float deltap, float dt, void *stream)
{
	(void)numParticles;
	int rc = sa_open_check(ctx, "compute_density_diffusion called without SA_BOUNDARY");
	if (rc != SPHX_OK) return rc;
	if (ctx->params.densitydiffusiontype != SPHX_BREZZI || !(ctx->params.simflags & SPHX_ENABLE_DENSITY_SUM) ||
		ctx->params.sph_formulation == SPHX_SPH_HA)
		return sphx_set_error(SPHX_ERR_UNSUPPORTED, "sphx_sa_compute_density_diffusion_io: built for Brezzi diffusion with density summation");
	SPHX_REQUIRE(forces && pos && vel && gGam && boundElements && vertPos0 && vertPos1 && vertPos2 && info && hash && cellStart && neibsList,
		"sphx_sa_compute_density_diffusion_io: missing buffer");
	if (!particleRangeEnd) return SPHX_OK;
	SaDiffusionArgs a = {};
	a.forces = (float4*)forces; a.pos = (const float4*)pos; a.vel = (const float4*)vel; a.gGam = (const float4*)gGam;
	a.info = (const particleinfo*)info; a.hash = hash; a.cellStart = cellStart; a.neibsList = neibsList;
	a.numParticles = particleRangeEnd; a.dt = dt;
	a.boundElement = (const float4*)boundElements;
	a.vertPos[0] = (const float2*)vertPos0; a.vertPos[1] = (const float2*)vertPos1; a.vertPos[2] = (const float2*)vertPos2; a.deltap = deltap;
	{
		// the fluid <- fluid sum is the solid-wall one: through the tiled window (SPHX_SA_TILE_DIFF finishes the row: / gamma / rho0);
		// the segments of the pressure-driven open faces add theirs to it with one element per lane (sa_wall.hip).  Round 6: such a run
		// rebuilds its lists in every step, and the tile lists of one rebuild cost a tenth of what the list walkers of one step did
		// (profiles/r06_sa_io_kernel_stats.txt)
		bool used = false;
		rc = sphx_sa_tiles_run(ctx, SPHX_SA_TILE_DIFF, forces, pos, vel, nullptr, info, hash, cellStart, neibsList, gGam,
			numParticles, 0u, particleRangeEnd, dt, (hipStream_t)stream, &used, &a.tileGuard);
		if (rc != SPHX_OK) return rc;
		if (used && ctx->sa_wall && ctx->sa_wall_neibslist == neibsList) {
			rc = sphx_sa_wall_density_diffusion_open(ctx, a, (hipStream_t)stream);
			if (rc != SPHX_OK) return rc;
			if (!a.tileGuard) return SPHX_OK;      // the host has seen the tiling succeed: no stand-by launch
		} else
			a.tileGuard = nullptr;      // (no list of wall particles: the walker is the whole pass, whatever the tiles left in FORCES.w)
	}
	sa_density_diffusion_kernel<true><<<div_up_u(particleRangeEnd, 128), 128, 0, (hipStream_t)stream>>>(ctx->dev, a);
	SPHX_LAUNCH_CHECK("sa_density_diffusion_kernel<open>");
	return SPHX_OK;
}
