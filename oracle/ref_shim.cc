// ref_shim.cc -- exports, with C linkage, the parts of the GPUSPH reference that compile
// AS THEY ARE (sources included from /root/reference where they lie, nothing copied, no
// stand-ins for missing headers / intrinsics / generated files) with g++ and the CUDA
// headers that ship inside this image's triton package:
//   src/cuda/sph_core.cu      W<>, F<> for all four kernel types
//   src/particleinfo.h        id(), PART_TYPE, flag predicates, object(), fluid_num()
//   src/hashkey.h, src/multi_gpu_defines.h   cellHashFromParticleHash, CELLTYPE_* constants
//   src/common_types.h        ENCODE_CELL / DECODE_CELL / NEIBINDEX_MASK / NEIBS_END
//   src/cuda/visc_avg.cu      visc_avg<FullViscSpec<...>> for every computational viscosity / averaging / constness
//   src/physparams.h, src/simparams.h   host parameter structs: defaults, EOS / viscosity setters, smoothing + influence radii
//   src/vector_math.h         float4 / float (= multiply by the reciprocal), dot3, length: operation order of the filters
//   src/utils.h               div_up / round_up: the arithmetic of getFmaxElements / round_particles (src/cuda/forces.cu:539-552,960-964)
//   src/predcorr_alloc_policy.{h,cc}   which buffers the predictor-corrector scheme double-buffers
//   src/timing.h              IPPSCounter: the definition of the reported metric (iterations x particles per second)
//   src/cuda/gamma.cuh        SA_BOUNDARY: wendlandOnSegment, the Gauss quadratures, calcVertexRelPos, gradGamma<WENDLAND>,
//                             Gamma<WENDLAND, PT_FLUID|PT_VERTEX> (host math library instead of the device one)
// (src/cuda/phys_core.cu does not link: its non-inline R() drags in the nvcc intrinsic __powf, which has no host definition.)
// Everything else on the hot path needs nvcc (__powf, texture references, thrust) or the
// Makefile-generated options/*.opt files and is therefore NOT built (DESIGN.md "Oracle").
// TEST INFRASTRUCTURE ONLY: used to pin oracle/sph_oracle.c and to generate tests/golden/ref_*.npz.
#include <cuda_runtime.h>
#include <math.h>
#include <climits>
#include "particledefine.h"
#include "sph_core.cu"
#include "visc_avg.cu"       // visc_avg<ViscSpec> in all its specialisations (+ average.h, visc_spec.h)
#include "physparams.h"      // PhysParams: defaults, add_fluid, set_equation_of_state, set_kinematic_visc / set_dynamic_visc
#include "simparams.h"       // SimParams: defaults, set_smoothing / set_kernel_radius / set_influenceradius
#include "vector_math.h"
#include "utils.h"
#include "predcorr_alloc_policy.cc"
#include "timing.h"
#include "gamma.cuh"
#include <thread>
#include <chrono>

// visc_avg of the reference for a Newtonian laminar MORRIS spec: compvisc 0/1 (KINEMATIC/DYNAMIC), avgop 0/1/2, is_const 0/1
template<ComputationalViscosityType cv, AverageOperator av, bool cst>
static float ref_visc_avg_t(float visc, float neib_visc, float rho, float neib_rho, float neib_mass)
{
	// non-constant viscosity = a multi-fluid framework: with ENABLE_NONE the kinematic non-constant case would re-derive
	// is_const_visc = true through with_computational_visc<DYNAMIC> (src/visc_spec.h:268-272,298-300)
	using Spec = FullViscSpec<NEWTONIAN, LAMINAR_FLOW, cv, MORRIS, av, cst ? ENABLE_NONE : ENABLE_MULTIFLUID, cst>;
	return visc_avg<Spec>(visc, neib_visc, rho, neib_rho, neib_mass);
}

// client code, as a Problem / framework would write it: access to PhysParams' protected fluid setters, and the static
// option members SimParams' constructor template reads from its framework type (src/simparams.h:261-274)
struct ShimPhysParams : PhysParams {
	ShimPhysParams() : PhysParams(NEWTONIAN) {}
	using PhysParams::add_fluid; using PhysParams::set_equation_of_state;
	using PhysParams::set_kinematic_visc; using PhysParams::set_dynamic_visc;
};
template<KernelType K> struct ShimFramework {
	static constexpr KernelType kerneltype = K;
	static constexpr SPHFormulation sph_formulation = SPH_F1;
	static constexpr DensityDiffusionType densitydiffusiontype = DENSITY_DIFFUSION_NONE;
	static constexpr RheologyType rheologytype = INVISCID;
	static constexpr TurbulenceModel turbmodel = ARTIFICIAL;
	static constexpr ComputationalViscosityType compvisc = KINEMATIC;
	static constexpr ViscousModel viscmodel = MORRIS;
	static constexpr AverageOperator viscavgop = ARITHMETIC;
	static constexpr bool is_const_visc = false;
	static constexpr BoundaryType boundarytype = DYN_BOUNDARY;
	static constexpr Periodicity periodicbound = PERIODIC_NONE;
	static constexpr flag_t simflags = ENABLE_DTADAPT;
};

extern "C" {

float ref_W(int kerneltype, float r, float slength, float coeff, float wsub_gaussian)
{
	switch (kerneltype) {
	case CUBICSPLINE: cusph::d_wcoeff_cubicspline = coeff; return cusph::W<CUBICSPLINE>(r, slength);
	case QUADRATIC:   cusph::d_wcoeff_quadratic = coeff;   return cusph::W<QUADRATIC>(r, slength);
	case WENDLAND:    cusph::d_wcoeff_wendland = coeff;    return cusph::W<WENDLAND>(r, slength);
	case GAUSSIAN:    cusph::d_wcoeff_gaussian = coeff; cusph::d_wsub_gaussian = wsub_gaussian;
	                  return cusph::W<GAUSSIAN>(r, slength);
	}
	return NAN;
}

float ref_F(int kerneltype, float r, float slength, float coeff)
{
	switch (kerneltype) {
	case CUBICSPLINE: cusph::d_fcoeff_cubicspline = coeff; return cusph::F<CUBICSPLINE>(r, slength);
	case QUADRATIC:   cusph::d_fcoeff_quadratic = coeff;   return cusph::F<QUADRATIC>(r, slength);
	case WENDLAND:    cusph::d_fcoeff_wendland = coeff;    return cusph::F<WENDLAND>(r, slength);
	case GAUSSIAN:    cusph::d_fcoeff_gaussian = coeff;    return cusph::F<GAUSSIAN>(r, slength);
	}
	return NAN;
}

static particleinfo mk(unsigned short x, unsigned short y, unsigned short z, unsigned short w)
{ particleinfo i; i.x = x; i.y = y; i.z = z; i.w = w; return i; }

unsigned ref_info_id(unsigned short x, unsigned short y, unsigned short z, unsigned short w) { return id(mk(x,y,z,w)); }
int ref_info_part_type(unsigned short x, unsigned short y, unsigned short z, unsigned short w) { return PART_TYPE(mk(x,y,z,w)); }
int ref_info_object(unsigned short x, unsigned short y, unsigned short z, unsigned short w) { return object(mk(x,y,z,w)); }
int ref_info_fluid_num(unsigned short x, unsigned short y, unsigned short z, unsigned short w) { return fluid_num(mk(x,y,z,w)); }
// bit0 FLUID, bit1 BOUNDARY, bit2 VERTEX, bit3 TESTPOINT, bit4 MOVING, bit5 FLOATING, bit6 COMPUTE_FORCE, bit7 SURFACE
unsigned ref_info_predicates(unsigned short x, unsigned short y, unsigned short z, unsigned short w)
{
	const particleinfo f = mk(x,y,z,w);
	return (FLUID(f) ? 1u : 0) | (BOUNDARY(f) ? 2u : 0) | (VERTEX(f) ? 4u : 0) | (TESTPOINT(f) ? 8u : 0) |
		(MOVING(f) ? 16u : 0) | (FLOATING(f) ? 32u : 0) | (COMPUTE_FORCE(f) ? 64u : 0) | (SURFACE(f) ? 128u : 0);
}
int ref_active(float w) { float4 p = make_float4(0, 0, 0, w); return ACTIVE(p) ? 1 : 0; }

unsigned ref_cell_hash_from_particle_hash(unsigned h, int preserve) { return cellHashFromParticleHash(h, preserve != 0); }
unsigned ref_encode_cell(unsigned cell) { return ENCODE_CELL(cell); }
int ref_decode_cell(unsigned data) { return DECODE_CELL(data); }
// 0 CELLTYPE_BITMASK, 1 CELL_HASH_MAX, 2 NEIBINDEX_MASK, 3 NEIBS_END, 4 CELLNUM_ENCODED,
// 5..8 CELLTYPE_*_SHIFTED (inner, inner edge, outer edge, outer), 9 EMPTY_SEGMENT, 10 MAX_CELLS
unsigned ref_constant(int which)
{
	switch (which) {
	case 0: return CELLTYPE_BITMASK;
	case 1: return CELL_HASH_MAX;
	case 2: return NEIBINDEX_MASK;
	case 3: return NEIBS_END;
	case 4: return CELLNUM_ENCODED;
	case 5: return CELLTYPE_INNER_CELL_SHIFTED;
	case 6: return CELLTYPE_INNER_EDGE_CELL_SHIFTED;
	case 7: return CELLTYPE_OUTER_EDGE_CELL_SHIFTED;
	case 8: return CELLTYPE_OUTER_CELL_SHIFTED;
	case 9: return EMPTY_SEGMENT;
	case 10: return MAX_CELLS;
	}
	return 0;
}
// enum values, to pin the numeric option codes of include/sphx.h
// 0..3 kernels, 4 SPH_F1, 5 COLAGROSSI, 6 DYN_BOUNDARY, 7 LJ_BOUNDARY, 8 PERIODIC_Z
int ref_enum(int which)
{
	switch (which) {
	case 0: return CUBICSPLINE; case 1: return QUADRATIC; case 2: return WENDLAND; case 3: return GAUSSIAN;
	case 4: return SPH_F1; case 5: return COLAGROSSI; case 6: return DYN_BOUNDARY; case 7: return LJ_BOUNDARY;
	case 8: return PERIODIC_Z; case 9: return SA_BOUNDARY; case 10: return FERRARI;
	// codes of the widened rows: filters, post-processing, turbulence models, run modes, flags, limits
	case 11: return SHEPARD_FILTER; case 12: return MLS_FILTER;
	case 13: return VORTICITY; case 14: return TESTPOINTS; case 15: return SURFACE_DETECTION;
	case 16: return ARTIFICIAL; case 17: return SPS; case 18: return LAMINAR_FLOW;
	case 19: return MK_BOUNDARY; case 20: return (int)ENABLE_PLANES; case 21: return (int)ENABLE_DTADAPT;
	case 22: return MAX_PLANES; case 23: return MAX_FLUID_TYPES; case 24: return (int)FG_SURFACE;
	case 25: return PT_TESTPOINT; case 26: return INVISCID;
	case 27: return (int)ENABLE_MULTIFLUID; case 28: return (int)ENABLE_REPACKING; case 29: return NEWTONIAN;
	case 30: return KINEMATIC; case 31: return DYNAMIC; case 32: return MORRIS; case 33: return ARITHMETIC;
	case 34: return HARMONIC; case 35: return GEOMETRIC; case 36: return REPACK; case 37: return SIMULATE;
	case 38: return INTERFACE_DETECTION; case 39: return FG_INTERFACE;
	}
	return -1;
}


// the same for a SINGLE-fluid framework (ENABLE_NONE) whose viscosity is forced non-constant (assume_const_visc<false>), kinematic
float ref_visc_avg_singlefluid_nonconst_kinematic(int avgop, float visc, float neib_visc, float rho, float neib_rho, float neib_mass)
{
	switch (avgop) {
	case 0: return visc_avg<FullViscSpec<NEWTONIAN, LAMINAR_FLOW, KINEMATIC, MORRIS, ARITHMETIC, ENABLE_NONE, false>>(visc, neib_visc, rho, neib_rho, neib_mass);
	case 1: return visc_avg<FullViscSpec<NEWTONIAN, LAMINAR_FLOW, KINEMATIC, MORRIS, HARMONIC, ENABLE_NONE, false>>(visc, neib_visc, rho, neib_rho, neib_mass);
	case 2: return visc_avg<FullViscSpec<NEWTONIAN, LAMINAR_FLOW, KINEMATIC, MORRIS, GEOMETRIC, ENABLE_NONE, false>>(visc, neib_visc, rho, neib_rho, neib_mass);
	}
	return NAN;
}
float ref_visc_avg(int compvisc, int avgop, int is_const, float visc, float neib_visc, float rho, float neib_rho, float neib_mass)
{
#define RVA(cv, av, cst) if (compvisc == cv && avgop == av && is_const == cst) \
		return ref_visc_avg_t<(ComputationalViscosityType)cv, (AverageOperator)av, (bool)cst>(visc, neib_visc, rho, neib_rho, neib_mass);
	RVA(0,0,0) RVA(0,1,0) RVA(0,2,0) RVA(1,0,0) RVA(1,1,0) RVA(1,2,0)
	RVA(0,0,1) RVA(0,1,1) RVA(0,2,1) RVA(1,0,1) RVA(1,1,1) RVA(1,2,1)
#undef RVA
	return NAN;
}

// PhysParams as the reference builds them: one fluid (rho0, gamma, c0, nu).  out[16]
void ref_physparams(float rho0, float gamma, float c0, float nu, float mu_second, float *out)
{
	ShimPhysParams pp;
	const size_t f = pp.add_fluid(rho0);
	pp.set_equation_of_state(f, gamma, c0);
	pp.set_kinematic_visc(f, nu);
	out[0] = pp.bcoeff[f]; out[1] = pp.gammacoeff[f]; out[2] = pp.sscoeff[f]; out[3] = pp.sspowercoeff[f];
	out[4] = pp.kinematicvisc[f]; out[5] = pp.visc_consistency[f];
	const size_t g = pp.add_fluid(rho0);
	pp.set_dynamic_visc(g, mu_second);
	out[6] = pp.kinematicvisc[g]; out[7] = pp.visc_consistency[g];
	out[8] = pp.artvisccoeff; out[9] = pp.p1coeff; out[10] = pp.p2coeff; out[11] = pp.MK_beta; out[12] = pp.partsurf;
	out[13] = pp.smagorinsky_constant; out[14] = pp.isotropic_sps_constant; out[15] = pp.cosconeanglefluid;
	out[16] = pp.cosconeanglenonfluid; out[17] = pp.gravity.z;
}
// SimParams: defaults and the smoothing / influence-radius arithmetic.  out[12] (doubles)
void ref_simparams(int gaussian, double sfactor, double deltap, double *out)
{
	ShimFramework<WENDLAND> fw; ShimFramework<GAUSSIAN> fg;
	SimParams sp = gaussian ? SimParams(&fg) : SimParams(&fw);
	out[0] = sp.sfactor; out[1] = sp.kernelradius; out[2] = sp.buildneibsfreq; out[3] = sp.dtadaptfactor;
	out[4] = sp.repack_maxiter; out[5] = sp.repack_a; out[6] = sp.repack_alpha; out[7] = sp.nlexpansionfactor;
	sp.set_smoothing(sfactor, deltap);
	out[8] = sp.slength; out[9] = sp.influenceRadius; out[10] = sp.nlInfluenceRadius; out[11] = sp.nlSqInfluenceRadius;
}

// out[4] = (x,y,z,w)/s with the reference's operator/ ; returns dot3 of the inputs' xyz; out[4] = length of xyz
float ref_float4_div(float x, float y, float z, float w, float s, float *out)
{
	const float4 v = make_float4(x, y, z, w);
	const float4 q = v/s;
	out[0] = q.x; out[1] = q.y; out[2] = q.z; out[3] = q.w;
	out[4] = length(make_float3(x, y, z));
	return dot3(v, v);
}
unsigned ref_div_up(unsigned a, unsigned b) { return div_up(a, b); }
unsigned ref_round_up(unsigned a, unsigned b) { return round_up(a, b); }
// number of copies the predictor-corrector allocation policy keeps of a buffer (flag_t key)
unsigned ref_predcorr_buffer_count(unsigned long long key) { PredCorrAllocPolicy p; return (unsigned)p.get_buffer_count((flag_t)key); }
unsigned long long ref_predcorr_multi_buffered(unsigned long long keys) { PredCorrAllocPolicy p; return p.get_multi_buffered((flag_t)keys); }
// buffer keys by name index: 0 POS 1 VEL 2 INFO 3 HASH 4 PARTINDEX 5 CELLSTART 6 CELLEND 7 NEIBSLIST 8 FORCES 9 TAU 10 CFL 11 XSPH
// 12 RB_FORCES 13 SPS_TURBVISC 14 VORTICITY 15 NORMALS 16 COMPACT_DEV_MAP 17 CFL_TEMP 18 RB_TORQUES 19 RB_KEYS
unsigned long long ref_buffer_key(int which)
{
	static const flag_t keys[] = { BUFFER_POS, BUFFER_VEL, BUFFER_INFO, BUFFER_HASH, BUFFER_PARTINDEX, BUFFER_CELLSTART, BUFFER_CELLEND,
		BUFFER_NEIBSLIST, BUFFER_FORCES, BUFFER_TAU, BUFFER_CFL, BUFFER_XSPH, BUFFER_RB_FORCES, BUFFER_SPS_TURBVISC, BUFFER_VORTICITY,
		BUFFER_NORMALS, BUFFER_COMPACT_DEV_MAP, BUFFER_CFL_TEMP, BUFFER_RB_TORQUES, BUFFER_RB_KEYS };
	return (which >= 0 && which < (int)(sizeof(keys)/sizeof(keys[0]))) ? keys[which] : 0;
}
// the metric: `increments` calls of incItersTimesParts(particles) spread over ~millis ms; out = {MIPPS, elapsed seconds}
void ref_ipps(unsigned long particles, int increments, int millis, double *out)
{
	IPPSCounter c;
	c.start();
	for (int i = 0; i < increments; ++i) {
		std::this_thread::sleep_for(std::chrono::microseconds(1000L*millis/increments));
		c.incItersTimesParts(particles);
	}
	out[0] = c.getMIPPS();
	out[1] = c.getElapsedSeconds();
}
// ---- src/cuda/gamma.cuh: vectors as float[3], q_vb as float[9] (three vertices) ----
float ref_wendlandOnSegment(float q) { return wendlandOnSegment(q); }
float ref_gaussQuadratureO5(const float *v0, const float *v1, const float *v2, const float *rel)
{
	return gaussQuadratureO5(make_float3(v0[0], v0[1], v0[2]), make_float3(v1[0], v1[1], v1[2]), make_float3(v2[0], v2[1], v2[2]),
		make_float3(rel[0], rel[1], rel[2]));
}
void ref_calcVertexRelPos(const float *ns, const float *vp0, const float *vp1, const float *vp2, float slength, float *out9)
{
	float3 q_vb[3];
	calcVertexRelPos(q_vb, make_float3(ns[0], ns[1], ns[2]), make_float2(vp0[0], vp0[1]), make_float2(vp1[0], vp1[1]),
		make_float2(vp2[0], vp2[1]), slength);
	for (int i = 0; i < 3; ++i) { out9[3*i] = q_vb[i].x; out9[3*i + 1] = q_vb[i].y; out9[3*i + 2] = q_vb[i].z; }
}
float ref_gradGamma(float slength, const float *q, const float *qvb9, const float *ns)
{
	float3 q_vb[3];
	for (int i = 0; i < 3; ++i) q_vb[i] = make_float3(qvb9[3*i], qvb9[3*i + 1], qvb9[3*i + 2]);
	return gradGamma<WENDLAND>(slength, make_float3(q[0], q[1], q[2]), q_vb, make_float3(ns[0], ns[1], ns[2]));
}
float ref_Gamma(int vertex, float slength, const float *q, const float *qvb9, const float *ns, const float *oldGGam, float epsilon)
{
	float3 q_vb[3];
	for (int i = 0; i < 3; ++i) q_vb[i] = make_float3(qvb9[3*i], qvb9[3*i + 1], qvb9[3*i + 2]);
	const float3 q3 = make_float3(q[0], q[1], q[2]), n3 = make_float3(ns[0], ns[1], ns[2]), g3 = make_float3(oldGGam[0], oldGGam[1], oldGGam[2]);
	return vertex ? Gamma<WENDLAND, PT_VERTEX>(slength, q3, q_vb, n3, g3, epsilon) : Gamma<WENDLAND, PT_FLUID>(slength, q3, q_vb, n3, g3, epsilon);
}
} // extern "C"
