/*
 * sphx.h -- C ABI of the MI355X-native WCSPH timestep engine (libsphx.so).
 *
 * This is the drop-in boundary for GPUSPH's per-step hot path.  Each entry point replaces
 * one virtual of the reference's abstract engines (paths relative to the GPUSPH tree):
 *
 *   AbstractNeibsEngine        src/engine_neibs.h:46-107
 *   AbstractForcesEngine       src/engine_forces.h:43-180
 *   AbstractViscEngine         src/engine_visc.h:42-109
 *   AbstractIntegrationEngine  src/engine_integration.h:42-144
 *
 * The reference selects physics at compile time through template arguments
 * (CUDASimFramework<kernel<>, boundary<>, ...>, src/cuda/cudasimframework.cu:379-606); here
 * the same options travel in the run-time POD `sphx_params` and select pre-compiled kernel
 * specialisations inside the library.
 *
 * Conventions
 *   - plain pointers and sizes only; every array pointer is a DEVICE pointer unless the name
 *     starts with h_.  Element types follow src/define_buffers.h:48-235:
 *       pos, vel, forces, rbforces, rbtorques : float4     info : ushort4 (particleinfo)
 *       hash, partIndex, cellStart, cellEnd   : uint32     neibsList : uint16 (neibdata)
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls are
 *     stream-ordered and do not synchronise the host unless documented ("sync").
 *   - one sphx_ctx per device (the per-device __constant__ state of the reference); a ctx may
 *     be used from one host thread at a time, different ctxs concurrently (GPUWorker model,
 *     src/GPUWorker.cc:3238-3331).
 *   - every function returns SPHX_OK (0) or a negative status; sphx_last_error() gives the
 *     message of the last failure on the calling thread.  The C++ adapters in
 *     gpusph_amd/host/ rethrow it as std::runtime_error / std::invalid_argument like
 *     CUDA_SAFE_CALL / KERNEL_CHECK_ERROR do (src/cuda/cuda_call.h:57-85).
 *   - callers clobber output buffers exactly as GPUWorker does (FORCES/CFL = 0,
 *     NEIBSLIST/CELLSTART/CELLEND = 0xFF; src/GPUWorker.cc:1847,1885,1949-1979).
 */
#ifndef SPHX_H
#define SPHX_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPHX_OK              0
#define SPHX_ERR_INVALID    -1   /* inconsistent arguments (std::invalid_argument in the reference) */
#define SPHX_ERR_RUNTIME    -2   /* HIP runtime / launch failure (std::runtime_error) */
#define SPHX_ERR_UNSUPPORTED -3  /* option combination not built into this library */

#define SPHX_MAX_FLUIDS 4        /* MAX_FLUID_TYPES, src/particledefine.h:327 */
#define SPHX_MAX_BODIES 16       /* MAX_BODIES, src/particledefine.h:329 */

/* Option codes carry the reference's enum values (src/particledefine.h:79-224, src/visc_spec.h) */
enum sphx_kernel      { SPHX_CUBICSPLINE = 1, SPHX_QUADRATIC = 2, SPHX_WENDLAND = 3, SPHX_GAUSSIAN = 4 };
enum sphx_formulation { SPHX_SPH_F1 = 1, SPHX_SPH_F2 = 2, SPHX_SPH_GRENIER = 3, SPHX_SPH_HA = 4 };
enum sphx_densitydiff { SPHX_DENSITY_DIFFUSION_NONE = 0, SPHX_FERRARI = 1, SPHX_COLAGROSSI = 2, SPHX_BREZZI = 3 };
enum sphx_boundary    { SPHX_LJ_BOUNDARY = 0, SPHX_MK_BOUNDARY = 1, SPHX_SA_BOUNDARY = 2, SPHX_DYN_BOUNDARY = 3 };
enum sphx_rheology    { SPHX_INVISCID = 0, SPHX_NEWTONIAN = 1, SPHX_GRANULAR = 2, SPHX_BINGHAM = 3, SPHX_PAPANASTASIOU = 4, SPHX_POWER_LAW = 5,
	SPHX_HERSCHEL_BULKLEY = 6, SPHX_ALEXANDROU = 7, SPHX_DEKEE_TURCOTTE = 8, SPHX_ZHU = 9 };   /* RheologyType, src/visc_spec.h:44-56 */
enum sphx_turbulence  { SPHX_LAMINAR_FLOW = 0, SPHX_ARTIFICIAL = 1, SPHX_SPS = 2, SPHX_KEPSILON = 3 };
enum sphx_compvisc    { SPHX_KINEMATIC = 0, SPHX_DYNAMIC = 1 };                          /* src/visc_spec.h */
enum sphx_viscmodel   { SPHX_MORRIS = 0, SPHX_MONAGHAN = 1, SPHX_ESPANOL_REVENGA = 2 };
enum sphx_avgop       { SPHX_ARITHMETIC = 0, SPHX_HARMONIC = 1, SPHX_GEOMETRIC = 2 };      /* src/average.h */
enum sphx_runmode     { SPHX_REPACK = 0, SPHX_SIMULATE = 1 };
enum sphx_filter      { SPHX_SHEPARD_FILTER = 0, SPHX_MLS_FILTER = 1 };   /* FilterType, src/particledefine.h:255-260 */
enum sphx_postproc    { SPHX_VORTICITY = 0, SPHX_TESTPOINTS = 1, SPHX_SURFACE_DETECTION = 2, SPHX_INTERFACE_DETECTION = 3 };   /* PostProcessType, :290-299 */
#define SPHX_PERIODIC_X 1u
#define SPHX_PERIODIC_Y 2u
#define SPHX_PERIODIC_Z 4u
/* simflags, src/simflags.h:62-160 */
#define SPHX_ENABLE_DTADAPT        (1ull << 0)
#define SPHX_ENABLE_XSPH           (1ull << 1)
#define SPHX_ENABLE_PLANES         (1ull << 2)
#define SPHX_ENABLE_DEM            (1ull << 3)
#define SPHX_ENABLE_MOVING_BODIES  (1ull << 4)
#define SPHX_ENABLE_INLET_OUTLET   (1ull << 5)
#define SPHX_ENABLE_WATER_DEPTH    (1ull << 6)
#define SPHX_ENABLE_DENSITY_SUM    (1ull << 7)
#define SPHX_ENABLE_GAMMA_QUADRATURE (1ull << 8)
#define SPHX_ENABLE_REPACKING      (1ull << 9)
#define SPHX_ENABLE_INTERNAL_ENERGY (1ull << 10)
#define SPHX_ENABLE_MULTIFLUID     (1ull << 11)

/* Everything the three setconstants() upload (src/cuda/forces.cu:268-399,
 * src/cuda/buildneibs.cu:85-96, src/cuda/euler.cu:51-69), as one POD. */
typedef struct sphx_params {
	/* grid: d_worldOrigin, d_cellSize, d_gridSize (src/cuda/cellgrid.cuh:63-66) */
	uint32_t gridSize[3];
	float    cellSize[3];
	float    worldOrigin[3];
	int32_t  coord[3];            /* axis (0=x,1=y,2=z) of COORD1,COORD2,COORD3 (src/linearization.h); yzx = {1,2,0} */
	uint32_t periodic;            /* Periodicity bits */
	/* neighbour list geometry: d_neiblistsize, d_neibboundpos, d_neiblist_stride */
	uint32_t neiblistsize;
	uint32_t neibboundpos;
	uint64_t neiblist_stride;     /* = allocated particles */
	/* framework options (template parameters in the reference) */
	int32_t  kerneltype, sph_formulation, densitydiffusiontype, boundarytype;
	int32_t  rheologytype, turbmodel, compvisc, viscmodel, avgop;
	uint64_t simflags;
	/* SimParams */
	float    slength, kernelradius, influenceradius, deltap, dtadaptfactor;
	float    densityDiffCoeff;    /* Colagrossi: xi*2h (src/ProblemCore.cc:1406-1416) */
	float    epsxsph;
	/* PhysParams */
	uint32_t numfluids;
	float    rho0[SPHX_MAX_FLUIDS], bcoeff[SPHX_MAX_FLUIDS], gammacoeff[SPHX_MAX_FLUIDS];
	float    sscoeff[SPHX_MAX_FLUIDS], sspowercoeff[SPHX_MAX_FLUIDS], visccoeff[SPHX_MAX_FLUIDS];
	float    gravity[3];
	float    artvisccoeff, epsartvisc;
	float    smagfactor, kspsfactor;
	float    dcoeff, p1coeff, p2coeff, r0;
	/* repacking (src/simparams.h:220-234): mixing intensity a, velocity damping alpha */
	float    repack_a, repack_alpha;
	/* Newtonian rheology: visccoeff[] holds nu (compvisc KINEMATIC) or mu (DYNAMIC) per fluid (GPUSPH.cc:1486-1502);
	 * is_const_visc = FullViscSpec::is_const_visc (src/visc_spec.h:265-282, forced true by KINEMATICVISC/SPSVISC);
	 * partsurf = particle surface for the wall friction of planes, 0 -> r0^2 (src/cuda/forces.cu:364-368) */
	int32_t  is_const_visc;
	float    partsurf;
	/* Monaghan-Kajtar boundary repulsion (src/physparams.h:336-338) */
	float    MK_K, MK_d, MK_beta;
	/* SPH_GRENIER: interface pressure coefficient between different fluids (src/physparams.h epsinterface,
	 * src/ProblemCore.cc:165-166 default 0.05; d_epsinterface src/cuda/forces_kernel.def:2237) */
	float    epsinterface;
	/* generalized Newtonian rheologies (rheologytype BINGHAM .. ZHU, src/visc_spec.h:44-56): per fluid the yield strength,
	 * the power-law exponent or exponential coefficient, the regularisation parameter m (src/physparams.h:185-243,
	 * uploaded by src/cuda/forces.cu:329-333); visccoeff[] then holds the consistency index (GPUSPH.cc:1503-1508);
	 * limiting_kinvisc clamps the effective viscosity (x rho0) */
	float    yield_strength[SPHX_MAX_FLUIDS], visc_nonlinear_param[SPHX_MAX_FLUIDS], visc_regularization_param[SPHX_MAX_FLUIDS];
	float    limiting_kinvisc;
	/* ENABLE_DEM (terrain as a height map, src/physparams.h:322-327, uploaded by src/cuda/forces.cu:353-359): cell sizes of the
	 * DEM, displacements used for the tangent plane, height above the terrain below which the terrain repels; the map itself
	 * goes through sphx_set_dem */
	float    ewres, nsres, demdx, demdy, demzmin;
	/* visc_model<MONAGHAN>: the coefficient of the (r.v)/(r.r) r form (src/physparams.h:266, default 10);
	 * visc_model<ESPANOL_REVENGA>: the bulk viscosity per fluid (d_visc2coeff, src/GPUSPH.cc:1511-1522) */
	float    monaghan_visc_coeff;
	float    visc2coeff[SPHX_MAX_FLUIDS];
} sphx_params;

/* TimingInfo fields filled by getinfo (src/timing.h:43-100, src/cuda/buildneibs.cu:137-145) */
typedef struct sphx_neibs_info {
	int32_t numInteractions;
	int32_t maxFluidBoundaryNeibs;
	int32_t maxVertexNeibs;
	int32_t hasTooManyNeibs;
	int32_t hasMaxNeibs[3];
} sphx_neibs_info;

typedef struct sphx_ctx sphx_ctx;

/* ---- lifetime / errors ------------------------------------------------------------------ */
const char *sphx_last_error(void);
const char *sphx_version(void);
/* creates the per-device state on HIP device `device` (hipSetDevice is called) */
int  sphx_create(sphx_ctx **out, int device);
void sphx_destroy(sphx_ctx *ctx);
/* pre-allocates internal scratch (sort bins, scan partials) so that no allocation happens
 * inside the step (required before stream capture into a hipGraph) */
int  sphx_reserve(sphx_ctx *ctx, uint32_t maxParticles);

/* ---- constants: {Neibs,Forces,Integration}Engine::setconstants --------------------------- */
int sphx_set_constants(sphx_ctx *ctx, const sphx_params *params);
int sphx_get_params(sphx_ctx *ctx, sphx_params *out);
/* AbstractForcesEngine::setgravity (src/engine_forces.h) */
int sphx_set_gravity(sphx_ctx *ctx, const float h_gravity[3]);
/* setplanes (src/engine_forces.h, CUDAForcesEngine::setplanes src/cuda/forces.cu:442-448): up to SPHX_MAX_PLANES
 * geometric planes, each a unit normal and a reference point given as grid cell + cell-local position
 * (plane_t, src/planes.h:43-47); arrays of 3*numPlanes values.  Used when SPHX_ENABLE_PLANES is set. */
#define SPHX_MAX_PLANES 8
int sphx_set_planes(sphx_ctx *ctx, const float *normals, const int32_t *gridPos, const float *pos, int numPlanes);
/* AbstractForcesEngine::setDEM / unsetDEM (src/engine_forces.h:103-110, src/cuda/forces.cu:937-958): the terrain height map of
 * ENABLE_DEM, width x height floats, row-major (what the reference copies into its 2D texture), taken from HOST memory;
 * hDem = NULL drops it.  Read by the finalize step of sphx_forces_basicstep with LJ_BOUNDARY (DemLJForce). */
int sphx_set_dem(sphx_ctx *ctx, const float *hDem, int width, int height);
/* AbstractForcesEngine::setrbcg / setrbstart; AbstractIntegrationEngine::setrbcg/setrbtrans/
 * setrbsteprot/setrblinearvel/setrbangularvel (src/engine_integration.h) */
int sphx_set_rb_cg(sphx_ctx *ctx, const int32_t *h_cgGridPos3, const float *h_cgPos3, int numbodies);
/* the two engines keep their own copy of the centres of gravity and the reference uploads them at different points of
 * a step (FORCES_UPLOAD_OBJECTS_CG after MOVE_BODIES, EULER_UPLOAD_OBJECTS_CG after the corrector,
 * src/integrators/PredictorCorrectorIntegrator.cc:331-332,566-570): sphx_set_rb_cg sets both, these set one */
int sphx_set_rb_cg_forces(sphx_ctx *ctx, const int32_t *h_cgGridPos3, const float *h_cgPos3, int numbodies);
int sphx_set_rb_cg_integration(sphx_ctx *ctx, const int32_t *h_cgGridPos3, const float *h_cgPos3, int numbodies);
int sphx_set_rb_start(sphx_ctx *ctx, const int32_t *h_rbfirstindex, int numbodies);
int sphx_set_rb_motion(sphx_ctx *ctx, const float *h_trans3, const float *h_steprot9,
	const float *h_linearvel3, const float *h_angularvel3, int numbodies);

/* ---- AbstractNeibsEngine ------------------------------------------------------------------ */
/* calcHash (src/cuda/buildneibs.cu:157-175): pos and hash updated in place, partIndex[i]=i */
int sphx_calc_hash(sphx_ctx *ctx, void *pos, uint32_t *hash, uint32_t *partIndex,
	const void *info, const uint32_t *compactDeviceMap, uint32_t numParticles, void *stream);
/* fixHash (src/cuda/buildneibs.cu:182-197) */
int sphx_fix_hash(sphx_ctx *ctx, uint32_t *hash, uint32_t *partIndex,
	const void *info, const uint32_t *compactDeviceMap, uint32_t numParticles, void *stream);
/* sort (src/cuda/buildneibs.cu:384-412): sorts (hash, info) keys and partIndex values in place
 * by (hash incl. cell-type bits, PART_TYPE, id) */
int sphx_sort(sphx_ctx *ctx, uint32_t *hash, void *info, uint32_t *partIndex,
	uint32_t numParticles, void *stream);
/* reorderDataAndFindCellStart (src/cuda/buildneibs.cu:213-354).  segmentStart (4 uint) and
 * newNumParticles (1 uint) are device pointers; segmentStart may be NULL. */
int sphx_reorder(sphx_ctx *ctx, uint32_t *segmentStart,
	uint32_t *cellStart, uint32_t *cellEnd,
	void *sortedPos, void *sortedVel,
	const void *unsortedPos, const void *unsortedVel,
	const void *sortedInfo, const uint32_t *sortedHash, const uint32_t *partIndex,
	uint32_t numParticles, uint32_t *newNumParticles, void *stream);
/* the optional per-particle arrays reorderDataAndFindCellStart gathers when the buffer lists hold them
 * (src/cuda/buildneibs.cu:263-311): sorted[i] = unsorted[partIndex[i]], rows of 4, 8 or 16 bytes */
int sphx_gather_rows(sphx_ctx *ctx, void *sorted, const void *unsorted, uint32_t rowBytes,
	const uint32_t *partIndex, uint32_t numParticles, void *stream);
/* cellStart/cellEnd of the already sorted range [fromParticle, toParticle) only (no gather).  The
 * reference rebuilds the cell ranges of imported halo cells on the host from the neighbour
 * device's s_dCellStarts (src/GPUWorker.cc:754-776,1391-1430); with the halo arriving in sorted
 * order this is the same adjacent-hash scan as reorderDataAndFindCellStart, run on the device. */
int sphx_find_cell_start(sphx_ctx *ctx, uint32_t *cellStart, uint32_t *cellEnd, const uint32_t *sortedHash,
	uint32_t fromParticle, uint32_t toParticle, void *stream);
/* buildNeibsList (src/cuda/buildneibs.cu:421-492) */
int sphx_build_neibs(sphx_ctx *ctx, uint16_t *neibsList,
	const void *pos, const void *info, const uint32_t *hash,
	const uint32_t *cellStart, const uint32_t *cellEnd,
	uint32_t numParticles, uint32_t particleRangeEnd, uint32_t gridCells,
	float sqinfluenceradius, float boundNlSqInflRad, void *stream);
/* the same with the SA_BOUNDARY members of buildneibs_params (src/cuda/buildneibs_params.h:66-115): BUFFER_VERTICES (uint4)
 * and BUFFER_BOUNDELEMENTS (float4) are read, the three arrays of BUFFER_VERTPOS (float2) are written for the segments,
 * boundary neighbours are kept out to boundNlSqInflRad, vertex neighbours go to the third list section.  All five may be
 * NULL for the other boundary types (= sphx_build_neibs). */
int sphx_build_neibs_sa(sphx_ctx *ctx, uint16_t *neibsList, void *vertPos0, void *vertPos1, void *vertPos2,
	const void *pos, const void *info, const void *vertices, const void *boundElements, const uint32_t *hash,
	const uint32_t *cellStart, const uint32_t *cellEnd,
	uint32_t numParticles, uint32_t particleRangeEnd, uint32_t gridCells,
	float sqinfluenceradius, float boundNlSqInflRad, void *stream);
/* resetinfo / getinfo (src/cuda/buildneibs.cu:119-145); getinfo is sync */
int sphx_neibs_resetinfo(sphx_ctx *ctx, void *stream);
int sphx_neibs_getinfo(sphx_ctx *ctx, sphx_neibs_info *h_out, void *stream);
/* numInteractions of the last list build as a 64-bit sum (the reference's counter is a 32-bit uint, src/timing.h:63: it wraps
 * beyond ~66 M particles with ~65 neighbours each; as a signed int beyond ~33 M); sync */
int sphx_neibs_interactions64(sphx_ctx *ctx, uint64_t *h_out, void *stream);

/* The part of finalizeforcesDevice that belongs to the boundary elements of bodies with FG_COMPUTE_FORCE under SA_BOUNDARY
 * (compute_boundary_pressure_force, src/cuda/forces_kernel.def:3258-3266,4115-4145): F = -P(rho~) A n per element into
 * rbforces / rbtorques (BUFFER_RB_FORCES / BUFFER_RB_TORQUES rows id + rbstart[object], torque about the forces engine's centre of
 * gravity: sphx_set_rb_start, sphx_set_rb_cg_forces) and into the element's own row of `forces` (w = 0).  Call it after
 * sphx_forces_basicstep_sa[_keps|_io] of the same range when compute_object_forces is asked; sphx_reduce_rb_forces sums the rows
 * (reduceRbForces, src/cuda/forces.cu:966-1004).  CompleteSaExample's floating cube is the reference's user. */
int sphx_sa_body_pressure_forces(sphx_ctx *ctx, void *forces, void *rbforces, void *rbtorques,
	const void *pos, const void *vel, const void *info, const uint32_t *hash, const void *boundElements,
	uint32_t fromParticle, uint32_t toParticle, void *stream);
/* basicstep of the forces engine with SA_BOUNDARY (src/cuda/forces.cu:717-806 with the SA members of forces_params): fluid <-
 * fluid, fluid <- vertex, fluid <- boundary element (through |grad gamma_as|, src/cuda/gamma.cuh), sums divided by gamma,
 * gravity, CFL maxima.  Built for solid walls, SPH_F1, laminar Newtonian or inviscid flow, in two forms: the continuity
 * equation with ENABLE_GAMMA_QUADRATURE (StillWaterRepackSA's simulation) and ENABLE_DENSITY_SUM with dynamic gamma
 * (StillWaterSA; then cflGamma = BUFFER_CFL_GAMMA receives the CFL condition of the gamma transport, per particle and, behind
 * round_up(numParticles, 4), per block -- NULL otherwise); everything else answers SPHX_ERR_UNSUPPORTED. */
int sphx_forces_basicstep_sa(sphx_ctx *ctx, void *forces, float *cfl, float *cflGamma,
	const void *pos, const void *vel, const void *info, const uint32_t *hash, const uint32_t *cellStart, const uint16_t *neibsList,
	const void *gGam, const void *boundElements, const void *vertPos0, const void *vertPos1, const void *vertPos2,
	uint32_t numParticles, uint32_t fromParticle, uint32_t toParticle,
	float deltap, float slength, float dtadaptfactor, float influenceradius, uint32_t cflOffset,
	int run_mode, int step, float dt, uint32_t *h_numBlocks, void *stream);
/* gamma part of dtreduce with dynamic gamma (src/cuda/forces.cu:576-585): dt = min(dt, 0.001/max(max CFL_gamma, 1e-5/dt)), on a
 * device scalar (stream-ordered) or on a host value (synchronous) */
int sphx_forces_dtreduce_gamma_device(sphx_ctx *ctx, const float *cflGamma, uint32_t numParticles, uint32_t numBlocks,
	float *d_dt, void *stream);
int sphx_forces_dtreduce_gamma(sphx_ctx *ctx, const float *cflGamma, uint32_t numParticles, uint32_t numBlocks,
	float *h_dt_inout, void *stream);
/* SA_BOUNDARY with ENABLE_MOVING_BODIES (bodies with prescribed motion; row f-2 of SURVEY.md 8).  What a caller does differently:
 *   - BUFFER_BOUNDELEMENTS is double buffered; behind sphx_euler_basicstep, sphx_sa_update_normals writes that of the new state:
 *     the normals of the moving segments and vertices turned by the body's step rotation (update_normals,
 *     src/cuda/euler_kernel.def:237-254; the rotation is what sphx_set_rb_motion uploaded), the rest copied;
 *   - density summation: sphx_sa_density_sum_moving with both boundary-element buffers (density_sum_impl, src/cuda/euler.cu:112-160;
 *     the boundary terms take the elements where they were and where they are, src/cuda/density_sum_kernel.cu:455-470); gamma of
 *     the VERTEX rows is integrated, not copied; the BOUNDARY rows of newGGam are not written (the segment condition re-derives
 *     gamma in every step, boundary_conditions_kernel.cu:1467);
 *   - gamma by quadrature: sphx_sa_integrate_gamma with the boundary elements of the NEW state; the vertex rows are integrated too
 *     (integrate_gamma_impl, src/cuda/euler.cu:254-258);
 *   - sphx_sa_segment_bc gives a moving segment the mean velocity of its vertices (moving_vertex_contrib,
 *     boundary_conditions_kernel.cu:781-800); sphx_forces_basicstep_sa reads the elements' velocities from BUFFER_VEL as always.
 * Not built: SA bodies that feel the fluid (object forces with SA_BOUNDARY), moving bodies together with open boundaries. */
int sphx_sa_update_normals(sphx_ctx *ctx, void *newBoundElements, const void *oldBoundElements, const void *info,
	uint32_t numParticles, uint32_t particleRangeEnd, void *stream);
int sphx_sa_density_sum_moving(sphx_ctx *ctx, void *newVel, void *newGGam, void *forces,
	const void *oldPos, const void *newPos, const void *oldVel, const void *oldGGam,
	const void *oldBoundElements, const void *newBoundElements,
	const void *vertPos0, const void *vertPos1, const void *vertPos2, const void *info,
	const uint32_t *hash, const uint32_t *cellStart, const uint16_t *neibsList,
	uint32_t numParticles, uint32_t particleRangeEnd, void *stream);
/* AbstractIntegrationEngine::density_sum (src/cuda/euler.cu:112-200; densitySumVolumicDevice / densitySumBoundaryDevice,
 * src/cuda/density_sum_kernel.cu:523-655): density and (dynamic) gamma of the fluid from the old and the new positions;
 * newVel.w and newGGam are written, forces.w is scratch, gGam rows of vertex / boundary particles are copied */
int sphx_sa_density_sum(sphx_ctx *ctx, void *newVel, void *newGGam, void *forces,
	const void *oldPos, const void *newPos, const void *oldVel, const void *oldGGam, const void *boundElements,
	const void *vertPos0, const void *vertPos1, const void *vertPos2, const void *info,
	const uint32_t *hash, const uint32_t *cellStart, const uint16_t *neibsList,
	uint32_t numParticles, uint32_t particleRangeEnd, float dt, int step, float t, float epsilon,
	float deltap, float slength, float influenceradius, void *stream);
/* AbstractForcesEngine::compute_density_diffusion (src/cuda/forces.cu:621-661) for Brezzi diffusion with density summation:
 * the diffusive density rate into forces.w; AbstractIntegrationEngine::apply_density_diffusion (src/cuda/euler.cu:307-326) */
int sphx_sa_compute_density_diffusion(sphx_ctx *ctx, void *forces, const void *pos, const void *vel, const void *gGam,
	const void *info, const uint32_t *hash, const uint32_t *cellStart, const uint16_t *neibsList,
	uint32_t numParticles, uint32_t particleRangeEnd, float deltap, float slength, float influenceradius, float dt, void *stream);
int sphx_apply_density_diffusion(sphx_ctx *ctx, void *vel, const void *forces, const void *info,
	uint32_t numParticles, uint32_t particleRangeEnd, float dt, void *stream);
/* AbstractIntegrationEngine::integrate_gamma (src/cuda/euler.cu:202-290) with ENABLE_GAMMA_QUADRATURE: gamma and grad gamma
 * of the fluid particles at their new positions by quadrature over the listed boundary elements
 * (integrateGammaDevice, src/cuda/density_sum_kernel.cu:690-765); rows of vertex and boundary particles are copied */
int sphx_sa_integrate_gamma(sphx_ctx *ctx, void *newGGam, const void *oldGGam, const void *newPos,
	const void *boundElements, const void *vertPos0, const void *vertPos1, const void *vertPos2, const void *info,
	const uint32_t *hash, const uint32_t *cellStart, const uint16_t *neibsList,
	uint32_t numParticles, uint32_t particleRangeEnd, float dt, int step, float t,
	float epsilon, float slength, float influenceradius, int run_mode, void *stream);

/* ---- AbstractBoundaryConditionsEngine (src/engine_boundary_conditions.h:46-186), SA_BOUNDARY, solid walls -------- */
/* computeVertexNormal (src/cuda/boundary_conditions.cu:417-452): area-weighted mean normal of the segments adjacent to each
 * vertex, written to the vertex rows of boundElements (in place; w = NaN) */
int sphx_sa_compute_vertex_normal(sphx_ctx *ctx, void *boundElements, const void *vertices, const void *info,
	const uint32_t *hash, const uint32_t *cellStart, const uint16_t *neibsList,
	uint32_t numParticles, uint32_t particleRangeEnd, void *stream);
/* ---- the same engine, open boundaries (ENABLE_INLET_OUTLET; gpusph_amd/csrc/sa_io.hip: one wave per open vertex / segment /
 * leaving particle over an index of the pass): initialisation, outgoing particles, the two condition passes, the
 * water depth, the flux; the passes over all fluid particles (density summation, forces, density diffusion) are the *_io entry
 * points further down.  A context with the flag set calls THESE (the plain sphx_sa_segment_bc / _vertex_bc / _density_sum /
 * sphx_forces_basicstep_sa do not know the open boundaries' terms). ---- */
/* saIdentifyCornerVertices (src/cuda/boundary_conditions.cu:667-700): a vertex of an open boundary that also belongs to a
 * segment not of that open boundary gets FG_CORNER in info (in place) */
int sphx_sa_identify_corner_vertices(sphx_ctx *ctx, const void *pos, void *info, const uint32_t *hash, const void *vertices,
	const uint32_t *cellStart, const uint16_t *neibsList, uint32_t numParticles, uint32_t particleRangeEnd, void *stream);
/* initIOmass_vertexCount (src/cuda/boundary_conditions.cu:578-606): per non-corner open-boundary vertex, the number of
 * non-corner vertices it shares an open-boundary segment with (one per shared segment), into forces.w (BUFFER_FORCES as
 * scratch, PredictorCorrectorIntegrator.cc:176-185).  pos: any float4 array of the particles (rows are prefetched, not used) */
int sphx_sa_init_io_mass_vertex_count(sphx_ctx *ctx, const void *vertices, const uint32_t *hash, const void *info,
	const uint32_t *cellStart, const uint16_t *neibsList, void *forces, const void *pos,
	uint32_t numParticles, uint32_t particleRangeEnd, void *stream);
/* initIOmass (src/cuda/boundary_conditions.cu:610-645): newPos = oldPos with the masses of the non-corner open-boundary
 * vertices moved towards half a fluid particle's mass: odd ids take the difference from the even ids they share segments with */
int sphx_sa_init_io_mass(sphx_ctx *ctx, const void *oldPos, const void *forces, const void *vertices, const uint32_t *hash,
	const void *info, const uint32_t *cellStart, const uint16_t *neibsList, void *newPos,
	uint32_t numParticles, uint32_t particleRangeEnd, float deltap, void *stream);
/* findOutgoingSegment (src/cuda/boundary_conditions.cu:238-278): a fluid particle behind an open-boundary segment and moving out
 * relative to it is marked with that segment's vertexinfo (vertices, in place) and its mass repartition over the segment's
 * vertices + its mass (gGam, in place: {beta0, beta1, beta2, mass}) */
int sphx_sa_find_outgoing_segment(sphx_ctx *ctx, const void *pos, const void *vel, void *vertices, void *gGam,
	const void *vertPos0, const void *vertPos1, const void *vertPos2, const void *boundElements, const void *info,
	const uint32_t *hash, const uint32_t *cellStart, const uint16_t *neibsList,
	uint32_t numParticles, uint32_t particleRangeEnd, float influenceradius, void *stream);
/* disableOutgoingParts (src/cuda/boundary_conditions.cu:76-104): marked fluid particles are disabled (NaN mass), marks cleared */
int sphx_sa_disable_outgoing_parts(sphx_ctx *ctx, void *pos, void *vertices, const void *info, uint32_t numParticles, void *stream);
/* saSegmentBoundaryConditions / saVertexBoundaryConditions with open boundaries (src/cuda/boundary_conditions.cu:108-235,280-410
 * with has_io; laminar, SPHX_SIMULATE).  The solid-wall rows go to the kernels of sphx_sa_segment_bc / sphx_sa_vertex_bc, the
 * segments and the non-corner vertices of the open faces get a wave each (gpusph_amd/csrc/sa_io.hip).  Held against the oracle
 * on an MI355X by tests/test_gpu_sa_io.py (round 5).  The driver that calls them: gpusph_amd/multigpu.py.
 * segment pass: vel, gGam, eulerVel in place (boundary rows): an open-boundary segment gets the Riemann-invariant condition from
 * the Shepard means of the fluid next to it and what IMPOSE_OPEN_BOUNDARY_CONDITION left in eulerVel, a solid one the wall density
 * and a cleared Eulerian velocity.
 * vertex pass: vel.w of every vertex; for the non-corner vertices of open boundaries eulerVel, the mass in newPos (mass flux of the
 * adjacent open segments over dt, the mass of marked outgoing particles in step 2) and, in step 2, new fluid particles appended at
 * *newNumParticles (device counter, starts at numParticles; rows up to totParticles must exist in newPos, vel, gGam, eulerVel,
 * forces, vertices, boundElements, info, hash, nextIDs), ids from nextIDs[vertex], which advances by numOpenVertices */
int sphx_sa_segment_bc_io(sphx_ctx *ctx, void *vel, void *gGam, void *eulerVel, const void *pos, const void *vertices,
	const void *boundElements, const void *info, const uint32_t *hash, const uint32_t *cellStart, const uint16_t *neibsList,
	uint32_t numParticles, uint32_t particleRangeEnd, int step, void *stream);
int sphx_sa_vertex_bc_io(sphx_ctx *ctx, void *vel, const void *pos, void *newPos, void *gGam, void *eulerVel, void *forces,
	void *vertices, void *boundElements, const void *vertPos0, const void *vertPos1, const void *vertPos2, void *info, uint32_t *hash,
	uint32_t *nextIDs, uint32_t *newNumParticles, const uint32_t *cellStart, const uint16_t *neibsList,
	uint32_t numParticles, uint32_t particleRangeEnd, uint32_t totParticles, float deltap, float dt, int step,
	uint32_t numOpenVertices, void *stream);
/* density_sum and the forces of SA_BOUNDARY with open boundaries (src/cuda/density_sum.cu + density_sum_kernel.cu:119-140,206-250,
 * 374-420,606-655; src/cuda/forces.cu:751-795 + forces_kernel.def:1485-1497,2494-2507,2703-2708; laminar or inviscid, Wendland).
 * Since round 6 the particle <- particle sums go through the tiled window as for solid walls and the open faces' terms ride with the
 * one-element-per-lane kernels of the boundary elements; the one-thread walkers of sphx_sa_density_sum /
 * sphx_forces_basicstep_sa with their open-boundary terms stay as the stand-by (no tiling, no list of wall particles).
 * sphx_sa_density_sum_io: sphx_sa_density_sum with the open boundaries' terms; oldEulerVel = BUFFER_EULERVEL of step n; dt = the
 *   integration interval of the Euler step just taken.  forces.w receives the volumic sums, as in sphx_sa_density_sum.  newVel and
 *   oldVel are two buffers in the reference's call (write list / read list); if they are one, the stand-by walker is the whole pass
 *   (the fast path hands the flux of gamma through the open segments over in newVel.w before the new density is written there).
 * sphx_forces_basicstep_sa_io: sphx_forces_basicstep_sa (SPHX_SIMULATE) with BUFFER_EULERVEL in the viscous terms and the gamma CFL;
 *   cfl / cflGamma as there (cflGamma: one value per particle, then one per block from round_up(numParticles, 4) on) */
int sphx_sa_density_sum_io(sphx_ctx *ctx, void *newVel, void *newGGam, void *forces, const void *oldPos, const void *newPos,
	const void *oldVel, const void *oldEulerVel, const void *oldGGam, const void *boundElements,
	const void *vertPos0, const void *vertPos1, const void *vertPos2, const void *info,
	const uint32_t *hash, const uint32_t *cellStart, const uint16_t *neibsList,
	uint32_t numParticles, uint32_t particleRangeEnd, float dt, void *stream);
/* ... and with ENABLE_MOVING_BODIES on top (CompleteSaExample.cu's option set, :46): the boundary elements of the state that is read
 * and of the one that is written, as in sphx_sa_density_sum_moving; io_gamma_contrib in the moving boundary loop
 * (src/cuda/density_sum_kernel.cu:422-484). */
int sphx_sa_density_sum_io_moving(sphx_ctx *ctx, void *newVel, void *newGGam, void *forces, const void *oldPos, const void *newPos,
	const void *oldVel, const void *oldEulerVel, const void *oldGGam, const void *oldBoundElements, const void *newBoundElements,
	const void *vertPos0, const void *vertPos1, const void *vertPos2, const void *info,
	const uint32_t *hash, const uint32_t *cellStart, const uint16_t *neibsList,
	uint32_t numParticles, uint32_t particleRangeEnd, float dt, void *stream);
int sphx_forces_basicstep_sa_io(sphx_ctx *ctx, void *forces, float *cfl, float *cflGamma, const void *pos, const void *vel,
	const void *eulerVel, const void *info, const uint32_t *hash, const uint32_t *cellStart, const uint16_t *neibsList,
	const void *gGam, const void *boundElements, const void *vertPos0, const void *vertPos1, const void *vertPos2,
	uint32_t numParticles, uint32_t fromParticle, uint32_t toParticle, float deltap, uint32_t cflOffset, uint32_t *h_numBlocks, void *stream);
/* The Brezzi diffusion with open boundaries and the water depth at the pressure-driven ones; the checkers are
 * orc_sa_density_diffusion_io / orc_sa_io_water_depth (tests/test_gpu_sa_io.py).
 * sphx_sa_compute_density_diffusion_io: computeDensityDiffusionDevice with ENABLE_INLET_OUTLET (src/cuda/forces.cu
 *   compute_density_diffusion + forces_kernel.def:4536-4582): sphx_sa_compute_density_diffusion plus the boundary term of the
 *   segments of pressure-driven open boundaries (:1836-1852), which reads the elements and their vertices' positions.
 * sphx_sa_io_water_depth: what forcesDevice<PT_VERTEX, PT_FLUID> leaves in IOwaterdepth with ENABLE_WATER_DEPTH (vertex_forces,
 *   src/cuda/forces.cu:676-686; forces_kernel.def:192-205, 1375-1389, 3285-3303): per open boundary (the object number of its
 *   vertices) the largest height of a fluid particle below one of its pressure-driven vertices, scaled to [0, UINT_MAX] over the
 *   domain's height.  IOwaterdepth (device, one uint per open boundary) is an atomicMax target: the caller clears it, as the
 *   problem's imposeBoundaryConditionHost does (src/problems/CompleteSaExample.cu:323-325). */
int sphx_sa_compute_density_diffusion_io(sphx_ctx *ctx, void *forces, const void *pos, const void *vel, const void *gGam,
	const void *boundElements, const void *vertPos0, const void *vertPos1, const void *vertPos2, const void *info,
	const uint32_t *hash, const uint32_t *cellStart, const uint16_t *neibsList,
	uint32_t numParticles, uint32_t particleRangeEnd, float deltap, float dt, void *stream);
int sphx_sa_io_water_depth(sphx_ctx *ctx, uint32_t *IOwaterdepth, const void *pos, const void *info, const uint32_t *hash,
	const uint32_t *cellStart, const uint16_t *neibsList, uint32_t numParticles, uint32_t fromParticle, uint32_t toParticle, void *stream);
/* FLUX_COMPUTATION of the post-processing engine (src/cuda/post_process.cu:485-570; fluxComputationDevice
 * src/cuda/post_process_kernel.cu:822-840): IOflux[numOpenBoundaries] (device) = per open boundary the volume flux
 * sum A_s (u_E . n_s) over its segments, from zero (the reference adds onto uncleared memory). */
int sphx_flux_computation(sphx_ctx *ctx, float *IOflux, const void *info, const void *eulerVel, const void *boundElements,
	uint32_t numParticles, uint32_t particleRangeEnd, uint32_t numOpenBoundaries, void *stream);
/* saInitGamma (src/cuda/boundary_conditions.cu:457-560): gamma and grad gamma of fluid and vertex particles at initialisation,
 * grad gamma from the analytical formula of a triangular element, gamma by Gauss quadrature / solid angles
 * (src/cuda/gamma.cuh).  Rows of boundary elements are not written.  oldGGam is accepted for interface parity (unused). */
int sphx_sa_init_gamma(sphx_ctx *ctx, void *newGGam, const void *oldGGam, const void *pos, const void *boundElements,
	const void *vertPos0, const void *vertPos1, const void *vertPos2, const void *info,
	const uint32_t *hash, const uint32_t *cellStart, const uint16_t *neibsList,
	float slength, float influenceradius, float deltap, float epsilon,
	uint32_t numParticles, uint32_t particleRangeEnd, void *stream);
/* saSegmentBoundaryConditions (src/cuda/boundary_conditions.cu:108-235): density (vel.w) and, for moving bodies, velocity of
 * the boundary elements from the fluid / their vertices; gamma of a segment = mean of its vertices when it is (re)computed
 * (step 0 or -1, non-finite gamma, moving bodies).  vel and gGam are updated in place (boundary rows only).
 * step: 0 (or -1) initialisation, 1 / 2 integrator steps; run_mode: SPHX_SIMULATE or SPHX_REPACK */
int sphx_sa_segment_bc(sphx_ctx *ctx, void *vel, void *gGam, const void *pos, const void *vertices,
	const void *boundElements, const void *info, const uint32_t *hash, const uint32_t *cellStart, const uint16_t *neibsList,
	uint32_t numParticles, uint32_t particleRangeEnd, float deltap, float slength, float influenceradius,
	int step, int run_mode, void *stream);
/* saVertexBoundaryConditions (src/cuda/boundary_conditions.cu:280-410) without open boundaries: density of the vertex
 * particles from the fluid (vel.w of vertex rows, in place) */
int sphx_sa_vertex_bc(sphx_ctx *ctx, void *vel, const void *gGam, const void *pos, const void *info,
	const uint32_t *hash, const uint32_t *cellStart, const uint16_t *neibsList,
	uint32_t numParticles, uint32_t particleRangeEnd, float deltap, float slength, float influenceradius,
	int step, int run_mode, void *stream);

/* ---- turbulence<KEPSILON> (k-epsilon model; SA_BOUNDARY with solid walls only, src/cuda/cudasimframework.cu:155) ------------
 * Buffers of the model: BUFFER_TKE, BUFFER_EPSILON, BUFFER_TURBVISC (float), BUFFER_EULERVEL (float4) -- particle properties,
 * re-sorted with the particles -- and the ephemeral BUFFER_DKDE (3 floats per particle: diffusion term of k, of epsilon,
 * Yap's C_e2) and BUFFER_CFL_KEPS (one float per forces block).  With KEPSILON uploaded the plain SA entry points answer
 * SPHX_ERR_INVALID for a SIMULATE pass; the repacking run mode has no k-epsilon and keeps using them.
 *   sphx_sa_segment_bc_keps   saSegmentBoundaryConditions with has_keps (boundary_conditions_kernel.cu:509-541,678-693,748-758,
 *                             806-826,1262-1277): also k = Shepard mean of the fluid's k (dk/dn = 0), epsilon from the wall law
 *                             de/dn = 4 c_mu^(3/4) k^(3/2)/(kappa r), Eulerian velocity = tangential mean of the three vertices
 *   sphx_sa_vertex_bc_keps    saVertexBoundaryConditions with has_keps (:918-927,1002-1021,1052-1072): k, epsilon = means over the
 *                             adjacent segments (floors 1e-6), Eulerian velocity made tangential to the vertex normal
 *   sphx_forces_basicstep_sa_keps  basicstep of the forces engine with keps_forces_params (forces_kernel.def:262-270,389-401,
 *                             2824-2878,2915-2980,3123-3170): P + 2/3 rho k in the pressure term, laminar + eddy viscosity in the
 *                             volumic MORRIS term, wall shear stress from the law of the wall instead of the laminar wall term,
 *                             DKDE from the boundary sums (the state the reference's last forcesDevice launch leaves), the strain
 *                             production term limited to 0.3 k S; cflKeps = per-block max eddy viscosity
 *   sphx_euler_keps           the k-epsilon part of eulerDevice (euler_kernel.def:219-231,262-274,325-337): semi-implicit k and
 *                             epsilon of fluid rows, eulerVel += dt force of wall rows, turbvisc = 0.9 k^2/epsilon of every row;
 *                             dt from d_dt[0]*dt_scale when d_dt is given, like sphx_euler_basicstep
 *   sphx_forces_dtreduce_keps_device  viscous part of dtreduce (forces.cu:585-598): d_dt = min(d_dt, 0.125 h^2/(max_kinematic + max cflKeps)) */
int sphx_sa_segment_bc_keps(sphx_ctx *ctx, void *vel, void *gGam, float *tke, float *eps, void *eulerVel,
	const void *pos, const void *vertices,
	const void *boundElements, const void *info, const uint32_t *hash, const uint32_t *cellStart, const uint16_t *neibsList,
	uint32_t numParticles, uint32_t particleRangeEnd, float deltap, float slength, float influenceradius,
	int step, int run_mode, void *stream);
int sphx_sa_vertex_bc_keps(sphx_ctx *ctx, void *vel, const void *gGam, float *tke, float *eps, void *eulerVel,
	const void *vertices, const void *boundElements, const void *pos, const void *info,
	const uint32_t *hash, const uint32_t *cellStart, const uint16_t *neibsList,
	uint32_t numParticles, uint32_t particleRangeEnd, float deltap, float slength, float influenceradius,
	int step, int run_mode, void *stream);
int sphx_forces_basicstep_sa_keps(sphx_ctx *ctx, void *forces, float *cfl, float *cflGamma, float *cflKeps, float *dkde,
	const void *pos, const void *vel, const void *info, const uint32_t *hash, const uint32_t *cellStart, const uint16_t *neibsList,
	const void *gGam, const void *boundElements, const void *vertPos0, const void *vertPos1, const void *vertPos2,
	const float *tke, const float *eps, const float *turbvisc, const void *eulerVel,
	uint32_t numParticles, uint32_t fromParticle, uint32_t toParticle,
	float deltap, float slength, float dtadaptfactor, float influenceradius, float epsilon, uint32_t cflOffset,
	int run_mode, int step, float dt, uint32_t *h_numBlocks, void *stream);
int sphx_euler_keps(sphx_ctx *ctx, float *newTke, float *newEps, float *newTurbVisc, void *newEulerVel,
	const float *oldTke, const float *oldEps, const void *oldEulerVel, const float *dkde, const void *forces,
	const void *oldPos, const void *info, uint32_t numParticles, uint32_t particleRangeEnd,
	float dt, const float *d_dt, float dt_scale, void *stream);
int sphx_forces_dtreduce_keps_device(sphx_ctx *ctx, const float *cflKeps, uint32_t numBlocks, float slength,
	float max_kinematic, float *d_dt, void *stream);
/* the same on a host value (synchronous), for the reference's blocking dtreduce */
int sphx_forces_dtreduce_keps(sphx_ctx *ctx, const float *cflKeps, uint32_t numBlocks, float slength,
	float max_kinematic, float *h_dt_inout, void *stream);

/* ---- AbstractForcesEngine ----------------------------------------------------------------- */
uint32_t sphx_forces_fmax_elements(uint32_t n);       /* getFmaxElements, src/cuda/forces.cu:539-543 */
uint32_t sphx_forces_fmax_temp_elements(uint32_t n);  /* getFmaxTempElements, :548-552 */
uint32_t sphx_forces_round_particles(uint32_t n);     /* round_particles, :960-964 */
/* basicstep (src/cuda/forces.cu:897-935): pair summation over [fromParticle,toParticle) +
 * finalize (gravity, /rho0, CFL block maxima into cfl[cflOffset..]).  tau0..2 may be NULL
 * (SPS only), rbforces/rbtorques may be NULL; xsph (float4, BUFFER_XSPH: mean velocity correction of fluid
 * particles, src/cuda/forces_params.h:213-221) is written with SPHX_ENABLE_XSPH and may be NULL otherwise.  *h_numBlocks receives the return value of
 * the reference's basicstep (#CFL elements written).
 * run_mode = SPHX_REPACK selects run_repack (src/cuda/forces.cu:828-896): the mixing force of the repacking
 * integrator on fluid particles (needs SPHX_ENABLE_REPACKING in simflags, src/main.cc:357-358). */
int sphx_forces_basicstep(sphx_ctx *ctx,
	void *forces, float *cfl, void *rbforces, void *rbtorques,
	const void *pos, const void *vel, const void *info, const uint32_t *hash,
	const uint32_t *cellStart, const uint16_t *neibsList,
	const void *tau0, const void *tau1, const void *tau2, void *xsph,
	uint32_t numParticles, uint32_t fromParticle, uint32_t toParticle,
	float deltap, float slength, float dtadaptfactor, float influenceradius,
	uint32_t cflOffset, int run_mode, int step, float dt, int compute_object_forces,
	uint32_t *h_numBlocks, void *stream);
/* dtreduce (src/cuda/forces.cu:556-606): sync, returns dt through *h_dt */
int sphx_forces_dtreduce(sphx_ctx *ctx, float slength, float dtadaptfactor, float sspeed_cfl,
	float max_kinematic, const float *cfl, float *cflTemp, uint32_t numBlocks,
	float *h_dt, void *stream);
/* device-resident variant (no host sync): d_dt[0] = (combine_min ? min(d_dt[0], dt) : dt) */
int sphx_forces_dtreduce_device(sphx_ctx *ctx, float slength, float dtadaptfactor, float sspeed_cfl,
	float max_kinematic, const float *cfl, float *cflTemp, uint32_t numBlocks,
	float *d_dt, int combine_min, void *stream);
/* reduceRbForces (src/cuda/forces.cu:966-1004): sync; h_total* receive 3 floats per body */
int sphx_reduce_rb_forces(sphx_ctx *ctx, void *rbforces, void *rbtorques, const uint32_t *rbnum,
	const uint32_t *h_lastindex, float *h_totalforce3, float *h_totaltorque3,
	uint32_t numforcesbodies, uint32_t numForcesBodiesParticles, void *stream);

/* Multi-GPU runs: leave `cus` compute units (rounded up to a multiple of 8, one per XCD) out of the persistent grid of
 * the tiled forces kernel, so that the communication kernels of the halo exchange (RCCL send/recv on a second stream) find
 * free CUs and LDS while the inner stripe computes.  The reference moves its halos with copy engines
 * (cudaMemcpyPeerAsync, src/GPUWorker.cc:396-407), which cost no SMs; this is the equivalent knob for kernel-based
 * transports.  0 (the default) = every CU runs a tile workgroup. */
int sphx_forces_reserve_cus(sphx_ctx *ctx, uint32_t cus);

/* Optional profiling hook (not a reference interface; used by bench.py for the roofline figure): when enabled,
 * every sphx_forces_basicstep records HIP events on its launch stream around its dominant kernel.  _read waits
 * for the recorded events, returns the summed elapsed time [ms] and the number of launches, and releases them. */
int sphx_forces_timing(sphx_ctx *ctx, int enable);
int sphx_forces_timing_read(sphx_ctx *ctx, double *total_ms, uint32_t *launches);

/* ---- AbstractViscEngine ------------------------------------------------------------------- */
/* calc_visc, SPS branch (src/cuda/visc.cu:175-234): tau0..2 are float2 arrays, turbvisc may be NULL */
int sphx_calc_visc(sphx_ctx *ctx, void *tau0, void *tau1, void *tau2, float *spsturbvisc,
	const void *pos, const void *vel, const void *info, const uint32_t *hash,
	const uint32_t *cellStart, const uint16_t *neibsList,
	uint32_t numParticles, uint32_t particleRangeEnd,
	float deltap, float slength, float influenceradius, void *stream);

/* ---- AbstractFilterEngine (density filters, run every N-th iteration) --------------------- */
/* process (src/engine_filter.h:69-77, CUDAFilterEngine src/cuda/forces.cu:1008-1147; kernels shepardDevice /
 * MlsDevice src/cuda/forces_kernel.cu:418-721): newVel = oldVel with rho~ replaced by the Shepard- or
 * MLS-corrected density (Shepard: fluid particles only, others are copied; inactive particles are not written).
 * newVel must not alias oldVel. */
int sphx_filter_process(sphx_ctx *ctx, int filtertype, void *newVel,
	const void *pos, const void *oldVel, const void *info, const uint32_t *hash,
	const uint32_t *cellStart, const uint16_t *neibsList,
	uint32_t numParticles, uint32_t particleRangeEnd, float slength, float influenceradius, void *stream);

/* ---- AbstractPostProcessEngine (run before writes) ---------------------------------------- */
/* process (src/engine_postprocess.h:80-87, CUDAPostProcessEngine src/cuda/post_process.cu:88-300; kernels
 * calcVortDevice / calcTestpointsVelocityDevice / calcSurfaceparticleDevice src/cuda/post_process_kernel.cu:58-392):
 *   SPHX_VORTICITY          writes vorticity (3 floats per particle, NaN for non-fluid / inactive); reads vel, info
 *   SPHX_TESTPOINTS         overwrites the velocity rows of test points in velInOut with the Shepard-normalised
 *                           velocity (xyz) and pressure (w) of their fluid neighbours; reads info
 *   SPHX_SURFACE_DETECTION  sets/clears FG_SURFACE of fluid particles in infoInOut; reads vel; normals (float4,
 *                           BUFFER_NORMALS option) may be NULL; cosconeangle*: PhysParams, src/physparams.h:370-371
 *   SPHX_INTERFACE_DETECTION  multi-fluid runs (calcInterfaceparticleDevice, post_process_kernel.cu:388-560): FG_SURFACE as
 *                           above plus FG_INTERFACE for fluid particles whose same-fluid cone is empty; same buffers
 * Buffers a type does not use may be NULL. */
int sphx_postprocess(sphx_ctx *ctx, int type,
	void *vorticity, void *velInOut, void *infoInOut, void *normals,
	const void *pos, const void *vel, const void *info, const uint32_t *hash,
	const uint32_t *cellStart, const uint16_t *neibsList,
	uint32_t numParticles, uint32_t particleRangeEnd, float cosconeanglefluid, float cosconeanglenonfluid, void *stream);

/* ---- AbstractIntegrationEngine ------------------------------------------------------------ */
/* basicstep (src/cuda/euler.cu:329-366).  dt is the step's dt or dt/2 exactly as the
 * reference passes it; if d_dt is not NULL the kernel instead uses d_dt[0]*dt_scale read on
 * the device (no host round trip of the adaptive dt).  run_mode = SPHX_REPACK: euler_repack_params
 * (src/cuda/euler.cu:346-353), only fluid positions and velocities are advanced. */
int sphx_euler_basicstep(sphx_ctx *ctx, void *newPos, void *newVel,
	const void *oldPos, const void *oldVel, const void *info, const uint32_t *hash,
	const void *forces, const void *xsph,
	uint32_t numParticles, uint32_t particleRangeEnd,
	float dt, const float *d_dt, float dt_scale, int step, float t,
	float slength, float influenceradius, int run_mode, void *stream);

/* ---- ENABLE_INTERNAL_ENERGY (AccuracyTest) --------------------------------------------------------------------------------------
 *   sphx_forces_internal_energy  the BUFFER_INTERNAL_ENERGY_UPD output of a forces pass (internal_energy_forces_params,
 *                                src/cuda/forces_params.h:296-303; add_internal_energy src/cuda/forces_kernel.def:3308-3320): call it with
 *                                the arguments of the sphx_forces_basicstep it belongs to; DEDt[from..to) is written (the reference
 *                                clobbers the buffer before the pass and accumulates over its three launches)
 *   sphx_euler_internal_energy   the BUFFER_INTERNAL_ENERGY part of the integration step (energy_euler_params,
 *                                src/cuda/euler_params.h:121-134): newEnergy = oldEnergy + dt DEDt for the particles eulerDevice
 *                                integrates; dt / d_dt / dt_scale as in sphx_euler_basicstep */
int sphx_forces_internal_energy(sphx_ctx *ctx, float *DEDt,
	const void *pos, const void *vel, const void *info, const uint32_t *hash,
	const uint32_t *cellStart, const uint16_t *neibsList,
	uint32_t numParticles, uint32_t fromParticle, uint32_t toParticle, void *stream);
int sphx_euler_internal_energy(sphx_ctx *ctx, float *newEnergy, const float *oldEnergy, const float *DEDt,
	const void *oldPos, const void *info, uint32_t numParticles, uint32_t particleRangeEnd,
	float dt, const float *d_dt, float dt_scale, void *stream);

/* ---- generalized Newtonian rheologies (rheology<BINGHAM | PAPANASTASIOU | POWER_LAW | HERSCHEL_BULKLEY | ALEXANDROU | DEKEE_TURCOTTE |
 * ZHU>, e.g. PoiseuillePapanastasiou) ------------------------------------------------------------------------------------------
 *   sphx_calc_effvisc              AbstractViscEngine::calc_visc when NEEDS_EFFECTIVE_VISC (src/cuda/visc.cu:86-170, effectiveViscDevice
 *                                  src/cuda/visc_kernel.cu:655-713; the CALC_VISC command before every forces pass,
 *                                  src/integrators/PredictorCorrectorIntegrator.cc:460-480): writes BUFFER_EFFVISC (mu_eff for compvisc
 *                                  DYNAMIC, mu_eff/rho for KINEMATIC); *h_max_kinvisc (may be NULL: no host synchronisation then) gets
 *                                  the largest kinematic viscosity, which the caller hands to sphx_forces_dtreduce as
 *                                  max_kinematic (src/GPUWorker.cc:2633-2645,2013-2030); NaN without ENABLE_DTADAPT like the reference
 *   sphx_forces_basicstep_effvisc  basicstep of the forces engine with effective_visc_forces_params: the arguments of
 *                                  sphx_forces_basicstep that the built option set uses, plus effvisc; sphx_forces_basicstep itself
 *                                  answers SPHX_ERR_INVALID for a SIMULATE pass of these rheologies */
int sphx_calc_effvisc(sphx_ctx *ctx, float *effvisc, float *h_max_kinvisc,
	const void *pos, const void *vel, const void *info, const uint32_t *hash,
	const uint32_t *cellStart, const uint16_t *neibsList,
	uint32_t numParticles, uint32_t particleRangeEnd, float deltap, float slength, float influenceradius, void *stream);
int sphx_forces_basicstep_effvisc(sphx_ctx *ctx, void *forces, float *cfl,
	const void *pos, const void *vel, const void *info, const uint32_t *hash,
	const uint32_t *cellStart, const uint16_t *neibsList, const float *effvisc,
	uint32_t numParticles, uint32_t fromParticle, uint32_t toParticle,
	float deltap, float slength, float dtadaptfactor, float influenceradius,
	uint32_t cflOffset, int run_mode, int step, float dt, uint32_t *h_numBlocks, void *stream);

/* ---- SPH_GRENIER (formulation<SPH_GRENIER>, boundary<DYN_BOUNDARY>: Bubble, LockExchange, RTInstability, OilJet) ----------
 * The volume formulation keeps two more buffers, BUFFER_VOLUME (float4: x initial volume, y log(current/initial), w current
 * volume) and BUFFER_SIGMA (float).
 *   sphx_init_volume               ProblemCore::init_volume (src/ProblemCore.cc:1586-1606), on the uploaded arrays
 *   sphx_compute_density           AbstractForcesEngine::compute_density (src/engine_forces.h:118-125, CUDADensityHelper<.., SPH_GRENIER, ..>
 *                                  src/cuda/forces.cu:208-246; the COMPUTE_DENSITY command before every forces pass,
 *                                  src/integrators/PredictorCorrectorIntegrator.cc:443-458): writes sigma, rewrites vel.w in
 *                                  place; the 'typical sigma' of boundary particles out of reach of the fluid uses
 *                                  maxFluidBoundaryNeibs of the last sphx_build_neibs on this context.  Does nothing for the
 *                                  other formulations, like the reference's helper.
 *   sphx_forces_basicstep_grenier  basicstep of the forces engine with grenier_forces_params (src/cuda/forces_params.h:224-240):
 *                                  the arguments of sphx_forces_basicstep that the built option set uses, plus sigma
 *   sphx_euler_basicstep_grenier   basicstep of the integration engine with Vol_params (src/cuda/euler_params.h:153-156):
 *                                  sphx_euler_basicstep plus BUFFER_VOLUME old/new; sphx_euler_basicstep itself answers
 *                                  SPHX_ERR_INVALID for a SIMULATE step of this formulation */
int sphx_init_volume(sphx_ctx *ctx, void *vol, const void *pos, const void *vel, const void *info,
	uint32_t numParticles, void *stream);
int sphx_compute_density(sphx_ctx *ctx, float *sigma, void *vel, const void *pos, const void *info,
	const uint32_t *hash, const void *vol, const uint32_t *cellStart, const uint16_t *neibsList,
	uint32_t numParticles, float slength, float influenceradius, void *stream);
int sphx_forces_basicstep_grenier(sphx_ctx *ctx, void *forces, float *cfl,
	const void *pos, const void *vel, const void *info, const uint32_t *hash,
	const uint32_t *cellStart, const uint16_t *neibsList, const float *sigma,
	uint32_t numParticles, uint32_t fromParticle, uint32_t toParticle,
	float deltap, float slength, float dtadaptfactor, float influenceradius,
	uint32_t cflOffset, int run_mode, int step, float dt, uint32_t *h_numBlocks, void *stream);
int sphx_euler_basicstep_grenier(sphx_ctx *ctx, void *newPos, void *newVel, void *newVol,
	const void *oldPos, const void *oldVel, const void *oldVol, const void *info, const uint32_t *hash,
	const void *forces, const void *xsph,
	uint32_t numParticles, uint32_t particleRangeEnd,
	float dt, const float *d_dt, float dt_scale, int step, float t,
	float slength, float influenceradius, int run_mode, void *stream);

/* disableFreeSurfParts (src/engine_integration.h:137, src/cuda/euler.cu:368-391): at the end of a repacking run,
 * disable (mass = NaN) the non-fluid particles flagged FG_SURFACE */
int sphx_disable_free_surf_parts(sphx_ctx *ctx, void *pos, const void *info,
	uint32_t numParticles, uint32_t particleRangeEnd, void *stream);

/* The EOS rows of the forces engine ({P/rho^2, c, P, rho} per particle, a scratch array of the context) are made by a pre-pass of
 * sphx_forces_basicstep over the velocity buffer it is given.  sphx_eos_rows_follow_euler(on != 0) makes sphx_euler_basicstep
 * (SPHX_SIMULATE, DYN / LJ boundaries) write them next to the densities they are functions of; a caller that then hands that very
 * buffer, unchanged, to sphx_forces_basicstep states so with sphx_eos_rows_current(vel, numParticles) right before the call, and
 * the pre-pass is skipped for that call.  "Unchanged" is the caller's statement (no filter, boundary-condition pass, sort, halo
 * import or upload since the Euler step, or since the previous forces call on the same buffer); a buffer / row count the rows
 * were not made for is ignored.  Off by default: the engine adapters of host/hip_engines.h, which cannot see what the
 * Integrator does between two commands, never make the statement; the native drivers (gpusph_amd/multigpu.py, slab_run.cpp) do.
 * No reference counterpart (the reference recomputes the EOS of the neighbour in every pair). */
int sphx_eos_rows_follow_euler(sphx_ctx *ctx, int on);
int sphx_eos_rows_current(sphx_ctx *ctx, const void *vel, uint32_t numParticles);

/* t += dt on the device (double += float, the sum GPUSPH::runSimulation keeps on the host, src/GPUSPH.cc:650-657): for
 * callers that keep dt device-resident (sphx_forces_dtreduce_device) and do not want a host round trip per step */
int sphx_time_advance(sphx_ctx *ctx, double *d_t, const float *d_dt, void *stream);

/* ---- small stream-ordered helpers the callers of the reference get from cudaMemset -------- */
int sphx_memset_async(void *ptr, int value, size_t bytes, void *stream);

/* ---- device memory service ------------------------------------------------------------------
 * What the reference's host code gets from the CUDA runtime: cudaMalloc / cudaFree / cudaMemset of
 * CUDABuffer (src/cuda/cudabuffer.h:47-131), the device selection of the worker threads
 * (checkCUDA, src/cuda/cudautil.cc:41; GPUWorker.cc:3245) and the blocking copies of GPUWorker's
 * uploads / dumps (src/GPUWorker.cc:1186-1330).  With these the host adapters (gpusph_amd/host/)
 * compile with a plain C++ compiler against GPUSPH's own headers: no HIP header on the host side of
 * the boundary.  All of them act on the calling thread's current device; copies and sphx_memset are
 * blocking like their cuda* counterparts. */
int sphx_device_count(int *count);
int sphx_set_device(int device);
int sphx_get_device(int *device);
int sphx_device_synchronize(void);
int sphx_malloc(void **ptr, size_t bytes);
int sphx_free(void *ptr);
int sphx_memset(void *ptr, int value, size_t bytes);
int sphx_memcpy_h2d(void *dst, const void *h_src, size_t bytes);
int sphx_memcpy_d2h(void *h_dst, const void *src, size_t bytes);
int sphx_memcpy_d2d(void *dst, const void *src, size_t bytes);

/* ---- halo exchange of the slab decomposition -----------------------------------------------------------------
 * What GPUWorker::importExternalCells / transferBursts / peerAsyncTransfer / networkTransfer do
 * (src/GPUWorker.cc:396-407, 711-822, 825-948): after a re-sort every device sends the particles of its inner-edge layers
 * (pos, vel, info, hash, and whatever particle state the option set adds) to the neighbouring devices and receives theirs
 * behind its own particles; after each forces pass the same for the forces.  REORDER leaves each layer one contiguous range
 * of rows (device map split on COORD3), so a layer of a buffer is [start, start + count) rows of rowBytes bytes.
 *
 * Two transports behind the same calls:
 *   sphx_halo_create_threads   one worker THREAD per device in one process, the reference's own model
 *                              (GPUWorker::simulationThread): a layer is one peer copy on the caller's stream
 *                              (hipMemcpyPeerAsync, over xGMI between two devices), the workers meet at a barrier around
 *                              it like the reference's threadSynchronizer.  Every worker calls each collective below.
 *   sphx_halo_create_rccl      one PROCESS per device: ncclSend / ncclRecv of the layers in one group per exchange on the
 *                              caller's stream (no host synchronisation), dt by ncclAllReduce(min).  The 128-byte id comes
 *                              from sphx_halo_unique_id on one rank and reaches the others by the host's own means.
 *                              RCCL is dlopen'ed on first use.
 * leftRank / rightRank: the ranks owning the neighbouring slabs, -1 for none.  All calls of a collective must be made by
 * every rank of the group (thread transport: they block on a barrier).
 * Failure: a rank that fails inside a collective of the thread transport (a HIP error, a layer that does not match what its
 * neighbour posted) still passes every barrier of that collective and flags the failure to the group: the call returns an
 * error on that rank AND on the ranks that depend on it, nobody is left blocked.  With RCCL an error is local to the rank. */
typedef struct sphx_halo sphx_halo;
typedef struct sphx_halo_group sphx_halo_group;
int sphx_halo_group_create(int world, sphx_halo_group **out);
int sphx_halo_group_destroy(sphx_halo_group *group);
int sphx_halo_create_threads(sphx_halo_group *group, sphx_ctx *ctx, int rank, sphx_halo **out);
int sphx_halo_unique_id(void *id128);
int sphx_halo_create_rccl(sphx_ctx *ctx, const void *id128, int rank, int world, sphx_halo **out);
int sphx_halo_destroy(sphx_halo *h);
/* UPDATE_EXTERNAL of `nbuf` buffers: send my two edge layers, receive the two halo layers */
int sphx_halo_exchange(sphx_halo *h, int nbuf, void *const *bufs, const uint32_t *rowBytes,
	int leftRank, uint32_t sendLeftStart, uint32_t sendLeftCount, uint32_t recvLeftStart, uint32_t recvLeftCount,
	int rightRank, uint32_t sendRightStart, uint32_t sendRightCount, uint32_t recvRightStart, uint32_t recvRightCount,
	void *stream);
/* dt of the step = the smallest of the devices' (GPUSPH.cc:650-657 over gdata->dts); total force / torque on a body */
int sphx_halo_allreduce_min_f32(sphx_halo *h, float *d_value, void *stream);
int sphx_halo_allreduce_sum_f32(sphx_halo *h, float *d_values, uint32_t n, void *stream);
/* ... in double, the precision the body totals are reduced in on the host (REDUCE_BODIES_FORCES_HOST, src/GPUSPH.cc:1905-1960);
 * worker threads: n <= 8 */
int sphx_halo_allreduce_sum_f64(sphx_halo *h, double *d_values, uint32_t n, void *stream);
/* the sizes of the layers every rank is about to send (host values: the receiver sizes its halo with them) */
int sphx_halo_allgather_u64x2(sphx_halo *h, const uint64_t mine[2], uint64_t *all /* [2*world] */, void *stream);
int sphx_halo_barrier(sphx_halo *h, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SPHX_H */
