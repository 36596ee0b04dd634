// sphx_api.hip -- context, constants and error plumbing of libsphx (C ABI in include/sphx.h).
#include "sphx_internal.h"
#include <cstdio>
#include <cmath>
#include <cstring>
#include <cstdlib>

static thread_local std::string g_last_error;

int sphx_set_error(int code, const std::string &msg)
{
	g_last_error = msg;
	return code;
}

extern "C" const char *sphx_last_error(void) { return g_last_error.c_str(); }
extern "C" const char *sphx_version(void) { return "sphx 0.1 (gfx950)"; }

extern "C" int sphx_create(sphx_ctx **out, int device)
{
	SPHX_REQUIRE(out != nullptr, "sphx_create: out is NULL");
	int ndev = 0;
	SPHX_HIP(hipGetDeviceCount(&ndev));
	if (device < 0 || device >= ndev)
		return sphx_set_error(SPHX_ERR_INVALID, "sphx_create: no such HIP device");
	SPHX_HIP(hipSetDevice(device));
	sphx_ctx *ctx = new sphx_ctx();
	memset((void*)ctx, 0, sizeof(*ctx));
	ctx->device = device;
	SPHX_HIP(hipMalloc((void**)&ctx->rb_dev, sizeof(RbParams)));
	SPHX_HIP(hipMemset(ctx->rb_dev, 0, sizeof(RbParams)));
	SPHX_HIP(hipHostMalloc((void**)&ctx->rb_staging, sizeof(RbParams)*SPHX_RB_RING, hipHostMallocDefault));
	for (int k = 0; k < SPHX_RB_RING; ++k) SPHX_HIP(hipEventCreateWithFlags(&ctx->rb_staged[k], hipEventDisableTiming));
	// the counters, then their NEIBS_SPREAD partial sets (see NeibsSpread)
	SPHX_HIP(hipMalloc((void**)&ctx->counters_dev, sizeof(NeibsCounters) + NEIBS_SPREAD*sizeof(NeibsSpread)));
	SPHX_HIP(hipMemset(ctx->counters_dev, 0, sizeof(NeibsCounters) + NEIBS_SPREAD*sizeof(NeibsSpread)));
	SPHX_HIP(hipMalloc((void**)&ctx->dt_scratch, 4*sizeof(float)));
	SPHX_HIP(hipMalloc((void**)&ctx->tile_ctl, 16*sizeof(uint32_t)));
	SPHX_HIP(hipMemset(ctx->tile_ctl, 0, 16*sizeof(uint32_t)));
	SPHX_HIP(hipHostMalloc((void**)&ctx->ovf_host, 2*sizeof(uint32_t), hipHostMallocDefault));
	SPHX_HIP(hipEventCreateWithFlags(&ctx->ovf_event, hipEventDisableTiming));
	ctx->tiles_overflow = -1;
	int cus = 0;
	SPHX_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
	ctx->tile_grid = (uint32_t)(cus > 0 ? cus : 256)*TILE_WGS_PER_CU;   // persistent grid: one 512-thread workgroup per CU (LDS bound)
	const char *dis = getenv("SPHX_DISABLE_TILES");
	ctx->disable_tiles = dis && dis[0] == '1';
	{ const char *mf = getenv("SPHX_NEIBS_MFMA"); ctx->neibs_mfma = mf && mf[0] == '1'; }
	// a tiled list build in parts, the tile lists of a part beside the list build of the next one (sphx_build_neibs_sa); 1 = one launch each
	{ const char *lp = getenv("SPHX_LIST_PARTS"); const int n = lp ? atoi(lp) : SPHX_LIST_PARTS_DEFAULT;
	  ctx->list_parts = n < 1 ? 1 : n > SPHX_LIST_PARTS_MAX ? SPHX_LIST_PARTS_MAX : n; }
	// SPHX_DISABLE_TILES=1: always the generic gather kernel (A/B runs, tests).
	// SPHX_TILE_DEBUG (ForcesArgs::dbg: timing experiments, some of which skip work and give wrong results) only exists in a
	// library built with -DSPHX_TILE_DEBUG_BUILD (make EXTRA=-DSPHX_TILE_DEBUG_BUILD); the product library ignores the variable
	ctx->tile_debug = 0;
#ifdef SPHX_TILE_DEBUG_BUILD
	const char *dbg = getenv("SPHX_TILE_DEBUG");
	ctx->tile_debug = dbg ? atoi(dbg) : 0;
#else
	if (getenv("SPHX_TILE_DEBUG"))
		fprintf(stderr, "libsphx: SPHX_TILE_DEBUG is ignored (library built without -DSPHX_TILE_DEBUG_BUILD)\n");
#endif
	*out = ctx;
	return SPHX_OK;
}

static void free_scratch(sphx_ctx *ctx)
{
	void *ptrs[] = { ctx->bin_count, ctx->bin_start, ctx->scan_partials, ctx->slot,
		ctx->tmp_hash, ctx->tmp_index, ctx->tmp_info, ctx->eos_aux, ctx->tau_pack, ctx->tiles, ctx->cell_end_copy, ctx->cell_fluid_end, ctx->tile_cols,
		ctx->tile_list, ctx->tile_runs, ctx->tile_rows, ctx->tile_lane_rec, ctx->tile_lane_index, ctx->neib_counts,
		ctx->sa_wall, ctx->sa_wall_vert, ctx->sa_wall_cache, ctx->sa_wall_tag, ctx->sa_wall_gsum, ctx->sa_wall_open, ctx->sa_rows_bound, ctx->sa_rows_vert };
	for (void *p : ptrs) if (p) (void)hipFree(p);
	ctx->bin_count = ctx->bin_start = ctx->scan_partials = ctx->slot = nullptr;
	ctx->tmp_hash = ctx->tmp_index = nullptr;
	ctx->tmp_info = nullptr;
	ctx->eos_aux = nullptr; ctx->tau_pack = nullptr;
	ctx->eos_tag_vel = nullptr; ctx->eos_tag_n = 0; ctx->eos_armed = false;
	ctx->sa_wall = nullptr; ctx->sa_wall_vert = nullptr; ctx->sa_wall_neibslist = nullptr;
	ctx->sa_rows_bound = nullptr; ctx->sa_rows_vert = nullptr; ctx->sa_rows_range = 0;
	ctx->sa_wall_cache = nullptr; ctx->sa_wall_tag = nullptr; ctx->sa_wall_gsum = nullptr; ctx->sa_wall_open = nullptr; ctx->sa_wall_open_neibslist = nullptr; ctx->sa_wall_capacity = 0;
	ctx->tile_list = nullptr; ctx->tile_runs = nullptr; ctx->tile_rows = nullptr; ctx->tile_lane_rec = nullptr; ctx->tile_lane_index = nullptr;
	ctx->tile_list_batches = ctx->tile_lane_cap = 0; ctx->neib_counts = nullptr;
	ctx->tiles = nullptr; ctx->cell_end_copy = nullptr; ctx->cell_fluid_end = nullptr; ctx->tile_cols = nullptr;
	ctx->tile_capacity = 0; ctx->cells_reserved = 0; ctx->tiles_built = false;
	ctx->reserved_particles = ctx->reserved_bins = 0;
}

extern "C" void sphx_destroy(sphx_ctx *ctx)
{
	if (!ctx) return;
	(void)hipSetDevice(ctx->device);
	free_scratch(ctx);
	if (ctx->rb_dev) (void)hipFree(ctx->rb_dev);
	if (ctx->rb_staging) {
		(void)hipHostFree(ctx->rb_staging);
		for (int k = 0; k < SPHX_RB_RING; ++k) (void)hipEventDestroy(ctx->rb_staged[k]);
	}
	if (ctx->counters_dev) (void)hipFree(ctx->counters_dev);
	if (ctx->dt_scratch) (void)hipFree(ctx->dt_scratch);
	if (ctx->tile_ctl) (void)hipFree(ctx->tile_ctl);
	if (ctx->ovf_host) { (void)hipHostFree(ctx->ovf_host); (void)hipEventDestroy(ctx->ovf_event); }
	if (ctx->side_stream) { (void)hipStreamDestroy(ctx->side_stream); (void)hipEventDestroy(ctx->side_fork); (void)hipEventDestroy(ctx->side_join); }
	if (ctx->list_part_events) for (int k = 0; k < SPHX_LIST_PARTS_MAX; ++k) (void)hipEventDestroy(ctx->list_part[k]);
	if (ctx->dem) (void)hipFree(ctx->dem);
	if (ctx->open_rows) (void)hipFree(ctx->open_rows);
	delete ctx->forces_events;
	delete ctx;
}

// number of sort bins: one per (cell type, cell) + 1 for inactive particles (CELL_HASH_MAX)
static uint32_t num_bins(const sphx_ctx *ctx)
{
	const uint64_t cells = (uint64_t)ctx->params.gridSize[0]*ctx->params.gridSize[1]*ctx->params.gridSize[2];
	return (uint32_t)(4*cells + 1);
}

int sphx_ensure_scratch(sphx_ctx *ctx, uint32_t numParticles)
{
	SPHX_REQUIRE(ctx->have_params, "sphx: set_constants must be called before using the engines");
	const uint32_t bins = num_bins(ctx);
	if (numParticles <= ctx->reserved_particles && bins <= ctx->reserved_bins)
		return SPHX_OK;
	hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
	(void)cs;
	const uint32_t n = numParticles > ctx->reserved_particles ? numParticles : ctx->reserved_particles;
	free_scratch(ctx);
	SPHX_HIP(hipMalloc((void**)&ctx->bin_count, sizeof(uint32_t)*((size_t)bins + 1)));
	SPHX_HIP(hipMalloc((void**)&ctx->bin_start, sizeof(uint32_t)*((size_t)bins + 1)));
	SPHX_HIP(hipMalloc((void**)&ctx->scan_partials, sizeof(uint32_t)*((size_t)(bins > n ? bins : n)/1024 + 2)));   // scans over bins and over particles
	SPHX_HIP(hipMalloc((void**)&ctx->slot, sizeof(uint32_t)*(size_t)n));
	SPHX_HIP(hipMalloc((void**)&ctx->tmp_hash, sizeof(uint32_t)*(size_t)n));
	SPHX_HIP(hipMalloc((void**)&ctx->tmp_index, sizeof(uint32_t)*(size_t)n));
	SPHX_HIP(hipMalloc((void**)&ctx->tmp_info, sizeof(uint2)*(size_t)n));
	SPHX_HIP(hipMalloc((void**)&ctx->eos_aux, sizeof(float4)*(size_t)n));
	SPHX_HIP(hipMalloc((void**)&ctx->neib_counts, sizeof(uint32_t)*(size_t)n));
	if (ctx->dev.turbmodel == SPHX_SPS)
		SPHX_HIP(hipMalloc((void**)&ctx->tau_pack, sizeof(float4)*2*(size_t)n));
	ctx->tile_capacity = n/8 + 4096;
	SPHX_HIP(hipMalloc((void**)&ctx->tiles, sizeof(uint32_t)*TILE_DESC*(size_t)ctx->tile_capacity));
	ctx->cells_reserved = (bins - 1)/4;
	SPHX_HIP(hipMalloc((void**)&ctx->cell_end_copy, sizeof(uint32_t)*(size_t)ctx->cells_reserved));
	SPHX_HIP(hipMalloc((void**)&ctx->cell_fluid_end, sizeof(uint32_t)*(size_t)ctx->cells_reserved));
	// row bundles x columns <= ceil(gs2/2) ceil(gs3/2) gs1 <= cells/4 rounded up generously for odd grid sizes
	SPHX_HIP(hipMalloc((void**)&ctx->tile_cols, sizeof(uint32_t)*((size_t)ctx->cells_reserved/2 + 1024)));
	ctx->reserved_particles = n;
	ctx->reserved_bins = bins;
	return SPHX_OK;
}

// Tile lists of the tiled forces kernel (forces.hip): the stream of list batches (room for 160 two-byte entries per particle on
// average incl. the padding of a chunk's lanes to its longest list, + a floor for small cases whose tiles are mostly padding),
// the lane tables and the run tables.  More than the neighbour list itself, so they are allocated on the first neighbour-list
// build that really tiles (sphx_build_neibs_sa decides: not for the formulations that have their own forces kernels, nor for
// SA_BOUNDARY with several fluids).  When the memory is not there the context simply keeps running on the generic kernels:
// tile_list stays NULL and tiles_built false.  A build that needs more than this reserves flags the tiling as overflowed.
int sphx_ensure_tile_lists(sphx_ctx *ctx)
{
	if (ctx->disable_tiles || !ctx->reserved_particles) return SPHX_OK;
	if (ctx->tile_list && ctx->tile_runs && ctx->tile_rows && ctx->tile_lane_rec && ctx->tile_lane_index) return SPHX_OK;
	const size_t n = ctx->reserved_particles;
	const size_t batches = n*5u/8u + 262144u;
	const size_t lanes = 2u*n + 262144u;
	if (batches >= ((size_t)1 << 31) || lanes >= ((size_t)1 << 31)) return SPHX_OK;      // 32-bit cursors
	// the ring of the walk reads TILE_AHEAD batches past the end of a wave's share without a clamp (acc_load_at, forces.hip): those
	// batches are allocated behind the capacity the tiling may fill, so a stream filled to the last batch still reads its own memory
	bool ok = hipMalloc((void**)&ctx->tile_list, sizeof(uint2)*64u*(batches + TILE_AHEAD)) == hipSuccess;
	ok = ok && hipMalloc((void**)&ctx->tile_runs, sizeof(uint32_t)*TILE_RUNTAB*(size_t)ctx->tile_capacity) == hipSuccess;
	ok = ok && hipMalloc((void**)&ctx->tile_rows, sizeof(uint32_t)*TILE_ROWDESC*(size_t)ctx->tile_capacity) == hipSuccess;
	ok = ok && hipMalloc((void**)&ctx->tile_lane_rec, sizeof(uint32_t)*lanes) == hipSuccess;
	ok = ok && hipMalloc((void**)&ctx->tile_lane_index, sizeof(uint32_t)*lanes) == hipSuccess;
	if (!ok) {
		(void)hipGetLastError();   // out of memory is not an error of the caller's command: the generic kernels take over
		void *ptrs[] = { ctx->tile_list, ctx->tile_runs, ctx->tile_rows, ctx->tile_lane_rec, ctx->tile_lane_index };
		for (void *q : ptrs) if (q) (void)hipFree(q);
		ctx->tile_list = nullptr; ctx->tile_runs = nullptr; ctx->tile_rows = nullptr; ctx->tile_lane_rec = nullptr; ctx->tile_lane_index = nullptr;
		ctx->tile_list_batches = ctx->tile_lane_cap = 0;
		return SPHX_OK;
	}
	ctx->tile_list_batches = (uint32_t)batches;
	ctx->tile_lane_cap = (uint32_t)lanes;
	return SPHX_OK;
}

extern "C" int sphx_reserve(sphx_ctx *ctx, uint32_t maxParticles)
{
	SPHX_REQUIRE(ctx != nullptr, "sphx_reserve: ctx is NULL");
	SPHX_HIP(hipSetDevice(ctx->device));
	return sphx_ensure_scratch(ctx, maxParticles);
}

// Kernel coefficients exactly as CUDAForcesEngine::setconstants computes them
// (src/cuda/forces.cu:274-309): float powers of h, double M_PI, result rounded to float.
static void kernel_coeffs(const sphx_params &sp, float &wcoeff, float &fcoeff, float &wsub)
{
	const float h = sp.slength;
	const float h2 = h*h;
	const float h3 = h2*h;
	const float h4 = h2*h2;
	const float h5 = h4*h;
	wsub = 0.0f;
	switch (sp.kerneltype) {
	case SPHX_CUBICSPLINE:
		wcoeff = 1.0f/(M_PI*h3); fcoeff = 3.0f/(4.0f*M_PI*h4); break;
	case SPHX_QUADRATIC:
		wcoeff = 15.0f/(16.0f*M_PI*h3); fcoeff = 15.0f/(32.0f*M_PI*h4); break;
	case SPHX_WENDLAND:
		wcoeff = 21.0f/(16.0f*M_PI*h3); fcoeff = 105.0f/(128.0f*M_PI*h5); break;
	case SPHX_GAUSSIAN: {
		const float R = sp.kernelradius;
		const float R2 = R*R;
		const float exp_R2 = exp(-R2);
		wsub = exp_R2;
		float kc = -2*exp_R2/3 * h3 * M_PI * R*(3+2*R2)
			+ h3 * 5.5683279968317078452848179821188357020136243902832439 * erf(R);
		kc = 1/kc;
		wcoeff = kc;
		kc *= 2/h2;
		fcoeff = kc;
		break;
	}
	default: wcoeff = fcoeff = NAN;
	}
}

extern "C" int sphx_set_constants(sphx_ctx *ctx, const sphx_params *sp)
{
	SPHX_REQUIRE(ctx && sp, "sphx_set_constants: NULL argument");
	SPHX_REQUIRE(sp->gridSize[0] && sp->gridSize[1] && sp->gridSize[2], "sphx_set_constants: empty grid");
	const uint64_t cells = (uint64_t)sp->gridSize[0]*sp->gridSize[1]*sp->gridSize[2];
	SPHX_REQUIRE(cells <= (0xFFFFFFFFull >> 2), "sphx_set_constants: more than MAX_CELLS cells"); // multi_gpu_defines.h:54
	int seen = 0;
	for (int i = 0; i < 3; ++i) {
		SPHX_REQUIRE(sp->coord[i] >= 0 && sp->coord[i] < 3, "sphx_set_constants: coord[] must be a permutation of 0,1,2");
		seen |= 1 << sp->coord[i];
	}
	SPHX_REQUIRE(seen == 7, "sphx_set_constants: coord[] must be a permutation of 0,1,2");
	SPHX_REQUIRE(sp->kerneltype >= SPHX_CUBICSPLINE && sp->kerneltype <= SPHX_GAUSSIAN, "sphx_set_constants: invalid kernel type");
	SPHX_REQUIRE(sp->numfluids >= 1 && sp->numfluids <= SPHX_MAX_FLUIDS, "sphx_set_constants: numfluids out of range");
	SPHX_REQUIRE(sp->neiblistsize >= 2 && sp->neibboundpos < sp->neiblistsize, "sphx_set_constants: invalid neighbour list geometry");
	// option combinations built into this library (the rest is SURVEY.md 8f "next")
	if (sp->sph_formulation < SPHX_SPH_F1 || sp->sph_formulation > SPHX_SPH_HA)
		return sphx_set_error(SPHX_ERR_INVALID, "sphx: invalid SPH formulation");
	if (sp->sph_formulation == SPHX_SPH_HA) {
		// Hu & Adams (BiFluidPoiseuille): rheology.hip's forces kernel
		if (sp->boundarytype != SPHX_DYN_BOUNDARY || sp->turbmodel != SPHX_LAMINAR_FLOW)
			return sphx_set_error(SPHX_ERR_UNSUPPORTED, "sphx: SPH_HA is built for DYN_BOUNDARY and LAMINAR_FLOW");
	}
	if (sp->sph_formulation == SPHX_SPH_GRENIER) {
		// the option set of the reference's Grenier problems (Bubble, LockExchange, RTInstability, OilJet): Wendland kernel,
		// dynamic boundaries, laminar flow, no density diffusion; grenier.hip
		if (sp->kerneltype != SPHX_WENDLAND || sp->boundarytype != SPHX_DYN_BOUNDARY)
			return sphx_set_error(SPHX_ERR_UNSUPPORTED, "sphx: SPH_GRENIER is built for the Wendland kernel with DYN_BOUNDARY");
		if (sp->densitydiffusiontype != SPHX_DENSITY_DIFFUSION_NONE)
			return sphx_set_error(SPHX_ERR_UNSUPPORTED, "sphx: SPH_GRENIER is built without density diffusion");
		if (sp->turbmodel != SPHX_LAMINAR_FLOW)
			return sphx_set_error(SPHX_ERR_UNSUPPORTED, "sphx: SPH_GRENIER is built for LAMINAR_FLOW (no artificial viscosity, no SPS)");
		if (sp->simflags & (SPHX_ENABLE_XSPH | SPHX_ENABLE_MOVING_BODIES))
			return sphx_set_error(SPHX_ERR_UNSUPPORTED, "sphx: SPH_GRENIER is built without XSPH and without moving bodies");
		SPHX_REQUIRE(sp->epsinterface == sp->epsinterface, "sphx_set_constants: SPH_GRENIER needs epsinterface (ProblemCore.cc:165-166 default 0.05)");
	}
	// SA_BOUNDARY: the neighbour engine (vertex section, VERTPOS) and the boundary-conditions engine of solid walls are built;
	// the forces / integration / filter engines answer SPHX_ERR_UNSUPPORTED for it (gamma terms, density summation)
	if (sp->boundarytype < SPHX_LJ_BOUNDARY || sp->boundarytype > SPHX_DYN_BOUNDARY)
		return sphx_set_error(SPHX_ERR_INVALID, "sphx: invalid boundary type");
	if (sp->boundarytype == SPHX_SA_BOUNDARY) {
		if (sp->kerneltype != SPHX_WENDLAND)
			return sphx_set_error(SPHX_ERR_UNSUPPORTED, "sphx: SA_BOUNDARY needs the Wendland kernel (src/cuda/gamma.cuh:241-250)");
		SPHX_REQUIRE(sp->neibboundpos + 2 <= sp->neiblistsize, "sphx_set_constants: SA_BOUNDARY needs a vertex section in the neighbour list");
	}
	// Brezzi diffusion is a term of the SA forces pass (not built); the engines that are built for SA_BOUNDARY do not read it
	if (sp->densitydiffusiontype != SPHX_DENSITY_DIFFUSION_NONE && sp->densitydiffusiontype != SPHX_COLAGROSSI &&
		sp->densitydiffusiontype != SPHX_FERRARI && sp->boundarytype != SPHX_SA_BOUNDARY)
		return sphx_set_error(SPHX_ERR_UNSUPPORTED, "sphx: Brezzi density diffusion (an SA_BOUNDARY option in the reference's problems) is not built");
	if (sp->rheologytype < SPHX_INVISCID || sp->rheologytype > SPHX_ZHU || sp->rheologytype == SPHX_GRANULAR)
		return sphx_set_error(SPHX_ERR_UNSUPPORTED, "sphx: INVISCID, NEWTONIAN and the generalized Newtonian rheologies (BINGHAM .. ZHU) are built, GRANULAR is not");
	if (sp->rheologytype > SPHX_NEWTONIAN) {
		// rheology.hip: effective viscosity + the forces that read it
		if ((sp->sph_formulation != SPHX_SPH_F1 && sp->sph_formulation != SPHX_SPH_HA) || sp->boundarytype != SPHX_DYN_BOUNDARY || sp->turbmodel != SPHX_LAMINAR_FLOW)
			return sphx_set_error(SPHX_ERR_UNSUPPORTED, "sphx: generalized Newtonian rheologies are built for SPH_F1 / SPH_HA, DYN_BOUNDARY and LAMINAR_FLOW");
		SPHX_REQUIRE(sp->limiting_kinvisc == sp->limiting_kinvisc, "sphx_set_constants: generalized Newtonian rheologies need limiting_kinvisc");
		for (uint32_t f = 0; f < sp->numfluids; ++f)
			SPHX_REQUIRE(sp->yield_strength[f] == sp->yield_strength[f] && sp->visc_nonlinear_param[f] == sp->visc_nonlinear_param[f] &&
				sp->visc_regularization_param[f] == sp->visc_regularization_param[f],
				"sphx_set_constants: generalized Newtonian rheologies need yield_strength / visc_nonlinear_param / visc_regularization_param for every fluid");
	}
	if (sp->rheologytype >= SPHX_NEWTONIAN) {
		if (sp->viscmodel != SPHX_MORRIS) {
			// visc_model<MONAGHAN | ESPANOL_REVENGA>: the forces kernel of rheology.hip
			SPHX_REQUIRE(sp->viscmodel == SPHX_MONAGHAN || sp->viscmodel == SPHX_ESPANOL_REVENGA, "sphx_set_constants: invalid viscous model");
			if (sp->rheologytype != SPHX_NEWTONIAN && sp->viscmodel == SPHX_ESPANOL_REVENGA)
				return sphx_set_error(SPHX_ERR_INVALID, "sphx: ESPANOL_REVENGA needs the NEWTONIAN rheology (src/cuda/cudasimframework.cu:186-193)");
			if ((sp->sph_formulation != SPHX_SPH_F1 && sp->sph_formulation != SPHX_SPH_HA) || sp->boundarytype != SPHX_DYN_BOUNDARY || sp->turbmodel != SPHX_LAMINAR_FLOW)
				return sphx_set_error(SPHX_ERR_UNSUPPORTED, "sphx: the MONAGHAN and ESPANOL_REVENGA viscous models are built for SPH_F1 / SPH_HA, DYN_BOUNDARY and LAMINAR_FLOW");
			if (sp->viscmodel == SPHX_ESPANOL_REVENGA)
				for (uint32_t f = 0; f < sp->numfluids; ++f)
					SPHX_REQUIRE(sp->visc2coeff[f] == sp->visc2coeff[f], "sphx_set_constants: ESPANOL_REVENGA needs the bulk viscosity (visc2coeff) of every fluid");
			if (sp->viscmodel == SPHX_MONAGHAN)
				SPHX_REQUIRE(sp->monaghan_visc_coeff == sp->monaghan_visc_coeff, "sphx_set_constants: MONAGHAN needs monaghan_visc_coeff");
		}
		if (sp->turbmodel == SPHX_ARTIFICIAL)
			return sphx_set_error(SPHX_ERR_UNSUPPORTED, "sphx: NEWTONIAN rheology is built with LAMINAR_FLOW or SPS");
		SPHX_REQUIRE(sp->compvisc == SPHX_KINEMATIC || sp->compvisc == SPHX_DYNAMIC, "sphx_set_constants: invalid computational viscosity");
		SPHX_REQUIRE(sp->avgop >= SPHX_ARITHMETIC && sp->avgop <= SPHX_GEOMETRIC, "sphx_set_constants: invalid averaging operator");
		for (uint32_t f = 0; f < sp->numfluids; ++f)
			SPHX_REQUIRE(sp->visccoeff[f] == sp->visccoeff[f], "sphx_set_constants: NEWTONIAN rheology needs visccoeff for every fluid");
	}
	if (sp->simflags & SPHX_ENABLE_DEM) {
		// DemLJForce is the LJ_BOUNDARY case of the finalize kernel (src/cuda/forces_kernel.def:4093-4102); other boundary models skip it
		SPHX_REQUIRE(sp->ewres > 0 && sp->nsres > 0 && sp->demdx == sp->demdx && sp->demdy == sp->demdy && sp->demzmin == sp->demzmin,
			"sphx_set_constants: ENABLE_DEM needs ewres, nsres, demdx, demdy, demzmin (computeDEMphysparams)");
	}
	if (sp->turbmodel == SPHX_KEPSILON) {
		// k-epsilon only with semi-analytical walls, as the reference (src/cuda/cudasimframework.cu:155)
		if (sp->boundarytype != SPHX_SA_BOUNDARY)
			return sphx_set_error(SPHX_ERR_INVALID, "sphx_set_constants: KEPSILON is only supported with SA_BOUNDARY");
		if (sp->rheologytype != SPHX_NEWTONIAN || sp->viscmodel != SPHX_MORRIS || sp->is_const_visc)
			return sphx_set_error(SPHX_ERR_UNSUPPORTED, "sphx: KEPSILON is built for a Newtonian fluid, the MORRIS viscous model and non-constant viscosity (FullViscSpec default)");
	} else
	if (sp->turbmodel != SPHX_ARTIFICIAL && sp->turbmodel != SPHX_SPS && sp->turbmodel != SPHX_LAMINAR_FLOW)
		return sphx_set_error(SPHX_ERR_UNSUPPORTED, "sphx: turbulence model not built");

	ctx->params = *sp;
	DevParams &d = ctx->dev;
	const DevParams planes_keep = d;   // planes are uploaded separately (setplanes) and survive a new setconstants
	memset(&d, 0, sizeof(d));
	d.numplanes = planes_keep.numplanes;
	memcpy(d.plane_normal, planes_keep.plane_normal, sizeof(d.plane_normal));
	memcpy(d.plane_gridpos, planes_keep.plane_gridpos, sizeof(d.plane_gridpos));
	memcpy(d.plane_pos, planes_keep.plane_pos, sizeof(d.plane_pos));
	for (int a = 0; a < 3; ++a) { d.gs[a] = (int)sp->gridSize[a]; d.cs[a] = sp->cellSize[a]; }
	d.c1 = sp->coord[0]; d.c2 = sp->coord[1]; d.c3 = sp->coord[2];
	d.gs1 = d.gs[d.c1];
	d.gs12 = d.gs[d.c1]*d.gs[d.c2];
	d.gsc2 = d.gs[d.c2]; d.gsc3 = d.gs[d.c3];
	d.csc1 = d.cs[d.c1]; d.csc2 = d.cs[d.c2]; d.csc3 = d.cs[d.c3];
	d.hs[d.c1] = 1; d.hs[d.c2] = d.gs1; d.hs[d.c3] = d.gs12;
	d.periodic = sp->periodic;
	d.neiblistsize = sp->neiblistsize; d.neibboundpos = sp->neibboundpos; d.stride = sp->neiblist_stride;
	d.kerneltype = sp->kerneltype; d.formulation = sp->sph_formulation; d.densitydiff = sp->densitydiffusiontype;
	d.boundarytype = sp->boundarytype; d.rheology = sp->rheologytype; d.turbmodel = sp->turbmodel;
	d.simflags = sp->simflags;
	d.slength = sp->slength; d.influenceradius = sp->influenceradius; d.deltap = sp->deltap;
	kernel_coeffs(*sp, d.wcoeff, d.fcoeff, d.wsub_gaussian);
	d.densityDiffCoeff = sp->densityDiffCoeff; d.epsxsph = sp->epsxsph;
	d.numfluids = sp->numfluids;
	for (int f = 0; f < SPHX_MAX_FLUIDS; ++f) {
		d.rho0[f] = sp->rho0[f]; d.bcoeff[f] = sp->bcoeff[f]; d.gammacoeff[f] = sp->gammacoeff[f];
		d.sscoeff[f] = sp->sscoeff[f]; d.sspowercoeff[f] = sp->sspowercoeff[f];
	}
	for (int a = 0; a < 3; ++a) d.gravity[a] = sp->gravity[a];
	d.artvisccoeff = sp->artvisccoeff; d.epsartvisc = sp->epsartvisc;
	d.smagfactor = sp->smagfactor; d.kspsfactor = sp->kspsfactor;
	d.dcoeff = sp->dcoeff; d.p1coeff = sp->p1coeff; d.p2coeff = sp->p2coeff; d.r0 = sp->r0;
	d.repack_a = sp->repack_a; d.repack_alpha = sp->repack_alpha;
	for (uint32_t f = 0; f < sp->numfluids; ++f)
		d.visccoeff[f] = (sp->rheologytype == SPHX_INVISCID || sp->visccoeff[f] != sp->visccoeff[f]) ? 0.0f : sp->visccoeff[f];
	d.compvisc = sp->compvisc; d.avgop = sp->avgop; d.is_const_visc = sp->is_const_visc;
	d.partsurf = (sp->partsurf == 0.0f) ? sp->r0*sp->r0 : sp->partsurf;
	// MK_BOUNDARY shares every code path of LJ_BOUNDARY (lists, sections, feedback bodies, Euler) but the force law
	d.MK_K = sp->MK_K; d.MK_d = sp->MK_d; d.MK_beta = sp->MK_beta;
	d.epsinterface = sp->epsinterface;
	for (int f = 0; f < SPHX_MAX_FLUIDS; ++f) {
		d.yield_strength[f] = sp->yield_strength[f]; d.visc_nonlinear_param[f] = sp->visc_nonlinear_param[f];
		d.visc_regularization_param[f] = sp->visc_regularization_param[f];
	}
	d.limiting_kinvisc = sp->limiting_kinvisc;
	d.ewres = sp->ewres; d.nsres = sp->nsres; d.demdx = sp->demdx; d.demdy = sp->demdy; d.demzmin = sp->demzmin;
	d.wo_z = sp->worldOrigin[2];
	d.viscmodel = sp->viscmodel; d.monaghan_visc_coeff = sp->monaghan_visc_coeff;
	for (int f = 0; f < SPHX_MAX_FLUIDS; ++f) d.visc2coeff[f] = sp->visc2coeff[f];
	d.dem = ctx->dem; d.dem_w = ctx->dem_w; d.dem_h = ctx->dem_h;
	d.mk_mask = (sp->boundarytype == SPHX_MK_BOUNDARY) ? 0xFFFFFFFFu : 0u;
	if (sp->boundarytype == SPHX_MK_BOUNDARY) d.boundarytype = SPHX_LJ_BOUNDARY;
	ctx->have_params = true;
	// EOS rows left behind by an Euler step were made with the coefficients this call replaces: a vouch for them must not hold
	ctx->eos_tag_vel = nullptr; ctx->eos_tag_n = 0; ctx->eos_armed = false;
	return SPHX_OK;
}

extern "C" int sphx_get_params(sphx_ctx *ctx, sphx_params *out)
{
	SPHX_REQUIRE(ctx && out, "sphx_get_params: NULL argument");
	SPHX_REQUIRE(ctx->have_params, "sphx_get_params: constants not set");
	*out = ctx->params;
	return SPHX_OK;
}

extern "C" int sphx_set_planes(sphx_ctx *ctx, const float *normals, const int32_t *gridPos, const float *pos, int numPlanes)
{
	SPHX_REQUIRE(ctx != nullptr, "sphx_set_planes: NULL ctx");
	SPHX_REQUIRE(numPlanes >= 0 && numPlanes <= SPHX_MAX_PLANES, "sphx_set_planes: too many planes");
	SPHX_REQUIRE(numPlanes == 0 || (normals && gridPos && pos), "sphx_set_planes: missing array");
	ctx->dev.numplanes = (uint32_t)numPlanes;
	for (int k = 0; k < numPlanes; ++k)
		for (int a = 0; a < 3; ++a) {
			ctx->dev.plane_normal[k][a] = normals[3*k + a];
			ctx->dev.plane_gridpos[k][a] = gridPos[3*k + a];
			ctx->dev.plane_pos[k][a] = pos[3*k + a];
		}
	return SPHX_OK;
}

// setDEM / unsetDEM (src/cuda/forces.cu:937-958): the reference keeps the map in a 2D texture with clamped addressing and linear
// filtering; here it is a plain row-major array and the filtering is written out (dem_interpol, sphx_internal.h)
extern "C" int sphx_set_dem(sphx_ctx *ctx, const float *hDem, int width, int height)
{
	SPHX_REQUIRE(ctx != nullptr, "sphx_set_dem: NULL ctx");
	if (ctx->dem) { (void)hipFree(ctx->dem); ctx->dem = nullptr; }
	ctx->dem_w = ctx->dem_h = 0;
	if (hDem) {
		SPHX_REQUIRE(width > 0 && height > 0, "sphx_set_dem: empty DEM");
		SPHX_HIP(hipMalloc((void**)&ctx->dem, sizeof(float)*(size_t)width*(size_t)height));
		SPHX_HIP(hipMemcpy(ctx->dem, hDem, sizeof(float)*(size_t)width*(size_t)height, hipMemcpyHostToDevice));
		ctx->dem_w = width; ctx->dem_h = height;
	}
	ctx->dev.dem = ctx->dem; ctx->dev.dem_w = ctx->dem_w; ctx->dev.dem_h = ctx->dem_h;
	return SPHX_OK;
}

extern "C" int sphx_set_gravity(sphx_ctx *ctx, const float g[3])
{
	SPHX_REQUIRE(ctx && g, "sphx_set_gravity: NULL argument");
	for (int a = 0; a < 3; ++a) { ctx->params.gravity[a] = g[a]; ctx->dev.gravity[a] = g[a]; }
	return SPHX_OK;
}

// The reference uploads these tables with cudaMemcpyToSymbol on the default stream, in order with its kernels.  Here the
// engines run on whatever stream the caller passes, so the upload is deferred to the next engine call that reads the
// tables (forces, Euler) and issued on THAT call's stream: kernels already enqueued there keep the old values, kernels
// enqueued there afterwards see the new ones.  One stream at a time: an engine call on another stream that reads the tables is
// not ordered against that copy -- a caller that runs forces and Euler on different streams orders them itself (the drivers
// here use one compute stream).  Not for stream capture: the copy reads a pinned staging slot when it executes, and a slot is
// re-used after SPHX_RB_RING uploads behind a host-side event wait.
static int upload_rb(sphx_ctx *ctx)
{
	ctx->rb_dirty = true;
	return SPHX_OK;
}

int sphx_rb_flush(sphx_ctx *ctx, hipStream_t st)
{
	if (!ctx->rb_dirty) return SPHX_OK;
	const uint32_t k = ctx->rb_ring++ % SPHX_RB_RING;
	if (ctx->rb_ring > SPHX_RB_RING)      // the slot was used before: its copy must have executed before it is overwritten
		SPHX_HIP(hipEventSynchronize(ctx->rb_staged[k]));
	ctx->rb_staging[k] = ctx->rb_host;
	SPHX_HIP(hipMemcpyAsync(ctx->rb_dev, &ctx->rb_staging[k], sizeof(RbParams), hipMemcpyHostToDevice, st));
	SPHX_HIP(hipEventRecord(ctx->rb_staged[k], st));
	ctx->rb_dirty = false;
	return SPHX_OK;
}

static int set_rb_cg(sphx_ctx *ctx, const int32_t *cgGridPos, const float *cgPos, int numbodies, bool forces, bool euler)
{
	SPHX_REQUIRE(ctx && cgGridPos && cgPos, "sphx_set_rb_cg: NULL argument");
	SPHX_REQUIRE(numbodies >= 0 && numbodies <= SPHX_MAX_BODIES, "sphx_set_rb_cg: too many bodies");
	for (int b = 0; b < numbodies; ++b)
		for (int a = 0; a < 3; ++a) {
			if (forces) { ctx->rb_host.cgGridPos[b][a] = cgGridPos[3*b + a]; ctx->rb_host.cgPos[b][a] = cgPos[3*b + a]; }
			if (euler) { ctx->rb_host.cgGridPosE[b][a] = cgGridPos[3*b + a]; ctx->rb_host.cgPosE[b][a] = cgPos[3*b + a]; }
		}
	return upload_rb(ctx);
}
extern "C" int sphx_set_rb_cg(sphx_ctx *ctx, const int32_t *cgGridPos, const float *cgPos, int numbodies)
{ return set_rb_cg(ctx, cgGridPos, cgPos, numbodies, true, true); }
extern "C" int sphx_set_rb_cg_forces(sphx_ctx *ctx, const int32_t *cgGridPos, const float *cgPos, int numbodies)
{ return set_rb_cg(ctx, cgGridPos, cgPos, numbodies, true, false); }
extern "C" int sphx_set_rb_cg_integration(sphx_ctx *ctx, const int32_t *cgGridPos, const float *cgPos, int numbodies)
{ return set_rb_cg(ctx, cgGridPos, cgPos, numbodies, false, true); }

extern "C" int sphx_set_rb_start(sphx_ctx *ctx, const int32_t *rbfirstindex, int numbodies)
{
	SPHX_REQUIRE(ctx && rbfirstindex, "sphx_set_rb_start: NULL argument");
	SPHX_REQUIRE(numbodies >= 0 && numbodies <= SPHX_MAX_BODIES, "sphx_set_rb_start: too many bodies");
	for (int b = 0; b < numbodies; ++b) ctx->rb_host.rbstart[b] = rbfirstindex[b];
	return upload_rb(ctx);
}

extern "C" int sphx_set_rb_motion(sphx_ctx *ctx, const float *trans, const float *steprot,
	const float *linearvel, const float *angularvel, int numbodies)
{
	SPHX_REQUIRE(ctx != nullptr, "sphx_set_rb_motion: NULL ctx");
	SPHX_REQUIRE(numbodies >= 0 && numbodies <= SPHX_MAX_BODIES, "sphx_set_rb_motion: too many bodies");
	for (int b = 0; b < numbodies; ++b) {
		for (int a = 0; a < 3; ++a) {
			if (trans) ctx->rb_host.trans[b][a] = trans[3*b + a];
			if (linearvel) ctx->rb_host.linearvel[b][a] = linearvel[3*b + a];
			if (angularvel) ctx->rb_host.angularvel[b][a] = angularvel[3*b + a];
		}
		if (steprot) for (int a = 0; a < 9; ++a) ctx->rb_host.steprot[b][a] = steprot[9*b + a];
	}
	return upload_rb(ctx);
}

extern "C" int sphx_memset_async(void *ptr, int value, size_t bytes, void *stream)
{
	if (!bytes) return SPHX_OK;
	SPHX_REQUIRE(ptr != nullptr, "sphx_memset_async: NULL pointer");
	SPHX_HIP(hipMemsetAsync(ptr, value, bytes, (hipStream_t)stream));
	return SPHX_OK;
}

// ---- device memory service (see include/sphx.h) ----
extern "C" int sphx_device_count(int *count)
{
	SPHX_REQUIRE(count != nullptr, "sphx_device_count: NULL argument");
	SPHX_HIP(hipGetDeviceCount(count));
	return SPHX_OK;
}
extern "C" int sphx_set_device(int device) { SPHX_HIP(hipSetDevice(device)); return SPHX_OK; }
extern "C" int sphx_get_device(int *device)
{
	SPHX_REQUIRE(device != nullptr, "sphx_get_device: NULL argument");
	SPHX_HIP(hipGetDevice(device));
	return SPHX_OK;
}
extern "C" int sphx_device_synchronize(void) { SPHX_HIP(hipDeviceSynchronize()); return SPHX_OK; }
extern "C" int sphx_malloc(void **ptr, size_t bytes)
{
	SPHX_REQUIRE(ptr != nullptr, "sphx_malloc: NULL argument");
	*ptr = nullptr;
	if (!bytes) return SPHX_OK;
	SPHX_HIP(hipMalloc(ptr, bytes));
	return SPHX_OK;
}
extern "C" int sphx_free(void *ptr)
{
	if (ptr) SPHX_HIP(hipFree(ptr));
	return SPHX_OK;
}
extern "C" int sphx_memset(void *ptr, int value, size_t bytes)
{
	if (!bytes) return SPHX_OK;
	SPHX_REQUIRE(ptr != nullptr, "sphx_memset: NULL pointer");
	SPHX_HIP(hipMemset(ptr, value, bytes));
	return SPHX_OK;
}
static int copy_blocking(void *dst, const void *src, size_t bytes, hipMemcpyKind kind, const char *what)
{
	if (!bytes) return SPHX_OK;
	if (!dst || !src) return sphx_set_error(SPHX_ERR_INVALID, std::string(what) + ": NULL pointer");
	SPHX_HIP(hipMemcpy(dst, src, bytes, kind));
	return SPHX_OK;
}
extern "C" int sphx_memcpy_h2d(void *dst, const void *src, size_t bytes) { return copy_blocking(dst, src, bytes, hipMemcpyHostToDevice, "sphx_memcpy_h2d"); }
extern "C" int sphx_memcpy_d2h(void *dst, const void *src, size_t bytes) { return copy_blocking(dst, src, bytes, hipMemcpyDeviceToHost, "sphx_memcpy_d2h"); }
extern "C" int sphx_memcpy_d2d(void *dst, const void *src, size_t bytes) { return copy_blocking(dst, src, bytes, hipMemcpyDeviceToDevice, "sphx_memcpy_d2d"); }
