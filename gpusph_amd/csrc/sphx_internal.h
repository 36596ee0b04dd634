// sphx_internal.h -- shared declarations of the gfx950 WCSPH engine (not part of the C ABI).
#ifndef SPHX_INTERNAL_H
#define SPHX_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include <utility>
#include "sphx.h"

// ---- data model (GPUSPH: src/particleinfo.h:79-300, src/multi_gpu_defines.h:56-83,
//      src/common_types.h:57-72, src/hashkey.h:44-47) ------------------------------------------
typedef ushort4 particleinfo;
typedef uint16_t neibdata;

#define CELLTYPE_BITMASK   (~(3u << 30))
#define CELL_HASH_MAX      0xFFFFFFFFu
#define EMPTY_SEGMENT      0xFFFFFFFFu
#define CELL_EMPTY         0xFFFFFFFFu
#define CELLNUM_SHIFT      11
#define CELLNUM_ENCODED    (1u << CELLNUM_SHIFT)
#define NEIBINDEX_MASK     (CELLNUM_ENCODED - 1u)
#define NEIBS_END          0xFFFFu

enum { PT_FLUID = 0, PT_BOUNDARY = 1, PT_VERTEX = 2, PT_TESTPOINT = 3, PT_NONE = 4 };
#define PART_FLAG_START      (1u << 3)
#define FG_COMPUTE_FORCE     (PART_FLAG_START << 0)
#define FG_MOVING_BOUNDARY   (PART_FLAG_START << 1)
#define FG_SURFACE           (PART_FLAG_START << 6)
#define FG_INTERFACE         (PART_FLAG_START << 7)

#define SPHX_BLOCK_FORCES 128   // one CFL entry per 128 particles (getFmaxElements contract)

// forces tiles (see forces.hip "Tiled path"): a tile is a k x 2 x 2 block of cells (k along COORD1)
#ifndef TILE_THREADS
#define TILE_THREADS  512                  // threads of the tiled kernel's workgroup: 8 waves, two per SIMD
#define TILE_PMAX     640                  // home particles per tile: ten chunks of 64, dealt out to the eight waves batch by batch
#define TILE_WCAP     2876                 // window records that fit LDS (48 B each, 1 workgroup per CU) next to the partial sums and the lane records
#define TILE_WCAP_SPS 1720                 // ... with the SPS stress tensor of every window particle (80 B each)
#define TILE_WCAP_SPS1 2156                // ... of an SPS run with one fluid (64 B each: no EOS rows, see tau_pack_kernel)
#define TILE_WGS_PER_CU 1                  // persistent workgroups per CU (LDS bound)
#endif
#define SA_WALL_CACHE_ENTRIES 96           // boundary-section entries per wall particle whose |grad gamma_as| is kept (more: recomputed)
#define TILE_HROWS    4                    // home rows: 2 (COORD2) x 2 (COORD3)
#define TILE_WROWS    16                   // window rows: 4 x 4
#define TILE_WAVES    (TILE_THREADS/64)    // waves of a tile's workgroup ...
#define TILE_RPW      (TILE_WROWS/TILE_WAVES)   // ... and the window rows each of them stages
#define TILE_CHUNKS   (TILE_PMAX/64)       // chunks of 64 home particles (one per lane of a wave) a tile can have
#define TILE_RUNS_MAX (TILE_CHUNKS + TILE_WAVES - 1)   // runs of a tile: a run is a stretch of ONE chunk's list batches walked by ONE wave
#define TILE_MAXCELLS 14                   // cells per tile along COORD1
#define TILE_KW       16                   // window columns (TILE_MAXCELLS + 2)
#define TILE_DESC     16                   // uint32 per tile: g2, g3, firstCell, numCells, first[4], count[4], window, flags (build_tiles_kernel),
                                           // first batch of its list stream, first entry of its lane tables (tile_lists_kernel)
#define TILE_ROWDESC  32                   // uint32 per tile (tile_rows, written by tile_lists_kernel): first record of each of the 16
                                           // window rows, then records per row and window slot of the row's first record as uint16 pairs
// uint32 per tile of the run table (tile_runs, tile_lists_kernel): who walks what.
//   [w], w < 8        wave w: first run | runs << 5 | first batch (tile-relative) << 10 | batches << 22
//   [8]               chunks | home particles << 8 | runs << 24
//   [9 + c], c < 10   chunk c: first run | runs << 8   (its partial sums, in list order)
//   [24 + r], r < 17  run r: chunk | batches of the fluid section << 4 | batches of the second section << 12 | TILE_RUN_LAST if it ends its chunk
#define TILE_RUNTAB   48
#define TILE_RUN_LAST (1u << 20)
#define TILE_RT_CHUNK 9
#define TILE_RT_RUN   24
#define TILE_NB       4                    // neighbours per batch in the tiled pair loop
#define TILE_AHEAD    4                    // list batches kept in flight (register ring)

// ---- per-kernel constants, passed BY VALUE as a kernel argument (kernarg/SGPR resident;
//      replaces the reference's ~70 __constant__ symbols, so there is no per-device global
//      state and one library instance serves any number of devices / host threads) ----------
struct DevParams {
	int      gs[3];           // grid size per axis
	float    cs[3];           // cell size per axis
	int      hs[3];           // hash stride per axis: hash = gx*hs[0] + gy*hs[1] + gz*hs[2]
	int      c1, c2, c3;      // axis index of COORD1..3
	int      gs1, gs12;       // gridSize[COORD1], gridSize[COORD1]*gridSize[COORD2]
	int      gsc2, gsc3;      // gridSize[COORD2], gridSize[COORD3]
	float    csc1, csc2, csc3;// cell size along COORD1..3 (kept as scalars: a run-time index into cs[] would send the kernel
	                          // argument block to scratch memory)
	uint32_t periodic;
	uint32_t neiblistsize, neibboundpos;
	uint64_t stride;
	int      kerneltype, formulation, densitydiff, boundarytype, rheology, turbmodel;
	uint64_t simflags;
	float    slength, influenceradius, deltap;
	float    wcoeff, fcoeff, wsub_gaussian;
	float    densityDiffCoeff, epsxsph;
	uint32_t numfluids;
	float    rho0[SPHX_MAX_FLUIDS], bcoeff[SPHX_MAX_FLUIDS], gammacoeff[SPHX_MAX_FLUIDS];
	float    sscoeff[SPHX_MAX_FLUIDS], sspowercoeff[SPHX_MAX_FLUIDS];
	float    gravity[3];
	float    artvisccoeff, epsartvisc, smagfactor, kspsfactor;
	float    dcoeff, p1coeff, p2coeff, r0;   // Lennard-Jones boundary repulsion
	float    repack_a, repack_alpha;         // repacking (d_repack_a, d_repack_alpha: src/cuda/phys_core.cu:93-94)
	float    visccoeff[SPHX_MAX_FLUIDS];     // Newtonian: nu (KINEMATIC) or mu (DYNAMIC) per fluid; 0 when inviscid
	int      compvisc, avgop, is_const_visc; // FullViscSpec (src/visc_spec.h:255-312)
	float    partsurf;                       // d_partsurf: wall friction of planes
	float    MK_K, MK_d, MK_beta;            // Monaghan-Kajtar repulsion
	float    epsinterface;                   // SPH_GRENIER interface term
	float    yield_strength[SPHX_MAX_FLUIDS], visc_nonlinear_param[SPHX_MAX_FLUIDS], visc_regularization_param[SPHX_MAX_FLUIDS];
	float    limiting_kinvisc;               // generalized Newtonian rheologies
	float    ewres, nsres, demdx, demdy, demzmin, wo_z;   // ENABLE_DEM (+ d_worldOrigin.z)
	const float *dem; int dem_w, dem_h;      // the height map (sphx_set_dem), row-major [h][w]
	int      viscmodel; float monaghan_visc_coeff; float visc2coeff[SPHX_MAX_FLUIDS];   // visc_model<MONAGHAN | ESPANOL_REVENGA>
	uint32_t mk_mask;                        // all ones when the repulsive boundary model is MK (boundarytype then reads LJ)
	uint32_t numplanes;                       // geometric planes (src/planes.h:43-47, MAX_PLANES src/particledefine.h:325)
	float    plane_normal[SPHX_MAX_PLANES][3];
	int      plane_gridpos[SPHX_MAX_PLANES][3];
	float    plane_pos[SPHX_MAX_PLANES][3];
};

// rigid-body tables live in device memory owned by the ctx (1.6 KB, too big for kernarg)
struct RbParams {
	int   cgGridPos[SPHX_MAX_BODIES][3];    // forces engine's copy (d_rbcgGridPos/d_rbcgPos of cuforces)
	float cgPos[SPHX_MAX_BODIES][3];
	int   cgGridPosE[SPHX_MAX_BODIES][3];   // integration engine's copy (cueuler): the reference uploads them at different
	float cgPosE[SPHX_MAX_BODIES][3];       // points of a step (FORCES_/EULER_UPLOAD_OBJECTS_CG)
	int   rbstart[SPHX_MAX_BODIES];
	float trans[SPHX_MAX_BODIES][3];
	float steprot[SPHX_MAX_BODIES][9];
	float linearvel[SPHX_MAX_BODIES][3];
	float angularvel[SPHX_MAX_BODIES][3];
};

struct NeibsCounters {   // device counters of cuneibs (src/cuda/buildneibs_kernel.cu:88-93)
	int numInteractions;
	int maxFluidBoundaryNeibs;
	int maxVertexNeibs;
	int hasTooManyNeibs;
	int hasMaxNeibs[3];
	int pad;
	unsigned long long numInteractions64;   // the same sum without the reference's 32-bit wrap (2^31 is 33 M particles x 65 neighbours)
};
// what every wave of the list build adds to those counters goes to one of NEIBS_SPREAD partial sets first (by block number) and
// is folded into the counters by a one-block kernel behind the build: three atomics per wave on ONE address were 1.5 M
// serialised read-modify-writes in one L2 channel per build of 32 M particles, and slowed the whole kernel down
#define NEIBS_SPREAD 256
struct NeibsSpread { int maxFluidBoundaryNeibs, maxVertexNeibs; unsigned long long numInteractions; };

#define SPHX_LIST_PARTS_MAX 16
#define SPHX_LIST_PARTS_DEFAULT 1
struct sphx_ctx {
	int         device;
	bool        have_params;
	sphx_params params;
	DevParams   dev;
	RbParams    rb_host;
	RbParams   *rb_dev;
	// rigid-body tables reach the device in stream order: the setters only mark them dirty; the next engine call that reads
	// them copies them from a ring of pinned staging slots on ITS stream before launching (sphx_rb_flush)
	bool        rb_dirty;
	RbParams   *rb_staging;              // [SPHX_RB_RING] pinned
	hipEvent_t  rb_staged[16];           // copy of slot k has executed
	uint32_t    rb_ring;
	NeibsCounters *counters_dev;
	// sort scratch
	uint32_t    reserved_particles;
	uint32_t    reserved_bins;
	uint32_t   *bin_count;     // [bins+1]
	uint32_t   *bin_start;     // [bins+1] exclusive scan
	uint32_t   *scan_partials; // per scan block
	uint32_t   *slot;          // [n]
	uint32_t   *tmp_hash;      // [n]
	uint32_t   *tmp_index;     // [n]
	uint2      *tmp_info;      // [n] particleinfo as 8 bytes
	float4     *eos_aux;       // [n] per-particle EOS pre-pass of the forces engine
	float      *dem; int dem_w, dem_h;   // ENABLE_DEM: the height map (sphx_set_dem)
	float4     *tau_pack;      // [2n] SPS: tau repacked as two float4 rows per particle for the LDS window (tiled kernel)
	float      *dt_scratch;    // 1 float, for the sync dtreduce
	// forces tiles, built by sphx_build_neibs
	uint32_t   *tiles;         // [tile_capacity][TILE_DESC]
	uint32_t   *tile_rows;     // [tile_capacity][TILE_ROWDESC]: the window rows of every tile, laid out once per neighbour-list build
	uint32_t   *tile_cols;     // [row bundles][gs1]: window records of a column (16 rows), bit 31 = a cell of it holds fluid
	uint32_t   *tile_ctl;      // [0] = number of tiles, [1] = overflow flag (generic kernel takes over), [2] finished groups, [4..11] tile tickets,
	                           // [12] batches of the list stream handed out, [13] entries of the lane tables handed out
	uint32_t   *cell_end_copy; // [cells] cellEnd of the build the tiles belong to
	uint32_t   *cell_fluid_end;// [cells] first non-fluid particle of each cell (neighbour-list build)
	uint32_t   *neib_counts;   // [n] entries of the fluid section | entries of the second section << 16 of every list (build_neibs_kernel)
	// tile lists (forces.hip "Tile lists"): the neighbour lists of the tiled particles as the tiled kernel walks them -- one
	// stream of 512-byte batches (64 lanes x 4 window offsets) per tile, in the order its waves consume them
	uint2      *tile_list;     // [tile_list_batches][64]
	uint32_t    tile_list_batches;
	uint32_t   *tile_lane_rec; // [tile_lane_cap] per lane of every chunk: window offset of the particle's own row | flags << 16
	uint32_t   *tile_lane_index;// [tile_lane_cap] ... its particle (0xFFFFFFFF: idle lane)
	uint32_t    tile_lane_cap;
	uint32_t   *tile_runs;     // [tile_capacity][TILE_RUNTAB]
	uint32_t    tile_capacity;
	uint32_t    cells_reserved;
	bool        tiles_built;
	int         tiles_overflow;// device-side overflow flag of the tiling as last seen by the host: -1 not seen yet (the forces
	                           // engine then launches the generic kernel as a guarded stand-by), 0 tiles usable, 1 generic kernels
	uint32_t   *ovf_host;      // pinned: copy of tile_ctl[0..1] made behind every tiled build
	hipEvent_t  ovf_event;     // ... has arrived
	// the tiling of a neighbour-list build (tile_columns_kernel, build_tiles_kernel: a few hundred waves walking serially) runs
	// on a stream of the context's own next to build_neibs_kernel, forked from and joined to the caller's stream by events
	hipStream_t side_stream;
	hipEvent_t  side_fork, side_join;
	// a tiled build in parts (sphx_build_neibs_sa): the lists of part k are there
	hipEvent_t  list_part[SPHX_LIST_PARTS_MAX];
	int         list_parts;    // parts of a tiled list build (SPHX_LIST_PARTS in the environment when the context was created)
	bool        list_part_events;
	bool        ovf_pending;
	bool        disable_tiles; // SPHX_DISABLE_TILES=1 in the environment (A/B testing)
	bool        neibs_mfma;    // SPHX_NEIBS_MFMA=1 in the environment when the context was created: the list build's prepass on the matrix cores (neibs_build.hip)
	int         tile_debug;    // SPHX_TILE_DEBUG (timing experiments; only with -DSPHX_TILE_DEBUG_BUILD)
	unsigned long long *tile_prof;   // ... & 16: phase timers of the tiled forces kernel
	bool        time_forces;   // sphx_forces_timing: bracket the dominant forces kernel with HIP events
	std::vector<std::pair<hipEvent_t, hipEvent_t> > *forces_events;
	const void *tiles_cellstart, *tiles_neibslist;
	// SA_BOUNDARY: [0] = number of, [1..] = the fluid particles whose list has boundary elements (built with the neighbour list;
	// the boundary-element terms are evaluated one element per lane for these, sa_bounds.hip); sa_wall_neibslist = that list
	uint32_t   *sa_wall;
	const void *sa_wall_neibslist;
	// ... and, of a run with ENABLE_MOVING_BODIES, the VERTEX particles whose list has boundary elements (same layout, same list):
	// the density summation integrates their gamma by the same boundary terms (sa_density_sum_wall_moving_kernel)
	uint32_t   *sa_wall_vert;
	// ... and ALL boundary elements / ALL vertex particles below the particleRangeEnd of that build (sa_rows_range), same layout: the two
	// boundary-condition passes take one thread per ROW of these instead of one per particle (sa_segment_bc_kernel, sa_vertex_bc_kernel)
	uint32_t   *sa_rows_bound, *sa_rows_vert;
	uint32_t    sa_rows_range;
	// |grad gamma_as| of every (wall particle, entry of its boundary section) as the density summation / gamma quadrature of a
	// step evaluates it at the new positions, kept for the forces pass that follows at those very positions (sa_wall.hip):
	// [wall particle][SA_WALL_CACHE_ENTRIES] floats + a tag per wall particle {position bits, list generation}
	float      *sa_wall_cache;
	float4     *sa_wall_tag;
	uint32_t    sa_wall_capacity, sa_wall_gen;
	// a run with open boundaries: two rows per wall particle, {sum_s grad gamma_as at step n, list generation} and the position
	// bits it was summed at -- from the forces pass of a step to its density summations (SaWallCache::gsum)
	float4     *sa_wall_gsum;
	// ... and the wall particles that have a segment of an OPEN face in reach ([0] = number of, [1..]; any order), left by the density
	// summation of such a run for the Brezzi diffusion that follows it on the same list (only those exchange density with a face)
	uint32_t   *sa_wall_open;
	const void *sa_wall_open_neibslist;
	uint32_t    sa_wall_open_gen;
	uint32_t    tile_grid;     // persistent grid size: 2 workgroups per CU
	// The EOS rows of the forces engine (eos_aux) written by the Euler step that writes the densities they are made of
	// (sphx_eos_rows_follow_euler): eos_tag_* = the velocity buffer and the row count of that step; eos_armed = the caller has
	// stated that this buffer is unchanged since (sphx_eos_rows_current), for the next sphx_forces_basicstep only
	bool        eos_follow, eos_armed;
	const void *eos_tag_vel;
	uint32_t    eos_tag_n;
	// open boundaries (sa_io.hip): the index of the pass in flight, [0] = how many rows, [1..] = which (one pass at a time per
	// context, in stream order)
	uint32_t   *open_rows;
	uint32_t    open_rows_cap;
};

// ---- error plumbing ---------------------------------------------------------------------------
int  sphx_set_error(int code, const std::string &msg);
#define SPHX_HIP(call) do { hipError_t _e = (call); if (_e != hipSuccess) \
	return sphx_set_error(SPHX_ERR_RUNTIME, std::string(#call) + ": " + hipGetErrorString(_e)); } while (0)
#define SPHX_LAUNCH_CHECK(name) do { hipError_t _e = hipGetLastError(); if (_e != hipSuccess) \
	return sphx_set_error(SPHX_ERR_RUNTIME, std::string("launch of " name ": ") + hipGetErrorString(_e)); } while (0)
#define SPHX_REQUIRE(cond, msg) do { if (!(cond)) return sphx_set_error(SPHX_ERR_INVALID, msg); } while (0)

static inline uint32_t div_up_u(uint32_t a, uint32_t b) { return (a + b - 1)/b; }
static inline uint32_t round_up_u(uint32_t a, uint32_t b) { return div_up_u(a, b)*b; }

int sphx_ensure_scratch(sphx_ctx *ctx, uint32_t numParticles);
int sphx_ensure_tile_lists(sphx_ctx *ctx);
static inline void sphx_tiles_overflow_poll(sphx_ctx *ctx)
{
	if (ctx->tiles_overflow == -1 && ctx->ovf_pending && hipEventQuery(ctx->ovf_event) == hipSuccess) {
		ctx->tiles_overflow = ctx->ovf_host[1] ? 1 : 0;
		ctx->ovf_pending = false;
	}
}
// repacking forces (filters.hip), reached through sphx_forces_basicstep(run_mode = SPHX_REPACK)
#define SPHX_RB_RING 16
int sphx_fidelity_forces_launch(sphx_ctx *ctx, void *forces, float *cfl,
	const void *pos, const void *vel, const void *info, const uint32_t *hash,
	const uint32_t *cellStart, const uint16_t *neibsList, const float *effvisc,
	uint32_t numParticles, uint32_t fromParticle, uint32_t toParticle,
	float slength, float influenceradius, uint32_t cflOffset, uint32_t *h_numBlocks, void *stream);   // rheology.hip
void sphx_fidelity_rows_launch(sphx_ctx *ctx, const void *vel, const void *info, uint32_t numParticles, hipStream_t st);   // rheology.hip
int sphx_rb_flush(sphx_ctx *ctx, hipStream_t st);
int sphx_xsph_launch(sphx_ctx *ctx, void *xsph, const void *pos, const void *vel, const void *info, const uint32_t *hash,
	const uint32_t *cellStart, const uint16_t *neibsList, uint32_t fromParticle, uint32_t toParticle, hipStream_t st);
int sphx_neibs_list_launch(sphx_ctx *ctx, uint16_t *neibsList, const void *pos, const void *info, const uint32_t *hash,
	const uint32_t *cellStart, const uint32_t *cellEnd, const void *vertices, const void *boundElements,
	void *vertPos0, void *vertPos1, void *vertPos2, uint32_t numParticles, uint32_t particleRangeEnd,
	float sqinfluenceradius, float boundNlSqInflRad, hipStream_t st);   // neibs_build.hip
int sphx_neibs_list_launch_part(sphx_ctx *ctx, uint16_t *neibsList, const void *pos, const void *info, const uint32_t *hash,
	const uint32_t *cellStart, const uint32_t *cellEnd, const void *vertices, const void *boundElements,
	void *vertPos0, void *vertPos1, void *vertPos2, uint32_t numParticles, uint32_t firstParticle, uint32_t particleRangeEnd,
	float sqinfluenceradius, float boundNlSqInflRad, hipStream_t st);   // ... of [firstParticle, particleRangeEnd)
// the tile lists of the tiles whose LAST home particle lies in [homeFrom, homeTo) (the whole tiling: 0, 0xFFFFFFFF)
int sphx_tile_lists_launch(sphx_ctx *ctx, const uint16_t *neibsList, const void *info, const uint32_t *hash, const uint32_t *cellStart, bool sa, hipStream_t st,
	uint32_t homeFrom = 0u, uint32_t homeTo = 0xFFFFFFFFu);
// SA_BOUNDARY engines over the tiles (forces.hip): which sums the tiled kernel forms
#define SPHX_SA_TILE_FORCES 0
#define SPHX_SA_TILE_DSUM 1
#define SPHX_SA_TILE_DIFF 2
int sphx_sa_tiles_run(sphx_ctx *ctx, int mode, void *forces, const void *pos, const void *vel, const void *newPos,
	const void *info, const uint32_t *hash, const uint32_t *cellStart, const uint16_t *neibsList, const void *gGam,
	uint32_t numParticles, uint32_t fromParticle, uint32_t toParticle, float dt, hipStream_t stream,
	bool *used, const uint32_t **guard);
int sphx_repack_launch(sphx_ctx *ctx, void *forces, float *cfl, void *rbforces, void *rbtorques,
	const void *pos, const void *vel, const void *info, const uint32_t *hash,
	const uint32_t *cellStart, const uint16_t *neibsList,
	uint32_t fromParticle, uint32_t toParticle, uint32_t cflOffset, uint32_t numBlocks, float deltap, hipStream_t st);

// ---- device helpers -----------------------------------------------------------------------------
#ifdef __HIPCC__
#define PART_TYPE(f)       ((f).x & 7u)
#define IS_FLUID(f)        (PART_TYPE(f) == PT_FLUID)
#define IS_BOUNDARY(f)     (PART_TYPE(f) == PT_BOUNDARY)
#define IS_VERTEX(f)       (PART_TYPE(f) == PT_VERTEX)
#define IS_TESTPOINT(f)    (PART_TYPE(f) == PT_TESTPOINT)
#define IS_MOVING(f)       ((f).x & FG_MOVING_BOUNDARY)
#define IS_FLOATING(f)     ((f).x & (FG_MOVING_BOUNDARY | FG_COMPUTE_FORCE))
#define HAS_COMPUTE_FORCE(f) ((f).x & FG_COMPUTE_FORCE)
#define IS_SURFACE(f)      ((f).x & FG_SURFACE)
#define FLUID_NUM(f)       ((f).y >> 12)
#define OBJECT_NUM(f)      ((f).y & 0xfffu)

__device__ __forceinline__ uint32_t info_id(const particleinfo &i) { return (uint32_t)i.z | ((uint32_t)i.w << 16); }
__device__ __forceinline__ bool is_active_w(float w) { return (__float_as_uint(w) & 0x7f800000u) != 0x7f800000u; }

// calcGridHash (src/cuda/cellgrid.cuh:98-104) with run-time linearisation
__device__ __forceinline__ uint32_t grid_hash(const DevParams &p, int gx, int gy, int gz)
{
	return (uint32_t)(gx*p.hs[0] + gy*p.hs[1] + gz*p.hs[2]);
}

// calcGridPosFromCellHash (src/cuda/cellgrid.cuh:115-127)
__device__ __forceinline__ int3 grid_pos_from_hash(const DevParams &p, uint32_t cellHash)
{
	const int q3 = (int)(cellHash / (uint32_t)p.gs12);
	const int rem = (int)cellHash - q3*p.gs12;
	const int q2 = rem / p.gs1;
	const int q1 = rem - q2*p.gs1;
	int3 g;
	g.x = (p.c1 == 0) ? q1 : ((p.c2 == 0) ? q2 : q3);
	g.y = (p.c1 == 1) ? q1 : ((p.c2 == 1) ? q2 : q3);
	g.z = (p.c1 == 2) ? q1 : ((p.c2 == 2) ? q2 : q3);
	return g;
}

// calcGridHashPeriodic (src/cuda/cellgrid.cuh:177-187)
// ---- ENABLE_DEM: terrain height map (src/cuda/geom_core.cu:103-182, src/cuda/forces_kernel.cu:205-226) ------------------------
// tex2D of an unnormalised, clamped, linearly filtered float texture: sample centres at i + 0.5; the fractional weights are kept
// in 1.8 fixed point as the texture unit of the reference's hardware does (CUDA programming guide, "Linear Filtering")
__device__ __forceinline__ float dem_interpol(const DevParams &p, float x, float y)
{
	const float xb = x - 0.5f, yb = y - 0.5f;
	const float fx = floorf(xb), fy = floorf(yb);
	const float al = rintf((xb - fx)*256.0f)*(1.0f/256.0f), be = rintf((yb - fy)*256.0f)*(1.0f/256.0f);
	const int i0 = min(max((int)fx, 0), p.dem_w - 1), i1 = min(max((int)fx + 1, 0), p.dem_w - 1);
	const int j0 = min(max((int)fy, 0), p.dem_h - 1), j1 = min(max((int)fy + 1, 0), p.dem_h - 1);
	const float t00 = p.dem[(size_t)j0*p.dem_w + i0], t10 = p.dem[(size_t)j0*p.dem_w + i1];
	const float t01 = p.dem[(size_t)j1*p.dem_w + i0], t11 = p.dem[(size_t)j1*p.dem_w + i1];
	return (1.0f - al)*(1.0f - be)*t00 + al*(1.0f - be)*t10 + (1.0f - al)*be*t01 + al*be*t11;
}

// DemLJForce: is the particle less than demzmin above the terrain?  Then the terrain acts as its tangent plane there.
// Returns true and the plane (unit normal, grid + local position of its reference point)
__device__ __forceinline__ bool dem_plane(const DevParams &p, const int3 &gridPos, float px, float py, float pz,
	float nrm[3], int pgp[3], float ppos[3])
{
	const float dx = (gridPos.x + 0.5f)*(p.cs[0]/p.ewres) + px/p.ewres + 0.5f;     // DemPos
	const float dy = (gridPos.y + 0.5f)*(p.cs[1]/p.nsres) + py/p.nsres + 0.5f;
	const float globalZ = p.wo_z + (gridPos.z + 0.5f)*p.cs[2] + pz;
	const float z0 = dem_interpol(p, dx, dy);
	if (!(globalZ - z0 < p.demzmin)) return false;
	const float z1 = dem_interpol(p, dx + 1*p.demdx/p.ewres, dy), z2 = dem_interpol(p, dx, dy + 1*p.demdy/p.nsres);
	const float a = p.demdy*(z0 - z1), b = p.demdx*(z0 - z2), c = p.demdx*p.demdy;
	const float inv = 1.0f/sqrtf(a*a + b*b + c*c);       // float3/float multiplies by the reciprocal (src/vector_math.h:526-530)
	nrm[0] = a*inv; nrm[1] = b*inv; nrm[2] = c*inv;
	pgp[0] = gridPos.x; pgp[1] = gridPos.y; pgp[2] = (int)floorf((z0 - p.wo_z)/p.cs[2]);
	ppos[0] = px; ppos[1] = py; ppos[2] = z0 - p.wo_z - (pgp[2] + 0.5f)*p.cs[2];
	return true;
}

__device__ __forceinline__ uint32_t grid_hash_periodic(const DevParams &p, int gx, int gy, int gz)
{
	if (gx < 0) gx = p.gs[0] - 1;
	if (gx >= p.gs[0]) gx = 0;
	if (gy < 0) gy = p.gs[1] - 1;
	if (gy >= p.gs[1]) gy = 0;
	if (gz < 0) gz = p.gs[2] - 1;
	if (gz >= p.gs[2]) gz = 0;
	return grid_hash(p, gx, gy, gz);
}

// window cell (row r of 16, column col) of a tile -> its cell hash, or 0xFFFFFFFF outside the tile's window / the grid
__device__ __forceinline__ uint32_t window_cell_hash(const DevParams &p, int g2, int g3, int ca, int ncells, int r, int col)
{
	if (col >= ncells + 2) return 0xFFFFFFFFu;
	const int v0 = ca - 1 + col, v1 = g2 + (r & 3) - 1, v2 = g3 + (r >> 2) - 1;
	int gx = (p.c1 == 0) ? v0 : (p.c2 == 0) ? v1 : v2;
	int gy = (p.c1 == 1) ? v0 : (p.c2 == 1) ? v1 : v2;
	int gz = (p.c1 == 2) ? v0 : (p.c2 == 2) ? v1 : v2;
	if (gx < 0) { if (p.periodic & SPHX_PERIODIC_X) gx = p.gs[0] - 1; else return 0xFFFFFFFFu; }
	else if (gx >= p.gs[0]) { if (p.periodic & SPHX_PERIODIC_X) gx = 0; else return 0xFFFFFFFFu; }
	if (gy < 0) { if (p.periodic & SPHX_PERIODIC_Y) gy = p.gs[1] - 1; else return 0xFFFFFFFFu; }
	else if (gy >= p.gs[1]) { if (p.periodic & SPHX_PERIODIC_Y) gy = 0; else return 0xFFFFFFFFu; }
	if (gz < 0) { if (p.periodic & SPHX_PERIODIC_Z) gz = p.gs[2] - 1; else return 0xFFFFFFFFu; }
	else if (gz >= p.gs[2]) { if (p.periodic & SPHX_PERIODIC_Z) gz = 0; else return 0xFFFFFFFFu; }
	return grid_hash(p, gx, gy, gz);
}

// The tiled forces kernel keeps its window in ONE frame per tile: a record of window cell (row r, column col) is stored as
// its cell-local position + tile_shift(r, col), the offset of that cell's centre from the centre of the window; home
// particles are shifted the same way, so that a pair needs no per-cell shift.  The same function is used for both sides.
__device__ __forceinline__ float3 tile_shift(const DevParams &p, int ncells, int r, int col)
{
	const float a1 = ((float)col - 0.5f*(float)(ncells + 1))*p.csc1;
	const float a2 = ((float)(r & 3) - 1.5f)*p.csc2;
	const float a3 = ((float)(r >> 2) - 1.5f)*p.csc3;
	float3 s;
	s.x = (p.c1 == 0) ? a1 : (p.c2 == 0) ? a2 : a3;
	s.y = (p.c1 == 1) ? a1 : (p.c2 == 1) ? a2 : a3;
	s.z = (p.c1 == 2) ? a1 : (p.c2 == 2) ? a2 : a3;
	return s;
}

// hash of the cell at COORD1 = 0 of window row r (grid row g2-1+(r&3), g3-1+(r>>2), wrapped where periodic): the COORD1
// index of a record of that row is its cell hash minus this.  -1 for a row outside the grid
__device__ __forceinline__ int window_row_hash0(const DevParams &p, int g2, int g3, int r)
{
	const int gs2 = p.gsc2, gs3 = p.gsc3;
	int v1 = g2 + (r & 3) - 1, v2 = g3 + (r >> 2) - 1;
	if (v1 < 0) { if (p.periodic & (1u << p.c2)) v1 = gs2 - 1; else return -1; }
	else if (v1 >= gs2) { if (p.periodic & (1u << p.c2)) v1 = 0; else return -1; }
	if (v2 < 0) { if (p.periodic & (1u << p.c3)) v2 = gs3 - 1; else return -1; }
	else if (v2 >= gs3) { if (p.periodic & (1u << p.c3)) v2 = 0; else return -1; }
	return v1*p.gs1 + v2*p.gs12;
}

// ... -> start/count of its particles
__device__ __forceinline__ void window_cell(const DevParams &p, const uint32_t *__restrict__ cellStart,
	const uint32_t *__restrict__ cellEnd, int g2, int g3, int ca, int ncells, int r, int col,
	uint32_t &start, uint32_t &cnt)
{
	start = 0; cnt = 0;
	const uint32_t h = window_cell_hash(p, g2, g3, ca, ncells, r, col);
	if (h == 0xFFFFFFFFu) return;
	const uint32_t cs = cellStart[h];
	if (cs != CELL_EMPTY) { start = cs; cnt = cellEnd[h] - cs; }
}


typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

// async global -> LDS copy of `count` float4 records (LDS-DMA, no VGPR round trip): every wave copies
// 64-record chunks, LDS destination = wave-uniform base + lane*16 (cdna_hip_programming.md "global_load_lds")
template<int NT>
__device__ __forceinline__ void stage_rows(const float4 *__restrict__ src, float4 *dst, uint32_t count, uint32_t tid)
{
	const uint32_t wave = tid >> 6, lane = tid & 63u;
	for (uint32_t c0 = wave*64u; c0 < count; c0 += NT) {
		const uint32_t cu = __builtin_amdgcn_readfirstlane(c0);
		if (cu + lane < count)
			__builtin_amdgcn_global_load_lds((gptr_t)(src + cu + lane), (lptr_t)(dst + cu), 16, 0, 0);
	}
}

// one wave copies a whole row: 64-record chunks
__device__ __forceinline__ void stage_row_wave(const float4 *__restrict__ src, float4 *dst, uint32_t count, uint32_t lane)
{
	for (uint32_t c0 = 0; c0 < count; c0 += 64u)
		if (c0 + lane < count)
			__builtin_amdgcn_global_load_lds((gptr_t)(src + c0 + lane), (lptr_t)(dst + c0), 16, 0, 0);
}

#endif // __HIPCC__

#endif
