// sa_args.h -- argument blocks of the SA_BOUNDARY engines that two translation units share: sa_bounds.hip (everything, in the
// reference's operation order: EXACT compile flags) and sa_wall.hip (the boundary-element terms with one element per lane).
#pragma once
#include "sphx_internal.h"

// |grad gamma_as| per (wall particle, boundary-section entry) handed from the density summation / gamma quadrature of a step to
// the forces pass that follows at the same positions (sphx_ctx::sa_wall_cache); all zero = not in use
struct SaWallCache {
	float *values;        // [wall particle][SA_WALL_CACHE_ENTRIES]
	float4 *tag;          // per wall particle: position bits of the particle when its row was written, generation of the list
	uint32_t capacity, gen;
	// a run with open boundaries: sum_s grad gamma_as of the particle's elements as the forces pass of step n evaluated it, for the
	// density summations of the step (both start from the positions of step n): [2 w] = {sum, generation}, [2 w + 1] = position bits
	float4 *gsum;
};

// open boundaries of a run with ENABLE_INLET_OUTLET: flags of particleinfo.x (src/particleinfo.h:153-156, 222-241)
#define SA_FG_INLET            (PART_FLAG_START << 2)
#define SA_FG_OUTLET           (PART_FLAG_START << 3)
#define SA_FG_VELOCITY_DRIVEN  (PART_FLAG_START << 4)
#define SA_FG_CORNER           (PART_FLAG_START << 5)
#define SA_IS_OPEN(f)          (((f).x & (SA_FG_INLET | SA_FG_OUTLET)) != 0)
#define SA_IS_VELOCITY_DRIVEN(f) (((f).x & SA_FG_VELOCITY_DRIVEN) != 0)
#define SA_IS_CORNER(f)        (((f).x & SA_FG_CORNER) != 0)

// the two boundary-condition passes: see sa_segment_bc_kernel / sa_vertex_bc_kernel (sa_bounds.hip)
struct SaArgs {
	float4 *vel;                 // in place: boundary rows (segment kernel) / vertex rows (vertex kernel) are written
	float4 *gGam;                // in place: boundary rows written when gamma is (re)computed
	float4 *boundElement;        // vertex-normal kernel: vertex rows written
	const float4 *pos;
	const uint4 *vertices;
	const particleinfo *info;
	const uint32_t *hash, *cellStart;
	const neibdata *neibsList;
	uint32_t numParticles;
	int step, repack;
	// KEPSILON (sa_segment_bc_params / sa_vertex_bc_params with has_keps, src/cuda/sa_bc_params.h:152-200,275-320): in place
	float *tke, *eps;
	float4 *eulerVel;
	float deltap;
	int openFaces;                // a run with open boundaries: their segments and (non-corner) vertices are left to sa_io.hip
	// [0] = number of, [1..] = the particles of the pass's type (sphx_ctx::sa_rows_bound / sa_rows_vert): thread t takes row t instead of
	// particle t, so that the lanes of a wave all have a list to walk (boundary elements and vertices are a few per cent of the particles,
	// a few lanes of every wave next to a wall); NULL: one thread per particle
	const uint32_t *rows;
};
// the solid-wall rows of the two passes in a run with open boundaries (sa_io.hip launches its own kernels for the open faces)
int sphx_sa_solid_rows_launch(sphx_ctx *ctx, const SaArgs &a, bool vertexPass, hipStream_t st);

// forces with SA_BOUNDARY: see sa_forces_kernel (sa_bounds.hip)
struct SaForcesArgs {
	float4 *forces;
	float *cfl;
	float *cflGamma, *cflGammaBlocks;   // BUFFER_CFL_GAMMA: per particle, and per block behind round_up(numParticles, 4)
	const float4 *pos, *vel, *gGam, *boundElement;
	const float2 *vertPos[3];
	const particleinfo *info;
	const uint32_t *hash, *cellStart;
	const neibdata *neibsList;
	uint32_t fromParticle, toParticle, cflOffset;
	float deltap;
	// KEPSILON (keps_forces_params, src/cuda/forces_params.h:283-320)
	const float *tke, *eps, *turbvisc;
	const float4 *eulerVel;
	float *dkde;          // BUFFER_DKDE: 3 floats per particle (diffusion term of k, of epsilon, Yap's C_e2)
	float *cflKeps;       // BUFFER_CFL_KEPS: one per block
	float epsilon;
	// the tiled kernel (forces.hip, SPHX_TURB_SA) has left the fluid <- fluid and fluid <- vertex sums in FORCES: only the
	// boundary elements and the fix-ups remain -- unless the tiling overflowed (*tileGuard != 0: the tiled kernel did nothing)
	int tiled;
	const uint32_t *tileGuard;
	int wallDone;         // ... and sa_forces_wall_kernel has added the boundary elements (needs `tiled`)
	SaWallCache wc;
	int open;             // a run with open boundaries (sa_forces_kernel<., true>): sa_forces_wall_kernel adds the OPEN terms as well
};

// integrateGammaDevice, quadrature flavour: see sa_integrate_gamma_kernel (sa_bounds.hip)
struct SaIntGammaArgs {
	float4 *newGGam;
	const float4 *oldGGam, *pos, *boundElement;
	const float2 *vertPos[3];
	const particleinfo *info;
	const uint32_t *hash, *cellStart;
	const neibdata *neibsList;
	uint32_t numParticles;
	float epsilon;
	int wallDone;      // the fluid particles with boundary elements in reach are done by sa_integrate_gamma_wall_kernel
	SaWallCache wc;
	int vertexRows;    // ENABLE_MOVING_BODIES: gamma of the vertex particles is integrated too
};

// density summation with dynamic gamma: see sa_density_sum_kernel (sa_bounds.hip)
struct SaDensitySumArgs {
	float4 *newVel, *newGGam, *forces;
	const float4 *oldPos, *pos /* new positions: what the walker prefetches is not used */, *oldVel, *oldGGam, *boundElement;
	const float2 *vertPos[3];
	const particleinfo *info;
	const uint32_t *hash, *cellStart;
	const neibdata *neibsList;
	uint32_t numParticles;
	int tiled;                    // the volumic sums are in FORCES.w already (tiled kernel, SPHX_TURB_SA_DSUM), see SaForcesArgs
	const uint32_t *tileGuard;
	int wallDone;                 // sa_density_sum_wall_kernel has left {sum grad gamma, sum grad gamma . dr} in newGGam (needs `tiled`)
	SaWallCache wc;
	const float4 *oldEulerVel;    // a run with open boundaries (sa_density_sum_kernel<true>)
	float dt;
	uint32_t *openList;           // ... sa_density_sum_wall_kernel<true> appends the particles with an open segment in reach ([0] = count)
	const float4 *boundElementNew; // ENABLE_MOVING_BODIES (sa_density_sum_kernel<.., true>): BUFFER_BOUNDELEMENTS of the new state
};

// Brezzi density diffusion with SA_BOUNDARY: see sa_density_diffusion_kernel (sa_bounds.hip)
struct SaDiffusionArgs {
	float4 *forces;
	const float4 *pos, *vel, *gGam;
	const particleinfo *info;
	const uint32_t *hash, *cellStart;
	const neibdata *neibsList;
	uint32_t numParticles;
	float dt;
	const uint32_t *tileGuard;    // stand-by launch behind the tiled kernel (SPHX_TURB_SA_DIFF): only if the tiling overflowed
	// a run with open boundaries (sa_density_diffusion_kernel<true>): the segments of the pressure-driven faces take part
	const float4 *boundElement;
	const float2 *vertPos[3];
	float deltap;
};

// sa_wall.hip: the boundary-element terms of the three engines for the particles of ctx->sa_wall
int sphx_sa_wall_forces(sphx_ctx *ctx, const SaForcesArgs &a, hipStream_t st);
int sphx_sa_wall_density_sum(sphx_ctx *ctx, const SaDensitySumArgs &a, hipStream_t st);
int sphx_sa_wall_density_sum_moving(sphx_ctx *ctx, const SaDensitySumArgs &a, hipStream_t st);   // ... with ENABLE_MOVING_BODIES (fluid rows)
int sphx_sa_wall_integrate_gamma(sphx_ctx *ctx, const SaIntGammaArgs &a, hipStream_t st);
int sphx_sa_wall_density_diffusion_open(sphx_ctx *ctx, const SaDiffusionArgs &a, hipStream_t st);   // the open segments' part, added to FORCES.w
