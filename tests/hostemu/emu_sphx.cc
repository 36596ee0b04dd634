// TEST HARNESS ONLY: the host-side API file and the open-boundary kernels of libsphx, compiled for the host through the stand-in
// tests/hostemu/hip/hip_runtime.h (read its header).  Built by tests/test_sa_io_hostemu.py into tests/hostemu/_build/.
#include <hip/hip_runtime.h>
thread_local dim3 blockIdx, threadIdx, blockDim, gridDim;
#include "../../gpusph_amd/csrc/sphx_api.hip"
#define ROW_GRID 2      // the row kernels stride over their index: two blocks of fibres do as well as a thousand
#include "../../gpusph_amd/csrc/sa_io.hip"
// sa_bounds.hip (the solid-wall SA passes: verified on the GPU, emulated for the regression value and because an open-boundary
// run calls some of them) with its launches rewritten into _build/sa_bounds_emu.inc by tests/hostemu_lib.py.  The tiled window and
// the wall-particle kernels it can hand over to live in other files: absent here, so every pass is its list walker.
#include "sa_bounds_emu.inc"
// euler.hip (the Euler step and, with moving SA bodies, the normals of the new state), launches rewritten the same way
#include "euler_emu.inc"
int sphx_sa_tiles_run(sphx_ctx *, int, void *, const void *, const void *, const void *, const void *, const uint32_t *, const uint32_t *,
	const uint16_t *, const void *, uint32_t, uint32_t, uint32_t, float, hipStream_t, bool *used, const uint32_t **guard)
{ if (used) *used = false; if (guard) *guard = nullptr; return SPHX_OK; }
int sphx_sa_wall_forces(sphx_ctx *, const SaForcesArgs &, hipStream_t) { return SPHX_OK; }
int sphx_sa_wall_density_sum(sphx_ctx *, const SaDensitySumArgs &, hipStream_t) { return SPHX_OK; }
int sphx_sa_wall_density_sum_moving(sphx_ctx *, const SaDensitySumArgs &, hipStream_t) { return SPHX_OK; }
int sphx_sa_wall_density_diffusion_open(sphx_ctx *, const SaDiffusionArgs &, hipStream_t) { return SPHX_OK; }
int sphx_sa_wall_integrate_gamma(sphx_ctx *, const SaIntGammaArgs &, hipStream_t) { return SPHX_OK; }

// the list build (neibs_build.hip: build_neibs_kernel with the prepass on the matrix cores, the v_mfma / v_permlane32_swap /
// v_alignbit of which the stand-in header emulates by their documented lane layouts), behind a test-only entry point: the caller
// brings the cell tables of a sorted state and gets the list, the section lengths and the folded counters
#include "../../gpusph_amd/csrc/neibs_build.hip"
extern "C" int emu_neibs_list(sphx_ctx *ctx, uint16_t *neibsList, const void *pos, const void *info, const uint32_t *hash,
	const uint32_t *cellStart, const uint32_t *cellEnd, const uint32_t *cellFluidEnd, uint32_t numParticles, uint32_t particleRangeEnd,
	float sqinfluenceradius, uint32_t *neibCounts, int *maxNeibs, unsigned long long *interactions)
{
	std::vector<char> cnt(sizeof(NeibsCounters) + NEIBS_SPREAD*sizeof(NeibsSpread), 0);
	NeibsCounters *c = reinterpret_cast<NeibsCounters*>(cnt.data());
	c->hasTooManyNeibs = -1;
	ctx->cell_fluid_end = const_cast<uint32_t*>(cellFluidEnd);
	ctx->counters_dev = c;
	ctx->neib_counts = neibCounts;
	const int rc = sphx_neibs_list_launch(ctx, neibsList, pos, info, hash, cellStart, cellEnd, nullptr, nullptr, nullptr, nullptr, nullptr,
		numParticles, particleRangeEnd, sqinfluenceradius, 0.0f, nullptr);
	const NeibsSpread *part = reinterpret_cast<const NeibsSpread*>(c + 1);
	int mx = 0; unsigned long long tot = 0;
	for (unsigned k = 0; k < NEIBS_SPREAD; ++k) { mx = part[k].maxFluidBoundaryNeibs > mx ? part[k].maxFluidBoundaryNeibs : mx; tot += part[k].numInteractions; }
	*maxNeibs = mx; *interactions = tot;
	ctx->cell_fluid_end = nullptr; ctx->counters_dev = nullptr; ctx->neib_counts = nullptr;
	return rc;
}
