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
int sphx_sa_wall_integrate_gamma(sphx_ctx *, const SaIntGammaArgs &, hipStream_t) { return SPHX_OK; }
