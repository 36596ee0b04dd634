// TEST HARNESS ONLY: the host-side API file and the open-boundary kernels of libsphx, compiled for the host through the stand-in
// tests/hostemu/hip/hip_runtime.h (read its header).  Built by tests/test_sa_io_hostemu.py into tests/hostemu/_build/.
#include <hip/hip_runtime.h>
thread_local dim3 blockIdx, threadIdx, blockDim, gridDim;
#include "../../gpusph_amd/csrc/sphx_api.hip"
#include "../../gpusph_amd/csrc/sa_io.hip"
