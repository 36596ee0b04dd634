"""gpusph_amd -- MI355X-native WCSPH timestep engine behind GPUSPH's engine interfaces.

Only what the per-step hot path needs lives here:
  csrc/      hand-written HIP kernels for gfx950 + the C ABI (include/sphx.h) -> libsphx.so
  host/      C++ adapters implementing GPUSPH's Abstract{Neibs,Forces,Visc,Integration}Engine
  capi.py    ctypes binding of the C ABI (fails loudly when libsphx.so is missing)
  params.py  SimParams/PhysParams mirror -> sphx_params
  problem.py Problem-style set-up of the synthetic DamBreak3D box (host side, numpy)
  engine.py  one-GPU timestep driver mirroring the Integrator/GPUWorker command sequence
  multigpu.py slab split + halo exchange over torch.distributed (RCCL on ROCm)
"""
from .defs import *  # noqa: F401,F403
