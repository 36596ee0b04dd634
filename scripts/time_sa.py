"""Step time of an SA_BOUNDARY tank (SABox mirror) on the GPU, per kernel.
usage: python scripts/time_sa.py [deltap] [StillWaterSA|StillWaterRepackSA] [steps]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gpusph_amd.problem import SABox
from gpusph_amd.engine import TimestepEngine

dp = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0125
options = sys.argv[2] if len(sys.argv) > 2 else "StillWaterSA"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
t0 = time.time()
prob = SABox(dp, l=1.6, w=1.6, h=1.0, H=0.8, options=options)
prob.simparams.buildneibsfreq = 10
print("SABox %s dp=%g: %d particles (%d fluid, %d segments, %d vertices), mesh in %.1f s" % (
    options, dp, prob.num_particles, prob.num_fluid, prob.num_segments, prob.num_vertices, time.time() - t0))
eng = TimestepEngine(prob, device="cuda:0")
eng.run(11)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
eng.run(steps)
ev[1].record(); torch.cuda.synchronize()
ms = ev[0].elapsed_time(ev[1]) / steps
info = eng.neibs_info()
print("%.3f ms/step, %.1f M particle-updates/s; max neighbours fluid+boundary %d, vertex %d; dt %.3g, t %.4g" % (
    ms, prob.num_particles / ms / 1e3, info.maxFluidBoundaryNeibs, info.maxVertexNeibs, eng.current_dt(), eng.time()))
v = eng.vel[:eng.n].cpu().numpy()
print("max |v| %.3g (c0 %.3g), finite %s" % (np.abs(v[:, :3]).max(), prob.physparams.sscoeff[0], np.isfinite(v).all()))
