"""Per-wave phase times of the tiled forces kernel; needs a library built with make EXTRA=-DSPHX_TILE_DEBUG_BUILD.
usage: tile_profile.py [particles]"""
import os, sys, ctypes as C
os.environ["SPHX_TILE_DEBUG"] = "16"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpusph_amd.problem import DamBreak3D
from gpusph_amd.engine import TimestepEngine
n = float(sys.argv[1]) if len(sys.argv) > 1 else 32e6
prob = DamBreak3D(DamBreak3D.deltap_for(n), obstacle=True, linearization="xzy")
eng = TimestepEngine(prob, track_particle_count=False)
for _ in range(12):
    eng.step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(4):
    eng.step()
e1.record(); torch.cuda.synchronize()
print("ms/step with timers on: %.3f" % (e0.elapsed_time(e1)/4))
f = eng.lib.sphx_dbg_tile_profile
f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]; f.restype = C.c_int
g = 256
buf = np.zeros((g, 8, 10), dtype=np.uint64)
got = f(eng.ctx.handle, buf.ctypes.data, g)
buf = buf[:got].astype(np.float64)
names = ["total", "top barrier", "desc + DMA issue", "finalize prev", "DMA landing", "conversion", "window barrier", "requests", "pair phase", "tiles"]
tot = buf[:, :, 0].mean()
print("groups", got, " cycles per launch (mean over workgroups and waves); share of the wave's total")
for k, nm in enumerate(names):
    v = buf[:, :, k]
    print("%-18s mean %12.0f  (%.1f %%)   per wave: %s" % (nm, v.mean(), 100*v.mean()/tot if k else 100.0, " ".join("%.0f" % x for x in v.mean(axis=0))))
