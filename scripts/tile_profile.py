"""Per-wave phase times of the tiled forces kernel (SPHX_TILE_DEBUG=16); timing experiment only.
usage: tile_profile.py [particles] ; SPHX_TILE_DEBUG adds experiment bits (see ForcesArgs::dbg)"""
import os, sys, ctypes as C
os.environ["SPHX_TILE_DEBUG"] = str(16 | int(os.environ.get("SPHX_TILE_DEBUG", "0")))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpusph_amd.problem import DamBreak3D
from gpusph_amd.engine import TimestepEngine

n = float(sys.argv[1]) if len(sys.argv) > 1 else 32e6
prob = DamBreak3D(DamBreak3D.deltap_for(n))
eng = TimestepEngine(prob, track_particle_count=False)
for _ in range(12):
    eng.step()
torch.cuda.synchronize()
buf = np.zeros((1024, 8, 10), dtype=np.uint64)
f = eng.lib.sphx_dbg_tile_profile
f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]; f.restype = C.c_int
g = f(eng.ctx.handle, buf.ctypes.data, 1024)
t = buf[:g].astype(np.float64) * 1e-5   # 100 MHz ticks -> ms
t[:, :, 9] *= 1e5
names = ["total", "top barrier", "dma issue", "landing", "conversion", "window barrier", "setup+requests", "pair loop", "drain+finalize", "tiles"]
print("SPHX_TILE_DEBUG", os.environ["SPHX_TILE_DEBUG"], "groups", g, " (ms per launch; mean over workgroups, per wave 0..7)")
for k, name in enumerate(names):
    print("%-15s all %.3f | " % (name, t[:, :, k].mean()) + " ".join("%.3f" % t[:, w, k].mean() for w in range(8)))
print("per-XCD mean total:", np.round(t[:, 0, 0].reshape(-1, 8).mean(axis=0), 3))
