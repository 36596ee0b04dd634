"""Per-workgroup phase times of the tiled forces kernel (SPHX_TILE_DEBUG=16); timing experiment only."""
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
buf = np.zeros((1024, 8), dtype=np.uint64)
f = eng.lib.sphx_dbg_tile_profile
f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]; f.restype = C.c_int
g = f(eng.ctx.handle, buf.ctypes.data, 1024)
t = buf[:g].astype(np.float64) * 1e-5   # 100 MHz ticks -> ms
print("groups", g)
for k, name in enumerate(["total", "stage", "pairs(wave0)", "tail", "tail:drain", "tail:finalize", "tail:cfl+barrier"]):
    print("%-13s mean %.3f  min %.3f  max %.3f ms" % (name, t[:, k].mean(), t[:, k].min(), t[:, k].max()))
x = t[:, 0].reshape(-1, 8)
print("per-XCD mean total:", np.round(x.mean(axis=0), 3))
