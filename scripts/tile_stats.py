"""Distribution of tile sizes (home particles, window records, cells) from the real tile descriptors."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpusph_amd.problem import DamBreak3D
from gpusph_amd.engine import TimestepEngine
n = float(sys.argv[1]) if len(sys.argv) > 1 else 32e6
prob = DamBreak3D(DamBreak3D.deltap_for(n))
eng = TimestepEngine(prob, track_particle_count=False)
eng.build_neibs(); torch.cuda.synchronize()
f = eng.lib.sphx_dbg_tiles
f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]; f.restype = C.c_int
maxT = 400000
t = np.zeros((maxT, 16), dtype=np.uint32)
nt = f(eng.ctx.handle, t.ctypes.data, maxT)
t = t[:nt]
hc = t[:, 8:12].sum(axis=1)
act = (t[:, 13] & 1) != 0
print("tiles", nt, "with fluid in window", act.sum(), "particles", eng.n)
for name, v in (("home particles", hc[act]), ("window records", t[act, 12]), ("cells along COORD1", t[act, 3])):
    print("%-20s mean %.1f  p10 %d  p50 %d  p90 %d  max %d" % (name, v.mean(), *np.percentile(v, [10, 50, 90]).astype(int), v.max()))
waves = (hc[act] + 63) // 64
print("lane fill of allocated waves: %.3f   mean waves/tile %.2f" % (hc[act].sum() / (waves * 64).sum(), waves.mean()))
lim_p = (hc[act] > 512 - 70).mean(); lim_w = (t[act, 12] > 3200 - 350).mean(); lim_c = (t[act, 3] >= 14).mean()
print("closed by: particles~%.2f window~%.2f cells~%.2f" % (lim_p, lim_w, lim_c))
