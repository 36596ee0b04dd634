"""Diagnostics of the tiled forces kernel against the generic one on the same state: who differs, and where they sit in the
tiling (chunk, lane, runs).  usage: diag_tiles_v2.py [particles] [steps]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpusph_amd.problem import DamBreak3D
from gpusph_amd.engine import TimestepEngine

n = float(sys.argv[1]) if len(sys.argv) > 1 else 3e5
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
prob = DamBreak3D(DamBreak3D.deltap_for(n), obstacle=True, linearization="xzy")
eng = TimestepEngine(prob, track_particle_count=False, clobber_neibslist=True)
for _ in range(steps):
    eng.step()
torch.cuda.synchronize()
f_t = eng.forces[:eng.n].cpu().numpy().copy()
lib, h = eng.lib, eng.ctx.handle
lib.sphx_dbg_tile_table.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
ctl = np.zeros(16, dtype=np.uint32); lib.sphx_dbg_tile_table(h, 5, ctl.ctypes.data, ctl.nbytes)
print("tiles", ctl[0], "overflow", ctl[1], "batches", ctl[12], "lanes", ctl[13], "usable", lib.sphx_dbg_tiles_usable(h))
nt = int(ctl[0])
runs = np.zeros((nt, 48), dtype=np.uint32); lib.sphx_dbg_tile_table(h, 1, runs.ctypes.data, runs.nbytes)
desc = np.zeros((nt, 16), dtype=np.uint32); lib.sphx_dbg_tile_table(h, 0, desc.ctypes.data, desc.nbytes)
C_ = runs[:, 8] & 255; P_ = (runs[:, 8] >> 8) & 0xFFFF; R_ = runs[:, 8] >> 24
print("chunks/tile mean %.2f  particles/tile mean %.1f max %d  runs/tile mean %.2f max %d" % (C_.mean(), P_.mean(), P_.max(), R_.mean(), R_.max()))
T = np.array([(runs[t, :8] >> 22).sum() for t in range(nt)])
print("batches/tile mean %.1f; sum %d (cursor %d)" % (T.mean(), T.sum(), ctl[12]))
# the same state through the generic kernel: a second engine on a copy of the state
os.environ["SPHX_DISABLE_TILES"] = "1"
eng2 = TimestepEngine(prob, track_particle_count=False, clobber_neibslist=True)
for _ in range(steps):
    eng2.step()
torch.cuda.synchronize()
f_g = eng2.forces[:eng2.n].cpu().numpy()
sc = np.abs(f_g).max(axis=0)
err = np.abs(f_t - f_g) / sc
bad = np.where(err.max(axis=1) > 1e-4)[0]
print("n", eng.n, "max rel err per component", err.max(axis=0), "bad particles", len(bad))
if len(bad):
    li = np.zeros(int(ctl[13]), dtype=np.uint32); lib.sphx_dbg_tile_table(h, 3, li.ctypes.data, li.nbytes)
    pos_of = {}
    for t in range(nt):
        lb, c = int(desc[t, 15]), int(C_[t])
        for L, idx in enumerate(li[lb:lb + 64*c]):
            if idx != 0xFFFFFFFF:
                pos_of[int(idx)] = (t, L)
    ptype = (eng.info[:eng.n, 0].cpu().numpy().astype(np.int32) & 7)
    for i in bad[:40]:
        t, L = pos_of.get(int(i), (-1, -1))
        c = L >> 6
        tab = runs[t, 9 + c] if t >= 0 else 0
        print("particle %d type %d tile %d lane %d chunk %d runs(first %d, n %d) of tile C %d P %d  err %s  tiled %s generic %s"
              % (i, ptype[i], t, L, c, tab & 255, (tab >> 8) & 255, C_[t], P_[t], err[i], f_t[i], f_g[i]))
    chunks = np.array([pos_of.get(int(i), (-1, -1))[1] >> 6 for i in bad])
    print("bad by chunk:", np.bincount(chunks[chunks >= 0]))
    notfound = [int(i) for i in bad if int(i) not in pos_of]
    print("bad particles not in any lane table:", len(notfound))
