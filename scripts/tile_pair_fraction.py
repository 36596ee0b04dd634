"""What share of the stored pairs of the tiled forces kernel has BOTH ends among the home particles of one tile (the pairs a
symmetric evaluation could compute once), and how much of the list stream is padding.  Read from the tile structures of a
built context (lane records = the home particles' window slots, list stream = window slots of the neighbours).
usage: tile_pair_fraction.py [particles]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpusph_amd.problem import DamBreak3D
from gpusph_amd.engine import TimestepEngine

n = float(sys.argv[1]) if len(sys.argv) > 1 else 2e6
prob = DamBreak3D(DamBreak3D.deltap_for(n), obstacle=True, linearization="xzy")
eng = TimestepEngine(prob, track_particle_count=False)
eng.step()
torch.cuda.synchronize()
lib, h = eng.lib, eng.ctx.handle
lib.sphx_dbg_tile_table.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
def table(which, shape, dtype):
    a = np.zeros(shape, dtype=dtype)
    assert lib.sphx_dbg_tile_table(h, which, a.ctypes.data, a.nbytes) == 0
    return a
ctl = table(5, 16, np.uint32)
nt, nbatch, nlanes = int(ctl[0]), int(ctl[12]), int(ctl[13])
assert lib.sphx_dbg_tiles_usable(h) == 1
runs = table(1, (nt, 48), np.uint32)
desc = table(0, (nt, 16), np.uint32)
rec = table(2, nlanes, np.uint32)
stream = table(4, (nbatch, 64, 4), np.uint16)          # [batch][lane][entry]: byte offset of the neighbour's window row
chunks = runs[:, 8] & 255
stored = inside = slots = 0
home_tot = 0
for t in range(nt):
    c = int(chunks[t])
    if c == 0:
        continue
    lb, nb = int(desc[t, 14]), int((runs[t, :8] >> 22).sum())
    r = rec[int(desc[t, 15]):int(desc[t, 15]) + 64*c]
    valid = (r >> 16) != 0
    home = np.zeros(4096, dtype=bool)
    home[(r[valid] & 0xFFFF) >> 4] = True
    home_tot += int(valid.sum())
    e = stream[lb:lb + nb].reshape(-1)
    real = e != 0
    slots += e.size
    stored += int(real.sum())
    inside += int(home[e[real] >> 4].sum())
print("particles %d, tiles %d, home particles per tile %.1f" % (eng.n, nt, home_tot/max(nt, 1)))
print("list stream: %d entry slots, %d stored pairs: %.1f %% padding" % (slots, stored, 100.0*(slots - stored)/slots))
print("stored pairs with both ends in the same tile: %.3f" % (inside/stored))
