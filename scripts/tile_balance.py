"""Lane/wave/tile utilisation of the tiled forces kernel from the real neighbour lists and tile descriptors,
and what re-ordering particles inside a tile by list length would give.  Timing experiment only."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpusph_amd.problem import DamBreak3D
from gpusph_amd.engine import TimestepEngine

n = float(sys.argv[1]) if len(sys.argv) > 1 else 8e6
prob = DamBreak3D(DamBreak3D.deltap_for(n))
eng = TimestepEngine(prob, track_particle_count=False, clobber_neibslist=True)
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 11):
    eng.step()
torch.cuda.synchronize()
A, N = eng.alloc, eng.n
nl = eng.neibslist.view(128, A)[:, :N]
end = nl == -1
nf = torch.argmax(end.to(torch.int8), dim=0)                         # first terminator from the front
nb = torch.argmax(torch.flip(end, dims=[0]).to(torch.int8), dim=0)   # ... from the back
ptype = (eng.info[:N, 0].to(torch.int32) & 7)
fluid = ptype == 0
w = torch.where(fluid, nf + nb, nf).cpu().numpy().astype(np.int64)    # entries walked (DYN boundary: fluid section only)
isf = fluid.cpu().numpy()
f = eng.lib.sphx_dbg_tiles
f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]; f.restype = C.c_int
maxT = 400000
tiles = np.zeros((maxT, 16), dtype=np.uint32)
nt = f(eng.ctx.handle, tiles.ctypes.data, maxT)
tiles = tiles[:nt]
print("particles", N, "tiles", nt, "mean walked entries", w.mean())

def steps(x):    # ring half-steps of 2 entries a wave needs for its longest lane (+1 entry for the terminator)
    return (x + 1 + 1) // 2

rng = np.random.default_rng(0)
sel = rng.choice(nt, size=min(nt, 20000), replace=False)
tot = dict(pairs=0, cur_wave=0, cur_tile=0, sort_wave=0, sort_tile=0, sort_bal=0, g16_wave=0, g16_bal=0, fb_wave=0)
for t in sel:
    d = tiles[t]
    idx = np.concatenate([np.arange(d[4 + r], d[4 + r] + d[8 + r]) for r in range(4)])
    if not (d[13] & 1) or len(idx) == 0:
        continue
    ww = w[idx]
    tot["pairs"] += ww.sum()
    def waves(order):
        x = ww[order]
        pad = (-len(x)) % 64
        x = np.concatenate([x, np.zeros(pad, dtype=x.dtype)]).reshape(-1, 64)
        return steps(x.max(axis=1)) * 2 * 64          # slots per wave
    def tile_time(ws, balanced):
        k = np.zeros(8, dtype=np.int64); k[:len(ws)] = ws
        if balanced:    # chunks sorted descending: SIMD s gets chunks s and 7-s
            simd = k[:4] + k[7:3:-1]
        else:
            simd = k[:4] + k[4:]
        return simd.max() * 4
    cur = waves(np.arange(len(ww)))
    tot["cur_wave"] += cur.sum(); tot["cur_tile"] += tile_time(cur, False)
    so = np.argsort(-ww, kind="stable")
    sw = waves(so)
    tot["sort_wave"] += sw.sum(); tot["sort_tile"] += tile_time(sw, False); tot["sort_bal"] += tile_time(sw, True)
    # groups of 16 consecutive particles sorted by their max
    pad = (-len(ww)) % 16
    g = np.concatenate([ww, np.zeros(pad, dtype=ww.dtype)]).reshape(-1, 16)
    go = np.argsort(-g.max(axis=1), kind="stable")
    gw = waves(np.arange(len(g) * 16).reshape(-1, 16)[go].ravel()[: len(g) * 16] % (len(ww) + pad) if False else np.arange(0))  if False else None
    x = g[go].ravel()
    pad2 = (-len(x)) % 64
    x = np.concatenate([x, np.zeros(pad2, dtype=x.dtype)]).reshape(-1, 64)
    gws = steps(x.max(axis=1)) * 2 * 64
    tot["g16_wave"] += gws.sum(); tot["g16_bal"] += tile_time(gws, True)
    # fluid first, then boundary (stable): separates the two populations only
    fo = np.argsort(~isf[idx], kind="stable")
    tot["fb_wave"] += waves(fo).sum()
p = tot["pairs"]
for k in ["cur_wave", "cur_tile", "sort_wave", "sort_tile", "sort_bal", "g16_wave", "g16_bal", "fb_wave"]:
    print("%-10s utilisation %.3f" % (k, p / tot[k]))
