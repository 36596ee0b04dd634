"""How much of the tiled kernel's pair loop is padding: every wave walks until its LONGEST list ends (uniform control),
in steps of 2 entries (TILE_HB).  From the real tile descriptors and the real neighbour list."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpusph_amd import defs as D
from gpusph_amd.problem import DamBreak3D
from gpusph_amd.engine import TimestepEngine
n = float(sys.argv[1]) if len(sys.argv) > 1 else 4e6
prob = DamBreak3D(DamBreak3D.deltap_for(n))
eng = TimestepEngine(prob, track_particle_count=False)
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 0):
    eng.step()
eng.build_neibs(); torch.cuda.synchronize()
sp = eng.sp
A = eng.neibslist.numel() // int(sp.neiblistsize)
lst = eng.neibslist.view(int(sp.neiblistsize), A)[:, :eng.n].to(torch.int32) & 0xFFFF
nbp = int(sp.neibboundpos)
end = lst == D.NEIBS_END
big = 10 ** 6
idx = torch.arange(lst.shape[0], device=lst.device)[:, None]
cntF = torch.where(end[:nbp + 1], idx[:nbp + 1], big).min(dim=0).values.clamp(max=nbp + 1)      # entries before the terminator
rev = torch.flip(end[:nbp + 1], dims=[0])
cntB = torch.where(rev, idx[:nbp + 1], big).min(dim=0).values.clamp(max=nbp + 1)
real = ((lst < D.NEIBS_END) & ~end).sum()   # not used
cntF = cntF.cpu().numpy(); cntB = cntB.cpu().numpy()
info = eng.info.cpu().numpy().reshape(-1, 4)[:eng.n] if eng.info.dim() == 1 else eng.info.cpu().numpy()[:eng.n]
ptype = info[:, 0] & 7
f = eng.lib.sphx_dbg_tiles
f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]; f.restype = C.c_int
t = np.zeros((400000, 16), dtype=np.uint32)
nt = f(eng.ctx.handle, t.ctypes.data, len(t)); t = t[:nt]
t = t[(t[:, 13] & 1) != 0]
used = 0; walked = 0; walked_sorted = 0; lanes = 0
usedB = 0; walkedB = 0
rng = np.random.default_rng(0)
sel = rng.choice(len(t), size=min(len(t), 4000), replace=False)
for d in t[sel]:
    ids = np.concatenate([np.arange(d[4 + r], d[4 + r] + d[8 + r]) for r in range(4)])
    fl = ptype[ids] == 0
    c = np.where(np.ones_like(fl), cntF[ids], 0)         # fluid and DYN boundary particles walk the fluid section
    cb = np.where(fl, cntB[ids], 0)
    for w0 in range(0, len(ids), 64):
        cw = c[w0:w0 + 64]; used += cw.sum(); walked += 64 * (2 * ((cw.max() + 2) // 2)); lanes += 64
        bw = cb[w0:w0 + 64]; usedB += bw.sum()
        if bw.max() > 0: walkedB += 64 * (2 * ((bw.max() + 2) // 2))
    cs = np.sort(c)[::-1]
    for w0 in range(0, len(ids), 64):
        cw = cs[w0:w0 + 64]; walked_sorted += 64 * (2 * ((cw.max() + 2) // 2))
print("particles %d tiles %d sampled %d" % (eng.n, len(t), len(sel)))
print("fluid section: list entries used %.3e, lane-slots walked %.3e  -> efficiency %.3f  (lanes sorted by length within the tile: %.3f)"
      % (used, walked, used / walked, used / walked_sorted))
print("boundary section: used %.3e walked %.3e -> %.3f ; share of all walked slots %.3f" % (usedB, walkedB, usedB / max(walkedB, 1), walkedB / (walked + walkedB)))
print("mean entries per lane: fluid %.1f boundary %.1f" % (used / lanes, usedB / lanes))
