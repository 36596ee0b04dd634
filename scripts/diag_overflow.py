import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import oracle_lib as ol
from gpusph_amd.problem import DamBreak3D
from gpusph_amd.engine import TimestepEngine
prob = DamBreak3D(deltap=0.045, obstacle=False, jitter=0.1, hydrostatic=False)
prob.simparams.neiblistsize = 64; prob.simparams.neibboundpos = 63
sim = ol.OracleSim(prob); sim.build_neibs()
eng = TimestepEngine(prob, clobber_neibslist=True); eng.build_neibs()
n = sim.n
gl = eng.neibslist.cpu().numpy().view(np.uint16).reshape(-1, eng.alloc)[:, :n]
ol_ = sim.nl.reshape(-1, len(sim.pos))[:, :n]
d = np.argwhere(gl != ol_)
print("diff entries", len(d), "particles", len(np.unique(d[:, 1])))
for i in np.unique(d[:, 1])[:3]:
    print("particle", i, "type", sim.info[i, 0] & 7)
    print(" gpu", gl[:, i]); print(" ref", ol_[:, i])
