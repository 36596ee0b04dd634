import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import oracle_lib as ol
from gpusph_amd.problem import PeriodicBox
from gpusph_amd.engine import TimestepEngine
prob = PeriodicBox(deltap=0.05, n=(30, 24, 20), jitter=0.2, velocity=(5.0, -3.0, 2.0))
sim = ol.OracleSim(prob); sim.build_neibs()
n = sim.n
rng = np.random.default_rng(21)
vel = sim.vel.copy()
vel[:, :3] += rng.uniform(-0.3, 0.3, size=(len(vel), 3)).astype(np.float32)
vel[:, 3] += rng.uniform(0, 2e-3, size=len(vel)).astype(np.float32)
sim.vel = vel
f_ref = sim.o.forces(sim.pos, sim.vel, sim.info, sim.hash, sim.cs, sim.nl, n)[0]
g = prob.grid_pos_from_hash(sim.hash[:n])
print("grid", prob.m_gridsize, "lin", prob.linearization)
for dis in ("0", "1"):
    os.environ["SPHX_DISABLE_TILES"] = dis
    eng = TimestepEngine(prob, clobber_neibslist=True)
    eng.build_neibs()
    eng.vel[:n] = torch.from_numpy(vel[:n]).to(eng.device)
    eng._forces(eng.pos, eng.vel, 1, 0)
    f = eng.forces[:n].cpu().numpy()
    err = np.abs(f[:, :3] - f_ref[:n, :3]).max(axis=1)
    bad = err > 1e-3
    print("disable_tiles", dis, "bad", bad.sum(), "of", n)
    if bad.any():
        gb = g[bad]
        for a in range(3):
            print(" axis", a, "cells of bad particles:", np.unique(gb[:, a]))
        i = np.where(bad)[0][:5]
        print(" idx", i, "\n f", f[i], "\n ref", f_ref[i])
