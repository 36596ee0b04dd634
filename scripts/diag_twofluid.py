import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import oracle_lib as ol
from gpusph_amd.problem import DamBreak3D
from gpusph_amd.engine import TimestepEngine
prob = DamBreak3D(deltap=0.045, obstacle=False, jitter=0.1, hydrostatic=False, two_fluids=True)
sim = ol.OracleSim(prob); sim.build_neibs()
eng = TimestepEngine(prob, clobber_neibslist=True); eng.build_neibs()
n = sim.n
rng = np.random.default_rng(34)
vel = sim.vel.copy()
fluid = (sim.info[:, 0] & 7) == 0
vel[fluid, :3] += rng.uniform(-0.3, 0.3, size=(fluid.sum(), 3)).astype(np.float32)
vel[:, 3] += rng.uniform(0, 2e-3, size=len(vel)).astype(np.float32)
sim.vel = vel; eng.vel[:n] = torch.from_numpy(vel[:n]).to(eng.device)
f_ref = sim.o.forces(sim.pos, sim.vel, sim.info, sim.hash, sim.cs, sim.nl, n)[0]
eng._forces(eng.pos, eng.vel, 1, 0)
f = eng.forces[:n].cpu().numpy()
err = np.abs(f[:, 3] - f_ref[:n, 3])
bad = np.argsort(-err)[:8]
gp = prob.global_pos(sim.pos[:n], sim.hash[:n])
for i in bad:
    print(i, "type", sim.info[i, 0] & 7, "fluid", sim.info[i, 1] >> 12, "z %.3f" % gp[i, 2], "w gpu %.6f ref %.6f" % (f[i, 3], f_ref[i, 3]), "xyz err %.2e" % np.abs(f[i, :3] - f_ref[i, :3]).max())
print("n bad > 1e-4:", (err > 1e-4).sum(), "of", n, " interface z", 0.5 * prob.H)
