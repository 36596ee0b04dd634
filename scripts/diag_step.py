import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
from gpusph_amd.problem import DamBreak3D
from gpusph_amd.engine import TimestepEngine
import oracle_lib as ol
prob = DamBreak3D(0.05, obstacle=True, jitter=0.05)
eng = TimestepEngine(prob, clobber_neibslist=True)
sim = ol.OracleSim(prob)
sim.build_neibs(); eng.build_neibs()
o = sim.o; n = sim.n
cof = 1
f1, cfl, nb, _, _ = o.forces(sim.pos, sim.vel, sim.info, sim.hash, sim.cs, sim.nl, n, compute_object_forces=cof, rb_count=prob.num_obstacle)
eng._forces(eng.pos, eng.vel, 1, 0)
g1 = eng.forces[:n].cpu().numpy()
types = sim.info[:n, 0] & 7
def rep(name, a, b):
    d = np.abs(a - b)
    i = np.unravel_index(d.argmax(), d.shape)
    print(name, "max diff", d.max(), "at", i, "type", types[i[0]], "flags", hex(sim.info[i[0], 0]), "vals", a[i], b[i], "scale", np.abs(b).max())
rep("F1 xyz", g1[:, :3], f1[:n, :3]); rep("F1 w", g1[:, 3:], f1[:n, 3:])
dt = float(np.float32(sim.dt)); hdt = float(np.float32(dt) / np.float32(2))
ps, vs = o.euler(sim.pos, sim.vel, sim.info, sim.hash, f1, n, hdt, 1)
eng._euler(1, 0.5)
rep("pos*", eng.pos2[:n].cpu().numpy(), ps[:n]); rep("vel*", eng.vel2[:n].cpu().numpy()[:, :3], vs[:n, :3]); rep("rho*", eng.vel2[:n].cpu().numpy()[:, 3:], vs[:n, 3:])
# now forces on the ORACLE's predicted state uploaded to the GPU (isolates the forces kernel)
eng.pos2[:n] = torch.from_numpy(ps[:n]).cuda(); eng.vel2[:n] = torch.from_numpy(vs[:n]).cuda()
f2, _, _, _, _ = o.forces(ps, vs, sim.info, sim.hash, sim.cs, sim.nl, n, compute_object_forces=cof, rb_count=prob.num_obstacle)
eng._forces(eng.pos2, eng.vel2, 2, 1)
g2 = eng.forces[:n].cpu().numpy()
rep("F2 xyz (same input)", g2[:, :3], f2[:n, :3]); rep("F2 w (same input)", g2[:, 3:], f2[:n, 3:])
for t in (0, 1):
    m = types == t
    print("type", t, "F2 xyz maxdiff", np.abs(g2[m, :3] - f2[:n][m, :3]).max(), "w", np.abs(g2[m, 3] - f2[:n][m, 3]).max())
