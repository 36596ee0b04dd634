"""diagnostic: two-fluid non-constant Newtonian viscosity: per-class sensitivities of one particle's force"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import oracle_lib as ol
from gpusph_amd import defs as D
from gpusph_amd.problem import DamBreak3D
from gpusph_amd.engine import TimestepEngine

def run(nu0, nu1):
    spec = dict(rheologytype=D.NEWTONIAN, turbmodel=D.LAMINAR_FLOW, compvisc=D.KINEMATIC, avgop=D.ARITHMETIC, is_const_visc=False)
    prob = DamBreak3D(deltap=0.045, obstacle=False, jitter=0.1, hydrostatic=False, viscosity=spec, kinematic_visc=0.05,
                      two_fluids=True, density_diffusion=D.DENSITY_DIFFUSION_NONE)
    prob.physparams.set_kinematic_visc(0, nu0); prob.physparams.set_kinematic_visc(1, nu1)
    eng = TimestepEngine(prob, device="cuda:0", clobber_neibslist=True)
    sim = ol.OracleSim(prob)
    sim.build_neibs(); eng.build_neibs()
    n = eng.n
    rng = np.random.default_rng(37)
    vel = sim.vel.copy()
    fluid = (sim.info[:, 0] & 7) == 0
    vel[fluid, :3] += rng.uniform(-0.3, 0.3, size=(fluid.sum(), 3)).astype(np.float32)
    sim.vel = vel
    eng.vel[:n] = torch.from_numpy(vel[:n]).to(eng.device)
    f_ref = sim.o.forces(sim.pos, sim.vel, sim.info, sim.hash, sim.cs, sim.nl, n)[0][:n].astype(np.float64)
    eng._forces(eng.pos, eng.vel, 1, 0)
    return eng.forces[:n].cpu().numpy().astype(np.float64)[:, :3], f_ref[:, :3], prob, sim, vel, n

a = run(0.05, 0.15); b = run(0.06, 0.15); c = run(0.05, 0.16)
prob, sim, vel, n = a[2], a[3], a[4], a[5]
fl = (sim.info[:n, 1] >> 12).astype(int); pt = (sim.info[:n, 0] & 7).astype(int)
gp = prob.global_pos(sim.pos[:n], sim.hash[:n])
from scipy.spatial import cKDTree
tree = cKDTree(gp)
h = float(np.float32(prob.simparams.slength)); fcoeff = 105.0 / (128.0 * np.pi * h ** 5)
rho = (vel[:n, 3].astype(np.float64) + 1.0) * np.array([1000.0, 850.0])[fl]
m = sim.pos[:n, 3].astype(np.float64); v = vel[:n, :3].astype(np.float64)
for i in (961, 960):
    nb = np.array([j for j in tree.query_ball_point(gp[i], 2 * h * (1 - 1e-7)) if j != i])
    r = np.linalg.norm(gp[i] - gp[nb], axis=1)
    F = (r / h - 2.0) ** 3 * fcoeff
    w = (m[nb] * F)[:, None] * (v[i] - v[nb])                     # pair weight vector
    own = (w / rho[nb][:, None])                                    # d/d nu_i  of (nu_i rho_i + nu_j rho_j)/(rho_i rho_j) * w
    oth = (w / rho[i])                                              # d/d nu_j
    cls = {"fluid0": (pt[nb] == 0) & (fl[nb] == 0), "fluid1": (pt[nb] == 0) & (fl[nb] == 1), "boundary": pt[nb] == 1}
    print("particle", i, "fluid", fl[i])
    print("  GPU    d/dnu0", (b[0][i] - a[0][i]) / 0.01, " d/dnu1", (c[0][i] - a[0][i]) / 0.01)
    print("  oracle d/dnu0", (b[1][i] - a[1][i]) / 0.01, " d/dnu1", (c[1][i] - a[1][i]) / 0.01)
    print("  own-viscosity part (all neighbours)", own.sum(axis=0))
    for k, cm in cls.items():
        print("  neighbour-viscosity part of class", k, oth[cm].sum(axis=0))
