"""SPH_GRENIER test infrastructure: a small two-fluid tank (the DamBreak3D mirror with the options of the reference's
Bubble / LockExchange / RTInstability problems: formulation<SPH_GRENIER>, viscosity<DYNAMICVISC>, boundary<DYN_BOUNDARY>,
ENABLE_MULTIFLUID) and a float64 all-pairs evaluation of Grenier's sums that shares no code with the oracle's list walk."""
import ctypes as C
import numpy as np

from gpusph_amd import defs as D
from gpusph_amd.problem import DamBreak3D, info_type
import oracle_lib as ol


def grenier_problem(deltap=0.04, viscosity="DYNAMICVISC", jitter=0.15, two_fluids=True):
    return DamBreak3D(deltap, obstacle=False, two_fluids=two_fluids, formulation=D.SPH_GRENIER, viscosity=viscosity,
                      density_diffusion=D.DENSITY_DIFFUSION_NONE, jitter=jitter)


def grenier_state(problem, seed=3, vel_scale=0.3):
    """sorted state + neighbour list + volumes, with a smooth random velocity field so that every term is exercised"""
    sim = ol.OracleSim(problem)
    sim.build_neibs()
    n = sim.n
    g = problem.global_pos(sim.pos[:n], sim.hash[:n])
    rng = np.random.default_rng(seed)
    A = rng.normal(size=(3, 3)) * vel_scale
    fluid = info_type(sim.info[:n]) == D.PT_FLUID
    sim.vel[:n, :3][fluid] = ((g[fluid] - g[fluid].mean(0)) @ A.T + 0.05 * np.sin(7 * g[fluid])).astype(np.float32)
    return sim, g


def fluid_num(info):
    return (info[:, 1] >> 12).astype(np.int64)


def wendland(q):
    """W and F = (1/r) dW/dr without their coefficients (src/cuda/sph_core.cu Wendland: (1 - q/2)^4 (2q + 1); F: (q - 2)^3)"""
    return (1 - q / 2) ** 4 * (2 * q + 1), (q - 2) ** 3


def brute_density(problem, sim, g, vol, max_fb):
    """sigma and rho~ of densityGrenierDevice by all pairs, float64"""
    n = sim.n
    p = sim.o.p
    h, R = float(p.slength), float(p.influenceradius)
    wc = float(sim.o.L.orc_wcoeff(C.c_int(D.WENDLAND), C.c_float(h), C.c_float(2.0)))
    t = info_type(sim.info[:n]); fl = fluid_num(sim.info[:n]); m = sim.pos[:n, 3].astype(np.float64)
    sigma = np.zeros(n); rho = np.zeros(n)
    for i in range(n):
        d = np.sqrt(((g - g[i]) ** 2).sum(1))
        near = (d < R)
        w = wendland(d[near] / h)[0] * wc
        tt, ff, mm = t[near], fl[near], m[near]
        s = w.sum()
        same = (tt == t[i]) & (ff == fl[i])
        if t[i] != D.PT_FLUID and not (tt == D.PT_FLUID).any():
            s = 3 * max_fb / (4 * np.pi * R ** 3)
        sigma[i] = s
        rho[i] = (mm[same] * w[same]).sum() / (w[same].sum() * vol[i, 3]) / float(p.rho0[fl[i]]) - 1
    return sigma, rho


def brute_forces(problem, sim, g, sigma):
    """Grenier's right-hand sides by all pairs, float64: (n, 4) with DvDt in xyz and D(log J)/Dt in w, after the fixup
    and gravity"""
    n = sim.n
    p = sim.o.p
    h, R = float(p.slength), float(p.influenceradius)
    fc = float(sim.o.L.orc_fcoeff(C.c_int(D.WENDLAND), C.c_float(h), C.c_float(2.0)))
    t = info_type(sim.info[:n]); fl = fluid_num(sim.info[:n])
    v = sim.vel[:n].astype(np.float64)
    rho0 = np.array([float(p.rho0[k]) for k in range(4)])
    rho = (v[:, 3] + 1) * rho0[fl]
    P = np.array([float(sim.o.L.orc_P(C.byref(p), C.c_float(sim.vel[i, 3]), C.c_int(int(fl[i])))) for i in range(n)], dtype=np.float64)
    pre = P / sigma
    visc = np.array([float(p.visccoeff[k]) for k in range(4)])
    mu = visc[fl] * (rho if p.compvisc == D.KINEMATIC else 1.0)
    eps = float(p.epsinterface)
    grav = np.array([float(p.gravity[a]) for a in range(3)])
    out = np.zeros((n, 4))
    for i in range(n):
        if t[i] not in (D.PT_FLUID, D.PT_BOUNDARY):
            continue
        rel = g[i] - g
        d = np.sqrt((rel ** 2).sum(1))
        near = (d < R)
        near[i] = False
        # fluid particles see fluid and boundary neighbours, boundary particles only fluid ones
        near &= (t == D.PT_FLUID) | ((t == D.PT_BOUNDARY) & (t[i] == D.PT_FLUID))
        r = rel[near]; f = wendland(d[near] / h)[1] * fc
        dv = v[i, :3] - v[near, :3]
        out[i, 3] = -((dv * r).sum(1) * f).sum() / sigma[i]
        if t[i] == D.PT_FLUID:
            pg = pre[i] + pre[near]
            iface = (t[near] == D.PT_FLUID) & (fl[near] != fl[i])
            pg = pg + np.where(iface, eps * (abs(pre[i]) + np.abs(pre[near])), 0.0)
            acc = -(pg * f)[:, None] * r
            if p.rheologytype == D.NEWTONIAN:
                a, b = mu[i], mu[near]
                avg = {D.ARITHMETIC: (a + b) / 2, D.HARMONIC: 2 * a * b / (a + b), D.GEOMETRIC: np.sqrt(a * b)}[int(p.avgop)]
                acc = acc + (avg * (1 / sigma[i] + 1 / sigma[near]) * f)[:, None] * dv
            out[i, :3] = acc.sum(0) / rho[i] + grav
    return out
