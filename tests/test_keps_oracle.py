"""turbulence<KEPSILON> in the CPU oracle (SA_BOUNDARY, solid walls): known answers of the restatement -- the closed form of the
semi-implicit Euler step, the boundary conditions of a uniform field, the boundary sums of DKDE recomputed in float64 from the
neighbour list, the law of the wall, and the launch-by-launch structure of the reference that decides what DKDE holds."""
import ctypes as C
import numpy as np
import pytest

from gpusph_amd import defs as D
from gpusph_amd.problem import SABox, info_type
from sa_helpers import OracleSaSim, list_sections

KEPS = dict(rheologytype=D.NEWTONIAN, turbmodel=D.KEPSILON)


@pytest.fixture(scope="module")
def sim():
    return OracleSaSim(SABox(deltap=0.05, viscosity=KEPS, jitter=0.05))


def _types(s):
    t = info_type(s.info[:s.n])
    return (np.where(t == k)[0] for k in (D.PT_FLUID, D.PT_BOUNDARY, D.PT_VERTEX))


def test_option_set_and_initial_state(sim):
    p = sim.o.p
    assert p.turbmodel == D.KEPSILON and p.is_const_visc == 0        # FullViscSpec default: not constant with k-epsilon
    k0, e0, nut0 = sim.problem.init_keps()
    c0 = float(sim.problem.physparams.sscoeff[0])
    assert k0 == pytest.approx((0.002 * c0) ** 2, rel=1e-6)
    assert e0 == pytest.approx(0.16 * k0 ** 1.5 / (2 * sim.problem.m_deltap), rel=1e-6)
    assert nut0 == pytest.approx(0.9 * k0 * k0 / e0, rel=1e-6)       # the reference's constant (0.9, not C_mu = 0.09)


def test_boundary_conditions_of_a_uniform_field(sim):
    """dk/dn = 0: a segment with fluid in reach takes the Shepard mean of a uniform k, i.e. k itself; its epsilon is the mean plus the
    wall-law term (larger than the fluid's); a vertex takes the plain mean over its adjacent segments; no Eulerian velocity appears"""
    fl, seg, vx = _types(sim)
    k0, e0, _ = sim.problem.init_keps()
    ke = sim.ke
    wet = seg[ke["tke"][seg] > 0]
    assert len(wet) > 200 and len(wet) < len(seg)                     # dry segments keep sum/max(0, 0.1 gamma) = 0
    assert (ke["tke"][wet] <= k0 * (1 + 1e-5)).all()                  # below k where the Shepard sum sits on its floor 0.1 gamma
    full = wet[ke["tke"][wet] > 0.99 * k0]
    assert len(full) > 200 and np.abs(ke["tke"][full] / k0 - 1).max() < 1e-5
    assert (ke["eps"][full] > e0).all()
    assert np.array_equal(ke["tke"][fl], np.full(len(fl), k0, np.float32))      # fluid rows untouched
    st = sim.st
    for v in vx[::37]:
        adj = [j for j in list_sections(st, v)[D.PT_BOUNDARY] if int(sim.info[v, 2]) | (int(sim.info[v, 3]) << 16) in sim.vertices[j, :3]]
        assert len(adj) >= 1
        assert ke["tke"][v] == pytest.approx(max(np.float32(ke["tke"][adj].astype(np.float64).mean()), 1e-6), rel=2e-6)
        assert ke["eps"][v] == pytest.approx(max(np.float32(ke["eps"][adj].astype(np.float64).mean()), 1e-6), rel=2e-6)
    assert not ke["eulervel"].any()


def test_euler_step_closed_form():
    """k' = (k + dt Dk)/(1 + dt e/k), e' = (e + dt De)/(1 + dt e/k C_e2); wall rows integrate their Eulerian velocity with the force"""
    s = OracleSaSim(SABox(deltap=0.08, viscosity=KEPS))
    n = s.n
    rng = np.random.default_rng(4)
    ke = dict(tke=rng.uniform(1e-3, 2e-3, n).astype(np.float32), eps=rng.uniform(1e-4, 3e-4, n).astype(np.float32),
              turbvisc=np.zeros(n, np.float32), eulervel=rng.normal(size=(n, 4)).astype(np.float32))
    dkde = np.stack([rng.normal(scale=1e-3, size=n), rng.normal(scale=1e-4, size=n), rng.uniform(0.5, 1.92, n)], 1).astype(np.float32)
    forces = rng.normal(size=(n, 4)).astype(np.float32)
    dt = 1e-3
    new = s.o.euler_keps(ke, dkde, forces, s.pos, s.info, n, dt)
    fl, seg, vx = _types(s)
    k, e = ke["tke"].astype(np.float64), ke["eps"].astype(np.float64)
    kk = (k + dt * dkde[:, 0]) / (1 + dt * e / k)
    ee = (e + dt * dkde[:, 1]) / (1 + dt * e / k * dkde[:, 2])
    assert np.abs(new["tke"][fl] / kk[fl] - 1).max() < 1e-6 and np.abs(new["eps"][fl] / ee[fl] - 1).max() < 1e-6
    wall = np.concatenate([seg, vx])
    assert np.array_equal(new["tke"][wall], ke["tke"][wall]) and np.array_equal(new["eps"][wall], ke["eps"][wall])
    assert np.allclose(new["eulervel"][wall], ke["eulervel"][wall] + np.float32(dt) * forces[wall], rtol=1e-6, atol=1e-9)
    assert np.array_equal(new["eulervel"][fl], ke["eulervel"][fl])
    assert np.abs(new["turbvisc"] / (0.9 * new["tke"].astype(np.float64) ** 2 / new["eps"]) - 1).max() < 1e-6


def _forces(s, vel=None, ke=None):
    return s.o.forces_sa_keps(s.pos, s.vel if vel is None else vel, s.info, s.hash, s.cs, s.nl, s.gg, s.be, s.vertpos,
                              s.ke if ke is None else ke, s.n, s.problem.m_deltap)


def test_dkde_holds_the_boundary_sums_and_nothing_else(sim):
    """every forcesDevice launch stores a fresh keps output (forces_kernel.def:1003-1013,3331-3339): after fluid<-fluid, fluid<-vertex,
    fluid<-boundary the DKDE / TAU rows of a fluid particle are those of the boundary launch.  A fluid particle without boundary
    elements in reach therefore has (0, 0, 1.92) whatever the k field around it does; vertex rows are cleared by their own launch."""
    s = sim
    fl, seg, vx = _types(s)
    rng = np.random.default_rng(1)
    ke = {k: v.copy() for k, v in s.ke.items()}
    ke["tke"][fl] *= rng.uniform(0.5, 1.5, len(fl)).astype(np.float32)         # strong gradients of k and epsilon in the fluid
    ke["eps"][fl] *= rng.uniform(0.5, 1.5, len(fl)).astype(np.float32)
    vel = s.vel.copy()
    vel[fl, :3] = rng.normal(scale=0.2, size=(len(fl), 3)).astype(np.float32)
    f, cfl, nb, dkde, strain = _forces(s, vel, ke)
    no_wall = np.array([i for i in fl if not list_sections(s.st, i)[D.PT_BOUNDARY]])
    assert len(no_wall) > 20
    assert not dkde[no_wall, :2].any() and (dkde[no_wall, 2] == np.float32(1.92)).all() and not strain[no_wall].any()
    assert not dkde[vx, :2].any() and (dkde[vx, 2] == np.float32(1.92)).all()
    near = np.setdiff1d(fl, no_wall)
    assert (np.abs(dkde[near, 1]) > 0).mean() > 0.7        # listed elements beyond the support contribute no |grad gamma|
    # and the eddy viscosity reaches the dt limit: per-block maxima over the fluid rows
    blk = s.o.cfl_keps[:nb]
    assert blk.max() == ke["turbvisc"][fl].max()


def test_boundary_sums_recomputed_in_float64(sim):
    """diffusion term of epsilon, Yap correction, strain rate and production for fluid particles at the wall, from the list and
    |grad gamma_as| of each element, in float64"""
    s = sim
    o, p = s.o, s.o.p
    fl, seg, vx = _types(s)
    rng = np.random.default_rng(7)
    vel = s.vel.copy()
    vel[fl, :3] = rng.normal(scale=0.3, size=(len(fl), 3)).astype(np.float32)
    ke = {k: v.copy() for k, v in s.ke.items()}
    ke["eulervel"][seg, :3] = rng.normal(scale=0.05, size=(len(seg), 3)).astype(np.float32)      # exercise relEulerVel
    f, cfl, nb, dkde, strain = _forces(s, vel, ke)
    gp = s.problem.global_pos(s.pos[:s.n], s.hash[:s.n])
    L = o.L
    L.orc_grad_gamma_vp.restype = C.c_float
    h, dp = float(p.slength), s.problem.m_deltap
    rho0 = float(p.rho0[0])
    checked, got_want = 0, []
    for i in fl[::11]:
        bnd = list_sections(s.st, i)[D.PT_BOUNDARY]
        if not bnd:
            continue
        k, e, nut = float(ke["tke"][i]), float(ke["eps"][i]), float(ke["turbvisc"][i])
        rho = (float(vel[i, 3]) + 1) * rho0
        de, ce2, T = 0.0, 1.92, np.zeros((3, 3))
        for j in bnd:
            r = gp[i] - gp[j]
            if np.linalg.norm(r) >= float(p.influenceradius) + dp:
                continue
            q = (r / h).astype(np.float32)
            be = s.be[j]
            gg = float(L.orc_grad_gamma_vp(C.c_float(h), C.c_float(q[0]), C.c_float(q[1]), C.c_float(q[2]), be.ctypes.data_as(C.c_void_p),
                                           s.vertpos[0][j].ctypes.data_as(C.c_void_p), s.vertpos[1][j].ctypes.data_as(C.c_void_p),
                                           s.vertpos[2][j].ctypes.data_as(C.c_void_p)))
            ns = be[:3].astype(np.float64)
            r_as = max(abs(r @ ns), dp)
            lyap = 0.400772603 * k ** 1.5 / (e * r_as)
            if lyap > 1:
                ce2 = min(ce2, max(1.92 - 0.83 * (lyap - 1) * lyap * lyap, 0.0))
            de += 0.276923077 * k * k / r_as * gg
            w = (vel[i, :3] - vel[j, :3]).astype(np.float64) + (ke["eulervel"][i, :3] - ke["eulervel"][j, :3])
            rho_s = (float(vel[j, 3]) + 1) * rho0
            T += np.outer(w, gg * ns * rho_s)
        gam = float(s.gg[i, 3])
        rg = rho * gam
        S2 = (2 * (T[0, 0] ** 2 + T[1, 1] ** 2 + T[2, 2] ** 2) + (T[0, 1] + T[1, 0]) ** 2 + (T[0, 2] + T[2, 0]) ** 2 + (T[1, 2] + T[2, 1]) ** 2)
        S = np.sqrt(S2) / rg
        Pt = min(nut * S2 / rg ** 2, 0.3 * k * S)
        want = np.array([Pt, de / rg + e * 1.44 * Pt / k, ce2])
        got_want.append((dkde[i].astype(np.float64), want))
        checked += 1
    assert checked > 20
    got, want = (np.array(x) for x in zip(*got_want))
    # |grad gamma_as| of an element at the edge of the support is sensitive to the float32 rounding of the relative position:
    # absolute tolerance on the scale of each column
    for c in range(3):
        assert np.abs(got[:, c] - want[:, c]).max() < 2e-3 * np.abs(want[:, c]).max(), c


def test_law_of_the_wall_and_the_pressure_of_k(sim):
    """fluid at rest: no wall shear (u_t = 0), no production; k only enters through P + 2/3 rho k.  Moving fluid: the wall term acts
    against the tangential velocity, and is bounded by the viscous-sublayer expression 2 |grad gamma| nu |u_t|/r applied per element"""
    s = sim
    fl, seg, vx = _types(s)
    vel0 = s.vel.copy(); vel0[:, :3] = 0
    f0, _, _, dk0, _ = _forces(s, vel0)
    assert not dk0[fl, 0].any()                                         # no strain, no production
    ke2 = {k: v.copy() for k, v in s.ke.items()}
    ke2["tke"] = (ke2["tke"] * 2).astype(np.float32)
    f1, *_ = _forces(s, vel0, ke2)
    inner = np.array([i for i in fl[::5] if not list_sections(s.st, i)[D.PT_BOUNDARY] and not list_sections(s.st, i)[D.PT_VERTEX]])
    # deep in the fluid a uniform k is a uniform pressure: its SPH gradient -2 (2k/3 rho^2) sum m F r is the discretisation error of a constant
    assert np.abs(f1[inner, :3] - f0[inner, :3]).max() < 0.05 * np.abs(f0[fl, :3]).max()
    velx = vel0.copy(); velx[fl, 0] = 0.3
    fx, *_ = _forces(s, velx)
    floor = np.array([i for i in fl if list_sections(s.st, i)[D.PT_BOUNDARY] and s.pos[i, 2] < -0.0 or False][:0] + [])
    near = np.array([i for i in fl if list_sections(s.st, i)[D.PT_BOUNDARY]])
    drag = fx[near, 0] - f0[near, 0]
    assert (drag < 0).mean() > 0.8 and drag.min() < -1e-3                # the walls brake the stream
    assert np.abs(fx[inner, 0] - f0[inner, 0]).max() < 1e-2 * np.abs(drag).max() + 1e-4   # uniform stream: no shear in the bulk
