"""Float64 ALL-PAIRS known answers for the headline option set (DamBreak3D as bench.py runs it: SPH_F1, Wendland, artificial
viscosity, Colagrossi density diffusion, DYN_BOUNDARY) and for the SPS stress tensor.

Whole-function parity of the pair loop is unpinned by reference outputs, so the restatement in
oracle/sph_oracle.c is held here by an evaluation that shares nothing with it: global float64 positions, neighbours by
distance (k-d tree -- no cells, no hash, no neighbour list, no list order), the formulas of the reference written from its
source text:
  continuity            m_j (v_ij . r_ij) F_ij                                        forces_kernel.def:2140-2151
  Colagrossi            - D c0 (rho_j/rho_i - 1) F_ij m_j, fluid neighbours of the same fluid, only where
                        |P_i - P_j| >= |g . r_ij| rho_i                               forces_kernel.def:1916-1952
  pressure              -(P_i/rho_i^2 + P_j/rho_j^2) m_j F_ij r_ij                     forces_kernel.def:2451-2466
  artificial viscosity  approaching pairs: (v.r) h alpha (c_i + c_j)/((r^2 + eps)(rho_i + rho_j)) m_j F_ij r_ij
                                                                                      visc_kernel.cu:74-85, forces_kernel.def:2748-2764
  who interacts         fluid <- fluid, fluid <- DYN boundary (no diffusion), DYN boundary <- fluid (continuity + diffusion;
                        momentum only with force feedback)                            forces_kernel.def:3565-3679
  finalize              drho~/dt = DrDt / rho0, gravity on fluid particles            forces_kernel.def:3212-3218,4032-4150
  SPS                   dv_a = -sum_j (v_i - v_j)_a r_ij F_ij m_j/rho_j over ALL neighbours, tau from it
                                                                                      visc_kernel.cu:307-407,759-811
The Colagrossi switch is decided by the last bits of P for pairs on its threshold; such pairs are identified (float64 margin
within a few float32 ulp of the pressures) and their term is allowed either way -- they are counted and must stay rare."""
import numpy as np
import pytest
from scipy.spatial import cKDTree

import oracle_lib as ol
from gpusph_amd import defs as D
from gpusph_amd.problem import DamBreak3D


def _state(viscosity=None, seed=5):
    prob = DamBreak3D(0.025, obstacle=True, jitter=0.15, hydrostatic=True, viscosity=viscosity, kinematic_visc=1.0e-6)
    sim = ol.OracleSim(prob)
    sim.build_neibs()
    n = sim.n
    rng = np.random.default_rng(seed)
    # a sheared, converging velocity field plus noise (approaching and separating pairs), densities off the hydrostatic
    # profile by up to 0.4 % (so that most pairs are far from the diffusion threshold)
    gp = prob.global_pos(sim.pos[:n], sim.hash[:n])
    v = np.stack([0.8*np.sin(3.0*gp[:, 2]) - 0.5*gp[:, 0], 0.6*np.cos(2.0*gp[:, 0]) - 0.4*gp[:, 1],
                  0.3*gp[:, 0]*gp[:, 1] - 0.5*gp[:, 2]], axis=1) + rng.normal(0, 0.15, size=(n, 3))
    sim.vel[:n, :3] = v.astype(np.float32)
    sim.vel[:n, 3] += rng.uniform(-4e-3, 4e-3, size=n).astype(np.float32)
    return prob, sim, gp


def _consts(sim):
    p = sim.o.p
    h = float(p.slength)
    return dict(h=h, rho0=float(p.rho0[0]), B=float(p.bcoeff[0]), gam=float(p.gammacoeff[0]), c0=float(p.sscoeff[0]),
                cpow=float(p.sspowercoeff[0]), g=np.array([p.gravity[0], p.gravity[1], p.gravity[2]], dtype=np.float64),
                alpha=float(p.artvisccoeff), eps=float(p.epsartvisc), Dc=float(p.densityDiffCoeff),
                fcoeff=105.0/(128.0*np.pi*h**5), R=float(p.influenceradius))


def test_headline_option_set_equals_a_float64_all_pairs_evaluation():
    prob, sim, gp = _state()
    n = sim.n
    p = sim.o.p
    assert p.kerneltype == D.WENDLAND and p.densitydiffusiontype == D.COLAGROSSI and p.boundarytype == D.DYN_BOUNDARY
    assert p.turbmodel == D.ARTIFICIAL and p.sph_formulation == D.SPH_F1
    k = _consts(sim)
    assert k["gam"] == 7.0 and k["cpow"] == 3.0 and k["alpha"] > 0 and k["Dc"] > 0
    f = sim.o.forces(sim.pos, sim.vel, sim.info, sim.hash, sim.cs, sim.nl, n)[0][:n].astype(np.float64)

    ptype = sim.info[:n, 0] & 7
    fluid, bound = ptype == D.PT_FLUID, ptype == D.PT_BOUNDARY
    assert fluid.sum() > 3000 and bound.sum() > 3000, (fluid.sum(), bound.sum())
    m = sim.pos[:n, 3].astype(np.float64)
    v = sim.vel[:n, :3].astype(np.float64)
    ratio = sim.vel[:n, 3].astype(np.float64) + 1.0
    rho = ratio*k["rho0"]
    P = k["B"]*(ratio**k["gam"] - 1.0)
    c = k["c0"]*ratio**k["cpow"]
    # float32 rounding of P: what can move a pair across the Colagrossi threshold
    Pulp = np.spacing(np.abs(P).astype(np.float32)).astype(np.float64)*4.0 + 1e-30

    tree = cKDTree(gp)
    acc = np.zeros((n, 3)); drdt = np.zeros(n); amb = np.zeros(n)
    pairs = ambiguous = 0
    nbs_all = tree.query_ball_point(gp, k["R"]*(1 - 1e-7))
    for i in range(n):
        nb = np.array([j for j in nbs_all[i] if j != i], dtype=np.int64)
        if bound[i]:
            nb = nb[fluid[nb]]                     # DYN boundary particles list fluid neighbours only
        if not len(nb):
            continue
        d = gp[i] - gp[nb]
        r2 = (d*d).sum(axis=1)
        r = np.sqrt(r2)
        F = (r/k["h"] - 2.0)**3*k["fcoeff"]
        vr = ((v[i] - v[nb])*d).sum(axis=1)
        mF = m[nb]*F
        cont = mF*vr
        nf = fluid[nb]
        margin = np.abs(P[i] - P[nb]) - np.abs((d @ k["g"])*rho[i])
        dterm = k["Dc"]*k["c0"]*(rho[nb]/rho[i] - 1.0)*mF
        on = nf & (margin >= 0.0)
        edge = nf & (np.abs(margin) <= Pulp[i] + Pulp[nb] + 1e-6*np.abs((d @ k["g"])*rho[i]))
        drdt[i] = cont.sum() - dterm[on & ~edge].sum()
        # threshold pairs: counted either way
        amb[i] = np.abs(dterm[edge]).sum()
        drdt[i] -= 0.5*dterm[edge].sum()
        pairs += int(nf.sum()); ambiguous += int(edge.sum())
        if fluid[i]:
            pg = P[i]/rho[i]**2 + P[nb]/rho[nb]**2
            visc = np.where(vr < 0.0, vr*k["h"]*k["alpha"]*(c[i] + c[nb])/((r2 + k["eps"])*(rho[i] + rho[nb])), 0.0)
            acc[i] = ((visc - pg)*mF) @ d + k["g"]
    drt = drdt/k["rho0"]
    amb = 0.5*amb/k["rho0"]

    # momentum equation: every fluid particle; walls without force feedback carry no acceleration
    ascale = np.abs(acc[fluid]).max()
    assert ascale > 50.0
    assert np.abs(f[fluid, :3] - acc[fluid]).max() <= 2e-5*ascale
    nofb = bound & ((sim.info[:n, 0] & D.FG_COMPUTE_FORCE) == 0)
    assert nofb.sum() > 0 and not f[nofb, :3].any()
    # continuity equation: fluid and wall particles; threshold pairs widen the bound of their two particles only
    dscale = np.abs(drt).max()
    assert dscale > 1.0
    err = np.abs(f[:, 3] - drt)
    assert (err <= 2e-5*dscale + amb*(1 + 1e-6)).all(), "drho/dt differs: %g of %g" % ((err - amb).max(), dscale)
    assert ambiguous <= 0.01*pairs, "too many pairs on the diffusion threshold to test anything: %d of %d" % (ambiguous, pairs)
    tight = amb == 0.0
    assert tight.mean() > 0.9 and np.abs(f[tight, 3] - drt[tight]).max() <= 2e-5*dscale
    # both branches of the switch and of the viscosity were exercised, on walls too
    assert np.abs(drt[bound]).max() > 0.05*dscale


def _tau_all_pairs(sim, gp, k, smag, ksps):
    n = sim.n
    m = sim.pos[:n, 3].astype(np.float64)
    v = sim.vel[:n, :3].astype(np.float64)
    rho = (sim.vel[:n, 3].astype(np.float64) + 1.0)*k["rho0"]
    ptype = sim.info[:n, 0] & 7
    bound = ptype == D.PT_BOUNDARY
    fluid = ptype == D.PT_FLUID
    tree = cKDTree(gp)
    tau = np.zeros((n, 6)); nu = np.zeros(n)
    nbs_all = tree.query_ball_point(gp, k["R"]*(1 - 1e-7))
    for i in range(n):
        nb = np.array([j for j in nbs_all[i] if j != i], dtype=np.int64)
        if bound[i]:
            nb = nb[fluid[nb]]
        if not len(nb):
            continue
        d = gp[i] - gp[nb]
        r = np.sqrt((d*d).sum(axis=1))
        w = (r/k["h"] - 2.0)**3*k["fcoeff"]*m[nb]/rho[nb]
        dv = -((v[i] - v[nb])[:, :, None]*(d*w[:, None])[:, None, :]).sum(axis=0)     # dv[a][b] = d v_a / d x_b
        xx, yy, zz = dv[0, 0], dv[1, 1], dv[2, 2]
        xy, xz, yz = dv[0, 1] + dv[1, 0], dv[0, 2] + dv[2, 0], dv[1, 2] + dv[2, 1]
        S2 = 2.0*(xx*xx + yy*yy + zz*zz) + xy*xy + xz*xz + yz*yz
        S = np.sqrt(S2)
        nu[i] = smag*S
        divu = (2.0/3.0)*nu[i]*(xx + yy + zz)
        bl = ksps*S2
        tau[i] = [(2*nu[i]*xx - divu - bl)/rho[i], nu[i]*xy/rho[i], nu[i]*xz/rho[i],
                  (2*nu[i]*yy - divu - bl)/rho[i], nu[i]*yz/rho[i], (2*nu[i]*zz - divu - bl)/rho[i]]
    return tau, nu


def test_sps_stress_tensor_equals_a_float64_all_pairs_evaluation():
    prob, sim, gp = _state(viscosity="SPSVISC", seed=9)
    n = sim.n
    p = sim.o.p
    assert p.turbmodel == D.SPS
    k = _consts(sim)
    smag, ksps = float(p.smagfactor), float(p.kspsfactor)
    dp = float(prob.m_deltap)
    assert smag == pytest.approx((0.12*dp)**2, rel=1e-6) and ksps == pytest.approx((2*0.0066/3)*dp*dp, rel=1e-6)   # GPUSPH.cc:1542-1554
    tau, tv = sim.o.sps(sim.pos, sim.vel, sim.info, sim.hash, sim.cs, sim.nl, n, n)
    ref, nu = _tau_all_pairs(sim, gp, k, smag, ksps)
    scale = np.abs(ref).max(axis=0)
    assert (scale > 0).all() and np.abs(nu).max() > 0
    assert (np.abs(tau[:n] - ref) <= 2e-5*scale.max()).all(), np.abs(tau[:n] - ref).max(axis=0)/scale.max()
    assert np.abs(tv[:n] - nu).max() <= 2e-5*np.abs(nu).max()
    # walls get a stress from their fluid neighbours too (their rows are read by the fluid's forces pass)
    bound = (sim.info[:n, 0] & 7) == D.PT_BOUNDARY
    assert np.abs(ref[bound]).max() > 0.01*scale.max()

    # the consumer: forces with tau minus forces with tau = 0 is sum_j m_j F_ij (tau_i + tau_j) . r_ij for the pairs that carry
    # the momentum equation (forces_kernel.def:2777-2798)
    # (the stress of this coarse, gently sheared state is 1e-4 of the pressure term: scaled up so that the term stands above the
    # float32 rounding of the accelerations it is the difference of)
    tau = (tau*np.float32(1000.0)).astype(np.float32)
    f_tau = sim.o.forces(sim.pos, sim.vel, sim.info, sim.hash, sim.cs, sim.nl, n, tau=tau)[0][:n].astype(np.float64)
    f_0 = sim.o.forces(sim.pos, sim.vel, sim.info, sim.hash, sim.cs, sim.nl, n, tau=np.zeros_like(tau))[0][:n].astype(np.float64)
    term = f_tau[:, :3] - f_0[:, :3]
    assert np.array_equal(f_tau[:, 3], f_0[:, 3])
    T = np.zeros((n, 3, 3))
    t64 = tau[:n].astype(np.float64)
    T[:, 0, 0], T[:, 0, 1], T[:, 0, 2], T[:, 1, 1], T[:, 1, 2], T[:, 2, 2] = t64.T
    T[:, 1, 0], T[:, 2, 0], T[:, 2, 1] = T[:, 0, 1], T[:, 0, 2], T[:, 1, 2]
    m = sim.pos[:n, 3].astype(np.float64)
    fluid = (sim.info[:n, 0] & 7) == D.PT_FLUID
    tree = cKDTree(gp)
    want = np.zeros((n, 3))
    for i in np.where(fluid)[0]:
        nb = np.array([j for j in tree.query_ball_point(gp[i], k["R"]*(1 - 1e-7)) if j != i], dtype=np.int64)
        d = gp[i] - gp[nb]
        r = np.sqrt((d*d).sum(axis=1))
        mF = m[nb]*(r/k["h"] - 2.0)**3*k["fcoeff"]
        want[i] = np.einsum("j,jab,jb->a", mF, T[i][None] + T[nb], d)
    ts = np.abs(want).max()
    fs = np.abs(f_0[fluid, :3]).max()
    assert ts > 0.1*fs
    # a difference of two float32 accelerations: their rounding adds to the bound of the term
    assert np.abs(term[fluid] - want[fluid]).max() <= 2e-5*ts + 1e-6*fs
