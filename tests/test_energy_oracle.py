"""ENABLE_INTERNAL_ENERGY (AccuracyTest.cu's flag), CPU side: the internal-energy rate the forces passes accumulate and its
integration, against the balance they are built for: what the pair forces take out of the kinetic energy goes into the
internal energy."""
import numpy as np
import pytest

from gpusph_amd import defs as D
from gpusph_amd.problem import DamBreak3D, PeriodicBox, info_type
import oracle_lib as ol


def _rates(prob, seed=5, amp=0.4):
    sim = ol.OracleSim(prob)
    sim.build_neibs()
    n = sim.n
    rng = np.random.default_rng(seed)
    fluid = info_type(sim.info[:n]) == D.PT_FLUID
    sim.vel[:n, :3][fluid] += rng.uniform(-amp, amp, size=(fluid.sum(), 3)).astype(np.float32)
    sim.vel[:n, 3] += rng.uniform(0, 2e-3, size=n).astype(np.float32)
    dedt = np.zeros(len(sim.pos), dtype=np.float32)
    f = sim.o.forces(sim.pos, sim.vel, sim.info, sim.hash, sim.cs, sim.nl, n, dedt=dedt)[0]
    return sim, f[:n], dedt[:n]


@pytest.mark.parametrize("visc", [None, "DYNAMICVISC"])
def test_pair_forces_conserve_kinetic_plus_internal_energy(visc):
    # periodic box without gravity: every pair is fluid-fluid and antisymmetric, so  sum m (v.a + de/dt) = 0
    prob = PeriodicBox(0.05, jitter=0.2, density_diffusion=D.DENSITY_DIFFUSION_NONE, viscosity=visc, kinematic_visc=0.05)
    prob.simparams.simflags |= D.ENABLE_INTERNAL_ENERGY
    sim, f, dedt = _rates(prob)
    n = sim.n
    m = sim.pos[:n, 3].astype(np.float64)
    v = sim.vel[:n, :3].astype(np.float64)
    a = f[:, :3].astype(np.float64)
    kin = (m * (v * a).sum(1)).sum()
    internal = (m * dedt.astype(np.float64)).sum()
    scale = (m * np.abs((v * a).sum(1))).sum()
    assert scale > 0 and abs(kin + internal) < 2e-5 * scale
    assert np.abs(dedt).max() > 0
    if visc is not None:      # viscosity only dissipates: with the pressure switched off every pair heats
        sim.vel[:n, 3] = 0.0
        d2 = np.zeros(len(sim.pos), dtype=np.float32)
        sim.o.forces(sim.pos, sim.vel, sim.info, sim.hash, sim.cs, sim.nl, n, dedt=d2)
        assert d2[:n].min() > -1e-6 * np.abs(d2[:n]).max() and d2[:n].max() > 0


def test_flag_off_leaves_the_buffer_alone_and_dyn_walls_get_a_rate_only_with_it():
    prob = DamBreak3D(0.05, obstacle=False, internal_energy=True, jitter=0.1)
    sim, f, dedt = _rates(prob)
    t = info_type(sim.info[:sim.n])
    assert np.abs(dedt[t == D.PT_BOUNDARY]).max() > 0            # forces_kernel.def:3661: walls compute the momentum terms for it
    assert not f[t == D.PT_BOUNDARY, :3].any()                    # ... without receiving an acceleration
    plain = DamBreak3D(0.05, obstacle=False, jitter=0.1)
    sim2 = ol.OracleSim(plain); sim2.build_neibs()
    buf = np.full(len(sim2.pos), 7.0, dtype=np.float32)
    sim2.o.forces(sim2.pos, sim2.vel, sim2.info, sim2.hash, sim2.cs, sim2.nl, sim2.n, dedt=buf)
    assert (buf == 0).all()                                      # cleared by the caller, never written


def test_lj_walls_enter_through_the_repulsion():
    prob = DamBreak3D(0.05, obstacle=False, internal_energy=True, boundary=D.LJ_BOUNDARY, jitter=0.1, hydrostatic=False)
    sim, f, dedt = _rates(prob, amp=0.8)
    n = sim.n
    a = sim.o.p.simflags
    sim.o.p.simflags = a & ~D.ENABLE_INTERNAL_ENERGY
    d0 = np.zeros(len(sim.pos), dtype=np.float32)
    sim.o.forces(sim.pos, sim.vel, sim.info, sim.hash, sim.cs, sim.nl, n, dedt=d0)
    assert not d0.any()
    assert np.isfinite(dedt).all() and np.abs(dedt).max() > 0


def test_euler_integrates_the_energy_and_steps_run():
    prob = DamBreak3D(0.05, obstacle=False, internal_energy=True)
    sim = ol.OracleSim(prob)
    for _ in range(12):
        sim.step()
    n = sim.n
    assert np.isfinite(sim.energy[:n]).all() and np.abs(sim.energy[:n]).max() > 0
    e = sim.o.euler_energy(sim.energy, sim.dedt, sim.pos, sim.info, n, 1e-3)
    t = info_type(sim.info[:n])
    np.testing.assert_allclose(e[:n], sim.energy[:n] + np.float32(1e-3) * sim.dedt[:n], rtol=1e-6, atol=1e-12)
    # an inactive particle keeps its energy
    pos = sim.pos.copy(); pos[0, 3] = np.nan
    e2 = sim.o.euler_energy(sim.energy, sim.dedt, pos, sim.info, n, 1e-3)
    assert e2[0] == sim.energy[0]
