"""SPH_GRENIER, CPU side: the oracle's list-walking restatement of densityGrenierDevice / the Grenier specialisations of the
forces kernel / the volume integration against a float64 all-pairs evaluation and closed forms.  The CUDA kernels of the
reference cannot be compiled here, so this is what pins the restatement."""
import numpy as np
import pytest

from gpusph_amd import defs as D
from gpusph_amd.problem import info_type
import oracle_lib as ol
from grenier_helpers import grenier_problem, grenier_state, brute_density, brute_forces, fluid_num


@pytest.fixture(scope="module")
def st():
    pr = grenier_problem(0.04)
    sim, g = grenier_state(pr)
    n = sim.n
    vel0 = sim.vel.copy()
    mfb = sim.neibs_info.maxFluidBoundaryNeibs
    sigma = sim.o.density_grenier(sim.pos, sim.vel, sim.info, sim.hash, sim.vol, sim.cs, sim.nl, n, mfb)
    return dict(problem=pr, sim=sim, g=g, n=n, vel0=vel0, sigma=sigma, mfb=mfb)


def test_defaults_follow_the_reference():
    pr = grenier_problem(0.05)
    assert pr.physparams.epsinterface == 0.05                  # ProblemCore.cc:165-166
    assert pr.simparams.avgop == D.HARMONIC                    # legacy viscosity name + Grenier, cudasimframework.cu:202-210
    sp = pr.sphx_params(pr.num_particles)
    assert sp.sph_formulation == D.SPH_GRENIER and abs(sp.epsinterface - 0.05) < 1e-9


def test_dyn_boundary_particles_list_each_other_only_with_grenier():
    # buildneibs_kernel.cu:598: the boundary-boundary exclusion of DYN_BOUNDARY is lifted for SPH_GRENIER (sigma needs them)
    from gpusph_amd.problem import DamBreak3D
    a = ol.OracleSim(grenier_problem(0.05, jitter=0.0)); a.build_neibs()
    b = ol.OracleSim(DamBreak3D(0.05, obstacle=False, two_fluids=True, viscosity="DYNAMICVISC",
                                density_diffusion=D.DENSITY_DIFFUSION_NONE)); b.build_neibs()
    t = info_type(a.info[:a.n])
    i = int(np.where(t == D.PT_BOUNDARY)[0][0])
    bp = int(a.o.p.neibboundpos); stride = int(a.o.p.neiblist_stride)
    assert a.nl[bp * stride + i] != 0xFFFF and b.nl[bp * stride + i] == 0xFFFF
    assert a.neibs_info.numInteractions > b.neibs_info.numInteractions


def test_init_volume(st):
    sim, n = st["sim"], st["n"]
    pp = st["problem"].physparams
    fl = fluid_num(sim.info[:n])
    rho0 = np.array(pp.rho0)[fl]
    vol = sim.o.init_volume(sim.pos, st["vel0"], sim.info, n)
    want = sim.pos[:n, 3] / ((st["vel0"][:n, 3] + np.float32(1)) * rho0.astype(np.float32))
    np.testing.assert_array_equal(vol[:n, 0], want.astype(np.float32))
    np.testing.assert_array_equal(vol[:n, 3], vol[:n, 0])
    assert not vol[:n, 1:3].any()


def test_sigma_and_density_equal_all_pairs(st):
    sim, n = st["sim"], st["n"]
    sig, rho = brute_density(st["problem"], sim, st["g"], sim.vol, st["mfb"])
    np.testing.assert_allclose(st["sigma"][:n], sig, rtol=2e-5)
    np.testing.assert_allclose(sim.vel[:n, 3], rho, atol=3e-6)
    # only the density changes, and the density of an undisturbed lattice of equal masses is the one the volumes were made from
    np.testing.assert_array_equal(sim.vel[:n, :3], st["vel0"][:n, :3])
    t = info_type(sim.info[:n])
    # boundary particles that see no fluid get the 'typical' specific volume
    R = float(sim.o.p.influenceradius)
    typical = np.float32(3 * st["mfb"]) / (np.float32(4) * np.float32(np.pi) * np.float32(R) ** 3)
    lonely = (t == D.PT_BOUNDARY) & (np.abs(st["sigma"][:n] - typical) < 1e-3 * typical)
    assert lonely.sum() > 50 and not (lonely & (t == D.PT_FLUID)).any()
    # partition of unity: sigma dp^3 ~ 1 for particles deep in the fluid
    dp = st["problem"].m_deltap
    deep = (t == D.PT_FLUID) & (sig * dp ** 3 > 0.97)
    assert deep.sum() > 100 and np.all(sig[deep] * dp ** 3 < 1.2)


def test_forces_equal_all_pairs(st):
    sim, n = st["sim"], st["n"]
    f, cfl, nb = sim.o.forces_grenier(sim.pos, sim.vel, sim.info, sim.hash, sim.cs, sim.nl, st["sigma"], n)
    want = brute_forces(st["problem"], sim, st["g"], st["sigma"][:n].astype(np.float64))
    t = info_type(sim.info[:n])
    fluid = t == D.PT_FLUID
    scale = np.abs(want[fluid, :3]).max()
    assert scale > 10.0
    np.testing.assert_allclose(f[:n, :3][fluid], want[fluid, :3], atol=2e-4 * scale)
    np.testing.assert_allclose(f[:n, 3], want[:, 3], atol=2e-4 * np.abs(want[:, 3]).max())
    # boundary particles of DYN_BOUNDARY integrate a volume too but receive no acceleration
    assert np.abs(want[~fluid, 3]).max() > 0 and not f[:n, :3][~fluid].any()
    # the interface term acts: without it the momentum of the particles next to the other fluid changes
    eps = sim.o.p.epsinterface
    sim.o.p.epsinterface = 0.0
    f0 = sim.o.forces_grenier(sim.pos, sim.vel, sim.info, sim.hash, sim.cs, sim.nl, st["sigma"], n)[0]
    sim.o.p.epsinterface = eps
    changed = np.abs(f0[:n, :3] - f[:n, :3]).max(1) > 1e-3 * scale
    assert 20 < changed.sum() < fluid.sum()
    # CFL entries: max(|a|, c^2/h) per block of 128
    assert nb == int(ol.lib().orc_fmax_elements(n)) and cfl[:nb].max() > 0


def test_linear_velocity_field_gives_its_divergence():
    # D(log J)/Dt = div v: exact for the discrete operator up to the kernel's first-moment error on a lattice
    pr = grenier_problem(0.03, jitter=0.0, two_fluids=False)
    sim = ol.OracleSim(pr); sim.build_neibs()
    n = sim.n
    g = pr.global_pos(sim.pos[:n], sim.hash[:n])
    A = np.array([[0.3, 0.1, 0.0], [-0.2, 0.5, 0.05], [0.0, 0.1, -0.4]])
    sim.vel[:n, :3] = (g @ A.T).astype(np.float32)       # walls too: the field is linear everywhere
    sigma = sim.o.density_grenier(sim.pos, sim.vel, sim.info, sim.hash, sim.vol, sim.cs, sim.nl, n, sim.neibs_info.maxFluidBoundaryNeibs)
    f = sim.o.forces_grenier(sim.pos, sim.vel, sim.info, sim.hash, sim.cs, sim.nl, sigma, n)[0]
    t = info_type(sim.info[:n])
    dp = pr.m_deltap
    fl = g[t == D.PT_FLUID]
    R = float(sim.o.p.influenceradius) + dp
    deep = (t == D.PT_FLUID) & np.all((g > fl.min(0) + R) & (g < fl.max(0) - R), axis=1)      # full support of fluid particles
    assert deep.sum() > 50
    np.testing.assert_allclose(f[:n, 3][deep], np.trace(A), rtol=0.05)      # h/dp = 1.3: a few per cent of first-moment error
    assert np.ptp(f[:n, 3][deep]) < 1e-4


def test_euler_integrates_the_volume_not_the_density(st):
    sim, n = st["sim"], st["n"]
    f = sim.o.forces_grenier(sim.pos, sim.vel, sim.info, sim.hash, sim.cs, sim.nl, st["sigma"], n)[0]
    dt = np.float32(1e-3)
    vol = sim.vol.copy(); vol[:n, 1] = 0.01
    ps, vs, nv = sim.o.euler_grenier(sim.pos, sim.vel, vol, sim.info, sim.hash, f, n, float(dt), 1)
    t = info_type(sim.info[:n])
    np.testing.assert_array_equal(vs[:n, 3], sim.vel[:n, 3])                    # density untouched
    y = vol[:n, 1].astype(np.float64) + float(dt) * f[:n, 3].astype(np.float64)
    np.testing.assert_allclose(nv[:n, 1], y, rtol=1e-6)
    np.testing.assert_allclose(nv[:n, 3], np.exp(nv[:n, 1].astype(np.float64)) * vol[:n, 0], rtol=1e-6)
    np.testing.assert_array_equal(nv[:n, 0], vol[:n, 0])
    fluid = t == D.PT_FLUID
    np.testing.assert_allclose(vs[:n, :3][fluid], sim.vel[:n, :3][fluid] + dt * f[:n, :3][fluid], rtol=1e-6, atol=1e-7)
    assert np.abs(nv[:n, 1][~fluid] - 0.01).max() > 0                           # DYN boundary particles integrate theirs too


def test_a_few_steps_stay_finite_and_keep_the_mass_identity():
    pr = grenier_problem(0.04, jitter=0.0)
    sim = ol.OracleSim(pr)
    for _ in range(4):
        sim.step()
    n = sim.n
    assert np.isfinite(sim.pos[:n]).all() and np.isfinite(sim.vel[:n]).all() and np.isfinite(sim.vol[:n]).all()
    assert 1e-5 < sim.dt < 1e-2
    # rho omega = m for the smoothed mass of an equal-mass neighbourhood: checked at the state the last density pass saw
    assert np.abs(sim.vol[:n, 1]).max() < 0.05
