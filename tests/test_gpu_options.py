"""GPU parity over the run-time options the library advertises (DESIGN.md "Option set built"): every cell
linearisation, every kernel, viscosity / diffusion switches, two fluids, moving rigid bodies in the integrator.
Each case: neighbour phase bit-exact, forces within 2e-5 of the largest component, tiled == generic where both exist."""
import numpy as np
import pytest

import oracle_lib as ol
from gpusph_amd import defs as D
from gpusph_amd.problem import DamBreak3D

pytestmark = pytest.mark.gpu


def _engine(problem, **kw):
    from gpusph_amd.engine import TimestepEngine
    return TimestepEngine(problem, device="cuda:0", **kw)


def _np(t, dtype=None):
    a = t.cpu().numpy()
    return a.view(dtype) if dtype is not None else a


def _perturb(sim, eng, seed):
    import torch
    n = eng.n
    rng = np.random.default_rng(seed)
    vel = sim.vel.copy()
    fluid = (sim.info[:, 0] & 7) == 0
    vel[fluid, :3] += rng.uniform(-0.3, 0.3, size=(fluid.sum(), 3)).astype(np.float32)
    vel[:, 3] += rng.uniform(0, 2e-3, size=len(vel)).astype(np.float32)
    sim.vel = vel
    eng.vel[:n] = torch.from_numpy(vel[:n]).to(eng.device)
    return vel


def _check_neibs_and_forces(prob, seed, monkeypatch=None, tol=2e-5, switch_flips=0, kernels_ulp=0):
    eng = _engine(prob, clobber_neibslist=True)
    sim = ol.OracleSim(prob)
    sim.build_neibs(); eng.build_neibs()
    n = eng.n
    assert n == sim.n
    assert np.array_equal(_np(eng.hash, np.uint32)[:n], sim.hash[:n])
    assert np.array_equal(_np(eng.info, np.uint16)[:n], sim.info[:n])
    assert np.array_equal(_np(eng.pos)[:n].view(np.uint32), sim.pos[:n].view(np.uint32))
    assert np.array_equal(_np(eng.neibslist, np.uint16).reshape(-1, eng.alloc)[:, :n], sim.nl.reshape(-1, len(sim.pos))[:, :n])
    vel = _perturb(sim, eng, seed)
    cof = 1 if prob.simparams.numforcesbodies else 0
    f_ref, cfl_ref, nb, rbf_ref, _ = sim.o.forces(sim.pos, sim.vel, sim.info, sim.hash, sim.cs, sim.nl, n,
                                                compute_object_forces=cof, rb_count=prob.num_obstacle)
    eng._forces(eng.pos, eng.vel, 1, 0)
    f = _np(eng.forces)[:n]
    assert np.abs(f[:, :3] - f_ref[:n, :3]).max() <= tol * np.abs(f_ref[:, :3]).max()
    werr = np.abs(f[:, 3] - f_ref[:n, 3])
    wtol = tol * np.abs(f_ref[:, 3]).max() + 1e-7
    # the Colagrossi switch |P_i - P_j| >= |rho g.r| flips on the last bit of P for a pair sitting on it:
    # where a case is known to have such pairs a handful of particles may differ by ONE pair's diffusion term
    assert (werr > wtol).sum() <= switch_flips and werr.max() <= (1e-3 if switch_flips else tol) * np.abs(f_ref[:, 3]).max() + 1e-7
    dt_ref = sim.o.dtreduce(cfl_ref, nb, sim.sspeed_cfl, sim.max_kinvisc)
    assert abs(float(eng.d_dt_next.item()) - dt_ref) <= tol * dt_ref
    if monkeypatch is not None:      # same state through the other forces kernel
        import torch
        monkeypatch.setenv("SPHX_DISABLE_TILES", "1")
        eng_g = _engine(prob, clobber_neibslist=True)
        eng_g.build_neibs()
        eng_g.vel[:n] = torch.from_numpy(vel[:n]).to(eng_g.device)
        eng_g._forces(eng_g.pos, eng_g.vel, 1, 0)
        monkeypatch.setenv("SPHX_DISABLE_TILES", "0")
        fg = _np(eng_g.forces)[:n]
        from kernel_agreement import assert_forces_agree
        assert_forces_agree(f, fg)


@pytest.mark.parametrize("lin", sorted(D.LINEARIZATIONS))
def test_every_cell_linearisation(lin, monkeypatch):
    prob = DamBreak3D(deltap=0.045, obstacle=True, jitter=0.1, hydrostatic=False, linearization=lin)
    _check_neibs_and_forces(prob, 31, monkeypatch)


@pytest.mark.parametrize("kernel", [D.CUBICSPLINE, D.QUADRATIC, D.WENDLAND, D.GAUSSIAN])
def test_every_kernel(kernel, monkeypatch):
    prob = DamBreak3D(deltap=0.05, obstacle=False, jitter=0.1, hydrostatic=False, kerneltype=kernel)
    _check_neibs_and_forces(prob, 32, monkeypatch)


@pytest.mark.parametrize("turb,diff", [(D.LAMINAR_FLOW, D.COLAGROSSI), (D.ARTIFICIAL, D.DENSITY_DIFFUSION_NONE),
                                       (D.LAMINAR_FLOW, D.DENSITY_DIFFUSION_NONE), (D.ARTIFICIAL, D.FERRARI)])
def test_viscosity_and_diffusion_switches(turb, diff, monkeypatch):
    prob = DamBreak3D(deltap=0.045, obstacle=True, jitter=0.1, hydrostatic=False, density_diffusion=diff)
    prob.simparams.turbmodel = turb
    _check_neibs_and_forces(prob, 33, monkeypatch)


def test_ferrari_diffusion_with_lj_boundaries_and_trajectory():
    """Spheric2LJ's option set on the dam-break mirror: LJ_BOUNDARY + Ferrari density diffusion + artificial viscosity
    (generic forces kernel), forces and a 10-step trajectory; also two fluids (the Ferrari term crosses the interface)"""
    prob = DamBreak3D(deltap=0.045, obstacle=True, jitter=0.1, hydrostatic=True, boundary=D.LJ_BOUNDARY, density_diffusion=D.FERRARI)
    _check_neibs_and_forces(prob, 39)
    prob2 = DamBreak3D(deltap=0.045, obstacle=False, jitter=0.1, hydrostatic=False, two_fluids=True, density_diffusion=D.FERRARI)
    _check_neibs_and_forces(prob2, 40)
    # the term matters at this tolerance
    sim = ol.OracleSim(prob); sim.build_neibs()
    n = sim.n
    sim.vel[:n, 3] += np.random.default_rng(39).uniform(0, 2e-3, size=n).astype(np.float32)
    f1 = sim.o.forces(sim.pos, sim.vel, sim.info, sim.hash, sim.cs, sim.nl, n)[0]
    sim.o.p.densityDiffCoeff = 0.0
    f0 = sim.o.forces(sim.pos, sim.vel, sim.info, sim.hash, sim.cs, sim.nl, n)[0]
    assert np.abs(f1[:n, 3] - f0[:n, 3]).max() > 100 * 2e-5 * np.abs(f1[:n, 3]).max()
    eng = _engine(prob); sim = ol.OracleSim(prob)
    steps = 10
    for _ in range(steps):
        sim.step(); eng.step()
    out = eng.download()
    n = eng.n
    assert np.array_equal(out["hash"], sim.hash[:n])
    cs = float(min(prob.m_cellsize))
    assert np.abs(out["pos"][:, :3] - sim.pos[:n, :3]).max() <= 1e-6 * cs * steps
    assert np.abs(out["vel"][:, :3] - sim.vel[:n, :3]).max() <= 1e-3 * max(np.abs(sim.vel[:n, :3]).max(), 1e-6)
    assert np.abs(out["vel"][:, 3] - sim.vel[:n, 3]).max() <= 1e-6 * steps


def test_sph_f2_formulation():
    """SPH_F2 (Bubble, LockExchange, RTInstability in the reference: two fluids): generic kernel, forces + trajectory;
    and single fluid, where it must not disturb anything"""
    prob = DamBreak3D(deltap=0.045, obstacle=False, jitter=0.1, hydrostatic=False, two_fluids=True, formulation=D.SPH_F2)
    _check_neibs_and_forces(prob, 41, switch_flips=3)
    prob1 = DamBreak3D(deltap=0.045, obstacle=True, jitter=0.1, hydrostatic=True, formulation=D.SPH_F2)
    _check_neibs_and_forces(prob1, 42)
    # F2 differs from F1 for two fluids by much more than the tolerance
    sims = {}
    for form in (D.SPH_F1, D.SPH_F2):
        pr = DamBreak3D(deltap=0.045, obstacle=False, jitter=0.1, hydrostatic=False, two_fluids=True, formulation=form)
        sim = ol.OracleSim(pr); sim.build_neibs()
        sim.vel[:sim.n, 3] += np.random.default_rng(41).uniform(0, 2e-3, size=sim.n).astype(np.float32)
        sims[form] = sim.o.forces(sim.pos, sim.vel, sim.info, sim.hash, sim.cs, sim.nl, sim.n)[0][:sim.n]
    assert np.abs(sims[D.SPH_F1] - sims[D.SPH_F2])[:, :3].max() > 100 * 2e-5 * np.abs(sims[D.SPH_F1][:, :3]).max()
    eng = _engine(prob); sim = ol.OracleSim(prob)
    steps = 8
    for _ in range(steps):
        sim.step(); eng.step()
    out = eng.download()
    n = eng.n
    assert np.array_equal(out["hash"], sim.hash[:n])
    cs = float(min(prob.m_cellsize))
    assert np.abs(out["pos"][:, :3] - sim.pos[:n, :3]).max() <= 1e-6 * cs * steps
    assert np.abs(out["vel"][:, :3] - sim.vel[:n, :3]).max() <= 1e-3 * max(np.abs(sim.vel[:n, :3]).max(), 1e-6)


def test_mk_boundary_forces_and_trajectory():
    """MK_BOUNDARY: same lists, sections and feedback-body handling as LJ_BOUNDARY, Monaghan-Kajtar force law"""
    prob = DamBreak3D(deltap=0.045, obstacle=True, jitter=0.2, hydrostatic=False, boundary=D.MK_BOUNDARY)
    _check_neibs_and_forces(prob, 43, monkeypatch=None)
    sim = ol.OracleSim(prob); sim.build_neibs()
    f1 = sim.o.forces(sim.pos, sim.vel, sim.info, sim.hash, sim.cs, sim.nl, sim.n)[0]
    sim.o.p.MK_K = 0.0
    f0 = sim.o.forces(sim.pos, sim.vel, sim.info, sim.hash, sim.cs, sim.nl, sim.n)[0]
    assert np.abs(f1[:sim.n, :3] - f0[:sim.n, :3]).max() > 100 * 2e-5 * np.abs(f1[:sim.n, :3]).max()
    eng = _engine(prob); sim = ol.OracleSim(prob)
    steps = 8
    for _ in range(steps):
        sim.step(); eng.step()
    out = eng.download()
    n = eng.n
    assert np.array_equal(out["hash"], sim.hash[:n])
    cs = float(min(prob.m_cellsize))
    assert np.abs(out["pos"][:, :3] - sim.pos[:n, :3]).max() <= 1e-6 * cs * steps
    assert np.abs(out["vel"][:, :3] - sim.vel[:n, :3]).max() <= 1e-3 * max(np.abs(sim.vel[:n, :3]).max(), 1e-6)


def test_dem_terrain_forces_and_trajectory(monkeypatch):
    """ENABLE_DEM (DEMExample's option set: LJ_BOUNDARY + terrain height map + side planes): DemLJForce in the finalize step of
    both forces kernels; the water column is dropped onto the hills"""
    from test_dem_oracle import dem_problem
    prob = dem_problem(0.04)
    # lower the column into reach of the terrain so that the DEM term is exercised in the single-pass comparison
    prob.parts.pos_global[:, 2] -= 0.03
    _check_neibs_and_forces(prob, 47, monkeypatch)
    sim = ol.OracleSim(prob); sim.build_neibs()
    f1 = sim.o.forces(sim.pos, sim.vel, sim.info, sim.hash, sim.cs, sim.nl, sim.n)[0]
    sim.o.p.simflags &= ~D.ENABLE_DEM
    f0 = sim.o.forces(sim.pos, sim.vel, sim.info, sim.hash, sim.cs, sim.nl, sim.n)[0]
    assert (np.abs(f1[:sim.n, :3] - f0[:sim.n, :3]).max(axis=1) > 1.0).sum() > 30          # the terrain acts on the bottom layer
    prob = dem_problem(0.04)
    eng = _engine(prob); sim = ol.OracleSim(prob)
    steps = 24
    for _ in range(steps):
        sim.step(); eng.step()
    out = eng.download()
    n = eng.n
    assert np.array_equal(out["hash"], sim.hash[:n])
    cs = float(min(prob.m_cellsize))
    assert np.abs(out["pos"][:, :3] - sim.pos[:n, :3]).max() <= 1e-6 * cs * steps
    assert np.abs(out["vel"][:, :3] - sim.vel[:n, :3]).max() <= 1e-3 * max(np.abs(sim.vel[:n, :3]).max(), 1e-6)
    # the map can be dropped again (unsetDEM): the entry point then ignores the flag's term
    from gpusph_amd import capi
    capi.check(eng.lib.sphx_set_dem(eng.ctx.handle, None, 0, 0))


def test_two_fluids():
    """multi-fluid branch (generic kernel): per-fluid EOS, Colagrossi diffusion only between particles of the same fluid"""
    prob = DamBreak3D(deltap=0.045, obstacle=False, jitter=0.1, hydrostatic=False, two_fluids=True)
    assert prob.physparams.numFluids() == 2
    _check_neibs_and_forces(prob, 34, switch_flips=3)
    prob = DamBreak3D(deltap=0.045, obstacle=False, jitter=0.1, hydrostatic=False, two_fluids=True,
                      density_diffusion=D.DENSITY_DIFFUSION_NONE)
    _check_neibs_and_forces(prob, 35)


VISC_FLAVOURS = [
    dict(compvisc=D.KINEMATIC, avgop=D.ARITHMETIC, is_const_visc=True),     # DYNAMICVISC of a single fluid
    dict(compvisc=D.KINEMATIC, avgop=D.HARMONIC, is_const_visc=True),       # KINEMATICVISC
    dict(compvisc=D.KINEMATIC, avgop=D.GEOMETRIC, is_const_visc=True),
    dict(compvisc=D.DYNAMIC, avgop=D.ARITHMETIC, is_const_visc=True),
    dict(compvisc=D.KINEMATIC, avgop=D.ARITHMETIC, is_const_visc=False),
    dict(compvisc=D.KINEMATIC, avgop=D.HARMONIC, is_const_visc=False),
    dict(compvisc=D.DYNAMIC, avgop=D.GEOMETRIC, is_const_visc=False),
    dict(compvisc=D.DYNAMIC, avgop=D.HARMONIC, is_const_visc=False),
]


@pytest.mark.parametrize("k", range(len(VISC_FLAVOURS)))
def test_newtonian_laminar_viscosity(k, monkeypatch):
    """NEWTONIAN + LAMINAR_FLOW + MORRIS in every averaging flavour of visc_avg: single fluid with a feedback body
    (tiled and generic kernels, bit-equal to each other), then two fluids of different viscosity (generic kernel).
    The two-fluid non-constant cases are the ones that caught a wave-uniform-branch version of the viscosity factor
    going wrong after the first list batch of the boundary section."""
    spec = dict(rheologytype=D.NEWTONIAN, turbmodel=D.LAMINAR_FLOW, **VISC_FLAVOURS[k])
    prob = DamBreak3D(deltap=0.045, obstacle=True, jitter=0.1, hydrostatic=False, viscosity=spec, kinematic_visc=0.05)
    _check_neibs_and_forces(prob, 36, monkeypatch)
    prob2 = DamBreak3D(deltap=0.045, obstacle=False, jitter=0.1, hydrostatic=False, viscosity=spec, kinematic_visc=0.05,
                       two_fluids=True, density_diffusion=D.DENSITY_DIFFUSION_NONE)
    _check_neibs_and_forces(prob2, 37)
    # the check above is sensitive to the term: halving the viscosities moves the oracle's forces by >> the tolerance
    sim = ol.OracleSim(prob2)
    sim.build_neibs()
    n = sim.n
    rng = np.random.default_rng(37)
    fluid = (sim.info[:, 0] & 7) == 0
    sim.vel[fluid, :3] += rng.uniform(-0.3, 0.3, size=(fluid.sum(), 3)).astype(np.float32)
    f1 = sim.o.forces(sim.pos, sim.vel, sim.info, sim.hash, sim.cs, sim.nl, n)[0]
    sim.o.p.visccoeff[0] *= 0.5; sim.o.p.visccoeff[1] *= 0.5
    f2 = sim.o.forces(sim.pos, sim.vel, sim.info, sim.hash, sim.cs, sim.nl, n)[0]
    assert np.abs(f1[:n, :3] - f2[:n, :3]).max() > 100 * 2e-5 * np.abs(f1[:n, :3]).max()


def test_newtonian_plane_friction_and_viscous_dt():
    """KINEMATICVISC against geometric planes (wall friction in finalize) and the viscous time-step limit"""
    prob = DamBreak3D(deltap=0.045, obstacle=False, jitter=0.3, hydrostatic=False, boundary=D.LJ_BOUNDARY, walls="planes",
                      viscosity="KINEMATICVISC", kinematic_visc=1.0)
    prob.physparams.partsurf = 3.0 * prob.m_deltap ** 2
    _check_neibs_and_forces(prob, 38)
    # the viscous limit 0.125 h^2/nu of dtreduce (src/cuda/forces.cu:586-601) on a quiet CFL buffer
    import ctypes as C
    import torch
    from gpusph_amd import capi
    eng = _engine(prob)
    h = float(eng.params.slength)
    cfl = torch.full((8,), 1.0e-3, dtype=torch.float32, device=eng.device)
    tmp = torch.zeros(8, dtype=torch.float32, device=eng.device)
    dt = C.c_float(0.0)
    capi.check(eng.lib.sphx_forces_dtreduce(eng.ctx.handle, h, eng.params.dtadaptfactor, eng.sspeed_cfl, eng.max_kinvisc,
                                            capi.ptr(cfl), capi.ptr(tmp), 8, C.byref(dt), None))
    assert eng.max_kinvisc == pytest.approx(1.0)
    assert dt.value == pytest.approx(0.125 * h * h / 1.0, rel=1e-6) and dt.value < 0.3 * h / eng.sspeed_cfl
    prob.physparams.partsurf = 0.0
    sim = ol.OracleSim(prob)
    sim.build_neibs()
    n = sim.n
    rng = np.random.default_rng(38)
    sim.vel[:n, :3] = rng.uniform(-0.3, 0.3, size=(n, 3)).astype(np.float32)
    fA = sim.o.forces(sim.pos, sim.vel, sim.info, sim.hash, sim.cs, sim.nl, n)[0]
    sim.o.p.partsurf = 3.0 * prob.m_deltap ** 2
    fB = sim.o.forces(sim.pos, sim.vel, sim.info, sim.hash, sim.cs, sim.nl, n)[0]
    assert np.abs(fA[:n, :3] - fB[:n, :3]).max() > 100 * 2e-5 * np.abs(fA[:n, :3]).max()   # the friction term matters here


def test_viscous_shear_wave_trajectory():
    """20 steps of a decaying shear wave in a periodic box, KINEMATICVISC without artificial viscosity"""
    from gpusph_amd.problem import PeriodicBox
    import torch
    prob = PeriodicBox(deltap=0.05, n=(16, 24, 12), jitter=0.05, viscosity="KINEMATICVISC", kinematic_visc=0.05)
    eng = _engine(prob); sim = ol.OracleSim(prob)
    k = 2 * np.pi / prob.m_size[1]
    u = (0.5 * np.sin(k * prob.parts.pos_global[:, 1])).astype(np.float32)
    sim.vel[:, 0] = u
    eng.vel[: len(u), 0] = torch.from_numpy(u).to(eng.device)
    steps = 20
    for _ in range(steps):
        sim.step(); eng.step()
    out = eng.download()
    n = eng.n
    assert np.array_equal(out["hash"], sim.hash[:n])
    cs = float(min(prob.m_cellsize))
    assert np.abs(out["pos"][:, :3] - sim.pos[:n, :3]).max() <= 1e-6 * cs * steps
    assert np.abs(out["vel"][:, :3] - sim.vel[:n, :3]).max() <= 1e-3 * np.abs(sim.vel[:n, :3]).max()
    g = prob.global_pos(out["pos"], out["hash"])
    amp = 2.0 * np.mean(out["vel"][:, 0] * np.sin(k * g[:, 1]))
    assert 0.5 * np.exp(-1.3 * 0.05 * k * k * eng.time()) < amp < 0.5 * np.exp(-0.7 * 0.05 * k * k * eng.time())


def test_spsvisc_trajectory_with_shepard_filter():
    """WaveTank's option set on the dam-break mirror: viscosity<SPSVISC> (nu = 1e-6), Shepard filter, a moving body;
    CALC_VISC runs before each forces pass (PredictorCorrectorIntegrator.cc:460-480)"""
    prob = DamBreak3D(deltap=0.04, obstacle=True, jitter=0.1, viscosity="SPSVISC", kinematic_visc=1.0e-6)
    assert prob.simparams.turbmodel == D.SPS and prob.simparams.rheologytype == D.NEWTONIAN
    assert prob.physparams.smagfactor == pytest.approx((0.12 * 0.04) ** 2, rel=1e-6)
    eng = _engine(prob); sim = ol.OracleSim(prob)
    eng.add_filter(D.SHEPARD_FILTER, 4); sim.filters = [(D.SHEPARD_FILTER, 4)]
    steps = 12
    for _ in range(steps):
        sim.step(); eng.step()
    out = eng.download()
    n = eng.n
    assert np.array_equal(out["hash"], sim.hash[:n])
    cs = float(min(prob.m_cellsize))
    assert np.abs(out["pos"][:, :3] - sim.pos[:n, :3]).max() <= 1e-6 * cs * steps
    assert np.abs(out["vel"][:, :3] - sim.vel[:n, :3]).max() <= 1e-3 * np.abs(sim.vel[:n, :3]).max()
    assert np.abs(out["vel"][:, 3] - sim.vel[:n, 3]).max() <= 1e-6 * steps
    assert abs(eng.current_dt() - sim.dt) <= 2e-5 * sim.dt


def test_euler_with_a_moving_rigid_body():
    """eulerDevice for particles of a moving object: position = rotation about the centre of gravity + translation,
    velocity = linear + angular x arm (src/cuda/euler_kernel.def:470-520, applyrot src/cuda/euler_kernel.cu:67-74)"""
    import torch
    from gpusph_amd import capi
    prob = DamBreak3D(deltap=0.04, obstacle=True)
    eng = _engine(prob)
    sim = ol.OracleSim(prob)
    sim.build_neibs(); eng.build_neibs()
    n = eng.n
    th = 0.01
    rot = np.array([np.cos(th), -np.sin(th), 0, np.sin(th), np.cos(th), 0, 0, 0, 1], dtype=np.float32)
    trans = np.array([1e-3, -2e-3, 5e-4], dtype=np.float32)
    lvel = np.array([0.3, -0.1, 0.05], dtype=np.float32)
    avel = np.array([0.0, 0.2, 1.5], dtype=np.float32)
    capi.check(eng.lib.sphx_set_rb_motion(eng.ctx.handle, trans.ctypes.data, rot.ctypes.data, lvel.ctypes.data, avel.ctypes.data, 1))
    for a in range(3):
        sim.o.p.rbtrans[0][a] = float(trans[a]); sim.o.p.rblinearvel[0][a] = float(lvel[a]); sim.o.p.rbangularvel[0][a] = float(avel[a])
    for a in range(9):
        sim.o.p.rbsteprot[0][a] = float(rot[a])
    rng = np.random.default_rng(5)
    forces = rng.normal(0, 5, size=(len(sim.pos), 4)).astype(np.float32)
    eng.forces[:n] = torch.from_numpy(forces[:n]).to(eng.device)
    dt = float(np.float32(2.7e-4))
    eng.d_dt.fill_(dt)
    body = (sim.info[:n, 0] & D.FG_MOVING_BOUNDARY) != 0
    assert body.sum() == prob.num_obstacle > 0
    for step, scale in ((1, 0.5), (2, 1.0)):
        pr, vr = sim.o.euler(sim.pos, sim.vel, sim.info, sim.hash, forces, n, float(np.float32(dt) * np.float32(scale)), step)
        eng._euler(step, scale)
        assert np.array_equal(_np(eng.pos2)[:n].view(np.uint32), pr[:n].view(np.uint32))
        assert np.array_equal(_np(eng.vel2)[:n].view(np.uint32), vr[:n].view(np.uint32))
        assert np.abs(vr[:n][body, :3]).max() > 0.1            # the body rows did move with the prescribed motion


def test_neighbour_list_overflow_is_reported():
    """a list that is too short: overflow is reported through getinfo (hasTooManyNeibs / hasMaxNeibs,
    src/cuda/buildneibs.cu:137-145), not as an error of the call; the host aborts the run on it (CHECK_NEIBSNUM).
    Particles whose list did fit are stored exactly as by the oracle.  The slots of a particle whose fluid and
    boundary sections collided are unspecified here: the reference resolves the collision in program order, the
    LDS-staged build in flush order, and nobody reads such a list."""
    from gpusph_amd import capi
    prob = DamBreak3D(deltap=0.045, obstacle=False, jitter=0.1, hydrostatic=False)
    prob.simparams.neiblistsize = 64
    prob.simparams.neibboundpos = 63
    eng = _engine(prob, clobber_neibslist=True)
    sim = ol.OracleSim(prob)
    sim.build_neibs(); eng.build_neibs()
    n = eng.n
    assert sim.neibs_info.hasTooManyNeibs >= 0
    gl = _np(eng.neibslist, np.uint16).reshape(-1, eng.alloc)[:, :n]
    ol_ = sim.nl.reshape(-1, len(sim.pos))[:, :n]
    assert gl.shape[0] == 64
    stored = (ol_ != 0xFFFF).sum(axis=0)
    fits = stored <= 63 - 2                       # both terminators had room: no collision
    assert fits.sum() > 0.5 * n and (~fits).sum() > 10
    assert np.array_equal(gl[:, fits], ol_[:, fits])
    with pytest.raises(capi.SphxError):
        eng.neibs_info()                      # CHECK_NEIBSNUM in the driver turns the report into an error
    info = eng.last_neibs_info
    assert info.hasTooManyNeibs >= 0
    assert info.hasMaxNeibs[0] + info.hasMaxNeibs[1] >= 63
    assert info.numInteractions == sim.neibs_info.numInteractions
    assert info.maxFluidBoundaryNeibs == sim.neibs_info.maxFluidBoundaryNeibs


def test_body_with_prescribed_motion_trajectory():
    """MOVE_BODIES: the engine's host kinematics (gpusph_amd/bodies.py) + separate forces / integration centres of gravity,
    against the oracle driver running the same callback; 14 steps spanning a rebuild, a body with force feedback"""
    from test_oracle_physics import _gate_callback
    prob = DamBreak3D(deltap=0.04, obstacle=True, jitter=0.05, hydrostatic=True)
    prob.moving_bodies_callback = _gate_callback(2.0, 60.0, 2.0)
    eng = _engine(prob); sim = ol.OracleSim(prob)
    steps = 14
    for _ in range(steps):
        sim.step(); eng.step()
    out = eng.download()
    n = eng.n
    assert np.array_equal(out["hash"], sim.hash[:n]) and np.array_equal(out["info"], sim.info[:n])
    body = (out["info"][:, 0] & D.FG_MOVING_BOUNDARY) != 0
    assert np.array_equal(out["pos"][body].view(np.uint32), sim.pos[:n][body].view(np.uint32))     # rigid rows: bit-exact
    assert np.array_equal(out["vel"][body, :3].view(np.uint32), sim.vel[:n][body, :3].view(np.uint32))
    cs = float(min(prob.m_cellsize))
    assert np.abs(out["pos"][:, :3] - sim.pos[:n, :3]).max() <= 1e-6 * cs * steps
    assert np.abs(out["vel"][:, :3] - sim.vel[:n, :3]).max() <= 1e-3 * np.abs(sim.vel[:n, :3]).max()
    assert abs(eng.time() - sim.t) <= 1e-6 * sim.t


def test_wavetank_mirror_trajectory():
    """BASELINE configs[4]'s option set in one run: LJ box particles + six planes (sloping beach), viscosity<SPSVISC> in the
    tiled kernel, Shepard filter, hinged paddle driven by the problem's callback -- 24 steps against the oracle driver"""
    from gpusph_amd.problem import WaveTank
    prob = WaveTank(0.03, paddle_tstart=0.0)
    eng = _engine(prob); sim = ol.OracleSim(prob)
    eng.add_filter(D.SHEPARD_FILTER, 10); sim.filters = [(D.SHEPARD_FILTER, 10)]
    steps = 24
    for _ in range(steps):
        sim.step(); eng.step()
    out = eng.download()
    n = eng.n
    assert n == sim.n and np.array_equal(out["hash"], sim.hash[:n]) and np.array_equal(out["info"], sim.info[:n])
    pad = (out["info"][:, 0] & D.FG_MOVING_BOUNDARY) != 0
    assert pad.sum() > 100
    assert np.array_equal(out["pos"][pad].view(np.uint32), sim.pos[:n][pad].view(np.uint32))      # rigid rows bit-exact
    cs = float(min(prob.m_cellsize))
    assert np.abs(out["pos"][:, :3] - sim.pos[:n, :3]).max() <= 1e-6 * cs * steps
    assert np.abs(out["vel"][:, :3] - sim.vel[:n, :3]).max() <= 1e-3 * np.abs(sim.vel[:n, :3]).max()
    assert np.abs(out["vel"][:, 3] - sim.vel[:n, 3]).max() <= 1e-6 * steps
    assert abs(eng.current_dt() - sim.dt) <= 2e-5 * sim.dt


@pytest.mark.parametrize("use_planes", [False, True])
def test_stillwater_mirror_trajectory(use_planes):
    """The StillWater mirror (BASELINE configs[2]'s problem with the boundary models built here): DYNAMICVISC + Ferrari
    diffusion in the tiled kernel, DYN walls (or planes), MLS filter -- 24 steps against the oracle driver across a
    neighbour-list rebuild"""
    from gpusph_amd.problem import StillWater
    prob = StillWater(10, use_planes=use_planes, jitter=0.05)
    eng = _engine(prob); sim = ol.OracleSim(prob)
    eng.add_filter(D.MLS_FILTER, 6); sim.filters = [(D.MLS_FILTER, 6)]
    steps = 24
    for _ in range(steps):
        sim.step(); eng.step()
    out = eng.download()
    n = eng.n
    assert n == sim.n and np.array_equal(out["hash"], sim.hash[:n]) and np.array_equal(out["info"], sim.info[:n])
    cs = float(min(prob.m_cellsize))
    assert np.abs(out["pos"][:, :3] - sim.pos[:n, :3]).max() <= 1e-6 * cs * steps
    vs = max(np.abs(sim.vel[:n, :3]).max(), 1e-3)
    assert np.abs(out["vel"][:, :3] - sim.vel[:n, :3]).max() <= 1e-3 * vs
    assert np.abs(out["vel"][:, 3] - sim.vel[:n, 3]).max() <= 1e-6 * steps
    assert abs(eng.current_dt() - sim.dt) <= 2e-5 * sim.dt


def test_interface_detection_bit_exact():
    """INTERFACE_DETECTION post-processing (calcInterfaceparticleDevice) on a jittered two-fluid column with a feedback body:
    flags and normals equal the oracle's bit for bit (polynomial kernel, IEEE division and sqrt, the reference's order)"""
    import torch
    prob = DamBreak3D(deltap=0.03, obstacle=True, jitter=0.2, hydrostatic=True, two_fluids=True)
    eng = _engine(prob); sim = ol.OracleSim(prob)
    eng.build_neibs(); sim.build_neibs()
    n = eng.n
    ref_info, ref_nrm = sim.o.interface(sim.pos, sim.vel, sim.info, sim.hash, sim.cs, sim.nl, n, normals=True)
    nrm = _np(eng.postprocess(D.INTERFACE_DETECTION, normals=True))
    got = _np(eng.info, np.uint16)[:n]
    assert np.array_equal(got, ref_info[:n])
    assert ((got[:, 0] & D.FG_INTERFACE) != 0).sum() > 100 and ((got[:, 0] & D.FG_SURFACE) != 0).sum() > 100
    assert np.array_equal(nrm.view(np.uint32), ref_nrm[:n].view(np.uint32))
    # running it again on the flagged INFO is idempotent (the pass clears both flags first)
    eng.postprocess(D.INTERFACE_DETECTION)
    assert np.array_equal(_np(eng.info, np.uint16)[:n], ref_info[:n])


def test_xsph_mean_velocity_and_trajectory():
    """ENABLE_XSPH (rows a9/a15): the forces pass writes 2 * mean neighbourhood velocity for the fluid particles
    (compute_mean_vel forces_kernel.def:2986-2994), Euler advects with v + eps * that (compute_corrected_velocity
    euler_kernel.def:171-180).  XSPH array bit-exact vs the oracle, then a trajectory with the usual tolerances"""
    import torch
    prob = DamBreak3D(deltap=0.04, obstacle=True, jitter=0.05, hydrostatic=False)
    prob.simparams.simflags |= D.ENABLE_XSPH
    eng = _engine(prob, clobber_neibslist=True)
    sim = ol.OracleSim(prob)
    sim.build_neibs(); eng.build_neibs()
    n = eng.n
    rng = np.random.default_rng(5)
    vel = sim.vel.copy()
    fluid = (sim.info[:, 0] & 7) == 0
    vel[fluid, :3] += rng.uniform(-0.3, 0.3, size=(int(fluid.sum()), 3)).astype(np.float32)
    vel[:, 3] += rng.uniform(0, 2e-3, size=len(vel)).astype(np.float32)
    sim.vel = vel
    eng.vel[:n] = torch.from_numpy(vel[:n]).to(eng.device)
    xs_ref = sim.o.xsph(sim.pos, sim.vel, sim.info, sim.hash, sim.cs, sim.nl, n)
    eng._forces(eng.pos, eng.vel, 1, 0)
    xs = eng.xsph[:n].cpu().numpy()
    assert np.abs(xs_ref[:n][fluid[:n], :3]).max() > 1e-3
    assert np.array_equal(xs.view(np.uint32), xs_ref[:n].view(np.uint32))
    assert not xs[~fluid[:n]].any()                     # boundary rows are never written
    # the correction changes where particles go: same run without the flag differs, with the flag follows the oracle
    for _ in range(12):
        sim.step(); eng.step()
    out = eng.download()
    same = (out["info"] == sim.info[:n]).all(axis=1)
    assert same.mean() > 0.999
    m = same
    assert np.abs(out["pos"][m, :3] - sim.pos[:n][m, :3]).max() <= 12e-6 * prob.m_cellsize.min()
    vscale = max(np.abs(sim.vel[:n, :3]).max(), 1e-3)
    assert np.abs(out["vel"][m, :3] - sim.vel[:n][m, :3]).max() <= 1e-3 * vscale
    plain = DamBreak3D(deltap=0.04, obstacle=True, jitter=0.05, hydrostatic=False)
    e2 = _engine(plain)
    e2.vel[:n] = torch.from_numpy(vel[:n]).to(e2.device)
    e3 = _engine(prob)
    e3.vel[:n] = torch.from_numpy(vel[:n]).to(e3.device)
    e2.step(); e3.step()
    assert np.abs(e2.download()["pos"][:, :3] - e3.download()["pos"][:, :3]).max() > 1e-7
