"""SA_BOUNDARY on the GPU (data path + boundary-conditions engine of solid walls) against the CPU oracle:
the neighbour phase with the SA buffers bit for bit, the boundary-conditions kernels to the tolerance of the math library."""
import ctypes as C
import numpy as np
import pytest

from gpusph_amd import defs as D
from gpusph_amd.problem import SABox, DamBreak3D, info_type
from sa_helpers import sa_oracle_state, assert_close_but_for_gamma_spikes, wall_rows

pytestmark = pytest.mark.gpu


# The run comparisons below (several steps of an engine against the oracle's sequence): measured on an MI355X with
# SPHX_TEST_REPORT (profiles/r06_sa_run_comparisons.txt), in units of the tolerance of each line: velocities after 6 steps worst
# 0.25 - 0.81, densities 0.06 - 0.53, every row WITHOUT a boundary element in reach <= 0.23.  So: no spike allowance to speak of
# (twice the tolerance for at most 2 per mille of the entries), and the plain tolerance for the rows away from the walls
AWAY = 1.0
RUN = dict(frac=0.002, spike=2.0)


def _engine(problem, **kw):
    import torch
    from gpusph_amd.engine import TimestepEngine
    assert torch.cuda.is_available()
    return TimestepEngine(problem, device="cuda:0", **kw)


def _tiles_usable(eng):
    """Whether the neighbour list built last left a usable tiling (then the tiled kernels form the particle <- particle sums)."""
    return int(eng.k.lib.sphx_dbg_tiles_usable(eng.k.ctx.handle))


@pytest.fixture(params=["tiled", "list-walkers"])
def kernels(request, monkeypatch):
    """Both ways the SA engines form their sums: tiled window + one boundary element per lane, or one thread per particle
    walking the whole list (the CPU oracle's mirror).  SPHX_DISABLE_TILES is read when a context is created."""
    monkeypatch.setenv("SPHX_DISABLE_TILES", "0" if request.param == "tiled" else "1")
    return request.param


def _np(t, dtype=None):
    a = t.cpu().numpy()
    return a.view(dtype) if dtype is not None else a


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.fixture(scope="module", params=[dict(deltap=0.05), dict(deltap=0.04, jitter=0.2, linearization="xzy")])
def pair(request):
    st = sa_oracle_state(**request.param)
    eng = _engine(SABox(**request.param), clobber_neibslist=True)
    eng.build_neibs()
    return st, eng


def test_neighbour_phase_with_sa_buffers_is_bit_exact(pair):
    st, eng = pair
    n = st["n"]
    assert eng.n == n
    assert np.array_equal(_np(eng.hash, np.uint32)[:n], st["hash"])
    assert np.array_equal(_np(eng.info, np.uint16)[:n], st["info"])
    assert np.array_equal(_np(eng.cellStart, np.uint32), st["cs"]) and np.array_equal(_np(eng.cellEnd, np.uint32), st["ce"])
    assert np.array_equal(_bits(_np(eng.pos)[:n]), _bits(st["pos"]))
    # the optional arrays of the re-sort: vertex ids, element normals/areas, gamma (NaN rows included)
    assert np.array_equal(_np(eng.vertices, np.uint32)[:n], st["vertices"])
    assert np.array_equal(_bits(_np(eng.boundelements)[:n]), _bits(st["boundelements"]))
    assert np.array_equal(_bits(_np(eng.gradgamma)[:n]), _bits(st["gradgamma"]))
    # all three sections of the list, the in-plane vertex offsets of the segments, the counters
    A = eng.alloc
    nl = _np(eng.neibslist, np.uint16).reshape(-1, A)[:, :n]
    assert np.array_equal(nl, st["nl"].reshape(-1, n))
    for k in range(3):
        assert np.array_equal(_bits(_np(eng.vertpos[k])[:n]), _bits(st["vertpos"][k])), k
    info = eng.neibs_info()
    want = st["neibs_info"]
    assert (info.numInteractions, info.maxFluidBoundaryNeibs, info.maxVertexNeibs, info.hasTooManyNeibs) == \
        (want.numInteractions, want.maxFluidBoundaryNeibs, want.maxVertexNeibs, -1)
    assert info.maxVertexNeibs > 10


def test_initialisation_sequence_of_the_boundary_conditions(pair):
    """SA_COMPUTE_VERTEX_NORMAL, SA_INIT_GAMMA, segment and vertex boundary conditions of step 0"""
    st, eng = pair
    o, p, n = st["oracle"], st["problem"], st["n"]
    t = info_type(st["info"])
    fl, seg, vx = (np.where(t == k)[0] for k in (D.PT_FLUID, D.PT_BOUNDARY, D.PT_VERTEX))
    be = o.sa_compute_vertex_normal(st["boundelements"], st["vertices"], st["info"], st["hash"], st["cs"], st["nl"], n)
    gg0 = o.sa_init_gamma(st["gradgamma"], st["pos"], be, st["vertpos"], st["info"], st["hash"], st["cs"], st["nl"], n, p.m_deltap)
    vel1, gg1 = o.sa_segment_bc(st["pos"], st["vel"], gg0, st["vertices"], be, st["info"], st["hash"], st["cs"], st["nl"], n, step=0)
    vel2 = o.sa_vertex_bc(st["pos"], vel1, gg1, st["info"], st["hash"], st["cs"], st["nl"], n)

    eng.sa_boundary_conditions(0)
    gbe, ggg, gvel = _np(eng.boundelements)[:n], _np(eng.gradgamma)[:n], _np(eng.vel)[:n]
    assert np.array_equal(_bits(gbe), _bits(be))                                   # vertex normals: same arithmetic
    # gamma: the same formulas through another math library (atan2f, acoshf): |grad gamma| ~ 10, gamma ~ 1
    scale = np.abs(gg0[np.concatenate([fl, vx]), :3]).max()
    for rows in (fl, vx, seg):
        assert_close_but_for_gamma_spikes(ggg[rows, :3], gg1[rows, :3], 2e-5, scale, what="grad gamma")
        assert np.abs(ggg[rows, 3] - gg1[rows, 3]).max() < 5e-6
    # wall densities (Tait equation inverted: powf twice)
    rho_scale = np.abs(vel2[:, 3]).max()
    assert np.abs(gvel[:, 3] - vel2[:, 3]).max() < 2e-5 * rho_scale + 2e-7
    assert np.array_equal(_bits(gvel[:, :3]), _bits(vel2[:, :3]))
    assert np.array_equal(_bits(gvel[fl]), _bits(st["vel"][fl]))                   # fluid rows untouched
    wet = seg[vel2[seg, 3] > 0]
    assert len(wet) > 200


def test_later_steps_and_repacking_mode(pair):
    st, eng = pair
    o, n = st["oracle"], st["n"]
    t = info_type(st["info"])
    seg = np.where(t == D.PT_BOUNDARY)[0]
    # the state the previous test left on the device is the input of both sides
    vel, gg, be = _np(eng.vel)[:n].copy(), _np(eng.gradgamma)[:n].copy(), _np(eng.boundelements)[:n].copy()
    rng = np.random.default_rng(9)
    fl = np.where(t == D.PT_FLUID)[0]
    vel[fl, 3] *= (1 + 0.05 * rng.standard_normal(len(fl))).astype(np.float32)    # a perturbed density field
    import torch
    eng.vel[:n] = torch.from_numpy(vel).to(eng.device)
    for step, repack in ((1, False), (2, False), (1, True)):
        v1, g1 = o.sa_segment_bc(st["pos"], vel, gg, st["vertices"], be, st["info"], st["hash"], st["cs"], st["nl"], n, step=step, repack=repack)
        v2 = o.sa_vertex_bc(st["pos"], v1, g1, st["info"], st["hash"], st["cs"], st["nl"], n)
        eng.sa_boundary_conditions(step, run_mode=D.REPACK if repack else D.SIMULATE)
        gvel, ggg = _np(eng.vel)[:n], _np(eng.gradgamma)[:n]
        assert np.array_equal(_bits(ggg), _bits(gg))                    # finite gamma, no moving bodies: left alone
        assert np.abs(gvel[:, 3] - v2[:, 3]).max() < 2e-5 * np.abs(v2[:, 3]).max() + 2e-7
        assert np.array_equal(_bits(gvel[:, :3]), _bits(v2[:, :3]))
        if step == 1 and not repack:
            assert np.abs(gvel[seg, 3] - vel[seg, 3]).max() > 1e-5       # it did respond to the perturbed field
        vel = gvel.copy()


def test_what_is_not_built_says_so(pair):
    from gpusph_amd import capi
    st, eng = pair
    # open boundaries: the driver has the sequence (tests/test_engine_sa_io.py, tests/test_gpu_sa_io.py) for a problem that
    # brings its imposed values and rebuilds in every iteration; a problem that only sets the flag is refused by the driver
    io = SABox(deltap=0.05)
    io.simparams.simflags |= D.ENABLE_INLET_OUTLET
    with pytest.raises((NotImplementedError, ValueError)):
        _engine(io)
    # and the boundary-conditions engine refuses a framework without SA_BOUNDARY, like the reference's SFINAE'd implementation
    other = _engine(DamBreak3D(deltap=0.05, obstacle=False))
    other.build_neibs()
    k = other.k
    with pytest.raises(capi.SphxInvalidArgument, match="without SA_BOUNDARY"):
        k.sa_vertex_bc(other.vel, other.vel, other.pos, other.info, other.hash, other.cellStart, other.neibslist, other.n, other.n, 1)


def test_cpp_adapters_run_the_sa_initialisation(tmp_path):
    """The engines the GPUSPH tree would load (HIPNeibsEngine, HIPBoundaryConditionsEngine behind a framework built by
    StillWaterSA's own SETUP_FRAMEWORK expression), driven through the abstract interfaces with BufferLists: neighbour phase
    with the SA buffers and the initialisation sequence of the boundary conditions, bit-equal to the Python driver."""
    import subprocess
    import host_case as hc
    prob = SABox(deltap=0.05, jitter=0.1)
    eng = _engine(prob, clobber_neibslist=True)
    n = prob.num_particles
    case = tmp_path / "case.txt"
    case.write_text("\n".join(hc.case_lines(prob, "StillWaterSA", allocated=eng.alloc) + hc.driver_lines(prob, eng, 0)) + "\n")
    hc.write_state(str(tmp_path / "state.bin"), prob.copy_to_array())
    subprocess.check_call([hc.exe("example_engines"), str(case), str(tmp_path / "state.bin"), str(tmp_path / "out.bin")])
    out = hc.read_out(str(tmp_path / "out.bin"))
    eng.build_neibs()
    eng.sa_boundary_conditions(0)
    assert out["n"] == n == eng.n
    for name, dt in (("pos", None), ("vel", None), ("boundelements", None), ("gradgamma", None)):
        assert np.array_equal(_bits(out[name]), _bits(_np(getattr(eng, name))[:n])), name
    assert np.array_equal(out["vertices"], _np(eng.vertices, np.uint32)[:n])
    assert np.array_equal(out["info"], _np(eng.info, np.uint16)[:n]) and np.array_equal(out["hash"], _np(eng.hash, np.uint32)[:n])
    for k in range(3):
        assert np.array_equal(_bits(out["vertpos"][k]), _bits(_np(eng.vertpos[k])[:n]))
    info = eng.neibs_info()
    assert tuple(out["counters"][:3]) == (info.numInteractions, info.maxFluidBoundaryNeibs, info.maxVertexNeibs)
    assert out["counters"][3] == n                      # no particle was created
    t = info_type(out["info"])
    assert np.isfinite(out["gradgamma"][t != D.PT_BOUNDARY]).all() and (out["vel"][t == D.PT_BOUNDARY, 3] > 0).sum() > 200


def test_sa_forces_gamma_integration_and_trajectory(kernels):
    """The SA forces engine (fluid, vertex and boundary-element terms, division by gamma), gamma by quadrature at new positions
    and the whole predictor-corrector sequence against the CPU oracle (option set of StillWaterRepackSA's simulation)."""
    from sa_helpers import OracleSaSim
    import torch
    kw = dict(deltap=0.05, jitter=0.15, options="StillWaterRepackSA")
    sim = OracleSaSim(SABox(**kw))
    eng = _engine(SABox(**kw), clobber_neibslist=True)
    eng.build_neibs()
    assert _tiles_usable(eng) == (kernels == "tiled")
    eng.sa_boundary_conditions(0)
    n, o, k = sim.n, sim.o, eng.k
    t = info_type(sim.info)
    fl = np.where(t == D.PT_FLUID)[0]
    # one forces evaluation on identical inputs (the oracle's initialised state uploaded)
    for name, arr in (("vel", sim.vel), ("gradgamma", sim.gg), ("boundelements", sim.be)):
        getattr(eng, name)[:n] = torch.from_numpy(arr).to(eng.device)
    f, cfl, nb = o.forces_sa(sim.pos, sim.vel, sim.info, sim.hash, sim.cs, sim.nl, sim.gg, sim.be, sim.vertpos, n, sim.problem.m_deltap)
    eng.k.memset(eng.cfl, 0)
    gnb = k.forces_sa(eng.forces, eng.cfl, eng.pos, eng.vel, eng.info, eng.hash, eng.cellStart, eng.neibslist, eng.gradgamma,
                      eng.boundelements, eng.vertpos, n, 0, n, 0)
    gf = _np(eng.forces)[:n]
    assert gnb == nb
    # tolerance: the state is hydrostatic, i.e. every force is the small remainder of pressure terms ~50x its size, and the
    # pressures come from powf of two math libraries (1 ulp of (1+rho~)^7 is 1e-5 of P): 1e-4 of the largest force
    scale = np.abs(f[fl, :3]).max()
    wall = wall_rows(sim.problem, sim.nl, sim.info, n)      # the gamma allowance is for particles next to a wall only
    assert 0.05 < wall[fl].mean() < 0.95
    # no fraction of outliers, no multiple of the tolerance: a row beyond 1e-4 of the scale must have boundary elements in reach and
    # lie, like the oracle's value, within the room that the conditioning of |grad gamma_as| of THOSE elements leaves around the
    # float64 value of the row's boundary terms (tests/sa_helpers.py assert_wall_rows_no_farther_from_float64)
    from sa_helpers import assert_wall_rows_no_farther_from_float64
    sw = max(np.abs(f[fl, 3]).max(), 1e-3)
    errn = np.abs(gf[fl].astype(np.float64) - f[fl])/np.array([scale, scale, scale, sw])
    assert errn[~wall[fl]].max() <= 1e-4, "a row with no boundary element in reach is %g of the scale off" % errn[~wall[fl]].max()
    arbitrated, total = assert_wall_rows_no_farther_from_float64(sim, gf, f, fl, 1e-4, scale, sw, what="SA forces")
    print("SA forces: %d of %d fluid rows beyond 1e-4 of the scale, each within the conditioning room of its own elements" % (arbitrated, total))
    assert not gf[t != D.PT_FLUID].any()
    assert_close_but_for_gamma_spikes(_np(eng.cfl)[:nb], cfl[:nb], 1e-4, what="SA CFL maxima")
    # gamma by quadrature at displaced positions
    rng = np.random.default_rng(3)
    newpos = sim.pos.copy()
    newpos[fl, :3] += (0.1 * sim.problem.m_deltap * rng.standard_normal((len(fl), 3))).astype(np.float32)
    g1 = o.sa_integrate_gamma(sim.gg, newpos, sim.be, sim.vertpos, sim.info, sim.hash, sim.cs, sim.nl, n)
    eng.pos2[:n] = torch.from_numpy(newpos).to(eng.device)
    k.sa_integrate_gamma(eng.gradgamma2, eng.gradgamma, eng.pos2, eng.boundelements, eng.vertpos, eng.info, eng.hash, eng.cellStart,
                         eng.neibslist, n, n)
    gg1 = _np(eng.gradgamma2)[:n]
    assert_close_but_for_gamma_spikes(gg1[fl, :3], g1[fl, :3], 5e-5, what="grad gamma by quadrature", wall=wall[fl])
    assert np.abs(gg1[fl, 3] - g1[fl, 3]).max() < 5e-6
    assert np.array_equal(_bits(gg1[t != D.PT_FLUID]), _bits(sim.gg[t != D.PT_FLUID]))      # walls: copied
    # six steps of the full sequence
    eng2 = _engine(SABox(**kw))
    for _ in range(6):
        sim.step(); eng2.step()
    assert eng2.iterations == sim.iterations == 6
    gp, gv, ggg = _np(eng2.pos)[:n], _np(eng2.vel)[:n], _np(eng2.gradgamma)[:n]
    cell = float(np.min(sim.problem.m_cellsize))
    assert np.abs(gp[:, :3] - sim.pos[:, :3]).max() < 6e-6 * cell
    W = wall_rows(sim.problem, sim.nl, sim.info, n)
    assert_close_but_for_gamma_spikes(gv[:, :3], sim.vel[:, :3], 1e-3, max(np.abs(sim.vel[:, :3]).max(), 1e-3), what="velocities after 6 steps (quadrature)", wall=W, away=AWAY, **RUN)
    assert_close_but_for_gamma_spikes(gv[:, 3], sim.vel[:, 3], 2e-6, 1.0, what="densities after 6 steps (quadrature)", wall=W, away=AWAY, **RUN)
    assert np.abs(ggg[fl, 3] - sim.gg[fl, 3]).max() < 2e-5
    assert abs(eng2.current_dt() - sim.dt) < 1e-4 * sim.dt and abs(eng2.time() - sim.t) < 1e-5 * sim.t


@pytest.mark.parametrize("options", ["StillWaterRepackSA", "StillWaterSA"])
def test_cpp_adapters_step_an_sa_problem(tmp_path, options):
    """Four predictor-corrector steps of the SA sequence through the abstract interfaces of the GPUSPH tree, framework from the
    problem's own SETUP_FRAMEWORK: forces, dtreduce (with the gamma condition), Euler, then integrate_gamma (StillWaterRepackSA)
    or density_sum + compute_/apply_density_diffusion (StillWaterSA), boundary conditions: bit-equal to the Python driver."""
    import subprocess
    import host_case as hc
    kw = dict(deltap=0.05, jitter=0.1, options=options)
    prob = SABox(**kw)
    eng = _engine(prob)
    n = prob.num_particles
    case = tmp_path / "case.txt"
    case.write_text("\n".join(hc.case_lines(prob, options, allocated=eng.alloc) + hc.driver_lines(prob, eng, 4)) + "\n")
    hc.write_state(str(tmp_path / "state.bin"), prob.copy_to_array())
    subprocess.check_call([hc.exe("example_engines"), str(case), str(tmp_path / "state.bin"), str(tmp_path / "out.bin")])
    out = hc.read_out(str(tmp_path / "out.bin"))
    eng.run(4)
    assert out["n"] == n
    for name in ("pos", "vel", "gradgamma", "boundelements"):
        assert np.array_equal(_bits(out[name]), _bits(_np(getattr(eng, name))[:n])), name
    assert np.float32(out["dt"]) == np.float32(eng.current_dt()) and abs(out["t"] - eng.time()) < 1e-12
    t = info_type(out["info"])
    assert np.abs(out["vel"][t == D.PT_FLUID, :3]).max() > 0            # it did move


def test_density_summation_form_on_the_gpu(kernels):
    """StillWaterSA's own option set -- density summation, dynamic gamma with its CFL condition, Brezzi diffusion: the three
    engines' calls one by one on identical inputs, then six steps of the whole sequence, against the CPU oracle."""
    from sa_helpers import OracleSaSim
    import torch
    kw = dict(deltap=0.05, jitter=0.15, options="StillWaterSA")
    sim = OracleSaSim(SABox(**kw))
    eng = _engine(SABox(**kw), clobber_neibslist=True)
    eng.build_neibs()
    assert _tiles_usable(eng) == (kernels == "tiled")
    eng.sa_boundary_conditions(0)
    n, o, k = sim.n, sim.o, eng.k
    t = info_type(sim.info)
    fl = np.where(t == D.PT_FLUID)[0]
    rng = np.random.default_rng(11)
    vel = sim.vel.copy()
    vel[fl, :3] = (2.0 * rng.standard_normal((len(fl), 3))).astype(np.float32)           # moving fluid: a gamma CFL to speak of
    for name, arr in (("vel", vel), ("gradgamma", sim.gg), ("boundelements", sim.be)):
        getattr(eng, name)[:n] = torch.from_numpy(arr).to(eng.device)
    f, cfl, nb = o.forces_sa(sim.pos, vel, sim.info, sim.hash, sim.cs, sim.nl, sim.gg, sim.be, sim.vertpos, n, sim.problem.m_deltap)
    k.memset(eng.cfl, 0)
    gnb = k.forces_sa(eng.forces, eng.cfl, eng.pos, eng.vel, eng.info, eng.hash, eng.cellStart, eng.neibslist, eng.gradgamma,
                      eng.boundelements, eng.vertpos, n, 0, n, 0, cfl_gamma=eng.cfl_gamma)
    assert gnb == nb
    gf = _np(eng.forces)[:n]
    wall = wall_rows(sim.problem, sim.nl, sim.info, n)      # the gamma allowance is for particles next to a wall only
    assert_close_but_for_gamma_spikes(gf[fl, :3], f[fl, :3], 1e-4, what="SA forces (density-summation form)", wall=wall[fl])
    assert not gf[fl, 3].any() and not f[fl, 3].any()
    n4 = (n + 3) // 4 * 4
    gcg = _np(eng.cfl_gamma)
    assert o.max_gamma_cfl > 0.1
    gscale = o.cfl_gamma[:n].max()
    assert_close_but_for_gamma_spikes(gcg[:n], o.cfl_gamma[:n], 1e-4, gscale, what="gamma CFL terms", wall=wall)
    assert_close_but_for_gamma_spikes(gcg[n4:n4 + nb], o.cfl_gamma[n4:n4 + nb], 1e-4, gscale, what="gamma CFL maxima")
    # dt with the gamma condition
    dt_ref = min(o.dtreduce(cfl, nb, sim.sspeed_cfl, sim.max_kinvisc), 1e9)
    dt_ref = min(dt_ref, float(o.L.orc_sa_gamma_dt(np.float32(dt_ref), np.float32(o.max_gamma_cfl))))
    k.dtreduce(eng.cfl, eng.cfl_temp, nb, eng.d_dt_next, 0)
    k.dtreduce_gamma(eng.cfl_gamma, n, nb, eng.d_dt_next)
    assert abs(float(eng.d_dt_next.item()) - dt_ref) < 5e-4 * dt_ref and dt_ref < 0.9 * o.dtreduce(cfl, nb, sim.sspeed_cfl, sim.max_kinvisc)
    # density summation between the state and a displaced copy; Brezzi diffusion on the result
    newpos = sim.pos.copy()
    newpos[fl, :3] += (0.05 * sim.problem.m_deltap * rng.standard_normal((len(fl), 3))).astype(np.float32)
    v1, g1 = o.sa_density_sum(vel, sim.pos, newpos, vel, sim.gg, sim.be, sim.vertpos, sim.info, sim.hash, sim.cs, sim.nl, n)
    eng.pos2[:n] = torch.from_numpy(newpos).to(eng.device)
    eng.vel2[:n] = torch.from_numpy(vel).to(eng.device)
    k.sa_density_sum(eng.vel2, eng.gradgamma2, eng.forces, eng.pos, eng.pos2, eng.vel, eng.gradgamma, eng.boundelements, eng.vertpos,
                     eng.info, eng.hash, eng.cellStart, eng.neibslist, n, n)
    gv1, gg1 = _np(eng.vel2)[:n], _np(eng.gradgamma2)[:n]
    assert_close_but_for_gamma_spikes(gv1[:, 3], v1[:, 3], 2e-6, 1.0, what="density summation", wall=wall)
    assert np.array_equal(_bits(gv1[:, :3]), _bits(v1[:, :3]))
    assert_close_but_for_gamma_spikes(gg1[fl, 3], g1[fl, 3], 5e-6, 1.0, what="dynamic gamma", wall=wall[fl])
    assert_close_but_for_gamma_spikes(gg1[fl, :3], g1[fl, :3], 5e-5, what="grad gamma at the new positions", wall=wall[fl])
    assert np.array_equal(_bits(gg1[t != D.PT_FLUID]), _bits(sim.gg[t != D.PT_FLUID]))
    dt = 3.0e-4
    v2, fd = o.sa_density_diffusion(newpos, v1, g1, sim.info, sim.hash, sim.cs, sim.nl, n, dt)
    eng.vel2[:n] = torch.from_numpy(v1).to(eng.device); eng.gradgamma2[:n] = torch.from_numpy(g1).to(eng.device)
    k.sa_density_diffusion(eng.forces, eng.pos2, eng.vel2, eng.gradgamma2, eng.info, eng.hash, eng.cellStart, eng.neibslist, n, n, dt)
    gfd, gv2 = _np(eng.forces)[:n], _np(eng.vel2)[:n]
    assert np.abs(fd[fl, 3]).max() > 0 and np.abs(gfd[fl, 3] - fd[fl, 3]).max() < 1e-4 * np.abs(fd[fl, 3]).max()
    assert np.abs(gv2[:, 3] - v2[:, 3]).max() < 2e-7
    # six steps of the full sequence
    sim2 = OracleSaSim(SABox(**kw))
    eng2 = _engine(SABox(**kw))
    for _ in range(6):
        sim2.step(); eng2.step()
    gp, gv, ggg = _np(eng2.pos)[:n], _np(eng2.vel)[:n], _np(eng2.gradgamma)[:n]
    cell = float(np.min(sim2.problem.m_cellsize))
    assert np.abs(gp[:, :3] - sim2.pos[:, :3]).max() < 6e-6 * cell
    W = wall_rows(sim2.problem, sim2.nl, sim2.info, n)
    assert_close_but_for_gamma_spikes(gv[:, :3], sim2.vel[:, :3], 1e-3, max(np.abs(sim2.vel[:, :3]).max(), 1e-3), what="velocities after 6 steps (density sum)", wall=W, away=AWAY, **RUN)
    assert_close_but_for_gamma_spikes(gv[:, 3], sim2.vel[:, 3], 2e-6, 1.0, what="densities after 6 steps (density sum)", wall=W, away=AWAY, **RUN)
    assert np.abs(ggg[fl, 3] - sim2.gg[fl, 3]).max() < 2e-5
    assert abs(eng2.current_dt() - sim2.dt) < 1e-5 * sim2.dt and abs(eng2.time() - sim2.t) < 1e-6 * sim2.t


def test_sa_repacking_run_follows_the_oracle():
    """the repacking run mode with SA_BOUNDARY (StillWaterRepackSA's ENABLE_REPACKING): REPACK variants of the initial boundary
    conditions, the mixing force with the wall term divided by gamma, Euler of the fluid, gamma by quadrature at the new positions"""
    import torch
    from sa_helpers import OracleSaSim
    kw = dict(deltap=0.05, options="StillWaterRepackSA", jitter=0.2)
    sim = OracleSaSim(SABox(**kw), repack=True)
    eng = _engine(SABox(**kw))
    n = sim.n
    # single pass first: forces of the first iteration
    eng.build_neibs()
    eng.sa_boundary_conditions(0, D.REPACK)
    f, cfl, nb = sim.o.repack_forces_sa(sim.pos, sim.vel, sim.info, sim.hash, sim.cs, sim.nl, sim.gg, sim.be, sim.vertpos, n, sim.problem.m_deltap)
    eng._forces(eng.pos, eng.vel, 1, 0, D.REPACK)
    gf = _np(eng.forces)[:n]
    scale = np.abs(f[:n, :3]).max()
    assert scale > 0
    assert_close_but_for_gamma_spikes(gf[:, :3], f[:n, :3], 5e-5, scale, what="SA repacking force")      # |grad gamma| of the elements: powf-free, sums of ~100 terms
    dt_ref = sim.o.dtreduce(cfl, nb, sim.sspeed_cfl, sim.max_kinvisc)
    assert abs(float(eng.d_dt_next.item()) - dt_ref) <= 5e-5 * dt_ref
    # then the loop (the engine's repack_step does its own initialisation on a fresh engine)
    eng = _engine(SABox(**kw))
    steps = 8
    for _ in range(steps):
        sim.repack_step(); eng.repack_step()
    cs = float(min(sim.problem.m_cellsize))
    assert np.array_equal(_np(eng.hash, np.uint32)[:n], sim.hash[:n])
    assert np.abs(_np(eng.pos)[:n, :3] - sim.pos[:n, :3]).max() <= 2e-6 * cs * steps
    assert np.abs(_np(eng.vel)[:n, :3] - sim.vel[:n, :3]).max() <= 1e-3 * max(np.abs(sim.vel[:n, :3]).max(), 1e-6)
    fluid = info_type(sim.info[:n]) == D.PT_FLUID
    np.testing.assert_allclose(_np(eng.gradgamma)[:n, 3][fluid], sim.gg[:n, 3][fluid], atol=2e-5)
    assert np.abs(sim.vel[:n, :3][fluid]).max() > 0


def test_sa_tank_at_4M_particles_through_properties():
    """BASELINE configs[2]'s wording is "StillWater 4M particles, SA boundary": the SA form of that tank (StillWaterSA's option set:
    density summation, dynamic gamma, Brezzi diffusion) at 4.3 M particles, which no CPU oracle run covers in test time, through
    what must hold whatever the arithmetic: the lists fit and are typed, gamma is 1 in the bulk, 1/2 / 1/4 / 1/8 for vertex
    particles on faces / edges / corners, between 0.1 and 1 everywhere, grad gamma points out of the fluid on every wall, a
    hydrostatic tank stays at rest over a rebuild (velocities a small fraction of sqrt(g H), densities near the hydrostatic ones)."""
    prob = SABox(0.008, l=1.6, w=1.6, h=1.0, H=0.8, options="StillWaterSA")
    assert prob.num_particles > 4.0e6
    eng = _engine(prob)
    steps = 12                    # crosses the rebuild at iteration 10
    for _ in range(steps):
        eng.step()
    n = eng.n
    assert n == prob.num_particles
    info = eng.neibs_info()
    assert info.hasTooManyNeibs == -1 and info.maxVertexNeibs > 10 and 0 < info.maxFluidBoundaryNeibs < prob.simparams.neibboundpos
    h = _np(eng.hash, np.uint32)[:n]
    assert (np.diff(h.astype(np.int64) & 0x3FFFFFFF) >= 0).all()
    pos, vel, gg, inf = _np(eng.pos)[:n], _np(eng.vel)[:n], _np(eng.gradgamma)[:n], _np(eng.info, np.uint16)[:n]
    t = info_type(inf)
    fl, vx = t == D.PT_FLUID, t == D.PT_VERTEX
    assert np.isfinite(pos).all() and np.isfinite(vel).all() and np.isfinite(gg[t != D.PT_BOUNDARY]).all()
    g = prob.global_pos(pos, h)
    dp = prob.m_deltap
    # gamma: 1 away from the walls, in [0.1, 1] everywhere, one half / quarter / eighth on the tank's faces / edges / corners
    wall_dist = np.minimum.reduce([g[:, 0], prob.l - g[:, 0], g[:, 1], prob.w - g[:, 1], g[:, 2]])
    bulk = fl & (wall_dist > prob.simparams.influenceRadius + dp)
    assert bulk.sum() > 3.0e6 and np.abs(gg[bulk, 3] - 1.0).max() < 1e-6 and np.abs(gg[bulk, :3]).max() < 1e-6
    assert gg[fl, 3].min() >= 0.1 - 1e-6 and gg[fl, 3].max() <= 1.0 + 1e-6
    dist = np.stack([g[:, 0], prob.l - g[:, 0], g[:, 1], prob.w - g[:, 1], g[:, 2]], axis=1)
    on_wall = dist < 0.25 * dp
    count = on_wall.sum(axis=1)
    clear = np.where(on_wall, np.inf, dist).min(axis=1) > 2 * dp         # the walls it is not on are out of the way
    below = g[:, 2] < prob.H - 3 * prob.simparams.influenceRadius      # wetted part of the walls, away from the free surface
    for k, want, tol in ((1, 0.5, 5e-3), (2, 0.25, 0.03), (3, 0.125, 5e-3)):
        sel = vx & (count == k) & below & clear
        assert sel.sum() > (1000 if k == 1 else 3), k
        assert np.abs(gg[sel, 3] - want).max() < tol, (k, np.abs(gg[sel, 3] - want).max())
    # grad gamma of fluid particles next to the floor points into the fluid
    near_floor = fl & (g[:, 2] < 0.6 * prob.simparams.influenceRadius) & (dist[:, :4].min(axis=1) > 3 * prob.simparams.influenceRadius)
    assert near_floor.sum() > 1000 and (gg[near_floor, 2] > 0).all()
    # still water stays still: velocities far below the gravity-wave speed, densities near hydrostatic
    c_wave = np.sqrt(9.81 * prob.H)
    assert np.abs(vel[fl, :3]).max() < 0.02 * c_wave
    pp = prob.physparams
    rho_hyd = (1.0 + pp.rho0[0] * 9.81 * np.clip(prob.H - g[fl, 2], 0, None) / pp.bcoeff[0]) ** (1.0 / pp.gammacoeff[0]) - 1.0
    deep = bulk[fl] & (g[fl, 2] < prob.H - 2 * prob.simparams.influenceRadius)       # away from walls and free surface
    assert deep.sum() > 2.0e6 and np.abs(vel[fl, 3] - rho_hyd)[deep].max() < 0.25 * rho_hyd.max()      # the start-up transient of the density summation
    assert np.abs(vel[fl, 3] - rho_hyd).max() < 0.5 * rho_hyd.max()
    assert 0 < eng.current_dt() <= prob.simparams.dt * 1.0001


@pytest.mark.parametrize("options", ["StillWaterSA", "StillWaterRepackSA"])
def test_sa_tiled_kernels_agree_with_the_list_walkers(options, monkeypatch):
    """The two ways of forming the SA sums (tiled window + a wave per wall particle / one thread per particle in list order) on a
    tank too large for the CPU oracle in test time: twelve steps across a neighbour-list rebuild, a sloshing start so that
    every term is alive.  They differ by the order of the sums and the fast reciprocal / square root of the tiled pair."""
    import torch
    def run(disable):
        monkeypatch.setenv("SPHX_DISABLE_TILES", disable)
        prob = SABox(0.02, l=1.2, w=0.8, h=0.8, H=0.6, jitter=0.1, options=options)
        eng = _engine(prob)
        n = prob.num_particles
        g = prob.global_pos(_np(eng.pos)[:n], _np(eng.hash, np.uint32)[:n])
        fl = info_type(_np(eng.info, np.uint16)[:n]) == D.PT_FLUID
        v = _np(eng.vel)[:n].copy()
        v[fl, 0] = 0.3 * np.sin(np.pi * g[fl, 0] / prob.l) * (g[fl, 2] / prob.H)        # a first sloshing mode
        v[fl, 2] = -0.3 * np.cos(np.pi * g[fl, 0] / prob.l) * (g[fl, 2] / prob.H) * 0.5
        eng.vel[:n] = torch.from_numpy(v.astype(np.float32)).to(eng.device)
        eng.build_neibs()
        usable = _tiles_usable(eng)
        for _ in range(12):
            eng.step()
        return prob, usable, _np(eng.pos)[:n].copy(), _np(eng.vel)[:n].copy(), _np(eng.gradgamma)[:n].copy(), \
            _np(eng.hash, np.uint32)[:n].copy(), eng.current_dt(), fl
    prob, used_t, pos_t, vel_t, gg_t, hash_t, dt_t, fl = run("0")
    _, used_w, pos_w, vel_w, gg_w, hash_w, dt_w, _ = run("1")
    assert used_t == 1 and used_w == 0
    assert prob.num_particles > 9.0e4
    assert np.array_equal(hash_t, hash_w)
    cell = float(np.min(prob.m_cellsize))
    vmax = np.abs(vel_w[:, :3]).max()
    assert vmax > 0.2
    # (a particle with an ill-conditioned element among its neighbours is pushed a little differently by the two: see
    # assert_close_but_for_gamma_spikes)
    assert_close_but_for_gamma_spikes(pos_t[:, :3], pos_w[:, :3], 2e-5, cell, frac=0.002, spike=2.0, what="positions, tiled against list walkers")      # measured worst 0.75
    # (velocities: the one or two worst particles end between 1.6e-3 and 2.1e-3 of max |v| after the twelve steps, depending on how
    # a build happens to group the partial sums of a tile -- the cut points of the waves' shares moved in round 4 --; the share
    # of the entries beyond the plain tolerance is 3e-4, held to 2e-3 here instead of the helper's 1e-2)
    assert_close_but_for_gamma_spikes(vel_t[:, :3], vel_w[:, :3], 2e-4, vmax, frac=2e-3, spike=15.0, what="velocities, tiled against list walkers")
    assert_close_but_for_gamma_spikes(vel_t[:, 3], vel_w[:, 3], 2e-6, 1.0, frac=0.002, spike=4.0, what="densities, tiled against list walkers")      # measured 1e-4 of the entries, worst 1.6
    assert np.abs(gg_t[fl, 3] - gg_w[fl, 3]).max() < 2e-5
    assert abs(dt_t - dt_w) < 1e-4 * dt_w


@pytest.mark.gpu
def test_force_on_a_body_that_feels_the_fluid():
    """sphx_sa_body_pressure_forces (compute_boundary_pressure_force + the BUFFER_RB_FORCES / RB_TORQUES rows of finalizeforcesDevice,
    src/cuda/forces_kernel.def:3258-3266,4115-4145) for the floor of SALoadBox: rows, torques and the elements' own forces rows
    against the oracle's restatement on identical inputs (the pressure goes through powf of two math libraries: 1 ulp of
    (1 + rho~)^7 is 1e-5 of P at these densities), the totals through sphx_reduce_rb_forces against the weight of the water, and
    the same rows left behind by the engine's own forces pass (the driver calls the entry point behind every SA forces pass when
    the problem has force bodies)."""
    from gpusph_amd.problem import SALoadBox
    from sa_helpers import OracleSaSim
    import torch
    kw = dict(deltap=0.05, jitter=0.1)
    sim = OracleSaSim(SALoadBox(**kw))
    eng = _engine(SALoadBox(**kw), clobber_neibslist=True)
    eng.build_neibs()
    eng.sa_boundary_conditions(0)
    n, o, k, p = sim.n, sim.o, eng.k, sim.problem
    assert np.array_equal(_np(eng.info, np.uint16)[:n], sim.info[:n])
    for name, arr in (("vel", sim.vel), ("boundelements", sim.be)):
        getattr(eng, name)[:n] = torch.from_numpy(arr).to(eng.device)
    f, rbf, rbt = o.sa_body_pressure_forces(sim.pos, sim.vel, sim.info, sim.hash, sim.be, n, p.num_obstacle)
    eng.forces.zero_(); eng.rbforces.zero_(); eng.rbtorques.zero_()
    k.sa_body_pressure_forces(eng.forces, eng.rbforces, eng.rbtorques, eng.pos, eng.vel, eng.info, eng.hash, eng.boundelements, 0, n)
    grf, grt, gf = _np(eng.rbforces), _np(eng.rbtorques), _np(eng.forces)[:n]
    scale = np.abs(rbf[:, 2]).max()
    assert np.abs(grf - rbf).max() <= 3e-5 * scale
    assert np.abs(grt - rbt).max() <= 3e-5 * scale * max(p.l, p.w)
    assert np.abs(gf - f[:n]).max() <= 3e-5 * scale and (gf[:, 3] == 0).all()
    load = (sim.info[:n, 0] & D.FG_COMPUTE_FORCE) != 0
    assert not gf[~load].any()
    # the totals: the weight of the water on the floor
    weight = p.physparams.rho0[0] * 9.81 * p.l * p.w * p.water_level
    tf, tt = eng.reduce_rb_forces()
    assert abs(tf[2] + weight) < 0.01 * weight and abs(tf[2] - rbf[:, 2].astype(np.float64).sum()) < 1e-4 * weight
    # a range launch touches its own rows only
    eng.rbforces.zero_(); eng.rbtorques.zero_()
    k.sa_body_pressure_forces(eng.forces, eng.rbforces, eng.rbtorques, eng.pos, eng.vel, eng.info, eng.hash, eng.boundelements, n // 2, n)
    idx = np.where(load)[0]
    rows = (sim.info[idx, 2].astype(np.int64) | (sim.info[idx, 3].astype(np.int64) << 16)) + int(p.rb_firstindex[0])
    part = _np(eng.rbforces)
    assert np.array_equal(part[rows[idx >= n // 2]].view(np.uint32), grf[rows[idx >= n // 2]].view(np.uint32))
    assert not part[rows[idx < n // 2]].any()
    # the engine's own step leaves the rows of its last forces pass
    for _ in range(2):
        eng.step()
    tf2, _ = eng.reduce_rb_forces()
    assert abs(tf2[2] + weight) < 0.02 * weight
