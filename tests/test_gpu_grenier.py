"""SPH_GRENIER on the GPU (grenier.hip, the Grenier branch of euler.hip and neibs.hip) against the CPU oracle: the list build
with boundary-boundary neighbours bit for bit, sigma / density / forces / volumes to the tolerance of the fp32 math library,
then whole steps of the driver."""
import numpy as np
import pytest

from gpusph_amd import defs as D
from gpusph_amd.problem import info_type
import oracle_lib as ol
from grenier_helpers import grenier_problem, grenier_state

pytestmark = pytest.mark.gpu


def _engine(problem, **kw):
    import torch
    from gpusph_amd.engine import TimestepEngine
    assert torch.cuda.is_available()
    return TimestepEngine(problem, device="cuda:0", **kw)


def _np(t, dtype=None):
    a = t.cpu().numpy()
    return a.view(dtype) if dtype is not None else a


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.fixture(scope="module", params=[dict(deltap=0.04), dict(deltap=0.035, viscosity=dict(rheologytype=D.NEWTONIAN, turbmodel=D.LAMINAR_FLOW,
                                                                                            compvisc=D.DYNAMIC, avgop=D.ARITHMETIC))])
def pair(request):
    import torch
    pr = grenier_problem(**request.param)
    sim, g = grenier_state(pr)
    eng = _engine(grenier_problem(**request.param), clobber_neibslist=True)
    eng.build_neibs()
    eng.vel.copy_(torch.from_numpy(np.pad(sim.vel, ((0, eng.alloc - len(sim.vel)), (0, 0)))).to(eng.device))
    return sim, eng


def test_list_build_keeps_boundary_boundary_pairs_bit_exact(pair):
    sim, eng = pair
    n = sim.n
    assert eng.n == n
    assert np.array_equal(_np(eng.hash, np.uint32)[:n], sim.hash[:n])
    assert np.array_equal(_bits(_np(eng.pos)[:n]), _bits(sim.pos[:n]))
    nl = _np(eng.neibslist, np.uint16).reshape(-1, eng.alloc)[:, :n]
    assert np.array_equal(nl, sim.nl.reshape(-1, len(sim.pos))[:, :n])
    info = eng.neibs_info()
    assert (info.numInteractions, info.maxFluidBoundaryNeibs) == (sim.neibs_info.numInteractions, sim.neibs_info.maxFluidBoundaryNeibs)
    t = info_type(sim.info[:n])
    bp = int(sim.o.p.neibboundpos)
    assert (nl[bp][t == D.PT_BOUNDARY] != 0xFFFF).all()          # every wall particle has wall neighbours
    # the re-sort carried the volumes
    assert np.array_equal(_bits(_np(eng.vol)[:n]), _bits(sim.vol[:n]))


def test_density_sigma_forces_and_volume_step(pair):
    import torch
    sim, eng = pair
    n = sim.n
    K = eng.k
    o = sim.o
    vel = sim.vel.copy()
    sigma = o.density_grenier(sim.pos, vel, sim.info, sim.hash, sim.vol, sim.cs, sim.nl, n, sim.neibs_info.maxFluidBoundaryNeibs)
    K.compute_density(eng.sigma, eng.vel, eng.pos, eng.info, eng.hash, eng.vol, eng.cellStart, eng.neibslist, n)
    got_sigma = _np(eng.sigma)[:n]
    np.testing.assert_allclose(got_sigma, sigma[:n], rtol=2e-6)
    t = info_type(sim.info[:n])
    R = np.float32(o.p.influenceradius)
    typical = np.float32(3 * sim.neibs_info.maxFluidBoundaryNeibs) / (np.float32(4) * np.float32(np.pi) * R * R * R)
    assert (np.abs(got_sigma - typical) < 1e-6 * typical).sum() > 50         # the 'typical sigma' rows took the device counter
    got_vel = _np(eng.vel)[:n]
    assert np.array_equal(_bits(got_vel[:, :3]), _bits(vel[:n, :3]))
    np.testing.assert_allclose(got_vel[:, 3], vel[:n, 3], atol=3e-7)
    # forces on the oracle's density and sigma (so that the comparison is of the forces kernel alone)
    eng.vel.copy_(torch.from_numpy(np.pad(vel, ((0, eng.alloc - len(vel)), (0, 0)))).to(eng.device))
    eng.sigma[:len(sigma)].copy_(torch.from_numpy(sigma).to(eng.device))
    f, cfl, nb = o.forces_grenier(sim.pos, vel, sim.info, sim.hash, sim.cs, sim.nl, sigma, n)
    K.memset(eng.forces, 0); K.memset(eng.cfl, 0)
    nb_g = K.forces_grenier(eng.forces, eng.cfl, eng.pos, eng.vel, eng.info, eng.hash, eng.cellStart, eng.neibslist, eng.sigma, n, 0, n, 0)
    assert nb_g == nb
    got = _np(eng.forces)[:n]
    scale = np.abs(f[:n, :3]).max()
    # tolerance: powf of the math library in P (conditioning ~ gamma/rho~ on the pressure), fp32 sums of ~100 terms
    assert np.abs(got[:, :3] - f[:n, :3]).max() <= 2e-5 * scale
    assert np.abs(got[:, 3] - f[:n, 3]).max() <= 2e-5 * np.abs(f[:n, 3]).max()
    np.testing.assert_allclose(_np(eng.cfl)[:nb], cfl[:nb], rtol=2e-5)
    # Euler: positions / velocities bit for bit from the same forces, the volume to one expf rounding
    import torch
    eng.forces[:len(f)].copy_(torch.from_numpy(f).to(eng.device))
    dt = np.float32(2e-4)
    d_dt = torch.full((1,), float(dt), dtype=torch.float32, device=eng.device)
    for step, scale_dt in ((1, 0.5), (2, 1.0)):
        ps, vs, vols = o.euler_grenier(sim.pos, vel, sim.vol, sim.info, sim.hash, f, n, float(np.float32(dt * np.float32(scale_dt))), step)
        K.euler_grenier(eng.pos2, eng.vel2, eng.vol2, eng.pos, eng.vel, eng.vol, eng.info, eng.hash, eng.forces, n, d_dt, scale_dt, step)
        assert np.array_equal(_bits(_np(eng.pos2)[:n]), _bits(ps[:n]))
        assert np.array_equal(_bits(_np(eng.vel2)[:n]), _bits(vs[:n]))
        gv = _np(eng.vol2)[:n]
        assert np.array_equal(_bits(gv[:, :3]), _bits(vols[:n, :3]))
        np.testing.assert_allclose(gv[:, 3], vols[:n, 3], rtol=3e-7)
        assert np.abs(gv[:, 1]).max() > 0


def test_entry_points_refuse_what_is_not_built():
    from gpusph_amd import capi
    from gpusph_amd.problem import DamBreak3D
    pr = grenier_problem(0.06)
    eng = _engine(pr)
    eng.build_neibs()
    K = eng.k
    n = eng.n
    with pytest.raises(capi.SphxInvalidArgument):       # the plain Euler entry does not know about BUFFER_VOLUME
        K.euler(eng.pos2, eng.vel2, eng.pos, eng.vel, eng.info, eng.hash, eng.forces, n, eng.d_dt, 0.5, 1)
    other = _engine(DamBreak3D(0.06, obstacle=False))
    other.build_neibs()
    with pytest.raises(capi.SphxInvalidArgument):
        other.k.forces_grenier(other.forces, other.cfl, other.pos, other.vel, other.info, other.hash, other.cellStart, other.neibslist,
                               other.forces, other.n, 0, other.n, 0)
    for kw in (dict(density_diffusion=D.COLAGROSSI), dict(viscosity="ARTVISC"), dict(boundary=D.LJ_BOUNDARY)):
        args = dict(obstacle=False, two_fluids=True, formulation=D.SPH_GRENIER, viscosity="DYNAMICVISC", density_diffusion=D.DENSITY_DIFFUSION_NONE)
        args.update(kw)
        with pytest.raises(capi.SphxUnsupported):
            _engine(DamBreak3D(0.06, **args))


@pytest.mark.parametrize("kw", [dict(deltap=0.04, jitter=0.0), dict(deltap=0.04, jitter=0.1, viscosity=dict(
    rheologytype=D.NEWTONIAN, turbmodel=D.LAMINAR_FLOW, compvisc=D.KINEMATIC, avgop=D.GEOMETRIC))])
def test_steps_follow_the_oracle(kw):
    pr = grenier_problem(**kw)
    sim = ol.OracleSim(pr)
    eng = _engine(grenier_problem(**kw))
    steps = 12                                   # crosses a neighbour rebuild (buildneibsfreq = 10)
    for _ in range(steps):
        sim.step(); eng.step()
    out = eng.download()
    n = eng.n
    assert n == sim.n and np.array_equal(out["hash"], sim.hash[:n])
    cs = float(min(pr.m_cellsize))
    assert np.abs(out["pos"][:, :3] - sim.pos[:n, :3]).max() <= 1e-6 * cs * steps
    assert np.abs(out["vel"][:, :3] - sim.vel[:n, :3]).max() <= 1e-3 * max(np.abs(sim.vel[:n, :3]).max(), 1e-6)
    np.testing.assert_allclose(_np(eng.vol)[:n, 3], sim.vol[:n, 3], rtol=1e-5)
    np.testing.assert_allclose(_np(eng.vol)[:n, 1], sim.vol[:n, 1], atol=1e-6)
    assert abs(eng.current_dt() - sim.dt) <= 2e-5 * sim.dt
    assert np.abs(sim.vol[:n, 1]).max() > 1e-6


def test_cpp_adapters_run_the_grenier_step_like_the_python_driver(tmp_path):
    """example_engines (built inside the GPUSPH tree against its own headers) with the Bubble framework: COMPUTE_DENSITY through
    AbstractForcesEngine::compute_density, the forces and Euler steps with BUFFER_SIGMA / BUFFER_VOLUME in the tree's BufferLists"""
    import os, subprocess
    import host_case as hc
    exe = hc.exe("example_engines")
    assert os.path.exists(exe), "gpusph_amd/host/example_engines is not built (make -C gpusph_amd/host, needs the GPUSPH tree)"
    prob = grenier_problem(0.04, jitter=0.05)
    prob.simparams.simflags &= ~D.ENABLE_REPACKING           # the Bubble selector list of problem_setup.h has no repacking flag
    eng = _engine(prob)
    steps = 12
    case, state, fout = tmp_path / "case.txt", tmp_path / "state.bin", tmp_path / "out.bin"
    case.write_text("\n".join(hc.case_lines(prob, "Bubble") + hc.driver_lines(prob, eng, steps)) + "\n")
    hc.write_state(state, prob.copy_to_array())
    r = subprocess.run([exe, str(case), str(state), str(fout)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    eng.run(steps)
    ref = eng.download()
    out = hc.read_out(fout)
    n = out["n"]
    assert n == eng.n and np.float32(eng.current_dt()) == out["dt"] and eng.time() == out["t"]
    assert np.array_equal(out["hash"], ref["hash"])
    assert np.array_equal(_bits(out["pos"]), _bits(ref["pos"])) and np.array_equal(_bits(out["vel"]), _bits(ref["vel"]))
    assert np.array_equal(_bits(out["vol"]), _bits(_np(eng.vol)[:n]))
    assert np.abs(out["vol"][:, 1]).max() > 1e-6
