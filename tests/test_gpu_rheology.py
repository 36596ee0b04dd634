"""Generalized Newtonian rheologies on the GPU (rheology.hip) against the CPU oracle: effective viscosity of the seven models,
forces with the per-particle viscosity, whole steps with the viscous dt limit from the largest effective viscosity, the C++
adapters with the tree's PoiseuillePapanastasiou framework."""
import numpy as np
import pytest

from gpusph_amd import defs as D
from gpusph_amd.problem import Poiseuille, info_type
import oracle_lib as ol
from test_rheology_oracle import RHEOLOGIES, _problem, _sheared

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


def _pair(pr_factory, g=0.8):
    import torch
    sim, gp = _sheared(pr_factory(), g=g)
    eng = _engine(pr_factory(), clobber_neibslist=True)
    eng.build_neibs()
    n = sim.n
    assert eng.n == n and np.array_equal(_np(eng.hash, np.uint32)[:n], sim.hash[:n])
    eng.vel[:n].copy_(torch.from_numpy(sim.vel[:n]).to(eng.device))
    return sim, eng


@pytest.mark.parametrize("rheology", RHEOLOGIES)
@pytest.mark.parametrize("compvisc", [D.KINEMATIC, D.DYNAMIC])
def test_effective_viscosity_and_forces(rheology, compvisc):
    import torch
    sim, eng = _pair(lambda: _problem(rheology, compvisc=compvisc, viscavg=(D.HARMONIC if compvisc == D.KINEMATIC else D.GEOMETRIC),
                                      density_diffusion=(D.COLAGROSSI if rheology % 2 else D.FERRARI)))
    n = sim.n
    K = eng.k
    eff, mx = sim.o.effective_visc(sim.pos, sim.vel, sim.info, sim.hash, sim.cs, sim.nl, n)
    got_mx = K.calc_effvisc(eng.effvisc, eng.pos, eng.vel, eng.info, eng.hash, eng.cellStart, eng.neibslist, n, n)
    got = _np(eng.effvisc)[:n]
    # tolerance: the shear rate norm is a sum of ~100 fp32 terms, then powf / expf of the math library
    np.testing.assert_allclose(got, eff[:n], rtol=3e-5)
    assert got_mx == pytest.approx(mx, rel=3e-5) and K.max_kinvisc == got_mx
    assert np.ptp(eff[:n]) > 0.05 * eff[:n].max()                      # the field really varies
    # forces on the oracle's viscosity field
    eng.effvisc[:n].copy_(torch.from_numpy(eff[:n]).to(eng.device))
    f, cfl, nb, _, _ = sim.o.forces(sim.pos, sim.vel, sim.info, sim.hash, sim.cs, sim.nl, n, effvisc=eff)
    K.memset(eng.forces, 0); K.memset(eng.cfl, 0)
    nb_g = K.forces_effvisc(eng.forces, eng.cfl, eng.pos, eng.vel, eng.info, eng.hash, eng.cellStart, eng.neibslist, eng.effvisc, n, 0, n, 0)
    assert nb_g == nb
    gf = _np(eng.forces)[:n]
    scale = np.abs(f[:n, :3]).max()
    assert np.abs(gf[:, :3] - f[:n, :3]).max() <= 2e-5 * scale
    assert np.abs(gf[:, 3] - f[:n, 3]).max() <= 2e-5 * max(np.abs(f[:n, 3]).max(), 1e-12)
    np.testing.assert_allclose(_np(eng.cfl)[:nb], cfl[:nb], rtol=2e-5)


@pytest.mark.parametrize("kw", [dict(rheology=D.PAPANASTASIOU), dict(rheology=D.HERSCHEL_BULKLEY, compvisc=D.DYNAMIC, viscavg=D.ARITHMETIC,
                                                                      power_law_n=0.8)])
def test_steps_follow_the_oracle(kw):
    pr = Poiseuille(12, **kw)
    sim = ol.OracleSim(pr)
    eng = _engine(Poiseuille(12, **kw))
    steps = 12                                   # crosses a neighbour rebuild
    for _ in range(steps):
        sim.step(); eng.step()
    out = eng.download()
    n = eng.n
    assert n == sim.n and np.array_equal(out["hash"], sim.hash[:n])
    cs = float(min(pr.m_cellsize))
    assert np.abs(out["pos"][:, :3] - sim.pos[:n, :3]).max() <= 1e-6 * cs * steps
    assert np.abs(out["vel"][:, :3] - sim.vel[:n, :3]).max() <= 1e-3 * max(np.abs(sim.vel[:n, :3]).max(), 1e-6)
    assert abs(eng.current_dt() - sim.dt) <= 3e-5 * sim.dt
    assert eng.k.max_kinvisc == pytest.approx(sim.max_kinvisc, rel=3e-5)
    assert np.abs(sim.vel[:n, 0]).max() > 0


def test_entry_points_refuse_what_is_not_built():
    from gpusph_amd import capi
    eng = _engine(Poiseuille(10, rheology=D.BINGHAM))
    eng.build_neibs()
    n = eng.n
    with pytest.raises(capi.SphxInvalidArgument):       # the plain forces entry does not read BUFFER_EFFVISC
        eng.k.forces(eng.forces, eng.cfl, None, None, eng.pos, eng.vel, eng.info, eng.hash, eng.cellStart, eng.neibslist, n, 0, n, 0)
    nw = _engine(Poiseuille(10))
    nw.build_neibs()
    import torch
    scratch = torch.zeros(nw.alloc, dtype=torch.float32, device=nw.device)
    with pytest.raises(capi.SphxInvalidArgument):
        nw.k.calc_effvisc(scratch, nw.pos, nw.vel, nw.info, nw.hash, nw.cellStart, nw.neibslist, nw.n, nw.n)
    pr = Poiseuille(10, rheology=D.BINGHAM)
    pr.simparams.rheologytype = D.GRANULAR
    with pytest.raises(capi.SphxUnsupported):
        _engine(pr)


@pytest.mark.parametrize("compvisc,viscavg", [(D.KINEMATIC, D.HARMONIC), (D.DYNAMIC, D.ARITHMETIC)])
def test_cpp_adapters_run_the_papanastasiou_step_like_the_python_driver(tmp_path, compvisc, viscavg):
    """example_engines (built inside the GPUSPH tree) with the framework of PoiseuillePapanastasiou: CALC_VISC through
    AbstractViscEngine::calc_visc (its return value limits dt), forces basicstep with BUFFER_EFFVISC"""
    import os, subprocess
    import host_case as hc
    exe = hc.exe("example_engines")
    assert os.path.exists(exe), "gpusph_amd/host/example_engines is not built (make -C gpusph_amd/host, needs the GPUSPH tree)"
    prob = Poiseuille(12, rheology=D.PAPANASTASIOU, compvisc=compvisc, viscavg=viscavg)
    eng = _engine(prob)
    steps = 12
    case, state, fout = tmp_path / "case.txt", tmp_path / "state.bin", tmp_path / "out.bin"
    lines = hc.case_lines(prob, "PoiseuillePapanastasiou", rhodiff=0, compvisc=compvisc, viscavg=viscavg) + hc.driver_lines(prob, eng, steps)
    case.write_text("\n".join(lines) + "\n")
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
    assert np.abs(out["vel"][:, 0]).max() > 0


HA_VISC = dict(rheologytype=D.NEWTONIAN, turbmodel=D.LAMINAR_FLOW, compvisc=D.DYNAMIC, avgop=D.HARMONIC)


@pytest.mark.parametrize("kw", [dict(density_diffusion=D.DENSITY_DIFFUSION_NONE), dict(density_diffusion=D.COLAGROSSI),
                                dict(density_diffusion=D.FERRARI), dict(two_fluids=False, viscosity=dict(rheologytype=D.INVISCID, turbmodel=D.LAMINAR_FLOW))])
def test_sph_ha_forces_and_trajectory(kw):
    """SPH_HA (Hu & Adams, BiFluidPoiseuille's formulation) through sphx_forces_basicstep, which routes it to rheology.hip's kernel"""
    import torch
    from gpusph_amd.problem import DamBreak3D
    args = dict(deltap=0.045, obstacle=False, two_fluids=True, formulation=D.SPH_HA, viscosity=HA_VISC, jitter=0.15, hydrostatic=False)
    args.update(kw)
    sim = ol.OracleSim(DamBreak3D(**args)); sim.build_neibs()
    eng = _engine(DamBreak3D(**args), clobber_neibslist=True); eng.build_neibs()
    n = sim.n
    rng = np.random.default_rng(21)
    fluid = info_type(sim.info[:n]) == D.PT_FLUID
    sim.vel[:n, :3][fluid] += rng.uniform(-0.3, 0.3, size=(fluid.sum(), 3)).astype(np.float32)
    sim.vel[:n, 3] += rng.uniform(0, 3e-3, size=n).astype(np.float32)
    eng.vel[:n].copy_(torch.from_numpy(sim.vel[:n]).to(eng.device))
    f, cfl, nb, _, _ = sim.o.forces(sim.pos, sim.vel, sim.info, sim.hash, sim.cs, sim.nl, n)
    eng._forces(eng.pos, eng.vel, 1, 0)
    gf = _np(eng.forces)[:n]
    scale = np.abs(f[:n, :3]).max()
    assert np.abs(gf[:, :3] - f[:n, :3]).max() <= 2e-5 * scale
    assert np.abs(gf[:, 3] - f[:n, 3]).max() <= 2e-5 * np.abs(f[:n, 3]).max() + 1e-7
    # differs from SPH_F1 on the same state once the densities are not uniform
    sim.o.p.sph_formulation = D.SPH_F1
    f1 = sim.o.forces(sim.pos, sim.vel, sim.info, sim.hash, sim.cs, sim.nl, n)[0]
    if args["two_fluids"]:      # with equal masses the two formulations coincide: (P_a V_a^2 + P_b V_b^2)/m = m (P_a/rho_a^2 + P_b/rho_b^2)
        assert np.abs(f1[:n, :3] - f[:n, :3]).max() > 1e-3 * scale
    sim2 = ol.OracleSim(DamBreak3D(**args)); eng2 = _engine(DamBreak3D(**args))
    steps = 8
    for _ in range(steps):
        sim2.step(); eng2.step()
    out = eng2.download()
    assert np.array_equal(out["hash"], sim2.hash[:n])
    assert np.abs(out["vel"][:, :3] - sim2.vel[:n, :3]).max() <= 1e-3 * max(np.abs(sim2.vel[:n, :3]).max(), 1e-6)


def test_cpp_adapters_with_the_bifluid_poiseuille_framework(tmp_path):
    import os, subprocess
    import host_case as hc
    from gpusph_amd.problem import DamBreak3D
    prob = DamBreak3D(0.045, obstacle=False, two_fluids=True, formulation=D.SPH_HA, viscosity=HA_VISC, jitter=0.05,
                      density_diffusion=D.COLAGROSSI)
    prob.simparams.simflags &= ~D.ENABLE_REPACKING
    # the framework of BiFluidPoiseuilleDYN is periodic in x and y; the engines do not care whether particles use it: this tank has walls
    prob.simparams.periodicbound = D.PERIODIC_X | D.PERIODIC_Y
    eng = _engine(prob)
    steps = 12
    case, state, fout = tmp_path / "case.txt", tmp_path / "state.bin", tmp_path / "out.bin"
    case.write_text("\n".join(hc.case_lines(prob, "BiFluidPoiseuilleDYN", rhodiff=D.COLAGROSSI) + hc.driver_lines(prob, eng, steps)) + "\n")
    hc.write_state(state, prob.copy_to_array())
    r = subprocess.run([hc.exe("example_engines"), str(case), str(state), str(fout)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    eng.run(steps)
    ref = eng.download()
    out = hc.read_out(fout)
    assert out["n"] == eng.n and np.float32(eng.current_dt()) == out["dt"]
    assert np.array_equal(_bits(out["pos"]), _bits(ref["pos"])) and np.array_equal(_bits(out["vel"]), _bits(ref["vel"]))


@pytest.mark.parametrize("kw", [dict(viscmodel=D.MONAGHAN, compvisc=D.KINEMATIC, viscavg=D.HARMONIC),
                                dict(viscmodel=D.MONAGHAN, compvisc=D.DYNAMIC, viscavg=D.ARITHMETIC),
                                dict(viscmodel=D.ESPANOL_REVENGA, compvisc=D.DYNAMIC, viscavg=D.ARITHMETIC, bulk_visc=0.04),
                                dict(viscmodel=D.ESPANOL_REVENGA, compvisc=D.KINEMATIC, viscavg=D.GEOMETRIC, bulk_visc=0.02)])
def test_monaghan_and_espanol_revenga_viscous_models(kw, tmp_path):
    """visc_model<MONAGHAN | ESPANOL_REVENGA>: forces vs oracle, a trajectory with the tightened viscous dt limit, and the C++
    adapters with Poiseuille.inc's framework (its fourth run-time selector) bit-equal to the Python driver"""
    import os, subprocess
    import host_case as hc
    sim, eng = _pair(lambda: Poiseuille(12, **kw), g=1.5)
    n = sim.n
    f, cfl, nb, _, _ = sim.o.forces(sim.pos, sim.vel, sim.info, sim.hash, sim.cs, sim.nl, n)
    eng._forces(eng.pos, eng.vel, 1, 0)
    gf = _np(eng.forces)[:n]
    scale = np.abs(f[:n, :3]).max()
    assert np.abs(gf[:, :3] - f[:n, :3]).max() <= 2e-5 * scale
    dt_ref = sim.o.dtreduce(cfl, nb, sim.sspeed_cfl, sim.max_kinvisc)
    assert abs(float(eng.d_dt_next.item()) - dt_ref) <= 2e-5 * dt_ref
    prob = Poiseuille(12, **kw)
    sim2 = ol.OracleSim(prob); eng2 = _engine(prob)
    steps = 12
    for _ in range(steps):
        sim2.step(); eng2.step()
    out = eng2.download()
    assert np.array_equal(out["hash"], sim2.hash[:n])
    assert np.abs(out["vel"][:, :3] - sim2.vel[:n, :3]).max() <= 1e-3 * max(np.abs(sim2.vel[:n, :3]).max(), 1e-6)
    assert abs(eng2.current_dt() - sim2.dt) <= 3e-5 * sim2.dt
    eng3 = _engine(prob)
    case, state, fout = tmp_path / "case.txt", tmp_path / "state.bin", tmp_path / "out.bin"
    lines = hc.case_lines(prob, "PoiseuilleViscModel", rhodiff=0, compvisc=kw["compvisc"], viscavg=kw["viscavg"], viscmodel=kw["viscmodel"]) + \
        hc.driver_lines(prob, eng3, steps)
    case.write_text("\n".join(lines) + "\n")
    hc.write_state(state, prob.copy_to_array())
    r = subprocess.run([hc.exe("example_engines"), str(case), str(state), str(fout)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    res = hc.read_out(fout)
    assert res["n"] == n and np.float32(eng2.current_dt()) == res["dt"]
    assert np.array_equal(_bits(res["pos"]), _bits(out["pos"])) and np.array_equal(_bits(res["vel"]), _bits(out["vel"]))
