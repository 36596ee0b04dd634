"""ENABLE_INTERNAL_ENERGY on the GPU (energy.hip) against the CPU oracle: the energy rate of a forces pass for the pair-force
options it is built for, the Euler update, whole steps, and the C++ adapters with AccuracyTest's framework."""
import numpy as np
import pytest

from gpusph_amd import defs as D
from gpusph_amd.problem import DamBreak3D, info_type
import oracle_lib as ol

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


CASES = [dict(), dict(viscosity="DYNAMICVISC", kinematic_visc=0.05), dict(boundary=D.LJ_BOUNDARY, obstacle=True),
         dict(boundary=D.MK_BOUNDARY), dict(formulation=D.SPH_F2, two_fluids=True, viscosity="KINEMATICVISC", kinematic_visc=0.03),
         dict(kerneltype=D.CUBICSPLINE, density_diffusion=D.FERRARI)]


@pytest.mark.parametrize("kw", CASES)
def test_energy_rate_of_a_forces_pass(kw):
    import torch
    args = dict(deltap=0.045, obstacle=False, jitter=0.15, hydrostatic=False, internal_energy=True)
    args.update(kw)
    sim = ol.OracleSim(DamBreak3D(**args)); sim.build_neibs()
    eng = _engine(DamBreak3D(**args), clobber_neibslist=True); eng.build_neibs()
    n = sim.n
    rng = np.random.default_rng(9)
    fluid = info_type(sim.info[:n]) == D.PT_FLUID
    sim.vel[:n, :3][fluid] += rng.uniform(-0.4, 0.4, size=(fluid.sum(), 3)).astype(np.float32)
    sim.vel[:n, 3] += rng.uniform(0, 2e-3, size=n).astype(np.float32)
    eng.vel[:n].copy_(torch.from_numpy(sim.vel[:n]).to(eng.device))
    dedt = np.zeros(len(sim.pos), dtype=np.float32)
    cof = 1 if sim.problem.simparams.numforcesbodies else 0
    sim.o.forces(sim.pos, sim.vel, sim.info, sim.hash, sim.cs, sim.nl, n, dedt=dedt, compute_object_forces=cof,
                 rb_count=getattr(sim.problem, "num_obstacle", 0))
    eng.dedt.fill_(7.0)
    eng.k.forces_internal_energy(eng.dedt, eng.pos, eng.vel, eng.info, eng.hash, eng.cellStart, eng.neibslist, n, 0, n)
    got = _np(eng.dedt)[:n]
    scale = np.abs(dedt[:n]).max()
    assert scale > 0
    # tolerance: powf of the equation of state inside every pair term, ~100 fp32 terms per particle
    assert np.abs(got - dedt[:n]).max() <= 3e-5 * scale
    t = info_type(sim.info[:n])
    if args.get("boundary", D.DYN_BOUNDARY) == D.DYN_BOUNDARY:
        assert np.abs(got[t == D.PT_BOUNDARY]).max() > 0


def test_steps_follow_the_oracle_and_the_adapters_follow_the_driver(tmp_path):
    import os, subprocess
    import host_case as hc
    kw = dict(deltap=0.045, obstacle=False, jitter=0.05, internal_energy=True)
    prob = DamBreak3D(**kw)
    prob.simparams.simflags &= ~D.ENABLE_REPACKING
    prob.simparams.densitydiffusiontype = D.DENSITY_DIFFUSION_NONE       # AccuracyTest.cu selects none
    sim = ol.OracleSim(prob)
    eng = _engine(prob)
    steps = 12
    for _ in range(steps):
        sim.step(); eng.step()
    out = eng.download()
    n = eng.n
    assert np.array_equal(out["hash"], sim.hash[:n])
    assert np.abs(out["vel"][:, :3] - sim.vel[:n, :3]).max() <= 1e-3 * max(np.abs(sim.vel[:n, :3]).max(), 1e-6)
    e = _np(eng.energy)[:n]
    scale = np.abs(sim.energy[:n]).max()
    assert scale > 0 and np.abs(e - sim.energy[:n]).max() <= 2e-3 * scale
    # the same run through the tree's interfaces (AccuracyTest framework: BUFFER_INTERNAL_ENERGY / _UPD in the BufferLists)
    exe = hc.exe("example_engines")
    assert os.path.exists(exe)
    eng2 = _engine(prob)
    case, state, fout = tmp_path / "case.txt", tmp_path / "state.bin", tmp_path / "out.bin"
    case.write_text("\n".join(hc.case_lines(prob, "AccuracyTest") + hc.driver_lines(prob, eng2, steps)) + "\n")
    hc.write_state(state, prob.copy_to_array())
    r = subprocess.run([exe, str(case), str(state), str(fout)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    res = hc.read_out(fout)
    assert res["n"] == n and np.array_equal(_bits(res["pos"]), _bits(out["pos"])) and np.array_equal(_bits(res["vel"]), _bits(out["vel"]))
    assert np.array_equal(_bits(res["energy"]), _bits(e))


def test_refusals():
    from gpusph_amd import capi
    eng = _engine(DamBreak3D(0.06, obstacle=False))
    eng.build_neibs()
    import torch
    buf = torch.zeros(eng.alloc, dtype=torch.float32, device=eng.device)
    with pytest.raises(capi.SphxInvalidArgument):
        eng.k.forces_internal_energy(buf, eng.pos, eng.vel, eng.info, eng.hash, eng.cellStart, eng.neibslist, eng.n, 0, eng.n)
    sps = _engine(DamBreak3D(0.06, obstacle=False, internal_energy=True, viscosity="SPSVISC"))
    sps.build_neibs()
    with pytest.raises(capi.SphxUnsupported):
        sps.k.forces_internal_energy(sps.dedt, sps.pos, sps.vel, sps.info, sps.hash, sps.cellStart, sps.neibslist, sps.n, 0, sps.n)
