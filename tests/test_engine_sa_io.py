"""The driver sequence of open boundaries in gpusph_amd.multigpu (ENABLE_INLET_OUTLET: imposed values, the condition passes with
`step`, FIND_OUTGOING_SEGMENT / DISABLE_OUTGOING_PARTS in the last step, BUFFER_EULERVEL and BUFFER_NEXTID through the re-sort, a
particle count that grows inside the allocation and shrinks at the re-sort, the water depth between the forces and the imposed
pressure) run on the CPU over the oracle's kernels (tests/oracle_kernels.py) and held, bit for bit, against the step-by-step
restatement of the reference's command sequence in tests/sa_helpers.py OracleSaIoSim -- written independently of the engine, on
numpy arrays, for tests/test_sa_io_oracle.py.  The same driver runs the HIP kernels on a GPU once those are verified there."""
import numpy as np
import pytest
import torch

from gpusph_amd import defs as D
from gpusph_amd.multigpu import MultiGpuEngine
from gpusph_amd.problem import SAChannelIO, info_type
from oracle_kernels import OracleKernels
from sa_helpers import OracleSaIoSim


def _engine(problem, alloc):
    return MultiGpuEngine(problem, "cpu", 0, 1, kernels=OracleKernels(problem, alloc), allocated=alloc)


def test_open_channel_driver_equals_the_oracle_sequence():
    mk = lambda: SAChannelIO(0.05, U=0.6)
    sim = OracleSaIoSim(mk(), 0.6, brezzi=True, water_depth=True)
    eng = _engine(mk(), sim.cap)
    assert eng.io and eng.num_open_vertices == sim.num_open_vertices
    for it in range(30):
        eng.step(); sim.step()
        n = sim.n
        assert eng.n_local == n, it
        assert eng.io_created == sim.created
        # rows in the order the two hold them: both append released particles in vertex order and re-sort stably
        assert np.array_equal(eng.info[:n].numpy().view(np.uint16), sim.info[:n]), it
        assert np.array_equal(eng.hash[:n].numpy().view(np.uint32), sim.hash[:n]), it
        for name, got, want in (("pos", eng.pos, sim.pos), ("vel", eng.vel, sim.vel), ("gamma", eng.gradgamma, sim.gg),
                                ("eulervel", eng.eulervel, sim.ev)):
            a, b = got[:n].numpy(), want[:n]
            assert np.array_equal(np.isnan(a), np.isnan(b)), (name, it)
            assert np.array_equal(a[~np.isnan(a)].view(np.uint32), b[~np.isnan(b)].view(np.uint32)), (name, it)
        assert np.array_equal(eng.vertices[:n].numpy().view(np.uint32), sim.vertices[:n]), it
        assert np.array_equal(eng.next_ids[:n].numpy().view(np.uint32), sim.next_ids[:n]), it
        assert float(np.float32(eng.current_dt())) == float(np.float32(sim.dt)), it
    assert sim.created > 0 and eng.time() == pytest.approx(sim.t, rel=1e-6)
    # particles leave through the outlet later than 30 steps in this tank; what the re-sort drops is counted all the same
    assert eng.io_removed == sim.removed


def test_particles_that_leave_are_dropped_at_the_next_rebuild():
    """a faster stream, so that the first layer crosses the outlet within the run"""
    mk = lambda: SAChannelIO(0.05, U=2.0, l=0.5, w=0.25, h=0.3, H=0.2)
    sim = OracleSaIoSim(mk(), 2.0, brezzi=True, water_depth=True)
    eng = _engine(mk(), sim.cap)
    for it in range(90):
        eng.step(); sim.step()
        assert eng.n_local == sim.n
        if sim.removed > 2:
            break
    assert sim.removed > 0
    eng.build_neibs()          # the particles disabled in the last step are dropped by the next re-sort
    assert eng.io_removed == sim.removed
    t = info_type(eng.info[:eng.n_local].numpy().view(np.uint16))
    assert np.isfinite(eng.pos[:eng.n_local].numpy()[t == D.PT_FLUID]).all()


def test_open_boundaries_need_what_the_driver_is_built_for():
    p = SAChannelIO(0.05)
    p.simparams.buildneibsfreq = 10
    with pytest.raises(NotImplementedError):
        _engine(p, p.num_particles + 4096)


def test_flux_through_the_open_boundaries():
    """FLUX_COMPUTATION: the inlet carries U over its whole wall (the imposed u_E is the same on dry segments), the outlet draws
    about what the wet part of the inlet delivers"""
    p = SAChannelIO(0.05, U=0.6)
    eng = _engine(p, p.num_particles + 4096)
    for _ in range(12):
        eng.step()
    flux = eng.open_boundary_flux().numpy()
    assert flux.shape == (2,)
    assert flux[0] == pytest.approx(0.6 * p.w * p.h, rel=1e-5)
    assert -1.3 * 0.6 * p.w * p.water_level < flux[1] < -0.6 * 0.6 * p.w * p.water_level
    # by hand
    n = eng.n_local
    info = eng.info[:n].numpy().view(np.uint16)
    seg = (info_type(info) == D.PT_BOUNDARY) & ((info[:, 0] & D.FG_OUTLET) != 0)
    be, ev = eng.boundelements[:n].numpy()[seg].astype(np.float64), eng.eulervel[:n].numpy()[seg].astype(np.float64)
    assert flux[1] == pytest.approx(float((be[:, 3] * (ev[:, :3] * be[:, :3]).sum(axis=1)).sum()), rel=1e-5)


def test_private_post_processing_is_the_problems_hook():
    """CALC_PRIVATE: the engine hands itself to Problem.calc_private and returns its rows; a problem without one is told so"""
    p = SAChannelIO(0.05)
    eng = _engine(p, p.num_particles + 4096)
    eng.step()
    with pytest.raises(NotImplementedError):
        eng.calc_private()
    p.calc_private = lambda e: (e.vel[:e.n_local, :3] ** 2).sum(dim=1).sqrt()          # e.g. the speed
    speed = eng.calc_private().numpy()
    n = eng.n_local
    fl = info_type(eng.info[:n].numpy().view(np.uint16)) == D.PT_FLUID
    assert speed.shape == (n,) and abs(np.median(speed[fl]) - 0.6) < 0.02
