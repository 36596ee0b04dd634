"""The OpenChannel mirror (gpusph_amd.problem.OpenChannel; options and dimensions of src/problems/OpenChannel.cu) through the
engine's driver on the CPU backend: a gravity-driven stream over a DYN_BOUNDARY bed, periodic along the stream."""
import numpy as np
import pytest

from gpusph_amd import defs as D
from gpusph_amd.multigpu import MultiGpuEngine
from gpusph_amd.problem import OpenChannel, info_type
from oracle_kernels import OracleKernels


@pytest.mark.parametrize("sidewalls", [True, False])
def test_the_stream_starts_to_run_downhill(sidewalls):
    p = OpenChannel(0.05, sidewalls=sidewalls)
    sp, pp = p.simparams, p.physparams
    assert sp.periodicbound == (D.PERIODIC_X if sidewalls else D.PERIODIC_X | D.PERIODIC_Y)
    assert sp.rheologytype == D.NEWTONIAN and sp.turbmodel == D.LAMINAR_FLOW and sp.compvisc == D.KINEMATIC and sp.avgop == D.HARMONIC
    assert pp.kinematicvisc[0] == pytest.approx(110.0 / 2650.0) and pp.gammacoeff[0] == 2.0 and pp.sscoeff[0] == 20.0
    assert p.dyn_layers == int(np.ceil(sp.influenceRadius / p.m_deltap)) + 1
    # the lattice wraps seamlessly along the stream
    x = np.unique(np.round(p.parts.pos_global[:, 0] / p.m_deltap, 6))
    assert x[0] == 0.5 and x[-1] == round(p.l / p.m_deltap) - 0.5 and abs(p.m_size[0] - p.l) < 1e-12 and p.m_origin[0] == 0.0
    alloc = p.num_particles + 4096
    eng = MultiGpuEngine(p, "cpu", 0, 1, kernels=OracleKernels(p, alloc), allocated=alloc)
    steps = 12
    for _ in range(steps):
        eng.step()
    n = eng.n_local
    assert n == p.num_particles
    t = info_type(eng.info[:n].numpy().view(np.uint16))
    fl = t == D.PT_FLUID
    vel = eng.vel[:n].numpy()
    assert np.isfinite(vel).all() and np.isfinite(eng.pos[:n].numpy()).all()
    T = eng.time()
    gx = pp.gravity[0]
    g = p.global_pos(eng.pos[:n].numpy(), eng.hash[:n].numpy().view(np.uint32))
    core = fl & (g[:, 2] > 0.2) & (g[:, 2] < p.H - 0.1) & ((g[:, 1] > 0.3) & (g[:, 1] < p.a - 0.3) if sidewalls else True)
    assert core.sum() > 200
    # away from the bed (and the walls) the fluid has only felt gravity's component along the bed so far
    assert vel[core, 0].mean() == pytest.approx(gx * T, rel=0.1)
    # (the lattice settles under its own weight meanwhile: acoustic, a thousandth of the sound speed)
    assert np.abs(vel[core, 2]).max() < 1e-3 * pp.sscoeff[0] and np.abs(vel[core, 1]).max() < 1e-3 * pp.sscoeff[0]
    assert abs(vel[core, 1].mean()) < 0.05 * gx * T
    # the bed holds: the layer next to it is slower
    low = fl & (g[:, 2] < 1.5 * p.m_deltap)
    assert vel[low, 0].mean() < vel[core, 0].mean()
    assert (vel[t == D.PT_BOUNDARY, :3] == 0).all()
