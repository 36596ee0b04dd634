"""SA_BOUNDARY bodies with prescribed motion on the device (ENABLE_MOVING_BODIES): the kernels against the
oracle on one moved state (normals bit for bit, the density summation between the old and the new elements, gamma of the vertex
rows by both forms), and the engine's whole sequence -- body motion, Euler, normals, density summation / gamma, boundary conditions
with the elements of the new state, forces -- against the independent restatement of tests/sa_helpers.py OracleSaSim.step_moving
(which tests/test_sa_moving.py holds bit for bit against the same driver over the oracle's kernels on the CPU)."""
import numpy as np
import pytest

from gpusph_amd import defs as D
from gpusph_amd.problem import SAPaddleBox, info_type
from sa_helpers import OracleSaSim, assert_close_but_for_gamma_spikes, wall_rows
from test_sa_moving import _moved_state, _bits

pytestmark = pytest.mark.gpu


def _np(t):
    return t.cpu().numpy()


def _engine(problem, **kw):
    import torch
    from gpusph_amd.engine import TimestepEngine
    assert torch.cuda.is_available()
    return TimestepEngine(problem, device="cuda:0", **kw)


@pytest.mark.parametrize("options", ["StillWaterSA", "StillWaterRepackSA"])
def test_kernels_against_the_oracle_on_a_moved_state(options):
    import torch
    sim, ps, vs, be_new, rot = _moved_state(options)
    o, n, p = sim.o, sim.n, sim.problem
    eng = _engine(SAPaddleBox(0.05, jitter=0.1, options=options))
    eng.build_neibs()
    dev = eng.device
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    K = eng.k
    assert np.array_equal(_np(eng.info)[:n].view(np.uint16).reshape(-1, 4), sim.info[:n])      # same order as the oracle's state
    z3 = np.zeros((1, 3), np.float32)
    K.set_body_motion(dict(trans=np.array([[0.004, 0, 0]], np.float32), rot=rot.reshape(1, 9).copy(), lvel=z3, avel=z3,
                           cg_grid=np.zeros((1, 3), np.int32), cg_pos=z3), False)
    d_be, d_ben = up(sim.be), torch.zeros_like(eng.boundelements)
    K.sa_update_normals(d_ben, d_be, eng.info, n, n)
    got = _np(d_ben)[:n]
    assert np.array_equal(np.isnan(got), np.isnan(be_new[:n]))
    assert np.array_equal(_bits(got[~np.isnan(got)]), _bits(be_new[:n][~np.isnan(be_new[:n])]))
    t = info_type(sim.info[:n])
    fl, vt, bd = t == D.PT_FLUID, t == D.PT_VERTEX, t == D.PT_BOUNDARY
    wall = wall_rows(p, sim.nl, sim.info, n)
    d_pos, d_ps, d_vel, d_vs, d_gg = up(sim.pos), up(ps), up(sim.vel), up(vs), up(sim.gg)
    vp = [up(v) for v in sim.vertpos]
    if options == "StillWaterSA":
        want_v, want_g = o.sa_density_sum_moving(vs, sim.pos, ps, sim.vel, sim.gg, sim.gg, sim.be, be_new, sim.vertpos, sim.info, sim.hash,
                                                 sim.cs, sim.nl, n)
        d_ng, d_f = up(sim.gg), torch.zeros_like(d_vel)
        K.sa_density_sum_moving(d_vs, d_ng, d_f, d_pos, d_ps, d_vel, d_gg, d_be, d_ben, vp, eng.info, eng.hash, eng.cellStart, eng.neibslist, n, n)
        gv, gg = _np(d_vs)[:n], _np(d_ng)[:n]
        assert_close_but_for_gamma_spikes(gv[fl, 3], want_v[:n][fl, 3], 2e-6, 1.0, what="density after the summation", wall=wall[fl], frac=0.03)
        assert_close_but_for_gamma_spikes(gg[fl], want_g[:n][fl], 2e-5, np.abs(want_g[:n][fl, :3]).max(), what="gamma of the fluid", wall=wall[fl], frac=0.03)
        assert_close_but_for_gamma_spikes(gg[vt], want_g[:n][vt], 2e-5, np.abs(want_g[:n][vt, :3]).max(), what="gamma of the vertices", frac=0.05)
        assert np.array_equal(_bits(gg[bd]), _bits(sim.gg[:n][bd]))
    else:
        want = o.sa_integrate_gamma_moving(sim.gg, ps, be_new, sim.vertpos, sim.info, sim.hash, sim.cs, sim.nl, n)
        d_ng = torch.zeros_like(d_gg)
        K.sa_integrate_gamma(d_ng, d_gg, d_ps, d_ben, vp, eng.info, eng.hash, eng.cellStart, eng.neibslist, n, n)
        gg = _np(d_ng)[:n]
        assert np.abs(gg[fl | vt, 3] - want[:n][fl | vt, 3]).max() < 2e-6
        assert_close_but_for_gamma_spikes(gg[fl, :3], want[:n][fl, :3], 5e-5, what="grad gamma of the fluid", wall=wall[fl], frac=0.03)
        assert_close_but_for_gamma_spikes(gg[vt, :3], want[:n][vt, :3], 5e-5, what="grad gamma of the vertices", frac=0.05)
        assert np.array_equal(_bits(gg[bd]), _bits(sim.gg[:n][bd]))


@pytest.mark.parametrize("options", ["StillWaterSA", "StillWaterRepackSA"])
def test_engine_follows_the_independent_sequence(options):
    import torch
    mk = lambda: SAPaddleBox(0.05, jitter=0.1, options=options)
    sim = OracleSaSim(mk())
    eng = _engine(mk())
    assert eng.sa_moving
    for _ in range(5):
        sim.step(); eng.step()
    torch.cuda.synchronize()
    n = sim.n
    assert eng.n == n and np.array_equal(_np(eng.info)[:n].view(np.uint16).reshape(-1, 4), sim.info[:n])
    gp, gv, gg, gb = _np(eng.pos)[:n], _np(eng.vel)[:n], _np(eng.gradgamma)[:n], _np(eng.boundelements)[:n]
    # the flap: positions, velocities and normals follow the prescribed motion, to rounding (the body's rotation is uploaded as
    # float, both sides apply the same float operations)
    moving = (sim.info[:n, 0] & D.FG_MOVING_BOUNDARY) != 0
    t = info_type(sim.info[:n])
    assert np.array_equal(_bits(gp[moving, :3]), _bits(sim.pos[:n][moving, :3]))
    seg = moving & (t == D.PT_BOUNDARY)
    assert np.array_equal(_bits(gb[seg]), _bits(sim.be[:n][seg]))
    assert gb[seg, 2].max() < -1e-3                      # it did turn
    cell = float(sim.o.p.cellSize[0])
    from sa_helpers import wall_rows
    W = wall_rows(sim.problem, sim.nl, sim.info, n)
    assert_close_but_for_gamma_spikes(gp[:, :3], sim.pos[:n, :3], 2e-5, cell, frac=0.002, spike=2.0, what="positions after 5 steps (moving)", wall=W)      # measured worst 0.12 of the tolerance
    assert_close_but_for_gamma_spikes(gv[:, :3], sim.vel[:n, :3], 1e-3, max(np.abs(sim.vel[:n, :3]).max(), 1e-3), frac=0.002, spike=2.0, what="velocities after 5 steps (moving)", wall=W)      # 0.11
    assert_close_but_for_gamma_spikes(gv[:, 3], sim.vel[:n, 3], 2e-6, 1.0, spike=10.0, frac=0.03, what="densities after 5 steps (moving)", wall=W)      # 1.4 % beyond, worst 7.4; rows away from the walls 0.23
    fin = np.isfinite(sim.gg[:n, 3])
    assert_close_but_for_gamma_spikes(gg[fin, 3], sim.gg[:n][fin, 3], 5e-6, 1.0, spike=10.0, frac=0.03, what="gamma after 5 steps")
    assert abs(eng.current_dt() - sim.dt) <= 1e-4*sim.dt
