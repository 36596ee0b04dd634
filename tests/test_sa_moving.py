"""SA_BOUNDARY bodies with prescribed motion (ENABLE_MOVING_BODIES) on the CPU:
  * known answers of the oracle's restatements (update_normals; the density summation between an old and a new state of the
    elements; gamma of vertex particles): what must hold whatever the code looks like -- a rotation turns the normals of the moving
    rows only and keeps their length, elements that did not move give the solid-wall numbers bit for bit, a wall that ADVANCES on a
    particle lowers its gamma by the volume it sweeps (to first order in the displacement);
  * the kernels' source (sa_bounds.hip / euler.hip through tests/hostemu) against the oracle;
  * the driver of gpusph_amd.multigpu over the oracle's kernels against the independent restatement of the command sequence in
    tests/sa_helpers.py OracleSaSim.step_moving, bit for bit, for both forms of the continuity equation."""
import math

import numpy as np
import pytest

from gpusph_amd import defs as D
from gpusph_amd.problem import SABox, SAPaddleBox, info_type
from sa_helpers import OracleSaSim


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def _rotation_y(angle):
    c, s = math.cos(angle), math.sin(angle)
    return np.array([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]], dtype=np.float32)


def _set_motion(o, rot, trans=(0.0, 0.0, 0.0)):
    for a in range(9):
        o.p.rbsteprot[0][a] = float(np.asarray(rot, dtype=np.float32).reshape(9)[a])
    for a in range(3):
        o.p.rbtrans[0][a] = float(trans[a])


def test_normals_turn_with_the_body_and_only_with_it():
    sim = OracleSaSim(SAPaddleBox(0.05))
    o, n, p = sim.o, sim.n, sim.problem
    moving = (sim.info[:n, 0] & D.FG_MOVING_BOUNDARY) != 0
    assert 0 < moving.sum() < n
    _set_motion(o, _rotation_y(0.1))
    out = o.sa_update_normals(sim.be, sim.info, n)
    t = info_type(sim.info[:n])
    rows = moving & (t != D.PT_FLUID)
    assert np.array_equal(_bits(out[~rows]), _bits(sim.be[:n][~rows]) ) or np.array_equal(np.isnan(out[~rows]), np.isnan(sim.be[:n][~rows]))
    seg = rows & (t == D.PT_BOUNDARY)
    want = (sim.be[:n][seg, :3].astype(np.float64) @ _rotation_y(0.1).astype(np.float64).T)
    assert np.abs(out[seg, :3] - want).max() < 1e-6
    assert np.abs(np.linalg.norm(out[seg, :3], axis=1) - 1.0).max() < 1e-6
    assert np.array_equal(out[seg, 3], sim.be[:n][seg, 3])            # the area rides along
    # the flap's normal (1, 0, 0) turned about y by +0.1: it tips towards -z
    assert out[seg, 2].max() < -0.09 and out[seg, 0].min() > 0.99
    # the identity leaves everything as it is, bit for bit
    _set_motion(o, np.eye(3))
    same = o.sa_update_normals(sim.be, sim.info, n)
    assert np.array_equal(_bits(same[~np.isnan(same)]), _bits(sim.be[:n][~np.isnan(sim.be[:n])]))


def test_density_summation_between_two_states_of_the_elements():
    sim = OracleSaSim(SAPaddleBox(0.05, jitter=0.1))
    o, n, p = sim.o, sim.n, sim.problem
    t = info_type(sim.info[:n])
    fl, vt = t == D.PT_FLUID, t == D.PT_VERTEX
    # (1) nothing moved: the fluid rows equal the solid-wall summation bit for bit; gamma of the vertices stays what it was
    v0, g0 = o.sa_density_sum(sim.vel, sim.pos, sim.pos, sim.vel, sim.gg, sim.be, sim.vertpos, sim.info, sim.hash, sim.cs, sim.nl, n)
    v1, g1 = o.sa_density_sum_moving(sim.vel, sim.pos, sim.pos, sim.vel, sim.gg, sim.gg, sim.be, sim.be, sim.vertpos, sim.info, sim.hash,
                                     sim.cs, sim.nl, n)
    assert np.array_equal(_bits(v1[:n][fl]), _bits(v0[:n][fl])) and np.array_equal(_bits(g1[:n][fl]), _bits(g0[:n][fl]))
    assert np.array_equal(_bits(g1[:n][vt, 3]), _bits(sim.gg[:n][vt, 3]))      # gamma + 0
    # (2) the flap advances by d along x (a translation: normals unchanged): a fluid particle in front of it loses
    #     gamma by grad gamma . (-d) to first order -- the same number as if the particle had moved by -d towards a wall at rest
    d = 0.02*p.m_deltap
    moving = (sim.info[:n, 0] & D.FG_MOVING_BOUNDARY) != 0
    new_pos = sim.pos.copy(); new_pos[:n][moving, 0] += np.float32(d)
    v2, g2 = o.sa_density_sum_moving(sim.vel, sim.pos, new_pos, sim.vel, sim.gg, sim.gg, sim.be, sim.be, sim.vertpos, sim.info, sim.hash,
                                     sim.cs, sim.nl, n)
    gpos = p.global_pos(sim.pos[:n], sim.hash[:n])
    near = fl & (gpos[:, 0] < 1.2*p.m_deltap) & (gpos[:, 1] > 2.5*p.m_deltap) & (gpos[:, 1] < p.w - 2.5*p.m_deltap) & \
        (gpos[:, 2] > 2.5*p.m_deltap) & (gpos[:, 2] < p.water_level - 1.5*p.m_deltap)
    assert near.sum() >= 6
    dgam = g2[:n][near, 3].astype(np.float64) - sim.gg[:n][near, 3]
    # in front of a plane wall at x = 0 with inward normal +x: grad gamma = -|grad gamma| ex... gamma falls as the wall comes closer
    assert (dgam < 0).all()
    other = sim.pos.copy(); other[:n][near, 0] -= np.float32(d)
    v3, g3 = o.sa_density_sum(sim.vel, sim.pos, other, sim.vel, sim.gg, sim.be, sim.vertpos, sim.info, sim.hash, sim.cs, sim.nl, n)
    dref = g3[:n][near, 3].astype(np.float64) - sim.gg[:n][near, 3]
    assert np.abs(dgam - dref).max() < 0.08*np.abs(dref).max()      # (the side walls and the floor did not move in (2): away from them)
    far = fl & (gpos[:, 0] > 0.5*p.l)
    assert np.array_equal(_bits(g2[:n][far, 3]), _bits(g0[:n][far, 3]))


@pytest.mark.parametrize("options", ["StillWaterSA", "StillWaterRepackSA"])
def test_driver_equals_the_independent_sequence(options):
    import torch
    from gpusph_amd.multigpu import MultiGpuEngine
    from oracle_kernels import OracleKernels
    mk = lambda: SAPaddleBox(0.05, jitter=0.1, options=options)
    sim = OracleSaSim(mk())
    alloc = sim.n + 64
    eng = MultiGpuEngine(mk(), "cpu", 0, 1, kernels=OracleKernels(mk(), alloc), allocated=alloc)
    assert eng.sa_moving and eng.bodies is not None
    for it in range(5):
        eng.step(); sim.step()
        n = sim.n
        for name, got, want in (("pos", eng.pos, sim.pos), ("vel", eng.vel, sim.vel), ("gamma", eng.gradgamma, sim.gg),
                                ("boundelements", eng.boundelements, sim.be)):
            a, b = got[:n].numpy(), want[:n]
            assert np.array_equal(np.isnan(a), np.isnan(b)), (name, it)
            assert np.array_equal(a[~np.isnan(a)].view(np.uint32), b[~np.isnan(b)].view(np.uint32)), (name, it)
        assert float(np.float32(eng.current_dt())) == float(np.float32(sim.dt)), it
    # the flap did move and turn, and the fluid felt it
    p = sim.problem
    moving = (sim.info[:n, 0] & D.FG_MOVING_BOUNDARY) != 0
    seg = moving & (info_type(sim.info[:n]) == D.PT_BOUNDARY)
    assert sim.be[:n][seg, 2].max() < -1e-3 and np.abs(sim.vel[:n][seg, 0]).max() > 0.1


# ---- the kernels' source (tests/hostemu) against the oracle ---------------------------------------------------------------------
def _moved_state(options):
    """a state, a motion of the flap (rotation about y + slide), and the new state the Euler step makes of it (by the oracle)"""
    sim = OracleSaSim(SAPaddleBox(0.05, jitter=0.1, options=options))
    o, n, p = sim.o, sim.n, sim.problem
    rot = _rotation_y(0.02)
    _set_motion(o, rot, (0.004, 0.0, 0.0))
    for a in range(3):
        o.p.rblinearvel[0][a] = (0.8, 0.0, 0.0)[a]; o.p.rbangularvel[0][a] = (0.0, 4.0, 0.0)[a]
        o.p.rbcgGridPosE[0][a] = int(p.rb_cg_gridpos[0][a]); o.p.rbcgPosE[0][a] = float(p.rb_cg_pos[0][a])
    f = np.zeros_like(sim.vel)
    rng = np.random.default_rng(2)
    f[:n, :3] = rng.normal(0, 2.0, size=(n, 3)).astype(np.float32)
    ps, vs = o.euler(sim.pos, sim.vel, sim.info, sim.hash, f, n, 1.0e-3, 1)
    be_new = o.sa_update_normals(sim.be, sim.info, n)
    return sim, ps, vs, be_new, rot


@pytest.fixture(scope="module")
def emu_of():
    from hostemu_lib import Emu
    made = {}
    def get(sim):
        key = sim.problem.options
        if key not in made:
            made[key] = Emu(sim.problem.sphx_params(sim.n))
        return made[key]
    yield get
    for e in made.values():
        e.close()


def test_kernels_in_emulation_density_summation(emu_of):
    from sa_helpers import assert_close_but_for_gamma_spikes, wall_rows
    sim, ps, vs, be_new, rot = _moved_state("StillWaterSA")
    o, n, p = sim.o, sim.n, sim.problem
    emu = emu_of(sim)
    z3 = np.zeros(3, np.float32)
    emu.call("sphx_set_rb_motion", np.array([0.004, 0, 0], np.float32), rot.reshape(9).copy(), z3, z3, 1)
    got_be = np.zeros_like(sim.be)
    emu.call("sphx_sa_update_normals", got_be, sim.be, sim.info, n, n, None)
    assert np.array_equal(np.isnan(got_be[:n]), np.isnan(be_new[:n]))
    assert np.array_equal(_bits(got_be[:n][~np.isnan(got_be[:n])]), _bits(be_new[:n][~np.isnan(be_new[:n])]))
    want_v, want_g = o.sa_density_sum_moving(vs, sim.pos, ps, sim.vel, sim.gg, sim.gg, sim.be, be_new, sim.vertpos, sim.info, sim.hash,
                                             sim.cs, sim.nl, n)
    d_v, d_g, d_f = vs.copy(), sim.gg.copy(), np.zeros_like(vs)
    vp = [np.ascontiguousarray(v) for v in sim.vertpos]
    emu.call("sphx_sa_density_sum_moving", d_v, d_g, d_f, sim.pos, ps, sim.vel, sim.gg, sim.be, be_new, vp[0], vp[1], vp[2], sim.info,
             sim.hash, sim.cs, sim.nl, n, n, None)
    t = info_type(sim.info[:n])
    fl, vt, bd = t == D.PT_FLUID, t == D.PT_VERTEX, t == D.PT_BOUNDARY
    wall = wall_rows(p, sim.nl, sim.info, n)
    assert_close_but_for_gamma_spikes(d_v[:n][fl, 3], want_v[:n][fl, 3], 2e-6, 1.0, what="density after the summation", wall=wall[fl], frac=0.03)
    assert_close_but_for_gamma_spikes(d_g[:n][fl], want_g[:n][fl], 2e-5, np.abs(want_g[:n][fl, :3]).max(), what="gamma of the fluid", wall=wall[fl], frac=0.03)
    assert_close_but_for_gamma_spikes(d_g[:n][vt], want_g[:n][vt], 2e-5, np.abs(want_g[:n][vt, :3]).max(), what="gamma of the vertices", frac=0.05)
    assert np.array_equal(_bits(d_g[:n][bd]), _bits(sim.gg[:n][bd]))          # boundary rows: not written
    # the vertices of the moving flap did get another gamma than they had
    moving = (sim.info[:n, 0] & D.FG_MOVING_BOUNDARY) != 0
    assert np.abs(want_g[:n][vt & moving, 3] - sim.gg[:n][vt & moving, 3]).max() > 1e-4


def test_kernels_in_emulation_gamma_by_quadrature(emu_of):
    from sa_helpers import assert_close_but_for_gamma_spikes, wall_rows
    sim, ps, vs, be_new, rot = _moved_state("StillWaterRepackSA")
    o, n, p = sim.o, sim.n, sim.problem
    emu = emu_of(sim)
    P = emu.params
    want = o.sa_integrate_gamma_moving(sim.gg, ps, be_new, sim.vertpos, sim.info, sim.hash, sim.cs, sim.nl, n)
    got = np.zeros_like(sim.gg)
    vp = [np.ascontiguousarray(v) for v in sim.vertpos]
    emu.call("sphx_sa_integrate_gamma", got, sim.gg, ps, be_new, vp[0], vp[1], vp[2], sim.info, sim.hash, sim.cs, sim.nl, n, n,
             0.0, 1, 0.0, 5e-5, float(P.slength), float(P.influenceradius), D.SIMULATE, None)
    t = info_type(sim.info[:n])
    fl, vt, bd = t == D.PT_FLUID, t == D.PT_VERTEX, t == D.PT_BOUNDARY
    wall = wall_rows(p, sim.nl, sim.info, n)
    assert np.abs(got[:n][fl | vt, 3] - want[:n][fl | vt, 3]).max() < 2e-6
    assert_close_but_for_gamma_spikes(got[:n][fl, :3], want[:n][fl, :3], 5e-5, what="grad gamma of the fluid", wall=wall[fl], frac=0.03)
    assert_close_but_for_gamma_spikes(got[:n][vt, :3], want[:n][vt, :3], 5e-5, what="grad gamma of the vertices", frac=0.05)
    assert np.array_equal(_bits(got[:n][bd]), _bits(sim.gg[:n][bd]))          # copied
    moving = (sim.info[:n, 0] & D.FG_MOVING_BOUNDARY) != 0
    assert np.abs(want[:n][vt & ~moving, 3] - sim.gg[:n][vt & ~moving, 3]).max() > 1e-5      # a fixed vertex next to the flap sees it turn


def test_kernels_in_emulation_gamma_while_repacking_leaves_the_body_alone(emu_of):
    """REPACK with ENABLE_MOVING_BODIES in the flags: integrate_gamma_impl's repack branch (src/cuda/euler.cu:222-239) integrates
    the fluid rows only, against the elements of the state that is read, and copies the vertex and boundary rows — it never takes
    the moving-bodies branch.  (the entry point ignored run_mode and integrated the vertex rows.)"""
    sim, ps, vs, be_new, rot = _moved_state("StillWaterRepackSA")
    o, n = sim.o, sim.n
    emu = emu_of(sim)
    P = emu.params
    want = o.sa_integrate_gamma(sim.gg, ps, sim.be, sim.vertpos, sim.info, sim.hash, sim.cs, sim.nl, n)    # fluid only, old elements
    got = np.zeros_like(sim.gg)
    vp = [np.ascontiguousarray(v) for v in sim.vertpos]
    emu.call("sphx_sa_integrate_gamma", got, sim.gg, ps, sim.be, vp[0], vp[1], vp[2], sim.info, sim.hash, sim.cs, sim.nl, n, n,
             0.0, 1, 0.0, 5e-5, float(P.slength), float(P.influenceradius), D.REPACK, None)
    t = info_type(sim.info[:n])
    fl, vt, bd = t == D.PT_FLUID, t == D.PT_VERTEX, t == D.PT_BOUNDARY
    assert np.array_equal(_bits(got[:n][vt]), _bits(sim.gg[:n][vt]))          # copied, not integrated
    assert np.array_equal(_bits(got[:n][bd]), _bits(sim.gg[:n][bd]))
    assert np.abs(got[:n][fl, 3] - want[:n][fl, 3]).max() < 2e-6
