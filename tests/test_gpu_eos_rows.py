"""The EOS rows of the forces engine written by the Euler step (sphx_eos_rows_follow_euler / sphx_eos_rows_current, include/sphx.h):
a run that lets the forces pass skip its EOS pre-pass wherever the driver can vouch for the velocity buffer is bit-identical to a
run that makes the rows in every pass -- across rebuilds, a density filter, and a rewrite of the velocities through torch."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _engine(follow, **kw):
    import torch
    from gpusph_amd.engine import TimestepEngine
    from gpusph_amd.problem import DamBreak3D
    assert torch.cuda.is_available()
    eng = TimestepEngine(DamBreak3D(0.03, obstacle=True, hydrostatic=False, **kw), device="cuda:0")
    if not follow:
        eng.k.eos_rows_follow_euler(False)
        eng._rows_follow = False
    return eng


def _bits(t, n):
    return t[:n].cpu().numpy().view(np.uint32)


def test_rows_from_the_euler_step_change_nothing():
    import torch
    from gpusph_amd import defs as D
    a, b = _engine(True), _engine(False)
    assert a._rows_follow and not b._rows_follow
    a.add_filter(D.SHEPARD_FILTER, 7); b.add_filter(D.SHEPARD_FILTER, 7)
    vouched = []
    orig = a.k.eos_rows_current
    a.k.eos_rows_current = lambda vel, n: (vouched.append(a.iterations), orig(vel, n))
    for it in range(24):
        if it == 13:      # the caller rewrites the densities through torch between two steps: the driver must not vouch for that buffer
            for e in (a, b):
                e.vel[: e.n, 3] *= 1.0005
        a.step(); b.step()
    n = a.n
    assert n == b.n
    assert np.array_equal(_bits(a.pos, n), _bits(b.pos, n))
    assert np.array_equal(_bits(a.vel, n), _bits(b.vel, n))
    assert np.array_equal(_bits(a.forces, n), _bits(b.forces, n))
    assert float(a.d_dt.item()) == float(b.d_dt.item())
    # the corrector pass of every step is vouched for, the predictor pass of the steps without a rebuild, a filter or the rewrite
    per_step = {i: vouched.count(i) for i in range(24)}
    assert all(per_step[i] >= 1 for i in range(24))
    assert per_step[0] == 1 and per_step[10] == 1 and per_step[20] == 1          # rebuilds
    assert per_step[7] == 1 and per_step[14] == 1 and per_step[21] == 1          # filter
    assert per_step[13] == 1                                                      # the rewrite through torch
    assert per_step[5] == 2 and per_step[15] == 2


def test_a_statement_about_another_buffer_is_ignored():
    """the library takes the caller's word only for the buffer and the row count the rows were made for"""
    import torch
    a, b = _engine(True), _engine(False)
    for _ in range(3):
        a.step(); b.step()
    n = a.n
    # vouch for a buffer that holds other densities: ignored, the pre-pass runs on what is handed over
    other = a.vel.clone()
    other[:n, 3] += 0.01
    for e, v in ((a, other), (b, other.clone())):
        e.k.memset(e.cfl, 0)
    a.k.eos_rows_current(other, n)              # not the buffer of the last Euler step
    a.k.forces(a.forces, a.cfl, None, None, a.pos, other, a.info, a.hash, a.cellStart, a.neibslist, n, 0, n, 0)
    b.k.forces(b.forces, b.cfl, None, None, b.pos, other, b.info, b.hash, b.cellStart, b.neibslist, n, 0, n, 0)
    torch.cuda.synchronize()
    assert np.array_equal(_bits(a.forces, n), _bits(b.forces, n))
    # and a wrong row count
    a.k.eos_rows_current(a.vel, n - 1)
    a.k.forces(a.forces, a.cfl, None, None, a.pos, a.vel, a.info, a.hash, a.cellStart, a.neibslist, n, 0, n, 0)
    b.k.forces(b.forces, b.cfl, None, None, b.pos, b.vel, b.info, b.hash, b.cellStart, b.neibslist, n, 0, n, 0)
    torch.cuda.synchronize()
    assert np.array_equal(_bits(a.forces, n), _bits(b.forces, n))
