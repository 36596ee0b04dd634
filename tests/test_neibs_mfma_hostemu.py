"""The list build with its distance tests on the matrix cores (gpusph_amd/csrc/neibs_build.hip, round 6), the kernel's SOURCE run
on the CPU by tests/hostemu (64 fibres per wave; v_mfma_f32_32x32x2_f32, v_permlane32_swap and v_alignbit emulated by their
documented lane layouts) against the oracle's list: every entry bit for bit, the section lengths and the counters.  This holds the
grouping of a wave's lanes by grid row, the candidate ranges, the order of emission, the encodings and the hand-over of unusual
lanes to the general walk; the device run (tests/test_gpu_parity.py and the full-size cases) holds the matrix unit itself."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gpusph_amd.defs as D
from gpusph_amd.problem import DamBreak3D, PeriodicBox
import oracle_lib as ol
from hostemu_lib import Emu


def _cell_fluid_end(sim):
    """first non-fluid particle of every cell (cell_fluid_end_kernel, neibs.hip)"""
    cs, ce = np.asarray(sim.cs).astype(np.int64), np.asarray(sim.ce).astype(np.int64)
    fe = np.asarray(sim.cs).copy()
    ptype = np.asarray(sim.info).view(np.uint16).reshape(-1, 4)[:, 0] & 7
    fluid_prefix = np.concatenate([[0], np.cumsum(ptype[:sim.n] == D.PT_FLUID)])
    ne = np.nonzero(cs != 0xFFFFFFFF)[0]
    fe[ne] = (cs[ne] + (fluid_prefix[ce[ne]] - fluid_prefix[cs[ne]])).astype(np.uint32)
    return fe


def _emulated_list(sim, env=None):
    p = sim.problem
    os.environ["SPHX_NEIBS_MFMA"] = "1"      # read when the context is created
    try:
        emu = Emu(p.sphx_params(sim.alloc))
    finally:
        del os.environ["SPHX_NEIBS_MFMA"]
    fn = emu.lib.emu_neibs_list
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p] * 8 + [C.c_uint32, C.c_uint32, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
    nl = np.zeros_like(np.asarray(sim.nl))
    counts = np.zeros(sim.alloc, dtype=np.uint32)
    mx, tot = C.c_int(0), C.c_ulonglong(0)
    fe = _cell_fluid_end(sim)
    sq = float(np.float32(p.simparams.nlSqInfluenceRadius))
    a = [np.ascontiguousarray(x) for x in (sim.pos, sim.info, sim.hash, sim.cs, sim.ce, fe)]
    env = dict(env or {})
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        rc = fn(emu.h, nl.ctypes.data, *[x.ctypes.data for x in a], sim.alloc, sim.n, sq, counts.ctypes.data, C.addressof(mx), C.addressof(tot))
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert rc == 0, emu.lib.sphx_last_error()
    emu.close()
    return nl, counts, mx.value, tot.value


def _check(sim, nl, counts, mx, tot):
    rows = int(sim.problem.sphx_params(sim.alloc).neiblistsize)
    want = np.asarray(sim.nl).view(np.uint16).reshape(rows, -1)[:, :sim.n]
    got = nl.view(np.uint16).reshape(rows, -1)[:, :sim.n]
    # the written part of every list: up to and including the terminators
    end = want == 0xFFFF
    nF = np.argmax(end, axis=0)
    nbp = int(sim.problem.sphx_params(sim.alloc).neibboundpos)
    nB = np.argmax(end[nbp::-1], axis=0)
    slot = np.arange(rows)[:, None]
    written = (slot <= nF[None, :]) | ((slot <= nbp) & (slot >= nbp - nB[None, :]))
    bad = np.nonzero((want != got) & written)
    assert len(bad[0]) == 0, "first differing (slot, particle): %s; want %s got %s" % (
        [(int(bad[0][k]), int(bad[1][k])) for k in range(min(5, len(bad[0])))], want[bad][:5], got[bad][:5])
    assert np.array_equal(counts[:sim.n] & 0xFFFF, nF) and np.array_equal(counts[:sim.n] >> 16, nB)
    info = sim.neibs_info
    assert mx == int(info.maxFluidBoundaryNeibs) and tot == int(info.numInteractions)


@pytest.mark.parametrize("lin", ["xzy", "yzx", "xyz", "yxz"])
def test_dam_break_lists_by_the_matrix_cores_equal_the_oracle(lin):
    prob = DamBreak3D(0.04, obstacle=True, linearization=lin)
    sim = ol.OracleSim(prob)
    for _ in range(3):
        sim.step()
    sim.iterations = 10
    sim.build_neibs()
    _check(sim, *_emulated_list(sim))


@pytest.mark.parametrize("case", ["lj-testpoints", "mk", "periodic-yz"])
def test_option_sets_by_the_matrix_cores_equal_the_oracle(case):
    """repulsive boundaries (boundary particles list no boundary neighbours; MK does), test points (they build lists and are nobody's
    neighbour), periodic faces across COORD2 / COORD3 (the wrapped rows).  Cases in which
    some lane is left to the general walk cannot run here: that walk relies on the execution mask around its ballots, which fibres
    do not have -- the device suite covers them (periodic COORD1, COORD1 = z, SA, the Gaussian kernel's cells of 60 particles whose rows exceed the prepass's six tiles)"""
    if case == "lj-testpoints":
        prob = DamBreak3D(0.05, boundary=D.LJ_BOUNDARY, linearization="yzx", testpoints=((0.5, 0.3, 0.1), (1.2, 0.33, 0.2)))
    elif case == "mk":
        prob = DamBreak3D(0.05, boundary=D.MK_BOUNDARY, linearization="xzy")
    else:
        prob = PeriodicBox(periodic=D.PERIODIC_Y | D.PERIODIC_Z, linearization="xzy", velocity=(0.0, 0.3, -0.2))
    sim = ol.OracleSim(prob)
    for _ in range(2):
        sim.step()
    sim.iterations = 10
    sim.build_neibs()
    _check(sim, *_emulated_list(sim))
