"""The float64 arbitration of wall rows (tests/sa_helpers.py assert_wall_rows_no_farther_from_float64) on the CPU: the SA forces
of the kernels' own source (sa_bounds.hip's list walker, run by tests/hostemu) against the oracle on a jittered tank.  Rows
without a boundary element in reach hold the plain tolerance; every row beyond it is a wall row whose difference the float64
value of |grad gamma_as| of its elements explains -- the product is no farther from that value than the oracle is.  The GPU test
of the same name in tests/test_gpu_sa.py asks the same of the device build."""
import ctypes as C

import numpy as np

from gpusph_amd import defs as D
from gpusph_amd.problem import SABox, info_type
from hostemu_lib import Emu
from sa_helpers import OracleSaSim, assert_wall_rows_no_farther_from_float64, wall_rows


def test_wall_rows_beyond_tolerance_are_explained_by_float64_grad_gamma():
    sim = OracleSaSim(SABox(deltap=0.05, jitter=0.15, options="StillWaterRepackSA"))
    n, o, p = sim.n, sim.o, sim.problem
    emu = Emu(p.sphx_params(n))
    P = emu.params
    fl = np.where(info_type(sim.info) == D.PT_FLUID)[0]
    want, cfl, nb = o.forces_sa(sim.pos, sim.vel, sim.info, sim.hash, sim.cs, sim.nl, sim.gg, sim.be, sim.vertpos, n, p.m_deltap)
    got = np.zeros_like(sim.vel)
    d_cfl = np.zeros(4*nb + 64, dtype=np.float32)
    hnb = C.c_uint32(0)
    vp = [np.ascontiguousarray(v) for v in sim.vertpos]
    emu.call("sphx_forces_basicstep_sa", got, d_cfl, None, sim.pos, sim.vel, sim.info, sim.hash, sim.cs, sim.nl, sim.gg, sim.be,
             vp[0], vp[1], vp[2], n, 0, n, float(np.float32(p.m_deltap)), float(P.slength), float(P.dtadaptfactor),
             float(P.influenceradius), 0, D.SIMULATE, 1, 0.0, C.addressof(hnb), None)
    emu.close()
    assert hnb.value == nb
    wall = wall_rows(p, sim.nl, sim.info, n)
    sx = np.abs(want[fl, :3]).max(); sw = max(np.abs(want[fl, 3]).max(), 1e-3)
    # a tolerance at which the conditioning of |grad gamma| shows: a tenth of the bar of the GPU test; the walker adds in the order of the oracle, so all that differs is the formulation of |grad gamma|
    tol = 1e-5
    err = np.abs(got[fl].astype(np.float64) - want[fl])/np.array([sx, sx, sx, sw])
    away = ~wall[fl]
    assert away.sum() > 30 and err[away].max() <= tol, "rows away from the walls: %g" % err[away].max()
    arbitrated, total = assert_wall_rows_no_farther_from_float64(sim, got, want, fl, tol, sx, sw, what="SA forces (emulated walker)")
    print("wall rows beyond %.0e of the scale, arbitrated by float64: %d of %d fluid rows; largest difference %.2g of the scale"
          % (tol, arbitrated, total, err.max()))
    assert arbitrated > 0, "nothing exceeded the tolerance: the arbitration was not exercised"
