"""ENABLE_INLET_OUTLET | ENABLE_DENSITY_SUM | ENABLE_MOVING_BODIES, the option set of the reference's CompleteSaExample.cu (:46), in the
one pass where the two features meet: the density summation (io_gamma_contrib inside the moving boundary loop,
src/cuda/density_sum_kernel.cu:422-484).  The oracle's restatement is pinned to the two restatements it combines (nothing moved: the
open-boundary pass bit for bit; nothing open: the moving-bodies pass bit for bit), the kernel's source (sphx_sa_density_sum_io_moving,
sa_density_sum_kernel<OPEN, MOVING>) runs against it on the CPU (tests/hostemu); the device run is tests/test_gpu_sa_io.py."""
import numpy as np
import pytest

from gpusph_amd import defs as D
from gpusph_amd.problem import SABox, info_type
from sa_helpers import sa_oracle_state, wall_rows, assert_close_but_for_gamma_spikes


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def io_moving_state(open_face=True, turned=True, deltap=0.05):
    """an SABox whose x = 0 wall is a velocity-driven inlet with a stream U through it and whose x = l wall is a body that turned by a
    small angle about y and slid along x between the two states: every input of the pass"""
    p = SABox(deltap)
    p.simparams.simflags |= D.ENABLE_INLET_OUTLET | D.ENABLE_MOVING_BODIES
    st = sa_oracle_state(p)
    o, n, dp = st["oracle"], st["n"], p.m_deltap
    U, dt = 0.2, 1.0e-3
    g = p.global_pos(st["pos"], st["hash"])
    t = info_type(st["info"])
    fl = t == D.PT_FLUID
    info = st["info"].copy()
    wall0 = np.abs(g[:, 0]) < 1e-6
    seg = (t == D.PT_BOUNDARY) & (st["boundelements"][:, 0] > 0.5) & wall0
    vtx = (t == D.PT_VERTEX) & wall0
    if open_face:
        info[seg | vtx, 0] |= D.FG_INLET | D.FG_VELOCITY_DRIVEN
        info[seg | vtx, 1] = (info[seg | vtx, 1] & 0xF000) | 1
    wall1 = np.abs(g[:, 0] - p.l) < 1e-6
    body = ((t == D.PT_BOUNDARY) & (st["boundelements"][:, 0] < -0.5) & wall1) | ((t == D.PT_VERTEX) & wall1)
    info[body, 0] |= D.FG_MOVING_BOUNDARY
    be = o.sa_compute_vertex_normal(st["boundelements"], st["vertices"], st["info"], st["hash"], st["cs"], st["nl"], n)
    gg = o.sa_init_gamma(st["gradgamma"], st["pos"], be, st["vertpos"], st["info"], st["hash"], st["cs"], st["nl"], n, dp)
    vel = st["vel"].copy(); vel[fl, 0] = U
    ev = np.zeros_like(vel); ev[seg | vtx, 0] = U if open_face else 0.0
    new_pos = st["pos"].copy(); new_pos[fl, 0] = st["pos"][fl, 0] + np.float32(dt) * np.float32(U)
    be_new = be.copy()
    if turned:
        a = 0.02
        c, s = np.float32(np.cos(a)), np.float32(np.sin(a))
        nx, nz = be[body, 0].copy(), be[body, 2].copy()
        be_new[body, 0] = c * nx + s * nz
        be_new[body, 2] = -s * nx + c * nz
        new_pos[body, 0] = st["pos"][body, 0] - np.float32(0.02 * dp)
    return dict(st=st, p=p, o=o, n=n, info=info, fl=fl, vt=(t == D.PT_VERTEX), bd=(t == D.PT_BOUNDARY), body=body, be=be, be_new=be_new,
                gg=gg, vel=vel, ev=ev, new_pos=new_pos, dt=dt)


def test_the_combined_pass_reduces_to_the_two_passes_it_combines():
    # nothing moved: the open-boundary pass, fluid rows bit for bit
    c = io_moving_state(turned=False)
    st, o, n = c["st"], c["o"], c["n"]
    A = (c["info"], st["hash"], st["cs"], st["nl"], n, c["dt"])
    v1, g1, s1 = o.sa_density_sum_io_moving(c["vel"], st["pos"], c["new_pos"], c["vel"], c["ev"], c["gg"], c["be"], c["be_new"], st["vertpos"], *A)
    v0, g0, s0 = o.sa_density_sum_io(c["vel"], st["pos"], c["new_pos"], c["vel"], c["ev"], c["gg"], c["be"], st["vertpos"], *A)
    fl = c["fl"]
    assert np.array_equal(_bits(v1[fl]), _bits(v0[fl])) and np.array_equal(_bits(g1[fl]), _bits(g0[fl])) and np.array_equal(_bits(s1[fl]), _bits(s0[fl]))
    assert np.abs(s1[fl]).max() > 0 and (c["info"][:n, 0] & D.FG_INLET).any()
    # nothing open: the moving-bodies pass, fluid and vertex rows bit for bit
    c = io_moving_state(open_face=False)
    st, o, n = c["st"], c["o"], c["n"]
    v1, g1, s1 = o.sa_density_sum_io_moving(c["vel"], st["pos"], c["new_pos"], c["vel"], c["ev"], c["gg"], c["be"], c["be_new"], st["vertpos"],
                                            c["info"], st["hash"], st["cs"], st["nl"], n, c["dt"])
    v0, g0 = o.sa_density_sum_moving(c["vel"], st["pos"], c["new_pos"], c["vel"], c["gg"], c["gg"], c["be"], c["be_new"], st["vertpos"],
                                     c["info"], st["hash"], st["cs"], st["nl"], n)
    rows = c["fl"] | c["vt"]
    assert np.array_equal(_bits(v1[c["fl"]]), _bits(v0[c["fl"]])) and np.array_equal(_bits(g1[rows]), _bits(g0[rows]))
    # ... and the turn of the body is felt: gamma of the vertices next to it changes
    assert np.abs(g1[c["vt"], 3] - c["gg"][c["vt"], 3]).max() > 1e-5


def test_kernel_source_in_emulation():
    from hostemu_lib import Emu
    c = io_moving_state()
    st, o, n, p = c["st"], c["o"], c["n"], c["p"]
    emu = Emu(p.sphx_params(n))
    vp = [np.ascontiguousarray(v) for v in st["vertpos"]]
    want_v, want_g, want_s = o.sa_density_sum_io_moving(c["vel"], st["pos"], c["new_pos"], c["vel"], c["ev"], c["gg"], c["be"], c["be_new"],
                                                        st["vertpos"], c["info"], st["hash"], st["cs"], st["nl"], n, c["dt"])
    d_nv, d_ng, d_f = c["vel"].copy(), c["gg"].copy(), np.zeros_like(c["vel"])
    emu.call("sphx_sa_density_sum_io_moving", d_nv, d_ng, d_f, st["pos"], c["new_pos"], c["vel"], c["ev"], c["gg"], c["be"], c["be_new"],
             vp[0], vp[1], vp[2], c["info"], st["hash"], st["cs"], st["nl"], n, n, float(np.float32(c["dt"])), None)
    fl, vt, bd = c["fl"], c["vt"], c["bd"]
    wall = wall_rows(p, st["nl"], c["info"], n)
    assert np.abs(d_f[fl, 3] - want_s[fl]).max() < 2e-5 * np.abs(want_s[fl]).max() + 1e-3
    assert_close_but_for_gamma_spikes(d_nv[fl, 3], want_v[fl, 3], 2e-6, 1.0, what="density after the summation", wall=wall[fl], frac=0.03)
    assert_close_but_for_gamma_spikes(d_ng[fl], want_g[fl], 2e-5, np.abs(want_g[fl, :3]).max(), what="gamma of the fluid", wall=wall[fl], frac=0.03)
    assert_close_but_for_gamma_spikes(d_ng[vt], want_g[vt], 2e-5, np.abs(want_g[vt, :3]).max(), what="gamma of the vertices", frac=0.05)
    assert np.array_equal(_bits(d_ng[bd]), _bits(c["gg"][bd]))
    # the entry points of the single features say where to go with this option set
    with pytest.raises(RuntimeError, match="sphx_sa_density_sum_io_moving"):
        emu.call("sphx_sa_density_sum_moving", d_nv, d_ng, d_f, st["pos"], c["new_pos"], c["vel"], c["gg"], c["be"], c["be_new"],
                 vp[0], vp[1], vp[2], c["info"], st["hash"], st["cs"], st["nl"], n, n, None)
    emu.close()
