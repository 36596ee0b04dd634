"""The product's formulation of gamma / |grad gamma| of a boundary element (gpusph_amd/csrc/sa_wall_gamma.h, the code the
SA kernels run) evaluated on the CPU and held against the reference's own numbers: tests/golden/ref_gamma.npz was produced by
src/cuda/gamma.cuh compiled into oracle/_ref (tests/golden/make_ref_goldens.py).  The header is plain C++ to g++.

The product does not repeat the reference's operation order (the oracle does, and is pinned bit for bit), so the comparison
is a bound.  For gamma (a quadrature of smooth terms) the bound is rounding.  For |grad gamma| the closed form is
ill-conditioned for some configurations: there the REFERENCE's float value is itself up to ~2e-3 away from the float64 value
of its own formula, and nothing can agree with it better than that without copying its rounding.  The bound used:
  * on average the product is at least as close to the float64 value as the reference is;
  * for every element, |product - reference| <= 2e-5 of the scale + twice the reference's own distance from float64."""
import ctypes as C
import math
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


@pytest.fixture(scope="module")
def wg(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("wg") / "wall_gamma_host.so")
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-o", so, os.path.join(HERE, "wall_gamma_host.cc")],
                   check=True, capture_output=True)
    L = C.CDLL(so)
    L.wg_grad_gamma.restype = C.c_float
    L.wg_gamma.restype = C.c_float
    return L


def _p(a):
    return np.ascontiguousarray(a, dtype=np.float32).ctypes.data_as(C.c_void_p)


def _grad_gamma_float64(ns, qvb, q, h):
    """the closed form of gamma.cuh:248-370 in double precision"""
    ns = ns.astype(np.float64); q = q.astype(np.float64); qv = qvb.reshape(3, 3).astype(np.float64)
    pas = float(ns @ q); a = abs(pas)
    if a >= 2:
        return 0.0
    g = tot = inside = 0.0
    for e in range(3):
        v0, v1 = qv[e], qv[(e + 1) % 3]
        t = (v0 - v1) / np.linalg.norm(v0 - v1)
        m = np.cross(ns, t); m /= np.linalg.norm(m)
        b = float(m @ (q - v0)); c = math.hypot(pas, b)
        s0, s1 = float(-(q - v0) @ t), float(-(q - v1) @ t)
        ang = math.copysign(math.atan2(s1, abs(b)) - math.atan2(s0, abs(b)), b)
        tot += ang
        if c < 2:
            lim = math.sqrt(4 - c * c)
            s0 = math.copysign(min(abs(s0), lim), s0); s1 = math.copysign(min(abs(s1), lim), s1)
            d0 = min(math.hypot(c, s0), 2.0); d1 = min(math.hypot(c, s1), 2.0)

            def pol(s, d):
                s2 = s * s
                return s * (3 * a**4 * (-420 + 29 * d) + b**4 * (-420 + 33 * d) + 2 * a * a * (-210 * (8 + s2) + 756 * d + 19 * s2 * d)
                            + 4 * (336 + s2 * (s2 * (-21 + 2 * d) + 28 * (-5 + 3 * d)))
                            + 2 * b * b * (420 * (-2 + d) + 6 * a * a * (-105 + 8 * d) + s2 * (-140 + 13 * d)))

            def angle(s, d):
                return math.atan2(a * s, b * d) - math.atan2(s, b)

            def lg(s, d):
                return math.copysign(1, s) * math.acosh(max(d / max(c, 1e-7), 1.0))
            K = 5 * b**6 + 21 * b**4 * (8 + a * a) + 35 * b * b * a * a * (16 + a * a) + 35 * a**4 * (24 + a * a)
            g += 0.00015542474911 * (48 * a**5 * (28 + a * a) * (angle(s1, d1) - angle(s0, d0))
                                     + b * (pol(s1, d1) - pol(s0, d0) + 3 * K * (lg(s1, d1) - lg(s0, d0))))
            inside += math.copysign(math.atan2(s1, abs(b)) - math.atan2(s0, abs(b)), b)
    tt = 1 - a / 2
    g += (inside - tot) * 0.05968310365947 * tt**5 * (2 + 5 * a + 4 * a * a)
    return g / h


def test_wall_gamma_against_the_reference(wg):
    d = np.load(os.path.join(GOLD, "ref_gamma.npz"))
    h = float(d["h"]); n = len(d["q"])
    gg = np.zeros(n, np.float32); gf = np.zeros(n, np.float32); gv = np.zeros(n, np.float32); corners = np.zeros((n, 9), np.float32)
    for i in range(n):
        ns, vp = d["ns"][i], d["vp"][i]
        gg[i] = wg.wg_grad_gamma(_p(ns), _p(vp), C.c_float(h), _p(d["q"][i]))
        gf[i] = wg.wg_gamma(0, _p(ns), _p(vp), C.c_float(h), _p(d["q"][i]), _p(d["ggam"][i]), C.c_float(5e-5))
        gv[i] = wg.wg_gamma(1, _p(ns), _p(vp), C.c_float(h), _p(d["qv"][i]), _p(d["ggam"][i]), C.c_float(5e-5))
        out = np.zeros(9, np.float32)
        wg.wg_corners(_p(ns), _p(vp), C.c_float(h), out.ctypes.data_as(C.c_void_p))
        corners[i] = out
    # the corners of an element from BUFFER_VERTPOS: the reference's numbers exactly (a zero may carry the other sign)
    assert np.array_equal(corners, d["q_vb"])
    # gamma of fluid and of vertex particles, incl. the solid-angle branch of a vertex sitting on a corner
    for got, name in ((gf, "gamma_fluid"), (gv, "gamma_vertex")):
        ref = d[name]
        assert (ref != 0).sum() > 300
        assert np.abs(got - ref).max() <= 2e-7 * np.abs(ref).max(), name
    on_corner = (np.arange(n) % 8 == 0) & (d["gamma_vertex"] != 0)
    assert on_corner.sum() > 30 and np.abs(gv[on_corner] - d["gamma_vertex"][on_corner]).max() <= 1e-6
    # |grad gamma_as|
    ref = d["grad_gamma"]
    exact = np.array([_grad_gamma_float64(d["ns"][i], d["q_vb"][i], d["q"][i], h) for i in range(n)])
    scale = np.abs(ref).max()
    err_ref, err_own = np.abs(ref - exact), np.abs(gg - exact)
    assert (ref != 0).sum() > 300
    assert err_own.mean() <= err_ref.mean() and err_own.max() <= err_ref.max()
    assert (np.abs(gg - ref) <= 2e-5 * scale + 2.0 * err_ref).all()
    assert np.median(np.abs(gg - ref)) <= 3e-6 * scale
