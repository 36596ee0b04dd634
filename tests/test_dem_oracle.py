"""ENABLE_DEM, CPU side: the oracle's restatement of the terrain lookup (a linearly filtered, clamped 2D texture in the
reference) and of DemLJForce against closed forms on a planar terrain."""
import ctypes as C
import math
import numpy as np
import pytest

from gpusph_amd import defs as D
from gpusph_amd.problem import DamBreak3D, info_type
import oracle_lib as ol


def dem_problem(deltap=0.04, **kw):
    args = dict(obstacle=False, boundary=D.LJ_BOUNDARY, walls="planes", dem=True, hydrostatic=False, jitter=0.1)
    args.update(kw)
    return DamBreak3D(deltap, **args)


def test_parameters_follow_compute_dem_physparams():
    pr = dem_problem()
    pp, sp = pr.physparams, pr.simparams
    assert sp.simflags & D.ENABLE_DEM and sp.simflags & D.ENABLE_PLANES and len(pr.planes) == 4      # sides only
    assert pp.ewres == pytest.approx(1.6 / 32) and pp.nsres == pytest.approx(0.67 / 14)
    assert pp.demdx == pytest.approx(pp.ewres / 5) and pp.demdy == pytest.approx(pp.nsres / 5)
    assert pp.demzmin == pytest.approx(5 * pr.m_deltap)
    p = pr.sphx_params(pr.num_particles)
    assert abs(p.ewres - pp.ewres) < 1e-9 and abs(p.demzmin - pp.demzmin) < 1e-9
    plain = DamBreak3D(0.04, obstacle=False).sphx_params(100)
    assert math.isnan(plain.ewres) and math.isnan(plain.demzmin)              # PhysParams leaves them unset without a DEM


def test_texture_lookup_is_a_clamped_bilinear_filter_with_8_bit_weights():
    rng = np.random.default_rng(4)
    dem = rng.uniform(0, 1, size=(7, 9)).astype(np.float32)
    L = ol.lib()
    L.orc_set_dem(ol.P(dem), C.c_int(9), C.c_int(7))
    # sample centres: texel (i, j) is at (i + 0.5, j + 0.5)
    for j in range(7):
        for i in range(9):
            assert float(L.orc_dem_interpol(C.c_float(i + 0.5), C.c_float(j + 0.5))) == dem[j, i]
    # between samples: bilinear to the 1/256 quantisation of the weights
    for _ in range(200):
        x, y = rng.uniform(0.5, 8.5), rng.uniform(0.5, 6.5)
        i, j = int(math.floor(x - 0.5)), int(math.floor(y - 0.5))
        a, b = x - 0.5 - i, y - 0.5 - j
        i1, j1 = min(i + 1, 8), min(j + 1, 6)
        want = (1 - a) * (1 - b) * dem[j, i] + a * (1 - b) * dem[j, i1] + (1 - a) * b * dem[j1, i] + a * b * dem[j1, i1]
        assert abs(float(L.orc_dem_interpol(C.c_float(x), C.c_float(y))) - want) <= 2.0 / 512 + 1e-6
    # clamped addressing
    assert float(L.orc_dem_interpol(C.c_float(-3.0), C.c_float(2.5))) == dem[2, 0]
    assert float(L.orc_dem_interpol(C.c_float(40.0), C.c_float(-1.0))) == dem[0, 8]


def _forces(pr, sim):
    n = sim.n
    return sim.o.forces(sim.pos, sim.vel, sim.info, sim.hash, sim.cs, sim.nl, n)[0][:n]


def test_planar_terrain_repels_along_its_normal_with_the_lj_law():
    pr = dem_problem(jitter=0.0)
    # replace the hills by an inclined plane z = z00 + sx x + sy y; put the water column right above it
    L = pr.m_size
    nrows, ncols = pr.dem.shape
    x = np.arange(ncols) * pr.physparams.ewres
    y = np.arange(nrows) * pr.physparams.nsres
    sx, sy, z00 = 0.03, -0.02, 0.085
    pr.dem = (z00 + sx * x[None, :] + sy * y[:, None]).astype(np.float32)
    sim = ol.OracleSim(pr)
    sim.build_neibs()
    n = sim.n
    g = pr.global_pos(sim.pos[:n], sim.hash[:n])
    with_dem = _forces(pr, sim)
    sim.o.p.simflags &= ~D.ENABLE_DEM
    without = _forces(pr, sim)
    sim.o.p.simflags |= D.ENABLE_DEM
    got = (with_dem - without)[:, :3].astype(np.float64)
    nrm = np.array([-sx, -sy, 1.0]); nrm /= np.linalg.norm(nrm)
    r = (g[:, 2] - (z00 + sx * g[:, 0] + sy * g[:, 1])) * nrm[2]            # distance to the plane
    pp = pr.physparams
    r0, Dc = pp.r0, pp.dcoeff
    t = info_type(sim.info[:n])
    above = g[:, 2] - (z00 + sx * g[:, 0] + sy * g[:, 1])
    acts = (t == D.PT_FLUID) & (r < r0) & (above < pp.demzmin)
    assert acts.sum() > 30
    want = np.zeros_like(got)
    lj = Dc * ((r0 / r[acts]) ** pp.p1coeff - (r0 / r[acts]) ** pp.p2coeff) / (r[acts] ** 2)
    want[acts] = (lj * r[acts])[:, None] * nrm[None, :]
    # tolerance: the 1/256 weights of the filter tilt the tangent plane by < 1e-3; the LJ law is steep near r0
    scale = np.abs(want).max()
    assert scale > 1.0
    assert np.abs(got[acts] - want[acts]).max() < 0.03 * scale
    assert np.abs(got[~acts]).max() < 1e-6 * scale                           # nothing beyond r0 or above demzmin
    # direction: along the terrain normal
    gn = got[acts] / np.linalg.norm(got[acts], axis=1, keepdims=True)
    big = np.linalg.norm(got[acts], axis=1) > 1e-2 * scale
    assert np.abs(gn[big] - nrm).max() < 5e-3


def test_hilly_terrain_keeps_the_water_above_it():
    pr = dem_problem(0.05)
    sim = ol.OracleSim(pr)
    for _ in range(30):
        sim.step()
    n = sim.n
    g = pr.global_pos(sim.pos[:n], sim.hash[:n])
    assert np.isfinite(g).all()
    ew, ns = pr.physparams.ewres, pr.physparams.nsres
    L = ol.lib()
    z0 = np.array([float(L.orc_dem_interpol(C.c_float(x / ew + 0.5), C.c_float(y / ns + 0.5))) for x, y in g[:, :2]])
    assert (g[:, 2] - z0).min() > 0.2 * pr.physparams.r0         # nobody went through the terrain
    assert g[:, 2].min() < pr.dem.max() + pr.m_deltap            # and the column did come down onto it
