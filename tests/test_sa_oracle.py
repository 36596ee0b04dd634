"""SA_BOUNDARY data path and boundary-conditions engine, CPU side: the oracle against answers that do not depend on
its own restatement (brute force, geometry, hydrostatics)."""
import numpy as np
import pytest

from gpusph_amd import defs as D
from gpusph_amd.problem import SABox, info_id, info_type
from sa_helpers import sa_oracle_state, list_sections, analytic_vertex_gamma


@pytest.fixture(scope="module")
def st():
    return sa_oracle_state(deltap=0.05)


def test_list_geometry_follows_the_reference_rule():
    # ProblemCore::check_neiblistsize: boundary section marker from the un-expanded size, 1.5x list for the vertex section
    from gpusph_amd.params import SimParams, PhysParams, check_neiblistsize
    sp, pp = SimParams(), PhysParams()
    sp.boundarytype = D.SA_BOUNDARY
    sp.set_smoothing(1.3, 0.05)
    pp.r0 = 0.05
    assert check_neiblistsize(sp, pp, 0.05) == (192, 127)
    p = SABox(0.05)       # the mirror resizes it: resize_neiblist(256, 64)
    assert (p.simparams.neiblistsize, p.simparams.neibboundpos) == (320, 255)


def test_three_sections_equal_brute_force(st):
    p = st["problem"]
    g = p.global_pos(st["pos"], st["hash"])
    t = info_type(st["info"])
    r2 = st["sqinfl"]; rb2 = st["bound_sqinfl"]
    rng = np.random.default_rng(5)
    picks = np.concatenate([rng.choice(np.where(t == k)[0], 25, replace=False) for k in (D.PT_FLUID, D.PT_BOUNDARY, D.PT_VERTEX)])
    cells = p.grid_pos_from_hash(st["hash"])
    for i in picks:
        d2 = ((g - g[i]) ** 2).sum(axis=1)
        d2[i] = np.inf
        d2[np.abs(cells - cells[i]).max(axis=1) > 1] = np.inf    # the search covers the 27 cells around the particle's own
        fl, bd, vx = list_sections(st, int(i))
        assert sorted(fl) == sorted(np.where((t == D.PT_FLUID) & (d2 < r2 * (1 - 1e-6)))[0].tolist()) or \
            set(np.where((t == D.PT_FLUID) & (d2 < r2 * (1 - 1e-5)))[0]) <= set(fl) <= set(np.where((t == D.PT_FLUID) & (d2 < r2 * (1 + 1e-5)))[0])
        # boundary neighbours are kept out to the wider radius (isCloseEnough<SA_BOUNDARY>)
        assert set(np.where((t == D.PT_BOUNDARY) & (d2 < rb2 * (1 - 1e-5)))[0]) <= set(bd) <= set(np.where((t == D.PT_BOUNDARY) & (d2 < rb2 * (1 + 1e-5)))[0])
        assert set(np.where((t == D.PT_VERTEX) & (d2 < r2 * (1 - 1e-5)))[0]) <= set(vx) <= set(np.where((t == D.PT_VERTEX) & (d2 < r2 * (1 + 1e-5)))[0])
    assert rb2 > r2
    assert st["neibs_info"].maxVertexNeibs > 0 and st["neibs_info"].hasTooManyNeibs == -1


def test_segments_list_their_own_vertices_and_vertpos_is_the_in_plane_offset(st):
    p = st["problem"]
    g = p.global_pos(st["pos"], st["hash"])
    t = info_type(st["info"])
    ids = info_id(st["info"])
    where = {int(v): k for k, v in enumerate(ids)}
    segs = np.where(t == D.PT_BOUNDARY)[0]
    for i in segs[::7]:
        vx = list_sections(st, int(i))[2]
        own = [where[int(v)] for v in st["vertices"][i, :3]]
        assert set(own) <= set(vx)
        nrm = st["boundelements"][i, :3].astype(np.float64)
        for k, j in enumerate(own):
            rel = g[i] - g[j]                       # relPos = segment - vertex
            vp = st["vertpos"][k][i].astype(np.float64)
            # (coord1, coord2, normal) is an orthonormal frame: the projection keeps the in-plane length
            assert abs(np.hypot(*vp) - np.linalg.norm(rel - nrm * (rel @ nrm))) < 1e-6
        # the three offsets of a triangle about its centroid sum to zero
        assert np.abs(sum(st["vertpos"][k][i].astype(np.float64) for k in range(3))).max() < 1e-6
    fluid = np.where(t == D.PT_FLUID)[0]
    assert not any(st["vertpos"][k][fluid].any() for k in range(3))


def test_vertex_normals_average_the_adjacent_segments(st):
    o, p = st["oracle"], st["problem"]
    be = o.sa_compute_vertex_normal(st["boundelements"], st["vertices"], st["info"], st["hash"], st["cs"], st["nl"], st["n"])
    g = p.global_pos(st["pos"], st["hash"])
    t = info_type(st["info"])
    dp = p.m_deltap
    vx = np.where(t == D.PT_VERTEX)[0]
    assert np.allclose(np.linalg.norm(be[vx, :3], axis=1), 1.0, atol=1e-6) and np.isnan(be[vx, 3]).all()
    # a vertex in the middle of the floor points up; one on the floor/x=0 edge points along (1,0,1)/sqrt2
    mid = [i for i in vx if abs(g[i, 2]) < 1e-5 and 2 * dp < g[i, 0] < p.l - 2 * dp and 2 * dp < g[i, 1] < p.w - 2 * dp]
    assert mid and np.allclose(be[mid, :3], [0, 0, 1], atol=1e-6)
    edge = [i for i in vx if abs(g[i, 2]) < 1e-5 and abs(g[i, 0]) < 1e-5 and 2 * dp < g[i, 1] < p.w - 2 * dp]
    assert edge and np.allclose(be[edge, :3], np.array([1, 0, 1]) / np.sqrt(2), atol=1e-6)
    seg = np.where(t == D.PT_BOUNDARY)[0]
    assert np.array_equal(be[seg], st["boundelements"][seg])


def test_wall_density_of_a_hydrostatic_tank(st):
    """Every fluid neighbour j contributes P_j + rho g (z_j - z_wall) = rho g (depth of the wall point): the Shepard mean the
    boundary conditions impose must be the hydrostatic density AT the segment / vertex, whatever the kernel sum looks like."""
    o, p = st["oracle"], st["problem"]
    t = info_type(st["info"])
    gg = st["gradgamma"].copy()
    vx = np.where(t == D.PT_VERTEX)[0]; seg = np.where(t == D.PT_BOUNDARY)[0]; fl = np.where(t == D.PT_FLUID)[0]
    gg[vx] = 0; gg[vx, 3] = analytic_vertex_gamma(p, st)[vx]
    vel, gg1 = o.sa_segment_bc(st["pos"], st["vel"], gg, st["vertices"], st["boundelements"], st["info"], st["hash"], st["cs"],
                               st["nl"], st["n"], step=0)
    g = p.global_pos(st["pos"], st["hash"])
    expect = p.initial_density(g)
    wet = seg[g[seg, 2] < p.water_level - 1.5 * p.m_deltap]
    assert len(wet) > 300
    assert np.abs(vel[wet, 3] - expect[wet]).max() < 0.02 * expect.max()
    dry = seg[g[seg, 2] > p.water_level + p.simparams.influenceRadius]
    assert len(dry) and np.all(vel[dry, 3] == 0)                       # no fluid in reach: P = 0 -> rho~ = 0
    assert np.all(vel[seg, :3] == 0)
    assert np.array_equal(vel[fl], st["vel"][fl]) and np.array_equal(vel[vx], st["vel"][vx])
    # gamma of a segment = mean of its three vertices (step 0): 1/2 in the middle of a wall
    own = {int(v): k for k, v in enumerate(info_id(st["info"]))}
    for i in seg[::11]:
        m = np.mean([gg[own[int(v)], 3] for v in st["vertices"][i, :3]], dtype=np.float64)
        assert abs(gg1[i, 3] - m) < 1e-6
    assert np.array_equal(gg1[vx], gg[vx])
    # later steps with a finite gamma leave gGam alone
    vel2, gg2 = o.sa_segment_bc(st["pos"], vel, gg1, st["vertices"], st["boundelements"], st["info"], st["hash"], st["cs"],
                                st["nl"], st["n"], step=1)
    assert np.array_equal(gg2[seg], gg1[seg]) and np.array_equal(vel2[seg], vel[seg])
    # vertices
    velv = o.sa_vertex_bc(st["pos"], vel, gg1, st["info"], st["hash"], st["cs"], st["nl"], st["n"])
    # (not the vertical edges of the tank, where a vertex sees a handful of fluid particles only)
    on_edge = ((np.abs(g[:, 0]) < 1e-5) | (np.abs(g[:, 0] - p.l) < 1e-5)) & ((np.abs(g[:, 1]) < 1e-5) | (np.abs(g[:, 1] - p.w) < 1e-5))
    wetv = vx[(g[vx, 2] < p.water_level - 1.5 * p.m_deltap) & ~on_edge[vx]]
    assert len(wetv) > 200
    assert np.abs(velv[wetv, 3] - expect[wetv]).max() < 0.02 * expect.max()
    assert np.array_equal(velv[:, :3], vel[:, :3]) and np.array_equal(velv[seg], vel[seg]) and np.array_equal(velv[fl], vel[fl])


def _plane_integrals(h, d):
    """for a particle at distance d from an infinite planar wall, Wendland kernel of smoothing length h (radius 2h):
    |grad gamma| = integral of W over the plane, 1 - gamma = kernel volume beyond the plane (numerical, float64)"""
    from scipy import integrate
    W = lambda r: 21.0 / (16.0 * np.pi * h ** 3) * (1 - r / (2 * h)) ** 4 * (1 + 2 * r / h) if r < 2 * h else 0.0
    gg = integrate.quad(lambda s: 2 * np.pi * s * W(np.hypot(d, s)), 0, np.sqrt(max(4 * h * h - d * d, 0)))[0]
    cap = integrate.quad(lambda z: integrate.quad(lambda s: 2 * np.pi * s * W(np.hypot(z, s)), 0, np.sqrt(max(4 * h * h - z * z, 0)))[0], d, 2 * h)[0]
    return gg, cap


def test_initial_gamma_of_a_planar_wall(st):
    o, p = st["oracle"], st["problem"]
    t = info_type(st["info"])
    be = o.sa_compute_vertex_normal(st["boundelements"], st["vertices"], st["info"], st["hash"], st["cs"], st["nl"], st["n"])
    gg = o.sa_init_gamma(st["gradgamma"], st["pos"], be, st["vertpos"], st["info"], st["hash"], st["cs"], st["nl"], st["n"], p.m_deltap)
    g = p.global_pos(st["pos"], st["hash"])
    dp, h, R = p.m_deltap, p.simparams.slength, p.simparams.influenceRadius
    seg = np.where(t == D.PT_BOUNDARY)[0]
    assert np.isnan(gg[seg]).all()                               # segments get theirs from the vertices (segment BC, step 0)
    fl = np.where(t == D.PT_FLUID)[0]; vx = np.where(t == D.PT_VERTEX)[0]
    assert np.isfinite(gg[fl]).all() and np.isfinite(gg[vx]).all()
    # fluid over the middle of the floor: only the floor is in reach
    mid = [i for i in fl if R + dp < g[i, 0] < p.l - R - dp and R + dp < g[i, 1] < p.w - R - dp]
    assert len(mid) > 20
    for i in mid:
        d = g[i, 2]
        want_gg, cap = _plane_integrals(h, d)
        if d >= R:
            assert gg[i, 3] == 1.0 and not gg[i, :3].any()
            continue
        # grad gamma is analytical in the reference (gamma.cuh:248-370): exact up to the mesh being finite
        assert abs(gg[i, 2] - want_gg) < 2e-3 * max(want_gg, 1.0) and abs(gg[i, 0]) < 1e-3 * want_gg + 1e-4 and abs(gg[i, 1]) < 1e-3 * want_gg + 1e-4
        # gamma itself is a 5th-order Gauss rule over elements of size dp whose centroid point is taken twice (see
        # gauss_quadrature_O5 in the oracle) applied to an integrand with a 1/q^3 pole at the wall: the missing volume comes
        # out at 1.23x the true cap one and a half dp from the wall and at 1.36x half a dp from it (pinned to the reference
        # bit for bit in test_oracle_pinned.py; here only that it is the cap, to that accuracy)
        miss = 1.0 - gg[i, 3]
        assert 0.9 * cap - 1e-4 <= miss <= 1.45 * cap + 1e-4, (d, cap, miss)
    # vertex particles: the solid angle of the adjacent elements as seen along -grad gamma
    def pick(cond):
        return [i for i in vx if cond(g[i])]
    inner = lambda v, L: 2 * dp < v < L - 2 * dp
    face = pick(lambda q: abs(q[2]) < 1e-5 and inner(q[0], p.l) and inner(q[1], p.w))
    edge = pick(lambda q: abs(q[2]) < 1e-5 and abs(q[0]) < 1e-5 and inner(q[1], p.w))
    corner = pick(lambda q: abs(q[2]) < 1e-5 and abs(q[0]) < 1e-5 and abs(q[1]) < 1e-5)
    assert len(face) > 20 and len(edge) > 3 and len(corner) == 1
    assert np.abs(gg[face, 3] - 0.5).max() < 2e-3
    assert np.abs(gg[edge, 3] - 0.25).max() < 0.03
    assert abs(gg[corner[0], 3] - 0.125) < 2e-3          # three walls meet: an eighth of the kernel support is fluid
    assert np.allclose(gg[face, :3] / np.linalg.norm(gg[face, :3], axis=1, keepdims=True), [0, 0, 1], atol=5e-3)
    # every wall: grad gamma of the fluid next to it points into the fluid (this is what the anticlockwise vertex order
    # of the elements is for: the analytical formula changes sign with the orientation of the edges)
    zmid = (g[:, 2] > R + dp) & (g[:, 2] < p.water_level - dp)
    for axis, L in ((0, p.l), (1, p.w)):
        other = 1 - axis
        Lo = p.w if axis == 0 else p.l
        sel = np.array([i for i in fl if zmid[i] and R + dp < g[i, other] < Lo - R - dp])
        near_lo = sel[np.abs(g[sel, axis] - dp) < 1e-5]; near_hi = sel[np.abs(g[sel, axis] - (L - dp)) < 1e-5]
        assert len(near_lo) > 3 and len(near_hi) > 3
        assert (gg[near_lo, axis] > 1.0).all() and (gg[near_hi, axis] < -1.0).all()
        assert np.abs(gg[near_lo, other]).max() < 0.2 and np.abs(gg[near_lo, 2]).max() < 0.2


def test_sa_forces_keep_a_hydrostatic_tank_at_rest():
    """SA forces (fluid, vertex and boundary-element terms, divided by gamma) + gravity on a hydrostatic tank: the residual
    acceleration is a small fraction of g for the bulk of the fluid, and twelve predictor-corrector steps of the whole SA
    sequence (forces, Euler, gamma by quadrature, boundary conditions) leave the water where it is."""
    from sa_helpers import OracleSaSim
    prob = SABox(0.05, options="StillWaterRepackSA")
    sim = OracleSaSim(prob)
    n, o = sim.n, sim.o
    t = info_type(sim.info)
    fl = np.where(t == D.PT_FLUID)[0]
    f, cfl, nb = o.forces_sa(sim.pos, sim.vel, sim.info, sim.hash, sim.cs, sim.nl, sim.gg, sim.be, sim.vertpos, n, prob.m_deltap)
    assert np.isfinite(f[fl]).all() and not f[t != D.PT_FLUID].any()
    g = prob.global_pos(sim.pos, sim.hash)
    acc = np.linalg.norm(f[fl, :3], axis=1)
    # without the wall terms the first layers would fall with g; with them the residual is the discretisation error of a
    # hydrostatic state on a lattice (free-surface layer excluded: SPH has no boundary condition there)
    below = fl[g[fl, 2] < prob.water_level - 1.5 * prob.m_deltap]
    res = np.linalg.norm(f[below, :3], axis=1)
    assert np.median(res) < 0.2 * 9.81 and res.max() < 0.6 * 9.81, (np.median(res), res.max())
    assert acc.max() < 1.5 * 9.81
    # the wall terms matter: the layer next to the floor would otherwise be pushed up by the fluid below... there is none:
    # it would fall with more than g/2
    first = fl[np.abs(g[fl, 2] - prob.m_deltap) < 1e-5]
    assert len(first) > 50 and np.abs(f[first, 2]).max() < 0.6 * 9.81
    z0 = g[fl, 2].copy()
    for _ in range(12):
        sim.step()
    assert np.isfinite(sim.pos).all() and np.isfinite(sim.vel).all() and np.isfinite(sim.gg[t != D.PT_BOUNDARY]).all()
    g1 = prob.global_pos(sim.pos, sim.hash)
    c0 = float(np.float32(prob.physparams.sscoeff[0]))
    assert np.abs(sim.vel[fl, :3]).max() < 0.02 * c0
    assert np.abs(g1[fl, 2] - z0).max() < 0.05 * prob.m_deltap
    assert 0.1 <= sim.gg[fl, 3].min() and sim.gg[fl, 3].max() <= 1.0 + 1e-6
    assert sim.t > 0 and sim.dt > 0


def test_density_summation_form_keeps_the_tank_at_rest_and_follows_a_compression():
    """StillWaterSA's own option set: density summation, dynamic gamma, Brezzi diffusion.  (1) Hydrostatic tank: twelve steps
    of the whole sequence leave it at rest, gamma stays what the initialisation found.  (2) Known answer of the summation: a
    uniform compression x -> (1 - e) x of all positions raises the density of interior particles by 3 e."""
    from sa_helpers import OracleSaSim
    prob = SABox(0.05, l=0.8, w=0.7, options="StillWaterSA")
    sim = OracleSaSim(prob)
    n, o = sim.n, sim.o
    t = info_type(sim.info)
    fl = np.where(t == D.PT_FLUID)[0]
    g0 = prob.global_pos(sim.pos, sim.hash)
    gam0 = sim.gg[fl, 3].copy()
    for _ in range(12):
        sim.step()
    assert np.isfinite(sim.pos).all() and np.isfinite(sim.vel).all()
    g1 = prob.global_pos(sim.pos, sim.hash)
    c0 = float(np.float32(prob.physparams.sscoeff[0]))
    assert np.abs(sim.vel[fl, :3]).max() < 0.02 * c0 and np.abs(g1[fl] - g0[fl]).max() < 0.05 * prob.m_deltap
    assert np.abs(sim.gg[fl, 3] - gam0).max() < 5e-3                 # dynamic gamma: integrated, not recomputed
    # (the lattice is not the discrete equilibrium: the column settles with an acoustic oscillation of a few tenths of rho~)
    assert np.abs(sim.vel[fl, 3] - prob.initial_density(g0)[fl]).max() < 0.4 * np.abs(prob.initial_density(g0)).max()
    # (2) density summation alone, on a fresh state of a deeper tank
    prob = SABox(0.05, l=0.8, w=0.7, h=0.6, H=0.55, options="StillWaterSA")
    sim = OracleSaSim(prob)
    n, o = sim.n, sim.o
    t = info_type(sim.info)
    fl = np.where(t == D.PT_FLUID)[0]
    e = 1e-3
    R, dp = prob.simparams.influenceRadius, prob.m_deltap
    g = prob.global_pos(sim.pos, sim.hash)
    centre = np.array([prob.l / 2, prob.w / 2, prob.water_level / 2])
    newpos = sim.pos.copy()
    newpos[:, :3] = sim.pos[:, :3] - (e * (g - centre)).astype(np.float32)         # everything moves, walls too: no gamma change
    v, gg = o.sa_density_sum(sim.vel, sim.pos, newpos, sim.vel, sim.gg, sim.be, sim.vertpos, sim.info, sim.hash, sim.cs, sim.nl, n)
    inner = [i for i in fl if R + dp < g[i, 0] < prob.l - R - dp and R + dp < g[i, 1] < prob.w - R - dp and R + dp < g[i, 2] < prob.water_level - R - dp]
    assert len(inner) >= 4, len(inner)
    drho = (v[inner, 3] - sim.vel[inner, 3]) / (1.0 + sim.vel[inner, 3])
    assert np.abs(drho / (3 * e) - 1).max() < 0.05
    assert np.array_equal(v[t != D.PT_FLUID], sim.vel[t != D.PT_FLUID]) and np.array_equal(gg[t != D.PT_FLUID].view(np.uint32), sim.gg[t != D.PT_FLUID].view(np.uint32))


def test_sa_repacking_force_vanishes_in_a_filled_lattice_and_pushes_a_displaced_particle_back():
    """run_repack with SA_BOUNDARY: -a c0^2 grad Gamma with the wall term of the boundary elements and the division by gamma.  On
    the regular lattice the discrete gradient of the kernel-sum is small next to a c0^2/h everywhere (walls included: the boundary
    integral replaces the missing neighbours); a particle displaced inside the bulk is pushed back towards its site."""
    from sa_helpers import OracleSaSim
    pr = SABox(0.04, l=0.8, w=0.8, h=0.8, H=0.64, options="StillWaterRepackSA")
    sim = OracleSaSim(pr, repack=True)
    n = sim.n
    f, cfl, nb = sim.o.repack_forces_sa(sim.pos, sim.vel, sim.info, sim.hash, sim.cs, sim.nl, sim.gg, sim.be, sim.vertpos, n, pr.m_deltap)
    t = info_type(sim.info[:n])
    fluid = t == D.PT_FLUID
    a, c0, h = pr.simparams.repack_a, pr.physparams.sscoeff[0], pr.simparams.slength
    scale = a * c0 * c0 / h
    g = pr.global_pos(sim.pos[:n], sim.hash[:n])
    R = float(sim.o.p.influenceradius)
    surface = g[:, 2] > g[fluid, 2].max() - 1.01 * R
    interior = np.all((g[:, :2] > g[fluid, :2].min(0) + R) & (g[:, :2] < g[fluid, :2].max(0) - R), axis=1) & (g[:, 2] > g[fluid, 2].min() + R)
    calm = fluid & ~surface & interior
    assert calm.sum() > 20 and np.abs(f[:n, :3][calm]).max() < 0.02 * scale
    # next to the walls the boundary integral stands in for the missing neighbours; the vertex particles enter the reference's
    # sum with a c0 instead of a c0^2 (forces_kernel.def:3057-3072, reproduced), so the balance there is only partial
    walls = fluid & ~surface & ~interior
    assert np.abs(f[:n, :3][walls]).max() < 0.5 * scale
    assert np.abs(f[:n, :3][fluid & surface]).max() > 0.2 * scale  # the free surface is what the lid of a repacking run is for
    assert not f[:n][~fluid].any()
    # displace one bulk particle: the force points back
    bulk = np.where(calm & (np.abs(g[:, 0] - g[fluid, 0].mean()) < 0.06) & (np.abs(g[:, 1] - g[fluid, 1].mean()) < 0.06) &
                    (np.abs(g[:, 2] - 0.3) < 0.04))[0]
    i = int(bulk[0])
    pos = sim.pos.copy(); pos[i, 0] += np.float32(0.3 * pr.m_deltap)
    f2 = sim.o.repack_forces_sa(pos, sim.vel, sim.info, sim.hash, sim.cs, sim.nl, sim.gg, sim.be, sim.vertpos, n, pr.m_deltap)[0]
    assert f2[i, 0] < -0.1 * scale
    # a jittered tank relaxes: the largest mixing force shrinks over a few iterations
    prj = SABox(0.04, l=0.8, w=0.8, h=0.8, H=0.64, options="StillWaterRepackSA", jitter=0.2)
    s2 = OracleSaSim(prj, repack=True)
    first = None
    for _ in range(8):
        s2.repack_step()
        m = np.abs(s2.forces[:n, :3][calm]).mean()
        first = m if first is None else first
    assert np.isfinite(s2.pos[:n]).all() and m < first


def test_floor_that_feels_the_fluid_carries_the_weight_of_the_water():
    """compute_boundary_pressure_force (src/cuda/forces_kernel.def:3258-3266,4115-4145), the forces engine's part of
    CompleteSaExample.cu's option set (a body with FG_COMPUTE_FORCE under SA_BOUNDARY): every COMPUTE_FORCE boundary element takes
    F = -P A n.  Known answer: the floor of a hydrostatic tank carries the weight of the water above it, rho g (l w level); the
    torque about the floor's centre vanishes by symmetry; the rows lie where id + rbstart[object] says; vertices and the segments of
    the other walls write nothing."""
    from gpusph_amd.problem import SALoadBox
    from sa_helpers import OracleSaSim
    p = SALoadBox(0.05)
    sim = OracleSaSim(p)
    o, n = sim.o, sim.n
    f, rbf, rbt = o.sa_body_pressure_forces(sim.pos, sim.vel, sim.info, sim.hash, sim.be, n, p.num_obstacle)
    tot, tq = rbf[:, :3].astype(np.float64).sum(0), rbt[:, :3].astype(np.float64).sum(0)
    weight = p.physparams.rho0[0] * 9.81 * p.l * p.w * p.water_level
    assert abs(tot[2] + weight) < 0.01 * weight and abs(tot[0]) < 1e-6 * weight and abs(tot[1]) < 1e-6 * weight
    assert np.abs(tq).max() < 1e-5 * weight * p.l
    load = (sim.info[:n, 0] & D.FG_COMPUTE_FORCE) != 0
    t = info_type(sim.info[:n])
    assert load.sum() == p.num_obstacle and (t[load] == D.PT_BOUNDARY).all()
    # each element's row: id - first id of the body; its own forces row holds the same vector with w = 0
    rows = info_id(sim.info[:n][load]).astype(np.int64) + int(p.rb_firstindex[0])
    assert np.array_equal(np.sort(rows), np.arange(p.num_obstacle))
    assert np.array_equal(rbf[rows], f[:n][load]) and (f[:n][load, 3] == 0).all() and (f[:n][load, 2] < 0).all()
    assert not f[:n][~load].any()
    # the force follows the pressure: after some steps of a tank at rest it is still the weight (the run does not drift)
    for _ in range(4):
        sim.step()
    _, rbf2, _ = o.sa_body_pressure_forces(sim.pos, sim.vel, sim.info, sim.hash, sim.be, n, p.num_obstacle)
    assert abs(rbf2[:, 2].astype(np.float64).sum() + weight) < 0.01 * weight
