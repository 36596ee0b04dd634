"""Open boundaries of SA_BOUNDARY (the half that is not built): known answers for the GROUNDWORK in the oracle --
the Riemann-invariant boundary condition, the mass repartition of a segment, corner identification, the initial masses of the
open-boundary vertices, the removal of outgoing particles.  There is no product counterpart yet: these tests pin the checker a
later round builds the engines against (oracle/sph_oracle.c "Open boundaries"; src/cuda/boundary_conditions_kernel.cu:111-283,
1999-2172, 2319-2398 of the reference).  Parity unpinned: the reference holds no fixture for any of it."""
import numpy as np
import pytest

from gpusph_amd import defs as D
from gpusph_amd.problem import SABox, info_id, info_type
from sa_helpers import sa_oracle_state


@pytest.fixture(scope="module")
def st():
    return sa_oracle_state(deltap=0.05)


def _eos(st):
    pp = st["problem"].physparams
    return float(pp.gammacoeff[0]), float(pp.sscoeff[0]), float(pp.bcoeff[0]), float(pp.rho0[0])


def test_riemann_celerity_and_its_inverse(st):
    o = st["oracle"]
    gam, c0, _, _ = _eos(st)
    for rt in (-0.01, 0.0, 0.003, 0.02, 0.05):
        assert abs(o.riemann_RHOR(o.riemann_R(rt)) - rt) < 2e-6
        # R = 2 c / (gamma - 1)
        assert abs(o.riemann_R(rt) * (gam - 1.0) / 2.0 - o.sound_speed(rt)) < 2e-6 * c0
        assert abs(o.eos_RHO(o.eos_P(rt)) - rt) < 2e-6


def test_velocity_inlet_takes_its_density_from_the_outgoing_invariant(st):
    o = st["oracle"]
    gam, c0, B, rho0 = _eos(st)
    n = np.array([1.0, 0.0, 0.0])
    rho_int = 0.01
    # imposed normal velocity equal to the fluid's: nothing to adjust
    ev = o.io_boundary_condition([0.3, 0, 0, 0], True, rho_int, 0.0, [0.3, 0, 0], 0.3, 0.3, n)
    assert abs(ev[3] - rho_int) < 2e-6 and np.allclose(ev[:3], [0.3, 0, 0])
    # expansion (the boundary recedes from the fluid: unExt < unInt): c_b = c_int + (gamma - 1)/2 (unExt - unInt)
    for un_ext in (0.2, 0.0, -0.5):
        ev = o.io_boundary_condition([un_ext, 0, 0, 0], True, rho_int, 0.0, [0.3, 0, 0], 0.3, un_ext, n)
        c_int = c0 * (1.0 + rho_int) ** ((gam - 1.0) / 2.0)
        c_b = c_int + 0.5 * (gam - 1.0) * (un_ext - 0.3)
        want = (c_b / c0) ** (2.0 / (gam - 1.0)) - 1.0
        assert abs(ev[3] - want) < 5e-6 and ev[3] < rho_int
    # compression (unExt > unInt): the density rises; the momentum jump P_b = P_int + rho u_int (u_int - u_ext) only when the
    # wave it implies outruns the fluid's own characteristic, the fluid's state otherwise (a contact discontinuity)
    for un_int, un_ext in ((0.3, 0.6), (-0.4, -0.1)):
        ev = o.io_boundary_condition([un_ext, 0, 0, 0], True, rho_int, 0.0, [un_int, 0, 0], un_int, un_ext, n)
        P_int = B * ((1.0 + rho_int) ** gam - 1.0)
        P_b = P_int + rho0 * (1.0 + rho_int) * un_int * (un_int - un_ext)
        rho_b = (P_b / B + 1.0) ** (1.0 / gam) - 1.0
        c = lambda r: c0 * (1.0 + r) ** ((gam - 1.0) / 2.0)
        want = rho_b if un_ext + c(rho_b) > un_int + c(rho_int) else rho_int
        assert abs(ev[3] - want) < 5e-6


def test_pressure_outlet_takes_its_normal_velocity_from_the_invariant(st):
    o = st["oracle"]
    gam, c0, _, _ = _eos(st)
    n = np.array([0.0, 0.0, 1.0])
    u_int = np.array([0.2, -0.1, -0.3])          # leaving through a boundary whose normal points into the fluid
    un = float(u_int @ n)
    # the imposed density is the fluid's: the normal velocity is the fluid's, the tangential one is kept (dv/dn = 0 on outflow)
    ev = o.io_boundary_condition([9, 9, 9, 0.01], False, 0.01, 0.01, u_int, un, 0.0, n)
    assert np.allclose(ev[:3], u_int, atol=2e-6) and ev[3] == np.float32(0.01)
    # a lower imposed pressure draws the fluid out faster, by the difference of the Riemann celerities
    ev = o.io_boundary_condition([0, 0, 0, 0.0], False, 0.01, 0.0, u_int, un, 0.0, n)
    want = un + (o.riemann_R(0.0) - o.riemann_R(0.01))
    assert abs(ev[2] - want) < 2e-5 and ev[2] < un and np.allclose(ev[:2], u_int[:2], atol=1e-6) and ev[3] == 0.0
    # inflow (flux > 0): no tangential velocity is imposed
    u_in = np.array([0.2, -0.1, 0.3])
    ev = o.io_boundary_condition([0, 0, 0, 0.01], False, 0.01, 0.01, u_in, 0.3, 0.0, n)
    assert abs(ev[2] - 0.3) < 2e-6 and ev[0] == 0.0 and ev[1] == 0.0
    # a negative imposed pressure never lets fluid in
    ev = o.io_boundary_condition([0, 0, 0, -0.001], False, -0.02, -0.001, [0, 0, 0.05], 0.05, 0.0, n)
    assert ev[2] <= 0.0 and ev[3] == np.float32(-0.001)


def test_mass_repartition_is_barycentric_inside_and_clipped_outside(st):
    o = st["oracle"]
    rng = np.random.default_rng(11)
    tri = np.array([[0.0, 0.0, 0.0], [0.05, 0.0, 0.0], [0.0, 0.05, 0.0]])
    nrm = np.array([0.0, 0.0, 1.0])
    assert np.allclose(o.mass_repartition(tri - tri.mean(axis=0), nrm), 1.0 / 3.0, atol=1e-6)
    for k in range(3):       # at a vertex: all of it
        want = np.zeros(3); want[k] = 1.0
        assert np.allclose(o.mass_repartition(tri - tri[k], nrm), want, atol=1e-6)
    for _ in range(200):
        lam = rng.dirichlet(np.ones(3))
        x = lam @ tri + np.array([0, 0, rng.uniform(-0.02, 0.02)])       # a point above / below the segment projects onto it
        assert np.allclose(o.mass_repartition(tri - x, nrm), lam, atol=2e-5)
    for _ in range(200):     # outside: the weights stay a partition of the mass over the three vertices
        x = np.array([rng.uniform(-0.1, 0.15), rng.uniform(-0.1, 0.15), 0.0])
        b = o.mass_repartition(tri - x, nrm)
        assert (b >= -1e-6).all() and abs(b.sum() - 1.0) < 1e-5


def _open_wall(st, flags):
    """the x = 0 wall of the tank as open boundary number 1: (info with the flags set, masks of its segments and vertices)"""
    p = st["problem"]
    g = p.global_pos(st["pos"], st["hash"])
    t = info_type(st["info"])
    seg = (t == D.PT_BOUNDARY) & (st["boundelements"][:, 0] > 0.5) & (np.abs(g[:, 0]) < 1e-6)
    vtx = (t == D.PT_VERTEX) & (np.abs(g[:, 0]) < 1e-6)
    info = st["info"].copy()
    info[seg | vtx, 0] |= flags
    info[seg | vtx, 1] = (info[seg | vtx, 1] & 0xF000) | 1
    return info, seg, vtx, g


def test_corner_vertices_are_those_shared_with_another_wall(st):
    p, o = st["problem"], st["oracle"]
    info, seg, vtx, g = _open_wall(st, D.FG_INLET | D.FG_VELOCITY_DRIVEN)
    out = o.sa_identify_corner_vertices(st["pos"], info, st["hash"], st["vertices"], st["cs"], st["nl"], st["n"])
    corner = (out[:, 0] & D.FG_CORNER) != 0
    eps = 1e-6
    on_edge = vtx & ((np.abs(g[:, 1]) < eps) | (np.abs(g[:, 1] - p.w) < eps) | (np.abs(g[:, 2]) < eps))   # y = 0, y = w, floor; open on top
    assert np.array_equal(corner, on_edge) and corner.sum() > 0 and (vtx & ~corner).sum() > 0
    assert np.array_equal(out[:, 0] & ~np.uint16(D.FG_CORNER), info[:, 0])      # nothing else touched


def _mesh_counts(st, info, deltap):
    """vertex counts and new masses from the connectivity alone (no neighbour lists)"""
    ids = info_id(st["info"])
    where = {int(v): k for k, v in enumerate(ids)}
    io = (info[:, 0] & (D.FG_INLET | D.FG_OUTLET)) != 0
    corner = (info[:, 0] & D.FG_CORNER) != 0
    t = info_type(info)
    segs = np.where((t == D.PT_BOUNDARY) & io)[0]
    partners = {}       # vertex index -> list of partner vertex indices, one per shared segment
    for s in segs:
        vs = [where[int(v)] for v in st["vertices"][s, :3]]
        for a in vs:
            partners.setdefault(a, []).extend(b for b in vs if b != a)
    count = np.zeros(len(info), dtype=np.float64)
    for a, lst in partners.items():
        if not corner[a]:
            count[a] = sum(1 for b in lst if not corner[b])
    ref = 0.5 * deltap ** 3 * st["problem"].physparams.rho0[0]
    mass = st["pos"][:, 3].astype(np.float64).copy()
    new = mass.copy()
    for a, lst in partners.items():
        if corner[a]:
            continue
        get_a = int(ids[a]) % 2
        for b in lst:
            if corner[b] or int(ids[b]) % 2 == get_a:
                continue
            if get_a:
                if ref - mass[a] > 0:
                    new[a] += (ref - mass[a]) / count[a]
            elif ref - mass[b] > 0:
                new[a] -= (ref - mass[b]) / count[b]
    return count, new


def test_initial_masses_of_the_open_boundary_vertices(st):
    p, o = st["problem"], st["oracle"]
    dp = p.m_deltap
    info, seg, vtx, g = _open_wall(st, D.FG_INLET | D.FG_VELOCITY_DRIVEN)
    info = o.sa_identify_corner_vertices(st["pos"], info, st["hash"], st["vertices"], st["cs"], st["nl"], st["n"])
    # lighter vertices than half a particle, so that there is something to hand over
    pos = st["pos"].copy()
    inner = vtx & ((info[:, 0] & D.FG_CORNER) == 0)
    rng = np.random.default_rng(3)
    pos[inner, 3] *= rng.uniform(0.5, 0.9, size=int(inner.sum())).astype(np.float32)
    st2 = dict(st, pos=pos)
    count, new_pos = o.sa_init_io_mass(pos, info, st["hash"], st["vertices"], st["cs"], st["nl"], st["n"], dp)
    want_count, want_mass = _mesh_counts(st2, info, dp)
    assert np.array_equal(count, want_count.astype(np.float32)) and count[inner].min() >= 2
    assert np.allclose(new_pos[:, 3], want_mass, rtol=2e-6, atol=0)
    assert np.array_equal(new_pos[:, :3], pos[:, :3])
    untouched = ~inner
    assert np.array_equal(new_pos[untouched, 3], pos[untouched, 3])
    # what the odd vertices take, the even ones give: the wall's mass is conserved
    assert abs(new_pos[inner, 3].astype(np.float64).sum() - pos[inner, 3].astype(np.float64).sum()) < 1e-6 * pos[inner, 3].sum()
    odd = inner & (info_id(st["info"]) % 2 == 1)
    assert (new_pos[odd, 3] >= pos[odd, 3]).all() and (new_pos[inner & ~odd, 3] <= pos[inner & ~odd, 3]).all()
    assert (new_pos[odd, 3] > pos[odd, 3]).any()


def test_outgoing_particles_are_disabled_and_their_marks_cleared(st):
    o = st["oracle"]
    t = info_type(st["info"])
    fl = np.where(t == D.PT_FLUID)[0]
    vertices = st["vertices"].copy()
    vertices[fl[3]] = (17, 0, 0, 5)            # marked by findOutgoingSegment: the segment's vertices and the weights' tag
    vertices[fl[8]] = (0, 23, 41, 0)
    pos, v2 = o.disable_outgoing_parts(st["pos"], vertices, st["info"], st["n"])
    gone = np.zeros(len(pos), dtype=bool); gone[[fl[3], fl[8]]] = True
    assert np.isnan(pos[gone, 3]).all() and np.array_equal(pos[~gone], st["pos"][~gone])
    assert (v2[gone] == 0).all() and np.array_equal(v2[~gone], st["vertices"][~gone])


def _with_gamma(st):
    """gradgamma of the walls as the initialisation leaves it: gamma 1/2 on the vertices of a face (any finite value does here)"""
    g = st["gradgamma"].copy()
    t = info_type(st["info"])
    g[t == D.PT_VERTEX] = (0.0, 0.0, 0.0, 0.5)
    g[t == D.PT_FLUID] = (0.0, 0.0, 0.0, 1.0)
    return g


def test_segment_conditions_with_open_boundaries_enabled_leave_solid_walls_as_they_are(st):
    o = st["oracle"]
    gg = _with_gamma(st)
    ev0 = np.full_like(st["vel"], 7.0)       # stale Eulerian velocities: solid segments get theirs cleared
    v_io, g_io, ev = o.sa_segment_bc_io(st["pos"], st["vel"], gg, ev0, st["vertices"], st["boundelements"], st["info"],
                                        st["hash"], st["cs"], st["nl"], st["n"], 0)
    v, g = o.sa_segment_bc(st["pos"], st["vel"], gg, st["vertices"], st["boundelements"], st["info"], st["hash"], st["cs"],
                           st["nl"], st["n"], 0)
    assert np.array_equal(v_io, v) and np.array_equal(g_io, g, equal_nan=True)
    seg = info_type(st["info"]) == D.PT_BOUNDARY
    assert (ev[seg] == 0).all() and (ev[~seg] == 7.0).all()


def test_velocity_inlet_segments_take_the_fluids_pressure(st):
    p, o = st["problem"], st["oracle"]
    info, seg, vtx, g = _open_wall(st, D.FG_INLET | D.FG_VELOCITY_DRIVEN)
    U = 0.2
    vel = st["vel"].copy()
    fl = info_type(info) == D.PT_FLUID
    vel[fl, 0] = U
    ev0 = np.zeros_like(vel)
    ev0[seg | vtx, 0] = U                     # IMPOSE_OPEN_BOUNDARY_CONDITION of a velocity inlet
    v, _, ev = o.sa_segment_bc_io(st["pos"], vel, _with_gamma(st), ev0, st["vertices"], st["boundelements"], info,
                                  st["hash"], st["cs"], st["nl"], st["n"], 1)
    # the imposed velocity stays; the stream arrives with that same velocity, so the density is the interior one:
    # the Shepard mean of the fluid's pressure at the segment (no gravity correction in this sum), i.e. about hydrostatic
    assert np.allclose(ev[seg, :3], [U, 0, 0], atol=1e-6)
    assert np.array_equal(v[seg, 3], ev[seg, 3])
    # (away from the tank's edges: a segment tucked into a corner has too little fluid in reach for the Shepard mean -- less than
    # a tenth of its gamma -- and is treated like a dry one)
    dp = p.m_deltap
    wet = seg & (g[:, 2] < p.water_level - 2 * dp) & (g[:, 2] > 1.5 * dp) & (g[:, 1] > 1.5 * dp) & (g[:, 1] < p.w - 1.5 * dp)
    dry = seg & (g[:, 2] > p.water_level + 3 * dp)
    assert wet.sum() > 10 and dry.sum() > 5
    hyd = p.initial_density(g)
    assert np.abs(ev[wet, 3] - hyd[wet]).max() < 0.25 * hyd[wet].max() and (ev[wet, 3] > 0).all()
    depth_order = np.argsort(g[wet, 2])
    assert ev[wet, 3][depth_order][:5].mean() > ev[wet, 3][depth_order][-5:].mean()      # denser at the bottom
    assert (ev[dry, 3] == 0).all()             # no fluid in reach: the reference density
    # the other walls are solid walls
    other = (info_type(info) == D.PT_BOUNDARY) & ~seg
    v_solid, _ = o.sa_segment_bc(st["pos"], vel, _with_gamma(st), st["vertices"], st["boundelements"], st["info"], st["hash"],
                                 st["cs"], st["nl"], st["n"], 1)
    assert np.array_equal(v[other], v_solid[other]) and (ev[other] == 0).all()


def test_pressure_outlet_segments_draw_by_the_difference_of_the_celerities(st):
    p, o = st["problem"], st["oracle"]
    info, seg, vtx, g = _open_wall(st, D.FG_OUTLET)
    hyd = p.initial_density(g)
    ev0 = np.zeros_like(st["vel"])
    ev0[seg | vtx, 3] = hyd[seg | vtx]        # imposed: the hydrostatic pressure
    v, _, ev = o.sa_segment_bc_io(st["pos"], st["vel"], _with_gamma(st), ev0, st["vertices"], st["boundelements"], info,
                                  st["hash"], st["cs"], st["nl"], st["n"], 1)
    assert np.array_equal(ev[seg, 3], ev0[seg, 3]) and np.array_equal(v[seg, 3], ev0[seg, 3])
    dp = p.m_deltap
    wet = np.where(seg & (g[:, 2] < p.water_level - 2 * dp) & (g[:, 2] > 1.5 * dp) & (g[:, 1] > 1.5 * dp) & (g[:, 1] < p.w - 1.5 * dp))[0]
    # fluid at rest: the normal velocity is R(imposed) - R(interior), the tangential one zero; the interior density is the
    # Shepard mean the velocity inlet reads (previous test), so the two passes can be held against each other
    info_v, _, _, _ = _open_wall(st, D.FG_INLET | D.FG_VELOCITY_DRIVEN)
    _, _, ev_v = o.sa_segment_bc_io(st["pos"], st["vel"], _with_gamma(st), np.zeros_like(st["vel"]), st["vertices"],
                                    st["boundelements"], info_v, st["hash"], st["cs"], st["nl"], st["n"], 1)
    for i in wet[::5]:
        rho_int, rho_ext = float(ev_v[i, 3]), float(ev0[i, 3])
        want = o.riemann_R(rho_ext) - o.riemann_R(rho_int)
        if rho_ext <= rho_int:
            assert abs(ev[i, 0] - want) < 1e-4 * max(1.0, abs(want)) + 2e-5 or ev[i, 0] == 0.0
        assert ev[i, 1] == 0.0 and ev[i, 2] == 0.0
        assert abs(ev[i, 0]) < 0.1 * p.physparams.sscoeff[0]


def test_particles_that_cross_an_open_boundary_are_marked_with_the_segment_they_crossed(st):
    p, o = st["problem"], st["oracle"]
    dp = p.m_deltap
    info, seg, vtx, g = _open_wall(st, D.FG_OUTLET)
    fl = np.where(info_type(info) == D.PT_FLUID)[0]
    near = fl[np.abs(g[fl, 0] - dp) < 1e-6]                  # the fluid layer next to the x = 0 wall
    pos, vel = st["pos"].copy(), st["vel"].copy()
    out, back, still = near[5], near[9], near[13]
    for i in (out, back, still):
        pos[i, 0] -= np.float32(1.3 * dp)                    # 0.3 dp behind the wall (the lists are those of the last build)
    pos[out, 1] += np.float32(0.22 * dp); pos[out, 2] += np.float32(0.09 * dp)      # off the lattice: one closest segment
    vel[out, 0] = -0.4                                       # leaving
    vel[back, 0] = +0.4                                      # behind the wall but coming back in
    vel[still, 0] = 0.0                                      # not moving relative to the segment
    inside = near[21]
    vel[inside, 0] = -0.4                                    # moving out, but still in front of the wall
    infl = float(p.simparams.influenceRadius)
    v2, g2 = o.find_outgoing_segment(pos, vel, st["vertices"], _with_gamma(st), st["vertpos"], st["boundelements"], info,
                                     st["hash"], st["cs"], st["nl"], st["n"], infl)
    marked = np.where((v2[:, 0] | v2[:, 1]) != 0)[0]
    assert set(marked) - set(np.where(info_type(info) != D.PT_FLUID)[0]) == {int(out)}
    # the mark: the vertices of the closest open-boundary segment behind which the particle lies, its barycentric shares, its mass
    gp = p.global_pos(pos, st["hash"])
    segs = np.where(seg)[0]
    d = np.linalg.norm(g[segs] - gp[out], axis=1)
    closest = segs[np.argmin(d)]
    assert np.array_equal(v2[out], st["vertices"][closest])
    w = g2[out]
    assert abs(w[:3].sum() - 1.0) < 1e-5 and (w[:3] >= -1e-6).all() and w[3] == st["pos"][out, 3]
    ids = info_id(st["info"])
    where = {int(v): k for k, v in enumerate(ids)}
    corners = np.array([g[where[int(v)]] for v in st["vertices"][closest, :3]])
    proj = gp[out].copy(); proj[0] = 0.0                      # the particle projected onto the wall
    assert np.allclose(w[:3] @ corners, proj, atol=2e-6 + 1e-3 * dp) or not _inside(corners, proj)
    # everybody else keeps grad gamma and has no mark
    others = np.ones(len(pos), dtype=bool); others[out] = False
    assert np.array_equal(g2[others], _with_gamma(st)[others], equal_nan=True)
    # ... and the marked particle is what disableOutgoingParts removes
    pos3, v3 = o.disable_outgoing_parts(pos, v2, info, st["n"])
    assert np.isnan(pos3[out, 3]) and (v3[out] == 0).all() and np.isfinite(pos3[[back, still, inside], 3]).all()


def _inside(tri, x):
    """is x (in the triangle's plane) inside the triangle"""
    a, b, c = tri
    n = np.cross(b - a, c - a)
    s = [np.dot(np.cross(q - p_, x - p_), n) for p_, q in ((a, b), (b, c), (c, a))]
    return min(s) >= 0 or max(s) <= 0


def _inlet_state(st, U, flags):
    """the x = 0 wall as an open boundary in a uniform stream U ex: arrays after corner identification, vertex normals and the
    segment conditions of the last step"""
    p, o = st["problem"], st["oracle"]
    info, seg, vtx, g = _open_wall(st, flags)
    info = o.sa_identify_corner_vertices(st["pos"], info, st["hash"], st["vertices"], st["cs"], st["nl"], st["n"])
    be = o.sa_compute_vertex_normal(st["boundelements"], st["vertices"], info, st["hash"], st["cs"], st["nl"], st["n"])
    vel = st["vel"].copy()
    vel[info_type(info) == D.PT_FLUID, 0] = U
    ev0 = np.zeros_like(vel)
    if flags & D.FG_VELOCITY_DRIVEN:
        ev0[seg | vtx, 0] = U
    else:
        ev0[seg | vtx, 3] = p.initial_density(g)[seg | vtx]
    gg = _with_gamma(st)
    v, gg2, ev = o.sa_segment_bc_io(st["pos"], vel, gg, ev0, st["vertices"], be, info, st["hash"], st["cs"], st["nl"], st["n"], 2)
    return dict(info=info, seg=seg, vtx=vtx, g=g, be=be, vel=v, ggam=gg2, ev=ev)


def test_inlet_vertices_collect_the_mass_flux_of_their_segments_and_release_particles(st):
    p, o = st["problem"], st["oracle"]
    dp = p.m_deltap
    rho0 = float(p.physparams.rho0[0])
    U, dt = 0.2, 2.0e-3
    s = _inlet_state(st, U, D.FG_INLET | D.FG_VELOCITY_DRIVEN)
    info, vtx, g = s["info"], s["vtx"], s["g"]
    inner = vtx & ((info[:, 0] & D.FG_CORNER) == 0)
    nopen = int(vtx.sum())
    next_ids = np.full(st["n"], 0xFFFFFFFF, dtype=np.uint32)
    next_ids[vtx] = st["n"] + np.arange(nopen, dtype=np.uint32)          # as the host hands them out
    ref = rho0 * dp ** 3
    # --- a step that is not the last: the masses integrate the flux, nobody is created
    a = o.sa_vertex_bc_io(st["pos"], s["vel"], s["ggam"], s["ev"], st["vertices"], s["be"], st["vertpos"], info, st["hash"],
                          next_ids, st["cs"], st["nl"], st["n"], dp, dt, 1, nopen)
    assert a["n"] == st["n"]
    gain = a["new_pos"][:st["n"], 3].astype(np.float64) - st["pos"][:, 3]
    assert (gain[~inner] == 0).all()
    # a vertex in the middle of the wall belongs to six triangles of area dp^2/2 and takes a third of each: rho U dp^2 per unit time
    mid = inner & (g[:, 1] > 1.5 * dp) & (g[:, 1] < p.w - 1.5 * dp) & (g[:, 2] > 1.5 * dp) & (g[:, 2] < p.h - 1.5 * dp)
    rho_seg = rho0 * (1.0 + np.abs(s["ev"][s["seg"], 3]).max())
    assert mid.sum() > 20
    assert (gain[mid] > dt * rho0 * U * dp * dp * (1 - 1e-4)).all() and (gain[mid] < dt * rho_seg * U * dp * dp * (1 + 1e-4)).all()
    # the whole wall: the flux through the open-boundary segments, less the thirds that belong to corner vertices
    assert 0.6 * dt * rho0 * U * p.w * p.h < gain[inner].sum() < dt * rho_seg * U * p.w * p.h
    # the vertex takes the imposed velocity and, where the wall is wet, the fluid's density
    assert np.allclose(a["euler_vel"][:st["n"]][inner, :3], [U, 0, 0], atol=1e-6)
    assert np.array_equal(a["vel"][:st["n"]][inner, 3], a["euler_vel"][:st["n"]][inner, 3])
    wet = mid & (g[:, 2] < p.water_level - 2 * dp)
    hyd = p.initial_density(g)
    assert np.abs(a["euler_vel"][:st["n"]][wet, 3] - hyd[wet]).max() < 0.25 * hyd[wet].max()
    # --- the last step: a vertex that holds more than half a particle lets one go
    b = o.sa_vertex_bc_io(st["pos"], s["vel"], s["ggam"], s["ev"], st["vertices"], s["be"], st["vertpos"], info, st["hash"],
                          next_ids, st["cs"], st["nl"], st["n"], dp, dt, 2, nopen)
    half = inner & (st["pos"][:, 3] + gain > 0.5 * ref)
    assert half.sum() > 20 and b["n"] == st["n"] + int(half.sum())
    clones = slice(st["n"], b["n"])
    parents = np.where(half)[0]
    # (the pass runs over the particles in order here, so the clones follow their parents' order)
    assert np.array_equal(b["new_pos"][clones, :3], st["pos"][parents, :3]) and np.allclose(b["new_pos"][clones, 3], ref)
    assert np.array_equal(b["vel"][clones], b["euler_vel"][:st["n"]][parents])
    assert (info_type(b["info"][clones]) == D.PT_FLUID).all() and (b["info"][clones, 0] == D.PT_FLUID).all()
    assert np.array_equal(info_id(b["info"][clones]), next_ids[parents])
    assert np.array_equal(b["next_ids"][:st["n"]][parents], next_ids[parents] + nopen)
    assert np.array_equal(b["hash"][clones], st["hash"][parents] & D.CELLTYPE_BITMASK)
    assert (b["euler_vel"][clones] == 0).all() and (b["vertices"][clones] == 0).all() and np.isnan(b["boundelements"][clones]).all()
    # mass: what the vertices gained from the flux went into the released particles or stayed with the vertices
    new_m = b["new_pos"][:st["n"], 3].astype(np.float64)
    assert abs((new_m[inner].sum() + ref * half.sum()) - (st["pos"][inner, 3].astype(np.float64).sum() + gain[inner].sum())) < 1e-6 * ref * half.sum()


def test_outlet_vertices_take_over_the_mass_of_the_particles_that_left(st):
    p, o = st["problem"], st["oracle"]
    dp = p.m_deltap
    s = _inlet_state(st, -0.2, D.FG_OUTLET)           # the stream leaves through x = 0
    info, vtx, g = s["info"], s["vtx"], s["g"]
    inner = vtx & ((info[:, 0] & D.FG_CORNER) == 0)
    fl = np.where(info_type(info) == D.PT_FLUID)[0]
    near = fl[np.abs(g[fl, 0] - dp) < 1e-6]
    out = near[len(near) // 2]
    pos, vel = st["pos"].copy(), s["vel"].copy()
    pos[out, 0] -= np.float32(1.3 * dp); pos[out, 1] += np.float32(0.22 * dp); pos[out, 2] += np.float32(0.09 * dp)
    v2, g2 = o.find_outgoing_segment(pos, vel, st["vertices"], s["ggam"], st["vertpos"], s["be"], info, st["hash"], st["cs"],
                                     st["nl"], st["n"], float(p.simparams.influenceRadius))
    assert (v2[out, 0] | v2[out, 1]) != 0
    next_ids = np.full(st["n"], 0xFFFFFFFF, dtype=np.uint32)
    kw = dict(deltap=dp, dt=1.0e-3, step=2, num_open_vertices=int(vtx.sum()))
    a = o.sa_vertex_bc_io(pos, vel, g2, s["ev"], v2, s["be"], st["vertpos"], info, st["hash"], next_ids, st["cs"], st["nl"], st["n"], **kw)
    b = o.sa_vertex_bc_io(pos, vel, s["ggam"], s["ev"], st["vertices"], s["be"], st["vertpos"], info, st["hash"], next_ids, st["cs"],
                          st["nl"], st["n"], **kw)      # the same pass without the mark
    assert a["n"] == st["n"] and b["n"] == st["n"]     # an outlet creates nobody
    extra = a["new_pos"][:st["n"], 3].astype(np.float64) - b["new_pos"][:st["n"], 3]
    ids = info_id(st["info"])
    where = {int(v): k for k, v in enumerate(ids)}
    owners = [where[int(v)] for v in v2[out, :3]]
    w = g2[out]
    for k, own in enumerate(owners):
        want = float(w[k]) * float(w[3]) if inner[own] else 0.0      # corner vertices do not take part
        assert abs(extra[own] - want) < 1e-6 * float(w[3])
    others = np.ones(st["n"], dtype=bool); others[owners] = False
    assert (extra[others] == 0).all()
    if all(inner[o_] for o_ in owners):
        assert abs(extra.sum() - float(st["pos"][out, 3])) < 1e-5 * float(st["pos"][out, 3])


def _stream_state(st, U, dt):
    """a uniform stream U ex through the x = 0 wall (velocity inlet with u_E = U ex), the fluid displaced by U dt: arrays for
    the density summation with the gradient of gamma of the fluid initialised by the oracle"""
    p, o = st["problem"], st["oracle"]
    info, seg, vtx, g = _open_wall(st, D.FG_INLET | D.FG_VELOCITY_DRIVEN)
    fl = info_type(info) == D.PT_FLUID
    be = o.sa_compute_vertex_normal(st["boundelements"], st["vertices"], st["info"], st["hash"], st["cs"], st["nl"], st["n"])
    gg = o.sa_init_gamma(st["gradgamma"], st["pos"], be, st["vertpos"], st["info"], st["hash"], st["cs"], st["nl"], st["n"], p.m_deltap)
    vel = st["vel"].copy(); vel[fl, 0] = U
    ev = np.zeros_like(vel); ev[seg | vtx, 0] = U
    new_pos = st["pos"].copy()
    new_pos[fl, 0] = st["pos"][fl, 0] + np.float32(dt) * np.float32(U)
    return dict(info=info, seg=seg, vtx=vtx, g=g, fl=fl, be=be, gg=gg, vel=vel, ev=ev, new_pos=new_pos)


def test_density_summation_with_open_boundaries_enabled_equals_the_plain_one_without_any(st):
    o = st["oracle"]
    s = _stream_state(st, 0.2, 1e-3)
    args = (st["pos"], s["new_pos"], s["vel"])
    v0, g0 = o.sa_density_sum(s["vel"], *args, s["gg"], s["be"], st["vertpos"], st["info"], st["hash"], st["cs"], st["nl"], st["n"])
    v1, g1, _ = o.sa_density_sum_io(s["vel"], *args, s["ev"], s["gg"], s["be"], st["vertpos"], st["info"], st["hash"], st["cs"],
                                    st["nl"], st["n"], 1e-3)      # no open-boundary flags in info: the Eulerian velocities are not read
    assert np.array_equal(v0, v1) and np.array_equal(g0, g1, equal_nan=True)


def test_a_stream_entering_at_the_inlets_own_velocity_does_not_see_the_inlet(st):
    p, o = st["problem"], st["oracle"]
    dp = p.m_deltap
    U, dt = 0.2, 1e-3
    s = _stream_state(st, U, dt)
    args = (st["pos"], s["new_pos"], s["vel"], s["ev"], s["gg"], s["be"], st["vertpos"], s["info"], st["hash"], st["cs"], st["nl"],
            st["n"], dt)
    v, g, sums = o.sa_density_sum_io(s["vel"], *args)
    # the inlet's vertices carried along with the stream: take their mass away and the volumic sums are the same numbers
    pos_light = st["pos"].copy(); pos_light[s["vtx"], 3] = 0.0
    new_light = s["new_pos"].copy(); new_light[s["vtx"], 3] = 0.0
    _, _, sums_light = o.sa_density_sum_io(s["vel"], pos_light, new_light, *args[2:])
    fl = s["fl"]
    # (to rounding: the sums hold terms of ~100 kg/m^3 each; the particle's displacement (x + U dt) - x is U dt to one ulp only)
    assert np.abs(sums[fl] - sums_light[fl]).max() < 1e-6 * p.physparams.rho0[0]
    # ... which the plain summation does not do: a particle next to the inlet, leaving it behind, loses density there
    v_plain, _ = o.sa_density_sum(s["vel"], st["pos"], s["new_pos"], s["vel"], s["gg"], s["be"], st["vertpos"], st["info"], st["hash"],
                                  st["cs"], st["nl"], st["n"])
    near = fl & (np.abs(s["g"][:, 0] - dp) < 1e-6) & (s["g"][:, 2] < p.water_level - 2 * dp) & (s["g"][:, 2] > 2.5 * dp) \
        & (s["g"][:, 1] > 2.5 * dp) & (s["g"][:, 1] < p.w - 2.5 * dp)
    far = fl & (s["g"][:, 0] > 0.3) & (s["g"][:, 0] < p.l - 0.3)
    assert near.sum() >= 8 and far.sum() > 10
    assert (v_plain[near, 3] < v[near, 3]).all()
    assert np.array_equal(v[far, 3], v_plain[far, 3])           # out of reach of the inlet the two are the same pass
    # next to the inlet the density of the uniform stream stays what it was: the translated neighbourhood is the same
    # neighbourhood, and gamma is advanced on both sides of the quotient (imposed gamma = new gamma when only the inlet is in reach)
    assert np.abs(v[near, 3] - s["vel"][near, 3]).max() < 2e-4
    assert np.abs(v_plain[near, 3] - s["vel"][near, 3]).max() > 5 * np.abs(v[near, 3] - s["vel"][near, 3]).max()


def test_viscous_terms_see_the_eulerian_velocity_of_an_open_boundary(st):
    """A stream at U next to an inlet whose Eulerian velocity is U is not braked by it; next to a solid wall it is.  (The forces
    pass of the pressure-driven open vertices and the water depth are not restated.)"""
    p, o = st["problem"], st["oracle"]
    dp = p.m_deltap
    U = 0.2
    s = _stream_state(st, U, 1e-3)
    n = st["n"]
    common = (st["hash"], st["cs"], st["nl"], s["gg"], s["be"], st["vertpos"], n, dp)
    f_plain, _, _ = o.forces_sa(st["pos"], s["vel"], st["info"], *common)
    # open boundaries enabled but none flagged, stale Eulerian velocities cleared: the plain pass
    f_none, _, _ = o.forces_sa_io(st["pos"], s["vel"], np.zeros_like(s["vel"]), st["info"], *common)
    assert np.array_equal(f_plain, f_none)
    f_io, _, _ = o.forces_sa_io(st["pos"], s["vel"], s["ev"], s["info"], *common)
    g, fl = s["g"], s["fl"]
    near = fl & (np.abs(g[:, 0] - dp) < 1e-6) & (g[:, 2] < p.water_level - 2 * dp) & (g[:, 2] > 2.5 * dp) \
        & (g[:, 1] > 2.5 * dp) & (g[:, 1] < p.w - 2.5 * dp)
    far = fl & (g[:, 0] > 0.3) & (g[:, 0] < p.l - 0.3)
    assert near.sum() >= 8 and far.sum() > 10
    assert np.array_equal(f_io[far], f_plain[far])             # out of reach of the inlet nothing changes
    # the wall at x = 0 as a solid wall brakes the layer next to it (the relative velocity is U); as an inlet moving with the
    # stream it does not: the x component of the two passes differs by that viscous force, and with the inlet it is the
    # pressure gradient alone, the same as for a fluid at rest next to a solid wall
    rest = s["vel"].copy(); rest[fl, 0] = 0.0
    f_rest, _, _ = o.forces_sa(st["pos"], rest, st["info"], *common)
    brake = f_plain[near, 0] - f_rest[near, 0]
    assert (brake < 0).all()
    assert np.abs(f_io[near, 0] - f_rest[near, 0]).max() < 0.02 * np.abs(brake).max()
    assert np.allclose(f_io[near, 1:3], f_rest[near, 1:3], atol=1e-3 * np.abs(f_rest[near, 1:3]).max())


def _outlet_wall(st):
    """the x = l wall as pressure-driven open boundary number 2 (on top of whatever _open_wall made of the x = 0 wall)"""
    p = st["problem"]
    g = p.global_pos(st["pos"], st["hash"])
    t = info_type(st["info"])
    seg = (t == D.PT_BOUNDARY) & (st["boundelements"][:, 0] < -0.5) & (np.abs(g[:, 0] - p.l) < 1e-6)
    vtx = (t == D.PT_VERTEX) & (np.abs(g[:, 0] - p.l) < 1e-6)
    return seg, vtx, g


def test_water_depth_is_the_highest_fluid_particle_an_outlet_vertex_looks_down_on(st):
    p, o = st["problem"], st["oracle"]
    info, _, _, _ = _open_wall(st, D.FG_INLET | D.FG_VELOCITY_DRIVEN)          # object 1: velocity driven, never measured
    seg, vtx, g = _outlet_wall(st)
    info[seg | vtx, 0] |= D.FG_OUTLET
    info[seg | vtx, 1] = (info[seg | vtx, 1] & 0xF000) | 2
    n = st["n"]
    depth = np.zeros(3, dtype=np.uint32)
    o.sa_io_water_depth(depth, st["pos"], info, st["hash"], st["cs"], st["nl"], n)
    assert depth[0] == 0 and depth[1] == 0 and depth[2] > 0
    z = o.sa_io_water_depth_z(depth[2])
    # by hand, in double on the global positions: fluid particles within the influence radius of an outlet vertex that is
    # not below them; pairs at the edge of either test (lattice!) may fall on either side in float
    R = float(p.simparams.influenceRadius)
    fl = np.where(info_type(info) == D.PT_FLUID)[0]
    lo, hi = -np.inf, -np.inf
    for v in np.where(vtx)[0]:
        d = g[fl] - g[v]
        r = np.sqrt((d * d).sum(axis=1))
        dz = g[v, 2] - g[fl, 2]
        sure = (r < R - 1e-5) & (dz > 1e-6)
        maybe = (r < R + 1e-5) & (dz > -1e-6)
        if sure.any():
            lo = max(lo, g[fl[sure], 2].max())
        if maybe.any():
            hi = max(hi, g[fl[maybe], 2].max())
    assert np.isfinite(lo) and lo - 1e-5 <= z <= hi + 1e-5
    # the tank's wall reaches above the water: that is the top layer of the fluid
    assert abs(z - g[fl, 2].max()) < 1e-5
    # a maximum that is never cleared here: with the fluid's top layers taken out the number stays; on a cleared array it drops
    pos2 = st["pos"].copy()
    pos2[fl[g[fl, 2] > p.water_level - 2.5 * p.m_deltap], 3] = np.nan
    before = depth.copy()
    o.sa_io_water_depth(depth, pos2, info, st["hash"], st["cs"], st["nl"], n)
    assert np.array_equal(depth, before)
    low = o.sa_io_water_depth(np.zeros(3, dtype=np.uint32), pos2, info, st["hash"], st["cs"], st["nl"], n)
    assert 1.5 * p.m_deltap < z - o.sa_io_water_depth_z(low[2]) < 2.5 * p.m_deltap
    # the scale: 0 is the bottom of the domain, UINT_MAX its top
    assert o.sa_io_water_depth_z(0) == pytest.approx(float(o.p.worldOrigin[2]), abs=1e-7)
    top = float(o.p.worldOrigin[2]) + float(o.p.cellSize[2]) * int(o.p.gridSize[2])
    assert o.sa_io_water_depth_z(0xFFFFFFFF) == pytest.approx(top, rel=1e-6)


def _brezzi_state(st, delta):
    """fluid and walls at the reference density, the segments of the x = 0 wall -- a pressure outlet -- at 1 + delta"""
    p, o = st["problem"], st["oracle"]
    info, seg, vtx, g = _open_wall(st, D.FG_OUTLET)
    be = o.sa_compute_vertex_normal(st["boundelements"], st["vertices"], st["info"], st["hash"], st["cs"], st["nl"], st["n"])
    gg = o.sa_init_gamma(st["gradgamma"], st["pos"], be, st["vertpos"], st["info"], st["hash"], st["cs"], st["nl"], st["n"], p.m_deltap)
    vel = np.zeros_like(st["vel"])
    vel[seg, 3] = delta
    return info, seg, g, be, gg, vel


def test_brezzi_diffusion_without_pressure_driven_segments_is_the_plain_one(st):
    p, o = st["problem"], st["oracle"]
    n, dt = st["n"], 1e-3
    _, _, _, be, gg, _ = _brezzi_state(st, 0.0)
    common = (st["hash"], st["cs"], st["nl"])
    v0, f0 = o.sa_density_diffusion(st["pos"], st["vel"], gg, st["info"], *common, n, dt)
    v1, f1 = o.sa_density_diffusion_io(st["pos"], st["vel"], gg, st["info"], *common, be, st["vertpos"], n, dt, p.m_deltap)
    assert np.array_equal(f0, f1) and np.array_equal(v0, v1)
    # a velocity-driven inlet has no such term either
    info_v, _, _, _ = _open_wall(st, D.FG_INLET | D.FG_VELOCITY_DRIVEN)
    _, f2 = o.sa_density_diffusion_io(st["pos"], st["vel"], gg, info_v, *common, be, st["vertpos"], n, dt, p.m_deltap)
    assert np.array_equal(f0, f2)


def test_brezzi_boundary_term_of_a_pressure_outlet(st):
    """The fluid term with V_b grad W -> |grad gamma_as| / r_as: against the plane x = 0 held at another pressure, a particle whose
    only wall in reach is that plane gets  -2 rho dt (2/(rho + rho_s)) (P - P_s) |grad gamma| / r_as / gamma / rho0."""
    p, o = st["problem"], st["oracle"]
    n, dt, dp = st["n"], 1e-3, p.m_deltap
    rho0 = float(p.physparams.rho0[0])
    common = (st["hash"], st["cs"], st["nl"])
    res = {}
    for delta in (0.0, 0.01, 0.02):
        info, seg, g, be, gg, vel = _brezzi_state(st, delta)
        _, f = o.sa_density_diffusion_io(st["pos"], vel, gg, info, *common, be, st["vertpos"], n, dt, dp)
        res[delta] = f[:, 3].astype(np.float64)
    fl = info_type(st["info"]) == D.PT_FLUID
    R = float(p.simparams.influenceRadius)
    far = fl & (g[:, 0] > R + dp + 1e-4)
    assert far.sum() > 100 and np.array_equal(res[0.0][far], res[0.01][far])            # out of reach of the wall: nothing
    # in reach of the plane and of no other wall (their segments are not pressure driven, but gamma's gradient must be the plane's)
    near = fl & (g[:, 0] < 1.6 * dp) & (g[:, 2] > R + 1.1 * dp) & (g[:, 1] > R + 1.1 * dp) & (g[:, 1] < p.w - R - 1.1 * dp)
    assert near.sum() >= 4
    for delta in (0.01, 0.02):
        rho_s = rho0 * (1 + delta)
        Ps = o.eos_P(delta)
        r_as = np.maximum(g[near, 0], dp)
        want = -((2.0 / (rho0 + rho_s)) * (0.0 - Ps)) * gg[near, 0] / r_as * dt * 2.0 * rho0 / gg[near, 3] / rho0
        got = res[delta][near] - res[0.0][near]
        assert (got > 0).all()                       # a boundary at higher pressure raises the density next to it
        assert np.abs(got - want).max() < 2e-3 * np.abs(want).max()


def test_brezzi_boundary_term_leaves_a_hydrostatic_column_alone(st):
    """(2/(rho + rho_s)) (P - P_s) - g.r vanishes between two points of a hydrostatic column: with the outlet's segments at the
    hydrostatic pressure of their height the term is what the linearisation of the density leaves, orders below the term of a
    boundary that is off by the pressure of the whole column."""
    p, o = st["problem"], st["oracle"]
    n, dt, dp = st["n"], 1e-3, p.m_deltap
    common = (st["hash"], st["cs"], st["nl"])
    info, seg, g, be, gg, _ = _brezzi_state(st, 0.0)
    vel = st["vel"].copy()                                     # the problem's initial state is the hydrostatic one
    hyd = p.initial_density(g)
    vel[seg, 3] = hyd[seg]
    _, f_plain = o.sa_density_diffusion(st["pos"], vel, gg, st["info"], *common, n, dt)
    _, f_hyd = o.sa_density_diffusion_io(st["pos"], vel, gg, info, *common, be, st["vertpos"], n, dt, dp)
    vel_b = vel.copy(); vel_b[seg, 3] = hyd[seg] + np.float32(hyd.max())        # the whole column's pressure on top, everywhere
    _, f_off = o.sa_density_diffusion_io(st["pos"], vel_b, gg, info, *common, be, st["vertpos"], n, dt, dp)
    fl = info_type(st["info"]) == D.PT_FLUID
    near = fl & (g[:, 0] < 1.6 * dp) & (g[:, 2] > 2.5 * dp) & (g[:, 2] < p.water_level - 2 * dp)
    assert near.sum() >= 6
    still = np.abs(f_hyd[near, 3] - f_plain[near, 3])
    pushed = np.abs(f_off[near, 3] - f_plain[near, 3])
    assert still.max() < 1e-3 * pushed.min()


def test_an_open_channel_runs_through_the_whole_sequence():
    """Inlet on the left, pressure outlet on the right, a rebuild before every step: particles are released at the inlet at the
    rate the imposed velocity asks for, those that cross the outlet are taken out, the stream in between keeps its velocity and
    its hydrostatic density, every id is handed out once.  (160 steps of the oracle sequence, tests/sa_helpers.py OracleSaIoSim.)"""
    from sa_helpers import OracleSaIoSim
    p = SABox(0.05, l=1.0, w=0.4, h=0.4, H=0.25)
    dp, U = p.m_deltap, 0.6
    sim = OracleSaIoSim(p, U)
    n0 = sim.n
    t = info_type(sim.info[:n0])
    inlet_vertices = int(((sim.info[:n0, 0] & D.FG_INLET) != 0).__and__(t == D.PT_VERTEX).__and__((sim.info[:n0, 0] & D.FG_CORNER) == 0).sum())
    fluid0 = int((t == D.PT_FLUID).sum())
    ref = float(p.physparams.rho0[0]) * dp ** 3
    for _ in range(160):
        sim.step()
    n = sim.n
    t = info_type(sim.info[:n])
    active = np.isfinite(sim.pos[:n, 3])
    fl = (t == D.PT_FLUID) & active
    g = p.global_pos(sim.pos[:n], sim.hash[:n])
    assert np.isfinite(sim.vel[:n][fl]).all() and np.isfinite(sim.pos[:n][fl]).all()
    # released: one per inlet vertex in the first step (they start at half a particle), then one per vertex every dp / U
    layers = U * sim.t / dp
    assert inlet_vertices <= sim.created <= inlet_vertices * (2 + int(layers))
    assert sim.created >= inlet_vertices * int(layers)
    # taken out: the first layer next to the outlet had dp to go; nobody is left beyond it
    assert sim.removed > 0 and int(fl.sum()) == fluid0 + sim.created - sim.removed
    assert g[fl, 0].max() < p.l + 2 * U * sim.dt and g[fl, 0].min() > -1e-6
    # ids: everybody who was ever released got a different one
    ids = info_id(sim.info[:n])
    assert len(np.unique(ids[t == D.PT_FLUID])) == int((t == D.PT_FLUID).sum())
    # the stream below the surface keeps its velocity and its hydrostatic density
    bulk = fl & (g[:, 0] > 0.2) & (g[:, 0] < p.l - 0.2) & (g[:, 2] < p.water_level - dp)
    assert bulk.sum() > 100
    assert 0.85 * U < sim.vel[:n][bulk, 0].mean() < 1.25 * U
    assert np.abs(sim.vel[:n][bulk, 3] - p.initial_density(g)[bulk]).max() < 0.01
    # the open vertices' masses stay within the clip of +/- 2 reference masses
    ov = (t == D.PT_VERTEX) & ((sim.info[:n, 0] & (D.FG_INLET | D.FG_OUTLET)) != 0)
    assert np.abs(sim.pos[:n][ov, 3]).max() <= 2 * ref * (1 + 1e-6)


def test_the_channel_with_brezzi_diffusion_and_the_measured_water_depth():
    """ChannelIO's options (densitydiffusion<BREZZI>, ENABLE_WATER_DEPTH, src/problems/ChannelIO.cu:45-46): the outlet's pressure
    follows the level the vertex pass measured -- nothing measured yet when the initial conditions are imposed, the top layer of the
    fluid afterwards -- and the diffusion with its open-boundary term keeps the stream as the plain sequence does."""
    from sa_helpers import OracleSaIoSim
    p = SABox(0.05, l=1.0, w=0.4, h=0.4, H=0.25)
    dp, U = p.m_deltap, 0.6
    sim = OracleSaIoSim(p, U, brezzi=True, water_depth=True)
    bottom = float(sim.o.p.worldOrigin[2])
    assert sim.level_seen == [pytest.approx(bottom, abs=1e-7)]         # (the reference reads an array it never cleared here)
    for _ in range(60):
        sim.step()
    n = sim.n
    t = info_type(sim.info[:n])
    fl = (t == D.PT_FLUID) & np.isfinite(sim.pos[:n, 3])
    g = p.global_pos(sim.pos[:n], sim.hash[:n])
    assert np.isfinite(sim.vel[:n][fl]).all() and np.isfinite(sim.pos[:n][fl]).all()
    assert len(sim.level_seen) == 1 + 2 * 60
    levels = np.array(sim.level_seen[1:])
    top = p.water_level - 0.5 * dp                                   # the lattice's top layer
    assert np.abs(levels - top).max() < dp
    assert sim.created > 0
    bulk = fl & (g[:, 0] > 0.2) & (g[:, 0] < p.l - 0.2) & (g[:, 2] < p.water_level - dp)
    assert bulk.sum() > 100
    assert 0.85 * U < sim.vel[:n][bulk, 0].mean() < 1.25 * U
    assert np.abs(sim.vel[:n][bulk, 3] - p.initial_density(g)[bulk]).max() < 0.01
