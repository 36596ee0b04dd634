"""The open-boundary kernels of gpusph_amd/csrc/sa_io.hip run on the CPU, from their own source, against the oracle -- a way to find
logic errors in kernels that have not run on a GPU yet (tests/hostemu_lib.py: sphx_api.hip + sa_io.hip compiled by g++ through a
stand-in for the HIP header, every launch a serial loop over its threads).  These tests mirror tests/test_gpu_sa_io.py, which is
the parity test proper; what passes here is the source's logic, not the device build."""
import numpy as np
import pytest

from gpusph_amd import defs as D
from gpusph_amd.problem import SABox, info_type
from sa_helpers import sa_oracle_state, wall_rows, assert_close_but_for_gamma_spikes
from hostemu_lib import Emu


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.fixture(scope="module")
def ctx():
    st = sa_oracle_state(deltap=0.05)
    p, n = st["problem"], st["n"]
    emu = Emu(p.sphx_params(n))
    g = p.global_pos(st["pos"], st["hash"])
    t = info_type(st["info"])
    seg = (t == D.PT_BOUNDARY) & (st["boundelements"][:, 0] > 0.5) & (np.abs(g[:, 0]) < 1e-6)
    vtx = (t == D.PT_VERTEX) & (np.abs(g[:, 0]) < 1e-6)
    yield dict(st=st, emu=emu, g=g, t=t, seg=seg, vtx=vtx)
    emu.close()


def _flag(c, flags):
    st = c["st"]
    info = st["info"].copy()
    w = c["seg"] | c["vtx"]
    info[w, 0] |= flags
    info[w, 1] = (info[w, 1] & 0xF000) | 1
    return info


def _vp(st):
    return [np.ascontiguousarray(v) for v in st["vertpos"]]


def test_the_five_verified_kernels_also_hold_in_emulation(ctx):
    """what ran bit-exact on the GPU: the harness itself is held to the same answers"""
    st, emu = ctx["st"], ctx["emu"]
    p, o, n = st["problem"], st["oracle"], st["n"]
    info = _flag(ctx, D.FG_INLET | D.FG_VELOCITY_DRIVEN)
    want_info = o.sa_identify_corner_vertices(st["pos"], info, st["hash"], st["vertices"], st["cs"], st["nl"], n)
    got = info.copy()
    emu.call("sphx_sa_identify_corner_vertices", st["pos"], got, st["hash"], st["vertices"], st["cs"], st["nl"], n, n, None)
    assert np.array_equal(got, want_info) and ((want_info[:, 0] & D.FG_CORNER) != 0).sum() > 0
    pos = st["pos"].copy()
    inner = ctx["vtx"] & ((want_info[:, 0] & D.FG_CORNER) == 0)
    pos[inner, 3] *= np.random.default_rng(3).uniform(0.5, 0.9, size=int(inner.sum())).astype(np.float32)
    want_count, want_pos = o.sa_init_io_mass(pos, want_info, st["hash"], st["vertices"], st["cs"], st["nl"], n, p.m_deltap)
    forces = np.zeros_like(pos); newpos = np.zeros_like(pos)
    emu.call("sphx_sa_init_io_mass_vertex_count", st["vertices"], st["hash"], want_info, st["cs"], st["nl"], forces, pos, n, n, None)
    assert np.array_equal(forces[:, 3], want_count)
    emu.call("sphx_sa_init_io_mass", pos, forces, st["vertices"], st["hash"], want_info, st["cs"], st["nl"], newpos, n, n,
             float(np.float32(p.m_deltap)), None)
    assert np.array_equal(_bits(newpos), _bits(want_pos[:n]))


def test_boundary_condition_passes_in_emulation(ctx):
    st, emu, t, seg, vtx = ctx["st"], ctx["emu"], ctx["t"], ctx["seg"], ctx["vtx"]
    p, o, n = st["problem"], st["oracle"], st["n"]
    dp, U, dt = p.m_deltap, 0.2, 2.0e-3
    info = _flag(ctx, D.FG_INLET | D.FG_VELOCITY_DRIVEN)
    info = o.sa_identify_corner_vertices(st["pos"], info, st["hash"], st["vertices"], st["cs"], st["nl"], n)
    be = o.sa_compute_vertex_normal(st["boundelements"], st["vertices"], info, st["hash"], st["cs"], st["nl"], n)
    vel = st["vel"].copy(); vel[t == D.PT_FLUID, 0] = U
    ev0 = np.zeros_like(vel); ev0[seg | vtx, 0] = U
    gg = st["gradgamma"].copy()
    gg[t == D.PT_VERTEX] = (0.0, 0.0, 0.0, 0.5); gg[t == D.PT_FLUID] = (0.0, 0.0, 0.0, 1.0)
    nopen = int(vtx.sum())
    next_ids = np.full(n, 0xFFFFFFFF, dtype=np.uint32)
    next_ids[vtx] = n + np.arange(nopen, dtype=np.uint32)
    A = n + nopen
    vp = _vp(st)

    def grow(a, fill=0):
        out = np.full((A,) + a.shape[1:], fill, dtype=a.dtype)
        out[:n] = a
        return out
    for step in (1, 2):
        want_v, want_g, want_e = o.sa_segment_bc_io(st["pos"], vel, gg, ev0, st["vertices"], be, info, st["hash"], st["cs"], st["nl"], n, step)
        d_vel, d_gg, d_ev = vel.copy(), gg.copy(), ev0.copy()
        emu.call("sphx_sa_segment_bc_io", d_vel, d_gg, d_ev, st["pos"], st["vertices"], be, info, st["hash"], st["cs"], st["nl"],
                 n, n, step, None)
        scale = np.abs(want_e[:, :3]).max()
        assert np.abs(d_ev[:, :3] - want_e[:, :3]).max() < 2e-5 * scale
        assert np.abs(d_ev[:, 3] - want_e[:, 3]).max() < 2e-5 * np.abs(want_e[:, 3]).max() + 2e-7
        assert np.abs(d_vel[:, 3] - want_v[:, 3]).max() < 2e-5 * np.abs(want_v[:, 3]).max() + 2e-7
        assert np.array_equal(_bits(d_vel[:, :3]), _bits(want_v[:, :3]))
        assert np.array_equal(_bits(d_gg), _bits(want_g))
        a = o.sa_vertex_bc_io(st["pos"], want_v, want_g, want_e, st["vertices"], be, st["vertpos"], info, st["hash"], next_ids,
                              st["cs"], st["nl"], n, dp, dt, step, nopen)
        d_vel, d_gg, d_ev = grow(want_v), grow(want_g), grow(want_e)
        d_pos, d_newpos = grow(st["pos"]), grow(st["pos"])
        d_forces = np.zeros((A, 4), dtype=np.float32)
        d_vert, d_be2, d_info2, d_hash = grow(st["vertices"]), grow(be), grow(info), grow(st["hash"])
        d_ids = grow(next_ids, 0xFFFFFFFF)
        d_count = np.array([n], dtype=np.uint32)
        emu.call("sphx_sa_vertex_bc_io", d_vel, d_pos, d_newpos, d_gg, d_ev, d_forces, d_vert, d_be2, vp[0], vp[1], vp[2], d_info2,
                 d_hash, d_ids, d_count, st["cs"], st["nl"], n, n, A, float(np.float32(dp)), float(np.float32(dt)), step, nopen, None)
        n2 = int(d_count[0])
        assert n2 == a["n"]
        mref = float(p.physparams.rho0[0]) * dp ** 3
        assert np.abs(d_newpos[:n, 3] - a["new_pos"][:n, 3]).max() < 2e-5 * mref
        assert np.array_equal(_bits(d_newpos[:n, :3]), _bits(a["new_pos"][:n, :3]))
        assert np.abs(d_ev[:n] - a["euler_vel"][:n]).max() < 2e-5 * max(scale, 1e-3)
        assert np.abs(d_vel[:n, 3] - a["vel"][:n, 3]).max() < 2e-5 * np.abs(a["vel"][:n, 3]).max() + 2e-7
        assert np.abs(d_gg[:n] - a["ggam"][:n]).max() < 2e-6
        if step == 2:
            assert n2 > n
            got_id = d_info2[n:n2, 2].astype(np.uint32) | (d_info2[n:n2, 3].astype(np.uint32) << 16)
            want_id = a["info"][n:n2, 2].astype(np.uint32) | (a["info"][n:n2, 3].astype(np.uint32) << 16)
            go, wo = np.argsort(got_id), np.argsort(want_id)
            assert np.array_equal(got_id[go], want_id[wo])
            assert np.array_equal(d_info2[n:n2][go], a["info"][n:n2][wo])
            assert np.array_equal(_bits(d_newpos[n:n2][go]), _bits(a["new_pos"][n:n2][wo]))
            assert np.abs(d_vel[n:n2][go] - a["vel"][n:n2][wo]).max() < 2e-5 * max(scale, 1e-3)
            assert np.array_equal(d_hash[n:n2][go], a["hash"][n:n2][wo])
            assert (d_ev[n:n2] == 0).all() and (d_vert[n:n2] == 0).all()
            assert np.isnan(d_be2[n:n2]).all()
            assert np.array_equal(d_ids[:n], a["next_ids"][:n])
            assert np.array_equal(d_ids[n:n2][go], a["next_ids"][n:n2][wo])


def test_outlet_vertices_take_over_outgoing_particles_in_emulation(ctx):
    """the last step at an outlet: the marked particle's mass goes to the three vertices of the segment it crossed"""
    st, emu, t, seg, vtx, g = ctx["st"], ctx["emu"], ctx["t"], ctx["seg"], ctx["vtx"], ctx["g"]
    p, o, n = st["problem"], st["oracle"], st["n"]
    dp, dt = p.m_deltap, 2.0e-3
    info = _flag(ctx, D.FG_OUTLET)
    info = o.sa_identify_corner_vertices(st["pos"], info, st["hash"], st["vertices"], st["cs"], st["nl"], n)
    be = o.sa_compute_vertex_normal(st["boundelements"], st["vertices"], info, st["hash"], st["cs"], st["nl"], n)
    fl = np.where(t == D.PT_FLUID)[0]
    near = fl[np.abs(g[fl, 0] - dp) < 1e-6]
    out = near[len(near) // 2]
    pos2, vel2 = st["pos"].copy(), st["vel"].copy()
    pos2[out, 0] -= np.float32(1.3 * dp); pos2[out, 1] += np.float32(0.22 * dp); pos2[out, 2] += np.float32(0.09 * dp)
    vel2[out, 0] = -0.4
    gg = st["gradgamma"].copy()
    gg[t == D.PT_VERTEX] = (0.0, 0.0, 0.0, 0.5); gg[t == D.PT_FLUID] = (0.0, 0.0, 0.0, 1.0)
    infl = float(np.float32(p.simparams.influenceRadius))
    vert2, gg2 = o.find_outgoing_segment(pos2, vel2, st["vertices"], gg, st["vertpos"], be, info, st["hash"], st["cs"], st["nl"], n, infl)
    assert (vert2[out, 0] | vert2[out, 1]) != 0
    # the emulated marking agrees (it did on the GPU)
    d_vert, d_gg = st["vertices"].copy(), gg.copy()
    vp = _vp(st)
    emu.call("sphx_sa_find_outgoing_segment", pos2, vel2, d_vert, d_gg, vp[0], vp[1], vp[2], be, info, st["hash"], st["cs"], st["nl"],
             n, n, infl, None)
    assert np.array_equal(d_vert, vert2) and np.array_equal(_bits(d_gg), _bits(gg2))
    ev0 = np.zeros_like(vel2); ev0[seg | vtx, 3] = p.initial_density(g)[seg | vtx]
    nopen = int(vtx.sum())
    next_ids = np.full(n, 0xFFFFFFFF, dtype=np.uint32)
    next_ids[vtx] = n + np.arange(nopen, dtype=np.uint32)
    a = o.sa_vertex_bc_io(pos2, vel2, gg2, ev0, vert2, be, st["vertpos"], info, st["hash"], next_ids, st["cs"], st["nl"], n, dp, dt,
                          2, nopen, room=0)
    d_vel, d_gg, d_ev, d_newpos = vel2.copy(), gg2.copy(), ev0.copy(), pos2.copy()
    d_forces = np.zeros_like(pos2)
    d_vert, d_be2, d_info2, d_hash, d_ids = vert2.copy(), be.copy(), info.copy(), st["hash"].copy(), next_ids.copy()
    d_count = np.array([n], dtype=np.uint32)
    emu.call("sphx_sa_vertex_bc_io", d_vel, pos2, d_newpos, d_gg, d_ev, d_forces, d_vert, d_be2, vp[0], vp[1], vp[2], d_info2,
             d_hash, d_ids, d_count, st["cs"], st["nl"], n, n, n, float(np.float32(dp)), float(np.float32(dt)), 2, nopen, None)
    assert int(d_count[0]) == n == a["n"]
    mref = float(p.physparams.rho0[0]) * dp ** 3
    moved = np.abs(a["new_pos"][:n, 3] - pos2[:, 3]) > 1e-3 * mref
    assert moved.sum() >= 2
    assert np.abs(d_newpos[:, 3] - a["new_pos"][:n, 3]).max() < 2e-5 * mref
    assert np.array_equal(d_vert, a["vertices"][:n])
    assert np.abs(d_ev - a["euler_vel"][:n]).max() < 2e-5 * max(np.abs(a["euler_vel"][:n]).max(), 1e-3)


def _stream(ctx, U, dt):
    st, t, seg, vtx = ctx["st"], ctx["t"], ctx["seg"], ctx["vtx"]
    p, o, n = st["problem"], st["oracle"], st["n"]
    info = _flag(ctx, D.FG_INLET | D.FG_VELOCITY_DRIVEN)
    fl = t == D.PT_FLUID
    be = o.sa_compute_vertex_normal(st["boundelements"], st["vertices"], st["info"], st["hash"], st["cs"], st["nl"], n)
    gg = o.sa_init_gamma(st["gradgamma"], st["pos"], be, st["vertpos"], st["info"], st["hash"], st["cs"], st["nl"], n, p.m_deltap)
    vel = st["vel"].copy(); vel[fl, 0] = U
    ev = np.zeros_like(vel); ev[seg | vtx, 0] = U
    new_pos = st["pos"].copy(); new_pos[fl, 0] = st["pos"][fl, 0] + np.float32(dt) * np.float32(U)
    return info, fl, be, gg, vel, ev, new_pos


def test_density_summation_and_forces_in_emulation(ctx):
    st, emu = ctx["st"], ctx["emu"]
    p, o, n = st["problem"], st["oracle"], st["n"]
    dp, U, dt = p.m_deltap, 0.2, 1.0e-3
    info, fl, be, gg, vel, ev, new_pos = _stream(ctx, U, dt)
    vp = _vp(st)
    wall = wall_rows(p, st["nl"], info, n)
    want_v, want_g, want_s = o.sa_density_sum_io(vel, st["pos"], new_pos, vel, ev, gg, be, st["vertpos"], info, st["hash"], st["cs"],
                                                 st["nl"], n, dt)
    d_nv, d_ng, d_f = vel.copy(), np.zeros_like(gg), np.zeros_like(vel)
    emu.call("sphx_sa_density_sum_io", d_nv, d_ng, d_f, st["pos"], new_pos, vel, ev, gg, be, vp[0], vp[1], vp[2], info, st["hash"],
             st["cs"], st["nl"], n, n, float(np.float32(dt)), None)
    assert np.abs(d_f[fl, 3] - want_s[fl]).max() < 2e-5 * np.abs(want_s[fl]).max() + 1e-3
    assert_close_but_for_gamma_spikes(d_nv[fl, 3], want_v[fl, 3], 2e-6, 1.0, what="density after the summation", wall=wall[fl])
    assert_close_but_for_gamma_spikes(d_ng[fl], want_g[fl], 2e-5, np.abs(want_g[fl, :3]).max(), what="gamma after the summation", wall=wall[fl])
    nf = ~fl
    assert np.array_equal(_bits(d_ng[nf]), _bits(gg[nf]))
    # forces (the CFL maxima are block reductions: not emulated)
    import ctypes as C
    want_f, want_cfl, nb = o.forces_sa_io(st["pos"], vel, ev, info, st["hash"], st["cs"], st["nl"], gg, be, st["vertpos"], n, dp)
    d_forces = np.zeros_like(vel)
    d_cfl = np.zeros(4 * nb + 64, dtype=np.float32)
    d_cflg = np.zeros(((n + 3) // 4) * 4 + 4 * nb + 64, dtype=np.float32)
    hnb = C.c_uint32(0)
    emu.call("sphx_forces_basicstep_sa_io", d_forces, d_cfl, d_cflg, st["pos"], vel, ev, info, st["hash"], st["cs"], st["nl"], gg, be,
             vp[0], vp[1], vp[2], n, 0, n, float(np.float32(dp)), 0, C.addressof(hnb), None)
    assert hnb.value == nb
    scale = np.abs(want_f[fl, :3]).max()
    assert_close_but_for_gamma_spikes(d_forces[fl, :3], want_f[fl, :3], 1e-4, scale, what="forces with open boundaries", wall=wall[fl])


def test_brezzi_diffusion_and_water_depth_in_emulation(ctx):
    st, emu, seg, vtx = ctx["st"], ctx["emu"], ctx["seg"], ctx["vtx"]
    p, o, n = st["problem"], st["oracle"], st["n"]
    dp, dt = p.m_deltap, 1.0e-3
    info = _flag(ctx, D.FG_OUTLET)
    fl = ctx["t"] == D.PT_FLUID
    be = o.sa_compute_vertex_normal(st["boundelements"], st["vertices"], st["info"], st["hash"], st["cs"], st["nl"], n)
    gg = o.sa_init_gamma(st["gradgamma"], st["pos"], be, st["vertpos"], st["info"], st["hash"], st["cs"], st["nl"], n, dp)
    vel = st["vel"].copy()
    vel[seg, 3] = vel[seg, 3] + np.float32(0.01)
    vp = _vp(st)
    _, want = o.sa_density_diffusion_io(st["pos"], vel, gg, info, st["hash"], st["cs"], st["nl"], be, st["vertpos"], n, dt, dp)
    _, plain = o.sa_density_diffusion(st["pos"], vel, gg, st["info"], st["hash"], st["cs"], st["nl"], n, dt)
    assert np.abs(want[fl, 3] - plain[fl, 3]).max() > 100 * np.abs(plain[fl, 3]).max()
    d_f = np.zeros_like(vel)
    emu.call("sphx_sa_compute_density_diffusion_io", d_f, st["pos"], vel, gg, be, vp[0], vp[1], vp[2], info, st["hash"], st["cs"],
             st["nl"], n, n, float(np.float32(dp)), float(np.float32(dt)), None)
    wall = wall_rows(p, st["nl"], info, n)
    assert_close_but_for_gamma_spikes(d_f[fl, 3], want[fl, 3], 3e-5, np.abs(want[fl, 3]).max(),
                                      what="Brezzi diffusion with a pressure outlet", wall=wall[fl])
    want_d = o.sa_io_water_depth(np.zeros(2, dtype=np.uint32), st["pos"], info, st["hash"], st["cs"], st["nl"], n)
    assert want_d[1] > 0
    d_depth = np.zeros(2, dtype=np.uint32)
    emu.call("sphx_sa_io_water_depth", d_depth, st["pos"], info, st["hash"], st["cs"], st["nl"], n, 0, n, None)
    assert np.array_equal(d_depth, want_d)


def test_the_open_channel_over_the_emulated_kernels_follows_the_oracle():
    """tests/sa_helpers.py OracleSaIoSim twice, with ChannelIO's options (Brezzi diffusion, water depth): once as it is and once with
    every open-boundary pass done by the emulated kernels (EmuPasses) -- 40 steps with a rebuild before each, particles released
    and re-sorted.  Same counts, same ids, the states within what the product's own |grad gamma_as| allows."""
    from sa_helpers import OracleSaIoSim
    from hostemu_lib import EmuPasses
    p = SABox(0.05, l=1.0, w=0.4, h=0.4, H=0.25)
    U = 0.6
    ref = OracleSaIoSim(p, U, brezzi=True, water_depth=True)
    sim = OracleSaIoSim(SABox(0.05, l=1.0, w=0.4, h=0.4, H=0.25), U, brezzi=True, water_depth=True)
    emu = Emu(sim.problem.sphx_params(sim.cap))
    sim.o = EmuPasses(sim.o, emu)
    # the initialisation ran on the oracle on both sides; redo it through the kernels on the same initial state
    # (corner flags, vertex masses, first conditions): same answers expected, so simply check them from here on
    for _ in range(40):
        ref.step(); sim.step()
        assert sim.n == ref.n and sim.created == ref.created and sim.removed == ref.removed
    n = sim.n
    assert sim.created > 0
    for name in ("sphx_sa_segment_bc_io", "sphx_sa_vertex_bc_io", "sphx_sa_density_sum_io", "sphx_forces_basicstep_sa_io",
                 "sphx_sa_compute_density_diffusion_io", "sphx_sa_io_water_depth", "sphx_sa_find_outgoing_segment",
                 "sphx_sa_disable_outgoing_parts"):
        assert sim.o.calls.get(name, 0) >= 40, name
    # row for row by particle id (the rows of the particles released in one step are handed out in any order, and a cell keeps
    # the order its particles arrive in)
    from gpusph_amd.problem import info_id
    a, b = np.argsort(info_id(sim.info[:n]), kind="stable"), np.argsort(info_id(ref.info[:n]), kind="stable")
    assert np.array_equal(sim.info[:n][a], ref.info[:n][b])
    # (a particle within rounding of a cell face may be hashed to either side: global positions are compared)
    assert (sim.hash[:n][a] != ref.hash[:n][b]).mean() < 0.01
    spos, svel, rpos, rvel = sim.pos[:n][a].copy(), sim.vel[:n][a], ref.pos[:n][b].copy(), ref.vel[:n][b]
    spos[:, :3] = sim.problem.global_pos(sim.pos[:n], sim.hash[:n])[a]
    rpos[:, :3] = p.global_pos(ref.pos[:n], ref.hash[:n])[b]
    act = np.isfinite(rpos[:, 3])
    assert np.array_equal(act, np.isfinite(spos[:, 3]))
    cell = float(sim.o.p.cellSize[0])
    assert_close_but_for_gamma_spikes(spos[act, :3], rpos[act, :3], 2e-5, cell, spike=5.0, what="positions after 40 steps")      # measured (SPHX_TEST_REPORT): 0.26 % beyond, worst 2.5 tolerances
    assert_close_but_for_gamma_spikes(svel[act, :3], rvel[act, :3], 1e-3, U, frac=0.002, spike=2.0, what="velocities after 40 steps")      # worst 0.34
    # densities: at the far end of an element's support the closed form of |grad gamma_as| cancels to ~4e-3 of the wall's own
    # gradient in EITHER formulation (both are that far from the float64 value there, tests/test_sa_wall_gamma.py), a fifth of
    # the local value for a particle 1.5 h from an open boundary; the density summation integrates it, and after 40 steps of a
    # stream that crosses two such zones a tenth of the particles carry more than 2e-6, none more than 1e-4 (hydrostatic
    # density of this tank: 4e-3)
    assert_close_but_for_gamma_spikes(svel[act, 3], rvel[act, 3], 2e-6, 1.0, frac=0.15, spike=15.0, what="densities after 40 steps")      # 12.8 % beyond, worst 6.6
    mref = float(p.physparams.rho0[0]) * p.m_deltap ** 3
    assert np.abs(spos[act, 3] - rpos[act, 3]).max() < 1e-3 * mref
    assert np.abs(np.array(sim.level_seen) - np.array(ref.level_seen)).max() < 1e-5
    emu.close()


def test_the_engines_driver_over_the_bindings_and_the_emulated_kernels():
    """gpusph_amd.multigpu's open-boundary sequence with the open-boundary passes going through HipKernels' OWN binding code
    (gpusph_amd/kernels.py: argument order, scalars, the two-call passes) into the emulated library, everything else through the
    oracle backend -- against the same driver over the oracle alone (which tests/test_engine_sa_io.py holds bit for bit against
    the independent restatement of the command sequence)."""
    import types
    import torch
    from gpusph_amd.kernels import HipKernels
    from gpusph_amd.multigpu import MultiGpuEngine
    from gpusph_amd.problem import SAChannelIO, info_id
    from oracle_kernels import OracleKernels

    IO = ("sa_identify_corner_vertices", "sa_init_io_mass_vertex_count", "sa_init_io_mass", "sa_segment_bc_io", "sa_vertex_bc_io", "sa_find_outgoing_segment",
          "sa_disable_outgoing_parts", "sa_density_sum_io", "sa_io_water_depth")

    class EmuIoKernels(OracleKernels):
        def __init__(self, problem, alloc):
            super().__init__(problem, alloc)
            self.emu = Emu(problem.sphx_params(alloc))
            shim = object.__new__(HipKernels)
            shim.lib, shim.ctx, shim.params = self.emu.lib, types.SimpleNamespace(handle=self.emu.h), self.sp
            shim._s = lambda: None
            shim.memset = lambda t, byte: t.view(torch.uint8).fill_(byte)
            self.shim = shim
            self.calls = {}
            for name in IO:
                setattr(self, name, self._bound(name))

        def _bound(self, name):
            fn = getattr(HipKernels, name)

            def call(*a, **kw):
                self.calls[name] = self.calls.get(name, 0) + 1
                return fn(self.shim, *a, **kw)
            return call

        def forces_sa_io(self, forces, cfl, pos, vel, eulervel, *rest, **kw):
            # the CFL maxima are block reductions (not emulated): the oracle's; the forces: the emulated kernel's
            nb = OracleKernels.forces_sa_io(self, forces, cfl, pos, vel, eulervel, *rest, **kw)
            keep = forces.clone()
            scr_cfl = torch.zeros_like(cfl)
            scr_g = torch.zeros_like(kw["cfl_gamma"]) if kw.get("cfl_gamma") is not None else None
            self.calls["forces_sa_io"] = self.calls.get("forces_sa_io", 0) + 1
            nb2 = HipKernels.forces_sa_io(self.shim, forces, scr_cfl, pos, vel, eulervel, *rest, cfl_gamma=scr_g)
            assert nb2 == nb
            self.last_force_gap = float((forces - keep).abs().max())
            return nb

    def diffusion(self_k):
        # HipKernels.sa_density_diffusion_io is two calls; the second (sphx_apply_density_diffusion) belongs to the integration
        # engine, which is not among the emulated files: the first call as the binding makes it, the update by hand
        def call(forces, pos, vel, ggam, boundelements, vertpos, info, hash_, cellStart, neibslist, n, range_end, dt):
            p = capi_ptr
            self_k.calls["sa_density_diffusion_io"] = self_k.calls.get("sa_density_diffusion_io", 0) + 1
            rc = self_k.emu.lib.sphx_sa_compute_density_diffusion_io(self_k.emu.h, p(forces), p(pos), p(vel), p(ggam), p(boundelements),
                                                                     p(vertpos[0]), p(vertpos[1]), p(vertpos[2]), p(info), p(hash_),
                                                                     p(cellStart), p(neibslist), n, range_end, self_k.sp.deltap,
                                                                     float(np.float32(dt)), None)
            assert rc == 0
            fluid = (info[:range_end, 0].to(torch.int32) & 7) == 0
            rows = torch.nonzero(fluid).flatten()
            vel[rows, 3] = vel[rows, 3] + forces[rows, 3] * np.float32(dt)
        return call
    from gpusph_amd.capi import ptr as capi_ptr

    mk = lambda: SAChannelIO(0.05, U=0.6)
    alloc = int(mk().num_particles * 1.6)
    ref = MultiGpuEngine(mk(), "cpu", 0, 1, kernels=OracleKernels(mk(), alloc), allocated=alloc)
    ke = EmuIoKernels(mk(), alloc)
    ke.sa_density_diffusion_io = diffusion(ke)
    eng = MultiGpuEngine(mk(), "cpu", 0, 1, kernels=ke, allocated=alloc)
    for it in range(30):
        ref.step(); eng.step()
        assert eng.n_local == ref.n_local and eng.io_created == ref.io_created, it
    for name in IO + ("forces_sa_io", "sa_density_diffusion_io"):
        assert ke.calls.get(name, 0) >= (1 if name in ("sa_identify_corner_vertices", "sa_init_io_mass_vertex_count", "sa_init_io_mass") else 30), name
    n = eng.n_local
    a = np.argsort(info_id(eng.info[:n].numpy().view(np.uint16)), kind="stable")
    b = np.argsort(info_id(ref.info[:n].numpy().view(np.uint16)), kind="stable")
    assert np.array_equal(eng.info[:n].numpy()[a], ref.info[:n].numpy()[b])
    p = eng.problem
    gp = p.global_pos(eng.pos[:n].numpy(), eng.hash[:n].numpy().view(np.uint32))[a]
    gr = p.global_pos(ref.pos[:n].numpy(), ref.hash[:n].numpy().view(np.uint32))[b]
    act = np.isfinite(ref.pos[:n].numpy()[b][:, 3])
    cell = float(p.m_cellsize[0])
    assert_close_but_for_gamma_spikes(gp[act], gr[act], 2e-5, cell, spike=4.0, what="positions after 30 steps")      # measured: 0.12 % beyond, worst 1.7 tolerances
    assert_close_but_for_gamma_spikes(eng.vel[:n].numpy()[a][act, :3], ref.vel[:n].numpy()[b][act, :3], 1e-3, 0.6, frac=0.002, spike=2.0,
                                      what="velocities after 30 steps")      # worst 0.59
    assert_close_but_for_gamma_spikes(eng.vel[:n].numpy()[a][act, 3], ref.vel[:n].numpy()[b][act, 3], 2e-6, 1.0, frac=0.15, spike=15.0,
                                      what="densities after 30 steps")      # 13.4 % beyond, worst 7.7
    assert np.array_equal(eng.next_ids[:n].numpy()[a], ref.next_ids[:n].numpy()[b])
    ke.emu.close()


def test_flux_computation_in_emulation(ctx):
    import ctypes as C
    st, emu, seg = ctx["st"], ctx["emu"], ctx["seg"]
    o, n = st["oracle"], st["n"]
    info = _flag(ctx, D.FG_INLET | D.FG_VELOCITY_DRIVEN)
    ev = np.random.default_rng(9).normal(size=(n, 4)).astype(np.float32)
    be = st["boundelements"]
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)
    want = np.full(3, 7.0, dtype=np.float32)                     # (the sums start from zero whatever the array held)
    o.L.orc_flux_computation(ptr(want), ptr(info), ptr(ev), ptr(be), C.c_uint32(n), C.c_uint32(3))
    got = np.full(3, -3.0, dtype=np.float32)
    emu.call("sphx_flux_computation", got, info, ev, be, n, n, 3, None)
    assert np.array_equal(got, want) and want[1] != 0 and want[0] == 0 and want[2] == 0
    by_hand = (be[seg, 3].astype(np.float64) * (ev[seg, :3].astype(np.float64) * be[seg, :3]).sum(axis=1)).sum()
    assert want[1] == pytest.approx(by_hand, rel=1e-4)


def test_solid_wall_passes_in_emulation(ctx):
    """sa_bounds.hip (verified on the GPU; its launches rewritten by the harness, the source untouched) over its list walkers: the
    vertex normals -- bit for bit, with and without open-boundary flags: a vertex of an open boundary averages over that boundary's
    segments only --, the initial gamma, the segment and vertex conditions and the density summation to the tolerances of their
    GPU tests (tests/test_gpu_sa.py)."""
    from gpusph_amd import defs as DD
    st, emu = ctx["st"], ctx["emu"]
    p, o, n = st["problem"], st["oracle"], st["n"]
    P = emu.params
    vp = _vp(st)
    for info in (st["info"], _flag(ctx, D.FG_INLET | D.FG_VELOCITY_DRIVEN)):
        want = o.sa_compute_vertex_normal(st["boundelements"], st["vertices"], info, st["hash"], st["cs"], st["nl"], n)
        got = st["boundelements"].copy()
        emu.call("sphx_sa_compute_vertex_normal", got, st["vertices"], info, st["hash"], st["cs"], st["nl"], n, n, None)
        assert np.array_equal(np.isnan(got), np.isnan(want))
        assert np.array_equal(_bits(got)[~np.isnan(got)], _bits(want)[~np.isnan(want)])
    flagged = o.sa_compute_vertex_normal(st["boundelements"], st["vertices"], _flag(ctx, D.FG_INLET), st["hash"], st["cs"], st["nl"], n)
    plain = o.sa_compute_vertex_normal(st["boundelements"], st["vertices"], st["info"], st["hash"], st["cs"], st["nl"], n)
    edge = ctx["vtx"] & (np.abs(ctx["g"][:, 2]) < 1e-6)            # on the edge between the open wall and the floor
    assert edge.sum() > 0 and not np.array_equal(flagged[edge, :3], plain[edge, :3])
    be = plain
    want_g = o.sa_init_gamma(st["gradgamma"], st["pos"], be, st["vertpos"], st["info"], st["hash"], st["cs"], st["nl"], n, p.m_deltap)
    got_g = st["gradgamma"].copy()
    emu.call("sphx_sa_init_gamma", got_g, st["gradgamma"], st["pos"], be, vp[0], vp[1], vp[2], st["info"], st["hash"], st["cs"], st["nl"],
             float(P.slength), float(P.influenceradius), float(np.float32(p.m_deltap)), 5e-5, n, n, None)
    rows = info_type(st["info"]) != DD.PT_BOUNDARY
    assert np.abs(got_g[rows, 3] - want_g[rows, 3]).max() < 2e-6
    assert_close_but_for_gamma_spikes(got_g[rows, :3], want_g[rows, :3], 2e-5, np.abs(want_g[rows, :3]).max(), what="grad gamma")
    # boundary conditions of step 1 on the initial state
    vel = st["vel"].copy()
    want_v, want_gg = o.sa_segment_bc(st["pos"], vel, want_g, st["vertices"], be, st["info"], st["hash"], st["cs"], st["nl"], n, 1)
    gv, gg = vel.copy(), want_g.copy()
    emu.call("sphx_sa_segment_bc", gv, gg, st["pos"], st["vertices"], be, st["info"], st["hash"], st["cs"], st["nl"], n, n,
             float(np.float32(p.m_deltap)), float(P.slength), float(P.influenceradius), 1, DD.SIMULATE, None)
    assert np.abs(gv[:, 3] - want_v[:, 3]).max() < 2e-5 * np.abs(want_v[:, 3]).max() + 2e-7
    assert np.array_equal(np.isnan(gg), np.isnan(want_gg)) and np.allclose(gg[~np.isnan(gg)], want_gg[~np.isnan(want_gg)], rtol=0, atol=1e-6)
    want_v2 = o.sa_vertex_bc(st["pos"], want_v, want_gg, st["info"], st["hash"], st["cs"], st["nl"], n)
    gv2 = want_v.copy()
    emu.call("sphx_sa_vertex_bc", gv2, want_gg, st["pos"], st["info"], st["hash"], st["cs"], st["nl"], n, n,
             float(np.float32(p.m_deltap)), float(P.slength), float(P.influenceradius), 1, DD.SIMULATE, None)
    assert np.abs(gv2[:, 3] - want_v2[:, 3]).max() < 2e-5 * np.abs(want_v2[:, 3]).max() + 2e-7
