"""Shared set-up of the SA_BOUNDARY tests: the SABox mirror through the neighbour phase of the ORACLE."""
import numpy as np

from gpusph_amd import defs as D
from gpusph_amd.problem import SABox, info_id, info_type
from oracle_lib import Oracle, orc_params_from


def wall_rows(problem, nl, info, n):
    """which of the n sorted particles have a boundary element in reach: a non-empty boundary section of the neighbour list
    (walls and vertices sit on the elements themselves: always)"""
    sp = problem.simparams
    nl2 = np.asarray(nl).view(np.uint16).reshape(sp.neiblistsize, -1)[:, :n]
    fluid = (np.asarray(info).view(np.uint16).reshape(-1, 4)[:n, 0] & 7) == D.PT_FLUID
    return (nl2[sp.neibboundpos] != 0xFFFF) | ~fluid


def assert_close_but_for_gamma_spikes(got, want, tol, scale=None, frac=0.01, spike=200.0, what="", wall=None, away=1.0):
    """|got - want| <= tol * scale for all but a fraction `frac` of the entries, and <= spike * tol * scale for those.
    `wall` (one flag per row of got / want, see wall_rows): the allowance is for the rows that have a boundary element in
    reach ONLY -- every other row must hold away * tol * scale (away = 1 for a single pass; after some steps of a run the
    difference of a wall row has reached its neighbours through the pair sums, diluted: the measured factor of each call site is
    in its line), so that an indexing or ordering bug of the particle <- particle sums cannot hide behind it.

    The closed form of |grad gamma_as| (edge antiderivatives that cancel against each other and against the angle
    bookkeeping) is ill-conditioned for some positions of a particle relative to an element: two float evaluations of the
    SAME formula in different operation orders differ there by up to ~1e-3 of |grad gamma|, and the reference's own float
    evaluation is that far from the float64 value of its formula (tests/test_sa_wall_gamma.py measures both against the
    reference's numbers).  The kernels integrate the same expressions in their own order of operations (set-up of an element
    once, polynomial collected by powers of the distance), so the few particles that have such an element in reach carry a
    difference the bulk does not; everything derived from gamma (SA forces, density summation, trajectories) inherits it.

    SPHX_TEST_REPORT=<file>: every call appends what it measured (share of the entries beyond the tolerance, worst entry, worst
    row without an element in reach, in units of tol * scale) -- how the allowances of the call sites were set."""
    import os
    got = np.asarray(got, dtype=np.float64); want = np.asarray(want, dtype=np.float64)
    s = float(np.abs(want).max() if scale is None else scale)
    err = np.abs(got - want)
    bad = float((err > tol * s).mean()) if err.size else 0.0
    worst = float(err.max()) if err.size else 0.0
    report = os.environ.get("SPHX_TEST_REPORT")
    if report:
        w = np.asarray(wall, dtype=bool) if wall is not None else None
        aw = float(err[~w].max()) if (w is not None and (~w).any()) else float("nan")
        with open(report, "a") as f:
            f.write("%-60s n=%d bad=%.4f (allowed %.4f) worst=%.2f tol (allowed %.1f) away_worst=%.2f tol (allowed %.1f) wall_share=%.3f\n" % (
                what, err.size, bad, frac, worst / max(tol * s, 1e-300), spike, aw / max(tol * s, 1e-300), away,
                float(w.mean()) if w is not None else float("nan")))
    assert bad <= frac and worst <= spike * tol * s, "%s: %.3g of the entries beyond %.1e of the scale %.3g (allowed %.3g), worst %.3g of the scale (allowed %.3g)" % (
        what, bad, tol, s, frac, worst / max(s, 1e-300), spike * tol)
    if wall is not None:
        wall = np.asarray(wall, dtype=bool)
        assert len(wall) == len(err), "%s: one wall flag per row" % what
        away_err = err[~wall]
        assert away_err.size == 0 or away_err.max() <= away * tol * s, "%s: a row with no boundary element in reach is %.3g of the scale off (allowed %.1e)" % (
            what, away_err.max() / max(s, 1e-300), away * tol)


def analytic_vertex_gamma(problem, st):
    """gamma of a particle ON a planar wall is 1/2, on an edge 1/4, in a corner 1/8 (the fraction of the kernel support
    inside the tank); rim vertices of the open top are treated like face/edge vertices of an infinitely tall wall"""
    g = problem.global_pos(st["pos"], st["hash"])
    dp = problem.m_deltap
    gam = np.ones(len(g))
    for a, L in enumerate((problem.l, problem.w)):
        gam *= np.where((np.abs(g[:, a]) < 0.25 * dp) | (np.abs(g[:, a] - L) < 0.25 * dp), 0.5, 1.0)
    gam *= np.where(np.abs(g[:, 2]) < 0.25 * dp, 0.5, 1.0)
    return gam.astype(np.float32)


def sa_oracle_state(problem=None, **kw):
    """sorted arrays + neighbour list of an SABox, all by the oracle"""
    problem = problem or SABox(**kw)
    n = problem.num_particles
    o = Oracle(orc_params_from(problem.sphx_params(n), problem))
    a = problem.copy_to_array()
    hash_, info = a["hash"].copy(), a["info"].copy()
    pidx = o.fix_hash(hash_, info)
    o.sort(hash_, info, pidx)
    cs, ce, _, pos, vel, newn = o.reorder(a["pos"], a["vel"], info, hash_, pidx, problem.grid_cells, segments=False)
    assert newn == n
    st = dict(problem=problem, oracle=o, n=n, pos=pos, vel=vel, info=info, hash=hash_, cs=cs, ce=ce, pidx=pidx,
              vertices=a["vertices"][pidx], boundelements=a["boundelements"][pidx], gradgamma=a["gradgamma"][pidx])
    sp = problem.simparams
    st["sqinfl"] = float(np.float32(sp.nlSqInfluenceRadius))
    # GPUWorker.cc:1890: (sqrt(nlSqInfluenceRadius) + slength/sfactor/2)^2, in float
    f32 = np.float32
    st["bound_sqinfl"] = float(np.power(f32(np.sqrt(f32(sp.nlSqInfluenceRadius))) + f32(sp.slength) / f32(sp.sfactor) / f32(2.0), f32(2.0), dtype=np.float32))
    st["nl"], st["vertpos"], st["neibs_info"] = o.build_neibs_sa(pos, info, st["vertices"], st["boundelements"], hash_, cs, ce, n, n,
                                                                 st["sqinfl"], st["bound_sqinfl"])
    return st


def list_sections(st, i):
    """(fluid, boundary, vertex) neighbour indices of sorted particle i, decoded from the oracle-format list"""
    p = st["problem"]
    sp = p.simparams
    stride = st["n"]
    nl = st["nl"].reshape(sp.neiblistsize, stride)
    g0 = p.grid_pos_from_hash(st["hash"][i:i + 1])[0]
    out = []
    for first, step in ((0, 1), (sp.neibboundpos, -1), (sp.neibboundpos + 1, 1)):
        res, slot, base = [], first, 0
        while True:
            d = int(nl[slot, i])
            if d == 0xFFFF:
                break
            if d >= D.CELLNUM_ENCODED:
                c = (d >> D.CELLNUM_SHIFT) - 1
                off = np.array([c % 3 - 1, (c // 3) % 3 - 1, c // 9 - 1])
                base = int(st["cs"][int(p.calc_grid_hash((g0 + off)[None, :])[0])])
                d &= D.NEIBINDEX_MASK
            res.append(base + d)
            slot += step
        out.append(res)
    return out


class OracleSaSim:
    """The predictor-corrector sequence of an SA_BOUNDARY run without density summation and with gamma by quadrature
    (PredictorCorrectorIntegrator.cc: initializeBoundaryConditionsSequence<SA_BOUNDARY> :117-290, the step phases :386-685),
    executed by the CPU oracle.  Test infrastructure (the GPU tests compare MultiGpuEngine.step against it)."""

    def __init__(self, problem, repack=False):
        st = sa_oracle_state(problem)
        self.st, self.problem, self.o, self.n = st, problem, st["oracle"], st["n"]
        o, n, p = self.o, self.n, problem
        self.pos, self.vel, self.info, self.hash, self.cs, self.nl = st["pos"], st["vel"], st["info"], st["hash"], st["cs"], st["nl"]
        self.vertices, self.vertpos = st["vertices"], st["vertpos"]
        self.be = o.sa_compute_vertex_normal(st["boundelements"], self.vertices, self.info, self.hash, self.cs, self.nl, n)
        gg = o.sa_init_gamma(st["gradgamma"], self.pos, self.be, self.vertpos, self.info, self.hash, self.cs, self.nl, n, p.m_deltap)
        self.keps = p.simparams.turbmodel == D.KEPSILON and not repack
        if self.keps:        # ProblemCore::init_keps / init_turbvisc: uniform k, epsilon, eddy viscosity; no Eulerian velocity
            k0, e0, nut0 = p.init_keps()
            N = len(self.pos)
            self.ke = dict(tke=np.full(N, k0, np.float32), eps=np.full(N, e0, np.float32), turbvisc=np.full(N, nut0, np.float32),
                           eulervel=np.zeros((N, 4), np.float32))
            self.vel, self.gg, self.ke = o.sa_bc_keps(self.pos, self.vel, gg, self.ke, self.vertices, self.be, self.info, self.hash,
                                                      self.cs, self.nl, n, 0, p.m_deltap)
        else:
            self.vel, self.gg = o.sa_segment_bc(self.pos, self.vel, gg, self.vertices, self.be, self.info, self.hash, self.cs, self.nl, n,
                                                step=0, repack=repack)
            self.vel = o.sa_vertex_bc(self.pos, self.vel, self.gg, self.info, self.hash, self.cs, self.nl, n)
        self.dt = float(np.float32(p.simparams.dt))
        self.t = 0.0
        self.iterations = 0
        pp, sp = p.physparams, p.simparams
        self.sspeed_cfl = float(np.float32(np.float64(np.float32(max(pp.sscoeff))) * 1.1))
        self.max_kinvisc = float(np.float32(max(pp.kinematicvisc))) if sp.rheologytype == D.NEWTONIAN else 0.0

    def _bc(self, pos, vel, gg, step):
        o, n = self.o, self.n
        vel, gg = o.sa_segment_bc(pos, vel, gg, self.vertices, self.be, self.info, self.hash, self.cs, self.nl, n, step=step)
        return o.sa_vertex_bc(pos, vel, gg, self.info, self.hash, self.cs, self.nl, n), gg

    def _dt(self, cfl, nb):
        dt = self.o.dtreduce(cfl, nb, self.sspeed_cfl, self.max_kinvisc)
        sf = self.problem.simparams.simflags
        if not (sf & D.ENABLE_GAMMA_QUADRATURE):          # dynamic gamma: its own CFL condition (src/cuda/forces.cu:576-585)
            dt = min(dt, float(self.o.L.orc_sa_gamma_dt(np.float32(dt), np.float32(self.o.max_gamma_cfl))))
        return dt

    def _post_euler(self, ps, vs, hdt):
        """what follows EULER on the new state (PredictorCorrectorIntegrator.cc:607-684): density summation (+ Brezzi diffusion)
        or, with the continuity equation, the gamma integration; always from the gamma / density of step n"""
        o, n = self.o, self.n
        sp = self.problem.simparams
        if sp.simflags & D.ENABLE_DENSITY_SUM:
            vs, gs = o.sa_density_sum(vs, self.pos, ps, self.vel, self.gg, self.be, self.vertpos, self.info, self.hash, self.cs, self.nl, n)
            if sp.densitydiffusiontype == D.BREZZI:
                vs, _ = o.sa_density_diffusion(ps, vs, gs, self.info, self.hash, self.cs, self.nl, n, hdt)
        else:
            gs = o.sa_integrate_gamma(self.gg, ps, self.be, self.vertpos, self.info, self.hash, self.cs, self.nl, n)
        return vs, gs

    def _dt_keps(self, cfl, nb):
        """dtreduce with the viscous limit of the largest eddy viscosity (src/cuda/forces.cu:585-598)"""
        f = np.float32
        dt = self._dt(cfl, nb)
        h = f(self.o.p.slength)
        dt_visc = f(f(h * h) / f(f(self.max_kinvisc) + f(self.o.cfl_keps[:nb].max()))) * f(0.125)
        return float(min(f(dt), dt_visc))

    def step_keps(self):
        """the step with turbulence<KEPSILON>: k, epsilon, the eddy viscosity and the Eulerian velocity are part of the state, the
        forces passes write DKDE, Euler integrates it, the boundary conditions treat the wall rows"""
        o, n, p = self.o, self.n, self.problem
        dp = p.m_deltap
        dt = float(np.float32(self.dt))
        hdt = float(np.float32(dt) / np.float32(2))
        A = (self.info, self.hash, self.cs, self.nl)
        f1, cfl, nb, dk1, _ = o.forces_sa_keps(self.pos, self.vel, *A, self.gg, self.be, self.vertpos, self.ke, n, dp)
        dt1 = self._dt_keps(cfl, nb)
        ps, vs = o.euler(self.pos, self.vel, self.info, self.hash, f1, n, hdt, 1)
        ks = o.euler_keps(self.ke, dk1, f1, self.pos, self.info, n, hdt)
        vs, gs = self._post_euler(ps, vs, hdt)
        vs, gs, ks = o.sa_bc_keps(ps, vs, gs, ks, self.vertices, self.be, *A, n, 1, dp)
        f2, cfl, nb, dk2, _ = o.forces_sa_keps(ps, vs, *A, gs, self.be, self.vertpos, ks, n, dp)
        dt2 = self._dt_keps(cfl, nb)
        pn, vn = o.euler(self.pos, self.vel, self.info, self.hash, f2, n, dt, 2)
        kn = o.euler_keps(self.ke, dk2, f2, self.pos, self.info, n, dt)
        vn, gn = self._post_euler(pn, vn, dt)
        vn, gn, kn = o.sa_bc_keps(pn, vn, gn, kn, self.vertices, self.be, *A, n, 2, dp)
        self.forces, self.dkde = f2, dk2
        self.pos, self.vel, self.gg, self.ke = pn, vn, gn, kn
        self.t += dt
        self.dt = min(dt1, dt2)
        self.iterations += 1

    def repack_step(self):
        """one iteration of the repacking integrator with SA_BOUNDARY (RepackingIntegrator.cc:278-420): forces(REPACK), one full-dt
        Euler step of the fluid, INTEGRATE_GAMMA of the new positions; no neighbour rebuild here"""
        o, n, p = self.o, self.n, self.problem
        dt = float(np.float32(self.dt))
        f, cfl, nb = o.repack_forces_sa(self.pos, self.vel, self.info, self.hash, self.cs, self.nl, self.gg, self.be, self.vertpos, n, p.m_deltap)
        dt1 = o.dtreduce(cfl, nb, self.sspeed_cfl, self.max_kinvisc)
        ps, vs = o.euler_repack(self.pos, self.vel, self.info, self.hash, f, n, dt, 1)
        self.gg = o.sa_integrate_gamma(self.gg, ps, self.be, self.vertpos, self.info, self.hash, self.cs, self.nl, n)
        self.pos, self.vel, self.forces = ps, vs, f
        self.t += dt
        self.iterations += 1
        self.dt = dt1

    # ---- SA bodies with prescribed motion (ENABLE_MOVING_BODIES): the command sequence of PredictorCorrectorIntegrator.cc:386-685
    # with BUFFER_BOUNDELEMENTS as a state buffer (:408-418), restated independently of gpusph_amd.multigpu
    def _move_bodies(self, step, dt):
        m = self.bodies.timestep(step, dt, self.t)
        p = self.o.p
        for b in range(len(self.bodies)):
            for a in range(3):
                p.rbtrans[b][a] = float(m["trans"][b][a]); p.rblinearvel[b][a] = float(m["lvel"][b][a])
                p.rbangularvel[b][a] = float(m["avel"][b][a])
            for a in range(9):
                p.rbsteprot[b][a] = float(m["rot"][b][a])
        self._last_motion = m

    def _post_euler_moving(self, ps, vs, be_new, hdt):
        o, n = self.o, self.n
        sp = self.problem.simparams
        if sp.simflags & D.ENABLE_DENSITY_SUM:
            vs, gs = o.sa_density_sum_moving(vs, self.pos, ps, self.vel, self.gg, self.gg, self.be, be_new, self.vertpos, self.info,
                                             self.hash, self.cs, self.nl, n)
            if sp.densitydiffusiontype == D.BREZZI:
                vs, _ = o.sa_density_diffusion(ps, vs, gs, self.info, self.hash, self.cs, self.nl, n, hdt)
        else:
            gs = o.sa_integrate_gamma_moving(self.gg, ps, be_new, self.vertpos, self.info, self.hash, self.cs, self.nl, n)
        return vs, gs

    def _bc_moving(self, pos, vel, gg, be, step):
        o, n = self.o, self.n
        vel, gg = o.sa_segment_bc(pos, vel, gg, self.vertices, be, self.info, self.hash, self.cs, self.nl, n, step=step)
        return o.sa_vertex_bc(pos, vel, gg, self.info, self.hash, self.cs, self.nl, n), gg

    def step_moving(self):
        from gpusph_amd.bodies import MovingBodies
        o, n, p = self.o, self.n, self.problem
        if getattr(self, "bodies", None) is None:
            self.bodies = MovingBodies(p, p.rb_cg_global)
            for b in range(len(self.bodies)):      # the centre of rotation the Euler step reads (uploaded at start and after every step)
                for a in range(3):
                    o.p.rbcgGridPosE[b][a] = int(p.rb_cg_gridpos[b][a]); o.p.rbcgPosE[b][a] = float(p.rb_cg_pos[b][a])
        dp = p.m_deltap
        dt = float(np.float32(self.dt))
        hdt = float(np.float32(dt)/np.float32(2))
        A = (self.info, self.hash, self.cs, self.nl)
        f1, cfl, nb = o.forces_sa(self.pos, self.vel, *A, self.gg, self.be, self.vertpos, n, dp)
        dt1 = self._dt(cfl, nb)
        self._move_bodies(1, dt)
        ps, vs = o.euler(self.pos, self.vel, self.info, self.hash, f1, n, hdt, 1)
        bes = o.sa_update_normals(self.be, self.info, n)
        vs, gs = self._post_euler_moving(ps, vs, bes, hdt)
        vs, gs = self._bc_moving(ps, vs, gs, bes, 1)
        f2, cfl, nb = o.forces_sa(ps, vs, *A, gs, bes, self.vertpos, n, dp)
        dt2 = self._dt(cfl, nb)
        self._move_bodies(2, dt)
        pn, vn = o.euler(self.pos, self.vel, self.info, self.hash, f2, n, dt, 2)
        ben = o.sa_update_normals(self.be, self.info, n)
        vn, gn = self._post_euler_moving(pn, vn, ben, dt)
        vn, gn = self._bc_moving(pn, vn, gn, ben, 2)
        m = self._last_motion
        for b in range(len(self.bodies)):
            for a in range(3):
                o.p.rbcgGridPosE[b][a] = int(m["cg_grid"][b][a]); o.p.rbcgPosE[b][a] = float(m["cg_pos"][b][a])
        self.forces = f2
        self.pos, self.vel, self.gg, self.be = pn, vn, gn, ben
        self.t += dt
        self.dt = min(dt1, dt2)
        self.iterations += 1

    def step(self):
        """no neighbour rebuild here: the runs compared are shorter than buildneibsfreq"""
        if self.keps:
            return self.step_keps()
        if self.problem.simparams.simflags & D.ENABLE_MOVING_BODIES:
            return self.step_moving()
        o, n, p = self.o, self.n, self.problem
        dp = p.m_deltap
        dt = float(np.float32(self.dt))
        hdt = float(np.float32(dt) / np.float32(2))
        f1, cfl, nb = o.forces_sa(self.pos, self.vel, self.info, self.hash, self.cs, self.nl, self.gg, self.be, self.vertpos, n, dp)
        dt1 = self._dt(cfl, nb)
        ps, vs = o.euler(self.pos, self.vel, self.info, self.hash, f1, n, hdt, 1)
        vs, gs = self._post_euler(ps, vs, hdt)
        vs, gs = self._bc(ps, vs, gs, 1)
        f2, cfl, nb = o.forces_sa(ps, vs, self.info, self.hash, self.cs, self.nl, gs, self.be, self.vertpos, n, dp)
        dt2 = self._dt(cfl, nb)
        pn, vn = o.euler(self.pos, self.vel, self.info, self.hash, f2, n, dt, 2)
        vn, gn = self._post_euler(pn, vn, dt)
        vn, gn = self._bc(pn, vn, gn, 2)
        self.forces = f2
        self.pos, self.vel, self.gg = pn, vn, gn
        self.t += dt
        self.dt = min(dt1, dt2)
        self.iterations += 1


class OracleSaIoSim:
    """An open channel on the CPU oracle: an SABox (density summation form) whose x = 0 wall is a velocity inlet (u_E = U ex) and
    whose x = l wall is a pressure outlet (hydrostatic for the still water level), the command sequence of
    PredictorCorrectorIntegrator.cc with ENABLE_INLET_OUTLET (:127-300 the boundary-condition phases, :386-685 the step), a
    neighbour-list rebuild before every step as the reference's ChannelIO does (src/problems/ChannelIO.cu:63) -- the particles an
    inlet vertex releases exist from the next rebuild on -- and the imposed values of that problem's callback (:104-131).
    GROUNDWORK: the checker of engines that are not built yet; nothing in the product mirrors it."""

    def __init__(self, problem, U, room=1.6, brezzi=False, water_depth=False):
        """brezzi: the Brezzi diffusion after every density summation (its open-boundary term on the pressure outlet);
        water_depth: ENABLE_WATER_DEPTH -- the outlet's hydrostatic pressure follows the level the vertex pass of the forces
        measured, as ChannelIO's callback does (:124-127), instead of the still water level"""
        self.problem, self.U = problem, float(U)
        self.brezzi, self.water_depth = brezzi, water_depth
        self.depth = np.zeros(3, dtype=np.uint32)             # IOwaterdepth[numOpenBoundaries + 1]: objects 1 (inlet) and 2 (outlet)
        self.level_seen = []
        p = problem
        a = p.copy_to_array()
        n0 = p.num_particles
        self.cap = cap = int(n0 * room)
        self.o = o = Oracle(orc_params_from(p.sphx_params(cap), p))
        # open boundaries: flags and object numbers on the unsorted arrays (global positions known)
        g = p.global_pos(a["pos"], a["hash"])
        t = info_type(a["info"])
        wall = (t == D.PT_BOUNDARY) | (t == D.PT_VERTEX)
        nrm = a["boundelements"]
        inlet = wall & (np.abs(g[:, 0]) < 1e-6) & ((t == D.PT_VERTEX) | (nrm[:, 0] > 0.5))
        outlet = wall & (np.abs(g[:, 0] - p.l) < 1e-6) & ((t == D.PT_VERTEX) | (nrm[:, 0] < -0.5))
        info = a["info"].copy()
        # a problem that is an open channel itself (gpusph_amd.problem.SAChannelIO) comes flagged, numbers its open boundaries
        # from 0 and imposes its own values (the engine's driver is held against this sim with such a problem)
        self.own_problem = hasattr(problem, "impose_open_boundaries")
        if self.own_problem:
            assert ((info[inlet, 0] & D.FG_INLET) != 0).all() and ((info[outlet, 0] & D.FG_OUTLET) != 0).all()
            self.outlet_obj = 1
            self.depth = np.zeros(problem.num_open_boundaries, dtype=np.uint32)
        else:
            self.outlet_obj = 2
            info[inlet, 0] |= D.FG_INLET | D.FG_VELOCITY_DRIVEN
            info[inlet, 1] = (info[inlet, 1] & 0xF000) | 1
            info[outlet, 0] |= D.FG_OUTLET
            info[outlet, 1] = (info[outlet, 1] & 0xF000) | 2
        self.num_open_vertices = int(((inlet | outlet) & (t == D.PT_VERTEX)).sum())

        def pad(x, fill):
            out = np.full((cap,) + x.shape[1:], fill, dtype=x.dtype)
            out[:n0] = x
            return out
        self.pos = pad(a["pos"], np.nan)                       # unused rows: inactive particles, sorted behind the active ones
        vel = a["vel"].copy()
        vel[t == D.PT_FLUID, 0] = np.float32(self.U)           # the stream is there from the start
        self.vel = pad(vel, 0)
        self.info = pad(info, 0)
        self.hash = pad(a["hash"], D.CELL_HASH_MAX)
        self.vertices = pad(a["vertices"], 0)
        self.be = pad(a["boundelements"], np.nan)
        self.gg = pad(a["gradgamma"], np.nan)
        self.ev = np.zeros((cap, 4), dtype=np.float32)
        ids = info_id(info)
        nid = np.full(n0, 0xFFFFFFFF, dtype=np.uint32)
        openv = (inlet | outlet) & (t == D.PT_VERTEX)
        nid[openv] = int(ids.max()) + 1 + np.arange(int(openv.sum()), dtype=np.uint32)      # GPUSPH hands out ids past the last one
        self.next_ids = pad(nid, 0xFFFFFFFF)
        self.n = n0
        self.t, self.iterations = 0.0, 0
        self.dt = float(np.float32(p.simparams.dt))
        pp, sp = p.physparams, p.simparams
        self.sspeed_cfl = float(np.float32(np.float64(np.float32(max(pp.sscoeff))) * 1.1))
        self.max_kinvisc = float(np.float32(max(pp.kinematicvisc))) if sp.rheologytype == D.NEWTONIAN else 0.0
        f32 = np.float32
        self.sqinfl = float(f32(sp.nlSqInfluenceRadius))
        self.bound_sqinfl = float(np.power(f32(np.sqrt(f32(sp.nlSqInfluenceRadius))) + f32(sp.slength) / f32(sp.sfactor) / f32(2.0), f32(2.0), dtype=f32))
        self.created = self.removed = 0
        # --- initialisation (initializeBoundaryConditionsSequence<SA_BOUNDARY>, init_step)
        self._rebuild()
        o, n = self.o, self.n
        A = (self.hash, self.cs, self.nl)
        self.be = o.sa_compute_vertex_normal(self.be, self.vertices, self.info, *A, n)
        self.gg = o.sa_init_gamma(self.gg, self.pos, self.be, self.vertpos, self.info, *A, n, p.m_deltap)
        self.info = o.sa_identify_corner_vertices(self.pos, self.info, self.hash, self.vertices, self.cs, self.nl, n)
        _, self.pos = o.sa_init_io_mass(self.pos, self.info, self.hash, self.vertices, self.cs, self.nl, n, p.m_deltap)
        self.vel, self.ev = self._impose(self.pos, self.vel, self.ev)
        self.vel, self.gg, self.ev = o.sa_segment_bc_io(self.pos, self.vel, self.gg, self.ev, self.vertices, self.be, self.info, *A, n, 0)
        a = self._vertex_bc(self.pos, self.vel, self.gg, self.ev, self.vertices, 0.0, 0)      # in place in the reference: kept
        self.pos, self.vel, self.ev = a["new_pos"], a["vel"], a["euler_vel"]

    # ChannelIO_imposeBoundaryCondition (src/problems/ChannelIO.cu:104-131): the Lagrangian velocity of the open boundaries'
    # particles is cleared, a velocity boundary gets u_E = U ex, a pressure boundary the density of the hydrostatic pressure
    def _impose(self, pos, vel, ev):
        p = self.problem
        n = self.n
        if self.own_problem:
            import torch
            vel, ev = vel.copy(), ev.copy()
            depth = torch.from_numpy(self.depth.view(np.int32)) if self.water_depth else None
            if self.water_depth:
                self.level_seen.append(self.o.sa_io_water_depth_z(self.depth[self.outlet_obj]))
            p.impose_open_boundaries(torch.from_numpy(pos), torch.from_numpy(vel), torch.from_numpy(ev),
                                     torch.from_numpy(self.info.view(np.int16)), torch.from_numpy(self.hash.view(np.int32)), depth,
                                     self.t, n)
            return vel, ev
        io = (self.info[:n, 0] & (D.FG_INLET | D.FG_OUTLET)) != 0
        vdriven = (self.info[:n, 0] & D.FG_VELOCITY_DRIVEN) != 0
        vel, ev = vel.copy(), ev.copy()
        vel[:n][io] = 0.0
        ev[:n][io] = 0.0
        rows = np.where(io & vdriven)[0]
        ev[rows, 0] = np.float32(self.U)
        rows = np.where(io & ~vdriven)[0]
        z = p.global_pos(pos[:n], self.hash[:n])[rows, 2]
        level = np.float32(p.water_level)
        if self.water_depth:
            # <Problem>_imposeBoundaryConditionDevice (problems/CompleteSaExample.cu:266-272): the scaled maximum back to a height;
            # imposeBoundaryConditionHost clears the array after the launch (:323-325)
            level = np.float32(self.o.sa_io_water_depth_z(self.depth[self.outlet_obj]))
            self.level_seen.append(float(level))
            self.depth[:] = 0
        pres = np.float32(9.81) * np.maximum(level - z.astype(np.float32), np.float32(0)) * np.float32(p.physparams.rho0[0])
        ev[rows, 3] = [self.o.eos_RHO(float(x)) for x in pres]
        return vel, ev

    def _forces(self, pos, vel, ev, gg):
        """the forces of one half step; with ENABLE_WATER_DEPTH the vertex pass that follows the fluid's (vertex_forces,
        src/cuda/forces.cu:676-686) leaves the water depth of the pressure-driven open boundaries"""
        o, n = self.o, self.n
        out = o.forces_sa_io(pos, vel, ev, self.info, self.hash, self.cs, self.nl, gg, self.be, self.vertpos, n, self.problem.m_deltap)
        if self.water_depth:
            o.sa_io_water_depth(self.depth, pos, self.info, self.hash, self.cs, self.nl, n)
        return out

    def _density(self, vs, ps, dt):
        """density summation (+ the Brezzi diffusion with the open boundaries' term) on the moved particles"""
        o, n = self.o, self.n
        A = (self.hash, self.cs, self.nl)
        vs, gs, _ = o.sa_density_sum_io(vs, self.pos, ps, self.vel, self.ev, self.gg, self.be, self.vertpos, self.info, *A, n, dt)
        if self.brezzi:
            vs, _ = o.sa_density_diffusion_io(ps, vs, gs, self.info, *A, self.be, self.vertpos, n, dt, self.problem.m_deltap)
        return vs, gs

    def _rebuild(self):
        o = self.o
        self.pos[self.n:] = np.nan          # rows behind the particles in use: inactive (Euler and the reorder leave them cleared)
        pidx = o.fix_hash(self.hash, self.info) if self.iterations == 0 else o.calc_hash(self.pos, self.hash, self.info)
        o.sort(self.hash, self.info, pidx)
        self.cs, self.ce, _, self.pos, self.vel, self.n = o.reorder(self.pos, self.vel, self.info, self.hash, pidx,
                                                                    self.problem.grid_cells, segments=False)
        for name in ("vertices", "be", "gg", "ev", "next_ids"):
            setattr(self, name, np.ascontiguousarray(getattr(self, name)[pidx]))
        n = self.n
        self.nl, self.vertpos, self.neibs_info = o.build_neibs_sa(self.pos, self.info, self.vertices, self.be, self.hash, self.cs,
                                                                  self.ce, n, n, self.sqinfl, self.bound_sqinfl)
        assert self.neibs_info.hasTooManyNeibs == -1

    def _vertex_bc(self, pos, vel, gg, ev, vertices, dt, step):
        """SA_CALC_VERTEX_BOUNDARY_CONDITIONS on the given state; the arrays it may extend (last step) become the sim's"""
        a = self.o.sa_vertex_bc_io(pos, vel, gg, ev, vertices, self.be, self.vertpos, self.info, self.hash, self.next_ids, self.cs,
                                   self.nl, self.n, self.problem.m_deltap, dt, step, self.num_open_vertices, room=0)
        if step == 2:
            self.created += a["n"] - self.n
            self.info, self.hash, self.next_ids, self.be = a["info"], a["hash"], a["next_ids"], a["boundelements"]
        return a

    def _dtmin(self, cfl, nb):
        dt = self.o.dtreduce(cfl, nb, self.sspeed_cfl, self.max_kinvisc)
        return min(dt, float(self.o.L.orc_sa_gamma_dt(np.float32(dt), np.float32(self.o.max_gamma_cfl))))

    def step(self):
        o, p = self.o, self.problem
        dp = p.m_deltap
        if self.iterations > 0:
            self._rebuild()
        n = self.n
        A = (self.hash, self.cs, self.nl)
        dt = float(np.float32(self.dt)); hdt = float(np.float32(dt) / np.float32(2))
        infl = float(p.simparams.influenceRadius)
        # predictor
        f1, cfl, nb = self._forces(self.pos, self.vel, self.ev, self.gg)
        dt1 = self._dtmin(cfl, nb)
        ps, vs = o.euler(self.pos, self.vel, self.info, self.hash, f1, n, hdt, 1)
        vs, gs = self._density(vs, ps, hdt)
        vs, evs = self._impose(ps, vs, self.ev)
        vs, gs, evs = o.sa_segment_bc_io(ps, vs, gs, evs, self.vertices, self.be, self.info, *A, n, 1)
        a = self._vertex_bc(ps, vs, gs, evs, self.vertices, hdt, 1)
        ps, vs, evs = a["new_pos"], a["vel"], a["euler_vel"]
        # corrector
        f2, cfl, nb = self._forces(ps, vs, evs, gs)
        dt2 = self._dtmin(cfl, nb)
        pn, vn = o.euler(self.pos, self.vel, self.info, self.hash, f2, n, dt, 2)
        vn, gn = self._density(vn, pn, dt)
        vn, evn = self._impose(pn, vn, self.ev)
        vn, gn, evn = o.sa_segment_bc_io(pn, vn, gn, evn, self.vertices, self.be, self.info, *A, n, 2)
        vert2, gn = o.find_outgoing_segment(pn, vn, self.vertices, gn, self.vertpos, self.be, self.info, *A, n, infl)
        a = self._vertex_bc(pn, vn, gn, evn, vert2, dt, 2)
        n2 = a["n"]
        pn, vert3 = o.disable_outgoing_parts(a["new_pos"], a["vertices"], self.info, n2)
        self.removed += int(np.isnan(pn[:n2, 3]).sum() - np.isnan(a["new_pos"][:n2, 3]).sum())
        self.pos, self.vel, self.gg, self.ev, self.vertices = pn, a["vel"], a["ggam"], a["euler_vel"], vert3
        self.forces = f2
        self.n = n2
        self.t += dt
        self.dt = min(dt1, dt2)
        self.iterations += 1


# ---- per-entry float64 arbitration of the boundary-element terms ---------------------------------------------------------------
def _grad_gamma_float64(ns, qvb, q, h):
    """(|grad gamma_as| by the closed form of gamma.cuh:248-370 in double precision, the sum of the MAGNITUDES of the terms it
    adds up).  The second number is what makes the closed form ill-conditioned: edge antiderivatives (a polynomial of monomials of
    both signs, an angle difference, a logarithm difference) that cancel against each other and against the angle bookkeeping; a
    float32 evaluation, in whatever order, carries a rounding error of a few 2^-24 of THAT, not of the result."""
    import math
    ns = np.asarray(ns, dtype=np.float64); q = np.asarray(q, dtype=np.float64); qv = np.asarray(qvb, dtype=np.float64).reshape(3, 3)
    pas = float(ns @ q); a = abs(pas)
    if a >= 2:
        return 0.0, 0.0
    g = mag = tot = inside = tot_mag = 0.0
    for e in range(3):
        v0, v1 = qv[e], qv[(e + 1) % 3]
        t = (v0 - v1)/np.linalg.norm(v0 - v1)
        m = np.cross(ns, t); m /= np.linalg.norm(m)
        b = float(m @ (q - v0)); c = math.hypot(pas, b)
        s0, s1 = float(-(q - v0) @ t), float(-(q - v1) @ t)
        ang = math.copysign(math.atan2(s1, abs(b)) - math.atan2(s0, abs(b)), b)
        tot += ang; tot_mag += abs(math.atan2(s1, abs(b))) + abs(math.atan2(s0, abs(b)))
        if c < 2:
            lim = math.sqrt(4 - c*c)
            s0 = math.copysign(min(abs(s0), lim), s0); s1 = math.copysign(min(abs(s1), lim), s1)
            d0 = min(math.hypot(c, s0), 2.0); d1 = min(math.hypot(c, s1), 2.0)

            def pol(s, d):
                s2 = s*s
                terms = [3*a**4*(-420), 3*a**4*29*d, b**4*(-420), b**4*33*d, 2*a*a*(-210*8), 2*a*a*(-210*s2), 2*a*a*756*d, 2*a*a*19*s2*d,
                         4*336, 4*s2*s2*(-21), 4*s2*s2*2*d, 4*s2*28*(-5), 4*s2*28*3*d,
                         2*b*b*420*(-2), 2*b*b*420*d, 2*b*b*6*a*a*(-105), 2*b*b*6*a*a*8*d, 2*b*b*s2*(-140), 2*b*b*s2*13*d]
                return s*sum(terms), abs(s)*sum(abs(x) for x in terms)

            def angle(s, d):
                return math.atan2(a*s, b*d) - math.atan2(s, b), abs(math.atan2(a*s, b*d)) + abs(math.atan2(s, b))

            def lg(s, d):
                v = math.copysign(1, s)*math.acosh(max(d/max(c, 1e-7), 1.0))
                return v, abs(v)
            K = 5*b**6 + 21*b**4*(8 + a*a) + 35*b*b*a*a*(16 + a*a) + 35*a**4*(24 + a*a)
            (p1, mp1), (p0, mp0) = pol(s1, d1), pol(s0, d0)
            (a1, ma1), (a0, ma0) = angle(s1, d1), angle(s0, d0)
            (l1, ml1), (l0, ml0) = lg(s1, d1), lg(s0, d0)
            ca = 48*a**5*(28 + a*a)
            g += 0.00015542474911*(ca*(a1 - a0) + b*(p1 - p0 + 3*K*(l1 - l0)))
            mag += 0.00015542474911*(ca*(ma1 + ma0) + abs(b)*(mp1 + mp0 + 3*K*(ml1 + ml0)))
            inside += math.copysign(math.atan2(s1, abs(b)) - math.atan2(s0, abs(b)), b)
    tt = 1 - a/2
    w = 0.05968310365947*tt**5*(2 + 5*a + 4*a*a)
    g += (inside - tot)*w
    mag += 2*tot_mag*w
    return g/h, mag/h


def _boundary_section(problem, nl2, cs, hash_, pos, i):
    """the boundary section of particle i's list, decoded: [(neighbour index, own position as seen from the neighbour's cell in
    float32, as the walkers form it)]"""
    sp = problem.simparams
    g = problem.grid_pos_from_hash(hash_[i:i + 1])[0]
    csz = problem.m_cellsize.astype(np.float32)
    out, slot, base, shift = [], int(sp.neibboundpos), None, None
    while slot >= 0:
        d = int(nl2[slot, i])
        if d == D.NEIBS_END:
            break
        if d >= D.CELLNUM_ENCODED:
            c = (d >> D.CELLNUM_SHIFT) - 1
            off = np.array([c % 3 - 1, (c % 9)//3 - 1, c//9 - 1])
            base = int(cs[int(problem.calc_grid_hash((g + off)[None])[0])])
            shift = off
        # fmaf(-(float)off, cs, pos): one rounding
        own = np.array([np.float32(np.float64(pos[i, a]) - np.float64(shift[a])*np.float64(csz[a])) for a in range(3)], dtype=np.float32)
        out.append((base + (d & D.NEIBINDEX_MASK), own))
        slot -= 1
    return out


ROUNDINGS = 8.0


def assert_wall_rows_no_farther_from_float64(sim, got, want, rows, tol, scale_xyz, scale_w, what=""):
    """`got` (the product) and `want` (the oracle) are the SA forces of the laminar option set on the same state; every row of
    `rows` (fluid particles) must hold |got - want| <= tol * scale, or -- where the row has boundary elements in reach --
    |got - want| <= tol * scale + 2 |want - f64|, f64 being the value the SAME sums take with |grad gamma_as| of every element in
    reach evaluated in float64 (the closed form of gamma.cuh in double, tests/test_sa_wall_gamma.py): a row is granted what
    float32 demonstrably does to ITS OWN boundary terms, measured on the oracle's evaluation of them, and nothing else (the bound
    of tests/test_sa_wall_gamma.py for one element, carried to the row's sum; accelerations as vectors).  Over the rows so
    arbitrated the product must on average be as close to the float64 values as the oracle is (within a quarter).

    What is asserted per flagged row: BOTH evaluations lie within tol * scale + ROUNDINGS 2^-24 sum_s M_s |K_s| / gamma of f64, where
    M_s is the sum of the magnitudes of the terms the closed form of |grad gamma_as| adds up (the cancelling edge antiderivatives;
    _grad_gamma_float64) -- the room rounding has in ANY float32 evaluation of that element, whatever its operation order -- and
    ROUNDINGS = 8 roundings of relative size 2^-24 per term (the terms are products and transcendental functions of rounded
    inputs).  The oracle is held to the same room: if it were outside, the arbitration, not the product, would be wrong.

    Why this is the right question: the force of a particle next to a wall is (S + sum_s g_s K_s) / gamma with g_s = |grad gamma_as|
    the only ill-conditioned input (two float evaluations of its closed form in different operation orders differ by up to ~1e-3
    of it for some positions; the reference's own float value is that far from the float64 value).  The oracle repeats the
    reference's operation order, the kernels have their own: where they differ by more than rounding, the float64 value of the
    formula says which is closer -- no fraction of outliers and no multiple of the tolerance is granted any more.  K_s is formed
    here in float64 from the state: pressure (P_a/rho_a^2 + P_s/rho_s^2) rho_s n_s, the laminar wall term -2 mu_avg/(r_as rho_a)
    (v_as - (v_as.n_s) n_s), continuity -rho_a (v_as.n_s) (forces_kernel.def:2079-2090,2414-2427,2680-2718)."""
    import ctypes as C
    import oracle_lib as ol
    got = np.asarray(got, dtype=np.float64); want = np.asarray(want, dtype=np.float64)
    p, problem, n = sim.o.p, sim.problem, sim.n
    sp = problem.simparams
    h = float(p.slength)
    nl2 = np.asarray(sim.nl).view(np.uint16).reshape(sp.neiblistsize, -1)
    L = ol.lib()
    L.orc_grad_gamma_vp.restype = C.c_float
    L.orc_grad_gamma_vp.argtypes = [C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    scale = np.array([scale_xyz]*3 + [scale_w])
    err = np.abs(got - want)
    flagged = [i for i in rows if (err[i] > tol*scale).any()]
    fluid_of = lambda j: int(sim.info[j, 1]) >> 12
    density_sum = bool(sp.simflags & D.ENABLE_DENSITY_SUM)
    newtonian = sp.rheologytype != D.INVISCID
    arbitrated = 0
    dist_own, dist_orc, rooms, aparts = [], [], [], []
    for i in flagged:
        sect = _boundary_section(problem, nl2, sim.cs, sim.hash, sim.pos, i)
        assert sect, "%s: row %d is %s of the scale off and has no boundary element in reach" % (what, i, err[i]/scale)
        fl = fluid_of(i)
        ri = (float(sim.vel[i, 3]) + 1.0)*float(p.rho0[fl])
        Pi = float(p.bcoeff[fl])*((float(sim.vel[i, 3]) + 1.0)**float(p.gammacoeff[fl]) - 1.0)
        gam = float(sim.gg[i, 3])
        delta = np.zeros(4)
        room = np.zeros(2)          # what float32 rounding of the cancelling terms of |grad gamma_as| can move this row by
        for s, own in sect:
            be = sim.be[s].astype(np.float64)
            ns = be[:3]
            rel32 = (own - sim.pos[s, :3]).astype(np.float32)
            r = float(np.sqrt((rel32.astype(np.float64)**2).sum()))
            if not np.isfinite(sim.pos[s, 3]) or r >= float(p.influenceradius) + problem.m_deltap:
                continue
            vp = [np.ascontiguousarray(sim.vertpos[k][s], dtype=np.float32) for k in range(3)]
            q32 = (rel32*np.float32(np.float32(1.0)/np.float32(h))).astype(np.float32)
            bel = np.ascontiguousarray(sim.be[s], dtype=np.float32)
            g_orc = float(L.orc_grad_gamma_vp(C.c_float(h), C.c_float(q32[0]), C.c_float(q32[1]), C.c_float(q32[2]),
                                               bel.ctypes.data_as(C.c_void_p), *[v.ctypes.data_as(C.c_void_p) for v in vp]))
            # corners of the element in units of h, from BUFFER_VERTPOS (the basis the neighbour-list build fixed)
            a = np.abs(ns); j = 1 if a[0] > a[1] else 0
            if (a[0] if j == 0 else a[1]) > a[2]:
                j = 2
            ej = np.zeros(3); ej[j] = 1.0
            u = np.cross(ns, ej); u /= np.linalg.norm(u); v = np.cross(ns, u)
            qvb = np.concatenate([-(u*float(vp[k][0]) + v*float(vp[k][1]))/h for k in range(3)])
            g_64, g_mag = _grad_gamma_float64(sim.be[s, :3], qvb, q32.astype(np.float64), h)
            fs = fluid_of(s)
            rs = (float(sim.vel[s, 3]) + 1.0)*float(p.rho0[fs])
            Ps = float(p.bcoeff[fs])*((float(sim.vel[s, 3]) + 1.0)**float(p.gammacoeff[fs]) - 1.0)
            vrel = sim.vel[i, :3].astype(np.float64) - sim.vel[s, :3].astype(np.float64)
            vn = float(vrel @ ns)
            K = (Pi/ri**2 + Ps/rs**2)*rs*ns
            if newtonian:
                r_as = max(abs(float(rel32.astype(np.float64) @ ns)), problem.m_deltap)
                kin = p.compvisc == D.KINEMATIC
                mu_i = float(p.visccoeff[fl])*(ri if kin else 1.0); mu_s = float(p.visccoeff[fs])*(rs if kin else 1.0)
                avg = 0.5*(mu_i + mu_s) if p.avgop == D.ARITHMETIC else (2*mu_i*mu_s/(mu_i + mu_s) if p.avgop == D.HARMONIC else np.sqrt(mu_i*mu_s))
                K = K - (2.0*avg/r_as)*(vrel - vn*ns)/ri
            Kw = 0.0 if density_sum else -ri*vn/float(p.rho0[fl])
            delta += (g_64 - g_orc)*np.concatenate([K, [Kw]])/gam
            room += ROUNDINGS*2.0**-24*g_mag*np.array([np.linalg.norm(K), abs(Kw)])/abs(gam)
        f64 = want[i] + delta
        vec = lambda d: np.array([np.linalg.norm(d[:3]), abs(d[3])])
        s2 = np.array([scale_xyz, scale_w])
        own, theirs, apart = vec(got[i] - f64), vec(want[i] - f64), vec(got[i] - want[i])
        assert (theirs <= tol*s2 + room).all(), "%s: row %d: the ORACLE is %s of the scale from the float64 value (conditioning allows %s): the arbitration itself is off" % (
            what, i, theirs/s2, room/s2)
        assert (own <= tol*s2 + room).all(), ("%s: row %d: the product is %s of the scale from the float64 value of its boundary terms; "
            "the conditioning of |grad gamma| of the elements in reach allows %s (the oracle is %s away)") % (what, i, own/s2, room/s2, theirs/s2)
        dist_own.append(own/s2); dist_orc.append(theirs/s2); rooms.append(room/s2); aparts.append(apart/s2)
        arbitrated += 1
    if arbitrated >= 8:
        mo, mr = np.mean(dist_own, axis=0), np.mean(dist_orc, axis=0)
        assert (mo <= 1.25*mr + tol).all(), "%s: over the arbitrated rows the product is %s of the scale from float64 on average, the oracle %s" % (what, mo, mr)
    if arbitrated:
        r, o_, a_ = np.array(rooms)[:, 0], np.array(dist_own)[:, 0], np.array(aparts)[:, 0]
        print("   arbitrated rows: product-oracle difference median %.2g / max %.2g of the scale; room granted median %.2g / max %.2g; "
              "share of the room the product uses: median %.2f, max %.2f" % (np.median(a_), a_.max(), np.median(r), r.max(),
                                                                         np.median(o_/(tol + r)), (o_/(tol + r)).max()))
    return arbitrated, len(rows)
