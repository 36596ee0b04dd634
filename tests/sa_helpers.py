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


def assert_close_but_for_gamma_spikes(got, want, tol, scale=None, frac=0.01, spike=200.0, what="", wall=None):
    """|got - want| <= tol * scale for all but a fraction `frac` of the entries, and <= spike * tol * scale for those.
    `wall` (one flag per row of got / want, see wall_rows): the allowance is for the rows that have a boundary element in
    reach ONLY -- every other row must hold tol * scale, so that an indexing or ordering bug of the particle <- particle
    sums cannot hide behind it.

    The closed form of |grad gamma_as| (edge antiderivatives that cancel against each other and against the angle
    bookkeeping) is ill-conditioned for some positions of a particle relative to an element: two float evaluations of the
    SAME formula in different operation orders differ there by up to ~1e-3 of |grad gamma|, and the reference's own float
    evaluation is that far from the float64 value of its formula (tests/test_sa_wall_gamma.py measures both against the
    reference's numbers).  The kernels integrate the same expressions in their own order of operations (set-up of an element
    once, polynomial collected by powers of the distance), so the few particles that have such an element in reach carry a
    difference the bulk does not; everything derived from gamma (SA forces, density summation, trajectories) inherits it."""
    got = np.asarray(got, dtype=np.float64); want = np.asarray(want, dtype=np.float64)
    s = float(np.abs(want).max() if scale is None else scale)
    err = np.abs(got - want)
    bad = float((err > tol * s).mean()) if err.size else 0.0
    worst = float(err.max()) if err.size else 0.0
    assert bad <= frac and worst <= spike * tol * s, "%s: %.3g of the entries beyond %.1e of the scale %.3g (allowed %.3g), worst %.3g of the scale (allowed %.3g)" % (
        what, bad, tol, s, frac, worst / max(s, 1e-300), spike * tol)
    if wall is not None:
        wall = np.asarray(wall, dtype=bool)
        assert len(wall) == len(err), "%s: one wall flag per row" % what
        away = err[~wall]
        assert away.size == 0 or away.max() <= tol * s, "%s: a row with no boundary element in reach is %.3g of the scale off (allowed %.1e)" % (
            what, away.max() / max(s, 1e-300), tol)


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

    def step(self):
        """no neighbour rebuild here: the runs compared are shorter than buildneibsfreq"""
        if self.keps:
            return self.step_keps()
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
