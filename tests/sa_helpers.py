"""Shared set-up of the SA_BOUNDARY tests: the SABox mirror through the neighbour phase of the ORACLE."""
import numpy as np

from gpusph_amd import defs as D
from gpusph_amd.problem import SABox
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
