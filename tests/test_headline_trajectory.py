"""A float64 TRAJECTORY known answer for the headline option set: ten whole steps (predictor, corrector, adaptive dt) of DamBreak3D as
bench.py runs it, integrated here in float64 from GLOBAL positions with neighbours by distance -- no cells, no hash, no neighbour
list, no sort, no shared code with oracle/ or the library -- and held against the oracle's ten steps (and, on a GPU, the engine's).

tests/test_headline_allpairs.py holds one evaluation of the pair sums; this holds what is built on them: the Euler update of both
half steps with the neighbours of the last rebuild (src/cuda/euler_kernel.def:396-470: the predictor advances with the velocity of step n over dt/2, the corrector from
step n with v_n + f* dt/2 over dt; densities of fluid AND DYN-boundary particles follow the continuity equation, walls do not
move), the CFL term (max over the fluid of max(|a|, c^2/h), forces_kernel.def:3436-3457), the dt rule (dtadaptfactor min(sqrt(h /
max cfl), h / (1.1 c0)), src/cuda/forces.cu:556-606) and its use (dt of the next step = min(dt after the predictor's forces, dt after
the corrector's), GPUSPH.cc:636-699).

Two comparisons.
(1) STEP BY STEP: each of the ten steps is repeated in float64 from the float32 state the tested path had before it, and the
    state after it is compared.  Nothing feeds back, so the bar is that of test_headline_allpairs carried through ONE step: a
    float32 evaluation of the pair sums is held to 2e-5 of the largest acceleration A (density rate Q) of the state, which over a
    step of dt allows a velocity error of 2e-5 A dt, a density error of 2e-5 Q dt and a position error of 2e-5 A dt^2 / 2, plus the
    rounding of the stored state (two updates of 6e-8 of the stored magnitude; positions are cell-local).  The dt the step hands
    on must agree to 1e-6.  A and Q are the largest values the float64 step meets; nothing is fitted per particle.
(2) FREE RUN: the float64 integrator runs its own ten steps from the common start; the same bar over the whole time T of the
    run, 2e-5 {A T, Q T, A T^2/2}: nothing drifts.
Pairs on the threshold of the Colagrossi switch may be decided either way by the last bits of a float32 pressure.  Their terms are
summed per particle along the float64 run and widen that particle's density bound; what such a density offset can do to the
accelerations of the particle and of its neighbours in the evaluations that follow (through P/rho^2, to first order) widens their
velocity and position bounds.  Few particles are touched (asserted)."""
import numpy as np
import pytest
from scipy.spatial import cKDTree

import oracle_lib as ol
from gpusph_amd import defs as D
from gpusph_amd.problem import DamBreak3D

STEPS = 10


def _problem():
    return DamBreak3D(0.025, obstacle=True, jitter=0.15, hydrostatic=True)


def _perturb(prob, pos, hsh, vel, info, seed=11):
    """a moving start for the fluid (sheared, converging field + noise), so that ten steps integrate something; walls at rest"""
    n = len(pos)
    fluid = (info.reshape(-1, 4)[:, 0] & 7) == D.PT_FLUID
    rng = np.random.default_rng(seed)
    gp = prob.global_pos(pos, hsh)
    v = np.stack([0.8*np.sin(3.0*gp[:, 2]) - 0.5*gp[:, 0], 0.6*np.cos(2.0*gp[:, 0]) - 0.4*gp[:, 1],
                  0.3*gp[:, 0]*gp[:, 1] - 0.5*gp[:, 2]], axis=1) + rng.normal(0, 0.1, size=(n, 3))
    out = vel.copy()
    out[fluid, :3] = v.astype(np.float32)[fluid]
    out[:, 3] += rng.uniform(-2e-3, 2e-3, size=n).astype(np.float32)
    return out


class Float64Run:
    """the integrator described in the module docstring; state in float64, keyed by particle id"""

    def __init__(self, prob, gp, vel, info, params):
        p = params
        self.h = float(p.slength); self.rho0 = float(p.rho0[0]); self.B = float(p.bcoeff[0]); self.gam = float(p.gammacoeff[0])
        self.c0 = float(p.sscoeff[0]); self.cpow = float(p.sspowercoeff[0])
        self.g = np.array([p.gravity[0], p.gravity[1], p.gravity[2]], dtype=np.float64)
        self.alpha = float(p.artvisccoeff); self.eps = float(p.epsartvisc); self.Dc = float(p.densityDiffCoeff)
        self.fcoeff = 105.0/(128.0*np.pi*self.h**5); self.R = float(p.influenceradius)
        self.dtadapt = float(p.dtadaptfactor)
        self.sspeed_cfl = float(np.float32(np.float64(np.float32(self.c0))*1.1))
        self.x = gp.astype(np.float64).copy()
        self.v = vel[:, :3].astype(np.float64).copy()
        self.r = vel[:, 3].astype(np.float64).copy()            # relative density
        ptype = info[:, 0] & 7
        self.fluid, self.bound = ptype == D.PT_FLUID, ptype == D.PT_BOUNDARY
        assert (self.fluid | self.bound).all()
        self.dt = float(np.float32(prob.simparams.dt))
        self.rebuild = int(prob.simparams.buildneibsfreq)
        self.t = 0.0
        self.dts = []
        self.amb = np.zeros(len(self.x))                          # sum over the run of dt |threshold terms| / rho0, per particle
        self.amax = self.qmax = 0.0                               # largest acceleration / density rate met along the run
        self.vunc = np.zeros(len(self.x)); self.xunc = np.zeros(len(self.x))

    def rates(self, x, v, r, m, runc=None):
        """(accelerations, d(relative density)/dt, dt this state allows, threshold terms per particle, and -- given runc, how far
        each relative density may be off because a threshold pair was decided the other way earlier -- how far that can move each
        acceleration, to first order: a reads the densities through P/rho^2, d(P/rho^2) ~ c^2 rho0 / rho^2 d(rho~))"""
        n = len(x)
        ratio = r + 1.0
        rho = ratio*self.rho0
        P = self.B*(ratio**self.gam - 1.0)
        c = self.c0*ratio**self.cpow
        # what float32 can do to a pressure: P = B (ratio^gamma - 1) carries the rounding of ratio^gamma (near the free surface
        # P is a small difference of numbers close to one), and the powf of the tested path a few ulp more
        Pulp = self.B*np.spacing((ratio**self.gam).astype(np.float32)).astype(np.float64)*8.0
        i, j = self.pairs
        d = x[i] - x[j]
        r2 = (d*d).sum(axis=1)
        inside = r2 < self.R*self.R            # a listed pair counts while it is within the support
        i, j, d, r2 = i[inside], j[inside], d[inside], r2[inside]
        dist = np.sqrt(r2)
        mF = m[j]*(dist/self.h - 2.0)**3*self.fcoeff
        vr = ((v[i] - v[j])*d).sum(axis=1)
        drdt = np.bincount(i, weights=mF*vr, minlength=n)
        # Colagrossi: fluid neighbours, where the pressure difference exceeds the hydrostatic one
        nf = self.fluid[j]
        gd = np.abs(d @ self.g)*rho[i]
        margin = np.abs(P[i] - P[j]) - gd
        dterm = self.Dc*self.c0*(rho[j]/rho[i] - 1.0)*mF
        edge = nf & (np.abs(margin) <= Pulp[i] + Pulp[j] + 1e-6*gd)
        on = nf & (margin >= 0.0) & ~edge
        drdt -= np.bincount(i, weights=np.where(on, dterm, 0.0) + 0.5*np.where(edge, dterm, 0.0), minlength=n)
        amb = 0.5*np.bincount(i, weights=np.where(edge, np.abs(dterm), 0.0), minlength=n)/self.rho0
        # momentum: fluid particles only (walls without force feedback)
        fi = self.fluid[i]
        pg = P[i]/rho[i]**2 + P[j]/rho[j]**2
        visc = np.where(vr < 0.0, vr*self.h*self.alpha*(c[i] + c[j])/((r2 + self.eps)*(rho[i] + rho[j])), 0.0)
        w = np.where(fi, (visc - pg)*mF, 0.0)
        acc = np.stack([np.bincount(i, weights=w*d[:, a], minlength=n) for a in range(3)], axis=1)
        acc[self.fluid] += self.g
        self.aunc = None
        if runc is not None:
            s = c*c*self.rho0/(rho*rho)*runc                      # |d(P/rho^2)| per particle
            lever = np.where(fi, np.abs(mF)*dist, 0.0)            # m_j |F_ij| |r_ij|
            self.aunc = np.bincount(i, weights=lever*(s[i] + s[j]), minlength=n)
        cfl = np.maximum(np.sqrt((acc[self.fluid]**2).sum(axis=1)), c[self.fluid]**2/self.h).max()
        dt = self.dtadapt*min(np.sqrt(self.h/cfl), self.h/self.sspeed_cfl)
        self.amax = max(self.amax, np.abs(acc).max()); self.qmax = max(self.qmax, np.abs(drdt).max()/self.rho0)
        return acc, drdt/self.rho0, dt, amb

    def step(self, m):
        dt = self.dt
        f = self.fluid
        if len(self.dts) % self.rebuild == 0:
            # the neighbours of a particle are those within the influence radius at the last rebuild (every buildneibsfreq
            # iterations, src/Integrator.cc:94-250), here by distance; both directions of every pair that interacts
            pairs = cKDTree(self.x).query_pairs(self.R*(1 - 1e-7), output_type="ndarray")
            i = np.concatenate([pairs[:, 0], pairs[:, 1]]); j = np.concatenate([pairs[:, 1], pairs[:, 0]])
            keep = (self.fluid[i] & (self.fluid[j] | self.bound[j])) | (self.bound[i] & self.fluid[j])
            self.pairs = (i[keep], j[keep])
        a1, q1, dt1, amb1 = self.rates(self.x, self.v, self.r, m)
        xs, vs, rs = self.x.copy(), self.v.copy(), self.r.copy()
        xs[f] += self.v[f]*(dt/2)
        vs[f] += a1[f]*(dt/2)
        rs += q1*(dt/2)                      # fluid and DYN-boundary rows
        a2, q2, dt2, amb2 = self.rates(xs, vs, rs, m, runc=self.amb + amb1*(dt/2))
        self.x[f] += (self.v[f] + a2[f]*(dt/2))*dt
        self.v[f] += a2[f]*dt
        self.r += q2*dt
        self.vunc += self.aunc*dt                # what the threshold pairs' densities can do to the velocities ...
        self.xunc += self.vunc*dt                # ... and through them to the positions
        self.amb += (amb1*0.5 + amb2)*dt
        self.t += dt
        self.dts.append(dt)
        self.dt = min(dt1, dt2)


def _ids(info):
    return info[:, 2].astype(np.uint32) | (info[:, 3].astype(np.uint32) << 16)


def _start():
    """the common start: (problem, oracle sim with the perturbed velocities, ids in ascending order)"""
    prob = _problem()
    sim = ol.OracleSim(prob)
    n = sim.n
    sim.vel[:n] = _perturb(prob, sim.pos[:n], sim.hash[:n], sim.vel[:n], sim.info[:n])
    return prob, sim, np.sort(_ids(sim.info[:n]))


def _by_id(prob, pos, vel, info, hsh):
    o = np.argsort(_ids(info), kind="stable")
    return prob.global_pos(pos, hsh)[o], vel[o, :3].astype(np.float64), vel[o, 3].astype(np.float64), o


class _OraclePath:
    """the tested path behind one face: state() by id, dt of the step to come, step()"""
    def __init__(self, prob, sim):
        self.prob, self.sim = prob, sim
    def state(self):
        s = self.sim; n = s.n
        return _by_id(self.prob, s.pos[:n], s.vel[:n], s.info[:n], s.hash[:n])[:3]
    def dt(self):
        return float(self.sim.dt)
    def step(self):
        self.sim.step()
    def time(self):
        return float(self.sim.t)
    def info_mass(self):
        s = self.sim; n = s.n
        o = np.argsort(_ids(s.info[:n]), kind="stable")
        return s.info[:n][o], s.pos[:n, 3].astype(np.float64)[o]


class _EnginePath:
    def __init__(self, eng):
        self.eng = eng
    def state(self):
        import torch
        torch.cuda.synchronize()
        out = self.eng.download()
        return _by_id(self.eng.problem, out["pos"], out["vel"], out["info"].reshape(-1, 4), out["hash"])[:3]
    def dt(self):
        return float(self.eng.current_dt())
    def step(self):
        self.eng.step()
    def time(self):
        return float(self.eng.d_t.item())
    def info_mass(self):
        out = self.eng.download()
        info = out["info"].reshape(-1, 4)
        o = np.argsort(_ids(info), kind="stable")
        return info[o], out["pos"][:, 3].astype(np.float64)[o]


def _step_by_step(path, prob, params, what):
    cell = float(max(prob.m_cellsize))
    info, m = path.info_mass()
    x, v, r = path.state()
    run = Float64Run(prob, x, np.concatenate([v, r[:, None]], axis=1), info, params)
    f = run.fluid
    worst = dict(v=0.0, r=0.0, x=0.0, dt=0.0)
    for k in range(STEPS):
        x0, v0, r0 = path.state()
        run.x, run.v, run.r = x0.copy(), v0.copy(), r0.copy()      # the float32 state before the step (the list is that of step 0)
        run.dt = path.dt()
        run.amb[:] = 0.0; run.vunc[:] = 0.0; run.xunc[:] = 0.0; run.amax = run.qmax = 0.0
        run.step(m)
        path.step()
        x1, v1, r1 = path.state()
        dt = run.dts[-1]
        bound_v = 2e-5*run.amax*dt + 2*6e-8*np.abs(v1).max()
        bound_r = 2e-5*run.qmax*dt + 2*6e-8*np.abs(r1).max()
        bound_x = 2e-5*run.amax*dt*dt/2 + 2*6e-8*cell
        ev = (np.abs(v1 - run.v).max(axis=1) - run.vunc)[f].max(); er = (np.abs(r1 - run.r) - run.amb).max()
        ex = (np.abs(x1 - run.x).max(axis=1) - run.xunc).max()
        edt = abs(path.dt() - run.dt)/run.dt
        worst = dict(v=max(worst["v"], ev/bound_v), r=max(worst["r"], er/bound_r), x=max(worst["x"], ex/bound_x), dt=max(worst["dt"], edt))
        assert np.abs(x1[~f] - x0[~f]).max() == 0.0, "walls moved"
        assert ev <= bound_v, "step %d, velocities: %g, allowed %g" % (k, ev, bound_v)
        assert er <= bound_r, "step %d, densities: %g, allowed %g" % (k, er, bound_r)
        assert ex <= bound_x, "step %d, positions: %g, allowed %g" % (k, ex, bound_x)
        assert edt <= 1e-6, "step %d, the dt handed on: %g against %g" % (k, path.dt(), run.dt)
        # the bounds are far below what a step changes
        assert bound_v < 2e-3*np.abs(run.v - v0).max() and bound_r < 2e-2*np.abs(run.r - r0).max()
        assert (run.amb > 0).mean() < 0.05 and (run.vunc > bound_v).mean() < 0.05
    print("%s, step by step: worst share of the bound: velocities %.2f, densities %.2f, positions %.2f; dt handed on to %.1e"
          % (what, worst["v"], worst["r"], worst["x"], worst["dt"]))


def _free_run(path, prob, params, what):
    cell = float(max(prob.m_cellsize))
    info, m = path.info_mass()
    x, v, r = path.state()
    run = Float64Run(prob, x, np.concatenate([v, r[:, None]], axis=1), info, params)
    start = (run.x.copy(), run.v.copy(), run.r.copy())
    f = run.fluid
    dts = []
    for _ in range(STEPS):
        run.step(m)
        dts.append(path.dt())
        path.step()
    x1, v1, r1 = path.state()
    dxs = np.abs(run.x - start[0]).max(); dvs = np.abs(run.v - start[1]).max(); drs = np.abs(run.r - start[2]).max()
    assert dxs > 1e-4 and dvs > 1e-2 and drs > 1e-5, "the run does not move enough to test an integrator"
    T = run.t
    K = 1.0
    bound_v = K*2e-5*run.amax*T; bound_r = K*2e-5*run.qmax*T; bound_x = K*2e-5*run.amax*T*T/2 + STEPS*2*6e-8*cell
    ev = (np.abs(v1 - run.v).max(axis=1) - run.vunc)[f].max(); er = (np.abs(r1 - run.r) - run.amb).max()
    ex = (np.abs(x1 - run.x).max(axis=1) - run.xunc).max()
    edt = np.abs(np.array(dts) - np.array(run.dts)).max()/min(run.dts)
    print("%s, free run: T %.3g, A %.3g, Q %.3g; dv %.3g = %.2f of its bound (changed by %.3g), drho %.3g = %.2f (changed by %.3g), "
          "dx %.3g = %.2f (moved by %.3g), dt rel %.3g, threshold rows %d"
          % (what, T, run.amax, run.qmax, ev, ev/bound_v, dvs, er, er/bound_r, drs, ex, ex/bound_x, dxs, edt, int((run.amb > 0).sum())))
    assert ev <= bound_v and er <= bound_r and ex <= bound_x
    loose = ((run.vunc > bound_v) & f).sum()/f.sum()
    print("   rows whose velocity bound the threshold pairs more than double: %.1f %%" % (100*loose))
    assert loose < 0.10
    assert bound_v < 1e-3*dvs and bound_r < 1e-2*drs and bound_x < 1e-3*dxs, "bounds too loose to mean anything"
    assert edt <= 1e-5, "dt sequence differs: %g" % edt
    assert abs(path.time() - run.t) <= 1e-5*run.t


def test_every_step_of_the_oracle_equals_a_float64_step_from_the_same_state():
    prob, sim, _ = _start()
    _step_by_step(_OraclePath(prob, sim), prob, sim.o.p, "oracle")


def test_ten_steps_of_the_oracle_follow_the_float64_trajectory():
    prob, sim, _ = _start()
    _free_run(_OraclePath(prob, sim), prob, sim.o.p, "oracle")


def _engine():
    import torch
    from gpusph_amd.engine import TimestepEngine
    eng = TimestepEngine(_problem(), device="cuda:0")
    n = eng.n
    pos0, hsh0 = eng.pos[:n].cpu().numpy(), eng.hash[:n].cpu().numpy().view(np.uint32)
    info0 = eng.info[:n].cpu().numpy().view(np.uint16)
    eng.vel[:n] = torch.from_numpy(_perturb(eng.problem, pos0, hsh0, eng.vel[:n].cpu().numpy(), info0)).to(eng.device)
    return eng


@pytest.mark.gpu
def test_every_step_of_the_engine_equals_a_float64_step_from_the_same_state():
    prob, sim, _ = _start()
    eng = _engine()
    _step_by_step(_EnginePath(eng), eng.problem, sim.o.p, "engine")


@pytest.mark.gpu
def test_ten_steps_of_the_engine_follow_the_float64_trajectory():
    prob, sim, _ = _start()
    eng = _engine()
    _free_run(_EnginePath(eng), eng.problem, sim.o.p, "engine")
