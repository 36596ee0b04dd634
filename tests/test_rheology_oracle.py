"""Generalized Newtonian rheologies (BINGHAM .. ZHU), CPU side: the oracle's restatement of effectiveViscDevice and of the
forces with a per-particle viscosity against closed forms, a float64 all-pairs shear rate, and the Newtonian limit."""
import ctypes as C
import math
import numpy as np
import pytest

from gpusph_amd import defs as D
from gpusph_amd.problem import Poiseuille, info_type
import oracle_lib as ol

RHEOLOGIES = [D.BINGHAM, D.PAPANASTASIOU, D.POWER_LAW, D.HERSCHEL_BULKLEY, D.ALEXANDROU, D.DEKEE_TURCOTTE, D.ZHU]


def _problem(rheology, **kw):
    extra = {}
    if D.POWER_LAW <= rheology < D.DEKEE_TURCOTTE:
        extra["power_law_n"] = 0.7
    if rheology >= D.DEKEE_TURCOTTE:
        extra["exponential_coeff"] = 0.3
    extra.update(kw)
    return Poiseuille(12, rheology=rheology, **extra)


def _closed_form(pp, rheology, S, limit):
    k, ys, nl, m = pp.visc_consistency[0], pp.yield_strength[0], pp.visc_nonlinear_param[0], pp.visc_regularization_param[0]
    if rheology >= D.DEKEE_TURCOTTE:
        shear = k * math.exp(-nl * S)
    elif rheology >= D.POWER_LAW:
        shear = k * S ** (nl - 1) if S > 0 else (k if nl == 1 else math.inf)
    else:
        shear = k
    if rheology in (D.PAPANASTASIOU, D.ALEXANDROU, D.ZHU):
        y = ys * (m if S == 0 else -math.expm1(-m * S) / S)
    elif rheology == D.POWER_LAW:
        y = 0.0
    else:
        y = ys / S if S > 0 else math.inf
    return min(shear + y, limit)


@pytest.mark.parametrize("rheology", RHEOLOGIES)
def test_effective_viscosity_closed_forms(rheology):
    pr = _problem(rheology)
    pp = pr.physparams
    o = ol.Oracle(ol.orc_params_from(pr.sphx_params(pr.num_particles), pr))
    assert pp.visccoeff[0] == pp.visc_consistency[0]          # generalized Newtonian: always the dynamic value (GPUSPH.cc:1503-1508)
    limit = pp.limiting_kinvisc * pp.rho0[0]
    for S in [0.0, 1e-6, 1e-4, 5e-4, 9.9e-4, 1.0e-3, 1.1e-3, 1e-2, 0.3, 1.0, 7.0, 150.0]:
        got = float(o.L.orc_effective_visc_value(C.byref(o.p), C.c_float(S), C.c_int(0)))
        want = _closed_form(pp, rheology, float(np.float32(S)), limit)
        assert got == pytest.approx(want, rel=3e-6), (rheology, S)
    # the regularised yield term is continuous where the Horner form hands over to the exponential (m S = 1)
    if rheology in (D.PAPANASTASIOU, D.ALEXANDROU, D.ZHU):
        m = pp.visc_regularization_param[0]
        a = float(o.L.orc_effective_visc_value(C.byref(o.p), C.c_float(np.nextafter(np.float32(1 / m), np.float32(0))), C.c_int(0)))
        b = float(o.L.orc_effective_visc_value(C.byref(o.p), C.c_float(np.float32(1 / m)), C.c_int(0)))
        assert abs(a - b) < 3e-6 * abs(b)


def test_parameter_mirror_follows_physparams():
    pr = _problem(D.HERSCHEL_BULKLEY)
    pp = pr.physparams
    # ys = F rho lz/4 (Poiseuille.inc:72), raised limiting viscosity = max(1e3, ys m + k) (physparams.h:599-603)
    assert pp.yield_strength[0] == pytest.approx(0.05 * 1.0 * 1.0 / 4) and pp.limiting_kinvisc == 1000.0
    pp.set_yield_strength(0, 2.0)
    assert pp.limiting_kinvisc == pytest.approx(2.0 * 1000.0 + 0.1)
    with pytest.raises(ValueError):
        pp.set_visc_exponential_coeff(0, 1.0)             # must_be_exponential_rheology
    assert _problem(D.ZHU).physparams.visc_nonlinear_param == [pytest.approx(0.3)]
    assert Poiseuille(12, rheology=D.ZHU).physparams.visc_nonlinear_param == [0.0]          # Newtonian-reducing defaults
    assert Poiseuille(12, rheology=D.POWER_LAW).physparams.visc_nonlinear_param == [1.0]
    sp = pr.sphx_params(pr.num_particles)
    assert sp.is_const_visc == 0 and sp.rheologytype == D.HERSCHEL_BULKLEY


def _sheared(pr, g=0.8):
    sim = ol.OracleSim(pr)
    sim.build_neibs()
    n = sim.n
    gp = pr.global_pos(sim.pos[:n], sim.hash[:n])
    fluid = info_type(sim.info[:n]) == D.PT_FLUID
    sim.vel[:n, 0] = (g * gp[:, 2] + 0.03 * np.sin(5 * gp[:, 1])).astype(np.float32)      # walls too: no jump at the wall
    sim.vel[:n, 1] = (0.1 * g * gp[:, 0] * fluid).astype(np.float32) * 0
    return sim, gp


def test_shear_rate_norm_equals_all_pairs_and_the_imposed_shear():
    pr = _problem(D.PAPANASTASIOU, compvisc=D.DYNAMIC)
    sim, gp = _sheared(pr)
    n = sim.n
    eff, mx = sim.o.effective_visc(sim.pos, sim.vel, sim.info, sim.hash, sim.cs, sim.nl, n)
    p = sim.o.p
    h, R = float(p.slength), float(p.influenceradius)
    fc = float(sim.o.L.orc_fcoeff(C.c_int(D.WENDLAND), C.c_float(h), C.c_float(2.0)))
    t = info_type(sim.info[:n])
    L = np.array([pr.lx, pr.ly, 0.0])
    v = sim.vel[:n, :3].astype(np.float64)
    rho = (sim.vel[:n, 3].astype(np.float64) + 1) * float(p.rho0[0])
    mass = sim.pos[:n, 3].astype(np.float64)
    rng = np.random.default_rng(2)
    picks = rng.choice(np.where(t == D.PT_FLUID)[0], 40, replace=False)
    limit = pr.physparams.limiting_kinvisc * pr.physparams.rho0[0]
    for i in picks:
        rel = gp[i] - gp
        rel[:, :2] -= np.round(rel[:, :2] / L[:2]) * L[:2]          # periodic in x and y
        d = np.sqrt((rel ** 2).sum(1))
        near = (d < R) & (d > 0)
        if t[i] != D.PT_FLUID:
            near &= t == D.PT_FLUID
        f = (d[near] / h - 2) ** 3 * fc
        w = f * mass[near] / rho[near]
        dv = v[i] - v[near]
        G = -(dv[:, :, None] * (rel[near] * w[:, None])[:, None, :]).sum(0)      # G[a][b] = d v_a / d x_b
        Dm = G + G.T
        S = math.sqrt((Dm * Dm).sum() / 2)
        want = _closed_form(pr.physparams, D.PAPANASTASIOU, S, limit)
        assert eff[i] == pytest.approx(want, rel=2e-4)
    # deep in the channel the discrete gradient sees the imposed shear rate (to the kernel's first-moment error)
    deep = (t == D.PT_FLUID) & (np.abs(gp[:, 2]) < 0.1)
    k, ys, m = pr.physparams.visc_consistency[0], pr.physparams.yield_strength[0], pr.physparams.visc_regularization_param[0]
    ideal = k + ys * (1 - math.exp(-m * 0.8)) / 0.8
    assert np.abs(eff[:n][deep] / ideal - 1).max() < 0.06
    # the reduction: largest kinematic viscosity (= the clamp-free rest value of the wall rows here)
    assert mx == pytest.approx(float((eff[:n] / rho).max()), rel=1e-6)


def test_kinematic_storage_divides_by_the_density():
    a, _ = _sheared(_problem(D.BINGHAM, compvisc=D.DYNAMIC))
    b, _ = _sheared(_problem(D.BINGHAM, compvisc=D.KINEMATIC))
    for s in (a, b):
        s.vel[:s.n, 3] = 0.01
    ea, ma = a.o.effective_visc(a.pos, a.vel, a.info, a.hash, a.cs, a.nl, a.n)
    eb, mb = b.o.effective_visc(b.pos, b.vel, b.info, b.hash, b.cs, b.nl, b.n)
    rho = np.float32(1.01) * np.float32(a.o.p.rho0[0])
    np.testing.assert_allclose(eb[:a.n], ea[:a.n] / rho, rtol=1e-6)
    assert ma == pytest.approx(mb, rel=1e-6)


@pytest.mark.parametrize("compvisc", [D.KINEMATIC, D.DYNAMIC])
def test_newtonian_limit_of_the_forces(compvisc):
    # a power law with n = 1 and no yield strength IS the Newtonian fluid: same forces up to the rounding of the
    # non-constant averaging flavour (the Newtonian single-fluid framework takes the constant-viscosity shortcut)
    gn, _ = _sheared(Poiseuille(12, rheology=D.POWER_LAW, compvisc=compvisc, viscavg=D.ARITHMETIC))
    nw, _ = _sheared(Poiseuille(12, rheology=D.NEWTONIAN, compvisc=compvisc, viscavg=D.ARITHMETIC))
    n = gn.n
    eff, mx = gn.o.effective_visc(gn.pos, gn.vel, gn.info, gn.hash, gn.cs, gn.nl, n)
    want = 0.1 * 1.0 if compvisc == D.DYNAMIC else 0.1
    np.testing.assert_allclose(eff[:n], want, rtol=1e-6)
    assert mx == pytest.approx(0.1, rel=1e-6)
    f_gn = gn.o.forces(gn.pos, gn.vel, gn.info, gn.hash, gn.cs, gn.nl, n, effvisc=eff)[0]
    f_nw = nw.o.forces(nw.pos, nw.vel, nw.info, nw.hash, nw.cs, nw.nl, n)[0]
    scale = np.abs(f_nw[:n, :3]).max()
    assert scale > 0.05 and np.abs(f_gn[:n] - f_nw[:n]).max() < 2e-6 * scale
    # and the viscosity matters: a shear-thinning fluid feels less viscous force in the same flow
    thin, _ = _sheared(Poiseuille(12, rheology=D.POWER_LAW, compvisc=compvisc, viscavg=D.ARITHMETIC, power_law_n=0.5), g=4.0)
    e2, _ = thin.o.effective_visc(thin.pos, thin.vel, thin.info, thin.hash, thin.cs, thin.nl, n)
    t = info_type(thin.info[:n])
    deep = (t == D.PT_FLUID)
    assert np.median(e2[:n][deep]) < 0.7 * want


def test_steps_use_the_viscous_limit_of_the_largest_effective_viscosity():
    pr = _problem(D.PAPANASTASIOU)
    sim = ol.OracleSim(pr)
    for _ in range(3):
        sim.step()
    n = sim.n
    assert np.isfinite(sim.vel[:n]).all() and np.isfinite(sim.pos[:n]).all()
    h = float(sim.o.p.slength)
    # dt = 0.125 h^2 / nu_max with nu_max = (k + ys m)/rho at rest (dtreduce, src/cuda/forces.cu:586-600)
    nu_max = (0.1 + pr.physparams.yield_strength[0] * 1000.0) / 1.0
    assert sim.max_kinvisc == pytest.approx(nu_max, rel=1e-3)
    assert sim.dt == pytest.approx(0.125 * h * h / sim.max_kinvisc, rel=1e-5)
    assert np.abs(sim.vel[:n, 0]).max() > 0


@pytest.mark.parametrize("viscmodel", [D.MONAGHAN, D.ESPANOL_REVENGA])
@pytest.mark.parametrize("compvisc", [D.KINEMATIC, D.DYNAMIC])
def test_monaghan_and_espanol_revenga_viscous_models_equal_all_pairs(viscmodel, compvisc):
    """visc_model<MONAGHAN> directs the Morris coefficient along r with (r.v)/(r.r) for approaching pairs; visc_model<ESPANOL_REVENGA>
    has a shear and a bulk viscosity acting along v and along r: float64 all-pairs evaluation of the viscous acceleration
    (forces with viscosity minus forces without)."""
    kw = dict(viscmodel=viscmodel, compvisc=compvisc, viscavg=D.ARITHMETIC)
    if viscmodel == D.ESPANOL_REVENGA:
        kw["bulk_visc"] = 0.04
    pr = Poiseuille(12, **kw)
    sim, gp = _sheared(pr, g=1.5)
    n = sim.n
    rng = np.random.default_rng(1)
    fluid = info_type(sim.info[:n]) == D.PT_FLUID
    sim.vel[:n, :3][fluid] += rng.uniform(-0.2, 0.2, size=(fluid.sum(), 3)).astype(np.float32)
    f = sim.o.forces(sim.pos, sim.vel, sim.info, sim.hash, sim.cs, sim.nl, n)[0][:n]
    keep = sim.o.p.rheologytype
    sim.o.p.rheologytype = D.INVISCID
    f0 = sim.o.forces(sim.pos, sim.vel, sim.info, sim.hash, sim.cs, sim.nl, n)[0][:n]
    sim.o.p.rheologytype = keep
    got = (f - f0)[:, :3].astype(np.float64)
    p = sim.o.p
    h, R = float(p.slength), float(p.influenceradius)
    fc = float(sim.o.L.orc_fcoeff(C.c_int(D.WENDLAND), C.c_float(h), C.c_float(2.0)))
    L = np.array([pr.lx, pr.ly])
    v = sim.vel[:n, :3].astype(np.float64)
    rho = (sim.vel[:n, 3].astype(np.float64) + 1) * float(p.rho0[0])
    m = sim.pos[:n, 3].astype(np.float64)
    eps = float(p.epsartvisc)
    mu, zeta = 0.1 * 1.0, 0.04           # dynamic shear viscosity (nu rho0), bulk viscosity
    t = info_type(sim.info[:n])
    scale = np.abs(got).max()
    assert scale > 0.05
    for i in rng.choice(np.where(t == D.PT_FLUID)[0], 40, replace=False):
        rel = gp[i] - gp
        rel[:, :2] -= np.round(rel[:, :2] / L) * L
        d2 = (rel ** 2).sum(1); d = np.sqrt(d2)
        near = (d < R) & (d > 0)
        F = (d[near] / h - 2) ** 3 * fc
        dv = v[i] - v[near]
        vdp = (dv * rel[near]).sum(1)
        vol = m[near] / (rho[i] * rho[near])
        if viscmodel == D.MONAGHAN:
            # visc_avg (arithmetic): m (mu_a + mu_b)/(rho_a rho_b) with mu = nu rho (kinematic) or mu (dynamic)
            mu_a = 0.1 * rho[i] if compvisc == D.KINEMATIC else mu
            mu_b = 0.1 * rho[near] if compvisc == D.KINEMATIC else mu
            c = np.where(vdp < 0, 10.0 * vdp / (d2[near] + eps), 0.0)
            want = ((vol * (mu_a + mu_b) * F * c)[:, None] * rel[near]).sum(0)
        else:
            mu_a = 0.1 * rho[i] if compvisc == D.KINEMATIC else mu
            mu_b = 0.1 * rho[near] if compvisc == D.KINEMATIC else mu
            vt = np.broadcast_to((mu_a + mu_b) / 2 / 3, vdp.shape)
            want = ((vol * F)[:, None] * ((5 * vt - zeta)[:, None] * dv + (5 * (vt + zeta) * vdp / (d2[near] + eps))[:, None] * rel[near])).sum(0)
        assert np.abs(got[i] - want).max() <= 3e-4 * scale
    # the models tighten the viscous dt limit: x monaghan_visc_coeff (10) and x 5 (GPUWorker.cc:2013-2022)
    sim2 = ol.OracleSim(Poiseuille(12, **kw))
    sim2.step()
    hh = float(sim2.o.p.slength)
    assert sim2.dt == pytest.approx(0.125 * hh * hh / (0.1 * (10.0 if viscmodel == D.MONAGHAN else 5.0)), rel=1e-5)
