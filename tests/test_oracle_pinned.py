"""Pins the oracle against the REFERENCE: golden vectors produced by oracle/_ref (the reference's own
sph_core.cu / particleinfo.h / hashkey.h / common_types.h compiled here, tests/golden/make_golden.py),
and -- when oracle/_ref is present -- against the live reference library too."""
import ctypes as C
import os
import numpy as np
import pytest

import oracle_lib as ol
from gpusph_amd import defs as D

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _w_f(k, r, h):
    """oracle W/F through the same entry points the oracle's force loop uses"""
    L = ol.lib()
    return L.orc_W(int(k), float(r), float(h)), L.orc_F(int(k), float(r), float(h))


def test_kernel_functions_match_reference_golden():
    g = np.load(os.path.join(GOLD, "ref_kernels.npz"))
    L = ol.lib()
    for i in range(len(g["r"])):
        k, r, h = int(g["kerneltype"][i]), float(g["r"][i]), float(g["slength"][i])
        kr = 3.0 if k == D.GAUSSIAN else 2.0
        # coefficient formulas restate src/cuda/forces.cu:274-309; W/F bodies are pinned bit-for-bit
        assert np.float32(L.orc_wcoeff(k, h, kr)) == g["wcoeff"][i]
        assert np.float32(L.orc_fcoeff(k, h, kr)) == g["fcoeff"][i]
        w, f = _w_f(k, r, h)
        assert np.float32(w) == g["W"][i], (k, r, h)
        assert np.float32(f) == g["F"][i], (k, r, h)


def test_kernel_functions_match_live_reference():
    ref = ol.ref()
    if ref is None:
        pytest.skip("oracle/_ref not built here (needs /root/reference)")
    L = ol.lib()
    rng = np.random.default_rng(5)
    for k in (1, 2, 3, 4):
        kr = 3.0 if k == 4 else 2.0
        h = float(np.float32(rng.uniform(1e-3, 0.5)))
        wc, fc = L.orc_wcoeff(k, h, kr), L.orc_fcoeff(k, h, kr)
        wsub = float(np.float32(np.exp(np.float32(-kr * kr))))
        for r in rng.uniform(1e-4, kr, 200).astype(np.float32) * np.float32(h):
            w, f = _w_f(k, float(r), h)
            assert np.float32(w) == np.float32(ref.ref_W(k, float(r), h, wc, wsub))
            assert np.float32(f) == np.float32(ref.ref_F(k, float(r), h, fc))


def test_wendland_kernel_analytic():
    """known answers independent of any code: W(0) = 21/(16 pi h^3), W(2h) = 0, F(2h) = 0"""
    h = 0.0195
    w0, _ = _w_f(D.WENDLAND, 0.0, h)
    assert abs(w0 - 21.0 / (16.0 * np.pi * h ** 3)) <= 1e-6 * w0
    w2, f2 = _w_f(D.WENDLAND, 2 * h, h)
    assert abs(w2) <= 1e-6 * w0 and abs(f2) <= 1e-12 * abs(_w_f(D.WENDLAND, h, h)[1]) + 1e-3


def test_datamodel_matches_reference_golden():
    g = np.load(os.path.join(GOLD, "ref_datamodel.npz"))
    L = ol.lib()

    class Info(C.Structure):
        _fields_ = [("x", C.c_uint16), ("y", C.c_uint16), ("z", C.c_uint16), ("w", C.c_uint16)]
    L.orc_info_id.restype = C.c_uint32; L.orc_info_id.argtypes = [Info]
    L.orc_info_type.restype = C.c_int; L.orc_info_type.argtypes = [Info]
    for i, row in enumerate(g["info"]):
        inf = Info(*map(int, row))
        assert L.orc_info_id(inf) == g["id"][i]
        assert L.orc_info_type(inf) == g["ptype"][i]
    # python-side mirrors used by the host code
    from gpusph_amd.problem import info_id, info_type
    assert np.array_equal(info_id(g["info"]), g["id"])
    assert np.array_equal(info_type(g["info"]).astype(np.int32), g["ptype"])
    assert np.array_equal((g["info"][:, 1] & 0xFFF).astype(np.int32), g["object"])
    assert np.array_equal((g["info"][:, 1] >> 12).astype(np.int32), g["fluid"])
    x = g["info"][:, 0].astype(np.uint32)
    t = x & 7
    pred = ((t == 0) * 1 | (t == 1) * 2 | (t == 2) * 4 | (t == 3) * 8 | ((x & D.FG_MOVING_BOUNDARY) != 0) * 16 |
            ((x & (D.FG_MOVING_BOUNDARY | D.FG_COMPUTE_FORCE)) != 0) * 32 | ((x & D.FG_COMPUTE_FORCE) != 0) * 64 |
            ((x & D.FG_SURFACE) != 0) * 128).astype(np.uint32)
    assert np.array_equal(pred, g["predicates"])
    assert np.array_equal(g["hashes"] & D.CELLTYPE_BITMASK, g["hash_reset"])
    assert np.array_equal(g["hashes"], g["hash_keep"])
    assert np.array_equal((g["cells"] + 1) << D.CELLNUM_SHIFT, g["encoded"])
    assert np.array_equal((((g["encoded"] + 37) >> D.CELLNUM_SHIFT).astype(np.int32) - 1), g["decoded"])
    c = g["constants"]
    assert list(c) == [D.CELLTYPE_BITMASK, D.CELL_HASH_MAX, D.NEIBINDEX_MASK, D.NEIBS_END, D.CELLNUM_ENCODED,
                       0 << 30, 1 << 30, 2 << 30, (3 << 30) & 0xFFFFFFFF, D.EMPTY_SEGMENT, D.MAX_CELLS]
    e = g["enums"]
    assert list(e) == [D.CUBICSPLINE, D.QUADRATIC, D.WENDLAND, D.GAUSSIAN, D.SPH_F1, D.COLAGROSSI, D.DYN_BOUNDARY,
                       D.LJ_BOUNDARY, D.PERIODIC_Z, D.SA_BOUNDARY, D.FERRARI,
                       D.SHEPARD_FILTER, D.MLS_FILTER, D.VORTICITY, D.TESTPOINTS, D.SURFACE_DETECTION,
                       D.ARTIFICIAL, D.SPS, D.LAMINAR_FLOW, D.MK_BOUNDARY, D.ENABLE_PLANES, D.ENABLE_DTADAPT,
                       8, 4, D.FG_SURFACE, D.PT_TESTPOINT, D.INVISCID,     # 8 = SPHX_MAX_PLANES, 4 = SPHX_MAX_FLUIDS
                       D.ENABLE_MULTIFLUID, D.ENABLE_REPACKING, D.NEWTONIAN, D.KINEMATIC, D.DYNAMIC, D.MORRIS, D.ARITHMETIC,
                       D.HARMONIC, D.GEOMETRIC, D.REPACK, D.SIMULATE, D.INTERFACE_DETECTION, D.FG_INTERFACE]
    assert list(np.isfinite(g["wvals"]).astype(np.int32)) == list(g["active"])


def test_oracle_pipeline_regression():
    """the oracle reproduces its own committed outputs (guards the fixture the GPU tests compare with)"""
    from gpusph_amd.problem import DamBreak3D
    g = np.load(os.path.join(GOLD, "oracle_pipeline.npz"))
    prob = DamBreak3D(float(g["deltap"]), obstacle=True, jitter=0.05, hydrostatic=False)
    arrs = prob.copy_to_array()
    assert np.array_equal(arrs["pos"].view(np.uint32), g["in_pos"].view(np.uint32))
    assert np.array_equal(arrs["hash"], g["in_hash"])
    sim = ol.OracleSim(prob)
    sim.step()
    assert np.array_equal(sim.hash, g["s1_hash"]) and np.array_equal(sim.nl, g["s1_neibs"])
    assert np.array_equal(sim.cs, g["s1_cellStart"]) and np.array_equal(sim.ce, g["s1_cellEnd"])
    assert np.array_equal(sim.pos.view(np.uint32), g["s1_pos"].view(np.uint32))
    assert np.array_equal(sim.forces.view(np.uint32), g["s1_forces"].view(np.uint32))
    for _ in range(10):
        sim.step()
    assert np.array_equal(sim.pos.view(np.uint32), g["s11_pos"].view(np.uint32))
    assert np.float32(sim.dt) == g["s11_dt"]


def test_oracle_features_regression():
    """tests/golden/oracle_features.npz: the oracle still produces the committed SPS / filter / post-processing /
    LJ / planes / moving-body vectors bit for bit (guards the restatement against accidental edits)."""
    import oracle_lib as ol
    from gpusph_amd import defs as D
    from gpusph_amd.problem import DamBreak3D
    g = np.load(os.path.join(GOLD, "oracle_features.npz"))
    prob = DamBreak3D(float(g["a_deltap"]), obstacle=True, jitter=0.1, hydrostatic=False, testpoints=[(0.2, 0.3, 0.2), (0.3, 0.4, 0.1)])
    prob.simparams.turbmodel = D.SPS
    dp = prob.m_deltap
    prob.physparams.smagfactor = float(np.float32((0.12 * dp) ** 2))
    prob.physparams.kspsfactor = float(np.float32((2.0 / 3.0) * 0.0066 * dp * dp))
    sim = ol.OracleSim(prob)
    sim.build_neibs()
    n = sim.n
    vel = g["a_vel"]
    eq = lambda a, b: np.array_equal(np.ascontiguousarray(a).view(np.uint8), np.ascontiguousarray(b).view(np.uint8))
    tau, tv = sim.o.sps(sim.pos, vel, sim.info, sim.hash, sim.cs, sim.nl, n, n)
    assert eq(tau, g["a_tau"]) and eq(tv, g["a_turbvisc"])
    assert eq(sim.o.filter(0, sim.pos, vel, sim.info, sim.hash, sim.cs, sim.nl, n), g["a_shepard"])
    assert eq(sim.o.filter(1, sim.pos, vel, sim.info, sim.hash, sim.cs, sim.nl, n), g["a_mls"])
    assert eq(sim.o.vorticity(sim.pos, vel, sim.info, sim.hash, sim.cs, sim.nl, n), g["a_vorticity"])
    info_s, nrm = sim.o.surface(sim.pos, vel, sim.info, sim.hash, sim.cs, sim.nl, n, normals=True)
    assert eq(info_s, g["a_surface_info"]) and eq(nrm, g["a_normals"])
    assert eq(sim.o.testpoints(sim.pos, vel, sim.info, sim.hash, sim.cs, sim.nl, n), g["a_testpoints"])


def test_oracle_features2_regression():
    """tests/golden/oracle_features2.npz: repacking forces + Euler, Newtonian viscosity (KINEMATICVISC with a feedback body,
    two fluids with non-constant kinematic/harmonic and dynamic/geometric averaging, planes with wall friction)"""
    import importlib.util
    import oracle_lib as ol
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLD, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)
    g = np.load(os.path.join(GOLD, "oracle_features2.npz"))
    eq = lambda a, b: np.array_equal(np.ascontiguousarray(a).view(np.uint8), np.ascontiguousarray(b).view(np.uint8))
    for tag, kind, make in mg.features2_cases():
        prob = make()
        sim = ol.OracleSim(prob)
        sim.build_neibs()
        n = sim.n
        vel = np.ascontiguousarray(g[tag + "_vel"])
        rb = getattr(prob, "num_obstacle", 0)
        if kind == "repack":
            f, cfl, nb, _, _ = sim.o.repack_forces(sim.pos, vel, sim.info, sim.hash, sim.cs, sim.nl, n, rb_count=rb)
            pr, vr = sim.o.euler_repack(sim.pos, vel, sim.info, sim.hash, f, n, float(np.float32(1.3e-4)), 1)
            assert eq(pr, g[tag + "_euler_pos"]) and eq(vr, g[tag + "_euler_vel"]), tag
        else:
            cof = 1 if prob.simparams.numforcesbodies else 0
            f, cfl, nb, _, _ = sim.o.forces(sim.pos, vel, sim.info, sim.hash, sim.cs, sim.nl, n, compute_object_forces=cof, rb_count=rb)
        assert eq(f, g[tag + "_forces"]), tag
        assert np.float32(sim.o.dtreduce(cfl, nb, sim.sspeed_cfl, sim.max_kinvisc)) == g[tag + "_dt"], tag


def test_visc_avg_matches_reference_golden():
    """tests/golden/ref_viscavg.npz was produced by the reference's own src/cuda/visc_avg.cu (compiled unmodified into
    oracle/_ref): the oracle's restatement of all 12 visc_avg flavours must reproduce it bit for bit."""
    import ctypes as C
    import oracle_lib as ol
    g = np.load(os.path.join(GOLD, "ref_viscavg.npz"))
    L = ol.lib()
    p = ol.OrcParams()
    p.rheologytype = 1       # the golden flavours are FullViscSpec<NEWTONIAN, ...> (the re-derived constness asks for it)
    n = len(g["visc"])
    for av in (0, 1, 2):     # single-fluid framework forced non-constant, kinematic: the reference lands in 2 m mu_i/(rho_i rho_j)
        p.compvisc, p.avgop, p.is_const_visc, p.simflags = 0, av, 0, 0
        got = np.array([L.orc_visc_avg(C.byref(p), float(g["visc"][i]), float(g["nvisc"][i]), float(g["rho"][i]),
                                       float(g["nrho"][i]), float(g["mass"][i])) for i in range(n)], dtype=np.float32)
        assert np.array_equal(got.view(np.uint32), g["va_single_0%d0" % av].view(np.uint32)), av
    for cv in (0, 1):
        for av in (0, 1, 2):
            for cst in (0, 1):
                p.compvisc, p.avgop, p.is_const_visc = cv, av, cst
                p.simflags = 0 if cst else (1 << 11)        # non-constant viscosity = multi-fluid framework
                got = np.array([L.orc_visc_avg(C.byref(p), float(g["visc"][i]), float(g["nvisc"][i]), float(g["rho"][i]),
                                               float(g["nrho"][i]), float(g["mass"][i])) for i in range(n)], dtype=np.float32)
                ref = g["va_%d%d%d" % (cv, av, cst)]
                if (cv, av, cst) == (0, 2, 1):      # 2 m rsqrt(rho rho'): rsqrt is not correctly rounded (host double 1/sqrt here,
                    np.testing.assert_allclose(got, ref, rtol=2.5e-7, atol=0)       # MUFU.RSQ on the device): one ulp
                else:
                    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (cv, av, cst, np.abs(got - ref).max())


def test_host_parameter_mirrors_match_reference_golden():
    """tests/golden/ref_hostparams.npz was produced by the reference's own PhysParams / SimParams (compiled into oracle/_ref):
    gpusph_amd/params.py must give the same defaults, equation-of-state and viscosity coefficients and radii."""
    from gpusph_amd.params import PhysParams, SimParams
    from gpusph_amd import defs as D
    g = np.load(os.path.join(GOLD, "ref_hostparams.npz"))
    for c, out in zip(g["phys_in"], g["phys"]):
        pp = PhysParams()
        f = pp.add_fluid(float(c[0]))
        pp.set_equation_of_state(f, float(c[1]), float(c[2]))
        pp.set_kinematic_visc(f, float(c[3]))
        h = pp.add_fluid(float(c[0]))
        pp.set_dynamic_visc(h, float(c[4]))
        mine = [pp.bcoeff[f], pp.gammacoeff[f], pp.sscoeff[f], pp.sspowercoeff[f], pp.kinematicvisc[f], pp.visc_consistency[f],
                pp.kinematicvisc[h], pp.visc_consistency[h], pp.artvisccoeff, pp.p1coeff, pp.p2coeff, pp.MK_beta, pp.partsurf,
                pp.smagorinsky_constant, pp.isotropic_sps_constant, pp.cosconeanglefluid, pp.cosconeanglenonfluid, pp.gravity[2]]
        assert np.array_equal(np.array(mine, dtype=np.float32).view(np.uint32), out.view(np.uint32)), (c, mine, out)
    for c, out in zip(g["sim_in"], g["sim"]):
        sp = SimParams()
        if int(c[0]):
            sp.kerneltype = D.GAUSSIAN
            sp.kernelradius = 3.0
        defaults = [sp.sfactor, sp.kernelradius, sp.buildneibsfreq, sp.dtadaptfactor, sp.repack_maxiter, sp.repack_a,
                    sp.repack_alpha, sp.nlexpansionfactor]
        assert np.array_equal(np.array(defaults, dtype=np.float32), out[:8].astype(np.float32)), (defaults, out[:8])
        sp.set_smoothing(float(c[1]), float(c[2]))
        assert [sp.slength, sp.influenceRadius, sp.nlInfluenceRadius, sp.nlSqInfluenceRadius] == list(out[8:12])


def test_small_arithmetic_and_policies_match_reference_golden():
    """tests/golden/ref_misc.npz was produced by the reference's own src/vector_math.h, src/utils.h and
    src/predcorr_alloc_policy.cc (compiled into oracle/_ref):
     * float4/float multiplies by the reciprocal (the operation order of the filters and of MLS), dot3, length
     * div_up / round_up, i.e. the arithmetic of getFmaxElements / round_particles (src/cuda/forces.cu:539-552,960-964)
     * the predictor-corrector scheme keeps two copies of POS and VEL and one of everything else the drivers allocate"""
    import ctypes as C
    g = np.load(os.path.join(GOLD, "ref_misc.npz"))
    L = ol.lib()
    for v, s, want in zip(g["v"], g["s"], g["v_over_s"]):
        out = (C.c_float * 4)()
        L.orc_f4_div(np.ascontiguousarray(v).ctypes.data, float(s), out)
        assert np.array_equal(np.array(list(out), dtype=np.float32).view(np.uint32), want.view(np.uint32))
        assert np.array_equal((v * (np.float32(1.0) / s)).view(np.uint32), want.view(np.uint32))
    d3 = (g["v"][:, 0] * g["v"][:, 0] + g["v"][:, 1] * g["v"][:, 1]) + g["v"][:, 2] * g["v"][:, 2]
    assert np.allclose(d3, g["dot3"], rtol=2e-7)
    from gpusph_amd import capi
    lib = capi.load()
    for (a, b), du, ru in zip(g["ab"], g["div_up"], g["round_up"]):
        assert du == (int(a) + int(b) - 1) // int(b) and ru == du * int(b)
    for n in (0, 1, 127, 128, 129, 100000, 31844148):
        k = list(map(tuple, g["ab"])).index((n, 128)) if (n, 128) in set(map(tuple, g["ab"])) else None
        if k is not None:     # getFmaxElements = round_up(div_up(n, 128), 4)
            want = (int(g["div_up"][k]) + 3) // 4 * 4
            assert lib.sphx_forces_fmax_elements(n) == want == L.orc_fmax_elements(n)
    keys = {name: int(k) for name, k in zip(
        ["POS", "VEL", "INFO", "HASH", "PARTINDEX", "CELLSTART", "CELLEND", "NEIBSLIST", "FORCES", "TAU", "CFL", "XSPH",
         "RB_FORCES", "SPS_TURBVISC", "VORTICITY", "NORMALS", "COMPACT_DEV_MAP", "CFL_TEMP", "RB_TORQUES", "RB_KEYS"], g["buffer_keys"])}
    counts = dict(zip(keys, (int(c) for c in g["predcorr_buffer_count"])))
    assert counts["POS"] == 2 and counts["VEL"] == 2
    assert all(c == 1 for name, c in counts.items() if name not in ("POS", "VEL"))
    assert len(set(keys.values())) == len(keys) and all(k and (k & (k - 1)) == 0 for k in keys.values())   # distinct single bits


def test_metric_definition_of_the_reference():
    """IPPSCounter (src/timing.h:103-164, compiled into oracle/_ref) is the metric bench.py reports: the sum over the timed
    iterations of the particle count, divided by the wall time; MIPPS = that / 1e6"""
    import ctypes as C
    ref = ol.ref()
    if ref is None:
        pytest.skip("oracle/_ref not built")
    out = (C.c_double * 2)()
    ref.ref_ipps(1_000_000, 20, 200, out)
    mipps, elapsed = out[0], out[1]
    assert 0.19 < elapsed < 1.0
    assert abs(mipps * elapsed - 20.0) < 0.02 * 20.0        # 20 iterations x 1e6 particles / elapsed / 1e6


def test_gamma_quadrature_matches_reference():
    """src/cuda/gamma.cuh (compiled into oracle/_ref from where it lies) against the oracle's restatement, bit for bit:
    the integrated Wendland kernel, the 5th-order Gauss rule on a triangle, the vertex frame of a segment, the analytical
    grad gamma of a segment and gamma for fluid and vertex particles (incl. the solid-angle branch for a particle on a vertex)"""
    import ctypes as C
    d = np.load(os.path.join(GOLD, "ref_gamma.npz"))
    L = ol.lib()
    F3 = C.c_float * 3
    h = float(d["h"])
    got = np.array([L.orc_wendland_on_segment(float(x)) for x in d["qs"]], dtype=np.float32)
    assert np.array_equal(got.view(np.uint32), d["wendland_on_segment"].view(np.uint32))
    assert d["wendland_on_segment"][-2] == 0.0 and d["wendland_on_segment"][0] > 1.0       # q = 2: outside; q -> 0: the 1/q^3 pole
    n = len(d["q"])
    on_vertex = 0
    for i in range(n):
        out = (C.c_float * 9)()
        vp = d["vp"][i]
        L.orc_calc_vertex_rel_pos(F3(*d["ns"][i]), (C.c_float * 2)(*vp[0]), (C.c_float * 2)(*vp[1]), (C.c_float * 2)(*vp[2]), h, out)
        qvb = np.array(list(out), dtype=np.float32)
        assert np.array_equal(qvb.view(np.uint32), d["q_vb"][i].view(np.uint32)), i
        v = [F3(*(-qvb[3 * k:3 * k + 3])) for k in range(3)]
        f32 = lambda x: np.float32(x).view(np.uint32)
        assert f32(L.orc_gauss_quadrature_O5(v[0], v[1], v[2], F3(*d["q"][i]))) == f32(d["gauss_quadrature_O5"][i]), i
        assert f32(L.orc_grad_gamma(h, F3(*d["q"][i]), out, F3(*d["ns"][i]))) == f32(d["grad_gamma"][i]), i
        assert f32(L.orc_gamma(0, h, F3(*d["q"][i]), out, F3(*d["ns"][i]), F3(*d["ggam"][i]), 5e-5)) == f32(d["gamma_fluid"][i]), i
        assert f32(L.orc_gamma(1, h, F3(*d["qv"][i]), out, F3(*d["ns"][i]), F3(*d["ggam"][i]), 5e-5)) == f32(d["gamma_vertex"][i]), i
        on_vertex += (i % 8 == 0) and d["gamma_vertex"][i] != 0
    assert on_vertex > 30          # the solid-angle branch was really exercised
    ref = ol.ref()                 # and live, when the reference build is present
    if ref is not None:
        for i in range(0, n, 7):
            out = (C.c_float * 9)(*d["q_vb"][i])
            assert np.float32(ref.ref_gradGamma(h, F3(*d["q"][i]), out, F3(*d["ns"][i]))) == d["grad_gamma"][i]
