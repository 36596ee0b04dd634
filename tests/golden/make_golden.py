#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/.

  ref_kernels.npz, ref_datamodel.npz   outputs of the REFERENCE's own code (oracle/_ref, compiled here
                                       from /root/reference/src/cuda/sph_core.cu, src/particleinfo.h, ...)
  oracle_pipeline.npz                  outputs of the CPU oracle on a small DamBreak3D (regression data for
                                       the GPU path; the oracle itself is unpinned for these stages)

Run in the build container (needs /root/reference for oracle/_ref):  python tests/golden/make_golden.py
"""
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import oracle_lib as ol  # noqa: E402
from gpusph_amd.problem import DamBreak3D  # noqa: E402


def kernels():
    ref = ol.ref()
    assert ref is not None, "oracle/_ref not built (needs /root/reference)"
    L = ol.lib()
    rng = np.random.default_rng(2024)
    rows = []
    for k in (1, 2, 3, 4):
        kr = 3.0 if k == 4 else 2.0
        for h in (np.float32(0.013), np.float32(0.0052), np.float32(0.00208), np.float32(0.39)):
            wc = np.float32(L.orc_wcoeff(k, float(h), kr)); fc = np.float32(L.orc_fcoeff(k, float(h), kr))
            wsub = np.float32(np.exp(np.float32(-kr * kr)))
            rs = np.concatenate([rng.uniform(1e-4, kr, 40).astype(np.float32) * h,
                                 np.array([0.5, 1.0, 1.5, 1.9999, 2.0], dtype=np.float32) * h])
            for r in rs:
                rows.append((k, float(r), float(h), float(wc), float(fc), float(wsub),
                             ref.ref_W(k, float(r), float(h), float(wc), float(wsub)),
                             ref.ref_F(k, float(r), float(h), float(fc))))
    a = np.array(rows, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "ref_kernels.npz"), kerneltype=a[:, 0].astype(np.int32),
                        r=a[:, 1].astype(np.float32), slength=a[:, 2].astype(np.float32),
                        wcoeff=a[:, 3].astype(np.float32), fcoeff=a[:, 4].astype(np.float32),
                        wsub=a[:, 5].astype(np.float32), W=a[:, 6].astype(np.float32), F=a[:, 7].astype(np.float32))


def datamodel():
    ref = ol.ref()
    rng = np.random.default_rng(99)
    info = rng.integers(0, 65536, size=(512, 4), dtype=np.uint16)
    info[:64, 0] = np.arange(64, dtype=np.uint16)          # all type/flag low-bit combinations
    ids = np.array([ref.ref_info_id(*map(int, i)) for i in info], dtype=np.uint32)
    ptype = np.array([ref.ref_info_part_type(*map(int, i)) for i in info], dtype=np.int32)
    obj = np.array([ref.ref_info_object(*map(int, i)) for i in info], dtype=np.int32)
    fl = np.array([ref.ref_info_fluid_num(*map(int, i)) for i in info], dtype=np.int32)
    pred = np.array([ref.ref_info_predicates(*map(int, i)) for i in info], dtype=np.uint32)
    hashes = np.concatenate([rng.integers(0, 2**32, size=64, dtype=np.uint64).astype(np.uint32),
                             np.array([0, 1, 0x3FFFFFFF, 0x40000000, 0x80000000, 0xC0000000, 0xFFFFFFFF], dtype=np.uint32)])
    h_reset = np.array([ref.ref_cell_hash_from_particle_hash(int(h), 0) for h in hashes], dtype=np.uint32)
    h_keep = np.array([ref.ref_cell_hash_from_particle_hash(int(h), 1) for h in hashes], dtype=np.uint32)
    cells = np.arange(27, dtype=np.uint32)
    enc = np.array([ref.ref_encode_cell(int(c)) for c in cells], dtype=np.uint32)
    dec = np.array([ref.ref_decode_cell(int(e) + 37) for e in enc], dtype=np.int32)
    consts = np.array([ref.ref_constant(i) for i in range(11)], dtype=np.uint32)
    enums = np.array([ref.ref_enum(i) for i in range(40)], dtype=np.int32)
    wvals = np.array([0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-45], dtype=np.float32)
    active = np.array([ref.ref_active(float(w)) for w in wvals], dtype=np.int32)
    np.savez_compressed(os.path.join(HERE, "ref_datamodel.npz"), info=info, id=ids, ptype=ptype, object=obj, fluid=fl,
                        predicates=pred, hashes=hashes, hash_reset=h_reset, hash_keep=h_keep, cells=cells,
                        encoded=enc, decoded=dec, constants=consts, enums=enums, wvals=wvals, active=active)


def pipeline():
    prob = DamBreak3D(0.05, obstacle=True, jitter=0.05, hydrostatic=False)   # off the Colagrossi switch, see tests/test_gpu_parity.py
    sim = ol.OracleSim(prob)
    arrs = prob.copy_to_array()
    out = {"in_pos": arrs["pos"], "in_vel": arrs["vel"], "in_info": arrs["info"], "in_hash": arrs["hash"]}
    sim.step()
    out.update(s1_hash=sim.hash.copy(), s1_info=sim.info.copy(), s1_partindex=sim.partindex.copy(),
               s1_cellStart=sim.cs.copy(), s1_cellEnd=sim.ce.copy(), s1_neibs=sim.nl.copy(),
               s1_numInteractions=np.int64(sim.neibs_info.numInteractions),
               s1_maxneibs=np.int64(sim.neibs_info.maxFluidBoundaryNeibs),
               s1_pos=sim.pos.copy(), s1_vel=sim.vel.copy(), s1_forces=sim.forces.copy(), s1_dt=np.float32(sim.dt))
    for _ in range(10):
        sim.step()
    out.update(s11_hash=sim.hash.copy(), s11_info=sim.info.copy(), s11_pos=sim.pos.copy(), s11_vel=sim.vel.copy(),
               s11_dt=np.float32(sim.dt), deltap=np.float32(0.05))
    np.savez_compressed(os.path.join(HERE, "oracle_pipeline.npz"), **out)


def features():
    """SURVEY 8c fixtures (iv), (v) and the widened rows: SPS tau, moving-body Euler rows, density filters,
    post-processing, LJ + planes forces -- inputs are rebuilt from the problem mirror, outputs are stored."""
    from gpusph_amd import defs as D
    out = {}
    # --- one DYN problem with an obstacle and test points: SPS, filters, post-processing, moving-body Euler
    pts = [(0.2, 0.3, 0.2), (0.3, 0.4, 0.1)]
    prob = DamBreak3D(0.05, obstacle=True, jitter=0.1, hydrostatic=False, testpoints=pts)
    prob.simparams.turbmodel = D.SPS
    dp = prob.m_deltap
    prob.physparams.smagfactor = float(np.float32((0.12 * dp) ** 2))
    prob.physparams.kspsfactor = float(np.float32((2.0 / 3.0) * 0.0066 * dp * dp))
    sim = ol.OracleSim(prob)
    sim.build_neibs()
    n = sim.n
    rng = np.random.default_rng(77)
    vel = sim.vel.copy()
    vel[:, :3] += rng.uniform(-0.4, 0.4, size=(len(vel), 3)).astype(np.float32)
    vel[:, 3] += rng.uniform(0, 2e-3, size=len(vel)).astype(np.float32)
    out["a_vel"] = vel
    tau, tv = sim.o.sps(sim.pos, vel, sim.info, sim.hash, sim.cs, sim.nl, n, n)
    out["a_tau"] = tau; out["a_turbvisc"] = tv
    out["a_forces_sps"] = sim.o.forces(sim.pos, vel, sim.info, sim.hash, sim.cs, sim.nl, n, tau=tau)[0]
    out["a_shepard"] = sim.o.filter(0, sim.pos, vel, sim.info, sim.hash, sim.cs, sim.nl, n)
    out["a_mls"] = sim.o.filter(1, sim.pos, vel, sim.info, sim.hash, sim.cs, sim.nl, n)
    out["a_vorticity"] = sim.o.vorticity(sim.pos, vel, sim.info, sim.hash, sim.cs, sim.nl, n)
    info_s, nrm = sim.o.surface(sim.pos, vel, sim.info, sim.hash, sim.cs, sim.nl, n, normals=True)
    out["a_surface_info"] = info_s; out["a_normals"] = nrm
    out["a_testpoints"] = sim.o.testpoints(sim.pos, vel, sim.info, sim.hash, sim.cs, sim.nl, n)
    th = 0.01
    rot = np.array([np.cos(th), -np.sin(th), 0, np.sin(th), np.cos(th), 0, 0, 0, 1], dtype=np.float32)
    motion = dict(trans=np.array([1e-3, -2e-3, 5e-4], np.float32), rot=rot, lvel=np.array([0.3, -0.1, 0.05], np.float32),
                  avel=np.array([0.0, 0.2, 1.5], np.float32))
    for a in range(3):
        sim.o.p.rbtrans[0][a] = float(motion["trans"][a]); sim.o.p.rblinearvel[0][a] = float(motion["lvel"][a])
        sim.o.p.rbangularvel[0][a] = float(motion["avel"][a])
    for a in range(9):
        sim.o.p.rbsteprot[0][a] = float(rot[a])
    frc = rng.normal(0, 5, size=(len(sim.pos), 4)).astype(np.float32)
    out["a_euler_forces"] = frc
    out.update({"a_rb_" + k: v for k, v in motion.items()})
    out["a_euler_dt"] = np.float32(2.7e-4)
    for step, scale in ((1, 0.5), (2, 1.0)):
        pr, vr = sim.o.euler(sim.pos, vel, sim.info, sim.hash, frc, n, float(np.float32(2.7e-4) * np.float32(scale)), step)
        out["a_euler%d_pos" % step] = pr; out["a_euler%d_vel" % step] = vr
    out["a_deltap"] = np.float32(0.05)
    # --- LJ boundary particles + a feedback body; planes instead of walls
    for tag, kw in (("b", dict(obstacle=True, boundary=D.LJ_BOUNDARY)), ("c", dict(obstacle=False, boundary=D.LJ_BOUNDARY, walls="planes"))):
        prob = DamBreak3D(0.05, jitter=0.25, hydrostatic=False, **kw)
        sim = ol.OracleSim(prob)
        sim.build_neibs()
        n = sim.n
        vel = sim.vel.copy()
        fl = (sim.info[:, 0] & 7) == 0
        vel[fl, :3] += rng.uniform(-0.3, 0.3, size=(fl.sum(), 3)).astype(np.float32)
        vel[fl, 3] += rng.uniform(0, 2e-3, size=fl.sum()).astype(np.float32)
        cof = 1 if prob.simparams.numforcesbodies else 0
        f, cfl, nb, rbf, rbt = sim.o.forces(sim.pos, vel, sim.info, sim.hash, sim.cs, sim.nl, n, compute_object_forces=cof,
                                            rb_count=prob.num_obstacle)
        out[tag + "_vel"] = vel; out[tag + "_forces"] = f; out[tag + "_neibs"] = sim.nl.copy()
        out[tag + "_dt"] = np.float32(sim.o.dtreduce(cfl, nb, sim.sspeed_cfl))
        if prob.num_obstacle:
            out[tag + "_rbforces"] = rbf
    np.savez_compressed(os.path.join(HERE, "oracle_features.npz"), **out)


def viscavg():
    """the reference's own visc_avg<FullViscSpec<...>> (src/cuda/visc_avg.cu compiled as it is into oracle/_ref) on a
    grid of viscosities, densities and masses, for the 12 (computational viscosity, average, constness) flavours"""
    ref = ol.ref()
    rng = np.random.default_rng(99)
    n = 400
    visc = rng.uniform(1e-6, 0.3, n).astype(np.float32); nvisc = rng.uniform(1e-6, 0.3, n).astype(np.float32)
    rho = rng.uniform(800, 1100, n).astype(np.float32); nrho = rng.uniform(800, 1100, n).astype(np.float32)
    mass = rng.uniform(1e-6, 0.2, n).astype(np.float32)
    out = dict(visc=visc, nvisc=nvisc, rho=rho, nrho=nrho, mass=mass)
    for cv in (0, 1):
        for av in (0, 1, 2):
            for cst in (0, 1):
                out["va_%d%d%d" % (cv, av, cst)] = np.array(
                    [ref.ref_visc_avg(cv, av, cst, float(visc[i]), float(nvisc[i]), float(rho[i]), float(nrho[i]), float(mass[i]))
                     for i in range(n)], dtype=np.float32)
    for av in (0, 1, 2):     # single-fluid framework forced to non-constant kinematic viscosity (with_computational_visc quirk)
        out["va_single_0%d0" % av] = np.array(
            [ref.ref_visc_avg_singlefluid_nonconst_kinematic(av, float(visc[i]), float(nvisc[i]), float(rho[i]), float(nrho[i]), float(mass[i]))
             for i in range(n)], dtype=np.float32)
    np.savez_compressed(os.path.join(HERE, "ref_viscavg.npz"), **out)


def hostparams():
    """the reference's own PhysParams / SimParams (src/physparams.h, src/simparams.h compiled into oracle/_ref): defaults,
    equation-of-state and viscosity setters, smoothing length and influence radii"""
    ref = ol.ref()
    cases = [(1000.0, 7.0, 20.0, 1.0e-2, 1.0e-3), (850.0, 7.0, 22.0, 3.0e-2, 0.5), (1.0, 1.4, 340.0, 1.5e-5, 1.8e-5)]
    phys = np.zeros((len(cases), 18), dtype=np.float32)
    for k, c in enumerate(cases):
        ref.ref_physparams(*[float(np.float32(x)) for x in c], phys[k].ctypes.data)
    sims = [(0, 1.3, 0.0159), (0, 1.3, 0.04), (1, 1.3, 0.05), (0, 1.5, 0.001590)]
    sim = np.zeros((len(sims), 12), dtype=np.float64)
    for k, c in enumerate(sims):
        ref.ref_simparams(c[0], c[1], c[2], sim[k].ctypes.data)
    np.savez_compressed(os.path.join(HERE, "ref_hostparams.npz"), phys_in=np.array(cases, dtype=np.float32), phys=phys,
                        sim_in=np.array(sims, dtype=np.float64), sim=sim)


def features2_cases():
    """(tag, problem factory) of the second feature fixture: repacking run mode and Newtonian viscosity.  Shared by
    the generator and the tests so that both rebuild the same inputs."""
    from gpusph_amd import defs as D
    newt = lambda **kw: dict(rheologytype=D.NEWTONIAN, turbmodel=D.LAMINAR_FLOW, **kw)
    return [
        ("r", "repack", lambda: DamBreak3D(0.05, obstacle=True, jitter=0.2, hydrostatic=True)),
        ("k", "forces", lambda: DamBreak3D(0.05, obstacle=True, jitter=0.15, hydrostatic=False, viscosity="KINEMATICVISC",
                                           kinematic_visc=0.05)),
        ("m", "forces", lambda: DamBreak3D(0.05, obstacle=False, jitter=0.15, hydrostatic=False, two_fluids=True, kinematic_visc=0.05,
                                           viscosity=newt(compvisc=D.KINEMATIC, avgop=D.HARMONIC, is_const_visc=False))),
        ("g", "forces", lambda: DamBreak3D(0.05, obstacle=False, jitter=0.15, hydrostatic=False, two_fluids=True, kinematic_visc=0.05,
                                           density_diffusion=D.DENSITY_DIFFUSION_NONE,
                                           viscosity=newt(compvisc=D.DYNAMIC, avgop=D.GEOMETRIC, is_const_visc=False))),
        ("w", "forces", lambda: DamBreak3D(0.05, obstacle=False, jitter=0.3, hydrostatic=False, boundary=D.LJ_BOUNDARY, walls="planes",
                                           viscosity="KINEMATICVISC", kinematic_visc=0.2)),
    ]


def features2():
    out = {}
    rng = np.random.default_rng(78)
    for tag, kind, make in features2_cases():
        prob = make()
        sim = ol.OracleSim(prob)
        sim.build_neibs()
        n = sim.n
        vel = sim.vel.copy()
        fl = (sim.info[:, 0] & 7) == 0
        vel[fl, :3] += rng.uniform(-0.3, 0.3, size=(fl.sum(), 3)).astype(np.float32)
        if kind == "forces":
            vel[:, 3] += rng.uniform(0, 2e-3, size=len(vel)).astype(np.float32)
        out[tag + "_vel"] = vel
        rb = getattr(prob, "num_obstacle", 0)
        if kind == "repack":
            f, cfl, nb, _, _ = sim.o.repack_forces(sim.pos, vel, sim.info, sim.hash, sim.cs, sim.nl, n, rb_count=rb)
            pr, vr = sim.o.euler_repack(sim.pos, vel, sim.info, sim.hash, f, n, float(np.float32(1.3e-4)), 1)
            out[tag + "_euler_pos"] = pr; out[tag + "_euler_vel"] = vr
        else:
            cof = 1 if prob.simparams.numforcesbodies else 0
            f, cfl, nb, _, _ = sim.o.forces(sim.pos, vel, sim.info, sim.hash, sim.cs, sim.nl, n, compute_object_forces=cof, rb_count=rb)
        out[tag + "_forces"] = f
        out[tag + "_dt"] = np.float32(sim.o.dtreduce(cfl, nb, sim.sspeed_cfl, sim.max_kinvisc))
    np.savez_compressed(os.path.join(HERE, "oracle_features2.npz"), **out)


def refmisc():
    """ref_misc.npz: small arithmetic and policy facts of the reference's own code (oracle/_ref): float4/float of src/vector_math.h, div_up/round_up of src/utils.h, the predictor-corrector
    buffer allocation policy, buffer keys"""
    import ctypes as C
    ref = ol.ref()
    rng = np.random.default_rng(31)
    v = rng.normal(0, 3, size=(200, 4)).astype(np.float32)
    sdiv = rng.uniform(0.01, 7, 200).astype(np.float32)
    q = np.zeros((200, 5), dtype=np.float32); d3 = np.zeros(200, dtype=np.float32)
    for i in range(200):
        buf = (C.c_float * 5)()
        d3[i] = ref.ref_float4_div(*[float(x) for x in v[i]], float(sdiv[i]), buf)
        q[i] = list(buf)
    ab = np.array([(a, b) for a in (0, 1, 127, 128, 129, 511, 512, 513, 100000, 31844148) for b in (4, 128, 256)], dtype=np.uint32)
    du = np.array([ref.ref_div_up(int(a), int(b)) for a, b in ab], dtype=np.uint32)
    ru = np.array([ref.ref_round_up(int(a), int(b)) for a, b in ab], dtype=np.uint32)
    keys = np.array([ref.ref_buffer_key(k) for k in range(20)], dtype=np.uint64)
    counts = np.array([ref.ref_predcorr_buffer_count(int(k)) for k in keys], dtype=np.uint32)
    np.savez_compressed(os.path.join(HERE, "ref_misc.npz"), v=v, s=sdiv, v_over_s=q[:, :4], length3=q[:, 4], dot3=d3, ab=ab, div_up=du, round_up=ru,
                        buffer_keys=keys, predcorr_buffer_count=counts)


def gamma_inputs():
    """inputs for the gamma.cuh pin: right-angled and random triangles of side ~deltap around the origin (as q_vb: vertex
    positions relative to the barycentre over h), particles at random offsets within the kernel support, plus the special
    positions the vertex specialisation tests for (on a vertex, in the plane)"""
    rng = np.random.default_rng(77)
    h = np.float32(0.065)
    n = 400
    ns = rng.normal(size=(n, 3)); ns[:60] = np.eye(3)[rng.integers(0, 3, 60)] * rng.choice([-1, 1], (60, 1))
    ns = (ns / np.linalg.norm(ns, axis=1, keepdims=True)).astype(np.float32)
    vp = rng.uniform(-0.05, 0.05, size=(n, 3, 2))
    vp = (vp - vp.mean(axis=1, keepdims=True)).astype(np.float32)          # offsets about the barycentre
    q = (rng.normal(size=(n, 3)) * rng.uniform(0.0, 1.6, (n, 1))).astype(np.float32)
    ggam = rng.normal(size=(n, 3)).astype(np.float32)
    return h, ns, vp, q, ggam


def refgamma():
    """ref_gamma.npz: src/cuda/gamma.cuh as compiled into oracle/_ref (host math library): wendlandOnSegment,
    gaussQuadratureO5, calcVertexRelPos, gradGamma<WENDLAND>, Gamma<WENDLAND, PT_FLUID|PT_VERTEX>"""
    import ctypes as C
    ref = ol.ref()
    h, ns, vp, q, ggam = gamma_inputs()
    n = len(q)
    F3 = C.c_float * 3
    qs = np.concatenate([np.linspace(0.01, 2.2, 120), [2.0, 1e-3]]).astype(np.float32)
    wos = np.array([ref.ref_wendlandOnSegment(float(x)) for x in qs], dtype=np.float32)
    qvb = np.zeros((n, 9), dtype=np.float32)
    gq = np.zeros(n, dtype=np.float32); gg = np.zeros(n, dtype=np.float32); gaf = np.zeros(n, dtype=np.float32); gav = np.zeros(n, dtype=np.float32)
    qv = q.copy()
    for i in range(n):
        out = (C.c_float * 9)()
        ref.ref_calcVertexRelPos(F3(*ns[i]), (C.c_float * 2)(*vp[i, 0]), (C.c_float * 2)(*vp[i, 1]), (C.c_float * 2)(*vp[i, 2]), float(h), out)
        qvb[i] = list(out)
        if i % 8 == 0:      # a particle ON a vertex of the element (the vertex specialisation's solid-angle branch)
            qv[i] = qvb[i, 3 * (i // 8 % 3):3 * (i // 8 % 3) + 3]
        v = [F3(*(-qvb[i, 3 * k:3 * k + 3])) for k in range(3)]
        gq[i] = ref.ref_gaussQuadratureO5(v[0], v[1], v[2], F3(*q[i]))
        gg[i] = ref.ref_gradGamma(float(h), F3(*q[i]), out, F3(*ns[i]))
        gaf[i] = ref.ref_Gamma(0, float(h), F3(*q[i]), out, F3(*ns[i]), F3(*ggam[i]), 5e-5)
        gav[i] = ref.ref_Gamma(1, float(h), F3(*qv[i]), out, F3(*ns[i]), F3(*ggam[i]), 5e-5)
    np.savez_compressed(os.path.join(HERE, "ref_gamma.npz"), h=h, ns=ns, vp=vp, q=q, qv=qv, ggam=ggam, qs=qs, wendland_on_segment=wos,
                        q_vb=qvb, gauss_quadrature_O5=gq, grad_gamma=gg, gamma_fluid=gaf, gamma_vertex=gav)


if __name__ == "__main__":
    kernels(); datamodel(); viscavg(); hostparams(); refmisc(); refgamma(); pipeline(); features(); features2()
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))
