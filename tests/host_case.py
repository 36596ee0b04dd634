"""Case files for the two host programs built against the GPUSPH tree (gpusph_amd/host/framework_check,
example_engines): the parameters of a Python problem mirror written as "key value..." lines (see
gpusph_amd/host/problem_setup.h).  Test infrastructure."""
import math
import os
import subprocess
import numpy as np

from gpusph_amd import defs as D

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_DIR = os.path.join(ROOT, "gpusph_amd", "host")


def exe(name):
    """path of a prebuilt host program; builds it when the GPUSPH tree is here and it is missing"""
    path = os.path.join(HOST_DIR, name)
    if not os.path.exists(path) and os.path.isdir("/root/reference/src"):
        subprocess.check_call(["make", "-s", "-C", HOST_DIR, name])
    return path


def _g(v):
    return "%.17g" % float(v)


def _nz(v):
    return float("nan") if v is None else v        # unset coefficients travel as "nan"


def case_lines(prob, framework, allocated=None, **selectors):
    """parameter lines for `prob` (after Problem.initialize), framework = name of a SETUP_FRAMEWORK expression in
    problem_setup.h, selectors = its run-time arguments (rhodiff=..., use_planes=...)"""
    sp, pp = prob.simparams, prob.physparams
    L = ["framework %s" % framework]
    for k, v in selectors.items():
        L.append("%s %d" % (k, int(v)))
    L += ["deltap %s" % _g(prob.m_deltap), "sfactor %s" % _g(sp.sfactor), "kernelradius %s" % _g(sp.kernelradius),
          "neiblistsize %d" % sp.neiblistsize, "neibboundpos %d" % sp.neibboundpos,
          "dtadaptfactor %s" % _g(np.float32(sp.dtadaptfactor)),
          "densityDiffCoeff %s" % _g(_nz(sp.densityDiffCoeff)),
          "repack_a %s" % _g(sp.repack_a), "repack_alpha %s" % _g(sp.repack_alpha),
          "buildneibsfreq %d" % sp.buildneibsfreq, "numbodies %d %d" % (sp.numbodies, sp.numforcesbodies),
          "nfluids %d" % pp.numFluids()]
    for f in range(pp.numFluids()):
        nu = pp.kinematicvisc[f]
        kind = "none" if (nu is None or math.isnan(nu)) else "kin"
        L.append("fluid%d %s %s %s %s %s" % (f, _g(pp.rho0[f]), _g(pp.gammacoeff[f]), _g(pp.sscoeff[f]), kind,
                                            _g(0.0 if kind == "none" else nu)))
        if sp.viscmodel == D.ESPANOL_REVENGA:
            L.append("bulkvisc%d %s" % (f, _g(pp.bulkvisc[f] if not math.isnan(pp.bulkvisc[f]) else 0.0)))
        if sp.rheologytype > D.NEWTONIAN:     # generalized Newtonian: yield strength; power-law / exponential parameter and m when set
            nl = pp.visc_nonlinear_param[f]
            default_nl = 0.0 if sp.rheologytype >= D.DEKEE_TURCOTTE else 1.0
            L.append("rheology%d %s %s %s" % (f, _g(pp.yield_strength[f]), _g(float("nan") if nl == default_nl else nl),
                                              _g(float("nan") if pp.visc_regularization_param[f] == 1000.0 else pp.visc_regularization_param[f])))
    L += ["gravity %s %s %s" % tuple(_g(x) for x in pp.gravity), "artvisccoeff %s" % _g(pp.artvisccoeff),
          "epsartvisc %s" % _g(pp.epsartvisc), "r0 %s" % _g(_nz(pp.r0)), "dcoeff %s" % _g(_nz(pp.dcoeff)),
          "p1coeff %s" % _g(pp.p1coeff), "p2coeff %s" % _g(pp.p2coeff),
          "smagfactor %s" % _g(_nz(pp.smagfactor)), "kspsfactor %s" % _g(_nz(pp.kspsfactor)),
          "MK_K %s" % _g(_nz(pp.MK_K)), "MK_d %s" % _g(_nz(pp.MK_d)), "MK_beta %s" % _g(_nz(pp.MK_beta)),
          "partsurf %s" % _g(pp.partsurf), "epsinterface %s" % _g(_nz(pp.epsinterface)),
          *(["demparams %s %s %s %s %s" % tuple(_g(v) for v in (pp.ewres, pp.nsres, pp.demdx, pp.demdy, pp.demzmin))]
            if (sp.simflags & D.ENABLE_DEM) else []), "epsxsph %s" % _g(getattr(pp, "epsxsph", sp.epsxsph)),
          "origin %s %s %s" % tuple(_g(x) for x in prob.m_origin), "grid %d %d %d" % tuple(int(x) for x in prob.m_gridsize),
          "cell %s %s %s" % tuple(_g(x) for x in prob.m_cellsize),
          "allocated %d" % int(allocated if allocated is not None else prob.num_particles)]
    return L


def driver_lines(prob, eng, steps, filters=(), final_surface=False):
    """what example_engines needs on top of case_lines: run control, planes, bodies, filters"""
    L = ["steps %d" % steps, "dt0 %s" % _g(np.float32(eng.dt)), "sspeed_cfl %s" % _g(np.float32(eng.sspeed_cfl)),
         "max_kinvisc %s" % _g(np.float32(eng.max_kinvisc))]
    if getattr(prob, "planes", None):
        nrm, gpos, lpos = prob.plane_tables()
        vals = []
        for k in range(len(nrm)):
            vals += [_g(x) for x in nrm[k]] + ["%d" % x for x in gpos[k]] + [_g(x) for x in lpos[k]]
        L.append("plane " + " ".join(vals))
    if getattr(prob, "dem", None) is not None:
        dem = np.asarray(prob.dem, dtype=np.float32)
        L.append("dem %d %d " % (dem.shape[1], dem.shape[0]) + " ".join(_g(v) for v in dem.ravel()))
    nobj = getattr(prob, "num_obstacle", 0)
    if nobj:
        cg = getattr(prob, "rb_cg_global", None)
        vals = []
        for b in range(len(prob.rb_firstindex)):
            if cg is not None:
                c = np.asarray(cg[b], dtype=np.float64)
            else:
                c = prob.m_origin + (np.asarray(prob.rb_cg_gridpos[b]) + 0.5) * prob.m_cellsize + np.asarray(prob.rb_cg_pos[b], dtype=np.float64)
            if getattr(prob, "moving_bodies_callback", None) is not None and hasattr(prob, "paddle_amplitude"):
                motion = ["paddle_y", _g(prob.paddle_amplitude), _g(prob.paddle_omega), _g(prob.paddle_tstart), _g(prob.paddle_tend)]
            else:
                motion = ["static", "0", "0", "0", "0"]
            vals += ["%d" % int(prob.rb_firstindex[b]), "%d" % int(nobj)] + [_g(x) for x in c] + motion
        L.append("body " + " ".join(vals))
    if filters:
        L.append("filter " + " ".join("%d %d" % (int(t), int(fq)) for t, fq in filters))
    if final_surface:
        L.append("final_surface 1")
    return L


def write_state(path, arrs):
    n = len(arrs["hash"])
    with open(path, "wb") as f:
        f.write(np.uint32(n).tobytes())
        f.write(np.ascontiguousarray(arrs["pos"], dtype=np.float32).tobytes())
        f.write(np.ascontiguousarray(arrs["vel"], dtype=np.float32).tobytes())
        f.write(np.ascontiguousarray(arrs["info"]).view(np.uint16).tobytes())
        f.write(np.ascontiguousarray(arrs["hash"]).view(np.uint32).tobytes())
        if "vertices" in arrs:      # SA_BOUNDARY: BUFFER_VERTICES, BUFFER_BOUNDELEMENTS, BUFFER_GRADGAMMA
            f.write(np.ascontiguousarray(arrs["vertices"], dtype=np.uint32).tobytes())
            f.write(np.ascontiguousarray(arrs["boundelements"], dtype=np.float32).tobytes())
            f.write(np.ascontiguousarray(arrs["gradgamma"], dtype=np.float32).tobytes())


def read_out(path):
    raw = open(path, "rb").read()
    n = int(np.frombuffer(raw, np.uint32, 1, 0)[0])
    dt = np.frombuffer(raw, np.float32, 1, 4)[0]
    t = np.frombuffer(raw, np.float64, 1, 8)[0]
    o = 16
    pos = np.frombuffer(raw, np.float32, 4 * n, o).reshape(n, 4); o += 16 * n
    vel = np.frombuffer(raw, np.float32, 4 * n, o).reshape(n, 4); o += 16 * n
    info = np.frombuffer(raw, np.uint16, 4 * n, o).reshape(n, 4); o += 8 * n
    hsh = np.frombuffer(raw, np.uint32, n, o); o += 4 * n
    out = dict(n=n, dt=dt, t=t, pos=pos, vel=vel, info=info, hash=hsh)
    if len(raw) - o == 4 * n:       # ENABLE_INTERNAL_ENERGY: BUFFER_INTERNAL_ENERGY
        out["energy"] = np.frombuffer(raw, np.float32, n, o)
    elif len(raw) - o == 16 * n:    # SPH_GRENIER: BUFFER_VOLUME
        out["vol"] = np.frombuffer(raw, np.float32, 4 * n, o).reshape(n, 4)
    elif len(raw) > o:              # the SA initialisation run appends its buffers and the list counters
        out["vertices"] = np.frombuffer(raw, np.uint32, 4 * n, o).reshape(n, 4); o += 16 * n
        out["boundelements"] = np.frombuffer(raw, np.float32, 4 * n, o).reshape(n, 4); o += 16 * n
        out["gradgamma"] = np.frombuffer(raw, np.float32, 4 * n, o).reshape(n, 4); o += 16 * n
        out["vertpos"] = [np.frombuffer(raw, np.float32, 2 * n, o + 8 * n * k).reshape(n, 2) for k in range(3)]; o += 24 * n
        out["counters"] = np.frombuffer(raw, np.int32, 4, o); o += 16
        if len(raw) - o == 12 * n:  # turbulence_model<KEPSILON>: k, epsilon, eddy viscosity
            for k, name in enumerate(("tke", "eps", "turbvisc")):
                out[name] = np.frombuffer(raw, np.float32, n, o + 4 * n * k)
    return out
