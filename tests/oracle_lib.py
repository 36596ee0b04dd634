"""ctypes binding of the CPU oracle (oracle/libsph_oracle.so) and of oracle/_ref -- tests only."""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")


class OrcParams(C.Structure):
    _fields_ = [
        ("gridSize", C.c_uint32 * 3), ("cellSize", C.c_float * 3), ("worldOrigin", C.c_float * 3),
        ("coord", C.c_int32 * 3), ("periodic", C.c_uint32),
        ("neiblistsize", C.c_uint32), ("neibboundpos", C.c_uint32), ("neiblist_stride", C.c_uint64),
        ("kerneltype", C.c_int32), ("sph_formulation", C.c_int32), ("densitydiffusiontype", C.c_int32),
        ("boundarytype", C.c_int32), ("rheologytype", C.c_int32), ("turbmodel", C.c_int32),
        ("compvisc", C.c_int32), ("viscmodel", C.c_int32), ("avgop", C.c_int32),
        ("simflags", C.c_uint64),
        ("slength", C.c_float), ("influenceradius", C.c_float), ("deltap", C.c_float), ("dtadaptfactor", C.c_float),
        ("densityDiffCoeff", C.c_float), ("epsxsph", C.c_float),
        ("numfluids", C.c_uint32),
        ("rho0", C.c_float * 4), ("bcoeff", C.c_float * 4), ("gammacoeff", C.c_float * 4),
        ("sscoeff", C.c_float * 4), ("sspowercoeff", C.c_float * 4), ("visccoeff", C.c_float * 4),
        ("gravity", C.c_float * 3),
        ("artvisccoeff", C.c_float), ("epsartvisc", C.c_float),
        ("smagfactor", C.c_float), ("kspsfactor", C.c_float),
        ("dcoeff", C.c_float), ("p1coeff", C.c_float), ("p2coeff", C.c_float), ("r0", C.c_float),
        ("repack_a", C.c_float), ("repack_alpha", C.c_float),
        ("is_const_visc", C.c_int32), ("partsurf", C.c_float),
        ("MK_K", C.c_float), ("MK_d", C.c_float), ("MK_beta", C.c_float),
        ("epsinterface", C.c_float),
        ("yield_strength", C.c_float * 4), ("visc_nonlinear_param", C.c_float * 4),
        ("visc_regularization_param", C.c_float * 4), ("limiting_kinvisc", C.c_float),
        ("ewres", C.c_float), ("nsres", C.c_float), ("demdx", C.c_float), ("demdy", C.c_float), ("demzmin", C.c_float),
        ("monaghan_visc_coeff", C.c_float), ("visc2coeff", C.c_float * 4),
        ("numplanes", C.c_uint32),
        ("plane_normal", (C.c_float * 3) * 8), ("plane_gridpos", (C.c_int32 * 3) * 8), ("plane_pos", (C.c_float * 3) * 8),
        ("rbcgGridPos", (C.c_int32 * 3) * 16), ("rbcgPos", (C.c_float * 3) * 16), ("rbstartindex", C.c_int32 * 16),
        ("rbtrans", (C.c_float * 3) * 16), ("rbsteprot", (C.c_float * 9) * 16),
        ("rblinearvel", (C.c_float * 3) * 16), ("rbangularvel", (C.c_float * 3) * 16),
        ("rbcgGridPosE", (C.c_int32 * 3) * 16), ("rbcgPosE", (C.c_float * 3) * 16),
    ]


class OrcNeibsInfo(C.Structure):
    _fields_ = [("numInteractions", C.c_int32), ("maxFluidBoundaryNeibs", C.c_int32),
                ("maxVertexNeibs", C.c_int32), ("hasTooManyNeibs", C.c_int32), ("hasMaxNeibs", C.c_int32 * 3)]


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR], stdout=subprocess.DEVNULL)


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(ORACLE_DIR, "libsph_oracle.so")
        src = [os.path.join(ORACLE_DIR, f) for f in ("sph_oracle.c", "sph_oracle.h")]
        override = os.environ.get("SPH_ORACLE_LIB")      # bench.py's CPU baseline: the same source built -O3 -march=native on the box it runs on
        if override and os.path.exists(override):
            path = override
        elif not os.path.exists(path) or any(os.path.getmtime(f) > os.path.getmtime(path) for f in src):
            build()      # a library older than its source would silently test yesterday's restatement
        _lib = C.CDLL(path)
        _lib.orc_W.restype = C.c_float; _lib.orc_W.argtypes = [C.c_int, C.c_float, C.c_float]
        _lib.orc_F.restype = C.c_float; _lib.orc_F.argtypes = [C.c_int, C.c_float, C.c_float]
        _lib.orc_wcoeff.restype = C.c_float; _lib.orc_wcoeff.argtypes = [C.c_int, C.c_float, C.c_float]
        _lib.orc_fcoeff.restype = C.c_float; _lib.orc_fcoeff.argtypes = [C.c_int, C.c_float, C.c_float]
        _lib.orc_P.restype = C.c_float; _lib.orc_P.argtypes = [C.c_void_p, C.c_float, C.c_int]
        _lib.orc_soundSpeed.restype = C.c_float; _lib.orc_soundSpeed.argtypes = [C.c_void_p, C.c_float, C.c_int]
        _lib.orc_R.restype = C.c_float; _lib.orc_R.argtypes = [C.c_void_p, C.c_float, C.c_int]
        _lib.orc_RHOR.restype = C.c_float; _lib.orc_RHOR.argtypes = [C.c_void_p, C.c_float, C.c_int]
        _lib.orc_RHO.restype = C.c_float; _lib.orc_RHO.argtypes = [C.c_void_p, C.c_float, C.c_int]
        _lib.orc_forces.restype = C.c_uint32
        _lib.orc_forces_sa.restype = C.c_uint32
        _lib.orc_forces_grenier.restype = C.c_uint32
        _lib.orc_forces_effvisc.restype = C.c_uint32
        _lib.orc_effective_visc.restype = C.c_float
        _lib.orc_dem_interpol.restype = C.c_float; _lib.orc_dem_interpol.argtypes = [C.c_float, C.c_float]
        _lib.orc_effective_visc_value.restype = C.c_float; _lib.orc_effective_visc_value.argtypes = [C.c_void_p, C.c_float, C.c_int]
        _lib.orc_sa_gamma_dt.restype = C.c_float; _lib.orc_sa_gamma_dt.argtypes = [C.c_float, C.c_float]
        _lib.orc_dtreduce.restype = C.c_float
        _lib.orc_dtreduce.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_float, C.c_float]
        _lib.orc_fmax_elements.restype = C.c_uint32; _lib.orc_fmax_elements.argtypes = [C.c_uint32]
        _lib.orc_f4_div.restype = None; _lib.orc_f4_div.argtypes = [C.c_void_p, C.c_float, C.c_void_p]
        _lib.orc_fmax_temp_elements.restype = C.c_uint32; _lib.orc_fmax_temp_elements.argtypes = [C.c_uint32]
        _lib.orc_round_particles.restype = C.c_uint32; _lib.orc_round_particles.argtypes = [C.c_uint32]
        _lib.orc_calc_grid_hash.restype = C.c_uint32
        _lib.orc_num_threads.restype = C.c_int
        _lib.orc_visc_avg.restype = C.c_float; _lib.orc_visc_avg.argtypes = [C.c_void_p] + [C.c_float] * 5
        _lib.orc_wendland_on_segment.restype = C.c_float; _lib.orc_wendland_on_segment.argtypes = [C.c_float]
        _lib.orc_gauss_quadrature_O5.restype = C.c_float; _lib.orc_gauss_quadrature_O5.argtypes = [C.c_void_p] * 4
        _lib.orc_calc_vertex_rel_pos.restype = None; _lib.orc_calc_vertex_rel_pos.argtypes = [C.c_void_p] * 4 + [C.c_float, C.c_void_p]
        _lib.orc_grad_gamma.restype = C.c_float; _lib.orc_grad_gamma.argtypes = [C.c_float] + [C.c_void_p] * 3
        _lib.orc_gamma.restype = C.c_float; _lib.orc_gamma.argtypes = [C.c_int, C.c_float] + [C.c_void_p] * 4 + [C.c_float]
        _lib.orc_RHO.restype = C.c_float; _lib.orc_RHO.argtypes = [C.c_void_p, C.c_float, C.c_int]
    return _lib


def ref():
    """oracle/_ref: the reference's own sources compiled here; None when not built (GPU box without it)."""
    global _ref
    if _ref is None:
        path = os.path.join(ORACLE_DIR, "_ref", "libgpusph_ref.so")
        if not os.path.exists(path):
            return None
        _ref = C.CDLL(path)
        _ref.ref_W.restype = C.c_float; _ref.ref_W.argtypes = [C.c_int, C.c_float, C.c_float, C.c_float, C.c_float]
        _ref.ref_F.restype = C.c_float; _ref.ref_F.argtypes = [C.c_int, C.c_float, C.c_float, C.c_float]
        for n in ("ref_info_id", "ref_info_predicates"):
            getattr(_ref, n).restype = C.c_uint32
            getattr(_ref, n).argtypes = [C.c_uint16] * 4
        for n in ("ref_info_part_type", "ref_info_object", "ref_info_fluid_num"):
            getattr(_ref, n).restype = C.c_int
            getattr(_ref, n).argtypes = [C.c_uint16] * 4
        _ref.ref_active.restype = C.c_int; _ref.ref_active.argtypes = [C.c_float]
        _ref.ref_cell_hash_from_particle_hash.restype = C.c_uint32
        _ref.ref_cell_hash_from_particle_hash.argtypes = [C.c_uint32, C.c_int]
        _ref.ref_encode_cell.restype = C.c_uint32; _ref.ref_encode_cell.argtypes = [C.c_uint32]
        _ref.ref_decode_cell.restype = C.c_int; _ref.ref_decode_cell.argtypes = [C.c_uint32]
        _ref.ref_constant.restype = C.c_uint32; _ref.ref_constant.argtypes = [C.c_int]
        _ref.ref_enum.restype = C.c_int; _ref.ref_enum.argtypes = [C.c_int]
        _ref.ref_visc_avg.restype = C.c_float; _ref.ref_visc_avg.argtypes = [C.c_int] * 3 + [C.c_float] * 5
        _ref.ref_physparams.restype = None; _ref.ref_physparams.argtypes = [C.c_float] * 5 + [C.c_void_p]
        _ref.ref_simparams.restype = None; _ref.ref_simparams.argtypes = [C.c_int, C.c_double, C.c_double, C.c_void_p]
        _ref.ref_float4_div.restype = C.c_float; _ref.ref_float4_div.argtypes = [C.c_float] * 5 + [C.c_void_p]
        _ref.ref_div_up.restype = C.c_uint32; _ref.ref_div_up.argtypes = [C.c_uint32, C.c_uint32]
        _ref.ref_round_up.restype = C.c_uint32; _ref.ref_round_up.argtypes = [C.c_uint32, C.c_uint32]
        _ref.ref_predcorr_buffer_count.restype = C.c_uint32; _ref.ref_predcorr_buffer_count.argtypes = [C.c_uint64]
        _ref.ref_predcorr_multi_buffered.restype = C.c_uint64; _ref.ref_predcorr_multi_buffered.argtypes = [C.c_uint64]
        _ref.ref_buffer_key.restype = C.c_uint64; _ref.ref_buffer_key.argtypes = [C.c_int]
        _ref.ref_ipps.restype = None; _ref.ref_ipps.argtypes = [C.c_ulong, C.c_int, C.c_int, C.c_void_p]
        _ref.ref_wendlandOnSegment.restype = C.c_float; _ref.ref_wendlandOnSegment.argtypes = [C.c_float]
        _ref.ref_gaussQuadratureO5.restype = C.c_float; _ref.ref_gaussQuadratureO5.argtypes = [C.c_void_p] * 4
        _ref.ref_calcVertexRelPos.restype = None; _ref.ref_calcVertexRelPos.argtypes = [C.c_void_p] * 4 + [C.c_float, C.c_void_p]
        _ref.ref_gradGamma.restype = C.c_float; _ref.ref_gradGamma.argtypes = [C.c_float] + [C.c_void_p] * 3
        _ref.ref_Gamma.restype = C.c_float; _ref.ref_Gamma.argtypes = [C.c_int, C.c_float] + [C.c_void_p] * 4 + [C.c_float]
        _ref.ref_visc_avg_singlefluid_nonconst_kinematic.restype = C.c_float
        _ref.ref_visc_avg_singlefluid_nonconst_kinematic.argtypes = [C.c_int] + [C.c_float] * 5
    return _ref


def orc_params_from(sphx_params, problem=None):
    """OrcParams from the product's SphxParams (same numbers, independent struct)."""
    o = OrcParams()
    names = {f[0] for f in OrcParams._fields_}
    for fname, _ in sphx_params._fields_:
        if fname in names:
            v = getattr(sphx_params, fname)
            if hasattr(v, "__len__"):
                dst = getattr(o, fname)
                for i in range(len(v)):
                    dst[i] = v[i]
            else:
                setattr(o, fname, v)
    for b in range(16):
        for a in (0, 4, 8):
            o.rbsteprot[b][a] = 1.0
    if problem is not None and getattr(problem, "planes", None):
        nrm, gpos, lpos = problem.plane_tables()
        o.numplanes = len(nrm)
        for k in range(len(nrm)):
            for a in range(3):
                o.plane_normal[k][a] = float(nrm[k][a]); o.plane_gridpos[k][a] = int(gpos[k][a]); o.plane_pos[k][a] = float(lpos[k][a])
    if problem is not None and getattr(problem, "dem", None) is not None:
        problem._orc_dem = np.ascontiguousarray(problem.dem, dtype=np.float32)      # kept alive for the oracle's pointer
        lib().orc_set_dem(P(problem._orc_dem), C.c_int(problem._orc_dem.shape[1]), C.c_int(problem._orc_dem.shape[0]))
    if problem is not None and getattr(problem, "num_obstacle", 0):
        for b in range(len(problem.rb_firstindex)):      # every body (SAChannelIOFlap: objects 0 and 1 are the open boundaries, 2 the flap)
            for a in range(3):
                o.rbcgGridPos[b][a] = o.rbcgGridPosE[b][a] = int(problem.rb_cg_gridpos[b][a])
                o.rbcgPos[b][a] = o.rbcgPosE[b][a] = float(problem.rb_cg_pos[b][a])
            o.rbstartindex[b] = int(problem.rb_firstindex[b])
    return o


def P(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Oracle:
    """numpy-level wrapper: every method takes/returns host arrays."""

    def __init__(self, params: OrcParams):
        self.p = params
        self.L = lib()

    def calc_hash(self, pos, hash_, info, devmap=None):
        n = len(hash_)
        pidx = np.empty(n, dtype=np.uint32)
        self.L.orc_calc_hash(C.byref(self.p), P(pos), P(hash_), P(pidx), P(info), P(devmap), C.c_uint32(n))
        return pidx

    def fix_hash(self, hash_, info, devmap=None):
        n = len(hash_)
        pidx = np.empty(n, dtype=np.uint32)
        self.L.orc_fix_hash(C.byref(self.p), P(hash_), P(pidx), P(info), P(devmap), C.c_uint32(n))
        return pidx

    def sort(self, hash_, info, pidx):
        self.L.orc_sort(P(hash_), P(info), P(pidx), C.c_uint32(len(hash_)))

    def reorder(self, upos, uvel, sinfo, shash, pidx, ncells, segments=True):
        n = len(shash)
        cs = np.full(ncells, 0xFFFFFFFF, dtype=np.uint32)
        ce = np.full(ncells, 0xFFFFFFFF, dtype=np.uint32)
        seg = np.zeros(4, dtype=np.uint32)
        spos = np.zeros_like(upos); svel = np.zeros_like(uvel)
        newn = np.zeros(1, dtype=np.uint32)
        self.L.orc_reorder(C.byref(self.p), P(cs), P(ce), P(seg) if segments else None, P(spos), P(svel),
                           P(upos), P(uvel), P(sinfo), P(shash), P(pidx), C.c_uint32(n), P(newn))
        return cs, ce, seg, spos, svel, int(newn[0])

    def build_neibs(self, pos, info, hash_, cs, ce, n, range_end, sqinfl):
        stride = int(self.p.neiblist_stride)
        size = int(self.p.neiblistsize) * stride
        if getattr(self, "reuse_nl", False) and getattr(self, "_nl", None) is not None and self._nl.size == size:
            nl = self._nl          # timing runs: clobber the previous list in place instead of faulting in 100 MB again
            nl.fill(0xFFFF)
        else:
            nl = np.full(size, 0xFFFF, dtype=np.uint16)
            self._nl = nl if getattr(self, "reuse_nl", False) else None
        out = OrcNeibsInfo()
        self.L.orc_build_neibs(C.byref(self.p), P(nl), P(pos), P(info), P(hash_), P(cs), P(ce),
                               C.c_uint32(n), C.c_uint32(range_end), C.c_float(sqinfl), C.byref(out))
        return nl, out

    # ---- SA_BOUNDARY (oracle/sph_oracle.c "Semi-analytical boundaries")
    def build_neibs_sa(self, pos, info, vertices, boundelements, hash_, cs, ce, n, range_end, sqinfl, bound_sqinfl):
        stride = int(self.p.neiblist_stride)
        nl = np.full(int(self.p.neiblistsize) * stride, 0xFFFF, dtype=np.uint16)
        vp = [np.zeros((len(pos), 2), dtype=np.float32) for _ in range(3)]
        out = OrcNeibsInfo()
        self.L.orc_build_neibs_sa(C.byref(self.p), P(nl), P(vp[0]), P(vp[1]), P(vp[2]), P(pos), P(info), P(vertices),
                                  P(boundelements), P(hash_), P(cs), P(ce), C.c_uint32(n), C.c_uint32(range_end),
                                  C.c_float(sqinfl), C.c_float(bound_sqinfl), C.byref(out))
        return nl, vp, out

    def sa_compute_vertex_normal(self, boundelements, vertices, info, hash_, cs, nl, n):
        be = boundelements.copy()
        self.L.orc_sa_compute_vertex_normal(C.byref(self.p), P(be), P(vertices), P(info), P(hash_), P(cs), P(nl), C.c_uint32(n))
        return be

    def sa_segment_bc(self, pos, vel, ggam, vertices, boundelements, info, hash_, cs, nl, n, step, repack=False):
        v, g = vel.copy(), ggam.copy()
        self.L.orc_sa_segment_bc(C.byref(self.p), P(v), P(g), P(pos), P(vertices), P(boundelements), P(info), P(hash_), P(cs),
                                 P(nl), C.c_uint32(n), C.c_int(step), C.c_int(1 if repack else 0))
        return v, g

    # ---- turbulence<KEPSILON>: ke = dict(tke, eps, turbvisc, eulervel) numpy arrays
    def sa_bc_keps(self, pos, vel, ggam, ke, vertices, boundelements, info, hash_, cs, nl, n, step, deltap, vertex=True):
        """segment then (vertex=True) vertex boundary conditions with the k-epsilon members; -> vel, ggam, ke (copies)"""
        v, g = vel.copy(), ggam.copy()
        k = {name: a.copy() for name, a in ke.items()}
        self.L.orc_sa_segment_bc_keps(C.byref(self.p), P(v), P(g), P(k["tke"]), P(k["eps"]), P(k["eulervel"]), P(pos), P(vertices),
                                      P(boundelements), P(info), P(hash_), P(cs), P(nl), C.c_uint32(n), C.c_int(step), C.c_float(deltap))
        if vertex:
            self.L.orc_sa_vertex_bc_keps(C.byref(self.p), P(v), P(g), P(k["tke"]), P(k["eps"]), P(k["eulervel"]), P(pos), P(vertices),
                                         P(boundelements), P(info), P(hash_), P(cs), P(nl), C.c_uint32(n))
        return v, g, k

    def forces_sa_keps(self, pos, vel, info, hash_, cs, nl, ggam, boundelements, vertpos, ke, n, deltap, epsilon=5e-5):
        """-> forces, cfl, numBlocks, dkde (n, 3), strain (n, 6); self.cfl_keps, self.max_gamma_cfl"""
        forces = np.zeros((len(pos), 4), dtype=np.float32)
        nblk = int(self.L.orc_fmax_elements(C.c_uint32(len(pos))))
        cfl = np.zeros(nblk, dtype=np.float32)
        self.cfl_keps = np.zeros(nblk, dtype=np.float32)
        self.cfl_gamma = np.zeros(((n + 3) // 4) * 4 + nblk, dtype=np.float32)
        dkde = np.zeros((len(pos), 3), dtype=np.float32); strain = np.zeros((len(pos), 6), dtype=np.float32)
        self.L.orc_forces_sa_keps.restype = C.c_uint32
        nb = self.L.orc_forces_sa_keps(C.byref(self.p), P(forces), P(cfl), P(self.cfl_gamma), P(self.cfl_keps), P(dkde), P(strain),
                                       P(pos), P(vel), P(info), P(hash_), P(cs), P(nl), P(ggam), P(boundelements),
                                       P(vertpos[0]), P(vertpos[1]), P(vertpos[2]), P(ke["tke"]), P(ke["eps"]), P(ke["turbvisc"]),
                                       P(ke["eulervel"]), C.c_uint32(n), C.c_uint32(0), C.c_uint32(n), C.c_uint32(0),
                                       C.c_float(deltap), C.c_float(epsilon))
        self.max_gamma_cfl = float(self.cfl_gamma[((n + 3) // 4) * 4:((n + 3) // 4) * 4 + int(nb)].max()) if nb else 0.0
        return forces, cfl, int(nb), dkde, strain

    def euler_keps(self, ke, dkde, forces, old_pos, info, n, dt):
        new = {name: a.copy() for name, a in ke.items()}
        self.L.orc_euler_keps(C.byref(self.p), P(new["tke"]), P(new["eps"]), P(new["turbvisc"]), P(new["eulervel"]), P(ke["tke"]),
                              P(ke["eps"]), P(ke["eulervel"]), P(dkde), P(forces), P(old_pos), P(info), C.c_uint32(n), C.c_float(dt))
        return new

    def sa_init_gamma(self, ggam, pos, boundelements, vertpos, info, hash_, cs, nl, n, deltap, epsilon=5e-5):
        g = ggam.copy()
        self.L.orc_sa_init_gamma(C.byref(self.p), P(g), P(pos), P(boundelements), P(vertpos[0]), P(vertpos[1]), P(vertpos[2]), P(info),
                                 P(hash_), P(cs), P(nl), C.c_uint32(n), C.c_float(deltap), C.c_float(epsilon))
        return g

    def forces_sa(self, pos, vel, info, hash_, cs, nl, ggam, boundelements, vertpos, n, deltap):
        forces = np.zeros((len(pos), 4), dtype=np.float32)
        nblk = int(self.L.orc_fmax_elements(C.c_uint32(len(pos))))
        cfl = np.zeros(nblk, dtype=np.float32)
        # BUFFER_CFL_GAMMA: one value per particle, then one per block from round_up(n, 4) on
        self.cfl_gamma = np.zeros(((n + 3) // 4) * 4 + nblk, dtype=np.float32)
        nb = self.L.orc_forces_sa(C.byref(self.p), P(forces), P(cfl), P(self.cfl_gamma), P(pos), P(vel), P(info), P(hash_), P(cs), P(nl),
                                  P(ggam), P(boundelements), P(vertpos[0]), P(vertpos[1]), P(vertpos[2]), C.c_uint32(n), C.c_uint32(0),
                                  C.c_uint32(n), C.c_uint32(0), C.c_float(deltap))
        self.max_gamma_cfl = float(self.cfl_gamma[((n + 3) // 4) * 4:((n + 3) // 4) * 4 + int(nb)].max()) if nb else 0.0
        return forces, cfl, int(nb)

    def forces_sa_io(self, pos, vel, euler_vel, info, hash_, cs, nl, ggam, boundelements, vertpos, n, deltap):
        """forces_sa with open boundaries enabled (oracle groundwork): the viscous terms and the gamma CFL see BUFFER_EULERVEL"""
        forces = np.zeros((len(pos), 4), dtype=np.float32)
        nblk = int(self.L.orc_fmax_elements(C.c_uint32(len(pos))))
        cfl = np.zeros(nblk, dtype=np.float32)
        self.cfl_gamma = np.zeros(((n + 3) // 4) * 4 + nblk, dtype=np.float32)
        self.L.orc_forces_sa_io.restype = C.c_uint32
        nb = self.L.orc_forces_sa_io(C.byref(self.p), P(forces), P(cfl), P(self.cfl_gamma), P(pos), P(vel), P(euler_vel), P(info), P(hash_),
                                     P(cs), P(nl), P(ggam), P(boundelements), P(vertpos[0]), P(vertpos[1]), P(vertpos[2]), C.c_uint32(n),
                                     C.c_uint32(0), C.c_uint32(n), C.c_uint32(0), C.c_float(deltap))
        self.max_gamma_cfl = float(self.cfl_gamma[((n + 3) // 4) * 4:((n + 3) // 4) * 4 + int(nb)].max()) if nb else 0.0
        return forces, cfl, int(nb)

    def repack_forces_sa(self, pos, vel, info, hash_, cs, nl, ggam, boundelements, vertpos, n, deltap):
        forces = np.zeros((len(pos), 4), dtype=np.float32)
        nblk = int(self.L.orc_fmax_elements(C.c_uint32(len(pos))))
        cfl = np.zeros(nblk, dtype=np.float32)
        self.L.orc_repack_forces_sa.restype = C.c_uint32
        nb = self.L.orc_repack_forces_sa(C.byref(self.p), P(forces), P(cfl), P(pos), P(vel), P(info), P(hash_), P(cs), P(nl), P(ggam),
                                         P(boundelements), P(vertpos[0]), P(vertpos[1]), P(vertpos[2]), C.c_uint32(n), C.c_uint32(0),
                                         C.c_uint32(n), C.c_uint32(0), C.c_float(deltap))
        return forces, cfl, int(nb)

    def sa_density_sum(self, new_vel, old_pos, new_pos, old_vel, old_ggam, boundelements, vertpos, info, hash_, cs, nl, n):
        """density_sum of the integration engine: new_vel.w and gamma of the fluid from the old and new positions"""
        v = new_vel.copy(); g = old_ggam.copy()
        scratch = np.zeros((len(old_pos), 4), dtype=np.float32)
        self.L.orc_sa_density_sum(C.byref(self.p), P(v), P(g), P(scratch), P(old_pos), P(new_pos), P(old_vel), P(old_ggam), P(boundelements),
                                  P(vertpos[0]), P(vertpos[1]), P(vertpos[2]), P(info), P(hash_), P(cs), P(nl), C.c_uint32(n))
        return v, g

    # ---- SA_BOUNDARY with moving bodies (oracle/sph_oracle.c "SA_BOUNDARY with MOVING bodies")
    def sa_update_normals(self, boundelements, info, n):
        """update_normals of the Euler step: the normals of moving segments and vertices turned by the body's step rotation"""
        out = boundelements.copy()
        self.L.orc_sa_update_normals(C.byref(self.p), P(out), P(boundelements), P(info), C.c_uint32(n))
        return out

    def sa_density_sum_moving(self, new_vel, old_pos, new_pos, old_vel, old_ggam, new_ggam_in, be_old, be_new, vertpos, info, hash_, cs, nl, n):
        """density_sum with ENABLE_MOVING_BODIES: fluid rows (density, gamma) and vertex rows (gamma); the boundary rows keep
        new_ggam_in's"""
        v = new_vel.copy(); g = new_ggam_in.copy()
        scratch = np.zeros((len(old_pos), 4), dtype=np.float32)
        self.L.orc_sa_density_sum_moving(C.byref(self.p), P(v), P(g), P(scratch), P(old_pos), P(new_pos), P(old_vel), P(old_ggam),
                                         P(be_old), P(be_new), P(vertpos[0]), P(vertpos[1]), P(vertpos[2]), P(info), P(hash_), P(cs), P(nl),
                                         C.c_uint32(n))
        return v, g

    def sa_integrate_gamma_moving(self, old_ggam, new_pos, be_new, vertpos, info, hash_, cs, nl, n, epsilon=5e-5):
        """integrate_gamma with ENABLE_GAMMA_QUADRATURE and ENABLE_MOVING_BODIES: fluid AND vertex rows by quadrature against the
        new elements, boundary rows copied"""
        g = old_ggam.copy()
        for cptype in (0, 2):
            self.L.orc_sa_integrate_gamma_quadrature(C.byref(self.p), P(g), P(old_ggam), P(new_pos), P(be_new), P(vertpos[0]),
                                                     P(vertpos[1]), P(vertpos[2]), P(info), P(hash_), P(cs), P(nl), C.c_uint32(n),
                                                     C.c_int(cptype), C.c_float(epsilon))
        return g

    def sa_body_pressure_forces(self, pos, vel, info, hash_, boundelements, n, rows):
        """-P A n of the FG_COMPUTE_FORCE boundary elements: (forces rows, BUFFER_RB_FORCES, BUFFER_RB_TORQUES)"""
        f = np.zeros((len(pos), 4), dtype=np.float32)
        rbf, rbt = np.zeros((rows, 4), dtype=np.float32), np.zeros((rows, 4), dtype=np.float32)
        self.L.orc_sa_body_pressure_forces(C.byref(self.p), P(f), P(rbf), P(rbt), P(pos), P(vel), P(info), P(hash_), P(boundelements),
                                           C.c_uint32(0), C.c_uint32(n))
        return f, rbf, rbt

    def sa_density_diffusion(self, pos, vel, ggam, info, hash_, cs, nl, n, dt):
        """compute_density_diffusion (Brezzi) + apply_density_diffusion: returns the updated velocity array"""
        f = np.zeros((len(pos), 4), dtype=np.float32)
        self.L.orc_sa_density_diffusion(C.byref(self.p), P(f), P(pos), P(vel), P(ggam), P(info), P(hash_), P(cs), P(nl), C.c_uint32(n),
                                        C.c_float(dt))
        v = vel.copy()
        fluid = (info[:n, 0] & 7) == 0
        v[:n][fluid, 3] = v[:n][fluid, 3] + f[:n][fluid, 3] * np.float32(dt)
        return v, f

    def sa_integrate_gamma(self, old_ggam, new_pos, boundelements, vertpos, info, hash_, cs, nl, n, epsilon=5e-5):
        """integrate_gamma with ENABLE_GAMMA_QUADRATURE: fluid rows by quadrature, the rest copied"""
        g = old_ggam.copy()
        self.L.orc_sa_integrate_gamma_quadrature(C.byref(self.p), P(g), P(old_ggam), P(new_pos), P(boundelements), P(vertpos[0]),
                                                 P(vertpos[1]), P(vertpos[2]), P(info), P(hash_), P(cs), P(nl), C.c_uint32(n),
                                                 C.c_int(0), C.c_float(epsilon))
        return g

    def sa_vertex_bc(self, pos, vel, ggam, info, hash_, cs, nl, n):
        v = vel.copy()
        self.L.orc_sa_vertex_bc(C.byref(self.p), P(v), P(ggam), P(pos), P(info), P(hash_), P(cs), P(nl), C.c_uint32(n))
        return v

    # ---- open boundaries of SA_BOUNDARY: oracle GROUNDWORK only (oracle/sph_oracle.c "Open boundaries"), no product counterpart
    def riemann_R(self, rho_tilde, fluid=0):
        return float(self.L.orc_R(C.byref(self.p), C.c_float(rho_tilde), C.c_int(fluid)))

    def riemann_RHOR(self, r, fluid=0):
        return float(self.L.orc_RHOR(C.byref(self.p), C.c_float(r), C.c_int(fluid)))

    def eos_P(self, rho_tilde, fluid=0):
        return float(self.L.orc_P(C.byref(self.p), C.c_float(rho_tilde), C.c_int(fluid)))

    def eos_RHO(self, pressure, fluid=0):
        return float(self.L.orc_RHO(C.byref(self.p), C.c_float(pressure), C.c_int(fluid)))

    def sound_speed(self, rho_tilde, fluid=0):
        return float(self.L.orc_soundSpeed(C.byref(self.p), C.c_float(rho_tilde), C.c_int(fluid)))

    def io_boundary_condition(self, euler_vel, velocity_driven, rho_int, rho_ext, u_int, un_int, un_ext, normal, fluid=0):
        """calculateIOboundaryCondition: the (u, rho~) imposed on an open-boundary element -> the completed condition"""
        ev = np.asarray(euler_vel, dtype=np.float32).copy()
        u = np.asarray(u_int, dtype=np.float32); nrm = np.asarray(normal, dtype=np.float32)
        self.L.orc_io_boundary_condition(C.byref(self.p), P(ev), C.c_int(1 if velocity_driven else 0), C.c_int(fluid),
                                         C.c_float(rho_int), C.c_float(rho_ext), P(u), C.c_float(un_int), C.c_float(un_ext), P(nrm))
        return ev

    def mass_repartition(self, vertex_rel_pos, normal):
        v = np.ascontiguousarray(np.asarray(vertex_rel_pos, dtype=np.float32).reshape(9))
        nrm = np.asarray(normal, dtype=np.float32)
        beta = np.zeros(3, dtype=np.float32)
        self.L.orc_mass_repartition(P(v), P(nrm), P(beta))
        return beta

    def sa_identify_corner_vertices(self, pos, info, hash_, vertices, cs, nl, n):
        out = info.copy()
        self.L.orc_sa_identify_corner_vertices(C.byref(self.p), P(pos), P(out), P(hash_), P(vertices), P(cs), P(nl), C.c_uint32(n))
        return out

    def sa_init_io_mass(self, pos, info, hash_, vertices, cs, nl, n, deltap):
        """INIT_IO_MASS_VERTEX_COUNT + INIT_IO_MASS: (vertex counts, positions with the new vertex masses)"""
        forces = np.zeros((len(pos), 4), dtype=np.float32)
        self.L.orc_sa_init_io_mass_vertex_count(C.byref(self.p), P(vertices), P(hash_), P(info), P(cs), P(nl), P(forces), C.c_uint32(n))
        new_pos = np.zeros_like(pos)
        self.L.orc_sa_init_io_mass(C.byref(self.p), P(pos), P(forces), P(vertices), P(hash_), P(info), P(cs), P(nl), P(new_pos),
                                   C.c_uint32(n), C.c_float(deltap))
        return forces[:, 3].copy(), new_pos

    def sa_segment_bc_io(self, pos, vel, ggam, euler_vel, vertices, boundelements, info, hash_, cs, nl, n, step):
        """the segment conditions with open boundaries enabled: (vel, gradgamma, eulerVel) after the pass"""
        v, g, e = vel.copy(), ggam.copy(), euler_vel.copy()
        self.L.orc_sa_segment_bc_io(C.byref(self.p), P(v), P(g), P(e), P(pos), P(vertices), P(boundelements), P(info), P(hash_), P(cs),
                                    P(nl), C.c_uint32(n), C.c_int(step))
        return v, g, e

    def find_outgoing_segment(self, pos, vel, vertices, ggam, vertpos, boundelements, info, hash_, cs, nl, n, influenceradius):
        """FIND_OUTGOING_SEGMENT: (vertices, gradgamma) with the marks of the fluid particles that left through an open boundary"""
        v, g = vertices.copy(), ggam.copy()
        self.L.orc_find_outgoing_segment(C.byref(self.p), P(pos), P(vel), P(v), P(g), P(vertpos[0]), P(vertpos[1]), P(vertpos[2]),
                                         P(boundelements), P(info), P(hash_), P(cs), P(nl), C.c_uint32(n), C.c_float(influenceradius))
        return v, g

    def sa_vertex_bc_io(self, pos, vel, ggam, euler_vel, vertices, boundelements, vertpos, info, hash_, next_ids, cs, nl, n,
                        deltap, dt, step, num_open_vertices, room=None):
        """the vertex conditions with open boundaries enabled.  Every per-particle array is grown by `room` rows for the particles
        the pass creates; returns a dict of the arrays after the pass and the new particle count"""
        room = int(num_open_vertices) if room is None else int(room)
        tot = len(pos) + room
        def grow(a, fill=0):
            out = np.full((tot,) + a.shape[1:], fill, dtype=a.dtype)
            out[:len(a)] = a
            return out
        a = dict(vel=grow(vel), new_pos=grow(pos), ggam=grow(ggam), euler_vel=grow(euler_vel),
                 forces=grow(np.zeros((len(pos), 4), dtype=np.float32)), vertices=grow(vertices), boundelements=grow(boundelements),
                 info=grow(info), hash=grow(hash_), next_ids=grow(next_ids.astype(np.uint32), 0xFFFFFFFF))
        old_pos = grow(pos)
        newn = C.c_uint32(n)
        self.L.orc_sa_vertex_bc_io(C.byref(self.p), P(a["vel"]), P(old_pos), P(a["new_pos"]), P(a["ggam"]), P(a["euler_vel"]),
                                   P(a["forces"]), P(a["vertices"]), P(a["boundelements"]), P(vertpos[0]), P(vertpos[1]), P(vertpos[2]),
                                   P(a["info"]), P(a["hash"]), P(a["next_ids"]), C.byref(newn), P(cs), P(nl), C.c_uint32(n),
                                   C.c_uint32(tot), C.c_float(deltap), C.c_float(dt), C.c_int(step), C.c_uint32(num_open_vertices))
        a["n"] = int(newn.value)
        return a

    def sa_density_sum_io(self, new_vel, old_pos, new_pos, old_vel, old_euler_vel, old_ggam, boundelements, vertpos, info, hash_,
                          cs, nl, n, dt):
        """density_sum with open boundaries enabled: (new_vel, gamma, the volumic sums in forces.w)"""
        v = new_vel.copy(); g = old_ggam.copy()
        scratch = np.zeros((len(old_pos), 4), dtype=np.float32)
        self.L.orc_sa_density_sum_io(C.byref(self.p), P(v), P(g), P(scratch), P(old_pos), P(new_pos), P(old_vel), P(old_euler_vel),
                                     P(old_ggam), P(boundelements), P(vertpos[0]), P(vertpos[1]), P(vertpos[2]), P(info), P(hash_),
                                     P(cs), P(nl), C.c_uint32(n), C.c_float(dt))
        return v, g, scratch[:, 3].copy()

    def sa_density_sum_io_moving(self, new_vel, old_pos, new_pos, old_vel, old_euler_vel, old_ggam, be_old, be_new, vertpos, info, hash_,
                                 cs, nl, n, dt):
        """density_sum with open boundaries AND moving bodies (CompleteSaExample's option set): (new_vel, gamma, the volumic sums);
        the BOUNDARY rows of gamma keep the old values here (the kernels leave them to the segment condition)"""
        v = new_vel.copy(); g = old_ggam.copy()
        scratch = np.zeros((len(old_pos), 4), dtype=np.float32)
        self.L.orc_sa_density_sum_io_moving(C.byref(self.p), P(v), P(g), P(scratch), P(old_pos), P(new_pos), P(old_vel), P(old_euler_vel),
                                            P(old_ggam), P(be_old), P(be_new), P(vertpos[0]), P(vertpos[1]), P(vertpos[2]), P(info),
                                            P(hash_), P(cs), P(nl), C.c_uint32(n), C.c_float(dt))
        return v, g, scratch[:, 3].copy()

    def sa_density_diffusion_io(self, pos, vel, ggam, info, hash_, cs, nl, boundelements, vertpos, n, dt, deltap):
        """the Brezzi diffusion with open boundaries enabled: (updated velocity, forces with the diffusion in .w)"""
        f = np.zeros((len(pos), 4), dtype=np.float32)
        self.L.orc_sa_density_diffusion_io(C.byref(self.p), P(f), P(pos), P(vel), P(ggam), P(info), P(hash_), P(cs), P(nl),
                                           P(boundelements), P(vertpos[0]), P(vertpos[1]), P(vertpos[2]), C.c_uint32(n),
                                           C.c_float(dt), C.c_float(deltap))
        v = vel.copy()
        fluid = (info[:n, 0] & 7) == 0
        v[:n][fluid, 3] = v[:n][fluid, 3] + f[:n][fluid, 3] * np.float32(dt)
        return v, f

    def sa_io_water_depth(self, depth, pos, info, hash_, cs, nl, n, frm=0):
        """the water depth the vertex pass of the forces leaves behind: depth (uint32 per open boundary) is updated in place"""
        self.L.orc_sa_io_water_depth(C.byref(self.p), P(depth), P(pos), P(info), P(hash_), P(cs), P(nl), C.c_uint32(frm),
                                     C.c_uint32(n))
        return depth

    def sa_io_water_depth_z(self, u):
        self.L.orc_sa_io_water_depth_z.restype = C.c_float
        return float(self.L.orc_sa_io_water_depth_z(C.byref(self.p), C.c_uint32(int(u))))

    def disable_outgoing_parts(self, pos, vertices, info, n):
        p2, v2 = pos.copy(), vertices.copy()
        self.L.orc_disable_outgoing_parts(P(p2), P(v2), P(info), C.c_uint32(n))
        return p2, v2

    def effective_visc(self, pos, vel, info, hash_, cs, nl, n, range_end=None):
        """CALC_VISC of the generalized Newtonian rheologies: (effvisc per particle, largest kinematic viscosity)"""
        range_end = n if range_end is None else range_end
        eff = np.zeros(len(pos), dtype=np.float32)
        nblk = int(self.L.orc_fmax_elements(C.c_uint32(len(pos))))
        cfl = np.zeros(nblk, dtype=np.float32)
        mx = self.L.orc_effective_visc(C.byref(self.p), P(eff), P(cfl), P(pos), P(vel), P(info), P(hash_), P(cs), P(nl),
                                       C.c_uint32(n), C.c_uint32(range_end))
        return eff, float(mx)

    def forces(self, pos, vel, info, hash_, cs, nl, n, frm=0, to=None, cfl_offset=0, compute_object_forces=0,
               rb_count=0, tau=None, effvisc=None, dedt=None):
        """dedt: BUFFER_INTERNAL_ENERGY_UPD (ENABLE_INTERNAL_ENERGY), cleared and accumulated by the three passes"""
        to = n if to is None else to
        forces = np.zeros((len(pos), 4), dtype=np.float32)
        if dedt is not None:
            dedt[:] = 0          # pre_forces clobbers it (GPUWorker.cc:1961-1963)
        self.L.orc_set_dedt(P(dedt))
        nblk = int(self.L.orc_fmax_elements(C.c_uint32(len(pos))))
        cfl = np.zeros(nblk + cfl_offset, dtype=np.float32)
        rbf = np.zeros((max(rb_count, 1), 4), dtype=np.float32)
        rbt = np.zeros((max(rb_count, 1), 4), dtype=np.float32)
        nb = self.L.orc_forces_effvisc(C.byref(self.p), P(forces), P(cfl), P(rbf) if rb_count else None,
                                       P(rbt) if rb_count else None, P(pos), P(vel), P(info), P(hash_), P(cs), P(nl), P(tau),
                                       C.c_uint32(n), C.c_uint32(frm), C.c_uint32(to), C.c_uint32(cfl_offset),
                                       C.c_int(compute_object_forces), P(effvisc))
        self.L.orc_set_dedt(None)
        return forces, cfl, int(nb), rbf, rbt

    # ---- SPH_GRENIER (oracle/sph_oracle.c "SPH_GRENIER")
    def init_volume(self, pos, vel, info, n):
        vol = np.zeros((len(pos), 4), dtype=np.float32)
        self.L.orc_init_volume(C.byref(self.p), P(vol), P(pos), P(vel), P(info), C.c_uint32(n))
        return vol

    def density_grenier(self, pos, vel, info, hash_, vol, cs, nl, n, max_fb_neibs):
        """COMPUTE_DENSITY: rewrites vel[:, 3] in place, returns sigma"""
        sigma = np.zeros(len(pos), dtype=np.float32)
        self.L.orc_density_grenier(C.byref(self.p), P(sigma), P(vel), P(pos), P(info), P(hash_), P(vol), P(cs), P(nl),
                                   C.c_uint32(n), C.c_int(int(max_fb_neibs)))
        return sigma

    def forces_grenier(self, pos, vel, info, hash_, cs, nl, sigma, n, frm=0, to=None, cfl_offset=0):
        to = n if to is None else to
        forces = np.zeros((len(pos), 4), dtype=np.float32)
        nblk = int(self.L.orc_fmax_elements(C.c_uint32(len(pos))))
        cfl = np.zeros(nblk + cfl_offset, dtype=np.float32)
        nb = self.L.orc_forces_grenier(C.byref(self.p), P(forces), P(cfl), P(pos), P(vel), P(info), P(hash_), P(cs), P(nl),
                                       P(sigma), C.c_uint32(n), C.c_uint32(frm), C.c_uint32(to), C.c_uint32(cfl_offset))
        return forces, cfl, int(nb)

    def euler_grenier(self, old_pos, old_vel, old_vol, info, hash_, forces, n, dt, step, xsph=None):
        npos = np.zeros_like(old_pos); nvel = np.zeros_like(old_vel); nvol = np.zeros_like(old_vol)
        self.L.orc_euler_grenier(C.byref(self.p), P(npos), P(nvel), P(nvol), P(old_pos), P(old_vel), P(old_vol), P(info),
                                 P(hash_), P(forces), P(xsph), C.c_uint32(n), C.c_float(dt), C.c_int(step))
        return npos, nvel, nvol

    def dtreduce(self, cfl, nblocks, sspeed_cfl, max_kinematic=0.0):
        # GPUWorker.cc:2013-2022: the MONAGHAN / ESPANOL_REVENGA viscous models tighten the viscous limit
        max_kinematic = max_kinematic*(float(self.p.monaghan_visc_coeff) if self.p.viscmodel == 1 else 5.0 if self.p.viscmodel == 2 else 1.0)
        return float(self.L.orc_dtreduce(C.byref(self.p), P(cfl), C.c_uint32(nblocks), C.c_float(sspeed_cfl),
                                         C.c_float(max_kinematic)))

    def euler(self, old_pos, old_vel, info, hash_, forces, n, dt, step, xsph=None):
        npos = np.zeros_like(old_pos); nvel = np.zeros_like(old_vel)
        self.L.orc_euler(C.byref(self.p), P(npos), P(nvel), P(old_pos), P(old_vel), P(info), P(hash_), P(forces), P(xsph),
                         C.c_uint32(n), C.c_float(dt), C.c_int(step))
        return npos, nvel

    def euler_energy(self, old_energy, dedt, old_pos, info, n, dt):
        new = np.zeros_like(old_energy)
        self.L.orc_euler_energy(C.byref(self.p), P(new), P(old_energy), P(dedt), P(old_pos), P(info), C.c_uint32(n), C.c_float(dt))
        return new

    def xsph(self, pos, vel, info, hash_, cs, nl, n, out=None):
        """mean neighbourhood velocity of the forces pass (ENABLE_XSPH); rows of non-fluid particles keep their content"""
        out = np.zeros((len(pos), 4), dtype=np.float32) if out is None else out
        self.L.orc_xsph(C.byref(self.p), P(out), P(pos), P(vel), P(info), P(hash_), P(cs), P(nl), C.c_uint32(0), C.c_uint32(n))
        return out

    def repack_forces(self, pos, vel, info, hash_, cs, nl, n, frm=0, to=None, cfl_offset=0, rb_count=0):
        to = n if to is None else to
        forces = np.zeros((len(pos), 4), dtype=np.float32)
        nblk = int(self.L.orc_fmax_elements(C.c_uint32(len(pos))))
        cfl = np.zeros(nblk + cfl_offset, dtype=np.float32)
        rbf = np.ones((max(rb_count, 1), 4), dtype=np.float32)
        rbt = np.ones((max(rb_count, 1), 4), dtype=np.float32)
        self.L.orc_repack_forces.restype = C.c_uint32
        nb = self.L.orc_repack_forces(C.byref(self.p), P(forces), P(cfl), P(rbf) if rb_count else None,
                                      P(rbt) if rb_count else None, P(pos), P(vel), P(info), P(hash_), P(cs), P(nl),
                                      C.c_uint32(n), C.c_uint32(frm), C.c_uint32(to), C.c_uint32(cfl_offset))
        return forces, cfl, int(nb), rbf, rbt

    def euler_repack(self, old_pos, old_vel, info, hash_, forces, n, dt, step=1):
        npos = np.zeros_like(old_pos); nvel = np.zeros_like(old_vel)
        self.L.orc_euler_repack(C.byref(self.p), P(npos), P(nvel), P(old_pos), P(old_vel), P(info), P(hash_), P(forces),
                                C.c_uint32(n), C.c_float(dt), C.c_int(step))
        return npos, nvel

    def disable_free_surf_parts(self, pos, info, n):
        self.L.orc_disable_free_surf_parts(P(pos), P(info), C.c_uint32(n))

    def filter(self, filtertype, pos, vel, info, hash_, cs, nl, range_end):
        """shepard (0) / MLS (1); inactive particles keep their input velocity in the returned copy"""
        new = vel.copy()
        fn = self.L.orc_shepard if filtertype == 0 else self.L.orc_mls
        fn(C.byref(self.p), P(new), P(pos), P(vel), P(info), P(hash_), P(cs), P(nl), C.c_uint32(range_end))
        return new

    def vorticity(self, pos, vel, info, hash_, cs, nl, range_end):
        out = np.zeros((len(pos), 3), dtype=np.float32)
        self.L.orc_vorticity(C.byref(self.p), P(out), P(pos), P(vel), P(info), P(hash_), P(cs), P(nl), C.c_uint32(range_end))
        return out

    def testpoints(self, pos, vel, info, hash_, cs, nl, range_end):
        new = vel.copy()
        self.L.orc_testpoints(C.byref(self.p), P(new), P(pos), P(info), P(hash_), P(cs), P(nl), C.c_uint32(range_end))
        return new

    def surface(self, pos, vel, info, hash_, cs, nl, range_end, normals=False, cosf=0.86, cosn=0.5):
        new = info.copy()
        nrm = np.zeros((len(pos), 4), dtype=np.float32) if normals else None
        self.L.orc_surface(C.byref(self.p), P(new), P(nrm), P(pos), P(vel), P(hash_), P(cs), P(nl), C.c_uint32(range_end),
                           C.c_float(cosf), C.c_float(cosn))
        return new, nrm

    def interface(self, pos, vel, info, hash_, cs, nl, range_end, normals=False, cosf=0.86, cosn=0.5):
        new = info.copy()
        nrm = np.zeros((len(pos), 4), dtype=np.float32) if normals else None
        self.L.orc_interface(C.byref(self.p), P(new), P(nrm), P(pos), P(vel), P(hash_), P(cs), P(nl), C.c_uint32(range_end),
                             C.c_float(cosf), C.c_float(cosn))
        return new, nrm

    def sps(self, pos, vel, info, hash_, cs, nl, n, range_end):
        tau = np.zeros((len(pos), 6), dtype=np.float32)
        tv = np.zeros(len(pos), dtype=np.float32)
        self.L.orc_sps(C.byref(self.p), P(tau), P(tv), P(pos), P(vel), P(info), P(hash_), P(cs), P(nl),
                       C.c_uint32(n), C.c_uint32(range_end))
        return tau, tv


class OracleSim:
    """Whole-step driver on the host following the reference's command sequence
    (Integrator::buildNeibsPhase src/Integrator.cc:94-250, PredictorCorrector
    src/integrators/PredictorCorrectorIntegrator.cc:386-685, dt: GPUWorker.cc:2226-2229, GPUSPH.cc:636-699)."""

    def __init__(self, problem, allocated=None):
        import gpusph_amd.defs as D
        self.D = D
        self.problem = problem
        arrs = problem.copy_to_array()
        self.n = len(arrs["hash"])
        self.alloc = allocated or self.n
        self.sp = problem.sphx_params(self.alloc)
        self.op = orc_params_from(self.sp, problem)
        self.o = Oracle(self.op)
        self.pos, self.vel, self.info, self.hash = arrs["pos"], arrs["vel"], arrs["info"], arrs["hash"]
        self.dt = float(np.float32(problem.simparams.dt))
        self.t = 0.0
        self.iterations = 0
        self.ncells = problem.grid_cells
        sscoeff = max(problem.physparams.sscoeff)
        self.sspeed_cfl = float(np.float32(np.float64(np.float32(sscoeff)) * 1.1))  # GPUWorker.cc:3010-3011
        pp, sp = problem.physparams, problem.simparams
        self.max_kinvisc = float(np.float32(max(pp.kinematicvisc))) if sp.rheologytype == D.NEWTONIAN else 0.0   # :3003-3006
        self.neibs_info = None
        self.bodies = None
        self.grenier = sp.sph_formulation == D.SPH_GRENIER
        self.effvisc_on = sp.rheologytype > D.NEWTONIAN         # NEEDS_EFFECTIVE_VISC
        self.energy_on = bool(sp.simflags & D.ENABLE_INTERNAL_ENERGY)
        if self.energy_on:      # init_internal_energy (src/ProblemCore.cc:1609-1618): zero
            self.energy = np.zeros(len(self.pos), dtype=np.float32)
            self.dedt = np.zeros(len(self.pos), dtype=np.float32)
        if self.grenier:    # GPUSPH.cc:495-496
            self.vol = self.o.init_volume(self.pos, self.vel, self.info, self.n)
        if getattr(problem, "moving_bodies_callback", None) is not None and getattr(problem, "num_obstacle", 0):
            from gpusph_amd.bodies import MovingBodies
            self.bodies = MovingBodies(problem, problem.rb_cg_global)

    def _move_bodies(self, step, dt, t):
        m = self.bodies.timestep(step, dt, t)
        p = self.o.p
        for b in range(len(self.bodies)):
            for a in range(3):
                p.rbtrans[b][a] = float(m["trans"][b][a]); p.rblinearvel[b][a] = float(m["lvel"][b][a])
                p.rbangularvel[b][a] = float(m["avel"][b][a])
                if self.problem.simparams.numforcesbodies > 0:
                    p.rbcgGridPos[b][a] = int(m["cg_grid"][b][a]); p.rbcgPos[b][a] = float(m["cg_pos"][b][a])
            for a in range(9):
                p.rbsteprot[b][a] = float(m["rot"][b][a])
        self._last_motion = m

    def build_neibs(self):
        o = self.o
        if self.iterations == 0:
            pidx = o.fix_hash(self.hash, self.info)
        else:
            pidx = o.calc_hash(self.pos, self.hash, self.info)
        o.sort(self.hash, self.info, pidx)
        self.cs, self.ce, self.seg, spos, svel, newn = o.reorder(self.pos, self.vel, self.info, self.hash, pidx, self.ncells,
                                                                 segments=False)
        self.pos, self.vel = spos, svel
        if getattr(self, "grenier", False):
            self.vol = np.ascontiguousarray(self.vol[pidx])
        if getattr(self, "energy_on", False):
            self.energy = np.ascontiguousarray(self.energy[pidx])
        self.partindex = pidx
        self.n = newn
        sq = float(np.float32(self.problem.simparams.nlSqInfluenceRadius))
        self.nl, self.neibs_info = o.build_neibs(self.pos, self.info, self.hash, self.cs, self.ce, self.n, self.n, sq)

    def repack_step(self):
        """RepackingIntegrator::initializeRepackingSequence (src/integrators/RepackingIntegrator.cc:278-420)"""
        sp = self.problem.simparams
        o = self.o
        if self.iterations % sp.buildneibsfreq == 0:
            self.build_neibs()
        n = self.n
        dt = float(np.float32(self.dt))
        rb = getattr(self.problem, "num_obstacle", 0)
        f, cfl, nb, self.rbf, self.rbt = o.repack_forces(self.pos, self.vel, self.info, self.hash, self.cs, self.nl, n, rb_count=rb)
        dt1 = o.dtreduce(cfl, nb, self.sspeed_cfl, self.max_kinvisc)
        self.pos, self.vel = o.euler_repack(self.pos, self.vel, self.info, self.hash, f, n, dt, 1)
        self.forces = f
        self.t += dt
        self.iterations += 1
        self.dt = dt1

    def repack(self, maxiter=None, reset=True):
        sp = self.problem.simparams
        for _ in range(int(sp.repack_maxiter if maxiter is None else maxiter)):
            self.repack_step()
        self.o.disable_free_surf_parts(self.pos, self.info, self.n)
        if self.iterations > 0:
            self.build_neibs()
        if not reset:
            return
        self.iterations = 0
        self.t = 0.0
        self.dt = float(np.float32(sp.dt))
        n = self.n
        rho = self.problem.initial_density(self.problem.global_pos(self.pos[:n], self.hash[:n]))
        self.vel[:n] = 0
        self.vel[:n, 3] = rho

    def step(self):
        sp = self.problem.simparams
        o = self.o
        if self.iterations % sp.buildneibsfreq == 0:
            self.build_neibs()
        n = self.n
        if self.iterations > 0:   # FILTER phases, PredictorCorrectorIntegrator.cc:1011-1041
            for ftype, freq in getattr(self, "filters", []):
                if self.iterations % freq == 0:
                    self.vel = o.filter(ftype, self.pos, self.vel, self.info, self.hash, self.cs, self.nl, n)
        cof = 1 if sp.numforcesbodies > 0 else 0
        rb = getattr(self.problem, "num_obstacle", 0)
        dt = float(np.float32(self.dt))
        sps = sp.turbmodel == self.D.SPS
        # predictor (CALC_VISC before the forces when SPS, PredictorCorrectorIntegrator.cc:460-480)
        tau = o.sps(self.pos, self.vel, self.info, self.hash, self.cs, self.nl, n, n)[0] if sps else None
        if self.grenier:
            return self._grenier_step(n, dt)
        ev = None
        if self.effvisc_on:    # CALC_VISC: effective viscosity of the state the forces read; its maximum limits dt (GPUWorker.cc:2633-2645)
            ev, self.max_kinvisc = o.effective_visc(self.pos, self.vel, self.info, self.hash, self.cs, self.nl, n)
            self.effvisc = ev
        f1, cfl, nb, self.rbf, self.rbt = o.forces(self.pos, self.vel, self.info, self.hash, self.cs, self.nl, n,
                                                   compute_object_forces=cof, rb_count=rb, tau=tau, effvisc=ev,
                                                   dedt=self.dedt if self.energy_on else None)
        dt1 = o.dtreduce(cfl, nb, self.sspeed_cfl, self.max_kinvisc)
        xsph_on = bool(sp.simflags & self.D.ENABLE_XSPH)
        if xsph_on:     # BUFFER_XSPH is allocated once and rewritten for the fluid particles by every forces pass
            self.xsph = o.xsph(self.pos, self.vel, self.info, self.hash, self.cs, self.nl, n, getattr(self, "xsph", None))
        xs = self.xsph if xsph_on else None
        if self.bodies is not None:
            self._move_bodies(1, dt, self.t)
        ps, vs = o.euler(self.pos, self.vel, self.info, self.hash, f1, n, float(np.float32(dt) / np.float32(2)), 1, xsph=xs)
        if self.energy_on:
            es = o.euler_energy(self.energy, self.dedt, self.pos, self.info, n, float(np.float32(dt) / np.float32(2)))
        # corrector
        tau = o.sps(ps, vs, self.info, self.hash, self.cs, self.nl, n, n)[0] if sps else None
        if self.effvisc_on:
            ev, self.max_kinvisc = o.effective_visc(ps, vs, self.info, self.hash, self.cs, self.nl, n)
        f2, cfl, nb, self.rbf, self.rbt = o.forces(ps, vs, self.info, self.hash, self.cs, self.nl, n,
                                                   compute_object_forces=cof, rb_count=rb, tau=tau, effvisc=ev,
                                                   dedt=self.dedt if self.energy_on else None)
        dt2 = o.dtreduce(cfl, nb, self.sspeed_cfl, self.max_kinvisc)
        if xsph_on:
            self.xsph = o.xsph(ps, vs, self.info, self.hash, self.cs, self.nl, n, self.xsph)
        if self.bodies is not None:
            self._move_bodies(2, dt, self.t)
        if self.energy_on:      # the corrector integrates from the energy of step n with the derivative of n*
            self.energy = o.euler_energy(self.energy, self.dedt, self.pos, self.info, n, dt)
        self.pos, self.vel = o.euler(self.pos, self.vel, self.info, self.hash, f2, n, dt, 2, xsph=xs)
        if self.bodies is not None:
            m = self._last_motion
            for b in range(len(self.bodies)):
                for a in range(3):
                    self.o.p.rbcgGridPosE[b][a] = int(m["cg_grid"][b][a]); self.o.p.rbcgPosE[b][a] = float(m["cg_pos"][b][a])
        self.forces = f2
        self.t += dt
        self.iterations += 1
        self.dt = min(dt1, dt2)

    def _grenier_step(self, n, dt):
        """predictor/corrector with SPH_GRENIER: COMPUTE_DENSITY rewrites the density of the state the forces are computed
        on (PredictorCorrectorIntegrator.cc:443-458), Euler integrates the volume"""
        o = self.o
        mfb = self.neibs_info.maxFluidBoundaryNeibs
        self.sigma = o.density_grenier(self.pos, self.vel, self.info, self.hash, self.vol, self.cs, self.nl, n, mfb)
        f1, cfl, nb = o.forces_grenier(self.pos, self.vel, self.info, self.hash, self.cs, self.nl, self.sigma, n)
        dt1 = o.dtreduce(cfl, nb, self.sspeed_cfl, self.max_kinvisc)
        ps, vs, vols = o.euler_grenier(self.pos, self.vel, self.vol, self.info, self.hash, f1, n,
                                       float(np.float32(dt) / np.float32(2)), 1)
        self.sigma_s = o.density_grenier(ps, vs, self.info, self.hash, vols, self.cs, self.nl, n, mfb)
        f2, cfl, nb = o.forces_grenier(ps, vs, self.info, self.hash, self.cs, self.nl, self.sigma_s, n)
        dt2 = o.dtreduce(cfl, nb, self.sspeed_cfl, self.max_kinvisc)
        self.pos, self.vel, self.vol = o.euler_grenier(self.pos, self.vel, self.vol, self.info, self.hash, f2, n, dt, 2)
        self.forces = f2
        self.t += dt
        self.iterations += 1
        self.dt = min(dt1, dt2)
