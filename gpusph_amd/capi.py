"""ctypes binding of the C ABI in include/sphx.h (libsphx.so).

There is NO fallback: if the HIP library is missing or a call fails, this raises.  The oracle
under oracle/ is test infrastructure and is never imported from here.
"""
import ctypes as C
import os
from .params import SphxParams

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SPHX_LIB") or os.path.join(_HERE, "libsphx.so")      # SPHX_LIB: A/B runs of another build

SPHX_OK, SPHX_ERR_INVALID, SPHX_ERR_RUNTIME, SPHX_ERR_UNSUPPORTED = 0, -1, -2, -3


class SphxError(RuntimeError):
    """std::runtime_error of CUDA_SAFE_CALL / KERNEL_CHECK_ERROR (src/cuda/cuda_call.h:57-85)."""


class SphxInvalidArgument(ValueError):
    """std::invalid_argument of the reference engines (e.g. src/cuda/buildneibs.cu:444-455)."""


class SphxUnsupported(NotImplementedError):
    """option combination not built into libsphx."""


class NeibsInfo(C.Structure):
    _fields_ = [("numInteractions", C.c_int32), ("maxFluidBoundaryNeibs", C.c_int32),
                ("maxVertexNeibs", C.c_int32), ("hasTooManyNeibs", C.c_int32),
                ("hasMaxNeibs", C.c_int32 * 3)]


_vp, _u32, _f, _i = C.c_void_p, C.c_uint32, C.c_float, C.c_int

# name -> (restype, argtypes); every symbol declared in include/sphx.h
SIGNATURES = {
    "sphx_last_error": (C.c_char_p, []),
    "sphx_version": (C.c_char_p, []),
    "sphx_create": (_i, [C.POINTER(_vp), _i]),
    "sphx_destroy": (None, [_vp]),
    "sphx_reserve": (_i, [_vp, _u32]),
    "sphx_neibs_interactions64": (_i, [_vp, C.POINTER(C.c_uint64), _vp]),
    "sphx_set_constants": (_i, [_vp, C.POINTER(SphxParams)]),
    "sphx_get_params": (_i, [_vp, C.POINTER(SphxParams)]),
    "sphx_set_planes": (_i, [_vp, _vp, _vp, _vp, _i]),
    "sphx_set_dem": (_i, [_vp, _vp, _i, _i]),
    "sphx_set_gravity": (_i, [_vp, C.POINTER(_f)]),
    "sphx_set_rb_cg": (_i, [_vp, _vp, _vp, _i]),
    "sphx_set_rb_cg_forces": (_i, [_vp, _vp, _vp, _i]),
    "sphx_set_rb_cg_integration": (_i, [_vp, _vp, _vp, _i]),
    "sphx_set_rb_start": (_i, [_vp, _vp, _i]),
    "sphx_set_rb_motion": (_i, [_vp, _vp, _vp, _vp, _vp, _i]),
    "sphx_calc_hash": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp]),
    "sphx_fix_hash": (_i, [_vp, _vp, _vp, _vp, _vp, _u32, _vp]),
    "sphx_sort": (_i, [_vp, _vp, _vp, _vp, _u32, _vp]),
    "sphx_reorder": (_i, [_vp] + [_vp] * 10 + [_u32, _vp, _vp]),
    "sphx_gather_rows": (_i, [_vp, _vp, _vp, _u32, _vp, _u32, _vp]),
    "sphx_find_cell_start": (_i, [_vp, _vp, _vp, _vp, _u32, _u32, _vp]),
    "sphx_build_neibs": (_i, [_vp] + [_vp] * 6 + [_u32, _u32, _u32, _f, _f, _vp]),
    "sphx_build_neibs_sa": (_i, [_vp] + [_vp] * 11 + [_u32, _u32, _u32, _f, _f, _vp]),
    "sphx_sa_compute_vertex_normal": (_i, [_vp] + [_vp] * 6 + [_u32, _u32, _vp]),
    "sphx_sa_identify_corner_vertices": (_i, [_vp] + [_vp] * 6 + [_u32, _u32, _vp]),
    "sphx_sa_init_io_mass_vertex_count": (_i, [_vp] + [_vp] * 7 + [_u32, _u32, _vp]),
    "sphx_sa_init_io_mass": (_i, [_vp] + [_vp] * 8 + [_u32, _u32, _f, _vp]),
    "sphx_sa_find_outgoing_segment": (_i, [_vp] + [_vp] * 12 + [_u32, _u32, _f, _vp]),
    "sphx_sa_disable_outgoing_parts": (_i, [_vp] + [_vp] * 3 + [_u32, _vp]),
    "sphx_sa_segment_bc_io": (_i, [_vp] + [_vp] * 10 + [_u32, _u32, _i, _vp]),
    "sphx_sa_vertex_bc_io": (_i, [_vp] + [_vp] * 17 + [_u32, _u32, _u32, _f, _f, _i, _u32, _vp]),
    "sphx_sa_density_sum_io": (_i, [_vp] + [_vp] * 16 + [_u32, _u32, _f, _vp]),
    "sphx_sa_density_sum_io_moving": (_i, [_vp] + [_vp] * 17 + [_u32, _u32, _f, _vp]),
    "sphx_forces_basicstep_sa_io": (_i, [_vp] + [_vp] * 15 + [_u32, _u32, _u32, _f, _u32, _vp, _vp]),
    "sphx_sa_compute_density_diffusion_io": (_i, [_vp] + [_vp] * 12 + [_u32, _u32, _f, _f, _vp]),
    "sphx_flux_computation": (_i, [_vp] + [_vp] * 4 + [_u32, _u32, _u32, _vp]),
    "sphx_sa_io_water_depth": (_i, [_vp] + [_vp] * 6 + [_u32, _u32, _u32, _vp]),
    "sphx_sa_body_pressure_forces": (_i, [_vp] * 9 + [_u32, _u32, _vp]),
    "sphx_forces_basicstep_sa": (_i, [_vp] + [_vp] * 14 + [_u32, _u32, _u32, _f, _f, _f, _f, _u32, _i, _i, _f, _vp, _vp]),
    "sphx_forces_dtreduce_gamma_device": (_i, [_vp, _vp, _u32, _u32, _vp, _vp]),
    "sphx_forces_dtreduce_gamma": (_i, [_vp, _vp, _u32, _u32, _vp, _vp]),
    "sphx_sa_density_sum": (_i, [_vp] + [_vp] * 15 + [_u32, _u32, _f, _i, _f, _f, _f, _f, _f, _vp]),
    "sphx_sa_compute_density_diffusion": (_i, [_vp] + [_vp] * 8 + [_u32, _u32, _f, _f, _f, _f, _vp]),
    "sphx_apply_density_diffusion": (_i, [_vp, _vp, _vp, _vp, _u32, _u32, _f, _vp]),
    "sphx_sa_update_normals": (_i, [_vp, _vp, _vp, _vp, _u32, _u32, _vp]),
    "sphx_sa_density_sum_moving": (_i, [_vp] + [_vp] * 16 + [_u32, _u32, _vp]),
    "sphx_sa_integrate_gamma": (_i, [_vp] + [_vp] * 11 + [_u32, _u32, _f, _i, _f, _f, _f, _f, _i, _vp]),
    "sphx_sa_init_gamma": (_i, [_vp] + [_vp] * 11 + [_f, _f, _f, _f, _u32, _u32, _vp]),
    "sphx_sa_segment_bc": (_i, [_vp] + [_vp] * 9 + [_u32, _u32, _f, _f, _f, _i, _i, _vp]),
    "sphx_sa_vertex_bc": (_i, [_vp] + [_vp] * 7 + [_u32, _u32, _f, _f, _f, _i, _i, _vp]),
    "sphx_sa_segment_bc_keps": (_i, [_vp] + [_vp] * 12 + [_u32, _u32, _f, _f, _f, _i, _i, _vp]),
    "sphx_sa_vertex_bc_keps": (_i, [_vp] + [_vp] * 12 + [_u32, _u32, _f, _f, _f, _i, _i, _vp]),
    "sphx_forces_basicstep_sa_keps": (_i, [_vp] + [_vp] * 20 + [_u32, _u32, _u32, _f, _f, _f, _f, _f, _u32, _i, _i, _f, _vp, _vp]),
    "sphx_euler_keps": (_i, [_vp] + [_vp] * 11 + [_u32, _u32, _f, _vp, _f, _vp]),
    "sphx_forces_dtreduce_keps_device": (_i, [_vp, _vp, _u32, _f, _f, _vp, _vp]),
    "sphx_forces_dtreduce_keps": (_i, [_vp, _vp, _u32, _f, _f, _vp, _vp]),
    "sphx_neibs_resetinfo": (_i, [_vp, _vp]),
    "sphx_neibs_getinfo": (_i, [_vp, C.POINTER(NeibsInfo), _vp]),
    "sphx_forces_fmax_elements": (_u32, [_u32]),
    "sphx_forces_fmax_temp_elements": (_u32, [_u32]),
    "sphx_forces_round_particles": (_u32, [_u32]),
    "sphx_forces_basicstep": (_i, [_vp] + [_vp] * 14 + [_u32, _u32, _u32, _f, _f, _f, _f, _u32, _i, _i, _f, _i,
                                                       C.POINTER(_u32), _vp]),
    "sphx_forces_dtreduce": (_i, [_vp, _f, _f, _f, _f, _vp, _vp, _u32, C.POINTER(_f), _vp]),
    "sphx_forces_dtreduce_device": (_i, [_vp, _f, _f, _f, _f, _vp, _vp, _u32, _vp, _i, _vp]),
    "sphx_reduce_rb_forces": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _u32, _vp]),
    "sphx_forces_reserve_cus": (_i, [_vp, _u32]),
    "sphx_forces_timing": (_i, [_vp, _i]),
    "sphx_forces_timing_read": (_i, [_vp, C.POINTER(C.c_double), C.POINTER(_u32)]),
    "sphx_calc_visc": (_i, [_vp] + [_vp] * 10 + [_u32, _u32, _f, _f, _f, _vp]),
    "sphx_filter_process": (_i, [_vp, _i] + [_vp] * 7 + [_u32, _u32, _f, _f, _vp]),
    "sphx_postprocess": (_i, [_vp, _i] + [_vp] * 10 + [_u32, _u32, _f, _f, _vp]),
    "sphx_euler_basicstep": (_i, [_vp] + [_vp] * 8 + [_u32, _u32, _f, _vp, _f, _i, _f, _f, _f, _i, _vp]),
    "sphx_euler_basicstep_grenier": (_i, [_vp] + [_vp] * 10 + [_u32, _u32, _f, _vp, _f, _i, _f, _f, _f, _i, _vp]),
    "sphx_init_volume": (_i, [_vp] * 5 + [_u32, _vp]),
    "sphx_forces_internal_energy": (_i, [_vp] * 8 + [_u32, _u32, _u32, _vp]),
    "sphx_euler_internal_energy": (_i, [_vp] * 6 + [_u32, _u32, _f, _vp, _f, _vp]),
    "sphx_calc_effvisc": (_i, [_vp] * 9 + [_u32, _u32, _f, _f, _f, _vp]),
    "sphx_forces_basicstep_effvisc": (_i, [_vp] * 10 + [_u32, _u32, _u32, _f, _f, _f, _f, _u32, _i, _i, _f, _vp, _vp]),
    "sphx_compute_density": (_i, [_vp] * 9 + [_u32, _f, _f, _vp]),
    "sphx_forces_basicstep_grenier": (_i, [_vp] * 10 + [_u32, _u32, _u32, _f, _f, _f, _f, _u32, _i, _i, _f, _vp, _vp]),
    "sphx_disable_free_surf_parts": (_i, [_vp, _vp, _vp, _u32, _u32, _vp]),
    "sphx_time_advance": (_i, [_vp, _vp, _vp, _vp]),
    "sphx_eos_rows_follow_euler": (_i, [_vp, _i]),
    "sphx_eos_rows_current": (_i, [_vp, _vp, _u32]),
    "sphx_halo_group_create": (_i, [_i, C.POINTER(_vp)]),
    "sphx_halo_group_destroy": (_i, [_vp]),
    "sphx_halo_create_threads": (_i, [_vp, _vp, _i, C.POINTER(_vp)]),
    "sphx_halo_unique_id": (_i, [_vp]),
    "sphx_halo_create_rccl": (_i, [_vp, _vp, _i, _i, C.POINTER(_vp)]),
    "sphx_halo_destroy": (_i, [_vp]),
    "sphx_halo_exchange": (_i, [_vp, _i, _vp, _vp, _i, _u32, _u32, _u32, _u32, _i, _u32, _u32, _u32, _u32, _vp]),
    "sphx_halo_allreduce_min_f32": (_i, [_vp, _vp, _vp]),
    "sphx_halo_allreduce_sum_f32": (_i, [_vp, _vp, _u32, _vp]),
    "sphx_halo_allreduce_sum_f64": (_i, [_vp, _vp, _u32, _vp]),
    "sphx_halo_allgather_u64x2": (_i, [_vp, _vp, _vp, _vp]),
    "sphx_halo_barrier": (_i, [_vp, _vp]),
    "sphx_memset_async": (_i, [_vp, _i, C.c_size_t, _vp]),
    "sphx_device_count": (_i, [C.POINTER(_i)]),
    "sphx_set_device": (_i, [_i]),
    "sphx_get_device": (_i, [C.POINTER(_i)]),
    "sphx_device_synchronize": (_i, []),
    "sphx_malloc": (_i, [C.POINTER(_vp), C.c_size_t]),
    "sphx_free": (_i, [_vp]),
    "sphx_memset": (_i, [_vp, _i, C.c_size_t]),
    "sphx_memcpy_h2d": (_i, [_vp, _vp, C.c_size_t]),
    "sphx_memcpy_d2h": (_i, [_vp, _vp, C.c_size_t]),
    "sphx_memcpy_d2d": (_i, [_vp, _vp, C.c_size_t]),
}

_lib = None


def _hip_runtimes_loaded():
    """paths of every libamdhip64 mapped into this process"""
    libs = set()
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamdhip64" in line:
                    libs.add(line.split()[-1])
    except OSError:
        pass
    return libs


def load(path=None):
    """dlopen libsphx.so and type every entry point.  Raises if the library is absent.

    torch is imported FIRST on purpose: its wheel bundles a libamdhip64.so with the same SONAME as
    /opt/rocm's, and device pointers / hipStream_t handles are shared between torch and libsphx, so
    both must resolve to ONE HIP runtime instance.  Loading libsphx first would map a second runtime
    (observed: "no ROCm-capable device is detected"); that situation is detected and refused.
    """
    global _lib
    if _lib is not None and path is None:
        return _lib
    import torch  # noqa: F401  (device memory / stream provider; must own the HIP runtime)
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise SphxError("libsphx.so not found at %s: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(there is no CPU fallback for the product path)" % p)
    lib = C.CDLL(p)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    rts = _hip_runtimes_loaded()
    if len(rts) > 1:
        raise SphxError("two HIP runtimes are mapped into this process (%s): import torch before anything "
                        "that links libamdhip64" % ", ".join(sorted(rts)))
    if path is None:
        _lib = lib
    return lib


def check(rc):
    if rc == SPHX_OK:
        return
    msg = load().sphx_last_error().decode("utf-8", "replace")
    if rc == SPHX_ERR_INVALID:
        raise SphxInvalidArgument(msg)
    if rc == SPHX_ERR_UNSUPPORTED:
        raise SphxUnsupported(msg)
    raise SphxError(msg)


def ptr(t):
    """device pointer of a torch tensor (or None / int passthrough)."""
    if t is None:
        return None
    if isinstance(t, int):
        return t
    return t.data_ptr()


class Context:
    """One per device: the per-device constant state of the reference engines."""

    def __init__(self, device=0):
        self.lib = load()
        h = C.c_void_p()
        check(self.lib.sphx_create(C.byref(h), int(device)))
        self.handle = h
        self.device = int(device)

    def close(self):
        if getattr(self, "handle", None):
            self.lib.sphx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_constants(self, params: SphxParams):
        check(self.lib.sphx_set_constants(self.handle, C.byref(params)))
        self.params = params

    def reserve(self, max_particles):
        check(self.lib.sphx_reserve(self.handle, int(max_particles)))
