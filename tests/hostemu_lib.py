"""TEST HARNESS ONLY: sphx_api.hip + sa_io.hip compiled for the HOST through the stand-in tests/hostemu/hip/hip_runtime.h, so that
the open-boundary kernels that have not run on a GPU yet can be held against the oracle on the CPU, one emulated thread after the
other.  It finds logic errors (indices, flags, signs, operation order); it says nothing about the device build, is never imported
by the package, and is no CPU path of the library.  Block-wide reductions (the CFL maxima) are not emulated."""
import ctypes as C
import os
import subprocess

import numpy as np

from gpusph_amd import capi

_HERE = os.path.dirname(os.path.abspath(__file__))
_EMU = os.path.join(_HERE, "hostemu")
_ROOT = os.path.dirname(_HERE)
_SO = os.path.join(_EMU, "_build", "libsphx_emu.so")
_SOURCES = [os.path.join(_EMU, "emu_sphx.cc"), os.path.join(_EMU, "hip", "hip_runtime.h")] + \
    [os.path.join(_ROOT, "gpusph_amd", "csrc", f) for f in ("sphx_api.hip", "sa_io.hip", "sphx_internal.h", "neib_iter.h",
                                                              "sa_wall_gamma.h", "sa_args.h")] + \
    [os.path.join(_ROOT, "include", "sphx.h")]

# the entry points of the two files
NAMES = ["sphx_create", "sphx_destroy", "sphx_set_constants", "sphx_last_error",
         "sphx_sa_identify_corner_vertices", "sphx_sa_init_io_mass_vertex_count", "sphx_sa_init_io_mass",
         "sphx_sa_find_outgoing_segment", "sphx_sa_disable_outgoing_parts", "sphx_sa_segment_bc_io", "sphx_sa_vertex_bc_io",
         "sphx_sa_density_sum_io", "sphx_forces_basicstep_sa_io", "sphx_sa_compute_density_diffusion_io", "sphx_sa_io_water_depth"]


def build():
    os.makedirs(os.path.dirname(_SO), exist_ok=True)
    if os.path.exists(_SO) and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in _SOURCES):
        return _SO
    # -ffp-contract=off as the library's own build of these files; -O1: compile time
    cmd = ["g++", "-O1", "-g0", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-Wno-attributes",
           "-I" + _EMU, "-I" + os.path.join(_ROOT, "include"), "-I" + os.path.join(_ROOT, "gpusph_amd", "csrc"),
           "-o", _SO, os.path.join(_EMU, "emu_sphx.cc")]
    subprocess.run(cmd, check=True, capture_output=True)
    return _SO


class Emu:
    """the emulated library with a context whose constants are set: emu.call("sphx_...", args...) with numpy arrays for buffers"""

    def __init__(self, params):
        self.lib = C.CDLL(build())
        for name in NAMES:
            res, args = capi.SIGNATURES[name]
            fn = getattr(self.lib, name)
            fn.restype, fn.argtypes = res, args
        self.h = C.c_void_p()
        self._check(self.lib.sphx_create(C.byref(self.h), 0))
        self.params = params
        self._check(self.lib.sphx_set_constants(self.h, C.byref(params)))

    def _check(self, rc):
        if rc != capi.SPHX_OK:
            raise RuntimeError("emulated libsphx: rc %d: %s" % (rc, self.lib.sphx_last_error().decode()))

    def call(self, name, *args):
        conv = []
        for a in args:
            if isinstance(a, np.ndarray):
                assert a.flags["C_CONTIGUOUS"]
                conv.append(a.ctypes.data)
            else:
                conv.append(a)
        self._check(getattr(self.lib, name)(self.h, *conv))

    def close(self):
        if self.h:
            self.lib.sphx_destroy(self.h)
            self.h = None
