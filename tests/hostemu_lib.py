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
_SO = os.path.join(_EMU, "_build", "libsphx_emu_asan.so" if os.environ.get("SPHX_HOSTEMU_ASAN") == "1" else "libsphx_emu.so")
_SOURCES = [os.path.join(_EMU, "emu_sphx.cc"), os.path.join(_EMU, "hip", "hip_runtime.h")] + \
    [os.path.join(_ROOT, "gpusph_amd", "csrc", f) for f in ("sphx_api.hip", "sa_io.hip", "sa_bounds.hip", "euler.hip", "neibs_build.hip", "sphx_internal.h",
                                                              "neib_iter.h", "wave_list.h", "sa_wall_gamma.h", "sa_args.h")] + \
    [os.path.join(_ROOT, "include", "sphx.h")]

# the entry points of the two files
NAMES = ["sphx_create", "sphx_destroy", "sphx_set_constants", "sphx_last_error",
         "sphx_sa_identify_corner_vertices", "sphx_sa_init_io_mass_vertex_count", "sphx_sa_init_io_mass",
         "sphx_sa_find_outgoing_segment", "sphx_sa_disable_outgoing_parts", "sphx_sa_segment_bc_io", "sphx_sa_vertex_bc_io",
         "sphx_sa_density_sum_io", "sphx_sa_density_sum_io_moving", "sphx_forces_basicstep_sa_io", "sphx_sa_compute_density_diffusion_io", "sphx_sa_io_water_depth", "sphx_flux_computation",
         # sa_bounds.hip, list walkers only
         "sphx_sa_compute_vertex_normal", "sphx_sa_init_gamma", "sphx_sa_segment_bc", "sphx_sa_vertex_bc", "sphx_sa_density_sum",
         "sphx_sa_compute_density_diffusion", "sphx_apply_density_diffusion", "sphx_sa_integrate_gamma", "sphx_forces_basicstep_sa",
         "sphx_sa_density_sum_moving",
         # euler.hip
         "sphx_sa_update_normals", "sphx_set_rb_motion"]


def _rewrite_launches(src, dst):
    """kernel<<<grid, block, 0, stream>>>(args -> SPHX_EMU_LAUNCH((kernel), grid, block, args  (the source stays as it is)"""
    import re
    text = open(src).read()
    pat = re.compile(r'(\b\w+(?:<[^<>;()]*>)?)<<<(.+?), (\w+), 0, ([^>;]+?)>>>\(')
    out, n = pat.subn(lambda m: 'SPHX_EMU_LAUNCH((%s), %s, %s, ' % (m.group(1), m.group(2), m.group(3)), text)
    assert n > 0 and '<<<' not in out, "a launch of %s was not understood" % src
    with open(dst, "w") as f:
        f.write(out)


def build():
    os.makedirs(os.path.dirname(_SO), exist_ok=True)
    if os.path.exists(_SO) and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in _SOURCES):
        return _SO
    _rewrite_launches(os.path.join(_ROOT, "gpusph_amd", "csrc", "sa_bounds.hip"), os.path.join(_EMU, "_build", "sa_bounds_emu.inc"))
    _rewrite_launches(os.path.join(_ROOT, "gpusph_amd", "csrc", "euler.hip"), os.path.join(_EMU, "_build", "euler_emu.inc"))
    # -ffp-contract=off as the library's own build of these files; -O1: compile time
    # SPHX_HOSTEMU_ASAN=1: an address-sanitised build, for a run under LD_PRELOAD=libasan.so (out-of-bounds reads and writes of the
    # kernels on the exact-size numpy buffers of these tests; see tests/hostemu/README)
    san = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-g"] if os.environ.get("SPHX_HOSTEMU_ASAN") == "1" else []
    cmd = ["g++"] + (["-DSPHX_EMU_DEBUG"] if os.environ.get("SPHX_EMU_DEBUG") else []) + ["-O1", "-g0", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-Wno-attributes"] + san + [
           "-I" + _EMU, "-I" + os.path.join(_EMU, "_build"), "-I" + os.path.join(_ROOT, "include"), "-I" + os.path.join(_ROOT, "gpusph_amd", "csrc"),
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


class EmuPasses:
    """The oracle with its open-boundary passes replaced by the emulated kernels: the same Python signatures as tests/oracle_lib.py,
    so that tests/sa_helpers.py OracleSaIoSim runs its whole sequence over the kernels' source.  Everything else (hashing, sort,
    reorder, neighbour list, Euler, time step) stays the oracle's; so do the CFL maxima of the forces (block reductions are not
    emulated)."""

    def __init__(self, oracle, emu):
        self._o, self._e = oracle, emu
        self.calls = {}

    def __getattr__(self, name):          # whatever is not overridden below
        return getattr(self._o, name)

    def _call(self, name, *args):
        self.calls[name] = self.calls.get(name, 0) + 1
        self._e.call(name, *args)

    @staticmethod
    def _vp(vertpos):
        return [np.ascontiguousarray(v) for v in vertpos]

    def sa_identify_corner_vertices(self, pos, info, hash_, vertices, cs, nl, n):
        out = info.copy()
        self._call("sphx_sa_identify_corner_vertices", pos, out, hash_, vertices, cs, nl, n, n, None)
        return out

    def sa_init_io_mass(self, pos, info, hash_, vertices, cs, nl, n, deltap):
        forces = np.zeros((len(pos), 4), dtype=np.float32)
        new_pos = np.zeros_like(pos)
        self._call("sphx_sa_init_io_mass_vertex_count", vertices, hash_, info, cs, nl, forces, pos, n, n, None)
        self._call("sphx_sa_init_io_mass", pos, forces, vertices, hash_, info, cs, nl, new_pos, n, n, float(np.float32(deltap)), None)
        return forces[:, 3].copy(), new_pos

    def sa_segment_bc_io(self, pos, vel, ggam, euler_vel, vertices, boundelements, info, hash_, cs, nl, n, step):
        v, g, e = vel.copy(), ggam.copy(), euler_vel.copy()
        self._call("sphx_sa_segment_bc_io", v, g, e, pos, vertices, boundelements, info, hash_, cs, nl, n, n, int(step), None)
        return v, g, e

    def find_outgoing_segment(self, pos, vel, vertices, ggam, vertpos, boundelements, info, hash_, cs, nl, n, influenceradius):
        v, g = vertices.copy(), ggam.copy()
        vp = self._vp(vertpos)
        self._call("sphx_sa_find_outgoing_segment", pos, vel, v, g, vp[0], vp[1], vp[2], boundelements, info, hash_, cs, nl, n, n,
                   float(np.float32(influenceradius)), None)
        return v, g

    def sa_vertex_bc_io(self, pos, vel, ggam, euler_vel, vertices, boundelements, vertpos, info, hash_, next_ids, cs, nl, n,
                        deltap, dt, step, num_open_vertices, room=None):
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
        count = np.array([n], dtype=np.uint32)
        vp = self._vp(vertpos)
        self._call("sphx_sa_vertex_bc_io", a["vel"], old_pos, a["new_pos"], a["ggam"], a["euler_vel"], a["forces"], a["vertices"],
                   a["boundelements"], vp[0], vp[1], vp[2], a["info"], a["hash"], a["next_ids"], count, cs, nl, n, n, tot,
                   float(np.float32(deltap)), float(np.float32(dt)), int(step), int(num_open_vertices), None)
        a["n"] = int(count[0])
        return a

    def disable_outgoing_parts(self, pos, vertices, info, n):
        p2, v2 = pos.copy(), vertices.copy()
        self._call("sphx_sa_disable_outgoing_parts", p2, v2, info, n, None)
        return p2, v2

    def sa_density_sum_io(self, new_vel, old_pos, new_pos, old_vel, old_euler_vel, old_ggam, boundelements, vertpos, info, hash_,
                          cs, nl, n, dt):
        v, g = new_vel.copy(), old_ggam.copy()
        scratch = np.zeros((len(old_pos), 4), dtype=np.float32)
        vp = self._vp(vertpos)
        self._call("sphx_sa_density_sum_io", v, g, scratch, old_pos, new_pos, old_vel, old_euler_vel, old_ggam, boundelements,
                   vp[0], vp[1], vp[2], info, hash_, cs, nl, n, n, float(np.float32(dt)), None)
        return v, g, scratch[:, 3].copy()

    def forces_sa_io(self, pos, vel, euler_vel, info, hash_, cs, nl, ggam, boundelements, vertpos, n, deltap):
        _, cfl, nb = self._o.forces_sa_io(pos, vel, euler_vel, info, hash_, cs, nl, ggam, boundelements, vertpos, n, deltap)
        self.max_gamma_cfl = self._o.max_gamma_cfl
        forces = np.zeros((len(pos), 4), dtype=np.float32)
        scr_cfl = np.zeros(len(cfl) * 4 + 64, dtype=np.float32)
        scr_g = np.zeros(((n + 3) // 4) * 4 + len(scr_cfl), dtype=np.float32)
        hnb = C.c_uint32(0)
        vp = self._vp(vertpos)
        self._call("sphx_forces_basicstep_sa_io", forces, scr_cfl, scr_g, pos, vel, euler_vel, info, hash_, cs, nl, ggam, boundelements,
                   vp[0], vp[1], vp[2], n, 0, n, float(np.float32(deltap)), 0, C.addressof(hnb), None)
        assert hnb.value == nb
        return forces, cfl, nb

    def sa_density_diffusion_io(self, pos, vel, ggam, info, hash_, cs, nl, boundelements, vertpos, n, dt, deltap):
        f = np.zeros((len(pos), 4), dtype=np.float32)
        vp = self._vp(vertpos)
        self._call("sphx_sa_compute_density_diffusion_io", f, pos, vel, ggam, boundelements, vp[0], vp[1], vp[2], info, hash_, cs, nl,
                   n, n, float(np.float32(deltap)), float(np.float32(dt)), None)
        v = vel.copy()
        fluid = (info[:n, 0] & 7) == 0
        v[:n][fluid, 3] = v[:n][fluid, 3] + f[:n][fluid, 3] * np.float32(dt)
        return v, f

    def sa_io_water_depth(self, depth, pos, info, hash_, cs, nl, n, frm=0):
        self._call("sphx_sa_io_water_depth", depth, pos, info, hash_, cs, nl, n, int(frm), n, None)
        return depth
