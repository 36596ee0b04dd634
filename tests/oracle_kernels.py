"""TEST-ONLY kernel backend: the interface of gpusph_amd.kernels.HipKernels implemented with the CPU
oracle on CPU torch tensors.  It exists so that the host logic of gpusph_amd.multigpu (slab partition,
segments, halo exchange order, index bookkeeping) can run under torch.distributed/gloo without a GPU.
It lives under tests/ on purpose: the product never imports it."""
import ctypes as C
import numpy as np
import torch

import oracle_lib as ol


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class OracleKernels:
    def __init__(self, problem, alloc):
        self._problem = problem
        self.sp = problem.sphx_params(alloc)
        self.op = ol.orc_params_from(self.sp, problem)
        self.L = ol.lib()
        self.ncells = problem.grid_cells
        s = problem.simparams
        self.compute_object_forces = 1 if s.numforcesbodies > 0 else 0
        self.sq = float(np.float32(s.nlSqInfluenceRadius))
        self.sspeed_cfl = float(np.float32(np.float64(np.float32(max(problem.physparams.sscoeff))) * 1.1))
        self.info_ = ol.OrcNeibsInfo()
        self.max_kinvisc = float(np.float32(max(problem.physparams.kinematicvisc))) if s.rheologytype == 1 else 0.0

    def fmax_elements(self, n):
        return int(self.L.orc_fmax_elements(C.c_uint32(n)))

    def fmax_temp_elements(self, n):
        return int(self.L.orc_fmax_temp_elements(C.c_uint32(n)))

    def memset(self, t, byte):
        t.view(torch.uint8).fill_(byte) if t.numel() else None

    def calc_hash(self, pos, hash_, partindex, info, devmap, n):
        self.L.orc_calc_hash(C.byref(self.op), _p(pos), _p(hash_), _p(partindex), _p(info), _p(devmap), C.c_uint32(n))

    def fix_hash(self, hash_, partindex, info, devmap, n):
        self.L.orc_fix_hash(C.byref(self.op), _p(hash_), _p(partindex), _p(info), _p(devmap), C.c_uint32(n))

    def sort(self, hash_, info, partindex, n):
        self.L.orc_sort(_p(hash_), _p(info), _p(partindex), C.c_uint32(n))

    def reorder(self, segment_start, cellStart, cellEnd, spos, svel, upos, uvel, info, hash_, partindex, n, new_num):
        self.L.orc_reorder(C.byref(self.op), _p(cellStart), _p(cellEnd), _p(segment_start), _p(spos), _p(svel), _p(upos),
                           _p(uvel), _p(info), _p(hash_), _p(partindex), C.c_uint32(n), _p(new_num))

    def find_cell_start(self, cellStart, cellEnd, hash_, frm, to):
        self.L.orc_find_cell_start(_p(cellStart), _p(cellEnd), _p(hash_), C.c_uint32(frm), C.c_uint32(to))

    def build_neibs(self, neibslist, pos, info, hash_, cellStart, cellEnd, n, range_end):
        self.L.orc_build_neibs(C.byref(self.op), _p(neibslist), _p(pos), _p(info), _p(hash_), _p(cellStart), _p(cellEnd),
                               C.c_uint32(n), C.c_uint32(range_end), C.c_float(self.sq), C.byref(self.info_))

    def neibs_info(self):
        return self.info_

    def forces(self, forces, cfl, rbforces, rbtorques, pos, vel, info, hash_, cellStart, neibslist, n, frm, to, cfl_offset, tau=None,
               xsph=None, run_mode=1, step=1):
        assert xsph is None and run_mode == 1, "the test backend runs SIMULATE without XSPH"
        if to > frm:
            forces[frm:to] = 0          # pre_forces clobber (GPUWorker.cc:1949); the oracle does the reference's RMW
        self.L.orc_forces.restype = C.c_uint32
        tau6 = torch.cat(list(tau), dim=1).contiguous() if tau is not None else None      # oracle layout: 6 floats per particle
        return int(self.L.orc_forces(C.byref(self.op), _p(forces), _p(cfl), _p(rbforces), _p(rbtorques), _p(pos), _p(vel),
                                     _p(info), _p(hash_), _p(cellStart), _p(neibslist), _p(tau6), C.c_uint32(n), C.c_uint32(frm),
                                     C.c_uint32(to), C.c_uint32(cfl_offset), C.c_int(self.compute_object_forces)))

    def set_body_motion(self, m, forces_cg):
        p = self.op
        for b in range(len(m["trans"])):
            for a in range(3):
                p.rbtrans[b][a] = float(m["trans"][b][a]); p.rblinearvel[b][a] = float(m["lvel"][b][a])
                p.rbangularvel[b][a] = float(m["avel"][b][a])
                if forces_cg:
                    p.rbcgGridPos[b][a] = int(m["cg_grid"][b][a]); p.rbcgPos[b][a] = float(m["cg_pos"][b][a])
            for a in range(9):
                p.rbsteprot[b][a] = float(m["rot"][b][a])

    def set_body_cg_integration(self, m):
        p = self.op
        for b in range(len(m["trans"])):
            for a in range(3):
                p.rbcgGridPosE[b][a] = int(m["cg_grid"][b][a]); p.rbcgPosE[b][a] = float(m["cg_pos"][b][a])

    def calc_visc(self, tau, pos, vel, info, hash_, cellStart, neibslist, n, range_end, turbvisc=None):
        tau6 = torch.zeros((len(pos), 6), dtype=torch.float32)
        self.L.orc_sps(C.byref(self.op), _p(tau6), None, _p(pos), _p(vel), _p(info), _p(hash_), _p(cellStart), _p(neibslist),
                       C.c_uint32(n), C.c_uint32(range_end))
        for k in range(3):
            tau[k][:range_end] = tau6[:range_end, 2 * k:2 * k + 2]

    def filter(self, filtertype, newvel, pos, oldvel, info, hash_, cellStart, neibslist, n, range_end):
        fn = self.L.orc_shepard if filtertype == 0 else self.L.orc_mls
        fn(C.byref(self.op), _p(newvel), _p(pos), _p(oldvel), _p(info), _p(hash_), _p(cellStart), _p(neibslist), C.c_uint32(range_end))

    def dtreduce(self, cfl, cfl_temp, nblocks, d_dt, combine_min):
        dt = float(self.L.orc_dtreduce(C.byref(self.op), _p(cfl), C.c_uint32(nblocks), C.c_float(self.sspeed_cfl), C.c_float(self.max_kinvisc)))
        d_dt[0] = min(float(d_dt[0]), dt) if combine_min else dt

    def euler(self, npos, nvel, opos, ovel, info, hash_, forces, n, d_dt, dt_scale, step, xsph=None, run_mode=1):
        dt = float(np.float32(d_dt[0].item()) * np.float32(dt_scale))
        self.L.orc_euler(C.byref(self.op), _p(npos), _p(nvel), _p(opos), _p(ovel), _p(info), _p(hash_), _p(forces), None,
                         C.c_uint32(n), C.c_float(dt), C.c_int(step))

    def time_advance(self, d_t, d_dt):
        d_t.add_(d_dt.double())

    # ---- the optional arrays of the re-sort
    def gather_rows(self, sorted_, unsorted, partindex, n):
        sorted_[:n] = unsorted[partindex[:n].long()]

    # ---- SPH_GRENIER
    def init_volume(self, vol, pos, vel, info, n):
        self.L.orc_init_volume(C.byref(self.op), _p(vol), _p(pos), _p(vel), _p(info), C.c_uint32(n))

    def compute_density(self, sigma, vel, pos, info, hash_, vol, cellStart, neibslist, n):
        self.L.orc_density_grenier(C.byref(self.op), _p(sigma), _p(vel), _p(pos), _p(info), _p(hash_), _p(vol), _p(cellStart), _p(neibslist),
                                   C.c_uint32(n), C.c_int(int(self.info_.maxFluidBoundaryNeibs)))

    def forces_grenier(self, forces, cfl, pos, vel, info, hash_, cellStart, neibslist, sigma, n, frm, to, cfl_offset=0):
        if to > frm:
            forces[frm:to] = 0
        self.L.orc_forces_grenier.restype = C.c_uint32
        return int(self.L.orc_forces_grenier(C.byref(self.op), _p(forces), _p(cfl), _p(pos), _p(vel), _p(info), _p(hash_), _p(cellStart),
                                             _p(neibslist), _p(sigma), C.c_uint32(n), C.c_uint32(frm), C.c_uint32(to), C.c_uint32(cfl_offset)))

    def euler_grenier(self, npos, nvel, nvol, opos, ovel, ovol, info, hash_, forces, n, d_dt, dt_scale, step):
        dt = float(np.float32(d_dt[0].item()) * np.float32(dt_scale))
        self.L.orc_euler_grenier(C.byref(self.op), _p(npos), _p(nvel), _p(nvol), _p(opos), _p(ovel), _p(ovol), _p(info), _p(hash_),
                                 _p(forces), None, C.c_uint32(n), C.c_float(dt), C.c_int(step))

    # ---- generalized Newtonian rheologies
    def calc_effvisc(self, effvisc, pos, vel, info, hash_, cellStart, neibslist, n, range_end):
        self.L.orc_effective_visc.restype = C.c_float
        mx = float(self.L.orc_effective_visc(C.byref(self.op), _p(effvisc), None, _p(pos), _p(vel), _p(info), _p(hash_), _p(cellStart),
                                             _p(neibslist), C.c_uint32(n), C.c_uint32(range_end)))
        if mx == mx:
            self.max_kinvisc = mx
        return mx

    def forces_effvisc(self, forces, cfl, pos, vel, info, hash_, cellStart, neibslist, effvisc, n, frm, to, cfl_offset=0):
        if to > frm:
            forces[frm:to] = 0
        self.L.orc_forces_effvisc.restype = C.c_uint32
        return int(self.L.orc_forces_effvisc(C.byref(self.op), _p(forces), _p(cfl), None, None, _p(pos), _p(vel), _p(info), _p(hash_),
                                             _p(cellStart), _p(neibslist), None, C.c_uint32(n), C.c_uint32(frm), C.c_uint32(to),
                                             C.c_uint32(cfl_offset), C.c_int(0), _p(effvisc)))

    # ---- ENABLE_INTERNAL_ENERGY: the energy rate is an output of the oracle's forces passes; run them once more into scratch
    def forces_internal_energy(self, dedt, pos, vel, info, hash_, cellStart, neibslist, n, frm, to):
        scratch = torch.zeros_like(pos)
        cfl = torch.zeros(self.fmax_elements(len(pos)) + 8, dtype=torch.float32)
        dedt[frm:to] = 0
        self.L.orc_set_dedt(_p(dedt))
        self.L.orc_forces(C.byref(self.op), _p(scratch), _p(cfl), None, None, _p(pos), _p(vel), _p(info), _p(hash_), _p(cellStart),
                          _p(neibslist), None, C.c_uint32(n), C.c_uint32(frm), C.c_uint32(to), C.c_uint32(0), C.c_int(0))
        self.L.orc_set_dedt(None)

    def euler_internal_energy(self, new_energy, old_energy, dedt, old_pos, info, n, d_dt, dt_scale):
        dt = float(np.float32(d_dt[0].item()) * np.float32(dt_scale))
        self.L.orc_euler_energy(C.byref(self.op), _p(new_energy), _p(old_energy), _p(dedt), _p(old_pos), _p(info), C.c_uint32(n), C.c_float(dt))

    # ---- SA_BOUNDARY (the interface of gpusph_amd.kernels.HipKernels' sa_* methods, in place on torch tensors)
    def _sa_radii(self):
        s = self._problem.simparams
        f32 = np.float32
        return float(np.power(f32(np.sqrt(f32(s.nlSqInfluenceRadius))) + f32(s.slength) / f32(s.sfactor) / f32(2.0), f32(2.0), dtype=np.float32))

    def build_neibs_sa(self, neibslist, vertpos, pos, info, vertices, boundelements, hash_, cellStart, cellEnd, n, range_end):
        self.L.orc_build_neibs_sa(C.byref(self.op), _p(neibslist), _p(vertpos[0]), _p(vertpos[1]), _p(vertpos[2]), _p(pos), _p(info),
                                  _p(vertices), _p(boundelements), _p(hash_), _p(cellStart), _p(cellEnd), C.c_uint32(n), C.c_uint32(range_end),
                                  C.c_float(self.sq), C.c_float(self._sa_radii()), C.byref(self.info_))

    def sa_compute_vertex_normal(self, boundelements, vertices, info, hash_, cellStart, neibslist, n, range_end):
        self.L.orc_sa_compute_vertex_normal(C.byref(self.op), _p(boundelements), _p(vertices), _p(info), _p(hash_), _p(cellStart), _p(neibslist),
                                            C.c_uint32(range_end))

    def sa_init_gamma(self, new_ggam, old_ggam, pos, boundelements, vertpos, info, hash_, cellStart, neibslist, n, range_end, epsilon=5e-5):
        new_ggam[:n] = old_ggam[:n]
        self.L.orc_sa_init_gamma(C.byref(self.op), _p(new_ggam), _p(pos), _p(boundelements), _p(vertpos[0]), _p(vertpos[1]), _p(vertpos[2]),
                                 _p(info), _p(hash_), _p(cellStart), _p(neibslist), C.c_uint32(range_end), C.c_float(self.sp.deltap),
                                 C.c_float(epsilon))

    def sa_segment_bc(self, vel, ggam, pos, vertices, boundelements, info, hash_, cellStart, neibslist, n, range_end, step, run_mode=1):
        self.L.orc_sa_segment_bc(C.byref(self.op), _p(vel), _p(ggam), _p(pos), _p(vertices), _p(boundelements), _p(info), _p(hash_),
                                 _p(cellStart), _p(neibslist), C.c_uint32(range_end), C.c_int(step), C.c_int(0 if run_mode == 1 else 1))

    def sa_vertex_bc(self, vel, ggam, pos, info, hash_, cellStart, neibslist, n, range_end, step, run_mode=1):
        self.L.orc_sa_vertex_bc(C.byref(self.op), _p(vel), _p(ggam), _p(pos), _p(info), _p(hash_), _p(cellStart), _p(neibslist),
                                C.c_uint32(range_end))

    def forces_sa(self, forces, cfl, pos, vel, info, hash_, cellStart, neibslist, ggam, boundelements, vertpos, n, frm, to, cfl_offset,
                  cfl_gamma=None, run_mode=1):
        assert run_mode == 1
        if to > frm:
            forces[frm:to] = 0
        self.L.orc_forces_sa.restype = C.c_uint32
        return int(self.L.orc_forces_sa(C.byref(self.op), _p(forces), _p(cfl), _p(cfl_gamma), _p(pos), _p(vel), _p(info), _p(hash_),
                                        _p(cellStart), _p(neibslist), _p(ggam), _p(boundelements), _p(vertpos[0]), _p(vertpos[1]),
                                        _p(vertpos[2]), C.c_uint32(n), C.c_uint32(frm), C.c_uint32(to), C.c_uint32(cfl_offset),
                                        C.c_float(self.sp.deltap)))

    def sa_density_sum_io_moving(self, new_vel, new_ggam, forces, old_pos, new_pos, old_vel, old_eulervel, old_ggam, be_old, be_new, vertpos,
                                 info, hash_, cellStart, neibslist, n, range_end, dt):
        new_ggam[:n] = old_ggam[:n]       # the BOUNDARY rows are copied, as the device kernel does (the segment condition re-derives them)
        self.L.orc_sa_density_sum_io_moving(C.byref(self.op), _p(new_vel), _p(new_ggam), _p(forces), _p(old_pos), _p(new_pos), _p(old_vel),
                                            _p(old_eulervel), _p(old_ggam), _p(be_old), _p(be_new), _p(vertpos[0]), _p(vertpos[1]),
                                            _p(vertpos[2]), _p(info), _p(hash_), _p(cellStart), _p(neibslist), C.c_uint32(range_end),
                                            C.c_float(dt))

    def sa_body_pressure_forces(self, forces, rbforces, rbtorques, pos, vel, info, hash_, boundelements, frm, to):
        self.L.orc_sa_body_pressure_forces(C.byref(self.op), _p(forces), _p(rbforces), _p(rbtorques), _p(pos), _p(vel), _p(info), _p(hash_),
                                           _p(boundelements), C.c_uint32(frm), C.c_uint32(to))

    # ---- turbulence<KEPSILON>: the interface of HipKernels' *_keps methods
    def sa_segment_bc_keps(self, vel, ggam, ke, pos, vertices, boundelements, info, hash_, cellStart, neibslist, n, range_end, step):
        self.L.orc_sa_segment_bc_keps(C.byref(self.op), _p(vel), _p(ggam), _p(ke["tke"]), _p(ke["eps"]), _p(ke["eulervel"]), _p(pos),
                                      _p(vertices), _p(boundelements), _p(info), _p(hash_), _p(cellStart), _p(neibslist),
                                      C.c_uint32(range_end), C.c_int(step), C.c_float(self.sp.deltap))

    def sa_vertex_bc_keps(self, vel, ggam, ke, vertices, boundelements, pos, info, hash_, cellStart, neibslist, n, range_end, step):
        self.L.orc_sa_vertex_bc_keps(C.byref(self.op), _p(vel), _p(ggam), _p(ke["tke"]), _p(ke["eps"]), _p(ke["eulervel"]), _p(pos),
                                     _p(vertices), _p(boundelements), _p(info), _p(hash_), _p(cellStart), _p(neibslist), C.c_uint32(range_end))

    def forces_sa_keps(self, forces, cfl, cfl_keps, dkde, pos, vel, info, hash_, cellStart, neibslist, ggam, boundelements, vertpos, ke,
                       n, frm, to, cfl_offset, cfl_gamma=None, epsilon=5e-5):
        if to > frm:
            forces[frm:to] = 0
        strain = torch.zeros((dkde.shape[0], 6), dtype=torch.float32)
        self.L.orc_forces_sa_keps.restype = C.c_uint32
        return int(self.L.orc_forces_sa_keps(C.byref(self.op), _p(forces), _p(cfl), _p(cfl_gamma), _p(cfl_keps), _p(dkde), _p(strain),
                                             _p(pos), _p(vel), _p(info), _p(hash_), _p(cellStart), _p(neibslist), _p(ggam), _p(boundelements),
                                             _p(vertpos[0]), _p(vertpos[1]), _p(vertpos[2]), _p(ke["tke"]), _p(ke["eps"]), _p(ke["turbvisc"]),
                                             _p(ke["eulervel"]), C.c_uint32(n), C.c_uint32(frm), C.c_uint32(to), C.c_uint32(cfl_offset),
                                             C.c_float(self.sp.deltap), C.c_float(epsilon)))

    def euler_keps(self, new, old, dkde, forces, old_pos, info, n, d_dt, dt_scale):
        dt = float(np.float32(d_dt[0].item()) * np.float32(dt_scale))
        self.L.orc_euler_keps(C.byref(self.op), _p(new["tke"]), _p(new["eps"]), _p(new["turbvisc"]), _p(new["eulervel"]), _p(old["tke"]),
                              _p(old["eps"]), _p(old["eulervel"]), _p(dkde), _p(forces), _p(old_pos), _p(info), C.c_uint32(n), C.c_float(dt))

    def dtreduce_keps(self, cfl_keps, nblocks, d_dt):
        f = np.float32
        h = f(self.op.slength)
        mx = f(cfl_keps[:nblocks].max().item()) if nblocks else f(0)
        dt_visc = f(f(h * h) / f(f(self.max_kinvisc) + mx)) * f(0.125)
        d_dt[0] = min(float(d_dt[0]), float(dt_visc))

    def dtreduce_gamma(self, cfl_gamma, n, nblocks, d_dt):
        base = ((n + 3) // 4) * 4
        mx = float(cfl_gamma[base:base + nblocks].max()) if nblocks else 0.0
        self.L.orc_sa_gamma_dt.restype = C.c_float
        dt = float(d_dt[0])
        d_dt[0] = min(dt, float(self.L.orc_sa_gamma_dt(C.c_float(dt), C.c_float(mx))))

    def sa_density_sum(self, new_vel, new_ggam, forces, old_pos, new_pos, old_vel, old_ggam, boundelements, vertpos, info, hash_, cellStart,
                       neibslist, n, range_end, dt=0.0, step=1):
        new_ggam[:n] = old_ggam[:n]       # rows of the other particle types are copied (copyTypeDataDevice)
        self.L.orc_sa_density_sum(C.byref(self.op), _p(new_vel), _p(new_ggam), _p(forces), _p(old_pos), _p(new_pos), _p(old_vel), _p(old_ggam),
                                  _p(boundelements), _p(vertpos[0]), _p(vertpos[1]), _p(vertpos[2]), _p(info), _p(hash_), _p(cellStart),
                                  _p(neibslist), C.c_uint32(range_end))

    def sa_density_diffusion(self, forces, pos, vel, ggam, info, hash_, cellStart, neibslist, n, range_end, dt):
        self.L.orc_sa_density_diffusion(C.byref(self.op), _p(forces), _p(pos), _p(vel), _p(ggam), _p(info), _p(hash_), _p(cellStart),
                                        _p(neibslist), C.c_uint32(range_end), C.c_float(dt))
        fluid = (info[:range_end, 0].to(torch.int32) & 7) == 0
        rows = torch.nonzero(fluid).flatten()
        vel[rows, 3] = vel[rows, 3] + forces[rows, 3] * np.float32(dt)

    def sa_integrate_gamma(self, new_ggam, old_ggam, new_pos, boundelements, vertpos, info, hash_, cellStart, neibslist, n, range_end,
                           epsilon=5e-5):
        new_ggam[:n] = old_ggam[:n]
        from gpusph_amd import defs as D
        moving = bool(int(self.op.simflags) & D.ENABLE_MOVING_BODIES)
        for cptype in ((0, 2) if moving else (0,)):      # with moving bodies the vertex rows are integrated too
            self.L.orc_sa_integrate_gamma_quadrature(C.byref(self.op), _p(new_ggam), _p(old_ggam), _p(new_pos), _p(boundelements), _p(vertpos[0]),
                                                     _p(vertpos[1]), _p(vertpos[2]), _p(info), _p(hash_), _p(cellStart), _p(neibslist),
                                                     C.c_uint32(range_end), C.c_int(cptype), C.c_float(epsilon))

    # ---- SA_BOUNDARY with moving bodies
    def sa_update_normals(self, new_be, old_be, info, n, range_end):
        self.L.orc_sa_update_normals(C.byref(self.op), _p(new_be), _p(old_be), _p(info), C.c_uint32(range_end))

    def sa_density_sum_moving(self, new_vel, new_ggam, forces, old_pos, new_pos, old_vel, old_ggam, old_be, new_be, vertpos, info, hash_,
                              cellStart, neibslist, n, range_end):
        self.L.orc_sa_density_sum_moving(C.byref(self.op), _p(new_vel), _p(new_ggam), _p(forces), _p(old_pos), _p(new_pos), _p(old_vel),
                                         _p(old_ggam), _p(old_be), _p(new_be), _p(vertpos[0]), _p(vertpos[1]), _p(vertpos[2]), _p(info),
                                         _p(hash_), _p(cellStart), _p(neibslist), C.c_uint32(range_end))

    # ---- SA open boundaries (ENABLE_INLET_OUTLET): the interface of HipKernels' *_io methods over the oracle's restatements
    def _vp(self, vertpos):
        return _p(vertpos[0]), _p(vertpos[1]), _p(vertpos[2])

    def sa_identify_corner_vertices(self, pos, info, hash_, vertices, cellStart, neibslist, n, range_end):
        self.L.orc_sa_identify_corner_vertices(C.byref(self.op), _p(pos), _p(info), _p(hash_), _p(vertices), _p(cellStart), _p(neibslist),
                                               C.c_uint32(range_end))

    def sa_init_io_mass_vertex_count(self, forces, pos, vertices, hash_, info, cellStart, neibslist, n, range_end):
        forces[:n] = 0
        self.L.orc_sa_init_io_mass_vertex_count(C.byref(self.op), _p(vertices), _p(hash_), _p(info), _p(cellStart), _p(neibslist),
                                                _p(forces), C.c_uint32(range_end))

    def sa_init_io_mass(self, new_pos, pos, forces, vertices, hash_, info, cellStart, neibslist, n, range_end):
        self.L.orc_sa_init_io_mass(C.byref(self.op), _p(pos), _p(forces), _p(vertices), _p(hash_), _p(info), _p(cellStart), _p(neibslist),
                                   _p(new_pos), C.c_uint32(range_end), C.c_float(self.sp.deltap))

    def sa_segment_bc_io(self, vel, ggam, eulervel, pos, vertices, boundelements, info, hash_, cellStart, neibslist, n, range_end, step):
        self.L.orc_sa_segment_bc_io(C.byref(self.op), _p(vel), _p(ggam), _p(eulervel), _p(pos), _p(vertices), _p(boundelements), _p(info),
                                    _p(hash_), _p(cellStart), _p(neibslist), C.c_uint32(range_end), C.c_int(step))

    def sa_vertex_bc_io(self, vel, old_pos, new_pos, ggam, eulervel, forces, vertices, boundelements, vertpos, info, hash_, next_ids,
                        count, cellStart, neibslist, n, range_end, max_particles, dt, step, num_open_vertices):
        self.L.orc_sa_vertex_bc_io(C.byref(self.op), _p(vel), _p(old_pos), _p(new_pos), _p(ggam), _p(eulervel), _p(forces), _p(vertices),
                                   _p(boundelements), *self._vp(vertpos), _p(info), _p(hash_), _p(next_ids), _p(count), _p(cellStart),
                                   _p(neibslist), C.c_uint32(range_end), C.c_uint32(max_particles), C.c_float(self.sp.deltap),
                                   C.c_float(dt), C.c_int(step), C.c_uint32(num_open_vertices))

    def sa_find_outgoing_segment(self, pos, vel, vertices, ggam, vertpos, boundelements, info, hash_, cellStart, neibslist, n, range_end):
        self.L.orc_find_outgoing_segment(C.byref(self.op), _p(pos), _p(vel), _p(vertices), _p(ggam), *self._vp(vertpos), _p(boundelements),
                                         _p(info), _p(hash_), _p(cellStart), _p(neibslist), C.c_uint32(range_end),
                                         C.c_float(self.sp.influenceradius))

    def sa_disable_outgoing_parts(self, pos, vertices, info, n):
        self.L.orc_disable_outgoing_parts(_p(pos), _p(vertices), _p(info), C.c_uint32(n))

    def sa_density_sum_io(self, new_vel, new_ggam, forces, old_pos, new_pos, old_vel, old_eulervel, old_ggam, boundelements, vertpos, info,
                          hash_, cellStart, neibslist, n, range_end, dt):
        new_ggam[:n] = old_ggam[:n]       # rows of the other particle types are copied
        self.L.orc_sa_density_sum_io(C.byref(self.op), _p(new_vel), _p(new_ggam), _p(forces), _p(old_pos), _p(new_pos), _p(old_vel),
                                     _p(old_eulervel), _p(old_ggam), _p(boundelements), *self._vp(vertpos), _p(info), _p(hash_),
                                     _p(cellStart), _p(neibslist), C.c_uint32(range_end), C.c_float(dt))

    def sa_density_diffusion_io(self, forces, pos, vel, ggam, boundelements, vertpos, info, hash_, cellStart, neibslist, n, range_end, dt):
        self.L.orc_sa_density_diffusion_io(C.byref(self.op), _p(forces), _p(pos), _p(vel), _p(ggam), _p(info), _p(hash_), _p(cellStart),
                                           _p(neibslist), _p(boundelements), *self._vp(vertpos), C.c_uint32(range_end), C.c_float(dt),
                                           C.c_float(self.sp.deltap))
        fluid = (info[:range_end, 0].to(torch.int32) & 7) == 0
        rows = torch.nonzero(fluid).flatten()
        vel[rows, 3] = vel[rows, 3] + forces[rows, 3] * np.float32(dt)

    def forces_sa_io(self, forces, cfl, pos, vel, eulervel, info, hash_, cellStart, neibslist, ggam, boundelements, vertpos, n, frm, to,
                     cfl_offset, cfl_gamma=None):
        if to > frm:
            forces[frm:to] = 0
        self.L.orc_forces_sa_io.restype = C.c_uint32
        return int(self.L.orc_forces_sa_io(C.byref(self.op), _p(forces), _p(cfl), _p(cfl_gamma), _p(pos), _p(vel), _p(eulervel), _p(info),
                                           _p(hash_), _p(cellStart), _p(neibslist), _p(ggam), _p(boundelements), *self._vp(vertpos),
                                           C.c_uint32(n), C.c_uint32(frm), C.c_uint32(to), C.c_uint32(cfl_offset),
                                           C.c_float(self.sp.deltap)))

    def sa_io_water_depth(self, depth, pos, info, hash_, cellStart, neibslist, n, frm, to):
        self.L.orc_sa_io_water_depth(C.byref(self.op), _p(depth), _p(pos), _p(info), _p(hash_), _p(cellStart), _p(neibslist),
                                     C.c_uint32(frm), C.c_uint32(to))

    def flux_computation(self, flux, info, eulervel, boundelements, n, num_open_boundaries):
        self.L.orc_flux_computation(_p(flux), _p(info), _p(eulervel), _p(boundelements), C.c_uint32(n), C.c_uint32(num_open_boundaries))
