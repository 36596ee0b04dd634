"""HipKernels -- the product's kernel backend: torch tensors in, C-ABI calls (include/sphx.h) out.

The multi-GPU driver (multigpu.py) talks to the device only through this small interface so that its
host logic -- slab partition, halo bookkeeping, exchange order -- can be exercised by the CPU test
suite with a test-only backend (tests/oracle_kernels.py).  There is no CPU implementation in the
package: constructing HipKernels without a HIP device raises.
"""
import ctypes as C
import numpy as np
import torch

from . import capi, defs as D


class HipKernels:
    def __init__(self, problem, alloc, device):
        if not torch.cuda.is_available():
            raise capi.SphxError("HipKernels needs a HIP device (there is no CPU fallback)")
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self.lib = capi.load()
        self.ctx = capi.Context(self.device.index or 0)
        self.params = problem.sphx_params(alloc)
        self.ctx.set_constants(self.params)
        self.ctx.reserve(alloc)
        if getattr(problem, "planes", None):
            nrm, gpos, lpos = problem.plane_tables()
            capi.check(self.lib.sphx_set_planes(self.ctx.handle, nrm.ctypes.data, gpos.ctypes.data, lpos.ctypes.data, len(nrm)))
        if getattr(problem, "dem", None) is not None:      # GPUWorker::allocateDeviceBuffers -> setDEM (GPUWorker.cc:1070-1076)
            dem = np.ascontiguousarray(problem.dem, dtype=np.float32)
            capi.check(self.lib.sphx_set_dem(self.ctx.handle, dem.ctypes.data, dem.shape[1], dem.shape[0]))
        self.ncells = problem.grid_cells
        sp = problem.simparams
        self.problem_simparams = sp
        self.compute_object_forces = 1 if sp.numforcesbodies > 0 else 0
        self.sq_nl_radius = float(np.float32(sp.nlSqInfluenceRadius))
        pp = problem.physparams
        self.sspeed_cfl = float(np.float32(np.float64(np.float32(max(pp.sscoeff))) * 1.1))  # GPUWorker.cc:3010-3011
        self.max_kinvisc = float(np.float32(max(pp.kinematicvisc))) if sp.rheologytype == D.NEWTONIAN else 0.0   # :3003-3006
        if getattr(problem, "num_obstacle", 0):
            gp = np.ascontiguousarray(problem.rb_cg_gridpos, dtype=np.int32)
            lp = np.ascontiguousarray(problem.rb_cg_pos, dtype=np.float32)
            fi = np.ascontiguousarray(problem.rb_firstindex, dtype=np.int32)
            nb = len(fi)
            capi.check(self.lib.sphx_set_rb_cg(self.ctx.handle, gp.ctypes.data, lp.ctypes.data, nb))
            capi.check(self.lib.sphx_set_rb_start(self.ctx.handle, fi.ctypes.data, nb))
            ident = np.tile(np.eye(3, dtype=np.float32).ravel(), nb)
            z3 = np.zeros(3 * nb, dtype=np.float32)
            capi.check(self.lib.sphx_set_rb_motion(self.ctx.handle, z3.ctypes.data, ident.ctypes.data,
                                                   z3.ctypes.data, z3.ctypes.data, nb))

    # ---- MOVE_BODIES uploads (see gpusph_amd/bodies.py); `m` keeps its host arrays alive until the calls return
    def set_body_motion(self, m, forces_cg):
        nb = len(m["trans"])
        capi.check(self.lib.sphx_set_rb_motion(self.ctx.handle, m["trans"].ctypes.data, m["rot"].ctypes.data,
                                               m["lvel"].ctypes.data, m["avel"].ctypes.data, nb))
        if forces_cg:
            capi.check(self.lib.sphx_set_rb_cg_forces(self.ctx.handle, m["cg_grid"].ctypes.data, m["cg_pos"].ctypes.data, nb))

    def set_body_cg_integration(self, m):
        capi.check(self.lib.sphx_set_rb_cg_integration(self.ctx.handle, m["cg_grid"].ctypes.data, m["cg_pos"].ctypes.data, len(m["trans"])))

    # ---- helpers
    def _s(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def fmax_elements(self, n):
        return int(self.lib.sphx_forces_fmax_elements(int(n)))

    def fmax_temp_elements(self, n):
        return int(self.lib.sphx_forces_fmax_temp_elements(int(n)))

    def memset(self, t, byte):
        capi.check(self.lib.sphx_memset_async(t.data_ptr(), byte, t.numel() * t.element_size(), self._s()))

    # ---- AbstractNeibsEngine
    def calc_hash(self, pos, hash_, partindex, info, devmap, n):
        p = capi.ptr
        capi.check(self.lib.sphx_calc_hash(self.ctx.handle, p(pos), p(hash_), p(partindex), p(info), p(devmap), n, self._s()))

    def fix_hash(self, hash_, partindex, info, devmap, n):
        p = capi.ptr
        capi.check(self.lib.sphx_fix_hash(self.ctx.handle, p(hash_), p(partindex), p(info), p(devmap), n, self._s()))

    def sort(self, hash_, info, partindex, n):
        p = capi.ptr
        capi.check(self.lib.sphx_sort(self.ctx.handle, p(hash_), p(info), p(partindex), n, self._s()))

    def reorder(self, segment_start, cellStart, cellEnd, spos, svel, upos, uvel, info, hash_, partindex, n, new_num):
        p = capi.ptr
        capi.check(self.lib.sphx_reorder(self.ctx.handle, p(segment_start), p(cellStart), p(cellEnd), p(spos), p(svel),
                                         p(upos), p(uvel), p(info), p(hash_), p(partindex), n, p(new_num), self._s()))

    def find_cell_start(self, cellStart, cellEnd, hash_, frm, to):
        p = capi.ptr
        capi.check(self.lib.sphx_find_cell_start(self.ctx.handle, p(cellStart), p(cellEnd), p(hash_), frm, to, self._s()))

    def build_neibs(self, neibslist, pos, info, hash_, cellStart, cellEnd, n, range_end):
        p = capi.ptr
        capi.check(self.lib.sphx_neibs_resetinfo(self.ctx.handle, self._s()))
        capi.check(self.lib.sphx_build_neibs(self.ctx.handle, p(neibslist), p(pos), p(info), p(hash_), p(cellStart),
                                             p(cellEnd), n, range_end, self.ncells, self.sq_nl_radius, self.sq_nl_radius,
                                             self._s()))

    def gather_rows(self, sorted_, unsorted, partindex, n):
        """the optional arrays of reorderDataAndFindCellStart: sorted[i] = unsorted[partindex[i]]"""
        row = unsorted.element_size() * (unsorted.numel() // unsorted.shape[0])
        capi.check(self.lib.sphx_gather_rows(self.ctx.handle, capi.ptr(sorted_), capi.ptr(unsorted), row, capi.ptr(partindex), n, self._s()))

    def build_neibs_sa(self, neibslist, vertpos, pos, info, vertices, boundelements, hash_, cellStart, cellEnd, n, range_end):
        p = capi.ptr
        sp = self.problem_simparams
        # GPUWorker.cc:1890: (sqrt(nlSqInfluenceRadius) + slength/sfactor/2)^2 in float
        f32 = np.float32
        bound_sq = float(np.power(f32(np.sqrt(f32(sp.nlSqInfluenceRadius))) + f32(sp.slength) / f32(sp.sfactor) / f32(2.0), f32(2.0), dtype=np.float32))
        capi.check(self.lib.sphx_neibs_resetinfo(self.ctx.handle, self._s()))
        capi.check(self.lib.sphx_build_neibs_sa(self.ctx.handle, p(neibslist), p(vertpos[0]), p(vertpos[1]), p(vertpos[2]), p(pos), p(info),
                                                p(vertices), p(boundelements), p(hash_), p(cellStart), p(cellEnd), n, range_end,
                                                self.ncells, self.sq_nl_radius, bound_sq, self._s()))

    # ---- AbstractBoundaryConditionsEngine (SA_BOUNDARY, solid walls)
    def sa_compute_vertex_normal(self, boundelements, vertices, info, hash_, cellStart, neibslist, n, range_end):
        p = capi.ptr
        capi.check(self.lib.sphx_sa_compute_vertex_normal(self.ctx.handle, p(boundelements), p(vertices), p(info), p(hash_), p(cellStart),
                                                          p(neibslist), n, range_end, self._s()))

    def sa_init_gamma(self, new_ggam, old_ggam, pos, boundelements, vertpos, info, hash_, cellStart, neibslist, n, range_end, epsilon=5e-5):
        p = capi.ptr
        P = self.params
        capi.check(self.lib.sphx_sa_init_gamma(self.ctx.handle, p(new_ggam), p(old_ggam), p(pos), p(boundelements), p(vertpos[0]),
                                               p(vertpos[1]), p(vertpos[2]), p(info), p(hash_), p(cellStart), p(neibslist),
                                               P.slength, P.influenceradius, P.deltap, float(np.float32(epsilon)), n, range_end, self._s()))

    def forces_sa(self, forces, cfl, pos, vel, info, hash_, cellStart, neibslist, ggam, boundelements, vertpos, n, frm, to, cfl_offset,
                  cfl_gamma=None, run_mode=D.SIMULATE):
        p = capi.ptr
        P = self.params
        nb = C.c_uint32(0)
        capi.check(self.lib.sphx_forces_basicstep_sa(self.ctx.handle, p(forces), p(cfl), p(cfl_gamma), p(pos), p(vel), p(info), p(hash_), p(cellStart),
                                                     p(neibslist), p(ggam), p(boundelements), p(vertpos[0]), p(vertpos[1]), p(vertpos[2]),
                                                     n, frm, to, P.deltap, P.slength, P.dtadaptfactor, P.influenceradius, cfl_offset,
                                                     run_mode, 1, 0.0, C.byref(nb), self._s()))
        return nb.value

    def sa_density_sum_io_moving(self, new_vel, new_ggam, forces, old_pos, new_pos, old_vel, old_eulervel, old_ggam, be_old, be_new,
                                 vertpos, info, hash_, cellStart, neibslist, n, range_end, dt):
        """DENSITY_SUM with ENABLE_INLET_OUTLET | ENABLE_MOVING_BODIES (CompleteSaExample.cu's option set)"""
        p = capi.ptr
        capi.check(self.lib.sphx_sa_density_sum_io_moving(self.ctx.handle, p(new_vel), p(new_ggam), p(forces), p(old_pos), p(new_pos),
                                                          p(old_vel), p(old_eulervel), p(old_ggam), p(be_old), p(be_new), p(vertpos[0]),
                                                          p(vertpos[1]), p(vertpos[2]), p(info), p(hash_), p(cellStart), p(neibslist),
                                                          n, range_end, float(np.float32(dt)), self._s()))

    def sa_body_pressure_forces(self, forces, rbforces, rbtorques, pos, vel, info, hash_, boundelements, frm, to):
        """F = -P A n on the FG_COMPUTE_FORCE boundary elements of [frm, to) -> BUFFER_RB_FORCES / RB_TORQUES rows (+ their forces row)"""
        p = capi.ptr
        capi.check(self.lib.sphx_sa_body_pressure_forces(self.ctx.handle, p(forces), p(rbforces), p(rbtorques), p(pos), p(vel), p(info),
                                                         p(hash_), p(boundelements), frm, to, self._s()))

    # ---- turbulence<KEPSILON> (SA_BOUNDARY, solid walls): ke = dict(tke, eps, turbvisc, eulervel) of the state
    def forces_sa_keps(self, forces, cfl, cfl_keps, dkde, pos, vel, info, hash_, cellStart, neibslist, ggam, boundelements, vertpos, ke,
                       n, frm, to, cfl_offset, cfl_gamma=None, epsilon=5e-5):
        p = capi.ptr
        P = self.params
        nb = C.c_uint32(0)
        capi.check(self.lib.sphx_forces_basicstep_sa_keps(
            self.ctx.handle, p(forces), p(cfl), p(cfl_gamma), p(cfl_keps), p(dkde), p(pos), p(vel), p(info), p(hash_), p(cellStart),
            p(neibslist), p(ggam), p(boundelements), p(vertpos[0]), p(vertpos[1]), p(vertpos[2]),
            p(ke["tke"]), p(ke["eps"]), p(ke["turbvisc"]), p(ke["eulervel"]),
            n, frm, to, P.deltap, P.slength, P.dtadaptfactor, P.influenceradius, float(epsilon), cfl_offset,
            D.SIMULATE, 1, 0.0, C.byref(nb), self._s()))
        return nb.value

    def sa_segment_bc_keps(self, vel, ggam, ke, pos, vertices, boundelements, info, hash_, cellStart, neibslist, n, range_end, step):
        p = capi.ptr
        P = self.params
        capi.check(self.lib.sphx_sa_segment_bc_keps(self.ctx.handle, p(vel), p(ggam), p(ke["tke"]), p(ke["eps"]), p(ke["eulervel"]), p(pos),
                                                    p(vertices), p(boundelements), p(info), p(hash_), p(cellStart), p(neibslist), n, range_end,
                                                    P.deltap, P.slength, P.influenceradius, int(step), D.SIMULATE, self._s()))

    def sa_vertex_bc_keps(self, vel, ggam, ke, vertices, boundelements, pos, info, hash_, cellStart, neibslist, n, range_end, step):
        p = capi.ptr
        P = self.params
        capi.check(self.lib.sphx_sa_vertex_bc_keps(self.ctx.handle, p(vel), p(ggam), p(ke["tke"]), p(ke["eps"]), p(ke["eulervel"]),
                                                   p(vertices), p(boundelements), p(pos), p(info), p(hash_), p(cellStart), p(neibslist),
                                                   n, range_end, P.deltap, P.slength, P.influenceradius, int(step), D.SIMULATE, self._s()))

    def euler_keps(self, new, old, dkde, forces, old_pos, info, n, d_dt, dt_scale):
        p = capi.ptr
        capi.check(self.lib.sphx_euler_keps(self.ctx.handle, p(new["tke"]), p(new["eps"]), p(new["turbvisc"]), p(new["eulervel"]),
                                            p(old["tke"]), p(old["eps"]), p(old["eulervel"]), p(dkde), p(forces), p(old_pos), p(info),
                                            n, n, 0.0, p(d_dt), dt_scale, self._s()))

    def dtreduce_keps(self, cfl_keps, nblocks, d_dt):
        capi.check(self.lib.sphx_forces_dtreduce_keps_device(self.ctx.handle, capi.ptr(cfl_keps), nblocks, self.params.slength,
                                                             self.max_kinvisc, capi.ptr(d_dt), self._s()))

    def dtreduce_gamma(self, cfl_gamma, n, nblocks, d_dt):
        capi.check(self.lib.sphx_forces_dtreduce_gamma_device(self.ctx.handle, capi.ptr(cfl_gamma), n, nblocks, capi.ptr(d_dt), self._s()))

    def sa_density_sum(self, new_vel, new_ggam, forces, old_pos, new_pos, old_vel, old_ggam, boundelements, vertpos, info, hash_, cellStart,
                       neibslist, n, range_end, dt=0.0, step=1):
        p = capi.ptr
        P = self.params
        capi.check(self.lib.sphx_sa_density_sum(self.ctx.handle, p(new_vel), p(new_ggam), p(forces), p(old_pos), p(new_pos), p(old_vel),
                                                p(old_ggam), p(boundelements), p(vertpos[0]), p(vertpos[1]), p(vertpos[2]), p(info), p(hash_),
                                                p(cellStart), p(neibslist), n, range_end, float(dt), int(step), 0.0, 5e-5, P.deltap, P.slength,
                                                P.influenceradius, self._s()))

    # ---- SA_BOUNDARY with moving bodies (include/sphx.h)
    def sa_update_normals(self, new_be, old_be, info, n, range_end):
        p = capi.ptr
        capi.check(self.lib.sphx_sa_update_normals(self.ctx.handle, p(new_be), p(old_be), p(info), n, range_end, self._s()))

    def sa_density_sum_moving(self, new_vel, new_ggam, forces, old_pos, new_pos, old_vel, old_ggam, old_be, new_be, vertpos, info, hash_,
                              cellStart, neibslist, n, range_end):
        p = capi.ptr
        capi.check(self.lib.sphx_sa_density_sum_moving(self.ctx.handle, p(new_vel), p(new_ggam), p(forces), p(old_pos), p(new_pos), p(old_vel),
                                                       p(old_ggam), p(old_be), p(new_be), p(vertpos[0]), p(vertpos[1]), p(vertpos[2]), p(info),
                                                       p(hash_), p(cellStart), p(neibslist), n, range_end, self._s()))

    def sa_density_diffusion(self, forces, pos, vel, ggam, info, hash_, cellStart, neibslist, n, range_end, dt):
        """compute_density_diffusion (forces engine) + apply_density_diffusion (integration engine)"""
        p = capi.ptr
        P = self.params
        capi.check(self.lib.sphx_sa_compute_density_diffusion(self.ctx.handle, p(forces), p(pos), p(vel), p(ggam), p(info), p(hash_), p(cellStart),
                                                              p(neibslist), n, range_end, P.deltap, P.slength, P.influenceradius, float(dt), self._s()))
        capi.check(self.lib.sphx_apply_density_diffusion(self.ctx.handle, p(vel), p(forces), p(info), n, range_end, float(dt), self._s()))

    def sa_integrate_gamma(self, new_ggam, old_ggam, new_pos, boundelements, vertpos, info, hash_, cellStart, neibslist, n, range_end,
                           epsilon=5e-5):
        p = capi.ptr
        P = self.params
        capi.check(self.lib.sphx_sa_integrate_gamma(self.ctx.handle, p(new_ggam), p(old_ggam), p(new_pos), p(boundelements), p(vertpos[0]),
                                                    p(vertpos[1]), p(vertpos[2]), p(info), p(hash_), p(cellStart), p(neibslist), n, range_end,
                                                    0.0, 1, 0.0, float(np.float32(epsilon)), P.slength, P.influenceradius, D.SIMULATE, self._s()))

    def sa_segment_bc(self, vel, ggam, pos, vertices, boundelements, info, hash_, cellStart, neibslist, n, range_end, step,
                      run_mode=D.SIMULATE):
        p = capi.ptr
        P = self.params
        capi.check(self.lib.sphx_sa_segment_bc(self.ctx.handle, p(vel), p(ggam), p(pos), p(vertices), p(boundelements), p(info), p(hash_),
                                               p(cellStart), p(neibslist), n, range_end, P.deltap, P.slength, P.influenceradius,
                                               int(step), int(run_mode), self._s()))

    def sa_vertex_bc(self, vel, ggam, pos, info, hash_, cellStart, neibslist, n, range_end, step, run_mode=D.SIMULATE):
        p = capi.ptr
        P = self.params
        capi.check(self.lib.sphx_sa_vertex_bc(self.ctx.handle, p(vel), p(ggam), p(pos), p(info), p(hash_), p(cellStart), p(neibslist),
                                              n, range_end, P.deltap, P.slength, P.influenceradius, int(step), int(run_mode), self._s()))

    def reserve_comm_cus(self, cus):
        """leave CUs out of the persistent forces grid for the communication kernels of the halo exchange"""
        capi.check(self.lib.sphx_forces_reserve_cus(self.ctx.handle, int(cus)))

    def neibs_info(self):
        info = capi.NeibsInfo()
        capi.check(self.lib.sphx_neibs_getinfo(self.ctx.handle, C.byref(info), self._s()))
        return info

    # ---- AbstractForcesEngine
    def forces(self, forces, cfl, rbforces, rbtorques, pos, vel, info, hash_, cellStart, neibslist, n, frm, to, cfl_offset, tau=None,
               xsph=None, run_mode=D.SIMULATE, step=1):
        p = capi.ptr
        t0, t1, t2 = (p(t) for t in tau) if tau is not None else (None, None, None)
        nb = C.c_uint32(0)
        P = self.params
        capi.check(self.lib.sphx_forces_basicstep(self.ctx.handle, p(forces), p(cfl), p(rbforces), p(rbtorques), p(pos), p(vel),
                                                  p(info), p(hash_), p(cellStart), p(neibslist), t0, t1, t2, p(xsph),
                                                  n, frm, to, P.deltap, P.slength, P.dtadaptfactor, P.influenceradius,
                                                  cfl_offset, run_mode, step, 0.0, self.compute_object_forces,
                                                  C.byref(nb), self._s()))
        return nb.value

    def dtreduce(self, cfl, cfl_temp, nblocks, d_dt, combine_min):
        p = capi.ptr
        P = self.params
        # stricter viscous limit of the MONAGHAN and ESPANOL_REVENGA models (GPUWorker.cc:2013-2022)
        mk = self.max_kinvisc*(float(P.monaghan_visc_coeff) if P.viscmodel == D.MONAGHAN else 5.0 if P.viscmodel == D.ESPANOL_REVENGA else 1.0)
        capi.check(self.lib.sphx_forces_dtreduce_device(self.ctx.handle, P.slength, P.dtadaptfactor, self.sspeed_cfl, mk,
                                                        p(cfl), p(cfl_temp), nblocks, p(d_dt), combine_min, self._s()))

    # ---- AbstractViscEngine / AbstractFilterEngine
    def calc_visc(self, tau, pos, vel, info, hash_, cellStart, neibslist, n, range_end, turbvisc=None):
        p = capi.ptr
        P = self.params
        capi.check(self.lib.sphx_calc_visc(self.ctx.handle, p(tau[0]), p(tau[1]), p(tau[2]), p(turbvisc), p(pos), p(vel), p(info), p(hash_),
                                           p(cellStart), p(neibslist), n, range_end, P.deltap, P.slength, P.influenceradius, self._s()))

    def filter(self, filtertype, newvel, pos, oldvel, info, hash_, cellStart, neibslist, n, range_end):
        p = capi.ptr
        P = self.params
        capi.check(self.lib.sphx_filter_process(self.ctx.handle, filtertype, p(newvel), p(pos), p(oldvel), p(info), p(hash_),
                                                p(cellStart), p(neibslist), n, range_end, P.slength, P.influenceradius, self._s()))

    # ---- AbstractIntegrationEngine
    def euler(self, npos, nvel, opos, ovel, info, hash_, forces, n, d_dt, dt_scale, step, xsph=None, run_mode=D.SIMULATE):
        p = capi.ptr
        P = self.params
        capi.check(self.lib.sphx_euler_basicstep(self.ctx.handle, p(npos), p(nvel), p(opos), p(ovel), p(info), p(hash_),
                                                 p(forces), p(xsph), n, n, 0.0, p(d_dt), dt_scale, step, 0.0,
                                                 P.slength, P.influenceradius, run_mode, self._s()))

    def eos_rows_follow_euler(self, on):
        """the forces engine's EOS rows are written by the Euler step (include/sphx.h)"""
        capi.check(self.lib.sphx_eos_rows_follow_euler(self.ctx.handle, 1 if on else 0))

    def eos_rows_current(self, vel, n):
        """the caller's statement that `vel` is unchanged since the rows were made for it; holds for the next forces call"""
        capi.check(self.lib.sphx_eos_rows_current(self.ctx.handle, capi.ptr(vel), n))

    def time_advance(self, d_t, d_dt):
        capi.check(self.lib.sphx_time_advance(self.ctx.handle, capi.ptr(d_t), capi.ptr(d_dt), self._s()))

    # ---- ENABLE_INTERNAL_ENERGY (energy.hip)
    def forces_internal_energy(self, dedt, pos, vel, info, hash_, cellStart, neibslist, n, frm, to):
        p = capi.ptr
        capi.check(self.lib.sphx_forces_internal_energy(self.ctx.handle, p(dedt), p(pos), p(vel), p(info), p(hash_), p(cellStart),
                                                        p(neibslist), n, frm, to, self._s()))

    def euler_internal_energy(self, new_energy, old_energy, dedt, old_pos, info, n, d_dt, dt_scale):
        p = capi.ptr
        capi.check(self.lib.sphx_euler_internal_energy(self.ctx.handle, p(new_energy), p(old_energy), p(dedt), p(old_pos), p(info), n, n,
                                                       0.0, p(d_dt), dt_scale, self._s()))

    # ---- generalized Newtonian rheologies (rheology.hip)
    def calc_effvisc(self, effvisc, pos, vel, info, hash_, cellStart, neibslist, n, range_end):
        """CALC_VISC: BUFFER_EFFVISC written; returns the largest kinematic viscosity (one host synchronisation) and makes it
        the viscous limit of the following dt reductions, as GPUWorker does with calc_visc's return value"""
        p = capi.ptr
        P = self.params
        mx = C.c_float(0.0)
        capi.check(self.lib.sphx_calc_effvisc(self.ctx.handle, p(effvisc), C.byref(mx), p(pos), p(vel), p(info), p(hash_), p(cellStart),
                                              p(neibslist), n, range_end, P.deltap, P.slength, P.influenceradius, self._s()))
        if mx.value == mx.value:
            self.max_kinvisc = float(mx.value)
        return float(mx.value)

    def forces_effvisc(self, forces, cfl, pos, vel, info, hash_, cellStart, neibslist, effvisc, n, frm, to, cfl_offset=0):
        p = capi.ptr
        P = self.params
        nb = C.c_uint32(0)
        capi.check(self.lib.sphx_forces_basicstep_effvisc(self.ctx.handle, p(forces), p(cfl), p(pos), p(vel), p(info), p(hash_),
                                                          p(cellStart), p(neibslist), p(effvisc), n, frm, to, P.deltap, P.slength,
                                                          P.dtadaptfactor, P.influenceradius, cfl_offset, D.SIMULATE, 1, 0.0,
                                                          C.byref(nb), self._s()))
        return int(nb.value)

    # ---- SPH_GRENIER (grenier.hip)
    def init_volume(self, vol, pos, vel, info, n):
        p = capi.ptr
        capi.check(self.lib.sphx_init_volume(self.ctx.handle, p(vol), p(pos), p(vel), p(info), n, self._s()))

    def compute_density(self, sigma, vel, pos, info, hash_, vol, cellStart, neibslist, n):
        """COMPUTE_DENSITY: sigma written, vel.w rewritten in place"""
        p = capi.ptr
        P = self.params
        capi.check(self.lib.sphx_compute_density(self.ctx.handle, p(sigma), p(vel), p(pos), p(info), p(hash_), p(vol), p(cellStart),
                                                 p(neibslist), n, P.slength, P.influenceradius, self._s()))

    def forces_grenier(self, forces, cfl, pos, vel, info, hash_, cellStart, neibslist, sigma, n, frm, to, cfl_offset=0):
        p = capi.ptr
        P = self.params
        nb = C.c_uint32(0)
        capi.check(self.lib.sphx_forces_basicstep_grenier(self.ctx.handle, p(forces), p(cfl), p(pos), p(vel), p(info), p(hash_),
                                                          p(cellStart), p(neibslist), p(sigma), n, frm, to, P.deltap, P.slength,
                                                          P.dtadaptfactor, P.influenceradius, cfl_offset, D.SIMULATE, 1, 0.0,
                                                          C.byref(nb), self._s()))
        return int(nb.value)

    def euler_grenier(self, npos, nvel, nvol, opos, ovel, ovol, info, hash_, forces, n, d_dt, dt_scale, step):
        p = capi.ptr
        P = self.params
        capi.check(self.lib.sphx_euler_basicstep_grenier(self.ctx.handle, p(npos), p(nvel), p(nvol), p(opos), p(ovel), p(ovol), p(info),
                                                         p(hash_), p(forces), None, n, n, 0.0, p(d_dt), dt_scale, step, 0.0,
                                                         P.slength, P.influenceradius, D.SIMULATE, self._s()))

    def disable_free_surf_parts(self, pos, info, n):
        capi.check(self.lib.sphx_disable_free_surf_parts(self.ctx.handle, capi.ptr(pos), capi.ptr(info), n, n, self._s()))

    # ---- AbstractPostProcessEngine
    def postprocess(self, pptype, vort, vel_inout, info_inout, normals, pos, vel, info, hash_, cellStart, neibslist, n, cosf, cosn):
        p = capi.ptr
        capi.check(self.lib.sphx_postprocess(self.ctx.handle, int(pptype), p(vort), p(vel_inout), p(info_inout), p(normals),
                                             p(pos), p(vel), p(info), p(hash_), p(cellStart), p(neibslist), n, n,
                                             float(cosf), float(cosn), self._s()))

    # ---- SA open boundaries (ENABLE_INLET_OUTLET): the passes of gpusph_amd/csrc/sa_io.hip behind the driver sequence of
    # gpusph_amd.multigpu (_sa_post_euler_io).  The first five entry points are verified on the GPU; the condition passes, the
    # density summation, the forces, the diffusion and the water depth have run over their own source on the CPU only
    # (tests/hostemu) and the library's SA entry points still refuse ENABLE_INLET_OUTLET: a run of such a problem on the device
    # stops at its first SA call with SphxUnsupported until those kernels have passed their GPU parity tests.
    def sa_identify_corner_vertices(self, pos, info, hash_, vertices, cellStart, neibslist, n, range_end):
        p = capi.ptr
        capi.check(self.lib.sphx_sa_identify_corner_vertices(self.ctx.handle, p(pos), p(info), p(hash_), p(vertices), p(cellStart),
                                                             p(neibslist), n, range_end, self._s()))

    def sa_init_io_mass_vertex_count(self, forces, pos, vertices, hash_, info, cellStart, neibslist, n, range_end):
        """INIT_IO_MASS_VERTEX_COUNT: the counts in forces.w (BUFFER_FORCES as scratch); the halo's come by UPDATE_EXTERNAL"""
        p = capi.ptr
        self.memset(forces, 0)
        capi.check(self.lib.sphx_sa_init_io_mass_vertex_count(self.ctx.handle, p(vertices), p(hash_), p(info), p(cellStart), p(neibslist),
                                                              p(forces), p(pos), n, range_end, self._s()))

    def sa_init_io_mass(self, new_pos, pos, forces, vertices, hash_, info, cellStart, neibslist, n, range_end):
        p = capi.ptr
        capi.check(self.lib.sphx_sa_init_io_mass(self.ctx.handle, p(pos), p(forces), p(vertices), p(hash_), p(info), p(cellStart),
                                                 p(neibslist), p(new_pos), n, range_end, self.params.deltap, self._s()))

    def sa_segment_bc_io(self, vel, ggam, eulervel, pos, vertices, boundelements, info, hash_, cellStart, neibslist, n, range_end, step):
        p = capi.ptr
        capi.check(self.lib.sphx_sa_segment_bc_io(self.ctx.handle, p(vel), p(ggam), p(eulervel), p(pos), p(vertices), p(boundelements),
                                                  p(info), p(hash_), p(cellStart), p(neibslist), n, range_end, int(step), self._s()))

    def sa_vertex_bc_io(self, vel, old_pos, new_pos, ggam, eulervel, forces, vertices, boundelements, vertpos, info, hash_, next_ids,
                        count, cellStart, neibslist, n, range_end, max_particles, dt, step, num_open_vertices):
        p = capi.ptr
        capi.check(self.lib.sphx_sa_vertex_bc_io(self.ctx.handle, p(vel), p(old_pos), p(new_pos), p(ggam), p(eulervel), p(forces),
                                                 p(vertices), p(boundelements), p(vertpos[0]), p(vertpos[1]), p(vertpos[2]), p(info),
                                                 p(hash_), p(next_ids), p(count), p(cellStart), p(neibslist), n, range_end,
                                                 int(max_particles), self.params.deltap, float(np.float32(dt)), int(step),
                                                 int(num_open_vertices), self._s()))

    def sa_find_outgoing_segment(self, pos, vel, vertices, ggam, vertpos, boundelements, info, hash_, cellStart, neibslist, n, range_end):
        p = capi.ptr
        capi.check(self.lib.sphx_sa_find_outgoing_segment(self.ctx.handle, p(pos), p(vel), p(vertices), p(ggam), p(vertpos[0]), p(vertpos[1]),
                                                          p(vertpos[2]), p(boundelements), p(info), p(hash_), p(cellStart), p(neibslist),
                                                          n, range_end, self.params.influenceradius, self._s()))

    def sa_disable_outgoing_parts(self, pos, vertices, info, n):
        p = capi.ptr
        capi.check(self.lib.sphx_sa_disable_outgoing_parts(self.ctx.handle, p(pos), p(vertices), p(info), n, self._s()))

    def sa_density_sum_io(self, new_vel, new_ggam, forces, old_pos, new_pos, old_vel, old_eulervel, old_ggam, boundelements, vertpos, info,
                          hash_, cellStart, neibslist, n, range_end, dt):
        p = capi.ptr
        capi.check(self.lib.sphx_sa_density_sum_io(self.ctx.handle, p(new_vel), p(new_ggam), p(forces), p(old_pos), p(new_pos), p(old_vel),
                                                   p(old_eulervel), p(old_ggam), p(boundelements), p(vertpos[0]), p(vertpos[1]),
                                                   p(vertpos[2]), p(info), p(hash_), p(cellStart), p(neibslist), n, range_end,
                                                   float(np.float32(dt)), self._s()))

    def sa_density_diffusion_io(self, forces, pos, vel, ggam, boundelements, vertpos, info, hash_, cellStart, neibslist, n, range_end, dt):
        p = capi.ptr
        capi.check(self.lib.sphx_sa_compute_density_diffusion_io(self.ctx.handle, p(forces), p(pos), p(vel), p(ggam), p(boundelements),
                                                                 p(vertpos[0]), p(vertpos[1]), p(vertpos[2]), p(info), p(hash_),
                                                                 p(cellStart), p(neibslist), n, range_end, self.params.deltap,
                                                                 float(np.float32(dt)), self._s()))
        capi.check(self.lib.sphx_apply_density_diffusion(self.ctx.handle, p(vel), p(forces), p(info), n, range_end, float(np.float32(dt)),
                                                         self._s()))

    def forces_sa_io(self, forces, cfl, pos, vel, eulervel, info, hash_, cellStart, neibslist, ggam, boundelements, vertpos, n, frm, to,
                     cfl_offset, cfl_gamma=None):
        p = capi.ptr
        nb = C.c_uint32(0)
        capi.check(self.lib.sphx_forces_basicstep_sa_io(self.ctx.handle, p(forces), p(cfl), p(cfl_gamma), p(pos), p(vel), p(eulervel),
                                                        p(info), p(hash_), p(cellStart), p(neibslist), p(ggam), p(boundelements),
                                                        p(vertpos[0]), p(vertpos[1]), p(vertpos[2]), n, frm, to, self.params.deltap,
                                                        cfl_offset, C.byref(nb), self._s()))
        return int(nb.value)

    def sa_io_water_depth(self, depth, pos, info, hash_, cellStart, neibslist, n, frm, to):
        p = capi.ptr
        capi.check(self.lib.sphx_sa_io_water_depth(self.ctx.handle, p(depth), p(pos), p(info), p(hash_), p(cellStart), p(neibslist),
                                                   n, frm, to, self._s()))

    def flux_computation(self, flux, info, eulervel, boundelements, n, num_open_boundaries):
        """FLUX_COMPUTATION: flux[num_open_boundaries] (float32, device) = sum A_s (u_E . n_s) over the segments of each open boundary"""
        p = capi.ptr
        capi.check(self.lib.sphx_flux_computation(self.ctx.handle, p(flux), p(info), p(eulervel), p(boundelements), n, n,
                                                  int(num_open_boundaries), self._s()))
