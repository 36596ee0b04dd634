"""One-GPU WCSPH timestep driver.

Mirrors the command stream GPUSPH's host side sends to one GPUWorker (paths relative to the
GPUSPH tree), calling the HIP engines through the C ABI (include/sphx.h):

  neighbour phase  Integrator::buildNeibsPhase          src/Integrator.cc:94-250
                   GPUWorker::runCommand<CALCHASH..BUILDNEIBS>   src/GPUWorker.cc:1779-1905
  predictor/corr.  PredictorCorrector::initializePredCorrSequence
                                                         src/integrators/PredictorCorrectorIntegrator.cc:386-685
                   runCommand<FORCES_SYNC>, <EULER>      src/GPUWorker.cc:2188-2270
  dt feedback      GPUWorker.cc:2226-2229, GPUSPH.cc:636-699 (dt_next = min over both passes)

MI355X-first differences from the reference's control flow (results are unchanged):
  * the adaptive dt stays on the device: dtreduce writes a device scalar that the next
    step's Euler kernels read, so a step issues no host<->device synchronisation at all
    (the reference does a blocking 4-byte D2H after every forces pass);
  * everything is enqueued on one HIP stream and can be captured into a hipGraph;
  * torch is only the allocator / stream provider here.
"""
import ctypes as C
import numpy as np
import torch

from . import defs as D
from . import capi


def _dev_u32(arr, device):
    return torch.from_numpy(np.ascontiguousarray(arr).view(np.int32)).to(device)


class TimestepEngine:
    def __init__(self, problem, device="cuda:0", allocated=None, clobber_neibslist=False,
                 track_particle_count=True):
        if not torch.cuda.is_available():
            raise capi.SphxError("TimestepEngine needs a HIP device (there is no CPU fallback)")
        self.problem = problem
        self.device = torch.device(device)
        self.dev_index = self.device.index or 0
        torch.cuda.set_device(self.device)
        self.lib = capi.load()
        self.ctx = capi.Context(self.dev_index)
        arrs = problem.copy_to_array()
        self.n = len(arrs["hash"])
        self.alloc = int(allocated or self.n)
        self.params = problem.sphx_params(self.alloc)
        self.ctx.set_constants(self.params)
        self.ctx.reserve(self.alloc)
        if getattr(problem, "planes", None):      # GPUWorker::uploadPlanes -> setplanes
            nrm, gpos, lpos = problem.plane_tables()
            capi.check(self.lib.sphx_set_planes(self.ctx.handle, nrm.ctypes.data, gpos.ctypes.data, lpos.ctypes.data, len(nrm)))
        sp, pp = problem.simparams, problem.physparams
        self.sp = sp
        self.ncells = problem.grid_cells
        self.clobber_neibslist = clobber_neibslist
        self.track_particle_count = track_particle_count
        dev = self.device
        A = self.alloc
        f32, i32, i16 = torch.float32, torch.int32, torch.int16

        def up4(a):
            t = torch.zeros((A, 4), dtype=f32, device=dev)
            t[: self.n] = torch.from_numpy(a).to(dev)
            return t

        # state "step n" and "step n*" (double buffered like BUFFER_POS / BUFFER_VEL)
        self.pos = up4(arrs["pos"]); self.vel = up4(arrs["vel"])
        self.pos2 = torch.zeros_like(self.pos); self.vel2 = torch.zeros_like(self.vel)
        self.info = torch.zeros((A, 4), dtype=i16, device=dev)
        self.info[: self.n] = torch.from_numpy(arrs["info"].view(np.int16)).to(dev)
        self.hash = torch.zeros(A, dtype=i32, device=dev)
        self.hash[: self.n] = _dev_u32(arrs["hash"], dev)
        self.partindex = torch.zeros(A, dtype=i32, device=dev)
        self.cellStart = torch.empty(self.ncells, dtype=i32, device=dev)
        self.cellEnd = torch.empty(self.ncells, dtype=i32, device=dev)
        self.neibslist = torch.empty(int(sp.neiblistsize) * A, dtype=i16, device=dev)
        self.forces = torch.zeros((A, 4), dtype=f32, device=dev)
        self.cfl_elems = int(self.lib.sphx_forces_fmax_elements(A))
        self.cfl = torch.zeros(self.cfl_elems, dtype=f32, device=dev)
        self.cfl_temp = torch.zeros(max(int(self.lib.sphx_forces_fmax_temp_elements(self.cfl_elems)), 4), dtype=f32, device=dev)
        self.new_num = torch.zeros(1, dtype=i32, device=dev)
        self.segment_start = torch.zeros(4, dtype=i32, device=dev)
        self.num_bodies_parts = getattr(problem, "num_obstacle", 0)
        self.rbforces = torch.zeros((max(self.num_bodies_parts, 1), 4), dtype=f32, device=dev)
        self.rbtorques = torch.zeros_like(self.rbforces)
        # device-resident time step: dt of the current step, and the running min for the next
        self.dt = float(np.float32(sp.dt))
        self.d_dt = torch.full((1,), self.dt, dtype=f32, device=dev)
        self.d_dt_next = torch.full((1,), self.dt, dtype=f32, device=dev)
        self.d_t = torch.zeros(1, dtype=torch.float64, device=dev)
        self.t_host = 0.0                  # host copy of t, kept only for the callbacks of moving bodies
        self.iterations = 0
        self.sspeed_cfl = float(np.float32(np.float64(np.float32(max(pp.sscoeff))) * 1.1))  # GPUWorker.cc:3010-3011
        # GPUWorker::uploadConstants (src/GPUWorker.cc:3003-3006): maximum kinematic viscosity for the viscous dt limit
        self.max_kinvisc = float(np.float32(max(pp.kinematicvisc))) if sp.rheologytype == D.NEWTONIAN else 0.0
        self.compute_object_forces = 1 if sp.numforcesbodies > 0 else 0
        if self.num_bodies_parts:
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
        self.sq_nl_radius = float(np.float32(sp.nlSqInfluenceRadius))
        self.last_neibs_info = None
        # SPS: BUFFER_TAU (3 x float2) and BUFFER_SPS_TURBVISC, recomputed by CALC_VISC before every forces pass
        self.sps = sp.turbmodel == D.SPS
        if self.sps:
            self.tau = [torch.zeros((A, 2), dtype=f32, device=dev) for _ in range(3)]
            self.turbvisc = torch.zeros(A, dtype=f32, device=dev)
        # bodies with prescribed motion: host kinematics per integrator step (MOVE_BODIES), see bodies.py
        self.bodies = None
        if getattr(problem, "moving_bodies_callback", None) is not None and self.num_bodies_parts:
            from .bodies import MovingBodies
            self.bodies = MovingBodies(problem, problem.rb_cg_global)
        # ENABLE_XSPH: BUFFER_XSPH, written by every forces pass for the fluid particles, read by the Euler steps
        self.xsph = torch.zeros((A, 4), dtype=f32, device=dev) if (sp.simflags & D.ENABLE_XSPH) else None
        self.filters = []            # [(FilterType, frequency)], Problem::addFilter order
        self.profile_forces = None   # list of (start,end) torch events around each forces launch when enabled

    # ------------------------------------------------------------------ helpers
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _memset(self, t, value, stream):
        capi.check(self.lib.sphx_memset_async(t.data_ptr(), value, t.numel() * t.element_size(), stream))

    # ------------------------------------------------------------------ neighbour phase
    def build_neibs(self):
        L, h, s = self.lib, self.ctx.handle, self._stream()
        n = self.n
        p = capi.ptr
        if self.iterations == 0:
            capi.check(L.sphx_fix_hash(h, p(self.hash), p(self.partindex), p(self.info), None, n, s))
        else:
            capi.check(L.sphx_calc_hash(h, p(self.pos), p(self.hash), p(self.partindex), p(self.info), None, n, s))
        capi.check(L.sphx_sort(h, p(self.hash), p(self.info), p(self.partindex), n, s))
        self._memset(self.cellStart, 0xFF, s)
        self._memset(self.cellEnd, 0xFF, s)
        capi.check(L.sphx_reorder(h, None, p(self.cellStart), p(self.cellEnd), p(self.pos2), p(self.vel2),
                                  p(self.pos), p(self.vel), p(self.info), p(self.hash), p(self.partindex),
                                  n, p(self.new_num), s))
        self.pos, self.pos2 = self.pos2, self.pos
        self.vel, self.vel2 = self.vel2, self.vel
        if self.track_particle_count:
            self.n = int(self.new_num.item()) & 0xFFFFFFFF   # DOWNLOAD_NEWNUMPARTS (sync), GPUWorker.cc:1471-1515
            n = self.n
        if self.clobber_neibslist:
            self._memset(self.neibslist, 0xFF, s)
        capi.check(L.sphx_neibs_resetinfo(h, s))
        capi.check(L.sphx_build_neibs(h, p(self.neibslist), p(self.pos), p(self.info), p(self.hash),
                                      p(self.cellStart), p(self.cellEnd), n, n, self.ncells,
                                      self.sq_nl_radius, self.sq_nl_radius, s))

    def neibs_info(self):
        """getinfo + CHECK_NEIBSNUM (GPUSPH.cc:1850-1880); synchronises."""
        info = capi.NeibsInfo()
        capi.check(self.lib.sphx_neibs_getinfo(self.ctx.handle, C.byref(info), self._stream()))
        self.last_neibs_info = info
        if info.hasTooManyNeibs >= 0:
            raise capi.SphxError("particle id %d has too many neighbours (%d fluid + %d boundary)"
                                 % (info.hasTooManyNeibs, info.hasMaxNeibs[0], info.hasMaxNeibs[1]))
        return info

    # ------------------------------------------------------------------ density filters
    def add_filter(self, filtertype, frequency):
        """ProblemCore::addFilter: run `filtertype` every `frequency` iterations (src/ProblemCore.h addFilter)."""
        self.filters.append((int(filtertype), int(frequency)))

    def apply_filter(self, filtertype):
        """FILTER_CALL phase (src/integrators/PredictorCorrectorIntegrator.cc:831-859): read the unfiltered
        velocities, write the filtered ones, swap the two VEL buffers."""
        L, h, s = self.lib, self.ctx.handle, self._stream()
        p = capi.ptr
        n = self.n
        capi.check(L.sphx_filter_process(h, int(filtertype), p(self.vel2), p(self.pos), p(self.vel), p(self.info), p(self.hash),
                                         p(self.cellStart), p(self.neibslist), n, n, self.params.slength,
                                         self.params.influenceradius, s))
        self.vel, self.vel2 = self.vel2, self.vel

    # ------------------------------------------------------------------ post-processing (before writes)
    def postprocess(self, pptype, normals=False):
        """POSTPROCESS command (src/GPUWorker.cc runCommand<POSTPROCESS>): VORTICITY returns a [n,3] tensor,
        TESTPOINTS updates the velocity rows of test points in place, SURFACE_DETECTION updates FG_SURFACE (and
        INTERFACE_DETECTION also FG_INTERFACE) in INFO in place, and returns the normals when asked."""
        L, h, s = self.lib, self.ctx.handle, self._stream()
        p = capi.ptr
        n = self.n
        pp = self.problem.physparams
        out = None
        vort = nrm = None
        if pptype == D.VORTICITY:
            vort = out = torch.empty((self.alloc, 3), dtype=torch.float32, device=self.device)
        detect = pptype in (D.SURFACE_DETECTION, D.INTERFACE_DETECTION)
        if detect and normals:
            nrm = out = torch.empty((self.alloc, 4), dtype=torch.float32, device=self.device)
        capi.check(L.sphx_postprocess(h, int(pptype), p(vort), p(self.vel) if pptype == D.TESTPOINTS else None,
                                      p(self.info) if detect else None, p(nrm),
                                      p(self.pos), p(self.vel), p(self.info), p(self.hash), p(self.cellStart),
                                      p(self.neibslist), n, n, float(getattr(pp, "cosconeanglefluid", 0.86)),
                                      float(getattr(pp, "cosconeanglenonfluid", 0.5)), s))
        return None if out is None else out[:n]

    # ------------------------------------------------------------------ forces / euler
    def _forces(self, pos, vel, step, combine_min, run_mode=D.SIMULATE):
        L, h, s = self.lib, self.ctx.handle, self._stream()
        p = capi.ptr
        sp = self.sp
        n = self.n
        nb = C.c_uint32(0)
        rb = self.num_bodies_parts > 0
        prof = self.profile_forces is not None
        tau = [None, None, None]
        if self.sps and run_mode == D.SIMULATE:
            # CALC_VISC on the state the forces read (PredictorCorrectorIntegrator.cc:460-480)
            capi.check(L.sphx_calc_visc(h, p(self.tau[0]), p(self.tau[1]), p(self.tau[2]), p(self.turbvisc), p(pos), p(vel),
                                        p(self.info), p(self.hash), p(self.cellStart), p(self.neibslist), n, n,
                                        self.params.deltap, self.params.slength, self.params.influenceradius, s))
            tau = [p(t) for t in self.tau]
        self._memset(self.cfl, 0, s)          # pre_forces clobbers BUFFER_CFL (GPUWorker.cc:1970-1972)
        if prof:
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
        capi.check(L.sphx_forces_basicstep(h, p(self.forces), p(self.cfl), p(self.rbforces) if rb else None,
                                           p(self.rbtorques) if rb else None, p(pos), p(vel), p(self.info), p(self.hash),
                                           p(self.cellStart), p(self.neibslist), tau[0], tau[1], tau[2], p(self.xsph),
                                           n, 0, n, self.params.deltap, self.params.slength, self.params.dtadaptfactor,
                                           self.params.influenceradius, 0, run_mode, step, self.dt,
                                           self.compute_object_forces, C.byref(nb), s))
        if prof:
            e1.record()
            self.profile_forces.append((e0, e1))
        capi.check(L.sphx_forces_dtreduce_device(h, self.params.slength, self.params.dtadaptfactor, self.sspeed_cfl,
                                                 self.max_kinvisc, p(self.cfl), p(self.cfl_temp), nb.value,
                                                 p(self.d_dt_next), combine_min, s))

    def _euler(self, step, dt_scale, run_mode=D.SIMULATE):
        L, h, s = self.lib, self.ctx.handle, self._stream()
        p = capi.ptr
        n = self.n
        capi.check(L.sphx_euler_basicstep(h, p(self.pos2), p(self.vel2), p(self.pos), p(self.vel), p(self.info),
                                          p(self.hash), p(self.forces), p(self.xsph), n, n, 0.0, p(self.d_dt), dt_scale,
                                          step, 0.0, self.params.slength, self.params.influenceradius, run_mode, s))

    def step(self):
        """one full predictor-corrector time step; no host synchronisation."""
        if self.iterations % self.sp.buildneibsfreq == 0:
            self.build_neibs()
        # filters run after the neighbour phase of every iteration > 0 whose number their frequency divides
        # (PredictorCorrector::next_phase, src/integrators/PredictorCorrectorIntegrator.cc:1011-1041)
        if self.iterations > 0:
            for ftype, freq in self.filters:
                if self.iterations % freq == 0:
                    self.apply_filter(ftype)
        if self.bodies is not None:     # the callback needs dt on the host: ONE synchronisation per step (t is summed on both sides)
            dt_host = float(self.d_dt.item())
            t_host = self.t_host
        # predictor: forces(step n) -> n* = n + dt/2 f
        self._forces(self.pos, self.vel, 1, 0)
        if self.bodies is not None:
            self._move_bodies(1, dt_host, t_host)
        self._euler(1, 0.5)
        # corrector: forces(step n*) -> n+1 = n + dt f*   (written over n*, then renamed to n)
        self._forces(self.pos2, self.vel2, 2, 1)
        if self.bodies is not None:
            self._move_bodies(2, dt_host, t_host)
        self._euler(2, 1.0)
        if self.bodies is not None:     # EULER_UPLOAD_OBJECTS_CG in the post-corrector phase (PredictorCorrectorIntegrator.cc:331-332)
            m = self._last_motion
            capi.check(self.lib.sphx_set_rb_cg_integration(self.ctx.handle, m["cg_grid"].ctypes.data, m["cg_pos"].ctypes.data, len(self.bodies)))
        self.pos, self.pos2 = self.pos2, self.pos
        self.vel, self.vel2 = self.vel2, self.vel
        # TIME_STEP_EPILOGUE: t += dt ; dt = min(dt_pred, dt_corr)
        self.d_t.add_(self.d_dt.double())
        if self.bodies is not None:
            self.t_host += dt_host         # the same double += float as on the device
        self.d_dt, self.d_dt_next = self.d_dt_next, self.d_dt
        self.iterations += 1

    def _move_bodies(self, step, dt, t):
        """MOVE_BODIES + UPLOAD_OBJECTS_MATRICES/VELOCITIES (+ FORCES_UPLOAD_OBJECTS_CG for bodies with force feedback),
        src/integrators/PredictorCorrectorIntegrator.cc:550-570"""
        m = self.bodies.timestep(step, dt, t)
        self._last_motion = m          # keeps the host arrays alive until the calls have consumed them
        nb = len(self.bodies)
        capi.check(self.lib.sphx_set_rb_motion(self.ctx.handle, m["trans"].ctypes.data, m["rot"].ctypes.data,
                                               m["lvel"].ctypes.data, m["avel"].ctypes.data, nb))
        if self.sp.numforcesbodies > 0:
            capi.check(self.lib.sphx_set_rb_cg_forces(self.ctx.handle, m["cg_grid"].ctypes.data, m["cg_pos"].ctypes.data, nb))

    def run(self, steps):
        for _ in range(steps):
            self.step()

    # ------------------------------------------------------------------ repacking run mode
    def repack_step(self):
        """one iteration of the repacking integrator (RepackingIntegrator::initializeRepackingSequence,
        src/integrators/RepackingIntegrator.cc:278-420): forces(REPACK) on step n, one full-dt Euler step."""
        if self.iterations % self.sp.buildneibsfreq == 0:
            self.build_neibs()
        self._forces(self.pos, self.vel, 1, 0, D.REPACK)
        self._euler(1, 1.0, D.REPACK)
        self.pos, self.pos2 = self.pos2, self.pos
        self.vel, self.vel2 = self.vel2, self.vel
        self.d_t.add_(self.d_dt.double())
        self.d_dt, self.d_dt_next = self.d_dt_next, self.d_dt
        self.iterations += 1

    def repack(self, maxiter=None, reset=True):
        """`GPUSPH --repack`: run the repacking integrator for repack_maxiter iterations (GPUSPH.cc:196-201,
        676-692), disable the free-surface lid particles, rebuild the neighbour list (FINISH_REPACKING -> NEIBS_LIST ->
        PREPARE_SIMULATION), then, like a run resumed from the repack file (GPUSPH.cc:425-450, ProblemCore::resetBuffers),
        restart the clock with zero velocities and the problem's initial density at the new positions."""
        if not (self.sp.simflags & D.ENABLE_REPACKING):
            raise ValueError("Repacking is not enabled in the problem")      # src/main.cc:357-358
        maxiter = int(self.sp.repack_maxiter if maxiter is None else maxiter)
        for _ in range(maxiter):
            self.repack_step()
        s = self._stream()
        capi.check(self.lib.sphx_disable_free_surf_parts(self.ctx.handle, capi.ptr(self.pos), capi.ptr(self.info),
                                                         self.n, self.n, s))
        if self.iterations > 0:
            self.build_neibs()
        if not reset:
            return
        self.iterations = 0
        self.d_t.zero_()
        self.t_host = 0.0
        self.dt = float(np.float32(self.sp.dt))
        self.d_dt.fill_(self.dt); self.d_dt_next.fill_(self.dt)
        n = self.n
        pos = self.pos[:n].cpu().numpy()
        hsh = self.hash[:n].cpu().numpy().view(np.uint32)
        rho = self.problem.initial_density(self.problem.global_pos(pos, hsh))
        vel = np.zeros((n, 4), dtype=np.float32)
        vel[:, 3] = rho
        self.vel[:n] = torch.from_numpy(vel).to(self.device)

    # ------------------------------------------------------------------ host views
    def current_dt(self):
        return float(self.d_dt.item())

    def time(self):
        return float(self.d_t.item())

    def download(self):
        n = self.n
        out = {
            "pos": self.pos[:n].cpu().numpy(), "vel": self.vel[:n].cpu().numpy(),
            "info": self.info[:n].cpu().numpy().view(np.uint16),
            "hash": self.hash[:n].cpu().numpy().view(np.uint32),
            "forces": self.forces[:n].cpu().numpy(),
        }
        return out

    # ------------------------------------------------------------------ output
    def write_vtp(self, path, vorticity=False, surface=False, forces=False):
        """VTKWriter::write (src/writers/VTKWriter.cc:610-830): a PART_*.vtp particle file; the post-processing
        engines run first when their output is asked for (SAVE command order of GPUSPH::doWrite)."""
        from . import vtkwriter
        vort = self.postprocess(D.VORTICITY).cpu().numpy() if vorticity else None
        nrm = self.postprocess(D.SURFACE_DETECTION, normals=True).cpu().numpy() if surface else None
        st = self.download()
        vtkwriter.write_vtp(path, self.problem, dict(pos=st["pos"], vel=st["vel"], info=st["info"].reshape(-1, 4), hash=st["hash"]),
                            vorticity=vort, normals=nrm, forces=st["forces"] if forces else None)

    # ------------------------------------------------------------------ checkpoints (GPUSPH HotFile v1)
    def _host_buffer_count(self):
        """size of GPUSPH's host buffer list for this option set, which HotFile::load compares with the header
        (src/writers/HotFile.cc:143): POS_GLOBAL, POS, VEL, INFO, HASH, + SPS_TURBVISC with SPS (GPUSPH.cc host allocation)"""
        return 5 + (1 if self.sps else 0)

    def save_hotfile(self, path):
        """HotFile::save of the current state (src/writers/HotFile.cc:86-118); readable by GPUSPH --resume and by
        the reference's scripts/hotdiff.py.  One body record per body, from the LIVE kinematic data (writeBody :285-340)."""
        from . import hotfile
        st = self.download()
        bodies = []
        if self.num_bodies_parts:
            pr = self.problem
            nb = len(pr.rb_firstindex)
            for b in range(nb):
                first = int(pr.rb_firstindex[b])
                forces_body = self.sp.numforcesbodies > b
                if self.bodies is not None:
                    kd, kd0 = self.bodies.kdata[b], self.bodies.initial[b]
                    crot, lvel, avel = kd.crot, kd.lvel, kd.avel
                    icrot = kd0.crot
                else:
                    crot = icrot = pr.m_origin + (pr.rb_cg_gridpos[b] + 0.5) * pr.m_cellsize + pr.rb_cg_pos[b]
                    lvel = avel = [0.0, 0.0, 0.0]
                bodies.append(dict(index=b, id=b, type=hotfile.MB_FORCES_MOVING if forces_body else hotfile.MB_MOVING,
                                   numparts=self.num_bodies_parts // nb,
                                   firstindex=first if forces_body else 0,
                                   lastindex=(self.num_bodies_parts // nb - 1) if forces_body else 0,
                                   crot=crot, lvel=lvel, avel=avel, orientation=[1, 0, 0, 0],
                                   initial_crot=icrot, initial_lvel=[0, 0, 0], initial_avel=[0, 0, 0],
                                   initial_orientation=[1, 0, 0, 0]))
        hotfile.write_hotfile(path, dict(pos=st["pos"], vel=st["vel"], info=st["info"].reshape(-1, 4), hash=st["hash"]),
                              self.iterations, self.time(), self.current_dt(), bodies=bodies,
                              host_buffer_count=self._host_buffer_count())

    def load_hotfile(self, path):
        """HotFile::load + resume: particle buffers, iteration count, t and dt come from the file; the neighbour
        phase of the next step re-hashes and re-sorts them (calcHash, not the iteration-0 fixHash).  Body records restore
        the kinematic data of the moving bodies (readBody) and the centres of rotation of both engines."""
        from . import hotfile
        hf = hotfile.read_hotfile(path)
        a = hf["arrays"]
        n = hf["particles"]
        if n > self.alloc:
            raise capi.SphxError("HotFile has %d particles, %d allocated" % (n, self.alloc))
        nb_sim = len(self.problem.rb_firstindex) if self.num_bodies_parts else 0
        if len(hf["bodies"]) != nb_sim:       # check_counts_match("body", ...), HotFile.cc:148
            raise capi.SphxError("mismatched body count; HotFile has %d, simulation has %d" % (len(hf["bodies"]), nb_sim))
        dev = self.device
        self.n = n
        self.pos[:n] = torch.from_numpy(a["pos"]).to(dev); self.vel[:n] = torch.from_numpy(a["vel"]).to(dev)
        self.info[:n] = torch.from_numpy(a["info"].view(np.int16)).to(dev)
        self.hash[:n] = torch.from_numpy(a["hash"].view(np.int32)).to(dev)
        self.iterations = int(hf["iterations"])
        self.dt = float(np.float32(hf["dt"]))
        self.d_dt.fill_(self.dt); self.d_dt_next.fill_(self.dt)
        self.d_t.fill_(float(hf["t"]))
        self.t_host = float(hf["t"])
        if self.bodies is not None:
            for rec in hf["bodies"]:
                b = int(rec["index"])
                kd = self.bodies.kdata[b]
                kd.crot = np.array(rec["crot"], dtype=np.float64)
                kd.lvel = np.array(rec["lvel"], dtype=np.float64); kd.avel = np.array(rec["avel"], dtype=np.float64)
                self.bodies.storage[b] = type(kd)(crot=kd.crot.copy(), lvel=kd.lvel.copy(), avel=kd.avel.copy())
                self.bodies.initial[b].crot = np.array(rec["initial_crot"], dtype=np.float64)
            nb = len(self.bodies)
            gp = np.zeros((nb, 3), dtype=np.int32); lp = np.zeros((nb, 3), dtype=np.float32)
            for b in range(nb):
                gp[b], lp[b] = self.bodies.grid_and_local(self.bodies.kdata[b].crot)
            capi.check(self.lib.sphx_set_rb_cg(self.ctx.handle, gp.ctypes.data, lp.ctypes.data, nb))
        if self.iterations % self.sp.buildneibsfreq != 0:
            self.build_neibs()      # a resumed run always starts with a neighbour phase (GPUSPH::runSimulation)
        return hf

    def reduce_rb_forces(self):
        """REDUCE_BODIES_FORCES for the single obstacle body; returns (force3, torque3)."""
        if not self.num_bodies_parts:
            return None
        nbp = self.num_bodies_parts
        keys = torch.zeros(nbp, dtype=torch.int32, device=self.device)
        last = np.array([nbp - 1], dtype=np.uint32)
        tf = np.zeros(3, dtype=np.float32); tt = np.zeros(3, dtype=np.float32)
        capi.check(self.lib.sphx_reduce_rb_forces(self.ctx.handle, capi.ptr(self.rbforces), capi.ptr(self.rbtorques),
                                                  capi.ptr(keys), last.ctypes.data, tf.ctypes.data, tt.ctypes.data,
                                                  1, nbp, self._stream()))
        return tf, tt
