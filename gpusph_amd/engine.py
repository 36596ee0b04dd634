"""One-GPU WCSPH timestep driver.

The command stream GPUSPH's host side sends to a GPUWorker lives in ONE place, gpusph_amd.multigpu.MultiGpuEngine
(neighbour phase, filters, predictor / corrector, dt feedback, body motion, halo exchange when there are several
devices).  TimestepEngine is that engine for a single domain (world = 1, no partition, no exchange) plus what is not
part of a time step: the repacking run mode, post-processing before writes, HotFile checkpoints and VTK output.

  neighbour phase  Integrator::buildNeibsPhase          src/Integrator.cc:94-250
                   GPUWorker::runCommand<CALCHASH..BUILDNEIBS>   src/GPUWorker.cc:1779-1905
  predictor/corr.  PredictorCorrector::initializePredCorrSequence
                                                         src/integrators/PredictorCorrectorIntegrator.cc:386-685
                   runCommand<FORCES_SYNC>, <EULER>      src/GPUWorker.cc:2188-2270
  dt feedback      GPUWorker.cc:2226-2229, GPUSPH.cc:636-699 (dt_next = min over both passes)
  repacking        RepackingIntegrator::initializeRepackingSequence   src/integrators/RepackingIntegrator.cc:278-420

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
from .multigpu import MultiGpuEngine


class TimestepEngine(MultiGpuEngine):
    def __init__(self, problem, device="cuda:0", allocated=None, clobber_neibslist=False,
                 track_particle_count=True):
        if not torch.cuda.is_available():
            raise capi.SphxError("TimestepEngine needs a HIP device (there is no CPU fallback)")
        n = len(problem.parts.info)
        if allocated is None and hasattr(problem, "max_parts"):      # a problem that creates particles says how many it may hold
            allocated = problem.max_parts(n)
        super().__init__(problem, device=device, rank=0, world=1, track_particle_count=track_particle_count,
                         allocated=int(allocated or n), clobber_neibslist=clobber_neibslist)
        self.lib, self.ctx, self.params = self.k.lib, self.k.ctx, self.k.params
        self.ncells = problem.grid_cells
        self.cfl_elems = self.cfl.numel()
        self.sspeed_cfl, self.max_kinvisc = self.k.sspeed_cfl, self.k.max_kinvisc
        self.sq_nl_radius = self.k.sq_nl_radius
        self.compute_object_forces = self.k.compute_object_forces
        self.num_bodies_parts = getattr(problem, "num_obstacle", 0)
        self.dt = float(np.float32(self.sp.dt))
        self.last_neibs_info = None

    # ------------------------------------------------------------------ compatibility views
    @property
    def n(self):
        return self.n_int

    @n.setter
    def n(self, value):
        self.n_int = self.n_local = self.edge_start = int(value)

    def _stream(self):
        return self.k._s()

    def _memset(self, t, value, stream=None):
        self.k.memset(t, value)

    def neibs_info(self):
        """getinfo + CHECK_NEIBSNUM (GPUSPH.cc:1850-1880); synchronises."""
        info = self.k.neibs_info()
        self.last_neibs_info = info
        if info.hasTooManyNeibs >= 0:
            raise capi.SphxError("particle id %d has too many neighbours (%d fluid + %d boundary)"
                                 % (info.hasTooManyNeibs, info.hasMaxNeibs[0], info.hasMaxNeibs[1]))
        return info

    def _forces(self, pos, vel, step, combine_min, run_mode=D.SIMULATE):
        """one forces pass (CALC_VISC, FORCES, dtreduce) on the given state"""
        self._forces_pass(pos, vel, combine_min, run_mode=run_mode, step=step)

    def _euler(self, step, dt_scale, run_mode=D.SIMULATE):
        kw = dict(xsph=self.xsph) if self.xsph is not None else {}
        if run_mode != D.SIMULATE:
            kw["run_mode"] = run_mode
        self.k.euler(self.pos2, self.vel2, self.pos, self.vel, self.info, self.hash, self.forces, self.n_local, self.d_dt,
                     dt_scale, step, **kw)

    def apply_filter(self, filtertype):
        """FILTER_CALL phase (src/integrators/PredictorCorrectorIntegrator.cc:831-859): read the unfiltered
        velocities, write the filtered ones, swap the two VEL buffers."""
        self._rows_for = None      # the velocity buffer may be rewritten behind torch's back (multigpu.py, _rows_for)
        n = self.n_local
        self.k.filter(int(filtertype), self.vel2, self.pos, self.vel, self.info, self.hash, self.cellStart, self.neibslist, n, n)
        self.vel, self.vel2 = self.vel2, self.vel

    # ------------------------------------------------------------------ post-processing (before writes)
    def postprocess(self, pptype, normals=False):
        """POSTPROCESS command (src/GPUWorker.cc runCommand<POSTPROCESS>): VORTICITY returns a [n,3] tensor,
        TESTPOINTS updates the velocity rows of test points in place, SURFACE_DETECTION updates FG_SURFACE (and
        INTERFACE_DETECTION also FG_INTERFACE) in INFO in place, and returns the normals when asked; FLUX_COMPUTATION returns the
        flux per open boundary, CALC_PRIVATE what the problem's calc_private makes of the state."""
        self._rows_for = None      # the velocity buffer may be rewritten behind torch's back (multigpu.py, _rows_for)
        if pptype == D.FLUX_COMPUTATION:
            return self.open_boundary_flux()
        if pptype == D.CALC_PRIVATE:
            return self.calc_private()
        n = self.n
        pp = self.problem.physparams
        out = vort = nrm = None
        if pptype == D.VORTICITY:
            vort = out = torch.empty((self.alloc, 3), dtype=torch.float32, device=self.device)
        detect = pptype in (D.SURFACE_DETECTION, D.INTERFACE_DETECTION)
        if detect and normals:
            nrm = out = torch.empty((self.alloc, 4), dtype=torch.float32, device=self.device)
        self.k.postprocess(pptype, vort, self.vel if pptype == D.TESTPOINTS else None, self.info if detect else None, nrm,
                           self.pos, self.vel, self.info, self.hash, self.cellStart, self.neibslist, n,
                           getattr(pp, "cosconeanglefluid", 0.86), getattr(pp, "cosconeanglenonfluid", 0.5))
        return None if out is None else out[:n]

    # ------------------------------------------------------------------ repacking run mode
    def repack_step(self):
        """one iteration of the repacking integrator (RepackingIntegrator::initializeRepackingSequence,
        src/integrators/RepackingIntegrator.cc:278-420): forces(REPACK) on step n, one full-dt Euler step."""
        self._rows_for = None      # the velocity buffer may be rewritten behind torch's back (multigpu.py, _rows_for)
        if self.iterations % self.sp.buildneibsfreq == 0:
            self.build_neibs()
            if self.sa and self.iterations == 0:     # initialisation step of the boundary conditions, REPACK variants (:80-235)
                self.sa_boundary_conditions(0, D.REPACK)
        self._forces(self.pos, self.vel, 1, 0, D.REPACK)
        self._euler(1, 1.0, D.REPACK)
        if self.sa:      # INTEGRATE_GAMMA of the new state (RepackingIntegrator.cc:386-405); gamma by quadrature only
            self.k.sa_integrate_gamma(self.gradgamma2, self.gradgamma, self.pos2, self.boundelements, self.vertpos, self.info, self.hash,
                                      self.cellStart, self.neibslist, self.n_local, self.n_local)
            self.gradgamma, self.gradgamma2 = self.gradgamma2, self.gradgamma
        self.pos, self.pos2 = self.pos2, self.pos
        self.vel, self.vel2 = self.vel2, self.vel
        self.k.time_advance(self.d_t, self.d_dt)
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
        self.k.disable_free_surf_parts(self.pos, self.info, self.n)
        if self.iterations > 0:
            self.build_neibs()
        if not reset:
            return
        # the restart below re-initialises velocities and densities only.  Option sets that carry more evolved state would need
        # what a run resumed from the repack file re-initialises as well (volumes: GPUSPH.cc:495; internal energy, k-epsilon
        # fields, the gamma of dynamic-gamma SA runs): refused rather than left half reset
        if self.grenier or self.energy_on or self.keps or (self.sa and getattr(self, "sa_dynamic_gamma", False)):
            raise ValueError("repack(reset=True) is built for runs whose evolved state is positions, velocities and densities: "
                             "not with SPH_GRENIER volumes, internal energy, k-epsilon or dynamic gamma (use reset=False and "
                             "re-initialise those buffers)")
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
    def download(self):
        return self.download_internal()

    # ------------------------------------------------------------------ output
    def write_vtp(self, path, vorticity=False, surface=False, forces=False):
        """VTKWriter::write (src/writers/VTKWriter.cc:610-830): a PART_*.vtp particle file; the post-processing
        engines run first when their output is asked for (SAVE command order of GPUSPH::doWrite)."""
        from . import vtkwriter
        vort = self.postprocess(D.VORTICITY).cpu().numpy() if vorticity else None
        nrm = self.postprocess(D.SURFACE_DETECTION, normals=True).cpu().numpy() if surface else None
        st = self.download()
        sa = {}
        if self.sa:      # SA_BOUNDARY: gamma with its gradient and the vertex ids of the segments (VTKWriter.cc:669-672,743-745)
            n = self.n
            sa = dict(gradgamma=self.gradgamma[:n].cpu().numpy(), vertices=self.vertices[:n].cpu().numpy().view(np.uint32))
        vtkwriter.write_vtp(path, self.problem, dict(pos=st["pos"], vel=st["vel"], info=st["info"].reshape(-1, 4), hash=st["hash"]),
                            vorticity=vort, normals=nrm, forces=st["forces"] if forces else None, **sa)

    # ------------------------------------------------------------------ checkpoints (GPUSPH HotFile v1)
    def _host_buffer_count(self):
        """size of GPUSPH's host buffer list for this option set, which HotFile::load compares with the header
        (src/writers/HotFile.cc:143): POS_GLOBAL, POS, VEL, INFO, HASH, + BOUNDELEMENTS, VERTICES, GRADGAMMA with
        SA_BOUNDARY, SPS_TURBVISC with SPS, EFFVISC with a generalized Newtonian rheology, VOLUME and SIGMA with
        SPH_GRENIER, INTERNAL_ENERGY with its flag (GPUSPH::allocateGlobalHostBuffers, GPUSPH.cc:868-941)"""
        return (5 + (3 if self.sa else 0) + (1 if self.sps else 0) + (1 if self.effvisc_on else 0)
                + (2 if self.grenier else 0) + (1 if self.energy_on else 0) + (4 if self.keps else 0))     # TKE, EPSILON, TURBVISC, EULERVEL

    def _hot_extra(self):
        """the option-dependent particle property buffers a HotFile stores beside pos/vel/info/hash: name -> device tensor"""
        ex = {}
        if self.energy_on:
            ex["energy"] = self.energy
        if self.sa:
            ex.update(boundelements=self.boundelements, gradgamma=self.gradgamma, vertices=self.vertices)
        if self.keps:
            ex.update(self.ke)
        if self.grenier:
            ex["vol"] = self.vol
        return ex

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
        arrays = dict(pos=st["pos"], vel=st["vel"], info=st["info"].reshape(-1, 4), hash=st["hash"])
        for k, t in self._hot_extra().items():
            arrays[k] = t[:self.n].cpu().numpy()
        hotfile.write_hotfile(path, arrays,
                              self.iterations, self.time(), self.current_dt(), bodies=bodies,
                              host_buffer_count=self._host_buffer_count())

    def load_hotfile(self, path):
        """HotFile::load + resume: particle buffers, iteration count, t and dt come from the file; the neighbour
        phase of the next step re-hashes and re-sorts them (calcHash, not the iteration-0 fixHash).  Body records restore
        the kinematic data of the moving bodies (readBody) and the centres of rotation of both engines."""
        self._rows_for = None      # the velocity buffer may be rewritten behind torch's back (multigpu.py, _rows_for)
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
        for k, t in self._hot_extra().items():
            if k not in a:
                raise capi.SphxError("HotFile lacks the '%s' buffer this option set evolves" % k)
            t[:n] = torch.from_numpy(a[k].view(np.int32) if k == "vertices" else a[k]).to(dev)
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
        """REDUCE_BODIES_FORCES for the single obstacle body through the library's reduction; returns (force3, torque3)."""
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
