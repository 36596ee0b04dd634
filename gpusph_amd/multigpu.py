"""Slab-decomposed multi-GPU timestep driver: one process per GPU, halo exchange over torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" on CPU in the test suite).

What it mirrors in GPUSPH (paths relative to the GPUSPH tree) and what it changes:
  * device map: equal split of the cells along COORD3, the slowest linearisation axis, so that each
    device's edge layer and each imported halo is ONE contiguous index range (src/linearization.h:28-35;
    ProblemCore::fillDeviceMapByAxis src/ProblemCore.cc:1061-1116).  DamBreak3D splits along Y
    (src/problems/DamBreak3D.cu:217-220): use linearisation "xzy" so that COORD3 = y.
  * cell types INNER / INNER_EDGE / OUTER_EDGE / OUTER in the two high hash bits
    (GPUWorker::createCompactDeviceMap src/GPUWorker.cc:1560-1634, src/multi_gpu_defines.h:56-83): the
    sort then orders particles [inner | inner edge | outer edge | outer] and REORDER returns the
    segment starts.
  * migration is implicit as in the reference (src/GPUWorker.cc:358-365,1432-1460): halo copies are
    integrated redundantly (EULER runs on all particles), a particle that crosses the slab boundary sorts
    into the other segment at the next rebuild, the old owner crops it, the new owner already has it.
  * halo import: the reference PULLS bursts with cudaMemcpyPeerAsync per (burst, buffer) and keeps the
    per-cell offsets on the host (src/GPUWorker.cc:711-921).  Here each rank SENDS its two edge layers
    (contiguous ranges) to its two neighbours and receives theirs, as ONE grouped send/recv per
    exchange on a dedicated stream: after each re-sort pos+vel+info+hash (44 B/particle), after each
    forces pass the forces (16 B/particle) -- strictly nearest-neighbour traffic, one xGMI link per
    pair of devices, no collective on the data path.  The forces of the edge stripe are computed
    first and exchanged while the inner stripe computes (FORCES_ENQUEUE/COMPLETE, GPUWorker.cc:2086-2185).
  * dt: min over devices (GPUSPH.cc:650-657) as an all_reduce(MIN) of one device float per step.
Results are bit-identical to the single-device run (same per-particle neighbour order and arithmetic).
"""
import os
import numpy as np
import torch

from . import defs as D


class SlabPartition:
    """equal split of the COORD3 planes among `world` devices"""

    def __init__(self, problem, world):
        c1, c2, c3 = D.LINEARIZATIONS[problem.linearization]
        gs = problem.m_gridsize
        self.gs3 = int(gs[c3])
        self.plane = int(gs[c1]) * int(gs[c2])          # cells per COORD3 plane = contiguous hash range
        self.world = world
        # a split axis that is periodic closes the chain of slabs into a ring: the first and the last slab are neighbours through
        # the periodic face and exchange their outermost planes like any two neighbours (in the reference this falls out of the
        # device map and the periodic neighbour cells, src/ProblemCore.cc:1061-1116).  Two slabs on a ring are each other's
        # neighbour on both sides: the transports order their receives for that (halo.py, halo.hip).
        self.ring = world > 1 and bool(problem.simparams.periodicbound & (1 << c3))
        # fillDeviceMapByAxis (src/ProblemCore.cc:1061-1116): fewer than 3 planes per device on average are refused; a device
        # takes round(planes / devices) planes, the last one what is left
        if world > 1 and self.gs3 / float(world) < 3.0:
            raise ValueError("not enough cells along the split axis (%d planes, %d devices: fewer than 3 per device; "
                             "src/ProblemCore.cc:1089-1090)" % (self.gs3, world))
        per = int(np.floor(self.gs3 / float(world) + 0.5))
        self.lo = [min(d * per, self.gs3) for d in range(world)]
        self.hi = [min((d + 1) * per, self.gs3) for d in range(world)]
        self.hi[-1] = self.gs3
        # the rounding rule can starve the last device (28 planes on 8 devices: 7 x 4 and nothing left; 21 on 6: one plane),
        # which the average test above does not see.  A slab needs its two edge planes to be different planes (each is sent
        # to one neighbour and has its own outer edge plane behind it), so such a split is replaced by the even one:
        # floor(planes / devices) each, the first planes % devices slabs one more
        if world > 1 and min(h - l for l, h in zip(self.lo, self.hi)) < 2:
            base, rem = divmod(self.gs3, world)
            counts = [base + (1 if d < rem else 0) for d in range(world)]
            self.lo = [sum(counts[:d]) for d in range(world)]
            self.hi = [self.lo[d] + counts[d] for d in range(world)]
        if world > 1 and min(h - l for l, h in zip(self.lo, self.hi)) < 2:
            raise ValueError("slab split of %d planes over %d devices leaves a device with fewer than 2 planes: %s"
                             % (self.gs3, world, list(zip(self.lo, self.hi))))

    def plane_types(self, rank):
        """CELLTYPE_* of every COORD3 plane as seen by `rank`"""
        t = np.full(self.gs3, D.CELLTYPE_OUTER_CELL, dtype=np.uint32)
        lo, hi = self.lo[rank], self.hi[rank]
        t[lo:hi] = D.CELLTYPE_INNER_CELL
        if rank > 0 or self.ring:
            t[lo] = D.CELLTYPE_INNER_EDGE_CELL
            t[(lo - 1) % self.gs3] = D.CELLTYPE_OUTER_EDGE_CELL
        if rank < self.world - 1 or self.ring:
            t[hi - 1] = D.CELLTYPE_INNER_EDGE_CELL
            t[hi % self.gs3] = D.CELLTYPE_OUTER_EDGE_CELL
        return t

    def compact_device_map(self, rank):
        """per-cell CELLTYPE_*_SHIFTED (BUFFER_COMPACT_DEV_MAP)"""
        return np.repeat(self.plane_types(rank) << np.uint32(30), self.plane).astype(np.uint32)

    def local_mask(self, rank, hashes):
        """particles a rank holds initially: its own planes plus the one-plane halo on each side"""
        c3 = (hashes & D.CELLTYPE_BITMASK) // self.plane
        return self.plane_types(rank)[c3] != D.CELLTYPE_OUTER_CELL


class MultiGpuEngine:
    def __init__(self, problem, device, rank, world, kernels=None, track_particle_count=True, margin=1.25,
                 overlap=True, allocated=None, clobber_neibslist=False, transport=None):
        self.sa = problem.simparams.boundarytype == D.SA_BOUNDARY
        self.grenier = problem.simparams.sph_formulation == D.SPH_GRENIER
        self.effvisc_on = problem.simparams.rheologytype > D.NEWTONIAN        # NEEDS_EFFECTIVE_VISC
        self.keps = problem.simparams.turbmodel == D.KEPSILON
        self.problem = problem
        self.rank, self.world = rank, world
        self.device = torch.device(device)
        self.is_cuda = self.device.type == "cuda"
        self.sp = problem.simparams
        self.part = SlabPartition(problem, world)
        arrs = problem.copy_to_array()
        mask = self.part.local_mask(rank, arrs["hash"]) if world > 1 else np.ones(len(arrs["hash"]), dtype=bool)
        n0 = int(mask.sum())
        self.alloc = int(allocated) if allocated is not None else int(n0 * margin) + 4096
        self.clobber_neibslist = clobber_neibslist
        if kernels is None:
            from .kernels import HipKernels
            kernels = HipKernels(problem, self.alloc, self.device)
        self.k = kernels
        # The EOS rows of the plain forces engine ride with the Euler step (include/sphx.h, sphx_eos_rows_follow_euler): this driver
        # owns the sequence of a step, so it can vouch for the velocity buffer between an Euler step and the forces pass that reads
        # it.  _rows_for = (the tensor the last Euler step wrote, its torch version counter): anything that rewrites it through
        # torch bumps the counter, the library's own rewrites (sort, filters, halo import) go to the other buffer or clear this
        self._rows_for = None
        self._rows_follow = (not self.sa and not self.grenier and not self.effvisc_on and hasattr(kernels, "eos_rows_follow_euler")
                             and problem.simparams.sph_formulation in (D.SPH_F1, D.SPH_F2))
        if self._rows_follow:
            kernels.eos_rows_follow_euler(True)
        # who moves the edge layers: torch.distributed by default, or a transport the caller built (halo.CapiTransport:
        # the library's own sphx_halo_* entry points; a callable gets the kernels of this rank and returns the transport)
        self.transport = None
        if world > 1:
            if callable(transport):
                transport = transport(self.k)
            if transport is None:
                import torch.distributed as dist
                from .halo import TorchTransport
                transport = TorchTransport(dist, self.is_cuda)
            self.transport = transport
        dev, A = self.device, self.alloc
        f32, i32, i16 = torch.float32, torch.int32, torch.int16

        def up(a, dtype, shape):
            t = torch.zeros(shape, dtype=dtype, device=dev)
            t[:n0] = torch.from_numpy(a[mask]).to(dev)
            return t

        self.n_local = n0
        self.pos = up(arrs["pos"], f32, (A, 4)); self.vel = up(arrs["vel"], f32, (A, 4))
        self.pos2 = torch.zeros_like(self.pos); self.vel2 = torch.zeros_like(self.vel)
        self.info = up(arrs["info"].view(np.int16), i16, (A, 4))
        self.hash = up(arrs["hash"].view(np.int32), i32, (A,))
        self.partindex = torch.zeros(A, dtype=i32, device=dev)
        self.ncells = problem.grid_cells
        self.cellStart = torch.empty(self.ncells, dtype=i32, device=dev)
        self.cellEnd = torch.empty(self.ncells, dtype=i32, device=dev)
        self.neibslist = torch.empty(int(self.sp.neiblistsize) * A, dtype=i16, device=dev)
        self.forces = torch.zeros((A, 4), dtype=f32, device=dev)
        # what every forces pass starts from zero -- the CFL maxima and, with bodies, the RB_FORCES / RB_TORQUES rows -- lies in
        # one allocation, so that it is one memset per pass (three fill kernels of a few microseconds each otherwise)
        ncfl = (self.k.fmax_elements(A) + 8 + 3) // 4 * 4
        nrb = max(getattr(problem, "num_obstacle", 0), 1)
        self._zeroed = torch.zeros(ncfl + 8 * nrb, dtype=f32, device=dev)
        self.cfl = self._zeroed[:ncfl]
        self.cfl_temp = torch.zeros(max(self.k.fmax_temp_elements(self.cfl.numel()), 4), dtype=f32, device=dev)
        self.new_num = torch.zeros(1, dtype=i32, device=dev)
        self.segment_start = torch.zeros(4, dtype=i32, device=dev)
        self.has_rb = bool(getattr(problem, "num_obstacle", 0))      # BUFFER_RB_FORCES / RB_TORQUES rows of body particles
        self.rbforces = self._zeroed[ncfl:ncfl + 4 * nrb].view(nrb, 4)
        self.rbtorques = self._zeroed[ncfl + 4 * nrb:].view(nrb, 4)
        self.devmap = (torch.from_numpy(self.part.compact_device_map(rank).view(np.int32)).to(dev)
                       if world > 1 else None)
        dt0 = float(np.float32(self.sp.dt))
        self.d_dt = torch.full((1,), dt0, dtype=f32, device=dev)
        self.d_dt_next = torch.full((1,), dt0, dtype=f32, device=dev)
        self.d_t = torch.zeros(1, dtype=torch.float64, device=dev)      # simulated time, summed on the device like dt
        self.iterations = 0
        self.n_int = n0
        self.edge_start = n0
        self.send_l = self.send_r = self.recv_l = self.recv_r = (0, 0)
        self.overlap = overlap and self.is_cuda and world > 1
        self.comm_stream = torch.cuda.Stream(device=dev) if self.overlap else None
        # The tiled forces kernel is a persistent grid of one workgroup per CU that owns almost all of the CU's LDS, so the
        # RCCL send/recv kernels of the overlapped exchange would queue behind the inner stripe.  Leave a few CUs (one per
        # XCD by default) out of that grid when the exchange is overlapped (SPHX_COMM_CUS overrides; 0 = none).
        self.comm_cus = int(os.environ.get("SPHX_COMM_CUS", "8")) if self.overlap else 0
        if self.comm_cus and hasattr(self.k, "reserve_comm_cus"):
            self.k.reserve_comm_cus(self.comm_cus)
        # exchange accounting (bench.py --gpus N): bytes sent + received per call, stall of the compute stream on the exchange
        self.halo_bytes = 0
        self.exchange_events = None      # list of (before, after) events on the compute stream around the wait, when enabled
        self.profile_forces = None
        self.track_particle_count = track_particle_count
        # SPS: BUFFER_TAU as three float2 arrays, computed for the internal particles before each forces pass and imported
        # for the halo (CALC_VISC + UPDATE_EXTERNAL, src/integrators/PredictorCorrectorIntegrator.cc:460-480)
        self.sps = self.sp.turbmodel == D.SPS
        self.tau = [torch.zeros((A, 2), dtype=f32, device=dev) for _ in range(3)] if self.sps else None
        self.turbvisc = torch.zeros(A, dtype=f32, device=dev) if self.sps else None      # BUFFER_SPS_TURBVISC
        # ENABLE_XSPH: BUFFER_XSPH, written by every forces pass for the fluid particles, read by the Euler steps
        self.xsph = torch.zeros((A, 4), dtype=f32, device=dev) if (self.sp.simflags & D.ENABLE_XSPH) else None
        # SA_BOUNDARY: BUFFER_VERTICES (uint4), BUFFER_BOUNDELEMENTS, BUFFER_GRADGAMMA (float4) travel through the re-sort like
        # pos/vel; BUFFER_VERTPOS (3 x float2) is written by the list build
        if self.sa:
            self.vertices = up(arrs["vertices"].view(np.int32), i32, (A, 4)); self.vertices2 = torch.zeros_like(self.vertices)
            self.boundelements = up(arrs["boundelements"], f32, (A, 4)); self.boundelements2 = torch.zeros_like(self.boundelements)
            self.gradgamma = up(arrs["gradgamma"], f32, (A, 4)); self.gradgamma2 = torch.zeros_like(self.gradgamma)
            self.vertpos = [torch.zeros((A, 2), dtype=f32, device=dev) for _ in range(3)]
            # dynamic gamma (no ENABLE_GAMMA_QUADRATURE): BUFFER_CFL_GAMMA, per particle and, behind round_up(n, 4), per block
            self.sa_dynamic_gamma = not (self.sp.simflags & D.ENABLE_GAMMA_QUADRATURE)
            self.sa_density_sum = bool(self.sp.simflags & D.ENABLE_DENSITY_SUM)
            self.cfl_gamma = (torch.zeros(A + 4 + self.cfl.numel(), dtype=f32, device=dev) if self.sa_dynamic_gamma else None)
        # SA open boundaries (ENABLE_INLET_OUTLET): BUFFER_EULERVEL (double buffered) and BUFFER_NEXTID travel through the re-sort,
        # IOwaterdepth is one uint per open boundary, the particle count grows inside the allocation.  The density
        # summation form, a rebuild in every iteration (the reference's open-boundary problems set buildneibsfreq = 1: a released
        # particle exists for the lists from the next rebuild on).  See _sa_post_euler_io.
        self.io = self.sa and bool(self.sp.simflags & D.ENABLE_INLET_OUTLET)
        # SA bodies with prescribed motion: BUFFER_BOUNDELEMENTS is a state buffer (the Euler step turns the normals of the moving
        # segments and vertices), boundelements2 holds it for the state n* / n+1 (PredictorCorrectorIntegrator.cc:408-418)
        self.sa_moving = self.sa and bool(self.sp.simflags & D.ENABLE_MOVING_BODIES)
        if self.sa_moving and self.keps:
            raise NotImplementedError("SA bodies with prescribed motion: not together with k-epsilon")
        if self.io:
            if not self.sa_density_sum or self.keps or self.sp.buildneibsfreq != 1:
                raise NotImplementedError("open boundaries are built for the density summation form without k-epsilon, buildneibsfreq = 1")
            if not hasattr(problem, "impose_open_boundaries"):
                raise ValueError("a problem with ENABLE_INLET_OUTLET needs impose_open_boundaries (its imposeBoundaryConditionHost)")
            self.eulervel = torch.zeros((A, 4), dtype=f32, device=dev); self.eulervel2 = torch.zeros_like(self.eulervel)
            # initializeNextIDs (src/GPUSPH.cc:1256-1302): the vertices of open boundaries get totParticles, totParticles + 1, ...
            # in particle order; nobody else has one
            flags = arrs["info"][:, 0]
            openv = ((flags & 7) == D.PT_VERTEX) & ((flags & (D.FG_INLET | D.FG_OUTLET)) != 0)
            nid = np.full(len(flags), 0xFFFFFFFF, dtype=np.uint32)
            nid[openv] = len(flags) + np.arange(int(openv.sum()), dtype=np.uint32)
            self.num_open_vertices = int(openv.sum())
            self.next_ids = torch.full((A,), -1, dtype=i32, device=dev)
            self.next_ids[:n0] = torch.from_numpy(nid.view(np.int32)[mask]).to(dev)
            self.next_ids2 = torch.full((A,), -1, dtype=i32, device=dev)
            self.water_depth_on = bool(self.sp.simflags & D.ENABLE_WATER_DEPTH)
            self.iowaterdepth = torch.zeros(max(int(problem.num_open_boundaries), 1), dtype=i32, device=dev) if self.water_depth_on else None
            self.io_count = torch.zeros(1, dtype=i32, device=dev)         # newNumParticles of the vertex pass
            self.io_created = self.io_removed = 0
        # turbulence<KEPSILON>: BUFFER_TKE / EPSILON / TURBVISC / EULERVEL are particle properties (double buffered, re-sorted);
        # ProblemCore::init_keps and init_turbvisc (src/ProblemCore.cc:1623-1659) give the uniform initial state; BUFFER_DKDE and
        # BUFFER_CFL_KEPS are outputs of the forces passes
        if self.keps:
            k0, e0, nut0 = problem.init_keps()
            self.ke = dict(tke=torch.full((A,), k0, dtype=f32, device=dev), eps=torch.full((A,), e0, dtype=f32, device=dev),
                           turbvisc=torch.full((A,), nut0, dtype=f32, device=dev), eulervel=torch.zeros((A, 4), dtype=f32, device=dev))
            self.ke2 = {k: torch.zeros_like(v) for k, v in self.ke.items()}
            self.dkde = torch.zeros((A, 3), dtype=f32, device=dev)
            self.cfl_keps = torch.zeros_like(self.cfl)
        self.effvisc = torch.zeros(A, dtype=f32, device=dev) if self.effvisc_on else None      # BUFFER_EFFVISC
        # ENABLE_INTERNAL_ENERGY: BUFFER_INTERNAL_ENERGY (double buffered, re-sorted; starts from zero: init_internal_energy) and its rate
        self.energy_on = bool(self.sp.simflags & D.ENABLE_INTERNAL_ENERGY)
        if self.energy_on:
            self.energy = torch.zeros(A, dtype=f32, device=dev); self.energy2 = torch.zeros_like(self.energy)
            self.dedt = torch.zeros(A, dtype=f32, device=dev)
        # SPH_GRENIER: BUFFER_VOLUME (double buffered like pos/vel, travels through the re-sort) and BUFFER_SIGMA
        if self.grenier:
            self.vol = torch.zeros((A, 4), dtype=f32, device=dev); self.vol2 = torch.zeros_like(self.vol)
            self.sigma = torch.zeros(A, dtype=f32, device=dev)
            self.k.init_volume(self.vol, self.pos, self.vel, self.info, n0)      # GPUSPH.cc:495-496
        self.filters = []            # [(FilterType, frequency)]
        # bodies with prescribed motion: every rank runs the same host kinematics (the callback is a pure function of time)
        self.bodies = None
        if getattr(problem, "moving_bodies_callback", None) is not None and getattr(problem, "num_obstacle", 0):
            from .bodies import MovingBodies
            self.bodies = MovingBodies(problem, problem.rb_cg_global)
        self.t_host = 0.0

    # ------------------------------------------------------------------ exchange
    def _neighbours(self):
        if self.world > 1 and self.part.ring:
            return (self.rank - 1) % self.world, (self.rank + 1) % self.world
        left = self.rank - 1 if self.rank > 0 else None
        right = self.rank + 1 if self.rank < self.world - 1 else None
        return left, right

    def _exchange(self, tensors):
        """send my edge layers / receive the halo layers of every tensor in `tensors` (dim-0 ranges)"""
        left, right = self._neighbours()
        self.halo_bytes += self.transport.exchange(tensors, left, right, self.send_l, self.recv_l, self.send_r, self.recv_r)

    # ------------------------------------------------------------------ neighbour phase
    def build_neibs(self):
        K = self.k
        n = self.n_local
        self._rows_for = None
        if self.iterations == 0:
            K.fix_hash(self.hash, self.partindex, self.info, self.devmap, n)
        else:
            K.calc_hash(self.pos, self.hash, self.partindex, self.info, self.devmap, n)
        K.sort(self.hash, self.info, self.partindex, n)
        K.memset(self.cellStart, 0xFF)
        K.memset(self.cellEnd, 0xFF)
        K.reorder(self.segment_start if self.world > 1 else None, self.cellStart, self.cellEnd, self.pos2, self.vel2,
                  self.pos, self.vel, self.info, self.hash, self.partindex, n, self.new_num)
        self.pos, self.pos2 = self.pos2, self.pos
        self.vel, self.vel2 = self.vel2, self.vel
        if self.sa:      # the optional arrays of reorderDataAndFindCellStart (src/cuda/buildneibs.cu:263-311)
            for name in ("vertices", "boundelements", "gradgamma"):
                src, dst = getattr(self, name), getattr(self, name + "2")
                K.gather_rows(dst, src, self.partindex, n)
                setattr(self, name, dst); setattr(self, name + "2", src)
        if self.io:      # BUFFER_EULERVEL and BUFFER_NEXTID are particle state too
            for name in ("eulervel", "next_ids"):
                src, dst = getattr(self, name), getattr(self, name + "2")
                K.gather_rows(dst, src, self.partindex, n)
                setattr(self, name, dst); setattr(self, name + "2", src)
        if self.energy_on:
            K.gather_rows(self.energy2, self.energy, self.partindex, n)
            self.energy, self.energy2 = self.energy2, self.energy
        if self.keps:
            for name in self.ke:
                K.gather_rows(self.ke2[name], self.ke[name], self.partindex, n)
            self.ke, self.ke2 = self.ke2, self.ke
        if self.grenier:
            K.gather_rows(self.vol2, self.vol, self.partindex, n)
            self.vol, self.vol2 = self.vol2, self.vol
        if self.world == 1:
            if self.track_particle_count:
                before = self.n_local
                self.n_local = int(self.new_num.item()) & 0xFFFFFFFF
                if self.io:
                    self.io_removed += before - self.n_local
            self.n_int = self.n_local
            self.edge_start = self.n_int
        else:
            self._update_segments_and_halo()
        if self.clobber_neibslist:
            K.memset(self.neibslist, 0xFF)
        if self.sa:
            K.build_neibs_sa(self.neibslist, self.vertpos, self.pos, self.info, self.vertices, self.boundelements, self.hash,
                             self.cellStart, self.cellEnd, self.n_local, self.n_int)
            if self.world > 1:       # the vertex offsets of the halo segments (UPDATE_EXTERNAL of BUFFER_VERTPOS after BUILDNEIBS)
                self._exchange(self.vertpos)
        else:
            K.build_neibs(self.neibslist, self.pos, self.info, self.hash, self.cellStart, self.cellEnd, self.n_local, self.n_int)

    def _sa_post_euler(self, step):
        """INTEGRATE_GAMMA on the new positions (always from the gamma of step n), then the boundary conditions of the new
        state (PredictorCorrectorIntegrator.cc:661-684 and the post-step phases); every pass covers the internal particles and is
        followed by the UPDATE_EXTERNAL of what it wrote"""
        K, n, ni = self.k, self.n_local, self.n_int
        ext = (lambda ts: self._exchange(ts)) if self.world > 1 else (lambda ts: None)
        be_new = self.boundelements2 if self.sa_moving else self.boundelements      # the elements of the new state
        if self.sa_density_sum:
            # DENSITY_SUM [+ CALC_DENSITY_DIFFUSION + APPLY_DENSITY_DIFFUSION] (PredictorCorrectorIntegrator.cc:607-659): density and
            # gamma of the new state from the positions of step n and of the new state; BUFFER_FORCES is their scratch
            if self.sa_moving:       # ... and from the elements where they were and where they are; gamma of the vertices too
                K.sa_density_sum_moving(self.vel2, self.gradgamma2, self.forces, self.pos, self.pos2, self.vel, self.gradgamma,
                                        self.boundelements, be_new, self.vertpos, self.info, self.hash, self.cellStart, self.neibslist, n, ni)
            else:
                K.sa_density_sum(self.vel2, self.gradgamma2, self.forces, self.pos, self.pos2, self.vel, self.gradgamma, self.boundelements,
                                 self.vertpos, self.info, self.hash, self.cellStart, self.neibslist, n, ni)
            ext([self.vel2, self.gradgamma2])
            if self.sp.densitydiffusiontype == D.BREZZI:
                dt = float(self.d_dt.item()) * (0.5 if step == 1 else 1.0)     # dt_op on the host, as the reference's command has it
                K.sa_density_diffusion(self.forces, self.pos2, self.vel2, self.gradgamma2, self.info, self.hash, self.cellStart,
                                       self.neibslist, n, ni, float(np.float32(dt)))
                ext([self.vel2])
        else:
            K.sa_integrate_gamma(self.gradgamma2, self.gradgamma, self.pos2, be_new, self.vertpos, self.info, self.hash,
                                 self.cellStart, self.neibslist, n, ni)
            ext([self.gradgamma2])
        if self.keps:
            ke = self.ke2
            K.sa_segment_bc_keps(self.vel2, self.gradgamma2, ke, self.pos2, self.vertices, self.boundelements, self.info, self.hash,
                                 self.cellStart, self.neibslist, n, ni, step)
            ext([self.vel2, self.gradgamma2, ke["tke"], ke["eps"], ke["eulervel"]])
            K.sa_vertex_bc_keps(self.vel2, self.gradgamma2, ke, self.vertices, self.boundelements, self.pos2, self.info, self.hash,
                                self.cellStart, self.neibslist, n, ni, step)
            ext([self.vel2, ke["tke"], ke["eps"], ke["eulervel"]])
            return
        K.sa_segment_bc(self.vel2, self.gradgamma2, self.pos2, self.vertices, be_new, self.info, self.hash, self.cellStart,
                        self.neibslist, n, ni, step, D.SIMULATE)
        ext([self.vel2, self.gradgamma2])
        K.sa_vertex_bc(self.vel2, self.gradgamma2, self.pos2, self.info, self.hash, self.cellStart, self.neibslist, n, ni, step, D.SIMULATE)
        ext([self.vel2])

    def _sa_post_euler_io(self, step):
        """The post-Euler commands of a step with ENABLE_INLET_OUTLET (src/integrators/PredictorCorrectorIntegrator.cc:127-300,
        607-684): DENSITY_SUM [+ density diffusion] with the open boundaries' terms from the state of step n,
        IMPOSE_OPEN_BOUNDARY_CONDITION (the problem's values; it consumes and clears the water depth, the maximum over the devices),
        the segment conditions, in the last step FIND_OUTGOING_SEGMENT, the vertex conditions (vertex masses; in the last step
        the take-over of outgoing particles and the release of new ones: the particle count grows), in the last step
        DISABLE_OUTGOING_PARTS.  Every pass covers the internal particles and is followed by the UPDATE_EXTERNAL of what it wrote."""
        K, n, ni = self.k, self.n_local, self.n_int
        ext = (lambda ts: self._exchange(ts)) if self.world > 1 else (lambda ts: None)
        dt = float(np.float32(np.float32(self.d_dt.item()) * np.float32(0.5 if step == 1 else 1.0)))      # dt_op, on the host
        # with bodies that move as well (the option set of CompleteSaExample.cu:46): the elements of step n and of the new state in
        # the density summation, those of the new state in everything that follows (as without open boundaries, _sa_post_euler)
        be_new = self.boundelements2 if self.sa_moving else self.boundelements
        if self.sa_moving:
            K.sa_density_sum_io_moving(self.vel2, self.gradgamma2, self.forces, self.pos, self.pos2, self.vel, self.eulervel, self.gradgamma,
                                       self.boundelements, be_new, self.vertpos, self.info, self.hash, self.cellStart, self.neibslist,
                                       n, ni, dt)
        else:
            K.sa_density_sum_io(self.vel2, self.gradgamma2, self.forces, self.pos, self.pos2, self.vel, self.eulervel, self.gradgamma,
                                self.boundelements, self.vertpos, self.info, self.hash, self.cellStart, self.neibslist, n, ni, dt)
        ext([self.vel2, self.gradgamma2])
        if self.sp.densitydiffusiontype == D.BREZZI:
            K.sa_density_diffusion_io(self.forces, self.pos2, self.vel2, self.gradgamma2, be_new, self.vertpos, self.info,
                                      self.hash, self.cellStart, self.neibslist, n, ni, dt)
            ext([self.vel2])
        self.eulervel2[:n] = self.eulervel[:n]
        self._io_impose(self.pos2, self.vel2, self.eulervel2)
        K.sa_segment_bc_io(self.vel2, self.gradgamma2, self.eulervel2, self.pos2, self.vertices, be_new, self.info,
                           self.hash, self.cellStart, self.neibslist, n, ni, step)
        ext([self.vel2, self.gradgamma2, self.eulervel2])
        if step == 2:
            K.sa_find_outgoing_segment(self.pos2, self.vel2, self.vertices, self.gradgamma2, self.vertpos, be_new,
                                       self.info, self.hash, self.cellStart, self.neibslist, n, ni)
            ext([self.vertices, self.gradgamma2])      # the marks: a vertex takes over from the halo's particles too
        self._io_vertex_bc(self.pos2, self.vel2, self.gradgamma2, self.eulervel2, dt, step, be_new)
        if step == 2:
            K.sa_disable_outgoing_parts(self.pos2, self.vertices, self.info, self.n_local)

    def _io_impose(self, pos, vel, eulervel):
        """IMPOSE_OPEN_BOUNDARY_CONDITION on every particle this device holds (the values are a function of position, time and
        the water level); over several devices the level is the maximum of theirs first (DOWNLOAD_IOWATERDEPTH,
        FIND_MAX_IOWATERDEPTH, UPLOAD_IOWATERDEPTH: src/integrators/PredictorCorrectorIntegrator.cc:214-222, GPUSPH.cc:2206-2226)"""
        if self.world > 1 and self.iowaterdepth is not None:
            mine = (self.iowaterdepth.to(torch.int64) & 0xFFFFFFFF).cpu().tolist()
            best = [max(v for v, _ in self.transport.allgather_pair(m, 0, self.device)) for m in mine]
            self.iowaterdepth.copy_(torch.tensor(best, dtype=torch.int64).to(torch.int32).to(self.device)
                                    if max(best) < 2 ** 31 else
                                    torch.from_numpy(np.array(best, dtype=np.uint32).view(np.int32)).to(self.device))
        self.problem.impose_open_boundaries(pos, vel, eulervel, self.info, self.hash, self.iowaterdepth, self.time(), self.n_local)

    def _io_vertex_bc(self, pos, vel, ggam, eulervel, dt, step, boundelements=None):
        """SA_CALC_VERTEX_BOUNDARY_CONDITIONS with open boundaries: the pass writes the vertex masses (and the rows of released
        particles) into `pos`, in place as the reference has it; the new particle count comes back in a device word.  Released
        particles are appended behind everything this device holds (its halo included); they belong to it (the hash of the vertex
        that released them) and are sorted in at the next rebuild"""
        K, n, ni = self.k, self.n_local, self.n_int
        self.io_count[0] = n
        be = self.boundelements if boundelements is None else boundelements      # (the rows of released particles are written there)
        K.sa_vertex_bc_io(vel, pos, pos, ggam, eulervel, self.forces, self.vertices, be, self.vertpos,
                          self.info, self.hash, self.next_ids, self.io_count, self.cellStart, self.neibslist, n, ni, self.alloc,
                          dt, step, self.num_open_vertices)
        n2 = int(self.io_count.item()) & 0xFFFFFFFF
        if n2 > self.alloc:
            raise RuntimeError("open boundaries released more particles than the allocation holds (%d > %d)" % (n2, self.alloc))
        if self.world > 1:
            self._exchange([pos, vel, eulervel])      # vertex masses, densities and Eulerian velocities of the halo's vertices
        self.io_created += n2 - n
        self.n_local = n2
        if self.world == 1:
            self.n_int = self.edge_start = n2

    def calc_private(self):
        """CALC_PRIVATE of the post-processing engine (src/cuda/post_process.cu:575-640: Problem::calcPrivate fills BUFFER_PRIVATE
        with whatever the problem defines, from positions, velocities, info, the hash and the neighbour list): the problem's
        `calc_private(engine)` gets the engine (its tensors live on the device) and returns one row per particle"""
        fn = getattr(self.problem, "calc_private", None)
        if fn is None:
            raise NotImplementedError("CALC_PRIVATE: the problem defines no calc_private (the reference's default throws as well)")
        out = fn(self)
        if out.shape[0] != self.n_local:
            raise ValueError("calc_private: one row per particle held by this device expected")
        return out

    def open_boundary_flux(self):
        """FLUX_COMPUTATION of the post-processing engine (src/cuda/post_process.cu:485-570) on the current state: per open boundary
        the volume flux sum A_s (u_E . n_s) through its segments, positive into the domain.  -> float32 tensor [num_open_boundaries]"""
        if not self.io:
            raise ValueError("the flux through open boundaries needs ENABLE_INLET_OUTLET")
        nob = int(self.problem.num_open_boundaries)
        flux = torch.zeros(max(nob, 1), dtype=torch.float32, device=self.device)
        self.k.flux_computation(flux, self.info, self.eulervel, self.boundelements, self.n_local, nob)
        return flux[:nob]

    def sa_boundary_conditions(self, step, run_mode=D.SIMULATE):
        """initializeBoundaryConditionsSequence<SA_BOUNDARY> (src/integrators/PredictorCorrectorIntegrator.cc:117-290) without
        open boundaries: at initialisation (step 0) the vertex normals and gamma, then in every step the segment and the vertex
        boundary conditions, in place on the current state; internal particles, then UPDATE_EXTERNAL."""
        if not self.sa:
            raise ValueError("boundary conditions sequence of a problem without SA_BOUNDARY")
        K, n, ni = self.k, self.n_local, self.n_int
        ext = (lambda ts: self._exchange(ts)) if self.world > 1 else (lambda ts: None)
        if step == 0:
            K.sa_compute_vertex_normal(self.boundelements, self.vertices, self.info, self.hash, self.cellStart, self.neibslist, n, ni)
            ext([self.boundelements])
            K.sa_init_gamma(self.gradgamma2, self.gradgamma, self.pos, self.boundelements, self.vertpos, self.info, self.hash,
                            self.cellStart, self.neibslist, n, ni)
            self.gradgamma, self.gradgamma2 = self.gradgamma2, self.gradgamma
            ext([self.gradgamma])
        if self.io:
            # with open boundaries (:150-290): corner vertices, the initial masses of the open vertices, the imposed values, then
            # the two condition passes of step 0
            if step != 0 or run_mode != D.SIMULATE:
                raise NotImplementedError("open boundaries: only the initialisation sequence goes through sa_boundary_conditions")
            K.sa_identify_corner_vertices(self.pos, self.info, self.hash, self.vertices, self.cellStart, self.neibslist, n, ni)
            ext([self.info])
            K.sa_init_io_mass_vertex_count(self.forces, self.pos, self.vertices, self.hash, self.info, self.cellStart, self.neibslist, n, ni)
            ext([self.forces])      # a vertex divides by the counts of the vertices it shares segments with, the halo's among them
            K.sa_init_io_mass(self.pos2, self.pos, self.forces, self.vertices, self.hash, self.info, self.cellStart, self.neibslist, n, ni)
            self.pos2[ni:n] = self.pos[ni:n]      # (the pass writes the internal rows)
            self.pos, self.pos2 = self.pos2, self.pos
            ext([self.pos])
            self._io_impose(self.pos, self.vel, self.eulervel)
            K.sa_segment_bc_io(self.vel, self.gradgamma, self.eulervel, self.pos, self.vertices, self.boundelements, self.info,
                               self.hash, self.cellStart, self.neibslist, n, ni, 0)
            ext([self.vel, self.gradgamma, self.eulervel])
            self._io_vertex_bc(self.pos, self.vel, self.gradgamma, self.eulervel, 0.0, 0)
            return
        if self.keps and run_mode == D.SIMULATE:
            ke = self.ke
            K.sa_segment_bc_keps(self.vel, self.gradgamma, ke, self.pos, self.vertices, self.boundelements, self.info, self.hash,
                                 self.cellStart, self.neibslist, n, ni, step)
            ext([self.vel, self.gradgamma, ke["tke"], ke["eps"], ke["eulervel"]])
            K.sa_vertex_bc_keps(self.vel, self.gradgamma, ke, self.vertices, self.boundelements, self.pos, self.info, self.hash,
                                self.cellStart, self.neibslist, n, ni, step)
            ext([self.vel, ke["tke"], ke["eps"], ke["eulervel"]])
            return
        K.sa_segment_bc(self.vel, self.gradgamma, self.pos, self.vertices, self.boundelements, self.info, self.hash, self.cellStart,
                        self.neibslist, n, ni, step, run_mode)
        ext([self.vel, self.gradgamma])
        K.sa_vertex_bc(self.vel, self.gradgamma, self.pos, self.info, self.hash, self.cellStart, self.neibslist, n, ni, step, run_mode)
        ext([self.vel])

    def _update_segments_and_halo(self):
        """UPDATE_SEGMENTS + CROP + APPEND_EXTERNAL (src/Integrator.cc:170-230)"""
        K = self.k
        seg = [int(v) & 0xFFFFFFFF for v in self.segment_start.cpu().tolist()]       # DOWNLOAD (sync)
        newn = int(self.new_num.item()) & 0xFFFFFFFF
        starts = list(seg) + [newn]
        for i in range(3, -1, -1):                 # EMPTY_SEGMENT -> start of the next non-empty one
            if starts[i] == D.EMPTY_SEGMENT:
                starts[i] = starts[i + 1]
        edge_start, n_int = starts[1], starts[2]   # internal = inner + inner edge; the rest is cropped
        self.edge_start, self.n_int = edge_start, n_int
        left, right = self._neighbours()
        # my inner-edge segment is sorted by hash: the plane facing the left neighbour comes first
        if left is not None and right is not None:
            key = (D.CELLTYPE_INNER_EDGE_CELL << 30) | ((self.part.hi[self.rank] - 1) * self.part.plane)
            h = self.hash[edge_start:n_int].to(torch.int64) & 0xFFFFFFFF
            split = edge_start + int(torch.searchsorted(h, torch.tensor([key], dtype=torch.int64, device=h.device)).item())
        elif left is not None:
            split = n_int
        else:
            split = edge_start
        self.send_l = (edge_start, split) if left is not None else (0, 0)
        self.send_r = (split, n_int) if right is not None else (0, 0)
        # counts of the layers my neighbours send me
        allc = self.transport.allgather_pair(self.send_l[1] - self.send_l[0], self.send_r[1] - self.send_r[0], self.device)
        rl = allc[left][1] if left is not None else 0      # left neighbour's right layer
        rr = allc[right][0] if right is not None else 0    # right neighbour's left layer
        if n_int + rl + rr > self.alloc:
            raise RuntimeError("rank %d: %d internal + %d halo particles exceed the %d allocated"
                               % (self.rank, n_int, rl + rr, self.alloc))
        self.recv_l = (n_int, n_int + rl)
        self.recv_r = (n_int + rl, n_int + rl + rr)
        self.n_local = n_int + rl + rr
        state = [self.pos, self.vel, self.info, self.hash]
        if self.grenier:
            state.append(self.vol)           # BUFFER_VOLUME is particle state: the halo copies integrate theirs from the exchanged forces
        if self.energy_on:
            state.append(self.energy)
        if self.sa:      # BUFFER_VERTICES / BOUNDELEMENTS / GRADGAMMA are particle state too
            state += [self.vertices, self.boundelements, self.gradgamma]
        if self.io:      # ... and BUFFER_EULERVEL (the halo's vertices and segments enter the forces with theirs)
            state.append(self.eulervel)
        if self.keps:
            state += list(self.ke.values())
        self._exchange(state)
        # imported cells are OUTER_EDGE cells here whatever they are at home
        if self.n_local > n_int:
            hs = self.hash[n_int:self.n_local]
            hs.bitwise_and_(D.CELLTYPE_BITMASK).bitwise_or_(-2147483648)          # CELLTYPE_OUTER_EDGE_CELL << 30
        # cell ranges: forget everything outside my own planes, then index the fresh halo
        lo, hi, plane = self.part.lo[self.rank] * self.part.plane, self.part.hi[self.rank] * self.part.plane, self.part.plane
        for t in (self.cellStart, self.cellEnd):
            if lo > 0:
                K.memset(t[:lo], 0xFF)
            if hi < self.ncells:
                K.memset(t[hi:], 0xFF)
        K.find_cell_start(self.cellStart, self.cellEnd, self.hash, n_int, self.n_local)

    # ------------------------------------------------------------------ forces / euler
    def _forces_pass(self, pos, vel, combine_min, run_mode=D.SIMULATE, step=1):
        K = self.k
        # (with bodies: rows of body particles owned by other ranks must read zero in the reduction)
        K.memset(self._zeroed if self.has_rb else self.cfl, 0)
        prof = self.profile_forces is not None and self.is_cuda
        if prof:
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
        sps = self.sps and run_mode == D.SIMULATE
        if sps:   # CALC_VISC on the state the forces read (PredictorCorrectorIntegrator.cc:460-480)
            K.calc_visc(self.tau, pos, vel, self.info, self.hash, self.cellStart, self.neibslist, self.n_local, self.n_int,
                        turbvisc=self.turbvisc)
            if self.world > 1:
                self._exchange(self.tau)
        # the forces entry of this option set, as a function of the particle range and the offset into the CFL array
        if self.sa:      # forces engine of SA_BOUNDARY: the state's gamma, the boundary elements, the vertex offsets of the segments
            ggam = self.gradgamma if pos is self.pos else self.gradgamma2
            be = self.boundelements2 if (self.sa_moving and pos is not self.pos) else self.boundelements

            ke = (self.ke if pos is self.pos else self.ke2) if (self.keps and run_mode == D.SIMULATE) else None
            if ke is not None:
                K.memset(self.cfl_keps, 0)
            # bodies that feel the fluid (FG_COMPUTE_FORCE segments): the pressure force on their elements rides with the pass
            # (compute_boundary_pressure_force inside finalizeforcesDevice, src/cuda/forces_kernel.def:4115-4145)
            body_forces = self.has_rb and self.sp.numforcesbodies > 0 and run_mode == D.SIMULATE

            def launch(frm, to, off):
                nb = launch_sa(frm, to, off)
                if body_forces:
                    K.sa_body_pressure_forces(self.forces, self.rbforces, self.rbtorques, pos, vel, self.info, self.hash, be, frm, to)
                return nb

            def launch_sa(frm, to, off):
                if ke is not None:
                    return K.forces_sa_keps(self.forces, self.cfl, self.cfl_keps, self.dkde, pos, vel, self.info, self.hash, self.cellStart,
                                            self.neibslist, ggam, self.boundelements, self.vertpos, ke, self.n_local, frm, to, off,
                                            cfl_gamma=self.cfl_gamma)
                if self.io:      # the Eulerian velocity of the open boundaries in the viscous terms and in the gamma CFL condition
                    ev = self.eulervel if pos is self.pos else self.eulervel2
                    return K.forces_sa_io(self.forces, self.cfl, pos, vel, ev, self.info, self.hash, self.cellStart, self.neibslist, ggam,
                                          be, self.vertpos, self.n_local, frm, to, off, cfl_gamma=self.cfl_gamma)
                return K.forces_sa(self.forces, self.cfl, pos, vel, self.info, self.hash, self.cellStart, self.neibslist, ggam,
                                   be, self.vertpos, self.n_local, frm, to, off, cfl_gamma=self.cfl_gamma, run_mode=run_mode)
        elif self.effvisc_on and run_mode == D.SIMULATE:
            # CALC_VISC on the state the forces read (internal particles, then UPDATE_EXTERNAL); its largest kinematic viscosity
            # is the viscous limit of this pass's dt on this device (the dt of the step is the minimum over the devices)
            K.calc_effvisc(self.effvisc, pos, vel, self.info, self.hash, self.cellStart, self.neibslist, self.n_local, self.n_int)
            if self.world > 1:
                self._exchange([self.effvisc])

            def launch(frm, to, off):
                return K.forces_effvisc(self.forces, self.cfl, pos, vel, self.info, self.hash, self.cellStart, self.neibslist,
                                        self.effvisc, self.n_local, frm, to, off)
        elif self.grenier and run_mode == D.SIMULATE:
            # COMPUTE_DENSITY on the state the forces read, UPDATE_EXTERNAL of sigma and of the rewritten velocity buffer
            # (PredictorCorrectorIntegrator.cc:443-458), then the Grenier forces
            vol = self.vol if pos is self.pos else self.vol2
            K.compute_density(self.sigma, vel, pos, self.info, self.hash, vol, self.cellStart, self.neibslist, self.n_int)
            if self.world > 1:
                self._exchange([self.sigma, vel])

            def launch(frm, to, off):
                return K.forces_grenier(self.forces, self.cfl, pos, vel, self.info, self.hash, self.cellStart, self.neibslist,
                                        self.sigma, self.n_local, frm, to, off)
        else:
            args = (self.forces, self.cfl, self.rbforces if self.has_rb else None, self.rbtorques if self.has_rb else None, pos, vel,
                    self.info, self.hash, self.cellStart, self.neibslist, self.n_local)
            kw = dict(tau=self.tau) if sps else {}
            if self.xsph is not None:
                kw["xsph"] = self.xsph
            if run_mode != D.SIMULATE or step != 1:
                kw.update(run_mode=run_mode, step=step)
            vouch = [self._rows_follow and run_mode == D.SIMULATE and self._rows_for is not None and self._rows_for[0] is vel and
                     vel._version == self._rows_for[1]]

            def launch(frm, to, off):
                if vouch[0]:
                    K.eos_rows_current(vel, self.n_local)
                vouch[0] = self._rows_follow and run_mode == D.SIMULATE      # a second stripe of this pass reads the buffer the first one read
                return K.forces(*args, frm, to, off, **kw)
        energy = self.energy_on and run_mode == D.SIMULATE      # the BUFFER_INTERNAL_ENERGY_UPD output of the pass travels with the forces
        if energy:
            K.forces_internal_energy(self.dedt, pos, vel, self.info, self.hash, self.cellStart, self.neibslist, self.n_local, 0, self.n_int)
        outputs = [self.forces, self.dedt] if energy else [self.forces]
        if self.xsph is not None and run_mode == D.SIMULATE:      # BUFFER_XSPH is a POST_FORCES_UPDATE_BUFFER: the halo copies are moved with it
            outputs.append(self.xsph)
        keps = self.keps and run_mode == D.SIMULATE
        if keps:         # BUFFER_DKDE is a POST_FORCES_UPDATE_BUFFER: the halo copies integrate k and epsilon from the exchanged rates
            outputs.append(self.dkde)
        if self.world > 1 and self.n_int > self.edge_start:
            # edge stripe first, then the inner stripe while the edge forces travel
            nb1 = launch(self.edge_start, self.n_int, 0)
            if self.overlap:
                ev = torch.cuda.Event(); ev.record()
            nb2 = launch(0, self.edge_start, nb1)
            if prof:
                e1.record(); self.profile_forces.append((e0, e1))
            if self.overlap:
                with torch.cuda.stream(self.comm_stream):
                    self.comm_stream.wait_event(ev)
                    self._exchange(outputs)
                if self.exchange_events is not None:     # how long the compute stream sits waiting for the halo forces
                    b = torch.cuda.Event(enable_timing=True); a_ = torch.cuda.Event(enable_timing=True)
                    b.record()
                torch.cuda.current_stream(self.device).wait_stream(self.comm_stream)
                if self.exchange_events is not None:
                    a_.record(); self.exchange_events.append((b, a_))
            else:
                self._exchange(outputs)
        else:
            nb1 = launch(0, self.n_int, 0)
            nb2 = 0
            if prof:
                e1.record(); self.profile_forces.append((e0, e1))
            if self.world > 1:
                self._exchange(outputs)
        K.dtreduce(self.cfl, self.cfl_temp, nb1 + nb2, self.d_dt_next, combine_min)
        if keps:         # viscous limit with the largest eddy viscosity (src/cuda/forces.cu:585-598)
            K.dtreduce_keps(self.cfl_keps, nb1 + nb2, self.d_dt_next)
        if self.sa and self.sa_dynamic_gamma and run_mode == D.SIMULATE:     # the CFL condition of the gamma transport (src/cuda/forces.cu:576-585)
            K.dtreduce_gamma(self.cfl_gamma, self.n_local, nb1 + nb2, self.d_dt_next)
        if self.io and self.water_depth_on:      # the vertex pass of the forces (vertex_forces, src/cuda/forces.cu:676-686)
            K.sa_io_water_depth(self.iowaterdepth, pos, self.info, self.hash, self.cellStart, self.neibslist, self.n_local, 0, self.n_int)

    def step(self):
        K = self.k
        if self.iterations % self.sp.buildneibsfreq == 0:
            self.build_neibs()
            if self.sa and self.iterations == 0:     # initialisation step of the boundary conditions (not after a resume)
                self.sa_boundary_conditions(0)
        n = self.n_local
        if self.iterations > 0:      # FILTER phases: internal particles, then UPDATE_EXTERNAL of the velocity buffer
            for ftype, freq in self.filters:
                if self.iterations % freq == 0:
                    self._rows_for = None
                    K.filter(ftype, self.vel2, self.pos, self.vel, self.info, self.hash, self.cellStart, self.neibslist, n, self.n_int)
                    if self.world > 1:
                        self._exchange([self.vel2])
                    self.vel, self.vel2 = self.vel2, self.vel
        ekw = dict(xsph=self.xsph) if self.xsph is not None else {}
        if self.bodies is not None:
            dt_host = float(self.d_dt.item())       # the callback needs dt on the host: one synchronisation per step
        # predictor: forces(step n) -> n* = n + dt/2 f
        self._forces_pass(self.pos, self.vel, 0, step=1)
        if self.bodies is not None:                 # MOVE_BODIES + uploads (PredictorCorrectorIntegrator.cc:550-570)
            m = self.bodies.timestep(1, dt_host, self.t_host); K.set_body_motion(m, self.sp.numforcesbodies > 0)
        if self.grenier:
            K.euler_grenier(self.pos2, self.vel2, self.vol2, self.pos, self.vel, self.vol, self.info, self.hash, self.forces, n, self.d_dt, 0.5, 1)
        else:
            K.euler(self.pos2, self.vel2, self.pos, self.vel, self.info, self.hash, self.forces, n, self.d_dt, 0.5, 1, **ekw)
            self._rows_for = (self.vel2, self.vel2._version)
        if self.sa_moving:           # update_normals of the Euler step: BUFFER_BOUNDELEMENTS of n* from that of n
            K.sa_update_normals(self.boundelements2, self.boundelements, self.info, n, n)
        if self.energy_on:
            K.euler_internal_energy(self.energy2, self.energy, self.dedt, self.pos, self.info, n, self.d_dt, 0.5)
        if self.keps:
            K.euler_keps(self.ke2, self.ke, self.dkde, self.forces, self.pos, self.info, n, self.d_dt, 0.5)
        if self.io:
            self._sa_post_euler_io(1)
        elif self.sa:
            self._sa_post_euler(1)
        # corrector: forces(step n*) -> n+1 = n + dt f*   (written over n*, then renamed to n)
        self._forces_pass(self.pos2, self.vel2, 1, step=2)
        if self.bodies is not None:
            m = self.bodies.timestep(2, dt_host, self.t_host); K.set_body_motion(m, self.sp.numforcesbodies > 0)
        if self.grenier:     # the volumes of n* are overwritten by those of n+1, like pos and vel
            K.euler_grenier(self.pos2, self.vel2, self.vol2, self.pos, self.vel, self.vol, self.info, self.hash, self.forces, n, self.d_dt, 1.0, 2)
            self.vol, self.vol2 = self.vol2, self.vol
        else:
            K.euler(self.pos2, self.vel2, self.pos, self.vel, self.info, self.hash, self.forces, n, self.d_dt, 1.0, 2, **ekw)
            self._rows_for = (self.vel2, self.vel2._version)
        if self.sa_moving:           # ... of n+1 from that of n (the full step's rotation)
            K.sa_update_normals(self.boundelements2, self.boundelements, self.info, n, n)
        if self.energy_on:
            K.euler_internal_energy(self.energy2, self.energy, self.dedt, self.pos, self.info, n, self.d_dt, 1.0)
            self.energy, self.energy2 = self.energy2, self.energy
        if self.keps:
            K.euler_keps(self.ke2, self.ke, self.dkde, self.forces, self.pos, self.info, n, self.d_dt, 1.0)
        if self.io:
            self._sa_post_euler_io(2)        # (the particle count has grown by the released particles: n is stale from here on)
            self.gradgamma, self.gradgamma2 = self.gradgamma2, self.gradgamma
            self.eulervel, self.eulervel2 = self.eulervel2, self.eulervel
            if self.sa_moving:
                self.boundelements, self.boundelements2 = self.boundelements2, self.boundelements
        elif self.sa:
            self._sa_post_euler(2)
            if self.keps:
                self.ke, self.ke2 = self.ke2, self.ke
            self.gradgamma, self.gradgamma2 = self.gradgamma2, self.gradgamma
            if self.sa_moving:
                self.boundelements, self.boundelements2 = self.boundelements2, self.boundelements
        if self.bodies is not None:                 # EULER_UPLOAD_OBJECTS_CG in the post-corrector phase (:331-332)
            K.set_body_cg_integration(m)
            self._last_motion = m
            self.t_host += dt_host                  # the same double += float as on the device
        self.pos, self.pos2 = self.pos2, self.pos
        self.vel, self.vel2 = self.vel2, self.vel
        # TIME_STEP_EPILOGUE: t += dt ; dt = min(dt_pred, dt_corr), over all devices (GPUSPH.cc:650-657)
        K.time_advance(self.d_t, self.d_dt)
        if self.world > 1:
            self.transport.allreduce_min(self.d_dt_next)
        self.d_dt, self.d_dt_next = self.d_dt_next, self.d_dt
        self.iterations += 1

    def run(self, steps):
        for _ in range(steps):
            self.step()

    def time(self):
        return float(self.d_t.item())

    # ------------------------------------------------------------------ views
    @property
    def n(self):
        return self.n_int

    def internal_particles(self):
        return self.n_int

    def neibs_info(self):
        return self.k.neibs_info()

    def add_filter(self, filtertype, frequency):
        self.filters.append((int(filtertype), int(frequency)))

    def current_dt(self):
        return float(self.d_dt.item())

    def reduce_rb_forces(self):
        """REDUCE_BODIES_FORCES + REDUCE_BODIES_FORCES_HOST (src/GPUSPH.cc): total force and torque on the feedback body,
        per device over the rows of its own particles, then summed over the devices (one all_reduce of 6 floats).
        The order of the float sum is implementation-defined in the reference too (thrust scan): tolerance-only."""
        if not self.has_rb:
            return None
        tot = torch.cat([self.rbforces[:, :3].double().sum(0), self.rbtorques[:, :3].double().sum(0)])
        if self.world > 1:
            self.transport.allreduce_sum(tot)
        tot = tot.cpu().numpy()
        return tot[:3].astype(np.float32), tot[3:].astype(np.float32)

    def download_internal(self):
        n = self.n_int
        return {"pos": self.pos[:n].cpu().numpy(), "vel": self.vel[:n].cpu().numpy(),
                "info": self.info[:n].cpu().numpy().view(np.uint16), "hash": self.hash[:n].cpu().numpy().view(np.uint32),
                "forces": self.forces[:n].cpu().numpy(),
                **({"vol": self.vol[:n].cpu().numpy()} if self.grenier else {}),
                **({"energy": self.energy[:n].cpu().numpy()} if self.energy_on else {}),
                **({"gradgamma": self.gradgamma[:n].cpu().numpy(), "boundelements": self.boundelements[:n].cpu().numpy()} if self.sa else {}),
                **({k: v[:n].cpu().numpy() for k, v in self.ke.items()} if self.keps else {})}
