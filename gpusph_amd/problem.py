"""Problem-style host set-up of the synthetic "uniform dam-break box" (SURVEY.md 8d).

Mirrors, on the host and in numpy, the parts of GPUSPH's Problem API that decide what the
engines see (paths relative to the GPUSPH tree):
  DamBreak3D ctor            src/problems/DamBreak3D.cu:37-214  (geometry, options, EOS)
  ProblemCore::set_grid_params           src/ProblemCore.cc:1432-1508
  ProblemCore::calc_localpos_and_hash    src/ProblemCore.cc:1553-1583
  ProblemCore::check_neiblistsize        src/ProblemCore.cc:806-887
  ProblemCore::check_dt                  src/ProblemCore.cc:748-800
  ProblemAPI<1>::copy_to_array (ids)     src/problem_api/ProblemAPI_1.cc:1766-1817
The particle fill is a deterministic lattice (no RNG unless `jitter` is requested, then
numpy default_rng(12345)); it follows DamBreak3D's layout -- 1.6 x 0.67 x 0.6 box with 3 layers
of dynamic-boundary particles on the inside of every face, a 0.4 x (Ly-6dp) x 0.4 water column
3 dp away from the walls, optional 0.12 x 0.12 x 0.6 obstacle at x = 0.9 -- but it is NOT claimed
to reproduce GPUSPH's Cube::Fill particle-for-particle.
"""
from dataclasses import dataclass
import math
import numpy as np
from . import defs as D
from .params import SimParams, PhysParams, check_neiblistsize, make_sphx_params

INFO_DTYPE = np.dtype([("x", "<u2"), ("y", "<u2"), ("z", "<u2"), ("w", "<u2")])


def make_particleinfo(ptype_flags, obj_fluid, ids):
    """particleinfo = ushort4 {type|flags, fluid<<12|object, id lo, id hi} (src/particleinfo.h:50-79)."""
    n = len(ids)
    info = np.zeros((n, 4), dtype=np.uint16)
    info[:, 0] = ptype_flags
    info[:, 1] = obj_fluid
    ids = np.asarray(ids, dtype=np.uint32)
    info[:, 2] = ids & 0xFFFF
    info[:, 3] = ids >> 16
    return info


def info_id(info):
    return info[:, 2].astype(np.uint32) | (info[:, 3].astype(np.uint32) << 16)


def info_type(info):
    return info[:, 0] & 7


def _lattice(i0, i1, j0, j1, k0, k1):
    """integer lattice indices of the closed box [i0,i1]x[j0,j1]x[k0,k1] as (n,3) int32."""
    if i1 < i0 or j1 < j0 or k1 < k0:
        return np.zeros((0, 3), dtype=np.int32)
    ii, jj, kk = np.meshgrid(np.arange(i0, i1 + 1, dtype=np.int32),
                             np.arange(j0, j1 + 1, dtype=np.int32),
                             np.arange(k0, k1 + 1, dtype=np.int32), indexing="ij")
    return np.stack([ii.ravel(), jj.ravel(), kk.ravel()], axis=1)


@dataclass
class HostParticles:
    pos_global: np.ndarray   # (n,4) float64 : x,y,z,mass
    vel: np.ndarray          # (n,4) float32 : vx,vy,vz,rho_tilde
    info: np.ndarray         # (n,4) uint16


class Problem:
    """The slice of ProblemCore the hot path depends on."""

    def __init__(self):
        self.simparams = SimParams()
        self.physparams = PhysParams()
        self.m_deltap = 0.0
        self.m_origin = np.zeros(3)
        self.m_size = np.zeros(3)
        self.m_gridsize = np.zeros(3, dtype=np.int64)
        self.m_cellsize = np.zeros(3)
        self.linearization = D.DEFAULT_LINEARIZATION
        self.m_name = "Problem"
        self.parts = None
        self.planes = []            # [(unit normal (3), reference point (3), global coordinates)], PlaneList
        self.moving_bodies_callback = None   # ProblemCore::moving_bodies_callback: f(index, t0, t1, initial_kdata, kdata) -> (dx, dr)
        self.m_maxFall = float("nan")        # ProblemAPI<1>::setMaxFall

    def plane_tables(self):
        """plane_t arrays as ProblemCore::copy_planes hands them to setplanes: unit normal, grid cell and
        cell-local position of the reference point (src/planes.h:43-59)."""
        n = len(self.planes)
        normals = np.zeros((n, 3), dtype=np.float32)
        gridpos = np.zeros((n, 3), dtype=np.int32)
        pos = np.zeros((n, 3), dtype=np.float32)
        for k, (nrm, pt) in enumerate(self.planes):
            nrm = np.asarray(nrm, dtype=np.float64)
            normals[k] = (nrm / np.linalg.norm(nrm)).astype(np.float32)
            g = self.calc_grid_pos(np.asarray(pt, dtype=np.float64)[None, :])[0]
            gridpos[k] = g
            pos[k] = (np.asarray(pt, dtype=np.float64) - self.m_origin - (g + 0.5) * self.m_cellsize).astype(np.float32)
        return normals, gridpos, pos

    # -- ProblemCore::set_deltap / set_smoothing --
    def set_deltap(self, dp):
        self.m_deltap = float(np.float32(dp))
        self.simparams.set_smoothing(self.simparams.sfactor, self.m_deltap)

    # -- ProblemCore::set_grid_params (src/ProblemCore.cc:1432-1508) --
    def set_grid_params(self):
        sp = self.simparams
        if sp.nlInfluenceRadius < sp.influenceRadius:
            raise RuntimeError("neighbor search radius < kernel influence radius")
        cellSide = sp.nlInfluenceRadius
        self.m_gridsize = np.floor(self.m_size / cellSide).astype(np.int64)
        if (self.m_gridsize == 0).any():
            raise RuntimeError("resolution %g is too low! Resulting grid size would be %s" % (sp.slength, self.m_gridsize))
        self.m_cellsize = self.m_size / self.m_gridsize
        if int(np.prod(self.m_gridsize)) > D.MAX_CELLS:
            raise RuntimeError("too many cells")

    # -- ProblemCore::check_dt (src/ProblemCore.cc:748-800), inviscid branch --
    def check_dt(self):
        sp, pp = self.simparams, self.physparams
        f32 = np.float32
        dt_ss = min(f32(sp.slength) / f32(c) for c in pp.sscoeff) * f32(sp.dtadaptfactor)
        g = math.sqrt(sum(x * x for x in pp.gravity))
        dt_g = f32(math.sqrt(sp.slength / g)) * f32(sp.dtadaptfactor) if g > 0 else np.inf
        cfl_dt = float(min(dt_ss, dt_g))
        if sp.rheologytype != D.INVISCID:      # get_dt_from_visc<NEWTONIAN>, ProblemCore.cc:728-745,767-786
            nu = max(pp.kinematicvisc)
            cfl_dt = float(min(f32(cfl_dt), f32(sp.slength) * f32(sp.slength) / f32(nu) * f32(0.125)))
        if not sp.dt:
            sp.dt = cfl_dt
        return sp.dt

    # -- ProblemCore::calc_grid_pos / calc_localpos_and_hash (src/ProblemCore.cc:1513-1583) --
    def calc_grid_pos(self, pos):
        g = np.floor((pos[:, :3] - self.m_origin) / self.m_cellsize).astype(np.int64)
        return np.clip(g, 0, self.m_gridsize - 1)

    def calc_grid_hash(self, g):
        c1, c2, c3 = D.LINEARIZATIONS[self.linearization]
        gs = self.m_gridsize
        return ((g[:, c3] * gs[c2]) * gs[c1] + g[:, c2] * gs[c1] + g[:, c1]).astype(np.uint32)

    def calc_localpos_and_hash(self, pos_global):
        g = self.calc_grid_pos(pos_global)
        h = self.calc_grid_hash(g)
        local = np.empty((len(pos_global), 4), dtype=np.float32)
        local[:, :3] = (pos_global[:, :3] - self.m_origin - (g + 0.5) * self.m_cellsize).astype(np.float32)
        local[:, 3] = pos_global[:, 3].astype(np.float32)
        return local, h

    def global_pos(self, local_pos, hashes):
        """inverse of calc_localpos_and_hash, in float64 (for analysis / tests)."""
        g = self.grid_pos_from_hash(hashes)
        return self.m_origin + (g + 0.5) * self.m_cellsize + local_pos[:, :3].astype(np.float64)

    def grid_pos_from_hash(self, hashes):
        c1, c2, c3 = D.LINEARIZATIONS[self.linearization]
        gs = self.m_gridsize
        h = (np.asarray(hashes, dtype=np.int64) & D.CELLTYPE_BITMASK)
        g = np.empty((len(h), 3), dtype=np.int64)
        t = gs[c2] * gs[c1]
        g[:, c3] = h // t
        rem = h - g[:, c3] * t
        g[:, c2] = rem // gs[c1]
        g[:, c1] = rem - g[:, c2] * gs[c1]
        return g

    def initialize(self):
        """ProblemCore::initialize order: grid, neighbour list size, defaults, dt."""
        sp, pp = self.simparams, self.physparams
        self.set_grid_params()
        if math.isnan(pp.r0):
            pp.r0 = self.m_deltap
        check_neiblistsize(sp, pp, self.m_deltap)
        if not math.isfinite(pp.dcoeff) and math.isfinite(self.m_maxFall):   # ProblemAPI<1>::initialize, ProblemAPI_1.cc:322-326
            g = float(np.float32(math.sqrt(sum(x * x for x in pp.gravity))))
            pp.dcoeff = float(np.float32(np.float32(5.0) * np.float32(g) * self.m_maxFall))
        if math.isnan(pp.epsartvisc):   # ProblemCore.cc:160-163
            pp.epsartvisc = float(np.float32(0.01 * sp.slength * sp.slength))
        if sp.sph_formulation == D.SPH_GRENIER and math.isnan(pp.epsinterface):   # ProblemCore.cc:165-166
            pp.epsinterface = 0.05
        if sp.densitydiffusiontype == D.COLAGROSSI:  # ProblemCore.cc:1406-1416
            if math.isnan(sp.densityDiffCoeff):
                sp.densityDiffCoeff = float(np.float32(0.1))
            sp.densityDiffCoeff = float(np.float32(np.float32(sp.densityDiffCoeff) * np.float32(2.0) * np.float32(sp.slength)))
        if sp.boundarytype == D.MK_BOUNDARY:         # ProblemCore.cc:141-154
            if math.isnan(pp.MK_d):
                pp.MK_d = float(np.float32(1.1 * self.m_deltap / pp.MK_beta))
            if math.isnan(pp.MK_K):
                pp.MK_K = float(np.float32(math.sqrt(sum(x * x for x in pp.gravity))))
        if sp.densitydiffusiontype == D.FERRARI:     # ProblemCore.cc:1379-1396
            if math.isnan(sp.densityDiffCoeff):
                ls = getattr(sp, "ferrariLengthScale", float("nan"))
                sp.densityDiffCoeff = 0.0 if math.isnan(ls) else float(np.float32(np.float32(ls) * np.float32(1e-3) / np.float32(self.m_deltap)))
        if sp.turbmodel == D.SPS:        # GPUSPH.cc:1540-1556: (Cs dp)^2 and (2/3) Ci dp^2, in double
            dp = self.m_deltap
            if math.isnan(pp.smagfactor):
                x = float(np.float32(np.float64(np.float32(pp.smagorinsky_constant)) * dp))
                pp.smagfactor = float(np.float32(x) * np.float32(x))
            if math.isnan(pp.kspsfactor):
                pp.kspsfactor = float(np.float32((2 * np.float64(np.float32(pp.isotropic_sps_constant)) / 3) * dp * dp))
        self.check_dt()

    def sphx_params(self, allocated):
        return make_sphx_params(self.simparams, self.physparams, gridsize=self.m_gridsize,
                                cellsize=self.m_cellsize, origin=self.m_origin, deltap=self.m_deltap,
                                allocated=allocated, linearization=self.linearization)

    def set_viscosity(self, spec):
        """viscosity<...> selector of the framework (src/visc_spec.h:314-391): a legacy name ("ARTVISC",
        "KINEMATICVISC", "DYNAMICVISC", "SPSVISC") or a dict(rheologytype, turbmodel, compvisc, viscmodel, avgop,
        is_const_visc) of FullViscSpec parameters; None = ARTVISC."""
        sp = self.simparams
        legacy = {
            None: dict(rheologytype=D.INVISCID, turbmodel=D.ARTIFICIAL),
            "ARTVISC": dict(rheologytype=D.INVISCID, turbmodel=D.ARTIFICIAL),
            "KINEMATICVISC": dict(rheologytype=D.NEWTONIAN, turbmodel=D.LAMINAR_FLOW, avgop=D.HARMONIC, is_const_visc=True),
            "DYNAMICVISC": dict(rheologytype=D.NEWTONIAN, turbmodel=D.LAMINAR_FLOW),
            # constant viscosity only by the single-fluid rule: the framework looks at the legacy NAME, and that is not
            # KINEMATICVISC (src/cuda/cudasimframework.cu:123-128); make_sphx_params applies the rule for None
            "SPSVISC": dict(rheologytype=D.NEWTONIAN, turbmodel=D.SPS, avgop=D.HARMONIC),
        }
        opts = dict(compvisc=D.KINEMATIC, viscmodel=D.MORRIS, avgop=D.ARITHMETIC, is_const_visc=None)
        opts.update(legacy[spec] if (spec is None or isinstance(spec, str)) else spec)
        for k, v in opts.items():
            setattr(sp, k, v)
        self.physparams.rheologytype = sp.rheologytype      # PhysParams(rheologytype), physparams.h:380

    def init_keps(self):
        """ProblemCore::init_keps + init_turbvisc (src/ProblemCore.cc:1623-1659): the uniform initial k, epsilon and eddy
        viscosity of turbulence<KEPSILON>, with the reference's mix of float and double arithmetic.  -> (k0, e0, nu_t0)"""
        f = np.float32
        Lm = max(2.0 * float(self.m_deltap), float(f(1e-5)))                           # fmax(2*m_deltap, 1e-5f), double
        k0 = f(float(f(0.002) * f(self.physparams.sscoeff[0])) ** 2)                 # pow(float, int) is double
        e0 = f(float(f(0.16) * np.power(k0, f(1.5))) / Lm)                           # 0.16f*powf(k0, 1.5f) / double
        nut0 = f(0.9 * float(k0) * float(k0) / float(e0))                            # 0.9*ki*ki/ei in double
        return float(k0), float(e0), float(nut0)

    def initial_density(self, pos_global):
        """rho~ the problem starts from at the given global positions; also used to reset the state at the end of a
        repacking run (ProblemCore::resetBuffers, src/ProblemCore.cc:1773-1796)."""
        return np.zeros(len(pos_global), dtype=np.float32)

    def copy_to_array(self):
        """host arrays as GPUSPH uploads them: cell-local float4 pos, float4 vel, ushort4 info, hash."""
        local, h = self.calc_localpos_and_hash(self.parts.pos_global)
        return {"pos": local, "vel": self.parts.vel.copy(), "info": self.parts.info.copy(), "hash": h}

    @property
    def num_particles(self):
        return len(self.parts.info)

    @property
    def grid_cells(self):
        return int(np.prod(self.m_gridsize))


class DamBreak3D(Problem):
    """Synthetic DamBreak3D (src/problems/DamBreak3D.cu:37-214): WENDLAND, SPH_F1, COLAGROSSI,
    ARTVISC (INVISCID + ARTIFICIAL), DYN_BOUNDARY, no periodicity, ENABLE_DTADAPT|ENABLE_REPACKING,
    3 dynamic-boundary layers, 128-slot neighbour list, c0 = 20, gamma = 7, rho0 = 1000, xi = 0.1."""

    DIM = (1.6, 0.67, 0.6)
    WATER_LENGTH = 0.4
    H = 0.4
    OBSTACLE_SIDE = 0.12
    OBSTACLE_XPOS = 0.9
    LAYERS = 3

    def __init__(self, deltap=0.015, *, obstacle=True, density_diffusion=D.COLAGROSSI, hydrostatic=True,
                 jitter=0.0, linearization=D.DEFAULT_LINEARIZATION, kerneltype=D.WENDLAND, boundary=D.DYN_BOUNDARY,
                 walls="particles", testpoints=(), two_fluids=False, viscosity=None, kinematic_visc=1.0e-2,
                 formulation=D.SPH_F1, dem=False, internal_energy=False):
        super().__init__()
        self.m_name = "DamBreak3D"
        sp, pp = self.simparams, self.physparams
        sp.kerneltype = kerneltype
        if kerneltype == D.GAUSSIAN:
            sp.kernelradius = 3.0
        if boundary not in (D.DYN_BOUNDARY, D.LJ_BOUNDARY, D.MK_BOUNDARY):
            raise ValueError("DamBreak3D mirror: DYN_BOUNDARY, LJ_BOUNDARY or MK_BOUNDARY")
        sp.boundarytype = boundary
        if walls not in ("particles", "planes") or (walls == "planes" and boundary == D.DYN_BOUNDARY):
            raise ValueError("walls: 'particles', or 'planes' with LJ_BOUNDARY (m_usePlanes)")
        self.walls = walls
        if boundary in (D.LJ_BOUNDARY, D.MK_BOUNDARY):
            # one layer of repulsive particles on the box faces instead of three dynamic layers
            # (DamBreak3D.cu:74,131-134: layers only for DYN_BOUNDARY)
            self.LAYERS = 1
        self.set_viscosity(viscosity)       # DamBreak3D.cu:54 viscosity<ARTVISC>
        sp.sph_formulation = formulation
        if formulation == D.SPH_GRENIER and (viscosity is None or isinstance(viscosity, str)):
            sp.avgop = D.HARMONIC           # legacy viscosity names average harmonically with Grenier (cudasimframework.cu:202-210)
        sp.densitydiffusiontype = density_diffusion
        sp.simflags = D.ENABLE_DTADAPT | D.ENABLE_REPACKING | \
            (D.ENABLE_PLANES if walls == "planes" else 0) | (D.ENABLE_MULTIFLUID if two_fluids else 0) | (D.ENABLE_DEM if dem else 0) | \
            (D.ENABLE_INTERNAL_ENERGY if internal_energy else 0)      # AccuracyTest.cu: add_flags<ENABLE_INTERNAL_ENERGY>
        # ENABLE_DEM (DEMExample.cu's option set: LJ_BOUNDARY + DEM + side planes): a synthetic terrain instead of the floor
        if dem and not (boundary == D.LJ_BOUNDARY and walls == "planes"):
            raise ValueError("dem=True: LJ_BOUNDARY with walls='planes' (addDEM + addDEMPlanes)")
        self.use_dem = bool(dem)
        self.dem = None
        sp.neiblistsize = 128               # resize_neiblist(128), DamBreak3D.cu:76
        if kerneltype == D.GAUSSIAN:
            sp.neiblistsize = 384           # radius 3h: ~250 neighbours in the bulk
        sp.densityDiffCoeff = 0.1           # DamBreak3D.cu:95
        self.linearization = linearization
        self.set_deltap(deltap)
        pp.gravity = (0.0, 0.0, -9.81)
        if boundary in (D.LJ_BOUNDARY, D.MK_BOUNDARY):
            # ProblemCore::check_dt/initialize defaults: r0 = deltap, D = 5|g| (src/ProblemCore.cc:132-150)
            pp.r0 = self.m_deltap
            pp.dcoeff = 5.0 * 9.81
        pp.add_fluid(1000.0)
        pp.set_equation_of_state(0, 7.0, 20.0)
        pp.set_kinematic_visc(0, kinematic_visc)            # DamBreak3D.cu:89
        self.two_fluids = bool(two_fluids)
        if two_fluids:      # a lighter fluid on top of the water column (multi-fluid branch of the forces engine)
            pp.add_fluid(850.0)
            pp.set_equation_of_state(1, 7.0, 22.0)
            pp.set_kinematic_visc(1, 3.0 * kinematic_visc)
        self.m_origin = np.zeros(3)
        self.m_size = np.array(self.DIM, dtype=np.float64)
        self.obstacle = obstacle
        self.testpoints = np.asarray(list(testpoints), dtype=np.float64).reshape(-1, 3)   # add_testpoint()
        self.hydrostatic = hydrostatic
        self.jitter = jitter
        if obstacle:
            sp.numbodies = 1
            sp.numforcesbodies = 1
        self.initialize()
        self.fill_parts()

    def _make_dem(self, fluid):
        """a smooth synthetic terrain on a node-centred grid (TopoCube: ewres = sizex/(ncols-1), world origin at DEM (0, 0)) and
        the parameters computeDEMphysparams derives from it (src/problem_api/ProblemAPI_1.cc:1399-1418: displacement scale 5,
        zmin scale 5); the water column is lifted above the highest point"""
        pp = self.physparams
        L = self.m_size
        ncols, nrows = 33, 15
        ewres, nsres = L[0] / (ncols - 1), L[1] / (nrows - 1)
        x = np.arange(ncols) * ewres
        y = np.arange(nrows) * nsres
        self.dem = (0.04 + 0.03 * np.sin(2 * np.pi * x[None, :] / 0.9) * np.cos(2 * np.pi * y[:, None] / 0.5)).astype(np.float32)
        pp.ewres = float(np.float32(ewres)); pp.nsres = float(np.float32(nsres))
        pp.demdx = float(np.float32(pp.ewres) / np.float32(5.0)); pp.demdy = float(np.float32(pp.nsres) / np.float32(5.0))
        pp.demzmin = float(np.float32(5.0 * self.m_deltap))
        fluid = fluid.copy()
        fluid[:, 2] += float(self.dem.max())
        return fluid

    # analytic particle count for a given dp (used to hit a target N)
    @classmethod
    def count(cls, dp, obstacle=True):
        n = [int(round(L / dp)) for L in cls.DIM]
        total = (n[0] + 1) * (n[1] + 1) * (n[2] + 1)
        inner = max(n[0] + 1 - 2 * cls.LAYERS, 0) * max(n[1] + 1 - 2 * cls.LAYERS, 0) * max(n[2] + 1 - 2 * cls.LAYERS, 0)
        walls = total - inner
        bd = cls.LAYERS * dp
        fl = [int(round(L / dp)) + 1 for L in (cls.WATER_LENGTH - bd, cls.DIM[1] - 2 * bd, cls.H - bd)]
        fluid = fl[0] * fl[1] * fl[2]
        obst = 0
        if obstacle:
            m = int(round(cls.OBSTACLE_SIDE / dp)) + 1
            mz = n[2] + 1 - 2 * cls.LAYERS
            obst = (m * m - max(m - 2 * cls.LAYERS, 0) ** 2) * mz
        return fluid + walls + obst

    @classmethod
    def deltap_for(cls, target, obstacle=True):
        lo, hi = 1e-4, 0.2
        for _ in range(80):
            mid = math.sqrt(lo * hi)
            if cls.count(mid, obstacle) > target:
                lo = mid
            else:
                hi = mid
        dp = float(np.float32(hi))
        while cls.count(dp, obstacle) > target:     # float32 rounding of dp may tip a lattice count over
            dp = float(np.nextafter(np.float32(dp), np.float32(1.0)))
            dp *= 1.0005
            dp = float(np.float32(dp))
        return dp

    def initial_density(self, pos_global):
        if not self.hydrostatic:
            return np.zeros(len(pos_global), dtype=np.float32)
        # rho~ from the hydrostatic pressure under the initial free surface (inverse Tait EOS)
        rho0 = self.physparams.rho0[0]
        B = self.physparams.bcoeff[0]
        gam = self.physparams.gammacoeff[0]
        g = -self.physparams.gravity[2]
        depth = np.clip(self.H - pos_global[:, 2], 0.0, None)
        in_col = pos_global[:, 0] <= self.WATER_LENGTH + 0.5 * self.m_deltap
        depth = np.where(in_col, depth, 0.0)
        return (np.power(1.0 + rho0 * g * depth / B, 1.0 / gam) - 1.0).astype(np.float32)

    def fill_parts(self):
        dp = self.m_deltap
        L = self.m_size
        n = [int(round(L[a] / dp)) for a in range(3)]
        dx = [L[a] / n[a] for a in range(3)]
        Lr = self.LAYERS
        # --- walls: three layers on the inside of each face (DYN_BOUNDARY, FillIn with -layers) ---
        slabs = [
            _lattice(0, n[0], 0, n[1], 0, Lr - 1), _lattice(0, n[0], 0, n[1], n[2] - Lr + 1, n[2]),
            _lattice(0, n[0], 0, Lr - 1, Lr, n[2] - Lr), _lattice(0, n[0], n[1] - Lr + 1, n[1], Lr, n[2] - Lr),
            _lattice(0, Lr - 1, Lr, n[1] - Lr, Lr, n[2] - Lr), _lattice(n[0] - Lr + 1, n[0], Lr, n[1] - Lr, Lr, n[2] - Lr),
        ]
        wall_idx = np.concatenate(slabs, axis=0)
        wall = wall_idx.astype(np.float64) * np.array(dx)
        if self.walls == "planes":
            # geometric planes instead of boundary particles: floor and the four side walls, normals pointing inwards
            wall = np.zeros((0, 3))
            self.planes = [((0, 0, 1), (0, 0, 0)), ((1, 0, 0), (0, 0, 0)), ((-1, 0, 0), (L[0], 0, 0)),
                           ((0, 1, 0), (0, 0, 0)), ((0, -1, 0), (0, L[1], 0))]
            if self.use_dem:      # addDEMPlanes: the four sides only, the terrain is the floor
                self.planes = self.planes[1:]
        # --- water column (DamBreak3D.cu:139-145) ---
        bd = Lr * dp
        fsize = np.array([self.WATER_LENGTH - bd, L[1] - 2 * bd, self.H - bd])
        fn = [max(int(round(fsize[a] / dp)), 1) for a in range(3)]
        fdx = fsize / np.array(fn)
        fluid = _lattice(0, fn[0], 0, fn[1], 0, fn[2]).astype(np.float64) * fdx + bd
        if self.jitter:
            rng = np.random.default_rng(12345)
            fluid = fluid + rng.uniform(-self.jitter * dp, self.jitter * dp, size=fluid.shape)
        if self.use_dem:
            fluid = self._make_dem(fluid)
        # --- obstacle: axis-aligned 3-layer shell standing on the floor layers (DamBreak3D.cu:160-178) ---
        obst = np.zeros((0, 3))
        if self.obstacle:
            m = max(int(round(self.OBSTACLE_SIDE / dp)), 1)
            odx = self.OBSTACLE_SIDE / m
            o = _lattice(0, m, 0, m, 0, n[2] - 2 * Lr)
            shell = ((o[:, 0] < Lr) | (o[:, 0] > m - Lr) | (o[:, 1] < Lr) | (o[:, 1] > m - Lr))
            o = o[shell].astype(np.float64)
            ox0 = self.OBSTACLE_XPOS
            oy0 = L[1] / 2 - self.OBSTACLE_SIDE / 2
            obst = np.stack([ox0 + o[:, 0] * odx, oy0 + o[:, 1] * odx, (Lr + o[:, 2]) * dx[2]], axis=1)
        nf, nw, no = len(fluid), len(wall), len(obst)
        nt = len(self.testpoints)
        ntot = nf + nw + no + nt
        pos = np.empty((ntot, 4), dtype=np.float64)
        pos[:nf, :3] = fluid
        pos[nf:nf + nw, :3] = wall
        pos[nf + nw:nf + nw + no, :3] = obst
        pos[nf + nw + no:, :3] = self.testpoints     # appended last here (the reference numbers them first)
        rho0 = self.physparams.rho0[0]
        pos[:, 3] = rho0 * dp ** 3           # mass = rho0 dp^3
        vel = np.zeros((ntot, 4), dtype=np.float32)
        vel[:, 3] = self.initial_density(pos)
        # ids: sequential, fluid first then boundary then bodies (ProblemAPI_1.cc:1766-1817)
        ids = np.arange(ntot, dtype=np.uint32)
        tf = np.empty(ntot, dtype=np.uint16)
        tf[:nf] = D.PT_FLUID
        tf[nf:nf + nw] = D.PT_BOUNDARY
        tf[nf + nw:nf + nw + no] = D.PT_BOUNDARY | D.FG_MOVING_BOUNDARY | D.FG_COMPUTE_FORCE
        tf[nf + nw + no:] = D.PT_TESTPOINT
        objfl = np.zeros(ntot, dtype=np.uint16)   # fluid number 0; object number 0 for the obstacle
        if self.two_fluids:
            upper = np.zeros(ntot, dtype=bool)
            upper[:nf] = pos[:nf, 2] > 0.5 * self.H
            objfl[upper] = 1 << 12                  # fluid number lives in the high 4 bits (src/particleinfo.h:144-161)
            pos[upper, 3] = self.physparams.rho0[1] * dp ** 3
        info = make_particleinfo(tf, objfl, ids)
        self.parts = HostParticles(pos, vel, info)
        self.num_fluid, self.num_wall, self.num_obstacle = nf, nw, no
        # rigid body bookkeeping (GPUSPH.cc: s_hRbFirstIndex = -(first id of the body), rbcg)
        self.rb_firstindex = np.array([-(nf + nw)], dtype=np.int32) if no else np.zeros(0, dtype=np.int32)
        if no:
            cg = np.array([[self.OBSTACLE_XPOS + self.OBSTACLE_SIDE / 2, L[1] / 2, L[2] / 2, 0.0]])
            self.rb_cg_global = cg[:, :3].copy()
            g = self.calc_grid_pos(cg)
            self.rb_cg_gridpos = g.astype(np.int32)
            self.rb_cg_pos = (cg[:, :3] - self.m_origin - (g + 0.5) * self.m_cellsize).astype(np.float32)
        else:
            self.rb_cg_gridpos = np.zeros((0, 3), dtype=np.int32)
            self.rb_cg_pos = np.zeros((0, 3), dtype=np.float32)


class PeriodicBox(Problem):
    """A box of fluid with periodic faces and no walls: the smallest problem that exercises the periodic branches of
    the hot path (clampGridPos / calcGridHashPeriodic, src/cuda/buildneibs_kernel.cu:225-298, cellgrid.cuh:163-196).
    The lattice wraps seamlessly (box side = n dp), so with jitter = 0 every particle sees the same neighbourhood."""

    def __init__(self, deltap=0.05, n=(12, 10, 9), periodic=D.PERIODIC_X | D.PERIODIC_Y | D.PERIODIC_Z, jitter=0.1, velocity=(0.0, 0.0, 0.0),
                 gravity=(0.0, 0.0, 0.0), linearization=D.DEFAULT_LINEARIZATION, kerneltype=D.WENDLAND,
                 density_diffusion=D.COLAGROSSI, repacking=False, viscosity=None, kinematic_visc=1.0e-2):
        super().__init__()
        self.m_name = "PeriodicBox"
        sp, pp = self.simparams, self.physparams
        sp.kerneltype = kerneltype
        sp.boundarytype = D.DYN_BOUNDARY
        self.set_viscosity(viscosity)
        sp.densitydiffusiontype = density_diffusion
        sp.periodicbound = periodic
        sp.simflags = D.ENABLE_DTADAPT | (D.ENABLE_REPACKING if repacking else 0)
        sp.neiblistsize = 128
        sp.densityDiffCoeff = 0.1
        self.linearization = linearization
        self.set_deltap(deltap)
        pp.gravity = tuple(float(g) for g in gravity)
        pp.add_fluid(1000.0)
        pp.set_equation_of_state(0, 7.0, 20.0)
        pp.set_kinematic_visc(0, kinematic_visc)
        self.m_origin = np.zeros(3)
        self.m_size = np.array(n, dtype=np.float64) * self.m_deltap
        self.initialize()
        idx = _lattice(0, n[0] - 1, 0, n[1] - 1, 0, n[2] - 1).astype(np.float64)
        pos3 = (idx + 0.5) * self.m_deltap
        if jitter:
            rng = np.random.default_rng(4321)
            pos3 = pos3 + rng.uniform(-jitter * self.m_deltap, jitter * self.m_deltap, size=pos3.shape)
            pos3 = np.mod(pos3, self.m_size)
        ntot = len(pos3)
        pos = np.empty((ntot, 4), dtype=np.float64)
        pos[:, :3] = pos3
        pos[:, 3] = pp.rho0[0] * self.m_deltap ** 3
        vel = np.zeros((ntot, 4), dtype=np.float32)
        vel[:, :3] = np.asarray(velocity, dtype=np.float32)
        info = make_particleinfo(np.full(ntot, D.PT_FLUID, dtype=np.uint16), np.zeros(ntot, dtype=np.uint16),
                                 np.arange(ntot, dtype=np.uint32))
        self.parts = HostParticles(pos, vel, info)
        self.num_fluid, self.num_wall, self.num_obstacle = ntot, 0, 0
        self.rb_firstindex = np.zeros(0, dtype=np.int32)
        self.rb_cg_gridpos = np.zeros((0, 3), dtype=np.int32)
        self.rb_cg_pos = np.zeros((0, 3), dtype=np.float32)


class Poiseuille(Problem):
    """Mirror of src/problems/Poiseuille.inc (the problem behind the reference's own analytic validator,
    scripts/validate-poiseuille.py): a unit cube of fluid between two DYN_BOUNDARY walls at z = +-lz/2, periodic in x and
    y, driven by a body force along x; Newtonian rheology, laminar flow, MORRIS, computational viscosity and averaging
    operator selectable as the validator does (--compvisc kin|dyn, --viscavg arithmetic|harmonic|geometric), Wendland
    kernel, no density diffusion by default, c0 = 20 max(sqrt(2 F lz), u_max).  Geometry as the problem places it
    (Poiseuille.inc:115-131): wall planes of (influence layers + 1) particle layers starting at z = +-lz/2 and growing
    outward, fluid lattice from -lz/2 + dp to lz/2 - dp, ppH particles per height.  The steady state is
    u(z) = F/(2 nu) (lz^2/4 - z^2) (compute_poiseuille_vel)."""

    def __init__(self, ppH=16, *, compvisc=D.KINEMATIC, viscavg=D.HARMONIC, rho=1.0, kinvisc=0.1, driving_force=0.05,
                 density_diffusion=D.DENSITY_DIFFUSION_NONE, steady_init=False, linearization=D.DEFAULT_LINEARIZATION,
                 rheology=D.NEWTONIAN, power_law_n=None, exponential_coeff=None, regularization=None, viscmodel=D.MORRIS,
                 bulk_visc=None):
        super().__init__()
        self.m_name = "Poiseuille"
        self.lz = self.ly = self.lx = 1.0
        self.rho, self.kinvisc, self.driving_force = float(rho), float(kinvisc), float(driving_force)
        sp, pp = self.simparams, self.physparams
        sp.kerneltype = D.WENDLAND
        sp.boundarytype = D.DYN_BOUNDARY
        # POISEUILLE_RHEOLOGY: NEWTONIAN (Poiseuille.cu) or a generalized Newtonian one (PoiseuillePapanastasiou.cu: PAPANASTASIOU)
        self.rheology = rheology
        yielding = rheology > D.NEWTONIAN and rheology not in (D.POWER_LAW, D.GRANULAR)        # YIELDING_RHEOLOGY
        self.ys = float(np.float32(np.float32(driving_force) * np.float32(rho) * np.float32(self.lz) / np.float32(4))) if yielding else 0.0
        self.set_viscosity(dict(rheologytype=rheology, turbmodel=D.LAMINAR_FLOW, compvisc=compvisc, viscmodel=viscmodel,
                                avgop=viscavg))
        sp.densitydiffusiontype = density_diffusion
        sp.periodicbound = D.PERIODIC_X | D.PERIODIC_Y
        sp.simflags = D.ENABLE_DTADAPT
        self.linearization = linearization
        self.set_deltap(self.lz / ppH)
        dp = self.m_deltap
        pp.gravity = (self.driving_force, 0.0, 0.0)
        pp.add_fluid(self.rho)
        pp.set_kinematic_visc(0, self.kinvisc)
        if bulk_visc is not None:
            pp.set_bulk_visc(0, bulk_visc)
        if yielding:
            pp.set_yield_strength(0, self.ys)             # Poiseuille.inc:131-132
        if power_law_n is not None:
            pp.set_visc_power_law(0, power_law_n)
        if exponential_coeff is not None:
            pp.set_visc_exponential_coeff(0, exponential_coeff)
        if regularization is not None:
            pp.set_visc_regularization_param(0, regularization)
        self.max_vel = self.compute_poiseuille_vel(0.0)
        hydrostatic_vel = math.sqrt(2.0 * self.driving_force * self.lz)
        pp.set_equation_of_state(0, 7.0, 20.0 * max(hydrostatic_vel, self.max_vel))
        self.dyn_layers = int(math.ceil(sp.sfactor * sp.kernelradius)) + 1      # suggestedDynamicBoundaryLayers
        # world box (ProblemAPI<1>::initialize, ProblemAPI_1.cc:255-300): bounding box of the geometries + dp/2, + the
        # extra boundary layers in the non-periodic direction
        half = np.array([(self.lx - dp) / 2, (self.ly - dp) / 2, self.lz / 2])
        gmin = -half - dp / 2
        gmax = half + dp / 2
        gmin[2] -= (self.dyn_layers - 1) * dp
        gmax[2] += (self.dyn_layers - 1) * dp
        self.m_origin = gmin
        self.m_size = gmax - gmin
        self.m_maxFall = float(self.lz + (self.dyn_layers - 1) * dp + dp / 2)   # water level - lowest point (autocomputed)
        self.initialize()
        # particles: lattice centred on the origin in x and y; fluid rows k = 1..ppH-1 at z = -lz/2 + k dp
        nxy = int(round((self.lx - dp) / dp)) + 1
        x0 = -(self.lx - dp) / 2
        ij = _lattice(0, nxy - 1, 0, nxy - 1, 0, 0)[:, :2].astype(np.float64) * dp + x0
        nz_f = ppH - 1
        fl = np.concatenate([np.column_stack([ij, np.full(len(ij), -self.lz / 2 + k * dp)]) for k in range(1, nz_f + 1)])
        lo = np.concatenate([np.column_stack([ij, np.full(len(ij), -self.lz / 2 - k * dp)]) for k in range(self.dyn_layers)])
        hi = np.concatenate([np.column_stack([ij, np.full(len(ij), self.lz / 2 + k * dp)]) for k in range(self.dyn_layers)])
        pos3 = np.concatenate([fl, lo, hi])
        nf, nw = len(fl), len(lo) + len(hi)
        ntot = nf + nw
        pos = np.empty((ntot, 4), dtype=np.float64)
        pos[:, :3] = pos3
        pos[:, 3] = pp.rho0[0] * dp ** 3
        vel = np.zeros((ntot, 4), dtype=np.float32)
        if steady_init:      # --steady-init: fluid starts from the analytic profile (Poiseuille.inc:136-150)
            vel[:nf, 0] = [self.compute_poiseuille_vel(z) for z in pos3[:nf, 2]]
        ptype = np.concatenate([np.full(nf, D.PT_FLUID, dtype=np.uint16), np.full(nw, D.PT_BOUNDARY, dtype=np.uint16)])
        info = make_particleinfo(ptype, np.zeros(ntot, dtype=np.uint16), np.arange(ntot, dtype=np.uint32))
        self.parts = HostParticles(pos, vel, info)
        self.num_fluid, self.num_wall, self.num_obstacle = nf, nw, 0
        self.rb_firstindex = np.zeros(0, dtype=np.int32)
        self.rb_cg_gridpos = np.zeros((0, 3), dtype=np.int32)
        self.rb_cg_pos = np.zeros((0, 3), dtype=np.float32)

    def compute_poiseuille_vel(self, z):
        """Poiseuille::compute_poiseuille_vel for n = 1 (Newtonian, or Bingham-like with a plug of half-width
        ys/(rho F)): Poiseuille.inc:187-229, in float like the reference (the reference evaluates the sheared branch with
        z - plug, i.e. for z >= 0; |z| here); scripts/validate-poiseuille.py:33-38 is the Newtonian formula in double"""
        f32 = np.float32
        plug = f32(f32(self.ys) / f32(f32(self.rho) * f32(self.driving_force))) if self.ys else f32(0.0)
        A = f32(f32(self.driving_force) / f32(self.kinvisc))
        A = f32(f32(1.0) * A) / f32(2.0)
        B = f32(f32(self.lz) / f32(2.0) - plug); B = f32(B * B)
        if abs(z) > self.lz / 2:
            return 0.0
        if abs(z) < plug:
            return float(f32(A * B))
        C = f32(f32(abs(z)) - plug); C = f32(C * C)
        return float(f32(A * f32(B - C)))


class WaveTank(Problem):
    """Mirror of src/problems/WaveTank.cu (BASELINE configs[4]'s option set): a 9 x 0.6 x 1 m flume with a hinged paddle
    (a moving body with prescribed rotation about y), a sloping beach as a geometric plane, Lennard-Jones box particles
    plus five wall planes, viscosity<SPSVISC>, a Shepard filter every 20 iterations (the caller adds it:
    `engine.add_filter(SHEPARD_FILTER, 20)`), dtadaptfactor 0.2.  The particle layout follows the problem's own loops
    (fluid rows between paddle and beach, WaveTank.cu:158-168); box and paddle surfaces are dp lattices."""
    LX, LY, LZ = 9.0, 0.6, 1.0
    SLOPE_LENGTH, H_LENGTH, HEIGHT, H = 8.5, 0.5, 0.63, 0.45
    BETA = 4.2364 * math.pi / 180.0

    def __init__(self, deltap=0.03, *, paddle_tstart=0.5, linearization=D.DEFAULT_LINEARIZATION, viscosity="SPSVISC"):
        super().__init__()
        self.m_name = "WaveTank"
        sp, pp = self.simparams, self.physparams
        sp.kerneltype = D.WENDLAND
        sp.boundarytype = D.LJ_BOUNDARY
        self.set_viscosity(viscosity)                      # WaveTank.cu:58
        sp.densitydiffusiontype = D.DENSITY_DIFFUSION_NONE
        sp.simflags = D.ENABLE_DTADAPT | D.ENABLE_PLANES      # WaveTank.cu:55-62 (ENABLE_MOVING_BODIES only matters to SA_BOUNDARY)
        sp.dtadaptfactor = 0.2
        sp.dt = 1.0e-4                                     # set_timestep(0.0001)
        self.linearization = linearization
        self.set_deltap(deltap)
        pp.gravity = (0.0, 0.0, -9.81)
        r0 = self.m_deltap
        pp.r0 = r0
        pp.dcoeff = 5.0 * 9.81 * self.H                    # setMaxFall(H): D = 5 g H (ProblemAPI_1.cc)
        pp.add_fluid(1000.0)
        pp.set_equation_of_state(0, 7.0, 20.0)
        pp.set_kinematic_visc(0, 1.0e-6)
        pp.artvisccoeff = 0.2
        self.m_origin = np.zeros(3)
        self.m_size = np.array([self.LX, self.LY, self.LZ], dtype=np.float64)
        self.paddle_origin = np.array([0.25, r0, 0.0])
        self.paddle_length = 0.7
        self.paddle_tstart, self.paddle_tend = float(paddle_tstart), 30.0
        self.paddle_amplitude = math.atan(0.2 / (2.0 * (self.H - self.paddle_origin[2])))
        self.paddle_omega = 2.0 * math.pi / 0.8
        sp.numbodies = 1
        sp.numforcesbodies = 0
        sb, cb = math.sin(self.BETA), math.cos(self.BETA)
        L = self.H_LENGTH + self.SLOPE_LENGTH
        # copy_planes (WaveTank.cu:245-258): unit normal n and a point on n.x + d = 0
        self.planes = [((0, 0, 1), (0, 0, 0)), ((0, 1, 0), (0, 0, 0)), ((0, -1, 0), (0, self.LY, 0)), ((1, 0, 0), (0, 0, 0)),
                       ((-1, 0, 0), (L, 0, 0)), ((-sb, 0, cb), (self.H_LENGTH, 0, 0))]
        self.moving_bodies_callback = self._paddle
        self.initialize()
        self.fill_parts()

    # WaveTank::moving_bodies_callback (WaveTank.cu:221-243): hinge rotation about y, first-order quaternion step
    def _paddle(self, index, t0, t1, kd0, kd):
        kd.lvel = np.zeros(3)
        if self.paddle_tstart < t1 < self.paddle_tend:
            w = self.paddle_amplitude * self.paddle_omega * math.sin(self.paddle_omega * (t1 - self.paddle_tstart))
            kd.avel = np.array([0.0, w, 0.0])
            # dr = normalize(1 + (t1 - t0)/2 (0, avel)): a rotation about y by 2 atan(w dt / 2)
            e0, e2 = 1.0, 0.5 * (t1 - t0) * w
            nrm = math.hypot(e0, e2)
            e0, e2 = e0 / nrm, e2 / nrm
            c, s = e0 * e0 - e2 * e2, 2.0 * e0 * e2          # cos, sin of the step angle
            return np.zeros(3), np.array([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]])
        kd.avel = np.zeros(3)
        return np.zeros(3), np.eye(3)

    def fill_parts(self):
        dp = self.m_deltap
        r0 = dp
        amp = -self.paddle_amplitude
        po = self.paddle_origin
        # fluid rows (WaveTank.cu:158-168)
        rows = []
        z, n = 0.0, 0
        while z < self.H:
            z = n * dp + 1.5 * r0
            x = po[0] + (z - po[2]) * math.tan(amp) + 1.0 * r0 / math.cos(amp)
            l = self.H_LENGTH + z / math.tan(self.BETA) - 1.5 * r0 / math.sin(self.BETA) - x
            nx, ny = int(l / dp) + 1, int((self.LY - 2.0 * r0) / dp) + 1
            ix, iy = np.meshgrid(np.arange(nx), np.arange(ny), indexing="ij")
            rows.append(np.stack([x + ix.ravel() * dp, r0 + iy.ravel() * dp, np.full(ix.size, z)], axis=1))
            n += 1
        fluid = np.concatenate(rows)
        # experiment box (FT_BORDER): dp lattice on the six faces of [0, h_length + slope_length] x [0, ly] x [0, height]
        Lx = self.H_LENGTH + self.SLOPE_LENGTH
        nb = [int(round(Lx / dp)), int(round(self.LY / dp)), int(round(self.HEIGHT / dp))]
        g = _lattice(0, nb[0], 0, nb[1], 0, nb[2])
        surf = (g[:, 0] == 0) | (g[:, 0] == nb[0]) | (g[:, 1] == 0) | (g[:, 1] == nb[1]) | (g[:, 2] == 0) | (g[:, 2] == nb[2])
        wall = g[surf].astype(np.float64) * np.array([Lx / nb[0], self.LY / nb[1], self.HEIGHT / nb[2]])
        # paddle: a plate of width ly - 2 r0 and length 0.7 from the hinge, tilted by -amplitude about y
        npy, npz = int(round((self.LY - 2 * r0) / dp)), int(round(self.paddle_length / dp))
        jy, jz = np.meshgrid(np.arange(npy + 1), np.arange(npz + 1), indexing="ij")
        s = jz.ravel() * (self.paddle_length / npz)
        padd = np.stack([po[0] + s * math.sin(-amp) * (-1.0), po[1] + jy.ravel() * ((self.LY - 2 * r0) / npy), po[2] + s * math.cos(amp)], axis=1)
        nf, nw, no = len(fluid), len(wall), len(padd)
        ntot = nf + nw + no
        pos = np.empty((ntot, 4), dtype=np.float64)
        pos[:nf, :3] = fluid; pos[nf:nf + nw, :3] = wall; pos[nf + nw:, :3] = padd
        pos[:, :3] = np.clip(pos[:, :3], 1e-9, self.m_size - 1e-9)
        pos[:, 3] = self.physparams.rho0[0] * dp ** 3
        vel = np.zeros((ntot, 4), dtype=np.float32)
        vel[:, 3] = self.initial_density(pos)
        tf = np.empty(ntot, dtype=np.uint16)
        tf[:nf] = D.PT_FLUID
        tf[nf:nf + nw] = D.PT_BOUNDARY
        tf[nf + nw:] = D.PT_BOUNDARY | D.FG_MOVING_BOUNDARY
        info = make_particleinfo(tf, np.zeros(ntot, dtype=np.uint16), np.arange(ntot, dtype=np.uint32))
        self.parts = HostParticles(pos, vel, info)
        self.num_fluid, self.num_wall, self.num_obstacle = nf, nw, no
        self.rb_firstindex = np.zeros(1, dtype=np.int32)
        cg = self.paddle_origin[None, :].copy()
        self.rb_cg_global = cg.copy()
        gcell = self.calc_grid_pos(cg)
        self.rb_cg_gridpos = gcell.astype(np.int32)
        self.rb_cg_pos = (cg - self.m_origin - (gcell + 0.5) * self.m_cellsize).astype(np.float32)

    def initial_density(self, pos_global):
        # hydrostatic filling under the still-water level H (m_hydrostaticFilling, m_waterLevel)
        pp = self.physparams
        depth = np.clip(self.H - pos_global[:, 2], 0.0, None)
        return (np.power(1.0 + pp.rho0[0] * 9.81 * depth / pp.bcoeff[0], 1.0 / pp.gammacoeff[0]) - 1.0).astype(np.float32)


class StillWater(Problem):
    """Mirror of src/problems/StillWater.cu (the problem of BASELINE configs[2], with the boundary models built here): a
    sqrt(2)H x sqrt(2)H x 1.1H box of still water of depth H = 1, viscosity<DYNAMICVISC> (nu = 3e-2), DYN_BOUNDARY walls of
    ceil(kernelradius sfactor) particle layers (or one layer plus five planes with use_planes), Ferrari density diffusion
    with length scale H by default (--density-diffusion), deltap = H/ppH, c0 = ceil(10 sqrt(2 g H)), dt0 = 4e-5,
    dtadaptfactor 0.3, neighbour rebuild every 20 iterations, optional MLS filter (the caller adds it, like --mls N).
    StillWater.cu:50-143 for the parameters and the box / fluid extents."""
    H = 1.0

    def __init__(self, ppH=16, *, use_planes=False, density_diffusion=D.FERRARI, viscosity="DYNAMICVISC",
                 linearization=D.DEFAULT_LINEARIZATION, jitter=0.0):
        super().__init__()
        self.m_name = "StillWater"
        sp, pp = self.simparams, self.physparams
        sp.kerneltype = D.WENDLAND
        sp.boundarytype = D.DYN_BOUNDARY
        self.set_viscosity(viscosity)
        sp.densitydiffusiontype = density_diffusion
        sp.simflags = D.ENABLE_DTADAPT | (D.ENABLE_PLANES if use_planes else 0)
        sp.dtadaptfactor = 0.3
        sp.buildneibsfreq = 20
        sp.dt = 4.0e-5
        sp.ferrariLengthScale = self.H
        self.linearization = linearization
        self.use_planes = bool(use_planes)
        self.jitter = jitter
        self.set_deltap(self.H / ppH)
        dp = self.m_deltap
        l = w = math.sqrt(2.0) * self.H
        h = 1.1 * self.H
        self.l, self.w, self.h = l, w, h
        self.m_origin = np.zeros(3)
        self.m_size = np.array([l, w, h], dtype=np.float64)
        if not use_planes:   # the box grows by the extra boundary layers (StillWater.cu:82-87)
            self.dyn_layers = int(math.ceil(sp.kernelradius * sp.sfactor))
            extra = (self.dyn_layers - 1) * dp
            self.m_origin = self.m_origin - extra
            self.m_size = self.m_size + 2 * extra
        else:
            self.dyn_layers = 1
        g = 9.81
        pp.gravity = (0.0, 0.0, -g)
        self.m_maxFall = self.H                             # setMaxFall(H), StillWater.cu:70
        c0 = math.ceil(10.0 * math.sqrt(2.0 * g * self.H))
        pp.add_fluid(1000.0)
        pp.set_equation_of_state(0, 7.0, float(c0))
        pp.set_kinematic_visc(0, 3.0e-2)
        if use_planes:   # copy_planes (StillWater.cu:145-154): floor and the four side walls
            o = self.m_origin
            self.planes = [((0, 0, 1), (0, 0, o[2])), ((0, 1, 0), (0, o[0], 0)), ((0, -1, 0), (0, o[0] + w, 0)),
                           ((1, 0, 0), (o[1], 0, 0)), ((-1, 0, 0), (o[1] + l, 0, 0))]
        self.initialize()
        self.fill_parts()

    def fill_parts(self):
        dp = self.m_deltap
        Lr = self.dyn_layers
        L = self.m_size
        n = [int(round(L[a] / dp)) for a in range(3)]
        dx = np.array([L[a] / n[a] for a in range(3)])
        # box walls (FT_BORDER filled inwards with dyn_layers layers): lattice nodes within Lr layers of a face
        g = _lattice(0, n[0], 0, n[1], 0, n[2])
        near = np.zeros(len(g), dtype=bool)
        for a in range(3):
            near |= (g[:, a] < Lr) | (g[:, a] > n[a] - Lr)
        wall = g[near].astype(np.float64) * dx + self.m_origin
        # fluid box (StillWater.cu:125-136)
        wd = dp
        fo = self.m_origin.copy()
        if Lr > 1:
            fo = fo + Lr * dp
        fo = fo + wd
        shift = 2 * wd if Lr == 1 else (Lr - 1) * dp * 2
        fsize = np.array([self.l - shift, self.w - shift, self.H - shift])
        fn = [max(int(round(fsize[a] / dp)), 1) for a in range(3)]
        fluid = _lattice(0, fn[0], 0, fn[1], 0, fn[2]).astype(np.float64) * (fsize / np.array(fn)) + fo
        if self.jitter:
            rng = np.random.default_rng(4242)
            fluid = fluid + rng.uniform(-self.jitter * dp, self.jitter * dp, size=fluid.shape)
        self.water_level = float(fo[2] + fsize[2])
        nf, nw = len(fluid), len(wall)
        ntot = nf + nw
        pos = np.empty((ntot, 4), dtype=np.float64)
        pos[:nf, :3] = fluid
        pos[nf:, :3] = wall
        eps = 1e-9
        pos[:, :3] = np.clip(pos[:, :3], self.m_origin + eps, self.m_origin + self.m_size - eps)
        pos[:, 3] = self.physparams.rho0[0] * dp ** 3
        vel = np.zeros((ntot, 4), dtype=np.float32)
        vel[:, 3] = self.initial_density(pos)
        tf = np.empty(ntot, dtype=np.uint16)
        tf[:nf] = D.PT_FLUID
        tf[nf:] = D.PT_BOUNDARY
        info = make_particleinfo(tf, np.zeros(ntot, dtype=np.uint16), np.arange(ntot, dtype=np.uint32))
        self.parts = HostParticles(pos, vel, info)
        self.num_fluid, self.num_wall, self.num_obstacle = nf, nw, 0
        self.rb_firstindex = np.zeros(0, dtype=np.int32)
        self.rb_cg_gridpos = np.zeros((0, 3), dtype=np.int32)
        self.rb_cg_pos = np.zeros((0, 3), dtype=np.float32)

    def initial_density(self, pos_global):
        # hydrostatic filling below the still-water level (m_hydrostaticFilling, ProblemAPI_1.cc)
        pp = self.physparams
        depth = np.clip(self.water_level - pos_global[:, 2], 0.0, None)
        return (np.power(1.0 + pp.rho0[0] * 9.81 * depth / pp.bcoeff[0], 1.0 / pp.gammacoeff[0]) - 1.0).astype(np.float32)

    @classmethod
    def ppH_for(cls, target):
        """ppH whose particle count is closest to `target` from below (fluid ~ 2 ppH^3, plus the walls)"""
        p = max(int((target / 2.0) ** (1.0 / 3.0)), 4)
        while p > 4 and cls.count(p) > target:
            p -= 1
        return p

    @classmethod
    def count(cls, ppH):
        dp = cls.H / ppH
        Lr = 3
        ext = 2 * (Lr - 1) * dp
        n = [int(round((math.sqrt(2.0) * cls.H + ext) / dp)), int(round((math.sqrt(2.0) * cls.H + ext) / dp)), int(round((1.1 * cls.H + ext) / dp))]
        full = (n[0] + 1) * (n[1] + 1) * (n[2] + 1)
        inner = max(n[0] + 1 - 2 * Lr, 0) * max(n[1] + 1 - 2 * Lr, 0) * max(n[2] + 1 - 2 * Lr, 0)
        shift = (Lr - 1) * dp * 2
        fs = [math.sqrt(2.0) * cls.H - shift, math.sqrt(2.0) * cls.H - shift, cls.H - shift]
        fl = 1
        for a in range(3):
            fl *= max(int(round(fs[a] / dp)), 1) + 1
        return full - inner + fl


class SABox(Problem):
    """A tank with SEMI-ANALYTICAL walls (SA_BOUNDARY), the synthetic counterpart of the reference's Crixus-meshed SA problems
    (e.g. src/problems/CompleteSaExample.cu; the reference reads such geometry from HDF5 files, src/HDF5SphReader.cc): the floor
    and the four side walls of an l x w x h tank are meshed by vertex particles (PT_VERTEX) on a square lattice of pitch deltap,
    every lattice square is cut into two triangular boundary elements (PT_BOUNDARY, placed at the centroid), and water of depth
    H fills the tank on the same lattice, one deltap off the walls (a vertex particle carries the half, quarter or eighth
    cell next to the wall, as in Crixus meshes).  Per particle, next to pos/vel/info (src/define_buffers.h:168-196):
      vertices       uint4   ids of the three vertices of a segment (0 elsewhere)
      boundelements  float4  unit normal towards the fluid and area of a segment (vertices: filled by computeVertexNormal)
      gradgamma      float4  (grad gamma, gamma); NaN until the boundary-conditions engine initialises it
    Framework options, list size, deltap and smoothing are StillWaterSA's (src/problems/StillWaterSA.cu:38-57) or, with
    options="StillWaterRepackSA", those of that problem's simulation (src/problems/StillWaterRepackSA.cu:38-44: continuity
    equation instead of density summation, gamma by quadrature, no density diffusion) -- the set the SA forces /
    integration engines are built for.  Wendland kernel (the only one the reference's SA code supports, src/cuda/gamma.cuh:241-250)."""

    def __init__(self, deltap=0.05, *, l=0.6, w=0.5, h=0.5, H=0.35, viscosity="DYNAMICVISC", jitter=0.0,
                 linearization=D.DEFAULT_LINEARIZATION, options="StillWaterSA"):
        super().__init__()
        self.m_name = "SABox"
        sp, pp = self.simparams, self.physparams
        sp.kerneltype = D.WENDLAND
        sp.boundarytype = D.SA_BOUNDARY
        self.set_viscosity(viscosity)
        if options == "StillWaterSA":
            sp.densitydiffusiontype = D.BREZZI
            sp.densityDiffCoeff = 0.05
            sp.simflags = D.ENABLE_DTADAPT | D.ENABLE_DENSITY_SUM
        elif options == "StillWaterRepackSA":
            sp.densitydiffusiontype = D.DENSITY_DIFFUSION_NONE
            sp.simflags = D.ENABLE_DTADAPT | D.ENABLE_REPACKING | D.ENABLE_GAMMA_QUADRATURE
        else:
            raise ValueError(options)
        self.options = options
        self.linearization = linearization
        self.jitter = jitter
        self.set_deltap(deltap)
        dp = self.m_deltap
        self.n_l, self.n_w, self.n_h = (int(round(v / dp)) for v in (l, w, h))
        self.l, self.w, self.h, self.H = self.n_l * dp, self.n_w * dp, self.n_h * dp, float(H)
        self.m_origin = np.full(3, -2.0 * dp)
        self.m_size = np.array([self.l, self.w, self.h], dtype=np.float64) + 4.0 * dp
        g = 9.81
        pp.gravity = (0.0, 0.0, -g)
        self.m_maxFall = self.H
        c0 = math.ceil(10.0 * math.sqrt(2.0 * g * self.H))
        pp.add_fluid(1000.0)
        pp.set_equation_of_state(0, 7.0, float(c0))
        pp.set_kinematic_visc(0, 1.0e-2)
        # two boundary elements per lattice square inside the wider boundary radius: near a corner of the tank a particle
        # sees ~150 fluid + boundary neighbours, more than the default section: resize_neiblist(128+128, 64)
        # (src/problems/StillWaterSA.cu:54, src/ProblemCore.h:341-353)
        sp.neibboundpos = 256 - 1
        sp.neiblistsize = 256 + 64
        self.initialize()
        self.fill_parts()

    def fill_parts(self):
        dp = self.m_deltap
        nl, nw, nh = self.n_l, self.n_w, self.n_h
        rho0 = self.physparams.rho0[0]
        # --- wall mesh: faces as (origin node, u step, v step, nu, nv, inward normal) on the integer lattice
        faces = [((0, 0, 0), (1, 0, 0), (0, 1, 0), nl, nw, (0, 0, 1)),      # floor
                 ((0, 0, 0), (0, 1, 0), (0, 0, 1), nw, nh, (1, 0, 0)),      # x = 0
                 ((nl, 0, 0), (0, 1, 0), (0, 0, 1), nw, nh, (-1, 0, 0)),    # x = l
                 ((0, 0, 0), (1, 0, 0), (0, 0, 1), nl, nh, (0, 1, 0)),      # y = 0
                 ((0, nw, 0), (1, 0, 0), (0, 0, 1), nl, nh, (0, -1, 0))]    # y = w
        vid = {}
        tris, normals = [], []
        def node(c):
            return vid.setdefault(c, len(vid))
        for o, du, dv, nu, nv, nrm in faces:
            o, du, dv = np.array(o), np.array(du), np.array(dv)
            # the vertices of an element run anticlockwise as seen from the fluid (the side its normal points to): the
            # analytical grad gamma of an element takes its edges in that sense (src/cuda/gamma.cuh:282-287, initConnectivity)
            ccw = np.dot(np.cross(du, dv), nrm) > 0
            for i in range(nu):
                for j in range(nv):
                    a = tuple(o + i * du + j * dv); b = tuple(o + (i + 1) * du + j * dv)
                    c = tuple(o + (i + 1) * du + (j + 1) * dv); d = tuple(o + i * du + (j + 1) * dv)
                    for tri in ((a, b, c), (a, c, d)):
                        tri = tri if ccw else (tri[0], tri[2], tri[1])
                        tris.append(tuple(node(v) for v in tri)); normals.append(nrm)
        vnodes = np.array(sorted(vid, key=vid.get), dtype=np.float64)
        vpos = vnodes * dp
        tris = np.array(tris, dtype=np.int64)
        spos = vpos[tris].mean(axis=1)
        normals = np.array(normals, dtype=np.float64)
        # vertex mass: the part of the dp^3 cube around the node that lies inside the tank
        frac = np.ones(len(vnodes))
        for a, n in enumerate((nl, nw, nh)):
            frac *= np.where((vnodes[:, a] == 0) | (vnodes[:, a] == n), 0.5, 1.0)
        # --- water
        fn = [nl - 1, nw - 1, max(int(round(self.H / dp)) - 1, 1)]
        fluid = _lattice(1, fn[0], 1, fn[1], 1, fn[2]).astype(np.float64) * dp
        if self.jitter:
            rng = np.random.default_rng(777)
            fluid = fluid + rng.uniform(-self.jitter * dp, self.jitter * dp, size=fluid.shape)
        self.water_level = (fn[2] + 0.5) * dp
        nf, ns, nv = len(fluid), len(spos), len(vpos)
        ntot = nf + ns + nv
        pos = np.empty((ntot, 4), dtype=np.float64)
        pos[:nf, :3] = fluid; pos[nf:nf + ns, :3] = spos; pos[nf + ns:, :3] = vpos
        pos[:nf + ns, 3] = rho0 * dp ** 3
        pos[nf + ns:, 3] = rho0 * dp ** 3 * frac
        vel = np.zeros((ntot, 4), dtype=np.float32)
        vel[:, 3] = self.initial_density(pos)
        tf = np.empty(ntot, dtype=np.uint16)
        tf[:nf] = D.PT_FLUID; tf[nf:nf + ns] = D.PT_BOUNDARY; tf[nf + ns:] = D.PT_VERTEX
        ids = np.arange(ntot, dtype=np.uint32)
        info = make_particleinfo(tf, np.zeros(ntot, dtype=np.uint16), ids)
        self.parts = HostParticles(pos, vel, info)
        self.vertices = np.zeros((ntot, 4), dtype=np.uint32)
        self.vertices[nf:nf + ns, :3] = (tris + nf + ns).astype(np.uint32)          # vertex ids
        self.boundelements = np.full((ntot, 4), np.nan, dtype=np.float32)
        self.boundelements[nf:nf + ns, :3] = normals
        self.boundelements[nf:nf + ns, 3] = 0.5 * dp * dp
        self.gradgamma = np.full((ntot, 4), np.nan, dtype=np.float32)
        self.num_fluid, self.num_segments, self.num_vertices, self.num_obstacle = nf, ns, nv, 0
        self.num_wall = ns + nv
        self.rb_firstindex = np.zeros(0, dtype=np.int32)
        self.rb_cg_gridpos = np.zeros((0, 3), dtype=np.int32)
        self.rb_cg_pos = np.zeros((0, 3), dtype=np.float32)

    def initial_density(self, pos_global):
        pp = self.physparams
        depth = np.clip(self.water_level - pos_global[:, 2], 0.0, None)
        return (np.power(1.0 + pp.rho0[0] * 9.81 * depth / pp.bcoeff[0], 1.0 / pp.gammacoeff[0]) - 1.0).astype(np.float32)

    def copy_to_array(self):
        a = super().copy_to_array()
        a.update(vertices=self.vertices.copy(), boundelements=self.boundelements.copy(), gradgamma=self.gradgamma.copy())
        return a


class SALoadBox(SABox):
    """SABox whose FLOOR is a body that feels the fluid: its segments carry FG_COMPUTE_FORCE and object number 0, so the forces
    engine writes the pressure force -P A n of every such element to BUFFER_RB_FORCES / BUFFER_RB_TORQUES
    (compute_boundary_pressure_force, src/cuda/forces_kernel.def:3258-3266,4115-4145) and REDUCE_BODIES_FORCES sums them -- the part
    of CompleteSaExample.cu's option set (a GT_FLOATING_BODY under SA_BOUNDARY, :122-124) that concerns the forces engine.  The body
    is fixed ("we might have fixed objects which still want to measure the force", the reference's comment at :3254): the known
    answer is the weight of the water, rho g V, pressing on the floor.  Vertices write no object forces with SA_BOUNDARY (:4120)."""

    def __init__(self, deltap=0.05, **kw):
        super().__init__(deltap, **kw)
        self.m_name = "SALoadBox"
        sp = self.simparams
        sp.numbodies = 1
        sp.numforcesbodies = 1
        info = self.parts.info
        nf = self.num_fluid
        nfloor = 2 * self.n_l * self.n_w                 # the floor is the first face meshed: its segments lead the PT_BOUNDARY rows
        load = np.zeros(len(info), dtype=bool)
        load[nf:nf + nfloor] = True
        assert (np.abs(self.boundelements[load, 2] - 1.0) < 1e-6).all()
        info[load, 0] |= D.FG_COMPUTE_FORCE
        info[load, 1] = (info[load, 1] & 0xF000) | 0
        self.load = load
        self.num_obstacle = int(nfloor)
        self.rb_firstindex = np.array([-nf], dtype=np.int32)      # row = id + rb_firstindex[object] (rb_particle_data, forces_kernel.def:526)
        cg = np.array([[0.5*self.l, 0.5*self.w, 0.0]])
        self.rb_cg_global = cg.copy()
        gcell = self.calc_grid_pos(cg)
        self.rb_cg_gridpos = gcell.astype(np.int32)
        self.rb_cg_pos = (cg - self.m_origin - (gcell + 0.5)*self.m_cellsize).astype(np.float32)


class SAPaddleBox(SABox):
    """SABox whose x = 0 wall is a MOVING body with prescribed motion (SA_BOUNDARY + ENABLE_MOVING_BODIES):
    its segments and vertices carry FG_MOVING_BOUNDARY and object number 0, and turn about the hinge line x = 0, z = 0 (a flap
    wave-maker, the SA counterpart of WaveTank's DYN paddle) while sliding along x -- rotation AND translation, so that the
    normals (sphx_sa_update_normals), the old / new elements of the density summation and the velocities of the moving segments
    all take part.  The reference has no ready-made problem of this kind without Crixus geometry files (CompleteSaExample.cu
    builds its moving parts from HDF5 meshes); what this mirror pins is the option set and, pass by pass, the oracle's
    restatement of the kernels.  A test geometry: the flap shares its edge vertices with the side walls and the floor, whose own
    segments stay where they are (a real flap would be a separate plate); over the few steps of a parity run that gap is
    1e-3 of deltap."""

    def __init__(self, deltap=0.05, *, omega=3.0, slide=0.2, **kw):
        super().__init__(deltap, **kw)
        self.m_name = "SAPaddleBox"
        sp = self.simparams
        sp.simflags |= D.ENABLE_MOVING_BODIES
        sp.numbodies = 1
        sp.numforcesbodies = 0
        self.flap_omega, self.flap_slide = float(omega), float(slide)
        info, g = self.parts.info, self.parts.pos_global
        t = info_type(info)
        nrm = self.boundelements
        flap = ((t == D.PT_VERTEX) & (np.abs(g[:, 0]) < 1e-9)) | ((t == D.PT_BOUNDARY) & (np.abs(g[:, 0]) < 1e-9) & (nrm[:, 0] > 0.5))
        info[flap, 0] |= D.FG_MOVING_BOUNDARY
        info[flap, 1] = (info[flap, 1] & 0xF000) | 0
        self.flap = flap
        self.num_obstacle = int(flap.sum())
        self.rb_firstindex = np.zeros(1, dtype=np.int32)
        cg = np.array([[0.0, 0.5*self.w, 0.0]])
        self.rb_cg_global = cg.copy()
        gcell = self.calc_grid_pos(cg)
        self.rb_cg_gridpos = gcell.astype(np.int32)
        self.rb_cg_pos = (cg - self.m_origin - (gcell + 0.5)*self.m_cellsize).astype(np.float32)
        self.moving_bodies_callback = self._flap

    def _flap(self, index, t0, t1, kd0, kd):
        """constant angular velocity about y through the hinge + constant slide along x; the hinge travels with the slide"""
        import math
        w, u = self.flap_omega, self.flap_slide
        kd.avel = np.array([0.0, w, 0.0]); kd.lvel = np.array([u, 0.0, 0.0])
        a = w*(t1 - t0)
        c, s = math.cos(a), math.sin(a)
        dx = np.array([u*(t1 - t0), 0.0, 0.0])
        kd.crot = kd.crot + dx
        return dx, np.array([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]])


class SAChannelIO(SABox):
    """An open channel on the SABox mesh, the synthetic counterpart of src/problems/ChannelIO.cu (whose geometry is a set of
    Crixus HDF5 files): the x = 0 wall of the tank is a VELOCITY-driven open boundary (u_E = U ex, setVelocityDriven(inlet, 1),
    :85-88), the x = l wall a PRESSURE-driven one, the fluid starts as a stream at U.  Framework options of that problem (:38-47):
    Brezzi diffusion, density summation, ENABLE_INLET_OUTLET | ENABLE_WATER_DEPTH, a neighbour-list rebuild in every iteration
    (:62).  Open boundaries are numbered 0 (inlet) and 1 (outlet): the object number of their segments and vertices, the index
    into BUFFER IOwaterdepth.

    Where this mirror is NOT ChannelIO.cu: (1) the imposed values.  ChannelIO's callback (:142-230) reads IOwaterdepth for its
    VELOCITY-driven boundary and holds the level of the pressure outlet at a constant 1.0, with U = 0.05; this mirror imposes the
    hydrostatic pressure under the MEASURED level at the pressure-driven outlet -- the rule of src/problems/CompleteSaExample.cu:268
    -- and a faster stream (U = 0.6 by default), so that the tests see particles enter and leave within tens of steps.  Nobody
    should expect number-for-number parity with a run of the tree's ChannelIO; what is held to the reference is the option set
    (test_channelio_framework_and_constants) and, pass by pass, the oracle's restatement of the kernels.  (2) the geometry (a box
    mesh made here instead of Crixus files).

    The passes run on the device (gpusph_amd/csrc/sa_io.hip, sa_bounds.hip; tests/test_gpu_sa_io.py)."""

    def __init__(self, deltap=0.05, *, U=0.6, l=1.0, w=0.4, h=0.4, H=0.25, water_depth=True, **kw):
        super().__init__(deltap, l=l, w=w, h=h, H=H, **kw)
        self.m_name = "SAChannelIO"
        self.U = float(U)
        sp = self.simparams
        sp.simflags |= D.ENABLE_INLET_OUTLET | (D.ENABLE_WATER_DEPTH if water_depth else 0)
        sp.buildneibsfreq = 1
        self.num_open_boundaries = 2
        info, g = self.parts.info, self.parts.pos_global
        t = info_type(info)
        wall = (t == D.PT_BOUNDARY) | (t == D.PT_VERTEX)
        nrm = self.boundelements
        inlet = wall & (np.abs(g[:, 0]) < 1e-6) & ((t == D.PT_VERTEX) | (nrm[:, 0] > 0.5))
        outlet = wall & (np.abs(g[:, 0] - self.l) < 1e-6) & ((t == D.PT_VERTEX) | (nrm[:, 0] < -0.5))
        info[inlet, 0] |= D.FG_INLET | D.FG_VELOCITY_DRIVEN
        info[inlet, 1] = (info[inlet, 1] & 0xF000) | 0
        info[outlet, 0] |= D.FG_OUTLET
        info[outlet, 1] = (info[outlet, 1] & 0xF000) | 1
        self.parts.vel[t == D.PT_FLUID, 0] = np.float32(self.U)

    def max_parts(self, numpart):
        """room for the particles the inlet releases (ChannelIO::max_parts, src/problems/ChannelIO.cu:95-98)"""
        return int(np.float32(numpart) * np.float32(1.2))

    def open_boundary_condition(self, io_info, abs_pos, waterdepth, t):
        """ChannelIO_imposeBoundaryCondition (src/problems/ChannelIO.cu:104-139) for the rows of the open boundaries' particles, on
        torch tensors of any device: -> (vel, eulerVel) rows.  The Lagrangian velocity is cleared; a velocity-driven boundary gets
        u_E = U ex, a pressure-driven one the density of the hydrostatic pressure under `waterdepth` (absolute z)."""
        import torch
        f32 = torch.float32
        m = io_info.shape[0]
        vel = torch.zeros((m, 4), dtype=f32, device=abs_pos.device)
        ev = torch.zeros((m, 4), dtype=f32, device=abs_pos.device)
        vdriven = (io_info[:, 0].to(torch.int32) & D.FG_VELOCITY_DRIVEN) != 0
        pp = self.physparams
        localdepth = torch.clamp(waterdepth - abs_pos[:, 2], min=0.0)
        pressure = np.float32(9.81) * localdepth * np.float32(pp.rho0[0])
        # RHO (src/cuda/phys_core.cu:108-112)
        rho = torch.pow(pressure / np.float32(pp.bcoeff[0]) + np.float32(1.0), float(np.float32(1.0) / np.float32(pp.gammacoeff[0]))) - np.float32(1.0)
        ev[:, 3] = torch.where(vdriven, torch.zeros_like(rho), rho)
        ev[:, 0] = torch.where(vdriven, torch.full_like(rho, np.float32(self.U)), torch.zeros_like(rho))
        return vel, ev

    def impose_open_boundaries(self, pos, vel, euler_vel, info, hash_, iowaterdepth, t, n):
        """<Problem>_imposeBoundaryConditionDevice + imposeBoundaryConditionHost (src/problems/ChannelIO.cu:142-230): for every
        particle of an open boundary the absolute position, the water level of its boundary from IOwaterdepth (pressure-driven ones;
        0 -> the bottom of the domain), the problem's condition, written to vel / eulerVel in place; then IOwaterdepth is cleared.
        Tensors of the engine (device or CPU), the first n rows."""
        import torch
        flags = info[:n, 0].to(torch.int32)
        rows = torch.nonzero((flags & (D.FG_INLET | D.FG_OUTLET)) != 0).flatten()
        if rows.numel():
            f32 = torch.float32
            dev = pos.device
            c1, c2, c3 = D.LINEARIZATIONS[self.linearization]
            gs = [int(v) for v in self.m_gridsize]
            h = hash_[rows].to(torch.int64) & D.CELLTYPE_BITMASK
            g = [None, None, None]
            tt = gs[c2] * gs[c1]
            g[c3] = torch.div(h, tt, rounding_mode="floor")
            rem = h - g[c3] * tt
            g[c2] = torch.div(rem, gs[c1], rounding_mode="floor")
            g[c1] = rem - g[c2] * gs[c1]
            cell = [np.float32(v) for v in self.m_cellsize]
            org = [np.float32(v) for v in self.m_origin]
            # d_worldOrigin + pos + gridPos*d_cellSize + 0.5f*d_cellSize, left to right in float
            ap = torch.stack([((org[a] + pos[rows, a]) + g[a].to(f32) * cell[a]) + np.float32(0.5) * cell[a] for a in range(3)], dim=1)
            io_info = info[rows]
            wd = torch.zeros(rows.numel(), dtype=f32, device=dev)
            if iowaterdepth is not None:
                obj = io_info[:, 1].to(torch.int32) & 0xFFF
                u = (iowaterdepth.to(torch.int64) & 0xFFFFFFFF)[obj.to(torch.int64)].to(f32)
                wd = u / np.float32(4294967296.0)                           # ((float)IOwaterdepth)/((float)UINT_MAX)
                wd = wd * (cell[2] * np.float32(gs[2]))
                wd = wd + org[2]
                vdriven = (io_info[:, 0].to(torch.int32) & D.FG_VELOCITY_DRIVEN) != 0
                wd = torch.where(vdriven, torch.zeros_like(wd), wd)
            v, e = self.open_boundary_condition(io_info, ap, wd, t)
            vel[rows] = v
            euler_vel[rows] = e
        if iowaterdepth is not None:
            iowaterdepth.zero_()



class SAChannelIOFlap(SAChannelIO):
    """SAChannelIO with a MOVING body on top: ENABLE_INLET_OUTLET | ENABLE_DENSITY_SUM | ENABLE_MOVING_BODIES, the option set of the
    reference's src/problems/CompleteSaExample.cu (:46; there an inlet, an outlet and a floating cube from Crixus meshes).  The
    y = w side wall of the channel is a flap with prescribed motion -- its segments and the vertices that are not part of an open
    face carry FG_MOVING_BOUNDARY and object number 2 (objects 0 and 1 are the open boundaries: the object number is shared by
    bodies and open boundaries as in the reference, so bodies 0 and 1 of the rigid-body tables are there and at rest) -- turning about
    the hinge line y = w, z = 0 while sliding along y.  What this mirror exercises is the command sequence where the two features
    meet: update_normals behind both Euler steps, the density summation between two states of the elements WITH the open faces'
    terms (sphx_sa_density_sum_io_moving), the condition passes, the take-over and release of particles and the per-step rebuild
    with BUFFER_BOUNDELEMENTS as state.  A test geometry, like SAPaddleBox (the flap shares its edge vertices with the floor)."""

    def __init__(self, deltap=0.05, *, omega=2.0, slide=0.1, **kw):
        super().__init__(deltap, **kw)
        self.m_name = "SAChannelIOFlap"
        sp = self.simparams
        sp.simflags |= D.ENABLE_MOVING_BODIES
        sp.numbodies = 3
        sp.numforcesbodies = 0
        self.flap_omega, self.flap_slide = float(omega), float(slide)
        info, g = self.parts.info, self.parts.pos_global
        t = info_type(info)
        nrm = self.boundelements
        opened = (info[:, 0] & (D.FG_INLET | D.FG_OUTLET)) != 0
        onwall = np.abs(g[:, 1] - self.w) < 1e-9
        flap = (((t == D.PT_VERTEX) & onwall) | ((t == D.PT_BOUNDARY) & onwall & (nrm[:, 1] < -0.5))) & ~opened
        info[flap, 0] |= D.FG_MOVING_BOUNDARY
        info[flap, 1] = (info[flap, 1] & 0xF000) | 2
        self.flap = flap
        self.num_obstacle = int(flap.sum())
        self.rb_firstindex = np.zeros(3, dtype=np.int32)
        cg = np.array([[0.0, 0.0, 0.0], [self.l, 0.0, 0.0], [0.5*self.l, self.w, 0.0]])
        self.rb_cg_global = cg.copy()
        gcell = self.calc_grid_pos(cg)
        self.rb_cg_gridpos = gcell.astype(np.int32)
        self.rb_cg_pos = (cg - self.m_origin - (gcell + 0.5)*self.m_cellsize).astype(np.float32)
        self.moving_bodies_callback = self._flap

    def _flap(self, index, t0, t1, kd0, kd):
        """body 2: constant angular velocity about x through the hinge + constant slide along y (the hinge travels with the slide);
        bodies 0 and 1 (the open boundaries' object numbers) do not move"""
        import math
        if index != 2:
            return np.zeros(3), np.eye(3)
        w, u = self.flap_omega, self.flap_slide
        kd.avel = np.array([w, 0.0, 0.0]); kd.lvel = np.array([0.0, u, 0.0])
        a = w*(t1 - t0)
        c, s = math.cos(a), math.sin(a)
        dx = np.array([0.0, u*(t1 - t0), 0.0])
        kd.crot = kd.crot + dx
        return dx, np.array([[1.0, 0.0, 0.0], [0.0, c, -s], [0.0, s, c]])

class OpenChannel(Problem):
    """The option set and dimensions of src/problems/OpenChannel.cu: a channel inclined by 4.5 degrees whose flow is driven by
    gravity, periodic along the stream (and across it without side walls), DYN_BOUNDARY bottom (and side walls) of
    ceil(influenceRadius/dp) + 1 particle layers, viscosity<KINEMATICVISC> with mu = 110 Pa s of a fluid of 2650 kg/m^3,
    EOS gamma = 2, c0 = 20 m/s, dt0 = 4e-5, dtadaptfactor 0.3, rebuild every 10 iterations, water level H = 0.5, width
    a = 1, length 15 influence radii (:41-112).  The particle placement is this repository's lattice (seamless across the periodic
    faces: first layer at dp/2, as the reference's periodicity_gap has it), not a reproduction of the tree's Cube fill."""

    def __init__(self, deltap=0.02, *, sidewalls=True, linearization=D.DEFAULT_LINEARIZATION):
        super().__init__()
        self.m_name = "OpenChannel"
        self.sidewalls = bool(sidewalls)
        sp, pp = self.simparams, self.physparams
        sp.kerneltype = D.WENDLAND
        sp.boundarytype = D.DYN_BOUNDARY
        self.set_viscosity("KINEMATICVISC")
        sp.densitydiffusiontype = D.DENSITY_DIFFUSION_NONE
        sp.periodicbound = D.PERIODIC_X if self.sidewalls else (D.PERIODIC_X | D.PERIODIC_Y)
        sp.simflags = D.ENABLE_DTADAPT
        self.linearization = linearization
        self.set_deltap(deltap)
        dp = self.m_deltap
        sp.dt = 0.00004
        sp.dtadaptfactor = 0.3
        sp.buildneibsfreq = 10
        self.H = 0.5
        self.m_maxFall = self.H
        infl = float(sp.influenceRadius)
        self.dyn_layers = int(math.ceil(infl / dp)) + 1
        ru = lambda x: math.ceil(round(x / dp, 9)) * dp           # round_up(x, dp)
        self.a, self.h, self.l = ru(1.0), ru(self.H * 1.4), ru(15.0 * infl)
        margin = np.array([0.0, 0.1 if self.sidewalls else 0.0, 0.1])
        lay = (self.dyn_layers - 1) * dp
        # the world holds the boundary layers (the reference's 0.1 margin does at its deltap = 0.02; kept at least that wide)
        margin = np.maximum(margin, np.array([0.0, lay + dp if self.sidewalls else 0.0, lay + dp]))
        self.m_size = np.array([self.l, self.a, self.h]) + 2.0 * margin
        self.m_origin = -margin
        angle, g = 4.5, float(np.float32(9.81))
        pp.gravity = (g * math.sin(math.pi * angle / 180.0), 0.0, -g * math.cos(math.pi * angle / 180.0))
        pp.add_fluid(2650.0)
        pp.set_equation_of_state(0, 2.0, 20.0)
        pp.set_dynamic_visc(0, 110.0)
        self.initialize()
        nl, na, nH, nh = (int(round(v / dp)) for v in (self.l, self.a, self.H, self.h))
        x = (np.arange(nl) + 0.5) * dp
        if self.sidewalls:
            yf = np.arange(1, na) * dp                        # fluid one dp off the walls at y = 0 and y = a
            yb = np.arange(-(self.dyn_layers - 1), na + self.dyn_layers) * dp
        else:
            yf = (np.arange(na) + 0.5) * dp
            yb = yf
        zf = np.arange(1, nH) * dp
        grid = lambda xs, ys, zs: np.stack(np.meshgrid(xs, ys, zs, indexing="ij"), axis=-1).reshape(-1, 3)
        fl = grid(x, yf, zf)
        walls = [grid(x, yb, -np.arange(self.dyn_layers) * dp)]             # bottom: layers at z = 0, -dp, ...
        if self.sidewalls:
            zw = np.arange(1, nh) * dp
            walls.append(grid(x, -np.arange(self.dyn_layers) * dp, zw))
            walls.append(grid(x, self.a + np.arange(self.dyn_layers) * dp, zw))
        wl = np.concatenate(walls)
        nf, nw = len(fl), len(wl)
        pos = np.empty((nf + nw, 4), dtype=np.float64)
        pos[:nf, :3], pos[nf:, :3] = fl, wl
        pos[:, 3] = pp.rho0[0] * dp ** 3
        vel = np.zeros((nf + nw, 4), dtype=np.float32)
        # hydrostatic filling normal to the bed
        depth = np.clip(self.H - pos[:, 2], 0.0, None)
        vel[:, 3] = (np.power(1.0 + pp.rho0[0] * (-pp.gravity[2]) * depth / pp.bcoeff[0], 1.0 / pp.gammacoeff[0]) - 1.0).astype(np.float32)
        types = np.concatenate([np.full(nf, D.PT_FLUID, dtype=np.uint16), np.full(nw, D.PT_BOUNDARY, dtype=np.uint16)])
        info = make_particleinfo(types, np.zeros(nf + nw, dtype=np.uint16), np.arange(nf + nw, dtype=np.uint32))
        self.parts = HostParticles(pos, vel, info)
        self.num_fluid, self.num_wall, self.num_obstacle = nf, nw, 0
        self.rb_firstindex = np.zeros(0, dtype=np.int32)
        self.rb_cg_gridpos = np.zeros((0, 3), dtype=np.int32)
        self.rb_cg_pos = np.zeros((0, 3), dtype=np.float32)
