"""SimParams / PhysParams mirror (GPUSPH src/simparams.h:262-389, src/physparams.h) and their
flattening into the C ABI's sphx_params (include/sphx.h)."""
import ctypes as C
import math
from dataclasses import dataclass, field
import numpy as np
from . import defs as D


class SphxParams(C.Structure):
    """ctypes image of `struct sphx_params` (include/sphx.h) -- keep field order in sync."""
    _fields_ = [
        ("gridSize", C.c_uint32 * 3), ("cellSize", C.c_float * 3), ("worldOrigin", C.c_float * 3),
        ("coord", C.c_int32 * 3), ("periodic", C.c_uint32),
        ("neiblistsize", C.c_uint32), ("neibboundpos", C.c_uint32), ("neiblist_stride", C.c_uint64),
        ("kerneltype", C.c_int32), ("sph_formulation", C.c_int32), ("densitydiffusiontype", C.c_int32),
        ("boundarytype", C.c_int32), ("rheologytype", C.c_int32), ("turbmodel", C.c_int32),
        ("compvisc", C.c_int32), ("viscmodel", C.c_int32), ("avgop", C.c_int32),
        ("simflags", C.c_uint64),
        ("slength", C.c_float), ("kernelradius", C.c_float), ("influenceradius", C.c_float),
        ("deltap", C.c_float), ("dtadaptfactor", C.c_float), ("densityDiffCoeff", C.c_float),
        ("epsxsph", C.c_float),
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
    ]


@dataclass
class PhysParams:
    """src/physparams.h: defaults :380-420, set_equation_of_state :506-516."""
    rho0: list = field(default_factory=list)
    bcoeff: list = field(default_factory=list)
    gammacoeff: list = field(default_factory=list)
    sscoeff: list = field(default_factory=list)
    sspowercoeff: list = field(default_factory=list)
    visccoeff: list = field(default_factory=list)
    kinematicvisc: list = field(default_factory=list)      # nu per fluid (physparams.h:175)
    visc_consistency: list = field(default_factory=list)   # mu per fluid for Newtonian fluids
    # generalized Newtonian rheologies (physparams.h:185-243): yield strength, power-law exponent / exponential coefficient,
    # regularisation parameter per fluid; the effective viscosity is clamped to limiting_kinvisc (x rho0)
    yield_strength: list = field(default_factory=list)
    visc_nonlinear_param: list = field(default_factory=list)
    visc_regularization_param: list = field(default_factory=list)
    limiting_kinvisc: float = 1.0e3                         # physparams.h:395
    rheologytype: int = 0                                   # PhysParams is built for the framework's rheology (physparams.h:380)
    monaghan_visc_coeff: float = 10.0                       # physparams.h:266,396
    bulkvisc: list = field(default_factory=list)            # Espanol & Revenga (physparams.h:176)
    visc2coeff: list = field(default_factory=list)          # d_visc2coeff (physparams.h:247)
    # ENABLE_DEM (physparams.h:322-327; computeDEMphysparams, src/problem_api/ProblemAPI_1.cc:1399-1418)
    ewres: float = float("nan")
    nsres: float = float("nan")
    demdx: float = float("nan")
    demdy: float = float("nan")
    demzmin: float = float("nan")
    partsurf: float = 0.0                                  # physparams.h:328,403 (0 -> r0^2 on upload)
    MK_K: float = float("nan")                             # physparams.h:336-338,405-407
    MK_d: float = float("nan")
    MK_beta: float = 2.0
    epsinterface: float = float("nan")                     # physparams.h; ProblemCore.cc:165-166 -> 0.05 for SPH_GRENIER
    gravity: tuple = (0.0, 0.0, -9.81)
    artvisccoeff: float = 0.3          # physparams.h:392
    epsartvisc: float = float("nan")   # defaulted to 0.01 h^2 in ProblemCore.cc:160-163
    smagorinsky_constant: float = 0.12  # physparams.h:411-412
    isotropic_sps_constant: float = 0.0066
    smagfactor: float = float("nan")
    kspsfactor: float = float("nan")
    dcoeff: float = float("nan")       # physparams.h:399 (defaulted for LJ_BOUNDARY in ProblemCore.cc:126-138)
    p1coeff: float = 12.0
    p2coeff: float = 6.0
    r0: float = float("nan")
    cosconeanglefluid: float = 0.86     # free-surface detection cone angles (src/physparams.h:418-419)
    cosconeanglenonfluid: float = 0.5

    def numFluids(self):
        return len(self.rho0)

    def add_fluid(self, rho):
        self.rho0.append(float(np.float32(rho)))
        for lst in (self.bcoeff, self.gammacoeff, self.sscoeff, self.sspowercoeff, self.visccoeff,
                    self.kinematicvisc, self.visc_consistency):
            lst.append(float("nan"))
        # the values that reduce back to a Newtonian rheology (physparams.h:469-480)
        self.yield_strength.append(0.0)
        self.bulkvisc.append(float("nan")); self.visc2coeff.append(float("nan"))
        self.visc_nonlinear_param.append(0.0 if self.rheologytype >= D.DEKEE_TURCOTTE else 1.0)
        self.visc_regularization_param.append(1000.0)
        return len(self.rho0) - 1

    def update_limiting_kinvisc(self, fluid_idx):
        """physparams.h:599-603"""
        new_limit = np.float32(self.yield_strength[fluid_idx]) * np.float32(self.visc_regularization_param[fluid_idx]) + \
            np.float32(self.visc_consistency[fluid_idx])
        self.limiting_kinvisc = float(np.fmax(np.float32(self.limiting_kinvisc), new_limit))      # fmaxf: a NaN operand loses

    def set_bulk_visc(self, fluid_idx, zeta):
        self.bulkvisc[fluid_idx] = float(np.float32(zeta))

    def set_consistency_index(self, fluid_idx, k):
        self.set_dynamic_visc(fluid_idx, k)

    def set_yield_strength(self, fluid_idx, ys):
        self.yield_strength[fluid_idx] = float(np.float32(ys))
        self.update_limiting_kinvisc(fluid_idx)

    def set_visc_power_law(self, fluid_idx, n):
        if not (D.POWER_LAW <= self.rheologytype < D.DEKEE_TURCOTTE):
            raise ValueError("the rheological model is not power-law")          # must_be_power_law_rheology
        self.visc_nonlinear_param[fluid_idx] = float(np.float32(n))

    def set_visc_exponential_coeff(self, fluid_idx, t1):
        if self.rheologytype < D.DEKEE_TURCOTTE:
            raise ValueError("the rheological model is not exponential")
        self.visc_nonlinear_param[fluid_idx] = float(np.float32(t1))

    def set_visc_regularization_param(self, fluid_idx, m):
        self.visc_regularization_param[fluid_idx] = float(np.float32(m))

    def set_limiting_kinvisc(self, max_visc):
        self.limiting_kinvisc = float(np.float32(max_visc))

    def set_kinematic_visc(self, fluid_idx, nu):
        """physparams.h:609-616"""
        nu = np.float32(nu)
        self.kinematicvisc[fluid_idx] = float(nu)
        self.visc_consistency[fluid_idx] = float(nu * np.float32(self.rho0[fluid_idx]))
        self.update_limiting_kinvisc(fluid_idx)

    def set_dynamic_visc(self, fluid_idx, mu):
        """physparams.h:622-629"""
        mu = np.float32(mu)
        self.kinematicvisc[fluid_idx] = float(mu / np.float32(self.rho0[fluid_idx]))
        self.visc_consistency[fluid_idx] = float(mu)
        self.update_limiting_kinvisc(fluid_idx)

    def update_visccoeff(self, sp):
        """GPUSPH::setViscosityCoefficient (src/GPUSPH.cc:1480-1508): what d_visccoeff holds"""
        for f in range(self.numFluids()):
            if sp.rheologytype == D.INVISCID:
                self.visccoeff[f] = float("nan")
            elif sp.rheologytype == D.NEWTONIAN and sp.compvisc == D.KINEMATIC:
                self.visccoeff[f] = self.kinematicvisc[f]
            else:
                self.visccoeff[f] = self.visc_consistency[f]
        if sp.viscmodel == D.ESPANOL_REVENGA:        # GPUSPH.cc:1511-1522: unset bulk viscosity is zero; 3 zeta <= 5 mu
            for f in range(self.numFluids()):
                if math.isnan(self.bulkvisc[f]):
                    self.bulkvisc[f] = 0.0
                if self.bulkvisc[f] * 3 > self.visc_consistency[f] * 5:
                    raise ValueError("fluid %d cannot be modelled with Espanol & Revenga (bulk viscosity too large)" % f)
                self.visc2coeff[f] = self.bulkvisc[f]

    def set_equation_of_state(self, fluid_idx, gamma, c0):
        if fluid_idx >= self.numFluids():
            raise IndexError("trying to set equation of state for a non-existing fluid")
        f32 = np.float32
        g, c = f32(gamma), f32(c0)
        self.gammacoeff[fluid_idx] = float(g)
        self.bcoeff[fluid_idx] = float(f32(f32(f32(self.rho0[fluid_idx]) * c) * c) / g)
        self.sscoeff[fluid_idx] = float(c)
        self.sspowercoeff[fluid_idx] = float((g - f32(1)) / f32(2))


@dataclass
class SimParams:
    """src/simparams.h:262-314 defaults."""
    kerneltype: int = D.WENDLAND
    sph_formulation: int = D.SPH_F1
    densitydiffusiontype: int = D.DENSITY_DIFFUSION_NONE
    boundarytype: int = D.LJ_BOUNDARY
    rheologytype: int = D.INVISCID
    turbmodel: int = D.ARTIFICIAL
    compvisc: int = D.KINEMATIC
    viscmodel: int = D.MORRIS
    avgop: int = D.ARITHMETIC
    periodicbound: int = D.PERIODIC_NONE
    simflags: int = D.ENABLE_DTADAPT
    sfactor: float = 1.3
    kernelradius: float = 2.0
    slength: float = 0.0
    influenceRadius: float = 0.0
    nlexpansionfactor: float = 1.0
    nlInfluenceRadius: float = 0.0
    nlSqInfluenceRadius: float = 0.0
    dtadaptfactor: float = 0.3
    buildneibsfreq: int = 10
    neiblistsize: int = 0
    neibboundpos: int = 0
    densityDiffCoeff: float = float("nan")
    epsxsph: float = 0.5
    dt: float = 0.0
    numbodies: int = 0
    numforcesbodies: int = 0
    repack_maxiter: int = 2000         # simparams.h:308-310
    repack_a: float = 0.1
    repack_alpha: float = 0.01
    is_const_visc: object = None       # None: FullViscSpec default (single fluid, NEWTONIAN, not k-epsilon; visc_spec.h:268-272)

    def set_smoothing(self, smooth, deltap):
        """simparams.h:325-336 (double arithmetic)."""
        self.sfactor = smooth
        self.slength = smooth * deltap
        self.set_influenceradius()
        return self.slength

    def set_influenceradius(self):
        """simparams.h:368-375."""
        self.influenceRadius = self.slength * self.kernelradius
        self.nlInfluenceRadius = self.nlexpansionfactor * self.influenceRadius
        self.nlSqInfluenceRadius = self.nlInfluenceRadius * self.nlInfluenceRadius
        return self.influenceRadius


def check_neiblistsize(sp: SimParams, pp: PhysParams, deltap: float):
    """ProblemCore::check_neiblistsize (src/ProblemCore.cc:806-887)."""
    r = math.ceil(sp.sfactor * sp.kernelradius)
    vol = math.ceil(4 * 3.2 * r * r * r / 3)
    neiblistsize = ((int(vol) + 31) // 32) * 32
    qq = deltap / pp.r0 if pp.r0 and not math.isnan(pp.r0) else 1.0
    ratio = max(qq * qq / r, 1.0)
    neiblistsize = int(math.ceil(ratio * neiblistsize))
    neiblistsize = ((neiblistsize + 31) // 32) * 32
    neibboundpos = neiblistsize - 1
    if sp.boundarytype == D.SA_BOUNDARY:     # "boundary particles are doubled": the vertex section, :859-865
        neiblistsize = ((3 * neiblistsize // 2 + 31) // 32) * 32
    if sp.neiblistsize == 0:
        sp.neiblistsize = neiblistsize
    if sp.neibboundpos == 0:
        sp.neibboundpos = neibboundpos if sp.boundarytype == D.SA_BOUNDARY else sp.neiblistsize - 1
    return sp.neiblistsize, sp.neibboundpos


def make_sphx_params(sp: SimParams, pp: PhysParams, *, gridsize, cellsize, origin, deltap,
                     allocated, linearization=D.DEFAULT_LINEARIZATION) -> SphxParams:
    """What the three setconstants() calls upload (src/cuda/forces.cu:268-399 etc.)."""
    pp.update_visccoeff(sp)
    p = SphxParams()
    f32 = lambda v: float(np.float32(v))
    for a in range(3):
        p.gridSize[a] = int(gridsize[a])
        p.cellSize[a] = f32(cellsize[a])
        p.worldOrigin[a] = f32(origin[a])
        p.coord[a] = D.LINEARIZATIONS[linearization][a]
        p.gravity[a] = f32(pp.gravity[a])
    p.periodic = sp.periodicbound
    p.neiblistsize = sp.neiblistsize
    p.neibboundpos = sp.neibboundpos
    p.neiblist_stride = int(allocated)
    p.kerneltype = sp.kerneltype
    p.sph_formulation = sp.sph_formulation
    p.densitydiffusiontype = sp.densitydiffusiontype
    p.boundarytype = sp.boundarytype
    p.rheologytype = sp.rheologytype
    p.turbmodel = sp.turbmodel
    p.compvisc = sp.compvisc
    p.viscmodel = sp.viscmodel
    p.avgop = sp.avgop
    p.simflags = sp.simflags
    p.slength = f32(sp.slength)
    p.kernelradius = f32(sp.kernelradius)
    p.influenceradius = f32(sp.influenceRadius)
    p.deltap = f32(deltap)
    p.dtadaptfactor = f32(sp.dtadaptfactor)
    p.densityDiffCoeff = f32(sp.densityDiffCoeff)      # NaN when no density diffusion is selected (simparams.h:287)
    p.epsxsph = f32(sp.epsxsph)
    p.numfluids = pp.numFluids()
    for f in range(pp.numFluids()):
        p.rho0[f] = f32(pp.rho0[f]); p.bcoeff[f] = f32(pp.bcoeff[f]); p.gammacoeff[f] = f32(pp.gammacoeff[f])
        p.sscoeff[f] = f32(pp.sscoeff[f]); p.sspowercoeff[f] = f32(pp.sspowercoeff[f])
        p.visccoeff[f] = f32(pp.visccoeff[f])      # NaN for INVISCID, as GPUSPH::setViscosityCoefficient leaves it
    p.artvisccoeff = f32(pp.artvisccoeff)
    p.epsartvisc = f32(pp.epsartvisc)
    nz = lambda v: float('nan') if v is None else f32(v)   # unset coefficients stay NaN, as in the tree's PhysParams
    p.smagfactor = nz(pp.smagfactor); p.kspsfactor = nz(pp.kspsfactor)
    p.dcoeff = nz(pp.dcoeff); p.p1coeff = nz(pp.p1coeff); p.p2coeff = nz(pp.p2coeff); p.r0 = nz(pp.r0)
    p.repack_a = f32(sp.repack_a); p.repack_alpha = f32(sp.repack_alpha)
    const = sp.is_const_visc
    if const is None:
        # IS_SINGLEFLUID && NEWTONIAN && turbmodel != KEPSILON
        const = not (sp.simflags & D.ENABLE_MULTIFLUID) and sp.rheologytype == D.NEWTONIAN and sp.turbmodel != D.KEPSILON
    p.is_const_visc = 1 if const else 0
    p.partsurf = f32(pp.partsurf)
    p.MK_K = nz(pp.MK_K); p.MK_d = nz(pp.MK_d); p.MK_beta = nz(pp.MK_beta)
    p.epsinterface = nz(pp.epsinterface)
    for f in range(pp.numFluids()):
        p.yield_strength[f] = f32(pp.yield_strength[f]); p.visc_nonlinear_param[f] = f32(pp.visc_nonlinear_param[f])
        p.visc_regularization_param[f] = f32(pp.visc_regularization_param[f])
    p.limiting_kinvisc = f32(pp.limiting_kinvisc)
    p.monaghan_visc_coeff = f32(pp.monaghan_visc_coeff)
    for f in range(pp.numFluids()):
        p.visc2coeff[f] = nz(pp.visc2coeff[f])
    p.ewres = nz(pp.ewres); p.nsres = nz(pp.nsres); p.demdx = nz(pp.demdx); p.demdy = nz(pp.demdy); p.demzmin = nz(pp.demzmin)
    return p
