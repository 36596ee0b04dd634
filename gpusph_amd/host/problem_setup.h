// problem_setup.h -- what stands where a GPUSPH Problem stands in the two host programs of this directory
// (framework_check.cc, example_engines.cpp): the SETUP_FRAMEWORK(...) expressions of the reference's problems, and the
// parameter set-up a Problem's constructor + ProblemCore::initialize + GPUSPH::setViscosityCoefficient perform on the
// tree's own SimParams / PhysParams.  Cases are text files with one "key value..." pair per line, written by the tests
// from the Python problem mirrors.
#ifndef SPHX_PROBLEM_SETUP_H
#define SPHX_PROBLEM_SETUP_H

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "cudasimframework.cu"

// the two buffer-debugging switches src/buffer.h:49-50 declares extern; GPUSPH's main() sets them from --debug
// (src/debugflags.cc:43-51, which needs a generated header and is not built here): off
bool debug_inspect_buffer = false;
bool debug_clobber_invalid_buffers = false;

// a Problem is a friend of PhysParams (src/physparams.h:428-429); this program stands where a Problem stands
struct ProblemPhysParams : PhysParams {
	ProblemPhysParams(RheologyType r) : PhysParams(r) {}
	using PhysParams::add_fluid; using PhysParams::set_equation_of_state;
	using PhysParams::set_kinematic_visc; using PhysParams::set_dynamic_visc;
	using PhysParams::set_artificial_visc;
	using PhysParams::set_yield_strength; using PhysParams::set_visc_power_law; using PhysParams::set_visc_exponential_coeff;
	using PhysParams::set_visc_regularization_param; using PhysParams::is_exponential_rheology; using PhysParams::set_bulk_visc;
};

typedef std::map<std::string, std::vector<std::string> > Case;

static Case read_case(const char *path)
{
	Case c;
	std::ifstream in(path);
	if (!in) throw std::runtime_error(std::string("cannot open ") + path);
	std::string line;
	while (std::getline(in, line)) {
		std::istringstream ls(line);
		std::string key, v;
		if (!(ls >> key)) continue;
		while (ls >> v) c[key].push_back(v);
	}
	return c;
}
static double num(Case const& c, const char *key, size_t i = 0)
{
	Case::const_iterator it = c.find(key);
	if (it == c.end() || it->second.size() <= i) throw std::runtime_error(std::string("case lacks ") + key);
	return strtod(it->second[i].c_str(), NULL);
}
static bool has(Case const& c, const char *key) { return c.count(key) != 0; }
static std::string str(Case const& c, const char *key) { return c.at(key).at(0); }

// The SETUP_FRAMEWORK(...) expressions of the reference's problems, verbatim in their selector lists; the macro itself is
// `this->simframework() = CUDASimFramework< __VA_ARGS__ >()` (src/ProblemCore.h:117)
#define SETUP_FRAMEWORK(...) simframework = CUDASimFramework< __VA_ARGS__ >()

static SimFramework *make_framework(Case const& c)
{
	const std::string name = str(c, "framework");
	SimFramework *simframework = NULL;
	if (name == "DamBreak3D") {            // src/problems/DamBreak3D.cu:45-60
		const DensityDiffusionType RHODIFF = (DensityDiffusionType)(int)num(c, "rhodiff");
		const bool USE_PLANES = num(c, "use_planes") != 0;
		SETUP_FRAMEWORK(
			viscosity<ARTVISC>,
			boundary<DYN_BOUNDARY>,
			add_flags<ENABLE_REPACKING>
		).select_options(
			RHODIFF,
			USE_PLANES, add_flags<ENABLE_PLANES>()
		);
	} else if (name == "StillWater") {     // src/problems/StillWater.cu:54-65
		const DensityDiffusionType rhodiff = (DensityDiffusionType)(int)num(c, "rhodiff");
		const bool m_usePlanes = num(c, "use_planes") != 0;
		SETUP_FRAMEWORK(
			viscosity<DYNAMICVISC>,
			boundary<DYN_BOUNDARY>
		).select_options(
			rhodiff,
			m_usePlanes, add_flags<ENABLE_PLANES>()
		);
	} else if (name == "StillWaterSPS") {  // the variant config 3 asks for: SPS viscosity (the commented-out selectors of StillWater.cu:57-59)
		const DensityDiffusionType rhodiff = (DensityDiffusionType)(int)num(c, "rhodiff");
		SETUP_FRAMEWORK(
			viscosity<SPSVISC>,
			boundary<DYN_BOUNDARY>
		).select_options(rhodiff);
	} else if (name == "WaveTank") {       // src/problems/WaveTank.cu:55-62
		SETUP_FRAMEWORK(
			viscosity<SPSVISC>,
			boundary<LJ_BOUNDARY>,
			add_flags<ENABLE_PLANES>
		);
	} else if (name == "DamBreakGate") {   // src/problems/DamBreakGate.cu:53-58
		SETUP_FRAMEWORK(
			viscosity<ARTVISC>,
			boundary<LJ_BOUNDARY>,
			add_flags<ENABLE_MOVING_BODIES>
		);
	} else if (name == "OpenChannel") {    // src/problems/OpenChannel.cu:46-53
		const bool use_side_walls = num(c, "use_side_walls") != 0;
		SETUP_FRAMEWORK(
			viscosity<KINEMATICVISC>,
			boundary<DYN_BOUNDARY>,
			periodicity<PERIODIC_XY>
		).select_options(
			use_side_walls, periodicity<PERIODIC_X>()
		);
	} else if (name == "Spheric2LJ") {     // src/problems/Spheric2LJ.cu:63-72
		const bool m_usePlanes = num(c, "use_planes") != 0;
		SETUP_FRAMEWORK(
			kernel<WENDLAND>,
			viscosity<ARTVISC>,
			boundary<LJ_BOUNDARY>,
			densitydiffusion<FERRARI>
		).select_options(
			m_usePlanes, add_flags<ENABLE_PLANES>()
		);
	} else if (name == "LockExchangeF2") { // src/problems/LockExchange.cu:44-53 with the SPH_F2 line it carries commented out
		const bool USE_PLANES = num(c, "use_planes") != 0;
		SETUP_FRAMEWORK(
			formulation<SPH_F2>,
			viscosity<DYNAMICVISC>,
			boundary<DYN_BOUNDARY>,
			add_flags<ENABLE_MULTIFLUID>
		).select_options(
			USE_PLANES, add_flags<ENABLE_PLANES>()
		);
	} else if (name == "Bubble") {         // src/problems/Bubble.cu:55-61 (Grenier: harmonic averaging by legacy rule)
		SETUP_FRAMEWORK(
			formulation<SPH_GRENIER>,
			viscosity<DYNAMICVISC>,
			boundary<DYN_BOUNDARY>,
			add_flags<ENABLE_MULTIFLUID>
		);
	} else if (name == "MultiFluidSPS") {  // legacy SPSVISC with several fluids: not a constant-viscosity specification
		SETUP_FRAMEWORK(
			viscosity<SPSVISC>,
			boundary<DYN_BOUNDARY>,
			add_flags<ENABLE_MULTIFLUID>
		);
	} else if (name == "StillWaterSA") {   // src/problems/StillWaterSA.cu:38-46: solid SA walls (the option set of the SABox mirror)
		SETUP_FRAMEWORK(
			kernel<WENDLAND>,
			formulation<SPH_F1>,
			viscosity<DYNAMICVISC>,
			boundary<SA_BOUNDARY>,
			periodicity<PERIODIC_NONE>,
			densitydiffusion<BREZZI>,
			add_flags<ENABLE_DTADAPT | ENABLE_DENSITY_SUM>
		);
	} else if (name == "ChannelIO") {   // src/problems/ChannelIO.cu:38-47: SA walls with open boundaries (the option set of the SAChannelIO mirror)
		SETUP_FRAMEWORK(
			kernel<WENDLAND>,
			formulation<SPH_F1>,
			viscosity<DYNAMICVISC>,
			boundary<SA_BOUNDARY>,
			periodicity<PERIODIC_NONE>,
			densitydiffusion<BREZZI>,
			add_flags<ENABLE_DTADAPT | ENABLE_INLET_OUTLET | ENABLE_DENSITY_SUM | ENABLE_WATER_DEPTH>
		);
	} else if (name == "StillWaterRepackSA") {   // src/problems/StillWaterRepackSA.cu:38-44: continuity equation, gamma by quadrature
		SETUP_FRAMEWORK(
			kernel<WENDLAND>,
			viscosity<DYNAMICVISC>,
			boundary<SA_BOUNDARY>,
			periodicity<PERIODIC_NONE>,
			add_flags<ENABLE_DTADAPT | ENABLE_REPACKING | ENABLE_GAMMA_QUADRATURE>
		);
	} else if (name == "StillWaterSAKeps") {   // StillWaterSA's options with turbulence_model<KEPSILON> (the GenericProblem selector, src/problems/GenericProblem.h:201)
		SETUP_FRAMEWORK(
			kernel<WENDLAND>,
			formulation<SPH_F1>,
			rheology<NEWTONIAN>,
			turbulence_model<KEPSILON>,
			boundary<SA_BOUNDARY>,
			periodicity<PERIODIC_NONE>,
			densitydiffusion<BREZZI>,
			add_flags<ENABLE_DTADAPT | ENABLE_DENSITY_SUM>
		);
	} else if (name == "CompleteSaExample") {   // src/problems/CompleteSaExample.cu:39-47: compiles and constructs; its physics is not built
		SETUP_FRAMEWORK(
			kernel<WENDLAND>,
			formulation<SPH_F1>,
			viscosity<DYNAMICVISC>,
			boundary<SA_BOUNDARY>,
			periodicity<PERIODIC_NONE>,
			densitydiffusion<BREZZI>,
			add_flags<ENABLE_INLET_OUTLET | ENABLE_DENSITY_SUM | ENABLE_MOVING_BODIES>
		);
	} else if (name == "PoiseuilleViscModel") {   // src/problems/Poiseuille.inc:102-119 (Newtonian), all four run-time selectors
		const DensityDiffusionType RHODIFF = (DensityDiffusionType)(int)num(c, "rhodiff");
		const ComputationalViscosityType compvisc = (ComputationalViscosityType)(int)num(c, "compvisc");
		const AverageOperator viscavg = (AverageOperator)(int)num(c, "viscavg");
		const ViscousModel viscmodel = (ViscousModel)(int)num(c, "viscmodel");
		SETUP_FRAMEWORK(
			kernel<WENDLAND>,
			rheology<NEWTONIAN>,
			turbulence_model<LAMINAR_FLOW>,
			computational_visc<KINEMATIC>,
			visc_model<MORRIS>,
			visc_average<ARITHMETIC>,
			periodicity<PERIODIC_XY>,
			boundary<DYN_BOUNDARY>
		).select_options
			( RHODIFF  // switch to the user-selected density diffusion
			, compvisc // switch to the user-selected computational viscosity
			, viscavg  // switch to the user-selected viscous averaging operator
			, viscmodel // switch to the user-selected viscous model
			);
	} else if (name == "BiFluidPoiseuilleDYN") {   // src/problems/BiFluidPoiseuille.inc:45-59 with the DYN defines of BiFluidPoiseuilleDYN.cu:27-28
		const DensityDiffusionType RHODIFF = (DensityDiffusionType)(int)num(c, "rhodiff");
		SETUP_FRAMEWORK(
			formulation<SPH_HA>,
			rheology<NEWTONIAN>,
			turbulence_model<LAMINAR_FLOW>,
			computational_visc<DYNAMIC>,
			visc_model<MORRIS>,
			visc_average<HARMONIC>,
			boundary<DYN_BOUNDARY>,
			periodicity<PERIODIC_XY>,
			add_flags<ENABLE_MULTIFLUID | ENABLE_DTADAPT>
			).select_options(
			RHODIFF == FERRARI, densitydiffusion<FERRARI>(),
			RHODIFF == BREZZI, densitydiffusion<BREZZI>(),
			RHODIFF == COLAGROSSI, densitydiffusion<COLAGROSSI>()
		);
	} else if (name == "AccuracyTest") {   // src/problems/AccuracyTest.cu:51-55
		SETUP_FRAMEWORK(
			viscosity<ARTVISC>,
			boundary<DYN_BOUNDARY>,
			add_flags<ENABLE_INTERNAL_ENERGY>
		);
	} else if (name == "DEMExample") {     // src/problems/DEMExample.cu:47-52
		const DensityDiffusionType rhodiff = (DensityDiffusionType)(int)num(c, "rhodiff");
		SETUP_FRAMEWORK(
			viscosity<ARTVISC>,
			boundary<LJ_BOUNDARY>,
			add_flags<ENABLE_DEM | ENABLE_PLANES>
		).select_options(
			rhodiff
		);
	} else if (name == "PoiseuillePapanastasiou") {   // src/problems/Poiseuille.inc:102-119 with POISEUILLE_RHEOLOGY = PAPANASTASIOU
		const DensityDiffusionType RHODIFF = (DensityDiffusionType)(int)num(c, "rhodiff");
		const ComputationalViscosityType compvisc = (ComputationalViscosityType)(int)num(c, "compvisc");
		const AverageOperator viscavg = (AverageOperator)(int)num(c, "viscavg");
		SETUP_FRAMEWORK(
			kernel<WENDLAND>,
			rheology<PAPANASTASIOU>,
			turbulence_model<LAMINAR_FLOW>,
			computational_visc<KINEMATIC>,
			visc_model<MORRIS>,
			visc_average<ARITHMETIC>,
			periodicity<PERIODIC_XY>,
			boundary<DYN_BOUNDARY>
		).select_options
			( RHODIFF  // switch to the user-selected density diffusion
			, compvisc // switch to the user-selected computational viscosity
			, viscavg  // switch to the user-selected viscous averaging operator
			);
	} else if (name == "GenericRuntime") { // every selector named (src/problems/GenericProblem.cu:55-65), kernel and periodicity chosen at run time
		simframework = CUDASimFramework<
			kernel<CUBICSPLINE>,
			formulation<SPH_F1>,
			densitydiffusion<COLAGROSSI>,
			rheology<NEWTONIAN>,
			turbulence_model<LAMINAR_FLOW>,
			computational_visc<DYNAMIC>,
			visc_model<MORRIS>,
			visc_average<GEOMETRIC>,
			boundary<DYN_BOUNDARY>,
			periodicity<PERIODIC_NONE>,
			add_flags<ENABLE_XSPH | ENABLE_MULTIFLUID>,
			disable_flags<ENABLE_DTADAPT>
		>().select_options(
			(KernelType)(int)num(c, "kernel"),
			(Periodicity)(int)num(c, "periodicity"));
	} else if (name == "Default") {
		simframework = CUDASimFramework<>();
	} else
		throw std::runtime_error("unknown framework case " + name);
	return simframework;
}


// ---- what a Problem's constructor and ProblemCore::initialize do to the parameter structures ----
static void configure_params(Case const& c, SimParams *sp, ProblemPhysParams &pp)
{
	const double deltap = num(c, "deltap");
	if (has(c, "kernelradius")) sp->set_kernel_radius(num(c, "kernelradius"));
	sp->set_smoothing(num(c, "sfactor"), deltap);   // ProblemCore::set_deltap -> set_smoothing, src/simparams.h:325-336
	sp->neiblistsize = (uint)num(c, "neiblistsize"); sp->neibboundpos = (uint)num(c, "neibboundpos");
	sp->dtadaptfactor = (float)num(c, "dtadaptfactor");
	sp->densityDiffCoeff = (float)num(c, "densityDiffCoeff");
	sp->repack_a = (float)num(c, "repack_a"); sp->repack_alpha = (float)num(c, "repack_alpha");
	if (has(c, "buildneibsfreq")) sp->buildneibsfreq = (uint)num(c, "buildneibsfreq");
	if (has(c, "numbodies")) { sp->numbodies = (uint)num(c, "numbodies", 0); sp->numforcesbodies = (uint)num(c, "numbodies", 1); }
	const size_t nfluids = (size_t)num(c, "nfluids");
	for (size_t f = 0; f < nfluids; ++f) {
		const std::string key = "fluid" + std::to_string(f);
		pp.add_fluid((float)num(c, key.c_str(), 0));
		pp.set_equation_of_state(f, (float)num(c, key.c_str(), 1), (float)num(c, key.c_str(), 2));
		const std::string kind = c.at(key).at(3);
		if (kind == "kin") pp.set_kinematic_visc(f, (float)num(c, key.c_str(), 4));
		else if (kind == "dyn") pp.set_dynamic_visc(f, (float)num(c, key.c_str(), 4));
		if (has(c, ("bulkvisc" + std::to_string(f)).c_str())) pp.set_bulk_visc(f, (float)num(c, ("bulkvisc" + std::to_string(f)).c_str()));
		// generalized Newtonian parameters (Poiseuille.inc:131-132 sets the yield strength; the others keep their defaults
		// unless the case says otherwise)
		const std::string rkey = "rheology" + std::to_string(f);     // yield strength, nonlinear parameter (NaN: keep), m (NaN: keep)
		if (has(c, rkey.c_str())) {
			pp.set_yield_strength(f, (float)num(c, rkey.c_str(), 0));
			const float nl = (float)num(c, rkey.c_str(), 1), m = (float)num(c, rkey.c_str(), 2);
			if (!std::isnan(nl)) { if (pp.is_exponential_rheology()) pp.set_visc_exponential_coeff(f, nl); else pp.set_visc_power_law(f, nl); }
			if (!std::isnan(m)) pp.set_visc_regularization_param(f, m);
		}
	}
	pp.gravity = make_float3((float)num(c, "gravity", 0), (float)num(c, "gravity", 1), (float)num(c, "gravity", 2));
	if (has(c, "artvisccoeff")) pp.set_artificial_visc((float)num(c, "artvisccoeff"));
	pp.epsartvisc = (float)num(c, "epsartvisc");
	pp.r0 = (float)num(c, "r0"); pp.dcoeff = (float)num(c, "dcoeff");
	pp.p1coeff = (float)num(c, "p1coeff"); pp.p2coeff = (float)num(c, "p2coeff");
	pp.smagfactor = (float)num(c, "smagfactor"); pp.kspsfactor = (float)num(c, "kspsfactor");
	pp.MK_K = (float)num(c, "MK_K"); pp.MK_d = (float)num(c, "MK_d"); pp.MK_beta = (float)num(c, "MK_beta");
	pp.partsurf = (float)num(c, "partsurf");
	if (c.count("epsinterface")) pp.epsinterface = (float)num(c, "epsinterface");
	if (has(c, "demparams")) {     // ProblemAPI<1>::computeDEMphysparams (src/problem_api/ProblemAPI_1.cc:1399-1418)
		pp.ewres = (float)num(c, "demparams", 0); pp.nsres = (float)num(c, "demparams", 1);
		pp.demdx = (float)num(c, "demparams", 2); pp.demdy = (float)num(c, "demparams", 3);
		pp.demdxdy = pp.demdx*pp.demdy; pp.demzmin = (float)num(c, "demparams", 4);
	}
	pp.epsxsph = (float)num(c, "epsxsph");
	// GPUSPH::setViscosityCoefficient (src/GPUSPH.cc:1481-1508), which runs between problem set-up and uploadConstants
	for (size_t f = 0; f < pp.numFluids(); ++f)
		pp.visccoeff[f] = sp->rheologytype == INVISCID ? NAN :
			(sp->rheologytype == NEWTONIAN && sp->compvisc == KINEMATIC) ? pp.kinematicvisc[f] : pp.visc_consistency[f];
	if (sp->viscmodel == ESPANOL_REVENGA)       // GPUSPH.cc:1511-1522
		for (size_t f = 0; f < pp.numFluids(); ++f) {
			if (std::isnan(pp.bulkvisc[f])) pp.bulkvisc[f] = 0;
			pp.visc2coeff[f] = pp.bulkvisc[f];
		}
}

struct GridSetup { float3 origin; uint3 gridSize; float3 cellSize; idx_t allocated; };
static GridSetup read_grid(Case const& c)
{
	GridSetup g;
	g.origin = make_float3((float)num(c, "origin", 0), (float)num(c, "origin", 1), (float)num(c, "origin", 2));
	g.gridSize = make_uint3((uint)num(c, "grid", 0), (uint)num(c, "grid", 1), (uint)num(c, "grid", 2));
	g.cellSize = make_float3((float)num(c, "cell", 0), (float)num(c, "cell", 1), (float)num(c, "cell", 2));
	g.allocated = (idx_t)num(c, "allocated");
	return g;
}

#endif // SPHX_PROBLEM_SETUP_H
