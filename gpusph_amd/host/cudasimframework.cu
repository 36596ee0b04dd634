// cudasimframework.cu -- the framework factory of the MI355X engines, under the file name GPUSPH's problems include
// (`#include "cudasimframework.cu"`, e.g. src/problems/DamBreak3D.cu:35).  Put this directory ahead of src/cuda in the
// include path and every problem's SETUP_FRAMEWORK(...) (src/ProblemCore.h:117) builds a SimFramework whose engines are the
// HIP*Engine classes of hip_engines.h instead of the CUDA*Engine templates of src/cuda/cudasimframework.cu:56-268.
//
// User-visible surface, same names and meaning as the reference (src/cuda/cudasimframework.cu:379-606):
//   selectors     kernel<>, formulation<>, densitydiffusion<>, rheology<>, turbulence_model<>, computational_visc<>,
//                 visc_model<>, visc_average<>, viscosity<LegacyViscosityType>, boundary<>, periodicity<>,
//                 add_flags<>, disable_flags<>
//   factory       CUDASimFramework<selectors...>  with  operator SimFramework*()  and
//                 .select_options(bool, selector [, ...]) / .select_options(option value [, ...])  (run-time overrides)
//   framework     static members kerneltype ... simflags, is_const_visc, ViscSpec read by SimParams' constructor template
//                 (src/simparams.h:261-274); PredCorrAllocPolicy + SimParams created with the engines
//                 (src/cuda/cudasimframework.cu:221-232); newFilterEngine / newPostProcessEngine (:236-268)
//
// How it differs inside: the reference resolves the named arguments through virtual multiple inheritance of up to twelve
// argument classes and instantiates every engine template for the chosen option set.  Here the option set is one type
// (OptionSet<...>) that the selectors rewrite one after the other, left to right, and the engines are not templates at
// all: the options reach libsphx as a run-time POD (sphx_params) filled by the engines' setconstants.  An option set the
// library has no kernels for (SA_BOUNDARY, SPH_GRENIER, k-epsilon, ...) still compiles, exactly like in the reference
// tree; the engines answer std::runtime_error("... not built") when it is uploaded.
#ifndef _CUDASIMFRAMEWORK_H
#define _CUDASIMFRAMEWORK_H

#include <memory>
#include <stdexcept>

#include "simframework.h"
#include "predcorr_alloc_policy.h"
#include "simflags.h"
#include "option_range.h"
#include "visc_spec.h"
#include "utils.h"   // round_up / div_up: reach problem sources through this file in the reference too (src/problems/OpenChannel.cu:81)

#include "hip_engines.h"

// part of the de-facto interface: the reference's file opens namespace std at file scope (src/cuda/cudasimframework.cu:53-55)
// and problem sources rely on it (e.g. the bare `cout` of src/problems/WaveTank.cu:117)
using namespace std;

// ---- the option set: one type carrying the twelve options --------------------------------------------------------------
template<
	KernelType _kerneltype,
	SPHFormulation _sph_formulation,
	DensityDiffusionType _densitydiffusiontype,
	RheologyType _rheologytype,
	TurbulenceModel _turbmodel,
	ComputationalViscosityType _compvisc,
	ViscousModel _viscmodel,
	AverageOperator _viscavgop,
	LegacyViscosityType _legacyvisctype,
	BoundaryType _boundarytype,
	Periodicity _periodicbound,
	flag_t _simflags>
struct OptionSet
{
	static constexpr KernelType kernel_v = _kerneltype;
	static constexpr SPHFormulation formulation_v = _sph_formulation;
	static constexpr DensityDiffusionType densitydiffusion_v = _densitydiffusiontype;
	static constexpr RheologyType rheology_v = _rheologytype;
	static constexpr TurbulenceModel turbulence_v = _turbmodel;
	static constexpr ComputationalViscosityType compvisc_v = _compvisc;
	static constexpr ViscousModel viscmodel_v = _viscmodel;
	static constexpr AverageOperator viscavg_v = _viscavgop;
	static constexpr LegacyViscosityType legacyvisc_v = _legacyvisctype;
	static constexpr BoundaryType boundary_v = _boundarytype;
	static constexpr Periodicity periodic_v = _periodicbound;
	static constexpr flag_t flags_v = _simflags;

#define SPHX_REBIND(name, Type, k, f, d, r, t, c, m, a, l, b, p, s) \
	template<Type v> using name = OptionSet<k, f, d, r, t, c, m, a, l, b, p, s>
	SPHX_REBIND(with_kernel, KernelType, v, _sph_formulation, _densitydiffusiontype, _rheologytype, _turbmodel, _compvisc, _viscmodel, _viscavgop, _legacyvisctype, _boundarytype, _periodicbound, _simflags);
	SPHX_REBIND(with_formulation, SPHFormulation, _kerneltype, v, _densitydiffusiontype, _rheologytype, _turbmodel, _compvisc, _viscmodel, _viscavgop, _legacyvisctype, _boundarytype, _periodicbound, _simflags);
	SPHX_REBIND(with_densitydiffusion, DensityDiffusionType, _kerneltype, _sph_formulation, v, _rheologytype, _turbmodel, _compvisc, _viscmodel, _viscavgop, _legacyvisctype, _boundarytype, _periodicbound, _simflags);
	SPHX_REBIND(with_rheology, RheologyType, _kerneltype, _sph_formulation, _densitydiffusiontype, v, _turbmodel, _compvisc, _viscmodel, _viscavgop, _legacyvisctype, _boundarytype, _periodicbound, _simflags);
	SPHX_REBIND(with_turbulence, TurbulenceModel, _kerneltype, _sph_formulation, _densitydiffusiontype, _rheologytype, v, _compvisc, _viscmodel, _viscavgop, _legacyvisctype, _boundarytype, _periodicbound, _simflags);
	SPHX_REBIND(with_compvisc, ComputationalViscosityType, _kerneltype, _sph_formulation, _densitydiffusiontype, _rheologytype, _turbmodel, v, _viscmodel, _viscavgop, _legacyvisctype, _boundarytype, _periodicbound, _simflags);
	SPHX_REBIND(with_viscmodel, ViscousModel, _kerneltype, _sph_formulation, _densitydiffusiontype, _rheologytype, _turbmodel, _compvisc, v, _viscavgop, _legacyvisctype, _boundarytype, _periodicbound, _simflags);
	SPHX_REBIND(with_viscavg, AverageOperator, _kerneltype, _sph_formulation, _densitydiffusiontype, _rheologytype, _turbmodel, _compvisc, _viscmodel, v, _legacyvisctype, _boundarytype, _periodicbound, _simflags);
	SPHX_REBIND(with_legacyvisc, LegacyViscosityType, _kerneltype, _sph_formulation, _densitydiffusiontype, _rheologytype, _turbmodel, _compvisc, _viscmodel, _viscavgop, v, _boundarytype, _periodicbound, _simflags);
	SPHX_REBIND(with_boundary, BoundaryType, _kerneltype, _sph_formulation, _densitydiffusiontype, _rheologytype, _turbmodel, _compvisc, _viscmodel, _viscavgop, _legacyvisctype, v, _periodicbound, _simflags);
	SPHX_REBIND(with_periodicity, Periodicity, _kerneltype, _sph_formulation, _densitydiffusiontype, _rheologytype, _turbmodel, _compvisc, _viscmodel, _viscavgop, _legacyvisctype, _boundarytype, v, _simflags);
	SPHX_REBIND(with_flags, flag_t, _kerneltype, _sph_formulation, _densitydiffusiontype, _rheologytype, _turbmodel, _compvisc, _viscmodel, _viscavgop, _legacyvisctype, _boundarytype, _periodicbound, v);
#undef SPHX_REBIND
};

// the defaults of the reference (TypeDefaults, src/cuda/cudasimframework.cu:346-360)
typedef OptionSet<WENDLAND, SPH_F1, DENSITY_DIFFUSION_NONE, INVISCID, ARTIFICIAL, KINEMATIC, MORRIS, ARITHMETIC,
	INVALID_VISCOSITY, LJ_BOUNDARY, PERIODIC_NONE, DEFAULT_FLAGS> DefaultOptionSet;

// ---- the named selectors: each rewrites an option set --------------------------------------------------------------------
template<KernelType v> struct kernel
{ template<typename O> using apply = typename O::template with_kernel<v>; };
template<SPHFormulation v> struct formulation
{ template<typename O> using apply = typename O::template with_formulation<v>; };
template<DensityDiffusionType v> struct densitydiffusion
{ template<typename O> using apply = typename O::template with_densitydiffusion<v>; };
template<RheologyType v> struct rheology
{ template<typename O> using apply = typename O::template with_rheology<v>; };
template<TurbulenceModel v> struct turbulence_model
{ template<typename O> using apply = typename O::template with_turbulence<v>; };
template<ComputationalViscosityType v> struct computational_visc
{ template<typename O> using apply = typename O::template with_compvisc<v>; };
template<ViscousModel v> struct visc_model
{ template<typename O> using apply = typename O::template with_viscmodel<v>; };
template<AverageOperator v> struct visc_average
{ template<typename O> using apply = typename O::template with_viscavg<v>; };
template<BoundaryType v> struct boundary
{ template<typename O> using apply = typename O::template with_boundary<v>; };
template<Periodicity v> struct periodicity
{ template<typename O> using apply = typename O::template with_periodicity<v>; };

// legacy viscosity names: the five viscous options at once (ConvertLegacyVisc, src/visc_spec.h:345-391), and the fact
// that a legacy name was used (KINEMATICVISC forces constant viscosity; Grenier + legacy name means harmonic averaging)
template<LegacyViscosityType v> struct viscosity
{
	typedef typename ConvertLegacyVisc<v>::type Spec;
	template<typename O> using apply = typename O
		::template with_legacyvisc<v>
		::template with_rheology<Spec::rheologytype>
		::template with_turbulence<Spec::turbmodel>
		::template with_compvisc<Spec::compvisc>
		::template with_viscmodel<Spec::viscmodel>
		::template with_viscavg<Spec::avgop>;
};

// flags are added to / removed from whatever the selectors to the left produced
template<flag_t flags> struct add_flags
{ template<typename O> using apply = typename O::template with_flags<(O::flags_v | flags)>; };
template<flag_t flags> struct disable_flags
{ template<typename O> using apply = typename O::template with_flags<DISABLE_FLAGS(O::flags_v, flags)>; };

// selector of an option type given as a value: selector_for<KernelType, WENDLAND> is kernel<WENDLAND>
template<typename Option, Option value> struct selector_for;
template<KernelType v> struct selector_for<KernelType, v> : kernel<v> {};
template<SPHFormulation v> struct selector_for<SPHFormulation, v> : formulation<v> {};
template<DensityDiffusionType v> struct selector_for<DensityDiffusionType, v> : densitydiffusion<v> {};
template<RheologyType v> struct selector_for<RheologyType, v> : rheology<v> {};
template<TurbulenceModel v> struct selector_for<TurbulenceModel, v> : turbulence_model<v> {};
template<ComputationalViscosityType v> struct selector_for<ComputationalViscosityType, v> : computational_visc<v> {};
template<ViscousModel v> struct selector_for<ViscousModel, v> : visc_model<v> {};
template<AverageOperator v> struct selector_for<AverageOperator, v> : visc_average<v> {};
template<BoundaryType v> struct selector_for<BoundaryType, v> : boundary<v> {};
template<Periodicity v> struct selector_for<Periodicity, v> : periodicity<v> {};

// left-to-right application of a list of selectors
template<typename O, typename... Selectors> struct ApplySelectors
{ typedef O type; };
template<typename O, typename First, typename... Rest> struct ApplySelectors<O, First, Rest...>
{ typedef typename ApplySelectors<typename First::template apply<O>, Rest...>::type type; };

// ---- the framework: option set -> SimFramework with HIP engines ----------------------------------------------------------
template<typename Options>
class HIPSimFrameworkImpl : public SimFramework
{
public:
	static const KernelType kerneltype = Options::kernel_v;
	static const SPHFormulation sph_formulation = Options::formulation_v;
	static const DensityDiffusionType densitydiffusiontype = Options::densitydiffusion_v;
	static const RheologyType rheologytype = Options::rheology_v;
	static const TurbulenceModel turbmodel = Options::turbulence_v;
	static const ComputationalViscosityType compvisc = Options::compvisc_v;
	static const ViscousModel viscmodel = Options::viscmodel_v;
	// Grenier's formulation with a legacy viscosity name keeps its historical harmonic averaging
	static const AverageOperator viscavgop =
		(Options::formulation_v == SPH_GRENIER && Options::legacyvisc_v != INVALID_VISCOSITY) ? HARMONIC : Options::viscavg_v;
	static const BoundaryType boundarytype = Options::boundary_v;
	static const Periodicity periodicbound = Options::periodic_v;
	static const flag_t simflags = Options::flags_v;
	// constant viscosity: by decree (KINEMATICVISC) or because there is one Newtonian fluid and no k-epsilon
	static const bool is_const_visc = (Options::legacyvisc_v == KINEMATICVISC) ||
		(IS_SINGLEFLUID(Options::flags_v) && Options::rheology_v == NEWTONIAN && Options::turbulence_v != KEPSILON);

	using ViscSpec = FullViscSpec<rheologytype, turbmodel, compvisc, viscmodel, viscavgop, simflags, is_const_visc>;

	// option combinations the reference refuses to compile (src/cuda/cudasimframework.cu:137-181), refused here as well,
	// one message per rule
	static_assert(!(Options::legacyvisc_v == KINEMATICVISC && IS_MULTIFLUID(Options::flags_v)),
		"viscosity<KINEMATICVISC> is a single-fluid specification");
	static_assert(!(turbmodel == KEPSILON && boundarytype != SA_BOUNDARY), "k-epsilon needs SA_BOUNDARY");
	static_assert(boundarytype != SA_BOUNDARY || viscmodel == MORRIS, "SA_BOUNDARY: only the MORRIS viscous model");
	static_assert(boundarytype != SA_BOUNDARY || Options::viscavg_v == ARITHMETIC || rheologytype == GRANULAR || sph_formulation == SPH_HA,
		"SA_BOUNDARY: only ARITHMETIC viscous averaging (except GRANULAR rheology / SPH_HA)");
	static_assert(boundarytype != SA_BOUNDARY || (turbmodel != SPS && turbmodel != ARTIFICIAL), "SA_BOUNDARY: no SPS, no artificial viscosity");
	static_assert(boundarytype != SA_BOUNDARY || kerneltype == WENDLAND, "SA_BOUNDARY: only the Wendland kernel has gamma formulas");
	static_assert(boundarytype != SA_BOUNDARY || sph_formulation != SPH_GRENIER, "SA_BOUNDARY: no SPH_GRENIER");
	static_assert(boundarytype != SA_BOUNDARY || !(simflags & (ENABLE_XSPH | ENABLE_DEM)), "SA_BOUNDARY: no XSPH, no DEM");
	static_assert(boundarytype != SA_BOUNDARY || !(simflags & ENABLE_INLET_OUTLET) || (simflags & ENABLE_DENSITY_SUM),
		"SA_BOUNDARY: open boundaries need ENABLE_DENSITY_SUM");
	static_assert(boundarytype != SA_BOUNDARY || !((simflags & ENABLE_DENSITY_SUM) && (simflags & ENABLE_GAMMA_QUADRATURE)),
		"SA_BOUNDARY: ENABLE_DENSITY_SUM excludes ENABLE_GAMMA_QUADRATURE");
	static_assert(boundarytype == SA_BOUNDARY || !(simflags & ENABLE_DENSITY_SUM), "ENABLE_DENSITY_SUM needs SA_BOUNDARY");
	static_assert(!(viscmodel == ESPANOL_REVENGA && rheologytype != NEWTONIAN), "ESPANOL_REVENGA: Newtonian fluids only");

private:
	HIPEngineContextPtr m_context;

public:
	HIPSimFrameworkImpl() : SimFramework(), m_context(std::make_shared<HIPEngineContext>())
	{
		m_neibsEngine = new HIPNeibsEngine(m_context);
		m_integrationEngine = new HIPPredCorrEngine(m_context);
		m_viscEngine = new HIPViscEngine(m_context);
		m_forcesEngine = new HIPForcesEngine(m_context);
		m_bcEngine = (boundarytype == SA_BOUNDARY) ? new HIPBoundaryConditionsEngine(m_context) : NULL;

		m_allocPolicy = std::make_shared<PredCorrAllocPolicy>();

		m_simparams = new SimParams(this);
	}

protected:
	AbstractFilterEngine* newFilterEngine(FilterType filtertype, int frequency)
	{
		switch (filtertype) {
		case SHEPARD_FILTER:
		case MLS_FILTER:
			return new HIPFilterEngine(m_context, filtertype, frequency);
		case INVALID_FILTER:
			throw std::runtime_error("Invalid filter type");
		}
		throw std::runtime_error("Unknown filter type");
	}

	AbstractPostProcessEngine* newPostProcessEngine(PostProcessType pptype, flag_t options = NO_FLAGS)
	{
		switch (pptype) {
		case VORTICITY:
		case TESTPOINTS:
		case SURFACE_DETECTION:
		case INTERFACE_DETECTION:
			return new HIPPostProcessEngine(m_context, pptype, options);
		case FLUX_COMPUTATION:
			throw std::runtime_error("FLUX_COMPUTATION (open boundaries): not built into libsphx");
		case CALC_PRIVATE:
			throw std::runtime_error("CALC_PRIVATE (problem-specific kernel): not built into libsphx");
		case INVALID_POSTPROC:
			throw std::runtime_error("Invalid filter type");
		}
		throw std::runtime_error("Unknown filter type");
	}
};

// ---- the factory problems name in SETUP_FRAMEWORK -------------------------------------------------------------------------
template<typename... Selectors>
class CUDASimFramework
{
	typedef typename ApplySelectors<DefaultOptionSet, Selectors...>::type Options;

	template<typename Extra>
	CUDASimFramework<Selectors..., Extra> extend() { return CUDASimFramework<Selectors..., Extra>(); }

	// run-time value of an option type -> the factory extended by the matching selector, by walking the option's range
	template<typename Option, Option check, bool in_range = is_in_range(check)>
	struct ValueWalk
	{
		template<typename... Rest>
		static SimFramework *go(CUDASimFramework &self, Option selector, Rest... rest)
		{
			if (selector == check)
				return self.template extend< selector_for<Option, check> >().select_options(rest...);
			return ValueWalk<Option, Option(check + 1)>::go(self, selector, rest...);
		}
	};
	template<typename Option, Option check>
	struct ValueWalk<Option, check, false>
	{
		template<typename... Rest>
		static SimFramework *go(CUDASimFramework&, Option, Rest...)
		{ throw std::runtime_error("invalid selector value"); }
	};

public:
	/// the framework itself: made when the factory is assigned to a SimFramework*
	operator SimFramework *() { return new HIPSimFrameworkImpl<Options>(); }

	/// end of a chain of run-time overrides
	SimFramework *select_options() { return *this; }

	/// override with Extra if the flag is true, then go on with the rest
	template<typename Extra, typename... Rest>
	SimFramework *select_options(bool selector, Extra, Rest... rest)
	{
		if (selector)
			return extend<Extra>().select_options(rest...);
		return this->select_options(rest...);
	}

	/// override the option of type Option with a value known at run time, then go on with the rest
	template<typename Option, typename... Rest>
	enable_if_t<option_range<Option>::defined, SimFramework *>
	select_options(Option selector, Rest... rest)
	{ return ValueWalk<Option, option_range<Option>::min>::go(*this, selector, rest...); }
};

#endif

/* vim: set ft=cuda sw=4 ts=4 : */
