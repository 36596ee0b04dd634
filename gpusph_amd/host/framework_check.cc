// framework_check.cc -- builds frameworks the way GPUSPH's problems do, through cudasimframework.cu of this directory and
// the reference's own headers (SimFramework, SimParams, PhysParams, option_range, visc_spec, ...), and prints what the
// HIP engines would upload.  tests/test_intree_boundary.py compares the output field by field with the Python mirrors
// (gpusph_amd/problem.py, params.py) that the GPU tests and bench.py drive the library with.
//
// Built only where the GPUSPH tree is present (host code, no device needed): see Makefile, target intree.
//   framework_check <case file>      one "key value..." pair per line, see problem_setup.h; JSON on stdout
#define GPUSPH_MAIN   // this translation unit holds the option-name tables, like src/GPUSPH.cc:38
#include "problem_setup.h"

static void hex(std::ostream &o, const void *p, size_t n)
{
	static const char *d = "0123456789abcdef";
	const unsigned char *b = (const unsigned char*)p;
	for (size_t i = 0; i < n; ++i) o << d[b[i] >> 4] << d[b[i] & 15];
}

int main(int argc, char **argv)
{
	if (argc < 2) { fprintf(stderr, "usage: %s <case file>\n", argv[0]); return 2; }
	try {
		const Case c = read_case(argv[1]);
		std::unique_ptr<SimFramework> fw(make_framework(c));
		SimParams *sp = fw->simparams();
		std::cout.precision(17);

		std::cout << "{\"options\": {"
			<< "\"kerneltype\": " << sp->kerneltype << ", \"sph_formulation\": " << sp->sph_formulation
			<< ", \"densitydiffusiontype\": " << sp->densitydiffusiontype << ", \"rheologytype\": " << sp->rheologytype
			<< ", \"turbmodel\": " << sp->turbmodel << ", \"compvisc\": " << sp->compvisc << ", \"viscmodel\": " << sp->viscmodel
			<< ", \"viscavgop\": " << sp->viscavgop << ", \"is_const_visc\": " << (sp->is_const_visc ? 1 : 0)
			<< ", \"boundarytype\": " << sp->boundarytype << ", \"periodicbound\": " << sp->periodicbound
			<< ", \"simflags\": " << sp->simflags << "}";
		std::cout << ", \"engines\": {\"neibs\": " << (fw->getNeibsEngine() != NULL) << ", \"forces\": " << (fw->getForcesEngine() != NULL)
			<< ", \"visc\": " << (fw->getViscEngine() != NULL) << ", \"integration\": " << (fw->getIntegrationEngine() != NULL)
			<< ", \"bc\": " << (fw->getBCEngine() != NULL) << ", \"alloc_policy\": " << (fw->getAllocPolicy() ? 1 : 0) << "}";

		if (has(c, "filter")) {            // Problem::addFilter -> SimFramework::addFilterEngine -> newFilterEngine
			AbstractFilterEngine *flt = fw->addFilterEngine((FilterType)(int)num(c, "filter", 0), (int)num(c, "filter", 1));
			std::cout << ", \"filter_frequency\": " << flt->frequency() << ", \"filters\": " << fw->getFilterEngines().size();
		}
		if (has(c, "postprocess")) {       // Problem::addPostProcess -> addPostProcessEngine -> newPostProcessEngine
			const PostProcessType pt = (PostProcessType)(int)num(c, "postprocess", 0);
			fw->addPostProcessEngine(pt, (flag_t)num(c, "postprocess", 1));
			AbstractPostProcessEngine *pp = fw->hasPostProcessEngine(pt);
			std::cout << ", \"pp_written\": " << pp->get_written_buffers() << ", \"pp_updated\": " << pp->get_updated_buffers();
		}

		if (has(c, "deltap")) {
			ProblemPhysParams pp(sp->rheologytype);      // ProblemCore::physparams(), src/ProblemCore.h:412-418
			configure_params(c, sp, pp);
			const GridSetup g = read_grid(c);
			sphx_params P;
			HIPEngineContext::fill_params(P, sp, &pp, g.origin, g.gridSize, g.cellSize, g.allocated);
			std::cout << ", \"slength\": " << sp->slength << ", \"influenceRadius\": " << sp->influenceRadius
				<< ", \"nlSqInfluenceRadius\": " << sp->nlSqInfluenceRadius << ", \"params_hex\": \"";
			hex(std::cout, &P, sizeof(P));
			std::cout << "\"";
		}
		std::cout << "}" << std::endl;
	} catch (std::exception const& e) {
		fprintf(stderr, "framework_check: %s\n", e.what());
		return 1;
	}
	return 0;
}
