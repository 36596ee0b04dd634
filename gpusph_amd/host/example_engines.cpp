// example_engines.cpp -- drives the four HIP*Engine adapters (sphx_host.h) through the command sequence
// GPUWorker runs for one device (src/GPUWorker.cc:1779-1905 CALCHASH/SORT/REORDER/BUILDNEIBS,
// :2188-2230 FORCES_SYNC with the blocking dtreduce, :2232-2270 EULER; predictor/corrector order of
// src/integrators/PredictorCorrectorIntegrator.cc:386-685).  Input: a problem dump written by
// tests/helpers (header + pos/vel/info/hash); output: pos/vel/info/hash after `steps` time steps.
//   example_engines <in.bin> <out.bin>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include "sphx_host.h"

struct DumpHeader {
	char magic[8];
	uint32_t n, alloc, steps, num_rb_particles;
	float dt, sspeed_cfl, nlSqInfluenceRadius;
	int32_t numforcesbodies;
	int32_t rb_cgGridPos[3];
	float rb_cgPos[3];
	int32_t rb_firstindex;
	int32_t filter_type, filter_freq;   // density filter (FilterType) every filter_freq iterations; freq 0 = none
	sphx_params params;
};

template<flag_t Key> static typename BufferTraits<Key>::element_type *alloc_buf(BufferList &bl, size_t n, int init = 0)
{
	bl.addBuffer<Key>(init);
	bl[Key]->alloc(n);
	bl[Key]->mark_valid();
	return static_cast<typename BufferTraits<Key>::element_type*>(bl[Key]->get_buffer());
}

int main(int argc, char **argv)
{
	if (argc < 3) { fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]); return 2; }
	FILE *f = fopen(argv[1], "rb");
	if (!f) { perror("open"); return 2; }
	DumpHeader h;
	if (fread(&h, sizeof(h), 1, f) != 1 || strncmp(h.magic, "SPHXDMP1", 8)) { fprintf(stderr, "bad dump\n"); return 2; }
	const uint32_t n0 = h.n, A = h.alloc;
	std::vector<float4> hpos(n0), hvel(n0);
	std::vector<particleinfo> hinfo(n0);
	std::vector<hashKey> hhash(n0);
	if (fread(hpos.data(), 16, n0, f) != n0 || fread(hvel.data(), 16, n0, f) != n0 ||
		fread(hinfo.data(), 8, n0, f) != n0 || fread(hhash.data(), 4, n0, f) != n0) { fprintf(stderr, "short dump\n"); return 2; }
	fclose(f);

	try {
		hip_throw(hipSetDevice(0), "hipSetDevice");
		const sphx_params &P = h.params;
		SimParams sp; PhysParams pp;
		sp.kerneltype = P.kerneltype; sp.sph_formulation = P.sph_formulation; sp.densitydiffusiontype = P.densitydiffusiontype;
		sp.boundarytype = P.boundarytype; sp.rheologytype = P.rheologytype; sp.turbmodel = P.turbmodel;
		sp.periodicbound = P.periodic; sp.simflags = P.simflags;
		sp.slength = P.slength; sp.kernelradius = P.kernelradius; sp.influenceRadius = P.influenceradius;
		sp.nlSqInfluenceRadius = h.nlSqInfluenceRadius; sp.dtadaptfactor = P.dtadaptfactor;
		sp.densityDiffCoeff = P.densityDiffCoeff; sp.neiblistsize = P.neiblistsize; sp.neibboundpos = P.neibboundpos;
		sp.deltap = P.deltap; sp.numforcesbodies = h.numforcesbodies;
		for (int a = 0; a < 3; ++a) sp.coord[a] = P.coord[a];
		for (uint32_t fl = 0; fl < P.numfluids; ++fl) {
			pp.rho0.push_back(P.rho0[fl]); pp.bcoeff.push_back(P.bcoeff[fl]); pp.gammacoeff.push_back(P.gammacoeff[fl]);
			pp.sscoeff.push_back(P.sscoeff[fl]); pp.sspowercoeff.push_back(P.sspowercoeff[fl]); pp.visccoeff.push_back(P.visccoeff[fl]);
		}
		pp.gravity = make_float3(P.gravity[0], P.gravity[1], P.gravity[2]);
		pp.artvisccoeff = P.artvisccoeff; pp.epsartvisc = P.epsartvisc;
		const float3 origin = make_float3(P.worldOrigin[0], P.worldOrigin[1], P.worldOrigin[2]);
		const uint3 gridSize = make_uint3(P.gridSize[0], P.gridSize[1], P.gridSize[2]);
		const float3 cellSize = make_float3(P.cellSize[0], P.cellSize[1], P.cellSize[2]);
		const uint gridCells = gridSize.x*gridSize.y*gridSize.z;

		HIPSimFramework fw;
		AbstractNeibsEngine *neibsEngine = fw.getNeibsEngine();
		AbstractForcesEngine *forcesEngine = fw.getForcesEngine();
		AbstractIntegrationEngine *integrationEngine = fw.getIntegrationEngine();
		// GPUWorker::uploadConstants (src/GPUWorker.cc:2989-3001)
		forcesEngine->setconstants(&sp, &pp, origin, gridSize, cellSize, A);
		integrationEngine->setconstants(&pp, origin, gridSize, cellSize, A, sp.neiblistsize, (float)sp.slength);
		neibsEngine->setconstants(&sp, &pp, origin, gridSize, cellSize, A);
		if (h.num_rb_particles) {
			const int3 g = make_int3(h.rb_cgGridPos[0], h.rb_cgGridPos[1], h.rb_cgGridPos[2]);
			const float3 c = make_float3(h.rb_cgPos[0], h.rb_cgPos[1], h.rb_cgPos[2]);
			forcesEngine->setrbcg(&g, &c, 1);
			integrationEngine->setrbcg(&g, &c, 1);   // each engine keeps its own copy (uploadForces/EulerBodiesCentersOfGravity)
			forcesEngine->setrbstart(&h.rb_firstindex, 1);
			const float ident[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
			const float3 z = make_float3(0, 0, 0);
			integrationEngine->setrbsteprot(ident, 1); integrationEngine->setrbtrans(&z, 1);
			integrationEngine->setrblinearvel(&z, 1); integrationEngine->setrbangularvel(&z, 1);
		}

		// buffers: state "n" (a) and "n*" / "sorted" (b) as in GPUWorker::m_dBuffers
		BufferList a, b, shared;
		float4 *posA = alloc_buf<BUFFER_POS>(a, A), *velA = alloc_buf<BUFFER_VEL>(a, A);
		alloc_buf<BUFFER_POS>(b, A); alloc_buf<BUFFER_VEL>(b, A);
		particleinfo *info = alloc_buf<BUFFER_INFO>(shared, A);
		hashKey *hash = alloc_buf<BUFFER_HASH>(shared, A);
		alloc_buf<BUFFER_PARTINDEX>(shared, A);
		alloc_buf<BUFFER_CELLSTART>(shared, gridCells, 0xFF); alloc_buf<BUFFER_CELLEND>(shared, gridCells, 0xFF);
		alloc_buf<BUFFER_NEIBSLIST>(shared, (size_t)sp.neiblistsize*A, 0xFF);
		alloc_buf<BUFFER_FORCES>(shared, A);
		alloc_buf<BUFFER_CFL>(shared, forcesEngine->getFmaxElements(A));
		alloc_buf<BUFFER_CFL_TEMP>(shared, std::max(4u, forcesEngine->getFmaxTempElements(forcesEngine->getFmaxElements(A))));
		if (h.num_rb_particles) { alloc_buf<BUFFER_RB_FORCES>(shared, h.num_rb_particles); alloc_buf<BUFFER_RB_TORQUES>(shared, h.num_rb_particles); }
		for (flag_t k : { BUFFER_INFO, BUFFER_HASH, BUFFER_PARTINDEX, BUFFER_CELLSTART, BUFFER_CELLEND, BUFFER_NEIBSLIST,
				BUFFER_FORCES, BUFFER_CFL, BUFFER_CFL_TEMP, BUFFER_RB_FORCES, BUFFER_RB_TORQUES })
			if (shared.has(k)) { a.add(k, shared[k]); b.add(k, shared[k]); }
		hip_throw(hipMemcpy(posA, hpos.data(), 16*(size_t)n0, hipMemcpyHostToDevice), "upload");
		hip_throw(hipMemcpy(velA, hvel.data(), 16*(size_t)n0, hipMemcpyHostToDevice), "upload");
		hip_throw(hipMemcpy(info, hinfo.data(), 8*(size_t)n0, hipMemcpyHostToDevice), "upload");
		hip_throw(hipMemcpy(hash, hhash.data(), 4*(size_t)n0, hipMemcpyHostToDevice), "upload");
		uint *d_newNum = nullptr;
		hip_throw(hipMalloc((void**)&d_newNum, 4), "hipMalloc");

		std::unique_ptr<AbstractFilterEngine> filterEngine;
		if (h.filter_freq > 0) filterEngine.reset(fw.newFilterEngine((FilterType)h.filter_type, h.filter_freq));

		BufferList *cur = &a, *oth = &b;
		uint n = n0;
		float dt = h.dt;
		for (uint32_t it = 0; it < h.steps; ++it) {
			if (it % sp.buildneibsfreq == 0) {
				if (it == 0) neibsEngine->fixHash(*cur, *cur, n); else neibsEngine->calcHash(*cur, *cur, n);
				neibsEngine->sort(*cur, *cur, n);
				shared[BUFFER_CELLSTART]->clobber(); shared[BUFFER_CELLEND]->clobber();
				neibsEngine->reorderDataAndFindCellStart(nullptr, *oth, *cur, n, d_newNum);
				std::swap(cur, oth);
				hip_throw(hipMemcpy(&n, d_newNum, 4, hipMemcpyDeviceToHost), "DOWNLOAD_NEWNUMPARTS");
				neibsEngine->resetinfo();
				shared[BUFFER_NEIBSLIST]->clobber();
				neibsEngine->buildNeibsList(*cur, *cur, n, n, gridCells, h.nlSqInfluenceRadius, h.nlSqInfluenceRadius);
				TimingInfo ti; neibsEngine->getinfo(ti);
				if (ti.hasTooManyNeibs >= 0) throw std::runtime_error("too many neighbours");
			}
			// FILTER_CALL + SWAP_STATE_BUFFERS(BUFFER_VEL) (src/integrators/PredictorCorrectorIntegrator.cc:831-859,1011-1041)
			if (filterEngine && it > 0 && it % filterEngine->frequency() == 0) {
				filterEngine->process(*cur, *oth, n, n, (float)sp.slength, (float)sp.influenceRadius);
				std::shared_ptr<AbstractBuffer> filtered = (*oth)[BUFFER_VEL], unfiltered = (*cur)[BUFFER_VEL];
				cur->add(BUFFER_VEL, filtered); oth->add(BUFFER_VEL, unfiltered);
			}
			float dts[2];
			for (int step = 1; step <= 2; ++step) {
				BufferList &state = (step == 1) ? *cur : *oth;      // forces on n (predictor) or n* (corrector)
				shared[BUFFER_FORCES]->clobber(); shared[BUFFER_CFL]->clobber();
				const uint nb = forcesEngine->basicstep(state, state, n, 0, n, sp.deltap, (float)sp.slength, sp.dtadaptfactor,
					(float)sp.influenceRadius, 0.0f, nullptr, 0, SIMULATE, step, dt, sp.numforcesbodies > 0);
				dts[step - 1] = forcesEngine->dtreduce((float)sp.slength, sp.dtadaptfactor, h.sspeed_cfl, 0.0f, state, state, nb, n);
				// EULER always reads step n, writes n* (src/integrators/PredictorCorrectorIntegrator.cc:587-609)
				integrationEngine->basicstep(*cur, *oth, n, n, step == 1 ? dt/2 : dt, step, 0.0f, (float)sp.slength,
					(float)sp.influenceRadius, SIMULATE);
			}
			std::swap(cur, oth);
			dt = std::min(dts[0], dts[1]);     // TIME_STEP_EPILOGUE (src/GPUSPH.cc:650-657)
		}
		// POSTPROCESS before a write (src/GPUWorker.cc runCommand<POSTPROCESS>): free-surface flags into INFO
		std::unique_ptr<AbstractPostProcessEngine> surf(fw.newPostProcessEngine(SURFACE_DETECTION));
		surf->setconstants(&sp, &pp, A);
		surf->process(*cur, *cur, n, n, 0, nullptr);
		hip_throw(hipDeviceSynchronize(), "sync");
		hip_throw(hipMemcpy(hpos.data(), cur->getData<BUFFER_POS>(), 16*(size_t)n, hipMemcpyDeviceToHost), "download");
		hip_throw(hipMemcpy(hvel.data(), cur->getData<BUFFER_VEL>(), 16*(size_t)n, hipMemcpyDeviceToHost), "download");
		hip_throw(hipMemcpy(hinfo.data(), info, 8*(size_t)n, hipMemcpyDeviceToHost), "download");
		hip_throw(hipMemcpy(hhash.data(), hash, 4*(size_t)n, hipMemcpyDeviceToHost), "download");
		FILE *o = fopen(argv[2], "wb");
		fwrite(&n, 4, 1, o); fwrite(&dt, 4, 1, o);
		fwrite(hpos.data(), 16, n, o); fwrite(hvel.data(), 16, n, o); fwrite(hinfo.data(), 8, n, o); fwrite(hhash.data(), 4, n, o);
		fclose(o);
		printf("example_engines: %u particles, %u steps, dt=%g\n", n, h.steps, dt);
	} catch (const std::exception &e) {
		fprintf(stderr, "example_engines: %s\n", e.what());
		return 1;
	}
	return 0;
}
