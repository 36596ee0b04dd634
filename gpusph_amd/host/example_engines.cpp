// example_engines.cpp -- one GPUWorker's command stream, through GPUSPH's own interfaces, onto the MI355X engines.
//
// Everything this program touches that is not named HIP* / sphx_* is the reference's: the framework comes out of
// SETUP_FRAMEWORK-style expressions (problem_setup.h) via cudasimframework.cu of this directory, the engines are reached
// as SimFramework::get*Engine() -> Abstract*Engine virtuals, buffers are BufferList / Buffer<Key> objects with the
// HIPBuffer allocation policy, parameters are the tree's SimParams / PhysParams.  The command order is the one the
// integrator sends to a worker:
//   neighbour phase   CALCHASH|fixHash, SORT, REORDER, BUILDNEIBS     src/Integrator.cc:94-250, src/GPUWorker.cc:1779-1905
//   filters           FILTER + swap of the VEL buffers                src/integrators/PredictorCorrectorIntegrator.cc:831-859
//   predictor/corr.   CALC_VISC, FORCES_SYNC (+ dtreduce), MOVE_BODIES uploads, EULER    :386-685, src/GPUWorker.cc:2188-2270
//   post-processing   before a write                                   src/GPUWorker.cc:2540-2590
// tests/test_gpu_parity.py runs it next to the Python driver (gpusph_amd/engine.py) on the same inputs: results bit-equal.
//
//   example_engines <case file> <state.bin> <out.bin>
#define GPUSPH_MAIN
#include <algorithm>
#include <cmath>
#include <cstring>
#include "problem_setup.h"

// a BufferList holding one freshly allocated device buffer (lists are then combined with operator|)
template<flag_t Key>
static BufferList one_buffer(size_t elems, int init = 0)
{
	BufferList l;
	l.addBuffer<HIPBuffer, Key>(init);
	l[Key]->alloc(elems);
	l.mark_valid();
	return l;
}

// ---- bodies with prescribed motion: the Chrono-free part of ProblemCore::bodies_timestep (src/ProblemCore.cc:484-610) ----
struct Kinematics { double crot[3], lvel[3], avel[3]; };
struct Body {
	int firstindex; uint numparts;
	std::string motion;             // "static" | "paddle_y"
	double amplitude, omega, tstart, tend;
	Kinematics initial, kdata, storage;
};

// WaveTank::moving_bodies_callback (src/problems/WaveTank.cu:221-243): rotation about y through the hinge, first-order
// quaternion step dr = normalize(1 + dt/2 (0, avel))
static void body_callback(Body &b, double t0, double t1, double dx[3], double dr[9])
{
	const double ident[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
	std::memcpy(dr, ident, sizeof(ident));
	dx[0] = dx[1] = dx[2] = 0;
	Kinematics &kd = b.kdata;
	kd.lvel[0] = kd.lvel[1] = kd.lvel[2] = 0;
	kd.avel[0] = kd.avel[1] = kd.avel[2] = 0;
	if (b.motion == "paddle_y" && b.tstart < t1 && t1 < b.tend) {
		const double w = b.amplitude*b.omega*sin(b.omega*(t1 - b.tstart));
		kd.avel[1] = w;
		double e0 = 1.0, e2 = 0.5*(t1 - t0)*w;
		const double nrm = hypot(e0, e2);
		e0 /= nrm; e2 /= nrm;
		const double c = e0*e0 - e2*e2, s = 2.0*e0*e2;
		dr[0] = c; dr[2] = s; dr[6] = -s; dr[8] = c;
	}
}

int main(int argc, char **argv)
{
	if (argc < 4) { fprintf(stderr, "usage: %s <case file> <state.bin> <out.bin>\n", argv[0]); return 2; }
	try {
		const Case c = read_case(argv[1]);
		sphx_throw(sphx_set_device(0));                                   // checkCUDA / cudaSetDevice of the worker thread
		std::unique_ptr<SimFramework> fw(make_framework(c));
		SimParams *sp = fw->simparams();
		ProblemPhysParams pp(sp->rheologytype);
		configure_params(c, sp, pp);
		const GridSetup g = read_grid(c);
		const uint A = (uint)g.allocated;
		const uint gridCells = g.gridSize.x*g.gridSize.y*g.gridSize.z;
		const double origin[3] = { num(c, "origin", 0), num(c, "origin", 1), num(c, "origin", 2) };
		const double cell[3] = { num(c, "cell", 0), num(c, "cell", 1), num(c, "cell", 2) };
		const float deltap = (float)num(c, "deltap");
		const float slength = (float)sp->slength, influenceRadius = (float)sp->influenceRadius;
		const float sqNlRadius = (float)sp->nlSqInfluenceRadius;
		const uint steps = (uint)num(c, "steps");
		const float sspeed_cfl = (float)num(c, "sspeed_cfl");
		float max_kinvisc = (float)num(c, "max_kinvisc");     // GPUWorker::m_max_kinvisc: constant, or what calc_visc returns

		// ---- initial particle state ----
		FILE *f = fopen(argv[2], "rb");
		if (!f) throw std::runtime_error(std::string("cannot open ") + argv[2]);
		uint n0 = 0;
		if (fread(&n0, 4, 1, f) != 1 || n0 > A) throw std::runtime_error("bad state file");
		std::vector<float4> hpos(n0), hvel(n0);
		std::vector<particleinfo> hinfo(n0);
		std::vector<hashKey> hhash(n0);
		if (fread(hpos.data(), 16, n0, f) != n0 || fread(hvel.data(), 16, n0, f) != n0 ||
			fread(hinfo.data(), 8, n0, f) != n0 || fread(hhash.data(), 4, n0, f) != n0) throw std::runtime_error("short state file");
		fclose(f);

		AbstractNeibsEngine *neibsEngine = fw->getNeibsEngine();
		AbstractForcesEngine *forcesEngine = fw->getForcesEngine();
		AbstractViscEngine *viscEngine = fw->getViscEngine();
		AbstractIntegrationEngine *integrationEngine = fw->getIntegrationEngine();

		// GPUWorker::uploadConstants (src/GPUWorker.cc:2989-3001)
		forcesEngine->setconstants(sp, &pp, g.origin, g.gridSize, g.cellSize, A);
		integrationEngine->setconstants(&pp, g.origin, g.gridSize, g.cellSize, A, sp->neiblistsize, slength);
		neibsEngine->setconstants(sp, &pp, g.origin, g.gridSize, g.cellSize, A);

		// GPUWorker::uploadPlanes
		if (has(c, "plane")) {
			PlaneList planes;
			const size_t np = c.at("plane").size()/9;
			for (size_t k = 0; k < np; ++k)
				planes.push_back(make_plane(
					make_float3((float)num(c, "plane", 9*k), (float)num(c, "plane", 9*k + 1), (float)num(c, "plane", 9*k + 2)),
					make_int3((int)num(c, "plane", 9*k + 3), (int)num(c, "plane", 9*k + 4), (int)num(c, "plane", 9*k + 5)),
					make_float3((float)num(c, "plane", 9*k + 6), (float)num(c, "plane", 9*k + 7), (float)num(c, "plane", 9*k + 8))));
			forcesEngine->setplanes(planes);
		}

		// GPUWorker::allocateDeviceBuffers -> setDEM (src/GPUWorker.cc:1070-1076): "dem <ncols> <nrows> <heights, row-major>"
		if (has(c, "dem")) {
			const int ncols = (int)num(c, "dem", 0), nrows = (int)num(c, "dem", 1);
			std::vector<float> hdem((size_t)ncols*nrows);
			for (size_t k = 0; k < hdem.size(); ++k) hdem[k] = (float)num(c, "dem", 2 + k);
			forcesEngine->setDEM(hdem.data(), ncols, nrows);
		}

		// ---- bodies ----
		std::vector<Body> bodies;
		uint numBodiesParticles = 0;
		if (has(c, "body")) {
			const size_t per = 10, nb = c.at("body").size()/per;   // firstindex numparts cgx cgy cgz motion amplitude omega tstart tend
			for (size_t k = 0; k < nb; ++k) {
				Body b;
				b.firstindex = (int)num(c, "body", per*k); b.numparts = (uint)num(c, "body", per*k + 1);
				for (int a = 0; a < 3; ++a) { b.initial.crot[a] = num(c, "body", per*k + 2 + a); b.initial.lvel[a] = b.initial.avel[a] = 0; }
				b.motion = c.at("body").at(per*k + 5);
				b.amplitude = num(c, "body", per*k + 6); b.omega = num(c, "body", per*k + 7);
				b.tstart = num(c, "body", per*k + 8); b.tend = num(c, "body", per*k + 9);
				b.kdata = b.storage = b.initial;
				numBodiesParticles += b.numparts;
				bodies.push_back(b);
			}
		}
		const int numbodies = (int)bodies.size();
		std::vector<int3> cgGrid(std::max(numbodies, 1));
		std::vector<float3> cgPos(std::max(numbodies, 1)), trans(std::max(numbodies, 1)), lvel(std::max(numbodies, 1)), avel(std::max(numbodies, 1));
		std::vector<float> steprot(9*std::max(numbodies, 1));
		// calc_grid_and_local_pos of the centres of rotation (src/ProblemCore.cc:1560-1583)
		auto grid_and_local = [&](const double x[3], int3 &gp, float3 &lp) {
			int gi[3]; float li[3];
			const uint gs[3] = { g.gridSize.x, g.gridSize.y, g.gridSize.z };
			for (int a = 0; a < 3; ++a) {
				long v = (long)floor((x[a] - origin[a])/cell[a]);
				v = std::max(0L, std::min(v, (long)gs[a] - 1));
				gi[a] = (int)v;
				li[a] = (float)(x[a] - origin[a] - (v + 0.5)*cell[a]);
			}
			gp = make_int3(gi[0], gi[1], gi[2]); lp = make_float3(li[0], li[1], li[2]);
		};
		if (numbodies) {
			std::vector<int> first(numbodies);
			for (int b = 0; b < numbodies; ++b) { grid_and_local(bodies[b].initial.crot, cgGrid[b], cgPos[b]); first[b] = bodies[b].firstindex; }
			integrationEngine->setrbcg(cgGrid.data(), cgPos.data(), numbodies);     // uploadEulerBodiesCentersOfGravity
			forcesEngine->setrbcg(cgGrid.data(), cgPos.data(), numbodies);          // uploadForcesBodiesCentersOfGravity
			forcesEngine->setrbstart(first.data(), numbodies);
			for (int b = 0; b < numbodies; ++b) {
				trans[b] = lvel[b] = avel[b] = make_float3(0, 0, 0);
				for (int a = 0; a < 9; ++a) steprot[9*b + a] = (a % 4 == 0) ? 1.0f : 0.0f;
			}
			integrationEngine->setrbtrans(trans.data(), numbodies); integrationEngine->setrbsteprot(steprot.data(), numbodies);
			integrationEngine->setrblinearvel(lvel.data(), numbodies); integrationEngine->setrbangularvel(avel.data(), numbodies);
		}
		const bool moving = std::any_of(bodies.begin(), bodies.end(), [](Body const& b) { return b.motion != "static"; });

		// ---- device buffers: double-buffered POS / VEL ("step n", "step n*"), the rest shared (GPUWorker::m_dBuffers) ----
		BufferList posA = one_buffer<BUFFER_POS>(A), posB = one_buffer<BUFFER_POS>(A);
		BufferList velA = one_buffer<BUFFER_VEL>(A), velB = one_buffer<BUFFER_VEL>(A);
		BufferList shared = one_buffer<BUFFER_INFO>(A) | one_buffer<BUFFER_HASH>(A) | one_buffer<BUFFER_PARTINDEX>(A) |
			one_buffer<BUFFER_CELLSTART>(gridCells, 0xFF) | one_buffer<BUFFER_CELLEND>(gridCells, 0xFF) |
			one_buffer<BUFFER_NEIBSLIST>((size_t)sp->neiblistsize*A, 0xFF) | one_buffer<BUFFER_FORCES>(A) |
			one_buffer<BUFFER_CFL>(forcesEngine->getFmaxElements(A)) |
			one_buffer<BUFFER_CFL_TEMP>(std::max(4u, forcesEngine->getFmaxTempElements(forcesEngine->getFmaxElements(A))));
		if (numBodiesParticles)
			shared |= one_buffer<BUFFER_RB_FORCES>(numBodiesParticles) | one_buffer<BUFFER_RB_TORQUES>(numBodiesParticles);
		if (sp->turbmodel == SPS)
			shared |= one_buffer<BUFFER_TAU>(A) | one_buffer<BUFFER_SPS_TURBVISC>(A);
		if (sp->simflags & ENABLE_XSPH)
			shared |= one_buffer<BUFFER_XSPH>(A);
		if (NEEDS_EFFECTIVE_VISC(sp->rheologytype))
			shared |= one_buffer<BUFFER_EFFVISC>(A);
		sphx_throw(sphx_memcpy_h2d(posA.getData<BUFFER_POS>(), hpos.data(), 16*(size_t)n0));
		sphx_throw(sphx_memcpy_h2d(velA.getData<BUFFER_VEL>(), hvel.data(), 16*(size_t)n0));
		sphx_throw(sphx_memcpy_h2d(shared.getData<BUFFER_INFO>(), hinfo.data(), 8*(size_t)n0));
		sphx_throw(sphx_memcpy_h2d(shared.getData<BUFFER_HASH>(), hhash.data(), 4*(size_t)n0));
		uint *d_newNum = NULL;
		sphx_throw(sphx_malloc((void**)&d_newNum, 4));

		// Problem::addFilter (SimFramework::addFilterEngine keeps them ordered by type)
		if (has(c, "filter"))
			for (size_t k = 0; k + 1 < c.at("filter").size(); k += 2)
				fw->addFilterEngine((FilterType)(int)num(c, "filter", k), (int)num(c, "filter", k + 1));

		// ---- SA_BOUNDARY: the neighbour phase with the vertex / segment buffers and the initialisation sequence of the
		// boundary conditions (initializeBoundaryConditionsSequence<SA_BOUNDARY>, PredictorCorrectorIntegrator.cc:117-290);
		// then `steps` predictor-corrector steps: forces, Euler, INTEGRATE_GAMMA, boundary conditions (:386-685), for the option sets
		// whose SA forces are built (continuity equation, gamma by quadrature); no neighbour rebuild in between
		if (sp->boundarytype == SA_BOUNDARY) {
			std::vector<vertexinfo> hvert(n0);
			std::vector<float4> hbe(n0), hgg(n0);
			FILE *fs = fopen(argv[2], "rb");
			fseek(fs, 4 + (long)n0*(16 + 16 + 8 + 4), SEEK_SET);
			if (fread(hvert.data(), 16, n0, fs) != n0 || fread(hbe.data(), 16, n0, fs) != n0 || fread(hgg.data(), 16, n0, fs) != n0)
				throw std::runtime_error("short state file (SA buffers)");
			fclose(fs);
			BufferList saA = one_buffer<BUFFER_VERTICES>(A) | one_buffer<BUFFER_BOUNDELEMENTS>(A) | one_buffer<BUFFER_GRADGAMMA>(A);
			BufferList saB = one_buffer<BUFFER_VERTICES>(A) | one_buffer<BUFFER_BOUNDELEMENTS>(A) | one_buffer<BUFFER_GRADGAMMA>(A);
			shared |= one_buffer<BUFFER_VERTPOS>(A);
			const bool dynamic_gamma = USING_DYNAMIC_GAMMA(sp->simflags), density_sum = (sp->simflags & ENABLE_DENSITY_SUM) != 0;
			if (dynamic_gamma)      // per particle + per block behind round_up(numParticles, 4) (src/cuda/forces.cu:576-581)
				shared |= one_buffer<BUFFER_CFL_GAMMA>((size_t)A + 4 + forcesEngine->getFmaxElements(A));
			sphx_throw(sphx_memcpy_h2d(saA.getData<BUFFER_VERTICES>(), hvert.data(), 16*(size_t)n0));
			sphx_throw(sphx_memcpy_h2d(saA.getData<BUFFER_BOUNDELEMENTS>(), hbe.data(), 16*(size_t)n0));
			sphx_throw(sphx_memcpy_h2d(saA.getData<BUFFER_GRADGAMMA>(), hgg.data(), 16*(size_t)n0));
			// turbulence_model<KEPSILON>: BUFFER_TKE / EPSILON / TURBVISC / EULERVEL are part of both states (GPUWorker.cc:158-190),
			// BUFFER_DKDE and BUFFER_CFL_KEPS are written by the forces; uniform start (ProblemCore::init_keps, init_turbvisc: the case
			// file carries the three values)
			const bool keps = sp->turbmodel == KEPSILON;
			if (keps) {
				saA |= one_buffer<BUFFER_TKE>(A) | one_buffer<BUFFER_EPSILON>(A) | one_buffer<BUFFER_TURBVISC>(A) | one_buffer<BUFFER_EULERVEL>(A);
				saB |= one_buffer<BUFFER_TKE>(A) | one_buffer<BUFFER_EPSILON>(A) | one_buffer<BUFFER_TURBVISC>(A) | one_buffer<BUFFER_EULERVEL>(A);
				shared |= one_buffer<BUFFER_DKDE>(A) | one_buffer<BUFFER_CFL_KEPS>(forcesEngine->getFmaxElements(A));
				std::vector<float> hk(n0, (float)num(c, "keps0", 0)), he(n0, (float)num(c, "keps0", 1)), hn(n0, (float)num(c, "keps0", 2));
				std::vector<float4> hev(n0, make_float4(0, 0, 0, 0));
				sphx_throw(sphx_memcpy_h2d(saA.getData<BUFFER_TKE>(), hk.data(), 4*(size_t)n0));
				sphx_throw(sphx_memcpy_h2d(saA.getData<BUFFER_EPSILON>(), he.data(), 4*(size_t)n0));
				sphx_throw(sphx_memcpy_h2d(saA.getData<BUFFER_TURBVISC>(), hn.data(), 4*(size_t)n0));
				sphx_throw(sphx_memcpy_h2d(saA.getData<BUFFER_EULERVEL>(), hev.data(), 16*(size_t)n0));
			}
			uint n = n0;
			BufferList unsorted = posA | velA | saA | shared, sorted = posB | velB | saB | shared;
			neibsEngine->fixHash(unsorted, unsorted, n);
			neibsEngine->sort(unsorted, unsorted, n);
			shared[BUFFER_CELLSTART]->clobber(); shared[BUFFER_CELLEND]->clobber();
			neibsEngine->reorderDataAndFindCellStart(NULL, sorted, unsorted, n, d_newNum);
			sphx_throw(sphx_memcpy_d2h(&n, d_newNum, 4));
			neibsEngine->resetinfo();
			shared[BUFFER_NEIBSLIST]->clobber();
			BufferList state = posB | velB | saB | shared;
			// GPUWorker::runCommand<BUILDNEIBS> (src/GPUWorker.cc:1890): boundary elements are searched a little farther out
			const float boundNlSqInflRad = powf(sqrt(sp->nlSqInfluenceRadius) + sp->slength/sp->sfactor/2.0f, 2.0f);
			neibsEngine->buildNeibsList(state, state, n, n, gridCells, sqNlRadius, boundNlSqInflRad);
			TimingInfo ti; neibsEngine->getinfo(ti);
			if (ti.hasTooManyNeibs >= 0) throw std::runtime_error("too many neighbours");
			AbstractBoundaryConditionsEngine *bc = fw->getBCEngine();
			if (!bc) throw std::runtime_error("no boundary conditions engine loaded");
			bc->computeVertexNormal(state, state, n, n);
			bc->saInitGamma(state, state, slength, influenceRadius, deltap, sp->epsilon, n, n);
			bc->saSegmentBoundaryConditions(state, state, n, n, deltap, slength, influenceRadius, 0, SIMULATE);
			uint newNum = n;
			bc->saVertexBoundaryConditions(state, state, n, n, deltap, slength, influenceRadius, 0, false, 0.0f, &newNum, 0, 1, n, SIMULATE);
			// the boundary elements and vertex ids do not change: both states of the integrator see the same ones
			sphx_throw(sphx_memcpy_d2d(saA.getData<BUFFER_VERTICES>(), as_const(saB).getData<BUFFER_VERTICES>(), 16*(size_t)n));
			sphx_throw(sphx_memcpy_d2d(saA.getData<BUFFER_BOUNDELEMENTS>(), as_const(saB).getData<BUFFER_BOUNDELEMENTS>(), 16*(size_t)n));
			if (keps)       // the boundary conditions wrote wall rows of k / epsilon / eulerVel in state B: state A starts from the same fields
				for (flag_t b : { BUFFER_TKE, BUFFER_EPSILON, BUFFER_TURBVISC, BUFFER_EULERVEL })
					sphx_throw(sphx_memcpy_d2d(saA[b]->get_buffer(0), as_const(saB)[b]->get_buffer(0), (size_t)n*saB[b]->get_element_size()));
			BufferList stN = posB | velB | saB | shared, stS = posA | velA | saA | shared;
			BufferList *cur = &stN, *oth = &stS;
			float sdt = (float)num(c, "dt0");
			double st = 0;
			for (uint it = 0; it < steps; ++it) {
				float dts[2];
				for (int step = 1; step <= 2; ++step) {
					BufferList &rd = (step == 1) ? *cur : *oth;     // forces on step n (predictor) or n* (corrector)
					shared[BUFFER_FORCES]->clobber(); shared[BUFFER_CFL]->clobber();
					const uint nb = forcesEngine->basicstep(rd, rd, n, 0, n, deltap, slength, sp->dtadaptfactor, influenceRadius,
						sp->epsilon, NULL, 0, SIMULATE, step, sdt, false);
					dts[step - 1] = forcesEngine->dtreduce(slength, sp->dtadaptfactor, sspeed_cfl, max_kinvisc, rd, rd, nb, n);
					const float hdt = step == 1 ? sdt/2 : sdt;
					integrationEngine->basicstep(*cur, *oth, n, n, hdt, step, (float)st, slength, influenceRadius, SIMULATE);
					if (density_sum) {     // DENSITY_SUM, then CALC_ / APPLY_DENSITY_DIFFUSION on the new state (:607-659)
						integrationEngine->density_sum(*cur, *oth, n, n, hdt, step, (float)st, sp->epsilon, deltap, slength, influenceRadius);
						if (sp->densitydiffusiontype != DENSITY_DIFFUSION_NONE) {
							forcesEngine->compute_density_diffusion(*oth, *oth, n, n, deltap, slength, influenceRadius, hdt);
							integrationEngine->apply_density_diffusion(*oth, *oth, n, n, hdt);
						}
					} else
						integrationEngine->integrate_gamma(*cur, *oth, n, n, hdt, step, (float)st, sp->epsilon, slength, influenceRadius, SIMULATE);
					bc->saSegmentBoundaryConditions(*oth, *oth, n, n, deltap, slength, influenceRadius, step, SIMULATE);
					bc->saVertexBoundaryConditions(*oth, *oth, n, n, deltap, slength, influenceRadius, step, false, hdt, &newNum, 0, 1, n, SIMULATE);
				}
				std::swap(cur, oth);
				st += sdt;
				sdt = std::min(dts[0], dts[1]);
			}
			sphx_throw(sphx_device_synchronize());
			const BufferList &cs = *cur;
			hpos.resize(n); hvel.resize(n); hinfo.resize(n); hhash.resize(n); hvert.resize(n); hbe.resize(n); hgg.resize(n);
			sphx_throw(sphx_memcpy_d2h(hpos.data(), cs.getData<BUFFER_POS>(), 16*(size_t)n));
			sphx_throw(sphx_memcpy_d2h(hvel.data(), cs.getData<BUFFER_VEL>(), 16*(size_t)n));
			sphx_throw(sphx_memcpy_d2h(hinfo.data(), cs.getData<BUFFER_INFO>(), 8*(size_t)n));
			sphx_throw(sphx_memcpy_d2h(hhash.data(), cs.getData<BUFFER_HASH>(), 4*(size_t)n));
			sphx_throw(sphx_memcpy_d2h(hvert.data(), cs.getData<BUFFER_VERTICES>(), 16*(size_t)n));
			sphx_throw(sphx_memcpy_d2h(hbe.data(), cs.getData<BUFFER_BOUNDELEMENTS>(), 16*(size_t)n));
			sphx_throw(sphx_memcpy_d2h(hgg.data(), cs.getData<BUFFER_GRADGAMMA>(), 16*(size_t)n));
			std::vector<float2> hvp(3*(size_t)n);
			const float2 * const *vp = cs.getRawPtr<BUFFER_VERTPOS>();
			for (int k = 0; k < 3; ++k) sphx_throw(sphx_memcpy_d2h(hvp.data() + (size_t)k*n, vp[k], 8*(size_t)n));
			sphx_throw(sphx_free(d_newNum));
			FILE *o = fopen(argv[3], "wb");
			if (!o) throw std::runtime_error(std::string("cannot write ") + argv[3]);
			fwrite(&n, 4, 1, o); fwrite(&sdt, 4, 1, o); fwrite(&st, 8, 1, o);
			fwrite(hpos.data(), 16, n, o); fwrite(hvel.data(), 16, n, o); fwrite(hinfo.data(), 8, n, o); fwrite(hhash.data(), 4, n, o);
			fwrite(hvert.data(), 16, n, o); fwrite(hbe.data(), 16, n, o); fwrite(hgg.data(), 16, n, o); fwrite(hvp.data(), 8, 3*(size_t)n, o);
			const int32_t counters[4] = { (int32_t)ti.numInteractions, (int32_t)ti.maxFluidBoundaryNeibs, (int32_t)ti.maxVertexNeibs, (int32_t)newNum };
			fwrite(counters, 4, 4, o);
			if (keps) {      // k, epsilon, eddy viscosity behind the counters
				std::vector<float> hk(n);
				for (flag_t b : { BUFFER_TKE, BUFFER_EPSILON, BUFFER_TURBVISC }) {
					sphx_throw(sphx_memcpy_d2h(hk.data(), cs[b]->get_buffer(0), 4*(size_t)n));
					fwrite(hk.data(), 4, n, o);
				}
			}
			fclose(o);
			printf("example_engines: %s, %u particles, SA initialisation + %u steps, max vertex neighbours %d, t=%g dt=%g\n",
				str(c, "framework").c_str(), n, steps, (int)ti.maxVertexNeibs, st, sdt);
			return 0;
		}

		BufferList *curPos = &posA, *othPos = &posB, *curVel = &velA, *othVel = &velB;
		// SPH_GRENIER: BUFFER_VOLUME is part of the particle state (double buffered, re-sorted), BUFFER_SIGMA is ephemeral
		// (GPUWorker.cc:195-198); the volumes start from mass/density (ProblemCore::init_volume, src/ProblemCore.cc:1586-1606)
		const bool grenier = sp->sph_formulation == SPH_GRENIER;
		BufferList volA, volB, noVol;
		if (grenier) {
			volA = one_buffer<BUFFER_VOLUME>(A); volB = one_buffer<BUFFER_VOLUME>(A);
			shared |= one_buffer<BUFFER_SIGMA>(A);
			std::vector<float4> hvol(n0);
			for (uint i = 0; i < n0; ++i) {
				const float v = hpos[i].w/((hvel[i].w + 1.0f)*pp.rho0[fluid_num(hinfo[i])]);
				hvol[i] = make_float4(v, 0.0f, 0.0f, v);
			}
			sphx_throw(sphx_memcpy_h2d(volA.getData<BUFFER_VOLUME>(), hvol.data(), 16*(size_t)n0));
		}
		// ENABLE_INTERNAL_ENERGY: BUFFER_INTERNAL_ENERGY travels with the particle state like the volumes (zero at the start,
		// ProblemCore::init_internal_energy), BUFFER_INTERNAL_ENERGY_UPD is written by every forces pass (GPUWorker.cc:208-211)
		const bool energy = (sp->simflags & ENABLE_INTERNAL_ENERGY) != 0;
		if (energy) {
			if (grenier) throw std::runtime_error("internal energy with SPH_GRENIER is not built");
			volA = one_buffer<BUFFER_INTERNAL_ENERGY>(A); volB = one_buffer<BUFFER_INTERNAL_ENERGY>(A);
			shared |= one_buffer<BUFFER_INTERNAL_ENERGY_UPD>(A);
		}
		BufferList *curVol = (grenier || energy) ? &volA : &noVol, *othVol = (grenier || energy) ? &volB : &noVol;
		uint n = n0;
		float dt = (float)num(c, "dt0");
		double t = 0;
		for (uint it = 0; it < steps; ++it) {
			if (it % sp->buildneibsfreq == 0) {
				BufferList unsorted = *curPos | *curVel | *curVol | shared, sorted = *othPos | *othVel | *othVol | shared;
				if (it == 0) neibsEngine->fixHash(unsorted, unsorted, n); else neibsEngine->calcHash(unsorted, unsorted, n);
				neibsEngine->sort(unsorted, unsorted, n);
				shared[BUFFER_CELLSTART]->clobber(); shared[BUFFER_CELLEND]->clobber();
				neibsEngine->reorderDataAndFindCellStart(NULL, sorted, unsorted, n, d_newNum);
				std::swap(curPos, othPos); std::swap(curVel, othVel); std::swap(curVol, othVol);
				sphx_throw(sphx_memcpy_d2h(&n, d_newNum, 4));                     // DOWNLOAD_NEWNUMPARTS
				neibsEngine->resetinfo();
				shared[BUFFER_NEIBSLIST]->clobber();
				BufferList state = *curPos | *curVel | shared;
				neibsEngine->buildNeibsList(state, state, n, n, gridCells, sqNlRadius, sqNlRadius);
				TimingInfo ti; neibsEngine->getinfo(ti);
				if (ti.hasTooManyNeibs >= 0) throw std::runtime_error("too many neighbours");   // CHECK_NEIBSNUM
			}
			if (it > 0)
				for (FilterFreqList::const_iterator flt = fw->getFilterFreqList().begin(); flt != fw->getFilterFreqList().end(); ++flt)
					if (it % flt->second == 0) {
						BufferList rd = *curPos | *curVel | shared, wr = *othVel;
						fw->getFilterEngines().at(flt->first)->process(rd, wr, n, n, slength, influenceRadius);
						std::swap(curVel, othVel);                                    // SWAP_STATE_BUFFERS(BUFFER_VEL)
					}
			float dts[2];
			for (int step = 1; step <= 2; ++step) {
				// forces on step n (predictor) or on n* (corrector); Euler always reads n and writes n*
				BufferList state = (step == 1) ? (*curPos | *curVel | *curVol | shared) : (*othPos | *othVel | *othVol | shared);
				if (grenier)                                                          // COMPUTE_DENSITY (:443-458): VEL in place, SIGMA
					forcesEngine->compute_density(state, state, n, slength, influenceRadius);
				if (sp->turbmodel == SPS || NEEDS_EFFECTIVE_VISC(sp->rheologytype)) {   // CALC_VISC (GPUWorker.cc:2633-2645)
					const float mk = viscEngine->calc_visc(state, state, n, n, deltap, slength, influenceRadius);
					if (!std::isnan(mk)) max_kinvisc = mk;
				}
				shared[BUFFER_FORCES]->clobber(); shared[BUFFER_CFL]->clobber();      // pre_forces
				forcesEngine->bind_textures(state, n, SIMULATE);
				const uint nb = forcesEngine->basicstep(state, state, n, 0, n, deltap, slength, sp->dtadaptfactor,
					influenceRadius, sp->epsilon, NULL, 0, SIMULATE, step, dt, sp->numforcesbodies > 0);
				forcesEngine->unbind_textures(SIMULATE);
				// runCommand<FORCES_SYNC> (src/GPUWorker.cc:2013-2022): stricter viscous limit of the MONAGHAN / ESPANOL_REVENGA models
				const float max_kinvisc_for_dt = max_kinvisc*(sp->viscmodel == MONAGHAN ? pp.monaghan_visc_coeff : sp->viscmodel == ESPANOL_REVENGA ? 5.0f : 1.0f);
				dts[step - 1] = forcesEngine->dtreduce(slength, sp->dtadaptfactor, sspeed_cfl, max_kinvisc_for_dt, state, state, nb, n);
				if (moving) {                                                         // MOVE_BODIES + uploads
					const double dt1 = (step == 1) ? dt/2.0 : (double)dt;
					for (int b = 0; b < numbodies; ++b) {
						Body &B = bodies[b];
						if (step == 1) B.storage = B.kdata; else B.kdata = B.storage;
						double dx[3], dr[9];
						body_callback(B, t, t + dt1, dx, dr);
						trans[b] = make_float3((float)dx[0], (float)dx[1], (float)dx[2]);
						for (int a = 0; a < 9; ++a) steprot[9*b + a] = (float)dr[a];
						lvel[b] = make_float3((float)B.kdata.lvel[0], (float)B.kdata.lvel[1], (float)B.kdata.lvel[2]);
						avel[b] = make_float3((float)B.kdata.avel[0], (float)B.kdata.avel[1], (float)B.kdata.avel[2]);
						grid_and_local(B.kdata.crot, cgGrid[b], cgPos[b]);
					}
					integrationEngine->setrbtrans(trans.data(), numbodies); integrationEngine->setrbsteprot(steprot.data(), numbodies);
					integrationEngine->setrblinearvel(lvel.data(), numbodies); integrationEngine->setrbangularvel(avel.data(), numbodies);
					if (sp->numforcesbodies > 0) forcesEngine->setrbcg(cgGrid.data(), cgPos.data(), numbodies);   // FORCES_UPLOAD_OBJECTS_CG
				}
				BufferList rd = *curPos | *curVel | *curVol | shared, wr = *othPos | *othVel | *othVol;
				integrationEngine->basicstep(rd, wr, n, n, step == 1 ? dt/2 : dt, step, (float)t, slength, influenceRadius, SIMULATE);
			}
			if (moving) integrationEngine->setrbcg(cgGrid.data(), cgPos.data(), numbodies);   // EULER_UPLOAD_OBJECTS_CG
			std::swap(curPos, othPos); std::swap(curVel, othVel); std::swap(curVol, othVol);
			t += dt;
			dt = std::min(dts[0], dts[1]);                                            // TIME_STEP_EPILOGUE (src/GPUSPH.cc:650-657)
		}

		if (has(c, "final_surface")) {       // POSTPROCESS before a write: free-surface flags into INFO
			fw->addPostProcessEngine(SURFACE_DETECTION);
			AbstractPostProcessEngine *surf = fw->hasPostProcessEngine(SURFACE_DETECTION);
			surf->setconstants(sp, &pp, A);
			BufferList rd = *curPos | *curVel | shared, wr = shared;   // INFO is updated in place (get_updated_buffers)
			surf->process(rd, wr, n, n, 0, NULL);
		}
		sphx_throw(sphx_device_synchronize());
		sphx_throw(sphx_memcpy_d2h(hpos.data(), as_const(*curPos).getData<BUFFER_POS>(), 16*(size_t)n));
		sphx_throw(sphx_memcpy_d2h(hvel.data(), as_const(*curVel).getData<BUFFER_VEL>(), 16*(size_t)n));
		sphx_throw(sphx_memcpy_d2h(hinfo.data(), as_const(shared).getData<BUFFER_INFO>(), 8*(size_t)n));
		sphx_throw(sphx_memcpy_d2h(hhash.data(), as_const(shared).getData<BUFFER_HASH>(), 4*(size_t)n));
		sphx_throw(sphx_free(d_newNum));
		FILE *o = fopen(argv[3], "wb");
		if (!o) throw std::runtime_error(std::string("cannot write ") + argv[3]);
		fwrite(&n, 4, 1, o); fwrite(&dt, 4, 1, o); fwrite(&t, 8, 1, o);
		fwrite(hpos.data(), 16, n, o); fwrite(hvel.data(), 16, n, o); fwrite(hinfo.data(), 8, n, o); fwrite(hhash.data(), 4, n, o);
		if (energy) {       // the internal energies behind the usual state
			std::vector<float> he(n);
			sphx_throw(sphx_memcpy_d2h(he.data(), as_const(*curVol).getData<BUFFER_INTERNAL_ENERGY>(), 4*(size_t)n));
			fwrite(he.data(), 4, n, o);
		}
		if (grenier) {      // the volumes behind the usual state
			std::vector<float4> hvol(n);
			sphx_throw(sphx_memcpy_d2h(hvol.data(), as_const(*curVol).getData<BUFFER_VOLUME>(), 16*(size_t)n));
			fwrite(hvol.data(), 16, n, o);
		}
		fclose(o);
		printf("example_engines: %s, %u particles, %u steps, t=%g dt=%g\n", str(c, "framework").c_str(), n, steps, t, dt);
	} catch (const std::exception &e) {
		fprintf(stderr, "example_engines: %s\n", e.what());
		return 1;
	}
	return 0;
}
