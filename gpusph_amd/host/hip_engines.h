// hip_engines.h -- the MI355X engines behind GPUSPH's abstract engine interfaces.
//
// This header is compiled INSIDE the GPUSPH source tree: every type it uses that is not named HIP* or sphx_*
// is GPUSPH's own (AbstractNeibsEngine src/engine_neibs.h:46-107, AbstractForcesEngine src/engine_forces.h:43-180,
// AbstractViscEngine src/engine_visc.h:42-109, AbstractIntegrationEngine src/engine_integration.h:42-144,
// AbstractFilterEngine src/engine_filter.h:41-78, AbstractPostProcessEngine src/engine_postprocess.h:49-105,
// BufferList / Buffer<Key> src/buffer.h, the keys of src/define_buffers.h, SimParams, PhysParams, TimingInfo,
// PlaneList, RunMode).  The classes below take the place of the CUDA*Engine templates of src/cuda/{buildneibs,
// forces,visc,euler,post_process}.cu: they unpack the buffer lists into raw device pointers and call the C ABI of
// libsphx.so (include/sphx.h).  The physics options that the reference bakes into template arguments travel in
// the run-time POD sphx_params, so none of these classes is a template.
//
// No HIP header is needed on this side of the boundary (device memory comes from the sphx_* memory service), so
// the translation unit builds with the host compiler GPUSPH's .cc files are built with.
//
// Semantics kept from the reference engines:
//  * non-const BufferList::getData<>() is called for exactly the buffers a method writes (it marks them dirty and
//    records them for GPUWorker, src/buffer.h:643-674, src/GPUWorker.cc:1826); everything else is read through the
//    const list; an optional buffer that is absent yields NULL = "feature off" (src/buffer.h:633-637)
//  * errors of the library are rethrown as CUDA_SAFE_CALL / KERNEL_CHECK_ERROR throw them
//    (src/cuda/cuda_call.h:57-85): std::invalid_argument for inconsistent arguments, std::runtime_error otherwise
//  * one engine object serves every GPUWorker thread: per-device state is looked up by the calling thread's current
//    device (the reference relies on per-device __constant__ memory and thread_local scratch, src/cuda/forces.cu:89-96)
//  * every pure virtual of the abstract classes is implemented; the ones whose physics is not built into libsphx
//    (SA boundaries, density sum, DEM, Jacobi solver of the granular rheology) throw std::runtime_error naming the method
#ifndef SPHX_HIP_ENGINES_H
#define SPHX_HIP_ENGINES_H

#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "engine_neibs.h"
#include "engine_forces.h"
#include "engine_visc.h"
#include "engine_integration.h"
#include "engine_filter.h"
#include "engine_postprocess.h"
#include "engine_boundary_conditions.h"
#include "linearization.h"

#include "sphx.h"

inline void sphx_throw(int rc)
{
	if (rc == SPHX_OK) return;
	const std::string msg = sphx_last_error();
	if (rc == SPHX_ERR_INVALID) throw std::invalid_argument(msg);
	throw std::runtime_error(msg);
}

[[noreturn]] inline void sphx_not_built(const char *method)
{
	throw std::runtime_error(std::string(method) + ": not built into libsphx (see DESIGN.md, out of scope of the hot path)");
}

// ---- HIPBuffer<Key>: device allocation policy of a Buffer<Key> (role of CUDABuffer, src/cuda/cudabuffer.h:47-131) ----
template<flag_t Key>
class HIPBuffer : public Buffer<Key>
{
	typedef Buffer<Key> baseclass;
public:
	typedef typename baseclass::element_type element_type;

	HIPBuffer(int _init = -1) : Buffer<Key>(_init) {}

	virtual ~HIPBuffer() {
		element_type **bufs = baseclass::get_raw_ptr();
		for (int i = 0; i < baseclass::array_count; ++i) {
			if (bufs[i]) (void)sphx_free(bufs[i]);   // a destructor must not throw
			bufs[i] = NULL;
		}
	}

	// memset of every array to the initial value (0, or 0xFF for neighbour list / cell start / cell end)
	virtual void clobber() {
		const size_t bufmem = AbstractBuffer::get_allocated_elements()*sizeof(element_type);
		element_type **bufs = baseclass::get_raw_ptr();
		for (int i = 0; i < baseclass::array_count; ++i)
			sphx_throw(sphx_memset(bufs[i], baseclass::get_init_value(), bufmem));
	}

	virtual size_t alloc(size_t elems) {
		AbstractBuffer::set_allocated_elements(elems);
		const size_t bufmem = elems*sizeof(element_type);
		element_type **bufs = baseclass::get_raw_ptr();
		for (int i = 0; i < baseclass::array_count; ++i) {
			sphx_throw(sphx_malloc((void**)(bufs + i), bufmem));
			sphx_throw(sphx_memset(bufs[i], baseclass::get_init_value(), bufmem));
		}
		return bufmem*baseclass::array_count;
	}

	virtual void swap_elements(uint idx1, uint idx2, uint _buf = 0) {
		element_type tmp;
		sphx_throw(sphx_memcpy_d2h(&tmp, this->get_offset_buffer(_buf, idx1), sizeof(element_type)));
		sphx_throw(sphx_memcpy_d2d(this->get_offset_buffer(_buf, idx1), this->get_offset_buffer(_buf, idx2), sizeof(element_type)));
		sphx_throw(sphx_memcpy_h2d(this->get_offset_buffer(_buf, idx2), &tmp, sizeof(element_type)));
	}

	virtual const char* get_buffer_class() const
	{ return "HIPBuffer"; }
};

// ---- per-device state shared by the engines of one framework ----
class HIPEngineContext
{
	std::mutex m_lock;
	std::map<int, sphx_ctx*> m_ctx;
	sphx_params m_params;

public:
	HIPEngineContext() { std::memset(&m_params, 0, sizeof(m_params)); }
	~HIPEngineContext() { for (auto &kv : m_ctx) sphx_destroy(kv.second); }

	sphx_ctx *ctx() {
		int dev = 0;
		sphx_throw(sphx_get_device(&dev));
		std::lock_guard<std::mutex> guard(m_lock);
		std::map<int, sphx_ctx*>::iterator it = m_ctx.find(dev);
		if (it != m_ctx.end()) return it->second;
		sphx_ctx *c = NULL;
		sphx_throw(sphx_create(&c, dev));
		m_ctx[dev] = c;
		return c;
	}

	sphx_params const& params() const { return m_params; }

	// host part of the three setconstants(): SimParams + PhysParams + grid -> sphx_params.  Pure host code
	static void fill_params(sphx_params &p, const SimParams *sp, const PhysParams *pp, float3 const& worldOrigin,
		uint3 const& gridSize, float3 const& cellSize, idx_t allocatedParticles)
	{
		// option codes of libsphx are the reference's enum values (pinned by tests/test_oracle_pinned.py), so the
		// conversion is a cast; the linearisation comes from the tree's COORD1..3 (src/linearization.h)
		std::memset(&p, 0, sizeof(p));
		p.gridSize[0] = gridSize.x; p.gridSize[1] = gridSize.y; p.gridSize[2] = gridSize.z;
		p.cellSize[0] = cellSize.x; p.cellSize[1] = cellSize.y; p.cellSize[2] = cellSize.z;
		p.worldOrigin[0] = worldOrigin.x; p.worldOrigin[1] = worldOrigin.y; p.worldOrigin[2] = worldOrigin.z;
		const int3 axes = make_int3(0, 1, 2);
		p.coord[0] = axes.COORD1; p.coord[1] = axes.COORD2; p.coord[2] = axes.COORD3;
		p.periodic = (uint32_t)sp->periodicbound;
		p.neiblistsize = sp->neiblistsize; p.neibboundpos = sp->neibboundpos; p.neiblist_stride = allocatedParticles;
		p.kerneltype = (int32_t)sp->kerneltype; p.sph_formulation = (int32_t)sp->sph_formulation;
		p.densitydiffusiontype = (int32_t)sp->densitydiffusiontype; p.boundarytype = (int32_t)sp->boundarytype;
		p.rheologytype = (int32_t)sp->rheologytype; p.turbmodel = (int32_t)sp->turbmodel;
		p.compvisc = (int32_t)sp->compvisc; p.viscmodel = (int32_t)sp->viscmodel; p.avgop = (int32_t)sp->viscavgop;
		p.is_const_visc = sp->is_const_visc ? 1 : 0;
		p.simflags = sp->simflags;
		p.slength = (float)sp->slength; p.kernelradius = (float)sp->kernelradius;
		p.influenceradius = (float)sp->influenceRadius;
		p.deltap = 0;   // not a constant of the reference engines: forces basicstep receives it per call
		p.dtadaptfactor = sp->dtadaptfactor; p.densityDiffCoeff = sp->densityDiffCoeff; p.epsxsph = pp->epsxsph;
		p.numfluids = (uint32_t)pp->numFluids();
		if (pp->numFluids() > SPHX_MAX_FLUIDS) throw std::invalid_argument("setconstants: more fluids than MAX_FLUID_TYPES");
		for (size_t f = 0; f < pp->numFluids(); ++f) {
			p.rho0[f] = pp->rho0[f]; p.bcoeff[f] = pp->bcoeff[f]; p.gammacoeff[f] = pp->gammacoeff[f];
			p.sscoeff[f] = pp->sscoeff[f]; p.sspowercoeff[f] = pp->sspowercoeff[f];
			p.visccoeff[f] = f < pp->visccoeff.size() ? pp->visccoeff[f] : 0.0f;
		}
		p.gravity[0] = pp->gravity.x; p.gravity[1] = pp->gravity.y; p.gravity[2] = pp->gravity.z;
		p.artvisccoeff = pp->artvisccoeff; p.epsartvisc = pp->epsartvisc;
		p.smagfactor = pp->smagfactor; p.kspsfactor = pp->kspsfactor;
		p.dcoeff = pp->dcoeff; p.p1coeff = pp->p1coeff; p.p2coeff = pp->p2coeff; p.r0 = pp->r0;
		p.repack_a = sp->repack_a; p.repack_alpha = sp->repack_alpha;
		p.partsurf = pp->partsurf;
		p.MK_K = pp->MK_K; p.MK_d = pp->MK_d; p.MK_beta = pp->MK_beta;
		p.epsinterface = pp->epsinterface;
		for (uint f = 0; f < MAX_FLUID_TYPES && f < SPHX_MAX_FLUIDS; ++f) {
			p.yield_strength[f] = f < pp->yield_strength.size() ? pp->yield_strength[f] : 0.0f;
			p.visc_nonlinear_param[f] = f < pp->visc_nonlinear_param.size() ? pp->visc_nonlinear_param[f] : 0.0f;
			p.visc_regularization_param[f] = f < pp->visc_regularization_param.size() ? pp->visc_regularization_param[f] : 0.0f;
		}
		p.limiting_kinvisc = pp->limiting_kinvisc;
		// PhysParams leaves the DEM members uninitialised unless a DEM was added (src/physparams.h:322-327): they travel only with ENABLE_DEM
		const bool dem = (sp->simflags & ENABLE_DEM) != 0;
		p.ewres = dem ? pp->ewres : NAN; p.nsres = dem ? pp->nsres : NAN; p.demdx = dem ? pp->demdx : NAN;
		p.demdy = dem ? pp->demdy : NAN; p.demzmin = dem ? pp->demzmin : NAN;
		p.monaghan_visc_coeff = pp->monaghan_visc_coeff;
		for (size_t f = 0; f < pp->numFluids(); ++f) p.visc2coeff[f] = f < pp->visc2coeff.size() ? pp->visc2coeff[f] : NAN;
	}

	void upload(const SimParams *sp, const PhysParams *pp, float3 const& worldOrigin, uint3 const& gridSize,
		float3 const& cellSize, idx_t allocatedParticles)
	{
		sphx_params p;
		fill_params(p, sp, pp, worldOrigin, gridSize, cellSize, allocatedParticles);
		sphx_ctx *c = ctx();
		sphx_throw(sphx_set_constants(c, &p));
		sphx_throw(sphx_reserve(c, (uint32_t)allocatedParticles));
		std::lock_guard<std::mutex> guard(m_lock);
		m_params = p;
	}
};

typedef std::shared_ptr<HIPEngineContext> HIPEngineContextPtr;

// ---- neighbour engine (CUDANeibsEngine, src/cuda/buildneibs.cu:70-500) ----
class HIPNeibsEngine : public AbstractNeibsEngine
{
	HIPEngineContextPtr m_c;

	// one optional per-particle array of the reorder (src/cuda/buildneibs.cu:263-311)
	template<flag_t Key>
	void gather_optional(BufferList& sorted_buffers, BufferList const& unsorted_buffers, const uint *partIndex, uint numParticles)
	{
		typedef typename BufferTraits<Key>::element_type T;
		for (uint idx = 0; idx < (uint)BufferTraits<Key>::num_buffers; ++idx) {
			const T *oldp = unsorted_buffers.getData<Key>(idx);
			if (!oldp) return;
			T *newp = sorted_buffers.getData<Key>(idx);
			if (!newp) throw std::invalid_argument(std::string("reorderDataAndFindCellStart: sorted ") + BufferTraits<Key>::name + " is null");
			sphx_throw(sphx_gather_rows(m_c->ctx(), newp, oldp, (uint32_t)sizeof(T), partIndex, numParticles, NULL));
		}
	}
public:
	explicit HIPNeibsEngine(HIPEngineContextPtr c) : m_c(c) {}

	void setconstants(const SimParams *simparams, const PhysParams *physparams, float3 const& worldOrigin,
		uint3 const& gridSize, float3 const& cellSize, idx_t const& allocatedParticles)
	{ m_c->upload(simparams, physparams, worldOrigin, gridSize, cellSize, allocatedParticles); }

	void getconstants(SimParams *simparams, PhysParams *)
	{
		sphx_params p;
		sphx_throw(sphx_get_params(m_c->ctx(), &p));
		simparams->neibboundpos = p.neibboundpos;
	}

	void resetinfo() { sphx_throw(sphx_neibs_resetinfo(m_c->ctx(), NULL)); }

	void getinfo(TimingInfo &ti)
	{
		sphx_neibs_info i;
		sphx_throw(sphx_neibs_getinfo(m_c->ctx(), &i, NULL));
		ti.numInteractions = i.numInteractions; ti.maxFluidBoundaryNeibs = i.maxFluidBoundaryNeibs;
		ti.maxVertexNeibs = i.maxVertexNeibs; ti.hasTooManyNeibs = i.hasTooManyNeibs;
		for (int k = 0; k < PT_TESTPOINT; ++k) ti.hasMaxNeibs[k] = i.hasMaxNeibs[k];
	}

	void calcHash(const BufferList& bufread, BufferList& bufwrite, const uint numParticles)
	{
		sphx_throw(sphx_calc_hash(m_c->ctx(), bufwrite.getData<BUFFER_POS>(), bufwrite.getData<BUFFER_HASH>(),
			bufwrite.getData<BUFFER_PARTINDEX>(), bufread.getData<BUFFER_INFO>(),
			bufread.getData<BUFFER_COMPACT_DEV_MAP>(), numParticles, NULL));
	}

	void fixHash(const BufferList& bufread, BufferList& bufwrite, const uint numParticles)
	{
		sphx_throw(sphx_fix_hash(m_c->ctx(), bufwrite.getData<BUFFER_HASH>(), bufwrite.getData<BUFFER_PARTINDEX>(),
			bufread.getData<BUFFER_INFO>(), bufread.getData<BUFFER_COMPACT_DEV_MAP>(), numParticles, NULL));
	}

	// keys (HASH, INFO) and values (PARTINDEX) are sorted in place (src/cuda/buildneibs.cu:384-412)
	void sort(const BufferList&, BufferList& bufwrite, uint numParticles)
	{
		sphx_throw(sphx_sort(m_c->ctx(), bufwrite.getData<BUFFER_HASH>(), bufwrite.getData<BUFFER_INFO>(),
			bufwrite.getData<BUFFER_PARTINDEX>(), numParticles, NULL));
	}

	void reorderDataAndFindCellStart(uint *segmentStart, BufferList& sorted_buffers,
		const BufferList& unsorted_buffers, const uint numParticles, uint *newNumParticles)
	{
		// HASH, PARTINDEX and INFO were sorted in place and are only read here
		const hashKey *particleHash = sorted_buffers.getConstData<BUFFER_HASH>();
		const uint *particleIndex = sorted_buffers.getConstData<BUFFER_PARTINDEX>();
		const particleinfo *particleInfo = sorted_buffers.getConstData<BUFFER_INFO>();
		uint *cellStart = sorted_buffers.getData<BUFFER_CELLSTART>();
		uint *cellEnd = sorted_buffers.getData<BUFFER_CELLEND>();
		sphx_throw(sphx_reorder(m_c->ctx(), segmentStart, cellStart, cellEnd,
			sorted_buffers.getData<BUFFER_POS>(), sorted_buffers.getData<BUFFER_VEL>(),
			unsorted_buffers.getData<BUFFER_POS>(), unsorted_buffers.getData<BUFFER_VEL>(),
			particleInfo, particleHash, particleIndex, numParticles, newNumParticles, NULL));
		gather_optional<BUFFER_VOLUME>(sorted_buffers, unsorted_buffers, particleIndex, numParticles);
		gather_optional<BUFFER_INTERNAL_ENERGY>(sorted_buffers, unsorted_buffers, particleIndex, numParticles);
		gather_optional<BUFFER_BOUNDELEMENTS>(sorted_buffers, unsorted_buffers, particleIndex, numParticles);
		gather_optional<BUFFER_GRADGAMMA>(sorted_buffers, unsorted_buffers, particleIndex, numParticles);
		gather_optional<BUFFER_VERTICES>(sorted_buffers, unsorted_buffers, particleIndex, numParticles);
		gather_optional<BUFFER_TKE>(sorted_buffers, unsorted_buffers, particleIndex, numParticles);
		gather_optional<BUFFER_EPSILON>(sorted_buffers, unsorted_buffers, particleIndex, numParticles);
		gather_optional<BUFFER_TURBVISC>(sorted_buffers, unsorted_buffers, particleIndex, numParticles);
		gather_optional<BUFFER_EFFPRES>(sorted_buffers, unsorted_buffers, particleIndex, numParticles);
		gather_optional<BUFFER_EULERVEL>(sorted_buffers, unsorted_buffers, particleIndex, numParticles);
		// BUFFER_NEXTID (particle creation at open boundaries) is reordered and extended by the SA path only
		if (unsorted_buffers.getData<BUFFER_NEXTID>())
			sphx_not_built("reorderDataAndFindCellStart with BUFFER_NEXTID (open boundaries)");
	}

	void buildNeibsList(const BufferList& bufread, BufferList& bufwrite, const uint numParticles,
		const uint particleRangeEnd, const uint gridCells, const float sqinfluenceradius, const float boundNlSqInflRad)
	{
		// consistency rule of the reference (src/cuda/buildneibs.cu:440-455): the SA arrays come together or not at all
		const vertexinfo *vertices = bufread.getData<BUFFER_VERTICES>();
		const float4 *boundelem = bufread.getData<BUFFER_BOUNDELEMENTS>();
		float2 **vertPos = bufwrite.getRawPtr<BUFFER_VERTPOS>();
		if (vertices || boundelem || vertPos) {
			if (!vertices || !boundelem || !vertPos)
				throw std::invalid_argument("inconsistent params to buildNeibsList");
		}
		sphx_params prm;
		sphx_throw(sphx_get_params(m_c->ctx(), &prm));
		if (prm.boundarytype == SPHX_SA_BOUNDARY && !vertices)
			throw std::invalid_argument("missing data");
		sphx_throw(sphx_build_neibs_sa(m_c->ctx(), bufwrite.getData<BUFFER_NEIBSLIST>(),
			vertPos ? vertPos[0] : NULL, vertPos ? vertPos[1] : NULL, vertPos ? vertPos[2] : NULL,
			bufread.getData<BUFFER_POS>(), bufread.getData<BUFFER_INFO>(), vertices, boundelem, bufread.getData<BUFFER_HASH>(),
			bufread.getData<BUFFER_CELLSTART>(), bufread.getData<BUFFER_CELLEND>(),
			numParticles, particleRangeEnd, gridCells, sqinfluenceradius, boundNlSqInflRad, NULL));
	}
};

// ---- forces engine (CUDAForcesEngine, src/cuda/forces.cu:180-1005) ----
class HIPForcesEngine : public AbstractForcesEngine
{
	HIPEngineContextPtr m_c;
public:
	explicit HIPForcesEngine(HIPEngineContextPtr c) : m_c(c) {}

	void setconstants(const SimParams *simparams, const PhysParams *physparams, float3 const& worldOrigin,
		uint3 const& gridSize, float3 const& cellSize, idx_t const& allocatedParticles)
	{ m_c->upload(simparams, physparams, worldOrigin, gridSize, cellSize, allocatedParticles); }

	// read back what setconstants uploaded (src/cuda/forces.cu:403-440)
	void getconstants(PhysParams *physparams)
	{
		sphx_params p;
		sphx_throw(sphx_get_params(m_c->ctx(), &p));
		if (p.numfluids != physparams->numFluids())
			throw std::runtime_error("wrong number of fluids");
		for (uint32_t f = 0; f < p.numfluids; ++f) {
			physparams->visccoeff[f] = p.visccoeff[f];
			physparams->rho0[f] = p.rho0[f]; physparams->bcoeff[f] = p.bcoeff[f]; physparams->gammacoeff[f] = p.gammacoeff[f];
			physparams->sscoeff[f] = p.sscoeff[f]; physparams->sspowercoeff[f] = p.sspowercoeff[f];
		}
		physparams->gravity = make_float3(p.gravity[0], p.gravity[1], p.gravity[2]);
		physparams->dcoeff = p.dcoeff; physparams->p1coeff = p.p1coeff; physparams->p2coeff = p.p2coeff;
		physparams->MK_K = p.MK_K; physparams->MK_d = p.MK_d; physparams->MK_beta = p.MK_beta;
		physparams->epsinterface = p.epsinterface;
		physparams->r0 = p.r0; physparams->epsartvisc = p.epsartvisc;
	}

	void setplanes(PlaneList const& planes)
	{
		std::vector<float> nrm, pos; std::vector<int32_t> gp;
		for (PlaneList::const_iterator pl = planes.begin(); pl != planes.end(); ++pl) {
			nrm.push_back(pl->normal.x); nrm.push_back(pl->normal.y); nrm.push_back(pl->normal.z);
			gp.push_back(pl->gridPos.x); gp.push_back(pl->gridPos.y); gp.push_back(pl->gridPos.z);
			pos.push_back(pl->pos.x); pos.push_back(pl->pos.y); pos.push_back(pl->pos.z);
		}
		sphx_throw(sphx_set_planes(m_c->ctx(), nrm.data(), gp.data(), pos.data(), (int)planes.size()));
	}

	void setgravity(float3 const& g)
	{
		const float v[3] = { g.x, g.y, g.z };
		sphx_throw(sphx_set_gravity(m_c->ctx(), v));
	}

	// the forces engine's own copy of the centres of gravity (cuforces::d_rbcgGridPos/d_rbcgPos)
	void setrbcg(const int3 *cgGridPos, const float3 *cgPos, int numbodies)
	{ sphx_throw(sphx_set_rb_cg_forces(m_c->ctx(), (const int32_t*)cgGridPos, (const float*)cgPos, numbodies)); }

	void setrbstart(const int *rbfirstindex, int numbodies)
	{ sphx_throw(sphx_set_rb_start(m_c->ctx(), rbfirstindex, numbodies)); }

	void reduceRbForces(BufferList& bufwrite, uint *lastindex, float3 *totalforce, float3 *totaltorque,
		uint numbodies, uint numBodiesParticles)
	{
		float4 *forces = bufwrite.getData<BUFFER_RB_FORCES>();
		float4 *torques = bufwrite.getData<BUFFER_RB_TORQUES>();
		const uint *rbnum = bufwrite.getConstData<BUFFER_RB_KEYS>();
		sphx_throw(sphx_reduce_rb_forces(m_c->ctx(), forces, torques, rbnum, lastindex,
			(float*)totalforce, (float*)totaltorque, numbodies, numBodiesParticles, NULL));
	}

	// there are no texture references on this path: loads go through L2 / LDS (src/cuda/forces.cu:469-532 has no equivalent)
	void bind_textures(const BufferList&, uint, RunMode) {}
	void unbind_textures(RunMode) {}

	void setDEM(const float *hDem, int width, int height) { sphx_throw(sphx_set_dem(m_c->ctx(), hDem, width, height)); }
	void unsetDEM() { sphx_throw(sphx_set_dem(m_c->ctx(), NULL, 0, 0)); }

	uint round_particles(uint numparts) { return sphx_forces_round_particles(numparts); }

	// CUDADensityHelper<kerneltype, SPH_GRENIER, boundarytype>::process (src/cuda/forces.cu:208-246): vel of the write list is
	// updated in place, sigma written; nothing for the other formulations
	void compute_density(const BufferList& bufread, BufferList& bufwrite, uint numParticles, float slength, float influenceradius)
	{
		if (m_c->params().sph_formulation != SPH_GRENIER) return;
		sphx_throw(sphx_compute_density(m_c->ctx(), bufwrite.getData<BUFFER_SIGMA>(), bufwrite.getData<BUFFER_VEL>(),
			bufread.getData<BUFFER_POS>(), bufread.getData<BUFFER_INFO>(), bufread.getData<BUFFER_HASH>(),
			bufread.getData<BUFFER_VOLUME>(), bufread.getData<BUFFER_CELLSTART>(), bufread.getData<BUFFER_NEIBSLIST>(),
			numParticles, slength, influenceradius, NULL));
	}

	// Brezzi diffusion after the density summation (src/cuda/forces.cu:621-661): the diffusive density rate into FORCES.w
	void compute_density_diffusion(const BufferList& bufread, BufferList& bufwrite, const uint numParticles,
		const uint particleRangeEnd, const float deltap, const float slength, const float influenceRadius, const float dt)
	{
		if (m_c->params().simflags & ENABLE_INLET_OUTLET) {
			// with open boundaries the term of the pressure-driven segments reads the elements and their vertices
			// (sa_boundary_density_diffusion_params, src/cuda/density_diffusion_params.h:80-100)
			const float2 * const *vertPos = bufread.getRawPtr<BUFFER_VERTPOS>();
			if (!vertPos) throw std::invalid_argument("compute_density_diffusion: open boundaries need BUFFER_VERTPOS");
			sphx_throw(sphx_sa_compute_density_diffusion_io(m_c->ctx(), bufwrite.getData<BUFFER_FORCES>(),
				bufread.getData<BUFFER_POS>(), bufread.getData<BUFFER_VEL>(), bufread.getData<BUFFER_GRADGAMMA>(),
				bufread.getData<BUFFER_BOUNDELEMENTS>(), vertPos[0], vertPos[1], vertPos[2], bufread.getData<BUFFER_INFO>(),
				bufread.getData<BUFFER_HASH>(), bufread.getData<BUFFER_CELLSTART>(), bufread.getData<BUFFER_NEIBSLIST>(),
				numParticles, particleRangeEnd, deltap, dt, NULL));
			return;
		}
		sphx_throw(sphx_sa_compute_density_diffusion(m_c->ctx(), bufwrite.getData<BUFFER_FORCES>(),
			bufread.getData<BUFFER_POS>(), bufread.getData<BUFFER_VEL>(), bufread.getData<BUFFER_GRADGAMMA>(),
			bufread.getData<BUFFER_INFO>(), bufread.getData<BUFFER_HASH>(), bufread.getData<BUFFER_CELLSTART>(),
			bufread.getData<BUFFER_NEIBSLIST>(), numParticles, particleRangeEnd, deltap, slength, influenceRadius, dt, NULL));
	}

	uint basicstep(const BufferList& bufread, BufferList& bufwrite, uint numParticles, uint fromParticle,
		uint toParticle, float deltap, float slength, float dtadaptfactor, float influenceradius,
		const float epsilon, uint *IOwaterdepth, uint cflOffset, const RunMode run_mode, const int step, const float dt,
		const bool compute_object_forces)
	{
		const sphx_params &P = m_c->params();
		// the write set follows the structures forces_params is assembled from (src/cuda/forces_params.h:86-220)
		float4 *forces = bufwrite.getData<BUFFER_FORCES>();
		float *cfl = (P.simflags & ENABLE_DTADAPT) ? bufwrite.getData<BUFFER_CFL>() : NULL;
		if (P.boundarytype == SA_BOUNDARY) {
			// sa_boundary_forces_params (src/cuda/forces_params.h): gamma of the state that is read, the boundary elements,
			// the vertex offsets of the segments
			const float2 * const *vertPos = bufread.getRawPtr<BUFFER_VERTPOS>();
			if (!vertPos) throw std::invalid_argument("forces basicstep: SA_BOUNDARY needs BUFFER_VERTPOS");
			uint32_t nb = 0;
			// BUFFER_CFL_GAMMA is there with dynamic gamma and adaptive dt (src/cuda/forces_params.h, dyndt + gamma)
			const bool gcfl = !(P.simflags & ENABLE_GAMMA_QUADRATURE) && (P.simflags & ENABLE_DTADAPT);
			// bodies that feel the fluid: with SA_BOUNDARY the force on a COMPUTE_FORCE element is the pressure on its area
			// (compute_boundary_pressure_force in finalizeforcesDevice, src/cuda/forces_kernel.def:3258-3266,4115-4145), written to
			// BUFFER_RB_FORCES / BUFFER_RB_TORQUES behind whichever SA forces entry ran on this range
			auto bodyForces = [&]() {
				if (!compute_object_forces || run_mode != SIMULATE) return;
				float4 *rbf = bufwrite.getData<BUFFER_RB_FORCES>(), *rbt = bufwrite.getData<BUFFER_RB_TORQUES>();
				if (!rbf || !rbt) return;
				sphx_throw(sphx_sa_body_pressure_forces(m_c->ctx(), forces, rbf, rbt, bufread.getData<BUFFER_POS>(),
					bufread.getData<BUFFER_VEL>(), bufread.getData<BUFFER_INFO>(), bufread.getData<BUFFER_HASH>(),
					bufread.getData<BUFFER_BOUNDELEMENTS>(), fromParticle, toParticle, NULL));
			};
			if (P.turbmodel == KEPSILON && run_mode == SIMULATE) {
				// keps_forces_params (src/cuda/forces_params.h:283-320): k, epsilon, eddy viscosity and Eulerian velocity of the
				// state that is read; BUFFER_DKDE and BUFFER_CFL_KEPS written (BUFFER_TAU is not needed: one launch)
				sphx_throw(sphx_forces_basicstep_sa_keps(m_c->ctx(), forces, cfl, gcfl ? bufwrite.getData<BUFFER_CFL_GAMMA>() : NULL,
					(P.simflags & ENABLE_DTADAPT) ? bufwrite.getData<BUFFER_CFL_KEPS>() : NULL, (float*)bufwrite.getData<BUFFER_DKDE>(),
					bufread.getData<BUFFER_POS>(), bufread.getData<BUFFER_VEL>(), bufread.getData<BUFFER_INFO>(),
					bufread.getData<BUFFER_HASH>(), bufread.getData<BUFFER_CELLSTART>(), bufread.getData<BUFFER_NEIBSLIST>(),
					bufread.getData<BUFFER_GRADGAMMA>(), bufread.getData<BUFFER_BOUNDELEMENTS>(), vertPos[0], vertPos[1], vertPos[2],
					bufread.getData<BUFFER_TKE>(), bufread.getData<BUFFER_EPSILON>(), bufread.getData<BUFFER_TURBVISC>(),
					bufread.getData<BUFFER_EULERVEL>(), numParticles, fromParticle, toParticle, deltap, slength, dtadaptfactor,
					influenceradius, epsilon, cflOffset, (int)run_mode, step, dt, &nb, NULL));
				bodyForces();
				return nb;
			}
			if ((P.simflags & ENABLE_INLET_OUTLET) && run_mode == SIMULATE) {
				// open boundaries: BUFFER_EULERVEL of the state that is read in the viscous terms and in the gamma CFL condition;
				// then what the vertex pass leaves behind with ENABLE_WATER_DEPTH (vertex_forces, src/cuda/forces.cu:676-686)
				sphx_throw(sphx_forces_basicstep_sa_io(m_c->ctx(), forces, cfl, gcfl ? bufwrite.getData<BUFFER_CFL_GAMMA>() : NULL,
					bufread.getData<BUFFER_POS>(), bufread.getData<BUFFER_VEL>(), bufread.getData<BUFFER_EULERVEL>(),
					bufread.getData<BUFFER_INFO>(), bufread.getData<BUFFER_HASH>(), bufread.getData<BUFFER_CELLSTART>(),
					bufread.getData<BUFFER_NEIBSLIST>(), bufread.getData<BUFFER_GRADGAMMA>(), bufread.getData<BUFFER_BOUNDELEMENTS>(),
					vertPos[0], vertPos[1], vertPos[2], numParticles, fromParticle, toParticle, deltap, cflOffset, &nb, NULL));
				if ((P.simflags & ENABLE_WATER_DEPTH) && IOwaterdepth)
					sphx_throw(sphx_sa_io_water_depth(m_c->ctx(), IOwaterdepth, bufread.getData<BUFFER_POS>(),
						bufread.getData<BUFFER_INFO>(), bufread.getData<BUFFER_HASH>(), bufread.getData<BUFFER_CELLSTART>(),
						bufread.getData<BUFFER_NEIBSLIST>(), numParticles, fromParticle, toParticle, NULL));
				bodyForces();
				return nb;
			}
			sphx_throw(sphx_forces_basicstep_sa(m_c->ctx(), forces, cfl, gcfl ? bufwrite.getData<BUFFER_CFL_GAMMA>() : NULL,
				bufread.getData<BUFFER_POS>(), bufread.getData<BUFFER_VEL>(), bufread.getData<BUFFER_INFO>(),
				bufread.getData<BUFFER_HASH>(), bufread.getData<BUFFER_CELLSTART>(), bufread.getData<BUFFER_NEIBSLIST>(),
				bufread.getData<BUFFER_GRADGAMMA>(), bufread.getData<BUFFER_BOUNDELEMENTS>(), vertPos[0], vertPos[1], vertPos[2],
				numParticles, fromParticle, toParticle, deltap, slength, dtadaptfactor, influenceradius,
				cflOffset, (int)run_mode, step, dt, &nb, NULL));
			bodyForces();
			return nb;
		}
		if (NEEDS_EFFECTIVE_VISC(P.rheologytype) && run_mode == SIMULATE) {
			// effective_visc_forces_params (src/cuda/forces_params.h): the viscosity of every particle from BUFFER_EFFVISC
			if (compute_object_forces)
				sphx_not_built("forces basicstep: generalized Newtonian rheologies with bodies that feel the fluid (BUFFER_RB_FORCES)");
			uint32_t nb = 0;
			sphx_throw(sphx_forces_basicstep_effvisc(m_c->ctx(), forces, cfl,
				bufread.getData<BUFFER_POS>(), bufread.getData<BUFFER_VEL>(), bufread.getData<BUFFER_INFO>(),
				bufread.getData<BUFFER_HASH>(), bufread.getData<BUFFER_CELLSTART>(), bufread.getData<BUFFER_NEIBSLIST>(),
				bufread.getData<BUFFER_EFFVISC>(), numParticles, fromParticle, toParticle, deltap, slength, dtadaptfactor,
				influenceradius, cflOffset, (int)run_mode, step, dt, &nb, NULL));
			return nb;
		}
		if (P.sph_formulation == SPH_GRENIER && run_mode == SIMULATE) {
			// grenier_forces_params (src/cuda/forces_params.h:224-240): sigma of the state that is read
			if (compute_object_forces)
				sphx_not_built("forces basicstep: SPH_GRENIER with bodies that feel the fluid (BUFFER_RB_FORCES)");
			uint32_t nb = 0;
			sphx_throw(sphx_forces_basicstep_grenier(m_c->ctx(), forces, cfl,
				bufread.getData<BUFFER_POS>(), bufread.getData<BUFFER_VEL>(), bufread.getData<BUFFER_INFO>(),
				bufread.getData<BUFFER_HASH>(), bufread.getData<BUFFER_CELLSTART>(), bufread.getData<BUFFER_NEIBSLIST>(),
				bufread.getData<BUFFER_SIGMA>(), numParticles, fromParticle, toParticle, deltap, slength, dtadaptfactor,
				influenceradius, cflOffset, (int)run_mode, step, dt, &nb, NULL));
			return nb;
		}
		float4 *rbforces = bufwrite.getData<BUFFER_RB_FORCES>();
		float4 *rbtorques = bufwrite.getData<BUFFER_RB_TORQUES>();
		float4 *xsph = (P.simflags & ENABLE_XSPH) ? bufwrite.getData<BUFFER_XSPH>() : NULL;
		const float2 *const *tau = (P.turbmodel == SPS && run_mode == SIMULATE) ? bufread.getRawPtr<BUFFER_TAU>() : NULL;
		uint32_t numBlocks = 0;
		sphx_throw(sphx_forces_basicstep(m_c->ctx(), forces, cfl, rbforces, rbtorques,
			bufread.getData<BUFFER_POS>(), bufread.getData<BUFFER_VEL>(), bufread.getData<BUFFER_INFO>(),
			bufread.getData<BUFFER_HASH>(), bufread.getData<BUFFER_CELLSTART>(), bufread.getData<BUFFER_NEIBSLIST>(),
			tau ? tau[0] : NULL, tau ? tau[1] : NULL, tau ? tau[2] : NULL, xsph,
			numParticles, fromParticle, toParticle, deltap, slength, dtadaptfactor, influenceradius,
			cflOffset, (int)run_mode, step, dt, compute_object_forces ? 1 : 0, &numBlocks, NULL));
		if ((P.simflags & ENABLE_INTERNAL_ENERGY) && run_mode == SIMULATE)
			// internal_energy_forces_params (src/cuda/forces_params.h:296-303): DEDt of this pass
			sphx_throw(sphx_forces_internal_energy(m_c->ctx(), bufwrite.getData<BUFFER_INTERNAL_ENERGY_UPD>(),
				bufread.getData<BUFFER_POS>(), bufread.getData<BUFFER_VEL>(), bufread.getData<BUFFER_INFO>(),
				bufread.getData<BUFFER_HASH>(), bufread.getData<BUFFER_CELLSTART>(), bufread.getData<BUFFER_NEIBSLIST>(),
				numParticles, fromParticle, toParticle, NULL));
		return numBlocks;
	}

	uint getFmaxElements(const uint n) { return sphx_forces_fmax_elements(n); }
	uint getFmaxTempElements(const uint n) { return sphx_forces_fmax_temp_elements(n); }

	float dtreduce(float slength, float dtadaptfactor, float sspeed_cfl, float max_kinematic,
		BufferList const& bufread, BufferList& bufwrite, uint numBlocks, uint numParticles)
	{
		float dt = 0;
		sphx_throw(sphx_forces_dtreduce(m_c->ctx(), slength, dtadaptfactor, sspeed_cfl, max_kinematic,
			bufread.getData<BUFFER_CFL>(), bufwrite.getData<BUFFER_CFL_TEMP>(), numBlocks, &dt, NULL));
		// SA_BOUNDARY with dynamic gamma: the CFL condition of the gamma transport (src/cuda/forces.cu:576-585).  The library
		// applies it after the viscous limit; the result is the same (the 1e-5/dt floor only matters when gamma does not limit)
		const sphx_params &P = m_c->params();
		if (P.boundarytype == SA_BOUNDARY && USING_DYNAMIC_GAMMA(P.simflags) && bufread.getData<BUFFER_CFL_GAMMA>())
			sphx_throw(sphx_forces_dtreduce_gamma(m_c->ctx(), bufread.getData<BUFFER_CFL_GAMMA>(), numParticles, numBlocks, &dt, NULL));
		if (P.turbmodel == KEPSILON && bufread.getData<BUFFER_CFL_KEPS>() && numBlocks) {
			// the viscous limit with the largest eddy viscosity added to max_kinematic (src/cuda/forces.cu:585-598)
			sphx_throw(sphx_forces_dtreduce_keps(m_c->ctx(), bufread.getData<BUFFER_CFL_KEPS>(), numBlocks, slength, max_kinematic, &dt, NULL));
		}
		return dt;
	}
};

// ---- viscous engine (CUDAViscEngine, src/cuda/visc.cu:60-640) ----
class HIPViscEngine : public AbstractViscEngine
{
	HIPEngineContextPtr m_c;
public:
	explicit HIPViscEngine(HIPEngineContextPtr c) : m_c(c) {}

	void setconstants() {}
	void getconstants() {}

	// SPS: stress tensor + turbulent viscosity; Newtonian rheologies have nothing to compute per particle
	// (CUDAViscEngineHelper's default, src/cuda/visc.cu:66-85).  Returns NAN as the reference does when there is
	// no per-particle kinematic viscosity to bound the time step with (src/cuda/visc.cu:232)
	float calc_visc(const BufferList& bufread, BufferList& bufwrite, const uint numParticles,
		const uint particleRangeEnd, const float deltap, const float slength, const float influenceradius)
	{
		const sphx_params &P = m_c->params();
		if (P.rheologytype == GRANULAR)
			sphx_not_built("calc_visc for the granular rheology");
		if (NEEDS_EFFECTIVE_VISC(P.rheologytype)) {
			// effective viscosity of the generalized Newtonian rheologies (src/cuda/visc.cu:86-170): BUFFER_EFFVISC written, the
			// largest kinematic viscosity returned for the viscous limit of dt
			float max_kinvisc = NAN;
			sphx_throw(sphx_calc_effvisc(m_c->ctx(), bufwrite.getData<BUFFER_EFFVISC>(), &max_kinvisc,
				bufread.getData<BUFFER_POS>(), bufread.getData<BUFFER_VEL>(), bufread.getData<BUFFER_INFO>(),
				bufread.getData<BUFFER_HASH>(), bufread.getData<BUFFER_CELLSTART>(), bufread.getData<BUFFER_NEIBSLIST>(),
				numParticles, particleRangeEnd, deltap, slength, influenceradius, NULL));
			return max_kinvisc;
		}
		if (P.turbmodel != SPS)
			return NAN;
		float2 **tau = bufwrite.getRawPtr<BUFFER_TAU>();
		float *turbvisc = bufwrite.getData<BUFFER_SPS_TURBVISC>();
		if (!tau) throw std::invalid_argument("calc_visc: SPS needs BUFFER_TAU");
		sphx_throw(sphx_calc_visc(m_c->ctx(), tau[0], tau[1], tau[2], turbvisc,
			bufread.getData<BUFFER_POS>(), bufread.getData<BUFFER_VEL>(), bufread.getData<BUFFER_INFO>(),
			bufread.getData<BUFFER_HASH>(), bufread.getData<BUFFER_CELLSTART>(), bufread.getData<BUFFER_NEIBSLIST>(),
			numParticles, particleRangeEnd, deltap, slength, influenceradius, NULL));
		return NAN;
	}

	void enforce_jacobi_fs_boundary_conditions(const BufferList&, BufferList&, const uint, const uint, const float, const float, const float)
	{ sphx_not_built("enforce_jacobi_fs_boundary_conditions (GRANULAR rheology)"); }
	float enforce_jacobi_wall_boundary_conditions(const BufferList&, BufferList&, const uint, const uint, const float, const float, const float)
	{ sphx_not_built("enforce_jacobi_wall_boundary_conditions (GRANULAR rheology)"); }
	void build_jacobi_vectors(const BufferList&, BufferList&, const uint, const uint, const float, const float, const float)
	{ sphx_not_built("build_jacobi_vectors (GRANULAR rheology)"); }
	float update_jacobi_effpres(const BufferList&, BufferList&, const uint, const uint, const float, const float, const float)
	{ sphx_not_built("update_jacobi_effpres (GRANULAR rheology)"); }
};

// ---- predictor-corrector integration engine (CUDAPredCorrEngine, src/cuda/euler.cu:40-395) ----
class HIPPredCorrEngine : public AbstractIntegrationEngine
{
	HIPEngineContextPtr m_c;
public:
	explicit HIPPredCorrEngine(HIPEngineContextPtr c) : m_c(c) {}

	// the constants of cueuler (grid, EOS, neighbour list geometry, src/cuda/euler.cu:51-69) are those the forces and
	// neighbour engines upload into the same per-device state
	void setconstants(const PhysParams *, float3 const&, uint3 const&, float3 const&, idx_t const&, int const&, float const&) {}
	void getconstants(PhysParams *) {}

	// the integration engine's own copy of the centres of gravity (cueuler::d_rbcgGridPos/d_rbcgPos)
	void setrbcg(const int3 *cgGridPos, const float3 *cgPos, int numbodies)
	{ sphx_throw(sphx_set_rb_cg_integration(m_c->ctx(), (const int32_t*)cgGridPos, (const float*)cgPos, numbodies)); }
	void setrbtrans(const float3 *trans, int n)
	{ sphx_throw(sphx_set_rb_motion(m_c->ctx(), (const float*)trans, NULL, NULL, NULL, n)); }
	void setrbsteprot(const float *rot, int n)
	{ sphx_throw(sphx_set_rb_motion(m_c->ctx(), NULL, rot, NULL, NULL, n)); }
	void setrblinearvel(const float3 *v, int n)
	{ sphx_throw(sphx_set_rb_motion(m_c->ctx(), NULL, NULL, (const float*)v, NULL, n)); }
	void setrbangularvel(const float3 *w, int n)
	{ sphx_throw(sphx_set_rb_motion(m_c->ctx(), NULL, NULL, NULL, (const float*)w, n)); }

	// common_density_sum_params (src/cuda/density_sum_params.h:60-110): old positions / velocities / gamma and the boundary
	// elements from the state that is read; the NEW positions (read), the new density and gamma (written) and the FORCES
	// scratch from the write list
	void density_sum(const BufferList& bufread, BufferList& bufwrite, const uint numParticles, const uint particleRangeEnd,
		const float dt, const int step, const float t, const float epsilon, const float deltap, const float slength,
		const float influenceRadius)
	{
		const float2 * const *vertPos = bufread.getRawPtr<BUFFER_VERTPOS>();
		if (!vertPos) throw std::invalid_argument("density_sum: BUFFER_VERTPOS missing");
		const float4 *newPos = bufwrite.getConstData<BUFFER_POS>();
		if ((m_c->params().simflags & ENABLE_INLET_OUTLET) && (m_c->params().simflags & ENABLE_MOVING_BODIES)) {
			// both (the option set of CompleteSaExample.cu:46): the Eulerian velocities of the state that is read AND the boundary
			// elements of both states (io_density_sum_params + boundelements_density_sum_params, src/cuda/density_sum_params.h)
			sphx_throw(sphx_sa_density_sum_io_moving(m_c->ctx(), bufwrite.getData<BUFFER_VEL>(), bufwrite.getData<BUFFER_GRADGAMMA>(),
				bufwrite.getData<BUFFER_FORCES>(), bufread.getData<BUFFER_POS>(), newPos, bufread.getData<BUFFER_VEL>(),
				bufread.getData<BUFFER_EULERVEL>(), bufread.getData<BUFFER_GRADGAMMA>(), bufread.getData<BUFFER_BOUNDELEMENTS>(),
				bufwrite.getConstData<BUFFER_BOUNDELEMENTS>(), vertPos[0], vertPos[1], vertPos[2], bufread.getData<BUFFER_INFO>(),
				bufread.getData<BUFFER_HASH>(), bufread.getData<BUFFER_CELLSTART>(), bufread.getData<BUFFER_NEIBSLIST>(), numParticles,
				particleRangeEnd, dt, NULL));
			return;
		}
		if (m_c->params().simflags & ENABLE_INLET_OUTLET) {
			// io_density_sum_params (src/cuda/density_sum_params.h): BUFFER_EULERVEL of the state that is read
			sphx_throw(sphx_sa_density_sum_io(m_c->ctx(), bufwrite.getData<BUFFER_VEL>(), bufwrite.getData<BUFFER_GRADGAMMA>(),
				bufwrite.getData<BUFFER_FORCES>(), bufread.getData<BUFFER_POS>(), newPos, bufread.getData<BUFFER_VEL>(),
				bufread.getData<BUFFER_EULERVEL>(), bufread.getData<BUFFER_GRADGAMMA>(), bufread.getData<BUFFER_BOUNDELEMENTS>(),
				vertPos[0], vertPos[1], vertPos[2], bufread.getData<BUFFER_INFO>(), bufread.getData<BUFFER_HASH>(),
				bufread.getData<BUFFER_CELLSTART>(), bufread.getData<BUFFER_NEIBSLIST>(), numParticles, particleRangeEnd, dt, NULL));
			return;
		}
		if (m_c->params().simflags & ENABLE_MOVING_BODIES) {
			// boundelements_density_sum_params (src/cuda/density_sum_params.h): with moving bodies BUFFER_BOUNDELEMENTS is a state
			// buffer, the old one in the read list, the new one (read only) in the write list
			sphx_throw(sphx_sa_density_sum_moving(m_c->ctx(), bufwrite.getData<BUFFER_VEL>(), bufwrite.getData<BUFFER_GRADGAMMA>(),
				bufwrite.getData<BUFFER_FORCES>(), bufread.getData<BUFFER_POS>(), newPos, bufread.getData<BUFFER_VEL>(),
				bufread.getData<BUFFER_GRADGAMMA>(), bufread.getData<BUFFER_BOUNDELEMENTS>(), bufwrite.getConstData<BUFFER_BOUNDELEMENTS>(),
				vertPos[0], vertPos[1], vertPos[2], bufread.getData<BUFFER_INFO>(), bufread.getData<BUFFER_HASH>(),
				bufread.getData<BUFFER_CELLSTART>(), bufread.getData<BUFFER_NEIBSLIST>(), numParticles, particleRangeEnd, NULL));
			return;
		}
		sphx_throw(sphx_sa_density_sum(m_c->ctx(), bufwrite.getData<BUFFER_VEL>(), bufwrite.getData<BUFFER_GRADGAMMA>(),
			bufwrite.getData<BUFFER_FORCES>(), bufread.getData<BUFFER_POS>(), newPos, bufread.getData<BUFFER_VEL>(),
			bufread.getData<BUFFER_GRADGAMMA>(), bufread.getData<BUFFER_BOUNDELEMENTS>(), vertPos[0], vertPos[1], vertPos[2],
			bufread.getData<BUFFER_INFO>(), bufread.getData<BUFFER_HASH>(), bufread.getData<BUFFER_CELLSTART>(),
			bufread.getData<BUFFER_NEIBSLIST>(), numParticles, particleRangeEnd, dt, step, t, epsilon, deltap, slength,
			influenceRadius, NULL));
	}
	// quadrature_gamma_params (src/cuda/density_sum_params.h:224-262): old gamma, vertex offsets, boundary elements and the
	// list from the state that is read; the NEW positions and the new gamma from the state that is updated
	void integrate_gamma(const BufferList& bufread, BufferList& bufreadUpdate, const uint numParticles,
		const uint particleRangeEnd, const float dt, const int step, const float t, const float epsilon,
		const float slength, const float influenceRadius, const RunMode run_mode)
	{
		const float2 * const *vertPos = bufread.getRawPtr<BUFFER_VERTPOS>();
		if (!vertPos) throw std::invalid_argument("integrate_gamma: BUFFER_VERTPOS missing");
		const float4 *newPos = bufreadUpdate.getConstData<BUFFER_POS>();
		// with moving bodies the quadrature reads the elements of the NEW state (quadrature_gamma_neib_data,
		// src/cuda/density_sum_kernel.cu:713-727) and the library integrates the vertex rows as well
		// (not while repacking: that branch of integrate_gamma_impl, src/cuda/euler.cu:222-239, reads the state's own elements and
		// nobody has written the new ones)
		const float4 *belem = ((m_c->params().simflags & ENABLE_MOVING_BODIES) && run_mode != REPACK)
			? bufreadUpdate.getConstData<BUFFER_BOUNDELEMENTS>() : bufread.getData<BUFFER_BOUNDELEMENTS>();
		sphx_throw(sphx_sa_integrate_gamma(m_c->ctx(), bufreadUpdate.getData<BUFFER_GRADGAMMA>(), bufread.getData<BUFFER_GRADGAMMA>(),
			newPos, belem, vertPos[0], vertPos[1], vertPos[2],
			bufread.getData<BUFFER_INFO>(), bufread.getData<BUFFER_HASH>(), bufread.getData<BUFFER_CELLSTART>(),
			bufread.getData<BUFFER_NEIBSLIST>(), numParticles, particleRangeEnd, dt, step, t, epsilon, slength, influenceRadius,
			run_mode == REPACK ? SPHX_REPACK : SPHX_SIMULATE, NULL));
	}
	// updateDensityDevice (src/cuda/euler.cu:307-326): rho~ += FORCES.w dt for the fluid
	void apply_density_diffusion(const BufferList& bufread, BufferList& bufwrite, const uint numParticles,
		const uint particleRangeEnd, const float dt)
	{
		sphx_throw(sphx_apply_density_diffusion(m_c->ctx(), bufwrite.getData<BUFFER_VEL>(), bufread.getData<BUFFER_FORCES>(),
			bufread.getData<BUFFER_INFO>(), numParticles, particleRangeEnd, dt, NULL));
	}

	void basicstep(const BufferList& bufread, BufferList& bufwrite, const uint numParticles,
		const uint particleRangeEnd, const float dt, const int step, const float t, const float slength,
		const float influenceRadius, const RunMode run_mode)
	{
		if (m_c->params().turbmodel == KEPSILON && run_mode == SIMULATE)
			// keps_euler_params (src/cuda/euler_params.h:101-119): k, epsilon old/new, the eddy viscosity, DKDE; the Eulerian velocity
			sphx_throw(sphx_euler_keps(m_c->ctx(), bufwrite.getData<BUFFER_TKE>(), bufwrite.getData<BUFFER_EPSILON>(),
				bufwrite.getData<BUFFER_TURBVISC>(), bufwrite.getData<BUFFER_EULERVEL>(), bufread.getData<BUFFER_TKE>(),
				bufread.getData<BUFFER_EPSILON>(), bufread.getData<BUFFER_EULERVEL>(), (const float*)bufread.getData<BUFFER_DKDE>(),
				bufread.getData<BUFFER_FORCES>(), bufread.getData<BUFFER_POS>(), bufread.getData<BUFFER_INFO>(),
				numParticles, particleRangeEnd, dt, NULL, 1.0f, NULL));
		if ((m_c->params().simflags & ENABLE_INTERNAL_ENERGY) && run_mode == SIMULATE)
			// energy_euler_params (src/cuda/euler_params.h:121-134): BUFFER_INTERNAL_ENERGY old/new, its rate from the forces pass
			sphx_throw(sphx_euler_internal_energy(m_c->ctx(), bufwrite.getData<BUFFER_INTERNAL_ENERGY>(),
				bufread.getData<BUFFER_INTERNAL_ENERGY>(), bufread.getData<BUFFER_INTERNAL_ENERGY_UPD>(),
				bufread.getData<BUFFER_POS>(), bufread.getData<BUFFER_INFO>(), numParticles, particleRangeEnd, dt, NULL, 1.0f, NULL));
		if (m_c->params().sph_formulation == SPH_GRENIER && run_mode == SIMULATE) {
			// Vol_params (src/cuda/euler_params.h:153-156): BUFFER_VOLUME of the read and of the write list
			sphx_throw(sphx_euler_basicstep_grenier(m_c->ctx(), bufwrite.getData<BUFFER_POS>(), bufwrite.getData<BUFFER_VEL>(),
				bufwrite.getData<BUFFER_VOLUME>(), bufread.getData<BUFFER_POS>(), bufread.getData<BUFFER_VEL>(),
				bufread.getData<BUFFER_VOLUME>(), bufread.getData<BUFFER_INFO>(), bufread.getData<BUFFER_HASH>(),
				bufread.getData<BUFFER_FORCES>(), bufread.getData<BUFFER_XSPH>(),
				numParticles, particleRangeEnd, dt, NULL, 1.0f, step, t, slength, influenceRadius, (int)run_mode, NULL));
			return;
		}
		sphx_throw(sphx_euler_basicstep(m_c->ctx(), bufwrite.getData<BUFFER_POS>(), bufwrite.getData<BUFFER_VEL>(),
			bufread.getData<BUFFER_POS>(), bufread.getData<BUFFER_VEL>(), bufread.getData<BUFFER_INFO>(),
			bufread.getData<BUFFER_HASH>(), bufread.getData<BUFFER_FORCES>(), bufread.getData<BUFFER_XSPH>(),
			numParticles, particleRangeEnd, dt, NULL, 1.0f, step, t, slength, influenceRadius, (int)run_mode, NULL));
		// update_normals (src/cuda/euler_kernel.def:237-254, sa_boundary_moving_euler_params): part of eulerDevice in the reference
		if (m_c->params().boundarytype == SA_BOUNDARY && (m_c->params().simflags & ENABLE_MOVING_BODIES) && run_mode == SIMULATE)
			sphx_throw(sphx_sa_update_normals(m_c->ctx(), bufwrite.getData<BUFFER_BOUNDELEMENTS>(), bufread.getData<BUFFER_BOUNDELEMENTS>(),
				bufread.getData<BUFFER_INFO>(), numParticles, particleRangeEnd, NULL));
	}

	void disableFreeSurfParts(float4 *pos, const particleinfo *info, const uint numParticles, const uint particleRangeEnd)
	{ sphx_throw(sphx_disable_free_surf_parts(m_c->ctx(), pos, info, numParticles, particleRangeEnd, NULL)); }
};

// ---- density filters (CUDAFilterEngine<filtertype, kerneltype, boundarytype>, src/cuda/forces.cu:1008-1147) ----
class HIPFilterEngine : public AbstractFilterEngine
{
	HIPEngineContextPtr m_c;
	FilterType m_type;
public:
	HIPFilterEngine(HIPEngineContextPtr c, FilterType type, uint frequency) :
		AbstractFilterEngine(frequency), m_c(c), m_type(type) {}
	void setconstants() {}
	void getconstants() {}
	void process(const BufferList& bufread, BufferList& bufwrite, uint numParticles, uint particleRangeEnd,
		float slength, float influenceradius)
	{
		sphx_throw(sphx_filter_process(m_c->ctx(), (int)m_type, bufwrite.getData<BUFFER_VEL>(),
			bufread.getData<BUFFER_POS>(), bufread.getData<BUFFER_VEL>(), bufread.getData<BUFFER_INFO>(),
			bufread.getData<BUFFER_HASH>(), bufread.getData<BUFFER_CELLSTART>(), bufread.getData<BUFFER_NEIBSLIST>(),
			numParticles, particleRangeEnd, slength, influenceradius, NULL));
	}
};

// ---- boundary-conditions engine of SA_BOUNDARY (CUDABoundaryConditionsEngine, src/cuda/boundary_conditions.cu) ----
// Solid walls: vertex normals, initial gamma, segment and vertex boundary conditions.  Open boundaries: corner vertices, the initial
// masses of the open vertices, the marking and removal of outgoing particles (sa_io.hip, verified on the GPU); the boundary-condition
// passes with open boundaries (Riemann conditions, mass evolution, particle creation) and the water depth are bound to entry points
// the library refuses until they have passed their GPU tests (SPHX_ERR_UNSUPPORTED -> the exception a "not built" call throws).
#include "engine_boundary_conditions.h"
class HIPBoundaryConditionsEngine : public AbstractBoundaryConditionsEngine
{
	HIPEngineContextPtr m_c;
	uint m_numOpenVertices;      // d_numOpenVertices of cubounds (src/cuda/boundary_conditions_kernel.cu:57): a member, passed by value
public:
	explicit HIPBoundaryConditionsEngine(HIPEngineContextPtr c) : m_c(c), m_numOpenVertices(0) {}

	void uploadNumOpenVertices(const uint &numOpenVertices) { m_numOpenVertices = numOpenVertices; }

	// vel and gGam are updated in place for the boundary elements (src/cuda/boundary_conditions.cu:134-148)
	void saSegmentBoundaryConditions(BufferList &bufwrite, BufferList const& bufread, const uint numParticles,
		const uint particleRangeEnd, const float deltap, const float slength, const float influenceradius,
		const int step, const RunMode run_mode)
	{
		if (m_c->params().turbmodel == KEPSILON && run_mode != REPACK) {
			// sa_segment_bc_params with has_keps (src/cuda/sa_bc_params.h:152-200): k, epsilon and the Eulerian velocity in place
			sphx_throw(sphx_sa_segment_bc_keps(m_c->ctx(), bufwrite.getData<BUFFER_VEL>(), bufwrite.getData<BUFFER_GRADGAMMA>(),
				bufwrite.getData<BUFFER_TKE>(), bufwrite.getData<BUFFER_EPSILON>(), bufwrite.getData<BUFFER_EULERVEL>(),
				bufread.getData<BUFFER_POS>(), bufread.getData<BUFFER_VERTICES>(), bufread.getData<BUFFER_BOUNDELEMENTS>(),
				bufread.getData<BUFFER_INFO>(), bufread.getData<BUFFER_HASH>(), bufread.getData<BUFFER_CELLSTART>(),
				bufread.getData<BUFFER_NEIBSLIST>(), numParticles, particleRangeEnd, deltap, slength, influenceradius,
				step, SPHX_SIMULATE, NULL));
			return;
		}
		if ((m_c->params().simflags & ENABLE_INLET_OUTLET) && run_mode != REPACK) {
			// sa_segment_bc_params with has_io (src/cuda/sa_bc_params.h): the Eulerian velocity in place
			sphx_throw(sphx_sa_segment_bc_io(m_c->ctx(), bufwrite.getData<BUFFER_VEL>(), bufwrite.getData<BUFFER_GRADGAMMA>(),
				bufwrite.getData<BUFFER_EULERVEL>(), bufread.getData<BUFFER_POS>(), bufread.getData<BUFFER_VERTICES>(),
				bufread.getData<BUFFER_BOUNDELEMENTS>(), bufread.getData<BUFFER_INFO>(), bufread.getData<BUFFER_HASH>(),
				bufread.getData<BUFFER_CELLSTART>(), bufread.getData<BUFFER_NEIBSLIST>(), numParticles, particleRangeEnd, step, NULL));
			return;
		}
		sphx_throw(sphx_sa_segment_bc(m_c->ctx(), bufwrite.getData<BUFFER_VEL>(), bufwrite.getData<BUFFER_GRADGAMMA>(),
			bufread.getData<BUFFER_POS>(), bufread.getData<BUFFER_VERTICES>(), bufread.getData<BUFFER_BOUNDELEMENTS>(),
			bufread.getData<BUFFER_INFO>(), bufread.getData<BUFFER_HASH>(), bufread.getData<BUFFER_CELLSTART>(),
			bufread.getData<BUFFER_NEIBSLIST>(), numParticles, particleRangeEnd, deltap, slength, influenceradius,
			step, run_mode == REPACK ? SPHX_REPACK : SPHX_SIMULATE, NULL));
	}

	// marks in BUFFER_VERTICES and BUFFER_GRADGAMMA of the write list, as the reference (src/cuda/boundary_conditions.cu:238-278: the
	// vertices array is taken with MULTISTATE_SAFE because it is shared between states)
	void findOutgoingSegment(BufferList &bufwrite, BufferList const& bufread, const uint numParticles, const uint particleRangeEnd,
		const float, const float, const float influenceradius)
	{
		const float2 * const *vertPos = bufread.getRawPtr<BUFFER_VERTPOS>();
		if (!vertPos)
			throw std::invalid_argument("findOutgoingSegment: BUFFER_VERTPOS missing");
		sphx_throw(sphx_sa_find_outgoing_segment(m_c->ctx(), bufread.getData<BUFFER_POS>(), bufread.getData<BUFFER_VEL>(),
			bufwrite.getData<BUFFER_VERTICES, BufferList::AccessSafety::MULTISTATE_SAFE>(), bufwrite.getData<BUFFER_GRADGAMMA>(),
			vertPos[0], vertPos[1], vertPos[2], bufread.getData<BUFFER_BOUNDELEMENTS>(), bufread.getData<BUFFER_INFO>(),
			bufread.getData<BUFFER_HASH>(), bufread.getData<BUFFER_CELLSTART>(), bufread.getData<BUFFER_NEIBSLIST>(),
			numParticles, particleRangeEnd, influenceradius, NULL));
	}

	// without open boundaries: the density of the vertex particles; no particle is created, *newNumParticles is left alone.
	// With them (sa_io_params + sa_cloning_params, src/cuda/sa_bc_params.h:209-270): the vertex masses into BUFFER_POS of the write
	// list and, in the last step, new particles at *newNumParticles (a device word) in the arrays the reference opens
	// MULTISTATE_SAFE (ids come from BUFFER_NEXTID, unique over the devices: deviceId / numDevices do not enter the kernel)
	void saVertexBoundaryConditions(BufferList &bufwrite, BufferList const& bufread, const uint numParticles,
		const uint particleRangeEnd, const float deltap, const float slength, const float influenceradius,
		const int step, const bool, const float dt, uint *newNumParticles, const uint, const uint, const uint totParticles,
		const RunMode run_mode)
	{
		if ((m_c->params().simflags & ENABLE_INLET_OUTLET) && run_mode != REPACK && m_c->params().turbmodel != KEPSILON) {
			const float2 * const *vertPos = bufread.getRawPtr<BUFFER_VERTPOS>();
			if (!vertPos)
				throw std::invalid_argument("saVertexBoundaryConditions: BUFFER_VERTPOS missing");
			typedef BufferList::AccessSafety AS;
			sphx_throw(sphx_sa_vertex_bc_io(m_c->ctx(), bufwrite.getData<BUFFER_VEL>(), bufread.getData<BUFFER_POS>(),
				bufwrite.getData<BUFFER_POS>(), bufwrite.getData<BUFFER_GRADGAMMA>(), bufwrite.getData<BUFFER_EULERVEL>(),
				bufwrite.getData<BUFFER_FORCES>(), bufwrite.getData<BUFFER_VERTICES, AS::MULTISTATE_SAFE>(),
				bufwrite.getData<BUFFER_BOUNDELEMENTS, AS::MULTISTATE_SAFE>(), vertPos[0], vertPos[1], vertPos[2],
				bufwrite.getData<BUFFER_INFO, AS::MULTISTATE_SAFE>(), bufwrite.getData<BUFFER_HASH, AS::MULTISTATE_SAFE>(),
				bufwrite.getData<BUFFER_NEXTID, AS::MULTISTATE_SAFE>(), newNumParticles, bufread.getData<BUFFER_CELLSTART>(),
				bufread.getData<BUFFER_NEIBSLIST>(), numParticles, particleRangeEnd, totParticles, deltap, dt, step,
				m_numOpenVertices, NULL));
			return;
		}
		// sa_vertex_bc_params takes pos from the read list and vel / gGam from the write list (src/cuda/sa_bc_params.h)
		if (m_c->params().turbmodel == KEPSILON && run_mode != REPACK) {
			sphx_throw(sphx_sa_vertex_bc_keps(m_c->ctx(), bufwrite.getData<BUFFER_VEL>(), bufwrite.getData<BUFFER_GRADGAMMA>(),
				bufwrite.getData<BUFFER_TKE>(), bufwrite.getData<BUFFER_EPSILON>(), bufwrite.getData<BUFFER_EULERVEL>(),
				bufread.getData<BUFFER_VERTICES>(), bufread.getData<BUFFER_BOUNDELEMENTS>(),
				bufread.getData<BUFFER_POS>(), bufread.getData<BUFFER_INFO>(), bufread.getData<BUFFER_HASH>(),
				bufread.getData<BUFFER_CELLSTART>(), bufread.getData<BUFFER_NEIBSLIST>(), numParticles, particleRangeEnd,
				deltap, slength, influenceradius, step, SPHX_SIMULATE, NULL));
			return;
		}
		sphx_throw(sphx_sa_vertex_bc(m_c->ctx(), bufwrite.getData<BUFFER_VEL>(), bufwrite.getData<BUFFER_GRADGAMMA>(),
			bufread.getData<BUFFER_POS>(), bufread.getData<BUFFER_INFO>(), bufread.getData<BUFFER_HASH>(),
			bufread.getData<BUFFER_CELLSTART>(), bufread.getData<BUFFER_NEIBSLIST>(), numParticles, particleRangeEnd,
			deltap, slength, influenceradius, step, run_mode == REPACK ? SPHX_REPACK : SPHX_SIMULATE, NULL));
	}

	void computeVertexNormal(const BufferList& bufread, BufferList& bufwrite, const uint numParticles, const uint particleRangeEnd)
	{
		sphx_throw(sphx_sa_compute_vertex_normal(m_c->ctx(), bufwrite.getData<BUFFER_BOUNDELEMENTS>(),
			bufread.getData<BUFFER_VERTICES>(), bufread.getData<BUFFER_INFO>(), bufread.getData<BUFFER_HASH>(),
			bufread.getData<BUFFER_CELLSTART>(), bufread.getData<BUFFER_NEIBSLIST>(), numParticles, particleRangeEnd, NULL));
	}

	void saInitGamma(const BufferList& bufread, BufferList& bufwrite, const float slength, const float influenceradius,
		const float deltap, const float epsilon, const uint numParticles, const uint particleRangeEnd)
	{
		const float2 * const *vertPos = bufread.getRawPtr<BUFFER_VERTPOS>();
		if (!vertPos)
			throw std::invalid_argument("saInitGamma: BUFFER_VERTPOS missing");
		sphx_throw(sphx_sa_init_gamma(m_c->ctx(), bufwrite.getData<BUFFER_GRADGAMMA>(), bufread.getData<BUFFER_GRADGAMMA>(),
			bufread.getData<BUFFER_POS>(), bufread.getData<BUFFER_BOUNDELEMENTS>(), vertPos[0], vertPos[1], vertPos[2],
			bufread.getData<BUFFER_INFO>(), bufread.getData<BUFFER_HASH>(), bufread.getData<BUFFER_CELLSTART>(),
			bufread.getData<BUFFER_NEIBSLIST>(), slength, influenceradius, deltap, epsilon, numParticles, particleRangeEnd, NULL));
	}

	// BUFFER_FORCES of the write list is the scratch that carries the counts to initIOmass (PredictorCorrectorIntegrator.cc:176-195)
	void initIOmass_vertexCount(BufferList &bufwrite, const BufferList &bufread, const uint numParticles, const uint particleRangeEnd)
	{
		sphx_throw(sphx_sa_init_io_mass_vertex_count(m_c->ctx(), bufread.getData<BUFFER_VERTICES>(), bufread.getData<BUFFER_HASH>(),
			bufread.getData<BUFFER_INFO>(), bufread.getData<BUFFER_CELLSTART>(), bufread.getData<BUFFER_NEIBSLIST>(),
			bufwrite.getData<BUFFER_FORCES>(), bufwrite.getData<BUFFER_FORCES>(), numParticles, particleRangeEnd, NULL));
	}
	void initIOmass(BufferList &bufwrite, const BufferList &bufread, const uint numParticles, const uint particleRangeEnd, const float deltap)
	{
		sphx_throw(sphx_sa_init_io_mass(m_c->ctx(), bufread.getData<BUFFER_POS>(), bufread.getData<BUFFER_FORCES>(),
			bufread.getData<BUFFER_VERTICES>(), bufread.getData<BUFFER_HASH>(), bufread.getData<BUFFER_INFO>(),
			bufread.getData<BUFFER_CELLSTART>(), bufread.getData<BUFFER_NEIBSLIST>(), bufwrite.getData<BUFFER_POS>(),
			numParticles, particleRangeEnd, deltap, NULL));
	}
	void disableOutgoingParts(const BufferList &bufread, BufferList &bufwrite, const uint, const uint particleRangeEnd)
	{
		sphx_throw(sphx_sa_disable_outgoing_parts(m_c->ctx(), bufwrite.getData<BUFFER_POS>(),
			bufwrite.getData<BUFFER_VERTICES, BufferList::AccessSafety::MULTISTATE_SAFE>(), bufread.getData<BUFFER_INFO>(),
			particleRangeEnd, NULL));
	}
	// src/cuda/boundary_conditions.cu:644-663: the per-device maxima to the host and the global ones back
	void downloadIOwaterdepth(uint *h_IOwaterdepth, const uint *d_IOwaterdepth, const uint numOpenBoundaries)
	{ sphx_throw(sphx_memcpy_d2h(h_IOwaterdepth, d_IOwaterdepth, numOpenBoundaries*sizeof(uint))); }
	void uploadIOwaterdepth(const uint *h_IOwaterdepth, uint *d_IOwaterdepth, const uint numOpenBoundaries)
	{ sphx_throw(sphx_memcpy_h2d(d_IOwaterdepth, h_IOwaterdepth, numOpenBoundaries*sizeof(uint))); }
	void saIdentifyCornerVertices(const BufferList &bufread, BufferList &bufwrite, const uint numParticles, const uint particleRangeEnd,
		const float, const float)
	{
		sphx_throw(sphx_sa_identify_corner_vertices(m_c->ctx(), bufread.getData<BUFFER_POS>(), bufwrite.getData<BUFFER_INFO>(),
			bufread.getData<BUFFER_HASH>(), bufread.getData<BUFFER_VERTICES>(), bufread.getData<BUFFER_CELLSTART>(),
			bufread.getData<BUFFER_NEIBSLIST>(), numParticles, particleRangeEnd, NULL));
	}
};

// ---- post-processing (CUDAPostProcessEngine<pptype, kerneltype, boundarytype, simflags>, src/cuda/post_process.cu:88-660) ----
class HIPPostProcessEngine : public AbstractPostProcessEngine
{
	HIPEngineContextPtr m_c;
	PostProcessType m_type;
	mutable float m_cosf, m_cosn;   // PhysParams::cosconeangle{fluid,nonfluid}, read by setconstants

	bool detects() const { return m_type == SURFACE_DETECTION || m_type == INTERFACE_DETECTION; }
public:
	HIPPostProcessEngine(HIPEngineContextPtr c, PostProcessType type, flag_t options) :
		AbstractPostProcessEngine(options), m_c(c), m_type(type), m_cosf(0.86f), m_cosn(0.5f) {}

	void setconstants(const SimParams *, const PhysParams *physparams, idx_t const&) const
	{
		if (physparams) { m_cosf = physparams->cosconeanglefluid; m_cosn = physparams->cosconeanglenonfluid; }
	}
	void getconstants() {}

	flag_t get_written_buffers() const
	{ return m_type == VORTICITY ? BUFFER_VORTICITY : detects() ? (m_options & BUFFER_NORMALS) : BUFFER_NONE; }
	flag_t get_updated_buffers() const
	{ return m_type == TESTPOINTS ? BUFFER_VEL : detects() ? BUFFER_INFO : BUFFER_NONE; }

	void process(const BufferList& bufread, BufferList& bufwrite, uint numParticles, uint particleRangeEnd,
		uint, const GlobalData * const)
	{
		// test points and surface flags are updated in place: GPUWorker puts the buffers named by get_updated_buffers()
		// in the write list too (src/GPUWorker.cc:2559-2572) and the reference takes them from there
		// (src/cuda/post_process.cu:175-177,252-254; INFO is shared between states, hence MULTISTATE_SAFE)
		float4 *velInOut = m_type == TESTPOINTS ? bufwrite.getData<BUFFER_VEL>() : NULL;
		particleinfo *infoInOut = detects() ? bufwrite.getData<BUFFER_INFO, BufferList::MULTISTATE_SAFE>() : NULL;
		sphx_throw(sphx_postprocess(m_c->ctx(), (int)m_type,
			m_type == VORTICITY ? bufwrite.getData<BUFFER_VORTICITY>() : NULL,
			velInOut, infoInOut,
			(detects() && (m_options & BUFFER_NORMALS)) ? bufwrite.getData<BUFFER_NORMALS>() : NULL,
			bufread.getData<BUFFER_POS>(), bufread.getData<BUFFER_VEL>(), bufread.getData<BUFFER_INFO>(),
			bufread.getData<BUFFER_HASH>(), bufread.getData<BUFFER_CELLSTART>(), bufread.getData<BUFFER_NEIBSLIST>(),
			numParticles, particleRangeEnd, m_cosf, m_cosn, NULL));
	}

	// host-side hooks: only FLUX_COMPUTATION has any (src/cuda/post_process.cu:64-75,534-568)
	void hostAllocate(const GlobalData * const) {}
	void hostProcess(const GlobalData * const) {}
	void write(WriterMap, double) {}
};

#endif // SPHX_HIP_ENGINES_H
