// sphx_host.h -- host-side C++ mirror of GPUSPH's engine interfaces for the MI355X engine.
//
// GPUSPH's host code (GPUWorker, Integrator) talks to the device through four abstract classes
//   AbstractNeibsEngine        src/engine_neibs.h:46-107
//   AbstractForcesEngine       src/engine_forces.h:43-180
//   AbstractViscEngine         src/engine_visc.h:42-109
//   AbstractIntegrationEngine  src/engine_integration.h:42-144
// whose methods take (BufferList const& read, BufferList& write, scalars) and pull raw device
// pointers out of the lists by compile-time key (src/buffer.h:629-772).  This header provides
//   * the same interfaces (same method names, argument order, meaning and error behaviour),
//   * a BufferList / typed-key registry with the semantics those engines rely on (NULL for a
//     missing optional buffer, dirty-marking on non-const access),
//   * concrete HIP*Engine classes that unpack the lists and call the C ABI (include/sphx.h).
// Inside the GPUSPH tree the concrete classes derive from the tree's own abstract classes and
// use its own buffer.h instead of the stand-ins below -- see INTEGRATION.md for the diff.
//
// Errors: every C-ABI status is rethrown like CUDA_SAFE_CALL / KERNEL_CHECK_ERROR do
// (src/cuda/cuda_call.h:57-85): std::invalid_argument for inconsistent buffer sets,
// std::runtime_error otherwise.
#ifndef SPHX_HOST_H
#define SPHX_HOST_H

#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>
#include "sphx.h"

typedef unsigned int uint;
typedef ushort4 particleinfo;          // src/particleinfo.h:79
typedef unsigned int hashKey;          // src/hashkey.h:44
typedef unsigned short neibdata;       // src/common_types.h:57
typedef uint64_t flag_t;               // src/common_types.h:83
typedef size_t idx_t;                  // src/common_types.h:75

// ---- buffer keys (names and element types of src/define_buffers.h:48-235; only the keys the
//      hot path touches) ---------------------------------------------------------------------------
#define BUFFER_NONE             ((flag_t)0U)
#define BUFFER_POS              ((flag_t)1U << 1)
#define BUFFER_VEL              ((flag_t)1U << 2)
#define BUFFER_INFO             ((flag_t)1U << 3)
#define BUFFER_HASH             ((flag_t)1U << 4)
#define BUFFER_PARTINDEX        ((flag_t)1U << 5)
#define BUFFER_CELLSTART        ((flag_t)1U << 6)
#define BUFFER_CELLEND          ((flag_t)1U << 7)
#define BUFFER_COMPACT_DEV_MAP  ((flag_t)1U << 8)
#define BUFFER_NEIBSLIST        ((flag_t)1U << 9)
#define BUFFER_FORCES           ((flag_t)1U << 10)
#define BUFFER_RB_FORCES        ((flag_t)1U << 11)
#define BUFFER_RB_TORQUES       ((flag_t)1U << 12)
#define BUFFER_RB_KEYS          ((flag_t)1U << 13)
#define BUFFER_XSPH             ((flag_t)1U << 16)
#define BUFFER_TAU              ((flag_t)1U << 17)
#define BUFFER_VORTICITY        (BUFFER_TAU << 1)
#define BUFFER_NORMALS          (BUFFER_VORTICITY << 1)
#define BUFFER_CFL              ((flag_t)1ULL << 33)
#define BUFFER_CFL_TEMP         ((flag_t)1ULL << 35)
#define BUFFER_SPS_TURBVISC     ((flag_t)1ULL << 37)

template<flag_t Key> struct BufferTraits;
#define SPHX_BUFFER_TRAITS(key, type, n, label) \
	template<> struct BufferTraits<key> { typedef type element_type; enum { num_buffers = n }; \
		static const char *name() { return label; } }
SPHX_BUFFER_TRAITS(BUFFER_POS, float4, 1, "Position");
SPHX_BUFFER_TRAITS(BUFFER_VEL, float4, 1, "Velocity");
SPHX_BUFFER_TRAITS(BUFFER_INFO, particleinfo, 1, "Info");
SPHX_BUFFER_TRAITS(BUFFER_HASH, hashKey, 1, "Hash");
SPHX_BUFFER_TRAITS(BUFFER_PARTINDEX, uint, 1, "Particle Index");
SPHX_BUFFER_TRAITS(BUFFER_CELLSTART, uint, 1, "Cell Start");
SPHX_BUFFER_TRAITS(BUFFER_CELLEND, uint, 1, "Cell End");
SPHX_BUFFER_TRAITS(BUFFER_COMPACT_DEV_MAP, uint, 1, "Compact device map");
SPHX_BUFFER_TRAITS(BUFFER_NEIBSLIST, neibdata, 1, "Neighbor List");
SPHX_BUFFER_TRAITS(BUFFER_FORCES, float4, 1, "Force");
SPHX_BUFFER_TRAITS(BUFFER_RB_FORCES, float4, 1, "Object forces");
SPHX_BUFFER_TRAITS(BUFFER_RB_TORQUES, float4, 1, "Object torques");
SPHX_BUFFER_TRAITS(BUFFER_RB_KEYS, uint, 1, "Object particle key");
SPHX_BUFFER_TRAITS(BUFFER_VORTICITY, float3, 1, "Vorticity");
SPHX_BUFFER_TRAITS(BUFFER_NORMALS, float4, 1, "Normals");
SPHX_BUFFER_TRAITS(BUFFER_XSPH, float4, 1, "XSPH");
SPHX_BUFFER_TRAITS(BUFFER_TAU, float2, 3, "Tau");
SPHX_BUFFER_TRAITS(BUFFER_CFL, float, 1, "CFL array");
SPHX_BUFFER_TRAITS(BUFFER_CFL_TEMP, float, 1, "CFL aux array");
SPHX_BUFFER_TRAITS(BUFFER_SPS_TURBVISC, float, 1, "SPS Turbulent viscosity");

inline void sphx_throw(int rc)
{
	if (rc == SPHX_OK) return;
	const std::string msg = sphx_last_error();
	if (rc == SPHX_ERR_INVALID) throw std::invalid_argument(msg);
	throw std::runtime_error(msg);
}
inline void hip_throw(hipError_t e, const char *what)
{
	if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}

// ---- AbstractBuffer / HIPBuffer (role of src/buffer.h:75-260 and src/cuda/cudabuffer.h:47-131) ----
class AbstractBuffer {
public:
	enum Validity { BUFFER_VALID, BUFFER_DIRTY, BUFFER_INVALID };
	virtual ~AbstractBuffer() {}
	virtual size_t get_element_size() const = 0;
	virtual uint get_array_count() const = 0;
	virtual const char *get_buffer_name() const = 0;
	virtual void *get_buffer(uint idx = 0) = 0;
	virtual const void *get_buffer(uint idx = 0) const = 0;
	virtual size_t alloc(size_t elems) = 0;
	virtual void clobber() = 0;
	Validity validity() const { return m_validity; }
	void mark_valid() { m_validity = BUFFER_VALID; }
	void mark_dirty() { m_validity = BUFFER_DIRTY; }
	void mark_invalid() { m_validity = BUFFER_INVALID; }
	bool is_invalid() const { return m_validity == BUFFER_INVALID; }
protected:
	Validity m_validity = BUFFER_INVALID;
};

template<flag_t Key>
class HIPBuffer : public AbstractBuffer {
	typedef typename BufferTraits<Key>::element_type T;
	enum { N = BufferTraits<Key>::num_buffers };
	T *m_ptr[N];
	size_t m_elems;
	int m_init;
public:
	explicit HIPBuffer(int init = 0) : m_elems(0), m_init(init) { for (int i = 0; i < N; ++i) m_ptr[i] = nullptr; }
	~HIPBuffer() override { for (int i = 0; i < N; ++i) if (m_ptr[i]) (void)hipFree(m_ptr[i]); }
	size_t get_element_size() const override { return sizeof(T); }
	uint get_array_count() const override { return N; }
	const char *get_buffer_name() const override { return BufferTraits<Key>::name(); }
	void *get_buffer(uint idx = 0) override { return idx < (uint)N ? m_ptr[idx] : nullptr; }
	const void *get_buffer(uint idx = 0) const override { return idx < (uint)N ? m_ptr[idx] : nullptr; }
	T **get_raw_ptr() { return m_ptr; }
	size_t alloc(size_t elems) override {
		m_elems = elems;
		for (int i = 0; i < N; ++i) {
			hip_throw(hipMalloc((void**)&m_ptr[i], elems*sizeof(T)), "hipMalloc");
			hip_throw(hipMemset(m_ptr[i], m_init, elems*sizeof(T)), "hipMemset");
		}
		return elems*sizeof(T)*N;
	}
	// cudaMemset to the buffer's init value (0, or 0xFF for neighbour list / cell start / cell end)
	void clobber() override {
		for (int i = 0; i < N; ++i)
			hip_throw(hipMemsetAsync(m_ptr[i], m_init, m_elems*sizeof(T), 0), "hipMemsetAsync");
	}
	size_t size() const { return m_elems; }
};

// ---- BufferList (role of src/buffer.h:560-800): key -> shared buffer, typed access ----
class BufferList {
	typedef std::map<flag_t, std::shared_ptr<AbstractBuffer> > map_type;
	map_type m_map;
	std::set<flag_t> m_updated;
public:
	template<flag_t Key> void addBuffer(int init = 0) { m_map[Key] = std::make_shared<HIPBuffer<Key> >(init); }
	void add(flag_t key, std::shared_ptr<AbstractBuffer> buf) { m_map[key] = buf; }
	std::shared_ptr<AbstractBuffer> operator[](flag_t key) const {
		map_type::const_iterator it = m_map.find(key);
		return it == m_map.end() ? std::shared_ptr<AbstractBuffer>() : it->second;
	}
	bool has(flag_t key) const { return m_map.count(key) != 0; }
	template<flag_t Key> std::shared_ptr<HIPBuffer<Key> > get() const {
		return std::static_pointer_cast<HIPBuffer<Key> >((*this)[Key]);
	}
	// const access: NULL when the buffer is absent ("feature off", src/buffer.h:633-637);
	// reading an INVALID buffer is an error (src/buffer.h:689-695)
	template<flag_t Key> const typename BufferTraits<Key>::element_type *getData(uint idx = 0) const {
		std::shared_ptr<AbstractBuffer> b = (*this)[Key];
		if (!b) return nullptr;
		if (b->is_invalid())
			throw std::invalid_argument(std::string("trying to read invalid buffer ") + b->get_buffer_name());
		return static_cast<const typename BufferTraits<Key>::element_type*>(b->get_buffer(idx));
	}
	// non-const access has the side effects GPUWorker relies on (src/buffer.h:643-674): the buffer
	// becomes DIRTY and is recorded among the updated buffers
	template<flag_t Key> typename BufferTraits<Key>::element_type *getData(uint idx = 0) {
		std::shared_ptr<AbstractBuffer> b = (*this)[Key];
		if (!b) return nullptr;
		b->mark_dirty();
		m_updated.insert(Key);
		return static_cast<typename BufferTraits<Key>::element_type*>(b->get_buffer(idx));
	}
	template<flag_t Key> typename BufferTraits<Key>::element_type **getRawPtr() {
		std::shared_ptr<HIPBuffer<Key> > b = get<Key>();
		if (!b) return nullptr;
		b->mark_dirty();
		m_updated.insert(Key);
		return b->get_raw_ptr();
	}
	const std::set<flag_t> &get_updated_buffers() const { return m_updated; }
	void clear_updated_buffers() { m_updated.clear(); }
	void mark_valid() { for (map_type::iterator it = m_map.begin(); it != m_map.end(); ++it) it->second->mark_valid(); }
};

// src/planes.h:43-61
struct plane_t { float3 normal; int3 gridPos; float3 pos; };
typedef std::vector<plane_t> PlaneList;

// ---- parameter structs: the fields setconstants() consumes (src/simparams.h, src/physparams.h) ----
struct SimParams {
	int kerneltype = SPHX_WENDLAND, sph_formulation = SPHX_SPH_F1, densitydiffusiontype = SPHX_DENSITY_DIFFUSION_NONE;
	int boundarytype = SPHX_LJ_BOUNDARY, rheologytype = SPHX_INVISCID, turbmodel = SPHX_ARTIFICIAL;
	int compvisc = 0, viscmodel = 0, avgop = 0, periodicbound = 0;
	bool is_const_visc = false;   // FullViscSpec::is_const_visc of the framework (src/visc_spec.h:265-282, simparams.h:261-274)
	flag_t simflags = SPHX_ENABLE_DTADAPT;
	double sfactor = 1.3, slength = 0, kernelradius = 2.0, influenceRadius = 0, nlSqInfluenceRadius = 0;
	float dtadaptfactor = 0.3f, densityDiffCoeff = 0, epsxsph = 0.5f, dt = 0;
	uint neiblistsize = 0, neibboundpos = 0, buildneibsfreq = 10, numbodies = 0, numforcesbodies = 0;
	float deltap = 0;          // Problem::m_deltap
	uint repack_maxiter = 2000; float repack_a = 0.1f, repack_alpha = 0.01f;   // src/simparams.h:308-310
	int coord[3] = {1, 2, 0};  // linearisation, yzx by default (src/linearization.h, Makefile:517-519)
};
struct PhysParams {
	std::vector<float> rho0, bcoeff, gammacoeff, sscoeff, sspowercoeff, visccoeff;
	float3 gravity = make_float3(0, 0, -9.81f);
	float artvisccoeff = 0.3f, epsartvisc = 0, smagfactor = 0, kspsfactor = 0;
	float dcoeff = 0, p1coeff = 12, p2coeff = 6, r0 = 0;
	float cosconeanglefluid = 0.86f, cosconeanglenonfluid = 0.5f;
	float partsurf = 0;
	float MK_K = 0, MK_d = 0, MK_beta = 2;
	size_t numFluids() const { return rho0.size(); }
};
struct TimingInfo {   // src/timing.h:43-100
	int numInteractions = 0, maxFluidBoundaryNeibs = 0, maxVertexNeibs = 0, hasTooManyNeibs = -1;
	int hasMaxNeibs[3] = {0, 0, 0};
};
enum RunMode { REPACK = SPHX_REPACK, SIMULATE = SPHX_SIMULATE };

// ---- the four abstract interfaces ----
class AbstractNeibsEngine {
public:
	virtual ~AbstractNeibsEngine() {}
	virtual void setconstants(const SimParams *simparams, const PhysParams *physparams,
		float3 const& worldOrigin, uint3 const& gridSize, float3 const& cellSize, idx_t const& allocatedParticles) = 0;
	virtual void getconstants(SimParams *simparams, PhysParams *physparams) = 0;
	virtual void resetinfo() = 0;
	virtual void getinfo(TimingInfo &timingInfo) = 0;
	virtual void calcHash(const BufferList& bufread, BufferList& bufwrite, const uint numParticles) = 0;
	virtual void fixHash(const BufferList& bufread, BufferList& bufwrite, const uint numParticles) = 0;
	virtual void reorderDataAndFindCellStart(uint *segmentStart, BufferList& sorted_buffers,
		const BufferList& unsorted_buffers, const uint numParticles, uint *newNumParticles) = 0;
	virtual void sort(const BufferList& bufread, BufferList& bufwrite, uint numParticles) = 0;
	virtual void buildNeibsList(const BufferList& bufread, BufferList& bufwrite, const uint numParticles,
		const uint particleRangeEnd, const uint gridCells, const float sqinfluenceradius, const float boundNlSqInflRad) = 0;
};

class AbstractForcesEngine {
public:
	virtual ~AbstractForcesEngine() {}
	virtual void setconstants(const SimParams *simparams, const PhysParams *physparams,
		float3 const& worldOrigin, uint3 const& gridSize, float3 const& cellSize, idx_t const& allocatedParticles) = 0;
	virtual void setgravity(float3 const& gravity) = 0;
	virtual void setplanes(PlaneList const& planes) = 0;
	virtual void setrbcg(const int3 *cgGridPos, const float3 *cgPos, int numbodies) = 0;
	virtual void setrbstart(const int *rbfirstindex, int numbodies) = 0;
	virtual void reduceRbForces(BufferList& bufwrite, uint *lastindex, float3 *totalforce, float3 *totaltorque,
		uint numbodies, uint numBodiesParticles) = 0;
	virtual void bind_textures(const BufferList& bufread, uint numParticles, RunMode run_mode) = 0;
	virtual void unbind_textures(RunMode run_mode) = 0;
	virtual uint round_particles(uint numparts) = 0;
	virtual uint basicstep(const BufferList& bufread, BufferList& bufwrite, uint numParticles, uint fromParticle,
		uint toParticle, float deltap, float slength, float dtadaptfactor, float influenceradius,
		const float epsilon, uint *IOwaterdepth, uint cflOffset, const RunMode run_mode, const int step,
		const float dt, const bool compute_object_forces) = 0;
	virtual uint getFmaxElements(const uint n) = 0;
	virtual uint getFmaxTempElements(const uint n) = 0;
	virtual float dtreduce(float slength, float dtadaptfactor, float sspeed_cfl, float max_kinematic,
		BufferList const& bufread, BufferList& bufwrite, uint numBlocks, uint numParticles) = 0;
};

class AbstractViscEngine {
public:
	virtual ~AbstractViscEngine() {}
	virtual float calc_visc(const BufferList& bufread, BufferList& bufwrite, const uint numParticles,
		const uint particleRangeEnd, const float deltap, const float slength, const float influenceradius,
		const RunMode run_mode) = 0;
};

// src/engine_filter.h:41-78
enum FilterType { FIRST_FILTER = 0, SHEPARD_FILTER = FIRST_FILTER, MLS_FILTER, INVALID_FILTER };   // src/particledefine.h:255-260

class AbstractFilterEngine {
	uint m_frequency;   // frequency of the filter (iterations)
public:
	explicit AbstractFilterEngine(uint _frequency) : m_frequency(_frequency) {}
	virtual ~AbstractFilterEngine() {}
	void set_frequency(uint _frequency) { m_frequency = _frequency; }
	uint frequency() const { return m_frequency; }
	virtual void setconstants() = 0;
	virtual void getconstants() = 0;
	virtual void process(const BufferList& bufread, BufferList& bufwrite, uint numParticles, uint particleRangeEnd,
		float slength, float influenceradius) = 0;
};

// src/engine_postprocess.h:49-105 (host-side hooks hostAllocate/hostProcess/write belong to FLUX/CALC_PRIVATE and
// the writers: not part of the device path)
enum PostProcessType { FIRST_POSTPROC = 0, VORTICITY = FIRST_POSTPROC, TESTPOINTS, SURFACE_DETECTION,
	INTERFACE_DETECTION, FLUX_COMPUTATION, CALC_PRIVATE, INVALID_POSTPROC };   // src/particledefine.h:290-299
#define NO_FLAGS ((flag_t)0)
struct GlobalData;

class AbstractPostProcessEngine {
protected:
	flag_t m_options;
public:
	explicit AbstractPostProcessEngine(flag_t options = NO_FLAGS) : m_options(options) {}
	virtual ~AbstractPostProcessEngine() {}
	flag_t const& get_options() const { return m_options; }
	virtual void setconstants(const SimParams *simparams, const PhysParams *physparams, idx_t const& allocatedParticles) const = 0;
	virtual void getconstants() = 0;
	virtual void process(const BufferList& bufread, BufferList& bufwrite, uint numParticles, uint particleRangeEnd,
		uint deviceIndex, const GlobalData * const gdata) = 0;
	virtual flag_t get_written_buffers() const = 0;
	virtual flag_t get_updated_buffers() const = 0;
};

class AbstractIntegrationEngine {
public:
	virtual ~AbstractIntegrationEngine() {}
	virtual void setconstants(const PhysParams *physparams, float3 const& worldOrigin, uint3 const& gridSize,
		float3 const& cellSize, idx_t const& allocatedParticles, int const& neiblistsize, float const& slength) = 0;
	virtual void setrbcg(const int3 *cgGridPos, const float3 *cgPos, int numbodies) = 0;
	virtual void setrbtrans(const float3 *trans, int numbodies) = 0;
	virtual void setrbsteprot(const float *rot, int numbodies) = 0;
	virtual void setrblinearvel(const float3 *linearvel, int numbodies) = 0;
	virtual void setrbangularvel(const float3 *angularvel, int numbodies) = 0;
	virtual void basicstep(BufferList const& bufread, BufferList& bufwrite, const uint numParticles,
		const uint particleRangeEnd, const float dt, const int step, const float t, const float slength,
		const float influenceradius, const RunMode run_mode) = 0;
	// end of a repacking run (src/engine_integration.h:137-142)
	virtual void disableFreeSurfParts(float4 *pos, const particleinfo *info, const uint numParticles,
		const uint particleRangeEnd) = 0;
};

// ---- shared per-device state of the HIP engines ----
// The reference keeps per-device constants in __constant__ symbols selected by cudaSetDevice; here
// each host thread (= GPUWorker = device) owns one sphx_ctx, looked up by the current HIP device.
class HIPEngineContext {
	std::map<int, sphx_ctx*> m_ctx;
	sphx_params m_params;
public:
	HIPEngineContext() { std::memset(&m_params, 0, sizeof(m_params)); }
	~HIPEngineContext() { for (auto &kv : m_ctx) sphx_destroy(kv.second); }
	sphx_ctx *ctx() {
		int dev = 0;
		hip_throw(hipGetDevice(&dev), "hipGetDevice");
		std::map<int, sphx_ctx*>::iterator it = m_ctx.find(dev);
		if (it != m_ctx.end()) return it->second;
		sphx_ctx *c = nullptr;
		sphx_throw(sphx_create(&c, dev));
		m_ctx[dev] = c;
		return c;
	}
	sphx_params &params() { return m_params; }
	void fill(const SimParams *sp, const PhysParams *pp, float3 const& worldOrigin, uint3 const& gridSize,
		float3 const& cellSize, idx_t allocatedParticles)
	{
		sphx_params &p = m_params;
		p.gridSize[0] = gridSize.x; p.gridSize[1] = gridSize.y; p.gridSize[2] = gridSize.z;
		p.cellSize[0] = cellSize.x; p.cellSize[1] = cellSize.y; p.cellSize[2] = cellSize.z;
		p.worldOrigin[0] = worldOrigin.x; p.worldOrigin[1] = worldOrigin.y; p.worldOrigin[2] = worldOrigin.z;
		for (int a = 0; a < 3; ++a) p.coord[a] = sp->coord[a];
		p.periodic = sp->periodicbound;
		p.neiblistsize = sp->neiblistsize; p.neibboundpos = sp->neibboundpos; p.neiblist_stride = allocatedParticles;
		p.kerneltype = sp->kerneltype; p.sph_formulation = sp->sph_formulation;
		p.densitydiffusiontype = sp->densitydiffusiontype; p.boundarytype = sp->boundarytype;
		p.rheologytype = sp->rheologytype; p.turbmodel = sp->turbmodel;
		p.compvisc = sp->compvisc; p.viscmodel = sp->viscmodel; p.avgop = sp->avgop;
		p.simflags = sp->simflags;
		p.slength = (float)sp->slength; p.kernelradius = (float)sp->kernelradius;
		p.influenceradius = (float)sp->influenceRadius; p.deltap = sp->deltap;
		p.dtadaptfactor = sp->dtadaptfactor; p.densityDiffCoeff = sp->densityDiffCoeff; p.epsxsph = sp->epsxsph;
		p.numfluids = (uint32_t)pp->numFluids();
		for (size_t f = 0; f < pp->numFluids() && f < SPHX_MAX_FLUIDS; ++f) {
			p.rho0[f] = pp->rho0[f]; p.bcoeff[f] = pp->bcoeff[f]; p.gammacoeff[f] = pp->gammacoeff[f];
			p.sscoeff[f] = pp->sscoeff[f]; p.sspowercoeff[f] = pp->sspowercoeff[f];
			p.visccoeff[f] = f < pp->visccoeff.size() ? pp->visccoeff[f] : 0.0f;
		}
		p.gravity[0] = pp->gravity.x; p.gravity[1] = pp->gravity.y; p.gravity[2] = pp->gravity.z;
		p.artvisccoeff = pp->artvisccoeff; p.epsartvisc = pp->epsartvisc;
		p.smagfactor = pp->smagfactor; p.kspsfactor = pp->kspsfactor;
		p.dcoeff = pp->dcoeff; p.p1coeff = pp->p1coeff; p.p2coeff = pp->p2coeff; p.r0 = pp->r0;
		p.repack_a = sp->repack_a; p.repack_alpha = sp->repack_alpha;
		p.is_const_visc = sp->is_const_visc ? 1 : 0; p.partsurf = pp->partsurf;
		p.MK_K = pp->MK_K; p.MK_d = pp->MK_d; p.MK_beta = pp->MK_beta;
		sphx_throw(sphx_set_constants(ctx(), &p));
		sphx_throw(sphx_reserve(ctx(), (uint32_t)allocatedParticles));
	}
};

class HIPNeibsEngine : public AbstractNeibsEngine {
	std::shared_ptr<HIPEngineContext> m_c;
public:
	explicit HIPNeibsEngine(std::shared_ptr<HIPEngineContext> c) : m_c(c) {}
	void setconstants(const SimParams *simparams, const PhysParams *physparams, float3 const& worldOrigin,
		uint3 const& gridSize, float3 const& cellSize, idx_t const& allocatedParticles) override
	{ m_c->fill(simparams, physparams, worldOrigin, gridSize, cellSize, allocatedParticles); }
	void getconstants(SimParams *simparams, PhysParams *) override {
		sphx_params p; sphx_throw(sphx_get_params(m_c->ctx(), &p)); simparams->neibboundpos = p.neibboundpos;
	}
	void resetinfo() override { sphx_throw(sphx_neibs_resetinfo(m_c->ctx(), nullptr)); }
	void getinfo(TimingInfo &ti) override {
		sphx_neibs_info i; sphx_throw(sphx_neibs_getinfo(m_c->ctx(), &i, nullptr));
		ti.numInteractions = i.numInteractions; ti.maxFluidBoundaryNeibs = i.maxFluidBoundaryNeibs;
		ti.maxVertexNeibs = i.maxVertexNeibs; ti.hasTooManyNeibs = i.hasTooManyNeibs;
		for (int k = 0; k < 3; ++k) ti.hasMaxNeibs[k] = i.hasMaxNeibs[k];
	}
	void calcHash(const BufferList& bufread, BufferList& bufwrite, const uint numParticles) override {
		sphx_throw(sphx_calc_hash(m_c->ctx(), bufwrite.getData<BUFFER_POS>(), bufwrite.getData<BUFFER_HASH>(),
			bufwrite.getData<BUFFER_PARTINDEX>(), bufread.getData<BUFFER_INFO>(),
			bufread.getData<BUFFER_COMPACT_DEV_MAP>(), numParticles, nullptr));
	}
	void fixHash(const BufferList& bufread, BufferList& bufwrite, const uint numParticles) override {
		sphx_throw(sphx_fix_hash(m_c->ctx(), bufwrite.getData<BUFFER_HASH>(), bufwrite.getData<BUFFER_PARTINDEX>(),
			bufread.getData<BUFFER_INFO>(), bufread.getData<BUFFER_COMPACT_DEV_MAP>(), numParticles, nullptr));
	}
	void sort(const BufferList&, BufferList& bufwrite, uint numParticles) override {
		sphx_throw(sphx_sort(m_c->ctx(), bufwrite.getData<BUFFER_HASH>(), bufwrite.getData<BUFFER_INFO>(),
			bufwrite.getData<BUFFER_PARTINDEX>(), numParticles, nullptr));
	}
	void reorderDataAndFindCellStart(uint *segmentStart, BufferList& sorted_buffers,
		const BufferList& unsorted_buffers, const uint numParticles, uint *newNumParticles) override
	{
		// INFO/HASH/PARTINDEX were sorted in place and are read here (src/cuda/buildneibs.cu:219-231)
		const BufferList& sorted_ro = sorted_buffers;
		const particleinfo *info = sorted_ro.getData<BUFFER_INFO>();
		const hashKey *hash = sorted_ro.getData<BUFFER_HASH>();
		const uint *partIndex = sorted_ro.getData<BUFFER_PARTINDEX>();
		sphx_throw(sphx_reorder(m_c->ctx(), segmentStart,
			sorted_buffers.getData<BUFFER_CELLSTART>(), sorted_buffers.getData<BUFFER_CELLEND>(),
			sorted_buffers.getData<BUFFER_POS>(), sorted_buffers.getData<BUFFER_VEL>(),
			unsorted_buffers.getData<BUFFER_POS>(), unsorted_buffers.getData<BUFFER_VEL>(),
			info, hash, partIndex, numParticles, newNumParticles, nullptr));
	}
	void buildNeibsList(const BufferList& bufread, BufferList& bufwrite, const uint numParticles,
		const uint particleRangeEnd, const uint gridCells, const float sqinfluenceradius, const float boundNlSqInflRad) override
	{
		sphx_throw(sphx_build_neibs(m_c->ctx(), bufwrite.getData<BUFFER_NEIBSLIST>(),
			bufread.getData<BUFFER_POS>(), bufread.getData<BUFFER_INFO>(), bufread.getData<BUFFER_HASH>(),
			bufread.getData<BUFFER_CELLSTART>(), bufread.getData<BUFFER_CELLEND>(),
			numParticles, particleRangeEnd, gridCells, sqinfluenceradius, boundNlSqInflRad, nullptr));
	}
};

class HIPForcesEngine : public AbstractForcesEngine {
	std::shared_ptr<HIPEngineContext> m_c;
public:
	explicit HIPForcesEngine(std::shared_ptr<HIPEngineContext> c) : m_c(c) {}
	void setconstants(const SimParams *simparams, const PhysParams *physparams, float3 const& worldOrigin,
		uint3 const& gridSize, float3 const& cellSize, idx_t const& allocatedParticles) override
	{ m_c->fill(simparams, physparams, worldOrigin, gridSize, cellSize, allocatedParticles); }
	void setgravity(float3 const& g) override { const float v[3] = { g.x, g.y, g.z }; sphx_throw(sphx_set_gravity(m_c->ctx(), v)); }
	void setplanes(PlaneList const& planes) override {   // CUDAForcesEngine::setplanes, src/cuda/forces.cu:442-448
		std::vector<float> nrm, pos; std::vector<int32_t> gp;
		for (plane_t const& pl : planes) {
			nrm.insert(nrm.end(), { pl.normal.x, pl.normal.y, pl.normal.z });
			gp.insert(gp.end(), { pl.gridPos.x, pl.gridPos.y, pl.gridPos.z });
			pos.insert(pos.end(), { pl.pos.x, pl.pos.y, pl.pos.z });
		}
		sphx_throw(sphx_set_planes(m_c->ctx(), nrm.data(), gp.data(), pos.data(), (int)planes.size()));
	}
	void setrbcg(const int3 *cgGridPos, const float3 *cgPos, int numbodies) override {   // cuforces' copy
		sphx_throw(sphx_set_rb_cg_forces(m_c->ctx(), (const int32_t*)cgGridPos, (const float*)cgPos, numbodies));
	}
	void setrbstart(const int *rbfirstindex, int numbodies) override { sphx_throw(sphx_set_rb_start(m_c->ctx(), rbfirstindex, numbodies)); }
	void reduceRbForces(BufferList& bufwrite, uint *lastindex, float3 *totalforce, float3 *totaltorque,
		uint numbodies, uint numBodiesParticles) override
	{
		const BufferList& ro = bufwrite;
		sphx_throw(sphx_reduce_rb_forces(m_c->ctx(), bufwrite.getData<BUFFER_RB_FORCES>(), bufwrite.getData<BUFFER_RB_TORQUES>(),
			ro.getData<BUFFER_RB_KEYS>(), lastindex, (float*)totalforce, (float*)totaltorque, numbodies, numBodiesParticles, nullptr));
	}
	// textures do not exist on this path: plain loads through L2 (src/cuda/forces.cu:469-532 has no equivalent)
	void bind_textures(const BufferList&, uint, RunMode) override {}
	void unbind_textures(RunMode) override {}
	uint round_particles(uint numparts) override { return sphx_forces_round_particles(numparts); }
	uint getFmaxElements(const uint n) override { return sphx_forces_fmax_elements(n); }
	uint getFmaxTempElements(const uint n) override { return sphx_forces_fmax_temp_elements(n); }
	uint basicstep(const BufferList& bufread, BufferList& bufwrite, uint numParticles, uint fromParticle,
		uint toParticle, float deltap, float slength, float dtadaptfactor, float influenceradius,
		const float, uint *, uint cflOffset, const RunMode run_mode, const int step, const float dt,
		const bool compute_object_forces) override
	{
		uint32_t numBlocks = 0;
		sphx_throw(sphx_forces_basicstep(m_c->ctx(), bufwrite.getData<BUFFER_FORCES>(), bufwrite.getData<BUFFER_CFL>(),
			bufwrite.getData<BUFFER_RB_FORCES>(), bufwrite.getData<BUFFER_RB_TORQUES>(),
			bufread.getData<BUFFER_POS>(), bufread.getData<BUFFER_VEL>(), bufread.getData<BUFFER_INFO>(),
			bufread.getData<BUFFER_HASH>(), bufread.getData<BUFFER_CELLSTART>(), bufread.getData<BUFFER_NEIBSLIST>(),
			bufread.getData<BUFFER_TAU>(0), bufread.getData<BUFFER_TAU>(1), bufread.getData<BUFFER_TAU>(2),
			numParticles, fromParticle, toParticle, deltap, slength, dtadaptfactor, influenceradius,
			cflOffset, run_mode, step, dt, compute_object_forces ? 1 : 0, &numBlocks, nullptr));
		return numBlocks;
	}
	float dtreduce(float slength, float dtadaptfactor, float sspeed_cfl, float max_kinematic,
		BufferList const& bufread, BufferList& bufwrite, uint numBlocks, uint) override
	{
		float dt = 0;
		sphx_throw(sphx_forces_dtreduce(m_c->ctx(), slength, dtadaptfactor, sspeed_cfl, max_kinematic,
			bufread.getData<BUFFER_CFL>(), bufwrite.getData<BUFFER_CFL_TEMP>(), numBlocks, &dt, nullptr));
		return dt;
	}
};

class HIPViscEngine : public AbstractViscEngine {
	std::shared_ptr<HIPEngineContext> m_c;
public:
	explicit HIPViscEngine(std::shared_ptr<HIPEngineContext> c) : m_c(c) {}
	float calc_visc(const BufferList& bufread, BufferList& bufwrite, const uint numParticles,
		const uint particleRangeEnd, const float deltap, const float slength, const float influenceradius,
		const RunMode) override
	{
		sphx_throw(sphx_calc_visc(m_c->ctx(), bufwrite.getData<BUFFER_TAU>(0), bufwrite.getData<BUFFER_TAU>(1),
			bufwrite.getData<BUFFER_TAU>(2), bufwrite.getData<BUFFER_SPS_TURBVISC>(),
			bufread.getData<BUFFER_POS>(), bufread.getData<BUFFER_VEL>(), bufread.getData<BUFFER_INFO>(),
			bufread.getData<BUFFER_HASH>(), bufread.getData<BUFFER_CELLSTART>(), bufread.getData<BUFFER_NEIBSLIST>(),
			numParticles, particleRangeEnd, deltap, slength, influenceradius, nullptr));
		return NAN;   // SPS: no kinematic-viscosity maximum to report (src/cuda/visc.cu:232)
	}
};

class HIPPredCorrEngine : public AbstractIntegrationEngine {
	std::shared_ptr<HIPEngineContext> m_c;
public:
	explicit HIPPredCorrEngine(std::shared_ptr<HIPEngineContext> c) : m_c(c) {}
	void setconstants(const PhysParams *, float3 const&, uint3 const&, float3 const&, idx_t const&, int const&, float const&) override {}
	void setrbcg(const int3 *cgGridPos, const float3 *cgPos, int numbodies) override {   // cueuler's copy
		sphx_throw(sphx_set_rb_cg_integration(m_c->ctx(), (const int32_t*)cgGridPos, (const float*)cgPos, numbodies));
	}
	void setrbtrans(const float3 *trans, int n) override { sphx_throw(sphx_set_rb_motion(m_c->ctx(), (const float*)trans, nullptr, nullptr, nullptr, n)); }
	void setrbsteprot(const float *rot, int n) override { sphx_throw(sphx_set_rb_motion(m_c->ctx(), nullptr, rot, nullptr, nullptr, n)); }
	void setrblinearvel(const float3 *v, int n) override { sphx_throw(sphx_set_rb_motion(m_c->ctx(), nullptr, nullptr, (const float*)v, nullptr, n)); }
	void setrbangularvel(const float3 *w, int n) override { sphx_throw(sphx_set_rb_motion(m_c->ctx(), nullptr, nullptr, nullptr, (const float*)w, n)); }
	void basicstep(BufferList const& bufread, BufferList& bufwrite, const uint numParticles,
		const uint particleRangeEnd, const float dt, const int step, const float t, const float slength,
		const float influenceradius, const RunMode run_mode) override
	{
		sphx_throw(sphx_euler_basicstep(m_c->ctx(), bufwrite.getData<BUFFER_POS>(), bufwrite.getData<BUFFER_VEL>(),
			bufread.getData<BUFFER_POS>(), bufread.getData<BUFFER_VEL>(), bufread.getData<BUFFER_INFO>(),
			bufread.getData<BUFFER_HASH>(), bufread.getData<BUFFER_FORCES>(), bufread.getData<BUFFER_XSPH>(),
			numParticles, particleRangeEnd, dt, nullptr, 1.0f, step, t, slength, influenceradius, run_mode, nullptr));
	}
	void disableFreeSurfParts(float4 *pos, const particleinfo *info, const uint numParticles,
		const uint particleRangeEnd) override
	{
		sphx_throw(sphx_disable_free_surf_parts(m_c->ctx(), pos, info, numParticles, particleRangeEnd, nullptr));
	}
};

// CUDAFilterEngine<filtertype, kerneltype, boundarytype> (src/cuda/forces.cu:1008-1147): the kernel and boundary
// type are run-time options of the context here, so one class serves both filters
class HIPFilterEngine : public AbstractFilterEngine {
	std::shared_ptr<HIPEngineContext> m_c;
	FilterType m_type;
public:
	HIPFilterEngine(std::shared_ptr<HIPEngineContext> c, FilterType type, uint frequency) :
		AbstractFilterEngine(frequency), m_c(c), m_type(type) {}
	void setconstants() override {}
	void getconstants() override {}
	void process(const BufferList& bufread, BufferList& bufwrite, uint numParticles, uint particleRangeEnd,
		float slength, float influenceradius) override
	{
		sphx_throw(sphx_filter_process(m_c->ctx(), (int)m_type, bufwrite.getData<BUFFER_VEL>(),
			bufread.getData<BUFFER_POS>(), bufread.getData<BUFFER_VEL>(), bufread.getData<BUFFER_INFO>(),
			bufread.getData<BUFFER_HASH>(), bufread.getData<BUFFER_CELLSTART>(), bufread.getData<BUFFER_NEIBSLIST>(),
			numParticles, particleRangeEnd, slength, influenceradius, nullptr));
	}
};

// CUDAPostProcessEngine<pptype, kerneltype, boundarytype, simflags> (src/cuda/post_process.cu:88-300)
class HIPPostProcessEngine : public AbstractPostProcessEngine {
	std::shared_ptr<HIPEngineContext> m_c;
	PostProcessType m_type;
	mutable float m_cosf = 0.86f, m_cosn = 0.5f;   // PhysParams::cosconeangle{fluid,nonfluid}, src/physparams.h:418-419
public:
	HIPPostProcessEngine(std::shared_ptr<HIPEngineContext> c, PostProcessType type, flag_t options) :
		AbstractPostProcessEngine(options), m_c(c), m_type(type) {}
	void setconstants(const SimParams *, const PhysParams *pp, idx_t const&) const override {
		if (pp) { m_cosf = pp->cosconeanglefluid; m_cosn = pp->cosconeanglenonfluid; }
	}
	void getconstants() override {}
	bool detects() const { return m_type == SURFACE_DETECTION || m_type == INTERFACE_DETECTION; }
	flag_t get_written_buffers() const override {
		return m_type == VORTICITY ? BUFFER_VORTICITY : detects() ? (m_options & BUFFER_NORMALS) : BUFFER_NONE;
	}
	flag_t get_updated_buffers() const override {
		return m_type == TESTPOINTS ? BUFFER_VEL : detects() ? BUFFER_INFO : BUFFER_NONE;
	}
	void process(const BufferList& bufread, BufferList& bufwrite, uint numParticles, uint particleRangeEnd,
		uint, const GlobalData * const) override
	{
		sphx_throw(sphx_postprocess(m_c->ctx(), (int)m_type,
			m_type == VORTICITY ? bufwrite.getData<BUFFER_VORTICITY>() : nullptr,
			m_type == TESTPOINTS ? bufwrite.getData<BUFFER_VEL>() : nullptr,
			detects() ? bufwrite.getData<BUFFER_INFO>() : nullptr,
			(detects() && (m_options & BUFFER_NORMALS)) ? bufwrite.getData<BUFFER_NORMALS>() : nullptr,
			bufread.getData<BUFFER_POS>(), bufread.getData<BUFFER_VEL>(), bufread.getData<BUFFER_INFO>(),
			bufread.getData<BUFFER_HASH>(), bufread.getData<BUFFER_CELLSTART>(), bufread.getData<BUFFER_NEIBSLIST>(),
			numParticles, particleRangeEnd, m_cosf, m_cosn, nullptr));
	}
};

// ---- factory: the engine-owning part of SimFramework (src/simframework.h:65-136) ----
class HIPSimFramework {
	std::shared_ptr<HIPEngineContext> m_c;
	std::unique_ptr<AbstractNeibsEngine> m_neibs;
	std::unique_ptr<AbstractForcesEngine> m_forces;
	std::unique_ptr<AbstractViscEngine> m_visc;
	std::unique_ptr<AbstractIntegrationEngine> m_integration;
public:
	HIPSimFramework() : m_c(std::make_shared<HIPEngineContext>()),
		m_neibs(new HIPNeibsEngine(m_c)), m_forces(new HIPForcesEngine(m_c)),
		m_visc(new HIPViscEngine(m_c)), m_integration(new HIPPredCorrEngine(m_c)) {}
	AbstractNeibsEngine *getNeibsEngine() { return m_neibs.get(); }
	AbstractForcesEngine *getForcesEngine() { return m_forces.get(); }
	AbstractViscEngine *getViscEngine() { return m_visc.get(); }
	AbstractIntegrationEngine *getIntegrationEngine() { return m_integration.get(); }
	// SimFramework::addFilterEngine -> newFilterEngine (src/cuda/cudasimframework.cu:236-251): caller owns the engine
	AbstractFilterEngine *newFilterEngine(FilterType filtertype, int frequency)
	{
		if (filtertype != SHEPARD_FILTER && filtertype != MLS_FILTER)
			throw std::runtime_error("Invalid filter type");
		return new HIPFilterEngine(m_c, filtertype, (uint)frequency);
	}
	// newPostProcessEngine (src/cuda/cudasimframework.cu:249-268): the engines not built here throw like the reference's default
	AbstractPostProcessEngine *newPostProcessEngine(PostProcessType pptype, flag_t options = NO_FLAGS)
	{
		if (pptype != VORTICITY && pptype != TESTPOINTS && pptype != SURFACE_DETECTION && pptype != INTERFACE_DETECTION)
			throw std::runtime_error("Unknown filter type");
		return new HIPPostProcessEngine(m_c, pptype, options);
	}
};

#endif // SPHX_HOST_H
