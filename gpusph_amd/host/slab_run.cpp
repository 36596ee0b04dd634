// slab_run -- a decomposed run driven from C++ through the C ABI alone: one worker thread per slab, each with its own sphx_ctx,
// the way GPUWorker threads drive their devices (src/GPUWorker.cc: one thread per device, importExternalCells / transferBursts
// between them).  What a worker does per step is the command stream of the reference's integrator for the plain WCSPH path
// (DYN / LJ boundaries, no filters, no bodies):
//   neighbour phase  CALCHASH (or fixHash), SORT, REORDER with the device map -> UPDATE_SEGMENTS, CROP, APPEND_EXTERNAL (the edge
//                    layers of the neighbours arrive as halo rows), cell ranges of the halo, BUILDNEIBS      src/Integrator.cc:94-250
//   predictor / corrector  FORCES on the internal particles, UPDATE_EXTERNAL of BUFFER_FORCES, dt reduction, EULER on every row
//                    (the halo copies are integrated with the forces they were sent)          PredictorCorrectorIntegrator.cc:386-685
//   epilogue         t += dt, dt = min over the devices                                                    src/GPUSPH.cc:650-657
// It is the C++ twin of gpusph_amd/multigpu.py's plain path; tests/test_gpu_halo.py runs it on a dam break and holds the
// particles it leaves against the single-domain run of the Python driver: bit for bit with the gather kernels.
// All slabs share device 0 on a one-GPU box (the copies of the thread transport are then same-device copies); on a node every
// worker would sphx_create on its own device, nothing else changes.  Needs libsphx.so only (no GPUSPH tree).
//
//   slab_run <case.bin> <out prefix>          writes <out prefix>.<rank>.bin
//
// case.bin (little endian; written by the test from the Python problem mirror):
//   u32 magic 'SLB1', world, steps, n, ncells, alloc, plane (cells per COORD3 plane), gs3, neiblistsize, sizeof(sphx_params)
//   sphx_params
//   f32 dt0, sspeed_cfl, max_kinvisc, sq_nl_radius; u32 buildneibsfreq, ring (1: the split axis is periodic, the slabs form a ring)
//   u32 lo[world], hi[world]                   COORD3 planes [lo, hi) of every slab
//   f32 pos[n][4], vel[n][4]; u16 info[n][4]; u32 hash[n]
// <out>.<rank>.bin: u32 n_int, f32 dt, f64 t, then pos, vel, info, hash of the n_int internal particles
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include "sphx.h"

static int g_fail = 0;
#define CALL(x) do { const int rc_ = (x); if (rc_ != 0) { std::fprintf(stderr, "slab_run: %s -> %d: %s\n", #x, rc_, sphx_last_error()); \
	__sync_fetch_and_add(&g_fail, 1); std::exit(2); } } while (0)

static const uint32_t CELLTYPE_INNER = 0, CELLTYPE_INNER_EDGE = 1, CELLTYPE_OUTER_EDGE = 2, CELLTYPE_OUTER = 3;
static const uint32_t CELLTYPE_BITMASK = 0x3FFFFFFFu, EMPTY_SEGMENT = 0xFFFFFFFFu;

struct Case {
	uint32_t world, steps, n, ncells, alloc, plane, gs3, neiblistsize, buildneibsfreq, ring;
	sphx_params params;
	float dt0, sspeed_cfl, max_kinvisc, sq_nl_radius;
	std::vector<uint32_t> lo, hi;
	std::vector<float> pos, vel;
	std::vector<uint16_t> info;
	std::vector<uint32_t> hash;
};

template<class T> static void rd(FILE *f, T *p, size_t count)
{
	if (std::fread(p, sizeof(T), count, f) != count) { std::fprintf(stderr, "slab_run: short case file\n"); std::exit(2); }
}

static Case read_case(const char *path)
{
	Case c;
	FILE *f = std::fopen(path, "rb");
	if (!f) { std::fprintf(stderr, "slab_run: cannot open %s\n", path); std::exit(2); }
	uint32_t h[10];
	rd(f, h, 10);
	if (h[0] != 0x31424C53u || h[9] != sizeof(sphx_params)) {
		std::fprintf(stderr, "slab_run: %s is not a case of this build (magic %08x, sphx_params %u vs %zu bytes)\n", path, h[0], h[9], sizeof(sphx_params));
		std::exit(2);
	}
	c.world = h[1]; c.steps = h[2]; c.n = h[3]; c.ncells = h[4]; c.alloc = h[5]; c.plane = h[6]; c.gs3 = h[7]; c.neiblistsize = h[8];
	rd(f, &c.params, 1);
	float s[4];
	rd(f, s, 4);
	c.dt0 = s[0]; c.sspeed_cfl = s[1]; c.max_kinvisc = s[2]; c.sq_nl_radius = s[3];
	rd(f, &c.buildneibsfreq, 1);
	rd(f, &c.ring, 1);
	c.lo.resize(c.world); c.hi.resize(c.world);
	rd(f, c.lo.data(), c.world); rd(f, c.hi.data(), c.world);
	c.pos.resize(4*(size_t)c.n); c.vel.resize(4*(size_t)c.n); c.info.resize(4*(size_t)c.n); c.hash.resize(c.n);
	rd(f, c.pos.data(), c.pos.size()); rd(f, c.vel.data(), c.vel.size()); rd(f, c.info.data(), c.info.size()); rd(f, c.hash.data(), c.hash.size());
	std::fclose(f);
	return c;
}

// everything one slab holds on its device
struct Worker {
	const Case *c;
	int rank, world;
	sphx_ctx *ctx;
	sphx_halo *halo;
	uint32_t n_local, n_int, edge_start;
	uint32_t sendL[2], sendR[2], recvL[2], recvR[2];      // [start, count] in rows
	void *pos, *pos2, *vel, *vel2, *info, *forces;
	uint32_t *hash, *partindex, *cellStart, *cellEnd, *devmap, *segmentStart, *newNum;
	uint16_t *neibslist;
	float *cfl, *cflTemp, *d_dt, *d_dt_next;
	double *d_t;
	uint32_t cflElems;
	uint64_t iterations;
	const void *eulers;      // the velocity buffer the last Euler step wrote, untouched since (sphx_eos_rows_current), or NULL
};

// the neighbours of a slab; on a periodic split axis the first and the last slab are neighbours through the periodic face
static int left_of(const Worker &w) { return w.rank > 0 ? w.rank - 1 : (w.c->ring && w.world > 1 ? w.world - 1 : -1); }
static int right_of(const Worker &w) { return w.rank < w.world - 1 ? w.rank + 1 : (w.c->ring && w.world > 1 ? 0 : -1); }

// UPDATE_EXTERNAL of row buffers: my edge layers out, my neighbours' edge layers into my halo rows
static void exchange(Worker &w, int nbuf, void *const *bufs, const uint32_t *rowBytes)
{
	CALL(sphx_halo_exchange(w.halo, nbuf, bufs, rowBytes,
		left_of(w), w.sendL[0], w.sendL[1], w.recvL[0], w.recvL[1],
		right_of(w), w.sendR[0], w.sendR[1], w.recvR[0], w.recvR[1], nullptr));
}

// UPDATE_SEGMENTS + CROP + APPEND_EXTERNAL (src/Integrator.cc:170-230)
static void update_segments_and_halo(Worker &w)
{
	const Case &c = *w.c;
	uint32_t seg[5];
	CALL(sphx_memcpy_d2h(seg, w.segmentStart, 4*sizeof(uint32_t)));
	CALL(sphx_memcpy_d2h(&seg[4], w.newNum, sizeof(uint32_t)));
	for (int i = 3; i >= 0; --i) if (seg[i] == EMPTY_SEGMENT) seg[i] = seg[i + 1];      // an empty segment starts where the next one does
	const uint32_t edge_start = seg[CELLTYPE_INNER_EDGE], n_int = seg[CELLTYPE_OUTER_EDGE];   // internal = inner + inner edge; the rest is cropped
	w.edge_start = edge_start; w.n_int = n_int;
	const int left = left_of(w), right = right_of(w);
	// the inner-edge segment is sorted by hash: the plane that faces the left neighbour comes first
	uint32_t split;
	if (left >= 0 && right >= 0) {
		std::vector<uint32_t> h(n_int - edge_start);
		if (!h.empty()) CALL(sphx_memcpy_d2h(h.data(), w.hash + edge_start, h.size()*sizeof(uint32_t)));
		const uint32_t key = (CELLTYPE_INNER_EDGE << 30) | ((c.hi[w.rank] - 1u)*c.plane);
		split = edge_start + (uint32_t)(std::lower_bound(h.begin(), h.end(), key) - h.begin());
	} else
		split = left >= 0 ? n_int : edge_start;
	w.sendL[0] = edge_start; w.sendL[1] = left >= 0 ? split - edge_start : 0u;
	w.sendR[0] = split;      w.sendR[1] = right >= 0 ? n_int - split : 0u;
	// the sizes of the layers my neighbours send me
	const uint64_t mine[2] = { w.sendL[1], w.sendR[1] };
	std::vector<uint64_t> all(2*(size_t)w.world);
	CALL(sphx_halo_allgather_u64x2(w.halo, mine, all.data(), nullptr));
	const uint32_t rl = left >= 0 ? (uint32_t)all[2*left + 1] : 0u;        // the left neighbour's right layer
	const uint32_t rr = right >= 0 ? (uint32_t)all[2*right] : 0u;         // the right neighbour's left layer
	if ((uint64_t)n_int + rl + rr > c.alloc) {
		std::fprintf(stderr, "slab_run: rank %d: %u internal + %u halo particles exceed the %u allocated\n", w.rank, n_int, rl + rr, c.alloc);
		std::exit(2);
	}
	w.recvL[0] = n_int; w.recvL[1] = rl;
	w.recvR[0] = n_int + rl; w.recvR[1] = rr;
	w.n_local = n_int + rl + rr;
	void *state[4] = { w.pos, w.vel, w.info, w.hash };
	const uint32_t rows[4] = { 16, 16, 8, 4 };
	exchange(w, 4, state, rows);
	CALL(sphx_device_synchronize());
	if (w.n_local > n_int) {      // what arrived are OUTER_EDGE cells here
		std::vector<uint32_t> h(w.n_local - n_int);
		CALL(sphx_memcpy_d2h(h.data(), w.hash + n_int, h.size()*sizeof(uint32_t)));
		for (size_t i = 0; i < h.size(); ++i) h[i] = (h[i] & CELLTYPE_BITMASK) | (CELLTYPE_OUTER_EDGE << 30);
		CALL(sphx_memcpy_h2d(w.hash + n_int, h.data(), h.size()*sizeof(uint32_t)));
	}
	// cell ranges: forget everything outside my own planes, then index the fresh halo
	const size_t lo = (size_t)c.lo[w.rank]*c.plane, hi = (size_t)c.hi[w.rank]*c.plane;
	uint32_t *tabs[2] = { w.cellStart, w.cellEnd };
	for (int k = 0; k < 2; ++k) {
		if (lo > 0) CALL(sphx_memset(tabs[k], 0xFF, lo*sizeof(uint32_t)));
		if (hi < c.ncells) CALL(sphx_memset(tabs[k] + hi, 0xFF, (c.ncells - hi)*sizeof(uint32_t)));
	}
	CALL(sphx_find_cell_start(w.ctx, w.cellStart, w.cellEnd, w.hash, n_int, w.n_local, nullptr));
}

static void build_neibs(Worker &w)
{
	const Case &c = *w.c;
	const uint32_t n = w.n_local;
	w.eulers = nullptr;      // the sort rewrites the buffers
	if (w.iterations == 0) CALL(sphx_fix_hash(w.ctx, w.hash, w.partindex, w.info, w.devmap, n, nullptr));
	else CALL(sphx_calc_hash(w.ctx, w.pos, w.hash, w.partindex, w.info, w.devmap, n, nullptr));
	CALL(sphx_sort(w.ctx, w.hash, w.info, w.partindex, n, nullptr));
	CALL(sphx_memset_async(w.cellStart, 0xFF, c.ncells*sizeof(uint32_t), nullptr));
	CALL(sphx_memset_async(w.cellEnd, 0xFF, c.ncells*sizeof(uint32_t), nullptr));
	CALL(sphx_reorder(w.ctx, w.world > 1 ? w.segmentStart : nullptr, w.cellStart, w.cellEnd, w.pos2, w.vel2, w.pos, w.vel,
		w.info, w.hash, w.partindex, n, w.newNum, nullptr));
	std::swap(w.pos, w.pos2); std::swap(w.vel, w.vel2);
	if (w.world == 1) {
		uint32_t nn = 0;
		CALL(sphx_memcpy_d2h(&nn, w.newNum, sizeof(uint32_t)));
		w.n_local = w.n_int = w.edge_start = nn;
	} else
		update_segments_and_halo(w);
	CALL(sphx_neibs_resetinfo(w.ctx, nullptr));
	CALL(sphx_build_neibs(w.ctx, w.neibslist, w.pos, w.info, w.hash, w.cellStart, w.cellEnd, w.n_local, w.n_int, c.ncells,
		c.sq_nl_radius, c.sq_nl_radius, nullptr));
}

// FORCES on the internal particles (edge stripe, then inner stripe), UPDATE_EXTERNAL of the forces, dt of this pass
static void forces_pass(Worker &w, const void *pos, const void *vel, int combine_min, int step)
{
	const sphx_params &P = w.c->params;
	CALL(sphx_memset_async(w.cfl, 0, w.cflElems*sizeof(float), nullptr));
	uint32_t nb1 = 0, nb2 = 0;
	bool vouch = w.eulers == vel;      // nothing has touched the densities since the Euler step: its EOS rows stand
	auto launch = [&](uint32_t from, uint32_t to, uint32_t off, uint32_t *nb) {
		if (vouch) CALL(sphx_eos_rows_current(w.ctx, vel, w.n_local));
		vouch = true;                  // the second stripe reads what the first one read
		CALL(sphx_forces_basicstep(w.ctx, w.forces, w.cfl, nullptr, nullptr, pos, vel, w.info, w.hash, w.cellStart, w.neibslist,
			nullptr, nullptr, nullptr, nullptr, w.n_local, from, to, P.deltap, P.slength, P.dtadaptfactor, P.influenceradius,
			off, SPHX_SIMULATE, step, 0.0f, 0, nb, nullptr));
	};
	if (w.world > 1 && w.n_int > w.edge_start) {
		launch(w.edge_start, w.n_int, 0u, &nb1);
		launch(0u, w.edge_start, nb1, &nb2);
	} else
		launch(0u, w.n_int, 0u, &nb1);
	if (w.world > 1) {
		void *bufs[1] = { w.forces };
		const uint32_t rows[1] = { 16 };
		exchange(w, 1, bufs, rows);
	}
	CALL(sphx_forces_dtreduce_device(w.ctx, P.slength, P.dtadaptfactor, w.c->sspeed_cfl, w.c->max_kinvisc, w.cfl, w.cflTemp,
		nb1 + nb2, w.d_dt_next, combine_min, nullptr));
}

static void step(Worker &w)
{
	const sphx_params &P = w.c->params;
	if (w.iterations % w.c->buildneibsfreq == 0) build_neibs(w);
	const uint32_t n = w.n_local;
	auto euler = [&](float dt_scale, int stepnum) {
		CALL(sphx_euler_basicstep(w.ctx, w.pos2, w.vel2, w.pos, w.vel, w.info, w.hash, w.forces, nullptr, n, n, 0.0f, w.d_dt, dt_scale,
			stepnum, 0.0f, P.slength, P.influenceradius, SPHX_SIMULATE, nullptr));
		w.eulers = w.vel2;
	};
	forces_pass(w, w.pos, w.vel, 0, 1);          // predictor: forces(n) -> n* = n + dt/2 f
	euler(0.5f, 1);
	forces_pass(w, w.pos2, w.vel2, 1, 2);        // corrector: forces(n*) -> n+1 = n + dt f*
	euler(1.0f, 2);
	std::swap(w.pos, w.pos2); std::swap(w.vel, w.vel2);
	CALL(sphx_time_advance(w.ctx, w.d_t, w.d_dt, nullptr));
	if (w.world > 1) CALL(sphx_halo_allreduce_min_f32(w.halo, w.d_dt_next, nullptr));
	std::swap(w.d_dt, w.d_dt_next);
	++w.iterations;
}

template<class T> static T *dev_alloc(size_t count, int fill = 0)
{
	void *p = nullptr;
	CALL(sphx_malloc(&p, count*sizeof(T)));
	CALL(sphx_memset(p, fill, count*sizeof(T)));
	return (T*)p;
}

static void worker(const Case *c, int rank, sphx_halo_group *group, const std::string &prefix)
{
	Worker w;
	std::memset(&w, 0, sizeof(w));
	w.c = c; w.rank = rank; w.world = (int)c->world;
	CALL(sphx_create(&w.ctx, 0));
	CALL(sphx_set_constants(w.ctx, &c->params));
	CALL(sphx_reserve(w.ctx, c->alloc));
	CALL(sphx_eos_rows_follow_euler(w.ctx, 1));
	if (w.world > 1) CALL(sphx_halo_create_threads(group, w.ctx, rank, &w.halo));
	// the particles this slab starts with: its own planes plus one plane of each neighbour (they are sorted out by the first
	// neighbour phase); the device map: CELLTYPE of every cell as seen from here (fillDeviceMapByAxis, src/ProblemCore.cc:1061-1116)
	const uint32_t lo = c->lo[rank], hi = c->hi[rank];
	std::vector<uint32_t> planeType(c->gs3, w.world > 1 ? CELLTYPE_OUTER : CELLTYPE_INNER);
	if (w.world > 1) {
		const bool hasLeft = rank > 0 || c->ring, hasRight = rank < w.world - 1 || c->ring;
		for (uint32_t p = lo; p < hi; ++p) planeType[p] = CELLTYPE_INNER;
		if (hasLeft) { planeType[lo] = CELLTYPE_INNER_EDGE; planeType[(lo + c->gs3 - 1u) % c->gs3] = CELLTYPE_OUTER_EDGE; }
		if (hasRight) { planeType[hi - 1u] = CELLTYPE_INNER_EDGE; planeType[hi % c->gs3] = CELLTYPE_OUTER_EDGE; }
	}
	std::vector<float> pos, vel; std::vector<uint16_t> info; std::vector<uint32_t> hash;
	for (uint32_t i = 0; i < c->n; ++i) {
		const uint32_t plane = (c->hash[i] & CELLTYPE_BITMASK)/c->plane;
		if (planeType[plane] == CELLTYPE_OUTER) continue;
		pos.insert(pos.end(), &c->pos[4*(size_t)i], &c->pos[4*(size_t)i] + 4);
		vel.insert(vel.end(), &c->vel[4*(size_t)i], &c->vel[4*(size_t)i] + 4);
		info.insert(info.end(), &c->info[4*(size_t)i], &c->info[4*(size_t)i] + 4);
		hash.push_back(c->hash[i]);
	}
	const uint32_t n0 = (uint32_t)hash.size(), A = c->alloc;
	if (n0 > A) { std::fprintf(stderr, "slab_run: rank %d starts with %u particles, %u allocated\n", rank, n0, A); std::exit(2); }
	w.n_local = w.n_int = w.edge_start = n0;
	w.pos = dev_alloc<float>(4*(size_t)A); w.pos2 = dev_alloc<float>(4*(size_t)A);
	w.vel = dev_alloc<float>(4*(size_t)A); w.vel2 = dev_alloc<float>(4*(size_t)A);
	w.info = dev_alloc<uint16_t>(4*(size_t)A); w.forces = dev_alloc<float>(4*(size_t)A);
	w.hash = dev_alloc<uint32_t>(A); w.partindex = dev_alloc<uint32_t>(A);
	w.cellStart = dev_alloc<uint32_t>(c->ncells, 0xFF); w.cellEnd = dev_alloc<uint32_t>(c->ncells, 0xFF);
	w.neibslist = dev_alloc<uint16_t>((size_t)c->neiblistsize*A);
	w.cflElems = sphx_forces_fmax_elements(A) + 8u;
	w.cfl = dev_alloc<float>(w.cflElems);
	w.cflTemp = dev_alloc<float>(std::max(sphx_forces_fmax_temp_elements(w.cflElems), 4u));
	w.segmentStart = dev_alloc<uint32_t>(4); w.newNum = dev_alloc<uint32_t>(1);
	w.d_dt = dev_alloc<float>(1); w.d_dt_next = dev_alloc<float>(1); w.d_t = dev_alloc<double>(1);
	CALL(sphx_memcpy_h2d(w.d_dt, &c->dt0, sizeof(float))); CALL(sphx_memcpy_h2d(w.d_dt_next, &c->dt0, sizeof(float)));
	CALL(sphx_memcpy_h2d(w.pos, pos.data(), pos.size()*sizeof(float))); CALL(sphx_memcpy_h2d(w.vel, vel.data(), vel.size()*sizeof(float)));
	CALL(sphx_memcpy_h2d(w.info, info.data(), info.size()*sizeof(uint16_t))); CALL(sphx_memcpy_h2d(w.hash, hash.data(), hash.size()*sizeof(uint32_t)));
	if (w.world > 1) {
		std::vector<uint32_t> map(c->ncells);
		for (uint32_t p = 0; p < c->gs3; ++p)
			std::fill(map.begin() + (size_t)p*c->plane, map.begin() + (size_t)(p + 1)*c->plane, planeType[p] << 30);
		w.devmap = dev_alloc<uint32_t>(c->ncells);
		CALL(sphx_memcpy_h2d(w.devmap, map.data(), map.size()*sizeof(uint32_t)));
	}

	for (uint32_t s = 0; s < c->steps; ++s) step(w);
	CALL(sphx_device_synchronize());

	const uint32_t n = w.n_int;
	std::vector<float> opos(4*(size_t)n), ovel(4*(size_t)n); std::vector<uint16_t> oinfo(4*(size_t)n); std::vector<uint32_t> ohash(n);
	float dt = 0.0f; double t = 0.0;
	CALL(sphx_memcpy_d2h(opos.data(), w.pos, opos.size()*sizeof(float))); CALL(sphx_memcpy_d2h(ovel.data(), w.vel, ovel.size()*sizeof(float)));
	CALL(sphx_memcpy_d2h(oinfo.data(), w.info, oinfo.size()*sizeof(uint16_t))); CALL(sphx_memcpy_d2h(ohash.data(), w.hash, ohash.size()*sizeof(uint32_t)));
	CALL(sphx_memcpy_d2h(&dt, w.d_dt, sizeof(float))); CALL(sphx_memcpy_d2h(&t, w.d_t, sizeof(double)));
	const std::string path = prefix + "." + std::to_string(rank) + ".bin";
	FILE *f = std::fopen(path.c_str(), "wb");
	if (!f) { std::fprintf(stderr, "slab_run: cannot write %s\n", path.c_str()); std::exit(2); }
	std::fwrite(&n, sizeof(n), 1, f); std::fwrite(&dt, sizeof(dt), 1, f); std::fwrite(&t, sizeof(t), 1, f);
	std::fwrite(opos.data(), sizeof(float), opos.size(), f); std::fwrite(ovel.data(), sizeof(float), ovel.size(), f);
	std::fwrite(oinfo.data(), sizeof(uint16_t), oinfo.size(), f); std::fwrite(ohash.data(), sizeof(uint32_t), ohash.size(), f);
	std::fclose(f);
	std::printf("slab_run: rank %d of %d: %u internal particles after %u steps, dt %.6g, t %.6g\n", rank, w.world, n, c->steps, dt, t);
	if (w.halo) CALL(sphx_halo_destroy(w.halo));
	sphx_destroy(w.ctx);
}

int main(int argc, char **argv)
{
	if (argc != 3) { std::fprintf(stderr, "usage: slab_run <case.bin> <out prefix>\n"); return 2; }
	const Case c = read_case(argv[1]);
	sphx_halo_group *group = nullptr;
	if (c.world > 1) CALL(sphx_halo_group_create((int)c.world, &group));
	std::vector<std::thread> threads;
	for (uint32_t r = 0; r < c.world; ++r) threads.emplace_back(worker, &c, (int)r, group, std::string(argv[2]));
	for (auto &t : threads) t.join();
	if (group) CALL(sphx_halo_group_destroy(group));
	return g_fail ? 1 : 0;
}
