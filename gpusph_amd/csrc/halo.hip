// halo.hip -- exchange of the slab decomposition behind the C ABI: what GPUWorker::importExternalCells / transferBursts /
// peerAsyncTransfer / networkTransfer do in the reference (src/GPUWorker.cc:396-407, 711-822, 825-948), for the two ways a
// GPUSPH-like host can be laid out on one node:
//   * one THREAD per device in one process (the reference's own model, GPUWorker::simulationThread): the devices' buffers live in
//     one address space, a halo layer is one peer copy, the worker threads meet at a barrier around it as the reference's do
//     (gdata->threadSynchronizer); hipMemcpyPeerAsync rides the xGMI link between two MI355X of a node;
//   * one PROCESS per device (torch.distributed.run, MPI): RCCL send / recv of the two edge layers in one group per exchange,
//     over the direct xGMI link between neighbouring slabs; dt is an all-reduce(min) of one device float.
// Both sit behind the same entry points (sphx_halo_exchange, _allreduce_min, _allreduce_sum, _allgather); the host chooses the
// transport when it creates its sphx_halo.  RCCL is loaded on first use (dlopen), so a host that never creates an RCCL
// transport never maps it.  Ranges are rows of the caller's particle arrays: [start, start + count) of `rowBytes` bytes each,
// the inner-edge segments REORDER leaves contiguous.
#include "sphx_internal.h"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <pthread.h>
#include <cstring>
#include <mutex>

namespace {

struct Rccl {
	void *lib;
	ncclResult_t (*GetUniqueId)(ncclUniqueId*);
	ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
	ncclResult_t (*CommDestroy)(ncclComm_t);
	ncclResult_t (*GroupStart)();
	ncclResult_t (*GroupEnd)();
	ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
	ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
	ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
	ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
	const char *(*GetErrorString)(ncclResult_t);
};

std::mutex g_rccl_mutex;
Rccl g_rccl;

// the process may already hold an RCCL (PyTorch bundles one under the same SONAME): dlopen hands that one back
const Rccl *rccl()
{
	std::lock_guard<std::mutex> lock(g_rccl_mutex);
	if (g_rccl.lib) return &g_rccl;
	void *lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
	if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
	if (!lib) return nullptr;
	Rccl r;
	r.lib = lib;
#define SPHX_SYM(field, name) do { *(void**)(&r.field) = dlsym(lib, name); if (!r.field) return nullptr; } while (0)
	SPHX_SYM(GetUniqueId, "ncclGetUniqueId"); SPHX_SYM(CommInitRank, "ncclCommInitRank"); SPHX_SYM(CommDestroy, "ncclCommDestroy");
	SPHX_SYM(GroupStart, "ncclGroupStart"); SPHX_SYM(GroupEnd, "ncclGroupEnd"); SPHX_SYM(Send, "ncclSend"); SPHX_SYM(Recv, "ncclRecv");
	SPHX_SYM(AllReduce, "ncclAllReduce"); SPHX_SYM(AllGather, "ncclAllGather"); SPHX_SYM(GetErrorString, "ncclGetErrorString");
#undef SPHX_SYM
	g_rccl = r;
	return &g_rccl;
}

#define SPHX_NCCL(R, call) do { ncclResult_t _r = (call); if (_r != ncclSuccess) \
	return sphx_set_error(SPHX_ERR_RUNTIME, std::string(#call) + ": " + (R)->GetErrorString(_r)); } while (0)

enum { MAX_BUFS = 32 };

// what a worker thread publishes for its neighbours to pull from (thread transport)
struct Posted {
	const void *buf[MAX_BUFS];
	uint32_t rowBytes[MAX_BUFS];
	int nbuf;
	uint32_t sendStart[2], sendCount[2];   // [0] towards the left neighbour, [1] towards the right
	int device;
	hipEvent_t ready, done;
	double scalar[8];                      // all-reduce / all-gather payload
	int err;                               // this rank failed inside the collective in flight: its peers return an error too
};

}   // namespace

struct sphx_halo_group {
	int world;
	pthread_barrier_t barrier;
	Posted *posted;      // [world]
	int attached;
	std::mutex mutex;
};

struct sphx_halo {
	sphx_ctx *ctx;
	int rank, world;
	// thread transport
	sphx_halo_group *group;
	// RCCL transport
	ncclComm_t comm;
	uint64_t *gather_dev;     // [2*world] scratch of the all-gather
	int peer_seen[2];         // devices peer access has been asked for (thread transport)
};

// ------------------------------------------------------------------------------------------
// creation
// ------------------------------------------------------------------------------------------
extern "C" int sphx_halo_group_create(int world, sphx_halo_group **out)
{
	SPHX_REQUIRE(out && world >= 1, "sphx_halo_group_create: invalid argument");
	sphx_halo_group *g = new sphx_halo_group;
	g->world = world; g->attached = 0;
	g->posted = new Posted[world];
	std::memset(g->posted, 0, sizeof(Posted)*(size_t)world);
	if (pthread_barrier_init(&g->barrier, nullptr, (unsigned)world) != 0) {
		delete[] g->posted; delete g;
		return sphx_set_error(SPHX_ERR_RUNTIME, "sphx_halo_group_create: pthread_barrier_init failed");
	}
	*out = g;
	return SPHX_OK;
}

extern "C" int sphx_halo_group_destroy(sphx_halo_group *g)
{
	if (!g) return SPHX_OK;
	pthread_barrier_destroy(&g->barrier);
	delete[] g->posted;
	delete g;
	return SPHX_OK;
}

extern "C" int sphx_halo_create_threads(sphx_halo_group *group, sphx_ctx *ctx, int rank, sphx_halo **out)
{
	SPHX_REQUIRE(group && ctx && out, "sphx_halo_create_threads: NULL argument");
	SPHX_REQUIRE(rank >= 0 && rank < group->world, "sphx_halo_create_threads: rank out of range");
	SPHX_HIP(hipSetDevice(ctx->device));
	sphx_halo *h = new sphx_halo;
	h->ctx = ctx; h->rank = rank; h->world = group->world; h->group = group; h->comm = nullptr; h->gather_dev = nullptr; h->peer_seen[0] = h->peer_seen[1] = -1;
	Posted &p = group->posted[rank];
	p.device = ctx->device;
	SPHX_HIP(hipEventCreateWithFlags(&p.ready, hipEventDisableTiming));
	SPHX_HIP(hipEventCreateWithFlags(&p.done, hipEventDisableTiming));
	{ std::lock_guard<std::mutex> lock(group->mutex); group->attached++; }
	*out = h;
	return SPHX_OK;
}

extern "C" int sphx_halo_unique_id(void *id128)
{
	SPHX_REQUIRE(id128 != nullptr, "sphx_halo_unique_id: NULL argument");
	const Rccl *R = rccl();
	if (!R) return sphx_set_error(SPHX_ERR_RUNTIME, "sphx_halo_unique_id: librccl.so not found");
	static_assert(sizeof(ncclUniqueId) == 128, "the id travels as 128 opaque bytes");
	ncclUniqueId id;
	SPHX_NCCL(R, R->GetUniqueId(&id));
	std::memcpy(id128, &id, sizeof(id));
	return SPHX_OK;
}

extern "C" int sphx_halo_create_rccl(sphx_ctx *ctx, const void *id128, int rank, int world, sphx_halo **out)
{
	SPHX_REQUIRE(ctx && id128 && out, "sphx_halo_create_rccl: NULL argument");
	SPHX_REQUIRE(world >= 1 && rank >= 0 && rank < world, "sphx_halo_create_rccl: rank out of range");
	const Rccl *R = rccl();
	if (!R) return sphx_set_error(SPHX_ERR_RUNTIME, "sphx_halo_create_rccl: librccl.so not found");
	SPHX_HIP(hipSetDevice(ctx->device));
	ncclUniqueId id;
	std::memcpy(&id, id128, sizeof(id));
	sphx_halo *h = new sphx_halo;
	h->ctx = ctx; h->rank = rank; h->world = world; h->group = nullptr; h->comm = nullptr; h->gather_dev = nullptr; h->peer_seen[0] = h->peer_seen[1] = -1;
	ncclResult_t r = R->CommInitRank(&h->comm, world, id, rank);
	if (r != ncclSuccess) { delete h; return sphx_set_error(SPHX_ERR_RUNTIME, std::string("ncclCommInitRank: ") + R->GetErrorString(r)); }
	if (hipMalloc((void**)&h->gather_dev, sizeof(uint64_t)*2*(size_t)(world + 1)) != hipSuccess) {
		R->CommDestroy(h->comm); delete h;
		return sphx_set_error(SPHX_ERR_RUNTIME, "sphx_halo_create_rccl: hipMalloc failed");
	}
	*out = h;
	return SPHX_OK;
}

extern "C" int sphx_halo_destroy(sphx_halo *h)
{
	if (!h) return SPHX_OK;
	if (h->group) {
		Posted &p = h->group->posted[h->rank];
		if (p.ready) (void)hipEventDestroy(p.ready);
		if (p.done) (void)hipEventDestroy(p.done);
		p.ready = p.done = nullptr;
	}
	if (h->comm) { const Rccl *R = rccl(); if (R) R->CommDestroy(h->comm); }
	if (h->gather_dev) (void)hipFree(h->gather_dev);
	delete h;
	return SPHX_OK;
}

// ------------------------------------------------------------------------------------------
// the exchange: my two edge layers out, the two halo layers in, for every buffer of the list
// ------------------------------------------------------------------------------------------
extern "C" int sphx_halo_exchange(sphx_halo *h, int nbuf, void *const *bufs, const uint32_t *rowBytes,
	int leftRank, uint32_t sendLeftStart, uint32_t sendLeftCount, uint32_t recvLeftStart, uint32_t recvLeftCount,
	int rightRank, uint32_t sendRightStart, uint32_t sendRightCount, uint32_t recvRightStart, uint32_t recvRightCount,
	void *stream_)
{
	SPHX_REQUIRE(h != nullptr, "sphx_halo_exchange: NULL handle");
	SPHX_REQUIRE(nbuf >= 0 && nbuf <= MAX_BUFS && (nbuf == 0 || (bufs && rowBytes)), "sphx_halo_exchange: invalid buffer list");
	SPHX_REQUIRE(leftRank < h->world && rightRank < h->world, "sphx_halo_exchange: neighbour rank out of range");
	hipStream_t stream = (hipStream_t)stream_;
	const int peer[2] = { leftRank, rightRank };
	const uint32_t sendStart[2] = { sendLeftStart, sendRightStart }, sendCount[2] = { sendLeftCount, sendRightCount };
	const uint32_t recvStart[2] = { recvLeftStart, recvRightStart }, recvCount[2] = { recvLeftCount, recvRightCount };
	for (int b = 0; b < nbuf; ++b) SPHX_REQUIRE(bufs[b] != nullptr && rowBytes[b] > 0, "sphx_halo_exchange: NULL buffer or empty rows");

	if (h->comm) {      // one process per device: RCCL send / recv, all of them in one group
		const Rccl *R = rccl();
		// Two slabs on a ring (a periodic split axis) are each other's neighbour on both sides; sends and receives between one
		// pair of ranks are matched in posting order and my LEFT layer is my peer's RIGHT halo: the receives go right halo first.
		const bool samePeer = peer[0] >= 0 && peer[0] == peer[1];
		SPHX_NCCL(R, R->GroupStart());
		for (int b = 0; b < nbuf; ++b) {
			char *base = (char*)bufs[b];
			for (int s = 0; s < 2; ++s)
				if (peer[s] >= 0 && sendCount[s])
					SPHX_NCCL(R, R->Send(base + (size_t)sendStart[s]*rowBytes[b], (size_t)sendCount[s]*rowBytes[b], ncclUint8, peer[s], h->comm, stream));
			for (int k = 0; k < 2; ++k) {
				const int s = samePeer ? 1 - k : k;
				if (peer[s] >= 0 && recvCount[s])
					SPHX_NCCL(R, R->Recv(base + (size_t)recvStart[s]*rowBytes[b], (size_t)recvCount[s]*rowBytes[b], ncclUint8, peer[s], h->comm, stream));
			}
		}
		SPHX_NCCL(R, R->GroupEnd());
		return SPHX_OK;
	}

	// one thread per device: publish, meet, pull, meet (GPUWorker::transferBursts + the synchroniser's barriers).
	// Failure semantics: once a rank has entered the collective it passes ALL its barriers whatever happens to it (a rank that
	// returned early would leave every other worker thread blocked in pthread_barrier_wait for good); it records the error,
	// raises its `err` flag for the peers, and every rank that sees a flag returns an error after the last barrier.
	sphx_halo_group *g = h->group;
	SPHX_REQUIRE(g != nullptr, "sphx_halo_exchange: handle without a transport");
	int rc = SPHX_OK;
	auto fail = [&](int code, const std::string &msg) { if (rc == SPHX_OK) rc = sphx_set_error(code, msg); };
	Posted &me = g->posted[h->rank];
	me.err = 0;
	if (hipSetDevice(h->ctx->device) != hipSuccess) fail(SPHX_ERR_RUNTIME, "sphx_halo_exchange: hipSetDevice failed");
	me.nbuf = nbuf;
	for (int b = 0; b < nbuf; ++b) { me.buf[b] = bufs[b]; me.rowBytes[b] = rowBytes[b]; }
	for (int s = 0; s < 2; ++s) { me.sendStart[s] = sendStart[s]; me.sendCount[s] = peer[s] >= 0 ? sendCount[s] : 0u; }
	// what I send has been produced once this event has happened
	if (rc == SPHX_OK && hipEventRecord(me.ready, stream) != hipSuccess) fail(SPHX_ERR_RUNTIME, "sphx_halo_exchange: hipEventRecord failed");
	if (rc != SPHX_OK) me.err = 1;
	pthread_barrier_wait(&g->barrier);
	for (int s = 0; s < 2 && rc == SPHX_OK; ++s) {
		if (peer[s] < 0 || !recvCount[s]) continue;
		const Posted &nb = g->posted[peer[s]];
		const int theirs = 1 - s;                     // my left neighbour sends me its RIGHT layer
		if (nb.err) { fail(SPHX_ERR_RUNTIME, "sphx_halo_exchange: a neighbour rank failed before posting its layers"); break; }
		if (nb.nbuf != nbuf || nb.sendCount[theirs] != recvCount[s]) {
			fail(SPHX_ERR_INVALID, "sphx_halo_exchange: neighbour posted a different layer than this rank expects");
			break;
		}
		if (nb.device != h->ctx->device && h->peer_seen[s] != nb.device) {     // direct loads over the link instead of a staged copy
			(void)hipDeviceEnablePeerAccess(nb.device, 0);
			(void)hipGetLastError();                                            // "already enabled" is fine
			h->peer_seen[s] = nb.device;
		}
		if (hipStreamWaitEvent(stream, nb.ready, 0) != hipSuccess) { fail(SPHX_ERR_RUNTIME, "sphx_halo_exchange: hipStreamWaitEvent failed"); break; }
		for (int b = 0; b < nbuf; ++b) {
			if (nb.rowBytes[b] != rowBytes[b]) { fail(SPHX_ERR_INVALID, "sphx_halo_exchange: row size mismatch between neighbours"); break; }
			const size_t bytes = (size_t)recvCount[s]*rowBytes[b];
			char *dst = (char*)bufs[b] + (size_t)recvStart[s]*rowBytes[b];
			const char *src = (const char*)nb.buf[b] + (size_t)nb.sendStart[theirs]*rowBytes[b];
			const hipError_t e = (nb.device == h->ctx->device) ? hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream)
				: hipMemcpyPeerAsync(dst, h->ctx->device, src, nb.device, bytes, stream);
			if (e != hipSuccess) { fail(SPHX_ERR_RUNTIME, std::string("sphx_halo_exchange: peer copy: ") + hipGetErrorString(e)); break; }
		}
	}
	if (hipEventRecord(me.done, stream) != hipSuccess) fail(SPHX_ERR_RUNTIME, "sphx_halo_exchange: hipEventRecord failed");
	if (rc != SPHX_OK) me.err = 1;
	pthread_barrier_wait(&g->barrier);
	// nobody may overwrite its edge rows before its neighbours have pulled them: my stream waits for their copies
	for (int s = 0; s < 2; ++s) {
		if (peer[s] < 0) continue;
		if (g->posted[peer[s]].err) fail(SPHX_ERR_RUNTIME, "sphx_halo_exchange: a neighbour rank failed during the exchange");
		else if (sendCount[s] && hipStreamWaitEvent(stream, g->posted[peer[s]].done, 0) != hipSuccess)
			fail(SPHX_ERR_RUNTIME, "sphx_halo_exchange: hipStreamWaitEvent failed");
	}
	if (rc != SPHX_OK) me.err = 1;
	pthread_barrier_wait(&g->barrier);
	// the verdict is the group's, not the neighbourhood's: a rank two slabs away from the failure would otherwise return
	// SPHX_OK, enter the next collective and wait there for ranks that have returned an error and will not come
	for (int r = 0; r < g->world; ++r)
		if (g->posted[r].err) fail(SPHX_ERR_RUNTIME, "sphx_halo_exchange: a rank of the group failed during the exchange");
	pthread_barrier_wait(&g->barrier);               // the slots may be rewritten from here on
	return rc;
}

// ------------------------------------------------------------------------------------------
// the small collectives of a step: dt (min over the devices, GPUSPH.cc:650-657 through gdata->dts), body forces (sum),
// layer counts after a re-sort (all-gather; host values: the caller sizes its halo with them)
// ------------------------------------------------------------------------------------------
// T = float or double.  Same failure rule as the exchange: every rank passes both barriers; a rank that fails raises its flag
// and every rank returns an error.
template<typename T>
static int thread_allreduce(sphx_halo *h, T *d_vals, uint32_t n, bool take_min, hipStream_t stream)
{
	sphx_halo_group *g = h->group;
	int rc = SPHX_OK;
	auto fail = [&](const char *msg) { if (rc == SPHX_OK) rc = sphx_set_error(SPHX_ERR_RUNTIME, msg); };
	T mine[8], out[8];
	Posted &me = g->posted[h->rank];
	me.err = 0;
	if (n > 8) { n = 0; rc = sphx_set_error(SPHX_ERR_INVALID, "sphx_halo_allreduce: at most 8 values between worker threads"); }
	if (hipSetDevice(h->ctx->device) != hipSuccess) fail("sphx_halo_allreduce: hipSetDevice failed");
	// through the host, like the reference's dt (gdata->dts[], one float per device)
	if (rc == SPHX_OK && (hipMemcpyAsync(mine, d_vals, sizeof(T)*n, hipMemcpyDeviceToHost, stream) != hipSuccess ||
	                      hipStreamSynchronize(stream) != hipSuccess)) fail("sphx_halo_allreduce: device to host copy failed");
	if (rc == SPHX_OK) for (uint32_t k = 0; k < n; ++k) me.scalar[k] = (double)mine[k];
	if (rc != SPHX_OK) me.err = 1;
	pthread_barrier_wait(&g->barrier);
	for (int r = 0; r < g->world; ++r)
		if (g->posted[r].err) fail("sphx_halo_allreduce: a rank of the group failed");
	if (rc == SPHX_OK)
		for (uint32_t k = 0; k < n; ++k) {
			T acc = (T)g->posted[0].scalar[k];
			for (int r = 1; r < g->world; ++r) {
				const T v = (T)g->posted[r].scalar[k];
				acc = take_min ? (v < acc ? v : acc) : acc + v;       // rank order: every thread computes the same sum
			}
			out[k] = acc;
		}
	pthread_barrier_wait(&g->barrier);
	if (rc == SPHX_OK && (hipMemcpyAsync(d_vals, out, sizeof(T)*n, hipMemcpyHostToDevice, stream) != hipSuccess ||
	                      hipStreamSynchronize(stream) != hipSuccess)) fail("sphx_halo_allreduce: host to device copy failed");   // `out` is a stack array
	return rc;
}

extern "C" int sphx_halo_allreduce_min_f32(sphx_halo *h, float *d_value, void *stream)
{
	SPHX_REQUIRE(h && d_value, "sphx_halo_allreduce_min_f32: NULL argument");
	if (h->comm) {
		const Rccl *R = rccl();
		SPHX_NCCL(R, R->AllReduce(d_value, d_value, 1, ncclFloat, ncclMin, h->comm, (hipStream_t)stream));
		return SPHX_OK;
	}
	return thread_allreduce<float>(h, d_value, 1, true, (hipStream_t)stream);
}

extern "C" int sphx_halo_allreduce_sum_f32(sphx_halo *h, float *d_values, uint32_t n, void *stream)
{
	SPHX_REQUIRE(h && d_values, "sphx_halo_allreduce_sum_f32: NULL argument");
	if (!n) return SPHX_OK;
	if (h->comm) {
		const Rccl *R = rccl();
		SPHX_NCCL(R, R->AllReduce(d_values, d_values, n, ncclFloat, ncclSum, h->comm, (hipStream_t)stream));
		return SPHX_OK;
	}
	return thread_allreduce<float>(h, d_values, n, false, (hipStream_t)stream);
}

// the rigid-body force / torque totals are summed in double by the driver: the same wire format here, so that a decomposed run
// gives the totals of the single domain whatever the transport
extern "C" int sphx_halo_allreduce_sum_f64(sphx_halo *h, double *d_values, uint32_t n, void *stream)
{
	SPHX_REQUIRE(h && d_values, "sphx_halo_allreduce_sum_f64: NULL argument");
	if (!n) return SPHX_OK;
	if (h->comm) {
		const Rccl *R = rccl();
		SPHX_NCCL(R, R->AllReduce(d_values, d_values, n, ncclDouble, ncclSum, h->comm, (hipStream_t)stream));
		return SPHX_OK;
	}
	return thread_allreduce<double>(h, d_values, n, false, (hipStream_t)stream);
}

extern "C" int sphx_halo_allgather_u64x2(sphx_halo *h, const uint64_t mine[2], uint64_t *all /* [2*world], host */, void *stream_)
{
	SPHX_REQUIRE(h && mine && all, "sphx_halo_allgather_u64x2: NULL argument");
	hipStream_t stream = (hipStream_t)stream_;
	if (h->comm) {
		const Rccl *R = rccl();
		uint64_t *send = h->gather_dev + 2*(size_t)h->world;
		SPHX_HIP(hipMemcpyAsync(send, mine, 2*sizeof(uint64_t), hipMemcpyHostToDevice, stream));
		SPHX_NCCL(R, R->AllGather(send, h->gather_dev, 2, ncclUint64, h->comm, stream));
		SPHX_HIP(hipMemcpyAsync(all, h->gather_dev, 2*sizeof(uint64_t)*(size_t)h->world, hipMemcpyDeviceToHost, stream));
		SPHX_HIP(hipStreamSynchronize(stream));
		return SPHX_OK;
	}
	sphx_halo_group *g = h->group;
	Posted &me = g->posted[h->rank];
	std::memcpy(&me.scalar[0], &mine[0], sizeof(uint64_t));
	std::memcpy(&me.scalar[1], &mine[1], sizeof(uint64_t));
	pthread_barrier_wait(&g->barrier);
	for (int r = 0; r < g->world; ++r) {
		std::memcpy(&all[2*r], &g->posted[r].scalar[0], sizeof(uint64_t));
		std::memcpy(&all[2*r + 1], &g->posted[r].scalar[1], sizeof(uint64_t));
	}
	pthread_barrier_wait(&g->barrier);
	return SPHX_OK;
}

extern "C" int sphx_halo_barrier(sphx_halo *h, void *stream)
{
	SPHX_REQUIRE(h != nullptr, "sphx_halo_barrier: NULL handle");
	if (h->comm) {      // a one-float all-reduce is the barrier of a stream-ordered transport
		const Rccl *R = rccl();
		float *scratch = (float*)(h->gather_dev);
		SPHX_NCCL(R, R->AllReduce(scratch, scratch, 1, ncclFloat, ncclMin, h->comm, (hipStream_t)stream));
		SPHX_HIP(hipStreamSynchronize((hipStream_t)stream));
		return SPHX_OK;
	}
	SPHX_HIP(hipStreamSynchronize((hipStream_t)stream));
	pthread_barrier_wait(&h->group->barrier);
	return SPHX_OK;
}
