// halo_check -- the slab decomposition's exchange through the C ABI from C++, the way GPUWorker's threads would call it:
// two worker threads, one sphx_ctx each (both on device 0 of a one-GPU box; on a node they would sit on different devices and
// the copies would be peer copies), buffers of the shapes of BUFFER_POS / BUFFER_INFO / BUFFER_HASH.  Every thread fills its
// own rows with a pattern, exchanges its two edge layers with its neighbour (sphx_halo_exchange = importExternalCells /
// transferBursts, src/GPUWorker.cc:396-407,825-948), reduces dt (gdata->dts) and a body total, gathers the layer counts, and
// checks what arrived.  Needs libsphx.so only (no GPUSPH tree).  Exit code 0 = all checks passed.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include "sphx.h"

static int g_fail = 0;
#define CHECK(cond, ...) do { if (!(cond)) { std::fprintf(stderr, "halo_check: " __VA_ARGS__); std::fprintf(stderr, "\n"); __sync_fetch_and_add(&g_fail, 1); } } while (0)
#define CALL(x) do { const int rc_ = (x); if (rc_ != 0) { std::fprintf(stderr, "halo_check: %s -> %d: %s\n", #x, rc_, sphx_last_error()); __sync_fetch_and_add(&g_fail, 1); } } while (0)

struct Layers { uint32_t sendL, nSendL, sendR, nSendR, recvL, nRecvL, recvR, nRecvR; };

static void worker(int rank, int world, sphx_halo_group *group)
{
	sphx_ctx *ctx = nullptr;
	CALL(sphx_create(&ctx, 0));
	sphx_halo *h = nullptr;
	CALL(sphx_halo_create_threads(group, ctx, rank, &h));
	const uint32_t rows = 4096;
	void *pos = nullptr, *info = nullptr, *hash = nullptr; float *d_dt = nullptr, *d_tot = nullptr;
	CALL(sphx_malloc(&pos, rows*16)); CALL(sphx_malloc(&info, rows*8)); CALL(sphx_malloc(&hash, rows*4));
	CALL(sphx_malloc((void**)&d_dt, 4)); CALL(sphx_malloc((void**)&d_tot, 6*4));
	// rank r owns rows [0, 1000 + 100 r); its left edge layer is its first 50 + 10 r rows, its right edge layer its last 70 + 10 r
	const uint32_t nint = 1000u + 100u*(uint32_t)rank;
	Layers L;
	L.sendL = 0; L.nSendL = rank > 0 ? 50u + 10u*(uint32_t)rank : 0u;
	L.nSendR = rank < world - 1 ? 70u + 10u*(uint32_t)rank : 0u; L.sendR = nint - L.nSendR;
	uint64_t mine[2] = { L.nSendL, L.nSendR };
	std::vector<uint64_t> all(2*(size_t)world);
	CALL(sphx_halo_allgather_u64x2(h, mine, all.data(), nullptr));
	L.nRecvL = rank > 0 ? (uint32_t)all[2*(rank - 1) + 1] : 0u;            // my left neighbour's right layer
	L.nRecvR = rank < world - 1 ? (uint32_t)all[2*(rank + 1)] : 0u;        // my right neighbour's left layer
	L.recvL = nint; L.recvR = nint + L.nRecvL;
	std::vector<float> hp(rows*4); std::vector<uint16_t> hi(rows*4); std::vector<uint32_t> hh(rows);
	for (uint32_t i = 0; i < rows; ++i) {
		for (int c = 0; c < 4; ++c) { hp[4*i + c] = (float)(1000*rank) + (float)i + 0.25f*(float)c; hi[4*i + c] = (uint16_t)(7*rank + i + c); }
		hh[i] = 0x10000u*(uint32_t)rank + i;
	}
	CALL(sphx_memcpy_h2d(pos, hp.data(), rows*16)); CALL(sphx_memcpy_h2d(info, hi.data(), rows*8)); CALL(sphx_memcpy_h2d(hash, hh.data(), rows*4));
	void *bufs[3] = { pos, info, hash };
	const uint32_t rowBytes[3] = { 16, 8, 4 };
	for (int rep = 0; rep < 3; ++rep)      // the slots of the group are re-used: several exchanges in a row
		CALL(sphx_halo_exchange(h, 3, bufs, rowBytes,
			rank > 0 ? rank - 1 : -1, L.sendL, L.nSendL, L.recvL, L.nRecvL,
			rank < world - 1 ? rank + 1 : -1, L.sendR, L.nSendR, L.recvR, L.nRecvR, nullptr));
	CALL(sphx_device_synchronize());
	std::vector<float> gp(rows*4); std::vector<uint16_t> gi(rows*4); std::vector<uint32_t> gh(rows);
	CALL(sphx_memcpy_d2h(gp.data(), pos, rows*16)); CALL(sphx_memcpy_d2h(gi.data(), info, rows*8)); CALL(sphx_memcpy_d2h(gh.data(), hash, rows*4));
	// my own rows are untouched; the halo rows hold my neighbours' edge rows
	for (uint32_t i = 0; i < nint; ++i) CHECK(gh[i] == hh[i] && gp[4*i] == hp[4*i], "rank %d: own row %u changed", rank, i);
	if (rank > 0) {
		const int nb = rank - 1; const uint32_t nbInt = 1000u + 100u*(uint32_t)nb, nbStart = nbInt - L.nRecvL;
		for (uint32_t k = 0; k < L.nRecvL; ++k) {
			const uint32_t src = nbStart + k, dst = L.recvL + k;
			CHECK(gh[dst] == 0x10000u*(uint32_t)nb + src, "rank %d: hash of left halo row %u", rank, k);
			CHECK(gp[4*dst + 3] == (float)(1000*nb) + (float)src + 0.75f, "rank %d: pos of left halo row %u", rank, k);
			CHECK(gi[4*dst + 2] == (uint16_t)(7*nb + src + 2), "rank %d: info of left halo row %u", rank, k);
		}
	}
	if (rank < world - 1) {
		const int nb = rank + 1;
		for (uint32_t k = 0; k < L.nRecvR; ++k) {
			const uint32_t src = k, dst = L.recvR + k;
			CHECK(gh[dst] == 0x10000u*(uint32_t)nb + src, "rank %d: hash of right halo row %u", rank, k);
			CHECK(gp[4*dst] == (float)(1000*nb) + (float)src, "rank %d: pos of right halo row %u", rank, k);
		}
	}
	// dt = min over the devices, body totals = sum
	const float dt = 1.0e-4f*(float)(rank + 2);
	float tot[6]; for (int c = 0; c < 6; ++c) tot[c] = (float)(rank + 1)*(float)(c + 1);
	CALL(sphx_memcpy_h2d(d_dt, &dt, 4)); CALL(sphx_memcpy_h2d(d_tot, tot, 24));
	CALL(sphx_halo_allreduce_min_f32(h, d_dt, nullptr));
	CALL(sphx_halo_allreduce_sum_f32(h, d_tot, 6, nullptr));
	float gdt = 0; float gtot[6];
	CALL(sphx_memcpy_d2h(&gdt, d_dt, 4)); CALL(sphx_memcpy_d2h(gtot, d_tot, 24));
	CHECK(gdt == 2.0e-4f, "rank %d: dt %g", rank, gdt);
	for (int c = 0; c < 6; ++c) CHECK(gtot[c] == (float)(world*(world + 1)/2)*(float)(c + 1), "rank %d: total %d = %g", rank, c, gtot[c]);
	CALL(sphx_halo_barrier(h, nullptr));
	CALL(sphx_halo_destroy(h));
	sphx_free(pos); sphx_free(info); sphx_free(hash); sphx_free(d_dt); sphx_free(d_tot);
	sphx_destroy(ctx);
}

int main(int argc, char **argv)
{
	const int world = argc > 1 ? std::atoi(argv[1]) : 2;
	sphx_halo_group *group = nullptr;
	if (sphx_halo_group_create(world, &group) != 0) { std::fprintf(stderr, "halo_check: %s\n", sphx_last_error()); return 2; }
	std::vector<std::thread> threads;
	for (int r = 0; r < world; ++r) threads.emplace_back(worker, r, world, group);
	for (auto &t : threads) t.join();
	sphx_halo_group_destroy(group);
	std::printf("halo_check: %d worker threads, %s\n", world, g_fail ? "FAILED" : "all layers, dt and totals as expected");
	return g_fail ? 1 : 0;
}
