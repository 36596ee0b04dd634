"""The slab decomposition's exchange behind the C ABI (include/sphx.h sphx_halo_*, gpusph_amd/csrc/halo.hip) on the GPU.

* thread transport (one worker thread per context, the reference's GPUWorker model): two slabs on the one device of this box,
  driven by two Python threads through halo.CapiTransport -- peer copies, barriers, dt through the host -- bit-identical to
  the single-domain run (gather kernels: every particle sees the same arithmetic whatever the decomposition);
* RCCL transport: a one-rank communicator on the box's device (what one GPU can run: RCCL refuses two ranks on one device):
  library load, ncclCommInitRank, all-reduce, all-gather and a grouped send / recv of a layer to the rank itself."""
import threading

import numpy as np
import pytest

from gpusph_amd import defs as D
from gpusph_amd.problem import DamBreak3D

pytestmark = pytest.mark.gpu


def _np(t, dtype=None):
    a = t.cpu().numpy()
    return a.view(dtype) if dtype is not None else a


@pytest.mark.timeout(600)
@pytest.mark.parametrize("case", [dict(), dict(viscosity="SPSVISC", kinematic_visc=1.0e-6)])
def test_two_worker_threads_exchange_through_the_c_abi(monkeypatch, case):
    import torch
    from gpusph_amd import capi
    from gpusph_amd.engine import TimestepEngine
    from gpusph_amd.halo import CapiTransport
    from gpusph_amd.multigpu import MultiGpuEngine
    monkeypatch.setenv("SPHX_DISABLE_TILES", "1")
    kw = dict(deltap=0.03, obstacle=True, jitter=0.05, linearization="xzy", **case)
    steps = 12
    lib = capi.load()
    group = CapiTransport.new_group(lib, 2)
    out, errors = [None, None], []

    def worker(rank):
        try:
            torch.cuda.set_device(0)
            eng = MultiGpuEngine(DamBreak3D(**kw), "cuda:0", rank, 2,
                                 transport=lambda k: CapiTransport(k, rank, 2, group=group))
            for _ in range(steps):
                eng.step()
            torch.cuda.synchronize()
            out[rank] = (eng.download_internal(), eng.current_dt(), eng.halo_bytes, eng.n_local)
            eng.transport.close()
        except BaseException as e:      # the other thread would wait at a barrier for ever: pytest's timeout ends the test
            errors.append((rank, repr(e)))
            raise

    threads = [threading.Thread(target=worker, args=(r,), daemon=True) for r in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=500)
    assert not errors, errors
    assert all(o is not None for o in out), "a worker did not finish"
    lib.sphx_halo_group_destroy(group)

    ref = TimestepEngine(DamBreak3D(**kw), device="cuda:0")
    for _ in range(steps):
        ref.step()
    n = ref.n
    parts = [o[0] for o in out]
    ids = np.concatenate([p["info"][:, 2].astype(np.uint32) | (p["info"][:, 3].astype(np.uint32) << 16) for p in parts])
    order = np.argsort(ids)
    rinfo = _np(ref.info, np.uint16)[:n]
    rid = rinfo[:, 2].astype(np.uint32) | (rinfo[:, 3].astype(np.uint32) << 16)
    ro = np.argsort(rid)
    assert np.array_equal(ids[order], rid[ro])                       # every particle owned by exactly one slab
    for k, t in (("pos", ref.pos), ("vel", ref.vel)):
        got = np.concatenate([p[k] for p in parts])[order]
        assert np.array_equal(got.view(np.uint32), _np(t)[:n][ro].view(np.uint32)), k
    assert all(o[1] == ref.current_dt() for o in out)
    assert all(o[2] > 0 and o[3] > len(o[0]["pos"]) for o in out)   # bytes did move, every slab holds a halo


def test_rccl_transport_on_one_rank():
    import ctypes as C
    import torch
    from gpusph_amd import capi
    from gpusph_amd.halo import CapiTransport
    from gpusph_amd.kernels import HipKernels
    prob = DamBreak3D(0.05, obstacle=False)
    k = HipKernels(prob, prob.num_particles + 64, torch.device("cuda:0"))
    uid = CapiTransport.new_unique_id(k.lib)
    assert len(uid) == 128 and any(uid)
    tr = CapiTransport(k, 0, 1, unique_id=uid)
    dt = torch.tensor([3.5e-4], dtype=torch.float32, device="cuda:0")
    tr.allreduce_min(dt)
    tot = torch.arange(6, dtype=torch.float32, device="cuda:0")
    tr.allreduce_sum(tot)
    counts = tr.allgather_pair(123456789012, 7, torch.device("cuda:0"))
    # a layer sent to the rank itself: rows [10, 30) land in rows [100, 120) of every buffer of the list
    pos = torch.rand((256, 4), dtype=torch.float32, device="cuda:0")
    info = torch.randint(0, 30000, (256, 4), dtype=torch.int16, device="cuda:0")
    want_pos, want_info = pos[10:30].clone(), info[10:30].clone()
    moved = tr.exchange([pos, info], 0, None, (10, 30), (100, 120), (0, 0), (0, 0))
    torch.cuda.synchronize()
    assert float(dt.item()) == np.float32(3.5e-4) and torch.equal(tot.cpu(), torch.arange(6, dtype=torch.float32))
    assert counts == [(123456789012, 7)]
    assert torch.equal(pos[100:120], want_pos) and torch.equal(info[100:120], want_info)
    assert moved == 2 * 20 * (16 + 8)
    tr.close()


def test_cpp_worker_threads_exchange_through_the_c_abi():
    """gpusph_amd/host/halo_check: C++ worker threads (one context each, as GPUWorker's) exchange edge layers of buffers shaped
    like BUFFER_POS / INFO / HASH, reduce dt and a body total and gather layer counts through sphx_halo_*; three slabs, so that
    the middle one has both neighbours"""
    import os, subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpusph_amd", "host", "halo_check")
    assert os.path.exists(exe), "gpusph_amd/host/halo_check is not built (make -C gpusph_amd/host halo_check)"
    for world in (2, 3):
        r = subprocess.run([exe, str(world)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "as expected" in r.stdout


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world,lin", [(2, "xzy"), (3, "yzx"), (2, "ring"), (3, "ring")])
def test_cpp_workers_run_a_decomposed_dam_break(tmp_path, monkeypatch, world, lin):
    """gpusph_amd/host/slab_run: the decomposed run driven from C++ through the C ABI alone (one worker thread and one context
    per slab; neighbour phase with the device map, segments, halo import, stripes of the forces, UPDATE_EXTERNAL of the forces,
    Euler on every row, dt = minimum over the slabs), on the dam break of the headline benchmark (DYN walls, artificial
    viscosity, Colagrossi diffusion) over a neighbour-list rebuild.  With the gather kernels the particles it leaves are
    bit-identical to the single-domain run of the Python driver."""
    import ctypes as C
    import os, struct, subprocess
    import torch
    from gpusph_amd.engine import TimestepEngine
    from gpusph_amd.kernels import HipKernels
    from gpusph_amd.multigpu import SlabPartition
    monkeypatch.setenv("SPHX_DISABLE_TILES", "1")
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpusph_amd", "host", "slab_run")
    assert os.path.exists(exe), "gpusph_amd/host/slab_run is not built (make -C gpusph_amd/host slab_run)"
    steps = 12
    if lin == "ring":      # a box periodic along the split axis with a stream through the periodic face: the slabs form a ring
        from gpusph_amd.problem import PeriodicBox
        kw = dict(deltap=0.05, n=(12, 12, 40), linearization="xyz", jitter=0.2, velocity=(0.1, 0.0, 1.5))
        make = lambda: PeriodicBox(**kw)
    else:
        kw = dict(deltap=0.03, obstacle=False, jitter=0.05, linearization=lin)
        make = lambda: DamBreak3D(**kw)
    prob = make()
    part = SlabPartition(prob, world)
    arrs = prob.copy_to_array()
    n = len(arrs["hash"])
    alloc = int(max(part.local_mask(r, arrs["hash"]).sum() for r in range(world)) * 1.3) + 4096
    k = HipKernels(prob, alloc, torch.device("cuda:0"))      # the uploaded constants and the host-side numbers of the Python driver
    params = bytes(k.params)
    sp = prob.simparams
    case = tmp_path / "case.bin"
    with open(case, "wb") as f:
        f.write(struct.pack("<10I", 0x31424C53, world, steps, n, prob.grid_cells, alloc, part.plane, part.gs3, int(sp.neiblistsize), len(params)))
        f.write(params)
        f.write(struct.pack("<4fII", float(np.float32(sp.dt)), k.sspeed_cfl, k.max_kinvisc, k.sq_nl_radius, int(sp.buildneibsfreq), int(part.ring)))
        f.write(np.asarray(part.lo, dtype=np.uint32).tobytes()); f.write(np.asarray(part.hi, dtype=np.uint32).tobytes())
        f.write(np.ascontiguousarray(arrs["pos"], dtype=np.float32).tobytes())
        f.write(np.ascontiguousarray(arrs["vel"], dtype=np.float32).tobytes())
        f.write(np.ascontiguousarray(arrs["info"]).view(np.uint16).tobytes())
        f.write(np.ascontiguousarray(arrs["hash"]).view(np.uint32).tobytes())
    del k
    env = dict(os.environ, SPHX_DISABLE_TILES="1")
    r = subprocess.run([exe, str(case), str(tmp_path / "out")], capture_output=True, text=True, timeout=500, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    parts = []
    for rank in range(world):
        raw = open(tmp_path / ("out.%d.bin" % rank), "rb").read()
        ni, dt, t = struct.unpack_from("<Ifd", raw, 0)
        off = 16
        pos = np.frombuffer(raw, np.float32, 4 * ni, off).reshape(ni, 4); off += 16 * ni
        vel = np.frombuffer(raw, np.float32, 4 * ni, off).reshape(ni, 4); off += 16 * ni
        info = np.frombuffer(raw, np.uint16, 4 * ni, off).reshape(ni, 4); off += 8 * ni
        parts.append(dict(pos=pos, vel=vel, info=info, dt=dt, t=t, n=ni))
    assert all(p["n"] > 0 for p in parts)
    ref = TimestepEngine(make(), device="cuda:0")
    for _ in range(steps):
        ref.step()
    nr = ref.n
    assert sum(p["n"] for p in parts) == nr
    ids = np.concatenate([p["info"][:, 2].astype(np.uint32) | (p["info"][:, 3].astype(np.uint32) << 16) for p in parts])
    order = np.argsort(ids)
    rinfo = _np(ref.info, np.uint16)[:nr]
    rid = rinfo[:, 2].astype(np.uint32) | (rinfo[:, 3].astype(np.uint32) << 16)
    ro = np.argsort(rid)
    assert np.array_equal(ids[order], rid[ro])                       # every particle owned by exactly one slab
    for key, tns in (("pos", ref.pos), ("vel", ref.vel)):
        got = np.concatenate([p[key] for p in parts])[order]
        assert np.array_equal(got.view(np.uint32), _np(tns)[:nr][ro].view(np.uint32)), key
    assert all(p["dt"] == np.float32(ref.current_dt()) for p in parts)
    assert all(p["t"] == ref.time() for p in parts)


@pytest.mark.timeout(900)
def test_two_slabs_of_8M_particles_on_one_gpu():
    """BASELINE configs[3]'s decomposition at a size that matters: DamBreak3D with 8 M particles cut into two slabs on COORD3,
    both on the one device of this box, two worker threads exchanging their edge layers through sphx_halo_* (peer copies),
    edge stripe first and the inner stripe overlapped, the TILED kernels (the ones the bench runs).  Against the single
    domain after 4 steps incl. the neighbour-list build: every particle owned by exactly one slab, same cells, positions to
    1e-6 of a cell per step, velocities to 1e-4 of the largest (the slabs tile differently: agreement to rounding)."""
    import torch
    from gpusph_amd import capi
    from gpusph_amd.engine import TimestepEngine
    from gpusph_amd.halo import CapiTransport
    from gpusph_amd.multigpu import MultiGpuEngine
    dp = DamBreak3D.deltap_for(8e6)
    kw = dict(deltap=dp, obstacle=True, linearization="xzy")
    steps = 4
    lib = capi.load()
    group = CapiTransport.new_group(lib, 2)
    out, errors = [None, None], []

    def worker(rank):
        try:
            torch.cuda.set_device(0)
            eng = MultiGpuEngine(DamBreak3D(**kw), "cuda:0", rank, 2, track_particle_count=False,
                                 transport=lambda k: CapiTransport(k, rank, 2, group=group))
            for _ in range(steps):
                eng.step()
            torch.cuda.synchronize()
            usable = int(eng.k.lib.sphx_dbg_tiles_usable(eng.k.ctx.handle))
            out[rank] = (eng.download_internal(), eng.current_dt(), eng.halo_bytes, eng.n_local, usable)
            eng.transport.close()
            del eng
        except BaseException as e:
            errors.append((rank, repr(e)))
            raise

    threads = [threading.Thread(target=worker, args=(r,), daemon=True) for r in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=800)
    assert not errors, errors
    assert all(o is not None for o in out), "a worker did not finish"
    lib.sphx_halo_group_destroy(group)
    assert all(o[4] == 1 for o in out), "the slabs did not run on the tiled kernels"

    prob = DamBreak3D(**kw)
    ref = TimestepEngine(prob, device="cuda:0", track_particle_count=False)
    for _ in range(steps):
        ref.step()
    n = ref.n
    assert n > 7.5e6
    parts = [o[0] for o in out]
    ids = np.concatenate([p["info"][:, 2].astype(np.uint32) | (p["info"][:, 3].astype(np.uint32) << 16) for p in parts])
    order = np.argsort(ids)
    rinfo = _np(ref.info, np.uint16)[:n]
    rid = rinfo[:, 2].astype(np.uint32) | (rinfo[:, 3].astype(np.uint32) << 16)
    ro = np.argsort(rid)
    assert np.array_equal(ids[order], rid[ro])                       # every particle owned by exactly one slab
    cs = float(min(prob.m_cellsize))
    got = np.concatenate([p["pos"] for p in parts])[order]; want = _np(ref.pos)[:n][ro]
    assert np.array_equal(got[:, 3].view(np.uint32), want[:, 3].view(np.uint32))
    hg = np.concatenate([p["hash"] for p in parts])[order]
    same_cell = (hg & 0x3FFFFFFF) == (_np(ref.hash, np.uint32)[:n][ro] & 0x3FFFFFFF)
    assert same_cell.mean() > 0.9999
    assert np.abs(got[same_cell, :3] - want[same_cell, :3]).max() <= steps * 1e-6 * cs
    got = np.concatenate([p["vel"] for p in parts])[order]; want = _np(ref.vel)[:n][ro]
    assert np.abs(got[:, :3] - want[:, :3]).max() <= 1e-4 * max(np.abs(want[:, :3]).max(), 1e-3)
    assert np.abs(got[:, 3] - want[:, 3]).max() <= 3e-6
    assert all(abs(o[1] - ref.current_dt()) <= 1e-4 * ref.current_dt() for o in out)
    assert all(o[2] > 0 and o[3] > len(o[0]["pos"]) for o in out)   # bytes did move, every slab holds a halo
    # the slabs are halves: the edge layers are two planes of ~200 x 200 cells
    assert abs(len(parts[0]["pos"]) - len(parts[1]["pos"])) < 0.2 * n
