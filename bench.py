#!/usr/bin/env python3
"""bench.py -- M particle-updates/s of the WCSPH timestep hot path on MI355X.

A "step" is one full predictor-corrector time step (forces x2, Euler x2, dt reduction, and the
neighbour-list rebuild every 10th step) over the synthetic DamBreak3D box with
all inputs resident in HBM.  Metric = 1e-6 * sum over timed steps of internal particles / wall
seconds = GPUSPH's MIPPS (src/timing.h:136-164).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--particles P]

For N > 1 the driver launches one rank per GPU (torch.distributed.run); the domain is slab-split
(strong scaling: the total particle count is fixed).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402

DEFAULT_PARTICLES = 32_000_000     # BASELINE.json configs[3]: DamBreak3D 32M (fits one MI355X)
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: 8 TB/s spec
FP32_PEAK_TFLOPS = 157.3           # MI355X_MICROARCH.md: fp32 vector (packed) peak


def usable_cpus():
    """CPUs this process may really run on: the affinity mask capped by the cgroup CPU quota (the GPU boxes show 256 CPUs
    with a quota of 16; 128 OpenMP threads there run 2.3x slower than 16)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(int(int(quota) / int(period)), 1))
    except (OSError, ValueError):
        pass
    return max(n, 1)


def cpu_baseline(target_particles=4_000_000, steps=6):
    """Oracle (OpenMP port of the reference algorithm) on a bounded sample of the same workload."""
    # the baseline is the oracle compiled -O3 -march=native -fopenmp.  The committed build of the oracle (the parity
    # checker) is -O2 without -march so that it runs on any box; for the timing the same source is compiled here, on the CPU
    # it is timed on (same results: no fast-math, no contraction)
    import subprocess, tempfile
    flags = "-O2 -fopenmp (prebuilt)"
    odir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle")
    tmp = os.path.join(tempfile.gettempdir(), "libsph_oracle_native_%d.so" % os.getpid())
    try:
        subprocess.run(["gcc", "-O3", "-march=native", "-g0", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fopenmp", "-std=gnu11",
                        "-shared", "-o", tmp, os.path.join(odir, "sph_oracle.c"), "-lm"], check=True, capture_output=True, timeout=300)
        os.environ["SPH_ORACLE_LIB"] = tmp
        flags = "-O3 -march=native -fopenmp"
    except Exception:
        pass
    import oracle_lib as ol
    from gpusph_amd.problem import DamBreak3D
    ol.lib().orc_set_num_threads(usable_cpus())
    dp = DamBreak3D.deltap_for(target_particles)
    prob = DamBreak3D(dp, obstacle=True)
    sim = ol.OracleSim(prob)
    sim.o.reuse_nl = True            # the list buffer is allocated once, like the device buffer it stands for
    sim.step()                       # includes the neighbour build (iteration 0)
    t0 = time.perf_counter()
    done = 0
    for _ in range(steps):
        sim.step()
        done += sim.n
    t1 = time.perf_counter()
    # amortise the rebuild like the GPU run does (1 rebuild per 10 steps)
    t_r0 = time.perf_counter()
    sim.iterations = 10
    sim.build_neibs()
    t_rebuild = time.perf_counter() - t_r0
    per_step = (t1 - t0) / steps + t_rebuild / 10.0
    return {
        "value": round(1e-6 * sim.n / per_step, 4), "unit": "M particle-updates/s",
        "cores": int(ol.lib().orc_num_threads()), "kind": "port",
        "sample": "DamBreak3D %d particles (dp=%.5f), %d steps + 1 rebuild amortised over 10 steps, "
                  "oracle/sph_oracle.c %s" % (sim.n, dp, steps, flags),
    }


def pmc_traffic(n_total, kernel):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes
    (profiles/*pmc_traffic*.json: separate FETCH_SIZE / WRITE_SIZE runs of this same command, FETCH doubled
    per the gfx950 correction).  PMC collection needs rocprofv3 around the process, so it cannot be taken
    live here; the number is only reported when the profiled workload is the one being run."""
    import glob
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "scripts"))
    from make_traffic_json import kernel_source_sha
    sha = kernel_source_sha()
    for path in sorted(glob.glob(os.path.join(here, "profiles", "*pmc_traffic*.json")), reverse=True):
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        if int(d.get("particles", -1)) != int(n_total):
            continue
        if d.get("kernel_source_sha") != sha:      # a profile of another build says nothing about this one
            continue
        for name, k in d.get("kernels", {}).items():
            if name.startswith(kernel + "<") or name == kernel:
                return int(k["hbm_bytes_per_launch"])
    return None


def self_launch(n):
    """one rank per GPU under torch.distributed.run on this node (what the driver's command line does for N > 1), same arguments"""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL between processes needs it on this host driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def sphx_env():
    """the SPHX_* switches in effect: part of the configuration of a measured line"""
    return {k: v for k, v in sorted(os.environ.items()) if k.startswith("SPHX_")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--particles", type=float, default=DEFAULT_PARTICLES)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-obstacle", action="store_true")
    ap.add_argument("--no-tiles", action="store_true", help="A/B: use the generic gather forces kernel")
    ap.add_argument("--two-fluids", action="store_true", help="A/B: a lighter second fluid on top (multi-fluid code path)")
    ap.add_argument("--viscosity", default=None, help="A/B: legacy viscosity selector instead of ARTVISC (KINEMATICVISC, DYNAMICVISC, SPSVISC)")
    ap.add_argument("--linearization", default="xzy", help="cell linearisation; the default makes COORD3 = y, DamBreak3D's own split "
                    "axis (src/problems/DamBreak3D.cu:217-220), for every --gpus so that the points of a scaling curve share one memory layout")
    args = ap.parse_args()

    if args.no_tiles:
        os.environ["SPHX_DISABLE_TILES"] = "1"
    import torch
    from gpusph_amd.problem import DamBreak3D

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` typed as it is: become the launcher of N ranks (one per GPU) and hand their line through
        sys.exit(self_launch(args.gpus))
    if args.gpus != world:
        raise SystemExit("bench.py --gpus %d inside a job of WORLD_SIZE %d: launch it as `python bench.py --gpus N` or under "
                         "torch.distributed.run --nproc-per-node N" % (args.gpus, world))
    if os.environ.get("SPHX_BENCH_BACKEND", "nccl") != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("SPHX_BENCH_BACKEND", "nccl")   # "gloo": test rigs with several ranks on ONE GPU
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend=backend)

    dp = DamBreak3D.deltap_for(args.particles, obstacle=not args.no_obstacle)
    lin = args.linearization
    prob = DamBreak3D(dp, obstacle=not args.no_obstacle, linearization=lin, two_fluids=args.two_fluids,
                      viscosity=args.viscosity, kinematic_visc=1.0e-6)
    n_total = prob.num_particles

    if world > 1:
        from gpusph_amd.multigpu import MultiGpuEngine
        # the exchange goes through the library's own RCCL entry points (include/sphx.h sphx_halo_*: what a GPUSPH host calls
        # in place of GPUWorker::transferBursts) unless SPHX_HALO=torch asks for torch.distributed's; torch.distributed only
        # carries the 128-byte communicator id and the bench's own bookkeeping then
        transport, transport_name = None, "torch.distributed"
        if os.environ.get("SPHX_HALO", "capi") == "capi" and os.environ.get("SPHX_BENCH_BACKEND", "nccl") == "nccl":
            from gpusph_amd import capi
            from gpusph_amd.halo import CapiTransport
            box = [None]
            if rank == 0:
                try:
                    box[0] = CapiTransport.new_unique_id(capi.load())
                except Exception as exc:       # no librccl for the library: every rank takes the torch transport
                    print("bench.py: sphx_halo_unique_id failed (%s); exchange over torch.distributed" % exc, file=sys.stderr)
            dist.broadcast_object_list(box, src=0)
            if box[0] is not None:
                # every rank creates its communicator, then all agree (over torch.distributed) on whether every one succeeded: a
                # rank left alone with another transport than its neighbours would hang the run.  The exchange is the same
                # either way; the line says which transport carried it
                name_box = [transport_name]

                def make_transport(k):
                    from gpusph_amd.halo import TorchTransport
                    t, ok = None, 1
                    try:
                        t = CapiTransport(k, rank, world, unique_id=box[0])
                    except Exception as exc:
                        ok = 0
                        print("bench.py: rank %d: sphx_halo_create_rccl failed (%s)" % (rank, exc), file=sys.stderr)
                    flag = torch.tensor([ok], dtype=torch.int32, device=device)
                    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                    if int(flag.item()) == 1:
                        name_box[0] = "sphx_halo (RCCL)"
                        return t
                    if t is not None:
                        t.close()
                    return TorchTransport(dist, True)

                transport = make_transport
        eng = MultiGpuEngine(prob, device=device, rank=rank, world=world, track_particle_count=True, transport=transport)
        if transport is not None:
            transport_name = name_box[0]
    else:
        from gpusph_amd.engine import TimestepEngine
        eng = TimestepEngine(prob, device=device, track_particle_count=False)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        eng.step()
    if world > 1:                           # exchange accounting over the timed region
        eng.halo_bytes = 0
        eng.exchange_events = []
    info = eng.neibs_info()                 # also checks for neighbour-list overflow
    # HIP events on the launch stream around the dominant kernel of every forces pass, recorded by the library
    import ctypes as C
    from gpusph_amd import capi
    ctx = eng.ctx if hasattr(eng, "ctx") else eng.k.ctx
    lib = eng.lib if hasattr(eng, "lib") else eng.k.lib
    capi.check(lib.sphx_forces_timing(ctx.handle, 1))
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    tot_ms, launches = C.c_double(0.0), C.c_uint32(0)
    capi.check(lib.sphx_forces_timing_read(ctx.handle, C.byref(tot_ms), C.byref(launches)))
    capi.check(lib.sphx_forces_timing(ctx.handle, 0))
    # one figure per forces PASS (a multi-GPU pass is an edge-stripe plus an inner-stripe launch)
    forces_ms = [tot_ms.value / (2 * args.steps)] if launches.value else []
    n_internal = eng.internal_particles() if hasattr(eng, "internal_particles") else eng.n
    n64 = C.c_uint64(0)      # the 32-bit counter of the reference's interface wraps at 33 M particles x 65 neighbours
    capi.check(lib.sphx_neibs_interactions64(ctx.handle, C.byref(n64), None))
    interactions = int(n64.value)
    if dist is not None:
        c = torch.tensor([n_internal, interactions], dtype=torch.float64, device=device)
        dist.all_reduce(c)
        n_sum, interactions_sum = int(c[0].item()), int(c[1].item())
    else:
        n_sum, interactions_sum = n_internal, interactions

    exch = None
    if dist is not None:                    # per rank: halo bytes per step and the compute stream's stall on the exchange
        torch.cuda.synchronize()
        stall_ms = sum(b.elapsed_time(a) for b, a in (eng.exchange_events or []))
        mine = torch.tensor([eng.halo_bytes / args.steps, stall_ms / args.steps, float(n_internal)], dtype=torch.float64, device=device)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        exch = {"halo_bytes_per_step": [int(r[0].item()) for r in allr],
                "exposed_exchange_ms_per_step": [round(float(r[1].item()), 4) for r in allr],
                "internal_particles": [int(r[2].item()) for r in allr],
                "comm_cus_reserved": int(getattr(eng, "comm_cus", 0)), "transport": transport_name}
    if rank == 0:
        updates = n_sum * args.steps
        value = 1e-6 * updates / elapsed
        nbar = interactions / max(n_internal, 1)
        # algorithmic bytes of one forces launch on this rank: (64 + 2*Nbar) B per particle
        # (own pos+vel+info+hash 44, list 2(Nbar+2), forces write 16) -- BASELINE.md section 3
        bytes_per_launch = n_internal * (64.0 + 2.0 * nbar)
        avg_ms = float(np.mean(forces_ms)) if forces_ms else float("nan")
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if forces_ms else float("nan")
        kernel = "forces_kernel" if args.no_tiles else "forces_tile_kernel"
        traffic = None if (args.no_tiles or world > 1) else pmc_traffic(n_total, kernel)
        out = {
            "metric": "M particle-updates/sec, DamBreak3D", "value": round(value, 2),
            "unit": "M particle-updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "DamBreak3D %d particles (dp=%.6f), Wendland, WCSPH + %s, "
                                   "Colagrossi diffusion, DYN boundary, neib rebuild every 10 steps"
                                   % (n_total, dp, "viscosity<%s>" % args.viscosity if args.viscosity else "artificial viscosity"),
                       "particles": n_total, "parallelism": "slab%d" % world if world > 1 else "single", "linearization": lin,
                       "mean_neibs": round(nbar, 2), "env": sphx_env()},
            "roofline": {"bound": "hbm", "kernel": kernel, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "launch_ms": round(avg_ms, 4), "bytes_per_launch": int(bytes_per_launch),
                         # what actually binds the pair loop: fp32 vector issue.  ~60 flop per stored pair
                         # against the packed-fp32 peak of the chip
                         "valu": {"bound": "valu", "achieved": round(interactions * 60.0 / (avg_ms * 1e-3) / 1e12, 2),
                                  "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                  "frac": round(interactions * 60.0 / (avg_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4),
                                  "flop_per_pair": 60}},
        }
        if exch is not None:
            out["exchange"] = exch
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
