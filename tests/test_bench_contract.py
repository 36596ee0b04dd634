"""The bench line (bench.py): the JSON contract the driver reads, on a small run; and that there is no CPU stand-in for the
product path -- without a GPU the bench ends with an error and prints no line."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout,
                          cwd=ROOT, env=env)


@pytest.mark.skipif(torch.cuda.is_available(), reason="needs a box WITHOUT a GPU")
def test_no_gpu_no_bench_line():
    r = _run(["--particles", "2e4", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], 300)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.lstrip().startswith("{")]
    assert "GPU" in r.stderr or "HIP" in r.stderr


@pytest.mark.gpu
def test_bench_line_contract():
    r = _run(["--particles", "3e5", "--steps", "12", "--warmup", "11", "--no-cpu-baseline"], 600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.lstrip().startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 12 and d["warmup"] == 11 and d["higher_is_better"] is True
    assert d["unit"] == "M particle-updates/s" and "DamBreak3D" in d["metric"] and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert d["vs_baseline"] is None                  # BASELINE.md holds no published number for this metric
    cfg = d["config"]
    assert "workload" in cfg and "model" not in cfg and cfg["particles"] > 2.5e5 and 30 < cfg["mean_neibs"] < 90
    # value and ms_per_step say the same thing
    assert abs(d["value"] - cfg["particles"] / d["ms_per_step"] / 1e3) <= 2e-3 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and rf["kernel"] == "forces_tile_kernel"
    assert 0 < rf["achieved"] < rf["peak"] and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    # achieved = algorithmic bytes per launch / mean launch time of the dominant kernel (HIP events on its stream)
    assert abs(rf["achieved"] - rf["bytes_per_launch"] / (rf["launch_ms"] * 1e-3) / 1e9) <= 2e-3 * rf["achieved"]
    assert abs(rf["bytes_per_launch"] - (64 + 2 * cfg["mean_neibs"]) * cfg["particles"]) <= 1e-3 * rf["bytes_per_launch"]
    assert 2 * rf["launch_ms"] < d["ms_per_step"]    # two forces passes fit into a step
    assert rf["traffic"] is None                     # PMC traffic is only quoted for the profiled workload and build
    assert "cpu_baseline" not in d                   # --no-cpu-baseline


def test_self_launch_builds_the_drivers_command_line(monkeypatch):
    """`python bench.py --gpus N` typed as it is must become N ranks under torch.distributed.run on 127.0.0.1 with the same
    arguments (the driver's own command form), not an error"""
    sys.path[:0] = [ROOT]
    import importlib
    bench = importlib.import_module("bench")
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7", "--warmup", "3"])
    assert bench.self_launch(4) == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    k = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[k + 1:] == ["--gpus", "4", "--steps", "7", "--warmup", "3"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def _two_rank_line(extra_env, particles="2e5"):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--particles", particles, "--steps", "12",
                        "--warmup", "11", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.lstrip().startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_gpus_2_as_typed_on_one_gpu():
    """`python bench.py --gpus 2` with no launcher around it: two ranks (here on ONE device, over gloo, which is what a one-GPU
    box can run), one JSON line from rank 0 with the exchange accounting"""
    d = _two_rank_line({"SPHX_BENCH_BACKEND": "gloo"})
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["parallelism"] == "slab2"
    ex = d["exchange"]
    assert len(ex["halo_bytes_per_step"]) == 2 and min(ex["halo_bytes_per_step"]) > 0
    assert sum(ex["internal_particles"]) == d["config"]["particles"]
    assert ex["transport"] == "torch.distributed"            # gloo rig: the library's RCCL transport needs one device per rank
    assert d["config"]["env"].get("SPHX_BENCH_BACKEND") == "gloo"
    # what the driver's scaling numbers are read against: bytes moved and time exposed, per rank; and the two ranks together
    # step the particles the single domain steps (the line of N = 1 on the same workload)
    assert len(ex["exposed_exchange_ms_per_step"]) == 2 and min(ex["exposed_exchange_ms_per_step"]) >= 0.0
    d1 = json.loads([l for l in _run(["--particles", "2e5", "--steps", "12", "--warmup", "11", "--no-cpu-baseline"], 900).stdout.splitlines()
                     if l.lstrip().startswith("{")][0])
    assert d1["n_gpus"] == 1
    assert d1["config"]["particles"] == d["config"]["particles"] == sum(ex["internal_particles"])
    assert d1["config"]["workload"] == d["config"]["workload"] and d1["metric"] == d["metric"]


@pytest.mark.gpu
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: RCCL between two devices")
def test_bench_gpus_2_over_rccl_between_two_devices():
    """the first box with two devices proves row e: one rank per GPU, the exchange through the library's own sphx_halo_* RCCL
    entry points, overlap stall reported, the same particle count as the single domain"""
    d = _two_rank_line({}, particles="2e6")
    assert d["n_gpus"] == 2 and d["exchange"]["transport"] == "sphx_halo (RCCL)"
    assert sum(d["exchange"]["internal_particles"]) == d["config"]["particles"]
    assert len(d["exchange"]["exposed_exchange_ms_per_step"]) == 2
    d1 = json.loads([l for l in _run(["--particles", "2e6", "--steps", "12", "--warmup", "11", "--no-cpu-baseline"], 900).stdout.splitlines()
                     if l.lstrip().startswith("{")][0])
    assert d1["config"]["particles"] == d["config"]["particles"]
    assert d["value"] > 0.8 * d1["value"]                      # two devices are not slower than one (2 M particles: latency bound)
