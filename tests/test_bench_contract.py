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
