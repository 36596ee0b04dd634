"""The committed profile set belongs to the committed kernel sources: bench.py quotes `roofline.traffic` only from a
profiles/*pmc_traffic*.json whose hash of the kernel sources matches the build, so a round that ends with a stale set would
report `null`.  This holds the newest set against the tree."""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_newest_traffic_profile_is_of_these_sources():
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    from make_traffic_json import kernel_source_sha
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic*.json")))
    assert files, "no PMC traffic profile committed"
    matching = [f for f in files if json.load(open(f)).get("kernel_source_sha") == kernel_source_sha()]
    assert matching, "no committed traffic profile was taken on the current kernel sources: run scripts/profile_round.sh"
    d = json.load(open(matching[-1]))
    tile = [v for k, v in d["kernels"].items() if k.startswith("forces_tile_kernel<")]
    assert tile and tile[0]["hbm_bytes_per_launch"] > 0 and d["particles"] > 3.0e7
    tag = os.path.basename(matching[-1]).split("_")[0]
    for tail in ("bench32M.json", "bench32M_kernel_stats.csv", "sq_32M.csv", "sq_8M.csv", "bench8M.json"):
        assert os.path.exists(os.path.join(ROOT, "profiles", "%s_%s" % (tag, tail))), tail
