"""Build profiles/*pmc_traffic*.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py.
usage: make_traffic_json.py <fetch counter_collection.csv> <write counter_collection.csv> <particles> <out.json>"""
import csv, hashlib, json, os, sys, collections

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_sha():
    """hash of the kernel sources the profiled library was built from: bench.py reports the traffic of a profile only
    for the build it was taken on"""
    h = hashlib.sha256()
    for name in ("forces.hip", "neibs.hip", "neibs_build.hip", "euler.hip", "sphx_internal.h"):
        h.update(open(os.path.join(ROOT, "gpusph_amd", "csrc", name), "rb").read())
    return h.hexdigest()


def load(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


if __name__ == "__main__":
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    out = {"particles": int(sys.argv[3]), "kernel_source_sha": kernel_source_sha(),
           "workload": "DamBreak3D %d particles (bench.py default)" % int(sys.argv[3]),
           "units": "bytes per launch",
           "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2) around bench.py; counters are "
                     "in KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced reads; checked on "
                     "euler_kernel: 2*FETCH = 1.79 GB vs 1.91 GB algorithmic read, WRITE = 1.019 GB vs 1.019 GB algorithmic write)",
           "kernels": {}}
    for name in sorted(set(fetch) | set(write)):
        f, w = fetch.get(name, [0.0]), write.get(name, [0.0])
        short = name.split("(")[0].replace("void ", "")
        fa, wa = sum(f) / len(f), sum(w) / len(w)
        if fa + wa < 64:
            continue
        out["kernels"][short] = {"FETCH_SIZE_KiB_avg": fa, "launches_FETCH_SIZE": len(f), "WRITE_SIZE_KiB_avg": wa,
                                 "launches_WRITE_SIZE": len(w), "hbm_bytes_per_launch": int((2 * fa + wa) * 1024)}
    json.dump(out, open(sys.argv[4], "w"), indent=1)
    for k, v in out["kernels"].items():
        print("%-40s %8.3f GB" % (k, v["hbm_bytes_per_launch"] / 1e9))
