"""Per-kernel averages of rocprofv3 --pmc CSV passes (counter_collection.csv).
usage: sq_summary.py <out.csv> <dir> [<dir> ...]      dirs as written by scripts/profile_sq.sh"""
import collections, csv, glob, os, sys

out, dirs = sys.argv[1], sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in dirs:
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
counters = sorted({c for k in acc.values() for c in k})
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Kernel", "Launches"] + counters)
    for name, cs in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0]))):
        n = max(len(v) for v in cs.values())
        w.writerow([name, n] + ["%.0f" % (sum(cs[c]) / len(cs[c])) if c in cs else "" for c in counters])
for name, cs in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0])))[:8]:
    print(name[:60], {c: "%.3g" % (sum(v) / len(v)) for c, v in cs.items()})
