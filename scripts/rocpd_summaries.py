"""Summaries of rocprofv3 SQLite outputs (the default output format of this ROCm): per-kernel duration statistics of a
--kernel-trace --stats run, per-kernel counter averages of --pmc runs, and the profiles/*pmc_traffic*.json bench.py reads.

usage: rocpd_summaries.py <stats.db> <fetch.db> <write.db> <particles> <out prefix>      e.g. profiles/r01d"""
import csv, json, sqlite3, sys, collections


def kernel_stats(db, out):
    c = sqlite3.connect(db)
    rows = collections.defaultdict(list)
    for name, dur in c.execute("select name, duration from kernels"):
        rows[name].append(dur)
    tot = sum(sum(v) for v in rows.values())
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for name, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([name, len(v), sum(v), "%.1f" % (sum(v) / len(v)), "%.3f" % (100.0 * sum(v) / tot), min(v), max(v)])
    return rows


def pmc(db, counter, out):
    c = sqlite3.connect(db)
    acc = collections.defaultdict(list)
    for name, val in c.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)):
        acc[name].append(val)
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Counter_Name", "Launches", "Average_Value_KiB", "Min", "Max"])
        for name, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([name, counter, len(v), "%.3f" % (sum(v) / len(v)), min(v), max(v)])
    return acc


if __name__ == "__main__":
    stats_db, fetch_db, write_db, particles, prefix = sys.argv[1:6]
    ks = kernel_stats(stats_db, prefix + "_bench32M_kernel_stats.csv")
    fetch = pmc(fetch_db, "FETCH_SIZE", prefix + "_pmc_fetch_summary.csv")
    write = pmc(write_db, "WRITE_SIZE", prefix + "_pmc_write_summary.csv")
    out = {"particles": int(particles), "workload": "DamBreak3D %d particles (bench.py default)" % int(particles),
           "units": "bytes per launch",
           "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2) around bench.py; counters are "
                     "in KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced reads; checked on "
                     "euler_kernel: 2*FETCH vs the algorithmic read, WRITE vs the algorithmic write)",
           "kernels": {}}
    for name in sorted(set(fetch) | set(write)):
        f, w = fetch.get(name, [0.0]), write.get(name, [0.0])
        short = name.split("(")[0].replace("void ", "")
        fa, wa = sum(f) / len(f), sum(w) / len(w)
        if fa + wa < 64:
            continue
        out["kernels"][short] = {"FETCH_SIZE_KiB_avg": fa, "launches_FETCH_SIZE": len(f), "WRITE_SIZE_KiB_avg": wa,
                                 "launches_WRITE_SIZE": len(w), "hbm_bytes_per_launch": int((2 * fa + wa) * 1024)}
    json.dump(out, open(prefix + "_pmc_traffic_32M.json", "w"), indent=1)
    for name, v in sorted(ks.items(), key=lambda kv: -sum(kv[1]))[:14]:
        short = name.split("(")[0].replace("void ", "")
        t = out["kernels"].get(short, {}).get("hbm_bytes_per_launch")
        print("%-52s calls %4d  avg %9.3f us   HBM %s" % (short[:52], len(v), sum(v) / len(v) / 1e3, "%.3f GB" % (t / 1e9) if t else "-"))
