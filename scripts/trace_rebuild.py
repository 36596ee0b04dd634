#!/usr/bin/env python3
"""Timeline of the last neighbour-list rebuild of a rocprofv3 --kernel-trace run (scripts/time_neibs.py): start and end of every
launch of the list build and of the tile lists relative to the first, so that one sees what ran beside what."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if any(k in r["Kernel_Name"] for k in ("build_neibs_kernel", "tile_lists_kernel", "build_tiles_kernel", "tile_columns_kernel"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
last = rows[-n:]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    print("%-22s queue %-4s %9.3f .. %9.3f ms" % (r["Kernel_Name"].split("(")[0].replace("void ", "")[:22], r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0)/1e6, (int(r["End_Timestamp"]) - t0)/1e6))
