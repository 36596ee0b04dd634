#!/bin/bash
# round 6, GPU call 17: where the step of the open channel with SA walls goes (SAChannelIO mirror, 8.6 M particles)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06_call17
mkdir -p $OUT
for c in SAChannelIO; do
  rm -rf gpurun_out/sa_$c
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sa_$c -- python scripts/time_sa_case_one.py $c 0.008 10 2>&1 | grep "ms/step"
  python - <<PY
import csv,glob
f=glob.glob('gpurun_out/sa_$c/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:22]:
    print("  %-70s calls %5s avg %9.1f us total %8.1f ms %5s%%"%(r['Name'].replace('void ','')[:70], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6, r['Percentage'][:5]))
PY
done 2>&1 | tee $OUT/sa_io_kernel_stats.txt
