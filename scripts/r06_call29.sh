#!/bin/bash
# round 6, GPU call 29: the two boundary-condition passes with one thread per ROW (boundary element / vertex particle) instead of one per
# particle: the SA suites and the two-slab parity, then the SA mirrors (before: profiles/r06d_sa_mirrors.txt)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06_call29
mkdir -p $OUT
timeout 1800 python -m pytest tests/test_gpu_sa.py tests/test_gpu_sa_io.py tests/test_gpu_sa_moving.py tests/test_gpu_openchannel.py tests/test_gpu_parity.py -q -m gpu -k "sa or SA or open or channel" > $OUT/pytest.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest.txt
tail -5 $OUT/pytest.txt
for c in SABox SAChannelIO; do
  rm -rf gpurun_out/sa_$c
  steps=20; [ $c = SAChannelIO ] && steps=10
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sa_$c -- python scripts/time_sa_case_one.py $c 0.008 $steps 2>&1 | grep "ms/step"
  python - <<PY
import csv,glob
f=glob.glob('gpurun_out/sa_$c/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f))):
    if 'bc_kernel' in r['Name'] or 'type_list' in r['Name']:
        print("  %-70s calls %5s avg %9.1f us total %8.1f ms %5s%%"%(r['Name'].replace('void ','')[:70], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6, r['Percentage'][:5]))
PY
done 2>&1 | tee $OUT/sa_bc_rows.txt
