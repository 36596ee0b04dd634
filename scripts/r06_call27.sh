#!/bin/bash
# round 6, GPU call 27: the density summation of a run with open boundaries AND moving bodies through the tiles + one element per lane
# (sa_density_sum_wall_moving_kernel<true>): the SA suites, then the SAChannelIOFlap mirror at 8.6 M particles (before: 61.2 ms/step)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06_call27
mkdir -p $OUT
timeout 1800 python -m pytest tests/test_gpu_sa_io.py tests/test_gpu_openchannel.py tests/test_gpu_sa.py tests/test_gpu_sa_moving.py -q -m gpu > $OUT/pytest.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest.txt
tail -12 $OUT/pytest.txt
for c in SAChannelIOFlap; do
  rm -rf gpurun_out/sa_$c
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sa_$c -- python scripts/time_sa_case_one.py $c 0.008 10 2>&1 | grep "ms/step"
  python - <<PY
import csv,glob
f=glob.glob('gpurun_out/sa_$c/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print("  %-70s calls %5s avg %9.1f us total %8.1f ms %5s%%"%(r['Name'].replace('void ','')[:70], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6, r['Percentage'][:5]))
PY
done 2>&1 | tee $OUT/sa_io_flap_kernel_stats.txt
