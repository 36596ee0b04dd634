#!/bin/bash
# round 6, GPU call 20: the boundary-element terms with the wave's lanes packed (sa_forces_wall_packed_kernel,
# sa_density_sum_wall_packed_kernel): the SA suites and the two-slab parity, then the SA mirrors with one wave per particle
# (SPHX_SA_WALL_PACKED=0) and packed
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06_call20
mkdir -p $OUT
timeout 1800 python -m pytest tests/test_gpu_sa.py tests/test_gpu_sa_io.py tests/test_gpu_sa_moving.py tests/test_gpu_openchannel.py tests/test_gpu_parity.py -q -m gpu -x > $OUT/pytest.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest.txt
tail -25 $OUT/pytest.txt
for mode in 0 1; do
for c in SABox SAChannelIO; do
  rm -rf gpurun_out/sa_$c
  steps=20; [ $c = SAChannelIO ] && steps=10
  SPHX_SA_WALL_PACKED=$mode rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sa_$c -- python scripts/time_sa_case_one.py $c 0.008 $steps 2>&1 | grep "ms/step" | sed "s/^/packed=$mode /"
  python - <<PY
import csv,glob
f=glob.glob('gpurun_out/sa_$c/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print("  %-70s calls %5s avg %9.1f us total %8.1f ms %5s%%"%(r['Name'].replace('void ','')[:70], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6, r['Percentage'][:5]))
PY
done
done 2>&1 | tee $OUT/sa_wall_packed_ab.txt
