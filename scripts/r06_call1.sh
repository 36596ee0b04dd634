#!/bin/bash
# round 6, first GPU call: (1) the list build after its move to neibs_build.hip + the round's small fixes, (2) A/B of the forces
# parts compiled with "amdgpu-agpr-alloc"="16" (scripts/hipcc_agpr_alloc.py), (3) tile_lists_kernel at one and two workgroups per CU
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06_call1
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sa_moving.py tests/test_gpu_eos_rows.py -x -q -m gpu > $OUT/pytest.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest.txt
tail -3 $OUT/pytest.txt
for lib in gpusph_amd/libsphx.so gpusph_amd/variants/libsphx_a16.so; do
  tag=$(basename $lib .so)
  for n in 32e6 8e6; do
    SPHX_LIB=$PWD/$lib python bench.py --no-cpu-baseline --particles $n --steps 30 --warmup 11 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$tag', '$n', d['value'], 'M/s', d['ms_per_step'], 'ms/step  forces', d['roofline']['launch_ms'], 'ms', d['roofline']['frac'])"
  done
done | tee $OUT/ab_forces.txt
for lib in gpusph_amd/libsphx.so gpusph_amd/variants/libsphx_tlA.so gpusph_amd/variants/libsphx_tlB.so gpusph_amd/variants/libsphx_tlC.so; do
  tag=$(basename $lib .so)
  rm -rf /tmp/q_$tag
  SPHX_LIB=$PWD/$lib rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/q_$tag -- python scripts/time_neibs.py 32e6 > $OUT/neibs_$tag.log 2>&1
  python - "$tag" <<'PY'
import csv, glob, sys
tag = sys.argv[1]
f = glob.glob('/tmp/q_%s/**/*kernel_stats.csv' % tag, recursive=True)
if not f:
    print(tag, "no stats"); sys.exit(0)
for r in csv.DictReader(open(f[0])):
    if any(k in r['Name'] for k in ('tile_lists', 'build_neibs', 'build_tiles', 'sort_rank', 'reorder')):
        print("%-12s %-40s calls %4s avg %10.1f us" % (tag, r['Name'].replace('void ', '')[:40], r['Calls'], float(r['AverageNs'])/1e3))
PY
  grep "rebuild ms" $OUT/neibs_$tag.log
done | tee $OUT/ab_tile_lists.txt
