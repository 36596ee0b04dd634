#!/bin/bash
# round 6, second GPU call: the list build with its distance tests on the matrix cores (neibs_build.hip): the whole -m gpu suite
# (lists bit-exact against the oracle up to 32 M particles), then the neighbour phase timed with and without the prepass
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06_call2
mkdir -p $OUT
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest.txt
tail -5 $OUT/pytest.txt
for mode in mfma general; do
  for n in 32e6 8e6; do
    rm -rf /tmp/q_$mode
    if [ $mode = general ]; then export SPHX_NEIBS_NO_MFMA=1; else unset SPHX_NEIBS_NO_MFMA; fi
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/q_$mode -- python scripts/time_neibs.py $n > $OUT/neibs_${mode}_$n.log 2>&1
    python - "$mode" "$n" <<'PY'
import csv, glob, sys
tag, n = sys.argv[1], sys.argv[2]
f = glob.glob('/tmp/q_%s/**/*kernel_stats.csv' % tag, recursive=True)
if not f:
    print(tag, "no stats"); sys.exit(0)
for r in csv.DictReader(open(f[0])):
    if any(k in r['Name'] for k in ('tile_lists', 'build_neibs', 'build_tiles')):
        print("%-8s %-5s %-44s calls %4s avg %10.1f us" % (tag, n, r['Name'].replace('void ', '')[:44], r['Calls'], float(r['AverageNs'])/1e3))
PY
    grep "rebuild ms" $OUT/neibs_${mode}_$n.log
  done
done | tee $OUT/neibs_ab.txt
unset SPHX_NEIBS_NO_MFMA
python bench.py --no-cpu-baseline --steps 30 --warmup 11 > $OUT/bench32M.json 2> $OUT/bench32M.err
python bench.py --no-cpu-baseline --particles 8e6 --steps 30 --warmup 11 > $OUT/bench8M.json 2>/dev/null
cat $OUT/bench32M.json | cut -c1-400
