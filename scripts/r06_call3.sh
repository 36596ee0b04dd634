#!/bin/bash
# round 6, third GPU call: where the time of the list build with the prepass on the matrix cores goes (variants without the
# emission / without the tiles), after the row-batched loads; lists of the real build checked first
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06_call3
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sa.py -x -q -m gpu -k "neibs_phase or periodic or inactive or dambreak_8M or dambreak_1M or feels_the_fluid" > $OUT/pytest.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest.txt
tail -3 $OUT/pytest.txt
for lib in gpusph_amd/libsphx.so gpusph_amd/variants/libsphx_skipemit.so gpusph_amd/variants/libsphx_skiptiles.so; do
  tag=$(basename $lib .so)
  rm -rf /tmp/q_$tag
  SPHX_DISABLE_TILES=1 SPHX_LIB=$PWD/$lib rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/q_$tag -- python scripts/time_neibs.py 32e6 > $OUT/neibs_$tag.log 2>&1
  python - "$tag" <<'PY'
import csv, glob, sys
tag = sys.argv[1]
f = glob.glob('/tmp/q_%s/**/*kernel_stats.csv' % tag, recursive=True)
if not f:
    print(tag, "no stats"); sys.exit(0)
for r in csv.DictReader(open(f[0])):
    if 'build_neibs' in r['Name']:
        print("%-20s %-44s calls %4s avg %10.1f us" % (tag, r['Name'].replace('void ', '')[:44], r['Calls'], float(r['AverageNs'])/1e3))
PY
done | tee $OUT/neibs_variants.txt
