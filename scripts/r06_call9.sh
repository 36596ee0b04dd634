#!/bin/bash
# round 6, ninth GPU call: the list build in parts (SPHX_LIST_PARTS: tile lists of part k beside the list build of part k + 1):
# the parity test, then the rebuild and the step at 32 M and 8 M for 1 (= the build of the earlier rounds), 3, 4, 6, 8, 12 parts
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06_call9
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "list_build_in_parts or neibs_phase or tiled_and_generic" > $OUT/pytest.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest.txt
tail -3 $OUT/pytest.txt
for parts in 1 3 4 6 8 12; do
  export SPHX_LIST_PARTS=$parts
  echo "parts $parts: $(python scripts/time_neibs.py 32e6 2>&1 | tail -1)  8M: $(python scripts/time_neibs.py 8e6 2>&1 | tail -1 | sed 's/.*rebuild/rebuild/')"
  python bench.py --no-cpu-baseline --steps 30 --warmup 11 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('parts $parts', '32M', d['value'], 'M/s', d['ms_per_step'], 'ms/step  forces', d['roofline']['launch_ms'], 'ms')"
  python bench.py --no-cpu-baseline --particles 8e6 --steps 30 --warmup 11 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('parts $parts', '8M', d['value'], 'M/s', d['ms_per_step'], 'ms/step  forces', d['roofline']['launch_ms'], 'ms')"
done 2>&1 | tee $OUT/list_parts_ab.txt
