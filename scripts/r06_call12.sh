#!/bin/bash
# round 6, GPU call 12: job staging (the window of a tile staged and converted from a table of 64-record jobs): parity, then the
# forces launch against the committed state on the same box
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06_call12
mkdir -p $OUT
export SPHX_LIB=$PWD/gpusph_amd/variants/libsphx_jobs.so
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_options.py -q -m gpu -x -k "not cpp_adapters" > $OUT/pytest.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest.txt
tail -5 $OUT/pytest.txt
for rep in 1 2; do
for tag in base jobs; do
  export SPHX_LIB=$PWD/gpusph_amd/variants/libsphx_$tag.so
  python bench.py --no-cpu-baseline --steps 30 --warmup 11 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$tag', '32M', d['value'], 'M/s', d['ms_per_step'], 'ms/step  forces', d['roofline']['launch_ms'], 'ms')"
  python bench.py --no-cpu-baseline --particles 8e6 --steps 30 --warmup 11 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$tag', '8M', d['value'], 'M/s', d['ms_per_step'], 'ms/step  forces', d['roofline']['launch_ms'], 'ms')"
done
done 2>&1 | tee $OUT/ab.txt
