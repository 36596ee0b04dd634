#!/bin/bash
# round 6, GPU call 11b: (W) no wait in the first three ring steps of a tile, alone and on top of H and H + F
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06_call11
mkdir -p $OUT
for rep in 1 2; do
for tag in base warm hash_warm hash_fsplit_warm; do
  export SPHX_LIB=$PWD/gpusph_amd/variants/libsphx_$tag.so
  python bench.py --no-cpu-baseline --steps 30 --warmup 11 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$tag', '32M', d['value'], 'M/s', d['ms_per_step'], 'ms/step  forces', d['roofline']['launch_ms'], 'ms')"
  python bench.py --no-cpu-baseline --particles 8e6 --steps 30 --warmup 11 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$tag', '8M', d['value'], 'M/s', d['ms_per_step'], 'ms/step  forces', d['roofline']['launch_ms'], 'ms')"
done
done 2>&1 | tee $OUT/ab_b.txt
