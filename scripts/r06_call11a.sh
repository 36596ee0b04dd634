#!/bin/bash
# round 6, GPU call 11a: (H) the window records' cell hashes requested one tile ahead, (F) the second wave of every SIMD finalizes the
# previous tile before it queues its rows: the committed state (base) against H, F and H + F
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06_call11
mkdir -p $OUT
for rep in 1 2; do
for tag in base hash fsplit hash_fsplit; do
  export SPHX_LIB=$PWD/gpusph_amd/variants/libsphx_$tag.so
  python bench.py --no-cpu-baseline --steps 30 --warmup 11 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$tag', '32M', d['value'], 'M/s', d['ms_per_step'], 'ms/step  forces', d['roofline']['launch_ms'], 'ms')"
  python bench.py --no-cpu-baseline --particles 8e6 --steps 30 --warmup 11 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$tag', '8M', d['value'], 'M/s', d['ms_per_step'], 'ms/step  forces', d['roofline']['launch_ms'], 'ms')"
done
done 2>&1 | tee $OUT/ab_a.txt
