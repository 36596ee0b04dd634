#!/bin/bash
# round 6, GPU call 13: the parity file on the tree with the ring warm-up; phase timers of the committed staging against the job staging
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06_call13
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_options.py -q -m gpu > $OUT/pytest.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest.txt
tail -4 $OUT/pytest.txt
for tag in base_dbg jobs_dbg; do
  echo "== $tag"
  SPHX_LIB=$PWD/gpusph_amd/variants/libsphx_$tag.so python scripts/tile_profile.py 32e6 2>&1 | tail -12
done | tee $OUT/phases.txt
