#!/bin/bash
# round 6, GPU call 18: the passes of a run with open boundaries through the tiles (fluid <- fluid sums) + one-element-per-lane add-ons:
# the open-boundary suites, then the SAChannelIO mirror at 8.6 M particles (kernel stats)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06_call18
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_sa_io.py tests/test_gpu_openchannel.py tests/test_gpu_sa.py tests/test_gpu_sa_moving.py -q -m gpu > $OUT/pytest.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest.txt
tail -6 $OUT/pytest.txt
bash scripts/r06_call17.sh 2>&1 | head -14 | tee $OUT/sa_io_kernel_stats.txt
