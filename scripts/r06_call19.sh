#!/bin/bash
# round 6, GPU call 19: the density summation of a run with open boundaries through the tiles + one element per lane
# (sa_density_sum_wall_kernel<true>): the open-boundary suites, then the SAChannelIO mirror at 8.6 M particles (kernel stats)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06_call19
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_sa_io.py tests/test_gpu_openchannel.py tests/test_gpu_sa.py tests/test_gpu_sa_moving.py -q -m gpu > $OUT/pytest.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest.txt
tail -30 $OUT/pytest.txt
bash scripts/r06_call17.sh 2>&1 | head -24 | tee $OUT/sa_io_kernel_stats.txt
