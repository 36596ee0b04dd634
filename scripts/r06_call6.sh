#!/bin/bash
# round 6, sixth GPU call: the SA suites with the tightened allowances, the open channel with a moving flap on the device
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06_call6
mkdir -p $OUT
rm -f $OUT/sa_report.txt
SPHX_TEST_REPORT=$PWD/$OUT/sa_report.txt timeout 1500 python -m pytest tests/test_gpu_sa.py tests/test_gpu_sa_io.py tests/test_gpu_sa_moving.py tests/test_gpu_keps.py tests/test_gpu_openchannel.py -q -m gpu > $OUT/pytest.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest.txt
tail -8 $OUT/pytest.txt
grep "flap" $OUT/sa_report.txt
