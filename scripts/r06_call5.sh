#!/bin/bash
# round 6, fifth GPU call: what the SA run comparisons actually measure (SPHX_TEST_REPORT), to set their allowances from numbers
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06_call5
mkdir -p $OUT
rm -f $OUT/sa_report.txt
SPHX_TEST_REPORT=$PWD/$OUT/sa_report.txt timeout 1500 python -m pytest tests/test_gpu_sa.py tests/test_gpu_sa_io.py tests/test_gpu_sa_moving.py tests/test_gpu_keps.py tests/test_bench_contract.py -q -m gpu > $OUT/pytest.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest.txt
tail -5 $OUT/pytest.txt
cat $OUT/sa_report.txt
