#!/bin/bash
# round 6, fourth GPU call: the whole -m gpu suite on the tree with the round's additions (force on SA bodies, density summation with
# open boundaries and moving bodies, tiled sums of the moving-bodies density summation, the list build on the matrix cores as an
# opt-in held bit-exact, the 32 M oracle case)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06_call4
mkdir -p $OUT
timeout 2400 python -m pytest tests -q -m gpu > $OUT/pytest.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest.txt
tail -15 $OUT/pytest.txt
