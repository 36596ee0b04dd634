#!/bin/bash
# round 6, eighth GPU call: the whole GPU suite on the tree of the re-entry, then the profile set of the round (scripts/profile_round.sh)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06_call8
mkdir -p $OUT
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke rc=$?" >> $OUT/smoke.txt
timeout 1500 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.txt
tail -5 $OUT/pytest_gpu.txt
scripts/profile_round.sh r06a > $OUT/profile_round.txt 2>&1
tail -12 $OUT/profile_round.txt
cat gpurun_out/profiles_r06a/r06a_bench32M.json
