#!/bin/bash
# round 6, GPU call 32: the final tree -- smoke, the whole GPU suite, the profile set (scripts/profile_round.sh r06f), the default bench line with
# the CPU baseline, the SA mirrors
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06_call32
mkdir -p $OUT
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke rc=$?" >> $OUT/smoke.txt
tail -2 $OUT/smoke.txt
timeout 2700 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.txt
grep -E "passed|failed" $OUT/pytest_gpu.txt | tail -2
scripts/profile_round.sh r06f > $OUT/profile_round.txt 2>&1
tail -8 $OUT/profile_round.txt
for c in SABox SAPaddleBox SAChannelIO SAChannelIOFlap; do
  steps=20; case $c in SAChannelIO*) steps=10;; esac
  timeout 600 python scripts/time_sa_case_one.py $c 0.008 $steps 2>&1 | grep "ms/step"
done | tee $OUT/sa_mirrors.txt
python bench.py > $OUT/bench32M.json 2> $OUT/bench32M.err; cat $OUT/bench32M.json
python scripts/time_neibs.py 32e6 2>&1 | tail -1 | tee $OUT/neibs_32M.txt
