#!/bin/bash
# round 6, GPU call 25: the final state of the round -- smoke, the whole GPU suite, the profile set (scripts/profile_round.sh r06c), every quoted
# number (scripts/measure_all.sh) and the three SA mirrors
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06_call25
mkdir -p $OUT
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke rc=$?" >> $OUT/smoke.txt
tail -2 $OUT/smoke.txt
timeout 2700 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.txt
tail -5 $OUT/pytest_gpu.txt
scripts/profile_round.sh r06c > $OUT/profile_round.txt 2>&1
tail -12 $OUT/profile_round.txt
cat gpurun_out/profiles_r06c/r06c_bench32M.json
bash scripts/measure_all.sh > $OUT/measure_all.txt 2>&1
tail -30 $OUT/measure_all.txt
for c in SABox SAPaddleBox SAChannelIO SAChannelIOFlap; do
  steps=20; case $c in SAChannelIO*) steps=10;; esac
  timeout 600 python scripts/time_sa_case_one.py $c 0.008 $steps 2>&1 | grep "ms/step"
done | tee $OUT/sa_mirrors.txt
