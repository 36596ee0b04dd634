#!/bin/bash
# round 6, GPU call 14: on the tree with the ring warm-up -- smoke, the whole GPU suite, the profile set (scripts/profile_round.sh r06b),
# every quoted number (scripts/measure_all.sh) and the SA mirrors with moving bodies / open boundaries next to the tank at rest
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06_call14
mkdir -p $OUT
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke rc=$?" >> $OUT/smoke.txt
timeout 1500 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.txt
tail -5 $OUT/pytest_gpu.txt
scripts/profile_round.sh r06b > $OUT/profile_round.txt 2>&1
tail -3 $OUT/profile_round.txt
bash scripts/measure_all.sh > $OUT/measure_all.txt 2>&1
tail -30 $OUT/measure_all.txt
timeout 1200 python scripts/time_sa_cases.py 0.008 20 2>&1 | grep -v "^SABox\|amdgpu.ids" | tee $OUT/sa_cases_4M.txt
