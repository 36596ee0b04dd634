#!/bin/bash
# round 6, GPU call 26: the SPS stress pass with packed pair arithmetic (stress_interact_pk, bit-identical sums): the SPS tests, then the
# two passes of an SPS step at 8 M particles and the StillWater / WaveTank mirrors (before: profiles/r06c_measure_all.txt)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06_call26
mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu -k "sps or SPS or wavetank or stillwater or allpairs or options" > $OUT/pytest.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest.txt
tail -4 $OUT/pytest.txt
bash scripts/sps_roofline.sh 8e6 > $OUT/sps_roofline_8M.txt 2>&1; cat $OUT/sps_roofline_8M.txt
python scripts/time_stillwater.py 4e6 SPSVISC 2>&1 | tail -1 | tee $OUT/stillwater_sps.txt
python scripts/time_wavetank.py 2>&1 | tail -1 | tee $OUT/wavetank.txt
