#!/bin/bash
# round 6, seventh GPU call: the SPS instantiations of the tiled kernel on the hand-managed ring (SPHX_ASMRING_ALL=1, possible with
# the accumulation-register need stated in the bitcode: AGPR_ALLOC=16) against the committed build
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06_call7
mkdir -p $OUT
SPHX_LIB=$PWD/gpusph_amd/variants/libsphx_spsring.so timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_options.py -q -m gpu -k "sps or SPS or stillwater or wavetank or two_fluids or options" > $OUT/pytest.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest.txt
tail -3 $OUT/pytest.txt
for lib in gpusph_amd/libsphx.so gpusph_amd/variants/libsphx_spsring.so; do
  tag=$(basename $lib .so)
  export SPHX_LIB=$PWD/$lib
  python bench.py --no-cpu-baseline --particles 8e6 --viscosity SPSVISC 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$tag', '8M SPSVISC', d['value'], 'M/s', d['ms_per_step'], 'ms/step  forces', d['roofline']['launch_ms'], 'ms')"
  python bench.py --no-cpu-baseline --particles 8e6 --two-fluids 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$tag', '8M two fluids', d['value'], 'M/s', d['ms_per_step'], 'ms/step  forces', d['roofline']['launch_ms'], 'ms')"
  echo "$tag stillwater 4M SPSVISC: $(python scripts/time_stillwater.py 4e6 SPSVISC 2>&1 | grep -i "ms/step" | tail -1)"
  echo "$tag stillwater 4M DYNAMICVISC: $(python scripts/time_stillwater.py 4e6 DYNAMICVISC 2>&1 | grep -i "ms/step" | tail -1)"
  echo "$tag wavetank: $(python scripts/time_wavetank.py 2>&1 | grep -i "ms/step" | tail -1)"
  python bench.py --no-cpu-baseline --steps 30 --warmup 11 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$tag', '32M plain', d['value'], 'M/s', d['ms_per_step'], 'ms/step  forces', d['roofline']['launch_ms'], 'ms')"
done | tee $OUT/sps_ab.txt
