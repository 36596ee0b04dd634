#!/bin/bash
# A/B of builds of the tiled forces kernel: the bench line at 32 M and 8 M particles for the committed library and for every
# library under gpusph_amd/variants/ (SPHX_LIB selects the build)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for lib in gpusph_amd/libsphx.so gpusph_amd/variants/libsphx_*.so; do
  tag=$(basename $lib .so)
  for n in 32e6 8e6; do
    SPHX_LIB=$PWD/$lib python bench.py --no-cpu-baseline --particles $n --steps 30 --warmup 11 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$tag', '$n', d['value'], 'M/s', d['ms_per_step'], 'ms/step  forces', d['roofline']['launch_ms'], 'ms', d['roofline']['frac'])"
  done
done
