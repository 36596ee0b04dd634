#!/bin/bash
# the option-set lines of scripts/measure_all.sh alone (A/B of builds: SPHX_LIB selects the library)
cd "$(dirname "$0")/.."
for args in "--viscosity SPSVISC" "--two-fluids" ""; do
  python bench.py --no-cpu-baseline --particles 8e6 $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('8M $args', d['value'], 'M/s', d['ms_per_step'], 'ms/step  forces', d['roofline']['launch_ms'], 'ms')"
done
python scripts/time_stillwater.py 4e6 DYNAMICVISC 2>&1 | tail -1
python scripts/time_stillwater.py 4e6 SPSVISC 2>&1 | tail -1
python scripts/time_wavetank.py 2>&1 | tail -1
