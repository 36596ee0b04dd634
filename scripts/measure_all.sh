#!/bin/bash
# Every number DESIGN.md and README.md quote, in one place: run on an MI355X box (through gpurun: `gpurun -- 'bash scripts/measure_all.sh'`),
# results under gpurun_out/measure/.  About 8 GPU-minutes.  The profile set (profiles/<tag>_*) is scripts/profile_round.sh <tag>.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/measure
mkdir -p $OUT
export TMPDIR=/tmp
# headline: DamBreak3D 31.8 M particles (bench default, with the CPU baseline), 8 M, 1 M, 128 M
python bench.py > $OUT/bench32M.json 2> $OUT/bench32M.err
python bench.py --no-cpu-baseline --particles 8e6 > $OUT/bench8M.json 2> /dev/null
python bench.py --no-cpu-baseline --particles 1e6 --steps 200 --warmup 21 > $OUT/bench1M.json 2> /dev/null
python bench.py --no-cpu-baseline --particles 128e6 --steps 12 --warmup 11 > $OUT/bench128M.json 2> /dev/null
# option sets on the same build
python bench.py --no-cpu-baseline --particles 8e6 --viscosity SPSVISC > $OUT/bench8M_spsvisc.json 2> /dev/null
python bench.py --no-cpu-baseline --particles 8e6 --two-fluids > $OUT/bench8M_two_fluids.json 2> /dev/null
python scripts/time_stillwater.py 4e6 DYNAMICVISC > $OUT/stillwater_4M_dynamicvisc.txt 2>&1      # configs[2], DYN walls
python scripts/time_stillwater.py 4e6 SPSVISC > $OUT/stillwater_4M_spsvisc.txt 2>&1
python scripts/time_wavetank.py > $OUT/wavetank_5M.txt 2>&1                                       # configs[4]'s option set
# configs[2] with SA walls: density summation + dynamic gamma + Brezzi (StillWaterSA), and continuity + gamma by quadrature
python scripts/time_sa.py 0.008 StillWaterSA 20 > $OUT/sa_4M_density_sum.txt 2>&1
python scripts/time_sa.py 0.008 StillWaterRepackSA 20 > $OUT/sa_4M_quadrature.txt 2>&1
python scripts/time_sa.py 0.0045 StillWaterSA 20 > $OUT/sa_23M_density_sum.txt 2>&1
SPHX_DISABLE_TILES=1 python scripts/time_sa.py 0.008 StillWaterSA 20 > $OUT/sa_4M_density_sum_list_walkers.txt 2>&1
# both tiled kernels of an SPS step against the HBM roofline
bash scripts/sps_roofline.sh 8e6 > $OUT/sps_roofline_8M.txt 2>&1
# the neighbour phase alone
python scripts/time_neibs.py 32e6 > $OUT/neibs_32M.txt 2>&1
grep -h "ms/step\|rebuild ms\|of the HBM roofline" $OUT/*.txt
for f in $OUT/bench*.json; do python - "$f" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("%-40s %8.1f M updates/s  %8.3f ms/step  forces %.3f ms = %.3f of the HBM roofline" % (
    sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], d["roofline"]["launch_ms"], d["roofline"]["frac"]))
PY
done
