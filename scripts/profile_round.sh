#!/bin/bash
# Everything profiles/ holds for one state of the build, taken on the GPU box in one gpurun call:
#   <tag>_bench32M_kernel_stats.csv   rocprofv3 --kernel-trace --stats of bench.py (default workload)
#   <tag>_pmc_traffic_32M.json        HBM bytes per launch per kernel (separate FETCH_SIZE / WRITE_SIZE passes), keyed on the
#                                     hash of the kernel sources
#   <tag>_sq_32M.csv, <tag>_sq_8M.csv SQ counters (VALU/SALU/LDS instructions, LDS bank conflicts, wait / active cycles)
# usage (through gpurun): scripts/profile_round.sh <tag>; then copy gpurun_out/profiles_<tag>/* to profiles/
set -u
TAG=$1
cd "$(dirname "$0")/.."
OUT=gpurun_out/profiles_$TAG
mkdir -p $OUT
scripts/profile_traffic.sh ${TAG}t 32e6
scripts/profile_sq.sh ${TAG}s32 32e6
scripts/profile_sq.sh ${TAG}s8 8e6
python scripts/make_traffic_json.py $(ls gpurun_out/${TAG}t_fetch/*/*counter_collection.csv) $(ls gpurun_out/${TAG}t_write/*/*counter_collection.csv) 31844148 $OUT/${TAG}_pmc_traffic_32M.json > /dev/null
cp $(ls gpurun_out/${TAG}t_stats/*/*kernel_stats.csv) $OUT/${TAG}_bench32M_kernel_stats.csv
python scripts/sq_summary.py $OUT/${TAG}_sq_32M.csv gpurun_out/${TAG}s32_sq1 gpurun_out/${TAG}s32_sq2 gpurun_out/${TAG}s32_sq3 > /dev/null
python scripts/sq_summary.py $OUT/${TAG}_sq_8M.csv gpurun_out/${TAG}s8_sq1 gpurun_out/${TAG}s8_sq2 gpurun_out/${TAG}s8_sq3 > /dev/null
python bench.py --no-cpu-baseline > $OUT/${TAG}_bench32M.json 2> /dev/null
python bench.py --no-cpu-baseline --particles 8e6 > $OUT/${TAG}_bench8M.json 2> /dev/null
ls -la $OUT
