#!/bin/bash
# SQ counter passes of bench.py (run on the GPU box through gpurun): two passes of <= 4 counters so that no
# multiplexing is needed.  usage: scripts/profile_sq.sh <tag> <particles> [extra bench args]
# outputs gpurun_out/<tag>_sq{1,2}/ (rocprofv3 SQLite/CSV) + .log
set -u
TAG=$1; N=$2; shift 2
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for pass in 1 2 3; do
  case $pass in
    1) C="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES";;
    2) C="SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES";;
    3) C="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS";;
  esac
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/${TAG}_sq$pass -- \
    python bench.py --particles $N --steps 4 --warmup 11 --no-cpu-baseline "$@" > gpurun_out/${TAG}_sq$pass.log 2>&1
  echo "pass $pass rc=$?"
done
