#!/bin/bash
# HBM traffic passes of bench.py (separate --pmc runs, MI355X_MICROARCH.md): FETCH_SIZE, WRITE_SIZE, plus a kernel-trace --stats pass.
# usage: scripts/profile_traffic.sh <tag> <particles> [extra bench args]     outputs gpurun_out/<tag>_{fetch,write,stats}/
set -u
TAG=$1; N=$2; shift 2
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/${TAG}_fetch -- python bench.py --particles $N --steps 10 --warmup 11 --no-cpu-baseline "$@" > gpurun_out/${TAG}_fetch.log 2>&1; echo "fetch rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/${TAG}_write -- python bench.py --particles $N --steps 10 --warmup 11 --no-cpu-baseline "$@" > gpurun_out/${TAG}_write.log 2>&1; echo "write rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats -- python bench.py --particles $N --steps 30 --warmup 11 --no-cpu-baseline "$@" > gpurun_out/${TAG}_stats.log 2>&1; echo "stats rc=$?"
