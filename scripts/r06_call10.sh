#!/bin/bash
# round 6, tenth GPU call: does the side stream's tile_lists_kernel run beside the next part's build_neibs_kernel at all?  Timelines
# of a rebuild in 4 parts (kernel trace), without and with the side stream at the highest priority; and the parity test
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06_call10
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "list_build_in_parts" > $OUT/pytest.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest.txt
tail -3 $OUT/pytest.txt
for prio in 0 1; do
  if [ $prio = 1 ]; then export SPHX_SIDE_PRIORITY=1; else unset SPHX_SIDE_PRIORITY; fi
  for parts in 1 4; do
    export SPHX_LIST_PARTS=$parts
    echo "prio $prio parts $parts: $(python scripts/time_neibs.py 32e6 2>&1 | tail -1)"
    rm -rf gpurun_out/tr_${prio}_$parts
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr_${prio}_$parts -- python scripts/time_neibs.py 32e6 > /dev/null 2>&1
    python scripts/trace_rebuild.py gpurun_out/tr_${prio}_$parts $((2*parts + 2))
  done
done 2>&1 | tee $OUT/timelines.txt
