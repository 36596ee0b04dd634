#!/bin/bash
# The first GPU call of round 5, in the order DESIGN.md 9 item 3 asks for: the open-boundary passes that were written at the end of
# round 4 and have run over their own source on the CPU only (tests/hostemu) meet the device.
#   1. the guarded parity tests of tests/test_gpu_sa_io.py (SPHX_TEST_SA_IO_BC=1: segment / vertex conditions, density summation,
#      forces, Brezzi diffusion, water depth, then SAChannelIO through the engine's driver against the CPU run), each test on its own
#      so that one failure does not hide the others;
#   1b. the OpenChannel mirror on the device (tests/test_gpu_openchannel.py, SPHX_TEST_OPENCHANNEL=1);
#   2. the whole GPU suite as the driver runs it (-x), to see that nothing else moved;
#   3. a bench line.
# before: delete the gpusph_amd/variants/ line from .gpurunignore (and run scripts/build_variants.sh if the directory is empty)
# usage: /usr/local/graft/bin/gpurun --timeout 1800 -- 'bash scripts/round5_first_gpu_call.sh'
# output: gpurun_out/round5_first/*.log.  When 1. is green: drop the skipif marks of those tests, then the refusal in
# sa_io_bc_check (gpusph_amd/csrc/sa_io.hip) and in sa_check (gpusph_amd/csrc/sa_bounds.hip), then the "WRITTEN, NOT YET RUN" notes.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/round5_first
mkdir -p $OUT
export SPHX_TEST_SA_IO_BC=1
for t in test_open_boundary_kernels_are_bit_exact test_boundary_condition_passes_with_open_boundaries \
         test_density_summation_and_forces_with_open_boundaries test_brezzi_diffusion_and_water_depth_with_open_boundaries \
         test_open_channel_on_the_device_follows_the_cpu_run; do
	timeout 300 python -m pytest tests/test_gpu_sa_io.py -q -m gpu -k $t > $OUT/io_$t.log 2>&1
	echo "$t: exit $?" | tee -a $OUT/summary.txt
done
unset SPHX_TEST_SA_IO_BC
SPHX_TEST_OPENCHANNEL=1 timeout 300 python -m pytest tests/test_gpu_openchannel.py -q -m gpu > $OUT/openchannel.log 2>&1
echo "openchannel mirror: exit $?" | tee -a $OUT/summary.txt
timeout 900 python -m pytest tests -x -q -m gpu > $OUT/gpu_suite.log 2>&1
echo "gpu suite: exit $?" | tee -a $OUT/summary.txt
tail -3 $OUT/gpu_suite.log | tee -a $OUT/summary.txt
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json | tee -a $OUT/summary.txt
#   4. the prepared experiments against the committed build (scripts/build_variants.sh puts them under gpusph_amd/variants/ first:
#      the list ring with a running base, + one list descriptor per walk for the SPS passes), bench lines at 32 M and 8 M each
if ls gpusph_amd/variants/libsphx_*.so > /dev/null 2>&1; then
	timeout 600 bash scripts/ab_forces.sh > $OUT/ab_forces.txt 2>&1
	cat $OUT/ab_forces.txt | tee -a $OUT/summary.txt
fi

