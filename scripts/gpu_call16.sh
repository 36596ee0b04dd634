cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c16
python scripts/diag_tileframe.py > gpurun_out/c16/diag.txt 2>&1
python scripts/time_neibs.py 32e6 > gpurun_out/c16/neibs32.txt 2>&1
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c16/stats -- python bench.py --no-cpu-baseline --steps 20 --warmup 11 > gpurun_out/c16/bench.log 2>&1
cp $(ls gpurun_out/c16/stats/*/*kernel_stats.csv) gpurun_out/c16/kernel_stats.csv; rm -rf gpurun_out/c16/stats
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | grep -v '^E   +\|^E  +' | grep 'passed\|failed\|FAILED\|^>\|^E  ' > gpurun_out/c16/pytest.txt
