cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c14
python scripts/time_neibs.py 32e6 > gpurun_out/c14/neibs32.txt 2>&1
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c14/stats -- python bench.py --no-cpu-baseline --steps 20 --warmup 11 > gpurun_out/c14/bench.log 2>&1
cp $(ls gpurun_out/c14/stats/*/*kernel_stats.csv) gpurun_out/c14/kernel_stats.csv; rm -rf gpurun_out/c14/stats
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "neibs_phase or full_size_against or golden or inactive" 2>&1 | tail -3 > gpurun_out/c14/pytest.txt
