cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c27
python bench.py --no-cpu-baseline > gpurun_out/c27/bench32.json 2> gpurun_out/c27/bench32.err
SPHX_TILE_DEBUG=256 python bench.py --no-cpu-baseline > gpurun_out/c27/bench32_noprio.json 2>/dev/null
SPHX_TILE_DEBUG=64 python bench.py --no-cpu-baseline > gpurun_out/c27/bench32_prio1.json 2>/dev/null
python bench.py --no-cpu-baseline --particles 8e6 > gpurun_out/c27/bench8.json 2>/dev/null
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_options.py -q -m gpu -x 2>&1 | grep -v '^E   +\|^E  +' | grep 'passed\|failed\|FAILED\|^>\|^E  ' > gpurun_out/c27/pytest.txt
