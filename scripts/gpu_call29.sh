cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c29
python bench.py --no-cpu-baseline --particles 8e6 --viscosity SPSVISC > gpurun_out/c29/bench8_sps.json 2> gpurun_out/c29/bench8_sps.err
python scripts/time_wavetank.py > gpurun_out/c29/wavetank.txt 2>&1
python scripts/time_stillwater.py > gpurun_out/c29/stillwater.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_options.py -q -m gpu -x 2>&1 | grep -v '^E   +\|^E  +' | grep 'passed\|failed\|FAILED\|^>\|^E  ' > gpurun_out/c29/pytest.txt
