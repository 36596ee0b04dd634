cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c7
timeout 900 python -m pytest tests/test_gpu_halo.py tests/test_gpu_sa.py tests/test_gpu_keps.py -q -m gpu 2>&1 | grep -v '^E   +\|^E  +' | tail -60 > gpurun_out/c7/pytest.txt
python bench.py --no-cpu-baseline > gpurun_out/c7/bench32_xzy.json 2> gpurun_out/c7/bench32_xzy.err
python bench.py --no-cpu-baseline --linearization yzx > gpurun_out/c7/bench32_yzx.json 2> gpurun_out/c7/bench32_yzx.err
python bench.py --no-cpu-baseline --particles 8e6 > gpurun_out/c7/bench8_xzy.json 2> gpurun_out/c7/bench8_xzy.err
python bench.py --no-cpu-baseline --particles 8e6 --linearization yzx > gpurun_out/c7/bench8_yzx.json 2> gpurun_out/c7/bench8_yzx.err
