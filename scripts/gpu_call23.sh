cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c23
timeout 900 python -m pytest tests/test_gpu_sa.py tests/test_gpu_keps.py -q -m gpu -x 2>&1 | grep -v '^E   +\|^E  +' | grep 'passed\|failed\|FAILED\|^>\|^E  ' > gpurun_out/c23/pytest_sa.txt
python scripts/time_sa.py 0.008 StillWaterSA 20 > gpurun_out/c23/sa_4M_tiled.txt 2>&1
python scripts/time_sa.py 0.008 StillWaterRepackSA 20 > gpurun_out/c23/sa_4M_quad_tiled.txt 2>&1
python scripts/time_sa.py 0.0125 StillWaterSA 20 > gpurun_out/c23/sa_1M_tiled.txt 2>&1
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c23/stats -- python scripts/time_sa.py 0.0045 StillWaterSA 20 > gpurun_out/c23/sa_23M_tiled.txt 2>&1
cp $(ls gpurun_out/c23/stats/*/*kernel_stats.csv) gpurun_out/c23/kernel_stats_23M.csv; rm -rf gpurun_out/c23/stats
