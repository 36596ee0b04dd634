cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c21
timeout 900 python -m pytest tests/test_gpu_sa.py -q -m gpu -x 2>&1 | grep -v '^E   +\|^E  +' | grep 'passed\|failed\|FAILED\|^>\|^E  ' > gpurun_out/c21/pytest_sa.txt
python scripts/time_sa.py 0.0125 StillWaterSA 20 > gpurun_out/c21/sa_1M_tiled.txt 2>&1
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c21/stats1 -- python scripts/time_sa.py 0.0125 StillWaterRepackSA 20 > gpurun_out/c21/sa_1M_quad_tiled.txt 2>&1
cp $(ls gpurun_out/c21/stats1/*/*kernel_stats.csv) gpurun_out/c21/kernel_stats_1M_quad.csv; rm -rf gpurun_out/c21/stats1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c21/stats -- python scripts/time_sa.py 0.0045 StillWaterSA 20 > gpurun_out/c21/sa_23M_tiled.txt 2>&1
cp $(ls gpurun_out/c21/stats/*/*kernel_stats.csv) gpurun_out/c21/kernel_stats_23M.csv; rm -rf gpurun_out/c21/stats
