cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c25
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | grep -v '^E   +\|^E  +' | grep 'passed\|failed\|FAILED\|^>\|^E  ' > gpurun_out/c25/pytest.txt
python scripts/time_sa.py 0.008 StillWaterSA 20 > gpurun_out/c25/sa_4M_tiled.txt 2>&1
python scripts/time_sa.py 0.0045 StillWaterSA 20 > gpurun_out/c25/sa_23M_tiled.txt 2>&1
