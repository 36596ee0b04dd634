cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c6
timeout 900 python -m pytest tests/test_gpu_sa.py tests/test_gpu_keps.py -q -m gpu 2>&1 | grep -v '^E   +\|^E  +' > gpurun_out/c6/pytest_sa.txt
SA_CASES=0.0045 python scripts/time_fidelity.py > gpurun_out/c6/time_fidelity_sa.txt 2>&1
