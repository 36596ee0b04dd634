cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c9
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sa.py -q -m gpu -k "n_steps_trajectory or full_size or sa_forces_gamma or 4M" 2>&1 | grep -v '^E   +\|^E  +' > gpurun_out/c9/pytest.txt
