cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c10
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -v '^E   +\|^E  +' | tail -150 > gpurun_out/c10/pytest.txt
