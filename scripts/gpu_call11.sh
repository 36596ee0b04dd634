cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c11
python scripts/diag_tileframe.py > gpurun_out/c11/diag_tileframe.txt 2>&1
python scripts/tile_profile.py 32e6 > gpurun_out/c11/prof.txt 2>&1
python bench.py --no-cpu-baseline > gpurun_out/c11/bench32.json 2> gpurun_out/c11/bench32.err
python bench.py --no-cpu-baseline --particles 8e6 > gpurun_out/c11/bench8.json 2> gpurun_out/c11/bench8.err
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | grep -v '^E   +\|^E  +' | tail -60 > gpurun_out/c11/pytest.txt
