cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c13
python scripts/tile_profile.py 32e6 > gpurun_out/c13/prof.txt 2>&1
python bench.py --no-cpu-baseline > gpurun_out/c13/bench32.json 2> gpurun_out/c13/bench32.err
python bench.py --no-cpu-baseline --particles 8e6 > gpurun_out/c13/bench8.json 2> gpurun_out/c13/bench8.err
