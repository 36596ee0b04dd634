set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c2
scripts/ubench/valu_rate > gpurun_out/c2/valu_rate.txt 2>&1
python scripts/tile_profile.py 32e6 > gpurun_out/c2/prof_base.txt 2>&1
SPHX_TILE_DEBUG=32 python scripts/tile_profile.py 32e6 > gpurun_out/c2/prof_nolist.txt 2>&1
SPHX_TILE_DEBUG=1 python scripts/tile_profile.py 32e6 > gpurun_out/c2/prof_nopairs.txt 2>&1
python bench.py --no-cpu-baseline > gpurun_out/c2/bench32.json 2> gpurun_out/c2/bench32.err
cat gpurun_out/c2/prof_base.txt gpurun_out/c2/prof_nolist.txt gpurun_out/c2/prof_nopairs.txt
