cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c12
for L in 1 2 3 6; do
  if [ $L = 6 ]; then unset SPHX_LIB; else export SPHX_LIB=$GRAFT_REPO_ROOT/gpusph_amd/libsphx_L$L.so; fi
  python scripts/tile_profile.py 32e6 > gpurun_out/c12/prof_L$L.txt 2>&1
  python bench.py --no-cpu-baseline > gpurun_out/c12/bench32_L$L.json 2> gpurun_out/c12/bench32_L$L.err
  python bench.py --no-cpu-baseline --particles 8e6 > gpurun_out/c12/bench8_L$L.json 2> gpurun_out/c12/bench8_L$L.err
done
