cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c19
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c19/stats -- python scripts/time_sa.py 0.0045 StillWaterSA 20 > gpurun_out/c19/sa_23M_tiled.txt 2>&1
cp $(ls gpurun_out/c19/stats/*/*kernel_stats.csv) gpurun_out/c19/kernel_stats_23M.csv; rm -rf gpurun_out/c19/stats
