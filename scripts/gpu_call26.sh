cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash scripts/profile_round.sh r03c > gpurun_out/profile_round_r03c.log 2>&1
python bench.py > gpurun_out/profiles_r03c/r03c_bench32M_with_cpu_baseline.json 2> gpurun_out/profiles_r03c/bench_cpu.err
for dp in 0.008 0.0045; do
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sa_stats_$dp -- python scripts/time_sa.py $dp StillWaterSA 20 > gpurun_out/profiles_r03c/sa_time_$dp.txt 2>&1
cp $(ls gpurun_out/sa_stats_$dp/*/*kernel_stats.csv) gpurun_out/profiles_r03c/r03c_sa_stillwatersa_dp${dp}_kernel_stats.csv; rm -rf gpurun_out/sa_stats_$dp
done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sa_stats_q -- python scripts/time_sa.py 0.008 StillWaterRepackSA 20 > gpurun_out/profiles_r03c/sa_time_quad_0.008.txt 2>&1
cp $(ls gpurun_out/sa_stats_q/*/*kernel_stats.csv) gpurun_out/profiles_r03c/r03c_sa_quadrature_dp0.008_kernel_stats.csv; rm -rf gpurun_out/sa_stats_q
