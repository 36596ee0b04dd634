cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash scripts/profile_round.sh r03d > gpurun_out/profile_round_r03d.log 2>&1
python bench.py > gpurun_out/profiles_r03d/r03d_bench32M_with_cpu_baseline.json 2> gpurun_out/profiles_r03d/bench_cpu.err
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | grep -v '^E   +\|^E  +' | grep 'passed\|failed\|FAILED\|^>\|^E  ' > gpurun_out/profiles_r03d/pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/profiles_r03d/smoke.txt 2>&1
