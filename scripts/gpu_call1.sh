set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c1
scripts/ubench/valu_rate > gpurun_out/c1/valu_rate.txt 2>&1
python scripts/diag_tileframe.py > gpurun_out/c1/diag_tileframe.txt 2>&1
timeout 900 python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_parity.py::test_full_size_32M_tiled_equals_generic_and_invariants 2>&1 | tail -40 > gpurun_out/c1/pytest.txt
python bench.py --no-cpu-baseline > gpurun_out/c1/bench32.json 2> gpurun_out/c1/bench32.err
python bench.py --no-cpu-baseline --particles 8e6 > gpurun_out/c1/bench8.json 2> gpurun_out/c1/bench8.err
python scripts/tile_profile.py 32e6 > gpurun_out/c1/tileprof32.txt 2>&1
tail -5 gpurun_out/c1/*.txt; cat gpurun_out/c1/bench32.json gpurun_out/c1/bench8.json
