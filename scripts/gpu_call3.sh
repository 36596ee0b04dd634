set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c3
python scripts/diag_tileframe.py > gpurun_out/c3/diag_tileframe.txt 2>&1
python scripts/tile_profile.py 32e6 > gpurun_out/c3/prof_base.txt 2>&1
SPHX_TILE_DEBUG=64 python scripts/tile_profile.py 32e6 > gpurun_out/c3/prof_prio.txt 2>&1
python bench.py --no-cpu-baseline > gpurun_out/c3/bench32.json 2> gpurun_out/c3/bench32.err
SPHX_TILE_DEBUG=64 python bench.py --no-cpu-baseline > gpurun_out/c3/bench32_prio.json 2> gpurun_out/c3/bench32_prio.err
timeout 900 python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_parity.py::test_full_size_32M_tiled_equals_generic_and_invariants 2>&1 | tail -30 > gpurun_out/c3/pytest.txt
