set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c4
python scripts/diag_tileframe.py > gpurun_out/c4/diag_tileframe.txt 2>&1
for d in 0 64 128 192; do
SPHX_TILE_DEBUG=$d python scripts/tile_profile.py 32e6 > gpurun_out/c4/prof_$d.txt 2>&1
SPHX_TILE_DEBUG=$d python bench.py --no-cpu-baseline > gpurun_out/c4/bench32_$d.json 2> gpurun_out/c4/bench32_$d.err
done
SPHX_TILE_DEBUG=64 python bench.py --no-cpu-baseline --particles 8e6 > gpurun_out/c4/bench8_64.json 2> gpurun_out/c4/bench8_64.err
python bench.py --no-cpu-baseline --particles 8e6 > gpurun_out/c4/bench8_0.json 2> gpurun_out/c4/bench8_0.err
timeout 900 python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_parity.py::test_full_size_32M_tiled_equals_generic_and_invariants 2>&1 | tail -30 > gpurun_out/c4/pytest.txt
