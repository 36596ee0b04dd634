cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c15
gpusph_amd/host/halo_check 2 > gpurun_out/c15/halo2.txt 2>&1; echo "rc=$?" >> gpurun_out/c15/halo2.txt
gpusph_amd/host/halo_check 3 > gpurun_out/c15/halo3.txt 2>&1; echo "rc=$?" >> gpurun_out/c15/halo3.txt
timeout 900 python -m pytest tests/test_gpu_halo.py tests/test_gpu_parity.py -q -m gpu -k "halo or repack or worker or rccl" 2>&1 | tail -5 > gpurun_out/c15/pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/c15/smoke.txt 2>&1
