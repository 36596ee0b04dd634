set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c5
timeout 900 python -m pytest tests/test_gpu_sa.py tests/test_gpu_keps.py -q -m gpu 2>&1 | tail -30 > gpurun_out/c5/pytest_sa.txt
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c5/sa_stats -- python scripts/time_sa.py 0.0045 StillWaterSA 20 > gpurun_out/c5/time_sa.txt 2>&1
python - <<'PY' > gpurun_out/c5/sa_kernels.txt
import csv,glob
f=glob.glob('gpurun_out/c5/sa_stats/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print("%-60s calls %4s avg %10.1f us"%(r['Name'].replace('void ','')[:60], r['Calls'], float(r['AverageNs'])/1e3))
PY
cp $(ls gpurun_out/c5/sa_stats/*/*kernel_stats.csv) gpurun_out/c5/sa_kernel_stats.csv
rm -rf gpurun_out/c5/sa_stats
