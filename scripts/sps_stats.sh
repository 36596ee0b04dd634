export TMPDIR=/tmp
cd /root/repo
rm -rf gpurun_out/sps_stats
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sps_stats -- python bench.py --no-cpu-baseline --particles 8e6 --viscosity SPSVISC --steps 20 --warmup 11 > gpurun_out/sps.log 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/sps_stats/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print("%-60s calls %4s avg %10.1f us total %8.1f ms"%(r['Name'].replace('void ','')[:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
tail -1 gpurun_out/sps.log | cut -c1-160
