export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/q_stats -- python bench.py --no-cpu-baseline --steps 20 --warmup 11 > gpurun_out/q.log 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/q_stats/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print("%-36s calls %4s avg %10.1f us"%(r['Name'].replace('void ','')[:36], r['Calls'], float(r['AverageNs'])/1e3))
PY
tail -1 gpurun_out/q.log | cut -c1-120

