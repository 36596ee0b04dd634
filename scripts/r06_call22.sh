#!/bin/bash
# round 6, GPU call 22: what the wall kernels of the SA tank wait for -- SQ / TA / TCP / TCC counters of one SABox run (4.3 M)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06_call22
mkdir -p $OUT
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ|TA|TCP|TCC|TD)_[A-Za-z0-9_]+" | sort -u > $OUT/counters_available.txt
wc -l $OUT/counters_available.txt
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAVES SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM" "TA_BUSY_avr TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_TRANS SQ_THREAD_CYCLES_VALU SQ_LEVEL_WAVES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pass$i -- python scripts/time_sa_case_one.py SABox 0.008 4 > $OUT/pass$i.log 2>&1
  echo "pass $i ($C) rc=$?"; tail -2 $OUT/pass$i.log
done
python - <<'PY'
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/r06_call22/pass*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name'].split('(')[0].replace('void ','')
        acc[n][r['Counter_Name']].append(float(r['Counter_Value']))
with open('gpurun_out/r06_call22/summary.txt','w') as out:
    for n in ['sa_density_sum_wall_kernel<false>','sa_density_sum_wall_kernel','sa_forces_wall_kernel','forces_tile_kernel<3, 72, 0, false>','sa_segment_bc_kernel<3, false>']:
        if n in acc:
            line=n+': '+', '.join('%s=%.4g'%(c,sum(v)/len(v)) for c,v in sorted(acc[n].items()))
            print(line); out.write(line+'\n')
PY
rm -rf $OUT/pass*/*/*kernel_trace.csv
