#!/bin/bash
# round 6, GPU call 23: the vector instructions of the PACKED wall kernels (scripts/experiments/sa_wall_packed.patch built as
# gpusph_amd/variants/libsphx_wallpacked.so) next to the committed ones -- why do they lose?
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06_call23
mkdir -p $OUT
export SPHX_LIB=$PWD/gpusph_amd/variants/libsphx_wallpacked.so
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pass$i -- python scripts/time_sa_case_one.py SABox 0.008 4 > $OUT/pass$i.log 2>&1
  echo "pass $i rc=$?"
done
python - <<'PY'
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/r06_call23/pass*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name'].split('(')[0].replace('void ','')
        acc[n][r['Counter_Name']].append(float(r['Counter_Value']))
with open('gpurun_out/r06_call23/summary.txt','w') as out:
    for n in acc:
        if 'wall' in n:
            line=n+': '+', '.join('%s=%.4g'%(c,sum(v)/len(v)) for c,v in sorted(acc[n].items()))
            print(line); out.write(line+'\n')
PY
rm -rf $OUT/pass*/*/*kernel_trace.csv
