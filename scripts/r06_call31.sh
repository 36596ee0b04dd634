#!/bin/bash
# round 6, GPU call 31: tile_lists_kernel with the loads of sections 2 and 5 (list lengths, hash, info of the thread's home particle) asked for at the
# top of a tile's setup (gpusph_amd/variants/libsphx_tlpre.so) against the committed build: four rebuilds at 32 M and 8 M, then the bench lines
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06_call31
mkdir -p $OUT
for lib in gpusph_amd/libsphx.so gpusph_amd/variants/libsphx_tlpipe.so gpusph_amd/variants/libsphx_tlpipe2.so gpusph_amd/libsphx.so gpusph_amd/variants/libsphx_tlpipe.so gpusph_amd/variants/libsphx_tlpipe2.so; do
  tag=$(basename $lib .so)
  for n in 32e6 8e6; do
    rm -rf gpurun_out/ab_$tag
    SPHX_LIB=$PWD/$lib rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_$tag -- python scripts/time_neibs.py $n > gpurun_out/ab_$tag.log 2>&1
    python - <<PY
import csv,glob
f=glob.glob('gpurun_out/ab_$tag/*/*kernel_stats.csv')[0]
rows={r['Name'].split('(')[0].replace('void ',''):r for r in csv.DictReader(open(f))}
print("$tag $n", " ".join("%s %.3f ms" % (k[:24], float(rows[k]['AverageNs'])/1e6) for k in rows if k.startswith(('tile_lists','build_neibs','build_tiles'))), open('gpurun_out/ab_$tag.log').read().strip().split('\n')[-1][-20:])
PY
  done
  SPHX_LIB=$PWD/$lib python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tag bench32M', d['value'], d['ms_per_step'])"
done 2>&1 | tee $OUT/tile_lists_prefetch_ab.txt
SPHX_LIB=$PWD/gpusph_amd/variants/libsphx_tlpipe2.so timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -2 | tee -a $OUT/tile_lists_prefetch_ab.txt
