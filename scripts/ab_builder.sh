#!/bin/bash
# A/B of builds of the tile-list builder: per-kernel times of four neighbour-list rebuilds at 32 M particles for every
# library under gpusph_amd/variants/ (SPHX_LIB selects the build)
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
for lib in gpusph_amd/variants/libsphx_*.so; do
  tag=$(basename $lib .so)
  SPHX_LIB=$PWD/$lib rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_$tag -- python scripts/time_neibs.py 32e6 $LIN > gpurun_out/ab_$tag.log 2>&1
  python - <<PY
import csv,glob
f=glob.glob('gpurun_out/ab_$tag/*/*kernel_stats.csv')[0]
rows={r['Name'].split('(')[0].replace('void ',''):r for r in csv.DictReader(open(f))}
print("$tag", " ".join("%s %.2f ms" % (k, float(rows[k]['AverageNs'])/1e6) for k in ('tile_lists_kernel','build_neibs_kernel<true, false>','build_tiles_kernel') if k in rows))
PY
done
