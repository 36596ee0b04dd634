#!/bin/bash
# The two tiled kernels of an SPS step against the HBM roofline (DamBreak3D 8 M particles, viscosity<SPSVISC>): kernel times from
# rocprofv3 --kernel-trace --stats, algorithmic bytes per launch as for the headline:
#   forces pass  (forces_tile_kernel<..SPS..>)   own pos + vel + info + hash 44 B, own tau 24 B, list 2 (Nbar + 2) B, forces 16 B  = (88 + 2 Nbar) B per particle
#   stress pass  (forces_tile_kernel<..STRESS..>) own pos + vel + info + hash 44 B, list 2 (Nbar + 2) B, tau written 24 B            = (72 + 2 Nbar) B per particle
# usage (through gpurun): bash scripts/sps_roofline.sh [particles]      output: stdout + gpurun_out/sps_roofline.txt
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
N=${1:-8e6}
rm -rf gpurun_out/sps_stats
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sps_stats -- python bench.py --no-cpu-baseline --particles $N --viscosity SPSVISC --steps 20 --warmup 11 > gpurun_out/sps.log 2>&1
python - <<'PY' | tee gpurun_out/sps_roofline.txt
import csv, glob, json
line = [l for l in open('gpurun_out/sps.log') if l.startswith('{')][-1]
d = json.loads(line)
n, nbar = d["config"]["particles"], d["config"]["mean_neibs"]
print("DamBreak3D %d particles, viscosity<SPSVISC>: %.1f M updates/s, %.3f ms/step, mean neighbours %.1f" % (n, d["value"], d["ms_per_step"], nbar))
f = glob.glob('gpurun_out/sps_stats/*/*kernel_stats.csv')[0]
rows = [r for r in csv.DictReader(open(f)) if "forces_tile_kernel" in r["Name"]]
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:2]:
    ms = float(r["AverageNs"])/1e6
    # the stress instantiation carries the SPHX_TURB_STRESS bit (32) in its second template argument
    turb = int(r["Name"].split("<")[1].split(",")[1])
    stress = bool(turb & 32)
    per = (72.0 if stress else 88.0) + 2.0*nbar
    gbs = n*per/(ms*1e-3)/1e9
    print("%-14s %-52s calls %4s  %.3f ms per launch  %.0f B per particle  %.0f GB/s = %.3f of the HBM roofline" % (
        "stress pass" if stress else "forces pass", r["Name"].replace("void ", "")[:52], r["Calls"], ms, per, gbs, gbs/8000.0))
PY
rm -rf gpurun_out/sps_stats
