"""time the filter / post-processing / repacking kernels at a given size (HIP events around each call)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
from gpusph_amd import defs as D
from gpusph_amd.problem import DamBreak3D
from gpusph_amd.engine import TimestepEngine

n = float(sys.argv[1]) if len(sys.argv) > 1 else 8e6
prob = DamBreak3D(DamBreak3D.deltap_for(n), obstacle=True)
eng = TimestepEngine(prob, device="cuda:0", track_particle_count=False)
for _ in range(3):
    eng.step()

def timed(name, fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    print("%-28s %8.3f ms" % (name, e0.elapsed_time(e1) / reps))

print("particles", eng.n)
timed("shepard", lambda: eng.apply_filter(D.SHEPARD_FILTER))
timed("mls", lambda: eng.apply_filter(D.MLS_FILTER))
timed("vorticity", lambda: eng.postprocess(D.VORTICITY))
timed("surface detection", lambda: eng.postprocess(D.SURFACE_DETECTION, normals=True))
timed("repack forces", lambda: eng._forces(eng.pos, eng.vel, 1, 0, D.REPACK))
