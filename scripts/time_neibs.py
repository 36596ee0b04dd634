import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
from gpusph_amd.problem import DamBreak3D
from gpusph_amd.engine import TimestepEngine
n = float(sys.argv[1]) if len(sys.argv) > 1 else 8e6
lin = sys.argv[2] if len(sys.argv) > 2 else "xzy"      # the bench's linearisation
prob = DamBreak3D(DamBreak3D.deltap_for(n), obstacle=True, linearization=lin)
eng = TimestepEngine(prob, track_particle_count=False)
eng.build_neibs(); eng.iterations = 1
torch.cuda.synchronize()
for rep in range(3):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); eng.build_neibs(); e1.record(); torch.cuda.synchronize()
print("SPHX_TILE_DEBUG=%s SPHX_DISABLE_TILES=%s rebuild ms: %.3f" % (os.environ.get("SPHX_TILE_DEBUG"), os.environ.get("SPHX_DISABLE_TILES"), e0.elapsed_time(e1)))
