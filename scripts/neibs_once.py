import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
from gpusph_amd.problem import DamBreak3D
from gpusph_amd.engine import TimestepEngine
n = float(sys.argv[1]) if len(sys.argv) > 1 else 8e6
prob = DamBreak3D(DamBreak3D.deltap_for(n))
eng = TimestepEngine(prob, track_particle_count=False)
eng.build_neibs(); eng.iterations = 1
eng.build_neibs()
torch.cuda.synchronize()
