"""M particle-updates/s of the WaveTank mirror (SPSVISC, planes, LJ box, Shepard every 20, moving paddle) on one GPU"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
from gpusph_amd import defs as D
from gpusph_amd.problem import WaveTank
from gpusph_amd.engine import TimestepEngine

dp = float(sys.argv[1]) if len(sys.argv) > 1 else 0.00572
prob = WaveTank(dp, paddle_tstart=0.0)
eng = TimestepEngine(prob, device="cuda:0")
eng.add_filter(D.SHEPARD_FILTER, 20)
for _ in range(10):
    eng.step()
torch.cuda.synchronize()
steps = 40
t0 = time.perf_counter()
for _ in range(steps):
    eng.step()
torch.cuda.synchronize()
el = time.perf_counter() - t0
info = eng.neibs_info()
print("particles %d (fluid %d)  %.3f ms/step  %.1f M particle-updates/s  mean neibs %.1f  overflow %d" % (
    eng.n, prob.num_fluid, 1e3 * el / steps, 1e-6 * eng.n * steps / el, info.numInteractions / eng.n, info.hasTooManyNeibs))
