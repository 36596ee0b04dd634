"""M particle-updates/s of the StillWater mirror (DYNAMICVISC, Ferrari diffusion, DYN walls, MLS optional) on one GPU"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
from gpusph_amd import defs as D
from gpusph_amd.problem import StillWater
from gpusph_amd.engine import TimestepEngine

target = float(sys.argv[1]) if len(sys.argv) > 1 else 4e6
visc = sys.argv[2] if len(sys.argv) > 2 else "DYNAMICVISC"
prob = StillWater(StillWater.ppH_for(target), viscosity=visc)
eng = TimestepEngine(prob, device="cuda:0")
for _ in range(20):
    eng.step()
torch.cuda.synchronize()
steps = 40
t0 = time.perf_counter()
for _ in range(steps):
    eng.step()
torch.cuda.synchronize()
el = time.perf_counter() - t0
info = eng.neibs_info()
print("particles %d (fluid %d) viscosity<%s>  %.3f ms/step  %.1f M particle-updates/s  mean neibs %.1f  overflow %d" % (
    eng.n, prob.num_fluid, visc, 1e3 * el / steps, 1e-6 * eng.n * steps / el, info.numInteractions / eng.n, info.hasTooManyNeibs))
