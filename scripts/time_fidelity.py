"""Step time of the option sets that run through the fidelity kernels (one thread per particle over the u16 list): SPH_GRENIER,
generalized Newtonian rheology, SPH_HA, internal energy, and the same tank through the optimised kernels for comparison.
usage: python scripts/time_fidelity.py [deltap] [steps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gpusph_amd import defs as D
from gpusph_amd.problem import DamBreak3D, Poiseuille
from gpusph_amd.engine import TimestepEngine

dp = float(sys.argv[1]) if len(sys.argv) > 1 else 0.006
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
NEWT = dict(rheologytype=D.NEWTONIAN, turbmodel=D.LAMINAR_FLOW, compvisc=D.DYNAMIC, avgop=D.HARMONIC)
cases = [
    ("SPH_F1 two fluids, optimised kernels", lambda: DamBreak3D(dp, obstacle=False, two_fluids=True, viscosity=NEWT, density_diffusion=D.DENSITY_DIFFUSION_NONE)),
    ("SPH_GRENIER two fluids", lambda: DamBreak3D(dp, obstacle=False, two_fluids=True, formulation=D.SPH_GRENIER, viscosity="DYNAMICVISC",
                                                   density_diffusion=D.DENSITY_DIFFUSION_NONE)),
    ("SPH_HA two fluids", lambda: DamBreak3D(dp, obstacle=False, two_fluids=True, formulation=D.SPH_HA, viscosity=NEWT,
                                             density_diffusion=D.DENSITY_DIFFUSION_NONE)),
    ("internal energy (AccuracyTest options)", lambda: DamBreak3D(dp, obstacle=False, internal_energy=True, density_diffusion=D.DENSITY_DIFFUSION_NONE)),
    ("PAPANASTASIOU Poiseuille", lambda: Poiseuille(int(round(1.0 / (dp * 1.6))), rheology=D.PAPANASTASIOU)),
]
if os.environ.get("SA_CASES"):      # the SA tank (StillWaterSA's options), laminar and with the k-epsilon model
    from gpusph_amd.problem import SABox
    sdp = float(os.environ["SA_CASES"])
    cases = [("SA walls, laminar", lambda: SABox(sdp, jitter=0.05)),
             ("SA walls, k-epsilon", lambda: SABox(sdp, jitter=0.05, viscosity=dict(rheologytype=D.NEWTONIAN, turbmodel=D.KEPSILON)))]
for name, make in cases:
    prob = make()
    eng = TimestepEngine(prob, device="cuda:0")
    eng.run(11)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    eng.run(steps)
    ev[1].record(); torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / steps
    v = eng.vel[:eng.n].cpu().numpy()
    print("%-42s %9d particles  %8.3f ms/step  %8.1f M updates/s  finite %s" % (name, prob.num_particles, ms, prob.num_particles / ms / 1e3,
                                                                           np.isfinite(v).all()))
    del eng
    torch.cuda.empty_cache()
