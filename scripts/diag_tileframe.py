"""How far the tiled forces kernel (positions in one frame per tile, mass * fcoeff in the window) is from the generic gather
kernel and from the oracle on the same state: the numbers behind tests/kernel_agreement.py.  GPU box."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
from gpusph_amd.problem import DamBreak3D
from gpusph_amd.engine import TimestepEngine
import oracle_lib as ol

def run(case, name):
    prob = DamBreak3D(**case)
    sim = ol.OracleSim(prob); sim.build_neibs()
    n = sim.n
    rng = np.random.default_rng(11)
    vel = sim.vel.copy()
    vel[:n, :3] += rng.uniform(-0.3, 0.3, size=(n, 3)).astype(np.float32)
    vel[:n, 3] += rng.uniform(0, 2e-3, size=n).astype(np.float32)
    cof = 1 if prob.simparams.numforcesbodies else 0
    f_ref, _, _, _, _ = sim.o.forces(sim.pos, vel, sim.info, sim.hash, sim.cs, sim.nl, n, compute_object_forces=cof, rb_count=prob.num_obstacle)
    out = {}
    for mode in ("0", "1"):
        os.environ["SPHX_DISABLE_TILES"] = mode
        eng = TimestepEngine(prob, clobber_neibslist=True)
        eng.build_neibs()
        eng.vel[:n] = torch.from_numpy(vel[:n]).to(eng.device)
        eng._forces(eng.pos, eng.vel, 1, 0)
        out[mode] = (eng.forces[:n].cpu().numpy().astype(np.float64), float(eng.d_dt_next.item()))
    os.environ["SPHX_DISABLE_TILES"] = "0"
    ft, fg = out["0"][0], out["1"][0]
    sx, sw = np.abs(f_ref[:n, :3]).max(), np.abs(f_ref[:n, 3]).max()
    print("%-28s n=%7d  tiled-generic: xyz %.2e w %.2e | tiled-oracle: xyz %.2e w %.2e | generic-oracle: xyz %.2e w %.2e | dt rel %.1e" % (
        name, n, np.abs(ft[:, :3] - fg[:, :3]).max()/sx, np.abs(ft[:, 3] - fg[:, 3]).max()/sw,
        np.abs(ft[:, :3] - f_ref[:n, :3]).max()/sx, np.abs(ft[:, 3] - f_ref[:n, 3]).max()/sw,
        np.abs(fg[:, :3] - f_ref[:n, :3]).max()/sx, np.abs(fg[:, 3] - f_ref[:n, 3]).max()/sw,
        abs(out["0"][1] - out["1"][1])/out["1"][1]), flush=True)

run(dict(deltap=0.04, obstacle=True), "lattice 0.04")
run(dict(deltap=0.03, obstacle=False, jitter=0.1), "jitter 0.03")
run(dict(deltap=0.025, obstacle=True, jitter=0.05, linearization="xzy"), "xzy 0.025")
run(dict(deltap=0.02, obstacle=True, jitter=0.05, hydrostatic=True), "hydrostatic 0.02")
