"""One SA mirror for a kernel trace: python scripts/time_sa_case_one.py <SABox|SAPaddleBox|SAChannelIO|...> [deltap] [steps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpusph_amd.problem as P
from gpusph_amd.engine import TimestepEngine
name = sys.argv[1]
dp = float(sys.argv[2]) if len(sys.argv) > 2 else 0.008
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
kw = dict(l=3.2, w=1.6, h=1.0, H=0.8) if "Channel" in name else dict(l=1.6, w=1.6, h=1.0, H=0.8)
if name == "SABox": kw["options"] = "StillWaterSA"
prob = getattr(P, name)(dp, **kw)
eng = TimestepEngine(prob, device="cuda:0")
eng.run(11); torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record(); eng.run(steps); ev[1].record(); torch.cuda.synchronize()
print("%s: %d particles, %.3f ms/step" % (name, eng.n, ev[0].elapsed_time(ev[1])/steps))
