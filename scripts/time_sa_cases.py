"""Step time of the SA_BOUNDARY mirrors with moving bodies / open boundaries next to the tank with walls at rest, same box and spacing:
the numbers DESIGN.md section 0 (row f-2) quotes.
usage: python scripts/time_sa_cases.py [deltap] [steps]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpusph_amd.problem import SABox, SAPaddleBox, SALoadBox, SAChannelIO, SAChannelIOFlap
from gpusph_amd.engine import TimestepEngine

dp = float(sys.argv[1]) if len(sys.argv) > 1 else 0.008
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
box = dict(l=1.6, w=1.6, h=1.0, H=0.8)
cases = [("SABox (walls at rest)", lambda: SABox(dp, options="StillWaterSA", **box)),
         ("SAPaddleBox (moving flap)", lambda: SAPaddleBox(dp, **box)),
         ("SALoadBox (force on a body)", lambda: SALoadBox(dp, **box)),
         ("SAChannelIO (open boundaries)", lambda: SAChannelIO(dp, l=3.2, w=1.6, h=1.0, H=0.8)),
         ("SAChannelIOFlap (open + moving)", lambda: SAChannelIOFlap(dp, l=3.2, w=1.6, h=1.0, H=0.8))]
for name, make in cases:
    try:
        t0 = time.time()
        prob = make()
        eng = TimestepEngine(prob, device="cuda:0")
        eng.run(11)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record(); eng.run(steps); ev[1].record(); torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1])/steps
        print("%-34s %9d particles  %8.3f ms/step  %7.1f M particle-updates/s  (rebuild every %d steps; set-up %.0f s)" % (
            name, eng.n, ms, eng.n/ms/1e3, prob.simparams.buildneibsfreq, time.time() - t0), flush=True)
        del eng, prob
        torch.cuda.empty_cache()
    except Exception as e:      # a case that does not build at this size says so; the others still run
        print("%-34s FAILED: %s" % (name, e), flush=True)
