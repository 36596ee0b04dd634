"""cpu_baseline of bench.py for a given OMP_NUM_THREADS (set in the environment before the oracle is loaded)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import bench
t = time.perf_counter()
r = bench.cpu_baseline()
print(os.environ.get("OMP_NUM_THREADS"), r["cores"], r["value"], "%.1f s" % (time.perf_counter() - t))
