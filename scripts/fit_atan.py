"""The arctangent of gpusph_amd/csrc/sa_wall.hip (wall_fast_atan2): fits q in atan(t) = t + t s q(s), s = t^2, on [0, 1] (minimax of
the error of atan(t)/t by reweighted least squares), prints the coefficients and measures the error of a float32 emulation of the
whole atan2 (reciprocal perturbed by 1 ulp + one correction step, fused multiply-adds) against numpy's arctan2 in float64.
usage: python scripts/fit_atan.py [degree]"""
import sys
import numpy as np

deg = int(sys.argv[1]) if len(sys.argv) > 1 else 9
dense = np.linspace(0.0, 1.0, 20001)
t = np.sqrt(dense)
target = np.where(t > 0, np.arctan(t) / np.where(t > 0, t, 1.0), 1.0) - 1.0
A = np.vander(dense, deg + 1, increasing=True)[:, 1:]
w = np.ones_like(dense)
for _ in range(80):
    cw = np.linalg.lstsq(A * w[:, None], target * w, rcond=None)[0]
    err = np.abs(A @ cw - target)
    w = w * (1 + 3 * err / err.max()); w /= w.max()
print("q coefficients, lowest power first:")
for c in cw:
    print("   %r" % float(c))
print("max error of atan(t)/t in float64: %.2e" % err.max())

f32 = np.float32
def fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)
rng = np.random.default_rng(0)
n = 2000000
y = (rng.standard_normal(n) * rng.choice([1e-3, 1, 30], n)).astype(f32)
x = (rng.standard_normal(n) * rng.choice([1e-3, 1, 30], n)).astype(f32)
ax, ay = np.abs(x), np.abs(y)
mx, mn = np.maximum(ax, ay), np.minimum(ax, ay)
r = (1.0 / mx.astype(np.float64)).astype(f32)
r = np.nextafter(r, np.where(rng.random(n) < 0.5, f32(np.inf), f32(-np.inf))).astype(f32)
q = (mn * r).astype(f32)
q = fma(fma(-mx, q, mn), r, q)
s = (q * q).astype(f32)
c32 = cw.astype(f32)
p = np.full_like(s, c32[-1])
for cc in c32[-2::-1]:
    p = fma(p, s, np.full_like(s, cc))
a = fma((q * s).astype(f32), p, q)
a = np.where(ay > ax, (f32(np.pi / 2) - a).astype(f32), a)
a = np.where(x < 0, (f32(np.pi) - a).astype(f32), a)
a = np.copysign(a, y)
ref = np.arctan2(y.astype(np.float64), x.astype(np.float64))
ulp = np.spacing(np.abs(ref).astype(f32))
e = np.abs(a - ref) / ulp
print("atan2 in float32 against the exact value: max %.2f ulp, mean %.2f ulp over %d random arguments" % (e.max(), e.mean(), n))
