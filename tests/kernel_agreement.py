"""How closely the two HIP implementations of a forces pass (LDS-tiled, gather) have to agree.

They sum the same pairs in the same order, but the tiled kernel keeps positions in one frame per tile (a pair's r_ij is the
difference of two shifted positions, not a shift applied to x_i alone) and folds the kernel's constant factor into the
neighbour's mass, so the two differ by rounding: the bound is a fraction of the bar both have to meet against the oracle
(2e-5 of the largest component, tests/test_gpu_parity.py)."""
import numpy as np

KERNELS_TOL = 1.0e-5      # of the largest |component| of the reference side


def assert_forces_agree(f_tiled, f_generic, tol=KERNELS_TOL, what="forces"):
    f_tiled = np.asarray(f_tiled, dtype=np.float64); f_generic = np.asarray(f_generic, dtype=np.float64)
    assert f_tiled.shape == f_generic.shape
    assert np.isfinite(f_tiled).all() and np.isfinite(f_generic).all(), what
    if f_tiled.ndim == 2 and f_tiled.shape[1] == 4:
        cols = ((slice(0, 3), "acceleration"), (slice(3, 4), "density rate"))
    else:
        cols = ((Ellipsis, what),)
    for sl, name in cols:
        a, b = f_tiled[:, sl] if sl is not Ellipsis else f_tiled, f_generic[:, sl] if sl is not Ellipsis else f_generic
        scale = np.abs(b).max()
        err = np.abs(a - b).max() if a.size else 0.0
        assert err <= tol * scale + 1e-30, "%s: tiled and generic %s differ by %.3g of the scale %.3g (tolerance %.1g)" % (
            what, name, err / max(scale, 1e-300), scale, tol)
