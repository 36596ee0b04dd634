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
        cols = ((slice(0, 3), "acceleration", False), (slice(3, 4), "density rate", True))
    else:
        cols = ((Ellipsis, what, False),)
    for sl, name, switched in cols:
        a, b = f_tiled[:, sl] if sl is not Ellipsis else f_tiled, f_generic[:, sl] if sl is not Ellipsis else f_generic
        scale = np.abs(b).max()
        err = np.abs(a - b)
        worst = err.max() if a.size else 0.0
        if switched:
            # the Colagrossi term of a pair is switched by |P_i - P_j| >= |rho_i g.r_ij|: g.r_ij of the tile's frame and of the
            # cell's frame differ in the last bits, so a pair sitting on the threshold may be decided differently by the two
            # kernels (as it is between the GPU and the oracle, test_full_size_against_the_oracle): such a particle differs by
            # that one pair's term.  A handful per million, each far below the field's scale
            beyond = int((err > tol * scale + 1e-30).sum())
            assert beyond <= max(2, int(3e-5 * len(a))) and worst <= 1e-3 * scale, "%s: %d particles beyond %.1g of the scale %.3g, worst %.3g" % (
                name, beyond, tol, scale, worst / max(scale, 1e-300))
            continue
        assert worst <= tol * scale + 1e-30, "%s: tiled and generic %s differ by %.3g of the scale %.3g (tolerance %.1g)" % (
            what, name, worst / max(scale, 1e-300), scale, tol)


def colagrossi_flips_are_single_pairs(prob, sim, n, f_gpu_w, f_ref_w, flipped, vel=None):
    """The Colagrossi density-diffusion term of a pair is switched by |P_i - P_j| >= |rho_i g.r_ij| (forces_kernel.def:1933-1936).
    A pair sitting within rounding of that threshold is decided differently by two pow implementations, and the particle's
    d(rho~)/dt then differs by that ONE pair's term.  For every particle in `flipped` this recomputes, in float64 from global
    positions, the diffusion term and the switch margin of every fluid neighbour, and asserts that the difference between the
    GPU and the oracle is the term of a pair whose switch margin is within rounding of the pressures (a few ulp of (rho/rho0)^gamma
    times B), or the sum of a few such pairs' terms, to 5 % plus the rounding of the rest of the sum."""
    idx = np.nonzero(flipped)[0]
    if not len(idx):
        return
    pp, sp = prob.physparams, prob.simparams
    vel = sim.vel if vel is None else vel
    h = float(sp.slength); R = float(sp.influenceRadius)
    rho0, B, gam, c0 = float(pp.rho0[0]), float(pp.bcoeff[0]), float(pp.gammacoeff[0]), float(pp.sscoeff[0])
    fcoeff = 105.0 / (128.0 * np.pi * h**5)              # Wendland: F = (q - 2)^3 * 105 / (128 pi h^5)
    coeff = float(sp.densityDiffCoeff) if hasattr(sp, "densityDiffCoeff") else float(prob.sphx_params(n).densityDiffCoeff)
    g = np.asarray(pp.gravity[:3], dtype=np.float64)
    gpos = prob.global_pos(sim.pos[:n], sim.hash[:n])
    fluid = (sim.info[:n, 0] & 7) == 0
    wscale = np.abs(f_ref_w).max()
    for i in idx:
        d = gpos[i] - gpos
        r = np.sqrt((d * d).sum(1))
        nb = np.nonzero(fluid & (r < R) & (r > 0))[0]
        rho_i = rho0 * (1.0 + float(vel[i, 3])); rho_j = rho0 * (1.0 + vel[nb, 3].astype(np.float64))
        P_i = B * ((rho_i / rho0)**gam - 1.0); P_j = B * ((rho_j / rho0)**gam - 1.0)
        F = (r[nb] / h - 2.0)**3 * fcoeff
        term = coeff * c0 * (rho_j / rho_i - 1.0) * sim.pos[nb, 3].astype(np.float64) * F / rho0
        margin = np.abs(P_i - P_j) - np.abs(rho_i * (d[nb] @ g))
        diff = abs(float(f_gpu_w[i]) - float(f_ref_w[i]))
        # the pair nearest to its threshold, in units of what two pow implementations can differ by: one ulp of (rho/rho0)^gamma
        # is B * 1.2e-7 in P
        knife = np.abs(margin) / (2.0 * B * 1.2e-7 + 2e-7 * (np.abs(P_i) + np.abs(P_j)))
        on_edge = np.nonzero(knife <= 4.0)[0]
        assert len(on_edge) >= 1, (i, float(knife.min()))
        # on a lattice mirror-image neighbours carry the same pressure, so several pairs of one particle can sit on the threshold
        # together: the difference is the sum of the terms of a subset of them
        t = np.abs(term[on_edge][:12])
        sums = np.array([sum(t[b] for b in range(len(t)) if m >> b & 1) for m in range(1, 1 << len(t))])
        best = sums[np.argmin(np.abs(sums - diff))]
        assert abs(best - diff) <= 0.05 * best + 2e-5 * wscale, (i, diff, t.tolist(), margin[on_edge].tolist())
