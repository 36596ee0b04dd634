"""Shared set-up of the SA_BOUNDARY tests: the SABox mirror through the neighbour phase of the ORACLE."""
import numpy as np

from gpusph_amd import defs as D
from gpusph_amd.problem import SABox
from oracle_lib import Oracle, orc_params_from


def analytic_vertex_gamma(problem, st):
    """gamma of a particle ON a planar wall is 1/2, on an edge 1/4, in a corner 1/8 (the fraction of the kernel support
    inside the tank); rim vertices of the open top are treated like face/edge vertices of an infinitely tall wall"""
    g = problem.global_pos(st["pos"], st["hash"])
    dp = problem.m_deltap
    gam = np.ones(len(g))
    for a, L in enumerate((problem.l, problem.w)):
        gam *= np.where((np.abs(g[:, a]) < 0.25 * dp) | (np.abs(g[:, a] - L) < 0.25 * dp), 0.5, 1.0)
    gam *= np.where(np.abs(g[:, 2]) < 0.25 * dp, 0.5, 1.0)
    return gam.astype(np.float32)


def sa_oracle_state(problem=None, **kw):
    """sorted arrays + neighbour list of an SABox, all by the oracle"""
    problem = problem or SABox(**kw)
    n = problem.num_particles
    o = Oracle(orc_params_from(problem.sphx_params(n), problem))
    a = problem.copy_to_array()
    hash_, info = a["hash"].copy(), a["info"].copy()
    pidx = o.fix_hash(hash_, info)
    o.sort(hash_, info, pidx)
    cs, ce, _, pos, vel, newn = o.reorder(a["pos"], a["vel"], info, hash_, pidx, problem.grid_cells, segments=False)
    assert newn == n
    st = dict(problem=problem, oracle=o, n=n, pos=pos, vel=vel, info=info, hash=hash_, cs=cs, ce=ce, pidx=pidx,
              vertices=a["vertices"][pidx], boundelements=a["boundelements"][pidx], gradgamma=a["gradgamma"][pidx])
    sp = problem.simparams
    st["sqinfl"] = float(np.float32(sp.nlSqInfluenceRadius))
    # GPUWorker.cc:1890: (sqrt(nlSqInfluenceRadius) + slength/sfactor/2)^2, in float
    f32 = np.float32
    st["bound_sqinfl"] = float(np.power(f32(np.sqrt(f32(sp.nlSqInfluenceRadius))) + f32(sp.slength) / f32(sp.sfactor) / f32(2.0), f32(2.0), dtype=np.float32))
    st["nl"], st["vertpos"], st["neibs_info"] = o.build_neibs_sa(pos, info, st["vertices"], st["boundelements"], hash_, cs, ce, n, n,
                                                                 st["sqinfl"], st["bound_sqinfl"])
    return st


def list_sections(st, i):
    """(fluid, boundary, vertex) neighbour indices of sorted particle i, decoded from the oracle-format list"""
    p = st["problem"]
    sp = p.simparams
    stride = st["n"]
    nl = st["nl"].reshape(sp.neiblistsize, stride)
    g0 = p.grid_pos_from_hash(st["hash"][i:i + 1])[0]
    out = []
    for first, step in ((0, 1), (sp.neibboundpos, -1), (sp.neibboundpos + 1, 1)):
        res, slot, base = [], first, 0
        while True:
            d = int(nl[slot, i])
            if d == 0xFFFF:
                break
            if d >= D.CELLNUM_ENCODED:
                c = (d >> D.CELLNUM_SHIFT) - 1
                off = np.array([c % 3 - 1, (c // 3) % 3 - 1, c // 9 - 1])
                base = int(st["cs"][int(p.calc_grid_hash((g0 + off)[None, :])[0])])
                d &= D.NEIBINDEX_MASK
            res.append(base + d)
            slot += step
        out.append(res)
    return out
