"""CPU model of the tiled forces kernel's pair phase from the oracle's neighbour lists (no GPU): what the tile size cap,
the order of the home particles inside a tile and the split of a tile's list batches over the eight waves do to the time of
the pair phase.  Issue-time constants from profiles/r03_valu_rate_microbench.txt (ns per batch of 4 pairs: 238.5 per wave
and SIMD when two waves share a SIMD, 350 for a wave that has its SIMD to itself).

  python scripts/tile_model.py [particles] [linearization]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import gpusph_amd.defs as D
from gpusph_amd.problem import DamBreak3D
import oracle_lib as ol

n_target = float(sys.argv[1]) if len(sys.argv) > 1 else 1e6
lin = sys.argv[2] if len(sys.argv) > 2 else "xzy"
prob = DamBreak3D(DamBreak3D.deltap_for(n_target), obstacle=True, linearization=lin)
sim = ol.OracleSim(prob)
for _ in range(int(sys.argv[3]) if len(sys.argv) > 3 else 1):
    sim.step()
sim.iterations = 10
sim.build_neibs()
n = sim.n
A = sim.alloc
nl = np.asarray(sim.nl).view(np.uint16).reshape(128, -1)[:, :n]
end = nl == 0xFFFF
nF = np.argmax(end, axis=0).astype(np.int64)
nB = np.argmax(end[::-1], axis=0).astype(np.int64)
ptype = (np.asarray(sim.info).view(np.uint16).reshape(-1, 4)[:n, 0] & 7)
fluid = ptype == 0
wF = nF.copy()                       # every tiled particle type walks its fluid section (DYN boundary too)
wB = np.where(fluid, nB, 0)          # the boundary section only for fluid particles
print("particles", n, "mean stored neighbours", (nF + nB).mean(), "walked", (wF + wB).mean())

c1, c2, c3 = D.LINEARIZATIONS[lin]
gs = prob.m_gridsize
gs1, gs2, gs3 = int(gs[c1]), int(gs[c2]), int(gs[c3])
cs = np.asarray(sim.cs).astype(np.int64); ce = np.asarray(sim.ce).astype(np.int64)
EMPTY = 0xFFFFFFFF
cnt = np.where(cs == EMPTY, 0, ce - cs).reshape(gs3, gs2, gs1)
start = np.where(cs == EMPTY, 0, cs).reshape(gs3, gs2, gs1)


def build_tiles(pmax, wcap, maxcells=14):
    tiles = []
    for b3 in range((gs3 + 1)//2):
        for b2 in range((gs2 + 1)//2):
            g2, g3 = 2*b2, 2*b3
            rows = [(g2 + (r & 1), g3 + (r >> 1)) for r in range(4)]
            rows = [(a, b) for a, b in rows if a < gs2 and b < gs3]
            wrows = [(g2 - 1 + i, g3 - 1 + j) for j in range(4) for i in range(4)]
            wrows = [(a, b) for a, b in wrows if 0 <= a < gs2 and 0 <= b < gs3]
            colw = np.zeros(gs1 + 2, dtype=np.int64)
            for a, b in wrows:
                colw[1:-1] += cnt[b, a, :]
            colh = np.zeros(gs1, dtype=np.int64)
            for a, b in rows:
                colh += cnt[b, a, :]
            ca, hsum, wc = 0, 0, 0
            for c in range(gs1):
                ncol = colh[c]
                fits = hsum and ncol and hsum + ncol <= pmax and wc + colw[c + 2] <= wcap and c - ca + 1 <= maxcells
                if fits:
                    hsum += ncol; wc += colw[c + 2]
                else:
                    if hsum:
                        tiles.append((rows, ca, c, wc))
                    hsum = 0
                    if ncol:
                        ca, hsum = c, ncol
                        wc = colw[c] + colw[c + 1] + colw[c + 2]
            if hsum:
                tiles.append((rows, ca, gs1, wc))
    return tiles


def home(t):
    rows, ca, cb, wc = t
    idx = []
    for a, b in rows:
        k = cnt[b, a, ca:cb].sum()
        if k:
            nz = np.nonzero(cnt[b, a, ca:cb])[0][0]
            s = start[b, a, ca + nz]
            idx.append(np.arange(s, s + k))
    return np.concatenate(idx) if idx else np.zeros(0, dtype=np.int64)


def r4(x):
    return (x + 3)//4*4


def model(pmax, wcap, lanes, key):
    tiles = build_tiles(pmax, wcap)
    tot = dict(ideal=0.0, cur=0.0, split=0.0, tiles=0, win=0, pairs=0, slots=0, runs=0)
    for t in tiles:
        idx = home(t)
        f, b = wF[idx], wB[idx]
        if (f + b).sum() == 0:
            continue
        tot["tiles"] += 1; tot["win"] += t[3]; tot["pairs"] += (f + b).sum()
        tot["ideal"] += 238.5*(f + b).sum()/4/64/4
        # current scheme: chunks of 64 in home order on `lanes` lanes, paired by rank on the SIMDs
        nch = (len(idx) + 63)//64
        bat = []
        for c in range(nch):
            s = slice(64*c, 64*c + 64)
            bat.append((r4(f[s].max()) + r4(b[s].max()))//4)
        bat = sorted(bat, reverse=True) + [0]*(8 - nch) if nch <= 8 else sorted(bat, reverse=True)
        if nch <= 8:
            simd = [(bat[k], bat[7 - k]) for k in range(4)]
            tot["cur"] += max(350.0*a + 127.0*bb for a, bb in simd)
        else:
            tot["cur"] += 238.5*sum(bat)/4*1.2
        # new scheme: sorted by key, all batches of the tile split evenly over 8 waves
        if key == "F":
            o = np.lexsort((-b, -f))
        elif key == "F+B":
            o = np.argsort(-(f + b), kind="stable")
        else:
            o = np.arange(len(idx))
        fs, bs = f[o], b[o]
        T = 0
        sb = []
        for c in range(nch):
            s = slice(64*c, 64*c + 64)
            sb.append((r4(fs[s].max()) + r4(bs[s].max()))//4)
            T += sb[-1]
        if nch <= 8:      # sorted lanes, whole chunks per wave, paired by rank on the SIMDs (no split)
            sb = sorted(sb, reverse=True) + [0]*(8 - nch)
            tot["sortpair"] = tot.get("sortpair", 0.0) + max(350.0*sb[k] + 127.0*sb[7 - k] for k in range(4))
        else:
            tot["sortpair"] = tot.get("sortpair", 0.0) + 477.0*(-(-T//8))*1.15
        tot["slots"] += T*4*64
        share = -(-T//8)
        tot["split"] += 477.0*share
        tot["runs"] += nch + 7
    return tot, len(tiles)


for pmax, wcap, key in ((512, 3200, "F"), (512, 3200, "none"), (640, 2900, "F"), (640, 3200, "F")):
    tot, nt = model(pmax, wcap, 512, key)
    print("pmax %d wcap %d key %-4s: tiles %d (with pairs %d) home/tile %.0f window/home %.2f | pair-phase time / ideal: current %.3f  sorted+paired %.3f  split %.3f | lane use of the split lists %.3f"
          % (pmax, wcap, key, nt, tot["tiles"], n/ max(nt, 1), tot["win"]/n, tot["cur"]/tot["ideal"], tot.get("sortpair", 0)/tot["ideal"], tot["split"]/tot["ideal"], tot["pairs"]/max(tot["slots"], 1)))
