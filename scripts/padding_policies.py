"""CPU study (oracle lists, no GPU) of what pads the tile lists' stream: per tile, the home particles are cut into chunks of 64
lanes and every chunk's section is padded to its longest list, rounded to the batch.  Policies:
  cur     sort by F (longest first), sections F and B padded separately, batches of 4   (tile_lists_kernel today)
  r2      ... batches of 2
  merged  ONE section per lane = F entries then B entries, sorted by F + B, batches of 4
  merged2 ... batches of 2
  python scripts/padding_policies.py [particles] [linearization] [steps]
"""
import os, sys
sys.argv = sys.argv[:1] + sys.argv[1:]
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tile_model.py")).read().split("def r4(x):")[0])


def rnd(x, q):
    return (x + q - 1)//q*q


tiles = build_tiles(640, 2876)
used = 0
slots = dict(cur=0, r2=0, merged=0, merged2=0, mergedF=0)
idle = 0
for t in tiles:
    idx = home(t)
    f, b = wF[idx], wB[idx]
    if (f + b).sum() == 0:
        continue
    used += (f + b).sum()
    o = np.lexsort((-b, -f)); fs, bs = f[o], b[o]
    o2 = np.argsort(-(f + b), kind="stable"); ts = (f + b)[o2]
    tF = (f + b)[o]
    nch = (len(idx) + 63)//64
    idle += (nch*64 - len(idx))
    for c in range(nch):
        s = slice(64*c, 64*c + 64)
        slots["cur"] += 64*(rnd(fs[s].max(), 4) + rnd(bs[s].max(), 4))
        slots["r2"] += 64*(rnd(fs[s].max(), 2) + rnd(bs[s].max(), 2))
        slots["merged"] += 64*rnd(ts[s].max(), 4)
        slots["merged2"] += 64*rnd(ts[s].max(), 2)
        slots["mergedF"] += 64*rnd(tF[s].max(), 4)
print("tiles", len(tiles), "entries walked", used, "idle lanes per tile %.1f" % (idle/len(tiles)))
for k, v in slots.items():
    print("%-8s slots %.4e  padding %.2f %% of the slots" % (k, v, 100.0*(v - used)/v))
