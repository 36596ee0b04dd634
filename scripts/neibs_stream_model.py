#!/usr/bin/env python3
"""What per-lane candidate streams and cell pruning could give build_neibs_kernel, counted on a real particle distribution before
writing either (round 5).  build_neibs_kernel walks the 27 neighbour cells in a wave-uniform loop: for every cell the wave runs
as many batches of four candidates as its fullest lane needs.  Alternatives counted here, per wave of 64 consecutive particles
of the sorted DamBreak3D state (~1.25 M particles, the bench's problem at a smaller size), batches of four candidates:
  now            sum over the 27 cells of the maximum over the lanes
  streams        every lane walks its own concatenated stream; the wave runs until its longest lane is done (max over lanes of the sums)
  streams+prune  ... and a lane skips the cells whose box lies farther than the influence radius from the particle
Result (printed below; fluid-only waves): 142.7 -> 126.6 (-11 %) -> 121.8 (-15 %), although the MEAN lane would need 93: every wave
has a lane near its cell's centre that can prune nothing.  A stream costs ~15 more instructions per batch of ~82 (per-lane cell
advance under partial exec in almost every batch) and LDS for the lanes' cell tables (occupancy 6.5 -> 2.4 waves per SIMD):
the counts do not pay for it.  Not built."""
import sys; import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from gpusph_amd.problem import DamBreak3D
from gpusph_amd import defs as D
import oracle_lib as ol
prob = DamBreak3D(0.0052, obstacle=True, linearization="xzy")
sim = ol.OracleSim(prob)
sim.build_neibs()
n = sim.n
print("particles", n)
# after a few steps positions are off-lattice; approximate with jitter? use as is + 3 oracle steps would be slow; use lattice state
pos = sim.pos[:n]; hsh = sim.hash[:n] & D.CELLTYPE_BITMASK
g = prob.grid_pos_from_hash(hsh)
gs = prob.m_gridsize
cs = sim.cs; ce = sim.ce
ncells = len(cs)
info = sim.info[:n]
fluid = (info[:,0] & 7) == 0
# fluid end per cell
start = cs.astype(np.int64); end = ce.astype(np.int64)
valid = start != 0xFFFFFFFF
fend = start.copy()
# count fluid per cell
cellcount_f = np.bincount(hsh[fluid].astype(np.int64), minlength=ncells)
fend = np.where(valid, start + cellcount_f, 0)
flen = np.where(valid, cellcount_f, 0)
R2 = float(prob.simparams.influenceRadius)**2
csz = prob.m_cellsize
c1,c2,c3 = D.LINEARIZATIONS[prob.linearization]
def cell_hash(gx,gy,gz):
    gg = np.stack([gx,gy,gz],axis=1)
    ok = ((gg>=0)&(gg<gs)).all(axis=1)
    gg = np.clip(gg,0,gs-1)
    return np.where(ok, (gg[:,c3]*gs[c2]+gg[:,c2])*gs[c1]+gg[:,c1], -1)
# particles that walk: all (DYN boundary)
nw = (n//64)*64
cur = np.zeros(nw//64); stream = np.zeros((nw,)); streamp = np.zeros((nw,))
for c in range(27):
    dx,dy,dz = c%3-1,(c//3)%3-1,c//9-1
    h = cell_hash(g[:nw,0]+dx, g[:nw,1]+dy, g[:nw,2]+dz)
    L = np.where(h>=0, flen[np.clip(h,0,None)], 0)
    b = (L+3)//4
    cur += b.reshape(-1,64).max(axis=1)
    stream += b
    # pruning: distance from particle to the cell box
    off = np.array([dx,dy,dz])*csz
    d = np.maximum(0, np.abs(pos[:nw,:3]-off) - csz/2)
    pr = (d*d).sum(axis=1) > R2*1.0001
    streamp += np.where(pr, 0, b)
ideal = (np.where(True, 0, 0))
print("batches per wave now (sum over cells of max over lanes): mean %.1f" % cur.mean())
print("per-lane streams, max over lanes: mean %.1f" % stream.reshape(-1,64).max(axis=1).mean())
print("per-lane streams + pruning: mean %.1f ; mean per lane %.1f" % (streamp.reshape(-1,64).max(axis=1).mean(), streamp.mean()))
print("fluid home lanes only:")
fl = fluid[:nw]
w = fl.reshape(-1,64).all(axis=1)      # waves of fluid particles only
print("waves all-fluid: %d of %d" % (w.sum(), len(w)))
print("  now %.1f, streams %.1f, streams+pruning %.1f (mean per lane %.1f)" % (cur[w].mean(), stream.reshape(-1,64).max(axis=1)[w].mean(),
      streamp.reshape(-1,64).max(axis=1)[w].mean(), streamp.reshape(-1,64)[w].mean()))
