"""Row-distance-dependent x window for phase 1 of the pair kernel: what would it save?

Phase 1 tests, per lane, the candidates of a neighbouring row of cells inside the lane's x window -- its own x sub-bin
+- (NSUB + 1) = +-9 sub-bins.  A particle at fractional position (fy, fz) inside its cell can only reach a row at (dy, dz)
within sqrt(1 - ddy^2 - ddz^2) cells along x (ddy = 0 for dy = 0, 1 - fy for dy = +1, fy for dy = -1, same for z), so its
window could shrink to ceil(8 * that) + 1 sub-bins.  But the candidate loop runs in groups of 8 candidates whose trip
count is WAVE-UNIFORM -- the maximum over the 64 lanes of a wavefront -- and every candidate the smaller window drops is
one the fp32 distance test rejects anyway (the hit masks, and with them phase 2, are identical).  So the only thing a
smaller window can save is loop trips, and those follow the LARGEST window of 64 lanes.

This script takes the bench's jittered lattice (S-cube), orders it as the kernel does (fine keys: cell * 8 + x sub-bin),
cuts it into wavefronts of 64 consecutive particles and prints, per class of row step (centre / edge / corner):
the mean per-lane window, the mean per-wavefront MAXIMUM window, and the number of 8-candidate groups each implies.

    python tools/sim/xwindow_sim.py [n1=100]
"""
import sys

import numpy as np

n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 100
NSUB = 8
rng = np.random.default_rng(1234)
dx = 1.0 / n1
g = np.arange(n1) * dx
x, y, z = [a.ravel().copy() for a in np.meshgrid(g, g, g, indexing='ij')]
for a in (x, y, z):
    a += 0.1 * dx * rng.uniform(-1, 1, a.size)
P = np.stack([x, y, z], 1)
cs = 2.0 * 1.3 * dx                                  # WendlandQuintic, hdx 1.3
mn = P.min(0) - 0.01 * (P.max(0) - P.min(0))
u = (P - mn) / cs
c = np.floor(u).astype(np.int64)
f = u - c                                            # fractional position inside the cell
nc = c.max(0) + 1
sub = np.minimum((f[:, 0] * NSUB).astype(np.int64), NSUB - 1)
key = (c[:, 0] + nc[0] * (c[:, 1] + nc[1] * c[:, 2])) * NSUB + sub
o = np.argsort(key, kind='stable')
f = f[o]
per_bin = P.shape[0] / float(nc.prod() * NSUB)       # particles per x sub-bin (candidates per sub-bin of a row)
nw = f.shape[0] // 64
F = f[:nw * 64].reshape(nw, 64, 3)
full = 2 * (NSUB + 1) + 1                            # 19 sub-bins: today's window


def win(ddy, ddz):
    r2 = np.maximum(0.0, 1.0 - ddy ** 2 - ddz ** 2)
    half = np.where(r2 > 0, np.ceil(NSUB * np.sqrt(r2)) + 1, 0)     # +1: the slack of the sub-bin index, as XWIN
    return np.minimum(2 * half + 1, full) * (r2 > 0)


def dd(frac, d):
    return np.zeros_like(frac) if d == 0 else (1.0 - frac if d > 0 else frac)


print('S-cube %d^3, %.2f particles per x sub-bin, %d wavefronts; window of today: %d sub-bins = %.1f candidates per row'
      % (n1, per_bin, nw, full, full * per_bin))
print('%-8s %5s  %22s  %26s  %s' % ('class', 'rows', 'mean window per LANE', 'mean MAX window per WAVE', '8-candidate groups per row step: today / per-lane mean / wave max'))
tot = {'today': 0.0, 'lane': 0.0, 'wave': 0.0}
for name, steps in (('centre', [(0, 0)]), ('edge', [(1, 0), (-1, 0), (0, 1), (0, -1)]),
                    ('corner', [(1, 1), (1, -1), (-1, 1), (-1, -1)])):
    lane, wave = [], []
    for dy, dz in steps:
        w = win(dd(F[:, :, 1], dy), dd(F[:, :, 2], dz))
        lane.append(w.mean())
        wave.append(w.max(axis=1).mean())
    lane, wave = float(np.mean(lane)), float(np.mean(wave))
    grp = lambda wdw: np.ceil(wdw * per_bin / 8.0)
    print('%-8s %5d  %10.2f (%5.1f %%)  %14.2f (%5.1f %%)        %d / %.2f / %.2f'
          % (name, len(steps), lane, 100 * lane / full, wave, 100 * wave / full, grp(full), lane * per_bin / 8, wave * per_bin / 8))
    tot['today'] += len(steps) * full
    tot['lane'] += len(steps) * lane
    tot['wave'] += len(steps) * wave
print('all 9 row steps: candidate TESTS per lane %.1f %% of today (what a lane could skip), loop TRIPS %.1f %% of today'
      % (100 * tot['lane'] / tot['today'], 100 * tot['wave'] / tot['today']))
