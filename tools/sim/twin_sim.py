"""Design simulation for the pair kernel: how many phase-2 iterations does a
wavefront need when every lane owns ND consecutive destinations (in a given
sort order) and walks the UNION of their neighbour sets?

    python tools/sim/twin_sim.py [n1]
"""
import sys
import numpy as np
from scipy.spatial import cKDTree

n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 48
rng = np.random.default_rng(1234)
dx = 1.0 / n1
g = np.arange(n1) * dx
x, y, z = [a.ravel().copy() for a in np.meshgrid(g, g, g, indexing='ij')]
for a in (x, y, z):
    a += 0.1 * dx * rng.uniform(-1, 1, a.size)
P = np.stack([x, y, z], 1)
h = 1.3 * dx
R = 2 * h
cell = R
mn = P.min(0) - 0.01 * (P.max(0) - P.min(0))
c = np.floor((P - mn) / cell).astype(np.int64)
nc = c.max(0) + 1
frac = (P - mn) / cell - c            # position inside the cell, [0,1)
tree = cKDTree(P)
nbr = tree.query_ball_point(P, R * (1 - 1e-12))
nbr = [np.array(v) for v in nbr]
cnt = np.array([len(v) for v in nbr])
print('particles', P.shape[0], 'mean nbrs', cnt.mean())


def order(kind):
    cid = c[:, 0] + nc[0] * (c[:, 1] + nc[1] * c[:, 2])
    if kind == 'cell':
        sub = rng.permutation(P.shape[0]) % 1024      # arbitrary inside the cell
        return np.lexsort((sub, cid))
    sx = lambda b: np.minimum((frac[:, 0] * b).astype(int), b - 1)
    sy = lambda b: np.minimum((frac[:, 1] * b).astype(int), b - 1)
    sz = lambda b: np.minimum((frac[:, 2] * b).astype(int), b - 1)
    if kind == 'x8':
        return np.lexsort((sx(8), cid))
    if kind == 'z2y2x4':
        return np.lexsort((sx(4), sy(2), sz(2), cid))
    if kind == 'z2y2x2':
        return np.lexsort((sx(2), sy(2), sz(2), cid))
    if kind == 'z2y2x8':
        return np.lexsort((sx(8), sy(2), sz(2), cid))
    if kind == 'z3y3x3':
        return np.lexsort((sx(3), sy(3), sz(3), cid))
    if kind == 'z4y4x4':
        return np.lexsort((sx(4), sy(4), sz(4), cid))
    if kind == 'morton8':   # 8x8x8 sub-grid, bit-interleaved
        a, b, d = sx(8), sy(8), sz(8)
        m = np.zeros(P.shape[0], dtype=int)
        for k in range(3):
            m |= ((a >> k) & 1) << (3 * k) | ((b >> k) & 1) << (3 * k + 1) | ((d >> k) & 1) << (3 * k + 2)
        return np.lexsort((m, cid))
    raise ValueError(kind)


def sim(kind, ND):
    o = order(kind)
    n = (len(o) // (64 * ND)) * 64 * ND
    # interior waves only (skip boundary effects): use counts as they are
    its = []
    hits = 0
    un_tot = 0
    for w0 in range(0, n, 64 * ND):
        mx = 0
        for l in range(64):
            ids = o[w0 + l * ND: w0 + (l + 1) * ND]
            if ND == 1:
                u = len(nbr[ids[0]])
            else:
                u = len(np.unique(np.concatenate([nbr[i] for i in ids])))
            un_tot += u
            hits += sum(len(nbr[i]) for i in ids)
            mx = max(mx, u)
        its.append(mx)
    its = np.array(its)
    nd = n
    print('%-8s ND=%d: gathers/dest %.1f  wave iterations/dest-per-lane %.1f (mean union %.1f)  pair-eval efficiency %.2f' % (
        kind, ND, un_tot / nd, its.mean() / 1, un_tot / (n / ND), hits / (its.sum() * 64 * ND)))


for kind in ['cell', 'x8', 'z2y2x2', 'z2y2x4', 'z2y2x8', 'z3y3x3', 'z4y4x4', 'morton8']:
    for ND in (1, 2, 3, 4):
        sim(kind, ND)
