"""Which share of a destination's true neighbours lies in its own row of cells
(cy, cz equal), in the dz=0 plane, and inside its own wavefront / workgroup tile
(consecutive cell-ordered destinations)?"""
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
R = 2.6 * dx
mn = P.min(0) - 0.01 * (P.max(0) - P.min(0))
c = np.floor((P - mn) / R).astype(np.int64)
nc = c.max(0) + 1
cid = c[:, 0] + nc[0] * (c[:, 1] + nc[1] * c[:, 2])
o = np.argsort(cid, kind='stable')
pos = np.empty_like(o); pos[o] = np.arange(o.size)
row = c[:, 1] + nc[1] * c[:, 2]
tree = cKDTree(P)
pairs = tree.query_pairs(R * (1 - 1e-12), output_type='ndarray')
i = np.concatenate([pairs[:, 0], pairs[:, 1], np.arange(P.shape[0])])
j = np.concatenate([pairs[:, 1], pairs[:, 0], np.arange(P.shape[0])])
# interior destinations only
inner = np.all((c[i] >= 2) & (c[i] < nc - 2), axis=1)
i, j = i[inner], j[inner]
print('pairs/dest %.1f' % (len(i) / inner_count if (inner_count := len(np.unique(i))) else 0))
print('own row        %.3f' % np.mean(row[i] == row[j]))
print('dz=0 plane     %.3f' % np.mean(c[i, 2] == c[j, 2]))
print('dy=0 or dz=0   %.3f' % np.mean((c[i, 2] == c[j, 2]) | (c[i, 1] == c[j, 1])))
for T in (64, 256, 512, 1024):
    print('same %4d-tile  %.3f   same tile or +-1 tile in the row %.3f' % (T, np.mean(pos[i] // T == pos[j] // T),
          np.mean((np.abs(pos[i] // T - pos[j] // T) <= 1) & (row[i] == row[j]))))
