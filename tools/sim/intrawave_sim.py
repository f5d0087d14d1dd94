"""Phase-2 iteration counts if the hits INSIDE a wavefront's own 64 destinations
were served from registers (ds_bpermute) in a loop of their own: per wave
max_l(intra) + max_l(extra) against max_l(intra + extra) today; only the
`extra` iterations issue global gathers."""
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
u = (P - mn) / R
c = np.floor(u).astype(np.int64)
nc = c.max(0) + 1
sub = np.minimum(((u[:, 0] - c[:, 0]) * 8).astype(int), 7)
key = (c[:, 0] + nc[0] * (c[:, 1] + nc[1] * c[:, 2])) * 8 + sub
o = np.argsort(key, kind='stable')
pos = np.empty_like(o); pos[o] = np.arange(o.size)
tree = cKDTree(P)
pairs = tree.query_pairs(R * (1 - 1e-12), output_type='ndarray')
i = np.concatenate([pairs[:, 0], pairs[:, 1], np.arange(P.shape[0])])
j = np.concatenate([pairs[:, 1], pairs[:, 0], np.arange(P.shape[0])])
wi, wj = pos[i] // 64, pos[j] // 64
intra = wi == wj
n = P.shape[0]
a = np.bincount(i[intra], minlength=n)
b = np.bincount(i[~intra], minlength=n)
nw = n // 64
A = a[o][:nw * 64].reshape(nw, 64)
B = b[o][:nw * 64].reshape(nw, 64)
# interior waves only
inner = np.all((c[o][:nw * 64].reshape(nw, 64, 3) >= 2) & (c[o][:nw * 64].reshape(nw, 64, 3) < nc - 2), axis=(1, 2))
A, B = A[inner], B[inner]
print('waves', A.shape[0], 'mean hits/lane %.1f intra %.1f extra %.1f' % ((A + B).mean(), A.mean(), B.mean()))
print('iterations today      %.1f' % (A + B).max(1).mean())
print('split: intra %.1f + extra %.1f = %.1f   (gather iterations: %.1f -> %.1f, %.0f %%)' % (
    A.max(1).mean(), B.max(1).mean(), (A.max(1) + B.max(1)).mean(), (A + B).max(1).mean(), B.max(1).mean(),
    100 * B.max(1).mean() / (A + B).max(1).mean()))
