"""Phase 2 with the row tile's RECORDS staged in LDS: per row tile the wavefront
runs k_r iterations whose records come from LDS (ds_read_b128, ~11 cycles per
piece), a lane's hits beyond k_r stay in its deferred list and are gathered from
global memory at the end as today (~44 cycles per piece).  What k_r policy keeps
the total iteration count near today's max_l(sum_r H[l, r])?"""
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
n = P.shape[0]
i = np.concatenate([pairs[:, 0], pairs[:, 1], np.arange(n)])
j = np.concatenate([pairs[:, 1], pairs[:, 0], np.arange(n)])
row = c[:, 1] + nc[1] * c[:, 2]
nw = n // 64
co = c[o][:nw * 64].reshape(nw, 64, 3)
inner = np.all((co >= 2) & (co < nc - 2), axis=(1, 2))
w_of = pos // 64
# H[wave][lane, row]
order = np.argsort(w_of[i], kind='stable')
i, j = i[order], j[order]
bounds = np.searchsorted(w_of[i], np.arange(nw + 1))
res = {}
tot_today = []
for w in np.nonzero(inner)[0][::7]:
    ii, jj = i[bounds[w]:bounds[w + 1]], j[bounds[w]:bounds[w + 1]]
    lane = pos[ii] - w * 64
    rows, rid = np.unique(row[jj], return_inverse=True)
    H = np.zeros((64, rows.size), int)
    np.add.at(H, (lane, rid), 1)
    today = H.sum(1).max()
    tot_today.append(today)
    for name, kfun in (('all rows to the end (=today)', lambda h: 0),
                       ('k = max (no deferral)', lambda h: h.max()),
                       ('k = min', lambda h: h.min()),
                       ('k = q25', lambda h: int(np.quantile(h, 0.25))),
                       ('k = median', lambda h: int(np.median(h))),
                       ('k = q75', lambda h: int(np.quantile(h, 0.75))),
                       ('k = mean-1', lambda h: max(int(h.mean()) - 1, 0))):
        k = np.array([kfun(H[:, r]) for r in range(rows.size)])
        lds_it = k.sum()
        D = np.maximum(H - k[None, :], 0).sum(1)
        useful_lds = np.minimum(H, k[None, :]).sum()
        res.setdefault(name, []).append((lds_it, D.max(), useful_lds / max(lds_it * 64, 1), rows.size))
print('waves', len(tot_today), 'iterations today %.1f' % np.mean(tot_today))
for name, v in res.items():
    v = np.array(v, float)
    lds, gat = v[:, 0].mean(), v[:, 1].mean()
    valu = (lds + gat) * 400 / 4 + 4000
    ta = gat * 220 + 1100
    ldsc = lds * 55 + 700
    print('%-30s LDS it %5.1f (lane use %.2f) + gather it %5.1f = %5.1f | rows %.1f | per-CU cycles/wave: VALU %5.0f TA %5.0f LDS %5.0f' % (
        name, lds, v[:, 2].mean(), gat, lds + gat, v[:, 3].mean(), valu, ta, ldsc))

# ---- symmetric row groups: a lane short of hits in row (dy=+1) has more in (dy=-1)
print('\nphase 2 per GROUP of rows (all of a group staged together), no deferral:')
def groups_of(kind):
    if kind == 'rows':
        return [[(dy, dz)] for dz in (-1, 0, 1) for dy in (-1, 0, 1)]
    if kind == 'centre | dy pair | dz pair | corners':
        return [[(0, 0)], [(-1, 0), (1, 0)], [(0, -1), (0, 1)], [(-1, -1), (1, -1), (-1, 1), (1, 1)]]
    if kind == 'dz planes':
        return [[(dy, dz) for dy in (-1, 0, 1)] for dz in (-1, 0, 1)]
    if kind == 'centre+dy pair | rest':
        return [[(0, 0), (-1, 0), (1, 0)], [(0, -1), (0, 1), (-1, -1), (1, -1), (-1, 1), (1, 1)]]
    if kind == 'centre | faces | corners':
        return [[(0, 0)], [(-1, 0), (1, 0), (0, -1), (0, 1)], [(-1, -1), (1, -1), (-1, 1), (1, 1)]]
    if kind == 'all':
        return [[(dy, dz) for dy in (-1, 0, 1) for dz in (-1, 0, 1)]]
out = {}
for w in np.nonzero(inner)[0][::7]:
    ii, jj = i[bounds[w]:bounds[w + 1]], j[bounds[w]:bounds[w + 1]]
    lane = pos[ii] - w * 64
    # a wavefront that straddles a row end is skipped here (two row segments)
    if np.unique(row[ii]).size != 1:
        continue
    dy = c[jj, 1] - c[ii, 1]
    dz = c[jj, 2] - c[ii, 2]
    for kind in ('rows', 'centre | dy pair | dz pair | corners', 'dz planes', 'centre+dy pair | rest',
                 'centre | faces | corners', 'all'):
        tot, recs = 0, []
        for grp in groups_of(kind):
            sel = np.zeros(ii.size, bool)
            for (a, b) in grp:
                sel |= (dy == a) & (dz == b)
            h = np.bincount(lane[sel], minlength=64)
            tot += h.max()
            recs.append(np.unique(jj[sel]).size)
        out.setdefault(kind, []).append((tot, max(recs), sum(recs)))
for kind, v in out.items():
    v = np.array(v, float)
    print('%-40s iterations %6.1f   largest staged group %5.0f records (%.1f KB)   staged per wave %.0f' % (
        kind, v[:, 0].mean(), v[:, 1].mean(), v[:, 1].mean() * 80 / 1024, v[:, 2].mean()))
