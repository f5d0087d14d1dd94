"""Slow, bit-defining oracle: drive the REFERENCE's own Python classes.

TEST INFRASTRUCTURE ONLY, and only usable in the build container (it imports
``/root/reference``, which does not exist on the GPU box).  Used by
``tests/golden/make_golden.py`` to produce the committed golden vectors.

What runs here is the reference's code: ``pysph.sph.equation.Group`` (real
precomputed-symbol code blocks and their dependency order,
equation.py:188-344,589-625), ``pysph.sph.acceleration_eval.AccelerationEval``
/ ``MegaGroup`` (real regrouping, acceleration_eval.py:94-162), the equation
classes' ``initialize/loop/post_loop`` bodies and ``pysph.base.kernels``
executed as plain Python floats -- the way the reference's
``sph/tests/test_equations.py:102-120`` exercises them.  What is restated in
Python (because the Cython modules need the absent ``cyarray``) is only

  * the linked-list neighbour search (linked_list_nnps.pyx:92-196,235-383,
    nnps_base.pyx:942-978,1520-1575), checked below against brute force, and
  * the loop nest that the mako template emits
    (acceleration_eval_cython.mako:10-154).
"""
import math
import os
import sys
from inspect import getfullargspec

_HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = '/root/reference'


def setup_reference_imports():
    if not os.path.isdir(REFERENCE):
        raise RuntimeError('%s not present (build container only)' % REFERENCE)
    stubs = os.path.join(_HERE, '_stubs')
    for p in (REFERENCE, stubs):
        if p not in sys.path:
            sys.path.insert(0, p)


UINT_MAX = 2 ** 32 - 1


class ListPA(object):
    """Particle array whose properties are Python lists of floats, so every
    arithmetic operation is a CPython float op (IEEE double, libm pow/sqrt)."""

    def __init__(self, name, props, n_real=None, constants=None):
        self.name = name
        self.properties = {k: [float(x) for x in v] for k, v in props.items()}
        self.constants = dict(constants or {})
        self.n = len(next(iter(self.properties.values())))
        self.n_real = self.n if n_real is None else n_real

    def get_number_of_particles(self, real=False):
        return self.n_real if real else self.n


class PyLinkedListNNPS(object):
    def __init__(self, dim, particles, radius_scale=2.0):
        self.dim = dim
        self.particles = particles
        self.radius_scale = radius_scale

    def update(self):
        rs = self.radius_scale
        hmax, hmin = -1.0, sys.float_info.max
        for pa in self.particles:
            h = pa.properties['h']
            hmax = max(hmax, max(h))
            hmin = min(hmin, min(h))
        cell_size = rs * hmax
        if cell_size < 1e-6:
            cell_size = 1.0
        self.cell_size = cell_size
        self.hmin = rs * hmin
        mx = [-1e100] * 3
        mn = [1e100] * 3
        for pa in self.particles:
            for k, c in enumerate('xyz'):
                mx[k] = max(mx[k], max(pa.properties[c]))
                mn[k] = min(mn[k], min(pa.properties[c]))
        ext = [mx[k] - mn[k] for k in range(3)]
        for k in range(3):
            mn[k] -= ext[k] * 0.01
            mx[k] += ext[k] * 0.01
        if all(abs(mx[k] - mn[k]) < 1e-12 for k in range(3)):
            for k in range(3):
                mn[k] -= 0.5
                mx[k] += 0.5
        self.xmin, self.xmax = mn, mx
        c1 = 1. / cell_size
        nc = [int(math.ceil(c1 * (mx[k] - mn[k]))) for k in range(3)]
        nc = [1 if v == 0 else v for v in nc]
        self.nc = nc
        n_cells = nc[0]
        if self.dim == 2:
            n_cells = nc[0] * nc[1]
        if self.dim == 3:
            n_cells = nc[0] * nc[1] * nc[2]
        self.n_cells = n_cells
        self.heads, self.nexts = [], []
        for pa in self.particles:
            head = [UINT_MAX] * n_cells
            nxt = [UINT_MAX] * pa.n
            x, y, z = (pa.properties[c] for c in 'xyz')
            for i in range(pa.n):
                cx = int(math.floor((x[i] - mn[0]) / cell_size))
                cy = int(math.floor((y[i] - mn[1]) / cell_size))
                cz = int(math.floor((z[i] - mn[2]) / cell_size))
                cid = cx + nc[0] * cy + nc[0] * nc[1] * cz
                nxt[i] = head[cid]
                head[cid] = i
            self.heads.append(head)
            self.nexts.append(nxt)

    def neighbors(self, src, dst, d_idx):
        S, D = self.particles[src].properties, self.particles[dst].properties
        head, nxt = self.heads[src], self.nexts[src]
        rs, cs = self.radius_scale, self.cell_size
        x, y, z = D['x'][d_idx], D['y'][d_idx], D['z'][d_idx]
        mn, nc = self.xmin, self.nc
        c0 = [int(math.floor((x - mn[0]) / cs)), int(math.floor((y - mn[1]) / cs)),
              int(math.floor((z - mn[2]) / cs))]
        hi2 = rs * D['h'][d_idx]
        hi2 *= hi2
        sx, sy, sz, sh = S['x'], S['y'], S['z'], S['h']
        out = []
        for ix in (-1, 0, 1):
            for iy in (-1, 0, 1):
                for iz in (-1, 0, 1):
                    cx, cy, cz = c0[0] + ix, c0[1] + iy, c0[2] + iz
                    if not (nc[0] > cx > -1 and nc[1] > cy > -1 and nc[2] > cz > -1):
                        continue
                    ci = cx + nc[0] * cy + nc[0] * nc[1] * cz
                    if not (-1 < ci < self.n_cells):
                        continue
                    j = head[ci]
                    while j != UINT_MAX:
                        hj2 = rs * sh[j]
                        hj2 *= hj2
                        dx, dy, dz = sx[j] - x, sy[j] - y, sz[j] - z
                        r2 = dx * dx + dy * dy + dz * dz
                        if (r2 < hi2) or (r2 < hj2):
                            out.append(j)
                        j = nxt[j]
        return out

    def brute_force(self, src, dst, d_idx):
        S, D = self.particles[src].properties, self.particles[dst].properties
        rs = self.radius_scale
        xi, yi, zi = D['x'][d_idx], D['y'][d_idx], D['z'][d_idx]
        hi = D['h'][d_idx] * rs
        out = []
        for j in range(self.particles[src].n):
            hj = rs * S['h'][j]
            dx, dy, dz = xi - S['x'][j], yi - S['y'][j], zi - S['z'][j]
            r2 = dx * dx + dy * dy + dz * dz
            if (r2 < hi * hi) or (r2 < hj * hj):
                out.append(j)
        return out


class RefEval(object):
    """Execute ``AccelerationEval.compute`` semantics with the reference's
    objects.  `a_eval` is a real ``pysph.sph.acceleration_eval.AccelerationEval``."""

    def __init__(self, a_eval, nnps):
        self.a_eval = a_eval
        self.nnps = nnps
        self.arrays = dict((pa.name, pa) for pa in a_eval.particle_arrays)
        self.index = dict((pa.name, i)
                          for i, pa in enumerate(a_eval.particle_arrays))
        self.kernel = a_eval.kernel
        self._code = {}

    def _call(self, eq, kind, ns):
        meth = getattr(eq, kind)
        args = [a for a in getfullargspec(meth).args if a != 'self']
        meth(*[ns[a] for a in args])

    def _ns(self, dst, src, t, dt):
        ns = {'t': t, 'dt': dt, 'SPH_KERNEL': self.kernel}
        for k, v in dst.properties.items():
            ns['d_' + k] = v
        for k, v in dst.constants.items():
            ns['d_' + k] = v
        if src is not None:
            for k, v in src.properties.items():
                ns['s_' + k] = v
            for k, v in src.constants.items():
                ns['s_' + k] = v
        return ns

    def _precomp(self, group):
        key = id(group)
        if key not in self._code:
            blocks = []
            for name, cb in group.precomputed.items():
                blocks.append(compile(cb.code, '<precomputed %s>' % name, 'exec'))
            ctx = {}
            for name, cb in group.precomputed.items():
                v = cb.context[name]
                ctx[name] = list(v) if isinstance(v, (list, tuple)) else v
            self._code[key] = (blocks, ctx)
        return self._code[key]

    def compute(self, t, dt):
        for mg in self.a_eval.mega_groups:
            assert not mg.has_subgroups and not mg.iterate
            self._do_group(mg, t, dt)

    def _do_group(self, group, t, dt):
        K = self.kernel
        for dest, (no_src, sources, all_eqs) in group.data.items():
            dst = self.arrays[dest]
            start = group.start_idx
            if isinstance(start, str):
                start = int(dst.properties.get(start, dst.constants.get(start))[0])
            if group.stop_idx is None:
                stop = dst.get_number_of_particles(group.real)
            elif isinstance(group.stop_idx, str):
                s = group.stop_idx
                stop = int(dst.properties.get(s, dst.constants.get(s))[0])
            else:
                stop = group.stop_idx
            ns = self._ns(dst, None, t, dt)
            for d_idx in range(start, stop):
                ns['d_idx'] = d_idx
                for eq in all_eqs.equations:
                    if hasattr(eq, 'initialize'):
                        self._call(eq, 'initialize', ns)
            for d_idx in range(start, stop):
                ns['d_idx'] = d_idx
                for eq in no_src.equations:
                    if hasattr(eq, 'loop'):
                        self._call(eq, 'loop', ns)
            for source, eq_group in sources.items():
                src = self.arrays[source]
                ns = self._ns(dst, src, t, dt)
                blocks, ctx = self._precomp(eq_group)
                ns.update(ctx)
                ns.update(KERNEL=K.kernel, GRADIENT=K.gradient, DWDQ=K.dwdq,
                          GRADH=K.gradient_h, DELTAP=K.get_deltap(),
                          sqrt=math.sqrt)
                si, di = self.index[source], self.index[dest]
                loops = [eq for eq in eq_group.equations if hasattr(eq, 'loop')]
                for d_idx in range(start, stop):
                    ns['d_idx'] = d_idx
                    for s_idx in self.nnps.neighbors(si, di, d_idx):
                        ns['s_idx'] = s_idx
                        for b in blocks:
                            exec(b, ns)
                        for eq in loops:
                            self._call(eq, 'loop', ns)
            ns = self._ns(dst, None, t, dt)
            for d_idx in range(start, stop):
                ns['d_idx'] = d_idx
                for eq in all_eqs.equations:
                    if hasattr(eq, 'post_loop'):
                        self._call(eq, 'post_loop', ns)
