"""Slow reference evaluator for ARBITRARY Equation objects with Python bodies.

TEST INFRASTRUCTURE ONLY (see oracle/sph_oracle.h): only tests/ may import this.

It runs the ``initialize / loop / post_loop`` methods of equation objects as
plain Python -- exactly how the reference's own tests exercise equations
(pysph/sph/tests/test_equations.py:102-120) -- inside a restatement of the loop
nest the reference generates (acceleration_eval_cython.mako:10-154):

    for every group, for every destination (first-appearance order):
        initialize (all equations)            mako :39-47
        loop of equations without sources     mako :50-58
        for every source (first appearance):  mako :61-127
            for every destination particle: neighbours, precomputed symbols
            (equation.py:188-297), loop of the equations having this source
        post_loop (all equations)             mako :130-138

Neighbours come from the C oracle's cell list (oracle.OracleNNPS.get_csr, itself
pinned to the reference's neighbour sets by tests/test_oracle_golden.py); pair
symbols use ``math.sqrt`` and true divisions as the reference's precomputed
code blocks do.  It is the checker for the *generated-family* path
(pysph_amd/codegen.py), which translates the same method bodies to HIP; the
anchor to the reference itself is tests/golden/tvf_wall.npz (the reference's own
classes, executed in the build container).
"""
import math
from inspect import getfullargspec

import numpy as np

VEC = ('XIJ', 'VIJ', 'DWIJ', 'DWI', 'DWJ')


class PyEval(object):
    def __init__(self, arrays, groups, kernel, nnps):
        """arrays: particle arrays with .properties/.constants (numpy);
        groups: Group objects (equations with Python bodies);
        kernel: object with kernel(xij, rij, h) and gradient(xij, rij, h, grad);
        nnps: oracle.OracleNNPS over the same arrays (updated)."""
        self.arrays = list(arrays)
        self.names = [a.name for a in arrays]
        self.groups = groups
        self.kernel = kernel
        self.nnps = nnps

    def _arr(self, pa, name):
        if name in pa.properties:
            return pa.properties[name]
        return pa.constants[name]

    def _call(self, eq, meth, ns):
        fn = getattr(type(eq), meth, None)
        if fn is None:
            return
        args = getfullargspec(fn).args[1:]
        fn(eq, *[ns[a] for a in args])

    def _dest_ns(self, dst, t, dt):
        ns = {'t': t, 'dt': dt}
        for k in list(dst.properties) + list(dst.constants):
            ns['d_' + k] = self._arr(dst, k)
        return ns

    def compute(self, t, dt):
        for g in self.groups:
            if getattr(g, 'has_subgroups', False):
                raise NotImplementedError('sub-groups')
            self._group(g, t, dt)

    def _group(self, g, t, dt):
        dests = []
        for eq in g.equations:
            if eq.dest not in dests:
                dests.append(eq.dest)
        for dname in dests:
            dst = self.arrays[self.names.index(dname)]
            eqs = [e for e in g.equations if e.dest == dname]
            n = dst.get_number_of_particles(bool(g.real))
            start = int(g.start_idx or 0)
            stop = n if g.stop_idx is None else min(int(g.stop_idx), n)
            ns = self._dest_ns(dst, t, dt)
            for i in range(start, stop):
                ns['d_idx'] = i
                for e in eqs:
                    self._call(e, 'initialize', ns)
            for i in range(start, stop):
                ns['d_idx'] = i
                for e in eqs:
                    if not e.sources:
                        self._call(e, 'loop', ns)
            sources = []
            for e in eqs:
                for s in (e.sources or []):
                    if s not in sources:
                        sources.append(s)
            for sname in sources:
                src = self.arrays[self.names.index(sname)]
                seqs = [e for e in eqs if e.sources and sname in e.sources]
                if any(getattr(type(e), 'initialize_pair', None) for e in seqs):
                    pns = dict(ns)
                    for k in list(src.properties) + list(src.constants):
                        pns['s_' + k] = self._arr(src, k)
                    for i in range(start, stop):
                        pns['d_idx'] = i
                        for e in seqs:
                            self._call(e, 'initialize_pair', pns)
                # loop_all first, then the pair loop (mako :62-110)
                if any(getattr(type(e), 'loop_all', None) for e in seqs):
                    self._all_nbrs(dst, src, seqs, ns, start, stop)
                if any(getattr(type(e), 'loop', None) for e in seqs):
                    self._pairs(dst, src, seqs, ns, start, stop)
            for i in range(start, stop):
                ns['d_idx'] = i
                for e in eqs:
                    self._call(e, 'post_loop', ns)

    def _all_nbrs(self, dst, src, eqs, ns, start, stop):
        si, di = self.names.index(src.name), self.names.index(dst.name)
        cs, nb = self.nnps.get_csr(si, di)
        pns = dict(ns)
        for k in list(src.properties) + list(src.constants):
            pns['s_' + k] = self._arr(src, k)
        pns['SPH_KERNEL'] = self.kernel
        for i in range(start, stop):
            pns['d_idx'] = i
            lst = [int(j) for j in nb[cs[i]:cs[i + 1]]]
            pns['NBRS'], pns['N_NBRS'] = lst, len(lst)
            for e in eqs:
                self._call(e, 'loop_all', pns)

    def _pairs(self, dst, src, eqs, ns, start, stop):
        K = self.kernel
        si, di = self.names.index(src.name), self.names.index(dst.name)
        cs, nb = self.nnps.get_csr(si, di)
        pns = dict(ns)
        for k in list(src.properties) + list(src.constants):
            pns['s_' + k] = self._arr(src, k)
        dx, dy, dz, dh = dst.x, dst.y, dst.z, dst.h
        sx, sy, sz, sh = src.x, src.y, src.z, src.h
        has = lambda pa, k: k in pa.properties
        for i in range(start, stop):
            pns['d_idx'] = i
            for j in nb[cs[i]:cs[i + 1]]:
                j = int(j)
                pns['s_idx'] = j
                xij = [float(dx[i] - sx[j]), float(dy[i] - sy[j]), float(dz[i] - sz[j])]
                r2 = xij[0] * xij[0] + xij[1] * xij[1] + xij[2] * xij[2]
                rij = math.sqrt(r2)
                hij = 0.5 * (float(dh[i]) + float(sh[j]))
                pns['XIJ'], pns['R2IJ'], pns['RIJ'], pns['HIJ'] = xij, r2, rij, hij
                pns['EPS'] = 0.01 * hij * hij
                if has(dst, 'u') and has(src, 'u'):
                    pns['VIJ'] = [float(dst.u[i] - src.u[j]), float(dst.v[i] - src.v[j]),
                                  float(dst.w[i] - src.w[j])]
                if has(dst, 'rho') and has(src, 'rho'):
                    rhoij = 0.5 * (float(dst.rho[i]) + float(src.rho[j]))
                    pns['RHOIJ'] = rhoij
                    pns['RHOIJ1'] = 1.0 / rhoij if rhoij else float('inf')   # C: 1/0 = inf, no trap
                pns['WIJ'] = K.kernel(xij, rij, hij)
                pns['WI'] = K.kernel(xij, rij, float(dh[i]))
                pns['WJ'] = K.kernel(xij, rij, float(sh[j]))
                pns['GHI'] = float(K.gradient_h(xij, rij, float(dh[i])))
                pns['GHJ'] = float(K.gradient_h(xij, rij, float(sh[j])))
                pns['GHIJ'] = float(K.gradient_h(xij, rij, hij))
                for nm, hh in (('DWIJ', hij), ('DWI', float(dh[i])), ('DWJ', float(sh[j]))):
                    gr = [0.0, 0.0, 0.0]
                    K.gradient(xij, rij, hh, gr)
                    pns[nm] = gr
                for e in eqs:
                    self._call(e, 'loop', pns)


def declare(spec, *a):
    """stand-in for compyle.api.declare inside equation bodies run as Python"""
    spec = spec.replace(' ', '')
    if a and isinstance(a[0], int):         # declare('int', 2) -> two values
        return tuple(0 for _ in range(a[0])) if spec in ('int', 'long') else \
            tuple(0.0 for _ in range(a[0]))
    if spec in ('int', 'long'):
        return 0
    if spec.startswith('matrix('):
        n = 1
        for d in spec[7:-1].strip('()').split(','):
            if d:
                n *= int(d)
        return [0.0] * n
    return 0.0
