"""Ghost generation pinned to the reference's algorithm:
oracle/ghosts.py restates CPUDomainManager._create_ghosts_periodic / _mirror /
_box_wrap_periodic (pysph/base/nnps_base.pyx:506-940) list by list; the host
DomainManager (CPU) and the device HipDomainManager (GPU) must produce the SAME
ghost set -- source particle, image position (bit for bit: the image is x +
translate resp. x + 2 (plane - x), one addition in both), velocity signs -- for
random particle clouds with single, double and triple periodicity / mirrors.
The oracle itself is pinned by the reference's own known answer: a lattice of
n^d points in a periodic box gets (n + 2 l)^d - n^d images
(base/tests/test_domain_manager.py:60-118, test_periodic_nnps.py)."""
import itertools

import numpy as np
import pytest

from pysph_amd.particle_array import get_particle_array_wcsph


def _cloud(n=900, seed=3, lo=0.0, hi=1.0):
    rng = np.random.default_rng(seed)
    x, y, z = (rng.uniform(lo, hi, n) for _ in range(3))
    pa = get_particle_array_wcsph(name='fluid', x=x, y=y, z=z, h=0.05 * np.ones(n),
                                  m=np.arange(n, dtype=float),           # marks the source particle
                                  rho=np.ones(n), u=rng.uniform(-1, 1, n),
                                  v=rng.uniform(-1, 1, n), w=rng.uniform(-1, 1, n))
    return pa


def _expected(pa, kind, axes, n_layers=2.0, radius_scale=2.0):
    from oracle import ghosts as G
    x, y, z = (list(map(float, pa.get(c))) for c in 'xyz')
    lims = [(0.0, 1.0)] * 3
    width = n_layers * radius_scale * 0.05
    if kind == 'periodic':
        gl = G.periodic_ghosts(x, y, z, lims, axes, [1.0, 1.0, 1.0], width)
    else:
        gl = G.mirror_ghosts(x, y, z, lims, axes, width)
    u, v, w = pa.get('u'), pa.get('v'), pa.get('w')
    rows = [(g[0], g[1], g[2], g[3], g[4] * u[g[0]], g[5] * v[g[0]], g[6] * w[g[0]]) for g in gl]
    return np.array(sorted(rows)) if rows else np.zeros((0, 7))


def _got(m, x, y, z, u, v, w, nreal):
    rows = np.stack([m[nreal:], x[nreal:], y[nreal:], z[nreal:], u[nreal:], v[nreal:], w[nreal:]], 1)
    return rows[np.lexsort(rows.T[::-1])] if rows.size else np.zeros((0, 7))


AXES = [(True, False, False), (False, True, False), (True, True, False), (True, False, True),
        (True, True, True)]


def test_oracle_ghost_counts_of_a_lattice():
    """the reference's known answer: (n + 2 l)^d - n^d images"""
    from oracle import ghosts as G
    n, dx = 10, 0.1
    g = (np.arange(n) + 0.5) * dx
    x, y, z = [list(map(float, a.ravel())) for a in np.meshgrid(g, g, g, indexing='ij')]
    width = 2 * dx * 1.01          # two layers of lattice planes
    for axes, d in (((True, False, False), 1), ((True, True, False), 2), ((True, True, True), 3)):
        got = G.periodic_ghosts(x, y, z, [(0.0, 1.0)] * 3, axes, [1.0] * 3, width)
        assert len(got) == (n + 4) ** d * n ** (3 - d) - n ** 3
        # every image is a distinct lattice point outside the box
        pts = set((round(a[1], 9), round(a[2], 9), round(a[3], 9)) for a in got)
        assert len(pts) == len(got)
        got_m = G.mirror_ghosts(x, y, z, [(0.0, 1.0)] * 3, axes, width)
        assert len(got_m) == len(got)


@pytest.mark.parametrize('kind', ['periodic', 'mirror'])
@pytest.mark.parametrize('axes', AXES)
def test_host_domain_manager_ghost_set(kind, axes):
    from pysph_amd.domain import DomainManager
    pa = _cloud()
    exp = _expected(pa, kind, axes)
    kw = dict(xmin=0, xmax=1, ymin=0, ymax=1, zmin=0, zmax=1)
    for ax, name in enumerate('xyz'):
        kw['%s_in_%s' % (kind, name)] = axes[ax]
    dm = DomainManager(**kw)
    dm.set_particles([pa], 2.0)
    nreal = pa.get_number_of_particles(True)
    dm.update()
    get = lambda c: np.asarray(pa.get(c, only_real_particles=False))
    got = _got(get('m'), get('x'), get('y'), get('z'), get('u'), get('v'), get('w'), nreal)
    assert got.shape == exp.shape and got.shape[0] > 0
    assert np.array_equal(got, exp)            # positions bit for bit
    assert np.all(np.asarray(pa.get('tag', only_real_particles=False))[nreal:] != 0)


@pytest.mark.gpu
@pytest.mark.parametrize('kind', ['periodic', 'mirror'])
@pytest.mark.parametrize('axes', AXES)
def test_device_domain_manager_ghost_set(kind, axes):
    from pysph_amd import device as dev
    from pysph_amd.domain import HipDomainManager
    pa = _cloud(seed=11)
    exp = _expected(pa, kind, axes)
    ctx = dev.HipContext(0)
    kw = dict(xmin=0, xmax=1, ymin=0, ymax=1, zmin=0, zmax=1, ctx=ctx)
    for ax, name in enumerate('xyz'):
        kw['%s_in_%s' % (kind, name)] = axes[ax]
    g = dev.attach(pa, ctx)
    g.push()
    dm = HipDomainManager(**kw)
    dm.set_particles([pa], 2.0)
    dm.update()
    n, nreal = g.get_number_of_particles(), g.get_number_of_particles(True)
    cols = {}
    for c in ('m', 'x', 'y', 'z', 'u', 'v', 'w'):
        cols[c] = np.empty(n)
        g.pull_into(c, cols[c])
    got = _got(cols['m'], cols['x'], cols['y'], cols['z'], cols['u'], cols['v'], cols['w'], nreal)
    assert got.shape == exp.shape and got.shape[0] > 0
    assert np.array_equal(got, exp)
    ctx.close()


def test_box_wrap_matches_reference_rule():
    from oracle import ghosts as G
    rng = np.random.default_rng(0)
    x = list(rng.uniform(-0.3, 1.3, 200))
    y = list(rng.uniform(-0.3, 1.3, 200))
    z = list(rng.uniform(0, 1, 200))
    x0, y0 = np.array(x), np.array(y)
    G.box_wrap(x, y, z, [(0.0, 1.0)] * 3, (True, True, False), [1.0] * 3)
    assert np.all((np.array(x) >= 0) & (np.array(x) <= 1))
    assert np.array_equal(np.array(x), np.where(x0 < 0, x0 + 1.0, np.where(x0 > 1, x0 - 1.0, x0)))
    assert np.array_equal(np.array(y), np.where(y0 < 0, y0 + 1.0, np.where(y0 > 1, y0 - 1.0, y0)))
