"""Known-answer tests of the smoothing kernels, after the reference's
pysph/base/tests/test_kernel.py: value at the origin / outside the support,
and the moment identities  int W = 1, int x W = 0, int dW = 0,
int x dW/dx = -1.

CPU: the host kernel classes (1-D by quadrature, 2-D/3-D by the same lattice
sums the reference uses).  GPU: the same lattice sums evaluated by the device
pair loop itself -- a probe particle at the centre of a unit lattice is the
destination of a generated equation accumulating  fac * WIJ  and  fac * DWIJ
over the lattice -- so the device kernel functions and their gradients are
pinned to analytic values, not only to the oracle."""
import numpy as np
import pytest

from pysph_amd import kernels as K
from pysph_amd.equations import Equation, Group

# (kernel, dim) -> W(0) at h = 1 (test_kernel.py:152,191,229,271,294,316,343,350,357,435,446)
W0 = {('CubicSpline', 1): 2.0 / 3, ('CubicSpline', 2): 10.0 / (7 * np.pi),
      ('CubicSpline', 3): 1.0 / np.pi,
      ('Gaussian', 1): 1.0 / np.sqrt(np.pi), ('Gaussian', 2): 1.0 / np.pi,
      ('Gaussian', 3): 1.0 / np.pi ** 1.5,
      ('QuinticSpline', 1): 0.55, ('QuinticSpline', 2): 66.0 * 7.0 / (478.0 * np.pi),
      ('QuinticSpline', 3): 66.0 / (120.0 * np.pi),
      ('WendlandQuintic', 2): 7.0 / (4.0 * np.pi),
      ('WendlandQuintic', 3): 21.0 / (16.0 * np.pi)}

# places of agreement the reference asks of the lattice sums
# (zeroth moment, first grad moment): test_kernel.py:195,212,233,251,298,302,320,324,439
PLACES = {('CubicSpline', 2): (7, 6), ('CubicSpline', 3): (6, 4),
          ('Gaussian', 2): (3, 2), ('Gaussian', 3): (2, 2),
          ('QuinticSpline', 2): (7, 6), ('QuinticSpline', 3): (2, 2),
          ('WendlandQuintic', 2): (6, 6), ('WendlandQuintic', 3): (2, 2)}


def _w(k, x, xj, h):
    return float(k.kernel(rij=abs(x - xj), h=h))


def _dw(k, x, xj, h):
    """d/dx of W(|x - xj|, h) -- gradient()[0] of the reference wrappers"""
    r = abs(x - xj)
    if r < 1e-12:
        return 0.0
    return float(k.dwdq(rij=r, h=h)) / h * (x - xj) / r


@pytest.mark.parametrize('name,dim', sorted(W0))
def test_kernel_at_origin_and_outside(name, dim):
    k = getattr(K, name)(dim=dim)
    assert abs(float(k.kernel(rij=0.0, h=1.0)) - W0[(name, dim)]) < 1e-7
    if name != 'Gaussian':                      # the reference cuts the Gaussian at 3h too
        assert float(k.kernel(rij=3.0, h=1.0)) == 0.0
    assert abs(float(k.kernel(rij=3.0, h=1.0))) < 1e-7
    assert float(k.dwdq(rij=3.0, h=1.0)) == 0.0 or name == 'Gaussian'


@pytest.mark.parametrize('name', ['CubicSpline', 'QuinticSpline', 'Gaussian'])
def test_1d_moments_by_quadrature(name):
    """test_kernel.py:154-182 (Gaussian: :273-287, looser because of the cut-off)"""
    from scipy.integrate import quad
    k = getattr(K, name)(dim=1)
    kh = k.radius_scale
    places = 8 if name != 'Gaussian' else 3
    tol = 0.5 * 10.0 ** -places

    def q(f, a, b, pts):
        return quad(f, a, b, points=pts, limit=200)[0]

    for a, b, h, xj in ((-kh, kh, 1.0, 0.0), (-kh, kh, 0.5, 0.0), (0.0, 2 * kh, 1.0, kh)):
        pts = [xj + s * h for s in (-3, -2, -1, 0, 1, 2, 3) if a < xj + s * h < b]
        assert abs(q(lambda x: _w(k, x, xj, h), a, b, pts) - 1.0) < tol
        assert abs(q(lambda x: _dw(k, x, xj, h), a, b, pts)) < tol
    pts = [-2.0, -1.0, 0.0, 1.0, 2.0]
    assert abs(q(lambda x: x * _w(k, x, 0.0, 1.0), -kh, kh, pts)) < tol
    pts = [kh + s for s in (-2, -1, 0, 1, 2)]
    assert abs(q(lambda x: (x - kh) * _dw(k, x, kh, 1.0), 0.0, 2 * kh, pts) + 1.0) < 2 * tol


def _lattice(dim):
    n = 101 if dim == 2 else 51
    ax = np.linspace(0.0, 1.0, n)
    if dim == 2:
        x, y = np.meshgrid(ax, ax, indexing='ij')
        z = np.zeros_like(x)
        vol = (1.0 / (n - 1)) ** 2
    else:
        x, y, z = np.meshgrid(ax, ax, ax, indexing='ij')
        vol = (1.0 / (n - 1)) ** 3
    return x.ravel(), y.ravel(), z.ravel(), vol


@pytest.mark.parametrize('name,dim', sorted(PLACES))
def test_lattice_moments_host_kernels(name, dim):
    """test_kernel.py:62-110 with the host kernel classes (vectorised)"""
    k = getattr(K, name)(dim=dim)
    x, y, z, vol = _lattice(dim)
    c = 0.5
    dx, dy, dz = x - c, y - c, (z - c if dim == 3 else z)
    r = np.sqrt(dx * dx + dy * dy + dz * dz)
    h = 0.15
    w = k.kernel(rij=r, h=h)
    p0, p1 = PLACES[(name, dim)]
    assert abs(np.sum(w) * vol - 1.0) < 0.5 * 10.0 ** -p0
    assert abs(np.sum(dx * w) * vol) < 0.5e-7
    g = np.where(r > 1e-12, k.dwdq(rij=r, h=h) / h / np.maximum(r, 1e-300), 0.0)
    assert abs(np.sum(g * dx) * vol) < 0.5e-7
    assert abs(np.sum(dx * g * dx) * vol + 1.0) < 0.5 * 10.0 ** -p1
    assert abs(np.sum(dy * g * dx) * vol) < 0.5e-8


class KernelMoments(Equation):
    """sum_j fac_j * W_ij * m_j  and  sum_j fac_j * grad W_ij * m_j  with
    fac = (x_j - x_i)^l (y_j - y_i)^m (z_j - z_i)^n;  the gradient is taken at
    the lattice point, as test_kernel.py does: grad_j W = -DWIJ."""

    def __init__(self, dest, sources, l=0, m=0, n=0):
        self.l = float(l)
        self.m = float(m)
        self.n = float(n)
        super(KernelMoments, self).__init__(dest, sources)

    def initialize(self, d_idx, d_mom, d_gx, d_gy, d_gz):
        d_mom[d_idx] = 0.0
        d_gx[d_idx] = 0.0
        d_gy[d_idx] = 0.0
        d_gz[d_idx] = 0.0

    def loop(self, d_idx, s_idx, d_mom, d_gx, d_gy, d_gz, s_m, XIJ, WIJ, DWIJ):
        fac = pow(-XIJ[0], self.l) * pow(-XIJ[1], self.m) * pow(-XIJ[2], self.n) * s_m[s_idx]
        d_mom[d_idx] += fac * WIJ
        d_gx[d_idx] -= fac * DWIJ[0]
        d_gy[d_idx] -= fac * DWIJ[1]
        d_gz[d_idx] -= fac * DWIJ[2]


def moment_arrays(dim):
    from pysph_amd.particle_array import get_particle_array
    x, y, z, vol = _lattice(dim)
    lat = get_particle_array(name='lattice', x=x, y=y, z=z, h=np.full(x.size, 0.15),
                             m=np.full(x.size, vol))
    c = 0.5
    probe = get_particle_array(name='probe', x=[c], y=[c], z=[c if dim == 3 else 0.0],
                               h=[0.15], m=[vol])
    for p in ('mom', 'gx', 'gy', 'gz'):
        probe.add_property(p)
    return probe, lat


def moment_equations(l=0, m=0, n=0):
    return [Group(equations=[KernelMoments('probe', ['lattice'], l=l, m=m, n=n)])]


@pytest.mark.gpu
@pytest.mark.parametrize('name,dim', sorted(PLACES))
def test_lattice_moments_device_pair_loop(name, dim):
    from pysph_amd import device as dev
    from pysph_amd.acceleration_eval import AccelerationEval, SPHCompiler
    from pysph_amd.nnps import HipNNPS
    kernel = getattr(K, name)(dim=dim)
    p0, p1 = PLACES[(name, dim)]
    probe, lat = moment_arrays(dim)
    expect = {}
    for lmn in ((0, 0, 0), (1, 0, 0), (0, 1, 0)) + (((0, 0, 1),) if dim == 3 else ()):
        ctx = dev.HipContext(0)
        a_eval = AccelerationEval([probe, lat], moment_equations(*lmn), kernel)
        SPHCompiler(a_eval, ctx=ctx).compile()
        nnps = HipNNPS(dim, [probe, lat], radius_scale=kernel.radius_scale, ctx=ctx)
        a_eval.set_nnps(nnps)
        a_eval.compute(0.0, 0.1)
        expect[lmn] = (probe.mom[0], probe.gx[0], probe.gy[0], probe.gz[0])
        ctx.close()
    mom, gx, gy, gz = expect[(0, 0, 0)]
    assert abs(mom - 1.0) < 0.5 * 10.0 ** -p0
    assert max(abs(gx), abs(gy), abs(gz)) < 0.5e-7
    for axis, lmn in enumerate(((1, 0, 0), (0, 1, 0), (0, 0, 1))[:dim]):
        mom, *g = expect[lmn]
        assert abs(mom) < 0.5e-7                       # first kernel moment
        for other in range(3):
            if other == axis:
                assert abs(g[other] + 1.0) < 0.5 * 10.0 ** -p1, (lmn, g)
            else:
                assert abs(g[other]) < 0.5e-6, (lmn, g)
