"""LinkedListNNPS scenarios of the reference's pysph/base/tests/test_nnps.py
that are not data-parallel bulk cases (those: test_hip_parity.py): cell index
positivity, the 2^28-cell limit, neighbour lists sorted by gid, thousands of
neighbours in ONE cell, a two-particle corner case with radius_scale 0.7, and
1-D data searched with dim = 2 / 3.  CPU: the oracle; GPU: HipNNPS, each
against brute force / the values the reference asserts."""
import numpy as np
import pytest

from pysph_amd.particle_array import get_particle_array


def brute_force(pa, i, radius_scale):
    x, y, z, h = pa.x, pa.y, pa.z, pa.h
    r2 = (x - x[i]) ** 2 + (y - y[i]) ** 2 + (z - z[i]) ** 2
    keep = (r2 < (radius_scale * h[i]) ** 2) | (r2 < (radius_scale * h) ** 2)
    return np.nonzero(keep)[0]


def far_away_particles(nx=20):
    """test_nnps.py:1003-1017: a unit cube of particles plus one at 1000"""
    rng = np.random.default_rng(3)
    x, y, z = rng.random((3, nx ** 3))
    h = np.ones_like(x) * 1.3 / nx
    return get_particle_array(name='fluid', x=np.append(x, 1000.0), y=np.append(y, 1000.0),
                              z=np.append(z, 1000.0), h=np.append(h, h[0]))


def line_particles(nx=10):
    x = np.linspace(0, 1, nx)
    return get_particle_array(name='fluid', x=x, h=np.ones_like(x) / (nx - 1))


def one_cell_particles(n):
    x = np.random.default_rng(5).random(n) * 0.1
    return get_particle_array(name='fluid', x=x, y=x.copy(), z=x.copy(), h=np.ones_like(x))


def corner_pair():
    return get_particle_array(name='fluid', x=[0.131, 0.359], y=[1.544, 1.809],
                              z=[-3.6489999, -2.8559999], h=1.0)


def y_line():
    y = np.array([1.0, 1.5])
    return get_particle_array(name='fluid', y=y, h=np.ones_like(y))


# ---------------------------------------------------------------------------
# CPU: the oracle
# ---------------------------------------------------------------------------
def test_oracle_cell_index_positivity(oracle):              # :985-999
    rng = np.random.default_rng(0)
    pa = get_particle_array(name='a', x=rng.uniform(-1, 1, 100), y=rng.uniform(-1, 1, 100),
                            z=rng.uniform(-1, 1, 100), h=0.2 * np.ones(100))
    nn = oracle.OracleNNPS(3, [pa], radius_scale=2.0)
    nn.update()
    for cell in range(int(nn.n_cells)):
        assert min(oracle.unflatten(cell, nn.ncells_per_dim, 3)) > -1


def test_oracle_raises_for_large_domain(oracle):             # :1019-1026
    nn = oracle.OracleNNPS(3, [far_away_particles(8)], radius_scale=2.0)
    with pytest.raises(RuntimeError):
        nn.update()


def test_oracle_corner_cases(oracle):                        # :1285-1297, :1327-1346
    pa = corner_pair()
    nn = oracle.OracleNNPS(3, [pa], radius_scale=0.7)
    nn.update()
    for i in range(2):
        assert sorted(nn.get_nearest_particles(0, 0, i)) == sorted(brute_force(pa, i, 0.7))
    for dim in (2, 3):
        nn = oracle.OracleNNPS(dim, [y_line()], radius_scale=2.0)
        nn.update()
        assert len(nn.get_nearest_particles(0, 0, 0)) == 2


def test_oracle_many_neighbours_in_one_cell(oracle):         # :1236-1247
    pa = one_cell_particles(1 << 12)
    nn = oracle.OracleNNPS(3, [pa], radius_scale=2.0)
    nn.update()
    assert len(nn.get_nearest_particles(0, 0, 0)) == pa.x.size


# ---------------------------------------------------------------------------
# GPU
# ---------------------------------------------------------------------------
def hip_nnps(arrays, dim, radius_scale=2.0, **kw):
    from pysph_amd import device as dev
    from pysph_amd.nnps import HipNNPS
    return HipNNPS(dim, arrays, radius_scale=radius_scale, ctx=dev.HipContext(0), **kw)


@pytest.mark.gpu
def test_raises_for_large_domain():                          # :1019-1026
    with pytest.raises(RuntimeError):
        hip_nnps([far_away_particles(20)], 3, cache=True)


@pytest.mark.gpu
def test_cell_index_positivity_and_bounds(oracle):           # :985-999
    rng = np.random.default_rng(0)
    pa = get_particle_array(name='a', x=rng.uniform(-1, 1, 100), y=rng.uniform(-1, 1, 100),
                            z=rng.uniform(-1, 1, 100), h=0.2 * np.ones(100))
    nn = hip_nnps([pa], 3)
    on = oracle.OracleNNPS(3, [pa], radius_scale=2.0)
    on.update()
    assert list(nn.ncells_per_dim) == list(on.ncells_per_dim) and nn.n_cells == on.n_cells
    assert nn.cell_size == on.cell_size
    assert np.array_equal(nn.xmin, on.xmin) and np.array_equal(nn.xmax, on.xmax)


@pytest.mark.gpu
def test_sorts_neighbours_by_gid():                          # :1163-1199
    pa = line_particles(10)
    nn = hip_nnps([pa], 1, sort_gids=True)
    assert pa.gid.max() == pa.gid.min() and pa.gid.max() > pa.gid.size      # invalid gids
    for i in range(10):
        nb = nn.get_nearest_particles(0, 0, i)
        assert list(nb) == sorted(nb) and len(nb) == len(brute_force(pa, i, 2.0))
    # valid gids in reverse order: the lists come back ordered by gid
    pa.gid[:] = np.arange(pa.x.size)[::-1]
    nn.update()
    for i in range(10):
        nb = nn.get_nearest_particles(0, 0, i)
        assert list(pa.gid[nb]) == sorted(pa.gid[nb])
        assert sorted(nb) == list(brute_force(pa, i, 2.0))


@pytest.mark.gpu
def test_large_number_of_neighbours_in_one_cell(oracle):     # :1236-1247
    """every particle sees all 4096 others through a single cell: the pair
    kernel walks one row in many 480-candidate tiles; SummationDensity must
    still equal the oracle's"""
    from pysph_amd import device as dev
    from pysph_amd import kernels as K
    from pysph_amd.acceleration_eval import AccelerationEval, SPHCompiler
    from pysph_amd.equations import Group, SummationDensity
    from pysph_amd.nnps import HipNNPS
    pa, ref = one_cell_particles(1 << 12), one_cell_particles(1 << 12)
    for a in (pa, ref):
        a.m[:] = 1.0 / a.x.size
    kernel = K.CubicSpline(dim=3)
    eqs = [Group(equations=[SummationDensity(dest='fluid', sources=['fluid'])])]
    ctx = dev.HipContext(0)
    a_eval = AccelerationEval([pa], eqs, kernel)
    SPHCompiler(a_eval, ctx=ctx).compile()
    nn = HipNNPS(3, [pa], radius_scale=2.0, ctx=ctx)
    a_eval.set_nnps(nn)
    assert nn.count_neighbors(0, 0) == pa.x.size ** 2
    assert len(nn.get_nearest_particles(0, 0, 0)) == pa.x.size
    a_eval.compute(0.0, 0.1)
    on = oracle.OracleNNPS(3, [ref], radius_scale=2.0)
    on.update()
    oev = oracle.OracleEval([ref], eqs, kernel, nthreads=8)
    oev.set_nnps(on)
    oev.compute(0.0, 0.1)
    assert np.max(np.abs(pa.rho - ref.rho) / np.abs(ref.rho)) < 1e-12


@pytest.mark.gpu
def test_corner_case_few_cells_and_lower_dimensional_data():  # :1285-1297, :1327-1346
    pa = corner_pair()
    nn = hip_nnps([pa], 3, radius_scale=0.7)
    for i in range(2):
        assert sorted(nn.get_nearest_particles(0, 0, i)) == sorted(brute_force(pa, i, 0.7))
    for dim in (2, 3):
        nn = hip_nnps([y_line()], dim, cache=False)
        assert len(nn.get_nearest_particles(0, 0, 0)) == 2
