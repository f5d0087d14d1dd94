"""N>1 path on CPU: world_size-2 gloo run of the slab-halo exchange.

The rank topology, counts handshake, payload exchange and ghost-append logic
of ``pysph_amd.parallel.SlabHalo`` are exercised with a numpy test double for
the device primitives (the product's ``DeviceHaloOps`` are HIP kernels, tested
under -m gpu).  Correctness criterion (mirrors
pysph/parallel/tests/example_test_case.py:143-166): every rank evaluates its
REAL particles with its ghosts as extra sources, and the result must equal the
single-domain evaluation matched by global index -- here BIT-EXACT, because the
oracle walks neighbours in a rank-independent order only up to the sums, so we
compare with the 1e-13 mixed tolerance.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))

PROPS = ('x', 'y', 'z', 'u', 'v', 'w', 'rho', 'h', 'm')


class NumpyHaloOps(object):
    """Test double with the interface of pysph_amd.parallel.DeviceHaloOps."""

    def __init__(self, pa, axis):
        self.pa = pa
        self.axis = 'xyz'[axis]
        self.nprops = len(PROPS)
        self._sel = {}

    def n_real(self):
        return self.pa.get_number_of_particles(True)

    def drop_ghosts(self):
        n = self.n_real()
        if self.pa.get_number_of_particles() != n:
            self.pa.resize(n)

    def select(self, lo_cut, hi_cut):
        c = self.pa.properties[self.axis][:self.n_real()]
        self._sel = {0: np.nonzero(c < lo_cut)[0], 1: np.nonzero(c >= hi_cut)[0]}
        return len(self._sel[0]), len(self._sel[1])

    def new_buffer(self, count):
        return torch.empty(max(count * self.nprops, 1), dtype=torch.float64)

    def int_tensor(self, values):
        return torch.tensor(values, dtype=torch.int64)

    def pack(self, side, count, shift):
        buf = self.new_buffer(count)
        idx = self._sel[side]
        out = buf.numpy()
        for k, p in enumerate(PROPS):
            v = self.pa.properties[p][idx]
            if p == self.axis:
                v = v + shift
            out[k * count:(k + 1) * count] = v
        return buf

    def append(self, buf, count):
        n0 = self.pa.get_number_of_particles()
        nreal = self.n_real()
        self.pa.resize(n0 + count)
        self.pa.set_num_real_particles(nreal)
        arr = buf.numpy()
        for k, p in enumerate(PROPS):
            self.pa.properties[p][n0:] = arr[k * count:(k + 1) * count]
        self.pa.properties['tag'][n0:] = 1  # Remote


def _worker(rank, world, port, periodic, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from test_hip_parity import make_cube, cube_equations
        from pysph_amd import kernels as K
        from pysph_amd.parallel import SlabHalo
        from pysph_amd.particle_array import ParticleArray
        from oracle import oracle as orc
        full, dx = make_cube(14)
        n = full.get_number_of_particles()
        gid = np.arange(n)
        x = full.x
        lo, hi = (0.0, 0.5) if rank == 0 else (0.5, 1.0)
        if periodic:      # make the wrap meaningful: domain [0,1) periodic in x
            pass
        own = np.nonzero((x >= lo) & (x < hi) if rank else (x < hi))[0]
        if rank == world - 1:
            own = np.nonzero(x >= lo)[0]
        pa = ParticleArray(name='fluid', **{k: v[own].copy() for k, v in
                                            full.properties.items()})
        kernel = K.WendlandQuintic(dim=3)
        width = kernel.radius_scale * 1.3 * dx
        halo = SlabHalo(pa, None, rank, world, axis=0, width=width, lo=lo,
                        hi=hi, periodic=periodic, period=1.0,
                        ops=NumpyHaloOps(pa, 0), dist=dist)
        halo.exchange()
        halo.exchange()   # idempotent: old ghosts are dropped first
        nreal = pa.get_number_of_particles(True)
        assert nreal == own.size and pa.get_number_of_particles() > nreal
        eqs = cube_equations(dx)
        nn = orc.OracleNNPS(3, [pa], 2.0)
        nn.update()
        ev = orc.OracleEval([pa], eqs, kernel)
        ev.set_nnps(nn)
        ev.compute(0.0, 1e-5)
        np.savez(out % rank, gid=gid[own],
                 counts=np.array(halo.last_counts),
                 **{k: pa.properties[k][:nreal] for k in
                    ('arho', 'au', 'av', 'aw', 'ax', 'ay', 'az', 'p', 'cs',
                     'dt_cfl', 'dt_force')})
        from pysph_amd.parallel import allreduce_scalars
        mx = allreduce_scalars([float(pa.dt_cfl[:nreal].max())], 'max', dist=dist)
        np.save((out % rank) + '.max.npy', np.array(mx))
    finally:
        dist.destroy_process_group()


def _free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize('periodic', [False])
def test_two_rank_halo_matches_single_domain(tmp_path, oracle, periodic):
    from test_hip_parity import make_cube, cube_equations
    from helpers import rel_err
    from pysph_amd import kernels as K
    out = str(tmp_path / 'rank%d.npz')
    mp.spawn(_worker, args=(2, _free_port(), periodic, out), nprocs=2, join=True)
    full, dx = make_cube(14)
    kernel = K.WendlandQuintic(dim=3)
    nn = oracle.OracleNNPS(3, [full], 2.0)
    nn.update()
    ev = oracle.OracleEval([full], cube_equations(dx), kernel)
    ev.set_nnps(nn)
    ev.compute(0.0, 1e-5)
    seen = 0
    gmax = 0.0
    for r in range(2):
        d = np.load(out % r)
        gid = d['gid']
        seen += gid.size
        assert d['counts'][0] + d['counts'][1] > 0      # something was sent
        for k in ('arho', 'au', 'av', 'aw', 'ax', 'ay', 'az', 'p', 'cs',
                  'dt_cfl', 'dt_force'):
            assert rel_err(d[k], full.properties[k][gid]) < 1e-13, (r, k)
        gmax = float(np.load((out % r) + '.max.npy')[0])
    assert seen == full.get_number_of_particles()
    assert gmax == full.dt_cfl.max()      # all_reduce(MAX) of the dt input


def test_slab_bounds_equal_counts():
    from pysph_amd.parallel import slab_bounds
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(0, 1.228, 9000), rng.uniform(0, 3.22, 1000)])
    cuts = slab_bounds(x, 8)
    counts = np.histogram(x, bins=cuts)[0]
    assert counts.sum() == x.size
    assert counts.max() - counts.min() <= 2


def test_neighbour_topology():
    from pysph_amd.parallel import SlabHalo

    class Dummy(object):
        pass
    for world in (1, 2, 8):
        for rank in range(world):
            h = SlabHalo(None, None, rank, world, 0, 0.1, 0, 1, ops=Dummy(),
                         dist=Dummy())
            peers = [p for _, p, _ in h.neighbours()]
            assert peers == [p for p in (rank - 1, rank + 1) if 0 <= p < world]
            hp = SlabHalo(None, None, rank, world, 0, 0.1, 0, 1, periodic=True,
                          period=float(world), ops=Dummy(), dist=Dummy())
            if world > 1:
                nb = hp.neighbours()
                assert len(nb) == 2
                assert nb[0][1] == (rank - 1) % world
                assert nb[1][1] == (rank + 1) % world
                if rank == 0:
                    assert nb[0][2] == float(world)
                if rank == world - 1:
                    assert nb[1][2] == -float(world)
