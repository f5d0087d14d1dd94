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

    def __init__(self, pa, axis, props=None):
        self.pa = pa
        self.axis = 'xyz'[axis]
        self.all_halo_props = list(props or PROPS)
        self.props = list(self.all_halo_props)   # what a message carries (set_promise may take h / m out)
        self.nprops = len(self.props)
        self.fill = []
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

    def new_buffer(self, count, nprops=None):
        return torch.empty(max(count * (nprops or self.nprops), 1), dtype=torch.float64)

    def int_tensor(self, values):
        return torch.tensor(values, dtype=torch.int64)

    def pack(self, side, count, shift):
        buf = self.new_buffer(count)
        idx = self._sel[side]
        out = buf.numpy()
        for k, p in enumerate(self.props):
            v = self.pa.properties[p][idx]
            if p == self.axis:
                v = v + shift
            out[k * count:(k + 1) * count] = v
        return buf

    def append(self, buf, count, stride=None):
        stride = count if stride is None else stride
        n0 = self.pa.get_number_of_particles()
        nreal = self.n_real()
        self.pa.resize(n0 + count)
        self.pa.set_num_real_particles(nreal)
        arr = buf.numpy()
        for k, p in enumerate(self.props):
            self.pa.properties[p][n0:] = arr[k * stride:k * stride + count]
        for p, v in self.fill:              # promised h / m did not travel
            self.pa.properties[p][n0:] = v
        self.pa.properties['tag'][n0:] = 1  # Remote

    # -- migration ---------------------------------------------------------
    def all_props(self):
        return sorted(k for k, v in self.pa.properties.items()
                      if v.dtype == np.float64)

    def pack_all(self, side, count, shift):
        names = self.all_props()
        buf = self.new_buffer(count, len(names))
        idx = self._sel[side]
        out = buf.numpy()
        for k, p in enumerate(names):
            v = self.pa.properties[p][idx]
            if p == self.axis:
                v = v + shift
            out[k * count:(k + 1) * count] = v
        return buf, len(names)

    def axis_row(self):
        return self.all_props().index(self.axis)

    def remove_selected(self):
        gone = np.concatenate([self._sel[0], self._sel[1]])
        keep = np.setdiff1d(np.arange(self.pa.get_number_of_particles()), gone)
        for k in list(self.pa.properties):
            self.pa.properties[k] = self.pa.properties[k][keep].copy()
        self.pa._n = keep.size
        self.pa.num_real_particles = keep.size
        return keep.size

    def append_real(self, buf, count):
        names = self.all_props()
        n0 = self.pa.get_number_of_particles()
        self.pa.resize(n0 + count)
        self.pa.set_num_real_particles(n0 + count)
        arr = buf.numpy()
        for k, p in enumerate(names):
            self.pa.properties[p][n0:] = arr[k * count:(k + 1) * count]

    def coords(self):
        return self.pa.properties[self.axis][:self.n_real()].copy()


class NumpyDirectHaloOps(NumpyHaloOps):
    """... with the device-side selection + packing of sph_halo_select_pack: the
    host never sees the counts, they travel in the headers of the messages"""

    def select_pack(self, lo_cut, hi_cut, shifts, caps, bufs):
        self.select(lo_cut, hi_cut)
        for s in (0, 1):
            if bufs[s] is None:
                continue
            cap, cnt = int(caps[s]), len(self._sel[s])
            out = bufs[s].numpy()
            if cnt <= cap:
                rows = self.pack(s, cnt, shifts[s]).numpy()
                for k in range(self.nprops):
                    out[k * cap:k * cap + cnt] = rows[k * cnt:(k + 1) * cnt]
            else:                    # what fits, as the device kernel does; the header says "incomplete"
                for k, p in enumerate(self.props):
                    v = self.pa.properties[p][self._sel[s][:cap]]
                    out[k * cap:(k + 1) * cap] = v + (shifts[s] if p == self.axis else 0.0)
            # the sender checks every selected row against a promise that does not travel (+ 0.5 in the header)
            broken = any(np.any(self.pa.properties[p][self._sel[s]] != v) for p, v in self.fill)
            hdr = cnt + (0.5 if broken else 0.0)
            out[self.nprops * cap] = float(hdr if cnt <= cap else -hdr)
        self._sel = {}               # no index lists on the host side of this protocol


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
        lo, hi = rank / float(world), (rank + 1) / float(world)    # equal slabs; the end slabs are open outwards
        own = np.nonzero(((x >= lo) | (rank == 0)) & ((x < hi) | (rank == world - 1)))[0]
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


def _worker_migrate(rank, world, port, out):
    """ownership starts WRONG (cut at x = 0.3 while the slabs are [.,0.5) and
    [0.5,.)) and lopsided; SlabDecomposition.update() must migrate, then
    rebalance() must even out the counts; results equal the single domain."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from test_hip_parity import make_cube, cube_equations
        from pysph_amd import kernels as K
        from pysph_amd.parallel import SlabDecomposition
        from pysph_amd.particle_array import ParticleArray
        from oracle import oracle as orc
        full, dx = make_cube(14)
        n = full.get_number_of_particles()
        x = full.x
        # two ranks: cut at 0.3 for slabs [0, .5) [.5, 1]; three: cuts at 0.2 / 0.5 for thirds -- the middle rank both
        # gives (to its lower neighbour) and takes (from its upper one)
        wrong = {2: [-1e30, 0.3, 1e30], 3: [-1e30, 0.2, 0.5, 1e30]}[world]
        own = np.nonzero((x >= wrong[rank]) & (x < wrong[rank + 1]))[0]
        props = {k: v[own].copy() for k, v in full.properties.items()}
        props['e0'] = own.astype(np.float64)        # global id rides along
        pa = ParticleArray(name='fluid', **props)
        kernel = K.WendlandQuintic(dim=3)
        width = kernel.radius_scale * 1.3 * dx
        lo, hi = rank / float(world), (rank + 1) / float(world)
        dec = SlabDecomposition([pa], None, rank, world, axis=0, width=width,
                                lo=lo, hi=hi,
                                ops_factory=lambda a, ax, p: NumpyHaloOps(a, ax),
                                dist=dist)
        dec.update()
        nreal = pa.get_number_of_particles(True)
        xr = pa.x[:nreal]
        assert ((xr >= lo) | (rank == 0)).all() and ((xr < hi) | (rank == world - 1)).all()
        moved = dec.halos[0].last_migrated
        if rank > 0:
            assert moved[0] > 0             # gave particles to the lower neighbour
        if rank < world - 1:
            assert moved[3] > 0             # ... and took some from the upper one
        # lopsided on purpose -> rebalance moves the faces to the quantiles
        lop = {2: [0.0, 0.8, 1.0], 3: [0.0, 0.6, 0.8, 1.0]}[world]
        dec.halos[0].lo, dec.halos[0].hi = lop[rank], lop[rank + 1]
        dec.update()
        n_before = pa.get_number_of_particles(True)
        faces, rounds = dec.rebalance(nbins=512)
        dec.exchange()
        nreal = pa.get_number_of_particles(True)
        assert abs(nreal - n // world) <= 0.02 * n, (n_before, nreal)
        nn = orc.OracleNNPS(3, [pa], 2.0)
        nn.update()
        ev = orc.OracleEval([pa], cube_equations(dx), kernel)
        ev.set_nnps(nn)
        ev.compute(0.0, 1e-5)
        np.savez(out % rank, gid=pa.e0[:nreal].astype(np.int64), face=faces[1],
                 **{k: pa.properties[k][:nreal] for k in
                    ('arho', 'au', 'av', 'aw', 'ax', 'ay', 'az')})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_two_rank_migration_and_rebalance(tmp_path, oracle, world):
    from test_hip_parity import make_cube, cube_equations
    from helpers import rel_err
    from pysph_amd import kernels as K
    out = str(tmp_path / 'mig%d.npz')
    mp.spawn(_worker_migrate, args=(world, _free_port(), out), nprocs=world, join=True)
    full, dx = make_cube(14)
    kernel = K.WendlandQuintic(dim=3)
    nn = oracle.OracleNNPS(3, [full], 2.0)
    nn.update()
    ev = oracle.OracleEval([full], cube_equations(dx), kernel)
    ev.set_nnps(nn)
    ev.compute(0.0, 1e-5)
    gids = []
    for r in range(world):
        d = np.load(out % r)
        gids.append(d['gid'])
        for k in ('arho', 'au', 'av', 'aw', 'ax', 'ay', 'az'):
            assert rel_err(d[k], full.properties[k][d['gid']]) < 1e-13, (r, k)
    allg = np.sort(np.concatenate(gids))
    assert (allg == np.arange(full.get_number_of_particles())).all()   # nobody lost or duplicated


def _free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize('world', [2, 3])
@pytest.mark.parametrize('periodic', [False, True])
def test_two_rank_halo_matches_single_domain(tmp_path, oracle, periodic, world):
    """(... and with three ranks, where the middle one has two different peers: ranks 1..6 of an 8-GPU run)"""
    from test_hip_parity import make_cube, cube_equations
    from helpers import rel_err
    from pysph_amd import kernels as K
    out = str(tmp_path / 'rank%d.npz')
    mp.spawn(_worker, args=(world, _free_port(), periodic, out), nprocs=world, join=True)
    full, dx = make_cube(14)
    nfull = full.get_number_of_particles()
    kernel = K.WendlandQuintic(dim=3)
    if periodic:
        # single-domain reference: periodic images in x (both faces talk to the
        # SAME peer with two ranks: the message-order case of SlabHalo._swap)
        from pysph_amd.domain import DomainManager
        dom = DomainManager(xmin=0.0, xmax=1.0, periodic_in_x=True, n_layers=1.0)
        dom.set_particles([full], kernel.radius_scale)
        dom.update()
        assert full.get_number_of_particles() > nfull
    nn = oracle.OracleNNPS(3, [full], 2.0)
    nn.update()
    ev = oracle.OracleEval([full], cube_equations(dx), kernel)
    ev.set_nnps(nn)
    ev.compute(0.0, 1e-5)
    seen = 0
    gmax = 0.0
    for r in range(world):
        d = np.load(out % r)
        gid = d['gid']
        seen += gid.size
        assert d['counts'][0] + d['counts'][1] > 0      # something was sent
        for k in ('arho', 'au', 'av', 'aw', 'ax', 'ay', 'az', 'p', 'cs',
                  'dt_cfl', 'dt_force'):
            assert rel_err(d[k], full.properties[k][gid]) < 1e-13, (r, k)
        gmax = float(np.load((out % r) + '.max.npy')[0])
    assert seen == nfull
    assert gmax == full.dt_cfl[:nfull].max()      # all_reduce(MAX) of the dt input


def _worker_cost(rank, world, port, out):
    """rebalance(cost=...): rank 0 reports three times the time per particle of the others -- the faces move so that the
    measured TIME is equal, i.e. rank 0 keeps about a third of the particles per unit of the others' share"""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from test_hip_parity import make_cube
        from pysph_amd.parallel import SlabDecomposition
        from pysph_amd.particle_array import ParticleArray
        full, dx = make_cube(16)
        x = full.x
        lo, hi = rank / float(world), (rank + 1) / float(world)
        own = np.nonzero(((x >= lo) | (rank == 0)) & ((x < hi) | (rank == world - 1)))[0]
        pa = ParticleArray(name='fluid', **{k: v[own].copy() for k, v in full.properties.items()})
        dec = SlabDecomposition([pa], None, rank, world, axis=0, width=2.6 * dx, lo=lo, hi=hi,
                                ops_factory=lambda a, ax, p: NumpyHaloOps(a, ax), dist=dist)
        dec.update()
        n0 = pa.get_number_of_particles(True)
        dec.rebalance(nbins=512, cost=(3.0 if rank == 0 else 1.0) * n0 * 1e-9)
        dec.exchange()
        np.save(out % rank, np.array([n0, pa.get_number_of_particles(True)]))
    finally:
        dist.destroy_process_group()


def test_rebalance_by_measured_time(tmp_path):
    out = str(tmp_path / 'cost%d.npy')
    mp.spawn(_worker_cost, args=(3, _free_port(), out), nprocs=3, join=True)
    n = [np.load(out % r) for r in range(3)]
    total = sum(v[0] for v in n)
    assert sum(v[1] for v in n) == total                      # nobody lost
    # the measured cost belongs to the REGION a rank held: the first third of the cube weighs 3 per particle, the rest 1;
    # thirds of that total (5 n / 9 each) are 5/27, 7/27 and 15/27 of the particles (a lattice plane of 256 either way)
    for got, share in zip(n, (5, 7, 15)):
        assert abs(got[1] - share * total / 27.0) < 300, [v[1] for v in n]


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


def _worker_fused(rank, world, port, periodic, out):
    """SlabDecomposition.exchange() with TWO arrays: one counts handshake and one
    batch of transfers for both; each array must receive exactly the ghosts the
    per-array exchange gives it (same order: lo face first, ascending index)."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from test_hip_parity import make_cube
        from pysph_amd.parallel import SlabDecomposition, SlabHalo
        from pysph_amd.particle_array import ParticleArray
        full, dx = make_cube(12)
        x = full.x
        lo, hi = rank / float(world), (rank + 1) / float(world)
        own = np.nonzero(((x >= lo) | (rank == 0)) & ((x < hi) | (rank == world - 1)))[0]     # (the end slabs are open outwards)
        odd, even = own[own % 2 == 1], own[own % 2 == 0]

        def make(idx, name):
            return ParticleArray(name=name, **{k: v[idx].copy() for k, v in full.properties.items()})
        width = 2.6 * dx
        a, b = make(odd, 'a'), make(even, 'b')
        a2, b2 = make(odd, 'a'), make(even, 'b')
        dec = SlabDecomposition([a, b], None, rank, world, axis=0, width=width, lo=lo, hi=hi,
                                periodic=periodic, period=1.0,
                                ops_factory=lambda pa, ax, p: NumpyHaloOps(pa, ax), dist=dist)
        dec.exchange()
        dec.exchange()           # idempotent
        # the two-phase form (round 4): transports whose device primitives cannot overlap (this numpy stand-in has no
        # streams) complete in the first half; same ghosts either way
        st = dec.exchange_begin()
        assert st is None
        dec.exchange_finish(st)
        flo, fhi = dec.faces()
        if not periodic:
            assert (flo, fhi) == (-float('inf') if rank == 0 else lo, float('inf') if rank == world - 1 else hi)
        else:
            assert (flo, fhi) == (lo, hi)
        for pa, ref in ((a, a2), (b, b2)):
            h = SlabHalo(ref, None, rank, world, axis=0, width=width, lo=lo, hi=hi,
                         periodic=periodic, period=1.0, ops=NumpyHaloOps(ref, 0), dist=dist)
            h.exchange()
            assert pa.get_number_of_particles() == ref.get_number_of_particles() > ref.get_number_of_particles(True)
            for k in PROPS:
                assert np.array_equal(pa.properties[k], ref.properties[k]), (pa.name, k)
        np.save(out % rank, np.array([a.get_number_of_particles(), b.get_number_of_particles()]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
@pytest.mark.parametrize('periodic', [False, True])
def test_fused_two_array_exchange_equals_per_array(tmp_path, periodic, world):
    out = str(tmp_path / 'f%d.npy')
    mp.spawn(_worker_fused, args=(world, _free_port(), periodic, out), nprocs=world, join=True)
    for r in range(world):
        assert np.load(out % r).min() > 0


class NumpyPaddedHaloOps(NumpyDirectHaloOps):
    """... and with the receive side of sph_halo_append_padded: every row of a
    message is appended, the rows behind the ghosts parked at 1e18; counts and the flag
    word are looked at one exchange later"""

    def message_buffer(self, key, size):
        cache = self.__dict__.setdefault('_messages', {})
        t = cache.get(key)
        if t is None or t.numel() != size:
            t = cache[key] = torch.zeros(size, dtype=torch.float64)
        return t

    def flag_words(self, n=1):
        f = self.__dict__.setdefault('_flags', [])
        while len(f) < n:
            f.append(0)
        return f

    def clear_flags(self):
        self._flags = [0] * len(self.__dict__.get('_flags', []))

    def append_padded(self, buf, cap, h_promise, m_promise, flags=None, slot=0):
        flags = self.flag_words(slot + 1) if flags is None else flags
        arr = buf.numpy()
        hdr = arr[self.nprops * cap]
        count = int(min(abs(hdr), cap))
        flags[slot] |= (1 if hdr < 0 else 0)
        n0 = self.pa.get_number_of_particles()
        nreal = self.n_real()
        self.pa.resize(n0 + cap)
        self.pa.set_num_real_particles(nreal)
        for k, p in enumerate(self.props):
            col = np.full(cap, 1e18 if p in 'xyz' else 0.0)
            col[:count] = arr[k * cap:k * cap + count]
            if p == 'h' and h_promise == h_promise and np.any(col[:count] != h_promise):
                flags[slot] |= 2
            if p == 'm' and m_promise == m_promise and np.any(col[:count] != m_promise):
                flags[slot] |= 2
            self.pa.properties[p][n0:] = col
        for p, v in self.fill:          # a promised property that did not travel: written into ALL rows
            self.pa.properties[p][n0:] = v
        self.pa.properties['tag'][n0:] = 1

    def queue_headers(self, tensors, nflags=1):
        return [float(t[-1]) for t in tensors], list(self.flag_words(nflags)[:nflags])

    def collect_headers(self, handle):
        return handle

    def mark_hm_written(self):
        pass

    def set_promise(self, h_promise, m_promise, send=True):
        self.props = [p for p in self.all_halo_props if send or not ((p == 'h' and h_promise == h_promise) or
                                                                     (p == 'm' and m_promise == m_promise))]
        self.nprops = len(self.props)
        self.fill = [(p, v) for p, v in (('h', h_promise), ('m', m_promise)) if v == v and p not in self.props]
        self._messages = {}

    def hm_range(self):
        n = self.n_real()
        if n == 0:
            return [np.inf, -np.inf, np.inf, -np.inf]
        h, m = self.pa.properties['h'][:n], self.pa.properties['m'][:n]
        return [h.min(), h.max(), m.min(), m.max()]


def _worker_padded(rank, world, port, periodic, out):
    """the 'padded' protocol against the counted one: after every exchange the
    first rows behind the real particles are the same ghosts, the rows behind
    them parked far away; the promised h and m do not travel (7 of 9 properties)
    and are written into the rows by the receiver; counts are read by `verify_halos`
    once the evaluation would have been queued and move the capacities on both ends
    alike; a face that outgrew its capacity is REPEATED by verify (the two ranks of
    the face only), unverified it is an error at the next exchange; a row that
    breaks the promise is an error at verify"""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import pysph_amd.parallel as par
        from test_hip_parity import make_cube
        from pysph_amd.particle_array import ParticleArray
        full, dx = make_cube(12)
        x = full.x
        # `world` equal slabs of the unit cube: with three ranks the middle one talks to two DIFFERENT peers (what
        # ranks 1..6 of an 8-GPU run do), with two periodic ranks both faces of a rank talk to the same peer
        lo, hi = rank / float(world), (rank + 1) / float(world)
        own = np.nonzero(((x >= lo) | (rank == 0)) & ((x < hi) | (rank == world - 1)))[0]     # (the end slabs are open outwards)

        def build(protocol, ops):
            pa = ParticleArray(name='fluid', **{k: v[own].copy() for k, v in full.properties.items()})
            h = par.SlabHalo(pa, None, rank, world, axis=0, width=0.1, lo=lo, hi=hi,
                             periodic=periodic, period=1.0, ops=ops(pa, 0), dist=dist, protocol=protocol)
            return pa, h
        pa_c, hc = build('capacity', NumpyDirectHaloOps)
        pa_p, hp = build('padded', NumpyPaddedHaloOps)
        log = []

        def same_ghosts(step):
            nr = pa_c.get_number_of_particles(True)
            nc = pa_c.get_number_of_particles()
            npad = pa_p.get_number_of_particles()
            assert npad >= nc
            # every face's message is appended whole: its ghosts, then its padding rows
            live = np.abs(pa_p.properties['x'][:npad]) < 1e17
            assert live[:nr].all() and live.sum() == nc
            for k in PROPS:
                assert np.array_equal(pa_c.properties[k][:nc], pa_p.properties[k][:npad][live]), (step, k)
                parked = 1e18 if k in 'xyz' else (hp.h_promise if k == 'h' else hp.m_promise if k == 'm' else 0.0)
                assert np.all(pa_p.properties[k][:npad][~live] == parked), (step, k)
            return npad - nc
        for step, width in enumerate((0.1, 0.1, 0.12, 0.05, 0.15, 0.15)):
            hc.width = hp.width = width
            hc.exchange()
            hp.exchange()
            assert par.verify_halos([hp])            # (the evaluation would be queued before this)
            log.append((hp.padded_exchanges, same_ghosts(step)))
        assert hp.handshakes == 1 and hp.padded_exchanges == 5 and hp.repaired_exchanges == 0
        assert hp.h_promise == hp.h_promise and hp.m_promise == hp.m_promise     # the cube has one h and one m
        assert hp.ops.props == [p for p in PROPS if p not in ('h', 'm')]         # ... which do not travel
        assert hp.last_counts == hc.last_counts
        # a face that outgrows its capacity in ONE exchange: verify repeats it the counted way -- same ghosts after
        hp.cap_send = {s: 8 for s in hp.cap_send}
        hp.cap_recv = {s: 8 for s in hp.cap_recv}
        hc.exchange()
        hp.exchange()
        assert not par.verify_halos([hp])
        assert hp.repaired_exchanges == 1
        same_ghosts('repaired')
        assert par.verify_halos([hp])                # nothing outstanding: nothing to do
        hc.exchange()
        hp.exchange()                                # capacities followed the counts: complete again
        assert par.verify_halos([hp])
        same_ghosts('after repair')
        # ... unverified: loud at the next exchange
        hp.cap_send = {s: 8 for s in hp.cap_send}
        hp.cap_recv = {s: 8 for s in hp.cap_recv}
        hp.exchange()
        try:
            hp.exchange()
            raised = False
        except par.GhostsIncomplete as e:
            raised = 'outgrew its message capacity' in str(e)
        assert raised
        np.save(out % rank, np.array(log))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3, 4])
@pytest.mark.parametrize('periodic', [False, True])
def test_padded_protocol_equals_capacity_protocol(tmp_path, periodic, world):
    out = str(tmp_path / 'padded_%d.npy')
    mp.spawn(_worker_padded, args=(world, _free_port(), periodic, out), nprocs=world, join=True)
    for rank in range(world):
        a = np.load(out % rank)
        assert a.shape == (6, 2) and a[-1, 0] == 5 and a[1:, 1].min() > 0


def _worker_promises(rank, world, port, send_promised, out):
    """TWO arrays through one padded exchange: the flag words are per array (the
    round-5 exchange looked at the first array's word only); a row of the SECOND
    array that breaks its promise is found -- by the receiver's check when the
    promised properties travel, by the sender's when they do not -- and verify
    names the array; renew_promises (collective) repairs the contract"""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import pysph_amd.parallel as par
        from test_hip_parity import make_cube
        from pysph_amd.particle_array import ParticleArray
        full, dx = make_cube(12)
        x = full.x
        lo, hi = rank / float(world), (rank + 1) / float(world)
        own = np.nonzero(((x >= lo) | (rank == 0)) & ((x < hi) | (rank == world - 1)))[0]
        odd, even = own[own % 2 == 1], own[own % 2 == 0]

        def make(idx, name):
            return ParticleArray(name=name, **{k: v[idx].copy() for k, v in full.properties.items()})
        a, b = make(odd, 'a'), make(even, 'b')
        dec = par.SlabDecomposition([a, b], None, rank, world, axis=0, width=2.6 * dx, lo=lo, hi=hi,
                                    ops_factory=lambda pa, ax, p: NumpyPaddedHaloOps(pa, ax), dist=dist,
                                    protocol='padded', send_promised=send_promised)
        for _ in range(3):
            dec.exchange()
            assert dec.verify()
        assert dec.halos[1].padded_exchanges == 2
        assert (len(dec.halos[1].ops.props) == 9) == send_promised
        # rank 0 writes h of ONE particle of array b next to its upper face
        if rank == 0:
            nr = b.get_number_of_particles(True)
            i = int(np.argmax(b.x[:nr]))
            b.h[i] *= 1.5
        dec.exchange()
        seen = ''
        try:
            dec.verify()
        except RuntimeError as e:
            seen = str(e)
        if send_promised:
            # the receiver's check: rank 1 finds it in the word of array 1
            assert ("['b']" in seen) == (rank == 1), (rank, seen)
        else:
            # the sender's check travels in the header: both ends of the face see it
            assert ("['b']" in seen) == (rank in (0, 1)), (rank, seen)
        # every rank looks at its ranges again: no promise for b's h any more, its h travels, the exchange goes on
        dec.renew_promises()
        for _ in range(3):
            dec.exchange()
            assert dec.verify()
        hb = dec.halos[1]
        assert hb.h_promise != hb.h_promise and hb.m_promise == hb.m_promise
        assert 'h' in hb.ops.props and (('m' in hb.ops.props) == send_promised)
        if rank == 1:
            ng = b.get_number_of_particles()
            assert np.sum(b.h[:ng] == 1.5 * full.h[0]) == 1      # the odd one arrived as a ghost with ITS h
        np.save(out % rank, np.array([dec.halos[0].padded_exchanges, dec.halos[1].padded_exchanges]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('send_promised', [False, True])
def test_padded_promises_are_checked_per_array(tmp_path, send_promised):
    out = str(tmp_path / 'prom_%d.npy')
    mp.spawn(_worker_promises, args=(3, _free_port(), send_promised, out), nprocs=3, join=True)
    for rank in range(3):
        assert np.load(out % rank).min() >= 4


def _worker_protocols(rank, world, port, periodic, out):
    """the fixed-capacity ghost messages against the counts handshake: same ghost
    rows after every exchange while the slab faces (and so the counts) move;
    tiny capacities force the repeat-with-exact-size path on every face"""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import pysph_amd.parallel as par
        from test_hip_parity import make_cube
        from pysph_amd.particle_array import ParticleArray
        full, dx = make_cube(12)
        x = full.x
        lo, hi = rank / float(world), (rank + 1) / float(world)     # equal slabs (three ranks: two different peers)
        own = np.nonzero(((x >= lo) | (rank == 0)) & ((x < hi) | (rank == world - 1)))[0]     # (the end slabs are open outwards)

        def build(protocol, ops=NumpyHaloOps):
            pa = ParticleArray(name='fluid', **{k: v[own].copy() for k, v in full.properties.items()})
            h = par.SlabHalo(pa, None, rank, world, axis=0, width=0.1, lo=lo, hi=hi,
                             periodic=periodic, period=1.0, ops=ops(pa, 0), dist=dist)
            h.protocol = protocol
            return pa, h
        pa_h, hh = build('handshake')
        pa_c, hc = build('capacity')
        pa_o, ho = build('capacity')
        # the same two with selection + packing "on the device" (counts only in the headers)
        pa_d, hd = build('capacity', NumpyDirectHaloOps)
        pa_e, he = build('capacity', NumpyDirectHaloOps)
        log = []
        for step, width in enumerate((0.1, 0.1, 0.22, 0.05, 0.3, 0.3)):
            for h in (hh, hc, ho, hd, he):
                h.width = width
            hh.exchange()
            hc.exchange()
            hd.exchange()
            if step:                 # capacities of 8 rows: every face overflows
                for h in (ho, he):
                    h.cap_send = {s: 8 for s in h.cap_send}
                    h.cap_recv = {s: 8 for s in h.cap_recv}
            ho.exchange()
            he.exchange()
            n = pa_h.get_number_of_particles()
            for pa in (pa_c, pa_o, pa_d, pa_e):
                assert pa.get_number_of_particles() == n
                for k in PROPS:
                    assert np.array_equal(pa_h.properties[k][:n], pa.properties[k][:n]), (step, k)
            assert hh.last_counts == hc.last_counts == ho.last_counts == hd.last_counts == he.last_counts
            log.append(hc.last_counts)
        # the handshake ran once (first exchange) under 'capacity', every time under 'handshake'
        assert hc.handshakes == 1 and hh.handshakes == 6 and ho.handshakes == 1
        # capacities follow the counts on both ends without a word being exchanged
        for s, c in hc.cap_send.items():
            assert c >= hc.last_counts[s] and c % 1024 == 0
        np.save(out % rank, np.array(log))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
@pytest.mark.parametrize('periodic', [False, True])
def test_capacity_protocol_equals_handshake(tmp_path, periodic, world):
    out = str(tmp_path / 'proto_%d.npy')
    mp.spawn(_worker_protocols, args=(world, _free_port(), periodic, out), nprocs=world, join=True)
    logs = [np.load(out % r) for r in range(world)]      # per exchange: (sent lo, sent hi, received lo, received hi)
    assert logs[0].shape == (6, 4) and logs[0][:, :2].sum() > 0
    for r in range(world):
        up = (r + 1) % world
        if r + 1 < world or periodic:
            # what rank r sent through its hi face arrived at its upper neighbour's lo face, and back
            assert np.array_equal(logs[r][:, 1], logs[up][:, 2]) and np.array_equal(logs[up][:, 0], logs[r][:, 3])


def test_capacity_rule_is_symmetric_and_stable():
    """both ends of a face apply _next_capacity to the SAME (capacity, count) pair:
    it must be a pure function, grow before the count reaches the capacity, not
    oscillate on a steady count and shrink after a large drop"""
    from pysph_amd.parallel import _capacity, _next_capacity
    for count in (0, 1, 4095, 4096, 75843, 10 ** 6, 3 * 10 ** 7):
        cap = _capacity(count)
        assert cap % 1024 == 0 and cap >= count + count // 4 + 4096
        assert _next_capacity(None, count) == cap
        assert _next_capacity(cap, count) == cap                 # steady state: no change
        assert _next_capacity(cap, count + count // 16) == cap   # small growth fits the headroom
        assert _next_capacity(cap, cap + 1) == _capacity(cap + 1)            # overflow -> resized
    big = _capacity(10 ** 6)
    assert _next_capacity(big, 1000) == _capacity(1000)          # a large drop shrinks the messages
    # a slowly growing count is re-sized BEFORE it overflows
    cap, resized, overflowed = _capacity(50000), 0, 0
    for count in range(50000, 200000, 500):
        overflowed += count > cap
        new = _next_capacity(cap, count)
        resized += new != cap
        cap = new
    assert overflowed == 0 and 0 < resized < 12


def test_capacity_adapts_to_a_steady_count():
    """with a history both ends of a face keep (fed the same counts) the messages
    SHRINK once the count has been steady: 25 % + 4096 rows of headroom -> 1/16 + 1024;
    a count that comes close to the tight capacity returns the face to the generous
    rule before it overflows; two ends fed the same sequence stay in step"""
    from pysph_amd.parallel import STEADY_EXCHANGES, _capacity, _capacity_tight, _next_capacity
    rng = np.random.default_rng(3)
    seq = [115000 + int(rng.integers(-300, 300)) for _ in range(30)]             # a dam-break face at rest
    seq += [115000 + 400 * k for k in range(1, 40)]                               # the front arrives
    seq += [131000] * 20
    ends = []
    for _ in range(2):
        cap, hist, caps = None, [None, 0, False], []
        for c in seq:
            cap = _next_capacity(cap, c, hist)
            caps.append(cap)
        ends.append(caps)
    assert ends[0] == ends[1]
    caps = ends[0]
    assert caps[0] == _capacity(seq[0])
    assert caps[STEADY_EXCHANGES + 1] == _capacity_tight(seq[STEADY_EXCHANGES + 1]) < caps[0]
    assert all(c >= n for c, n in zip([caps[0]] + caps[:-1], seq))               # the capacity in force always held the count
    assert caps[29] * 8 * 7 < 0.8 * caps[0] * 8 * 9                               # tight AND without h, m: < 80 % of the bytes... of round 5
    assert caps[-1] == _capacity_tight(131000)                                    # steady again: tight again
    # without a history: the rule of the counted protocol, unchanged
    assert _next_capacity(None, 1000) == _capacity(1000)


def test_null_stream_handle_is_not_a_shared_stream():
    """HipContext(dev, torch.cuda.current_stream().cuda_stream) on torch's
    default stream passes the NULL handle; the library then creates a stream of
    its own, so the halo's pack / append kernels and the transport do NOT share a
    stream and the two host synchronisations must stay (a skipped one lets RCCL
    send a payload the pack kernel has not finished)."""
    import ctypes
    import types
    from pysph_amd.parallel import DeviceHaloOps

    def fake(handle, current):
        ops = types.SimpleNamespace()
        ops.ctx = types.SimpleNamespace(stream=handle)
        ops.device = None
        ops.torch = types.SimpleNamespace(cuda=types.SimpleNamespace(
            current_stream=lambda d: types.SimpleNamespace(cuda_stream=current)))
        return ops
    shares = DeviceHaloOps._shares_torch_stream
    assert not shares(fake(None, 0))
    assert not shares(fake(0, 0))                       # null handle: library-owned stream
    assert not shares(fake(ctypes.c_void_p(0), 0))
    assert shares(fake(0x7f00, 0x7f00))
    assert shares(fake(ctypes.c_void_p(0x7f00), 0x7f00))
    assert not shares(fake(0x7f00, 0x7f10))
