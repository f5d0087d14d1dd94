"""EIGHT ranks before a node exists (CPU, gloo): the workloads `bench.py --gpus 8`
runs are built by bench.py's own `build_workload` for every rank -- the weak-scaling
cube (eight unit cubes side by side), BASELINE config 4's strong-scaling dam break
(ONE three-array tank cut into eight equal-count slabs), the elastic block -- and
go through `SlabDecomposition` on the round-trip-free ('padded') protocol that
`bench.py --gpus N` defaults to, with a numpy test double for the device
primitives: rendezvous of eight processes, the counted first exchange, promises
all-reduced over eight ranks, capacities following the counts on both ends of
seven faces, ranks 1..6 talking to two different peers, verify() after every
exchange; then every rank evaluates its real particles with the oracle and the
result equals the whole domain evaluated alone, matched by gid (the reference's
recipe: pysph/parallel/tests/example_test_case.py:143-166).  What is NOT covered
here is RCCL itself and the device kernels (GPU tests, world size 1 and two
thread-ranks)."""
import os
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))

WORLD = 8
ARGV = {
    'cube': ['--workload', 'cube', '--n1', '8'],
    'dam_break': ['--workload', 'dam_break', '--dx', '0.04'],
    'elastic_block': ['--workload', 'elastic_block', '--n1', '32'],
}


def _worker(rank, world, port, workload, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ['OMP_NUM_THREADS'] = '1'
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import bench
        import pysph_amd.parallel as par
        from oracle import oracle as orc
        from test_parallel_gloo import NumpyPaddedHaloOps
        args = bench.parse_args(ARGV[workload] + ['--gpus', str(world)])
        w = bench.build_workload(args, rank, world)
        lo, hi, periodic, period = w.slab
        props = {'elastic_block': par.ELASTIC_HALO_PROPS}.get(workload, par.WCSPH_HALO_PROPS)
        dec = par.SlabDecomposition(w.arrays, None, rank, world, axis=w.slab_axis, width=w.halo_width, lo=lo, hi=hi,
                                    props=props, periodic=periodic, period=period, dist=dist, protocol='padded',
                                    ops_factory=lambda pa, ax, p: NumpyPaddedHaloOps(pa, ax, props=p))
        for _ in range(4):
            dec.exchange()
            assert dec.verify()
        hs = dec.halos
        assert all(h.padded_exchanges == 3 and h.handshakes == 1 and h.repaired_exchanges == 0 for h in hs)
        nb = len(hs[0].neighbours())
        assert nb == (1 if rank in (0, world - 1) else 2)
        # a ghost travels without its promised h and m (every workload here has ONE of each per array)
        assert all('h' not in h.ops.props and 'm' not in h.ops.props for h in hs)
        live = [np.abs(a.x) < 1e17 for a in w.arrays]              # the rows of an array: real, ghosts, parked padding
        arrays = [a.extract_particles(np.nonzero(m)[0], name=a.name) for a, m in zip(w.arrays, live)]
        for a, b in zip(arrays, w.arrays):
            a.set_num_real_particles(b.get_number_of_particles(True))
        nn = orc.OracleNNPS(3, arrays, w.kernel.radius_scale)
        nn.update()
        ev = orc.OracleEval(arrays, w.eqs, w.kernel, nthreads=1)
        ev.set_nnps(nn)
        ev.compute(0.0, 1e-5)
        res = {}
        for a in arrays:
            n = a.get_number_of_particles(True)
            res[a.name + '/gid'] = np.asarray(a.gid[:n], dtype=np.int64)
            res[a.name + '/ghosts'] = np.array([a.get_number_of_particles() - n])
            for f in w.fields:
                if f in a.properties:
                    res[a.name + '/' + f] = a.get(f)[:n].copy()
        np.savez(out % rank, **res)
    finally:
        dist.destroy_process_group()


def _free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize('workload', ['cube', 'dam_break', 'elastic_block'])
def test_bench_workloads_on_eight_ranks_match_one_domain(tmp_path, oracle, workload):
    import bench
    from helpers import rel_err
    out = str(tmp_path / 'r%d.npz')
    mp.spawn(_worker, args=(WORLD, _free_port(), workload, out), nprocs=WORLD, join=True)
    # the whole domain alone: ONE problem for the strong-scaling workloads, the eight cubes side by side for the weak one
    args = bench.parse_args(ARGV[workload] + ['--gpus', '1'])
    if workload == 'cube':
        parts = [bench.build_workload(args, r, WORLD) for r in range(WORLD)]
        w1 = parts[0]
        for p in parts[1:]:
            w1.arrays[0].append_parray(p.arrays[0])
        w1.arrays[0].set_num_real_particles(w1.arrays[0].get_number_of_particles())
    else:
        w1 = bench.build_workload(args, 0, 1)
        if workload == 'elastic_block':
            # (the decomposed run recomputes p and the artificial stress on its ghosts: the same scheme here)
            from pysph_amd.solid_mech import ElasticSolidsScheme
            w1.eqs = ElasticSolidsScheme(['solid'], [], dim=3, ghost_recompute=True).get_equations()
    nn = oracle.OracleNNPS(3, w1.arrays, w1.kernel.radius_scale)
    nn.update()
    ev = oracle.OracleEval(w1.arrays, w1.eqs, w1.kernel, nthreads=4)
    ev.set_nnps(nn)
    ev.compute(0.0, 1e-5)
    ranks = [np.load(out % r) for r in range(WORLD)]
    for a in w1.arrays:
        n = a.get_number_of_particles(True)
        order = np.argsort(np.asarray(a.gid[:n]))
        gids = np.concatenate([d[a.name + '/gid'] for d in ranks])
        assert np.array_equal(np.sort(gids), np.asarray(a.gid[:n])[order]), a.name     # nobody lost or duplicated
        pos = np.searchsorted(np.asarray(a.gid[:n])[order], gids)
        for f in w1.fields:
            if f not in a.properties:
                continue
            got = np.concatenate([d[a.name + '/' + f] for d in ranks])
            e = rel_err(got, a.get(f)[:n][order][pos], scale=max(np.abs(a.get(f)[:n]).max(), 1e-300))
            assert e < 1e-12, (a.name, f, e)
    if workload == 'dam_break':
        # equal-count slabs: the fluid fills 38 % of the tank, geometric slabs would idle
        per_rank = [sum(d[a.name + '/gid'].size for a in w1.arrays) for d in ranks]
        assert max(per_rank) - min(per_rank) <= 3600, per_rank     # (the tank is cut along y: a lattice plane holds 2300-3500 particles, a rank three of the 27)
    assert all(int(d[w1.arrays[0].name + '/ghosts'][0]) > 0 for d in ranks[:4])


# ---------------------------------------------------------------------------
# The protocol UNDER MOTION on CPU ranks: what tests/test_integrator.py runs with two thread-ranks on a GPU, here
# with three gloo processes, the oracle as the evaluator and the numpy double as the device.
# ---------------------------------------------------------------------------
def _worker_motion(rank, world, port, out, nsteps, tight, lazy=0):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ['OMP_NUM_THREADS'] = '1'
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import pysph_amd.parallel as par
        from oracle import oracle as orc
        from oracle import steppers
        from pysph_amd.examples import dam_break_3d as db
        from test_parallel_gloo import NumpyPaddedHaloOps
        dx = 0.08
        if tight:      # capacities without headroom: the lattice plane that crosses a face outgrows its message
            par._capacity = lambda c: ((c + 8 + 7) // 8) * 8
            par._capacity_tight = par._capacity
        full = db.create_particles(dx)
        g0 = 0
        for a in full:
            n = a.get_number_of_particles()
            a.add_property('e0', data=np.arange(g0, g0 + n, dtype=np.float64))
            g0 += n
        full[0].u[:] = 4.0
        cuts = par.slab_bounds(np.concatenate([a.x for a in full]), world)
        lo, hi = float(cuts[rank]), float(cuts[rank + 1])
        arrays = [a.extract_particles(np.nonzero((a.x >= lo) & (a.x < hi))[0], name=a.name) for a in full]
        eqs = db.create_scheme(dx).get_equations()
        kernel = db.create_kernel()
        support = 2.0 * db.hdx * dx
        margin = 0.25 * support if lazy else 0.05 * support
        dec = par.SlabDecomposition(arrays, None, rank, world, axis=0, width=support + margin,
                                    lo=max(lo, -1e30), hi=min(hi, 1e30), dist=dist, protocol='padded',
                                    ops_factory=lambda pa, ax, p: NumpyPaddedHaloOps(pa, ax))
        pm = par.HipParallelManager(dec, rebalance_every=25, migrate_every=lazy or 1, margin=margin)
        dt = 0.125 * db.hdx * dx / (1.1 * db.c0)
        outs = ('arho', 'au', 'av', 'aw', 'ax', 'ay', 'az', 'p', 'cs', 'dt_cfl', 'dt_force')

        def evaluate(t):
            live = [a.extract_particles(np.nonzero(np.abs(a.x) < 1e17)[0], name=a.name) for a in arrays]
            for b, a in zip(live, arrays):
                b.set_num_real_particles(a.get_number_of_particles(True))
            nn = orc.OracleNNPS(3, live, kernel.radius_scale)
            nn.update()
            ev = orc.OracleEval(live, eqs, kernel, nthreads=1)
            ev.set_nnps(nn)
            ev.compute(t, dt)
            for b, a in zip(live, arrays):
                nr = a.get_number_of_particles(True)
                for f in outs:
                    if f in a.properties:
                        a.properties[f][:nr] = b.properties[f][:nr]

        def accel(t):          # Integrator.compute_accelerations
            pm.update()
            evaluate(t)
            while not pm.verify():
                evaluate(t)
        fluid = arrays[0]
        t = 0.0
        for _ in range(nsteps):          # EPECIntegrator.one_timestep with WCSPHStep (integrator.py:401-420)
            steppers.wcsph_initialize(fluid)
            accel(t)
            steppers.wcsph_stage(fluid, dt, 1)
            accel(t)
            steppers.wcsph_stage(fluid, dt, 2)
            t += dt
        res = {}
        for a in arrays:
            nr = a.get_number_of_particles(True)
            for f in ('e0', 'x', 'y', 'z', 'u', 'v', 'w', 'rho', 'au', 'av', 'aw', 'arho'):
                res[a.name + '/' + f] = a.properties[f][:nr].copy()
        hs = dec.halos
        res['stats'] = np.array([sum(h.padded_exchanges for h in hs), sum(h.repaired_exchanges for h in hs),
                                 sum(h.total_migrated for h in hs), pm.count, pm.max_excursion / support])
        np.savez(out % rank, **res)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('tight,lazy', [(False, 0), (True, 0), (False, 6)], ids=['roomy', 'tight', 'lazy-migration'])
def test_three_ranks_step_the_dam_break_through_the_padded_exchange(tmp_path, oracle, tight, lazy):
    """30 EPEC steps of the three-array dam break on three slab ranks (gloo), the round-trip-free exchange verified
    after every evaluation, particles migrating, a re-balance on the way, the front's lattice plane crossing a face --
    with capacities without headroom (`tight`) that plane outgrows its message and is repaired; with LAZY migration
    (`HipParallelManager(migrate_every=6, margin=...)`) ownership changes hands every sixth update only, the ghost layers a
    quarter of the kernel support wider, and the strayed particles are computed by their old rank meanwhile.  Same state
    as one domain stepped by the oracle, gid by gid."""
    from helpers import rel_err
    from oracle import steppers
    from pysph_amd.examples import dam_break_3d as db
    world, nsteps, dx = 3, 30, 0.08
    out = str(tmp_path / 'm%d.npz')
    mp.spawn(_worker_motion, args=(world, _free_port(), out, nsteps, tight, lazy), nprocs=world, join=True)
    full = db.create_particles(dx)
    g0 = 0
    for a in full:
        n = a.get_number_of_particles()
        a.add_property('e0', data=np.arange(g0, g0 + n, dtype=np.float64))
        g0 += n
    full[0].u[:] = 4.0
    eqs = db.create_scheme(dx).get_equations()
    kernel = db.create_kernel()
    dt = 0.125 * db.hdx * dx / (1.1 * db.c0)
    nn = oracle.OracleNNPS(3, full, kernel.radius_scale)
    ev = oracle.OracleEval(full, eqs, kernel, nthreads=4)
    ev.set_nnps(nn)
    t = 0.0
    for _ in range(nsteps):
        steppers.epec_step([full[0]], nn, ev, t, dt)
        t += dt
    ranks = [np.load(out % r) for r in range(world)]
    for a in full:
        gids = np.concatenate([d[a.name + '/e0'] for d in ranks]).astype(np.int64)
        base = int(a.e0.min()) if a.get_number_of_particles() else 0
        assert np.array_equal(np.sort(gids), np.arange(base, base + a.get_number_of_particles())), a.name
        for f in ('x', 'y', 'z', 'u', 'v', 'w', 'rho', 'au', 'av', 'aw', 'arho'):
            got = np.concatenate([d[a.name + '/' + f] for d in ranks])
            e = rel_err(got, a.get(f)[gids - base], scale=max(np.abs(a.get(f)).max(), 1e-300))
            assert e < 1e-9, (a.name, f, e)
    stats = np.array([d['stats'] for d in ranks])
    assert stats[:, 3].min() == 2 * nsteps                    # pm.update() before every evaluation
    assert stats[:, 0].sum() > 2 * nsteps                     # ... nearly all of them through the padded exchange
    assert stats[:, 2].sum() > 0                              # particles migrated (the column moves, the faces moved)
    assert (stats[:, 1].sum() > 0) == tight, stats            # repaired exactly where the capacities had no headroom
    if lazy:
        # the leavers had strayed beyond their faces, by less than the margin (a quarter of the support)
        assert 0.0 < stats[:, 4].max() < 0.25, stats[:, 4]
