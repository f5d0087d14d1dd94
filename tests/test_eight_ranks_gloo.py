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
    'dam_break': ['--workload', 'dam_break', '--dx', '0.05'],
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
        dec = par.SlabDecomposition(w.arrays, None, rank, world, axis=0, width=w.halo_width, lo=lo, hi=hi,
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
        assert max(per_rank) - min(per_rank) <= 600, per_rank      # (a lattice plane of the tank holds ~460 particles)
    assert all(int(d[w1.arrays[0].name + '/ghosts'][0]) > 0 for d in ranks[:4])
