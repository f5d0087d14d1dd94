"""The N>1 code path of bench.py on ONE GPU: two threads play two ranks, with
tests/helpers.ThreadDist standing in for torch.distributed (RCCL needs one
device per rank).  Everything else is the real thing: slab halos / slab
decomposition through the device primitives, the timed loop, the MAX reduction
of the elapsed time and the JSON contract."""
import json
import os
import sys
import threading

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def _run_two_ranks(argv):
    import bench
    from helpers import ThreadDist
    hub = ThreadDist(2)
    outs, errors = {}, []

    def rank_main(rank):
        try:
            args = bench.parse_args(argv)
            outs[rank] = bench.run(args, rank, 0, 2, hub.view(rank))
        except BaseException:
            import traceback
            errors.append(traceback.format_exc())
            try:
                hub.barrier.abort()
            except Exception:
                pass

    ths = [threading.Thread(target=rank_main, args=(r,)) for r in range(2)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(600)
    assert not errors, errors[0]
    assert outs[1] is None and outs[0] is not None
    return outs[0]


@pytest.mark.gpu
def test_bench_weak_scaling_path_two_ranks():
    out = _run_two_ranks(['--gpus', '2', '--n1', '32', '--steps', '3', '--warmup', '1',
                          '--no-cpu-baseline'])
    json.dumps(out)
    assert out['n_gpus'] == 2 and out['scaling'] == 'weak'
    assert out['config']['parallelism'] == 'slab2'
    assert out['config']['particles_per_gpu'] == 32 ** 3
    assert out['value'] > 0 and out['steps'] == 3 and out['warmup'] == 1
    assert abs(out['value'] - 2 * 32 ** 3 * 3 / (out['ms_per_step'] * 3e-3)) < 1e-6 * out['value']
    assert out['roofline']['bound'] == 'hbm' and 'cpu_baseline' not in out


@pytest.mark.gpu
def test_bench_dam_break_strong_scaling_path_two_ranks():
    out = _run_two_ranks(['--gpus', '2', '--workload', 'dam_break', '--dx', '0.05',
                          '--steps', '2', '--warmup', '1', '--no-cpu-baseline'])
    assert out['n_gpus'] == 2 and out['scaling'] == 'strong'
    from pysph_amd.examples import dam_break_3d as db
    total = sum(a.get_number_of_particles() for a in db.create_particles(0.05))
    assert abs(out['value'] * out['ms_per_step'] * 1e-3 - total) < 1e-6 * total


def _run_two_ranks_collect(argv, fields, one_problem=False, nsteps=1, counters=None):
    """two thread-ranks run one step of the workload; returns {gid: row} of the
    requested fields over both ranks, and the same from ONE domain"""
    import numpy as np
    import torch
    import bench
    from helpers import ThreadDist
    from pysph_amd import device as dev
    hub = ThreadDist(2)
    rows, errors = {}, []

    def rank_main(rank):
        try:
            args = bench.parse_args(argv + ['--gpus', '2'])
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                ctx = dev.HipContext(0, stream.cuda_stream)
                bench.apply_options(args, ctx)
                w = bench.build_workload(args, rank, 2)
                nnps, a_eval, halo, domain, step, _ = bench.setup(args, w, rank, 2, hub.view(rank), ctx)
                for _ in range(nsteps):
                    step()
                if counters is not None:
                    counters[rank] = (bool(getattr(w, 'overlap_halo', False)), ctx.timer_get('n_phase2')[1],
                                      halo.halos[0].padded_exchanges if halo is not None and hasattr(halo, 'halos') else 0,
                                      ctx.timer_get('n_async')[1], ctx.timer_get('n_mass_fused')[1], ctx.timer_get('n_merged')[1])
                pa = w.arrays[0]
                pa.gpu.sync_host()
                n = pa.get_number_of_particles(True)
                rows[rank] = (np.asarray(pa.gid[:n]).copy(),
                              np.stack([np.asarray(pa.get(f))[:n] for f in fields], 1))
                del nnps, a_eval, step
                ctx.close()
        except BaseException:
            import traceback
            errors.append(traceback.format_exc())
            try:
                hub.barrier.abort()
            except Exception:
                pass

    ths = [threading.Thread(target=rank_main, args=(r,)) for r in range(2)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(600)
    assert not errors, errors[0]
    gid = np.concatenate([rows[0][0], rows[1][0]])
    val = np.concatenate([rows[0][1], rows[1][1]], 0)
    args = bench.parse_args(argv + ['--gpus', '2'])
    if one_problem:
        # strong scaling: the ranks hold the two slabs of what rank 0 of 1 holds whole
        w1 = bench.build_workload(args, 0, 1)
    else:
        # one domain: both cubes in one array
        parts = [bench.build_workload(args, r, 2) for r in range(2)]
        w1 = parts[0]
        w1.arrays[0].append_parray(parts[1].arrays[0])
        w1.arrays[0].set_num_real_particles(w1.arrays[0].get_number_of_particles())
    ctx = dev.HipContext(0, torch.cuda.current_stream().cuda_stream)
    bench.apply_options(args, ctx)
    nnps, a_eval, halo, domain, step, _ = bench.setup(args, w1, 0, 1, None, ctx)
    step()
    pa = w1.arrays[0]
    pa.gpu.sync_host()
    n = pa.get_number_of_particles(True)
    gid1 = np.asarray(pa.gid[:n])
    val1 = np.stack([np.asarray(pa.get(f))[:n] for f in fields], 1)
    ctx.close()
    return gid, val, gid1, val1


@pytest.mark.gpu
def test_taylor_green_two_slabs_periodic_matches_one_domain():
    """Slab decomposition along x (periodic: slab 0 <-> slab 1 wrap with the
    coordinate shift, nnps_base.pyx:841-856) COMBINED with the device domain
    manager's periodic images in y and z: the remote ghosts are exchanged
    first, their y/z images made after (HipDomainManager(slab=...)).  Every real
    particle matches the single-domain evaluation, gid by gid -- the reference's
    own recipe for its parallel runs (parallel/tests/example_test_case.py:143-166)."""
    import numpy as np
    fields = ['rho', 'V', 'au', 'av', 'aw', 'auhat', 'avhat', 'awhat']
    gid, val, gid1, val1 = _run_two_ranks_collect(
        ['--workload', 'taylor_green', '--n1', '20', '--steps', '1', '--warmup', '0'], fields)
    assert gid.size == gid1.size == 2 * 20 ** 3
    a = val[np.argsort(gid)]
    b = val1[np.argsort(gid1)]
    scale_acc = np.max(np.abs(b[:, 2:]))
    assert np.max(np.abs(a[:, :2] - b[:, :2])) / np.max(np.abs(b[:, :2])) < 1e-12
    assert np.max(np.abs(a[:, 2:] - b[:, 2:])) / scale_acc < 1e-10


@pytest.mark.gpu
def test_cube_two_slabs_matches_one_domain_by_gid():
    import numpy as np
    fields = ['arho', 'au', 'av', 'aw', 'ax', 'ay', 'az']
    gid, val, gid1, val1 = _run_two_ranks_collect(
        ['--n1', '24', '--steps', '1', '--warmup', '0'], fields)
    a = val[np.argsort(gid)]
    b = val1[np.argsort(gid1)]
    for k in range(len(fields)):
        assert np.max(np.abs(a[:, k] - b[:, k])) / np.max(np.abs(b[:, k])) < 1e-10, fields[k]


@pytest.mark.gpu
@pytest.mark.parametrize('argv,one', [(['--n1', '24'], False), (['--n1', '24', '--vary-h', '0.1'], False),
                                      (['--workload', 'dam_break', '--dx', '0.04'], True)],
                         ids=['cube', 'cube-variable-h', 'dam-break-three-arrays'])
def test_padded_exchange_runs_without_round_trips_and_matches_one_domain(argv, one):
    """round 5, the 'padded' ghost exchange (default of bench.py): from the second exchange on the receiver appends
    whole fixed-capacity messages -- padding rows parked far outside the domain behind the ghosts -- so neither the exchange nor the neighbour
    update that follows waits for the device (counts and flags are read one exchange later); with ONE h and ONE mass
    per array on every rank the appends keep what the update knows of h and m (uniform-mass records stay).  Results
    against one domain, gid by gid."""
    import numpy as np
    fields = ['arho', 'au', 'av', 'aw', 'ax', 'ay', 'az', 'dt_cfl']
    cnt = {}
    gid, val, gid1, val1 = _run_two_ranks_collect(argv + ['--steps', '1', '--warmup', '0'], fields, one_problem=one,
                                                  nsteps=4, counters=cnt)
    for r in (0, 1):
        assert cnt[r][2] >= 3, cnt                  # padded exchanges
        if '--vary-h' not in argv:
            assert cnt[r][3] >= 2, cnt              # neighbour updates without a round trip
            assert cnt[r][4] >= 1, cnt              # uniform-mass records in use
        if one:
            # the three-array dam break stays on the ONE-launch merged evaluation on BOTH ranks: rank 0 owns no obstacle
            # particle -- its obstacle array is nothing but the padding rows of an empty message -- and such an array
            # adopts the promised h and m (round 5: "mass unknown" for ever, per-destination launches and a round trip
            # in every neighbour update on every rank without an obstacle)
            assert cnt[r][5] >= 2, cnt
    a = val[np.argsort(gid)]
    b = val1[np.argsort(gid1)]
    for k in range(len(fields)):
        assert np.max(np.abs(a[:, k] - b[:, k])) / np.max(np.abs(b[:, k])) < 1e-10, fields[k]


@pytest.mark.gpu
@pytest.mark.parametrize('extra,split', [(['--overlap-halo'], True), ([], False), (['--overlap-halo', '--vary-h', '0.1'], True),
                                         (['--overlap-halo', '--dtype', 'f32'], True),
                                         (['--overlap-halo', '--opt', 'split_pair=1'], True),
                                         (['--overlap-halo', '--opt', 'split_pair=1', '--vary-h', '0.1'], True)],
                         ids=['overlapped', 'plain', 'overlapped-variable-h', 'overlapped-fp32',
                              'overlapped-split-pair', 'overlapped-split-pair-variable-h'])
def test_cube_two_slabs_overlapped_exchange_matches_one_domain(extra, split):
    """round 4: the ghost exchange overlapped with the evaluation -- transfers posted; neighbour update, EOS and records
    of the real particles (option split_pair: and their interior wave tiles); THEN the ghosts appended, binned into
    tables of their own and read as a second source segment by the wave tiles that can reach them (sph_group.phase
    1 / 2) -- against one domain, gid by gid"""
    import numpy as np
    fields = ['arho', 'au', 'av', 'aw', 'ax', 'ay', 'az', 'dt_cfl']
    cnt = {}
    gid, val, gid1, val1 = _run_two_ranks_collect(['--n1', '24', '--steps', '1', '--warmup', '0'] + extra, fields,
                                                  nsteps=3, counters=cnt)
    for r in (0, 1):
        assert cnt[r][0] == split and (cnt[r][1] >= 2) == split, cnt   # (the set-up exchanges already sized the messages)
    a = val[np.argsort(gid)]
    b = val1[np.argsort(gid1)]
    tol = 5e-5 if '--dtype' in extra else 1e-10
    for k in range(len(fields)):
        assert np.max(np.abs(a[:, k] - b[:, k])) / np.max(np.abs(b[:, k])) < tol, fields[k]


@pytest.mark.gpu
@pytest.mark.parametrize('argv', [['--workload', 'elastic_block', '--n1', '24'],
                                  ['--workload', 'elastic', '--rings-dx', '2e-3', '--rings-spacing', '0.02']],
                         ids=['block', 'overlapping-shells'])
def test_elastic_two_slabs_match_one_domain_by_gid(argv):
    """The elastic set on N > 1: its second group reads p and the artificial
    stress r_ij of ghost SOURCES; the halo carries rho, cs and s_ij and the two
    no-source equations recompute p, r_ij on the ghosts
    (ElasticSolidsScheme(ghost_recompute=True)) where the reference refreshes
    remote properties in mid-evaluation (parallel_manager.pyx:159-210).  One body
    cut in two at the quantile of x (second case: the two shells pushed into each
    other so that the cut runs through material) against the whole on one rank."""
    import numpy as np
    fields = ['arho', 'au', 'av', 'aw', 'ax', 'ay', 'az', 'p', 'as00', 'as01', 'as02', 'as11', 'as12',
              'as22', 'r00', 'r11', 'r22', 'v00', 'v11', 'v22']
    gid, val, gid1, val1 = _run_two_ranks_collect(argv + ['--steps', '1', '--warmup', '0'], fields,
                                                  one_problem=True)
    assert gid.size == gid1.size and np.array_equal(np.sort(gid), np.sort(gid1))
    a = val[np.argsort(gid)]
    b = val1[np.argsort(gid1)]
    import bench
    for k, f in enumerate(fields):
        group = [fields.index(g) for g in bench._scale_group(f) if g in fields]
        scale = max(np.max(np.abs(b[:, j])) for j in group)
        if scale == 0.0:            # the block starts at the reference density: p = 0 everywhere
            assert np.max(np.abs(a[:, k])) == 0.0, f
        else:
            assert np.max(np.abs(a[:, k] - b[:, k])) / scale < 1e-10, f


@pytest.mark.gpu
def test_bench_elastic_strong_scaling_path_two_ranks():
    out = _run_two_ranks(['--gpus', '2', '--workload', 'elastic', '--rings-dx', '2e-3', '--steps', '2',
                          '--warmup', '1', '--no-cpu-baseline'])
    assert out['n_gpus'] == 2 and out['scaling'] == 'strong' and out['config']['parallelism'] == 'slab2'
    import bench
    total = bench.make_rings3d(2e-3)[0].get_number_of_particles()
    assert abs(out['value'] * out['ms_per_step'] * 1e-3 - total) < 1e-6 * total


@pytest.mark.gpu
def test_bench_taylor_green_path_two_ranks():
    out = _run_two_ranks(['--gpus', '2', '--workload', 'taylor_green', '--n1', '20', '--steps', '2',
                          '--warmup', '1', '--no-cpu-baseline'])
    assert out['n_gpus'] == 2 and out['config']['parallelism'] == 'slab2'
    assert out['config']['particles_per_gpu'] == 20 ** 3


@pytest.mark.gpu
def test_direct_rccl_transport_equals_torch_transport_on_one_rank():
    """round 6: the point-to-point transfers of the ghost exchange straight on RCCL on the context's stream
    (SphCommTransport -> libsphcomm.so sph_comm_sendrecv; the default of `bench.py --gpus N`) against
    torch.distributed's batch_isend_irecv: a self-periodic slab (world size 1: both faces talk to the same peer, the
    message-order case) gets the same ghosts and the same results, through the round-trip-free protocol."""
    import numpy as np
    import torch
    import torch.distributed as dist
    import bench
    from pysph_amd import device as dev
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29641')
    own = not dist.is_initialized()
    if own:
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    res = {}
    try:
        for transport in ('torch', 'sphcomm'):
            os.environ['SPH_HALO_TRANSPORT'] = transport
            args = bench.parse_args(['--n1', '40', '--self-slab', '--no-cpu-baseline', '--no-extras'])
            ts = torch.cuda.Stream()
            with torch.cuda.stream(ts):
                ctx = dev.HipContext(0, ts.cuda_stream)
                bench.apply_options(args, ctx)
                w = bench.build_workload(args, 0, 1)
                nnps, a_eval, halo, domain, step, _ = bench.setup(args, w, 0, 1, dist, ctx)
                assert (getattr(w, 'transport', None) is not None) == (transport == 'sphcomm')
                for _ in range(4):
                    step()
                pa = w.arrays[0]
                pa.gpu.sync_host()
                n, nr = pa.gpu.get_number_of_particles(), pa.gpu.get_number_of_particles(True)
                res[transport] = ({f: np.array(pa.get(f, only_real_particles=False)[:n]) for f in ('x', 'u', 'rho', 'au', 'arho', 'ax')},
                                  n, nr, halo.halos[0].padded_exchanges)
                del nnps, a_eval, step, halo
                ctx.close()
    finally:
        os.environ.pop('SPH_HALO_TRANSPORT', None)
        if own:
            dist.destroy_process_group()
    a, b = res['torch'], res['sphcomm']
    assert a[1:3] == b[1:3] and a[1] > a[2] and b[3] >= 3
    for f in a[0]:
        assert np.array_equal(a[0][f], b[0][f]), f
