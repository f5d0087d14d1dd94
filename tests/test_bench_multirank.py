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
