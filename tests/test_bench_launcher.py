"""`python bench.py --gpus N` must start N ranks by itself (the driver runs it
without torchrun and without RANK / WORLD_SIZE): the launcher re-executes the
script under torch.distributed.run with a 127.0.0.1 rendezvous.  Checked on CPU
with SPH_BENCH_DRYRUN=1 (gloo, no particles, no timing): N ranks come up, agree
on the world size, all-reduce, and rank 0 prints the one JSON line."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_flag_spawns_ranks():
    env = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    env['SPH_BENCH_DRYRUN'] = '1'
    r = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '2',
                        '--steps', '1', '--warmup', '0'], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, universal_newlines=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1]
    out = json.loads(line)
    assert out == {'dryrun': True, 'n_gpus': 2, 'rank_sum': 3.0}


def test_world_size_must_match_gpus():
    """under torchrun with the wrong --nproc-per-node the ranks refuse to run
    (the round-1 bench silently measured one rank and called it N)"""
    env = dict(os.environ)
    env.update(RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', SPH_BENCH_DRYRUN='1')
    r = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '4'],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       universal_newlines=True, timeout=120)
    assert r.returncode != 0 and 'WORLD_SIZE=1' in r.stderr


def test_launch_command_shape():
    sys.path.insert(0, REPO)
    import bench
    cmd = bench.launch_command(['--gpus', '8', '--steps', '3'], 8, port=12345)
    assert cmd[1:4] == ['-m', 'torch.distributed.run', '--nnodes=1']
    assert cmd[cmd.index('--nproc-per-node') + 1] == '8'
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    assert cmd[-4:] == ['--gpus', '8', '--steps', '3'] and cmd[-5].endswith('bench.py')


def test_field_error_treats_nan_as_failure():
    """the bench's own parity check: a NaN in the device result must not pass as
    'no error found' (np.max of NaN never compares greater than the running max)"""
    import numpy as np
    import bench
    b = np.array([1.0, -2.0, 0.5])
    assert bench.field_error(b.copy(), b, [b]) == 0.0
    a = b.copy(); a[1] += 2e-9
    assert abs(bench.field_error(a, b, [b]) - 1e-9) < 1e-12
    a[0] = np.nan
    assert bench.field_error(a, b, [b]) == 1e300
    a[0] = np.inf
    assert bench.field_error(a, b, [b]) == 1e300
    # a component that vanishes by symmetry borrows the scale of its vector
    z = np.zeros(3)
    assert bench.field_error(z + 1e-12, z, [z, b]) == 1e-12 / 2.0
    assert bench.field_error(z + 1e-12, z, [z]) == 1e-12       # no scale at all: absolute
    assert bench.field_error(np.zeros(0), np.zeros(0), []) == 0.0
