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


import pytest


@pytest.mark.parametrize('n', [2, 8])
def test_gpus_flag_spawns_ranks(n):
    """(8: the launch the driver's scaling run makes -- eight processes rendezvous on 127.0.0.1)"""
    env = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    env['SPH_BENCH_DRYRUN'] = '1'
    env['OMP_NUM_THREADS'] = '1'
    r = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', str(n),
                        '--steps', '1', '--warmup', '0'], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, universal_newlines=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1]
    out = json.loads(line)
    assert out == {'dryrun': True, 'n_gpus': n, 'rank_sum': n * (n + 1) / 2.0}


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


def test_field_error_elementwise_figure():
    """the stricter figure next to the norm-wise one: every element against its
    own magnitude, floored at 1e-6 of the field's scale"""
    import numpy as np
    import bench
    b = np.array([1.0, 1e-3, 1e-9, -2.0])
    a = b.copy(); a[1] += 1e-12
    nw, ew = bench.field_error(a, b, [b], elementwise=True)
    assert abs(nw - 0.5e-12) < 1e-20 and abs(ew - 1e-9) < 1e-15
    a = b.copy(); a[2] += 1e-12            # |b_i| below the floor: judged against 1e-6 * 2.0
    nw, ew = bench.field_error(a, b, [b], elementwise=True)
    assert abs(ew - 1e-12 / 2e-6) < 1e-15
    a[0] = np.nan
    assert bench.field_error(a, b, [b], elementwise=True) == (1e300, 1e300)
    assert bench.field_error(np.zeros(0), np.zeros(0), [], elementwise=True) == (0.0, 0.0)


def test_elementwise_error_of_a_reordered_sum():
    """What the element-wise figure can resolve: the oracle (the reference's
    arithmetic, term by term) against ITSELF with the particles in another order
    -- same pairs, same terms, another summation order.  Norm-wise the two agree
    to 1e-15; element-wise a particle whose sum nearly cancels differs by 1e-11
    and more.  This is why the GPU tests bound the element-wise figure at 1e-8
    while the norm-wise one stays at the north-star 1e-10."""
    import numpy as np
    import bench
    from oracle import oracle as orc
    from pysph_amd import kernels as K

    def run(pa):
        nn = orc.OracleNNPS(3, [pa], 2.0)
        nn.update()
        ev = orc.OracleEval([pa], eqs, K.WendlandQuintic(dim=3), nthreads=4)
        ev.set_nnps(nn)
        ev.compute(0.0, 1e-5)

    pa, dx = bench.make_cube(32)
    eqs = bench.cube_equations(dx)
    perm = np.random.default_rng(0).permutation(pa.get_number_of_particles())
    pb = pa.extract_particles(perm, name='fluid')
    run(pa)
    run(pb)
    worst_nw = worst_ew = 0.0
    for f in ('arho', 'au', 'av', 'aw', 'ax', 'ay', 'az'):
        g = [pa.get(x) for x in bench._scale_group(f) if x in pa.properties]
        nw, ew = bench.field_error(pb.get(f), pa.get(f)[perm], g, elementwise=True)
        worst_nw, worst_ew = max(worst_nw, nw), max(worst_ew, ew)
    assert worst_nw < 1e-14
    assert 1e-14 < worst_ew < 1e-8, worst_ew


def test_rings3d_workload_is_the_named_configuration():
    """BASELINE config 5 / SURVEY 8(d) S-rings3d: rings.py material, hdx 1.5,
    rho0 1, two bodies approaching at +-0.059 cs; 2.0 M particles at the default dx"""
    import numpy as np
    import bench
    pa, kernel = bench.make_rings3d(2e-3)
    n = pa.get_number_of_particles()
    assert n % 2 == 0 and type(kernel).__name__ == 'CubicSpline'
    assert abs(pa.h[0] / 2e-3 - 1.5) < 1e-12 and abs(pa.m[0] - 1.0 * 2e-3 ** 3) < 1e-20
    assert float(pa.E[0]) == 1e7 and float(pa.nu[0]) == 0.3975 and float(pa.rho_ref[0]) == 1.0
    cs = float(pa.cs[0])
    left = pa.x < 0.041
    assert left.sum() == n // 2
    assert abs(pa.u[left].mean() / cs - 0.059) < 1e-3 and abs(pa.u[~left].mean() / cs + 0.059) < 1e-3
    r = np.sqrt((pa.x[left]) ** 2 + pa.y[left] ** 2 + pa.z[left] ** 2)
    assert r.min() >= 0.03 - 1e-12 and r.max() < 0.04
    # the default spacing of the benchmark: 2.0 M particles
    g = np.arange(-0.04, 0.04, 5.372e-4)
    x, y, z = np.meshgrid(g, g, g, indexing='ij', sparse=True)
    d = x * x + y * y + z * z
    assert abs(2 * np.count_nonzero((d >= 0.03 ** 2) & (d < 0.04 ** 2)) - 2.0e6) < 1e4


def test_timed_region_is_exactly_k_steps_behind_the_breakdown_and_the_warm_up():
    """bench.timed: the run's first step (dropped from the class figures), K breakdown steps with every class timed,
    W warm-up steps and EXACTLY K timed steps with the pair launches timed only -- in that order"""
    import bench

    log = []

    class Ctx(object):
        def timer_enable(self, on):
            log.append(('enable', on))

        def timer_reset(self):
            log.append(('reset',))

        def timer_get(self, key):
            return (1.0, 1)

    def step():
        log.append(('step',))

    def barrier():
        log.append(('barrier',))

    elapsed, timers = bench.timed(5, 2, step, barrier, Ctx())
    assert elapsed >= 0.0 and 'pair' in timers and 'nnps' in timers and 'n_async' in timers
    names = [e[0] + (str(e[1]) if len(e) > 1 else '') for e in log]
    assert names == (['enable1', 'step', 'reset'] + ['step'] * 5 + ['barrier', 'enable2'] + ['step'] * 2 +
                     ['reset', 'barrier'] + ['step'] * 5 + ['barrier', 'enable0']), names
