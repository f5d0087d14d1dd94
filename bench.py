#!/usr/bin/env python
"""Benchmark of the SPH acceleration-eval hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one resident particle set:
``nnps.update()`` (bounds, cell keys, radix sort, cell ranges) followed by
``AccelerationEval.compute()`` (EOS + pack + fused WCSPH pair kernel), exactly
what ``Integrator.compute_accelerations`` runs (pysph/sph/integrator.py:274-286).
Inputs are already in HBM when the timed region starts.

Workload (BASELINE.json metric): WCSPH, dam-break parameter set
(WendlandQuintic, hdx 1.3, alpha 0.25, gamma 7, c0 = 10 sqrt(2 g 0.55),
gz = -9.81), synthetic uniform 3-D distribution "S-cube" of SURVEY.md 8(d):
159^3 = 4.02 M particles per GPU, jitter U(+-0.1 dx), seeded.  For N > 1 the
domain is N such cubes side by side along x (weak scaling), slab-decomposed one
cube per rank, with a ghost-particle halo exchange on RCCL (torch.distributed
nccl) before every step.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

ALGO_BYTES_PAIR = 160.0      # pair-loop kernel, fp64, per particle-update
ALGO_BYTES_UPDATE = 184.0    # whole compute() (EOS + pair pass)
FLOP_PER_PAIR = 130.0            # fused WCSPH fluid<-fluid group, reference operation count
FP64_VECTOR_PEAK_TFLOPS = 78.6   # MI355X fp64 vector (non-MFMA) peak
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8 TB/s spec


def make_cube(n1, x_offset=0.0, seed=1234, hdx=1.3, nx=None):
    from pysph_amd.particle_array import get_particle_array_wcsph
    from pysph_amd.examples import dam_break_3d as db
    rng = np.random.default_rng(seed)
    dx = 1.0 / n1
    nx = nx or n1
    gx = np.arange(nx) * dx + x_offset
    g = np.arange(n1) * dx
    x, y, z = [a.ravel().copy() for a in np.meshgrid(gx, g, g, indexing='ij')]
    n = x.size
    for a in (x, y, z):
        a += 0.1 * dx * rng.uniform(-1, 1, n)
    pa = get_particle_array_wcsph(
        name='fluid', x=x, y=y, z=z, h=hdx * dx * np.ones(n),
        m=db.ro * dx ** 3 * np.ones(n),
        rho=db.ro * (1 + 0.01 * rng.uniform(-1, 1, n)),
        u=0.1 * db.c0 * rng.uniform(-1, 1, n),
        v=0.1 * db.c0 * rng.uniform(-1, 1, n),
        w=0.1 * db.c0 * rng.uniform(-1, 1, n))
    return pa, dx


def cube_equations(dx, hdx=1.3):
    from pysph_amd.scheme import WCSPHScheme
    from pysph_amd.examples import dam_break_3d as db
    s = WCSPHScheme(['fluid'], [], dim=3, rho0=db.ro, c0=db.c0, h0=hdx * dx,
                    hdx=hdx, gz=-9.81, alpha=db.alpha, beta=db.beta,
                    gamma=db.gamma)
    return s.get_equations()


def cpu_baseline(n1=100, target_seconds=15.0):
    """The oracle ("port": C/OpenMP restatement of the reference's Cython path)
    timed on this box's host cores on a bounded sample of the same workload."""
    from oracle import oracle as orc
    from pysph_amd import kernels as K
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    pa, dx = make_cube(n1, seed=99)
    eqs = cube_equations(dx)
    nn = orc.OracleNNPS(3, [pa], 2.0)
    ev = orc.OracleEval([pa], eqs, K.WendlandQuintic(dim=3), nthreads=avail)
    ev.set_nnps(nn)

    def one():
        t0 = time.perf_counter()
        nn.update()
        ev.compute(0.0, 1e-5)
        return time.perf_counter() - t0
    # thread count: the reference advises tuning it (installation.rst:1043);
    # take the fastest of a short sweep, report the count actually used
    best = None
    for nt in sorted(set([avail, max(avail // 2, 1), max(avail // 4, 1)])):
        ev.nthreads = nt
        one()
        t = one()
        if best is None or t < best[0]:
            best = (t, nt)
    t_first, cores = best
    ev.nthreads = cores
    reps = int(max(1, min(10, target_seconds / max(t_first, 1e-3))))
    ts = [one() for _ in range(reps)]
    t = float(np.median(ts))
    n = n1 ** 3
    return {'value': n / t, 'unit': 'particle-updates/s', 'cores': cores,
            'kind': 'port',
            'sample': 'S-cube %d^3=%d particles, WCSPH db set, fp64, '
                      'nnps.update+compute, median of %d passes (OpenMP '
                      'schedule(dynamic,64), %d threads)' % (n1, n, reps, cores)}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--n1', type=int, default=159, help='lattice side per GPU')
    ap.add_argument('--workload', default='cube',
                    choices=['cube', 'dam_break', 'taylor_green', 'elastic'],
                    help='cube = S-cube WCSPH (headline); others: BASELINE configs 2/3/5')
    ap.add_argument('--dx', type=float, default=0.0087, help='dam_break spacing')
    ap.add_argument('--variant', type=int, default=3)
    ap.add_argument('--ablate', type=int, default=0, help='profiling only')
    ap.add_argument('--opt', action='append', default=[], help='key=value library option')
    ap.add_argument('--no-reorder', action='store_true',
                    help='skip Solver.reorder_particles() before timing')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-n1', type=int, default=100)
    return ap.parse_args(argv)


def main():
    args = parse_args()
    import torch
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU fallback)')
    torch.cuda.set_device(local_rank)
    dist = None
    # SPH_BENCH_FORCE_DIST=1: bring RCCL up even for one rank (checks the
    # process-group plumbing and the stdout ordering on a 1-GPU box)
    if world > 1 or os.environ.get('SPH_BENCH_FORCE_DIST') == '1':
        import torch.distributed as dist
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world,
                                device_id=torch.device('cuda', local_rank))
    out = run(args, rank, local_rank, world, dist)
    # The JSON line must be the LAST thing on stdout: RCCL's version banner
    # (NCCL_DEBUG=VERSION on the GPU boxes) sits in the C stdio buffer of every
    # rank until the process exits -- push it out before the final barrier,
    # tear the process group down, then print.
    import ctypes
    libc = ctypes.CDLL(None)
    libc.fflush(None)
    sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
        libc.fflush(None)
    if out is not None:
        print(json.dumps(out), flush=True)


def run(args, rank, local_rank, world, dist):
    """One rank of the benchmark; `dist` is torch.distributed (RCCL) or, in the
    one-GPU test of the N>1 code path, an in-process stand-in with the same
    calls (tests/helpers.ThreadDist).  Returns the JSON dict on rank 0."""
    import torch

    from pysph_amd import device as dev
    from pysph_amd import kernels as K
    from pysph_amd.acceleration_eval import AccelerationEval, SPHCompiler
    from pysph_amd.nnps import HipNNPS

    tstream = torch.cuda.Stream()      # kernels, copies and RCCL share it
    torch.cuda.set_stream(tstream)
    ctx = dev.HipContext(local_rank, tstream.cuda_stream)
    ctx.set_option('pair_variant', args.variant)
    if args.ablate:
        ctx.set_option('ablate', args.ablate)
    for kv in args.opt:
        k, v = kv.split('=')
        ctx.set_option(k, int(v))

    n1 = args.n1
    domain = None
    scaling = 'weak'
    algo_pair = ALGO_BYTES_PAIR
    if args.workload == 'cube':
        pa, dx = make_cube(n1, x_offset=float(rank), seed=1234 + rank)
        arrays = [pa]
        eqs = cube_equations(dx)
        kernel = K.WendlandQuintic(dim=3)
        wname = ('S-cube WCSPH dam-break parameter set (WendlandQuintic, hdx 1.3), '
                 '%d^3 = %d particles per GPU, jitter 0.1dx, seed 1234' %
                 (n1, pa.get_number_of_particles()))
    elif args.workload == 'dam_break':
        from pysph_amd.examples import dam_break_3d as db
        arrays = db.create_particles(args.dx)
        if world > 1:
            # C4: ONE tank cut into `world` slabs along x at the quantiles of
            # all particles' x (equal counts: the fluid fills 38 % of the tank)
            from pysph_amd.parallel import slab_bounds
            cuts = slab_bounds(np.concatenate([a.x for a in arrays]), world)
            slab_lo = -1e30 if rank == 0 else float(cuts[rank])
            slab_hi = 1e30 if rank == world - 1 else float(cuts[rank + 1])
            arrays = [a.extract_particles(np.nonzero((a.x >= slab_lo) & (a.x < slab_hi))[0],
                                          name=a.name) for a in arrays]
            scaling = 'strong'
        dx = args.dx
        eqs = db.create_scheme(dx).get_equations()
        kernel = db.create_kernel()
        wname = ('3D dam break (dam_break_3d.py geometry), dx=%g: %s' % (
            dx, ', '.join('%s %d' % (a.name, a.get_number_of_particles())
                          for a in arrays)))
    elif args.workload == 'taylor_green':
        from pysph_amd.domain import HipDomainManager
        from pysph_amd.particle_array import get_particle_array_tvf_fluid
        from pysph_amd.scheme import TVFScheme
        if world > 1:
            raise SystemExit('taylor_green workload: single GPU only in this round')
        dx = 1.0 / n1
        g = (np.arange(n1) + 0.5) * dx
        x, y, z = [a.ravel().copy() for a in np.meshgrid(g, g, g, indexing='ij')]
        pa = get_particle_array_tvf_fluid(
            name='fluid', x=x, y=y, z=z, h=dx * np.ones(x.size),
            m=dx ** 3 * np.ones(x.size), rho=np.ones(x.size),
            u=-np.cos(2 * np.pi * x) * np.sin(2 * np.pi * y),
            v=np.sin(2 * np.pi * x) * np.cos(2 * np.pi * y))
        pa.uhat[:] = pa.u
        pa.vhat[:] = pa.v
        arrays = [pa]
        eqs = TVFScheme(['fluid'], [], dim=3, rho0=1.0, c0=10.0, nu=0.01,
                        p0=100.0, pb=100.0, h0=dx).get_equations()
        kernel = K.QuinticSpline(dim=3)
        domain = HipDomainManager(ctx=ctx, xmin=0, xmax=1, ymin=0, ymax=1, zmin=0,
                                  zmax=1, periodic_in_x=True, periodic_in_y=True,
                                  periodic_in_z=True)
        algo_pair = 160.0   # force pass: 112 R + 48 W (SURVEY 8d TVF pass 2)
        wname = ('Taylor-Green 3D TVF (taylor_green.py parameters), periodic unit '
                 'cube %d^3 = %d particles, QuinticSpline hdx 1.0' % (n1, x.size))
    else:
        from pysph_amd.solid_mech import (ElasticSolidsScheme,
                                          get_particle_array_elastic_dynamics)
        if world > 1:
            raise SystemExit('elastic workload: single GPU only in this round')
        dx = 1.0 / n1
        g = (np.arange(n1) + 0.5) * dx
        x, y, z = [a.ravel().copy() for a in np.meshgrid(g, g, g, indexing='ij')]
        rng = np.random.default_rng(7)
        E, nu, rho0 = 1e7, 0.3975, 1.2            # rings.py:21-38
        kernel = K.CubicSpline(dim=3)
        h0 = 1.3 * dx
        pa = get_particle_array_elastic_dynamics(
            name='solid', x=x, y=y, z=z, h=h0 * np.ones(x.size),
            m=rho0 * dx ** 3 * np.ones(x.size), rho=rho0 * np.ones(x.size),
            u=1e-2 * rng.uniform(-1, 1, x.size),
            constants=dict(E=E, nu=nu, rho_ref=rho0, n=4,
                           wdeltap=float(kernel.kernel(rij=dx, h=h0))))
        arrays = [pa]
        eqs = ElasticSolidsScheme(['solid'], [], dim=3).get_equations()
        algo_pair = 8.0 * (22 + 7)
        wname = ('Elastic solid block (Gray 2001 equation set, rings.py material), '
                 '%d^3 = %d particles, CubicSpline hdx 1.3, fp64' % (n1, x.size))
    pa = arrays[0]
    n_local = sum(a.get_number_of_particles() for a in arrays)

    for a in arrays:
        dev.attach(a, ctx).push()       # everything resident in HBM
    halo = None
    if world > 1 and args.workload == 'dam_break':
        from pysph_amd.parallel import SlabDecomposition
        halo = SlabDecomposition(arrays, ctx, rank, world, axis=0,
                                 width=kernel.radius_scale * 1.3 * dx,
                                 lo=slab_lo, hi=slab_hi, dist=dist)
    elif world > 1:
        from pysph_amd.parallel import SlabHalo
        halo = SlabHalo(pa, ctx, rank, world, axis=0,
                        width=kernel.radius_scale * 1.3 * dx,
                        lo=float(rank), hi=float(rank + 1), dist=dist)
    a_eval = AccelerationEval(arrays, eqs, kernel)
    SPHCompiler(a_eval, ctx=ctx, sync='manual').compile()
    nnps = HipNNPS(3, arrays, radius_scale=kernel.radius_scale, ctx=ctx,
                   sync=False, domain=domain)
    a_eval.set_nnps(nnps)
    if not args.no_reorder and domain is None:
        # what the reference's Solver does for its GPU backends before the first
        # step and every 50 steps (solver.py:296-302, application.py:1157-1161):
        # put the particles in cell order so gathers/scatters coalesce
        for i in range(len(arrays)):
            nnps.spatially_order_particles(i)
            nnps.update()

    def step():
        if halo is not None:
            halo.exchange()
        if domain is not None:
            nnps.update_domain()        # periodic ghosts are rebuilt every step
        nnps.update()
        a_eval.compute(0.0, 1e-5)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ctx.timer_enable(True)
    ctx.timer_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    ctx.timer_enable(False)
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    timers = {k: ctx.timer_get(k) for k in ('nnps', 'pack', 'eos', 'pair')}
    # true neighbour pairs of one evaluation (outside the timed region): the
    # flop-side reading of the pair loop that SURVEY 8(d) asks for next to the
    # HBM one
    pairs = None
    if args.workload == 'cube' and rank == 0:
        pairs = nnps.count_neighbors(0, 0)
    pair_ms, pair_launches = timers['pair']
    n_total = n_local * world          # real particles only (ghosts are extra work)
    if scaling == 'strong':
        tn = torch.tensor([float(n_local)], dtype=torch.float64, device='cuda')
        dist.all_reduce(tn, op=dist.ReduceOp.SUM)
        n_total = int(tn.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = n_total * args.steps / elapsed

    if rank == 0:
        # HBM bytes per launch of the dominant kernel come from a separate
        # rocprofv3 --pmc run of this same command (profiles/); only quoted when
        # the configuration matches the profiled one, else null
        traffic = None
        try:
            pt = json.load(open(os.path.join(REPO, 'profiles', 'pmc_traffic.json')))
            c = pt['config']
            if (c['n1'], c['variant'], c['spatially_ordered']) == \
                    (n1, args.variant, not args.no_reorder) and world == 1 \
                    and args.workload == 'cube':
                traffic = pt['bytes_per_launch']
        except Exception:
            traffic = None
        pair_avg_s = pair_ms / max(pair_launches, 1) * 1e-3
        # several pair launches per step for multi-destination sets: bytes of ONE
        # step / total pair-kernel time of one step
        pair_step_s = pair_ms / args.steps * 1e-3
        achieved = algo_pair * n_local / pair_step_s / 1e9 if pair_step_s > 0 else 0.0
        out = {
            'metric': 'particle-updates/sec (nnps.update + AccelerationEval.compute), '
                      'WCSPH 3D, fp64',
            'value': value, 'unit': 'particle-updates/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms_per_step, 'higher_is_better': True,
            'scaling': scaling, 'vs_baseline': None, 'dtype': 'f64',
            'data': 'synthetic',
            'config': {
                'workload': wname,
                'particles_per_gpu': n_local, 'pair_variant': args.variant,
                'spatially_ordered': not args.no_reorder,
                'parallelism': 'slab%d' % world if world > 1 else 'single',
            },
            'roofline': {
                'bound': 'hbm', 'kernel': 'k_pair_%s<FamWCSPH,WendlandQuintic>' %
                {0: 'direct', 2: 'wg', 3: 'agg', 6: 'lean'}[args.variant],
                'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                'algorithmic_bytes_per_particle': algo_pair,
                'avg_kernel_ms': pair_avg_s * 1e3,
            },
            'kernel_ms_per_step': {k: v[0] / args.steps for k, v in timers.items()},
            'fp64_valu': None if not pairs else {
                'pairs_per_launch': pairs,
                'flop_per_pair': FLOP_PER_PAIR,
                'achieved': pairs * FLOP_PER_PAIR / pair_step_s / 1e12,
                'peak': FP64_VECTOR_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                'frac': pairs * FLOP_PER_PAIR / pair_step_s / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                'Gpairs_per_s': pairs / pair_step_s / 1e9},
            'algorithmic_GBs_whole_update': ALGO_BYTES_UPDATE * value / 1e9,
        }
        if not args.no_cpu_baseline and world == 1:
            out['cpu_baseline'] = cpu_baseline(args.cpu_n1)
        return out
    return None


if __name__ == '__main__':
    main()
