"""What a migration costs a stepping slab rank (one GPU, the slab as its own periodic neighbour over RCCL): EPEC time steps
of the 1 M cube with HipParallelManager(migrate_every=K) -- K = 1 is the reference's order (migrate before every
evaluation, parallel_manager.pyx:512-530), K > 1 the lazy migration of DESIGN.md section 6."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch
import torch.distributed as dist
import bench
from pysph_amd import device as dev
from pysph_amd import kernels as K
from pysph_amd.acceleration_eval import AccelerationEval, SPHCompiler
from pysph_amd.integrator import EPECIntegrator, WCSPHStep, setup_integrator
from pysph_amd.nnps import HipNNPS
from pysph_amd.parallel import HipParallelManager, SlabDecomposition, SphCommTransport

os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29655')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for every in ([int(os.environ['ONLY_EVERY'])] if os.environ.get('ONLY_EVERY') else (1, 4, 16)):
    ts = torch.cuda.Stream(); torch.cuda.set_stream(ts)
    ctx = dev.HipContext(0, ts.cuda_stream)
    pa, dx = bench.make_cube(n1)
    eqs = bench.cube_equations(dx)
    kernel = K.WendlandQuintic(dim=3)
    dev.attach(pa, ctx).push()
    a_eval = AccelerationEval([pa], eqs, kernel)
    SPHCompiler(a_eval, ctx=ctx, sync='manual').compile()
    nnps = HipNNPS(3, [pa], radius_scale=2.0, ctx=ctx, sync=False)
    a_eval.set_nnps(nnps)
    integ = EPECIntegrator(fluid=WCSPHStep())
    setup_integrator(integ, a_eval, nnps)
    support = 2.0 * 1.3 * dx
    margin = float(os.environ.get('MARGIN', '0.3')) * support
    tr = SphCommTransport(ctx, dist, 0, 1)
    dec = SlabDecomposition([pa], ctx, 0, 1, axis=0, width=support + margin, lo=0.0, hi=1.0, periodic=True, period=1.0,
                            dist=tr, protocol='padded')
    pm = HipParallelManager(dec, migrate_every=every, margin=margin)
    integ.set_parallel_manager(pm)
    nnps.spatially_order_particles(0)
    dt = 0.125 * 1.3 * dx / 32.85
    t = 0.0
    for _ in range(8):
        integ.step(t, dt); t += dt
        if os.environ.get('VERBOSE'): print('  warm-up step: strayed', pm.max_excursion / support, 'migrated', dec.halos[0].last_migrated, flush=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nsteps = 48
    for _ in range(nsteps):
        integ.step(t, dt); t += dt
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    h = dec.halos[0]
    print('migrate_every %2d: %.3f ms per EPEC time step (2 evaluations + 3 sweeps), migrated %d, strayed at most %.3f of the support, '
          'padded exchanges %d, repaired %d' % (every, el / nsteps * 1e3, h.total_migrated, pm.max_excursion / support,
                                                 h.padded_exchanges, h.repaired_exchanges), flush=True)
    del integ, a_eval, nnps, pm, dec
    ctx.close()
dist.destroy_process_group()
