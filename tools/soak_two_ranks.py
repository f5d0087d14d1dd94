"""Soak of the stepping slab path (round 6): the three-array dam break under gravity on TWO thread-ranks of one GPU, EPEC
steps through HipParallelManager on the padded exchange with lazy migration and periodic re-balancing -- hundreds of steps,
the column collapsing through the face.  No reference run (summation order amplifies over hundreds of steps): the
invariants instead -- every global id owned exactly once, nothing non-finite, the counters of what the protocol did.
    python tools/soak_two_ranks.py [dx] [steps] [migrate_every] [rebalance_every]      (SOAK_TIGHT=1: capacities without headroom)"""
import os, sys, threading, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'tests'))
import numpy as np
import torch
from helpers import ThreadDist
import pysph_amd.parallel as par
from pysph_amd import device as dev
from pysph_amd.acceleration_eval import AccelerationEval, SPHCompiler
from pysph_amd.examples import dam_break_3d as db
from pysph_amd.integrator import EPECIntegrator, WCSPHStep, setup_integrator
from pysph_amd.nnps import HipNNPS
from pysph_amd.particle_array import ParticleArray

dx = float(sys.argv[1]) if len(sys.argv) > 1 else 0.04
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
lazy = int(sys.argv[3]) if len(sys.argv) > 3 else 4
reb = int(sys.argv[4]) if len(sys.argv) > 4 else 50
if os.environ.get('SOAK_TIGHT') == '1':
    # capacities without headroom: every growth of a face's count outgrows its message and is repaired by verify()
    par._capacity = par._capacity_tight = lambda c: ((c + 8 + 7) // 8) * 8
full = db.create_particles(dx)
g0 = 0
for a in full:
    n = a.get_number_of_particles()
    a.add_property('e0', data=np.arange(g0, g0 + n, dtype=np.float64))
    g0 += n
eqs = db.create_scheme(dx).get_equations()
kernel = db.create_kernel()
dt = 0.125 * db.hdx * dx / (1.1 * db.c0)
allx = np.concatenate([a.x for a in full])
cut = float(np.median(allx))
hub = ThreadDist(2)
results, errors = {}, []


def copy_of(a, idx):
    return ParticleArray(name=a.name, **{k: v[idx].copy() for k, v in a.properties.items()})


def rank_main(rank):
    try:
        ts = torch.cuda.Stream()
        with torch.cuda.stream(ts):
            arrays = [copy_of(a, np.nonzero(a.x < cut)[0] if rank == 0 else np.nonzero(a.x >= cut)[0]) for a in full]
            ctx = dev.HipContext(0, ts.cuda_stream)
            ctx.timer_enable(True)
            for a in arrays:
                dev.attach(a, ctx).push()
            a_eval = AccelerationEval(arrays, eqs, kernel)
            SPHCompiler(a_eval, ctx=ctx, sync='manual').compile()
            nnps = HipNNPS(3, arrays, radius_scale=kernel.radius_scale, ctx=ctx, sync=False)
            a_eval.set_nnps(nnps)
            integ = EPECIntegrator(fluid=WCSPHStep())
            setup_integrator(integ, a_eval, nnps)
            support = 2.0 * db.hdx * dx
            margin = 0.25 * support
            lo, hi = (-1e30, cut) if rank == 0 else (cut, 1e30)
            dec = par.SlabDecomposition(arrays, ctx, rank, 2, axis=0, width=support + margin, lo=lo, hi=hi,
                                        dist=hub.view(rank), protocol='padded')
            pm = par.HipParallelManager(dec, rebalance_every=reb, migrate_every=lazy, margin=margin)
            integ.set_parallel_manager(pm)
            t = 0.0
            t0 = time.perf_counter()
            for k in range(nsteps):
                integ.step(t, dt)
                t += dt
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            out = {}
            for a in arrays:
                a.gpu.managed = True
                a.gpu.sync_host()
                nr = a.get_number_of_particles(True)
                out[a.name] = dict((p, a.properties[p][:nr].copy()) for p in ('x', 'y', 'z', 'u', 'v', 'w', 'rho', 'e0'))
            hs = dec.halos
            results[rank] = dict(out=out, ms=el / nsteps * 1e3, padded=[h.padded_exchanges for h in hs],
                                 repaired=[h.repaired_exchanges for h in hs], migrated=[h.total_migrated for h in hs],
                                 faces=(hs[0].lo, hs[0].hi), strayed=pm.max_excursion / support,
                                 n_merged=ctx.timer_get('n_merged')[1], n_async=ctx.timer_get('n_async')[1])
    except Exception:
        import traceback
        errors.append(traceback.format_exc())
        try:
            hub.barrier.abort()
        except Exception:
            pass


threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(2)]
for t_ in threads:
    t_.start()
for t_ in threads:
    t_.join(3000)
if errors:
    print(errors[0]); sys.exit(1)
ok = True
for name in ('fluid', 'boundary', 'obstacle'):
    gids = np.concatenate([results[r]['out'][name]['e0'] for r in range(2)]).astype(np.int64)
    ref = next(a for a in full if a.name == name).e0.astype(np.int64)
    once = gids.size == ref.size and (np.sort(gids) == np.sort(ref)).all()
    finite = all(np.isfinite(results[r]['out'][name][p]).all() for r in range(2) for p in ('x', 'u', 'rho'))
    print('%-9s %7d particles, every id owned once: %s, finite: %s, per rank %s' % (
        name, ref.size, once, finite, [results[r]['out'][name]['e0'].size for r in range(2)]))
    ok = ok and once and finite
fl = [results[r]['out']['fluid'] for r in range(2)]
xs = np.concatenate([f['x'] for f in fl]); us = np.concatenate([f['u'] for f in fl]); zs = np.concatenate([f['z'] for f in fl])
print('fluid after %d steps (t = %.4f): x in [%.3f, %.3f], z in [%.3f, %.3f], max |u| %.3f' % (
    nsteps, nsteps * dt, xs.min(), xs.max(), zs.min(), zs.max(), np.abs(us).max()))
for r in range(2):
    d = results[r]
    print('rank %d: %.3f ms per EPEC step, faces %s, padded exchanges %s, repaired %s, migrated %s, strayed %.3f of the support, '
          'merged evaluations %d, lagged updates %d' % (r, d['ms'], d['faces'], d['padded'], d['repaired'], d['migrated'], d['strayed'],
                                                       d['n_merged'], d['n_async']))
print('SOAK', 'OK' if ok else 'FAILED')
sys.exit(0 if ok else 1)
