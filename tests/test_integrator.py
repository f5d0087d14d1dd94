"""Integrator stage kernels + device-resident time stepping."""
import numpy as np
import pytest

from conftest import load_golden
from helpers import rel_err

STEP_PROPS = ['x', 'y', 'z', 'u', 'v', 'w', 'rho', 'x0', 'y0', 'z0', 'u0', 'v0',
              'w0', 'rho0', 'au', 'av', 'aw', 'ax', 'ay', 'az', 'arho', 'uhat',
              'vhat', 'what', 'auhat', 'avhat', 'awhat', 'vmag2',
              'e', 'ae', 'e0', 's00', 's01', 's02', 's11', 's12', 's22',
              'as00', 'as01', 'as02', 'as11', 'as12', 'as22',
              's000', 's010', 's020', 's110', 's120', 's220']


def _pa_from(g):
    from pysph_amd.particle_array import ParticleArray
    return ParticleArray(name='fluid', **{k: g['in/' + k].copy()
                                          for k in STEP_PROPS})


def test_oracle_steppers_match_reference_bitwise():
    """numpy stepper restatement vs the reference's WCSPHStep /
    TransportVelocityStep methods (golden): bit-exact."""
    from oracle import steppers as S
    g = load_golden('steppers.npz')
    dt = float(g['dt'])
    pa = _pa_from(g)
    S.wcsph_initialize(pa)
    for k in STEP_PROPS:
        assert np.array_equal(pa.properties[k], g['wcsph/initialize/' + k]), k
    S.wcsph_stage(pa, dt, 1)
    for k in STEP_PROPS:
        assert np.array_equal(pa.properties[k], g['wcsph/stage1/' + k]), k
    S.wcsph_stage(pa, dt, 2)
    for k in STEP_PROPS:
        assert np.array_equal(pa.properties[k], g['wcsph/stage2/' + k]), k
    pa = _pa_from(g)
    S.solid_initialize(pa)
    for k in STEP_PROPS:
        assert np.array_equal(pa.properties[k], g['solid/initialize/' + k]), k
    S.solid_stage(pa, dt, 1)
    for k in STEP_PROPS:
        assert np.array_equal(pa.properties[k], g['solid/stage1/' + k]), k
    S.solid_stage(pa, dt, 2)
    for k in STEP_PROPS:
        assert np.array_equal(pa.properties[k], g['solid/stage2/' + k]), k
    pa = _pa_from(g)
    S.tvf_stage1(pa, dt)
    for k in STEP_PROPS:
        assert np.array_equal(pa.properties[k], g['tvf/stage1/' + k]), k
    S.tvf_stage2(pa, dt)
    for k in STEP_PROPS:
        assert np.array_equal(pa.properties[k], g['tvf/stage2/' + k]), k


def test_integrator_api_surface():
    from pysph_amd.integrator import (Integrator, EPECIntegrator, WCSPHStep,
                                      TransportVelocityStep)
    with pytest.raises(ValueError):
        Integrator(fluid=object())
    i = EPECIntegrator(fluid=WCSPHStep(), solid=WCSPHStep())
    assert sorted(i.steppers) == ['fluid', 'solid']
    assert TransportVelocityStep()._kind == 2


@pytest.mark.gpu
def test_stage_kernels_match_reference_golden():
    """sph_integrate_stage vs the reference stepper methods (golden vectors).
    fp contraction (fma) may differ in the last bit: tolerance 4 ulp."""
    import ctypes as C
    from pysph_amd import device as dev
    g = load_golden('steppers.npz')
    dt = float(g['dt'])
    for kind, tag, stages in ((1, 'wcsph', [(0, 'initialize'), (1, 'stage1'), (2, 'stage2')]),
                              (2, 'tvf', [(1, 'stage1'), (2, 'stage2')]),
                              (3, 'solid', [(0, 'initialize'), (1, 'stage1'), (2, 'stage2')])):
        pa = _pa_from(g)
        ctx = dev.HipContext(0)
        gpu = dev.attach(pa, ctx)
        gpu.push()
        for st, name in stages:
            dev._check(ctx.lib.sph_integrate_stage(ctx._h, gpu.array_id, kind, st, dt))
            gpu.pull()
            for k in STEP_PROPS:
                ref = g['%s/%s/%s' % (tag, name, k)]
                assert np.allclose(pa.properties[k], ref, rtol=1e-15, atol=1e-16), (tag, name, k)


@pytest.mark.gpu
@pytest.mark.parametrize('sync', ['manual', 'auto'])
def test_epec_steps_device_resident_vs_oracle(oracle, sync):
    """Three EPEC steps of the WCSPH cube against the CPU oracle loop, with
    everything device-resident (sync='manual': one push, one pull) and with the
    host arrays authoritative (sync='auto': the Cython backend's behaviour --
    every evaluation and every stage moves its data); also the adaptive
    time-step reductions (integrator.py:161-200)."""
    from oracle import steppers as S
    from pysph_amd import device as dev
    from pysph_amd import kernels as K
    from pysph_amd.acceleration_eval import AccelerationEval, SPHCompiler
    from pysph_amd.integrator import EPECIntegrator, WCSPHStep, setup_integrator
    from pysph_amd.nnps import HipNNPS
    from test_hip_parity import make_cube, cube_equations, _copy_arrays
    pa, dx = make_cube(16)
    ref = _copy_arrays([pa])
    eqs = cube_equations(dx)
    kernel = K.WendlandQuintic(dim=3)
    dt = 0.25 * 1.3 * dx / 32.85
    ctx = dev.HipContext(0)
    dev.attach(pa, ctx).push()
    a_eval = AccelerationEval([pa], eqs, kernel)
    SPHCompiler(a_eval, ctx=ctx, sync=sync).compile()
    nnps = HipNNPS(3, [pa], radius_scale=2.0, ctx=ctx, sync=(sync == 'auto'))
    a_eval.set_nnps(nnps)
    integ = EPECIntegrator(fluid=WCSPHStep())
    setup_integrator(integ, a_eval, nnps)
    calls = []
    integ.set_post_stage_callback(lambda t, dt_, stage: calls.append((t, stage)))
    onn = oracle.OracleNNPS(3, ref, 2.0)
    oev = oracle.OracleEval(ref, eqs, kernel, nthreads=4)
    oev.set_nnps(onn)
    t = 0.0
    for _ in range(3):
        integ.step(t, dt)
        S.epec_step(ref, onn, oev, t, dt)
        t += dt
    assert len(calls) == 6 and calls[0][1] == 1 and calls[1][1] == 2
    assert abs(calls[0][0] - 0.5 * dt) < 1e-18
    if sync == 'manual':
        pa.gpu.pull()
    for prop in ('x', 'y', 'z', 'u', 'v', 'w', 'rho', 'au', 'av', 'aw', 'arho'):
        e = rel_err(pa.properties[prop], ref[0].properties[prop])
        assert e < 1e-10, (prop, e)
    # adaptive dt from device reductions == host formula on the oracle state
    got = integ.compute_time_step(dt, 0.3)
    r = ref[0]
    hmin = r.h.min()
    want = 0.3 * min(hmin / r.dt_cfl.max(),
                     np.sqrt(hmin / np.sqrt(r.dt_force.max())))
    assert abs(got - want) < 1e-9 * want


@pytest.mark.gpu
def test_epec_steps_two_slab_ranks_match_single_domain():
    """SURVEY.md 8(e) parity check: the same EPEC run on one context and on
    two slab ranks (two threads on one GPU, tests/helpers.ThreadDist standing in
    for torch.distributed) -- migration across the slab face, ghost refresh
    before every evaluation, re-balancing, all-reduced adaptive time step --
    ends in the same particle state, matched by global id."""
    import threading
    import torch
    from helpers import ThreadDist
    from pysph_amd import device as dev
    from pysph_amd import kernels as K
    from pysph_amd.acceleration_eval import AccelerationEval, SPHCompiler
    from pysph_amd.integrator import EPECIntegrator, WCSPHStep, setup_integrator
    from pysph_amd.nnps import HipNNPS
    from pysph_amd.parallel import HipParallelManager, SlabDecomposition
    from pysph_amd.particle_array import ParticleArray
    from test_hip_parity import make_cube, cube_equations
    full, dx = make_cube(16)
    n = full.get_number_of_particles()
    full.u[:] += 6.0               # drift in +x: particles cross the slab face
    full.add_property('e0', data=np.arange(n, dtype=np.float64))
    eqs = cube_equations(dx)
    kernel = K.WendlandQuintic(dim=3)
    dt = 0.25 * 1.3 * dx / 32.85
    nsteps = 6
    PROPS = ('x', 'y', 'z', 'u', 'v', 'w', 'rho', 'au', 'av', 'aw', 'arho')

    def run(pa, ctx, pm_factory=None):
        dev.attach(pa, ctx).push()
        a_eval = AccelerationEval([pa], eqs, kernel)
        SPHCompiler(a_eval, ctx=ctx, sync='manual').compile()
        nnps = HipNNPS(3, [pa], radius_scale=2.0, ctx=ctx, sync=False)
        a_eval.set_nnps(nnps)
        integ = EPECIntegrator(fluid=WCSPHStep())
        setup_integrator(integ, a_eval, nnps)
        pm = None
        if pm_factory:
            pm = pm_factory(pa, ctx)
            integ.set_parallel_manager(pm)
        t = 0.0
        for _ in range(nsteps):
            integ.step(t, dt)
            t += dt
        dtn = integ.compute_time_step(dt, 0.3)
        pa.gpu.managed = True
        pa.gpu.sync_host()
        return dict((k, pa.properties[k].copy()) for k in PROPS + ('e0',)), dtn, pm

    # single domain
    single = ParticleArray(name='fluid', **{k: v.copy() for k, v in full.properties.items()})
    ref, dt_ref, _ = run(single, dev.HipContext(0))

    hub = ThreadDist(2)
    results, errors = {}, []

    def rank_main(rank):
        try:
            ts = torch.cuda.Stream()
            with torch.cuda.stream(ts):
                x = full.x
                own = np.nonzero(x < 0.5)[0] if rank == 0 else np.nonzero(x >= 0.5)[0]
                pa = ParticleArray(name='fluid', **{k: v[own].copy() for k, v in full.properties.items()})
                ctx = dev.HipContext(0, ts.cuda_stream)
                lo, hi = (0.0, 0.5) if rank == 0 else (0.5, 1.0)

                def pmf(pa_, ctx_):
                    dec = SlabDecomposition([pa_], ctx_, rank, 2, axis=0,
                                            width=2.0 * 1.3 * dx * 1.05, lo=lo, hi=hi,
                                            dist=hub.view(rank))
                    return HipParallelManager(dec, rebalance_every=5)
                out, dtn, pm = run(pa, ctx, pmf)
                results[rank] = (out, dtn, pm.dec.halos[0].last_migrated, pm.count)
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
        t_.join(600)
    assert not errors, errors[0]
    order = np.argsort(ref['e0'])
    gids = []
    moved = 0
    for r in range(2):
        out, dtn, mig, count = results[r]
        gid = out['e0'].astype(np.int64)
        gids.append(gid)
        moved += sum(mig)
        assert count == 2 * nsteps                 # pm.update() before every evaluation
        assert abs(dtn - dt_ref) < 1e-9 * dt_ref   # all-reduced dt inputs
        for k in PROPS:
            e = rel_err(out[k], ref[k][order][gid])
            assert e < 1e-9, (r, k, e)
    assert (np.sort(np.concatenate(gids)) == np.arange(n)).all()
    # the drift really moved particles over the face at some point
    start_left = int((full.x < 0.5).sum())
    assert len(gids[0]) != start_left or moved > 0


@pytest.mark.gpu
@pytest.mark.parametrize('tight,lazy', [(False, 0), (True, 0), (False, 5)], ids=['roomy', 'tight', 'lazy-migration'])
def test_dam_break_epec_two_slab_ranks_padded_exchange_under_motion(monkeypatch, tight, lazy):
    """What BASELINE config 4 does, small: the THREE-array dam break (dx 0.06) cut in
    two slabs, 40 EPEC steps through HipParallelManager on the round-trip-free
    ('padded') exchange and the merged one-launch evaluation, with the fluid moving:
    the front's lattice plane crosses the slab face (rank 1 owns no fluid at the
    start and gets it by migration; its fluid ghost count for rank 0 jumps from
    nothing to a plane), a re-balance moves the face, and -- `tight` -- capacities
    without headroom make exactly that plane outgrow its message: verify() repeats
    the face the counted way and the evaluation runs again (the reference counts
    first and never evaluates on incomplete ghosts, parallel_manager.pyx:1085-1157;
    recipe of the comparison: parallel/tests/example_test_case.py:143-166).  `lazy`:
    ownership changes hands every fifth update only (HipParallelManager(migrate_every,
    margin): the ghost layers a quarter of the support wider, strays computed by their
    old rank meanwhile).  Same particle state as one domain, matched by global id."""
    import threading
    import torch
    from helpers import ThreadDist
    import pysph_amd.parallel as par
    from pysph_amd import device as dev
    from pysph_amd.acceleration_eval import AccelerationEval, SPHCompiler
    from pysph_amd.examples import dam_break_3d as db
    from pysph_amd.integrator import EPECIntegrator, WCSPHStep, setup_integrator
    from pysph_amd.nnps import HipNNPS
    from pysph_amd.particle_array import ParticleArray
    dx = 0.06
    full = db.create_particles(dx)
    g0 = 0
    for a in full:
        n = a.get_number_of_particles()
        a.add_property('e0', data=np.arange(g0, g0 + n, dtype=np.float64))
        g0 += n
    fl = full[0]
    fl.u[:] = 4.0                   # the column moves towards +x: its front plane (x = 1.2) crosses the face at 1.23
    eqs = db.create_scheme(dx).get_equations()
    kernel = db.create_kernel()
    dt = 0.125 * db.hdx * dx / (1.1 * db.c0)
    nsteps = 40
    cut = 1.23
    PROPS = ('x', 'y', 'z', 'u', 'v', 'w', 'rho', 'au', 'av', 'aw', 'arho')
    if tight:
        monkeypatch.setattr(par, '_capacity', lambda c: ((c + 8 + 7) // 8) * 8)
        monkeypatch.setattr(par, '_capacity_tight', lambda c: ((c + 8 + 7) // 8) * 8)

    def run(arrays, ctx, pm_factory=None):
        for a in arrays:
            dev.attach(a, ctx).push()
        a_eval = AccelerationEval(arrays, eqs, kernel)
        SPHCompiler(a_eval, ctx=ctx, sync='manual').compile()
        nnps = HipNNPS(3, arrays, radius_scale=kernel.radius_scale, ctx=ctx, sync=False)
        a_eval.set_nnps(nnps)
        integ = EPECIntegrator(fluid=WCSPHStep())
        setup_integrator(integ, a_eval, nnps)
        pm = None
        if pm_factory:
            pm = pm_factory(arrays, ctx)
            integ.set_parallel_manager(pm)
        t = 0.0
        for _ in range(nsteps):
            integ.step(t, dt)
            t += dt
        out = {}
        for a in arrays:
            a.gpu.managed = True
            a.gpu.sync_host()
            nr = a.get_number_of_particles(True)
            out[a.name] = dict((k, a.properties[k][:nr].copy()) for k in PROPS + ('e0',))
        return out, pm, ctx.timer_get('n_merged')[1]

    def copy_of(a, idx=None):
        props = {k: (v.copy() if idx is None else v[idx].copy()) for k, v in a.properties.items()}
        return ParticleArray(name=a.name, **props)

    ref, _, _ = run([copy_of(a) for a in full], dev.HipContext(0))
    hub = ThreadDist(2)
    results, errors = {}, []

    def rank_main(rank):
        try:
            ts = torch.cuda.Stream()
            with torch.cuda.stream(ts):
                arrays = [copy_of(a, np.nonzero(a.x < cut)[0] if rank == 0 else np.nonzero(a.x >= cut)[0]) for a in full]
                ctx = dev.HipContext(0, ts.cuda_stream)
                ctx.timer_enable(True)
                lo, hi = (-1e30, cut) if rank == 0 else (cut, 1e30)

                def pmf(arrs, ctx_):
                    support = 2.0 * db.hdx * dx
                    margin = (0.25 if lazy else 0.05) * support
                    dec = par.SlabDecomposition(arrs, ctx_, rank, 2, axis=0, width=support + margin,
                                                lo=lo, hi=hi, dist=hub.view(rank), protocol='padded')
                    return par.HipParallelManager(dec, rebalance_every=31, migrate_every=lazy or 1, margin=margin)
                out, pm, n_merged = run(arrays, ctx, pmf)
                hs = pm.dec.halos
                results[rank] = (out, [h.padded_exchanges for h in hs], [h.repaired_exchanges for h in hs],
                                 [h.last_migrated for h in hs], pm.count, n_merged, (hs[0].lo, hs[0].hi),
                                 [list(h.ops.props) for h in hs], pm.max_excursion / (2.0 * db.hdx * dx))
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
        t_.join(900)
    assert not errors, errors[0]
    fluid_on_1 = results[1][0]['fluid']['e0'].size
    assert fluid_on_1 > 0                            # the front crossed the face: rank 1 owns fluid now
    for name in ('fluid', 'boundary', 'obstacle'):
        order = np.argsort(ref[name]['e0'])
        gids = []
        for r in range(2):
            out = results[r][0][name]
            gid = out['e0'].astype(np.int64) - int(ref[name]['e0'].min())
            gids.append(gid)
            for k in PROPS:
                e = rel_err(out[k], ref[name][k][order][gid], scale=max(np.abs(ref[name][k]).max(), 1e-300))
                assert e < 1e-9, (name, r, k, e)
        assert (np.sort(np.concatenate(gids)) == np.arange(ref[name]['e0'].size)).all(), name
    for r in range(2):
        _, padded, repaired, _, count, n_merged, faces, props, strayed = results[r]
        assert count == 2 * nsteps
        assert min(padded) >= 2 * nsteps - 8, padded          # the steady state IS the round-trip-free exchange
        assert n_merged >= 2 * nsteps - 8                     # ... on the merged one-launch evaluation
        assert all(dev.prop_id('h') not in p and dev.prop_id('m') not in p for p in props)   # promised h, m do not travel
        if tight:
            assert sum(repaired) >= 1, repaired               # the plane that crossed outgrew its message: repaired
        else:
            assert sum(repaired) == 0, repaired
    assert results[0][6][1] != cut                            # the re-balance moved the face
    if lazy:
        assert 0.0 < max(results[r][8] for r in range(2)) < 0.25   # leavers had strayed, by less than the margin


@pytest.mark.gpu
def test_dam_break_time_loop_runs_and_stays_physical():
    """The example's device-resident time loop (EPEC + adaptive dt + periodic
    reordering) on the config-1 geometry at a coarse spacing: the column
    starts to collapse (front advances, fluid accelerates downwards), nothing
    blows up, the boundary does not move, dt follows the CFL/force limits."""
    from pysph_amd.examples import dam_break_3d as db
    arrays, st = db.run(dx=0.06, n_steps=40, reorder_freq=10)
    fluid = [a for a in arrays if a.name == 'fluid'][0]
    wall = [a for a in arrays if a.name == 'boundary'][0]
    ref = db.create_particles(0.06)
    assert st['steps'] == 40 and st['t'] > 0
    assert np.isfinite(fluid.x).all() and np.isfinite(fluid.rho).all()
    assert fluid.x.max() > ref[0].x.max()                 # front moved towards the obstacle
    assert fluid.w.mean() < 0                              # collapsing under gravity
    assert abs(fluid.rho / 1000.0 - 1).max() < 0.05        # weakly compressible
    assert all(0 < d < 1e-2 for d in st['dts'])
    # boundary particles have no stepper: the same set of positions as at t = 0
    assert np.array_equal(np.sort(wall.x), np.sort(ref[1].x))
    assert sorted(fluid.properties) == sorted(ref[0].properties)


@pytest.mark.gpu
def test_rings_collision_time_loop():
    """BASELINE config 5's problem (colliding elastic rings, rings.py) through
    the elastic equation set + SolidMechStep, device-resident PEC steps: the
    rings approach, linear momentum stays zero by symmetry, the stress state
    builds up only after contact, nothing blows up."""
    from pysph_amd.examples import rings
    dx = 0.001
    ref = rings.create_particles(dx)[0]
    arrays, st = rings.run(dx=dx, n_steps=150, dt=2e-8, reorder_freq=50)
    pa = arrays[0]
    assert st['steps'] == 150 and pa.get_number_of_particles() == ref.get_number_of_particles()
    for k in ('x', 'y', 'u', 'v', 'rho', 's00', 's01', 's11', 'p'):
        assert np.isfinite(pa.properties[k]).all(), k
    left, right = pa.x < np.median(pa.x), pa.x >= np.median(pa.x)
    gap0 = ref.x[ref.x >= np.median(ref.x)].min() - ref.x[ref.x < np.median(ref.x)].max()
    gap1 = pa.x[right].min() - pa.x[left].max()
    assert gap1 < gap0                                             # approaching
    mom = (pa.m * pa.u).sum() / (pa.m * np.abs(pa.u)).sum()
    assert abs(mom) < 1e-9                                         # mirror-symmetric set-up
    assert abs(pa.rho - 1.0).max() < 0.2
    assert abs((pa.m * pa.v).sum()) < 1e-9 * (pa.m * np.abs(pa.u)).sum()


@pytest.mark.gpu
def test_taylor_green_decay_time_loop():
    """The reference's Taylor-Green example (TVF, Re = 100, periodic box) with
    PEC + TransportVelocityStep and device-resident periodic images: the
    maximum velocity follows the exact decay exp(-8 pi^2 nu t) within a few
    percent over 400 steps at 100 x 100 (at 200 x 200 it is 0.03 %) (the error measure of taylor_green.py:38-50), the
    particles stay inside the box, density stays near rho0."""
    from pysph_amd.examples import taylor_green as tg
    arrays, st = tg.run(nx=100, n_steps=400)
    pa = arrays[0]
    n = pa.get_number_of_particles(True)
    assert n == 10000 and st['t'] > 0.05
    vmax = np.sqrt(pa.u[:n] ** 2 + pa.v[:n] ** 2).max()
    exact = tg.U * np.exp(-8 * np.pi ** 2 * st['nu'] * st['t'])
    assert abs(vmax - exact) < 0.02 * exact, (vmax, exact)
    assert (pa.x[:n] >= 0).all() and (pa.x[:n] <= 1).all()
    assert (pa.y[:n] >= 0).all() and (pa.y[:n] <= 1).all()
    assert abs(pa.rho[:n] - 1).max() < 0.05
