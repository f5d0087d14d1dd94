"""GPU parity tests: libsphhip (through the C-ABI / ctypes) against
  (a) the golden vectors produced by the reference's own Python classes, and
  (b) the C oracle on seeded inputs at sizes it finishes in seconds.

Bar (BASELINE.json north_star): fp64 results within 1e-10 relative of the
reference.  `rel_err` scales by max|field| (mixed abs/rel), see helpers.py.
Neighbour SETS must be identical (integer work: bit-exact).
"""
import os

import numpy as np
import pytest

from conftest import load_golden, arrays_from_golden
from helpers import golden_case, rel_err, WC_OUT

pytestmark = pytest.mark.gpu

TOL = 1e-10
CASES = ['sd_1d_line', 'wcsph_cube_varh', 'tvf_cube', 'wcsph_dam_dx0.1',
         'elastic_2d', 'elastic_3d']


def make_eval(arrays, eqs, kernel, dim, variant=6, sync='auto'):
    from pysph_amd import device as dev
    from pysph_amd.acceleration_eval import AccelerationEval, SPHCompiler
    from pysph_amd.nnps import HipNNPS
    ctx = dev.HipContext(0)
    ctx.set_option('pair_variant', variant)
    a_eval = AccelerationEval(arrays, eqs, kernel)
    SPHCompiler(a_eval, ctx=ctx, sync=sync).compile()
    nnps = HipNNPS(dim, arrays, radius_scale=kernel.radius_scale, ctx=ctx)
    a_eval.set_nnps(nnps)
    return a_eval, nnps, ctx


@pytest.mark.parametrize('variant', [0, 6])
@pytest.mark.parametrize('case', CASES)
def test_golden_parity(case, variant):
    g = load_golden(case + '.npz')
    arrays = arrays_from_golden(g, 'in')
    eqs, kernel, dim, outs = golden_case(case, g)
    a_eval, nnps, ctx = make_eval(arrays, eqs, kernel, dim, variant)
    # grid scalars are computed in the reference's arithmetic: exact
    assert nnps.cell_size == float(g['nnps/cell_size'])
    assert np.array_equal(nnps.xmin, g['nnps/xmin'])
    assert np.array_equal(nnps.xmax, g['nnps/xmax'])
    assert np.array_equal(nnps.ncells_per_dim, g['nnps/ncells_per_dim'])
    assert nnps.n_cells == int(g['nnps/n_cells'])
    a_eval.compute(float(g['t']), float(g['dt']))
    worst = 0.0
    for pa in arrays:
        for prop in outs:
            key = 'out/%s/%s' % (pa.name, prop)
            if key in g.files and prop in pa.properties:
                e = rel_err(pa.properties[prop], g[key])
                worst = max(worst, e)
                assert e < TOL, (case, pa.name, prop, e)
    print('golden %s variant %d: max rel err %.3e' % (case, variant, worst))


@pytest.mark.parametrize('case,counter', [('wcsph_dam_dx0.1', 'n_merged'), ('tvf_cube', 'n_mass_fused'),
                                          ('elastic_2d', 'n_mass_fused'), ('elastic_3d', 'n_mass_fused')])
def test_golden_parity_on_fused_records(case, counter):
    """The record layouts that need a promise or a look at the masses -- the merged
    order of a dam break's three arrays, TVF without p and V, the elastic rates
    without h and m -- only engage on device-resident state from the SECOND
    evaluation on (the first neighbour update has not looked at the masses yet; a
    host push of m forgets them): run two evaluations and compare the second with
    the reference's golden vectors.  The equations are idempotent on their inputs
    (EOS, density and stress groups recompute what the rate groups read)."""
    g = load_golden(case + '.npz')
    arrays = arrays_from_golden(g, 'in')
    eqs, kernel, dim, outs = golden_case(case, g)
    a_eval, nnps, ctx = make_eval(arrays, eqs, kernel, dim, 6, sync='manual')
    for pa in arrays:
        pa.gpu.push()
    nnps.sync = False
    nnps.update()          # the push of x, y, z, h invalidated the grid
    a_eval.compute(float(g['t']), float(g['dt']))
    before = ctx.timer_get(counter)[1]
    nnps.update()
    a_eval.compute(float(g['t']), float(g['dt']))
    assert ctx.timer_get(counter)[1] > before, 'the fused-record path did not run'
    a_eval.c_acceleration_eval.pull_outputs()
    worst = 0.0
    for pa in arrays:
        for prop in outs:
            key = 'out/%s/%s' % (pa.name, prop)
            if key in g.files and prop in pa.properties:
                e = rel_err(pa.properties[prop], g[key])
                worst = max(worst, e)
                assert e < TOL, (case, pa.name, prop, e)
    print('golden %s on fused records: max rel err %.3e' % (case, worst))


@pytest.mark.parametrize('which', ['product', 'restatement'])
def test_generated_wall_equations_match_reference(which):
    """TVF with solid walls: SetWallVelocity, SolidWallPressureBC and
    SolidWallNoSlipBC have no hand-written kernel -- their Python bodies
    (the product's pysph_amd/wall_bc.py, and the statement-by-statement
    restatement of the reference's in tests/wall_equations_fixture.py) are
    translated by pysph_amd.codegen, compiled for gfx950 and run through
    sph_eval_generated, next to the hand-written TVF momentum kernels on the same
    destination.  Golden = the reference's own TVFScheme(fluids, solids) classes
    executed by oracle/ref_driver.py."""
    g = load_golden('tvf_wall.npz')
    arrays = arrays_from_golden(g, 'in')
    we = None
    if which == 'restatement':
        import wall_equations_fixture as we
    eqs, kernel, dim, outs = golden_case('tvf_wall', g, wall_equations=we)
    a_eval, nnps, ctx = make_eval(arrays, eqs, kernel, dim, 6)
    a_eval.compute(float(g['t']), float(g['dt']))
    worst, checked = 0.0, 0
    for pa in arrays:
        for prop in outs:
            key = 'out/%s/%s' % (pa.name, prop)
            if key in g.files and prop in pa.properties:
                e = rel_err(pa.properties[prop], g[key])
                worst = max(worst, e)
                checked += 1
                assert e < TOL, (pa.name, prop, e)
    wall = [pa for pa in arrays if pa.name == 'wall'][0]
    assert (wall.wij == 0).any() and (wall.wij > 0).any()   # both branches of post_loop
    assert checked >= 18
    print('tvf_wall (generated families): max rel err %.3e' % worst)


def _custom_setup(varh, seed=5):
    from pysph_amd.equations import Group
    from pysph_amd.particle_array import get_particle_array
    from custom_equations import KitchenSink, PowerLawState, WallPush
    rng = np.random.default_rng(seed)
    n1 = 9
    dx = 1.0 / n1
    g = (np.arange(n1) + 0.5) * dx
    x, y, z = [a.ravel() for a in np.meshgrid(g, g, g, indexing='ij')]
    wall = y < 2.5 * dx
    arrays = []
    for name, msk in (('fluid', ~wall), ('solid', wall)):
        n = int(msk.sum())
        pa = get_particle_array(
            name=name, constants=dict(coef=np.array([1.25, -0.5])),
            x=x[msk] + 0.1 * dx * rng.uniform(-1, 1, n),
            y=y[msk] + 0.1 * dx * rng.uniform(-1, 1, n),
            z=z[msk] + 0.1 * dx * rng.uniform(-1, 1, n),
            u=rng.uniform(-1, 1, n), v=rng.uniform(-1, 1, n), w=rng.uniform(-1, 1, n),
            h=1.3 * dx * (1 + varh * rng.uniform(-1, 1, n)),
            m=dx ** 3 * np.ones(n), rho=1 + 0.1 * rng.uniform(-1, 1, n),
            additional_props=['q', 'gx', 'gy', 'gz', 'e'])
        for k in ('q', 'gx', 'gy', 'gz', 'e', 'p'):
            pa.properties[k][:] = rng.uniform(1, 2, n)     # junk the equations must overwrite
        arrays.append(pa)
    eqs = [
        Group(real=False, equations=[PowerLawState('fluid', None, k=1.5, n=1.4),
                                     PowerLawState('solid', None, k=0.5, n=2.0)]),
        Group(equations=[
            KitchenSink('fluid', ['fluid', 'solid'], a=0.3, b=0.05, flag=True),
            WallPush('fluid', ['solid'], c=0.7),
            KitchenSink('solid', ['fluid'], a=0.1, b=2.0, flag=False)]),
    ]
    return arrays, eqs


@pytest.mark.parametrize('varh', [0.0, 0.15])
@pytest.mark.parametrize('kname', ['CubicSpline', 'Gaussian'])
def test_generated_custom_equations_vs_python(oracle, varh, kname):
    """Arbitrary user equations (tests/custom_equations.py) through the
    generated-family path vs the same Python bodies executed by
    oracle/py_eval.py: no-source equations with an array constant, two
    equations with different source lists on one destination, WI/WJ/DWI/DWJ
    with per-particle h (and the uniform-h specialisation), local arrays,
    loops, early return, elif chains."""
    from oracle.py_eval import PyEval
    from pysph_amd import kernels as K
    arrays, eqs = _custom_setup(varh)
    ref = _copy_arrays(arrays)
    for r, a in zip(ref, arrays):
        r.constants = dict((k, v.copy()) for k, v in a.constants.items())
    kernel = getattr(K, kname)(dim=3)
    t, dt = 0.25, 1e-3
    a_eval, nnps, ctx = make_eval(arrays, eqs, kernel, 3)
    a_eval.compute(t, dt)
    onn = oracle.OracleNNPS(3, ref, radius_scale=kernel.radius_scale)
    onn.update()
    PyEval(ref, eqs, kernel, onn).compute(t, dt)
    worst = 0.0
    for pa, pr in zip(arrays, ref):
        for prop in ('p', 'e', 'q', 'gx', 'gy', 'gz'):
            e = rel_err(pa.properties[prop], pr.properties[prop])
            worst = max(worst, e)
            assert e < TOL, (pa.name, prop, e)
    print('custom equations %s varh=%g: max rel err %.3e' % (kname, varh, worst))


@pytest.mark.parametrize('tensile', [False, True])
def test_generated_wcsph_matches_handwritten_and_oracle(oracle, tensile):
    """Continuity + Momentum (+tensile correction) + XSPH written as Python
    bodies (tests/custom_equations.py) and pushed through the translator, vs
    the hand-written FamWCSPH kernel and the C oracle on the same 110 k-particle
    cube: the generated family sits on the same skeleton, so it must agree
    with both to rounding -- and its run time is printed next to the
    hand-written one."""
    import time
    from custom_equations import PyContinuity, PyMomentum, PyXSPH
    from pysph_amd.equations import Group, TaitEOS
    pa, dx = make_cube(48)
    kernel = K_Wendland()
    c0 = 32.85
    hand = cube_equations(dx) if not tensile else None
    from pysph_amd.equations import (ContinuityEquation, MomentumEquation,
                                     XSPHCorrection)
    kw = dict(c0=c0, alpha=0.25, beta=0.1, gz=-9.81, tensile_correction=tensile)
    eos = Group(real=False, equations=[TaitEOS(dest='fluid', sources=None, rho0=1000.0,
                                                c0=c0, gamma=7.0)])
    hand = [eos, Group(equations=[
        ContinuityEquation(dest='fluid', sources=['fluid']),
        MomentumEquation(dest='fluid', sources=['fluid'], **kw),
        XSPHCorrection(dest='fluid', sources=['fluid'], eps=0.5)])]
    gen = [eos, Group(equations=[
        PyContinuity(dest='fluid', sources=['fluid']),
        PyMomentum(dest='fluid', sources=['fluid'], **kw),
        PyXSPH(dest='fluid', sources=['fluid'], eps=0.5)])]
    out = {}
    for tag, eqs in (('hand', hand), ('gen', gen)):
        q = _copy_arrays([pa])
        a_eval, nnps, ctx = make_eval(q, eqs, kernel, 3, 6, sync='manual')
        q[0].gpu.push()
        nnps.update()
        a_eval.compute(0.0, 1e-5)            # warm-up (and the build of the family)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            a_eval.compute(0.0, 1e-5)
        ctx.synchronize()
        out[tag + '_ms'] = (time.perf_counter() - t0) / 5 * 1e3
        q[0].gpu.pull()
        out[tag] = q[0]
    ref = _copy_arrays([pa])
    onn = oracle.OracleNNPS(3, ref, 2.0)
    onn.update()
    oev = oracle.OracleEval(ref, hand, kernel, nthreads=8)
    oev.set_nnps(onn)
    oev.compute(0.0, 1e-5)
    for prop in WC_OUT:
        e1 = rel_err(out['gen'].properties[prop], out['hand'].properties[prop])
        e2 = rel_err(out['gen'].properties[prop], ref[0].properties[prop])
        assert e1 < TOL and e2 < TOL, (prop, e1, e2)
    print('WCSPH 110k, tensile=%s: hand-written %.3f ms, generated %.3f ms per compute'
          % (tensile, out['hand_ms'], out['gen_ms']))


def K_Wendland():
    from pysph_amd import kernels as K
    return K.WendlandQuintic(dim=3)


def test_generated_strided_properties_vs_python(oracle):
    """Multi-component properties (ParticleArray stride 3, indexed
    d_g3[d_idx*3 + k]) in generated families: every component travels as its
    own device property, the helper splits / re-interleaves the host array.
    Two groups: the first fills g3, the second reads it for destination AND
    source (the delta-SPH pattern of wc/basic.py:355-414)."""
    from oracle.py_eval import PyEval
    from custom_equations import StridedDiffusion, StridedGradient
    from pysph_amd import kernels as K
    from pysph_amd.equations import Group
    from pysph_amd.particle_array import get_particle_array_wcsph

    def build():
        rng = np.random.default_rng(9)
        n1 = 10
        dx = 1.0 / n1
        g = (np.arange(n1) + 0.5) * dx
        x, y, z = [a.ravel() for a in np.meshgrid(g, g, g, indexing='ij')]
        n = x.size
        pa = get_particle_array_wcsph(
            name='fluid', x=x + 0.1 * dx * rng.uniform(-1, 1, n),
            y=y + 0.1 * dx * rng.uniform(-1, 1, n), z=z + 0.1 * dx * rng.uniform(-1, 1, n),
            h=1.2 * dx * np.ones(n), m=dx ** 3 * np.ones(n),
            rho=1 + 0.1 * rng.uniform(-1, 1, n))
        pa.add_property('g3', stride=3)
        pa.g3[:] = rng.uniform(-1, 1, 3 * n)           # junk the first group must overwrite
        pa.arho[:] = 7.0
        return pa
    eqs = [Group(equations=[StridedGradient('fluid', ['fluid'])], real=False),
           Group(equations=[StridedDiffusion('fluid', ['fluid'], delta=0.1, c0=10.0)])]
    kernel = K.CubicSpline(dim=3)
    pa, ref = build(), build()
    a_eval, nnps, ctx = make_eval([pa], eqs, kernel, 3)
    a_eval.compute(0.0, 1e-4)
    onn = oracle.OracleNNPS(3, [ref], radius_scale=2.0)
    onn.update()
    PyEval([ref], eqs, kernel, onn).compute(0.0, 1e-4)
    assert rel_err(pa.g3, ref.g3) < TOL
    assert rel_err(pa.arho, ref.arho) < TOL
    assert np.abs(ref.g3).max() > 0 and np.abs(ref.arho).max() > 0


def test_generated_split_initialize_in_pair_mode(oracle):
    """initialize() writes qtmp, loop() reads s_qtmp: the packed source records
    must hold the values AFTER initialize ran for every particle (the reference
    finishes initialize first, mako :36-47), so the family launches initialize
    on its own before packing."""
    from oracle.py_eval import PyEval
    from custom_equations import SmoothCopy
    from pysph_amd import kernels as K
    from pysph_amd.acceleration_eval import _CGroup
    from pysph_amd.equations import Group
    from pysph_amd.particle_array import get_particle_array_wcsph

    def build():
        rng = np.random.default_rng(4)
        n1 = 10
        dx = 1.0 / n1
        g = (np.arange(n1) + 0.5) * dx
        x, y, z = [a.ravel() for a in np.meshgrid(g, g, g, indexing='ij')]
        n = x.size
        pa = get_particle_array_wcsph(name='fluid', x=x, y=y, z=z, h=1.3 * dx * np.ones(n),
                                      m=dx ** 3 * np.ones(n), rho=np.ones(n))
        pa.add_property('q')
        pa.add_property('qtmp')
        pa.q[:] = rng.uniform(0, 1, n)
        pa.qtmp[:] = 99.0
        return pa
    eqs = [Group(equations=[SmoothCopy('fluid', ['fluid'])])]
    kernel = K.CubicSpline(dim=3)
    pa, ref = build(), build()
    a_eval, nnps, ctx = make_eval([pa], eqs, kernel, 3)
    assert a_eval.c_acceleration_eval.plan[0][1].units[0].fam.split_init
    a_eval.compute(0.0, 1e-4)
    onn = oracle.OracleNNPS(3, [ref], radius_scale=2.0)
    onn.update()
    PyEval([ref], eqs, kernel, onn).compute(0.0, 1e-4)
    assert rel_err(pa.qtmp, ref.qtmp) == 0.0
    assert rel_err(pa.q, ref.q) < TOL


@pytest.mark.parametrize('kname', ['CubicSpline', 'WendlandQuintic'])
def test_generated_loop_all_vs_python(oracle, kname):
    """loop_all equations (the neighbour list NBRS / N_NBRS and the kernel
    object SPH_KERNEL handed to the equation, mako :62-80): a device CSR list per
    source, one thread per destination.  ShepardFilter's initialize() writes
    rhotmp, which its loop_all reads as a SOURCE property: the generated family
    runs initialize for all particles first (split launch), as the reference does.
    Checked against the same bodies executed as Python."""
    from oracle.py_eval import PyEval
    from custom_equations import GradientAllNbrs, ShepardFilter
    from pysph_amd import kernels as K
    from pysph_amd.equations import Group
    from pysph_amd.particle_array import get_particle_array_wcsph

    def build():
        rng = np.random.default_rng(21)
        out = []
        for name, n1, off in (('fluid', 11, 0.0), ('solid', 6, 0.02)):
            dx = 1.0 / n1
            g = (np.arange(n1) + 0.5) * dx
            x, y, z = [a.ravel() for a in np.meshgrid(g, g, g, indexing='ij')]
            n = x.size
            pa = get_particle_array_wcsph(
                name=name, x=x + off + 0.1 * dx * rng.uniform(-1, 1, n),
                y=y + 0.1 * dx * rng.uniform(-1, 1, n), z=z + 0.1 * dx * rng.uniform(-1, 1, n),
                h=1.2 / 11 * (1 + 0.1 * rng.uniform(-1, 1, n)), m=dx ** 3 * np.ones(n),
                rho=1 + 0.2 * rng.uniform(-1, 1, n))
            for extra in ('rhotmp', 'gx', 'gy', 'gz'):
                pa.add_property(extra)
            pa.rhotmp[:] = -5.0
            out.append(pa)
        return out
    eqs = [Group(equations=[ShepardFilter('fluid', ['fluid'])]),
           Group(equations=[GradientAllNbrs('fluid', ['fluid', 'solid'], scale=0.5),
                            GradientAllNbrs('solid', ['fluid'], scale=2.0)], real=False)]
    kernel = getattr(K, kname)(dim=3)
    arrays, ref = build(), build()
    a_eval, nnps, ctx = make_eval(arrays, eqs, kernel, 3)
    a_eval.compute(0.0, 1e-4)
    onn = oracle.OracleNNPS(3, ref, radius_scale=2.0)
    onn.update()
    PyEval(ref, eqs, kernel, onn).compute(0.0, 1e-4)
    for pa, pr in zip(arrays, ref):
        for prop in ('rho', 'rhotmp', 'gx', 'gy', 'gz'):
            e = rel_err(pa.properties[prop], pr.properties[prop])
            assert e < TOL, (pa.name, prop, e)
    assert abs(ref[0].rho - 1).max() < 0.5 and (ref[0].rhotmp != -5.0).all()


@pytest.mark.parametrize('case', ['sd_1d_line', 'wcsph_cube_varh',
                                  'wcsph_dam_dx0.1'])
def test_neighbour_sets_match_reference(case):
    g = load_golden(case + '.npz')
    arrays = arrays_from_golden(g, 'in')
    eqs, kernel, dim, outs = golden_case(case, g)
    a_eval, nnps, ctx = make_eval(arrays, eqs, kernel, dim)
    names = [pa.name for pa in arrays]
    pairs = set(k.split('/')[1] + '/' + k.split('/')[2]
                for k in g.files if k.startswith('nbrs/'))
    for pr in pairs:
        s, d = pr.split('/')
        start, idx = nnps.get_csr(names.index(s), names.index(d))
        gs, gi = g['nbrs/%s/start' % pr], g['nbrs/%s/idx' % pr]
        assert np.array_equal(start, gs)
        ref = np.concatenate([np.sort(gi[gs[i]:gs[i + 1]])
                              for i in range(len(gs) - 1)] or
                             [np.zeros(0, np.uint32)])
        assert np.array_equal(idx, ref)
    if case == 'sd_1d_line':  # test_acceleration_eval.py:341
        start, idx = nnps.get_csr(0, 0)
        assert list(np.diff(start.astype(int))) == [3, 4, 5, 5, 5, 5, 5, 5, 4, 3]


def _perturb(arrays, seed, c0, rho0, dx):
    rng = np.random.default_rng(seed)
    for pa in arrays:
        n = pa.get_number_of_particles()
        pa.rho[:] = rho0 * (1 + 0.02 * rng.uniform(-1, 1, n))
        if pa.name == 'fluid':
            for k in 'xyz':
                pa.properties[k] += 0.1 * dx * rng.uniform(-1, 1, n)
            for k in 'uvw':
                pa.properties[k][:] = 0.1 * c0 * rng.uniform(-1, 1, n)


def _copy_arrays(arrays):
    from pysph_amd.particle_array import ParticleArray
    out = []
    for pa in arrays:
        q = ParticleArray(name=pa.name, **{k: v.copy() for k, v in
                                           pa.properties.items()})
        q.set_num_real_particles(pa.get_number_of_particles(True))
        out.append(q)
    return out


@pytest.mark.parametrize('variant', [0, 6])
def test_dam_break_27k_vs_oracle(oracle, variant):
    """BASELINE config 1: dam_break_3d, dx=0.04 (9360+15152+160 particles)."""
    from pysph_amd.examples import dam_break_3d as db
    dx = 0.04
    arrays = db.create_particles(dx)
    _perturb(arrays, 42, db.c0, db.ro, dx)
    ref = _copy_arrays(arrays)
    eqs = db.create_scheme(dx).get_equations()
    kernel = db.create_kernel()
    a_eval, nnps, ctx = make_eval(arrays, eqs, kernel, 3, variant)
    a_eval.compute(0.0, 1e-5)
    onn = oracle.OracleNNPS(3, ref, radius_scale=kernel.radius_scale)
    onn.update()
    oev = oracle.OracleEval(ref, eqs, kernel, nthreads=8)
    oev.set_nnps(onn)
    oev.compute(0.0, 1e-5)
    for pa, pr in zip(arrays, ref):
        for prop in WC_OUT:
            e = rel_err(pa.properties[prop], pr.properties[prop])
            assert e < TOL, (pa.name, prop, e)
    # dt_cfl is a max over the neighbour set -> must be (near) identical
    assert rel_err(arrays[0].dt_cfl, ref[0].dt_cfl) < 1e-13


def make_cube(n1, seed=1234, hdx=1.3, varh=0.0, jitter=0.1):
    """S-cube of SURVEY 8(d): lattice mgrid[0:1:dx]^3, jitter U(+-0.1dx)."""
    from pysph_amd.particle_array import get_particle_array_wcsph
    from pysph_amd.examples import dam_break_3d as db
    rng = np.random.default_rng(seed)
    dx = 1.0 / n1
    g = np.arange(n1) * dx
    x, y, z = [a.ravel().copy() for a in np.meshgrid(g, g, g, indexing='ij')]
    n = x.size
    for a in (x, y, z):
        a += jitter * dx * rng.uniform(-1, 1, n)
    h = hdx * dx * (1 + varh * rng.uniform(-1, 1, n))
    pa = get_particle_array_wcsph(
        name='fluid', x=x, y=y, z=z, h=h, m=db.ro * dx ** 3 * np.ones(n),
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


@pytest.mark.parametrize('variant,varh', [(0, 0.0), (0, 0.2), (6, 0.0), (6, 0.2)])
def test_cube_100k_vs_oracle(oracle, variant, varh):
    from pysph_amd import kernels as K
    pa, dx = make_cube(46, varh=varh)
    ref = _copy_arrays([pa])
    eqs = cube_equations(dx)
    kernel = K.WendlandQuintic(dim=3)
    a_eval, nnps, ctx = make_eval([pa], eqs, kernel, 3, variant)
    a_eval.compute(0.0, 1e-5)
    onn = oracle.OracleNNPS(3, ref, radius_scale=2.0)
    onn.update()
    oev = oracle.OracleEval(ref, eqs, kernel, nthreads=8)
    oev.set_nnps(onn)
    oev.compute(0.0, 1e-5)
    for prop in WC_OUT:
        e = rel_err(pa.properties[prop], ref[0].properties[prop])
        assert e < TOL, (prop, e)
    # neighbour counts identical to the oracle's (integer work: exact)
    start, idx = nnps.get_csr(0, 0)
    ostart, oidx = onn.get_csr(0, 0, nthreads=8)
    assert np.array_equal(start, ostart)


def test_full_size_1m_properties():
    """At N=1M (oracle parity at the BASELINE sizes: tests/test_baseline_sizes.py)
    the two independent kernel schedules -- the plain per-lane 27-cell walk and
    the wavefront-tile kernel -- agree to rounding and outputs are finite."""
    from pysph_amd import kernels as K
    pa, dx = make_cube(100)
    eqs = cube_equations(dx)
    kernel = K.WendlandQuintic(dim=3)
    res = {}
    for variant in (0, 6):
        q = _copy_arrays([pa])
        a_eval, nnps, ctx = make_eval(q, eqs, kernel, 3, variant)
        a_eval.compute(0.0, 1e-5)
        res[variant] = q[0]
        ctx.close()
    for prop in WC_OUT:
        a, d = (res[v].properties[prop] for v in (0, 6))
        assert np.all(np.isfinite(d))
        assert rel_err(a, d) < 1e-12, prop


def test_group_semantics_real_start_stop(oracle):
    """Group(real=...), start_idx/stop_idx and ghost sources
    (equation.py:457-561; acceleration_eval_cython_helper.py:259-280)."""
    from pysph_amd import kernels as K
    from pysph_amd.equations import Group, ContinuityEquation, TaitEOS
    pa, dx = make_cube(16)
    n = pa.get_number_of_particles()
    pa.set_num_real_particles(n - 500)         # last 500 act as ghosts
    pa.arho[:] = 7.0
    pa.p[:] = -3.0
    ref = _copy_arrays([pa])
    eqs = [
        Group(equations=[TaitEOS(dest='fluid', sources=None, rho0=1000.,
                                 c0=10., gamma=7.0)], real=False),
        Group(equations=[ContinuityEquation(dest='fluid', sources=['fluid'])],
              real=True, start_idx=100, stop_idx=3000),
    ]
    kernel = K.CubicSpline(dim=3)
    a_eval, nnps, ctx = make_eval([pa], eqs, kernel, 3)
    a_eval.compute(0.0, 1e-5)
    onn = oracle.OracleNNPS(3, ref, radius_scale=2.0)
    onn.update()
    oev = oracle.OracleEval(ref, eqs, kernel)
    oev.set_nnps(onn)
    oev.compute(0.0, 1e-5)
    assert rel_err(pa.p, ref[0].p) < TOL             # all n incl. ghosts
    assert np.all(pa.arho[:100] == 7.0) and np.all(pa.arho[3000:] == 7.0)
    assert rel_err(pa.arho, ref[0].arho) < TOL


def test_group_control_flow_hooks():
    """pre/post/condition/iterate/update_nnps are host-side control
    (acceleration_eval_cython.mako:291-363; tests
    test_acceleration_eval.py:386-420 iterate)."""
    from pysph_amd import kernels as K
    from pysph_amd.equations import Group, SummationDensity
    pa, dx = make_cube(10)
    calls = []

    class CountingSD(SummationDensity):
        def __init__(self, dest, sources):
            SummationDensity.__init__(self, dest, sources)
            self.count = 0
            self.name = 'SummationDensity'

        def converged(self):   # SimpleEquation.converged, :148-154
            self.count += 1
            result = self.count - 1
            if result > 0:
                self.count = 0
            return result

    # resolve_equation matches on the class name
    CountingSD.__name__ = 'SummationDensity'
    eq = CountingSD(dest='fluid', sources=['fluid'])
    eqs = [
        Group(equations=[eq], iterate=True, max_iterations=5,
              pre=lambda: calls.append('pre'), post=lambda: calls.append('post'),
              update_nnps=True),
        Group(equations=[SummationDensity(dest='fluid', sources=['fluid'])],
              condition=lambda t, dt: t > 1.0),
    ]
    a_eval, nnps, ctx = make_eval([pa], eqs, K.CubicSpline(dim=3), 3)
    nupd = []
    orig = nnps.update
    nnps.update = lambda: (nupd.append(1), orig())[1]
    a_eval.compute(0.0, 0.1)
    assert calls == ['pre', 'post'] * 2       # converged on the 2nd iteration
    assert len(nupd) == 2
    rho1 = pa.rho.copy()
    pa.rho[:] = 0
    a_eval.compute(2.0, 0.1)                   # condition true now
    assert np.allclose(pa.rho, rho1, rtol=1e-13)


def test_edge_cases_empty_single_and_2d(oracle):
    from pysph_amd import kernels as K
    from pysph_amd.equations import Group, SummationDensity
    from pysph_amd.particle_array import get_particle_array_wcsph
    rng = np.random.default_rng(5)
    # empty source array + single destination particle
    a = get_particle_array_wcsph(name='a', x=[0.5], y=[0.5], z=[0.5], h=[0.1],
                                 m=[2.0])
    b = get_particle_array_wcsph(name='b')
    assert b.get_number_of_particles() == 0
    eqs = [Group(equations=[SummationDensity(dest='a', sources=['a', 'b'])])]
    kernel = K.CubicSpline(dim=3)
    a_eval, nnps, ctx = make_eval([a, b], eqs, kernel, 3)
    a_eval.compute(0.0, 0.1)
    assert abs(a.rho[0] - 2.0 / (np.pi * 0.1 ** 3)) < 1e-9 * a.rho[0]
    # 2-D problem in the z=0 plane, WendlandQuintic dim=2, ragged cells
    n = 3000
    x, y = rng.uniform(0, 1, n), rng.uniform(0, 1, n) ** 2
    pa = get_particle_array_wcsph(name='fluid', x=x, y=y, z=np.zeros(n),
                                  h=0.03 * (1 + 0.3 * rng.uniform(-1, 1, n)),
                                  m=np.ones(n))
    ref = _copy_arrays([pa])
    eqs = [Group(equations=[SummationDensity(dest='fluid', sources=['fluid'])])]
    kernel = K.WendlandQuintic(dim=2)
    for variant in (0, 6):
        q = _copy_arrays([pa])
        a_eval, nnps, ctx = make_eval(q, eqs, kernel, 2, variant)
        a_eval.compute(0.0, 0.1)
        onn = oracle.OracleNNPS(2, ref, radius_scale=2.0)
        onn.update()
        oev = oracle.OracleEval(ref, eqs, kernel)
        oev.set_nnps(onn)
        oev.compute(0.0, 0.1)
        assert rel_err(q[0].rho, ref[0].rho) < TOL
        start, idx = nnps.get_csr(0, 0)
        ostart, oidx = onn.get_csr(0, 0)
        assert np.array_equal(start, ostart)


@pytest.mark.parametrize('varh', [0.0, 0.1])
def test_dense_cells_coincident_particles(oracle, varh):
    """Stress the rarely taken paths of the aggregated kernel: > 96 candidates
    in a lane's 3-cell range (exact in-place tail), rows longer than one
    480-candidate tile (several tiles per row, mask slots flushed mid-source),
    and distinct particles at IDENTICAL positions (r = 0: the reference's
    r > 1e-12 gradient guard, linked-list neighbours include them)."""
    from pysph_amd import kernels as K
    from pysph_amd.particle_array import get_particle_array_wcsph
    rng = np.random.default_rng(17)
    n = 9000
    x, y, z = rng.uniform(0, 1, n), rng.uniform(0, 1, n), rng.uniform(0, 0.5, n)
    x[-60:], y[-60:], z[-60:] = x[:60], y[:60], z[:60]         # exact duplicates
    h0 = 0.11
    pa = get_particle_array_wcsph(
        name='fluid', x=x, y=y, z=z, h=h0 * (1 + varh * rng.uniform(-1, 1, n)),
        m=np.ones(n) / n, rho=1000.0 * (1 + 0.02 * rng.uniform(-1, 1, n)),
        u=rng.uniform(-1, 1, n), v=rng.uniform(-1, 1, n), w=rng.uniform(-1, 1, n))
    eqs = cube_equations(h0 / 1.3)
    kernel = K.WendlandQuintic(dim=3)
    ref = _copy_arrays([pa])
    onn = oracle.OracleNNPS(3, ref, 2.0)
    onn.update()
    oev = oracle.OracleEval(ref, eqs, kernel, nthreads=8)
    oev.set_nnps(onn)
    oev.compute(0.0, 1e-5)
    ostart, _ = onn.get_csr(0, 0)
    assert np.diff(ostart.astype(np.int64)).max() > 300       # really dense
    for variant in (0, 6):
        q = _copy_arrays([pa])
        a_eval, nnps, ctx = make_eval(q, eqs, kernel, 3, variant)
        a_eval.compute(0.0, 1e-5)
        start, idx = nnps.get_csr(0, 0)
        assert np.array_equal(start, ostart)
        for prop in WC_OUT:
            e = rel_err(q[0].properties[prop], ref[0].properties[prop])
            assert e < TOL, (variant, prop, e)


@pytest.mark.parametrize('seed', list(range(int(os.environ.get('SPH_FUZZ_SEEDS', '12')))))
def test_randomised_configurations_vs_oracle(oracle, seed):
    """Differential test over randomly drawn problem shapes: dimension,
    particle count, clustered vs uniform positions, constant or strongly
    varying h, one or two arrays (the second may be empty or tiny), kernel,
    Group(real, start_idx, stop_idx), pair-kernel variant.  Neighbour lists
    must be identical to the oracle's, fields within tolerance."""
    from pysph_amd import kernels as K
    from pysph_amd.equations import (ContinuityEquation, Group, MomentumEquation,
                                     SummationDensity, TaitEOS, XSPHCorrection)
    from pysph_amd.particle_array import get_particle_array_wcsph
    rng = np.random.default_rng(1000 + seed)
    dim = int(rng.choice([1, 2, 3], p=[0.15, 0.35, 0.5]))
    n = int(rng.choice([1, 7, 300, 2500, 9000]))
    if dim == 1:
        n = min(n, 2500)
    kname = str(rng.choice(['CubicSpline', 'WendlandQuintic', 'QuinticSpline', 'Gaussian']))
    if kname == 'WendlandQuintic' and dim == 1:
        kname = 'CubicSpline'
    kernel = getattr(K, kname)(dim=dim)
    varh = float(rng.choice([0.0, 0.0, 0.3]))
    clustered = bool(rng.integers(0, 2))
    spacing = (1.0 / max(n, 2)) ** (1.0 / dim)

    def coords(m):
        c = rng.uniform(0, 1, (m, 3))
        if clustered:
            c = c ** 2.5                      # dense corner, ragged cells
        c[:, dim:] = 0.0
        return c
    arrays = []
    sizes = [n, int(rng.choice([0, 1, max(n // 5, 1)]))]
    for name, m in zip(('fluid', 'solid'), sizes):
        c = coords(m)
        h = 1.3 * spacing * (1 + varh * rng.uniform(-1, 1, m))
        pa = get_particle_array_wcsph(
            name=name, x=c[:, 0], y=c[:, 1], z=c[:, 2], h=h, m=spacing ** dim * np.ones(m),
            rho=1000.0 * (1 + 0.02 * rng.uniform(-1, 1, m)), u=rng.uniform(-1, 1, m),
            v=rng.uniform(-1, 1, m), w=rng.uniform(-1, 1, m))
        arrays.append(pa)
    real = bool(rng.integers(0, 2))
    start = int(rng.integers(0, max(n // 3, 1)))
    stop = None if rng.integers(0, 2) else int(rng.integers(start, n + 1))
    c0 = 10.0
    eqs = [
        Group(real=False, equations=[TaitEOS(dest=a.name, sources=None, rho0=1000.0, c0=c0, gamma=7.0)
                                     for a in arrays]),
        Group(real=real, start_idx=start, stop_idx=stop, equations=[
            ContinuityEquation(dest='fluid', sources=['fluid', 'solid']),
            MomentumEquation(dest='fluid', sources=['fluid', 'solid'], c0=c0, alpha=0.3, beta=0.1,
                             gy=-1.0, tensile_correction=bool(rng.integers(0, 2))),
            XSPHCorrection(dest='fluid', sources=['fluid'], eps=0.4)]),
        # start/stop of a group apply to every destination in it: the solid
        # destination gets its own groups
        Group(equations=[ContinuityEquation(dest='solid', sources=['fluid'])]),
        Group(equations=[SummationDensity(dest='solid', sources=['fluid', 'solid'])]),
    ]
    # fields the groups do not touch keep their input values: give them some
    for a in arrays:
        for k in ('arho', 'au', 'av', 'aw', 'ax', 'ay', 'az'):
            a.properties[k][:] = rng.uniform(-1, 1, a.get_number_of_particles())
    ref = _copy_arrays(arrays)
    onn = oracle.OracleNNPS(dim, ref, radius_scale=kernel.radius_scale)
    onn.update()
    oev = oracle.OracleEval(ref, eqs, kernel, nthreads=4)
    oev.set_nnps(onn)
    oev.compute(0.1, 1e-4)
    variant = int(rng.choice([0, 6, 6]))
    q = _copy_arrays(arrays)
    a_eval, nnps, ctx = make_eval(q, eqs, kernel, dim, variant)
    a_eval.compute(0.1, 1e-4)
    for si in range(2):
        for di in range(2):
            if q[di].get_number_of_particles() == 0:
                continue
            s1, i1 = nnps.get_csr(si, di)
            s2, i2 = onn.get_csr(si, di)
            assert np.array_equal(s1, s2), (seed, si, di)
            assert np.array_equal(i1, np.concatenate(
                [np.sort(i2[s2[k]:s2[k + 1]]) for k in range(len(s2) - 1)] or [i2[:0]]))
    for pa, pr in zip(q, ref):
        for prop in WC_OUT:
            e = rel_err(pa.properties[prop], pr.properties[prop])
            assert e < TOL, (seed, dim, n, kname, varh, clustered, variant, pa.name, prop, e)


@pytest.mark.parametrize('resident', [False, True], ids=['host-owned', 'device-resident-2nd-eval'])
@pytest.mark.parametrize('seed', list(range(int(os.environ.get('SPH_FUZZ_SEEDS', '8')))))
def test_randomised_tvf_and_elastic_vs_oracle(oracle, seed, resident):
    """The same differential idea for the other two hand-written equation sets:
    TVF (QuinticSpline / Gaussian, optional artificial viscosity) and the
    elastic-solid set (CubicSpline / WendlandQuintic, 2-D or 3-D), random
    sizes, jitter, constant or varying h, random input fields."""
    from pysph_amd import kernels as K
    from pysph_amd.particle_array import get_particle_array_tvf_fluid
    from pysph_amd.scheme import TVFScheme
    from pysph_amd.solid_mech import ElasticSolidsScheme, get_particle_array_elastic_dynamics
    from helpers import TVF_OUT, EL_OUT
    rng = np.random.default_rng(5000 + seed)
    which = 'tvf' if seed % 2 == 0 else 'elastic'
    dim = int(rng.choice([2, 3]))
    n1 = int(rng.choice([5, 9, 14])) if dim == 3 else int(rng.choice([8, 30, 60]))
    dx = 1.0 / n1
    g = (np.arange(n1) + 0.5) * dx
    if dim == 3:
        x, y, z = [a.ravel() for a in np.meshgrid(g, g, g, indexing='ij')]
    else:
        x, y = [a.ravel() for a in np.meshgrid(g, g, indexing='ij')]
        z = np.zeros_like(x)
    n = x.size
    jit = float(rng.choice([0.0, 0.15]))
    x = x + jit * dx * rng.uniform(-1, 1, n)
    y = y + jit * dx * rng.uniform(-1, 1, n)
    if dim == 3:
        z = z + jit * dx * rng.uniform(-1, 1, n)
    varh = float(rng.choice([0.0, 0.2]))
    if which == 'tvf':
        kernel = getattr(K, str(rng.choice(['QuinticSpline', 'Gaussian'])))(dim=dim)
        pa = get_particle_array_tvf_fluid(
            name='fluid', x=x, y=y, z=z, h=dx * (1 + varh * rng.uniform(-1, 1, n)),
            m=dx ** dim * np.ones(n), rho=1 + 0.05 * rng.uniform(-1, 1, n),
            u=rng.uniform(-1, 1, n), v=rng.uniform(-1, 1, n), w=rng.uniform(-1, 1, n),
            uhat=rng.uniform(-1, 1, n), vhat=rng.uniform(-1, 1, n), what=rng.uniform(-1, 1, n))
        eqs = TVFScheme(['fluid'], [], dim=dim, rho0=1.0, c0=10.0, nu=float(rng.choice([0.0, 0.02])),
                        p0=100.0, pb=100.0, h0=dx, gx=0.3, alpha=float(rng.choice([0.0, 0.2]))
                        ).get_equations()
        outs = TVF_OUT
    else:
        kernel = getattr(K, str(rng.choice(['CubicSpline', 'WendlandQuintic'])))(dim=dim)
        h0 = 1.3 * dx
        pa = get_particle_array_elastic_dynamics(
            name='solid', x=x, y=y, z=z, h=h0 * (1 + varh * rng.uniform(-1, 1, n)),
            m=1.2 * dx ** dim * np.ones(n), rho=1.2 * (1 + 0.02 * rng.uniform(-1, 1, n)),
            u=0.1 * rng.uniform(-1, 1, n), v=0.1 * rng.uniform(-1, 1, n),
            w=0.1 * rng.uniform(-1, 1, n) * (dim == 3),
            constants=dict(E=1e7, nu=0.3975, rho_ref=1.2, n=4,
                           wdeltap=float(kernel.kernel(rij=dx, h=h0))))
        for k in ('s00', 's01', 's02', 's11', 's12', 's22'):
            pa.properties[k][:] = 1e3 * rng.uniform(-1, 1, n)
        eqs = ElasticSolidsScheme(['solid'], [], dim=dim).get_equations()
        outs = EL_OUT
    ref = _copy_arrays([pa])
    ref[0].constants = dict((k, v.copy()) for k, v in pa.constants.items())
    onn = oracle.OracleNNPS(dim, ref, radius_scale=kernel.radius_scale)
    onn.update()
    oev = oracle.OracleEval(ref, eqs, kernel, nthreads=4)
    oev.set_nnps(onn)
    oev.compute(0.0, 1e-5)
    variant = int(rng.choice([0, 6, 6]))
    if resident:
        # round 4: the SECOND evaluation on device-resident state -- TVF force records without p and V, elastic rate
        # records without h and m, the tension word -- (both equation sets are idempotent on their inputs)
        variant = 6
        a_eval, nnps, ctx = make_eval([pa], eqs, kernel, dim, variant, sync='manual')
        pa.gpu.push()
        nnps.sync = False
        nnps.update()
        a_eval.compute(0.0, 1e-5)
        nnps.update()
        a_eval.compute(0.0, 1e-5)
        if varh == 0.0:
            assert ctx.timer_get('n_mass_fused')[1] > 0, 'the one-mass-per-array records did not engage'
        a_eval.c_acceleration_eval.pull_outputs()
    else:
        a_eval, nnps, ctx = make_eval([pa], eqs, kernel, dim, variant)
        a_eval.compute(0.0, 1e-5)
    for prop in outs:
        if prop in pa.properties:
            e = rel_err(pa.properties[prop], ref[0].properties[prop])
            assert e < TOL, (seed, which, dim, n1, type(kernel).__name__, varh, variant, resident, prop, e)


@pytest.mark.parametrize('case', ['wcsph_cube_varh', 'tvf_cube', 'wcsph_dam_dx0.1', 'elastic_3d',
                                  'elastic_2d', 'tvf_wall'])
def test_record_f32_mode_vs_golden(case):
    """Option record_f32: the packed records (positions relative to the grid
    origin, h and every gathered property) are stored as floats -- about half the
    gather pieces per pair -- while the pair arithmetic and the accumulation stay
    fp64.  Results carry fp32 input precision: compared with the fp64 golden
    vectors at an fp32 tolerance (fp32 parity is against the
    fp64 oracle), for every hand-written family and a generated one."""
    g = load_golden(case + '.npz')
    arrays = arrays_from_golden(g, 'in')
    eqs, kernel, dim, outs = golden_case(case, g)
    a_eval, nnps, ctx = make_eval(arrays, eqs, kernel, dim, 6)
    ctx.set_option('record_f32', 1)
    a_eval.compute(float(g['t']), float(g['dt']))
    worst = 0.0
    for pa in arrays:
        for prop in outs:
            key = 'out/%s/%s' % (pa.name, prop)
            if key in g.files and prop in pa.properties:
                e = rel_err(pa.properties[prop], g[key])
                worst = max(worst, e)
                assert e < 2e-5, (case, pa.name, prop, e)
    assert worst > 1e-12          # really a different precision, not the fp64 path
    print('record_f32 %s: max rel err %.3e' % (case, worst))


@pytest.mark.parametrize('case', ['wcsph_cube_varh', 'tvf_cube', 'wcsph_dam_dx0.1', 'elastic_3d',
                                  'elastic_2d'])
def test_fp32_arithmetic_vs_golden(case):
    """Option arith_f32 (BASELINE config 5, SURVEY.md 7 "fp64 + fp32
    instantiations"): the hand-written families are instantiated with
    Real = float -- fp32 records, fp32 pair arithmetic (v_rcp_f32 / v_rsq_f32,
    no refinement), fp32 accumulators, one store per output -- as the
    reference's OpenCL / CUDA backends do unless --use-double
    (acceleration_eval_gpu_helper.py:281-283,437-441).  Tolerance: the fp64
    golden vectors (= the reference's Cython arithmetic) at 5e-5 of the field
    maximum -- a sum of ~100 fp32 terms with cancellation between neighbours
    (fp32 eps 6e-8 x sqrt(100) x a cancellation factor of a few tens)."""
    g = load_golden(case + '.npz')
    arrays = arrays_from_golden(g, 'in')
    eqs, kernel, dim, outs = golden_case(case, g)
    a_eval, nnps, ctx = make_eval(arrays, eqs, kernel, dim, 6)
    ctx.set_option('arith_f32', 1)
    a_eval.compute(float(g['t']), float(g['dt']))
    worst = 0.0
    for pa in arrays:
        for prop in outs:
            key = 'out/%s/%s' % (pa.name, prop)
            if key in g.files and prop in pa.properties:
                e = rel_err(pa.properties[prop], g[key])
                worst = max(worst, e)
                assert e < 5e-5, (case, pa.name, prop, e)
    assert worst > 1e-9           # really fp32 arithmetic, not the fp64 path
    print('arith_f32 %s: max rel err %.3e' % (case, worst))


def test_error_behaviour():
    """Same failures as the reference: RuntimeError for missing properties
    (acceleration_eval.py:32-73) and for >2^28 cells
    (linked_list_nnps.pyx:335-343); unknown equations fail loudly."""
    from pysph_amd import kernels as K
    from pysph_amd.acceleration_eval import AccelerationEval
    from pysph_amd.equations import Equation, SummationDensity
    from pysph_amd.particle_array import get_particle_array
    from pysph_amd.nnps import HipNNPS
    from pysph_amd import device as dev
    f = get_particle_array(name='f', x=[0.0, 1.0])
    with pytest.raises(RuntimeError):
        AccelerationEval([f], [SummationDensity(dest='fluid', sources=['f'])],
                         K.CubicSpline(dim=1))

    class Unknown(Equation):
        pass
    with pytest.raises(NotImplementedError):
        AccelerationEval([f], [Unknown(dest='f', sources=['f'])],
                         K.CubicSpline(dim=1))
    # test_nnps.py:1019: too many cells -> RuntimeError
    big = get_particle_array(name='big', x=[0.0, 1e6], y=[0.0, 1e6],
                             z=[0.0, 1e6], h=[1e-3, 1e-3])
    with pytest.raises(RuntimeError):
        HipNNPS(3, [big], radius_scale=2.0, ctx=dev.HipContext(0))


def test_dt_reductions_on_device():
    """max(dt_cfl), max(dt_force) as Integrator.compute_time_step reads them
    (integrator.py:161-200)."""
    from pysph_amd import kernels as K
    pa, dx = make_cube(20)
    eqs = cube_equations(dx)
    a_eval, nnps, ctx = make_eval([pa], eqs, K.WendlandQuintic(dim=3), 3)
    a_eval.compute(0.0, 1e-5)
    assert pa.gpu.max('dt_cfl') == pa.dt_cfl.max()
    assert pa.gpu.max('dt_force') == pa.dt_force.max()


def test_halo_device_ops_two_slabs_one_gpu(oracle):
    """Device side of the multi-GPU path on ONE GPU: two
    contexts play two slab ranks; ghosts are selected/packed/appended by the
    HIP halo kernels (sph_halo_select/pack/append) and handed over as torch
    buffers (what RCCL send/recv would move).  Real-particle results of both
    slabs must equal the single-domain oracle, matched by global index."""
    import torch
    from pysph_amd import device as dev
    from pysph_amd import kernels as K
    from pysph_amd.acceleration_eval import AccelerationEval, SPHCompiler
    from pysph_amd.nnps import HipNNPS
    from pysph_amd.parallel import DeviceHaloOps, WCSPH_HALO_PROPS
    from pysph_amd.particle_array import ParticleArray
    full, dx = make_cube(24)
    kernel = K.WendlandQuintic(dim=3)
    eqs = cube_equations(dx)
    width = kernel.radius_scale * 1.3 * dx
    x = full.x
    parts = [np.nonzero(x < 0.5)[0], np.nonzero(x >= 0.5)[0]]
    ranks = []
    for r, own in enumerate(parts):
        pa = ParticleArray(name='fluid', **{k: v[own].copy() for k, v in
                                            full.properties.items()})
        ctx = dev.HipContext(0)
        dev.attach(pa, ctx).push()
        ops = DeviceHaloOps(pa, ctx, WCSPH_HALO_PROPS, 0)
        lo, hi = (-1e30, 0.5) if r == 0 else (0.5, 1e30)
        ops.drop_ghosts()
        n_lo, n_hi = ops.select(lo + width, hi - width)
        ranks.append(dict(pa=pa, ctx=ctx, ops=ops, own=own, n=(n_lo, n_hi)))
    # rank 0 sends its hi list to rank 1, rank 1 its lo list to rank 0
    assert ranks[0]['n'][0] == 0 and ranks[1]['n'][1] == 0
    b01 = ranks[0]['ops'].pack(1, ranks[0]['n'][1], 0.0)
    b10 = ranks[1]['ops'].pack(0, ranks[1]['n'][0], 0.0)
    torch.cuda.synchronize()
    ranks[1]['ops'].append(b01, ranks[0]['n'][1])
    ranks[0]['ops'].append(b10, ranks[1]['n'][0])
    assert ranks[0]['n'][1] > 0 and ranks[1]['n'][0] > 0
    onn = oracle.OracleNNPS(3, [full], 2.0)
    onn.update()
    oev = oracle.OracleEval([full], eqs, kernel, nthreads=4)
    oev.set_nnps(onn)
    oev.compute(0.0, 1e-5)
    for rk in ranks:
        pa, ctx = rk['pa'], rk['ctx']
        a_eval = AccelerationEval([pa], eqs, kernel)
        SPHCompiler(a_eval, ctx=ctx, sync='manual').compile()
        nnps = HipNNPS(3, [pa], radius_scale=2.0, ctx=ctx, sync=False)
        a_eval.set_nnps(nnps)
        a_eval.compute(0.0, 1e-5)
        nreal = pa.get_number_of_particles()
        assert pa.gpu.get_number_of_particles() > nreal       # ghosts present
        assert pa.gpu.get_number_of_particles(True) == nreal
        pa.gpu.pull(*WC_OUT)
        for prop in WC_OUT:
            e = rel_err(pa.properties[prop], full.properties[prop][rk['own']])
            assert e < TOL, (prop, e)


def test_slab_decomposition_migrate_rebalance_one_gpu(oracle):
    """SlabDecomposition end to end with the DEVICE primitives
    (sph_halo_select/pack/remove_selected/append) on one GPU: two threads play
    two slab ranks (tests/helpers.ThreadDist stands in for torch.distributed).
    Ownership starts wrong and lopsided; after update() every particle is in
    its slab, after rebalance() the counts are even, and the evaluation of each
    slab's real particles equals the single-domain oracle matched by global id
    (carried in the spare fp64 property e0).  Mirrors
    pysph/parallel/tests/example_test_case.py:143-166."""
    import threading
    import torch
    from helpers import ThreadDist
    from pysph_amd import device as dev
    from pysph_amd import kernels as K
    from pysph_amd.acceleration_eval import AccelerationEval, SPHCompiler
    from pysph_amd.nnps import HipNNPS
    from pysph_amd.parallel import SlabDecomposition
    from pysph_amd.particle_array import ParticleArray
    full, dx = make_cube(24)
    n = full.get_number_of_particles()
    kernel = K.WendlandQuintic(dim=3)
    eqs = cube_equations(dx)
    width = kernel.radius_scale * 1.3 * dx
    hub = ThreadDist(2)
    results, errors = {}, []

    def rank_main(rank):
        try:
            ts = torch.cuda.Stream()
            with torch.cuda.stream(ts):
                x = full.x
                own = np.nonzero(x < 0.3)[0] if rank == 0 else np.nonzero(x >= 0.3)[0]
                props = {k: v[own].copy() for k, v in full.properties.items()}
                props['e0'] = own.astype(np.float64)
                pa = ParticleArray(name='fluid', **props)
                ctx = dev.HipContext(0, ts.cuda_stream)
                dev.attach(pa, ctx).push()
                lo, hi = (0.0, 0.5) if rank == 0 else (0.5, 1.0)
                dec = SlabDecomposition([pa], ctx, rank, 2, axis=0, width=width,
                                        lo=lo, hi=hi, dist=hub.view(rank))
                dec.update()
                gpu = pa.gpu
                xs = np.empty(gpu.get_number_of_particles(True))
                gpu.pull_into('x', xs)
                assert (xs < 0.5).all() if rank == 0 else (xs >= 0.5).all()
                assert gpu.get_number_of_particles() > xs.size        # ghosts
                faces, rounds = dec.rebalance(nbins=1024)
                dec.exchange()
                nreal = gpu.get_number_of_particles(True)
                assert abs(nreal - n // 2) <= 0.02 * n, nreal
                a_eval = AccelerationEval([pa], eqs, kernel)
                SPHCompiler(a_eval, ctx=ctx, sync='manual').compile()
                nnps = HipNNPS(3, [pa], radius_scale=2.0, ctx=ctx, sync=False)
                a_eval.set_nnps(nnps)
                a_eval.compute(0.0, 1e-5)
                gpu.sync_host()
                results[rank] = {k: pa.properties[k].copy() for k in WC_OUT + ['e0']}
        except Exception as e:      # surface in the main thread
            import traceback
            errors.append(traceback.format_exc())
            try:
                hub.barrier.abort()
            except Exception:
                pass

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    assert not errors, errors[0]
    onn = oracle.OracleNNPS(3, [full], 2.0)
    onn.update()
    oev = oracle.OracleEval([full], eqs, kernel, nthreads=4)
    oev.set_nnps(onn)
    oev.compute(0.0, 1e-5)
    gids = []
    for r in range(2):
        gid = results[r]['e0'].astype(np.int64)
        gids.append(gid)
        for prop in WC_OUT:
            e = rel_err(results[r][prop], full.properties[prop][gid])
            assert e < TOL, (r, prop, e)
    assert (np.sort(np.concatenate(gids)) == np.arange(n)).all()


def test_dam_break_two_slabs_three_arrays_one_gpu(oracle):
    """BASELINE config 4 in miniature: the dam-break tank (fluid + boundary +
    obstacle) cut into two x-slabs at the particle-count median, one thread per
    slab rank on one GPU.  Every array gets its own ghosts
    (SlabDecomposition); the slab that has no obstacle particles holds an EMPTY
    obstacle array.  real=False EOS groups recompute p, cs on the ghosts;
    results of all real particles equal the single-domain oracle by global id."""
    import threading
    import torch
    from helpers import ThreadDist
    from pysph_amd import device as dev
    from pysph_amd.acceleration_eval import AccelerationEval, SPHCompiler
    from pysph_amd.examples import dam_break_3d as db
    from pysph_amd.nnps import HipNNPS
    from pysph_amd.parallel import SlabDecomposition, slab_bounds
    dx = 0.06
    full = db.create_particles(dx)
    _perturb(full, 7, db.c0, db.ro, dx)
    for a in full:
        a.add_property('e0', data=np.arange(a.get_number_of_particles(), dtype=np.float64))
    ref = _copy_arrays(full)
    eqs = db.create_scheme(dx).get_equations()
    kernel = db.create_kernel()
    cuts = slab_bounds(np.concatenate([a.x for a in full]), 2)
    width = kernel.radius_scale * db.hdx * dx
    hub = ThreadDist(2)
    results, errors = {}, []

    def rank_main(rank):
        try:
            ts = torch.cuda.Stream()
            with torch.cuda.stream(ts):
                lo = -1e30 if rank == 0 else float(cuts[1])
                hi = float(cuts[1]) if rank == 0 else 1e30
                arrays = [a.extract_particles(np.nonzero((a.x >= lo) & (a.x < hi))[0], name=a.name)
                          for a in full]
                ctx = dev.HipContext(0, ts.cuda_stream)
                for a in arrays:
                    dev.attach(a, ctx).push()
                dec = SlabDecomposition(arrays, ctx, rank, 2, axis=0, width=width, lo=lo, hi=hi,
                                        dist=hub.view(rank))
                dec.update()
                a_eval = AccelerationEval(arrays, eqs, kernel)
                SPHCompiler(a_eval, ctx=ctx, sync='manual').compile()
                nnps = HipNNPS(3, arrays, radius_scale=kernel.radius_scale, ctx=ctx, sync=False)
                a_eval.set_nnps(nnps)
                a_eval.compute(0.0, 1e-5)
                out = {}
                for a in arrays:
                    a.gpu.sync_host()
                    out[a.name] = {k: a.properties[k].copy() for k in WC_OUT + ['e0']}
                results[rank] = out
        except Exception:
            import traceback
            errors.append(traceback.format_exc())
            try:
                hub.barrier.abort()
            except Exception:
                pass

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    assert not errors, errors[0]
    onn = oracle.OracleNNPS(3, ref, radius_scale=kernel.radius_scale)
    onn.update()
    oev = oracle.OracleEval(ref, eqs, kernel, nthreads=8)
    oev.set_nnps(onn)
    oev.compute(0.0, 1e-5)
    sizes = [len(results[r]['obstacle']['e0']) for r in range(2)]
    assert min(sizes) == 0 and max(sizes) > 0          # one slab has no obstacle
    for pr in ref:
        seen = 0
        for r in range(2):
            d = results[r][pr.name]
            gid = d['e0'].astype(np.int64)
            seen += gid.size
            if gid.size == 0:
                continue
            for prop in WC_OUT:
                e = rel_err(d[prop], pr.properties[prop][gid],
                            scale=max(np.abs(pr.properties[prop]).max(), 1e-300))
                assert e < TOL, (r, pr.name, prop, e)
        assert seen == pr.get_number_of_particles()


def test_rccl_transport_self_periodic_slab(oracle):
    """The REAL transport on one GPU: torch.distributed with the nccl backend
    (= RCCL) and world_size 1.  A slab that is periodic along its own axis is
    its own neighbour on both faces, so SlabHalo.exchange() runs its whole
    RCCL path -- all_gather_into_tensor for the counts, batch_isend_irecv for
    both payloads (two messages to the SAME peer: the posting-order case) --
    on the context's stream; all_reduce for the scalars.  Reference: the same
    periodic images from the host DomainManager, evaluated by the oracle."""
    import socket
    import torch
    import torch.distributed as dist
    from pysph_amd import device as dev
    from pysph_amd import kernels as K
    from pysph_amd.acceleration_eval import AccelerationEval, SPHCompiler
    from pysph_amd.domain import DomainManager
    from pysph_amd.nnps import HipNNPS
    from pysph_amd.parallel import SlabHalo, allreduce_scalars
    pa, dx = make_cube(20)
    nreal = pa.get_number_of_particles()
    kernel = K.WendlandQuintic(dim=3)
    eqs = cube_equations(dx)
    ref = _copy_arrays([pa])
    dom = DomainManager(xmin=0.0, xmax=1.0, periodic_in_x=True, n_layers=1.0)
    dom.set_particles(ref, kernel.radius_scale)
    dom.update()
    onn = oracle.OracleNNPS(3, ref, 2.0)
    onn.update()
    oev = oracle.OracleEval(ref, eqs, kernel, nthreads=4)
    oev.set_nnps(onn)
    oev.compute(0.0, 1e-5)
    sk = socket.socket()
    sk.bind(('127.0.0.1', 0))
    port = sk.getsockname()[1]
    sk.close()
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('nccl', rank=0, world_size=1,
                            device_id=torch.device('cuda', 0))
    try:
        ts = torch.cuda.Stream()
        with torch.cuda.stream(ts):
            ctx = dev.HipContext(0, ts.cuda_stream)
            dev.attach(pa, ctx).push()
            halo = SlabHalo(pa, ctx, 0, 1, axis=0, width=kernel.radius_scale * 1.3 * dx,
                            lo=0.0, hi=1.0, periodic=True, period=1.0, dist=dist)
            halo.exchange()
            halo.exchange()                      # ghosts are dropped and rebuilt
            sent_lo, sent_hi, got_lo, got_hi = halo.last_counts
            assert sent_lo > 0 and sent_hi > 0 and (got_lo, got_hi) == (sent_hi, sent_lo)
            assert pa.gpu.get_number_of_particles() == ref[0].get_number_of_particles()
            a_eval = AccelerationEval([pa], eqs, kernel)
            SPHCompiler(a_eval, ctx=ctx, sync='manual').compile()
            nnps = HipNNPS(3, [pa], radius_scale=2.0, ctx=ctx, sync=False)
            a_eval.set_nnps(nnps)
            a_eval.compute(0.0, 1e-5)
            mx = allreduce_scalars([pa.gpu.max('dt_cfl')], 'max', dist=dist,
                                   device=torch.device('cuda', 0))[0]
            pa.gpu.pull(*WC_OUT)
    finally:
        dist.destroy_process_group()
    assert mx == ref[0].dt_cfl[:nreal].max()
    for prop in WC_OUT:
        e = rel_err(pa.properties[prop][:nreal], ref[0].properties[prop][:nreal])
        assert e < TOL, (prop, e)


@pytest.mark.parametrize('seed', list(range(int(os.environ.get('SPH_FUZZ_SEEDS', '6')))))
def test_randomised_slab_decomposition_vs_single_domain(oracle, seed):
    """Random multi-rank layouts on one GPU (threads + ThreadDist): 2-4 slab
    ranks, uneven random cuts (a slab may start empty), optionally periodic
    along the slab axis, ownership scrambled so that migration has to move
    particles (possibly over several slabs), then ghosts; every rank's real
    particles must reproduce the single-domain oracle by global id."""
    import threading
    import torch
    from helpers import ThreadDist
    from pysph_amd import device as dev
    from pysph_amd import kernels as K
    from pysph_amd.acceleration_eval import AccelerationEval, SPHCompiler
    from pysph_amd.domain import DomainManager
    from pysph_amd.nnps import HipNNPS
    from pysph_amd.parallel import SlabDecomposition
    from pysph_amd.particle_array import ParticleArray
    rng = np.random.default_rng(9000 + seed)
    world = int(rng.choice([2, 3, 4]))
    periodic = bool(rng.integers(0, 2))
    n1 = int(rng.choice([18, 22]))
    full, dx = make_cube(n1, seed=int(rng.integers(1, 1000)))
    n = full.get_number_of_particles()
    full.add_property('e0', data=np.arange(n, dtype=np.float64))
    kernel = K.WendlandQuintic(dim=3)
    eqs = cube_equations(dx)
    width = kernel.radius_scale * 1.3 * dx
    # random faces in (0,1), at least one ghost width apart when periodic
    while True:
        cuts = np.sort(rng.uniform(0.05, 0.95, world - 1))
        faces = np.concatenate([[0.0], cuts, [1.0]])
        if np.diff(faces).min() > 1.1 * width:      # feasible: world * 1.1 * width < 1
            break
    # scrambled initial ownership: each particle starts on a random rank
    start_rank = rng.integers(0, world, n)
    ref = _copy_arrays([full])
    if periodic:
        dom = DomainManager(xmin=0.0, xmax=1.0, periodic_in_x=True, n_layers=1.0)
        dom.set_particles(ref, kernel.radius_scale)
        dom.update()
    onn = oracle.OracleNNPS(3, ref, 2.0)
    onn.update()
    oev = oracle.OracleEval(ref, eqs, kernel, nthreads=4)
    oev.set_nnps(onn)
    oev.compute(0.0, 1e-5)
    hub = ThreadDist(world)
    results, errors = {}, []

    def rank_main(rank):
        try:
            ts = torch.cuda.Stream()
            with torch.cuda.stream(ts):
                own = np.nonzero(start_rank == rank)[0]
                pa = ParticleArray(name='fluid', **{k: v[own].copy() for k, v in full.properties.items()})
                ctx = dev.HipContext(0, ts.cuda_stream)
                dev.attach(pa, ctx).push()
                dec = SlabDecomposition([pa], ctx, rank, world, axis=0, width=width,
                                        lo=float(faces[rank]), hi=float(faces[rank + 1]),
                                        periodic=periodic, period=1.0, dist=hub.view(rank))
                for _ in range(world):          # a particle moves one slab per call
                    dec.migrate()
                dec.exchange()
                a_eval = AccelerationEval([pa], eqs, kernel)
                SPHCompiler(a_eval, ctx=ctx, sync='manual').compile()
                nnps = HipNNPS(3, [pa], radius_scale=2.0, ctx=ctx, sync=False)
                a_eval.set_nnps(nnps)
                a_eval.compute(0.0, 1e-5)
                pa.gpu.sync_host()
                results[rank] = dict((k, pa.properties[k].copy()) for k in WC_OUT + ['e0', 'x'])
        except Exception:
            import traceback
            errors.append(traceback.format_exc())
            try:
                hub.barrier.abort()
            except Exception:
                pass

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(600)
    assert not errors, errors[0]
    gids = []
    for r in range(world):
        d = results[r]
        gid = d['e0'].astype(np.int64)
        gids.append(gid)
        if gid.size == 0:
            continue
        assert (d['x'] >= faces[r] - (1e30 if r == 0 and not periodic else 0)).all()
        assert (d['x'] < faces[r + 1] + (1e30 if r == world - 1 and not periodic else 0)).all()
        for prop in WC_OUT:
            e = rel_err(d[prop], ref[0].properties[prop][gid],
                        scale=max(np.abs(ref[0].properties[prop][:n]).max(), 1e-300))
            assert e < TOL, (seed, world, periodic, r, prop, e)
    assert (np.sort(np.concatenate(gids)) == np.arange(n)).all()


def test_device_reorder_keeps_results(oracle):
    """NNPS.spatially_order_particles on device-resident state
    (nnps_base.pyx:1615-1629; solver.py:296-302): properties are permuted by
    the cell order, results are unchanged when matched through that order."""
    from pysph_amd import device as dev
    from pysph_amd import kernels as K
    from pysph_amd.acceleration_eval import AccelerationEval, SPHCompiler
    from pysph_amd.nnps import HipNNPS
    pa, dx = make_cube(20)
    ref = _copy_arrays([pa])
    eqs = cube_equations(dx)
    kernel = K.WendlandQuintic(dim=3)
    ctx = dev.HipContext(0)
    dev.attach(pa, ctx).push()
    a_eval = AccelerationEval([pa], eqs, kernel)
    SPHCompiler(a_eval, ctx=ctx, sync='manual').compile()
    nnps = HipNNPS(3, [pa], radius_scale=2.0, ctx=ctx, sync=False)
    a_eval.set_nnps(nnps)
    order = nnps.get_spatially_ordered_indices(0).astype(np.int64)
    nnps.spatially_order_particles(0)
    nnps.update()
    order2 = nnps.get_spatially_ordered_indices(0)
    assert np.array_equal(order2, np.arange(order.size))   # already sorted now
    a_eval.compute(0.0, 1e-5)
    pa.gpu.pull()
    onn = oracle.OracleNNPS(3, ref, 2.0)
    onn.update()
    oev = oracle.OracleEval(ref, eqs, kernel, nthreads=4)
    oev.set_nnps(onn)
    oev.compute(0.0, 1e-5)
    assert np.array_equal(pa.x, ref[0].x[order])
    for prop in WC_OUT:
        assert rel_err(pa.properties[prop], ref[0].properties[prop][order]) < TOL, prop


def test_sph_evaluator_known_answers():
    """The reference's own known answers for SPHEvaluator
    (pysph/tools/tests/test_sph_evaluator.py:21-73): summation density of ten
    unit masses on [0, 1] seen from x = 0.5 is 9.0 (2 places), from x = 0 with a
    periodic domain 9.0, from x = 0 without it 7.0 (1 place); default Gaussian
    kernel, host arrays authoritative."""
    from pysph_amd.domain import DomainManager
    from pysph_amd.equations import SummationDensity
    from pysph_amd.particle_array import get_particle_array
    from pysph_amd.tools import SPHEvaluator
    x = np.linspace(0, 1, 10)
    dx = x[1] - x[0]
    src = get_particle_array(name='src', x=x, m=np.ones_like(x), h=np.ones_like(x) * dx)
    eqs = [SummationDensity(dest='dest', sources=['src'])]
    dest = get_particle_array(name='dest', x=[0.5], h=src.h[:1])
    ev = SPHEvaluator(arrays=[dest, src], equations=eqs, dim=1)
    ev.evaluate()
    assert abs(dest.rho[0] - 9.0) < 0.5e-2
    # with a periodic domain
    dest2 = get_particle_array(name='dest', x=[0.0], h=src.h[:1])
    src2 = get_particle_array(name='src', x=x, m=np.ones_like(x), h=np.ones_like(x) * dx)
    dm = DomainManager(xmin=-dx / 2, xmax=1.0 + dx / 2, periodic_in_x=True)
    ev2 = SPHEvaluator(arrays=[dest2, src2], equations=eqs, dim=1, domain_manager=dm)
    ev2.evaluate()
    assert abs(dest2.rho[0] - 9.0) < 0.5e-2
    # new particle arrays: the destination moved to the end of the line
    rho0 = dest.rho[0]
    dest.x[0] = 0.0
    ev.update_particle_arrays([dest, src])
    ev.evaluate()
    assert dest.rho[0] != rho0
    assert abs(dest.rho[0] - 7.0) < 0.5e-1


def test_degenerate_h_gives_unit_cell_size(oracle):
    """The reference's ten-particle fixture with h = 0 (test_nnps.py:26-115):
    cell size falls back to 1.0, the grid scalars equal the oracle's, nobody
    has neighbours (r2 < (k*0)^2 never holds)."""
    from pysph_amd import device as dev
    from pysph_amd.nnps import HipNNPS
    from test_oracle_golden import _ten_particles
    pa = _ten_particles()
    ref = _ten_particles()
    nnps = HipNNPS(3, [pa], radius_scale=1.0, ctx=dev.HipContext(0))
    onn = oracle.OracleNNPS(3, [ref], radius_scale=1.0)
    onn.update()
    assert nnps.cell_size == 1.0 == onn.cell_size
    assert np.array_equal(nnps.xmin, onn.xmin) and np.array_equal(nnps.xmax, onn.xmax)
    assert np.array_equal(nnps.ncells_per_dim, onn.ncells_per_dim)
    start, idx = nnps.get_csr(0, 0)
    assert start[-1] == 0 and idx.size == 0


def _correction_case():
    """jittered 9^3 cube carrying the linear field p = 2x - 3y + z + 1"""
    from pysph_amd.particle_array import get_particle_array_wcsph
    rng = np.random.default_rng(21)
    n1 = 9
    dx = 1.0 / n1
    g = (np.arange(n1) + 0.5) * dx
    x, y, z = [a.ravel() for a in np.meshgrid(g, g, g, indexing='ij')]
    n = x.size
    x = x + 0.15 * dx * rng.uniform(-1, 1, n)
    y = y + 0.15 * dx * rng.uniform(-1, 1, n)
    z = z + 0.15 * dx * rng.uniform(-1, 1, n)
    pa = get_particle_array_wcsph(name='fluid', x=x, y=y, z=z, h=1.3 * dx * np.ones(n),
                                  m=dx ** 3 * np.ones(n), rho=np.ones(n),
                                  p=2 * x - 3 * y + z + 1)
    for name, stride in (('lmat', 9), ('amat', 16), ('bvec', 4), ('pfit', 4)):
        pa.add_property(name, stride=stride)
        pa.properties[name][:] = rng.uniform(-1, 1, stride * n)     # junk to be overwritten
    for name in ('gx', 'gy', 'gz'):
        pa.add_property(name)
    return pa


def _correction_equations():
    from custom_equations import (CorrectGradient, CorrectionMatrix, GradientOfLinearField,
                                  MomentMatrix, SolveMoments)
    from pysph_amd.equations import Group
    return [Group(equations=[CorrectionMatrix('fluid', ['fluid'], dim=3)]),
            Group(equations=[CorrectGradient('fluid', ['fluid'], dim=3, tol=50.0),
                             GradientOfLinearField('fluid', ['fluid'])]),
            Group(equations=[MomentMatrix('fluid', ['fluid']),
                             SolveMoments('fluid', ['fluid'], dim=3)])]


def test_generated_matrix_helpers_unrolled_components_and_symbol_rewrite(oracle):
    """The constructs of the reference's kernel-correction and interpolation
    equations (kernel_correction.py:40-125, bc/interpolate.py:263-380): strided
    properties addressed through loops and int locals (unrolled at translation
    time), helper functions with matrix / int arguments, SPH_KERNEL.gradient in
    loop_all, and an equation that REWRITES DWIJ for the ones after it.
    Oracle: the same Python bodies run by oracle/py_eval.py; plus the analytic
    property that both corrections reproduce a linear field exactly."""
    from oracle.py_eval import PyEval
    from pysph_amd import kernels as K
    kernel = K.CubicSpline(dim=3)
    eqs = _correction_equations()
    pa, ref = _correction_case(), _correction_case()
    a_eval, nnps, ctx = make_eval([pa], eqs, kernel, 3)
    a_eval.compute(0.0, 1e-4)
    onn = oracle.OracleNNPS(3, [ref], radius_scale=2.0)
    onn.update()
    PyEval([ref], eqs, kernel, onn).compute(0.0, 1e-4)
    assert rel_err(pa.lmat, ref.lmat) < TOL
    assert rel_err(pa.amat, ref.amat) < TOL and rel_err(pa.bvec, ref.bvec) < TOL
    # the linear solves amplify last-bit differences (fma contraction) by the
    # condition number of the local matrices
    for prop in ('gx', 'gy', 'gz', 'pfit'):
        assert rel_err(pa.properties[prop], ref.properties[prop]) < 1e-8, prop
    # analytic: corrected gradient of the linear field, and its MLS fit
    assert np.abs(pa.gx - 2.0).max() < 1e-9 and np.abs(pa.gy + 3.0).max() < 1e-9
    assert np.abs(pa.gz - 1.0).max() < 1e-9
    fit = pa.pfit.reshape(-1, 4)
    assert np.abs(fit[:, 0] - pa.p).max() < 1e-9
    assert np.abs(fit[:, 1:] - np.array([2.0, -3.0, 1.0])).max() < 1e-8


def _random_generated_case(seed):
    """problem shape drawn from the seed; the equations are the made-up ones of
    tests/custom_equations.py (every translator construct)"""
    from custom_equations import KitchenSink, PowerLawState, WallPush
    from pysph_amd.equations import Group
    from pysph_amd.particle_array import get_particle_array
    rng = np.random.default_rng(5000 + seed)
    dim = int(rng.choice([1, 2, 3], p=[0.2, 0.4, 0.4]))
    n = int(rng.choice([1, 5, 60, 250]))
    kname = str(rng.choice(['CubicSpline', 'WendlandQuintic', 'QuinticSpline', 'Gaussian']))
    if kname == 'WendlandQuintic' and dim == 1:
        kname = 'QuinticSpline'
    varh = float(rng.choice([0.0, 0.0, 0.2]))
    clustered = bool(rng.integers(2))
    spacing = 1.0 / max(n, 2) ** (1.0 / dim)
    arrays = []
    for name, m in (('fluid', n), ('solid', int(rng.choice([0, 1, max(n // 4, 1)])))):
        c = rng.uniform(0, 1, (m, 3))
        if clustered:
            c = c ** 2.0
        c[:, dim:] = 0.0
        pa = get_particle_array(
            name=name, constants=dict(coef=np.array([1.25, -0.5])),
            x=c[:, 0], y=c[:, 1], z=c[:, 2],
            u=rng.uniform(-1, 1, m), v=rng.uniform(-1, 1, m), w=rng.uniform(-1, 1, m),
            h=1.2 * spacing * (1 + varh * rng.uniform(-1, 1, m)),
            m=spacing ** dim * np.ones(m), rho=1 + 0.1 * rng.uniform(-1, 1, m),
            additional_props=['q', 'gx', 'gy', 'gz', 'e'])
        for k in ('q', 'gx', 'gy', 'gz', 'e', 'p'):
            pa.properties[k][:] = rng.uniform(1, 2, m)
        arrays.append(pa)
    real = bool(rng.integers(2))
    start = int(rng.integers(0, max(n // 3, 1)))
    stop = None if rng.integers(2) else int(rng.integers(start, n + 1))
    eqs = [
        Group(real=False, equations=[PowerLawState('fluid', None, k=1.5, n=1.4),
                                     PowerLawState('solid', None, k=0.5, n=2.0)]),
        Group(real=real, start_idx=start, stop_idx=stop, equations=[
            KitchenSink('fluid', ['fluid', 'solid'], a=0.3, b=0.05, flag=bool(rng.integers(2))),
            WallPush('fluid', ['solid'], c=0.7)]),
        Group(equations=[KitchenSink('solid', ['fluid'], a=0.1, b=2.0, flag=False)]),
    ]
    return arrays, eqs, dim, kname


@pytest.mark.parametrize('seed', list(range(int(os.environ.get('SPH_FUZZ_SEEDS', '12')))))
def test_randomised_generated_families_vs_python(oracle, seed):
    """the generated-family path under randomly drawn dimension / kernel /
    particle distribution / per-particle h / empty and one-particle arrays /
    Group(real, start_idx, stop_idx), against the same Python bodies run by
    oracle/py_eval.py"""
    from oracle.py_eval import PyEval
    from pysph_amd import kernels as K
    arrays, eqs, dim, kname = _random_generated_case(seed)
    ref = _copy_arrays(arrays)
    for r, a in zip(ref, arrays):
        r.constants = dict((k, v.copy()) for k, v in a.constants.items())
    kernel = getattr(K, kname)(dim=dim)
    a_eval, nnps, ctx = make_eval(arrays, eqs, kernel, dim)
    a_eval.compute(0.25, 1e-3)
    onn = oracle.OracleNNPS(dim, ref, radius_scale=kernel.radius_scale)
    onn.update()
    PyEval(ref, eqs, kernel, onn).compute(0.25, 1e-3)
    for pa, pr in zip(arrays, ref):
        for prop in ('p', 'e', 'q', 'gx', 'gy', 'gz'):
            got, want = pa.properties[prop], pr.properties[prop]
            if got.size:
                scale = max(np.abs(want).max(), 1e-300)
                assert np.abs(got - want).max() <= TOL * scale, (seed, dim, kname, pa.name, prop)


F32_GENERATED_CASES = ['custom-0', 'custom-0.15', 'wcsph', 'random-0', 'random-3', 'random-7', 'random-9']


def _f32_generated_case(case):
    from pysph_amd import kernels as K
    if case.startswith('custom-'):
        arrays, eqs = _custom_setup(float(case[7:]))
        return arrays, eqs, K.CubicSpline(dim=3), 3, ('p', 'e', 'q', 'gx', 'gy', 'gz')
    if case.startswith('random-'):
        arrays, eqs, dim, kname = _random_generated_case(int(case[7:]))
        return arrays, eqs, getattr(K, kname)(dim=dim), dim, ('p', 'e', 'q', 'gx', 'gy', 'gz')
    from custom_equations import PyContinuity, PyMomentum, PyXSPH
    from pysph_amd.equations import Group
    pa, dx = make_cube(14)
    rng = np.random.default_rng(3)
    for f in 'uvw':
        pa.properties[f][:] = rng.uniform(-1, 1, pa.x.size)
    pa.rho[:] = 1000.0 * (1 + 0.01 * rng.uniform(-1, 1, pa.x.size))
    pa.p[:] = 1e3 * rng.uniform(0, 1, pa.x.size)
    kw = dict(c0=32.85, alpha=0.25, beta=0.1, gz=-9.81, tensile_correction=False)
    eqs = [Group(equations=[PyContinuity(dest='fluid', sources=['fluid']),
                            PyMomentum(dest='fluid', sources=['fluid'], **kw),
                            PyXSPH(dest='fluid', sources=['fluid'], eps=0.5)])]
    return [pa], eqs, K.WendlandQuintic(dim=3), 3, WC_OUT


def _wcsph_hand_equations():
    """the library equations the Python bodies of the 'wcsph' case restate (run by the C oracle: py_eval has no WDP)"""
    from pysph_amd.equations import ContinuityEquation, Group, MomentumEquation, XSPHCorrection
    kw = dict(c0=32.85, alpha=0.25, beta=0.1, gz=-9.81, tensile_correction=False)
    return [Group(equations=[ContinuityEquation(dest='fluid', sources=['fluid']),
                             MomentumEquation(dest='fluid', sources=['fluid'], **kw),
                             XSPHCorrection(dest='fluid', sources=['fluid'], eps=0.5)])]


@pytest.mark.parametrize('case', F32_GENERATED_CASES)
def test_generated_families_fp32_arithmetic_vs_python(oracle, case):
    """Option arith_f32 for GENERATED families: the pair
    launch runs the family's float build -- `codegen.source_f32`: fp32 records,
    every local / pair symbol / kernel value / literal / accumulator a float,
    arrays narrowed on load and widened on store -- as the reference's generated
    OpenCL / CUDA code does without --use-double
    (acceleration_eval_gpu_helper.py:281-283,437-441).  Checked against the same
    Python bodies run in fp64 by oracle/py_eval.py at 5e-5 of each field's
    maximum (the tolerance of the hand-written fp32 families), on the made-up
    equations of tests/custom_equations.py (uniform and per-particle h, two
    sources, no-source equations with parameters) and on WCSPH written as
    Python bodies (against the C oracle running the library equations they
    restate); and it must really be another precision."""
    from oracle.py_eval import PyEval
    arrays, eqs, kernel, dim, outs = _f32_generated_case(case)
    ref = _copy_arrays(arrays)
    for r, a in zip(ref, arrays):
        r.constants = dict((k, v.copy()) for k, v in getattr(a, 'constants', {}).items())
    a_eval, nnps, ctx = make_eval(arrays, eqs, kernel, dim)
    ctx.set_option('arith_f32', 1)
    a_eval.compute(0.25, 1e-3)
    onn = oracle.OracleNNPS(dim, ref, radius_scale=kernel.radius_scale)
    onn.update()
    if case == 'wcsph':
        oev = oracle.OracleEval(ref, _wcsph_hand_equations(), kernel, nthreads=4)
        oev.set_nnps(onn)
        oev.compute(0.25, 1e-3)
    else:
        PyEval(ref, eqs, kernel, onn).compute(0.25, 1e-3)
    worst = 0.0
    for pa, pr in zip(arrays, ref):
        for prop in outs:
            got, want = pa.properties[prop], pr.properties[prop]
            if got.size:
                scale = max(np.abs(want).max(), 1e-300)
                e = np.abs(got - want).max() / scale
                worst = max(worst, e)
                assert e <= 5e-5, (case, pa.name, prop, e)
    print('generated arith_f32 %s: max err %.3e of the field maximum' % (case, worst))
    if any(pa.get_number_of_particles() > 4 for pa in arrays):
        assert worst > 1e-10, 'the float build was not the one that ran'


def _image_case():
    from pysph_amd.particle_array import get_particle_array
    rng = np.random.default_rng(2)
    n = 500
    pa = get_particle_array(name='fluid', x=rng.uniform(0, 1, n), h=0.05 * np.ones(n),
                            rho=rng.uniform(1, 2, n), p=rng.uniform(1, 2, n))
    pa.add_property('q')
    pa.add_property('image')
    pa.add_property('orig_idx', type='int')
    pa.image[300:] = 1.0                              # the last 200 are images ...
    pa.orig_idx[:] = np.arange(n)
    pa.orig_idx[300:] = rng.integers(0, 300, 200)     # ... of particles among the first 300
    return pa


def _image_equations():
    from custom_equations import CopyFromOriginal
    from pysph_amd.equations import Group
    return [Group(equations=[CopyFromOriginal('fluid', None)], real=False)]


def test_generated_read_of_destination_at_runtime_index(oracle):
    """d_rho[idx] with idx read from an INTEGER property: the image particles
    end up with the values of their originals (which the launch does not write)"""
    from oracle.py_eval import PyEval
    from pysph_amd import kernels as K
    kernel = K.CubicSpline(dim=1)
    pa, ref = _image_case(), _image_case()
    a_eval, nnps, ctx = make_eval([pa], _image_equations(), kernel, 1)
    a_eval.compute(0.0, 0.1)
    onn = oracle.OracleNNPS(1, [ref], radius_scale=2.0)
    onn.update()
    PyEval([ref], _image_equations(), kernel, onn).compute(0.0, 0.1)
    for prop in ('rho', 'p', 'q'):
        assert np.array_equal(pa.properties[prop], ref.properties[prop]), prop
    assert np.array_equal(pa.rho[300:], pa.rho[pa.orig_idx[300:]])
    assert pa.orig_idx.dtype.kind == 'i' and np.array_equal(pa.orig_idx, ref.orig_idx)


def _gradh_case(kname):
    from custom_equations import DensityWithGradH
    from pysph_amd import kernels as K
    from pysph_amd.equations import Group
    from pysph_amd.particle_array import get_particle_array
    rng = np.random.default_rng(8)
    n1 = 9
    dx = 1.0 / n1
    g = (np.arange(n1) + 0.5) * dx
    x, y, z = [a.ravel() for a in np.meshgrid(g, g, g, indexing='ij')]
    n = x.size
    pa = get_particle_array(name='fluid', x=x + 0.1 * dx * rng.uniform(-1, 1, n),
                            y=y + 0.1 * dx * rng.uniform(-1, 1, n), z=z + 0.1 * dx * rng.uniform(-1, 1, n),
                            h=1.2 * dx * (1 + 0.2 * rng.uniform(-1, 1, n)), m=dx ** 3 * np.ones(n))
    pa.add_property('dwdh')
    pa.add_property('q')
    eqs = [Group(equations=[DensityWithGradH('fluid', ['fluid'])])]
    return pa, eqs, getattr(K, kname)(dim=3)


@pytest.mark.parametrize('kname', ['CubicSpline', 'WendlandQuintic', 'QuinticSpline', 'Gaussian'])
def test_generated_gradh_symbols_vs_python(oracle, kname):
    """GHI / GHJ / GHIJ = dW/dh (equation.py:285-295, kernels.py gradient_h) with
    per-particle h, all four kernels"""
    from oracle.py_eval import PyEval
    pa, eqs, kernel = _gradh_case(kname)
    ref, _, _ = _gradh_case(kname)
    a_eval, nnps, ctx = make_eval([pa], eqs, kernel, 3)
    a_eval.compute(0.0, 0.1)
    onn = oracle.OracleNNPS(3, [ref], radius_scale=kernel.radius_scale)
    onn.update()
    PyEval([ref], eqs, kernel, onn).compute(0.0, 0.1)
    for prop in ('rho', 'dwdh', 'q'):
        assert rel_err(pa.properties[prop], ref.properties[prop]) < TOL, (kname, prop)
    assert np.abs(ref.dwdh).max() > 0


def _systems_case(n):
    from custom_equations import SolveSystems
    from pysph_amd.equations import Group
    from pysph_amd.particle_array import get_particle_array
    rng = np.random.default_rng(31 + n)
    m = 600
    pa = get_particle_array(name='fluid', x=np.linspace(0, 1, m), h=0.01 * np.ones(m))
    for name, stride in (('amat', 16), ('bvec', 4), ('pfit', 4)):
        pa.add_property(name, stride=stride)
    pa.add_property('q')
    A = rng.uniform(-1, 1, (m, 4, 4))
    A[::3, 0, 0] = 0.0                     # a zero pivot: needs row exchange
    A[5] = 0.0                             # one singular system
    pa.amat[:] = A.ravel()
    pa.bvec[:] = rng.uniform(-1, 1, 4 * m)
    return pa, [Group(equations=[SolveSystems('fluid', None, n=n)])]


@pytest.mark.parametrize('n', [2, 3, 4])
def test_generated_linear_solver_helpers_vs_numpy(n):
    """sph/tests/test_linalg.py on the device: 600 systems per launch, the
    leading n x n block of a 4 x 4 matrix (augmented_matrix 'lower dimension'
    cases), zero pivots needing row exchange, one singular matrix"""
    from pysph_amd import kernels as K
    pa, eqs = _systems_case(n)
    a_eval, nnps, ctx = make_eval([pa], eqs, K.CubicSpline(dim=1), 1)
    a_eval.compute(0.0, 0.1)
    A = pa.amat.reshape(-1, 4, 4)[:, :n, :n]
    b = pa.bvec.reshape(-1, 4)[:, :n]
    x = pa.pfit.reshape(-1, 4)
    assert pa.q[5] == 1.0 and np.all(np.delete(pa.q, 5) == 0.0)      # singular flag
    ok = np.ones(pa.q.size, dtype=bool)
    ok[5] = False
    want = np.linalg.solve(A[ok], b[ok][..., None])[..., 0]
    cond = np.linalg.cond(A[ok])
    err = np.abs(x[ok][:, :n] - want).max(axis=1) / np.abs(want).max(axis=1)
    assert np.all(err < 1e-12 * np.maximum(cond, 10.0))
    assert np.all(x[ok][:, n:] == 0.0)


def test_tvf_two_slabs_two_halo_layers_one_gpu(oracle):
    """The TVF set across two slab ranks: its first group is a
    real=False summation density, i.e. V and rho of the GHOSTS are recomputed
    locally, which needs the neighbours of the ghosts -- a halo two support
    radii wide (the reason the reference defaults to two ghost layers).  The
    inner ghost layer then carries correct V, rho, p for the force group; results
    of all real particles equal the single-domain oracle by global id."""
    import threading
    import torch
    from helpers import ThreadDist
    from test_periodic import lattice
    from pysph_amd import device as dev
    from pysph_amd import kernels as K
    from pysph_amd.acceleration_eval import AccelerationEval, SPHCompiler
    from pysph_amd.nnps import HipNNPS
    from pysph_amd.parallel import TVF_HALO_PROPS, SlabHalo
    from pysph_amd.scheme import TVFScheme
    full, dx = lattice(14, dim=3, hdx=1.0, jitter=0.08)
    rng = np.random.default_rng(12)
    n = full.get_number_of_particles()
    for k in 'uvw':
        full.properties[k][:] = rng.uniform(-1, 1, n)
        full.properties[k + 'hat'][:] = full.properties[k] + 0.1 * rng.uniform(-1, 1, n)
    full.add_property('e0', data=np.arange(n, dtype=np.float64))
    kernel = K.QuinticSpline(dim=3)
    eqs = TVFScheme(['fluid'], [], dim=3, rho0=1.0, c0=10.0, nu=0.01, p0=100.0, pb=100.0,
                    h0=dx).get_equations()
    ref = _copy_arrays([full])
    cut = 0.5
    width = 2.0 * kernel.radius_scale * dx * 1.02          # two support radii
    outs = ['rho', 'V', 'p', 'au', 'av', 'aw', 'auhat', 'avhat', 'awhat']
    hub = ThreadDist(2)
    results, errors = {}, []

    def rank_main(rank):
        try:
            ts = torch.cuda.Stream()
            with torch.cuda.stream(ts):
                lo, hi = (-1e30, cut) if rank == 0 else (cut, 1e30)
                pa = full.extract_particles(np.nonzero((full.x >= lo) & (full.x < hi))[0], name='fluid')
                ctx = dev.HipContext(0, ts.cuda_stream)
                dev.attach(pa, ctx).push()
                halo = SlabHalo(pa, ctx, rank, 2, axis=0, width=width, lo=lo, hi=hi,
                                props=TVF_HALO_PROPS, dist=hub.view(rank))
                halo.exchange()
                a_eval = AccelerationEval([pa], eqs, kernel)
                SPHCompiler(a_eval, ctx=ctx, sync='manual').compile()
                nnps = HipNNPS(3, [pa], radius_scale=kernel.radius_scale, ctx=ctx, sync=False)
                a_eval.set_nnps(nnps)
                a_eval.compute(0.0, 1e-4)
                nreal = pa.gpu.get_number_of_particles(True)
                assert pa.gpu.get_number_of_particles() > nreal
                pa.gpu.sync_host()
                results[rank] = {k: pa.properties[k][:nreal].copy() for k in outs + ['e0']}
        except Exception:
            import traceback
            errors.append(traceback.format_exc())
            try:
                hub.barrier.abort()
            except Exception:
                pass

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    assert not errors, errors[0]
    onn = oracle.OracleNNPS(3, ref, radius_scale=kernel.radius_scale)
    onn.update()
    oev = oracle.OracleEval(ref, eqs, kernel, nthreads=8)
    oev.set_nnps(onn)
    oev.compute(0.0, 1e-4)
    seen = 0
    for r in range(2):
        d = results[r]
        gid = d['e0'].astype(np.int64)
        seen += gid.size
        for prop in outs:
            want = ref[0].properties[prop]
            e = rel_err(d[prop], want[gid], scale=max(np.abs(want).max(), 1e-300))
            assert e < TOL, (r, prop, e)
    assert seen == n


@pytest.mark.gpu
def test_shared_records_are_repacked_after_a_no_source_equation(oracle):
    """One group, two destinations, hand-written kernels only: the library packs
    every source array once for the group (option pack_group).  Here the second
    destination's EOS runs AFTER the first destination packed that array and
    changes p and cs, which its own record carries: the stale copy must not be
    used.  Reference order (acceleration_eval_cython.mako:50-154): per
    destination, no-source equations first, then the source loops -- so `a` sees
    b's OLD pressure and `b` sees both new ones; the oracle does the same."""
    from pysph_amd.equations import (ContinuityEquation, Group, MomentumEquation,
                                     TaitEOS, XSPHCorrection)
    from pysph_amd.examples import dam_break_3d as db
    from pysph_amd import kernels as K
    from pysph_amd.particle_array import ParticleArray
    full, dx = make_cube(16, seed=5)
    x = full.x
    halves = []
    for name, sel in (('a', x < 0.5), ('b', x >= 0.5)):
        idx = np.nonzero(sel)[0]
        pa = ParticleArray(name=name, **{k: v[idx].copy() for k, v in full.properties.items()})
        pa.constants = dict(getattr(full, 'constants', {}))
        halves.append(pa)
    both = ['a', 'b']
    eos = dict(rho0=db.ro, c0=db.c0, gamma=db.gamma)
    mom = dict(c0=db.c0, alpha=db.alpha, beta=db.beta, gz=-9.81)
    eqs = [Group(equations=[
        TaitEOS(dest='a', sources=None, **eos),
        ContinuityEquation(dest='a', sources=both),
        MomentumEquation(dest='a', sources=both, **mom),
        XSPHCorrection(dest='a', sources=['a']),
        TaitEOS(dest='b', sources=None, **eos),
        ContinuityEquation(dest='b', sources=both),
        MomentumEquation(dest='b', sources=both, **mom),
        XSPHCorrection(dest='b', sources=['b']),
    ])]
    for pa in halves:            # p, cs start wrong everywhere: only an EOS that ran makes them right
        pa.p[:] = 123.0
        pa.cs[:] = 3.0
    ref = _copy_arrays(halves)
    kernel = K.WendlandQuintic(dim=3)
    a_eval, nnps, ctx = make_eval(halves, eqs, kernel, 3, 6)
    a_eval.compute(0.0, 1e-5)
    onn = oracle.OracleNNPS(3, ref, radius_scale=kernel.radius_scale)
    onn.update()
    oev = oracle.OracleEval(ref, eqs, kernel, nthreads=8)
    oev.set_nnps(onn)
    oev.compute(0.0, 1e-5)
    for pa, pr in zip(halves, ref):
        for prop in WC_OUT:
            e = rel_err(pa.properties[prop], pr.properties[prop])
            assert e < TOL, (pa.name, prop, e)
    # the case really distinguishes stale from fresh: with b's old p / cs its own sums differ
    assert abs(ref[1].p - 123.0).max() > 1.0


@pytest.mark.parametrize('seed', list(range(int(os.environ.get('SPH_FUZZ_SEEDS', '12')))))
def test_randomised_device_resident_second_evaluation_vs_oracle(oracle, seed):
    """Round 4: the record layouts and launch shapes that only engage on device-resident state from the
    SECOND evaluation on (once the neighbour update has looked at the masses) -- the merged order of
    several arrays, uniform-mass / variable-h / state-fused records -- over randomly drawn problem shapes:
    2-D or 3-D, one fluid + up to two solid arrays (one may be empty), clustered or uniform positions,
    uniform or varying h, equal / per-class / mixed masses, trailing ghost particles (sources only),
    `WCSPHScheme` with or without the HG correction.  The second evaluation must match the oracle's single
    one (the equation set is idempotent on its inputs), neighbour lists included."""
    from pysph_amd import kernels as K
    from pysph_amd.particle_array import get_particle_array_wcsph
    from pysph_amd.scheme import WCSPHScheme
    rng = np.random.default_rng(9000 + seed)
    dim = int(rng.choice([2, 3]))
    n = int(rng.choice([400, 3000, 8000]))
    kname = str(rng.choice(['CubicSpline', 'WendlandQuintic', 'QuinticSpline']))
    kernel = getattr(K, kname)(dim=dim)
    varh = float(rng.choice([0.0, 0.0, 0.0, 0.2]))
    clustered = bool(rng.integers(0, 2))
    spacing = (1.0 / n) ** (1.0 / dim)
    mass_mode = str(rng.choice(['equal', 'equal', 'per-class', 'mixed']))
    sizes = {'fluid': n, 'wall': int(rng.choice([0, max(n // 4, 1), max(n // 2, 1)])),
             'block': int(rng.choice([0, 0, 5, max(n // 10, 1)]))}
    arrays = []
    for name, m in sizes.items():
        c = rng.uniform(0, 1, (m, 3))
        if clustered:
            c = c ** 2.0
        if name != 'fluid' and m:
            c[:, dim - 1] *= 0.15            # the solids hug one face: real fluid-solid neighbourhoods
        c[:, dim:] = 0.0
        mass = spacing ** dim * (1.0 if name == 'fluid' or mass_mode == 'equal' else 1.5)
        mvec = mass * np.ones(m)
        if mass_mode == 'mixed' and name == 'wall' and m:
            mvec = mvec * (1 + 0.3 * rng.uniform(-1, 1, m))
        pa = get_particle_array_wcsph(
            name=name, x=c[:, 0], y=c[:, 1], z=c[:, 2], h=1.3 * spacing * (1 + varh * rng.uniform(-1, 1, m)),
            m=mvec, rho=1000.0 * (1 + 0.02 * rng.uniform(-1, 1, m)), u=rng.uniform(-1, 1, m),
            v=rng.uniform(-1, 1, m), w=rng.uniform(-1, 1, m))
        if m > 10 and rng.integers(0, 2):
            pa.set_num_real_particles(m - int(rng.integers(1, m // 3)))   # trailing ghosts: sources only
        arrays.append(pa)
    solids = ['wall', 'block']
    scheme = WCSPHScheme(['fluid'], solids, dim=dim, rho0=1000.0, c0=10.0, h0=1.3 * spacing, hdx=1.3,
                         gamma=7.0, alpha=0.3, beta=0.1, gy=-1.0, hg_correction=bool(rng.integers(0, 2)))
    eqs = scheme.get_equations()
    ref = _copy_arrays(arrays)
    onn = oracle.OracleNNPS(dim, ref, radius_scale=kernel.radius_scale)
    onn.update()
    oev = oracle.OracleEval(ref, eqs, kernel, nthreads=4)
    oev.set_nnps(onn)
    oev.compute(0.1, 1e-4)
    q = _copy_arrays(arrays)
    a_eval, nnps, ctx = make_eval(q, eqs, kernel, dim, 6, sync='manual')
    for pa in q:
        pa.gpu.push()
    nnps.sync = False
    nnps.update()
    a_eval.compute(0.1, 1e-4)
    nnps.update()                      # this one looks at the masses
    a_eval.compute(0.1, 1e-4)
    took = {k: ctx.timer_get(k)[1] for k in ('n_merged', 'n_mass_fused', 'n_eos_fused')}
    a_eval.c_acceleration_eval.pull_outputs()
    for pa in q:
        pa.gpu.pull('rho', 'p', 'cs')
    print('seed %d: dim %d n %s %s varh %g masses %s -> %s' % (seed, dim, sizes, kname, varh, mass_mode, took))
    for si in range(3):
        for di in range(3):
            if q[di].get_number_of_particles() == 0:
                continue
            s1 = nnps.get_csr_start(si, di)
            s2, _ = onn.get_csr(si, di)
            assert np.array_equal(s1, s2), (seed, si, di)
    for pa, pr in zip(q, ref):
        nreal = pr.get_number_of_particles(True)
        for prop in WC_OUT + ['rho']:
            a, b = np.asarray(pa.properties[prop]), np.asarray(pr.properties[prop])
            # p, cs and the clamped rho are recomputed for every particle (real=False group), the rates for the real ones
            upto = a.size if prop in ('p', 'cs', 'rho') else nreal
            e = rel_err(a[:upto], b[:upto])
            assert e < TOL, (seed, dim, sizes, kname, varh, mass_mode, took, pa.name, prop, e)


def test_ghost_outside_the_h_range_of_the_update_is_rejected():
    """the cell size and the uniform-h path of an update rest on the h range of the particles it saw: a ghost
    binned afterwards with a larger h would lose neighbours silently -- it is an error (advisor, round 4)"""
    from pysph_amd import device as dev
    from pysph_amd.nnps import HipNNPS
    from pysph_amd.particle_array import get_particle_array_wcsph
    rng = np.random.default_rng(5)
    n, nreal = 600, 500
    c = rng.uniform(0, 1, (n, 3))
    c[nreal:, 0] += 1.0
    h = 0.1 * np.ones(n)
    h[-1] = 0.13
    pa = get_particle_array_wcsph(name='fluid', x=c[:, 0], y=c[:, 1], z=c[:, 2], h=h, m=np.ones(n), rho=np.ones(n))
    pa.set_num_real_particles(nreal)
    ctx = dev.HipContext(0)
    dev.attach(pa, ctx).push()
    dev._check(ctx.lib.sph_array_resize(ctx._h, pa.gpu.array_id, nreal, nreal))
    nnps = HipNNPS(3, [pa], radius_scale=2.0, ctx=ctx, sync=False)
    dev._check(ctx.lib.sph_array_resize(ctx._h, pa.gpu.array_id, n, nreal))
    with pytest.raises(dev.SphError, match='smoothing length outside'):
        nnps.update_ghosts(0, 0.0, 1.0)
    ctx.close()


@pytest.mark.parametrize('merge', [1, 0], ids=['merged-first', 'per-array'])
def test_ghost_segments_three_arrays_lazy_tables_vs_oracle(oracle, merge):
    """The ghost split with SEVERAL arrays: the neighbour update bins the real particles of the dam-break
    tank's three arrays in one merged order (per-array tables are lazy), the ghosts of every array arrive behind them
    and are binned into segments of their own, and the evaluation -- which cannot take the one-launch merged path
    with ghost segments -- asks for the per-array tables: they must cover the particles that were BINNED (n_binned),
    not the arrays' present sizes.  Real particles of all three arrays against the oracle on reals + ghosts, neighbour
    counts included; twice, so that the second round runs on kept buffers."""
    from pysph_amd import device as dev
    from pysph_amd.examples import dam_break_3d as db
    dx = 0.07
    full = db.create_particles(dx)
    _perturb(full, 11, db.c0, db.ro, dx)
    kernel = db.create_kernel()
    eqs = db.create_scheme(dx).get_equations()
    width = kernel.radius_scale * db.hdx * dx
    cut = float(np.median(np.concatenate([a.x for a in full])))
    arrays, nreal = [], []
    for a in full:                  # this rank: x < cut; its ghosts: the next halo width
        real = np.nonzero(a.x < cut)[0]
        ghost = np.nonzero((a.x >= cut) & (a.x < cut + width))[0]
        b = a.extract_particles(np.concatenate([real, ghost]), name=a.name)
        b.set_num_real_particles(real.size)
        arrays.append(b)
        nreal.append(int(real.size))
    assert sum(a.get_number_of_particles() - r for a, r in zip(arrays, nreal)) > 100 and min(nreal) >= 0
    ref = _copy_arrays(arrays)
    onn = oracle.OracleNNPS(3, ref, radius_scale=kernel.radius_scale)
    onn.update()
    oev = oracle.OracleEval(ref, eqs, kernel, nthreads=4)
    oev.set_nnps(onn)
    oev.compute(0.0, 1e-5)
    a_eval, nnps, ctx = make_eval(arrays, eqs, kernel, 3, 6, sync='manual')
    ctx.set_option('merge_arrays', merge)
    ctx.set_option('lazy_tables', 1)
    lib = ctx.lib
    for a in arrays:
        a.gpu.push()
    nnps.sync = False
    for rep in range(2):
        for a, r in zip(arrays, nreal):
            dev._check(lib.sph_array_resize(ctx._h, a.gpu.array_id, r, r))
        nnps.set_ghost_faces(0, -1e30, cut)
        nnps.update()
        for a, r in zip(arrays, nreal):
            dev._check(lib.sph_array_resize(ctx._h, a.gpu.array_id, a.get_number_of_particles(), r))
        nnps.update_ghosts(0, -1e30, cut)
        a_eval.compute(0.0, 1e-5)
    a_eval.c_acceleration_eval.pull_outputs()
    for i, (a, r) in enumerate(zip(arrays, nreal)):
        for j in range(len(arrays)):
            s1 = nnps.get_csr_start(j, i)
            s2, _ = onn.get_csr(j, i)
            assert np.array_equal(np.diff(s1.astype(np.int64))[:r], np.diff(s2.astype(np.int64))[:r]), (a.name, j)
    checked = 0
    for a, q, r in zip(arrays, ref, nreal):
        for prop in WC_OUT:
            if prop in ('p', 'cs', 'rho') or prop not in a.properties:
                continue
            want = np.asarray(q.properties[prop])[:r]
            if r and np.abs(want).max() > 0:
                e = rel_err(np.asarray(a.properties[prop])[:r], want)
                assert e < TOL, (a.name, prop, e)
                checked += 1
    assert checked >= 8
    ctx.close()


@pytest.mark.parametrize('seed', list(range(int(os.environ.get('SPH_FUZZ_SEEDS', '12')))))
def test_randomised_ghost_segments_vs_oracle(oracle, seed):
    """Round 4, ghost split (sph_nnps_update_ghosts): the neighbour update bins the REAL particles, the ghosts
    that "arrive" afterwards are binned on the same grid into tables of their own and read by the pair kernel
    as a second source segment -- guarded per wavefront when they lie beyond slab faces along x, unguarded
    along another axis, with the grid widened for them or not (ghosts outside it are clamped into its
    outermost cells).  One evaluation and the neighbour counts against the oracle on reals + ghosts."""
    from pysph_amd import device as dev
    from pysph_amd import kernels as K
    from pysph_amd.particle_array import get_particle_array_wcsph
    rng = np.random.default_rng(7000 + seed)
    nreal = int(rng.choice([300, 2500, 7000]))
    ng = int(rng.choice([1, nreal // 10, nreal // 3]))
    axis = int(rng.choice([0, 0, 1, 2]))
    varh = float(rng.choice([0.0, 0.0, 0.2]))
    extend = bool(rng.integers(0, 2))
    spacing = (1.0 / nreal) ** (1.0 / 3.0)
    width = 2.6 * spacing * (1 + varh)
    c = rng.uniform(0, 1, (nreal + ng, 3))
    # ghosts: beyond the faces [0, 1) of `axis`, within the halo width (both sides)
    side = rng.integers(0, 2, ng)
    c[nreal:, axis] = np.where(side == 0, -width * rng.uniform(0, 1, ng), 1.0 + width * rng.uniform(0, 1, ng))
    n = nreal + ng
    pa = get_particle_array_wcsph(
        name='fluid', x=c[:, 0], y=c[:, 1], z=c[:, 2], h=1.3 * spacing * (1 + varh * rng.uniform(-1, 1, n)),
        m=spacing ** 3 * np.ones(n), rho=1000.0 * (1 + 0.02 * rng.uniform(-1, 1, n)), u=rng.uniform(-1, 1, n),
        v=rng.uniform(-1, 1, n), w=rng.uniform(-1, 1, n))
    pa.set_num_real_particles(nreal)
    eqs = cube_equations(spacing)
    kernel = K.WendlandQuintic(dim=3)
    ref = _copy_arrays([pa])
    onn = oracle.OracleNNPS(3, ref, radius_scale=2.0)
    onn.update()
    oev = oracle.OracleEval(ref, eqs, kernel, nthreads=4)
    oev.set_nnps(onn)
    oev.compute(0.0, 1e-5)
    a_eval, nnps, ctx = make_eval([pa], eqs, kernel, 3, 6, sync='manual')
    lib, h = ctx.lib, pa.gpu
    pa.gpu.push()
    nnps.sync = False
    if varh:
        # ghosts are other ranks' particles: the update that bins the real ones must know the GLOBAL h range (a slab
        # run: fixed_h + h_range_reduce) -- since round 5 sph_nnps_update_ghosts rejects a ghost outside the range
        nnps._h_fixed, nnps._h_range = True, (float(pa.h.min()), float(pa.h.max()))
    for rep in range(2):               # the second round runs on the records that need a look at the masses
        # the step as a slab rank sees it: only the real particles are there when the neighbour update runs ...
        dev._check(lib.sph_array_resize(ctx._h, h.array_id, nreal, nreal))
        nnps.set_extend(*([width if extend and k == axis else 0.0 for k in range(3)]))
        nnps.set_ghost_faces(axis, 0.0, 1.0)
        nnps.update()
        # ... then the ghosts arrive behind them (their rows are still in the buffers: same capacity)
        dev._check(lib.sph_array_resize(ctx._h, h.array_id, n, nreal))
        nnps.update_ghosts(axis, 0.0, 1.0)
        a_eval.compute(0.0, 1e-5)
    a_eval.c_acceleration_eval.pull_outputs()
    pa.gpu.pull('p', 'cs')
    s1 = nnps.get_csr_start(0, 0)
    s2, _ = onn.get_csr(0, 0)
    assert np.array_equal(np.diff(s1.astype(np.int64))[:nreal], np.diff(s2.astype(np.int64))[:nreal]), seed
    assert not np.any(np.diff(s1.astype(np.int64))[nreal:])      # ghosts have no lists
    for prop in WC_OUT:
        a, b = np.asarray(pa.properties[prop]), np.asarray(ref[0].properties[prop])
        upto = n if prop in ('p', 'cs', 'rho') else nreal
        e = rel_err(a[:upto], b[:upto])
        assert e < TOL, (seed, nreal, ng, axis, varh, extend, prop, e)
