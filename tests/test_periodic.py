"""Periodic DomainManager (reference tests
pysph/base/tests/test_domain_manager.py:100-118, test_periodic_nnps.py)."""
import numpy as np
import pytest

from helpers import live_rows, rel_err


def lattice(n1, dim=3, hdx=1.0, name='fluid', jitter=0.0, seed=3):
    from pysph_amd.particle_array import get_particle_array_tvf_fluid
    dx = 1.0 / n1
    g = (np.arange(n1) + 0.5) * dx
    if dim == 2:
        x, y = [a.ravel().copy() for a in np.meshgrid(g, g, indexing='ij')]
        z = np.zeros_like(x)
    else:
        x, y, z = [a.ravel().copy() for a in np.meshgrid(g, g, g, indexing='ij')]
    n = x.size
    rng = np.random.default_rng(seed)
    if jitter:
        x += jitter * dx * rng.uniform(-1, 1, n)
        y += jitter * dx * rng.uniform(-1, 1, n)
        if dim == 3:
            z += jitter * dx * rng.uniform(-1, 1, n)
    pa = get_particle_array_tvf_fluid(
        name=name, x=x, y=y, z=z, h=hdx * dx * np.ones(n),
        m=dx ** dim * np.ones(n), rho=np.ones(n))
    return pa, dx


def test_host_periodic_ghosts_lattice_density(oracle):
    """Uniform lattice in a periodic box: every particle sees the same
    neighbourhood, so summation density is identical for all particles and the
    number density matches 1/vol (test_domain_manager.py:100-118)."""
    from pysph_amd import kernels as K
    from pysph_amd.domain import DomainManager
    from pysph_amd.equations import Group, TVFSummationDensity
    pa, dx = lattice(12, dim=2, hdx=1.2)
    kernel = K.QuinticSpline(dim=2)
    dom = DomainManager(xmin=0, xmax=1, ymin=0, ymax=1, periodic_in_x=True,
                        periodic_in_y=True)
    dom.set_particles([pa], kernel.radius_scale)
    dom.update()
    n, nreal = pa.get_number_of_particles(), pa.get_number_of_particles(True)
    assert nreal == 144 and n > nreal
    assert np.all(pa.tag[:nreal] == 0) and np.all(pa.tag[nreal:] == 2)
    dom.update()                                   # idempotent
    assert pa.get_number_of_particles() == n
    eqs = [Group(equations=[TVFSummationDensity(dest='fluid', sources=['fluid'])])]
    nn = oracle.OracleNNPS(2, [pa], kernel.radius_scale)
    nn.update()
    ev = oracle.OracleEval([pa], eqs, kernel)
    ev.set_nnps(nn)
    ev.compute(0.0, 0.1)
    V, rho = pa.V[:nreal], pa.rho[:nreal]
    assert np.max(np.abs(1.0 / V - dx ** 2)) < 1e-5 * dx ** 2 * 10
    assert np.max(np.abs(V - V[0])) < 1e-11 * V[0]
    assert np.max(np.abs(rho - pa.m[:nreal] * V)) < 1e-14


def test_box_wrap_host():
    from pysph_amd.domain import DomainManager
    pa, dx = lattice(6, dim=2)
    pa.x[0] = -0.01
    pa.y[1] = 1.02
    dom = DomainManager(xmin=0, xmax=1, ymin=0, ymax=1, periodic_in_x=True,
                        periodic_in_y=True)
    dom.set_particles([pa], 3.0)
    dom.update()
    assert abs(pa.x[0] - 0.99) < 1e-15 and abs(pa.y[1] - 0.02) < 1e-15


@pytest.mark.gpu
def test_tvf_periodic_taylor_green_vs_oracle(oracle):
    """BASELINE config 3 at test size: periodic unit cube, TVF equation set
    (taylor_green.py parameters: rho0 1, c0 10, p0 = pb = 100, nu 0.01),
    QuinticSpline.  Host-side ghosts (sync='auto')."""
    from pysph_amd import kernels as K
    from pysph_amd.domain import DomainManager
    from pysph_amd.scheme import TVFScheme
    from test_hip_parity import make_eval, _copy_arrays
    from pysph_amd import device as dev
    from pysph_amd.acceleration_eval import AccelerationEval, SPHCompiler
    from pysph_amd.nnps import HipNNPS
    pa, dx = lattice(12, dim=3, hdx=1.0, jitter=0.05)
    x, y = pa.x, pa.y
    pa.u[:] = -np.cos(2 * np.pi * x) * np.sin(2 * np.pi * y)
    pa.v[:] = np.sin(2 * np.pi * x) * np.cos(2 * np.pi * y)
    pa.uhat[:] = pa.u * 1.01
    pa.vhat[:] = pa.v * 0.99
    kernel = K.QuinticSpline(dim=3)
    eqs = TVFScheme(['fluid'], [], dim=3, rho0=1.0, c0=10.0, nu=0.01, p0=100.0,
                    pb=100.0, h0=dx).get_equations()
    ref = _copy_arrays([pa])
    kw = dict(xmin=0, xmax=1, ymin=0, ymax=1, zmin=0, zmax=1, periodic_in_x=True,
              periodic_in_y=True, periodic_in_z=True)
    # oracle on host-ghosted arrays
    dref = DomainManager(**kw)
    dref.set_particles(ref, kernel.radius_scale)
    dref.update()
    onn = oracle.OracleNNPS(3, ref, kernel.radius_scale)
    onn.update()
    oev = oracle.OracleEval(ref, eqs, kernel, nthreads=4)
    oev.set_nnps(onn)
    oev.compute(0.0, 1e-4)
    # HIP, host DomainManager
    ctx = dev.HipContext(0)
    a_eval = AccelerationEval([pa], eqs, kernel)
    SPHCompiler(a_eval, ctx=ctx).compile()
    nnps = HipNNPS(3, [pa], radius_scale=kernel.radius_scale, ctx=ctx,
                   domain=DomainManager(**kw))
    a_eval.set_nnps(nnps)
    a_eval.compute(0.0, 1e-4)
    nreal = pa.get_number_of_particles(True)
    assert pa.get_number_of_particles() == ref[0].get_number_of_particles()
    for prop in ('rho', 'V', 'p', 'au', 'av', 'aw', 'auhat', 'avhat', 'awhat'):
        e = rel_err(pa.properties[prop][:nreal], ref[0].properties[prop][:nreal])
        assert e < 1e-10, (prop, e)
    # summation density (real=False group) also filled the ghosts
    assert rel_err(pa.rho, ref[0].rho) < 1e-10


@pytest.mark.gpu
def test_device_domain_manager_matches_host(oracle):
    """HipDomainManager (device-resident ghosts) creates the same ghost SET as
    the host DomainManager and the evaluation agrees."""
    from pysph_amd import kernels as K
    from pysph_amd import device as dev
    from pysph_amd.domain import DomainManager, HipDomainManager
    from pysph_amd.scheme import TVFScheme
    from pysph_amd.acceleration_eval import AccelerationEval, SPHCompiler
    from pysph_amd.nnps import HipNNPS
    from test_hip_parity import _copy_arrays
    pa, dx = lattice(10, dim=3, hdx=1.0, jitter=0.05)
    pa.x[3] = -0.004           # needs box-wrapping
    pa.u[:] = np.sin(2 * np.pi * pa.y)
    kernel = K.QuinticSpline(dim=3)
    eqs = TVFScheme(['fluid'], [], dim=3, rho0=1.0, c0=10.0, nu=0.01, p0=100.0,
                    pb=100.0, h0=dx).get_equations()
    kw = dict(xmin=0, xmax=1, ymin=0, ymax=1, zmin=0, zmax=1, periodic_in_x=True,
              periodic_in_y=True, periodic_in_z=True)
    ref = _copy_arrays([pa])
    dref = DomainManager(**kw)
    dref.set_particles(ref, kernel.radius_scale)
    dref.update()
    onn = oracle.OracleNNPS(3, ref, kernel.radius_scale)
    onn.update()
    oev = oracle.OracleEval(ref, eqs, kernel, nthreads=4)
    oev.set_nnps(onn)
    oev.compute(0.0, 1e-4)

    ctx = dev.HipContext(0)
    nreal = pa.get_number_of_particles()
    dev.attach(pa, ctx).push()
    a_eval = AccelerationEval([pa], eqs, kernel)
    SPHCompiler(a_eval, ctx=ctx, sync='manual').compile()
    nnps = HipNNPS(3, [pa], radius_scale=kernel.radius_scale, ctx=ctx,
                   domain=HipDomainManager(ctx=ctx, **kw), sync=False)
    a_eval.set_nnps(nnps)
    assert pa.gpu.get_number_of_particles() == ref[0].get_number_of_particles()
    assert pa.gpu.get_number_of_particles(True) == nreal
    a_eval.compute(0.0, 1e-4)
    nnps.update_domain()       # again: ghosts are dropped and rebuilt
    nnps.update()
    a_eval.compute(0.0, 1e-4)
    # (updates after the first make the images into fixed capacities: count the rows that are particles)
    assert live_rows(pa).size == ref[0].get_number_of_particles()
    pa.gpu.pull('x', 'rho', 'V', 'p', 'au', 'av', 'aw', 'auhat', 'avhat', 'awhat')
    assert abs(pa.x[3] - 0.996) < 1e-15
    for prop in ('rho', 'V', 'p', 'au', 'av', 'aw', 'auhat', 'avhat', 'awhat'):
        e = rel_err(pa.properties[prop][:nreal], ref[0].properties[prop][:nreal])
        assert e < 1e-10, (prop, e)


def test_host_mirror_ghosts_lattice_density(oracle):
    """Reflecting planes (nnps_base.pyx:506-697): a lattice whose first layer
    sits half a spacing from the plane continues seamlessly across it, so the
    summation density of a closed mirrored box equals that of the periodic
    one for every particle; images carry the flipped normal velocity; corner
    images (x and y reflected) exist."""
    from pysph_amd import kernels as K
    from pysph_amd.domain import DomainManager
    from pysph_amd.equations import Group, TVFSummationDensity
    pa, dx = lattice(12, dim=2, hdx=1.2)
    pa.u[:] = 1.0 + pa.y
    pa.v[:] = -0.5
    kernel = K.QuinticSpline(dim=2)
    dom = DomainManager(xmin=0, xmax=1, ymin=0, ymax=1, mirror_in_x=True,
                        mirror_in_y=True)
    dom.set_particles([pa], kernel.radius_scale)
    dom.update()
    n, nreal = pa.get_number_of_particles(), pa.get_number_of_particles(True)
    assert nreal == 144 and n > nreal and np.all(pa.tag[nreal:] == 2)
    dom.update()
    assert pa.get_number_of_particles() == n                  # idempotent
    gx, gy, gu, gv = pa.x[nreal:], pa.y[nreal:], pa.u[nreal:], pa.v[nreal:]
    assert gx.min() < 0 and gx.max() > 1 and gy.min() < 0 and gy.max() > 1
    corner = (gx < 0) & (gy < 0)
    assert corner.any()
    # every image is the reflection of a real particle: fold back and match
    fx = np.where(gx < 0, -gx, np.where(gx > 1, 2 - gx, gx))
    fy = np.where(gy < 0, -gy, np.where(gy > 1, 2 - gy, gy))
    real = set(zip(np.round(pa.x[:nreal] / dx - 0.5).astype(int),
                   np.round(pa.y[:nreal] / dx - 0.5).astype(int)))
    assert all((int(round(a / dx - 0.5)), int(round(b / dx - 0.5))) in real
               for a, b in zip(fx, fy))
    xflip = (gx < 0) | (gx > 1)
    yflip = (gy < 0) | (gy > 1)
    assert np.allclose(np.where(xflip, -gu, gu), 1.0 + fy, atol=1e-14)
    assert np.allclose(np.where(yflip, -gv, gv), -0.5, atol=1e-14)
    eqs = [Group(equations=[TVFSummationDensity(dest='fluid', sources=['fluid'])])]
    nn = oracle.OracleNNPS(2, [pa], kernel.radius_scale)
    nn.update()
    ev = oracle.OracleEval([pa], eqs, kernel)
    ev.set_nnps(nn)
    ev.compute(0.0, 0.1)
    V = pa.V[:nreal]
    assert np.max(np.abs(V - V[0])) < 1e-11 * V[0]
    assert np.max(np.abs(1.0 / V - dx ** 2)) < 1e-4 * dx ** 2


@pytest.mark.gpu
def test_device_mirror_matches_host(oracle):
    """HipDomainManager with reflecting planes (sph_halo_pack_mirror) builds
    the same image set as the host DomainManager (periodic in z on top) and
    the WCSPH evaluation of the real particles agrees with the oracle."""
    from pysph_amd import kernels as K
    from pysph_amd import device as dev
    from pysph_amd.domain import DomainManager, HipDomainManager
    from pysph_amd.acceleration_eval import AccelerationEval, SPHCompiler
    from pysph_amd.nnps import HipNNPS
    from test_hip_parity import _copy_arrays, make_cube, cube_equations
    pa, dx = make_cube(12)
    kernel = K.WendlandQuintic(dim=3)
    eqs = cube_equations(dx)
    kw = dict(xmin=0, xmax=1, ymin=0, ymax=1, zmin=0, zmax=1, mirror_in_x=True,
              mirror_in_y=True, periodic_in_z=True, n_layers=1.0)
    ref = _copy_arrays([pa])
    dref = DomainManager(**kw)
    dref.set_particles(ref, kernel.radius_scale)
    dref.update()
    onn = oracle.OracleNNPS(3, ref, kernel.radius_scale)
    onn.update()
    oev = oracle.OracleEval(ref, eqs, kernel, nthreads=4)
    oev.set_nnps(onn)
    oev.compute(0.0, 1e-5)
    ctx = dev.HipContext(0)
    nreal = pa.get_number_of_particles()
    dev.attach(pa, ctx).push()
    a_eval = AccelerationEval([pa], eqs, kernel)
    SPHCompiler(a_eval, ctx=ctx, sync='manual').compile()
    nnps = HipNNPS(3, [pa], radius_scale=kernel.radius_scale, ctx=ctx,
                   domain=HipDomainManager(ctx=ctx, **kw), sync=False)
    a_eval.set_nnps(nnps)
    n_dev = pa.gpu.get_number_of_particles()
    assert n_dev == ref[0].get_number_of_particles() and n_dev > nreal
    # same image SET (positions + flipped velocities), order aside
    got = np.empty((5, n_dev))
    for k, p in enumerate(('x', 'y', 'z', 'u', 'v')):
        pa.gpu.pull_into(p, got[k])
    want = np.array([ref[0].properties[p] for p in ('x', 'y', 'z', 'u', 'v')])
    key = lambda a: a[:, np.lexsort(a[::-1])]
    assert np.array_equal(key(got[:, nreal:]), key(want[:, nreal:]))
    a_eval.compute(0.0, 1e-5)
    outs = ('arho', 'au', 'av', 'aw', 'ax', 'ay', 'az')
    pa.gpu.pull(*outs)
    for prop in outs:
        e = rel_err(pa.properties[prop][:nreal], ref[0].properties[prop][:nreal])
        assert e < 1e-10, (prop, e)


# ---------------------------------------------------------------------------
# pysph/base/tests/test_periodic_nnps.py: fluid between two plates, periodic
# along the channel (two particle arrays share the periodic images)
# ---------------------------------------------------------------------------
def periodic_channel_2d(n=100):
    """test_periodic_nnps.py:30-77"""
    from pysph_amd.particle_array import get_particle_array
    L, hdx = 1.0, 1.5
    dx = L / n
    _x = np.arange(dx / 2, L, dx)
    xx, yy = np.meshgrid(_x, _x)
    x, y = xx.ravel(), yy.ravel()
    fluid = get_particle_array(name='fluid', x=x, y=y, h=np.ones_like(x) * hdx * dx,
                               m=np.ones_like(x) * dx * dx, V=np.zeros_like(x))
    xt, yt = [a.ravel() for a in np.meshgrid(_x, np.arange(L + dx / 2, L + dx / 2 + 10 * dx, dx))]
    xb, yb = [a.ravel() for a in np.meshgrid(_x, np.arange(-dx / 2, -dx / 2 - 10 * dx, -dx))]
    x, y = np.concatenate((xt, xb)), np.concatenate((yt, yb))
    channel = get_particle_array(name='channel', x=x, y=y, h=np.ones_like(x) * hdx * dx,
                                 m=np.ones_like(x) * dx * dx, V=np.zeros_like(x))
    return [fluid, channel], dx * dx


def channel_density_equations():
    from pysph_amd.equations import Group, TVFSummationDensity
    return [Group(equations=[TVFSummationDensity(dest='fluid', sources=['fluid', 'channel'])])]


def check_channel(fluid, vol, nreal):
    """test_periodic_nnps.py:143-147: number density and density by summation,
    to six places, for EVERY fluid particle (also those next to the periodic
    faces and next to the plates)"""
    V, rho, m = fluid.V[:nreal], fluid.rho[:nreal], fluid.m[:nreal]
    assert np.max(np.abs(1.0 / V - vol)) < 0.5e-6
    assert np.max(np.abs(rho - m / (1.0 / V))) < 0.5e-6


def test_host_periodic_channel_two_arrays(oracle):
    from pysph_amd import kernels as K
    from pysph_amd.domain import DomainManager
    arrays, vol = periodic_channel_2d(40)
    kernel = K.Gaussian(dim=2)
    dom = DomainManager(xmin=0, xmax=1.0, periodic_in_x=True)
    assert dom.is_periodic and dom.periodic_in_x and not dom.periodic_in_y and not dom.periodic_in_z
    dom.set_particles(arrays, kernel.radius_scale)
    dom.update()
    nreal = arrays[0].get_number_of_particles(True)
    assert all(a.get_number_of_particles() > a.get_number_of_particles(True) for a in arrays)
    nn = oracle.OracleNNPS(2, arrays, kernel.radius_scale)
    nn.update()
    ev = oracle.OracleEval(arrays, channel_density_equations(), kernel, nthreads=8)
    ev.set_nnps(nn)
    ev.compute(0.0, 0.1)
    check_channel(arrays[0], vol, nreal)


@pytest.mark.gpu
def test_device_periodic_channel_two_arrays():
    """the reference's full-size case (n = 100: 10 000 fluid + 2 000 plate
    particles) with device-resident periodic images of BOTH arrays"""
    from pysph_amd import device as dev
    from pysph_amd import kernels as K
    from pysph_amd.acceleration_eval import AccelerationEval, SPHCompiler
    from pysph_amd.domain import HipDomainManager
    from pysph_amd.nnps import HipNNPS
    arrays, vol = periodic_channel_2d(100)
    kernel = K.Gaussian(dim=2)
    ctx = dev.HipContext(0)
    for a in arrays:
        dev.attach(a, ctx).push()
    a_eval = AccelerationEval(arrays, channel_density_equations(), kernel)
    SPHCompiler(a_eval, ctx=ctx, sync='manual').compile()
    dom = HipDomainManager(ctx=ctx, xmin=0, xmax=1.0, periodic_in_x=True)
    nnps = HipNNPS(2, arrays, radius_scale=kernel.radius_scale, ctx=ctx, domain=dom, sync=False)
    assert dom.is_periodic and dom.periodic_in_x and not dom.periodic_in_y
    a_eval.set_nnps(nnps)
    a_eval.compute(0.0, 0.1)
    fluid = arrays[0]
    nreal = fluid.gpu.get_number_of_particles(True)
    assert nreal == 10000 and fluid.gpu.get_number_of_particles() > nreal
    fluid.gpu.pull('V', 'rho')
    check_channel(fluid, vol, nreal)


@pytest.mark.gpu
@pytest.mark.parametrize('axes', ['xyz', 'z', 'xy'])
def test_device_periodic_3d_flag_combinations(axes):
    """test_periodic_nnps.py:227-310: the flag combinations; for each, a
    lattice filling the periodic directions has the lattice number density on
    every particle away from the open faces"""
    from pysph_amd import device as dev
    from pysph_amd import kernels as K
    from pysph_amd.acceleration_eval import AccelerationEval, SPHCompiler
    from pysph_amd.domain import HipDomainManager
    from pysph_amd.equations import Group, TVFSummationDensity
    from pysph_amd.nnps import HipNNPS
    pa, dx = lattice(12, dim=3, hdx=1.5)
    for p in ('x', 'y', 'z'):
        pa.properties[p] -= 0.5                     # box [-1/2, 1/2]^3 as in the reference
    kernel = K.Gaussian(dim=3)
    kw = {}
    for ax in axes:
        kw.update({ax + 'min': -0.5, ax + 'max': 0.5, 'periodic_in_' + ax: True})
    ctx = dev.HipContext(0)
    dev.attach(pa, ctx).push()
    a_eval = AccelerationEval([pa], [Group(equations=[TVFSummationDensity('fluid', ['fluid'])])],
                              kernel)
    SPHCompiler(a_eval, ctx=ctx, sync='manual').compile()
    dom = HipDomainManager(ctx=ctx, **kw)
    nnps = HipNNPS(3, [pa], radius_scale=kernel.radius_scale, ctx=ctx, domain=dom, sync=False)
    assert (dom.periodic_in_x, dom.periodic_in_y, dom.periodic_in_z) == \
        ('x' in axes, 'y' in axes, 'z' in axes)
    a_eval.set_nnps(nnps)
    a_eval.compute(0.0, 0.1)
    nreal = pa.gpu.get_number_of_particles(True)
    pa.gpu.pull('V')
    inner = np.ones(nreal, dtype=bool)
    for ax in 'xyz':
        if ax not in axes:                          # open direction: stay 3h from the faces
            inner &= np.abs(pa.properties[ax][:nreal]) < 0.5 - 3.2 * 1.5 * dx
    assert inner.any()
    assert np.max(np.abs(1.0 / pa.V[:nreal][inner] - dx ** 3)) < 0.5e-6


@pytest.mark.gpu
@pytest.mark.parametrize('axes', ['xyz', 'xy'])
def test_device_periodic_update_without_round_trip_equals_counted_update(axes):
    """Round 5 (opt-in, protocol='padded'): from its second update on the device domain manager makes the periodic
    images without a device->host round trip -- h known without looking, both faces' images of an axis appended into fixed capacities sized from
    the previous update's counts, the rows behind the counts parked (sph_domain_images_padded).  Same live ghosts in
    the same order as the counted (list-based) update, so the densities of a moving jittered lattice are
    BIT-IDENTICAL between the two protocols update after update, particles crossing the faces included; the
    neighbour updates in between make no round trip either; capacities that are too small are an error, one update
    late."""
    from pysph_amd import device as dev
    from pysph_amd import kernels as K
    from pysph_amd.acceleration_eval import AccelerationEval, SPHCompiler
    from pysph_amd.domain import HipDomainManager
    from pysph_amd.equations import Group, TVFSummationDensity
    from pysph_amd.nnps import HipNNPS
    kernel = K.QuinticSpline(dim=3)
    kw = {}
    for ax in axes:
        kw.update({ax + 'min': 0.0, ax + 'max': 1.0, 'periodic_in_' + ax: True})
    runs = {}
    for protocol in ('padded', 'counted'):
        pa, dx = lattice(16, dim=3, hdx=1.2)
        rng = np.random.default_rng(5)
        for q in 'xyz':
            pa.properties[q] += 0.2 * dx * rng.uniform(-1, 1, pa.x.size)
        ctx = dev.HipContext(0)
        ctx.timer_enable(True)
        dev.attach(pa, ctx).push()
        a_eval = AccelerationEval([pa], [Group(equations=[TVFSummationDensity('fluid', ['fluid'])])], kernel)
        SPHCompiler(a_eval, ctx=ctx, sync='manual').compile()
        # (the lattice drifts by half a spacing per update: whole planes cross the image thresholds at once -- headroom
        # for that; the default 12 % trips the overflow error here, as it should)
        dom = HipDomainManager(ctx=ctx, protocol=protocol, headroom=0.5, **kw)
        nnps = HipNNPS(3, [pa], radius_scale=kernel.radius_scale, ctx=ctx, domain=dom, sync=False)
        a_eval.set_nnps(nnps)
        nreal = pa.gpu.get_number_of_particles(True)
        out, live = [], []
        for step in range(5):
            if step:
                from helpers import device_add
                for q, amp in zip('xyz', (0.45, -0.3, 0.2)):        # a drift (on the device, as a stage kernel moves
                    device_add(pa, q, np.full(nreal, amp * dx))     # particles): they leave through the faces
                nnps.update_domain()        # ghosts dropped, particles wrapped, images made again
                nnps.update()
            a_eval.compute(0.0, 0.1)
            pa.gpu.pull('V')
            out.append(pa.V[:nreal].copy())
            # the live rows (real particles, then the images), in order
            live.append(np.stack([live_rows(pa, q) for q in 'xyz']))
        runs[protocol] = (out, live, dom.padded_updates, ctx.timer_get('n_async')[1], dom)
    for a, b in zip(runs['padded'][0], runs['counted'][0]):
        assert np.array_equal(a, b)
    for a, b in zip(runs['padded'][1], runs['counted'][1]):
        assert a.shape == b.shape and np.array_equal(a, b)
    assert not np.array_equal(runs['padded'][1][0], runs['padded'][1][2])      # (the ghost set really changed)
    assert runs['padded'][2] == 4 and runs['counted'][2] == 0
    assert runs['padded'][3] >= 3           # neighbour updates without a round trip
    # capacities that are too small: the images do not fit, the update after says so
    dom = runs['padded'][4]
    dom._collect_counts()               # (the counts of the last update: the capacities would follow them)
    for key in dom._caps:
        dom._caps[key] = [8, 8]
    dom.update()
    with pytest.raises(RuntimeError, match='did not fit'):
        dom.update()


@pytest.mark.gpu
def test_default_periodic_update_repairs_images_that_did_not_fit():
    """Round 6: the round-trip-free periodic update is the DEFAULT.  Its capacities follow the previous update's counts
    with some headroom -- which a lattice plane drifting across an image threshold overruns at scale (a whole layer
    enters at once; here: capacities of count + 8 rows make a 16^2 plane do it).  verify(), called with the
    evaluation queued, makes the missing images by a counted update and says so; the evaluation then runs again: the
    densities equal the counted protocol's update after update.  Images restricted to what the evaluation
    reads (set_image_props) give the same densities."""
    from helpers import device_add
    from pysph_amd import device as dev
    from pysph_amd import kernels as K
    from pysph_amd.acceleration_eval import AccelerationEval, SPHCompiler
    from pysph_amd.domain import HipDomainManager
    from pysph_amd.equations import Group, TVFSummationDensity
    from pysph_amd.nnps import HipNNPS
    kernel = K.QuinticSpline(dim=3)
    kw = {}
    for ax in 'xyz':
        kw.update({ax + 'min': 0.0, ax + 'max': 1.0, 'periodic_in_' + ax: True})
    runs = {}
    for mode in ('default', 'restricted', 'counted'):
        pa, dx = lattice(16, dim=3, hdx=1.2)
        ctx = dev.HipContext(0)
        dev.attach(pa, ctx).push()
        a_eval = AccelerationEval([pa], [Group(equations=[TVFSummationDensity('fluid', ['fluid'])])], kernel)
        SPHCompiler(a_eval, ctx=ctx, sync='manual').compile()
        dom = HipDomainManager(ctx=ctx, **(dict(kw, protocol='counted') if mode == 'counted' else kw))
        assert dom.protocol == ('counted' if mode == 'counted' else 'padded')
        if mode == 'restricted':
            dom.set_image_props(a_eval.c_acceleration_eval.inputs)
        dom._capacity = lambda count: int(count) + 8
        nnps = HipNNPS(3, [pa], radius_scale=kernel.radius_scale, ctx=ctx, domain=dom, sync=False)
        a_eval.set_nnps(nnps)
        nreal = pa.gpu.get_number_of_particles(True)
        out = []
        for step in range(6):
            if step:
                for q, amp in zip('xyz', (0.45, -0.3, 0.2)):        # the exact lattice drifts: whole planes cross the thresholds
                    device_add(pa, q, np.full(nreal, amp * dx))
                nnps.update_domain()
                nnps.update()
            a_eval.compute(0.0, 0.1)
            while not dom.verify():
                nnps.update()
                a_eval.compute(0.0, 0.1)
            pa.gpu.pull('V')
            out.append(pa.V[:nreal].copy())
        runs[mode] = (out, dom.padded_updates, dom.repaired_updates, len(dom._props_of(dom.helpers[0])))
    for mode in ('default', 'restricted'):
        for a, b in zip(runs[mode][0], runs['counted'][0]):
            # (same image SETS; a repaired step sorts its particles on another binning grid -- the bounds of its own first
            # update instead of the previous step's -- and sums in another order: equal to rounding)
            assert np.max(np.abs(a - b)) <= 1e-13 * np.max(np.abs(b)), mode
        assert runs[mode][1] >= 3 and runs[mode][2] >= 1, runs[mode][1:]       # padded updates; at least one was repaired
    assert runs['restricted'][3] < runs['default'][3]


# ---------------------------------------------------------------------------
# pysph/base/tests/test_domain_manager.py: periodic box, every property of a
# ghost is the property of its original; box wrapping of particles that left
# ---------------------------------------------------------------------------
def periodic_box(dim, n=10):
    """test_domain_manager.py:30-57, :249-293; p = mod(x) + mod(y) + mod(z)"""
    from pysph_amd.particle_array import get_particle_array
    L, hdx = 1.0, 1.5
    dx = L / n
    _x = np.arange(dx / 2, L, dx)
    if dim == 2:
        x, y = [a.ravel() for a in np.meshgrid(_x, _x)]
        z = np.zeros_like(x)
    else:
        x, y, z = [a.ravel() for a in np.meshgrid(_x, _x, _x)]
    p = np.mod(x, L) + np.mod(y, L) + np.mod(z, L)
    pa = get_particle_array(name='fluid', x=x, y=y, z=z, h=np.ones_like(x) * hdx * dx,
                            m=np.ones_like(x) * dx ** dim, V=np.zeros_like(x), p=p)
    kw = dict(xmin=0, xmax=L, ymin=0, ymax=L, periodic_in_x=True, periodic_in_y=True)
    if dim == 3:
        kw.update(zmin=0, zmax=L, periodic_in_z=True)
    return pa, dx ** dim, kw


@pytest.mark.parametrize('dim', [2, 3])
def test_host_ghosts_carry_every_property_and_box_wrapping(oracle, dim):
    from pysph_amd import kernels as K
    from pysph_amd.domain import DomainManager
    from pysph_amd.equations import Group, TVFSummationDensity
    pa, vol, kw = periodic_box(dim, 10 if dim == 2 else 8)
    kernel = K.Gaussian(dim=dim)
    orig_n = pa.get_number_of_particles()
    dom = DomainManager(**kw)
    dom.set_particles([pa], kernel.radius_scale)
    dom.update()
    # test_periodicity (:216-237): images beyond every periodic face, with the
    # pressure of the particle they are an image of
    assert pa.get_number_of_particles() > orig_n
    for ax in 'xyz'[:dim]:
        v = pa.properties[ax]
        assert v.min() < 0.0 and v.max() > 1.0
    expect = np.mod(pa.x, 1.0) + np.mod(pa.y, 1.0) + np.mod(pa.z, 1.0)
    assert np.allclose(pa.p, expect, atol=1e-14)
    # test_box_wrapping (:209-214): move everything by 0.35, update, density unchanged
    n = pa.get_number_of_particles(True)
    pa.x[:n] += 0.35
    pa.y[:n] += 0.35
    dom.update()
    n = pa.get_number_of_particles(True)
    assert n == orig_n and pa.x[:n].max() <= 1.0 and pa.x[:n].min() >= 0.0
    nn = oracle.OracleNNPS(dim, [pa], kernel.radius_scale)
    nn.update()
    ev = oracle.OracleEval([pa], [Group(equations=[TVFSummationDensity('fluid', ['fluid'])])],
                           kernel, nthreads=4)
    ev.set_nnps(nn)
    ev.compute(0.0, 0.1)
    V, rho = pa.V[:n], pa.rho[:n]
    assert np.max(np.abs(1.0 / V - vol)) < 0.5e-5                 # :115 (5 places)
    assert np.max(np.abs(rho - pa.m[:n] * V)) < 0.5e-14           # :116 (14 places)


@pytest.mark.gpu
@pytest.mark.parametrize('dim', [2, 3])
def test_device_ghosts_carry_every_property_and_box_wrapping(dim):
    from pysph_amd import device as dev
    from pysph_amd import kernels as K
    from pysph_amd.acceleration_eval import AccelerationEval, SPHCompiler
    from pysph_amd.domain import HipDomainManager
    from pysph_amd.equations import Group, TVFSummationDensity
    from pysph_amd.nnps import HipNNPS
    pa, vol, kw = periodic_box(dim, 10 if dim == 2 else 8)
    kernel = K.Gaussian(dim=dim)
    orig_n = pa.get_number_of_particles()
    ctx = dev.HipContext(0)
    dev.attach(pa, ctx).push()
    a_eval = AccelerationEval([pa], [Group(equations=[TVFSummationDensity('fluid', ['fluid'])])],
                              kernel)
    SPHCompiler(a_eval, ctx=ctx, sync='manual').compile()
    dom = HipDomainManager(ctx=ctx, **kw)
    nnps = HipNNPS(dim, [pa], radius_scale=kernel.radius_scale, ctx=ctx, domain=dom, sync=False)
    a_eval.set_nnps(nnps)
    n_all = pa.gpu.get_number_of_particles()
    assert n_all > orig_n and pa.gpu.get_number_of_particles(True) == orig_n
    x, y, z, p = [getattr(pa.gpu, k).get() for k in ('x', 'y', 'z', 'p')]
    for v in (x, y, z)[:dim]:
        assert v.min() < 0.0 and v.max() > 1.0
    assert np.allclose(p, np.mod(x, 1.0) + np.mod(y, 1.0) + np.mod(z, 1.0), atol=1e-14)
    # everything moves by 0.35: wrapped back into the box, density as before
    shifted = pa.gpu.x.get()
    shifted[:orig_n] += 0.35
    _push_all(pa, 'x', shifted)
    shifted = pa.gpu.y.get()
    shifted[:orig_n] += 0.35
    _push_all(pa, 'y', shifted)
    nnps.update_domain()
    nnps.update()
    a_eval.compute(0.0, 0.1)
    x = pa.gpu.x.get()[:orig_n]
    assert x.max() <= 1.0 and x.min() >= 0.0
    V = pa.gpu.V.get()[:orig_n]
    rho = pa.gpu.rho.get()[:orig_n]
    assert np.max(np.abs(1.0 / V - vol)) < 0.5e-5
    assert np.max(np.abs(rho - pa.m[:orig_n] * V)) < 0.5e-14


def _push_all(pa, prop, values):
    """overwrite a device property including its ghost tail"""
    import ctypes as C
    from pysph_amd import device as dev
    values = np.ascontiguousarray(values, dtype=np.float64)
    dev._check(pa.gpu.lib.sph_array_push(pa.gpu.ctx._h, pa.gpu.array_id, dev.prop_id(prop),
                                         values.ctypes.data_as(C.POINTER(C.c_double)), 0,
                                         values.size))
