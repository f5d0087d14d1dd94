#!/usr/bin/env python
"""Generate the golden vectors in tests/golden/*.npz from the REFERENCE.

Run in the build container only (needs /root/reference; ~2 min):

    python tests/golden/make_golden.py

Each case builds its equation list with the reference's own classes
(``pysph.sph.scheme.WCSPHScheme/TVFScheme``, ``pysph.sph.wc.*``,
``pysph.base.kernels``) and executes them as plain Python through
``oracle/ref_driver.py`` (the reference's equation bodies, kernels, pair-symbol
code blocks and MegaGroup regrouping are the reference's code; see that file
for what is restated).  The neighbour search used is checked against brute
force for every destination particle before anything is written.

Stored per case: the input state of every particle array, the output state
after one ``compute(t, dt)``, the NNPS scalars and (case 1) the neighbour
lists in the reference's traversal order.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

from oracle.ref_driver import (setup_reference_imports, ListPA,  # noqa: E402
                               PyLinkedListNNPS, RefEval)

setup_reference_imports()

from pysph.sph.acceleration_eval import AccelerationEval  # noqa: E402
from pysph.sph.scheme import WCSPHScheme, TVFScheme  # noqa: E402
from pysph.sph.equation import Group  # noqa: E402
from pysph.sph.basic_equations import SummationDensity  # noqa: E402
from pysph.base.kernels import (CubicSpline, WendlandQuintic,  # noqa: E402
                                QuinticSpline)

from pysph_amd.examples import dam_break_3d as db  # noqa: E402


def check_nnps(nnps):
    n = len(nnps.particles)
    for s in range(n):
        for d in range(n):
            for i in range(nnps.particles[d].n):
                a = sorted(nnps.neighbors(s, d, i))
                b = nnps.brute_force(s, d, i)
                assert a == b, (s, d, i)


def run_case(fname, arrays, equations, kernel, dim, t=0.0, dt=1e-4,
             store_nbrs=(), meta=None):
    """arrays: list of (name, dict prop->np.ndarray, n_real)."""
    pas = [ListPA(name, props, n_real) for name, props, n_real in arrays]
    out = {}
    for pa in pas:
        for k, v in pa.properties.items():
            out['in/%s/%s' % (pa.name, k)] = np.array(v)
        out['nreal/%s' % pa.name] = np.array(pa.n_real)
    a_eval = AccelerationEval(pas, equations, kernel)
    nnps = PyLinkedListNNPS(dim, pas, radius_scale=kernel.radius_scale)
    nnps.update()
    check_nnps(nnps)
    RefEval(a_eval, nnps).compute(t, dt)
    for pa in pas:
        for k, v in pa.properties.items():
            out['out/%s/%s' % (pa.name, k)] = np.array(v)
    out['nnps/cell_size'] = np.array(nnps.cell_size)
    out['nnps/xmin'] = np.array(nnps.xmin)
    out['nnps/xmax'] = np.array(nnps.xmax)
    out['nnps/ncells_per_dim'] = np.array(nnps.nc)
    out['nnps/n_cells'] = np.array(nnps.n_cells)
    names = [pa.name for pa in pas]
    for s, d in store_nbrs:
        si, di = names.index(s), names.index(d)
        start, nbrs = [0], []
        for i in range(pas[di].n):
            nb = nnps.neighbors(si, di, i)
            nbrs.extend(nb)
            start.append(len(nbrs))
        out['nbrs/%s/%s/start' % (s, d)] = np.array(start, dtype=np.uint32)
        out['nbrs/%s/%s/idx' % (s, d)] = np.array(nbrs, dtype=np.uint32)
    out['t'] = np.array(t)
    out['dt'] = np.array(dt)
    for k, v in (meta or {}).items():
        out['meta/' + k] = np.array(v)
    path = os.path.join(HERE, fname)
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB')


WC_IN = ['x', 'y', 'z', 'u', 'v', 'w', 'h', 'm', 'rho']
WC_OUT = ['p', 'cs', 'arho', 'au', 'av', 'aw', 'ax', 'ay', 'az', 'dt_cfl',
          'dt_force']


def case_wcsph_dam():
    """Mini dam break (dx=0.1), the dam_break_3d.py equation set from the
    reference's WCSPHScheme, perturbed state so every branch is taken."""
    rng = np.random.default_rng(20250925)
    dx = 0.1
    arrays = []
    for pa in db.create_particles(dx):
        n = pa.get_number_of_particles()
        props = {k: pa.properties[k].copy() for k in WC_IN}
        props['rho'] = db.ro * (1 + 0.02 * rng.uniform(-1, 1, n))
        if pa.name == 'fluid':
            for k in 'xyz':
                props[k] = props[k] + 0.1 * dx * rng.uniform(-1, 1, n)
            for k in 'uvw':
                props[k] = 0.1 * db.c0 * rng.uniform(-1, 1, n)
        for k in WC_OUT:
            props[k] = rng.uniform(-1, 1, n)  # garbage: must be overwritten
        arrays.append((pa.name, props, n))
    s = WCSPHScheme(['fluid'], ['boundary', 'obstacle'], dim=3, rho0=db.ro,
                    c0=db.c0, h0=dx * db.hdx, hdx=db.hdx, gz=-9.81,
                    alpha=db.alpha, beta=db.beta, gamma=db.gamma,
                    hg_correction=True, tensile_correction=False)
    run_case('wcsph_dam_dx0.1.npz', arrays, s.get_equations(),
             WendlandQuintic(dim=3), 3,
             store_nbrs=[('fluid', 'fluid'), ('boundary', 'fluid'),
                         ('fluid', 'boundary'), ('fluid', 'obstacle')],
             meta=dict(dx=dx))


def case_wcsph_dam_varh():
    """The mini dam break again with EVERY particle its own smoothing length (h +- 15 % in all three arrays): the
    scenario of the one-launch variable-h family (round 5) -- several arrays AND the `or` of the neighbour criterion AND
    HIJ-dependent kernels in one case.  No neighbour lists stored (wcsph_dam_dx0.1 and wcsph_cube_varh pin those)."""
    rng = np.random.default_rng(20250926)
    dx = 0.1
    arrays = []
    for pa in db.create_particles(dx):
        n = pa.get_number_of_particles()
        props = {k: pa.properties[k].copy() for k in WC_IN}
        props['rho'] = db.ro * (1 + 0.02 * rng.uniform(-1, 1, n))
        props['h'] = props['h'] * (1 + 0.15 * rng.uniform(-1, 1, n))
        if pa.name == 'fluid':
            for k in 'xyz':
                props[k] = props[k] + 0.1 * dx * rng.uniform(-1, 1, n)
            for k in 'uvw':
                props[k] = 0.1 * db.c0 * rng.uniform(-1, 1, n)
        for k in WC_OUT:
            props[k] = rng.uniform(-1, 1, n)  # garbage: must be overwritten
        arrays.append((pa.name, props, n))
    s = WCSPHScheme(['fluid'], ['boundary', 'obstacle'], dim=3, rho0=db.ro,
                    c0=db.c0, h0=dx * db.hdx, hdx=db.hdx, gz=-9.81,
                    alpha=db.alpha, beta=db.beta, gamma=db.gamma,
                    hg_correction=True, tensile_correction=False)
    run_case('wcsph_dam_varh.npz', arrays, s.get_equations(),
             WendlandQuintic(dim=3), 3, meta=dict(dx=dx))


def case_wcsph_cube_varh():
    """8^3 jittered cube, variable h (exercises the `or` of the neighbour
    criterion), CubicSpline, tensile correction + beta != 0, summation density,
    ghost particles (n_real < n) so Group(real=...) matters."""
    rng = np.random.default_rng(7)
    n1 = 8
    dx = 1.0 / n1
    g = (np.arange(n1) + 0.5) * dx
    x, y, z = [a.ravel() for a in np.meshgrid(g, g, g, indexing='ij')]
    n = x.size
    rho0, c0 = 1000.0, 10.0
    props = dict(
        x=x + 0.2 * dx * rng.uniform(-1, 1, n),
        y=y + 0.2 * dx * rng.uniform(-1, 1, n),
        z=z + 0.2 * dx * rng.uniform(-1, 1, n),
        u=rng.uniform(-1, 1, n), v=rng.uniform(-1, 1, n),
        w=rng.uniform(-1, 1, n),
        h=1.2 * dx * (1 + 0.25 * rng.uniform(-1, 1, n)),
        m=rho0 * dx ** 3 * np.ones(n),
        rho=rho0 * (1 + 0.05 * rng.uniform(-1, 1, n)))
    for k in WC_OUT:
        props[k] = rng.uniform(-1, 1, n)
    s = WCSPHScheme(['fluid'], [], dim=3, rho0=rho0, c0=c0, h0=1.2 * dx,
                    hdx=1.2, gx=0.5, gy=-0.25, gz=-9.81, alpha=1.0, beta=1.0,
                    gamma=7.0, tensile_correction=True,
                    summation_density=True)
    run_case('wcsph_cube_varh.npz', [('fluid', props, n - 40)],
             s.get_equations(), CubicSpline(dim=3), 3,
             store_nbrs=[('fluid', 'fluid')], meta=dict(dx=dx))


def case_sd_1d():
    """test_acceleration_eval.py:295-316 fixture: 10 particles on a line,
    h = 1.05 dx, CubicSpline(dim=1), SummationDensity."""
    n = 10
    dx = 1.0 / (n - 1)
    x = np.linspace(0, 1, n)
    z = np.zeros(n)
    props = dict(x=x, y=z.copy(), z=z.copy(), h=np.ones(n) * dx * 1.05,
                 m=np.ones(n), rho=z.copy())
    eqs = [Group(equations=[SummationDensity(dest='fluid', sources=['fluid'])])]
    run_case('sd_1d_line.npz', [('fluid', props, n)], eqs, CubicSpline(dim=1),
             1, store_nbrs=[('fluid', 'fluid')])


def case_tvf_cube():
    """7^3 jittered cube, the reference TVFScheme (fluids only) with nu>0 and
    alpha>0 so every TVF momentum term is present; QuinticSpline."""
    rng = np.random.default_rng(11)
    n1 = 7
    dx = 1.0 / n1
    g = (np.arange(n1) + 0.5) * dx
    x, y, z = [a.ravel() for a in np.meshgrid(g, g, g, indexing='ij')]
    n = x.size
    rho0, c0 = 1.0, 10.0
    props = dict(
        x=x + 0.1 * dx * rng.uniform(-1, 1, n),
        y=y + 0.1 * dx * rng.uniform(-1, 1, n),
        z=z + 0.1 * dx * rng.uniform(-1, 1, n),
        u=rng.uniform(-1, 1, n), v=rng.uniform(-1, 1, n),
        w=rng.uniform(-1, 1, n),
        uhat=rng.uniform(-1, 1, n), vhat=rng.uniform(-1, 1, n),
        what=rng.uniform(-1, 1, n),
        h=dx * np.ones(n), m=rho0 * dx ** 3 * np.ones(n),
        rho=rho0 * (1 + 0.05 * rng.uniform(-1, 1, n)))
    for k in ['p', 'V', 'au', 'av', 'aw', 'auhat', 'avhat', 'awhat']:
        props[k] = rng.uniform(0.5, 1, n)
    s = TVFScheme(['fluid'], [], dim=3, rho0=rho0, c0=c0, nu=0.01,
                  p0=c0 * c0 * rho0, pb=c0 * c0 * rho0, h0=dx, gx=0.1,
                  alpha=0.2)
    run_case('tvf_cube.npz', [('fluid', props, n)], s.get_equations(),
             QuinticSpline(dim=3), 3, t=0.01, meta=dict(dx=dx))


def case_tvf_wall():
    """Fluid block between two wall slabs, the reference TVFScheme WITH solids:
    SetWallVelocity, SolidWallPressureBC and SolidWallNoSlipBC of
    transport_velocity.py act next to the momentum equations (the wall
    equations have no hand-written HIP kernel: this pins the generated-family
    path against the reference's own classes)."""
    rng = np.random.default_rng(23)
    nx, ny, nz, nw = 7, 6, 6, 3
    dx = 1.0 / nx
    gx_ = (np.arange(nx) + 0.5) * dx
    gy_ = (np.arange(ny) + 0.5) * dx
    gz_ = (np.arange(nz) + 0.5) * dx
    x, y, z = [a.ravel() for a in np.meshgrid(gx_, gy_, gz_, indexing='ij')]
    n = x.size
    rho0, c0 = 1.0, 10.0
    fluid = dict(
        x=x + 0.1 * dx * rng.uniform(-1, 1, n),
        y=y + 0.1 * dx * rng.uniform(-1, 1, n),
        z=z + 0.1 * dx * rng.uniform(-1, 1, n),
        u=rng.uniform(-1, 1, n), v=rng.uniform(-1, 1, n),
        w=rng.uniform(-1, 1, n),
        uhat=rng.uniform(-1, 1, n), vhat=rng.uniform(-1, 1, n),
        what=rng.uniform(-1, 1, n),
        h=dx * np.ones(n), m=rho0 * dx ** 3 * np.ones(n),
        rho=rho0 * (1 + 0.05 * rng.uniform(-1, 1, n)))
    for k in ['p', 'V', 'au', 'av', 'aw', 'auhat', 'avhat', 'awhat']:
        fluid[k] = rng.uniform(0.5, 1, n)
    # walls: nw layers below y=0 and above y=ny*dx (two layers further than
    # the kernel support stay untouched by the fluid: wij = 0 branch)
    gyw = np.concatenate([-(np.arange(nw) + 0.5) * dx,
                          ny * dx + (np.arange(nw) + 0.5) * dx])
    xw, yw, zw = [a.ravel() for a in np.meshgrid(gx_, gyw, gz_, indexing='ij')]
    m_ = xw.size
    wall = dict(
        x=xw.copy(), y=yw.copy(), z=zw.copy(),
        u=0.3 * np.ones(m_) * (yw > 0), v=np.zeros(m_), w=0.1 * np.ones(m_) * (yw < 0),
        h=dx * np.ones(m_), m=rho0 * dx ** 3 * np.ones(m_),
        rho=rho0 * np.ones(m_), p=rng.uniform(0.5, 1, m_),
        V=np.ones(m_) / dx ** 3,
        au=0.05 * rng.uniform(-1, 1, m_), av=0.05 * rng.uniform(-1, 1, m_),
        aw=0.05 * rng.uniform(-1, 1, m_))
    for k in ['wij', 'uf', 'vf', 'wf', 'ug', 'vg', 'wg']:
        wall[k] = rng.uniform(0.5, 1, m_)
    s = TVFScheme(['fluid'], ['wall'], dim=3, rho0=rho0, c0=c0, nu=0.01,
                  p0=c0 * c0 * rho0, pb=c0 * c0 * rho0, h0=dx, gy=-0.5,
                  alpha=0.2)
    run_case('tvf_wall.npz', [('fluid', fluid, n), ('wall', wall, m_)],
             s.get_equations(), QuinticSpline(dim=3), 3, t=0.01,
             meta=dict(dx=dx))


def case_kernels():
    """Kernel known answers from the reference's kernel classes (the same
    functions pysph/base/tests/test_kernel.py integrates)."""
    from pysph.base.kernels import Gaussian
    out = {}
    rng = np.random.default_rng(3)
    for cls, dims in ((CubicSpline, (1, 2, 3)), (WendlandQuintic, (2, 3)),
                      (QuinticSpline, (1, 2, 3)), (Gaussian, (1, 2, 3))):
        for dim in dims:
            k = cls(dim=dim)
            h = 0.5 + rng.uniform(0, 1, 64)
            q = np.concatenate([[0.0, 1e-13, 1.0, 2.0, 3.0],
                                rng.uniform(0, 3.5, 59)])
            r = q * h
            xij = rng.uniform(-1, 1, (64, 3))
            w = [k.kernel(list(xij[i]), r[i], h[i]) for i in range(64)]
            dw = [k.dwdq(r[i], h[i]) for i in range(64)]
            gr = []
            for i in range(64):
                g = [0.0, 0.0, 0.0]
                k.gradient(list(xij[i]), r[i], h[i], g)
                gr.append(g)
            key = '%s/%d/' % (cls.__name__, dim)
            out[key + 'h'] = h
            out[key + 'r'] = r
            out[key + 'xij'] = xij
            out[key + 'w'] = np.array(w)
            out[key + 'dwdq'] = np.array(dw)
            out[key + 'grad'] = np.array(gr)
            out[key + 'fac'] = np.array(k.fac)
            out[key + 'deltap'] = np.array(k.get_deltap())
            out[key + 'radius_scale'] = np.array(k.radius_scale)
    path = os.path.join(HERE, 'kernels.npz')
    np.savez_compressed(path, **out)
    print('wrote', path)


def _import_solid_mech():
    """pysph.sph.solid_mech.basic imports pysph.base.utils (-> cyarray, absent)
    only for a particle-array factory; give it a placeholder module.  The one
    compiled helper its equations need, linalg3.eigen_decomposition /
    transform_diag_inv, is the reference's own linalg3.pyx built into
    oracle/_ref (oracle/build_ref.sh)."""
    import types
    if 'pysph.base.utils' not in sys.modules:
        m = types.ModuleType('pysph.base.utils')
        m.get_particle_array = lambda *a, **k: None
        sys.modules['pysph.base.utils'] = m
    sys.path.insert(0, os.path.join(REPO, 'oracle', '_ref'))
    import linalg3
    from pysph.sph.solid_mech import basic as sm

    def loop(self, d_idx, d_rho, d_p, d_s00, d_s01, d_s02, d_s11, d_s12, d_s22,
             d_r00, d_r01, d_r02, d_r11, d_r12, d_r22):
        # solid_mech/basic.py:170-242 with the cython-only matrix declarations
        # replaced by numpy arrays; eigen solver = compiled reference code
        rhoi = d_rho[d_idx]
        rhoi21 = 1. / (rhoi * rhoi)
        p = d_p[d_idx]
        S = np.array([[d_s00[d_idx] - p, d_s01[d_idx], d_s02[d_idx]],
                      [d_s01[d_idx], d_s11[d_idx] - p, d_s12[d_idx]],
                      [d_s02[d_idx], d_s12[d_idx], d_s22[d_idx] - p]])
        V, R = linalg3.py_eigen_decompose_eispack(S)
        rd = np.zeros(3)
        for k in range(3):
            rd[k] = -self.eps * V[k] * rhoi21 if V[k] > 0 else 0
        Rab = linalg3.py_transform_diag_inv(rd, np.ascontiguousarray(R))
        d_r00[d_idx] = float(Rab[0][0]); d_r11[d_idx] = float(Rab[1][1])
        d_r22[d_idx] = float(Rab[2][2]); d_r12[d_idx] = float(Rab[1][2])
        d_r02[d_idx] = float(Rab[0][2]); d_r01[d_idx] = float(Rab[0][1])
    sm.MonaghanArtificialStress.loop = loop
    return sm


EL_PROPS = ['x', 'y', 'z', 'u', 'v', 'w', 'h', 'm', 'rho', 'p', 'cs',
            'arho', 'au', 'av', 'aw', 'ax', 'ay', 'az',
            'v00', 'v01', 'v02', 'v10', 'v11', 'v12', 'v20', 'v21', 'v22',
            's00', 's01', 's02', 's11', 's12', 's22',
            'as00', 'as01', 'as02', 'as11', 'as12', 'as22',
            'r00', 'r01', 'r02', 'r11', 'r12', 'r22']


def _elastic_case(fname, dim, n1, seed):
    sm = _import_solid_mech()
    from pysph.sph.basic_equations import (
        ContinuityEquation, MonaghanArtificialViscosity, XSPHCorrection,
        VelocityGradient2D, VelocityGradient3D)
    rng = np.random.default_rng(seed)
    dx = 1.0 / n1
    g = (np.arange(n1) + 0.5) * dx
    if dim == 2:
        x, y = [a.ravel() for a in np.meshgrid(g, g, indexing='ij')]
        z = np.zeros_like(x)
    else:
        x, y, z = [a.ravel() for a in np.meshgrid(g, g, g, indexing='ij')]
    n = x.size
    hdx = 1.3
    rho0, E, nu = 1.2, 1e4, 0.3975           # rings.py parameters scaled
    G = E / (2. * (1. + nu))
    c0 = np.sqrt(E / (3 * (1. - 2 * nu) * rho0))
    kernel = CubicSpline(dim=dim)
    props = {k: rng.uniform(-1, 1, n) for k in EL_PROPS}
    props.update(
        x=x + 0.1 * dx * rng.uniform(-1, 1, n),
        y=y + 0.1 * dx * rng.uniform(-1, 1, n),
        z=(z + 0.1 * dx * rng.uniform(-1, 1, n)) if dim == 3 else z,
        h=hdx * dx * np.ones(n), m=rho0 * dx ** dim * np.ones(n),
        rho=rho0 * (1 + 0.02 * rng.uniform(-1, 1, n)),
        cs=c0 * np.ones(n))
    if dim == 2:
        props['w'] = np.zeros(n)
    for k in ('s00', 's01', 's02', 's11', 's12', 's22'):
        props[k] = 0.05 * E * rng.uniform(-1, 1, n)
    consts = dict(wdeltap=[kernel.kernel(rij=dx, h=hdx * dx)], n=[4.0], G=[G],
                  E=[E], nu=[nu], rho_ref=[rho0], c0_ref=[c0])
    VG = VelocityGradient2D if dim == 2 else VelocityGradient3D
    eqs = [
        Group(equations=[
            sm.IsothermalEOS('solid', sources=None),
            VG(dest='solid', sources=['solid']),
            sm.MonaghanArtificialStress(dest='solid', sources=None, eps=0.3)]),
        Group(equations=[
            ContinuityEquation(dest='solid', sources=['solid']),
            sm.MomentumEquationWithStress(dest='solid', sources=['solid']),
            MonaghanArtificialViscosity(dest='solid', sources=['solid'],
                                        alpha=1.0, beta=1.0),
            sm.HookesDeviatoricStressRate(dest='solid', sources=None),
            XSPHCorrection(dest='solid', sources=['solid'], eps=0.5)]),
    ]
    pas = [ListPA('solid', props, n, constants=consts)]
    out = {}
    for k, v in pas[0].properties.items():
        out['in/solid/%s' % k] = np.array(v)
    out['nreal/solid'] = np.array(n)
    for k, v in consts.items():
        out['const/solid/%s' % k] = np.array(v, dtype=float)
    a_eval = AccelerationEval(pas, eqs, kernel)
    nnps = PyLinkedListNNPS(dim, pas, radius_scale=kernel.radius_scale)
    nnps.update()
    check_nnps(nnps)
    RefEval(a_eval, nnps).compute(0.0, 1e-4)
    for k, v in pas[0].properties.items():
        out['out/solid/%s' % k] = np.array(v)
    out['nnps/cell_size'] = np.array(nnps.cell_size)
    out['nnps/xmin'] = np.array(nnps.xmin)
    out['nnps/xmax'] = np.array(nnps.xmax)
    out['nnps/ncells_per_dim'] = np.array(nnps.nc)
    out['nnps/n_cells'] = np.array(nnps.n_cells)
    out['t'] = np.array(0.0)
    out['dt'] = np.array(1e-4)
    out['meta/dx'] = np.array(dx)
    out['meta/dim'] = np.array(dim)
    path = os.path.join(HERE, fname)
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB')


def case_elastic_2d():
    """ElasticSolidsScheme equation set (solid_mech/basic.py:604-651) on a 2-D
    block, CubicSpline(dim=2): the reference's own equation classes; eigen
    solver = the reference's compiled linalg3."""
    _elastic_case('elastic_2d.npz', 2, 14, 5)


def case_elastic_3d():
    """Same set with VelocityGradient3D (what BASELINE config 5 needs)."""
    _elastic_case('elastic_3d.npz', 3, 7, 6)


def case_steppers():
    """The reference's stepper methods (integrator_step.py:38-93, 257-299)
    executed as plain Python on 24 random particles."""
    from inspect import getfullargspec
    from pysph.sph.integrator_step import WCSPHStep, TransportVelocityStep
    rng = np.random.default_rng(17)
    n = 24
    from pysph.sph.integrator_step import SolidMechStep
    names = ['x', 'y', 'z', 'u', 'v', 'w', 'rho', 'x0', 'y0', 'z0', 'u0', 'v0',
             'w0', 'rho0', 'au', 'av', 'aw', 'ax', 'ay', 'az', 'arho', 'uhat',
             'vhat', 'what', 'auhat', 'avhat', 'awhat', 'vmag2',
             'e', 'ae', 'e0', 's00', 's01', 's02', 's11', 's12', 's22',
             'as00', 'as01', 'as02', 'as11', 'as12', 'as22',
             's000', 's010', 's020', 's110', 's120', 's220']
    base = {k: rng.uniform(-1, 1, n) for k in names}
    out = {'in/' + k: v.copy() for k, v in base.items()}
    dt = 0.0123
    out['dt'] = np.array(dt)

    def run(step, methods, tag):
        st = {k: [float(x) for x in v] for k, v in base.items()}
        for meth in methods:
            f = getattr(step, meth)
            args = [a for a in getfullargspec(f).args if a != 'self']
            for i in range(n):
                ns = dict(('d_' + k, v) for k, v in st.items())
                ns.update(d_idx=i, dt=dt, t=0.0)
                f(*[ns[a] for a in args])
            for k, v in st.items():
                out['%s/%s/%s' % (tag, meth, k)] = np.array(v)
    run(WCSPHStep(), ['initialize', 'stage1', 'stage2'], 'wcsph')
    run(TransportVelocityStep(), ['stage1', 'stage2'], 'tvf')
    run(SolidMechStep(), ['initialize', 'stage1', 'stage2'], 'solid')
    path = os.path.join(HERE, 'steppers.npz')
    np.savez_compressed(path, **out)
    print('wrote', path)


if __name__ == '__main__':
    which = sys.argv[1:] or ['kernels', 'sd_1d', 'wcsph_cube_varh', 'tvf_cube',
                             'wcsph_dam', 'wcsph_dam_varh', 'steppers', 'elastic_2d', 'elastic_3d', 'tvf_wall']
    for w in which:
        globals()['case_' + w]()
