"""The scenarios of the reference's integrator tests
(pysph/sph/tests/test_integrator.py) on the HIP backend: user-defined steppers
translated to device code, ``py_stage*`` host hooks, helper functions, the
leap-frog / PEFRL / Euler integrators on a harmonic oscillator and the
adaptive-time-step rules."""
import numpy as np
import pytest

from pysph_amd.equations import Equation
from pysph_amd.integrator import (EulerIntegrator, IntegratorStep, LeapFrogIntegrator,
                                  LeapFrogStep, PECIntegrator, PEFRLIntegrator, PEFRLStep)


class SHM(Equation):
    """simple harmonic oscillator (test_integrator.py:22-27)"""

    def initialize(self, d_idx, d_x, d_au):
        d_au[d_idx] = -d_x[d_idx]


class EulerXStep(IntegratorStep):
    def stage1(self, d_idx, d_x, d_u, dt):
        d_x[d_idx] += dt * d_u[d_idx]


class S1Step(IntegratorStep):
    def py_stage1(self, dest, t, dt):
        self.called_with1 = t, dt
        dest.u[:] = 1.0

    def stage1(self, d_idx, d_u, d_au, dt):
        d_u[d_idx] += d_au[d_idx] * dt * 0.5

    def stage2(self, d_idx, d_x, d_u, d_au, dt):
        d_u[d_idx] += 0.5 * dt * d_au[d_idx]
        d_x[d_idx] += dt * d_u[d_idx]


class S12Step(IntegratorStep):
    """explicit push/pull in the hooks, as the reference's GPU variant does"""

    def py_stage1(self, dest, t, dt):
        self.called_with1 = t, dt
        dest.u[:] = 1.0
        if dest.gpu:
            dest.gpu.push('u')

    def stage1(self, d_idx, d_u, d_au, dt):
        d_u[d_idx] += d_au[d_idx] * dt * 0.5

    def py_stage2(self, dest, t, dt):
        self.called_with2 = t, dt
        if dest.gpu:
            dest.gpu.pull('u')
        dest.u += 0.5
        if dest.gpu:
            dest.gpu.push('u')

    def stage2(self, d_idx, d_x, d_u, d_au, dt):
        d_u[d_idx] += 0.5 * dt * d_au[d_idx]
        d_x[d_idx] += dt * d_u[d_idx]


class OnlyPyStep(IntegratorStep):
    def py_stage1(self, dest, t, dt):
        self.called_with1 = t, dt
        dest.x[:] = 0.0
        dest.u[:] = 1.0

    def py_stage2(self, dest, t, dt):
        self.called_with2 = t, dt
        dest.u += 0.5
        dest.x += 0.5


def twice(dt=0.0):
    return dt * 2.0


def scaled(a, fac=1.0):
    tmp = twice(a)
    if fac > 1.5:
        return tmp * fac
    return tmp


class StepWithHelper(IntegratorStep):
    def _get_helpers_(self):
        return [twice]

    def stage1(self, d_idx, d_u, d_au, dt):
        d_u[d_idx] += d_au[d_idx] * twice(dt)


class StepWithNestedHelper(IntegratorStep):
    def stage1(self, d_idx, d_u, d_au, dt):
        d_u[d_idx] += d_au[d_idx] * scaled(dt, fac=2.0)


def make_pa(n=1):
    from pysph_amd.particle_array import get_particle_array
    x = np.asarray([1.0] + [0.0] * (n - 1))
    pa = get_particle_array(name='fluid', x=x, u=np.zeros(n), h=np.ones(n), m=np.ones(n))
    for prop in ('ax', 'ay', 'az', 'ae', 'arho', 'e'):
        pa.add_property(prop)
    return pa


def setup(pa, integrator, sync='auto', extra=()):
    from pysph_amd import device as dev
    from pysph_amd import kernels as K
    from pysph_amd.acceleration_eval import AccelerationEval, SPHCompiler
    from pysph_amd.nnps import HipNNPS
    kernel = K.CubicSpline(dim=1)
    arrays = [pa] + list(extra)
    a_eval = AccelerationEval(arrays, [SHM(dest='fluid', sources=None)], kernel)
    ctx = dev.HipContext(0)
    SPHCompiler(a_eval, integrator=integrator, ctx=ctx, sync=sync).compile()
    if sync == 'manual':            # device-resident state: the caller moves the data
        for a in arrays:
            a.gpu.push()
    nnps = HipNNPS(kernel.dim, arrays, radius_scale=kernel.radius_scale, ctx=ctx,
                   sync=(sync == 'auto'))
    a_eval.set_nnps(nnps)
    integrator.set_nnps(nnps)
    return ctx


def integrate(integrator, dt, tf, callback):
    t = 0.0
    while t < tf:
        integrator.step(t, dt)
        callback(t + dt)
        t += dt


ALL_STEPPERS = (LeapFrogStep, PEFRLStep, EulerXStep, S1Step, S12Step, OnlyPyStep,
                StepWithHelper, StepWithNestedHelper)


def prebuild():
    """every stage family of this file, built without a GPU (called by
    tests/prebuild_generated.py)"""
    from pysph_amd import kernels as K
    from pysph_amd.integrator import generated_stages
    n = 0
    pa = make_pa()
    for cls in ALL_STEPPERS:
        n += len(generated_stages(cls(), pa, 0, K.kernel_id(K.CubicSpline(dim=1))))
    return n


# ---------------------------------------------------------------------------
# CPU: translation only
# ---------------------------------------------------------------------------
def test_stepper_stage_translates_with_helpers():
    from pysph_amd.integrator import stage_family
    fam = stage_family(StepWithNestedHelper(), 'stage1', make_pa(), 1)
    src = fam.source
    # helpers are emitted before use, callee first; keyword and default
    # arguments are resolved at the call site
    assert src.index('gen_helper_twice(double dt)') < src.index('gen_helper_scaled(double a_, double fac)')
    assert 'gen_helper_scaled(dt, 2.0)' in src
    assert fam.dout == ['u'] and fam.din == ['au']


def test_missing_stepper_arrays_are_detected():
    """test_integrator.py:31-72: RuntimeError naming what the array lacks"""
    from pysph_amd.integrator import stage_family
    from pysph_amd.particle_array import get_particle_array
    pa = get_particle_array(name='fluid', x=np.ones(1), h=np.ones(1), m=np.ones(1))
    with pytest.raises(RuntimeError) as e:
        stage_family(LeapFrogStep(), 'stage1', pa, 1)
    assert 'ax' in str(e.value)


# ---------------------------------------------------------------------------
# GPU
# ---------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize('sync', ['auto', 'manual'])
def test_leapfrog_conserves_energy_and_is_second_order(sync):
    """test_integrator.py:262-282, :362-398"""
    pa = make_pa()
    integrator = LeapFrogIntegrator(fluid=LeapFrogStep())
    setup(pa, integrator, sync)
    tf = np.pi
    errs = []
    for dt in (0.02 * tf, 0.01 * tf):
        pa.x[0], pa.u[0] = 1.0, 0.0
        if sync == 'manual':
            pa.gpu.push()
        energy = []

        def callback(t):
            if sync == 'manual':
                pa.gpu.pull('x', 'u')
            energy.append(0.5 * (pa.x[0] ** 2 + pa.u[0] ** 2))

        callback(0.0)
        integrate(integrator, dt, tf, callback)
        errs.append(np.max(np.abs(np.asarray(energy) - 0.5)))
    assert errs[0] < 5e-4
    assert errs[1] < errs[0]
    assert abs(errs[0] / errs[1] - 4.0) < 5e-3


@pytest.mark.gpu
def test_pefrl_is_fourth_order():
    """test_integrator.py:419-472"""
    pa = make_pa()
    integrator = PEFRLIntegrator(fluid=PEFRLStep())
    setup(pa, integrator)
    tf = np.pi
    errs = []
    for dt in (0.1 * tf, 0.05 * tf):
        pa.x[0], pa.u[0] = 1.0, 0.0
        energy = [0.5]
        integrate(integrator, dt, tf,
                  lambda t: energy.append(0.5 * (pa.x[0] ** 2 + pa.u[0] ** 2)))
        errs.append(np.max(np.abs(np.asarray(energy) - 0.5)))
    assert errs[0] < 5e-5
    assert errs[0] / errs[1] > 16.0


@pytest.mark.gpu
def test_integrator_calls_py_stage1():
    """test_integrator.py:284-309"""
    pa = make_pa()
    stepper = S1Step()
    integrator = LeapFrogIntegrator(fluid=stepper)
    setup(pa, integrator)
    calls = []
    integrate(integrator, 1.0, 1.0, calls.append)
    assert len(calls) == 1
    assert stepper.called_with1 == (0.0, 1.0)
    np.testing.assert_array_almost_equal(pa.x, [1.5])
    np.testing.assert_array_almost_equal(pa.u, [0.5])


@pytest.mark.gpu
@pytest.mark.parametrize('sync', ['auto', 'manual'])
def test_integrator_calls_py_stage1_stage2(sync):
    """test_integrator.py:311-338 (manual sync: the hooks' own push/pull is
    what moves the data, as with the reference's GPU backend)"""
    pa = make_pa()
    stepper = S12Step()
    integrator = LeapFrogIntegrator(fluid=stepper)
    setup(pa, integrator, sync)
    if sync == 'manual':
        pa.gpu.push()
    integrate(integrator, 1.0, 1.0, lambda t: None)
    if sync == 'manual':
        pa.gpu.pull('x', 'u')
    assert stepper.called_with1 == (0.0, 1.0)
    assert stepper.called_with2 == (0.5, 1.0)
    np.testing.assert_array_almost_equal(pa.x, [2.0])
    np.testing.assert_array_almost_equal(pa.u, [1.0])


@pytest.mark.gpu
def test_integrator_calls_only_py_when_no_stage():
    """test_integrator.py:340-360"""
    pa = make_pa()
    stepper = OnlyPyStep()
    integrator = LeapFrogIntegrator(fluid=stepper)
    setup(pa, integrator)
    integrate(integrator, 1.0, 1.0, lambda t: None)
    assert stepper.called_with1 == (0.0, 1.0)
    assert stepper.called_with2 == (0.5, 1.0)
    np.testing.assert_array_almost_equal(pa.x, [0.5])
    np.testing.assert_array_almost_equal(pa.u, [1.5])


@pytest.mark.gpu
@pytest.mark.parametrize('cls,fac', [(StepWithHelper, 2.0), (StepWithNestedHelper, 4.0)])
def test_helper_can_be_used_with_stepper(cls, fac):
    """test_integrator.py:400-417"""
    pa = make_pa()
    integrator = EulerIntegrator(fluid=cls())
    setup(pa, integrator)
    integrate(integrator, 0.5, 1.0, lambda t: None)
    assert pa.u[0] == -fac * pa.x[0]


@pytest.mark.gpu
def test_compiling_detects_missing_arrays():
    from pysph_amd.particle_array import get_particle_array, get_particle_array_wcsph
    fluid = get_particle_array_wcsph(name='fluid', x=np.ones(1), h=np.ones(1), m=np.ones(1))
    solid = get_particle_array(name='solid', x=np.ones(1), h=np.ones(1), m=np.ones(1))
    integrator = PECIntegrator(fluid=LeapFrogStep(), solid=LeapFrogStep())
    with pytest.raises(RuntimeError):
        setup(fluid, integrator, extra=[solid])


@pytest.mark.gpu
def test_compute_time_step_rules():
    """test_integrator.py:111-205: None without constraints, dt_adapt wins
    over dt_cfl, invalid dt_adapt is ignored, cfl*h/dt_cfl otherwise"""
    def run(**props):
        pa = make_pa(2)
        for k, v in props.items():
            pa.add_property(k)
            pa.properties[k][:] = v
        integrator = EulerIntegrator(fluid=EulerXStep())
        setup(pa, integrator)
        return integrator.compute_time_step(0.1, 0.5)

    assert run() is None
    assert run(dt_adapt=[0.1, 0.2]) == 0.1
    assert run(dt_adapt=[0.0, -2.0]) is None
    assert run(dt_adapt=[0.1, 0.2], dt_cfl=[1.0, 1.0]) == 0.1
    assert run(dt_cfl=[1.0, 2.0]) == 0.5 * 1.0 / 2.0


class PyWCSPHStep(IntegratorStep):
    """the WCSPH two-stage stepper as Python bodies (semantics of
    integrator_step.py:38-93; cf. oracle/steppers.py): save the state, then
    q = q0 + f*dt*aq with f = 1/2 and 1"""

    def initialize(self, d_idx, d_x0, d_y0, d_z0, d_x, d_y, d_z, d_u0, d_v0, d_w0, d_u, d_v,
                   d_w, d_rho0, d_rho):
        d_x0[d_idx] = d_x[d_idx]
        d_y0[d_idx] = d_y[d_idx]
        d_z0[d_idx] = d_z[d_idx]
        d_u0[d_idx] = d_u[d_idx]
        d_v0[d_idx] = d_v[d_idx]
        d_w0[d_idx] = d_w[d_idx]
        d_rho0[d_idx] = d_rho[d_idx]

    def stage1(self, d_idx, d_x0, d_y0, d_z0, d_x, d_y, d_z, d_u0, d_v0, d_w0, d_u, d_v, d_w,
               d_rho0, d_rho, d_au, d_av, d_aw, d_ax, d_ay, d_az, d_arho, dt):
        dtb2 = 0.5 * dt
        d_u[d_idx] = d_u0[d_idx] + dtb2 * d_au[d_idx]
        d_v[d_idx] = d_v0[d_idx] + dtb2 * d_av[d_idx]
        d_w[d_idx] = d_w0[d_idx] + dtb2 * d_aw[d_idx]
        d_x[d_idx] = d_x0[d_idx] + dtb2 * d_ax[d_idx]
        d_y[d_idx] = d_y0[d_idx] + dtb2 * d_ay[d_idx]
        d_z[d_idx] = d_z0[d_idx] + dtb2 * d_az[d_idx]
        d_rho[d_idx] = d_rho0[d_idx] + dtb2 * d_arho[d_idx]

    def stage2(self, d_idx, d_x0, d_y0, d_z0, d_x, d_y, d_z, d_u0, d_v0, d_w0, d_u, d_v, d_w,
               d_rho0, d_rho, d_au, d_av, d_aw, d_ax, d_ay, d_az, d_arho, dt):
        d_u[d_idx] = d_u0[d_idx] + dt * d_au[d_idx]
        d_v[d_idx] = d_v0[d_idx] + dt * d_av[d_idx]
        d_w[d_idx] = d_w0[d_idx] + dt * d_aw[d_idx]
        d_x[d_idx] = d_x0[d_idx] + dt * d_ax[d_idx]
        d_y[d_idx] = d_y0[d_idx] + dt * d_ay[d_idx]
        d_z[d_idx] = d_z0[d_idx] + dt * d_az[d_idx]
        d_rho[d_idx] = d_rho0[d_idx] + dt * d_arho[d_idx]


class PyTVFStep(IntegratorStep):
    """the transport-velocity kick-drift-kick stepper as Python bodies
    (semantics of integrator_step.py:257-299)"""

    def stage1(self, d_idx, d_u, d_v, d_w, d_au, d_av, d_aw, d_uhat, d_vhat, d_what, d_auhat,
               d_avhat, d_awhat, d_x, d_y, d_z, dt):
        dtb2 = 0.5 * dt
        d_u[d_idx] += dtb2 * d_au[d_idx]
        d_v[d_idx] += dtb2 * d_av[d_idx]
        d_w[d_idx] += dtb2 * d_aw[d_idx]
        d_uhat[d_idx] = d_u[d_idx] + dtb2 * d_auhat[d_idx]
        d_vhat[d_idx] = d_v[d_idx] + dtb2 * d_avhat[d_idx]
        d_what[d_idx] = d_w[d_idx] + dtb2 * d_awhat[d_idx]
        d_x[d_idx] += dt * d_uhat[d_idx]
        d_y[d_idx] += dt * d_vhat[d_idx]
        d_z[d_idx] += dt * d_what[d_idx]

    def stage2(self, d_idx, d_u, d_v, d_w, d_au, d_av, d_aw, d_vmag2, dt):
        dtb2 = 0.5 * dt
        d_u[d_idx] += dtb2 * d_au[d_idx]
        d_v[d_idx] += dtb2 * d_av[d_idx]
        d_w[d_idx] += dtb2 * d_aw[d_idx]
        d_vmag2[d_idx] = (d_u[d_idx] * d_u[d_idx] + d_v[d_idx] * d_v[d_idx] +
                          d_w[d_idx] * d_w[d_idx])


def golden_stepper_array():
    from conftest import load_golden
    from test_integrator import STEP_PROPS, _pa_from
    g = load_golden('steppers.npz')
    return g, _pa_from(g), STEP_PROPS


def prebuild_golden_steppers():
    from pysph_amd.integrator import generated_stages
    g, pa, _ = golden_stepper_array()
    return sum(len(generated_stages(cls(), pa, 0, 1)) for cls in (PyWCSPHStep, PyTVFStep))


@pytest.mark.gpu
def test_generated_steppers_match_reference_golden_vectors():
    """Python-bodied steppers through the generated path vs the outputs of the
    reference's own WCSPHStep / TransportVelocityStep methods
    (tests/golden/steppers.npz), stage by stage; fma contraction may move the
    last bit."""
    import ctypes as C
    from pysph_amd import device as dev
    from pysph_amd.integrator import generated_stages
    g, _, props = golden_stepper_array()
    dt = float(g['dt'])
    kern = dev.SphKernel(1, 3, 1.0, 2.0, 0.5)
    for cls, tag, names in ((PyWCSPHStep, 'wcsph', ['initialize', 'stage1', 'stage2']),
                            (PyTVFStep, 'tvf', ['stage1', 'stage2'])):
        _, pa, _ = golden_stepper_array()
        ctx = dev.HipContext(0)
        gpu = dev.attach(pa, ctx)
        gpu.push()
        stages = generated_stages(cls(), pa, gpu.array_id, 1)
        for name in names:
            st = stages[name]
            dev._check(ctx.lib.sph_eval_generated(ctx._h, C.addressof(kern), C.addressof(st.cf),
                                                  0.0, dt))
            gpu.pull()
            for k in props:
                ref = g['%s/%s/%s' % (tag, name, k)]
                assert np.allclose(pa.properties[k], ref, rtol=1e-15, atol=1e-16), (tag, name, k)


# ---------------------------------------------------------------------------
# sph/tests/test_multi_group_integrator.py: one acceleration evaluator per stage
# ---------------------------------------------------------------------------
class Eq1(Equation):
    def initialize(self, d_idx, d_au):
        d_au[d_idx] = 1.0


class Eq2(Equation):
    def initialize(self, d_idx, d_au):
        d_au[d_idx] += 1.0


class TwoKickStep(IntegratorStep):
    def stage1(self, d_idx, d_u, d_au, dt):
        d_u[d_idx] += d_au[d_idx] * dt

    def stage2(self, d_idx, d_u, d_au, dt):
        d_u[d_idx] += d_au[d_idx] * dt


def multi_stage_array():
    from pysph_amd.particle_array import get_particle_array
    n = 10
    x = np.linspace(0, 1, n)
    return get_particle_array(name='fluid', x=x, h=np.ones_like(x) * 1.05 / (n - 1),
                              m=np.ones_like(x), au=0.0, u=0.0)


def multi_stage_equations():
    from pysph_amd.equations import MultiStageEquations
    return MultiStageEquations([[Eq1(dest='fluid', sources=['fluid'])],
                                [Eq2(dest='fluid', sources=['fluid'])]])


def prebuild_multi_stage():
    from pysph_amd import kernels as K
    from pysph_amd.acceleration_eval import _CGroup, make_acceleration_evals
    from pysph_amd.integrator import generated_stages
    pa = multi_stage_array()
    n = len(generated_stages(TwoKickStep(), pa, 0, 1))
    for a in make_acceleration_evals([pa], multi_stage_equations(), K.CubicSpline(dim=1)):
        for g in a.equation_groups:
            n += len(_CGroup(g, {'fluid': 0}, {'fluid': pa}, 1).units)
    return n


@pytest.mark.gpu
def test_different_accels_per_stage():
    """test_multi_group_integrator.py:52-98: MultiStageEquations ->
    make_acceleration_evals -> SPHCompiler with a user integrator calling
    compute_accelerations(index, update_nnps=False)"""
    from pysph_amd import device as dev
    from pysph_amd import kernels as K
    from pysph_amd.acceleration_eval import SPHCompiler, make_acceleration_evals
    from pysph_amd.integrator import Integrator
    from pysph_amd.nnps import HipNNPS

    class TwoEvalIntegrator(Integrator):
        def one_timestep(self, t, dt):
            self.compute_accelerations(0, update_nnps=False)
            self.stage1()
            self.do_post_stage(dt, 1)
            self.compute_accelerations(1, update_nnps=False)
            self.stage2()
            self.update_domain()
            self.do_post_stage(dt, 2)

    pa = multi_stage_array()
    kernel = K.CubicSpline(dim=1)
    a_evals = make_acceleration_evals([pa], multi_stage_equations(), kernel)
    assert len(a_evals) == 2
    integrator = TwoEvalIntegrator(fluid=TwoKickStep())
    ctx = dev.HipContext(0)
    SPHCompiler(a_evals, integrator=integrator, ctx=ctx).compile()
    nnps = HipNNPS(kernel.dim, [pa], radius_scale=kernel.radius_scale, ctx=ctx)
    nnps.update()
    for ae in a_evals:
        ae.set_nnps(nnps)
    integrator.set_nnps(nnps)
    integrator.step(0.0, 0.1)
    one = np.ones_like(pa.x)
    np.testing.assert_array_almost_equal(pa.au, 2.0 * one)
    np.testing.assert_array_almost_equal(pa.u, 0.3 * one)
