"""Integrators and steppers driving the HIP stage kernels.

Mirrors (pypr/pysph):

* ``IntegratorStep`` subclasses ``WCSPHStep`` and ``TransportVelocityStep``
  (pysph/sph/integrator_step.py:38-93, 257-299) -- specification objects; the
  per-particle bodies are ``k_stage`` in csrc/sph_integrate.hip;
* ``Integrator`` / ``PECIntegrator`` / ``EPECIntegrator`` with the user-visible
  ``one_timestep(t, dt)`` written against ``self.initialize() / stage1() /
  stage2() / compute_accelerations() / update_domain() / do_post_stage()``
  (pysph/sph/integrator.py:20-420).  The reference *source-inlines*
  ``one_timestep`` into the generated Cython class
  (integrator_cython_helper.py:177-181); here the same function is executed with
  ``self`` bound to ``HipIntegrator`` -- the ``c_integrator`` protocol of
  integrator_cython.mako:24-85 (``t, dt, step, set_nnps,
  set_parallel_manager, set_post_stage_callback``);
* ``compute_time_step(dt, cfl)`` (integrator.py:161-200) with the
  ``max(dt_cfl), max(dt_force), max(dt_visc), min(h)`` reductions done on the
  device (``sph_reduce_max/min``).

State stays device-resident across steps (``sync='manual'`` acceleration
evals); ``pa.gpu.pull()`` brings results back for output.
"""
import math

from . import device as dev

STEP_WCSPH = 1
STEP_TVF = 2


class IntegratorStep(object):
    """pysph/sph/integrator_step.py:15-35."""
    _kind = 0

    def __repr__(self):
        return '%s()' % self.__class__.__name__


class WCSPHStep(IntegratorStep):
    """integrator_step.py:38-93."""
    _kind = STEP_WCSPH


class TransportVelocityStep(IntegratorStep):
    """integrator_step.py:257-299."""
    _kind = STEP_TVF


_STEP_KINDS = {'WCSPHStep': STEP_WCSPH, 'TransportVelocityStep': STEP_TVF}


def stepper_kind(step):
    name = type(step).__name__
    if name not in _STEP_KINDS:
        raise NotImplementedError(
            'HIP backend: stepper %s has no stage kernel (have %s)' %
            (name, sorted(_STEP_KINDS)))
    return _STEP_KINDS[name]


class Integrator(object):
    """pysph/sph/integrator.py:20-360 (default one_timestep == PEC)."""

    def __init__(self, **kw):
        for name, step in kw.items():
            if not isinstance(step, IntegratorStep) and \
                    type(step).__name__ not in _STEP_KINDS:
                raise ValueError('Stepper %s must be an instance of '
                                 'IntegratorStep' % (step,))
        self.steppers = kw
        self.parallel_manager = None
        self.nnps = None
        self.acceleration_evals = None
        self.c_integrator = None
        self.fixed_h = False
        self.h_minimum = None

    # -- reference public interface -------------------------------------
    def set_acceleration_evals(self, a_evals):
        self.acceleration_evals = a_evals

    def set_nnps(self, nnps):
        self.nnps = nnps
        self.c_integrator.set_nnps(nnps)

    def set_compiled_object(self, c_integrator):
        self.c_integrator = c_integrator

    def set_parallel_manager(self, pm):
        self.parallel_manager = pm
        self.c_integrator.set_parallel_manager(pm)

    def set_post_stage_callback(self, callback):
        self.c_integrator.set_post_stage_callback(callback)

    def set_fixed_h(self, fixed_h):
        self.fixed_h = fixed_h
        if fixed_h:
            self.compute_h_minimum()

    def step(self, time, dt):
        self.c_integrator.step(time, dt)

    def compute_accelerations(self, index=0, update_nnps=True):
        """integrator.py:274-286."""
        if update_nnps:
            if self.parallel_manager:
                self.parallel_manager.update()
            self.nnps.update()
        c = self.c_integrator
        self.acceleration_evals[index].compute(c.t, c.dt)

    def initial_acceleration(self, t, dt):
        self.acceleration_evals[0].compute(t, dt)

    def update_domain(self):
        self.nnps.update_domain()

    def compute_h_minimum(self):
        hmin = 1.0
        for pa in self.acceleration_evals[0].particle_arrays:
            if pa.get_number_of_particles(True):
                hmin = min(hmin, pa.gpu.min('h'))
        pm = self.parallel_manager
        if pm is not None and hasattr(pm, 'reduce_min'):
            hmin = pm.reduce_min([hmin])[0]        # parallel_manager.pyx:463
        self.h_minimum = hmin

    def _get_dt_adapt_factors(self):
        factors = [-1.0, -1.0, -1.0]
        for pa in self.acceleration_evals[0].particle_arrays:
            for i, name in enumerate(('dt_cfl', 'dt_force', 'dt_visc')):
                if name in pa.properties and dev.prop_id(name) >= 0 and \
                        pa.get_number_of_particles(True):
                    factors[i] = max(factors[i], pa.gpu.max(name))
        pm = self.parallel_manager
        if pm is not None and hasattr(pm, 'reduce_max'):
            factors = pm.reduce_max(factors)
        return factors

    def compute_time_step(self, dt, cfl):
        """integrator.py:161-200."""
        cfl_f, force_f, visc_f = self._get_dt_adapt_factors()
        if not self.fixed_h or self.h_minimum is None:
            self.compute_h_minimum()
        hmin = self.h_minimum
        dt_cfl = dt_force = dt_visc = float('inf')
        if cfl_f > 0:
            dt_cfl = hmin / cfl_f
        if force_f > 0:
            dt_force = math.sqrt(hmin / math.sqrt(force_f))
        if visc_f > 0:
            dt_visc = hmin / visc_f
        dt_min = min(dt_cfl, dt_force, dt_visc)
        if dt_min <= 0.0 or math.isinf(dt_min):
            return None
        return cfl * dt_min

    # -- user-overridable ----------------------------------------------------
    def one_timestep(self, t, dt):
        """integrator.py:227-246 (predict-evaluate-correct)."""
        self.initialize()
        self.stage1()
        self.update_domain()
        self.do_post_stage(0.5 * dt, 1)
        self.compute_accelerations()
        self.stage2()
        self.update_domain()
        self.do_post_stage(dt, 2)


class PECIntegrator(Integrator):
    """integrator.py:300-358."""


class EPECIntegrator(Integrator):
    """integrator.py:367-420."""

    def one_timestep(self, t, dt):
        self.initialize()
        self.compute_accelerations()
        self.stage1()
        self.update_domain()
        self.do_post_stage(0.5 * dt, 1)
        self.compute_accelerations()
        self.stage2()
        self.update_domain()
        self.do_post_stage(dt, 2)


class HipIntegrator(object):
    """The ``c_integrator`` object (integrator_cython.mako:24-113)."""

    def __init__(self, integrator, acceleration_eval_obj, ctx=None):
        self.integrator = integrator
        self.acceleration_eval = acceleration_eval_obj
        self.ctx = ctx or acceleration_eval_obj.ctx
        self.lib = self.ctx.lib
        self.t = self.dt = self.orig_t = 0.0
        self._post_stage_callback = None
        self._stages = []
        for name in sorted(integrator.steppers):
            helper = acceleration_eval_obj.helpers[name]
            self._stages.append((helper.array_id,
                                 stepper_kind(integrator.steppers[name])))

    def set_nnps(self, nnps):
        pass

    def set_parallel_manager(self, pm):
        pass

    def set_post_stage_callback(self, callback):
        self._post_stage_callback = callback

    def compute_accelerations(self, index=0, update_nnps=True):
        self.integrator.compute_accelerations(index, update_nnps)

    def update_domain(self):
        self.integrator.update_domain()

    def do_post_stage(self, stage_dt, stage):
        self.t = self.orig_t + stage_dt
        if self._post_stage_callback is not None:
            self._post_stage_callback(self.t, self.dt, stage)

    def _sweep(self, stage):
        for aid, kind in self._stages:
            dev._check(self.lib.sph_integrate_stage(self.ctx._h, aid, kind,
                                                    stage, self.dt))

    def initialize(self):
        self._sweep(0)

    def stage1(self):
        self._sweep(1)

    def stage2(self):
        self._sweep(2)

    def step(self, t, dt):
        self.orig_t = self.t = t
        self.dt = dt
        # the reference inlines the *source* of one_timestep into this class
        type(self.integrator).one_timestep(self, t, dt)


def setup_integrator(integrator, a_evals, nnps, ctx=None):
    """What ``SPHCompiler`` + ``Solver.setup`` do for the integrator
    (sph_compiler.py:27-59, solver.py:186-260): install the compiled object and
    hand over the acceleration evals and the NNPS."""
    if not isinstance(a_evals, (list, tuple)):
        a_evals = [a_evals]
    integrator.set_acceleration_evals(list(a_evals))
    obj = HipIntegrator(integrator, a_evals[0].c_acceleration_eval, ctx)
    integrator.set_compiled_object(obj)
    integrator.set_nnps(nnps)
    return obj
