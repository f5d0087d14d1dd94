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
    """C-ABI id of a hand-written stage kernel, or None: the stepper's own
    ``initialize / stage1 / stage2 ...`` bodies are then translated like
    no-source equations (pysph_amd/codegen.py) -- the reference compiles
    user-defined steppers the same way it compiles equations
    (integrator_cython_helper.py:24-120)."""
    return _STEP_KINDS.get(type(step).__name__)


STAGE_NAMES = ('initialize', 'stage1', 'stage2', 'stage3', 'stage4', 'stage5')


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
        pm = self.parallel_manager
        if update_nnps:
            if pm:
                pm.update()
            self.nnps.update()
        c = self.c_integrator
        self.acceleration_evals[index].compute(c.t, c.dt)
        # The reference counts its ghosts first and never evaluates on incomplete ones
        # (parallel_manager.pyx:1085-1157).  The round-trip-free exchange appends fixed-capacity
        # messages and learns the counts afterwards: they are read HERE, with the evaluation queued and
        # nothing having consumed it yet -- a face that had outgrown its message has been repeated the
        # counted way by verify(), and the update + evaluation (which only read the particles' state and
        # overwrite their own results) run again.
        checks = [getattr(pm, 'verify', None) if pm else None,
                  # ... and the periodic images of a device domain manager made without a round trip (update_domain)
                  getattr(getattr(self.nnps, 'domain', None), 'verify', None)]
        checks = [v for v in checks if v is not None and update_nnps]
        while not all([v() for v in checks]):
            self.nnps.update()
            self.acceleration_evals[index].compute(c.t, c.dt)

    def initial_acceleration(self, t, dt):
        self.acceleration_evals[0].compute(t, dt)

    def update_domain(self):
        self.nnps.update_domain()

    def _reduce(self, pa, name, op):
        """max/min of one property: on the device when the state is
        device-resident (sync='manual') and the property is one the device
        owns (h, or anything an equation writes); from the host array
        otherwise (sync='auto', or a property only the user sets)."""
        ev = getattr(self.acceleration_evals[0], 'c_acceleration_eval', None)
        manual = getattr(ev, 'sync', 'auto') == 'manual'
        written = set()
        if ev is not None:
            written = set(ev.outputs.get(pa.name, ())) | set(ev.outputs_exact.get(pa.name, ()))
        if manual and pa.gpu is not None and dev.prop_id(name) >= 0 and \
                (name == 'h' or name in written):
            return pa.gpu.max(name) if op == 'max' else pa.gpu.min(name)
        import numpy as np
        from .particle_array import get_npy
        arr = get_npy(pa, name)
        return float(np.max(arr) if op == 'max' else np.min(arr))

    def _get_explicit_dt_adapt(self):
        """integrator.py:83-117: a user-specified ``dt_adapt`` property is the
        allowed time step."""
        arrays = [pa for pa in self.acceleration_evals[0].particle_arrays
                  if 'dt_adapt' in pa.properties]
        if not arrays:
            return None
        dt_min = float('inf')
        for pa in arrays:
            if pa.get_number_of_particles() > 0:
                dt_min = min(dt_min, self._reduce(pa, 'dt_adapt', 'min'))
        pm = self.parallel_manager
        if pm is not None and hasattr(pm, 'reduce_min'):
            dt_min = pm.reduce_min([dt_min])[0]
        return dt_min if dt_min > 0.0 else None

    def compute_h_minimum(self):
        hmin = 1.0
        for pa in self.acceleration_evals[0].particle_arrays:
            if pa.get_number_of_particles(True):
                hmin = min(hmin, self._reduce(pa, 'h', 'min'))
        pm = self.parallel_manager
        if pm is not None and hasattr(pm, 'reduce_min'):
            hmin = pm.reduce_min([hmin])[0]        # parallel_manager.pyx:463
        self.h_minimum = hmin

    def _get_dt_adapt_factors(self):
        factors = [-1.0, -1.0, -1.0]
        for pa in self.acceleration_evals[0].particle_arrays:
            for i, name in enumerate(('dt_cfl', 'dt_force', 'dt_visc')):
                if name in pa.properties and pa.get_number_of_particles(True):
                    factors[i] = max(factors[i], self._reduce(pa, name, 'max'))
        pm = self.parallel_manager
        if pm is not None and hasattr(pm, 'reduce_max'):
            factors = pm.reduce_max(factors)
        return factors

    def compute_time_step(self, dt, cfl):
        """integrator.py:161-200."""
        dt_adapt = self._get_explicit_dt_adapt()
        if dt_adapt is not None:
            return dt_adapt
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
    # The order of operations of one time step is the user-visible protocol of
    # the reference (its generated integrator inlines the SOURCE of
    # one_timestep, integrator_cython_helper.py:177-181); subclasses may
    # override one_timestep with any sequence of the calls below.  The two
    # built-in sequences are data:
    #   'init'            -> self.initialize()
    #   'accel'           -> self.compute_accelerations()
    #   ('stage', k, f)   -> self.stage<k>(); self.update_domain();
    #                        self.do_post_stage(f * dt, k)
    SEQUENCE = ('init', ('stage', 1, 0.5), 'accel', ('stage', 2, 1.0))  # PEC: pysph/sph/integrator.py:227-246

    def one_timestep(self, t, dt):
        for op in self.SEQUENCE:
            if op == 'init':
                self.initialize()
            elif op == 'accel':
                self.compute_accelerations()
            else:
                _, k, f = op
                getattr(self, 'stage%d' % k)()
                self.update_domain()
                self.do_post_stage(f * dt, k)


class PECIntegrator(Integrator):
    """Predict-evaluate-correct: pysph/sph/integrator.py:300-358."""


class EPECIntegrator(Integrator):
    """Evaluate-predict-evaluate-correct (two acceleration evaluations per
    step, the integrator of the dam-break / Taylor-Green examples):
    pysph/sph/integrator.py:367-420."""
    SEQUENCE = ('init', 'accel', ('stage', 1, 0.5), 'accel', ('stage', 2, 1.0))


def method_properties_of(fn):
    """d_* array names in a stage method's signature"""
    import inspect
    return [a[2:] for a in inspect.signature(fn).parameters
            if a.startswith('d_') and a != 'd_idx']


def stage_family(stepper, method, pa, kernel_kind):
    """One stage method of a user-defined stepper as a generated no-source
    family: the body is per-particle code on d_* arrays, t and dt -- exactly a
    no-source ``Equation.loop``."""
    from .codegen import GeneratedFamily
    from .equations import Equation
    fn = getattr(type(stepper), method)
    body = {'loop': fn}
    if callable(getattr(type(stepper), '_get_helpers_', None)):
        body['_get_helpers_'] = getattr(type(stepper), '_get_helpers_')
    adapter_cls = type('%s_%s' % (type(stepper).__name__, method), (Equation,), body)
    eq = adapter_cls(pa.name, None)
    for k, v in stepper.__dict__.items():       # scalar parameters of the stepper
        if not hasattr(eq, k):
            setattr(eq, k, v)
    missing = [p for p in method_properties_of(fn) if p not in pa.properties
               and p not in getattr(pa, 'constants', {})]
    if missing:
        # sph_compiler.py / integrator_cython_helper.py:122-150 check_array_properties
        raise RuntimeError('stepper %s.%s needs %s, which array %r lacks' % (
            type(stepper).__name__, method, sorted(missing), pa.name))
    return GeneratedFamily(pa.name, [eq], {pa.name: pa}, kernel_kind, name='step')


def generated_stages(stepper, pa, array_id, kernel_kind):
    """{stage name: _GeneratedStage} for every stage method the stepper defines"""
    out = {}
    for m in STAGE_NAMES:
        if callable(getattr(type(stepper), m, None)):
            out[m] = _GeneratedStage(stepper, m, pa, array_id, kernel_kind)
    return out


class _GeneratedStage(object):
    """A built stage family and its ``sph_gen_family`` descriptor."""

    def __init__(self, stepper, method, pa, array_id, kernel_kind):
        import ctypes as C
        self.fam = stage_family(stepper, method, pa, kernel_kind)
        lib = self.fam.load()
        cf = dev.SphGenFamily()
        cf.launch = C.cast(lib.sphgen_launch, C.c_void_p).value
        cf.dest = array_id
        cf.nsrc = 0
        cf.n_din = len(self.fam.din)
        for k, p in enumerate(self.fam.din):
            cf.din[k] = dev.prop_register(p)
        cf.n_dout = len(self.fam.dout)
        for k, p in enumerate(self.fam.dout):
            cf.dout[k] = dev.prop_register(p)
        cf.npar = len(self.fam.params)
        cf.real = 1                                   # steppers act on real particles only
        cf.start_idx, cf.stop_idx = 0, -1
        self.cf = cf

    def run(self, integ, t, dt):
        import ctypes as C
        for k, v in enumerate(self.fam.param_values()):
            self.cf.par[k] = v
        ev = integ.acceleration_eval
        dev._check(integ.lib.sph_eval_generated(ev.ctx._h, C.addressof(ev.ckernel),
                                                C.addressof(self.cf), t, dt))


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
            stepper = integrator.steppers[name]
            pa = acceleration_eval_obj.arrays[name]
            kind = stepper_kind(stepper)
            gen = {}
            if kind is None:
                gen = generated_stages(stepper, pa, helper.array_id,
                                       acceleration_eval_obj.ckernel.kind)
            self._stages.append((helper, stepper, pa, kind, gen))

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
        """One stage over every stepped array: the optional host hook
        ``py_<stage>(dest, t, dt)`` first, then the per-particle stage
        (integrator_cython.mako:87-113).  With the acceleration eval in
        sync='auto' the host arrays are authoritative: they are pushed before
        and pulled after the sweep (and around the hook)."""
        name = STAGE_NAMES[stage]
        auto = getattr(self.acceleration_eval, 'sync', 'manual') == 'auto'
        for helper, stepper, pa, kind, gen in self._stages:
            hook = getattr(stepper, 'py_' + name, None)
            if hook is not None:
                hook(pa, self.t, self.dt)
            if kind is not None:
                if auto:
                    helper.push()
                dev._check(self.lib.sph_integrate_stage(self.ctx._h, helper.array_id, kind,
                                                        stage, self.dt))
                if auto:
                    helper.pull()
            elif name in gen:
                st = gen[name]
                if auto:
                    helper.push(*[p for p in st.fam.dprops if p in pa.properties])
                st.run(self, self.t, self.dt)
                if auto:
                    helper.pull(*[p for p in st.fam.dout if p in pa.properties])

    def initialize(self):
        self._sweep(0)

    def stage1(self):
        self._sweep(1)

    def stage2(self):
        self._sweep(2)

    def stage3(self):
        self._sweep(3)

    def stage4(self):
        self._sweep(4)

    def stage5(self):
        self._sweep(5)

    def step(self, t, dt):
        self.orig_t = self.t = t
        self.dt = dt
        # the reference inlines the *source* of one_timestep into this class
        self.SEQUENCE = type(self.integrator).SEQUENCE
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


# ---------------------------------------------------------------------------
# The reference's other explicit integrators and their steppers
# (pysph/sph/integrator.py:426-517, pysph/sph/integrator_step.py:22-35,
# 708-830).  The integrators are SEQUENCE data like PEC / EPEC above; the
# steppers have no hand-written stage kernel: their stage methods below are
# per-particle bodies that run as generated stage families (`stage_family`),
# written as "kick" (velocities, density, energy by the rates) and "drift"
# (positions by velocity + XSPH correction) with the weights of each scheme.
# ---------------------------------------------------------------------------
class EulerIntegrator(Integrator):
    """Forward Euler: evaluate, then one full step (integrator.py:426-437)."""
    SEQUENCE = ('accel', ('stage', 1, 1.0))


class LeapFrogIntegrator(Integrator):
    """Drift-kick-drift leap-frog: half drift, evaluate at t + dt/2, kick and
    second half drift (integrator.py:464-477)."""
    SEQUENCE = (('stage', 1, 0.5), 'accel', ('stage', 2, 1.0))


class EulerStep(IntegratorStep):
    """Forward-Euler stepper (integrator_step.py:22-35): kick, then drift with
    the NEW velocity, then the density."""

    def stage1(self, d_idx, d_x, d_y, d_z, d_u, d_v, d_w, d_au, d_av, d_aw,
               d_rho, d_arho, dt):
        un = d_u[d_idx] + dt * d_au[d_idx]
        vn = d_v[d_idx] + dt * d_av[d_idx]
        wn = d_w[d_idx] + dt * d_aw[d_idx]
        d_u[d_idx] = un
        d_v[d_idx] = vn
        d_w[d_idx] = wn
        d_x[d_idx] += dt * un
        d_y[d_idx] += dt * vn
        d_z[d_idx] += dt * wn
        d_rho[d_idx] += dt * d_arho[d_idx]


class LeapFrogStep(IntegratorStep):
    """Stepper of `LeapFrogIntegrator` (integrator_step.py:708-730): positions
    move with u + ax (XSPH-corrected velocity), half a step either side of the
    kick."""

    def stage1(self, d_idx, d_x, d_y, d_z, d_u, d_v, d_w, d_ax, d_ay, d_az, dt):
        hdt = 0.5 * dt
        d_x[d_idx] += hdt * (d_u[d_idx] + d_ax[d_idx])
        d_y[d_idx] += hdt * (d_v[d_idx] + d_ay[d_idx])
        d_z[d_idx] += hdt * (d_w[d_idx] + d_az[d_idx])

    def stage2(self, d_idx, d_x, d_y, d_z, d_u, d_v, d_w, d_au, d_av, d_aw,
               d_ax, d_ay, d_az, d_rho, d_arho, d_e, d_ae, dt):
        hdt = 0.5 * dt
        # kick
        d_u[d_idx] += dt * d_au[d_idx]
        d_v[d_idx] += dt * d_av[d_idx]
        d_w[d_idx] += dt * d_aw[d_idx]
        d_rho[d_idx] += dt * d_arho[d_idx]
        d_e[d_idx] += dt * d_ae[d_idx]
        # second half drift with the kicked velocity
        d_x[d_idx] += hdt * (d_u[d_idx] + d_ax[d_idx])
        d_y[d_idx] += hdt * (d_v[d_idx] + d_ay[d_idx])
        d_z[d_idx] += hdt * (d_w[d_idx] + d_az[d_idx])


# Omelyan, Mryglod & Folk, Comput. Phys. Commun. 146 (2002) 188, eq. (20)
PEFRL_XI = 0.1786178958448091
PEFRL_LAMBDA = -0.2123418310626054
PEFRL_CHI = -0.06626458266981849


class PEFRLIntegrator(Integrator):
    """Position-extended Forest-Ruth-like, 4th order, four evaluations per step
    (integrator.py:481-517).  The post-stage times are the cumulative drift
    weights xi, xi + chi, 1 - (xi + chi), 1 - xi, 1."""
    SEQUENCE = (('stage', 1, PEFRL_XI), 'accel',
                ('stage', 2, PEFRL_XI + PEFRL_CHI), 'accel',
                ('stage', 3, 1.0 - (PEFRL_XI + PEFRL_CHI)), 'accel',
                ('stage', 4, 1.0 - PEFRL_XI), 'accel',
                ('stage', 5, 1.0))


class PEFRLStep(IntegratorStep):
    """Stepper of `PEFRLIntegrator` (integrator_step.py:738-830): drifts with
    weights (xi, chi, 1 - 2 (chi + xi), chi, xi), kicks in between with weights
    ((1 - 2 lambda) / 2, lambda, lambda, (1 - 2 lambda) / 2).  The weights are
    scalar attributes, frozen into the generated stage kernels at build time."""

    def __init__(self):
        self.xi = PEFRL_XI
        self.lam = PEFRL_LAMBDA
        self.chi = PEFRL_CHI
        self.kick_outer = 0.5 * (1.0 - 2.0 * PEFRL_LAMBDA)
        self.drift_mid = 1.0 - 2.0 * (PEFRL_CHI + PEFRL_XI)

    def stage1(self, d_idx, d_x, d_y, d_z, d_u, d_v, d_w, d_ax, d_ay, d_az, dt):
        wx = self.xi * dt
        d_x[d_idx] += wx * (d_u[d_idx] + d_ax[d_idx])
        d_y[d_idx] += wx * (d_v[d_idx] + d_ay[d_idx])
        d_z[d_idx] += wx * (d_w[d_idx] + d_az[d_idx])

    def stage2(self, d_idx, d_x, d_y, d_z, d_u, d_v, d_w, d_au, d_av, d_aw,
               d_ax, d_ay, d_az, d_rho, d_arho, d_e, d_ae, dt):
        wv = self.kick_outer * dt
        wx = self.chi * dt
        d_u[d_idx] += wv * d_au[d_idx]
        d_v[d_idx] += wv * d_av[d_idx]
        d_w[d_idx] += wv * d_aw[d_idx]
        d_rho[d_idx] += wv * d_arho[d_idx]
        d_e[d_idx] += wv * d_ae[d_idx]
        d_x[d_idx] += wx * (d_u[d_idx] + d_ax[d_idx])
        d_y[d_idx] += wx * (d_v[d_idx] + d_ay[d_idx])
        d_z[d_idx] += wx * (d_w[d_idx] + d_az[d_idx])

    def stage3(self, d_idx, d_x, d_y, d_z, d_u, d_v, d_w, d_au, d_av, d_aw,
               d_ax, d_ay, d_az, d_rho, d_arho, d_e, d_ae, dt):
        wv = self.lam * dt
        wx = self.drift_mid * dt
        d_u[d_idx] += wv * d_au[d_idx]
        d_v[d_idx] += wv * d_av[d_idx]
        d_w[d_idx] += wv * d_aw[d_idx]
        d_rho[d_idx] += wv * d_arho[d_idx]
        d_e[d_idx] += wv * d_ae[d_idx]
        d_x[d_idx] += wx * (d_u[d_idx] + d_ax[d_idx])
        d_y[d_idx] += wx * (d_v[d_idx] + d_ay[d_idx])
        d_z[d_idx] += wx * (d_w[d_idx] + d_az[d_idx])

    def stage4(self, d_idx, d_x, d_y, d_z, d_u, d_v, d_w, d_au, d_av, d_aw,
               d_ax, d_ay, d_az, d_rho, d_arho, d_e, d_ae, dt):
        wv = self.lam * dt
        wx = self.chi * dt
        d_u[d_idx] += wv * d_au[d_idx]
        d_v[d_idx] += wv * d_av[d_idx]
        d_w[d_idx] += wv * d_aw[d_idx]
        d_rho[d_idx] += wv * d_arho[d_idx]
        d_e[d_idx] += wv * d_ae[d_idx]
        d_x[d_idx] += wx * (d_u[d_idx] + d_ax[d_idx])
        d_y[d_idx] += wx * (d_v[d_idx] + d_ay[d_idx])
        d_z[d_idx] += wx * (d_w[d_idx] + d_az[d_idx])

    def stage5(self, d_idx, d_x, d_y, d_z, d_u, d_v, d_w, d_au, d_av, d_aw,
               d_ax, d_ay, d_az, d_rho, d_arho, d_e, d_ae, dt):
        wv = self.kick_outer * dt
        wx = self.xi * dt
        d_u[d_idx] += wv * d_au[d_idx]
        d_v[d_idx] += wv * d_av[d_idx]
        d_w[d_idx] += wv * d_aw[d_idx]
        d_rho[d_idx] += wv * d_arho[d_idx]
        d_e[d_idx] += wv * d_ae[d_idx]
        d_x[d_idx] += wx * (d_u[d_idx] + d_ax[d_idx])
        d_y[d_idx] += wx * (d_v[d_idx] + d_ay[d_idx])
        d_z[d_idx] += wx * (d_w[d_idx] + d_az[d_idx])
