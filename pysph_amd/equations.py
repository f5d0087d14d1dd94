"""Equation / Group specification objects for the HIP backend.

In the reference an ``Equation`` carries Python method bodies that ``compyle``
transpiles (pysph/sph/equation.py:392-446); here the bodies are hand-written
HIP device code, so these classes are *specifications only*: same class names,
constructor arguments, defaults and derived attributes as the reference
classes they stand for, so user scripts and the reference's ``Scheme`` classes
read identically.  The backend matches equations by **class name** (see
``EQUATION_TABLE``), which is why a real ``pysph`` equation object works at
the same boundary.

``EQUATION_TABLE[name]`` = (kind id of the C-ABI, parameter names in C-ABI
order, destination properties touched, source properties read).
"""
from collections import OrderedDict
import itertools

_group_counter = itertools.count()


class Equation(object):
    """pysph/sph/equation.py:392-446."""

    def __init__(self, dest, sources):
        self.dest = dest
        self.sources = sources if sources else None
        self.no_source = self.sources is None
        self.name = self.__class__.__name__
        self.var_name = ''

    def converged(self):
        return 1.0

    def __repr__(self):
        args = ', '.join('%s=%r' % kv for kv in sorted(self.__dict__.items())
                         if kv[0] not in ('name', 'var_name', 'no_source'))
        return '%s(%s)' % (self.name, args)


class Group(object):
    """pysph/sph/equation.py:448-561 (constructor surface and flags)."""

    def __init__(self, equations, real=True, update_nnps=False, iterate=False,
                 max_iterations=1, min_iterations=0, pre=None, post=None,
                 condition=None, start_idx=0, stop_idx=None, name=None):
        self.equations = equations
        self.real = real
        self.update_nnps = update_nnps
        self.iterate = iterate
        self.max_iterations = max_iterations
        self.min_iterations = min_iterations
        self.pre = pre
        self.post = post
        self.condition = condition
        self.start_idx = start_idx
        self.stop_idx = stop_idx
        self.name = name if name is not None else 'Group_%d' % next(_group_counter)
        n_sub = sum(isinstance(e, Group) for e in equations)
        if n_sub and n_sub != len(equations):
            raise ValueError('All elements must be Groups if you use sub groups.')
        self.has_subgroups = n_sub > 0


class MultiStageEquations(object):
    """pysph/sph/equation.py:966-1004."""

    def __init__(self, groups):
        self.groups = groups


# --------------------------------------------------------------------------
# WCSPH  (pysph/sph/wc/basic.py, pysph/sph/basic_equations.py)
# --------------------------------------------------------------------------
class TaitEOS(Equation):
    """wc/basic.py:9-65."""

    def __init__(self, dest, sources, rho0, c0, gamma, p0=0.0):
        self.rho0 = rho0
        self.rho01 = 1.0 / rho0
        self.c0 = c0
        self.gamma = gamma
        self.gamma1 = 0.5 * (gamma - 1.0)
        self.B = rho0 * c0 * c0 / gamma
        self.p0 = p0
        super(TaitEOS, self).__init__(dest, sources)


class TaitEOSHGCorrection(Equation):
    """wc/basic.py:68-126."""

    def __init__(self, dest, sources, rho0, c0, gamma):
        self.rho0 = rho0
        self.rho01 = 1.0 / rho0
        self.c0 = c0
        self.gamma = gamma
        self.gamma1 = 0.5 * (gamma - 1.0)
        self.B = rho0 * c0 * c0 / gamma
        super(TaitEOSHGCorrection, self).__init__(dest, sources)


class ContinuityEquation(Equation):
    """basic_equations.py:177-192."""


class MomentumEquation(Equation):
    """wc/basic.py:129-271."""

    def __init__(self, dest, sources, c0, alpha=1.0, beta=1.0, gx=0.0, gy=0.0,
                 gz=0.0, tensile_correction=False):
        self.alpha = alpha
        self.beta = beta
        self.gx = gx
        self.gy = gy
        self.gz = gz
        self.c0 = c0
        self.tensile_correction = tensile_correction
        super(MomentumEquation, self).__init__(dest, sources)


class XSPHCorrection(Equation):
    """basic_equations.py:260-300."""

    def __init__(self, dest, sources, eps=0.5):
        self.eps = eps
        super(XSPHCorrection, self).__init__(dest, sources)


class SummationDensity(Equation):
    """basic_equations.py:19-29."""


class IsothermalEOS(Equation):
    """basic_equations.py:151-176."""

    def __init__(self, dest, sources, rho0, c0, p0):
        self.rho0 = rho0
        self.c0 = c0
        self.c02 = c0 * c0
        self.p0 = p0
        super(IsothermalEOS, self).__init__(dest, sources)


class MonaghanArtificialViscosity(Equation):
    """basic_equations.py:195-257."""

    def __init__(self, dest, sources, alpha=1.0, beta=1.0):
        self.alpha = alpha
        self.beta = beta
        super(MonaghanArtificialViscosity, self).__init__(dest, sources)


# --------------------------------------------------------------------------
# Transport-velocity formulation (pysph/sph/wc/transport_velocity.py)
# --------------------------------------------------------------------------
class TVFSummationDensity(Equation):
    """transport_velocity.py:24-58 (class ``SummationDensity`` there; the
    backend tells the two apart by the 'V' property -- see
    ``resolve_equation``)."""

    def __init__(self, dest, sources):
        super(TVFSummationDensity, self).__init__(dest, sources)
        self.name = 'SummationDensity'


class StateEquation(Equation):
    """transport_velocity.py:190-216."""

    def __init__(self, dest, sources, p0, rho0, b=1.0):
        self.b = b
        self.p0 = p0
        self.rho0 = rho0
        super(StateEquation, self).__init__(dest, sources)


class MomentumEquationPressureGradient(Equation):
    """transport_velocity.py:219-325."""

    def __init__(self, dest, sources, pb, gx=0., gy=0., gz=0., tdamp=0.0):
        self.pb = pb
        self.gx = gx
        self.gy = gy
        self.gz = gz
        self.tdamp = tdamp
        super(MomentumEquationPressureGradient, self).__init__(dest, sources)


class MomentumEquationViscosity(Equation):
    """transport_velocity.py:328-386."""

    def __init__(self, dest, sources, nu):
        self.nu = nu
        super(MomentumEquationViscosity, self).__init__(dest, sources)


class MomentumEquationArtificialViscosity(Equation):
    """transport_velocity.py:389-436."""

    def __init__(self, dest, sources, c0, alpha=0.1):
        self.alpha = alpha
        self.c0 = c0
        super(MomentumEquationArtificialViscosity, self).__init__(dest, sources)


class MomentumEquationArtificialStress(Equation):
    """transport_velocity.py:439-545."""


# --------------------------------------------------------------------------
# kind ids (must match include/sphhip.h) and per-equation metadata
# --------------------------------------------------------------------------
EQ_TAIT_EOS = 1
EQ_TAIT_EOS_HG = 2
EQ_CONTINUITY = 3
EQ_MOMENTUM = 4
EQ_XSPH = 5
EQ_SUMMATION_DENSITY = 6
EQ_TVF_SUMMATION_DENSITY = 7
EQ_TVF_STATE_EQUATION = 8
EQ_TVF_MOM_PRESSURE = 9
EQ_TVF_MOM_VISCOSITY = 10
EQ_TVF_MOM_ART_VISCOSITY = 11
EQ_TVF_MOM_ART_STRESS = 12
EQ_ISOTHERMAL_EOS = 13
EQ_MONAGHAN_ART_VISCOSITY = 14

_XV = ('x', 'y', 'z', 'h')
EQUATION_TABLE = OrderedDict([
    # name: (kind, params, dest props, source props)
    ('TaitEOS', (EQ_TAIT_EOS, ('rho0', 'c0', 'gamma', 'p0'),
                 ('rho', 'p', 'cs'), ())),
    ('TaitEOSHGCorrection', (EQ_TAIT_EOS_HG, ('rho0', 'c0', 'gamma'),
                             ('rho', 'p', 'cs'), ())),
    ('ContinuityEquation', (EQ_CONTINUITY, (),
                            _XV + ('u', 'v', 'w', 'arho'),
                            _XV + ('u', 'v', 'w', 'm'))),
    ('MomentumEquation', (EQ_MOMENTUM,
                          ('c0', 'alpha', 'beta', 'gx', 'gy', 'gz',
                           'tensile_correction'),
                          _XV + ('u', 'v', 'w', 'rho', 'p', 'cs', 'au', 'av',
                                 'aw', 'dt_cfl', 'dt_force'),
                          _XV + ('u', 'v', 'w', 'rho', 'p', 'cs', 'm'))),
    ('XSPHCorrection', (EQ_XSPH, ('eps',),
                        _XV + ('u', 'v', 'w', 'rho', 'ax', 'ay', 'az'),
                        _XV + ('u', 'v', 'w', 'rho', 'm'))),
    ('SummationDensity', (EQ_SUMMATION_DENSITY, (),
                          _XV + ('rho',), _XV + ('m',))),
    ('TVFSummationDensity', (EQ_TVF_SUMMATION_DENSITY, (),
                             _XV + ('rho', 'V', 'm'), _XV)),
    ('StateEquation', (EQ_TVF_STATE_EQUATION, ('p0', 'rho0', 'b'),
                       ('rho', 'p'), ())),
    ('MomentumEquationPressureGradient', (
        EQ_TVF_MOM_PRESSURE, ('pb', 'gx', 'gy', 'gz', 'tdamp'),
        _XV + ('m', 'rho', 'p', 'V', 'au', 'av', 'aw', 'auhat', 'avhat',
               'awhat'),
        _XV + ('rho', 'p', 'V'))),
    ('MomentumEquationViscosity', (
        EQ_TVF_MOM_VISCOSITY, ('nu',),
        _XV + ('u', 'v', 'w', 'm', 'rho', 'V', 'au', 'av', 'aw'),
        _XV + ('u', 'v', 'w', 'rho', 'V'))),
    ('MomentumEquationArtificialViscosity', (
        EQ_TVF_MOM_ART_VISCOSITY, ('c0', 'alpha'),
        _XV + ('u', 'v', 'w', 'rho', 'au', 'av', 'aw'),
        _XV + ('u', 'v', 'w', 'rho', 'm'))),
    ('MomentumEquationArtificialStress', (
        EQ_TVF_MOM_ART_STRESS, (),
        _XV + ('u', 'v', 'w', 'uhat', 'vhat', 'what', 'rho', 'V', 'm', 'au',
               'av', 'aw'),
        _XV + ('u', 'v', 'w', 'uhat', 'vhat', 'what', 'rho', 'V'))),
    ('IsothermalEOS', (EQ_ISOTHERMAL_EOS, ('rho0', 'c0', 'p0'),
                       ('rho', 'p'), ())),
    ('MonaghanArtificialViscosity', (
        EQ_MONAGHAN_ART_VISCOSITY, ('alpha', 'beta'),
        _XV + ('u', 'v', 'w', 'rho', 'cs', 'au', 'av', 'aw'),
        _XV + ('u', 'v', 'w', 'rho', 'cs', 'm'))),
])

# equations that have no neighbour loop
NO_SOURCE_KINDS = (EQ_TAIT_EOS, EQ_TAIT_EOS_HG, EQ_TVF_STATE_EQUATION,
                   EQ_ISOTHERMAL_EOS, 17, 19, 21)


def resolve_equation(eq):
    """(kind, [param values], dest props, src props) for an equation object
    (this module's or the reference's, matched by class name).  Unknown
    equations fail loudly -- there is no generic fallback."""
    name = type(eq).__name__
    if name == 'SummationDensity' and 'transport_velocity' in type(eq).__module__:
        name = 'TVFSummationDensity'
    if name == 'IsothermalEOS' and 'solid_mech' in type(eq).__module__:
        name = 'SolidIsothermalEOS'
    if name not in EQUATION_TABLE:
        try:   # the elastic family registers itself on import
            from . import solid_mech  # noqa: F401
        except ImportError:
            pass
    if name not in EQUATION_TABLE:
        raise NotImplementedError(
            'HIP backend: equation %s has no hand-written kernel; supported: %s'
            % (name, ', '.join(EQUATION_TABLE)))
    kind, params, dprops, sprops = EQUATION_TABLE[name]
    # '@name' parameters are array constants resolved at compute time
    vals = [p if p.startswith('@') else float(getattr(eq, p)) for p in params]
    return kind, vals, dprops, sprops
