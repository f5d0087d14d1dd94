"""Elastic-solid (Gray, Monaghan & Swift 2001) equation specifications.

Mirrors pysph/sph/solid_mech/basic.py: ``get_particle_array_elastic_dynamics``
(:34-84), ``IsothermalEOS`` (:93-101, NOT the basic_equations one: it reads the
array constants ``c0_ref``/``rho_ref``), ``MonaghanArtificialStress`` (:104-242),
``MomentumEquationWithStress`` (:245-387), ``HookesDeviatoricStressRate``
(:390-505), ``ElasticSolidsScheme`` (:592-651), plus ``VelocityGradient2D/3D``
(pysph/sph/basic_equations.py:63-148) and ``SolidMechStep``
(pysph/sph/integrator_step.py:173-255).  Specification objects only; the bodies
are HIP device code.  Array constants (``G, wdeltap, n, c0_ref, rho_ref``) are
read from the destination array at every ``compute`` (the reference indexes
``d_G[0]`` etc. at run time).
"""
import numpy as np

from .equations import (Equation, Group, ContinuityEquation,
                        MonaghanArtificialViscosity, XSPHCorrection,
                        EQUATION_TABLE, NO_SOURCE_KINDS)
from .integrator import IntegratorStep, _STEP_KINDS
from .particle_array import get_particle_array


def get_bulk_mod(G, nu):
    return 2.0 * G * (1 + nu) / (3 * (1 - 2 * nu))


def get_speed_of_sound(E, nu, rho0):
    return np.sqrt(E / (3 * (1. - 2 * nu) * rho0))


def get_shear_modulus(E, nu):
    return E / (2. * (1. + nu))


SOLID_PROPS = [
    'cs', 'e', 'v00', 'v01', 'v02', 'v10', 'v11', 'v12', 'v20', 'v21', 'v22',
    'r00', 'r01', 'r02', 'r11', 'r12', 'r22', 's00', 's01', 's02', 's11',
    's12', 's22', 'as00', 'as01', 'as02', 'as11', 'as12', 'as22', 's000',
    's010', 's020', 's110', 's120', 's220', 'arho', 'au', 'av', 'aw', 'ax',
    'ay', 'az', 'ae', 'rho0', 'u0', 'v0', 'w0', 'x0', 'y0', 'z0', 'e0']


def get_particle_array_elastic_dynamics(constants=None, **props):
    """solid_mech/basic.py:34-84."""
    consts = {'wdeltap': -1., 'n': 4, 'G': 0.0, 'E': 0.0, 'nu': 0.0,
              'rho_ref': 1000.0, 'c0_ref': 0.0}
    if constants:
        consts.update(constants)
    pa = get_particle_array(constants=consts, additional_props=SOLID_PROPS,
                            **props)
    pa.G[0] = get_shear_modulus(pa.E[0], pa.nu[0])
    pa.cs[:] = get_speed_of_sound(pa.E[0], pa.nu[0], pa.rho_ref[0])
    pa.c0_ref[0] = get_speed_of_sound(pa.E[0], pa.nu[0], pa.rho_ref[0])
    pa.set_output_arrays(['x', 'y', 'z', 'u', 'v', 'w', 'rho', 'm', 'h', 'pid',
                          'gid', 'tag', 'p'])
    return pa


class IsothermalEOS(Equation):
    """solid_mech/basic.py:93-101."""


class MonaghanArtificialStress(Equation):
    """solid_mech/basic.py:104-242."""

    def __init__(self, dest, sources, eps=0.3):
        self.eps = eps
        super(MonaghanArtificialStress, self).__init__(dest, sources)


class MomentumEquationWithStress(Equation):
    """solid_mech/basic.py:245-387."""


class HookesDeviatoricStressRate(Equation):
    """solid_mech/basic.py:390-505."""


class VelocityGradient2D(Equation):
    """basic_equations.py:63-98."""


class VelocityGradient3D(Equation):
    """basic_equations.py:101-148."""


class SolidMechStep(IntegratorStep):
    """integrator_step.py:173-255."""
    _kind = 3


_STEP_KINDS['SolidMechStep'] = 3

EQ_VELOCITY_GRADIENT_3D = 15
EQ_VELOCITY_GRADIENT_2D = 16
EQ_HOOKES_DEVIATORIC_STRESS_RATE = 17
EQ_MOMENTUM_WITH_STRESS = 18
EQ_MONAGHAN_ART_STRESS = 19
EQ_SOLID_ISOTHERMAL_EOS = 21

_XV = ('x', 'y', 'z', 'h')
_S6 = ('s00', 's01', 's02', 's11', 's12', 's22')
_R6 = ('r00', 'r01', 'r02', 'r11', 'r12', 'r22')
_V9 = ('v00', 'v01', 'v02', 'v10', 'v11', 'v12', 'v20', 'v21', 'v22')
# '@name': first value of the destination array's constant `name`
EQUATION_TABLE.update({
    'VelocityGradient3D': (EQ_VELOCITY_GRADIENT_3D, (),
                           _XV + ('u', 'v', 'w') + _V9,
                           _XV + ('u', 'v', 'w', 'm', 'rho')),
    'VelocityGradient2D': (EQ_VELOCITY_GRADIENT_2D, (),
                           _XV + ('u', 'v', 'w', 'v00', 'v01', 'v10', 'v11'),
                           _XV + ('u', 'v', 'w', 'm', 'rho')),
    'HookesDeviatoricStressRate': (
        EQ_HOOKES_DEVIATORIC_STRESS_RATE, ('@G',),
        _S6 + _V9 + ('as00', 'as01', 'as02', 'as11', 'as12', 'as22', 'G'), ()),
    'MomentumEquationWithStress': (
        EQ_MOMENTUM_WITH_STRESS, ('@wdeltap', '@n'),
        _XV + ('rho', 'p', 'au', 'av', 'aw', 'wdeltap', 'n') + _S6 + _R6,
        _XV + ('rho', 'p', 'm') + _S6 + _R6),
    'MonaghanArtificialStress': (EQ_MONAGHAN_ART_STRESS, ('eps',),
                                 ('rho', 'p') + _S6 + _R6, ()),
    'SolidIsothermalEOS': (EQ_SOLID_ISOTHERMAL_EOS, ('@c0_ref', '@rho_ref'),
                           ('rho', 'p', 'c0_ref', 'rho_ref'), ()),
})


class ElasticSolidsScheme(object):
    """solid_mech/basic.py:592-651.  `dim` selects VelocityGradient2D (what the
    reference scheme hard-codes) or VelocityGradient3D (needed for 3-D runs)."""

    def __init__(self, elastic_solids, solids, dim, artificial_stress_eps=0.3,
                 xsph_eps=0.5, alpha=1.0, beta=1.0, ghost_recompute=False):
        """ghost_recompute: for slab-decomposed (multi-GPU) runs.  The second
        group reads p and the artificial stress r_ij of its SOURCES, ghosts
        included; both are per-particle functions of rho and s_ij, which the
        ghost exchange carries (parallel.ELASTIC_HALO_PROPS).  With this flag the
        two no-source equations that compute them form a group of their own over
        ALL particles (real=False, like the EOS groups of the fluid schemes)
        instead of the reference's mid-evaluation refresh of remote properties
        (ParallelManager.update_remote_particle_properties,
        pysph/parallel/parallel_manager.pyx:159-210): same values on the real
        particles, one exchange per evaluation."""
        self.ghost_recompute = ghost_recompute
        self.elastic_solids = list(elastic_solids)
        self.solids = list(solids)
        self.dim = dim
        self.alpha = alpha
        self.beta = beta
        self.xsph_eps = xsph_eps
        self.artificial_stress_eps = artificial_stress_eps

    def get_equations(self):
        everyone = self.solids + self.elastic_solids
        VG = VelocityGradient3D if self.dim == 3 else VelocityGradient2D
        g0, g1, g2 = [], [], []
        for es in self.elastic_solids:
            # equations without sources run before the pair loops of their group
            # (mako :50-58): hoisting these two in front of VG changes nothing
            pre = g0 if self.ghost_recompute else g1
            pre.append(IsothermalEOS(es, sources=None))
            g1.append(VG(dest=es, sources=everyone))
            pre.append(MonaghanArtificialStress(
                dest=es, sources=None, eps=self.artificial_stress_eps))
        for es in self.elastic_solids:
            g2.append(ContinuityEquation(dest=es, sources=everyone))
            g2.append(MomentumEquationWithStress(dest=es, sources=everyone))
            g2.append(MonaghanArtificialViscosity(
                dest=es, sources=everyone, alpha=self.alpha, beta=self.beta))
            g2.append(HookesDeviatoricStressRate(dest=es, sources=None))
            g2.append(XSPHCorrection(dest=es, sources=[es], eps=self.xsph_eps))
        if self.ghost_recompute:
            return [Group(equations=g0, real=False), Group(equations=g1), Group(equations=g2)]
        return [Group(equations=g1), Group(equations=g2)]

    def setup_properties(self, particles, clean=True):
        for pa in particles:
            for p in SOLID_PROPS:
                if p not in pa.properties:
                    pa.add_property(p)
