"""Equation-set builders for the formulations the HIP backend accelerates.

Mirrors the *output* (group list, order, flags, parameters) of the
reference's ``WCSPHScheme.get_equations`` (pysph/sph/scheme.py:388-506) and
``TVFScheme.get_equations`` (:616-687) for the options the BASELINE configs
use.  Options that select equations without a hand-written kernel
(delta-SPH, laminar viscosity, update_h, TVF solid walls) raise instead of
silently dropping terms.  With the reference installed, its own ``Scheme``
objects can be used instead: the backend consumes the resulting group list.
"""
from .equations import (
    Group, TaitEOS, TaitEOSHGCorrection, ContinuityEquation, MomentumEquation,
    XSPHCorrection, SummationDensity, TVFSummationDensity, StateEquation,
    MomentumEquationPressureGradient, MomentumEquationViscosity,
    MomentumEquationArtificialViscosity, MomentumEquationArtificialStress)
from .particle_array import WCSPH_PROPS, TVF_FLUID_PROPS


def _ensure_props(particles, props):
    for pa in particles:
        for p in props:
            if p not in pa.properties:
                pa.add_property(p)


class WCSPHScheme(object):
    def __init__(self, fluids, solids, dim, rho0, c0, h0, hdx, gamma=7.0,
                 gx=0.0, gy=0.0, gz=0.0, alpha=0.1, beta=0.0, delta=0.1,
                 nu=0.0, tensile_correction=False, hg_correction=False,
                 update_h=False, delta_sph=False, summation_density=False):
        if delta_sph or update_h or abs(nu) > 1e-14:
            raise NotImplementedError(
                'HIP backend: delta_sph / update_h / nu!=0 select equations '
                'that have no hand-written kernel yet')
        self.fluids = list(fluids)
        self.solids = list(solids)
        self.dim = dim
        self.rho0 = rho0
        self.c0 = c0
        self.h0 = h0
        self.hdx = hdx
        self.gamma = gamma
        self.gx, self.gy, self.gz = gx, gy, gz
        self.alpha = alpha
        self.beta = beta
        self.tensile_correction = tensile_correction
        self.hg_correction = hg_correction
        self.summation_density = summation_density

    def get_timestep(self, cfl=0.5):
        return cfl * self.h0 / self.c0

    def get_equations(self):
        everyone = self.fluids + self.solids
        groups = []
        if self.summation_density:
            groups.append(Group(real=False, equations=[
                SummationDensity(dest=f, sources=everyone)
                for f in self.fluids]))
        eos = [TaitEOS(dest=f, sources=None, rho0=self.rho0, c0=self.c0,
                       gamma=self.gamma) for f in self.fluids]
        solid_eos = TaitEOSHGCorrection if self.hg_correction else TaitEOS
        eos += [solid_eos(dest=s, sources=None, rho0=self.rho0, c0=self.c0,
                          gamma=self.gamma) for s in self.solids]
        groups.append(Group(equations=eos, real=False))

        rates = [ContinuityEquation(dest=s, sources=self.fluids)
                 for s in self.solids]
        for f in self.fluids:
            if not self.summation_density:
                rates.append(ContinuityEquation(dest=f, sources=everyone))
            rates.append(MomentumEquation(
                dest=f, sources=everyone, c0=self.c0, alpha=self.alpha,
                beta=self.beta, gx=self.gx, gy=self.gy, gz=self.gz,
                tensile_correction=self.tensile_correction))
            rates.append(XSPHCorrection(dest=f, sources=[f]))
        groups.append(Group(equations=rates))
        return groups

    def setup_properties(self, particles, clean=True):
        _ensure_props(particles, WCSPH_PROPS)


class TVFScheme(object):
    def __init__(self, fluids, solids, dim, rho0, c0, nu, p0, pb, h0,
                 gx=0.0, gy=0.0, gz=0.0, alpha=0.0, tdamp=0.0,
                 wall_equations=None):
        """wall_equations: module (or object) providing ``SetWallVelocity``,
        ``SolidWallPressureBC`` and ``SolidWallNoSlipBC`` as Equation classes
        with Python bodies (they have no hand-written kernel and run through
        pysph_amd.codegen); default: ``pysph_amd.wall_bc``.  The reference's
        own ``pysph.sph.wc.transport_velocity`` works as well.  Only used when
        `solids` is given."""
        self.wall_equations = wall_equations
        self.fluids = list(fluids)
        self.solids = list(solids or [])
        self.dim = dim
        self.rho0, self.c0, self.nu, self.p0, self.pb, self.h0 = \
            rho0, c0, nu, p0, pb, h0
        self.gx, self.gy, self.gz = gx, gy, gz
        self.alpha = alpha
        self.tdamp = 0.0  # sic: scheme.py:545 ignores the argument

    def get_equations(self):
        """scheme.py:616-687.  The wall equations (``solids`` given) have no
        hand-written kernel: they run through the generated-family path."""
        SetWallVelocity = SolidWallNoSlipBC = SolidWallPressureBC = None
        if self.solids:
            we = self.wall_equations
            if we is None:
                from . import wall_bc as we
            SetWallVelocity = we.SetWallVelocity
            SolidWallNoSlipBC = we.SolidWallNoSlipBC
            SolidWallPressureBC = we.SolidWallPressureBC
        everyone = self.fluids + self.solids
        groups = [
            Group(real=False, equations=[
                TVFSummationDensity(dest=f, sources=everyone)
                for f in self.fluids]),
        ]
        g2 = [StateEquation(dest=f, sources=None, p0=self.p0, rho0=self.rho0,
                            b=1.0) for f in self.fluids]
        g2 += [SetWallVelocity(dest=s, sources=self.fluids) for s in self.solids]
        groups.append(Group(real=False, equations=g2))
        if self.solids:
            groups.append(Group(real=False, equations=[
                SolidWallPressureBC(dest=s, sources=self.fluids, b=1.0,
                                    rho0=self.rho0, p0=self.p0, gx=self.gx,
                                    gy=self.gy, gz=self.gz)
                for s in self.solids]))
        force = []
        for f in self.fluids:
            force.append(MomentumEquationPressureGradient(
                dest=f, sources=everyone, pb=self.pb, gx=self.gx,
                gy=self.gy, gz=self.gz, tdamp=self.tdamp))
            if self.alpha > 0.0:
                force.append(MomentumEquationArtificialViscosity(
                    dest=f, sources=everyone, c0=self.c0, alpha=self.alpha))
            if self.nu > 0.0:
                force.append(MomentumEquationViscosity(
                    dest=f, sources=self.fluids, nu=self.nu))
                if self.solids:
                    force.append(SolidWallNoSlipBC(
                        dest=f, sources=self.solids, nu=self.nu))
            force.append(MomentumEquationArtificialStress(
                dest=f, sources=self.fluids))
        groups.append(Group(equations=force))
        return groups

    def setup_properties(self, particles, clean=True):
        from .particle_array import TVF_SOLID_PROPS
        _ensure_props([p for p in particles if p.name in self.fluids],
                      TVF_FLUID_PROPS)
        _ensure_props([p for p in particles if p.name in self.solids],
                      TVF_SOLID_PROPS)
