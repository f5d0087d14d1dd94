"""3-D dam break over a dry bed (SPHERIC test 2) -- problem set-up.

Inputs only: the particle-placement rule of the reference's
``DamBreak3DGeometry.create_particles`` (pysph/examples/_db_geometry.py:250-400)
re-expressed with boolean masks (the reference walks ``for i in range(x.size)``
in Python, unusable at 1e7 particles), and the parameters of
``pysph/examples/dam_break_3d.py:20-32,54-60``.
"""
import numpy as np

from ..kernels import WendlandQuintic
from ..particle_array import get_particle_array_wcsph
from ..scheme import WCSPHScheme

dim = 3
nboundary_layers = 1
hdx = 1.3
ro = 1000.0
gamma = 7.0
alpha = 0.25
beta = 0.0
c0 = 10.0 * np.sqrt(2.0 * 9.81 * 0.55)


class DamBreak3DGeometry(object):
    def __init__(self, container_height=1.0, container_width=1.0,
                 container_length=3.22, fluid_column_height=0.55,
                 fluid_column_width=1.0, fluid_column_length=1.228,
                 obstacle_center_x=2.5, obstacle_center_y=0,
                 obstacle_length=0.16, obstacle_height=0.161,
                 obstacle_width=0.4, nboundary_layers=5, with_obstacle=True,
                 dx=0.02, hdx=1.2, rho0=1000.0):
        self.__dict__.update(locals())
        del self.__dict__['self']

    def get_max_speed(self, g=9.81):
        return np.sqrt(2 * g * self.fluid_column_height)

    def create_particles(self):
        dx = self.dx
        ghost = self.nboundary_layers * dx
        cw2 = 0.5 * self.container_width
        eps = 0.1 * dx
        # lattice: same mgrid limits as _db_geometry.py:303-316
        gx = np.mgrid[0.0 - ghost:self.container_length + ghost + eps:dx]
        gy = np.mgrid[-cw2 - ghost:cw2 + ghost + eps:dx]
        gz = np.mgrid[0.0 - ghost:self.container_height + ghost + eps:dx]
        x, y, z = [a.ravel() for a in np.meshgrid(gx, gy, gz, indexing='ij')]

        fluid = ((0 < x) & (x <= self.fluid_column_length) &
                 (-cw2 < y) & (y < cw2) &
                 (0 < z) & (z <= self.fluid_column_height))
        obl2, obw2 = 0.5 * self.obstacle_length, 0.5 * self.obstacle_width
        ocx, ocy = self.obstacle_center_x, self.obstacle_center_y
        obstacle = ((ocx - obl2 <= x) & (x <= ocx + obl2) &
                    (ocy - obw2 <= y) & (y <= ocy + obw2) &
                    (0 < z) & (z <= self.obstacle_height))
        wall = ((y <= -cw2) | (y >= cw2) | (x >= self.container_length) |
                (x <= 0) | (z <= 0))

        h0 = self.hdx * dx
        m0 = self.rho0 * dx ** 3
        out = []
        sets = [('fluid', fluid), ('boundary', wall)]
        if self.with_obstacle:
            sets.append(('obstacle', obstacle))
        for name, mask in sets:
            idx = np.nonzero(mask)[0]
            pa = get_particle_array_wcsph(name=name, x=x[idx], y=y[idx],
                                          z=z[idx])
            pa.m[:] = m0
            pa.h[:] = h0
            pa.rho[:] = self.rho0
            out.append(pa)
        return out


def create_scheme(dx=0.02, hdx=hdx):
    """dam_break_3d.py:54-60."""
    return WCSPHScheme(['fluid'], ['boundary', 'obstacle'], dim=dim, rho0=ro,
                       c0=c0, h0=dx * hdx, hdx=hdx, gz=-9.81, alpha=alpha,
                       beta=beta, gamma=gamma, hg_correction=True,
                       tensile_correction=False)


def create_kernel():
    return WendlandQuintic(dim=dim)


def create_particles(dx=0.02, hdx=hdx):
    geom = DamBreak3DGeometry(dx=dx, nboundary_layers=nboundary_layers,
                              hdx=hdx, rho0=ro)
    return geom.create_particles()
