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


def run(dx=0.04, n_steps=20, tf=None, cfl=0.3, reorder_freq=50, ctx=None,
        adaptive=True, log=None):
    """The time loop of ``Solver.solve`` (pysph/solver/solver.py:430-520) for
    this problem with everything device-resident: EPEC integrator with
    ``WCSPHStep`` for the fluid (boundary and obstacle have no stepper, as in
    ``WCSPHScheme.configure_solver`` scheme.py:360-386), adaptive time step
    from the device reductions (integrator.py:161-200), particles re-ordered
    into cell order every ``reorder_freq`` steps (solver.py:296-302).  One push
    before the loop, one pull after.  Returns (arrays, stats)."""
    import time

    from .. import device as dev
    from ..acceleration_eval import AccelerationEval, SPHCompiler
    from ..integrator import EPECIntegrator, WCSPHStep, setup_integrator
    from ..nnps import HipNNPS
    ctx = ctx or dev.HipContext(0)
    arrays = create_particles(dx)
    kernel = create_kernel()
    eqs = create_scheme(dx).get_equations()
    for a in arrays:
        dev.attach(a, ctx).push()
    a_eval = AccelerationEval(arrays, eqs, kernel)
    SPHCompiler(a_eval, ctx=ctx, sync='manual').compile()
    nnps = HipNNPS(dim, arrays, radius_scale=kernel.radius_scale, ctx=ctx,
                   sync=False)
    a_eval.set_nnps(nnps)
    integ = EPECIntegrator(fluid=WCSPHStep())
    setup_integrator(integ, a_eval, nnps)
    dt = 0.125 * hdx * dx / (1.1 * c0)        # dam_break_3d.py:33 (dt = 0.125 h0/(1.1 c0))
    t, step, t0 = 0.0, 0, time.perf_counter()
    dts = []
    while step < n_steps and (tf is None or t < tf):
        if reorder_freq and step % reorder_freq == 0:
            for i in range(len(arrays)):
                nnps.spatially_order_particles(i)
            nnps.update()
        integ.step(t, dt)
        t += dt
        step += 1
        dts.append(dt)
        if adaptive:
            new = integ.compute_time_step(dt, cfl)
            if new is not None:
                dt = new
        if log and step % log == 0:
            print('step %d  t = %.5f  dt = %.3e' % (step, t, dt))
    ctx.synchronize()
    wall = time.perf_counter() - t0
    for a in arrays:
        a.gpu.pull()
    n = sum(a.get_number_of_particles() for a in arrays)
    stats = dict(steps=step, t=t, wall_s=wall, steps_per_s=step / wall,
                 particles=n, particle_steps_per_s=n * step / wall, dts=dts)
    return arrays, stats


if __name__ == '__main__':
    import argparse
    ap = argparse.ArgumentParser(description='3-D dam break on one MI355X')
    ap.add_argument('--dx', type=float, default=0.02)
    ap.add_argument('--steps', type=int, default=200)
    args = ap.parse_args()
    _, st = run(dx=args.dx, n_steps=args.steps, log=50)
    print('%d particles, %d steps to t = %.4f s in %.2f s wall: %.1f steps/s, '
          '%.3g particle-steps/s' % (st['particles'], st['steps'], st['t'],
                                     st['wall_s'], st['steps_per_s'],
                                     st['particle_steps_per_s']))
