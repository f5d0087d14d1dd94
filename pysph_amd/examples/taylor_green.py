"""2-D Taylor-Green vortex with the transport-velocity formulation -- problem
set-up and a device-resident time loop.

Parameters of ``pysph/examples/taylor_green.py:29-35,140-158,263-298``: unit
periodic box, U = 1, rho0 = 1, c0 = 10, p0 = pb = c0^2 rho0, Re = 100
(nu = U L / Re), nx = 50, hdx = 1, QuinticSpline, initial field
u = -U cos(2 pi x) sin(2 pi y), v = U sin(2 pi x) cos(2 pi y); ``TVFScheme``
integrated by PEC + ``TransportVelocityStep`` with periodic images rebuilt after
every stage (``Integrator.update_domain``).  The exact solution decays as
exp(-8 pi^2 nu t / L^2) (taylor_green.py:38-50), which the test uses.
"""
import numpy as np

from ..integrator import TransportVelocityStep
from ..kernels import QuinticSpline
from ..particle_array import get_particle_array_tvf_fluid
from ..scheme import TVFScheme

L, U, rho0 = 1.0, 1.0, 1.0
c0 = 10 * U
p0 = c0 ** 2 * rho0
hdx = 1.0


def create_particles(nx=50):
    dx = L / nx
    g = (np.arange(nx) + 0.5) * dx
    x, y = [a.ravel().copy() for a in np.meshgrid(g, g, indexing='ij')]
    pa = get_particle_array_tvf_fluid(
        name='fluid', x=x, y=y, h=hdx * dx * np.ones_like(x),
        m=rho0 * dx * dx * np.ones_like(x), rho=rho0 * np.ones_like(x),
        u=-U * np.cos(2 * np.pi * x) * np.sin(2 * np.pi * y),
        v=U * np.sin(2 * np.pi * x) * np.cos(2 * np.pi * y))
    pa.V[:] = 1.0 / (dx * dx)
    pa.uhat[:] = pa.u
    pa.vhat[:] = pa.v
    return [pa], dx


def run(nx=50, re=100.0, n_steps=200, ctx=None):
    """PEC steps, device-resident incl. the periodic images."""
    import time

    from .. import device as dev
    from ..acceleration_eval import AccelerationEval, SPHCompiler
    from ..domain import HipDomainManager
    from ..integrator import PECIntegrator, setup_integrator
    from ..nnps import HipNNPS
    ctx = ctx or dev.HipContext(0)
    arrays, dx = create_particles(nx)
    nu = U * L / re
    kernel = QuinticSpline(dim=2)
    eqs = TVFScheme(['fluid'], [], dim=2, rho0=rho0, c0=c0, nu=nu, p0=p0, pb=p0,
                    h0=hdx * dx).get_equations()
    h0 = hdx * dx
    dt = min(0.25 * h0 / (c0 + U), 0.125 * h0 * h0 / nu)      # taylor_green.py:166-171
    for a in arrays:
        dev.attach(a, ctx).push()
    a_eval = AccelerationEval(arrays, eqs, kernel)
    SPHCompiler(a_eval, ctx=ctx, sync='manual').compile()
    dom = HipDomainManager(ctx=ctx, xmin=0, xmax=L, ymin=0, ymax=L, periodic_in_x=True,
                           periodic_in_y=True)
    nnps = HipNNPS(2, arrays, radius_scale=kernel.radius_scale, ctx=ctx, sync=False,
                   domain=dom)
    a_eval.set_nnps(nnps)
    integ = PECIntegrator(fluid=TransportVelocityStep())
    setup_integrator(integ, a_eval, nnps)
    t, t0 = 0.0, time.perf_counter()
    for _ in range(n_steps):
        integ.step(t, dt)
        t += dt
    ctx.synchronize()
    wall = time.perf_counter() - t0
    arrays[0].gpu.sync_host()
    return arrays, dict(steps=n_steps, t=t, dt=dt, nu=nu, wall_s=wall,
                        steps_per_s=n_steps / wall)


if __name__ == '__main__':
    import argparse
    ap = argparse.ArgumentParser(description='2-D Taylor-Green vortex (TVF) on one MI355X')
    ap.add_argument('--nx', type=int, default=200)
    ap.add_argument('--steps', type=int, default=1000)
    args = ap.parse_args()
    arrs, st = run(nx=args.nx, n_steps=args.steps)
    pa = arrs[0]
    vmax = np.sqrt(pa.u ** 2 + pa.v ** 2).max()
    print('%d particles, %d steps to t = %.4f in %.2f s wall (%.0f steps/s); max|v| = %.4f, '
          'exact %.4f' % (pa.get_number_of_particles(True), st['steps'], st['t'], st['wall_s'],
                          st['steps_per_s'], vmax, U * np.exp(-8 * np.pi ** 2 * st['nu'] * st['t'])))
