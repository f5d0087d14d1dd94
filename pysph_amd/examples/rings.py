"""Colliding elastic rings (Gray, Monaghan & Swift 2001) -- problem set-up and
a device-resident time loop.

Inputs and parameters follow ``pysph/examples/solid_mech/rings.py:21-91``:
two rings (inner radius 0.03, outer 0.04) whose centres are 2 x 0.041 apart,
E = 1e7, nu = 0.3975, rho0 = 1, CubicSpline, hdx 1.5, approaching each other
with u = +-0.059 cs; ``ElasticSolidsScheme`` (velocity gradient, artificial
stress, momentum with stress, Hooke's deviatoric rate, artificial viscosity,
XSPH) integrated by PEC + ``SolidMechStep``.  The reference runs it in fp64 on
its Cython backend; so does this (fp32 is not built).
"""
import numpy as np

from ..kernels import CubicSpline
from ..solid_mech import (ElasticSolidsScheme, SolidMechStep,
                          get_particle_array_elastic_dynamics)

E, nu, rho0 = 1e7, 0.3975, 1.0
hdx = 1.5
ri, ro, spacing = 0.03, 0.04, 0.041
u_f = 0.059


def create_particles(dx=0.0005):
    x, y = np.mgrid[-ro:ro:dx, -ro:ro:dx]
    x, y = x.ravel(), y.ravel()
    d = x * x + y * y
    keep = np.flatnonzero((ri * ri <= d) * (d < ro * ro))
    x, y = x[keep], y[keep]
    x = np.concatenate([x - spacing, x + spacing])
    y = np.concatenate([y, y])
    h = hdx * dx
    kernel = CubicSpline(dim=2)
    pa = get_particle_array_elastic_dynamics(
        name='solid', x=x + spacing, y=y, m=np.ones_like(x) * dx * dx,
        rho=np.ones_like(x), h=np.ones_like(x) * h,
        constants=dict(wdeltap=kernel.kernel(rij=dx, h=h), n=4, rho_ref=rho0,
                       E=E, nu=nu))
    pa.u[:] = pa.cs * u_f * (2 * (x < 0) - 1)
    return [pa]


def create_scheme():
    return ElasticSolidsScheme(elastic_solids=['solid'], solids=[], dim=2)


def run(dx=0.0005, n_steps=200, dt=1e-8, ctx=None, reorder_freq=100):
    """PEC steps with everything device-resident (one push, one pull)."""
    import time

    from .. import device as dev
    from ..acceleration_eval import AccelerationEval, SPHCompiler
    from ..integrator import PECIntegrator, setup_integrator
    from ..nnps import HipNNPS
    ctx = ctx or dev.HipContext(0)
    arrays = create_particles(dx)
    kernel = CubicSpline(dim=2)
    eqs = create_scheme().get_equations()
    for a in arrays:
        dev.attach(a, ctx).push()
    a_eval = AccelerationEval(arrays, eqs, kernel)
    SPHCompiler(a_eval, ctx=ctx, sync='manual').compile()
    nnps = HipNNPS(2, arrays, radius_scale=kernel.radius_scale, ctx=ctx, sync=False)
    a_eval.set_nnps(nnps)
    integ = PECIntegrator(solid=SolidMechStep())
    setup_integrator(integ, a_eval, nnps)
    t, t0 = 0.0, time.perf_counter()
    for step in range(n_steps):
        if reorder_freq and step % reorder_freq == 0:
            nnps.spatially_order_particles(0)
            nnps.update()
        integ.step(t, dt)
        t += dt
    ctx.synchronize()
    wall = time.perf_counter() - t0
    arrays[0].gpu.pull()
    n = arrays[0].get_number_of_particles()
    return arrays, dict(steps=n_steps, t=t, wall_s=wall, steps_per_s=n_steps / wall,
                        particles=n)


if __name__ == '__main__':
    import argparse
    ap = argparse.ArgumentParser(description='colliding elastic rings on one MI355X')
    ap.add_argument('--dx', type=float, default=0.0005)
    ap.add_argument('--steps', type=int, default=1000)
    args = ap.parse_args()
    arrs, st = run(dx=args.dx, n_steps=args.steps)
    print('%d particles, %d steps in %.2f s wall: %.1f steps/s' % (
        st['particles'], st['steps'], st['wall_s'], st['steps_per_s']))
