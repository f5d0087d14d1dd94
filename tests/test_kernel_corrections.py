"""The gradient-correction scenarios of the reference's
pysph/sph/tests/test_kernel_corrections.py (:63-131, :254-275) on the HIP
backend: four (2-D) or eight (3-D) particles on the corners of a box carrying
u = x + y (+ z); after SummationDensity -> correction matrix -> corrected DWIJ,
the SPH gradient of u is exactly (1, 1[, 1]) -- also for perturbed positions.

The correction equations are the restatements in tests/custom_equations.py
(their reference counterparts, GradientCorrectionPreStep / GradientCorrection of
wc/kernel_correction.py, translate too: tests/reference_census.py); GradPhi is
the reference test's own probe equation.  Every group runs as a generated
family, through SPHEvaluator like in the reference's test."""
import numpy as np
import pytest

from pysph_amd.equations import Equation, Group, SummationDensity


class GradPhi(Equation):                            # test_kernel_corrections.py:20-30
    def initialize(self, d_idx, d_gradu):
        d_gradu[3 * d_idx] = 0.0
        d_gradu[3 * d_idx + 1] = 0.0
        d_gradu[3 * d_idx + 2] = 0.0

    def loop(self, d_idx, d_gradu, d_u, s_idx, s_m, s_rho, s_u, DWIJ):
        fac = s_m[s_idx] / s_rho[s_idx] * (s_u[s_idx] - d_u[d_idx])
        d_gradu[3 * d_idx] += fac * DWIJ[0]
        d_gradu[3 * d_idx + 1] += fac * DWIJ[1]
        d_gradu[3 * d_idx + 2] += fac * DWIJ[2]


def corner_particles(dim, perturbed=False):
    from pysph_amd.particle_array import get_particle_array
    if dim == 2:
        x, y = [a.ravel() for a in np.mgrid[0.5:1:2j, 0.5:1:2j]]
        z = np.zeros_like(x)
    else:
        x, y, z = [a.ravel() for a in np.mgrid[0.5:1:2j, 0.5:1:2j, 0.5:1:2j]]
    if perturbed:                                   # :94-98
        d = np.resize([0.1, 0.05, -0.1, -0.05], x.size)
        x, y = x + d, y + d
    u = x + y + (z if dim == 3 else 0.0)
    pa = get_particle_array(name='fluid', x=x, y=y, z=z, h=0.5 * np.ones_like(x),
                            m=np.ones_like(x), u=u)
    pa.add_property('gradu', stride=3)
    pa.add_property('lmat', stride=9)
    return pa


def correction_equations(dim):
    from custom_equations import CorrectGradient, CorrectionMatrix
    return [Group(equations=[SummationDensity(dest='fluid', sources=['fluid'])]),
            Group(equations=[CorrectionMatrix(dest='fluid', sources=['fluid'], dim=dim)]),
            Group(equations=[CorrectGradient(dest='fluid', sources=['fluid'], dim=dim, tol=100.0),
                             GradPhi(dest='fluid', sources=['fluid'])])]


def expected(dim, n):
    e = np.ones((n, 3))
    if dim == 2:
        e[:, 2] = 0.0
    return e.ravel()


@pytest.mark.parametrize('dim', [2, 3])
@pytest.mark.parametrize('perturbed', [False, True])
def test_gradient_correction_python_evaluator(oracle, dim, perturbed):
    """the restated equations give the reference test's expected values when
    run as plain Python (CPU)"""
    from oracle.py_eval import PyEval
    from pysph_amd.kernels import CubicSpline
    pa = corner_particles(dim, perturbed)
    kernel = CubicSpline(dim=dim)
    onn = oracle.OracleNNPS(dim, [pa], radius_scale=kernel.radius_scale)
    onn.update()
    eqs = correction_equations(dim)
    oev = oracle.OracleEval([pa], eqs[:1], kernel)          # SummationDensity: C oracle
    oev.set_nnps(onn)
    oev.compute(0.0, 0.1)
    PyEval([pa], eqs[1:], kernel, onn).compute(0.0, 0.1)
    np.testing.assert_array_almost_equal(pa.gradu, expected(dim, pa.x.size))


@pytest.mark.gpu
@pytest.mark.parametrize('dim', [2, 3])
@pytest.mark.parametrize('perturbed', [False, True])
def test_gradient_correction(dim, perturbed):       # :101-131
    from pysph_amd.kernels import CubicSpline
    from pysph_amd.tools import SPHEvaluator
    pa = corner_particles(dim, perturbed)
    seval = SPHEvaluator(arrays=[pa], equations=correction_equations(dim), dim=dim,
                         kernel=CubicSpline(dim=dim))
    seval.evaluate(0.0, 0.1)
    np.testing.assert_array_almost_equal(pa.gradu, expected(dim, pa.x.size))
