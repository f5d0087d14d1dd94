"""SPH smoothing-kernel descriptors for the HIP backend.

Mirror of the reference's kernel classes (pysph/base/kernels.py): same class
names, constructor argument (``dim``) and attributes (``dim``, ``fac``,
``radius_scale``, ``get_deltap()``) -- which is all that crosses the drop-in
boundary (``AccelerationEval.kernel`` is read by value at compile time,
acceleration_eval_cython_helper.py:240-245).  The per-pair arithmetic itself
lives in the HIP device functions (``csrc/sph_kernels.h``); the vectorised
numpy ``kernel``/``gradient`` helpers below are host conveniences (initial
conditions, ``get_correction``), never part of the accelerated path.

    CubicSpline      kernels.py:29-163     radius_scale 2, deltap 2/3
    WendlandQuintic  kernels.py:274-380    radius_scale 2, deltap 1/2
    QuinticSpline    kernels.py:1050-1210  radius_scale 3, deltap 0.7593...
    Gaussian         kernels.py:830-930    radius_scale 3, deltap 1/sqrt(2)
"""
import math
import numpy as np

KERNEL_IDS = {'CubicSpline': 1, 'WendlandQuintic': 2, 'QuinticSpline': 3,
              'Gaussian': 4}


class _Kernel(object):
    radius_scale = 2.0
    _sigma = {}
    _deltap = 0.5

    def __init__(self, dim=2):
        if dim not in self._sigma:
            raise ValueError('%s: Dim %d not supported' %
                             (type(self).__name__, dim))
        self.dim = dim
        self.fac = self._sigma[dim]
        self.radius_scale = type(self).radius_scale

    def get_deltap(self):
        return self._deltap

    # host-side helpers -------------------------------------------------
    def _norm(self, h):
        return self.fac / np.asarray(h, dtype=float) ** self.dim

    def kernel(self, xij=(0., 0., 0.), rij=1.0, h=1.0):
        q = np.asarray(rij, dtype=float) / h
        return self._w(q) * self._norm(h)

    def dwdq(self, rij=1.0, h=1.0):
        r = np.asarray(rij, dtype=float)
        return np.where(r > 1e-12, self._dw(r / h), 0.0) * self._norm(h)

    def gradient_h(self, xij=(0., 0., 0.), rij=1.0, h=1.0):
        """dW/dh (kernels.py gradient_h, e.g. :138-163): -fac h1 (dw q + w dim)"""
        q = np.asarray(rij, dtype=float) / h
        return -self._norm(h) / h * (self._dw(q) * q + self._w(q) * self.dim)

    def gradient(self, xij=(0., 0., 0.), rij=1.0, h=1.0, grad=None):
        r = np.asarray(rij, dtype=float)
        safe = np.where(r > 1e-12, r, 1.0)
        tmp = np.where(r > 1e-12, self.dwdq(r, h) / (h * safe), 0.0)
        g = [tmp * xij[0], tmp * xij[1], tmp * xij[2]]
        if grad is not None:
            grad[0], grad[1], grad[2] = g
        return g


class CubicSpline(_Kernel):
    radius_scale = 2.0
    _sigma = {1: 2.0 / 3.0, 2: 10 * (1.0 / math.pi) / 7.0, 3: 1.0 / math.pi}
    _deltap = 2. / 3

    def __init__(self, dim=1):
        _Kernel.__init__(self, dim)

    @staticmethod
    def _w(q):
        return np.where(q > 2.0, 0.0, np.where(
            q > 1.0, 0.25 * (2. - q) ** 3, 1 - 1.5 * q * q * (1 - 0.5 * q)))

    @staticmethod
    def _dw(q):
        return np.where(q > 2.0, 0.0, np.where(
            q > 1.0, -0.75 * (2. - q) ** 2, -3.0 * q * (1 - 0.75 * q)))


class WendlandQuintic(_Kernel):
    radius_scale = 2.0
    _sigma = {2: 7.0 * (1.0 / math.pi) / 4.0, 3: (1.0 / math.pi) * 21.0 / 16.0}
    _deltap = 0.5

    @staticmethod
    def _w(q):
        return np.where(q < 2.0, (1. - 0.5 * q) ** 4 * (2.0 * q + 1.0), 0.0)

    @staticmethod
    def _dw(q):
        return np.where(q < 2.0, -5.0 * q * (1. - 0.5 * q) ** 3, 0.0)


class QuinticSpline(_Kernel):
    radius_scale = 3.0
    _sigma = {1: 1.0 / 120.0, 2: (1.0 / math.pi) * 7.0 / 478.0,
              3: (1.0 / math.pi) * 1.0 / 120.0}
    _deltap = 0.759298480738450

    @staticmethod
    def _w(q):
        c = lambda a: np.clip(a - q, 0.0, None) ** 5
        return c(3.) - 6.0 * c(2.) + 15.0 * c(1.)

    @staticmethod
    def _dw(q):
        c = lambda a: np.clip(a - q, 0.0, None) ** 4
        return -5.0 * c(3.) + 30.0 * c(2.) - 75.0 * c(1.)


class Gaussian(_Kernel):
    radius_scale = 3.0
    _g = 0.5 * (2.0 / math.sqrt(math.pi))
    _sigma = {1: _g, 2: _g * _g, 3: _g * _g * _g}
    _deltap = 0.70710678118654746

    @staticmethod
    def _w(q):
        return np.where(q < 3.0, np.exp(-q * q), 0.0)

    @staticmethod
    def _dw(q):
        return np.where(q < 3.0, -2.0 * q * np.exp(-q * q), 0.0)


def get_correction(kernel, h0):
    """W(deltap*h0, h0)  (kernels.py:10-12)."""
    return float(kernel.kernel(rij=kernel.get_deltap() * h0, h=h0))


def kernel_id(kernel):
    """Integer id used by the C-ABI for a kernel object (ours or the
    reference's: matched by class name)."""
    name = type(kernel).__name__
    if name not in KERNEL_IDS:
        raise NotImplementedError(
            'HIP backend: smoothing kernel %s is not implemented '
            '(have %s)' % (name, sorted(KERNEL_IDS)))
    return KERNEL_IDS[name]
