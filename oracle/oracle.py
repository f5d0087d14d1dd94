"""ctypes front-end of the CPU ORACLE (``oracle/sph_oracle.c``).

TEST INFRASTRUCTURE ONLY: imported by ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` -- never by ``pysph_amd``.

It mirrors the reference call sequence of
``pysph/sph/tests/test_acceleration_eval.py:305-316``::

    nnps = OracleNNPS(dim, particles, radius_scale); nnps.update()
    ev = OracleEval(particles, equations, kernel); ev.set_nnps(nnps)
    ev.compute(t, dt)

operating in place on the host arrays (numpy views) of the particle arrays.
Equation objects may be ``pysph_amd.equations`` specs or the reference's own
classes; they are matched by class name with a table kept separately from the
product's, so a wrong id on either side shows up as a parity failure.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

# enum orc_prop order (sph_oracle.h)
PROPS = ['x', 'y', 'z', 'u', 'v', 'w', 'h', 'm', 'rho', 'p', 'cs',
         'arho', 'au', 'av', 'aw', 'ax', 'ay', 'az', 'dt_cfl', 'dt_force',
         'V', 'uhat', 'vhat', 'what', 'auhat', 'avhat', 'awhat',
         'v00', 'v01', 'v02', 'v10', 'v11', 'v12', 'v20', 'v21', 'v22',
         's00', 's01', 's02', 's11', 's12', 's22',
         'as00', 'as01', 'as02', 'as11', 'as12', 'as22',
         'r00', 'r01', 'r02', 'r11', 'r12', 'r22', 'e', 'ae']
NPROP = len(PROPS)
MAX_ARRAYS = 8
MAX_PAR = 16

KERNELS = {'CubicSpline': 1, 'WendlandQuintic': 2, 'QuinticSpline': 3,
           'Gaussian': 4}

# class name -> (oracle kind, parameter attribute names)
EQS = {
    'TaitEOS': (1, ('rho0', 'c0', 'gamma', 'p0')),
    'TaitEOSHGCorrection': (2, ('rho0', 'c0', 'gamma')),
    'ContinuityEquation': (3, ()),
    'MomentumEquation': (4, ('c0', 'alpha', 'beta', 'gx', 'gy', 'gz',
                             'tensile_correction')),
    'XSPHCorrection': (5, ('eps',)),
    'SummationDensity': (6, ()),
    'TVFSummationDensity': (7, ()),
    'StateEquation': (8, ('p0', 'rho0', 'b')),
    'MomentumEquationPressureGradient': (9, ('pb', 'gx', 'gy', 'gz', 'tdamp')),
    'MomentumEquationViscosity': (10, ('nu',)),
    'MomentumEquationArtificialViscosity': (11, ('c0', 'alpha')),
    'MomentumEquationArtificialStress': (12, ()),
    'IsothermalEOS': (13, ('rho0', 'c0', 'p0')),
    'MonaghanArtificialViscosity': (14, ('alpha', 'beta')),
    'VelocityGradient3D': (15, ()),
    'VelocityGradient2D': (16, ()),
    # '@name' = first value of the destination array's constant `name`
    # (d_G[0], d_wdeltap[0], d_n[0], d_c0_ref[0], d_rho_ref[0])
    'HookesDeviatoricStressRate': (17, ('@G',)),
    'MomentumEquationWithStress': (18, ('@wdeltap', '@n')),
    'MonaghanArtificialStress': (19, ('eps',)),
    'SolidIsothermalEOS': (21, ('@c0_ref', '@rho_ref')),
}


class _Array(C.Structure):
    _fields_ = [('n', C.c_long), ('n_real', C.c_long),
                ('p', C.POINTER(C.c_double) * NPROP)]


class _Kernel(C.Structure):
    _fields_ = [('kind', C.c_int), ('dim', C.c_int), ('fac', C.c_double),
                ('radius_scale', C.c_double), ('deltap', C.c_double)]


class _Equation(C.Structure):
    _fields_ = [('kind', C.c_int), ('dest', C.c_int), ('nsrc', C.c_int),
                ('src', C.c_int * MAX_ARRAYS), ('par', C.c_double * MAX_PAR)]


class _Group(C.Structure):
    _fields_ = [('real', C.c_int), ('start_idx', C.c_long),
                ('stop_idx', C.c_long), ('neq', C.c_int),
                ('eqs', C.POINTER(_Equation))]


def build(force=False):
    """Compile oracle/libsphoracle.so with gcc (see oracle/Makefile)."""
    so = os.path.join(_HERE, 'libsphoracle.so')
    src = [os.path.join(_HERE, f) for f in ('sph_oracle.c', 'sph_oracle.h')]
    if force or not os.path.exists(so) or \
            os.path.getmtime(so) < max(os.path.getmtime(s) for s in src):
        subprocess.check_call(['make', '-C', _HERE, 'libsphoracle.so'])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_nnps_create.restype = C.c_void_p
        L.orc_nnps_create.argtypes = [C.c_int, C.c_int, C.c_double]
        L.orc_nnps_destroy.argtypes = [C.c_void_p]
        L.orc_nnps_set_array.argtypes = [C.c_void_p, C.c_int, C.POINTER(_Array)]
        L.orc_nnps_update.argtypes = [C.c_void_p]
        L.orc_nnps_info.argtypes = [C.c_void_p, C.POINTER(C.c_double),
                                    C.POINTER(C.c_long)]
        for f in (L.orc_nnps_neighbors, L.orc_nnps_brute_force):
            f.restype = C.c_long
            f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_long,
                          C.POINTER(C.c_uint), C.c_long]
        L.orc_flatten.restype = C.c_long
        L.orc_flatten.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
        L.orc_unflatten.restype = None
        L.orc_unflatten.argtypes = [C.c_long, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int)]
        L.orc_valid_cell_index.restype = C.c_long
        L.orc_valid_cell_index.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_long]
        L.orc_nnps_csr.restype = C.c_long
        L.orc_nnps_csr.argtypes = [C.c_void_p, C.c_int, C.c_int,
                                   C.POINTER(C.c_uint), C.POINTER(C.c_uint),
                                   C.c_int]
        L.orc_compute.argtypes = [C.c_void_p, C.POINTER(_Kernel),
                                  C.POINTER(_Group), C.c_int, C.c_double,
                                  C.c_double, C.c_int]
        L.orc_kernel_w.restype = C.c_double
        L.orc_kernel_w.argtypes = [C.POINTER(_Kernel), C.c_double, C.c_double]
        L.orc_kernel_dwdq.restype = C.c_double
        L.orc_kernel_dwdq.argtypes = [C.POINTER(_Kernel), C.c_double, C.c_double]
        L.orc_kernel_gradient.argtypes = [C.POINTER(_Kernel),
                                          C.POINTER(C.c_double), C.c_double,
                                          C.c_double, C.POINTER(C.c_double)]
        L.orc_eigen3.argtypes = [C.POINTER(C.c_double)] * 3
        L.orc_last_error.restype = C.c_char_p
        _LIB = L
    return _LIB


def _npy(pa, name):
    if hasattr(pa, 'properties') and name in pa.properties and \
            isinstance(pa.properties[name], np.ndarray):
        return pa.properties[name]
    return pa.get_carray(name).get_npy_array()


def make_kernel(kernel):
    return _Kernel(KERNELS[type(kernel).__name__], int(kernel.dim),
                   float(kernel.fac), float(kernel.radius_scale),
                   float(kernel.get_deltap()))


class OracleNNPS(object):
    """LinkedListNNPS restatement (linked_list_nnps.pyx:28-383)."""

    def __init__(self, dim, particles, radius_scale=2.0):
        self.dim = dim
        self.particles = list(particles)
        self.radius_scale = radius_scale
        self.narrays = len(self.particles)
        self._L = lib()
        self._h = self._L.orc_nnps_create(dim, self.narrays, radius_scale)
        self._arrs = [_Array() for _ in self.particles]
        self._bind()

    def _bind(self):
        for i, pa in enumerate(self.particles):
            a = self._arrs[i]
            a.n = pa.get_number_of_particles()
            a.n_real = pa.get_number_of_particles(True)
            for k, prop in enumerate(PROPS):
                if prop in pa.properties:
                    arr = _npy(pa, prop)
                    assert arr.dtype == np.float64 and arr.flags.c_contiguous
                    a.p[k] = arr.ctypes.data_as(C.POINTER(C.c_double))
                else:
                    a.p[k] = None
            self._L.orc_nnps_set_array(self._h, i, C.byref(a))

    def update(self):
        self._bind()  # host arrays may have been reallocated
        rc = self._L.orc_nnps_update(self._h)
        if rc:
            raise RuntimeError(self._L.orc_last_error().decode())
        d8 = (C.c_double * 8)()
        i4 = (C.c_long * 4)()
        self._L.orc_nnps_info(self._h, d8, i4)
        self.cell_size, self.hmin = d8[0], d8[1]
        self.xmin = np.array(d8[2:5])
        self.xmax = np.array(d8[5:8])
        self.ncells_per_dim = np.array(i4[0:3])
        self.n_cells = i4[3]

    def get_nearest_particles(self, src_index, dst_index, d_idx):
        cap = 4096
        while True:
            out = np.empty(cap, dtype=np.uint32)
            n = self._L.orc_nnps_neighbors(
                self._h, src_index, dst_index, d_idx,
                out.ctypes.data_as(C.POINTER(C.c_uint)), cap)
            if n <= cap:
                return out[:n].copy()
            cap = n

    def brute_force_neighbors(self, src_index, dst_index, d_idx):
        cap = self.particles[src_index].get_number_of_particles()
        out = np.empty(max(cap, 1), dtype=np.uint32)
        n = self._L.orc_nnps_brute_force(
            self._h, src_index, dst_index, d_idx,
            out.ctypes.data_as(C.POINTER(C.c_uint)), cap)
        return out[:n].copy()

    def count_csr(self, src_index, dst_index, nthreads=1):
        """start[nd+1] only (exclusive scan of the neighbour counts)"""
        nd = self.particles[dst_index].get_number_of_particles()
        start = np.zeros(nd + 1, dtype=np.uint32)
        self._L.orc_nnps_csr(self._h, src_index, dst_index,
                             start.ctypes.data_as(C.POINTER(C.c_uint)), None, nthreads)
        return start

    def get_csr(self, src_index, dst_index, nthreads=1):
        nd = self.particles[dst_index].get_number_of_particles()
        start = np.zeros(nd + 1, dtype=np.uint32)
        sp = start.ctypes.data_as(C.POINTER(C.c_uint))
        total = self._L.orc_nnps_csr(self._h, src_index, dst_index, sp, None,
                                     nthreads)
        nbrs = np.empty(max(total, 1), dtype=np.uint32)
        self._L.orc_nnps_csr(self._h, src_index, dst_index, sp,
                             nbrs.ctypes.data_as(C.POINTER(C.c_uint)), nthreads)
        return start, nbrs[:total]

    def __del__(self):
        try:
            self._L.orc_nnps_destroy(self._h)
        except Exception:
            pass


def _eq_name(eq):
    name = type(eq).__name__
    if name == 'SummationDensity' and 'transport_velocity' in type(eq).__module__:
        name = 'TVFSummationDensity'
    if name == 'IsothermalEOS' and 'solid_mech' in type(eq).__module__:
        name = 'SolidIsothermalEOS'
    return name


class OracleEval(object):
    """Generated ``AccelerationEval.compute`` restatement
    (acceleration_eval_cython.mako:262-363)."""

    def __init__(self, particle_arrays, equations, kernel, nthreads=1):
        self.particle_arrays = list(particle_arrays)
        names = [pa.name for pa in self.particle_arrays]
        self.kernel = make_kernel(kernel)
        self.nthreads = nthreads
        groups = equations
        if not all(hasattr(g, 'equations') for g in groups):
            groups = [type('G', (), dict(equations=list(equations), real=True,
                                         start_idx=0, stop_idx=None))()]
        self._keep = []
        cg = (_Group * len(groups))()
        for gi, g in enumerate(groups):
            if getattr(g, 'has_subgroups', False) or getattr(g, 'iterate', False):
                raise NotImplementedError('oracle: sub-groups / iterate')
            ce = (_Equation * len(g.equations))()
            for ei, eq in enumerate(g.equations):
                kind, pars = EQS[_eq_name(eq)]
                ce[ei].kind = kind
                ce[ei].dest = names.index(eq.dest)
                srcs = eq.sources or []
                ce[ei].nsrc = len(srcs)
                for k, s in enumerate(srcs):
                    ce[ei].src[k] = names.index(s)
                dest_pa = self.particle_arrays[names.index(eq.dest)]
                for k, p in enumerate(pars):
                    if p.startswith('@'):
                        ce[ei].par[k] = float(np.ravel(dest_pa.constants[p[1:]])[0])
                    else:
                        ce[ei].par[k] = float(getattr(eq, p))
            self._keep.append(ce)
            cg[gi].real = 1 if g.real else 0
            cg[gi].start_idx = self._idx(g.start_idx, g, 0)
            cg[gi].stop_idx = self._idx(g.stop_idx, g, -1)
            cg[gi].neq = len(g.equations)
            cg[gi].eqs = ce
        self._groups = cg
        self._ng = len(groups)
        self.nnps = None

    def _idx(self, v, g, default):
        if v is None:
            return default
        if isinstance(v, str):
            dest = [pa for pa in self.particle_arrays
                    if pa.name == g.equations[0].dest][0]
            return int(_npy(dest, v)[0])
        return int(v)

    def set_nnps(self, nnps):
        self.nnps = nnps

    def compute(self, t, dt):
        self.nnps._bind()
        rc = lib().orc_compute(self.nnps._h, C.byref(self.kernel), self._groups,
                               self._ng, t, dt, self.nthreads)
        if rc:
            raise RuntimeError(lib().orc_last_error().decode())


def kernel_w(kernel, rij, h):
    k = make_kernel(kernel)
    return lib().orc_kernel_w(C.byref(k), rij, h)


def kernel_dwdq(kernel, rij, h):
    k = make_kernel(kernel)
    return lib().orc_kernel_dwdq(C.byref(k), rij, h)


def kernel_gradient(kernel, xij, rij, h):
    k = make_kernel(kernel)
    x = (C.c_double * 3)(*xij)
    g = (C.c_double * 3)()
    lib().orc_kernel_gradient(C.byref(k), x, rij, h, g)
    return [g[0], g[1], g[2]]


def eigen3(a):
    """linalg3.eigen_decomposition restatement: returns (d[3], V[3,3])."""
    A = np.ascontiguousarray(a, dtype=float).ravel()
    V = np.zeros(9)
    d = np.zeros(3)
    P = C.POINTER(C.c_double)
    lib().orc_eigen3(A.ctypes.data_as(P), V.ctypes.data_as(P), d.ctypes.data_as(P))
    return d, V.reshape(3, 3)


def flatten(cid, ncells_per_dim):
    """nnps_base.pyx:55-59 ``py_flatten``."""
    nc = (C.c_int * 3)(*[int(v) for v in ncells_per_dim])
    return int(lib().orc_flatten(int(cid[0]), int(cid[1]), int(cid[2]), nc))


def unflatten(cell_index, ncells_per_dim, dim):
    """nnps_base.pyx:97-101 ``py_unflatten``."""
    nc = (C.c_int * 3)(*[int(v) for v in ncells_per_dim])
    out = (C.c_int * 3)()
    lib().orc_unflatten(int(cell_index), nc, int(dim), out)
    return tuple(out)


def get_valid_cell_index(cid, ncells_per_dim, n_cells):
    """nnps_base.pyx:61-65 ``py_get_valid_cell_index``."""
    nc = (C.c_int * 3)(*[int(v) for v in ncells_per_dim])
    return int(lib().orc_valid_cell_index(int(cid[0]), int(cid[1]), int(cid[2]), nc,
                                          int(n_cells)))
