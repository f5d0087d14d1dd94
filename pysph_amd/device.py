"""ctypes binding of ``libsphhip.so`` + the device mirror of a ParticleArray.

``HipDeviceHelper`` plays the role of the reference's ``DeviceHelper``
(pysph/base/device_helper.py:47-672, reached as ``pa.gpu``): ``push`` /
``pull`` / ``resize`` / ``max`` with the same names and argument meaning, but
over an explicit host<->HIP buffer table behind the C-ABI
(``include/sphhip.h``) instead of compyle arrays.

There is NO CPU fallback: if the shared library cannot be loaded, or no GPU is
visible, constructing a context raises.
"""
import ctypes as C
import os

import numpy as np

from .particle_array import get_npy

_HERE = os.path.dirname(os.path.abspath(__file__))
# (SPH_LIBRARY: another build of the same sources, for A/B measurements of compile-time switches)
LIB_PATH = os.environ.get('SPH_LIBRARY') or os.path.join(_HERE, 'libsphhip.so')

MAX_ARRAYS = 8
MAX_PAR = 16


class SphKernel(C.Structure):
    _fields_ = [('kind', C.c_int), ('dim', C.c_int), ('fac', C.c_double),
                ('radius_scale', C.c_double), ('deltap', C.c_double)]


class SphEquation(C.Structure):
    _fields_ = [('kind', C.c_int), ('dest', C.c_int), ('nsrc', C.c_int),
                ('src', C.c_int * MAX_ARRAYS), ('par', C.c_double * MAX_PAR)]


class SphGroup(C.Structure):
    _fields_ = [('real', C.c_int), ('start_idx', C.c_long),
                ('stop_idx', C.c_long), ('neq', C.c_int),
                ('eqs', C.POINTER(SphEquation)),
                # optional promises about this group inside one evaluation
                # (include/sphhip.h; set by acceleration_eval.annotate_plan)
                ('src_eos', C.c_int), ('eos_par', C.c_double * 4),
                ('nl_mode', C.c_int), ('phase', C.c_int)]


# every symbol include/sphhip.h declares, with its ctypes signature
_P = C.c_void_p
_PD = C.POINTER(C.c_double)
_PU = C.POINTER(C.c_uint32)
SIGNATURES = {
    'sph_abi_sizeof': (C.c_long, [C.c_char_p]),
    'sph_abi_offsetof': (C.c_long, [C.c_char_p, C.c_char_p]),
    'sph_ctx_create': (C.c_int, [C.c_int, _P, C.POINTER(_P)]),
    'sph_ctx_destroy': (C.c_int, [_P]),
    'sph_ctx_synchronize': (C.c_int, [_P]),
    'sph_last_error': (C.c_char_p, []),
    'sph_version': (C.c_char_p, []),
    'sph_prop_id': (C.c_int, [C.c_char_p]),
    'sph_array_resize': (C.c_int, [_P, C.c_int, C.c_size_t, C.c_size_t]),
    'sph_array_size': (C.c_int, [_P, C.c_int, C.POINTER(C.c_size_t),
                                 C.POINTER(C.c_size_t)]),
    'sph_array_ensure_prop': (C.c_int, [_P, C.c_int, C.c_int]),
    'sph_array_permute': (C.c_int, [_P, C.c_int, C.POINTER(C.c_uint32), C.c_size_t,
                                    C.c_size_t]),
    'sph_array_push': (C.c_int, [_P, C.c_int, C.c_int, _PD, C.c_size_t,
                                 C.c_size_t]),
    'sph_array_pull': (C.c_int, [_P, C.c_int, C.c_int, _PD, C.c_size_t,
                                 C.c_size_t]),
    'sph_array_device_ptr': (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(_P)]),
    'sph_nnps_update': (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_int),
                                  C.c_double, C.c_double, _PD]),
    'sph_nnps_info': (C.c_int, [_P, _PD, C.POINTER(C.c_long)]),
    'sph_nnps_minmax': (C.c_int, [_P, C.c_int, C.POINTER(C.c_int), _PD]),
    'sph_nnps_get_csr': (C.c_int, [_P, C.c_int, C.c_int, _PU, C.c_size_t, _PU,
                                   C.c_size_t, C.POINTER(C.c_size_t)]),
    'sph_nnps_get_order': (C.c_int, [_P, C.c_int, _PU]),
    'sph_nnps_reorder_array': (C.c_int, [_P, C.c_int]),
    'sph_eval_group': (C.c_int, [_P, C.POINTER(SphKernel),
                                 C.POINTER(SphGroup), C.c_double, C.c_double]),
    'sph_halo_select': (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_double,
                                  C.c_double, C.c_double, C.c_size_t,
                                  C.POINTER(C.c_size_t)]),
    'sph_domain_box_wrap': (C.c_int, [_P, C.c_int, C.c_int, C.c_double,
                                      C.c_double, C.c_double]),
    'sph_domain_images_padded': (C.c_int, [_P, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double,
                                          C.POINTER(C.c_size_t), C.c_int, C.POINTER(C.c_int)]),
    'sph_domain_counts_queue': (C.c_int, [_P]),
    'sph_domain_counts_collect': (C.c_int, [_P, _PD]),
    'sph_array_h_known': (C.c_int, [_P, C.c_int, _PD, _PD]),
    'sph_array_props': (C.c_int, [_P, C.c_int, C.POINTER(C.c_int), C.c_int,
                                  C.POINTER(C.c_int)]),
    'sph_halo_pack': (C.c_int, [_P, C.c_int, C.c_int, C.c_int,
                                C.POINTER(C.c_int), C.c_int, C.c_double, _P]),
    'sph_halo_pack_mirror': (C.c_int, [_P, C.c_int, C.c_int, C.c_int,
                                       C.POINTER(C.c_int), C.c_int, C.c_double, _P]),
    'sph_halo_image': (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int,
                                 C.c_double, C.POINTER(C.c_size_t)]),
    'sph_halo_append': (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_int),
                                  _P, C.c_size_t]),
    'sph_halo_append_padded': (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_int), _P, C.c_size_t,
                                        C.c_double, C.c_double, _P]),
    'sph_halo_append_padded2': (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_int), _P, C.c_size_t, _P, C.c_size_t,
                                         C.c_double, C.c_double, _P]),
    'sph_halo_append_strided': (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_int),
                                          _P, C.c_size_t, C.c_size_t]),
    'sph_halo_append_promised': (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_int), _P, C.c_size_t,
                                           C.c_double, C.c_double, _P]),
    'sph_halo_select_pack': (C.c_int, [_P, C.c_int, C.c_int, C.c_double, C.c_double, C.c_size_t,
                                       C.c_int, C.POINTER(C.c_int), _PD, C.POINTER(C.c_size_t),
                                       C.POINTER(_P)]),
    'sph_halo_select_pack_promised': (C.c_int, [_P, C.c_int, C.c_int, C.c_double, C.c_double, C.c_size_t,
                                                C.c_int, C.POINTER(C.c_int), _PD, C.POINTER(C.c_size_t),
                                                C.POINTER(_P), C.c_double, C.c_double]),
    'sph_array_fill': (C.c_int, [_P, C.c_int, C.c_int, C.c_double, C.c_size_t, C.c_size_t]),
    'sph_coord_histogram': (C.c_int, [_P, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.POINTER(C.c_uint32)]),
    'sph_queue_values': (C.c_int, [_P, C.c_int, C.POINTER(_P), C.c_int, _P, _P]),
    'sph_array_mark_written': (C.c_int, [_P, C.c_int, C.c_int]),
    'sph_nnps_set_h_range': (C.c_int, [_P, C.c_double, C.c_double]),
    'sph_nnps_set_extend': (C.c_int, [_P, C.c_double, C.c_double, C.c_double]),
    'sph_nnps_update_ghosts': (C.c_int, [_P, C.c_int, C.c_double, C.c_double]),
    'sph_nnps_set_ghost_faces': (C.c_int, [_P, C.c_int, C.c_double, C.c_double]),
    'sph_read_values': (C.c_int, [_P, C.c_int, C.POINTER(_P), _PD]),
    'sph_halo_remove_selected': (C.c_int, [_P, C.c_int, C.POINTER(C.c_size_t)]),
    'sph_prop_register': (C.c_int, [C.c_char_p]),
    'sph_eval_generated': (C.c_int, [_P, _P, _P, C.c_double, C.c_double]),
    'sph_reduce_max': (C.c_int, [_P, C.c_int, C.c_int, _PD]),
    'sph_reduce_min': (C.c_int, [_P, C.c_int, C.c_int, _PD]),
    'sph_integrate_stage': (C.c_int, [_P, C.c_int, C.c_int, C.c_int,
                                      C.c_double]),
    'sph_set_option': (C.c_int, [_P, C.c_char_p, C.c_long]),
    'sph_timer_enable': (C.c_int, [_P, C.c_int]),
    'sph_timer_reset': (C.c_int, [_P]),
    'sph_timer_get': (C.c_int, [_P, C.c_char_p, _PD, C.POINTER(C.c_long)]),
}

_LIB = None


class SphError(RuntimeError):
    pass


def load_library(path=None):
    """dlopen libsphhip.so and bind every declared symbol.  Raises if the
    extension has not been built -- there is no fallback path."""
    global _LIB
    if _LIB is None:
        path = path or LIB_PATH
        if not os.path.exists(path):
            raise SphError(
                'HIP extension %s is missing: run `python -c "import '
                '__graft_entry__ as g; g.build()"` (or make -C pysph_amd/csrc)'
                % path)
        # One HIP/HSA runtime per process: PyTorch-ROCm bundles its own
        # libamdhip64/libhsa-runtime64, and whichever runtime initialises
        # first owns the device (measured: creating a context from this
        # library before `import torch` leaves torch without GPUs).  torch is
        # this package's plumbing for streams, device buffers and RCCL, so let
        # it load its runtime first; libsphhip then binds to the same one.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if not exported
            fn.restype = res
            fn.argtypes = args
        _LIB = lib
    return _LIB


def _check(rc):
    if rc != 0:
        raise SphError('libsphhip error %d: %s' %
                       (rc, load_library().sph_last_error().decode()))


def prop_id(name):
    pid = load_library().sph_prop_id(name.encode())
    return pid


def prop_register(name):
    """id of a property, giving it a user slot if it is not built in
    (equation-specific arrays of generated families)."""
    pid = load_library().sph_prop_register(name.encode())
    if pid < 0:
        raise SphError('sph_prop_register(%s): %s' % (
            name, load_library().sph_last_error().decode()))
    return pid


GEN_MAX_PROPS, GEN_MAX_SPROPS, GEN_MAX_PAR = 48, 20, 64


class SphGenFamily(C.Structure):
    """``sph_gen_family`` of include/sphhip.h."""
    _fields_ = [('launch', C.c_void_p), ('dest', C.c_int), ('nsrc', C.c_int),
                ('src', C.c_int * MAX_ARRAYS),
                ('src_flags', C.c_uint32 * MAX_ARRAYS),
                ('n_sprops', C.c_int), ('sprops', C.c_int * GEN_MAX_SPROPS),
                ('n_din', C.c_int), ('din', C.c_int * GEN_MAX_PROPS),
                ('n_dout', C.c_int), ('dout', C.c_int * GEN_MAX_PROPS),
                ('npar', C.c_int), ('par', C.c_double * GEN_MAX_PAR),
                ('real', C.c_int), ('start_idx', C.c_long),
                ('stop_idx', C.c_long), ('split_init', C.c_int),
                ('loop_all', C.c_int), ('also_pair', C.c_int),
                ('init_pair', C.c_int), ('nstate', C.c_int),
                ('state', C.c_double * 16), ('launch_f32', C.c_void_p)]


class HipContext(object):
    """One GPU, one stream, up to 8 particle arrays (``sph_ctx``)."""

    def __init__(self, device=0, stream=None):
        self.lib = load_library()
        self._h = _P()
        if isinstance(stream, C.c_void_p):
            stream = stream.value
        stream = int(stream) if stream else None
        _check(self.lib.sph_ctx_create(device, stream, C.byref(self._h)))
        # the caller's HIP stream handle, or None: the library made its own.  A
        # null handle (torch's legacy default stream reports 0) is NOT a shared
        # stream: sph_ctx_create then creates a non-blocking stream of its own
        self.stream = stream
        self.device = device
        self._ids = {}
        self.options = {}

    def array_id(self, name, owner=None):
        """Device slot of the particle array called `name`.  A context mirrors
        ONE set of arrays: a second, different array object with the same name
        would silently share (and resize) the first one's device storage, so
        that is refused -- use a context of its own (HipContext()) for it."""
        if name not in self._ids:
            if len(self._ids) >= MAX_ARRAYS:
                raise SphError('at most %d particle arrays' % MAX_ARRAYS)
            self._ids[name] = len(self._ids)
        if owner is not None:
            import weakref
            owners = self.__dict__.setdefault('_owners', {})
            prev = owners.get(name)
            prev = prev() if prev is not None else None
            if prev is not None and prev is not owner:
                raise SphError('two different particle arrays named %r in one HipContext; '
                               'create a separate HipContext for the second set' % name)
            try:
                owners[name] = weakref.ref(owner)
            except TypeError:
                pass
        return self._ids[name]

    def synchronize(self):
        _check(self.lib.sph_ctx_synchronize(self._h))

    def set_option(self, key, value):
        _check(self.lib.sph_set_option(self._h, key.encode(), int(value)))
        self.options[key] = int(value)      # host mirror (what the generated families are built for)

    # timers ----------------------------------------------------------
    def timer_enable(self, on=True):
        _check(self.lib.sph_timer_enable(self._h, int(on)))

    def timer_reset(self):
        _check(self.lib.sph_timer_reset(self._h))

    def timer_get(self, key):
        ms = C.c_double()
        cnt = C.c_long()
        _check(self.lib.sph_timer_get(self._h, key.encode(), C.byref(ms),
                                      C.byref(cnt)))
        return ms.value, cnt.value

    def close(self):
        if self._h:
            t = self.__dict__.pop('_transport', None)
            if t is not None:       # a libsphcomm communicator goes before its context (sphcomm.h)
                t.close()
            self.lib.sph_ctx_destroy(self._h)
            self._h = _P()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_DEFAULT_CTX = {}


def get_context(device=0, stream=None):
    """Process-wide default context per device."""
    key = (device, stream)
    if key not in _DEFAULT_CTX:
        _DEFAULT_CTX[key] = HipContext(device, stream)
    return _DEFAULT_CTX[key]


def _as_f64(arr, name):
    if arr.dtype.kind in 'iu':
        # an integer property an equation reads (orig_idx, ...): the device copy
        # holds the same values as doubles (exact below 2^53)
        return np.ascontiguousarray(arr, dtype=np.float64)
    if arr.dtype != np.float64:
        raise SphError('property %s: the HIP backend mirrors fp64 arrays '
                       '(got %s)' % (name, arr.dtype))
    return np.ascontiguousarray(arr)


class DeviceProperty(object):
    """``pa.gpu.x``: what the reference's DeviceHelper exposes as a compyle
    Array (device_helper.py:96-128) -- ``get()`` / ``set()`` move the whole
    property, ``[a:b].get()`` a range of particles.  Integer properties (tag,
    pid, gid) are host-resident metadata of the device state and are served
    from the host array."""

    def __init__(self, helper, name, lo=0, hi=None):
        self._h, self.name, self._lo, self._hi = helper, name, lo, hi

    def _host(self):
        return get_npy(self._h._pa, self.name)

    def _stride(self):
        return getattr(self._h._pa, 'stride', {}).get(self.name, 1)

    @property
    def dtype(self):
        return self._host().dtype

    @property
    def data(self):
        return self._h.device_ptr(self.name)

    def __len__(self):
        return self._h.get_number_of_particles() * self._stride()

    def __getitem__(self, sl):
        if not isinstance(sl, slice) or sl.step not in (None, 1):
            raise IndexError('DeviceProperty: contiguous slices only')
        lo, hi, _ = sl.indices(len(self))
        return DeviceProperty(self._h, self.name, lo, hi)

    def get(self):
        h, host = self._h, self._host()
        if host.dtype != np.float64 or (self._stride() == 1 and prop_id(self.name) < 0):
            return host[self._lo:self._hi].copy()
        stride, n = self._stride(), h.get_number_of_particles()
        out = np.empty(n * stride)
        if stride == 1:
            h.pull_into(self.name, out)
        else:
            for k in range(stride):
                col = np.empty(n)
                _check(h.lib.sph_array_pull(h.ctx._h, h.array_id,
                                            prop_register('%s__%d' % (self.name, k)),
                                            col.ctypes.data_as(_PD), 0, n))
                out[k::stride] = col
        return out[self._lo:self._hi]

    def __setitem__(self, key, value):
        """``gpu.rho[0] = 1.0`` / ``gpu.tag[:] = 1``: read-modify-write of the
        whole property (a convenience, not a fast path)"""
        host = self._host()
        if host.dtype != np.float64:
            host[key] = value
            return
        cur = DeviceProperty(self._h, self.name).get()
        cur[key] = value
        DeviceProperty(self._h, self.name).set(cur)

    def set(self, values):
        h, host = self._h, self._host()
        values = np.asarray(values)
        if self._lo != 0 or self._hi is not None:
            raise SphError('DeviceProperty.set: whole properties only')
        if values.size != host.size:
            raise SphError('%s.set: %d values for %d slots' % (self.name, values.size, host.size))
        host[:] = values
        if host.dtype == np.float64:
            stride = self._stride()
            h.push(*([self.name] if stride == 1 else
                     ['%s__%d' % (self.name, k) for k in range(stride)]))


class HipDeviceHelper(object):
    """``pa.gpu`` for the HIP backend: the DeviceHelper surface of
    pysph/base/device_helper.py -- push/pull/resize/max/min (:130-239),
    property access ``gpu.x.get()``, and the structural operations align,
    align_particles, remove_particles, remove_tagged_particles, add_particles,
    extend, append_parray, extract_particles, empty_clone (:241-672).  The
    structural operations restructure the device arrays (``sph_array_permute``)
    AND the host arrays, so the two stay the same shape; integer properties
    (tag, pid, gid) live on the host."""

    def __init__(self, pa, ctx=None):
        self._pa = pa
        self.ctx = ctx or get_context()
        self.lib = self.ctx.lib
        self.array_id = self.ctx.array_id(pa.name, pa)
        self._n = -1
        # True when the device holds extra (ghost) particles the host array
        # does not have: sizes are then managed by the halo exchange
        self.managed = False
        self.resize(pa.get_number_of_particles())

    def get_number_of_particles(self, real=False):
        n, nr = C.c_size_t(), C.c_size_t()
        _check(self.lib.sph_array_size(self.ctx._h, self.array_id,
                                       C.byref(n), C.byref(nr)))
        return nr.value if real else n.value

    def resize(self, n=None):
        pa = self._pa
        if n is not None and n != pa.get_number_of_particles() and not self.managed \
                and hasattr(pa, 'resize'):
            # an explicit device resize (device_helper.py:130-150) keeps the host
            # array -- which holds the integer properties -- the same shape
            grow = n > pa.get_number_of_particles()
            nr0 = pa.get_number_of_particles(True)
            pa.resize(n)
            pa.set_num_real_particles(n if grow and nr0 == self._n else min(nr0, n))
        n = pa.get_number_of_particles() if n is None else n
        nreal = min(pa.get_number_of_particles(True), n)
        _check(self.lib.sph_array_resize(self.ctx._h, self.array_id, n, nreal))
        self._n = n

    def _sync_size(self):
        if self.managed:
            return
        pa = self._pa
        n = pa.get_number_of_particles()
        nreal = pa.get_number_of_particles(True)
        if n != self._n or nreal != self.get_number_of_particles(True):
            self.resize(n)

    def _component(self, name):
        """``base__k`` -> (host array of the stride-S property `base`, S, k) for
        the components generated families use; None for scalar properties."""
        base, sep, k = name.rpartition('__')
        pa = self._pa
        if sep and k.isdigit() and base in pa.properties:
            stride = getattr(pa, 'stride', {}).get(base, 1)
            if stride > 1 and int(k) < stride:
                return get_npy(pa, base), stride, int(k)
        return None

    def push(self, *props):
        """host -> device.  No args: every fp64 property the device knows."""
        pa = self._pa
        self._sync_size()
        if not props:
            props = [p for p in pa.properties if prop_id(p) >= 0]
        for p in props:
            if p in pa.properties and get_npy(pa, p).dtype != np.float64 and prop_id(p) < 0:
                continue              # tag / pid / gid: host-resident metadata
            comp = self._component(p)
            if comp is not None:      # one component of a strided property
                host, stride, k = comp
                col = np.ascontiguousarray(host[k::stride], dtype=np.float64)
                _check(self.lib.sph_array_push(
                    self.ctx._h, self.array_id, prop_register(p),
                    col.ctypes.data_as(_PD), 0, col.size))
                continue
            pid = prop_id(p)
            if pid < 0 and p in pa.properties:
                pid = prop_register(p)          # a user property, pushed by name
            if pid < 0:
                raise SphError('property %r has no device mirror' % p)
            arr = _as_f64(get_npy(pa, p), p)
            _check(self.lib.sph_array_push(
                self.ctx._h, self.array_id, pid,
                arr.ctypes.data_as(_PD), 0, arr.size))

    def pull(self, *props):
        """device -> host, into the host array's own buffer."""
        pa = self._pa
        if not props:
            props = [p for p in pa.properties if prop_id(p) >= 0]
        for p in props:
            if p in pa.properties and get_npy(pa, p).dtype != np.float64 and prop_id(p) < 0:
                continue              # tag / pid / gid: host-resident metadata
            comp = self._component(p)
            if comp is not None:
                host, stride, k = comp
                n = min(host.size // stride, self.get_number_of_particles())
                col = np.empty(n)
                _check(self.lib.sph_array_pull(
                    self.ctx._h, self.array_id, prop_register(p),
                    col.ctypes.data_as(_PD), 0, n))
                host[k:n * stride:stride] = col
                continue
            pid = prop_id(p)
            if pid < 0:
                raise SphError('property %r has no device mirror' % p)
            arr = get_npy(pa, p)
            if arr.dtype.kind in 'iu':
                tmp = np.empty(min(arr.size, self.get_number_of_particles()))
                _check(self.lib.sph_array_pull(self.ctx._h, self.array_id, pid,
                                               tmp.ctypes.data_as(_PD), 0, tmp.size))
                arr[:tmp.size] = tmp.astype(arr.dtype)
                continue
            if arr.dtype != np.float64 or not arr.flags.c_contiguous:
                raise SphError('pull(%s): host array must be contiguous fp64'
                               % p)
            _check(self.lib.sph_array_pull(
                self.ctx._h, self.array_id, pid, arr.ctypes.data_as(_PD), 0,
                min(arr.size, self.get_number_of_particles())))

    def pull_into(self, prop, out):
        """device -> a caller-provided fp64 buffer (first out.size values)."""
        if out.dtype != np.float64 or not out.flags.c_contiguous:
            raise SphError('pull_into(%s): buffer must be contiguous fp64' % prop)
        _check(self.lib.sph_array_pull(
            self.ctx._h, self.array_id, prop_id(prop), out.ctypes.data_as(_PD),
            0, min(out.size, self.get_number_of_particles())))
        return out

    def device_props(self):
        """ids of the properties that have device storage, ascending"""
        cap = 512                       # > SPH_PROP_COUNT; the library checks it
        ids = (C.c_int * cap)()
        cnt = C.c_int(0)
        _check(self.lib.sph_array_props(self.ctx._h, self.array_id, ids, cap,
                                        C.byref(cnt)))
        return [int(ids[k]) for k in range(cnt.value)]

    def sync_host(self, ghosts=False):
        """Make the host array mirror a device-managed array (after halo
        exchange / migration the device owns the particle count): resize the
        host array, pull every mirrored property.  ghosts=False keeps only the
        real particles on the host (what dump_output wants)."""
        pa = self._pa
        nreal = self.get_number_of_particles(True)
        n = self.get_number_of_particles() if ghosts else nreal
        pa.resize(n)
        pa.set_num_real_particles(nreal)
        on_device = set(self.device_props())
        for p in pa.properties:
            pid = prop_id(p)
            if pid not in on_device:
                continue
            arr = get_npy(pa, p)
            if arr.dtype == np.float64:
                _check(self.lib.sph_array_pull(self.ctx._h, self.array_id, pid,
                                               arr.ctypes.data_as(_PD), 0, n))
            elif arr.dtype.kind in 'iu':
                # integer metadata mirrored as doubles (gid of a slab-decomposed
                # array: it migrates with its particle)
                # The ghost exchange does not carry it (sph_halo_append writes the
                # listed fp64 properties only): rows >= nreal are uninitialised on
                # the device and get the reference's "no gid" sentinel
                # (UINT_MAX, particle_array.pyx) instead of a cast of garbage
                tmp = np.empty(nreal)
                if nreal:
                    _check(self.lib.sph_array_pull(self.ctx._h, self.array_id, pid,
                                                   tmp.ctypes.data_as(_PD), 0, nreal))
                arr[:nreal] = tmp.astype(arr.dtype)
                if n > nreal:
                    arr[nreal:n] = np.iinfo(arr.dtype).max

    def max(self, prop):
        out = C.c_double()
        _check(self.lib.sph_reduce_max(self.ctx._h, self.array_id,
                                       prop_id(prop), C.byref(out)))
        return out.value

    def min(self, prop):
        out = C.c_double()
        _check(self.lib.sph_reduce_min(self.ctx._h, self.array_id,
                                       prop_id(prop), C.byref(out)))
        return out.value

    def device_ptr(self, prop):
        ptr = _P()
        _check(self.lib.sph_array_device_ptr(self.ctx._h, self.array_id,
                                             prop_id(prop), C.byref(ptr)))
        return ptr.value

    # -- DeviceHelper surface: properties as attributes ------------------------
    def __getattr__(self, name):
        if name.startswith('_'):
            raise AttributeError(name)
        pa = self.__dict__.get('_pa')
        if pa is not None and name in pa.properties:
            return DeviceProperty(self, name)
        raise AttributeError('HipDeviceHelper has no attribute or property %r' % name)

    @property
    def properties(self):
        return list(self._pa.properties)

    # -- structural operations ---------------------------------------------------
    def _host_take(self, idx):
        """host arrays <- rows idx (all properties, stride-aware)"""
        pa = self._pa
        for key, arr in list(pa.properties.items()):
            st = getattr(pa, 'stride', {}).get(key, 1)
            pa.properties[key] = (arr[idx] if st == 1 else
                                  arr.reshape(-1, st)[idx].ravel()).copy()
        pa._n = int(len(idx))

    def align(self, indices):
        """new particle i = old particle indices[i] (device_helper.py:241-288);
        fewer indices than particles drops the rest"""
        idx = np.ascontiguousarray(np.asarray(indices).ravel(), dtype=np.uint32)
        pa = self._pa
        tag = get_npy(pa, 'tag')[idx]
        nreal = int(np.count_nonzero(tag == 0))
        if nreal and not np.all(tag[:nreal] == 0):
            nreal = min(self.get_number_of_particles(True), idx.size)   # caller aligns later
        _check(self.lib.sph_array_permute(
            self.ctx._h, self.array_id, idx.ctypes.data_as(C.POINTER(C.c_uint32)),
            idx.size, nreal))
        self._host_take(idx.astype(np.int64))
        pa.set_num_real_particles(nreal)
        self._n = idx.size

    def align_particles(self):
        """real (Local) particles first, stable (device_helper.py:290-345)"""
        tag = get_npy(self._pa, 'tag')
        order = np.argsort(tag != 0, kind='stable')
        self.align(order)
        nreal = int(np.count_nonzero(get_npy(self._pa, 'tag') == 0))
        self._pa.set_num_real_particles(nreal)
        _check(self.lib.sph_array_resize(self.ctx._h, self.array_id, order.size, nreal))

    def remove_particles(self, indices, align=True):
        """device_helper.py:480-525"""
        n = self.get_number_of_particles()
        keep = np.ones(n, dtype=bool)
        keep[np.asarray(indices, dtype=np.int64)] = False
        self.align(np.nonzero(keep)[0])
        if align:
            self.align_particles()

    def remove_tagged_particles(self, tag, align=True):
        """device_helper.py:527-560"""
        self.remove_particles(np.nonzero(get_npy(self._pa, 'tag') == tag)[0], align)

    def extend(self, num_particles):
        """num_particles new (Local, zero-valued) particles at the end
        (device_helper.py:596-610)"""
        if num_particles <= 0:
            return
        pa = self._pa
        n0 = self.get_number_of_particles()
        pa.resize(n0 + num_particles)
        get_npy(pa, 'tag')[n0:] = 0
        _check(self.lib.sph_array_resize(self.ctx._h, self.array_id, n0 + num_particles,
                                         self.get_number_of_particles(True)))
        self._n = n0 + num_particles

    def add_particles(self, align=True, **particle_props):
        """device_helper.py:562-594: append particles with the given property
        values (others default), then align"""
        if not particle_props:
            return
        pa = self._pa
        count = max(np.asarray(v).size // getattr(pa, 'stride', {}).get(k, 1)
                    for k, v in particle_props.items())
        n0 = self.get_number_of_particles()
        self.extend(count)
        for key, val in particle_props.items():
            st = getattr(pa, 'stride', {}).get(key, 1)
            get_npy(pa, key)[n0 * st:] = np.asarray(val).ravel()
        self._push_range(n0, n0 + count)
        if align:
            self.align_particles()

    def _push_range(self, lo, hi):
        pa = self._pa
        for key, arr in pa.properties.items():
            if arr.dtype != np.float64:
                continue
            st = getattr(pa, 'stride', {}).get(key, 1)
            for k in range(st):
                name = key if st == 1 else '%s__%d' % (key, k)
                pid = prop_id(name)
                if pid < 0:
                    continue
                _check(self.lib.sph_array_ensure_prop(self.ctx._h, self.array_id, pid))
                col = np.ascontiguousarray(arr[lo * st + k:hi * st:st] if st > 1 else arr[lo:hi])
                _check(self.lib.sph_array_push(self.ctx._h, self.array_id, pid,
                                               col.ctypes.data_as(_PD), lo, hi - lo))

    def append_parray(self, parray, align=True, update_constants=False):
        """device_helper.py:612-640"""
        n1 = parray.get_number_of_particles()
        if n1 == 0:
            return
        pa = self._pa
        n0 = self.get_number_of_particles()
        self.extend(n1)
        for key in pa.properties:
            if key in parray.properties:
                st = getattr(pa, 'stride', {}).get(key, 1)
                get_npy(pa, key)[n0 * st:] = get_npy(parray, key)
        self._push_range(n0, n0 + n1)
        if align:
            self.align_particles()

    def empty_clone(self, props=None):
        """a particle array with the same properties and no particles
        (device_helper.py:642-658)"""
        from .particle_array import ParticleArray
        pa = self._pa
        out = ParticleArray(name=pa.name)
        for key, arr in pa.properties.items():
            if props is None or key in props or key in ('tag', 'pid', 'gid'):
                st = getattr(pa, 'stride', {}).get(key, 1)
                out.properties[key] = np.empty(0, dtype=arr.dtype)
                if st != 1:
                    out.stride[key] = st
        for key, val in pa.constants.items():
            out.constants[key] = val.copy()
        return out

    def extract_particles(self, indices, dest_array=None, align=True, props=None):
        """the DEVICE values of the chosen particles as a (host) particle array
        (device_helper.py:660-672)"""
        idx = np.asarray(indices, dtype=np.int64).ravel()
        pa = self._pa
        out = dest_array if dest_array is not None else self.empty_clone(props)
        n0 = out.get_number_of_particles()
        out.resize(n0 + idx.size)
        for key in out.properties:
            if key not in pa.properties:
                continue
            st = getattr(pa, 'stride', {}).get(key, 1)
            vals = DeviceProperty(self, key).get()
            vals = vals[idx] if st == 1 else vals.reshape(-1, st)[idx].ravel()
            out.properties[key][n0 * st:] = vals
        if align:
            out.align_particles()
        return out


_HELPERS = {}


def attach(pa, ctx=None):
    """Give a particle array its device mirror (``pa.gpu``)."""
    gpu = getattr(pa, 'gpu', None)
    if not isinstance(gpu, HipDeviceHelper):
        gpu = _HELPERS.get(id(pa))
    if not isinstance(gpu, HipDeviceHelper) or (ctx and gpu.ctx is not ctx):
        gpu = HipDeviceHelper(pa, ctx)
        try:
            pa.gpu = gpu
        except AttributeError:  # a ParticleArray type without a settable gpu
            _HELPERS[id(pa)] = gpu
    return gpu
