"""Host-side particle container used at the drop-in boundary.

The reference's ``ParticleArray`` (pysph/base/particle_array.pyx:109-1453) is a
Cython class over ``cyarray`` buffers and cannot be built without ``cyarray``.
The MI355X backend only needs the *boundary surface* of that type:

  * ``name``, ``properties`` (name -> array), ``constants``,
  * ``num_real_particles`` / ``get_number_of_particles(real)`` with the real
    particles stored first (particle_array.pyx:423, :1092),
  * attribute access (``pa.x``), ``add_property`` / ``add_constant``,
  * ``get_carray(name).get_npy_array()`` (zero-copy view),
  * ``gpu`` -- the device mirror (reference: ``DeviceHelper``,
    pysph/base/device_helper.py:47; here :class:`pysph_amd.device.HipDeviceHelper`).

This numpy-backed class provides exactly that surface so the backend, its
tests and the benchmark run without the reference installed.  A real
``pysph.base.particle_array.ParticleArray`` is accepted wherever this class is
(see :func:`get_npy`): the host keeps owning the data either way.
"""
import numpy as np

UINT_MAX = 2 ** 32 - 1


class ParticleTAGS(object):
    """Tags for particles (particle_array.pyx ``ParticleTAGS``)."""
    Local = 0
    Remote = 1
    Ghost = 2


class _ArrayView(object):
    """What ``ParticleArray.get_carray`` returns: a thin handle with the
    ``get_npy_array()`` accessor the reference's carrays have."""

    def __init__(self, data):
        self._data = data

    def get_npy_array(self):
        return self._data

    @property
    def length(self):
        return self._data.size

    def __len__(self):
        return self._data.size


DEFAULT_PROPS = ('x', 'y', 'z', 'u', 'v', 'w', 'm', 'h', 'rho', 'p',
                 'au', 'av', 'aw', 'gid', 'pid', 'tag')

_INT_PROPS = {'tag': np.int32, 'pid': np.int32, 'gid': np.uint32}


class ParticleArray(object):
    def __init__(self, name='array', constants=None, **props):
        object.__setattr__(self, 'properties', {})
        object.__setattr__(self, 'constants', {})
        object.__setattr__(self, 'stride', {})
        self.name = name
        self.gpu = None
        self.output_property_arrays = []
        n = 0
        for key, val in props.items():
            n = max(n, np.asarray(val).size)
        self._n = n
        self.num_real_particles = n
        for key, val in props.items():
            self.add_property(key, data=val)
        for key in ('tag', 'pid', 'gid'):
            if key not in self.properties:
                self.add_property(key)
        if constants:
            for key, val in constants.items():
                self.add_constant(key, val)

    # -- attribute access mirrors the reference (pa.x is the numpy view) --
    def __getattr__(self, key):
        props = object.__getattribute__(self, 'properties')
        if key in props:
            return props[key]
        consts = object.__getattribute__(self, 'constants')
        if key in consts:
            return consts[key]
        raise AttributeError("ParticleArray %r has no property %r" % (
            object.__getattribute__(self, '__dict__').get('name'), key))

    def __setattr__(self, key, value):
        props = object.__getattribute__(self, 'properties')
        consts = object.__getattribute__(self, 'constants')
        if key in props:
            props[key][...] = value
        elif key in consts:
            consts[key][...] = value
        else:
            object.__setattr__(self, key, value)

    # -- reference API subset -------------------------------------------
    def get_number_of_particles(self, real=False):
        return self.num_real_particles if real else self._n

    def set_num_real_particles(self, n):
        self.num_real_particles = int(n)

    def add_property(self, name, type='double', default=None, data=None,
                     stride=1):
        dtype = _INT_PROPS.get(name)
        if dtype is None:
            dtype = {'double': np.float64, 'float': np.float32,
                     'int': np.int32, 'unsigned int': np.uint32,
                     'long': np.int64}[type]
        if default is None:
            default = UINT_MAX if name == 'gid' else 0
        arr = np.full(self._n * stride, default, dtype=dtype)
        if data is not None:
            d = np.asarray(data).ravel()
            if d.size == 1:
                arr[:] = d[0]
            elif d.size == arr.size:
                arr[:] = d
            else:
                raise ValueError('property %s: size %d != %d' %
                                 (name, d.size, arr.size))
        self.properties[name] = arr
        if stride != 1:
            self.stride[name] = stride

    def add_constant(self, name, data):
        self.constants[name] = np.atleast_1d(np.asarray(data, dtype=float)).copy()

    def get_carray(self, name):
        if name in self.properties:
            return _ArrayView(self.properties[name])
        return _ArrayView(self.constants[name])

    def get(self, *names, only_real_particles=True):
        n = self.num_real_particles if only_real_particles else self._n
        out = [self.properties[k][:n] for k in names]
        return out[0] if len(out) == 1 else out

    def set_output_arrays(self, names):
        self.output_property_arrays = list(names)

    def set_name(self, name):
        self.name = name

    def resize(self, n):
        """Grow/shrink every property (new slots zero / default)."""
        for key, arr in list(self.properties.items()):
            stride = self.stride.get(key, 1)
            new = np.zeros(n * stride, dtype=arr.dtype)
            if key == 'gid':
                new[:] = UINT_MAX
            m = min(arr.size, new.size)
            new[:m] = arr[:m]
            self.properties[key] = new
        self._n = n

    def append_parray(self, other, tag=None):
        """Append `other`'s particles (e.g. ghosts); they land after the
        existing ones, so real particles stay first when tag != Local."""
        n0, n1 = self._n, other.get_number_of_particles()
        self.resize(n0 + n1)
        for key, arr in self.properties.items():
            if key in other.properties:
                s = self.stride.get(key, 1)
                arr[n0 * s:] = other.properties[key]
        if tag is not None:
            self.properties['tag'][n0:] = tag

    def extract_particles(self, indices, name=None):
        idx = np.asarray(indices, dtype=np.int64)
        out = ParticleArray(name=name or self.name)
        out._n = idx.size
        out.num_real_particles = idx.size
        for key, arr in self.properties.items():
            s = self.stride.get(key, 1)
            if s == 1:
                out.properties[key] = arr[idx].copy()
            else:
                out.properties[key] = arr.reshape(-1, s)[idx].ravel().copy()
                out.stride[key] = s
        for key, val in self.constants.items():
            out.constants[key] = val.copy()
        return out

    def remove_tagged_particles(self, tag):
        keep = np.nonzero(self.properties['tag'] != tag)[0]
        for key, arr in list(self.properties.items()):
            s = self.stride.get(key, 1)
            self.properties[key] = (arr[keep].copy() if s == 1 else
                                    arr.reshape(-1, s)[keep].ravel().copy())
        self._n = keep.size
        self.num_real_particles = int(
            np.count_nonzero(self.properties['tag'] == ParticleTAGS.Local))

    def align_particles(self):
        """Real (Local) particles first (particle_array.pyx:1092)."""
        tag = self.properties['tag']
        order = np.argsort(tag != ParticleTAGS.Local, kind='stable')
        for key, arr in list(self.properties.items()):
            s = self.stride.get(key, 1)
            self.properties[key] = (arr[order] if s == 1 else
                                    arr.reshape(-1, s)[order].ravel())
        self.num_real_particles = int(np.count_nonzero(tag == ParticleTAGS.Local))


def get_npy(pa, name):
    """Numpy view of property/constant `name` for either this module's
    ParticleArray or the reference's (particle_array.pxd:36-138)."""
    if isinstance(pa, ParticleArray):
        if name in pa.properties:
            return pa.properties[name]
        return pa.constants[name]
    return pa.get_carray(name).get_npy_array()


def has_prop(pa, name):
    return name in pa.properties or name in pa.constants


def get_particle_array(additional_props=None, constants=None, **props):
    """Particle array with the default SPH properties (reference factory:
    pysph/base/utils.py:41-131, ``DEFAULT_PROPS`` :36-39)."""
    name = props.pop('name', 'array')
    pa = ParticleArray(name=name, constants=constants, **props)
    wanted = list(DEFAULT_PROPS) + list(additional_props or [])
    for prop in wanted:
        if prop not in pa.properties:
            pa.add_property(prop)
    pa.set_output_arrays(['x', 'y', 'z', 'u', 'v', 'w', 'rho', 'm', 'h',
                          'pid', 'gid', 'tag'])
    return pa


WCSPH_PROPS = ['cs', 'ax', 'ay', 'az', 'arho', 'x0', 'y0', 'z0', 'u0', 'v0',
               'w0', 'rho0', 'div', 'dt_cfl', 'dt_force']


def get_particle_array_wcsph(constants=None, **props):
    """utils.py:134-166."""
    pa = get_particle_array(constants=constants, additional_props=WCSPH_PROPS,
                            **props)
    pa.set_output_arrays(['x', 'y', 'z', 'u', 'v', 'w', 'rho', 'm', 'h',
                          'pid', 'gid', 'tag', 'p'])
    return pa


TVF_FLUID_PROPS = ['uhat', 'vhat', 'what', 'auhat', 'avhat', 'awhat', 'vmag2',
                   'V', 'arho', 'x0', 'y0', 'z0', 'u0', 'v0', 'w0', 'cs']


TVF_SOLID_PROPS = ['u0', 'v0', 'w0', 'V', 'wij', 'ax', 'ay', 'az', 'uf', 'vf',
                   'wf', 'ug', 'vg', 'wg']


def get_particle_array_tvf_solid(constants=None, **props):
    """utils.py ``get_particle_array_tvf_solid`` (:329-360)."""
    pa = get_particle_array(constants=constants,
                            additional_props=TVF_SOLID_PROPS, **props)
    pa.set_output_arrays(['x', 'y', 'z', 'u', 'v', 'w', 'rho', 'p', 'h', 'm',
                          'V', 'pid', 'gid', 'tag'])
    return pa


def get_particle_array_tvf_fluid(constants=None, **props):
    """utils.py ``get_particle_array_tvf_fluid``."""
    pa = get_particle_array(constants=constants,
                            additional_props=TVF_FLUID_PROPS, **props)
    pa.set_output_arrays(['x', 'y', 'z', 'u', 'v', 'w', 'rho', 'p', 'h', 'm',
                          'au', 'av', 'aw', 'V', 'vmag2', 'pid', 'gid', 'tag'])
    return pa
