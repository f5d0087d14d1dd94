"""``pa.gpu`` (HipDeviceHelper) against the scenarios of the reference's
pysph/base/tests/test_device_helper.py (line numbers per test).  Differences
by design: the device mirrors the fp64 properties; tag / pid / gid are
host-resident metadata that the structural operations keep in step; a clone
made by ``empty_clone`` is a host array (two arrays of one name cannot share
a context)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture
def pa():
    from pysph_amd.particle_array import get_particle_array
    return get_particle_array(name='f', x=[0.0, 1.0], m=1.0, rho=2.0)


@pytest.fixture
def h(pa):
    from pysph_amd import device as dev
    gpu = dev.attach(pa, dev.HipContext(0))
    gpu.push()
    return gpu


def same(pa, h, props=('x', 'y', 'm', 'rho', 'tag')):
    for p in props:
        assert np.allclose(pa.properties[p], getattr(h, p).get()), p


def test_simple(pa, h):                                         # :34-50
    same(pa, h)


def test_push_with_args(pa, h):                                 # :52-73
    pa.x[:] = [2.0, 3.0]
    pa.rho[0] = 1.0
    pa.tag[:] = 1
    h.push('x', 'rho', 'tag')
    same(pa, h)
    assert list(h.x.get()) == [2.0, 3.0] and h.rho.get()[0] == 1.0


def test_push_with_no_args(pa, h):                              # :75-96
    pa.x[:] = 1.0
    pa.rho[:] = 1.0
    pa.m[:] = 1.0
    pa.tag[:] = [1, 2]
    h.push()
    same(pa, h)


def test_pull_with_args(pa, h):                                 # :98-119
    h.x.set(np.array([2.0, 3.0]))
    h.rho[0] = 1.0
    h.tag[:] = 1
    pa.x[:] = -7.0                       # host made stale on purpose
    h.pull('x', 'rho', 'tag')
    same(pa, h)
    assert list(pa.x) == [2.0, 3.0] and pa.rho[0] == 1.0 and list(pa.tag) == [1, 1]


def test_pull_with_no_args(pa, h):                              # :121-142
    h.x[:] = 1.0
    h.rho[:] = 1.0
    h.m[:] = 1.0
    h.tag[:] = np.array([1, 2])
    pa.m[:] = 99.0
    h.pull()
    same(pa, h)
    assert list(pa.m) == [1.0, 1.0]


def test_max_min(pa, h):                                        # :144-153
    assert h.max('x') == 1.0 and h.min('x') == 0.0


def test_adding_property_updates_gpu(pa, h):                    # :155-176
    pa.add_property('test', data=[3.0, 4.0])
    h.push('test')
    assert np.allclose(pa.test, h.test.get())
    assert not hasattr(h, 'not_a_property')


def test_resize(pa, h):                                         # :178-210
    h.resize(4)
    assert h.get_number_of_particles() == 4 and pa.get_number_of_particles() == 4
    same(pa, h)
    assert list(h.x.get()[:2]) == [0.0, 1.0]
    h.remove_particles([2, 3])
    ptr = h.x.data
    h.resize(2)
    assert h.x.data == ptr               # shrinking keeps the allocation (:200-203)
    same(pa, h)


def test_get_number_of_particles(pa, h):                        # :212-230
    h.resize(5)
    h.x.set(np.array([2.0, 3.0, 4.0, 5.0, 6.0]))
    h.tag.set(np.array([0, 0, 1, 0, 1]))
    h.align_particles()
    assert h.get_number_of_particles() == 5
    assert h.get_number_of_particles(real=True) == 3


def test_align_with_strided_property(pa):                       # :232-258
    from pysph_amd import device as dev
    pa.add_property('force', stride=3)
    h = dev.attach(pa, dev.HipContext(0))
    n = 5
    h.resize(n)
    h.x.set(np.array([2.0, 3.0, 4.0, 5.0, 6.0]))
    h.force.set(np.arange(n * 3, dtype=float))
    h.align(np.arange(4, -1, -1, dtype=np.int32))
    assert np.all(h.x.get() == np.array([6., 5., 4., 3., 2.]))
    expect = np.arange(n * 3).reshape(n, 3)[::-1, :].ravel()
    assert np.all(h.force.get() == expect)
    assert np.all(pa.force == expect)


def test_align_particles(pa, h):                                # :260-278
    h.resize(5)
    h.x.set(np.array([2.0, 3.0, 4.0, 5.0, 6.0]))
    h.tag.set(np.array([0, 0, 1, 0, 1]))
    h.align_particles()
    x = h.x.get()
    assert np.all(np.sort(x[:-2]) == np.array([2., 3., 5.]))
    assert list(pa.tag) == [0, 0, 0, 1, 1]


def test_remove_particles(pa, h):                               # :280-299
    h.resize(4)
    h.x.set(np.array([2.0, 3.0, 4.0, 5.0]))
    h.remove_particles(np.array([1, 2], dtype=np.uint32))
    assert np.all(np.sort(h.x.get()) == np.array([2., 5.]))
    assert h.get_number_of_particles() == 2 and pa.get_number_of_particles() == 2


def test_remove_tagged_particles(pa, h):                        # :301-318
    h.resize(5)
    h.x.set(np.array([2.0, 3.0, 4.0, 5.0, 6.0]))
    h.tag.set(np.array([0, 0, 1, 0, 1]))
    h.remove_tagged_particles(1)
    assert np.all(np.sort(h.x.get()) == np.array([2., 3., 5.]))


def test_add_particles(pa, h):                                  # :320-335
    h.add_particles(x=np.zeros(4, np.float32))
    assert np.all(np.sort(h.x.get()) == np.array([0., 0., 0., 0., 0., 1.]))
    # the other properties of the new particles are zero on the device too
    assert np.all(np.sort(h.rho.get()) == np.array([0., 0., 0., 0., 2., 2.]))


def test_extend(pa, h):                                         # :337-351
    h.extend(4)
    assert h.get_number_of_particles() == 6


def test_append_parray(pa, h):                                  # :353-367
    from pysph_amd.particle_array import get_particle_array
    pa2 = get_particle_array(name='s', x=[0.0, 1.0], m=1.0, rho=2.0)
    h.append_parray(pa2)
    assert h.get_number_of_particles() == 4
    assert list(h.x.get()) == [0.0, 1.0, 0.0, 1.0] and list(h.rho.get()) == [2.0] * 4


def test_empty_clone_and_extract_particles():                   # :369-403
    from pysph_amd import device as dev
    from pysph_amd.particle_array import get_particle_array
    pa = get_particle_array(name='f', x=[0.0, 1.0, 2.0, 3.0], m=1.0, rho=2.0)
    h = dev.attach(pa, dev.HipContext(0))
    h.push()
    clone = h.empty_clone()
    assert clone.get_number_of_particles() == 0 and clone.name == 'f'
    h.x.set(np.array([10.0, 11.0, 12.0, 13.0]))         # device differs from the host copy
    pa.x[:] = -1.0
    h.extract_particles(np.array([1, 2], dtype=np.uint32), clone)
    assert clone.get_number_of_particles() == 2
    assert list(clone.x) == [11.0, 12.0] and list(clone.rho) == [2.0, 2.0]
