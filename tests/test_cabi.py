"""CPU-only checks of the drop-in boundary: the C-ABI library is built, loads,
and exports exactly the symbols include/sphhip.h declares; the product never
imports the oracle; enum tables on both sides of ctypes agree."""
import ctypes
import os
import re

import pytest

from conftest import REPO


def _declared():
    text = open(os.path.join(REPO, 'include', 'sphhip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(sph_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from pysph_amd import device as dev
    names = _declared()
    assert len(names) >= 20
    assert sorted(dev.SIGNATURES) == names
    lib = ctypes.CDLL(dev.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), n
    dev.load_library()  # binds restype/argtypes for all of them


def test_property_and_enum_tables_agree():
    from pysph_amd import device as dev
    from pysph_amd import equations as E
    from pysph_amd import solid_mech as SM
    hdr = open(os.path.join(REPO, 'include', 'sphhip.h')).read()
    for name in ['x', 'y', 'z', 'u', 'v', 'w', 'h', 'm', 'rho', 'p', 'cs',
                 'arho', 'au', 'av', 'aw', 'ax', 'ay', 'az', 'dt_cfl',
                 'dt_force', 'V', 'uhat', 'auhat']:
        assert dev.prop_id(name) >= 0
    assert dev.prop_id('x') == 0 and dev.prop_id('nonsense') == -1
    for cname, val in re.findall(r'(SPH_EQ_[A-Z_]+)\s*=\s*(\d+)', hdr):
        py = getattr(E, cname[4:], None)
        if py is None:
            py = getattr(SM, cname[4:])
        assert py == int(val), cname


def test_no_gpu_means_loud_failure():
    """Without a GPU the product must raise, not fall back to a CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from pysph_amd import device as dev
    with pytest.raises(dev.SphError):
        dev.HipContext(0)


def test_product_does_not_import_oracle():
    pkg = os.path.join(REPO, 'pysph_amd')
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                src = open(os.path.join(root, f)).read()
                assert 'oracle' not in src.replace('Oracle', 'oracle') or \
                    f in (), (f, 'mentions the oracle')


def test_comm_library_exports_every_declared_symbol():
    """libsphcomm.so (RCCL transport for non-Python hosts): header == exports"""
    text = open(os.path.join(REPO, 'include', 'sphcomm.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    names = sorted(set(re.findall(r'\b(sph_[a-z0-9_]+)\s*\(', text)))
    assert names == ['sph_allreduce', 'sph_comm_destroy', 'sph_comm_init_all', 'sph_comm_init_rank',
                     'sph_comm_unique_id', 'sph_halo_exchange', 'sph_halo_exchange_all']
    from pysph_amd import device as dev
    dev.load_library()                       # libsphhip.so first (libsphcomm links it)
    lib = ctypes.CDLL(os.path.join(REPO, 'pysph_amd', 'libsphcomm.so'))
    for n in names:
        assert hasattr(lib, n), n


@pytest.mark.gpu
def test_comm_library_self_periodic_exchange_matches_torch_path():
    """The C-ABI transport on real RCCL with one rank that is its own neighbour
    on both faces of a periodic slab (the message-order case a 2-rank periodic
    run also hits): same ghosts, in the same order, as the torch.distributed
    path of pysph_amd/parallel.py -- and the min/max/sum all-reduce."""
    import numpy as np
    from pysph_amd import device as dev
    from pysph_amd.particle_array import get_particle_array_wcsph
    from pysph_amd.parallel import WCSPH_HALO_PROPS
    lib = dev.load_library()
    comm = ctypes.CDLL(os.path.join(REPO, 'pysph_amd', 'libsphcomm.so'))
    for f in ('sph_comm_unique_id', 'sph_comm_init_rank', 'sph_comm_destroy', 'sph_halo_exchange',
              'sph_allreduce'):
        getattr(comm, f).restype = ctypes.c_int
    rng = np.random.default_rng(5)
    n = 5000
    x = rng.uniform(0, 1, n)
    pa = get_particle_array_wcsph(name='fluid', x=x, y=rng.uniform(0, 1, n), z=rng.uniform(0, 1, n),
                                  h=0.05 * np.ones(n), m=np.ones(n), rho=1000 + rng.uniform(0, 1, n),
                                  u=rng.uniform(-1, 1, n))
    ctx = dev.HipContext(0)
    g = dev.attach(pa, ctx)
    g.managed = True
    g.push()
    uid = (ctypes.c_char * 128)()
    assert comm.sph_comm_unique_id(uid) == 0
    assert comm.sph_comm_init_rank(ctx._h, 0, 1, uid) == 0, lib.sph_last_error()
    props = (ctypes.c_int * len(WCSPH_HALO_PROPS))(*[dev.prop_id(p) for p in WCSPH_HALO_PROPS])
    counts = (ctypes.c_size_t * 4)()
    width = 0.1
    rc = comm.sph_halo_exchange(ctx._h, g.array_id, 0, ctypes.c_double(0.0), ctypes.c_double(1.0),
                                ctypes.c_double(width), 1, ctypes.c_double(1.0), len(props), props, 1, counts)
    assert rc == 0, lib.sph_last_error()
    n_lo, n_hi = int(np.count_nonzero(x < width)), int(np.count_nonzero(x >= 1.0 - width))
    assert list(counts) == [n_lo, n_hi, n_hi, n_lo]   # my hi list arrives on my lo face
    assert g.get_number_of_particles() == n + n_lo + n_hi and g.get_number_of_particles(True) == n
    gx = np.empty(n + n_lo + n_hi)
    g.pull_into('x', gx)
    gu = np.empty(n + n_lo + n_hi)
    g.pull_into('u', gu)
    # ghosts: lo face first (= the peer's hi list shifted by -period), ascending index order
    hi_idx, lo_idx = np.nonzero(x >= 1.0 - width)[0], np.nonzero(x < width)[0]
    assert np.array_equal(gx[n:n + n_hi], x[hi_idx] - 1.0)
    assert np.array_equal(gx[n + n_hi:], x[lo_idx] + 1.0)
    assert np.array_equal(gu[n:n + n_hi], pa.u[hi_idx])
    v = (ctypes.c_double * 3)(1.5, -2.0, 7.0)
    assert comm.sph_allreduce(ctx._h, v, 3, 1) == 0 and list(v) == [1.5, -2.0, 7.0]
    assert comm.sph_allreduce(ctx._h, v, 3, 2) == 0 and list(v) == [1.5, -2.0, 7.0]
    assert comm.sph_comm_destroy(ctx._h) == 0
    ctx.close()
