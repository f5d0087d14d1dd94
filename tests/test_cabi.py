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


def test_ctypes_structures_match_the_library_they_are_loaded_against():
    """every ctypes Structure the product passes across the boundary has the
    size, the field order and the field offsets of the C struct AS THE LOADED
    LIBRARY WAS COMPILED (sph_abi_sizeof / sph_abi_offsetof), and the stub
    printed in INTEGRATION.md section 3 -- executed, not read -- has them too"""
    from pysph_amd import device as dev
    lib = dev.load_library()
    pairs = {'sph_kernel': dev.SphKernel, 'sph_equation': dev.SphEquation, 'sph_group': dev.SphGroup,
             'sph_gen_family': dev.SphGenFamily}
    assert lib.sph_abi_sizeof(b'no_such_struct') == -1
    assert lib.sph_abi_offsetof(b'sph_group', b'no_such_field') == -1

    def check(cname, st):
        assert ctypes.sizeof(st) == lib.sph_abi_sizeof(cname.encode()), cname
        for fname, _ in st._fields_:
            off = lib.sph_abi_offsetof(cname.encode(), fname.encode())
            assert off == getattr(st, fname).offset, (cname, fname, off)
    for cname, st in pairs.items():
        check(cname, st)
    assert lib.sph_abi_sizeof(b'sph_gen_args') > 0
    # the stub of INTEGRATION.md: the first python block after the section-3 heading
    text = open(os.path.join(REPO, 'INTEGRATION.md')).read()
    sec = text[text.index('## 3. The ctypes stub'):]
    code = sec[sec.index('```python') + 9:]
    code = code[:code.index('```')]
    decl = code[:code.index('ctx = C.c_void_p()')].replace('lib = C.CDLL("libsphhip.so")', '')
    ns = {}
    exec(decl, ns)
    for cname in ('sph_kernel', 'sph_equation', 'sph_group'):
        check(cname, ns[cname])


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
    assert names == ['sph_allreduce', 'sph_comm_destroy', 'sph_comm_init_all', 'sph_comm_init_rank', 'sph_comm_sendrecv',
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


@pytest.mark.gpu
@pytest.mark.parametrize('n', [70001, 1300007], ids=['one-chunk-trip', 'three-chunk-trips'])
def test_select_pack_equals_list_based_select_and_pack(n):
    """sph_halo_select_pack (selection + packing of both faces in one device
    pass, counts only in the message headers) against the list-based
    sph_halo_select / sph_halo_pack: same rows in the same order, the header
    carries the count, a capacity that is too small gives a NEGATIVE header,
    an open face (NULL message) is left alone; sph_read_values returns the
    headers in one round trip."""
    import ctypes as C
    import numpy as np
    import torch
    from pysph_amd import device as dev
    from pysph_amd.particle_array import get_particle_array_wcsph
    rng = np.random.default_rng(3)
    # (ragged last block; the larger array takes several 256-particle trips per chunk counter)
    pa = get_particle_array_wcsph(name='fluid', x=rng.uniform(0, 1, n), y=rng.uniform(0, 1, n),
                                  z=rng.uniform(0, 1, n), u=rng.uniform(-1, 1, n), rho=rng.uniform(1, 2, n),
                                  h=0.01 * np.ones(n), m=np.ones(n))
    # torch fills the message buffers, the library packs into them: both on ONE stream (torch's default
    # stream has the NULL handle, for which the library would create a stream of its own)
    ts = torch.cuda.Stream()
    torch.cuda.set_stream(ts)
    ctx = dev.HipContext(0, ts.cuda_stream)
    g = dev.attach(pa, ctx)
    g.push()
    lib = ctx.lib
    names = ('x', 'y', 'z', 'u', 'rho', 'h', 'm')
    props = (C.c_int * len(names))(*[dev.prop_id(p) for p in names])
    npr = len(names)
    lo_cut, hi_cut, shift = 0.07, 0.91, (0.5, -2.0)
    counts = (C.c_size_t * 2)()
    dev._check(lib.sph_halo_select(ctx._h, g.array_id, 0, 0, lo_cut, hi_cut, 0.0, 0, counts))
    cnt = [int(counts[0]), int(counts[1])]
    assert cnt == [int((pa.x < lo_cut).sum()), int((pa.x >= hi_cut).sum())] and min(cnt) > 1000
    ref = []
    for s in (0, 1):
        buf = torch.empty(cnt[s] * npr, dtype=torch.float64, device='cuda')
        dev._check(lib.sph_halo_pack(ctx._h, g.array_id, s, npr, props, 0, shift[s], C.c_void_p(buf.data_ptr())))
        ref.append(buf.cpu().numpy().reshape(npr, cnt[s]))
    # rows in ascending particle index, the axis coordinate shifted
    idx = np.nonzero(pa.x < lo_cut)[0]
    assert np.array_equal(ref[0][0], pa.x[idx] + shift[0]) and np.array_equal(ref[0][3], pa.u[idx])

    def select_pack(caps, with_lo=True):
        bufs = [torch.full((caps[s] * npr + 1,), -7.0, dtype=torch.float64, device='cuda') for s in (0, 1)]
        sh = (C.c_double * 2)(*shift)
        cp = (C.c_size_t * 2)(*caps)
        ds = (C.c_void_p * 2)(bufs[0].data_ptr() if with_lo else None, bufs[1].data_ptr())
        dev._check(lib.sph_halo_select_pack(ctx._h, g.array_id, 0, lo_cut, hi_cut, 0, npr, props, sh, cp, ds))
        ptrs = (C.c_void_p * 2)(*[b.data_ptr() + caps[s] * npr * 8 for s, b in enumerate(bufs)])
        hdr = (C.c_double * 2)()
        dev._check(lib.sph_read_values(ctx._h, 2, ptrs, hdr))
        return [b.cpu().numpy() for b in bufs], [hdr[0], hdr[1]]
    caps = [cnt[0] + 500, cnt[1] + 37]
    bufs, hdr = select_pack(caps)
    assert hdr == [float(cnt[0]), float(cnt[1])]
    for s in (0, 1):
        rows = bufs[s][:-1].reshape(npr, caps[s])
        assert np.array_equal(rows[:, :cnt[s]], ref[s]), s
        assert np.all(rows[:, cnt[s]:] == -7.0)             # nothing written behind the rows
        assert bufs[s][-1] == cnt[s]
    # a capacity that is too small: negative header, what fits is the head of the list
    small = [cnt[0] - 100, cnt[1] + 1]
    bufs, hdr = select_pack(small)
    assert hdr == [-float(cnt[0]), float(cnt[1])]
    assert np.array_equal(bufs[0][:-1].reshape(npr, small[0]), ref[0][:, :small[0]])
    # open low face: its message is not touched, the high face is complete
    bufs, hdr = select_pack(caps, with_lo=False)
    assert np.all(bufs[0] == -7.0) and hdr[1] == float(cnt[1])
    assert np.array_equal(bufs[1][:-1].reshape(npr, caps[1])[:, :cnt[1]], ref[1])
    # argument errors are errors, not overruns
    with pytest.raises(dev.SphError):
        dev._check(lib.sph_read_values(ctx._h, 65, None, None))
    with pytest.raises(dev.SphError):
        dev._check(lib.sph_halo_append_strided(ctx._h, g.array_id, npr, props, C.c_void_p(0), 10, 5))
    ctx.close()
    torch.cuda.set_stream(torch.cuda.default_stream())


@pytest.mark.gpu
def test_fixed_h_and_bounds_skip_the_reduction_but_not_the_result():
    """LinkedListNNPS(fixed_h=True) + nnps.bounds: sph_nnps_update takes the grid
    from the caller and the h range from the first update (no min/max pass, no
    round trip); the neighbour lists are those of the default path, also after
    the particles moved inside the bounds."""
    import numpy as np
    from test_hip_parity import make_cube
    from pysph_amd import device as dev
    from pysph_amd.nnps import HipNNPS
    pa, dx = make_cube(14)
    pb, _ = make_cube(14)
    ctx_a, ctx_b = dev.HipContext(0), dev.HipContext(0)
    na = HipNNPS(3, [pa], radius_scale=2.0, ctx=ctx_a)
    nb = HipNNPS(3, [pb], radius_scale=2.0, ctx=ctx_b, fixed_h=True)
    assert nb._h_fixed
    nb.bounds = (-0.2, -0.2, -0.2, 1.2, 1.2, 1.2)
    rng = np.random.default_rng(0)
    for step in range(3):
        d = 0.3 * dx * rng.uniform(-1, 1, (3, pa.get_number_of_particles()))
        for p in (pa, pb):
            p.x += d[0]; p.y += d[1]; p.z += d[2]
        na.update()
        nb.update()
        assert tuple(nb.xmin) == (-0.2, -0.2, -0.2) and nb.cell_size == na.cell_size
        sa, ia = na.get_csr(0, 0)
        sb, ib = nb.get_csr(0, 0)
        assert np.array_equal(sa, sb) and np.array_equal(ia, ib)
    ctx_a.close()
    ctx_b.close()


@pytest.mark.gpu
def test_fixed_h_range_is_per_nnps_not_per_context():
    """Two neighbour searches on ONE context (dev.get_context() is process-wide):
    the h range a fixed_h search found must not leak into a later search over
    other particles with other smoothing lengths (round-3 advisor finding: the
    range was sticky context state, giving a stale cell size / uniform-h flag)."""
    import numpy as np
    from test_hip_parity import make_cube
    from pysph_amd import device as dev
    from pysph_amd.nnps import HipNNPS
    ctx = dev.HipContext(0)
    ref_ctx = dev.HipContext(0)
    pa, dx = make_cube(12)
    n1 = HipNNPS(3, [pa], radius_scale=2.0, ctx=ctx, fixed_h=True)
    n1.update()
    assert n1._h_fixed and n1._h_range[0] == n1._h_range[1]
    # another problem on the same context: larger, NON-uniform h, not fixed
    pb, _ = make_cube(10)
    pc, _ = make_cube(10)
    pb.name = 'other'            # a context mirrors ONE array per name
    rng = np.random.default_rng(3)
    hb = pb.h * (1.5 + 0.3 * rng.uniform(-1, 1, pb.get_number_of_particles()))
    pb.h[:] = hb
    pc.h[:] = hb
    n2 = HipNNPS(3, [pb], radius_scale=2.0, ctx=ctx)
    n3 = HipNNPS(3, [pc], radius_scale=2.0, ctx=ref_ctx)     # never saw a fixed range
    for _ in range(2):
        n2.update()
        n3.update()
        assert n2.cell_size == n3.cell_size and n2.hmin == n3.hmin
        s2, i2 = n2.get_csr(0, 0)
        s3, i3 = n3.get_csr(0, 0)
        assert np.array_equal(s2, s3) and np.array_equal(i2, i3)
    # ... and the fixed one still gets ITS range back on its next update
    n1.update()
    assert abs(n1.cell_size - 2.0 * float(pa.h[0])) < 1e-15
    ctx.close()
    ref_ctx.close()


@pytest.mark.gpu
def test_coordinate_histogram_matches_numpy():
    """sph_coord_histogram (the re-balancing of a slab decomposition looks at the distribution of the real particles along
    the slab axis on the device; `SlabDecomposition.rebalance` pulled every coordinate to the host before): equal to
    numpy's bincount of the same binning rule, ghosts and parked padding rows not counted."""
    import numpy as np
    from pysph_amd import device as dev
    from pysph_amd.parallel import DeviceHaloOps, WCSPH_HALO_PROPS
    from pysph_amd.particle_array import get_particle_array_wcsph
    rng = np.random.default_rng(11)
    n = 50000
    x = np.concatenate([rng.uniform(0.0, 1.228, n), rng.uniform(0.0, 3.22, n // 5)])
    pa = get_particle_array_wcsph(name='fluid', x=x, y=rng.random(x.size), z=rng.random(x.size))
    pa.tag[-1000:] = 1                       # the last thousand are ghosts
    pa.x[-500:] = 1e18                       # ... half of them parked padding rows
    pa.align_particles() if hasattr(pa, 'align_particles') else None
    pa.set_num_real_particles(x.size - 1000)
    ctx = dev.HipContext(0)
    dev.attach(pa, ctx).push()
    ops = DeviceHaloOps(pa, ctx, WCSPH_HALO_PROPS, 0)
    nr = ops.n_real()
    assert nr == x.size - 1000
    lo, hi = ops.coord_range()
    assert lo == pa.x[:nr].min() and hi == pa.x[:nr].max()
    nbins = 512
    got = ops.histogram(lo, hi - lo, nbins)
    b = np.minimum(np.floor((pa.x[:nr] - lo) * (nbins / (hi - lo))).astype(np.int64), nbins - 1)
    want = np.bincount(b, minlength=nbins)
    assert got.sum() == nr and np.array_equal(got, want)
    ctx.close()
