"""The hand-written particle sort of the neighbour update (round 5:
k_bin_keys / k_bucket_scatter / k_bucket_sort, csrc/sph_nnps.hip) and the
update without a device->host round trip (bounds of the previous update,
DevArray::h_dirty / m_dirty).

The sort must give what a STABLE sort of the fine keys gives (the order inside
a key is ascending particle index: linked_list_nnps.pyx:235-291 visits a cell's
particles in a fixed order, and the sums of a destination are taken in cell
order) for every bucket shape: LDS-staged buckets, buckets beyond the LDS stage
(sorted in global memory), bins of more than 32 particles (ranked by counting),
more such bins than the queue holds, several arrays in one merged order.  The
tables that fall out of it are compared with a host restatement."""
import ctypes as C

import numpy as np
import pytest

from pysph_amd.particle_array import get_particle_array

pytestmark = pytest.mark.gpu

NSUB = 8


def _ctx():
    from pysph_amd import device as dev
    return dev.HipContext(0)


def host_fine_keys(pa, nn):
    """csrc/sph_nnps.hip fine_key on the host (same arithmetic, fp64)"""
    xmin, cs, nc = nn.xmin, nn.cell_size, nn.ncells_per_dim
    # cell_coord(): ONE multiplication by the reciprocal of a binning cell 4 ulp wider than the reference's (round 6; the
    # reference divides by cell_size, nnps_base.pxd:39-57 -- its own cell ids are what the grid attributes report)
    inv = (1.0 / cs) * (1.0 - 4.0 * np.finfo(np.float64).eps)
    ux = (np.asarray(pa.x) - xmin[0]) * inv
    cx = np.floor(ux).astype(np.int64)
    cy = np.floor((np.asarray(pa.y) - xmin[1]) * inv).astype(np.int64)
    cz = np.floor((np.asarray(pa.z) - xmin[2]) * inv).astype(np.int64)
    sub = np.floor((ux - cx) * NSUB).astype(np.int64)
    lo, hi = cx < 0, cx > nc[0] - 1
    cx = np.clip(cx, 0, nc[0] - 1)
    sub = np.where(lo, 0, np.where(hi, NSUB - 1, np.clip(sub, 0, NSUB - 1)))
    cy = np.clip(cy, 0, nc[1] - 1)
    cz = np.clip(cz, 0, nc[2] - 1)
    return (cx + nc[0] * (cy + nc[1] * cz)) * NSUB + sub


def check_order(nn, arrays):
    for i, pa in enumerate(arrays):
        n = pa.get_number_of_particles()
        if n == 0:
            continue
        keys = host_fine_keys(pa, nn)
        want = np.argsort(keys, kind='stable')
        got = nn.get_spatially_ordered_indices(i)
        assert np.array_equal(got, want.astype(np.uint32)), (pa.name, n)


def cloud(n, seed, name='a', box=1.0, h=0.05):
    rng = np.random.default_rng(seed)
    x, y, z = rng.random((3, n)) * box
    return get_particle_array(name=name, x=x, y=y, z=z, h=h * np.ones(n), m=np.ones(n))


@pytest.mark.parametrize('n,box,h', [(1, 1.0, 0.1), (63, 1.0, 0.1), (5000, 1.0, 0.05), (200000, 1.0, 0.01),
                                     (30000, 0.2, 0.05),     # a handful of cells: one bucket far beyond the LDS stage
                                     (300000, 4.0, 0.011)],  # many buckets, sparse keys
                         ids=['one', 'wave-1', '5k', '200k', 'one-bucket', 'sparse'])
def test_sorted_order_is_the_stable_order(n, box, h):
    from pysph_amd.nnps import HipNNPS
    pa = cloud(n, n, box=box, h=h)
    ctx = _ctx()
    nn = HipNNPS(3, [pa], radius_scale=2.0, ctx=ctx)
    check_order(nn, [pa])
    ctx.close()


def test_dense_bins_and_coincident_particles():
    """one bin of 3000 coincident particles (ranked by counting), 150 bins of 40
    (more than the queue of big bins holds: the rest by insertion), on top of a cloud"""
    from pysph_amd.nnps import HipNNPS
    rng = np.random.default_rng(11)
    base = cloud(20000, 1, h=0.03)
    xs, ys, zs = [np.asarray(base.x)], [np.asarray(base.y)], [np.asarray(base.z)]
    xs.append(np.full(3000, 0.5)); ys.append(np.full(3000, 0.5)); zs.append(np.full(3000, 0.5))
    for _ in range(150):
        p = rng.random(3) * 0.05 + 0.2      # 150 piles inside one or two cells
        xs.append(np.full(40, p[0])); ys.append(np.full(40, p[1])); zs.append(np.full(40, p[2]))
    x, y, z = np.concatenate(xs), np.concatenate(ys), np.concatenate(zs)
    order = rng.permutation(x.size)          # the piles scattered over the index range
    pa = get_particle_array(name='a', x=x[order], y=y[order], z=z[order], h=0.03 * np.ones(x.size))
    for lbits in (0, 9, 11):         # (11: the piles' bucket is split into sub-ranges; the 3000-pile goes to global memory)
        ctx = _ctx()
        ctx.set_option('sort_lbits', lbits)
        nn = HipNNPS(3, [pa], radius_scale=2.0, ctx=ctx)
        check_order(nn, [pa])
        ctx.close()


@pytest.mark.parametrize('merge', [1, 0])
def test_several_arrays_one_empty(merge):
    from pysph_amd.nnps import HipNNPS
    arrays = [cloud(40000, 2, 'fluid', h=0.02), get_particle_array(name='none', x=np.zeros(0)),
              cloud(9000, 3, 'solid', h=0.02), cloud(100, 4, 'obstacle', box=0.3, h=0.02)]
    ctx = _ctx()
    ctx.set_option('merge_arrays', merge)
    nn = HipNNPS(3, arrays, radius_scale=2.0, ctx=ctx)
    check_order(nn, arrays)
    # ... and the neighbour lists that come out of the tables, against brute force for a few particles
    for (s, d) in ((0, 2), (2, 0), (3, 0), (0, 0)):
        start, nbrs = nn.get_csr(s, d)
        src, dst = arrays[s], arrays[d]
        for i in (0, dst.get_number_of_particles() // 2, dst.get_number_of_particles() - 1):
            r2 = (src.x - dst.x[i]) ** 2 + (src.y - dst.y[i]) ** 2 + (src.z - dst.z[i]) ** 2
            want = np.nonzero((r2 < (2.0 * dst.h[i]) ** 2) | (r2 < (2.0 * src.h) ** 2))[0]
            assert np.array_equal(nbrs[start[i]:start[i + 1]], want), (s, d, i)


def _moved(pa, seed, amp, on_device=False):
    """(on_device: in place in device memory, as a stage kernel moves particles; the host copy follows)"""
    from helpers import device_add
    rng = np.random.default_rng(seed)
    for f in 'xyz':
        d = amp * (rng.random(pa.x.size) - 0.5)
        if on_device:
            device_add(pa, f, d)
        else:
            getattr(pa, f)[:] = getattr(pa, f) + d


@pytest.mark.parametrize('narr', [1, 3], ids=['one-array', 'merged'])
def test_update_without_round_trip_reports_the_exact_grid_and_the_same_neighbours(narr):
    """Device-resident positions move between updates; h and m are not written.
    From the second update on no update waits for the device (`n_async`), the
    grid it REPORTS is the reference's for the new positions (equal to a fresh
    neighbour search's), and the neighbour lists are the same sets -- also for
    particles that left the box of the previous update (clamped into the
    outermost cells)."""
    from pysph_amd import device as dev
    from pysph_amd.nnps import HipNNPS
    arrays = [cloud(30000, 5, 'fluid', h=0.03), cloud(5000, 6, 'solid', h=0.03),
              cloud(50, 7, 'obstacle', box=0.2, h=0.03)][:narr]
    ctx = _ctx()
    ctx.timer_enable(True)
    for a in arrays:
        dev.attach(a, ctx).push()
    nn = HipNNPS(3, arrays, radius_scale=2.0, ctx=ctx, sync=False)
    for step in range(3):
        for a in arrays:
            _moved(a, 100 + step, 0.02 if step < 2 else 0.3, on_device=True)     # the last move throws particles far outside
        nn.update()
        fresh_ctx = _ctx()
        copies = [get_particle_array(name=a.name, x=a.x.copy(), y=a.y.copy(), z=a.z.copy(), h=a.h.copy(), m=a.m.copy())
                  for a in arrays]
        fresh = HipNNPS(3, copies, radius_scale=2.0, ctx=fresh_ctx)       # host data pushed, looked at: the exact grid
        assert nn.cell_size == fresh.cell_size and nn.n_cells == fresh.n_cells
        assert np.array_equal(nn.xmin, fresh.xmin) and np.array_equal(nn.xmax, fresh.xmax)
        assert np.array_equal(nn.ncells_per_dim, fresh.ncells_per_dim)
        for (s, d) in ((0, 0), (narr - 1, 0), (0, narr - 1)):
            a0, a1 = nn.get_csr(s, d)
            b0, b1 = fresh.get_csr(s, d)
            assert np.array_equal(a0, b0) and np.array_equal(a1, b1), (step, s, d)
        fresh_ctx.close()
    assert ctx.timer_get('n_async')[1] == 3
    # a write of h: the next update looks again (and finds the new cell size)
    arrays[0].h[:] = 0.04
    arrays[0].gpu.push('h')
    nn.update()
    assert ctx.timer_get('n_async')[1] == 3
    assert nn.cell_size == 2.0 * 0.04
    nn.update()
    assert ctx.timer_get('n_async')[1] == 4
    ctx.close()


def test_evaluation_after_particles_left_the_box_of_the_previous_update_vs_oracle(oracle):
    """the update that bins on the PREVIOUS update's bounds (no round trip) clamps particles that left that box into its
    outermost cells: the evaluation on that grid equals the oracle's on the reference's own grid -- every field and the
    neighbour count of every particle (the lagged path had only been compared with a fresh HipNNPS)."""
    from helpers import rel_err
    from pysph_amd import device as dev
    from pysph_amd import kernels as K
    from pysph_amd.acceleration_eval import AccelerationEval, SPHCompiler
    from pysph_amd.nnps import HipNNPS
    from test_hip_parity import cube_equations, make_cube
    pa, dx = make_cube(22)
    eqs = cube_equations(dx)
    kernel = K.WendlandQuintic(dim=3)
    ctx = _ctx()
    ctx.timer_enable(True)
    dev.attach(pa, ctx).push()
    a_eval = AccelerationEval([pa], eqs, kernel)
    SPHCompiler(a_eval, ctx=ctx, sync='manual').compile()
    nn = HipNNPS(3, [pa], radius_scale=2.0, ctx=ctx, sync=False)
    a_eval.set_nnps(nn)
    nn.update()
    a_eval.compute(0.0, 1e-5)
    for step, amp in enumerate((0.01, 0.3, 0.02)):           # the second move throws particles far outside the old box
        _moved(pa, 300 + step, amp, on_device=True)
        nn.update()
        a_eval.compute(0.0, 1e-5)
        ref = pa.extract_particles(np.arange(pa.get_number_of_particles()), name='fluid')
        onn = oracle.OracleNNPS(3, [ref], 2.0)
        onn.update()
        oev = oracle.OracleEval([ref], eqs, kernel, nthreads=4)
        oev.set_nnps(onn)
        oev.compute(0.0, 1e-5)
        pa.gpu.pull('arho', 'au', 'av', 'aw', 'ax', 'ay', 'az', 'p', 'cs', 'dt_cfl')
        for f in ('arho', 'au', 'av', 'aw', 'ax', 'ay', 'az', 'p', 'cs', 'dt_cfl'):
            assert rel_err(pa.get(f), ref.get(f)) < 1e-10, (step, f)
        start = nn.get_csr_start(0, 0)
        want = np.array([len(onn.get_nearest_particles(0, 0, i)) for i in range(0, ref.get_number_of_particles(), 97)])
        assert np.array_equal(np.diff(start.astype(np.int64))[::97], want), step
    assert ctx.timer_get('n_async')[1] == 3                  # none of the three updates looked at the particles first
    ctx.close()


def test_too_many_cells_after_an_update_without_round_trip_is_raised_by_the_next_update():
    """linked_list_nnps.pyx:307-343 raises inside update() when the grid needs more than 2^28 cells.  An update that bins on
    the previous update's bounds learns its own bounds late: the error comes when the grid attributes are read or --
    nobody reads them in a stepping loop -- from the NEXT update, which would have to bin on that grid (never silently on
    a clamped one for more than the one update)."""
    from pysph_amd import device as dev
    from pysph_amd.nnps import HipNNPS
    pa = cloud(20000, 9, h=0.002)                # cells of 0.004: a unit box has 250^3 = 1.6e7 of them
    ctx = _ctx()
    ctx.timer_enable(True)
    dev.attach(pa, ctx).push()
    nn = HipNNPS(3, [pa], radius_scale=2.0, ctx=ctx, sync=False)
    nn.update()
    _moved(pa, 1, 0.001, on_device=True)
    nn.update()
    from helpers import device_add
    far = np.zeros(pa.x.size)
    far[:10] = 5.0                               # ten particles leave: a 6 x 6 x 6 box of 0.004 cells is 3.4e9 cells
    for f in 'xyz':
        device_add(pa, f, far)
    nn.update()                                  # bins on the old box (the ten are clamped into its outermost cells) ...
    assert ctx.timer_get('n_async')[1] == 3      # (the constructor made the first, looked-at update)
    with pytest.raises(RuntimeError, match='too many cells'):
        nn.update()                              # ... and the next update reports what that one found
    ctx.close()


def test_array_in_no_spatial_order_is_visited_in_the_previous_cell_order():
    """an array that lies in memory in random order makes every lane of the key pass its own atomic group; from the third
    update on (the figure travels with the bounds) the passes visit it through the previous update's cell order instead.
    Same sorted order either way (ascending index inside a key), also after the particles moved."""
    from pysph_amd import device as dev
    from pysph_amd.nnps import HipNNPS
    pa = cloud(150000, 21, h=0.012)             # random positions in index order = no spatial order
    orders = {}
    for via in (1, 0):
        ctx = _ctx()
        ctx.set_option('via_unordered', via)
        dev.attach(pa, ctx).push()
        nn = HipNNPS(3, [pa], radius_scale=2.0, ctx=ctx, sync=False)
        got = []
        for step in range(5):
            nn.update()
            check_order(nn, [pa])
            got.append(nn.get_spatially_ordered_indices(0).copy())
        orders[via] = got
        ctx.close()
    for a, b in zip(orders[0], orders[1]):
        assert np.array_equal(a, b)


def test_async_update_can_be_switched_off():
    from pysph_amd import device as dev
    from pysph_amd.nnps import HipNNPS
    pa = cloud(5000, 8)
    ctx = _ctx()
    ctx.set_option('async_update', 0)
    ctx.timer_enable(True)
    dev.attach(pa, ctx).push()
    nn = HipNNPS(3, [pa], radius_scale=2.0, ctx=ctx, sync=False)
    nn.update()
    nn.update()
    assert ctx.timer_get('n_async')[1] == 0
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize('frac', [0.0005, 0.02, 0.2, 0.6], ids=['a-few', '2-percent', 'a-fifth', 'most'])
def test_removal_of_the_leavers_moves_tail_rows_into_the_holes(frac):
    """sph_halo_remove_selected (parallel_manager.pyx:1085-1157: the exported particles leave): with few leavers the
    kept rows of the tail move into the holes (option fill_holes, the default; what the reference's own removal does
    with the ends of its property arrays), with many the stable compaction runs -- either way exactly the kept
    particles remain, every property of a row with it; fill_holes = 0 keeps their order"""
    import torch
    from pysph_amd import device as dev
    from pysph_amd.parallel import DeviceHaloOps, WCSPH_HALO_PROPS
    from pysph_amd.particle_array import get_particle_array_wcsph
    rng = np.random.default_rng(5)
    n = 20000
    x = rng.uniform(0.0, 1.0, n)
    lo, hi = frac / 2, 1.0 - frac / 2

    def run(fill):
        pa = get_particle_array_wcsph(name='fluid', x=x.copy(), y=rng.uniform(size=n), z=np.zeros(n), h=0.01 * np.ones(n),
                                      m=np.ones(n), rho=np.arange(n, dtype=np.float64))
        pa.u[:] = 2.0 * np.arange(n)
        ctx = dev.HipContext(0, torch.cuda.current_stream().cuda_stream)
        ctx.set_option('fill_holes', fill)
        ops = DeviceHaloOps(pa, ctx, WCSPH_HALO_PROPS, 0)
        ops.gpu.push()
        n_lo, n_hi = ops.select(lo, hi)
        assert n_lo == int((x < lo).sum()) and n_hi == int((x >= hi).sum())
        left = ops.remove_selected()
        pa.gpu.sync_host()                                      # (the host copy follows the device's row count)
        nr = pa.get_number_of_particles(True)
        out = dict((k, np.array(pa.get(k)[:nr])) for k in ('x', 'rho', 'u'))
        ctx.close()
        return left, out

    keep = (x >= lo) & (x < hi)
    for fill in (1, 0):
        left, out = run(fill)
        assert left == int(keep.sum()) == out['x'].size
        ids = out['rho'].astype(np.int64)                       # rho carries the original index
        assert np.array_equal(np.sort(ids), np.nonzero(keep)[0])
        assert np.array_equal(out['x'], x[ids]) and np.array_equal(out['u'], 2.0 * ids)
        if fill == 0 or frac >= 0.25:
            assert np.array_equal(ids, np.nonzero(keep)[0])     # stable
        else:
            moved = int((ids != np.arange(left)).sum())
            assert 0 < moved <= n - left                         # only holes were filled, every other row stayed where it was


@pytest.mark.gpu
def test_migrants_under_a_promise_keep_the_update_without_round_trip():
    """sph_halo_append_promised (the arrivals of a migration, parallel_manager.pyx:1085-1157, in a run whose ranks
    agreed on ONE h and m per array): the neighbour update after it still bins on the previous update's bounds (a plain
    append makes it look at h and m first), and an arrival that carries another h (or, where the library holds the
    array's one mass, another m) sets bit 1 of the flag word"""
    import torch
    from pysph_amd import device as dev
    from pysph_amd.nnps import HipNNPS
    from pysph_amd.parallel import DeviceHaloOps, WCSPH_HALO_PROPS
    pa = cloud(20000, 11, h=0.02)
    pa.m[:] = 0.5
    ctx = _ctx()
    ctx.timer_enable(True)
    ops = DeviceHaloOps(pa, ctx, WCSPH_HALO_PROPS, 0)
    ops.gpu.push()
    ops.set_promise(0.02, 0.5)
    nn = HipNNPS(3, [pa], radius_scale=2.0, ctx=ctx, sync=False)
    nn.update()
    nn.update()
    base = ctx.timer_get('n_async')[1]
    ids = list(ops.all_props())
    kx, kh, km = [ids.index(dev.prop_id(p)) for p in ('x', 'h', 'm')]

    def arrivals(count, h):
        buf = torch.zeros(len(ids) * count, dtype=torch.float64, device='cuda')
        rows = buf.view(len(ids), count)
        rows[kx] = torch.linspace(0.1, 0.9, count, dtype=torch.float64)
        rows[kh] = h
        rows[km] = 0.5
        return buf

    flags = ops.flag_words(1)
    ops.append_real(arrivals(50, 0.02), 50, flags, 0)         # the promised h: what the library knows survives
    nn.update()
    assert ctx.timer_get('n_async')[1] == base + 1
    assert int(flags[0].item()) == 0
    ops.append_real(arrivals(50, 0.02), 50)                    # the plain append: h and m are unknown again
    nn.update()
    assert ctx.timer_get('n_async')[1] == base + 1
    nn.update()
    assert ctx.timer_get('n_async')[1] == base + 2
    ops.append_real(arrivals(7, 0.03), 7, flags, 0)            # arrivals with another smoothing length
    torch.cuda.synchronize()
    assert int(flags[0].item()) & 2
    assert pa.gpu.get_number_of_particles(True) == 20000 + 107
    ctx.close()
