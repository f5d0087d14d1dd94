"""The schedule options of round 3 change speed, never results:

* ``eos_fuse``  -- WCSPH pair kernel on 64-byte records with p, cs recomputed
  from rho (taken when the group list promises the Tait EOS, sph_group.src_eos);
* ``nl_reuse``  -- the second pair pass of an evaluation (TVF force, elastic
  rates) starts from the hit lists the first pass kept (sph_group.nl_mode);
  off by default: measured 2-6 % slower;
* ``norm_masks`` -- a row's hit bits shifted down to the lane's first hit;
* ``mass_fuse`` -- EOS-fused records whose mass slot carries p / rho^2 when every
  source array has ONE mass (seen by the neighbour update's reduction);
* ``merge_arrays`` -- a WCSPH group over several arrays (a dam break) as ONE launch
  over the merged cell order of all arrays, the per-source equation difference as
  a class bit in the record (round 4);
* ``row_mod3``  -- the order in which a wavefront visits its 3x3 rows of cells
  (the sums of a destination are taken in that order: equal to rounding).

Each is compared ON against OFF (to rounding where the arithmetic is regrouped,
bit for bit where only the schedule differs) and against the oracle, and the
launch counters prove the path under test really ran."""
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

pytestmark = pytest.mark.gpu


def _run(argv, opts, steps=2):
    """one bench workload through the C-ABI with the given library options;
    returns ({field: array}, counters, neighbour counts)"""
    import torch
    import bench
    from pysph_amd import device as dev
    args = bench.parse_args(argv + ['--no-cpu-baseline', '--no-extras'])
    ctx = dev.HipContext(0, torch.cuda.current_stream().cuda_stream)
    bench.apply_options(args, ctx)
    for k, v in opts.items():
        ctx.set_option(k, v)
    w = bench.build_workload(args, 0, 1)
    nnps, a_eval, halo, domain, step, ordered = bench.setup(args, w, 0, 1, None, ctx)
    for a in w.arrays:
        a.gpu.pull()
    host_in = bench.copy_arrays(w.arrays)
    for _ in range(steps):
        step()
    out = {}
    for pa in w.arrays:
        nreal = pa.get_number_of_particles(True)
        pa.gpu.pull(*[f for f in w.fields if f in pa.properties])
        for f in w.fields:
            if f in pa.properties:
                out[pa.name + '.' + f] = np.array(pa.get(f)[:nreal])
    cnt = {k: ctx.timer_get(k)[1] for k in ('n_eos_fused', 'n_nl_keep', 'n_nl_reuse', 'n_mass_fused', 'n_merged', 'n_tension_flag',
                                            'n_dest_list', 'n_row_lds')}
    res = bench.parity_check(w, host_in, nnps, domain,
                             bench.PARITY_TOL if args.dtype == 'f64' else 5e-5)
    del nnps, a_eval, step
    ctx.close()
    torch.cuda.empty_cache()
    return out, cnt, res


def _max_rel(a, b):
    """as bench.parity_check: the components of one vector share their scale (a
    component that cancels to nothing, the forces of a lattice, has none)"""
    import bench
    worst = 0.0
    for k in a:
        arr, f = k.rsplit('.', 1)
        group = [b[arr + '.' + g] for g in bench._scale_group(f) if arr + '.' + g in b]
        worst = max(worst, bench.field_error(a[k], b[k], group))
    return worst


@pytest.mark.parametrize('argv', [['--n1', '64'], ['--n1', '64', '--no-reorder'],
                                  ['--workload', 'dam_break', '--dx', '0.03']],
                         ids=['cube', 'cube-unsorted', 'dam-break'])
def test_eos_fused_records_match_gathered_records(argv):
    on, c_on, r_on = _run(argv, {})
    off, c_off, r_off = _run(argv, {'eos_fuse': 0})
    assert c_on['n_eos_fused'] > 0 and c_off['n_eos_fused'] == 0
    assert r_on['parity_ok'] and r_off['parity_ok'], (r_on, r_off)
    assert _max_rel(on, off) < 1e-13


@pytest.mark.parametrize('gamma', [1.0, 3.0, 5.0])
@pytest.mark.parametrize('argv', [['--n1', '48'], ['--n1', '48', '--dtype', 'f32'], ['--workload', 'dam_break', '--dx', '0.03']],
                         ids=['cube', 'cube-fp32', 'dam-break'])
def test_eos_fused_records_with_other_odd_exponents(argv, gamma):
    """TaitEOS with gamma = 1, 3, 5 (wc/basic.py:60-65): the powers by multiplication like 7's, in k_nosrc and in the
    per-destination fused decoders (FamWCSPHEG_T, with and without the uniform-mass slot) alike: they agree with the
    gathered p / cs and with the oracle's pow().  The merged one-launch kernel and the variable-h records fold the
    exponent 7 and are not taken."""
    argv = argv + ['--gamma', repr(gamma)]
    on, c_on, r_on = _run(argv, {}, steps=3)
    off, c_off, r_off = _run(argv, {'eos_fuse': 0}, steps=3)
    assert c_on['n_eos_fused'] > 0 and c_off['n_eos_fused'] == 0, (c_on, c_off)
    assert c_on['n_merged'] == 0, c_on
    if '--dtype' in argv:
        assert r_on['parity_max_rel'] < 5e-5 and r_off['parity_max_rel'] < 5e-5, (r_on, r_off)
        assert r_on['parity_neighbour_count_mismatches'] == 0
    else:
        assert r_on['parity_ok'] and r_off['parity_ok'], (r_on, r_off)
        assert _max_rel(on, off) < 1e-13


def test_variable_h_records_fuse_the_exponent_seven_only():
    out, cnt, res = _run(['--n1', '48', '--vary-h', '0.15', '--gamma', '3.0'], {}, steps=3)
    assert cnt['n_eos_fused'] == 0 and res['parity_ok'], (cnt, res)
    out, cnt, res = _run(['--n1', '48', '--vary-h', '0.15'], {}, steps=3)
    assert cnt['n_eos_fused'] > 0 and res['parity_ok'], (cnt, res)


def test_eos_is_not_fused_for_an_exponent_without_integer_powers():
    out, cnt, res = _run(['--n1', '48', '--gamma', '1.4'], {}, steps=3)
    assert cnt['n_eos_fused'] == 0 and res['parity_ok'], (cnt, res)


def test_eos_fused_records_fp32():
    on, c_on, r_on = _run(['--n1', '64', '--dtype', 'f32'], {})
    off, c_off, r_off = _run(['--n1', '64', '--dtype', 'f32'], {'eos_fuse': 0})
    assert c_on['n_eos_fused'] > 0 and c_off['n_eos_fused'] == 0
    assert r_on['parity_max_rel'] < 5e-5 and r_off['parity_max_rel'] < 5e-5
    assert r_on['parity_neighbour_count_mismatches'] == 0


def test_eos_not_fused_when_h_varies_and_masses_are_not_fused():
    """the promise is there, the conditions of the 64-byte layouts are not (variable h needs
    the one-mass-per-array records, switched off here)"""
    out, cnt, res = _run(['--n1', '48', '--vary-h', '0.1'], {'mass_fuse': 0})
    assert cnt['n_eos_fused'] == 0 and res['parity_ok']


@pytest.mark.parametrize('argv', [['--n1', '64', '--vary-h', '0.15'], ['--n1', '64', '--vary-h', '0.15', '--no-reorder'],
                                  ['--n1', '64', '--vary-h', '0.15', '--dtype', 'f32']],
                         ids=['variable-h', 'variable-h-unsorted', 'variable-h-fp32'])
def test_variable_h_on_64_byte_records(argv):
    """round 4: variable h with ONE mass per array -- [x y z h u v w rho], p, cs, p / rho^2 recomputed
    from rho, m a constant of the source -- against the 96-byte records and the oracle"""
    on, c_on, r_on = _run(argv, {}, steps=3)
    off, c_off, r_off = _run(argv, {'mass_fuse': 0}, steps=3)
    assert c_on['n_eos_fused'] == 2 and c_on['n_mass_fused'] == 2 and c_off['n_eos_fused'] == 0, (c_on, c_off)
    assert r_on['parity_neighbour_count_mismatches'] == 0 and r_off['parity_neighbour_count_mismatches'] == 0
    if '--dtype' in argv:
        assert r_on['parity_max_rel'] < 5e-5 and r_off['parity_max_rel'] < 5e-5, (r_on, r_off)
        assert _max_rel(on, off) < 5e-5
    else:
        assert r_on['parity_ok'] and r_off['parity_ok'], (r_on, r_off)
        assert _max_rel(on, off) < 1e-13


@pytest.mark.parametrize('argv', [['--workload', 'taylor_green', '--n1', '48'],
                                  ['--workload', 'elastic', '--rings-dx', '1.6e-3'],
                                  ['--workload', 'elastic', '--rings-dx', '1.6e-3', '--rings-spacing', '0.0416'],
                                  ['--workload', 'elastic_block', '--n1', '40'],
                                  ['--workload', 'elastic', '--rings-dx', '1.6e-3', '--dtype', 'f32']],
                         ids=['taylor-green', 'rings', 'rings-in-contact', 'block', 'rings-fp32'])
def test_second_pass_on_kept_lists_is_bit_identical(argv):
    """only the schedule changes: same pairs, same order within a lane's list"""
    on, c_on, r_on = _run(argv, {'nl_reuse': 1})
    off, c_off, r_off = _run(argv, {})
    assert c_on['n_nl_keep'] > 0 and c_on['n_nl_reuse'] == c_on['n_nl_keep']
    assert c_off['n_nl_keep'] == 0 and c_off['n_nl_reuse'] == 0
    assert r_on['parity_ok'] or '--dtype' in argv, r_on
    for k in on:
        assert np.array_equal(on[k], off[k]), k


def test_kept_lists_are_dropped_by_a_neighbour_update():
    """the lists belong to ONE sph_nnps_update: the next evaluation's first
    pass keeps new ones, a second pass never starts from the previous step's"""
    on, cnt, res = _run(['--workload', 'taylor_green', '--n1', '40'], {'nl_reuse': 1}, steps=3)
    assert cnt['n_nl_keep'] == 3 and cnt['n_nl_reuse'] == 3 and res['parity_ok']


@pytest.mark.parametrize('argv', [['--workload', 'taylor_green', '--n1', '48'],
                                  ['--params', 'cube', '--n1', '64'],
                                  ['--workload', 'elastic', '--rings-dx', '1.6e-3']],
                         ids=['taylor-green', 'cube.py', 'rings'])
def test_normalised_masks_are_bit_identical(argv):
    on, _, r_on = _run(argv, {})
    off, _, r_off = _run(argv, {'norm_masks': 0})
    assert r_on['parity_ok'] and r_off['parity_ok']
    assert r_on['parity_neighbour_count_mismatches'] == 0
    for k in on:
        assert np.array_equal(on[k], off[k]), k




def test_wave_tiles_of_the_real_particles_are_bit_identical():
    """round 6 (option dest_list): a pair launch whose destinations are exactly the real particles takes its wave tiles
    from the list of real positions the neighbour update made -- 64 consecutive REAL particles per wavefront instead of 64
    consecutive positions of the cell order, of which a periodic box's images (here 40 % of the rows) are idle lanes.
    Which wavefront serves a destination changes; its candidates, their order and every sum do not: bit-identical.  The
    density pass (Group.real = False: every row is a destination) does not take the list."""
    argv = ['--workload', 'taylor_green', '--n1', '40']
    on, c_on, r_on = _run(argv, {'dest_list': 2})
    auto, c_auto, r_auto = _run(argv, {})
    off, c_off, r_off = _run(argv, {'dest_list': 0})
    assert r_on['parity_ok'] and r_off['parity_ok'] and r_on['parity_neighbour_count_mismatches'] == 0
    assert c_on['n_dest_list'] == 2 and c_auto['n_dest_list'] == 2 and c_off['n_dest_list'] == 0     # the force pass of two steps
    for k in on:
        assert np.array_equal(on[k], off[k]) and np.array_equal(auto[k], off[k]), k


def test_wave_tiles_of_the_real_particles_on_the_merged_order():
    """... and on the merged order of a dam break's three arrays: a slab rank (`--emulate-rank`: real particles + the ghost
    layers its two neighbours send, 18 % of the rows at this size) evaluated with the list forced on, left to the rule
    (>= 1/8 of the rows: on) and off -- bit-identical, checked against the oracle, ONE merged launch either way."""
    argv = ['--workload', 'dam_break', '--dx', '0.04', '--emulate-rank', '1/3']
    on, c_on, r_on = _run(argv, {'dest_list': 2}, steps=3)
    off, c_off, r_off = _run(argv, {'dest_list': 0}, steps=3)
    assert r_on['parity_ok'] and r_off['parity_ok'] and r_on['parity_neighbour_count_mismatches'] == 0
    assert c_on['n_merged'] == 2 and c_off['n_merged'] == 2       # (from the second evaluation on: the masses have been seen)
    assert c_on['n_dest_list'] >= 2 and c_off['n_dest_list'] == 0
    for k in on:
        assert np.array_equal(on[k], off[k]), k


@pytest.mark.parametrize('argv', [['--workload', 'taylor_green', '--n1', '48'],
                                  ['--n1', '64'],
                                  ['--workload', 'dam_break', '--dx', '0.03'],
                                  ['--workload', 'elastic', '--rings-dx', '1.6e-3']],
                         ids=['taylor-green', 'cube', 'dam-break', 'rings'])
def test_row_order_changes_the_summation_order_only(argv):
    """every order visits the same 9 rows: identical neighbour counts, results
    equal to rounding, each within the tolerance of the oracle"""
    ref, _, r_ref = _run(argv, {'row_mod3': 0})
    assert r_ref['parity_ok'] and r_ref['parity_neighbour_count_mismatches'] == 0
    for mode in (1, 2, 3, 4):
        out, _, res = _run(argv, {'row_mod3': mode})
        assert res['parity_ok'], (mode, res)
        assert res['parity_neighbour_count_mismatches'] == 0
        assert _max_rel(out, ref) < 2e-10, mode   # each is within 1e-10 of the oracle


def test_row_order_option_is_validated():
    import torch
    from pysph_amd import device as dev
    ctx = dev.HipContext(0, torch.cuda.current_stream().cuda_stream)
    with pytest.raises(dev.SphError):
        ctx.set_option('row_mod3', 5)
    ctx.close()


@pytest.mark.parametrize('argv', [['--n1', '64'], ['--n1', '64', '--no-reorder'], ['--n1', '64', '--dtype', 'f32'],
                                  ['--params', 'cube', '--n1', '48']],
                         ids=['cube', 'cube-unsorted', 'cube-fp32', 'cube.py'])
def test_uniform_mass_records_match_mass_carrying_records(argv):
    on, c_on, r_on = _run(argv, {})
    off, c_off, r_off = _run(argv, {'mass_fuse': 0})
    assert 0 < c_on['n_mass_fused'] < c_on['n_eos_fused']   # from the second neighbour update on
    assert c_off['n_mass_fused'] == 0 and c_off['n_eos_fused'] > 0
    tol = 5e-5 if '--dtype' in argv else bench_tol()
    assert r_on['parity_max_rel'] < tol and r_off['parity_max_rel'] < tol, (r_on, r_off)
    assert r_on['parity_neighbour_count_mismatches'] == 0
    assert _max_rel(on, off) < (1e-5 if '--dtype' in argv else 1e-13)


def bench_tol():
    import bench
    return bench.PARITY_TOL


def test_fixed_bounds_updates_still_learn_the_masses():
    """an update that needs no bounds (given) and no h range (known) still looks
    at the masses ONCE when an evaluation could use one mass per array, and keeps
    what it found while nothing writes m (round 5: DevArray::m_dirty) -- until
    round 4 such updates forgot the masses and ran on mass-carrying records"""
    out, cnt, res = _run(['--n1', '48', '--fixed-bounds'], {})
    assert cnt['n_mass_fused'] > 0 and res['parity_ok'], (cnt, res)
    assert res['parity_neighbour_count_mismatches'] == 0


@pytest.mark.parametrize('argv', [['--n1', '48', '--vary-m', '0.2'], ['--workload', 'dam_break', '--dx', '0.03']],
                         ids=['masses-differ', 'run-time-flags'])
def test_records_keep_the_mass_when_masses_differ_or_flags_are_not_constant(argv):
    # (the per-destination path: since round 4 a dam break's group runs on the merged order of its arrays,
    # where the flags are a compile-time class table -- tests above)
    out, cnt, res = _run(argv, {'merge_arrays': 0})
    assert cnt['n_eos_fused'] > 0 and cnt['n_mass_fused'] == 0 and res['parity_ok'], (cnt, res)
    assert res['parity_neighbour_count_mismatches'] == 0


def test_a_push_of_the_masses_forgets_their_uniformity():
    """the reduction of the next neighbour update looks again; until then the
    records carry the mass"""
    import torch
    import bench
    from pysph_amd import device as dev
    args = bench.parse_args(['--n1', '32', '--no-cpu-baseline', '--no-extras'])
    ctx = dev.HipContext(0, torch.cuda.current_stream().cuda_stream)
    w = bench.build_workload(args, 0, 1)
    nnps, a_eval, halo, domain, step, ordered = bench.setup(args, w, 0, 1, None, ctx)
    step()
    step()
    n1 = ctx.timer_get('n_mass_fused')[1]
    assert n1 > 0
    pa = w.arrays[0]
    pa.gpu.pull('m')
    m = np.array(pa.m)
    m[::2] *= 1.5
    pa.m[:] = m
    pa.gpu.push('m')                       # not uniform any more, and no neighbour update since
    a_eval.compute(0.0, 1e-5)
    assert ctx.timer_get('n_mass_fused')[1] == n1
    host_in = None
    for a in w.arrays:
        a.gpu.pull()
    host_in = bench.copy_arrays(w.arrays)
    step()                                 # the update's reduction sees two masses
    assert ctx.timer_get('n_mass_fused')[1] == n1
    res = bench.parity_check(w, host_in, nnps, domain, bench.PARITY_TOL)
    assert res['parity_ok'], res
    del nnps, a_eval, step
    ctx.close()


# ---------------------------------------------------------------------------
# round 4: several arrays as ONE record stream (sph_ctx::merged, FamWCSPHM_T)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize('argv', [['--workload', 'dam_break', '--dx', '0.03'],
                                  ['--workload', 'dam_break', '--dx', '0.03', '--no-reorder'],
                                  ['--workload', 'dam_break', '--dx', '0.02', '--dtype', 'f32']],
                         ids=['dam-break', 'dam-break-unsorted', 'dam-break-fp32'])
def test_merged_arrays_match_per_destination_path(argv):
    """fluid <- {fluid, boundary, obstacle} and walls <- fluid in one launch over the
    merged order against the per-destination launches: the same pairs (neighbour
    counts exact on both), sums associated differently (all arrays in cell order
    instead of source by source), so equal to rounding; both against the oracle."""
    f32 = '--dtype' in argv
    on, c_on, r_on = _run(argv, {}, steps=3)
    off, c_off, r_off = _run(argv, {'merge_arrays': 0}, steps=3)
    # the first evaluation has not seen the masses yet; from the second on ONE pair launch per evaluation (n_merged counts them)
    assert c_on['n_merged'] == 2 and c_off['n_merged'] == 0, (c_on, c_off)
    assert r_on['parity_neighbour_count_mismatches'] == 0 and r_off['parity_neighbour_count_mismatches'] == 0
    if f32:
        assert r_on['parity_max_rel'] < 5e-5 and r_off['parity_max_rel'] < 5e-5, (r_on, r_off)
        assert _max_rel(on, off) < 5e-5
    else:
        assert r_on['parity_ok'] and r_off['parity_ok'], (r_on, r_off)
        assert _max_rel(on, off) < 1e-13


def test_merged_arrays_with_variable_h_match_per_destination_path():
    """round 5: a dam break whose particles carry their own h runs its rate group as ONE launch too
    (FamWCSPHMV_T: records [x y | z h | u v | w +-rho], p / rho^2 and cs recomputed from the gathered rho)"""
    argv = ['--workload', 'dam_break', '--dx', '0.03', '--vary-h', '0.15']
    on, c_on, r_on = _run(argv, {}, steps=3)
    off, c_off, r_off = _run(argv, {'merge_arrays': 0}, steps=3)
    assert c_on['n_merged'] == 2 and c_off['n_merged'] == 0, (c_on, c_off)
    assert r_on['parity_ok'] and r_off['parity_ok'], (r_on, r_off)
    assert r_on['parity_neighbour_count_mismatches'] == 0
    assert _max_rel(on, off) < 1e-13
    on32, c32, r32 = _run(argv + ['--dtype', 'f32'], {}, steps=3)
    assert c32['n_merged'] == 2 and r32['parity_max_rel'] < 5e-5, (c32, r32)


def test_merged_arrays_not_taken_without_the_promises():
    """without the EOS promise (option eos_fuse 0) there are no 64-byte records and the per-destination path runs;
    the uniform-h specialisation switched off (option uniform_h 0) takes the variable-h merged family since round 5"""
    import bench
    args = ['--workload', 'dam_break', '--dx', '0.03']
    out, cnt, res = _run(args, {'eos_fuse': 0}, steps=3)
    assert cnt['n_merged'] == 0 and res['parity_ok'], (cnt, res)
    out, cnt, res = _run(args + ['--opt', 'uniform_h=0'], {}, steps=3)
    assert cnt['n_merged'] == 2 and res['parity_ok'], (cnt, res)


def test_merged_records_refuse_a_density_that_is_not_positive():
    """the merged records carry a particle's class in the SIGN of rho: a density <= 0 would change its class silently.
    k_pack_merged raises a device word, the next neighbour updates carry it to the host with their bounds, and the
    update that sees it fails loudly and switches the context to the per-destination path (round 5)"""
    import torch
    import bench
    from pysph_amd import device as dev
    args = bench.parse_args(['--workload', 'dam_break', '--dx', '0.03', '--no-cpu-baseline', '--no-extras'])
    ctx = dev.HipContext(0, torch.cuda.current_stream().cuda_stream)
    bench.apply_options(args, ctx)
    w = bench.build_workload(args, 0, 1)
    nnps, a_eval, halo, domain, step, ordered = bench.setup(args, w, 0, 1, None, ctx)
    for _ in range(3):
        step()
    n0 = ctx.timer_get('n_merged')[1]
    assert n0 == 2
    fluid = w.arrays[0]
    fluid.gpu.pull('rho')
    fluid.rho[7] = -fluid.rho[7]
    fluid.gpu.push('rho')
    with pytest.raises(dev.SphError, match='density <= 0'):
        for _ in range(4):          # the evaluation that packs it, then at most two updates until the word has arrived
            step()
    n1 = ctx.timer_get('n_merged')[1]
    for _ in range(2):              # from now on the per-destination path
        step()
    assert ctx.timer_get('n_merged')[1] == n1
    del nnps, a_eval, step
    ctx.close()
    torch.cuda.empty_cache()


def test_merged_arrays_with_masses_per_class():
    """walls of another (uniform) mass than the fluid: the class carries its mass;
    walls whose masses differ among themselves: no uniform mass per class, the
    per-destination path runs -- same results either way"""
    import torch
    import bench
    from pysph_amd import device as dev

    def run(scale_boundary, scale_obstacle):
        args = bench.parse_args(['--workload', 'dam_break', '--dx', '0.03', '--no-cpu-baseline', '--no-extras'])
        ctx = dev.HipContext(0, torch.cuda.current_stream().cuda_stream)
        bench.apply_options(args, ctx)
        w = bench.build_workload(args, 0, 1)
        for a in w.arrays:
            if a.name == 'boundary':
                a.m[:] = a.m * scale_boundary
            if a.name == 'obstacle':
                a.m[:] = a.m * scale_obstacle
        nnps, a_eval, halo, domain, step, ordered = bench.setup(args, w, 0, 1, None, ctx)
        for a in w.arrays:
            a.gpu.pull()
        host_in = bench.copy_arrays(w.arrays)
        for _ in range(3):
            step()
        n_merged = ctx.timer_get('n_merged')[1]
        res = bench.parity_check(w, host_in, nnps, domain)
        del nnps, a_eval, step
        ctx.close()
        torch.cuda.empty_cache()
        return n_merged, res

    n, res = run(1.25, 1.25)
    assert n == 2 and res['parity_ok'], (n, res)
    n, res = run(1.25, 0.75)
    assert n == 0 and res['parity_ok'], (n, res)


@pytest.mark.parametrize('argv,tol', [(['--workload', 'taylor_green', '--n1', '48'], 1e-10),
                                      (['--workload', 'elastic', '--rings-dx', '1.6e-3'], 1e-13),
                                      (['--workload', 'elastic_block', '--n1', '40'], 1e-13),
                                      (['--workload', 'elastic', '--rings-dx', '1.6e-3', '--dtype', 'f32'], 5e-5)],
                         ids=['taylor-green', 'rings', 'block', 'rings-fp32'])
def test_state_fused_tvf_and_uniform_elastic_records(argv, tol):
    """TVF force records without p and V (recomputed from rho and the one mass), elastic
    rate records without h and m: against the full records (mass_fuse = 0) and the oracle.
    Taylor-Green's lattice sums cancel, so record-level rounding shows at the 1e-11 level."""
    on, c_on, r_on = _run(argv, {}, steps=3)
    off, c_off, r_off = _run(argv, {'mass_fuse': 0}, steps=3)
    assert c_on['n_mass_fused'] == 2 and c_off['n_mass_fused'] == 0, (c_on, c_off)
    assert r_on['parity_neighbour_count_mismatches'] == 0
    if '--dtype' in argv:
        assert r_on['parity_max_rel'] < 5e-5 and r_off['parity_max_rel'] < 5e-5, (r_on, r_off)
    else:
        assert r_on['parity_ok'] and r_off['parity_ok'], (r_on, r_off)
    assert _max_rel(on, off) < tol


@pytest.mark.parametrize('argv', [['--workload', 'elastic', '--rings-dx', '1.6e-3', '--rings-unperturbed'],
                                  ['--workload', 'elastic', '--rings-dx', '1.6e-3'],
                                  ['--workload', 'elastic', '--rings-dx', '1.6e-3', '--rings-unperturbed', '--dtype', 'f32']],
                         ids=['no-tension', 'tension', 'no-tension-fp32'])
def test_artificial_stress_is_gathered_only_under_tension(argv):
    """the rates kernel reads a device word the artificial-stress kernel sets when any r_ij is non-zero
    and skips the r_ij pieces of every record otherwise: bit-identical to always gathering them, with
    (perturbed rings) and without (rings.py's initial state) particles in tension"""
    on, c_on, r_on = _run(argv, {}, steps=3)
    off, c_off, r_off = _run(argv, {'tension_flag': 0}, steps=3)
    assert c_on['n_tension_flag'] == 2 and c_off['n_tension_flag'] == 0, (c_on, c_off)
    if '--dtype' in argv:
        assert r_on['parity_max_rel'] < 5e-5, r_on
    else:
        assert r_on['parity_ok'], r_on
    assert r_on['parity_neighbour_count_mismatches'] == 0
    for k in on:
        assert np.array_equal(on[k], off[k]), k
    if '--rings-unperturbed' in argv:
        assert not np.any(on['solid.r00']) and not np.any(on['solid.r12'])
    else:
        assert np.any(on['solid.r00'])


@pytest.mark.parametrize('argv,tol', [(['--workload', 'elastic', '--rings-dx', '1.6e-3'], 1e-12),
                                      (['--workload', 'elastic', '--rings-dx', '1.6e-3', '--rings-unperturbed'], 1e-12),
                                      (['--workload', 'elastic_block', '--n1', '40'], 1e-12),
                                      (['--workload', 'elastic', '--rings-dx', '1.6e-3', '--dtype', 'f32'], 5e-5)],
                         ids=['rings', 'rings-no-tension', 'block', 'rings-fp32'])
def test_row_tiles_of_source_records_in_lds(argv, tol):
    """option row_lds (k_pair_rowlds: the elastic rates evaluate a row tile's hits from an LDS copy of its records,
    tile after tile): the same pairs in another order -- against the default kernel and the oracle"""
    on, c_on, r_on = _run(argv, {'row_lds': 1}, steps=3)
    off, c_off, r_off = _run(argv, {}, steps=3)
    assert c_on['n_row_lds'] >= 2 and c_off['n_row_lds'] == 0, (c_on, c_off)
    assert r_on['parity_neighbour_count_mismatches'] == 0
    if '--dtype' in argv:
        assert r_on['parity_max_rel'] < 5e-5, r_on
    else:
        assert r_on['parity_ok'], r_on
    assert _max_rel(on, off) < tol


def test_merged_arrays_with_an_empty_array():
    """a dam break whose obstacle array holds no particle (the reference's with_obstacle=False keeps the
    equations but an application may also drain an array): the merged order skips it, results as on the
    per-destination path"""
    import torch
    import bench
    from pysph_amd import device as dev

    def run(opts):
        args = bench.parse_args(['--workload', 'dam_break', '--dx', '0.03', '--no-cpu-baseline', '--no-extras'])
        ctx = dev.HipContext(0, torch.cuda.current_stream().cuda_stream)
        bench.apply_options(args, ctx)
        for k, v in opts.items():
            ctx.set_option(k, v)
        w = bench.build_workload(args, 0, 1)
        k = [a.name for a in w.arrays].index('obstacle')
        w.arrays[k] = w.arrays[k].extract_particles(np.arange(0), name='obstacle')   # the array stays, without particles
        assert w.arrays[k].get_number_of_particles() == 0
        nnps, a_eval, halo, domain, step, ordered = bench.setup(args, w, 0, 1, None, ctx)
        for a in w.arrays:
            a.gpu.pull()
        host_in = bench.copy_arrays(w.arrays)
        for _ in range(3):
            step()
        n_merged = ctx.timer_get('n_merged')[1]
        out = {}
        for pa in w.arrays:
            n = pa.get_number_of_particles(True)
            if n:
                pa.gpu.pull('arho', 'au', 'av', 'aw', 'ax', 'ay', 'az')
                for f in ('arho', 'au', 'ax'):
                    out[pa.name + '.' + f] = np.array(pa.get(f)[:n])
        res = bench.parity_check(w, host_in, nnps, domain)
        del nnps, a_eval, step
        ctx.close()
        torch.cuda.empty_cache()
        return n_merged, out, res

    n_on, on, r_on = run({})
    n_off, off, r_off = run({'merge_arrays': 0})
    assert n_on == 2 and n_off == 0, (n_on, n_off)
    assert r_on['parity_ok'] and r_off['parity_ok'], (r_on, r_off)
    assert _max_rel(on, off) < 1e-13
