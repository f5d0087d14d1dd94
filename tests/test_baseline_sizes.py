"""Parity against the CPU oracle AT THE BASELINE SIZES (BASELINE.json configs):
the 159^3 = 4.02 M headline cube (spatially ordered and not, uniform and +-15 %
variable h, both parameter sets), the C2 dam break (dx 0.0087: 1.05 M fluid +
0.26 M boundary + 16 k obstacle, three arrays) and the 4 M periodic
Taylor-Green TVF case.  Every output field is compared (max|a-b| / max|b| <
1e-10, BASELINE.json north_star) and, for the non-periodic cases, the neighbour
COUNT of every destination exactly -- index arithmetic (32-bit offsets, tile
order, slot overflow, long rows) only shows up at size.

Same code path as ``bench.py``'s own post-run check (``parity_check``): the
oracle (oracle/sph_oracle.c, OpenMP) needs 3-6 s per case on the GPU box's
host cores."""
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

pytestmark = pytest.mark.gpu

# Element-wise figure (bench.field_error): max_i |a_i - b_i| / max(|b_i|, 1e-6 max|b|).
# A pair sum evaluated in another order already differs from itself by ~1e-16 of the
# field's scale per particle, i.e. by up to 1e-16 / 1e-6 = 1e-10 (times the handful
# of ulps a 80-term sum collects) where |b_i| sits at the floor: the reference's own
# restatement against itself in a permuted particle order gives 3.5e-11 at 216 k
# particles (tests/test_bench_launcher.py::test_elementwise_error_of_a_reordered_sum).
ELEMENTWISE_BOUND = 1e-8
# ... and where the sums cancel to zero (the Taylor-Green lattice) every particle sits
# at the floor: the figure is the norm-wise error / floor, bounded by 1e-10 / 1e-6
# (bench.ELEMENTWISE_TOL_CANCELLING); parity_check asserts each workload's own bound.


def _case(argv):
    import torch
    import bench
    from pysph_amd import device as dev
    args = bench.parse_args(argv + ['--no-cpu-baseline', '--no-extras'])
    stream = torch.cuda.current_stream()
    ctx = dev.HipContext(0, stream.cuda_stream)
    bench.apply_options(args, ctx)
    w = bench.build_workload(args, 0, 1)
    nnps, a_eval, halo, domain, step, ordered = bench.setup(args, w, 0, 1, None, ctx)
    for a in w.arrays:
        a.gpu.pull()
    host_in = bench.copy_arrays(w.arrays)
    step()
    step()          # the second evaluation runs on buffers the first one sized
    res = bench.parity_check(w, host_in, nnps, domain)
    res['n_merged'] = ctx.timer_get('n_merged')[1]
    n = sum(a.get_number_of_particles() for a in w.arrays)
    del nnps, a_eval, step
    ctx.close()
    torch.cuda.empty_cache()
    return res, n, ordered


@pytest.mark.parametrize('extra', [
    [],
    ['--no-reorder'],
    ['--vary-h', '0.15'],
    ['--vary-h', '0.15', '--no-reorder'],
    ['--params', 'cube'],
], ids=['sorted-uniform-h', 'unsorted-uniform-h', 'sorted-variable-h', 'unsorted-variable-h',
        'cube.py-parameters'])
def test_cube_4m_vs_oracle(extra):
    res, n, ordered = _case(['--n1', '159'] + extra)
    assert n == 159 ** 3
    assert ordered == ('--no-reorder' not in extra)
    assert res['parity_neighbour_count_mismatches'] == 0, res
    assert res['parity_max_rel'] < 1e-10, res
    assert res['parity_elementwise_max_rel'] < ELEMENTWISE_BOUND, res


def test_dam_break_c2_vs_oracle():
    """BASELINE config 2: three arrays, fluid <- {fluid, boundary, obstacle},
    solids <- fluid, sparse boundary shells (long empty rows in the cell tables)."""
    res, n, _ = _case(['--workload', 'dam_break', '--dx', '0.0087'])
    assert n > 1.3e6
    assert res['parity_neighbour_count_mismatches'] == 0, res
    assert res['parity_max_rel'] < 1e-10, res
    assert res['parity_elementwise_max_rel'] < ELEMENTWISE_BOUND, res


def test_dam_break_4m_vs_oracle():
    """The size BASELINE's metric names: dx 0.0055 = 4.06 M fluid + 0.53 M
    boundary + 64 k obstacle."""
    res, n, _ = _case(['--workload', 'dam_break', '--dx', '0.0055'])
    assert n > 4.6e6
    assert res['parity_neighbour_count_mismatches'] == 0, res
    assert res['parity_max_rel'] < 1e-10, res
    assert res['parity_elementwise_max_rel'] < ELEMENTWISE_BOUND, res


def test_dam_break_4m_variable_h_runs_merged_vs_oracle():
    """round 5: the same tank with every particle carrying its own h (+-15 %) still runs its rate group as ONE launch
    over the merged order (FamWCSPHMV_T) from the second evaluation on"""
    res, n, _ = _case(['--workload', 'dam_break', '--dx', '0.0055', '--vary-h', '0.15'])
    assert n > 4.6e6 and res['n_merged'] >= 1, res
    assert res['parity_neighbour_count_mismatches'] == 0, res
    assert res['parity_max_rel'] < 1e-10, res
    assert res['parity_elementwise_max_rel'] < ELEMENTWISE_BOUND and res['parity_ok'], res


def test_dam_break_16m_one_gpu_vs_oracle():
    """BASELINE config 4's workload (S-dam dx 0.0035) on ONE GPU -- the anchor
    its 8-GPU strong-scaling number is divided by: 15.8 M fluid + 1.3 M boundary +
    0.25 M obstacle on a ~4.6 M-cell grid (37 M fine_start entries per array, long
    empty rows, packed indices beyond 2^24)."""
    res, n, _ = _case(['--workload', 'dam_break', '--dx', '0.0035'])
    assert n > 16e6
    assert res['parity_neighbour_count_mismatches'] == 0, res
    assert res['parity_max_rel'] < 1e-10, res
    assert res['parity_elementwise_max_rel'] < ELEMENTWISE_BOUND and res['parity_ok'], res


def test_taylor_green_4m_periodic_vs_oracle():
    """BASELINE config 3: 159^3 periodic TVF; the oracle runs on the ghosts of
    the host DomainManager, the device on those of HipDomainManager."""
    res, n, _ = _case(['--workload', 'taylor_green', '--n1', '159'])
    assert n >= 159 ** 3          # + the device's periodic ghosts
    # every real destination's neighbour count, ghost images counted as sources
    assert res['parity_neighbour_count_mismatches'] == 0, res
    assert res['parity_max_rel'] < 1e-10, res
    import bench
    assert res['parity_elementwise_tolerance'] == bench.ELEMENTWISE_TOL_CANCELLING
    assert res['parity_elementwise_max_rel'] < bench.ELEMENTWISE_TOL_CANCELLING and res['parity_ok'], res


@pytest.mark.parametrize('extra', [[], ['--rings-spacing', '0.0405372']],
                         ids=['rings.py-spacing', 'in-contact'])
def test_rings_2m_vs_oracle(extra):
    """BASELINE config 5 as specified (S-rings3d): two hollow
    spheres, CubicSpline hdx 1.5 (113-neighbour rows), free surfaces; at the
    reference's spacing and with the two bodies one dx apart (cross-body pairs)."""
    res, n, _ = _case(['--workload', 'elastic'] + extra)
    assert 1.9e6 < n < 2.1e6
    assert res['parity_neighbour_count_mismatches'] == 0, res
    assert res['parity_max_rel'] < 1e-10, res
    # no systematic cancellation in the rings' fields (seeded perturbation: every term acts)
    assert res['parity_elementwise_max_rel'] < ELEMENTWISE_BOUND and res['parity_ok'], res


def test_rings_2m_fp32_vs_oracle():
    """BASELINE config 5 as named: S-rings3d, 2 M, fp32 arithmetic -- against
    the fp64 oracle at the fp32 tolerance."""
    res, n, _ = _case(['--workload', 'elastic', '--dtype', 'f32'])
    assert 1.9e6 < n < 2.1e6
    assert res['parity_neighbour_count_mismatches'] == 0, res
    assert 1e-9 < res['parity_max_rel'] < 5e-5, res


def test_elastic_2m_vs_oracle():
    """The elastic equation set on a solid block, 126^3 = 2.0 M (fp64 arithmetic)."""
    res, n, _ = _case(['--workload', 'elastic_block', '--n1', '126'])
    assert n == 126 ** 3
    assert res['parity_neighbour_count_mismatches'] == 0, res
    assert res['parity_max_rel'] < 1e-10, res
    assert res['parity_elementwise_max_rel'] < ELEMENTWISE_BOUND and res['parity_ok'], res


def test_elastic_2m_fp32_vs_oracle():
    """The same block in fp32 arithmetic -- against the fp64 oracle at the fp32
    tolerance (see tests/test_hip_parity.py::test_fp32_arithmetic_vs_golden)."""
    res, n, _ = _case(['--workload', 'elastic_block', '--n1', '126', '--dtype', 'f32'])
    assert n == 126 ** 3
    assert 1e-9 < res['parity_max_rel'] < 5e-5, res


def test_cube_4m_fp32_vs_oracle():
    res, n, _ = _case(['--n1', '159', '--dtype', 'f32'])
    assert 1e-9 < res['parity_max_rel'] < 5e-5, res
