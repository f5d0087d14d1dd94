"""Host-side structure checks of the reference's
pysph/sph/tests/test_acceleration_eval.py that need no device: property
checking of equations against the particle arrays (:46-135) and the MegaGroup
regrouping (:242-291)."""
import numpy as np
import pytest

from pysph_amd.acceleration_eval import (AccelerationEval, MegaGroup,
                                         check_equation_array_properties)
from pysph_amd.equations import Equation, Group, SummationDensity
from pysph_amd.particle_array import get_particle_array


class DummyEquation(Equation):                                  # :23-34
    def initialize(self, d_idx, d_rho, d_V):
        d_rho[d_idx] = d_V[d_idx]

    def loop(self, d_idx, d_rho, s_idx, s_m, s_u, WIJ):
        d_rho[d_idx] += s_m[s_idx] * WIJ

    def post_loop(self, d_idx, d_rho, s_idx, s_m, s_V):
        d_rho[d_idx] += s_m[d_idx]


class FindTotalMass(Equation):                                  # :37-45
    def initialize(self, d_idx, d_m, d_total_mass):
        d_total_mass[0] = 0.0

    def post_loop(self, d_idx, d_m, d_total_mass):
        d_total_mass[0] += d_m[d_idx]


class SimpleEquation(Equation):
    def initialize(self, d_idx, d_u, d_au):
        d_u[d_idx] = 0.0
        d_au[d_idx] = 0.0

    def loop(self, d_idx, d_au, s_idx, s_m):
        d_au[d_idx] += s_m[s_idx]


class MixedTypeEquation(Equation):
    def loop(self, d_idx, d_au, s_idx, s_m, s_pid, s_tag):
        d_au[d_idx] += s_m[s_idx] + s_pid[s_idx] + s_tag[s_idx]


def test_should_raise_runtime_error_when_invalid_dest_source():    # :47-69
    f = get_particle_array(name='f')
    with pytest.raises(RuntimeError):
        check_equation_array_properties(SummationDensity(dest='fluid', sources=['f']), [f])
    with pytest.raises(RuntimeError):
        check_equation_array_properties(SummationDensity(dest='f', sources=['fluid']), [f])


def test_should_pass_when_properties_exist():                      # :71-81
    f = get_particle_array(name='f')
    check_equation_array_properties(SummationDensity(dest='f', sources=['f']), [f])


def test_should_fail_when_props_dont_exist():                      # :83-92
    f = get_particle_array(name='f')
    with pytest.raises(RuntimeError):
        check_equation_array_properties(DummyEquation(dest='f', sources=['f']), [f])


def test_source_properties_are_checked_too():                      # :94-118
    f = get_particle_array(name='f')
    f.add_property('V')
    s = get_particle_array(name='s')
    eq = DummyEquation(dest='f', sources=['f', 's'])
    with pytest.raises(RuntimeError):
        check_equation_array_properties(eq, [f, s])
    s.add_property('V')
    check_equation_array_properties(eq, [f, s])


def test_should_check_constants():                                 # :120-135
    f = get_particle_array(name='f')
    eq = FindTotalMass(dest='f', sources=['f'])
    with pytest.raises(RuntimeError):
        check_equation_array_properties(eq, [f])
    f.add_constant('total_mass', 0.0)
    check_equation_array_properties(eq, [f])


def test_acceleration_eval_checks_at_construction():
    from pysph_amd.kernels import CubicSpline
    f = get_particle_array(name='f', x=np.zeros(2))
    with pytest.raises(RuntimeError):
        AccelerationEval([f], [DummyEquation(dest='f', sources=['f'])], CubicSpline(dim=1))


def test_mega_group_retains_user_order_of_equations():             # :243-270
    group = Group(equations=[SimpleEquation(dest='f', sources=['s', 'f']),
                             DummyEquation(dest='f', sources=['s', 'f']),
                             MixedTypeEquation(dest='f', sources=['f'])])
    mg = MegaGroup(group, Group)
    assert list(mg.data.keys()) == ['f']
    eqs_with_no_source, sources, all_eqs = mg.data['f']
    assert len(eqs_with_no_source.equations) == 0
    names = lambda g: [type(x).__name__ for x in g.equations]
    assert names(all_eqs) == ['SimpleEquation', 'DummyEquation', 'MixedTypeEquation']
    assert sorted(sources.keys()) == ['f', 's']
    assert names(sources['s']) == ['SimpleEquation', 'DummyEquation']
    assert names(sources['f']) == ['SimpleEquation', 'DummyEquation', 'MixedTypeEquation']


def test_mega_group_copies_props_of_group():                        # :272-291
    def nothing():
        pass
    g = Group(equations=[], real=False, update_nnps=True, iterate=True, max_iterations=20,
              min_iterations=2, pre=nothing, post=nothing, start_idx=1, stop_idx=2, name='Dummy')
    mg = MegaGroup(g, Group)
    for prop in ('real update_nnps iterate max_iterations condition min_iterations pre post '
                 'start_idx stop_idx name').split():
        assert getattr(mg, prop) == getattr(g, prop)
