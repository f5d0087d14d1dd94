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


# ---------------------------------------------------------------------------
# what the structure of a group list lets the library assume inside one
# evaluation (sph_group.src_eos / nl_mode): derived on the host, no device needed
# ---------------------------------------------------------------------------
def _plan(arrays, groups, kernel_kind=2):
    from pysph_amd.acceleration_eval import _CGroup, annotate_plan
    ids = dict((pa.name, i) for i, pa in enumerate(arrays))
    amap = dict((pa.name, pa) for pa in arrays)
    plan = [(g, _CGroup(g, ids, amap, kernel_kind)) for g in groups]
    annotate_plan(plan)
    return plan


def _modes(plan):
    return [[(u.cg.src_eos, u.cg.nl_mode) for u in cg.units] for _, cg in plan]


def test_dam_break_groups_promise_the_tait_eos_to_the_pair_group():
    from pysph_amd.examples import dam_break_3d as db
    arrays = db.create_particles(0.2)
    plan = _plan(arrays, db.create_scheme(0.2).get_equations())
    m = _modes(plan)
    assert all(e == 0 for grp in m[:-1] for e, _ in grp)          # the EOS group itself: nothing
    assert [e for e, _ in m[-1]] == [1] * len(m[-1])              # fluid, boundary, obstacle destinations
    u = plan[-1][1].units[0]
    assert list(u.cg.eos_par) == [db.ro, db.c0, db.gamma, 0.0]
    assert all(n == 0 for grp in m for _, n in grp)               # three sources: no list reuse


def test_eos_promise_needs_every_array_and_all_particles():
    from pysph_amd.equations import (ContinuityEquation, Group, MomentumEquation, TaitEOS)
    from pysph_amd.particle_array import get_particle_array_wcsph
    a = get_particle_array_wcsph(name='a', x=np.zeros(2))
    b = get_particle_array_wcsph(name='b', x=np.zeros(2))
    pair = Group(equations=[ContinuityEquation('a', ['a', 'b']),
                            MomentumEquation('a', ['a', 'b'], c0=10.0)])
    eos_a = TaitEOS('a', None, rho0=1000., c0=10., gamma=7.)
    eos_b = TaitEOS('b', None, rho0=1000., c0=10., gamma=7.)
    full = _plan([a, b], [Group(equations=[eos_a, eos_b], real=False), pair])
    assert _modes(full)[-1][0][0] == 1
    # b has no EOS / the EOS skips the ghosts / another c0 / a neighbour update in
    # between / a group in between: no promise
    between = Group(equations=[ContinuityEquation('b', ['a'])])
    for groups in ([Group(equations=[eos_a], real=False), pair],
                   [Group(equations=[eos_a, eos_b], real=True), pair],
                   [Group(equations=[eos_a, TaitEOS('b', None, rho0=1000., c0=11., gamma=7.)],
                          real=False), pair],
                   [Group(equations=[eos_a, eos_b], real=False, update_nnps=True), pair],
                   [Group(equations=[eos_a, eos_b], real=False), between, pair]):
        assert _modes(_plan([a, b], groups))[-1][0][0] == 0, groups


def test_tvf_and_elastic_second_passes_reuse_the_first_pass_lists():
    from pysph_amd import kernels as K
    from pysph_amd.particle_array import get_particle_array_tvf_fluid
    from pysph_amd.scheme import TVFScheme
    from pysph_amd.solid_mech import (ElasticSolidsScheme,
                                      get_particle_array_elastic_dynamics)
    f = get_particle_array_tvf_fluid(name='fluid', x=np.zeros(2))
    groups = TVFScheme(['fluid'], [], dim=3, rho0=1.0, c0=10.0, nu=0.01, p0=100.0,
                       pb=100.0, h0=0.1).get_equations()
    m = _modes(_plan([f], groups, kernel_kind=3))
    assert [n for grp in m for _, n in grp] == [1, 0, 2]          # density keeps, state equation, force reuses
    s = get_particle_array_elastic_dynamics(name='solid', x=np.zeros(2))
    m = _modes(_plan([s], ElasticSolidsScheme(['solid'], [], dim=3).get_equations(), kernel_kind=1))
    assert [n for grp in m for _, n in grp] == [1, 2]             # velocity gradient keeps, rates reuse
    # a real=True first pass does not cover the ghosts a real=False second pass loops over
    from pysph_amd.equations import Group, SummationDensity
    w = get_particle_array(name='w')
    g1 = Group(equations=[SummationDensity('w', ['w'])], real=True)
    g2 = Group(equations=[SummationDensity('w', ['w'])], real=False)
    assert [n for grp in _modes(_plan([w], [g1, g2])) for _, n in grp] == [0, 0]
    assert [n for grp in _modes(_plan([w], [g2, g1])) for _, n in grp] == [1, 2]
    # a neighbour update between the passes: nothing is kept
    g2u = Group(equations=[SummationDensity('w', ['w'])], real=False, update_nnps=True)
    assert [n for grp in _modes(_plan([w], [g2u, g1])) for _, n in grp] == [0, 0]


def test_tvf_force_group_gets_the_state_promise():
    """round 4: [TVF SummationDensity | StateEquation | force group], each over all
    particles: p and V = rho / m are functions of rho when the force group runs
    (sph_group.src_eos = 2, eos_par = p0 rho0 b) -- and not when a link is missing"""
    from pysph_amd.equations import (Group, MomentumEquationPressureGradient, StateEquation,
                                     TVFSummationDensity)
    from pysph_amd.particle_array import get_particle_array_tvf_fluid
    from pysph_amd.scheme import TVFScheme
    f = get_particle_array_tvf_fluid(name='fluid', x=np.zeros(2))
    groups = TVFScheme(['fluid'], [], dim=3, rho0=2.0, c0=10.0, nu=0.01, p0=100.0,
                       pb=100.0, h0=0.1).get_equations()
    plan = _plan([f], groups, kernel_kind=3)
    u = plan[-1][1].units[0]
    assert u.cg.src_eos == 2 and list(u.cg.eos_par)[:3] == [100.0, 2.0, 1.0]
    assert [e for grp in _modes(plan)[:-1] for e, _ in grp] == [0, 0]
    dens = Group(equations=[TVFSummationDensity('fluid', ['fluid'])], real=False)
    state = Group(equations=[StateEquation('fluid', None, p0=100.0, rho0=2.0, b=1.0)], real=False)
    force = Group(equations=[MomentumEquationPressureGradient('fluid', ['fluid'], pb=100.0)])
    assert _plan([f], [dens, state, force], 3)[-1][1].units[0].cg.src_eos == 2
    state_real = Group(equations=[StateEquation('fluid', None, p0=100.0, rho0=2.0, b=1.0)], real=True)
    dens_real = Group(equations=[TVFSummationDensity('fluid', ['fluid'])], real=True)
    for gl in ([state, force],                      # no density summation in this evaluation
               [dens, state_real, force],           # the state equation skips the ghosts
               [dens_real, state, force],           # the density summation skips the ghosts
               [dens, state, Group(equations=[TVFSummationDensity('fluid', ['fluid'])], real=False), force]):
        assert _plan([f], gl, 3)[-1][1].units[-1].cg.src_eos != 2, gl


def test_a_group_with_several_destinations_goes_to_the_library_in_one_call():
    """round 4: the dam break's rate group (boundary, obstacle, fluid destinations) is ONE
    sph_eval_group call carrying every unit's equations and the promises they share, so that
    the library can run it as one launch over the merged order of the arrays"""
    from pysph_amd.examples import dam_break_3d as db
    arrays = db.create_particles(0.2)
    plan = _plan(arrays, db.create_scheme(0.2).get_equations())
    cg = plan[-1][1]
    assert len(cg.units) == 3
    calls = []

    class Lib(object):
        def sph_eval_group(self, ctx, kernel, group, t, dt):
            g = group._obj
            calls.append((g.neq, g.src_eos, list(g.eos_par), g.real, g.start_idx, g.stop_idx,
                          [(g.eqs[i].kind, g.eqs[i].dest, g.eqs[i].nsrc) for i in range(g.neq)]))
            return 0

    class Ev(object):
        lib = Lib()
        ckernel = None

        class ctx(object):
            _h = None
    import ctypes
    Ev.ckernel = ctypes.c_int(0)
    cg.refresh_range()
    cg.run(Ev, 0.0, 1e-5)
    assert len(calls) == 1
    neq, src_eos, eos_par, real, start, stop, eqs = calls[0]
    assert neq == sum(len(u.eqs) for u in cg.units) == 5 and src_eos == 1
    assert eos_par == [db.ro, db.c0, db.gamma, 0.0] and real == 1 and (start, stop) == (0, -1)
    # boundary <- fluid, obstacle <- fluid (continuity), then the fluid's three equations
    assert [(k, n) for k, _, n in eqs] == [(3, 1), (3, 1), (3, 3), (4, 3), (5, 1)]
    assert len(set(d for _, d, _ in eqs)) == 3
