"""Build (hipcc, gfx950) every generated family the -m gpu tests use, without
a GPU: called by __graft_entry__.build() so that the shared objects under
pysph_amd/_gen/ travel with the snapshot and the GPU run does not compile."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def main():
    from pysph_amd import kernels as K
    from pysph_amd.acceleration_eval import AccelerationEval, _CGroup
    from conftest import load_golden, arrays_from_golden
    from helpers import golden_case
    import test_hip_parity as T
    from custom_equations import PyContinuity, PyMomentum, PyXSPH
    from pysph_amd.equations import Group
    n = 0

    f32_all = os.environ.get('SPHGEN_PREBUILD_F32') == 'all'     # compile shake-out of the float builds of EVERY family

    def plan(arrays, eqs, kernel, f32=False):
        count = 0
        a = AccelerationEval(arrays, eqs, kernel)
        ids = dict((pa.name, i) for i, pa in enumerate(arrays))
        amap = dict((pa.name, pa) for pa in arrays)
        for g in a.equation_groups:
            cg = _CGroup(g, ids, amap, K.kernel_id(kernel))
            count += sum(hasattr(u, 'fam') for u in cg.units)
            if f32 or f32_all:        # the float builds the option arith_f32 launches (tests that set it)
                for u in cg.units:
                    if hasattr(u, 'fam'):
                        u.fam.flavour_f32().load()
                        count += 1
        return count

    g = load_golden('tvf_wall.npz')
    eqs, kernel, dim, outs = golden_case('tvf_wall', g)
    n += plan(arrays_from_golden(g, 'in'), eqs, kernel)
    import wall_equations_fixture
    eqs, kernel, dim, outs = golden_case('tvf_wall', g, wall_equations=wall_equations_fixture)
    n += plan(arrays_from_golden(g, 'in'), eqs, kernel)
    for kname in ('CubicSpline', 'Gaussian'):
        arrays, eqs = T._custom_setup(0.0)
        n += plan(arrays, eqs, getattr(K, kname)(dim=3))
    for kname in ('CubicSpline', 'WendlandQuintic', 'QuinticSpline', 'Gaussian'):
        arrays, eqs, dim, _ = T._random_generated_case(7)
        n += plan(arrays, eqs, getattr(K, kname)(dim=3))
    for case in T.F32_GENERATED_CASES:      # the float builds of test_generated_families_fp32_arithmetic_vs_python
        arrays, eqs, kernel, dim, _ = T._f32_generated_case(case)
        n += plan(arrays, eqs, kernel, f32=True)
    n += plan([T._image_case()], T._image_equations(), K.CubicSpline(dim=1))
    for kname in ('CubicSpline', 'WendlandQuintic', 'QuinticSpline', 'Gaussian'):
        gp, geqs, gk = T._gradh_case(kname)
        n += plan([gp], geqs, gk)
    for nsys in (2, 3, 4):
        sp, seqs = T._systems_case(nsys)
        n += plan([sp], seqs, K.CubicSpline(dim=1))
    pa, dx = T.make_cube(6)
    for tensile in (False, True):
        kw = dict(c0=32.85, alpha=0.25, beta=0.1, gz=-9.81, tensile_correction=tensile)
        eqs = [Group(equations=[PyContinuity(dest='fluid', sources=['fluid']),
                                PyMomentum(dest='fluid', sources=['fluid'], **kw),
                                PyXSPH(dest='fluid', sources=['fluid'], eps=0.5)])]
        n += plan([pa], eqs, K.WendlandQuintic(dim=3))
    from custom_equations import StridedDiffusion, StridedGradient
    from pysph_amd.particle_array import get_particle_array_wcsph
    pb = get_particle_array_wcsph(name='fluid', x=np.zeros(2))
    pb.add_property('g3', stride=3)
    n += plan([pb], [Group(equations=[StridedGradient('fluid', ['fluid'])], real=False),
                     Group(equations=[StridedDiffusion('fluid', ['fluid'])])],
              K.CubicSpline(dim=3))
    from custom_equations import GradientAllNbrs, ShepardFilter, SmoothCopy
    pq = get_particle_array_wcsph(name='fluid', x=np.zeros(2))
    pq.add_property('q')
    pq.add_property('qtmp')
    n += plan([pq], [Group(equations=[SmoothCopy('fluid', ['fluid'])])], K.CubicSpline(dim=3))
    arrs = []
    for name in ('fluid', 'solid'):
        pc = get_particle_array_wcsph(name=name, x=np.zeros(2))
        for extra in ('rhotmp', 'gx', 'gy', 'gz'):
            pc.add_property(extra)
        arrs.append(pc)
    for kname in ('CubicSpline', 'WendlandQuintic'):
        n += plan(arrs, [Group(equations=[ShepardFilter('fluid', ['fluid'])]),
                         Group(equations=[GradientAllNbrs('fluid', ['fluid', 'solid'], scale=0.5),
                                          GradientAllNbrs('solid', ['fluid'], scale=2.0)], real=False)],
                  getattr(K, kname)(dim=3))
    import test_reference_scenarios as RS
    from pysph_amd.particle_array import get_particle_array
    ps = get_particle_array(name='fluid', x=np.zeros(3))
    ps.add_constant('total_mass', 0.0)
    ps.add_constant('reduce_calls', 0)
    for cls, src in ((RS.SimpleEquation, ['fluid']), (RS.SimpleReduction, ['fluid']), (RS.PyInit, None),
                     (RS.LoopAllEquation, ['fluid']), (RS.DumbEquation, ['fluid']),
                     (RS.EqWithTime, ['fluid']), (RS.SillyEquation, ['fluid'])):
        n += plan([ps], [Group(equations=[cls('fluid', src)])], K.CubicSpline(dim=1))
    n += plan([ps], [Group(equations=[RS.SimpleEquation('fluid', ['fluid']),
                                      RS.SimpleEquation('fluid', ['fluid'])])], K.CubicSpline(dim=1))
    n += plan([ps], [Group(equations=[RS.InitializePair('fluid', ['fluid'])])], K.CubicSpline(dim=1))
    n += plan([RS.newton_array()], RS.newton_equations(RS.NewtonSqrt('fluid', None)), K.CubicSpline(dim=1))
    n += plan(RS.ghost_copy_arrays(), RS.ghost_copy_equations(), K.CubicSpline(dim=1))
    n += plan([T._correction_case()], T._correction_equations(), K.CubicSpline(dim=3))
    for cls in (RS.HelperEquation, RS.MixedTypeEquation):
        n += plan([ps], [Group(equations=[cls('fluid', ['fluid'])])], K.CubicSpline(dim=1))
    n += plan([ps], [Group(equations=[RS.HelperEquation('fluid', ['fluid']),
                                      RS.HelperEquation('fluid', ['fluid'])])], K.CubicSpline(dim=1))
    import test_kernel_corrections as KC
    for dim in (2, 3):
        n += plan([KC.corner_particles(dim)], KC.correction_equations(dim), K.CubicSpline(dim=dim))
    import test_kernel_moments as KM
    for kname, dim in sorted(KM.PLACES):
        n += plan(list(KM.moment_arrays(2)), KM.moment_equations(), getattr(K, kname)(dim=dim))
    import test_reference_integrators as RI
    n += RI.prebuild()
    n += RI.prebuild_golden_steppers()
    n += RI.prebuild_multi_stage()
    n += plan([RI.make_pa()], [RI.SHM(dest='fluid', sources=None)], K.CubicSpline(dim=1))
    return n


if __name__ == '__main__':
    # pass 1 collects what is missing from the cache, build_deferred() compiles it in
    # parallel, pass 2 walks everything again with the real modules (loads every one)
    from pysph_amd import codegen
    codegen.DEFERRED = []
    main()
    print('compiled in parallel:', codegen.build_deferred())
    print('generated families built:', main())
