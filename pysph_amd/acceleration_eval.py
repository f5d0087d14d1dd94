"""The drop-in boundary: ``AccelerationEval`` and its HIP compiled object.

Reference interfaces mirrored here (pypr/pysph):

* ``AccelerationEval(particle_arrays, equations, kernel, mode, backend)`` with
  ``compute / set_nnps / update_particle_arrays / set_compiled_object``
  -- pysph/sph/acceleration_eval.py:166-248;
* ``MegaGroup`` regrouping ``{dest: (eqs_with_no_source, sources, all_eqs)}``
  -- acceleration_eval.py:94-162;
* the ``c_acceleration_eval`` protocol ``compute(t, dt)``, ``set_nnps(nnps)``,
  ``update_particle_arrays(arrays)`` implemented by the generated Cython class
  (acceleration_eval_cython.mako:197-363) and by ``GPUAccelerationEval``
  (acceleration_eval_gpu_helper.py:204-335) -- here ``HipAccelerationEval``;
* the helper protocol ``SPHCompiler`` drives: ``Helper(a_eval)``,
  ``get_code()``, ``compile(code)``, ``setup_compiled_module(module)``
  (sph_compiler.py:27-59) -- here ``AccelerationEvalHipHelper``.

``HipAccelerationEval`` accepts either this package's spec objects or a real
``pysph.sph.acceleration_eval.AccelerationEval`` (it only reads
``particle_arrays``, ``equation_groups``/``mega_groups`` and ``kernel``), see
INTEGRATION.md.

Group control flow that is host-side in the reference stays host-side here:
``pre/post/condition`` callbacks, ``iterate`` with ``converged()``,
``update_nnps`` and ``py_initialize`` / ``reduce`` hooks
(acceleration_eval_cython.mako:291-363).
"""
import ctypes as C
from collections import OrderedDict, defaultdict

from . import device as dev
from .equations import Group, MultiStageEquations, resolve_equation
from .kernels import kernel_id
from .particle_array import has_prop


def group_equations(equations):
    """acceleration_eval.py:14-28."""
    only_groups = [x for x in equations if hasattr(x, 'has_subgroups')]
    if only_groups and len(only_groups) != len(equations):
        raise ValueError('All elements must be Groups if you use groups.')
    if not only_groups:
        return [Group(equations)]
    return equations


def check_equation_array_properties(equation, particle_arrays):
    """acceleration_eval.py:32-73: RuntimeError when a destination/source array
    or one of the properties the equation touches is missing."""
    arrays = dict((pa.name, pa) for pa in particle_arrays)
    try:
        kind, vals, dprops, sprops = resolve_equation(equation)
    except NotImplementedError:
        from .codegen import has_python_body, method_properties
        if not has_python_body(equation):
            raise
        dprops, sprops = method_properties(equation)
    if equation.dest not in arrays:
        raise RuntimeError("ERROR: Equation %s has invalid dest: '%s'" %
                           (equation.name, equation.dest))
    for src in (equation.sources or []):
        if src not in arrays:
            raise RuntimeError("ERROR: Equation %s has invalid source: '%s'" %
                               (equation.name, src))
    missing = defaultdict(set)
    for p in dprops:
        if not has_prop(arrays[equation.dest], p):
            missing[equation.dest].add(p)
    for src in (equation.sources or []):
        for p in sprops:
            if not has_prop(arrays[src], p):
                missing[src].add(p)
    if missing:
        msg = 'ERROR: Missing array properties for equation: %s\n' % equation.name
        for name, props in missing.items():
            msg += "Array '%s' missing properties %s.\n" % (name, sorted(props))
        raise RuntimeError(msg)


class MegaGroup(object):
    """acceleration_eval.py:94-162."""

    def __init__(self, group, group_cls=Group):
        self._orig_group = group
        self.Group = group_cls
        for key in ('real', 'update_nnps', 'iterate', 'pre', 'post',
                    'max_iterations', 'min_iterations', 'has_subgroups',
                    'condition', 'start_idx', 'stop_idx', 'name'):
            setattr(self, key, getattr(group, key))
        self.data = self._make_data(group)

    def _make_data(self, group):
        if group.has_subgroups:
            return [MegaGroup(g, self.Group) for g in group.equations]
        dests = OrderedDict()
        for eq in group.equations:
            if eq.dest not in dests:
                dests[eq.dest] = ([], OrderedDict(), [])
            no_src, sources, all_eqs = dests[eq.dest]
            if eq not in all_eqs:
                all_eqs.append(eq)
            if eq.no_source:
                no_src.append(eq)
            else:
                for s in eq.sources:
                    sources.setdefault(s, []).append(eq)
        # dest -> (eqs_with_no_source: Group, {source: Group}, all_eqs: Group), the user's
        # equation order kept in each (acceleration_eval.py:126-162)
        out = OrderedDict()
        for dest, (no_src, sources, all_eqs) in dests.items():
            out[dest] = (self.Group(equations=no_src),
                         OrderedDict((s_, self.Group(equations=e_)) for s_, e_ in sources.items()),
                         self.Group(equations=all_eqs))
        return out


class AccelerationEval(object):
    """acceleration_eval.py:166-248 (backend is always the HIP one here)."""

    def __init__(self, particle_arrays, equations, kernel, mode='serial',
                 backend='hip'):
        assert backend in ('hip', '', None)
        self.backend = 'hip'
        self.particle_arrays = particle_arrays
        self.equation_groups = group_equations(equations)
        self.kernel = kernel
        self.nnps = None
        self.mode = mode
        all_eqs = []
        for g in self.equation_groups:
            if g.has_subgroups:
                for sg in g.equations:
                    all_eqs.extend(sg.equations)
            else:
                all_eqs.extend(g.equations)
        self.all_equations = all_eqs
        for eq in all_eqs:
            check_equation_array_properties(eq, particle_arrays)
        self.mega_groups = [MegaGroup(g) for g in self.equation_groups]
        self.c_acceleration_eval = None

    def compute(self, t, dt):
        self.c_acceleration_eval.compute(t, dt)

    def set_compiled_object(self, c_acceleration_eval):
        self.c_acceleration_eval = c_acceleration_eval

    def set_nnps(self, nnps):
        self.nnps = nnps
        self.c_acceleration_eval.set_nnps(nnps)

    def update_particle_arrays(self, particle_arrays):
        self.c_acceleration_eval.update_particle_arrays(particle_arrays)


def make_acceleration_evals(particle_arrays, equations, kernel, mode='serial',
                            backend='hip'):
    """acceleration_eval.py:76-90."""
    groups = equations.groups if isinstance(equations, MultiStageEquations) \
        else [equations]
    return [AccelerationEval(particle_arrays, g, kernel, mode, backend)
            for g in groups]


def is_builtin(eq):
    """hand-written kernel available (matched by class name, as
    ``resolve_equation`` does)?"""
    try:
        resolve_equation(eq)
        return True
    except NotImplementedError:
        return False


class _BuiltinUnit(object):
    """Equations of one destination with hand-written kernels, marshalled for
    ``sph_eval_group``."""

    def __init__(self, group, eqs, array_ids, arrays, owner):
        self.group = group
        self.eqs = eqs
        self.ceqs = (dev.SphEquation * max(len(eqs), 1))()
        self._const_params = []
        for i, eq in enumerate(eqs):
            kind, vals, dprops, sprops = resolve_equation(eq)
            ce = self.ceqs[i]
            ce.kind = kind
            ce.dest = array_ids[eq.dest]
            srcs = eq.sources or []
            ce.nsrc = len(srcs)
            for k, s in enumerate(srcs):
                ce.src[k] = array_ids[s]
            for k, v in enumerate(vals):
                if isinstance(v, str):           # '@const' -> read every compute
                    self._const_params.append((i, k, eq.dest, v[1:]))
                else:
                    ce.par[k] = v
            owner.inputs[eq.dest].update(dprops)
            owner.outputs[eq.dest].update(dprops)
            for s in srcs:
                owner.inputs[s].update(sprops)
        self.cg = dev.SphGroup()
        self.cg.real = 1 if group.real else 0
        self.cg.neq = len(eqs)
        self.cg.eqs = self.ceqs
        self._arrays = arrays

    @property
    def has_pair(self):
        return any(self.ceqs[k].nsrc > 0 for k in range(len(self.eqs)))

    def single_pair_key(self):
        """(dest id, source id) when every pair equation of this unit loops over
        the same single source, else None"""
        keys = set()
        for k in range(len(self.eqs)):
            ce = self.ceqs[k]
            if ce.nsrc == 0:
                continue
            if ce.nsrc != 1:
                return None
            keys.add((ce.dest, ce.src[0]))
        return keys.pop() if len(keys) == 1 else None

    def refresh(self, start, stop):
        from .particle_array import get_npy as _get
        for i, k, dest, cname in self._const_params:
            self.ceqs[i].par[k] = float(_get(self._arrays[dest], cname)[0])
        self.cg.start_idx, self.cg.stop_idx = start, stop

    def run(self, ev, t, dt, phase=0):
        self.cg.phase = phase
        try:
            dev._check(ev.lib.sph_eval_group(
                ev.ctx._h, C.byref(ev.ckernel), C.byref(self.cg), t, dt))
        finally:
            self.cg.phase = 0


class _GeneratedUnit(object):
    """Equations of one destination WITHOUT hand-written kernels: translated
    and compiled by ``pysph_amd.codegen`` and run through
    ``sph_eval_generated``."""

    def __init__(self, group, dest, eqs, array_ids, arrays, kernel_kind, owner,
                 skip_initialize=()):
        from .codegen import GeneratedFamily
        self.group = group
        self.eqs = eqs
        self.fam = GeneratedFamily(dest, eqs, arrays, kernel_kind,
                                   name='%s_%s' % (group.name, dest),
                                   skip_initialize=skip_initialize)
        f = self.fam
        lib = f.load()
        self.cf = dev.SphGenFamily()
        self.cf.launch = C.cast(lib.sphgen_launch, C.c_void_p).value
        self.cf.dest = array_ids[dest]
        self.cf.nsrc = len(f.sources)
        for j, sname in enumerate(f.sources):
            self.cf.src[j] = array_ids[sname]
            self.cf.src_flags[j] = f.src_flags[sname]
        self.cf.n_sprops = len(f.sprops)
        for k, p in enumerate(f.sprops):
            self.cf.sprops[k] = dev.prop_register(p)
        self.cf.n_din = len(f.din)
        for k, p in enumerate(f.din):
            self.cf.din[k] = dev.prop_register(p)
        self.cf.n_dout = len(f.dout)
        for k, p in enumerate(f.dout):
            self.cf.dout[k] = dev.prop_register(p)
        self.cf.npar = len(f.params)
        self.cf.real = 1 if group.real else 0
        self.cf.split_init = 1 if f.split_init else 0
        self.cf.loop_all = 1 if f.loop_all else 0
        self.cf.also_pair = 1 if f.also_pair else 0
        self.cf.init_pair = int(f.init_pair)
        self.cf.nstate = len(f.state)
        owner.inputs[dest].update(f.dprops)
        owner.outputs_exact[dest].update(f.dout)    # exactly what the bodies write
        for sname in f.sources:
            owner.inputs[sname].update(list(f.sprops) + ['x', 'y', 'z', 'h'])

    def refresh(self, start, stop):
        for k, v in enumerate(self.fam.param_values()):
            self.cf.par[k] = v
        self.cf.start_idx, self.cf.stop_idx = start, stop

    def _float_build(self, ev):
        """option arith_f32: the pair launch of this family runs its float build
        (codegen.source_f32), compiled / loaded the first time the option is seen"""
        f = self.fam
        if f.abs_src_pos and ev.ctx.options.get('record_f32'):
            raise RuntimeError(
                "generated family '%s' reads a neighbour's absolute position (s_x[s_idx]); fp32 records "
                "(option record_f32) hold positions relative to the grid origin" % f.name)
        if ev.ctx.options.get('arith_f32') and not self.cf.launch_f32 and f.sources and f.bodies['loop'] \
                and not f.abs_src_pos:
            self._lib32 = f.flavour_f32().load()
            self.cf.launch_f32 = C.cast(self._lib32.sphgen_launch, C.c_void_p).value

    def run(self, ev, t, dt):
        f = self.fam
        self._float_build(ev)
        for k, v in enumerate(f.state_values()):
            self.cf.state[k] = v
        dev._check(ev.lib.sph_eval_generated(
            ev.ctx._h, C.addressof(ev.ckernel), C.addressof(self.cf), t, dt))
        if f.state:
            f.store_state([self.cf.state[k] for k in range(len(f.state))])


class _CGroup(object):
    """One leaf group: per destination (first-appearance order,
    acceleration_eval.py:126-131) a unit with hand-written kernels and/or a
    generated unit.

    When a destination mixes both kinds, the hand-written unit runs first and
    the generated one continues from the values in memory (``d_au[d_idx] += ..``
    keeps its meaning; the sums only associate differently).  That is exact
    with respect to the reference's initialize -> loops -> post_loop order as
    long as the generated equations have no ``initialize`` (it would reset what
    the first pass accumulated) -- which is checked."""

    def __init__(self, group, array_ids, arrays, kernel_kind=None):
        self.group = group
        self.inputs = defaultdict(set)    # array name -> props read
        self.outputs = defaultdict(set)   # array name -> props written (hand-written kernels: a superset)
        self.outputs_exact = defaultdict(set)  # generated families: exactly the written properties
        self.units = []
        self._arrays = arrays
        dests = []
        for eq in group.equations:
            if eq.dest not in dests:
                dests.append(eq.dest)
        for dest in dests:
            eqs = [eq for eq in group.equations if eq.dest == dest]
            builtin = [eq for eq in eqs if is_builtin(eq)]
            custom = [eq for eq in eqs if not is_builtin(eq)]
            if builtin:
                self.units.append(_BuiltinUnit(group, builtin, array_ids, arrays, self))
            if custom:
                from .codegen import has_python_body, initialize_only_resets
                skip = []
                reset_by_builtin = set()
                for eq in builtin:
                    reset_by_builtin.update(resolve_equation(eq)[2])
                for eq in custom:
                    if not has_python_body(eq):
                        resolve_equation(eq)     # raises the "no kernel" error
                    if builtin and callable(getattr(type(eq), 'initialize', None)):
                        # the hand-written pass has already run: an initialize
                        # that only repeats its resets is dropped, anything
                        # else cannot be ordered correctly
                        if not initialize_only_resets(eq, reset_by_builtin):
                            raise NotImplementedError(
                                'destination %r mixes hand-written equations with '
                                'the generated %s, whose initialize() would run '
                                'after the hand-written pass; put it in its own '
                                'group' % (dest, type(eq).__name__))
                        skip.append(eq)
                self.units.append(_GeneratedUnit(group, dest, custom, array_ids,
                                                 arrays, kernel_kind, self, skip))

    def _combined_group(self):
        """ONE sph_group holding the equations of every (hand-written) unit: a
        group with several destinations goes to the library in one call, so
        that it can run all of them in one launch over the merged cell order of
        the arrays (sph_eval.hip: eval_group_merged) -- or, when that does not
        apply, loop over the destinations itself with shared source records."""
        units = self.units
        total = sum(len(u.eqs) for u in units)
        ceqs = (dev.SphEquation * max(total, 1))()
        cg = dev.SphGroup()
        cg.real = units[0].cg.real
        cg.neq = total
        cg.eqs = ceqs
        return cg, ceqs

    def refresh_range(self):
        g = self.group
        dest = self._arrays[g.equations[0].dest] if g.equations else None

        def resolve(v, default):
            if v is None:
                return default
            if isinstance(v, str):  # property/constant name: first value
                from .particle_array import get_npy
                return int(get_npy(dest, v)[0])
            return int(v)
        start, stop = resolve(g.start_idx, 0), resolve(g.stop_idx, -1)
        for u in self.units:
            u.refresh(start, stop)
        self._start_stop = (start, stop)

    def run(self, ev, t, dt):
        # several destinations with hand-written kernels only: ONE call for the
        # whole group (the equations by value, as the units hold them now)
        shared = len(self.units) > 1 and all(isinstance(u, _BuiltinUnit) for u in self.units)
        if shared:
            # ... and only when ONE set of promises describes every destination: units that were promised
            # different things (or a neighbour-list mode, which belongs to one (destination, source) pair) keep
            # their own calls -- a combined call would have to drop what differs
            u0 = self.units[0].cg
            shared = all(u.cg.src_eos == u0.src_eos and list(u.cg.eos_par) == list(u0.eos_par) and u.cg.nl_mode == 0
                         for u in self.units)
        if not shared:
            for u in self.units:
                u.run(ev, t, dt)
            return
        if getattr(self, '_combined', None) is None:
            self._combined = self._combined_group()
        cg, ceqs = self._combined
        i = 0
        for u in self.units:
            for k in range(len(u.eqs)):
                ceqs[i] = u.ceqs[k]
                i += 1
        cg.start_idx, cg.stop_idx = self._start_stop
        # the promises annotate_plan made for every unit (the same for all: checked above) hold for the group
        u0 = self.units[0].cg
        cg.src_eos = u0.src_eos
        for k in range(4):
            cg.eos_par[k] = u0.eos_par[k]
        cg.nl_mode = 0
        cg.phase = 0
        dev._check(ev.lib.sph_eval_group(
            ev.ctx._h, C.byref(ev.ckernel), C.byref(cg), t, dt))


_TAIT_KINDS = (1, 2)            # SPH_EQ_TAIT_EOS, SPH_EQ_TAIT_EOS_HG
_WCSPH_PAIR_KINDS = (3, 4, 5)   # SPH_EQ_CONTINUITY, SPH_EQ_MOMENTUM, SPH_EQ_XSPH
_TVF_DENSITY_KIND = 7           # SPH_EQ_TVF_SUMMATION_DENSITY
_TVF_STATE_KIND = 8             # SPH_EQ_TVF_STATE_EQUATION
_TVF_FORCE_KINDS = (9, 10, 11, 12)  # pressure gradient, viscosity, artificial viscosity, artificial stress


def _plain_leaf(g, cg):
    """a leaf group whose execution is exactly one pass over its units: no
    host callbacks, no iteration, no condition, no neighbour update after it"""
    if g.has_subgroups or g.iterate or g.condition is not None or g.pre or g.post \
            or g.update_nnps:
        return False
    for eq in g.equations:
        if hasattr(eq, 'py_initialize') or hasattr(eq, 'reduce'):
            return False
    return all(isinstance(u, _BuiltinUnit) for u in cg.units)


def _ranged(g):
    """Group(start_idx=..., stop_idx=...) restricts the destinations"""
    return g.start_idx not in (None, 0) or g.stop_idx is not None


def annotate_plan(plan):
    """What the STRUCTURE of the group list allows the library to assume inside
    one evaluation (sph_group.src_eos / nl_mode, include/sphhip.h).  Only
    sequences of plain leaf groups with hand-written kernels are considered;
    everything else keeps the general path.

    * EOS fusion: a WCSPH pair group (Continuity / Momentum / XSPH) directly
      preceded by a group that is nothing but TaitEOS / TaitEOSHGCorrection over
      ALL particles (real=False, no index range) of every array the pair group
      touches, all with the same rho0, c0, gamma and p0: p and cs of those
      arrays are then functions of rho when the pair group runs.
    * neighbour-list reuse: two pair units of the same evaluation over the same
      (destination, single source) with no neighbour update in between -- TVF's
      density and force passes, the elastic set's velocity-gradient and rate
      passes: the first keeps its per-lane hit lists, the second starts from
      them (the reference's NeighborCache, nnps_base.pyx:1144-1257)."""
    if any(isinstance(item, list) for _, item in plan):
        return
    leaves = [(g, cg) for g, cg in plan]
    plain = [_plain_leaf(g, cg) for g, cg in leaves]
    for g, cg in leaves:
        for u in cg.units:
            if isinstance(u, _BuiltinUnit):
                u.cg.src_eos = 0
                u.cg.nl_mode = 0
    # -- EOS fusion ---------------------------------------------------------
    for i in range(1, len(leaves)):
        if not (plain[i] and plain[i - 1]):
            continue
        g0, cg0 = leaves[i - 1]
        g1, cg1 = leaves[i]
        if g0.real or _ranged(g0):
            continue
        eos = {}
        ok = True
        for u in cg0.units:
            for k in range(len(u.eqs)):
                ce = u.ceqs[k]
                if ce.kind not in _TAIT_KINDS or u._const_params:
                    ok = False
                    break
                par = (ce.par[0], ce.par[1], ce.par[2], ce.par[3] if ce.kind == 1 else 0.0)
                if eos.setdefault(ce.dest, par) != par:
                    ok = False
        if not ok or not eos or len(set(eos.values())) != 1:
            continue
        par = list(eos.values())[0]
        for u in cg1.units:
            used = set()
            pair = True
            for k in range(len(u.eqs)):
                ce = u.ceqs[k]
                if ce.kind not in _WCSPH_PAIR_KINDS or ce.nsrc == 0:
                    pair = False
                used.add(ce.dest)
                used.update(ce.src[j] for j in range(ce.nsrc))
            if pair and used and used <= set(eos):
                u.cg.src_eos = 1
                for k in range(4):
                    u.cg.eos_par[k] = par[k]
    # -- TVF state fusion: [TVF SummationDensity | StateEquation | force group], each over ALL
    #    particles (real=False, no range) of the arrays the force group reads: p and V = rho / m
    #    are functions of rho when the force group runs (src_eos = 2, eos_par = p0 rho0 b)
    for i in range(2, len(leaves)):
        if not (plain[i] and plain[i - 1] and plain[i - 2]):
            continue
        (gd, cgd), (gs, cgs), (gf, cgf) = leaves[i - 2], leaves[i - 1], leaves[i]
        if gd.real or gs.real or _ranged(gd) or _ranged(gs):
            continue
        dens = set()
        ok = True
        for u in cgd.units:
            for k in range(len(u.eqs)):
                if u.ceqs[k].kind != _TVF_DENSITY_KIND:
                    ok = False
                dens.add(u.ceqs[k].dest)
        state = {}
        for u in cgs.units:
            for k in range(len(u.eqs)):
                ce = u.ceqs[k]
                if ce.kind != _TVF_STATE_KIND or u._const_params:
                    ok = False
                    break
                par = (ce.par[0], ce.par[1], ce.par[2])
                if state.setdefault(ce.dest, par) != par:
                    ok = False
        if not ok or not state or len(set(state.values())) != 1:
            continue
        par = list(state.values())[0]
        for u in cgf.units:
            used = set()
            force = True
            for k in range(len(u.eqs)):
                ce = u.ceqs[k]
                if ce.kind not in _TVF_FORCE_KINDS or ce.nsrc == 0:
                    force = False
                used.add(ce.dest)
                used.update(ce.src[j] for j in range(ce.nsrc))
            if force and used and used <= set(state) and used <= dens:
                u.cg.src_eos = 2
                for k in range(3):
                    u.cg.eos_par[k] = par[k]
                u.cg.eos_par[3] = 0.0
    # -- neighbour-list reuse ---------------------------------------------
    keeper = None                    # (key, unit, group) of the pass whose lists would be the kept ones
    for (g, cg), ok in zip(leaves, plain):
        if not ok:
            keeper = None
            continue
        for u in cg.units:
            if not u.has_pair:
                continue             # equations without sources leave positions and h alone
            key = u.single_pair_key()
            if key is None or _ranged(g):
                keeper = None
                continue
            if keeper is not None and keeper[0] == key and (not keeper[2].real or g.real):
                keeper[1].cg.nl_mode = 1     # its destination range covers this pass's
                u.cg.nl_mode = 2
            else:
                keeper = (key, u, g)


class HipAccelerationEval(object):
    """The compiled object behind ``AccelerationEval.set_compiled_object``.

    sync='auto'   (default) host arrays are authoritative, as with the Cython
                  backend: inputs are pushed before and outputs pulled after
                  every ``compute``;
    sync='manual' device-resident state, as with the reference's GPU backends
                  (explicit ``pa.gpu.push()/pull()``, cf.
                  test_acceleration_eval.py:178-191).
    """

    def __init__(self, a_eval, ctx=None, sync='auto'):
        self.a_eval = a_eval
        self.ctx = ctx or dev.get_context()
        self.lib = self.ctx.lib
        self.sync = sync
        self.nnps = None
        k = a_eval.kernel
        self.ckernel = dev.SphKernel(kernel_id(k), int(k.dim), float(k.fac),
                                     float(k.radius_scale),
                                     float(k.get_deltap()))
        self._setup(list(a_eval.particle_arrays))

    def _setup(self, arrays):
        self.particle_arrays = arrays
        self.arrays = OrderedDict((pa.name, pa) for pa in arrays)
        self.helpers = OrderedDict(
            (pa.name, dev.attach(pa, self.ctx)) for pa in arrays)
        ids = dict((n, h.array_id) for n, h in self.helpers.items())
        groups = getattr(self.a_eval, 'equation_groups', None)
        if groups is None:
            groups = group_equations(self.a_eval.equations)
        self.plan = [self._plan_group(g, ids) for g in groups]
        annotate_plan(self.plan)
        self.inputs = defaultdict(set)
        self.outputs = defaultdict(set)
        self.outputs_exact = defaultdict(set)
        for cg in self._leaves(self.plan):
            for n, p in cg.inputs.items():
                self.inputs[n].update(p)
            for n, p in cg.outputs.items():
                self.outputs[n].update(p)
            for n, p in cg.outputs_exact.items():
                self.outputs_exact[n].update(p)

    def _plan_group(self, g, ids):
        if g.has_subgroups:
            return (g, [self._plan_group(sg, ids) for sg in g.equations])
        return (g, _CGroup(g, ids, self.arrays, self.ckernel.kind))

    def _leaves(self, plan):
        for g, item in plan:
            if isinstance(item, list):
                for leaf in self._leaves(item):
                    yield leaf
            else:
                yield item

    # -- c_acceleration_eval protocol ---------------------------------------
    def set_nnps(self, nnps):
        self.nnps = nnps

    def update_particle_arrays(self, particle_arrays):
        self._setup(list(particle_arrays))

    def compute(self, t, dt):
        if self.nnps is None:
            raise RuntimeError('HipAccelerationEval.compute: call set_nnps first')
        if self.sync == 'auto':
            self.push_inputs()
        for entry in self.plan:
            self._run(entry, t, dt)
        if self.sync == 'auto':
            self.pull_outputs()

    # -- an evaluation in two halves around the arrival of the ghosts ---------
    def can_split(self):
        """May `compute` run as `compute_begin` (before the ghosts of a slab
        exchange arrive) + `compute_end` (after `nnps.update_ghosts()`)?  Yes for
        a device-resident plan of plain leaf groups with hand-written kernels
        over ONE particle array whose pair loops visit real particles only
        (ghosts are sources, never destinations): equations without sources run
        over the particles present in each half, the pair loops of the wavefronts
        that cannot reach a ghost run in the first half, the others in the
        second (sph_group.phase)."""
        if self.sync != 'manual' or len(self.particle_arrays) != 1:
            return False
        pair_groups = []
        for k, (g, item) in enumerate(self.plan):
            if isinstance(item, list) or not _plain_leaf(g, item) or _ranged(g):
                return False
            for u in item.units:
                if u.has_pair:
                    if not g.real:
                        return False
                    pair_groups.append(k)
        # nothing may read what a pair loop writes before the SECOND half has completed it for the particles near the
        # faces: exactly one group with pair loops, and it is the last one (equations without sources of that group run
        # before its loops).  WCSPH's [EOS | rates] qualifies; the elastic set (stress rates from the velocity gradient
        # of the same evaluation) does not.
        return len(set(pair_groups)) == 1 and pair_groups[0] == len(self.plan) - 1

    def compute_begin(self, t, dt):
        if not self.can_split():
            raise RuntimeError('this evaluation cannot be split around the ghost exchange (can_split())')
        self._phase(1, t, dt)

    def compute_end(self, t, dt):
        self._phase(2, t, dt)

    def _phase(self, phase, t, dt):
        for g, item in self.plan:
            item.refresh_range()
            for u in item.units:
                u.run(self, t, dt, phase)

    # -- data movement --------------------------------------------------------
    def push_inputs(self):
        # positions and h were pushed by nnps.update() (HipNNPS sync=True) and
        # define the cell grid: re-pushing them would invalidate it
        skip = ('x', 'y', 'z', 'h') if getattr(self.nnps, 'sync', False) else ()
        for name, props in self.inputs.items():
            pa = self.arrays[name]
            have = [p for p in sorted(props)
                    if ((p in pa.properties and dev.prop_id(p) >= 0)
                        or self.helpers[name]._component(p) is not None)
                    and p not in skip]
            self.helpers[name].push(*have)

    def pull_outputs(self):
        # the hand-written kernels' tables list every property an equation
        # touches on its destination, inputs included: those never change
        never_written = ('x', 'y', 'z', 'h', 'm', 'u', 'v', 'w', 'uhat', 'vhat', 'what')
        for name in set(self.outputs) | set(self.outputs_exact):
            pa = self.arrays[name]
            props = set(p for p in self.outputs.get(name, ()) if p not in never_written)
            props |= set(self.outputs_exact.get(name, ()))
            out = [p for p in sorted(props)
                   if (p in pa.properties and dev.prop_id(p) >= 0)
                   or self.helpers[name]._component(p) is not None]
            self.helpers[name].pull(*out)

    # -- group execution (acceleration_eval_cython.mako:291-363) -------------
    def _run(self, entry, t, dt):
        g, item = entry
        if g.condition is not None and not g.condition(t, dt):
            return
        if g.iterate:
            count = 1
            while True:
                self._run_once(g, item, t, dt)
                done = self._converged(g)
                if count >= g.min_iterations and \
                        (done or count == g.max_iterations):
                    break
                count += 1
        else:
            self._run_once(g, item, t, dt)

    def _converged(self, g):
        ok = True
        for eq in self._group_equations(g):
            ok &= eq.converged() > 0
        return ok

    def _group_equations(self, g):
        if g.has_subgroups:
            for sg in g.equations:
                for eq in self._group_equations(sg):
                    yield eq
        else:
            for eq in g.equations:
                yield eq

    def _host_hook(self, fn, *args):
        """Run a host-side callback (Group.pre/post, Equation.py_initialize,
        Equation.reduce).  With sync='auto' the host arrays are authoritative:
        bring them up to date first and push whatever the callback changed
        (the reference's own GPU tests do the same pull/push by hand inside
        their callbacks, test_acceleration_eval.py:178-191)."""
        if self.sync == 'auto':
            self.pull_outputs()
        fn(*args)
        if self.sync == 'auto':
            self.push_inputs()

    def _run_once(self, g, item, t, dt):
        if g.pre:
            self._host_hook(g.pre)
        if isinstance(item, list):
            for sub in item:
                sg = sub[0]
                if sg.condition is not None and not sg.condition(t, dt):
                    continue
                self._run_once(sg, sub[1], t, dt)
        else:
            for eq in g.equations:
                if hasattr(eq, 'py_initialize'):
                    self._host_hook(eq.py_initialize, self.arrays[eq.dest], t, dt)
            item.refresh_range()
            item.run(self, t, dt)
            for eq in g.equations:
                if hasattr(eq, 'reduce'):
                    self._host_hook(eq.reduce, self.arrays[eq.dest], t, dt)
        if g.update_nnps:
            self.nnps.update_domain()
            self.nnps.update()
        if g.post:
            self._host_hook(g.post)


class AccelerationEvalHipHelper(object):
    """Helper with the protocol ``SPHCompiler`` expects
    (sph_compiler.py:27-59; Cython twin:
    acceleration_eval_cython_helper.py:113-181).  Nothing is generated or
    compiled at run time: the kernels are prebuilt in libsphhip.so."""

    def __init__(self, acceleration_eval, ctx=None, sync='auto'):
        self.object = acceleration_eval
        self.ctx = ctx
        self.sync = sync

    def get_code(self):
        return '# HIP backend: hand-written kernels in libsphhip.so\n'

    def compile(self, code):
        dev.load_library()
        return None

    def setup_compiled_module(self, module=None):
        obj = HipAccelerationEval(self.object, ctx=self.ctx, sync=self.sync)
        self.object.set_compiled_object(obj)
        return obj


class SPHCompiler(object):
    """sph_compiler.py:1-94 for the HIP backend (acceleration evals only)."""

    def __init__(self, acceleration_evals, integrator=None, ctx=None,
                 sync='auto'):
        if not isinstance(acceleration_evals, (list, tuple)):
            acceleration_evals = [acceleration_evals]
        self.acceleration_evals = list(acceleration_evals)
        self.integrator = integrator
        self.ctx = ctx
        self.helpers = [AccelerationEvalHipHelper(a, ctx, sync)
                        for a in self.acceleration_evals]

    def compile(self):
        for h in self.helpers:
            h.compile(h.get_code())
            h.setup_compiled_module(None)
        if self.integrator is not None:
            # sph_compiler.py:27-59: the integrator gets its compiled object too
            # (stage sweeps = sph_integrate_stage); the caller hands it the NNPS
            # afterwards with integrator.set_nnps(nnps), as Solver.setup does
            from .integrator import HipIntegrator
            integ = self.integrator
            integ.set_acceleration_evals(self.acceleration_evals)
            integ.set_compiled_object(HipIntegrator(
                integ, self.acceleration_evals[0].c_acceleration_eval, self.ctx))
