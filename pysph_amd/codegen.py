"""Equation bodies -> HIP: the generated-family path of the HIP backend.

The reference turns every ``Equation``'s Python methods into C through
``compyle`` (pysph/sph/equation.py:748-892 ``get_*_code``,
acceleration_eval_cython_helper.py:147-305) and drops them into the loop nest
of acceleration_eval_cython.mako:10-154.  The hand-written families of
``libsphhip`` cover the benchmark equation sets; for every other equation this
module does the analogous step for gfx950:

* the ``initialize / loop / post_loop`` methods of all equations acting on one
  destination in one group are parsed (``ast``) and re-emitted as the
  ``load / pair / finish`` members of a *family struct* for the pair-loop
  skeleton ``csrc/sph_pair.h`` (the same aggregated two-phase kernel the
  hand-written families use -- cell-ordered records, fp32 prefilter, exact
  fp64 criterion, register accumulators, one write per output);
* the struct is compiled by ``hipcc --offload-arch=gfx950`` into a shared
  object of its own under ``pysph_amd/libsphhip_gen/`` (cached by source hash) that
  exports ``sphgen_launch``;
* ``HipAccelerationEval`` hands that function pointer and the property / source
  lists to ``sph_eval_generated`` (include/sphhip.h).

Supported Python subset (what the reference's equations use): float
arithmetic, comparisons, ``and/or/not``, conditional expressions, ``if/elif/
else``, ``for i in range(..)``, local scalars, ``declare('matrix(n)')`` local
arrays, augmented assignment, bare ``return``, the libm calls compyle maps
(``sqrt pow exp log sin cos tan tanh fabs abs max min floor ceil atan2``),
``M_PI``/``pi``, ``self.<scalar attribute>`` (copied by value when the family is
built, like equation.py:885-892 does), ``d_<prop>[d_idx]``,
``s_<prop>[s_idx]``, components of strided properties ``d_<prop>[d_idx*S + k]``,
``d_<constant>[k]`` and the precomputed symbols ``XIJ
VIJ R2IJ RIJ HIJ RHOIJ RHOIJ1 EPS WIJ DWIJ WI WJ DWI DWJ WDP t dt``
(equation.py:188-297).  Anything else raises ``CodegenError`` -- the equation
then has to be hand-written or simplified; there is no silent fallback.
"""
import ast
import ctypes as C
from collections import OrderedDict
import hashlib
import inspect
import os
import re
import subprocess
import textwrap

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, 'csrc')
INCLUDE = os.path.join(os.path.dirname(_HERE), 'include')
# the directory name sorts behind 'libsphhip.so' in an alphabetical listing of the loaded
# libraries (a truncated listing then still shows the library itself)
GEN_DIR = os.path.join(_HERE, 'libsphhip_gen')

VEC_SYMBOLS = ('XIJ', 'VIJ', 'DWIJ', 'DWI', 'DWJ')
SCALAR_SYMBOLS = ('R2IJ', 'RIJ', 'HIJ', 'RHOIJ', 'RHOIJ1', 'EPS', 'WIJ', 'WI',
                  'WJ', 'WDP', 'GHI', 'GHJ', 'GHIJ', 't', 'dt')
MATH_1 = {'sqrt': 'sqrt', 'exp': 'exp', 'log': 'log', 'sin': 'sin',
          'cos': 'cos', 'tan': 'tan', 'tanh': 'tanh', 'fabs': 'fabs',
          'abs': 'fabs', 'floor': 'floor', 'ceil': 'ceil', 'log10': 'log10',
          'asin': 'asin', 'acos': 'acos', 'atan': 'atan', 'sinh': 'sinh',
          'cosh': 'cosh', 'erf': 'erf'}
MATH_2 = {'pow': 'pow', 'atan2': 'atan2', 'fmod': 'fmod'}
CONSTANTS = {'M_PI': 'M_PI', 'pi': 'M_PI', 'M_1_PI': 'M_1_PI',
             'M_2_SQRTPI': 'M_2_SQRTPI', 'M_PI_2': 'M_PI_2', 'INFINITY': 'INFINITY'}
METHODS = ('initialize', 'initialize_pair', 'loop', 'loop_all', 'post_loop')
UNSUPPORTED_METHODS = ()


class CodegenError(Exception):
    pass


# identifiers a Python local may not keep in the generated C++: keywords and the
# names the emitted skeleton itself uses around the bodies
_RESERVED = set("""alignas alignof and and_eq asm auto bitand bitor bool break case catch char
    class compl const constexpr const_cast continue decltype default delete do double
    dynamic_cast else enum explicit export extern false float for friend goto if inline int
    long mutable namespace new noexcept not not_eq nullptr operator or or_eq private
    protected public register reinterpret_cast return short signed sizeof static
    static_assert static_cast struct switch template this thread_local throw true try
    typedef typeid typename union unsigned using virtual void volatile wchar_t while xor
    xor_eq restrict a D g s sj pi pj fl o PAR KK UH A NBRS N_NBRS r2 hi2 hj2 gi gj
    tg_ tgi_ tgj_ uint32_t size_t""".split())


def _cn(name):
    """C++ spelling of a Python local / loop variable / helper argument"""
    return name + '_' if name in _RESERVED else name


_SKELETON = None


def _skeleton_digest():
    global _SKELETON
    if _SKELETON is None:
        h = hashlib.sha1()
        for path in (os.path.join(CSRC, 'sph_pair.h'), os.path.join(CSRC, 'sph_kernels.h'),
                     os.path.join(INCLUDE, 'sphhip.h')):
            with open(path, 'rb') as f:
                h.update(f.read())
        _SKELETON = h.hexdigest()
    return _SKELETON


def has_python_body(eq):
    """True when the object carries translatable method bodies (a reference
    equation, or a user's own ``Equation`` subclass)."""
    return any(callable(getattr(type(eq), m, None)) for m in METHODS)


def method_properties(eq):
    """(destination properties, source properties) named in the argument
    lists of the equation's methods -- what the reference's
    check_equation_array_properties inspects (acceleration_eval.py:32-73)."""
    d, s = set(), set()
    for m in METHODS:
        fn = getattr(type(eq), m, None)
        if fn is None:
            continue
        for a in inspect.getfullargspec(fn).args:
            if a.startswith('d_') and a != 'd_idx':
                d.add(a[2:])
            elif a.startswith('s_') and a != 's_idx':
                s.add(a[2:])
    return sorted(d), sorted(s)


def initialize_only_resets(eq, props):
    """True when ``eq.initialize`` does nothing but assign literals to
    ``d_<p>[d_idx]`` with every p in `props` -- i.e. it only repeats resets that
    another equation on the same destination performs anyway."""
    fdef = _method_ast(eq, 'initialize')
    if fdef is None:
        return True
    for st in fdef.body:
        if isinstance(st, ast.Expr) and isinstance(st.value, ast.Constant):
            continue
        if not (isinstance(st, ast.Assign) and len(st.targets) == 1
                and isinstance(st.value, ast.Constant)
                and isinstance(st.targets[0], ast.Subscript)
                and isinstance(st.targets[0].value, ast.Name)
                and st.targets[0].value.id.startswith('d_')
                and st.targets[0].value.id[2:] in props):
            return False
    return True


def _method_ast(eq, name):
    fn = getattr(type(eq), name, None)
    if fn is None:
        return None
    try:
        src = textwrap.dedent(inspect.getsource(fn))
    except (OSError, TypeError) as e:
        raise CodegenError('%s.%s: source not available (%s)' %
                           (type(eq).__name__, name, e))
    tree = ast.parse(src)
    return tree.body[0]


class _Body(object):
    """Translation of one method of one equation."""

    def __init__(self, fam, eq, eq_index, kind, fdef):
        self.fam = fam
        self.eq = eq
        self.k = eq_index
        self.kind = kind            # 'initialize' | 'loop' | 'post_loop'
        self.pair = kind == 'loop' and not fam.is_no_source(eq)
        self.all_nbrs = kind == 'loop_all'
        # whole source arrays are visible (any index), not one packed neighbour
        self.raw_src = kind in ('loop_all', 'initialize_pair')
        self.fdef = fdef
        self.locals = {}            # name -> ('double', None) | ('array', n) | ('int', None)
        self.loop_vars = set()
        self.unrolled = {}          # loop variables of statically unrolled loops -> value
        self.affine_ints = {}       # int locals with a known value a*d_idx + b*s_idx + c
        self.where = '%s.%s' % (type(eq).__name__, kind)
        self.lines = []
        fam._writes = set()
        self._emit_block(fdef.body, 1)
        self.writes = fam._writes
        fam._writes = None

    # -- helpers -----------------------------------------------------------
    def err(self, node, msg):
        raise CodegenError('%s line %d: %s' % (self.where,
                                               getattr(node, 'lineno', 0), msg))

    def _self_param(self, node, attr):
        if not hasattr(self.eq, attr):
            self.err(node, 'self.%s is not set on the equation object' % attr)
        val = getattr(self.eq, attr)
        if isinstance(val, bool):
            val = 1.0 if val else 0.0
        if not isinstance(val, (int, float)):
            try:
                val = float(val)
            except Exception:
                self.err(node, 'self.%s is not a scalar (%r)' % (attr, type(val)))
        return self.fam.param(('self', self.k, attr), float(val))

    # -- integer index algebra ----------------------------------------------
    def _assign_counts(self):
        if not hasattr(self, '_acount'):
            cnt = {}
            for node in ast.walk(self.fdef):
                tgts = []
                if isinstance(node, ast.Assign):
                    vals = node.value.elts if isinstance(node.value, ast.Tuple) else [node.value]
                    if all(self._declare_call(v) is not None for v in vals):
                        continue                    # x = declare(...) is a declaration
                    tgts = node.targets
                elif isinstance(node, (ast.AugAssign, ast.For)):
                    tgts = [node.target]
                for t_ in tgts:
                    for e in (t_.elts if isinstance(t_, ast.Tuple) else [t_]):
                        if isinstance(e, ast.Name):
                            cnt[e.id] = cnt.get(e.id, 0) + 1
            self._acount = cnt
        return self._acount

    def _affine(self, n):
        """integer expression as ({symbol: coefficient}, constant) or None.
        Symbols: d_idx, s_idx (pair loops) and run-time int names."""
        if isinstance(n, ast.Constant) and isinstance(n.value, int) and not isinstance(n.value, bool):
            return {}, n.value
        if isinstance(n, ast.Name):
            if n.id in self.unrolled:
                return {}, self.unrolled[n.id]
            if n.id in self.affine_ints:
                syms, c = self.affine_ints[n.id]
                return dict(syms), c
            if n.id == 'd_idx' or (n.id == 's_idx' and self.pair):
                return {n.id: 1}, 0
            if n.id in self.loop_vars or self.locals.get(n.id, ('',))[0] == 'int':
                return {n.id: 1}, 0
            return None
        if isinstance(n, ast.Attribute) and isinstance(n.value, ast.Name) and n.value.id == 'self' \
                and self.eq is not None and isinstance(getattr(self.eq, n.attr, None), int) \
                and not isinstance(getattr(self.eq, n.attr), bool):
            return {}, int(getattr(self.eq, n.attr))
        if isinstance(n, ast.UnaryOp) and isinstance(n.op, ast.USub):
            a = self._affine(n.operand)
            return None if a is None else (dict((k, -v) for k, v in a[0].items()), -a[1])
        if isinstance(n, ast.BinOp) and isinstance(n.op, (ast.Add, ast.Sub, ast.Mult)):
            a, b = self._affine(n.left), self._affine(n.right)
            if a is None or b is None:
                return None
            if isinstance(n.op, ast.Mult):
                if a[0] and b[0]:
                    return None
                if a[0]:
                    a, b = b, a
                return dict((k, v * a[1]) for k, v in b[0].items() if v * a[1]), a[1] * b[1]
            sg = 1 if isinstance(n.op, ast.Add) else -1
            syms = dict(a[0])
            for k, v in b[0].items():
                syms[k] = syms.get(k, 0) + sg * v
            return dict((k, v) for k, v in syms.items() if v), a[1] + sg * b[1]
        return None

    def _const_int(self, n):
        a = self._affine(n)
        return a[1] if a is not None and not a[0] else None

    def _component(self, sl, idx_name):
        """``S*<idx_name> + K`` (any arrangement that reduces to it, K from
        literals / unrolled loop variables / known ints) -> (S, K); S = 1,
        K = 0 is the plain scalar access"""
        a = self._affine(sl)
        if a is None or set(a[0]) != {idx_name}:
            return None
        S, K = a[0][idx_name], a[1]
        if S >= 1 and 0 <= K < S:
            return S, K
        return None

    def _mentions_in_property_index(self, body, var):
        for st in body:
            for node in ast.walk(st):
                if isinstance(node, ast.Subscript) and isinstance(node.value, ast.Name) and \
                        (node.value.id.startswith('d_') or node.value.id.startswith('s_')) and \
                        node.value.id not in self.locals:
                    sl = node.slice
                    names = set(x.id for x in ast.walk(sl) if isinstance(x, ast.Name))
                    if var in names:
                        return True
                    # through an int local that was computed from the loop variable
                    for nm in names:
                        if nm in self._depends_on(var):
                            return True
        return False

    def _depends_on(self, var):
        """int locals assigned (anywhere in the method) from expressions that
        mention `var`"""
        out, changed = set(), True
        while changed:
            changed = False
            for node in ast.walk(self.fdef):
                if isinstance(node, ast.Assign) and len(node.targets) == 1 and \
                        isinstance(node.targets[0], ast.Name) and node.targets[0].id not in out:
                    names = set(x.id for x in ast.walk(node.value) if isinstance(x, ast.Name))
                    if var in names or names & out:
                        out.add(node.targets[0].id)
                        changed = True
        return out

    # -- expressions -------------------------------------------------------
    def expr(self, n):
        if isinstance(n, ast.Constant):
            if isinstance(n.value, bool):
                return '1.0' if n.value else '0.0'
            if isinstance(n.value, (int, float)):
                return repr(float(n.value))
            self.err(n, 'constant %r' % (n.value,))
        if isinstance(n, ast.Name):
            return self.name(n)
        if isinstance(n, ast.Attribute):
            if isinstance(n.value, ast.Name) and n.value.id == 'self':
                return self._self_param(n, n.attr)
            if isinstance(n.value, ast.Name) and n.value.id in ('math', 'np', 'numpy', 'M'):
                if n.attr in CONSTANTS:
                    return CONSTANTS[n.attr]
            self.err(n, 'attribute access %s' % ast.dump(n))
        if isinstance(n, ast.Subscript):
            return self.subscript(n, store=False)
        if isinstance(n, ast.BinOp):
            a, b = self.expr(n.left), self.expr(n.right)
            if isinstance(n.op, ast.Add):
                return '(%s + %s)' % (a, b)
            if isinstance(n.op, ast.Sub):
                return '(%s - %s)' % (a, b)
            if isinstance(n.op, ast.Mult):
                return '(%s * %s)' % (a, b)
            if isinstance(n.op, ast.Div):
                return '(%s / %s)' % (a, b)
            if isinstance(n.op, ast.Mod):
                return 'fmod(%s, %s)' % (a, b)
            if isinstance(n.op, ast.Pow):
                if isinstance(n.right, ast.Constant) and n.right.value == 2:
                    return '(%s * %s)' % (a, a)
                return 'pow(%s, %s)' % (a, b)
            self.err(n, 'operator %s' % type(n.op).__name__)
        if isinstance(n, ast.UnaryOp):
            v = self.expr(n.operand)
            if isinstance(n.op, ast.USub):
                return '(-%s)' % v
            if isinstance(n.op, ast.UAdd):
                return v
            if isinstance(n.op, ast.Not):
                return '(!(%s))' % v
            self.err(n, 'unary operator')
        if isinstance(n, ast.BoolOp):
            op = ' && ' if isinstance(n.op, ast.And) else ' || '
            return '(' + op.join('(%s)' % self.expr(v) for v in n.values) + ')'
        if isinstance(n, ast.Compare):
            ops = {ast.Lt: '<', ast.Gt: '>', ast.LtE: '<=', ast.GtE: '>=',
                   ast.Eq: '==', ast.NotEq: '!='}
            parts, left = [], n.left
            for op, right in zip(n.ops, n.comparators):
                if type(op) not in ops:
                    self.err(n, 'comparison %s' % type(op).__name__)
                parts.append('(%s %s %s)' % (self.expr(left), ops[type(op)],
                                             self.expr(right)))
                left = right
            return '(' + ' && '.join(parts) + ')'
        if isinstance(n, ast.IfExp):
            return '((%s) ? (%s) : (%s))' % (self.expr(n.test), self.expr(n.body),
                                             self.expr(n.orelse))
        if isinstance(n, ast.Call):
            return self.call(n)
        self.err(n, 'expression %s' % type(n).__name__)

    def call(self, n):
        f = n.func
        if isinstance(f, ast.Attribute) and isinstance(f.value, ast.Name) and f.value.id == 'SPH_KERNEL':
            if not (self.all_nbrs or self.pair):
                self.err(n, 'SPH_KERNEL is only available in loop and loop_all')
            args = [_cn(a.id) if (isinstance(a, ast.Name) and (
                        self.locals.get(a.id, ('',))[0] in ('array', 'arrayarg') or a.id in VEC_SYMBOLS))
                    else self.expr(a) for a in n.args]
            if f.attr == 'kernel' and len(args) == 3:       # kernel(xij, rij, h)
                return 'gen_kernel_w<KK>(%s, %s, a)' % (args[1], args[2])
            if f.attr == 'dwdq' and len(args) == 2:
                return 'gen_kernel_dwdq<KK>(%s, %s, a)' % (args[0], args[1])
            if f.attr == 'get_deltap' and not args:
                return 'a.k.deltap'
            self.err(n, 'SPH_KERNEL.%s' % f.attr)
        fname = f.id if isinstance(f, ast.Name) else (
            f.attr if isinstance(f, ast.Attribute) and isinstance(f.value, ast.Name)
            and f.value.id in ('math', 'np', 'numpy', 'M') else None)
        if fname is None:
            self.err(n, 'call %s' % ast.dump(f))
        helper = None
        if isinstance(f, ast.Name) and fname not in MATH_1 and fname not in MATH_2 \
                and fname not in ('max', 'min', 'float', 'double'):
            helper = self.fam.helper(self, n, fname)
        if helper is not None:
            # plain Python helper functions (Equation._get_helpers_,
            # equation.py:860-872).  Argument types follow the defaults, as in
            # the reference's translator: a list is a double array, an int an
            # int, anything else a double; the result is a double.
            names = helper.argnames
            if len(n.args) > len(names):
                self.err(n, '%s(): too many arguments' % fname)

            def passed(i, node):
                kind = helper.argtypes[i]
                if kind == 'array':
                    if isinstance(node, ast.Name) and \
                            self.locals.get(node.id, ('',))[0] in ('array', 'arrayarg'):
                        return _cn(node.id)
                    if isinstance(node, ast.Name) and node.id in VEC_SYMBOLS and self.pair:
                        self.fam.use_symbol(node.id)
                        self.fam.sym_written.add(node.id)    # the callee may write it
                        return node.id
                    self.err(n, '%s(): argument %s must be a local matrix' % (fname, names[i]))
                if kind == 'int':
                    return self.index(node)
                return self.expr(node)

            slots = [passed(i, a) for i, a in enumerate(n.args)]
            slots += [None] * (len(names) - len(slots))
            for kw in n.keywords:
                if kw.arg not in names or slots[names.index(kw.arg)] is not None:
                    self.err(n, '%s(): keyword %s' % (fname, kw.arg))
                slots[names.index(kw.arg)] = passed(names.index(kw.arg), kw.value)
            for i, v in enumerate(slots):
                if v is None:
                    if helper.defaults[i] is None:
                        self.err(n, '%s(): argument %s missing' % (fname, names[i]))
                    slots[i] = helper.defaults[i]
            return '%s(%s)' % (helper.cname, ', '.join(slots))
        if n.keywords:
            self.err(n, 'call %s' % ast.dump(f))
        args = [self.expr(a) for a in n.args]
        if fname in MATH_1 and len(args) == 1:
            return '%s(%s)' % (MATH_1[fname], args[0])
        if fname in MATH_2 and len(args) == 2:
            return '%s(%s, %s)' % (MATH_2[fname], args[0], args[1])
        if fname in ('max', 'min') and len(args) >= 2:
            fn = 'fmax' if fname == 'max' else 'fmin'
            out = args[0]
            for a in args[1:]:
                out = '%s(%s, %s)' % (fn, out, a)
            return out
        if fname in ('float', 'double') and len(args) == 1:
            return '((double)(%s))' % args[0]
        self.err(n, 'call to %s()' % fname)

    def name(self, n):
        v = n.id
        if v in self.unrolled:
            return repr(float(self.unrolled[v]))
        if v in self.loop_vars:
            return _cn(v)
        if v in self.locals:
            return _cn(v)
        if v == 'N_NBRS' and self.all_nbrs:
            return 'N_NBRS'
        if v in CONSTANTS:
            return CONSTANTS[v]
        if v in ('True', 'False'):
            return '1.0' if v == 'True' else '0.0'
        if v in SCALAR_SYMBOLS:
            if v not in ('t', 'dt') and not self.pair:
                self.err(n, 'pair symbol %s outside a pair loop' % v)
            self.fam.use_symbol(v)
            return v
        if v in VEC_SYMBOLS:
            self.err(n, 'vector symbol %s must be subscripted' % v)
        if v == 'd_idx' and self.raw_src:
            return '((double)d_idx)'            # compared with NBRS[k] in loop_all bodies
        if v == 'd_idx' and not self.pair and self.kind in ('initialize', 'loop', 'post_loop'):
            return '((double)o)'                # load()/finish(): o is the particle's own index
        if v in ('d_idx', 's_idx'):
            self.err(n, '%s may only index a property array' % v)
        self.err(n, 'unknown name %r (locals must be assigned before use)' % v)

    def index(self, n):
        """integer index expression (loop variables, literals, + - *)"""
        if isinstance(n, ast.Constant) and isinstance(n.value, int):
            return str(n.value)
        if isinstance(n, ast.Name) and n.id in self.unrolled:
            return str(self.unrolled[n.id])
        if isinstance(n, ast.Name) and n.id in self.affine_ints and not self.affine_ints[n.id][0]:
            return str(self.affine_ints[n.id][1])
        if isinstance(n, ast.Name) and (n.id in self.loop_vars or
                                        self.locals.get(n.id, ('', 0))[0] == 'int'):
            return _cn(n.id)
        if isinstance(n, ast.Name) and n.id == 'N_NBRS' and self.all_nbrs:
            return 'N_NBRS'
        if isinstance(n, ast.Name) and n.id == 'd_idx' and self.raw_src:
            return '((int)d_idx)'
        if isinstance(n, ast.Subscript) and isinstance(n.value, ast.Name) and n.value.id == 'NBRS' \
                and self.all_nbrs:
            return self.subscript(n, store=False)           # s_m[NBRS[i]]
        if isinstance(n, ast.Attribute) and isinstance(n.value, ast.Name) and n.value.id == 'self' \
                and isinstance(getattr(self.eq, n.attr, None), int):
            return str(int(getattr(self.eq, n.attr)))       # frozen at build time
        if isinstance(n, ast.BinOp) and isinstance(n.op, (ast.Add, ast.Sub, ast.Mult)):
            op = {ast.Add: '+', ast.Sub: '-', ast.Mult: '*'}[type(n.op)]
            return '(%s %s %s)' % (self.index(n.left), op, self.index(n.right))
        self.err(n, 'array index must be an integer literal or a loop variable')

    def _strided(self, sl, idx_name):
        """``<idx>*S + K`` / ``S*<idx> + K`` with literal S, K -> (S, K): one
        component of a multi-component property (ParticleArray stride,
        particle_array.pyx add_property(stride=...)); None otherwise."""
        k = 0
        if isinstance(sl, ast.BinOp) and isinstance(sl.op, ast.Add):
            if isinstance(sl.right, ast.Constant) and isinstance(sl.right.value, int):
                k, sl = sl.right.value, sl.left
            elif isinstance(sl.left, ast.Constant) and isinstance(sl.left.value, int):
                k, sl = sl.left.value, sl.right
            else:
                return None
        if isinstance(sl, ast.BinOp) and isinstance(sl.op, ast.Mult):
            a, b = sl.left, sl.right
            if isinstance(b, ast.Name):
                a, b = b, a
            if isinstance(a, ast.Name) and a.id == idx_name and \
                    isinstance(b, ast.Constant) and isinstance(b.value, int) and b.value > 1 \
                    and 0 <= k < b.value:
                return b.value, k
        return None

    def _strided_any(self, sl):
        """``<int index>*S + K`` with literal S, K -> (S, index text, K)"""
        k = 0
        if isinstance(sl, ast.BinOp) and isinstance(sl.op, ast.Add):
            if isinstance(sl.right, ast.Constant) and isinstance(sl.right.value, int):
                k, sl = sl.right.value, sl.left
            elif isinstance(sl.left, ast.Constant) and isinstance(sl.left.value, int):
                k, sl = sl.left.value, sl.right
            else:
                return None
        if isinstance(sl, ast.BinOp) and isinstance(sl.op, ast.Mult):
            a, b = sl.left, sl.right
            if isinstance(a, ast.Constant):
                a, b = b, a
            if isinstance(b, ast.Constant) and isinstance(b.value, int) and b.value > 1 \
                    and 0 <= k < b.value and isinstance(a, ast.Name):
                return b.value, self.index(a), k
        return None

    def subscript(self, n, store):
        if not isinstance(n.value, ast.Name):
            self.err(n, 'subscript of %s' % type(n.value).__name__)
        base = n.value.id
        sl = n.slice
        if isinstance(sl, ast.Index):      # py<3.9
            sl = sl.value
        if base.startswith('d_'):
            prop = base[2:]
            if isinstance(sl, ast.Name) and sl.id == 'd_idx':
                return self.fam.dest_prop(prop, store)
            sk = self._strided(sl, 'd_idx') or self._component(sl, 'd_idx')
            if sk is not None and sk[0] == 1:
                return self.fam.dest_prop(prop, store)
            if sk is not None:
                self.fam.note_stride(prop, sk[0], self, n)
                return self.fam.dest_prop('%s__%d' % (prop, sk[1]), store)
            if self.fam.is_dest_constant(prop):
                if store:
                    self.err(n, 'constants are read-only here')
                if not (isinstance(sl, ast.Constant) and isinstance(sl.value, int)):
                    self.err(n, 'constant %s needs a literal index' % base)
                return self.fam.param(('const', prop, sl.value), None)
            if not store and not self.pair and isinstance(sl, ast.Name) and \
                    self.locals.get(sl.id, ('',))[0] == 'int':
                # d_rho[idx] with a run-time index: another particle of the SAME array, read
                # from memory as it is when the launch reads it (the ghost-update equations,
                # e.g. iisph.py:250-261, copy from the particle a ghost is an image of)
                return '%s[%s]' % (self.fam.raw_dest_prop(prop), _cn(sl.id))
            self.err(n, '%s must be indexed with d_idx' % base)
        if base == 'NBRS' and self.all_nbrs:
            if store:
                self.err(n, 'NBRS is read-only')
            return '((int)NBRS[%s])' % self.index(sl)
        if base.startswith('s_') and self.raw_src:
            if store:
                self.err(n, 'source arrays are read-only (gather formulation)')
            sk = self._strided_any(sl)
            if sk is None:
                a = self._affine(sl)
                if a is not None and len(a[0]) == 1:
                    (nm, S), = a[0].items()
                    if S > 1 and 0 <= a[1] < S:
                        sk = (S, '((int)d_idx)' if nm == 'd_idx' else _cn(nm), a[1])
            if sk is not None:
                self.fam.note_stride(base[2:], sk[0], self, n)
                return 'S_%s[%s]' % (self.fam.raw_src_prop('%s__%d' % (base[2:], sk[2])), sk[1])
            return 'S_%s[%s]' % (self.fam.raw_src_prop(base[2:]), self.index(sl))
        if base.startswith('s_'):
            prop = base[2:]
            if store:
                self.err(n, 'source arrays are read-only (gather formulation)')
            if not self.pair:
                self.err(n, '%s outside a pair loop' % base)
            if isinstance(sl, ast.Name) and sl.id == 's_idx':
                return self.fam.src_prop(prop)
            sk = self._strided(sl, 's_idx') or self._component(sl, 's_idx')
            if sk is not None and sk[0] == 1:
                return self.fam.src_prop(prop)
            if sk is not None:
                self.fam.note_stride(prop, sk[0], self, n)
                return self.fam.src_prop('%s__%d' % (prop, sk[1]))
            self.err(n, '%s must be indexed with s_idx' % base)
        if base in VEC_SYMBOLS:
            if store and self.pair:
                # e.g. GradientCorrection (kernel_correction.py:95-125) rewrites
                # DWIJ for the equations after it in the group
                self.fam.sym_written.add(base)
            elif store:
                self.err(n, 'precomputed symbols are read-only here')
            if not self.pair:
                self.err(n, 'pair symbol %s outside a pair loop' % base)
            self.fam.use_symbol(base)
            return '%s[%s]' % (_cn(base), self.index(sl))
        if base in self.locals and self.locals[base][0] in ('array', 'arrayarg'):
            return '%s[%s]' % (_cn(base), self.index(sl))
        self.err(n, 'subscript of unknown array %r' % base)

    # -- statements --------------------------------------------------------
    def _emit(self, ind, text):
        self.lines.append('    ' * ind + text)

    def _emit_block(self, body, ind):
        for st in body:
            self._emit_stmt(st, ind)

    def _declare_call(self, value):
        """x = declare('matrix(3)') / declare('double')"""
        if isinstance(value, ast.Call) and isinstance(value.func, ast.Name) \
                and value.func.id == 'declare' and value.args \
                and isinstance(value.args[0], ast.Constant):
            spec = str(value.args[0].value).replace(' ', '')
            if spec.startswith('matrix(') and spec.endswith(')'):
                dims = spec[7:-1].strip('()').split(',')
                n = 1
                for d in dims:
                    if d:
                        n *= int(d)
                return ('array', n)
            if spec in ('int', 'long', 'unsigned int', 'unsignedint'):
                return ('int', None)
            if spec in ('double', 'float'):
                return ('double', None)
        return None

    def _emit_stmt(self, st, ind):
        if isinstance(st, ast.Expr):
            if isinstance(st.value, ast.Constant):      # docstring
                return
            v = st.value
            if isinstance(v, ast.Call) and isinstance(v.func, ast.Attribute) and \
                    isinstance(v.func.value, ast.Name) and v.func.value.id == 'SPH_KERNEL' and \
                    v.func.attr == 'gradient' and len(v.args) == 4 and (self.all_nbrs or self.pair) and \
                    all(isinstance(v.args[k], ast.Name) and (
                        self.locals.get(v.args[k].id, ('',))[0] in ('array', 'arrayarg')
                        or (k == 0 and v.args[k].id in VEC_SYMBOLS)) for k in (0, 3)):
                # gradient(xij, rij, h, grad): kernels.py:126-137
                self._emit(ind, 'gen_kernel_gradient<KK>(%s, %s, %s, %s, a);' % (
                    _cn(v.args[0].id), self.expr(v.args[1]), self.expr(v.args[2]), _cn(v.args[3].id)))
                return
            if isinstance(v, ast.Call) and isinstance(v.func, ast.Name) and \
                    self.fam.helper(self, v, v.func.id) is not None:
                self._emit(ind, '(void)%s;' % self.call(v))      # result ignored
                return
            self.err(st, 'expression statement')
        if isinstance(st, ast.Pass):
            return
        if isinstance(st, (ast.Continue, ast.Break)):
            if not getattr(self, '_runtime_loops', 0):
                self.err(st, '%s outside a run-time loop' % type(st).__name__.lower())
            self._emit(ind, 'continue;' if isinstance(st, ast.Continue) else 'break;')
            return
        if isinstance(st, ast.Return):
            if self.kind == 'helper':
                self._emit(ind, 'return %s;' % (self.expr(st.value) if st.value is not None
                                                else '0.0'))
                return
            if st.value is not None:
                self.err(st, 'return with a value')
            self._emit(ind, 'return;')
            return
        if isinstance(st, ast.Assign):
            if len(st.targets) != 1:
                self.err(st, 'chained assignment')
            tgt = st.targets[0]
            if isinstance(tgt, ast.Tuple) and self._declare_call(st.value) is not None:
                for t_ in tgt.elts:       # i, j = declare('int', 2)
                    self._declare(t_, self._declare_call(st.value), st)
                return
            if isinstance(tgt, ast.Tuple):
                if not isinstance(st.value, ast.Tuple) or len(tgt.elts) != len(st.value.elts):
                    self.err(st, 'tuple assignment')
                decl = [self._declare_call(v) for v in st.value.elts]
                if all(d is not None for d in decl):
                    for t_, d in zip(tgt.elts, decl):
                        self._declare(t_, d, st)
                    return
                vals = [self.expr(v) for v in st.value.elts]
                tmp = ['_t%d_%d' % (st.lineno, i) for i in range(len(vals))]
                for tn, v in zip(tmp, vals):
                    self._emit(ind, 'const double %s = %s;' % (tn, v))
                for t_, tn in zip(tgt.elts, tmp):
                    self._emit(ind, '%s = %s;' % (self._target(t_, st), tn))
                return
            d = self._declare_call(st.value)
            if d is not None:
                self._declare(tgt, d, st)
                return
            if isinstance(tgt, ast.Name) and self.locals.get(tgt.id, ('',))[0] == 'int':
                self.affine_ints.pop(tgt.id, None)
                aff = self._affine(st.value)
                if aff is not None and not (set(aff[0]) - {'d_idx', 's_idx'}) and \
                        (self._assign_counts().get(tgt.id, 0) == 1 or self.unrolled):
                    # i16 = 16*d_idx / n = self.dim: a value known at translation
                    # time (single assignment, or inside an unrolled loop)
                    self.affine_ints[tgt.id] = aff
                    if aff[0]:
                        return                      # only ever used inside indices
                    self._emit(ind, '%s = %d;' % (_cn(tgt.id), aff[1]))
                    return
                if isinstance(st.value, ast.Subscript) and isinstance(st.value.value, ast.Name) \
                        and st.value.value.id == 'NBRS':
                    rhs = self.subscript(st.value, store=False)
                else:
                    try:
                        rhs = self.index(st.value)
                    except CodegenError:
                        # idx = d_orig_idx[d_idx]: an index held in a (double) property
                        rhs = '(int)(%s)' % self.expr(st.value)
                self._emit(ind, '%s = %s;' % (_cn(tgt.id), rhs))
                return
            rhs = self.expr(st.value)
            self._emit(ind, '%s = %s;' % (self._target(tgt, st), rhs))
            return
        if isinstance(st, ast.AugAssign):
            ops = {ast.Add: '+=', ast.Sub: '-=', ast.Mult: '*=', ast.Div: '/='}
            if type(st.op) not in ops:
                self.err(st, 'augmented operator')
            rhs = self.expr(st.value)
            self._emit(ind, '%s %s %s;' % (self._target(st.target, st, aug=True),
                                           ops[type(st.op)], rhs))
            return
        if isinstance(st, ast.If):
            self._emit(ind, 'if (%s) {' % self.expr(st.test))
            self._emit_block(st.body, ind + 1)
            if st.orelse:
                self._emit(ind, '} else {')
                self._emit_block(st.orelse, ind + 1)
            self._emit(ind, '}')
            return
        if isinstance(st, ast.For):
            it = st.iter
            if not (isinstance(st.target, ast.Name) and isinstance(it, ast.Call)
                    and isinstance(it.func, ast.Name) and it.func.id == 'range'
                    and 1 <= len(it.args) <= 2) or st.orelse:
                self.err(st, 'only "for i in range(a[, b])" loops')
            var = st.target.id
            clo = 0 if len(it.args) == 1 else self._const_int(it.args[0])
            chi = self._const_int(it.args[-1])
            if clo is not None and chi is not None and chi - clo <= 64 and \
                    self._mentions_in_property_index(st.body, var):
                # components of strided properties live in separate registers:
                # a loop over them is unrolled at translation time
                saved = self.unrolled.get(var)
                outer_loops, self._runtime_loops = getattr(self, '_runtime_loops', 0), 0
                for val in range(clo, chi):
                    self.unrolled[var] = val
                    self._emit(ind, '{   // %s = %d' % (var, val))
                    self._emit_block(st.body, ind + 1)
                    self._emit(ind, '}')
                self._runtime_loops = outer_loops
                if saved is None:
                    self.unrolled.pop(var, None)
                else:
                    self.unrolled[var] = saved
                for nm in self._depends_on(var):
                    self.affine_ints.pop(nm, None)
                return
            lo, hi = ('0', self.index(it.args[0])) if len(it.args) == 1 else \
                (self.index(it.args[0]), self.index(it.args[1]))
            fresh = var not in self.loop_vars
            self.loop_vars.add(var)
            cv = _cn(var)
            self._emit(ind, 'for (int %s = %s; %s < %s; %s++) {' % (cv, lo, cv, hi, cv))
            self._runtime_loops = getattr(self, '_runtime_loops', 0) + 1
            self._emit_block(st.body, ind + 1)
            self._runtime_loops -= 1
            self._emit(ind, '}')
            if fresh:
                self.loop_vars.discard(var)
            return
        self.err(st, 'statement %s' % type(st).__name__)

    def _declare(self, tgt, d, st):
        if not isinstance(tgt, ast.Name):
            self.err(st, 'declare() target')
        self.locals[tgt.id] = d

    def _target(self, tgt, st, aug=False):
        if isinstance(tgt, ast.Name):
            if tgt.id in SCALAR_SYMBOLS and tgt.id not in ('t', 'dt') and self.pair:
                # e.g. CRKSPHSymmetric (crksph.py) rescales WI / WJ in place
                self.fam.use_symbol(tgt.id)
                self.fam.sym_written.add(tgt.id)
                return tgt.id
            if tgt.id in self.loop_vars or tgt.id in SCALAR_SYMBOLS or tgt.id in VEC_SYMBOLS:
                self.err(st, 'assignment to %s' % tgt.id)
            if tgt.id not in self.locals:
                if aug:
                    self.err(st, '%s used before assignment' % tgt.id)
                self.locals[tgt.id] = ('double', None)
            return _cn(tgt.id)
        if isinstance(tgt, ast.Subscript):
            return self.subscript(tgt, store=True)
        if isinstance(tgt, ast.Attribute) and isinstance(tgt.value, ast.Name) and tgt.value.id == 'self' \
                and self.eq is not None and not aug:
            # self.equation_has_converged = -1 (gas_dynamics/basic.py:121-160, swe/basic.py):
            # a flag the host reads back (converged()); every lane stores the same constant
            if not isinstance(getattr(self.eq, tgt.attr, None), (int, float)):
                self.err(st, 'self.%s is not a scalar attribute of the equation' % tgt.attr)
            return self.fam.state_slot(self.k, tgt.attr, self.kind)
        self.err(st, 'assignment target %s' % type(tgt).__name__)

    def code(self, ind):
        """the body wrapped in an immediately invoked lambda: locals are
        hoisted to its top, a bare ``return`` leaves just this equation"""
        pad = '    ' * ind
        out = [pad + '[&]() {  // %s' % self.where]
        for name, (kind, n) in sorted(self.locals.items()):
            if kind == 'array':
                out.append(pad + '    double %s[%d] = {};' % (_cn(name), n))
            elif kind == 'int':
                out.append(pad + '    int %s = 0; (void)%s;' % (_cn(name), _cn(name)))
            else:
                out.append(pad + '    double %s = 0.0;' % _cn(name))
        for ln in self.lines:
            out.append(pad + ln)
        out.append(pad + '}();')
        return '\n'.join(out)


class _HelperBody(_Body):
    """A module-level Python function called from an equation/stepper body,
    emitted as a ``__device__`` function of doubles."""

    def __init__(self, fam, fn, name):
        self.fam = fam
        self.eq = None
        self.k = -1
        self.kind = 'helper'
        self.pair = self.all_nbrs = self.raw_src = False
        try:
            src = textwrap.dedent(inspect.getsource(fn))
        except (OSError, TypeError) as e:
            raise CodegenError('helper %s: source not available (%s)' % (name, e))
        self.fdef = fdef = ast.parse(src).body[0]
        self.fn = fn
        self.where = 'helper %s' % name
        self.cname = 'gen_helper_%s' % name
        a = fdef.args
        if a.vararg or a.kwarg or a.kwonlyargs:
            raise CodegenError('helper %s: only plain scalar arguments' % name)
        self.argnames = [x.arg for x in a.args]
        nd = len(a.args) - len(a.defaults)
        self.defaults = [None] * nd
        self.argtypes = ['double'] * nd
        for d in a.defaults:
            if isinstance(d, ast.List):                     # m=[1., 0.]  -> double *
                self.defaults.append(None)
                self.argtypes.append('array')
            elif isinstance(d, ast.Constant) and isinstance(d.value, bool):
                raise CodegenError('helper %s: boolean defaults' % name)
            elif isinstance(d, ast.Constant) and isinstance(d.value, int):
                self.defaults.append(str(d.value))
                self.argtypes.append('int')
            elif isinstance(d, ast.Constant) and isinstance(d.value, float):
                self.defaults.append(repr(d.value))
                self.argtypes.append('double')
            else:
                raise CodegenError('helper %s: default values must be numbers or lists' % name)
        kinds = {'double': 'arg', 'int': 'int', 'array': 'arrayarg'}
        self.locals = dict((n_, (kinds[t_], None)) for n_, t_ in zip(self.argnames, self.argtypes))
        self.loop_vars = set()
        self.unrolled = {}
        self.affine_ints = {}
        self.lines = []
        self._emit_block(fdef.body, 1)
        self.writes = set()

    def _self_param(self, node, attr):
        self.err(node, 'self is not available in a helper function')

    def definition(self):
        ctype = {'double': 'double ', 'int': 'int ', 'array': 'double *'}
        out = ['__device__ __forceinline__ double %s(%s)' % (
            self.cname, ', '.join(ctype[t] + _cn(n) for n, t in zip(self.argnames, self.argtypes))), '{']
        for name, (kind, n) in sorted(self.locals.items()):
            if name in self.argnames:
                continue
            if kind == 'array':
                out.append('    double %s[%d] = {};' % (_cn(name), n))
            elif kind == 'int':
                out.append('    int %s = 0; (void)%s;' % (_cn(name), _cn(name)))
            else:
                out.append('    double %s = 0.0;' % _cn(name))
        out += self.lines
        out += ['    return 0.0;', '}']
        return '\n'.join(out)


# Ahead-of-time builds (tests/prebuild_generated.py, __graft_entry__.build()): while
# DEFERRED is a list, families that are not in the cache are only collected; then
# build_deferred() compiles them side by side.
DEFERRED = None


class _DeferredModule(object):
    sphgen_launch = C.CFUNCTYPE(C.c_int, C.c_void_p)(lambda args: -1)


def build_deferred(jobs=None):
    """compile the collected families with `jobs` hipcc processes at a time"""
    global DEFERRED
    from concurrent.futures import ThreadPoolExecutor
    pending, DEFERRED = DEFERRED or [], None
    unique = list(dict((f.hash, f) for f in pending).values())
    if unique:
        with ThreadPoolExecutor(max_workers=jobs or min(16, os.cpu_count() or 4)) as pool:
            list(pool.map(lambda f: f.build(), unique))
    return len(unique)


_F32_LITERAL = re.compile(r'(?<![\w.])((?:\d+\.\d*|\.\d+)(?:[eE][+-]?\d+)?|\d+[eE][+-]?\d+)(?![\w.])')
_F32_VIEW = (
    '#define SPHGEN_F32 1\n'
    '// fp64 memory read as float operands (scalar parameters, whole source arrays of loop_all / initialize_pair)\n'
    'struct GenF32View {\n'
    '    const double *p;\n'
    '    __device__ __forceinline__ float operator[](long i) const { return (float)p[i]; }\n'
    '};\n')


def source_f32(src):
    """The family's source with its ARITHMETIC in float: the fp32 mode of the
    reference's generated GPU code (acceleration_eval_gpu_helper.py:281-283,
    437-441 casts every array and scalar to float32 when use_double is off).
    Particle arrays stay fp64 in memory (values narrowed on load, widened on
    store), the records of the pair loop are fp32 (option record_f32 layout),
    every local, pair symbol, kernel evaluation, literal and accumulator is
    float.  Text-level: the emitted skeleton is regular, the interface parts
    (Params, the fp64 record reader, the launch wrapper) are kept as they are."""
    out, keep = [], None
    for ln in src.split('\n'):
        st = ln.strip()
        if keep is None:
            if st == 'struct Params {':
                keep = '    };'
            elif st.startswith('template <> __device__ __forceinline__ void load_record<FamGen, true>'):
                keep = '}'
            elif st.startswith('extern "C" int sphgen_kernel_kind'):
                keep = '\0'                     # to the end: the launch wrapper
        if keep is not None:
            out.append(ln)
            if ln == keep:
                keep = None
            continue
        if st.startswith('#') or st.startswith('//'):
            out.append(ln)
            if st == '#include <cstring>':
                out.append(_F32_VIEW)
            continue
        m = re.match(r'^(\s*)const double \*(PAR|S_\w+) = (a\.p\.[\w\[\]\.]+); \(void\)\2;$', ln)
        if m:
            out.append('%sconst GenF32View %s{%s}; (void)%s;' % (m.group(1), m.group(2), m.group(3), m.group(2)))
            continue
        ln = ln.replace('((double)d_idx)', '@IDX_D@').replace('((double)o)', '@IDX_O@')
        ln = re.sub(r'\bdouble4\b', 'float4', ln)
        ln = re.sub(r'\bdouble\b', 'float', ln)
        ln = re.sub(r'\bPairGeom\b', 'PairGeomT<float>', ln)
        ln = re.sub(r'\b(a\.k\.sigma|a\.k\.deltap|a\.hu)\b', r'((float)\1)', ln)
        ln = re.sub(r'\b(M_PI|M_1_PI|M_2_SQRTPI|M_PI_2)\b', r'((float)\1)', ln)
        ln = _F32_LITERAL.sub(r'\1f', ln)
        ln = ln.replace('@IDX_D@', '((double)d_idx)').replace('@IDX_O@', '((double)o)')
        out.append(ln)
    return '\n'.join(out)


class GeneratedFamily(object):
    """All equations of one group acting on one destination, generated."""

    def __init__(self, dest, equations, arrays, kernel_kind, name=None,
                 skip_initialize=()):
        if len(equations) > 32:
            raise CodegenError('more than 32 equations on one destination')
        self.dest = dest
        self.equations = list(equations)
        self.arrays = arrays
        self.kernel_kind = int(kernel_kind)
        self.name = name or 'gen'
        self.dprops = []        # dest props, order of first use
        self.dwritten = set()
        self.sprops = []        # source props packed into the records
        self.symbols = set()
        self.strides = {}       # multi-component properties: name -> stride
        self.params = []        # [(key, value)]
        self.sources = []       # first-appearance order (acceleration_eval.py:136-151)
        self.src_flags = {}
        for k, eq in enumerate(self.equations):
            for m in UNSUPPORTED_METHODS:
                if callable(getattr(type(eq), m, None)):
                    raise CodegenError('%s.%s is not supported by the generated path'
                                       % (type(eq).__name__, m))
            for s in (eq.sources or []):
                if s not in self.sources:
                    self.sources.append(s)
                    self.src_flags[s] = 0
                self.src_flags[s] |= 1 << k
        self.abs_src_pos = False
        self.helpers = OrderedDict()    # name -> _HelperBody, in dependency order
        self.sym_written = set()        # pair symbols some equation assigns to
        self.raw_dest = set()           # destination properties also read at a run-time index
        self.state = []                 # [(equation index, attribute)] assigned by device code
        self.state_in_init = False
        self.bodies = {m: [] for m in METHODS}
        self.nosrc_loops = []
        for k, eq in enumerate(self.equations):
            for m in METHODS:
                fdef = _method_ast(eq, m)
                if fdef is None or (m == 'initialize' and eq in skip_initialize):
                    continue
                b = _Body(self, eq, k, m, fdef)
                if m == 'loop' and self.is_no_source(eq):
                    self.nosrc_loops.append(b)
                else:
                    self.bodies[m].append(b)
        # loop_all families: one thread per destination walks its neighbour list
        self.loop_all = bool(self.bodies['loop_all'])
        # loop_all AND loop on one destination (mako :62-110 runs loop_all, then
        # the pair loop, per source): initialize gets its own launch, then the
        # loop_all launch (no post_loop), then the pair launch (with post_loop)
        self.also_pair = self.loop_all and bool(self.bodies['loop'])
        # initialize_pair (mako :62-75): per source, before its loops, one sweep
        # over the destinations with the source ARRAYS in view.  1: loops
        # follow; 2: nothing else walks neighbours (post_loop ends the sweep)
        self.init_pair = 0
        if self.bodies['initialize_pair']:
            self.init_pair = 1 if (self.loop_all or self.bodies['loop']) else 2
        # initialize()/no-source loops of ALL particles must be finished before a
        # loop reads what they wrote as a SOURCE property (mako :36-58): then
        # they run as a launch of their own, before the records are packed
        early = set()
        for b in self.bodies['initialize'] + self.nosrc_loops:
            early |= b.writes
        self.split_init = self.also_pair or bool(self.init_pair) or self.state_in_init or (
            bool(self.sources) and dest in self.sources and bool(early & set(self.sprops)))
        if 'VIJ' in self.symbols:
            for p in 'uvw':
                self.dest_prop(p, False)
                self.src_prop(p)
        if self.symbols & {'RHOIJ', 'RHOIJ1'}:
            self.dest_prop('rho', False)
            self.src_prop('rho')
        if len(self.sprops) > 20:
            raise CodegenError('more than 20 source properties in one family')
        if len(self.dprops) > 48:
            raise CodegenError('more than 48 destination properties in one family')
        if len(self.params) > 64:
            raise CodegenError('more than 64 scalar parameters in one family')
        self.source = self._emit_source()
        # content address: the generated struct AND the skeleton it is compiled
        # against (a changed sph_pair.h must not pick up a stale binary)
        self.hash = hashlib.sha1((self.source + _skeleton_digest()).encode()).hexdigest()[:16]
        self.lib = None

    # -- bookkeeping used by the bodies -------------------------------------
    @staticmethod
    def is_no_source(eq):
        return not eq.sources

    def is_dest_constant(self, name):
        pa = self.arrays.get(self.dest)
        return pa is not None and name in getattr(pa, 'constants', {})

    def dest_prop(self, prop, store):
        if prop not in self.dprops:
            self.dprops.append(prop)
        if store:
            self.dwritten.add(prop)
            if getattr(self, '_writes', None) is not None:
                self._writes.add(prop)
        return 'D.d_%s' % prop

    def src_prop(self, prop):
        if prop in ('x', 'y', 'z', 'h'):
            if prop != 'h':
                # a body reads a neighbour's ABSOLUTE position (not XIJ): fp32 records hold positions
                # relative to the grid origin, so this family keeps fp64 records
                self.abs_src_pos = True
            return {'x': 'pj.x', 'y': 'pj.y', 'z': 'pj.z', 'h': 's_h'}[prop]
        if prop not in self.sprops:
            self.sprops.append(prop)
        return 's_%s' % prop

    def use_symbol(self, s):
        self.symbols.add(s)

    def state_slot(self, k, attr, kind):
        key = (k, attr)
        if key not in self.state:
            if len(self.state) >= 16:
                raise CodegenError('more than 16 equation attributes assigned by device code')
            self.state.append(key)
        if kind == 'initialize':
            self.state_in_init = True
        return 'a.p.state[%d]' % self.state.index(key)

    def state_values(self):
        return [float(getattr(self.equations[k], attr)) for k, attr in self.state]

    def store_state(self, values):
        """what the device code left in the attributes goes back to the equation
        objects (type of the current value kept)"""
        for (k, attr), v in zip(self.state, values):
            cur = getattr(self.equations[k], attr)
            setattr(self.equations[k], attr, type(cur)(v) if isinstance(cur, (int, bool)) else float(v))

    def raw_dest_prop(self, prop):
        """placeholder for the memory pointer of a destination property (resolved
        to din[]/dout[] once every body is translated)"""
        self.dest_prop(prop, False)
        self.raw_dest.add(prop)
        return '@RAWD_%s@' % prop

    def helper(self, body, node, fname):
        """the Python function `fname` as seen from the calling body: listed by
        the equation's/stepper's ``_get_helpers_()`` or a plain function in the
        globals of the calling method"""
        if fname in self.helpers:
            return self.helpers[fname]
        fn = None
        owner = body.eq if body.eq is not None else None
        cands = []
        if owner is not None:
            get = getattr(owner, '_get_helpers_', None)
            if callable(get):
                cands += list(get() or [])
            for m in METHODS:
                meth = getattr(type(owner), m, None)
                if meth is not None and hasattr(meth, '__globals__'):
                    g = meth.__globals__.get(fname)
                    if g is not None:
                        cands.append(g)
        else:
            g = body.fn.__globals__.get(fname)
            if g is not None:
                cands.append(g)
        for c in cands:
            if inspect.isfunction(c) and c.__name__ == fname:
                fn = c
                break
        if fn is None:
            return None
        saved = getattr(self, '_writes', None)
        h = _HelperBody(self, fn, fname)       # may pull in further helpers first
        self._writes = saved
        self.helpers[fname] = h
        return h

    def raw_src_prop(self, prop):
        """loop_all: source properties are read from the arrays themselves
        (original order), x y z h included"""
        if prop not in self.sprops:
            self.sprops.append(prop)
        return prop

    def note_stride(self, prop, stride, body, node):
        """component k of a stride-S property travels as the scalar device
        property ``<prop>__k`` (split / re-interleaved by HipDeviceHelper)"""
        if self.strides.setdefault(prop, stride) != stride:
            body.err(node, 'property %s used with two strides' % prop)

    def param(self, key, value):
        for i, (k, _) in enumerate(self.params):
            if k == key:
                return 'PAR[%d]' % i
        self.params.append((key, value))
        return 'PAR[%d]' % (len(self.params) - 1)

    def param_values(self):
        """current values: self.* are frozen at build time (equation.py:885-892),
        array constants are read at every compute (they may be updated)."""
        from .particle_array import get_npy
        out = []
        for key, val in self.params:
            if key[0] == 'const':
                out.append(float(get_npy(self.arrays[self.dest], key[1])[key[2]]))
            else:
                out.append(val)
        return out

    # -- emission -----------------------------------------------------------
    def _emit_source(self):
        din = [p for p in self.dprops if p not in self.dwritten]
        dout = [p for p in self.dprops if p in self.dwritten]
        self.din, self.dout = din, dout
        na = max(2, (len(self.sprops) + 1) & ~1)
        S = self.symbols
        L = []
        A = L.append
        A('// generated by pysph_amd/codegen.py -- do not edit')
        A('// destination: %s; equations: %s' % (
            self.dest, ', '.join(type(e).__name__ for e in self.equations)))
        A('#ifndef SPHGEN_MINB')
        A('#define SPHGEN_MINB 3')
        A('#endif')
        A('#include "sph_pair.h"')
        A('#include <cstring>')
        A('')
        A('// kernel functions by (r, h) for loop_all bodies: SPH_KERNEL.kernel / gradient / dwdq')
        A('template <int KK, class A> __device__ __forceinline__ double gen_kernel_w(double rij, double h, const A &a)')
        A('{')
        A('    const double h1 = 1.0 / h;')
        A('    return kernel_norm(a.k.sigma, h1, a.k.dim) * SphKernel<KK>::template w<false>(rij * h1);')
        A('}')
        A('template <int KK, class A> __device__ __forceinline__ double gen_kernel_dwdq(double rij, double h, const A &a)')
        A('{')
        A('    const double h1 = 1.0 / h;')
        A('    return kernel_norm(a.k.sigma, h1, a.k.dim) * SphKernel<KK>::template dw<false>(rij * h1);')
        A('}')
        A('template <int KK, class A>')
        A('__device__ __forceinline__ void gen_kernel_gradient(const double *xij, double rij, double h, double *grad, const A &a)')
        A('{')
        A('    const double h1 = 1.0 / h;')
        A('    const double wdash = kernel_norm(a.k.sigma, h1, a.k.dim) * SphKernel<KK>::template dw<false>(rij * h1);')
        A('    const double tmp = rij > 1e-12 ? wdash * h1 / rij : 0.0;     // kernels.py:126-137')
        A('    grad[0] = tmp * xij[0]; grad[1] = tmp * xij[1]; grad[2] = tmp * xij[2];')
        A('}')
        A('')
        for h in self.helpers.values():
            L.extend(h.definition().split('\n'))
            A('')
        A('struct FamGen {')
        A('    static constexpr int MINB = SPHGEN_MINB;   // workgroups per CU, chosen at build time')
        A('    typedef double Real;                       // generated bodies compute in fp64')
        A('    static constexpr bool PRED = false;        // the neighbour criterion stays a branch around pair()')
        A('    static constexpr uint32_t CF0 = 0;         // equation flags are run-time values')
        A('    static constexpr int NA = %d;' % na)
        A('    static constexpr int NR = 4 + NA;')
        A('    struct Params {')
        A('        const double *din[%d];' % max(len(din), 1))
        A('        double *dout[%d];' % max(len(dout), 1))
        A('        double par[%d];' % max(len(self.params), 1))
        A('        const uint32_t *csr_start[SPH_MAX_ARRAYS], *csr_nbrs[SPH_MAX_ARRAYS];   // loop_all')
        A('        double *state;   // equation attributes assigned by the bodies')
        A('        const double *sraw[SPH_MAX_ARRAYS][%d];' % max(len(self.sprops), 1))
        A('    };')
        A('    struct Dest {')
        for p in self.dprops:
            A('        double d_%s;' % p)
        if not self.dprops:
            A('        double unused_;')
        A('    };')
        # ---- load = memory -> registers, initialize, no-source loops
        A('    template <class A> static __device__ __forceinline__ void load(Dest &D, const double *, const A &a, uint32_t o)')
        A('    {')
        A('        const double *PAR = a.p.par; (void)PAR;')
        A('        const double t = a.t, dt = a.dt; (void)t; (void)dt;')
        for i, p in enumerate(din):
            A('        D.d_%s = a.p.din[%d][o];' % (p, i))
        for i, p in enumerate(dout):
            A('        D.d_%s = a.p.dout[%d][o];' % (p, i))
        A('        if (!a.skip_init) {')
        for b in self.bodies['initialize']:
            A(b.code(3))
        for b in self.nosrc_loops:
            A(b.code(3))
        A('        }')
        A('    }')
        # ---- pair
        A('    template <int KK, bool UH, class A>')
        A('    static __device__ __forceinline__ void pair(Dest &D, const double4 &pi, const double4 &pj, double r2,')
        A('                                                const double (&s)[NA], uint32_t fl, const A &a)')
        A('    {')
        A('        const double *PAR = a.p.par; (void)PAR;')
        A('        const double t = a.t, dt = a.dt; (void)t; (void)dt;')
        A('        PairGeom g;')
        A('        pair_geom<KK, UH>(g, pi, pj, r2, a);')
        A('        const double XIJ[3] = {g.xij[0], g.xij[1], g.xij[2]}; (void)XIJ;')
        A('        const double R2IJ = r2, RIJ = g.rij, HIJ = g.hij, EPS = g.eps; (void)R2IJ; (void)RIJ; (void)HIJ; (void)EPS;')
        A('        const double s_h = UH ? a.hu : pj.w; (void)s_h;')
        for i, p in enumerate(self.sprops):
            if p not in ('x', 'y', 'z', 'h'):
                A('        const double s_%s = s[%d];' % (p, i))
        if 'VIJ' in S:
            A('        const double VIJ[3] = {D.d_u - s_u, D.d_v - s_v, D.d_w - s_w};')
        if S & {'RHOIJ', 'RHOIJ1'}:
            A('        const double RHOIJ = 0.5 * (D.d_rho + s_rho); (void)RHOIJ;')
            A('        const double RHOIJ1 = 1.0 / RHOIJ; (void)RHOIJ1;')
        if 'WIJ' in S:
            A('        const double WIJ = pair_w<KK, UH>(g);')
        if 'WDP' in S:   # KERNEL(XIJ, DELTAP*HIJ, HIJ), equation.py:243-246
            A('        const double WDP = SphKernel<KK>::template w<false>((a.k.deltap * g.hij) * g.h1) * g.fac;')
        if 'DWIJ' in S:
            A('        const double tg_ = pair_gradfac<KK, UH>(g);')
            A('        const double DWIJ[3] = {tg_ * XIJ[0], tg_ * XIJ[1], tg_ * XIJ[2]};')
        if 'GHIJ' in S:     # dW/dh at HIJ (equation.py:285-295 GRADH)
            A('        const double GHIJ = pair_gradh<KK, UH>(g, a.k.dim);')
        if S & {'WI', 'DWI', 'WJ', 'DWJ', 'GHI', 'GHJ'}:
            # kernels evaluated with h_d / h_s (equation.py:262-297)
            A('        PairGeom gi = g, gj = g;')
            A('        if (!UH) {')
            A('            gi.hij = pi.w; gi.h1 = 1.0 / pi.w; gi.fac = kernel_norm(a.k.sigma, gi.h1, a.k.dim); gi.q = g.rij * gi.h1;')
            A('            gj.hij = pj.w; gj.h1 = 1.0 / pj.w; gj.fac = kernel_norm(a.k.sigma, gj.h1, a.k.dim); gj.q = g.rij * gj.h1;')
            A('        }')
            if 'WI' in S:
                A('        const double WI = pair_w<KK, false>(gi);')
            if 'WJ' in S:
                A('        const double WJ = pair_w<KK, false>(gj);')
            if 'GHI' in S:
                A('        const double GHI = pair_gradh<KK, false>(gi, a.k.dim);')
            if 'GHJ' in S:
                A('        const double GHJ = pair_gradh<KK, false>(gj, a.k.dim);')
            if 'DWI' in S:
                A('        const double tgi_ = pair_gradfac<KK, false>(gi);')
                A('        const double DWI[3] = {tgi_ * XIJ[0], tgi_ * XIJ[1], tgi_ * XIJ[2]};')
            if 'DWJ' in S:
                A('        const double tgj_ = pair_gradfac<KK, false>(gj);')
                A('        const double DWJ[3] = {tgj_ * XIJ[0], tgj_ * XIJ[1], tgj_ * XIJ[2]};')
        for b in self.bodies['loop']:
            A('        if (fl & %du) {' % (1 << b.k))
            A(b.code(3))
            A('        }')
        A('    }')
        # ---- finish = post_loop + stores
        A('    template <class A> static __device__ __forceinline__ void finish(Dest &D, const A &a, uint32_t o)')
        A('    {')
        A('        const double *PAR = a.p.par; (void)PAR;')
        A('        const double t = a.t, dt = a.dt; (void)t; (void)dt;')
        if self.bodies['post_loop']:
            A('        if (!a.skip_post) {   // an intermediate launch of a sequenced family: stores only')
            for b in self.bodies['post_loop']:
                A(b.code(3))
            A('        }')
        A('        store(D, a, o);')
        A('    }')
        A('    template <class A> static __device__ __forceinline__ void store(Dest &D, const A &a, uint32_t o)')
        A('    {')
        for i, p in enumerate(dout):
            A('        a.p.dout[%d][o] = D.d_%s;' % (i, p))
        A('    }')
        # ---- loop_all bodies of one source
        A('    template <int KK, class A>')
        A('    static __device__ __forceinline__ void all_nbrs(Dest &D, const A &a, int j, uint32_t d_idx)')
        A('    {')
        A('        const double *PAR = a.p.par; (void)PAR;')
        A('        const double t = a.t, dt = a.dt; (void)t; (void)dt;')
        A('        const uint32_t fl = a.src[j].flags; (void)fl;')
        A('        const uint32_t *NBRS = a.p.csr_nbrs[j] + a.p.csr_start[j][d_idx]; (void)NBRS;')
        A('        const int N_NBRS = (int)(a.p.csr_start[j][d_idx + 1] - a.p.csr_start[j][d_idx]); (void)N_NBRS;')
        if self.loop_all:
            for i, p in enumerate(self.sprops):
                A('        const double *S_%s = a.p.sraw[j][%d]; (void)S_%s;' % (p, i, p))
        for b in self.bodies['loop_all']:
            A('        if (fl & %du) {' % (1 << b.k))
            A(b.code(3))
            A('        }')
        A('    }')
        # ---- initialize_pair bodies of one source
        A('    template <class A>')
        A('    static __device__ __forceinline__ void init_pair(Dest &D, const A &a, int j, uint32_t d_idx)')
        A('    {')
        A('        const double *PAR = a.p.par; (void)PAR;')
        A('        const double t = a.t, dt = a.dt; (void)t; (void)dt;')
        A('        const uint32_t fl = a.src[j].flags; (void)fl;')
        if self.init_pair:
            for i, p in enumerate(self.sprops):
                A('        const double *S_%s = a.p.sraw[j][%d]; (void)S_%s;' % (p, i, p))
        for b in self.bodies['initialize_pair']:
            A('        if (fl & %du) {' % (1 << b.k))
            A(b.code(3))
            A('        }')
        A('    }')
        A('};')
        A('')
        ns = len(self.sprops)
        nrc = (3 + ns + 1) & ~1
        A('// records under uniform h: [x y z | source props...] (%d doubles)' % nrc)
        A('template <> __device__ __forceinline__ void load_record<FamGen, true>(const double *__restrict__ rj, uint32_t,')
        A('                                                                      double4 &pj, double (&s)[FamGen::NA])')
        A('{')
        A('    const double2 *r2 = reinterpret_cast<const double2 *>(rj);')
        A('    double w[%d];' % nrc)
        A('#pragma unroll')
        A('    for (int q = 0; q < %d; q++) { const double2 t = r2[q]; w[2 * q] = t.x; w[2 * q + 1] = t.y; }' % (nrc // 2))
        A('    pj.x = w[0]; pj.y = w[1]; pj.z = w[2]; pj.w = 0.0;')
        A('#pragma unroll')
        A('    for (int k = 0; k < FamGen::NA; k++) s[k] = k < %d ? w[3 + (k < %d ? k : 0)] : 0.0;' % (ns, max(ns, 1)))
        A('}')
        A('')
        A('__global__ __launch_bounds__(256) void k_gen_nosrc(PairArgs<FamGen> a)')
        A('{')
        A('    const size_t i = (size_t)a.d_start + (size_t)blockIdx.x * 256 + threadIdx.x;')
        A('    if (i >= a.d_stop) return;')
        A('    FamGen::Dest D;')
        A('    FamGen::load(D, nullptr, a, (uint32_t)i);')
        A('    FamGen::finish(D, a, (uint32_t)i);')
        A('}')
        A('')
        A('// initialize + no-source loops only (mode 1): results go to memory')
        A('__global__ __launch_bounds__(256) void k_gen_init(PairArgs<FamGen> a)')
        A('{')
        A('    const size_t i = (size_t)a.d_start + (size_t)blockIdx.x * 256 + threadIdx.x;')
        A('    if (i >= a.d_stop) return;')
        A('    FamGen::Dest D;')
        A('    FamGen::load(D, nullptr, a, (uint32_t)i);')
        A('    FamGen::store(D, a, (uint32_t)i);')
        A('}')
        A('')
        A('// loop_all equations (mode 2): one thread per destination, neighbour lists in CSR form')
        A('__global__ __launch_bounds__(256) void k_gen_loop_all(PairArgs<FamGen> a)')
        A('{')
        A('    const size_t i = (size_t)a.d_start + (size_t)blockIdx.x * 256 + threadIdx.x;')
        A('    if (i >= a.d_stop) return;')
        A('    FamGen::Dest D;')
        A('    FamGen::load(D, nullptr, a, (uint32_t)i);')
        A('    for (int j = 0; j < a.nsrc; j++) FamGen::all_nbrs<%d>(D, a, j, (uint32_t)i);' % self.kernel_kind)
        A('    FamGen::finish(D, a, (uint32_t)i);')
        A('}')
        A('')
        A('// initialize_pair equations (mode 3): the sources whose flags are set, in order')
        A('__global__ __launch_bounds__(256) void k_gen_init_pair(PairArgs<FamGen> a)')
        A('{')
        A('    const size_t i = (size_t)a.d_start + (size_t)blockIdx.x * 256 + threadIdx.x;')
        A('    if (i >= a.d_stop) return;')
        A('    FamGen::Dest D;')
        A('    FamGen::load(D, nullptr, a, (uint32_t)i);')
        A('    for (int j = 0; j < a.nsrc; j++) if (a.src[j].flags) FamGen::init_pair(D, a, j, (uint32_t)i);')
        A('    FamGen::finish(D, a, (uint32_t)i);')
        A('}')
        A('')
        A('extern "C" int sphgen_kernel_kind(void) { return %d; }' % self.kernel_kind)
        A('')
        A('extern "C" int sphgen_launch(const sph_gen_args *g)')
        A('{')
        A('    if (g->kernel_kind != %d) return -1000;' % self.kernel_kind)
        A('    if (g->n_din != %d || g->n_dout != %d || g->npar != %d) return -1001;' % (
            len(din), len(dout), len(self.params)))
        A('    PairArgs<FamGen> a;')
        A('    memset(&a, 0, sizeof a);')
        A('    a.gfx_lo = -0x7fffffff; a.gfx_hi = 0x7fffffff; // no ghost split for generated families')
        A('    a.norm_masks = g->norm_masks;')
        A('    a.row_mod3 = g->row_mod3;')
        A('    a.nsrc = g->nsrc;')
        A('    for (int j = 0; j < g->nsrc; j++) a.src[j] = {g->src_cell_start[j], g->src_off[j], g->src_flags[j], g->src_fine_start[j]};')
        A('    a.rec = g->rec; a.nrec = g->nrec; a.fpos = (const float4 *)g->fpos; a.dom_extent = g->dom_extent;')
        A('    a.d_off = g->d_off; a.nd = g->nd; a.d_keys = g->d_keys; a.d_fkeys = g->d_fkeys; a.d_perm = g->d_perm;')
        A('    a.d_tile_order = g->d_tile_order;')
        A('    a.d_start = g->d_start; a.d_stop = g->d_stop; a.dflags = g->dflags;')
        A('    for (int k = 0; k < 3; k++) { a.nc[k] = g->nc[k]; a.xmin[k] = g->xmin[k]; }')
        A('    a.cell_size = g->cell_size; a.radius_scale = g->radius_scale;')
        A('    a.k.sigma = g->sigma; a.k.deltap = g->deltap; a.k.dim = g->dim;')
        A('    a.t = g->t; a.dt = g->dt;')
        A('    a.hu = g->hu; a.h1u = g->h1u; a.facu = g->facu; a.epsu = g->epsu; a.hr2u = g->hr2u;')
        A('    for (int k = 0; k < g->n_din; k++) a.p.din[k] = g->din[k];')
        A('    for (int k = 0; k < g->n_dout; k++) a.p.dout[k] = g->dout[k];')
        A('    for (int k = 0; k < g->npar; k++) a.p.par[k] = g->par[k];')
        A('    a.skip_init = g->skip_init;')
        A('    a.p.state = g->state;')
        A('    a.skip_post = g->skip_post;')
        A('    for (int j = 0; j < SPH_MAX_ARRAYS; j++) {')
        A('        a.p.csr_start[j] = g->csr_start[j]; a.p.csr_nbrs[j] = g->csr_nbrs[j];')
        A('        for (int k = 0; k < %d; k++) a.p.sraw[j][k] = g->sraw[j][k];' % max(len(self.sprops), 1))
        A('    }')
        A('    hipStream_t st = (hipStream_t)g->stream;')
        A('    const size_t nrange = (size_t)g->d_stop - g->d_start;')
        A('    const dim3 lin((unsigned)((nrange + 255) / 256));')
        A('    if (g->mode == 1) {')
        A('        hipLaunchKernelGGL(k_gen_init, lin, dim3(256), 0, st, a);')
        A('    } else if (g->mode == 2) {')
        A('        hipLaunchKernelGGL(k_gen_loop_all, lin, dim3(256), 0, st, a);')
        A('    } else if (g->mode == 3) {')
        A('        hipLaunchKernelGGL(k_gen_init_pair, lin, dim3(256), 0, st, a);')
        A('    } else if (g->nsrc == 0) {')
        A('        const size_t n = (size_t)g->d_stop - g->d_start;')
        A('        hipLaunchKernelGGL(k_gen_nosrc, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a);')
        A('    } else {')
        A('        if (g->nrec != (g->rec_f32 ? ((4 + FamGen::NA + 3) & ~3) : g->uniform_h ? %d : FamGen::NR)) return -1002;' % nrc)
        A('        dim3 grid(4 * ((a.nd + 255) / 256) / WPB), block(64 * WPB);')
        A('#ifdef SPHGEN_F32   // Real = float reads fp32 records only')
        A('        if (!g->rec_f32) return -1003;')
        A('#endif')
        A('        if (g->rec_f32 && g->uniform_h) hipLaunchKernelGGL((k_pair_wave<FamGen, %d, true, true>), grid, block, 0, st, a);' % self.kernel_kind)
        A('        else if (g->rec_f32) hipLaunchKernelGGL((k_pair_wave<FamGen, %d, false, true>), grid, block, 0, st, a);' % self.kernel_kind)
        A('#ifndef SPHGEN_F32')
        A('        else if (g->uniform_h) hipLaunchKernelGGL((k_pair_wave<FamGen, %d, true>), grid, block, 0, st, a);' % self.kernel_kind)
        A('        else hipLaunchKernelGGL((k_pair_wave<FamGen, %d, false>), grid, block, 0, st, a);' % self.kernel_kind)
        A('#endif')
        A('    }')
        A('    return (int)hipGetLastError();')
        A('}')
        src = '\n'.join(L) + '\n'
        for prop in sorted(self.raw_dest):
            ptr = 'a.p.dout[%d]' % dout.index(prop) if prop in dout else 'a.p.din[%d]' % din.index(prop)
            src = src.replace('@RAWD_%s@' % prop, ptr)
        for sym in sorted(self.sym_written):     # symbols an equation assigns to lose their const
            src = src.replace('const double %s[3] =' % sym, 'double %s[3] =' % sym)
            src = src.replace('const double %s =' % sym, 'double %s =' % sym)
        return src

    # -- build / load ---------------------------------------------------------
    def so_path(self):
        return os.path.join(GEN_DIR, 'fam_%s.so' % self.hash)   # content-addressed

    def build(self, force=False):
        """hipcc the family into its own shared object (cached by hash)."""
        os.makedirs(GEN_DIR, exist_ok=True)
        so = self.so_path()
        if os.path.exists(so) and not force:
            return so
        if DEFERRED is not None and not force:
            DEFERRED.append(self)       # built later, in parallel (build_deferred)
            return so
        src = so[:-3] + '.hip'
        with open(src, 'w') as f:
            f.write(self.source)
        hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
        import shutil
        if not (os.path.isfile(hipcc) or shutil.which(hipcc)):
            # a box without the compiler only runs what was prebuilt into the
            # cache (tests/prebuild_generated.py, __graft_entry__.build())
            raise CodegenError(
                "equation family '%s' is not in the prebuilt cache (%s) and there is no "
                "hipcc at '%s' to compile it: build it where ROCm is installed (the cache "
                "directory travels with the package) or set HIPCC" % (self.name, so, hipcc))
        # occupancy: 4 workgroups/CU (128 VGPRs) when the pair kernel fits
        # without scratch, else 3 (168), else 2 (256) -- what the hand-written
        # families fix by hand (Fam::MINB), read here from the compiler's
        # resource-usage remarks
        log = ''
        for minb in (4, 3, 2):
            cmd = [hipcc, '-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC',
                   '-shared', '-DSPHGEN_MINB=%d' % minb,
                   '-Rpass-analysis=kernel-resource-usage', '-I', CSRC, '-I',
                   INCLUDE, src, '-o', so + '.tmp']
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                               universal_newlines=True)
            log = r.stdout
            if r.returncode != 0:
                errs = [ln for ln in log.splitlines() if 'remark:' not in ln]
                raise CodegenError('hipcc failed for %s:\n%s\n--- source: %s' %
                                   (self.name, '\n'.join(errs)[-4000:], src))
            scratch, in_pair = 0, False
            for ln in log.splitlines():
                if 'Function Name:' in ln:
                    in_pair = 'k_pair_wave' in ln
                elif in_pair and 'ScratchSize' in ln:
                    scratch = max(scratch, int(ln.split(':')[-1].split('[')[0]))
            self.minb, self.scratch = minb, scratch
            if scratch == 0 or minb == 2:
                break
        os.replace(so + '.tmp', so)
        return so

    def flavour_f32(self):
        """the same family with float arithmetic (`source_f32`): its own content
        address, built and cached like the fp64 one"""
        if getattr(self, '_f32', None) is None:
            import copy
            f = copy.copy(self)
            f.source = source_f32(self.source)
            f.hash = hashlib.sha1((f.source + _skeleton_digest()).encode()).hexdigest()[:16]
            f.name = self.name + '_f32'
            f.lib = None
            f._f32 = f
            self._f32 = f
        return self._f32

    def load(self):
        if self.lib is None:
            so = self.build()
            if DEFERRED is not None and not os.path.exists(so):
                return _DeferredModule()      # collection pass: nothing is launched
            self.lib = C.CDLL(so)
            self.lib.sphgen_launch.restype = C.c_int
            self.lib.sphgen_launch.argtypes = [C.c_void_p]
        return self.lib
