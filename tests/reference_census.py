"""Census of the reference's Equation classes through pysph_amd.codegen.

Run in a process of its own (it plants stand-ins for compyle / cyarray / the
Cython particle array in sys.modules so that as many pysph.sph modules as
possible import) by tests/test_codegen.py; prints one JSON object
{"ok": [...], "bad": {class: reason}, "import_failed": [...]}.
Needs /root/reference: build-container only."""
import sys, os, types, inspect, importlib, pkgutil, collections, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
for p in (REF, os.path.join(REPO,'oracle','_stubs'), REPO):
    sys.path.insert(0,p)
import numpy as np
# ---- extra stubs (census only) ----
def mod(name, **attrs):
    m = types.ModuleType(name); m.__dict__.update(attrs); sys.modules[name]=m; return m
class _Any(object):
    def __init__(self,*a,**k): pass
    def __getattr__(self,n): return _Any()
    def __call__(self,*a,**k): return _Any()
def ident_deco(*a, **k):
    if len(a)==1 and callable(a[0]) and not k: return a[0]
    return lambda f: f
import compyle.api as capi
for nm in ('get_config','Elementwise','Reduction','Scan','wrap','elementwise','profile','Array'):
    if not hasattr(capi,nm): setattr(capi,nm,_Any())
capi.annotate = ident_deco
import contextlib
@contextlib.contextmanager
def _ctx(*a, **k):
    yield
mod('compyle.profile', profile_ctx=_ctx, profile=ident_deco, profile_kernel=ident_deco, named_profile=ident_deco, get_profile_info=lambda:{})
mod('compyle.types', KnownType=capi.KnownType, annotate=ident_deco, declare=capi.declare)
mod('compyle.array', Array=_Any, wrap=_Any(), get_backend=lambda *a:'cython', to_device=_Any(), empty=_Any(), zeros=_Any())
mod('compyle.parallel', Elementwise=_Any, Reduction=_Any, Scan=_Any, elementwise=ident_deco)
mod('compyle.low_level', cast=lambda x,t:x, atomic_inc=_Any(), address=_Any())
mod('compyle.utils', ArgumentParser=_Any)
mod('compyle.ext_module', ExtModule=_Any, get_platform_dir=_Any(), get_md5=_Any())
mod('compyle.cython_generator', KnownType=capi.KnownType, CythonGenerator=_Any, get_parallel_range=_Any(), get_func_definition=_Any())
mod('compyle.opencl', get_context=_Any(), get_queue=_Any(), profile_kernel=_Any())
mod('compyle.template', Template=object)
mod('cyarray'); mod('cyarray.api', UIntArray=_Any, DoubleArray=_Any, IntArray=_Any, LongArray=_Any, BaseArray=_Any)
mod('cyarray.carray', UIntArray=_Any, DoubleArray=_Any, IntArray=_Any, LongArray=_Any, BaseArray=_Any)
import pysph, pysph.base
class FakePA(object): pass
mod('pysph.base.particle_array', ParticleArray=FakePA, get_ghost_tag=lambda:2, get_local_tag=lambda:0, get_remote_tag=lambda:1, is_local=_Any(), is_ghost=_Any(), is_remote=_Any())
mod('pysph.base.nnps', LinkedListNNPS=_Any, DomainManager=_Any, NNPS=_Any, get_number_of_threads=lambda:1)
mod('pysph.base.reduce_array', serial_reduce_array=_Any(), parallel_reduce_array=_Any(), dummy_reduce_array=_Any())
mod('pysph.cpy', ) 
mod('mako'); mod('mako.template', Template=_Any)
mod('pysph.base.linalg3', eigen_decomposition=_Any(), transform=_Any(), transform_diag=_Any(), transform_diag_inv=_Any(), py_det=_Any())
mod('pysph.base.tree', )
from pysph_amd.codegen import GeneratedFamily, CodegenError, METHODS
import pysph.sph
mods=[]; impfail=[]
for m in pkgutil.walk_packages(pysph.sph.__path__, 'pysph.sph.'):
    if '.tests' in m.name: continue
    try: mods.append(importlib.import_module(m.name))
    except Exception as e: impfail.append((m.name, type(e).__name__, str(e)[:90]))
from pysph.sph.equation import Equation
class AnyArray(object):
    name='fluid'
    class _P(dict):
        def __contains__(self,k): return True
        def __getitem__(self,k): return np.zeros(2)
    properties=_P(); constants={}; stride={}
seen=set(); ok=[]; bad=collections.OrderedDict(); cons=[]; builds=[]
for mod_ in mods:
    for name,cls in inspect.getmembers(mod_, inspect.isclass):
        if not issubclass(cls,Equation) or cls is Equation or cls in seen or cls.__module__!=mod_.__name__: continue
        seen.add(cls)
        if not any(callable(getattr(cls,m,None)) for m in METHODS): continue
        kw={}
        try: sig=inspect.signature(cls.__init__)
        except Exception: continue
        for pn,pp in list(sig.parameters.items())[1:]:
            if pn=='dest': kw[pn]='fluid'
            elif pn=='sources': kw[pn]=['fluid']
            elif pp.default is inspect._empty and pp.kind in (pp.POSITIONAL_OR_KEYWORD,):
                kw[pn]= 2 if pn=='dim' else 1.0
        # arrays indexed with a literal (d_G[0]) are ParticleArray constants
        consts = {}
        for m_ in METHODS:
            fn = getattr(cls, m_, None)
            if fn is None: continue
            try: tree = __import__('ast').parse(__import__('textwrap').dedent(inspect.getsource(fn)))
            except Exception: continue
            import ast as _ast
            for nd in _ast.walk(tree):
                if isinstance(nd, _ast.Subscript) and isinstance(nd.value, _ast.Name) and \
                        nd.value.id.startswith('d_') and isinstance(nd.slice, _ast.Constant):
                    consts[nd.value.id[2:]] = np.zeros(4)
        AnyArray.constants = consts
        try: eq=cls(**kw)
        except Exception as e:
            cons.append((cls.__module__+'.'+name, str(e)[:60])); continue
        try:
            fam_ = GeneratedFamily('fluid',[eq],{'fluid':AnyArray()},2,'c'); ok.append(cls.__module__+'.'+name)
            if os.environ.get('CENSUS_BUILD') == 'f32':       # the float build of every family (codegen.source_f32): compiled below
                builds.append((cls.__module__+'.'+name, fam_))
            elif os.environ.get('CENSUS_BUILD'):
                try: fam_.build()
                except Exception as e: bad[cls.__module__+'.'+name] = 'BUILD ' + str(e)[-400:]
        except CodegenError as e: bad[cls.__module__+'.'+name]=str(e)[:160]
        except Exception as e: bad[cls.__module__+'.'+name]='EXC '+type(e).__name__+': '+str(e)[:120]
if builds:
    from concurrent.futures import ThreadPoolExecutor
    def _b(item):
        try: item[1].flavour_f32().build(); return None
        except Exception as e: return (item[0], 'BUILD f32 ' + str(e)[-600:])
    with ThreadPoolExecutor(max_workers=os.cpu_count() or 4) as pool:
        for r in pool.map(_b, builds):
            if r: bad[r[0]] = r[1]
print(json.dumps({'ok': ok, 'bad': bad, 'construct_failed': [c[0] for c in cons],
                  'import_failed': [f[0] for f in impfail]}))
