import ast


class KnownType(object):
    def __init__(self, type_str, base_type=''):
        self.type = type_str
        self.base_type = base_type


class _NoCodegen(object):
    def __init__(self, *a, **kw):
        pass

    def __getattr__(self, name):
        raise RuntimeError("compyle stub: code generation is not available")


CythonGenerator = _NoCodegen
OpenCLConverter = _NoCodegen


def get_symbols(code, ctx=(ast.Load, ast.Store)):
    """Names (ast.Name ids) used in `code` (a string or an AST)."""
    tree = ast.parse(code) if isinstance(code, str) else code
    return set(n.id for n in ast.walk(tree)
               if isinstance(n, ast.Name) and isinstance(n.ctx, ctx))


def declare(*args, **kw):
    return None


def annotate(*a, **kw):
    def wrap(f):
        return f
    return wrap
