"""Minimal stand-in for the third-party `compyle` package (absent here).

TEST INFRASTRUCTURE ONLY.  It exists so that the reference's pure-Python
equation / kernel / scheme classes under /root/reference can be *imported and
executed as plain Python* by `tests/golden/make_golden.py` (the same way the
reference's own `pysph/sph/tests/test_equations.py:102-120` calls
`Equation.loop(...)` directly).  Nothing here translates or compiles code.
"""
