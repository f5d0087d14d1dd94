"""pysph_amd -- MI355X (gfx950) acceleration-eval backend for PySPH-style SPH.

One hot path, built MI355X-first: cell-list neighbour search + the per-pair
summation loops of ``AccelerationEval.compute()``, as hand-written HIP kernels
behind a C-ABI (``include/sphhip.h``), driven from Python through ctypes.
See DESIGN.md.
"""
__version__ = '0.1.0'
