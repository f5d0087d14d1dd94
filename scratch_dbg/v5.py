import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import numpy as np
import bench
from pysph_amd import device as dev, kernels as K
from pysph_amd.acceleration_eval import AccelerationEval, SPHCompiler
from pysph_amd.nnps import HipNNPS
from helpers import rel_err

def run(n1, lds, varh=0.0, reorder=True):
    pa, dx = bench.make_cube(n1)
    if varh:
        rng = np.random.default_rng(3)
        pa.h[:] *= 1 + varh * rng.uniform(-1, 1, pa.h.size)
    eqs = bench.cube_equations(dx)
    kernel = K.WendlandQuintic(dim=3)
    ctx = dev.HipContext(0)
    ctx.set_option('lds_records', lds)
    a_eval = AccelerationEval([pa], eqs, kernel)
    SPHCompiler(a_eval, ctx=ctx).compile()
    nnps = HipNNPS(3, [pa], radius_scale=2.0, ctx=ctx)
    a_eval.set_nnps(nnps)
    a_eval.compute(0.0, 1e-5)
    return pa

for n1, varh in ((30, 0.0), (30, 0.1), (64, 0.0)):
    a = run(n1, 0, varh); b = run(n1, 1, varh)
    worst = max(rel_err(a.properties[p], b.properties[p]) for p in ('arho','au','av','aw','ax','ay','az','dt_cfl','dt_force'))
    print('n1', n1, 'varh', varh, 'max rel err lds vs agg', worst)
