"""Time the neighbour-list (CSR) passes at the headline size: count pass and
count + fill (device lists of the generated loop_all path), variant 6 (the
wave-tile pair kernel) against variant 0 (per-particle 27-cell walk)."""
import sys
import time
sys.path.insert(0, '/root/repo')
import numpy as np
import torch, bench
from pysph_amd import device as dev

for variant in (6, 0):
    args = bench.parse_args(['--n1', '159', '--variant', str(variant)])
    ctx = dev.HipContext(0, torch.cuda.current_stream().cuda_stream)
    bench.apply_options(args, ctx)
    w = bench.build_workload(args, 0, 1)
    nnps, a_eval, halo, domain, step, ordered = bench.setup(args, w, 0, 1, None, ctx)
    step()
    for rep in range(3):
        ctx.synchronize()
        t0 = time.perf_counter()
        tot = nnps.count_neighbors(0, 0)
        ctx.synchronize()
        t1 = time.perf_counter()
    print('variant %d: count pass %d pairs in %.3f ms (host round trip and %d-entry copy included)' % (
        variant, tot, (t1 - t0) * 1e3, 159 ** 3), flush=True)
    del nnps, a_eval, step
    ctx.close()
    torch.cuda.empty_cache()
