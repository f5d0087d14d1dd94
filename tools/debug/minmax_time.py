import os, sys, time, ctypes as C
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import torch, bench
from pysph_amd import device as dev
for n1 in (159, 100):
    args = bench.parse_args(['--n1', str(n1)])
    ts = torch.cuda.Stream(); torch.cuda.set_stream(ts)
    ctx = dev.HipContext(0, ts.cuda_stream)
    w = bench.build_workload(args, 0, 1)
    nnps, a_eval, halo, domain, step, _ = bench.setup(args, w, 0, 1, None, ctx)
    step(); step()
    ids = (C.c_int * 1)(w.arrays[0].gpu.array_id)
    out = (C.c_double * 8)()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        ctx.lib.sph_nnps_minmax(ctx._h, 1, ids, out)       # k_bin_keys (bounds, h, m: no keys, no histogram) + finish + readback
    torch.cuda.synchronize()
    print(n1, 'sph_nnps_minmax (sync each): %.1f us per call' % ((time.perf_counter() - t0) / 50 * 1e6), flush=True)
    del nnps, a_eval, step
    ctx.close()
