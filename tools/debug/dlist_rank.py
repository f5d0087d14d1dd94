import os, sys, json
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import torch, bench
from pysph_amd import device as dev
for opt in (1, 0):
    args = bench.parse_args(['--workload', 'dam_break', '--dx', '0.0035', '--emulate-rank', '3/8', '--opt', 'dest_list=%d' % opt])
    ts = torch.cuda.Stream(); torch.cuda.set_stream(ts)
    ctx = dev.HipContext(0, ts.cuda_stream)
    bench.apply_options(args, ctx)
    w = bench.build_workload(args, 0, 1)
    nnps, a_eval, halo, domain, step, _ = bench.setup(args, w, 0, 1, None, ctx)
    el, tm = bench.timed(20, 5, step, torch.cuda.synchronize, ctx)
    print(opt, el / 20 * 1e3, {k: v[0] / 20 for k, v in tm.items() if k in ('nnps', 'pack', 'eos', 'pair')},
          'dlist launches', ctx.timer_get('n_dest_list')[1], 'merged', ctx.timer_get('n_merged')[1],
          [(a.gpu.get_number_of_particles(True), a.gpu.get_number_of_particles()) for a in w.arrays], flush=True)
    del nnps, a_eval, step
    ctx.close()
