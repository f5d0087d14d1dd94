import sys, json
sys.path.insert(0, '/root/repo')
import torch, bench
from pysph_amd import device as dev
for argv in (['--n1','159'], ['--n1','159','--workload','taylor_green'], ['--workload','elastic','--n1','126']):
    args = bench.parse_args(argv)
    ctx = dev.HipContext(0, torch.cuda.current_stream().cuda_stream)
    bench.apply_options(args, ctx)
    w = bench.build_workload(args, 0, 1)
    nnps, a_eval, halo, domain, step, ordered = bench.setup(args, w, 0, 1, None, ctx)
    step()
    ctx.set_option('count_iters', 1)
    step()
    print(argv, flush=True)
    ctx.set_option('dump_counters', 1)
    ctx.close()
