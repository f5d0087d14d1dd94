import sys
sys.path.insert(0, '/root/repo')
import torch, bench
from pysph_amd import device as dev
args = bench.parse_args(['--no-extras', '--no-cpu-baseline'] + sys.argv[1:])
ctx = dev.HipContext(0, torch.cuda.current_stream().cuda_stream)
bench.apply_options(args, ctx)
w = bench.build_workload(args, 0, 1)
nnps, a_eval, halo, domain, step, ordered = bench.setup(args, w, 0, 1, None, ctx)
step(); step()
torch.cuda.synchronize()
ctx.timer_enable(1); ctx.timer_reset()
for k in range(8):
    step()
    torch.cuda.synchronize()
    print(k, {n: (round(ctx.timer_get(n)[0], 4), ctx.timer_get(n)[1]) for n in ('nnps', 'pack', 'eos', 'pair')})
