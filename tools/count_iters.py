"""Phase-2 iteration counters of the pair kernel (option count_iters) for the
headline cube, Taylor-Green and S-rings3d, with and without the mask
normalisation: iterations / calls / wavefronts / row tiles per evaluation
(calls > wavefronts: lists flushed early).  Prints to stderr (dump_counters)."""
import sys
sys.path.insert(0, '/root/repo')
import torch, bench
from pysph_amd import device as dev
for argv in (['--n1', '159'], ['--n1', '159', '--workload', 'taylor_green'], ['--workload', 'elastic']):
    for norm in (1, 0):
        args = bench.parse_args(argv)
        ctx = dev.HipContext(0, torch.cuda.current_stream().cuda_stream)
        bench.apply_options(args, ctx)
        ctx.set_option('norm_masks', norm)
        ctx.set_option('nl_reuse', 0)
        w = bench.build_workload(args, 0, 1)
        nnps, a_eval, halo, domain, step, ordered = bench.setup(args, w, 0, 1, None, ctx)
        step()
        ctx.set_option('count_iters', 1)
        step()
        print(argv, 'norm_masks', norm, flush=True)
        sys.stderr.write('%s norm_masks=%d: ' % (' '.join(argv), norm))
        ctx.set_option('dump_counters', 1)
        del nnps, a_eval, step
        ctx.close()
        torch.cuda.empty_cache()
