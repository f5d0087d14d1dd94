import os, sys, json
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import torch, bench
import torch.distributed as dist
from pysph_amd import device as dev
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29577')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
for ss in (False, True, True):
    argv = ['--workload', 'dam_break', '--dx', '0.0035', '--emulate-rank', '4/8'] + (['--self-slab'] if ss else [])
    args = bench.parse_args(argv)
    ts = torch.cuda.Stream(); torch.cuda.set_stream(ts)
    ctx = dev.HipContext(0, ts.cuda_stream)
    bench.apply_options(args, ctx)
    w = bench.build_workload(args, 0, 1)
    nnps, a_eval, halo, domain, step, _ = bench.setup(args, w, 0, 1, dist if ss else None, ctx)
    el, tm = bench.timed(20, 5, step, torch.cuda.synchronize, ctx)
    print(ss, round(el / 20 * 1e3, 4), {k: round(v[0] / 20, 4) for k, v in tm.items() if k in ('nnps', 'pack', 'eos', 'pair')},
          'merged', ctx.timer_get('n_merged')[1], 'async', ctx.timer_get('n_async')[1],
          [(a.gpu.get_number_of_particles(True), a.gpu.get_number_of_particles()) for a in w.arrays],
          None if halo is None else [(h.padded_exchanges, h.ops.nprops, dict(h.cap_send), h.__dict__.get('cap_hist'), h.last_counts) for h in halo.halos], flush=True)
    del nnps, a_eval, step, halo
    ctx.close()
dist.destroy_process_group()
