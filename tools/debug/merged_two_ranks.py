"""debug: does the merged one-launch path engage on two slab ranks of a dam break?"""
import os, sys, threading
import numpy as np
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'tests'))
import bench
from helpers import ThreadDist
from pysph_amd import device as dev
hub = ThreadDist(2)
def rank_main(rank):
    try:
        args = bench.parse_args(['--workload', 'dam_break', '--dx', '0.05', '--gpus', '2'])
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            ctx = dev.HipContext(0, stream.cuda_stream)
            bench.apply_options(args, ctx)
            w = bench.build_workload(args, rank, 2)
            nnps, a_eval, halo, domain, step, _ = bench.setup(args, w, rank, 2, hub.view(rank), ctx)
            for k in range(6):
                step()
                print(rank, k, 'merged', ctx.timer_get('n_merged')[1], 'async', ctx.timer_get('n_async')[1],
                      'umass', ctx.timer_get('n_mass_fused')[1], 'padded', [h.padded_exchanges for h in halo.halos],
                      'n', [(a.gpu.get_number_of_particles(True), a.gpu.get_number_of_particles()) for a in w.arrays], flush=True)
    except BaseException:
        import traceback; traceback.print_exc()
        hub.barrier.abort()
ths = [threading.Thread(target=rank_main, args=(r,)) for r in range(2)]
[t.start() for t in ths]; [t.join(600) for t in ths]
