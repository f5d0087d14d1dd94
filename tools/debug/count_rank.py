"""Row tiles and phase-2 trips of the merged pair kernel per wavefront: the whole 4.65 M tank against ONE rank's slab of the
17.3 M tank (needs the profiling build: SPH_LIBRARY=pysph_amd/libsphhip_prof.so)"""
import sys
sys.path.insert(0, '/root/repo')
import torch, bench
from pysph_amd import device as dev
for argv in (['--workload', 'dam_break', '--dx', '0.0055'], ['--workload', 'dam_break', '--dx', '0.0035', '--emulate-rank', '4/8'],
             ['--workload', 'dam_break', '--dx', '0.0035', '--emulate-rank', '4/8', '--slab-axis', '1']):
    args = bench.parse_args(argv + ['--no-extras', '--no-cpu-baseline'])
    ctx = dev.HipContext(0, torch.cuda.current_stream().cuda_stream)
    bench.apply_options(args, ctx)
    w = bench.build_workload(args, 0, 1)
    nnps, a_eval, halo, domain, step, ordered = bench.setup(args, w, 0, 1, None, ctx)
    for _ in range(3):
        step()
    ctx.set_option('count_iters', 1)
    step()
    n = sum(a.get_number_of_particles() for a in w.arrays)
    nr = sum(a.get_number_of_particles(True) for a in w.arrays)
    sys.stderr.write('%s: rows %d real %d: ' % (' '.join(argv), n, nr))
    sys.stderr.flush()
    ctx.set_option('dump_counters', 1)
    del nnps, a_eval, step
    ctx.close()
    torch.cuda.empty_cache()
