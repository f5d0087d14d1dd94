#!/usr/bin/env python
"""Per-phase cost of one ghost exchange on ONE GPU (the slab is its own periodic
neighbour over RCCL): every phase bracketed by a device synchronisation, so the
numbers are upper bounds of what the unsynchronised exchange pays.

    python tools/halo_profile.py [--workload cube|taylor_green] [--n1 159]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    import torch
    import torch.distributed as dist
    argv = ['--self-slab', '--no-check', '--no-extras', '--no-cpu-baseline'] + sys.argv[1:]
    sys.argv = [sys.argv[0]] + argv
    args = bench.parse_args()
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29544')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    from pysph_amd import device as dev
    ts = torch.cuda.Stream()
    torch.cuda.set_stream(ts)
    ctx = dev.HipContext(0, ts.cuda_stream)
    bench.apply_options(args, ctx)
    w = bench.build_workload(args, 0, 1)
    nnps, a_eval, halo, domain, step, _ = bench.setup(args, w, 0, 1, dist, ctx)
    for _ in range(3):
        step()
    h = halo.halos[0]
    ops = h.ops
    acc = {}

    def tick(name, t0):
        torch.cuda.synchronize()
        acc[name] = acc.get(name, 0.0) + (time.perf_counter() - t0)
        return time.perf_counter()

    reps = 10
    for _ in range(reps):
        if domain is not None:
            # the domain manager drops every ghost first (HipDomainManager.update)
            n = ops.n_real()
            dev._check(ops.lib.sph_array_resize(ctx._h, ops.id, n, n))
        torch.cuda.synchronize()
        t = time.perf_counter()
        ops.drop_ghosts(); t = tick('drop', t)
        n_lo, n_hi = ops.select(h.lo + h.width, h.hi - h.width); t = tick('select (flags, scans, count readback)', t)
        nbrs = h.neighbours()
        send = {0: n_lo, 1: n_hi}
        out = {s: ops.pack(s, send[s], shift) for s, _, shift in nbrs}; t = tick('pack x2', t)
        mine = ops.int_tensor([send[0], send[1]])
        allc = ops.int_tensor([0, 0])
        dist.all_gather_into_tensor(allc, mine)
        allc = [int(v) for v in allc.cpu()]; t = tick('counts all_gather + readback', t)
        inb = {s: ops.new_buffer(allc[1 - s], ops.nprops) for s, _, _ in nbrs}
        reqs = [dist.P2POp(dist.isend, out[s], 0) for s in (1, 0)] + [dist.P2POp(dist.irecv, inb[s], 0) for s in (0, 1)]
        for wk in dist.batch_isend_irecv(reqs):
            wk.wait()
        t = tick('send/recv', t)
        for s in (0, 1):
            ops.append(inb[s], allc[1 - s])
        t = tick('append x2', t)
    total = 0.0
    for k, v in acc.items():
        print('%-42s %7.1f us' % (k, v / reps * 1e6))
        total += v / reps
    print('%-42s %7.1f us   (ghosts per face %d / %d, %d properties)' % ('sum', total * 1e6, n_lo, n_hi, ops.nprops))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        if domain is not None:
            nnps.update_domain()
        else:
            halo.exchange()
    torch.cuda.synchronize()
    print('%-42s %7.1f us' % ('exchange as the step runs it', (time.perf_counter() - t0) / reps * 1e6))
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
