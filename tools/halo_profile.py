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
    # the round-3 exchange, phase by phase: device-side select + pack, transfer of the
    # fixed-capacity messages, header readback, strided append
    acc3 = {}

    def tick3(name, t0):
        torch.cuda.synchronize()
        acc3[name] = acc3.get(name, 0.0) + (time.perf_counter() - t0)
        return time.perf_counter()
    cap = {s: h.cap_send[s] for s in (0, 1)}
    capr = {s: h.cap_recv[s] for s in (0, 1)}
    shift_of = {s: shift for s, _, shift in h.neighbours()}
    for _ in range(reps):
        if domain is not None:
            n = ops.n_real()
            dev._check(ops.lib.sph_array_resize(ctx._h, ops.id, n, n))
        torch.cuda.synchronize()
        t = time.perf_counter()
        ops.drop_ghosts(); t = tick3('drop', t)
        out = {s: ops.message_buffer(('send', s), cap[s] * ops.nprops + 1) for s in (0, 1)}
        inb = {s: ops.message_buffer(('recv', s), capr[s] * ops.nprops + 1) for s in (0, 1)}
        t = tick3('buffers (kept from exchange to exchange)', t)
        ops.select_pack(h.lo + h.width, h.hi - h.width, [shift_of[0], shift_of[1]], [cap[0], cap[1]],
                        [out[0], out[1]]); t = tick3('select_pack (block counts, scan, pack)', t)
        reqs = [dist.P2POp(dist.isend, out[s], 0) for s in (1, 0)] + [dist.P2POp(dist.irecv, inb[s], 0) for s in (0, 1)]
        for wk in dist.batch_isend_irecv(reqs):
            wk.wait()
        t = tick3('send/recv of the capacity-sized messages', t)
        hdr = ops.read_headers([out[0], out[1], inb[0], inb[1]])
        t = tick3('header readback (sph_read_values)', t)
        for s in (0, 1):
            ops.append(inb[s], int(abs(hdr[2 + s])), stride=capr[s])
        t = tick3('append x2 (strided)', t)
    print('round-3 protocol (capacities %d / %d rows):' % (cap[0], cap[1]))
    tot3 = 0.0
    for k, v in acc3.items():
        print('  %-40s %7.1f us' % (k, v / reps * 1e6))
        tot3 += v / reps
    print('  %-40s %7.1f us' % ('sum', tot3 * 1e6))
    print('list-based protocol of round 2:')
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
