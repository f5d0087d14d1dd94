"""Slab decomposition + ghost-particle halo exchange (one process per GPU).

Replaces, for the single-node 8-GPU case, what the reference does with MPI +
Zoltan in ``pysph/parallel/parallel_manager.pyx`` (``ParallelManager.update``
:512-530 = remove Remote particles -> compute_remote_particles :1159-1243 ->
remote_exchange_data :159-210 appends ghosts tagged Remote):

* the domain is cut into slabs along one axis, one slab per rank;
* before every acceleration evaluation each rank drops its old ghosts, selects
  its REAL particles within ``width`` (= n_layers * radius_scale * hmax, one
  cell) of each slab face, packs the halo properties on the device
  (``sph_halo_select/pack``) and swaps them with its <=2 neighbours by RCCL
  point-to-point send/recv (``torch.distributed`` batch_isend_irecv: direct
  xGMI links, no all-to-all); received particles are appended as ghosts
  (index >= n_real: sources only -- exactly the Remote-tag semantics,
  ``Group(real=True)`` skips them as destinations, ``real=False`` groups (EOS)
  recompute p, cs on them locally: scheme.py:412,432);
* scalars (adaptive dt inputs, global bounds) use all_reduce MIN/MAX, replacing
  parallel_manager.pyx:463,937-945.

The transport layer is written against a small ``ops`` object so that the
rank topology / counts handshake / periodic shift logic is exercised on CPU
under gloo in tests/ with a numpy test double; the product ops
(``DeviceHaloOps``) call the HIP kernels through the C-ABI and there is no
host fallback inside the package.
"""
import ctypes as C
import os

from . import device as dev

# what a WCSPH ghost needs (72 B/particle); p, cs are recomputed
WCSPH_HALO_PROPS = ('x', 'y', 'z', 'u', 'v', 'w', 'rho', 'h', 'm')
TVF_HALO_PROPS = WCSPH_HALO_PROPS + ('uhat', 'vhat', 'what')
# elastic solids: cs and the deviatoric stress travel; p and the artificial stress
# r_ij are recomputed on the ghosts (ElasticSolidsScheme(ghost_recompute=True))
ELASTIC_HALO_PROPS = WCSPH_HALO_PROPS + ('cs', 's00', 's01', 's02', 's11', 's12', 's22')


class DeviceHaloOps(object):
    """HIP implementation of the pack/append primitives (C-ABI)."""

    def __init__(self, pa, ctx, props, axis):
        import torch
        self.torch = torch
        self.pa = pa
        self.ctx = ctx
        self.lib = ctx.lib
        self.gpu = dev.attach(pa, ctx)
        # Both this halo and HipDomainManager keep their ghosts as the rows
        # behind n_real and drop ALL of them on update (the reference removes
        # only Remote- resp. Ghost-tagged rows, parallel_manager.pyx:512-530,
        # nnps_base.pyx:485-520): on one array each would delete the other's.
        # The periodic wrap ALONG the slab axis is this halo's own job
        # (periodic=True); other periodic / mirror axes are not supported
        # together with slabs.
        if getattr(self.gpu, 'ghost_owner', None) not in (None, 'slab', 'domain+slab'):
            raise RuntimeError(
                "array '%s' already has device ghosts managed by %s; a slab halo "
                "on the same array would delete them" % (pa.name, self.gpu.ghost_owner))
        if getattr(self.gpu, 'ghost_owner', None) is None:
            self.gpu.ghost_owner = 'slab'
        self.gpu.managed = True
        self.id = self.gpu.array_id
        # gid identifies a particle across ranks (1-vs-N parity checks, output):
        # mirror it on the device (as a double, exact below 2^53) so that
        # migration carries it along with the fp64 properties; tag and pid are
        # re-derived on the host (rows >= n_real are Remote, pid = rank)
        if 'gid' in getattr(pa, 'properties', {}):
            dev.prop_register('gid')
            self.gpu.push('gid')
        self.axis = axis
        self.nprops = len(props)
        self.props = (C.c_int * self.nprops)(*[dev.prop_id(p) for p in props])
        self.device = torch.device('cuda', ctx.device)

    def n_real(self):
        return self.gpu.get_number_of_particles(True)

    def drop_ghosts(self):
        n = self.n_real()
        dev._check(self.lib.sph_array_resize(self.ctx._h, self.id, n, n))

    def select(self, lo_cut, hi_cut):
        counts = (C.c_size_t * 2)()
        dev._check(self.lib.sph_halo_select(self.ctx._h, self.id, self.axis,
                                            0, lo_cut, hi_cut, 0.0, 0, counts))
        return int(counts[0]), int(counts[1])

    def new_buffer(self, count, nprops=None):
        return self.torch.empty(max(count * (nprops or self.nprops), 1),
                                dtype=self.torch.float64, device=self.device)

    def int_tensor(self, values):
        return self.torch.tensor(values, dtype=self.torch.int64,
                                 device=self.device)

    # -- ordering between this context's HIP stream and the transport -------
    # RCCL orders its work after (and `work.wait()` orders it before) torch's
    # CURRENT stream.  When the context runs on that very stream (bench.py and
    # the integrator pipeline create it that way) the pack kernels, the
    # transfers and the append kernels are ordered by the stream itself and no
    # host synchronisation is needed; a context with a stream of its own needs
    # two host-side syncs per exchange.
    def _shares_torch_stream(self):
        mine = self.ctx.stream
        mine = getattr(mine, 'value', mine)         # a ctypes.c_void_p handle
        if not mine:
            # no caller-supplied stream (None, or the null handle of torch's
            # default stream): the library runs on a stream of its own
            return False
        cur = self.torch.cuda.current_stream(self.device).cuda_stream
        return int(mine) == int(cur)

    def before_comm(self):
        if not self._shares_torch_stream():
            self.ctx.synchronize()      # payloads are complete before they are sent

    def after_comm(self):
        if not self._shares_torch_stream():
            # the receives are complete before the append kernels read them
            self.torch.cuda.current_stream(self.device).synchronize()

    def pack(self, side, count, shift, out=None):
        """[nprops][count] payload of the rows selected for `side`; `out`: a larger
        buffer to fill from its start (the fixed-capacity messages)"""
        buf = out if out is not None else self.new_buffer(count)
        dev._check(self.lib.sph_halo_pack(
            self.ctx._h, self.id, side, self.nprops, self.props, self.axis,
            float(shift), C.c_void_p(buf.data_ptr())))
        return buf

    def append(self, buf, count, stride=None):
        """`stride`: rows of a fixed-capacity message (property k of row i at
        buf[k * stride + i])"""
        dev._check(self.lib.sph_halo_append_strided(
            self.ctx._h, self.id, self.nprops, self.props,
            C.c_void_p(buf.data_ptr()), count, count if stride is None else stride))

    def message_buffer(self, key, size):
        """the fixed-capacity message `key` (face, direction): one device buffer
        kept from exchange to exchange while its capacity stays"""
        cache = self.__dict__.setdefault('_messages', {})
        t = cache.get(key)
        if t is None or t.numel() != size:
            t = cache[key] = self.torch.empty(size, dtype=self.torch.float64, device=self.device)
        return t

    def read_headers(self, tensors):
        """the last element of each message, in ONE device->host round trip
        (sph_read_values: all copies queued on the context's stream, one sync)"""
        n = len(tensors)
        ptrs = (C.c_void_p * n)(*[t.data_ptr() + (t.numel() - 1) * 8 for t in tensors])
        out = (C.c_double * n)()
        dev._check(self.lib.sph_read_values(self.ctx._h, n, ptrs, out))
        return [out[k] for k in range(n)]

    def read_headers_overlapped(self, tensors, works):
        """the same headers WITHOUT waiting for what the context's stream still has
        queued (the part of the evaluation that runs while the ghosts are in
        flight): a side stream waits for the transfers, gathers the headers and
        copies them to pinned memory; the host waits for that copy only."""
        torch = self.torch
        side = self.__dict__.get('_side')
        if side is None:
            side = self._side = torch.cuda.Stream(self.device)
            self._hdr_pin = torch.empty(64, dtype=torch.float64).pin_memory()
        n = len(tensors)
        with torch.cuda.stream(side):
            for w in works:
                w.wait()                     # stream-level: the side stream runs after the transfers
            h = torch.stack([t[-1] for t in tensors])
            self._hdr_pin[:n].copy_(h, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(side)
        ev.synchronize()
        return self._hdr_pin[:n].tolist()

    # -- the exchange without a device->host round trip ('padded' protocol) ----
    def append_padded(self, buf, cap, h_promise, m_promise):
        """all `cap` rows of a fixed-capacity message behind the particles: the
        first |header| are the ghosts, the rest padding rows parked at 1e18 (inert on
        the whole path); the host learns the count an exchange later
        (sph_halo_append_padded)"""
        dev._check(self.lib.sph_halo_append_padded(
            self.ctx._h, self.id, self.nprops, self.props, C.c_void_p(buf.data_ptr()), int(cap),
            float(h_promise), float(m_promise), C.c_void_p(self.flag_word().data_ptr())))

    def flag_word(self):
        """device word sph_halo_append_padded ORs into (bit 0: incomplete message, bit 1: broken promise)"""
        t = self.__dict__.get('_flag')
        if t is None:
            t = self._flag = self.torch.zeros(2, dtype=self.torch.int32, device=self.device)
        return t

    def queue_headers(self, tensors):
        """the last element of each message and the flag word on their way to pinned
        memory, behind an event nobody waits for now: `collect_headers` an exchange later"""
        torch = self.torch
        n = len(tensors)
        pin = self.__dict__.get('_hdr_pin2')
        if pin is None or pin.numel() < n + 1:
            pin = self._hdr_pin2 = torch.empty(max(n + 1, 64), dtype=torch.float64).pin_memory()
        vals = torch.stack([t[-1] for t in tensors] + [self.flag_word()[0].to(torch.float64)])
        pin[:n + 1].copy_(vals, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        return (ev, n)

    def collect_headers(self, handle):
        ev, n = handle
        ev.synchronize()
        vals = self._hdr_pin2[:n + 1].tolist()
        return vals[:n], int(vals[n])

    def hm_range(self):
        """(hmin, hmax, mmin, mmax) of the real particles (set-up only: one round trip each)"""
        out = []
        for prop, fn in (('h', self.lib.sph_reduce_min), ('h', self.lib.sph_reduce_max),
                         ('m', self.lib.sph_reduce_min), ('m', self.lib.sph_reduce_max)):
            v = C.c_double()
            if self.n_real() == 0 or dev.prop_id(prop) not in list(self.props):
                out.append(float('inf') if fn is self.lib.sph_reduce_min else -float('inf'))
                continue
            dev._check(fn(self.ctx._h, self.id, dev.prop_id(prop), C.byref(v)))
            out.append(v.value)
        return out

    def select_pack(self, lo_cut, hi_cut, shifts, caps, bufs):
        """Both faces selected AND packed on the device, no host round trip:
        bufs[side] (or None) is a message of caps[side] * nprops + 1 doubles,
        rows laid out [nprops][cap], the row count (negative: more than cap) in
        its last element (sph_halo_select_pack)."""
        sh = (C.c_double * 2)(float(shifts[0]), float(shifts[1]))
        cp = (C.c_size_t * 2)(int(caps[0]), int(caps[1]))
        ds = (C.c_void_p * 2)(*[b.data_ptr() if b is not None else None for b in bufs])
        dev._check(self.lib.sph_halo_select_pack(
            self.ctx._h, self.id, self.axis, float(lo_cut), float(hi_cut), 0,
            self.nprops, self.props, sh, cp, ds))

    # -- migration of owned particles (every device property travels) -------
    def all_props(self):
        """ids of the properties with device storage, ascending: identical
        on every rank because every rank runs the same equations"""
        return self.gpu.device_props()

    def pack_all(self, side, count, shift):
        ids = self.all_props()
        buf = self.new_buffer(count, len(ids))
        dev._check(self.lib.sph_halo_pack(
            self.ctx._h, self.id, side, len(ids), (C.c_int * len(ids))(*ids),
            self.axis, float(shift), C.c_void_p(buf.data_ptr())))
        return buf, len(ids)

    def remove_selected(self):
        left = C.c_size_t(0)
        dev._check(self.lib.sph_halo_remove_selected(self.ctx._h, self.id,
                                                     C.byref(left)))
        return int(left.value)

    def append_real(self, buf, count):
        ids = self.all_props()
        dev._check(self.lib.sph_halo_append(
            self.ctx._h, self.id, len(ids), (C.c_int * len(ids))(*ids),
            C.c_void_p(buf.data_ptr()), count))
        n = self.gpu.get_number_of_particles()
        dev._check(self.lib.sph_array_resize(self.ctx._h, self.id, n, n))

    def coords(self):
        """host copy of the real particles' coordinate along the slab axis
        (re-balancing only; every k steps)"""
        import numpy as np
        n = self.n_real()
        out = np.empty(n)
        if n:
            self.gpu.pull_into('xyz'[self.axis], out)
        return out


class SlabHalo(object):
    """Ghost exchange + migration for one particle array of a
    slab-decomposed domain.

    rank r owns coordinate range [lo, hi) along `axis`; its neighbours are
    r-1 and r+1 (wrapping with a coordinate shift of -+`period` when
    `periodic`)."""

    def __init__(self, pa, ctx, rank, world, axis, width, lo, hi,
                 props=WCSPH_HALO_PROPS, periodic=False, period=0.0,
                 ops=None, dist=None, protocol=None):
        if dist is None:
            import torch.distributed as dist
        self.dist = dist
        self.rank, self.world = rank, world
        self.axis = axis
        self.width = float(width)
        self.lo, self.hi = float(lo), float(hi)
        self.periodic = periodic
        self.period = float(period)
        self.ops = ops or DeviceHaloOps(pa, ctx, props, axis)
        self.last_counts = (0, 0, 0, 0)   # sent lo/hi, received lo/hi
        self.last_migrated = (0, 0, 0, 0)
        # ghost exchange protocol (exchange_halos): 'capacity' = fixed-size
        # messages with the row count in their last element, sized from the
        # count both ends saw in the previous exchange; 'handshake' = counts
        # all_gather before exactly-sized messages (always used the first time)
        # 'padded' = 'capacity' without the readback: the receiver appends every row of a message, the rows behind the
        # ghosts as padding parked far outside the domain, and learns the counts an exchange later (_exchange_padded)
        self.protocol = protocol or os.environ.get('SPH_HALO_PROTOCOL', 'capacity')
        self.cap_send, self.cap_recv = {}, {}
        self.handshakes = 0               # exchanges that needed the counts round
        # 'padded' protocol: the promises every ghost keeps (NaN: none), what the last padded exchange left to collect
        self.h_promise = self.m_promise = float('nan')
        self.promised = False
        self.padded_pending = None
        self.padded_exchanges = 0

    def neighbours(self):
        """[(side, peer rank, coordinate shift applied to what we SEND)]"""
        out = []
        r, w = self.rank, self.world
        if r > 0:
            out.append((0, r - 1, 0.0))
        elif self.periodic:          # w == 1: the slab is its own neighbour
            out.append((0, w - 1, +self.period))
        if r < w - 1:
            out.append((1, r + 1, 0.0))
        elif self.periodic:
            out.append((1, 0, -self.period))
        return out

    def _swap(self, nbrs, send_cnt, out_buf, nprops):
        """counts handshake (one tiny all_gather) + one batch of point-to-point
        send/recv with the <=2 neighbours; returns ({side: recv buffer},
        {side: recv count})"""
        ops, dist = self.ops, self.dist
        sync = getattr(ops, 'before_comm', None)
        if sync is not None:
            sync()
        mine = ops.int_tensor([send_cnt[0], send_cnt[1]])
        allc = ops.int_tensor([0] * (2 * self.world))
        dist.all_gather_into_tensor(allc, mine)
        allc = [int(v) for v in allc.cpu()]
        # what peer sends to me: its hi list if it is my lo neighbour, else lo
        recv_cnt = {s: allc[2 * peer + (1 - s)] for s, peer, _ in nbrs}
        in_buf = {s: ops.new_buffer(recv_cnt[s], nprops) for s, _, _ in nbrs}
        # Between one pair of ranks messages match in posting order.  With a
        # periodic axis and <= 2 ranks both faces talk to the SAME peer, whose
        # side-0 buffer expects my side-1 list: post the sends hi-face first,
        # the receives lo-face first.
        reqs = []
        for s, peer, _ in sorted(nbrs, key=lambda nb: -nb[0]):
            if send_cnt[s]:
                reqs.append(dist.P2POp(dist.isend, out_buf[s], peer))
        for s, peer, _ in sorted(nbrs, key=lambda nb: nb[0]):
            if recv_cnt[s]:
                reqs.append(dist.P2POp(dist.irecv, in_buf[s], peer))
        if reqs:
            for w in dist.batch_isend_irecv(reqs):
                w.wait()
        sync = getattr(ops, 'after_comm', None)
        if sync is not None:
            sync()
        return in_buf, recv_cnt

    def exchange(self, drop=True):
        """refresh the ghosts (ParallelManager.update, :512-530).  drop=False:
        the caller (HipDomainManager with slab=...) has already removed every
        ghost row and will add the periodic images of the other axes after."""
        exchange_halos([self], drop=drop)

    def migrate(self):
        """hand the REAL particles that left [lo, hi) to the neighbouring
        slab, with every property (parallel_manager.pyx:1085-1157: exported
        particles are sent, removed locally and arrive as Local particles).
        A particle moves at most one slab per call; returns the number that
        left this rank."""
        ops = self.ops
        ops.drop_ghosts()
        nbrs = self.neighbours()
        if not nbrs:
            return 0
        sides = [s for s, _, _ in nbrs]
        inf = float('inf')
        # an open (non-periodic) outer face keeps its particles
        n_lo, n_hi = ops.select(self.lo if 0 in sides else -inf,
                                self.hi if 1 in sides else inf)
        send_cnt = {0: n_lo, 1: n_hi}
        out_buf, nprops = {}, None
        for s, _, shift in nbrs:
            out_buf[s], nprops = ops.pack_all(s, send_cnt[s], shift)
        ops.remove_selected()
        in_buf, recv_cnt = self._swap(nbrs, send_cnt, out_buf, nprops)
        for s, _, _ in nbrs:
            if recv_cnt[s]:
                ops.append_real(in_buf[s], recv_cnt[s])
        self.last_migrated = (n_lo, n_hi, recv_cnt.get(0, 0),
                              recv_cnt.get(1, 0))
        return n_lo + n_hi


def _capacity(count):
    """rows a fixed-size ghost message is sized for, from the count both ends of
    the face saw last: a quarter of headroom + 4096 rows, in units of 1024"""
    return ((count + count // 4 + 4096 + 1023) // 1024) * 1024


def _next_capacity(cap, count):
    """the SAME rule on both ends of a face (sender: the count it sent;
    receiver: the count in the header), so the two stay in step without talking"""
    if cap is None or count > cap or count + count // 8 + 1024 > cap or 4 * count + 16384 < cap:
        return _capacity(count)
    return cap


def exchange_halos(hs, drop=True):
    """Ghost refresh of the arrays `hs`: see _exchange_steps (run to completion)."""
    for _ in _exchange_steps(hs, drop, False):
        pass


def exchange_halos_begin(hs, drop=True):
    """The first half of a ghost refresh: ghosts dropped, both faces selected and
    packed, the point-to-point transfers POSTED -- and control back to the caller,
    who can run whatever needs no ghosts (the neighbour update of the real
    particles, the interior part of the evaluation) before
    `exchange_halos_finish`.  Only the steady-state protocol splits (fixed-capacity
    messages packed on the device); the first exchange and the handshake
    protocol complete here.  Returns the state to hand to the second half."""
    steps = _exchange_steps(hs, drop, True)
    try:
        next(steps)
        return steps
    except StopIteration:
        return None


def exchange_halos_finish(steps):
    """The second half: wait for the transfers (the host only for the message
    headers, read on a side stream), append the ghosts behind the real particles."""
    if steps is not None:
        for _ in steps:
            pass


def _exchange_steps(hs, drop, overlap):
    """Ghost refresh of the arrays `hs` (SlabHalo objects of ONE rank, same slab),
    as a generator: with `overlap` it yields once, after the transfers were posted.

    'capacity' protocol (default): every face carries ONE fixed-size message per
    array -- [nprops][count] rows packed at its start, the row count in its last
    element -- whose size both ends derived from the previous exchange's count;
    one batch_isend_irecv, then ONE small device->host copy of the headers (the
    host needs the counts to resize the arrays).  A face that outgrew its
    capacity sends a negative header and the pair repeats that face with the
    exact size.  The very first exchange (no capacities yet) and
    SPH_HALO_PROTOCOL=handshake use a counts all_gather before exactly-sized
    messages.  Between one pair of ranks messages match in posting order: sends
    hi-face first, receives lo-face first (periodic axis with <= 2 ranks: both
    faces talk to the same peer), arrays in the same order on both sides."""
    h0 = hs[0]
    dist, ops0, world, na = h0.dist, h0.ops, h0.world, len(hs)
    nbrs = h0.neighbours()
    sides = [s for s, _, _ in nbrs]
    if nbrs and drop and not overlap and h0.protocol == 'padded' and all(
            hasattr(h.ops, 'append_padded') and hasattr(h.ops, 'select_pack') and h.promised and
            getattr(h.ops, '_shares_torch_stream', lambda: True)() and
            all(h.cap_send.get(s) is not None and h.cap_recv.get(s) is not None for s in sides) for h in hs):
        _exchange_padded(hs, nbrs)
        return
    _padded_collect(hs)           # (a padded exchange may be outstanding: its counts size this one)
    for h in hs:
        if drop:
            h.ops.drop_ghosts()
    if not nbrs:
        return
    if h0.protocol == 'padded' and not all(h.promised for h in hs):
        _establish_promises(hs)   # collective; this exchange goes the counted way anyway
    fixed = h0.protocol in ('capacity', 'padded') and all(
        h.cap_send.get(s) is not None and h.cap_recv.get(s) is not None for h in hs for s in sides)
    # with fixed-capacity messages and device-side packing the host never sees
    # the selection: the counts come back with the headers
    device_pack = fixed and all(hasattr(h.ops, 'select_pack') for h in hs)
    send = [{0: 0, 1: 0} for _ in hs]
    if not device_pack:
        for c, h in zip(send, hs):
            c[0], c[1] = h.ops.select(h.lo + h.width, h.hi - h.width)
            for s in (0, 1):        # nothing goes out through an open face
                if s not in sides:
                    c[s] = 0
    send_order = sorted(nbrs, key=lambda nb: -nb[0])
    recv_order = sorted(nbrs, key=lambda nb: nb[0])

    def comm_sync(which):
        for h in hs:
            f = getattr(h.ops, which, None)
            if f is not None:
                f()

    def run(reqs):
        if reqs:
            for w in dist.batch_isend_irecv(reqs):
                w.wait()

    if not fixed:
        out = [{s: h.ops.pack(s, c[s], shift) for s, _, shift in nbrs} for h, c in zip(hs, send)]
        comm_sync('before_comm')
        mine = ops0.int_tensor([c[s] for c in send for s in (0, 1)])
        allc = ops0.int_tensor([0] * (2 * na * world))
        dist.all_gather_into_tensor(allc, mine)
        allc = [int(v) for v in allc.cpu()]
        # what peer sends to me: its hi list if it is my lo neighbour, else lo
        recv = [{s: allc[2 * na * peer + 2 * a + (1 - s)] for s, peer, _ in nbrs} for a in range(na)]
        inb = [{s: h.ops.new_buffer(recv[a][s], h.ops.nprops) for s, _, _ in nbrs}
               for a, h in enumerate(hs)]
        reqs = []
        for s, peer, _ in send_order:
            for a in range(na):
                if send[a][s]:
                    reqs.append(dist.P2POp(dist.isend, out[a][s], peer))
        for s, peer, _ in recv_order:
            for a in range(na):
                if recv[a][s]:
                    reqs.append(dist.P2POp(dist.irecv, inb[a][s], peer))
        run(reqs)
        comm_sync('after_comm')
        for h in hs:
            h.handshakes += 1
    else:
        import torch
        shift_of = {s: shift for s, _, shift in nbrs}
        out, inb = [], []
        for a, h in enumerate(hs):
            npr, oa, ia = h.ops.nprops, {}, {}
            persistent = getattr(h.ops, 'message_buffer', None)
            for s in sides:
                if persistent is not None:
                    oa[s] = persistent(('send', s), h.cap_send[s] * npr + 1)
                    ia[s] = persistent(('recv', s), h.cap_recv[s] * npr + 1)
                else:
                    oa[s] = h.ops.new_buffer(h.cap_send[s] * npr + 1, 1)
                    ia[s] = h.ops.new_buffer(h.cap_recv[s] * npr + 1, 1)
            if device_pack:
                # selection and packing of both faces in one device pass, the row
                # counts stay on the device (headers of the messages)
                h.ops.select_pack(h.lo + h.width, h.hi - h.width,
                                  [shift_of.get(0, 0.0), shift_of.get(1, 0.0)],
                                  [h.cap_send.get(0, 0), h.cap_send.get(1, 0)],
                                  [oa.get(0), oa.get(1)])
            else:
                for s in sides:
                    cap, cnt = h.cap_send[s], send[a][s]
                    if cnt <= cap:
                        packed = h.ops.pack(s, cnt, shift_of[s])
                        # rows [nprops][cap]: property k of row i at k * cap + i
                        oa[s][:cap * npr].view(npr, cap)[:, :cnt] = packed[:cnt * npr].view(npr, cnt)
                    oa[s][-1] = float(cnt if cnt <= cap else -cnt)
            out.append(oa)
            inb.append(ia)
        comm_sync('before_comm')
        reqs = [dist.P2POp(dist.isend, out[a][s], peer) for s, peer, _ in send_order for a in range(na)]
        reqs += [dist.P2POp(dist.irecv, inb[a][s], peer) for s, peer, _ in recv_order for a in range(na)]
        split = overlap and device_pack and hasattr(ops0, 'read_headers_overlapped') and \
            ops0._shares_torch_stream()
        keys = [(a, s) for a in range(na) for s in sides]
        msgs = [out[a][s] for a, s in keys] + [inb[a][s] for a, s in keys]
        if split:
            works = list(dist.batch_isend_irecv(reqs)) if reqs else []
            yield 'posted'              # the caller's ghost-free work goes here
            hdr_early = ops0.read_headers_overlapped(msgs, works)
            for w in works:
                w.wait()                # stream-level: what follows on the context's stream runs after the transfers
        else:
            hdr_early = None
            run(reqs)
        # the ONE readback of the exchange: the row counts this rank packed and
        # the ones its peers packed (the host sizes the arrays with them)
        if hdr_early is not None:
            hdr = hdr_early
        elif hasattr(ops0, 'read_headers'):
            comm_sync('after_comm')     # (a context with a stream of its own: the receives are complete first)
            hdr = ops0.read_headers(msgs)
        else:
            hdr = torch.stack([m[-1] for m in msgs]).cpu().tolist()
        sent = {k: int(v) for k, v in zip(keys, hdr[:len(keys)])}
        hdr = {k: int(v) for k, v in zip(keys, hdr[len(keys):])}
        for a, s in keys:
            send[a][s] = abs(sent[(a, s)])
        recv = [{s: abs(hdr[(a, s)]) for s in sides} for a in range(na)]
        stride = [{s: hs[a].cap_recv[s] for s in sides} for a in range(na)]
        # faces that outgrew their capacity: the pair repeats them, exactly sized
        over_send = [k for k in keys if sent[k] < 0]
        over_recv = [k for k in keys if hdr[k] < 0]
        if over_send or over_recv:
            if over_send and device_pack:
                for a in sorted(set(a for a, _ in over_send)):
                    h = hs[a]               # the index lists of the faces, this time
                    h.ops.select(h.lo + h.width, h.hi - h.width)
            for a, s in over_send:
                out[a][s] = hs[a].ops.pack(s, send[a][s], shift_of[s])
            for a, s in over_recv:
                inb[a][s] = hs[a].ops.new_buffer(recv[a][s], hs[a].ops.nprops)
                stride[a][s] = recv[a][s]
            comm_sync('before_comm')
            reqs = [dist.P2POp(dist.isend, out[a][s], peer) for s, peer, _ in send_order
                    for a in range(na) if (a, s) in over_send]
            reqs += [dist.P2POp(dist.irecv, inb[a][s], peer) for s, peer, _ in recv_order
                     for a in range(na) if (a, s) in over_recv]
            run(reqs)
        comm_sync('after_comm')
    # ghosts go behind the real particles (lo side first: deterministic)
    for a, h in enumerate(hs):
        for s, _, _ in nbrs:
            if recv[a][s]:
                if fixed:
                    h.ops.append(inb[a][s], recv[a][s], stride=stride[a][s])
                else:
                    h.ops.append(inb[a][s], recv[a][s])
            h.cap_send[s] = _next_capacity(h.cap_send.get(s), send[a][s])
            h.cap_recv[s] = _next_capacity(h.cap_recv.get(s), recv[a][s])
        h.last_counts = (send[a][0], send[a][1], recv[a].get(0, 0), recv[a].get(1, 0))


def _establish_promises(hs):
    """'padded' protocol, once: does every rank hold ONE smoothing length and ONE
    mass per array?  (min/max over the ranks of each array's own range.)  Then
    every ghost is promised to carry them, and the neighbour update keeps what it
    knows of h and m across the appends (no look, no round trip); sph_halo_append_padded
    checks the promise on the device."""
    h0 = hs[0]
    lo, hi = [], []
    for h in hs:
        r = h.ops.hm_range() if hasattr(h.ops, 'hm_range') else [float('inf'), -float('inf')] * 2
        lo += [r[0], r[2]]
        hi += [r[1], r[3]]
    dev_t = getattr(h0.ops, 'device', None)
    glo = allreduce_scalars(lo, 'min', dist=h0.dist, device=dev_t)
    ghi = allreduce_scalars(hi, 'max', dist=h0.dist, device=dev_t)
    for a, h in enumerate(hs):
        h.h_promise = glo[2 * a] if glo[2 * a] == ghi[2 * a] else float('nan')
        h.m_promise = glo[2 * a + 1] if glo[2 * a + 1] == ghi[2 * a + 1] else float('nan')
        h.promised = True


def _padded_collect(hs):
    """what the last 'padded' exchange sent on its way -- the row counts both ends
    packed and the device flag word -- read now, one exchange later: capacities
    follow the counts (the same rule on both ends of a face, from the same pair),
    an incomplete message or a broken promise is an error (the step that used the
    ghosts is invalid; it cannot be repaired after the fact)."""
    h0 = hs[0]
    pend = h0.padded_pending
    if pend is None:
        return
    h0.padded_pending = None
    handle, keys = pend
    vals, flag = h0.ops.collect_headers(handle)
    nk = len(keys)
    sent = {k: int(v) for k, v in zip(keys, vals[:nk])}
    recv = {k: int(v) for k, v in zip(keys, vals[nk:])}
    for a, h in enumerate(hs):
        sides = sorted(set(s for aa, s in keys if aa == a))
        h.last_counts = tuple(abs(sent.get((a, s), 0)) for s in (0, 1)) + tuple(abs(recv.get((a, s), 0)) for s in (0, 1))
        for s in sides:
            h.cap_send[s] = _next_capacity(h.cap_send.get(s), abs(sent[(a, s)]))
            h.cap_recv[s] = _next_capacity(h.cap_recv.get(s), abs(recv[(a, s)]))
    bad = [k for k in keys if sent[k] < 0 or recv[k] < 0]
    if bad or flag & 1:
        raise RuntimeError('padded ghost exchange: a face outgrew its message capacity in ONE exchange %r (counts %r / %r): '
                           'the ghosts of the last step were incomplete.  SPH_HALO_PROTOCOL=capacity repeats such faces '
                           'at the price of a device->host round trip per exchange' % (bad, sent, recv))
    if flag & 2:
        raise RuntimeError('padded ghost exchange: a ghost arrived with a smoothing length or mass other than the one promised '
                           'for its array (h or m was written on some rank after the promise was made): the last step '
                           'ran on wrong record layouts')


def _exchange_padded(hs, nbrs):
    """Ghost refresh with NO device->host round trip ('padded' protocol, steady state):
    fixed-capacity messages packed on the device as in 'capacity', but the receiver
    appends ALL rows of a message -- the ghosts and, behind them, padding rows parked
    far outside the domain that are inert on the whole path -- so nothing has to be counted before the
    neighbour update and the evaluation are queued.  Counts and flags follow one
    exchange later (_padded_collect)."""
    h0 = hs[0]
    dist, ops0, na = h0.dist, h0.ops, len(hs)
    sides = [s for s, _, _ in nbrs]
    _padded_collect(hs)
    for h in hs:
        h.ops.drop_ghosts()
    shift_of = {s: shift for s, _, shift in nbrs}
    send_order = sorted(nbrs, key=lambda nb: -nb[0])
    recv_order = sorted(nbrs, key=lambda nb: nb[0])
    out, inb = [], []
    for a, h in enumerate(hs):
        npr, oa, ia = h.ops.nprops, {}, {}
        for s in sides:
            oa[s] = h.ops.message_buffer(('send', s), h.cap_send[s] * npr + 1)
            ia[s] = h.ops.message_buffer(('recv', s), h.cap_recv[s] * npr + 1)
        h.ops.select_pack(h.lo + h.width, h.hi - h.width,
                          [shift_of.get(0, 0.0), shift_of.get(1, 0.0)],
                          [h.cap_send.get(0, 0), h.cap_send.get(1, 0)],
                          [oa.get(0), oa.get(1)])
        out.append(oa)
        inb.append(ia)
    for h in hs:
        f = getattr(h.ops, 'before_comm', None)
        if f is not None:
            f()
    reqs = [dist.P2POp(dist.isend, out[a][s], peer) for s, peer, _ in send_order for a in range(na)]
    reqs += [dist.P2POp(dist.irecv, inb[a][s], peer) for s, peer, _ in recv_order for a in range(na)]
    if reqs:
        for w in dist.batch_isend_irecv(reqs):
            w.wait()                  # stream-level for device transports: nothing here blocks the host
    for h in hs:
        f = getattr(h.ops, 'after_comm', None)
        if f is not None:
            f()
    keys = [(a, s) for a in range(na) for s in sides]
    for a, h in enumerate(hs):
        for s, _, _ in nbrs:          # lo side first: deterministic
            h.ops.append_padded(inb[a][s], h.cap_recv[s], h.h_promise, h.m_promise)
        h.padded_exchanges += 1
    h0.padded_pending = (ops0.queue_headers([out[a][s] for a, s in keys] + [inb[a][s] for a, s in keys]), keys)


class SlabDecomposition(object):
    """All particle arrays of one rank: what ``ParallelManager.update()``
    (parallel_manager.pyx:512-530) does before an acceleration evaluation --
    migrate owned particles that crossed a slab face, then refresh the ghosts
    -- plus ``rebalance()`` (the reference's Zoltan load balance,
    parallel_manager.pyx:577-640, reduced to moving the slab faces)."""

    def __init__(self, arrays, ctx, rank, world, axis, width, lo, hi,
                 props=WCSPH_HALO_PROPS, periodic=False, period=0.0,
                 ops_factory=None, dist=None, protocol=None):
        if dist is None:
            import torch.distributed as dist
        self.dist = dist
        self.rank, self.world, self.axis = rank, world, axis
        self.halos = []
        for pa in arrays:
            p = props[pa.name] if isinstance(props, dict) else props
            ops = ops_factory(pa, axis, p) if ops_factory else None
            self.halos.append(SlabHalo(pa, ctx, rank, world, axis, width, lo,
                                       hi, props=p, periodic=periodic,
                                       period=period, ops=ops, dist=dist, protocol=protocol))

    @property
    def lo(self):
        return self.halos[0].lo

    @property
    def hi(self):
        return self.halos[0].hi

    def migrate(self):
        return sum(h.migrate() for h in self.halos)

    def exchange(self, drop=True):
        """Ghost refresh of ALL arrays with one batch of point-to-point transfers
        (a dam break has three arrays): see exchange_halos."""
        exchange_halos(self.halos, drop=drop)

    def faces(self):
        """(lo, hi) along the slab axis outside which every ghost of this rank
        lies; -inf / +inf for a face without a neighbour"""
        h = self.halos[0]
        sides = [s for s, _, _ in h.neighbours()]
        return (h.lo if 0 in sides else -float('inf'), h.hi if 1 in sides else float('inf'))

    def exchange_begin(self, drop=True):
        """post the ghost transfers and return (exchange_halos_begin); the ghosts
        are there after `exchange_finish(state)`"""
        return exchange_halos_begin(self.halos, drop=drop)

    def exchange_finish(self, state):
        exchange_halos_finish(state)

    def update(self):
        self.migrate()
        self.exchange()

    def rebalance(self, nbins=4096, weights=None):
        """Move the slab faces so that every rank owns about the same number
        of real particles (optionally weighted per array, e.g. fluid particles
        cost more than boundary ones): global histogram of the slab-axis
        coordinate (all_reduce SUM), faces at its quantiles, then migrate until
        every particle is home (a particle moves one slab per round)."""
        import numpy as np
        dist = self.dist
        ops0 = self.halos[0].ops
        coords = [h.ops.coords() for h in self.halos]
        lmin = min([c.min() for c in coords if c.size] or [float('inf')])
        lmax = max([c.max() for c in coords if c.size] or [-float('inf')])
        gmin = -allreduce_scalars([-lmin], 'max', dist=dist,
                                  device=getattr(ops0, 'device', None))[0]
        gmax = allreduce_scalars([lmax], 'max', dist=dist,
                                 device=getattr(ops0, 'device', None))[0]
        span = max(gmax - gmin, 1e-300)
        hist = np.zeros(nbins)
        for k, c in enumerate(coords):
            w = 1.0 if weights is None else float(weights[k])
            if c.size:
                b = np.minimum(((c - gmin) / span * nbins).astype(np.int64),
                               nbins - 1)
                hist += w * np.bincount(b, minlength=nbins)
        import torch
        t = torch.tensor(hist, dtype=torch.float64,
                         device=getattr(ops0, 'device', None))
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        cum = np.concatenate([[0.0], np.cumsum(t.cpu().numpy())])
        edges = gmin + span * np.arange(nbins + 1) / nbins
        targets = cum[-1] * np.arange(1, self.world) / self.world
        cuts = np.interp(targets, cum, edges)
        faces = np.concatenate([[-np.inf if not self.halos[0].periodic else gmin],
                                cuts, [np.inf if not self.halos[0].periodic else gmax]])
        for h in self.halos:
            # the outer faces keep their meaning (open, or the periodic box)
            if 0 < self.rank:
                h.lo = float(faces[self.rank])
            if self.rank < self.world - 1:
                h.hi = float(faces[self.rank + 1])
        rounds = 0
        while True:
            moved = self.migrate()
            total = allreduce_scalars([float(moved)], 'max', dist=dist,
                                      device=getattr(ops0, 'device', None))[0]
            rounds += 1
            if total == 0 or rounds > self.world:
                break
        # the faces moved: the ghost counts of the next exchange have nothing to do with the last ones -- every rank
        # (rebalance is collective) sizes its messages anew with a counted exchange
        for h in self.halos:
            h.cap_send.clear()
            h.cap_recv.clear()
        return faces, rounds


class HipParallelManager(object):
    """The object ``Integrator.set_parallel_manager`` expects
    (integrator.py:274-286 calls ``pm.update()`` before every ``nnps.update()``;
    solver.py:476-480 and parallel_manager.pyx:463 reduce the adaptive
    time-step inputs over all ranks): a ``SlabDecomposition`` plus the scalar
    all-reduces, with an optional re-balance every ``rebalance_every`` updates
    (the reference's ``lb_freq``)."""

    def __init__(self, decomposition, rebalance_every=0, weights=None):
        self.dec = decomposition
        self.dist = decomposition.dist
        self.rebalance_every = int(rebalance_every)
        self.weights = weights
        self.count = 0
        self._device = getattr(decomposition.halos[0].ops, 'device', None)

    def update(self):
        self.count += 1
        if self.rebalance_every and self.count % self.rebalance_every == 0:
            self.dec.rebalance(weights=self.weights)
            self.dec.exchange()
        else:
            self.dec.update()

    def reduce_max(self, values):
        return allreduce_scalars(values, 'max', dist=self.dist, device=self._device)

    def reduce_min(self, values):
        return allreduce_scalars(values, 'min', dist=self.dist, device=self._device)


def allreduce_scalars(values, op, dist=None, device=None):
    """MIN/MAX of a few doubles over all ranks (dt, dt_cfl, dt_force, bounds,
    hmax): parallel_manager.pyx:463 ``update_time_steps`` and :937-945
    ``_compute_bounds``."""
    import torch
    if dist is None:
        import torch.distributed as dist
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    dist.all_reduce(t, op={'min': dist.ReduceOp.MIN,
                           'max': dist.ReduceOp.MAX}[op])
    return [float(v) for v in t.cpu()]


def slab_bounds(coord, world, weights=None):
    """Equal-particle-count slab faces along one axis (the
    dam-break fluid fills 38 % of the tank, geometric slabs would idle):
    returns world+1 ascending cut positions from the quantiles of `coord`.
    `weights` (one per particle): equal WORK instead of equal count -- a fluid
    particle of a dam break costs several boundary particles
    (`SlabDecomposition.rebalance(weights=...)` is the run-time counterpart)."""
    import numpy as np
    coord = np.asarray(coord)
    if weights is None:
        q = np.quantile(coord, np.linspace(0.0, 1.0, world + 1))
    else:
        order = np.argsort(coord, kind='stable')
        cw = np.cumsum(np.asarray(weights, dtype=np.float64)[order])
        targets = cw[-1] * np.linspace(0.0, 1.0, world + 1)
        idx = np.minimum(np.searchsorted(cw, targets, side='left'), coord.size - 1)
        q = coord[order][idx].astype(np.float64)
    q[0], q[-1] = -np.inf, np.inf
    return q
