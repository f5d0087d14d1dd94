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

    def empty_ints(self, n):
        return self.torch.empty(int(n), dtype=self.torch.int64, device=self.device)

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
        n0 = self.gpu.get_number_of_particles()
        dev._check(self.lib.sph_halo_append_strided(
            self.ctx._h, self.id, self.nprops, self.props,
            C.c_void_p(buf.data_ptr()), count, count if stride is None else stride))
        for p, v in self.__dict__.get('fill', ()):      # promised h / m did not travel
            dev._check(self.lib.sph_array_fill(self.ctx._h, self.id, p, v, n0, count))

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
    def append_padded(self, buf, cap, h_promise, m_promise, flags=None, slot=0):
        """all `cap` rows of a fixed-capacity message behind the particles: the
        first |header| are the ghosts, the rest padding rows parked at 1e18 (inert on
        the whole path); the host learns the count after the evaluation was queued
        (sph_halo_append_padded).  `flags` / `slot`: the flag words of the exchange
        (ONE tensor for all its arrays, `flag_words` of the first) and this array's
        word in it.  A promised h / m that is not among the message's properties is
        written into the rows by the library (it did not travel)."""
        flags = self.flag_words(slot + 1) if flags is None else flags
        dev._check(self.lib.sph_halo_append_padded(
            self.ctx._h, self.id, self.nprops, self.props, C.c_void_p(buf.data_ptr()), int(cap),
            float(h_promise), float(m_promise), C.c_void_p(flags.data_ptr() + 4 * int(slot))))

    def append_padded_faces(self, bufs, caps, h_promise, m_promise, flags, slot):
        """the messages of all faces of this array (lo face first) in ONE launch (sph_halo_append_padded2)"""
        bufs, caps = list(bufs), [int(c) for c in caps]
        if len(bufs) > 2:
            raise ValueError('an array has at most two faces')
        while len(bufs) < 2:
            bufs.append(None)
            caps.append(0)
        dev._check(self.lib.sph_halo_append_padded2(
            self.ctx._h, self.id, self.nprops, self.props,
            C.c_void_p(bufs[0].data_ptr()) if bufs[0] is not None else None, caps[0],
            C.c_void_p(bufs[1].data_ptr()) if bufs[1] is not None else None, caps[1],
            float(h_promise), float(m_promise), C.c_void_p(flags.data_ptr() + 4 * int(slot))))

    def flag_words(self, n=1):
        """device words sph_halo_append_padded ORs into, one per array of the exchange
        (bit 0: incomplete message, bit 1: a ghost broke the promise of its array)"""
        t = self.__dict__.get('_flag')
        if t is None or t.numel() < n:
            t = self._flag = self.torch.zeros(max(n, 8), dtype=self.torch.int32, device=self.device)
        return t

    def queue_headers(self, tensors, nflags=1):
        """the header of each message and the `nflags` flag words on their way to pinned
        memory behind an event: `collect_headers` waits for that event only (it was recorded
        right behind the transfers, so a host that has queued the evaluation since does not
        drain the stream)"""
        torch = self.torch
        n = len(tensors)
        pin = self.__dict__.get('_hdr_pin2')
        if pin is None:
            pin = self._hdr_pin2 = torch.empty(64, dtype=torch.float64).pin_memory()
        if n <= 48 and n + nflags <= 64 and self._shares_torch_stream():
            # ONE launch writes the headers and the flag words into the pinned buffer (sph_queue_values)
            ptrs = (C.c_void_p * n)(*[t.data_ptr() + (t.numel() - 1) * 8 for t in tensors])
            dev._check(self.lib.sph_queue_values(self.ctx._h, n, ptrs, nflags,
                                                 C.c_void_p(self.flag_words(nflags).data_ptr()),
                                                 C.c_void_p(pin.data_ptr())))
        else:
            if pin.numel() < n + nflags:
                pin = self._hdr_pin2 = torch.empty(n + nflags, dtype=torch.float64).pin_memory()
            vals = torch.cat([torch.stack([t[-1] for t in tensors]),
                              self.flag_words(nflags)[:nflags].to(torch.float64)])
            pin[:n + nflags].copy_(vals, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        return (ev, n, nflags)

    def collect_headers(self, handle):
        ev, n, nflags = handle
        ev.synchronize()
        vals = self._hdr_pin2[:n + nflags].tolist()
        return vals[:n], [int(v) for v in vals[n:]]

    def clear_flags(self):
        if self.__dict__.get('_flag') is not None:
            self._flag.zero_()

    def mark_hm_written(self):
        """what the neighbour update knows of this array's h and m is void (a ghost
        arrived with other values than promised): its next update looks again"""
        for p in ('h', 'm'):
            dev._check(self.lib.sph_array_mark_written(self.ctx._h, self.id, dev.prop_id(p)))

    # -- promised-uniform h / m do not travel ------------------------------------
    def set_promise(self, h_promise, m_promise, send=True):
        """the properties of this array's messages from now on: without h / m where the
        ranks hold ONE value of it (`send` False keeps them in the messages: the receiver
        then checks every ghost against the promise instead of the sender every row it
        packs).  Rows appended through `append` get the promised values written in."""
        full = self.__dict__.setdefault('_full_props', list(self.props))
        keep = [p for p in full if send or not (
            (p == dev.prop_id('h') and h_promise == h_promise) or
            (p == dev.prop_id('m') and m_promise == m_promise))]
        self.nprops = len(keep)
        self.props = (C.c_int * self.nprops)(*keep)
        self._messages = {}
        self.fill = [(dev.prop_id('h'), float(h_promise)), (dev.prop_id('m'), float(m_promise))]
        self.fill = [(p, v) for p, v in self.fill if v == v and p in full and p not in keep]
        self.promise = (float(h_promise), float(m_promise))

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
        rows laid out [nprops][cap], the row count (negative: more than cap; + 0.5: a
        packed row carries another h / m than promised) in its last element
        (sph_halo_select_pack_promised)."""
        sh = (C.c_double * 2)(float(shifts[0]), float(shifts[1]))
        cp = (C.c_size_t * 2)(int(caps[0]), int(caps[1]))
        ds = (C.c_void_p * 2)(*[b.data_ptr() if b is not None else None for b in bufs])
        nan = float('nan')
        hp, mp = self.__dict__.get('promise', (nan, nan))
        fill = dict(self.__dict__.get('fill', ()))
        dev._check(self.lib.sph_halo_select_pack_promised(
            self.ctx._h, self.id, self.axis, float(lo_cut), float(hi_cut), 0,
            self.nprops, self.props, sh, cp, ds,
            hp if dev.prop_id('h') in fill else nan, mp if dev.prop_id('m') in fill else nan))

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

    def axis_row(self):
        """row of the slab-axis coordinate in a `pack_all` buffer"""
        return list(self.all_props()).index(dev.prop_id('xyz'[self.axis]))

    def remove_selected(self):
        left = C.c_size_t(0)
        dev._check(self.lib.sph_halo_remove_selected(self.ctx._h, self.id,
                                                     C.byref(left)))
        return int(left.value)

    def append_real(self, buf, count, flags=None, slot=0):
        """migrants become real particles of this array.  With the flag words of the
        exchange (`flags`, `slot`) and a promise in place (`set_promise`) the library
        keeps what it knows of h and m and checks the arrivals against the promise on the
        device (sph_halo_append_promised): the neighbour update after the migration makes
        no device->host round trip."""
        ids = self.all_props()
        nan = float('nan')
        hp, mp = self.__dict__.get('promise', (nan, nan))
        if flags is not None and (hp == hp or mp == mp):
            dev._check(self.lib.sph_halo_append_promised(
                self.ctx._h, self.id, len(ids), (C.c_int * len(ids))(*ids), C.c_void_p(buf.data_ptr()), count,
                hp, mp, C.c_void_p(flags.data_ptr() + 4 * int(slot))))
        else:
            dev._check(self.lib.sph_halo_append(
                self.ctx._h, self.id, len(ids), (C.c_int * len(ids))(*ids),
                C.c_void_p(buf.data_ptr()), count))
        n = self.gpu.get_number_of_particles()
        dev._check(self.lib.sph_array_resize(self.ctx._h, self.id, n, n))

    def coord_range(self):
        """(min, max) of the real particles' coordinate along the slab axis (device reductions)"""
        if self.n_real() == 0:
            return float('inf'), -float('inf')
        lo, hi = C.c_double(), C.c_double()
        p = dev.prop_id('xyz'[self.axis])
        dev._check(self.lib.sph_reduce_min(self.ctx._h, self.id, p, C.byref(lo)))
        dev._check(self.lib.sph_reduce_max(self.ctx._h, self.id, p, C.byref(hi)))
        return lo.value, hi.value

    def histogram(self, vmin, span, nbins):
        """counts of the real particles per bin of the slab-axis coordinate (sph_coord_histogram: the particles stay on
        the device, nbins words come back)"""
        import numpy as np
        out = np.zeros(nbins, dtype=np.uint32)
        dev._check(self.lib.sph_coord_histogram(self.ctx._h, self.id, self.axis, float(vmin), float(span), int(nbins),
                                                out.ctypes.data_as(C.POINTER(C.c_uint32))))
        return out.astype(np.float64)

    def coords(self):
        """host copy of the real particles' coordinate along the slab axis
        (test doubles and diagnostics; re-balancing uses `histogram`)"""
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
                 ops=None, dist=None, protocol=None, promise=True, send_promised=False):
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
        self.total_migrated = 0           # particles this rank has handed over since the start
        self.track_excursion = False      # migrate() measures how far its leavers had strayed (one more readback)
        self.max_excursion = 0.0
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
        self.promise = promise            # False: no promises are made (h and m travel, every neighbour update looks at them)
        self.send_promised = send_promised  # True: promised h / m travel all the same (the receiver checks them)
        self.promised = not promise       # the promises stand (or none are wanted)
        self.padded_ok = None             # can every rank run the round-trip-free exchange?  (agreed collectively with the
                                          # promises; without promises each rank decides for itself: the two protocols send
                                          # the same messages)
        self.padded_pending = None
        self.padded_exchanges = 0
        self.repaired_exchanges = 0       # exchanges `verify_halos` found incomplete and repeated the counted way
        self.recv_rows = {}

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
        self.max_excursion = 0.0
        for s, _, shift in nbrs:
            out_buf[s], nprops = ops.pack_all(s, send_cnt[s], shift)
            if send_cnt[s] and self.track_excursion and hasattr(ops, 'axis_row'):
                # how far beyond its face the farthest leaver had travelled (lazy migration: HipParallelManager)
                k, cnt = ops.axis_row(), send_cnt[s]
                row = out_buf[s][k * cnt:(k + 1) * cnt] - shift
                face = self.lo if s == 0 else self.hi
                far = float((face - row).max()) if s == 0 else float((row - face).max())
                self.max_excursion = max(self.max_excursion, far)
        ops.remove_selected()
        in_buf, recv_cnt = self._swap(nbrs, send_cnt, out_buf, nprops)
        for s, _, _ in nbrs:
            if recv_cnt[s]:
                ops.append_real(in_buf[s], recv_cnt[s])
        self.last_migrated = (n_lo, n_hi, recv_cnt.get(0, 0),
                              recv_cnt.get(1, 0))
        self.total_migrated += n_lo + n_hi
        return n_lo + n_hi


def migrate_halos(hs):
    """`SlabHalo.migrate` for all arrays of a rank together (a dam break has three):
    the leavers of every array are selected, packed and removed, then ONE tiny
    all_gather carries every array's counts and ONE batch of point-to-point transfers
    every array's rows -- per array that was a host round trip and a transfer group
    each (parallel_manager.pyx:1085-1157 sends all arrays' exported particles in one
    exchange as well).  Returns the number of particles that left this rank."""
    h0 = hs[0]
    dist = h0.dist
    for h in hs:
        h.ops.drop_ghosts()
    nbrs = h0.neighbours()
    if not nbrs:
        return 0
    sides = [s for s, _, _ in nbrs]
    inf = float('inf')
    send_cnt, out_buf, nprops = [], [], []
    fars = []
    for h in hs:
        ops = h.ops
        n_lo, n_hi = ops.select(h.lo if 0 in sides else -inf, h.hi if 1 in sides else inf)
        cnt = {0: n_lo, 1: n_hi}
        bufs, npr = {}, None
        h.max_excursion = 0.0
        for s, _, shift in nbrs:
            bufs[s], npr = ops.pack_all(s, cnt[s], shift)
            if cnt[s] and h.track_excursion and hasattr(ops, 'axis_row'):
                # how far beyond its face the farthest leaver had travelled (lazy migration: HipParallelManager); the
                # maxima stay where they were computed until all arrays are through: ONE read for all of them
                k = ops.axis_row()
                row = bufs[s][k * cnt[s]:(k + 1) * cnt[s]] - shift
                face = h.lo if s == 0 else h.hi
                fars.append((h, (face - row).max() if s == 0 else (row - face).max()))
        ops.remove_selected()
        send_cnt.append(cnt)
        out_buf.append(bufs)
        nprops.append(npr)
    ops0 = h0.ops
    if fars:
        vals = [f for _, f in fars]
        if hasattr(vals[0], 'device'):              # torch scalars (on the device: one transfer for all of them)
            import torch
            vals = torch.stack(vals).cpu().tolist()
        for (h, _), v in zip(fars, vals):
            h.max_excursion = max(h.max_excursion, float(v))
    sync = getattr(ops0, 'before_comm', None)
    if sync is not None:
        sync()
    na, world = len(hs), h0.world
    mine = ops0.int_tensor([c[s] for c in send_cnt for s in (0, 1)])
    empty = getattr(ops0, 'empty_ints', None)        # (no host->device copy for a buffer that is overwritten)
    allc = empty(2 * na * world) if empty is not None else ops0.int_tensor([0] * (2 * na * world))
    dist.all_gather_into_tensor(allc, mine)
    allc = [int(v) for v in allc.cpu()]
    # what a peer sends to me: its hi list if it is my lo neighbour, else its lo list
    recv_cnt = [{s: allc[2 * na * peer + 2 * a + (1 - s)] for s, peer, _ in nbrs} for a in range(na)]
    in_buf = [{s: hs[a].ops.new_buffer(recv_cnt[a][s], nprops[a]) for s, _, _ in nbrs} for a in range(na)]
    # Between one pair of ranks messages match in posting order: sends hi-face first, receives lo-face first (a periodic
    # axis with <= 2 ranks has both faces talk to the SAME peer), the arrays in their order inside each face.
    reqs = []
    for s, peer, _ in sorted(nbrs, key=lambda nb: -nb[0]):
        for a in range(na):
            if send_cnt[a][s]:
                reqs.append(dist.P2POp(dist.isend, out_buf[a][s], peer))
    for s, peer, _ in sorted(nbrs, key=lambda nb: nb[0]):
        for a in range(na):
            if recv_cnt[a][s]:
                reqs.append(dist.P2POp(dist.irecv, in_buf[a][s], peer))
    if reqs:
        for w in dist.batch_isend_irecv(reqs):
            w.wait()
    sync = getattr(ops0, 'after_comm', None)
    if sync is not None:
        sync()
    left = 0
    # the flag words the padded exchange reads with its headers: an arrival that breaks its array's promise shows there
    flags = _flag_words(hs) if (h0.promise and getattr(h0, 'padded_ok', False)) else None
    code = getattr(h0.ops.append_real, '__code__', None)
    if flags is not None and (code is None or 'flags' not in code.co_varnames):
        flags = None        # (primitives without the promised append: the plain one)
    for a, h in enumerate(hs):
        for s, _, _ in nbrs:
            if recv_cnt[a][s]:
                if flags is not None:
                    h.ops.append_real(in_buf[a][s], recv_cnt[a][s], flags, a)
                else:
                    h.ops.append_real(in_buf[a][s], recv_cnt[a][s])
        h.last_migrated = (send_cnt[a][0], send_cnt[a][1], recv_cnt[a].get(0, 0), recv_cnt[a].get(1, 0))
        h.total_migrated += send_cnt[a][0] + send_cnt[a][1]
        left += send_cnt[a][0] + send_cnt[a][1]
    return left


def _capacity(count):
    """rows a fixed-size ghost message is sized for, from the count both ends of
    the face saw last: a quarter of headroom + 4096 rows, in units of 1024"""
    return ((count + count // 4 + 4096 + 1023) // 1024) * 1024


def _capacity_tight(count):
    """... once the count has been steady: a sixteenth of headroom + 1024 rows (a face
    that outgrows it anyway is repeated by the counted exchange: `verify_halos`)"""
    return ((count + count // 16 + 1024 + 1023) // 1024) * 1024


STEADY_EXCHANGES = 8     # exchanges a face's count must stay within 1/64 of itself before its messages shrink


def _next_capacity(cap, count, hist=None):
    """the SAME rule on both ends of a face (sender: the count it sent;
    receiver: the count in the header), so the two stay in step without talking.
    `hist` (a list [previous count, steady exchanges, tight?], updated in place;
    both ends keep one per face and direction and feed it the same counts) lets
    the capacity ADAPT: after STEADY_EXCHANGES exchanges whose count moved by less
    than 1/64 the message shrinks to `_capacity_tight`; a count that then comes
    within 1/32 + 512 rows of that capacity returns it to the generous rule."""
    tight = False
    if hist is not None:
        prev, steady, tight = hist
        steady = steady + 1 if (prev is not None and abs(count - prev) <= max(prev // 64, 16)) else 0
        if tight and (count > cap or count + count // 32 + 512 > cap):
            tight, steady = False, 0
        elif not tight and steady >= STEADY_EXCHANGES and cap is not None and count <= cap:
            tight = True
        hist[:] = [count, steady, tight]
    if tight:
        want = _capacity_tight(count)
        # (never grow into the tight size: a tight face only ever shrinks or stays)
        return want if cap is None or want < cap or count > cap else cap
    if cap is None or count > cap or count + count // 8 + 1024 > cap or 4 * count + 16384 < cap:
        return _capacity(count)
    return cap


def _track_capacities(h, s, sent, received):
    """capacities of face `s` of halo `h` after an exchange that sent / received
    these many rows (None: that direction did not run)"""
    hist = h.__dict__.setdefault('cap_hist', {})
    if sent is not None:
        h.cap_send[s] = _next_capacity(h.cap_send.get(s), sent, hist.setdefault(('send', s), [None, 0, False]))
    if received is not None:
        h.cap_recv[s] = _next_capacity(h.cap_recv.get(s), received, hist.setdefault(('recv', s), [None, 0, False]))


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
    if hs and hs[0].protocol == 'padded' and hs[0].promise and hs[0].neighbours() and not all(h.promised for h in hs):
        _padded_collect(_active(hs))
        _establish_promises(hs)   # collective, over ALL arrays; the exchange after it goes the counted way
    hs = _active(hs)              # (arrays that hold no particle on ANY rank take no part)
    if not hs:
        return
    h0 = hs[0]
    dist, ops0, world, na = h0.dist, h0.ops, h0.world, len(hs)
    nbrs = h0.neighbours()
    sides = [s for s, _, _ in nbrs]
    if len(nbrs) == 2 and world > 2 and h0.hi - h0.lo < h0.width:
        # the ghost layer of a neighbour would have to reach THROUGH this slab into the next one: ghosts come from the
        # two adjacent slabs only (the reference's Zoltan exchange has no such limit, parallel_manager.pyx:1159-1243)
        raise RuntimeError('slab %d is %.6g wide, thinner than the ghost layer (%.6g): use fewer ranks or re-balance'
                           % (h0.rank, h0.hi - h0.lo, h0.width))
    if nbrs and drop and not overlap and h0.protocol == 'padded' and all(
            h.promised and _padded_eligible(h) and
            all(h.cap_send.get(s) is not None and h.cap_recv.get(s) is not None for s in sides) for h in hs):
        _exchange_padded(hs, nbrs)
        return
    _padded_collect(hs)           # (a padded exchange may be outstanding: its counts size this one)
    for h in hs:
        if drop:
            h.ops.drop_ghosts()
    if not nbrs:
        return
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
        broken = sorted(set(k[0] for k, v in zip(keys + keys, hdr) if v != int(v)))
        if broken:                      # a packed row carries another h / m than the one promised (and not sent)
            raise RuntimeError(PROMISE_ERROR % [getattr(getattr(hs[a].ops, 'pa', None), 'name', a) for a in broken])
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
            _track_capacities(h, s, send[a][s], recv[a][s])
        h.last_counts = (send[a][0], send[a][1], recv[a].get(0, 0), recv[a].get(1, 0))


def _establish_promises(hs):
    """'padded' protocol, once (and again after every `rebalance`): does every rank
    hold ONE smoothing length and ONE mass per array?  (min/max over the ranks of
    each array's own range.)  Then every ghost is promised to carry them: the
    neighbour update keeps what it knows of h and m across the appends (no look, no
    round trip), and -- `send_promised=False`, the default -- the promised
    properties do not TRAVEL: the messages shrink from 9 to 7 properties for a
    WCSPH ghost, the receiver writes the promised values into the rows, the SENDER
    checks every row it packs against the promise (header + 0.5 otherwise:
    `verify_halos`).  Also agreed here, collectively: whether EVERY rank can run the
    round-trip-free exchange (kernels and transport on one stream); one that cannot
    makes all of them take the counted one."""
    h0 = hs[0]
    lo, hi = [], []
    for h in hs:
        r = h.ops.hm_range() if hasattr(h.ops, 'hm_range') else [float('inf'), -float('inf')] * 2
        lo += [r[0], r[2]]
        hi += [r[1], r[3]]
    can = all(hasattr(h.ops, 'append_padded') and hasattr(h.ops, 'select_pack') and
              getattr(h.ops, '_shares_torch_stream', lambda: True)() for h in hs)
    lo.append(1.0 if can else 0.0)
    dev_t = getattr(h0.ops, 'device', None)
    glo = allreduce_scalars(lo, 'min', dist=h0.dist, device=dev_t)
    ghi = allreduce_scalars(hi, 'max', dist=h0.dist, device=dev_t)
    for a, h in enumerate(hs):
        h.h_promise = glo[2 * a] if glo[2 * a] == ghi[2 * a] else float('nan')
        h.m_promise = glo[2 * a + 1] if glo[2 * a + 1] == ghi[2 * a + 1] else float('nan')
        h.promised = True
        # an array without a single particle on ANY rank (a dam break without its obstacle; min = +inf, max = -inf) takes no
        # part in the exchanges until the promises are renewed: its empty messages would leave nothing but padding rows
        # behind, an array "with rows but no mass" that keeps the rank off the one-launch merged evaluation and makes every
        # neighbour update look at h and m
        h.skip = hasattr(h.ops, 'hm_range') and glo[2 * a] > ghi[2 * a] and glo[2 * a + 1] > ghi[2 * a + 1]
        h.padded_ok = glo[-1] > 0.5
        if hasattr(h.ops, 'set_promise'):
            h.ops.set_promise(h.h_promise, h.m_promise, send=h.send_promised or not h.padded_ok)
        if hasattr(h.ops, 'clear_flags'):
            h.ops.clear_flags()       # (what an earlier promise's check left behind)
        # message sizes may have changed with the property list
        h.cap_send.clear()
        h.cap_recv.clear()
        h.__dict__.pop('cap_hist', None)


def _padded_eligible(h):
    if h.padded_ok is None:
        h.padded_ok = bool(hasattr(h.ops, 'append_padded') and hasattr(h.ops, 'select_pack') and
                           getattr(h.ops, '_shares_torch_stream', lambda: True)())
    return h.padded_ok


def _active(hs):
    return [h for h in hs if not getattr(h, 'skip', False)]


class GhostsIncomplete(RuntimeError):
    """a 'padded' ghost exchange was found incomplete too late to repair"""


def _padded_headers(hs):
    """the counts and flags of the outstanding 'padded' exchange (None: there is
    none): waits for the event recorded right behind its transfers.  Returns
    (keys, sent, recv, flags, out, inb) with the raw header values."""
    h0 = hs[0]
    pend = h0.padded_pending
    if pend is None:
        return None
    h0.padded_pending = None
    handle, keys, out, inb = pend
    vals, flags = h0.ops.collect_headers(handle)
    nk = len(keys)
    sent = dict(zip(keys, vals[:nk]))
    recv = dict(zip(keys, vals[nk:]))
    for a, h in enumerate(hs):
        sides = sorted(set(s for aa, s in keys if aa == a))
        h.last_counts = tuple(int(abs(sent.get((a, s), 0))) for s in (0, 1)) + \
            tuple(int(abs(recv.get((a, s), 0))) for s in (0, 1))
        for s in sides:
            _track_capacities(h, s, int(abs(sent[(a, s)])), int(abs(recv[(a, s)])))
    return keys, sent, recv, flags, out, inb


def _broken_promise(hs, sent, recv, flags):
    """arrays of which a row travelled (or was packed) with another h / m than promised"""
    bad = set(a for (a, s), v in list(sent.items()) + list(recv.items()) if v != int(v))
    bad |= set(a for a, f in enumerate(flags) if f & 2)
    return sorted(bad)


PROMISE_ERROR = ('padded ghost exchange: array %r carries a smoothing length or mass other than the ONE value '
                 'promised for it when the exchange was set up (h or m was written afterwards).  Write h / m before '
                 'the first exchange, call SlabDecomposition.renew_promises() on every rank after writing them, or '
                 'construct the decomposition with promise=False')


def _padded_collect(hs):
    """what the last 'padded' exchange left to collect, when nobody asked
    `verify_halos` before the next exchange starts: capacities follow the counts;
    an incomplete message or a broken promise is an error NOW (the evaluation
    that used those ghosts has been consumed).  Callers that verify after queueing
    the evaluation (Integrator.compute_accelerations, bench.py) never get here
    with anything wrong."""
    hs = _active(hs)
    got = _padded_headers(hs) if hs else None
    if got is None:
        return
    keys, sent, recv, flags, _, _ = got
    bad = [k for k in keys if sent[k] < 0 or recv[k] < 0]
    if bad or any(f & 1 for f in flags):
        raise GhostsIncomplete(
            'padded ghost exchange: a face outgrew its message capacity in ONE exchange %r (counts %r / %r) and '
            'nobody verified the ghosts before the next exchange: the last evaluation ran on incomplete ghosts.  '
            'Call SlabDecomposition.verify() after queueing the evaluation (it repeats such a face with a counted '
            'exchange and says so; Integrator.compute_accelerations does), or SPH_HALO_PROTOCOL=capacity' % (bad, sent, recv))
    broken = _broken_promise(hs, sent, recv, flags)
    if broken:
        raise RuntimeError(PROMISE_ERROR % [getattr(getattr(hs[a].ops, 'pa', None), 'name', a) for a in broken])


def verify_halos(hs):
    """Were the ghosts of the last exchange complete?  To be called AFTER the
    neighbour update and the evaluation that use them were queued and BEFORE
    anything consumes the evaluation's results (Integrator.compute_accelerations
    does; the reference never evaluates on incomplete ghosts,
    parallel_manager.pyx:1085-1157, it counts first).  The 'padded' protocol appends
    fixed-capacity messages without knowing their counts; the counts were queued
    to pinned memory right behind the transfers, so reading them here waits for
    the TRANSFERS only -- the host stays one evaluation ahead of the device
    instead of two, it does not drain the stream.

    True: nothing to repair.  False: a face had outgrown its capacity; the two
    ranks of that face (both read the negative header, nobody else is involved)
    have repeated it with exactly sized messages -- point-to-point, the counted
    way -- and rebuilt the ghost rows of the arrays concerned: the caller repeats
    `nnps.update()` and the evaluation (both only read the real particles' state
    and overwrite their own results).  A ghost that broke the promise of its
    array's ONE h / m raises: the sender's properties have to be looked at again
    collectively (`SlabDecomposition.renew_promises`)."""
    hs = _active(hs)
    got = _padded_headers(hs) if hs else None
    if got is None:
        return True
    keys, sent, recv, flags, out, inb = got
    h0 = hs[0]
    broken = _broken_promise(hs, sent, recv, flags)
    if broken:
        raise RuntimeError(PROMISE_ERROR % [getattr(getattr(hs[a].ops, 'pa', None), 'name', a) for a in broken])
    over_send = [k for k in keys if sent[k] < 0]
    over_recv = [k for k in keys if recv[k] < 0]
    if not over_send and not over_recv:
        return True
    dist = h0.dist
    nbrs = h0.neighbours()
    shift_of = {s: shift for s, _, shift in nbrs}
    send_order = sorted(nbrs, key=lambda nb: -nb[0])
    recv_order = sorted(nbrs, key=lambda nb: nb[0])
    na = len(hs)
    xo, xi = {}, {}
    for a in sorted(set(a for a, _ in over_send)):
        h = hs[a]
        h.ops.select(h.lo + h.width, h.hi - h.width)       # the index lists of the faces, this time (real particles only)
    for a, s in over_send:
        xo[(a, s)] = hs[a].ops.pack(s, int(-sent[(a, s)]), shift_of[s])
    for a, s in over_recv:
        xi[(a, s)] = hs[a].ops.new_buffer(int(-recv[(a, s)]), hs[a].ops.nprops)
    for h in hs:
        f = getattr(h.ops, 'before_comm', None)
        if f is not None:
            f()
    reqs = [dist.P2POp(dist.isend, xo[(a, s)], peer) for s, peer, _ in send_order for a in range(na) if (a, s) in xo]
    reqs += [dist.P2POp(dist.irecv, xi[(a, s)], peer) for s, peer, _ in recv_order for a in range(na) if (a, s) in xi]
    if reqs:
        for w in dist.batch_isend_irecv(reqs):
            w.wait()
    for h in hs:
        f = getattr(h.ops, 'after_comm', None)
        if f is not None:
            f()
    # the ghost rows of an array that received an incomplete face are rebuilt: its complete faces from their
    # messages (still in their buffers), the repeated one from the exactly sized message
    for a in sorted(set(a for a, _ in over_recv)):
        h = hs[a]
        h.ops.drop_ghosts()
        for s, _, _ in nbrs:
            if (a, s) in xi:
                h.ops.append(xi[(a, s)], int(-recv[(a, s)]))
            else:
                h.ops.append_padded(inb[a][s], h.recv_rows[s], h.h_promise, h.m_promise, _flag_words(hs), a)
    if hasattr(h0.ops, 'clear_flags'):
        h0.ops.clear_flags()
    for h in hs:
        h.repaired_exchanges += 1
    return False


def _flag_words(hs):
    f = getattr(hs[0].ops, 'flag_words', None)
    return f(len(hs)) if f is not None else None


def _exchange_padded(hs, nbrs):
    """Ghost refresh with NO device->host round trip ('padded' protocol, steady state):
    fixed-capacity messages packed on the device as in 'capacity', but the receiver
    appends ALL rows of a message -- the ghosts and, behind them, padding rows parked
    far outside the domain that are inert on the whole path -- so nothing has to be counted before the
    neighbour update and the evaluation are queued.  Counts and flags are read by
    `verify_halos` once those are queued (or, unverified, at the next exchange)."""
    h0 = hs[0]
    dist, ops0, na = h0.dist, h0.ops, len(hs)
    sides = [s for s, _, _ in nbrs]
    _padded_collect(hs)
    for h in hs:
        h.ops.drop_ghosts()
    shift_of = {s: shift for s, _, shift in nbrs}
    send_order = sorted(nbrs, key=lambda nb: -nb[0])
    recv_order = sorted(nbrs, key=lambda nb: nb[0])
    # ONE message per face and direction: the arrays' messages ([nprops][capacity] rows + header each) back to back in one
    # buffer -- 2 sends + 2 receives per exchange whatever the number of arrays (a dam break's three were 12 point-to-point
    # operations for torch to queue; both ends derive the same layout from the capacities they share)
    out, inb = [{} for _ in hs], [{} for _ in hs]
    face_out, face_in = {}, {}
    for s in sides:
        so = [h.cap_send[s] * h.ops.nprops + 1 for h in hs]
        si = [h.cap_recv[s] * h.ops.nprops + 1 for h in hs]
        face_out[s] = ops0.message_buffer(('face-send', s), sum(so))
        face_in[s] = ops0.message_buffer(('face-recv', s), sum(si))
        oo = oi = 0
        for a in range(na):
            out[a][s] = face_out[s][oo:oo + so[a]]
            inb[a][s] = face_in[s][oi:oi + si[a]]
            oo += so[a]
            oi += si[a]
    for a, h in enumerate(hs):
        h.ops.select_pack(h.lo + h.width, h.hi - h.width,
                          [shift_of.get(0, 0.0), shift_of.get(1, 0.0)],
                          [h.cap_send.get(0, 0), h.cap_send.get(1, 0)],
                          [out[a].get(0), out[a].get(1)])
    for h in hs:
        f = getattr(h.ops, 'before_comm', None)
        if f is not None:
            f()
    reqs = [dist.P2POp(dist.isend, face_out[s], peer) for s, peer, _ in send_order]
    reqs += [dist.P2POp(dist.irecv, face_in[s], peer) for s, peer, _ in recv_order]
    if reqs:
        for w in dist.batch_isend_irecv(reqs):
            w.wait()                  # stream-level for device transports: nothing here blocks the host
    for h in hs:
        f = getattr(h.ops, 'after_comm', None)
        if f is not None:
            f()
    keys = [(a, s) for a in range(na) for s in sides]
    flags = _flag_words(hs)           # ONE set of flag words for all arrays of the exchange, a word per array
    for a, h in enumerate(hs):
        h.recv_rows = dict(h.cap_recv)    # rows each face's message was appended with (a repair re-appends them)
        if hasattr(h.ops, 'append_padded_faces'):
            h.ops.append_padded_faces([inb[a][s] for s, _, _ in nbrs], [h.cap_recv[s] for s, _, _ in nbrs],
                                      h.h_promise, h.m_promise, flags, a)
        else:
            for s, _, _ in nbrs:      # lo side first: deterministic
                h.ops.append_padded(inb[a][s], h.cap_recv[s], h.h_promise, h.m_promise, flags, a)
        h.padded_exchanges += 1
    msgs = [out[a][s] for a, s in keys] + [inb[a][s] for a, s in keys]
    h0.padded_pending = (ops0.queue_headers(msgs, na), keys, out, inb)


class SlabDecomposition(object):
    """All particle arrays of one rank: what ``ParallelManager.update()``
    (parallel_manager.pyx:512-530) does before an acceleration evaluation --
    migrate owned particles that crossed a slab face, then refresh the ghosts
    -- plus ``rebalance()`` (the reference's Zoltan load balance,
    parallel_manager.pyx:577-640, reduced to moving the slab faces)."""

    def __init__(self, arrays, ctx, rank, world, axis, width, lo, hi,
                 props=WCSPH_HALO_PROPS, periodic=False, period=0.0,
                 ops_factory=None, dist=None, protocol=None, promise=True, send_promised=False):
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
                                       period=period, ops=ops, dist=dist, protocol=protocol,
                                       promise=promise, send_promised=send_promised))

    @property
    def lo(self):
        return self.halos[0].lo

    @property
    def hi(self):
        return self.halos[0].hi

    def migrate(self):
        """every array's leavers in ONE counts handshake and ONE batch of transfers
        (`migrate_halos`; round 5: one of each per array)"""
        return migrate_halos(self.halos)

    def exchange(self, drop=True):
        """Ghost refresh of ALL arrays with one batch of point-to-point transfers
        (a dam break has three arrays): see exchange_halos."""
        exchange_halos(self.halos, drop=drop)

    def verify(self):
        """True when the ghosts of the last exchange were complete; False after an
        incomplete face was repeated the counted way -- repeat the neighbour update
        and the evaluation then (`verify_halos`)."""
        return verify_halos(self.halos)

    def renew_promises(self):
        """COLLECTIVE: look at every array's h and m range again (after h or m was
        written) and agree anew on what the ghosts are promised to carry; the next
        exchange is a counted one."""
        _padded_collect(self.halos)
        if self.halos[0].promise:
            _establish_promises(self.halos)     # (also decides anew which arrays are empty everywhere)

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

    def rebalance(self, nbins=4096, weights=None, cost=None):
        """Move the slab faces so that every rank owns about the same number
        of real particles (optionally weighted per array, e.g. fluid particles
        cost more than boundary ones): global histogram of the slab-axis
        coordinate (all_reduce SUM), faces at its quantiles, then migrate until
        every particle is home (a particle moves one slab per round).

        `cost` (seconds this rank spent computing per step, MEASURED: device
        timers, HipParallelManager(cost_fn=...)): equal TIME instead of equal
        count -- every particle of this rank weighs cost / n_local, so a rank that
        was slow per particle (the end slab of a dam-break tank: a sparse grid,
        most of the wall particles) gives particles away.  The reference balances
        by Zoltan's object weights (parallel_manager.pyx:577-640)."""
        import numpy as np
        dist = self.dist
        ops0 = self.halos[0].ops
        # an outstanding 'padded' exchange is settled first: its counts belong to the OLD faces and must not size
        # the messages of the new ones
        _padded_collect(self.halos)
        on_device = all(hasattr(h.ops, 'histogram') and hasattr(h.ops, 'coord_range') for h in self.halos)
        coords = None
        if on_device:
            rng = [h.ops.coord_range() for h in self.halos]
            counts = [h.ops.n_real() for h in self.halos]
            lmin = min(r[0] for r in rng)
            lmax = max(r[1] for r in rng)
        else:
            coords = [h.ops.coords() for h in self.halos]
            counts = [c.size for c in coords]
            lmin = min([c.min() for c in coords if c.size] or [float('inf')])
            lmax = max([c.max() for c in coords if c.size] or [-float('inf')])
        gmin = -allreduce_scalars([-lmin], 'max', dist=dist,
                                  device=getattr(ops0, 'device', None))[0]
        gmax = allreduce_scalars([lmax], 'max', dist=dist,
                                 device=getattr(ops0, 'device', None))[0]
        span = max(gmax - gmin, 1e-300)
        hist = np.zeros(nbins)
        per_particle = 1.0
        if cost is not None:
            nw = sum((1.0 if weights is None else float(weights[k])) * n for k, n in enumerate(counts))
            per_particle = float(cost) / nw if nw > 0 else 0.0
        for k, h in enumerate(self.halos):
            w = per_particle * (1.0 if weights is None else float(weights[k]))
            if on_device:
                if counts[k]:
                    hist += w * h.ops.histogram(gmin, span, nbins)
            else:
                c = coords[k]
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
            h.__dict__.pop('cap_hist', None)
        return faces, rounds


class HipParallelManager(object):
    """The object ``Integrator.set_parallel_manager`` expects
    (integrator.py:274-286 calls ``pm.update()`` before every ``nnps.update()``;
    solver.py:476-480 and parallel_manager.pyx:463 reduce the adaptive
    time-step inputs over all ranks): a ``SlabDecomposition`` plus the scalar
    all-reduces, with an optional re-balance every ``rebalance_every`` updates
    (the reference's ``lb_freq``)."""

    def __init__(self, decomposition, rebalance_every=0, weights=None, cost_fn=None, migrate_every=1, margin=0.0):
        """cost_fn: a callable returning the seconds THIS rank spent computing since its
        last call (e.g. `device_cost_fn(ctx)`: the library's kernel timers); the
        re-balance then equalises measured time instead of particle counts.

        migrate_every = K > 1 (LAZY migration): ownership changes hands every K-th update
        only.  The reference migrates before every evaluation
        (parallel_manager.pyx:512-530); here a migration costs the host two round trips
        (the counts of the leavers, the counts of the arrivals), after which every launch
        of the update is exposed to the launch latency -- the one part of a stepping
        rank's update that the round-trip-free exchange and neighbour update do not
        cover.  Between migrations a particle that crossed a face stays with its old
        rank; that is the same computation as long as the ghost layers reach `margin`
        farther than the kernel support: construct the decomposition with width =
        radius_scale * hmax + margin.  Every migration measures how far its leavers had
        strayed (max over the ranks); more than `margin` is an error -- the evaluations
        since the previous migration may have missed neighbours."""
        self.dec = decomposition
        self.dist = decomposition.dist
        self.rebalance_every = int(rebalance_every)
        self.weights = weights
        self.cost_fn = cost_fn
        self.migrate_every = max(1, int(migrate_every))
        self.margin = float(margin)
        self.max_excursion = 0.0          # the farthest a leaver had strayed at any migration so far
        if self.migrate_every > 1:
            for h in decomposition.halos:
                h.track_excursion = True
        self.count = 0
        self._device = getattr(decomposition.halos[0].ops, 'device', None)

    def update(self):
        self.count += 1
        if self.rebalance_every and self.count % self.rebalance_every == 0:
            cost = self.cost_fn() if self.cost_fn is not None else None
            self.dec.rebalance(weights=self.weights, cost=cost)
            self.dec.exchange()
        elif self.count % self.migrate_every == 0:
            self.dec.update()
            if self.migrate_every > 1:
                far = max(h.max_excursion for h in self.dec.halos)
                far = allreduce_scalars([far], 'max', dist=self.dist, device=self._device)[0]   # (every rank or none raises)
                self.max_excursion = max(self.max_excursion, far)
                if far > self.margin:
                    raise RuntimeError('lazy migration: a particle had travelled %.6g beyond its slab when it was handed over, '
                                       'the ghost layers reach only %.6g beyond the kernel support: migrate more often '
                                       '(migrate_every) or widen them (margin)' % (far, self.margin))
        else:
            self.dec.exchange()

    def verify(self):
        """Integrator.compute_accelerations asks this once the evaluation is queued:
        False = the ghosts were incomplete and have been repaired, evaluate again"""
        return self.dec.verify()

    def reduce_max(self, values):
        return allreduce_scalars(values, 'max', dist=self.dist, device=self._device)

    def reduce_min(self, values):
        return allreduce_scalars(values, 'min', dist=self.dist, device=self._device)


class SphCommTransport(object):
    """`dist` for SlabHalo / SlabDecomposition with the POINT-TO-POINT transfers on
    libsphcomm.so -- ncclSend / ncclRecv in one group on the context's own stream
    (`sph_comm_sendrecv`) -- and everything else (the few collectives of set-up,
    migration and re-balancing) on torch.distributed.  The default of
    `bench.py --gpus N` (SPH_HALO_TRANSPORT=torch selects the process group's
    batch_isend_irecv): the process group's stream hand-over around its RCCL
    kernel costs a slab rank ~45 us of idle stream per exchange.
    The kernels, the messages and the protocol are the same; needs the context
    to run on torch's current stream like the torch transport's fast path."""

    class P2POp(object):
        def __init__(self, op, tensor, peer):
            self.op, self.tensor, self.peer = op, tensor, peer

    isend, irecv = 'isend', 'irecv'

    def __init__(self, ctx, torch_dist, rank, world):
        import torch
        self.ctx, self.td = ctx, torch_dist
        self.ReduceOp = torch_dist.ReduceOp
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libsphcomm.so')
        self.lib = C.CDLL(path)
        self.lib.sph_comm_sendrecv.restype = C.c_int
        ident = (C.c_ubyte * 128)()
        if rank == 0:
            dev._check(self.lib.sph_comm_unique_id(ident))
        t = torch.tensor(list(ident), dtype=torch.uint8, device=torch.device('cuda', ctx.device))
        if world > 1:
            torch_dist.broadcast(t, src=0)
        ident = (C.c_ubyte * 128)(*[int(v) for v in t.cpu()])
        dev._check(self.lib.sph_comm_init_rank(C.c_void_p(ctx._h.value if hasattr(ctx._h, 'value') else ctx._h), rank, world, ident))
        ctx._transport = self          # (HipContext.close destroys the communicator before the context)

    def batch_isend_irecv(self, reqs):
        sends = [r for r in reqs if r.op == 'isend']
        recvs = [r for r in reqs if r.op == 'irecv']

        def arrays(rs):
            n = len(rs)
            return ((C.c_void_p * max(n, 1))(*[r.tensor.data_ptr() for r in rs]),
                    (C.c_size_t * max(n, 1))(*[r.tensor.numel() for r in rs]),
                    (C.c_int * max(n, 1))(*[int(r.peer) for r in rs]))
        sb, sc, sp = arrays(sends)
        rb, rc, rp = arrays(recvs)
        h = self.ctx._h
        dev._check(self.lib.sph_comm_sendrecv(C.c_void_p(h.value if hasattr(h, 'value') else h), len(sends), sb, sc, sp,
                                              len(recvs), rb, rc, rp))
        return []                      # ordered by the stream itself: nothing to wait for

    def close(self):
        h = self.ctx._h
        if h:
            self.lib.sph_comm_destroy(C.c_void_p(h.value if hasattr(h, 'value') else h))

    def __getattr__(self, name):       # all_gather_into_tensor, all_reduce, barrier, ...: torch.distributed
        return getattr(self.td, name)


def device_cost_fn(ctx, keys=('nnps', 'pack', 'eos', 'pair', 'stage')):
    """seconds of kernel time the context has spent since the last call (the
    library's event timers, switched on here): HipParallelManager(cost_fn=...)"""
    ctx.timer_enable(True)

    def cost():
        t = sum(ctx.timer_get(k)[0] for k in keys) * 1e-3
        ctx.timer_reset()
        return t
    return cost


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
