"""Neighbour search front-end with the reference's ``LinkedListNNPS`` surface.

Mirrors the Python-visible protocol of ``pysph.base.nnps.LinkedListNNPS``
(pysph/base/linked_list_nnps.pyx:28-90 constructor, nnps_base.pyx:1430-1510
``update`` / ``update_domain`` / ``set_context`` / ``get_nearest_particles``,
attributes of nnps_base.pxd:279-371): same constructor arguments, same
attribute names (``cell_size, hmin, xmin, xmax, n_cells, ncells_per_dim,
particles, dim, radius_scale``), same error for a bad cell count
(``RuntimeError``, linked_list_nnps.pyx:307-343).

The data structure behind it is the device-side sorted cell list built by
``sph_nnps_update`` (csrc/sph_nnps.hip); neighbour *lists* are only
materialised by the query methods here -- the pair kernels never need them.
"""
import ctypes as C

import numpy as np

from . import device as dev

from .particle_array import get_npy

_XYZH = ('x', 'y', 'z', 'h')


class HipNNPS(object):
    def __init__(self, dim, particles, radius_scale=2.0, ghost_layers=1,
                 domain=None, fixed_h=False, cache=False, sort_gids=False,
                 ctx=None, sync=True, h_range_reduce=None):
        """`h_range_reduce`: with ``fixed_h`` in a multi-rank run, a callable
        ``(hmin, hmax) -> (global hmin, global hmax)`` (e.g. a MIN/MAX
        all-reduce): ghosts and migrants carry other ranks' smoothing lengths.
        `sync`: push x, y, z, h of every array from the host before each
        ``update()`` (drop-in behaviour: the host owns the data).  With
        ``sync=False`` the positions already on the device are used
        (device-resident pipelines)."""
        self.dim = dim
        self.particles = list(particles)
        self.narrays = len(self.particles)
        self.radius_scale = float(radius_scale)
        self.ghost_layers = ghost_layers
        self.fixed_h = fixed_h
        self.use_cache = cache
        self.sort_gids = sort_gids
        self.domain = domain
        self.sync = sync
        self.ctx = ctx or dev.get_context()
        self.lib = self.ctx.lib
        self.helpers = [dev.attach(pa, self.ctx) for pa in self.particles]
        self.src_index = self.dst_index = 0
        self._grid = None        # (cell_size, hmin, xmin, xmax, ncells_per_dim, n_cells) of the last update, read on demand
        self._extend = (0.0, 0.0, 0.0)   # state of THIS object, handed to the library by every update()
        self._faces = (-1, -float('inf'), float('inf'))
        self.bounds = None       # optional fixed global bounds (multi-GPU)
        self._h_fixed = False
        self._h_range = None     # (hmin, hmax) of THIS neighbour search once fixed_h has seen them
        self.h_range_reduce = h_range_reduce
        self.cell_size_override = -1.0
        if self.domain is not None:
            self.domain.set_particles(self.particles, self.radius_scale)
            self.domain.update()       # LinkedListNNPS.__init__: linked_list_nnps.pyx:88
        self.update()

    # -- reference protocol ----------------------------------------------
    def set_context(self, src_index, dst_index):
        self.src_index, self.dst_index = src_index, dst_index

    def set_in_parallel(self, in_parallel):
        pass

    def set_use_cache(self, use_cache):
        self.use_cache = use_cache

    def update_domain(self):
        """DomainManager.update (nnps_base.pyx:450-483): periodic ghosts; the
        cell size itself is recomputed inside ``update``."""
        if self.domain is not None:
            self.domain.update()
            self._csr_key = None

    def update(self):
        self._csr_key = None    # cached neighbour lists describe the previous particle state
        if self.sync:
            for h in self.helpers:
                h.push(*_XYZH)
        else:
            for h in self.helpers:
                h._sync_size()
        ids = (C.c_int * self.narrays)(*[h.array_id for h in self.helpers])
        # The known h range is state of THIS object, not of the (process-wide)
        # context: every update hands the library its own range, or clears the
        # one another HipNNPS on the same context may have left there.
        if self._h_fixed:
            dev._check(self.lib.sph_nnps_set_h_range(self.ctx._h, *self._h_range))
        else:
            dev._check(self.lib.sph_nnps_set_h_range(self.ctx._h, 0.0, -1.0))
        b = None
        if self.bounds is not None:
            b = (C.c_double * 6)(*self.bounds)
        # ... and so are the ghost-split settings (the grid's extra room, the slab faces)
        dev._check(self.lib.sph_nnps_set_extend(self.ctx._h, *self._extend))
        dev._check(self.lib.sph_nnps_set_ghost_faces(self.ctx._h, *self._faces))
        rc = self.lib.sph_nnps_update(self.ctx._h, self.dim, self.narrays, ids,
                                      self.radius_scale,
                                      self.cell_size_override, b)
        if rc == -3:  # SPH_ERR_CELLS -> the reference raises RuntimeError
            raise RuntimeError(self.lib.sph_last_error().decode())
        dev._check(rc)
        # cell_size, xmin, xmax, n_cells ... are read from the library when
        # somebody looks at them (properties below): a steady-state update()
        # makes no device->host round trip
        self._grid = None
        # the library holds ONE grid per context: remember which update of the context this object's attributes belong to
        self.ctx._nnps_updates = getattr(self.ctx, '_nnps_updates', 0) + 1
        self._update_no = self.ctx._nnps_updates
        if self.fixed_h and not self._h_fixed:
            # fixed_h (linked_list_nnps.pyx:54): the smoothing lengths never
            # change, so the range found by this first update stays: later updates
            # skip the h reduction (and, with `bounds`, the whole min/max pass and
            # its device->host round trip)
            mm = (C.c_double * 8)()
            dev._check(self.lib.sph_nnps_minmax(self.ctx._h, self.narrays, ids, mm))
            lo, hi = mm[3], mm[7]          # (+max, -max) of an empty rank
            if self.h_range_reduce is not None:
                # slab-decomposed runs: ghosts and migrants carry the other
                # ranks' smoothing lengths, so the range must be the global
                # one; a collective, so every rank takes part on its first
                # update -- an empty rank contributes the neutral element
                lo, hi = self.h_range_reduce(lo, hi)
            if hi >= lo:
                self._h_range = (float(lo), float(hi))
                self._h_fixed = True

    def _read_grid(self):
        """cell_size, hmin, xmin, xmax, ncells_per_dim, n_cells of THIS object's last
        update(), read from the library on first use.  Can raise: a bad cell count
        of an update that made no round trip is reported here (or by the next
        update, whichever comes first), and the library keeps one grid per context
        -- after another neighbour search updated the same context this object's
        grid is gone."""
        if self._grid is None:
            if getattr(self, '_update_no', None) != getattr(self.ctx, '_nnps_updates', None):
                raise RuntimeError('the grid attributes of this neighbour search are those of its last update(); another '
                                   'search has updated the same context since (one grid per context): update() again')
            d8 = (C.c_double * 8)()
            i4 = (C.c_long * 4)()
            rc = self.lib.sph_nnps_info(self.ctx._h, d8, i4)
            if rc == -3:   # an update without a round trip reports a bad cell count when its bounds are looked at
                raise RuntimeError(self.lib.sph_last_error().decode())
            dev._check(rc)
            self._grid = (d8[0], d8[1], np.array(d8[2:5]), np.array(d8[5:8]),
                          np.array(i4[0:3], dtype=np.int32), int(i4[3]))
        return self._grid

    # attributes of nnps_base.pxd:279-371, the reference's values for the particles of the last update()
    cell_size = property(lambda self: self._read_grid()[0])
    hmin = property(lambda self: self._read_grid()[1])
    xmin = property(lambda self: self._read_grid()[2])
    xmax = property(lambda self: self._read_grid()[3])
    ncells_per_dim = property(lambda self: self._read_grid()[4])
    n_cells = property(lambda self: self._read_grid()[5])

    def set_extend(self, ex=0.0, ey=0.0, ez=0.0):
        """widen the computed bounds on both sides of every axis (before the
        reference's 1 % padding): the grid of an `update()` that runs while the
        ghosts of a slab exchange are still in flight must hold them.  State of
        this object: every `update()` hands it to the library."""
        self._extend = (float(ex), float(ey), float(ez))

    def set_ghost_faces(self, axis=-1, lo=-float('inf'), hi=float('inf')):
        """name the slab faces outside which this rank's ghosts lie BEFORE `update()`:
        the first half of a split evaluation then leaves the wavefronts that can
        reach a ghost to the second half (axis -1: no faces).  State of this
        object, as `set_extend`."""
        self._faces = (int(axis), float(lo), float(hi))

    def update_ghosts(self, axis=0, lo=-float('inf'), hi=float('inf')):
        """bin the particles that arrived behind the ones the last `update()` saw
        (the ghosts of this step) on the same grid, into tables of their own: the
        second half of a neighbour update that overlaps the ghost exchange.
        `lo`, `hi`: the slab faces along `axis` (every ghost lies outside [lo, hi))."""
        self._csr_key = None
        if self._faces[0] >= 0 and (int(axis), float(lo), float(hi)) != self._faces:
            raise ValueError('update_ghosts: the slab faces %r differ from those named before update() %r'
                             % ((axis, lo, hi), self._faces))
        dev._check(self.lib.sph_nnps_update_ghosts(self.ctx._h, int(axis), float(lo), float(hi)))

    def get_csr(self, src_index, dst_index):
        """(start[nd+1], nbrs) with each list sorted ascending.  nd is the
        DEVICE particle count of the destination (it holds the ghosts of a
        device domain manager / slab halo that the host array never sees); the
        library checks the buffer length it is given against it."""
        nd = self.helpers[dst_index].get_number_of_particles()
        start = np.zeros(nd + 1, dtype=np.uint32)
        total = C.c_size_t()
        s, d = self.helpers[src_index].array_id, self.helpers[dst_index].array_id
        sp = start.ctypes.data_as(dev._PU)
        dev._check(self.lib.sph_nnps_get_csr(self.ctx._h, s, d, sp, nd + 1, None,
                                             0, C.byref(total)))
        nbrs = np.empty(max(total.value, 1), dtype=np.uint32)
        dev._check(self.lib.sph_nnps_get_csr(
            self.ctx._h, s, d, sp, nd + 1, nbrs.ctypes.data_as(dev._PU),
            nbrs.size, C.byref(total)))
        return start, nbrs[:total.value]

    def get_csr_start(self, src_index, dst_index):
        """start[nd+1] only: the exclusive scan of every destination's
        neighbour count (pass 1 of the CSR query; no lists are materialised)."""
        nd = self.helpers[dst_index].get_number_of_particles()
        start = np.zeros(nd + 1, dtype=np.uint32)
        total = C.c_size_t()
        s, d = self.helpers[src_index].array_id, self.helpers[dst_index].array_id
        dev._check(self.lib.sph_nnps_get_csr(self.ctx._h, s, d,
                                             start.ctypes.data_as(dev._PU), nd + 1,
                                             None, 0, C.byref(total)))
        return start

    def count_neighbors(self, src_index, dst_index):
        """total number of (destination, source) neighbour pairs -- pass 1 of
        the CSR query only (nd+1 counters come back, no lists)."""
        nd = self.helpers[dst_index].get_number_of_particles()
        start = np.zeros(nd + 1, dtype=np.uint32)
        total = C.c_size_t()
        s, d = self.helpers[src_index].array_id, self.helpers[dst_index].array_id
        dev._check(self.lib.sph_nnps_get_csr(self.ctx._h, s, d,
                                             start.ctypes.data_as(dev._PU), nd + 1,
                                             None, 0, C.byref(total)))
        return int(total.value)

    def get_nearest_particles(self, src_index, dst_index, d_idx, nbrs=None):
        """Neighbours of destination particle `d_idx` (ascending ids; the
        reference returns cell-traversal order unless sort_gids)."""
        key = (src_index, dst_index)
        if getattr(self, '_csr_key', None) != key:
            self._csr = self.get_csr(src_index, dst_index)
            self._csr_key = key
        start, idx = self._csr
        out = idx[start[d_idx]:start[d_idx + 1]].copy()
        if self.sort_gids and out.size:
            # nnps_base.pyx:1577-1611 _sort_neighbors: by gid when the source
            # has valid gids, by index otherwise (gids[0] == UINT_MAX)
            gid = np.asarray(get_npy(self.particles[src_index], 'gid'))
            if gid.size and gid[0] != np.iinfo(np.uint32).max:
                out = out[np.argsort(gid[out], kind='stable')]
        if nbrs is not None and hasattr(nbrs, 'set_data'):
            nbrs.set_data(out)
        return out

    def get_spatially_ordered_indices(self, pa_index):
        n = self.particles[pa_index].get_number_of_particles()
        perm = np.empty(max(n, 1), dtype=np.uint32)
        dev._check(self.lib.sph_nnps_get_order(
            self.ctx._h, self.helpers[pa_index].array_id,
            perm.ctypes.data_as(dev._PU)))
        return perm[:n]

    def spatially_order_particles(self, pa_index):
        """Reorder the particles into cell order (nnps_base.pyx:1615-1629).
        Host arrays when they are authoritative (sync=True); otherwise the
        device-resident properties are permuted in place.  As in the reference
        (solver.py:296-302) the caller must ``update()`` afterwards."""
        if not self.sync:
            h = self.helpers[pa_index]
            if h.get_number_of_particles() != h.get_number_of_particles(True):
                # device ghosts (periodic images, remote halo) are transient and
                # must stay behind the real particles: drop them, bin the real
                # particles alone, permute; the caller's update_domain() /
                # halo exchange rebuilds the ghosts in the new order
                for g in self.helpers:
                    nreal = g.get_number_of_particles(True)
                    dev._check(self.lib.sph_array_resize(self.ctx._h, g.array_id,
                                                         nreal, nreal))
                self.update()
            dev._check(self.lib.sph_nnps_reorder_array(self.ctx._h, h.array_id))
            self._csr_key = None
            return
        pa = self.particles[pa_index]
        order = self.get_spatially_ordered_indices(pa_index).astype(np.int64)
        for name, arr in pa.properties.items():
            stride = pa.stride.get(name, 1)
            if stride == 1:
                arr[...] = arr[order]
            else:
                arr[...] = arr.reshape(-1, stride)[order].ravel()
        self._csr_key = None


# name used by reference scripts: ``from pysph.base.nnps import LinkedListNNPS``
LinkedListNNPS = HipNNPS
