"""Periodic domains: ghost-image particles for the neighbour search.

Mirrors ``DomainManager(xmin, xmax, ymin, ymax, zmin, zmax, periodic_in_x,
periodic_in_y, periodic_in_z, n_layers)`` and its ``update()``
(pysph/base/nnps_base.pyx:227-290, CPUDomainManager :425-483):

    remove the ghosts of the previous call -> box-wrap real particles
    (:699-748) -> create image copies of every particle within
    ``n_layers * cell_size`` of a periodic face, axis by axis, corner images
    included (:751-940); images are tagged Ghost and sit behind the real
    particles (sources only).

Two implementations with the same interface:

* ``DomainManager``     host arrays are authoritative (``sync='auto'``): the
  ghost bookkeeping is done on the host ParticleArray exactly where the
  reference does it (``Integrator.update_domain`` -> ``nnps.update_domain()``);
* ``HipDomainManager``  device-resident state (``sync='manual'``): the same
  steps run as HIP kernels through ``sph_domain_box_wrap`` /
  ``sph_halo_select(mode=1)`` / ``sph_halo_image`` -- the ghost copy is the
  multi-GPU halo copy (``sph_halo_pack`` + ``sph_halo_append``) with a
  coordinate shift, done in place.

Mirror (reflecting) boundaries (``mirror_in_x/y/z``,
``_create_ghosts_mirror`` :506-697) use the same machinery: every particle
(images of earlier axes included -- the corner reflections) within
``n_layers * cell_size`` of a mirror plane gets an image with the coordinate
``x + 2*(plane - x)`` and the normal velocity component negated.
"""
import ctypes as C
import os

import numpy as np

from . import device as dev
from .particle_array import ParticleTAGS


class _DomainBase(object):
    def __init__(self, xmin=-1000., xmax=1000., ymin=0., ymax=0., zmin=0.,
                 zmax=0., periodic_in_x=False, periodic_in_y=False,
                 periodic_in_z=False, n_layers=2.0, props=None,
                 mirror_in_x=False, mirror_in_y=False, mirror_in_z=False):
        self.mirror = [bool(mirror_in_x), bool(mirror_in_y), bool(mirror_in_z)]
        self.is_mirror = any(self.mirror)
        for lo, hi in ((xmin, xmax), (ymin, ymax), (zmin, zmax)):
            if hi < lo:
                raise ValueError('Invalid domain limits')   # _check_limits
        self.lims = [(float(xmin), float(xmax)), (float(ymin), float(ymax)),
                     (float(zmin), float(zmax))]
        self.periodic = [bool(periodic_in_x), bool(periodic_in_y),
                         bool(periodic_in_z)]
        self.is_periodic = any(self.periodic)
        self.n_layers = float(n_layers)
        self.props = props
        self.translate = [hi - lo for lo, hi in self.lims]
        self.radius_scale = 2.0
        self.cell_size = 1.0
        self.particles = []

    # attributes the reference exposes
    xmin = property(lambda self: self.lims[0][0])
    xmax = property(lambda self: self.lims[0][1])
    ymin = property(lambda self: self.lims[1][0])
    ymax = property(lambda self: self.lims[1][1])
    zmin = property(lambda self: self.lims[2][0])
    zmax = property(lambda self: self.lims[2][1])
    periodic_in_x = property(lambda self: self.periodic[0])
    periodic_in_y = property(lambda self: self.periodic[1])
    periodic_in_z = property(lambda self: self.periodic[2])
    mirror_in_x = property(lambda self: self.mirror[0])
    mirror_in_y = property(lambda self: self.mirror[1])
    mirror_in_z = property(lambda self: self.mirror[2])

    @property
    def manager(self):
        """the reference's DomainManager is a facade over a CPU/GPU manager
        (``domain.manager.periodic_in_x``, nnps_base.pyx:227-290); here the
        object is its own manager"""
        return self

    def set_particles(self, particles, radius_scale):
        self.particles = list(particles)
        self.radius_scale = float(radius_scale)


class DomainManager(_DomainBase):
    """Host-side periodic ghosts on the ParticleArrays (sync='auto')."""

    def _cell_size(self):
        # _compute_cell_size_for_binning (nnps_base.pyx:942-978)
        hmax = -1.0
        for pa in self.particles:
            h = pa.properties['h']
            if h.size:
                hmax = max(hmax, float(h.max()))
        cs = self.radius_scale * hmax
        return 1.0 if cs < 1e-6 else cs

    def update(self):
        if not (self.is_periodic or self.is_mirror):
            return
        for pa in self.particles:
            if pa.get_number_of_particles() != pa.get_number_of_particles(True):
                pa.remove_tagged_particles(ParticleTAGS.Ghost)
        self.cell_size = self._cell_size()
        width = self.n_layers * self.cell_size
        for pa in self.particles:
            nreal = pa.get_number_of_particles(True)
            for ax, name in enumerate('xyz'):
                if not self.periodic[ax]:
                    continue
                lo, hi = self.lims[ax]
                v = pa.properties[name]
                v[v < lo] += self.translate[ax]
                v[v > hi] -= self.translate[ax]
            for ax, name in enumerate('xyz'):
                if not self.periodic[ax]:
                    continue
                lo, hi = self.lims[ax]
                v = pa.properties[name]
                low = np.nonzero((v - lo) <= width)[0]
                high = np.nonzero((hi - v) <= width)[0]
                for idx, shift in ((low, self.translate[ax]),
                                   (high, -self.translate[ax])):
                    if idx.size == 0:
                        continue
                    g = pa.extract_particles(idx)
                    g.properties[name] += shift
                    pa.append_parray(g, tag=ParticleTAGS.Ghost)
                    pa.set_num_real_particles(nreal)
            # reflecting planes, after the periodic images (nnps_base.pyx:471-480)
            for ax, (name, vel) in enumerate(zip('xyz', 'uvw')):
                if not self.mirror[ax]:
                    continue
                lo, hi = self.lims[ax]
                v = pa.properties[name]
                low = np.nonzero((v - lo) <= width)[0]
                high = np.nonzero((hi - v) <= width)[0]
                for idx, plane in ((low, lo), (high, hi)):
                    if idx.size == 0:
                        continue
                    g = pa.extract_particles(idx)
                    c = g.properties[name]
                    c += 2.0 * (plane - c)
                    if vel in g.properties:
                        g.properties[vel] *= -1.0
                    pa.append_parray(g, tag=ParticleTAGS.Ghost)
                    pa.set_num_real_particles(nreal)


class HipDomainManager(_DomainBase):
    """Device-resident periodic ghosts (sync='manual')."""

    def __init__(self, *args, **kw):
        """slab: a SlabHalo / SlabDecomposition of the SAME arrays (multi-GPU).
        Its axis is then the slab transport's business -- remote ghosts, the
        periodic wrap 0 <-> P-1 with its coordinate shift
        (nnps_base.pyx:841-856 done between ranks) -- and this manager makes the
        images of the remaining axes AFTER that exchange, so that the corner
        images of remote ghosts exist (the reference gets the same effect from
        processing the axes one after the other, nnps_base.pyx:751-940)."""
        ctx = kw.pop('ctx', None)
        self.slab = kw.pop('slab', None)
        # 'padded' (default since round 6): a steady-state update makes the images without a device->host round trip,
        # into fixed capacities sized from the previous update's counts plus `headroom` (sph_domain_images_padded); the
        # counts are read by `verify()` once the evaluation that uses the images is QUEUED (the host waits for the image
        # kernels only) and images that did not fit -- for a lattice the counts are quantised: a plane that drifts across
        # the threshold brings a whole layer at once -- are made again by a counted update before anything consumes the
        # evaluation (Integrator.compute_accelerations and bench.py repeat it then).  Unverified, images that did not fit
        # are an error of the next update.  'counted': every periodic axis reads its two image counts back (exact sizes,
        # four round trips per update; the reference's own order of things, nnps_base.pyx:751-940).
        self.protocol = kw.pop('protocol', os.environ.get('SPH_DOMAIN_PROTOCOL', 'padded'))
        self.headroom = float(kw.pop('headroom', os.environ.get('SPH_DOMAIN_HEADROOM', '0.125')))
        # the properties an image carries: None = every device property of the array (what the reference copies,
        # nnps_base.pyx:828-856); {array name: names} = those only (set_image_props: an evaluator's inputs -- a dozen
        # of the ~35 properties of a Taylor-Green particle; the rest of an image's row is undefined)
        self.image_props = kw.pop('image_props', None)
        _DomainBase.__init__(self, *args, **kw)
        self.ctx = ctx or dev.get_context()
        self.lib = self.ctx.lib
        self._caps = None           # {(helper index, axis): [cap_lo, cap_hi]} once a counted update has run
        self._queued = False        # counts of the last padded update are on their way to the host
        self.padded_updates = 0
        self.repaired_updates = 0   # padded updates whose images did not fit: made again, counted (verify)

    def set_image_props(self, mapping):
        """restrict the images to the properties somebody reads: {array name: property names}, e.g. the `inputs` of the
        compiled acceleration evaluator; x, y, z, h, m always travel"""
        self.image_props = {k: sorted(set(v) | set(('x', 'y', 'z', 'h', 'm'))) for k, v in dict(mapping).items()}

    def _props_of(self, helper):
        have = helper.device_props()
        if self.image_props is None or helper._pa.name not in self.image_props:
            return have
        want = set(dev.prop_id(p) for p in self.image_props[helper._pa.name])
        return [p for p in have if p in want]

    def verify(self):
        """True when the images of the last update were complete; False after images that did not fit their capacities
        were made again by a counted update -- repeat the neighbour update and the evaluation then.  To be called with
        the evaluation queued and nothing having consumed it (as SlabDecomposition.verify)."""
        if not self._queued:
            return True
        if self._collect_counts(raise_on_overflow=False):
            return True
        self.repaired_updates += 1
        self._caps = None
        self.update()               # counted (no capacities): exact sizes; box-wrapping again changes nothing
        return False

    def set_particles(self, particles, radius_scale):
        _DomainBase.set_particles(self, particles, radius_scale)
        self.helpers = [dev.attach(pa, self.ctx) for pa in self.particles]
        for h in self.helpers:
            owner = getattr(h, 'ghost_owner', None)
            if owner == 'slab' and self.slab is None:
                # both keep their ghosts behind n_real and drop ALL of them on
                # update: uncoordinated, each would delete the other's
                raise RuntimeError(
                    "array '%s' has a slab halo; pass it as HipDomainManager(slab=...) so "
                    "that the two ghost layers are built in one ordered update" % h._pa.name)
            h.ghost_owner = 'domain+slab' if self.slab is not None else 'domain'
            h.managed = True

    def _hmax_known(self):
        """max h over the arrays when every array's range is known without looking (nothing wrote h since the last
        neighbour update / reduction saw it), else None"""
        best = None
        lo, hi = C.c_double(), C.c_double()
        for h in self.helpers:
            if h.get_number_of_particles(True) == 0:
                continue
            if not self.lib.sph_array_h_known(self.ctx._h, h.array_id, C.byref(lo), C.byref(hi)):
                return None
            best = hi.value if best is None else max(best, hi.value)
        return best

    def _capacity(self, count):
        # the parked rows behind the count cost the sort and the record packing their share and the pair passes a little
        # (empty destination tiles): measured on Taylor-Green, 9 % padding rows = +0.15 ms, 2 % = nothing
        return int(count) + int(int(count) * self.headroom) + 1024

    def _collect_counts(self, raise_on_overflow=True):
        """counts of the last padded update, read now: the capacities follow them.  Images that did not fit: False
        (verify repairs them) or, unverified at the next update, an error of the step that used them"""
        out = (C.c_double * (dev.MAX_ARRAYS * 6))()
        dev._check(self.lib.sph_domain_counts_collect(self.ctx._h, out))
        self._queued = False
        for (k, ax), caps in self._caps.items():
            aid = self.helpers[k].array_id
            for side in (0, 1):
                cnt = out[aid * 6 + ax * 2 + side]
                if cnt < 0:
                    self._caps = None
                    if not raise_on_overflow:
                        return False
                    raise RuntimeError(
                        "periodic images of array '%s', axis %d: %d rows did not fit their capacity of %d and nobody "
                        "verified the update -- the last evaluation ran with images missing (call verify() once the "
                        "evaluation is queued: it makes them again; SPH_DOMAIN_PROTOCOL=counted sizes every update exactly)"
                        % (self.helpers[k]._pa.name, ax, int(-cnt), caps[side]))
                cnt = int(cnt)
                if cnt + int(cnt * self.headroom * 0.6) + 512 > caps[side] or \
                        cnt + int(cnt * self.headroom * 2) + 4096 < caps[side]:
                    caps[side] = self._capacity(cnt)
        return True

    def _hmax(self):
        ids = (C.c_int * len(self.helpers))(*[h.array_id for h in self.helpers])
        out = (C.c_double * 8)()
        dev._check(self.lib.sph_nnps_minmax(self.ctx._h, len(self.helpers),
                                            ids, out))
        return out[7]

    def update(self):
        if not (self.is_periodic or self.is_mirror or self.slab is not None):
            return
        lib, ctx = self.lib, self.ctx._h
        slab_axis = self.slab.axis if self.slab is not None else -1
        if self._queued:
            self._collect_counts()
        for h in self.helpers:
            nreal = h.get_number_of_particles(True)
            dev._check(lib.sph_array_resize(ctx, h.array_id, nreal, nreal))
        # a steady-state periodic update needs nothing from the device: h is known while nothing wrote it, the image
        # capacities come from the counts of the previous update
        hmax = self._hmax_known() if (self.protocol == 'padded' and self._caps is not None
                                      and not any(self.mirror)) else None
        padded = hmax is not None
        if not padded:
            self._caps = None
            hmax = self._hmax()
        cs = self.radius_scale * hmax
        self.cell_size = 1.0 if cs < 1e-6 else cs
        width = self.n_layers * self.cell_size
        for h in self.helpers:
            for ax in range(3):
                if self.periodic[ax] and ax != slab_axis:
                    lo, hi = self.lims[ax]
                    dev._check(lib.sph_domain_box_wrap(ctx, h.array_id, ax, lo, hi,
                                                       self.translate[ax]))
        if self.slab is not None:
            self.slab.exchange(drop=False)     # remote ghosts along the slab axis first
        new_caps = {}
        for k, h in enumerate(self.helpers):
            aid = h.array_id
            props = self._props_of(h)
            nprops = len(props)
            pr = (C.c_int * max(nprops, 1))(*props)
            for ax in range(3):
                if not self.periodic[ax] or ax == slab_axis:
                    continue
                lo, hi = self.lims[ax]
                if padded:
                    # both faces' images of all particles present, into the capacities of this (array, axis): no counts
                    # come back now (sph_domain_images_padded)
                    cp = (C.c_size_t * 2)(*self._caps[(k, ax)])
                    dev._check(lib.sph_domain_images_padded(ctx, aid, ax, lo, hi, width, self.translate[ax], cp,
                                                            nprops, pr))
                    continue
                counts = (C.c_size_t * 2)()
                n_all = h.get_number_of_particles()
                dev._check(lib.sph_halo_select(ctx, aid, ax, 1, lo, hi, width,
                                               n_all, counts))
                new_caps[(k, ax)] = [self._capacity(counts[0]), self._capacity(counts[1])]
                # both sides were selected among the n_all particles present
                # before either side's images exist (nnps_base.pyx:805-856)
                for side, shift in ((0, self.translate[ax]),
                                    (1, -self.translate[ax])):
                    if counts[side]:
                        dev._check(lib.sph_halo_image(ctx, aid, side, nprops, pr, ax,
                                                      0, shift, None))
            for ax in range(3):
                if not self.mirror[ax]:
                    continue
                lo, hi = self.lims[ax]
                counts = (C.c_size_t * 2)()
                n_all = h.get_number_of_particles()
                dev._check(lib.sph_halo_select(ctx, aid, ax, 1, lo, hi, width,
                                               n_all, counts))
                for side, plane in ((0, lo), (1, hi)):
                    if counts[side]:
                        dev._check(lib.sph_halo_image(ctx, aid, side, nprops, pr, ax,
                                                      1, plane, None))
        if padded:
            dev._check(lib.sph_domain_counts_queue(ctx))
            self._queued = True
            self.padded_updates += 1
        elif self.protocol == 'padded' and not any(self.mirror):
            self._caps = new_caps
