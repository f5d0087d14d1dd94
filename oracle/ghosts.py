"""CPU restatement of the reference's ghost generation -- TEST INFRASTRUCTURE
(see oracle/sph_oracle.c): only tests/ may import it.

Follows, statement by statement and list by list,
  CPUDomainManager._create_ghosts_periodic   pysph/base/nnps_base.pyx:751-940
  CPUDomainManager._create_ghosts_mirror     pysph/base/nnps_base.pyx:506-697
  CPUDomainManager._box_wrap_periodic        pysph/base/nnps_base.pyx:699-748
in plain Python over lists (the Cython original cimports cyarray and cannot be
built here): flag lists from the real particles first, then per
axis the images of what the ghost array already holds (the "corners"), then the
images of the real particles, in the reference's order.  A ghost is returned as
(source index of the REAL particle it descends from, x, y, z, su, sv, sw) with
s* the sign its velocity components carry (mirror images flip the normal one).

Pinned by the reference's own known answers: the image counts of
base/tests/test_domain_manager.py (a lattice in a periodic box gets
(n + 2 l)^d - n^d ghosts) in tests/test_ghost_sets.py.
"""


def box_wrap(x, y, z, lims, periodic, translate):
    """nnps_base.pyx:735-748 (in place)"""
    for c, ax in ((x, 0), (y, 1), (z, 2)):
        if not periodic[ax]:
            continue
        lo, hi = lims[ax]
        for i in range(len(c)):
            if c[i] < lo:
                c[i] = c[i] + translate[ax]
            if c[i] > hi:
                c[i] = c[i] - translate[ax]


def periodic_ghosts(x, y, z, lims, periodic, translate, width):
    """nnps_base.pyx:751-940 for one particle array; `width` = n_layers * cell_size"""
    n = len(x)
    real = [x, y, z]
    flags = {}
    for ax in range(3):
        lo, hi = lims[ax]
        flags[(ax, 0)] = [i for i in range(n) if periodic[ax] and (real[ax][i] - lo) <= width]
        flags[(ax, 1)] = [i for i in range(n) if periodic[ax] and (hi - real[ax][i]) <= width]
    ghosts = []                      # [src, x, y, z]

    def from_real(idx, ax, shift):
        for i in idx:
            g = [i, x[i], y[i], z[i]]
            g[1 + ax] = g[1 + ax] + shift
            ghosts.append(g)

    def from_ghosts(idx, ax, shift):
        new = []
        for k in idx:
            g = list(ghosts[k])
            g[1 + ax] = g[1 + ax] + shift
            new.append(g)
        ghosts.extend(new)

    if periodic[0]:                                        # :834-849
        from_real(flags[(0, 0)], 0, translate[0])
        from_real(flags[(0, 1)], 0, -translate[0])
    for ax in (1, 2):                                      # :851-935
        if not periodic[ax]:
            continue
        lo, hi = lims[ax]
        low = [k for k in range(len(ghosts)) if (ghosts[k][1 + ax] - lo) <= width]
        high = [k for k in range(len(ghosts)) if (hi - ghosts[k][1 + ax]) <= width]
        from_ghosts(low, ax, translate[ax])
        from_ghosts(high, ax, -translate[ax])
        from_real(flags[(ax, 1)], ax, -translate[ax])      # "y_high" first, then "y_low"
        from_real(flags[(ax, 0)], ax, translate[ax])
    return [tuple(g) + (1.0, 1.0, 1.0) for g in ghosts]


def mirror_ghosts(x, y, z, lims, mirror, width):
    """nnps_base.pyx:506-697 for one particle array"""
    n = len(x)
    real = [x, y, z]
    flags, shifts = {}, {}
    for ax in range(3):
        lo, hi = lims[ax]
        flags[(ax, 0)] = [i for i in range(n) if mirror[ax] and (real[ax][i] - lo) <= width]
        shifts[(ax, 0)] = [-2 * (real[ax][i] - lo) for i in flags[(ax, 0)]]
        flags[(ax, 1)] = [i for i in range(n) if mirror[ax] and (hi - real[ax][i]) <= width]
        shifts[(ax, 1)] = [2 * (hi - real[ax][i]) for i in flags[(ax, 1)]]
    added = []                       # [src, x, y, z, su, sv, sw]

    def from_real(side, ax):
        for i, t in zip(flags[(ax, side)], shifts[(ax, side)]):
            g = [i, x[i], y[i], z[i], 1.0, 1.0, 1.0]
            g[1 + ax] = g[1 + ax] + t
            g[4 + ax] = -g[4 + ax]
            added.append(g)

    def from_added(idx, tr, ax):
        new = []
        for k, t in zip(idx, tr):
            g = list(added[k])
            g[1 + ax] = g[1 + ax] + t
            g[4 + ax] = -g[4 + ax]
            new.append(g)
        added.extend(new)

    if mirror[0]:                                          # :596-609
        from_real(0, 0)
        from_real(1, 0)
    for ax in (1, 2):                                      # :611-693
        if not mirror[ax]:
            continue
        lo, hi = lims[ax]
        low = [k for k in range(len(added)) if (added[k][1 + ax] - lo) <= width]
        lt = [-2 * (added[k][1 + ax] - lo) for k in low]
        high = [k for k in range(len(added)) if (hi - added[k][1 + ax]) <= width]
        ht = [2 * (hi - added[k][1 + ax]) for k in high]
        from_added(low, lt, ax)
        from_added(high, ht, ax)
        from_real(1, ax)                                   # "y_high" first, then "y_low"
        from_real(0, ax)
    return [tuple(g) for g in added]
