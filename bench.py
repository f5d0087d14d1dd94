#!/usr/bin/env python
"""Benchmark of the SPH acceleration-eval hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one resident particle set:
``nnps.update()`` (bounds, cell keys, radix sort, cell ranges) followed by
``AccelerationEval.compute()`` (EOS + pack + fused pair kernel), exactly what
``Integrator.compute_accelerations`` runs (pysph/sph/integrator.py:274-286).
Inputs are already in HBM when the timed region starts.

Workload (BASELINE.json metric): WCSPH, dam-break parameter set
(WendlandQuintic, hdx 1.3, alpha 0.25, gamma 7, c0 = 10 sqrt(2 g 0.55),
gz = -9.81), synthetic uniform 3-D distribution "S-cube" of SURVEY.md 8(d):
159^3 = 4.02 M particles per GPU, jitter U(+-0.1 dx), seeded.  For N > 1 the
domain is N such cubes side by side along x (weak scaling), slab-decomposed one
cube per rank, with a ghost-particle halo exchange on RCCL (torch.distributed
nccl) before every step.

``--gpus N`` with N > 1 and no torchrun environment starts the N ranks itself
(``python -m torch.distributed.run --nproc-per-node N``, rendezvous on
127.0.0.1); under torchrun (RANK / WORLD_SIZE set) it is one of the ranks.

After the timed loop rank 0 (N = 1) checks the device results of the SAME
particle state against the CPU oracle (``parity_max_rel``, ``--no-check`` to
skip) and, for the default workload, measures the secondary configurations of
BASELINE.md section 4 (``extra``: other sizes, unsorted, variable h, the
reference's own "cube" parameterisation; ``--no-extras`` to skip).

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

ALGO_BYTES_PAIR = 160.0      # pair-loop kernel, fp64, per particle-update
ALGO_BYTES_UPDATE = 184.0    # whole compute() (EOS + pair pass)
FLOP_PER_PAIR = 130.0            # fused WCSPH fluid<-fluid group, reference operation count
FP64_VECTOR_PEAK_TFLOPS = 78.6   # MI355X fp64 vector (non-MFMA) peak
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8 TB/s spec
L2_PEAK_TBS = 34.5           # MI355X_MICROARCH.md, L2 section: aggregate L2 bandwidth of the 8 XCDs
PARITY_TOL = 1e-10           # BASELINE.json north_star
PAIR_FAMILIES = ('pair_wcsph', 'pair_density', 'pair_tvf', 'pair_vgrad', 'pair_elastic')
ELEMENTWISE_FLOOR = 1e-6     # element-wise parity: |b_i| floored at this fraction of the field's scale
# Element-wise bound that `parity_ok` ASSERTS (fp64), per field class:
#  * pair sums and per-particle functions whose terms do not cancel systematically (every WCSPH / elastic output):
#    1e-8 -- a sum taken in another order differs by ~1e-16 of the field's scale per particle, i.e. up to 1e-10 at the
#    1e-6 floor times the handful of ulps an 80-term sum collects (the oracle against ITSELF in a permuted particle
#    order: 3.5e-11 at 216 k particles, tests/test_bench_launcher.py); the worst field is the EOS pressure
#    B((rho/rho0)^7 - 1) of particles at rho = rho0 +- 1e-6, an absolute error of 2e-16 B judged against the floor;
#  * lattice sums that cancel to zero (Taylor-Green on the exact lattice: 113 background-pressure terms of magnitude
#    3e4 adding up to nothing, p = p0 (rho/rho0 - 1) = rounding noise x 100): no value of its own to be relative to,
#    every particle sits at the floor, so the figure is the norm-wise error / floor and its bound 1e-10 / 1e-6 = 1e-4.
ELEMENTWISE_TOL = 1e-8
ELEMENTWISE_TOL_CANCELLING = PARITY_TOL / ELEMENTWISE_FLOOR
# fp32 arithmetic (tolerance 5e-5 norm-wise against the fp64 oracle): every particle judged against its own value
# floored at a thousandth of the field's scale, bound = norm-wise tolerance / floor
PARITY_TOL_F32 = 5e-5
ELEMENTWISE_FLOOR_F32 = 1e-3
ELEMENTWISE_TOL_F32 = PARITY_TOL_F32 / ELEMENTWISE_FLOOR_F32


# ---------------------------------------------------------------------------
# workloads
# ---------------------------------------------------------------------------
def make_cube(n1, x_offset=0.0, seed=1234, hdx=1.3, nx=None, vary_h=0.0, gid0=0, vary_m=0.0):
    """S-cube of SURVEY.md 8(d): jittered lattice, random velocities (so the
    artificial-viscosity branch is taken for about half of the pairs)."""
    from pysph_amd.particle_array import get_particle_array_wcsph
    from pysph_amd.examples import dam_break_3d as db
    rng = np.random.default_rng(seed)
    dx = 1.0 / n1
    nx = nx or n1
    gx = np.arange(nx) * dx + x_offset
    g = np.arange(n1) * dx
    x, y, z = [a.ravel().copy() for a in np.meshgrid(gx, g, g, indexing='ij')]
    n = x.size
    for a in (x, y, z):
        a += 0.1 * dx * rng.uniform(-1, 1, n)
    h = hdx * dx * np.ones(n)
    pa = get_particle_array_wcsph(
        name='fluid', x=x, y=y, z=z, h=h,
        m=db.ro * dx ** 3 * np.ones(n),
        rho=db.ro * (1 + 0.01 * rng.uniform(-1, 1, n)),
        u=0.1 * db.c0 * rng.uniform(-1, 1, n),
        v=0.1 * db.c0 * rng.uniform(-1, 1, n),
        w=0.1 * db.c0 * rng.uniform(-1, 1, n))
    if vary_h:
        pa.h[:] = h * (1.0 + vary_h * rng.uniform(-1, 1, n))
    if vary_m:
        pa.m[:] = pa.m * (1.0 + vary_m * np.random.default_rng(seed + 7).uniform(-1, 1, n))
    pa.gid[:] = np.arange(gid0, gid0 + n, dtype=pa.gid.dtype)
    return pa, dx


def cube_equations(dx, hdx=1.3, params='db', gamma=None):
    """'db': dam_break_3d.py parameters; 'cube': the reference's own benchmark
    example (pysph/examples/cube.py:40-55: alpha 0.5, c0 10, hdx 1.5)."""
    from pysph_amd.scheme import WCSPHScheme
    from pysph_amd.examples import dam_break_3d as db
    if params == 'cube':
        s = WCSPHScheme(['fluid'], [], dim=3, rho0=db.ro, c0=10.0, h0=hdx * dx,
                        hdx=hdx, gz=-9.81, alpha=0.5, beta=0.0, gamma=gamma or 7.0)
    else:
        s = WCSPHScheme(['fluid'], [], dim=3, rho0=db.ro, c0=db.c0, h0=hdx * dx,
                        hdx=hdx, gz=-9.81, alpha=db.alpha, beta=db.beta,
                        gamma=gamma or db.gamma)
    return s.get_equations()


def make_taylor_green(n1, x_offset=0.0):
    from pysph_amd.particle_array import get_particle_array_tvf_fluid
    dx = 1.0 / n1
    g = (np.arange(n1) + 0.5) * dx
    x, y, z = [a.ravel().copy() for a in np.meshgrid(g + x_offset, g, g, indexing='ij')]
    pa = get_particle_array_tvf_fluid(
        name='fluid', x=x, y=y, z=z, h=dx * np.ones(x.size),
        m=dx ** 3 * np.ones(x.size), rho=np.ones(x.size),
        u=-np.cos(2 * np.pi * x) * np.sin(2 * np.pi * y),
        v=np.sin(2 * np.pi * x) * np.cos(2 * np.pi * y))
    pa.uhat[:] = pa.u
    pa.vhat[:] = pa.v
    return pa, dx


def make_rings3d(dx, spacing=0.041, seed=7, perturb=True):
    """S-rings3d of SURVEY.md 8(d) (BASELINE config 5): the colliding rings of
    pysph/examples/solid_mech/rings.py:20-80 taken to 3-D -- two hollow spheres
    (inner radius 0.03, outer 0.04) on a lattice, centres 2 x `spacing` apart
    along x, ONE particle array, CubicSpline hdx 1.5, E 1e7, nu 0.3975, rho0 1,
    approaching each other with u = +-0.059 cs.  dx = 5.372e-4 gives 2.0 M
    particles.  The reference starts from the stress-free state, where every
    rate but the velocity-gradient one vanishes inside a body; a seeded
    perturbation (rho +-1 %, velocities +-0.01 cs on top of the approach
    speed, deviatoric stresses +-1e-3 E) makes every term of the equation set
    act, tension included (the artificial-stress eigen-decomposition).
    `perturb=False`: the state rings.py starts from (uniform density, no stress,
    the approach speed only) -- no particle in tension, the artificial stress
    r_ij is zero everywhere and the rates kernel does not gather it."""
    from pysph_amd import kernels as K
    from pysph_amd.solid_mech import get_particle_array_elastic_dynamics
    E, nu, rho0, hdx = 1e7, 0.3975, 1.0, 1.5     # rings.py:21-28
    ri, ro = 0.03, 0.04                           # rings.py:31-32
    g = np.arange(-ro, ro, dx)                    # numpy.mgrid[-ro:ro:dx], rings.py:42
    x, y, z = [a.ravel() for a in np.meshgrid(g, g, g, indexing='ij')]
    d = x * x + y * y + z * z
    keep = np.flatnonzero((ri * ri <= d) & (d < ro * ro))
    x, y, z = x[keep], y[keep], z[keep]
    side = np.concatenate([np.ones(x.size), -np.ones(x.size)])   # +1: left body, moving right
    x = np.concatenate([x - spacing, x + spacing])
    y = np.concatenate([y, y])
    z = np.concatenate([z, z])
    n = x.size
    kernel = K.CubicSpline(dim=3)
    h0 = hdx * dx
    rng = np.random.default_rng(seed)
    pa = get_particle_array_elastic_dynamics(
        name='solid', x=x + spacing, y=y, z=z, h=h0 * np.ones(n),
        m=rho0 * dx ** 3 * np.ones(n),
        rho=rho0 * (1 + (0.01 if perturb else 0.0) * rng.uniform(-1, 1, n)),
        constants=dict(E=E, nu=nu, rho_ref=rho0, n=4,
                       wdeltap=float(kernel.kernel(rij=dx, h=h0))))
    cs = float(pa.cs[0])
    amp = 0.01 if perturb else 0.0
    pa.u[:] = cs * 0.059 * side + amp * cs * rng.uniform(-1, 1, n)   # rings.py:76-77
    pa.v[:] = amp * cs * rng.uniform(-1, 1, n)
    pa.w[:] = amp * cs * rng.uniform(-1, 1, n)
    for c in ('s00', 's01', 's02', 's11', 's12', 's22'):
        pa.get(c)[:] = (1e-3 if perturb else 0.0) * E * rng.uniform(-1, 1, n)
    pa.gid[:] = np.arange(n, dtype=pa.gid.dtype)
    return pa, kernel


def make_elastic(n1):
    from pysph_amd import kernels as K
    from pysph_amd.solid_mech import get_particle_array_elastic_dynamics
    dx = 1.0 / n1
    g = (np.arange(n1) + 0.5) * dx
    x, y, z = [a.ravel().copy() for a in np.meshgrid(g, g, g, indexing='ij')]
    rng = np.random.default_rng(7)
    E, nu, rho0 = 1e7, 0.3975, 1.2            # rings.py:21-38
    kernel = K.CubicSpline(dim=3)
    h0 = 1.3 * dx
    pa = get_particle_array_elastic_dynamics(
        name='solid', x=x, y=y, z=z, h=h0 * np.ones(x.size),
        m=rho0 * dx ** 3 * np.ones(x.size), rho=rho0 * np.ones(x.size),
        u=1e-2 * rng.uniform(-1, 1, x.size),
        constants=dict(E=E, nu=nu, rho_ref=rho0, n=4,
                       wdeltap=float(kernel.kernel(rij=dx, h=h0))))
    pa.gid[:] = np.arange(x.size, dtype=pa.gid.dtype)
    return pa, dx, kernel


class Workload(object):
    """particle arrays + equations + kernel (+ periodic domain) of one rank"""
    ghost_parts = None    # --emulate-rank: per array, the ghost rows a rank's neighbours would send (appended by setup)
    domain_kw = None      # HipDomainManager arguments (periodic workloads)
    scaling = 'weak'
    algo_pair = ALGO_BYTES_PAIR   # algorithmic bytes of the pair passes of one evaluation, per real particle of arrays[0]
    algo_solid = 0.0              # ... per particle of the other arrays (dam break: boundary, obstacle)
    fields = ()           # output properties the parity checks compare
    ew_tol = ELEMENTWISE_TOL  # asserted element-wise bound (fp64 runs), see ELEMENTWISE_TOL
    slab = None           # (lo, hi, periodic, period) of this rank's slab
    slab_axis = 0         # ... along this axis
    halo_width = 0.0


def _dam_weights(arrays, args):
    """work per particle for the slab cut of a dam break (--slab-weight-solid; default 1: equal particle counts.
    A boundary / obstacle particle is a destination of the continuity equation over the fluid only, which argues for
    less -- measured on the emulated ranks of the 17 M tank it does not pay: the end slab that holds most of them also
    holds the sparsest grid, whose neighbour update costs 0.25 ms against 0.11: slowest rank 1.77 / 1.65 / 1.53 / 1.42 ms
    at weight 0.25 / 0.5 / 0.75 / 1)"""
    ws = float(args.slab_weight_solid)
    return np.concatenate([np.full(a.get_number_of_particles(), 1.0 if a.name == 'fluid' else ws) for a in arrays])


_DAM_CACHE = {}     # --emulate-rank: the whole tank, built once for all emulated ranks of one invocation


def cut_slab(w, pa, rank, world, halo_width):
    """N > 1, ONE problem (strong scaling): this rank's slab of `pa` along x,
    cut at the quantiles of x (equal particle counts)"""
    if world == 1:
        return pa
    from pysph_amd.parallel import slab_bounds
    cuts = slab_bounds(pa.x, world)
    lo = -1e30 if rank == 0 else float(cuts[rank])
    hi = 1e30 if rank == world - 1 else float(cuts[rank + 1])
    w.scaling = 'strong'
    w.slab = (lo, hi, False, 0.0)
    w.halo_width = halo_width
    return pa.extract_particles(np.nonzero((pa.x >= lo) & (pa.x < hi))[0], name=pa.name)


def build_workload(args, rank, world):
    from pysph_amd import kernels as K
    w = Workload()
    n1 = args.n1
    if args.workload == 'cube':
        hdx = args.hdx or (1.5 if args.params == 'cube' else 1.3)
        pa, dx = make_cube(n1, x_offset=float(rank), seed=1234 + rank, hdx=hdx,
                           vary_h=args.vary_h, gid0=rank * n1 ** 3, vary_m=args.vary_m)
        w.arrays = [pa]
        w.eqs = cube_equations(dx, hdx=hdx, params=args.params, gamma=args.gamma)
        w.kernel = K.CubicSpline(dim=3) if args.params == 'cube' else K.WendlandQuintic(dim=3)
        w.name = ('S-cube WCSPH %s parameter set (%s, hdx %g), %d^3 = %d particles per GPU, '
                  'jitter 0.1dx, seed 1234%s' % (
                      'cube.py' if args.params == 'cube' else 'dam-break',
                      type(w.kernel).__name__, hdx, n1, pa.get_number_of_particles(),
                      ', h +-%g %%' % (100 * args.vary_h) if args.vary_h else ''))
        w.halo_width = w.kernel.radius_scale * hdx * dx * (1.0 + args.vary_h)
        w.slab = (float(rank), float(rank + 1), False, 0.0)
        if args.self_slab and world == 1:
            # timing of the halo path only: the cube made periodic in x through the
            # slab transport (a different problem than the oracle's: no parity check)
            w.slab = (0.0, 1.0, True, 1.0)
            args.no_check = True
        w.fields = ('arho', 'au', 'av', 'aw', 'ax', 'ay', 'az', 'p', 'cs')
    elif args.workload == 'dam_break':
        from pysph_amd.examples import dam_break_3d as db
        cached = _DAM_CACHE.get(args.dx) if args.emulate_rank else None
        arrays = [a.extract_particles(np.arange(a.get_number_of_particles()), name=a.name) for a in cached] \
            if cached else db.create_particles(args.dx)
        if args.emulate_rank and not cached:
            # (the tank as created -- BEFORE the seeded perturbation below, which every use applies itself)
            _DAM_CACHE.clear()
            _DAM_CACHE[args.dx] = [a.extract_particles(np.arange(a.get_number_of_particles()), name=a.name) for a in arrays]
        gid0 = 0
        # the example starts from rest at uniform density, where every pair
        # term but gravity vanishes: give the fluid the S-cube's seeded
        # perturbation (rho +-1 %, velocities +-0.1 c0) so that the pressure and
        # viscosity branches run and the parity check has something to compare
        rng = np.random.default_rng(4321)
        for a in arrays:
            n = a.get_number_of_particles()
            a.gid[:] = np.arange(gid0, gid0 + n, dtype=a.gid.dtype)
            gid0 += n
            if a.name == 'fluid':
                a.rho[:] = a.rho * (1 + 0.01 * rng.uniform(-1, 1, n))
                for c in 'uvw':
                    a.get(c)[:] = 0.1 * db.c0 * rng.uniform(-1, 1, n)
            if args.vary_h:
                # --vary-h: every particle its own smoothing length (a dam break with evolving h: the variable-h records)
                a.h[:] = a.h * (1 + args.vary_h * np.random.default_rng(977 + len(a.name)).uniform(-1, 1, n))
        lo, hi = -1e30, 1e30
        if args.emulate_rank:
            # ONE rank of an N-rank strong-scaling run WITHOUT the other ranks: this rank's slab of the tank (cut at the
            # quantiles of x like `--gpus N`) and, behind its real particles, the ghost layers its two neighbours
            # would send (their particles within the halo width of the faces).  Times what a rank computes per step;
            # the exchange that delivers the ghosts is measured separately (--self-slab).
            from pysph_amd.parallel import slab_bounds
            er, ew = args.emulate_rank
            sc = 'xyz'[args.slab_axis]      # the slab axis (--slab-axis)
            cuts = slab_bounds(np.concatenate([a.get(sc, only_real_particles=False) for a in arrays]), ew, weights=_dam_weights(arrays, args))
            elo = -1e30 if er == 0 else float(cuts[er])
            ehi = 1e30 if er == ew - 1 else float(cuts[er + 1])
            width = db.create_kernel().radius_scale * 1.3 * args.dx
            cut, ghosts_of = [], []
            for a in arrays:
                ac = a.get(sc, only_real_particles=False)
                real = np.nonzero((ac >= elo) & (ac < ehi))[0]
                ghost = np.nonzero(((ac >= elo - width) & (ac < elo)) | ((ac >= ehi) & (ac < ehi + width)))[0]
                if args.self_slab:
                    # --self-slab: no ghosts are built here -- the rank gets them through the slab transport itself, as its
                    # own periodic neighbour (its low-face particles arrive beyond its high face and the other way round: as
                    # many ghosts per array and face as the real neighbours would send, through the real selection, packing,
                    # RCCL send / recv and append of every array): what a rank's exchange costs LOCALLY, everything but the link
                    ghost = ghost[:0]
                # the ghosts go behind the real particles AFTER the device-side reorder of `setup` (which drops every row
                # behind the real particles -- round 5's emulated ranks ran WITHOUT their ghosts after it: their step
                # times, 1.33-1.38 ms, left the ghosts' share of the neighbour update, the records and the pair loops out)
                b = a.extract_particles(real, name=a.name)
                # (as they arrive: the low face's ghosts, then the high face's, each in its sender's cell order)
                def cell_order(idx):
                    cx, cy, cz = (np.floor(a.get(q)[idx] / width).astype(np.int64) for q in 'xyz')
                    return idx[np.lexsort((a.x[idx], cx, cy, cz))]
                gh = a.extract_particles(np.concatenate([cell_order(ghost[ac[ghost] < elo]), cell_order(ghost[ac[ghost] >= ehi])]),
                                         name=a.name)
                gh.tag[:] = 1
                cut.append(b)
                ghosts_of.append(gh)
            arrays = cut
            w.ghost_parts = ghosts_of
            w.scaling = 'strong'
        elif world > 1:
            # C4: ONE tank cut into `world` slabs along x at the quantiles of
            # all particles' x (equal counts: the fluid fills 38 % of the tank)
            from pysph_amd.parallel import slab_bounds
            sc = 'xyz'[args.slab_axis]
            cuts = slab_bounds(np.concatenate([a.get(sc, only_real_particles=False) for a in arrays]), world, weights=_dam_weights(arrays, args))
            lo = -1e30 if rank == 0 else float(cuts[rank])
            hi = 1e30 if rank == world - 1 else float(cuts[rank + 1])
            arrays = [a.extract_particles(np.nonzero((a.get(sc, only_real_particles=False) >= lo) & (a.get(sc, only_real_particles=False) < hi))[0],
                                          name=a.name) for a in arrays]
            w.scaling = 'strong'
        w.arrays = arrays
        dx = args.dx
        scheme = db.create_scheme(dx)
        if args.gamma:
            scheme.gamma = args.gamma
        w.eqs = scheme.get_equations()
        w.kernel = db.create_kernel()
        w.name = ('3D dam break (dam_break_3d.py geometry), dx=%g: %s' % (
            dx, ', '.join('%s %d' % (a.name, a.get_number_of_particles())
                          for a in arrays)))
        if args.emulate_rank:
            w.name += ' -- rank %d of %d emulated: %s real particles + the ghost layers of its neighbours' % (
                args.emulate_rank[0], args.emulate_rank[1], sum(a.get_number_of_particles(True) for a in arrays))
        w.halo_width = w.kernel.radius_scale * 1.3 * dx * (1.0 + args.vary_h)
        # solids <- fluid continuity: x,y,z,h,u,v,w read (56 B), arho written (8 B): SURVEY 8(d)
        w.algo_solid = 64.0
        w.slab = (lo, hi, False, 0.0)
        w.slab_axis = args.slab_axis
        if args.emulate_rank and args.self_slab:
            if args.emulate_rank[0] in (0, args.emulate_rank[1] - 1):
                raise SystemExit('--emulate-rank with --self-slab needs an interior rank (two faces)')
            w.slab = (elo, ehi, True, ehi - elo)
            args.no_check = True
        w.fields = ('arho', 'au', 'av', 'aw', 'ax', 'ay', 'az', 'p', 'cs')
    elif args.workload == 'taylor_green':
        from pysph_amd.scheme import TVFScheme
        pa, dx = make_taylor_green(n1, x_offset=float(rank))
        pa.gid[:] = np.arange(rank * n1 ** 3, (rank + 1) * n1 ** 3, dtype=pa.gid.dtype)
        w.arrays = [pa]
        w.eqs = TVFScheme(['fluid'], [], dim=3, rho0=1.0, c0=10.0, nu=0.01,
                          p0=100.0, pb=100.0, h0=dx).get_equations()
        w.kernel = K.QuinticSpline(dim=3)
        # unit cube per rank, periodic box [0, world] x [0,1] x [0,1]
        w.domain_kw = dict(xmin=0, xmax=float(world), ymin=0, ymax=1, zmin=0, zmax=1,
                           periodic_in_x=True, periodic_in_y=True, periodic_in_z=True)
        # density pass 40 R + 16 W, force pass 112 R + 48 W (TVF passes 1 and 2)
        w.algo_pair = 216.0
        w.name = ('Taylor-Green 3D TVF (taylor_green.py parameters), periodic unit '
                  'cube %d^3 = %d particles per GPU, QuinticSpline hdx 1.0' % (
                      n1, pa.get_number_of_particles()))
        # two ghost layers (nnps_base.pyx:231): the real=False density group
        # recomputes V, rho on the inner layer from the outer one
        w.halo_width = 2.0 * w.kernel.radius_scale * dx
        w.slab = (float(rank), float(rank + 1), True, float(world))
        # p = p0 (rho / rho0 - 1) is left out: on the lattice rho = rho0 to 1e-15,
        # p is pure cancellation noise with no scale of its own; rho and V carry it
        w.fields = ('rho', 'V', 'au', 'av', 'aw', 'auhat', 'avhat', 'awhat')
        w.ew_tol = ELEMENTWISE_TOL_CANCELLING   # exact lattice: the force sums cancel to zero
        if args.dtype == 'f32':
            # On the exact lattice both force sums are pure cancellation: the
            # background-pressure sum is ~113 terms of magnitude 3e4 adding up
            # to zero, and p = p0 (rho/rho0 - 1) is the rounding noise of rho
            # times 100.  fp32 (as in the reference's own fp32 GPU backends)
            # leaves errors of order 0.1 of the acceleration scale there -- a
            # property of the formulation in fp32, not of this kernel; only the
            # density sum is comparable.
            w.fields = ('rho', 'V')
    elif args.workload == 'elastic':
        from pysph_amd.solid_mech import ElasticSolidsScheme
        spacing = args.rings_spacing if args.rings_spacing > 0 else 0.041
        pa, kernel = make_rings3d(args.rings_dx, spacing=spacing, perturb=not args.rings_unperturbed)
        pa = cut_slab(w, pa, rank, world, kernel.radius_scale * 1.5 * args.rings_dx)
        w.arrays = [pa]
        w.kernel = kernel
        w.eqs = ElasticSolidsScheme(['solid'], [], dim=3,
                                    ghost_recompute=world > 1).get_equations()
        # velocity-gradient pass: x,y,z,h,u,v,w,m,rho read (72 B), v00..v22 written (72 B);
        # rates pass: x,y,z,h,u,v,w,m,rho,cs,p,s_ij,r_ij read (184 B), arho,au..aw,ax..az written (56 B)
        w.algo_pair = 144.0 + 240.0
        w.name = ('S-rings3d: two hollow spheres (ri 0.03, ro 0.04, centres %g apart), '
                  'Gray 2001 elastic set, rings.py material and approach speed, '
                  'CubicSpline hdx 1.5, dx %g: %d particles' % (
                      2 * spacing, args.rings_dx, pa.get_number_of_particles()))
        w.fields = ('arho', 'au', 'av', 'aw', 'ax', 'ay', 'az', 'p',
                    'as00', 'as01', 'as02', 'as11', 'as12', 'as22',
                    'v00', 'v01', 'v02', 'v10', 'v11', 'v12', 'v20', 'v21', 'v22',
                    'r00', 'r01', 'r02', 'r11', 'r12', 'r22')
    else:
        from pysph_amd.solid_mech import ElasticSolidsScheme
        pa, dx, kernel = make_elastic(n1)
        pa = cut_slab(w, pa, rank, world, kernel.radius_scale * 1.3 * dx)
        w.arrays = [pa]
        w.kernel = kernel
        w.eqs = ElasticSolidsScheme(['solid'], [], dim=3,
                                    ghost_recompute=world > 1).get_equations()
        w.algo_pair = 144.0 + 240.0
        w.name = ('Elastic solid block (Gray 2001 equation set, rings.py material), '
                  '%d^3 = %d particles, CubicSpline hdx 1.3' % (n1, pa.get_number_of_particles()))
        w.fields = ('arho', 'au', 'av', 'aw', 'ax', 'ay', 'az', 'p',
                    'as00', 'as01', 'as02', 'as11', 'as12', 'as22')
    return w


# ---------------------------------------------------------------------------
# CPU legs (rank 0, outside the timed region): baseline and checker
# ---------------------------------------------------------------------------
def _host_threads():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def cpu_baseline(n1=100, target_seconds=15.0):
    """The oracle ("port": C/OpenMP restatement of the reference's Cython path)
    timed on this box's host cores on a bounded sample of the same workload."""
    from oracle import oracle as orc
    from pysph_amd import kernels as K
    avail = _host_threads()
    pa, dx = make_cube(n1, seed=99)
    eqs = cube_equations(dx)
    nn = orc.OracleNNPS(3, [pa], 2.0)
    ev = orc.OracleEval([pa], eqs, K.WendlandQuintic(dim=3), nthreads=avail)
    ev.set_nnps(nn)

    def one():
        t0 = time.perf_counter()
        nn.update()
        ev.compute(0.0, 1e-5)
        return time.perf_counter() - t0
    # thread count: the reference advises tuning it (installation.rst:1043);
    # take the fastest of a short sweep, report the count actually used
    best = None
    for nt in sorted(set([avail, max(avail // 2, 1), max(avail // 4, 1)])):
        ev.nthreads = nt
        one()
        t = one()
        if best is None or t < best[0]:
            best = (t, nt)
    t_first, cores = best
    ev.nthreads = cores
    reps = int(max(1, min(10, target_seconds / max(t_first, 1e-3))))
    ts = [one() for _ in range(reps)]
    t = float(np.median(ts))
    n = n1 ** 3
    return {'value': n / t, 'unit': 'particle-updates/s', 'cores': cores,
            'kind': 'port',
            'sample': 'S-cube %d^3=%d particles, WCSPH db set, fp64, '
                      'nnps.update+compute, median of %d passes (OpenMP '
                      'schedule(dynamic,64), %d threads)' % (n1, n, reps, cores)}


def field_error(a, b, scale_fields, elementwise=False, floor=None):
    """max|a-b| / max|b|, the components of one vector (`scale_fields`) sharing
    their scale -- a component that vanishes by symmetry, the y force of a
    lattice, has no scale of its own; absolute when the scale is zero.  A NaN or
    Inf anywhere in `a` is the worst possible error (a NaN never compares
    greater, so it must not reach the comparison).

    `elementwise`: also the stricter element-wise figure
    max_i |a_i - b_i| / max(|b_i|, 1e-6 max|b|) -- every particle judged against
    its OWN value, floored at a millionth of the field's scale (a sum that
    cancels to nothing has no relative accuracy in any summation order);
    returns the pair (norm-wise, element-wise)."""
    scale = max([float(np.max(np.abs(g))) for g in scale_fields if g.size] + [0.0])
    if a.size == 0:
        return (0.0, 0.0) if elementwise else 0.0
    if not np.all(np.isfinite(a)):
        return (1e300, 1e300) if elementwise else 1e300
    d = np.abs(a - b)
    err = float(np.max(np.abs(a))) if scale == 0.0 else float(np.max(d) / scale)
    err = err if np.isfinite(err) else 1e300
    if not elementwise:
        return err
    if scale == 0.0:
        return err, err
    ew = float(np.max(d / np.maximum(np.abs(b), (ELEMENTWISE_FLOOR if floor is None else floor) * scale)))
    return err, (ew if np.isfinite(ew) else 1e300)


def parity_check(w, host_in, nnps, domain, tol=PARITY_TOL, ew_tol='auto'):
    """`parity_ok` = norm-wise error < tol AND element-wise error < ew_tol AND no
    neighbour-count mismatch.  ew_tol 'auto': the workload's bound (`Workload.ew_tol`,
    see ELEMENTWISE_TOL) for fp64 tolerances; for the fp32 tolerance the floor is
    ELEMENTWISE_FLOOR_F32 = 1e-3 of the field's scale (an fp32 value judged against
    a 1e-6 floor says nothing) and the bound ELEMENTWISE_TOL_F32 = tol / floor.

    Device results of the state the timed loop ran on vs the CPU oracle on
    the SAME inputs (tests/ and this leg are the only users of oracle/): every
    output field, norm-wise (max|a-b| / max|b| per field) AND element-wise
    (`field_error`), plus the neighbour COUNT of every real destination (exact;
    periodic workloads: ghosts count as sources on both sides).  `host_in`:
    pristine host copies of the inputs in the device's particle order."""
    from oracle import oracle as orc
    ref = host_in
    nt = _host_threads()
    if domain is not None:
        # periodic workload: the oracle runs on the host domain manager's ghosts
        from pysph_amd.domain import DomainManager
        dm = DomainManager(**w.domain_kw)
        dm.set_particles(ref, w.kernel.radius_scale)
        dm.update()
    onn = orc.OracleNNPS(3, ref, w.kernel.radius_scale)
    onn.update()
    oev = orc.OracleEval(ref, w.eqs, w.kernel, nthreads=nt)
    oev.set_nnps(onn)
    t0 = time.perf_counter()
    oev.compute(0.0, 1e-5)
    t_oracle = time.perf_counter() - t0
    worst, worst_field = 0.0, None
    worst_ew, worst_ew_field = 0.0, None
    f32 = tol > PARITY_TOL
    floor = ELEMENTWISE_FLOOR_F32 if f32 else ELEMENTWISE_FLOOR
    for pa, pr in zip(w.arrays, ref):
        nreal = pr.get_number_of_particles(True)
        if nreal == 0:
            continue
        pa.gpu.pull(*[f for f in w.fields if f in pa.properties])
        for f in w.fields:
            if f not in pa.properties or f not in pr.properties:
                continue
            err, ew = field_error(
                np.asarray(pa.get(f))[:nreal], np.asarray(pr.get(f))[:nreal],
                [np.asarray(pr.get(g))[:nreal] for g in _scale_group(f) if g in pr.properties],
                elementwise=True, floor=floor)
            if err > worst:
                worst, worst_field = err, '%s.%s' % (pa.name, f)
            if ew > worst_ew:
                worst_ew, worst_ew_field = ew, '%s.%s' % (pa.name, f)
    if ew_tol == 'auto':
        ew_tol = w.ew_tol if not f32 else ELEMENTWISE_TOL_F32
    out = {'parity_max_rel': worst, 'parity_worst_field': worst_field,
           'parity_elementwise_max_rel': worst_ew,
           'parity_elementwise_worst_field': worst_ew_field,
           'parity_elementwise_floor': floor,
           'parity_elementwise_tolerance': ew_tol,
           'parity_tolerance': tol,
           'parity_ok': bool(worst < tol and (ew_tol is None or worst_ew < ew_tol)),
           'parity_oracle_seconds': t_oracle, 'parity_oracle_threads': nt}
    # neighbour counts of every REAL destination, exact (the oracle's criterion
    # is the reference's, linked_list_nnps.pyx:176-184); sources are all
    # particles of the source array -- with a periodic domain that includes the
    # ghost images, whose ORDER differs between the host and the device domain
    # managers while their set does not (tests/test_ghost_sets.py)
    mism = 0
    for di in range(len(w.arrays)):
        nreal = ref[di].get_number_of_particles(True)
        for si in range(len(w.arrays)):
            if w.arrays[di].get_number_of_particles() == 0 or \
                    w.arrays[si].get_number_of_particles() == 0:
                continue
            start = np.asarray(nnps.get_csr_start(si, di))
            ostart = np.asarray(onn.count_csr(si, di, nthreads=nt))
            if start.size <= nreal or ostart.size <= nreal:
                mism += nreal
                continue
            mism += int(np.count_nonzero(np.diff(start[:nreal + 1].astype(np.int64)) !=
                                         np.diff(ostart[:nreal + 1].astype(np.int64))))
    out['parity_neighbour_count_mismatches'] = mism
    out['parity_ok'] = bool(out['parity_ok'] and mism == 0)
    return out


_VECTORS = (('au', 'av', 'aw', 'auhat', 'avhat', 'awhat'), ('ax', 'ay', 'az'),
            ('as00', 'as01', 'as02', 'as11', 'as12', 'as22'),
            ('v00', 'v01', 'v02', 'v10', 'v11', 'v12', 'v20', 'v21', 'v22'),
            ('r00', 'r01', 'r02', 'r11', 'r12', 'r22'))


def _scale_group(f):
    for g in _VECTORS:
        if f in g:
            return g
    return (f,)


def copy_arrays(arrays):
    """pristine host copies (same particle order) for the oracle"""
    out = []
    for a in arrays:
        b = a.extract_particles(np.arange(a.get_number_of_particles()), name=a.name)
        b.set_num_real_particles(a.get_number_of_particles(True))     # (--emulate-rank: ghosts behind the real particles)
        out.append(b)
    return out


# ---------------------------------------------------------------------------
def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--n1', type=int, default=159, help='lattice side per GPU')
    ap.add_argument('--workload', default='cube',
                    choices=['cube', 'dam_break', 'taylor_green', 'elastic', 'elastic_block'],
                    help='cube = S-cube WCSPH (headline); dam_break / taylor_green / elastic (= S-rings3d): '
                         'BASELINE configs 2/3/5; elastic_block: a solid block, hdx 1.3 (round-2 stand-in)')
    ap.add_argument('--rings-dx', type=float, default=5.372e-4, dest='rings_dx',
                    help='elastic workload: lattice spacing (5.372e-4: 2.0 M particles)')
    ap.add_argument('--rings-spacing', type=float, default=0.0, dest='rings_spacing',
                    help='elastic workload: half the distance of the centres (default 0.041, rings.py:34; '
                         '0.04 + dx puts the bodies in contact)')
    ap.add_argument('--params', default='db', choices=['db', 'cube'],
                    help='cube workload: dam_break_3d.py or cube.py parameter set')
    ap.add_argument('--hdx', type=float, default=0.0)
    ap.add_argument('--vary-m', type=float, default=0.0, dest='vary_m',
                    help='cube workload: m = m0 (1 +- vary_m U(-1,1)) (no uniform-mass records)')
    ap.add_argument('--vary-h', type=float, default=0.0, dest='vary_h',
                    help='cube workload: h = h0 (1 +- vary_h U(-1,1))')
    ap.add_argument('--dx', type=float, default=0.0087, help='dam_break spacing')
    ap.add_argument('--gamma', type=float, default=None,
                    help='cube / dam_break: exponent of the Tait EOS (default: the examples\' 7)')
    ap.add_argument('--dtype', default='f64', choices=['f64', 'f32'],
                    help='arithmetic type of the pair kernels')
    ap.add_argument('--variant', type=int, default=6)
    ap.add_argument('--ablate', type=int, default=0, help='profiling only')
    ap.add_argument('--overlap-halo', action='store_true', dest='overlap_halo',
                    help='slab runs whose evaluation can be split (one array, WCSPH): post the ghost transfers, run the '
                         'neighbour update and the interior wave tiles, then the ghosts and the face tiles '
                         '(default: the plain exchange -> neighbour update -> evaluation order; DESIGN.md section 6)')
    ap.add_argument('--rings-unperturbed', action='store_true', dest='rings_unperturbed',
                    help='--workload elastic: the state rings.py starts from (no stress, no particle in tension)')
    ap.add_argument('--opt', action='append', default=[], help='key=value library option')
    ap.add_argument('--no-reorder', action='store_true',
                    help='skip Solver.reorder_particles() before timing')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-check', action='store_true', help='skip the oracle parity check')
    ap.add_argument('--no-extras', action='store_true', help='skip the secondary configurations')
    ap.add_argument('--no-counters', action='store_true', dest='no_counters',
                    help='do not re-run under rocprofv3 --pmc for roofline.traffic (replay profiles/pmc_traffic.json)')
    ap.add_argument('--slab-axis', type=int, default=0, choices=(0, 1, 2), dest='slab_axis',
                    help='dam break: the axis the tank is cut along.  Measured on emulated ranks of the 17.3 M tank: cut along '
                         'y (whole rows of cells along x, 14 %% ghosts, every rank the same mix of arrays) an interior rank '
                         'takes 1.42 ms against 1.35 cut along x (rows of ~50 particles, 10 %% ghosts): x stays the default')
    ap.add_argument('--image-all-props', action='store_true', help='periodic images carry every device property (the reference\'s copy) instead of the evaluation\'s inputs')
    ap.add_argument('--cpu-n1', type=int, default=0, help='side of the CPU baseline sample (0: the workload\'s own size)')
    ap.add_argument('--fixed-bounds', action='store_true', dest='fixed_bounds',
                    help='hand nnps.update() the grid bounds and the (constant) h range instead of '
                         'reducing them every update: LinkedListNNPS(fixed_h=True) plus bounds a '
                         'stepping host knows from its own reductions; removes the min/max pass and '
                         'its device->host round trip from the step (reported in config)')
    ap.add_argument('--slab-weight-solid', type=float, default=1.0, dest='slab_weight_solid',
                    help='dam_break over several ranks: work of a boundary / obstacle particle relative to a fluid '
                         'particle when the slab faces are cut (1: equal particle counts)')
    ap.add_argument('--halo-protocol', default='padded', dest='halo_protocol', choices=['padded', 'capacity', 'handshake'],
                    help='ghost exchange of slab runs: padded (default) = fixed-capacity messages appended whole, padding '
                         'rows parked far away behind the ghosts, no device->host round trip; capacity = the same messages, the row counts '
                         'read back every exchange; handshake = counts all_gather before exactly sized messages')
    ap.add_argument('--emulate-rank', default=None, dest='emulate_rank', metavar='r/N',
                    help='dam_break on ONE GPU: rank r of an N-rank strong-scaling run, its slab plus the ghost layers '
                         'its neighbours would send, built locally (no communication): what a rank computes per step')
    ap.add_argument('--self-slab', action='store_true', dest='self_slab',
                    help='taylor_green on ONE GPU through the N>1 code path: the periodic x axis '
                         'is a slab whose two faces are exchanged with this rank itself over RCCL')
    args = ap.parse_args(argv)
    if args.emulate_rank:
        r, n = (int(v) for v in str(args.emulate_rank).split('/'))
        if not (0 <= r < n) or args.workload != 'dam_break':
            ap.error('--emulate-rank r/N needs 0 <= r < N and --workload dam_break')
        args.emulate_rank = (r, n)
    return args


def _free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_command(args_list, gpus, port=None):
    """the torchrun command line `bench.py --gpus N` re-executes itself under"""
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
            '--nproc-per-node', str(gpus), '--master-addr', '127.0.0.1',
            '--master-port', str(port or _free_port()),
            os.path.abspath(__file__)] + list(args_list)


def main():
    args = parse_args()
    # (the host driver of this pool supports dmabuf IPC only: RCCL between processes needs it whoever launched the ranks)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # start the ranks ourselves: one process per GPU, RCCL over xGMI
        env = dict(os.environ)
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        sys.exit(subprocess.call(launch_command(sys.argv[1:], args.gpus), env=env))
    import torch
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit('bench.py --gpus %d but WORLD_SIZE=%d: launch with '
                         '--nproc-per-node equal to --gpus' % (args.gpus, world))
    if os.environ.get('SPH_BENCH_DRYRUN') == '1':
        # launcher / rank plumbing only (tests/test_bench_launcher.py, CPU, gloo):
        # no particle is touched and nothing is measured
        import torch.distributed as dist
        dist.init_process_group('gloo', rank=rank, world_size=world)
        assert dist.get_world_size() == args.gpus
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t)
        dist.barrier()
        dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({'dryrun': True, 'n_gpus': world, 'rank_sum': float(t.item())}), flush=True)
        return
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU fallback)')
    torch.cuda.set_device(local_rank)
    dist = None
    # SPH_BENCH_FORCE_DIST=1: bring RCCL up even for one rank (checks the
    # process-group plumbing and the stdout ordering on a 1-GPU box)
    if world > 1 or args.self_slab or os.environ.get('SPH_BENCH_FORCE_DIST') == '1':
        import torch.distributed as dist
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world,
                                device_id=torch.device('cuda', local_rank))
        assert dist.get_world_size() == args.gpus
    out = run(args, rank, local_rank, world, dist)
    # The JSON line must be the LAST thing on stdout: RCCL's version banner
    # (NCCL_DEBUG=VERSION on the GPU boxes) sits in the C stdio buffer of every
    # rank until the process exits -- push it out before the final barrier,
    # tear the process group down, then print.
    import ctypes
    libc = ctypes.CDLL(None)
    libc.fflush(None)
    sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
        libc.fflush(None)
    if out is not None:
        print(json.dumps(out), flush=True)


def apply_options(args, ctx):
    ctx.set_option('pair_variant', args.variant)
    if args.ablate:
        ctx.set_option('ablate', args.ablate)
    if args.dtype == 'f32':
        ctx.set_option('arith_f32', 1)
    for kv in args.opt:
        k, v = kv.split('=')
        ctx.set_option(k, int(v))


def setup(args, w, rank, world, dist, ctx):
    """attach + push the arrays, build halo / domain / evaluator / NNPS"""
    from pysph_amd import device as dev
    from pysph_amd.acceleration_eval import AccelerationEval, SPHCompiler
    from pysph_amd.nnps import HipNNPS
    for a in w.arrays:
        dev.attach(a, ctx).push()       # everything resident in HBM
    halo = None
    if world > 1 or (args.self_slab and w.slab is not None and w.slab[2]):
        from pysph_amd.parallel import (ELASTIC_HALO_PROPS, SlabDecomposition,
                                        TVF_HALO_PROPS, WCSPH_HALO_PROPS)
        lo, hi, periodic, period = w.slab
        if os.environ.get('SPH_HALO_TRANSPORT', 'sphcomm') == 'sphcomm' and dist is not None and not hasattr(dist, 'hub'):
            # the point-to-point transfers straight on RCCL on the context's stream (libsphcomm.so: sph_comm_sendrecv),
            # the few collectives on torch.distributed: the process group's stream hand-over around its RCCL kernel cost a
            # slab rank of the 16 M dam break 0.09 ms per exchange.  SPH_HALO_TRANSPORT=torch: batch_isend_irecv.
            from pysph_amd.parallel import SphCommTransport, allreduce_scalars
            torch_dist, transport = dist, None
            try:
                transport = SphCommTransport(ctx, dist, rank, world)
            except Exception as e:
                sys.stderr.write('bench: libsphcomm transport unavailable on rank %d (%s)\n' % (rank, e))
            # every rank or none: the ranks of one exchange must speak through the same communicator
            import torch
            ok = allreduce_scalars([1.0 if transport is not None else 0.0], 'min', dist=torch_dist,
                                   device=torch.device('cuda', torch.cuda.current_device()) if torch_dist.get_backend() == 'nccl' else None)[0]
            if ok >= 1.0:
                dist = w.transport = transport
            else:
                if transport is not None:
                    transport.close()
                if rank == 0:
                    sys.stderr.write('bench: point-to-point transfers through torch.distributed\n')
        props = {'taylor_green': TVF_HALO_PROPS, 'elastic': ELASTIC_HALO_PROPS,
                 'elastic_block': ELASTIC_HALO_PROPS}.get(args.workload, WCSPH_HALO_PROPS)
        halo = SlabDecomposition(w.arrays, ctx, rank, world, axis=w.slab_axis,
                                 width=w.halo_width, lo=lo, hi=hi, props=props,
                                 periodic=periodic, period=period, dist=dist,
                                 protocol=os.environ.get('SPH_HALO_PROTOCOL', args.halo_protocol))
    domain = None
    if w.domain_kw is not None:
        from pysph_amd.domain import HipDomainManager
        domain = HipDomainManager(ctx=ctx, slab=halo, **w.domain_kw)
    a_eval = AccelerationEval(w.arrays, w.eqs, w.kernel)
    SPHCompiler(a_eval, ctx=ctx, sync='manual').compile()
    if domain is not None and not args.image_all_props:
        # an image carries what the evaluation reads (a dozen of a Taylor-Green particle's ~35 device properties)
        domain.set_image_props(a_eval.c_acceleration_eval.inputs)
    h_reduce = None
    if world > 1:
        from pysph_amd.parallel import allreduce_scalars
        dev_t = getattr(halo.halos[0].ops, 'device', None)   # where the transport's tensors live (as SlabDecomposition.rebalance)

        def h_reduce(lo, hi):
            # the known h range of a fixed_h run is the GLOBAL one: ghosts and
            # migrants bring the other ranks' smoothing lengths
            return (allreduce_scalars([lo], 'min', dist=dist, device=dev_t)[0],
                    allreduce_scalars([hi], 'max', dist=dist, device=dev_t)[0])
    # the overlapped exchange bins the real particles BEFORE the ghosts arrive: the cell size and the uniform-h
    # decision of that update must rest on the GLOBAL h range (the particles of this benchmark keep their h), as
    # sph_nnps_update_ghosts checks since round 5
    nnps = HipNNPS(3, w.arrays, radius_scale=w.kernel.radius_scale, ctx=ctx,
                   sync=False, domain=domain,
                   fixed_h=bool(args.fixed_bounds) or (halo is not None and args.overlap_halo),
                   h_range_reduce=h_reduce)
    a_eval.set_nnps(nnps)
    ordered = False
    if not args.no_reorder:
        # what the reference's Solver does for its GPU backends before the first
        # step and every 50 steps (solver.py:296-302, application.py:1157-1161):
        # put the particles in cell order so gathers/scatters coalesce (ghosts
        # are dropped for it and rebuilt by the first step)
        for i in range(len(w.arrays)):
            nnps.spatially_order_particles(i)
            if domain is not None:
                nnps.update_domain()
            elif halo is not None:
                halo.exchange()
            nnps.update()
        ordered = True
    if getattr(w, 'ghost_parts', None):
        # --emulate-rank: the ghost layers of the two neighbours, behind the (re-ordered) real particles
        for a, gh in zip(w.arrays, w.ghost_parts):
            if gh.get_number_of_particles():
                a.gpu.append_parray(gh, align=True)
        nnps.update()

    if args.fixed_bounds:
        # the particles of this benchmark do not move: the box of one full update
        # (ghosts included), padded by a cell, holds every later one
        if domain is not None:
            nnps.update_domain()
        elif halo is not None:
            halo.exchange()
        nnps.update()
        pad = nnps.cell_size
        nnps.bounds = tuple(nnps.xmin - pad) + tuple(nnps.xmax + pad)

    # Slab runs whose evaluation can be split around the arrival of the ghosts (one particle array, hand-written
    # kernels, pair loops over real particles: the cube) overlap the exchange with the neighbour update of the real
    # particles and the interior part of the pair loops
    ce = a_eval.c_acceleration_eval
    overlap = (halo is not None and domain is None and args.overlap_halo
               and hasattr(halo, 'exchange_begin') and ce.can_split())
    if overlap:
        nnps.set_extend(w.halo_width, 0.0, 0.0)
    w.overlap_halo = bool(overlap)

    def step():
        if domain is not None:
            nnps.update_domain()        # periodic (and, with a slab, remote) ghosts rebuilt every step
        elif overlap:
            st = halo.exchange_begin()
            if st is not None:          # transfers posted: everything that needs no ghosts runs now
                nnps.set_ghost_faces(0, *halo.faces())
                nnps.update()
                ce.compute_begin(0.0, 1e-5)
                halo.exchange_finish(st)
                nnps.update_ghosts(0, *halo.faces())
                ce.compute_end(0.0, 1e-5)
                return
        elif halo is not None:
            halo.exchange()
        if overlap:
            nnps.set_ghost_faces(-1)    # (an exchange that completed in one piece: the plain order)
        nnps.update()
        a_eval.compute(0.0, 1e-5)
        # the round-trip-free exchange learns its ghost counts HERE, with the evaluation queued (the host waits for the
        # transfers only); a face that had outgrown its message was repeated the counted way and the evaluation runs
        # again (as Integrator.compute_accelerations does; never on a benchmark whose particles do not move)
        while not all([v.verify() for v in (halo, domain) if v is not None]):
            nnps.update()
            a_eval.compute(0.0, 1e-5)
    return nnps, a_eval, halo, domain, step, ordered


SETTLE_MS = 100.0    # load after which the clock has settled: the warm-up of the extras, never of the headline


def timed(steps, warmup, step, barrier, ctx, breakdown_steps=None, settle_ms=0.0):
    """The per-class breakdown, W warm-up steps, then EXACTLY `steps` timed ones between barrier + synchronize.  In the
    timed region the library times the PAIR launches only (HIP events on the launch stream: roofline.achieved comes from
    them); the per-class breakdown (nnps / pack / eos, the pair families) is taken in `steps` extra steps with every class
    timed -- the event markers around each region cost the stream ~5 us each, eight per step were 2 % of the headline step
    (round 5 timed every class inside the timed region).  Round 6 takes those steps BEFORE the warm-up instead of behind
    the timed region: the GPU's clock needs ~40 steps of load (80 ms) to settle after the idle seconds of the set-up and
    the CPU oracle -- the same 20 timed steps take 2.11 ms each behind 5 warm-up steps, 2.06 behind 11, 2.04 behind 20, 2.035 behind 40
     --, and steps that have to run anyway may as well run there.  Their class figures carry that
    ramp (a few per cent high); the run's very first step, which allocates inside the timed classes, is not among them."""
    ctx.timer_enable(1)
    step()                                           # the first step of a run allocates (outputs, record buffers, tables)
    if settle_ms:
        barrier()
    t_begin = time.perf_counter()
    ctx.timer_reset()                                # inside the timed classes: not part of the breakdown
    nb = max(1, min(breakdown_steps or steps, steps))   # as many steps as the timed region: the class figures are not extrapolated
    for _ in range(nb):
        step()
    barrier()
    scale = float(steps) / nb                        # (reported per `steps`, like the pair figure)
    timers = {k: (ctx.timer_get(k)[0] * scale, int(round(ctx.timer_get(k)[1] * scale)))
              for k in ('nnps', 'pack', 'eos') + PAIR_FAMILIES}
    ctx.timer_enable(2)
    for _ in range(warmup):
        step()
    if settle_ms:                                    # (extras only: steady-state figures whatever the size of the step)
        barrier()
        while (time.perf_counter() - t_begin) * 1e3 < settle_ms:
            for _ in range(4):
                step()
            barrier()
    ctx.timer_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    timers['pair'] = ctx.timer_get('pair')
    timers['n_async'] = ctx.timer_get('n_async')     # neighbour updates that made no device->host round trip
    ctx.timer_enable(0)
    return elapsed, timers


def run(args, rank, local_rank, world, dist):
    """One rank of the benchmark; `dist` is torch.distributed (RCCL) or, in the
    one-GPU test of the N>1 code path, an in-process stand-in with the same
    calls (tests/helpers.ThreadDist).  Returns the JSON dict on rank 0."""
    import torch

    from pysph_amd import device as dev

    tstream = torch.cuda.Stream()      # kernels, copies and RCCL share it
    torch.cuda.set_stream(tstream)
    ctx = dev.HipContext(local_rank, tstream.cuda_stream)
    apply_options(args, ctx)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    extra = {}
    if world > 1 and hasattr(dist, 'all_gather'):
        # 1-vs-N parity of a small instance before anything is timed
        extra['parity_1_vs_n_ranks_max_rel'] = multi_rank_parity(
            args, rank, local_rank, world, dist, tstream)

    w = build_workload(args, rank, world)
    n_local = sum(a.get_number_of_particles() for a in w.arrays)
    nnps, a_eval, halo, domain, step, ordered = setup(args, w, rank, world, dist, ctx)
    host_in = None
    if rank == 0 and world == 1 and not args.no_check:
        # inputs as the device holds them now (after the spatial reordering)
        for a in w.arrays:
            a.gpu.pull()
        host_in = copy_arrays(w.arrays)
    elapsed, timers = timed(args.steps, args.warmup, step, barrier, ctx)
    extra['halo_transport'] = None if halo is None else (
        'libsphcomm (ncclSend/Recv on the context stream)' if getattr(w, 'transport', None) is not None else 'torch.distributed')
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # true neighbour pairs of one evaluation (outside the timed region): the
    # flop-side reading of the pair loop that SURVEY 8(d) asks for next to the
    # HBM one
    pairs = None
    if args.workload == 'cube' and rank == 0:
        pairs = nnps.count_neighbors(0, 0)
    pair_ms, pair_launches = timers['pair']
    n_total = n_local * world          # real particles only (ghosts are extra work)
    if w.scaling == 'strong' and dist is not None:
        tn = torch.tensor([float(n_local)], dtype=torch.float64, device='cuda')
        dist.all_reduce(tn, op=dist.ReduceOp.SUM)
        n_total = int(tn.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = n_total * args.steps / elapsed
    if getattr(w, 'transport', None) is not None:
        # the library's own RCCL communicator goes while every rank is still here (before torch's process group)
        torch.cuda.synchronize()
        ctx.__dict__.pop('_transport', None)
        w.transport.close()
        w.transport = None
    if rank != 0:
        return None

    # HBM bytes per launch of the dominant kernel come from a separate
    # rocprofv3 --pmc run of this same command (profiles/); only quoted when
    # the configuration matches the profiled one, else null
    traffic, traffic_source, l1_fill, traffic_box_ms = None, None, None, None
    try:
        pt = json.load(open(os.path.join(REPO, 'profiles', 'pmc_traffic.json')))
        c = pt['config']
        if (c['n1'], c['variant'], c['spatially_ordered'], c.get('workload', 'cube'),
                c.get('dtype', 'f64')) == \
                (args.n1, args.variant, ordered, args.workload, args.dtype) \
                and world == 1 and args.params == 'db' and not args.vary_h:
            traffic = pt['bytes_per_launch']
            l1_fill = pt.get('l1_fill_bytes_per_launch')
            traffic_box_ms = pt.get('profiled_box_kernel_ms')
            traffic_source = pt.get('source', 'profiles/pmc_traffic.json: separate rocprofv3 --pmc '
                                              'passes of this command, not measured in this run')
    except Exception:
        traffic = None
    # ... unless rocprofv3 is on this box: then the counters are taken in THIS run (three short child runs)
    measured_kernel = None
    if world == 1 and not args.no_counters and not args.ablate:
        wl_flags = ['--workload', args.workload, '--n1', str(args.n1), '--dtype', args.dtype, '--params', args.params,
                    '--dx', repr(args.dx), '--variant', str(args.variant)]
        if args.vary_h:
            wl_flags += ['--vary-h', repr(args.vary_h)]
        if args.gamma:
            wl_flags += ['--gamma', repr(args.gamma)]
        if args.no_reorder:
            wl_flags += ['--no-reorder']
        for kv in args.opt:
            wl_flags += ['--opt', kv]
        cm = counters_in_this_run(wl_flags)
        if cm:
            traffic, l1_fill = cm['bytes_per_launch'], cm['l1_fill_bytes_per_launch']
            traffic_box_ms, measured_kernel = cm['kernel_ms_under_counters'], cm['kernel']
            traffic_source = ('measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCP_TCC_READ_REQ_sum '
                              '--kernel-trace, one 3-step child run of this command per counter set on this box, '
                              'per-launch mean of this kernel; FETCH_SIZE x 2 (gfx950: 64 B counted per 128-B request, '
                              'MI355X_MICROARCH.md) + WRITE_SIZE')
    pair_avg_s = pair_ms / max(pair_launches, 1) * 1e-3
    # several pair launches per step for multi-destination sets: bytes of ONE
    # step / total pair-kernel time of one step
    pair_step_s = pair_ms / args.steps * 1e-3
    bytes_scale = 0.5 if args.dtype == 'f32' else 1.0
    algo_pair = w.algo_pair * bytes_scale
    n_first = w.arrays[0].get_number_of_particles(True)
    algo_bytes = algo_pair * n_first + w.algo_solid * bytes_scale * (n_local - n_first)
    achieved = algo_bytes / pair_step_s / 1e9 if pair_step_s > 0 else 0.0
    fam = {'cube': 'FamWCSPH', 'dam_break': 'FamWCSPH', 'taylor_green': 'FamDensity+FamTVF',
           'elastic': 'FamVGrad+FamElastic', 'elastic_block': 'FamVGrad+FamElastic'}[args.workload]
    valu_peak = FP64_VECTOR_PEAK_TFLOPS * (2.0 if args.dtype == 'f32' else 1.0)
    out = {
        'metric': 'particle-updates/sec (nnps.update + AccelerationEval.compute), '
                  '%s 3D, %s' % ({'cube': 'WCSPH', 'dam_break': 'WCSPH', 'taylor_green': 'TVF',
                                  'elastic': 'elastic', 'elastic_block': 'elastic'}[args.workload],
                                 'fp64' if args.dtype == 'f64' else 'fp32'),
        'value': value, 'unit': 'particle-updates/s', 'n_gpus': world,
        'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': ms_per_step, 'higher_is_better': True,
        'scaling': w.scaling, 'vs_baseline': None, 'dtype': args.dtype,
        'data': 'synthetic',
        'config': {
            'workload': w.name,
            'particles_per_gpu': n_local, 'pair_variant': args.variant,
            'spatially_ordered': ordered,
            'parallelism': 'slab%d' % world if world > 1 else ('slab1-self-exchange' if halo is not None else 'single'),
            'rccl_ranks': world if dist is not None else 0,
            'fixed_bounds': bool(args.fixed_bounds),
            'halo_overlap': bool(getattr(w, 'overlap_halo', False)),
        },
        'roofline': {
            # the kernel's name as rocprofv3 prints it when the counters were taken in this run
            'bound': 'hbm', 'kernel': measured_kernel or 'k_pair_%s<%s,%s>' % (
                {0: 'direct', 2: 'wg', 3: 'agg', 6: 'wave'}[args.variant], fam,
                type(w.kernel).__name__),
            'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
            'traffic_source': traffic_source,
            # the kernel's mean duration in the run the counters come from (a child run of this command on this box, or --
            # replayed counters -- another box of the pool), next to this run's avg_kernel_ms
            'traffic_profiled_kernel_ms': traffic_box_ms,
            'algorithmic_bytes_per_particle': algo_pair,
            'avg_kernel_ms': pair_avg_s * 1e3,
            'pair_kernel_ms_per_step': pair_step_s * 1e3,
            # what the kernel is actually bound by: the per-lane record
            # gathers re-fetch their 128-B lines from the L2 into the CUs' 32-KB L1s
            'l2_to_l1': None if not l1_fill or pair_avg_s <= 0 else {
                'bytes_per_launch': l1_fill, 'achieved': l1_fill / pair_avg_s / 1e12,
                'peak': L2_PEAK_TBS, 'unit': 'TB/s', 'frac': l1_fill / pair_avg_s / 1e12 / L2_PEAK_TBS,
                'frac_on_profiled_box': None if not traffic_box_ms else l1_fill / (traffic_box_ms * 1e-3) / 1e12 / L2_PEAK_TBS,
                'source': 'TCP_TCC_READ_REQ x 128 B of the run the counters come from (roofline.traffic_source), this run\'s kernel time'},
        },
        'kernel_ms_per_step': {k: v[0] / args.steps for k, v in timers.items() if k not in PAIR_FAMILIES and k != 'n_async'},
        'nnps_updates_without_round_trip': timers['n_async'][1],
        'periodic_updates_without_round_trip': None if domain is None else int(getattr(domain, 'padded_updates', 0)),
        'pair_ms_per_family': family_ms(timers, args.steps),
        'fp64_valu': None if not pairs else {
            'pairs_per_launch': pairs,
            'flop_per_pair': FLOP_PER_PAIR,
            'achieved': pairs * FLOP_PER_PAIR / pair_step_s / 1e12,
            'peak': valu_peak, 'unit': 'TFLOP/s',
            'frac': pairs * FLOP_PER_PAIR / pair_step_s / 1e12 / valu_peak,
            'Gpairs_per_s': pairs / pair_step_s / 1e9},
        'algorithmic_GBs_whole_update': ALGO_BYTES_UPDATE * bytes_scale * value / 1e9,
    }
    if host_in is not None:
        tol = PARITY_TOL if args.dtype == 'f64' else 5e-5   # fp32: tests/test_hip_parity.py::test_fp32_arithmetic_vs_golden
        extra.update(parity_check(w, host_in, nnps, domain, tol))
    if world == 1 and not args.no_extras and args.workload == 'cube' \
            and args.params == 'db' and not args.vary_h and args.n1 == 159 \
            and args.dtype == 'f64' and not args.ablate:
        del nnps, a_eval, step      # free this workload's device state first (252^3 needs room)
        extra['secondary'] = secondary_runs(args, local_rank, tstream)
        extra['step_vs_n'] = step_vs_n(args, local_rank, tstream)
        t1 = extra['secondary'].get('C4 dam break dx 0.0035 (16 M) on one GPU', {}).get('ms_per_step')
        extra['projected_strong_scaling_8'] = projected_strong_scaling(args, local_rank, tstream, t_one_gpu_ms=t1)
        extra['time_stepping'] = time_stepping(local_rank, tstream)
    if not args.no_cpu_baseline and world == 1:
        # on the size of the reported workload itself (159^3 for the headline line: ~1.1 s per pass on the box's cores);
        # --cpu-n1 overrides the sample
        n_cpu = args.cpu_n1 or (args.n1 if args.workload == 'cube' else 100)
        out['cpu_baseline'] = cpu_baseline(n_cpu, target_seconds=12.0)
        if not args.no_extras and args.workload == 'cube' and n_cpu != 100:
            b1 = cpu_baseline(100, target_seconds=4.0)      # (the 1 M sample of the earlier rounds' lines)
            extra['cpu_baseline_100'] = {k: b1[k] for k in ('value', 'cores', 'sample')}
    if extra:
        out['extra'] = extra
    return out


def time_stepping(local_rank, tstream, dx=0.0087, n_steps=30):
    """The callers either side of the path: whole EPEC time steps
    of the C2 dam break, everything device-resident -- per step two
    nnps.update() + compute() evaluations, three WCSPHStep stage sweeps and the
    adaptive time step from device reductions (pysph/sph/integrator.py:161-200,
    401-420), particles re-ordered into cell order every 50 steps as the
    reference's Solver does for its GPU backends.  Reported next to the
    headline, not part of it."""
    import torch
    from pysph_amd import device as dev
    from pysph_amd.examples import dam_break_3d as db
    ctx = dev.HipContext(local_rank, tstream.cuda_stream)
    try:
        db.run(dx=dx, n_steps=3, ctx=ctx)                      # warm-up (buffers, generated code)
        ctx.close()
        ctx = dev.HipContext(local_rank, tstream.cuda_stream)
        arrays, st = db.run(dx=dx, n_steps=n_steps, ctx=ctx)
        n_fluid = arrays[0].get_number_of_particles()
        return {'workload': 'C2 dam break dx %g, EPEC + WCSPHStep, adaptive dt, %d steps from rest' % (dx, n_steps),
                'particles': st['particles'], 'fluid_particles': n_fluid,
                'ms_per_time_step': st['wall_s'] / st['steps'] * 1e3,
                'time_steps_per_s': st['steps_per_s'],
                'particle_steps_per_s': st['particle_steps_per_s'],
                'simulated_time': st['t']}
    except Exception as e:
        return {'error': '%s: %s' % (type(e).__name__, e)}
    finally:
        ctx.close()
        torch.cuda.empty_cache()


def counters_in_this_run(argv_workload, timeout_s=240):
    """HBM-side bytes of the dominant pair kernel MEASURED IN THIS RUN: bench.py
    re-executes itself (same workload flags, 3 steps) under `rocprofv3 --pmc <one
    set> --kernel-trace`, one child per counter set as the guide prescribes
    (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass; --pmc only
    with --kernel-trace), and reads the per-dispatch counters of the pair kernel
    with the largest total duration.  gfx950 correction of the same guide:
    FETCH_SIZE counts 64 B per 128-B request of a wide coalesced read -> x 2;
    WRITE_SIZE as reported (both in KiB).  Returns None when rocprofv3 is not on
    the box or a pass fails -- the caller then replays profiles/pmc_traffic.json
    and says so."""
    import csv
    import glob
    import shutil
    import tempfile
    if os.environ.get('SPH_BENCH_CHILD') or not shutil.which('rocprofv3'):
        return None
    if any(k.startswith(('ROCPROF', 'ROCP_')) for k in os.environ):   # this run is being profiled itself
        return None
    env = dict(os.environ, SPH_BENCH_CHILD='1', TMPDIR='/tmp')
    child = [sys.executable, os.path.join(REPO, 'bench.py'), '--no-cpu-baseline', '--no-check', '--no-extras',
             '--steps', '3', '--warmup', '1'] + argv_workload
    vals, kname, kms = {}, None, None
    for cset in ('FETCH_SIZE', 'WRITE_SIZE', 'TCP_TCC_READ_REQ_sum'):
        d = tempfile.mkdtemp(prefix='sph_pmc_', dir='/tmp')
        try:
            subprocess.run(['rocprofv3', '--pmc', cset, '--kernel-trace', '--output-format', 'csv', '-d', d, '-o', 'run', '--'] + child,
                           cwd='/tmp', env=env, timeout=timeout_s, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
            f = glob.glob(os.path.join(d, '**', '*_counter_collection.csv'), recursive=True)
            per = {}                       # kernel -> {dispatch: value}
            for r in csv.DictReader(open(f[0])):
                if 'k_pair_wave' in r['Kernel_Name'] and 'FamNbr' not in r['Kernel_Name'] and r['Counter_Name'] == cset:
                    per.setdefault(r['Kernel_Name'], {}).setdefault(r['Dispatch_Id'], 0.0)
                    per[r['Kernel_Name']][r['Dispatch_Id']] += float(r['Counter_Value'])
            if not per:
                return None
            # the kernel of the steady state: the one with the most dispatches (warm-up launches of other layouts are fewer)
            k = max(per, key=lambda q: len(per[q]))
            v = list(per[k].values())
            vals[cset] = sum(v) / len(v)
            kname = kname or k
            tr = glob.glob(os.path.join(d, '**', '*_kernel_trace.csv'), recursive=True)
            if tr and kms is None:
                du = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-6
                      for r in csv.DictReader(open(tr[0])) if r['Kernel_Name'] == k]
                kms = sum(du) / len(du) if du else None
        except Exception:
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return {'kernel': kname, 'bytes_per_launch': vals['FETCH_SIZE'] * 1024.0 * 2.0 + vals['WRITE_SIZE'] * 1024.0,
            'fetch_bytes_corrected': vals['FETCH_SIZE'] * 1024.0 * 2.0, 'write_bytes': vals['WRITE_SIZE'] * 1024.0,
            'l1_fill_bytes_per_launch': vals['TCP_TCC_READ_REQ_sum'] * 128.0, 'kernel_ms_under_counters': kms}


def family_ms(timers, steps):
    """pair-kernel ms per step of every equation family that ran"""
    return {k[5:]: v[0] / steps for k, v in timers.items()
            if k in PAIR_FAMILIES and v[1] > 0}


def secondary_runs(args, local_rank, tstream):
    """The other BASELINE configurations AS SPECIFIED, each
    measured in this same invocation and checked against the oracle on the
    state it ran on -- C2 dam break (dx 0.0087, 1 M fluid), the dam break at
    dx 0.0055 (4.06 M fluid: the size BASELINE's `metric` names), C3
    Taylor-Green 159^3 periodic TVF, C5 S-rings3d 2 M in fp32 (as named) and
    fp64 --, then the timing-only variants of the headline cube BASELINE.md
    section 4 quotes (other sizes, unsorted, variable h, cube.py parameters).
    Every checked entry carries ms_per_step, the pair-kernel ms per equation
    family, its own algorithmic-bytes roofline fraction and parity_max_rel."""
    import copy
    import torch
    from pysph_amd import device as dev
    res = {}
    cases = [
        ('C2 dam break dx 0.0087', dict(workload='dam_break', dx=0.0087), True),
        ('dam break dx 0.0055 (4 M fluid)', dict(workload='dam_break', dx=0.0055), True),
        # the same tank with every particle carrying its own h: the merged one-launch path on variable-h records (round 5)
        ('dam break dx 0.0055, h +-15 %', dict(workload='dam_break', dx=0.0055, vary_h=0.15), True),
        # BASELINE config 4's workload on ONE GPU: the N = 1 anchor of its 8-GPU strong-scaling number
        ('C4 dam break dx 0.0035 (16 M) on one GPU', dict(workload='dam_break', dx=0.0035), True),
        ('C3 Taylor-Green 159^3 TVF', dict(workload='taylor_green', n1=159), True),
        ('C5 S-rings3d 2 M fp32', dict(workload='elastic', dtype='f32'), True),
        ('C5 S-rings3d 2 M fp64', dict(workload='elastic'), True),
        # TIMING ONLY: at rest every output field of this state is (almost) identically zero, a parity figure would be vacuous
        ('C5 S-rings3d 2 M fp32 from rings.py\'s initial state (timing only; no stress yet: r_ij = 0, not gathered)',
         dict(workload='elastic', dtype='f32', rings_unperturbed=True), False),
        ('100^3', dict(n1=100), False),
        ('252^3', dict(n1=252), False),
        ('159^3 unsorted', dict(no_reorder=True), False),
        ('159^3 h +-15%', dict(vary_h=0.15), False),
        ('159^3 cube.py parameters (CubicSpline, hdx 1.5, alpha 0.5, c0 10)', dict(params='cube'), False),
    ]
    for name, kw, check in cases:
        a2 = copy.copy(args)
        steps, warmup = max(3, args.steps // 4), 2
        for k, v in kw.items():
            setattr(a2, k, v)
        ctx = dev.HipContext(local_rank, tstream.cuda_stream)
        apply_options(a2, ctx)
        try:
            w = build_workload(a2, 0, 1)
            nnps, a_eval, halo, domain, step, ordered = setup(a2, w, 0, 1, None, ctx)
            host_in = None
            if check:
                for a in w.arrays:
                    a.gpu.pull()
                host_in = copy_arrays(w.arrays)
            elapsed, timers = timed(steps, warmup, step, torch.cuda.synchronize, ctx, settle_ms=SETTLE_MS)
            n = sum(a.get_number_of_particles(True) for a in w.arrays)
            n_fluid = w.arrays[0].get_number_of_particles(True)
            pair_step_s = timers['pair'][0] / steps * 1e-3
            scale = 0.5 if a2.dtype == 'f32' else 1.0
            algo = w.algo_pair * scale
            algo_bytes = algo * n_fluid + w.algo_solid * scale * (n - n_fluid)
            achieved = algo_bytes / pair_step_s / 1e9 if pair_step_s > 0 else 0.0
            r = {'workload': w.name, 'dtype': a2.dtype, 'particles': n,
                 'ms_per_step': elapsed / steps * 1e3,
                 'particle_updates_per_s': n * steps / elapsed,
                 'pair_ms_per_step': timers['pair'][0] / steps,
                 'pair_ms_per_family': family_ms(timers, steps),
                 'kernel_ms_per_step': {k: timers[k][0] / steps for k in ('nnps', 'pack', 'eos', 'pair')},
                 'roofline': {'algorithmic_bytes_per_particle': algo,
                              'algorithmic_bytes_per_evaluation': algo_bytes,
                              'achieved_GBs': achieved, 'frac': achieved / HBM_PEAK_GBS},
                 'steps': steps}
            if check:
                tol = PARITY_TOL if a2.dtype == 'f64' else 5e-5
                pc = parity_check(w, host_in, nnps, domain, tol)
                for k in ('parity_max_rel', 'parity_worst_field', 'parity_elementwise_max_rel',
                          'parity_elementwise_worst_field', 'parity_tolerance', 'parity_elementwise_tolerance',
                          'parity_neighbour_count_mismatches', 'parity_ok', 'parity_oracle_seconds'):
                    r[k] = pc[k]
            res[name] = r
            del nnps, a_eval, step, w, host_in
        except Exception as e:         # a secondary number must not lose the headline
            res[name] = {'error': '%s: %s' % (type(e).__name__, e)}
        finally:
            ctx.close()
            torch.cuda.empty_cache()
    return res


def step_vs_n(args, local_rank, tstream, sides=(63, 79, 100, 126, 159)):
    """ms per step and per class (nnps / pack / eos / pair) of the headline cube at
    0.25 / 0.5 / 1 / 2 / 4 M particles: what is left of a step when the pair kernel
    shrinks is the latency-bound bookkeeping around it (round 5: the hand-written
    sort, no device->host round trip) -- and what a rank of a strong-scaling run
    pays.  Timing only (the sizes with a parity check are in extra.secondary)."""
    import copy
    import torch
    from pysph_amd import device as dev
    rows = {}
    for n1 in sides:
        a2 = copy.copy(args)
        a2.n1 = n1
        ctx = dev.HipContext(local_rank, tstream.cuda_stream)
        apply_options(a2, ctx)
        try:
            w = build_workload(a2, 0, 1)
            nnps, a_eval, halo, domain, step, ordered = setup(a2, w, 0, 1, None, ctx)
            steps = 10
            elapsed, timers = timed(steps, 3, step, torch.cuda.synchronize, ctx, settle_ms=SETTLE_MS)
            n = sum(a.get_number_of_particles(True) for a in w.arrays)
            rows['%d^3' % n1] = {'particles': n, 'ms_per_step': elapsed / steps * 1e3,
                                 'particle_updates_per_s': n * steps / elapsed,
                                 'kernel_ms_per_step': {k: timers[k][0] / steps for k in ('nnps', 'pack', 'eos', 'pair')},
                                 'updates_without_round_trip': timers['n_async'][1]}
            del nnps, a_eval, step, w
        except Exception as e:
            rows['%d^3' % n1] = {'error': '%s: %s' % (type(e).__name__, e)}
        finally:
            ctx.close()
            torch.cuda.empty_cache()
    return rows


def projected_strong_scaling(args, local_rank, tstream, world=8, dx=0.0035, t_one_gpu_ms=None):
    """BASELINE config 4 (the 16 M dam break over 8 GPUs) projected from ONE GPU:
    every rank's slab + the ghost layers its neighbours would send is built
    locally (`--emulate-rank r/8`) and timed (nnps.update + compute: what that
    rank computes per step), the exchange that delivers the ghosts is taken from
    the slab transport measured on this GPU (`--self-slab` against the plain step
    at a rank's particle count: select + pack + RCCL send/recv to itself + header
    round trip + append).  projected speed-up = t(1 GPU) / max_r (t_r + exchange).
    A PROJECTION, not a measurement: no byte crosses xGMI here."""
    import copy
    import torch
    from pysph_amd import device as dev
    out = {'world': world, 'dx': dx, 'slab_weight_solid': args.slab_weight_solid, 'halo_protocol': args.halo_protocol, 'ranks': {}}
    t_max, n_max = 0.0, 0
    for r in range(world):
        a2 = copy.copy(args)
        a2.workload, a2.dx, a2.emulate_rank = 'dam_break', dx, (r, world)
        ctx = dev.HipContext(local_rank, tstream.cuda_stream)
        apply_options(a2, ctx)
        try:
            w = build_workload(a2, 0, 1)
            nnps, a_eval, halo, domain, step, ordered = setup(a2, w, 0, 1, None, ctx)
            steps = 10
            elapsed, timers = timed(steps, 3, step, torch.cuda.synchronize, ctx, settle_ms=SETTLE_MS)
            nreal = sum(a.get_number_of_particles(True) for a in w.arrays)
            nall = sum(a.get_number_of_particles() for a in w.arrays)
            ms = elapsed / steps * 1e3
            # what this rank RECEIVES per exchange, face by face and array by array, as the round-trip-free protocol sends
            # it in its steady state: rows = the capacity of a steady face (pysph_amd.parallel._capacity_tight: the count
            # + 1/16 + 1024, in units of 1024), 7 doubles per row (x y z u v w rho: the promised-uniform h and m do not
            # travel) + the header.  Round 5 sent 9 doubles x (count + 1/4 + 4096) rows.
            from pysph_amd.parallel import _capacity, _capacity_tight
            faces = {}
            for a in w.arrays:
                nr, nt = a.get_number_of_particles(True), a.get_number_of_particles()
                gx = np.asarray(a.x[nr:nt])
                mid = 0.5 * (float(np.min(a.x[:nr])) + float(np.max(a.x[:nr]))) if nr else 0.0
                cnt = [int(np.sum(gx < mid)), int(np.sum(gx >= mid))]
                faces[a.name] = {'ghosts': cnt,
                                 'message_bytes': [(_capacity_tight(c) * 7 + 1) * 8 if c or (r > 0 if s == 0 else r < world - 1) else 0
                                                   for s, c in enumerate(cnt)],
                                 'message_bytes_round5': [(_capacity(c) * 9 + 1) * 8 if c or (r > 0 if s == 0 else r < world - 1) else 0
                                                          for s, c in enumerate(cnt)]}
            out['ranks'][str(r)] = {'real_particles': nreal, 'ghost_particles': nall - nreal, 'ms_per_step': ms,
                                    'real_per_array': {a.name: a.get_number_of_particles(True) for a in w.arrays},
                                    'received_per_face': faces,
                                    'bytes_per_face': [sum(f['message_bytes'][s] for f in faces.values()) for s in (0, 1)],
                                    'bytes_per_face_round5': [sum(f['message_bytes_round5'][s] for f in faces.values()) for s in (0, 1)],
                                    'kernel_ms_per_step': {k: timers[k][0] / steps for k in ('nnps', 'pack', 'eos', 'pair')}}
            if ms > t_max:
                t_max, n_max = ms, nreal
            del nnps, a_eval, step, w
        except Exception as e:
            out['ranks'][str(r)] = {'error': '%s: %s' % (type(e).__name__, e)}
        finally:
            ctx.close()
            torch.cuda.empty_cache()
    _DAM_CACHE.clear()
    # the exchange, LOCAL part: an interior rank of the same tank whose ghosts come through the slab transport itself (the
    # rank as its own periodic neighbour: the selection + packing of every array, the RCCL send / recv to itself, the
    # appends, on the round-trip-free protocol) against the same rank with its ghosts in place.  Round 5 took this
    # figure from a one-array cube of a rank's size; the dam break's three arrays make three times the launches.
    ex = {}
    own_group = False
    r_mid = world // 2
    try:
        import torch.distributed as dist
        if not dist.is_initialized():
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', str(_free_port()))
            dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', local_rank))
            own_group = True
        a2 = copy.copy(args)
        a2.workload, a2.dx, a2.emulate_rank, a2.self_slab = 'dam_break', dx, (r_mid, world), True
        ctx = dev.HipContext(local_rank, tstream.cuda_stream)
        apply_options(a2, ctx)
        try:
            w = build_workload(a2, 0, 1)
            nnps, a_eval, halo, domain, step, ordered = setup(a2, w, 0, 1, dist, ctx)
            elapsed, timers = timed(10, 4, step, torch.cuda.synchronize, ctx, settle_ms=SETTLE_MS)
            ex['self_slab'] = elapsed / 10 * 1e3
            ex['self_slab_kernels'] = sum(timers[k][0] for k in ('nnps', 'pack', 'eos', 'pair')) / 10
            ex['padded_exchanges'] = [h.padded_exchanges for h in halo.halos]
            ex['ghosts'] = [a.gpu.get_number_of_particles() - a.gpu.get_number_of_particles(True) for a in w.arrays]
            ex['properties_per_ghost'] = [int(h.ops.nprops) for h in halo.halos]
            del nnps, a_eval, step, w, halo
        finally:
            ctx.close()
            torch.cuda.empty_cache()
    except Exception as e:
        ex['self_slab'] = None
        ex['self_slab_error'] = '%s: %s' % (type(e).__name__, e)
    _DAM_CACHE.clear()
    if own_group:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    ex['plain'] = out['ranks'].get(str(r_mid), {}).get('ms_per_step')
    ex['plain_kernels'] = sum(out['ranks'].get(str(r_mid), {}).get('kernel_ms_per_step', {}).values()) or None
    out['exchange_stand_in'] = {'rank': r_mid, 'what': 'dam break dx %g, rank %d of %d as its own periodic neighbour over RCCL' % (dx, r_mid, world),
                                'ms_per_step': ex}
    if ex.get('plain') and ex.get('self_slab'):
        # what the slab transport adds to the step of a rank that already holds (and computes with) its ghosts
        out['exchange_ms'] = max(ex['self_slab'] - ex['plain'], 0.0)
    out['slowest_rank_ms'] = t_max
    if t_one_gpu_ms and t_max > 0 and out.get('exchange_ms') is not None:
        out['t_one_gpu_ms'] = t_one_gpu_ms
        out['projected_speedup_free_exchange'] = t_one_gpu_ms / t_max
        # The stand-in's "transfer" is a device copy; a real face crosses ONE xGMI link per direction (send and receive at
        # the same time, the two faces of a rank on two different links: the slower face sets the time).  The plain order
        # does not hide it: step = rank + local exchange work (selection, packing, RCCL launch, append: the stand-in's
        # figure) + bytes of the larger face / link rate + a link latency.  MI355X_MICROARCH.md: 153.6 GB/s peak per
        # link and direction; RCCL point-to-point reaches 45-75 GB/s of it on messages of a few MB.
        LINK_LATENCY_MS = 0.010
        ok = [v for v in out['ranks'].values() if 'ms_per_step' in v]
        face_bytes = max(max(v['bytes_per_face']) for v in ok)
        face_bytes5 = max(max(v['bytes_per_face_round5']) for v in ok)
        out['largest_face_message_bytes'] = face_bytes
        out['largest_face_message_bytes_round5_protocol'] = face_bytes5
        out['link_model'] = 'step = max_r(rank_ms) + exchange_ms (measured on this GPU: select + pack + RCCL launch + append of all three arrays) + ' \
                            'largest face bytes / rate + %.0f us; nothing overlapped' % (LINK_LATENCY_MS * 1e3)
        out['projected_speedup_with_link'] = {}
        for gbs in (45, 60, 75):
            t_link = face_bytes / (gbs * 1e9) * 1e3 + LINK_LATENCY_MS
            out['projected_speedup_with_link']['%d GB/s' % gbs] = {
                'transfer_ms': t_link, 'step_ms': t_max + out['exchange_ms'] + t_link,
                'speedup': t_one_gpu_ms / (t_max + out['exchange_ms'] + t_link),
                'speedup_round5_messages': t_one_gpu_ms / (t_max + out['exchange_ms'] + face_bytes5 / (gbs * 1e9) * 1e3 + LINK_LATENCY_MS)}
        # the headline of the projection: 60 GB/s per direction
        out['projected_speedup'] = out['projected_speedup_with_link']['60 GB/s']['speedup']
        out['projected_speedup_stand_in_transfer'] = t_one_gpu_ms / (t_max + out['exchange_ms'])
    return out


def multi_rank_parity(args, rank, local_rank, world, dist, tstream):
    """Before timing, N > 1: a small instance of the SAME workload decomposed
    over the ranks vs the whole domain evaluated on rank 0 alone, matched by
    gid (the reference's own recipe: parallel/tests/example_test_case.py:143-166).
    Returns max|a-b| / max|b| over the output fields on every rank."""
    import copy
    import torch
    from pysph_amd import device as dev
    a2 = copy.copy(args)
    one_problem = args.workload in ('dam_break', 'elastic', 'elastic_block')   # cut into slabs (strong scaling)
    if args.workload != 'dam_break':
        a2.n1 = 24
    a2.dx = max(args.dx, 0.05)
    a2.rings_dx = max(args.rings_dx, 2e-3)
    a2.no_reorder = True
    ctx = dev.HipContext(local_rank, tstream.cuda_stream)
    apply_options(a2, ctx)
    w = build_workload(a2, rank, world)
    nnps, a_eval, halo, domain, step, _ = setup(a2, w, rank, world, dist, ctx)
    for _ in range(3):      # the third evaluation runs the steady-state exchange (overlapped where the workload allows)
        step()
    fields = list(w.fields)
    gathered = []
    for pa in w.arrays:
        pa.gpu.sync_host()
        n = pa.get_number_of_particles(True)
        cols = [np.asarray(pa.gid[:n], dtype=np.float64)] + \
               [np.asarray(pa.get(f))[:n] if f in pa.properties else np.zeros(n) for f in fields]
        r = np.stack(cols, 1) if n else np.zeros((0, 1 + len(fields)))
        cnt = torch.tensor([r.shape[0]], dtype=torch.int64, device='cuda')
        cnts = [torch.zeros_like(cnt) for _ in range(world)]
        dist.all_gather(cnts, cnt)
        cap = max(int(c.item()) for c in cnts)
        buf = torch.zeros((max(cap, 1), 1 + len(fields)), dtype=torch.float64, device='cuda')
        if r.shape[0]:
            buf[:r.shape[0]] = torch.from_numpy(r).cuda()
        bufs = [torch.zeros_like(buf) for _ in range(world)]
        dist.all_gather(bufs, buf)
        gathered.append(np.concatenate(
            [b.cpu().numpy()[:int(c.item())] for b, c in zip(bufs, cnts)], 0))
    del nnps, a_eval, step, halo, domain
    ctx.close()
    worst = torch.zeros(1, dtype=torch.float64, device='cuda')
    if rank == 0:
        # the whole domain on this GPU alone
        ctx1 = dev.HipContext(local_rank, tstream.cuda_stream)
        apply_options(a2, ctx1)
        if one_problem:
            w1 = build_workload(a2, 0, 1)
        else:
            parts = [build_workload(a2, r, world) for r in range(world)]
            w1 = parts[0]
            for ai in range(len(w1.arrays)):
                for p in parts[1:]:
                    w1.arrays[ai].append_parray(p.arrays[ai])
                w1.arrays[ai].set_num_real_particles(w1.arrays[ai].get_number_of_particles())
        nn1, ev1, _, dom1, step1, _ = setup(a2, w1, 0, 1, None, ctx1)
        step1()
        for ai, pa in enumerate(w1.arrays):
            pa.gpu.sync_host()
            n = pa.get_number_of_particles(True)
            g = gathered[ai]
            if g.shape[0] != n:
                raise SystemExit('1-vs-N parity: %d particles on %d ranks, %d on one' % (
                    g.shape[0], world, n))
            if n == 0:
                continue
            o1 = np.argsort(np.asarray(pa.gid[:n]))
            oN = np.argsort(g[:, 0])
            for k, f in enumerate(fields):
                if f not in pa.properties:
                    continue
                b = np.asarray(pa.get(f))[:n][o1]
                a = g[oN, 1 + k]
                scale = np.max(np.abs(b))
                err = float(np.max(np.abs(a - b)) / scale) if scale > 0 else float(np.max(np.abs(a)))
                if not np.isfinite(err):
                    err = 1e300         # a NaN never compares greater
                worst[0] = max(float(worst[0]), err)
        del nn1, ev1, step1, dom1
        ctx1.close()
    dist.all_reduce(worst, op=dist.ReduceOp.MAX)
    val = float(worst.item())
    if val > PARITY_TOL:
        raise SystemExit('1-vs-%d-rank parity check failed: max relative difference %g' % (world, val))
    return val


if __name__ == '__main__':
    main()
