"""Shared builders for the parity tests (equation sets named like the
reference's schemes)."""
import numpy as np

from pysph_amd import kernels as K
from pysph_amd.equations import Group, SummationDensity
from pysph_amd.scheme import WCSPHScheme, TVFScheme
from pysph_amd.examples import dam_break_3d as db

WC_OUT = ['rho', 'p', 'cs', 'arho', 'au', 'av', 'aw', 'ax', 'ay', 'az',
          'dt_cfl', 'dt_force']
EL_OUT = ['p', 'v00', 'v01', 'v02', 'v10', 'v11', 'v12', 'v20', 'v21', 'v22',
          'r00', 'r01', 'r02', 'r11', 'r12', 'r22', 'arho', 'au', 'av', 'aw',
          'ax', 'ay', 'az', 'as00', 'as01', 'as02', 'as11', 'as12', 'as22']
TVF_OUT = ['rho', 'V', 'p', 'au', 'av', 'aw', 'auhat', 'avhat', 'awhat']


def golden_case(name, g, wall_equations=None):
    """(equations, kernel, dim, output props) matching make_golden.py."""
    if name in ('wcsph_dam_dx0.1', 'wcsph_dam_varh'):
        dx = float(g['meta/dx'])
        s = db.create_scheme(dx)
        return s.get_equations(), K.WendlandQuintic(dim=3), 3, WC_OUT
    if name == 'wcsph_cube_varh':
        dx = float(g['meta/dx'])
        s = WCSPHScheme(['fluid'], [], dim=3, rho0=1000.0, c0=10.0,
                        h0=1.2 * dx, hdx=1.2, gx=0.5, gy=-0.25, gz=-9.81,
                        alpha=1.0, beta=1.0, gamma=7.0,
                        tensile_correction=True, summation_density=True)
        return s.get_equations(), K.CubicSpline(dim=3), 3, WC_OUT
    if name == 'sd_1d_line':
        eqs = [Group(equations=[SummationDensity(dest='fluid',
                                                 sources=['fluid'])])]
        return eqs, K.CubicSpline(dim=1), 1, ['rho']
    if name == 'tvf_cube':
        dx = float(g['meta/dx'])
        s = TVFScheme(['fluid'], [], dim=3, rho0=1.0, c0=10.0, nu=0.01,
                      p0=100.0, pb=100.0, h0=dx, gx=0.1, alpha=0.2)
        return s.get_equations(), K.QuinticSpline(dim=3), 3, TVF_OUT
    if name == 'tvf_wall':
        dx = float(g['meta/dx'])
        # wall equations: the product's own (pysph_amd/wall_bc.py), or -- for the
        # cross-check against the statement-by-statement restatement of the
        # reference's bodies -- tests/wall_equations_fixture.py
        s = TVFScheme(['fluid'], ['wall'], dim=3, rho0=1.0, c0=10.0, nu=0.01,
                      p0=100.0, pb=100.0, h0=dx, gy=-0.5, alpha=0.2,
                      wall_equations=wall_equations)
        return s.get_equations(), K.QuinticSpline(dim=3), 3, TVF_OUT + [
            'wij', 'uf', 'vf', 'wf', 'ug', 'vg', 'wg']
    if name in ('elastic_2d', 'elastic_3d'):
        from pysph_amd.solid_mech import ElasticSolidsScheme
        dim = int(g['meta/dim'])
        s = ElasticSolidsScheme(['solid'], [], dim=dim)
        return s.get_equations(), K.CubicSpline(dim=dim), dim, EL_OUT
    raise KeyError(name)


def rel_err(a, b, scale=None):
    """max |a-b| / scale, scale = max|b| of the field unless given (mixed
    abs/rel measure: accelerations near cancellation are judged against the
    field's magnitude, SURVEY.md section 7 'Hard parts')."""
    a = np.asarray(a, dtype=float)
    b = np.asarray(b, dtype=float)
    if scale is None:
        scale = max(np.max(np.abs(b)), 1e-300) if b.size else 1.0
    return float(np.max(np.abs(a - b)) / scale) if b.size else 0.0


class ThreadDist(object):
    """In-process stand-in for ``torch.distributed`` used by the -m gpu tests:
    `world` threads play the ranks of one node on ONE GPU (RCCL cannot put two
    ranks on one device).  Tensors are handed over by reference and copied on
    the receiver's stream after a device synchronize -- the ordering RCCL's
    stream semantics give.  ``view(rank)`` returns the per-rank object to pass
    as ``dist=``."""

    def __init__(self, world):
        import queue
        import threading
        try:                      # initialise torch's HIP state once, in the
            import torch          # creating thread (lazy init from several
            if torch.cuda.is_available():      # threads at once races)
                torch.cuda.init()
                torch.zeros(1, device='cuda')      # forces context creation
                torch.cuda.synchronize()
        except ImportError:
            pass
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world
        self.q = {(a, b): queue.Queue() for a in range(world) for b in range(world)}

    def view(self, rank):
        return _ThreadDistRank(self, rank)


class _Done(object):
    def wait(self):
        return True


class _ThreadDistRank(object):
    class ReduceOp(object):
        MIN, MAX, SUM = 'min', 'max', 'sum'

    class P2POp(object):
        def __init__(self, op, tensor, peer):
            self.op, self.tensor, self.peer = op, tensor, peer

    isend, irecv = 'isend', 'irecv'

    def __init__(self, hub, rank):
        self.hub, self.rank = hub, rank

    def _sync(self):
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    def all_gather_into_tensor(self, out, inp):
        import torch
        hub = self.hub
        self._sync()
        hub.slots[self.rank] = inp.clone()
        self._sync()
        hub.barrier.wait()
        out.copy_(torch.cat([s.reshape(-1) for s in hub.slots]))
        self._sync()
        hub.barrier.wait()

    def all_reduce(self, t, op=None):
        import torch
        hub = self.hub
        self._sync()
        hub.slots[self.rank] = t.clone()
        self._sync()
        hub.barrier.wait()
        st = torch.stack(hub.slots)
        r = {'min': st.min(0).values, 'max': st.max(0).values, 'sum': st.sum(0)}[op]
        self._sync()
        hub.barrier.wait()
        t.copy_(r)
        self._sync()
        hub.barrier.wait()

    def batch_isend_irecv(self, reqs):
        hub = self.hub
        self._sync()
        for r in reqs:
            if r.op == 'isend':
                hub.q[(self.rank, r.peer)].put(r.tensor)
        for r in reqs:
            if r.op == 'irecv':
                src = hub.q[(r.peer, self.rank)].get(timeout=120)
                r.tensor.copy_(src[:r.tensor.numel()])
        self._sync()
        return [_Done()]

    def barrier(self):
        self.hub.barrier.wait()


def live_rows(pa, prop='x'):
    """property `prop` of every row the DEVICE array holds that is a particle: the padding rows of the round-trip-free
    ghost protocols (sph_halo_append_padded, sph_domain_images_padded) are parked at 1e18 and skipped"""
    import numpy as np
    from pysph_amd import device as dev
    g = pa.gpu
    n = g.get_number_of_particles()
    out = {}
    for q in ('x', prop):
        b = np.empty(n)
        dev._check(g.lib.sph_array_pull(g.ctx._h, g.array_id, dev.prop_id(q), b.ctypes.data_as(dev._PD), 0, n))
        out[q] = b
    return out[prop][np.abs(out['x']) < 1e17]


class _DeviceView(object):
    """n doubles at a raw device address, for torch.as_tensor (__cuda_array_interface__)"""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {'shape': (int(n),), 'typestr': '<f8', 'data': (int(ptr), False), 'version': 2}


def device_add(pa, prop, delta, n=None):
    """property `prop` of the DEVICE copy of `pa` += delta (host array, first n rows), written in place on the device
    through the property's raw pointer -- what a device-resident mover (a stage kernel) does; a host push of positions
    would make the next neighbour update look at the particles first (positions from the host may lie anywhere).  The
    host copy is updated alike."""
    import torch
    g = pa.gpu
    n = len(delta) if n is None else n
    g.ctx.synchronize()
    t = torch.as_tensor(_DeviceView(g.device_ptr(prop), n), device=torch.device('cuda', g.ctx.device))
    t += torch.from_numpy(np.ascontiguousarray(delta[:n], dtype=np.float64)).to(t.device)
    torch.cuda.synchronize()
    pa.properties[prop][:n] += delta[:n]
