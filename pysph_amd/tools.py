"""``SPHEvaluator``: particle arrays + equations (+ domain, kernel) -> one SPH
evaluation, the reference's convenience entry for post-processing
(pysph/tools/sph_evaluator.py:16-86) on the HIP backend.

Same constructor arguments and methods (``evaluate``, ``update``,
``update_particle_arrays``); the default kernel is the Gaussian like the
reference's, the default neighbour search is ``HipNNPS``.  Host arrays stay
authoritative (``sync='auto'``): inputs are pushed before and results pulled
after every ``evaluate``.
"""
from .acceleration_eval import AccelerationEval, SPHCompiler
from .kernels import Gaussian
from .nnps import HipNNPS


class SPHEvaluator(object):
    def __init__(self, arrays, equations, dim, kernel=None, domain_manager=None,
                 backend='hip', nnps_factory=HipNNPS, ctx=None):
        if backend not in ('hip', '', None):
            raise ValueError("pysph_amd.tools.SPHEvaluator: backend must be 'hip'")
        self.arrays = arrays
        self.equations = equations
        self.domain_manager = domain_manager
        self.dim = dim
        self.kernel = Gaussian(dim=dim) if kernel is None else kernel
        self.nnps_factory = nnps_factory
        self.backend = backend
        # an evaluator owns its device state: a context of its own unless one is given
        from .device import HipContext
        self.ctx = ctx if ctx is not None else HipContext(0)
        ctx = self.ctx
        self.func_eval = AccelerationEval(arrays, equations, self.kernel)
        SPHCompiler(self.func_eval, None, ctx=ctx).compile()
        self._create_nnps(arrays)

    def evaluate(self, t=0.0, dt=0.1):
        """Evaluate the equations (dummy t, dt may be passed)."""
        self.func_eval.compute(t, dt)

    def update(self, update_domain=True):
        """Particles moved (same arrays): rebuild the neighbour structure."""
        if update_domain:
            self.nnps.update_domain()
        self.nnps.update()

    def update_particle_arrays(self, arrays):
        """A new set of particle arrays with the same properties."""
        self.arrays = arrays
        self.func_eval.update_particle_arrays(arrays)
        self._create_nnps(arrays)

    def _create_nnps(self, arrays):
        kw = dict(dim=self.kernel.dim, particles=arrays,
                  radius_scale=self.kernel.radius_scale, domain=self.domain_manager,
                  cache=True)
        kw['ctx'] = self.ctx
        self.nnps = self.nnps_factory(**kw)
        self.func_eval.set_nnps(self.nnps)
