"""TEST FIXTURE (not product code): the solid-wall equations of the
transport-velocity scheme (Adami et al. 2012) as *Python bodies*, for the
generated-family path (``pysph_amd.codegen``) on a box without the reference.

The product consumes the reference's own classes
(``pysph.sph.wc.transport_velocity``) when ``pysph`` is importable
(``pysph_amd.scheme.TVFScheme``); the GPU box has no reference, so the tests
hand this restatement to ``TVFScheme(..., wall_equations=<this module>)``.
tests/test_codegen.py checks (where the reference is present) that both
translate to the same generated code.

Same class names, constructor arguments and array names as
pysph/sph/wc/transport_velocity.py (``SetWallVelocity`` :84-134,
``SolidWallNoSlipBC`` :548-638, ``SolidWallPressureBC`` :641-735,
``ContinuitySolid`` :157-173, ``VolumeSummation`` :61-75), so
``TVFScheme(fluids, solids)`` builds the reference's group list.  The bodies
follow the reference's semantics statement by statement (the order of the
floating-point operations matters for parity) but are written for the
translator's subset: plain float arithmetic on ``d_*[d_idx]`` / ``s_*[s_idx]``
and the precomputed pair symbols.
"""
from pysph_amd.equations import Equation


class SetWallVelocity(Equation):
    """Shepard-filtered fluid velocity on the wall particles, Eq. (22), and the
    dummy velocity u_g = 2 u_wall - u_filtered of Eq. (23).  Destination
    arrays need uf, vf, wf, wij, ug, vg, wg."""

    def initialize(self, d_idx, d_uf, d_vf, d_wf, d_wij):
        d_uf[d_idx] = 0.0
        d_vf[d_idx] = 0.0
        d_wf[d_idx] = 0.0
        d_wij[d_idx] = 0.0

    def loop(self, d_idx, s_idx, d_uf, d_vf, d_wf, s_u, s_v, s_w, d_wij, WIJ):
        d_wij[d_idx] += WIJ
        d_uf[d_idx] += s_u[s_idx] * WIJ
        d_vf[d_idx] += s_v[s_idx] * WIJ
        d_wf[d_idx] += s_w[s_idx] * WIJ

    def post_loop(self, d_uf, d_vf, d_wf, d_wij, d_idx, d_ug, d_vg, d_wg,
                  d_u, d_v, d_w):
        # wall particles far from the fluid keep wij = 0 (and uf = 0)
        if d_wij[d_idx] > 1e-12:
            d_uf[d_idx] /= d_wij[d_idx]
            d_vf[d_idx] /= d_wij[d_idx]
            d_wf[d_idx] /= d_wij[d_idx]
        d_ug[d_idx] = 2 * d_u[d_idx] - d_uf[d_idx]
        d_vg[d_idx] = 2 * d_v[d_idx] - d_vf[d_idx]
        d_wg[d_idx] = 2 * d_w[d_idx] - d_wf[d_idx]


class SolidWallPressureBC(Equation):
    """Wall pressure extrapolated from the fluid with the hydrostatic
    correction, Eq. (27), then the wall density from the linear state
    equation, Eq. (28).  ``d_au..d_aw`` are the prescribed wall accelerations."""

    def __init__(self, dest, sources, rho0, p0, b=1.0, gx=0.0, gy=0.0, gz=0.0):
        self.rho0 = rho0
        self.p0 = p0
        self.b = b
        self.gx = gx
        self.gy = gy
        self.gz = gz
        super(SolidWallPressureBC, self).__init__(dest, sources)

    def initialize(self, d_idx, d_p, d_wij):
        d_p[d_idx] = 0.0
        d_wij[d_idx] = 0.0

    def loop(self, d_idx, s_idx, d_p, s_p, d_wij, s_rho, d_au, d_av, d_aw,
             WIJ, XIJ):
        gdotxij = (self.gx - d_au[d_idx]) * XIJ[0] + \
            (self.gy - d_av[d_idx]) * XIJ[1] + \
            (self.gz - d_aw[d_idx]) * XIJ[2]
        d_p[d_idx] += s_p[s_idx] * WIJ + s_rho[s_idx] * gdotxij * WIJ
        d_wij[d_idx] += WIJ

    def post_loop(self, d_idx, d_wij, d_p, d_rho):
        if d_wij[d_idx] > 1e-14:
            d_p[d_idx] /= d_wij[d_idx]
        d_rho[d_idx] = self.rho0 * (d_p[d_idx] / self.p0 + self.b)


class SolidWallNoSlipBC(Equation):
    """Viscous force of the wall on the fluid with the dummy wall velocity
    (ug, vg, wg) of ``SetWallVelocity``: third term of Eq. (8) in Adami 2013.
    Destination = fluid, sources = walls."""

    def __init__(self, dest, sources, nu):
        self.nu = nu
        super(SolidWallNoSlipBC, self).__init__(dest, sources)

    def initialize(self, d_idx, d_au, d_av, d_aw):
        d_au[d_idx] = 0.0
        d_av[d_idx] = 0.0
        d_aw[d_idx] = 0.0

    def loop(self, d_idx, s_idx, d_m, d_rho, s_rho, d_V, s_V, d_u, d_v, d_w,
             d_au, d_av, d_aw, s_ug, s_vg, s_wg, DWIJ, R2IJ, EPS, XIJ):
        etai = self.nu * d_rho[d_idx]
        etaj = self.nu * s_rho[s_idx]
        etaij = 2 * (etai * etaj) / (etai + etaj)
        Vi = 1. / d_V[d_idx]
        Vj = 1. / s_V[s_idx]
        Vi2 = Vi * Vi
        Vj2 = Vj * Vj
        Fij = XIJ[0] * DWIJ[0] + XIJ[1] * DWIJ[1] + XIJ[2] * DWIJ[2]
        tmp = 1. / d_m[d_idx] * (Vi2 + Vj2) * (etaij * Fij / (R2IJ + EPS))
        d_au[d_idx] += tmp * (d_u[d_idx] - s_ug[s_idx])
        d_av[d_idx] += tmp * (d_v[d_idx] - s_vg[s_idx])
        d_aw[d_idx] += tmp * (d_w[d_idx] - s_wg[s_idx])


class ContinuitySolid(Equation):
    """Continuity contribution of wall particles using their dummy velocity."""

    def loop(self, d_idx, s_idx, d_rho, d_u, d_v, d_w, d_arho, s_m, s_rho,
             s_ug, s_vg, s_wg, DWIJ):
        Vj = s_m[s_idx] / s_rho[s_idx]
        rhoi = d_rho[d_idx]
        uij = d_u[d_idx] - s_ug[s_idx]
        vij = d_v[d_idx] - s_vg[s_idx]
        wij = d_w[d_idx] - s_wg[s_idx]
        vij_dot_dwij = uij * DWIJ[0] + vij * DWIJ[1] + wij * DWIJ[2]
        d_arho[d_idx] += rhoi * Vj * vij_dot_dwij


class VolumeSummation(Equation):
    """Number density V = sum_b W_ab (inverse particle volume)."""

    def initialize(self, d_idx, d_V):
        d_V[d_idx] = 0.0

    def loop(self, d_idx, d_V, WIJ):
        d_V[d_idx] += WIJ


class VolumeFromMassDensity(Equation):
    """V = rho / m, per particle (no sources)."""

    def loop(self, d_idx, d_V, d_rho, d_m):
        d_V[d_idx] = d_rho[d_idx] / d_m[d_idx]
