"""Made-up equations that exercise the translator subset of
pysph_amd/codegen.py (they are no physics; they touch every construct the
reference's equations use).  Run as Python by oracle/py_eval.py and as
generated HIP by the product; both must agree."""
from math import fabs, pow, sqrt  # noqa: F401  (used when the bodies run as Python)

from oracle.py_eval import declare  # noqa: F401
from pysph_amd.equations import Equation


class PowerLawState(Equation):
    """no sources; initialize + loop + post_loop; pow; array constant."""

    def __init__(self, dest, sources, k=1.5, n=1.4):
        self.k = k
        self.n = n
        super(PowerLawState, self).__init__(dest, sources)

    def initialize(self, d_idx, d_e):
        d_e[d_idx] = 0.0

    def loop(self, d_idx, d_p, d_rho, d_e, d_coef):
        d_p[d_idx] = self.k * pow(d_rho[d_idx], self.n) * d_coef[0]
        d_e[d_idx] = d_p[d_idx] / ((self.n - 1.0) * d_rho[d_idx]) + d_coef[1]

    def post_loop(self, d_idx, d_e, t, dt):
        d_e[d_idx] += 2 * t - dt


class KitchenSink(Equation):
    def __init__(self, dest, sources, a=0.3, b=2.0, flag=True):
        self.a = a
        self.b = b
        self.flag = flag
        super(KitchenSink, self).__init__(dest, sources)

    def initialize(self, d_idx, d_q, d_gx, d_gy, d_gz):
        d_q[d_idx] = 0.0
        d_gx[d_idx] = 0.0
        d_gy[d_idx] = 0.0
        d_gz[d_idx] = 0.0

    def loop(self, d_idx, s_idx, d_q, d_gx, d_gy, d_gz, s_m, s_rho, d_h, s_h,
             WI, WJ, DWI, DWJ, DWIJ, VIJ, XIJ, RIJ, R2IJ, HIJ, RHOIJ1, EPS, WIJ):
        tmp = declare('matrix(3)')
        if RIJ < 1e-12:
            return
        wbar = 0.5 * (WI + WJ)
        i = declare('int')
        for i in range(3):
            tmp[i] = 0.5 * (DWI[i] + DWJ[i]) * s_m[s_idx] * RHOIJ1
        vdotx = VIJ[0] * XIJ[0] + VIJ[1] * XIJ[1] + VIJ[2] * XIJ[2]
        if vdotx < 0 and self.flag:
            fac = self.a * vdotx / (R2IJ + EPS)
        elif vdotx > 0.01:
            fac = min(vdotx, self.b, 3.0)
        else:
            fac = -sqrt(fabs(vdotx)) * pow(HIJ, 0.5)
        d_q[d_idx] += s_m[s_idx] / s_rho[s_idx] * wbar * \
            (1.0 if d_h[d_idx] > s_h[s_idx] else 0.5) + WIJ * 1e-3
        d_gx[d_idx] += fac * tmp[0] - DWIJ[0] * 1e-3
        d_gy[d_idx] += fac * tmp[1] - DWIJ[1] * 1e-3
        d_gz[d_idx] -= -fac * tmp[2]

    def post_loop(self, d_idx, d_q, d_gx, d_x, t, dt):
        d_q[d_idx] = max(d_q[d_idx], 0.0) + t * dt
        if not (d_x[d_idx] > 0.5):
            d_gx[d_idx] *= 2.0


class WallPush(Equation):
    """second equation on the same destination, different source list."""

    def __init__(self, dest, sources, c=0.7):
        self.c = c
        super(WallPush, self).__init__(dest, sources)

    def loop(self, d_idx, s_idx, d_gy, s_p, d_rho, WIJ, RIJ, HIJ):
        d_gy[d_idx] += self.c * s_p[s_idx] / d_rho[d_idx] * WIJ * (RIJ / HIJ) ** 2
