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


# ---------------------------------------------------------------------------
# The WCSPH rate equations as Python bodies (semantics of
# pysph/sph/basic_equations.py:177-192, :285-300 and pysph/sph/wc/basic.py:198-271):
# the SAME physics the hand-written FamWCSPH kernel implements, here pushed
# through the translator -- generated vs hand-written on identical input.
# ---------------------------------------------------------------------------
class PyContinuity(Equation):
    def initialize(self, d_idx, d_arho):
        d_arho[d_idx] = 0.0

    def loop(self, d_idx, d_arho, s_idx, s_m, DWIJ, VIJ):
        vijdotdwij = DWIJ[0] * VIJ[0] + DWIJ[1] * VIJ[1] + DWIJ[2] * VIJ[2]
        d_arho[d_idx] += s_m[s_idx] * vijdotdwij


class PyMomentum(Equation):
    def __init__(self, dest, sources, c0, alpha=1.0, beta=1.0, gx=0.0, gy=0.0,
                 gz=0.0, tensile_correction=False):
        self.c0, self.alpha, self.beta = c0, alpha, beta
        self.gx, self.gy, self.gz = gx, gy, gz
        self.tensile_correction = tensile_correction
        super(PyMomentum, self).__init__(dest, sources)

    def initialize(self, d_idx, d_au, d_av, d_aw, d_dt_cfl):
        d_au[d_idx] = 0.0
        d_av[d_idx] = 0.0
        d_aw[d_idx] = 0.0
        d_dt_cfl[d_idx] = 0.0

    def loop(self, d_idx, s_idx, d_rho, d_cs, d_p, d_au, d_av, d_aw, s_m, s_rho,
             s_cs, s_p, VIJ, XIJ, HIJ, R2IJ, RHOIJ1, EPS, DWIJ, WIJ, WDP, d_dt_cfl):
        rhoi21 = 1.0 / (d_rho[d_idx] * d_rho[d_idx])
        rhoj21 = 1.0 / (s_rho[s_idx] * s_rho[s_idx])
        vijdotxij = VIJ[0] * XIJ[0] + VIJ[1] * XIJ[1] + VIJ[2] * XIJ[2]
        piij = 0.0
        if vijdotxij < 0:
            cij = 0.5 * (d_cs[d_idx] + s_cs[s_idx])
            muij = (HIJ * vijdotxij) / (R2IJ + EPS)
            piij = -self.alpha * cij * muij + self.beta * muij * muij
            piij = piij * RHOIJ1
        if R2IJ > 1e-12:
            _dt_cfl = abs(HIJ * vijdotxij / R2IJ) + self.c0
            d_dt_cfl[d_idx] = max(_dt_cfl, d_dt_cfl[d_idx])
        tmpi = d_p[d_idx] * rhoi21
        tmpj = s_p[s_idx] * rhoj21
        fij = WIJ / WDP
        Ri = 0.0
        Rj = 0.0
        if self.tensile_correction:
            fij = fij * fij
            fij = fij * fij
            if d_p[d_idx] > 0:
                Ri = 0.01 * tmpi
            else:
                Ri = 0.2 * abs(tmpi)
            if s_p[s_idx] > 0:
                Rj = 0.01 * tmpj
            else:
                Rj = 0.2 * abs(tmpj)
        tmp = (tmpi + tmpj) + (Ri + Rj) * fij
        d_au[d_idx] += -s_m[s_idx] * (tmp + piij) * DWIJ[0]
        d_av[d_idx] += -s_m[s_idx] * (tmp + piij) * DWIJ[1]
        d_aw[d_idx] += -s_m[s_idx] * (tmp + piij) * DWIJ[2]

    def post_loop(self, d_idx, d_au, d_av, d_aw, d_dt_force):
        d_au[d_idx] += self.gx
        d_av[d_idx] += self.gy
        d_aw[d_idx] += self.gz
        d_dt_force[d_idx] = d_au[d_idx] * d_au[d_idx] + d_av[d_idx] * d_av[d_idx] + \
            d_aw[d_idx] * d_aw[d_idx]


class PyXSPH(Equation):
    def __init__(self, dest, sources, eps=0.5):
        self.eps = eps
        super(PyXSPH, self).__init__(dest, sources)

    def initialize(self, d_idx, d_ax, d_ay, d_az):
        d_ax[d_idx] = 0.0
        d_ay[d_idx] = 0.0
        d_az[d_idx] = 0.0

    def loop(self, s_idx, d_idx, s_m, d_ax, d_ay, d_az, WIJ, RHOIJ1, VIJ):
        tmp = -self.eps * s_m[s_idx] * WIJ * RHOIJ1
        d_ax[d_idx] += tmp * VIJ[0]
        d_ay[d_idx] += tmp * VIJ[1]
        d_az[d_idx] += tmp * VIJ[2]

    def post_loop(self, d_idx, d_ax, d_ay, d_az, d_u, d_v, d_w):
        d_ax[d_idx] += d_u[d_idx]
        d_ay[d_idx] += d_v[d_idx]
        d_az[d_idx] += d_w[d_idx]


# ---------------------------------------------------------------------------
# multi-component (strided) properties: d_g3[d_idx*3 + k], s_g3[s_idx*3 + k]
# (the access pattern of the reference's delta-SPH equations,
# pysph/sph/wc/basic.py:355-414)
# ---------------------------------------------------------------------------
class StridedGradient(Equation):
    def initialize(self, d_idx, d_g3):
        d_g3[d_idx * 3 + 0] = 0.0
        d_g3[d_idx * 3 + 1] = 0.0
        d_g3[3 * d_idx + 2] = 0.0

    def loop(self, d_idx, s_idx, d_g3, d_rho, s_rho, s_m, DWIJ):
        Vj = s_m[s_idx] / s_rho[s_idx]
        d_g3[d_idx * 3 + 0] += (s_rho[s_idx] - d_rho[d_idx]) * DWIJ[0] * Vj
        d_g3[d_idx * 3 + 1] += (s_rho[s_idx] - d_rho[d_idx]) * DWIJ[1] * Vj
        d_g3[d_idx * 3 + 2] += (s_rho[s_idx] - d_rho[d_idx]) * DWIJ[2] * Vj


class StridedDiffusion(Equation):
    def __init__(self, dest, sources, delta=0.1, c0=10.0):
        self.delta = delta
        self.c0 = c0
        super(StridedDiffusion, self).__init__(dest, sources)

    def initialize(self, d_idx, d_arho):
        d_arho[d_idx] = 0.0

    def loop(self, d_idx, d_arho, s_idx, s_m, d_rho, s_rho, DWIJ, XIJ, R2IJ, HIJ,
             EPS, d_g3, s_g3):
        Vj = s_m[s_idx] / s_rho[s_idx]
        fac = -2.0 * (s_rho[s_idx] - d_rho[d_idx]) / (R2IJ + EPS)
        psix = fac * XIJ[0] - d_g3[d_idx * 3 + 0] - s_g3[s_idx * 3 + 0]
        psiy = fac * XIJ[1] - d_g3[d_idx * 3 + 1] - s_g3[s_idx * 3 + 1]
        psiz = fac * XIJ[2] - d_g3[d_idx * 3 + 2] - s_g3[s_idx * 3 + 2]
        d_arho[d_idx] += self.delta * HIJ * self.c0 * \
            (psix * DWIJ[0] + psiy * DWIJ[1] + psiz * DWIJ[2]) * Vj


# ---------------------------------------------------------------------------
# loop_all: the equation walks the neighbour list itself (NBRS, N_NBRS) and calls
# the kernel object -- semantics of ShepardFilter,
# pysph/sph/wc/density_correction.py:24-46 (zeroth-order density
# re-initialisation).  initialize() copies rho -> rhotmp for ALL particles
# before any neighbour sum reads s_rhotmp.
# ---------------------------------------------------------------------------
class ShepardFilter(Equation):
    def initialize(self, d_idx, d_rho, d_rhotmp):
        d_rhotmp[d_idx] = d_rho[d_idx]

    def loop_all(self, d_idx, d_rho, d_x, d_y, d_z, s_m, s_rhotmp, s_x, s_y,
                 s_z, d_h, s_h, SPH_KERNEL, NBRS, N_NBRS):
        i, s_idx = declare('int', 2)
        xij = declare('matrix(3)')
        tmp_w = 0.0
        x = d_x[d_idx]
        y = d_y[d_idx]
        z = d_z[d_idx]
        d_rho[d_idx] = 0.0
        for i in range(N_NBRS):
            s_idx = NBRS[i]
            xij[0] = x - s_x[s_idx]
            xij[1] = y - s_y[s_idx]
            xij[2] = z - s_z[s_idx]
            rij = sqrt(xij[0] * xij[0] + xij[1] * xij[1] + xij[2] * xij[2])
            hij = (d_h[d_idx] + s_h[s_idx]) * 0.5
            wij = SPH_KERNEL.kernel(xij, rij, hij)
            tmp_w += wij * s_m[s_idx] / s_rhotmp[s_idx]
            d_rho[d_idx] += wij * s_m[s_idx]
        d_rho[d_idx] /= tmp_w


class GradientAllNbrs(Equation):
    """made up: SPH_KERNEL.gradient + dwdq inside loop_all, two sources"""

    def __init__(self, dest, sources, scale=0.5):
        self.scale = scale
        super(GradientAllNbrs, self).__init__(dest, sources)

    def initialize(self, d_idx, d_gx, d_gy, d_gz):
        d_gx[d_idx] = 0.0
        d_gy[d_idx] = 0.0
        d_gz[d_idx] = 0.0

    def loop_all(self, d_idx, d_gx, d_gy, d_gz, d_x, d_y, d_z, d_h, s_x, s_y, s_z, s_m,
                 s_rho, SPH_KERNEL, NBRS, N_NBRS):
        k, j = declare('int', 2)
        xij = declare('matrix(3)')
        grad = declare('matrix(3)')
        for k in range(N_NBRS):
            j = NBRS[k]
            xij[0] = d_x[d_idx] - s_x[j]
            xij[1] = d_y[d_idx] - s_y[j]
            xij[2] = d_z[d_idx] - s_z[j]
            rij = sqrt(xij[0] * xij[0] + xij[1] * xij[1] + xij[2] * xij[2])
            SPH_KERNEL.gradient(xij, rij, d_h[d_idx], grad)
            vol = self.scale * s_m[j] / s_rho[j]
            d_gx[d_idx] += vol * grad[0]
            d_gy[d_idx] += vol * grad[1]
            d_gz[d_idx] += vol * (grad[2] + 1e-3 * SPH_KERNEL.dwdq(rij, d_h[d_idx]))


class SmoothCopy(Equation):
    """initialize() saves q in qtmp and clears q; loop() reads the SAVED value of
    the neighbours (s_qtmp): needs initialize finished for all particles first."""

    def initialize(self, d_idx, d_q, d_qtmp):
        d_qtmp[d_idx] = d_q[d_idx]
        d_q[d_idx] = 0.0

    def loop(self, d_idx, s_idx, d_q, s_qtmp, s_m, s_rho, WIJ):
        d_q[d_idx] += s_qtmp[s_idx] * s_m[s_idx] / s_rho[s_idx] * WIJ


# ---------------------------------------------------------------------------
# helper functions with array / int arguments (types follow the defaults, as in
# the reference's translator: list -> double*, int -> int, float -> double)
# ---------------------------------------------------------------------------
def stack_columns(A=[0.0, 0.0], b=[0.0, 0.0], n=3, nb=1, lda=3, out=[0.0, 0.0]):
    """[A | b] of the leading n x n block of a row-major lda x lda matrix and
    nb right-hand sides, row-major n x (n + nb) in `out`."""
    i, j, w = declare('int', 3)
    w = n + nb
    for i in range(n):
        for j in range(n):
            out[w * i + j] = A[lda * i + j]
        for j in range(nb):
            out[w * i + n + j] = b[nb * i + j]


def eliminate(m=[1.0, 0.0], n=3, nb=1, x=[0.0, 0.0]):
    """Gauss-Jordan with row pivoting on the augmented n x (n + nb) matrix m;
    the solution goes to x (n x nb).  Returns 1.0 for a singular matrix."""
    i, j, k, w, piv = declare('int', 5)
    w = n + nb
    for k in range(n):
        piv = k
        for i in range(k + 1, n):
            if fabs(m[w * i + k]) > fabs(m[w * piv + k]):
                piv = i
        if fabs(m[w * piv + k]) < 1e-14:
            return 1.0
        if piv != k:
            for j in range(w):
                tmp = m[w * k + j]
                m[w * k + j] = m[w * piv + j]
                m[w * piv + j] = tmp
        d = 1.0 / m[w * k + k]
        for j in range(w):
            m[w * k + j] = m[w * k + j] * d
        for i in range(n):
            if i != k:
                f = m[w * i + k]
                for j in range(w):
                    m[w * i + j] -= f * m[w * k + j]
    for i in range(n):
        for j in range(nb):
            x[nb * i + j] = m[w * i + n + j]
    return 0.0


class CorrectionMatrix(Equation):
    """Bonet-Lok gradient-correction matrix (the pattern of
    kernel_correction.py:40-78): a stride-9 property zeroed in a loop, filled
    from loop_all with SPH_KERNEL.gradient and a doubly nested component loop."""

    def __init__(self, dest, sources, dim=3):
        self.dim = dim
        super(CorrectionMatrix, self).__init__(dest, sources)

    def initialize(self, d_idx, d_lmat):
        i = declare('int')
        for i in range(9):
            d_lmat[9 * d_idx + i] = 0.0

    def loop_all(self, d_idx, d_lmat, d_x, d_y, d_z, d_h, s_x, s_y, s_z, s_h, s_m, s_rho,
                 SPH_KERNEL, NBRS, N_NBRS):
        i, j, k, s_idx, n = declare('int', 5)
        xij = declare('matrix(3)')
        dw = declare('matrix(3)')
        n = self.dim
        for k in range(N_NBRS):
            s_idx = NBRS[k]
            xij[0] = d_x[d_idx] - s_x[s_idx]
            xij[1] = d_y[d_idx] - s_y[s_idx]
            xij[2] = d_z[d_idx] - s_z[s_idx]
            r = sqrt(xij[0] * xij[0] + xij[1] * xij[1] + xij[2] * xij[2])
            SPH_KERNEL.gradient(xij, r, 0.5 * (d_h[d_idx] + s_h[s_idx]), dw)
            vol = s_m[s_idx] / s_rho[s_idx]
            if r > 1e-12:
                for i in range(n):
                    for j in range(n):
                        d_lmat[9 * d_idx + 3 * i + j] -= vol * dw[i] * xij[j]


class CorrectGradient(Equation):
    """kernel_correction.py:95-125 pattern: solve L x = DWIJ per pair with a
    helper taking local matrices and REPLACE DWIJ for the equations after this
    one in the group."""

    def __init__(self, dest, sources, dim=3, tol=0.5):
        self.dim = dim
        self.tol = tol
        super(CorrectGradient, self).__init__(dest, sources)

    def _get_helpers_(self):
        return [eliminate]

    def loop(self, d_idx, d_lmat, DWIJ, HIJ):
        i, j, n, w = declare('int', 4)
        n = self.dim
        w = n + 1
        aug = declare('matrix(12)')
        res = declare('matrix(3)')
        for i in range(n):
            for j in range(n):
                aug[w * i + j] = d_lmat[9 * d_idx + 3 * i + j]
            aug[w * i + n] = DWIJ[i]
        bad = eliminate(aug, n, 1, res)
        before = 0.0
        after = 0.0
        for i in range(n):
            before += fabs(DWIJ[i])
            after += fabs(res[i])
        if bad < 0.5 and fabs(after - before) < self.tol * (before + 1e-4 * HIJ):
            for i in range(n):
                DWIJ[i] = res[i]


class GradientOfLinearField(Equation):
    """consumer of the (corrected) DWIJ: gradient of f = 2x - 3y + z + 1, which
    the corrected kernel gradient reproduces to rounding"""

    def initialize(self, d_idx, d_gx, d_gy, d_gz):
        d_gx[d_idx] = 0.0
        d_gy[d_idx] = 0.0
        d_gz[d_idx] = 0.0

    def loop(self, d_idx, s_idx, d_gx, d_gy, d_gz, s_m, s_rho, XIJ, DWIJ):
        fji = -(2.0 * XIJ[0] - 3.0 * XIJ[1] + XIJ[2])
        vol = s_m[s_idx] / s_rho[s_idx]
        d_gx[d_idx] += vol * fji * DWIJ[0]
        d_gy[d_idx] += vol * fji * DWIJ[1]
        d_gz[d_idx] += vol * fji * DWIJ[2]


class MomentMatrix(Equation):
    """bc/interpolate.py:340-380 pattern: 4x4 moment matrix per destination in
    a stride-16 property, addressed through an int local holding 16*d_idx."""

    def initialize(self, d_idx, d_amat, d_bvec):
        i, j = declare('int', 2)
        for i in range(4):
            d_bvec[4 * d_idx + i] = 0.0
            for j in range(4):
                d_amat[16 * d_idx + j + 4 * i] = 0.0

    def loop(self, d_idx, s_idx, d_amat, d_bvec, s_m, s_rho, s_p, XIJ, WIJ):
        vol = s_m[s_idx] / s_rho[s_idx]
        i16, i4, i, j = declare('int', 4)
        i16 = 16 * d_idx
        i4 = 4 * d_idx
        basis = declare('matrix(4)')
        basis[0] = 1.0
        basis[1] = -XIJ[0]
        basis[2] = -XIJ[1]
        basis[3] = -XIJ[2]
        for i in range(4):
            d_bvec[i4 + i] += s_p[s_idx] * basis[i] * WIJ * vol
            for j in range(4):
                d_amat[i16 + 4 * i + j] += basis[i] * basis[j] * WIJ * vol


class SolveMoments(Equation):
    """bc/interpolate.py:292-313 pattern: post_loop gathers the strided
    properties into local matrices, calls two helpers, scatters the result."""

    def __init__(self, dest, sources, dim=3):
        self.dim = dim
        super(SolveMoments, self).__init__(dest, sources)

    def _get_helpers_(self):
        return [stack_columns, eliminate]

    def post_loop(self, d_idx, d_amat, d_bvec, d_pfit):
        a = declare('matrix(16)')
        aug = declare('matrix(20)')
        b = declare('matrix(4)')
        res = declare('matrix(4)')
        i, n, i16, i4 = declare('int', 4)
        i16 = 16 * d_idx
        i4 = 4 * d_idx
        for i in range(16):
            a[i] = d_amat[i16 + i]
        for i in range(4):
            b[i] = d_bvec[i4 + i]
            res[i] = 0.0
        n = self.dim + 1
        stack_columns(a, b, n, 1, 4, aug)
        eliminate(aug, n, 1, res)
        for i in range(4):
            d_pfit[i4 + i] = res[i]


class CopyFromOriginal(Equation):
    """the ghost-update pattern (iisph.py:243-261, gas_dynamics UpdateGhostProps):
    particles flagged as images copy properties from the particle they are an
    image of -- a read of the destination array at a run-time index held in an
    integer property"""

    def initialize(self, d_idx, d_image, d_orig_idx, d_rho, d_p, d_q):
        idx = declare('int')
        if d_image[d_idx] > 0.5:
            idx = d_orig_idx[d_idx]
            d_rho[d_idx] = d_rho[idx]
            d_p[d_idx] = 2.0 * d_p[idx]
            d_q[d_idx] = d_rho[idx] + d_p[idx]


class DensityWithGradH(Equation):
    """the adaptive-h summation pattern of gas_dynamics/basic.py:123-160: density
    with the kernel at h_i, and the dW/dh sums that correct for a varying h
    (symbols GHI, GHJ, GHIJ)"""

    def initialize(self, d_idx, d_rho, d_dwdh, d_q):
        d_rho[d_idx] = 0.0
        d_dwdh[d_idx] = 0.0
        d_q[d_idx] = 0.0

    def loop(self, d_idx, s_idx, d_rho, d_dwdh, d_q, s_m, WI, GHI, GHJ, GHIJ):
        d_rho[d_idx] += s_m[s_idx] * WI
        d_dwdh[d_idx] += s_m[s_idx] * GHI
        d_q[d_idx] += s_m[s_idx] * (GHJ - 0.5 * GHIJ)


class SolveSystems(Equation):
    """one n x n linear system per particle (n = self.n <= 4), matrix and
    right-hand side in strided properties: the reference's wc/linalg.py helpers'
    use case (sph/tests/test_linalg.py) on the device"""

    def __init__(self, dest, sources, n=4):
        self.n = n
        super(SolveSystems, self).__init__(dest, sources)

    def _get_helpers_(self):
        return [stack_columns, eliminate]

    def loop(self, d_idx, d_amat, d_bvec, d_pfit, d_q):
        a = declare('matrix(16)')
        b = declare('matrix(4)')
        aug = declare('matrix(20)')
        x = declare('matrix(4)')
        i, n = declare('int', 2)
        n = self.n
        for i in range(16):
            a[i] = d_amat[16 * d_idx + i]
        for i in range(4):
            b[i] = d_bvec[4 * d_idx + i]
            x[i] = 0.0
        stack_columns(a, b, n, 1, 4, aug)
        d_q[d_idx] = eliminate(aug, n, 1, x)
        for i in range(4):
            d_pfit[4 * d_idx + i] = x[i]
