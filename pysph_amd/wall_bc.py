"""Solid-wall boundary conditions of the transport-velocity formulation
(Adami, Hu & Adams, J. Comput. Phys. 231 (2012) 7057 and 241 (2013) 292).

These equations have no hand-written pair kernel: their ``initialize / loop /
post_loop`` methods below are per-particle Python bodies that
``pysph_amd.codegen`` translates into a generated family on the ``k_pair_wave``
skeleton.  Class names, constructor arguments and the
array names in the method signatures are the interface of
pysph/sph/wc/transport_velocity.py (``SetWallVelocity`` :84-134,
``SolidWallPressureBC`` :641-735, ``SolidWallNoSlipBC`` :548-638,
``ContinuitySolid`` :157-173, ``VolumeSummation`` :61-75), so that
``TVFScheme(fluids, solids)`` builds the reference's group list; the bodies are
written from the papers' formulae:

* wall velocity (2012, eqs. 22-23): the kernel-weighted mean of the fluid
  velocity seen by a wall particle, ``v~ = sum_f v_f W / sum_f W``, mirrored
  about the wall's own velocity, ``v_g = 2 v_w - v~``;
* wall pressure (2012, eq. 27): ``p_w = [sum_f p_f W + (g - a_w) . sum_f rho_f
  r_wf W] / sum_f W`` and the density that the linear state equation assigns
  to it (eq. 28);
* no-slip force (2013, eq. 8, third term): the viscous pair force of the fluid
  equation evaluated with the mirrored wall velocity ``v_g``.
"""
from .equations import Equation


class SetWallVelocity(Equation):
    """Destination (wall) arrays carry uf, vf, wf (filtered fluid velocity),
    wij (the kernel sum) and ug, vg, wg (the mirrored velocity)."""

    def initialize(self, d_idx, d_uf, d_vf, d_wf, d_wij):
        d_wij[d_idx] = 0.0
        d_uf[d_idx] = 0.0
        d_vf[d_idx] = 0.0
        d_wf[d_idx] = 0.0

    def loop(self, d_idx, s_idx, d_uf, d_vf, d_wf, d_wij, s_u, s_v, s_w, WIJ):
        # numerators and the common denominator of the weighted mean
        d_uf[d_idx] += s_u[s_idx] * WIJ
        d_vf[d_idx] += s_v[s_idx] * WIJ
        d_wf[d_idx] += s_w[s_idx] * WIJ
        d_wij[d_idx] += WIJ

    def post_loop(self, d_idx, d_uf, d_vf, d_wf, d_wij, d_u, d_v, d_w,
                  d_ug, d_vg, d_wg):
        ksum = d_wij[d_idx]
        um = d_uf[d_idx]
        vm = d_vf[d_idx]
        wm = d_wf[d_idx]
        if ksum > 1e-12:
            # a wall particle with no fluid in reach keeps a zero mean
            um = um / ksum
            vm = vm / ksum
            wm = wm / ksum
        d_uf[d_idx] = um
        d_vf[d_idx] = vm
        d_wf[d_idx] = wm
        d_ug[d_idx] = 2 * d_u[d_idx] - um
        d_vg[d_idx] = 2 * d_v[d_idx] - vm
        d_wg[d_idx] = 2 * d_w[d_idx] - wm


class SolidWallPressureBC(Equation):
    """``d_au, d_av, d_aw`` are the prescribed accelerations of the wall."""

    def __init__(self, dest, sources, rho0, p0, b=1.0, gx=0.0, gy=0.0, gz=0.0):
        self.rho0 = rho0
        self.p0 = p0
        self.b = b
        self.gx = gx
        self.gy = gy
        self.gz = gz
        super(SolidWallPressureBC, self).__init__(dest, sources)

    def initialize(self, d_idx, d_p, d_wij):
        d_wij[d_idx] = 0.0
        d_p[d_idx] = 0.0

    def loop(self, d_idx, s_idx, d_p, d_wij, d_au, d_av, d_aw, s_p, s_rho,
             WIJ, XIJ):
        # (g - a_w) . r_wf: the hydrostatic head between fluid and wall particle
        head = (self.gx - d_au[d_idx]) * XIJ[0] + \
            (self.gy - d_av[d_idx]) * XIJ[1] + \
            (self.gz - d_aw[d_idx]) * XIJ[2]
        d_p[d_idx] += s_p[s_idx] * WIJ + s_rho[s_idx] * head * WIJ
        d_wij[d_idx] += WIJ

    def post_loop(self, d_idx, d_p, d_wij, d_rho):
        ksum = d_wij[d_idx]
        pw = d_p[d_idx]
        if ksum > 1e-14:
            pw = pw / ksum
        d_p[d_idx] = pw
        # invert p = p0 (rho / rho0 - b)
        d_rho[d_idx] = self.rho0 * (pw / self.p0 + self.b)


class SolidWallNoSlipBC(Equation):
    """Destination = fluid, sources = walls (which carry ug, vg, wg)."""

    def __init__(self, dest, sources, nu):
        self.nu = nu
        super(SolidWallNoSlipBC, self).__init__(dest, sources)

    def initialize(self, d_idx, d_au, d_av, d_aw):
        d_au[d_idx] = 0.0
        d_av[d_idx] = 0.0
        d_aw[d_idx] = 0.0

    def loop(self, d_idx, s_idx, d_m, d_rho, d_V, d_u, d_v, d_w, d_au, d_av,
             d_aw, s_rho, s_V, s_ug, s_vg, s_wg, DWIJ, XIJ, R2IJ, EPS):
        # harmonic mean of the dynamic viscosities eta = nu rho
        ea = self.nu * d_rho[d_idx]
        eb = self.nu * s_rho[s_idx]
        eta = 2 * (ea * eb) / (ea + eb)
        # squared particle volumes (V holds the number density 1 / volume)
        va = 1. / d_V[d_idx]
        vb = 1. / s_V[s_idx]
        vol2 = va * va + vb * vb
        xdw = XIJ[0] * DWIJ[0] + XIJ[1] * DWIJ[1] + XIJ[2] * DWIJ[2]
        coef = 1. / d_m[d_idx] * vol2 * (eta * xdw / (R2IJ + EPS))
        d_au[d_idx] += coef * (d_u[d_idx] - s_ug[s_idx])
        d_av[d_idx] += coef * (d_v[d_idx] - s_vg[s_idx])
        d_aw[d_idx] += coef * (d_w[d_idx] - s_wg[s_idx])


class ContinuitySolid(Equation):
    """Density rate of a fluid particle from the walls, relative velocity taken
    against the mirrored wall velocity: ``rho_a sum_w V_w (v_a - v_g) . grad W``."""

    def loop(self, d_idx, s_idx, d_rho, d_arho, d_u, d_v, d_w, s_m, s_rho,
             s_ug, s_vg, s_wg, DWIJ):
        vol = s_m[s_idx] / s_rho[s_idx]
        du = d_u[d_idx] - s_ug[s_idx]
        dv = d_v[d_idx] - s_vg[s_idx]
        dw = d_w[d_idx] - s_wg[s_idx]
        flux = du * DWIJ[0] + dv * DWIJ[1] + dw * DWIJ[2]
        d_arho[d_idx] += d_rho[d_idx] * vol * flux


class VolumeSummation(Equation):
    """Number density ``V_a = sum_b W_ab``."""

    def initialize(self, d_idx, d_V):
        d_V[d_idx] = 0.0

    def loop(self, d_idx, d_V, WIJ):
        d_V[d_idx] += WIJ
