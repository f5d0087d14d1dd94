"""TEST FIXTURE (not product code): steppers and integrators a USER of the
integrator protocol would write -- leap-frog, PEFRL and Euler, after
pysph/sph/integrator_step.py:22-35,708-830 and pysph/sph/integrator.py:426-517
-- used by tests/test_reference_integrators.py to exercise generated stage
kernels (user-defined ``IntegratorStep`` bodies) and overridden
``one_timestep`` sequences.  The product ships only the integrators of the hot
path's own examples (PEC / EPEC, pysph_amd/integrator.py)."""
from pysph_amd.integrator import Integrator, IntegratorStep, PECIntegrator


class LeapFrogStep(IntegratorStep):
    """integrator_step.py:708-730 (runs as a generated stepper)."""

    def stage1(self, d_idx, d_x, d_y, d_z, d_u, d_v, d_w, d_ax, d_ay, d_az, dt):
        d_x[d_idx] += 0.5 * dt * (d_u[d_idx] + d_ax[d_idx])
        d_y[d_idx] += 0.5 * dt * (d_v[d_idx] + d_ay[d_idx])
        d_z[d_idx] += 0.5 * dt * (d_w[d_idx] + d_az[d_idx])

    def stage2(self, d_idx, d_x, d_y, d_z, d_u, d_au, d_v, d_av, d_w, d_aw, d_ax, d_ay,
               d_az, d_rho, d_arho, d_e, d_ae, dt):
        d_u[d_idx] += dt * d_au[d_idx]
        d_v[d_idx] += dt * d_av[d_idx]
        d_w[d_idx] += dt * d_aw[d_idx]
        d_rho[d_idx] += dt * d_arho[d_idx]
        d_e[d_idx] += dt * d_ae[d_idx]
        d_x[d_idx] += 0.5 * dt * (d_u[d_idx] + d_ax[d_idx])
        d_y[d_idx] += 0.5 * dt * (d_v[d_idx] + d_ay[d_idx])
        d_z[d_idx] += 0.5 * dt * (d_w[d_idx] + d_az[d_idx])


class EulerStep(IntegratorStep):
    """integrator_step.py:22-35 (runs as a generated stepper)."""

    def stage1(self, d_idx, d_u, d_v, d_w, d_au, d_av, d_aw, d_x, d_y, d_z, d_rho,
               d_arho, dt):
        d_u[d_idx] += dt * d_au[d_idx]
        d_v[d_idx] += dt * d_av[d_idx]
        d_w[d_idx] += dt * d_aw[d_idx]
        d_x[d_idx] += dt * d_u[d_idx]
        d_y[d_idx] += dt * d_v[d_idx]
        d_z[d_idx] += dt * d_w[d_idx]
        d_rho[d_idx] += dt * d_arho[d_idx]


class LeapFrogIntegrator(PECIntegrator):
    """integrator.py:464-477."""

    def one_timestep(self, t, dt):
        self.stage1()
        self.update_domain()
        self.do_post_stage(0.5 * dt, 1)
        self.compute_accelerations()
        self.stage2()
        self.update_domain()
        self.do_post_stage(dt, 2)


class EulerIntegrator(Integrator):
    """integrator.py:426-437."""

    def one_timestep(self, t, dt):
        self.compute_accelerations()
        self.stage1()
        self.update_domain()
        self.do_post_stage(dt, 1)


class PEFRLStep(IntegratorStep):
    """Position-extended Forest-Ruth-like scheme of Omelyan, Mryglod & Folk,
    Comput. Phys. Commun. 146 (2002) 188 (integrator_step.py:738-830): five
    position sub-steps with weights (xi, chi, 1-2(xi+chi), chi, xi) and four
    velocity sub-steps with weights ((1-2 lam)/2, lam, lam, (1-2 lam)/2).
    Runs as a generated stepper."""

    def stage1(self, d_idx, d_x, d_y, d_z, d_u, d_v, d_w, d_ax, d_ay, d_az, dt):
        cx = 0.1786178958448091 * dt
        d_x[d_idx] += cx * (d_u[d_idx] + d_ax[d_idx])
        d_y[d_idx] += cx * (d_v[d_idx] + d_ay[d_idx])
        d_z[d_idx] += cx * (d_w[d_idx] + d_az[d_idx])

    def stage2(self, d_idx, d_x, d_y, d_z, d_u, d_au, d_v, d_av, d_w, d_aw, d_ax, d_ay,
               d_az, d_rho, d_arho, d_e, d_ae, dt):
        cv = 0.5 * (1.0 - 2.0 * (-0.2123418310626054)) * dt
        cx = -0.06626458266981849 * dt
        d_u[d_idx] += cv * d_au[d_idx]
        d_v[d_idx] += cv * d_av[d_idx]
        d_w[d_idx] += cv * d_aw[d_idx]
        d_rho[d_idx] += cv * d_arho[d_idx]
        d_e[d_idx] += cv * d_ae[d_idx]
        d_x[d_idx] += cx * (d_u[d_idx] + d_ax[d_idx])
        d_y[d_idx] += cx * (d_v[d_idx] + d_ay[d_idx])
        d_z[d_idx] += cx * (d_w[d_idx] + d_az[d_idx])

    def stage3(self, d_idx, d_x, d_y, d_z, d_u, d_au, d_v, d_av, d_w, d_aw, d_ax, d_ay,
               d_az, d_rho, d_arho, d_e, d_ae, dt):
        cv = -0.2123418310626054 * dt
        cx = (1.0 - 2.0 * (0.1786178958448091 + (-0.06626458266981849))) * dt
        d_u[d_idx] += cv * d_au[d_idx]
        d_v[d_idx] += cv * d_av[d_idx]
        d_w[d_idx] += cv * d_aw[d_idx]
        d_rho[d_idx] += cv * d_arho[d_idx]
        d_e[d_idx] += cv * d_ae[d_idx]
        d_x[d_idx] += cx * (d_u[d_idx] + d_ax[d_idx])
        d_y[d_idx] += cx * (d_v[d_idx] + d_ay[d_idx])
        d_z[d_idx] += cx * (d_w[d_idx] + d_az[d_idx])

    def stage4(self, d_idx, d_x, d_y, d_z, d_u, d_au, d_v, d_av, d_w, d_aw, d_ax, d_ay,
               d_az, d_rho, d_arho, d_e, d_ae, dt):
        cv = -0.2123418310626054 * dt
        cx = -0.06626458266981849 * dt
        d_u[d_idx] += cv * d_au[d_idx]
        d_v[d_idx] += cv * d_av[d_idx]
        d_w[d_idx] += cv * d_aw[d_idx]
        d_rho[d_idx] += cv * d_arho[d_idx]
        d_e[d_idx] += cv * d_ae[d_idx]
        d_x[d_idx] += cx * (d_u[d_idx] + d_ax[d_idx])
        d_y[d_idx] += cx * (d_v[d_idx] + d_ay[d_idx])
        d_z[d_idx] += cx * (d_w[d_idx] + d_az[d_idx])

    def stage5(self, d_idx, d_x, d_y, d_z, d_u, d_au, d_v, d_av, d_w, d_aw, d_ax, d_ay,
               d_az, d_rho, d_arho, d_e, d_ae, dt):
        cv = 0.5 * (1.0 - 2.0 * (-0.2123418310626054)) * dt
        cx = 0.1786178958448091 * dt
        d_u[d_idx] += cv * d_au[d_idx]
        d_v[d_idx] += cv * d_av[d_idx]
        d_w[d_idx] += cv * d_aw[d_idx]
        d_rho[d_idx] += cv * d_arho[d_idx]
        d_e[d_idx] += cv * d_ae[d_idx]
        d_x[d_idx] += cx * (d_u[d_idx] + d_ax[d_idx])
        d_y[d_idx] += cx * (d_v[d_idx] + d_ay[d_idx])
        d_z[d_idx] += cx * (d_w[d_idx] + d_az[d_idx])


class PEFRLIntegrator(Integrator):
    """integrator.py:481-517: the stage times are the cumulative position
    weights xi, xi+chi, 1-(xi+chi), 1-xi, 1."""

    def one_timestep(self, t, dt):
        self.stage1()
        self.update_domain()
        self.do_post_stage(0.1786178958448091 * dt, 1)
        self.compute_accelerations()
        self.stage2()
        self.update_domain()
        self.do_post_stage(0.1123533131749906 * dt, 2)
        self.compute_accelerations()
        self.stage3()
        self.update_domain()
        self.do_post_stage(0.8876466868250094 * dt, 3)
        self.compute_accelerations()
        self.stage4()
        self.update_domain()
        self.do_post_stage(0.8213821041551909 * dt, 4)
        self.compute_accelerations()
        self.stage5()
        self.update_domain()
        self.do_post_stage(dt, 5)
