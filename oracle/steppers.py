"""CPU ORACLE (test infrastructure): numpy restatement of the reference's
integrator steppers and of one EPEC/PEC time step, operating in place on host
ParticleArrays (real particles only, integrator_cython.mako:103-104).

Pinned against the reference's own ``WCSPHStep`` / ``TransportVelocityStep``
Python methods by tests/golden/steppers.npz (tests/golden/make_golden.py).
"""


def wcsph_initialize(pa):
    """WCSPHStep.initialize  pysph/sph/integrator_step.py:51-61"""
    n = pa.get_number_of_particles(True)
    for a, b in (('x0', 'x'), ('y0', 'y'), ('z0', 'z'), ('u0', 'u'),
                 ('v0', 'v'), ('w0', 'w'), ('rho0', 'rho')):
        pa.properties[a][:n] = pa.properties[b][:n]


def wcsph_stage(pa, dt, stage):
    """WCSPHStep.stage1 (:63-76) / stage2 (:78-92)"""
    n = pa.get_number_of_particles(True)
    p = pa.properties
    f = 0.5 * dt if stage == 1 else dt
    for q, q0, aq in (('u', 'u0', 'au'), ('v', 'v0', 'av'), ('w', 'w0', 'aw'),
                      ('x', 'x0', 'ax'), ('y', 'y0', 'ay'), ('z', 'z0', 'az'),
                      ('rho', 'rho0', 'arho')):
        p[q][:n] = p[q0][:n] + f * p[aq][:n]


def tvf_stage1(pa, dt):
    """TransportVelocityStep.stage1  integrator_step.py:268-285"""
    n = pa.get_number_of_particles(True)
    p = pa.properties
    dtb2 = 0.5 * dt
    for q, aq in (('u', 'au'), ('v', 'av'), ('w', 'aw')):
        p[q][:n] += dtb2 * p[aq][:n]
    for qh, q, aq in (('uhat', 'u', 'auhat'), ('vhat', 'v', 'avhat'),
                      ('what', 'w', 'awhat')):
        p[qh][:n] = p[q][:n] + dtb2 * p[aq][:n]
    for x, qh in (('x', 'uhat'), ('y', 'vhat'), ('z', 'what')):
        p[x][:n] += dt * p[qh][:n]


def tvf_stage2(pa, dt):
    """TransportVelocityStep.stage2  integrator_step.py:287-299"""
    n = pa.get_number_of_particles(True)
    p = pa.properties
    dtb2 = 0.5 * dt
    for q, aq in (('u', 'au'), ('v', 'av'), ('w', 'aw')):
        p[q][:n] += dtb2 * p[aq][:n]
    p['vmag2'][:n] = p['u'][:n] * p['u'][:n] + p['v'][:n] * p['v'][:n] + \
        p['w'][:n] * p['w'][:n]


def epec_step(arrays, nnps, ev, t, dt, domain=None):
    """EPECIntegrator.one_timestep with WCSPHStep  (integrator.py:401-420)."""
    def accel():
        nnps.update()
        ev.compute(t, dt)

    def upd():
        if domain is not None:
            domain.update()
    for pa in arrays:
        wcsph_initialize(pa)
    accel()
    for pa in arrays:
        wcsph_stage(pa, dt, 1)
    upd()
    accel()
    for pa in arrays:
        wcsph_stage(pa, dt, 2)
    upd()


def pec_step_tvf(arrays, nnps, ev, t, dt, domain=None):
    """Integrator.one_timestep (PEC) with TransportVelocityStep
    (integrator.py:227-246)."""
    for pa in arrays:
        tvf_stage1(pa, dt)
    if domain is not None:
        domain.update()
    nnps.update()
    ev.compute(t, dt)
    for pa in arrays:
        tvf_stage2(pa, dt)
    if domain is not None:
        domain.update()


_SM = [('x', 'x0', 'ax'), ('y', 'y0', 'ay'), ('z', 'z0', 'az'), ('u', 'u0', 'au'),
       ('v', 'v0', 'av'), ('w', 'w0', 'aw'), ('rho', 'rho0', 'arho'),
       ('e', 'e0', 'ae'), ('s00', 's000', 'as00'), ('s01', 's010', 'as01'),
       ('s02', 's020', 'as02'), ('s11', 's110', 'as11'), ('s12', 's120', 'as12'),
       ('s22', 's220', 'as22')]


def solid_initialize(pa):
    """SolidMechStep.initialize  pysph/sph/integrator_step.py:175-197"""
    n = pa.get_number_of_particles(True)
    for q, q0, _ in _SM:
        pa.properties[q0][:n] = pa.properties[q][:n]


def solid_stage(pa, dt, stage):
    """SolidMechStep.stage1 (:199-226) / stage2 (:228-255)"""
    n = pa.get_number_of_particles(True)
    p = pa.properties
    f = 0.5 * dt if stage == 1 else dt
    for q, q0, aq in _SM:
        p[q][:n] = p[q0][:n] + f * p[aq][:n]
