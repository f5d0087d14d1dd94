// sph_integrate.hip -- integrator stage sweeps on the device.
//
// Replaces the per-particle loops the reference generates from
// pysph/sph/integrator_cython.mako:87-113 around the IntegratorStep methods of
// pysph/sph/integrator_step.py (WCSPHStep :38-93, TransportVelocityStep
// :257-299).  Real particles only, as in the template (:103-104).  Purely
// streaming (HBM-bound): ~112 B (initialize) / ~168 B (stage) per particle.
#include "sph_internal.h"

struct StageArgs {
    int stepper, stage;
    double dt;
    size_t n;
    double *p[SPH_PROP_COUNT];
};

__global__ __launch_bounds__(256) void k_stage(StageArgs a)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    double **p = a.p;
    const double dt = a.dt, dtb2 = 0.5 * a.dt;
    if (a.stepper == SPH_STEP_WCSPH) {
        if (a.stage == 0) { // integrator_step.py:51-61
            p[SPH_X0][i] = p[SPH_X][i]; p[SPH_Y0][i] = p[SPH_Y][i]; p[SPH_Z0][i] = p[SPH_Z][i];
            p[SPH_U0][i] = p[SPH_U][i]; p[SPH_V0][i] = p[SPH_V][i]; p[SPH_W0][i] = p[SPH_W][i];
            p[SPH_RHO0][i] = p[SPH_RHO][i];
        } else { // stage1 :63-76 (dt/2), stage2 :78-92 (dt)
            const double f = a.stage == 1 ? dtb2 : dt;
            p[SPH_U][i] = p[SPH_U0][i] + f * p[SPH_AU][i];
            p[SPH_V][i] = p[SPH_V0][i] + f * p[SPH_AV][i];
            p[SPH_W][i] = p[SPH_W0][i] + f * p[SPH_AW][i];
            p[SPH_X][i] = p[SPH_X0][i] + f * p[SPH_AX][i];
            p[SPH_Y][i] = p[SPH_Y0][i] + f * p[SPH_AY][i];
            p[SPH_Z][i] = p[SPH_Z0][i] + f * p[SPH_AZ][i];
            p[SPH_RHO][i] = p[SPH_RHO0][i] + f * p[SPH_ARHO][i];
        }
    } else if (a.stepper == SPH_STEP_SOLID_MECH) { // integrator_step.py:175-255
        constexpr int q[]  = {SPH_X, SPH_Y, SPH_Z, SPH_U, SPH_V, SPH_W, SPH_RHO, SPH_E,
                                 SPH_S00, SPH_S01, SPH_S02, SPH_S11, SPH_S12, SPH_S22};
        constexpr int q0[] = {SPH_X0, SPH_Y0, SPH_Z0, SPH_U0, SPH_V0, SPH_W0, SPH_RHO0, SPH_E0,
                                 SPH_S000, SPH_S010, SPH_S020, SPH_S110, SPH_S120, SPH_S220};
        constexpr int aq[] = {SPH_AX, SPH_AY, SPH_AZ, SPH_AU, SPH_AV, SPH_AW, SPH_ARHO, SPH_AE,
                                 SPH_AS00, SPH_AS01, SPH_AS02, SPH_AS11, SPH_AS12, SPH_AS22};
        if (a.stage == 0) {
#pragma unroll
            for (int k = 0; k < 14; k++) p[q0[k]][i] = p[q[k]][i];
        } else {
            const double f = a.stage == 1 ? dtb2 : dt;
#pragma unroll
            for (int k = 0; k < 14; k++) p[q[k]][i] = p[q0[k]][i] + f * p[aq[k]][i];
        }
    } else if (a.stepper == SPH_STEP_TVF) {
        if (a.stage == 1) { // :268-285
            double u = p[SPH_U][i] + dtb2 * p[SPH_AU][i];
            double v = p[SPH_V][i] + dtb2 * p[SPH_AV][i];
            double w = p[SPH_W][i] + dtb2 * p[SPH_AW][i];
            p[SPH_U][i] = u; p[SPH_V][i] = v; p[SPH_W][i] = w;
            double uh = u + dtb2 * p[SPH_AUHAT][i];
            double vh = v + dtb2 * p[SPH_AVHAT][i];
            double wh = w + dtb2 * p[SPH_AWHAT][i];
            p[SPH_UHAT][i] = uh; p[SPH_VHAT][i] = vh; p[SPH_WHAT][i] = wh;
            p[SPH_X][i] += dt * uh;
            p[SPH_Y][i] += dt * vh;
            p[SPH_Z][i] += dt * wh;
        } else if (a.stage == 2) { // :287-299
            double u = p[SPH_U][i] + dtb2 * p[SPH_AU][i];
            double v = p[SPH_V][i] + dtb2 * p[SPH_AV][i];
            double w = p[SPH_W][i] + dtb2 * p[SPH_AW][i];
            p[SPH_U][i] = u; p[SPH_V][i] = v; p[SPH_W][i] = w;
            p[SPH_VMAG2][i] = u * u + v * v + w * w;
        }
    }
}

extern "C" int sph_integrate_stage(sph_ctx *c, int id, int stepper, int stage, double dt)
{
    if (!c || id < 0 || id >= SPH_MAX_ARRAYS || stage < 0 || stage > 2) { sph_set_error("sph_integrate_stage: bad arguments"); return SPH_ERR_ARG; }
    HIP_TRY(hipSetDevice(c->device));
    DevArray &A = c->arr[id];
    static const int wc0[] = {SPH_X, SPH_Y, SPH_Z, SPH_U, SPH_V, SPH_W, SPH_RHO, SPH_X0, SPH_Y0, SPH_Z0, SPH_U0, SPH_V0, SPH_W0, SPH_RHO0, -1};
    static const int wc1[] = {SPH_X, SPH_Y, SPH_Z, SPH_U, SPH_V, SPH_W, SPH_RHO, SPH_X0, SPH_Y0, SPH_Z0, SPH_U0, SPH_V0, SPH_W0, SPH_RHO0,
                              SPH_AU, SPH_AV, SPH_AW, SPH_AX, SPH_AY, SPH_AZ, SPH_ARHO, -1};
    static const int tv1[] = {SPH_X, SPH_Y, SPH_Z, SPH_U, SPH_V, SPH_W, SPH_AU, SPH_AV, SPH_AW, SPH_UHAT, SPH_VHAT, SPH_WHAT,
                              SPH_AUHAT, SPH_AVHAT, SPH_AWHAT, -1};
    static const int tv2[] = {SPH_U, SPH_V, SPH_W, SPH_AU, SPH_AV, SPH_AW, SPH_VMAG2, -1};
    static const int sm0[] = {SPH_X, SPH_Y, SPH_Z, SPH_U, SPH_V, SPH_W, SPH_RHO, SPH_E, SPH_S00, SPH_S01, SPH_S02, SPH_S11, SPH_S12, SPH_S22,
                              SPH_X0, SPH_Y0, SPH_Z0, SPH_U0, SPH_V0, SPH_W0, SPH_RHO0, SPH_E0, SPH_S000, SPH_S010, SPH_S020, SPH_S110,
                              SPH_S120, SPH_S220, -1};
    static const int sm1[] = {SPH_X, SPH_Y, SPH_Z, SPH_U, SPH_V, SPH_W, SPH_RHO, SPH_E, SPH_S00, SPH_S01, SPH_S02, SPH_S11, SPH_S12, SPH_S22,
                              SPH_X0, SPH_Y0, SPH_Z0, SPH_U0, SPH_V0, SPH_W0, SPH_RHO0, SPH_E0, SPH_S000, SPH_S010, SPH_S020, SPH_S110,
                              SPH_S120, SPH_S220, SPH_AX, SPH_AY, SPH_AZ, SPH_AU, SPH_AV, SPH_AW, SPH_ARHO, SPH_AE, SPH_AS00, SPH_AS01,
                              SPH_AS02, SPH_AS11, SPH_AS12, SPH_AS22, -1};
    const int *need = nullptr;
    if (stepper == SPH_STEP_WCSPH) need = stage == 0 ? wc0 : wc1;
    else if (stepper == SPH_STEP_SOLID_MECH) need = stage == 0 ? sm0 : sm1;
    else if (stepper == SPH_STEP_TVF) { if (stage == 0) return SPH_OK; need = stage == 1 ? tv1 : tv2; }
    else { sph_set_error("sph_integrate_stage: unknown stepper %d", stepper); return SPH_ERR_UNSUPPORTED; }
    for (const int *q = need; *q >= 0; q++) SPH_TRY(sph_array_ensure_prop(c, id, *q));
    if (A.n_real == 0) return SPH_OK;
    StageArgs a;
    a.stepper = stepper; a.stage = stage; a.dt = dt; a.n = A.n_real;
    for (int k = 0; k < SPH_PROP_COUNT; k++) a.p[k] = A.prop[k];
    ScopedTimer tm(c, T_STAGE);
    hipLaunchKernelGGL(k_stage, dim3(div_up(A.n_real, 256)), dim3(256), 0, c->stream, a);
    if (stepper == SPH_STEP_TVF ? stage == 1 : stage > 0) c->nnps_valid = false; // positions moved
    return SPH_OK;
}
