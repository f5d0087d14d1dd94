/*
 * sph_oracle.h -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * A plain-C restatement of the reference's (pypr/pysph) hot path:
 *   - cell-list neighbour search  pysph/base/linked_list_nnps.pyx:92-196,235-383
 *                                 pysph/base/nnps_base.pyx:942-978,1471-1575
 *                                 pysph/base/nnps_base.pxd:39-135
 *   - the generated loop nest     pysph/sph/acceleration_eval_cython.mako:10-154
 *   - pair symbols                pysph/sph/equation.py:188-297
 *   - SPH kernels                 pysph/base/kernels.py
 *   - equation bodies             pysph/sph/wc/basic.py, basic_equations.py,
 *                                 wc/transport_velocity.py, solid_mech/basic.py
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (pysph_amd + libsphhip.so) never does.
 *
 * Parity pin: validated against golden vectors produced by executing the
 * reference's own Python Equation/kernel classes (tests/golden/make_golden.py).
 */
#ifndef SPH_ORACLE_H
#define SPH_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_ARRAYS 8
#define ORC_MAX_PAR 16

/* particle properties (all double, SoA) */
enum orc_prop {
    OP_X, OP_Y, OP_Z, OP_U, OP_V, OP_W, OP_H, OP_M, OP_RHO, OP_P, OP_CS,
    OP_ARHO, OP_AU, OP_AV, OP_AW, OP_AX, OP_AY, OP_AZ, OP_DT_CFL, OP_DT_FORCE,
    /* transport-velocity formulation */
    OP_VOL /* 'V' */, OP_UHAT, OP_VHAT, OP_WHAT, OP_AUHAT, OP_AVHAT, OP_AWHAT,
    /* elastic solids: velocity gradient, deviatoric stress, artificial stress */
    OP_V00, OP_V01, OP_V02, OP_V10, OP_V11, OP_V12, OP_V20, OP_V21, OP_V22,
    OP_S00, OP_S01, OP_S02, OP_S11, OP_S12, OP_S22,
    OP_AS00, OP_AS01, OP_AS02, OP_AS11, OP_AS12, OP_AS22,
    OP_R00, OP_R01, OP_R02, OP_R11, OP_R12, OP_R22,
    OP_E /* thermal energy */, OP_AE,
    OP_COUNT
};

enum orc_kernel_kind {
    OK_CUBIC_SPLINE = 1,
    OK_WENDLAND_QUINTIC = 2,
    OK_QUINTIC_SPLINE = 3,
    OK_GAUSSIAN = 4
};

enum orc_eq_kind {
    OE_TAIT_EOS = 1,            /* par: rho0 c0 gamma p0 */
    OE_TAIT_EOS_HG,             /* par: rho0 c0 gamma */
    OE_CONTINUITY,              /* no par */
    OE_MOMENTUM,                /* par: c0 alpha beta gx gy gz tensile_correction */
    OE_XSPH,                    /* par: eps */
    OE_SUMMATION_DENSITY,       /* basic_equations.SummationDensity */
    OE_TVF_SUMMATION_DENSITY,   /* transport_velocity.SummationDensity */
    OE_TVF_STATE_EQUATION,      /* par: p0 rho0 b */
    OE_TVF_MOM_PRESSURE,        /* par: pb gx gy gz tdamp */
    OE_TVF_MOM_VISCOSITY,       /* par: nu */
    OE_TVF_MOM_ART_VISCOSITY,   /* par: c0 alpha */
    OE_TVF_MOM_ART_STRESS,      /* no par */
    OE_ISOTHERMAL_EOS,          /* par: rho0 c0 p0 */
    OE_MONAGHAN_ART_VISCOSITY,  /* par: alpha beta */
    OE_VELOCITY_GRADIENT_3D,
    OE_VELOCITY_GRADIENT_2D,
    OE_HOOKES_DEVIATORIC_STRESS_RATE, /* par: G (shear modulus) */
    OE_MOMENTUM_WITH_STRESS,    /* par: wdeltap n */
    OE_MONAGHAN_ART_STRESS,     /* par: eps */
    OE_ENERGY_WITH_STRESS,      /* par: alpha beta eta */
    OE_SOLID_ISOTHERMAL_EOS     /* solid_mech/basic.py:93-101  par: c0_ref rho_ref */
};

typedef struct {
    long n;        /* all particles (real + ghost/remote) */
    long n_real;   /* real particles come first */
    double *p[OP_COUNT];
} orc_array;

typedef struct {
    int kind;
    int dim;
    double fac;
    double radius_scale;
    double deltap;
} orc_kernel;

typedef struct {
    int kind;
    int dest;                 /* index into the array list */
    int nsrc;                 /* 0 => equation with no sources */
    int src[ORC_MAX_ARRAYS];
    double par[ORC_MAX_PAR];
} orc_equation;

typedef struct {
    int real;                 /* Group(real=...)   equation.py:457 */
    long start_idx;           /* Group(start_idx=) */
    long stop_idx;            /* Group(stop_idx=), <0 => None */
    int neq;
    const orc_equation *eqs;
} orc_group;

typedef struct orc_nnps orc_nnps;

/* NNPS ---------------------------------------------------------------- */
orc_nnps *orc_nnps_create(int dim, int narrays, double radius_scale);
void orc_nnps_destroy(orc_nnps *);
void orc_nnps_set_array(orc_nnps *, int index, const orc_array *arr);
/* DomainManager.update (cell size) + NNPS.update (bounds, refresh, bin). */
int orc_nnps_update(orc_nnps *);
/* scalars out: cell_size, hmin, xmin[3], xmax[3]; ints: ncx ncy ncz n_cells */
void orc_nnps_info(const orc_nnps *, double *d8, long *i4);
/* get_nearest_particles(src, dst, d_idx): returns count; writes <= cap ids */
long orc_nnps_neighbors(const orc_nnps *, int src, int dst, long d_idx,
                        unsigned *out, long cap);
/* brute force (nnps_base.pyx:1325-1366), same criterion, ascending order */
long orc_nnps_brute_force(const orc_nnps *, int src, int dst, long d_idx,
                          unsigned *out, long cap);
/* CSR of every destination particle (counts pass if nbrs==NULL) */
long orc_nnps_csr(const orc_nnps *, int src, int dst, unsigned *start,
                  unsigned *nbrs, int nthreads);

/* AccelerationEval.compute ------------------------------------------- */
int orc_compute(orc_nnps *, const orc_kernel *, const orc_group *groups,
                int ngroups, double t, double dt, int nthreads);

/* kernel functions, exposed for KATs (test_kernel.py) */
double orc_kernel_w(const orc_kernel *, double rij, double h);
double orc_kernel_dwdq(const orc_kernel *, double rij, double h);
void orc_kernel_gradient(const orc_kernel *, const double *xij, double rij,
                         double h, double *grad);

/* linalg3.eigen_decomposition on a row-major 3x3 (KAT hook) */
void orc_eigen3(const double *A9, double *V9, double *d3);

const char *orc_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
