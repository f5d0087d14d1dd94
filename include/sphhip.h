/*
 * sphhip.h -- C-ABI of libsphhip.so, the MI355X (gfx950) SPH acceleration-eval
 * backend.  Plain pointers and sizes only; every function returns 0 on success
 * or a negative sph_status (message via sph_last_error()).  No exceptions cross
 * the boundary.  A context belongs to one GPU and one host thread.
 *
 * What each entry point replaces in the reference (pypr/pysph; paths relative
 * to the reference root) is cited at its declaration.  The reference-side
 * binding a maintainer would add is shown in INTEGRATION.md.
 */
#ifndef SPHHIP_H
#define SPHHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPH_MAX_ARRAYS 8
#define SPH_MAX_PAR 16
#define SPH_MAX_EQS 64

typedef struct sph_ctx sph_ctx;

enum sph_status {
    SPH_OK = 0,
    SPH_ERR_HIP = -1,          /* a HIP runtime call failed */
    SPH_ERR_ARG = -2,          /* bad argument */
    SPH_ERR_CELLS = -3,        /* LinkedListNNPS: bad number of cells (linked_list_nnps.pyx:307-343) */
    SPH_ERR_UNSUPPORTED = -4,  /* equation / kernel combination without a hand-written kernel */
    SPH_ERR_MISSING_PROP = -5, /* a property needed by an equation was never registered */
    SPH_ERR_STATE = -6         /* e.g. eval before nnps update */
};

/* Particle properties with device storage (all fp64, SoA).  Names via
 * sph_prop_id(); mirrors the property names of pysph/base/utils.py:36-39,
 * 152-155 (WCSPH) and get_particle_array_tvf_fluid.                      */
enum sph_prop {
    SPH_X, SPH_Y, SPH_Z, SPH_U, SPH_V, SPH_W, SPH_H, SPH_M, SPH_RHO, SPH_P, SPH_CS,
    SPH_ARHO, SPH_AU, SPH_AV, SPH_AW, SPH_AX, SPH_AY, SPH_AZ, SPH_DT_CFL, SPH_DT_FORCE,
    SPH_VOL /* 'V' */, SPH_UHAT, SPH_VHAT, SPH_WHAT, SPH_AUHAT, SPH_AVHAT, SPH_AWHAT,
    SPH_X0, SPH_Y0, SPH_Z0, SPH_U0, SPH_V0, SPH_W0, SPH_RHO0, SPH_VMAG2,
    /* elastic solids (pysph/sph/solid_mech/basic.py:34-60) */
    SPH_E, SPH_AE, SPH_E0,
    SPH_V00, SPH_V01, SPH_V02, SPH_V10, SPH_V11, SPH_V12, SPH_V20, SPH_V21, SPH_V22,
    SPH_S00, SPH_S01, SPH_S02, SPH_S11, SPH_S12, SPH_S22,
    SPH_AS00, SPH_AS01, SPH_AS02, SPH_AS11, SPH_AS12, SPH_AS22,
    SPH_R00, SPH_R01, SPH_R02, SPH_R11, SPH_R12, SPH_R22,
    SPH_S000, SPH_S010, SPH_S020, SPH_S110, SPH_S120, SPH_S220,
    /* user properties: slots named at run time with sph_prop_register (the
     * reference's ParticleArray.add_property for equation-specific arrays,
     * e.g. uf, ug, wij of the TVF wall equations)                           */
    SPH_USER0,
    SPH_PROP_COUNT = SPH_USER0 + 96
};

/* pysph/base/kernels.py class -> id */
enum sph_kernel_kind {
    SPH_K_CUBIC_SPLINE = 1,     /* kernels.py:29 */
    SPH_K_WENDLAND_QUINTIC = 2, /* kernels.py:274 */
    SPH_K_QUINTIC_SPLINE = 3,   /* kernels.py:1050 */
    SPH_K_GAUSSIAN = 4          /* kernels.py:830 */
};

/* Kernel object fields copied by value, as the generated Cython does
 * (acceleration_eval_cython_helper.py:240-245).                          */
typedef struct {
    int kind;
    int dim;
    double fac;
    double radius_scale;
    double deltap;
} sph_kernel;

/* Equation class -> id; `par` order listed per id (equation __init__ args). */
enum sph_eq_kind {
    SPH_EQ_TAIT_EOS = 1,              /* wc/basic.py:9      par: rho0 c0 gamma p0 */
    SPH_EQ_TAIT_EOS_HG = 2,           /* wc/basic.py:68     par: rho0 c0 gamma */
    SPH_EQ_CONTINUITY = 3,            /* basic_equations.py:177 */
    SPH_EQ_MOMENTUM = 4,              /* wc/basic.py:129    par: c0 alpha beta gx gy gz tensile */
    SPH_EQ_XSPH = 5,                  /* basic_equations.py:260  par: eps */
    SPH_EQ_SUMMATION_DENSITY = 6,     /* basic_equations.py:19 */
    SPH_EQ_TVF_SUMMATION_DENSITY = 7, /* wc/transport_velocity.py:24 */
    SPH_EQ_TVF_STATE_EQUATION = 8,    /* :190  par: p0 rho0 b */
    SPH_EQ_TVF_MOM_PRESSURE = 9,      /* :219  par: pb gx gy gz tdamp */
    SPH_EQ_TVF_MOM_VISCOSITY = 10,    /* :328  par: nu */
    SPH_EQ_TVF_MOM_ART_VISCOSITY = 11,/* :389  par: c0 alpha */
    SPH_EQ_TVF_MOM_ART_STRESS = 12,   /* :439 */
    SPH_EQ_ISOTHERMAL_EOS = 13,       /* basic_equations.py:151  par: rho0 c0 p0 */
    SPH_EQ_MONAGHAN_ART_VISCOSITY = 14,/* basic_equations.py:195  par: alpha beta */
    SPH_EQ_VELOCITY_GRADIENT_3D = 15, /* basic_equations.py:101 */
    SPH_EQ_VELOCITY_GRADIENT_2D = 16, /* basic_equations.py:63 */
    SPH_EQ_HOOKES_DEVIATORIC_STRESS_RATE = 17, /* solid_mech/basic.py:390  par: G (= d_G[0]) */
    SPH_EQ_MOMENTUM_WITH_STRESS = 18, /* solid_mech/basic.py:245  par: wdeltap n (= d_wdeltap[0], d_n[0]) */
    SPH_EQ_MONAGHAN_ART_STRESS = 19,  /* solid_mech/basic.py:104  par: eps */
    SPH_EQ_SOLID_ISOTHERMAL_EOS = 21  /* solid_mech/basic.py:93   par: c0_ref rho_ref */
};

/* One Equation(dest, sources) instance: pysph/sph/equation.py:392-420. */
typedef struct {
    int kind;
    int dest;                  /* array id */
    int nsrc;                  /* 0: equation without sources */
    int src[SPH_MAX_ARRAYS];   /* array ids, user order */
    double par[SPH_MAX_PAR];
} sph_equation;

/* One (leaf) Group: pysph/sph/equation.py:457-561.  `pre/post/condition/
 * iterate/update_nnps` are host-side control and stay in the Python caller.
 * ZERO-INITIALISE the struct (memset / `sph_group g = {0}`) before filling it:
 * the optional promises at its end grow with the library, and a field left
 * as stack garbage would be read as a promise.                              */
typedef struct {
    int real;                  /* loop over real particles only */
    long start_idx;            /* D_START_IDX  (acceleration_eval_cython_helper.py:263-268) */
    long stop_idx;             /* < 0: None -> dst.size(real) */
    int neq;
    const sph_equation *eqs;
    /* Optional promises of the caller about this group inside ONE evaluation
     * (all zero: none).  They only select faster schedules; results are the
     * same.  The Python host derives them from the structure of the group list
     * (pysph_amd/acceleration_eval.py: annotate_plan).
     * src_eos = 1: p and cs of every array this group's pair loops read are the
     *   Tait EOS of its rho with eos_par = {rho0, c0, gamma, p0}, as an
     *   SPH_EQ_TAIT_EOS[_HG] equation over ALL its particles left them (the
     *   group before this one) -- the pair kernel may recompute them from rho
     *   instead of gathering them (64-byte records).
     * src_eos = 2: p and V of every array this group's pair loops read are
     *   p = p0 (rho / rho0 - b) and V = rho / m with eos_par = {p0, rho0, b, -}, as
     *   SPH_EQ_TVF_STATE_EQUATION (the group before this one) and
     *   SPH_EQ_TVF_SUMMATION_DENSITY (earlier in this evaluation), each over ALL
     *   particles, left them -- the TVF force kernel may recompute both from rho
     *   (80-byte records) when the arrays have one mass each.
     * nl_mode: reuse of the neighbour lists between the pair passes of one
     *   evaluation (the reference's NeighborCache, nnps_base.pyx:1144-1257):
     *   1 = a later group of this evaluation loops over the same (destination,
     *   single source): keep this pass's per-lane hit lists; 2 = start from the
     *   lists kept by such a pass (ignored unless they were kept for the same
     *   arrays since the last sph_nnps_update).                               */
    int src_eos;
    double eos_par[4];
    int nl_mode;
    /* phase (ghost split, single-array groups): 0 = the whole group; 1 = the part that needs no ghosts, called
     * after sph_nnps_update and BEFORE the ghosts arrive (equations without sources over the particles present,
     * records of the real particles packed and -- option split_pair -- the pair loops of the wavefronts whose
     * candidates cannot include a ghost); 2 = the rest, after sph_nnps_update_ghosts (equations without sources
     * over the new particles, ghost records, the pair loops not run yet).  1 and 2 of one group must follow each other with only phase calls in between.   */
    int phase;
} sph_group;

/* ---------------------------------------------------------------------- */
/* context                                                                  */
/* ---------------------------------------------------------------------- */
/* `stream`: a hipStream_t (e.g. torch.cuda.current_stream().cuda_stream) or
 * NULL for a stream owned by the context.  All kernels of this context are
 * launched on it.                                                          */
int sph_ctx_create(int device, void *stream, sph_ctx **out);
int sph_ctx_destroy(sph_ctx *ctx);
int sph_ctx_synchronize(sph_ctx *ctx);
const char *sph_last_error(void);
const char *sph_version(void);
int sph_prop_id(const char *name); /* -1 if unknown */

/* ---------------------------------------------------------------------- */
/* device mirror of the host ParticleArrays                                 */
/* replaces DeviceHelper.push/pull/resize (pysph/base/device_helper.py:200, */
/* :219, :130) -- the host keeps owning the numpy/cyarray buffers.          */
/* ---------------------------------------------------------------------- */
/* (Re)size array `array_id` to n particles of which the first n_real are
 * real (particle_array.pyx:423).  Existing device data is preserved up to
 * min(old, new).                                                           */
int sph_array_resize(sph_ctx *ctx, int array_id, size_t n, size_t n_real);
int sph_array_size(sph_ctx *ctx, int array_id, size_t *n, size_t *n_real);
/* New particle i takes every device property of old particle indices[i]
 * (host array of n_new indices < n); afterwards n = n_new, of which the first
 * n_real_new are real.  DeviceHelper.align / remove_particles /
 * align_particles (pysph/base/device_helper.py:241-345, :480-560).          */
int sph_array_permute(sph_ctx *ctx, int array_id, const uint32_t *indices, size_t n_new, size_t n_real_new);
/* Make sure property `prop` has device storage (zero-filled when created). */
int sph_array_ensure_prop(sph_ctx *ctx, int array_id, int prop);
/* host -> device / device -> host of n doubles starting at particle `offset`. */
int sph_array_push(sph_ctx *ctx, int array_id, int prop, const double *host, size_t offset, size_t n);
int sph_array_pull(sph_ctx *ctx, int array_id, int prop, double *host, size_t offset, size_t n);
/* Raw device pointer of a property (device-resident pipelines, RCCL halos). */
int sph_array_device_ptr(sph_ctx *ctx, int array_id, int prop, void **dptr);
/* n rows of one property set to `value`, starting at particle `offset` (the ghost
 * rows of a promised-uniform h / m that did not travel: the reference sends every
 * property of a remote particle, parallel_manager.pyx:159-210).               */
int sph_array_fill(sph_ctx *ctx, int array_id, int prop, double value, size_t offset, size_t n);
/* The caller wrote `prop` through a raw pointer, or no longer trusts what the
 * library knows of it: for h / m the next sph_nnps_update looks at the values
 * again, for x / y / z it does not bin on the previous update's bounds.        */
int sph_array_mark_written(sph_ctx *ctx, int array_id, int prop);

/* ---------------------------------------------------------------------- */
/* neighbour search                                                         */
/* replaces DomainManager._compute_cell_size_for_binning                    */
/* (pysph/base/nnps_base.pyx:942-978), NNPS.update/_compute_bounds          */
/* (:1471-1575), LinkedListNNPS._refresh/_bin                               */
/* (pysph/base/linked_list_nnps.pyx:235-383).                               */
/* ---------------------------------------------------------------------- */
/* Bin every particle of the listed arrays on one common grid.  cell_size<=0:
 * radius_scale*max(h) as the reference does.  `bounds` (6 doubles
 * xmin,ymin,zmin,xmax,ymax,zmax) overrides the computed, 1%-padded bounds
 * when non-NULL (multi-GPU runs share the global grid).                    */
int sph_nnps_update(sph_ctx *ctx, int dim, int narrays, const int *array_ids,
                    double radius_scale, double cell_size, const double *bounds);
/* The caller knows the range of h (Integrator.set_fixed_h / LinkedListNNPS(fixed_h=True),
 * pysph/sph/integrator.py:62-81, linked_list_nnps.pyx:54: smoothing lengths that
 * never change): sph_nnps_update then skips the h reduction, and with `bounds`
 * given as well the whole min/max pass and its device->host round trip.
 * hmax < 0 forgets the range.                                                */
int sph_nnps_set_h_range(sph_ctx *ctx, double hmin, double hmax);
/* Ghost split -- overlapping the ghost exchange of a slab-decomposed run with the evaluation
 * (ParallelManager.update, pysph/parallel/parallel_manager.pyx:512-530, is serial in the reference):
 *   sph_nnps_set_extend   the computed bounds are widened by ext[k] on both sides of axis k before the 1 % padding
 *                         (the halo width along the slab axis: the grid of the REAL particles must hold the ghosts
 *                         that arrive later; ghosts outside the grid are clamped into its outermost cells, which
 *                         costs candidates, never neighbours);
 *   sph_nnps_update       then runs while the ghosts are still in flight (the arrays hold their real particles);
 *   sph_nnps_update_ghosts  after the ghosts were appended: bins the particles behind the ones the update saw into
 *                         ghost-only tables on the same grid; pair evaluations read them as a second source segment
 *                         of their array.  `axis`, `lo`, `hi`: every ghost has coordinate < lo or >= hi along `axis`
 *                         (the slab faces; -inf / +inf for an open face) -- with axis 0 wavefronts whose candidate
 *                         windows stay inside [lo, hi) skip the ghost segments, and sph_group.phase can split an
 *                         evaluation into the part that needs no ghosts and the rest.                             */
int sph_nnps_set_extend(sph_ctx *ctx, double ex, double ey, double ez);
/* The slab faces along `axis` outside which this rank's ghosts lie (coordinate < lo or >= hi; -inf / +inf: open face;
 * axis -1: none).  Named BEFORE sph_nnps_update so that the first half of a split evaluation already knows which
 * wavefronts can reach a ghost.                                                                                  */
int sph_nnps_set_ghost_faces(sph_ctx *ctx, int axis, double lo, double hi);
int sph_nnps_update_ghosts(sph_ctx *ctx, int axis, double lo, double hi);
/* d8: cell_size hmin xmin[3] xmax[3];  i4: ncx ncy ncz n_cells             */
int sph_nnps_info(sph_ctx *ctx, double *d8, long *i4);
/* min/max of x,y,z,h over the listed arrays, no padding (out: 8 doubles
 * xmin ymin zmin hmin xmax ymax zmax hmax) -- NNPS._compute_bounds input.  */
int sph_nnps_minmax(sph_ctx *ctx, int narrays, const int *array_ids, double *out8);
/* Neighbour lists as CSR, replacing NNPS.get_nearest_particles
 * (nnps_base.pyx:1290-1323) / GPUNeighborCache (gpu_nnps_base.pyx:54-117).
 * Pass 1 (nbrs==NULL): fills start[0..nd] (exclusive scan of the counts) and
 * *total.  Pass 2: fills nbrs[total]; each list is sorted ascending.  Host
 * buffers; start_len must equal nd + 1 with nd the DEVICE particle count of
 * dst (sph_array_info: ghosts included), nbrs_len >= total -- a mismatch is
 * SPH_ERR_ARG, never an overrun.                                            */
int sph_nnps_get_csr(sph_ctx *ctx, int src, int dst, uint32_t *start, size_t start_len, uint32_t *nbrs,
                     size_t nbrs_len, size_t *total);
/* Permutation of array `array_id` into cell order (sorted -> original
 * index), n entries: get_spatially_ordered_indices
 * (linked_list_nnps.pyx:198-209).                                          */
int sph_nnps_get_order(sph_ctx *ctx, int array_id, uint32_t *perm);

/* Physically reorder EVERY device property of `array_id` into the cell order
 * of the last sph_nnps_update (new[i] = old[perm[i]]): the device-resident
 * form of NNPS.spatially_order_particles (pysph/base/nnps_base.pyx:1615-1629),
 * which the reference's Solver runs before the first step and every
 * reorder_freq (=50 on GPU backends) steps (pysph/solver/solver.py:296-302,
 * application.py:1157-1161).  Invalidates the grid: update afterwards.      */
int sph_nnps_reorder_array(sph_ctx *ctx, int array_id);

/* ---------------------------------------------------------------------- */
/* acceleration evaluation                                                  */
/* replaces one leaf-group block of the generated AccelerationEval.compute  */
/* (pysph/sph/acceleration_eval_cython.mako:10-154), including the          */
/* destination/source regrouping of MegaGroup                               */
/* (pysph/sph/acceleration_eval.py:94-162).                                 */
/* ---------------------------------------------------------------------- */
int sph_eval_group(sph_ctx *ctx, const sph_kernel *kernel, const sph_group *group,
                   double t, double dt);

/* ---------------------------------------------------------------------- */
/* user properties and generated equation families                          */
/* replaces, for equations that are not hand-written in this library, the   */
/* transpile-and-compile step of the reference (pysph/sph/equation.py       */
/* :748-892 get_loop_code..., acceleration_eval_cython_helper.py:147-181):   */
/* the host language translates the Equation bodies into a family struct    */
/* for the pair-loop skeleton (pysph_amd/csrc/sph_pair.h), builds it with    */
/* hipcc into its own shared object and hands the library the module's      */
/* launch function.                                                          */
/* ---------------------------------------------------------------------- */
/* Id of the property called `name`: a built-in one, an already registered
 * user slot, or a new user slot (process-wide table; <0: table full).       */
int sph_prop_register(const char *name);

/* x sub-bins per cell in the device sort key (sph_nnps.hip k_cell_keys): particles of a
 * cell are ordered along x, fine_start has one entry per sub-bin. */
#define SPH_NSUB 8

#define SPH_GEN_MAX_PROPS 48
#define SPH_GEN_MAX_SPROPS 20
#define SPH_GEN_MAX_PAR 64
#define SPH_GEN_MAX_STATE 16

/* What the library hands to a generated module's launch function: plain
 * device pointers and scalars (the contents of PairArgs<Fam> in sph_pair.h). */
typedef struct sph_gen_args {
    void *stream;                 /* hipStream_t */
    int kernel_kind, uniform_h;
    int nsrc;                     /* 0: equations without sources only    */
    const uint32_t *src_cell_start[SPH_MAX_ARRAYS];
    const uint32_t *src_fine_start[SPH_MAX_ARRAYS]; /* per x sub-bin of a cell (sph_nnps.hip)     */
    uint32_t src_off[SPH_MAX_ARRAYS], src_flags[SPH_MAX_ARRAYS];
    const double *rec;            /* packed records [x y z h | sprops...]  */
    int nrec;                     /* doubles per record                    */
    const void *fpos;             /* float4 prefilter positions            */
    double dom_extent;
    uint32_t d_off, nd;
    const uint32_t *d_keys, *d_fkeys, *d_perm; /* cell ids, fine (sub-bin) keys, sorted -> original */
    const uint32_t *d_tile_order; /* traversal order of the 256-particle destination tiles, or NULL */
    uint32_t d_start, d_stop, dflags;
    int nc[3];
    double xmin[3], cell_size, radius_scale;
    double sigma, deltap;
    int dim;
    double t, dt;
    double hu, h1u, facu, epsu, hr2u;
    /* mode 0: no-source kernel (nsrc == 0) or fused pair kernel;
     * mode 1: initialize + no-source loops only, results stored (run before the
     *         records are packed when a later loop reads a property as s_<prop>
     *         that initialize writes -- the reference finishes initialize for
     *         all particles first, acceleration_eval_cython.mako:36-58);
     * mode 2: loop_all equations: one thread per destination walks its
     *         neighbour list (NBRS, N_NBRS; mako :62-80).                      */
    int mode;
    int skip_init;                /* mode 0/2: initialize already done by a mode-1 launch */
    int skip_post;                /* mode 2 followed by a pair launch of the same family: no post_loop yet */
    int rec_f32;                  /* records are floats [x-x0 y-y0 z-z0 h | sprops...] (option record_f32) */
    const uint32_t *csr_start[SPH_MAX_ARRAYS]; /* mode 2: per source, start[nd+1] */
    const uint32_t *csr_nbrs[SPH_MAX_ARRAYS];  /*         and neighbour indices   */
    const double *sraw[SPH_MAX_ARRAYS][SPH_GEN_MAX_SPROPS]; /* mode 2: source props, original order */
    int n_din, n_dout, npar;
    const double *din[SPH_GEN_MAX_PROPS];   /* destination props read      */
    double *dout[SPH_GEN_MAX_PROPS];        /* destination props read-modify-written */
    double par[SPH_GEN_MAX_PAR];
    double *state;                /* device copy of sph_gen_family.state (equation attributes the bodies write) */
    int row_mod3, norm_masks;     /* the context options of the same names (row order / hit-mask normalisation of the pair kernel) */
} sph_gen_args;

typedef int (*sph_gen_launch_fn)(const sph_gen_args *);

/* One destination of one group, all of its equations generated.  Equation k
 * of the family acts for source j iff bit k of src_flags[j] is set.          */
typedef struct sph_gen_family {
    sph_gen_launch_fn launch;
    int dest;
    int nsrc;
    int src[SPH_MAX_ARRAYS];
    uint32_t src_flags[SPH_MAX_ARRAYS];
    int n_sprops, sprops[SPH_GEN_MAX_SPROPS]; /* source props in record order (after x y z h) */
    int n_din, din[SPH_GEN_MAX_PROPS];
    int n_dout, dout[SPH_GEN_MAX_PROPS];
    int npar;
    double par[SPH_GEN_MAX_PAR];
    int real;                    /* Group(real=...)                          */
    long start_idx, stop_idx;    /* Group(start_idx, stop_idx); <0: None     */
    int split_init;              /* 1: run a mode-1 launch before packing    */
    int loop_all;                /* 1: the family has loop_all equations (mode 2 launch)   */
    int also_pair;               /* 1: ... and pair loops as well: mode 2 (no post_loop), then the pair launch */
    int init_pair;               /* initialize_pair equations: 1 = mode-3 launch per source, then the loops;
                                  * 2 = nothing else follows (the last mode-3 launch runs post_loop)      */
    int nstate;                  /* equation attributes the device code assigns (self.x = ...): in/out    */
    double state[SPH_GEN_MAX_STATE];
    sph_gen_launch_fn launch_f32; /* the same family compiled with float arithmetic (fp32 records, fp32 accumulators),
                                  * or NULL: taken for the pair launch under option arith_f32 -- the fp32 mode of the
                                  * reference's generated GPU code (acceleration_eval_gpu_helper.py:281-283,437-441) */
} sph_gen_family;

/* initialize -> no-source loops -> per-source pair loops -> post_loop of one
 * generated family (the loop nest of acceleration_eval_cython.mako:10-154).  */
int sph_eval_generated(sph_ctx *ctx, const sph_kernel *kernel, sph_gen_family *family,
                       double t, double dt);
/* max over the first n_real particles of a property (dt_cfl, dt_force:
 * pysph/sph/integrator.py:161-200).                                        */
int sph_reduce_max(sph_ctx *ctx, int array_id, int prop, double *out);

/* ---------------------------------------------------------------------- */
/* ghost-particle halos (slab decomposition, one process per GPU)           */
/* replaces ParallelManager.compute_remote_particles / remote_exchange_data */
/* (pysph/parallel/parallel_manager.pyx:1159-1243, :159-210); the transport */
/* itself is RCCL send/recv issued by the Python host (torch.distributed).  */
/* ---------------------------------------------------------------------- */
/* Select particles of `array_id` by their coordinate v along `axis` (0,1,2),
 * in ascending index order (deterministic); counts[2] receives the lengths of
 * the "lo" (side 0) and "hi" (side 1) lists.
 *   mode 0 (slab faces):    lo: v <  p0          hi: v >= p1
 *   mode 1 (periodic box):  lo: (v - p0) <= p2   hi: (p1 - v) <= p2
 *     -- the tests of CPUDomainManager._create_ghosts_periodic
 *        (pysph/base/nnps_base.pyx:805-817) with p0/p1 = domain min/max and
 *        p2 = n_layers*cell_size.
 * `upto`: consider the first `upto` particles (0: the real particles only).  */
int sph_halo_select(sph_ctx *ctx, int array_id, int axis, int mode, double p0, double p1, double p2,
                    size_t upto, size_t *counts);
/* Gather `nprops` properties of the particles selected for `side` into the
 * device buffer dst, laid out [nprops][count]; `shift` is added to the
 * `axis` coordinate (periodic wrap, nnps_base.pyx:841-856).                */
int sph_halo_pack(sph_ctx *ctx, int array_id, int side, int nprops, const int *props, int axis,
                  double shift, void *dst_device);
/* As sph_halo_pack for a reflecting boundary: the `axis` coordinate becomes
 * x + 2*(plane - x), the `axis` velocity component (u, v or w) changes sign
 * (CPUDomainManager._create_ghosts_mirror, pysph/base/nnps_base.pyx:506-697). */
int sph_halo_pack_mirror(sph_ctx *ctx, int array_id, int side, int nprops, const int *props, int axis,
                         double plane, void *dst_device);
/* Append `count` ghost particles (tag Remote: sources only) behind the
 * current ones from a device buffer laid out [nprops][count].  n grows,
 * n_real is unchanged.  Drop them again with sph_array_resize(n_real).      */
/* Local images in ONE step (periodic / mirror ghosts of a single-rank domain,
 * CPUDomainManager._create_ghosts_periodic / _mirror, nnps_base.pyx:506-940):
 * the particles selected for `side` are appended behind the array as ghosts,
 * mode 0: `axis` coordinate + val (periodic shift); mode 1: reflected about
 * the plane `val`, normal velocity negated.  Equivalent to sph_halo_pack(_mirror)
 * into a buffer followed by sph_halo_append of that buffer.  *count = images made. */
int sph_halo_image(sph_ctx *ctx, int array_id, int side, int nprops, const int *props, int axis, int mode,
                   double val, size_t *count);
int sph_halo_append(sph_ctx *ctx, int array_id, int nprops, const int *props, const void *src_device,
                    size_t count);
/* Selection AND packing of both slab faces (mode 0 of sph_halo_select: lo:
 * v < lo_cut, hi: v >= hi_cut) without a host round trip -- the ghost refresh of
 * ParallelManager.update (pysph/parallel/parallel_manager.pyx:512-530,
 * remote_exchange_data :159-210) as ONE device pass per array: the rows go
 * straight into two fixed-capacity device messages dst2[side], laid out
 * [nprops][cap2[side]] doubles followed by ONE header double = the row count,
 * NEGATED when it exceeds the capacity (the payload is then incomplete: repeat
 * that face through sph_halo_select / sph_halo_pack with the exact size).
 * shift2[side] is added to the `axis` coordinate (periodic wrap).  dst2[side]
 * may be NULL (open face).  At most 32 properties.  The host learns the counts
 * from the headers (the receiver's copy tells it how many rows to append with
 * sph_halo_append_strided).                                                  */
int sph_halo_select_pack(sph_ctx *ctx, int array_id, int axis, double lo_cut, double hi_cut, size_t upto,
                         int nprops, const int *props, const double *shift2, const size_t *cap2,
                         void *const *dst2);
/* ... when every particle of the array is PROMISED to have ONE smoothing length
 * h_promise and / or ONE mass m_promise (NaN: no promise; established by the
 * caller over all ranks): such a property need not be among `props` -- the
 * receiver writes the promised value into its ghost rows (sph_halo_append_padded,
 * sph_array_fill) and a WCSPH ghost travels as 7 doubles instead of the 9 of
 * ParallelManager.remote_exchange_data (parallel_manager.pyx:159-210) -- and every
 * SELECTED row is checked against the promise here, on the sender's side: a row
 * that breaks it adds 0.5 to the header of its message (|header| = count + 0.5). */
int sph_halo_select_pack_promised(sph_ctx *ctx, int array_id, int axis, double lo_cut, double hi_cut, size_t upto,
                                  int nprops, const int *props, const double *shift2, const size_t *cap2,
                                  void *const *dst2, double h_promise, double m_promise);
/* n <= 64 doubles, one from each DEVICE address, in one round trip (all copies
 * on the context's stream, one synchronisation): the headers of the ghost
 * messages, from which the host learns the counts.                          */
int sph_read_values(sph_ctx *ctx, int n, const void *const *dev_ptrs, double *out);
/* ... and WITHOUT the round trip: one launch gathers the n <= 48 doubles and nflags
 * uint32 flag words (as doubles, behind them; n + nflags <= 64) into host_out,
 * PAGE-LOCKED host memory the device can write; the caller records an event
 * behind it and reads the values when that has passed (the counts of the
 * round-trip-free ghost exchange, read once the evaluation is queued).         */
int sph_queue_values(sph_ctx *ctx, int n, const void *const *dev_ptrs, int nflags, const void *flag_words,
                     double *host_out);
/* sph_halo_append from a message whose rows are `stride` doubles apart
 * (property k of row i at src[k * stride + i], stride >= count).            */
/* The same message appended WITHOUT a device->host round trip (round 5): all `cap` rows go behind the particles, the
 * first |header| of them are the ghosts, the rest PADDING ROWS parked at x = y = z = 1e18 with every other listed property
 * zero -- inert on the whole path (the bounds and keys of sph_nnps_update skip / spread positions beyond 1e17, every
 * distance test fails by 1e36, ghosts are never destinations) and finite everywhere -- so the host needs no count to
 * size anything.  Real coordinates must stay below 1e17 in magnitude.
 * h_promise / m_promise (NaN: none): every ghost is promised to carry this smoothing length / mass (the ONE value of the
 * array on every rank, established collectively by the caller); then the neighbour update keeps what it knows of h and m.
 * A promised property that is NOT among `props` did not travel: the promised value is written into all `cap` rows.
 * flag_word: a DEVICE uint32 the kernel ORs into -- bit 0: the message was incomplete (negative header), bit 1: a ghost
 * broke a promise; the caller reads it when convenient (pysph_amd/parallel.py: with the counts, once the evaluation that
 * uses the ghosts is queued -- verify_halos -- and repeats an incomplete face the counted way).
 * Replaces the recv side of ParallelManager.remote_exchange_data (parallel_manager.pyx:159-210), whose counts travel over
 * MPI to the host first. */
int sph_halo_append_padded(sph_ctx *ctx, int array_id, int nprops, const int *props, const void *src_device, size_t cap,
                           double h_promise, double m_promise, void *flag_word);
/* ... the messages of both faces of the array in ONE launch: the rows of src_device, then those of src2_device (NULL / 0: none). */
int sph_halo_append_padded2(sph_ctx *ctx, int array_id, int nprops, const int *props, const void *src_device, size_t cap,
                            const void *src2_device, size_t cap2, double h_promise, double m_promise, void *flag_word);
int sph_halo_append_strided(sph_ctx *ctx, int array_id, int nprops, const int *props, const void *src_device,
                            size_t count, size_t stride);

/* sph_halo_append for MIGRANTS under a promise (the particles a neighbouring slab hands over,
 * pysph/parallel/parallel_manager.pyx:1085-1157, in a run whose ranks agreed that this array's h
 * and / or m is ONE value everywhere; NaN: no promise for that property): what the neighbour update
 * knows of h and m survives the append -- the update after a migration makes no device->host round
 * trip, like the one after sph_halo_append_padded -- and every arriving row is checked on the
 * device: one that carries another h or m sets bit 1 of *flag_word_device (the word
 * sph_halo_append_padded uses; sticky, read with the headers of the next exchange).  h and m must
 * be among `props` to be kept.                                                                  */
int sph_halo_append_promised(sph_ctx *ctx, int array_id, int nprops, const int *props, const void *src_device,
                             size_t count, double h_promise, double m_promise, void *flag_word_device);

/* Remove the particles of the last sph_halo_select (both sides) from the
 * array -- particles that migrated to a neighbouring slab, after their
 * properties were packed with sph_halo_pack (ParallelManager's exported
 * particles, pysph/parallel/parallel_manager.pyx:1085-1157).  When fewer
 * than a quarter of the particles leave, the kept rows at the end of the
 * array move into the holes (as the reference's removal does with the ends
 * of its property arrays; option "fill_holes" 0: never); otherwise a stable
 * compaction of every device property.  Requires n == n_real (ghosts dropped).
 * Afterwards n = n_real = *n_left.  Received particles are appended with
 * sph_halo_append and made real with sph_array_resize(n, n).                 */
int sph_halo_remove_selected(sph_ctx *ctx, int array_id, size_t *n_left);

/* Box-wrap the first n_real particles along `axis` into [vmin, vmax]:
 * v < vmin -> v + translate; v > vmax -> v - translate
 * (CPUDomainManager._box_wrap_periodic, pysph/base/nnps_base.pyx:699-748).   */
int sph_domain_box_wrap(sph_ctx *ctx, int array_id, int axis, double vmin, double vmax, double translate);
/* The periodic images of one axis (CPUDomainManager._create_ghosts_periodic,
 * pysph/base/nnps_base.pyx:751-940; selection rule :805-817, shift :841-856)
 * WITHOUT a device->host round trip: every particle present (n of them) within
 * `width` of the low face `lo` gets an image shifted by +shift, within `width`
 * of the high face `hi` one shifted by -shift, in ascending particle index, into
 * FIXED capacities behind the particles: rows [n, n + cap[0]) and [n + cap[0],
 * n + cap[0] + cap[1]).  The array grows by cap[0] + cap[1] (n_real unchanged);
 * rows behind the counts are parked like the padding rows of
 * sph_halo_append_padded (inert on the whole path).  `props`: the properties
 * the images carry (x, y and z among them).  The two counts stay on the device
 * (negated when they exceed their capacity: images are then MISSING);
 * sph_domain_counts_queue sends the counts of all calls so far towards pinned
 * memory, sph_domain_counts_collect returns them -- an update later, when the
 * copy has long completed -- as out[array_id * 6 + axis * 2 + side]
 * (SPH_MAX_ARRAYS * 6 doubles); the caller sizes the next capacities from them
 * and treats a negative one as an error of the step that used the images.   */
int sph_domain_images_padded(sph_ctx *ctx, int array_id, int axis, double lo, double hi, double width, double shift,
                             const size_t cap[2], int nprops, const int *props);
int sph_domain_counts_queue(sph_ctx *ctx);
int sph_domain_counts_collect(sph_ctx *ctx, double *out);
/* 1 and the range of h of the array when it is known WITHOUT looking (found by
 * the last neighbour update or reduction, nothing wrote h since), else 0.    */
int sph_array_h_known(sph_ctx *ctx, int array_id, double *hmin, double *hmax);
/* Ids of the properties that currently have device storage: out[cap], *n
 * receives the count (at most SPH_PROP_COUNT); SPH_ERR_ARG if cap is too
 * small.                                                                    */
int sph_array_props(sph_ctx *ctx, int array_id, int *out, int cap, int *n);
/* Histogram (nbins uint32 counters, HOST memory) of coordinate `axis` of the array's
 * real particles over [vmin, vmin + span): what a re-balancing of the slab
 * decomposition needs of the particles (the reference hands Zoltan every object,
 * pysph/parallel/parallel_manager.pyx:577-640).  One device->host copy of nbins
 * words; re-balancing is rare.                                                */
int sph_coord_histogram(sph_ctx *ctx, int array_id, int axis, double vmin, double span, int nbins, uint32_t *host_out);

/* ---------------------------------------------------------------------- */
/* integrator stage sweeps (the caller either side of the hot path)         */
/* replaces the generated per-particle stage loops                          */
/* (pysph/sph/integrator_cython.mako:87-113) for the steppers of            */
/* pysph/sph/integrator_step.py; stage 0 = initialize, 1 = stage1, 2 = stage2;*/
/* real particles only.                                                     */
/* ---------------------------------------------------------------------- */
enum sph_stepper {
    SPH_STEP_WCSPH = 1, /* WCSPHStep            integrator_step.py:38-93  */
    SPH_STEP_TVF = 2,   /* TransportVelocityStep integrator_step.py:257-299 */
    SPH_STEP_SOLID_MECH = 3 /* SolidMechStep       integrator_step.py:173-255 */
};
int sph_integrate_stage(sph_ctx *ctx, int array_id, int stepper, int stage, double dt);
/* min over the first n_real particles of a property (h_minimum,
 * pysph/sph/integrator.py:150-160).                                        */
int sph_reduce_min(sph_ctx *ctx, int array_id, int prop, double *out);

/* options (key, value):
 *   "pair_variant"   6 = one wavefront per 64 destinations, two-phase pair
 *                    kernel k_pair_wave (default); 0 = per-lane cell walk
 *                    k_pair_direct (cross-checks)
 *   "uniform_h"      0/1: allow the hmin == hmax specialisation (default 1)
 *   "arith_f32"      0/1: pair loops in fp32 arithmetic on fp32 records (the
 *                    reference's GPU backends without --use-double,
 *                    acceleration_eval_gpu_helper.py:281-283); default 0.
 *                    Generated families: the pair launch of those that carry
 *                    a float build (sph_gen_family.launch_f32)
 *   "record_f32"     0/1: fp32 records, fp64 arithmetic; default 0
 *   "const_flags"    0/1: equation-flag set compiled as a constant when it is
 *                    the family's usual one (default 1)
 *   "tile_block_rows" rows of 256-destination tiles per traversal block (XCD-aware
 *                    tile order); "invalidate_nnps": the next evaluation needs a
 *                    fresh sph_nnps_update
 *   "pack_group"     1 ... 0 around the sph_eval_group calls (one per destination)
 *                    of ONE group: WCSPH source records are packed once per
 *                    group instead of once per destination that reads them.
 *                    Nothing else may modify the arrays in between.
 *   "eos_fuse"       0: ignore sph_group.src_eos (default 1)
 *   "mass_fuse"      0: EOS-fused records always carry the mass (default 1: when every array read by a
 *                    launch had ONE mass at the last sph_nnps_update -- the reduction that finds the
 *                    bounds also looks at m once an evaluation could have used it, i.e. from the second
 *                    step on -- the slot carries p / rho^2 and the mass is a constant of
 *                    the source array; a later sph_array_push of m, or an update that skips the
 *                    reduction (bounds + sph_nnps_set_h_range), falls back to the mass-carrying records;
 *                    masses written through sph_array_device_ptr, or by a generated equation, AFTER the
 *                    last update are seen by the next one -- as a changed h is by the uniform-h path)
 *   "merge_arrays"   0: never run a multi-array WCSPH group as one launch over the merged cell order of all arrays
 *                    (default 1: taken when the group's (destination, source) equation matrix is the two-class table
 *                    of WCSPHScheme(fluids, solids), under the conditions of the uniform-mass EOS-fused records;
 *                    sph_nnps_update then sorts all arrays' keys once -- that sorted sequence is the merged order --
 *                    and derives per-array tables only on demand; "lazy_tables" 0 builds them at every update)
 *   "async_update"   0: every sph_nnps_update reduces the bounds and waits for them (default 1: an update that knows h
 *                    and m without looking -- nothing wrote them since the last reduction -- bins on the grid of the
 *                    PREVIOUS update's bounds while reducing its own, with no device->host round trip; sph_nnps_info
 *                    returns the reference's exact values for the current particles either way)
 *   "via_unordered"  0: the key passes of the sort always run in memory order (default 1: an array found to lie in
 *                    memory in no spatial order is visited in the previous update's cell order)
 *   "sort_lbits"     9..11: fixed low key bits of the particle sort's buckets (0: the fewest that keep the bucket count
 *                    under 6144, fewer if the mean bucket would not fit its stage; tests)
 *   "split_pair"     split evaluations (sph_group.phase): 0 (default) = phase 1 prepares (equations without sources,
 *                    records of the particles present), phase 2 launches every wave tile with the ghost segments
 *                    guarded per wavefront; 1 = the interior wave tiles already in phase 1, the face tiles in phase 2
 *   "tension_flag"   0: the elastic rates always gather the artificial stress r_ij (default 1: only while the word the
 *                    MonaghanArtificialStress kernel sets says a particle of the source array is in tension)
 *   "nl_reuse"       1: honour sph_group.nl_mode (default 0: measured slower)
 *   "norm_masks"     0: hit masks are not shifted down to a lane's first hit (default 1)
 *   "dest_list"      pair launches over the REAL particles of an array take their wave tiles from a list of the
 *                    real particles' sorted positions: 1 (default) = when ghosts / images / padding rows are at
 *                    least 1/8 of the rows, 2 = always, 0 = never (every row of the cell order gets a lane)
 *   "fill_holes"     0: sph_halo_remove_selected always compacts stably (default 1: see there)
 *   "row_lds"        1: the elastic rates on uniform-h records evaluate a row tile's hits from an LDS copy of its
 *                    records, tile after tile (default 0: measured 45 % slower); same pairs,
 *                    another summation order
 *   "row_mod3"       order in which a wavefront visits its 3x3 rows of cells: 3 (default) = the row whose
 *                    (y mod 3, z mod 3) equals the step, so that all wavefronts in flight walk rows of one
 *                    residue class at a time and find each other's lines in L1 / L2; 1 / 2 = y / z only;
 *                    0 = (dy, dz) order; 4 = (dy, dz) order rotated per tile.  Sums are taken in that order.
 *   profiling only:  "lds_pad", "wcsph_nr", and -- in a library built with
 *                    `make PROFILING=1` only, an error otherwise -- "ablate",
 *                    "count_iters" / "dump_counters"    */
int sph_set_option(sph_ctx *ctx, const char *key, long value);

/* ---------------------------------------------------------------------- */
/* timing (replaces compyle.profile's ProfileContext keys,                  */
/* acceleration_eval_cython.mako:14-144): accumulated hipEvent time per     */
/* kernel class since the last reset.                                       */
/* ---------------------------------------------------------------------- */
/* on = 1: every kernel class is timed (an event pair around each region); 2: the pair
 * launches only (T_PAIR) -- the event markers themselves cost the stream a few
 * microseconds each; 0: off.                                                   */
int sph_timer_enable(sph_ctx *ctx, int on);
int sph_timer_reset(sph_ctx *ctx);
/* keys: "nnps", "pack", "eos", "pair", "stage"; the pair launches once more per
 * equation family: "pair_wcsph", "pair_density", "pair_tvf", "pair_vgrad",
 * "pair_elastic"; out: total ms and launches.  Launch counters (ms = 0, counted
 * whether or not timing is enabled): "n_eos_fused" (pair launches on the 64-byte
 * EOS-fused records), "n_mass_fused" (launches on records that rely on one mass per array), "n_merged" (group
 * evaluations run as one launch over the merged order), "n_tension_flag" (elastic rate launches that read the tension word), "n_phase2" (pair launches of the second half of a
 * split evaluation: sph_group.phase 2),
 * "n_nl_keep" / "n_nl_reuse" (launches that kept / started from kept neighbour
 * lists), "n_async" (neighbour updates that made no device->host round trip). */
int sph_timer_get(sph_ctx *ctx, const char *key, double *ms, long *count);

/* ABI self-description: the size of a struct of this header ("sph_kernel", "sph_equation", "sph_group",
 * "sph_gen_family", "sph_gen_args") and the offset of one of its fields AS THIS LIBRARY WAS COMPILED, so that a
 * binding written in another language (the ctypes Structures of pysph_amd/device.py) can
 * check its own layout against the library it loaded instead of trusting a copy of this file.  -1: unknown name.
 * No reference counterpart (Cython reads the C headers at build time).  */
long sph_abi_sizeof(const char *struct_name);
long sph_abi_offsetof(const char *struct_name, const char *field);

#ifdef __cplusplus
}
#endif
#endif
