/*
 * sph_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * See sph_oracle.h for scope.  Every function cites the reference lines it
 * restates.  Arithmetic is written in the reference's operation order and
 * must be compiled WITHOUT fp contraction / fast-math (see Makefile) so that
 * it reproduces what the reference's Python classes compute (and hence what
 * compyle's Cython translation of them computes), bit for bit.
 */
#include "sph_oracle.h"

#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static char g_err[512];
const char *orc_last_error(void) { return g_err; }

/* ------------------------------------------------------------------ */
/* SPH kernels: pysph/base/kernels.py                                   */
/* ------------------------------------------------------------------ */

/* "fac = self.fac * h1 [* h1 [* h1]]" -- kernels.py:73-79 (and every kernel) */
static inline double k_fac(const orc_kernel *k, double h1)
{
    double fac = k->fac * h1;
    if (k->dim == 2) fac = k->fac * h1 * h1;
    else if (k->dim == 3) fac = k->fac * h1 * h1 * h1;
    return fac;
}

double orc_kernel_w(const orc_kernel *k, double rij, double h)
{
    double h1 = 1.0 / h;
    double q = rij * h1;
    double fac = k_fac(k, h1);
    double val = 0.0;
    switch (k->kind) {
    case OK_CUBIC_SPLINE: { /* kernels.py:68-90 */
        double tmp2 = 2. - q;
        if (q > 2.0) val = 0.0;
        else if (q > 1.0) val = 0.25 * tmp2 * tmp2 * tmp2;
        else val = 1 - 1.5 * q * q * (1 - 0.5 * q);
        return val * fac;
    }
    case OK_WENDLAND_QUINTIC: { /* kernels.py:306-324 */
        double tmp = 1. - 0.5 * q;
        if (q < 2.0) val = tmp * tmp * tmp * tmp * (2.0 * q + 1.0);
        return val * fac;
    }
    case OK_QUINTIC_SPLINE: { /* kernels.py:1094-1123 */
        double tmp3 = 3. - q, tmp2 = 2. - q, tmp1 = 1. - q;
        if (q > 3.0) val = 0.0;
        else if (q > 2.0) val = tmp3 * tmp3 * tmp3 * tmp3 * tmp3;
        else if (q > 1.0) {
            val = tmp3 * tmp3 * tmp3 * tmp3 * tmp3;
            val -= 6.0 * tmp2 * tmp2 * tmp2 * tmp2 * tmp2;
        } else {
            val = tmp3 * tmp3 * tmp3 * tmp3 * tmp3;
            val -= 6.0 * tmp2 * tmp2 * tmp2 * tmp2 * tmp2;
            val += 15. * tmp1 * tmp1 * tmp1 * tmp1 * tmp1;
        }
        return val * fac;
    }
    case OK_GAUSSIAN: /* kernels.py:868-884 */
        if (q < 3.0) val = exp(-q * q) * fac;
        return val;
    }
    return 0.0;
}

double orc_kernel_dwdq(const orc_kernel *k, double rij, double h)
{
    double h1 = 1.0 / h;
    double q = rij * h1;
    double fac = k_fac(k, h1);
    double val = 0.0;
    switch (k->kind) {
    case OK_CUBIC_SPLINE: { /* kernels.py:92-124 */
        double tmp2 = 2. - q;
        if (rij > 1e-12) {
            if (q > 2.0) val = 0.0;
            else if (q > 1.0) val = -0.75 * tmp2 * tmp2;
            else val = -3.0 * q * (1 - 0.75 * q);
        } else val = 0.0;
        return val * fac;
    }
    case OK_WENDLAND_QUINTIC: { /* kernels.py:326-346 */
        double tmp = 1.0 - 0.5 * q;
        if (q < 2.0)
            if (rij > 1e-12) val = -5.0 * q * tmp * tmp * tmp;
        return val * fac;
    }
    case OK_QUINTIC_SPLINE: { /* kernels.py:1125-1158 */
        double tmp3 = 3. - q, tmp2 = 2. - q, tmp1 = 1. - q;
        if (rij > 1e-12) {
            if (q > 3.0) val = 0.0;
            else if (q > 2.0) val = -5.0 * tmp3 * tmp3 * tmp3 * tmp3;
            else if (q > 1.0) {
                val = -5.0 * tmp3 * tmp3 * tmp3 * tmp3;
                val += 30.0 * tmp2 * tmp2 * tmp2 * tmp2;
            } else {
                val = -5.0 * tmp3 * tmp3 * tmp3 * tmp3;
                val += 30.0 * tmp2 * tmp2 * tmp2 * tmp2;
                val -= 75.0 * tmp1 * tmp1 * tmp1 * tmp1;
            }
        } else val = 0.0;
        return val * fac;
    }
    case OK_GAUSSIAN: /* kernels.py:886-905 */
        if (q < 3.0)
            if (rij > 1e-12) val = -2.0 * q * exp(-q * q);
        return val * fac;
    }
    return 0.0;
}

/* kernels.py:126-137 (identical body in every kernel class) */
void orc_kernel_gradient(const orc_kernel *k, const double *xij, double rij,
                         double h, double *grad)
{
    double h1 = 1. / h;
    double tmp;
    if (rij > 1e-12) {
        double wdash = orc_kernel_dwdq(k, rij, h);
        tmp = wdash * h1 / rij;
    } else tmp = 0.0;
    grad[0] = tmp * xij[0];
    grad[1] = tmp * xij[1];
    grad[2] = tmp * xij[2];
}

/* ------------------------------------------------------------------ */
/* NNPS: linked_list_nnps.pyx / nnps_base.pyx / nnps_base.pxd           */
/* ------------------------------------------------------------------ */
struct orc_nnps {
    int dim, narrays;
    double radius_scale;
    double cell_size, hmin;
    double xmin[3], xmax[3];
    int nc[3];
    long n_cells;
    const orc_array *arr[ORC_MAX_ARRAYS];
    unsigned *head[ORC_MAX_ARRAYS];
    unsigned *next[ORC_MAX_ARRAYS];
    long head_cap[ORC_MAX_ARRAYS], next_cap[ORC_MAX_ARRAYS];
};

orc_nnps *orc_nnps_create(int dim, int narrays, double radius_scale)
{
    if (narrays > ORC_MAX_ARRAYS) return NULL;
    orc_nnps *n = (orc_nnps *)calloc(1, sizeof(orc_nnps));
    n->dim = dim;
    n->narrays = narrays;
    n->radius_scale = radius_scale;
    return n;
}

void orc_nnps_destroy(orc_nnps *n)
{
    if (!n) return;
    for (int i = 0; i < ORC_MAX_ARRAYS; i++) { free(n->head[i]); free(n->next[i]); }
    free(n);
}

void orc_nnps_set_array(orc_nnps *n, int index, const orc_array *arr)
{
    n->arr[index] = arr;
}

/* nnps_base.pxd:39-57  real_to_int: <int>floor(real_val/step) */
static inline int real_to_int(double real_val, double step)
{
    return (int)floor(real_val / step);
}

/* nnps_base.pxd:113-135 get_valid_cell_index (+ flatten_raw :83-96) */
static inline long valid_cell_index(int cx, int cy, int cz, const int *nc, long n_cells)
{
    long ncx = nc[0], ncy = nc[1], ncz = nc[2];
    long cell_index = -1;
    int ok = (ncx > cx && cx > -1) && (ncy > cy && cy > -1) && (ncz > cz && cz > -1);
    if (ok) {
        cell_index = (long)(cx + ncx * cy + ncx * ncy * cz);
        if (!(-1 < cell_index && cell_index < n_cells)) cell_index = -1;
    }
    return cell_index;
}

/* exported for the tests that mirror test_nnps.py:1394-1467 */
long orc_flatten(int cx, int cy, int cz, const int *nc)
{
    /* nnps_base.pxd:83-96 flatten_raw: row-major in x, then y, then z; `dim` unused */
    return (long)(cx + (long)nc[0] * cy + (long)nc[0] * (long)nc[1] * cz);
}

void orc_unflatten(long cell_index, const int *nc, int dim, int *out3)
{
    /* nnps_base.pyx:70-95: integer division (cdivision), dim-aware */
    int ncx = nc[0], ncy = nc[1];
    int ix = 0, iy = 0, iz = 0;
    if (dim > 1) {
        if (dim > 2) {
            int plane = ncx * ncy;
            iz = (int)(cell_index / plane);
            cell_index -= (long)iz * plane;
        }
        iy = (int)(cell_index / ncx);
        ix = (int)(cell_index - (long)iy * ncx);
    } else {
        ix = (int)cell_index;
    }
    out3[0] = ix; out3[1] = iy; out3[2] = iz;
}

long orc_valid_cell_index(int cx, int cy, int cz, const int *nc, long n_cells)
{
    return valid_cell_index(cx, cy, cz, nc, n_cells);
}

int orc_nnps_update(orc_nnps *n)
{
    /* DomainManager._compute_cell_size_for_binning  nnps_base.pyx:942-978 */
    double hmax = -1.0, hmin = DBL_MAX;
    for (int a = 0; a < n->narrays; a++) {
        const orc_array *A = n->arr[a];
        const double *h = A->p[OP_H];
        for (long i = 0; i < A->n; i++) {
            if (h[i] > hmax) hmax = h[i];
            if (h[i] < hmin) hmin = h[i];
        }
    }
    double cell_size = n->radius_scale * hmax;
    n->hmin = n->radius_scale * hmin;
    if (cell_size < 1e-6) cell_size = 1.0;
    n->cell_size = cell_size;

    /* NNPS._compute_bounds  nnps_base.pyx:1520-1575 */
    double xmax = -1e100, ymax = -1e100, zmax = -1e100;
    double xmin = 1e100, ymin = 1e100, zmin = 1e100;
    for (int a = 0; a < n->narrays; a++) {
        const orc_array *A = n->arr[a];
        const double *x = A->p[OP_X], *y = A->p[OP_Y], *z = A->p[OP_Z];
        for (long i = 0; i < A->n; i++) {
            xmax = fmax(x[i], xmax); ymax = fmax(y[i], ymax); zmax = fmax(z[i], zmax);
            xmin = fmin(x[i], xmin); ymin = fmin(y[i], ymin); zmin = fmin(z[i], zmin);
        }
    }
    double lx = xmax - xmin, ly = ymax - ymin, lz = zmax - zmin;
    xmin -= lx * 0.01; ymin -= ly * 0.01; zmin -= lz * 0.01;
    xmax += lx * 0.01; ymax += ly * 0.01; zmax += lz * 0.01;
    double eps = 1e-12;
    if (fabs(xmax - xmin) < eps && fabs(ymax - ymin) < eps && fabs(zmax - zmin) < eps) {
        xmin -= 0.5; xmax += 0.5;
        ymin -= 0.5; ymax += 0.5;
        zmin -= 0.5; zmax += 0.5;
    }
    n->xmin[0] = xmin; n->xmin[1] = ymin; n->xmin[2] = zmin;
    n->xmax[0] = xmax; n->xmax[1] = ymax; n->xmax[2] = zmax;

    /* LinkedListNNPS._get_number_of_cells  linked_list_nnps.pyx:293-326 */
    double cell_size1 = 1. / cell_size;
    int ncx = (int)ceil(cell_size1 * (xmax - xmin));
    int ncy = (int)ceil(cell_size1 * (ymax - ymin));
    int ncz = (int)ceil(cell_size1 * (zmax - zmin));
    if (ncx < 0 || ncy < 0 || ncz < 0) {
        snprintf(g_err, sizeof g_err,
                 "LinkedListNNPS: Number of cells is negative (%d, %d, %d).", ncx, ncy, ncz);
        return -1;
    }
    ncx = ncx == 0 ? 1 : ncx;
    ncy = ncy == 0 ? 1 : ncy;
    ncz = ncz == 0 ? 1 : ncz;
    n->nc[0] = ncx; n->nc[1] = ncy; n->nc[2] = ncz;
    long ncells = ncx;
    if (n->dim == 2) ncells = (long)ncx * ncy;
    if (n->dim == 3) ncells = (long)ncx * ncy * ncz;
    /* _count_occupied_cells  linked_list_nnps.pyx:335-343 */
    if (ncells < 0 || ncells > (1L << 28)) {
        snprintf(g_err, sizeof g_err,
                 "ERROR: LinkedListNNPS requires too many cells (%ld).", ncells);
        return -2;
    }
    n->n_cells = ncells;

    /* _refresh :345-383 and _bin :235-286 (serial, ascending index) */
    for (int a = 0; a < n->narrays; a++) {
        const orc_array *A = n->arr[a];
        if (n->head_cap[a] < ncells) {
            free(n->head[a]);
            n->head[a] = (unsigned *)malloc(sizeof(unsigned) * (size_t)ncells);
            n->head_cap[a] = ncells;
        }
        if (n->next_cap[a] < A->n) {
            free(n->next[a]);
            n->next[a] = (unsigned *)malloc(sizeof(unsigned) * (size_t)(A->n > 0 ? A->n : 1));
            n->next_cap[a] = A->n;
        }
        unsigned *head = n->head[a], *next = n->next[a];
        for (long j = 0; j < ncells; j++) head[j] = UINT_MAX;
        for (long j = 0; j < A->n; j++) next[j] = UINT_MAX;
        const double *x = A->p[OP_X], *y = A->p[OP_Y], *z = A->p[OP_Z];
        for (long i = 0; i < A->n; i++) {
            int cx = real_to_int(x[i] - xmin, cell_size);
            int cy = real_to_int(y[i] - ymin, cell_size);
            int cz = real_to_int(z[i] - zmin, cell_size);
            long cid = (long)(cx + (long)ncx * cy + (long)ncx * ncy * cz); /* flatten_raw */
            next[i] = head[cid];
            head[cid] = (unsigned)i;
        }
    }
    return 0;
}

void orc_nnps_info(const orc_nnps *n, double *d8, long *i4)
{
    d8[0] = n->cell_size; d8[1] = n->hmin;
    for (int k = 0; k < 3; k++) { d8[2 + k] = n->xmin[k]; d8[5 + k] = n->xmax[k]; }
    i4[0] = n->nc[0]; i4[1] = n->nc[1]; i4[2] = n->nc[2]; i4[3] = n->n_cells;
}

/* LinkedListNNPS.find_nearest_neighbors  linked_list_nnps.pyx:92-196.
 * Appends at most `cap` ids to out, returns the true count.            */
static long find_nearest_neighbors(const orc_nnps *n, int src, int dst, long d_idx,
                                   unsigned *out, long cap)
{
    static const int shifts[3] = {-1, 0, 1};
    const orc_array *S = n->arr[src], *D = n->arr[dst];
    const double *s_x = S->p[OP_X], *s_y = S->p[OP_Y], *s_z = S->p[OP_Z], *s_h = S->p[OP_H];
    const unsigned *head = n->head[src], *next = n->next[src];
    double radius_scale = n->radius_scale, cell_size = n->cell_size;
    double x = D->p[OP_X][d_idx], y = D->p[OP_Y][d_idx], z = D->p[OP_Z][d_idx];
    int _cx = real_to_int(x - n->xmin[0], cell_size);
    int _cy = real_to_int(y - n->xmin[1], cell_size);
    int _cz = real_to_int(z - n->xmin[2], cell_size);
    double hi2 = radius_scale * D->p[OP_H][d_idx];
    hi2 *= hi2;
    long count = 0;
    for (int ix = 0; ix < 3; ix++)
        for (int iy = 0; iy < 3; iy++)
            for (int iz = 0; iz < 3; iz++) {
                long ci = valid_cell_index(_cx + shifts[ix], _cy + shifts[iy], _cz + shifts[iz],
                                           n->nc, n->n_cells);
                if (ci > -1) {
                    unsigned _next = head[ci];
                    while (_next != UINT_MAX) {
                        double hj2 = radius_scale * s_h[_next];
                        hj2 *= hj2;
                        double dx = s_x[_next] - x, dy = s_y[_next] - y, dz = s_z[_next] - z;
                        double xij2 = dx * dx + dy * dy + dz * dz; /* norm2 */
                        if ((xij2 < hi2) || (xij2 < hj2)) {
                            if (count < cap) out[count] = _next;
                            count++;
                        }
                        _next = next[_next];
                    }
                }
            }
    return count;
}

long orc_nnps_neighbors(const orc_nnps *n, int src, int dst, long d_idx, unsigned *out, long cap)
{
    return find_nearest_neighbors(n, src, dst, d_idx, out, cap);
}

/* NNPSBase.brute_force_neighbors  nnps_base.pyx:1325-1366 */
long orc_nnps_brute_force(const orc_nnps *n, int src, int dst, long d_idx, unsigned *out, long cap)
{
    const orc_array *S = n->arr[src], *D = n->arr[dst];
    double xi = D->p[OP_X][d_idx], yi = D->p[OP_Y][d_idx], zi = D->p[OP_Z][d_idx];
    double hi = D->p[OP_H][d_idx] * n->radius_scale;
    double hi2 = hi * hi;
    long count = 0;
    for (long j = 0; j < S->n; j++) {
        double hj = n->radius_scale * S->p[OP_H][j];
        double dx = xi - S->p[OP_X][j], dy = yi - S->p[OP_Y][j], dz = zi - S->p[OP_Z][j];
        double xij2 = dx * dx + dy * dy + dz * dz;
        if ((xij2 < hi2) || (xij2 < hj * hj)) {
            if (count < cap) out[count] = (unsigned)j;
            count++;
        }
    }
    return count;
}

long orc_nnps_csr(const orc_nnps *n, int src, int dst, unsigned *start, unsigned *nbrs, int nthreads)
{
    const orc_array *D = n->arr[dst];
    long nd = D->n;
    (void)nthreads;
    if (!nbrs) {
        /* counts pass -> exclusive scan into start[0..nd] */
#pragma omp parallel for schedule(dynamic, 64) num_threads(nthreads > 0 ? nthreads : 1)
        for (long i = 0; i < nd; i++)
            start[i + 1] = (unsigned)find_nearest_neighbors(n, src, dst, i, NULL, 0);
        start[0] = 0;
        for (long i = 0; i < nd; i++) start[i + 1] += start[i];
        return start[nd];
    }
#pragma omp parallel for schedule(dynamic, 64) num_threads(nthreads > 0 ? nthreads : 1)
    for (long i = 0; i < nd; i++)
        find_nearest_neighbors(n, src, dst, i, nbrs + start[i], start[i + 1] - start[i]);
    return start[nd];
}

/* ------------------------------------------------------------------ */
/* pair symbols: equation.py:188-297                                    */
/* ------------------------------------------------------------------ */
enum {
    PS_HIJ = 1 << 0, PS_RHOIJ = 1 << 1, PS_RHOIJ1 = 1 << 2, PS_EPS = 1 << 3,
    PS_XIJ = 1 << 4, PS_VIJ = 1 << 5, PS_R2IJ = 1 << 6, PS_RIJ = 1 << 7,
    PS_WIJ = 1 << 8, PS_DWIJ = 1 << 9, PS_WDP = 1 << 10
};

/* symbols named in each equation's loop() signature */
static unsigned eq_symbols(int kind)
{
    switch (kind) {
    case OE_CONTINUITY: return PS_DWIJ | PS_VIJ;                       /* basic_equations.py:187 */
    case OE_MOMENTUM:                                                   /* wc/basic.py:205-208 */
        return PS_VIJ | PS_XIJ | PS_HIJ | PS_R2IJ | PS_RHOIJ1 | PS_EPS | PS_DWIJ | PS_WIJ | PS_WDP;
    case OE_XSPH: return PS_WIJ | PS_RHOIJ1 | PS_VIJ;                   /* basic_equations.py:290 */
    case OE_SUMMATION_DENSITY: return PS_WIJ;                           /* basic_equations.py:28 */
    case OE_TVF_SUMMATION_DENSITY: return PS_WIJ;                       /* transport_velocity.py:56 */
    case OE_TVF_MOM_PRESSURE: return PS_DWIJ;                           /* :290 */
    case OE_TVF_MOM_VISCOSITY: return PS_R2IJ | PS_EPS | PS_DWIJ | PS_VIJ | PS_XIJ; /* :363 */
    case OE_TVF_MOM_ART_VISCOSITY:                                      /* :420 */
        return PS_RHOIJ1 | PS_R2IJ | PS_EPS | PS_DWIJ | PS_VIJ | PS_XIJ | PS_HIJ;
    case OE_TVF_MOM_ART_STRESS: return PS_DWIJ;                         /* :473 */
    case OE_MONAGHAN_ART_VISCOSITY:                                     /* basic_equations.py:236 */
        return PS_VIJ | PS_XIJ | PS_HIJ | PS_R2IJ | PS_RHOIJ1 | PS_EPS | PS_DWIJ;
    case OE_VELOCITY_GRADIENT_3D:
    case OE_VELOCITY_GRADIENT_2D: return PS_DWIJ | PS_VIJ;
    case OE_MOMENTUM_WITH_STRESS: return PS_WIJ | PS_DWIJ;
    default: return 0;
    }
}

/* Group._setup_precomputed: transitive closure  equation.py:589-625 */
static unsigned close_symbols(unsigned s)
{
    for (;;) {
        unsigned t = s;
        if (t & PS_EPS) t |= PS_HIJ;
        if (t & PS_RHOIJ1) t |= PS_RHOIJ;
        if (t & PS_R2IJ) t |= PS_XIJ;
        if (t & PS_RIJ) t |= PS_R2IJ;
        if (t & PS_WIJ) t |= PS_XIJ | PS_RIJ | PS_HIJ;
        if (t & PS_DWIJ) t |= PS_XIJ | PS_RIJ | PS_HIJ;
        if (t & PS_WDP) t |= PS_XIJ | PS_HIJ;
        if (t == s) return s;
        s = t;
    }
}

typedef struct {
    double HIJ, RHOIJ, RHOIJ1, EPS, R2IJ, RIJ, WIJ, WDP;
    double XIJ[3], VIJ[3], DWIJ[3];
} pair_t;


/* ------------------------------------------------------------------ */
/* 3x3 symmetric eigen-decomposition: pysph/base/linalg3.pyx:259-560    */
/* (EISPACK tred2 + tql2 as the reference carries them), operation order */
/* kept so results match the compiled reference bit for bit.            */
/* ------------------------------------------------------------------ */
#define EN 3
static inline double hypot2(double x, double y) { return sqrt(x * x + y * y); } /* linalg3.pyx:30-31 */

static void eig_tred2(double V[EN][EN], double *d, double *e) /* linalg3.pyx:259-378 */
{
    int i, j, k;
    double scale, f, g, h, hh;
    for (j = 0; j < EN; j++) d[j] = V[EN - 1][j];
    for (i = EN - 1; i > 0; i--) {
        scale = 0.0; h = 0.0;
        for (k = 0; k < i; k++) scale += fabs(d[k]);
        if (scale == 0.0) {
            e[i] = d[i - 1];
            for (j = 0; j < i; j++) { d[j] = V[i - 1][j]; V[i][j] = 0.0; V[j][i] = 0.0; }
        } else {
            for (k = 0; k < i; k++) { d[k] /= scale; h += d[k] * d[k]; }
            f = d[i - 1];
            g = sqrt(h);
            if (f > 0) g = -g;
            e[i] = scale * g;
            h = h - f * g;
            d[i - 1] = f - g;
            for (j = 0; j < i; j++) e[j] = 0.0;
            for (j = 0; j < i; j++) {
                f = d[j];
                V[j][i] = f;
                g = e[j] + V[j][j] * f;
                for (k = j + 1; k < i; k++) { g += V[k][j] * d[k]; e[k] += V[k][j] * f; }
                e[j] = g;
            }
            f = 0.0;
            for (j = 0; j < i; j++) { e[j] /= h; f += e[j] * d[j]; }
            hh = f / (h + h);
            for (j = 0; j < i; j++) e[j] -= hh * d[j];
            for (j = 0; j < i; j++) {
                f = d[j]; g = e[j];
                for (k = j; k < i; k++) V[k][j] -= (f * e[k] + g * d[k]);
                d[j] = V[i - 1][j];
                V[i][j] = 0.0;
            }
        }
        d[i] = h;
    }
    for (i = 0; i < EN - 1; i++) {
        V[EN - 1][i] = V[i][i];
        V[i][i] = 1.0;
        h = d[i + 1];
        if (h != 0.0) {
            for (k = 0; k < i + 1; k++) d[k] = V[k][i + 1] / h;
            for (j = 0; j < i + 1; j++) {
                g = 0.0;
                for (k = 0; k < i + 1; k++) g += V[k][i + 1] * V[k][j];
                for (k = 0; k < i + 1; k++) V[k][j] -= g * d[k];
            }
        }
        for (k = 0; k < i + 1; k++) V[k][i + 1] = 0.0;
    }
    for (j = 0; j < EN; j++) { d[j] = V[EN - 1][j]; V[EN - 1][j] = 0.0; }
    V[EN - 1][EN - 1] = 1.0;
    e[0] = 0.0;
}

static void eig_tql2(double V[EN][EN], double *d, double *e) /* linalg3.pyx:381-500 */
{
    int i, j, k, l, m;
    double f, tst1, eps, g, h, p, r, dl1, c, c2, c3, el1, s, s2;
    for (i = 1; i < EN; i++) e[i - 1] = e[i];
    e[EN - 1] = 0.0;
    f = 0.0; tst1 = 0.0;
    eps = pow(2.0, -52.0);
    for (l = 0; l < EN; l++) {
        tst1 = fmax(tst1, fabs(d[l]) + fabs(e[l]));
        m = l;
        while (m < EN) {
            if (fabs(e[m]) <= eps * tst1) break;
            m += 1;
        }
        if (m > l) {
            int cont = 1;
            while (cont) {
                g = d[l];
                p = (d[l + 1] - g) / (2.0 * e[l]);
                r = hypot2(p, 1.0);
                if (p < 0) r = -r;
                d[l] = e[l] / (p + r);
                d[l + 1] = e[l] * (p + r);
                dl1 = d[l + 1];
                h = g - d[l];
                for (i = l + 2; i < EN; i++) d[i] -= h;
                f += h;
                p = d[m];
                c = 1.0; c2 = c; c3 = c;
                el1 = e[l + 1];
                s = 0.0; s2 = 0.0;
                for (i = m - 1; i > l - 1; i--) {
                    c3 = c2; c2 = c; s2 = s;
                    g = c * e[i];
                    h = c * p;
                    r = hypot2(p, e[i]);
                    e[i + 1] = s * r;
                    s = e[i] / r;
                    c = p / r;
                    p = c * d[i] - s * g;
                    d[i + 1] = h + s * (c * g + s * d[i]);
                    for (k = 0; k < EN; k++) {
                        h = V[k][i + 1];
                        V[k][i + 1] = s * V[k][i] + c * h;
                        V[k][i] = c * V[k][i] - s * h;
                    }
                }
                p = -s * s2 * c3 * el1 * e[l] / dl1;
                e[l] = s * p;
                d[l] = c * p;
                cont = fabs(e[l]) > eps * tst1;
            }
        }
        d[l] += f;
        e[l] = 0.0;
    }
    for (i = 0; i < EN - 1; i++) {
        k = i; p = d[i];
        for (j = i + 1; j < EN; j++) if (d[j] < p) { k = j; p = d[j]; }
        if (k != i) {
            d[k] = d[i]; d[i] = p;
            for (j = 0; j < EN; j++) { p = V[j][i]; V[j][i] = V[j][k]; V[j][k] = p; }
        }
    }
}

/* linalg3.pyx:509-536 */
static void eigen_decomposition(double A[EN][EN], double V[EN][EN], double *d)
{
    double e[EN];
    double s = 0.0;
    int i, j;
    for (i = 0; i < EN; i++)
        for (j = 0; j < EN; j++) { V[i][j] = A[i][j]; s += fabs(V[i][j]); }
    if (s == 0) {
        for (i = 0; i < 3; i++) { d[i] = 0.0; for (j = 0; j < 3; j++) V[i][j] = (i == j); }
    } else {
        for (i = 0; i < EN; i++) for (j = 0; j < EN; j++) V[i][j] /= s;
        eig_tred2(V, d, e);
        eig_tql2(V, d, e);
        for (i = 0; i < EN; i++) d[i] *= s;
    }
}

/* P*A*P.T with A diagonal: linalg3.pyx:220-234 */
static void transform_diag_inv(const double *A, double P[3][3], double res[3][3])
{
    int i, j, k;
    for (i = 0; i < 3; i++) for (j = 0; j < 3; j++) res[i][j] = 0.0;
    for (i = 0; i < 3; i++) for (j = 0; j < 3; j++) for (k = 0; k < 3; k++) res[i][j] += P[i][k] * A[k] * P[j][k];
}

void orc_eigen3(const double *A9, double *V9, double *d3)
{
    double A[3][3], V[3][3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) A[i][j] = A9[3 * i + j];
    eigen_decomposition(A, V, d3);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) V9[3 * i + j] = V[i][j];
}

/* ------------------------------------------------------------------ */
/* equations                                                            */
/* ------------------------------------------------------------------ */
#define DP(prop) (D->p[prop])
#define SP(prop) (S->p[prop])

static void eq_initialize(const orc_equation *e, const orc_array *D, long d_idx)
{
    switch (e->kind) {
    case OE_CONTINUITY: DP(OP_ARHO)[d_idx] = 0.0; break;              /* basic_equations.py:184 */
    case OE_MOMENTUM:                                                  /* wc/basic.py:198-202 */
        DP(OP_AU)[d_idx] = 0.0; DP(OP_AV)[d_idx] = 0.0; DP(OP_AW)[d_idx] = 0.0;
        DP(OP_DT_CFL)[d_idx] = 0.0;
        break;
    case OE_XSPH:                                                      /* basic_equations.py:285-288 */
        DP(OP_AX)[d_idx] = 0.0; DP(OP_AY)[d_idx] = 0.0; DP(OP_AZ)[d_idx] = 0.0;
        break;
    case OE_SUMMATION_DENSITY: DP(OP_RHO)[d_idx] = 0.0; break;        /* basic_equations.py:25 */
    case OE_TVF_SUMMATION_DENSITY:                                     /* transport_velocity.py:52-54 */
        DP(OP_VOL)[d_idx] = 0.0; DP(OP_RHO)[d_idx] = 0.0;
        break;
    case OE_TVF_MOM_PRESSURE:                                          /* :282-288 */
        DP(OP_AU)[d_idx] = 0.0; DP(OP_AV)[d_idx] = 0.0; DP(OP_AW)[d_idx] = 0.0;
        DP(OP_AUHAT)[d_idx] = 0.0; DP(OP_AVHAT)[d_idx] = 0.0; DP(OP_AWHAT)[d_idx] = 0.0;
        break;
    case OE_TVF_MOM_VISCOSITY:
    case OE_TVF_MOM_ART_VISCOSITY:
    case OE_TVF_MOM_ART_STRESS:
    case OE_MONAGHAN_ART_VISCOSITY:
    case OE_MOMENTUM_WITH_STRESS: /* solid_mech/basic.py:262-265 */
        DP(OP_AU)[d_idx] = 0.0; DP(OP_AV)[d_idx] = 0.0; DP(OP_AW)[d_idx] = 0.0;
        break;
    case OE_VELOCITY_GRADIENT_2D: /* basic_equations.py:82-86 */
        DP(OP_V00)[d_idx] = 0.0; DP(OP_V01)[d_idx] = 0.0; DP(OP_V10)[d_idx] = 0.0; DP(OP_V11)[d_idx] = 0.0;
        break;
    case OE_VELOCITY_GRADIENT_3D: /* basic_equations.py:115-128 */
        DP(OP_V00)[d_idx] = 0.0; DP(OP_V01)[d_idx] = 0.0; DP(OP_V02)[d_idx] = 0.0;
        DP(OP_V10)[d_idx] = 0.0; DP(OP_V11)[d_idx] = 0.0; DP(OP_V12)[d_idx] = 0.0;
        DP(OP_V20)[d_idx] = 0.0; DP(OP_V21)[d_idx] = 0.0; DP(OP_V22)[d_idx] = 0.0;
        break;
    case OE_HOOKES_DEVIATORIC_STRESS_RATE: /* solid_mech/basic.py:409-416 */
        DP(OP_AS00)[d_idx] = 0.0; DP(OP_AS01)[d_idx] = 0.0; DP(OP_AS02)[d_idx] = 0.0;
        DP(OP_AS11)[d_idx] = 0.0; DP(OP_AS12)[d_idx] = 0.0; DP(OP_AS22)[d_idx] = 0.0;
        break;
    default: break;
    }
}

/* loop() of equations with no sources */
static void eq_loop_nosrc(const orc_equation *e, const orc_array *D, long d_idx)
{
    const double *par = e->par;
    switch (e->kind) {
    case OE_TAIT_EOS: { /* wc/basic.py:51-65; par rho0 c0 gamma p0 */
        double rho0 = par[0], c0 = par[1], gamma = par[2], p0 = par[3];
        double rho01 = 1.0 / rho0, gamma1 = 0.5 * (gamma - 1.0), B = rho0 * c0 * c0 / gamma;
        double ratio = DP(OP_RHO)[d_idx] * rho01;
        double tmp = pow(ratio, gamma);
        DP(OP_P)[d_idx] = p0 + B * (tmp - 1.0);
        DP(OP_CS)[d_idx] = c0 * pow(ratio, gamma1);
        break;
    }
    case OE_TAIT_EOS_HG: { /* wc/basic.py:110-126 */
        double rho0 = par[0], c0 = par[1], gamma = par[2];
        double rho01 = 1.0 / rho0, gamma1 = 0.5 * (gamma - 1.0), B = rho0 * c0 * c0 / gamma;
        if (DP(OP_RHO)[d_idx] < rho0) DP(OP_RHO)[d_idx] = rho0;
        double ratio = DP(OP_RHO)[d_idx] * rho01;
        double tmp = pow(ratio, gamma);
        DP(OP_P)[d_idx] = B * (tmp - 1.0);
        DP(OP_CS)[d_idx] = c0 * pow(ratio, gamma1);
        break;
    }
    case OE_TVF_STATE_EQUATION: { /* transport_velocity.py:215-216; par p0 rho0 b */
        DP(OP_P)[d_idx] = par[0] * (DP(OP_RHO)[d_idx] / par[1] - par[2]);
        break;
    }
    case OE_ISOTHERMAL_EOS: { /* basic_equations.py:151-176; par rho0 c0 p0 */
        double c02 = par[1] * par[1];
        DP(OP_P)[d_idx] = par[2] + c02 * (DP(OP_RHO)[d_idx] - par[0]);
        break;
    }
    case OE_SOLID_ISOTHERMAL_EOS: /* solid_mech/basic.py:100-101; par c0_ref rho_ref */
        DP(OP_P)[d_idx] = par[0] * par[0] * (DP(OP_RHO)[d_idx] - par[1]);
        break;
    case OE_MONAGHAN_ART_STRESS: { /* solid_mech/basic.py:170-242; par eps */
        double rhoi = DP(OP_RHO)[d_idx];
        double rhoi21 = 1. / (rhoi * rhoi);
        double R[3][3], Rab[3][3], S[3][3], V[3], rd[3];
        S[0][0] = DP(OP_S00)[d_idx] - DP(OP_P)[d_idx];
        S[1][1] = DP(OP_S11)[d_idx] - DP(OP_P)[d_idx];
        S[2][2] = DP(OP_S22)[d_idx] - DP(OP_P)[d_idx];
        S[1][2] = DP(OP_S12)[d_idx]; S[2][1] = DP(OP_S12)[d_idx];
        S[0][2] = DP(OP_S02)[d_idx]; S[2][0] = DP(OP_S02)[d_idx];
        S[0][1] = DP(OP_S01)[d_idx]; S[1][0] = DP(OP_S01)[d_idx];
        eigen_decomposition(S, R, V);
        for (int k = 0; k < 3; k++) rd[k] = V[k] > 0 ? -par[0] * V[k] * rhoi21 : 0;
        transform_diag_inv(rd, R, Rab);
        DP(OP_R00)[d_idx] = Rab[0][0]; DP(OP_R11)[d_idx] = Rab[1][1]; DP(OP_R22)[d_idx] = Rab[2][2];
        DP(OP_R12)[d_idx] = Rab[1][2]; DP(OP_R02)[d_idx] = Rab[0][2]; DP(OP_R01)[d_idx] = Rab[0][1];
        break;
    }
    case OE_HOOKES_DEVIATORIC_STRESS_RATE: { /* solid_mech/basic.py:418-505; par G */
        double v00 = DP(OP_V00)[d_idx], v01 = DP(OP_V01)[d_idx], v02 = DP(OP_V02)[d_idx];
        double v10 = DP(OP_V10)[d_idx], v11 = DP(OP_V11)[d_idx], v12 = DP(OP_V12)[d_idx];
        double v20 = DP(OP_V20)[d_idx], v21 = DP(OP_V21)[d_idx], v22 = DP(OP_V22)[d_idx];
        double s00 = DP(OP_S00)[d_idx], s01 = DP(OP_S01)[d_idx], s02 = DP(OP_S02)[d_idx];
        double s10 = s01, s11 = DP(OP_S11)[d_idx], s12 = DP(OP_S12)[d_idx];
        double s20 = s02, s21 = s12, s22 = DP(OP_S22)[d_idx];
        double eps00 = v00, eps01 = 0.5 * (v01 + v10), eps02 = 0.5 * (v02 + v20);
        double eps11 = v11, eps12 = 0.5 * (v12 + v21), eps22 = v22;
        double omega00 = 0.0, omega01 = 0.5 * (v01 - v10), omega02 = 0.5 * (v02 - v20);
        double omega10 = -omega01, omega11 = 0.0, omega12 = 0.5 * (v12 - v21);
        double omega20 = -omega02, omega21 = -omega12, omega22 = 0.0;
        double tmp = 2.0 * par[0];
        double trace = 1.0 / 3.0 * (eps00 + eps11 + eps22);
        DP(OP_AS00)[d_idx] = tmp * (eps00 - trace) + (s00 * omega00 + s01 * omega01 + s02 * omega02) +
                             (s00 * omega00 + s10 * omega01 + s20 * omega02);
        DP(OP_AS01)[d_idx] = tmp * (eps01) + (s00 * omega10 + s01 * omega11 + s02 * omega12) +
                             (s01 * omega00 + s11 * omega01 + s21 * omega02);
        DP(OP_AS02)[d_idx] = tmp * eps02 + (s00 * omega20 + s01 * omega21 + s02 * omega22) +
                             (s02 * omega00 + s12 * omega01 + s22 * omega02);
        DP(OP_AS11)[d_idx] = tmp * (eps11 - trace) + (s10 * omega10 + s11 * omega11 + s12 * omega12) +
                             (s01 * omega10 + s11 * omega11 + s21 * omega12);
        DP(OP_AS12)[d_idx] = tmp * eps12 + (s10 * omega20 + s11 * omega21 + s12 * omega22) +
                             (s02 * omega10 + s12 * omega11 + s22 * omega12);
        DP(OP_AS22)[d_idx] = tmp * (eps22 - trace) + (s20 * omega20 + s21 * omega21 + s22 * omega22) +
                             (s02 * omega20 + s12 * omega21 + s22 * omega22);
        break;
    }
    default: break;
    }
}

static void eq_loop(const orc_equation *e, const orc_array *D, const orc_array *S,
                    long d_idx, long s_idx, const pair_t *P)
{
    const double *par = e->par;
    const double *VIJ = P->VIJ, *XIJ = P->XIJ, *DWIJ = P->DWIJ;
    switch (e->kind) {
    case OE_CONTINUITY: { /* basic_equations.py:187-192 */
        double vijdotdwij = DWIJ[0] * VIJ[0] + DWIJ[1] * VIJ[1] + DWIJ[2] * VIJ[2];
        DP(OP_ARHO)[d_idx] += SP(OP_M)[s_idx] * vijdotdwij;
        break;
    }
    case OE_MOMENTUM: { /* wc/basic.py:204-259; par c0 alpha beta gx gy gz tensile */
        double c0 = par[0], alpha = par[1], beta = par[2];
        int tensile = par[6] != 0.0;
        double rhoi21 = 1.0 / (DP(OP_RHO)[d_idx] * DP(OP_RHO)[d_idx]);
        double rhoj21 = 1.0 / (SP(OP_RHO)[s_idx] * SP(OP_RHO)[s_idx]);
        double vijdotxij = VIJ[0] * XIJ[0] + VIJ[1] * XIJ[1] + VIJ[2] * XIJ[2];
        double piij = 0.0;
        if (vijdotxij < 0) {
            double cij = 0.5 * (DP(OP_CS)[d_idx] + SP(OP_CS)[s_idx]);
            double muij = (P->HIJ * vijdotxij) / (P->R2IJ + P->EPS);
            piij = -alpha * cij * muij + beta * muij * muij;
            piij = piij * P->RHOIJ1;
        }
        double _dt_cfl = 0.0;
        if (P->R2IJ > 1e-12) {
            _dt_cfl = fabs(P->HIJ * vijdotxij / P->R2IJ) + c0;
            DP(OP_DT_CFL)[d_idx] = fmax(_dt_cfl, DP(OP_DT_CFL)[d_idx]);
        }
        double tmpi = DP(OP_P)[d_idx] * rhoi21;
        double tmpj = SP(OP_P)[s_idx] * rhoj21;
        double fij = P->WIJ / P->WDP;
        double Ri = 0.0, Rj = 0.0;
        if (tensile) {
            fij = fij * fij;
            fij = fij * fij;
            if (DP(OP_P)[d_idx] > 0) Ri = 0.01 * tmpi; else Ri = 0.2 * fabs(tmpi);
            if (SP(OP_P)[s_idx] > 0) Rj = 0.01 * tmpj; else Rj = 0.2 * fabs(tmpj);
        }
        double tmp = (tmpi + tmpj) + (Ri + Rj) * fij;
        DP(OP_AU)[d_idx] += -SP(OP_M)[s_idx] * (tmp + piij) * DWIJ[0];
        DP(OP_AV)[d_idx] += -SP(OP_M)[s_idx] * (tmp + piij) * DWIJ[1];
        DP(OP_AW)[d_idx] += -SP(OP_M)[s_idx] * (tmp + piij) * DWIJ[2];
        break;
    }
    case OE_XSPH: { /* basic_equations.py:290-295; par eps */
        double tmp = -par[0] * SP(OP_M)[s_idx] * P->WIJ * P->RHOIJ1;
        DP(OP_AX)[d_idx] += tmp * VIJ[0];
        DP(OP_AY)[d_idx] += tmp * VIJ[1];
        DP(OP_AZ)[d_idx] += tmp * VIJ[2];
        break;
    }
    case OE_SUMMATION_DENSITY: /* basic_equations.py:28-29 */
        DP(OP_RHO)[d_idx] += SP(OP_M)[s_idx] * P->WIJ;
        break;
    case OE_TVF_SUMMATION_DENSITY: /* transport_velocity.py:56-58 */
        DP(OP_VOL)[d_idx] += P->WIJ;
        DP(OP_RHO)[d_idx] += DP(OP_M)[d_idx] * P->WIJ;
        break;
    case OE_TVF_MOM_PRESSURE: { /* :290-320; par pb gx gy gz tdamp */
        double rhoi = DP(OP_RHO)[d_idx], rhoj = SP(OP_RHO)[s_idx];
        double p_i = DP(OP_P)[d_idx], p_j = SP(OP_P)[s_idx];
        double pij = rhoj * p_i + rhoi * p_j;
        pij /= (rhoj + rhoi);
        double Vi = 1. / DP(OP_VOL)[d_idx], Vj = 1. / SP(OP_VOL)[s_idx];
        double Vi2 = Vi * Vi, Vj2 = Vj * Vj;
        double mi1 = 1.0 / DP(OP_M)[d_idx];
        double tmp = -pij * mi1 * (Vi2 + Vj2);
        DP(OP_AU)[d_idx] += tmp * DWIJ[0];
        DP(OP_AV)[d_idx] += tmp * DWIJ[1];
        DP(OP_AW)[d_idx] += tmp * DWIJ[2];
        tmp = -par[0] * mi1 * (Vi2 + Vj2);
        DP(OP_AUHAT)[d_idx] += tmp * DWIJ[0];
        DP(OP_AVHAT)[d_idx] += tmp * DWIJ[1];
        DP(OP_AWHAT)[d_idx] += tmp * DWIJ[2];
        break;
    }
    case OE_TVF_MOM_VISCOSITY: { /* :363-384; par nu */
        double etai = par[0] * DP(OP_RHO)[d_idx], etaj = par[0] * SP(OP_RHO)[s_idx];
        double etaij = 2 * (etai * etaj) / (etai + etaj);
        double Fij = DWIJ[0] * XIJ[0] + DWIJ[1] * XIJ[1] + DWIJ[2] * XIJ[2];
        double Vi = 1. / DP(OP_VOL)[d_idx], Vj = 1. / SP(OP_VOL)[s_idx];
        double Vi2 = Vi * Vi, Vj2 = Vj * Vj;
        double tmp = 1. / DP(OP_M)[d_idx] * (Vi2 + Vj2) * etaij * Fij / (P->R2IJ + P->EPS);
        DP(OP_AU)[d_idx] += tmp * VIJ[0];
        DP(OP_AV)[d_idx] += tmp * VIJ[1];
        DP(OP_AW)[d_idx] += tmp * VIJ[2];
        break;
    }
    case OE_TVF_MOM_ART_VISCOSITY: { /* :420-436; par c0 alpha */
        double vijdotrij = VIJ[0] * XIJ[0] + VIJ[1] * XIJ[1] + VIJ[2] * XIJ[2];
        double piij = 0.0;
        if (vijdotrij < 0) {
            double muij = (P->HIJ * vijdotrij) / (P->R2IJ + P->EPS);
            piij = -par[1] * par[0] * muij;
            piij = SP(OP_M)[s_idx] * piij * P->RHOIJ1;
        }
        DP(OP_AU)[d_idx] += -piij * DWIJ[0];
        DP(OP_AV)[d_idx] += -piij * DWIJ[1];
        DP(OP_AW)[d_idx] += -piij * DWIJ[2];
        break;
    }
    case OE_TVF_MOM_ART_STRESS: { /* :473-545 */
        double rhoi = DP(OP_RHO)[d_idx], rhoj = SP(OP_RHO)[s_idx];
        double ui = DP(OP_U)[d_idx], uhati = DP(OP_UHAT)[d_idx];
        double vi = DP(OP_V)[d_idx], vhati = DP(OP_VHAT)[d_idx];
        double wi = DP(OP_W)[d_idx], whati = DP(OP_WHAT)[d_idx];
        double uj = SP(OP_U)[s_idx], uhatj = SP(OP_UHAT)[s_idx];
        double vj = SP(OP_V)[s_idx], vhatj = SP(OP_VHAT)[s_idx];
        double wj = SP(OP_W)[s_idx], whatj = SP(OP_WHAT)[s_idx];
        double Vi = 1. / DP(OP_VOL)[d_idx], Vj = 1. / SP(OP_VOL)[s_idx];
        double Vi2 = Vi * Vi, Vj2 = Vj * Vj;
        double Axxi = rhoi * ui * (uhati - ui), Axyi = rhoi * ui * (vhati - vi), Axzi = rhoi * ui * (whati - wi);
        double Ayxi = rhoi * vi * (uhati - ui), Ayyi = rhoi * vi * (vhati - vi), Ayzi = rhoi * vi * (whati - wi);
        double Azxi = rhoi * wi * (uhati - ui), Azyi = rhoi * wi * (vhati - vi), Azzi = rhoi * wi * (whati - wi);
        double Axxj = rhoj * uj * (uhatj - uj), Axyj = rhoj * uj * (vhatj - vj), Axzj = rhoj * uj * (whatj - wj);
        double Ayxj = rhoj * vj * (uhatj - uj), Ayyj = rhoj * vj * (vhatj - vj), Ayzj = rhoj * vj * (whatj - wj);
        double Azxj = rhoj * wj * (uhatj - uj), Azyj = rhoj * wj * (vhatj - vj), Azzj = rhoj * wj * (whatj - wj);
        double Ax = 0.5 * ((Axxi + Axxj) * DWIJ[0] + (Axyi + Axyj) * DWIJ[1] + (Axzi + Axzj) * DWIJ[2]);
        double Ay = 0.5 * ((Ayxi + Ayxj) * DWIJ[0] + (Ayyi + Ayyj) * DWIJ[1] + (Ayzi + Ayzj) * DWIJ[2]);
        double Az = 0.5 * ((Azxi + Azxj) * DWIJ[0] + (Azyi + Azyj) * DWIJ[1] + (Azzi + Azzj) * DWIJ[2]);
        double tmp = 1. / DP(OP_M)[d_idx] * (Vi2 + Vj2);
        DP(OP_AU)[d_idx] += tmp * Ax;
        DP(OP_AV)[d_idx] += tmp * Ay;
        DP(OP_AW)[d_idx] += tmp * Az;
        break;
    }
    case OE_VELOCITY_GRADIENT_2D: { /* basic_equations.py:88-98 */
        double tmp = SP(OP_M)[s_idx] / SP(OP_RHO)[s_idx];
        DP(OP_V00)[d_idx] += tmp * -VIJ[0] * DWIJ[0];
        DP(OP_V01)[d_idx] += tmp * -VIJ[0] * DWIJ[1];
        DP(OP_V10)[d_idx] += tmp * -VIJ[1] * DWIJ[0];
        DP(OP_V11)[d_idx] += tmp * -VIJ[1] * DWIJ[1];
        break;
    }
    case OE_VELOCITY_GRADIENT_3D: { /* basic_equations.py:130-148 */
        double tmp = SP(OP_M)[s_idx] / SP(OP_RHO)[s_idx];
        DP(OP_V00)[d_idx] += tmp * -VIJ[0] * DWIJ[0];
        DP(OP_V01)[d_idx] += tmp * -VIJ[0] * DWIJ[1];
        DP(OP_V02)[d_idx] += tmp * -VIJ[0] * DWIJ[2];
        DP(OP_V10)[d_idx] += tmp * -VIJ[1] * DWIJ[0];
        DP(OP_V11)[d_idx] += tmp * -VIJ[1] * DWIJ[1];
        DP(OP_V12)[d_idx] += tmp * -VIJ[1] * DWIJ[2];
        DP(OP_V20)[d_idx] += tmp * -VIJ[2] * DWIJ[0];
        DP(OP_V21)[d_idx] += tmp * -VIJ[2] * DWIJ[1];
        DP(OP_V22)[d_idx] += tmp * -VIJ[2] * DWIJ[2];
        break;
    }
    case OE_MOMENTUM_WITH_STRESS: { /* solid_mech/basic.py:267-387; par wdeltap n */
        double pa = DP(OP_P)[d_idx], pb = SP(OP_P)[s_idx];
        double rhoa = DP(OP_RHO)[d_idx], rhob = SP(OP_RHO)[s_idx];
        double rhoa21 = 1. / (rhoa * rhoa), rhob21 = 1. / (rhob * rhob);
        double s00a = DP(OP_S00)[d_idx], s01a = DP(OP_S01)[d_idx], s02a = DP(OP_S02)[d_idx];
        double s10a = s01a, s11a = DP(OP_S11)[d_idx], s12a = DP(OP_S12)[d_idx];
        double s20a = s02a, s21a = s12a, s22a = DP(OP_S22)[d_idx];
        double s00b = SP(OP_S00)[s_idx], s01b = SP(OP_S01)[s_idx], s02b = SP(OP_S02)[s_idx];
        double s10b = s01b, s11b = SP(OP_S11)[s_idx], s12b = SP(OP_S12)[s_idx];
        double s20b = s02b, s21b = s12b, s22b = SP(OP_S22)[s_idx];
        double r00a = DP(OP_R00)[d_idx], r01a = DP(OP_R01)[d_idx], r02a = DP(OP_R02)[d_idx];
        double r11a = DP(OP_R11)[d_idx], r12a = DP(OP_R12)[d_idx], r22a = DP(OP_R22)[d_idx];
        double r00b = SP(OP_R00)[s_idx], r01b = SP(OP_R01)[s_idx], r02b = SP(OP_R02)[s_idx];
        double r11b = SP(OP_R11)[s_idx], r12b = SP(OP_R12)[s_idx], r22b = SP(OP_R22)[s_idx];
        s00a = s00a - pa; s00b = s00b - pb;
        s11a = s11a - pa; s11b = s11b - pb;
        s22a = s22a - pa; s22b = s22b - pb;
        double as00, as01, as02, as10, as11, as12, as20, as21, as22;
        if (par[0] > 0.) {
            double fab = P->WIJ / par[0];
            fab = pow(fab, par[1]);
            as00 = fab * (r00a + r00b); as01 = fab * (r01a + r01b); as02 = fab * (r02a + r02b);
            as10 = as01; as11 = fab * (r11a + r11b); as12 = fab * (r12a + r12b);
            as20 = as02; as21 = as12; as22 = fab * (r22a + r22b);
        } else {
            as00 = 0.0; as01 = 0.0; as02 = 0.0; as10 = as01; as11 = 0.0; as12 = 0.0;
            as20 = as02; as21 = as12; as22 = 0.0;
        }
        double mb = SP(OP_M)[s_idx];
        DP(OP_AU)[d_idx] += (mb * (s00a * rhoa21 + s00b * rhob21 + as00) * DWIJ[0] +
                             mb * (s01a * rhoa21 + s01b * rhob21 + as01) * DWIJ[1] +
                             mb * (s02a * rhoa21 + s02b * rhob21 + as02) * DWIJ[2]);
        DP(OP_AV)[d_idx] += (mb * (s10a * rhoa21 + s10b * rhob21 + as10) * DWIJ[0] +
                             mb * (s11a * rhoa21 + s11b * rhob21 + as11) * DWIJ[1] +
                             mb * (s12a * rhoa21 + s12b * rhob21 + as12) * DWIJ[2]);
        DP(OP_AW)[d_idx] += (mb * (s20a * rhoa21 + s20b * rhob21 + as20) * DWIJ[0] +
                             mb * (s21a * rhoa21 + s21b * rhob21 + as21) * DWIJ[1] +
                             mb * (s22a * rhoa21 + s22b * rhob21 + as22) * DWIJ[2]);
        break;
    }
    case OE_MONAGHAN_ART_VISCOSITY: { /* basic_equations.py:236-257; par alpha beta */
        double vijdotxij = VIJ[0] * XIJ[0] + VIJ[1] * XIJ[1] + VIJ[2] * XIJ[2];
        double piij = 0.0;
        if (vijdotxij < 0) {
            double cij = 0.5 * (DP(OP_CS)[d_idx] + SP(OP_CS)[s_idx]);
            double muij = (P->HIJ * vijdotxij) / (P->R2IJ + P->EPS);
            piij = -par[0] * cij * muij + par[1] * muij * muij;
            piij = piij * P->RHOIJ1;
        }
        DP(OP_AU)[d_idx] += -SP(OP_M)[s_idx] * piij * DWIJ[0];
        DP(OP_AV)[d_idx] += -SP(OP_M)[s_idx] * piij * DWIJ[1];
        DP(OP_AW)[d_idx] += -SP(OP_M)[s_idx] * piij * DWIJ[2];
        break;
    }
    default: break;
    }
}

static void eq_post_loop(const orc_equation *e, const orc_array *D, long d_idx, double t)
{
    const double *par = e->par;
    switch (e->kind) {
    case OE_MOMENTUM: { /* wc/basic.py:261-271 */
        DP(OP_AU)[d_idx] += par[3];
        DP(OP_AV)[d_idx] += par[4];
        DP(OP_AW)[d_idx] += par[5];
        double acc2 = (DP(OP_AU)[d_idx] * DP(OP_AU)[d_idx] +
                       DP(OP_AV)[d_idx] * DP(OP_AV)[d_idx] +
                       DP(OP_AW)[d_idx] * DP(OP_AW)[d_idx]);
        DP(OP_DT_FORCE)[d_idx] = acc2;
        break;
    }
    case OE_XSPH: /* basic_equations.py:297-300 */
        DP(OP_AX)[d_idx] += DP(OP_U)[d_idx];
        DP(OP_AY)[d_idx] += DP(OP_V)[d_idx];
        DP(OP_AZ)[d_idx] += DP(OP_W)[d_idx];
        break;
    case OE_TVF_MOM_PRESSURE: { /* transport_velocity.py:322-325; par pb gx gy gz tdamp */
        double damping_factor = 1.0;
        if (t < par[4]) damping_factor = 0.5 * (sin((-0.5 + t / par[4]) * M_PI) + 1.0);
        DP(OP_AU)[d_idx] += par[1] * damping_factor;
        DP(OP_AV)[d_idx] += par[2] * damping_factor;
        DP(OP_AW)[d_idx] += par[3] * damping_factor;
        break;
    }
    default: break;
    }
}

static int eq_has(int kind, int what) /* what: 0 initialize, 1 loop, 2 post_loop */
{
    switch (kind) {
    case OE_TAIT_EOS: case OE_TAIT_EOS_HG: case OE_TVF_STATE_EQUATION: case OE_ISOTHERMAL_EOS:
    case OE_SOLID_ISOTHERMAL_EOS: case OE_MONAGHAN_ART_STRESS:
        return what == 1;
    case OE_HOOKES_DEVIATORIC_STRESS_RATE: return what != 2;
    case OE_MOMENTUM: case OE_XSPH: case OE_TVF_MOM_PRESSURE: return 1;
    default: return what != 2;
    }
}

/* ------------------------------------------------------------------ */
/* AccelerationEval.compute: acceleration_eval_cython.mako:10-154       */
/* MegaGroup regrouping:      acceleration_eval.py:94-162               */
/* ------------------------------------------------------------------ */
static int do_group(orc_nnps *nn, const orc_kernel *K, const orc_group *g, double t, double dt,
                    int nthreads)
{
    (void)dt;
    /* destinations in first-appearance order */
    int dests[ORC_MAX_ARRAYS], ndest = 0;
    for (int i = 0; i < g->neq; i++) {
        int d = g->eqs[i].dest, seen = 0;
        for (int j = 0; j < ndest; j++) seen |= dests[j] == d;
        if (!seen) dests[ndest++] = d;
    }
    for (int di = 0; di < ndest; di++) {
        int dst = dests[di];
        const orc_array *D = nn->arr[dst];
        long start = g->start_idx;
        long np_dest = g->stop_idx >= 0 ? g->stop_idx : (g->real ? D->n_real : D->n);
        if (np_dest > D->n) np_dest = D->n; /* a stop_idx beyond a destination is out of bounds in the reference */

        /* sources in first-appearance order over this destination's equations */
        int srcs[ORC_MAX_ARRAYS], nsrcs = 0;
        for (int i = 0; i < g->neq; i++) {
            const orc_equation *e = &g->eqs[i];
            if (e->dest != dst) continue;
            for (int k = 0; k < e->nsrc; k++) {
                int seen = 0;
                for (int j = 0; j < nsrcs; j++) seen |= srcs[j] == e->src[k];
                if (!seen) srcs[nsrcs++] = e->src[k];
            }
        }
        /* initialize: all equations of this destination (mako :36-46) */
#pragma omp parallel for schedule(static) num_threads(nthreads)
        for (long d_idx = start; d_idx < np_dest; d_idx++)
            for (int i = 0; i < g->neq; i++)
                if (g->eqs[i].dest == dst && eq_has(g->eqs[i].kind, 0))
                    eq_initialize(&g->eqs[i], D, d_idx);
        /* equations with no source (mako :50-58) */
#pragma omp parallel for schedule(static) num_threads(nthreads)
        for (long d_idx = start; d_idx < np_dest; d_idx++)
            for (int i = 0; i < g->neq; i++)
                if (g->eqs[i].dest == dst && g->eqs[i].nsrc == 0)
                    eq_loop_nosrc(&g->eqs[i], D, d_idx);
        /* per source pair loops (mako :62-110) */
        for (int si = 0; si < nsrcs; si++) {
            int src = srcs[si];
            const orc_array *S = nn->arr[src];
            const orc_equation *act[64];
            int nact = 0;
            unsigned sym = 0;
            for (int i = 0; i < g->neq; i++) {
                const orc_equation *e = &g->eqs[i];
                if (e->dest != dst) continue;
                for (int k = 0; k < e->nsrc; k++)
                    if (e->src[k] == src) { act[nact++] = e; sym |= eq_symbols(e->kind); break; }
            }
            sym = close_symbols(sym);
#pragma omp parallel num_threads(nthreads)
            {
                long cap = 1024;
                unsigned *nbrs = (unsigned *)malloc(sizeof(unsigned) * (size_t)cap);
                pair_t P;
                memset(&P, 0, sizeof P);
#pragma omp for schedule(dynamic, 64)
                for (long d_idx = start; d_idx < np_dest; d_idx++) {
                    long nn_ = find_nearest_neighbors(nn, src, dst, d_idx, nbrs, cap);
                    if (nn_ > cap) {
                        cap = nn_ * 2;
                        nbrs = (unsigned *)realloc(nbrs, sizeof(unsigned) * (size_t)cap);
                        nn_ = find_nearest_neighbors(nn, src, dst, d_idx, nbrs, cap);
                    }
                    for (long ni = 0; ni < nn_; ni++) {
                        long s_idx = nbrs[ni];
                        /* precomputed symbols, dependency-level order (equation.py:300-344) */
                        if (sym & PS_HIJ) P.HIJ = 0.5 * (DP(OP_H)[d_idx] + SP(OP_H)[s_idx]);
                        if (sym & PS_RHOIJ) P.RHOIJ = 0.5 * (DP(OP_RHO)[d_idx] + SP(OP_RHO)[s_idx]);
                        if (sym & PS_VIJ) {
                            P.VIJ[0] = DP(OP_U)[d_idx] - SP(OP_U)[s_idx];
                            P.VIJ[1] = DP(OP_V)[d_idx] - SP(OP_V)[s_idx];
                            P.VIJ[2] = DP(OP_W)[d_idx] - SP(OP_W)[s_idx];
                        }
                        if (sym & PS_XIJ) {
                            P.XIJ[0] = DP(OP_X)[d_idx] - SP(OP_X)[s_idx];
                            P.XIJ[1] = DP(OP_Y)[d_idx] - SP(OP_Y)[s_idx];
                            P.XIJ[2] = DP(OP_Z)[d_idx] - SP(OP_Z)[s_idx];
                        }
                        if (sym & PS_EPS) P.EPS = 0.01 * P.HIJ * P.HIJ;
                        if (sym & PS_R2IJ)
                            P.R2IJ = P.XIJ[0] * P.XIJ[0] + P.XIJ[1] * P.XIJ[1] + P.XIJ[2] * P.XIJ[2];
                        if (sym & PS_RHOIJ1) P.RHOIJ1 = 1.0 / P.RHOIJ;
                        if (sym & PS_WDP) P.WDP = orc_kernel_w(K, K->deltap * P.HIJ, P.HIJ);
                        if (sym & PS_RIJ) P.RIJ = sqrt(P.R2IJ);
                        if (sym & PS_DWIJ) orc_kernel_gradient(K, P.XIJ, P.RIJ, P.HIJ, P.DWIJ);
                        if (sym & PS_WIJ) P.WIJ = orc_kernel_w(K, P.RIJ, P.HIJ);
                        for (int a = 0; a < nact; a++) eq_loop(act[a], D, S, d_idx, s_idx, &P);
                    }
                }
                free(nbrs);
            }
        }
        /* post_loop (mako :116-122) */
#pragma omp parallel for schedule(static) num_threads(nthreads)
        for (long d_idx = start; d_idx < np_dest; d_idx++)
            for (int i = 0; i < g->neq; i++)
                if (g->eqs[i].dest == dst && eq_has(g->eqs[i].kind, 2))
                    eq_post_loop(&g->eqs[i], D, d_idx, t);
    }
    return 0;
}

int orc_compute(orc_nnps *nn, const orc_kernel *K, const orc_group *groups, int ngroups,
                double t, double dt, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    for (int gi = 0; gi < ngroups; gi++) {
        int rc = do_group(nn, K, &groups[gi], t, dt, nthreads);
        if (rc) return rc;
    }
    return 0;
}
