// sph_eval.hip -- AccelerationEval.compute on the device.
//
// Replaces one leaf-group block of the code the reference generates from
// pysph/sph/acceleration_eval_cython.mako:10-154 (initialize -> no-source
// loops -> per-source [neighbours -> precomputed symbols -> Equation.loop] ->
// post_loop), with the destination/source regrouping of MegaGroup
// (pysph/sph/acceleration_eval.py:94-162) done here on the host.
//
// Design (gfx950, no MFMA -- an irregular gather, not a contraction):
//   * every array taking part in a (dest, sources) block is gathered into
//     cell order as packed records {x,y,z,h | equation-family aux} (+ a compact
//     fp32 position array for the prefilter) -> all pair-loop reads are
//     contiguous runs (a row of cells along x is one run, sorted along x to 1/8
//     of a cell);
//   * one fused kernel per destination does initialize + ALL sources + post_loop
//     with the sums held in registers and ONE write per output, no atomics;
//   * two schedules of the same arithmetic:
//       variant 6 (default) k_pair_wave   : one wavefront per 64 destinations,
//                                           fp32 prefilter tiles in LDS, per-lane
//                                           hit-mask slot lists (sph_pair.h),
//       variant 0           k_pair_direct : plain per-lane 27-cell walk, the
//                                           independent cross-check;
//     (schedules measured and dropped: DESIGN.md section 4);
//   * every family is a template on the arithmetic type: double (default) or
//     float (option arith_f32);
//   * the exact criterion of the reference (r2 < (k h_i)^2 or r2 < (k h_j)^2,
//     linked_list_nnps.pyx:176-184) decides neighbourhood in every schedule;
//     the fp32 test in front of it is a conservative superset filter.
#include "sph_internal.h"
#include "sph_kernels.h"
#include "sph_pair.h"

#include <cfloat>
#include <cmath>
#include <type_traits>

// ---------------------------------------------------------------------------
// equations without sources (elementwise)
// ---------------------------------------------------------------------------
struct EosArgs {
    int kind;
    double par[SPH_MAX_PAR];
    double *rho, *p, *cs;
    size_t start, stop;
    double *q[SPH_PROP_COUNT]; // every device property of the destination (elastic equations)
    uint32_t *tflag;           // MonaghanArtificialStress: set to 1 when any r_ij is non-zero (DevArray::tflag), or null
};

// Symmetric 3x3 eigen-decomposition by cyclic Jacobi rotations.  The reference
// uses EISPACK tred2/tql2 (pysph/base/linalg3.pyx:259-536); what the caller
// needs is the matrix function V f(lambda) V^T, which does not depend on the
// eigen-solver, its ordering or sign conventions -- only on its accuracy.
__device__ inline void jacobi_eigen3(double A[3][3], double V[3][3], double d[3])
{
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) V[i][j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 12; sweep++) {
        double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
        double diag = fabs(A[0][0]) + fabs(A[1][1]) + fabs(A[2][2]);
        if (off <= 1e-300 || off <= 1e-18 * diag) break;
#pragma unroll
        for (int pq = 0; pq < 3; pq++) {
            const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;
            double apq = A[p][q];
            if (apq == 0.0) continue;
            double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
            double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
            A[p][p] -= t * apq;
            A[q][q] += t * apq;
            A[p][q] = A[q][p] = 0.0;
            const int r = 3 - p - q;
            double arp = A[r][p], arq = A[r][q];
            A[r][p] = A[p][r] = c * arp - s * arq;
            A[r][q] = A[q][r] = s * arp + c * arq;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                double vkp = V[k][p], vkq = V[k][q];
                V[k][p] = c * vkp - s * vkq;
                V[k][q] = s * vkp + c * vkq;
            }
        }
    }
    d[0] = A[0][0]; d[1] = A[1][1]; d[2] = A[2][2];
}

__global__ __launch_bounds__(256) void k_nosrc(EosArgs a)
{
    size_t i = a.start + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.stop) return;
    switch (a.kind) {
    case SPH_EQ_TAIT_EOS: { // wc/basic.py:60-65
        double rho0 = a.par[0], c0 = a.par[1], gamma = a.par[2], p0 = a.par[3];
        double ratio = a.rho[i] * (1.0 / rho0);
        double tmp, csr;
        if (tait_gk(gamma) >= 0) { // odd integer exponents (7: water): the powers by multiplication (two pow() calls made this kernel compute-bound)
            tait_powers(ratio, tait_gk(gamma), tmp, csr);
        } else {
            tmp = pow(ratio, gamma);
            csr = pow(ratio, 0.5 * (gamma - 1.0));
        }
        a.p[i] = p0 + (rho0 * c0 * c0 / gamma) * (tmp - 1.0);
        a.cs[i] = c0 * csr;
        break;
    }
    case SPH_EQ_TAIT_EOS_HG: { // wc/basic.py:118-126
        double rho0 = a.par[0], c0 = a.par[1], gamma = a.par[2];
        double r = a.rho[i];
        if (r < rho0) { r = rho0; a.rho[i] = r; }
        double ratio = r * (1.0 / rho0);
        double tmp, csr;
        if (tait_gk(gamma) >= 0) { // as TaitEOS above: the EOS-fused pair kernel recomputes p, cs by the same multiplications
            tait_powers(ratio, tait_gk(gamma), tmp, csr);
        } else {
            tmp = pow(ratio, gamma);
            csr = pow(ratio, 0.5 * (gamma - 1.0));
        }
        a.p[i] = (rho0 * c0 * c0 / gamma) * (tmp - 1.0);
        a.cs[i] = c0 * csr;
        break;
    }
    case SPH_EQ_TVF_STATE_EQUATION: // transport_velocity.py:215-216; rho / rho0 as rho * (1 / rho0), like TaitEOS above: the
                                    // state-fused TVF pair kernel (FamTVFE_T) recomputes exactly this
        a.p[i] = a.par[0] * (a.rho[i] * (1.0 / a.par[1]) - a.par[2]);
        break;
    case SPH_EQ_ISOTHERMAL_EOS: // basic_equations.py:175-176
        a.p[i] = a.par[2] + (a.par[1] * a.par[1]) * (a.rho[i] - a.par[0]);
        break;
    case SPH_EQ_SOLID_ISOTHERMAL_EOS: // solid_mech/basic.py:100-101; par c0_ref rho_ref
        a.p[i] = a.par[0] * a.par[0] * (a.rho[i] - a.par[1]);
        break;
    case SPH_EQ_MONAGHAN_ART_STRESS: { // solid_mech/basic.py:170-242; par eps
        double **q = a.q;
        const double rhoi = a.rho[i], rhoi21 = 1. / (rhoi * rhoi), pr = a.p[i];
        double S[3][3], R[3][3], lam[3], rd[3];
        S[0][0] = q[SPH_S00][i] - pr; S[1][1] = q[SPH_S11][i] - pr; S[2][2] = q[SPH_S22][i] - pr;
        S[0][1] = S[1][0] = q[SPH_S01][i];
        S[0][2] = S[2][0] = q[SPH_S02][i];
        S[1][2] = S[2][1] = q[SPH_S12][i];
        // Gershgorin: every eigenvalue <= max_i (S_ii + sum_{j != i} |S_ij|).  When that bound is not positive no principal
        // stress is tensile, every rd below would be zero and R = 0: no eigen-decomposition (most particles of a body under
        // compression; an eigenvalue the reference's solver would round to +1e-17 of the stress scale is all this can miss)
        {
            const double g0 = S[0][0] + fabs(S[0][1]) + fabs(S[0][2]), g1 = S[1][1] + fabs(S[0][1]) + fabs(S[1][2]),
                         g2 = S[2][2] + fabs(S[0][2]) + fabs(S[1][2]);
            if (fmax(g0, fmax(g1, g2)) <= 0.0) {
                q[SPH_R00][i] = 0.0; q[SPH_R11][i] = 0.0; q[SPH_R22][i] = 0.0;
                q[SPH_R12][i] = 0.0; q[SPH_R02][i] = 0.0; q[SPH_R01][i] = 0.0;
                break;
            }
        }
        // same scaling as linalg3.eigen_decomposition (:520-536): tiny matrices stay accurate
        double sc = 0.0;
        for (int r = 0; r < 3; r++) for (int cc = 0; cc < 3; cc++) sc += fabs(S[r][cc]);
        if (sc == 0.0) {
            lam[0] = lam[1] = lam[2] = 0.0;
            for (int r = 0; r < 3; r++) for (int cc = 0; cc < 3; cc++) R[r][cc] = (r == cc);
        } else {
            for (int r = 0; r < 3; r++) for (int cc = 0; cc < 3; cc++) S[r][cc] /= sc;
            jacobi_eigen3(S, R, lam);
            for (int k = 0; k < 3; k++) lam[k] *= sc;
        }
        for (int k = 0; k < 3; k++) rd[k] = lam[k] > 0 ? -a.par[0] * lam[k] * rhoi21 : 0.0;
        if (a.tflag && (rd[0] != 0.0 || rd[1] != 0.0 || rd[2] != 0.0)) *a.tflag = 1u; // a particle in tension: r_ij != 0
        // transform_diag_inv: R diag(rd) R^T  (linalg3.pyx:220-234)
        double Rab[3][3];
        for (int r = 0; r < 3; r++)
            for (int cc = r; cc < 3; cc++) {
                double t = 0.0;
                for (int k = 0; k < 3; k++) t += R[r][k] * rd[k] * R[cc][k];
                Rab[r][cc] = t;
            }
        q[SPH_R00][i] = Rab[0][0]; q[SPH_R11][i] = Rab[1][1]; q[SPH_R22][i] = Rab[2][2];
        q[SPH_R12][i] = Rab[1][2]; q[SPH_R02][i] = Rab[0][2]; q[SPH_R01][i] = Rab[0][1];
        break;
    }
    case SPH_EQ_HOOKES_DEVIATORIC_STRESS_RATE: { // solid_mech/basic.py:418-505; par G
        double **q = a.q;
        const double v00 = q[SPH_V00][i], v01 = q[SPH_V01][i], v02 = q[SPH_V02][i];
        const double v10 = q[SPH_V10][i], v11 = q[SPH_V11][i], v12 = q[SPH_V12][i];
        const double v20 = q[SPH_V20][i], v21 = q[SPH_V21][i], v22 = q[SPH_V22][i];
        const double s00 = q[SPH_S00][i], s01 = q[SPH_S01][i], s02 = q[SPH_S02][i];
        const double s11 = q[SPH_S11][i], s12 = q[SPH_S12][i], s22 = q[SPH_S22][i];
        const double s10 = s01, s20 = s02, s21 = s12;
        const double eps00 = v00, eps01 = 0.5 * (v01 + v10), eps02 = 0.5 * (v02 + v20);
        const double eps11 = v11, eps12 = 0.5 * (v12 + v21), eps22 = v22;
        const double omega01 = 0.5 * (v01 - v10), omega02 = 0.5 * (v02 - v20), omega12 = 0.5 * (v12 - v21);
        const double omega10 = -omega01, omega20 = -omega02, omega21 = -omega12;
        const double tmp = 2.0 * a.par[0];
        const double trace = 1.0 / 3.0 * (eps00 + eps11 + eps22);
        q[SPH_AS00][i] = tmp * (eps00 - trace) + (s01 * omega01 + s02 * omega02) + (s10 * omega01 + s20 * omega02);
        q[SPH_AS01][i] = tmp * eps01 + (s00 * omega10 + s02 * omega12) + (s11 * omega01 + s21 * omega02);
        q[SPH_AS02][i] = tmp * eps02 + (s00 * omega20 + s01 * omega21) + (s12 * omega01 + s22 * omega02);
        q[SPH_AS11][i] = tmp * (eps11 - trace) + (s10 * omega10 + s12 * omega12) + (s01 * omega10 + s21 * omega12);
        q[SPH_AS12][i] = tmp * eps12 + (s10 * omega20 + s11 * omega21) + (s02 * omega10 + s22 * omega12);
        q[SPH_AS22][i] = tmp * (eps22 - trace) + (s20 * omega20 + s21 * omega21) + (s02 * omega20 + s12 * omega21);
        break;
    }
    }
}

// ---------------------------------------------------------------------------
// equation families
// ---------------------------------------------------------------------------
enum { F_VG2 = 1, F_VG3 = 2,                                            // velocity gradient
       F_ECONT = 1, F_ESTRESS = 2, F_EAV = 4, F_EXSPH = 8 };            // elastic rates
enum { F_CONT = 1, F_MOM = 2, F_XSPH = 4, F_TENSILE = 8,               // WCSPH
       F_SD = 1, F_TVFSD = 2,                                           // density
       F_TP = 1, F_TVISC = 2, F_TAV = 4, F_TAS = 8 };                   // TVF force

#define MAX_AUX 20
struct PackArgs {
    const uint32_t *perm;
    size_t n;
    size_t off;
    const double *x, *y, *z, *h;
    int na;                       // doubles in the aux record
    const double *src[MAX_AUX];   // nullptr -> 0.0
    int derived;                  // 1: aux[5] = p/(rho*rho) (p = aux[7], rho = aux[4]); 2: aux[10] = 1/V^2 (V = aux[8]);
                                  // 3: aux[6..11] = (s_ij - p delta_ij)/rho^2 (p = aux[18], rho = aux[4])
                                  // 4: aux[3] = m/rho (m = aux[3], rho = aux[4]; aux[4] itself is not stored)
                                  // 5: aux[0] = the particle's original index (neighbour lists)
    double4 *posh;
    double *aux;
    double *rec;                  // non-null: interleaved records [x y z h aux... pad], nr doubles each
    int nr;
    int layout;                   // 0: [x y z h | aux...]; 1: WCSPH [x y z cs | u v w m | rho tmpj | h p]; 2: density [x y z m];
                                  // 3: TVF [x y z rho | u v w p | Vj2 m uhat vhat | what -];
                                  // 4: generated, uniform h [x y z | aux...]; 5: fp32 records (floats, any family)
                                  // 6: WCSPH, p and cs recomputed by the pair kernel [x y | z u | v w | rho m] (64 B);
                                  // 7: the same in fp32 [x-x0 y-y0 z-z0 u | v w rho m] (32 B)
                                  // 8 / 9: TVF, p and V recomputed from rho (fp64 80 B / fp32 48 B)
                                  // 10 / 11: elastic rates without h and m (fp64 160 B / fp32 80 B)
                                  // 12 / 13: WCSPH with variable h, one mass per array, EOS recomputed (64 B / 32 B)
                                  // (1, 2: aggregated kernel only)
    int umass;                    // layouts 6 / 7: the last slot carries p / rho^2 (derived 1) instead of m
    float4 *fpos;                 // non-null: fp32 {x-xmin, y-ymin, z-zmin, radius_scale*h} for the prefilter tiles
    int lds_np;                   // 16-B pieces per record when the launch carries 256 * lds_np * 16 B of LDS, else 0
    double gmin[3];
    double radius_scale;
};

#define PACK_MAXP ((4 + MAX_AUX) / 2) // 16-B pieces of the widest record

// Store one record of `np` 16-B pieces per lane.  The 64 records of a wavefront are contiguous in
// the packed buffer: a whole block transposes them through (dynamic) LDS so that each store
// instruction writes 1 KiB of whole lines instead of 64 pieces a record apart (k_pack 0.29 -> 0.22 ms
// on the 4 M cube); the ragged last block, and launches without LDS, store per lane.
template <class PA>
__device__ __forceinline__ void emit_pieces(const PA &a, size_t i, const double2 (&pc)[PACK_MAXP], int np)
{
    extern __shared__ __attribute__((aligned(16))) double2 pack_lds[];
    double2 *const base = reinterpret_cast<double2 *>(a.rec);
    if (a.lds_np == np && ((size_t)blockIdx.x + 1) * 256 <= a.n) {
        const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
        double2 *const sw = pack_lds + (size_t)w * 64 * np;
#pragma unroll
        for (int q = 0; q < PACK_MAXP; q++)
            if (q < np) sw[l * np + q] = pc[q];
        // private to the wavefront: program order is execution order, the fence is for the compiler
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        double2 *const ow = base + (a.off + (size_t)blockIdx.x * 256 + (size_t)w * 64) * np;
#pragma unroll
        for (int q = 0; q < PACK_MAXP; q++)
            if (q < np) ow[q * 64 + l] = sw[q * 64 + l];
    } else {
        double2 *const r2 = base + (a.off + i) * np;
#pragma unroll
        for (int q = 0; q < PACK_MAXP; q++)
            if (q < np) r2[q] = pc[q];
    }
}

// LAYOUT >= 0: the record layout as a compile-time constant (the hot layouts of the hand-written families: the layout
// switch folds away, the piece count is a constant, and the record never passes through a dynamically indexed local array
// -- the one-kernel-for-all form kept 32 B per lane in scratch); -1: PackArgs::layout at run time (every other layout).
template <int LAYOUT>
__global__ __launch_bounds__(256) void k_pack(PackArgs a)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    uint32_t o = a.perm[i];
    const int layout = LAYOUT >= 0 ? LAYOUT : a.layout;
    double4 ph;
    ph.x = a.x[o]; ph.y = a.y[o]; ph.z = a.z[o]; ph.w = a.h[o];
    double v[MAX_AUX];
    // (compile-time layouts read the slots their record holds: WCSPH u v w m rho tmpj cs p = 0..7, elastic 0..17 + p)
    constexpr int NV = (LAYOUT == 6 || LAYOUT == 7 || LAYOUT == 12 || LAYOUT == 13) ? 8 : (LAYOUT == 2 ? 1 : (LAYOUT == 3 || LAYOUT == 8 || LAYOUT == 9) ? 11 : MAX_AUX);
#pragma unroll
    for (int k = 0; k < MAX_AUX; k++) v[k] = (k < NV && (k < a.na || (a.derived == 3 && k == 18) || (a.derived == 4 && k == 4)) && a.src[k]) ? a.src[k][o] : 0.0;
    if (a.derived == 1) v[5] = v[4] != 0.0 ? v[7] * (1.0 / (v[4] * v[4])) : 0.0; // tmpj = p*rhoj21, wc/basic.py:211,234
    if (a.derived == 2) { double Vj = 1. / v[8]; v[10] = Vj * Vj; } // Vj2, transport_velocity.py:303-306
    if (a.derived == 4) v[3] = v[3] / v[4]; // m/rho, basic_equations.py:103,139 (VelocityGradient tmp)
    if (a.derived == 5) v[0] = (double)o;
    if (a.derived == 3) { // (sigma_ij)/rho^2 with sigma = s - p I: solid_mech/basic.py:281-283,333-339,367-378
        const double r21 = 1. / (v[4] * v[4]), pr = v[18];
        v[6] = (v[6] - pr) * r21; v[7] *= r21; v[8] *= r21;
        v[9] = (v[9] - pr) * r21; v[10] *= r21; v[11] = (v[11] - pr) * r21;
        v[18] = 0.0;
    }
    if (a.fpos)
        a.fpos[a.off + i] = make_float4((float)(ph.x - a.gmin[0]), (float)(ph.y - a.gmin[1]), (float)(ph.z - a.gmin[2]),
                                        (float)(a.radius_scale * ph.w));
    if (a.rec) {
        // the record as 16-B pieces, then ONE way out (emit_pieces): the records of a wavefront
        // are contiguous, so they leave transposed through LDS as whole lines
        double2 pc[PACK_MAXP];
#pragma unroll
        for (int q = 0; q < PACK_MAXP; q++) pc[q] = make_double2(0.0, 0.0);
        int np;
        if (layout == 6) { // WCSPH with the EOS fused into the pair kernel: one 64-B half line per record
            pc[0] = make_double2(ph.x, ph.y); pc[1] = make_double2(ph.z, v[0]);
            pc[2] = make_double2(v[1], v[2]); pc[3] = make_double2(v[4], a.umass ? v[5] : v[3]); // [rho m], uniform mass: [rho p/rho^2]
            np = 4;
        } else if (layout == 7) {
            pc[0] = __builtin_bit_cast(double2, make_float4((float)(ph.x - a.gmin[0]), (float)(ph.y - a.gmin[1]),
                                                            (float)(ph.z - a.gmin[2]), (float)v[0]));
            pc[1] = __builtin_bit_cast(double2, make_float4((float)v[1], (float)v[2], (float)v[4], (float)(a.umass ? v[5] : v[3])));
            np = 2;
        } else if (layout == 8) { // TVF with the state equation fused: [x y | z rho | u v | w uhat | vhat what] (no artificial stress: 4 pieces)
            pc[0] = make_double2(ph.x, ph.y); pc[1] = make_double2(ph.z, v[6]);
            pc[2] = make_double2(v[0], v[1]); pc[3] = make_double2(v[2], v[3]);
            pc[4] = make_double2(v[4], v[5]);
            np = a.nr / 2;
        } else if (layout == 9) { // the same in fp32: [x-x0 y-y0 z-z0 rho | u v w uhat | vhat what - -]
            pc[0] = __builtin_bit_cast(double2, make_float4((float)(ph.x - a.gmin[0]), (float)(ph.y - a.gmin[1]),
                                                            (float)(ph.z - a.gmin[2]), (float)v[6]));
            pc[1] = __builtin_bit_cast(double2, make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]));
            pc[2] = __builtin_bit_cast(double2, make_float4((float)v[4], (float)v[5], 0.f, 0.f));
            np = a.nr / 4;
        } else if (layout == 12) { // WCSPH, variable h, one mass, EOS recomputed: [x y | z h | u v | w rho]
            pc[0] = make_double2(ph.x, ph.y); pc[1] = make_double2(ph.z, ph.w);
            pc[2] = make_double2(v[0], v[1]); pc[3] = make_double2(v[2], v[4]);
            np = 4;
        } else if (layout == 13) { // the same in fp32: [x y z h | u v w rho]
            pc[0] = __builtin_bit_cast(double2, make_float4((float)(ph.x - a.gmin[0]), (float)(ph.y - a.gmin[1]),
                                                            (float)(ph.z - a.gmin[2]), (float)ph.w));
            pc[1] = __builtin_bit_cast(double2, make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[4]));
            np = 2;
        } else if (layout == 10) { // elastic, uniform h and mass: [x y | z u | v w | rho cs | t x6 | r x6]
            pc[0] = make_double2(ph.x, ph.y); pc[1] = make_double2(ph.z, v[0]);
            pc[2] = make_double2(v[1], v[2]); pc[3] = make_double2(v[4], v[5]);
#pragma unroll
            for (int q = 0; q < 6; q++) pc[4 + q] = make_double2(v[6 + 2 * q], v[7 + 2 * q]);
            np = 10;
        } else if (layout == 11) { // the same in fp32: [x y z u | v w rho cs | t t t t | t t r r | r r r r]
            pc[0] = __builtin_bit_cast(double2, make_float4((float)(ph.x - a.gmin[0]), (float)(ph.y - a.gmin[1]),
                                                            (float)(ph.z - a.gmin[2]), (float)v[0]));
            pc[1] = __builtin_bit_cast(double2, make_float4((float)v[1], (float)v[2], (float)v[4], (float)v[5]));
#pragma unroll
            for (int q = 0; q < 3; q++)
                pc[2 + q] = __builtin_bit_cast(double2, make_float4((float)v[6 + 4 * q], (float)v[7 + 4 * q], (float)v[8 + 4 * q], (float)v[9 + 4 * q]));
            np = 5;
        } else if (layout == 1) { // WCSPH [x y | z cs | u v | w m | rho tmpj] (+ [h p]: variable h / tensile correction)
            pc[0] = make_double2(ph.x, ph.y); pc[1] = make_double2(ph.z, v[6]);
            pc[2] = make_double2(v[0], v[1]); pc[3] = make_double2(v[2], v[3]);
            pc[4] = make_double2(v[4], v[5]); pc[5] = make_double2(ph.w, v[7]);
            np = a.nr / 2;
        } else if (layout == 5) { // fp32 records [x-x0 y-y0 z-z0 h | aux...] (option record_f32), a.nr floats
            float w[4 + MAX_AUX];
            w[0] = (float)(ph.x - a.gmin[0]); w[1] = (float)(ph.y - a.gmin[1]); w[2] = (float)(ph.z - a.gmin[2]);
            w[3] = (float)ph.w;
#pragma unroll
            for (int k = 0; k < MAX_AUX; k++) w[4 + k] = k < a.na ? (float)v[k] : 0.f;
#pragma unroll
            for (int q = 0; q < (4 + MAX_AUX) / 4; q++)
                pc[q] = __builtin_bit_cast(double2, make_float4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]));
            np = a.nr / 4;
        } else if (layout == 4) { // generated families under uniform h: [x y z | aux...] (h is a launch constant)
            double w[3 + MAX_AUX + 1];
            w[0] = ph.x; w[1] = ph.y; w[2] = ph.z;
#pragma unroll
            for (int k = 0; k < MAX_AUX + 1; k++) w[3 + k] = k < a.na ? v[k < MAX_AUX ? k : 0] : 0.0;
#pragma unroll
            for (int q = 0; q < (3 + MAX_AUX + 1) / 2; q++) pc[q] = make_double2(w[2 * q], w[2 * q + 1]);
            np = (a.nr + 1) / 2;
        } else if (layout == 3) { // TVF under uniform h: [x y | z rho | u v | w p | Vj2 what | uhat vhat] (+ [m -] with artificial viscosity)
            pc[0] = make_double2(ph.x, ph.y); pc[1] = make_double2(ph.z, v[6]);
            pc[2] = make_double2(v[0], v[1]); pc[3] = make_double2(v[2], v[7]);
            pc[4] = make_double2(v[10], v[5]); pc[5] = make_double2(v[3], v[4]);
            pc[6] = make_double2(v[9], 0.0);
            np = a.nr > 12 ? 7 : 6;
        } else if (layout == 2) { // compact density records [x y z m] (uniform h)
            pc[0] = make_double2(ph.x, ph.y); pc[1] = make_double2(ph.z, v[0]);
            np = 2;
        } else { // [x y z h | aux...]
            pc[0] = make_double2(ph.x, ph.y); pc[1] = make_double2(ph.z, ph.w);
#pragma unroll
            for (int q = 0; q < MAX_AUX / 2; q++) pc[2 + q] = make_double2(v[2 * q], v[2 * q + 1]);
            np = a.nr / 2;
        }
        emit_pieces(a, i, pc, np);
    } else {
        a.posh[a.off + i] = ph;
        double *dst = a.aux + (a.off + i) * (size_t)a.na;
#pragma unroll
        for (int k = 0; k < MAX_AUX; k++) if (k < a.na) dst[k] = v[k];
    }
}

// Records of the MERGED order (FamWCSPHM_T): thread p packs the particle at merged position p -- array slot[p], original
// index perm[p] -- from ITS array's properties (pointer tables read per lane from the kernarg segment) into
// [x y | z u | v w | +-rho p/rho^2] (fp32: [x-x0 y-y0 z-z0 u | v w +-rho p/rho^2]); the sign of rho is the array's class.
// One launch for all arrays; consecutive merged positions are consecutive records, so whole blocks leave through LDS
// as full lines like k_pack's.
struct PackMArgs {
    const uint32_t *perm;
    const uint8_t *slot;
    size_t n, off;                 // off = 0 (emit_pieces)
    const double *prop[9][SPH_MAX_ARRAYS]; // x y z u v w rho p h, per slot (h: variable-h records only)
    int vh;                        // 1: variable h -- records [x y | z h | u v | w +-rho], p / rho^2 recomputed by the pair kernel
    uint32_t cls;                  // bit s: slot s is a class-1 array
    double *rec;
    float4 *fpos;
    int f32, lds_np;
    double gmin[3];
    double hr;                     // radius_scale * h (uniform h)
    uint32_t *rho_flag;            // device word: set to 1 when a density is not positive (the class bit is the SIGN of rho)
};

// (F32, VH: PackMArgs::f32 / vh as compile-time constants -- the piece count of the record is then one too)
template <bool F32, bool VH>
__global__ __launch_bounds__(256) void k_pack_merged(PackMArgs a)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const uint32_t o = a.perm[i], sl = a.slot[i];
    double v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const double *p = kernarg_read<const double *>(__builtin_offsetof(PackMArgs, prop) + ((size_t)k * SPH_MAX_ARRAYS + sl) * sizeof(double *));
        v[k] = p[o];
    }
    const double rho = v[6];
    const double q = rho != 0.0 ? v[7] * (1.0 / (rho * rho)) : 0.0; // tmpj = p*rhoj21, wc/basic.py:211,234 (as k_pack, derived 1)
    const double srho = ((a.cls >> sl) & 1u) ? -rho : rho;
    // rho <= 0 (or NaN): this record's class bit is not to be trusted (padding rows of sph_halo_append_padded, parked far
    // outside the domain with rho = 0, are nobody's neighbour)
    if (!(rho > 0.0) && fabs(v[0]) < SPH_PARKED_MIN) atomicOr(a.rho_flag, 1u);
    double hp = 0.0;
    if (VH) hp = kernarg_read<const double *>(__builtin_offsetof(PackMArgs, prop) + ((size_t)8 * SPH_MAX_ARRAYS + sl) * sizeof(double *))[o];
    a.fpos[i] = make_float4((float)(v[0] - a.gmin[0]), (float)(v[1] - a.gmin[1]), (float)(v[2] - a.gmin[2]),
                            VH ? (float)(a.hr * hp) : (float)a.hr); // hr: radius_scale * h, or radius_scale alone with variable h
    double2 pc[PACK_MAXP];
#pragma unroll
    for (int k = 0; k < PACK_MAXP; k++) pc[k] = make_double2(0.0, 0.0);
    int np;
    if (VH && F32) {
        pc[0] = __builtin_bit_cast(double2, make_float4((float)(v[0] - a.gmin[0]), (float)(v[1] - a.gmin[1]), (float)(v[2] - a.gmin[2]), (float)hp));
        pc[1] = __builtin_bit_cast(double2, make_float4((float)v[3], (float)v[4], (float)v[5], (float)srho));
        np = 2;
    } else if (VH) {
        pc[0] = make_double2(v[0], v[1]); pc[1] = make_double2(v[2], hp);
        pc[2] = make_double2(v[3], v[4]); pc[3] = make_double2(v[5], srho);
        np = 4;
    } else if (F32) {
        pc[0] = __builtin_bit_cast(double2, make_float4((float)(v[0] - a.gmin[0]), (float)(v[1] - a.gmin[1]), (float)(v[2] - a.gmin[2]), (float)v[3]));
        pc[1] = __builtin_bit_cast(double2, make_float4((float)v[4], (float)v[5], (float)srho, (float)q));
        np = 2;
    } else {
        pc[0] = make_double2(v[0], v[1]); pc[1] = make_double2(v[2], v[3]);
        pc[2] = make_double2(v[4], v[5]); pc[3] = make_double2(srho, q);
        np = 4;
    }
    emit_pieces(a, i, pc, np);
}

template <class T> struct FamWCSPH_T {
    typedef T Real; // arithmetic type of the pair loop
    static constexpr uint32_t CF0 = F_CONT | F_MOM | F_XSPH; // flag set compiled as a constant (variant 6)
#ifndef SPH_WCSPH_MINB
#define SPH_WCSPH_MINB 4
#endif
    static constexpr int MINB = SPH_WCSPH_MINB; // wavefronts per SIMD the pair kernel is compiled for (VGPR budget)
    static constexpr int NA = 8; // u v w m rho tmpj(=p/rho^2) cs p
    static constexpr int NR = 12; // x y z h + NA (128-B padded records measured slower: larger L2 footprint)
    struct Params {
        T c0, alpha, beta, gx, gy, gz, eps;
        double *arho, *au, *av, *aw, *ax, *ay, *az, *dt_cfl, *dt_force;
    };
    struct Dest {
        T u, v, w, rho, p, cs, tmpi;
        T arho, au, av, aw, ax, ay, az, dt_cfl;
    };
    template <class A> static __device__ __forceinline__ void load(Dest &D, const T *a, const A &, uint32_t)
    {
        D.u = a[0]; D.v = a[1]; D.w = a[2]; D.rho = a[4]; D.tmpi = a[5]; D.cs = a[6]; D.p = a[7];
        D.arho = D.au = D.av = D.aw = D.ax = D.ay = D.az = T(0.0);
        D.dt_cfl = T(-1.0); // running max of |h v.x / r2| over the pairs; -1: none yet (finish)
    }
    // The algebra below is the reference's (cited per term) regrouped so that a
    // pair costs one rsqrt and one reciprocal:  DWIJ = tg*XIJ  =>
    //   DWIJ.VIJ = tg*(XIJ.VIJ),   f*DWIJ = (f*tg)*XIJ,
    //   1/(R2IJ+EPS) and 1/RHOIJ from one reciprocal of their product,
    //   1/R2IJ = rinv^2.
    // PRED: the lean kernel calls pair() for every lane of every iteration and
    // passes the neighbour criterion as `pass`; a pair that fails contributes
    // exactly zero because every term carries the factor m_j (set to 0) -- no
    // branch around the accumulators, which then stay in their registers.
    static constexpr bool PRED = true;
    template <int KK, bool UH, class A>
    static __device__ __forceinline__ void pair(Dest &D, const real4<T> &pi, const real4<T> &pj, T r2,
                                                const T (&s)[NA], uint32_t fl, const A &a, bool pass = true)
    {
        PairGeomT<T> g;
        // (one transcendental per pair where the momentum equation acts: pair_geom_mom)
        constexpr bool MRG = SPH_MERGED_RSQ && sizeof(T) == 8 && KK != 4;
        T tt_m = T(0.0);
        if constexpr (MRG) {
            if (fl & F_MOM) tt_m = pair_geom_mom<KK, UH>(g, pi, pj, r2, T(0.5) * (D.rho + s[4]), a);
            else pair_geom<KK, UH>(g, pi, pj, r2, a);
        } else {
            pair_geom<KK, UH>(g, pi, pj, r2, a);
        }
        const T tg = pair_gradfac<KK, UH>(g);
        const T vij0 = D.u - s[0], vij1 = D.v - s[1], vij2 = D.w - s[2]; // VIJ equation.py:214-223
        const T vdotx = vij0 * g.xij[0] + vij1 * g.xij[1] + vij2 * g.xij[2];
        const T mj = pass ? s[3] : T(0.0);
        if (fl & F_CONT) D.arho = fma(mj * tg, vdotx, D.arho); // basic_equations.py:187-192
        if (fl & (F_MOM | F_XSPH)) {
            const T rhoij = T(0.5) * (D.rho + s[4]); // RHOIJ equation.py:196
            T rhoij1;                              // RHOIJ1 :199
            T wij = T(0.0);
            if (fl & (F_XSPH | F_TENSILE)) wij = pair_w<KK, UH>(g);
            if (fl & F_MOM) { // wc/basic.py:204-259
                const T re = r2 + g.eps;
                const T tt = MRG ? tt_m : fast_rcp(re * rhoij);
                const T inv_re = rhoij * tt;
                rhoij1 = re * tt;
                const T hv = g.hij * vdotx;
                // the viscosity acts for approaching pairs only (vdotx < 0; h > 0): muij from min(hv, 0) is exactly
                // zero otherwise and takes piij with it
                const T muij = raw_min(hv, T(0.0)) * inv_re;
                const T cij = T(0.5) * (D.cs + s[6]);
                const T piij = (a.p.beta * muij - a.p.alpha * cij) * muij * rhoij1;
                // dt_cfl = max_j (|h v.x / r2| + c0) = max_j |h v.x / r2| + c0 (rounding is monotonic): the running
                // maximum starts at -1 (no pair yet), finish() adds c0 once
                const T dtv = fabs(hv * (g.rinv * g.rinv));
                D.dt_cfl = raw_max(D.dt_cfl, (r2 > T(1e-12) && pass) ? dtv : T(-1.0));
                const T tmpj = s[5];
                T tmp = D.tmpi + tmpj;
                if (fl & F_TENSILE) {
                    // WDP = KERNEL(XIJ, DELTAP*HIJ, HIJ)  equation.py:243-246
                    const T qd = (a.k.deltap * g.hij) * g.h1;
                    const T wdp = SphKernel<KK>::w(qd) * g.fac;
                    T fij = wij * fast_rcp(wdp);
                    fij = fij * fij;
                    fij = fij * fij;
                    const T Ri = D.p > 0 ? T(0.01) * D.tmpi : T(0.2) * fabs(D.tmpi);
                    const T Rj = s[7] > 0 ? T(0.01) * tmpj : T(0.2) * fabs(tmpj);
                    tmp = (D.tmpi + tmpj) + (Ri + Rj) * fij;
                }
                const T ft = -mj * (tmp + piij) * tg;
                D.au = fma(ft, g.xij[0], D.au);
                D.av = fma(ft, g.xij[1], D.av);
                D.aw = fma(ft, g.xij[2], D.aw);
            } else {
                rhoij1 = fast_rcp(rhoij);
            }
            if (fl & F_XSPH) { // basic_equations.py:290-295
                const T tmp = -a.p.eps * mj * wij * rhoij1;
                D.ax = fma(tmp, vij0, D.ax);
                D.ay = fma(tmp, vij1, D.ay);
                D.az = fma(tmp, vij2, D.az);
            }
        }
    }
    template <class A> static __device__ __forceinline__ void finish(Dest &D, const A &a, uint32_t o)
    {
        if (a.dflags & F_CONT) a.p.arho[o] = D.arho;
        if (a.dflags & F_MOM) { // post_loop wc/basic.py:261-271
            T au = D.au + a.p.gx, av = D.av + a.p.gy, aw = D.aw + a.p.gz;
            a.p.au[o] = au; a.p.av[o] = av; a.p.aw[o] = aw;
            a.p.dt_cfl[o] = D.dt_cfl < T(0.0) ? T(0.0) : D.dt_cfl + a.p.c0;
            a.p.dt_force[o] = au * au + av * av + aw * aw;
        }
        if (a.dflags & F_XSPH) { // post_loop basic_equations.py:297-300
            a.p.ax[o] = D.ax + D.u; a.p.ay[o] = D.ay + D.v; a.p.az[o] = D.az + D.w;
        }
    }
};
typedef FamWCSPH_T<double> FamWCSPH;

// The same equations on 64-byte records [x y | z u | v w | rho m] (fp32: 32 bytes): when the group before
// this one was the Tait EOS over every array read here (sph_group.src_eos), p and cs are functions of rho
// and are recomputed per gathered record -- four 16-B pieces in ONE 64-B half line instead of five pieces
// that straddle a 128-B line half of the time.  ~14 more VALU operations per pair, hidden under the gathers.
// Arithmetic as k_nosrc's (TaitEOS, gamma = 7) and k_pack's p / rho^2.
// UM (uniform-mass records): every particle of a source array has the same mass (seen by the last sph_nnps_update),
// so the record's eighth slot carries p / rho^2 as k_pack computes it for the 80-byte records and m is a constant of
// the source: of the EOS only cs = c0 (rho/rho0)^3 is left in the loop (13 of its 17 instructions go).
template <class T, bool UM = false> struct FamWCSPHE_T : FamWCSPH_T<T> {
    static constexpr bool EOSF = true;
    static constexpr bool UMASS = UM;
    static constexpr int NR = 8;
    // one gathered record as it arrives (four / two 16-B pieces), and its decoding
    struct Raw {
        typename std::conditional<sizeof(T) == 8, double2, float4>::type q[sizeof(T) == 8 ? 4 : 2];
    };
    template <class A> static __device__ __forceinline__ void load_raw(const A &a, uint32_t jg, Raw &r)
    {
        if constexpr (sizeof(T) == 8) {
            const double2 *p = reinterpret_cast<const double2 *>(a.rec) + (unsigned long long)jg * 4;
            r.q[0] = p[0]; r.q[1] = p[1]; r.q[2] = p[2]; r.q[3] = p[3];
        } else {
            const float4 *p = reinterpret_cast<const float4 *>(a.rec) + (unsigned long long)jg * 2;
            r.q[0] = p[0]; r.q[1] = p[1];
        }
    }
    // gk: the exponent gamma = 2 gk + 1 -- the constant 3 here (water: the compiler folds the powers), the launch
    // argument in FamWCSPHEG_T below
    template <class A> static __device__ __forceinline__ void decode(const A &a, const Raw &r, uint32_t fl, T mu, real4<T> &pj, T (&s)[8],
                                                                     int gk = 3)
    {
        T rho;
        if constexpr (sizeof(T) == 8) {
            pj.x = r.q[0].x; pj.y = r.q[0].y; pj.z = r.q[1].x; pj.w = 0.0;
            s[0] = r.q[1].y; s[1] = r.q[2].x; s[2] = r.q[2].y; s[3] = r.q[3].y; rho = r.q[3].x;
        } else {
            pj.x = r.q[0].x; pj.y = r.q[0].y; pj.z = r.q[0].z; pj.w = 0.f;
            s[0] = r.q[0].w; s[1] = r.q[1].x; s[2] = r.q[1].y; s[3] = r.q[1].w; rho = r.q[1].z;
        }
        s[4] = rho;
        s[5] = s[6] = s[7] = T(0.0);
        if constexpr (UM) {
            const T q = s[3]; // p / rho^2
            s[3] = mu;
            if (fl & F_MOM) {
                const T ratio = rho * (T)a.e_rho01;
                s[5] = q;
                s[6] = (T)a.e_c0 * tait_cs_power(ratio, gk);
            }
        } else
        if (fl & F_MOM) { // the flags of this (destination, source): a compile-time constant in the common case; a
                          // continuity-only destination -- a dam break's walls -- reads neither
            const T ratio = rho * (T)a.e_rho01;
            T r7, r3; // (ratio^gamma, ratio^((gamma - 1) / 2): named for the usual exponent)
            tait_powers(ratio, gk, r7, r3);
            const T p = (T)a.e_p0 + (T)a.e_B * (r7 - T(1.0));
            s[5] = rho != T(0.0) ? p * fast_rcp(rho * rho) : T(0.0);
            s[6] = (T)a.e_c0 * r3;
            s[7] = p;
        }
    }
    template <class A> static __device__ __forceinline__ void load_fused(const A &a, uint32_t jg, uint32_t fl, T mu, real4<T> &pj, T (&s)[8])
    {
        Raw r;
        load_raw(a, jg, r);
        decode(a, r, fl, mu, pj, s);
    }
};

// ... with the exponent as a launch argument (gamma = 1, 3, 5: PairArgs::e_gk).  A family of its own because the select
// costs the water kernels 1.3-2 % of their time when they carry it (profiles/r06_ab_tables.txt).
template <class T, bool UM = false> struct FamWCSPHEG_T : FamWCSPHE_T<T, UM> {
    typedef FamWCSPHE_T<T, UM> Base;
    template <class A> static __device__ __forceinline__ void load_fused(const A &a, uint32_t jg, uint32_t fl, T mu, real4<T> &pj, T (&s)[8])
    {
        typename Base::Raw r;
        Base::load_raw(a, jg, r);
        Base::decode(a, r, fl, mu, pj, s, a.e_gk);
    }
};

// Variable h with ONE mass per source array: [x y | z h | u v | w rho] (fp32: [x y z h | u v w rho]) -- 64 bytes instead of
// the 96-byte records with cs, m, p / rho^2 and p: the mass is a constant of the source, p, cs and p / rho^2 are the
// Tait EOS of the gathered rho (as FamWCSPHE_T without the uniform-mass slot).  Run with UH = false.
template <class T> struct FamWCSPHV_T : FamWCSPH_T<T> {
    static constexpr bool EOSF = true;
    static constexpr int NR = 8;
    template <class A> static __device__ __forceinline__ void load_fused(const A &a, uint32_t jg, uint32_t fl, T mu, real4<T> &pj, T (&s)[8])
    {
        T rho;
        if constexpr (sizeof(T) == 8) {
            const double2 *p = reinterpret_cast<const double2 *>(a.rec) + (unsigned long long)jg * 4;
            const double2 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
            pj.x = q0.x; pj.y = q0.y; pj.z = q1.x; pj.w = q1.y;
            s[0] = q2.x; s[1] = q2.y; s[2] = q3.x; rho = q3.y;
        } else {
            const float4 *p = reinterpret_cast<const float4 *>(a.rec) + (unsigned long long)jg * 2;
            const float4 q0 = p[0], q1 = p[1];
            pj.x = q0.x; pj.y = q0.y; pj.z = q0.z; pj.w = q0.w;
            s[0] = q1.x; s[1] = q1.y; s[2] = q1.z; rho = q1.w;
        }
        s[3] = mu;
        s[4] = rho;
        s[5] = s[6] = s[7] = T(0.0);
        if (fl & F_MOM) {
            const T ratio = rho * (T)a.e_rho01;
            T r7, r3; // (ratio^gamma, ratio^((gamma - 1) / 2): named for the usual exponent)
            tait_powers(ratio, 3, r7, r3); // (gamma = 7 only: sph_eval_group checks)
            const T p = (T)a.e_p0 + (T)a.e_B * (r7 - T(1.0));
            s[5] = rho != T(0.0) ? p * fast_rcp(rho * rho) : T(0.0);
            s[6] = (T)a.e_c0 * r3;
            s[7] = p;
        }
    }
};

// ---- the same equations over the MERGED order of all arrays of the grid (sph_ctx::merged): ONE record stream, one
// phase 1 and one phase 2 per wavefront whatever the number of arrays, destinations of every array in one launch.
// A dam break -- fluid <- {fluid, boundary, obstacle}, walls <- fluid -- is two CLASSES of arrays: which equations act for
// a (destination, source) pair depends on the classes of the two particles only.  Records are the uniform-mass ones,
// [x y | z u | v w | +-rho p/rho^2]: the SIGN of rho is the particle's class (rho > 0 always), the mass a constant of
// the class.  The class table is the kernel's compile-time flag constant CF = f00 | f01 << 4 | f10 << 8 | f11 << 12
// (f<destination class><source class>: F_CONT | F_MOM | F_XSPH bits), so every "does equation e act for this pair" is
// a boolean expression of two sign bits, and the criterion stays a factor: a pair without equations adds exactly zero.
// Reference semantics: per-source equation lists of pysph/sph/acceleration_eval.py:126-151 (here per class pair),
// equations wc/basic.py:198-269, basic_equations.py:187-192,285-300.  Sums run over all arrays in cell order instead of
// source by source (another association of the same terms).
#define WCSPHM_CT(f00, f01, f10, f11) ((uint32_t)(f00) | ((uint32_t)(f01) << 4) | ((uint32_t)(f10) << 8) | ((uint32_t)(f11) << 12))
template <class T> struct FamWCSPHM_T : FamWCSPHE_T<T, true> {
    typedef FamWCSPHE_T<T, true> Base;
    typedef typename Base::Raw Raw;
    static constexpr bool MERGED = true;
    // WCSPHScheme(fluids, solids): fluid <- fluid all three, fluid <- solid no XSPH, solid <- fluid continuity only
    static constexpr uint32_t CF0 = WCSPHM_CT(F_CONT | F_MOM | F_XSPH, F_CONT | F_MOM, F_CONT, 0);
    struct Params {
        T c0, alpha, beta, gx, gy, gz, eps;
        double mu1;                                // mass of the class-1 arrays (class 0: SrcDesc::mu)
        uint32_t rng[SPH_MAX_ARRAYS][2];           // per slot: destination index range [start, stop) (empty: no destination)
        double *out[SPH_MAX_ARRAYS][9];            // per slot: arho au av aw ax ay az dt_cfl dt_force
    };
    static __device__ __forceinline__ size_t rng_offset(uint32_t slot) { return __builtin_offsetof(Params, rng) + (size_t)slot * 8; }
    struct Dest : Base::Dest { bool dc; };
    template <class A> static __device__ __forceinline__ void load(Dest &D, const T *a, const A &A_, uint32_t o, uint32_t)
    {
        Base::load(D, a, A_, o);
        D.dc = a[3] < T(0.0); // decode() left the class in the sign of the mass
    }
    template <class A> static __device__ __forceinline__ void decode(const A &a, const Raw &r, uint32_t, T mu, real4<T> &pj, T (&s)[8])
    {
        T rho, q;
        if constexpr (sizeof(T) == 8) {
            pj.x = r.q[0].x; pj.y = r.q[0].y; pj.z = r.q[1].x; pj.w = 0.0;
            s[0] = r.q[1].y; s[1] = r.q[2].x; s[2] = r.q[2].y; q = r.q[3].y; rho = r.q[3].x;
        } else {
            pj.x = r.q[0].x; pj.y = r.q[0].y; pj.z = r.q[0].z; pj.w = 0.f;
            s[0] = r.q[0].w; s[1] = r.q[1].x; s[2] = r.q[1].y; q = r.q[1].w; rho = r.q[1].z;
        }
        const bool c1 = rho < T(0.0);
        s[3] = c1 ? -(T)a.p.mu1 : mu;          // +-m: the sign is the class
        rho = fabs(rho);
        s[4] = rho;
        const T ratio = rho * (T)a.e_rho01;
        s[5] = q;                               // p / rho^2 as k_pack_merged computed it from the stored p
        s[6] = (T)a.e_c0 * tait_cs_power(ratio, 3); // (gamma = 7 only: sph_eval_group checks)
        s[7] = T(0.0);
    }
    template <class A> static __device__ __forceinline__ void load_fused(const A &a, uint32_t jg, uint32_t fl, T mu, real4<T> &pj, T (&s)[8])
    {
        Raw r;
        Base::load_raw(a, jg, r);
        decode(a, r, fl, mu, pj, s);
    }
    // does equation `bit` act for (destination class dc, source class sc)?  ct is a compile-time constant
    static __device__ __forceinline__ bool acts(uint32_t ct, uint32_t bit, bool dc, bool sc)
    {
        const bool c00 = ct & bit, c01 = (ct >> 4) & bit, c10 = (ct >> 8) & bit, c11 = (ct >> 12) & bit;
        return dc ? (sc ? c11 : c10) : (sc ? c01 : c00);
    }
    template <int KK, bool UH, class A>
    static __device__ __forceinline__ void pair(Dest &D, const real4<T> &pi, const real4<T> &pj, T r2,
                                                const T (&s)[8], uint32_t ct, const A &a, bool pass = true)
    {
        PairGeomT<T> g;
        constexpr bool MRG = SPH_MERGED_RSQ && sizeof(T) == 8 && KK != 4;
        T tt_m = T(0.0);
        if constexpr (MRG) tt_m = pair_geom_mom<KK, UH>(g, pi, pj, r2, T(0.5) * (D.rho + s[4]), a);
        else pair_geom<KK, UH>(g, pi, pj, r2, a);
        const T tg = pair_gradfac<KK, UH>(g);
        const T vij0 = D.u - s[0], vij1 = D.v - s[1], vij2 = D.w - s[2];
        const T vdotx = vij0 * g.xij[0] + vij1 * g.xij[1] + vij2 * g.xij[2];
        const bool sc = s[3] < T(0.0);
        const T m = fabs(s[3]);
        const bool on_c = pass && acts(ct, F_CONT, D.dc, sc), on_m = pass && acts(ct, F_MOM, D.dc, sc),
                   on_x = pass && acts(ct, F_XSPH, D.dc, sc);
        D.arho = fma((on_c ? m : T(0.0)) * tg, vdotx, D.arho); // basic_equations.py:187-192
        const T rhoij = T(0.5) * (D.rho + s[4]);
        const T wij = pair_w<KK, UH>(g);
        // wc/basic.py:204-259
        const T re = r2 + g.eps;
        const T tt = MRG ? tt_m : fast_rcp(re * rhoij);
        const T inv_re = rhoij * tt;
        const T rhoij1 = re * tt;
        const T hv = g.hij * vdotx;
        const T muij = raw_min(hv, T(0.0)) * inv_re; // approaching pairs only, as FamWCSPH_T::pair
        const T cij = T(0.5) * (D.cs + s[6]);
        const T piij = (a.p.beta * muij - a.p.alpha * cij) * muij * rhoij1;
        const T dtv = fabs(hv * (g.rinv * g.rinv));
        D.dt_cfl = raw_max(D.dt_cfl, (r2 > T(1e-12) && on_m) ? dtv : T(-1.0));
        const T ft = -(on_m ? m : T(0.0)) * ((D.tmpi + s[5]) + piij) * tg;
        D.au = fma(ft, g.xij[0], D.au);
        D.av = fma(ft, g.xij[1], D.av);
        D.aw = fma(ft, g.xij[2], D.aw);
        // basic_equations.py:290-295
        const T tmp = -a.p.eps * (on_x ? m : T(0.0)) * wij * rhoij1;
        D.ax = fma(tmp, vij0, D.ax);
        D.ay = fma(tmp, vij1, D.ay);
        D.az = fma(tmp, vij2, D.az);
    }
    template <class A> static __device__ __forceinline__ void finish(Dest &D, const A &a, uint32_t o, uint32_t slot)
    {
        // the equations with this lane's class as destination (class table row); its array's output pointers
        const uint32_t ct = a.dflags;
        const uint32_t row = D.dc ? ((ct >> 8) | (ct >> 12)) & 15u : (ct | (ct >> 4)) & 15u;
        const size_t base = __builtin_offsetof(A, p) + __builtin_offsetof(Params, out) + (size_t)slot * (9 * sizeof(double *));
        auto out = [&](int k) { return kernarg_read<double *>(base + (size_t)k * sizeof(double *)); };
        if (row & F_CONT) out(0)[o] = D.arho;
        if (row & F_MOM) { // post_loop wc/basic.py:261-271
            const T au = D.au + a.p.gx, av = D.av + a.p.gy, aw = D.aw + a.p.gz;
            out(1)[o] = au; out(2)[o] = av; out(3)[o] = aw;
            out(7)[o] = D.dt_cfl < T(0.0) ? T(0.0) : D.dt_cfl + a.p.c0;
            out(8)[o] = au * au + av * av + aw * aw;
        }
        if (row & F_XSPH) { // post_loop basic_equations.py:297-300
            out(4)[o] = D.ax + D.u; out(5)[o] = D.ay + D.v; out(6)[o] = D.az + D.w;
        }
    }
};

// ... and with VARIABLE h (round 5): records [x y | z h | u v | w +-rho] (fp32 [x y z h | u v w +-rho]) as FamWCSPHV_T's
// with the class in the sign of rho; p / rho^2 and cs are the Tait EOS of the gathered rho.  Run with UH = false.
template <class T> struct FamWCSPHMV_T : FamWCSPHM_T<T> {
    typedef FamWCSPHM_T<T> Base;
    template <class A> static __device__ __forceinline__ void load_fused(const A &a, uint32_t jg, uint32_t, T mu, real4<T> &pj, T (&s)[8])
    {
        T rho;
        if constexpr (sizeof(T) == 8) {
            const double2 *p = reinterpret_cast<const double2 *>(a.rec) + (unsigned long long)jg * 4;
            const double2 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
            pj.x = q0.x; pj.y = q0.y; pj.z = q1.x; pj.w = q1.y;
            s[0] = q2.x; s[1] = q2.y; s[2] = q3.x; rho = q3.y;
        } else {
            const float4 *p = reinterpret_cast<const float4 *>(a.rec) + (unsigned long long)jg * 2;
            const float4 q0 = p[0], q1 = p[1];
            pj.x = q0.x; pj.y = q0.y; pj.z = q0.z; pj.w = q0.w;
            s[0] = q1.x; s[1] = q1.y; s[2] = q1.z; rho = q1.w;
        }
        const bool c1 = rho < T(0.0);
        s[3] = c1 ? -(T)a.p.mu1 : mu; // +-m: the sign is the class
        rho = fabs(rho);
        s[4] = rho;
        const T ratio = rho * (T)a.e_rho01;
        T r7, r3;
        tait_powers(ratio, 3, r7, r3); // (gamma = 7 only: sph_eval_group checks)
        const T p = (T)a.e_p0 + (T)a.e_B * (r7 - T(1.0));
        s[5] = rho != T(0.0) ? p * fast_rcp(rho * rho) : T(0.0);
        s[6] = (T)a.e_c0 * r3;
        s[7] = T(0.0);
    }
};

// WCSPH records of the aggregated kernel use the layout
//   [x y z cs | u v w m | rho tmpj | h p]
// so that the common case (uniform h, no tensile correction) gathers 80 B =
// five 16-B pieces per pair and never touches the last piece.
template <bool UH>
__device__ __forceinline__ void load_record_wcsph(const double *__restrict__ rj, uint32_t fl, double4 &pj, double (&s)[8])
{
    const double2 *r2 = reinterpret_cast<const double2 *>(rj);
    const double2 a0 = r2[0], a1 = r2[1], b0 = r2[2], b1 = r2[3], c = r2[4];
    pj.x = a0.x; pj.y = a0.y; pj.z = a1.x; pj.w = 0.0;
    s[0] = b0.x; s[1] = b0.y; s[2] = b1.x; s[3] = b1.y; s[4] = c.x; s[5] = c.y; s[6] = a1.y; s[7] = 0.0;
    if (!UH || (fl & F_TENSILE)) {
        const double2 d = *reinterpret_cast<const double2 *>(rj + 10);
        pj.w = d.x; s[7] = d.y;
    }
}
template <> __device__ __forceinline__ void load_record<FamWCSPH, true>(const double *__restrict__ rj, uint32_t fl, double4 &pj, double (&s)[8]) { load_record_wcsph<true>(rj, fl, pj, s); }
template <> __device__ __forceinline__ void load_record<FamWCSPH, false>(const double *__restrict__ rj, uint32_t fl, double4 &pj, double (&s)[8]) { load_record_wcsph<false>(rj, fl, pj, s); }

// ---- density summations (basic_equations.py:19-29, transport_velocity.py:24-58)
template <class T> struct FamDensity_T {
    typedef T Real; // arithmetic type of the pair loop
    static constexpr bool PRED = true; // see FamWCSPH
    static constexpr uint32_t CF0 = F_TVFSD; // flag set compiled as a constant (variant 6)
    static constexpr int MINB = 4; // wavefronts per SIMD the pair kernel is compiled for (VGPR budget)
    static constexpr int NA = 1; // m
    static constexpr int NR = 6;  // x y z h m pad
    struct Params { double *rho, *V; };
    struct Dest { T m, rho, V; };
    template <class A> static __device__ __forceinline__ void load(Dest &D, const T *a, const A &, uint32_t) { D.m = a[0]; D.rho = T(0.0); D.V = T(0.0); }
    template <int KK, bool UH, class A>
    static __device__ __forceinline__ void pair(Dest &D, const real4<T> &pi, const real4<T> &pj, T r2,
                                                const T (&s)[NA], uint32_t fl, const A &a, bool pass = true)
    {
        PairGeomT<T> g;
        pair_geom<KK, UH>(g, pi, pj, r2, a);
        T wij = pair_w<KK, UH>(g);
        wij = pass ? wij : T(0.0); // PRED: a pair outside the criterion adds exactly zero
        if (fl & F_SD) D.rho += s[0] * wij;
        if (fl & F_TVFSD) { D.V += wij; D.rho += D.m * wij; }
    }
    template <class A> static __device__ __forceinline__ void finish(Dest &D, const A &a, uint32_t o)
    {
        a.p.rho[o] = D.rho;
        if (a.dflags & F_TVFSD) a.p.V[o] = D.V;
    }
};
typedef FamDensity_T<double> FamDensity;

// Density records of the aggregated kernel under uniform h: [x y z m] (32 B, two
// 16-B pieces per pair); otherwise the generic [x y z h | m pad].
template <> __device__ __forceinline__ void load_record<FamDensity, true>(const double *__restrict__ rj, uint32_t fl, double4 &pj, double (&s)[1])
{
    const double2 *r2 = reinterpret_cast<const double2 *>(rj);
    const double2 a0 = r2[0], a1 = r2[1];
    pj.x = a0.x; pj.y = a0.y; pj.z = a1.x; pj.w = 0.0;
    s[0] = a1.y;
}

// ---- neighbour lists (LinkedListNNPS.find_nearest_neighbors, linked_list_nnps.pyx:92-196; the reference's
//      NeighborCache, nnps_base.pyx:1144-1257) on the pair-kernel skeleton: the "equation" counts the pairs
//      that pass the exact criterion and, in the fill pass, writes the neighbour's ORIGINAL index (it travels
//      in the record as a double) behind the destination's list start.  Same candidate tiles, fp32 prefilter
//      and exact test as every other family, so the lists are the pair loops' own neighbour sets.
struct FamNbr {
    typedef double Real;
    static constexpr bool PRED = true;
    static constexpr uint32_t CF0 = 1;
    static constexpr int MINB = 4;
    static constexpr int NA = 1; // original index of the particle
    static constexpr int NR = 6; // x y z h idx pad  (uniform h: [x y z idx])
    struct Params { uint32_t *count; const uint32_t *start; uint32_t *nbrs; };
    struct Dest { uint32_t n, base; };
    template <class A> static __device__ __forceinline__ void load(Dest &D, const double *, const A &a, uint32_t o)
    {
        D.n = 0;
        D.base = a.p.start ? a.p.start[o] : 0u;
    }
    template <int KK, bool UH, class A>
    static __device__ __forceinline__ void pair(Dest &D, const double4 &, const double4 &, double, const double (&s)[NA], uint32_t,
                                                const A &a, bool pass = true)
    {
        if (pass) {
            if (a.p.nbrs) a.p.nbrs[D.base + D.n] = (uint32_t)s[0];
            D.n++;
        }
    }
    template <class A> static __device__ __forceinline__ void finish(Dest &D, const A &a, uint32_t o)
    {
        if (a.p.count) a.p.count[o] = D.n;
    }
};
template <> __device__ __forceinline__ void load_record<FamNbr, true>(const double *__restrict__ rj, uint32_t, double4 &pj, double (&s)[1])
{
    const double2 *r2 = reinterpret_cast<const double2 *>(rj);
    const double2 a0 = r2[0], a1 = r2[1];
    pj.x = a0.x; pj.y = a0.y; pj.z = a1.x; pj.w = 0.0;
    s[0] = a1.y;
}

// ---- TVF momentum terms (transport_velocity.py:219-545) -------------------
template <class T> struct FamTVF_T {
    typedef T Real; // arithmetic type of the pair loop
    static constexpr bool PRED = true; // see FamWCSPH
    static constexpr uint32_t CF0 = F_TP | F_TVISC | F_TAS; // flag set compiled as a constant (variant 6)
    static constexpr int MINB = 4; // wavefronts per SIMD the pair kernel is compiled for (VGPR budget; fp64: 125-129 registers)
    static constexpr int NA = 12; // u v w uhat vhat what rho p V m Vj2 pad
    static constexpr int NR = 16; // x y z h + NA
    struct Params {
        T pb, gx, gy, gz, tdamp, nu, c0, alpha;
        const double *m; // the destination's own mass (a neighbour's travels in the records only with F_TAV)
        double *au, *av, *aw, *auhat, *avhat, *awhat;
    };
    struct Dest {
        T u, v, w, uh, vh, wh, rho, p, Vi2, mi1;
        T au, av, aw, auh, avh, awh;
    };
    template <class A> static __device__ __forceinline__ void load(Dest &D, const T *a, const A &A_, uint32_t o)
    {
        D.u = a[0]; D.v = a[1]; D.w = a[2]; D.uh = a[3]; D.vh = a[4]; D.wh = a[5];
        D.rho = a[6]; D.p = a[7]; D.Vi2 = a[10]; D.mi1 = T(1.0) / T(A_.p.m[o]);
        D.au = D.av = D.aw = D.auh = D.avh = D.awh = T(0.0);
    }
    template <int KK, bool UH, class A>
    static __device__ __forceinline__ void pair(Dest &D, const real4<T> &pi, const real4<T> &pj, T r2,
                                                const T (&s)[NA], uint32_t fl, const A &a, bool pass = true)
    {
        PairGeomT<T> g;
        pair_geom<KK, UH>(g, pi, pj, r2, a);
        T tg = pair_gradfac<KK, UH>(g);
        tg = pass ? tg : T(0.0); // PRED: every term carries DWIJ
        T dw0 = tg * g.xij[0], dw1 = tg * g.xij[1], dw2 = tg * g.xij[2];
        T rhoj = s[6], Vj2 = s[10];
        T vsum = D.Vi2 + Vj2;
        T vij0 = D.u - s[0], vij1 = D.v - s[1], vij2 = D.w - s[2];
        if (fl & F_TP) { // :290-320
            T pij = rhoj * D.p + D.rho * s[7];
            pij *= fast_rcp(rhoj + D.rho);
            T tmp = -pij * D.mi1 * vsum;
            D.au += tmp * dw0; D.av += tmp * dw1; D.aw += tmp * dw2;
            tmp = -a.p.pb * D.mi1 * vsum;
            D.auh += tmp * dw0; D.avh += tmp * dw1; D.awh += tmp * dw2;
        }
        if (fl & F_TAV) { // :420-436
            T vijdotrij = vij0 * g.xij[0] + vij1 * g.xij[1] + vij2 * g.xij[2];
            T piij = T(0.0);
            if (vijdotrij < 0) {
                T muij = (g.hij * vijdotrij) * fast_rcp(r2 + g.eps);
                piij = -a.p.alpha * a.p.c0 * muij;
                piij = s[9] * piij * fast_rcp(T(0.5) * (D.rho + rhoj));
            }
            D.au += -piij * dw0; D.av += -piij * dw1; D.aw += -piij * dw2;
        }
        if (fl & F_TVISC) { // :363-384
            T etai = a.p.nu * D.rho, etaj = a.p.nu * rhoj;
            T etaij = 2 * (etai * etaj) * fast_rcp(etai + etaj);
            T Fij = dw0 * g.xij[0] + dw1 * g.xij[1] + dw2 * g.xij[2];
            T tmp = D.mi1 * vsum * etaij * Fij * fast_rcp(r2 + g.eps);
            D.au += tmp * vij0; D.av += tmp * vij1; D.aw += tmp * vij2;
        }
        if (fl & F_TAS) { // :473-545
            // A = rho v (x) (vhat - v);  0.5 (A_i + A_j) . DWIJ, with the dot
            // products (vhat - v) . DWIJ taken first (same terms as the
            // reference's 18 products, associated per particle)
            const T ddi = (D.uh - D.u) * dw0 + (D.vh - D.v) * dw1 + (D.wh - D.w) * dw2;
            const T ddj = (s[3] - s[0]) * dw0 + (s[4] - s[1]) * dw1 + (s[5] - s[2]) * dw2;
            const T ci = D.rho * ddi, cj = rhoj * ddj;
            const T tmp = T(0.5) * D.mi1 * vsum;
            D.au += tmp * (ci * D.u + cj * s[0]);
            D.av += tmp * (ci * D.v + cj * s[1]);
            D.aw += tmp * (ci * D.w + cj * s[2]);
        }
    }
    template <class A> static __device__ __forceinline__ void finish(Dest &D, const A &a, uint32_t o)
    {
        T damp = T(1.0); // post_loop :322-325
        if (a.t < a.p.tdamp) damp = T(0.5) * (sin((-T(0.5) + a.t / a.p.tdamp) * M_PI) + T(1.0));
        T gx = (a.dflags & F_TP) ? a.p.gx * damp : T(0.0);
        T gy = (a.dflags & F_TP) ? a.p.gy * damp : T(0.0);
        T gz = (a.dflags & F_TP) ? a.p.gz * damp : T(0.0);
        a.p.au[o] = D.au + gx; a.p.av[o] = D.av + gy; a.p.aw[o] = D.aw + gz;
        if (a.dflags & F_TP) { a.p.auhat[o] = D.auh; a.p.avhat[o] = D.avh; a.p.awhat[o] = D.awh; }
    }
};
typedef FamTVF_T<double> FamTVF;

// TVF records of the aggregated kernel under uniform h (see k_pack layout 3):
// five 16-B pieces per pair, a sixth when the artificial-stress term acts, a
// seventh (the neighbour's mass) only for the artificial viscosity.
template <> __device__ __forceinline__ void load_record<FamTVF, true>(const double *__restrict__ rj, uint32_t fl, double4 &pj, double (&s)[12])
{
    const double2 *r2 = reinterpret_cast<const double2 *>(rj);
    const double2 a0 = r2[0], a1 = r2[1], b0 = r2[2], b1 = r2[3], c0 = r2[4];
    pj.x = a0.x; pj.y = a0.y; pj.z = a1.x; pj.w = 0.0;
    s[0] = b0.x; s[1] = b0.y; s[2] = b1.x; s[6] = a1.y; s[7] = b1.y; s[8] = 0.0; s[9] = 0.0; s[10] = c0.x; s[11] = 0.0;
    s[3] = s[4] = 0.0; s[5] = c0.y;
    if (fl & F_TAS) {
        const double2 d0 = r2[5];
        s[3] = d0.x; s[4] = d0.y;
    }
    if (fl & F_TAV) s[9] = r2[6].x;
}

// The same terms on records WITHOUT p and V (sph_group.src_eos = 2): the group before this one was TVF's StateEquation
// over every particle of every array read here and its density summation ran earlier in the evaluation, so
// p = p0 (rho / rho0 - b) and V = rho / m are functions of the gathered rho and the source array's ONE mass:
// [x y | z rho | u v | w uhat | vhat what] -- 80 bytes, five 16-B pieces instead of six (four without the artificial
// stress); fp32: [x y z rho | u v w uhat | vhat what - -], three pieces instead of four.  One reciprocal more per pair.
template <class T> struct FamTVFE_T : FamTVF_T<T> {
    static constexpr bool EOSF = true;
    static constexpr int NA = 12;
    template <class A> static __device__ __forceinline__ void load_fused(const A &a, uint32_t jg, uint32_t fl, T mu, real4<T> &pj, T (&s)[12])
    {
        T rho;
        if constexpr (sizeof(T) == 8) {
            const double2 *p = reinterpret_cast<const double2 *>(a.rec) + (unsigned long long)jg * (unsigned)(a.nrec >> 1);
            const double2 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
            pj.x = q0.x; pj.y = q0.y; pj.z = q1.x; pj.w = 0.0; rho = q1.y;
            s[0] = q2.x; s[1] = q2.y; s[2] = q3.x; s[3] = q3.y; s[4] = s[5] = 0.0;
            if (fl & F_TAS) { const double2 q4 = p[4]; s[4] = q4.x; s[5] = q4.y; }
        } else {
            const float4 *p = reinterpret_cast<const float4 *>(a.rec) + (unsigned long long)jg * (unsigned)(a.nrec >> 2);
            const float4 q0 = p[0], q1 = p[1];
            pj.x = q0.x; pj.y = q0.y; pj.z = q0.z; pj.w = 0.f; rho = q0.w;
            s[0] = q1.x; s[1] = q1.y; s[2] = q1.z; s[3] = q1.w; s[4] = s[5] = 0.f;
            if (fl & F_TAS) { const float4 q2 = p[2]; s[4] = q2.x; s[5] = q2.y; }
        }
        s[6] = rho;
        s[7] = (T)a.e_p0 * (rho * (T)a.e_rho01 - (T)a.e_B); // StateEquation, as k_nosrc computes it
        s[8] = T(0.0);
        s[9] = mu;
        const T vj = mu * fast_rcp(rho);                     // 1 / V_j with V = sum W = rho / m (transport_velocity.py:52-58): Vj2 = (m / rho)^2, :303-306
        s[10] = vj * vj;
        s[11] = T(0.0);
    }
};

// ---- velocity gradient (basic_equations.py:63-148) -------------------------
template <class T> struct FamVGrad_T {
    typedef T Real; // arithmetic type of the pair loop
    static constexpr bool PRED = true; // see FamWCSPH
    static constexpr uint32_t CF0 = F_VG3; // flag set compiled as a constant (variant 6)
    static constexpr int MINB = 4; // wavefronts per SIMD the pair kernel is compiled for (VGPR budget)
    static constexpr int NA = 4; // u v w m/rho
    static constexpr int NR = 8;
    struct Params { double *v[9]; };
    struct Dest { T u, v, w; T g[9]; };
    template <class A> static __device__ __forceinline__ void load(Dest &D, const T *a, const A &, uint32_t)
    {
        D.u = a[0]; D.v = a[1]; D.w = a[2];
        for (int k = 0; k < 9; k++) D.g[k] = T(0.0);
    }
    template <int KK, bool UH, class A>
    static __device__ __forceinline__ void pair(Dest &D, const real4<T> &pi, const real4<T> &pj, T r2,
                                                const T (&s)[NA], uint32_t fl, const A &a, bool pass = true)
    {
        PairGeomT<T> g;
        pair_geom<KK, UH>(g, pi, pj, r2, a);
        T tg = pair_gradfac<KK, UH>(g);
        tg = pass ? tg : T(0.0); // PRED
        const T dw[3] = {tg * g.xij[0], tg * g.xij[1], tg * g.xij[2]};
        const T tmp = s[3]; // m/rho (divided once per particle by k_pack)
        const T nv[3] = {-(D.u - s[0]), -(D.v - s[1]), -(D.w - s[2])};
        const int n = (fl & F_VG3) ? 3 : 2;
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++)
                if (i < n && j < n) D.g[3 * i + j] += tmp * nv[i] * dw[j];
    }
    template <class A> static __device__ __forceinline__ void finish(Dest &D, const A &a, uint32_t o)
    {
        const int n = (a.dflags & F_VG3) ? 3 : 2;
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++)
                if (i < n && j < n) a.p.v[3 * i + j][o] = D.g[3 * i + j];
    }
};
typedef FamVGrad_T<double> FamVGrad;

// ---- elastic rates: Continuity + MomentumEquationWithStress +
//      MonaghanArtificialViscosity + XSPH  (solid_mech/basic.py:245-387,
//      basic_equations.py:177-300) -------------------------------------------
template <class T> struct FamElastic_T {
    typedef T Real; // arithmetic type of the pair loop
    static constexpr bool PRED = true; // see FamWCSPH
    static constexpr uint32_t CF0 = F_ECONT | F_ESTRESS | F_EAV | F_EXSPH; // flag set compiled as a constant (variant 6)
    // wavefronts per SIMD the pair kernel is compiled for (VGPR budget): fp32 120 registers; fp64 215-227, and at 3
    // (168 registers + 190 B of scratch) the rings take the same 2.99 ms
    static constexpr int MINB = sizeof(T) == 4 ? 4 : 2;
    static constexpr int NA = 18; // u v w m rho cs | t00 t01 t02 t11 t12 t22 (= sigma/rho^2) | r00 r01 r02 r11 r12 r22
    static constexpr int NR = 22;
    struct Params {
        T wdeltap, n, alpha, beta, eps;
        double *arho, *au, *av, *aw, *ax, *ay, *az;
        const uint32_t *tension; // FamElasticU_T: the source array's "a particle is in tension" word, or null (always gather r_ij)
    };
    // The destination's own tensors are NOT live in the pair loop: with S = sum_j m_j DWIJ and F = sum_j m_j f^n DWIJ,
    //   sum_j m_j ((t_i + t_j) + f^n (r_i + r_j)) . DWIJ = t_i . S + r_i . F + sum_j m_j (t_j + f^n r_j) . DWIJ,
    // so t_i, r_i (24 registers in fp64) wait in scratch until finish() and the loop carries S and F (12) instead.
    struct Dest {
        T u, v, w, rho, cs, t[6], r[6];
        T arho, au, av, aw, ax, ay, az;
        T s[3], f[3];
    };
    template <class A> static __device__ __forceinline__ void load(Dest &D, const T *a, const A &, uint32_t)
    {
        D.u = a[0]; D.v = a[1]; D.w = a[2]; D.rho = a[4]; D.cs = a[5];
        for (int k = 0; k < 6; k++) { D.t[k] = a[6 + k]; D.r[k] = a[12 + k]; }
        D.arho = D.au = D.av = D.aw = D.ax = D.ay = D.az = T(0.0);
        for (int k = 0; k < 3; k++) D.s[k] = D.f[k] = T(0.0);
    }
    template <int KK, bool UH, class A>
    static __device__ __forceinline__ void pair(Dest &D, const real4<T> &pi, const real4<T> &pj, T r2,
                                                const T (&s)[NA], uint32_t fl, const A &a, bool pass = true)
    {
        PairGeomT<T> g;
        pair_geom<KK, UH>(g, pi, pj, r2, a);
        T tg = pair_gradfac<KK, UH>(g);
        tg = pass ? tg : T(0.0); // PRED: terms carry DWIJ, the XSPH one WIJ
        const T dw0 = tg * g.xij[0], dw1 = tg * g.xij[1], dw2 = tg * g.xij[2];
        const T vij0 = D.u - s[0], vij1 = D.v - s[1], vij2 = D.w - s[2];
        const T vdotx = vij0 * g.xij[0] + vij1 * g.xij[1] + vij2 * g.xij[2];
        const T mj = s[3];
        if (fl & F_ECONT) D.arho = fma(mj * tg, vdotx, D.arho);
        T wij = T(0.0);
        if (fl & (F_ESTRESS | F_EXSPH)) wij = pair_w<KK, UH>(g);
        wij = pass ? wij : T(0.0);
        if (fl & F_ESTRESS) { // solid_mech/basic.py:267-387
            T fab = T(0.0);
            if (a.p.wdeltap > T(0.)) {
                const T f = wij * fast_rcp(a.p.wdeltap);
                if (a.p.n == T(4.0)) { const T f2 = f * f; fab = f2 * f2; }
                else if (a.p.n == T(2.0)) fab = f * f;
                else if (a.p.n == T(1.0)) fab = f;
                else fab = pow(f, a.p.n);
            }
            const T m0 = mj * dw0, m1 = mj * dw1, m2 = mj * dw2;
            D.s[0] += m0; D.s[1] += m1; D.s[2] += m2;
            D.f[0] = fma(fab, m0, D.f[0]); D.f[1] = fma(fab, m1, D.f[1]); D.f[2] = fma(fab, m2, D.f[2]);
            const T a00 = fma(fab, s[12], s[6]), a01 = fma(fab, s[13], s[7]), a02 = fma(fab, s[14], s[8]);
            const T a11 = fma(fab, s[15], s[9]), a12 = fma(fab, s[16], s[10]), a22 = fma(fab, s[17], s[11]);
            D.au += a00 * m0 + a01 * m1 + a02 * m2;
            D.av += a01 * m0 + a11 * m1 + a12 * m2;
            D.aw += a02 * m0 + a12 * m1 + a22 * m2;
        }
        if (fl & (F_EAV | F_EXSPH)) {
            const T rhoij = T(0.5) * (D.rho + s[4]);
            T rhoij1;
            if (fl & F_EAV) { // basic_equations.py:236-257
                const T re = r2 + g.eps;
                const T tt = fast_rcp(re * rhoij);
                rhoij1 = re * tt;
                const T muij = (g.hij * vdotx) * (rhoij * tt);
                const T cij = T(0.5) * (D.cs + s[5]);
                T piij = (a.p.beta * muij - a.p.alpha * cij) * muij * rhoij1;
                piij = vdotx < 0 ? piij : T(0.0);
                const T f = -mj * piij;
                D.au = fma(f, dw0, D.au); D.av = fma(f, dw1, D.av); D.aw = fma(f, dw2, D.aw);
            } else rhoij1 = fast_rcp(rhoij);
            if (fl & F_EXSPH) { // basic_equations.py:290-295
                const T tmp = -a.p.eps * mj * wij * rhoij1;
                D.ax = fma(tmp, vij0, D.ax); D.ay = fma(tmp, vij1, D.ay); D.az = fma(tmp, vij2, D.az);
            }
        }
    }
    template <class A> static __device__ __forceinline__ void finish(Dest &D, const A &a, uint32_t o)
    {
        if (a.dflags & F_ECONT) a.p.arho[o] = D.arho;
        if (a.dflags & F_ESTRESS) { // the destination's own share of the stress term
            D.au += D.t[0] * D.s[0] + D.t[1] * D.s[1] + D.t[2] * D.s[2] + D.r[0] * D.f[0] + D.r[1] * D.f[1] + D.r[2] * D.f[2];
            D.av += D.t[1] * D.s[0] + D.t[3] * D.s[1] + D.t[4] * D.s[2] + D.r[1] * D.f[0] + D.r[3] * D.f[1] + D.r[4] * D.f[2];
            D.aw += D.t[2] * D.s[0] + D.t[4] * D.s[1] + D.t[5] * D.s[2] + D.r[2] * D.f[0] + D.r[4] * D.f[1] + D.r[5] * D.f[2];
        }
        if (a.dflags & (F_ESTRESS | F_EAV)) { a.p.au[o] = D.au; a.p.av[o] = D.av; a.p.aw[o] = D.aw; }
        if (a.dflags & F_EXSPH) { a.p.ax[o] = D.ax + D.u; a.p.ay[o] = D.ay + D.v; a.p.az[o] = D.az + D.w; }
    }
};
typedef FamElastic_T<double> FamElastic;

// The same terms on records without h and m (uniform h, ONE mass per source array -- seen by the neighbour update's
// reduction, as for the WCSPH uniform-mass records): [x y | z u | v w | rho cs | t00 t01 | t02 t11 | t12 t22 | r00 r01 |
// r02 r11 | r12 r22], 160 bytes = ten 16-B pieces instead of eleven; fp32 [x y z u | v w rho cs | t00 t01 t02 t11 |
// t12 t22 r00 r01 | r02 r11 r12 r22], 80 bytes = five pieces instead of six.
template <class T> struct FamElasticU_T : FamElastic_T<T> {
    static constexpr bool EOSF = true;  // own record decoding (load_fused)
    static constexpr bool TOKEN = true; // wave token: 1 = gather the r_ij, 0 = no particle of the source is in tension (r_ij = 0:
                                        // MonaghanArtificialStress, solid_mech/basic.py:170-242, leaves them zero under compression)
    static constexpr int NA = 18;
    template <class A> static __device__ __forceinline__ uint32_t wave_token(const A &a)
    {
        return a.p.tension ? (uint32_t)__builtin_amdgcn_readfirstlane((int)*a.p.tension) : 1u;
    }
    // 16-B pieces of one record, and the record decoded from wherever its pieces are (the packed buffer, or the LDS copy
    // of a row tile: k_pair_rowlds)
    static constexpr int PIECES = sizeof(T) == 8 ? 10 : 5;
    static constexpr bool ROWLDS = true;
    typedef typename std::conditional<sizeof(T) == 8, double2, float4>::type Piece;
    template <class P> static __device__ __forceinline__ void decode(P p, T mu, real4<T> &pj, T (&s)[18], uint32_t with_r)
    {
        if constexpr (sizeof(T) == 8) {
            double2 q[10];
#pragma unroll
            for (int k = 0; k < 7; k++) q[k] = p[k];
            if (with_r) { q[7] = p[7]; q[8] = p[8]; q[9] = p[9]; }
            else q[7] = q[8] = q[9] = make_double2(0.0, 0.0);
            pj.x = q[0].x; pj.y = q[0].y; pj.z = q[1].x; pj.w = 0.0;
            s[0] = q[1].y; s[1] = q[2].x; s[2] = q[2].y; s[3] = mu; s[4] = q[3].x; s[5] = q[3].y;
#pragma unroll
            for (int k = 0; k < 6; k++) { s[6 + 2 * k] = q[4 + k].x; s[7 + 2 * k] = q[4 + k].y; }
        } else {
            float4 q[5];
#pragma unroll
            for (int k = 0; k < 4; k++) q[k] = p[k];
            if (with_r) q[4] = p[4];
            else { q[4] = make_float4(0.f, 0.f, 0.f, 0.f); q[3].z = 0.f; q[3].w = 0.f; }
            pj.x = q[0].x; pj.y = q[0].y; pj.z = q[0].z; pj.w = 0.f;
            s[0] = q[0].w; s[1] = q[1].x; s[2] = q[1].y; s[3] = mu; s[4] = q[1].z; s[5] = q[1].w;
#pragma unroll
            for (int k = 0; k < 3; k++) { s[6 + 4 * k] = q[2 + k].x; s[7 + 4 * k] = q[2 + k].y; s[8 + 4 * k] = q[2 + k].z; s[9 + 4 * k] = q[2 + k].w; }
        }
    }
    template <class A> static __device__ __forceinline__ void load_fused(const A &a, uint32_t jg, uint32_t, T mu, real4<T> &pj, T (&s)[18],
                                                                         uint32_t with_r = 1u)
    {
        decode(reinterpret_cast<const Piece *>(a.rec) + (unsigned long long)jg * PIECES, mu, pj, s, with_r);
    }
};

// ---------------------------------------------------------------------------
// variant 0: per-lane walk over the 3x3 rows of cells (x-contiguous ranges)
// ---------------------------------------------------------------------------
template <class Fam> __device__ __forceinline__ void load_aux(double (&s)[Fam::NA], const double *__restrict__ p)
{
#pragma unroll
    for (int k = 0; k < Fam::NA; k++) s[k] = p[k];
}

template <class Fam, int KK, bool UH> __global__ __launch_bounds__(256) void k_pair_direct(PairArgs<Fam> a)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.nd) return;
    uint32_t o = a.d_perm[i];
    if (o < a.d_start || o >= a.d_stop) return;
    double4 pi = a.posh[a.d_off + i];
    typename Fam::Dest D;
    Fam::load(D, a.aux + (size_t)(a.d_off + i) * Fam::NA, a, o);
    uint32_t key = a.d_keys[i];
    int cx = key % a.nc[0];
    int t = key / a.nc[0];
    int cy = t % a.nc[1], cz = t / a.nc[1];
    double hi2 = a.radius_scale * pi.w;
    hi2 *= hi2;
    for (int s = 0; s < a.nsrc; s++) {
        const SrcDesc sd = a.src[s];
        for (int oz = -1; oz <= 1; oz++)
            for (int oy = -1; oy <= 1; oy++) {
                int yy = cy + oy, zz = cz + oz;
                if (yy < 0 || yy >= a.nc[1] || zz < 0 || zz >= a.nc[2]) continue;
                int xa = max(cx - 1, 0), xb = min(cx + 1, a.nc[0] - 1);
                uint32_t row = (uint32_t)(a.nc[0] * (yy + a.nc[1] * zz));
                uint32_t j0 = sd.cell_start[row + xa], j1 = sd.cell_start[row + xb + 1];
                for (uint32_t j = j0; j < j1; j++) {
                    double4 pj = a.posh[sd.off + j];
                    double hj2 = a.radius_scale * pj.w;
                    hj2 *= hj2;
                    double r2 = r2_exact(pi.x - pj.x, pi.y - pj.y, pi.z - pj.z);
                    if ((r2 < hi2) || (r2 < hj2)) {
                        double sj[Fam::NA];
                        load_aux<Fam>(sj, a.aux + (size_t)(sd.off + j) * Fam::NA);
                        Fam::template pair<KK, UH>(D, pi, pj, r2, sj, sd.flags, a);
                    }
                }
            }
    }
    Fam::finish(D, a, o);
}

// ---------------------------------------------------------------------------
// host driver
// ---------------------------------------------------------------------------
static bool is_nosrc_kind(int k)
{
    return k == SPH_EQ_TAIT_EOS || k == SPH_EQ_TAIT_EOS_HG || k == SPH_EQ_TVF_STATE_EQUATION ||
           k == SPH_EQ_ISOTHERMAL_EOS || k == SPH_EQ_SOLID_ISOTHERMAL_EOS || k == SPH_EQ_MONAGHAN_ART_STRESS ||
           k == SPH_EQ_HOOKES_DEVIATORIC_STRESS_RATE;
}

static int need_prop(sph_ctx *c, int id, int prop, const char *who)
{
    if (!c->arr[id].prop[prop]) {
        sph_set_error("%s: array %d has no device copy of a required property (id %d); push it first", who, id, prop);
        return SPH_ERR_MISSING_PROP;
    }
    return SPH_OK;
}

static int run_nosrc(sph_ctx *c, const sph_equation &e, size_t start, size_t stop)
{
    if (stop <= start) return SPH_OK;
    DevArray &A = c->arr[e.dest];
    EosArgs a;
    a.kind = e.kind;
    memcpy(a.par, e.par, sizeof a.par);
    SPH_TRY(need_prop(c, e.dest, SPH_RHO, "EOS"));
    SPH_TRY(sph_array_ensure_prop(c, e.dest, SPH_P));
    a.rho = A.prop[SPH_RHO];
    a.p = A.prop[SPH_P];
    a.cs = nullptr;
    if (e.kind == SPH_EQ_TAIT_EOS || e.kind == SPH_EQ_TAIT_EOS_HG) {
        SPH_TRY(sph_array_ensure_prop(c, e.dest, SPH_CS));
        a.cs = A.prop[SPH_CS];
    }
    a.start = start;
    a.stop = stop;
    if (e.kind == SPH_EQ_MONAGHAN_ART_STRESS) {
        for (int p : {SPH_S00, SPH_S01, SPH_S02, SPH_S11, SPH_S12, SPH_S22}) SPH_TRY(need_prop(c, e.dest, p, "MonaghanArtificialStress"));
        for (int p : {SPH_R00, SPH_R01, SPH_R02, SPH_R11, SPH_R12, SPH_R22}) SPH_TRY(sph_array_ensure_prop(c, e.dest, p));
    }
    if (e.kind == SPH_EQ_HOOKES_DEVIATORIC_STRESS_RATE) {
        for (int p : {SPH_S00, SPH_S01, SPH_S02, SPH_S11, SPH_S12, SPH_S22}) SPH_TRY(need_prop(c, e.dest, p, "HookesDeviatoricStressRate"));
        for (int p : {SPH_V00, SPH_V01, SPH_V02, SPH_V10, SPH_V11, SPH_V12, SPH_V20, SPH_V21, SPH_V22, SPH_AS00, SPH_AS01, SPH_AS02,
                      SPH_AS11, SPH_AS12, SPH_AS22})
            SPH_TRY(sph_array_ensure_prop(c, e.dest, p));
    }
    for (int k = 0; k < SPH_PROP_COUNT; k++) a.q[k] = A.prop[k];
    a.rho = A.prop[SPH_RHO];
    a.p = A.prop[SPH_P];
    a.tflag = nullptr;
    if (e.kind == SPH_EQ_MONAGHAN_ART_STRESS) {
        // "no particle in tension" for the rates kernel: valid when this launch covers every particle of the array
        SPH_TRY(A.tflag.reserve(64));
        if (start == 0) HIP_TRY(hipMemsetAsync(A.tflag.ptr, 0, 4, c->stream)); // (a phase-2 launch over the ghosts only adds to the word)
        a.tflag = A.tflag.as<uint32_t>();
        A.tflag_valid = (start == 0 && stop == A.n) || (A.tflag_valid && start <= A.n_binned && stop == A.n);
    }
    ScopedTimer tm(c, T_EOS);
    hipLaunchKernelGGL(k_nosrc, dim3(div_up(stop - start, 256)), dim3(256), 0, c->stream, a);
    return SPH_OK;
}

enum Family { FAM_NONE, FAM_WCSPH, FAM_DENSITY, FAM_TVF, FAM_VGRAD, FAM_ELASTIC, FAM_NBR };

static int eq_family(int kind, uint32_t *flag, bool elastic)
{
    if (elastic) {
        switch (kind) {
        case SPH_EQ_CONTINUITY: *flag = F_ECONT; return FAM_ELASTIC;
        case SPH_EQ_MOMENTUM_WITH_STRESS: *flag = F_ESTRESS; return FAM_ELASTIC;
        case SPH_EQ_MONAGHAN_ART_VISCOSITY: *flag = F_EAV; return FAM_ELASTIC;
        case SPH_EQ_XSPH: *flag = F_EXSPH; return FAM_ELASTIC;
        default: break;
        }
    }
    switch (kind) {
    case SPH_EQ_CONTINUITY: *flag = F_CONT; return FAM_WCSPH;
    case SPH_EQ_MOMENTUM: *flag = F_MOM; return FAM_WCSPH;
    case SPH_EQ_XSPH: *flag = F_XSPH; return FAM_WCSPH;
    case SPH_EQ_SUMMATION_DENSITY: *flag = F_SD; return FAM_DENSITY;
    case SPH_EQ_TVF_SUMMATION_DENSITY: *flag = F_TVFSD; return FAM_DENSITY;
    case SPH_EQ_TVF_MOM_PRESSURE: *flag = F_TP; return FAM_TVF;
    case SPH_EQ_TVF_MOM_VISCOSITY: *flag = F_TVISC; return FAM_TVF;
    case SPH_EQ_TVF_MOM_ART_VISCOSITY: *flag = F_TAV; return FAM_TVF;
    case SPH_EQ_TVF_MOM_ART_STRESS: *flag = F_TAS; return FAM_TVF;
    case SPH_EQ_VELOCITY_GRADIENT_2D: *flag = F_VG2; return FAM_VGRAD;
    case SPH_EQ_VELOCITY_GRADIENT_3D: *flag = F_VG3; return FAM_VGRAD;
    default: return FAM_NONE;
    }
}

struct PackPlan {
    int nr; // doubles per interleaved record (== Fam::NR)
    int na;
    int props[MAX_AUX]; // sph_prop or -1
    int derived;
};

static PackPlan pack_plan(int fam)
{
    PackPlan p;
    for (auto &v : p.props) v = -1;
    p.derived = 0;
    if (fam == FAM_WCSPH) {
        p.nr = FamWCSPH::NR;
        p.na = 8;
        int pr[8] = {SPH_U, SPH_V, SPH_W, SPH_M, SPH_RHO, -1, SPH_CS, SPH_P};
        for (int k = 0; k < 8; k++) p.props[k] = pr[k];
        p.derived = 1;
    } else if (fam == FAM_DENSITY) {
        p.nr = FamDensity::NR;
        p.na = 1;
        p.props[0] = SPH_M;
    } else if (fam == FAM_NBR) {
        p.nr = FamNbr::NR;
        p.na = 1;
        p.derived = 5; // aux[0] = the particle's original index
    } else if (fam == FAM_VGRAD) {
        p.nr = FamVGrad::NR;
        p.na = 4;
        int pr[5] = {SPH_U, SPH_V, SPH_W, SPH_M, SPH_RHO};
        for (int k = 0; k < 5; k++) p.props[k] = pr[k];
        p.derived = 4;
    } else if (fam == FAM_ELASTIC) {
        p.nr = FamElastic::NR;
        p.na = 18;
        int pr[19] = {SPH_U, SPH_V, SPH_W, SPH_M, SPH_RHO, SPH_CS, SPH_S00, SPH_S01, SPH_S02, SPH_S11, SPH_S12, SPH_S22,
                      SPH_R00, SPH_R01, SPH_R02, SPH_R11, SPH_R12, SPH_R22, SPH_P};
        for (int k = 0; k < 19; k++) p.props[k] = pr[k];
        p.derived = 3;
    } else {
        p.nr = FamTVF::NR;
        p.na = 12;
        int pr[12] = {SPH_U, SPH_V, SPH_W, SPH_UHAT, SPH_VHAT, SPH_WHAT, SPH_RHO, SPH_P, SPH_VOL, SPH_M, -1, -1};
        for (int k = 0; k < 12; k++) p.props[k] = pr[k];
        p.derived = 2;
    }
    return p;
}

// which aux slots a family really needs given the union of flags (others may be absent -> 0)
static bool slot_required(int fam, uint32_t flags, int prop)
{
    if (fam == FAM_WCSPH) {
        if (prop == SPH_M || prop == SPH_U || prop == SPH_V || prop == SPH_W) return true;
        if (prop == SPH_RHO) return flags & (F_MOM | F_XSPH);
        if (prop == SPH_P || prop == SPH_CS) return flags & F_MOM;
        return false;
    }
    if (fam == FAM_DENSITY) return prop == SPH_M;
    if (fam == FAM_NBR) return false;
    if (fam == FAM_VGRAD) return true;
    if (fam == FAM_ELASTIC) {
        if (prop == SPH_U || prop == SPH_V || prop == SPH_W || prop == SPH_M) return true;
        if (prop == SPH_RHO) return flags & (F_ESTRESS | F_EAV | F_EXSPH);
        if (prop == SPH_CS) return flags & F_EAV;
        if (prop == SPH_P) return flags & F_ESTRESS;
        if (prop >= SPH_S00 && prop <= SPH_S22) return flags & F_ESTRESS;
        return false; // r_ij default to 0 when absent
    }
    if (fam == FAM_TVF) {
        if (prop == SPH_UHAT || prop == SPH_VHAT || prop == SPH_WHAT) return flags & F_TAS;
        return true;
    }
    return false;
}

// 16-B pieces of one packed record, as k_pack counts them; 0: records that are not a whole number of
// pieces (no LDS transposition, per-lane stores)
static int pack_pieces(const PackArgs &pa)
{
    if (!pa.rec) return 0;
    if (pa.layout == 6) return 4;
    if (pa.layout == 7) return 2;
    if (pa.layout == 8) return pa.nr / 2;
    if (pa.layout == 9) return pa.nr / 4;
    if (pa.layout == 10) return 10;
    if (pa.layout == 11) return 5;
    if (pa.layout == 12) return 4;
    if (pa.layout == 13) return 2;
    if (pa.layout == 5) return (pa.nr % 4) ? 0 : pa.nr / 4;
    if (pa.nr % 2) return 0;
    return pa.nr / 2;
}

// `launch` false: validate only (the records of this array are already in the packed buffer, see
// the pack cache in sph_eval_group)
// k_pack with the layout as a compile-time constant where the library has an instantiation
static void launch_pack(sph_ctx *c, const PackArgs &pa, size_t n)
{
    const dim3 g(div_up(n, 256)), b(256);
    const size_t lds = (size_t)pa.lds_np * 256 * 16;
    switch (pa.rec ? pa.layout : -1) {
    case 2: hipLaunchKernelGGL(k_pack<2>, g, b, lds, c->stream, pa); break;
    case 3: hipLaunchKernelGGL(k_pack<3>, g, b, lds, c->stream, pa); break;
    case 8: hipLaunchKernelGGL(k_pack<8>, g, b, lds, c->stream, pa); break;
    case 9: hipLaunchKernelGGL(k_pack<9>, g, b, lds, c->stream, pa); break;
    case 6: hipLaunchKernelGGL(k_pack<6>, g, b, lds, c->stream, pa); break;
    case 7: hipLaunchKernelGGL(k_pack<7>, g, b, lds, c->stream, pa); break;
    case 10: hipLaunchKernelGGL(k_pack<10>, g, b, lds, c->stream, pa); break;
    case 11: hipLaunchKernelGGL(k_pack<11>, g, b, lds, c->stream, pa); break;
    case 12: hipLaunchKernelGGL(k_pack<12>, g, b, lds, c->stream, pa); break;
    case 13: hipLaunchKernelGGL(k_pack<13>, g, b, lds, c->stream, pa); break;
    default: hipLaunchKernelGGL(k_pack<-1>, g, b, lds, c->stream, pa); break;
    }
}

static int pack_array(sph_ctx *c, int id, size_t off, const PackPlan &pl, int fam, uint32_t flags, bool dest_only = false,
                      bool launch = true, int seg = 0)
{
    DevArray &A = c->arr[id];
    // seg 0: the particles sph_nnps_update binned, through their cell order; seg 1: the ghosts binned after it
    // (sph_nnps_update_ghosts), through theirs, behind the first segment's records
    const size_t nseg = seg ? A.g_n : A.n_binned;
    if (nseg == 0) return SPH_OK;
    PackArgs pa;
    pa.perm = seg ? A.g_perm.as<uint32_t>() : A.perm.as<uint32_t>();
    pa.n = nseg;
    pa.off = off + (seg ? A.n_binned : 0);
    pa.x = A.prop[SPH_X]; pa.y = A.prop[SPH_Y]; pa.z = A.prop[SPH_Z]; pa.h = A.prop[SPH_H];
    pa.na = pl.na;
    for (int k = 0; k < MAX_AUX; k++) {
        pa.src[k] = nullptr;
        if ((k < pl.na || (pl.derived == 3 && k == 18) || (pl.derived == 4 && k == 4)) && pl.props[k] >= 0) {
            pa.src[k] = A.prop[pl.props[k]];
            // a destination that is not among the sources is only read through Fam::load:
            // it does not need a mass unless the equation uses the destination's own
            // (transport_velocity.SummationDensity: rho_i = m_i sum W)
            const bool mass_free = dest_only && pl.props[k] == SPH_M &&
                                   (fam == FAM_WCSPH || (fam == FAM_DENSITY && !(flags & F_TVFSD)));
            if (!pa.src[k] && !mass_free && slot_required(fam, flags, pl.props[k])) return need_prop(c, id, pl.props[k], "pair loop");
        }
    }
    pa.derived = pl.derived;
    pa.posh = c->posh.as<double4>();
    pa.aux = c->aux.as<double>();
    pa.rec = nullptr;
    pa.nr = pl.nr;
    pa.fpos = nullptr;
    for (int k = 0; k < 3; k++) pa.gmin[k] = c->xmin[k];
    pa.radius_scale = c->radius_scale;
    if (c->pair_variant >= 2) pa.rec = c->posh.as<double>();
    if (c->pair_variant >= 3) pa.fpos = c->fposb.as<float4>();
    pa.layout = (c->pair_variant >= 3 && fam == FAM_WCSPH) ? 1 : (c->pair_variant >= 3 && (fam == FAM_DENSITY || fam == FAM_NBR) && pl.nr == 4) ? 2
              : (c->pair_variant >= 3 && fam == FAM_TVF && (pl.nr == 14 || pl.nr == 12)) ? 3 : 0;
    if (c->pair_variant >= 3 && (c->record_f32 || c->arith_f32)) pa.layout = 5;
    pa.umass = 0;
    if (c->cur_eosf) {
        // p, cs (and p / rho^2) are recomputed by the pair kernel: not read here
        pa.layout = c->arith_f32 ? 7 : 6;
        pa.src[5] = pa.src[6] = nullptr;
        if (c->cur_umass) { // ... except p / rho^2, which takes the place of the (uniform) mass: derived 1 reads p
            pa.umass = 1;
        } else {
            pa.src[7] = nullptr;
            pa.derived = 0;
        }
    }
    if (c->cur_elu) pa.layout = c->arith_f32 ? 11 : 10; // h and m are launch / source constants
    if (c->cur_eosv) { // p, cs, p / rho^2 recomputed, m a source constant: none of them read here
        pa.layout = c->arith_f32 ? 13 : 12;
        pa.src[3] = pa.src[5] = pa.src[6] = pa.src[7] = nullptr;
        pa.derived = 0;
    }
    if (c->cur_tvff) { // p and V are functions of rho (and the one mass): not read here
        pa.layout = c->arith_f32 ? 9 : 8;
        pa.src[7] = pa.src[8] = pa.src[9] = nullptr;
        pa.derived = 0;
    }
    pa.lds_np = pack_pieces(pa);
    if (launch) launch_pack(c, pa, nseg);
    return SPH_OK;
}

template <class Fam> static int launch_pair(sph_ctx *c, int kk, const PairArgs<Fam> &a)
{
    if (a.nd == 0) return SPH_OK;
    const bool uh = c->uniform_h && c->use_uniform_h;
    constexpr bool FP32 = sizeof(typename Fam::Real) == 4;
    if (c->pair_variant == 6) {
        dim3 g2(div_up(4 * div_up(a.nd, 256), WPB)), b2(64 * WPB);
        // equation flags as a compile-time constant when every source carries the same set
        uint32_t cf = a.src[0].flags;
        for (int j = 1; j < a.nsrc; j++) if (a.src[j].flags != cf) cf = 0;
        if (c->const_flags == 0) cf = 0;
#define LAUNCH6F(K, UHV, F32V, CFV) hipLaunchKernelGGL((k_pair_wave<Fam, K, UHV, F32V, CFV>), g2, b2, (size_t)c->lds_pad, c->stream, a)
#define LAUNCH6U(K, F32V)                                                                                 \
        if (uh) { if (cf == Fam::CF0) LAUNCH6F(K, true, F32V, Fam::CF0); else LAUNCH6F(K, true, F32V, 0); }  \
        else { if (cf == Fam::CF0) LAUNCH6F(K, false, F32V, Fam::CF0); else LAUNCH6F(K, false, F32V, 0); }
#define LAUNCH6(K)                                                       \
        if constexpr (FP32) { LAUNCH6U(K, true) }                        \
        else { if (c->record_f32) { LAUNCH6U(K, true) } else { LAUNCH6U(K, false) } }
        switch (kk) {
        case 1: LAUNCH6(1); break;
        case 2: LAUNCH6(2); break;
        case 3: LAUNCH6(3); break;
        case 4: LAUNCH6(4); break;
        }
#undef LAUNCH6
#undef LAUNCH6U
#undef LAUNCH6F
        return SPH_OK;
    }
    if constexpr (FP32) {
        sph_set_error("fp32 arithmetic (arith_f32) needs pair_variant 6");
        return SPH_ERR_UNSUPPORTED;
    } else {
        dim3 gd(div_up(a.nd, 256)), bd(256);
#define LAUNCH(K)                                                                                \
        if (uh) hipLaunchKernelGGL((k_pair_direct<Fam, K, true>), gd, bd, 0, c->stream, a);      \
        else hipLaunchKernelGGL((k_pair_direct<Fam, K, false>), gd, bd, 0, c->stream, a)
        switch (kk) {
        case 1: LAUNCH(1); break;
        case 2: LAUNCH(2); break;
        case 3: LAUNCH(3); break;
        case 4: LAUNCH(4); break;
        }
#undef LAUNCH
        return SPH_OK;
    }
}

// the EOS-fused WCSPH family: uniform h, variant 6 only (sph_eval_group checks)
template <class Fam, bool UHV = true> static int launch_pair_fused(sph_ctx *c, int kk, const PairArgs<Fam> &a)
{
    if (a.nd == 0) return SPH_OK;
    constexpr bool FP32 = sizeof(typename Fam::Real) == 4;
    dim3 g2(div_up(4 * div_up(a.nd, 256), WPB)), b2(64 * WPB);
    uint32_t cf = a.src[0].flags;
    for (int j = 1; j < a.nsrc; j++) if (a.src[j].flags != cf) cf = 0;
    if (c->const_flags == 0) cf = 0;
    if constexpr (fam_rowlds<Fam>::value && UHV) {
        if (c->row_lds && !a.nl_mode && !a.face_mode) { // the experiment of round 6: source records of a row tile in LDS
            dim3 g1(4 * div_up(a.nd, 256)), b1(64);
#define LAUNCHL(K)                                                                                              \
            if (cf == Fam::CF0) hipLaunchKernelGGL((k_pair_rowlds<Fam, K, Fam::CF0>), g1, b1, 0, c->stream, a); \
            else hipLaunchKernelGGL((k_pair_rowlds<Fam, K, 0>), g1, b1, 0, c->stream, a)
            switch (kk) {
            case 1: LAUNCHL(1); break;
            case 2: LAUNCHL(2); break;
            case 3: LAUNCHL(3); break;
            case 4: LAUNCHL(4); break;
            }
#undef LAUNCHL
            c->timers[T_N_ROWLDS].count++;
            return SPH_OK;
        }
    }
#define LAUNCHE(K)                                                                                                              \
    if (cf == Fam::CF0) hipLaunchKernelGGL((k_pair_wave<Fam, K, UHV, FP32, Fam::CF0>), g2, b2, (size_t)c->lds_pad, c->stream, a); \
    else hipLaunchKernelGGL((k_pair_wave<Fam, K, UHV, FP32, 0>), g2, b2, (size_t)c->lds_pad, c->stream, a)
    switch (kk) {
    case 1: LAUNCHE(1); break;
    case 2: LAUNCHE(2); break;
    case 3: LAUNCHE(3); break;
    case 4: LAUNCHE(4); break;
    }
#undef LAUNCHE
    return SPH_OK;
}

template <class Fam>
static void fill_common(sph_ctx *c, PairArgs<Fam> &a, const sph_kernel *K, double t)
{
    a.posh = c->posh.as<double4>();
    a.aux = c->aux.as<double>();
    a.rec = c->posh.as<double>();
    a.nrec = c->cur_nrec;
    a.fpos = c->fposb.as<float4>();
    a.dom_extent = fmax(fmax(c->xmax[0] - c->xmin[0], c->xmax[1] - c->xmin[1]), c->xmax[2] - c->xmin[2]);
    for (int k = 0; k < 3; k++) { a.nc[k] = c->nc[k]; a.xmin[k] = c->xmin[k]; }
    a.cell_size = c->cell_size;
    a.radius_scale = c->radius_scale;
    a.k.sigma = K->fac;
    a.k.deltap = K->deltap;
    a.k.dim = K->dim;
    a.t = t;
    a.dt = c->cur_dt;
    a.ablate = (int)c->ablate;
    a.dbg = c->count_iters ? c->dbgc.as<unsigned long long>() : nullptr;
    // uniform-h constants, computed as the general path would per pair
    a.hu = 0.5 * (c->h_uniform + c->h_uniform);
    a.h1u = 1.0 / a.hu;
    a.facu = K->fac * a.h1u;
    if (K->dim > 1) a.facu *= a.h1u;
    if (K->dim > 2) a.facu *= a.h1u;
    a.epsu = 0.01 * a.hu * a.hu;
    a.hr2u = (c->radius_scale * c->h_uniform) * (c->radius_scale * c->h_uniform);
    a.norm_masks = (int)c->norm_masks;
    a.face_mode = 0;
    a.gfx_lo = c->gfx_lo; // (no faces named: -/+ INT_MAX, no face wavefronts)
    a.gfx_hi = c->gfx_hi;
}

// traversal order of the destination's tiles (sph_nnps_update builds it; option tile_block_rows) and of a tile's rows
template <class Fam> static void set_tile_order(sph_ctx *c, PairArgs<Fam> &a, const DevArray &D)
{
    a.d_tile_order = D.n_tiles ? D.tile_order.as<uint32_t>() : nullptr;
    a.row_mod3 = (int)c->row_mod3;
}
// the destinations of this launch are exactly the real particles of D's order: wave tiles from D.dlist (no idle ghost lanes)
template <class Fam> static void use_dest_list(sph_ctx *c, PairArgs<Fam> &a, const DevArray &D)
{
    if (!c->dest_list || c->pair_variant != 6 || D.dlist_n == 0 || a.nl_mode || a.face_mode) return;
    a.d_list = D.dlist.as<uint32_t>();
    a.nd = (uint32_t)D.dlist_n;
    c->timers[T_N_DLIST].count++; // (pair launches whose wave tiles came from the list: sph_timer_get "n_dest_list")
    a.d_tile_order = D.n_ctiles ? D.ctile_order.as<uint32_t>() : nullptr;
}

static int ensure_out(sph_ctx *c, int id, std::initializer_list<int> props)
{
    for (int p : props) SPH_TRY(sph_array_ensure_prop(c, id, p));
    return SPH_OK;
}

// Neighbour lists of `dst` among `src` through the wave-tile pair kernel (nnps_build_csr_device /
// sph_nnps_get_csr): count pass (count[nd] filled, start == nullptr) or fill pass (start[nd], nbrs filled).
int nnps_csr_pair_kernel(sph_ctx *c, int src, int dst, uint32_t *count, const uint32_t *start, uint32_t *nbrs)
{
    SPH_TRY(nnps_need_tables(c));
    DevArray &S = c->arr[src], &D = c->arr[dst];
    if (D.n == 0) return SPH_OK;
    // ghost split: the source's ghosts are its second segment; the destinations are the particles the update binned (the
    // ghosts behind them have no lists: the callers zero their counts)
    if (count && D.n > D.n_binned) HIP_TRY(hipMemsetAsync(count + D.n_binned, 0, (D.n - D.n_binned) * 4, c->stream));
    const bool uh = c->uniform_h && c->use_uniform_h;
    const bool dest_is_src = src == dst;
    const size_t total = S.n + (dest_is_src ? 0 : D.n);
    if (total >= (1ull << 32)) { sph_set_error("too many particles for 32-bit packed indices"); return SPH_ERR_ARG; }
    PackPlan pl = pack_plan(FAM_NBR);
    if (uh) pl.nr = 4;
    const bool was_f32 = c->record_f32 != 0, was_a32 = c->arith_f32 != 0; // lists are exact: fp64 records always
    c->record_f32 = 0; c->arith_f32 = 0;
    c->cur_eosf = false;
    c->cur_eosv = false;
    c->cur_tvff = false;
    c->cur_elu = false;
    c->cur_umass = false;
    c->cur_nrec = pl.nr;
    int rc = c->posh.reserve((total + 64) * sizeof(double) * pl.nr);
    if (rc == SPH_OK) rc = c->aux.reserve(64);
    if (rc == SPH_OK) rc = c->fposb.reserve((total + 64) * sizeof(float4));
    for (auto &pc : c->pack_cache) pc.epoch = 0;
    if (rc == SPH_OK) rc = pack_array(c, src, 0, pl, FAM_NBR, 1u, false, true, 0);
    if (rc == SPH_OK && S.g_n > 0) rc = pack_array(c, src, 0, pl, FAM_NBR, 1u, false, true, 1);
    if (rc == SPH_OK && !dest_is_src) rc = pack_array(c, dst, S.n, pl, FAM_NBR, 1u, true, true, 0);
    c->record_f32 = was_f32; c->arith_f32 = was_a32;
    if (rc != SPH_OK) return rc;
    sph_kernel K;
    K.kind = 1; K.dim = c->dim; K.fac = 1.0; K.radius_scale = c->radius_scale; K.deltap = 0.0;
    PairArgs<FamNbr> a;
    memset(&a, 0, sizeof a);
    fill_common(c, a, &K, 0.0);
    a.ablate = 0; a.dbg = nullptr;
    a.nsrc = 1;
    a.src[0] = {S.cell_start.as<uint32_t>(), 0u, 1u, S.fine_start.as<uint32_t>(), 0.0, 0u};
    if (S.g_n > 0) a.src[a.nsrc++] = {S.g_cell_start.as<uint32_t>(), (uint32_t)S.n_binned, 1u, S.g_fine_start.as<uint32_t>(), 0.0, 1u};
    a.d_off = dest_is_src ? 0u : (uint32_t)S.n;
    a.nd = (uint32_t)D.n_binned;
    a.d_keys = D.keys_sorted.as<uint32_t>(); a.d_fkeys = D.fkeys_sorted.as<uint32_t>(); a.d_perm = D.perm.as<uint32_t>();
    set_tile_order(c, a, D);
    a.d_start = 0; a.d_stop = (uint32_t)D.n; a.dflags = 1u;
    a.p.count = count; a.p.start = start; a.p.nbrs = nbrs;
    dim3 g2(div_up(4 * div_up(a.nd, 256), WPB)), b2(64 * WPB);
    if (uh) hipLaunchKernelGGL((k_pair_wave<FamNbr, 1, true, false, 1>), g2, b2, 0, c->stream, a);
    else hipLaunchKernelGGL((k_pair_wave<FamNbr, 1, false, false, 1>), g2, b2, 0, c->stream, a);
    return SPH_OK;
}

// ---------------------------------------------------------------------------
// A WCSPH group over several arrays as ONE launch on the merged order (sph_ctx::merged, FamWCSPHM_T): applicable
// when the group is nothing but Continuity / Momentum / XSPH equations whose (destination, source) flag matrix is a
// two-class table the library has a kernel for, over exactly the arrays of the grid, under the conditions of the
// EOS-fused uniform-mass records (uniform h, sph_group.src_eos, gamma = 7, no tensile correction, one mass per class).
// *done = false: not applicable, the caller takes the per-destination path.
// ---------------------------------------------------------------------------
static int eval_group_merged(sph_ctx *c, const sph_kernel *K, const sph_group *g, double t, bool *done)
{
    *done = false;
    if (!c->merged_valid || !c->merge_arrays || !c->nnps_valid || c->pair_variant != 6 || c->ablate || c->count_iters) return SPH_OK;
    if (c->merge_blocked || !c->xflag.ptr) return SPH_OK; // a non-positive density was seen: the sign of rho cannot carry the class
    if (g->phase != 0 || c->ghosts_binned) return SPH_OK; // ghost segments: the per-destination path reads them as extra sources
    if (!(g->src_eos == 1 && c->eos_fuse && c->mass_fuse && c->const_flags &&
          tait_gk(g->eos_par[2]) == 3 && g->eos_par[0] > 0.0 && !c->record_f32 && !c->wcsph_nr)) // (gamma = 7: the merged decoders fold it)
        return SPH_OK;
    const bool vh = !(c->uniform_h && c->use_uniform_h); // variable h: FamWCSPHMV_T on records that carry h (round 5)
    const int na = c->narrays;
    uint32_t F[SPH_MAX_ARRAYS][SPH_MAX_ARRAYS] = {};
    const sph_equation *me = nullptr, *xe = nullptr;
    for (int i = 0; i < g->neq; i++) {
        const sph_equation &e = g->eqs[i];
        uint32_t flag;
        if (e.kind == SPH_EQ_CONTINUITY) flag = F_CONT;
        else if (e.kind == SPH_EQ_MOMENTUM) flag = F_MOM;
        else if (e.kind == SPH_EQ_XSPH) flag = F_XSPH;
        else return SPH_OK;
        if (e.nsrc <= 0 || e.dest < 0 || e.dest >= SPH_MAX_ARRAYS || !c->arr[e.dest].used || c->arr[e.dest].nnps_slot < 0) return SPH_OK;
        if (e.kind == SPH_EQ_MOMENTUM) {
            if (e.par[6] != 0.0) return SPH_OK; // tensile correction reads p of the neighbours
            if (me && memcmp(me->par, e.par, 7 * sizeof(double)) != 0) return SPH_OK; // one parameter set per launch
            me = &e;
        }
        if (e.kind == SPH_EQ_XSPH) {
            if (xe && xe->par[0] != e.par[0]) return SPH_OK;
            xe = &e;
        }
        const int ds = c->arr[e.dest].nnps_slot;
        for (int k = 0; k < e.nsrc; k++) {
            const int sid = e.src[k];
            if (sid < 0 || sid >= SPH_MAX_ARRAYS || !c->arr[sid].used || c->arr[sid].nnps_slot < 0) return SPH_OK;
            const int ss = c->arr[sid].nnps_slot;
            if (F[ds][ss] & flag) return SPH_OK; // the same equation twice: the general path reports it
            F[ds][ss] |= flag;
        }
    }
    // every array of the grid is in the record stream: it must take part (an array without equations would act
    // as a source of whatever its class says)
    for (int a = 0; a < na; a++) {
        bool used = false;
        for (int b = 0; b < na; b++) used |= F[a][b] != 0 || F[b][a] != 0;
        if (!used) return SPH_OK;
    }
    // two classes: arrays a, b are alike when they see and are seen by every array -- each other included -- the same way
    auto alike = [&](int a, int b) {
        if (F[a][a] != F[b][b] || F[a][b] != F[b][a] || F[a][a] != F[a][b]) return false;
        for (int x = 0; x < na; x++)
            if (x != a && x != b && (F[a][x] != F[b][x] || F[x][a] != F[x][b])) return false;
        return true;
    };
    int cls[SPH_MAX_ARRAYS], rep[2] = {0, -1};
    for (int a = 0; a < na; a++) {
        if (a == rep[0] || alike(a, rep[0])) cls[a] = 0;
        else if (rep[1] < 0 || alike(a, rep[1])) { cls[a] = 1; if (rep[1] < 0) rep[1] = a; }
        else return SPH_OK;
    }
    if (rep[1] < 0) return SPH_OK; // one class of several arrays: the per-destination path has constant flags already
    auto tab = [&](int i, int j) {
        // an entry of the class table from two representatives (two DIFFERENT members when the class has them)
        for (int a = 0; a < na; a++) for (int b = 0; b < na; b++)
            if (cls[a] == i && cls[b] == j && (a != b || i != j)) return F[a][b];
        return F[rep[i]][rep[j]];
    };
    uint32_t T4[2][2] = {{tab(0, 0), tab(0, 1)}, {tab(1, 0), tab(1, 1)}};
    for (int a = 0; a < na; a++) for (int b = 0; b < na; b++) if (F[a][b] != T4[cls[a]][cls[b]]) return SPH_OK;
    if (T4[1][1] > T4[0][0]) { // class 0 = the one with the richer self-interaction (the fluids)
        for (int a = 0; a < na; a++) cls[a] ^= 1;
        std::swap(T4[0][0], T4[1][1]); std::swap(T4[0][1], T4[1][0]);
    }
    const uint32_t ct = WCSPHM_CT(T4[0][0], T4[0][1], T4[1][0], T4[1][1]);
    if (ct != FamWCSPHM_T<double>::CF0) return SPH_OK; // the class table the library has a kernel for (WCSPHScheme's)
    // one mass per class, seen by the last neighbour update
    c->want_mrange = true;
    double mu[2] = {0.0, 0.0};
    bool have[2] = {false, false};
    for (int a = 0; a < na; a++) {
        const DevArray &A = c->arr[c->ids[a]];
        if (A.n == 0) continue;
        if (!A.m_known) return SPH_OK;
        if (have[cls[a]] && mu[cls[a]] != A.m_value) return SPH_OK;
        mu[cls[a]] = A.m_value; have[cls[a]] = true;
    }
    DevArray &M = c->merged;
    if (M.n == 0) { *done = true; return SPH_OK; }

    // outputs and properties
    static const int outp[9] = {SPH_ARHO, SPH_AU, SPH_AV, SPH_AW, SPH_AX, SPH_AY, SPH_AZ, SPH_DT_CFL, SPH_DT_FORCE};
    static const int inp[9] = {SPH_X, SPH_Y, SPH_Z, SPH_U, SPH_V, SPH_W, SPH_RHO, SPH_P, SPH_H};
    const bool f32 = c->arith_f32 != 0;
    PackMArgs pa;
    memset(&pa, 0, sizeof pa);
    for (int a = 0; a < na; a++) {
        const int id = c->ids[a];
        DevArray &A = c->arr[id];
        for (int k = 0; k < 9; k++) {
            if (A.n && !A.prop[inp[k]]) return need_prop(c, id, inp[k], "pair loop");
            pa.prop[k][a] = A.prop[inp[k]];
        }
        if (cls[a]) pa.cls |= 1u << a;
    }
    c->cur_eosf = !vh; c->cur_umass = !vh; c->cur_tvff = false; c->cur_elu = false; c->cur_eosv = vh;
    c->cur_nrec = 8;
    SPH_TRY(c->posh.reserve((M.n + 64) * (f32 ? 32 : 64)));
    SPH_TRY(c->aux.reserve(64));
    SPH_TRY(c->fposb.reserve((M.n + 64) * sizeof(float4)));
    for (auto &pc : c->pack_cache) pc.epoch = 0; // the shared slots of the per-destination path are overwritten
    {
        ScopedTimer tm(c, T_PACK);
        pa.perm = M.perm.as<uint32_t>(); pa.slot = M.slot8.as<uint8_t>();
        pa.n = M.n; pa.off = 0;
        pa.rec = c->posh.as<double>(); pa.fpos = c->fposb.as<float4>();
        pa.f32 = f32 ? 1 : 0; pa.lds_np = f32 ? 2 : 4;
        for (int k = 0; k < 3; k++) pa.gmin[k] = c->xmin[k];
        pa.vh = vh ? 1 : 0;
        pa.hr = vh ? c->radius_scale : c->radius_scale * c->h_uniform;
        pa.rho_flag = c->xflag.as<uint32_t>();
        const dim3 pg(div_up(M.n, 256)), pb(256);
        const size_t plds = (size_t)pa.lds_np * 256 * 16;
        if (f32 && vh) hipLaunchKernelGGL((k_pack_merged<true, true>), pg, pb, plds, c->stream, pa);
        else if (f32) hipLaunchKernelGGL((k_pack_merged<true, false>), pg, pb, plds, c->stream, pa);
        else if (vh) hipLaunchKernelGGL((k_pack_merged<false, true>), pg, pb, plds, c->stream, pa);
        else hipLaunchKernelGGL((k_pack_merged<false, false>), pg, pb, plds, c->stream, pa);
    }
    ScopedTimer tm(c, T_PAIR);
    ScopedTimer tmf(c, T_PAIR_FAM + FAM_WCSPH);
    c->timers[T_N_EOSF].count++; c->timers[T_N_UMASS].count++; c->timers[T_N_MERGED].count++;
    auto run = [&](auto tag) -> int {
        typedef decltype(tag) Fm;
        PairArgs<Fm> a;
        memset(&a, 0, sizeof a);
        fill_common(c, a, K, t);
        a.nsrc = 1;
        a.src[0] = {nullptr, 0u, ct, M.fine_start.as<uint32_t>(), mu[0]};
        a.d_off = 0; a.nd = (uint32_t)M.n;
        a.d_mu = mu[0];
        a.d_keys = M.keys_sorted.as<uint32_t>(); a.d_fkeys = M.fkeys_sorted.as<uint32_t>(); a.d_perm = M.perm.as<uint32_t>();
        a.d_slot = M.slot8.as<uint8_t>();
        set_tile_order(c, a, M);
        a.d_start = 0; a.d_stop = 0xffffffffu; a.dflags = ct;
        a.e_rho01 = 1.0 / g->eos_par[0]; a.e_c0 = g->eos_par[1];
        a.e_B = g->eos_par[0] * g->eos_par[1] * g->eos_par[1] / g->eos_par[2];
        a.e_gk = tait_gk(g->eos_par[2]);
        a.e_p0 = g->eos_par[3];
        if (me) { a.p.c0 = me->par[0]; a.p.alpha = me->par[1]; a.p.beta = me->par[2]; a.p.gx = me->par[3]; a.p.gy = me->par[4]; a.p.gz = me->par[5]; }
        if (xe) a.p.eps = xe->par[0];
        a.p.mu1 = mu[1];
        for (int s = 0; s < na; s++) {
            const int id = c->ids[s];
            DevArray &A = c->arr[id];
            uint32_t row = 0;
            for (int b = 0; b < na; b++) row |= F[s][b];
            // NP_DEST / D_START_IDX (acceleration_eval_cython_helper.py:259-280), per destination array
            size_t start = g->start_idx > 0 ? (size_t)g->start_idx : 0;
            size_t stop = g->stop_idx >= 0 ? (size_t)g->stop_idx : (g->real ? A.n_real : A.n);
            if (stop > A.n) stop = A.n;
            if (!row || stop < start) start = stop = 0;
            a.p.rng[s][0] = (uint32_t)start; a.p.rng[s][1] = (uint32_t)stop;
            if (row & F_CONT) { SPH_TRY(ensure_out(c, id, {SPH_ARHO})); }
            if (row & F_MOM) { SPH_TRY(ensure_out(c, id, {SPH_AU, SPH_AV, SPH_AW, SPH_DT_CFL, SPH_DT_FORCE})); }
            if (row & F_XSPH) { SPH_TRY(ensure_out(c, id, {SPH_AX, SPH_AY, SPH_AZ})); }
            for (int k = 0; k < 9; k++) a.p.out[s][k] = A.prop[outp[k]];
        }
        {   // every destination range is [0, n_real) of its array (or empty): the wave tiles are the real particles' ones
            bool all_real = true;
            for (int s = 0; s < na; s++) {
                const DevArray &A = c->arr[c->ids[s]];
                all_real &= (a.p.rng[s][0] == 0 && a.p.rng[s][1] == (uint32_t)A.n_real) || A.n == 0;
            }
            if (all_real) use_dest_list(c, a, M);
        }
        constexpr bool FP32 = sizeof(typename Fm::Real) == 4;
        dim3 g2(div_up(4 * div_up(a.nd, 256), WPB)), b2(64 * WPB);
        constexpr bool UHM = !std::is_base_of<FamWCSPHMV_T<typename Fm::Real>, Fm>::value; // uniform h unless the family carries h
#define LAUNCHM(KV) hipLaunchKernelGGL((k_pair_wave<Fm, KV, UHM, FP32, Fm::CF0>), g2, b2, (size_t)c->lds_pad, c->stream, a)
        switch (K->kind) {
        case 1: LAUNCHM(1); break;
        case 2: LAUNCHM(2); break;
        case 3: LAUNCHM(3); break;
        case 4: LAUNCHM(4); break;
        }
#undef LAUNCHM
        return SPH_OK;
    };
    if (vh) SPH_TRY(f32 ? run(FamWCSPHMV_T<float>()) : run(FamWCSPHMV_T<double>()));
    else SPH_TRY(f32 ? run(FamWCSPHM_T<float>()) : run(FamWCSPHM_T<double>()));
    HIP_TRY(hipGetLastError());
    *done = true;
    return SPH_OK;
}

extern "C" int sph_eval_group(sph_ctx *c, const sph_kernel *K, const sph_group *g, double t, double dt)
{
    if (!c || !K || !g) { sph_set_error("sph_eval_group: NULL argument"); return SPH_ERR_ARG; }
    c->cur_dt = dt;
    if (g->neq > SPH_MAX_EQS) { sph_set_error("sph_eval_group: too many equations"); return SPH_ERR_ARG; }
    if (K->kind < 1 || K->kind > 4) { sph_set_error("sph_eval_group: unknown kernel kind %d", K->kind); return SPH_ERR_UNSUPPORTED; }
    HIP_TRY(hipSetDevice(c->device));
    if (!c->pack_group) c->pack_epoch++; // packed records are reused between the destinations of ONE group only (option pack_group)
    {
        bool done = false;
        SPH_TRY(eval_group_merged(c, K, g, t, &done));
        if (done) return SPH_OK;
    }

    // destinations in first-appearance order (acceleration_eval.py:126-131)
    int dests[SPH_MAX_ARRAYS], ndest = 0;
    for (int i = 0; i < g->neq; i++) {
        int d = g->eqs[i].dest;
        if (d < 0 || d >= SPH_MAX_ARRAYS || !c->arr[d].used) { sph_set_error("equation %d: bad dest array %d", i, d); return SPH_ERR_ARG; }
        bool seen = false;
        for (int j = 0; j < ndest; j++) seen |= dests[j] == d;
        if (!seen) dests[ndest++] = d;
    }
    const int phase = g->phase;
    if (phase < 0 || phase > 2 || (phase != 0 && ndest != 1)) {
        sph_set_error("sph_eval_group: phase %d needs a group with one destination array", phase);
        return SPH_ERR_ARG;
    }
    for (int di = 0; di < ndest; di++) {
        const int dst = dests[di];
        DevArray &D = c->arr[dst];
        // NP_DEST / D_START_IDX (acceleration_eval_cython_helper.py:259-280)
        size_t start = g->start_idx > 0 ? (size_t)g->start_idx : 0;
        size_t stop = g->stop_idx >= 0 ? (size_t)g->stop_idx : (g->real ? D.n_real : D.n);
        if (stop > D.n) stop = D.n;

        // 1. equations without sources run first (mako :50-58); phase 2: over the particles that arrived since phase 1
        const size_t ns_start = phase == 2 ? std::max(start, D.n_binned) : start;
        for (int i = 0; i < g->neq; i++) {
            const sph_equation &e = g->eqs[i];
            if (e.dest != dst || e.nsrc != 0) continue;
            if (!is_nosrc_kind(e.kind)) { sph_set_error("equation kind %d needs sources", e.kind); return SPH_ERR_UNSUPPORTED; }
            SPH_TRY(run_nosrc(c, e, ns_start, stop));
            c->pack_cache[dst].epoch = 0; // its properties changed: records packed for an earlier destination are stale
        }

        // 2. sources in first-appearance order with their equation flags (acceleration_eval.py:136-151)
        int srcs[SPH_MAX_ARRAYS], nsrcs = 0;
        uint32_t sflags[SPH_MAX_ARRAYS] = {};
        int fam = FAM_NONE;
        const sph_equation *eq_of[32] = {};
        bool elastic = false;
        for (int i = 0; i < g->neq; i++) {
            const sph_equation &e = g->eqs[i];
            if (e.dest == dst && e.nsrc > 0 &&
                (e.kind == SPH_EQ_MOMENTUM_WITH_STRESS || e.kind == SPH_EQ_MONAGHAN_ART_VISCOSITY))
                elastic = true;
        }
        for (int i = 0; i < g->neq; i++) {
            const sph_equation &e = g->eqs[i];
            if (e.dest != dst || e.nsrc == 0) continue;
            if (e.kind < 0 || e.kind >= 32) { sph_set_error("bad equation kind %d", e.kind); return SPH_ERR_ARG; }
            uint32_t flag = 0;
            int f = eq_family(e.kind, &flag, elastic);
            if (f == FAM_NONE) { sph_set_error("equation kind %d has no hand-written pair kernel", e.kind); return SPH_ERR_UNSUPPORTED; }
            if (fam != FAM_NONE && f != fam) {
                sph_set_error("group mixes equation families on destination %d (kinds are fused per family)", dst);
                return SPH_ERR_UNSUPPORTED;
            }
            fam = f;
            if (eq_of[e.kind]) {
                sph_set_error("equation kind %d appears twice for destination %d in one group", e.kind, dst);
                return SPH_ERR_UNSUPPORTED;
            }
            eq_of[e.kind] = &e;
            for (int k = 0; k < e.nsrc; k++) {
                int s = e.src[k], pos = -1;
                if (s < 0 || s >= SPH_MAX_ARRAYS || !c->arr[s].used) { sph_set_error("bad source array %d", s); return SPH_ERR_ARG; }
                for (int j = 0; j < nsrcs; j++) if (srcs[j] == s) pos = j;
                if (pos < 0) { pos = nsrcs; srcs[nsrcs++] = s; }
                sflags[pos] |= flag;
            }
        }
        if (fam == FAM_NONE) continue;
        if (!c->nnps_valid) { sph_set_error("sph_eval_group: neighbour grid is stale; call sph_nnps_update"); return SPH_ERR_STATE; }
        if (D.nnps_slot < 0) { sph_set_error("destination array %d is not part of the neighbour grid", dst); return SPH_ERR_STATE; }
        SPH_TRY(nnps_need_tables(c)); // the per-destination path reads every array's own cell order
        uint32_t dflags = 0;
        for (int j = 0; j < nsrcs; j++) {
            dflags |= sflags[j];
            if (c->arr[srcs[j]].nnps_slot < 0) { sph_set_error("source array %d is not part of the neighbour grid", srcs[j]); return SPH_ERR_STATE; }
        }
        if (fam == FAM_WCSPH && eq_of[SPH_EQ_MOMENTUM] && eq_of[SPH_EQ_MOMENTUM]->par[6] != 0.0) {
            dflags |= F_TENSILE;
            for (int j = 0; j < nsrcs; j++) if (sflags[j] & F_MOM) sflags[j] |= F_TENSILE;
        }

        // 3. pack destination + sources into cell order
        size_t total = 0, off_of[SPH_MAX_ARRAYS];
        bool dest_is_src = false;
        for (int j = 0; j < nsrcs; j++) { off_of[j] = total; total += c->arr[srcs[j]].n; dest_is_src |= srcs[j] == dst; }
        size_t d_off = total;
        if (!dest_is_src) total += D.n;
        else for (int j = 0; j < nsrcs; j++) if (srcs[j] == dst) d_off = off_of[j];
        // WCSPH groups with several destinations (a dam break: fluid, boundary, obstacle) read the
        // SAME source records, and nothing the family writes (accelerations) is part of a record:
        // one buffer slot per neighbour-grid array, each array packed once per group instead of
        // once per destination that reads it (dam break C2: 7 -> 3 k_pack launches per evaluation).
        // The host brackets the per-destination calls of one group with option pack_group 1 / 0.
        const bool share = fam == FAM_WCSPH && c->pair_variant >= 3 && (ndest > 1 || c->pack_group);
        if (share) {
            size_t goff[SPH_MAX_ARRAYS] = {};
            total = 0;
            for (int a = 0; a < c->narrays; a++) { goff[c->ids[a]] = total; total += c->arr[c->ids[a]].n; }
            for (int j = 0; j < nsrcs; j++) off_of[j] = goff[srcs[j]];
            d_off = goff[dst];
        }
        if (total >= (1ull << 32)) { sph_set_error("too many particles for 32-bit packed indices"); return SPH_ERR_ARG; }
        // ghost split (sph_nnps_update_ghosts): arrays whose ghosts were binned after the real particles carry them as a
        // second record segment and a second source; the destinations are the particles the update binned
        bool any_ghosts = false;
        for (int j = 0; j < nsrcs; j++) any_ghosts |= c->arr[srcs[j]].g_n > 0;
        any_ghosts |= D.g_n > 0;
        if (phase != 0 && !(nsrcs == 1 && srcs[0] == dst && !share)) {
            sph_set_error("sph_eval_group: phases need a group over ONE array (destination == its only source)");
            return SPH_ERR_UNSUPPORTED;
        }
        if ((any_ghosts || phase != 0) && c->pair_variant != 6) {
            sph_set_error("ghost segments (sph_nnps_update_ghosts) need pair_variant 6");
            return SPH_ERR_UNSUPPORTED;
        }
        if (!dest_is_src && D.n_binned != D.n) {
            // a destination-only array is read through its own records only: its ghosts are no destinations
        }
        PackPlan pl = pack_plan(fam);
        // compact 80-B WCSPH records when neither h nor p of a neighbour is read
        if (c->pair_variant >= 3 && fam == FAM_WCSPH && c->uniform_h && c->use_uniform_h && !(dflags & F_TENSILE)) pl.nr = c->wcsph_nr ? (int)c->wcsph_nr : 10;
        if (c->pair_variant >= 3 && fam == FAM_DENSITY && c->uniform_h && c->use_uniform_h) pl.nr = 4;
        if (c->pair_variant >= 3 && fam == FAM_TVF && c->uniform_h && c->use_uniform_h) pl.nr = (dflags & F_TAV) ? 14 : 12;
        if (c->pair_variant >= 3 && (c->record_f32 || c->arith_f32)) pl.nr = (4 + pl.na + 3) & ~3; // floats
        // The caller promised (sph_group.src_eos) that p, cs of every array read here are the Tait EOS of its
        // rho: with gamma = 7 (powers by multiplication), uniform h and no tensile correction the pair kernel
        // recomputes them and the records shrink to 64 bytes (32 in fp32).
        const bool eosf = g->src_eos == 1 && c->eos_fuse && fam == FAM_WCSPH && c->pair_variant == 6 && c->uniform_h &&
                          c->use_uniform_h && !(dflags & F_TENSILE) && tait_gk(g->eos_par[2]) >= 0 &&
                          g->eos_par[0] > 0.0 && !c->record_f32 && !c->wcsph_nr;
        if (eosf) pl.nr = 8; // doubles, or floats with arith_f32
        // ... and when every array read here had ONE mass at the last neighbour update, the record's mass slot
        // carries p / rho^2 and most of the per-record EOS goes as well (FamWCSPHE_T<T, true>)
        // (only with the equation flags as a compile-time constant: the run-time-flag kernel of a dam break's three
        // sources is at its scalar-register limit and 1-3 % slower with a mass per source)
        bool umass = eosf && c->mass_fuse && c->const_flags;
        for (int j = 0; j < nsrcs && umass; j++) umass = sflags[j] == FamWCSPH::CF0;
        if (umass) c->want_mrange = true; // the reduction of the neighbour updates from now on includes m (8 B per particle)
        for (int j = 0; j < nsrcs && umass; j++) umass = c->arr[srcs[j]].m_known;
        // ... and with VARIABLE h the same promise plus one mass per source array gives 64-byte records [x y z h u v w rho]
        bool eosv = !eosf && g->src_eos == 1 && c->eos_fuse && c->mass_fuse && fam == FAM_WCSPH && c->pair_variant == 6 &&
                    !(c->uniform_h && c->use_uniform_h) && !(dflags & F_TENSILE) && tait_gk(g->eos_par[2]) == 3 && g->eos_par[0] > 0.0 &&
                    !c->record_f32 && !c->wcsph_nr;
        if (eosv) c->want_mrange = true;
        for (int j = 0; j < nsrcs && eosv; j++) eosv = c->arr[srcs[j]].m_known;
        if (eosv) pl.nr = 8;
        c->cur_eosv = eosv;
        // TVF force pass after StateEquation + density summation (sph_group.src_eos = 2): p and V leave the records
        // when every array read here has ONE mass (V = rho / m)
        bool tvff = g->src_eos == 2 && c->eos_fuse && c->mass_fuse && fam == FAM_TVF && c->pair_variant == 6 && c->uniform_h &&
                    c->use_uniform_h && !c->record_f32 && g->eos_par[1] != 0.0; // (the artificial viscosity's m_j is the source's one mass)
        if (tvff) c->want_mrange = true;
        for (int j = 0; j < nsrcs && tvff; j++) tvff = c->arr[srcs[j]].m_known;
        if (tvff && !dest_is_src) tvff = D.m_known;
        if (tvff) pl.nr = c->arith_f32 ? ((dflags & F_TAS) ? 12 : 8) : ((dflags & F_TAS) ? 10 : 8);
        // elastic rates: uniform h and ONE mass per source array -> neither travels in the records
        bool elu = fam == FAM_ELASTIC && c->mass_fuse && c->pair_variant == 6 && c->uniform_h && c->use_uniform_h && !c->record_f32;
        if (elu) c->want_mrange = true;
        for (int j = 0; j < nsrcs && elu; j++) elu = c->arr[srcs[j]].m_known;
        if (elu) pl.nr = 20; // doubles, or floats with arith_f32
        c->cur_elu = elu;
        c->cur_tvff = tvff;
        c->cur_umass = umass;
        c->cur_eosf = eosf;
        c->cur_nrec = pl.nr;
        const void *rec_was = c->posh.ptr, *fpos_was = c->fposb.ptr;
        // phase 1 leaves room for the ghosts phase 2 packs behind the real particles' records (the array's capacity bounds
        // them); phase 2 keeps what phase 1 packed should the buffers have to grow after all
        const size_t total_res = phase == 1 ? std::max(total, D.cap) : total;
        SPH_TRY(c->posh.reserve((total_res + 64) * sizeof(double) * (c->pair_variant >= 2 ? pl.nr : 4), phase == 2, c->stream));
        SPH_TRY(c->aux.reserve((total_res + 64) * sizeof(double) * pl.na));
        if (c->pair_variant >= 3) SPH_TRY(c->fposb.reserve((total_res + 64) * sizeof(float4), phase == 2, c->stream));
        if (c->posh.ptr != rec_was || c->fposb.ptr != fpos_was) // the buffers moved: nothing packed earlier is there
            for (auto &pc : c->pack_cache) pc.epoch = 0;
        {
            ScopedTimer tm(c, T_PACK);
            // a source must hold what ITS equations read; the destination's own record what all of them read
            const int sig = pl.nr * 32 + (c->record_f32 ? 1 : 0) + (c->arith_f32 ? 2 : 0) + (eosf ? 4 : 0) + (umass ? 8 : 0) + (tvff || elu || eosv ? 16 : 0);
            // The shared slots live in the same buffers every other unit packs into from offset 0:
            // a unit that does not share (another family, a single-destination call), or one whose
            // record layout differs from what the cache holds, overwrites them -- nothing cached survives.
            // a split evaluation: the second half must read the layout the first half packed -- when what is known of the
            // masses changed in between (a ghost with another mass: sph_nnps_update_ghosts), it packs both segments again
            if (phase == 1) c->phase_sig = sig;
            const bool repack = phase == 2 && c->phase_sig != sig;
            bool keep = share;
            if (share)
                for (auto &pc : c->pack_cache)
                    if (pc.epoch == c->pack_epoch && (pc.fam != fam || pc.sig != sig)) keep = false;
            if (!keep)
                for (auto &pc : c->pack_cache) pc.epoch = 0;
            auto pack_once = [&](int id, size_t off, uint32_t fl, bool dest_only) -> int {
                PackCache &pc = c->pack_cache[id];
                const bool hit = share && pc.epoch == c->pack_epoch && pc.fam == fam && pc.sig == sig;
                // phase 1: the particles present (the real ones); phase 2: the ghosts that arrived since; else both segments
                if (phase != 2 || repack) SPH_TRY(pack_array(c, id, off, pl, fam, fl, dest_only, !hit, 0)); // a hit still validates
                if (phase != 1 && c->arr[id].g_n > 0) SPH_TRY(pack_array(c, id, off, pl, fam, fl, dest_only, !hit, 1));
                pc.epoch = share ? c->pack_epoch : 0; pc.fam = fam; pc.sig = sig;
                return SPH_OK;
            };
            for (int j = 0; j < nsrcs; j++) SPH_TRY(pack_once(srcs[j], off_of[j], srcs[j] == dst ? dflags : sflags[j], false));
            if (!dest_is_src) SPH_TRY(pack_once(dst, d_off, dflags, true));
        }

        // Neighbour-list reuse between the pair passes of one evaluation (sph_group.nl_mode): one source,
        // the wave-tile kernel, same grid.  A pass that keeps its lists (1) records what they belong to; a pass
        // that wants them (2) gets them only if that record matches -- otherwise it runs its own phase 1.
        int nl_mode = 0;
        if (c->nl_reuse && !c->row_lds && c->pair_variant == 6 && nsrcs == 1 && g->nl_mode && !c->ablate && !any_ghosts && phase == 0) {
            const size_t n_wt = (size_t)4 * div_up(D.n, 256);
            if (g->nl_mode == 1) {
                SPH_TRY(c->nlbuf.reserve(n_wt * NLW * sizeof(uint32_t)));
                c->nl.valid = true; c->nl.epoch = c->nnps_epoch; c->nl.dst = dst; c->nl.src = srcs[0];
                c->nl.start = start; c->nl.stop = stop; c->nl.nd = D.n;
                nl_mode = 1;
            } else if (g->nl_mode == 2 && c->nl.valid && c->nl.epoch == c->nnps_epoch && c->nl.dst == dst && c->nl.src == srcs[0] &&
                       c->nl.nd == D.n && c->nl.start <= start && c->nl.stop >= stop && c->nlbuf.bytes >= n_wt * NLW * sizeof(uint32_t)) {
                nl_mode = 2;
            }
        } else if (g->nl_mode != 2) {
            c->nl.valid = c->nl.valid && !(g->nl_mode == 1); // a keeping pass that could not keep: nothing to reuse
        }

        // Split evaluations (sph_group.phase): by default the first half only prepares (equations without sources, records
        // of the particles present) and the second half launches EVERY wave tile once the ghosts are in -- the ghost
        // segments guarded per wavefront; option split_pair 1: the interior tiles in the first half, the face tiles in
        // the second (hides a transfer longer than the neighbour update + packing; costs a second, sparse launch)
        if (phase == 1 && !c->split_pair) continue;
        // 4. fused pair kernel, in the arithmetic type of the context (fp64, or fp32 with option arith_f32)
        ScopedTimer tm(c, T_PAIR);
        ScopedTimer tmf(c, T_PAIR_FAM + fam);
        if (phase == 2) c->timers[T_N_PHASE2].count++;
        if (eosf || tvff || eosv) c->timers[T_N_EOSF].count++;
        if (umass || tvff || elu || eosv) c->timers[T_N_UMASS].count++;
        if (nl_mode == 1) c->timers[T_N_NLKEEP].count++;
        if (nl_mode == 2) c->timers[T_N_NLREUSE].count++;
        // the part of the launch arguments every family shares
        {
            int nseg = nsrcs;
            for (int j = 0; j < nsrcs; j++) nseg += c->arr[srcs[j]].g_n > 0 && phase != 1;
            if (nseg > SPH_MAX_ARRAYS) { sph_set_error("too many source segments (%d arrays with ghost segments)", nsrcs); return SPH_ERR_UNSUPPORTED; }
        }
        auto common = [&](auto &a) {
            fill_common(c, a, K, t);
            a.nsrc = 0;
            for (int j = 0; j < nsrcs; j++) {
                const DevArray &S = c->arr[srcs[j]];
                a.src[a.nsrc++] = {S.cell_start.as<uint32_t>(), (uint32_t)off_of[j], sflags[j], S.fine_start.as<uint32_t>(), S.m_value, 0u};
            }
            for (int j = 0; j < nsrcs && phase != 1; j++) { // the ghost segments, after every array's first segment
                const DevArray &S = c->arr[srcs[j]];
                if (S.g_n == 0) continue;
                a.src[a.nsrc++] = {S.g_cell_start.as<uint32_t>(), (uint32_t)(off_of[j] + S.n_binned), sflags[j],
                                   S.g_fine_start.as<uint32_t>(), S.m_value, 1u};
            }
            a.face_mode = c->split_pair ? phase : 0;
            a.d_off = (uint32_t)d_off; a.nd = (uint32_t)D.n_binned;
            a.d_mu = D.m_value;
            a.d_keys = D.keys_sorted.as<uint32_t>(); a.d_fkeys = D.fkeys_sorted.as<uint32_t>(); a.d_perm = D.perm.as<uint32_t>();
            set_tile_order(c, a, D);
            a.d_start = (uint32_t)start; a.d_stop = (uint32_t)stop; a.dflags = dflags;
            a.nl = nullptr; a.nl_mode = nl_mode;
            if (nl_mode) a.nl = c->nlbuf.as<uint32_t>();
            // Group.real = True without start / stop: the destinations are the real particles
            if (start == 0 && stop == D.n_real && D.n_binned == D.n && !any_ghosts) use_dest_list(c, a, D);
            if (eosf || eosv) {
                a.e_rho01 = 1.0 / g->eos_par[0]; a.e_c0 = g->eos_par[1];
                a.e_B = g->eos_par[0] * g->eos_par[1] * g->eos_par[1] / g->eos_par[2];
        a.e_gk = tait_gk(g->eos_par[2]);
                a.e_p0 = g->eos_par[3];
            }
            if (tvff) { a.e_p0 = g->eos_par[0]; a.e_rho01 = 1.0 / g->eos_par[1]; a.e_B = g->eos_par[2]; } // StateEquation: p0 rho0 b
        };
        auto run_wcsph = [&](auto tag) -> int {
            typedef decltype(tag) F;
            PairArgs<F> a;
            memset(&a, 0, sizeof a);
            common(a);
            const sph_equation *me = eq_of[SPH_EQ_MOMENTUM], *xe = eq_of[SPH_EQ_XSPH];
            if (me) { a.p.c0 = me->par[0]; a.p.alpha = me->par[1]; a.p.beta = me->par[2]; a.p.gx = me->par[3]; a.p.gy = me->par[4]; a.p.gz = me->par[5]; }
            if (xe) a.p.eps = xe->par[0];
            if (dflags & F_CONT) { SPH_TRY(ensure_out(c, dst, {SPH_ARHO})); a.p.arho = D.prop[SPH_ARHO]; }
            if (dflags & F_MOM) {
                SPH_TRY(ensure_out(c, dst, {SPH_AU, SPH_AV, SPH_AW, SPH_DT_CFL, SPH_DT_FORCE}));
                a.p.au = D.prop[SPH_AU]; a.p.av = D.prop[SPH_AV]; a.p.aw = D.prop[SPH_AW];
                a.p.dt_cfl = D.prop[SPH_DT_CFL]; a.p.dt_force = D.prop[SPH_DT_FORCE];
            }
            if (dflags & F_XSPH) {
                SPH_TRY(ensure_out(c, dst, {SPH_AX, SPH_AY, SPH_AZ}));
                a.p.ax = D.prop[SPH_AX]; a.p.ay = D.prop[SPH_AY]; a.p.az = D.prop[SPH_AZ];
            }
            if constexpr (std::is_same<F, FamWCSPHV_T<typename F::Real>>::value) return launch_pair_fused<F, false>(c, K->kind, a);
            else if constexpr (fam_eosf<F>::value) return launch_pair_fused<F>(c, K->kind, a);
            else return launch_pair<F>(c, K->kind, a);
        };
        auto run_density = [&](auto tag) -> int {
            typedef decltype(tag) F;
            PairArgs<F> a;
            memset(&a, 0, sizeof a);
            common(a);
            SPH_TRY(ensure_out(c, dst, {SPH_RHO}));
            a.p.rho = D.prop[SPH_RHO];
            if (dflags & F_TVFSD) { SPH_TRY(ensure_out(c, dst, {SPH_VOL})); a.p.V = D.prop[SPH_VOL]; }
            if ((dflags & F_SD) && (dflags & F_TVFSD)) { sph_set_error("both SummationDensity flavours on one destination"); return SPH_ERR_UNSUPPORTED; }
            return launch_pair<F>(c, K->kind, a);
        };
        auto run_vgrad = [&](auto tag) -> int {
            typedef decltype(tag) F;
            PairArgs<F> a;
            memset(&a, 0, sizeof a);
            common(a);
            static const int vp[9] = {SPH_V00, SPH_V01, SPH_V02, SPH_V10, SPH_V11, SPH_V12, SPH_V20, SPH_V21, SPH_V22};
            if ((dflags & F_VG2) && (dflags & F_VG3)) { sph_set_error("both VelocityGradient2D and 3D on one destination"); return SPH_ERR_UNSUPPORTED; }
            for (int k = 0; k < 9; k++) {
                const bool used = (dflags & F_VG3) || (k == 0 || k == 1 || k == 3 || k == 4);
                if (used) { SPH_TRY(sph_array_ensure_prop(c, dst, vp[k])); a.p.v[k] = D.prop[vp[k]]; }
            }
            return launch_pair<F>(c, K->kind, a);
        };
        auto run_elastic = [&](auto tag) -> int {
            typedef decltype(tag) F;
            PairArgs<F> a;
            memset(&a, 0, sizeof a);
            common(a);
            const sph_equation *se = eq_of[SPH_EQ_MOMENTUM_WITH_STRESS], *ae = eq_of[SPH_EQ_MONAGHAN_ART_VISCOSITY],
                               *xe = eq_of[SPH_EQ_XSPH];
            if (se) { a.p.wdeltap = se->par[0]; a.p.n = se->par[1]; }
            a.p.tension = nullptr;
            if (elu && c->tension_flag && nsrcs == 1 && c->arr[srcs[0]].tflag_valid && c->arr[srcs[0]].tflag.ptr) {
                a.p.tension = c->arr[srcs[0]].tflag.as<uint32_t>();
                c->timers[T_N_TFLAG].count++;
            }
            if (ae) { a.p.alpha = ae->par[0]; a.p.beta = ae->par[1]; }
            if (xe) a.p.eps = xe->par[0];
            if (dflags & F_ECONT) { SPH_TRY(ensure_out(c, dst, {SPH_ARHO})); a.p.arho = D.prop[SPH_ARHO]; }
            if (dflags & (F_ESTRESS | F_EAV)) {
                SPH_TRY(ensure_out(c, dst, {SPH_AU, SPH_AV, SPH_AW}));
                a.p.au = D.prop[SPH_AU]; a.p.av = D.prop[SPH_AV]; a.p.aw = D.prop[SPH_AW];
            }
            if (dflags & F_EXSPH) {
                SPH_TRY(ensure_out(c, dst, {SPH_AX, SPH_AY, SPH_AZ}));
                a.p.ax = D.prop[SPH_AX]; a.p.ay = D.prop[SPH_AY]; a.p.az = D.prop[SPH_AZ];
            }
            if constexpr (fam_eosf<F>::value) return launch_pair_fused<F>(c, K->kind, a);
            else return launch_pair<F>(c, K->kind, a);
        };
        auto run_tvf = [&](auto tag) -> int {
            typedef decltype(tag) F;
            PairArgs<F> a;
            memset(&a, 0, sizeof a);
            common(a);
            const sph_equation *pe = eq_of[SPH_EQ_TVF_MOM_PRESSURE], *ve = eq_of[SPH_EQ_TVF_MOM_VISCOSITY],
                               *ae = eq_of[SPH_EQ_TVF_MOM_ART_VISCOSITY];
            if (pe) { a.p.pb = pe->par[0]; a.p.gx = pe->par[1]; a.p.gy = pe->par[2]; a.p.gz = pe->par[3]; a.p.tdamp = pe->par[4]; }
            if (ve) a.p.nu = ve->par[0];
            if (ae) { a.p.c0 = ae->par[0]; a.p.alpha = ae->par[1]; }
            SPH_TRY(ensure_out(c, dst, {SPH_AU, SPH_AV, SPH_AW}));
            a.p.au = D.prop[SPH_AU]; a.p.av = D.prop[SPH_AV]; a.p.aw = D.prop[SPH_AW];
            a.p.m = D.prop[SPH_M];
            if (dflags & F_TP) {
                SPH_TRY(ensure_out(c, dst, {SPH_AUHAT, SPH_AVHAT, SPH_AWHAT}));
                a.p.auhat = D.prop[SPH_AUHAT]; a.p.avhat = D.prop[SPH_AVHAT]; a.p.awhat = D.prop[SPH_AWHAT];
            }
            if constexpr (fam_eosf<F>::value) return launch_pair_fused<F>(c, K->kind, a);
            else return launch_pair<F>(c, K->kind, a);
        };
        const bool f32 = c->arith_f32 != 0;
        const bool g7 = tait_gk(g->eos_par[2]) == 3; // the usual exponent: kernels that fold it
        if (fam == FAM_WCSPH && eosf && umass && g7) SPH_TRY(f32 ? run_wcsph(FamWCSPHE_T<float, true>()) : run_wcsph(FamWCSPHE_T<double, true>()));
        else if (fam == FAM_WCSPH && eosf && g7) SPH_TRY(f32 ? run_wcsph(FamWCSPHE_T<float>()) : run_wcsph(FamWCSPHE_T<double>()));
        else if (fam == FAM_WCSPH && eosf && umass) SPH_TRY(f32 ? run_wcsph(FamWCSPHEG_T<float, true>()) : run_wcsph(FamWCSPHEG_T<double, true>()));
        else if (fam == FAM_WCSPH && eosf) SPH_TRY(f32 ? run_wcsph(FamWCSPHEG_T<float>()) : run_wcsph(FamWCSPHEG_T<double>()));
        else if (fam == FAM_WCSPH && eosv) SPH_TRY(f32 ? run_wcsph(FamWCSPHV_T<float>()) : run_wcsph(FamWCSPHV_T<double>()));
        else if (fam == FAM_WCSPH) SPH_TRY(f32 ? run_wcsph(FamWCSPH_T<float>()) : run_wcsph(FamWCSPH()));
        else if (fam == FAM_DENSITY) SPH_TRY(f32 ? run_density(FamDensity_T<float>()) : run_density(FamDensity()));
        else if (fam == FAM_VGRAD) SPH_TRY(f32 ? run_vgrad(FamVGrad_T<float>()) : run_vgrad(FamVGrad()));
        else if (fam == FAM_ELASTIC && elu) SPH_TRY(f32 ? run_elastic(FamElasticU_T<float>()) : run_elastic(FamElasticU_T<double>()));
        else if (fam == FAM_ELASTIC) SPH_TRY(f32 ? run_elastic(FamElastic_T<float>()) : run_elastic(FamElastic()));
        else if (tvff) SPH_TRY(f32 ? run_tvf(FamTVFE_T<float>()) : run_tvf(FamTVFE_T<double>()));
        else SPH_TRY(f32 ? run_tvf(FamTVF_T<float>()) : run_tvf(FamTVF()));
    }
    HIP_TRY(hipGetLastError());
    return SPH_OK;
}

// ---------------------------------------------------------------------------
// generated equation families (pysph_amd/codegen.py): pack the sources'
// records with the properties the generated bodies read, then call the
// module's launch function with plain pointers.
// ---------------------------------------------------------------------------
static int pack_generic(sph_ctx *c, int id, size_t off, int nprops, const int *props, int nr, int layout)
{
    DevArray &A = c->arr[id];
    if (A.n == 0) return SPH_OK;
    PackArgs pa;
    memset(&pa, 0, sizeof pa);
    pa.perm = A.perm.as<uint32_t>();
    pa.n = A.n;
    pa.off = off;
    pa.x = A.prop[SPH_X]; pa.y = A.prop[SPH_Y]; pa.z = A.prop[SPH_Z]; pa.h = A.prop[SPH_H];
    pa.na = nprops;
    for (int k = 0; k < nprops; k++) {
        pa.src[k] = A.prop[props[k]];
        if (!pa.src[k]) return need_prop(c, id, props[k], "generated pair loop");
    }
    pa.posh = c->posh.as<double4>();
    pa.aux = c->aux.as<double>();
    pa.rec = c->posh.as<double>();
    pa.nr = nr;
    pa.layout = layout;
    pa.fpos = c->fposb.as<float4>();
    for (int k = 0; k < 3; k++) pa.gmin[k] = c->xmin[k];
    pa.radius_scale = c->radius_scale;
    pa.lds_np = pack_pieces(pa);
    launch_pack(c, pa, A.n);
    return SPH_OK;
}

// the loop nest of one generated family; sph_eval_generated wraps it with the
// round trip of the equation attributes the device code assigns
static int eval_generated_launches(sph_ctx *c, const sph_kernel *K, const sph_gen_family *f, double t, double dt, double *d_state);

extern "C" int sph_eval_generated(sph_ctx *c, const sph_kernel *K, sph_gen_family *f, double t, double dt)
{
    if (!c || !K || !f || !f->launch) { sph_set_error("sph_eval_generated: NULL argument"); return SPH_ERR_ARG; }
    if (f->nstate < 0 || f->nstate > SPH_GEN_MAX_STATE) { sph_set_error("sph_eval_generated: nstate out of range"); return SPH_ERR_ARG; }
    double *d_state = nullptr;
    if (f->nstate > 0) {
        // self.<attr> = ... in an equation body (e.g. the convergence flag of an iterated group,
        // gas_dynamics/basic.py:121-160): values go in before the launches and come back after
        HIP_TRY(hipSetDevice(c->device));
        SPH_TRY(c->gen_state.reserve(SPH_GEN_MAX_STATE * sizeof(double)));
        d_state = c->gen_state.as<double>();
        HIP_TRY(hipMemcpyAsync(d_state, f->state, f->nstate * sizeof(double), hipMemcpyHostToDevice, c->stream));
    }
    int rc = eval_generated_launches(c, K, f, t, dt, d_state);
    if (rc == SPH_OK && f->nstate > 0) {
        HIP_TRY(hipMemcpyAsync(f->state, d_state, f->nstate * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return rc;
}

static int eval_generated_launches(sph_ctx *c, const sph_kernel *K, const sph_gen_family *f, double t, double dt, double *d_state)
{
    if (!c || !K || !f || !f->launch) { sph_set_error("sph_eval_generated: NULL argument"); return SPH_ERR_ARG; }
    if (K->kind < 1 || K->kind > 4) { sph_set_error("sph_eval_generated: unknown kernel kind %d", K->kind); return SPH_ERR_UNSUPPORTED; }
    if (f->dest < 0 || f->dest >= SPH_MAX_ARRAYS || !c->arr[f->dest].used) { sph_set_error("sph_eval_generated: bad dest array %d", f->dest); return SPH_ERR_ARG; }
    if (f->nsrc < 0 || f->nsrc > SPH_MAX_ARRAYS || f->n_sprops < 0 || f->n_sprops > SPH_GEN_MAX_SPROPS || f->n_sprops > MAX_AUX ||
        f->n_din < 0 || f->n_din > SPH_GEN_MAX_PROPS || f->n_dout < 0 || f->n_dout > SPH_GEN_MAX_PROPS || f->npar < 0 ||
        f->npar > SPH_GEN_MAX_PAR) {
        sph_set_error("sph_eval_generated: family descriptor out of range");
        return SPH_ERR_ARG;
    }
    HIP_TRY(hipSetDevice(c->device));
    c->nl.valid = false; // a generated body may write positions or h: kept neighbour lists are not trusted across it
    const int dst = f->dest;
    DevArray &D = c->arr[dst];
    size_t start = f->start_idx > 0 ? (size_t)f->start_idx : 0;
    size_t stop = f->stop_idx >= 0 ? (size_t)f->stop_idx : (f->real ? D.n_real : D.n);
    if (stop > D.n) stop = D.n;

    sph_gen_args g;
    memset(&g, 0, sizeof g);
    g.stream = (void *)c->stream;
    g.kernel_kind = K->kind;
    g.uniform_h = (c->uniform_h && c->use_uniform_h) ? 1 : 0;
    g.nsrc = f->nsrc;
    g.t = t; g.dt = dt;
    g.state = d_state;
    g.n_din = f->n_din; g.n_dout = f->n_dout; g.npar = f->npar;
    for (int k = 0; k < f->npar; k++) g.par[k] = f->par[k];
    for (int k = 0; k < f->n_dout; k++) {
        int p = f->dout[k];
        if (p < 0 || p >= SPH_PROP_COUNT) { sph_set_error("sph_eval_generated: bad output property %d", p); return SPH_ERR_ARG; }
        SPH_TRY(sph_array_ensure_prop(c, dst, p));
        g.dout[k] = D.prop[p];
        sph_mark_written(D, p); // a generated body writes h or m: the next neighbour update looks at them again
        if (p >= SPH_R00 && p <= SPH_R22) D.tflag_valid = false;
    }
    for (int k = 0; k < f->n_din; k++) {
        int p = f->din[k];
        if (p < 0 || p >= SPH_PROP_COUNT) { sph_set_error("sph_eval_generated: bad input property %d", p); return SPH_ERR_ARG; }
        if (!D.prop[p]) return need_prop(c, dst, p, "generated equation (destination)");
        g.din[k] = D.prop[p];
    }
    g.d_start = (uint32_t)start; g.d_stop = (uint32_t)stop;
    g.nd = (uint32_t)D.n;
    g.sigma = K->fac; g.deltap = K->deltap; g.dim = K->dim;
    g.row_mod3 = (int)c->row_mod3; g.norm_masks = (int)c->norm_masks;
    if (D.n == 0 || stop <= start) return SPH_OK;

    if (f->split_init) { // initialize for ALL particles before anything reads it as a source
        sph_gen_args gi = g;
        gi.mode = 1;
        gi.nsrc = 0;
        ScopedTimer tm(c, T_EOS);
        int rc = f->launch(&gi);
        if (rc != 0) { sph_set_error("generated family initialize launch failed (code %d)", rc); return SPH_ERR_HIP; }
        g.skip_init = 1;
    }

    // The per-source part of the loop nest.  `only` < 0: every source in one fused pass (the
    // normal case); otherwise just source `only` (families with initialize_pair and several
    // sources, which the reference sequences source by source: mako :62-110).
    auto loops = [&](int only, bool last) -> int {
        int idx[SPH_MAX_ARRAYS], ns = 0;
        for (int j = 0; j < f->nsrc; j++) if (only < 0 || only == j) idx[ns++] = j;
        sph_gen_args q = g;
        q.nsrc = ns;
        q.dflags = 0;
        if (ns > 0 && f->loop_all) {
            if (!c->nnps_valid) { sph_set_error("sph_eval_generated: neighbour grid is stale; call sph_nnps_update"); return SPH_ERR_STATE; }
            for (int jj = 0; jj < ns; jj++) {
                const int j = idx[jj], s = f->src[j];
                size_t total = 0;
                SPH_TRY(nnps_build_csr_device(c, s, dst, c->csr_start[jj], c->csr_nbrs[jj], &total));
                q.csr_start[jj] = c->csr_start[jj].as<uint32_t>();
                q.csr_nbrs[jj] = c->csr_nbrs[jj].as<uint32_t>();
                q.src_flags[jj] = f->src_flags[j];
                q.dflags |= f->src_flags[j];
                for (int k = 0; k < f->n_sprops; k++) {
                    const double *p = c->arr[s].prop[f->sprops[k]];
                    if (!p && c->arr[s].n) return need_prop(c, s, f->sprops[k], "generated loop_all");
                    q.sraw[jj][k] = p;
                }
            }
            q.mode = 2;
            q.radius_scale = c->radius_scale;
            q.skip_post = (f->also_pair || !last) ? 1 : 0;
            {
                ScopedTimer tm(c, T_PAIR);
                int rc = f->launch(&q);
                if (rc != 0) { sph_set_error("generated loop_all launch failed (code %d)", rc); return SPH_ERR_HIP; }
            }
            if (!f->also_pair) return SPH_OK;
            // the same family's pair loops follow (mako :62-110: loop_all, then loop, per source);
            // initialize has run (split launch), post_loop runs at the end of the pair launch
            q.mode = 0;
            q.skip_init = 1;
            q.dflags = 0;
        }
        q.skip_post = last ? 0 : 1;
        bool a32 = false;

        if (ns > 0) {
            if (!c->nnps_valid) { sph_set_error("sph_eval_generated: neighbour grid is stale; call sph_nnps_update"); return SPH_ERR_STATE; }
            if (D.nnps_slot < 0) { sph_set_error("destination array %d is not part of the neighbour grid", dst); return SPH_ERR_STATE; }
            if (c->ghosts_binned) { sph_set_error("generated families do not read ghost segments (ghost split): use the plain exchange -> sph_nnps_update order"); return SPH_ERR_UNSUPPORTED; }
            SPH_TRY(nnps_need_tables(c));
            size_t total = 0, off_of[SPH_MAX_ARRAYS];
            bool dest_is_src = false;
            for (int jj = 0; jj < ns; jj++) {
                const int s = f->src[idx[jj]];
                if (c->arr[s].nnps_slot < 0) { sph_set_error("source array %d is not part of the neighbour grid", s); return SPH_ERR_STATE; }
                off_of[jj] = total; total += c->arr[s].n; dest_is_src |= s == dst;
            }
            size_t d_off = total;
            if (!dest_is_src) total += D.n;
            else for (int jj = 0; jj < ns; jj++) if (f->src[idx[jj]] == dst) d_off = off_of[jj];
            if (total >= (1ull << 32)) { sph_set_error("too many particles for 32-bit packed indices"); return SPH_ERR_ARG; }
            const int na = f->n_sprops;
            // whole 16-byte pieces; under uniform h the records drop h: [x y z | aux...]
            const bool compact = q.uniform_h != 0;
            const int na_fam = ((na + 1) & ~1) < 2 ? 2 : ((na + 1) & ~1); // FamGen::NA
            // option arith_f32: the family's float build (fp32 records, arithmetic and accumulators), when the host gave one
            a32 = c->arith_f32 && f->launch_f32 != nullptr;
            const bool rf32 = c->record_f32 || a32;
            const int nr = rf32 ? ((4 + na_fam + 3) & ~3) /* floats */
                         : compact ? ((3 + na + 1) & ~1) : 4 + ((na + 1) & ~1);
            const int layout = rf32 ? 5 : (compact ? 4 : 0);
            q.rec_f32 = rf32 ? 1 : 0;
            SPH_TRY(c->posh.reserve((total + 64) * sizeof(double) * nr));
            for (auto &pc : c->pack_cache) pc.epoch = 0; // the generated family's records overwrite the shared WCSPH slots
            SPH_TRY(c->aux.reserve(64));
            SPH_TRY(c->fposb.reserve((total + 64) * sizeof(float4)));
            {
                ScopedTimer tm(c, T_PACK);
                for (int jj = 0; jj < ns; jj++) SPH_TRY(pack_generic(c, f->src[idx[jj]], off_of[jj], na, f->sprops, nr, layout));
                if (!dest_is_src) {
                    // the destination only needs its position record; properties it lacks are not read
                    int have[SPH_GEN_MAX_SPROPS], nh = 0;
                    for (int k = 0; k < na; k++) if (D.prop[f->sprops[k]]) have[nh++] = f->sprops[k];
                    SPH_TRY(pack_generic(c, dst, d_off, nh == na ? na : 0, f->sprops, nr, layout));
                }
            }
            q.rec = c->posh.as<double>();
            q.nrec = nr;
            q.fpos = c->fposb.as<float4>();
            q.dom_extent = fmax(fmax(c->xmax[0] - c->xmin[0], c->xmax[1] - c->xmin[1]), c->xmax[2] - c->xmin[2]);
            for (int k = 0; k < 3; k++) { q.nc[k] = c->nc[k]; q.xmin[k] = c->xmin[k]; }
            q.cell_size = c->cell_size;
            q.radius_scale = c->radius_scale;
            q.hu = 0.5 * (c->h_uniform + c->h_uniform);
            q.h1u = 1.0 / q.hu;
            q.facu = K->fac * q.h1u;
            if (K->dim > 1) q.facu *= q.h1u;
            if (K->dim > 2) q.facu *= q.h1u;
            q.epsu = 0.01 * q.hu * q.hu;
            q.hr2u = (c->radius_scale * c->h_uniform) * (c->radius_scale * c->h_uniform);
            for (int jj = 0; jj < ns; jj++) {
                const int j = idx[jj];
                q.src_cell_start[jj] = c->arr[f->src[j]].cell_start.as<uint32_t>();
                q.src_fine_start[jj] = c->arr[f->src[j]].fine_start.as<uint32_t>();
                q.src_off[jj] = (uint32_t)off_of[jj];
                q.src_flags[jj] = f->src_flags[j];
                q.dflags |= f->src_flags[j];
            }
            q.d_off = (uint32_t)d_off;
            q.d_keys = D.keys_sorted.as<uint32_t>();
            q.d_fkeys = D.fkeys_sorted.as<uint32_t>();
            q.d_perm = D.perm.as<uint32_t>();
            q.d_tile_order = D.n_tiles ? D.tile_order.as<uint32_t>() : nullptr;
        }
        ScopedTimer tm(c, ns > 0 ? T_PAIR : T_EOS);
        int rc = (a32 ? f->launch_f32 : f->launch)(&q);
        if (rc != 0) { sph_set_error("generated family launch failed (code %d)", rc); return SPH_ERR_HIP; }
        return SPH_OK;
    };

    for (int j = 0; j < f->nsrc; j++) {
        const int s = f->src[j];
        if (s < 0 || s >= SPH_MAX_ARRAYS || !c->arr[s].used) { sph_set_error("bad source array %d", s); return SPH_ERR_ARG; }
        for (int i = 0; i < j; i++) if (f->src[i] == s) { sph_set_error("source array %d listed twice", s); return SPH_ERR_ARG; }
    }
    if (f->nsrc > 0 && f->init_pair) {
        // initialize_pair (mako :62-75): per source, before that source's loops, one sweep over
        // the destinations with the source's ARRAYS in view (mode 3; initialize ran as a split launch)
        g.skip_init = 1;
        for (int j = 0; j < f->nsrc; j++) {
            sph_gen_args gp = g;
            gp.mode = 3;
            gp.nsrc = 1;
            gp.src_flags[0] = f->src_flags[j];
            for (int k = 0; k < f->n_sprops; k++) gp.sraw[0][k] = c->arr[f->src[j]].prop[f->sprops[k]];
            const bool last = j == f->nsrc - 1;
            gp.skip_post = (f->init_pair == 2 && last) ? 0 : 1;
            {
                ScopedTimer tm(c, T_EOS);
                int rc = f->launch(&gp);
                if (rc != 0) { sph_set_error("generated initialize_pair launch failed (code %d)", rc); return SPH_ERR_HIP; }
            }
            if (f->init_pair == 1) SPH_TRY(loops(j, last));
        }
        return SPH_OK;
    }
    return loops(-1, true);
}

// ---------------------------------------------------------------------------
// max reduction (adaptive time step inputs: integrator.py:161-200)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_max(const double *__restrict__ v, size_t n, double *__restrict__ part, double sgn)
{
    double m = -DBL_MAX;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        m = fmax(m, sgn * v[i]);
    for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o, 64));
    __shared__ double s[4];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = fmax(fmax(s[0], s[1]), fmax(s[2], s[3]));
}

static int reduce_minmax(sph_ctx *c, int id, int prop, double *out, bool want_max);

extern "C" int sph_reduce_max(sph_ctx *c, int id, int prop, double *out) { return reduce_minmax(c, id, prop, out, true); }
extern "C" int sph_reduce_min(sph_ctx *c, int id, int prop, double *out) { return reduce_minmax(c, id, prop, out, false); }

static int reduce_minmax(sph_ctx *c, int id, int prop, double *out, bool want_max)
{
    if (id < 0 || id >= SPH_MAX_ARRAYS || prop < 0 || prop >= SPH_PROP_COUNT || !c->arr[id].prop[prop]) {
        sph_set_error("sph_reduce_max: array %d has no device property %d", id, prop);
        return SPH_ERR_MISSING_PROP;
    }
    HIP_TRY(hipSetDevice(c->device));
    DevArray &A = c->arr[id];
    size_t n = A.n_real;
    const double sgn = want_max ? 1.0 : -1.0;
    if (n == 0) { *out = want_max ? -DBL_MAX : DBL_MAX; return SPH_OK; }
    int nb = (int)std::min<size_t>(512, (n + 255) / 256);
    SPH_TRY(c->red_part.reserve(1024 * sizeof(double)));
    hipLaunchKernelGGL(k_max, dim3(nb), dim3(256), 0, c->stream, A.prop[prop], n, c->red_part.as<double>(), sgn);
    hipLaunchKernelGGL(k_max, dim3(1), dim3(256), 0, c->stream, c->red_part.as<double>(), (size_t)nb,
                       c->red_part.as<double>() + 512, 1.0);
    HIP_TRY(hipMemcpyAsync(c->pinned, c->red_part.as<double>() + 512, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    *out = sgn * c->pinned[0];
    return SPH_OK;
}
