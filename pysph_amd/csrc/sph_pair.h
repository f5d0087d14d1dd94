// sph_pair.h -- the pair-loop skeleton shared by the hand-written equation
// families (sph_eval.hip) and by generated ones (pysph_amd/codegen.py):
// launch arguments, per-pair geometry and kernel helpers, record access and the
// two-phase pair kernel k_pair_wave<Fam, KK, UH, F32, CF>.
//
// A family `Fam` supplies
//   Real                 double or float: the type the pair arithmetic runs in
//                        (float: BASELINE config 5 / the reference's GPU default,
//                        acceleration_eval_gpu_helper.py:281-283,437-441)
//   MINB                 wavefronts per SIMD to compile for (VGPR budget)
//   NA, NR               aux doubles per record, record length (>= 4 + NA)
//   Params               launch constants / output pointers
//   Dest                 per-destination registers (inputs + accumulators)
//   load(D, s, a, o)     initialize: from the destination's own record `s`
//                        (and, for generated families, memory at original index o)
//   pair<KK,UH>(D, pi, pj, r2, s, flags, a)   one neighbour pair
//   finish(D, a, o)      post_loop + the single write per output
// and may specialise load_record<Fam, UH> for a custom record layout.
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>

#include "sphhip.h"
#include "sph_kernels.h"

// ---------------------------------------------------------------------------
// exact (non-contracted) squared distance: must round like the reference's
// norm2 (nnps_base.pxd:36-37) so that neighbour SETS are identical.
// ---------------------------------------------------------------------------
template <class T> __device__ __forceinline__ T r2_exact(T dx, T dy, T dz)
{
#pragma clang fp contract(off)
    const T a = dx * dx, b = dy * dy, c = dz * dz;
    return (a + b) + c;
}

// 4-vector of the arithmetic type
template <class T> struct Real4Of;
template <> struct Real4Of<double> { typedef double4 type; };
template <> struct Real4Of<float> { typedef float4 type; };
template <class T> using real4 = typename Real4Of<T>::type;


struct KernelConst {
    double sigma;  // kernel.fac
    double deltap; // kernel.get_deltap()
    int dim;
};


struct SrcDesc {
    const uint32_t *cell_start;
    uint32_t off;    // offset of this source's segment in the packed buffers
    uint32_t flags;  // equations acting for this (dest, source) pair
    const uint32_t *fine_start; // first sorted position per x sub-bin (SPH_NSUB per cell)
    double mu;       // the one mass of this source's particles (families with uniform-mass records)
    uint32_t ghost;  // 1: the ghost segment of an array (sph_nnps_update_ghosts): read by face wavefronts only
};

// Powers of the Tait EOS (wc/basic.py:60-65: p = p0 + B ((rho/rho0)^gamma - 1), cs = c0 (rho/rho0)^((gamma-1)/2)) for the odd
// integer exponents gamma = 2 gk + 1 = 1, 3, 5, 7 by multiplication -- ONE spelling for k_nosrc and for every record
// decoder that recomputes p and cs, so that the fused and the gathered forms agree to the bit.  gk is wave-uniform.
static __host__ __device__ inline int tait_gk(double gamma)
{
    return gamma == 7.0 ? 3 : gamma == 5.0 ? 2 : gamma == 3.0 ? 1 : gamma == 1.0 ? 0 : -1;
}
template <class T> __device__ __forceinline__ T tait_cs_power(T ratio, int gk) // ratio^gk
{
    if (gk == 3) return (ratio * ratio) * ratio;
    if (gk == 2) return ratio * ratio;
    return gk == 1 ? ratio : T(1.0);
}
template <class T> __device__ __forceinline__ void tait_powers(T ratio, int gk, T &rg, T &rk) // ratio^(2 gk + 1), ratio^gk
{
    const T r2 = ratio * ratio;
    if (gk == 3) { rk = r2 * ratio; rg = (r2 * r2) * rk; }
    else if (gk == 2) { rk = r2; rg = (r2 * r2) * ratio; }
    else if (gk == 1) { rk = ratio; rg = r2 * ratio; }
    else { rk = T(1.0); rg = ratio; }
}

template <class Fam> struct PairArgs {
    int nsrc;
    SrcDesc src[SPH_MAX_ARRAYS];
    const double4 *posh;
    const double *aux;
    const double *rec; // variant 2: interleaved records, Fam::NR doubles each
    int nrec;          // variant 3: doubles per record (Fam::NR, or a compact layout's), floats with record_f32
    const float4 *fpos; // variant 3: fp32 grid-relative positions + radius_scale*h (prefilter only)
    double dom_extent;  // largest grid extent: bounds the fp32 rounding of fpos
    uint32_t d_off, nd;
    double d_mu;     // the one mass of the destination array (uniform-mass records: decoding the destination's own record)
    const uint32_t *d_keys, *d_fkeys, *d_perm; // cell ids / fine keys of the sorted destinations, sorted -> original index
    const uint8_t *d_slot; // merged order (families with MERGED): the array (nnps slot) of every sorted destination
    const uint32_t *d_tile_order; // traversal order of the destination tiles (null: memory order)
    const uint32_t *d_list;       // (optional) the destinations: lane t of wave tile w takes sorted position d_list[64 w + t], nd =
                                  // the length of the list (DevArray::dlist: the real particles of the order, ascending) -- wave
                                  // tiles without idle ghost lanes; null: position 64 w + t itself
    // ghost split: ghosts have a fine x index <= gfx_lo or >= gfx_hi; a wavefront whose candidate windows stay inside is
    // an INTERIOR one (it skips ghost segments); face_mode 1 = interior wavefronts only, 2 = the others only, 0 = all
    int face_mode, gfx_lo, gfx_hi;
    int row_mod3;      // order in which a wavefront visits its 3x3 rows of cells: bit 0 / bit 1 -- step (sy, sz) takes the
                       // row whose y / z is congruent to the step modulo 3, so that ALL wavefronts in flight walk rows of
                       // one residue class at a time (they share them in L2 / L1); 4 -- natural order rotated per wave tile
    uint32_t d_start, d_stop;
    int nc[3];
    double xmin[3];
    double cell_size;
    double radius_scale;
    KernelConst k;
    uint32_t dflags; // union of the source flags
    unsigned long long *dbg; // profiling only (option count_iters): [0] phase-2 iterations, [1] phase-2 calls, [2] wavefronts, [3] row tiles
    int ablate;      // profiling only: 1 = skip pair arithmetic, 2 = skip phase 2
    int skip_init;   // generated families: initialize() already ran in a launch of its own
    int skip_post;   // generated loop_all launch followed by a pair launch: post_loop runs there
    double t, dt;
    // constants of the uniform-h specialisation (hmin == hmax over all arrays)
    double hu, h1u, facu, epsu, hr2u;
    // neighbour-list reuse between the pair passes of one evaluation (the reference's NeighborCache,
    // nnps_base.pyx:1144-1257): nl_mode 1 = this pass keeps every lane's hit-mask slot list, 2 = this pass
    // starts from the lists a previous pass kept (one source only).  Per wave tile NLW words:
    // [slots held][per slot: mask lo, mask hi, index of bit 0 relative to the source's segment] x 64 lanes.
    uint32_t *nl;
    int nl_mode;
    int norm_masks;  // 1: a row's 96 hit bits are shifted down to the lane's first hit before they become slots
    // Tait EOS of the records' density (families with EOSF: p and cs are not gathered but recomputed)
    double e_rho01, e_c0, e_B, e_p0;
    int e_gk;                   // Tait exponent gamma = 2 e_gk + 1 (1, 3, 5, 7): powers by multiplication (tait_powers)
    typename Fam::Params p;
};

// does the family run on the merged order of all arrays (destinations of several arrays in one launch; per-slot
// destination ranges and output pointers in Fam::Params, read per lane with kernarg_read)?
template <class F, class = void> struct fam_merged { static constexpr bool value = false; };
template <class F> struct fam_merged<F, decltype((void)F::MERGED)> { static constexpr bool value = F::MERGED; };

// A per-lane (dynamically indexed) read of the kernel's own by-value argument struct, straight from the kernarg
// segment: `byte_offset` is the offset of the element inside the (single) kernel argument.  Indexing the by-value
// struct itself with a lane-dependent index could make the compiler copy the table to scratch first.  Scalar V only.
template <class V> __device__ __forceinline__ V kernarg_read(size_t byte_offset)
{
    typedef const char __attribute__((address_space(4))) *kptr;
    const kptr ka = (kptr)__builtin_amdgcn_kernarg_segment_ptr();
    return *reinterpret_cast<const V __attribute__((address_space(4))) *>(ka + byte_offset);
}

// does the family read a wave-uniform word once per wavefront and hand it to every record decode (Fam::wave_token,
// load_fused(..., token))?  (elastic rates: "no particle of the source array is in tension")
template <class F, class = void> struct fam_token { static constexpr bool value = false; };
template <class F> struct fam_token<F, decltype((void)F::TOKEN)> { static constexpr bool value = F::TOKEN; };

// does the family recompute p, cs from rho (64-byte WCSPH records)?
template <class F, class = void> struct fam_eosf { static constexpr bool value = false; };
template <class F> struct fam_eosf<F, decltype((void)F::EOSF)> { static constexpr bool value = F::EOSF; };


// ---------------------------------------------------------------------------
// fast fp64 reciprocal / square root: hardware estimate (v_rcp_f64 / v_rsq_f64,
// ~2^-26 relative) + SPH_NEWTON_STEPS Newton steps: one step leaves ~1e-14
// relative (measured: tools/microbench/rcp_accuracy.hip), two ~1 ulp.  No
// div_scale/div_fixup range handling -- operands here are densities, distances
// and smoothing lengths, far from the fp64 range limits.  The 1e-10 parity
// budget (BASELINE.json) is four orders of magnitude above the one-step error;
// neighbour SETS do not depend on it (exact r2 comparison).
// ---------------------------------------------------------------------------
#ifndef SPH_NEWTON_STEPS
#define SPH_NEWTON_STEPS 1
#endif
__device__ __forceinline__ double fast_rcp(double d)
{
    double x = __builtin_amdgcn_rcp(d);
    double e = fma(-d, x, 1.0);
    x = fma(x, e, x);
    if (SPH_NEWTON_STEPS > 1) {
        e = fma(-d, x, 1.0);
        x = fma(x, e, x);
    }
    return x;
}
// s = sqrt(a), rs = 1/sqrt(a).  a is clamped to 1e-300 so that coincident
// particles (a == 0: the self pair) give finite s ~ 1e-150, rs ~ 1e150; every
// use of rs multiplies it with a factor that is exactly 0 for such pairs
// (XIJ, or h*VIJ.XIJ), and the gradient has the reference's own r > 1e-12 guard.
__device__ __forceinline__ void fast_sqrt_rsqrt(double a, double &s, double &rs)
{
    a = raw_max(a, 1e-300);
    double y = __builtin_amdgcn_rsq(a);
    double g = a * y, h = 0.5 * y;
    double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    if (SPH_NEWTON_STEPS > 1) {
        r = fma(-h, g, 0.5);
        g = fma(g, r, g);
        h = fma(h, r, h);
    }
    s = g;
    rs = h + h;
}
// fp32: v_rcp_f32 / v_rsq_f32 are 1-ulp instructions, no refinement needed
__device__ __forceinline__ float fast_rcp(float d) { return __builtin_amdgcn_rcpf(d); }
__device__ __forceinline__ void fast_sqrt_rsqrt(float a, float &s, float &rs)
{
    a = fmaxf(a, 1e-35f);
    rs = __builtin_amdgcn_rsqf(a);
    s = a * rs;
}

// per-pair geometry shared by all families
template <class T> struct PairGeomT {
    T xij[3];
    T r2, rij, rinv, hij, h1, q, fac, eps;
};
typedef PairGeomT<double> PairGeom;

// UH: every particle has the same h -> HIJ, 1/HIJ, the kernel normalisation
// and EPS are launch constants.
template <int KK, bool UH, class T, class A>
__device__ __forceinline__ void pair_geom(PairGeomT<T> &g, const real4<T> &pi, const real4<T> &pj, T r2, const A &a)
{
    g.xij[0] = pi.x - pj.x; g.xij[1] = pi.y - pj.y; g.xij[2] = pi.z - pj.z; // XIJ equation.py:205-212
    g.r2 = r2;                                                               // R2IJ :226-233
    // The Gaussian is cut off at q = 3 where it is still 1.2e-4 of its peak: a
    // pair at r = 3h to within an ulp (exact lattices produce them) is in or out
    // depending on the last bit of q, so q must be the reference's own
    // q = sqrt(r2) * (1/h) with correctly rounded sqrt and division.  The
    // polynomial kernels vanish at their support radius; the ~1 ulp fast paths
    // are invisible there.
    constexpr bool EXACT_Q = (KK == 4);
    if (EXACT_Q) {
        g.rij = sqrt(r2);
        g.rinv = g.rij > T(0) ? T(1) / g.rij : T(0);
    } else {
        fast_sqrt_rsqrt(r2, g.rij, g.rinv);                                  // RIJ  :235
    }
    if (UH) {
        g.hij = (T)a.hu; g.h1 = (T)a.h1u; g.fac = (T)a.facu; g.eps = (T)a.epsu;
    } else {
        g.hij = T(0.5) * (pi.w + pj.w);                                      // HIJ  :192
        g.h1 = EXACT_Q ? T(1) / g.hij : fast_rcp(g.hij);
        g.fac = kernel_norm((T)a.k.sigma, g.h1, a.k.dim);
        g.eps = T(0.01) * g.hij * g.hij;                                     // EPS  :194
    }
    g.q = g.rij * g.h1;
}
// The same with ONE transcendental for the pair (fp64): the momentum equation needs 1 / ((r2 + eps) rhoij) next to
// r = sqrt(r2) and 1 / r.  With P = (r2 + eps) rhoij:  t = rsqrt(r2 P^2) = 1 / (r P)  =>  1/r = t P,  r = r2 (1/r),
// 1/P = t r -- v_rsq_f64 + one Newton step instead of v_rsq_f64 + v_rcp_f64 + a step each (the transcendentals issue at a
// quarter of the fp64 rate).  r2 is clamped to 1e-300 first: coincident particles get r ~ 1e-150 (below every r > 1e-12
// guard) and still the exact 1/P.  rho_ij: RHOIJ of equation.py:196; returns tt = 1 / ((r2 + eps) rhoij).
#ifndef SPH_MERGED_RSQ
#define SPH_MERGED_RSQ 0 // measured (profiles/r06_ab_tables.txt): 1.818-1.836 ms against 1.825-1.842 on the 4 M cube, 2.114 against 2.097 on the 4 M dam break -- nothing; off
#endif
template <int KK, bool UH, class A>
__device__ __forceinline__ double pair_geom_mom(PairGeomT<double> &g, const double4 &pi, const double4 &pj, double r2, double rhoij, const A &a)
{
    typedef double T;
    g.xij[0] = pi.x - pj.x; g.xij[1] = pi.y - pj.y; g.xij[2] = pi.z - pj.z;
    g.r2 = r2;
    if (UH) {
        g.hij = (T)a.hu; g.h1 = (T)a.h1u; g.fac = (T)a.facu; g.eps = (T)a.epsu;
    } else {
        g.hij = T(0.5) * (pi.w + pj.w);
        g.h1 = fast_rcp(g.hij);
        g.fac = kernel_norm((T)a.k.sigma, g.h1, a.k.dim);
        g.eps = T(0.01) * g.hij * g.hij;
    }
    const T P = (r2 + g.eps) * rhoij;
    const T r2c = raw_max(r2, 1e-300);
    const T aa = r2c * (P * P);
    const T y = __builtin_amdgcn_rsq(aa);
    T gg = aa * y, hh = 0.5 * y;
    const T rr = fma(-hh, gg, 0.5);
    hh = fma(hh, rr, hh);
    const T t = hh + hh;            // 1 / (r P)
    g.rinv = t * P;
    g.rij = r2c * g.rinv;
    g.q = g.rij * g.h1;
    return t * g.rij;               // 1 / P
}

template <int KK, bool UH, class T> __device__ __forceinline__ T pair_w(const PairGeomT<T> &g) { return SphKernel<KK>::template w<UH>(g.q) * g.fac; }
// GRADH(XIJ, RIJ, h) = dW/dh (kernels.py gradient_h, e.g. :138-163): -fac*h1*(dw*q + w*dim)
template <int KK, bool UH, class T> __device__ __forceinline__ T pair_gradh(const PairGeomT<T> &g, int dim)
{
    return -g.fac * g.h1 * (SphKernel<KK>::template dw<UH>(g.q) * g.q + SphKernel<KK>::template w<UH>(g.q) * (T)dim);
}
// GRADIENT(XIJ, RIJ, HIJ, DWIJ) (kernels.py:126-137) returns tmp*xij with
// tmp = dwdq*h1/rij; here tmp only.  dw(q)/rij = dwq(q)*h1 when the kernel has
// a closed form for dw/q.
template <int KK, bool UH, class T> __device__ __forceinline__ T pair_gradfac(const PairGeomT<T> &g)
{
    T t;
    if (SphKernel<KK>::HAS_DWQ) t = SphKernel<KK>::template dwq<UH>(g.q) * (g.fac * g.h1 * g.h1);
    else t = SphKernel<KK>::template dw<UH>(g.q) * (g.fac * g.h1) * g.rinv;
    return g.rij > T(1e-12) ? t : T(0);
}

// Record access of the pair kernel.  Default layout: [x y z h | aux...].
template <class Fam, bool UH>
__device__ __forceinline__ void load_record(const double *__restrict__ rj, uint32_t fl, double4 &pj, double (&s)[Fam::NA])
{
    pj = *reinterpret_cast<const double4 *>(rj);
#pragma unroll
    for (int k = 0; k < Fam::NA; k++) s[k] = rj[4 + k];
}

// fp32 RECORDS: [x-x0 y-y0 z-z0 h | aux...] as floats, 16-byte pieces of four;
// positions are relative to the grid origin so that their rounding is relative
// to the domain extent.  Read by the fp32 families (Real = float) as they are,
// and by fp64 families under option record_f32 (values widened on load:
// arithmetic and accumulation fp64, inputs of fp32 precision).
template <class Fam, class T>
__device__ __forceinline__ void load_record_f32(const float *__restrict__ rj, real4<T> &pj, T (&s)[Fam::NA])
{
    constexpr int NF = (4 + Fam::NA + 3) & ~3;
    float f[NF];
    const float4 *r4 = reinterpret_cast<const float4 *>(rj);
#pragma unroll
    for (int q = 0; q < NF / 4; q++) { const float4 t = r4[q]; f[4 * q] = t.x; f[4 * q + 1] = t.y; f[4 * q + 2] = t.z; f[4 * q + 3] = t.w; }
    pj.x = f[0]; pj.y = f[1]; pj.z = f[2]; pj.w = f[3];
#pragma unroll
    for (int k = 0; k < Fam::NA; k++) s[k] = f[4 + k];
}

// Branching kernels (`if (criterion) pair(...)`) call this between the gather
// and the criterion: the use keeps all pieces of the record in ONE batch of
// loads ahead of the branch.  Without it the optimiser sinks the non-position
// pieces into the branch (as misaligned loads): two memory latencies per hit.
template <int NA, class T>
__device__ __forceinline__ void pin_record(const real4<T> &pj, const T (&s)[NA])
{
    asm volatile("" ::"v"(pj.x), "v"(pj.y), "v"(pj.z));
#pragma unroll
    for (int k = 0; k < NA; k++)
        if (!__builtin_constant_p(s[k])) asm volatile("" ::"v"(s[k]));
}

// ---- WCSPH: Continuity + Momentum + XSPH (wc/basic.py, basic_equations.py) --
// ---------------------------------------------------------------------------
// XCD-aware block remap (MI355X: 8 XCDs, block b runs on XCD b % 8, each XCD
// has a private 4 MiB L2).  Consecutive tiles of the cell-ordered destination
// array share their 3x3 neighbour rows, so give every XCD one CONTIGUOUS chunk
// of tiles: the rows a workgroup gathers from are then already in its XCD's L2.
// Placement only affects speed, never results.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t xcd_tile(uint32_t b, uint32_t nb)
{
    const uint32_t xcd = b & 7u, idx = b >> 3;
    const uint32_t base = nb >> 3, rem = nb & 7u;
    return xcd * base + min(xcd, rem) + idx;
}

// wave-uniform maximum of a per-lane int
__device__ __forceinline__ int wave_max_i32(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return __builtin_amdgcn_readfirstlane(v);
}

// profiling hooks of the pair kernel (options ablate / count_iters): compiled in only with -DSPH_PROFILING --
// the two uniform values and their tests cost the phase-2 loop scalar registers it does not have
#ifdef SPH_PROFILING
#define ABLATE(a) ((a).ablate)
#define DBGC(a) ((a).dbg)
#else
#define ABLATE(a) 0
#define DBGC(a) ((unsigned long long *)nullptr)
#endif

#define AMAXLEN 96 // hit bits kept per row and lane; longer ranges take the slow tail

typedef float f2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------
// variant 6 (default): one WAVEFRONT per 64 consecutive cell-ordered
// destinations, no workgroup barriers.
//   Phase 1 (per row of cells, per source): the wavefront stages the fp32
//   positions of ITS candidate range of the row (x sub-bins [first lane - XWIN,
//   last lane + XWIN], ~100 candidates) and the row's fine_start slice in its own
//   LDS tile; every lane tests the candidates of its own x window (its sub-bin
//   +- XWIN instead of three whole cells), two per packed-fp32 instruction, one
//   sign bit each, and appends its NON-EMPTY 64-bit hit masks to a private slot
//   list {mask, absolute index of bit 0}.
//   Phase 2 (once per source): one loop per lane -- pop the lowest bit, step to
//   the next slot when the mask runs empty, gather the fp64 record, exact
//   criterion, pair -- wave-collective only through the loop condition.  Lanes
//   that have run out of hits gather their own record (in L1) and contribute
//   nothing; families with PRED take the criterion as a factor, so the loop body
//   is branch-free and the accumulators stay in their registers.  With CF != 0
//   the equation flags are a compile-time constant.
//   Measured on the 4 M cube against the 256-thread workgroup version of the
//   same schedule (barriers around every shared row tile): see DESIGN.md.
// ---------------------------------------------------------------------------
#ifndef SPH_WLQ
#define SPH_WLQ 10
#endif
#ifndef SPH_WCAP_UH
#define SPH_WCAP_UH 184
#endif
#define WLQ SPH_WLQ // slots per lane (9 rows of cells + one spare; a lane out of slots flushes its wavefront's phase 2 early)
#define WCAP_UH SPH_WCAP_UH // candidates per LDS tile piece (a wavefront's row range is ~105 for WCSPH, ~125 for TVF);
#define WCAP_VH 136 // variable h keeps a fourth plane (the candidates' own radii): same 2304 B
#define WCSL 96     // fine_start entries of one row segment kept in LDS (16-bit, relative to the segment start)
#define NLW (64 * (1 + 3 * WLQ)) // 32-bit words one wave tile keeps for the neighbour-list reuse
#define NL_NONE 0xFFFFFFFFu     // stored slot count of a wave tile whose lists are not reusable (flushed early / long rows)
#define XWIN (SPH_NSUB + 1) // a lane's candidate window: its own x sub-bin +- XWIN (one bin of slack for the
                            // rounding of the sub-bin index: |x_j - x_i| < cell_size spans at most SPH_NSUB bins exactly)

#ifndef WPB
#define WPB 1       // wavefronts per workgroup.  There are no barriers, so any value works; 4 (x-adjacent tiles on one
                    // CU, sharing gathered lines in its L1) measured 1 % faster on the uniform-h cube but loses a
                    // workgroup of occupancy to LDS granularity with variable h (4 x 10.7 KB > 40 KB)
#endif

template <class Fam, int KK, bool UH, bool F32 = false, uint32_t CF = 0>
__global__ __launch_bounds__(64 * WPB, Fam::MINB) void k_pair_wave(PairArgs<Fam> a)
{
    typedef typename Fam::Real T; // arithmetic type of the pair loop
    static_assert(F32 || sizeof(T) == 8, "fp32 arithmetic reads fp32 records");
    const uint32_t NR = (uint32_t)a.nrec;
    constexpr int WCAP = UH ? WCAP_UH : WCAP_VH;
    constexpr int TS = WCAP + 8;
    // every wavefront of the workgroup has its own LDS areas
    __shared__ __attribute__((aligned(16))) float tile_[WPB][(UH ? 3 : 4) * TS];
    __shared__ unsigned short csl_[WPB][WCSL];
    __shared__ unsigned long long smask_[WPB][WLQ][64];
    __shared__ uint32_t sjb_[WPB][WLQ][64];
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float *const tile = tile_[wv];
    float *const tx = tile, *const ty = tile + TS, *const tz = tile + 2 * TS, *const tw = tile + (UH ? 0 : 3 * TS);
    unsigned short *const csl = csl_[wv];
    unsigned long long (*const smask)[64] = smask_[wv];
    uint32_t (*const sjb)[64] = sjb_[wv];

    const int t = threadIdx.x & 63; // lane
    // 64-destination wave tiles inside the 256-destination tiles of the traversal order
    const uint32_t wt = xcd_tile(blockIdx.x, gridDim.x) * WPB + (uint32_t)wv;
    uint32_t dtile = wt >> 2;
    if (dtile * 256u >= a.nd) return; // the grid is rounded up to whole workgroups
    if (a.d_tile_order) dtile = a.d_tile_order[dtile];
    const uint32_t tbase = dtile * 256u + (wt & 3u) * 64u;
    const uint32_t i = tbase + t;
    if (tbase >= a.nd) return; // whole wavefront past the end
    const bool valid = i < a.nd;
    uint32_t ic = valid ? i : a.nd - 1;
    if (a.d_list) ic = a.d_list[ic]; // the destination's sorted position (ascending over the lanes either way)
    const uint32_t o = a.d_perm[ic];
    uint32_t slot = 0;
    bool active;
    if constexpr (fam_merged<Fam>::value) {
        // destinations of several arrays: this lane's array decides its index range (Group.real / start_idx / stop_idx;
        // an array without equations as destination has an empty one) and, in finish(), where its results go
        slot = a.d_slot[ic];
        const unsigned long long rg = kernarg_read<unsigned long long>(__builtin_offsetof(PairArgs<Fam>, p) + Fam::rng_offset(slot));
        active = valid && o >= (uint32_t)rg && o < (uint32_t)(rg >> 32);
    } else {
        active = valid && o >= a.d_start && o < a.d_stop;
    }
    const uint32_t fkey = a.d_fkeys[ic];
    const uint32_t key = fkey / SPH_NSUB;
    const int ncx = a.nc[0], ncy = a.nc[1], ncz = a.nc[2];
    const int row = key / ncx;
    const int cx = (int)(key % ncx) * SPH_NSUB + (int)(fkey % SPH_NSUB); // x sub-bin along the row
    bool facew = true; // may this wavefront's candidate windows hold ghosts?  (wave-uniform)
    if (a.face_mode != 0 || a.gfx_lo > -0x7fffffff || a.gfx_hi < 0x7fffffff) {
        int mn = active ? cx : 0x7fffffff, mx = active ? cx : -0x7fffffff;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { mn = min(mn, __shfl_xor(mn, o, 64)); mx = max(mx, __shfl_xor(mx, o, 64)); }
        mn = __builtin_amdgcn_readfirstlane(mn); mx = __builtin_amdgcn_readfirstlane(mx);
        facew = mn <= mx && (mn - XWIN <= a.gfx_lo || mx + XWIN >= a.gfx_hi);
        if (mn > mx && !a.nl_mode) return;         // no active destination in this wavefront
        if (a.face_mode == 1 && facew) return;     // the part that needs no ghosts
        if (a.face_mode == 2 && !facew) return;    // the rest
    }
    real4<T> pi;
    typename Fam::Dest D;
    uint32_t wtok = 0;
    if constexpr (fam_token<Fam>::value) wtok = Fam::wave_token(a);
    T cur_mu = (T)a.d_mu; // mass of the array being read (uniform-mass records): the destination's own first, then each source's
    // one record: fp32 records for Real = float (and for record_f32), else fp64
    auto fetch = [&](uint32_t jg, uint32_t flags, real4<T> &pj, T (&sj)[Fam::NA]) {
        if constexpr (fam_token<Fam>::value) Fam::load_fused(a, jg, flags, cur_mu, pj, sj, wtok);
        else if constexpr (fam_eosf<Fam>::value) Fam::load_fused(a, jg, flags, cur_mu, pj, sj);
        else if constexpr (F32) load_record_f32<Fam, T>(reinterpret_cast<const float *>(a.rec) + (unsigned long long)jg * NR, pj, sj);
        else load_record<Fam, UH>(a.rec + (unsigned long long)jg * NR, flags, pj, sj);
    };
    {
        T sd_[Fam::NA];
        // the destination's own h / p come with the same rules as a source's
        // (uniform h: the constant; p only for the tensile correction)
        fetch(a.d_off + ic, a.dflags, pi, sd_);
        if (UH) pi.w = (T)a.hu;
        if constexpr (fam_merged<Fam>::value) Fam::load(D, sd_, a, o, slot);
        else Fam::load(D, sd_, a, o);
    }
    const int nfx = ncx * SPH_NSUB;
    const T hi_r = (T)a.radius_scale * pi.w;
    const T hi2 = UH ? (T)a.hr2u : hi_r * hi_r;
    const int row_first = __builtin_amdgcn_readfirstlane(row), row_last = __builtin_amdgcn_readlane(row, 63);
    if (DBGC(a) && t == 0) atomicAdd(DBGC(a) + 2, 1ull);

    // exact criterion + pair arithmetic for one candidate record (branching form)
    auto do_pair = [&](uint32_t jg, uint32_t flags) {
        real4<T> pj;
        T sj[Fam::NA];
        fetch(jg, flags, pj, sj);
        pin_record<Fam::NA, T>(pj, sj);
        T hj2 = hi2;
        if (!UH) { hj2 = (T)a.radius_scale * pj.w; hj2 *= hj2; }
        const T r2 = r2_exact<T>(pi.x - pj.x, pi.y - pj.y, pi.z - pj.z);
        if (((r2 < hi2) || (r2 < hj2)) && ABLATE(a) != 1) Fam::template pair<KK, UH>(D, pi, pj, r2, sj, flags, a);
    };

    int cq = 0; // slots this lane holds
    auto phase2 = [&](uint32_t flags) {
        unsigned long long m = 0;
        uint32_t jb = 0;
        int q = 0;
        if (cq > 0 && ABLATE(a) != 2) { m = smask[0][t]; jb = sjb[0][t]; }
        const uint32_t self = a.d_off + ic;
        if (__any(m != 0)) {
            if (DBGC(a) && t == 0) atomicAdd(DBGC(a) + 1, 1ull);
            do {
                if (DBGC(a) && t == 0) atomicAdd(DBGC(a), 1ull);
                const bool has = m != 0;
                uint32_t j = has ? jb + (uint32_t)__builtin_ctzll(m) : self;
                m &= m - 1;
                if (m == 0 && q + 1 < cq) { ++q; m = smask[q][t]; jb = sjb[q][t]; }
                if (ABLATE(a) == 3) j = self;
                if constexpr (Fam::PRED) {
                    real4<T> pj;
                    T sj[Fam::NA];
                    fetch(j, flags, pj, sj);
                    T hj2 = hi2;
                    if (!UH) { hj2 = (T)a.radius_scale * pj.w; hj2 *= hj2; }
                    const T r2 = r2_exact<T>(pi.x - pj.x, pi.y - pj.y, pi.z - pj.z);
                    const bool pass = has && ((r2 < hi2) || (r2 < hj2)) && ABLATE(a) != 1;
                    Fam::template pair<KK, UH>(D, pi, pj, r2, sj, flags, a, pass);
                } else {
                    if (has) do_pair(j, flags);
                }
            } while (__any(m != 0));
        }
        cq = 0;
    };

    // neighbour-list reuse (one source): this wave tile's block of the list buffer
    uint32_t *const nlw = a.nl_mode ? a.nl + (size_t)wt * NLW : nullptr;
    bool reuse = false, unsaved = false; // wave-uniform
    if (a.nl_mode == 2) {
        const uint32_t nq = nlw[t];
        reuse = !__any(nq == NL_NONE);
        if (reuse) {
            const uint32_t off0 = a.src[0].off;
            const int nqmax = wave_max_i32((int)nq);
            for (int q = 0; q < nqmax; q++) {
                const uint32_t *w = nlw + 64 * (1 + 3 * q) + t;
                smask[q][t] = (unsigned long long)w[0] | ((unsigned long long)w[64] << 32);
                sjb[q][t] = w[128] + off0;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            cq = active ? (int)nq : 0; // a lane outside this pass's destination range holds no work
        }
    }

    // Sources outermost: a wavefront whose 64 destinations straddle a row
    // boundary does phase 1 once per row segment (only that segment's lanes
    // build masks) but ONE phase 2 per source for all its lanes together.
    // Phase 2 has ONE call site per source, outside the row loops: a lane that
    // runs out of slots (rare) makes the wavefront leave the loops with its
    // position (row segment, row step, tile, part) saved, work the lists off and
    // re-enter at that position, re-staging the tile it stopped in.  Nothing of
    // phase 1 is live across phase 2 except that position (wave-uniform), which
    // is what lets the register allocator give phase 2 the whole budget.
    for (int s = 0; s < a.nsrc; s++) {
    const SrcDesc sd = a.src[s];
    if (sd.ghost && !facew) continue; // interior wavefront: none of its candidates can be a ghost
    const uint32_t fl = CF ? CF : sd.flags;
    cur_mu = (T)sd.mu;
    int R = row_first, st = 0, part_resume = 0; // position in phase 1 (wave-uniform)
    uint32_t tb_resume = 0;
    bool resumed = false;
    for (;;) {
    bool more = false; // a lane is out of slots: phase 2 now, then back into the loops
    for (; R <= row_last && !more && !reuse && ABLATE(a) != 6; R++, st = 0) { // reuse: the lists are in LDS already; 6 (profiling): prologue + finish only
        const bool inseg = active && row == R;
        const unsigned long long segm = __ballot(inseg);
        if (!segm) continue;
        const int cxa = __builtin_amdgcn_readlane(cx, __builtin_ctzll(segm));
        const int cxb = __builtin_amdgcn_readlane(cx, 63 - __builtin_clzll(segm));
        const int cyR = R % ncy, czR = R / ncy;
        // x sub-bins this segment's destinations can reach, and mine
        const int xa = max(cxa - XWIN, 0), xb = min(cxb + XWIN, nfx - 1);
        const int ncs = xb - xa + 2; // fine_start entries needed: bins xa..xb and the end
        const double binw = a.cell_size * (1.0 / SPH_NSUB);
        // fp32 coordinates: grid-relative positions (fpos, rounded once from
        // fp64) minus this row segment's origin; every value carries at most
        // 2^-24 * dom_extent of rounding, covered by `slack`
        const float oxf = (float)(binw * xa);
        const float oyf = (float)(a.cell_size * (cyR - 1));
        const float ozf = (float)(a.cell_size * (czR - 1));
        const double L = fmax(a.cell_size * (double)max((xb - xa) / SPH_NSUB + 2, 4), a.dom_extent);
        const float slack = (float)(L * 1.5e-6);
        const float4 fpi = a.fpos[a.d_off + ic];
        const float fxs = fpi.x - oxf, fys = fpi.y - oyf, fzs = fpi.z - ozf;
        const f2 fx = {fxs, fxs}, fy = {fys, fys}, fz = {fzs, fzs};
        const float hif = (float)hi_r * 1.000001f + slack;
        const float hi2f = hif * hif;
        const int mycl = max(cx - XWIN, xa) - xa, mych = min(cx + XWIN, xb) + 1 - xa;

        for (; st < 9; st++) {
            // row order (PairArgs::row_mod3)
            const int st2 = a.row_mod3 == 4 ? (st + (int)(wt % 9u)) % 9 : st;
            const int sy = st2 % 3 - 1, sz = st2 / 3 - 1;
            const int dy = (a.row_mod3 & 1) ? (sy + 1 - cyR % 3 + 4) % 3 - 1 : sy;
            const int dz = (a.row_mod3 & 2) ? (sz + 1 - czR % 3 + 4) % 3 - 1 : sz;
            const int yy = cyR + dy, zz = czR + dz;
            if (yy < 0 || yy >= ncy || zz < 0 || zz >= ncz) continue;
            const uint32_t rowb = (uint32_t)(ncx * (yy + ncy * zz)) * SPH_NSUB; // fine index of the row's first sub-bin
            const uint32_t j0 = sd.fine_start[rowb + xa], j1 = sd.fine_start[rowb + xb + 1];
            const bool csl_ok = ncs <= WCSL && j1 - j0 < 65536u;
            uint32_t tb = j0;
            int part = 0;
            if (resumed) { tb = tb_resume; part = part_resume; resumed = false; }
            const bool csl_now = tb != j0; // re-entry in a later tile of the row: the fine_start slice is staged again
            for (; tb < j1; tb += WCAP, part = 0) {
                const int tn = (int)min((uint32_t)WCAP, j1 - tb);
                if (DBGC(a) && t == 0) atomicAdd(DBGC(a) + 3, 1ull);
                if (csl_ok && (tb == j0 || csl_now))
                    for (int q = t; q < ncs; q += 64) csl[q] = (unsigned short)(sd.fine_start[rowb + xa + q] - j0);
                for (int k = t; k < tn + 8; k += 64) {
                    float vx = 3.0e18f, vy = 3.0e18f, vz = 3.0e18f, vw = 0.f;
                    if (k < tn) {
                        const float4 fj = a.fpos[sd.off + tb + k];
                        vx = fj.x - oxf; vy = fj.y - oyf; vz = fj.z - ozf;
                        const float hjf = fj.w * 1.000001f + slack;
                        vw = hjf * hjf;
                    }
                    tx[k] = vx; ty[k] = vy; tz[k] = vz;
                    if (!UH) tw[k] = vw;
                }
                // the tile is private to this wavefront: its LDS accesses execute in
                // program order, the fence only keeps the compiler from reordering them
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                int s0 = 0, len = 0;
                if (inseg) {
                    int lo, hi;
                    if (csl_ok) { lo = (int)(j0 + csl[mycl] - tb); hi = (int)(j0 + csl[mych] - tb); }
                    else { lo = (int)(sd.fine_start[rowb + xa + mycl] - tb); hi = (int)(sd.fine_start[rowb + xa + mych] - tb); }
                    lo = max(lo, 0); hi = min(hi, tn);
                    s0 = lo & ~1;
                    len = hi - s0;
                }
                // A lane's range of this tile in parts of AMAXLEN candidates (one part unless the rows are
                // unusually dense: a second part takes the same path, nothing is evaluated outside phase 2)
                for (;; part++) {
                    const int plen = len - AMAXLEN * part;
                    const int ps0 = plen > 0 ? s0 + AMAXLEN * part : 0; // lanes without candidates in this part read the tile's start
                    if (part > 0 && !__any(plen > 0)) break;
                    const int lenc = ABLATE(a) == 7 ? 0 : min(plen, AMAXLEN); // 7 (profiling): staging only
                    // One sign bit per candidate: d = |x_i - x_j|^2 - thr^2 in packed fp32 FMAs,
                    // shifted into a 32-bit word with v_alignbit (1 VALU per candidate).
                    uint32_t wd[3] = {0u, 0u, 0u};
#pragma unroll
                    for (int gw = 0; gw < 3; gw++) {
                        if (!__any(32 * gw < lenc)) break; // wave-uniform
                        uint32_t mm = 0;
                        int g8 = 0;
                        for (; g8 < 4 && __any(32 * gw + 8 * g8 < lenc); g8++) {
                            const float *tb0 = tile + (ps0 + 32 * gw + 8 * g8); // one address, constant offsets below
#pragma unroll
                            for (int p = 0; p < 4; p++) {
                                const f2 X = *reinterpret_cast<const f2 *>(tb0 + 2 * p);
                                const f2 Y = *reinterpret_cast<const f2 *>(tb0 + TS + 2 * p);
                                const f2 Z = *reinterpret_cast<const f2 *>(tb0 + 2 * TS + 2 * p);
                                const f2 ex = fx - X, ey = fy - Y, ez = fz - Z;
                                f2 nthr = {-hi2f, -hi2f};
                                if (!UH) {
                                    const f2 W = *reinterpret_cast<const f2 *>(tb0 + 3 * TS + 2 * p);
                                    nthr.x = -fmaxf(hi2f, W.x); // r2 < hi^2 or r2 < hj^2
                                    nthr.y = -fmaxf(hi2f, W.y);
                                }
                                f2 d = __builtin_elementwise_fma(ex, ex, nthr);
                                d = __builtin_elementwise_fma(ey, ey, d);
                                d = __builtin_elementwise_fma(ez, ez, d);
                                mm = __builtin_amdgcn_alignbit(mm, __float_as_uint(d.x), 31);
                                mm = __builtin_amdgcn_alignbit(mm, __float_as_uint(d.y), 31);
                            }
                        }
                        if (g8 < 4) mm <<= 8 * (4 - g8);
                        mm = __builtin_bitreverse32(mm); // bit b <-> candidate 32*gw + b
                        // candidates beyond this lane's range (its own tail / other lanes' longer ranges)
                        const int rem = lenc - 32 * gw;
                        wd[gw] = rem >= 32 ? mm : (rem > 0 ? (mm & ((1u << rem) - 1u)) : 0u);
                    }
                    unsigned long long m0 = (unsigned long long)wd[0] | ((unsigned long long)wd[1] << 32);
                    uint32_t m2 = wd[2];
                    uint32_t jb0 = sd.off + tb + (uint32_t)ps0;
                    // Hits beyond bit 63 would cost a second slot although a lane's hits in one row
                    // span less than 64 candidates: shift the 96 bits down to the lane's first hit
                    // (wide windows only: QuinticSpline / radius_scale 3, where slots run out otherwise)
                    if (a.norm_masks && __any(m2 != 0)) {
                        const int sh = m0 ? __builtin_ctzll(m0) : (m2 ? 64 + __builtin_ctz(m2) : 0);
                        if (sh >= 64) { m0 = (unsigned long long)(m2 >> (sh - 64)); m2 = 0; }
                        else if (sh > 0) {
                            m0 = (m0 >> sh) | ((unsigned long long)m2 << (64 - sh));
                            m2 = sh >= 32 ? 0u : (m2 >> sh);
                        }
                        jb0 += (uint32_t)sh;
                    }
                    // a lane without room for this part's slots: leave, work the lists off, come back here (rare)
                    if (__any(cq + (m0 != 0) + (m2 != 0) > WLQ)) { more = true; break; }
                    if (m0) { smask[cq][t] = m0; sjb[cq][t] = jb0; cq++; }
                    if (m2) { smask[cq][t] = m2; sjb[cq][t] = jb0 + 64u; cq++; }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // tile reads before the next tile's writes
                if (more) { tb_resume = tb; part_resume = part; break; }
            }
            if (more) break;
        }
        if (more) break; // R and st keep their values
    }
    if (a.nl_mode == 1 && s == 0) {
        // keep the lists for the next pass of this evaluation over the same (destination, source)
        if (unsaved || more) nlw[t] = NL_NONE;
        else {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            nlw[t] = (uint32_t)cq;
            const int nqmax = wave_max_i32(cq);
            for (int q = 0; q < nqmax; q++) {
                uint32_t *w = nlw + 64 * (1 + 3 * q) + t;
                const unsigned long long m = q < cq ? smask[q][t] : 0ull;
                w[0] = (uint32_t)m; w[64] = (uint32_t)(m >> 32);
                w[128] = q < cq ? sjb[q][t] - sd.off : 0u;
            }
        }
    }
    phase2(fl);
    if (!more) break;
    unsaved = true;
    resumed = true;
    }
    }
    if (active) {
        if constexpr (fam_merged<Fam>::value) Fam::finish(D, a, o, slot);
        else Fam::finish(D, a, o);
    }
}


// ---------------------------------------------------------------------------
// k_pair_rowlds (round 6, an experiment behind the option "row_lds"): the same two phases, but a row tile's source
// RECORDS are staged in LDS next to its fp32 positions -- one coalesced copy of consecutive 16-B pieces -- and the
// tile's hits are evaluated from LDS right behind its phase 1, instead of being listed for ONE phase 2 over all nine
// rows that gathers every hit's pieces from L1.  For families whose phase 2 is bound by those gathers (the elastic
// rates: ten 16-B pieces per hit, TA 81 % busy, VALU 42 %): a candidate of a row tile is hit by ~3 of the
// wavefront's 64 destinations, so the tile costs one third of the L1 traffic, all of it coalesced.  The price: the
// wavefront runs nine short pair loops (per loop: as many trips as its busiest lane has hits in THAT row) instead
// of one long one.  Uniform h, one part per tile (WCAP_L <= AMAXLEN), no neighbour-list reuse, no split launches.
// Measured: DESIGN.md section 4, profiles/r06_ab_tables.txt.
// ---------------------------------------------------------------------------
template <class F, class = void> struct fam_rowlds { static constexpr bool value = false; };
template <class F> struct fam_rowlds<F, decltype((void)F::ROWLDS)> { static constexpr bool value = F::ROWLDS; };

#ifndef SPH_WCAP_L
#define SPH_WCAP_L 96
#endif
#define WCAP_L SPH_WCAP_L

template <class Fam, int KK, uint32_t CF = 0>
__global__ __launch_bounds__(64, Fam::MINB) void k_pair_rowlds(PairArgs<Fam> a)
{
    typedef typename Fam::Real T;
    typedef typename Fam::Piece Piece;
    constexpr int NP = Fam::PIECES;
    constexpr int TS = WCAP_L + 8;
    static_assert(WCAP_L <= AMAXLEN && WCAP_L % 8 == 0, "one part per tile");
    __shared__ __attribute__((aligned(16))) float tile[3 * TS];
    __shared__ unsigned short csl[WCSL];
    __shared__ __attribute__((aligned(16))) Piece recs[WCAP_L * NP];
    float *const tx = tile, *const ty = tile + TS, *const tz = tile + 2 * TS;

    const int t = threadIdx.x & 63;
    const uint32_t wt = xcd_tile(blockIdx.x, gridDim.x);
    uint32_t dtile = wt >> 2;
    if (dtile * 256u >= a.nd) return;
    if (a.d_tile_order) dtile = a.d_tile_order[dtile];
    const uint32_t tbase = dtile * 256u + (wt & 3u) * 64u;
    const uint32_t i = tbase + t;
    if (tbase >= a.nd) return;
    const bool valid = i < a.nd;
    uint32_t ic = valid ? i : a.nd - 1;
    if (a.d_list) ic = a.d_list[ic];
    const uint32_t o = a.d_perm[ic];
    const bool active = valid && o >= a.d_start && o < a.d_stop;
    const uint32_t fkey = a.d_fkeys[ic];
    const uint32_t key = fkey / SPH_NSUB;
    const int ncx = a.nc[0], ncy = a.nc[1], ncz = a.nc[2];
    const int row = key / ncx;
    const int cx = (int)(key % ncx) * SPH_NSUB + (int)(fkey % SPH_NSUB);
    if (!__any(active)) return;
    bool facew = true;
    if (a.gfx_lo > -0x7fffffff || a.gfx_hi < 0x7fffffff) {
        int mn = active ? cx : 0x7fffffff, mx = active ? cx : -0x7fffffff;
#pragma unroll
        for (int o2 = 32; o2 > 0; o2 >>= 1) { mn = min(mn, __shfl_xor(mn, o2, 64)); mx = max(mx, __shfl_xor(mx, o2, 64)); }
        mn = __builtin_amdgcn_readfirstlane(mn); mx = __builtin_amdgcn_readfirstlane(mx);
        facew = mn - XWIN <= a.gfx_lo || mx + XWIN >= a.gfx_hi;
    }
    real4<T> pi;
    typename Fam::Dest D;
    uint32_t wtok = 1u;
    if constexpr (fam_token<Fam>::value) wtok = Fam::wave_token(a);
    {
        T sd_[Fam::NA];
        Fam::decode(reinterpret_cast<const Piece *>(a.rec) + (unsigned long long)(a.d_off + ic) * NP, (T)a.d_mu, pi, sd_, wtok);
        pi.w = (T)a.hu;
        Fam::load(D, sd_, a, o);
    }
    const int nfx = ncx * SPH_NSUB;
    const T hi_r = (T)a.radius_scale * pi.w;
    const T hi2 = (T)a.hr2u;
    const int row_first = __builtin_amdgcn_readfirstlane(row), row_last = __builtin_amdgcn_readlane(row, 63);
    const float4 fpi = a.fpos[a.d_off + ic];

    for (int s = 0; s < a.nsrc; s++) {
        const SrcDesc sd = a.src[s];
        if (sd.ghost && !facew) continue;
        const uint32_t fl = CF ? CF : sd.flags;
        const T mu = (T)sd.mu;
        for (int R = row_first; R <= row_last; R++) {
            const bool inseg = active && row == R;
            const unsigned long long segm = __ballot(inseg);
            if (!segm) continue;
            const int cxa = __builtin_amdgcn_readlane(cx, __builtin_ctzll(segm));
            const int cxb = __builtin_amdgcn_readlane(cx, 63 - __builtin_clzll(segm));
            const int cyR = R % ncy, czR = R / ncy;
            const int xa = max(cxa - XWIN, 0), xb = min(cxb + XWIN, nfx - 1);
            const int ncs = xb - xa + 2;
            const double binw = a.cell_size * (1.0 / SPH_NSUB);
            const float oxf = (float)(binw * xa);
            const float oyf = (float)(a.cell_size * (cyR - 1));
            const float ozf = (float)(a.cell_size * (czR - 1));
            const double L = fmax(a.cell_size * (double)max((xb - xa) / SPH_NSUB + 2, 4), a.dom_extent);
            const float slack = (float)(L * 1.5e-6);
            const float fxs = fpi.x - oxf, fys = fpi.y - oyf, fzs = fpi.z - ozf;
            const f2 fx = {fxs, fxs}, fy = {fys, fys}, fz = {fzs, fzs};
            const float hif = (float)hi_r * 1.000001f + slack;
            const float hi2f = hif * hif;
            const int mycl = max(cx - XWIN, xa) - xa, mych = min(cx + XWIN, xb) + 1 - xa;
            for (int st = 0; st < 9; st++) {
                const int sy = st % 3 - 1, sz = st / 3 - 1;
                const int dy = (a.row_mod3 & 1) ? (sy + 1 - cyR % 3 + 4) % 3 - 1 : sy;
                const int dz = (a.row_mod3 & 2) ? (sz + 1 - czR % 3 + 4) % 3 - 1 : sz;
                const int yy = cyR + dy, zz = czR + dz;
                if (yy < 0 || yy >= ncy || zz < 0 || zz >= ncz) continue;
                const uint32_t rowb = (uint32_t)(ncx * (yy + ncy * zz)) * SPH_NSUB;
                const uint32_t j0 = sd.fine_start[rowb + xa], j1 = sd.fine_start[rowb + xb + 1];
                const bool csl_ok = ncs <= WCSL && j1 - j0 < 65536u;
                for (uint32_t tb = j0; tb < j1; tb += WCAP_L) {
                    const int tn = (int)min((uint32_t)WCAP_L, j1 - tb);
                    if (csl_ok && tb == j0)
                        for (int q = t; q < ncs; q += 64) csl[q] = (unsigned short)(sd.fine_start[rowb + xa + q] - j0);
                    for (int k = t; k < tn + 8; k += 64) {
                        float vx = 3.0e18f, vy = 3.0e18f, vz = 3.0e18f;
                        if (k < tn) {
                            const float4 fj = a.fpos[sd.off + tb + k];
                            vx = fj.x - oxf; vy = fj.y - oyf; vz = fj.z - ozf;
                        }
                        tx[k] = vx; ty[k] = vy; tz[k] = vz;
                    }
                    { // the tile's records: consecutive pieces of the packed buffer, lane after lane
                        const Piece *g = reinterpret_cast<const Piece *>(a.rec) + (unsigned long long)(sd.off + tb) * NP;
                        const int npc = tn * NP;
                        for (int k = t; k < npc; k += 64) recs[k] = g[k];
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    int s0 = 0, len = 0;
                    if (inseg) {
                        int lo, hi;
                        if (csl_ok) { lo = (int)(j0 + csl[mycl] - tb); hi = (int)(j0 + csl[mych] - tb); }
                        else { lo = (int)(sd.fine_start[rowb + xa + mycl] - tb); hi = (int)(sd.fine_start[rowb + xa + mych] - tb); }
                        lo = max(lo, 0); hi = min(hi, tn);
                        s0 = lo & ~1;
                        len = max(hi - s0, 0);
                    }
                    uint32_t wd[3] = {0u, 0u, 0u};
#pragma unroll
                    for (int gw = 0; gw < 3; gw++) {
                        if (!__any(32 * gw < len)) break;
                        uint32_t mm = 0;
                        int g8 = 0;
                        for (; g8 < 4 && __any(32 * gw + 8 * g8 < len); g8++) {
                            const float *tb0 = tile + ((len > 0 ? s0 : 0) + 32 * gw + 8 * g8);
#pragma unroll
                            for (int p = 0; p < 4; p++) {
                                const f2 X = *reinterpret_cast<const f2 *>(tb0 + 2 * p);
                                const f2 Y = *reinterpret_cast<const f2 *>(tb0 + TS + 2 * p);
                                const f2 Z = *reinterpret_cast<const f2 *>(tb0 + 2 * TS + 2 * p);
                                const f2 ex = fx - X, ey = fy - Y, ez = fz - Z;
                                const f2 nthr = {-hi2f, -hi2f};
                                f2 d = __builtin_elementwise_fma(ex, ex, nthr);
                                d = __builtin_elementwise_fma(ey, ey, d);
                                d = __builtin_elementwise_fma(ez, ez, d);
                                mm = __builtin_amdgcn_alignbit(mm, __float_as_uint(d.x), 31);
                                mm = __builtin_amdgcn_alignbit(mm, __float_as_uint(d.y), 31);
                            }
                        }
                        if (g8 < 4) mm <<= 8 * (4 - g8);
                        mm = __builtin_bitreverse32(mm);
                        const int rem = len - 32 * gw;
                        wd[gw] = rem >= 32 ? mm : (rem > 0 ? (mm & ((1u << rem) - 1u)) : 0u);
                    }
                    // this tile's hits, from LDS
                    unsigned long long m0 = (unsigned long long)wd[0] | ((unsigned long long)wd[1] << 32);
                    uint32_t m2 = wd[2];
                    while (__any((m0 | m2) != 0)) {
                        const bool has = (m0 | m2) != 0;
                        int j = 0;
                        if (m0) { j = s0 + __builtin_ctzll(m0); m0 &= m0 - 1; }
                        else if (m2) { j = s0 + 64 + __builtin_ctz(m2); m2 &= m2 - 1; }
                        real4<T> pj;
                        T sj[Fam::NA];
                        Fam::decode(recs + j * NP, mu, pj, sj, wtok);
                        const T r2 = r2_exact<T>(pi.x - pj.x, pi.y - pj.y, pi.z - pj.z);
                        const bool pass = has && r2 < hi2;
                        Fam::template pair<KK, true>(D, pi, pj, r2, sj, fl, a, pass);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                }
            }
        }
    }
    if (active) Fam::finish(D, a, o);
}
