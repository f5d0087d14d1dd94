// sph_nnps.hip -- cell-list neighbour search on the device.
//
// Reference behaviour being replaced (pypr/pysph):
//   DomainManager._compute_cell_size_for_binning  pysph/base/nnps_base.pyx:942-978
//   NNPS.update / _compute_bounds                  pysph/base/nnps_base.pyx:1471-1575
//   LinkedListNNPS._get_number_of_cells/_refresh   pysph/base/linked_list_nnps.pyx:293-383
//   LinkedListNNPS._bin (serial push-front list)   pysph/base/linked_list_nnps.pyx:235-286
//
// MI355X design: instead of head/next linked lists (pointer chasing, serial
// build) every array is radix-sorted by flattened cell id; a cell is then the
// contiguous range [cell_start[c], cell_start[c+1]) of the sorted order, and a
// row of cells along x is one contiguous range -- what the pair kernels stream.
// The grid itself (bounds, 1 % padding, cell size, cells per dimension, error
// conditions) is computed on the host in the reference's exact arithmetic.
#include "sph_internal.h"
#include "sph_kernels.h"

#include <algorithm>
#include <cfloat>
#include <cmath>

// ---------------------------------------------------------------------------
// bounds (min/max of x, y, z, h, m), fine keys and the bucket histogram of the sort: ONE pass over the positions
// ---------------------------------------------------------------------------
__device__ inline double wave_min(double v)
{
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ inline double wave_max(double v)
{
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}

struct GridDesc {
    double xmin[3];
    double cell_size;
    // 1 / (a cell wider than cell_size by 4 ulp): the BINNING cell.  A cell coordinate is floor((x - xmin) * inv_cell) --
    // one multiplication where the reference divides (nnps_base.pxd:39-57; three fp64 divisions were 45 of k_bin_keys'
    // ~310 instructions per 64 particles).  Any monotone map onto cells at least radius_scale * hmax wide finds the same
    // neighbours (section 3 of DESIGN.md); the reference's own cell ids are what sph_nnps_info REPORTS, computed apart.
    // Every site that derives a cell from a position uses cell_coord() below.
    double inv_cell;
    int nc[3];
    // PARKING bins behind the grid's own in the tables (or none): where parked padding rows are binned -- cells no
    // destination visits, at the end of the cell order (wavefronts made of them find no active destination)
    uint32_t park_base, park_bins, park_n; // first parking bin, their number, rows of the index range that maps onto them
};

// nnps_base.pxd:39-57 real_to_int = <int>floor(real_val/step), flatten_raw :83-96
// The sort key is FINER than the reference's cell id: key = cell * SPH_NSUB +
// sub, sub = the particle's x position inside its cell in SPH_NSUB sub-bins.
// Cells stay the reference's (cell id = key / SPH_NSUB, cell_start per cell);
// the sub-bins only order the particles of a cell along x, so that a pair
// kernel can cut a destination's candidate range in a row of cells from three
// whole cells to its x window (fine_start per sub-bin).
// A particle outside the grid is clamped into its outermost cells: neighbours
// are found by the distance criterion, and two particles closer than a cell size
// stay in adjacent (or the same) cells under the clamp.
static inline void grid_set_cell(GridDesc &g, double cell_size)
{
    g.cell_size = cell_size;
    g.inv_cell = cell_size > 0.0 ? (1.0 / cell_size) * (1.0 - 4.0 * DBL_EPSILON) : 0.0;
}
// position -> (unclamped) cell coordinate along one axis, in units of the binning cell
__device__ __forceinline__ double cell_coord(double v, double vmin, const GridDesc &g) { return (v - vmin) * g.inv_cell; }

__device__ __forceinline__ uint32_t fine_key(double x, double y, double z, const GridDesc &g)
{
    const double ux = cell_coord(x, g.xmin[0], g);
    int cx = (int)floor(ux);
    int cy = (int)floor(cell_coord(y, g.xmin[1], g));
    int cz = (int)floor(cell_coord(z, g.xmin[2], g));
    int sub = (int)floor((ux - (double)cx) * SPH_NSUB);
    if (cx < 0) { cx = 0; sub = 0; }
    if (cx > g.nc[0] - 1) { cx = g.nc[0] - 1; sub = SPH_NSUB - 1; }
    sub = min(max(sub, 0), SPH_NSUB - 1);
    cy = min(max(cy, 0), g.nc[1] - 1);
    cz = min(max(cz, 0), g.nc[2] - 1);
    return (uint32_t)(cx + g.nc[0] * (cy + g.nc[1] * cz)) * SPH_NSUB + (uint32_t)sub;
}
// ... of particle i: a parked position (a padding row of sph_halo_append_padded / sph_domain_images_padded: never
// anybody's neighbour) gets a key that depends on i only, spread over the parking bins behind the grid (no pile-up in one
// bin; measured: spread over the grid's own cells the 1.3 % padding rows of Taylor-Green cost its pair passes 2.6 % --
// idle lanes in every destination tile, a candidate more in every row) or, without parking bins, over the whole table
__device__ __forceinline__ bool is_parked(double x) { return fabs(x) >= SPH_PARKED_MIN; }
__device__ __forceinline__ uint32_t fine_key_of(double x, double y, double z, const GridDesc &g, size_t i)
{
    if (is_parked(x)) {
        // parking bins IN INDEX ORDER (row i of park_n -> bin i * park_bins / park_n): padding rows that are neighbours in
        // memory stay neighbours in the cell order -- hashed over the bins (the first version) every one of them was a
        // random 8-byte gather for the record packing and an atomic group of its own for the sort: Taylor-Green's 0.5 M
        // padding rows cost 0.15 ms of packing and tipped the array into the "no spatial order" traversal
        // ... ONE row per bin (row i -> bin i mod park_bins; round 5 mapped row i of park_n to bin i * park_bins / park_n,
        // 8 consecutive rows per bin: the 10 k padding rows behind a face's ghosts then fell into ONE sort bucket of 1024
        // bins, beyond its LDS stage -- k_bucket_sort 33 -> 76 us on a slab rank of the 16 M dam break)
        if (g.park_bins) return g.park_base + (uint32_t)(i % g.park_bins);
        const unsigned long long n_fine = (unsigned long long)g.nc[0] * g.nc[1] * g.nc[2] * SPH_NSUB;
        return (uint32_t)(((unsigned long long)i * 2654435761ull) % n_fine);
    }
    return fine_key(x, y, z, g);
}

// The particle sort, hand-written for gfx950 (it replaces hipCUB's Onesweep passes and their memsets):
//   fine key = hi : lo, lo = the `lbits` (9..11) low bits.  A BUCKET = the particles of one hi value = 2^lbits
//   consecutive fine keys = 2^(lbits - 3) consecutive cells of a row.
//   k_bin_keys        one pass over x, y, z (+ h, m when they have to be looked at): the min/max partials of this update,
//                     the fine key of every particle and the bucket histogram G (wavefront-aggregated atomics: the 64
//                     consecutive particles of a spatially coherent array hit one or two buckets)
//   k_bin_finish      one workgroup: the partials reduced, G scanned into the bucket starts, G and the cursors left
//                     zeroed for the next sort (no memset anywhere)
//   k_bucket_scatter  (key, index) pairs into their bucket in arrival order (wavefront-aggregated cursor atomics)
//   k_bucket_sort     one workgroup per bucket: counting sort by lo in LDS, every bin put into ascending index order
//                     (= what a stable sort gives: deterministic, bit-identical to the radix sort it replaces) and --
//                     the scan of a bucket's histogram IS its slice of fine_start -- the cell tables and the merged
//                     order's slot / index split written on the way
// Four launches per neighbour update.
#define SORT_LMIN 9
#define SORT_LMAX 11
#define SORT_BK_CAP 3840   // (lo, index) pairs a bucket's workgroup stages in LDS (four workgroups per CU: 4 x 38.6 KB); a larger bucket is sorted in global memory
#define SORT_BIN_SMALL 32  // bins up to this size: insertion sort by one thread (arrival order is nearly index order)
#define SORT_BIGQ 128      // larger ones: queued, ranked by counting by the whole workgroup (an LDS-staged bucket has at most 124)
#define SORT_BS 512        // threads of a k_bucket_sort workgroup
enum { MM_XYZ = 1, MM_H = 2, MM_M = 4 };
// out[]: 8 global values, 4 per array, then
#define MM_OUT_XFLAG 42    // the context's device flag word (sph_ctx::xflag: a non-positive density met by the merged records)
#define MM_OUT_GROUPS 43   // atomic groups k_bin_keys formed (one per bucket a wavefront's 64 particles hit): n / 64 * 1..2 for
                           // particles in cell order, n for particles in no spatial order
#define MM_OUT_N 44

struct BinArrays {   // the arrays of one update, concatenated in slot order
    const double *x[SPH_MAX_ARRAYS], *y[SPH_MAX_ARRAYS], *z[SPH_MAX_ARRAYS], *h[SPH_MAX_ARRAYS], *m[SPH_MAX_ARRAYS];
    uint32_t n[SPH_MAX_ARRAYS], off[SPH_MAX_ARRAYS]; // particles, first position in the concatenation
    uint32_t first[SPH_MAX_ARRAYS + 1];               // workgroups [first[a], first[a + 1]) belong to array a
    // (optional) the order in which an array's particles are VISITED: thread t takes particle via[t] -- the previous
    // update's cell order for an array that lies in memory in no spatial order, so that a wavefront's 64 particles are
    // neighbours again and hit one or two buckets (DevArray::unordered)
    const uint32_t *via[SPH_MAX_ARRAYS];
    int narrays;
};
struct BinWork {
    uint32_t *keys;              // fine key of every particle of the concatenation (null: bounds only)
    uint32_t *G, *bstart, *cur;  // bucket histogram (zero on entry and on exit), bucket starts [nbuckets + 1], cursors (zeroed here)
    double *part, *parta;        // partials per workgroup: [8] {xmin ymin zmin hmin xmax ymax zmax hmax}, [4] {hmin hmax mmin mmax} of its array
    uint32_t *grp;               // ... and its count of atomic groups
    double *out;                 // [0..7] as part, [8 + 4 a ..] {mmin mmax hmin hmax} of array a, [MM_OUT_XFLAG ...]
    const uint32_t *xflag;       // the context's flag word, copied to out[MM_OUT_XFLAG]
    uint32_t nbuckets;
    int lbits, mm;
};

template <class T> __device__ __forceinline__ T wave_incl_scan(T x, int lane)
{
    for (int o = 1; o < 64; o <<= 1) { const T t = __shfl_up(x, o, 64); if (lane >= o) x += t; }
    return x;
}

#define BIN_U 4 // trips of k_bin_keys whose position loads are in flight together
__global__ __launch_bounds__(256) void k_bin_keys(BinArrays t, GridDesc g, BinWork w)
{
    // this workgroup's array (uniform) and its share of it
    int a = 0;
    for (int k = 1; k < SPH_MAX_ARRAYS; k++) if (k < t.narrays && blockIdx.x >= t.first[k]) a = k;
    const double *__restrict__ x = t.x[a], *__restrict__ y = t.y[a], *__restrict__ z = t.z[a];
    const double *__restrict__ h = (w.mm & MM_H) ? t.h[a] : nullptr, *__restrict__ m = (w.mm & MM_M) ? t.m[a] : nullptr;
    const uint32_t n = t.n[a], lb = blockIdx.x - t.first[a], nba = t.first[a + 1] - t.first[a];
    const uint32_t *__restrict__ via = t.via[a];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const bool mmx = (w.mm & MM_XYZ) != 0;
    uint32_t ngroups = 0; // (wave-uniform)
    double mn[5] = {DBL_MAX, DBL_MAX, DBL_MAX, DBL_MAX, DBL_MAX};
    double mx[5] = {-DBL_MAX, -DBL_MAX, -DBL_MAX, -DBL_MAX, -DBL_MAX};
    // BIN_U trips at a time: their position loads (3 x BIN_U per lane) are in flight together -- one trip at a time the
    // kernel was a chain of memory latencies (52 us at 4 M, no shorter at 2.4 M)
    for (size_t ib = (size_t)lb * 256; ib < n; ib += (size_t)nba * 256 * BIN_U) { // the trip count is uniform over the workgroup
        double qx[BIN_U], qy[BIN_U], qz[BIN_U];
        size_t qj[BIN_U];
#pragma unroll
        for (int u = 0; u < BIN_U; u++) {
            const size_t i = ib + (size_t)u * nba * 256 + threadIdx.x;
            const bool valid = i < n;
            qj[u] = valid && via ? (size_t)via[i] : i; // the particle this thread takes
        }
#pragma unroll
        for (int u = 0; u < BIN_U; u++) {
            const bool valid = ib + (size_t)u * nba * 256 + threadIdx.x < n;
            qx[u] = valid ? x[qj[u]] : 0.0; qy[u] = valid ? y[qj[u]] : 0.0; qz[u] = valid ? z[qj[u]] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < BIN_U; u++) {
        const size_t i0 = ib + (size_t)u * nba * 256;
        if (i0 >= n) break; // (uniform)
        const size_t i = i0 + threadIdx.x;
        const bool valid = i < n;
        const double px = qx[u], py = qy[u], pz = qz[u];
        const size_t j = qj[u];
        const bool counts = valid && !is_parked(px); // padding rows (sph_halo_append_padded) are nobody's bounds, h or m
        // (raw_min / raw_max: ONE v_min_f64 / v_max_f64 each; fmin / fmax put a canonicalising v_max x, x in front)
        if (counts && mmx) {
            mn[0] = raw_min(mn[0], px); mx[0] = raw_max(mx[0], px);
            mn[1] = raw_min(mn[1], py); mx[1] = raw_max(mx[1], py);
            mn[2] = raw_min(mn[2], pz); mx[2] = raw_max(mx[2], pz);
        }
        if (counts && h) { const double v = h[j]; mn[3] = raw_min(mn[3], v); mx[3] = raw_max(mx[3], v); }
        if (counts && m) { const double v = m[j]; mn[4] = raw_min(mn[4], v); mx[4] = raw_max(mx[4], v); }
        if (w.keys) {
            uint32_t key = 0;
            if (valid) { key = fine_key_of(px, py, pz, g, (size_t)t.off[a] + j); w.keys[(size_t)t.off[a] + i] = key; }
            const uint32_t d = key >> w.lbits;
            unsigned long long todo = __ballot(valid);
            uint32_t cnt = 0; // this lane leads a group of `cnt` particles of one bucket
            while (todo) {
                const int l = __builtin_ctzll(todo);
                const uint32_t dl = (uint32_t)__builtin_amdgcn_readlane((int)d, l);
                const unsigned long long mk = __ballot(valid && d == dl);
                if (lane == l) cnt = (uint32_t)__builtin_popcountll(mk);
                todo &= ~mk;
                ngroups++;
            }
            // one atomic per bucket the wavefront's particles hit, all of them in ONE instruction (particles in no
            // spatial order make 64 groups: 64 single-lane atomic instructions cost 0.35 ms at 4 M)
            if (cnt) atomicAdd(&w.G[d], cnt);
        }
        }
    }
    __shared__ double s[4][10];
    if (w.keys) { // this workgroup's atomic groups: one plain store (8 k same-address atomics cost 70 us at 4 M)
        __shared__ uint32_t sgrp[4];
        if (lane == 0) sgrp[wv] = ngroups;
        __syncthreads();
        if (threadIdx.x == 0) w.grp[blockIdx.x] = sgrp[0] + sgrp[1] + sgrp[2] + sgrp[3];
    }
    if (w.mm) {
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const double lo = wave_min(mn[k]), hi = wave_max(mx[k]);
            if (lane == 0) {
                if (k < 4) { s[wv][k] = lo; s[wv][4 + k] = hi; }
                else { s[wv][8] = lo; s[wv][9] = hi; }
            }
        }
        __syncthreads();
        if (threadIdx.x < 10) {
            const int k = threadIdx.x;
            const bool is_min = k < 4 || k == 8;
            double v = s[0][k];
            for (int q = 1; q < 4; q++) v = is_min ? fmin(v, s[q][k]) : fmax(v, s[q][k]);
            if (k < 8) w.part[(size_t)blockIdx.x * 8 + k] = v;
            if (k == 3) w.parta[(size_t)blockIdx.x * 4 + 0] = v;
            if (k == 7) w.parta[(size_t)blockIdx.x * 4 + 1] = v;
            if (k >= 8) w.parta[(size_t)blockIdx.x * 4 + 2 + (k - 8)] = v;
        }
    }
}

// One workgroup after k_bin_keys: the partials reduced, the bucket histogram G scanned into the bucket starts, G and the
// cursors left zeroed for the next sort.  (A kernel of its own rather than the last workgroup of k_bin_keys behind a
// ticket: the agent-scope fence that pattern needs writes back an XCD's L2 once per wavefront -- measured 200 us at 4 M.)
__global__ __launch_bounds__(1024) void k_bin_finish(BinArrays t, BinWork w, uint32_t nblk, int have_keys)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (w.mm) {
        __shared__ double sred[128][8];
        const int k = threadIdx.x & 7, gq = threadIdx.x >> 3;
        double v = (k < 4) ? DBL_MAX : -DBL_MAX;
#pragma unroll 8
        for (uint32_t b = gq; b < nblk; b += 128) { // (independent loads: unrolled, they are in flight together)
            const double q = w.part[(size_t)b * 8 + k];
            v = (k < 4) ? fmin(v, q) : fmax(v, q);
        }
        sred[gq][k] = v;
        __syncthreads();
        if (threadIdx.x < 8) {
            for (int q = 1; q < 128; q++) v = (k < 4) ? fmin(v, sred[q][k]) : fmax(v, sred[q][k]);
            w.out[k] = v;
        }
        // the h and m range of every array: one wavefront per array
        if (wv < t.narrays) {
            const int a2 = wv;
            double hl = DBL_MAX, hh = -DBL_MAX, ml = DBL_MAX, mh = -DBL_MAX;
#pragma unroll 8
            for (uint32_t b = t.first[a2] + lane; b < t.first[a2 + 1]; b += 64) {
                hl = fmin(hl, w.parta[(size_t)b * 4 + 0]); hh = fmax(hh, w.parta[(size_t)b * 4 + 1]);
                ml = fmin(ml, w.parta[(size_t)b * 4 + 2]); mh = fmax(mh, w.parta[(size_t)b * 4 + 3]);
            }
            hl = wave_min(hl); hh = wave_max(hh); ml = wave_min(ml); mh = wave_max(mh);
            if (lane == 0) { w.out[8 + 4 * a2] = ml; w.out[9 + 4 * a2] = mh; w.out[10 + 4 * a2] = hl; w.out[11 + 4 * a2] = hh; }
        }
    }
    if (threadIdx.x == 0) w.out[MM_OUT_XFLAG] = (double)*w.xflag;
    if (have_keys) { // atomic groups of the key pass, summed over its workgroups
        __shared__ uint32_t sg[16];
        uint32_t gsum = 0;
        for (uint32_t b = threadIdx.x; b < nblk; b += 1024) gsum += w.grp[b];
        for (int o2 = 32; o2 > 0; o2 >>= 1) gsum += __shfl_xor(gsum, o2, 64);
        if (lane == 0) sg[wv] = gsum;
        __syncthreads();
        if (threadIdx.x == 0) { uint32_t tot = 0; for (int q = 0; q < 16; q++) tot += sg[q]; w.out[MM_OUT_GROUPS] = (double)tot; }
    } else if (threadIdx.x == 0) {
        w.out[MM_OUT_GROUPS] = 0.0;
    }
    if (have_keys) { // bucket starts = exclusive scan of G; G and the cursors zeroed
        // 32 consecutive buckets per thread and round (one round up to 32 Ki buckets: the loads of a round are in flight
        // together; a sparse grid -- the slab of a dam break's tank -- has tens of thousands of buckets)
        __shared__ uint32_t ws[16];
        constexpr int PT = 32;
        uint32_t carry = 0;
        for (uint32_t base = 0; base < w.nbuckets; base += 1024 * PT) {
            uint32_t v[PT], sum = 0;
            const uint32_t i0 = base + threadIdx.x * PT;
#pragma unroll
            for (int q = 0; q < PT; q += 4) {
                if (i0 + q + 3 < w.nbuckets) { // (tables are 16-byte aligned, i0 a multiple of 32)
                    const uint4 t4 = *reinterpret_cast<const uint4 *>(w.G + i0 + q);
                    v[q] = t4.x; v[q + 1] = t4.y; v[q + 2] = t4.z; v[q + 3] = t4.w;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; r++) v[q + r] = i0 + q + r < w.nbuckets ? w.G[i0 + q + r] : 0u;
                }
            }
#pragma unroll
            for (int q = 0; q < PT; q++) sum += v[q];
            const uint32_t incl = wave_incl_scan<uint32_t>(sum, lane);
            if (lane == 63) ws[wv] = incl;
            __syncthreads();
            uint32_t off = carry + incl - sum, tot = 0;
            for (int q = 0; q < 16; q++) { if (q < wv) off += ws[q]; tot += ws[q]; }
#pragma unroll
            for (int q = 0; q < PT; q++) {
                const uint32_t idx = i0 + q;
                if (idx < w.nbuckets) { w.bstart[idx] = off; off += v[q]; w.G[idx] = 0u; w.cur[idx] = 0u; }
            }
            carry += tot;
            __syncthreads();
        }
        if (threadIdx.x == 0) w.bstart[w.nbuckets] = carry;
    }
}

// (key, index) pairs into their buckets, in arrival order.  SCAT_ITEMS keys per thread: the cursor atomics of a
// wavefront's rounds are in flight together (one key per thread left the kernel waiting on one round trip per wavefront).
#define SCAT_ITEMS 4
// via (optional, one array): the particle thread t stands for is via[t] (k_bin_keys visited them in that order)
__global__ __launch_bounds__(256) void k_bucket_scatter(const uint32_t *__restrict__ keys, uint32_t n, int lbits,
                                                        const uint32_t *__restrict__ bstart, uint32_t *__restrict__ cur,
                                                        uint2 *__restrict__ pairs, const uint32_t *__restrict__ via)
{
    const int lane = threadIdx.x & 63;
    const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;
    const uint32_t i0 = blockIdx.x * (256u * SCAT_ITEMS) + threadIdx.x;
    uint32_t key[SCAT_ITEMS], rank[SCAT_ITEMS], base[SCAT_ITEMS], bs[SCAT_ITEMS];
    int leader[SCAT_ITEMS];
#pragma unroll
    for (int q = 0; q < SCAT_ITEMS; q++) { const uint32_t i = i0 + q * 256u; key[q] = i < n ? keys[i] : 0u; }
#pragma unroll
    for (int q = 0; q < SCAT_ITEMS; q++) {
        const bool valid = i0 + q * 256u < n;
        const uint32_t d = key[q] >> lbits;
        unsigned long long todo = __ballot(valid);
        uint32_t cnt = 0;
        leader[q] = lane; rank[q] = 0; base[q] = 0;
        while (todo) {
            const int l = __builtin_ctzll(todo);
            const uint32_t dl = (uint32_t)__builtin_amdgcn_readlane((int)d, l);
            const unsigned long long mk = __ballot(valid && d == dl);
            if (valid && d == dl) { leader[q] = l; rank[q] = (uint32_t)__builtin_popcountll(mk & lt); cnt = (uint32_t)__builtin_popcountll(mk); }
            todo &= ~mk;
        }
        bs[q] = valid ? bstart[d] : 0u;
        if (valid && lane == leader[q]) base[q] = atomicAdd(&cur[d], cnt); // every group's leader at once
    }
#pragma unroll
    for (int q = 0; q < SCAT_ITEMS; q++) {
        const uint32_t i = i0 + q * 256u;
        const uint32_t b = __shfl(base[q], leader[q], 64);
        if (i < n) pairs[(size_t)bs[q] + b + rank[q]] = make_uint2(key[q], via ? via[i] : i);
    }
}

struct CatOff { uint32_t off[SPH_MAX_ARRAYS + 1]; int narrays; }; // first position of every array in the concatenation

struct BucketOut {
    uint32_t *fkeys, *keys, *perm; // sorted fine keys, cell ids (optional), sorted position -> index (in its own array when slot != null)
    uint8_t *slot;                 // merged order of several arrays: the array slot of every sorted particle (else null)
    CatOff co;
    uint32_t *fine_start, *cell_start;
    uint32_t n_fine, n_cells;
    uint2 *scratch;                // n entries (big bins)
};

// One unit of the bucket sort: the `cnt_all` pairs at `src` whose low key bits lie in the sub-range [k_lo, k_hi) of the
// bucket's 2^lbits bins (the whole bucket: [0, NL)) -- counted, scanned, scattered into the LDS stage, every bin put into
// ascending index order, emitted at `out0` + their rank.  WHOLE: the unit is the whole bucket and fits the stage (its
// pairs are loaded once into registers and serve both sweeps); else the pairs are filtered from the bucket twice.
// Returns the number of pairs of the unit.  All threads of the workgroup call it together.
template <int BS, int CAP, bool WHOLE>
__device__ __forceinline__ uint32_t bucket_unit(const uint2 *__restrict__ src, uint32_t cnt_all, uint32_t k_lo, uint32_t k_hi, uint32_t NL,
                                                uint32_t mask, size_t fk0, uint32_t out0, const BucketOut &o, uint32_t *tab, uint2 *stage,
                                                uint32_t *ws, uint32_t *bigq, uint32_t *nbig, bool &overflow)
{
    constexpr int NW = BS / 64, PT = (CAP + BS - 1) / BS, PB = (1 << SORT_LMAX) / BS; // wavefronts; pairs / bins per thread
    const uint32_t tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    uint2 mine[PT];
    if (WHOLE) {
#pragma unroll
        for (int q = 0; q < PT; q++) { const uint32_t i = tid + q * BS; mine[q] = i < cnt_all ? src[i] : make_uint2(0u, 0u); }
    }
    for (uint32_t k = tid; k < NL; k += BS) tab[k] = 0u;
    if (tid == 0) *nbig = 0u;
    __syncthreads();
    if (WHOLE) {
#pragma unroll
        for (int q = 0; q < PT; q++) if (tid + q * BS < cnt_all) atomicAdd(&tab[mine[q].x & mask], 1u);
    } else {
        for (uint32_t i = tid; i < cnt_all; i += BS) { const uint32_t lo = src[i].x & mask; if (lo >= k_lo && lo < k_hi) atomicAdd(&tab[lo], 1u); }
    }
    __syncthreads();
    uint32_t total;
    { // exclusive scan of the bin counts: NL / BS consecutive bins per thread
        const uint32_t per = NL / BS;
        uint32_t v[PB], sum = 0;
#pragma unroll
        for (uint32_t q = 0; q < PB; q++) { v[q] = q < per ? tab[tid * per + q] : 0u; sum += v[q]; }
        const uint32_t incl = wave_incl_scan<uint32_t>(sum, lane);
        if (lane == 63) ws[wv] = incl;
        __syncthreads();
        uint32_t off = incl - sum;
        total = 0;
        for (int q = 0; q < NW; q++) { if (q < wv) off += ws[q]; total += ws[q]; }
#pragma unroll
        for (uint32_t q = 0; q < PB; q++) if (q < per) { tab[tid * per + q] = off; off += v[q]; }
    }
    __syncthreads();
    // the unit's slice of the tables: fine_start[k] = first sorted position whose key >= k (the bucket that holds key
    // n_fine writes the end entry), cell_start[c] = fine_start[c * SPH_NSUB]
    for (uint32_t k = k_lo + tid; k < k_hi; k += BS) {
        const size_t fk = fk0 + k;
        if (fk <= o.n_fine) {
            const uint32_t v = out0 + tab[k];
            o.fine_start[fk] = v;
            if (fk % SPH_NSUB == 0) o.cell_start[fk / SPH_NSUB] = v;
        }
    }
    overflow = total > (uint32_t)CAP; // (uniform: every thread has the same total)
    if (overflow) return total;       // the caller splits the unit further, or sorts it in global memory
    __syncthreads();
    if (WHOLE) {
#pragma unroll
        for (int q = 0; q < PT; q++)
            if (tid + q * BS < cnt_all) {
                const uint32_t lo = mine[q].x & mask, pos = atomicAdd(&tab[lo], 1u);
                stage[pos] = make_uint2(lo, mine[q].y);
            }
    } else {
        for (uint32_t i = tid; i < cnt_all; i += BS) {
            const uint2 p = src[i];
            const uint32_t lo = p.x & mask;
            if (lo >= k_lo && lo < k_hi) { const uint32_t pos = atomicAdd(&tab[lo], 1u); stage[pos] = make_uint2(lo, p.y); }
        }
    }
    __syncthreads();
    // every bin into ascending index order (tab[k] is now the END of bin k)
    for (uint32_t k = k_lo + tid; k < k_hi; k += BS) {
        const uint32_t b0 = k > k_lo ? tab[k - 1] : 0u, b1 = tab[k], c = b1 - b0;
        if (c < 2) continue;
        if (c > SORT_BIN_SMALL) {
            const uint32_t q = atomicAdd(nbig, 1u);
            if (q < SORT_BIGQ) { bigq[q] = k; continue; }
        }
        for (uint32_t i = b0 + 1; i < b1; i++) {
            const uint32_t xv = stage[i].y;
            uint32_t j = i;
            while (j > b0 && stage[j - 1].y > xv) { stage[j].y = stage[j - 1].y; j--; }
            stage[j].y = xv;
        }
    }
    __syncthreads();
    const uint32_t nq = min(*nbig, (uint32_t)SORT_BIGQ);
    for (uint32_t q = 0; q < nq; q++) { // dense bins (coincident particles): rank by counting
        const uint32_t k = bigq[q], b0 = k > k_lo ? tab[k - 1] : 0u, c = tab[k] - b0;
        uint2 *scr = o.scratch + out0 + b0;
        for (uint32_t i = tid; i < c; i += BS) scr[i] = stage[b0 + i];
        __syncthreads();
        for (uint32_t i = tid; i < c; i += BS) {
            const uint32_t xv = scr[i].y;
            uint32_t r = 0;
            for (uint32_t j = 0; j < c; j++) r += scr[j].y < xv;
            stage[b0 + r].y = xv;
        }
        __syncthreads();
    }
    for (uint32_t p = tid; p < total; p += BS) {
        const size_t j = (size_t)out0 + p;
        const uint32_t lo = stage[p].x, gpos = stage[p].y;
        const uint32_t fk = (uint32_t)fk0 + lo;
        o.fkeys[j] = fk;
        if (o.keys) o.keys[j] = fk / SPH_NSUB;
        if (o.slot) {
            uint32_t sl = 0, sb = 0;
#pragma unroll
            for (int b = 1; b < SPH_MAX_ARRAYS; b++)
                if (b < o.co.narrays && gpos >= o.co.off[b]) { sl = (uint32_t)b; sb = o.co.off[b]; }
            o.slot[j] = (uint8_t)sl;
            o.perm[j] = gpos - sb;
        } else {
            o.perm[j] = gpos;
        }
    }
    return total;
}

// The pairs of [k_lo, k_hi) sorted in GLOBAL memory (a sub-range that still exceeds the stage after splitting: thousands of
// particles in a few cells): the output arrays themselves are the stage.  tab[] holds the exclusive starts of the bins
// of the sub-range (bucket_unit left them there).
template <int BS>
__device__ __forceinline__ void bucket_unit_global(const uint2 *__restrict__ src, uint32_t cnt_all, uint32_t k_lo, uint32_t k_hi, uint32_t mask,
                                                   size_t fk0, uint32_t out0, uint32_t total, const BucketOut &o, uint32_t *tab,
                                                   uint32_t *bigq, uint32_t *nbig)
{
    const uint32_t tid = threadIdx.x;
    __syncthreads();
    for (uint32_t i = tid; i < cnt_all; i += BS) {
        const uint2 p = src[i];
        const uint32_t lo = p.x & mask;
        if (lo >= k_lo && lo < k_hi) { const uint32_t pos = atomicAdd(&tab[lo], 1u); o.fkeys[(size_t)out0 + pos] = lo; o.perm[(size_t)out0 + pos] = p.y; }
    }
    __syncthreads();
    uint32_t *pm = o.perm + out0;
    for (uint32_t k = k_lo + tid; k < k_hi; k += BS) {
        const uint32_t b0 = k > k_lo ? tab[k - 1] : 0u, b1 = tab[k], c = b1 - b0;
        if (c < 2) continue;
        if (c > SORT_BIN_SMALL) {
            const uint32_t q = atomicAdd(nbig, 1u);
            if (q < SORT_BIGQ) { bigq[q] = k; continue; }
        }
        for (uint32_t i = b0 + 1; i < b1; i++) {
            const uint32_t xv = pm[i];
            uint32_t j = i;
            while (j > b0 && pm[j - 1] > xv) { pm[j] = pm[j - 1]; j--; }
            pm[j] = xv;
        }
    }
    __syncthreads();
    const uint32_t nq = min(*nbig, (uint32_t)SORT_BIGQ);
    for (uint32_t q = 0; q < nq; q++) {
        const uint32_t k = bigq[q], b0 = k > k_lo ? tab[k - 1] : 0u, c = tab[k] - b0;
        uint2 *scr = o.scratch + out0 + b0;
        for (uint32_t i = tid; i < c; i += BS) scr[i] = make_uint2(k, pm[b0 + i]);
        __syncthreads();
        for (uint32_t i = tid; i < c; i += BS) {
            const uint32_t xv = scr[i].y;
            uint32_t r = 0;
            for (uint32_t j = 0; j < c; j++) r += scr[j].y < xv;
            pm[b0 + r] = xv;
        }
        __syncthreads();
    }
    for (uint32_t p = tid; p < total; p += BS) {
        const size_t j = (size_t)out0 + p;
        const uint32_t lo = o.fkeys[j], gpos = o.perm[j];
        const uint32_t fk = (uint32_t)fk0 + lo;
        o.fkeys[j] = fk;
        if (o.keys) o.keys[j] = fk / SPH_NSUB;
        if (o.slot) {
            uint32_t sl = 0, sb = 0;
#pragma unroll
            for (int b = 1; b < SPH_MAX_ARRAYS; b++)
                if (b < o.co.narrays && gpos >= o.co.off[b]) { sl = (uint32_t)b; sb = o.co.off[b]; }
            o.slot[j] = (uint8_t)sl;
            o.perm[j] = gpos - sb;
        } else {
            o.perm[j] = gpos;
        }
    }
}

// `bpb` consecutive buckets per workgroup: 1 on a dense grid (as many workgroups as buckets); more on a sparse one, whose
// tens of thousands of mostly EMPTY buckets would otherwise each cost a workgroup dispatch.  512 threads and a 3840-pair
// stage: four workgroups per CU.  A bucket beyond the stage (the walls of a tank) is split into sub-ranges of its bins
// that fit (each filtered from the bucket's pairs), and only a sub-range that still does not fit -- thousands of
// particles in a few cells -- is sorted in global memory.
// (MULTI: the one-bucket kernel keeps its 32 registers, the loop costs the other one 64 + spills)
template <bool MULTI, int BS, int CAP>
__global__ __launch_bounds__(BS, 8) void k_bucket_sort(const uint2 *__restrict__ pairs, const uint32_t *__restrict__ bstart, int lbits,
                                                       BucketOut o, uint32_t nbuckets, uint32_t bpb)
{
    __shared__ uint32_t tab[1 << SORT_LMAX];
    __shared__ uint2 stage[CAP];
    __shared__ uint32_t ws[BS / 64], bigq[SORT_BIGQ], nbig;
    const uint32_t tid = threadIdx.x, NL = 1u << lbits, mask = NL - 1u;
    const uint32_t bk0 = MULTI ? blockIdx.x * bpb : blockIdx.x, bk1 = MULTI ? min(nbuckets, (blockIdx.x + 1) * bpb) : blockIdx.x + 1;
    for (uint32_t bk = bk0; bk < bk1; bk++) {
        const uint32_t base = bstart[bk], s = bstart[bk + 1] - base;
        const size_t fk0 = (size_t)bk * NL;
        if (s == 0) { // an empty bucket (the air of a dam break's tank): its slice of the tables is one value
            for (uint32_t k = tid; k < NL; k += BS) {
                const size_t fk = fk0 + k;
                if (fk <= o.n_fine) { o.fine_start[fk] = base; if (fk % SPH_NSUB == 0) o.cell_start[fk / SPH_NSUB] = base; }
            }
            continue;
        }
        bool over = false;
        if (s <= (uint32_t)CAP) {
            bucket_unit<BS, CAP, true>(pairs + base, s, 0u, NL, NL, mask, fk0, base, o, tab, stage, ws, bigq, &nbig, over);
        } else {
            // sub-ranges of the bins: as many as would fit the stage 1.5 times over if the pairs were spread evenly
            uint32_t parts = 2;
            while (parts < NL && (size_t)parts * CAP * 2 < (size_t)3 * s) parts *= 2;
            const uint32_t width = NL / parts;
            uint32_t done = 0;
            for (uint32_t r = 0; r < parts; r++) {
                __syncthreads();
                const uint32_t t = bucket_unit<BS, CAP, false>(pairs + base, s, r * width, (r + 1) * width, NL, mask, fk0, base + done, o, tab,
                                                               stage, ws, bigq, &nbig, over);
                if (over) bucket_unit_global<BS>(pairs + base, s, r * width, (r + 1) * width, mask, fk0, base + done, t, o, tab, bigq, &nbig);
                done += t;
            }
        }
        if (MULTI) __syncthreads(); // the LDS tables serve the workgroup's next bucket
    }
}

// ---------------------------------------------------------------------------
// cell ranges of a sorted key sequence (the per-array tables derived from the merged order), tile order
// ---------------------------------------------------------------------------
// cell_start[c] = fine_start[c * SPH_NSUB]
// Runs after k_fill_gaps: also empties the gap queue for its next user (the queue is empty between uses).
__global__ __launch_bounds__(256) void k_coarse_start(const uint32_t *__restrict__ fine_start, uint32_t n_cells,
                                                      uint32_t *__restrict__ cell_start, uint32_t *__restrict__ gapq)
{
    size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c <= n_cells) cell_start[c] = fine_start[c * SPH_NSUB];
    if (c == 0) gapq[0] = 0u;
}

// Traversal order of the destination tiles (runs of SPH_TILE cell-sorted
// particles) of the aggregated pair kernel.  Memory order is x fastest, then y,
// then z: a tile's z-neighbour rows are a whole plane of tiles away, more than
// the 4 MiB L2 of an XCD holds, so every row used to be fetched from HBM once
// per plane that needs it.  Traversing blocks of `by` rows through ALL planes
// (key = ((cy / by) * ncz + cz) * by + cy % by, ties in memory order) brings
// the reuse distance of a row down to a few hundred KB.  Results do not depend
// on the order.
#define SPH_TILE 256
// (dlist, optional: tile t is the destinations dlist[t * SPH_TILE ...] -- DevArray::dlist -- instead of those positions)
__global__ __launch_bounds__(256) void k_tile_keys(const uint32_t *__restrict__ skeys, size_t n, uint32_t n_tiles, int ncx,
                                                   int ncy, int ncz, int by, uint32_t *__restrict__ key,
                                                   uint32_t *__restrict__ count, const uint32_t *__restrict__ dlist = nullptr)
{
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tiles) return;
    const uint32_t k = skeys[dlist ? (size_t)dlist[(size_t)t * SPH_TILE] : (size_t)t * SPH_TILE];
    const uint32_t row = k / (uint32_t)ncx;
    // A tile of parked padding rows lies in the parking cells behind the grid and has no work: such tiles are dealt over
    // the whole traversal (a key that depends on the tile index only).  Left together at the end of the order they are the
    // LAST XCD's contiguous share of the launch (xcd_tile): one of eight XCDs idle, +6 % on Taylor-Green's pair passes
    // with 9 % padding rows.
    if (row >= (uint32_t)ncy * (uint32_t)ncz) {
        const uint32_t nkeys = ((uint32_t)ncy + (uint32_t)by - 1u) / (uint32_t)by * (uint32_t)by * (uint32_t)ncz;
        const uint32_t kp = (uint32_t)(((unsigned long long)t * 2654435761ull) % nkeys);
        key[t] = kp;
        atomicAdd(&count[kp + 2], 1u);
        return;
    }
    const uint32_t cy = row % (uint32_t)ncy, cz = row / (uint32_t)ncy;
    const uint32_t kk = ((cy / (uint32_t)by) * (uint32_t)ncz + cz) * (uint32_t)by + cy % (uint32_t)by;
    key[t] = kk;
    atomicAdd(&count[kk + 2], 1u); // the bin sort's count pass
}

// start[k] = first sorted position whose key >= k, for k in [0, ntab]: thread i
// owns the boundary between sorted positions i-1 and i and fills the table
// entries in (key[i-1], key[i]] with i.  Short gaps are written by the owning
// lane, medium ones by its wavefront, and the long ones (empty rows / planes of
// a sparsely occupied grid: the obstacle of the dam break occupies 0.1 % of the
// tank's cells) are queued for k_fill_gaps, which spreads each of them over the
// whole grid -- O(n + ntab) stores, no atomics on the table, no per-entry search.
#define GAP_WAVE 64      // gaps of at least this many entries: wave-cooperative
#define GAP_GRID 8192    // ... and of at least this many: queued for k_fill_gaps
#define GAP_QUEUE 4096   // queue capacity (a full queue falls back to the wave-cooperative fill)
// `coarse` (optional): the cell ids of the sorted order, coarse[i] = skeys[i] / SPH_NSUB, written on the way.
__global__ __launch_bounds__(256) void k_cell_start(const uint32_t *__restrict__ skeys, size_t n, uint32_t ntab,
                                                    uint32_t *__restrict__ start, uint32_t *__restrict__ gapq,
                                                    uint32_t *__restrict__ coarse)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    long first = 0, last = -1; // entries first..last get the value i
    if (i <= n) {
        first = i > 0 ? (long)skeys[i - 1] + 1 : 0;
        last = i < n ? (long)skeys[i] : (long)ntab;
        if (coarse && i < n) coarse[i] = (uint32_t)last / SPH_NSUB;
    }
    bool big = last - first >= GAP_WAVE;
    if (!big)
        for (long k = first; k <= last; k++) start[k] = (uint32_t)i;
    if (big && last - first >= GAP_GRID) {
        const uint32_t slot = atomicAdd(&gapq[0], 1u);
        if (slot < GAP_QUEUE) {
            gapq[4 + 3 * slot] = (uint32_t)first; gapq[5 + 3 * slot] = (uint32_t)last; gapq[6 + 3 * slot] = (uint32_t)i;
            big = false;
        }
    }
    unsigned long long m = __ballot(big);
    while (m) {
        const int src = __builtin_ctzll(m);
        m &= m - 1;
        const long f = __shfl(first, src, 64), l = __shfl(last, src, 64);
        const uint32_t v = (uint32_t)__shfl((unsigned long long)i, src, 64);
        for (long k = f + lane; k <= l; k += 64) start[k] = v;
    }
}

__global__ __launch_bounds__(256) void k_fill_gaps(const uint32_t *__restrict__ gapq, uint32_t *__restrict__ start)
{
    const uint32_t ng = min(gapq[0], (uint32_t)GAP_QUEUE);
    for (uint32_t g = 0; g < ng; g++) {
        const size_t f = gapq[4 + 3 * g], l = gapq[5 + 3 * g];
        const uint32_t v = gapq[6 + 3 * g];
        for (size_t k = f + (size_t)blockIdx.x * blockDim.x + threadIdx.x; k <= l; k += (size_t)gridDim.x * blockDim.x) start[k] = v;
    }
}

__global__ __launch_bounds__(256) void k_fill_u32(uint32_t *p, size_t n, uint32_t v)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// ---------------------------------------------------------------------------
// host side of the pass above
// ---------------------------------------------------------------------------
#ifndef BIN_BLOCKS
#define BIN_BLOCKS 2048
#endif
// BIN_BLOCKS: workgroups of one k_bin_keys launch (all arrays together): eight per CU (1536 measured 25 % slower)

static int bin_arrays(sph_ctx *c, int narrays, const int *ids, BinArrays *ba, size_t *n_cat, uint32_t *nblocks)
{
    memset(ba, 0, sizeof *ba);
    ba->narrays = narrays;
    size_t total = 0;
    for (int a = 0; a < narrays; a++) total += c->arr[ids[a]].n;
    if (total >= (1ull << 31)) { sph_set_error("nnps: %zu particles (the sorted index is 31 bits wide)", total); return SPH_ERR_ARG; }
    uint32_t nb = 0, off = 0;
    for (int a = 0; a < narrays; a++) {
        DevArray &A = c->arr[ids[a]];
        ba->first[a] = nb;
        ba->off[a] = off;
        ba->n[a] = (uint32_t)A.n;
        off += (uint32_t)A.n;
        if (A.n == 0) continue;
        for (int p : {SPH_X, SPH_Y, SPH_Z, SPH_H})
            if (!A.prop[p]) {
                sph_set_error("nnps: array %d has no device copy of x/y/z/h", ids[a]);
                return SPH_ERR_MISSING_PROP;
            }
        ba->x[a] = A.prop[SPH_X]; ba->y[a] = A.prop[SPH_Y]; ba->z[a] = A.prop[SPH_Z]; ba->h[a] = A.prop[SPH_H];
        ba->m[a] = A.prop[SPH_M];
        // the array's share of the launch, at least one workgroup, at most one per 256 particles
        size_t share = (size_t)((double)BIN_BLOCKS * (double)A.n / (double)total);
        share = std::max<size_t>(1, std::min<size_t>(share, (A.n + 255) / 256));
        nb += (uint32_t)share;
    }
    for (int a = narrays; a <= SPH_MAX_ARRAYS; a++) ba->first[a] = nb;
    *n_cat = total;
    *nblocks = nb;
    return SPH_OK;
}

// tables of the sort: [G: cap][bstart: cap + 1][cur: cap]; G must be zero between sorts (k_bin_finish leaves it so)
static int sort_tables(sph_ctx *c, uint32_t nbuckets, BinWork *w)
{
    const size_t cap = std::max<size_t>(c->sort_tab_entries, 1024);
    if (!c->sort_tab.ptr || nbuckets + 1 > cap) {
        size_t ncap = std::max<size_t>((size_t)nbuckets + 1 + nbuckets / 4, 1024);
        DevBuf nb_;
        SPH_TRY(nb_.reserve((3 * ncap + 8) * 4));
        HIP_TRY(hipMemsetAsync(nb_.ptr, 0, nb_.bytes, c->stream));
        if (c->sort_tab.ptr) { HIP_TRY(hipStreamSynchronize(c->stream)); c->sort_tab.release(); }
        c->sort_tab = nb_;
        c->sort_tab_entries = ncap;
    }
    uint32_t *t = c->sort_tab.as<uint32_t>();
    const size_t e = c->sort_tab_entries;
    w->G = t; w->bstart = t + e; w->cur = t + 2 * e + 1;
    w->nbuckets = nbuckets;
    return SPH_OK;
}

// One k_bin_keys launch: `mm` = what to reduce (MM_*), `keys` = null (bounds only) or the key buffer of the concatenation.
// The reduced values land in c->red_out (device); the caller copies them where it wants them.
static int launch_bin_keys(sph_ctx *c, const BinArrays &ba, uint32_t nblocks, const GridDesc &g, int mm, uint32_t *keys, uint32_t nbuckets,
                           int lbits)
{
    BinWork w;
    memset(&w, 0, sizeof w);
    SPH_TRY(sort_tables(c, keys ? nbuckets : 1, &w));
    SPH_TRY(c->red_part.reserve((size_t)nblocks * 13 * sizeof(double)));
    SPH_TRY(c->red_out.reserve(64 * sizeof(double)));
    w.keys = keys;
    w.part = c->red_part.as<double>();
    w.parta = w.part + (size_t)nblocks * 8;
    w.grp = reinterpret_cast<uint32_t *>(w.part + (size_t)nblocks * 12);
    w.out = c->red_out.as<double>();
    if (!c->xflag.ptr) {
        SPH_TRY(c->xflag.reserve(16));
        HIP_TRY(hipMemsetAsync(c->xflag.ptr, 0, 16, c->stream));
    }
    w.xflag = c->xflag.as<uint32_t>();
    w.lbits = lbits;
    w.mm = mm;
    hipLaunchKernelGGL(k_bin_keys, dim3(nblocks), dim3(256), 0, c->stream, ba, g, w);
    hipLaunchKernelGGL(k_bin_finish, dim3(1), dim3(1024), 0, c->stream, ba, w, nblocks, keys ? 1 : 0);
    return SPH_OK;
}

// what a look at h / m found, per array
static void note_ranges(sph_ctx *c, int narrays, const int *ids, const double *out, int mm)
{
    for (int a = 0; a < narrays; a++) {
        DevArray &A = c->arr[ids[a]];
        if ((mm & MM_H) && !A.raw_hm) { A.h_seen = true; A.h_dirty = false; A.h_lo = out[10 + 4 * a]; A.h_hi = out[11 + 4 * a]; }
        if (mm & MM_M) {
            A.m_lo = out[8 + 4 * a]; A.m_hi = out[9 + 4 * a];
            if (!A.raw_hm) { A.m_seen = A.prop[SPH_M] != nullptr; A.m_dirty = false; }
            A.m_known = c->want_mrange && A.n > 0 && A.prop[SPH_M] && A.m_lo == A.m_hi && !A.m_mixed_ghosts;
            A.m_value = A.m_lo;
        }
    }
}

// bounds (and h / m ranges) of the arrays, synchronously: out8 = {xmin ymin zmin hmin xmax ymax zmax hmax}
int nnps_minmax(sph_ctx *c, int narrays, const int *ids, double *out8, int mm)
{
    BinArrays ba;
    size_t n_cat = 0;
    uint32_t nblocks = 0;
    SPH_TRY(bin_arrays(c, narrays, ids, &ba, &n_cat, &nblocks));
    if (nblocks == 0) {
        for (int k = 0; k < 4; k++) { out8[k] = DBL_MAX; out8[4 + k] = -DBL_MAX; }
        return SPH_OK;
    }
    GridDesc g;
    memset(&g, 0, sizeof g);
    SPH_TRY(launch_bin_keys(c, ba, nblocks, g, mm, nullptr, 0, SORT_LMAX));
    HIP_TRY(hipMemcpyAsync(c->pinned, c->red_out.ptr, MM_OUT_N * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    memcpy(out8, c->pinned, 8 * sizeof(double));
    note_ranges(c, narrays, ids, c->pinned, mm);
    return SPH_OK;
}

extern "C" int sph_nnps_minmax(sph_ctx *c, int narrays, const int *ids, double *out8)
{
    if (!c || narrays < 1 || narrays > SPH_MAX_ARRAYS) { sph_set_error("sph_nnps_minmax: bad arguments"); return SPH_ERR_ARG; }
    for (int a = 0; a < narrays; a++)
        if (ids[a] < 0 || ids[a] >= SPH_MAX_ARRAYS || !c->arr[ids[a]].used) { sph_set_error("sph_nnps_minmax: array id %d not registered", ids[a]); return SPH_ERR_ARG; }
    HIP_TRY(hipSetDevice(c->device));
    return nnps_minmax(c, narrays, ids, out8, MM_XYZ | MM_H | (c->want_mrange ? MM_M : 0));
}

// The grid of a set of bounds in the reference's arithmetic.  mm = {xmin ymin zmin hmin xmax ymax zmax hmax}.
static int grid_from_minmax(const sph_ctx *c, int dim, double radius_scale, double cell_size_in, const double *bounds, const double *mm,
                            GridHost *G)
{
    // DomainManager._compute_cell_size_for_binning (nnps_base.pyx:942-978)
    double hmax = -1.0, hmin = DBL_MAX;
    if (mm[7] > hmax) hmax = mm[7];
    if (mm[3] < hmin) hmin = mm[3];
    double cell_size = radius_scale * hmax;
    G->hmin = radius_scale * hmin;
    if (cell_size < 1e-6) cell_size = 1.0;
    if (cell_size_in > 0) cell_size = cell_size_in;
    G->uniform_h = (hmin == hmax);
    G->h_uniform = hmax;

    // NNPS._compute_bounds (nnps_base.pyx:1520-1575)
    double xmax = fmax(mm[4], -1e100), ymax = fmax(mm[5], -1e100), zmax = fmax(mm[6], -1e100);
    double xmin = fmin(mm[0], 1e100), ymin = fmin(mm[1], 1e100), zmin = fmin(mm[2], 1e100);
    // ghost split: room for the ghosts that arrive after this update (sph_nnps_set_extend)
    xmin -= c->extend[0]; xmax += c->extend[0];
    ymin -= c->extend[1]; ymax += c->extend[1];
    zmin -= c->extend[2]; zmax += c->extend[2];
    double lx = xmax - xmin, ly = ymax - ymin, lz = zmax - zmin;
    xmin -= lx * 0.01; ymin -= ly * 0.01; zmin -= lz * 0.01;
    xmax += lx * 0.01; ymax += ly * 0.01; zmax += lz * 0.01;
    const double eps = 1e-12;
    if (fabs(xmax - xmin) < eps && fabs(ymax - ymin) < eps && fabs(zmax - zmin) < eps) {
        xmin -= 0.5; xmax += 0.5;
        ymin -= 0.5; ymax += 0.5;
        zmin -= 0.5; zmax += 0.5;
    }
    if (bounds) {
        xmin = bounds[0]; ymin = bounds[1]; zmin = bounds[2];
        xmax = bounds[3]; ymax = bounds[4]; zmax = bounds[5];
    }

    // LinkedListNNPS._get_number_of_cells (linked_list_nnps.pyx:293-326)
    double cell_size1 = 1. / cell_size;
    int ncx = (int)ceil(cell_size1 * (xmax - xmin));
    int ncy = (int)ceil(cell_size1 * (ymax - ymin));
    int ncz = (int)ceil(cell_size1 * (zmax - zmin));
    if (ncx < 0 || ncy < 0 || ncz < 0) {
        sph_set_error("LinkedListNNPS: Number of cells is negative (%d, %d, %d).", ncx, ncy, ncz);
        return SPH_ERR_CELLS;
    }
    ncx = ncx == 0 ? 1 : ncx;
    ncy = ncy == 0 ? 1 : ncy;
    ncz = ncz == 0 ? 1 : ncz;
    long n_cells = ncx;
    if (dim == 2) n_cells = (long)ncx * ncy;
    if (dim == 3) n_cells = (long)ncx * ncy * ncz;
    // _count_occupied_cells (:328-343)
    if (n_cells < 0 || n_cells > (1L << 28)) {
        sph_set_error("ERROR: LinkedListNNPS requires too many cells (%ld).", n_cells);
        return SPH_ERR_CELLS;
    }
    // The reference indexes head[] with the full 3-D flattened id even when
    // dim < 3; particles of a dim<3 problem lie in one z (and y) plane so the
    // id stays < n_cells.  Keys here use the same flattening; the table is
    // sized for the full product so that a stray plane cannot overflow it.
    long n_cells_alloc = (long)ncx * ncy * ncz;
    if (n_cells_alloc > (1L << 28)) {
        sph_set_error("ERROR: LinkedListNNPS requires too many cells (%ld).", n_cells_alloc);
        return SPH_ERR_CELLS;
    }
    G->cell_size = cell_size;
    G->xmin[0] = xmin; G->xmin[1] = ymin; G->xmin[2] = zmin;
    G->xmax[0] = xmax; G->xmax[1] = ymax; G->xmax[2] = zmax;
    G->nc[0] = ncx; G->nc[1] = ncy; G->nc[2] = ncz;
    G->n_cells = n_cells;
    G->n_cells_alloc = n_cells_alloc;
    return SPH_OK;
}

// low key bits sorted inside a bucket.  Round 5 took the most the tables allow unless the MEAN bucket would not fit the
// stage ("a bucket costs its workgroup ~13 us of latency whatever its size, so fewer and larger buckets win").  Round 6
// measured the other side (tools/debug/lbits2.sh, nnps ms at 9 / 10 / 11 bits): the 4 M cube 0.145 / 0.154 / 0.204, the
// 1 M cube 0.075 / 0.088 / 0.135, a 2.4 M-row slab rank of the 17.3 M tank 0.123 / 0.132 / 0.160 -- few large buckets
// leave the chip idle and spill the stage in the dense part of a sparse grid -- but the 4.65 M tank 0.214 / 0.190 / 0.189
// and the 17.3 M one 0.588 / 0.502 / 0.491: what costs there is the NUMBER of buckets, the empty ones of a tank that is
// 60 % air included (tables, the scan, one workgroup each).  So: the fewest bits that keep the bucket count under
// SORT_NB_MAX, then fewer still if the mean bucket would not fit the stage (dense grids: wide kernels).
// Feedback from the largest bucket of the previous sort was tried and removed: the walls of a sparse tank drove the
// bucket size down for everybody (47 k buckets instead of 12 k: 0.27 -> 0.40 ms).
#define SORT_NB_MAX 6144
static int sort_choose_lbits(sph_ctx *c, size_t n, size_t n_fine)
{
    if (!c->hand_sort && c->sort_lbits >= SORT_LMIN && c->sort_lbits <= SORT_LMAX) return c->sort_lbits; // option sort_lbits
    const double per_key = (double)n / (double)std::max<size_t>(n_fine, 1);
    int lb = SORT_LMIN;
    while (lb < SORT_LMAX && (n_fine >> lb) > SORT_NB_MAX) lb++;
    while (lb > SORT_LMIN && per_key * (double)(1u << lb) > 0.6 * SORT_BK_CAP) lb--;
    c->sort_lbits = lb;
    return lb;
}

struct SortDest { DevArray *T; bool merged; CatOff co; const uint32_t *via = nullptr; };

// keys (already in c->tmp_u32a) -> sorted order + tables of T
static int sort_finish(sph_ctx *c, size_t n, size_t n_fine, long n_cells, int lbits, uint32_t nbuckets, const SortDest &d)
{
    BinWork w;
    SPH_TRY(sort_tables(c, nbuckets, &w));
    DevArray &T = *d.T;
    uint2 *pairs = c->tmp_u32b.as<uint2>();
    hipLaunchKernelGGL(k_bucket_scatter, dim3(div_up(n, 256 * SCAT_ITEMS)), dim3(256), 0, c->stream, c->tmp_u32a.as<uint32_t>(), (uint32_t)n, lbits,
                       (const uint32_t *)w.bstart, w.cur, pairs, d.via);
    BucketOut o;
    memset(&o, 0, sizeof o);
    o.fkeys = T.fkeys_sorted.as<uint32_t>(); o.keys = T.keys_sorted.as<uint32_t>(); o.perm = T.perm.as<uint32_t>();
    o.slot = d.merged ? T.slot8.as<uint8_t>() : nullptr;
    o.co = d.co;
    o.fine_start = T.fine_start.as<uint32_t>(); o.cell_start = T.cell_start.as<uint32_t>();
    o.n_fine = (uint32_t)n_fine; o.n_cells = (uint32_t)n_cells;
    o.scratch = c->tmp_u32a.as<uint2>();
    // Two variants for grids whose buckets hold a few hundred particles (the end slab of a dam-break tank: 11.8 k buckets
    // of 250, 24.5 M table entries for 2.9 M particles) were built, measured and removed: a small workgroup shape (128
    // threads, 1024-pair stage, ten per CU: the same 150 us) and one WAVEFRONT per small bucket with LDS areas of its own and
    // no barrier (0.258 -> 0.326 ms there, slower on every other workload too).  What such a grid pays for is its TABLE --
    // 98 MB of fine_start + cell_start written per update for 24 MB of keys --, not the buckets' barrier chains.
    const uint32_t bpb = std::min<uint32_t>(16u, std::max<uint32_t>(1u, nbuckets / 4096u));
    const uint2 *pp = (const uint2 *)pairs;
    const uint32_t *bs = (const uint32_t *)w.bstart;
    const dim3 grid(div_up(nbuckets, bpb));
    if (bpb > 1) hipLaunchKernelGGL((k_bucket_sort<true, SORT_BS, SORT_BK_CAP>), grid, dim3(SORT_BS), 0, c->stream, pp, bs, lbits, o, nbuckets, bpb);
    else hipLaunchKernelGGL((k_bucket_sort<false, SORT_BS, SORT_BK_CAP>), grid, dim3(SORT_BS), 0, c->stream, pp, bs, lbits, o, nbuckets, 1u);
    return SPH_OK;
}

// ---------------------------------------------------------------------------
// Hand-written prefix sums (dev_scan_u32 / _u64: the halo selection, the neighbour-list starts, the compaction below)
// and a hand-written COUNTING sort for keys that are bin ids of a small table (the traversal order of the tiles):
//   count   the key pass adds 1 to T[key + 2]                      (T: nbins + 2 entries, zeroed)
//   scan    inclusive, in place: T[k + 1] = first position of bin k
//   scatter pos = atomicAdd(&T[key + 1], 1): afterwards T[k] = first position of bin k for every k -- T IS fine_start
//   fix     the atomics hand out the positions of a bin in arrival order: one thread per bin puts its (one to three)
//           particles into ascending index order, i.e. the order a stable sort gives -- deterministic, bit-identical
//           between runs and to the radix sort this replaces -- and writes the sorted keys / cell ids / cell_start (and,
//           for the merged order of several arrays, slot and local index) on the way.  Bins of more than BIN_SMALL
//           particles (dense cells, coincident particles) are queued for one workgroup each (rank by counting).
// The PARTICLE sort stays a radix sort (hipCUB Onesweep): the counting sort through the fine_start table was built and
// measured for it too -- bit-identical results, no library kernel, and slower: with ~2 particles per x sub-bin and the
// particles nearly in cell order, a wavefront's 64 table updates hit one or two 128-byte lines and the L2 serialises
// them (4 M atomics: 112 us in the key pass + 176 us in the scatter against 120 us for three radix passes;
// profiles/r04_binsort_vs_radix.txt).
// ---------------------------------------------------------------------------
#define SCAN_ITEMS 16
#define SCAN_BLOCK (256 * SCAN_ITEMS)

template <class T>
__global__ __launch_bounds__(256) void k_scan_partials(const T *__restrict__ in, size_t n, T *__restrict__ partial)
{
    const size_t base = (size_t)blockIdx.x * SCAN_BLOCK + threadIdx.x;
    T sum = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) if (base + (size_t)k * 256 < n) sum += in[base + (size_t)k * 256];
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    __shared__ T ws[4];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

// exclusive scan of the block partials, in place: one workgroup walks them 1024 at a time
template <class T> __global__ __launch_bounds__(1024) void k_scan_spine(T *__restrict__ part, uint32_t nb)
{
    __shared__ T ws[17];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    T carry = 0;
    for (uint32_t base = 0; base < nb; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const T v = i < nb ? part[i] : T(0);
        const T x = wave_incl_scan<T>(v, lane);
        if (lane == 63) ws[wave] = x;
        __syncthreads();
        if (wave == 0) {
            const T w = lane < 16 ? ws[lane] : T(0);
            const T y = wave_incl_scan<T>(w, lane);
            if (lane < 16) ws[lane] = y - w;
            if (lane == 15) ws[16] = y;
        }
        __syncthreads();
        if (i < nb) part[i] = carry + ws[wave] + x - v;
        carry += ws[16];
        __syncthreads();
    }
}

// out[i] = partial[block] + scan of the block's items (inclusive, or exclusive), SCAN_ITEMS rounds of 256 coalesced
// items with the running total carried; in == out allowed (an item is read before it is written, by the same thread)
template <class T>
__global__ __launch_bounds__(256) void k_scan_apply(const T *in, T *out, size_t n, const T *__restrict__ partial, int exclusive)
{
    __shared__ T ws[2][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    T carry = partial[blockIdx.x];
    for (int r = 0; r < SCAN_ITEMS; r++) {
        const size_t i = (size_t)blockIdx.x * SCAN_BLOCK + (size_t)r * 256 + threadIdx.x;
        const T v = i < n ? in[i] : T(0);
        const T x = wave_incl_scan<T>(v, lane);
        if (lane == 63) ws[r & 1][wave] = x;
        __syncthreads();
        T off = carry;
        for (int w = 0; w < wave; w++) off += ws[r & 1][w];
        if (i < n) out[i] = off + x - (exclusive ? v : T(0));
        carry += ws[r & 1][0] + ws[r & 1][1] + ws[r & 1][2] + ws[r & 1][3];
    }
}

// up to 64 Ki counters: ONE workgroup (1024 threads, 16 consecutive items each per round, the loads of a round in flight
// together) instead of the three launches below -- the block counters of a ghost selection (n / 256 of them), the bins
// of the tile order: launch gaps, not bytes
template <class T> __global__ __launch_bounds__(1024) void k_scan_small(const T *in, T *out, uint32_t n, int exclusive)
{
    __shared__ T ws[16];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    constexpr int PT = 16;
    T carry = 0;
    for (uint32_t base = 0; base < n; base += 1024 * PT) {
        T v[PT], sum = 0;
        const uint32_t i0 = base + threadIdx.x * PT;
#pragma unroll
        for (int q = 0; q < PT; q++) { v[q] = i0 + q < n ? in[i0 + q] : T(0); sum += v[q]; }
        const T incl = wave_incl_scan<T>(sum, lane);
        if (lane == 63) ws[wv] = incl;
        __syncthreads();
        T off = carry + incl - sum, tot = 0;
        for (int q = 0; q < 16; q++) { if (q < wv) off += ws[q]; tot += ws[q]; }
#pragma unroll
        for (int q = 0; q < PT; q++) {
            if (i0 + q < n) out[i0 + q] = exclusive ? off : off + v[q];
            off += v[q];
        }
        carry += tot;
        __syncthreads();
    }
}

// prefix sums of n counters (device-wide, three launches; one for up to 64 Ki); in == out allowed
template <class T> static int dev_scan(sph_ctx *c, const T *in, T *out, size_t n, bool exclusive)
{
    if (n == 0) return SPH_OK;
    if (n <= 65536) {
        hipLaunchKernelGGL(k_scan_small<T>, dim3(1), dim3(1024), 0, c->stream, in, out, (uint32_t)n, exclusive ? 1 : 0);
        return SPH_OK;
    }
    const uint32_t nblk = (uint32_t)div_up(n, SCAN_BLOCK);
    SPH_TRY(c->scan_part.reserve(((size_t)nblk + 64) * sizeof(T)));
    T *part = c->scan_part.as<T>();
    hipLaunchKernelGGL(k_scan_partials<T>, dim3(nblk), dim3(256), 0, c->stream, in, n, part);
    hipLaunchKernelGGL(k_scan_spine<T>, dim3(1), dim3(1024), 0, c->stream, part, nblk);
    hipLaunchKernelGGL(k_scan_apply<T>, dim3(nblk), dim3(256), 0, c->stream, in, out, n, (const T *)part, exclusive ? 1 : 0);
    return SPH_OK;
}
int dev_scan_u32(sph_ctx *c, const uint32_t *in, uint32_t *out, size_t n, bool exclusive) { return dev_scan<uint32_t>(c, in, out, n, exclusive); }
int dev_scan_u64(sph_ctx *c, const unsigned long long *in, unsigned long long *out, size_t n, bool exclusive)
{
    return dev_scan<unsigned long long>(c, in, out, n, exclusive);
}

// pos = T[key + 1]++ ; perm[pos] = position of the particle in the key array
__global__ __launch_bounds__(256) void k_bin_scatter(const uint32_t *__restrict__ keys, size_t n, uint32_t *__restrict__ table,
                                                     uint32_t *__restrict__ perm)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t pos = atomicAdd(&table[keys[i] + 1], 1u);
    perm[pos] = (uint32_t)i;
}

#define BIN_SMALL 64     // bins up to this size: insertion sort by their thread (arrival order is nearly index order)
#define BIN_QUEUE 65536  // larger ones: one workgroup each (a full queue falls back to the thread)
struct BinFixArgs {
    const uint32_t *table;   // nbins + 1 first positions
    uint32_t nbins, nsub;    // nsub: bins per cell (cell_start[c] = table[c * nsub]); 0: no cells
    uint32_t *perm;          // sorted position -> position in the key array; rewritten to the local index when slot != null
    uint32_t *fkeys, *keys;  // optional: sorted fine keys / cell ids
    uint32_t *cell_start;    // optional (nsub > 0): nbins / nsub + 1 entries
    uint32_t perm_base;      // added to every perm entry (slot == null): the sorted keys describe particles perm_base ... of an array
    uint8_t *slot;           // optional: array slot of every sorted particle (merged order)
    struct { uint32_t off[SPH_MAX_ARRAYS + 1]; int narrays; } co;
    uint32_t *bigq;          // [0] count, then bin ids
};

__device__ __forceinline__ void bin_emit(const BinFixArgs &a, uint32_t k, uint32_t j, uint32_t g)
{
    if (a.fkeys) a.fkeys[j] = k;
    if (a.keys) a.keys[j] = k / SPH_NSUB;
    if (a.slot) {
        uint32_t s = 0, base = 0;
#pragma unroll
        for (int b = 1; b < SPH_MAX_ARRAYS; b++)
            if (b < a.co.narrays && g >= a.co.off[b]) { s = (uint32_t)b; base = a.co.off[b]; }
        a.slot[j] = (uint8_t)s;
        a.perm[j] = g - base;
    } else {
        a.perm[j] = g + a.perm_base;
    }
}

__global__ __launch_bounds__(256) void k_bin_fix(BinFixArgs a)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > a.nbins) return;
    const uint32_t lo = a.table[k];
    if (a.nsub && a.cell_start && k % a.nsub == 0) a.cell_start[k / a.nsub] = lo;
    if (k == a.nbins) return;
    const uint32_t hi = a.table[k + 1], cnt = hi - lo;
    if (cnt == 0) return;
    if (cnt > BIN_SMALL) {
        const uint32_t q = atomicAdd(&a.bigq[0], 1u);
        if (q < BIN_QUEUE) { a.bigq[1 + q] = k; return; }
    }
    if (cnt <= 4) { // the common case in registers
        uint32_t v[4];
#pragma unroll
        for (int t = 0; t < 4; t++) v[t] = (uint32_t)t < cnt ? a.perm[lo + t] : 0xffffffffu;
#define CSWAP(i, j) { const uint32_t x = min(v[i], v[j]), y = max(v[i], v[j]); v[i] = x; v[j] = y; }
        CSWAP(0, 1) CSWAP(2, 3) CSWAP(0, 2) CSWAP(1, 3) CSWAP(1, 2)
#undef CSWAP
#pragma unroll
        for (int t = 0; t < 4; t++) if ((uint32_t)t < cnt) bin_emit(a, k, lo + t, v[t]);
        return;
    }
    for (uint32_t i = lo + 1; i < hi; i++) { // insertion sort in place
        const uint32_t x = a.perm[i];
        uint32_t j = i;
        while (j > lo && a.perm[j - 1] > x) { a.perm[j] = a.perm[j - 1]; j--; }
        a.perm[j] = x;
    }
    for (uint32_t i = lo; i < hi; i++) bin_emit(a, k, i, a.perm[i]);
}

// queued bins: rank by counting, one workgroup per bin; `scratch` holds a copy of the bin's entries (n entries available)
__global__ __launch_bounds__(256) void k_bin_fix_big(BinFixArgs a, uint32_t *__restrict__ scratch)
{
    const uint32_t nq = min(a.bigq[0], (uint32_t)BIN_QUEUE);
    for (uint32_t q = blockIdx.x; q < nq; q += gridDim.x) {
        const uint32_t k = a.bigq[1 + q];
        const uint32_t lo = a.table[k], hi = a.table[k + 1];
        for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) scratch[i] = a.perm[i];
        __syncthreads();
        for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
            const uint32_t x = scratch[i];
            uint32_t rank = 0;
            for (uint32_t j = lo; j < hi; j++) rank += scratch[j] < x;
            bin_emit(a, k, lo + rank, x);
        }
        __syncthreads();
    }
}

__global__ void k_reset_u32(uint32_t *p) { p[0] = 0u; }

// ghosts against what the update knew of the real particles: bit 0 = a mass other than mu, bit 1 = an h outside [hlo, hhi]
__global__ __launch_bounds__(256) void k_ghost_hm_check(const double *__restrict__ h, const double *__restrict__ m, size_t n, double hlo,
                                                        double hhi, double mu, uint32_t *__restrict__ flag)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t f = 0;
    const double hv = h[i];
    if (!(hv >= hlo && hv <= hhi)) f |= 2u;
    if (m && m[i] != mu) f |= 1u;
    if (f) atomicOr(flag, f);
}

// Sort n keys < nbins through `table` (nbins + 2 entries; the count pass -- k_tile_keys, k_cell_keys_count -- has run on the
// zeroed table): leaves table[k] = first sorted position of bin k (k <= nbins) and fills the BinFixArgs outputs.
static int nnps_bin_sort_finish(sph_ctx *c, const uint32_t *keys, size_t n, BinFixArgs fa, uint32_t *table)
{
    SPH_TRY(dev_scan_u32(c, table, table, (size_t)fa.nbins + 2, false));
    SPH_TRY(c->bigq.reserve((1 + BIN_QUEUE) * 4));
    SPH_TRY(c->tmp_u32b.reserve((n + 64) * 4));
    hipLaunchKernelGGL(k_reset_u32, dim3(1), dim3(1), 0, c->stream, c->bigq.as<uint32_t>());
    hipLaunchKernelGGL(k_bin_scatter, dim3(div_up(n, 256)), dim3(256), 0, c->stream, keys, n, table, fa.perm);
    fa.table = table;
    fa.bigq = c->bigq.as<uint32_t>();
    hipLaunchKernelGGL(k_bin_fix, dim3(div_up((size_t)fa.nbins + 1, 256)), dim3(256), 0, c->stream, fa);
    hipLaunchKernelGGL(k_bin_fix_big, dim3(256), dim3(256), 0, c->stream, fa, c->tmp_u32b.as<uint32_t>());
    return SPH_OK;
}

// fine keys of a (small) set of particles + the counting sort's count pass
__global__ __launch_bounds__(256) void k_cell_keys_count(const double *__restrict__ x, const double *__restrict__ y,
                                                         const double *__restrict__ z, size_t n, GridDesc g,
                                                         uint32_t *__restrict__ keys, uint32_t *__restrict__ count)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t key = fine_key_of(x[i], y[i], z[i], g, i);
    keys[i] = key;
    atomicAdd(&count[key + 2], 1u);
}


// ---------------------------------------------------------------------------
// merged-first build (sph_nnps_update) and the per-array tables derived from it on demand
// ---------------------------------------------------------------------------
// Stable compaction of the merged order by slot = every array's own cell order: count per block of SPLIT_BLOCK merged
// positions and slot, exclusive scan per slot over the blocks, scatter with the ranks re-derived from wavefront ballots.
#define SPLIT_BLOCK 1024
struct SplitOut { uint32_t *fkeys[SPH_MAX_ARRAYS], *keys[SPH_MAX_ARRAYS], *perm[SPH_MAX_ARRAYS]; int narrays; };

__global__ __launch_bounds__(256) void k_slot_count(const uint8_t *__restrict__ slot, size_t n, int narrays, uint32_t nb,
                                                    uint32_t *__restrict__ blockcnt)
{
    __shared__ uint32_t cnt[SPH_MAX_ARRAYS];
    if (threadIdx.x < SPH_MAX_ARRAYS) cnt[threadIdx.x] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    for (int r = 0; r < SPLIT_BLOCK / 256; r++) {
        const size_t p = (size_t)blockIdx.x * SPLIT_BLOCK + (size_t)r * 256 + threadIdx.x;
        const int s = p < n ? (int)slot[p] : -1;
        for (int b = 0; b < narrays; b++) {
            const unsigned long long m = __ballot(s == b);
            if (lane == 0 && m) atomicAdd(&cnt[b], (uint32_t)__builtin_popcountll(m));
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < narrays) blockcnt[(size_t)threadIdx.x * nb + blockIdx.x] = cnt[threadIdx.x];
}

// exclusive scan of row blockIdx.x (one slot) of blockcnt[narrays][nb], in place
__global__ __launch_bounds__(1024) void k_slot_scan(uint32_t *__restrict__ blockcnt, uint32_t nb)
{
    __shared__ uint32_t ws[17];
    uint32_t *row = blockcnt + (size_t)blockIdx.x * nb;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < nb; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < nb ? row[i] : 0u;
        uint32_t x = v;
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(x, o, 64); if (lane >= o) x += t; }
        if (lane == 63) ws[wave] = x;
        __syncthreads();
        if (wave == 0) {
            const uint32_t w = lane < 16 ? ws[lane] : 0u;
            uint32_t y = w;
            for (int o = 1; o < 16; o <<= 1) { const uint32_t t = __shfl_up(y, o, 64); if (lane >= o) y += t; }
            if (lane < 16) ws[lane] = y - w;
            if (lane == 15) ws[16] = y;
        }
        __syncthreads();
        if (i < nb) row[i] = carry + ws[wave] + x - v;
        carry += ws[16];
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_slot_scatter(const uint8_t *__restrict__ slot, const uint32_t *__restrict__ fkeys,
                                                      const uint32_t *__restrict__ perm, size_t n, uint32_t nb,
                                                      const uint32_t *__restrict__ blockoff, SplitOut o)
{
    __shared__ uint32_t run[SPH_MAX_ARRAYS], wcnt[4][SPH_MAX_ARRAYS];
    if ((int)threadIdx.x < o.narrays) run[threadIdx.x] = blockoff[(size_t)threadIdx.x * nb + blockIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;
    for (int r = 0; r < SPLIT_BLOCK / 256; r++) {
        const size_t p = (size_t)blockIdx.x * SPLIT_BLOCK + (size_t)r * 256 + threadIdx.x;
        const int s = p < n ? (int)slot[p] : -1;
        uint32_t rank = 0;
        for (int b = 0; b < o.narrays; b++) {
            const unsigned long long m = __ballot(s == b);
            if (lane == 0) wcnt[wave][b] = (uint32_t)__builtin_popcountll(m);
            if (s == b) rank = (uint32_t)__builtin_popcountll(m & lt);
        }
        __syncthreads();
        if (s >= 0) {
            uint32_t pos = run[s] + rank;
            for (int w = 0; w < wave; w++) pos += wcnt[w][s];
            const uint32_t f = fkeys[p], q = perm[p];
#pragma unroll
            for (int b = 0; b < SPH_MAX_ARRAYS; b++)
                if (s == b) { o.fkeys[b][pos] = f; o.keys[b][pos] = f / SPH_NSUB; o.perm[b][pos] = q; }
        }
        __syncthreads();
        if ((int)threadIdx.x < o.narrays) run[threadIdx.x] += wcnt[0][threadIdx.x] + wcnt[1][threadIdx.x] + wcnt[2][threadIdx.x] + wcnt[3][threadIdx.x];
        __syncthreads();
    }
}

static int nnps_tile_order(sph_ctx *c, DevArray &A, size_t n);

static int nnps_reserve_tables(sph_ctx *c, DevArray &A, size_t n)
{
    const size_t n_fine = (size_t)c->n_cells * SPH_NSUB;
    SPH_TRY(A.keys_sorted.reserve((n + 1) * 4));
    SPH_TRY(A.perm.reserve((n + 1) * 4));
    SPH_TRY(A.fkeys_sorted.reserve((n + 1) * 4));
    SPH_TRY(A.cell_start.reserve(((size_t)c->n_cells + 1) * 4));
    SPH_TRY(A.fine_start.reserve((n_fine + 1) * 4));
    return SPH_OK;
}

static void nnps_empty_tables(sph_ctx *c, DevArray &A)
{
    const size_t n_fine = (size_t)c->n_cells * SPH_NSUB;
    hipLaunchKernelGGL(k_fill_u32, dim3(div_up((size_t)c->n_cells + 1, 256)), dim3(256), 0, c->stream, A.cell_start.as<uint32_t>(),
                       (size_t)c->n_cells + 1, 0u);
    hipLaunchKernelGGL(k_fill_u32, dim3(div_up(n_fine + 1, 256)), dim3(256), 0, c->stream, A.fine_start.as<uint32_t>(), n_fine + 1, 0u);
}

// fine_start / cell_start (and, `with_keys`, the cell ids) of one array from its sorted fine keys; its tile order
// (the first n_binned particles: ghosts appended after the update live in tables of their own)
static int nnps_finish_tables(sph_ctx *c, DevArray &A, bool with_keys)
{
    const size_t n = A.n_binned, n_fine = (size_t)c->n_cells * SPH_NSUB;
    hipLaunchKernelGGL(k_cell_start, dim3(div_up(n + 1, 256)), dim3(256), 0, c->stream, A.fkeys_sorted.as<uint32_t>(), n,
                       (uint32_t)n_fine, A.fine_start.as<uint32_t>(), c->gapq.as<uint32_t>(),
                       with_keys ? A.keys_sorted.as<uint32_t>() : (uint32_t *)nullptr);
    hipLaunchKernelGGL(k_fill_gaps, dim3(512), dim3(256), 0, c->stream, c->gapq.as<uint32_t>(), A.fine_start.as<uint32_t>());
    hipLaunchKernelGGL(k_coarse_start, dim3(div_up((size_t)c->n_cells + 1, 256)), dim3(256), 0, c->stream,
                       A.fine_start.as<uint32_t>(), (uint32_t)c->n_cells, A.cell_start.as<uint32_t>(), c->gapq.as<uint32_t>());
    return nnps_tile_order(c, A, n);
}

// The per-array cell orders and tables of a merged-first update, built when first asked for (per-destination pair
// paths, neighbour-list queries, reorder); a no-op otherwise.
int nnps_need_tables(sph_ctx *c)
{
    if (!c->merged_valid || c->tables_valid) return SPH_OK;
    DevArray &M = c->merged;
    const int na = c->narrays;
    const uint32_t nb = (uint32_t)div_up(M.n, SPLIT_BLOCK);
    SPH_TRY(c->splitcnt.reserve((size_t)na * nb * 4 + 64));
    SplitOut so;
    memset(&so, 0, sizeof so);
    so.narrays = na;
    for (int a = 0; a < na; a++) {
        DevArray &A = c->arr[c->ids[a]];
        SPH_TRY(nnps_reserve_tables(c, A, A.n_binned));
        so.fkeys[a] = A.fkeys_sorted.as<uint32_t>(); so.keys[a] = A.keys_sorted.as<uint32_t>(); so.perm[a] = A.perm.as<uint32_t>();
    }
    if (M.n) {
        uint32_t *bc = c->splitcnt.as<uint32_t>();
        hipLaunchKernelGGL(k_slot_count, dim3(nb), dim3(256), 0, c->stream, M.slot8.as<uint8_t>(), M.n, na, nb, bc);
        hipLaunchKernelGGL(k_slot_scan, dim3(na), dim3(1024), 0, c->stream, bc, nb);
        hipLaunchKernelGGL(k_slot_scatter, dim3(nb), dim3(256), 0, c->stream, M.slot8.as<uint8_t>(), M.fkeys_sorted.as<uint32_t>(),
                           M.perm.as<uint32_t>(), M.n, nb, bc, so);
    }
    for (int a = 0; a < na; a++) {
        DevArray &A = c->arr[c->ids[a]];
        if (A.n_binned == 0) { nnps_empty_tables(c, A); continue; }
        SPH_TRY(nnps_finish_tables(c, A, false));
    }
    HIP_TRY(hipGetLastError());
    c->tables_valid = true;
    return SPH_OK;
}

// Traversal order of the 256-particle destination tiles of one (cell-sorted) array: only worth it when there is more than
// one z plane of tiles.  The order is a permutation of the tile ids that only steers locality: any permutation of the same
// nt tiles gives the same results.  Particles move a fraction of a cell per step, so the order of the last build stays
// good: rebuilt when the tile count or the grid changed and every 16th update.
// (one traversal order: of the position tiles, or -- `compact` -- of the tiles of the real-particle list DevArray::dlist)
struct TileOrd { DevBuf *key, *id, *order; size_t *n_tiles; int *grid, *age; };
static int nnps_tile_order_of(sph_ctx *c, DevArray &A, size_t n, TileOrd T, const uint32_t *dlist)
{
    const bool want_tiles = c->tile_block_rows > 0 && c->nc[2] > 1 && n > 64 * SPH_TILE;
    const int tsig = (int)c->tile_block_rows;
    const uint32_t nt_now = (uint32_t)div_up(n, SPH_TILE);
    if (want_tiles && *T.n_tiles == nt_now && T.grid[0] == c->nc[0] && T.grid[1] == c->nc[1] &&
        T.grid[2] == c->nc[2] && T.grid[3] == tsig && ++*T.age < 16)
        return SPH_OK;
    *T.n_tiles = 0;
    if (want_tiles) {
        *T.age = 0;
        T.grid[0] = c->nc[0]; T.grid[1] = c->nc[1]; T.grid[2] = c->nc[2]; T.grid[3] = tsig;
        const uint32_t nt = nt_now;
        SPH_TRY(T.key->reserve((size_t)nt * 4 * 2));
        SPH_TRY(T.id->reserve((size_t)nt * 4));
        SPH_TRY(T.order->reserve((size_t)nt * 4));
        // keys = traversal rank of the tile's row (< ncy * ncz rounded up to whole blocks of rows): the same counting sort
        const uint32_t by = (uint32_t)c->tile_block_rows;
        const uint32_t nrow_keys = ((uint32_t)c->nc[1] + by - 1) / by * by * (uint32_t)c->nc[2];
        SPH_TRY(T.id->reserve(((size_t)nrow_keys + 2) * 4));
        uint32_t *tk = T.key->as<uint32_t>(), *tt = T.id->as<uint32_t>();
        HIP_TRY(hipMemsetAsync(tt, 0, ((size_t)nrow_keys + 2) * 4, c->stream));
        hipLaunchKernelGGL(k_tile_keys, dim3(div_up(nt, 256)), dim3(256), 0, c->stream, A.keys_sorted.as<uint32_t>(), n, nt,
                           c->nc[0], c->nc[1], c->nc[2], (int)c->tile_block_rows, tk, tt, dlist);
        BinFixArgs fa;
        memset(&fa, 0, sizeof fa);
        fa.nbins = nrow_keys; fa.nsub = 0;
        fa.perm = T.order->as<uint32_t>();
        SPH_TRY(nnps_bin_sort_finish(c, tk, nt, fa, tt));
        *T.n_tiles = nt;
    }
    return SPH_OK;
}
static int nnps_tile_order(sph_ctx *c, DevArray &A, size_t n)
{
    return nnps_tile_order_of(c, A, n, TileOrd{&A.tile_key, &A.tile_id, &A.tile_order, &A.n_tiles, A.tile_grid, &A.tile_age}, nullptr);
}

// ---------------------------------------------------------------------------
// The real particles of a cell order (DevArray::dlist).  Position p of the order holds original index perm[p] of the
// array in slot slot[p] (one array: slot 0); it is REAL when that index lies below the array's n_real -- ghosts, periodic
// images and parked padding rows all lie behind.  Two launches in the pattern of sph_halo_select_pack: one counter per
// chunk, then every workgroup sums the counters before its own and ranks its positions with wavefront ballots.
// ---------------------------------------------------------------------------
struct RealTest {
    const uint32_t *perm;
    const uint8_t *slot; // null: one array
    uint32_t nreal[SPH_MAX_ARRAYS];
    int narrays;
};
__device__ __forceinline__ bool real_at(const RealTest &r, size_t p)
{
    const uint32_t s = r.slot ? r.slot[p] : 0u;
    uint32_t nr = r.nreal[0];
#pragma unroll
    for (int k = 1; k < SPH_MAX_ARRAYS; k++) nr = (k < r.narrays && s == (uint32_t)k) ? r.nreal[k] : nr;
    return r.perm[p] < nr;
}
#define DLIST_MAX_CHUNKS 2048
__global__ __launch_bounds__(256) void k_dlist_counts(RealTest r, size_t n, int q256, uint32_t *__restrict__ blk)
{
    const size_t first = (size_t)blockIdx.x * 256 * q256;
    uint32_t cnt = 0; // (wave-uniform)
    for (int q = 0; q < q256; q++) {
        const size_t p = first + (size_t)q * 256 + threadIdx.x;
        cnt += (uint32_t)__popcll(__ballot(p < n && real_at(r, p)));
    }
    __shared__ uint32_t sc[4];
    if ((threadIdx.x & 63) == 0) sc[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) blk[blockIdx.x] = sc[0] + sc[1] + sc[2] + sc[3];
}
__global__ __launch_bounds__(256) void k_dlist_fill(RealTest r, size_t n, int q256, const uint32_t *__restrict__ blk,
                                                    uint32_t *__restrict__ dlist)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __shared__ uint32_t sbase[4], wcnt[4];
    {
        uint32_t b0 = 0;
        for (uint32_t b = threadIdx.x; b < blockIdx.x; b += 256) b0 += blk[b];
        for (int o = 32; o > 0; o >>= 1) b0 += __shfl_xor(b0, o, 64);
        if (lane == 0) sbase[wv] = b0;
    }
    __syncthreads();
    size_t run = (size_t)sbase[0] + sbase[1] + sbase[2] + sbase[3];
    const size_t first = (size_t)blockIdx.x * 256 * q256;
    for (int q = 0; q < q256; q++) {
        const size_t p = first + (size_t)q * 256 + threadIdx.x;
        const bool on = p < n && real_at(r, p);
        const unsigned long long m = __ballot(on);
        __syncthreads(); // (the previous trip's wcnt has been read)
        if (lane == 0) wcnt[wv] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t before = 0;
        for (int w = 0; w < wv; w++) before += wcnt[w];
        if (on) dlist[run + before + (size_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)p;
        run += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    }
}

// dlist of the order T (the merged one over `narrays` arrays, or one array's own): built when some array holds rows behind
// its real particles, dropped otherwise
static int nnps_dest_list(sph_ctx *c, DevArray &T, bool merged, int narrays, const int *ids, size_t n_cat)
{
    T.dlist_n = 0; // (the tile order of the list keeps its age: rebuilt when the list's length or the grid changes, and every 16th update)
    if (!c->dest_list || n_cat == 0) return SPH_OK;
    RealTest r;
    memset(&r, 0, sizeof r);
    size_t n_real = 0;
    bool ghosts = false;
    for (int a = 0; a < narrays; a++) {
        const DevArray &A = c->arr[ids[a]];
        if (!merged && A.n == 0) continue;
        r.nreal[merged ? a : 0] = (uint32_t)A.n_real;
        n_real += A.n_real;
        ghosts |= A.n > A.n_real;
    }
    if (!ghosts || n_real == 0) return SPH_OK;
    // Worth its two passes (~22 us at 5 M positions) where the rows behind the real particles are a large share of the
    // order: Taylor-Green's periodic images (20 %: force pass 3.32 -> 3.07 ms).  A slab rank of the 16 M dam break (9.6 %
    // ghosts) gains 0.02 ms on its pair launch -- a wave tile with idle ghost lanes ends early -- and pays 0.02 for the
    // list: option dest_list 1 (default) builds it from one eighth of the positions on, 2 always, 0 never.
    if (c->dest_list == 1 && (n_cat - n_real) * 8 < n_cat) return SPH_OK;
    r.narrays = merged ? narrays : 1;
    r.perm = T.perm.as<uint32_t>();
    r.slot = merged ? T.slot8.as<uint8_t>() : nullptr;
    const int q256 = (int)div_up(n_cat, (size_t)256 * DLIST_MAX_CHUNKS);
    const unsigned nb = div_up(n_cat, (size_t)256 * q256);
    SPH_TRY(T.dl_cnt.reserve(((size_t)nb + 1) * 4));
    SPH_TRY(T.dlist.reserve((n_real + 64) * 4));
    hipLaunchKernelGGL(k_dlist_counts, dim3(nb), dim3(256), 0, c->stream, r, n_cat, q256, T.dl_cnt.as<uint32_t>());
    hipLaunchKernelGGL(k_dlist_fill, dim3(nb), dim3(256), 0, c->stream, r, n_cat, q256, T.dl_cnt.as<uint32_t>(), T.dlist.as<uint32_t>());
    T.dlist_n = n_real;
    return nnps_tile_order_of(c, T, n_real, TileOrd{&T.ctile_key, &T.ctile_id, &T.ctile_order, &T.n_ctiles, T.ctile_grid, &T.ctile_age},
                              T.dlist.as<uint32_t>());
}

// fine x index (cell * SPH_NSUB + sub-bin along a row) beyond which ghosts lie, from the slab faces the host named
// (sph_nnps_set_ghost_faces) and the grid of the current update
static void nnps_face_planes(sph_ctx *c)
{
    c->gfx_lo = -0x7fffffff; c->gfx_hi = 0x7fffffff; // no faces named: no wavefront is a face wavefront
    if (c->face_axis < 0) return;
    if (c->face_axis != 0) { c->gfx_lo = 0x7fffffff; c->gfx_hi = -0x7fffffff; return; } // rows run along x: every wavefront may see ghosts
    // the fine x index of the face planes by the keys' own (monotone) map: a ghost at x < face_lo has an index <= the
    // plane's, one at x >= face_hi an index >= the plane's
    GridDesc g;
    grid_set_cell(g, c->cell_size);
    const double flo = floor((c->face_lo - c->xmin[0]) * g.inv_cell * SPH_NSUB), fhi = floor((c->face_hi - c->xmin[0]) * g.inv_cell * SPH_NSUB);
    c->gfx_lo = !(flo > -1e9) ? -0x7fffffff : (flo > 1e9 ? 0x7fffffff : (int)flo); // ghosts: fx <= gfx_lo ...
    c->gfx_hi = !(fhi < 1e9) ? 0x7fffffff : (fhi < -1e9 ? -0x7fffffff : (int)fhi);  // ... or fx >= gfx_hi
}

// the merged records met a density that is not positive (sph_ctx::xflag): loud, once; the merged path stays off
static int nnps_rho_flag(sph_ctx *c)
{
    if (c->merge_blocked) return SPH_OK;
    c->merge_blocked = true;
    sph_set_error("a density <= 0 (or NaN) reached the one-launch WCSPH path, whose records carry the particle's class in the "
                  "sign of rho: the evaluations since then ran with wrong classes.  The context takes the per-destination "
                  "path from now on (option merge_arrays 0 from the start avoids the encoding)");
    return SPH_ERR_STATE;
}

// the bounds the last update sent to pin_async have arrived
static int lag_wait(sph_ctx *c)
{
    if (c->lag.pending) {
        HIP_TRY(hipEventSynchronize(c->lag_ev));
        c->lag.pending = false;
        c->sort_groups = c->pin_async[MM_OUT_GROUPS];
        if (c->pin_async[MM_OUT_XFLAG] != 0.0) SPH_TRY(nnps_rho_flag(c));
    }
    return SPH_OK;
}

// the grid the reference would report for the last update (exact; the binning grid of an update without a round trip
// is the one of the update before)
static int nnps_reported_grid(sph_ctx *c)
{
    if (c->rep_valid) return SPH_OK;
    SPH_TRY(lag_wait(c));
    double mm[8];
    memcpy(mm, c->pin_async, sizeof mm);
    mm[3] = c->lag.hr[0]; mm[7] = c->lag.hr[1];
    SPH_TRY(grid_from_minmax(c, c->lag.dim, c->lag.radius_scale, c->lag.cell_size_in, nullptr, mm, &c->rep));
    c->rep_valid = true;
    return SPH_OK;
}

extern "C" int sph_nnps_update(sph_ctx *c, int dim, int narrays, const int *ids, double radius_scale,
                               double cell_size_in, const double *bounds)
{
    if (!c || narrays < 1 || narrays > SPH_MAX_ARRAYS || dim < 1 || dim > 3) {
        sph_set_error("sph_nnps_update: bad arguments");
        return SPH_ERR_ARG;
    }
    HIP_TRY(hipSetDevice(c->device));
    ScopedTimer tm(c, T_NNPS);
    c->nnps_valid = false;
    for (int a = 0; a < narrays; a++) {
        if (ids[a] < 0 || ids[a] >= SPH_MAX_ARRAYS || !c->arr[ids[a]].used) {
            sph_set_error("sph_nnps_update: array id %d not registered", ids[a]);
            return SPH_ERR_ARG;
        }
    }
    if (!c->pin_async) {
        HIP_TRY(hipHostMalloc((void **)&c->pin_async, 64 * sizeof(double), hipHostMallocDefault));
        HIP_TRY(hipEventCreateWithFlags(&c->lag_ev, hipEventDisableTiming));
    }
    BinArrays ba;
    size_t n_cat = 0;
    uint32_t nblocks = 0;
    SPH_TRY(bin_arrays(c, narrays, ids, &ba, &n_cat, &nblocks));

    // What is known without looking (DevArray::h_dirty / m_dirty).  h: the range the caller gave (sph_nnps_set_h_range), or
    // the union of the arrays' clean ranges; m: only looked at once an evaluation could have used one mass per array.
    const bool h_given = c->h_known[1] >= 0.0;
    bool h_clean = true, m_clean = true;
    double hr[2] = {DBL_MAX, -DBL_MAX};
    for (int a = 0; a < narrays; a++) {
        DevArray &A = c->arr[ids[a]];
        if (A.n == 0) continue;
        if (A.raw_hm || A.h_dirty || !A.h_seen) h_clean = false;
        else { hr[0] = fmin(hr[0], A.h_lo); hr[1] = fmax(hr[1], A.h_hi); }
        if (c->want_mrange && A.prop[SPH_M] && (A.raw_hm || A.m_dirty || !A.m_seen)) m_clean = false;
    }
    if (h_given) { hr[0] = c->h_known[0]; hr[1] = c->h_known[1]; }
    const bool have_h = h_given || h_clean;
    const bool lag_match = c->lag.valid && c->lag.dim == dim && c->lag.narrays == narrays && c->lag.radius_scale == radius_scale &&
                           c->lag.cell_size_in == cell_size_in && memcmp(c->lag.ids, ids, narrays * sizeof(int)) == 0 &&
                           memcmp(c->lag.extend, c->extend, sizeof c->extend) == 0;
    // the three ways through: no look at all (bounds given), bounds of the previous update (no round trip), a look now
    const bool no_look = bounds && have_h && m_clean;
    const bool lagged = !no_look && !bounds && c->async_update && have_h && m_clean && lag_match && n_cat > 0;

    double mm[8] = {DBL_MAX, DBL_MAX, DBL_MAX, DBL_MAX, -DBL_MAX, -DBL_MAX, -DBL_MAX, -DBL_MAX};
    int mm_async = 0; // what k_bin_keys reduces next to computing the keys
    if (no_look) {
        for (int k = 0; k < 3; k++) { mm[k] = bounds[k]; mm[4 + k] = bounds[3 + k]; }
        c->lag.valid = false;
    } else if (lagged) {
        SPH_TRY(lag_wait(c));
        memcpy(mm, c->pin_async, sizeof mm);
        mm_async = MM_XYZ;
        c->n_async_updates++;
        c->timers[T_N_ASYNC].count++;
    } else {
        const int look = (bounds ? 0 : MM_XYZ) | (have_h ? 0 : MM_H) | (m_clean ? 0 : MM_M);
        SPH_TRY(nnps_minmax(c, narrays, ids, mm, look));
        if (!(look & MM_H)) { mm[3] = hr[0]; mm[7] = hr[1]; }
        else if (h_given) { /* never: have_h */ }
        if (bounds) for (int k = 0; k < 3; k++) { mm[k] = bounds[k]; mm[4 + k] = bounds[3 + k]; }
        if (!h_given && (look & MM_H)) {
            hr[0] = DBL_MAX; hr[1] = -DBL_MAX;
            for (int a = 0; a < narrays; a++) {
                DevArray &A = c->arr[ids[a]];
                if (A.n == 0) continue;
                hr[0] = fmin(hr[0], c->pinned[10 + 4 * a]); hr[1] = fmax(hr[1], c->pinned[11 + 4 * a]);
            }
        }
        // these bounds are what the next update may bin on
        memcpy(c->pin_async, mm, sizeof mm);
        c->pin_async[MM_OUT_GROUPS] = c->pinned[MM_OUT_GROUPS];
        c->sort_groups = c->pinned[MM_OUT_GROUPS];
        if (nblocks && c->pinned[MM_OUT_XFLAG] != 0.0) SPH_TRY(nnps_rho_flag(c));
        c->lag.valid = !bounds;
        c->lag.pending = false;
    }
    mm[3] = hr[0]; mm[7] = hr[1];
    // masses: an update that did not look keeps what the last look found while nothing wrote m (m_dirty)
    for (int a = 0; a < narrays; a++) {
        DevArray &A = c->arr[ids[a]];
        if (!c->want_mrange || !A.prop[SPH_M] || A.n == 0) { A.m_known = false; continue; }
        if (!A.raw_hm && !A.m_dirty && A.m_seen) { A.m_known = A.m_lo == A.m_hi && !A.m_mixed_ghosts; A.m_value = A.m_lo; }
    }
    c->lag.dim = dim; c->lag.narrays = narrays; c->lag.radius_scale = radius_scale; c->lag.cell_size_in = cell_size_in;
    memcpy(c->lag.ids, ids, narrays * sizeof(int));
    memcpy(c->lag.extend, c->extend, sizeof c->extend);
    c->lag.hr[0] = hr[0]; c->lag.hr[1] = hr[1];

    GridHost G;
    SPH_TRY(grid_from_minmax(c, dim, radius_scale, cell_size_in, bounds, mm, &G));
    c->rep_valid = !lagged;
    if (!lagged) c->rep = G;
    c->hmin = G.hmin;
    c->uniform_h = G.uniform_h;
    c->h_uniform = G.h_uniform;
    const double cell_size = G.cell_size;
    const long n_cells_alloc = G.n_cells_alloc;

    c->dim = dim;
    c->narrays = narrays;
    c->radius_scale = radius_scale;
    c->cell_size = cell_size;
    for (int k = 0; k < 3; k++) { c->xmin[k] = G.xmin[k]; c->xmax[k] = G.xmax[k]; c->nc[k] = G.nc[k]; }
    c->n_cells = n_cells_alloc;
    for (auto &A : c->arr) A.nnps_slot = -1;

    // parked padding rows (an array that took a padded ghost message or padded images since its ghosts were last dropped)
    // are binned into PARKING cells behind the grid's own: the tables are sized for both, nothing else knows
    bool padding = false;
    for (int a = 0; a < narrays; a++) padding |= c->arr[ids[a]].has_padding && c->arr[ids[a]].n > 0;
    const long park_cells = padding ? (long)std::min<size_t>(std::max<size_t>(n_cat / 64, 512), 1u << 18) : 0;
    const long n_cells_tab = n_cells_alloc + park_cells;
    c->park_cells = park_cells;
    c->n_cells = n_cells_tab;
    GridDesc g;
    for (int k = 0; k < 3; k++) { g.xmin[k] = c->xmin[k]; g.nc[k] = c->nc[k]; }
    grid_set_cell(g, cell_size);
    g.park_base = (uint32_t)(n_cells_alloc * SPH_NSUB); g.park_bins = (uint32_t)(park_cells * SPH_NSUB); g.park_n = (uint32_t)n_cat;
    const size_t n_fine = (size_t)n_cells_tab * SPH_NSUB;

    // Several arrays (a dam break has three): ONE sort of all their keys.
    //  * merged-first (option merge_arrays, default): the arrays are concatenated in slot order and the sort is stable, so
    //    equal fine keys keep slot order: the sorted sequence IS the merged order of all arrays (sph_ctx::merged) the
    //    multi-array pair kernel runs on; the sorted value (position in the concatenation) gives slot and original index.
    //    Per-array tables are derived from it only when something asks for them (nnps_need_tables: per-destination pair
    //    paths, neighbour-list queries, reorder) -- a steady-state dam-break step builds ONE fine_start table.
    //  * otherwise (option merge_arrays 0, or one array): every array sorted into its own tables.
    int n_nonempty = 0;
    for (int a = 0; a < narrays; a++) n_nonempty += c->arr[ids[a]].n > 0;
    const bool merged_first = n_nonempty > 1 && c->merge_arrays;
    if (!c->gapq.ptr) { // first use: the queue starts empty; afterwards k_coarse_start leaves it empty
        SPH_TRY(c->gapq.reserve((4 + 3 * GAP_QUEUE) * 4));
        HIP_TRY(hipMemsetAsync(c->gapq.ptr, 0, 16, c->stream));
    }
    c->merged_valid = false;
    c->tables_valid = true;
    c->ghosts_binned = false;
    c->merged.dlist_n = 0;
    for (int a = 0; a < narrays; a++) c->arr[ids[a]].dlist_n = 0;
    nnps_face_planes(c); // ghost split: where this step's ghosts will lie (known before they arrive)
    for (int a = 0; a < narrays; a++) {
        c->ids[a] = ids[a];
        c->arr[ids[a]].nnps_slot = a;
        c->arr[ids[a]].perm_n = c->arr[ids[a]].n;
        c->arr[ids[a]].n_binned = c->arr[ids[a]].n;
        c->arr[ids[a]].g_n = 0;
        DevArray &A = c->arr[ids[a]];
        A.m_known_binned = A.m_known; A.hm_writes_binned = A.hm_writes;
        A.h_clean_binned = !A.raw_hm && !A.h_dirty && A.h_seen; A.m_clean_binned = !A.raw_hm && !A.m_dirty && A.m_seen;
    }
    SPH_TRY(c->tmp_u32a.reserve((n_cat + 64) * 8));
    SPH_TRY(c->tmp_u32b.reserve((n_cat + 64) * 8));
    const int lbits = sort_choose_lbits(c, n_cat, n_fine);
    const uint32_t nbuckets = (uint32_t)(n_fine >> lbits) + 1u;
    bool reduced = false; // mm_async has ridden on a launch
    if (merged_first || n_nonempty == 1) {
        DevArray *T = &c->merged;
        if (!merged_first)
            for (int a = 0; a < narrays; a++) if (c->arr[ids[a]].n) T = &c->arr[ids[a]];
        if (merged_first) { T->n = T->n_real = n_cat; SPH_TRY(T->slot8.reserve(n_cat + 64)); }
        SPH_TRY(nnps_reserve_tables(c, *T, n_cat));
        // An array that lies in memory in NO spatial order (the previous key pass, run in memory order, formed more than
        // n / 8 atomic groups: every lane its own bucket) is visited in the cell order of the previous update instead:
        // a wavefront's 64 particles are neighbours again -- gathers for the positions instead of 4 M atomics per pass.
        // Sticky until the array is reordered or resized (a pass run in cell order says nothing about the memory order).
        const uint32_t *via = nullptr;
        if (!merged_first) {
            if (!c->last_keys_via && c->last_keys_n > 0 && c->last_keys_array == T && c->sort_groups > (double)c->last_keys_n / 8.0) T->unordered = true;
            if (c->via_unordered && T->unordered && T->perm_direct_n == T->n) via = T->perm.as<uint32_t>();
            for (int a = 0; a < narrays; a++) if (&c->arr[ids[a]] == T) ba.via[a] = via;
        }
        SPH_TRY(launch_bin_keys(c, ba, nblocks, g, mm_async, c->tmp_u32a.as<uint32_t>(), nbuckets, lbits));
        c->last_keys_via = via != nullptr; c->last_keys_n = merged_first ? 0 : n_cat; c->last_keys_array = merged_first ? nullptr : T;
        reduced = true;
        SortDest d;
        d.T = T; d.merged = merged_first; d.via = via;
        d.co.narrays = narrays;
        for (int a = 0; a <= SPH_MAX_ARRAYS; a++) d.co.off[a] = a < narrays ? ba.off[a] : (uint32_t)n_cat;
        SPH_TRY(sort_finish(c, n_cat, n_fine, n_cells_tab, lbits, nbuckets, d));
        SPH_TRY(nnps_tile_order(c, *T, n_cat));
        SPH_TRY(nnps_dest_list(c, *T, merged_first, narrays, ids, n_cat));
        if (!merged_first) T->perm_direct_n = n_cat;
        if (merged_first) {
            c->merged_valid = true;
            c->tables_valid = false;
            if (!c->lazy_tables) SPH_TRY(nnps_need_tables(c));
        }
        for (int a = 0; a < narrays; a++) { // the empty arrays next to the one that was sorted
            DevArray &A = c->arr[ids[a]];
            if (A.n || merged_first) continue;
            SPH_TRY(nnps_reserve_tables(c, A, 0));
            nnps_empty_tables(c, A);
        }
    } else {
        for (int a = 0; a < narrays; a++) {
            DevArray &A = c->arr[ids[a]];
            SPH_TRY(nnps_reserve_tables(c, A, A.n));
            if (A.n == 0) { nnps_empty_tables(c, A); continue; }
            BinArrays b1;
            size_t n1 = 0;
            uint32_t nb1 = 0;
            SPH_TRY(bin_arrays(c, 1, &ids[a], &b1, &n1, &nb1));
            SPH_TRY(launch_bin_keys(c, b1, nb1, g, 0, c->tmp_u32a.as<uint32_t>(), nbuckets, lbits));
            SortDest d;
            d.T = &A; d.merged = false;
            d.co.narrays = 1;
            for (int k = 0; k <= SPH_MAX_ARRAYS; k++) d.co.off[k] = k ? (uint32_t)n1 : 0u;
            SPH_TRY(sort_finish(c, n1, n_fine, n_cells_tab, lbits, nbuckets, d));
            SPH_TRY(nnps_tile_order(c, A, n1));
            SPH_TRY(nnps_dest_list(c, A, false, 1, &ids[a], n1));
            A.perm_direct_n = n1;
            c->last_keys_n = 0; // (several arrays: no single pass to judge)
        }
    }
    if (lagged) {
        if (!reduced && nblocks) SPH_TRY(launch_bin_keys(c, ba, nblocks, g, mm_async, nullptr, 0, lbits));
        // this update's bounds: on their way, nobody waits
        HIP_TRY(hipMemcpyAsync(c->pin_async, c->red_out.ptr, MM_OUT_N * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipEventRecord(c->lag_ev, c->stream));
        c->lag.pending = true;
    }
    HIP_TRY(hipGetLastError());
    c->nnps_valid = true;
    c->nnps_epoch++;
    return SPH_OK;
}

extern "C" int sph_nnps_set_h_range(sph_ctx *c, double hmin, double hmax)
{
    if (!c || (hmax >= 0.0 && hmin > hmax)) { sph_set_error("sph_nnps_set_h_range: bad arguments"); return SPH_ERR_ARG; }
    c->h_known[0] = hmin; c->h_known[1] = hmax;
    return SPH_OK;
}

extern "C" int sph_nnps_set_extend(sph_ctx *c, double ex, double ey, double ez)
{
    if (!c || ex < 0.0 || ey < 0.0 || ez < 0.0) { sph_set_error("sph_nnps_set_extend: bad arguments"); return SPH_ERR_ARG; }
    c->extend[0] = ex; c->extend[1] = ey; c->extend[2] = ez;
    return SPH_OK;
}

extern "C" int sph_nnps_set_ghost_faces(sph_ctx *c, int axis, double lo, double hi)
{
    if (!c || axis < -1 || axis > 2) { sph_set_error("sph_nnps_set_ghost_faces: bad arguments"); return SPH_ERR_ARG; }
    c->face_axis = axis; c->face_lo = lo; c->face_hi = hi;
    if (c->n_cells > 0) nnps_face_planes(c);
    return SPH_OK;
}

// The particles that arrived behind the ones the last sph_nnps_update binned (the ghosts of this step), on the same
// grid, into tables of their own: the hand-written counting sort (few keys; rocPRIM would take its ~25-launch merge sort).
extern "C" int sph_nnps_update_ghosts(sph_ctx *c, int axis, double lo, double hi)
{
    if (!c || axis < 0 || axis > 2) { sph_set_error("sph_nnps_update_ghosts: bad arguments"); return SPH_ERR_ARG; }
    if (c->n_cells <= 0 || c->narrays < 1) { sph_set_error("sph_nnps_update_ghosts: call sph_nnps_update first"); return SPH_ERR_STATE; }
    HIP_TRY(hipSetDevice(c->device));
    ScopedTimer tm(c, T_NNPS);
    // the grid must still describe the real particles: nothing but appends since the update
    for (int a = 0; a < c->narrays; a++) {
        DevArray &A = c->arr[c->ids[a]];
        if (A.nnps_slot != a || A.n < A.n_binned || A.n_real > A.n_binned) {
            sph_set_error("sph_nnps_update_ghosts: array %d changed other than by appended ghosts since sph_nnps_update", c->ids[a]);
            return SPH_ERR_STATE;
        }
    }
    GridDesc g;
    for (int k = 0; k < 3; k++) { g.xmin[k] = c->xmin[k]; g.nc[k] = c->nc[k]; }
    grid_set_cell(g, c->cell_size);
    g.park_base = (uint32_t)((c->n_cells - c->park_cells) * SPH_NSUB); g.park_bins = (uint32_t)(c->park_cells * SPH_NSUB);
    g.park_n = 1; // (set per array below: the ghost segment's own row count)
    const size_t n_fine = (size_t)c->n_cells * SPH_NSUB;
    for (int a = 0; a < c->narrays; a++) {
        DevArray &A = c->arr[c->ids[a]];
        const size_t ng = A.n - A.n_binned;
        A.g_n = ng;
        if (ng == 0) continue;
        SPH_TRY(A.g_keys.reserve((ng + 1) * 4));
        SPH_TRY(A.g_fkeys.reserve((ng + 1) * 4));
        SPH_TRY(A.g_perm.reserve((ng + 1) * 4));
        SPH_TRY(A.g_fine_start.reserve((n_fine + 2) * 4));
        SPH_TRY(A.g_cell_start.reserve(((size_t)c->n_cells + 1) * 4));
        uint32_t *T = A.g_fine_start.as<uint32_t>();
        HIP_TRY(hipMemsetAsync(T, 0, (n_fine + 2) * 4, c->stream));
        const size_t nb = A.n_binned;
        g.park_n = (uint32_t)ng;
        hipLaunchKernelGGL(k_cell_keys_count, dim3(div_up(ng, 256)), dim3(256), 0, c->stream, A.prop[SPH_X] + nb, A.prop[SPH_Y] + nb,
                           A.prop[SPH_Z] + nb, ng, g, A.g_keys.as<uint32_t>(), T);
        BinFixArgs fa;
        memset(&fa, 0, sizeof fa);
        fa.nbins = (uint32_t)n_fine; fa.nsub = SPH_NSUB;
        fa.perm = A.g_perm.as<uint32_t>(); fa.fkeys = A.g_fkeys.as<uint32_t>();
        fa.cell_start = A.g_cell_start.as<uint32_t>();
        fa.perm_base = (uint32_t)nb; // original index of ghost k is n_binned + k
        SPH_TRY(nnps_bin_sort_finish(c, A.g_keys.as<uint32_t>(), ng, fa, T));
    }
    // Everything the update decided from the REAL particles -- the cell size and the uniform-h path from their h range, the
    // one-mass-per-array records from their masses -- must hold for the ghosts too: they are other ranks' particles.
    // Checked every time, on the device, one word per array back: a ghost with another mass switches the array to
    // mass-carrying records (the second half of a split evaluation repacks the real particles' records then); a ghost
    // whose h lies outside the range the grid was built for is an error (fixed_h with the global range avoids it).
    {
        SPH_TRY(c->bigq.reserve((1 + BIN_QUEUE) * 4));
        uint32_t *flags = c->bigq.as<uint32_t>(); // (the bin-sort queue: idle between the sorts)
        hipLaunchKernelGGL(k_fill_u32, dim3(1), dim3(256), 0, c->stream, flags, (size_t)SPH_MAX_ARRAYS, 0u);
        bool any = false;
        for (int a = 0; a < c->narrays; a++) {
            DevArray &A = c->arr[c->ids[a]];
            if (A.g_n == 0) continue;
            any = true;
            const size_t nb = A.n_binned;
            hipLaunchKernelGGL(k_ghost_hm_check, dim3(div_up(A.g_n, 256)), dim3(256), 0, c->stream, A.prop[SPH_H] + nb,
                               A.m_known_binned && A.prop[SPH_M] ? A.prop[SPH_M] + nb : (const double *)nullptr, A.g_n,
                               c->lag.hr[0], c->lag.hr[1], A.m_value, flags + a);
        }
        if (any) {
            uint32_t *pin = (uint32_t *)c->pinned;
            HIP_TRY(hipMemcpyAsync(pin, flags, SPH_MAX_ARRAYS * 4, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            for (int a = 0; a < c->narrays; a++) {
                DevArray &A = c->arr[c->ids[a]];
                if (A.g_n == 0) continue;
                if (pin[a] & 2u) {
                    sph_set_error("sph_nnps_update_ghosts: a ghost of array %d has a smoothing length outside [%g, %g], the range the "
                                  "grid of this update was built for (fixed_h with the global h range covers other ranks' particles)",
                                  c->ids[a], c->lag.hr[0], c->lag.hr[1]);
                    return SPH_ERR_STATE;
                }
                const bool untouched = A.hm_writes == A.hm_writes_binned; // nothing but the append since the update
                if (A.m_known_binned) {
                    if (pin[a] & 1u) { A.m_known = false; A.m_mixed_ghosts = true; }
                    else { A.m_known = true; if (untouched && A.m_clean_binned) A.m_dirty = false; }
                }
                if (untouched && A.h_clean_binned) {
                    // the ghosts were checked against the range the GRID was built for (the union over the arrays, or the
                    // caller's): this array's own range is known again, widened to that one (a superset: an array whose
                    // ONE smoothing length differs from the others' is no longer "one value" once foreign ghosts joined)
                    A.h_dirty = false;
                    A.h_lo = fmin(A.h_lo, c->lag.hr[0]); A.h_hi = fmax(A.h_hi, c->lag.hr[1]);
                }
            }
        }
    }
    c->face_axis = axis; c->face_lo = lo; c->face_hi = hi;
    nnps_face_planes(c);
    HIP_TRY(hipGetLastError());
    c->ghosts_binned = true;
    c->nnps_valid = true; // the appends invalidated it; reals and ghosts are both binned again
    return SPH_OK;
}

extern "C" int sph_nnps_info(sph_ctx *c, double *d8, long *i4)
{
    if (c->nnps_epoch == 0) { sph_set_error("sph_nnps_info: call sph_nnps_update first"); return SPH_ERR_STATE; }
    // the reference's values for the particles of the last update -- of an update without a round trip they are computed
    // now, from the bounds it sent on their way (the grid it BINNED on is the one of the update before)
    SPH_TRY(nnps_reported_grid(c));
    const GridHost &G = c->rep;
    d8[0] = G.cell_size; d8[1] = G.hmin;
    for (int k = 0; k < 3; k++) { d8[2 + k] = G.xmin[k]; d8[5 + k] = G.xmax[k]; }
    i4[0] = G.nc[0]; i4[1] = G.nc[1]; i4[2] = G.nc[2];
    // n_cells as the reference reports it (dim-aware, linked_list_nnps.pyx:321-325)
    i4[3] = G.n_cells;
    return SPH_OK;
}

extern "C" int sph_nnps_get_order(sph_ctx *c, int id, uint32_t *perm)
{
    if (!c->nnps_valid || c->arr[id].nnps_slot < 0) { sph_set_error("sph_nnps_get_order: array not binned"); return SPH_ERR_STATE; }
    SPH_TRY(nnps_need_tables(c));
    DevArray &A = c->arr[id];
    if (A.n == 0) return SPH_OK;
    HIP_TRY(hipMemcpyAsync(perm, A.perm.ptr, A.n * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SPH_OK;
}

// ---------------------------------------------------------------------------
// neighbour lists (query API; the pair kernels never materialise these)
// LinkedListNNPS.find_nearest_neighbors  linked_list_nnps.pyx:92-196
// ---------------------------------------------------------------------------
template <bool FILL>
__global__ __launch_bounds__(256) void k_csr(const double *__restrict__ dx, const double *__restrict__ dy,
                                             const double *__restrict__ dz, const double *__restrict__ dh, size_t nd,
                                             const double *__restrict__ sx, const double *__restrict__ sy,
                                             const double *__restrict__ sz, const double *__restrict__ sh,
                                             const uint32_t *__restrict__ sperm, const uint32_t *__restrict__ scell_start,
                                             GridDesc g, double radius_scale, uint32_t *__restrict__ start,
                                             uint32_t *__restrict__ nbrs)
{
#pragma clang fp contract(off) // r2 must round like the reference's norm2 (nnps_base.pxd:36-37)
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nd) return;
    double x = dx[i], y = dy[i], z = dz[i];
    // (clamped like the keys: a particle outside the grid lives in its outermost cells)
    int cx = min(max((int)floor(cell_coord(x, g.xmin[0], g)), 0), g.nc[0] - 1);
    int cy = min(max((int)floor(cell_coord(y, g.xmin[1], g)), 0), g.nc[1] - 1);
    int cz = min(max((int)floor(cell_coord(z, g.xmin[2], g)), 0), g.nc[2] - 1);
    double hi2 = radius_scale * dh[i];
    hi2 *= hi2;
    uint32_t count = 0;
    uint32_t base = FILL ? start[i] : 0;
    for (int oz = -1; oz <= 1; oz++)
        for (int oy = -1; oy <= 1; oy++) {
            int yy = cy + oy, zz = cz + oz;
            if (yy < 0 || yy >= g.nc[1] || zz < 0 || zz >= g.nc[2]) continue;
            int xa = max(cx - 1, 0), xb = min(cx + 1, g.nc[0] - 1);
            if (xa > xb) continue;
            uint32_t row = (uint32_t)(g.nc[0] * (yy + g.nc[1] * zz));
            uint32_t j0 = scell_start[row + xa], j1 = scell_start[row + xb + 1];
            for (uint32_t j = j0; j < j1; j++) {
                uint32_t s = sperm[j];
                double hj2 = radius_scale * sh[s];
                hj2 *= hj2;
                double ex = sx[s] - x, ey = sy[s] - y, ez = sz[s] - z;
                double r2 = ex * ex + ey * ey + ez * ez;
                if ((r2 < hi2) || (r2 < hj2)) {
                    if (FILL) nbrs[base + count] = s;
                    count++;
                }
            }
        }
    if (!FILL) start[i] = count;
}

extern "C" int sph_nnps_get_csr(sph_ctx *c, int src, int dst, uint32_t *start, size_t start_len, uint32_t *nbrs,
                                size_t nbrs_len, size_t *total)
{
    if (!c->nnps_valid) { sph_set_error("sph_nnps_get_csr: call sph_nnps_update first"); return SPH_ERR_STATE; }
    if (!start || !total) { sph_set_error("sph_nnps_get_csr: NULL start/total"); return SPH_ERR_ARG; }
    if (src < 0 || src >= SPH_MAX_ARRAYS || dst < 0 || dst >= SPH_MAX_ARRAYS || c->arr[src].nnps_slot < 0 ||
        c->arr[dst].nnps_slot < 0) {
        sph_set_error("sph_nnps_get_csr: arrays %d/%d are not part of the current grid", src, dst);
        return SPH_ERR_ARG;
    }
    HIP_TRY(hipSetDevice(c->device));
    SPH_TRY(nnps_need_tables(c));
    DevArray &S = c->arr[src], &D = c->arr[dst];
    size_t nd = D.n;
    // the caller sized `start` from ITS idea of the particle count; the device
    // array may hold ghosts the host never saw
    if (start_len != nd + 1) {
        sph_set_error("sph_nnps_get_csr: start has %zu entries, destination %d holds %zu particles on the device (needs %zu)",
                      start_len, dst, nd, nd + 1);
        return SPH_ERR_ARG;
    }
    GridDesc g;
    for (int k = 0; k < 3; k++) { g.xmin[k] = c->xmin[k]; g.nc[k] = c->nc[k]; }
    grid_set_cell(g, c->cell_size);
    g.park_base = g.park_bins = g.park_n = 0;
    SPH_TRY(c->tmp_u32a.reserve((nd + 1) * 4));
    uint32_t *d_start = c->tmp_u32a.as<uint32_t>();
    // variant 6 (default): the lists come from the wave-tile pair kernel itself (nnps_csr_pair_kernel);
    // variant 0 keeps the plain per-particle 27-cell walk as the independent cross-check
    const bool wave = c->pair_variant == 6;
    if (!wave && c->ghosts_binned) { sph_set_error("sph_nnps_get_csr: the per-particle cell walk (pair_variant 0) does not read ghost segments"); return SPH_ERR_UNSUPPORTED; }
    if (!nbrs) {
        if (nd && wave) SPH_TRY(nnps_csr_pair_kernel(c, src, dst, d_start, nullptr, nullptr));
        else if (nd)
            hipLaunchKernelGGL(k_csr<false>, dim3(div_up(nd, 256)), dim3(256), 0, c->stream, D.prop[SPH_X], D.prop[SPH_Y],
                               D.prop[SPH_Z], D.prop[SPH_H], nd, S.prop[SPH_X], S.prop[SPH_Y], S.prop[SPH_Z],
                               S.prop[SPH_H], S.perm.as<uint32_t>(), S.cell_start.as<uint32_t>(), g, c->radius_scale,
                               d_start, (uint32_t *)nullptr);
        std::vector<uint32_t> cnt(nd);
        if (nd) HIP_TRY(hipMemcpyAsync(cnt.data(), d_start, nd * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        start[0] = 0;
        for (size_t i = 0; i < nd; i++) start[i + 1] = start[i] + cnt[i];
        *total = start[nd];
        return SPH_OK;
    }
    size_t tot = start[nd];
    if (nbrs_len < tot) { sph_set_error("sph_nnps_get_csr: nbrs has %zu entries, %zu needed", nbrs_len, tot); return SPH_ERR_ARG; }
    SPH_TRY(c->tmp_u32b.reserve((tot + 1) * 4));
    HIP_TRY(hipMemcpyAsync(d_start, start, (nd + 1) * 4, hipMemcpyHostToDevice, c->stream));
    if (nd && wave) SPH_TRY(nnps_csr_pair_kernel(c, src, dst, nullptr, d_start, c->tmp_u32b.as<uint32_t>()));
    else if (nd)
        hipLaunchKernelGGL(k_csr<true>, dim3(div_up(nd, 256)), dim3(256), 0, c->stream, D.prop[SPH_X], D.prop[SPH_Y],
                           D.prop[SPH_Z], D.prop[SPH_H], nd, S.prop[SPH_X], S.prop[SPH_Y], S.prop[SPH_Z], S.prop[SPH_H],
                           S.perm.as<uint32_t>(), S.cell_start.as<uint32_t>(), g, c->radius_scale, d_start,
                           c->tmp_u32b.as<uint32_t>());
    if (tot) HIP_TRY(hipMemcpyAsync(nbrs, c->tmp_u32b.ptr, tot * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < nd; i++) std::sort(nbrs + start[i], nbrs + start[i + 1]);
    if (total) *total = tot;
    return SPH_OK;
}

// Device-resident CSR neighbour lists (original particle indices, cell-traversal
// order) for generated loop_all equations: counts -> exclusive scan -> fill.
int nnps_build_csr_device(sph_ctx *c, int src, int dst, DevBuf &start, DevBuf &nbrs, size_t *total)
{
    if (!c->nnps_valid) { sph_set_error("neighbour lists: call sph_nnps_update first"); return SPH_ERR_STATE; }
    if (c->arr[src].nnps_slot < 0 || c->arr[dst].nnps_slot < 0) {
        sph_set_error("neighbour lists: arrays %d/%d are not part of the current grid", src, dst);
        return SPH_ERR_ARG;
    }
    SPH_TRY(nnps_need_tables(c));
    DevArray &S = c->arr[src], &D = c->arr[dst];
    const size_t nd = D.n;
    GridDesc g;
    for (int k = 0; k < 3; k++) { g.xmin[k] = c->xmin[k]; g.nc[k] = c->nc[k]; }
    grid_set_cell(g, c->cell_size);
    g.park_base = g.park_bins = g.park_n = 0;
    SPH_TRY(start.reserve((nd + 2) * 4));
    SPH_TRY(c->tmp_u32a.reserve((nd + 2) * 4));
    *total = 0;
    if (nd == 0) return SPH_OK;
    uint32_t *cnt = c->tmp_u32a.as<uint32_t>();
    HIP_TRY(hipMemsetAsync(cnt + nd, 0, 4, c->stream));
    const bool wave = c->pair_variant == 6;
    if (!wave && c->ghosts_binned) { sph_set_error("neighbour lists: the per-particle cell walk (pair_variant 0) does not read ghost segments"); return SPH_ERR_UNSUPPORTED; }
    if (wave) SPH_TRY(nnps_csr_pair_kernel(c, src, dst, cnt, nullptr, nullptr));
    else
    hipLaunchKernelGGL(k_csr<false>, dim3(div_up(nd, 256)), dim3(256), 0, c->stream, D.prop[SPH_X], D.prop[SPH_Y],
                       D.prop[SPH_Z], D.prop[SPH_H], nd, S.prop[SPH_X], S.prop[SPH_Y], S.prop[SPH_Z], S.prop[SPH_H],
                       S.perm.as<uint32_t>(), S.cell_start.as<uint32_t>(), g, c->radius_scale, cnt, (uint32_t *)nullptr);
    SPH_TRY(dev_scan_u32(c, cnt, start.as<uint32_t>(), nd + 1, true));
    uint32_t *pin = (uint32_t *)c->pinned;
    HIP_TRY(hipMemcpyAsync(pin, start.as<uint32_t>() + nd, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    *total = pin[0];
    SPH_TRY(nbrs.reserve(((size_t)pin[0] + 1) * 4));
    if (wave) return nnps_csr_pair_kernel(c, src, dst, nullptr, start.as<uint32_t>(), nbrs.as<uint32_t>());
    hipLaunchKernelGGL(k_csr<true>, dim3(div_up(nd, 256)), dim3(256), 0, c->stream, D.prop[SPH_X], D.prop[SPH_Y],
                       D.prop[SPH_Z], D.prop[SPH_H], nd, S.prop[SPH_X], S.prop[SPH_Y], S.prop[SPH_Z], S.prop[SPH_H],
                       S.perm.as<uint32_t>(), S.cell_start.as<uint32_t>(), g, c->radius_scale, start.as<uint32_t>(),
                       nbrs.as<uint32_t>());
    return SPH_OK;
}

// ---------------------------------------------------------------------------
// physical reorder into cell order
// ---------------------------------------------------------------------------
// dst[i] = src[perm[i]] for i < n, 0 for n <= i < cap (the slack behind the particles stays zero without a memset)
__global__ __launch_bounds__(256) void k_gather_f64(const double *__restrict__ src, const uint32_t *__restrict__ perm,
                                                    size_t n, size_t cap, double *__restrict__ dst)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[perm[i]];
    else if (i < cap) dst[i] = 0.0;
}

// DeviceHelper.align(indices) (pysph/base/device_helper.py:241-288): new particle i
// takes every property of old particle indices[i]; n becomes n_new (<= old n drops
// particles: remove_particles / remove_tagged_particles are "keep list" gathers).
extern "C" int sph_array_permute(sph_ctx *c, int id, const uint32_t *indices, size_t n_new, size_t n_real_new)
{
    if (!c || id < 0 || id >= SPH_MAX_ARRAYS || (n_new && !indices) || n_real_new > n_new) {
        sph_set_error("sph_array_permute: bad arguments");
        return SPH_ERR_ARG;
    }
    HIP_TRY(hipSetDevice(c->device));
    DevArray &A = c->arr[id];
    if (n_new > A.n) { sph_set_error("sph_array_permute: %zu indices for %zu particles", n_new, A.n); return SPH_ERR_ARG; }
    for (size_t i = 0; i < n_new; i++)
        if (indices[i] >= A.n) { sph_set_error("sph_array_permute: index %u out of range (n=%zu)", indices[i], A.n); return SPH_ERR_ARG; }
    if (n_new) {
        SPH_TRY(c->aux.reserve(n_new * sizeof(uint32_t)));
        HIP_TRY(hipMemcpyAsync(c->aux.ptr, indices, n_new * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
        double *tmp = nullptr;
        HIP_TRY(hipMalloc((void **)&tmp, A.cap * sizeof(double)));
        for (int p = 0; p < SPH_PROP_COUNT; p++) {
            if (!A.prop[p]) continue;
            hipLaunchKernelGGL(k_gather_f64, dim3(div_up(A.cap, 256)), dim3(256), 0, c->stream, A.prop[p],
                               c->aux.as<uint32_t>(), n_new, A.cap, tmp);
            double *old = A.prop[p];
            A.prop[p] = tmp;
            tmp = old;
        }
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipFree(tmp));
    }
    A.perm_n = 0;
    A.perm_direct_n = 0; A.unordered = false; // another memory order
    c->nnps_valid = false;
    return sph_array_resize(c, id, n_new, n_real_new);
}

extern "C" int sph_nnps_reorder_array(sph_ctx *c, int id)
{
    if (!c || id < 0 || id >= SPH_MAX_ARRAYS) { sph_set_error("sph_nnps_reorder_array: bad arguments"); return SPH_ERR_ARG; }
    // the cell order of the LAST sph_nnps_update is applied (as the reference's
    // spatially_order_particles uses the cell lists of its last update); several
    // arrays can be reordered back to back, each at most once per update
    if (c->arr[id].nnps_slot < 0 || c->arr[id].perm_n != c->arr[id].n) {
        if (c->arr[id].n == 0) return SPH_OK;
        sph_set_error("sph_nnps_reorder_array: array not binned (call sph_nnps_update first; one reorder per update)");
        return SPH_ERR_STATE;
    }
    HIP_TRY(hipSetDevice(c->device));
    SPH_TRY(nnps_need_tables(c));
    DevArray &A = c->arr[id];
    if (A.n == 0) return SPH_OK;
    if (A.n_real != A.n) {
        // real particles must stay first (particle_array.pyx:1092): ghosts are
        // transient (halo / periodic images) and are rebuilt after a reorder
        sph_set_error("sph_nnps_reorder_array: drop ghost particles first (n=%zu, n_real=%zu)", A.n, A.n_real);
        return SPH_ERR_STATE;
    }
    double *tmp = nullptr;
    HIP_TRY(hipMalloc((void **)&tmp, A.cap * sizeof(double)));
    for (int p = 0; p < SPH_PROP_COUNT; p++) {
        if (!A.prop[p]) continue;
        hipLaunchKernelGGL(k_gather_f64, dim3(div_up(A.cap, 256)), dim3(256), 0, c->stream, A.prop[p], A.perm.as<uint32_t>(),
                           A.n, A.cap, tmp);
        double *old = A.prop[p];
        A.prop[p] = tmp;
        tmp = old;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipFree(tmp));
    A.perm_n = 0;
    A.perm_direct_n = 0; A.unordered = false; // the array lies in cell order now
    c->nnps_valid = false;
    return SPH_OK;
}
