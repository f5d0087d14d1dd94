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

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cfloat>
#include <cmath>

// ---------------------------------------------------------------------------
// min/max of x, y, z, h
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

struct BlockRanges { int first[SPH_MAX_ARRAYS + 1]; int narrays; }; // blocks [first[a], first[a+1]) belong to array a

// several arrays in ONE launch (a dam break has three: three short launches and their gaps cost more than the passes)
struct MinMaxMulti {
    const double *x[SPH_MAX_ARRAYS], *y[SPH_MAX_ARRAYS], *z[SPH_MAX_ARRAYS], *h[SPH_MAX_ARRAYS], *m[SPH_MAX_ARRAYS];
    size_t n[SPH_MAX_ARRAYS];
    BlockRanges br;
};

// part[block][8] = {xmin ymin zmin hmin xmax ymax zmax hmax}; partm[block][2] = {mmin mmax} of this array (m may be null)
__global__ __launch_bounds__(256) void k_minmax(MinMaxMulti t, double *__restrict__ part, double *__restrict__ partm)
{
    // this block's array (wave-uniform) and its share of it
    int a = 0;
    for (int k = 1; k < SPH_MAX_ARRAYS; k++) if (k < t.br.narrays && (int)blockIdx.x >= t.br.first[k]) a = k;
    const double *__restrict__ x = t.x[a], *__restrict__ y = t.y[a], *__restrict__ z = t.z[a], *__restrict__ h = t.h[a];
    const double *__restrict__ m = t.m[a];
    const size_t n = t.n[a];
    const size_t lb = blockIdx.x - t.br.first[a], nba = t.br.first[a + 1] - t.br.first[a];
    double mn[5] = {DBL_MAX, DBL_MAX, DBL_MAX, DBL_MAX, DBL_MAX};
    double mx[5] = {-DBL_MAX, -DBL_MAX, -DBL_MAX, -DBL_MAX, -DBL_MAX};
    const double *p[4] = {x, y, z, h};
    for (size_t i = lb * blockDim.x + threadIdx.x; i < n; i += nba * blockDim.x) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            double v = p[k][i];
            mn[k] = fmin(mn[k], v);
            mx[k] = fmax(mx[k], v);
        }
        if (m) { const double v = m[i]; mn[4] = fmin(mn[4], v); mx[4] = fmax(mx[4], v); }
    }
    __shared__ double s[4][10];
    int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        double a = wave_min(mn[k]), b = wave_max(mx[k]);
        if (lane == 0) {
            if (k < 4) { s[wv][k] = a; s[wv][4 + k] = b; }
            else { s[wv][8] = a; s[wv][9] = b; }
        }
    }
    __syncthreads();
    if (threadIdx.x < 10) {
        int k = threadIdx.x;
        const bool is_min = k < 4 || k == 8;
        double v = s[0][k];
        for (int w = 1; w < 4; w++) v = is_min ? fmin(v, s[w][k]) : fmax(v, s[w][k]);
        if (k < 8) part[(size_t)blockIdx.x * 8 + k] = v;
        else partm[(size_t)blockIdx.x * 2 + (k - 8)] = v;
    }
}


__global__ __launch_bounds__(512) void k_minmax_final(const double *__restrict__ part, int nblocks, double *__restrict__ out,
                                                      const double *__restrict__ partm, BlockRanges br)
{
    static_assert(SPH_MAX_ARRAYS <= 8, "one wavefront of this block per array");
    // the mass range of every array: wavefront a over the partials of array a -> out[8 + 2 a] = {mmin, mmax}
    {
        const int a = threadIdx.x >> 6, l = threadIdx.x & 63;
        if (a < br.narrays) {
            double lo = DBL_MAX, hi = -DBL_MAX;
            for (int b = br.first[a] + l; b < br.first[a + 1]; b += 64) { lo = fmin(lo, partm[2 * b]); hi = fmax(hi, partm[2 * b + 1]); }
            lo = wave_min(lo); hi = wave_max(hi);
            if (l == 0) { out[8 + 2 * a] = lo; out[8 + 2 * a + 1] = hi; }
        }
    }
    // 64 groups of 8 lanes stride over the partials (the single-wavefront
    // version spent 35-70 us on a chain of dependent loads)
    __shared__ double s[64][8];
    const int k = threadIdx.x & 7, g = threadIdx.x >> 3;
    double v = (k < 4) ? DBL_MAX : -DBL_MAX;
    for (int b = g; b < nblocks; b += 64) {
        double w = part[(size_t)b * 8 + k];
        v = (k < 4) ? fmin(v, w) : fmax(v, w);
    }
    s[g][k] = v;
    __syncthreads();
    if (threadIdx.x < 8) {
        for (int q = 1; q < 64; q++) v = (k < 4) ? fmin(v, s[q][k]) : fmax(v, s[q][k]);
        out[k] = v;
    }
}

int nnps_minmax(sph_ctx *c, int narrays, const int *ids, double *out8)
{
    const int BLOCKS = 1024;
    // partials: [narrays * BLOCKS][8] position / h, then [narrays * BLOCKS][2] mass; results: 8 + 2 per array
    SPH_TRY(c->red_part.reserve((size_t)narrays * BLOCKS * 10 * sizeof(double)));
    SPH_TRY(c->red_out.reserve((8 + 2 * SPH_MAX_ARRAYS) * sizeof(double)));
    double *const partm = c->red_part.as<double>() + (size_t)narrays * BLOCKS * 8;
    BlockRanges br;
    int *const first = br.first;
    br.narrays = narrays;
    int nb_total = 0;
    MinMaxMulti mt;
    memset(&mt, 0, sizeof mt);
    for (int a = 0; a < narrays; a++) {
        DevArray &A = c->arr[ids[a]];
        first[a] = nb_total;
        A.m_known = false;
        if (A.n == 0) continue;
        for (int p : {SPH_X, SPH_Y, SPH_Z, SPH_H})
            if (!A.prop[p]) {
                sph_set_error("nnps: array %d has no device copy of x/y/z/h", ids[a]);
                return SPH_ERR_MISSING_PROP;
            }
        int nb = (int)std::min<size_t>(BLOCKS, (A.n + 255) / 256);
        mt.x[a] = A.prop[SPH_X]; mt.y[a] = A.prop[SPH_Y]; mt.z[a] = A.prop[SPH_Z]; mt.h[a] = A.prop[SPH_H];
        mt.m[a] = c->want_mrange ? A.prop[SPH_M] : nullptr;
        mt.n[a] = A.n;
        nb_total += nb;
    }
    first[narrays] = nb_total;
    for (int a = narrays + 1; a <= SPH_MAX_ARRAYS; a++) first[a] = nb_total;
    mt.br = br;
    if (nb_total) hipLaunchKernelGGL(k_minmax, dim3(nb_total), dim3(256), 0, c->stream, mt, c->red_part.as<double>(), partm);
    if (nb_total == 0) {
        for (int k = 0; k < 4; k++) { out8[k] = DBL_MAX; out8[4 + k] = -DBL_MAX; }
        return SPH_OK;
    }
    // the mass range of every array rides on the same kernel and round trip (uniform-mass records of the WCSPH pair kernel)
    hipLaunchKernelGGL(k_minmax_final, dim3(1), dim3(512), 0, c->stream, c->red_part.as<double>(), nb_total,
                       c->red_out.as<double>(), partm, br);
    HIP_TRY(hipMemcpyAsync(c->pinned, c->red_out.ptr, (8 + 2 * narrays) * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    memcpy(out8, c->pinned, 8 * sizeof(double));
    for (int a = 0; a < narrays; a++) {
        DevArray &A = c->arr[ids[a]];
        const double lo = c->pinned[8 + 2 * a], hi = c->pinned[8 + 2 * a + 1];
        A.m_known = c->want_mrange && A.n > 0 && A.prop[SPH_M] && lo == hi && !A.m_mixed_ghosts;
        A.m_value = lo;
    }
    return SPH_OK;
}

// ---------------------------------------------------------------------------
// cell keys, sort, cell ranges
// ---------------------------------------------------------------------------
struct GridDesc {
    double xmin[3];
    double cell_size;
    int nc[3];
};

// nnps_base.pxd:39-57 real_to_int = <int>floor(real_val/step), flatten_raw :83-96
// The sort key is FINER than the reference's cell id: key = cell * SPH_NSUB +
// sub, sub = the particle's x position inside its cell in SPH_NSUB sub-bins.
// Cells stay the reference's (cell id = key / SPH_NSUB, cell_start per cell);
// the sub-bins only order the particles of a cell along x, so that a pair
// kernel can cut a destination's candidate range in a row of cells from three
// whole cells to its x window (fine_start per sub-bin).  Costs no extra radix
// pass (18 + 3 bits at 4 M particles).
__global__ __launch_bounds__(256) void k_cell_keys(const double *__restrict__ x, const double *__restrict__ y,
                                                   const double *__restrict__ z, size_t n, GridDesc g,
                                                   uint32_t *__restrict__ keys, uint32_t *__restrict__ idx,
                                                   uint32_t tag, uint32_t idx_base)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double ux = (x[i] - g.xmin[0]) / g.cell_size;
    int cx = (int)floor(ux);
    int cy = (int)floor((y[i] - g.xmin[1]) / g.cell_size);
    int cz = (int)floor((z[i] - g.xmin[2]) / g.cell_size);
    int sub = (int)floor((ux - (double)cx) * SPH_NSUB);
    if (cx < 0) { cx = 0; sub = 0; }
    if (cx > g.nc[0] - 1) { cx = g.nc[0] - 1; sub = SPH_NSUB - 1; }
    sub = min(max(sub, 0), SPH_NSUB - 1);
    cy = min(max(cy, 0), g.nc[1] - 1);
    cz = min(max(cz, 0), g.nc[2] - 1);
    keys[i] = ((uint32_t)(cx + g.nc[0] * (cy + g.nc[1] * cz)) * SPH_NSUB + (uint32_t)sub) | tag;
    idx[i] = (uint32_t)i + idx_base; // position in the concatenation of all arrays (merged-first build), else the local index
}

// the same for the concatenation of several arrays in ONE launch (merged-first build): block b belongs to array a with
// first[a] <= b < first[a + 1]; keys and values land at the array's offset in the concatenation, the value is the
// particle's position there
struct KeysMulti {
    const double *x[SPH_MAX_ARRAYS], *y[SPH_MAX_ARRAYS], *z[SPH_MAX_ARRAYS];
    uint32_t n[SPH_MAX_ARRAYS], off[SPH_MAX_ARRAYS], first[SPH_MAX_ARRAYS + 1];
    int narrays;
};
__global__ __launch_bounds__(256) void k_cell_keys_multi(KeysMulti t, GridDesc g, uint32_t *__restrict__ keys, uint32_t *__restrict__ idx)
{
    int a = 0;
    for (int k = 1; k < SPH_MAX_ARRAYS; k++) if (k < t.narrays && blockIdx.x >= t.first[k]) a = k;
    const size_t i = (size_t)(blockIdx.x - t.first[a]) * blockDim.x + threadIdx.x;
    if (i >= t.n[a]) return;
    const double *__restrict__ x = t.x[a], *__restrict__ y = t.y[a], *__restrict__ z = t.z[a];
    const double ux = (x[i] - g.xmin[0]) / g.cell_size;
    int cx = (int)floor(ux);
    int cy = (int)floor((y[i] - g.xmin[1]) / g.cell_size);
    int cz = (int)floor((z[i] - g.xmin[2]) / g.cell_size);
    int sub = (int)floor((ux - (double)cx) * SPH_NSUB);
    if (cx < 0) { cx = 0; sub = 0; }
    if (cx > g.nc[0] - 1) { cx = g.nc[0] - 1; sub = SPH_NSUB - 1; }
    sub = min(max(sub, 0), SPH_NSUB - 1);
    cy = min(max(cy, 0), g.nc[1] - 1);
    cz = min(max(cz, 0), g.nc[2] - 1);
    const size_t o = (size_t)t.off[a] + i;
    keys[o] = (uint32_t)(cx + g.nc[0] * (cy + g.nc[1] * cz)) * SPH_NSUB + (uint32_t)sub;
    idx[o] = (uint32_t)o;
}

// one array's segment of the concatenated sort: strip the array tag, split into the array's own tables
__global__ __launch_bounds__(256) void k_split_segment(const uint32_t *__restrict__ cat_keys, const uint32_t *__restrict__ cat_perm,
                                                       size_t n, uint32_t mask, uint32_t *__restrict__ fkeys,
                                                       uint32_t *__restrict__ keys, uint32_t *__restrict__ perm)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t f = cat_keys[i] & mask;
    fkeys[i] = f;
    keys[i] = f / SPH_NSUB;
    perm[i] = cat_perm[i];
}

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
__global__ __launch_bounds__(256) void k_tile_keys(const uint32_t *__restrict__ skeys, size_t n, uint32_t n_tiles, int ncx,
                                                   int ncy, int ncz, int by, uint32_t *__restrict__ key,
                                                   uint32_t *__restrict__ count)
{
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tiles) return;
    const uint32_t k = skeys[(size_t)t * SPH_TILE];
    const uint32_t row = k / (uint32_t)ncx;
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

static int bits_for(long n_cells)
{
    int b = 1;
    while ((1L << b) < n_cells) b++;
    return b;
}

extern "C" int sph_nnps_minmax(sph_ctx *c, int narrays, const int *ids, double *out8)
{
    if (!c || narrays < 1 || narrays > SPH_MAX_ARRAYS) { sph_set_error("sph_nnps_minmax: bad arguments"); return SPH_ERR_ARG; }
    HIP_TRY(hipSetDevice(c->device));
    return nnps_minmax(c, narrays, ids, out8);
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

template <class T> __device__ __forceinline__ T wave_incl_scan(T x, int lane)
{
    for (int o = 1; o < 64; o <<= 1) { const T t = __shfl_up(x, o, 64); if (lane >= o) x += t; }
    return x;
}

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

// prefix sums of n counters (device-wide, three launches); in == out allowed
template <class T> static int dev_scan(sph_ctx *c, const T *in, T *out, size_t n, bool exclusive)
{
    if (n == 0) return SPH_OK;
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

__global__ __launch_bounds__(256) void k_mass_differs(const double *__restrict__ m, size_t n, double mu, uint32_t *__restrict__ flag)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && m[i] != mu) *flag = 1u;
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
    const double ux = (x[i] - g.xmin[0]) / g.cell_size;
    int cx = (int)floor(ux);
    int cy = (int)floor((y[i] - g.xmin[1]) / g.cell_size);
    int cz = (int)floor((z[i] - g.xmin[2]) / g.cell_size);
    int sub = (int)floor((ux - (double)cx) * SPH_NSUB);
    if (cx < 0) { cx = 0; sub = 0; }
    if (cx > g.nc[0] - 1) { cx = g.nc[0] - 1; sub = SPH_NSUB - 1; }
    sub = min(max(sub, 0), SPH_NSUB - 1);
    cy = min(max(cy, 0), g.nc[1] - 1);
    cz = min(max(cz, 0), g.nc[2] - 1);
    const uint32_t key = (uint32_t)(cx + g.nc[0] * (cy + g.nc[1] * cz)) * SPH_NSUB + (uint32_t)sub;
    keys[i] = key;
    atomicAdd(&count[key + 2], 1u);
}


// ---------------------------------------------------------------------------
// merged-first build (sph_nnps_update) and the per-array tables derived from it on demand
// ---------------------------------------------------------------------------
struct CatOff { uint32_t off[SPH_MAX_ARRAYS + 1]; int narrays; }; // first position of every array in the concatenation

// The merged order straight from the ONE stable sort of all arrays' fine keys: the sorted value is the particle's
// position in the concatenation of the arrays -> its slot and its original index there.
__global__ __launch_bounds__(256) void k_merged_split(const uint32_t *__restrict__ skeys, const uint32_t *__restrict__ svals, size_t n,
                                                      CatOff co, uint32_t *__restrict__ fkeys, uint32_t *__restrict__ keys,
                                                      uint32_t *__restrict__ perm, uint8_t *__restrict__ slot)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t f = skeys[i], gpos = svals[i];
    uint32_t s = 0, base = 0;
#pragma unroll
    for (int b = 1; b < SPH_MAX_ARRAYS; b++)
        if (b < co.narrays && gpos >= co.off[b]) { s = (uint32_t)b; base = co.off[b]; }
    fkeys[i] = f;
    keys[i] = f / SPH_NSUB;
    perm[i] = gpos - base;
    slot[i] = (uint8_t)s;
}

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

static int nnps_reserve_tables(sph_ctx *c, DevArray &A)
{
    const size_t n = A.n, n_fine = (size_t)c->n_cells * SPH_NSUB;
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
static int nnps_finish_tables(sph_ctx *c, DevArray &A, bool with_keys)
{
    const size_t n = A.n, n_fine = (size_t)c->n_cells * SPH_NSUB;
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
        SPH_TRY(nnps_reserve_tables(c, A));
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
        if (A.n == 0) { nnps_empty_tables(c, A); continue; }
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
static int nnps_tile_order(sph_ctx *c, DevArray &A, size_t n)
{
    const bool want_tiles = c->tile_block_rows > 0 && c->nc[2] > 1 && n > 64 * SPH_TILE;
    const int tsig = (int)c->tile_block_rows;
    const uint32_t nt_now = (uint32_t)div_up(n, SPH_TILE);
    if (want_tiles && A.n_tiles == nt_now && A.tile_grid[0] == c->nc[0] && A.tile_grid[1] == c->nc[1] &&
        A.tile_grid[2] == c->nc[2] && A.tile_grid[3] == tsig && ++A.tile_age < 16)
        return SPH_OK;
    A.n_tiles = 0;
    if (want_tiles) {
        A.tile_age = 0;
        A.tile_grid[0] = c->nc[0]; A.tile_grid[1] = c->nc[1]; A.tile_grid[2] = c->nc[2]; A.tile_grid[3] = tsig;
        const uint32_t nt = nt_now;
        SPH_TRY(A.tile_key.reserve((size_t)nt * 4 * 2));
        SPH_TRY(A.tile_id.reserve((size_t)nt * 4));
        SPH_TRY(A.tile_order.reserve((size_t)nt * 4));
        // keys = traversal rank of the tile's row (< ncy * ncz rounded up to whole blocks of rows): the same counting sort
        const uint32_t by = (uint32_t)c->tile_block_rows;
        const uint32_t nrow_keys = ((uint32_t)c->nc[1] + by - 1) / by * by * (uint32_t)c->nc[2];
        SPH_TRY(A.tile_id.reserve(((size_t)nrow_keys + 2) * 4));
        uint32_t *tk = A.tile_key.as<uint32_t>(), *tt = A.tile_id.as<uint32_t>();
        HIP_TRY(hipMemsetAsync(tt, 0, ((size_t)nrow_keys + 2) * 4, c->stream));
        hipLaunchKernelGGL(k_tile_keys, dim3(div_up(nt, 256)), dim3(256), 0, c->stream, A.keys_sorted.as<uint32_t>(), n, nt,
                           c->nc[0], c->nc[1], c->nc[2], (int)c->tile_block_rows, tk, tt);
        BinFixArgs fa;
        memset(&fa, 0, sizeof fa);
        fa.nbins = nrow_keys; fa.nsub = 0;
        fa.perm = A.tile_order.as<uint32_t>();
        SPH_TRY(nnps_bin_sort_finish(c, tk, nt, fa, tt));
        A.n_tiles = nt;
    }
    return SPH_OK;
}

// fine x index (cell * SPH_NSUB + sub-bin along a row) beyond which ghosts lie, from the slab faces the host named
// (sph_nnps_set_ghost_faces) and the grid of the current update
static void nnps_face_planes(sph_ctx *c)
{
    c->gfx_lo = -0x7fffffff; c->gfx_hi = 0x7fffffff; // no faces named: no wavefront is a face wavefront
    if (c->face_axis < 0) return;
    if (c->face_axis != 0) { c->gfx_lo = 0x7fffffff; c->gfx_hi = -0x7fffffff; return; } // rows run along x: every wavefront may see ghosts
    const double binw = c->cell_size / SPH_NSUB;
    const double flo = floor((c->face_lo - c->xmin[0]) / binw), fhi = floor((c->face_hi - c->xmin[0]) / binw);
    c->gfx_lo = !(flo > -1e9) ? -0x7fffffff : (flo > 1e9 ? 0x7fffffff : (int)flo); // ghosts: fx <= gfx_lo ...
    c->gfx_hi = !(fhi < 1e9) ? 0x7fffffff : (fhi < -1e9 ? -0x7fffffff : (int)fhi);  // ... or fx >= gfx_hi
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
    double mm[8];
    if (bounds && c->h_known[1] >= 0.0) {
        // the grid is given and the h range known (sph_nnps_set_h_range): no reduction, no round trip
        for (int k = 0; k < 3; k++) { mm[k] = bounds[k]; mm[4 + k] = bounds[3 + k]; }
        mm[3] = c->h_known[0]; mm[7] = c->h_known[1];
        for (int a = 0; a < narrays; a++) {
            DevArray &A = c->arr[ids[a]];
            for (int p : {SPH_X, SPH_Y, SPH_Z, SPH_H})
                if (A.n && !A.prop[p]) { sph_set_error("nnps: array %d has no device copy of x/y/z/h", ids[a]); return SPH_ERR_MISSING_PROP; }
            A.m_known = false; // nobody looked at the masses (ghosts may have been appended since the last look)
        }
    } else {
        SPH_TRY(nnps_minmax(c, narrays, ids, mm));
        if (c->h_known[1] >= 0.0) { mm[3] = c->h_known[0]; mm[7] = c->h_known[1]; }
    }

    // DomainManager._compute_cell_size_for_binning (nnps_base.pyx:942-978)
    double hmax = -1.0, hmin = DBL_MAX;
    if (mm[7] > hmax) hmax = mm[7];
    if (mm[3] < hmin) hmin = mm[3];
    double cell_size = radius_scale * hmax;
    c->hmin = radius_scale * hmin;
    if (cell_size < 1e-6) cell_size = 1.0;
    if (cell_size_in > 0) cell_size = cell_size_in;
    c->uniform_h = (hmin == hmax);
    c->h_uniform = hmax;

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

    c->dim = dim;
    c->narrays = narrays;
    c->radius_scale = radius_scale;
    c->cell_size = cell_size;
    c->xmin[0] = xmin; c->xmin[1] = ymin; c->xmin[2] = zmin;
    c->xmax[0] = xmax; c->xmax[1] = ymax; c->xmax[2] = zmax;
    c->nc[0] = ncx; c->nc[1] = ncy; c->nc[2] = ncz;
    c->n_cells = n_cells_alloc;
    for (auto &A : c->arr) A.nnps_slot = -1;

    GridDesc g;
    for (int k = 0; k < 3; k++) { g.xmin[k] = c->xmin[k]; g.nc[k] = c->nc[k]; }
    g.cell_size = cell_size;
    const size_t n_fine = (size_t)n_cells_alloc * SPH_NSUB;
    int end_bit = bits_for((long)n_fine);

    // Several arrays (a dam break has three): ONE radix sort of all their keys -- rocPRIM sorts fewer than 1 Mi keys
    // with a merge sort of ~25 launch pairs per array, which is what three separate sorts cost.
    //  * merged-first (option merge_arrays, default): the keys carry NO array tag.  The sort is stable and the arrays are
    //    concatenated in slot order, so equal fine keys keep slot order: the sorted sequence IS the merged order of all
    //    arrays (sph_ctx::merged) the multi-array pair kernel runs on; the sorted value (position in the concatenation)
    //    gives slot and original index.  Per-array tables are derived from it only when something asks for them
    //    (nnps_need_tables: per-destination pair paths, neighbour-list queries, reorder) -- a steady-state dam-break
    //    step builds ONE fine_start table instead of four, with two sort bits less (a radix pass at 4 M particles).
    //  * otherwise: the array's slot in the bits above the cell key, one segment of the sorted sequence per array.
    size_t n_cat = 0, cat_off[SPH_MAX_ARRAYS + 1] = {};
    int tag_bits = 0, n_nonempty = 0;
    for (int a = 0; a < narrays; a++) { cat_off[a] = n_cat; n_cat += c->arr[ids[a]].n; n_nonempty += c->arr[ids[a]].n > 0; }
    cat_off[narrays] = n_cat;
    while ((1 << tag_bits) < narrays) tag_bits++;
    const bool cat = n_nonempty > 1 && end_bit + tag_bits <= 32 && n_cat < (1ull << 31);
    const bool merged_first = cat && c->merge_arrays;
    const size_t n_half = (n_cat + 63) & ~(size_t)63; // keys | values halves of the scratch buffers, 256-B aligned
    if (!c->gapq.ptr) { // first use: the queue starts empty; afterwards k_coarse_start leaves it empty
        SPH_TRY(c->gapq.reserve((4 + 3 * GAP_QUEUE) * 4));
        HIP_TRY(hipMemsetAsync(c->gapq.ptr, 0, 16, c->stream));
    }
    if (cat) {
        SPH_TRY(c->tmp_u32a.reserve((n_half + 64) * 4 * 2));
        SPH_TRY(c->tmp_u32b.reserve((n_half + 64) * 4 * 2));
        uint32_t *ck = c->tmp_u32a.as<uint32_t>(), *ci = ck + n_half, *cks = c->tmp_u32b.as<uint32_t>(), *cp = cks + n_half;
        if (merged_first) { // one launch for all arrays
            KeysMulti km;
            memset(&km, 0, sizeof km);
            km.narrays = narrays;
            uint32_t nbk = 0;
            for (int a = 0; a < narrays; a++) {
                DevArray &A = c->arr[ids[a]];
                km.x[a] = A.prop[SPH_X]; km.y[a] = A.prop[SPH_Y]; km.z[a] = A.prop[SPH_Z];
                km.n[a] = (uint32_t)A.n; km.off[a] = (uint32_t)cat_off[a]; km.first[a] = nbk;
                nbk += div_up(A.n, 256);
            }
            for (int a = narrays; a <= SPH_MAX_ARRAYS; a++) km.first[a] = nbk;
            hipLaunchKernelGGL(k_cell_keys_multi, dim3(nbk), dim3(256), 0, c->stream, km, g, ck, ci);
        } else
        for (int a = 0; a < narrays; a++) {
            DevArray &A = c->arr[ids[a]];
            if (A.n == 0) continue;
            hipLaunchKernelGGL(k_cell_keys, dim3(div_up(A.n, 256)), dim3(256), 0, c->stream, A.prop[SPH_X], A.prop[SPH_Y],
                               A.prop[SPH_Z], A.n, g, ck + cat_off[a], ci + cat_off[a],
                               end_bit < 32 ? (uint32_t)a << end_bit : 0u, 0u);
        }
        const int sort_bits = merged_first ? end_bit : end_bit + tag_bits;
        size_t tmp_bytes = 0;
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, ck, cks, ci, cp, (int)n_cat, 0, sort_bits, c->stream));
        SPH_TRY(c->cub_tmp.reserve(tmp_bytes));
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(c->cub_tmp.ptr, tmp_bytes, ck, cks, ci, cp, (int)n_cat, 0, sort_bits, c->stream));
    }
    c->merged_valid = false;
    c->tables_valid = true;
    c->ghosts_binned = false;
    nnps_face_planes(c); // ghost split: where this step's ghosts will lie (known before they arrive)
    for (int a = 0; a < narrays; a++) {
        c->ids[a] = ids[a];
        c->arr[ids[a]].nnps_slot = a;
        c->arr[ids[a]].perm_n = c->arr[ids[a]].n;
        c->arr[ids[a]].n_binned = c->arr[ids[a]].n;
        c->arr[ids[a]].g_n = 0;
    }
    if (merged_first) {
        DevArray &M = c->merged;
        M.n = M.n_real = n_cat;
        SPH_TRY(M.fkeys_sorted.reserve((n_cat + 1) * 4));
        SPH_TRY(M.keys_sorted.reserve((n_cat + 1) * 4));
        SPH_TRY(M.perm.reserve((n_cat + 1) * 4));
        SPH_TRY(M.slot8.reserve(n_cat + 64));
        SPH_TRY(M.fine_start.reserve((n_fine + 1) * 4));
        SPH_TRY(M.cell_start.reserve(((size_t)n_cells_alloc + 1) * 4));
        CatOff co;
        co.narrays = narrays;
        for (int a = 0; a <= SPH_MAX_ARRAYS; a++) co.off[a] = (uint32_t)cat_off[a < narrays ? a : narrays];
        const uint32_t *cks = c->tmp_u32b.as<uint32_t>();
        hipLaunchKernelGGL(k_merged_split, dim3(div_up(n_cat, 256)), dim3(256), 0, c->stream, cks, cks + n_half, n_cat, co,
                           M.fkeys_sorted.as<uint32_t>(), M.keys_sorted.as<uint32_t>(), M.perm.as<uint32_t>(), M.slot8.as<uint8_t>());
        hipLaunchKernelGGL(k_cell_start, dim3(div_up(n_cat + 1, 256)), dim3(256), 0, c->stream, M.fkeys_sorted.as<uint32_t>(),
                           n_cat, (uint32_t)n_fine, M.fine_start.as<uint32_t>(), c->gapq.as<uint32_t>(), (uint32_t *)nullptr);
        hipLaunchKernelGGL(k_fill_gaps, dim3(512), dim3(256), 0, c->stream, c->gapq.as<uint32_t>(), M.fine_start.as<uint32_t>());
        hipLaunchKernelGGL(k_coarse_start, dim3(div_up((size_t)n_cells_alloc + 1, 256)), dim3(256), 0, c->stream,
                           M.fine_start.as<uint32_t>(), (uint32_t)n_cells_alloc, M.cell_start.as<uint32_t>(), c->gapq.as<uint32_t>());
        SPH_TRY(nnps_tile_order(c, M, n_cat));
        c->merged_valid = true;
        c->tables_valid = false;
        if (!c->lazy_tables) SPH_TRY(nnps_need_tables(c));
    } else {
    for (int a = 0; a < narrays; a++) {
        DevArray &A = c->arr[ids[a]];
        size_t n = A.n;
        SPH_TRY(A.keys.reserve((n + 1) * 4));
        SPH_TRY(A.idx.reserve((n + 1) * 4));
        SPH_TRY(nnps_reserve_tables(c, A));
        if (n == 0) { nnps_empty_tables(c, A); continue; }
        if (cat) {
            const uint32_t *cks = c->tmp_u32b.as<uint32_t>();
            hipLaunchKernelGGL(k_split_segment, dim3(div_up(n, 256)), dim3(256), 0, c->stream, cks + cat_off[a],
                               cks + n_half + cat_off[a], n, end_bit < 32 ? (1u << end_bit) - 1u : 0xffffffffu,
                               A.fkeys_sorted.as<uint32_t>(), A.keys_sorted.as<uint32_t>(), A.perm.as<uint32_t>());
        } else {
        hipLaunchKernelGGL(k_cell_keys, dim3(div_up(n, 256)), dim3(256), 0, c->stream, A.prop[SPH_X], A.prop[SPH_Y],
                           A.prop[SPH_Z], n, g, A.keys.as<uint32_t>(), A.idx.as<uint32_t>(), 0u, 0u);
        size_t tmp_bytes = 0;
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, A.keys.as<uint32_t>(), A.fkeys_sorted.as<uint32_t>(),
                                                   A.idx.as<uint32_t>(), A.perm.as<uint32_t>(), (int)n, 0, end_bit,
                                                   c->stream));
        SPH_TRY(c->cub_tmp.reserve(tmp_bytes));
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(c->cub_tmp.ptr, tmp_bytes, A.keys.as<uint32_t>(),
                                                   A.fkeys_sorted.as<uint32_t>(), A.idx.as<uint32_t>(),
                                                   A.perm.as<uint32_t>(), (int)n, 0, end_bit, c->stream));
        }
        // the single-array path gets its cell ids (keys_sorted) from k_cell_start; the concatenated one has them already
        SPH_TRY(nnps_finish_tables(c, A, !cat));
    }
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
    g.cell_size = c->cell_size;
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
        hipLaunchKernelGGL(k_cell_keys_count, dim3(div_up(ng, 256)), dim3(256), 0, c->stream, A.prop[SPH_X] + nb, A.prop[SPH_Y] + nb,
                           A.prop[SPH_Z] + nb, ng, g, A.g_keys.as<uint32_t>(), T);
        BinFixArgs fa;
        memset(&fa, 0, sizeof fa);
        fa.nbins = (uint32_t)n_fine; fa.nsub = SPH_NSUB;
        fa.perm = A.g_perm.as<uint32_t>(); fa.fkeys = A.g_fkeys.as<uint32_t>();
        fa.cell_start = A.g_cell_start.as<uint32_t>();
        fa.perm_base = (uint32_t)nb; // original index of ghost k is n_binned + k
        SPH_TRY(nnps_bin_sort_finish(c, A.g_keys.as<uint32_t>(), ng, fa, T));
        // The uniform-mass records rest on what the update saw of the REAL particles' masses; ghosts are other ranks'
        // particles.  Look at theirs now and then (masses are constants of the motion): a ghost with another mass switches
        // the array back to mass-carrying records for good (until its masses are pushed again).
        if (A.m_known && A.prop[SPH_M] && --A.g_mcheck < 0) {
            A.g_mcheck = 64;
            SPH_TRY(c->bigq.reserve((1 + BIN_QUEUE) * 4));
            uint32_t *flag = c->bigq.as<uint32_t>(); // (the queue counter: idle between the sorts)
            hipLaunchKernelGGL(k_reset_u32, dim3(1), dim3(1), 0, c->stream, flag);
            hipLaunchKernelGGL(k_mass_differs, dim3(div_up(ng, 256)), dim3(256), 0, c->stream, A.prop[SPH_M] + nb, ng, A.m_value, flag);
            uint32_t *pin = (uint32_t *)c->pinned;
            HIP_TRY(hipMemcpyAsync(pin, flag, 4, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            if (pin[0]) { A.m_known = false; A.m_mixed_ghosts = true; }
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
    if (!c->nnps_valid) { sph_set_error("sph_nnps_info: call sph_nnps_update first"); return SPH_ERR_STATE; }
    d8[0] = c->cell_size; d8[1] = c->hmin;
    for (int k = 0; k < 3; k++) { d8[2 + k] = c->xmin[k]; d8[5 + k] = c->xmax[k]; }
    i4[0] = c->nc[0]; i4[1] = c->nc[1]; i4[2] = c->nc[2];
    // n_cells as the reference reports it (dim-aware, linked_list_nnps.pyx:321-325)
    long ncells = c->nc[0];
    if (c->dim == 2) ncells = (long)c->nc[0] * c->nc[1];
    if (c->dim == 3) ncells = (long)c->nc[0] * c->nc[1] * c->nc[2];
    i4[3] = ncells;
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
    int cx = (int)floor((x - g.xmin[0]) / g.cell_size);
    int cy = (int)floor((y - g.xmin[1]) / g.cell_size);
    int cz = (int)floor((z - g.xmin[2]) / g.cell_size);
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
    g.cell_size = c->cell_size;
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
    g.cell_size = c->cell_size;
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
__global__ __launch_bounds__(256) void k_gather_f64(const double *__restrict__ src, const uint32_t *__restrict__ perm,
                                                    size_t n, double *__restrict__ dst)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[perm[i]];
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
        HIP_TRY(hipMemsetAsync(tmp, 0, A.cap * sizeof(double), c->stream));
        for (int p = 0; p < SPH_PROP_COUNT; p++) {
            if (!A.prop[p]) continue;
            hipLaunchKernelGGL(k_gather_f64, dim3(div_up(n_new, 256)), dim3(256), 0, c->stream, A.prop[p],
                               c->aux.as<uint32_t>(), n_new, tmp);
            double *old = A.prop[p];
            A.prop[p] = tmp;
            tmp = old;
        }
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipFree(tmp));
    }
    A.perm_n = 0;
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
    HIP_TRY(hipMemsetAsync(tmp, 0, A.cap * sizeof(double), c->stream));
    for (int p = 0; p < SPH_PROP_COUNT; p++) {
        if (!A.prop[p]) continue;
        hipLaunchKernelGGL(k_gather_f64, dim3(div_up(A.n, 256)), dim3(256), 0, c->stream, A.prop[p], A.perm.as<uint32_t>(),
                           A.n, tmp);
        double *old = A.prop[p];
        A.prop[p] = tmp;
        tmp = old;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipFree(tmp));
    A.perm_n = 0;
    c->nnps_valid = false;
    return SPH_OK;
}
