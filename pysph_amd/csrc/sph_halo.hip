// sph_halo.hip -- ghost-particle selection / packing for slab-decomposed runs.
//
// Reference behaviour replaced: ParallelManager.compute_remote_particles
// (pysph/parallel/parallel_manager.pyx:1159-1243: which local particles must
// be copied to which neighbour) and the per-property packing of
// remote_exchange_data (:159-210).  The reference ships every load-balancing
// property with one Zoltan Comm_Do per property; here one flat buffer per
// neighbour is packed on the device and moved by a single RCCL send/recv.
#include "sph_internal.h"


// one 64-bit flag per particle: bit 0 = selected for the low side, bit 32 = for the high side (kept for
// sph_halo_remove_selected), plus, per 256-particle block, how many particles go to each side (lo | hi << 32): the lists
// are then built from a scan of one counter per BLOCK (k_halo_lists) instead of one flag per particle.
__global__ __launch_bounds__(256) void k_halo_flags_counts(const double *__restrict__ coord, size_t n, int mode, double p0,
                                                           double p1, double p2, unsigned long long *__restrict__ fl,
                                                           unsigned long long *__restrict__ blk)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    bool lo = false, hi = false;
    if (i < n) {
        const double v = coord[i];
        if (mode == 0) { lo = v < p0; hi = v >= p1; }
        else { // nnps_base.pyx:805-817; a parked padding row (sph_halo_append_padded) is nobody's image
            const bool live = fabs(v) < SPH_PARKED_MIN;
            lo = live && (v - p0) <= p2;
            hi = live && (p1 - v) <= p2;
        }
        fl[i] = (lo ? 1ull : 0ull) | (hi ? 1ull << 32 : 0ull);
    }
    const unsigned long long ml = __ballot(lo), mh = __ballot(hi);
    __shared__ uint32_t cl[4], ch[4];
    if ((threadIdx.x & 63) == 0) { cl[threadIdx.x >> 6] = (uint32_t)__popcll(ml); ch[threadIdx.x >> 6] = (uint32_t)__popcll(mh); }
    __syncthreads();
    if (threadIdx.x == 0)
        blk[blockIdx.x] = (unsigned long long)(cl[0] + cl[1] + cl[2] + cl[3]) | ((unsigned long long)(ch[0] + ch[1] + ch[2] + ch[3]) << 32);
}

// index lists of both sides, ascending (the order a scan of per-particle flags gives): rank inside the
// block by wavefront ballots, block offset from the scanned block counters
__global__ __launch_bounds__(256) void k_halo_lists(const unsigned long long *__restrict__ fl, size_t n,
                                                    const unsigned long long *__restrict__ blkpos,
                                                    uint32_t *__restrict__ list_lo, uint32_t *__restrict__ list_hi)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned long long f = i < n ? fl[i] : 0ull;
    const bool on[2] = {(f & 1ull) != 0, (f >> 32) != 0};
    __shared__ uint32_t wcnt[2][4];
    unsigned long long m[2];
#pragma unroll
    for (int s = 0; s < 2; s++) {
        m[s] = __ballot(on[s]);
        if (lane == 0) wcnt[s][wv] = (uint32_t)__popcll(m[s]);
    }
    __syncthreads();
    const unsigned long long base = blkpos[blockIdx.x];
#pragma unroll
    for (int s = 0; s < 2; s++) {
        uint32_t before = 0;
        for (int w = 0; w < wv; w++) before += wcnt[s][w];
        const uint32_t pl = (uint32_t)((s == 0) ? (base & 0xffffffffull) : (base >> 32)) + before +
                            (uint32_t)__popcll(m[s] & ((1ull << lane) - 1ull));
        if (on[s]) (s == 0 ? list_lo : list_hi)[pl] = (uint32_t)i;
    }
}

__global__ __launch_bounds__(256) void k_box_wrap(double *__restrict__ coord, size_t n, double vmin, double vmax,
                                                  double translate)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = coord[i];
    if (v < vmin) v = v + translate;
    if (v > vmax) v = v - translate;
    coord[i] = v;
}

__global__ __launch_bounds__(256) void k_list_scatter(const uint32_t *__restrict__ flag, const uint32_t *__restrict__ pos,
                                                      size_t n, uint32_t *__restrict__ list)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (flag[i]) list[pos[i]] = (uint32_t)i;
}

// Every property of one pack/append call in ONE launch (blockIdx.y = property):
// a periodic domain update is ~200 property gathers, launch-bound otherwise.
struct PropList {
    double *p[SPH_PROP_COUNT];
    int what[SPH_PROP_COUNT]; // 0 copy, 1 add `val`, 2 mirror about `val`, 3 negate
};

__global__ __launch_bounds__(256) void k_halo_gather_multi(PropList L, const uint32_t *__restrict__ list, size_t count,
                                                           double val, double *__restrict__ dst)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const int k = blockIdx.y, what = L.what[k];
    const double v = L.p[k][list[i]];
    // mirror: x + 2*(plane - x) as nnps_base.pyx:568-575,598-611 computes it
    dst[(size_t)k * count + i] = what == 1 ? v + val : (what == 2 ? v + 2.0 * (val - v) : (what == 3 ? -v : v));
}

__global__ __launch_bounds__(256) void k_halo_append_multi(PropList L, const double *__restrict__ src, size_t n0,
                                                           size_t count, size_t stride)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const int k = blockIdx.y;
    L.p[k][n0 + i] = src[(size_t)k * stride + i];
}


extern "C" int sph_halo_select(sph_ctx *c, int id, int axis, int mode, double p0, double p1, double p2, size_t upto,
                               size_t *counts)
{
    if (!c || id < 0 || id >= SPH_MAX_ARRAYS || axis < 0 || axis > 2 || !counts) {
        sph_set_error("sph_halo_select: bad arguments");
        return SPH_ERR_ARG;
    }
    HIP_TRY(hipSetDevice(c->device));
    DevArray &A = c->arr[id];
    HaloState &H = c->halo[id];
    size_t n = upto ? (upto < A.n ? upto : A.n) : A.n_real;
    counts[0] = counts[1] = 0;
    H.count[0] = H.count[1] = 0;
    H.nsel = n;
    if (n == 0) return SPH_OK;
    const double *coord = A.prop[SPH_X + axis];
    if (!coord) { sph_set_error("sph_halo_select: no device coordinates"); return SPH_ERR_MISSING_PROP; }
    const unsigned nb = div_up(n, 256);
    SPH_TRY(H.flag[0].reserve((n + 1) * 8));
    SPH_TRY(H.flag[1].reserve(((size_t)nb + 1) * 8));
    SPH_TRY(H.pos[1].reserve(((size_t)nb + 1) * 8));
    unsigned long long *fl = H.flag[0].as<unsigned long long>();
    unsigned long long *blk = H.flag[1].as<unsigned long long>(), *bps = H.pos[1].as<unsigned long long>();
    hipLaunchKernelGGL(k_halo_flags_counts, dim3(nb), dim3(256), 0, c->stream, coord, n, mode, p0, p1, p2, fl, blk);
    SPH_TRY(dev_scan_u64(c, blk, bps, nb, true));
    unsigned long long *pin = (unsigned long long *)c->pinned;
    HIP_TRY(hipMemcpyAsync(pin, bps + (nb - 1), 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(pin + 1, blk + (nb - 1), 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const unsigned long long tot = pin[0] + pin[1]; // no carry between the words: each count < 2^32
    H.count[0] = (size_t)(tot & 0xffffffffull);
    H.count[1] = (size_t)(tot >> 32);
    for (int s = 0; s < 2; s++) {
        counts[s] = H.count[s];
        SPH_TRY(H.list[s].reserve((H.count[s] + 1) * 4));
    }
    if (H.count[0] + H.count[1])
        hipLaunchKernelGGL(k_halo_lists, dim3(nb), dim3(256), 0, c->stream, fl, n, bps, H.list[0].as<uint32_t>(),
                           H.list[1].as<uint32_t>());
    return SPH_OK;
}

// Selection and packing of both faces WITHOUT the host: thread i owns particle i, its list positions come
// from the scan, its rows go straight into the two fixed-capacity messages [nprops][cap] (+ one header
// double behind them: the row count, negated when it exceeds the capacity -- the payload is then incomplete
// and the pair repeats that face with the exact size).  The thread of the last particle writes the headers.
struct DirectPack {
    const double *p[32];
    int nprops, axis_k; // index of the slab-axis coordinate among the packed properties, or -1
    double shift[2];
    size_t cap[2];
    double *dst[2];
};
// promised-uniform h / m that do NOT travel (sph_halo_select_pack_promised): every selected row is checked against the
// promise on the SENDER's side, in pass 1; a chunk with a row that breaks it sets bit 31 (low face) / bit 63 (high
// face) of its counter word, the workgroup that writes the headers adds 0.5 to the header of such a face
struct PromiseCheck {
    const double *h, *m; // null: no promise for that property
    double h_promise, m_promise;
};
#define HALO_CNT_MASK 0x7fffffffull

// pass 1: how many particles of each CHUNK (q256 x 256 consecutive particles, one workgroup) go to the low / high face
// (lo | hi << 32).  The chunk grows with n so that there are at most HALO_MAX_CHUNKS of them: pass 2 then finds a
// chunk's first list position by summing the counters before it (<= 16 MB of L2 reads over the whole launch) and no
// scan launches stand between the two passes.
#define HALO_MAX_CHUNKS 2048
// which face(s) a particle at coordinate v belongs to.  BOX = false (slab faces): lo: v < p0, hi: v >= p1.  BOX = true
// (periodic box, nnps_base.pyx:805-817): lo: (v - p0) <= p2, hi: (p1 - v) <= p2 -- the rule of k_halo_flags_counts; a
// parked padding row is nobody's image
template <bool BOX> __device__ __forceinline__ void face_rule(double v, double p0, double p1, double p2, bool &lo, bool &hi)
{
    if (!BOX) { lo = v < p0; hi = v >= p1; }
    else {
        const bool live = fabs(v) < SPH_PARKED_MIN;
        lo = live && (v - p0) <= p2;
        hi = live && (p1 - v) <= p2;
    }
}

template <bool BOX>
__global__ __launch_bounds__(256) void k_halo_chunk_counts(const double *__restrict__ coord, size_t n, double p0, double p1, double p2,
                                                           int q256, unsigned long long *__restrict__ blk, PromiseCheck pc = PromiseCheck{})
{
    const size_t first = (size_t)blockIdx.x * 256 * q256;
    uint32_t cl = 0, ch = 0; // (wave-uniform)
    bool bl = false, bh = false; // a selected row of this wavefront breaks the promise (wave-uniform)
    for (int q = 0; q < q256; q++) {
        const size_t i = first + (size_t)q * 256 + threadIdx.x;
        bool lo = false, hi = false;
        if (i < n) face_rule<BOX>(coord[i], p0, p1, p2, lo, hi);
        cl += (uint32_t)__popcll(__ballot(lo));
        ch += (uint32_t)__popcll(__ballot(hi));
        if (pc.h || pc.m) {
            bool bad = false;
            if (lo || hi) bad = (pc.h && pc.h[i] != pc.h_promise) || (pc.m && pc.m[i] != pc.m_promise);
            bl = bl || __any(bad && lo);
            bh = bh || __any(bad && hi);
        }
    }
    __shared__ uint32_t sl[4], sh[4];
    if ((threadIdx.x & 63) == 0) { sl[threadIdx.x >> 6] = cl | (bl ? 0x80000000u : 0u); sh[threadIdx.x >> 6] = ch | (bh ? 0x80000000u : 0u); }
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t tl = (sl[0] & 0x7fffffffu) + (sl[1] & 0x7fffffffu) + (sl[2] & 0x7fffffffu) + (sl[3] & 0x7fffffffu);
        const uint32_t th = (sh[0] & 0x7fffffffu) + (sh[1] & 0x7fffffffu) + (sh[2] & 0x7fffffffu) + (sh[3] & 0x7fffffffu);
        const uint32_t xl = (sl[0] | sl[1] | sl[2] | sl[3]) & 0x80000000u, xh = (sh[0] | sh[1] | sh[2] | sh[3]) & 0x80000000u;
        blk[blockIdx.x] = (unsigned long long)(tl | xl) | ((unsigned long long)(th | xh) << 32);
    }
}

// pass 2: every workgroup sums the counters of the chunks before its own, re-derives its flags, ranks its particles
// with wavefront ballots -- ascending index, the same order a full scan gives -- and writes their rows straight into
// the messages; the last workgroup writes the two headers.
__global__ __launch_bounds__(256) void k_halo_pack_direct(DirectPack a, const double *__restrict__ coord, size_t n, double p0,
                                                          double p1, int q256, const unsigned long long *__restrict__ blk)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __shared__ uint32_t sbase[2][4];
    __shared__ uint32_t wcnt[2][4];
    {
        // (bit 31 of each half: "a selected row of the chunk breaks the h / m promise" -- ORed, not summed; the workgroup of
        // the last chunk, which writes the headers, looks at its own chunk's word as well)
        uint32_t bl = 0, bh = 0, xl = 0, xh = 0;
        const uint32_t upto_b = blockIdx.x + (blockIdx.x == gridDim.x - 1 ? 1u : 0u);
        for (uint32_t b = threadIdx.x; b < upto_b; b += 256) {
            const unsigned long long v = blk[b];
            if (b < blockIdx.x) { bl += (uint32_t)(v & HALO_CNT_MASK); bh += (uint32_t)((v >> 32) & HALO_CNT_MASK); }
            xl |= (uint32_t)(v >> 31) & 1u; xh |= (uint32_t)(v >> 63) & 1u;
        }
        for (int o = 32; o > 0; o >>= 1) { bl += __shfl_xor(bl, o, 64); bh += __shfl_xor(bh, o, 64); xl |= __shfl_xor(xl, o, 64); xh |= __shfl_xor(xh, o, 64); }
        if (lane == 0) { sbase[0][wv] = bl | (xl << 31); sbase[1][wv] = bh | (xh << 31); }
    }
    __syncthreads();
    size_t run[2]; // list position of this chunk's next selected particle, per face (uniform over the workgroup)
    bool broken[2];
    for (int s = 0; s < 2; s++) {
        run[s] = (size_t)(sbase[s][0] & 0x7fffffffu) + (sbase[s][1] & 0x7fffffffu) + (sbase[s][2] & 0x7fffffffu) + (sbase[s][3] & 0x7fffffffu);
        broken[s] = ((sbase[s][0] | sbase[s][1] | sbase[s][2] | sbase[s][3]) >> 31) != 0u;
    }
    const size_t first = (size_t)blockIdx.x * 256 * q256;
    // the coordinates of (up to) eight trips in flight together: a trip is then ballots and LDS, not a memory latency
    double cv[8];
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const size_t i = first + (size_t)q * 256 + threadIdx.x;
        cv[q] = (q < q256 && i < n) ? coord[i] : 0.0;
    }
    for (int q = 0; q < q256; q++) {
        const size_t i = first + (size_t)q * 256 + threadIdx.x;
        bool on[2] = {false, false};
        if (i < n) {
            double v = 0.0;
            if (q < 8) {
#pragma unroll
                for (int r = 0; r < 8; r++) v = (r == q) ? cv[r] : v;
            } else {
                v = coord[i];
            }
            on[0] = v < p0;
            on[1] = v >= p1;
        }
        unsigned long long m[2];
        __syncthreads(); // (the previous trip's wcnt has been read)
#pragma unroll
        for (int s = 0; s < 2; s++) {
            m[s] = __ballot(on[s]);
            if (lane == 0) wcnt[s][wv] = (uint32_t)__popcll(m[s]);
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 2; s++) {
            uint32_t before = 0;
            for (int w = 0; w < wv; w++) before += wcnt[s][w];
            const size_t pl = run[s] + before + (size_t)__popcll(m[s] & ((1ull << lane) - 1ull));
            if (on[s] && a.dst[s] && pl < a.cap[s]) {
                // the row's properties in batches of eight INDEPENDENT loads, then the stores (one property at a time was a
                // chain of dependent pointer load -> gather -> store latencies: 41 us for the faces of a 2.1 M-particle slab)
                for (int k0 = 0; k0 < a.nprops; k0 += 8) {
                    double v[8];
#pragma unroll
                    for (int q = 0; q < 8; q++) v[q] = k0 + q < a.nprops ? a.p[k0 + q][i] : 0.0;
#pragma unroll
                    for (int q = 0; q < 8; q++)
                        if (k0 + q < a.nprops) a.dst[s][(size_t)(k0 + q) * a.cap[s] + pl] = (k0 + q == a.axis_k) ? v[q] + a.shift[s] : v[q];
                }
            }
            run[s] += wcnt[s][0] + wcnt[s][1] + wcnt[s][2] + wcnt[s][3];
        }
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0)
        for (int s = 0; s < 2; s++)
            if (a.dst[s]) {
                const double cnt = (double)run[s] + (broken[s] ? 0.5 : 0.0); // + 0.5: a selected row breaks the h / m promise
                a.dst[s][(size_t)a.nprops * a.cap[s]] = run[s] <= a.cap[s] ? cnt : -cnt;
            }
}

extern "C" int sph_halo_select_pack(sph_ctx *c, int id, int axis, double lo_cut, double hi_cut, size_t upto, int nprops,
                                    const int *props, const double *shift2, const size_t *cap2, void *const *dst2)
{
    const double nan = __builtin_nan("");
    return sph_halo_select_pack_promised(c, id, axis, lo_cut, hi_cut, upto, nprops, props, shift2, cap2, dst2, nan, nan);
}

// ... with the promise check of the round-trip-free exchange on the SENDER's side: h_promise / m_promise (NaN: none) are
// the ONE smoothing length / mass every particle of the array is promised to have -- then h / m need not be among the
// packed properties (the receiver writes the promised value into its ghost rows, sph_halo_append_padded), and every
// selected row is compared with the promise here: a row that breaks it adds 0.5 to its message's header.
extern "C" int sph_halo_select_pack_promised(sph_ctx *c, int id, int axis, double lo_cut, double hi_cut, size_t upto, int nprops,
                                             const int *props, const double *shift2, const size_t *cap2, void *const *dst2,
                                             double h_promise, double m_promise)
{
    if (!c || id < 0 || id >= SPH_MAX_ARRAYS || axis < 0 || axis > 2 || nprops < 1 || nprops > 32 || !props || !shift2 ||
        !cap2 || !dst2) {
        sph_set_error("sph_halo_select_pack: bad arguments (at most 32 properties)");
        return SPH_ERR_ARG;
    }
    HIP_TRY(hipSetDevice(c->device));
    DevArray &A = c->arr[id];
    HaloState &H = c->halo[id];
    const size_t n = upto ? (upto < A.n ? upto : A.n) : A.n_real;
    H.count[0] = H.count[1] = 0; // the host does not learn the counts: no list-based call may follow
    H.nsel = 0;
    DirectPack a;
    a.nprops = nprops;
    a.axis_k = -1;
    for (int k = 0; k < nprops; k++) {
        const int p = props[k];
        if (p < 0 || p >= SPH_PROP_COUNT || !A.prop[p]) {
            sph_set_error("sph_halo_select_pack: array %d has no device property %d", id, p);
            return SPH_ERR_MISSING_PROP;
        }
        a.p[k] = A.prop[p];
        if (p == SPH_X + axis) a.axis_k = k;
    }
    for (int s = 0; s < 2; s++) { a.shift[s] = shift2[s]; a.cap[s] = cap2[s]; a.dst[s] = (double *)dst2[s]; }
    if (n == 0) { // empty array: empty messages
        for (int s = 0; s < 2; s++)
            if (a.dst[s]) HIP_TRY(hipMemsetAsync(a.dst[s] + (size_t)nprops * a.cap[s], 0, sizeof(double), c->stream));
        return SPH_OK;
    }
    const double *coord = A.prop[SPH_X + axis];
    if (!coord) { sph_set_error("sph_halo_select_pack: no device coordinates"); return SPH_ERR_MISSING_PROP; }
    // two passes over the coordinate, TWO launches: one counter per chunk of the array, summed by the packing
    // workgroups themselves (round 4: a counter per 256 particles and a three-launch scan between the passes)
    const int q256 = (int)div_up(n, (size_t)256 * HALO_MAX_CHUNKS);
    const unsigned nb = div_up(n, (size_t)256 * q256);
    SPH_TRY(H.flag[1].reserve(((size_t)nb + 1) * 8));
    unsigned long long *blk = H.flag[1].as<unsigned long long>();
    PromiseCheck pc;
    pc.h = h_promise == h_promise ? A.prop[SPH_H] : nullptr;
    pc.m = m_promise == m_promise ? A.prop[SPH_M] : nullptr;
    pc.h_promise = h_promise; pc.m_promise = m_promise;
    if ((h_promise == h_promise && !pc.h) || (m_promise == m_promise && !pc.m)) {
        sph_set_error("sph_halo_select_pack_promised: array %d has no device h / m to check the promise against", id);
        return SPH_ERR_MISSING_PROP;
    }
    hipLaunchKernelGGL(k_halo_chunk_counts<false>, dim3(nb), dim3(256), 0, c->stream, coord, n, lo_cut, hi_cut, 0.0, q256, blk, pc);
    hipLaunchKernelGGL(k_halo_pack_direct, dim3(nb), dim3(256), 0, c->stream, a, coord, n, lo_cut, hi_cut, q256, blk);
    return SPH_OK;
}

// The headers of the ghost messages and the flag words of an exchange on their way to the host WITHOUT a round trip: one
// launch gathers n doubles (one from each device address) and nflags uint32 words (as doubles, behind them) straight into
// `host_out` -- page-locked host memory the device can write (torch's pinned tensors, hipHostMalloc) -- and the caller
// records an event behind it.  (torch.stack + cat + copy_ were four launches and 19 us per exchange.)
struct GatherPtrs { const double *p[48]; };
__global__ void k_gather_to_host(GatherPtrs g, int n, const uint32_t *__restrict__ flags, int nflags, double *__restrict__ out)
{
    const int t = threadIdx.x;
    if (t < n) out[t] = *g.p[t];
    else if (t < n + nflags) out[t] = (double)flags[t - n];
}
extern "C" int sph_queue_values(sph_ctx *c, int n, const void *const *dev_ptrs, int nflags, const void *flag_words, double *host_out)
{
    if (!c || n < 0 || n > 48 || nflags < 0 || n + nflags > 64 || (n && !dev_ptrs) || (nflags && !flag_words) || !host_out) {
        sph_set_error("sph_queue_values: bad arguments (at most 48 values + flag words up to 64 in all)");
        return SPH_ERR_ARG;
    }
    HIP_TRY(hipSetDevice(c->device));
    GatherPtrs g;
    for (int k = 0; k < 48; k++) g.p[k] = k < n ? (const double *)dev_ptrs[k] : nullptr;
    hipLaunchKernelGGL(k_gather_to_host, dim3(1), dim3(64), 0, c->stream, g, n, (const uint32_t *)flag_words, nflags, host_out);
    HIP_TRY(hipGetLastError());
    return SPH_OK;
}

// Small device -> host reads in one round trip (the headers of the ghost messages): n doubles, one from each
// device address, all copies queued on the context's stream before the single synchronisation.
extern "C" int sph_read_values(sph_ctx *c, int n, const void *const *dev_ptrs, double *out)
{
    if (!c || n < 0 || n > 64 || (n && (!dev_ptrs || !out))) { sph_set_error("sph_read_values: bad arguments (at most 64 values)"); return SPH_ERR_ARG; }
    HIP_TRY(hipSetDevice(c->device));
    for (int k = 0; k < n; k++)
        HIP_TRY(hipMemcpyAsync(c->pinned + k, dev_ptrs[k], sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int k = 0; k < n; k++) out[k] = c->pinned[k];
    return SPH_OK;
}

extern "C" int sph_halo_pack(sph_ctx *c, int id, int side, int nprops, const int *props, int axis, double shift,
                             void *dst)
{
    if (!c || id < 0 || id >= SPH_MAX_ARRAYS || side < 0 || side > 1 || nprops < 1 || nprops > SPH_PROP_COUNT) {
        sph_set_error("sph_halo_pack: bad arguments");
        return SPH_ERR_ARG;
    }
    HIP_TRY(hipSetDevice(c->device));
    DevArray &A = c->arr[id];
    HaloState &H = c->halo[id];
    size_t cnt = H.count[side];
    if (cnt == 0) return SPH_OK;
    PropList L;
    for (int k = 0; k < nprops; k++) {
        int p = props[k];
        if (p < 0 || p >= SPH_PROP_COUNT || !A.prop[p]) {
            sph_set_error("sph_halo_pack: array %d has no device property %d", id, p);
            return SPH_ERR_MISSING_PROP;
        }
        L.p[k] = A.prop[p];
        L.what[k] = (p == SPH_X + axis) ? 1 : 0;
    }
    hipLaunchKernelGGL(k_halo_gather_multi, dim3(div_up(cnt, 256), nprops), dim3(256), 0, c->stream, L,
                       H.list[side].as<uint32_t>(), cnt, shift, (double *)dst);
    return SPH_OK;
}

// Mirror images of the particles selected for `side` (CPUDomainManager.
// _create_ghosts_mirror, pysph/base/nnps_base.pyx:506-697): coordinate
// reflected about `plane`, normal velocity negated, everything else copied.
extern "C" int sph_halo_pack_mirror(sph_ctx *c, int id, int side, int nprops, const int *props, int axis, double plane,
                                    void *dst)
{
    if (!c || id < 0 || id >= SPH_MAX_ARRAYS || side < 0 || side > 1 || nprops < 1 || nprops > SPH_PROP_COUNT || axis < 0 || axis > 2) {
        sph_set_error("sph_halo_pack_mirror: bad arguments");
        return SPH_ERR_ARG;
    }
    HIP_TRY(hipSetDevice(c->device));
    DevArray &A = c->arr[id];
    HaloState &H = c->halo[id];
    size_t cnt = H.count[side];
    if (cnt == 0) return SPH_OK;
    PropList L;
    for (int k = 0; k < nprops; k++) {
        int p = props[k];
        if (p < 0 || p >= SPH_PROP_COUNT || !A.prop[p]) {
            sph_set_error("sph_halo_pack_mirror: array %d has no device property %d", id, p);
            return SPH_ERR_MISSING_PROP;
        }
        L.p[k] = A.prop[p];
        // the mirrored coordinate; the normal velocity component flips sign (_mul_to_array(.., -1))
        L.what[k] = p == SPH_X + axis ? 2 : (p == SPH_U + axis ? 3 : 0);
    }
    hipLaunchKernelGGL(k_halo_gather_multi, dim3(div_up(cnt, 256), nprops), dim3(256), 0, c->stream, L,
                       H.list[side].as<uint32_t>(), cnt, plane, (double *)dst);
    return SPH_OK;
}

extern "C" int sph_halo_append(sph_ctx *c, int id, int nprops, const int *props, const void *src, size_t count)
{
    return sph_halo_append_strided(c, id, nprops, props, src, count, count);
}

// rows of a fixed-capacity message: property k of row i at src[k * stride + i]
extern "C" int sph_halo_append_strided(sph_ctx *c, int id, int nprops, const int *props, const void *src, size_t count,
                                       size_t stride)
{
    if (!c || id < 0 || id >= SPH_MAX_ARRAYS || nprops < 1 || stride < count) { sph_set_error("sph_halo_append: bad arguments"); return SPH_ERR_ARG; }
    if (count == 0) return SPH_OK;
    DevArray &A = c->arr[id];
    size_t n0 = A.n;
    for (int k = 0; k < nprops; k++) SPH_TRY(sph_array_ensure_prop(c, id, props[k]));
    SPH_TRY(sph_array_resize(c, id, n0 + count, A.n_real));
    if (nprops > SPH_PROP_COUNT) { sph_set_error("sph_halo_append: too many properties"); return SPH_ERR_ARG; }
    PropList L;
    for (int k = 0; k < nprops; k++) { L.p[k] = A.prop[props[k]]; L.what[k] = 0; }
    hipLaunchKernelGGL(k_halo_append_multi, dim3(div_up(count, 256), nprops), dim3(256), 0, c->stream, L,
                       (const double *)src, n0, count, stride);
    c->nnps_valid = false;
    return SPH_OK;
}

// Migrants under a promise (round 6): the rows of sph_halo_append whose h / m are promised to be the array's ONE value
// (the ranks agreed on it: SlabDecomposition's promises).  What the neighbour update knows of h and m then survives
// the append -- the update after a migration stays without a device->host round trip, as the one after a padded ghost
// append does -- and every arriving row is checked on the device: a row with another h or m sets bit 1 of the flag
// word the padded exchange reads with its headers (sticky; the caller raises).
__global__ __launch_bounds__(256) void k_promise_check(const double *__restrict__ hcol, const double *__restrict__ mcol, size_t count,
                                                       double h_promise, double m_promise, uint32_t *__restrict__ flag)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const bool bad = (hcol && hcol[i] != h_promise) || (mcol && mcol[i] != m_promise);
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 2u);
}

extern "C" int sph_halo_append_promised(sph_ctx *c, int id, int nprops, const int *props, const void *src, size_t count,
                                        double h_promise, double m_promise, void *flag_word)
{
    if (!c || id < 0 || id >= SPH_MAX_ARRAYS || nprops < 1 || !props || !flag_word) { sph_set_error("sph_halo_append_promised: bad arguments"); return SPH_ERR_ARG; }
    if (count == 0) return SPH_OK;
    HIP_TRY(hipSetDevice(c->device));
    int k_h = -1, k_m = -1;
    for (int k = 0; k < nprops; k++) {
        if (props[k] == SPH_H) k_h = k;
        if (props[k] == SPH_M) k_m = k;
    }
    const DevArray &A0 = c->arr[id];
    const bool empty0 = A0.n == 0;
    // (a promised property must TRAVEL here: these are real particles, nothing is written in their place)
    const bool keep_h = k_h >= 0 && h_promise == h_promise && !A0.raw_hm &&
                        (empty0 || (!A0.h_dirty && A0.h_seen && A0.h_lo == h_promise && A0.h_hi == h_promise));
    const bool keep_m = k_m >= 0 && m_promise == m_promise && !A0.raw_hm &&
                        (empty0 || (!A0.m_dirty && A0.m_seen && A0.m_lo == m_promise && A0.m_hi == m_promise));
    const bool mk = A0.m_known || empty0;
    SPH_TRY(sph_halo_append_strided(c, id, nprops, props, src, count, count));
    DevArray &A = c->arr[id];
    if (keep_h) { A.h_dirty = false; if (empty0) { A.h_seen = true; A.h_lo = A.h_hi = h_promise; } }
    if (keep_m) {
        A.m_dirty = false; A.m_known = mk;
        if (empty0) { A.m_seen = true; A.m_lo = A.m_hi = m_promise; A.m_value = m_promise; }
    }
    if (keep_h || keep_m) {
        const double *base = (const double *)src;
        hipLaunchKernelGGL(k_promise_check, dim3(div_up(count, 256)), dim3(256), 0, c->stream, keep_h ? base + (size_t)k_h * count : nullptr,
                           keep_m ? base + (size_t)k_m * count : nullptr, count, h_promise, m_promise, (uint32_t *)flag_word);
    }
    return SPH_OK;
}

// Rows of a fixed-capacity message appended WITHOUT the host knowing how many there are: all `cap` rows of the message
// go behind the particles; the first |header| of them are the ghosts, the rest are PADDING ROWS: parked at
// x = y = z = SPH_PARKED (1e18, far outside any domain), every other listed property zero.  A parked row is inert on the
// whole path and FINITE everywhere (the pair kernels multiply failed candidates by zero, a NaN would poison the sums):
// the bounds reduction and the keys of sph_nnps_update skip / spread it (|x| >= SPH_PARKED_MIN), the fp32 prefilter and
// every exact distance test fail against it by 1e36, and ghosts are never destinations.  flag (device word, sticky):
// bit 0 = the message was incomplete (negative header: more rows than its capacity), bit 1 = a ghost's h or m differs
// from the promised one.
__global__ __launch_bounds__(256) void k_halo_append_padded(PropList L, const double *__restrict__ src0, size_t n0, size_t cap0,
                                                            int nprops, int k_h, int k_m, double h_promise, double m_promise,
                                                            uint32_t *__restrict__ flag, double *__restrict__ fill_h, double *__restrict__ fill_m,
                                                            const double *__restrict__ src1 = nullptr, size_t cap1 = 0)
{
    // rows [n0, n0 + cap0) take message 0, rows [n0 + cap0, n0 + cap0 + cap1) message 1 (the other face's: ONE launch)
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cap0 + cap1) return;
    const int k = blockIdx.y;
    if (k >= nprops) {
        // a promised property that did not travel: every row (ghost or padding) gets the promised value -- the array's
        // range stays the ONE value whatever looks at it
        if (k == nprops && fill_h) fill_h[n0 + i] = h_promise;
        if (k == nprops + 1 && fill_m) fill_m[n0 + i] = m_promise;
        return;
    }
    const bool second = i >= cap0;
    const double *__restrict__ src = second ? src1 : src0;
    const size_t cap = second ? cap1 : cap0, r = second ? i - cap0 : i;
    const double hdr = src[(size_t)nprops * cap];
    const size_t count = (size_t)fmin(fabs(hdr), (double)cap);
    if (r == 0 && k == 0 && hdr < 0.0) atomicOr(flag, 1u);
    double v = L.what[k] ? SPH_PARKED : 0.0;
    if (r < count) {
        v = src[(size_t)k * cap + r];
        if ((k == k_h && h_promise == h_promise && v != h_promise) || (k == k_m && m_promise == m_promise && v != m_promise)) atomicOr(flag, 2u);
    }
    L.p[k][n0 + i] = v;
}

extern "C" int sph_halo_append_padded(sph_ctx *c, int id, int nprops, const int *props, const void *src, size_t cap,
                                      double h_promise, double m_promise, void *flag_word)
{
    return sph_halo_append_padded2(c, id, nprops, props, src, cap, nullptr, 0, h_promise, m_promise, flag_word);
}

// ... the messages of BOTH faces of an array in one launch: the rows of `src` first, those of `src2` (may be NULL / 0 rows)
// behind them -- the order two calls give
extern "C" int sph_halo_append_padded2(sph_ctx *c, int id, int nprops, const int *props, const void *src, size_t cap,
                                       const void *src2, size_t cap2, double h_promise, double m_promise, void *flag_word)
{
    if (!c || id < 0 || id >= SPH_MAX_ARRAYS || nprops < 1 || nprops > SPH_PROP_COUNT || !props || (cap && !src) || (cap2 && !src2) || !flag_word) {
        sph_set_error("sph_halo_append_padded: bad arguments");
        return SPH_ERR_ARG;
    }
    if (cap == 0 && cap2 != 0) { src = src2; cap = cap2; src2 = nullptr; cap2 = 0; }
    if (cap == 0) return SPH_OK;
    HIP_TRY(hipSetDevice(c->device));
    size_t n0 = c->arr[id].n;
    int k_h = -1, k_m = -1;
    for (int k = 0; k < nprops; k++) {
        SPH_TRY(sph_array_ensure_prop(c, id, props[k]));
        if (props[k] == SPH_H) k_h = k;
        if (props[k] == SPH_M) k_m = k;
    }
    // what the neighbour update knows of h and m stays valid when every ghost is promised to carry the array's ONE value
    // (checked on the device, bit 1 of the flag word): the round-trip-free update goes on
    DevArray &A0 = c->arr[id];
    // (a promised property that is not in the message did not travel: it is written into the rows below)
    // An array that holds NOTHING when the first message arrives (a rank without obstacle particles, a slab the fluid has
    // not reached) has an empty range: whatever is promised IS its range from here on -- without this such an array was
    // "mass unknown" for ever (its padding rows are not particles), every neighbour update of the rank looked at h and m
    // (a round trip) and its dam-break evaluation left the one-launch merged path.
    const bool empty0 = A0.n == 0;
    const bool keep_h = h_promise == h_promise && !A0.raw_hm && (empty0 || (!A0.h_dirty && A0.h_seen && A0.h_lo == h_promise && A0.h_hi == h_promise));
    const bool keep_m = m_promise == m_promise && !A0.raw_hm && (empty0 || (!A0.m_dirty && A0.m_seen && A0.m_lo == m_promise && A0.m_hi == m_promise));
    const bool fill_h = k_h < 0 && h_promise == h_promise, fill_m = k_m < 0 && m_promise == m_promise;
    if (fill_h) SPH_TRY(sph_array_ensure_prop(c, id, SPH_H));
    if (fill_m) SPH_TRY(sph_array_ensure_prop(c, id, SPH_M));
    const bool mk = A0.m_known || empty0;
    SPH_TRY(sph_array_resize(c, id, n0 + cap + cap2, A0.n_real));
    DevArray &A = c->arr[id];
    if (keep_h) { A.h_dirty = false; if (empty0) { A.h_seen = true; A.h_lo = A.h_hi = h_promise; } }
    if (keep_m) {
        A.m_dirty = false; A.m_known = mk;
        if (empty0) { A.m_seen = true; A.m_lo = A.m_hi = m_promise; A.m_value = m_promise; }
    }
    PropList L;
    for (int k = 0; k < nprops; k++) { L.p[k] = A.prop[props[k]]; L.what[k] = props[k] == SPH_X || props[k] == SPH_Y || props[k] == SPH_Z; }
    A.has_padding = true;
    hipLaunchKernelGGL(k_halo_append_padded, dim3(div_up(cap + cap2, 256), nprops + 2), dim3(256), 0, c->stream, L, (const double *)src, n0, cap,
                       nprops, k_h, k_m, h_promise, m_promise, (uint32_t *)flag_word, fill_h ? A.prop[SPH_H] : nullptr,
                       fill_m ? A.prop[SPH_M] : nullptr, (const double *)src2, cap2);
    c->nnps_valid = false;
    return SPH_OK;
}

// ---------------------------------------------------------------------------
// Periodic images of one axis WITHOUT a device->host round trip (round 5).  The list-based update (sph_halo_select +
// 2 x sph_halo_image) reads the two counts back per axis because the host owns the array sizes; here the images go
// into FIXED capacities behind the particles present -- rows [n, n + cap0) for the low face's images, [n + cap0,
// n + cap0 + cap1) for the high face's -- in ascending particle index as before, the rows behind the counts PARKED
// (sph_halo_append_padded's inert padding: x = y = z = 1e18, everything else 0).  The counts stay on the device
// (negated when they exceed their capacity: images are then missing) and reach the host one update later
// (sph_domain_counts_queue / _collect), where the capacities follow them.
// ---------------------------------------------------------------------------
// pass 2: the two index lists (ascending particle index, clipped at their capacities) and the two counts
__global__ __launch_bounds__(256) void k_domain_lists(const double *__restrict__ coord, size_t n, double p0, double p1, double p2,
                                                      int q256, size_t cap0, size_t cap1, const unsigned long long *__restrict__ blk,
                                                      uint32_t *__restrict__ list_lo, uint32_t *__restrict__ list_hi,
                                                      double *__restrict__ counts)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __shared__ uint32_t sbase[2][4];
    __shared__ uint32_t wcnt[2][4];
    {
        uint32_t bl = 0, bh = 0;
        for (uint32_t b = threadIdx.x; b < blockIdx.x; b += 256) { const unsigned long long v = blk[b]; bl += (uint32_t)v; bh += (uint32_t)(v >> 32); }
        for (int o = 32; o > 0; o >>= 1) { bl += __shfl_xor(bl, o, 64); bh += __shfl_xor(bh, o, 64); }
        if (lane == 0) { sbase[0][wv] = bl; sbase[1][wv] = bh; }
    }
    __syncthreads();
    size_t run[2];
    for (int s = 0; s < 2; s++) run[s] = (size_t)sbase[s][0] + sbase[s][1] + sbase[s][2] + sbase[s][3];
    const size_t first = (size_t)blockIdx.x * 256 * q256;
    const size_t cap[2] = {cap0, cap1};
    for (int q = 0; q < q256; q++) {
        const size_t i = first + (size_t)q * 256 + threadIdx.x;
        bool on[2] = {false, false};
        if (i < n) face_rule<true>(coord[i], p0, p1, p2, on[0], on[1]);
        unsigned long long m[2];
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 2; s++) {
            m[s] = __ballot(on[s]);
            if (lane == 0) wcnt[s][wv] = (uint32_t)__popcll(m[s]);
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 2; s++) {
            uint32_t before = 0;
            for (int w = 0; w < wv; w++) before += wcnt[s][w];
            const size_t pl = run[s] + before + (size_t)__popcll(m[s] & ((1ull << lane) - 1ull));
            if (on[s] && pl < cap[s]) (s == 0 ? list_lo : list_hi)[pl] = (uint32_t)i;
            run[s] += wcnt[s][0] + wcnt[s][1] + wcnt[s][2] + wcnt[s][3];
        }
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0)
        for (int s = 0; s < 2; s++) counts[s] = run[s] <= cap[s] ? (double)run[s] : -(double)run[s];
}

// pass 3: every row of the two capacities, every property (blockIdx.y): an image of the listed particle -- the low face's
// images move up by the period, the high face's down (nnps_base.pyx:841-856) -- or, behind the count, a parked row
__global__ __launch_bounds__(256) void k_domain_images(PropList L, size_t n, size_t cap0, size_t cap1, double shift,
                                                       const uint32_t *__restrict__ list_lo, const uint32_t *__restrict__ list_hi,
                                                       const double *__restrict__ counts)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cap0 + cap1) return;
    const int s = i < cap0 ? 0 : 1;
    const size_t r = s == 0 ? i : i - cap0, cap = s == 0 ? cap0 : cap1;
    const size_t count = (size_t)fmin(fabs(counts[s]), (double)cap);
    const int k = blockIdx.y, what = L.what[k];
    double v;
    if (r < count) {
        v = L.p[k][(s == 0 ? list_lo : list_hi)[r]];
        if (what == 1) v += s == 0 ? shift : -shift;
    } else {
        v = what ? SPH_PARKED : 0.0;
    }
    L.p[k][n + i] = v;
}

extern "C" int sph_domain_images_padded(sph_ctx *c, int id, int axis, double lo, double hi, double width, double shift,
                                        const size_t *cap2, int nprops, const int *props)
{
    if (!c || id < 0 || id >= SPH_MAX_ARRAYS || axis < 0 || axis > 2 || !cap2 || nprops < 1 || nprops > SPH_PROP_COUNT || !props) {
        sph_set_error("sph_domain_images_padded: bad arguments");
        return SPH_ERR_ARG;
    }
    HIP_TRY(hipSetDevice(c->device));
    if (!c->dom_counts.ptr) {
        SPH_TRY(c->dom_counts.reserve(SPH_MAX_ARRAYS * 6 * sizeof(double)));
        HIP_TRY(hipMemsetAsync(c->dom_counts.ptr, 0, SPH_MAX_ARRAYS * 6 * sizeof(double), c->stream));
    }
    double *counts = c->dom_counts.as<double>() + (size_t)id * 6 + axis * 2;
    const size_t n = c->arr[id].n, cap0 = cap2[0], cap1 = cap2[1];
    if (n + cap0 + cap1 >= (1ull << 32)) { sph_set_error("sph_domain_images_padded: too many particles"); return SPH_ERR_ARG; }
    bool has_h = false, has_m = false, has_x = false;
    for (int k = 0; k < nprops; k++) {
        const int p = props[k];
        if (p < 0 || p >= SPH_PROP_COUNT || !c->arr[id].prop[p]) {
            sph_set_error("sph_domain_images_padded: array %d has no device property %d", id, p);
            return SPH_ERR_MISSING_PROP;
        }
        has_h |= p == SPH_H; has_m |= p == SPH_M; has_x |= p == SPH_X + axis;
    }
    bool all_pos = has_x;
    for (int q = 0; q < 3; q++) { bool in = false; for (int k = 0; k < nprops; k++) in |= props[k] == SPH_X + q; all_pos &= in; }
    if (!all_pos) { sph_set_error("sph_domain_images_padded: x, y and z must all be among the imaged properties"); return SPH_ERR_ARG; }
    if (n == 0 || cap0 + cap1 == 0) { // nothing to image (or no room): zero counts
        HIP_TRY(hipMemsetAsync(counts, 0, 2 * sizeof(double), c->stream));
        if (n == 0) return SPH_OK;
    }
    // images (and padding rows) are copies of the array's own particles: h and m keep their range and their cleanliness
    const bool hd = c->arr[id].h_dirty, md = c->arr[id].m_dirty, mk = c->arr[id].m_known;
    SPH_TRY(sph_array_resize(c, id, n + cap0 + cap1, c->arr[id].n_real)); // may move the property buffers
    DevArray &A = c->arr[id];
    if (has_h) A.h_dirty = hd;
    if (has_m) { A.m_dirty = md; A.m_known = mk; }
    A.has_padding = true;
    PropList L;
    for (int k = 0; k < nprops; k++) {
        const int p = props[k];
        L.p[k] = A.prop[p];
        // 1: the axis coordinate (shifted in an image), 2: another coordinate (both parked in a padding row), 0: the rest
        L.what[k] = p == SPH_X + axis ? 1 : ((p == SPH_X || p == SPH_Y || p == SPH_Z) ? 2 : 0);
    }
    HaloState &H = c->halo[id];
    H.count[0] = H.count[1] = 0; // the host does not learn the counts: no list-based call may follow
    H.nsel = 0;
    const double *coord = A.prop[SPH_X + axis];
    const int q256 = (int)div_up(n, (size_t)256 * HALO_MAX_CHUNKS);
    const unsigned nb = div_up(n, (size_t)256 * q256);
    SPH_TRY(H.flag[1].reserve(((size_t)nb + 1) * 8));
    unsigned long long *blk = H.flag[1].as<unsigned long long>();
    SPH_TRY(H.list[0].reserve((cap0 + 1) * 4));
    SPH_TRY(H.list[1].reserve((cap1 + 1) * 4));
    hipLaunchKernelGGL(k_halo_chunk_counts<true>, dim3(nb), dim3(256), 0, c->stream, coord, n, lo, hi, width, q256, blk);
    hipLaunchKernelGGL(k_domain_lists, dim3(nb), dim3(256), 0, c->stream, coord, n, lo, hi, width, q256, cap0, cap1, blk,
                       H.list[0].as<uint32_t>(), H.list[1].as<uint32_t>(), counts);
    if (cap0 + cap1)
        hipLaunchKernelGGL(k_domain_images, dim3(div_up(cap0 + cap1, 256), nprops), dim3(256), 0, c->stream, L, n, cap0, cap1, shift,
                           H.list[0].as<uint32_t>(), H.list[1].as<uint32_t>(), counts);
    c->nnps_valid = false;
    return SPH_OK;
}

// the image counts of every sph_domain_images_padded call so far on their way to pinned memory, behind an event nobody
// waits for now ...
extern "C" int sph_domain_counts_queue(sph_ctx *c)
{
    if (!c) { sph_set_error("sph_domain_counts_queue: NULL context"); return SPH_ERR_ARG; }
    if (!c->dom_counts.ptr) return SPH_OK;
    HIP_TRY(hipSetDevice(c->device));
    if (!c->dom_pin) HIP_TRY(hipHostMalloc((void **)&c->dom_pin, SPH_MAX_ARRAYS * 6 * sizeof(double), hipHostMallocDefault));
    if (!c->dom_ev) HIP_TRY(hipEventCreateWithFlags(&c->dom_ev, hipEventDisableTiming));
    HIP_TRY(hipMemcpyAsync(c->dom_pin, c->dom_counts.ptr, SPH_MAX_ARRAYS * 6 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipEventRecord(c->dom_ev, c->stream));
    c->dom_queued = true;
    return SPH_OK;
}

// ... and read an update later: out[array * 6 + axis * 2 + side], negative = that face's images did not fit
extern "C" int sph_domain_counts_collect(sph_ctx *c, double *out)
{
    if (!c || !out) { sph_set_error("sph_domain_counts_collect: bad arguments"); return SPH_ERR_ARG; }
    if (!c->dom_queued) { sph_set_error("sph_domain_counts_collect: nothing was queued"); return SPH_ERR_STATE; }
    HIP_TRY(hipEventSynchronize(c->dom_ev));
    for (int k = 0; k < SPH_MAX_ARRAYS * 6; k++) out[k] = c->dom_pin[k];
    c->dom_queued = false;
    return SPH_OK;
}

// the h range of an array when it is known WITHOUT looking (seen by a neighbour update or a reduction and not written
// since): 1 and the range, else 0
extern "C" int sph_array_h_known(sph_ctx *c, int id, double *hmin, double *hmax)
{
    if (!c || id < 0 || id >= SPH_MAX_ARRAYS || !hmin || !hmax) return 0;
    const DevArray &A = c->arr[id];
    if (!A.used || !A.h_seen || A.h_dirty || A.raw_hm) return 0; // (a raw pointer to h was handed out: writes are not tracked)
    *hmin = A.h_lo; *hmax = A.h_hi;
    return 1;
}

__global__ __launch_bounds__(256) void k_halo_image_multi(PropList L, const uint32_t *__restrict__ list, size_t count,
                                                          double val, size_t n0)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const int k = blockIdx.y, what = L.what[k];
    const double v = L.p[k][list[i]];
    // same arithmetic as k_halo_gather_multi (the image position must be bit-identical to the packed one)
    L.p[k][n0 + i] = what == 1 ? v + val : (what == 2 ? v + 2.0 * (val - v) : (what == 3 ? -v : v));
}

extern "C" int sph_halo_image(sph_ctx *c, int id, int side, int nprops, const int *props, int axis, int mode, double val,
                              size_t *count)
{
    if (!c || id < 0 || id >= SPH_MAX_ARRAYS || side < 0 || side > 1 || nprops < 1 || nprops > SPH_PROP_COUNT ||
        axis < 0 || axis > 2 || mode < 0 || mode > 1) {
        sph_set_error("sph_halo_image: bad arguments");
        return SPH_ERR_ARG;
    }
    HIP_TRY(hipSetDevice(c->device));
    HaloState &H = c->halo[id];
    const size_t cnt = H.count[side];
    if (count) *count = cnt;
    if (cnt == 0) return SPH_OK;
    {
        DevArray &A0 = c->arr[id];
        if (H.nsel > A0.n) { sph_set_error("sph_halo_image: selection is stale"); return SPH_ERR_STATE; }
        for (int k = 0; k < nprops; k++) {
            int p = props[k];
            if (p < 0 || p >= SPH_PROP_COUNT || !A0.prop[p]) {
                sph_set_error("sph_halo_image: array %d has no device property %d", id, p);
                return SPH_ERR_MISSING_PROP;
            }
        }
    }
    const size_t n0 = c->arr[id].n;
    // images are copies of the array's own particles: a property among `props` keeps its range (and its cleanliness)
    bool has_h = false, has_m = false;
    for (int k = 0; k < nprops; k++) { has_h |= props[k] == SPH_H; has_m |= props[k] == SPH_M; }
    const bool hd = c->arr[id].h_dirty, md = c->arr[id].m_dirty, mk = c->arr[id].m_known;
    SPH_TRY(sph_array_resize(c, id, n0 + cnt, c->arr[id].n_real)); // may move the property buffers
    DevArray &A = c->arr[id];
    if (has_h && n0 > 0) A.h_dirty = hd;
    if (has_m && n0 > 0) { A.m_dirty = md; A.m_known = mk; }
    PropList L;
    for (int k = 0; k < nprops; k++) {
        const int p = props[k];
        L.p[k] = A.prop[p];
        L.what[k] = mode == 0 ? (p == SPH_X + axis ? 1 : 0) : (p == SPH_X + axis ? 2 : (p == SPH_U + axis ? 3 : 0));
    }
    hipLaunchKernelGGL(k_halo_image_multi, dim3(div_up(cnt, 256), nprops), dim3(256), 0, c->stream, L,
                       H.list[side].as<uint32_t>(), cnt, val, n0);
    c->nnps_valid = false;
    return SPH_OK;
}

__global__ __launch_bounds__(256) void k_keep_flags(const unsigned long long *__restrict__ fl, size_t nsel, size_t n,
                                                    uint32_t *__restrict__ keep)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    keep[i] = (i < nsel && fl[i]) ? 0u : 1u;
}

__global__ __launch_bounds__(256) void k_compact_f64(const double *__restrict__ src, const uint32_t *__restrict__ list,
                                                     size_t count, double *__restrict__ dst)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    dst[i] = src[list[i]];
}

// Few leavers among many particles: the kept rows of the tail [keepn, n) move into the holes the leavers left below keepn
// (what the reference's own removal does: ParticleArray.remove_particles hands the indices to the property arrays'
// `remove`, which copies elements from the end into the removed slots -- particle_array.pyx remove_particles).  Work and
// launches no longer grow with the particles that STAY: four small launches instead of a full scan and one compaction
// launch per property (~40 launches, 0.17 ms at 1 M particles with 35 properties).
__global__ __launch_bounds__(256) void k_tail_keep(const unsigned long long *__restrict__ fl, size_t nsel, size_t keepn, size_t gone,
                                                   uint32_t *__restrict__ keep)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > gone) return;
    const size_t r = keepn + i;
    keep[i] = (i == gone || (r < nsel && fl[r])) ? 0u : 1u; // (one entry more: the scan's total)
}

__global__ __launch_bounds__(256) void k_tail_list(const uint32_t *__restrict__ keep, const uint32_t *__restrict__ pos, size_t gone,
                                                   uint32_t keepn, uint32_t *__restrict__ filler)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < gone && keep[i]) filler[pos[i]] = keepn + (uint32_t)i;
}

struct FillArgs {
    double *p[SPH_PROP_COUNT];
    int np;
};

__global__ __launch_bounds__(256) void k_fill_holes(FillArgs a, const uint32_t *__restrict__ list0, uint32_t c0,
                                                    const uint32_t *__restrict__ list1, uint32_t keepn,
                                                    const uint32_t *__restrict__ filler, const uint32_t *__restrict__ pos, uint32_t gone)
{
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= pos[gone]) return; // the kept rows of the tail = the holes below keepn
    // the holes: the entries below keepn of the two ascending selection lists, the low face's first
    uint32_t lo = 0, hi = c0;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (list0[mid] < keepn) lo = mid + 1; else hi = mid; }
    const uint32_t hole = j < lo ? list0[j] : list1[j - lo];
    const uint32_t src = filler[j];
    for (int p = blockIdx.y; p < a.np; p += gridDim.y) a.p[p][hole] = a.p[p][src];
}

// Particles that left this rank's slab (ParallelManager's "exported" particles,
// pysph/parallel/parallel_manager.pyx:1085-1157 remove_particles after the
// lb_exchange_data send): stable compaction of every device property.
extern "C" int sph_halo_remove_selected(sph_ctx *c, int id, size_t *n_left)
{
    if (!c || id < 0 || id >= SPH_MAX_ARRAYS) { sph_set_error("sph_halo_remove_selected: bad arguments"); return SPH_ERR_ARG; }
    HIP_TRY(hipSetDevice(c->device));
    DevArray &A = c->arr[id];
    HaloState &H = c->halo[id];
    if (A.n != A.n_real) { sph_set_error("sph_halo_remove_selected: drop ghost particles first (n=%zu, n_real=%zu)", A.n, A.n_real); return SPH_ERR_STATE; }
    if (H.nsel > A.n) { sph_set_error("sph_halo_remove_selected: selection is stale"); return SPH_ERR_STATE; }
    const size_t n = A.n, gone = H.count[0] + H.count[1];
    if (n_left) *n_left = n - gone;
    if (gone == 0 || n == 0) return SPH_OK;
    const size_t keepn = n - gone;
    if (c->fill_holes && gone * 4 < n && H.nsel <= n) {
        SPH_TRY(c->tmp_u32a.reserve((gone + 64) * 4));
        SPH_TRY(c->tmp_u32b.reserve((gone + 64) * 4));
        SPH_TRY(c->dkeys.reserve((gone + 64) * 4));
        uint32_t *keep = c->tmp_u32a.as<uint32_t>(), *pos = c->tmp_u32b.as<uint32_t>(), *filler = c->dkeys.as<uint32_t>();
        hipLaunchKernelGGL(k_tail_keep, dim3(div_up(gone + 1, 256)), dim3(256), 0, c->stream, H.flag[0].as<unsigned long long>(),
                           H.nsel, keepn, gone, keep);
        SPH_TRY(dev_scan_u32(c, keep, pos, gone + 1, true));
        hipLaunchKernelGGL(k_tail_list, dim3(div_up(gone, 256)), dim3(256), 0, c->stream, keep, pos, gone, (uint32_t)keepn, filler);
        FillArgs fa;
        fa.np = 0;
        for (int p = 0; p < SPH_PROP_COUNT; p++) if (A.prop[p]) fa.p[fa.np++] = A.prop[p];
        if (fa.np)
            hipLaunchKernelGGL(k_fill_holes, dim3(div_up(gone, 256), (unsigned)std::min(fa.np, 16)), dim3(256), 0, c->stream, fa,
                               H.list[0].as<uint32_t>(), (uint32_t)H.count[0], H.list[1].as<uint32_t>(), (uint32_t)keepn, filler, pos,
                               (uint32_t)gone);
        sph_mark_removed(A, keepn);
        A.n = A.n_real = keepn;
        H.count[0] = H.count[1] = 0;
        c->nnps_valid = false;
        return SPH_OK;
    }
    // keep flags -> positions -> list of kept indices (ascending: order is preserved).  Scratch of the context and a
    // spare property buffer of the array that stay allocated: round 5 allocated and freed four device buffers and drained
    // the stream in here, ~1 ms per migration at 1 M particles (tools/stepping_selfslab.py).
    SPH_TRY(c->tmp_u32a.reserve((n + 64) * 8));
    SPH_TRY(c->tmp_u32b.reserve((n + 64) * 8));
    SPH_TRY(c->dkeys.reserve((keepn + 64) * 4));
    uint32_t *keep = c->tmp_u32a.as<uint32_t>(), *pos = c->tmp_u32b.as<uint32_t>(), *list = c->dkeys.as<uint32_t>();
    hipLaunchKernelGGL(k_keep_flags, dim3(div_up(n, 256)), dim3(256), 0, c->stream,
                       H.flag[0].as<unsigned long long>(), H.nsel, n, keep);
    SPH_TRY(dev_scan_u32(c, keep, pos, n, true));
    if (keepn)
        hipLaunchKernelGGL(k_list_scatter, dim3(div_up(n, 256)), dim3(256), 0, c->stream, keep, pos, n, list);
    if (!A.spare || A.spare_cap != A.cap) {
        if (A.spare) { HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipFree(A.spare)); A.spare = nullptr; }
        HIP_TRY(hipMalloc((void **)&A.spare, A.cap * sizeof(double)));
        A.spare_cap = A.cap;
    }
    double *tmp = A.spare;
    for (int p = 0; p < SPH_PROP_COUNT; p++) {
        if (!A.prop[p]) continue;
        if (keepn)
            hipLaunchKernelGGL(k_compact_f64, dim3(div_up(keepn, 256)), dim3(256), 0, c->stream, A.prop[p], list, keepn, tmp);
        double *old = A.prop[p];
        A.prop[p] = tmp;
        tmp = old;
    }
    A.spare = tmp; // (the buffer the last property left behind: stream order keeps its readers ahead of the next writer)
    sph_mark_removed(A, keepn);
    A.n = A.n_real = keepn;
    H.count[0] = H.count[1] = 0;
    c->nnps_valid = false;
    return SPH_OK;
}

extern "C" int sph_domain_box_wrap(sph_ctx *c, int id, int axis, double vmin, double vmax, double translate)
{
    if (!c || id < 0 || id >= SPH_MAX_ARRAYS || axis < 0 || axis > 2) { sph_set_error("sph_domain_box_wrap: bad arguments"); return SPH_ERR_ARG; }
    HIP_TRY(hipSetDevice(c->device));
    DevArray &A = c->arr[id];
    if (A.n_real == 0) return SPH_OK;
    if (!A.prop[SPH_X + axis]) { sph_set_error("sph_domain_box_wrap: no device coordinates"); return SPH_ERR_MISSING_PROP; }
    hipLaunchKernelGGL(k_box_wrap, dim3(div_up(A.n_real, 256)), dim3(256), 0, c->stream, A.prop[SPH_X + axis], A.n_real,
                       vmin, vmax, translate);
    c->nnps_valid = false;
    return SPH_OK;
}

// Histogram of one coordinate of the REAL particles over [vmin, vmin + span) in nbins bins (the last bin holds the upper
// edge): the re-balancing of a slab decomposition needs the distribution of the particles along the slab axis, not the
// particles (SlabDecomposition.rebalance pulled every coordinate to the host before).
__global__ __launch_bounds__(256) void k_coord_histogram(const double *__restrict__ coord, size_t n, double vmin, double inv_w, int nbins,
                                                         uint32_t *__restrict__ hist)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = coord[i];
    if (!(fabs(v) < SPH_PARKED_MIN)) return;
    int b = (int)floor((v - vmin) * inv_w);
    b = min(max(b, 0), nbins - 1);
    atomicAdd(&hist[b], 1u);
}

extern "C" int sph_coord_histogram(sph_ctx *c, int id, int axis, double vmin, double span, int nbins, uint32_t *host_out)
{
    if (!c || id < 0 || id >= SPH_MAX_ARRAYS || axis < 0 || axis > 2 || nbins < 1 || nbins > (1 << 20) || !(span > 0.0) || !host_out) {
        sph_set_error("sph_coord_histogram: bad arguments");
        return SPH_ERR_ARG;
    }
    HIP_TRY(hipSetDevice(c->device));
    DevArray &A = c->arr[id];
    memset(host_out, 0, (size_t)nbins * 4);
    if (A.n_real == 0) return SPH_OK;
    if (!A.prop[SPH_X + axis]) { sph_set_error("sph_coord_histogram: no device coordinates"); return SPH_ERR_MISSING_PROP; }
    HaloState &H = c->halo[id];
    SPH_TRY(H.pos[0].reserve((size_t)nbins * 4));
    uint32_t *hist = H.pos[0].as<uint32_t>();
    HIP_TRY(hipMemsetAsync(hist, 0, (size_t)nbins * 4, c->stream));
    hipLaunchKernelGGL(k_coord_histogram, dim3(div_up(A.n_real, 256)), dim3(256), 0, c->stream, A.prop[SPH_X + axis], A.n_real, vmin,
                       (double)nbins / span, nbins, hist);
    HIP_TRY(hipMemcpyAsync(host_out, hist, (size_t)nbins * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SPH_OK;
}

extern "C" int sph_array_props(sph_ctx *c, int id, int *out, int cap, int *n)
{
    if (!c || id < 0 || id >= SPH_MAX_ARRAYS || !out || !n) { sph_set_error("sph_array_props: bad arguments"); return SPH_ERR_ARG; }
    int k = 0;
    for (int p = 0; p < SPH_PROP_COUNT; p++)
        if (c->arr[id].prop[p]) {
            if (k >= cap) { sph_set_error("sph_array_props: buffer of %d ints is too small (SPH_PROP_COUNT = %d)", cap, SPH_PROP_COUNT); return SPH_ERR_ARG; }
            out[k++] = p;
        }
    *n = k;
    return SPH_OK;
}
