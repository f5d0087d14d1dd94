// sph_internal.h -- private declarations shared by the libsphhip translation units.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/sphhip.h"

void sph_set_error(const char *fmt, ...);

#define HIP_TRY(expr)                                                              \
    do {                                                                           \
        hipError_t _e = (expr);                                                    \
        if (_e != hipSuccess) {                                                    \
            sph_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr,             \
                          hipGetErrorString(_e));                                  \
            return SPH_ERR_HIP;                                                    \
        }                                                                          \
    } while (0)

#define SPH_TRY(expr)                \
    do {                             \
        int _rc = (expr);            \
        if (_rc != SPH_OK) return _rc; \
    } while (0)

// A growable device buffer.
struct DevBuf {
    void *ptr = nullptr;
    size_t bytes = 0;
    int reserve(size_t need, bool keep = false, hipStream_t stream = nullptr);
    void release();
    template <class T> T *as() const { return static_cast<T *>(ptr); }
};

// Device mirror of one host ParticleArray + its cell-list state.
struct DevArray {
    bool used = false;
    size_t n = 0, n_real = 0, cap = 0;
    double *prop[SPH_PROP_COUNT] = {};
    // neighbour-search state (valid after sph_nnps_update)
    DevBuf keys, keys_sorted, idx, perm; // uint32 each; perm: sorted position -> original index
    DevBuf cell_start;                   // uint32[n_cells + 1]
    DevBuf fkeys_sorted, fine_start;     // sorted fine keys (cell * SPH_NSUB + x sub-bin); uint32[n_cells * SPH_NSUB + 1]
    DevBuf tile_key, tile_id, tile_order; // traversal order of the 256-particle destination tiles (aggregated kernel)
    size_t n_tiles = 0;
    int tile_grid[4] = {0, 0, 0, 0};     // grid (ncx, ncy, ncz, block rows) the tile order was built for
    int tile_age = 0;                    // updates since the tile order was built
    // The REAL particles of the cell order (round 6): dlist[g] = sorted position of the g-th particle that is not a ghost /
    // image / padding row (original index < n_real of its array), ascending.  Built by the neighbour update when an array
    // holds rows behind its real particles; pair launches whose destinations are exactly the real particles
    // (Group.real = True, no start / stop) take their wave tiles from it: 64 consecutive REAL particles per wavefront
    // instead of 64 consecutive positions of which the ghosts are idle lanes (a slab rank of the 16 M dam break: 9.6 % of
    // the positions; Taylor-Green's periodic images: 20 %).  ctile_*: the traversal order of those tiles.
    DevBuf dlist, dl_cnt;
    size_t dlist_n = 0;                  // entries of dlist (0: none -- every position is a destination candidate)
    DevBuf ctile_key, ctile_id, ctile_order;
    size_t n_ctiles = 0;
    int ctile_grid[4] = {0, 0, 0, 0};
    int ctile_age = 0;
    bool m_known = false;                // every particle has the same mass: the last look found one value and nothing wrote m since
    double m_value = 0.0;                // ... this one
    // What the neighbour update knows about h and m WITHOUT looking: the range its last reduction found (`*_seen`), valid
    // while nothing that can write the property ran since (`*_dirty`: every entry point that writes h / m, appends
    // particles from outside, or hands out the raw pointer sets it -- sph_mark_written / _grown / _removed below).
    // Clean ranges let sph_nnps_update skip the reduction of h and m and, with them, the device->host round trip.
    bool h_dirty = true, m_dirty = true, h_seen = false, m_seen = false;
    bool has_padding = false;            // parked padding rows may sit behind the real particles (sph_halo_append_padded,
                                         // sph_domain_images_padded): the next neighbour update gives them cells of their own
    bool raw_hm = false;                 // a raw device pointer to h or m was handed out: writes through it cannot be tracked
    unsigned hm_writes = 0;              // writes of h / m (sph_mark_written) so far
    // ghost split: what sph_nnps_update knew when it binned the real particles (sph_nnps_update_ghosts verifies the ghosts
    // that arrived since against it, every time)
    bool m_known_binned = false, h_clean_binned = false, m_clean_binned = false;
    unsigned hm_writes_binned = 0;
    double h_lo = 0.0, h_hi = 0.0, m_lo = 0.0, m_hi = 0.0;
    int nnps_slot = -1;                  // position in the last sph_nnps_update list
    size_t perm_n = 0;                   // particles `perm` was built for (0: none / already applied)
    size_t perm_direct_n = 0;            // ... and `perm` is this array's own cell order written by the last sort (not a lazily derived one)
    bool unordered = false;              // the array lies in memory in no spatial order (the key pass said so): visited through perm
    // Ghost split (sph_nnps_update_ghosts): the tables above cover the first n_binned particles (the real ones: the
    // update ran before the ghosts of this step arrived); the g_n ghosts behind them are binned on the same grid
    // into tables of their own and read by the pair kernels as a second source segment of the array.
    size_t n_binned = 0, g_n = 0;
    bool m_mixed_ghosts = false;         // ghosts with another mass than the real particles' one were seen: no uniform-mass records
    DevBuf g_keys, g_fkeys, g_perm, g_fine_start, g_cell_start;
    // "no particle of this array is in tension": a device word the artificial-stress kernel (k_nosrc) sets to 1 when
    // any r_ij is non-zero; valid while that kernel covered every particle and nothing wrote r_ij since
    DevBuf tflag;
    bool tflag_valid = false;
    double *spare = nullptr;             // one more property-sized buffer: the compaction of sph_halo_remove_selected rotates through it
    size_t spare_cap = 0;                // ... holding this many doubles
    DevBuf slot8;                        // merged order only (sph_ctx::merged): uint8 nnps slot of every particle's array
};

// ghost selection lists of one array (sph_halo.hip)
struct HaloState {
    DevBuf flag[2], pos[2], list[2];
    size_t count[2] = {0, 0};
    size_t nsel = 0; // particles the flags of the last sph_halo_select cover
};

// T_PAIR: every pair launch; T_PAIR_FAM + family (sph_eval.hip enum Family): the same launches per equation family
// T_N_*: launch counters only (no time): pair launches on EOS-fused records, launches that kept / reused neighbour lists
enum TimerKey { T_NNPS, T_PACK, T_EOS, T_PAIR, T_STAGE, T_PAIR_FAM, T_N_EOSF = T_PAIR_FAM + 6, T_N_NLKEEP, T_N_NLREUSE, T_N_UMASS, T_N_MERGED, T_N_TFLAG, T_N_PHASE2, T_N_ASYNC, T_N_DLIST, T_N_ROWLDS, T_COUNT };

struct Timer {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    double ms = 0;
    long count = 0;
};

// records of one array in the packed buffer of the current sph_eval_group call
struct PackCache {
    unsigned long long epoch = 0;
    int fam = -1, sig = 0;
};

// padding rows of sph_halo_append_padded: parked far outside any domain, skipped by the bounds and spread by the keys
#define SPH_PARKED 1e18
#define SPH_PARKED_MIN 1e17

// property `prop` of the array is (or may be) written
static inline void sph_mark_written(DevArray &A, int prop)
{
    if (prop == SPH_H) { A.h_dirty = true; A.hm_writes++; }
    if (prop == SPH_M) { A.m_dirty = true; A.m_known = false; A.hm_writes++; }
}
// particles with values from outside (other ranks' ghosts, host appends) arrive
static inline void sph_mark_grown(DevArray &A) { A.h_dirty = A.m_dirty = true; A.m_known = false; }
// particles leave: a range of ONE value stays what it is while a particle is left, any other range may shrink
static inline void sph_mark_removed(DevArray &A, size_t n_left)
{
    if (!(A.h_seen && A.h_lo == A.h_hi && n_left > 0)) A.h_dirty = true;
    if (!(A.m_seen && A.m_lo == A.m_hi && n_left > 0)) { A.m_dirty = true; A.m_known = false; }
}

// a cell grid in the reference's arithmetic (nnps_base.pyx:942-978,1520-1575, linked_list_nnps.pyx:293-343)
struct GridHost {
    double cell_size = 0, hmin = 0, xmin[3] = {}, xmax[3] = {};
    int nc[3] = {1, 1, 1};
    long n_cells = 0;       // as the reference reports it (dim-aware)
    long n_cells_alloc = 0; // ncx * ncy * ncz: what the tables are sized for
    bool uniform_h = false;
    double h_uniform = 0;
};

struct sph_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    DevArray arr[SPH_MAX_ARRAYS];
    HaloState halo[SPH_MAX_ARRAYS];

    // grid of the last sph_nnps_update
    bool nnps_valid = false;
    int dim = 3;
    int narrays = 0;
    int ids[SPH_MAX_ARRAYS] = {};
    double radius_scale = 2.0, cell_size = 0, hmin = 0;
    double xmin[3] = {}, xmax[3] = {};
    int nc[3] = {1, 1, 1};
    long n_cells = 0;
    bool uniform_h = false; // hmin == hmax over all arrays
    double h_uniform = 0;
    // ONE cell order over the particles of ALL arrays of the grid (several arrays only; option merge_arrays): the array's
    // slot is the LEAST significant part of the sort key, so the particles of every array interleave per x sub-bin and a
    // row of cells is one contiguous run of the merged order.  `merged` uses the DevArray fields fkeys_sorted, keys_sorted,
    // perm (merged position -> original index in its own array), slot8, fine_start and the tile order; n = all particles.
    DevArray merged;
    bool merged_valid = false;
    long merge_arrays = 1;
    int phase_sig = -1;     // record layout the first half of a split evaluation packed (the second half repacks when its own differs)
    // The merged records carry a particle's class in the SIGN of rho.  k_pack_merged raises this device word when it meets a
    // density that is not positive; the neighbour updates carry the word to the host with their bounds (no round trip of
    // its own): the update that sees it returns an error -- the evaluations since the density went bad ran with wrong
    // classes -- and the context takes the per-destination path from then on.
    DevBuf xflag;
    bool merge_blocked = false;
    long split_pair = 0;    // split evaluations: 1 = interior wave tiles in phase 1, face tiles in phase 2 (default: all tiles in phase 2)
    long tension_flag = 1;  // elastic rates: r_ij gathered only when the source array's tension word says so
    // ... built FIRST by sph_nnps_update (one stable sort of all arrays' keys); the per-array orders and tables are a
    // stable compaction of it by slot, made when something first asks for them (nnps_need_tables)
    bool tables_valid = true;
    long lazy_tables = 1;
    // ghost split: the grid is padded by `extend` on both sides of every axis (the real particles' bounds must hold the
    // ghosts that arrive later); ghosts have a fine x index <= gfx_lo or >= gfx_hi (wavefronts whose candidate windows
    // stay inside never read a ghost segment)
    double extend[3] = {0.0, 0.0, 0.0};
    int gfx_lo = -0x7fffffff, gfx_hi = 0x7fffffff;
    int face_axis = -1;      // sph_nnps_set_ghost_faces: the slab axis (-1: none) and the faces outside which ghosts lie
    double face_lo = 0.0, face_hi = 0.0;
    bool ghosts_binned = false;
    DevBuf splitcnt, scan_part, bigq;
    // The hand-written particle sort (sph_nnps.hip): bucket histogram / starts / cursors, ticket words
    DevBuf sort_tab;
    size_t sort_tab_entries = 0;  // buckets the tables were zeroed for
    int sort_lbits = 0;           // low key bits sorted inside a bucket (chosen from the mean density, or fixed by option sort_lbits)
    double sort_groups = 0;       // atomic groups of the last key pass (n / 64 * 1..2 in cell order, n in no order)
    bool last_keys_via = false;   // ... which visited its array through the previous cell order
    size_t last_keys_n = 0;       // ... over this many particles of
    const DevArray *last_keys_array = nullptr; // ... this (single) array
    long via_unordered = 1;       // option: 0 = never visit through the previous order
    long hand_sort = 1;           // 0: option sort_lbits fixed the bucket size (profiling / tests)
    // The update without a device->host round trip (option async_update, default 1).  When h and m are known without
    // looking (DevArray::h_dirty / m_dirty) only the bounds of the positions are missing for the grid -- and ANY grid
    // whose cells are at least radius_scale * hmax wide gives the same neighbours (particles outside it are clamped into
    // its outermost cells).  Such an update bins on the grid of the PREVIOUS update's bounds (they arrived long ago) in
    // the same pass that reduces this update's bounds; those travel to `pin_async` behind an event nobody waits for.
    // The grid REPORTED (sph_nnps_info: cell_size, xmin, xmax, n_cells -- the reference's values, exactly) is computed
    // from this update's bounds when somebody asks.
    long async_update = 1;
    struct {
        bool valid = false;      // pin_async holds (or will hold, once `ev` has passed) the bounds of the last update
        bool pending = false;    // ... behind `ev`
        int dim = 0, narrays = 0, ids[SPH_MAX_ARRAYS] = {};
        double radius_scale = 0, cell_size_in = 0, extend[3] = {}, hr[2] = {};
    } lag;
    long park_cells = 0;          // cells behind the grid's own in the tables of the last update: where parked rows are binned
    // image counts of sph_domain_images_padded: device words [array][axis][side], their pinned copy and its event
    DevBuf dom_counts;
    double *dom_pin = nullptr;    // pinned: SPH_MAX_ARRAYS * 6 doubles
    hipEvent_t dom_ev = nullptr;
    bool dom_queued = false;
    hipEvent_t lag_ev = nullptr;
    double *pin_async = nullptr;  // pinned: 64 doubles
    GridHost rep;                 // the reported grid of the last update
    bool rep_valid = false;       // ... computed (else: from pin_async on demand)
    long n_async_updates = 0;     // updates that ran without a round trip (sph_timer_get "n_async")

    // scratch
    DevBuf cub_tmp, red_part, red_out, posh, aux, fposb, dkeys, dperm, tmp_u32a, tmp_u32b, gen_state, gapq;
    double *pinned = nullptr; // small pinned host buffer (64 doubles)

    // options
    long pair_variant = 6;
    long ablate = 0;
    long count_iters = 0;   // profiling: the pair kernel counts its phase-2 iterations (option count_iters; dump_counters prints them)
    DevBuf dbgc;
    long const_flags = 1;   // variant 6: compile-time equation flags when all sources agree (0: always run-time flags)
    long use_uniform_h = 1;
    long arith_f32 = 0;     // hand-written families compute in fp32 (fp32 records, fp32 accumulators; BASELINE config 5)
    long record_f32 = 0;    // packed records in fp32 (inputs rounded, arithmetic fp64): aggregated kernel only
    long tile_block_rows = 8; // destination tiles are traversed in blocks of this many cell rows (y) through all z planes; 0: memory order
    long row_mod3 = 3;        // the pair kernel visits a wavefront's 3x3 rows of cells in (y mod 3, z mod 3) order (PairArgs::row_mod3)
    double cur_dt = 0.0;    // dt of the sph_eval_group call being set up
    DevBuf csr_start[SPH_MAX_ARRAYS], csr_nbrs[SPH_MAX_ARRAYS]; // neighbour lists of generated loop_all families
    unsigned long long pack_epoch = 0;
    long pack_group = 0;    // 1 between the per-destination sph_eval_group calls of one host group (records shared)
    PackCache pack_cache[SPH_MAX_ARRAYS];
    long wcsph_nr = 0;      // profiling: doubles per compact WCSPH record (default 10; 12/14/16 pad the stride)
    long lds_pad = 0;       // profiling: extra dynamic LDS per pair-kernel workgroup (limits wavefronts per CU)
    int cur_nrec = 0;       // doubles per packed record of the pair launch being set up
    bool cur_eosf = false;  // the pair launch being set up reads the 64-byte WCSPH records (EOS recomputed per record)
    bool cur_umass = false; // ... with p / rho^2 in the mass slot (every source array has one mass)
    bool cur_eosv = false;  // ... the 64-byte variable-h WCSPH records [x y z h u v w rho] (one mass per source array)
    bool cur_elu = false;   // ... the elastic-rates records without h and m (uniform h, one mass per source array)
    bool cur_tvff = false;  // the pair launch being set up reads the state-fused TVF records (p and V recomputed from rho)
    long mass_fuse = 1;     // allow that
    bool want_mrange = false; // an evaluation could have used uniform-mass records: the next neighbour updates look at the masses
    long block_sorted_outputs = 0;
    long eos_fuse = 1;      // honour sph_group.src_eos (64-byte WCSPH records, p and cs recomputed from rho)
    long nl_reuse = 0;      // honour sph_group.nl_mode (neighbour lists kept between the pair passes of one evaluation): built,
                            // bit-identical, and measured SLOWER on MI355X (phase 1 overlaps other wavefronts' gathers)
    long fill_holes = 1;    // sph_halo_remove_selected moves the tail's kept rows into the holes when few particles leave (0: always the stable compaction)
    long row_lds = 0;       // experiment: k_pair_rowlds for the families that have it (the elastic rates on uniform-h records)
    long dest_list = 1;     // pair launches over the real particles take their wave tiles from DevArray::dlist
    long norm_masks = 1;    // shift a row's hit bits down to the lane's first hit
    // neighbour lists kept by the last nl_mode-1 pair pass
    DevBuf nlbuf;
    struct { bool valid = false; unsigned long long epoch = 0; int dst = -1, src = -1; size_t start = 0, stop = 0, nd = 0; } nl;
    unsigned long long nnps_epoch = 0; // bumped by every sph_nnps_update
    double h_known[2] = {0.0, -1.0};   // sph_nnps_set_h_range: hmin, hmax (hmax < 0: unknown)

    void *comm = nullptr;   // SphComm of libsphcomm.so (sph_comm.hip), or nullptr

    // timers
    bool timers_on = false;
    int timer_mask = 1;     // 1: every class; 2: the pair launches only (sph_timer_enable)
    Timer timers[T_COUNT];
};

struct ScopedTimer {
    sph_ctx *c;
    int key;
    hipEvent_t a = nullptr, b = nullptr;
    ScopedTimer(sph_ctx *ctx, int k);
    ~ScopedTimer();
};

// nnps.hip
int nnps_minmax(sph_ctx *c, int narrays, const int *ids, double *out8, int mm); // mm: 1 positions | 2 h | 4 m
int dev_scan_u32(sph_ctx *c, const uint32_t *in, uint32_t *out, size_t n, bool exclusive); // prefix sums, hand-written (sph_nnps.hip)
int dev_scan_u64(sph_ctx *c, const unsigned long long *in, unsigned long long *out, size_t n, bool exclusive);
int nnps_need_tables(sph_ctx *c); // per-array cell orders / tables of a merged-first update, on first use
int nnps_build_csr_device(sph_ctx *c, int src, int dst, DevBuf &start, DevBuf &nbrs, size_t *total);
// eval.hip: the same lists through the wave-tile pair kernel (count pass: start == nullptr; fill pass: start, nbrs)
int nnps_csr_pair_kernel(sph_ctx *c, int src, int dst, uint32_t *count, const uint32_t *start, uint32_t *nbrs);

// eval.hip helpers
static inline unsigned div_up(size_t a, unsigned b) { return (unsigned)((a + b - 1) / b); }
