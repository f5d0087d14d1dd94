// sph_ctx.hip -- context, device mirror of the host ParticleArrays, timers.
//
// Replaces the reference's DeviceHelper (pysph/base/device_helper.py:47-672:
// push/pull/resize of `pa.gpu`) with an explicit host<->HIP buffer table:
// the Python host keeps owning the numpy buffers and decides when to move data.
#include <cstddef>
#include <cstdlib>
#include "sph_internal.h"

static thread_local char g_err[1024] = "";

void sph_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

int DevBuf::reserve(size_t need, bool keep, hipStream_t stream)
{
    if (need <= bytes) return SPH_OK;
    size_t nb = need + need / 4 + 256;
    void *np = nullptr;
    HIP_TRY(hipMalloc(&np, nb));
    if (keep && ptr && bytes) {
        HIP_TRY(hipMemcpyAsync(np, ptr, bytes, hipMemcpyDeviceToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));
    }
    if (ptr) (void)hipFree(ptr);
    ptr = np;
    bytes = nb;
    return SPH_OK;
}

void DevBuf::release()
{
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    bytes = 0;
}

ScopedTimer::ScopedTimer(sph_ctx *ctx, int k) : c(ctx), key(k)
{
    if (!c->timers_on) return;
    // (timer_mask: sph_timer_enable(ctx, 2) times the pair launches only -- an event pair costs the stream a marker either
    // side of the region, ~5 us each on this runtime: eight of them per step were 2 % of the headline step)
    if (c->timer_mask == 2 && k != T_PAIR) return;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { a = b = nullptr; return; }
    (void)hipEventRecord(a, c->stream);
}

ScopedTimer::~ScopedTimer()
{
    if (!a) return;
    (void)hipEventRecord(b, c->stream);
    c->timers[key].pending.emplace_back(a, b);
}

static const char *PROP_NAMES[SPH_USER0] = {
    "x", "y", "z", "u", "v", "w", "h", "m", "rho", "p", "cs",
    "arho", "au", "av", "aw", "ax", "ay", "az", "dt_cfl", "dt_force",
    "V", "uhat", "vhat", "what", "auhat", "avhat", "awhat",
    "x0", "y0", "z0", "u0", "v0", "w0", "rho0", "vmag2",
    "e", "ae", "e0",
    "v00", "v01", "v02", "v10", "v11", "v12", "v20", "v21", "v22",
    "s00", "s01", "s02", "s11", "s12", "s22",
    "as00", "as01", "as02", "as11", "as12", "as22",
    "r00", "r01", "r02", "r11", "r12", "r22",
    "s000", "s010", "s020", "s110", "s120", "s220"};

extern "C" {

const char *sph_last_error(void) { return g_err; }
const char *sph_version(void) { return "sphhip 0.1 (gfx950)"; }

// user property names (process-wide: ids must agree between contexts/ranks
// that register the same names in the same order)
static char g_user_names[SPH_PROP_COUNT - SPH_USER0][48];
static int g_n_user = 0;

int sph_prop_id(const char *name)
{
    if (!name) return -1;
    for (int i = 0; i < SPH_USER0; i++)
        if (strcmp(name, PROP_NAMES[i]) == 0) return i;
    for (int i = 0; i < g_n_user; i++)
        if (strcmp(name, g_user_names[i]) == 0) return SPH_USER0 + i;
    return -1;
}

int sph_prop_register(const char *name)
{
    int id = sph_prop_id(name);
    if (id >= 0) return id;
    if (!name || !name[0] || strlen(name) >= sizeof g_user_names[0]) { sph_set_error("sph_prop_register: bad name"); return SPH_ERR_ARG; }
    if (g_n_user >= SPH_PROP_COUNT - SPH_USER0) { sph_set_error("sph_prop_register: all %d user property slots are taken", SPH_PROP_COUNT - SPH_USER0); return SPH_ERR_ARG; }
    strcpy(g_user_names[g_n_user], name);
    return SPH_USER0 + g_n_user++;
}

int sph_ctx_create(int device, void *stream, sph_ctx **out)
{
    if (!out) { sph_set_error("sph_ctx_create: out is NULL"); return SPH_ERR_ARG; }
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) {
        sph_set_error("sph_ctx_create: device %d not available (%d visible)", device, ndev);
        return SPH_ERR_ARG;
    }
    HIP_TRY(hipSetDevice(device));
    sph_ctx *c = new sph_ctx();
    c->device = device;
    if (stream) {
        c->stream = (hipStream_t)stream;
    } else {
        HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->own_stream = true;
    }
    HIP_TRY(hipHostMalloc((void **)&c->pinned, 64 * sizeof(double), hipHostMallocDefault));
    // development aid: SPH_PAIR_VARIANT overrides the default pair-kernel schedule
    if (const char *pv = getenv("SPH_PAIR_VARIANT")) c->pair_variant = atol(pv);
    *out = c;
    return SPH_OK;
}

int sph_ctx_destroy(sph_ctx *c)
{
    if (!c) return SPH_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (auto &A : c->arr) {
        for (auto &p : A.prop) if (p) (void)hipFree(p);
        if (A.spare) (void)hipFree(A.spare);
        A.keys.release(); A.keys_sorted.release(); A.idx.release(); A.perm.release();
        A.tile_key.release(); A.tile_id.release(); A.tile_order.release();
        A.dlist.release(); A.dl_cnt.release(); A.ctile_key.release(); A.ctile_id.release(); A.ctile_order.release();
        A.cell_start.release(); A.fkeys_sorted.release(); A.fine_start.release(); A.tflag.release();
        A.g_keys.release(); A.g_fkeys.release(); A.g_perm.release(); A.g_fine_start.release(); A.g_cell_start.release();
    }
    {
        DevArray &A = c->merged;
        A.keys_sorted.release(); A.perm.release(); A.fkeys_sorted.release(); A.fine_start.release(); A.slot8.release(); A.cell_start.release();
        A.tile_key.release(); A.tile_id.release(); A.tile_order.release();
        A.dlist.release(); A.dl_cnt.release(); A.ctile_key.release(); A.ctile_id.release(); A.ctile_order.release();
    }
    for (auto &H : c->halo)
        for (int s = 0; s < 2; s++) { H.flag[s].release(); H.pos[s].release(); H.list[s].release(); }
    for (DevBuf *b : {&c->dbgc, &c->gapq, &c->cub_tmp, &c->red_part, &c->red_out, &c->posh, &c->aux, &c->fposb, &c->dkeys, &c->dperm,
                      &c->tmp_u32a, &c->tmp_u32b, &c->gen_state, &c->nlbuf, &c->splitcnt, &c->scan_part, &c->bigq, &c->sort_tab, &c->xflag, &c->dom_counts})
        b->release();
    for (auto &b : c->csr_start) b.release();
    for (auto &b : c->csr_nbrs) b.release();
    for (auto &t : c->timers)
        for (auto &pr : t.pending) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    if (c->pinned) (void)hipHostFree(c->pinned);
    if (c->pin_async) (void)hipHostFree(c->pin_async);
    if (c->lag_ev) (void)hipEventDestroy(c->lag_ev);
    if (c->dom_pin) (void)hipHostFree(c->dom_pin);
    if (c->dom_ev) (void)hipEventDestroy(c->dom_ev);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return SPH_OK;
}

int sph_ctx_synchronize(sph_ctx *c)
{
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SPH_OK;
}

static int check_array(sph_ctx *c, int id, const char *who)
{
    if (!c) { sph_set_error("%s: ctx is NULL", who); return SPH_ERR_ARG; }
    if (id < 0 || id >= SPH_MAX_ARRAYS) { sph_set_error("%s: bad array id %d", who, id); return SPH_ERR_ARG; }
    return SPH_OK;
}

int sph_array_resize(sph_ctx *c, int id, size_t n, size_t n_real)
{
    SPH_TRY(check_array(c, id, "sph_array_resize"));
    if (n_real > n) { sph_set_error("sph_array_resize: n_real %zu > n %zu", n_real, n); return SPH_ERR_ARG; }
    HIP_TRY(hipSetDevice(c->device));
    DevArray &A = c->arr[id];
    A.used = true;
    if (n > A.cap) {
        size_t ncap = n + n / 8 + 64;
        for (int p = 0; p < SPH_PROP_COUNT; p++) {
            if (!A.prop[p]) continue;
            double *np = nullptr;
            HIP_TRY(hipMalloc((void **)&np, ncap * sizeof(double)));
            HIP_TRY(hipMemsetAsync(np, 0, ncap * sizeof(double), c->stream));
            if (A.n) HIP_TRY(hipMemcpyAsync(np, A.prop[p], A.n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            HIP_TRY(hipFree(A.prop[p]));
            A.prop[p] = np;
        }
        A.cap = ncap;
    }
    if (n != A.n) c->nnps_valid = false;
    if (n > A.n) A.tflag_valid = false; // new particles (ghosts, migrants): their r_ij were not looked at
    if (n > A.n) sph_mark_grown(A);     // ... nor their h and m
    if (n < A.n) sph_mark_removed(A, n);
    if (n <= n_real) A.has_padding = false; // (ghosts dropped: the padding rows went with them)
    A.n = n;
    A.n_real = n_real;
    return SPH_OK;
}

int sph_array_size(sph_ctx *c, int id, size_t *n, size_t *n_real)
{
    SPH_TRY(check_array(c, id, "sph_array_size"));
    if (n) *n = c->arr[id].n;
    if (n_real) *n_real = c->arr[id].n_real;
    return SPH_OK;
}

int sph_array_ensure_prop(sph_ctx *c, int id, int prop)
{
    SPH_TRY(check_array(c, id, "sph_array_ensure_prop"));
    if (prop < 0 || prop >= SPH_PROP_COUNT) { sph_set_error("bad property id %d", prop); return SPH_ERR_ARG; }
    DevArray &A = c->arr[id];
    if (!A.used) { sph_set_error("array %d was never sized (sph_array_resize)", id); return SPH_ERR_STATE; }
    if (A.prop[prop]) return SPH_OK;
    HIP_TRY(hipSetDevice(c->device));
    size_t cap = A.cap ? A.cap : 64;
    A.cap = cap;
    HIP_TRY(hipMalloc((void **)&A.prop[prop], cap * sizeof(double)));
    HIP_TRY(hipMemsetAsync(A.prop[prop], 0, cap * sizeof(double), c->stream));
    sph_mark_written(A, prop);
    return SPH_OK;
}

int sph_array_push(sph_ctx *c, int id, int prop, const double *host, size_t offset, size_t n)
{
    SPH_TRY(sph_array_ensure_prop(c, id, prop));
    DevArray &A = c->arr[id];
    if (offset + n > A.n) { sph_set_error("sph_array_push: %zu+%zu > n=%zu", offset, n, A.n); return SPH_ERR_ARG; }
    if (n == 0) return SPH_OK;
    HIP_TRY(hipMemcpyAsync(A.prop[prop] + offset, host, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    // pageable host memory: the runtime has staged the data once this returns
    // only after a sync; keep the call synchronous so Python may reuse `host`.
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (prop == SPH_X || prop == SPH_Y || prop == SPH_Z || prop == SPH_H) c->nnps_valid = false;
    // positions from the host may lie anywhere (a restart, a translated body): the next neighbour update does not bin on
    // the grid of the previous update's bounds but looks at the particles first
    if (prop == SPH_X || prop == SPH_Y || prop == SPH_Z) c->lag.valid = false;
    if (prop == SPH_M) A.m_mixed_ghosts = false;
    sph_mark_written(A, prop); // h / m: until the next sph_nnps_update has looked at them
    if (prop >= SPH_R00 && prop <= SPH_R22) A.tflag_valid = false;
    return SPH_OK;
}

int sph_array_pull(sph_ctx *c, int id, int prop, double *host, size_t offset, size_t n)
{
    SPH_TRY(check_array(c, id, "sph_array_pull"));
    DevArray &A = c->arr[id];
    if (prop < 0 || prop >= SPH_PROP_COUNT || !A.prop[prop]) {
        sph_set_error("sph_array_pull: array %d has no device copy of property %d", id, prop);
        return SPH_ERR_MISSING_PROP;
    }
    if (offset + n > A.n) { sph_set_error("sph_array_pull: %zu+%zu > n=%zu", offset, n, A.n); return SPH_ERR_ARG; }
    if (n == 0) return SPH_OK;
    HIP_TRY(hipMemcpyAsync(host, A.prop[prop] + offset, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SPH_OK;
}

// n rows of one property set to `value`, from row `offset` on (ghost rows whose promised-uniform h / m did not travel,
// pysph_amd/parallel.py).  A value the neighbour update already knows as the array's ONE h / m leaves that knowledge valid.
__global__ __launch_bounds__(256) static void k_fill_f64(double *__restrict__ p, size_t n, double v)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

int sph_array_fill(sph_ctx *c, int id, int prop, double value, size_t offset, size_t n)
{
    SPH_TRY(sph_array_ensure_prop(c, id, prop));
    DevArray &A = c->arr[id];
    if (offset + n > A.n) { sph_set_error("sph_array_fill: %zu+%zu > n=%zu", offset, n, A.n); return SPH_ERR_ARG; }
    if (n == 0) return SPH_OK;
    HIP_TRY(hipSetDevice(c->device));
    hipLaunchKernelGGL(k_fill_f64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, A.prop[prop] + offset, n, value);
    if (prop == SPH_X || prop == SPH_Y || prop == SPH_Z || prop == SPH_H) c->nnps_valid = false;
    const bool same_h = prop == SPH_H && A.h_seen && !A.h_dirty && A.h_lo == value && A.h_hi == value;
    const bool same_m = prop == SPH_M && A.m_seen && !A.m_dirty && A.m_lo == value && A.m_hi == value;
    if (!same_h && !same_m) sph_mark_written(A, prop);
    if (prop >= SPH_R00 && prop <= SPH_R22) A.tflag_valid = false;
    return SPH_OK;
}

// "property `prop` of the array was written behind the library's back" / "do not trust what you know of it": for h and m
// the next neighbour update looks at the values again (one update with a device->host round trip)
int sph_array_mark_written(sph_ctx *c, int id, int prop)
{
    SPH_TRY(check_array(c, id, "sph_array_mark_written"));
    if (prop < 0 || prop >= SPH_PROP_COUNT) { sph_set_error("sph_array_mark_written: bad property %d", prop); return SPH_ERR_ARG; }
    DevArray &A = c->arr[id];
    sph_mark_written(A, prop);
    if (prop == SPH_X || prop == SPH_Y || prop == SPH_Z || prop == SPH_H) c->nnps_valid = false;
    if (prop == SPH_X || prop == SPH_Y || prop == SPH_Z) c->lag.valid = false;
    if (prop == SPH_M) A.m_mixed_ghosts = false;
    if (prop >= SPH_R00 && prop <= SPH_R22) A.tflag_valid = false;
    return SPH_OK;
}

int sph_array_device_ptr(sph_ctx *c, int id, int prop, void **dptr)
{
    SPH_TRY(sph_array_ensure_prop(c, id, prop));
    *dptr = c->arr[id].prop[prop];
    // whoever holds the raw pointer may write h or m at any time: every neighbour update looks at them from now on
    if (prop == SPH_M || prop == SPH_H) { c->arr[id].raw_hm = true; sph_mark_written(c->arr[id], prop); }
    if (prop >= SPH_R00 && prop <= SPH_R22) c->arr[id].tflag_valid = false;
    return SPH_OK;
}

int sph_set_option(sph_ctx *c, const char *key, long value)
{
    if (strcmp(key, "pair_variant") == 0) {
        if (value != 0 && value != 6) { sph_set_error("pair_variant must be 0 (direct per-lane walk) or 6 (wavefront tiles, the default)"); return SPH_ERR_ARG; }
        c->pair_variant = value;
        return SPH_OK;
    }
    if (strcmp(key, "uniform_h") == 0) { c->use_uniform_h = value; return SPH_OK; }
    if (strcmp(key, "record_f32") == 0) { c->record_f32 = value ? 1 : 0; return SPH_OK; }
    if (strcmp(key, "arith_f32") == 0) { c->arith_f32 = value ? 1 : 0; return SPH_OK; }
    if (strcmp(key, "tile_block_rows") == 0) {
        if (value < 0 || value > 4096) { sph_set_error("tile_block_rows out of range"); return SPH_ERR_ARG; }
        c->tile_block_rows = value; c->nnps_valid = false; return SPH_OK;
    }
    if (strcmp(key, "row_mod3") == 0) {
        if (value < 0 || value > 4) { sph_set_error("row_mod3 must be 0..4"); return SPH_ERR_ARG; }
        c->row_mod3 = value; return SPH_OK;
    }
    if (strcmp(key, "pack_group") == 0) { c->pack_group = value ? 1 : 0; c->pack_epoch++; return SPH_OK; }
    if (strcmp(key, "wcsph_nr") == 0) { c->wcsph_nr = value; return SPH_OK; }
    if (strcmp(key, "lds_pad") == 0) { c->lds_pad = value; return SPH_OK; }
#ifndef SPH_PROFILING
    if ((strcmp(key, "ablate") == 0 || strcmp(key, "count_iters") == 0) && value != 0) {
        sph_set_error("option '%s': this libsphhip.so was built without the pair kernel's profiling hooks (make PROFILING=1)", key);
        return SPH_ERR_ARG;
    }
#endif
    if (strcmp(key, "ablate") == 0) { c->ablate = value; return SPH_OK; }
    if (strcmp(key, "count_iters") == 0) {
        c->count_iters = value;
        if (value) { SPH_TRY(c->dbgc.reserve(64)); HIP_TRY(hipMemset(c->dbgc.ptr, 0, 64)); }
        return SPH_OK;
    }
    if (strcmp(key, "dump_counters") == 0) {
        unsigned long long h[4] = {};
        if (c->dbgc.ptr) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipMemcpy(h, c->dbgc.ptr, 32, hipMemcpyDeviceToHost)); }
        fprintf(stderr, "pair kernel counters: phase-2 iterations %llu, phase-2 calls %llu, wavefronts %llu, row tiles %llu\n", h[0], h[1], h[2], h[3]);
        return SPH_OK;
    }
    if (strcmp(key, "const_flags") == 0) { c->const_flags = value; return SPH_OK; }
    if (strcmp(key, "invalidate_nnps") == 0) { c->nnps_valid = false; return SPH_OK; }
    if (strcmp(key, "eos_fuse") == 0) { c->eos_fuse = value; return SPH_OK; }
    if (strcmp(key, "mass_fuse") == 0) { c->mass_fuse = value; return SPH_OK; }
    if (strcmp(key, "nl_reuse") == 0) { c->nl_reuse = value; c->nl.valid = false; return SPH_OK; }
    if (strcmp(key, "norm_masks") == 0) { c->norm_masks = value; return SPH_OK; }
    if (strcmp(key, "row_lds") == 0) { c->row_lds = value; return SPH_OK; }
    if (strcmp(key, "fill_holes") == 0) { c->fill_holes = value ? 1 : 0; return SPH_OK; }
    if (strcmp(key, "dest_list") == 0) { c->dest_list = value < 0 ? 0 : (value > 2 ? 2 : value); c->nnps_valid = false; return SPH_OK; }
    if (strcmp(key, "merge_arrays") == 0) { c->merge_arrays = value ? 1 : 0; c->nnps_valid = false; return SPH_OK; }
    if (strcmp(key, "split_pair") == 0) { c->split_pair = value ? 1 : 0; return SPH_OK; }
    if (strcmp(key, "tension_flag") == 0) { c->tension_flag = value ? 1 : 0; return SPH_OK; }
    if (strcmp(key, "lazy_tables") == 0) { c->lazy_tables = value ? 1 : 0; c->nnps_valid = false; return SPH_OK; }
    if (strcmp(key, "async_update") == 0) { c->async_update = value ? 1 : 0; return SPH_OK; }
    if (strcmp(key, "via_unordered") == 0) { c->via_unordered = value ? 1 : 0; return SPH_OK; }
    if (strcmp(key, "sort_lbits") == 0) { // 0: from the mean density; 9..11: fixed low key bits of the particle sort (profiling / tests)
        if (value != 0 && (value < 9 || value > 11)) { sph_set_error("sort_lbits must be 0 or 9..11"); return SPH_ERR_ARG; }
        c->sort_lbits = (int)value; c->hand_sort = value == 0; return SPH_OK;
    }
    sph_set_error("sph_set_option: unknown key '%s'", key);
    return SPH_ERR_ARG;
}

int sph_timer_enable(sph_ctx *c, int on) { c->timers_on = on != 0; c->timer_mask = on; return SPH_OK; }

// ---- ABI self-description ---------------------------------------------------
struct AbiField { const char *st, *field; long off; };
#define ABI_F(S, F) {#S, #F, (long)offsetof(S, F)}
static const AbiField ABI_FIELDS[] = {
    ABI_F(sph_kernel, kind), ABI_F(sph_kernel, dim), ABI_F(sph_kernel, fac), ABI_F(sph_kernel, radius_scale), ABI_F(sph_kernel, deltap),
    ABI_F(sph_equation, kind), ABI_F(sph_equation, dest), ABI_F(sph_equation, nsrc), ABI_F(sph_equation, src), ABI_F(sph_equation, par),
    ABI_F(sph_group, real), ABI_F(sph_group, start_idx), ABI_F(sph_group, stop_idx), ABI_F(sph_group, neq), ABI_F(sph_group, eqs),
    ABI_F(sph_group, src_eos), ABI_F(sph_group, eos_par), ABI_F(sph_group, nl_mode), ABI_F(sph_group, phase),
    ABI_F(sph_gen_family, launch), ABI_F(sph_gen_family, dest), ABI_F(sph_gen_family, nsrc), ABI_F(sph_gen_family, src),
    ABI_F(sph_gen_family, src_flags), ABI_F(sph_gen_family, n_sprops), ABI_F(sph_gen_family, sprops), ABI_F(sph_gen_family, n_din),
    ABI_F(sph_gen_family, din), ABI_F(sph_gen_family, n_dout), ABI_F(sph_gen_family, dout), ABI_F(sph_gen_family, npar),
    ABI_F(sph_gen_family, par), ABI_F(sph_gen_family, real), ABI_F(sph_gen_family, start_idx), ABI_F(sph_gen_family, stop_idx),
    ABI_F(sph_gen_family, split_init), ABI_F(sph_gen_family, loop_all), ABI_F(sph_gen_family, also_pair), ABI_F(sph_gen_family, init_pair),
    ABI_F(sph_gen_family, nstate), ABI_F(sph_gen_family, state), ABI_F(sph_gen_family, launch_f32),
    ABI_F(sph_gen_args, stream), ABI_F(sph_gen_args, rec), ABI_F(sph_gen_args, mode), ABI_F(sph_gen_args, par), ABI_F(sph_gen_args, state),
    ABI_F(sph_gen_args, row_mod3),
};
#undef ABI_F

long sph_abi_sizeof(const char *name)
{
    if (!name) return -1;
    if (!strcmp(name, "sph_kernel")) return (long)sizeof(sph_kernel);
    if (!strcmp(name, "sph_equation")) return (long)sizeof(sph_equation);
    if (!strcmp(name, "sph_group")) return (long)sizeof(sph_group);
    if (!strcmp(name, "sph_gen_family")) return (long)sizeof(sph_gen_family);
    if (!strcmp(name, "sph_gen_args")) return (long)sizeof(sph_gen_args);
    return -1;
}

long sph_abi_offsetof(const char *name, const char *field)
{
    if (!name || !field) return -1;
    for (const AbiField &f : ABI_FIELDS)
        if (!strcmp(f.st, name) && !strcmp(f.field, field)) return f.off;
    return -1;
}

static int timer_drain(sph_ctx *c)
{
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (auto &t : c->timers) {
        for (auto &pr : t.pending) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) { t.ms += ms; t.count++; }
            (void)hipEventDestroy(pr.first);
            (void)hipEventDestroy(pr.second);
        }
        t.pending.clear();
    }
    return SPH_OK;
}

int sph_timer_reset(sph_ctx *c)
{
    SPH_TRY(timer_drain(c));
    for (auto &t : c->timers) { t.ms = 0; t.count = 0; }
    return SPH_OK;
}

int sph_timer_get(sph_ctx *c, const char *key, double *ms, long *count)
{
    static const char *names[T_COUNT] = {"nnps", "pack", "eos", "pair", "stage",
                                         "pair_none", "pair_wcsph", "pair_density", "pair_tvf", "pair_vgrad", "pair_elastic",
                                         "n_eos_fused", "n_nl_keep", "n_nl_reuse", "n_mass_fused", "n_merged", "n_tension_flag", "n_phase2", "n_async", "n_dest_list", "n_row_lds"};
    SPH_TRY(timer_drain(c));
    for (int i = 0; i < T_COUNT; i++)
        if (strcmp(key, names[i]) == 0) {
            if (ms) *ms = c->timers[i].ms;
            if (count) *count = c->timers[i].count;
            return SPH_OK;
        }
    sph_set_error("sph_timer_get: unknown key '%s'", key);
    return SPH_ERR_ARG;
}

} // extern "C"
