// sph_comm.hip -- ghost-halo transport on RCCL for hosts WITHOUT torch.distributed.
//
// The Python host reaches RCCL through torch.distributed (pysph_amd/parallel.py);
// a C / C++ / Fortran host binding this library cannot.  These entry points give
// it the same exchange (SURVEY.md 8b "sph_comm_init_all / sph_halo_exchange /
// sph_allreduce_*"): ParallelManager.update's ghost refresh
// (pysph/parallel/parallel_manager.pyx:512-530, remote_exchange_data :159-210)
// and the dt / bounds reductions (:463, :937-945) as
//   select (device) -> counts to the two neighbours -> pack (device)
//   -> ncclGroupStart; ncclSend/ncclRecv x <= 2; ncclGroupEnd -> append (device)
// all on the context's stream, point-to-point over xGMI.  Built as its own
// shared object (libsphcomm.so, links librccl) so that libsphhip.so has no RCCL
// dependency and a process that already holds torch's bundled RCCL never sees
// two copies through one library.
#include "sph_internal.h"

#include <rccl/rccl.h>

#include <cmath>
#include <cstdlib>

#define NCCL_TRY(expr)                                                                   \
    do {                                                                                 \
        ncclResult_t _r = (expr);                                                        \
        if (_r != ncclSuccess) {                                                         \
            sph_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, ncclGetErrorString(_r)); \
            return SPH_ERR_HIP;                                                          \
        }                                                                                \
    } while (0)

// inside ncclGroupStart / ncclGroupEnd: an error closes the group before it is reported (an open group
// would leave the peers blocked in their matching receive and every later call on the communicator broken)
#define NCCL_TRY_IN_GROUP(expr)                                                          \
    do {                                                                                 \
        ncclResult_t _r = (expr);                                                        \
        if (_r != ncclSuccess) {                                                         \
            (void)ncclGroupEnd();                                                        \
            sph_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, ncclGetErrorString(_r)); \
            return SPH_ERR_HIP;                                                          \
        }                                                                                \
    } while (0)

struct SphComm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    DevBuf send[2], recv[2], cnt; // payloads per face, 4 counters (send lo/hi, recv lo/hi)
    unsigned long long *h_cnt = nullptr; // pinned, 16 words: [0..3] counters (sent lo/hi, received lo/hi), [4..7] (as doubles) the 4 message headers
    // fixed-capacity protocol: rows the messages of (array, face) are sized for; 0 = not known yet.
    // Both ends of a face derive them from the SAME count with the same rule, so they stay in step.
    size_t cap_send[SPH_MAX_ARRAYS][2] = {}, cap_recv[SPH_MAX_ARRAYS][2] = {};
    bool cap_known[SPH_MAX_ARRAYS] = {};
};

// pysph_amd/parallel.py _capacity / _next_capacity -- the two transports must agree
static size_t halo_capacity(size_t count) { return ((count + count / 4 + 4096 + 1023) / 1024) * 1024; }
static size_t halo_next_capacity(size_t cap, bool known, size_t count)
{
    if (!known || count > cap || count + count / 8 + 1024 > cap || 4 * count + 16384 < cap) return halo_capacity(count);
    return cap;
}

static SphComm *comm_of(sph_ctx *c) { return static_cast<SphComm *>(c->comm); }

extern "C" {

int sph_comm_unique_id(void *id128)
{
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
    if (!id128) { sph_set_error("sph_comm_unique_id: NULL"); return SPH_ERR_ARG; }
    NCCL_TRY(ncclGetUniqueId(static_cast<ncclUniqueId *>(id128)));
    return SPH_OK;
}

static int comm_attach(sph_ctx *c, ncclComm_t comm, int rank, int world)
{
    SphComm *m = new SphComm();
    m->comm = comm; m->rank = rank; m->world = world;
    HIP_TRY(hipHostMalloc((void **)&m->h_cnt, 16 * sizeof(unsigned long long), hipHostMallocDefault));
    c->comm = m;
    return SPH_OK;
}

int sph_comm_init_rank(sph_ctx *c, int rank, int world, const void *id128)
{
    if (!c || !id128 || world < 1 || rank < 0 || rank >= world) { sph_set_error("sph_comm_init_rank: bad arguments"); return SPH_ERR_ARG; }
    if (c->comm) { sph_set_error("sph_comm_init_rank: context already has a communicator"); return SPH_ERR_STATE; }
    HIP_TRY(hipSetDevice(c->device));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    ncclComm_t comm;
    NCCL_TRY(ncclCommInitRank(&comm, world, id, rank));
    return comm_attach(c, comm, rank, world);
}

int sph_comm_init_all(int ndev, sph_ctx **ctxs)
{
    if (ndev < 1 || ndev > 64 || !ctxs) { sph_set_error("sph_comm_init_all: bad arguments"); return SPH_ERR_ARG; }
    std::vector<int> devs(ndev);
    for (int i = 0; i < ndev; i++) {
        if (!ctxs[i] || ctxs[i]->comm) { sph_set_error("sph_comm_init_all: context %d missing or already initialised", i); return SPH_ERR_ARG; }
        devs[i] = ctxs[i]->device;
    }
    std::vector<ncclComm_t> comms(ndev);
    NCCL_TRY(ncclCommInitAll(comms.data(), ndev, devs.data()));
    for (int i = 0; i < ndev; i++) SPH_TRY(comm_attach(ctxs[i], comms[i], i, ndev));
    return SPH_OK;
}

// One NCCL group of point-to-point transfers of device doubles on the CONTEXT'S stream: what the Python transport hands
// to torch.distributed.batch_isend_irecv, without the process group's stream hand-over in front of and behind the RCCL
// kernel (measured on a slab rank of the 16 M dam break: ~30 us + ~16 us of idle stream per exchange).  Sends are posted
// in the order given, then the receives: between one pair of ranks messages match in posting order.
int sph_comm_sendrecv(sph_ctx *c, int nsend, const void *const *sendbufs, const size_t *sendcounts, const int *sendpeers,
                      int nrecv, void *const *recvbufs, const size_t *recvcounts, const int *recvpeers)
{
    if (!c || !c->comm || nsend < 0 || nrecv < 0 || (nsend && (!sendbufs || !sendcounts || !sendpeers)) ||
        (nrecv && (!recvbufs || !recvcounts || !recvpeers))) {
        sph_set_error("sph_comm_sendrecv: bad arguments (or no communicator: sph_comm_init_rank first)");
        return SPH_ERR_ARG;
    }
    SphComm *m = comm_of(c);
    for (int k = 0; k < nsend; k++)
        if (sendpeers[k] < 0 || sendpeers[k] >= m->world) { sph_set_error("sph_comm_sendrecv: peer %d of %d ranks", sendpeers[k], m->world); return SPH_ERR_ARG; }
    for (int k = 0; k < nrecv; k++)
        if (recvpeers[k] < 0 || recvpeers[k] >= m->world) { sph_set_error("sph_comm_sendrecv: peer %d of %d ranks", recvpeers[k], m->world); return SPH_ERR_ARG; }
    if (nsend + nrecv == 0) return SPH_OK;
    HIP_TRY(hipSetDevice(c->device));
    NCCL_TRY(ncclGroupStart());
    for (int k = 0; k < nsend; k++)
        NCCL_TRY_IN_GROUP(ncclSend(sendbufs[k], sendcounts[k], ncclDouble, sendpeers[k], m->comm, c->stream));
    for (int k = 0; k < nrecv; k++)
        NCCL_TRY_IN_GROUP(ncclRecv(recvbufs[k], recvcounts[k], ncclDouble, recvpeers[k], m->comm, c->stream));
    NCCL_TRY(ncclGroupEnd());
    return SPH_OK;
}

int sph_comm_destroy(sph_ctx *c)
{
    if (!c || !c->comm) return SPH_OK;
    SphComm *m = comm_of(c);
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (m->comm) (void)ncclCommDestroy(m->comm);
    for (DevBuf *b : {&m->send[0], &m->send[1], &m->recv[0], &m->recv[1], &m->cnt}) b->release();
    if (m->h_cnt) (void)hipHostFree(m->h_cnt);
    delete m;
    c->comm = nullptr;
    return SPH_OK;
}

// neighbours of `rank` in a 1-D slab decomposition: side 0 = lower face
static int peer_of(const SphComm *m, int side, int periodic, double period, double *shift)
{
    *shift = 0.0;
    if (side == 0) {
        if (m->rank > 0) return m->rank - 1;
        if (periodic) { *shift = +period; return m->world - 1; }
    } else {
        if (m->rank < m->world - 1) return m->rank + 1;
        if (periodic) { *shift = -period; return 0; }
    }
    return -1;
}

// Ghost refresh of `n` contexts in one call: n == 1 in the one-process-per-GPU
// model; n == ndev when one process drives all GPUs (sph_comm_init_all), where
// the sends and receives of all devices must sit in one NCCL group.
int sph_halo_exchange_all(int n, sph_ctx **ctxs, int array_id, int axis, const double *lo, const double *hi,
                          double width, int periodic, double period, int nprops, const int *props, int drop,
                          size_t *counts4)
{
    if (n < 1 || !ctxs || !lo || !hi || !props || nprops < 1 || nprops > SPH_PROP_COUNT || axis < 0 || axis > 2) {
        sph_set_error("sph_halo_exchange: bad arguments");
        return SPH_ERR_ARG;
    }
    if (array_id < 0 || array_id >= SPH_MAX_ARRAYS) { sph_set_error("sph_halo_exchange: bad array id"); return SPH_ERR_ARG; }
    struct Face { int peer; double shift; size_t ns, nr; };
    std::vector<Face> F(2 * n);
    // Fixed-capacity protocol (the default of both transports, pysph_amd/parallel.py exchange_halos): once
    // an exchange has told both ends of every face how many rows cross it, the messages have a fixed size
    // derived from that count, selection + packing run on the device without the host seeing the counts
    // (sph_halo_select_pack), and ONE small readback per device returns the headers.  The first exchange,
    // and SPH_HALO_PROTOCOL=handshake, use the counts handshake below.
    const char *proto = getenv("SPH_HALO_PROTOCOL");
    bool fixed = !(proto && strcmp(proto, "handshake") == 0);
    for (int i = 0; i < n && fixed; i++) {
        SphComm *m = ctxs[i] ? comm_of(ctxs[i]) : nullptr;
        if (!m) { sph_set_error("sph_halo_exchange: context %d has no communicator (sph_comm_init_rank / _all)", i); return SPH_ERR_STATE; }
        fixed = m->cap_known[array_id];
    }
    if (fixed) {
        // 1. drop, buffers, select + pack on every device
        for (int i = 0; i < n; i++) {
            sph_ctx *c = ctxs[i];
            SphComm *m = comm_of(c);
            HIP_TRY(hipSetDevice(c->device));
            size_t nn, nreal;
            SPH_TRY(sph_array_size(c, array_id, &nn, &nreal));
            if (drop) SPH_TRY(sph_array_resize(c, array_id, nreal, nreal));
            double shift2[2];
            size_t cap2[2];
            void *dst2[2];
            for (int s = 0; s < 2; s++) {
                Face &f = F[2 * i + s];
                f.peer = peer_of(m, s, periodic, period, &f.shift);
                f.ns = f.nr = 0;
                shift2[s] = f.shift;
                cap2[s] = m->cap_send[array_id][s];
                SPH_TRY(m->send[s].reserve((cap2[s] * nprops + 1) * sizeof(double)));
                SPH_TRY(m->recv[s].reserve((m->cap_recv[array_id][s] * nprops + 1) * sizeof(double)));
                dst2[s] = f.peer >= 0 ? m->send[s].ptr : nullptr;
            }
            SPH_TRY(sph_halo_select_pack(c, array_id, axis, lo[i] + width, hi[i] - width, 0, nprops, props, shift2, cap2, dst2));
        }
        // 2. one group of fixed-size point-to-point transfers (hi face sent first, lo face received first)
        NCCL_TRY(ncclGroupStart());
        for (int i = 0; i < n; i++) {
            sph_ctx *c = ctxs[i];
            SphComm *m = comm_of(c);
            if (hipSetDevice(c->device) != hipSuccess) { (void)ncclGroupEnd(); sph_set_error("sph_halo_exchange: hipSetDevice failed"); return SPH_ERR_HIP; }
            for (int s = 1; s >= 0; s--)
                if (F[2 * i + s].peer >= 0)
                    NCCL_TRY_IN_GROUP(ncclSend(m->send[s].ptr, m->cap_send[array_id][s] * nprops + 1, ncclDouble, F[2 * i + s].peer, m->comm, c->stream));
            for (int s = 0; s < 2; s++)
                if (F[2 * i + s].peer >= 0)
                    NCCL_TRY_IN_GROUP(ncclRecv(m->recv[s].ptr, m->cap_recv[array_id][s] * nprops + 1, ncclDouble, F[2 * i + s].peer, m->comm, c->stream));
        }
        NCCL_TRY(ncclGroupEnd());
        // 3. the one readback: the headers this device packed and the ones it received
        std::vector<double> hdr(4 * n, 0.0);
        bool over = false;
        for (int i = 0; i < n; i++) {
            sph_ctx *c = ctxs[i];
            SphComm *m = comm_of(c);
            HIP_TRY(hipSetDevice(c->device));
            double *h = reinterpret_cast<double *>(m->h_cnt + 4);
            for (int s = 0; s < 2; s++) {
                h[s] = h[2 + s] = 0.0;
                if (F[2 * i + s].peer < 0) continue;
                HIP_TRY(hipMemcpyAsync(h + s, m->send[s].as<double>() + m->cap_send[array_id][s] * nprops, sizeof(double), hipMemcpyDeviceToHost, c->stream));
                HIP_TRY(hipMemcpyAsync(h + 2 + s, m->recv[s].as<double>() + m->cap_recv[array_id][s] * nprops, sizeof(double), hipMemcpyDeviceToHost, c->stream));
            }
            HIP_TRY(hipStreamSynchronize(c->stream));
            for (int k = 0; k < 4; k++) { hdr[4 * i + k] = h[k]; over |= h[k] < 0.0; }
            for (int s = 0; s < 2; s++) { F[2 * i + s].ns = (size_t)fabs(h[s]); F[2 * i + s].nr = (size_t)fabs(h[2 + s]); }
        }
        // 4. faces that outgrew their capacity: the pair repeats them with the exact size
        std::vector<size_t> stride(2 * n);
        for (int i = 0; i < n; i++)
            for (int s = 0; s < 2; s++) stride[2 * i + s] = comm_of(ctxs[i])->cap_recv[array_id][s];
        if (over) {
            for (int i = 0; i < n; i++) {
                sph_ctx *c = ctxs[i];
                SphComm *m = comm_of(c);
                HIP_TRY(hipSetDevice(c->device));
                if (hdr[4 * i] < 0.0 || hdr[4 * i + 1] < 0.0) { // the index lists of the faces, this time
                    size_t cnt[2];
                    SPH_TRY(sph_halo_select(c, array_id, axis, 0, lo[i] + width, hi[i] - width, 0.0, 0, cnt));
                }
                for (int s = 0; s < 2; s++) {
                    Face &f = F[2 * i + s];
                    if (hdr[4 * i + s] < 0.0) {
                        SPH_TRY(m->send[s].reserve((f.ns * nprops + 1) * sizeof(double)));
                        SPH_TRY(sph_halo_pack(c, array_id, s, nprops, props, axis, f.shift, m->send[s].ptr));
                    }
                    if (hdr[4 * i + 2 + s] < 0.0) {
                        SPH_TRY(m->recv[s].reserve((f.nr * nprops + 1) * sizeof(double)));
                        stride[2 * i + s] = f.nr;
                    }
                }
            }
            NCCL_TRY(ncclGroupStart());
            for (int i = 0; i < n; i++) {
                sph_ctx *c = ctxs[i];
                SphComm *m = comm_of(c);
                if (hipSetDevice(c->device) != hipSuccess) { (void)ncclGroupEnd(); sph_set_error("sph_halo_exchange: hipSetDevice failed"); return SPH_ERR_HIP; }
                for (int s = 1; s >= 0; s--)
                    if (hdr[4 * i + s] < 0.0)
                        NCCL_TRY_IN_GROUP(ncclSend(m->send[s].ptr, F[2 * i + s].ns * nprops, ncclDouble, F[2 * i + s].peer, m->comm, c->stream));
                for (int s = 0; s < 2; s++)
                    if (hdr[4 * i + 2 + s] < 0.0)
                        NCCL_TRY_IN_GROUP(ncclRecv(m->recv[s].ptr, F[2 * i + s].nr * nprops, ncclDouble, F[2 * i + s].peer, m->comm, c->stream));
            }
            NCCL_TRY(ncclGroupEnd());
        }
        // 5. append (lo side first), capacities for the next exchange
        for (int i = 0; i < n; i++) {
            sph_ctx *c = ctxs[i];
            SphComm *m = comm_of(c);
            HIP_TRY(hipSetDevice(c->device));
            for (int s = 0; s < 2; s++) {
                Face &f = F[2 * i + s];
                if (f.nr) SPH_TRY(sph_halo_append_strided(c, array_id, nprops, props, m->recv[s].ptr, f.nr, stride[2 * i + s]));
                m->cap_send[array_id][s] = halo_next_capacity(m->cap_send[array_id][s], true, f.ns);
                m->cap_recv[array_id][s] = halo_next_capacity(m->cap_recv[array_id][s], true, f.nr);
            }
            if (counts4) {
                counts4[4 * i + 0] = F[2 * i].ns; counts4[4 * i + 1] = F[2 * i + 1].ns;
                counts4[4 * i + 2] = F[2 * i].nr; counts4[4 * i + 3] = F[2 * i + 1].nr;
            }
        }
        return SPH_OK;
    }
    // 1. selection on every device, send counts to the host
    for (int i = 0; i < n; i++) {
        sph_ctx *c = ctxs[i];
        SphComm *m = c ? comm_of(c) : nullptr;
        if (!m) { sph_set_error("sph_halo_exchange: context %d has no communicator (sph_comm_init_rank / _all)", i); return SPH_ERR_STATE; }
        HIP_TRY(hipSetDevice(c->device));
        size_t nn, nreal;
        SPH_TRY(sph_array_size(c, array_id, &nn, &nreal));
        if (drop) SPH_TRY(sph_array_resize(c, array_id, nreal, nreal)); // ghosts of the previous step
        size_t cnt[2];
        SPH_TRY(sph_halo_select(c, array_id, axis, 0, lo[i] + width, hi[i] - width, 0.0, 0, cnt));
        for (int s = 0; s < 2; s++) {
            Face &f = F[2 * i + s];
            f.peer = peer_of(m, s, periodic, period, &f.shift);
            f.ns = f.peer >= 0 ? cnt[s] : 0;
            f.nr = 0;
        }
    }
    // 2. counts handshake: my lo face talks to the peer's hi face and vice versa
    for (int i = 0; i < n; i++) { // allocations and copies BEFORE the group opens
        sph_ctx *c = ctxs[i];
        SphComm *m = comm_of(c);
        HIP_TRY(hipSetDevice(c->device));
        SPH_TRY(m->cnt.reserve(8 * sizeof(unsigned long long)));
        for (int s = 0; s < 2; s++) m->h_cnt[s] = F[2 * i + s].ns;
        HIP_TRY(hipMemcpyAsync(m->cnt.ptr, m->h_cnt, 2 * sizeof(unsigned long long), hipMemcpyHostToDevice, c->stream));
    }
    NCCL_TRY(ncclGroupStart());
    for (int i = 0; i < n; i++) {
        sph_ctx *c = ctxs[i];
        SphComm *m = comm_of(c);
        if (hipSetDevice(c->device) != hipSuccess) { (void)ncclGroupEnd(); sph_set_error("sph_halo_exchange: hipSetDevice failed"); return SPH_ERR_HIP; }
        unsigned long long *d = m->cnt.as<unsigned long long>();
        // between one pair of ranks messages match in posting order: with a periodic
        // axis and <= 2 ranks both faces talk to the SAME peer -- send hi first, receive lo first
        for (int s = 1; s >= 0; s--)
            if (F[2 * i + s].peer >= 0) NCCL_TRY_IN_GROUP(ncclSend(d + s, 1, ncclUint64, F[2 * i + s].peer, m->comm, c->stream));
        for (int s = 0; s < 2; s++)
            if (F[2 * i + s].peer >= 0) NCCL_TRY_IN_GROUP(ncclRecv(d + 2 + s, 1, ncclUint64, F[2 * i + s].peer, m->comm, c->stream));
    }
    NCCL_TRY(ncclGroupEnd());
    for (int i = 0; i < n; i++) {
        sph_ctx *c = ctxs[i];
        SphComm *m = comm_of(c);
        HIP_TRY(hipSetDevice(c->device));
        HIP_TRY(hipMemcpyAsync(m->h_cnt + 2, m->cnt.as<unsigned long long>() + 2, 2 * sizeof(unsigned long long),
                               hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        for (int s = 0; s < 2; s++) F[2 * i + s].nr = F[2 * i + s].peer >= 0 ? (size_t)m->h_cnt[2 + s] : 0;
    }
    // 3. pack, 4. one group of point-to-point transfers
    for (int i = 0; i < n; i++) {
        sph_ctx *c = ctxs[i];
        SphComm *m = comm_of(c);
        HIP_TRY(hipSetDevice(c->device));
        for (int s = 0; s < 2; s++) {
            Face &f = F[2 * i + s];
            SPH_TRY(m->send[s].reserve((f.ns * nprops + 1) * sizeof(double)));
            SPH_TRY(m->recv[s].reserve((f.nr * nprops + 1) * sizeof(double)));
            if (f.ns) SPH_TRY(sph_halo_pack(c, array_id, s, nprops, props, axis, f.shift, m->send[s].ptr));
        }
    }
    NCCL_TRY(ncclGroupStart());
    for (int i = 0; i < n; i++) {
        sph_ctx *c = ctxs[i];
        SphComm *m = comm_of(c);
        if (hipSetDevice(c->device) != hipSuccess) { (void)ncclGroupEnd(); sph_set_error("sph_halo_exchange: hipSetDevice failed"); return SPH_ERR_HIP; }
        for (int s = 1; s >= 0; s--) {
            Face &f = F[2 * i + s];
            if (f.ns) NCCL_TRY_IN_GROUP(ncclSend(m->send[s].ptr, f.ns * nprops, ncclDouble, f.peer, m->comm, c->stream));
        }
        for (int s = 0; s < 2; s++) {
            Face &f = F[2 * i + s];
            if (f.nr) NCCL_TRY_IN_GROUP(ncclRecv(m->recv[s].ptr, f.nr * nprops, ncclDouble, f.peer, m->comm, c->stream));
        }
    }
    NCCL_TRY(ncclGroupEnd());
    // 5. ghosts go behind the real particles (lo side first: deterministic)
    for (int i = 0; i < n; i++) {
        sph_ctx *c = ctxs[i];
        SphComm *m = comm_of(c);
        HIP_TRY(hipSetDevice(c->device));
        for (int s = 0; s < 2; s++) {
            Face &f = F[2 * i + s];
            if (f.nr) SPH_TRY(sph_halo_append(c, array_id, nprops, props, m->recv[s].ptr, f.nr));
            m->cap_send[array_id][s] = halo_capacity(f.ns);
            m->cap_recv[array_id][s] = halo_capacity(f.nr);
        }
        m->cap_known[array_id] = true;
        if (counts4) {
            counts4[4 * i + 0] = F[2 * i].ns; counts4[4 * i + 1] = F[2 * i + 1].ns;
            counts4[4 * i + 2] = F[2 * i].nr; counts4[4 * i + 3] = F[2 * i + 1].nr;
        }
    }
    return SPH_OK;
}

int sph_halo_exchange(sph_ctx *c, int array_id, int axis, double lo, double hi, double width, int periodic, double period,
                      int nprops, const int *props, int drop, size_t *counts4)
{
    return sph_halo_exchange_all(1, &c, array_id, axis, &lo, &hi, width, periodic, period, nprops, props, drop, counts4);
}

// MIN / MAX / SUM of a few doubles over all ranks (dt, dt_cfl, dt_force, bounds, hmax)
int sph_allreduce(sph_ctx *c, double *vals, int nvals, int op)
{
    SphComm *m = c ? comm_of(c) : nullptr;
    if (!m || !vals || nvals < 1 || nvals > 64 || op < 0 || op > 2) { sph_set_error("sph_allreduce: bad arguments / no communicator"); return SPH_ERR_ARG; }
    HIP_TRY(hipSetDevice(c->device));
    SPH_TRY(m->send[0].reserve(64 * sizeof(double)));
    HIP_TRY(hipMemcpyAsync(m->send[0].ptr, vals, nvals * sizeof(double), hipMemcpyHostToDevice, c->stream));
    const ncclRedOp_t ops[3] = {ncclMin, ncclMax, ncclSum};
    NCCL_TRY(ncclAllReduce(m->send[0].ptr, m->send[0].ptr, nvals, ncclDouble, ops[op], m->comm, c->stream));
    HIP_TRY(hipMemcpyAsync(vals, m->send[0].ptr, nvals * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SPH_OK;
}

} // extern "C"
