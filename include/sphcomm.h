/* sphcomm.h -- C ABI of libsphcomm.so: the ghost-halo transport on RCCL for
 * hosts that do not have torch.distributed (C, C++, Fortran, Julia ...).
 *
 * The Python host of this repository reaches RCCL through torch.distributed
 * (pysph_amd/parallel.py: SlabHalo / SlabDecomposition); these entry points are
 * the same exchange behind plain pointers, for the binding a non-Python
 * reference-side caller would write.  They replace, on one node,
 *   ParallelManager.update's ghost refresh      pysph/parallel/parallel_manager.pyx:512-530
 *   remote_exchange_data (one Comm_Do per prop)  pysph/parallel/parallel_manager.pyx:159-210
 *   update_time_steps / _compute_bounds          pysph/parallel/parallel_manager.pyx:463, 937-945
 * with, on the context's stream and point-to-point over xGMI: the first exchange
 * of an array -- select (device) -> counts to the <= 2 slab neighbours -> pack
 * (device) -> ncclGroupStart; ncclSend / ncclRecv; ncclGroupEnd -> append
 * (device); every later one -- selection + packing of both faces in one device
 * pass into fixed-capacity messages sized from the count both ends saw last
 * (sph_halo_select_pack; the row count travels in the message header) -> one
 * NCCL group -> ONE small device->host copy of the headers -> append; a face
 * that outgrew its capacity is repeated with the exact size.  The same
 * protocol as pysph_amd/parallel.py (SPH_HALO_PROTOCOL=handshake forces the
 * first form).
 *
 * Link: -lsphcomm -lsphhip (libsphcomm.so links librccl itself).  All functions
 * return SPH_OK (0) or a negative SPH_ERR_* code; sph_last_error() of
 * libsphhip.so has the message.  Call sph_comm_destroy before sph_ctx_destroy.
 */
#ifndef SPHCOMM_H
#define SPHCOMM_H

#include "sphhip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* 128-byte rendezvous token (ncclUniqueId): rank 0 creates it and hands it to
 * the other ranks by whatever means the host has (MPI_Bcast, a file, a socket). */
int sph_comm_unique_id(void *id128);
/* One process per GPU (the model of this backend): attach a communicator of
 * `world` ranks to the context.                                              */
int sph_comm_init_rank(sph_ctx *ctx, int rank, int world, const void *id128);
/* One process driving `ndev` GPUs: contexts i = 0..ndev-1 become ranks 0..ndev-1
 * (ncclCommInitAll).  Use sph_halo_exchange_all with these.                   */
int sph_comm_init_all(int ndev, sph_ctx **ctxs);
int sph_comm_destroy(sph_ctx *ctx);

/* Refresh the ghosts of `array_id` in a 1-D slab decomposition along `axis`:
 * this rank owns [lo, hi); particles within `width` of a face go to the
 * neighbouring rank (rank-1 / rank+1; with `periodic`, rank 0 and world-1 are
 * neighbours and what crosses that face is shifted by -+`period`,
 * pysph/base/nnps_base.pyx:841-856).  `props[nprops]`: the property ids a ghost
 * carries (sph_prop_id: x y z u v w rho h m = 72 B for WCSPH).  drop != 0
 * removes the ghosts of the previous call first.  Received ghosts are appended
 * behind the real particles (tag Remote semantics: sources only).
 * counts4 (optional): sent lo, sent hi, received lo, received hi.            */
int sph_halo_exchange(sph_ctx *ctx, int array_id, int axis, double lo, double hi, double width, int periodic,
                      double period, int nprops, const int *props, int drop, size_t *counts4);
/* The same for n contexts of ONE process (sph_comm_init_all): lo[n], hi[n],
 * counts4[4 n]; every device's sends and receives sit in one NCCL group.     */
int sph_halo_exchange_all(int n, sph_ctx **ctxs, int array_id, int axis, const double *lo, const double *hi,
                          double width, int periodic, double period, int nprops, const int *props, int drop,
                          size_t *counts4);
/* One NCCL group of point-to-point transfers of DEVICE doubles on the context's
 * stream: nsend sends (buffer, count, peer rank), then nrecv receives -- the
 * batch of ParallelManager.remote_exchange_data's transfers
 * (pysph/parallel/parallel_manager.pyx:159-210) for a caller that packs and
 * appends itself (pysph_amd/parallel.py with SPH_HALO_TRANSPORT=sphcomm: the
 * round-trip-free exchange without torch.distributed's stream hand-over).     */
int sph_comm_sendrecv(sph_ctx *ctx, int nsend, const void *const *sendbufs, const size_t *sendcounts,
                      const int *sendpeers, int nrecv, void *const *recvbufs, const size_t *recvcounts,
                      const int *recvpeers);
/* In-place MIN (op 0) / MAX (1) / SUM (2) of nvals <= 64 host doubles over all
 * ranks: the time-step and bounds reductions.                                */
int sph_allreduce(sph_ctx *ctx, double *vals, int nvals, int op);

#ifdef __cplusplus
}
#endif
#endif
