/* comm_internal.h — shared by comm_rccl.hip (the communicator object and its RCCL transport) and
 * comm_ipc.hip (the one-shot peer-to-peer transport over hipIpc-mapped mailboxes). */
#ifndef PA_COMM_INTERNAL_H
#define PA_COMM_INTERNAL_H

#include "hipk_internal.h"
#include "primme_amd_comm.h"
#include <rccl/rccl.h>

/* ---- the peer-to-peer transport (comm_ipc.hip) -------------------------------------------------------
 * Every rank exports two allocations through hipIpcGetMemHandle and maps those of all peers:
 *   mailbox : 8-byte {data, tag} granules for the <= 64 KB reductions, barrier words, halo flags, and
 *             the halo landing zones (two generations);
 *   window  : bulk landing zone of the all-gather / reduce-scatter (two generations), grown on demand.
 * Rendez-vous (handles, sizes, the optional RCCL id) goes through a POSIX shared-memory segment named
 * by the 128-byte id: one node, which is the scope of the row partition (8 GPUs on xGMI). */
struct pa_ipc;

int pa_ipc_unique_id(void *id128);
int pa_ipc_is_ipc_id(const void *id128);
int pa_ipc_attach(pa_ipc **out, const void *id128, int rank, int nranks);
void pa_ipc_detach(pa_ipc *x);
/* host-side exchange through the shared segment: `bytes` (<= PA_IPC_PAYLOAD) from every rank, in rank order */
#define PA_IPC_PAYLOAD 256
int pa_ipc_host_allgather(pa_ipc *x, const void *mine, size_t bytes, void *all);
/* 1 when every rank sits on its own device (RCCL can be brought up next to the mailboxes) */
int pa_ipc_distinct_devices(pa_ipc *x);
/* 0: only the host rendez-vous of this object works (the mailboxes did not come up on every rank) */
int pa_ipc_gpu_ok(pa_ipc *x);
/* in place: dbuf[0:count) <- sum over ranks, summed in rank order on every rank (identical bits everywhere).
 * mirror (device address of pinned host memory, may be NULL) receives a copy; `fin` (flag == NULL: none) is
 * published after the results are visible to the host: reduction + publication = ONE launch. */
int pa_ipc_allreduce(pa_ipc *x, hipStream_t st, double *dbuf, int count, double *mirror, hipk_fin_flag fin);
/* neighbour exchange into the mailbox' landing zones; *lo_out / *hi_out = where the rows from rank-1 / rank+1
 * landed (valid until the next-but-one exchange).  max_side_bytes: the largest ncols*count*elem of ANY rank
 * (identical on all ranks: it sizes the zones, and growing them is collective). */
int pa_ipc_halo(pa_ipc *x, hipStream_t st, const void *xv, int64_t ldx, int64_t nrows, int ncols, size_t elem,
      int64_t send_lo_cnt, int64_t send_hi_cnt, int64_t recv_lo_cnt, int64_t recv_hi_cnt, size_t max_side_bytes,
      void **lo_out, void **hi_out);
int pa_ipc_allgather_cols(pa_ipc *x, hipStream_t st, const void *send, int64_t ld_send, void *recv, int64_t ld_recv,
      size_t bytes_per_rank, size_t elem, int ncols);
int pa_ipc_reduce_scatter_cols(pa_ipc *x, hipStream_t st, const void *send, int64_t ld_send, void *recv, int64_t ld_recv,
      size_t count_per_rank, int is_double, int ncols);
/* non-zero after a device-side wait ran into its time limit (a peer died or left the collective sequence) */
int pa_ipc_error(pa_ipc *x);
/* the pieces the fused second stage of the reductions needs (hipk_core.hip): see hipk_xreduce in hipk_internal.h */
hipk_xreduce *pa_ipc_xreduce(pa_ipc *x);

/* ---- the communicator ------------------------------------------------------------------------------- */
enum { PA_COMM_RCCL = 0, PA_COMM_IPC = 1, PA_COMM_HYBRID = 2 };
struct primme_amd_comm {
   int kind;               /* PA_COMM_RCCL: everything on RCCL; PA_COMM_IPC: everything on the mailboxes (ranks may share a
                              device); PA_COMM_HYBRID: reductions and neighbour halos on the mailboxes, bulk collectives on RCCL */
   ncclComm_t comm;        /* RCCL and HYBRID */
   int rank, nranks;
   hipStream_t stream;     /* for the host-buffer callback path */
   double *dbuf;           /* staging for the host-buffer path */
   double *hbuf;           /* its pinned host twin: the caller's (pageable) buffers never meet an asynchronous copy */
   size_t dbuf_cap;
   pa_ipc *ipc;            /* IPC and HYBRID: the mailboxes.  RCCL created from a mailbox id whose device side failed: NULL, the
                              rendez-vous object lives on in `boot` for the host-side exchanges */
   pa_ipc *boot;
};


extern "C" int pa_comm_halo_auto(primme_amd_comm *c, void *hip_stream, const void *x, int64_t ldx, int64_t nrows, int ncols,
      size_t elem, int64_t send_lo_cnt, int64_t send_hi_cnt, void *lo_buf, int64_t recv_lo_cnt, void *hi_buf,
      int64_t recv_hi_cnt, int64_t max_side_rows, void **lo_out, void **hi_out);

#endif
