/* comm_rccl.hip — the communication backend of the row-partitioned solve.
 *
 * The reference has no communication code of its own: the application supplies
 * globalSumReal / broadcastReal (reference include/primme_eigs.h:192-198,
 * examples/ex_eigs_mpi.c:209-218 wraps MPI_Allreduce).  Here the library ships
 * that callback, implemented on RCCL over xGMI:
 *   - primme_amd_global_sum : the callback with the reference's contract
 *     (host buffers, send may equal recv);
 *   - pa_comm_allreduce_device : what the solver uses instead when it sees this
 *     callback installed — the <= 4 KB partial results are already in HBM, so
 *     they are reduced in place on the solver's stream (no host round trip);
 *   - halo exchange for the distributed matvec (grouped ncclSend/ncclRecv with
 *     the two neighbours, or an all-gather for unstructured column patterns).
 * One process per GPU; the ncclUniqueId is created on rank 0 and distributed by
 * the launcher (bench.py uses torch.distributed's store for the 128 bytes).
 * All broadcasts of the reference disappear: every rank solves the identical
 * small projected problem deterministically.
 */
#include "hipk_internal.h"
#include "primme_amd.h"
#include "primme_amd_comm.h"
#include <rccl/rccl.h>

struct primme_amd_comm {
   ncclComm_t comm;
   int rank, nranks;
   hipStream_t stream;     /* for the host-buffer callback path */
   double *dbuf;           /* staging for the host-buffer path */
   double *hbuf;           /* its pinned host twin: the caller's (pageable) buffers never meet an asynchronous copy */
   size_t dbuf_cap;
};

#define NCCL_CHECK(call)                                                              \
   do {                                                                               \
      ncclResult_t r_ = (call);                                                       \
      if (r_ != ncclSuccess) {                                                        \
         fprintf(stderr, "primme_amd: %s failed: %s\n", #call, ncclGetErrorString(r_)); \
         return -43;                                                                  \
      }                                                                               \
   } while (0)

extern "C" int primme_amd_comm_unique_id(void *id128) {
   ncclUniqueId id;
   NCCL_CHECK(ncclGetUniqueId(&id));
   memcpy(id128, &id, sizeof(id) < 128 ? sizeof(id) : 128);
   return 0;
}

extern "C" int primme_amd_comm_create(primme_amd_comm **out, const void *id128, int rank, int nranks) {
   primme_amd_comm *c = (primme_amd_comm *)calloc(1, sizeof(primme_amd_comm));
   if (!c) return -2;
   ncclUniqueId id;
   memset(&id, 0, sizeof(id));
   memcpy(&id, id128, sizeof(id) < 128 ? sizeof(id) : 128);
   c->rank = rank; c->nranks = nranks;
   NCCL_CHECK(ncclCommInitRank(&c->comm, nranks, id, rank));
   HIPK_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
   c->dbuf_cap = 8192;
   HIPK_CHECK(hipMalloc((void **)&c->dbuf, c->dbuf_cap * sizeof(double)));
   HIPK_CHECK(hipHostMalloc((void **)&c->hbuf, c->dbuf_cap * sizeof(double), hipHostMallocDefault));
   *out = c;
   return 0;
}

extern "C" int primme_amd_comm_destroy(primme_amd_comm *c) {
   if (!c) return 0;
   hipStreamSynchronize(c->stream);
   ncclCommDestroy(c->comm);
   (void)hipFree(c->dbuf);
   if (c->hbuf) (void)hipHostFree(c->hbuf);
   (void)hipStreamDestroy(c->stream);
   free(c);
   return 0;
}

extern "C" int primme_amd_comm_rank(const primme_amd_comm *c) { return c->rank; }
extern "C" int primme_amd_comm_size(const primme_amd_comm *c) { return c->nranks; }

extern "C" int pa_comm_allreduce_device(void *commInfo, double *dbuf, int count, void *hip_stream) {
   primme_amd_comm *c = (primme_amd_comm *)commInfo;
   if (!c) return -43;
   NCCL_CHECK(ncclAllReduce(dbuf, dbuf, (size_t)count, ncclDouble, ncclSum, c->comm, (hipStream_t)hip_stream));
   return 0;
}

extern "C" void primme_amd_global_sum(void *sendBuf, void *recvBuf, int *count,
      struct primme_params *primme, int *ierr) {
   primme_amd_comm *c = (primme_amd_comm *)primme->commInfo;
   *ierr = 1;
   if (!c || *count < 0) return;
   const size_t n = (size_t)*count;
   if (n == 0) { *ierr = 0; return; }
   if (n > c->dbuf_cap) {
      (void)hipFree(c->dbuf);
      if (c->hbuf) (void)hipHostFree(c->hbuf);
      c->hbuf = NULL;
      c->dbuf_cap = 2 * n;
      if (hipMalloc((void **)&c->dbuf, c->dbuf_cap * sizeof(double)) != hipSuccess) return;
      if (hipHostMalloc((void **)&c->hbuf, c->dbuf_cap * sizeof(double), hipHostMallocDefault) != hipSuccess) return;
   }
   /* globalSumReal_type: this callback handles double (the solver always reduces doubles) */
   memcpy(c->hbuf, sendBuf, n * sizeof(double));
   if (hipMemcpyAsync(c->dbuf, c->hbuf, n * sizeof(double), hipMemcpyHostToDevice, c->stream) != hipSuccess) return;
   if (ncclAllReduce(c->dbuf, c->dbuf, n, ncclDouble, ncclSum, c->comm, c->stream) != ncclSuccess) return;
   if (hipMemcpyAsync(c->hbuf, c->dbuf, n * sizeof(double), hipMemcpyDeviceToHost, c->stream) != hipSuccess) return;
   if (hipStreamSynchronize(c->stream) != hipSuccess) return;
   memcpy(recvBuf, c->hbuf, n * sizeof(double));
   *ierr = 0;
}

/* Exchange with the two neighbouring ranks: send my first `send_lo_cnt` elements of
 * every column to rank-1 and my last `send_hi_cnt` to rank+1; receive `recv_lo_cnt`
 * from rank-1 into lo and `recv_hi_cnt` from rank+1 into hi.  Columns of x are ldx
 * apart; halo buffers are packed (column stride = count).  elem = bytes/element. */
extern "C" int primme_amd_comm_halo(primme_amd_comm *c, void *hip_stream, const void *x, int64_t ldx,
      int64_t nrows, int ncols, size_t elem, int64_t send_lo_cnt, int64_t send_hi_cnt, void *lo,
      int64_t recv_lo_cnt, void *hi, int64_t recv_hi_cnt) {
   hipStream_t st = (hipStream_t)hip_stream;
   NCCL_CHECK(ncclGroupStart());
   for (int col = 0; col < ncols; col++) {
      const char *xc = (const char *)x + (size_t)col * ldx * elem;
      if (c->rank > 0) {
         if (send_lo_cnt > 0) NCCL_CHECK(ncclSend(xc, (size_t)send_lo_cnt * elem, ncclChar, c->rank - 1, c->comm, st));
         if (recv_lo_cnt > 0) NCCL_CHECK(ncclRecv((char *)lo + (size_t)col * recv_lo_cnt * elem, (size_t)recv_lo_cnt * elem, ncclChar, c->rank - 1, c->comm, st));
      }
      if (c->rank < c->nranks - 1) {
         if (send_hi_cnt > 0) NCCL_CHECK(ncclSend(xc + (size_t)(nrows - send_hi_cnt) * elem, (size_t)send_hi_cnt * elem, ncclChar, c->rank + 1, c->comm, st));
         if (recv_hi_cnt > 0) NCCL_CHECK(ncclRecv((char *)hi + (size_t)col * recv_hi_cnt * elem, (size_t)recv_hi_cnt * elem, ncclChar, c->rank + 1, c->comm, st));
      }
   }
   NCCL_CHECK(ncclGroupEnd());
   return 0;
}

/* all-gather of equal-sized slabs (used for unstructured column patterns) */
extern "C" int primme_amd_comm_allgather(primme_amd_comm *c, void *hip_stream, const void *send,
      void *recv, size_t bytes_per_rank) {
   NCCL_CHECK(ncclAllGather(send, recv, bytes_per_rank, ncclChar, c->comm, (hipStream_t)hip_stream));
   return 0;
}

extern "C" int primme_amd_comm_reduce_scatter(primme_amd_comm *c, void *hip_stream, const void *send,
      void *recv, size_t count_per_rank, int is_double) {
   NCCL_CHECK(ncclReduceScatter(send, recv, count_per_rank, is_double ? ncclDouble : ncclFloat, ncclSum, c->comm,
         (hipStream_t)hip_stream));
   return 0;
}

/* block forms: column c of the send panel (ld_send elements apart) -> column c of the receive panel.
 * The calls are issued inside one RCCL group, i.e. one launch for the whole block. */
extern "C" int primme_amd_comm_allgather_cols(primme_amd_comm *c, void *hip_stream, const void *send, int64_t ld_send,
      void *recv, int64_t ld_recv, size_t bytes_per_rank, size_t elem, int ncols) {
   NCCL_CHECK(ncclGroupStart());
   for (int col = 0; col < ncols; col++)
      NCCL_CHECK(ncclAllGather((const char *)send + (size_t)col * ld_send * elem, (char *)recv + (size_t)col * ld_recv * elem,
            bytes_per_rank, ncclChar, c->comm, (hipStream_t)hip_stream));
   NCCL_CHECK(ncclGroupEnd());
   return 0;
}
extern "C" int primme_amd_comm_reduce_scatter_cols(primme_amd_comm *c, void *hip_stream, const void *send, int64_t ld_send,
      void *recv, int64_t ld_recv, size_t count_per_rank, int is_double, int ncols) {
   const size_t elem = is_double ? 8 : 4;
   NCCL_CHECK(ncclGroupStart());
   for (int col = 0; col < ncols; col++)
      NCCL_CHECK(ncclReduceScatter((const char *)send + (size_t)col * ld_send * elem, (char *)recv + (size_t)col * ld_recv * elem,
            count_per_rank, is_double ? ncclDouble : ncclFloat, ncclSum, c->comm, (hipStream_t)hip_stream));
   NCCL_CHECK(ncclGroupEnd());
   return 0;
}

/* globalSumReal with the primme_svds signature (host buffers), same communicator */
#include "primme_amd_svds.h"
extern "C" void primme_amd_svds_global_sum(void *sendBuf, void *recvBuf, int *count,
      struct primme_svds_params *ps, int *ierr) {
   struct primme_params tmp;
   memset(&tmp, 0, sizeof(tmp));
   tmp.commInfo = ps->commInfo;
   primme_amd_global_sum(sendBuf, recvBuf, count, &tmp, ierr);
}

/* small integer exchange at set-up time (neighbour halo sizes) */
extern "C" int primme_amd_comm_allgather_i64(primme_amd_comm *c, const int64_t *mine, int n, int64_t *all) {
   int64_t *d = NULL;
   HIPK_CHECK(hipMalloc((void **)&d, (size_t)(c->nranks + 1) * n * sizeof(int64_t)));
   int64_t *h = NULL;      /* pinned twin of the exchange buffer */
   HIPK_CHECK(hipHostMalloc((void **)&h, (size_t)(c->nranks + 1) * n * sizeof(int64_t), hipHostMallocDefault));
   memcpy(h + (size_t)c->nranks * n, mine, (size_t)n * sizeof(int64_t));
   HIPK_CHECK(hipMemcpyAsync(d + (size_t)c->nranks * n, h + (size_t)c->nranks * n, (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
   NCCL_CHECK(ncclAllGather(d + (size_t)c->nranks * n, d, (size_t)n * sizeof(int64_t), ncclChar, c->comm, c->stream));
   HIPK_CHECK(hipMemcpyAsync(h, d, (size_t)c->nranks * n * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
   HIPK_CHECK(hipStreamSynchronize(c->stream));
   memcpy(all, h, (size_t)c->nranks * n * sizeof(int64_t));
   (void)hipHostFree(h);
   (void)hipFree(d);
   return 0;
}
