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
#include "comm_internal.h"
#include "primme_amd.h"

#define NCCL_CHECK(call)                                                              \
   do {                                                                               \
      ncclResult_t r_ = (call);                                                       \
      if (r_ != ncclSuccess) {                                                        \
         fprintf(stderr, "primme_amd: %s failed: %s\n", #call, ncclGetErrorString(r_)); \
         return -43;                                                                  \
      }                                                                               \
   } while (0)

/* PRIMME_AMD_COMM selects the transport when the id is created (rank 0) — the id tells the other ranks:
 *   rccl          : everything on RCCL (an ncclUniqueId; the only form that spans nodes)
 *   ipc           : everything on the peer-to-peer mailboxes (comm_ipc.hip); ranks may share one device
 *   unset / auto  : mailboxes for the <= 32 KB reductions and the neighbour halos, RCCL for the bulk all-gather /
 *                   reduce-scatter (ranks on distinct devices); everything on the mailboxes when ranks share a
 *                   device; RCCL alone when the mailboxes cannot be brought up */
static int comm_mode(void) {
   const char *e = getenv("PRIMME_AMD_COMM");
   if (!e || !*e || !strcmp(e, "auto") || !strcmp(e, "hybrid")) return PA_COMM_HYBRID;
   if (!strcmp(e, "ipc")) return PA_COMM_IPC;
   if (!strcmp(e, "rccl")) return PA_COMM_RCCL;
   fprintf(stderr, "primme_amd: PRIMME_AMD_COMM=%s is not one of rccl | ipc | auto; using auto\n", e);
   return PA_COMM_HYBRID;
}

static int rccl_unique_id(void *id128) {
   ncclUniqueId id;
   NCCL_CHECK(ncclGetUniqueId(&id));
   memcpy(id128, &id, sizeof(id) < 128 ? sizeof(id) : 128);
   return 0;
}
extern "C" int primme_amd_comm_unique_id(void *id128) {
   if (comm_mode() != PA_COMM_RCCL) return pa_ipc_unique_id(id128);
   return rccl_unique_id(id128);
}
/* for a launcher that knows the job: a mailbox id only when the mailboxes can serve it (<= 16 ranks on one node).
 * primme_amd_comm_id_kind_for: the decision alone, without side effects — 1 mailbox id, 0 ncclUniqueId, -43 refused */
extern "C" int primme_amd_comm_id_kind_for(int nranks, int spans_nodes) {
   const int mode = comm_mode();
   const int servable = nranks >= 1 && nranks <= HIPK_XR_MAXRANKS && !spans_nodes;
   if (mode == PA_COMM_RCCL) return 0;
   if (!servable) return mode == PA_COMM_IPC ? -43 : 0;      /* auto: RCCL is what serves it; ipc asked for explicitly: refused */
   return 1;
}
extern "C" int primme_amd_comm_unique_id_for(void *id128, int nranks, int spans_nodes) {
   const int kind = primme_amd_comm_id_kind_for(nranks, spans_nodes);
   if (kind < 0) {
      fprintf(stderr, "primme_amd: PRIMME_AMD_COMM=ipc cannot serve this job (%d ranks%s): the peer-to-peer mailboxes take at most %d ranks "
            "on one node; unset PRIMME_AMD_COMM or set it to rccl\n", nranks, spans_nodes ? ", several nodes" : "", HIPK_XR_MAXRANKS);
      return kind;
   }
   return kind == 1 ? pa_ipc_unique_id(id128) : rccl_unique_id(id128);
}

static int comm_staging(primme_amd_comm *c) {
   HIPK_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
   c->dbuf_cap = 8192;
   HIPK_CHECK(hipMalloc((void **)&c->dbuf, c->dbuf_cap * sizeof(double)));
   HIPK_CHECK(hipHostMalloc((void **)&c->hbuf, c->dbuf_cap * sizeof(double), hipHostMallocMapped));
   return 0;
}

extern "C" int primme_amd_comm_create(primme_amd_comm **out, const void *id128, int rank, int nranks) {
   primme_amd_comm *c = (primme_amd_comm *)calloc(1, sizeof(primme_amd_comm));
   if (!c) return -2;
   c->rank = rank; c->nranks = nranks;
   ncclUniqueId id;
   memset(&id, 0, sizeof(id));
   if (pa_ipc_is_ipc_id(id128) && nranks > HIPK_XR_MAXRANKS) {
      /* every rank sees the same id and the same count: all of them leave here, nobody waits at a rendez-vous */
      fprintf(stderr, "primme_amd: rank %d: the communicator id is one of the peer-to-peer mailbox transport, which serves at most %d ranks (this job has %d); "
            "create the id with PRIMME_AMD_COMM=rccl (or with primme_amd_comm_unique_id_for, which picks RCCL for such a job)\n", rank, HIPK_XR_MAXRANKS, nranks);
      free(c);
      return -43;
   }
   if (pa_ipc_is_ipc_id(id128)) {
      /* the mode of the rank that made the id decides (a stray PRIMME_AMD_COMM on one rank must not split the job) */
      int mode = comm_mode(), modes[HIPK_XR_MAXRANKS];
      if (pa_ipc_attach(&c->ipc, id128, rank, nranks)) { free(c); return -43; }
      if (pa_ipc_host_allgather(c->ipc, &mode, sizeof(int), modes)) { pa_ipc_detach(c->ipc); free(c); return -43; }
      mode = modes[0] == PA_COMM_RCCL ? PA_COMM_HYBRID : modes[0];
      const int gpu_ok = pa_ipc_gpu_ok(c->ipc);
      /* PRIMME_AMD_COMM_STRICT=1: mailboxes that do not map across the devices are an ERROR, not a reason to fall back to RCCL
       * quietly (tests/test_multigpu_rccl.py: a silent fall-back must not pass for the peer-to-peer result) */
      const int strict = getenv("PRIMME_AMD_COMM_STRICT") != NULL && modes[0] != PA_COMM_RCCL;
      if (!gpu_ok && strict)
         fprintf(stderr, "primme_amd: rank %d: PRIMME_AMD_COMM_STRICT: the peer-to-peer mailboxes did not come up across the devices of this job (no fall-back to RCCL)\n", rank);
      if (!gpu_ok && (strict || mode == PA_COMM_IPC || !pa_ipc_distinct_devices(c->ipc))) {
         /* the mailboxes were asked for explicitly, or ranks share a device (RCCL cannot serve that): no fall-back */
         pa_ipc_detach(c->ipc); free(c);
         return -43;
      }
      c->kind = !gpu_ok ? PA_COMM_RCCL
              : (mode == PA_COMM_HYBRID && pa_ipc_distinct_devices(c->ipc) && nranks > 1) ? PA_COMM_HYBRID : PA_COMM_IPC;
      if (c->kind != PA_COMM_IPC) {
         /* RCCL next to (or instead of) the mailboxes: its id travels through the rendez-vous segment */
         char all[HIPK_XR_MAXRANKS][128];
         if (rank == 0 && ncclGetUniqueId(&id) != ncclSuccess) memset(&id, 0, sizeof(id));
         if (pa_ipc_host_allgather(c->ipc, &id, 128, all)) { pa_ipc_detach(c->ipc); free(c); return -43; }
         memcpy(&id, all[0], 128);
         const ncclResult_t nr = ncclCommInitRank(&c->comm, nranks, id, rank);
         if (nr != ncclSuccess) {
            fprintf(stderr, "primme_amd: ncclCommInitRank failed: %s\n", ncclGetErrorString(nr));
            pa_ipc_detach(c->ipc); free(c);
            return -43;
         }
      }
      if (!gpu_ok) { c->boot = c->ipc; c->ipc = NULL; }
   } else {
      c->kind = PA_COMM_RCCL;
      memcpy(&id, id128, sizeof(id) < 128 ? sizeof(id) : 128);
      const ncclResult_t nr = ncclCommInitRank(&c->comm, nranks, id, rank);
      if (nr != ncclSuccess) {
         fprintf(stderr, "primme_amd: ncclCommInitRank failed: %s\n", ncclGetErrorString(nr));
         free(c);
         return -43;
      }
   }
   if (comm_staging(c)) { primme_amd_comm_destroy(c); return -1; }
   if (getenv("PRIMME_AMD_COMM_VERBOSE") && rank == 0)
      fprintf(stderr, "primme_amd: communicator of %d ranks: %s\n", nranks,
            c->kind == PA_COMM_RCCL ? "rccl" : c->kind == PA_COMM_IPC ? "ipc (peer-to-peer mailboxes)" : "mailboxes + rccl");
   *out = c;
   return 0;
}

extern "C" int primme_amd_comm_destroy(primme_amd_comm *c) {
   if (!c) return 0;
   hipStreamSynchronize(c->stream);
   if (c->kind != PA_COMM_IPC) ncclCommDestroy(c->comm);
   if (c->ipc) pa_ipc_detach(c->ipc);
   if (c->boot) pa_ipc_detach(c->boot);
   (void)hipFree(c->dbuf);
   if (c->hbuf) (void)hipHostFree(c->hbuf);
   (void)hipStreamDestroy(c->stream);
   free(c);
   return 0;
}

extern "C" int primme_amd_comm_rank(const primme_amd_comm *c) { return c->rank; }
extern "C" int primme_amd_comm_size(const primme_amd_comm *c) { return c->nranks; }
/* "rccl", "ipc" or "hybrid" */
extern "C" const char *primme_amd_comm_transport(const primme_amd_comm *c) {
   return !c ? "none" : c->kind == PA_COMM_RCCL ? "rccl" : c->kind == PA_COMM_IPC ? "ipc" : "hybrid";
}
/* non-zero after a device-side wait of the peer-to-peer transport ran into its time limit */
extern "C" int primme_amd_comm_error(const primme_amd_comm *c) { return c && c->ipc ? pa_ipc_error(c->ipc) : 0; }

extern "C" int pa_comm_allreduce_device(void *commInfo, double *dbuf, int count, void *hip_stream) {
   primme_amd_comm *c = (primme_amd_comm *)commInfo;
   if (!c) return -43;
   if (c->ipc) {
      hipk_fin_flag nof = {NULL, NULL, 0};
      return pa_ipc_allreduce(c->ipc, (hipStream_t)hip_stream, dbuf, count, NULL, nof);
   }
   NCCL_CHECK(ncclAllReduce(dbuf, dbuf, (size_t)count, ncclDouble, ncclSum, c->comm, (hipStream_t)hip_stream));
   return 0;
}
/* public form: dbuf[0:count) <- sum over the ranks, in place, on `hip_stream` (doubles) */
extern "C" int primme_amd_comm_allreduce(primme_amd_comm *c, void *hip_stream, double *dbuf, int count) {
   return pa_comm_allreduce_device(c, dbuf, count, hip_stream);
}
/* reduction + copy into the context's pinned mirror + completion flag in ONE launch (peer-to-peer transport);
 * returns 1 when the transport cannot (RCCL: the caller publishes with a launch of its own) */
extern "C" int pa_comm_allreduce_publish(void *commInfo, hipk_ctx *ctx, double *dbuf, int count) {
   primme_amd_comm *c = (primme_amd_comm *)commInfo;
   if (!c) return -43;
   if (!c->ipc || count <= 0) return 1;
   double *mh = hipk_mirror_of(ctx, dbuf);
   if (!mh || !ctx->flag_dev || !ctx->spin_wait || !hipk_mirror_of(ctx, dbuf + count - 1)) return 1;
   return pa_ipc_allreduce(c->ipc, ctx->stream, dbuf, count, mh, hipk_next_flag(ctx, dbuf)) ? -43 : 0;
}
/* the context's second-stage launches may reduce across the ranks themselves (hipk_xreduce_arm) */
extern "C" int pa_comm_attach_ctx(void *commInfo, hipk_ctx *ctx) {
   primme_amd_comm *c = (primme_amd_comm *)commInfo;
   static int off = -1;              /* PRIMME_AMD_NO_XREDUCE=1: keep reduction launches separate (A/B knob) */
   if (off < 0) off = getenv("PRIMME_AMD_NO_XREDUCE") != NULL;
   ctx->xr = (c && c->ipc && !off) ? pa_ipc_xreduce(c->ipc) : NULL;
   ctx->xr_armed = 0; ctx->xr_lo = NULL; ctx->xr_count = 0;
   return ctx->xr ? 0 : 1;
}
extern "C" int pa_comm_failed(void *commInfo) { return primme_amd_comm_error((const primme_amd_comm *)commInfo); }

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
      if (hipHostMalloc((void **)&c->hbuf, c->dbuf_cap * sizeof(double), hipHostMallocMapped) != hipSuccess) return;
   }
   /* globalSumReal_type: this callback handles double (the solver always reduces doubles) */
   /* The mailbox tags are one sequence per communicator and a slot is reused two reductions later: reductions must reach the
    * device in the order their tags were taken.  The solver's own reductions run on ITS stream and may still be queued
    * there; this host-buffer path runs on the communicator's stream: drain the device first (this path synchronises anyway). */
   if (c->ipc && hipDeviceSynchronize() != hipSuccess) return;
   memcpy(c->hbuf, sendBuf, n * sizeof(double));
   if (hipMemcpyAsync(c->dbuf, c->hbuf, n * sizeof(double), hipMemcpyHostToDevice, c->stream) != hipSuccess) return;
   if (c->ipc) {
      hipk_fin_flag nof = {NULL, NULL, 0};
      if (pa_ipc_allreduce(c->ipc, c->stream, c->dbuf, (int)n, NULL, nof)) return;
   } else if (ncclAllReduce(c->dbuf, c->dbuf, n, ncclDouble, ncclSum, c->comm, c->stream) != ncclSuccess) return;
   if (hipMemcpyAsync(c->hbuf, c->dbuf, n * sizeof(double), hipMemcpyDeviceToHost, c->stream) != hipSuccess) return;
   if (hipStreamSynchronize(c->stream) != hipSuccess) return;
   if (c->ipc && pa_ipc_error(c->ipc)) return;
   memcpy(recvBuf, c->hbuf, n * sizeof(double));
   *ierr = 0;
}

/* the neighbour exchange with the landing buffers chosen by the transport: RCCL receives into the caller's
 * buffers (lo_buf / hi_buf), the mailboxes have the rows land in their own zones; *lo_out / *hi_out say where.
 * max_side_rows: the largest halo count of ANY rank (the mailboxes size their zones with it, collectively). */
extern "C" int pa_comm_halo_auto(primme_amd_comm *c, void *hip_stream, const void *x, int64_t ldx, int64_t nrows, int ncols,
      size_t elem, int64_t send_lo_cnt, int64_t send_hi_cnt, void *lo_buf, int64_t recv_lo_cnt, void *hi_buf,
      int64_t recv_hi_cnt, int64_t max_side_rows, void **lo_out, void **hi_out) {
   if (c->ipc)
      return pa_ipc_halo(c->ipc, (hipStream_t)hip_stream, x, ldx, nrows, ncols, elem, send_lo_cnt, send_hi_cnt, recv_lo_cnt,
            recv_hi_cnt, (size_t)max_side_rows * ncols * elem, lo_out, hi_out);
   *lo_out = lo_buf; *hi_out = hi_buf;
   return primme_amd_comm_halo(c, hip_stream, x, ldx, nrows, ncols, elem, send_lo_cnt, send_hi_cnt, lo_buf, recv_lo_cnt, hi_buf, recv_hi_cnt);
}

/* Exchange with the two neighbouring ranks: send my first `send_lo_cnt` elements of
 * every column to rank-1 and my last `send_hi_cnt` to rank+1; receive `recv_lo_cnt`
 * from rank-1 into lo and `recv_hi_cnt` from rank+1 into hi.  Columns of x are ldx
 * apart; halo buffers are packed (column stride = count).  elem = bytes/element. */
extern "C" int primme_amd_comm_halo(primme_amd_comm *c, void *hip_stream, const void *x, int64_t ldx,
      int64_t nrows, int ncols, size_t elem, int64_t send_lo_cnt, int64_t send_hi_cnt, void *lo,
      int64_t recv_lo_cnt, void *hi, int64_t recv_hi_cnt) {
   hipStream_t st = (hipStream_t)hip_stream;
   if (c->ipc) {
      /* caller-owned landing buffers: agree on the zone size (a host rendez-vous: this entry point synchronises;
       * the operator uses pa_comm_halo_auto, which does not), exchange, copy out */
      int64_t mine = send_lo_cnt, all[HIPK_XR_MAXRANKS], mx = 0;
      if (send_hi_cnt > mine) mine = send_hi_cnt;
      if (recv_lo_cnt > mine) mine = recv_lo_cnt;
      if (recv_hi_cnt > mine) mine = recv_hi_cnt;
      if (pa_ipc_host_allgather(c->ipc, &mine, sizeof(mine), all)) return -43;
      for (int p = 0; p < c->nranks; p++) if (all[p] > mx) mx = all[p];
      void *zl = NULL, *zh = NULL;
      int rc = pa_ipc_halo(c->ipc, st, x, ldx, nrows, ncols, elem, send_lo_cnt, send_hi_cnt, recv_lo_cnt, recv_hi_cnt,
            (size_t)mx * ncols * elem, &zl, &zh);
      if (rc) return rc;
      if (c->rank > 0 && recv_lo_cnt > 0) HIPK_CHECK(hipMemcpyAsync(lo, zl, (size_t)recv_lo_cnt * ncols * elem, hipMemcpyDeviceToDevice, st));
      if (c->rank < c->nranks - 1 && recv_hi_cnt > 0) HIPK_CHECK(hipMemcpyAsync(hi, zh, (size_t)recv_hi_cnt * ncols * elem, hipMemcpyDeviceToDevice, st));
      return 0;
   }
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
   if (c->kind == PA_COMM_IPC) return pa_ipc_allgather_cols(c->ipc, (hipStream_t)hip_stream, send, 0, recv, 0, bytes_per_rank, bytes_per_rank % 8 ? 4 : 8, 1);
   NCCL_CHECK(ncclAllGather(send, recv, bytes_per_rank, ncclChar, c->comm, (hipStream_t)hip_stream));
   return 0;
}

extern "C" int primme_amd_comm_reduce_scatter(primme_amd_comm *c, void *hip_stream, const void *send,
      void *recv, size_t count_per_rank, int is_double) {
   if (c->kind == PA_COMM_IPC) return pa_ipc_reduce_scatter_cols(c->ipc, (hipStream_t)hip_stream, send, 0, recv, 0, count_per_rank, is_double, 1);
   NCCL_CHECK(ncclReduceScatter(send, recv, count_per_rank, is_double ? ncclDouble : ncclFloat, ncclSum, c->comm,
         (hipStream_t)hip_stream));
   return 0;
}

/* block forms: column c of the send panel (ld_send elements apart) -> column c of the receive panel.
 * The calls are issued inside one RCCL group, i.e. one launch for the whole block. */
extern "C" int primme_amd_comm_allgather_cols(primme_amd_comm *c, void *hip_stream, const void *send, int64_t ld_send,
      void *recv, int64_t ld_recv, size_t bytes_per_rank, size_t elem, int ncols) {
   if (c->kind == PA_COMM_IPC) return pa_ipc_allgather_cols(c->ipc, (hipStream_t)hip_stream, send, ld_send, recv, ld_recv, bytes_per_rank, elem, ncols);
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
   if (c->kind == PA_COMM_IPC) return pa_ipc_reduce_scatter_cols(c->ipc, (hipStream_t)hip_stream, send, ld_send, recv, ld_recv, count_per_rank, is_double, ncols);
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
   if (c->ipc || c->boot) {
      pa_ipc *hx = c->ipc ? c->ipc : c->boot;
      /* through the rendez-vous segment, PA_IPC_PAYLOAD bytes per rank and round */
      const int per = PA_IPC_PAYLOAD / (int)sizeof(int64_t);
      int64_t tmp[HIPK_XR_MAXRANKS * (PA_IPC_PAYLOAD / sizeof(int64_t))];
      for (int i0 = 0; i0 < n; i0 += per) {
         const int k = n - i0 < per ? n - i0 : per;
         if (pa_ipc_host_allgather(hx, mine + i0, (size_t)k * sizeof(int64_t), tmp)) return -43;
         for (int p = 0; p < c->nranks; p++)
            for (int i = 0; i < k; i++) all[(size_t)p * n + i0 + i] = tmp[(size_t)p * k + i];
      }
      return 0;
   }
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

/* ---- self-test: what a launcher runs before it trusts a communicator it has never used on this machine ---------------
 * Every collective of this header with known data on whatever transport the communicator came up on: all-reduce of
 * 1 .. 4096 doubles (rank-ordered sums on the mailboxes: exact; RCCL: to rounding), the neighbour halo, the bulk
 * all-gather and reduce-scatter of a block of columns, the integer exchange, then `reps` back-to-back 8-double
 * all-reduces for the latency a block-size-1 iteration pays per reduction.  Collective: every rank calls it.
 * Returns 0 when everything matched on this rank, a positive count of mismatching checks otherwise (negative: an error of
 * the transport); *allreduce_us (optional) = wall-clock microseconds per all-reduce of the timed loop. */
static double st_now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
static double st_val(int rank, int i, int salt) { return (double)(rank + 1) + 1e-3 * (double)i + 0.25 * (double)salt; }
extern "C" int primme_amd_comm_selftest(primme_amd_comm *c, void *hip_stream, int reps, double *allreduce_us) {
   if (!c) return -43;
   hipStream_t st = hip_stream ? (hipStream_t)hip_stream : c->stream;
   const int P = c->nranks, me = c->rank, NMAX = 4096, per = 257, nc = 2;
   const size_t words = (size_t)NMAX + (size_t)per * P * nc * 2 + 64;
   double *d = NULL, *h = NULL;
   if (hipMalloc((void **)&d, words * sizeof(double)) != hipSuccess) return -2;
   if (hipHostMalloc((void **)&h, words * sizeof(double), hipHostMallocDefault) != hipSuccess) { (void)hipFree(d); return -2; }
   int bad = 0, rc = 0;
#define ST_UP(n) (hipMemcpyAsync(d, h, (size_t)(n) * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess)
#define ST_DOWN(n) (hipMemcpyAsync(h, d, (size_t)(n) * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
   const int sizes[] = {1, 2, 15, 16, 17, 300, 4096};
   for (int s = 0; s < 7 && !rc; s++) {
      const int n = sizes[s];
      for (int i = 0; i < n; i++) h[i] = st_val(me, i, s);
      if (ST_UP(n)) { rc = -1; break; }
      rc = primme_amd_comm_allreduce(c, st, d, n);
      if (rc || ST_DOWN(n)) { rc = rc ? rc : -1; break; }
      for (int i = 0; i < n; i++) {
         double want = 0.0;
         for (int p = 0; p < P; p++) want += st_val(p, i, s);        /* rank order: what the mailboxes produce bit for bit */
         if (!(fabs(h[i] - want) <= 1e-12 * fabs(want))) { bad++; break; }
      }
   }
   /* neighbour halo: 3 rows down, 2 rows up, two columns */
   if (!rc && P > 1) {
      const int nrows = 40, ld = 48, lo_n = 2, hi_n = 3;          /* I receive 2 rows from below (rank-1's last 2) and 3 from above */
      for (int col = 0; col < nc; col++) for (int i = 0; i < nrows; i++) h[col * ld + i] = st_val(me, i, 100 + col);
      double *dx = d, *dlo = d + nc * ld, *dhi = dlo + nc * lo_n;
      if (hipMemcpyAsync(dx, h, (size_t)nc * ld * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess) rc = -1;
      /* what rank-1 needs from ME is my first hi_n rows (they are ITS rows from above), what rank+1 needs my last lo_n */
      if (!rc) rc = primme_amd_comm_halo(c, st, dx, ld, nrows, nc, sizeof(double), hi_n, lo_n, dlo, lo_n, dhi, hi_n);
      if (!rc && (hipMemcpyAsync(h, dlo, (size_t)nc * (lo_n + hi_n) * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess ||
                  hipStreamSynchronize(st) != hipSuccess)) rc = -1;
      if (!rc) {
         const double *glo = h, *ghi = h + nc * lo_n;
         for (int col = 0; col < nc; col++) {
            if (me > 0) for (int i = 0; i < lo_n; i++) if (glo[col * lo_n + i] != st_val(me - 1, nrows - lo_n + i, 100 + col)) { bad++; break; }
            if (me < P - 1) for (int i = 0; i < hi_n; i++) if (ghi[col * hi_n + i] != st_val(me + 1, i, 100 + col)) { bad++; break; }
         }
      }
   }
   /* bulk window: all-gather and reduce-scatter of a block of columns */
   if (!rc) {
      double *dsend = d, *drecv = d + (size_t)per * P * nc;
      for (int col = 0; col < nc; col++) for (int i = 0; i < per; i++) h[col * per + i] = st_val(me, i, 200 + col);
      if (hipMemcpyAsync(dsend, h, (size_t)nc * per * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess) rc = -1;
      if (!rc) rc = primme_amd_comm_allgather_cols(c, st, dsend, per, drecv, (int64_t)per * P, (size_t)per * sizeof(double), sizeof(double), nc);
      if (!rc && (hipMemcpyAsync(h, drecv, (size_t)nc * per * P * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess ||
                  hipStreamSynchronize(st) != hipSuccess)) rc = -1;
      if (!rc) for (int col = 0; col < nc && !bad; col++) for (int p = 0; p < P; p++) for (int i = 0; i < per; i++)
         if (h[(size_t)col * per * P + (size_t)p * per + i] != st_val(p, i, 200 + col)) { bad++; p = P; break; }
      for (int col = 0; col < nc; col++) for (int i = 0; i < per * P; i++) h[(size_t)col * per * P + i] = st_val(me, i, 300 + col);
      if (!rc && hipMemcpyAsync(dsend, h, (size_t)nc * per * P * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess) rc = -1;
      if (!rc) rc = primme_amd_comm_reduce_scatter_cols(c, st, dsend, (int64_t)per * P, drecv, per, per, 1, nc);
      if (!rc && (hipMemcpyAsync(h, drecv, (size_t)nc * per * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess ||
                  hipStreamSynchronize(st) != hipSuccess)) rc = -1;
      if (!rc) for (int col = 0; col < nc; col++) for (int i = 0; i < per; i++) {
         double want = 0.0;
         for (int p = 0; p < P; p++) want += st_val(p, me * per + i, 300 + col);
         if (!(fabs(h[col * per + i] - want) <= 1e-12 * fabs(want))) { bad++; break; }
      }
   }
   if (!rc) {
      int64_t mine[2] = {7 + me, -(int64_t)me}, all[2 * HIPK_XR_MAXRANKS > 64 ? 2 * HIPK_XR_MAXRANKS : 64];
      if (P <= 32) {
         rc = primme_amd_comm_allgather_i64(c, mine, 2, all);
         if (!rc) for (int p = 0; p < P; p++) if (all[2 * p] != 7 + p || all[2 * p + 1] != -(int64_t)p) { bad++; break; }
      }
   }
   /* latency: back-to-back 8-double all-reduces (a block-size-1 iteration makes three) */
   if (!rc && reps > 0) {
      for (int i = 0; i < 8; i++) h[i] = 1.0;
      if (ST_UP(8) || hipStreamSynchronize(st) != hipSuccess) rc = -1;
      for (int w = 0; w < 10 && !rc; w++) rc = primme_amd_comm_allreduce(c, st, d, 8);
      if (!rc && hipStreamSynchronize(st) != hipSuccess) rc = -1;
      const double t0 = st_now();
      for (int r = 0; r < reps && !rc; r++) rc = primme_amd_comm_allreduce(c, st, d, 8);
      if (!rc && hipStreamSynchronize(st) != hipSuccess) rc = -1;
      if (!rc && allreduce_us) *allreduce_us = 1e6 * (st_now() - t0) / reps;
   }
#undef ST_UP
#undef ST_DOWN
   if (!rc && primme_amd_comm_error(c)) rc = -43;
   (void)hipHostFree(h);
   (void)hipFree(d);
   if (rc) { fprintf(stderr, "primme_amd: communicator self-test: rank %d: transport error %d\n", me, rc); return rc < 0 ? rc : -43; }
   if (bad) fprintf(stderr, "primme_amd: communicator self-test: rank %d: %d check(s) did not match (transport %s)\n", me, bad, primme_amd_comm_transport(c));
   return bad;
}
