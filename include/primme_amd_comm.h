/* primme_amd_comm.h — C-ABI of the communicator and of the distributed operator
 * handle behind the ready-made callbacks of primme_amd.h.
 *
 * Replaces what every application of the reference writes by hand around
 * MPI: the globalSumReal callback (reference include/primme_eigs.h:192-195,
 * examples/ex_eigs_mpi.c:209-218) and the row-partitioned matvec with its
 * neighbour exchange (examples/ex_eigs_mpi.c:150-207).
 */
#ifndef PRIMME_AMD_COMM_H
#define PRIMME_AMD_COMM_H

#include <stddef.h>
#include <stdint.h>
#include "primme_amd_kernels.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct primme_amd_comm primme_amd_comm;

/* rank 0 creates the 128-byte id; the launcher distributes it to all ranks.
 * Transports (environment variable PRIMME_AMD_COMM, read by the rank that creates the id):
 *   rccl : every collective on RCCL (the id is an ncclUniqueId; the only form that spans nodes);
 *   ipc  : every collective on the peer-to-peer mailboxes of csrc/comm_ipc.hip — device memory exported with
 *          hipIpcGetMemHandle and written by the peers directly (xGMI, or the same device when ranks share one);
 *          the id names a POSIX shared-memory segment (one node); at most 16 ranks;
 *   auto (default) : the <= 32 KB reductions (reference call sites src/eigs/ortho.c:249, :290,
 *          update_projection.c:136) and the neighbour halos on the mailboxes — one launch per reduction, the
 *          local second stage included — and the bulk all-gather / reduce-scatter on RCCL; "ipc" when ranks
 *          share a device.
 * Sums are formed in rank order on every rank: all ranks hold identical bits. */
int primme_amd_comm_unique_id(void *id128);
/* The same for a launcher that knows the shape of the job: `auto` then hands out a mailbox id only when the mailboxes can
 * serve it — at most 16 ranks, all on ONE node (one /dev/shm) — and an ncclUniqueId otherwise, so that default settings
 * keep working for jobs that span nodes or have more ranks.  An explicit PRIMME_AMD_COMM=ipc on a job the mailboxes cannot
 * serve fails here (-43, with a message) instead of at the rendez-vous. */
int primme_amd_comm_unique_id_for(void *id128, int nranks, int spans_nodes);
/* the decision alone, without creating anything: 1 = a mailbox id, 0 = an ncclUniqueId, -43 = refused */
int primme_amd_comm_id_kind_for(int nranks, int spans_nodes);
/* Fails fast (-43 and a message saying what to set), the same way on every rank, when a mailbox id meets more than 16 ranks; a
 * rank that cannot see the rendez-vous segment (another node / IPC namespace) returns -43 at once and the ranks that can
 * see it give up after PRIMME_AMD_IPC_ATTACH_TIMEOUT_S seconds (default 60; the later rendez-vous keep
 * PRIMME_AMD_IPC_TIMEOUT_S, default 300) with the same advice: PRIMME_AMD_COMM=rccl. */
int primme_amd_comm_create(primme_amd_comm **comm, const void *id128, int rank, int nranks);
int primme_amd_comm_destroy(primme_amd_comm *comm);
int primme_amd_comm_rank(const primme_amd_comm *comm);
int primme_amd_comm_size(const primme_amd_comm *comm);
const char *primme_amd_comm_transport(const primme_amd_comm *comm);     /* "rccl", "ipc" or "hybrid" */
/* non-zero after a device-side wait of the peer-to-peer transport ran into its time limit
 * (PRIMME_AMD_IPC_DEVICE_TIMEOUT_S, default 60): a rank died or left the collective call sequence */
int primme_amd_comm_error(const primme_amd_comm *comm);
/* dbuf[0:count) <- sum over the ranks (doubles, in place, stream-ordered): what primme_amd_global_sum does for
 * host buffers and what the solver does with its device-resident partial sums */
int primme_amd_comm_allreduce(primme_amd_comm *comm, void *hip_stream, double *dbuf, int count);
/* Neighbour exchange into caller-owned buffers: my first `send_lo_cnt` rows of every column go to rank-1 (they land in ITS `hi`),
 * my last `send_hi_cnt` to rank+1 (ITS `lo`); column c of a halo buffer starts at c * count elements.  Stream-ordered on RCCL.
 * On the peer-to-peer transport the rows land in the communicator's own zones first; this entry point then agrees on the zone
 * size through the host rendez-vous (it SYNCHRONISES the ranks on the host) and copies out — the ready-made operator
 * (primme_amd_operator_apply) uses the zones in place and does neither. */
int primme_amd_comm_halo(primme_amd_comm *c, void *hip_stream, const void *x, int64_t ldx,
      int64_t nrows, int ncols, size_t elem, int64_t send_lo_cnt, int64_t send_hi_cnt, void *lo,
      int64_t recv_lo_cnt, void *hi, int64_t recv_hi_cnt);
int primme_amd_comm_allgather(primme_amd_comm *c, void *hip_stream, const void *send, void *recv,
      size_t bytes_per_rank);
/* recv[0:count) = sum over ranks of send[rank*count : (rank+1)*count), count elements per rank */
int primme_amd_comm_reduce_scatter(primme_amd_comm *c, void *hip_stream, const void *send, void *recv,
      size_t count_per_rank, int is_double);
/* a block of columns in one grouped exchange: column c of `send` (ld_send elements apart) to column c
 * of `recv` (examples/ex_eigs_mpi.c issues one MPI call per vector) */
int primme_amd_comm_allgather_cols(primme_amd_comm *c, void *hip_stream, const void *send, int64_t ld_send,
      void *recv, int64_t ld_recv, size_t bytes_per_rank, size_t elem, int ncols);
int primme_amd_comm_reduce_scatter_cols(primme_amd_comm *c, void *hip_stream, const void *send, int64_t ld_send,
      void *recv, int64_t ld_recv, size_t count_per_rank, int is_double, int ncols);
int primme_amd_comm_allgather_i64(primme_amd_comm *c, const int64_t *mine, int n, int64_t *all);
/* Self-test of a communicator on whatever transport it came up on (collective: every rank calls it): all-reduce of
 * 1 .. 4096 doubles, the neighbour halo, the bulk all-gather / reduce-scatter, the integer exchange — each against known
 * data — then `reps` back-to-back 8-double all-reduces; *allreduce_us (optional) = wall-clock microseconds per reduction of
 * that loop.  Returns 0 when every check matched on this rank, the number of failed checks otherwise, negative on a
 * transport error.  bench.py --gpus N runs it before the timed region and prints transport and latency. */
int primme_amd_comm_selftest(primme_amd_comm *c, void *hip_stream, int reps, double *allreduce_us);

/* Operator handle for primme->matrix / primme->preconditioner: a local sparse
 * operator plus (optionally) the communicator that feeds its halo. */
typedef struct primme_amd_operator primme_amd_operator;
int primme_amd_operator_create(primme_amd_operator **op, hipk_csr *A, primme_amd_comm *comm_or_null);
int primme_amd_operator_destroy(primme_amd_operator *op);
/* flavour of primme_amd_jacobi_precond for this operator: fixed = 0 (default) divides by
 * diag(A) - ShiftsForPreconditioner[c] (reference tests/COMMON/mat.c:187-193, examples/
 * ex_eigs_dseq.c:187-202); fixed = 1 by diag(A) - shift (mat.c:137-147, :166-170) */
int primme_amd_operator_set_jacobi(primme_amd_operator *op, int fixed, double shift);
/* on = 1: the matrix is the real-equivalent form (primme_amd_csr_complex_to_real) of a Hermitian
 * matrix and primme_amd_matvec / primme_amd_jacobi_precond are called by hip_zprimme /
 * hip_cprimme with leading dimensions counted in complex elements */
int primme_amd_operator_set_complex(primme_amd_operator *op, int on);
hipk_csr *primme_amd_operator_matrix(primme_amd_operator *op);
/* y = A (a x), xout = a x, dot_dev[0] = xout' y, a = 1/sqrt(norm2_dev[0]), halo exchange included: the
 * fused tail of the solver's one-synchronisation iteration (one column; CSR operators).
 * primme_amd_operator_can_fuse tells whether the operator supports it. */
int primme_amd_operator_can_fuse(const primme_amd_operator *op);
int primme_amd_operator_apply_scaled(primme_amd_operator *op, hipk_ctx *ctx /* the caller's context */, const void *x,
      const double *norm2_dev, void *xout, void *y, double *dot_dev);
/* y = A x - shifts[c] x(:,c) (single-rank CSR operators; returns 1 when not covered) and the data of the
 * operator's Jacobi preconditioner, for the fused steps of the JDQMR inner iteration (eigs_jd.c) */
int primme_amd_operator_apply_shifted(primme_amd_operator *op, void *hip_stream, const void *x, int64_t ldx,
      void *y, int64_t ldy, int ncols, const double *shifts_host);
int primme_amd_operator_jacobi_data(primme_amd_operator *op, const void **diag, int *fixed, double *shift);
/* y = A x on `hip_stream` including the halo exchange */
int primme_amd_operator_apply(primme_amd_operator *op, void *hip_stream, const void *x, int64_t ldx,
      void *y, int64_t ldy, int ncols);

#ifdef __cplusplus
}
#endif
#endif
