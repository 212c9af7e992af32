/* eigs_internal.h — declarations shared by the host side of the solver. */
#ifndef EIGS_INTERNAL_H
#define EIGS_INTERNAL_H

#include <stdint.h>
#include "primme_amd.h"
#include "primme_amd_kernels.h"
#include "eigs_scalar.h"

#define PA_MIN(a, b) ((a) < (b) ? (a) : (b))
#define PA_MAX(a, b) ((a) > (b) ? (a) : (b))
#define PA_EPS 2.220446049250313e-16 /* double */

/* dense helpers (eigs_dense.c); HS = double, or double complex in the complex objects (Hermitian / unitary
 * variants: ' reads as conjugate transpose) */
int  pa_sym_eig(int n, const HS *A, int lda, double *evals, HS *Z, int ldz);
int  pa_sym_eig_gen(int n, const HS *H, int ldh, const HS *G, int ldg, double *evals,
      HS *Z, int ldz);
int  pa_potrf_upper(int n, HS *A, int lda);
void pa_trsm_left_upper_trans(int n, int nb, const HS *U, int ldu, HS *B, int ldb);
void pa_trsm_left_upper(int n, int nb, const HS *U, int ldu, HS *B, int ldb);
void pa_trsm_right_upper(int mb, int n, const HS *U, int ldu, HS *B, int ldb);
void pa_permute_cols(HS *A, int mrows, int n, int lda, const int *perm);
void pa_permute_reals(double *A, int mrows, int n, int lda, const int *perm);   /* Ritz values and other real rows */
void pa_permute_ints(int *a, int n, const int *perm);
void pa_submatrix(const HS *X, int nx, int ldx, const HS *H, int nh, int ldh, HS *R,
      int ldr);
void pa_larnv_uniform11(int64_t iseed[4], int64_t n, double *x);

/* user callbacks called with the operand type they declare (eigs_callbacks.c) */
struct primme_svds_params;
int pa_call_global_sum(primme_params *p, double *buf, int count);
int pa_call_conv_test(primme_params *p, double eval, void *evec, double rnorm, int *isconv);
int pa_call_monitor(primme_params *p, double *basisEvals, int basisSize, int *basisFlags, int *iblock, int blockSize,
      double *basisNorms, int numConverged, double *lockedEvals, int numLocked, int *lockedFlags, double *lockedNorms,
      primme_event event);
int pa_call_monitor_inner(primme_params *p, double eval, double resNorm, int counts, int innerIts, double lsRes);
int pa_svds_call_global_sum(struct primme_svds_params *ps, double *buf, int count);
int pa_svds_call_conv_test(struct primme_svds_params *ps, double sval, void *leftsvec, void *rightsvec, double rnorm,
      int *method, int *isconv);

/* complex-independence sweep shared by hip_zprimme and hip_zprimme_svds (eigs_complex.c) */
typedef int (*pa_sum_fn)(void *who, double *buf, int count);
int pa_complex_sweep(hipk_ctx *ctx, hipk_dtype dtr, int64_t mr, int64_t ldr, char *cand, int ncand, char *Z, char *rot,
      int nwant, double *d_s, double *h_s, pa_sum_fn sum, void *who, int *picked, int *nacc);

/* parameter handling (eigs_params.c) */
int pa_check_input(const void *evals, const void *evecs, const void *resNorms,
      const primme_params *primme, double machine_eps);

/* communicator (comm_rccl.c): device all-reduce used when primme->globalSumReal ==
 * primme_amd_global_sum */
int pa_comm_allreduce_device(void *commInfo, double *dbuf, int count, void *hip_stream);
/* peer-to-peer transport (comm_ipc.hip): reduction + pinned mirror + completion flag in one launch; 1 = not on this transport */
int pa_comm_allreduce_publish(void *commInfo, hipk_ctx *ctx, double *dbuf, int count);
/* let the context's second-stage launches reduce across the ranks themselves (no-op on RCCL); 0 = attached */
int pa_comm_attach_ctx(void *commInfo, hipk_ctx *ctx);
/* arm the NEXT second-stage launch on the context as a cross-rank one (only with an attached communicator) */
void hipk_xreduce_arm(hipk_ctx *ctx);
/* 1 when a communicator of the peer-to-peer transport is attached to the context (arming has an effect) */
int hipk_xreduce_available(hipk_ctx *ctx);
/* 1 when [buf, buf+count) was produced by an armed launch since the last call: the sums are already global and
 * published (the record is consumed) */
int hipk_xreduce_covered(hipk_ctx *ctx, const double *buf, int count);
/* non-zero after a device-side wait of the transport ran into its time limit (a rank left the collective sequence) */
int pa_comm_failed(void *commInfo);
extern long pa_last_pre[2];   /* eigs_conv.c: pre-enqueued iterations of the last solve (launched, adopted) */

/* the matvec handle understands the communicator for halo exchange */
#endif
