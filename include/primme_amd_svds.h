/* primme_amd_svds.h — boundary B1 for singular value problems.
 *
 * Same types, field order and calling conventions as the reference's
 * include/primme_svds.h:46-168 (sizeof(primme_svds_params) = 1720 on x86-64), so a program
 * written against dprimme_svds() / cublas_dprimme_svds() switches by renaming the solver call:
 *
 *     reference                                   this library
 *     primme_svds_initialize  (interface.c:108)   primme_svds_initialize
 *     primme_svds_set_method  (interface.c:218)   primme_svds_set_method
 *     cublas_dprimme_svds     (primme_svds.h:264) hip_dprimme_svds   (svecs = DEVICE pointer)
 *     cublas_sprimme_svds     (primme_svds.h:260) hip_sprimme_svds
 *
 * Covered on the device path: the normal-equations method (A'A when n <= m, AA' otherwise;
 * reference primme_svds_interface.c:231-235), the augmented operator [0 A'; A 0] and the hybrid
 * default, for the largest / smallest / closest_abs singular triplets.  What the eigensolver does
 * not cover (the refined extraction with explicit_I, i.e. interior targets with block size > 1 or
 * in single precision) comes back as PRIMME_FUNCTION_UNAVAILABLE - 100 / - 200, never a CPU
 * fallback.
 */
#ifndef PRIMME_AMD_SVDS_H
#define PRIMME_AMD_SVDS_H

#include "primme_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { primme_svds_largest, primme_svds_smallest, primme_svds_closest_abs } primme_svds_target;
typedef enum {
   primme_svds_default, primme_svds_hybrid, primme_svds_normalequations, primme_svds_augmented
} primme_svds_preset_method;
typedef enum {
   primme_svds_op_none, primme_svds_op_AtA, primme_svds_op_AAt, primme_svds_op_augmented
} primme_svds_operator;

typedef struct primme_svds_stats {
   PRIMME_INT numOuterIterations, numRestarts, numMatvecs, numPreconds;
   PRIMME_INT numGlobalSum, numBroadcast, volumeGlobalSum, volumeBroadcast;
   double numOrthoInnerProds, elapsedTime, timeMatvec, timePrecond, timeOrtho;
   double timeGlobalSum, timeBroadcast;
   PRIMME_INT lockingIssue;
} primme_svds_stats;

struct primme_svds_params;
typedef void (*primme_svds_block_op)(void *x, PRIMME_INT *ldx, void *y, PRIMME_INT *ldy,
      int *blockSize, int *transpose, struct primme_svds_params *primme_svds, int *ierr);

typedef struct primme_svds_params {
   primme_params primme;        /* first-stage eigenproblem (must stay the first field) */
   primme_params primmeStage2;  /* hybrid second stage */
   PRIMME_INT m, n;             /* rows, columns of A */
   primme_svds_block_op matrixMatvec;        primme_op_datatype matrixMatvec_type;
   primme_svds_block_op applyPreconditioner; primme_op_datatype applyPreconditioner_type;
   int numProcs, procID;
   PRIMME_INT mLocal, nLocal;
   void *commInfo;
   void (*globalSumReal)(void *sendBuf, void *recvBuf, int *count,
         struct primme_svds_params *primme_svds, int *ierr);
   primme_op_datatype globalSumReal_type;
   void (*broadcastReal)(void *buffer, int *count, struct primme_svds_params *primme_svds, int *ierr);
   primme_op_datatype broadcastReal_type;
   int numSvals;
   primme_svds_target target;
   int numTargetShifts;
   double *targetShifts;
   primme_svds_operator method, methodStage2;
   void *matrix, *preconditioner;
   int locking, numOrthoConst;
   double aNorm, eps;
   int precondition, initSize, maxBasisSize, maxBlockSize;
   PRIMME_INT maxMatvecs;
   PRIMME_INT iseed[4];
   int printLevel;
   primme_op_datatype internalPrecision;
   FILE *outputFile;
   struct primme_svds_stats stats;
   void (*convTestFun)(double *sval, void *leftsvec, void *rightsvec, double *rNorm, int *method,
         int *isconv, struct primme_svds_params *primme_svds, int *ierr);
   primme_op_datatype convTestFun_type;
   void *convtest;
   void (*monitorFun)(void *basisSvals, int *basisSize, int *basisFlags, int *iblock, int *blockSize,
         void *basisNorms, int *numConverged, void *lockedSvals, int *numLocked, int *lockedFlags,
         void *lockedNorms, int *inner_its, void *LSRes, const char *msg, double *time,
         primme_event *event, int *stage, struct primme_svds_params *primme_svds, int *err);
   primme_op_datatype monitorFun_type;
   void *monitor;
   void *queue;            /* hipStream_t* of the caller, or NULL */
   const char *profile;
} primme_svds_params;

primme_svds_params *primme_svds_params_create(void);
int primme_svds_params_destroy(primme_svds_params *primme_svds);
void primme_svds_initialize(primme_svds_params *primme_svds);
int primme_svds_set_method(primme_svds_preset_method method, primme_preset_method methodStage1,
      primme_preset_method methodStage2, primme_svds_params *primme_svds);
void primme_svds_set_defaults(primme_svds_params *primme_svds);
void primme_svds_free(primme_svds_params *primme_svds);

/* svecs: DEVICE array [U (mLocal x numSvals, ld mLocal) | V (nLocal x numSvals, ld nLocal)] behind
 * numOrthoConst constraint columns of each; svals, resNorms: host arrays.  The user matvec gets
 * device pointers and runs on the stream in *primme_svds->queue (set by the solver when NULL). */
int hip_dprimme_svds(double *svals, double *svecs, double *resNorms, primme_svds_params *primme_svds);
int hip_sprimme_svds(float *svals, float *svecs, float *resNorms, primme_svds_params *primme_svds);
/* complex matrices (reference primme_svds.h:242-243, :268-271: cublas_zprimme_svds / cublas_cprimme_svds): svecs holds
 * complex vectors (re, im interleaved); through the real-equivalent form, csrc/svds_complex.c */
int hip_zprimme_svds(double *svals, void *svecs, double *resNorms, primme_svds_params *primme_svds);
int hip_cprimme_svds(float *svals, void *svecs, float *resNorms, primme_svds_params *primme_svds);

/* The reference's CPU entry points with their HOST-pointer contract (reference include/primme_svds.h:236-243,
 * src/svds/primme_svds_c.c:113-118; BASELINE configs[4] is worded with dprimme_svds): svecs is a host array
 * [Uc U | Vc V], the callbacks get host pointers, primme_svds->queue must be NULL.  A program written against the
 * CPU library (examples/ex_svds_dseq.c) relinks unchanged; the solve runs on the device and every operator
 * application is staged through pinned host memory (csrc/svds_hostapi.c) — the plumbing path; hand
 * hip_?primme_svds a device callback or the library's CSR operator for the device rate. */
int dprimme_svds(double *svals, double *svecs, double *resNorms, primme_svds_params *primme_svds);
int sprimme_svds(float *svals, float *svecs, float *resNorms, primme_svds_params *primme_svds);
int zprimme_svds(double *svals, void *svecs, double *resNorms, primme_svds_params *primme_svds);
int cprimme_svds(float *svals, void *svecs, float *resNorms, primme_svds_params *primme_svds);

/* ---- ready-made matvec for a device-resident CSR matrix and its transpose ----------------
 * primme_svds->matrix = handle from primme_amd_svds_operator_create (A and A' are both kept in
 * CSR: the transposed product is then the same coalesced row-tile kernel instead of a scatter
 * with atomics).  Replaces the hand-written callbacks of the reference's examples
 * (examples/ex_svds_dseq.c) and of tests/COMMON/mat.c (CSRMatrixMatvecSVD). */
typedef struct primme_amd_svds_operator primme_amd_svds_operator;
struct hipk_ctx;
int primme_amd_svds_operator_create(primme_amd_svds_operator **op, struct hipk_ctx *ctx, int dt,
      int64_t m, int64_t n, const int32_t *rowptr_host, const int32_t *colind_host,
      const void *values_host);
/* Row-partitioned A across the ranks of a communicator (BASELINE configs[4]: 8 M x 2 M over 8
 * GPUs): this rank owns rows [row0, row0+mLocal) of A and entries [col0, col0+nLocal) of every
 * n-vector, nLocal equal on all ranks.  rowptr/colind (GLOBAL column numbers)/values describe the
 * local rows.  y = A x: all-gather of x, local product.  y = A' x: local product with the local
 * rows' transpose into a full n-vector, reduce-scatter.  comm = primme_amd_comm* (primme_amd_comm.h). */
int primme_amd_svds_operator_create_dist(primme_amd_svds_operator **op, struct hipk_ctx *ctx, int dt,
      int64_t mLocal, int64_t n, int64_t nLocal, const int32_t *rowptr_host, const int32_t *colind_host,
      const void *values_host, void *comm);
int primme_amd_svds_operator_destroy(primme_amd_svds_operator *op);
/* on = 1: the matrix is the real-equivalent form (2m x 2n, primme_amd_csr_complex_to_real) of a complex one and
 * primme_amd_svds_matvec is called by hip_zprimme_svds / hip_cprimme_svds with leading dimensions counted in
 * complex elements (single rank) */
int primme_amd_svds_operator_set_complex(primme_amd_svds_operator *op, int on);
/* globalSumReal with the primme_svds signature over the same communicator
 * (primme_svds->commInfo = primme_amd_comm*).  When hip_dprimme_svds sees this function installed
 * it gives the eigensolver the in-stream RCCL reduction of primme_amd_global_sum. */
void primme_amd_svds_global_sum(void *sendBuf, void *recvBuf, int *count,
      struct primme_svds_params *primme_svds, int *ierr);
void primme_amd_svds_matvec(void *x, PRIMME_INT *ldx, void *y, PRIMME_INT *ldy, int *blockSize,
      int *transpose, struct primme_svds_params *primme_svds, int *ierr);
/* Ready-made applyPreconditioner: the diagonal preconditioner of the reference's test driver for
 * singular value problems (tests/COMMON/mat.c:353-426, driver.PrecChoice = jacobi): y = x ./
 * (diag(A'A) - shift^2) for the A'A operator, x ./ (diag(AA') - shift^2) for AA', both halves for the
 * augmented operator.  set_jacobi builds the two diagonals from the host CSR arrays the operator
 * was created from (single-rank operators); primme_svds->preconditioner = the operator handle. */
int primme_amd_svds_operator_set_jacobi(primme_amd_svds_operator *op, const int32_t *rowptr_host,
      const int32_t *colind_host, const void *values_host, double shift);
void primme_amd_svds_jacobi_precond(void *x, PRIMME_INT *ldx, void *y, PRIMME_INT *ldy, int *blockSize,
      int *mode, struct primme_svds_params *primme_svds, int *ierr);

#ifdef __cplusplus
}
#endif
#endif
