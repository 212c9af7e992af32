/* primme_amd.h — public solver ABI of the MI355X-native Davidson/GD+k path.
 *
 * This is boundary B1 of SURVEY.md §8(b): the structs, enums and entry points a
 * caller of the reference's eigensolver binds to.  A program written against the
 * reference's GPU flavour (cublas_dprimme: device `evecs`, device pointers handed
 * to matrixMatvec, host evals/resNorms — reference examples/ex_eigs_dhipblas.c:174-182,
 * :239-264) switches to this library by calling hip_dprimme() instead.
 *
 * LAYOUT IS ABI.  Field order, widths and enum values below reproduce
 * reference include/primme_eigs.h:47-253 (sizeof(primme_params) == 640,
 * sizeof(primme_stats) == 200 on x86-64; tests/test_abi_and_params.py checks every offset
 * against the table captured from the reference build).  Only declarations are
 * shared; every definition in this repository is new code.
 */
#ifndef PRIMME_AMD_H
#define PRIMME_AMD_H

#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef PRIMME_INT
#define PRIMME_INT int64_t /* reference include/primme.h:82-87 */
#endif

/* ---- return codes (reference include/primme.h:116-123; -4..-39 are the
 *      argument checks of reference src/eigs/primme_c.c:438-535) ------------- */
#define PRIMME_UNEXPECTED_FAILURE   (-1)
#define PRIMME_MALLOC_FAILURE       (-2)
#define PRIMME_MAIN_ITER_FAILURE    (-3)
#define PRIMME_LAPACK_FAILURE       (-40)
#define PRIMME_USER_FAILURE         (-41)
#define PRIMME_ORTHO_CONST_FAILURE  (-42)
#define PRIMME_PARALLEL_FAILURE     (-43)
#define PRIMME_FUNCTION_UNAVAILABLE (-44)

/* ---- enums: values are ABI (reference include/primme_eigs.h:47-107) --------- */
typedef enum {
   primme_smallest = 0, primme_largest, primme_closest_geq, primme_closest_leq,
   primme_closest_abs, primme_largest_abs
} primme_target;

typedef enum {
   primme_proj_default = 0, primme_proj_RR, primme_proj_harmonic, primme_proj_refined
} primme_projection;

typedef enum {
   primme_init_default = 0, primme_init_krylov, primme_init_random, primme_init_user
} primme_init;

typedef enum {
   primme_full_LTolerance = 0, primme_decreasing_LTolerance,
   primme_adaptive_ETolerance, primme_adaptive
} primme_convergencetest;

typedef enum {
   primme_event_outer_iteration = 0, primme_event_inner_iteration, primme_event_restart,
   primme_event_reset, primme_event_converged, primme_event_locked,
   primme_event_message, primme_event_profile
} primme_event;

typedef enum {
   primme_orth_default = 0, primme_orth_implicit_I, primme_orth_explicit_I
} primme_orth;

typedef enum {
   primme_op_default = 0, primme_op_half, primme_op_float, primme_op_double,
   primme_op_quad, primme_op_int
} primme_op_datatype;

/* ---- counters (reference include/primme_eigs.h:109-135) --------------------- */
typedef struct primme_stats {
   PRIMME_INT numOuterIterations, numRestarts, numMatvecs, numPreconds;
   PRIMME_INT numGlobalSum, numBroadcast, volumeGlobalSum, volumeBroadcast;
   double flopsDense;          /* flops of the fused Ritz/residual panel updates   */
   double numOrthoInnerProds;  /* one per basis column touched by the orthogonaliser */
   double elapsedTime, timeMatvec, timePrecond, timeOrtho, timeGlobalSum,
          timeBroadcast, timeDense;
   double estimateMinEVal, estimateMaxEVal, estimateLargestSVal;
   double estimateBNorm, estimateInvBNorm;
   double maxConvTol;             /* largest residual norm among locked pairs */
   double estimateResidualError;  /* accumulated error in V, W                */
   PRIMME_INT lockingIssue;
} primme_stats;

typedef struct JD_projectors { int LeftQ, LeftX, RightQ, RightX, SkewQ, SkewX; } JD_projectors;
typedef struct projection_params { primme_projection projection; } projection_params;
typedef struct correction_params {
   int precondition, robustShifts, maxInnerIterations;
   struct JD_projectors projectors;
   primme_convergencetest convTest;
   double relTolBase;
} correction_params;
typedef struct restarting_params { int maxPrevRetain; } restarting_params;

struct primme_params;
/* Block operator callback: y(:,0:bs) = Op * x(:,0:bs); everything by pointer,
 * *ierr != 0 aborts the solve with PRIMME_USER_FAILURE
 * (reference include/primme_eigs.h:170-185, src/eigs/auxiliary_eigs.c:183-230).
 * In hip_?primme x and y are DEVICE pointers, ld = ldOPs. */
typedef void (*primme_block_op)(void *x, PRIMME_INT *ldx, void *y, PRIMME_INT *ldy,
      int *blockSize, struct primme_params *primme, int *ierr);

/* ---- the parameter block (reference include/primme_eigs.h:166-253) ---------- */
typedef struct primme_params {
   PRIMME_INT n;
   primme_block_op matrixMatvec;        primme_op_datatype matrixMatvec_type;
   primme_block_op applyPreconditioner; primme_op_datatype applyPreconditioner_type;
   primme_block_op massMatrixMatvec;    primme_op_datatype massMatrixMatvec_type;

   /* row partition (reference :187-198, examples/ex_eigs_mpi.c:100-123) */
   int numProcs, procID;
   PRIMME_INT nLocal;
   void *commInfo;
   void (*globalSumReal)(void *sendBuf, void *recvBuf, int *count,
         struct primme_params *primme, int *ierr);
   primme_op_datatype globalSumReal_type;
   void (*broadcastReal)(void *buffer, int *count, struct primme_params *primme, int *ierr);
   primme_op_datatype broadcastReal_type;

   int numEvals;
   primme_target target;
   int numTargetShifts;
   double *targetShifts;

   int dynamicMethodSwitch, locking, initSize, numOrthoConst;
   int maxBasisSize, minRestartSize, maxBlockSize;
   PRIMME_INT maxMatvecs, maxOuterIterations;
   PRIMME_INT iseed[4];
   double aNorm, BNorm, invBNorm, eps;
   primme_orth orth;
   primme_op_datatype internalPrecision;

   int printLevel;
   FILE *outputFile;

   void *matrix, *preconditioner, *massMatrix;
   double *ShiftsForPreconditioner;
   primme_init initBasisMode;
   PRIMME_INT ldevecs, ldOPs;

   struct projection_params projectionParams;
   struct restarting_params restartingParams;
   struct correction_params correctionParams;
   struct primme_stats stats;

   void (*convTestFun)(double *eval, void *evec, double *rNorm, int *isconv,
         struct primme_params *primme, int *ierr);
   primme_op_datatype convTestFun_type;
   void *convtest;
   void (*monitorFun)(void *basisEvals, int *basisSize, int *basisFlags, int *iblock,
         int *blockSize, void *basisNorms, int *numConverged, void *lockedEvals,
         int *numLocked, int *lockedFlags, void *lockedNorms, int *inner_its,
         void *LSRes, const char *msg, double *time, primme_event *event,
         struct primme_params *primme, int *err);
   primme_op_datatype monitorFun_type;
   void *monitor;
   void *queue;          /* hip_?primme: optional hipStream_t* (NULL = library stream) */
   const char *profile;
} primme_params;

typedef enum {
   PRIMME_DEFAULT_METHOD = 0, PRIMME_DYNAMIC, PRIMME_DEFAULT_MIN_TIME,
   PRIMME_DEFAULT_MIN_MATVECS, PRIMME_Arnoldi, PRIMME_GD, PRIMME_GD_plusK,
   PRIMME_GD_Olsen_plusK, PRIMME_JD_Olsen_plusK, PRIMME_RQI, PRIMME_JDQR,
   PRIMME_JDQMR, PRIMME_JDQMR_ETol, PRIMME_STEEPEST_DESCENT,
   PRIMME_LOBPCG_OrthoBasis, PRIMME_LOBPCG_OrthoBasis_Window
} primme_preset_method;

/* convergence flags reported to monitorFun (reference src/eigs/common_eigs.h:41-46) */
enum primme_amd_conv_flags {
   PRIMME_AMD_UNCONVERGED = 0, PRIMME_AMD_SKIP_UNTIL_RESTART, PRIMME_AMD_CONVERGED,
   PRIMME_AMD_PRACTICALLY_CONVERGED
};

/* ---- entry points ----------------------------------------------------------- */
/* reference src/eigs/primme_interface.c:101, :293, :543, :228 */
void primme_initialize(primme_params *primme);
int  primme_set_method(primme_preset_method method, primme_params *primme);
void primme_set_defaults(primme_params *primme);
void primme_free(primme_params *primme);
primme_params *primme_params_create(void);
int  primme_params_destroy(primme_params *primme);

/* Solvers: same contract as the reference's cublas_?primme / magma_?primme
 * (reference include/primme_eigs.h:394-417, src/eigs/primme_c.c:103-108):
 *   evals[numEvals], resNorms[numEvals]   HOST arrays
 *   evecs[ldevecs*(numOrthoConst+max(numEvals,initSize))]  DEVICE array, column-major;
 *        on input: numOrthoConst constraint vectors then initSize guesses
 *   returns 0, or <0 (codes above); primme->initSize = converged pairs returned;
 *   primme->stats filled; primme->aNorm back-filled when it was <= 0.
 * Requires the GPU: there is no CPU fallback; a missing/failed device returns
 * PRIMME_UNEXPECTED_FAILURE.
 * hip_zprimme / hip_cprimme (Hermitian problems; evecs and the callbacks' vectors are DEVICE
 * arrays of (re, im) pairs, leading dimensions in complex elements exactly as for
 * cublas_zprimme) run natively on complex panels (csrc/hipk_complex.hip, the host solver compiled
 * for double complex, csrc/eigs_*_z.c): the GD family, JDQMR and the dynamic switch, with
 * Rayleigh-Ritz, harmonic or refined extraction — zprimme's eigenpairs, residual norms and, on the
 * committed extremal fixtures, its outer-iteration / matvec / restart counts.  PRIMME_AMD_COMPLEX_REAL_FORM=1
 * (csrc/eigs_complex.c) selects the real-equivalent 2n form instead: same eigenpairs, about
 * twice the operator applications; kept as a cross-check. */
int hip_dprimme(double *evals, double *evecs, double *resNorms, primme_params *primme);
int hip_zprimme(double *evals, void *evecs, double *resNorms, primme_params *primme);
int hip_sprimme(float *evals, float *evecs, float *resNorms, primme_params *primme);
int hip_cprimme(float *evals, void *evecs, float *resNorms, primme_params *primme);

/* The reference's CPU entry points with their HOST-pointer contract (include/primme_eigs.h:386-393,
 * src/eigs/primme_c.c:103-108): evecs is a host array [constraints | guesses], the callbacks get host
 * pointers with leading dimension ldOPs, primme->queue must be NULL.  A program written against the
 * CPU library (examples/ex_eigs_dseq.c) relinks unchanged; the solve runs on the device and every
 * operator application is staged through pinned host memory (csrc/eigs_hostapi.c), so this is the
 * plumbing path — use hip_?primme with a device callback for the device rate. */
int dprimme(double *evals, double *evecs, double *resNorms, primme_params *primme);
int zprimme(double *evals, void *evecs, double *resNorms, primme_params *primme);
int sprimme(float *evals, float *evecs, float *resNorms, primme_params *primme);
int cprimme(float *evals, void *evecs, float *resNorms, primme_params *primme);

/* ---- ready-made callbacks (what examples/ex_eigs_dhipblas.c:239-264 and
 *      examples/ex_eigs_mpi.c:209-218 hand-write for every application) ------- */

/* matrixMatvec for a device CSR matrix: set primme->matrix = handle returned by
 * primme_amd_operator_create() over a hipk_csr_create() / hipk_stencil_create() matrix
 * (primme_amd_comm.h, primme_amd_kernels.h). */
void primme_amd_matvec(void *x, PRIMME_INT *ldx, void *y, PRIMME_INT *ldy, int *blockSize,
      struct primme_params *primme, int *ierr);
/* massMatrixMatvec for a device CSR mass matrix B of a generalised problem A x = lambda B x (round 6; reference
 * primme_eigs.h:182-185, auxiliary_eigs.c:250-290): set primme->massMatrix = another operator handle. */
void primme_amd_mass_matvec(void *x, PRIMME_INT *ldx, void *y, PRIMME_INT *ldy, int *blockSize,
      struct primme_params *primme, int *ierr);
/* Jacobi (diagonal) preconditioner y = (diag(A) - shift)^-1 x with the shifts the
 * solver publishes in primme->ShiftsForPreconditioner; primme->preconditioner =
 * the same operator handle, flavour chosen with
 * primme_amd_operator_set_jacobi() (cf. reference tests/COMMON/mat.c
 * createInvDiagPrecNative / examples/ex_eigs_dseq.c:187-202). */
void primme_amd_jacobi_precond(void *x, PRIMME_INT *ldx, void *y, PRIMME_INT *ldy,
      int *blockSize, struct primme_params *primme, int *ierr);
/* globalSumReal over RCCL: set primme->commInfo = handle from primme_amd_comm_create().
 * When the solver sees this exact function installed it reduces its DEVICE partials
 * with ncclAllReduce before the (single) device->host copy; called directly it also
 * honours the reference contract on HOST buffers (send may equal recv). */
void primme_amd_global_sum(void *sendBuf, void *recvBuf, int *count,
      struct primme_params *primme, int *ierr);

/* Diagnostics of the LAST solve of this process (block size 1, GD+k family): how many outer iterations were enqueued before
 * the host had seen the previous one and how many of those the host then adopted.  No reference
 * counterpart (the reference has no device queue to run ahead of). */
void primme_amd_prelaunch_stats(long *launched, long *adopted);
/* Diagnostics of this process (JDQMR): inner QMR steps taken since the last call, and how many of them ran with ONE host
 * synchronisation (block sizes 2 .. 8 with the library's Jacobi preconditioner: DESIGN.md section 4b); the call resets both. */
void primme_amd_qmr_step_stats(long *steps, long *one_synchronisation);

#ifdef __cplusplus
}
#endif
#endif /* PRIMME_AMD_H */
