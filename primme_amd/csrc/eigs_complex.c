/* eigs_complex.c — hip_zprimme / hip_cprimme: Hermitian problems through their real-equivalent form.
 *
 * Boundary: reference include/primme_eigs.h:394-417 (cublas_zprimme / cublas_cprimme), front end
 * src/eigs/primme_c.c:103-108 with SCALAR = complex.  The reference instantiates every routine
 * of the solver once more for complex scalars.  This path does not (yet) have complex panel
 * kernels; instead it uses the fact that a complex n-vector in memory IS a real 2n-vector
 * (re0, im0, re1, im1, ...), and that in this ordering
 *
 *     z -> A z          is a real-linear, symmetric map M of R^2n when A is Hermitian,
 *     Re(x^H y)         is the real inner product of the two 2n-vectors,
 *     i z               is the pair rotation (re, im) -> (-im, re)      (hipk_pair_rotate).
 *
 * Every eigenvalue of A is an eigenvalue of M with twice the multiplicity, and every real
 * eigenvector of M, read back as a complex vector, is an eigenvector of A (u and i u span the same
 * complex line).  So the user's complex callbacks are applied, untouched, to the real solver's
 * 2n-vectors; the real solver (eigs_main.c, all methods, all extractions) is asked for
 * 2 numEvals pairs; and the complex-linearly independent ones are selected at the end by a
 * complex Gram-Schmidt sweep built from the real TN/NN panel kernels
 * ( z^H u = dot(z,u) + i dot(iz,u),   u - c z = u - Re(c) z - Im(c) (iz) ).
 *
 * Cost: twice the eigenpairs on vectors of the same byte length as the complex ones, i.e. the
 * iteration counts are NOT those of the reference's zprimme (the eigenvalues, eigenvectors and
 * residual norms are, to the tolerance).  Native complex kernels are the follow-up.
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include "primme_amd.h"
#include "primme_amd_kernels.h"
#include "primme_amd_comm.h"
#include "eigs_internal.h"

int pa_eigs_solve(void *evals_out, void *evecs, void *resNorms_out, primme_params *p, hipk_dtype dt, int out_double);
/* the complex instantiation of the host solver (eigs_main_z.c and friends, eigs_scalar.h) */
int pa_eigs_solve_z(void *evals_out, void *evecs, void *resNorms_out, primme_params *p, hipk_dtype dt, int out_double);

/* NATIVE path: complex panels, complex Hermitian projected problem, the reference's zprimme iteration for iteration.
 * Covers the Generalized-Davidson family (GD, GD+k, Olsen variants, LOBPCG-like presets; locking and soft locking;
 * any block size), which includes the default method and BASELINE configs[3], the JDQMR inner-outer iteration
 * (real QMR recurrences on the 2m-real view of the panels, complex projectors), the dynamic switch between the two,
 * and the three extractions (Rayleigh-Ritz; harmonic and refined for interior targets, eigs_harm.c).
 * Only the library's operator when it was built on the real-equivalent CSR expansion takes the
 * real-equivalent form below.  PRIMME_AMD_COMPLEX_REAL_FORM=1 forces the latter (A/B measurements). */
static int native_complex_ok(const primme_params *primme) {
   if (getenv("PRIMME_AMD_COMPLEX_REAL_FORM")) return 0;
   if (primme->matrixMatvec == primme_amd_matvec) {
      /* the library's operator: native only over a complex CSR matrix */
      if (!primme->matrix) return 0;
      const hipk_dtype odt = hipk_csr_dtype(primme_amd_operator_matrix((primme_amd_operator *)primme->matrix));
      if (odt != HIPK_C64 && odt != HIPK_C32) return 0;
   }
   if (primme->applyPreconditioner == primme_amd_jacobi_precond) {
      if (!primme->preconditioner) return 0;
      const hipk_dtype odt = hipk_csr_dtype(primme_amd_operator_matrix((primme_amd_operator *)primme->preconditioner));
      if (odt != HIPK_C64 && odt != HIPK_C32) return 0;
   }
   return 1;
}

typedef struct {
   primme_params *user;   /* the caller's struct: what its callbacks expect to receive */
   primme_params q;       /* the real problem of twice the size handed to the solver */
} cplx_side;

#define SIDE_OF(pp) ((cplx_side *)((char *)(pp) - offsetof(cplx_side, q)))

static void sync_user(cplx_side *sd) {
   sd->user->queue = sd->q.queue;
   sd->user->ShiftsForPreconditioner = sd->q.ShiftsForPreconditioner;
   sd->user->stats = sd->q.stats;
   sd->user->aNorm = sd->q.aNorm;
}

static void cx_matvec(void *x, PRIMME_INT *ldx, void *y, PRIMME_INT *ldy, int *blockSize, primme_params *qp, int *ierr) {
   cplx_side *sd = SIDE_OF(qp);
   PRIMME_INT lx = *ldx / 2, ly = *ldy / 2;
   sync_user(sd);
   sd->user->matrixMatvec(x, &lx, y, &ly, blockSize, sd->user, ierr);
}
static void cx_precond(void *x, PRIMME_INT *ldx, void *y, PRIMME_INT *ldy, int *blockSize, primme_params *qp, int *ierr) {
   cplx_side *sd = SIDE_OF(qp);
   PRIMME_INT lx = *ldx / 2, ly = *ldy / 2;
   sync_user(sd);
   sd->user->applyPreconditioner(x, &lx, y, &ly, blockSize, sd->user, ierr);
}
static void cx_global_sum(void *s, void *r, int *count, primme_params *qp, int *ierr) {
   cplx_side *sd = SIDE_OF(qp);
   if (s != r) memcpy(r, s, sizeof(double) * (size_t)*count);
   *ierr = pa_call_global_sum(sd->user, (double *)r, *count) ? 1 : 0;
}
static void cx_broadcast(void *buf, int *count, primme_params *qp, int *ierr) {
   cplx_side *sd = SIDE_OF(qp);
   sd->user->broadcastReal(buf, count, sd->user, ierr);
}
static void cx_conv_test(double *eval, void *evec, double *rNorm, int *isconv, primme_params *qp, int *ierr) {
   cplx_side *sd = SIDE_OF(qp);
   sync_user(sd);
   *ierr = pa_call_conv_test(sd->user, *eval, evec, *rNorm, isconv) ? 1 : 0;
}
static void cx_monitor(void *basisEvals, int *basisSize, int *basisFlags, int *iblock, int *blockSize,
      void *basisNorms, int *numConverged, void *lockedEvals, int *numLocked, int *lockedFlags,
      void *lockedNorms, int *inner_its, void *LSRes, const char *msg, double *time, primme_event *event,
      primme_params *qp, int *ierr) {
   cplx_side *sd = SIDE_OF(qp);
   sync_user(sd);
   sd->user->monitorFun(basisEvals, basisSize, basisFlags, iblock, blockSize, basisNorms, numConverged,
         lockedEvals, numLocked, lockedFlags, lockedNorms, inner_its, LSRes, msg, time, event, sd->user, ierr);
}

/* Complex Gram-Schmidt sweep: out of `ncand` real 2n-vectors (columns of `cand`, leading dimension ldr, in
 * order) take up to `nwant` that are independent as COMPLEX vectors -- u and i u span the same complex line, and the
 * real solver returns both.  An accepted vector is orthonormalised (complex sense) against the ones taken before,
 * stored in Z(:,a) and its i-multiple in rot(:,a); picked[a] = its candidate index.  z^H u = dot(z,u) + i dot(iz,u),
 * u - c z = u - Re(c) z - Im(c) (iz): real TN / NN panel kernels.  d_s / h_s: 4 nwant + 8 doubles (device / pinned). */
int pa_complex_sweep(hipk_ctx *ctx, hipk_dtype dtr, int64_t mr, int64_t ldr, char *cand, int ncand, char *Z, char *rot,
      int nwant, double *d_s, double *h_s, pa_sum_fn sum, void *who, int *picked, int *nacc) {
   const size_t colB = (size_t)(ldr > 0 ? ldr : 1) * ((dtr == HIPK_F64) ? 8 : 4);
   int acc = 0;
   unsigned char *used = (unsigned char *)calloc((size_t)ncand + 1, 1);
   if (!used) return PRIMME_MALLOC_FAILURE;
   const double thresholds[2] = {0.25, 1e-6};
   for (int pass = 0; pass < 2 && acc < nwant; pass++) {
      for (int j = 0; j < ncand && acc < nwant; j++) {
         if (used[j]) continue;
         char *u = cand + colB * (size_t)j;
         double nrm2 = 1.0;
         if (acc > 0) {
            hipk_seg segs[2] = {{Z, ldr, acc}, {rot, ldr, acc}};
            int cnt = 2 * acc;
            if (hipk_panel_dots(ctx, dtr, mr, segs, 2, u, ldr, 1, d_s, cnt) ||
                hipk_d2h(ctx, h_s, d_s, sizeof(double) * (size_t)cnt) || hipk_sync(ctx)) { free(used); return PRIMME_UNEXPECTED_FAILURE; }
            if (sum && sum(who, h_s, cnt)) { free(used); return PRIMME_USER_FAILURE; }
            if (hipk_h2d(ctx, d_s, h_s, sizeof(double) * (size_t)(2 * acc)) ||
                hipk_panel_project(ctx, dtr, mr, segs, 2, d_s, 2 * acc, u, ldr, 1, d_s + 2 * acc) ||
                hipk_d2h(ctx, h_s, d_s + 2 * acc, sizeof(double)) || hipk_sync(ctx)) { free(used); return PRIMME_UNEXPECTED_FAILURE; }
            if (sum && sum(who, h_s, 1)) { free(used); return PRIMME_USER_FAILURE; }
            nrm2 = h_s[0];
         }
         if (!(nrm2 > thresholds[pass])) continue;     /* i times (a combination of) vectors already taken */
         used[j] = 1;
         if (acc > 0) {
            const double a = 1.0 / sqrt(nrm2);
            if (hipk_scale_cols(ctx, dtr, mr, u, ldr, 1, &a)) { free(used); return PRIMME_UNEXPECTED_FAILURE; }
         }
         if (hipk_copy_cols(ctx, dtr, mr, u, ldr, Z + colB * (size_t)acc, ldr, 1) ||
             hipk_pair_rotate(ctx, dtr, mr / 2, u, ldr, rot + colB * (size_t)acc, ldr, 1)) { free(used); return PRIMME_UNEXPECTED_FAILURE; }
         picked[acc++] = j;
      }
   }
   free(used);
   *nacc = acc;
   return 0;
}
static int sum_eigs(void *who, double *buf, int count) { return pa_call_global_sum((primme_params *)who, buf, count); }

#define CX(call) do { int rc__ = (call); if (rc__) { ret = rc__ < 0 ? rc__ : PRIMME_UNEXPECTED_FAILURE; goto done; } } while (0)

static int solve_complex(void *evals_out, void *evecs, void *resNorms_out, primme_params *primme, hipk_dtype dtr) {
   if (!primme) return -4;
   if (!evals_out && !evecs && !resNorms_out) {   /* defaults query (reference primme_c.c:301-306) */
      if (primme->numProcs <= 1) { primme->nLocal = primme->n; primme->procID = 0; }
      primme_set_defaults(primme);
      return 0;
   }
   /* generalised Hermitian problems (round 6): on the native complex panels only */
   if (primme->massMatrixMatvec && !native_complex_ok(primme)) return PRIMME_FUNCTION_UNAVAILABLE;
   if (!evals_out) return -30;
   if (!evecs || !hipk_is_device_ptr(evecs)) return -31;
   if (!resNorms_out) return -32;
   if (!primme->matrixMatvec) return -6;
   if (primme->n < 0 || primme->numEvals < 0 || primme->numEvals > primme->n) return primme->n < 0 ? -5 : -11;

   if (native_complex_ok(primme))
      return pa_eigs_solve_z(evals_out, evecs, resNorms_out, primme, dtr == HIPK_F64 ? HIPK_C64 : HIPK_C32, 0);

   cplx_side *sd = (cplx_side *)calloc(1, sizeof(cplx_side));
   if (!sd) return PRIMME_MALLOC_FAILURE;
   sd->user = primme;
   sd->q = *primme;                               /* before defaults: unset sizes stay unset */
   {
      /* basis sizes the caller's primme_set_method derived from the complex problem's n are derived
       * again for the real problem; sizes the caller (or a preset) chose explicitly are kept */
      primme_params t = *primme;
      if (t.numProcs <= 1) t.nLocal = t.n;
      t.maxBasisSize = 0; t.minRestartSize = 0;
      primme_set_defaults(&t);
      if (t.maxBasisSize == primme->maxBasisSize) {
         sd->q.maxBasisSize = 0;
         if (t.minRestartSize == primme->minRestartSize) sd->q.minRestartSize = 0;
      } else {
         t = *primme;
         if (t.numProcs <= 1) t.nLocal = t.n;
         t.minRestartSize = 0;
         primme_set_defaults(&t);
         if (t.minRestartSize == primme->minRestartSize) sd->q.minRestartSize = 0;
      }
   }
   if (primme->numProcs <= 1) { primme->nLocal = primme->n; primme->procID = 0; }
   primme_set_defaults(primme);
   if (primme->ldOPs == -1) primme->ldOPs = primme->nLocal;

   primme_params *q = &sd->q;
   const int nev = primme->numEvals, nOC = primme->numOrthoConst, init = primme->initSize;
   const int64_t ldu = primme->ldevecs;            /* complex elements */
   const int64_t mr = 2 * (int64_t)primme->nLocal, ldr = 2 * ldu;
   const size_t esr = (dtr == HIPK_F64) ? 8 : 4;
   q->n = 2 * primme->n; q->nLocal = mr; q->ldevecs = ldr;
   q->ldOPs = primme->ldOPs > 0 ? 2 * primme->ldOPs : primme->ldOPs;
   q->numEvals = 2 * nev; q->numOrthoConst = 2 * nOC; q->initSize = 2 * init;
   /* the reference's own rule for the default (primme_interface.c:601-607: lock when the wanted
    * pairs do not fit in the restarted basis), re-applied to the doubled count */
   primme_set_defaults(q);
   if (q->minRestartSize <= 0 && q->n > 2) q->minRestartSize = 1;   /* presets with fixed tiny bases */
   if (q->locking == 0 && q->numEvals > q->minRestartSize) q->locking = 1;
   q->matrixMatvec = cx_matvec;
   if (primme->applyPreconditioner) q->applyPreconditioner = cx_precond;
   if (primme->globalSumReal && primme->globalSumReal != primme_amd_global_sum) { q->globalSumReal = cx_global_sum; q->globalSumReal_type = primme_op_double; }
   if (primme->broadcastReal) q->broadcastReal = cx_broadcast;
   if (primme->convTestFun) { q->convTestFun = cx_conv_test; q->convTestFun_type = primme_op_double; }
   if (primme->monitorFun) { q->monitorFun = cx_monitor; q->monitorFun_type = primme_op_double; }

   int ret = 0;
   hipk_ctx *ctx = NULL;
   char *work = NULL, *rot = NULL;
   double *d_s = NULL, *h_s = NULL, *evr = NULL, *rnr = NULL;
   void *user_queue = primme->queue;
   if (hipk_ctx_create(&ctx, primme->queue)) { free(sd); return PRIMME_UNEXPECTED_FAILURE; }
   void *stream = hipk_ctx_stream(ctx);
   q->queue = &stream;

   const int ncand = 2 * (nev > init ? nev : init);
   const size_t colB = (size_t)(ldr > 0 ? ldr : 1) * esr;
   CX(hipk_malloc(ctx, colB * (size_t)(2 * nOC + ncand + 1), (void **)&work));
   CX(hipk_malloc(ctx, colB * (size_t)(nev + 1), (void **)&rot));
   CX(hipk_malloc(ctx, sizeof(double) * (size_t)(4 * nev + 8), (void **)&d_s));
   CX(hipk_host_alloc(ctx, sizeof(double) * (size_t)(4 * nev + 8), (void **)&h_s));
   evr = (double *)calloc((size_t)2 * nev + 1, sizeof(double));
   rnr = (double *)calloc((size_t)2 * nev + 1, sizeof(double));
   if (!evr || !rnr) { ret = PRIMME_MALLOC_FAILURE; goto done; }

   /* constraints [Q | iQ], then the initial guesses [X0 | iX0] */
   CX(hipk_copy_cols(ctx, dtr, mr, evecs, ldr, work, ldr, nOC));
   CX(hipk_pair_rotate(ctx, dtr, mr / 2, evecs, ldr, work + colB * (size_t)nOC, ldr, nOC));
   CX(hipk_copy_cols(ctx, dtr, mr, (char *)evecs + colB * (size_t)nOC, ldr, work + colB * (size_t)(2 * nOC), ldr, init));
   CX(hipk_pair_rotate(ctx, dtr, mr / 2, (char *)evecs + colB * (size_t)nOC, ldr, work + colB * (size_t)(2 * nOC + init), ldr, init));
   CX(hipk_sync(ctx));

   ret = pa_eigs_solve(evr, work, rnr, q, dtr, 1);
   sync_user(sd);
   primme->dynamicMethodSwitch = q->dynamicMethodSwitch;
   memcpy(primme->iseed, q->iseed, sizeof(primme->iseed));
   primme->initSize = 0;
   if (ret != 0 && ret != PRIMME_MAIN_ITER_FAILURE) goto done;

   /* complex Gram-Schmidt sweep over the converged real pairs, in the solver's order */
   {
      const int nconv = q->initSize;
      char *Z = (char *)evecs + colB * (size_t)nOC;      /* accepted vectors, in place in the caller's array */
      int acc = 0;
      int *picked = (int *)calloc((size_t)nev + 1, sizeof(int));
      if (!picked) { ret = PRIMME_MALLOC_FAILURE; goto done; }
      const int rcs = pa_complex_sweep(ctx, dtr, mr, ldr, work + colB * (size_t)(2 * nOC), nconv, Z, rot, nev, d_s, h_s,
            (primme->numProcs > 1 && primme->globalSumReal) ? sum_eigs : NULL, primme, picked, &acc);
      if (rcs) { free(picked); ret = rcs; goto done; }
      for (int a = 0; a < acc; a++) {
         if (dtr == HIPK_F64) { ((double *)evals_out)[a] = evr[picked[a]]; ((double *)resNorms_out)[a] = rnr[picked[a]]; }
         else { ((float *)evals_out)[a] = (float)evr[picked[a]]; ((float *)resNorms_out)[a] = (float)rnr[picked[a]]; }
      }
      free(picked);
      if (hipk_sync(ctx)) { ret = PRIMME_UNEXPECTED_FAILURE; goto done; }
      primme->initSize = acc;
      /* (fewer than numEvals with ret = 0: the real solver exhausted the space, as dprimme does
       * for closest_geq / closest_leq when fewer eigenvalues exist on the wanted side) */

      /* the sweep may have changed the vectors (components along pairs already taken removed):
       * report the residual norms of what is returned, ||A z - lambda z|| */
      if (acc > 0) {
         int ierr = 0, nb = acc, cnt = acc;
         PRIMME_INT ldc = ldu;
         double *theta = (double *)malloc(sizeof(double) * (size_t)acc);
         if (!theta) { ret = PRIMME_MALLOC_FAILURE; goto done; }
         for (int i = 0; i < acc; i++) theta[i] = (dtr == HIPK_F64) ? ((double *)evals_out)[i] : (double)((float *)evals_out)[i];
         primme->queue = q->queue;
         primme->matrixMatvec(Z, &ldc, work, &ldc, &nb, primme, &ierr);
         if (ierr) { free(theta); ret = PRIMME_USER_FAILURE; goto done; }
         primme->stats.numMatvecs += acc;
         if (hipk_residual_cols(ctx, dtr, mr, Z, ldr, work, ldr, acc, theta, d_s) ||
             hipk_d2h(ctx, h_s, d_s, sizeof(double) * (size_t)acc) || hipk_sync(ctx)) { free(theta); ret = PRIMME_UNEXPECTED_FAILURE; goto done; }
         free(theta);
         if (primme->numProcs > 1 && primme->globalSumReal) {
            if (pa_call_global_sum(primme, h_s, cnt)) { ret = PRIMME_USER_FAILURE; goto done; }
         }
         for (int i = 0; i < acc; i++) {
            if (dtr == HIPK_F64) ((double *)resNorms_out)[i] = sqrt(h_s[i]);
            else ((float *)resNorms_out)[i] = (float)sqrt(h_s[i]);
         }
      }
   }

done:
   primme->queue = user_queue;
   primme->ShiftsForPreconditioner = NULL;
   free(evr); free(rnr);
   if (ctx) {
      if (h_s) hipk_host_free(ctx, h_s);
      if (d_s) hipk_free(ctx, d_s);
      if (rot) hipk_free(ctx, rot);
      if (work) hipk_free(ctx, work);
      hipk_ctx_destroy(ctx);
   }
   free(sd);
   return ret;
}

int hip_zprimme(double *evals, void *evecs, double *resNorms, primme_params *primme) {
   return solve_complex(evals, evecs, resNorms, primme, HIPK_F64);
}
int hip_cprimme(float *evals, void *evecs, float *resNorms, primme_params *primme) {
   return solve_complex(evals, evecs, resNorms, primme, HIPK_F32);
}
