/* svds_complex.c — hip_zprimme_svds / hip_cprimme_svds: the entry points, and the REAL-EQUIVALENT form of the complex singular
 * value problem (the default until round 4; since round 5 the native complex front end of svds_main.c is the default and this
 * form is the A/B alternative behind PRIMME_AMD_COMPLEX_REAL_FORM=1).
 *
 * Boundary: reference include/primme_svds.h:242-243, :268-271 (zprimme_svds / cprimme_svds and their GPU
 * flavours), front end src/svds/primme_svds_c.c:103-108 with SCALAR = complex.  Same construction as
 * eigs_complex.c for Hermitian problems: a complex n-vector in memory IS a real 2n-vector, x -> A x is a
 * real-linear map M of R^2n into R^2m whose transpose is x -> A^H x, every singular value of A is a singular
 * value of M with twice the multiplicity, and (u, v) and (i u, i v) are the two real triplets of one complex one.
 * The user's complex callbacks are applied, untouched, to the real solver's vectors; the real front end
 * (svds_main.c: normal equations, augmented, hybrid) is asked for 2 numSvals triplets; the complex-independent
 * right vectors are selected by the sweep of eigs_complex.c, the left ones follow as u = A v / sigma, and the
 * residual norms of what is returned are recomputed.
 *
 * Cost: twice the triplets; the operator-application counts are NOT those of zprimme_svds (the values, vectors and
 * residual norms are, to the tolerance).
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include "primme_amd.h"
#include "primme_amd_svds.h"
#include "primme_amd_kernels.h"
#include "eigs_internal.h"

typedef struct {
   primme_svds_params *user;   /* the caller's struct: what its callbacks expect to receive */
   primme_svds_params q;       /* the real problem of twice the size handed to the front end */
} csv_side;
#define CSV_OF(qq) ((csv_side *)((char *)(qq) - offsetof(csv_side, q)))

static void sync_user(csv_side *sd) {
   sd->user->queue = sd->q.queue;
   sd->user->aNorm = sd->q.aNorm;
}
static void cs_matvec(void *x, PRIMME_INT *ldx, void *y, PRIMME_INT *ldy, int *blockSize, int *transpose,
      primme_svds_params *qs, int *ierr) {
   csv_side *sd = CSV_OF(qs);
   PRIMME_INT lx = *ldx / 2, ly = *ldy / 2;
   sync_user(sd);
   sd->user->matrixMatvec(x, &lx, y, &ly, blockSize, transpose, sd->user, ierr);
}
static void cs_precond(void *x, PRIMME_INT *ldx, void *y, PRIMME_INT *ldy, int *blockSize, int *mode,
      primme_svds_params *qs, int *ierr) {
   csv_side *sd = CSV_OF(qs);
   PRIMME_INT lx = *ldx / 2, ly = *ldy / 2;
   sync_user(sd);
   sd->user->applyPreconditioner(x, &lx, y, &ly, blockSize, mode, sd->user, ierr);
}
static void cs_global_sum(void *s, void *r, int *count, primme_svds_params *qs, int *ierr) {
   csv_side *sd = CSV_OF(qs);
   if (s != r) memcpy(r, s, sizeof(double) * (size_t)*count);
   *ierr = pa_svds_call_global_sum(sd->user, (double *)r, *count) ? 1 : 0;
}
static void cs_broadcast(void *buf, int *count, primme_svds_params *qs, int *ierr) {
   csv_side *sd = CSV_OF(qs);
   sd->user->broadcastReal(buf, count, sd->user, ierr);
}
static void cs_conv_test(double *sval, void *leftsvec, void *rightsvec, double *rNorm, int *method, int *isconv,
      primme_svds_params *qs, int *ierr) {
   csv_side *sd = CSV_OF(qs);
   sync_user(sd);
   *ierr = pa_svds_call_conv_test(sd->user, *sval, leftsvec, rightsvec, *rNorm, method, isconv) ? 1 : 0;
}
static void cs_monitor(void *basisSvals, int *basisSize, int *basisFlags, int *iblock, int *blockSize, void *basisNorms,
      int *numConverged, void *lockedSvals, int *numLocked, int *lockedFlags, void *lockedNorms, int *inner_its,
      void *LSRes, const char *msg, double *time, primme_event *event, int *stage, primme_svds_params *qs, int *err) {
   csv_side *sd = CSV_OF(qs);
   sync_user(sd);
   sd->user->monitorFun(basisSvals, basisSize, basisFlags, iblock, blockSize, basisNorms, numConverged, lockedSvals,
         numLocked, lockedFlags, lockedNorms, inner_its, LSRes, msg, time, event, stage, sd->user, err);
}
static int sum_svds(void *who, double *buf, int count) { return pa_svds_call_global_sum((primme_svds_params *)who, buf, count); }

#define CX(call) do { int rc__ = (call); if (rc__) { ret = rc__ < 0 ? rc__ : PRIMME_UNEXPECTED_FAILURE; goto done; } } while (0)

int pa_svds_solve_native_complex(void *svals, void *svecs, void *resNorms, primme_svds_params *ps, int single);

static int solve_svds_complex(void *svals_out, void *svecs_, void *resNorms_out, primme_svds_params *ps, hipk_dtype dtr) {
   if (!ps) return -4;
   if (!svals_out && !svecs_ && !resNorms_out)       /* defaults query: the same for every precision */
      return hip_dprimme_svds(NULL, NULL, NULL, ps);
   /* Round 5: the NATIVE complex front end is the default — svds_main.c instantiated for complex panels over the native complex
    * eigensolver, like the reference's primme_svds_c.c for SCALAR = complex: numSvals triplets, the reference's operator counts.
    * The real-equivalent form below (twice the triplets, twice the bytes per application) stays behind PRIMME_AMD_COMPLEX_REAL_FORM=1
    * (A/B measurements, tests of both). */
   if (!getenv("PRIMME_AMD_COMPLEX_REAL_FORM")) {
      if (svecs_ && !hipk_is_device_ptr(svecs_)) return -18;
      return pa_svds_solve_native_complex(svals_out, svecs_, resNorms_out, ps, dtr == HIPK_F32);
   }
   if (ps->numProcs <= 1) { ps->mLocal = ps->m; ps->nLocal = ps->n; ps->procID = 0; ps->numProcs = 1; }
   primme_svds_set_defaults(ps);
   if (ps->n < 0 || ps->m < 0 || ps->nLocal < 0 || ps->mLocal < 0 || ps->nLocal > ps->n || ps->mLocal > ps->m) return -5;
   if (!ps->matrixMatvec) return -7;
   if (ps->numSvals > PA_MIN(ps->n, ps->m)) return -10;
   if (ps->numSvals < 1) return -11;
   if (!svals_out) return -17;
   if (!svecs_ || !hipk_is_device_ptr(svecs_)) return -18;
   if (!resNorms_out) return -19;

   csv_side *sd = (csv_side *)calloc(1, sizeof(csv_side));
   if (!sd) return PRIMME_MALLOC_FAILURE;
   sd->user = ps;
   sd->q = *ps;
   primme_svds_params *q = &sd->q;
   const int k = ps->numSvals, nOC = ps->numOrthoConst, init = ps->initSize;
   const int64_t mL = ps->mLocal, nL = ps->nLocal;             /* complex elements */
   const size_t esr = (dtr == HIPK_F64) ? 8 : 4, esc = 2 * esr;
   q->m = 2 * ps->m; q->n = 2 * ps->n; q->mLocal = 2 * mL; q->nLocal = 2 * nL;
   q->numSvals = 2 * k; q->numOrthoConst = 2 * nOC; q->initSize = 2 * init;
   /* sizes the stage structs derived from the complex problem are derived again */
   q->primme.nLocal = -1; q->primmeStage2.nLocal = -1;
   q->primme.ldOPs = -1; q->primmeStage2.ldOPs = -1; q->primme.ldevecs = -1; q->primmeStage2.ldevecs = -1;
   if (ps->locking < 0) { q->primme.locking = -1; q->primmeStage2.locking = -1; }   /* depends on the doubled count */
   if (ps->maxBasisSize == 0) {
      q->primme.maxBasisSize = 0; q->primme.minRestartSize = 0;
      q->primmeStage2.maxBasisSize = 0; q->primmeStage2.minRestartSize = 0;
   }
   q->matrixMatvec = cs_matvec;
   if (ps->applyPreconditioner) q->applyPreconditioner = cs_precond;
   if (ps->globalSumReal && ps->globalSumReal != primme_amd_svds_global_sum) { q->globalSumReal = cs_global_sum; q->globalSumReal_type = primme_op_double; }
   if (ps->broadcastReal) q->broadcastReal = cs_broadcast;
   if (ps->convTestFun) { q->convTestFun = cs_conv_test; q->convTestFun_type = primme_op_double; }
   if (ps->monitorFun) { q->monitorFun = cs_monitor; q->monitorFun_type = primme_op_double; }
   primme_svds_set_defaults(q);

   int ret = 0;
   hipk_ctx *ctx = NULL;
   char *work = NULL, *Zv = NULL, *rot = NULL, *Uz = NULL, *Wv = NULL, *Vc = NULL;
   double *d_s = NULL, *h_s = NULL, *svr = NULL, *rnr = NULL;
   int *picked = NULL;
   void *user_queue = ps->queue;
   if (hipk_ctx_create(&ctx, ps->queue)) { free(sd); return PRIMME_UNEXPECTED_FAILURE; }
   void *stream = hipk_ctx_stream(ctx);
   q->queue = &stream;

   char *svecs = (char *)svecs_;
   const int nMaxC = PA_MAX(init, k) + nOC, nMaxR = 2 * nMaxC;
   const size_t colU = (size_t)(mL > 0 ? mL : 1) * esc, colV = (size_t)(nL > 0 ? nL : 1) * esc;   /* bytes per column */
   CX(hipk_malloc(ctx, (colU + colV) * (size_t)nMaxR + 64, (void **)&work));
   CX(hipk_malloc(ctx, colV * (size_t)(k + 1), (void **)&Zv));
   CX(hipk_malloc(ctx, colV * (size_t)(k + 1), (void **)&rot));
   CX(hipk_malloc(ctx, colV * (size_t)(k + 1), (void **)&Wv));
   CX(hipk_malloc(ctx, colU * (size_t)(k + 1), (void **)&Uz));
   CX(hipk_malloc(ctx, colV * (size_t)(nOC + 1), (void **)&Vc));
   CX(hipk_malloc(ctx, sizeof(double) * (size_t)(4 * k + 8), (void **)&d_s));
   CX(hipk_host_alloc(ctx, sizeof(double) * (size_t)(4 * k + 8), (void **)&h_s));
   svr = (double *)calloc((size_t)2 * k + 1, sizeof(double));
   rnr = (double *)calloc((size_t)2 * k + 1, sizeof(double));
   picked = (int *)calloc((size_t)k + 1, sizeof(int));
   if (!svr || !rnr || !picked) { ret = PRIMME_MALLOC_FAILURE; goto done; }

   {
      /* real-equivalent input: constraints [C | iC], then the initial guesses [X0 | iX0], on both sides */
      char *Ur = work, *Vr = work + colU * (size_t)nMaxR;
      const char *Uin = svecs, *Vin = svecs + colU * (size_t)nMaxC;
      CX(hipk_memset0(ctx, work, (colU + colV) * (size_t)nMaxR));
      CX(hipk_copy_cols(ctx, dtr, 2 * mL, Uin, 2 * mL, Ur, 2 * mL, nOC));
      CX(hipk_pair_rotate(ctx, dtr, mL, Uin, 2 * mL, Ur + colU * (size_t)nOC, 2 * mL, nOC));
      CX(hipk_copy_cols(ctx, dtr, 2 * mL, Uin + colU * (size_t)nOC, 2 * mL, Ur + colU * (size_t)(2 * nOC), 2 * mL, init));
      CX(hipk_pair_rotate(ctx, dtr, mL, Uin + colU * (size_t)nOC, 2 * mL, Ur + colU * (size_t)(2 * nOC + init), 2 * mL, init));
      CX(hipk_copy_cols(ctx, dtr, 2 * nL, Vin, 2 * nL, Vr, 2 * nL, nOC));
      CX(hipk_pair_rotate(ctx, dtr, nL, Vin, 2 * nL, Vr + colV * (size_t)nOC, 2 * nL, nOC));
      CX(hipk_copy_cols(ctx, dtr, 2 * nL, Vin + colV * (size_t)nOC, 2 * nL, Vr + colV * (size_t)(2 * nOC), 2 * nL, init));
      CX(hipk_pair_rotate(ctx, dtr, nL, Vin + colV * (size_t)nOC, 2 * nL, Vr + colV * (size_t)(2 * nOC + init), 2 * nL, init));
      CX(hipk_copy_cols(ctx, dtr, 2 * nL, Vin, 2 * nL, Vc, 2 * nL, nOC));         /* the right constraints move on output */
      CX(hipk_sync(ctx));
   }

   ret = (dtr == HIPK_F64) ? hip_dprimme_svds(svr, (double *)work, rnr, q) : -44;
   if (dtr == HIPK_F32) {
      float *sf = (float *)svr, *rf = (float *)rnr;     /* the float front end writes floats: widen in place afterwards */
      ret = hip_sprimme_svds(sf, (float *)work, rf, q);
      for (int i = 2 * k - 1; i >= 0; i--) { rnr[i] = (double)rf[i]; }
      for (int i = 2 * k - 1; i >= 0; i--) { svr[i] = (double)sf[i]; }
   }
   sync_user(sd);
   ps->stats = q->stats;
   ps->primme.stats = q->primme.stats; ps->primmeStage2.stats = q->primmeStage2.stats;
   memcpy(ps->iseed, q->iseed, sizeof(ps->iseed));
   ps->initSize = 0;
   if (ret != 0 && ret != PRIMME_MAIN_ITER_FAILURE - 100 && ret != PRIMME_MAIN_ITER_FAILURE - 200) goto done;

   {
      const int nconv = q->initSize, n1R = nconv + 2 * nOC;
      char *Vr = work + colU * (size_t)n1R;                         /* packed right after the left block */
      int acc = 0;
      const int rcs = pa_complex_sweep(ctx, dtr, 2 * nL, 2 * nL, Vr + colV * (size_t)(2 * nOC), nconv, Zv, rot, k, d_s, h_s,
            (ps->numProcs > 1 && ps->globalSumReal) ? sum_svds : NULL, ps, picked, &acc);
      if (rcs) { ret = rcs; goto done; }
      if (acc > 0) {
         /* u = A v / sigma (the normalised left candidate of the real solve where sigma is not usable), and the
          * residual of what is returned: A v - sigma u = 0 by construction, so ||A^H u - sigma v|| is all of it */
         int ierr = 0, nb = acc, tr = 0;
         PRIMME_INT lv = nL, lu = mL;
         double *sig = (double *)malloc(sizeof(double) * (size_t)acc), *inv = (double *)malloc(sizeof(double) * (size_t)acc);
         if (!sig || !inv) { free(sig); free(inv); ret = PRIMME_MALLOC_FAILURE; goto done; }
         ps->queue = q->queue;
         ps->matrixMatvec(Zv, &lv, Uz, &lu, &nb, &tr, ps, &ierr);
         if (ierr) { free(sig); free(inv); ret = PRIMME_USER_FAILURE; goto done; }
         for (int a = 0; a < acc; a++) {
            sig[a] = svr[picked[a]];
            if (sig[a] > 0.0 && 1.0 / sig[a] < 1.79e308) inv[a] = 1.0 / sig[a];
            else {
               inv[a] = 1.0;
               if (hipk_copy_cols(ctx, dtr, 2 * mL, work + colU * (size_t)(2 * nOC + picked[a]), 2 * mL, Uz + colU * (size_t)a, 2 * mL, 1)) {
                  free(sig); free(inv); ret = PRIMME_UNEXPECTED_FAILURE; goto done;
               }
            }
         }
         if (hipk_scale_cols(ctx, dtr, 2 * mL, Uz, 2 * mL, acc, inv)) { free(sig); free(inv); ret = PRIMME_UNEXPECTED_FAILURE; goto done; }
         tr = 1;
         ps->matrixMatvec(Uz, &lu, Wv, &lv, &nb, &tr, ps, &ierr);
         if (ierr) { free(sig); free(inv); ret = PRIMME_USER_FAILURE; goto done; }
         ps->stats.numMatvecs += 2 * acc;
         if (hipk_residual_cols(ctx, dtr, 2 * nL, Zv, 2 * nL, Wv, 2 * nL, acc, sig, d_s) ||
             hipk_d2h(ctx, h_s, d_s, sizeof(double) * (size_t)acc) || hipk_sync(ctx)) { free(sig); free(inv); ret = PRIMME_UNEXPECTED_FAILURE; goto done; }
         if (ps->numProcs > 1 && ps->globalSumReal && pa_svds_call_global_sum(ps, h_s, acc)) { free(sig); free(inv); ret = PRIMME_USER_FAILURE; goto done; }
         for (int a = 0; a < acc; a++) {
            if (dtr == HIPK_F64) { ((double *)svals_out)[a] = sig[a]; ((double *)resNorms_out)[a] = sqrt(h_s[a]); }
            else { ((float *)svals_out)[a] = (float)sig[a]; ((float *)resNorms_out)[a] = (float)sqrt(h_s[a]); }
         }
         free(sig); free(inv);
      }
      /* output layout [U: constraints, found | V: constraints, found], each block acc + numOrthoConst columns */
      char *Vout = svecs + colU * (size_t)(nOC + acc);
      CX(hipk_copy_cols(ctx, dtr, 2 * mL, Uz, 2 * mL, svecs + colU * (size_t)nOC, 2 * mL, acc));
      CX(hipk_copy_cols(ctx, dtr, 2 * nL, Vc, 2 * nL, Vout, 2 * nL, nOC));
      CX(hipk_copy_cols(ctx, dtr, 2 * nL, Zv, 2 * nL, Vout + colV * (size_t)nOC, 2 * nL, acc));
      CX(hipk_sync(ctx));
      ps->initSize = acc;
   }

done:
   ps->queue = user_queue;
   free(svr); free(rnr); free(picked);
   if (ctx) {
      if (h_s) hipk_host_free(ctx, h_s);
      hipk_free(ctx, d_s); hipk_free(ctx, Vc); hipk_free(ctx, Uz); hipk_free(ctx, Wv); hipk_free(ctx, rot); hipk_free(ctx, Zv); hipk_free(ctx, work);
      hipk_ctx_destroy(ctx);
   }
   free(sd);
   return ret;
}

int hip_zprimme_svds(double *svals, void *svecs, double *resNorms, primme_svds_params *ps) {
   return solve_svds_complex(svals, svecs, resNorms, ps, HIPK_F64);
}
int hip_cprimme_svds(float *svals, void *svecs, float *resNorms, primme_svds_params *ps) {
   return solve_svds_complex(svals, svecs, resNorms, ps, HIPK_F32);
}
