/* eigs_main.c — hip_?primme entry points and the outer block-Davidson loop.
 *
 *   hip_dprimme / front end   <- reference src/eigs/primme_c.c:103-108, :277-422
 *   main loop                 <- reference src/eigs/main_iter.c:175-1414
 *   init_basis / block Krylov <- reference src/eigs/init.c:125-323
 *   GD correction             <- reference src/eigs/correction.c:335-381
 *   verify_norms              <- reference src/eigs/main_iter.c:1864-1897
 *   copy_back_candidates      <- reference src/eigs/main_iter.c:1743-1832
 *
 * Scope of this translation unit: Hermitian standard problem, Rayleigh-Ritz
 * extraction, Generalized-Davidson family (maxInnerIterations == 0: GD, GD+k,
 * GD_Olsen+k, LOBPCG-like presets), hard and soft locking, all targets.
 * The control flow is the reference's; every n-length operation is a launch of
 * the device layer on panels that stay in HBM for the whole solve.
 */
#include "eigs_solver.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>

double pa_problem_norm(int overrideUser, const primme_params *p);
int pa_reduce(pa_solver *s, double *d_buf, int count, int keep_dev, int defer_sync);
int pa_refresh_wtq(pa_solver *s, int basisSize, int nLk);
int pa_block_first_reorder(pa_solver *s, int basisSize, int *flags, const int *iev, int blockSize, int numConverged, int numLocked);
int pa_matvec(pa_solver *s, char *Vp, int64_t ldV, char *Wp, int64_t ldW, int c0, int nc);
int pa_precond(pa_solver *s, char *X, int64_t ldX, char *Y, int64_t ldY, int nc);
int pa_random_col(pa_solver *s, char *col);
int pa_ortho_cgs(pa_solver *s, char *Vp, int64_t ldV, int b1, int b2, char *locked,
      int64_t ldLocked, int numLocked, HS *RLocked, int ldRLocked, int *b2_out);
int pa_update_projection(pa_solver *s, int numCols, int blockSize);
int pa_solve_H(pa_solver *s, int basisSize, int numLocked, int numConverged);
int pa_push_coefficients(pa_solver *s, int basisSize, int ldh);
int pa_ritz_update(pa_solver *s, int basisSize, const hipk_job *jobs, int njobs, double *norms_out,
      int nslots, int64_t flop_cols);
void pa_conv_test_absolute(double *eval, void *evec, double *rNorm, int *isConv,
      primme_params *p, int *ierr);
int pa_check_convergence(pa_solver *s, char *X, int64_t ldX, int givenX, char *R, int64_t ldR,
      int givenR, int numLocked, int left, int right, int *flags, double *blockNorms,
      const double *hVals, int *reset, int practConvCheck);
void pa_map_vecs(const HS *Vp, int mrows, int nV, int ldV, const HS *Wn, int n0, int n,
      int ldW, int *pm);
void pa_monitor(pa_solver *s, double *basisEvals, int basisSize, int *basisFlags, int *iblock,
      int blockSize, double *basisNorms, int numConverged, double *lockedEvals, int numLocked,
      int *lockedFlags, double *lockedNorms, primme_event event);
int pa_prepare_candidates(pa_solver *s, int basisSize, char *X, char *R, int computeXR,
      int *flags, int remainedEvals, double *blockNorms, int blockNormsSize, int maxBlockSize,
      int numLocked, double *evals, double *resNorms, int *iev, int *blockSize,
      int *recentlyConverged, double *smallestResNorm, int numConverged, double *basisNorms,
      int *reset, int nprevhVecs, int practConvChecking, int *map);
int pa_restart(pa_solver *s, int basisSize, int *flags, int *iev, int *ievSize, double *blockNorms,
      int *evecsPerm, double *evals, double *resNorms, int *numConverged, int *numLocked,
      int *lockedFlags, int nprevhVecs, int numGuesses, int *restartSizeOutput,
      int *restartsSinceReset);

/* ---- block orthogonalisation dispatcher (reference ortho.c:429-439, :522-530):
 *      implicit_I -> vector-by-vector CGS. ----------------------------------------- */
int pa_ortho_block_gram(pa_solver *s, char *Vp, int64_t ldV, int b1, int b2, char *locked,
      int64_t ldLocked, int numLocked, HS *RLocked, int ldRLocked, int maxRank, int *b2_out);

static int ortho_block(pa_solver *s, char *Vp, int64_t ldV, int b1, int b2, char *locked,
      int64_t ldLocked, int numLocked, HS *RLocked, int ldRLocked, int *b2_out) {
   if (b2 < b1) { *b2_out = b2 + 1; return 0; }
   if (s->VtBV)   /* explicit_I: iterative CholQR/SVQB with the tracked Gram matrix */
      return pa_ortho_block_gram(s, Vp, ldV, b1, b2, locked, ldLocked, numLocked, RLocked, ldRLocked, s->maxRank, b2_out);
   return pa_ortho_cgs(s, Vp, ldV, b1, b2, locked, ldLocked, numLocked, RLocked, ldRLocked, b2_out);
}

/* ---- initial basis ------------------------------------------------------------- */
static int init_block_krylov(pa_solver *s, int dv1, int dv2) {
   primme_params *p = s->p;
   const int numNew = dv2 - dv1 + 1;
   if (numNew <= 0) return 0;
   const int blockSize = numNew <= p->maxBlockSize ? 1 : p->maxBlockSize;
   for (int i = dv1; i < dv1 + blockSize; i++) CHK(pa_random_col(s, VCOL(s, i)));
   int nV = 0;
   CHK(ortho_block(s, s->V, s->ld, dv1, dv1 + blockSize - 1, s->evecs, s->ldevecs, p->numOrthoConst, NULL, 0, &nV));
   if (nV != dv1 + blockSize) return PRIMME_UNEXPECTED_FAILURE;
   int mm = blockSize;
   for (int i = dv1 + blockSize; i <= dv2; i += mm) {
      mm = PA_MIN(mm, dv2 - i + 1);
      /* V(:,i..) = A V(:,i-bs..), also stored as W(:,i-bs..) */
      CHK(pa_matvec(s, s->V, s->ld, s->V + (size_t)blockSize * s->ld * s->es, s->ld, i - blockSize, mm));
      CHK(hipk_copy_cols(s->ctx, s->dt, s->m, VCOL(s, i), s->ld, WCOL(s, i - blockSize), s->ld, mm));
      CHK(ortho_block(s, s->V, s->ld, i, i + mm - 1, s->evecs, s->ldevecs, p->numOrthoConst, NULL, 0, &nV));
      for (int j = nV; j < i + mm; j++) CHK(pa_random_col(s, VCOL(s, j)));
      CHK(ortho_block(s, s->V, s->ld, nV, i + mm - 1, s->evecs, s->ldevecs, p->numOrthoConst, NULL, 0, &nV));
      if (nV != i + mm) return PRIMME_UNEXPECTED_FAILURE;
   }
   CHK(pa_matvec(s, s->V, s->ld, s->W, s->ld, dv2 - blockSize + 1, blockSize));
   return 0;
}

static int init_basis(pa_solver *s, int *basisSize, int *nextGuess, int *numGuesses) {
   primme_params *p = s->p;
   if (p->numOrthoConst > 0) {
      int nV = 0;
      CHK(ortho_block(s, s->evecs, s->ldevecs, 0, p->numOrthoConst - 1, NULL, 0, 0, NULL, 0, &nV));
      if (nV != p->numOrthoConst) return PRIMME_ORTHO_CONST_FAILURE;
   }
   int initSize = p->locking ? PA_MIN(p->minRestartSize, p->initSize) : PA_MIN(p->maxBasisSize, p->initSize);
   initSize = (int)PA_MAX(0, PA_MIN(p->n - p->numOrthoConst, (int64_t)initSize));
   *numGuesses = p->initSize - initSize;
   *nextGuess = p->numOrthoConst + initSize;
   CHK(hipk_copy_cols(s->ctx, s->dt, s->m, ECOL(s, p->numOrthoConst), s->ldevecs, s->V, s->ld, initSize));

   int nrandom = 0;
   switch (p->initBasisMode) {
   case primme_init_krylov: nrandom = 0; break;
   case primme_init_random: nrandom = PA_MAX(0, p->minRestartSize - initSize); break;
   case primme_init_user: nrandom = PA_MAX(p->maxBlockSize - initSize, 0); break;
   default: return PRIMME_UNEXPECTED_FAILURE;
   }
   nrandom = (int)PA_MAX(0, PA_MIN(p->n - p->numOrthoConst - initSize, (int64_t)nrandom));
   for (int i = 0; i < nrandom; i++) CHK(pa_random_col(s, VCOL(s, initSize + i)));
   *basisSize = initSize + nrandom;
   CHK(ortho_block(s, s->V, s->ld, 0, *basisSize - 1, s->evecs, s->ldevecs, p->numOrthoConst, NULL, 0, basisSize));
   CHK(pa_matvec(s, s->V, s->ld, s->W, s->ld, 0, *basisSize));
   if (p->initBasisMode == primme_init_krylov) {
      int minRestartSize = (int)PA_MIN((int64_t)p->minRestartSize, p->n - p->numOrthoConst);
      CHK(init_block_krylov(s, *basisSize, minRestartSize - 1));
      *basisSize = minRestartSize;
   }
   return 0;
}

/* merge the sorted locked values with the current Ritz values (reference
 * correction.c:700-760 mergeSort): sortedRitzVals and, for each block vector, its
 * position ilev in the merged list. */
static void merge_sort(const double *lockedEvals, int numLocked, const double *ritzVals,
      const int *flags, int basisSize, double *sorted, int *ilev, int blockSize,
      const primme_params *p) {
   int li = 0, ri = 0, si = 0, bi = 0;
   while (li < numLocked || ri < basisSize) {
      int takeRitz;
      if (li >= numLocked) takeRitz = 1;
      else if (ri >= basisSize) takeRitz = 0;
      else if (p->target == primme_smallest) takeRitz = ritzVals[ri] <= lockedEvals[li];
      else takeRitz = ritzVals[ri] >= lockedEvals[li];
      if (takeRitz) {
         sorted[si] = ritzVals[ri];
         if (bi < blockSize && flags[ri] == UNCONV) ilev[bi++] = si;
         ri++;
      } else {
         sorted[si] = lockedEvals[li++];
      }
      si++;
   }
}

/* GD correction: t = K^-1 (r - eps x) (approximate Olsen when RightX) or K^-1 r;
 * without a preconditioner a device copy.  Writes the correction over X. */
void pa_dyn_init(pa_cost_model *c, const primme_params *p);
int pa_dyn_observe(pa_cost_model *c, primme_params *p, double now, int recentConv, int atRestart,
      int numConverged, double currentResNorm);
int pa_dyn_leave_gd(pa_solver *s, pa_cost_model *c);
int pa_dyn_leave_jdqmr(pa_solver *s, pa_cost_model *c);
void pa_dyn_recommend(const pa_cost_model *c, primme_params *p);
int pa_update_Q(pa_solver *s, double tau, int col0, int bs, int *nQ);
int pa_update_QtV(pa_solver *s, int col0, int bs);
int pa_prepare_vecs(pa_solver *s, int basisSize, int i0, int blockSize, int *arbitraryVecs, double smallestResNorm,
      const int *flags, int RRForAll);
int pa_evecs_hat_init(pa_solver *s);
int pa_evecs_hat_update(pa_solver *s, int *numConvergedStored, int numConverged);
int pa_correction_jdqmr(pa_solver *s, int basisSize, int blockSize, const double *blockNorms, const int *iev,
      double *shifts, int numLocked, int numConvergedStored, int *touch);

static int solve_correction_gd(pa_solver *s, const double *lockedEvals, int numLocked, int *flags,
      int basisSize, double *blockNorms, int *iev, int blockSize, int numConvergedStored, int *touch) {
   primme_params *p = s->p;
   if (blockSize <= 0) return 0;
   double *shifts = (double *)malloc((size_t)blockSize * sizeof(double));
   double *olsen = (double *)malloc((size_t)blockSize * sizeof(double));
   HS *colsen = (HS *)malloc((size_t)blockSize * sizeof(HS));      /* axpy factors in the panels' scalar type */
   double *sorted = s->hVals;
   int *ilev = iev, own = 0;
   if (!shifts || !olsen || !colsen) return PRIMME_MALLOC_FAILURE;
   const int extremal = (p->target == primme_smallest || p->target == primme_largest);
   if (p->locking && extremal) {
      sorted = (double *)malloc((size_t)(numLocked + basisSize) * sizeof(double));
      ilev = (int *)malloc((size_t)blockSize * sizeof(int));
      own = 1;
      for (int b = 0; b < blockSize; b++) ilev[b] = 0;
      merge_sort(lockedEvals, numLocked, s->hVals, flags, basisSize, sorted, ilev, blockSize, p);
   }
   if (!extremal) {
      const double targetShift = p->numTargetShifts > 0
            ? p->targetShifts[PA_MIN(p->numTargetShifts - 1, numLocked)] : 0.0;
      for (int b = 0; b < blockSize; b++) {
         const int si = ilev[b];
         const double bn = blockNorms[b] * sqrt(p->stats.estimateInvBNorm);
         if (fabs(sorted[si] - targetShift) < bn) shifts[b] = targetShift;
         else if (s->refined) shifts[b] = sorted[si];    /* |theta - tau| <= sigma: trust the Ritz value */
         else shifts[b] = sorted[si] + bn * (targetShift - sorted[si]) / fabs(targetShift - sorted[si]);
         olsen[b] = (si < s->numPrevRitzVals) ? fabs(s->prevRitzVals[si] - sorted[si]) : bn;
      }
      s->numPrevRitzVals = basisSize;
      memcpy(s->prevRitzVals, sorted, (size_t)basisSize * sizeof(double));
   } else {
      for (int b = 0; b < blockSize; b++) {
         const int si = ilev[b];
         if (p->correctionParams.robustShifts) {
            /* robust shift from the Davis-Kahan bounds (reference correction.c:525-613) */
            const int nS = numLocked + basisSize;
            const double rn = blockNorms[b];
            double eps1;
            if (p->stats.numOuterIterations <= 1) {
               eps1 = olsen[b] = rn * sqrt(p->stats.estimateInvBNorm);
            } else {
               double gap, lowerGap, delta;
               if (si == 0 && nS >= 2) {
                  lowerGap = 1.79769313486231571e+308;
                  gap = fabs(sorted[1] - sorted[0]);
               } else if (si > 0 && nS >= 2 && si + 1 < nS) {
                  lowerGap = fabs(sorted[si] - sorted[si - 1]);
                  gap = PA_MIN(lowerGap, fabs(sorted[si + 1] - sorted[si]));
               } else {
                  lowerGap = (si > 0) ? fabs(sorted[si] - sorted[si - 1]) : 1.79769313486231571e+308;
                  gap = lowerGap;
               }
               delta = (si < s->numPrevRitzVals) ? fabs(s->prevRitzVals[si] - sorted[si]) : 1.79769313486231571e+308;
               if (gap > rn) eps1 = PA_MIN(delta, PA_MIN(rn * rn * p->stats.estimateInvBNorm / gap, lowerGap));
               else eps1 = PA_MIN(rn * sqrt(p->stats.estimateInvBNorm), lowerGap);
               olsen[b] = PA_MIN(delta, eps1);
            }
            double sh = (p->target == primme_smallest) ? sorted[si] - eps1 : sorted[si] + eps1;
            if (si > 0) sh = (p->target == primme_smallest) ? PA_MAX(sh, sorted[si - 1]) : PA_MIN(sh, sorted[si - 1]);
            shifts[b] = sh;
         } else {
            shifts[b] = s->hVals[iev[b]];
            olsen[b] = (si < s->numPrevRitzVals) ? fabs(s->prevRitzVals[si] - sorted[si])
                                                 : blockNorms[b] * sqrt(p->stats.estimateInvBNorm);
         }
      }
      s->numPrevRitzVals = numLocked + basisSize;
      memcpy(s->prevRitzVals, sorted, (size_t)s->numPrevRitzVals * sizeof(double));
   }
   p->ShiftsForPreconditioner = shifts;

   char *r = WCOL(s, basisSize), *x = VCOL(s, basisSize);
   int rc = 0;
   if (p->correctionParams.maxInnerIterations != 0) {
      /* JDQMR: inner-outer iteration (reference correction.c:385-467) */
      rc = pa_correction_jdqmr(s, basisSize, blockSize, blockNorms, iev, shifts, numLocked, numConvergedStored, touch);
   } else if (p->correctionParams.projectors.RightX && p->correctionParams.projectors.SkewX) {
      /* exact Olsen: x <- K^-1 r - (x'K^-1 r / x'K^-1 x) K^-1 x  (reference correction.c:718-777);
       * K^-1 [x r] live in the scratch panel */
      if (2 * blockSize > s->nT) rc = PRIMME_UNEXPECTED_FAILURE;
      char *Kx = s->T, *Kr = TCOL(s, blockSize);
      char *xm = x;                                /* K^-1 B x for a generalised problem (correction.c:738-741) */
      if (s->B) {
         if (blockSize > s->nBT) rc = PRIMME_FUNCTION_UNAVAILABLE;
         if (!rc) rc = pa_apply_B(s, x, s->ld, s->BT, s->ld, blockSize);
         xm = s->BT;
      }
      if (!rc) rc = pa_precond(s, xm, s->ld, Kx, s->ld, blockSize);
      if (!rc) rc = pa_precond(s, r, s->ld, Kr, s->ld, blockSize);
      if (!rc) rc = hipk_pair_dots(s->ctx, s->dt, s->m, x, s->ld, Kx, s->ld, blockSize, s->d_red);
      if (!rc) rc = hipk_pair_dots(s->ctx, s->dt, s->m, x, s->ld, Kr, s->ld, blockSize, s->d_red + SD * blockSize);
      if (!rc) rc = pa_reduce(s, s->d_red, SD * 2 * blockSize, 0, 0);
      if (!rc) {
         const HS *hr = (const HS *)s->h_red;
         for (int b = 0; b < blockSize; b++)
            colsen[b] = (HS_ABS(hr[b]) > 0.0) ? -hr[blockSize + b] / hr[b] : 0.0;
         rc = hipk_copy_cols(s->ctx, s->dt, s->m, Kr, s->ld, x, s->ld, blockSize);
         if (!rc) rc = hipk_axpy_cols(s->ctx, s->dt, s->m, (const double *)colsen, Kx, s->ld, x, s->ld, blockSize);
      }
   } else {
      if (p->correctionParams.projectors.RightX &&
            ((p->correctionParams.precondition && p->applyPreconditioner) || s->B ||      /* (`Bx != x`, correction.c:359) */
                  (p->locking && p->orth == primme_orth_implicit_I))) {
         for (int b = 0; b < blockSize; b++) colsen[b] = -olsen[b];
         const char *xm = x;                       /* r -= eps B x for a generalised problem (correction.c:340-352) */
         if (s->B) {
            if (blockSize > s->nBT) rc = PRIMME_FUNCTION_UNAVAILABLE;
            if (!rc) rc = pa_apply_B(s, x, s->ld, s->BT, s->ld, blockSize);
            xm = s->BT;
         }
         if (!rc) rc = hipk_axpy_cols(s->ctx, s->dt, s->m, (const double *)colsen, xm, s->ld, r, s->ld, blockSize);
      }
      if (!rc && !s->fuse_gd) rc = pa_precond(s, r, s->ld, x, s->ld, blockSize);
   }
   p->ShiftsForPreconditioner = NULL;
   if (own) { free(sorted); free(ilev); }
   free(shifts);
   free(olsen);
   free(colsen);
   return rc;
}

/* residual norms of the first nb basis vectors taken as Ritz vectors (soft locking,
 * just after a restart: V holds Ritz vectors, W = A V) */
static int verify_norms(pa_solver *s, int nb, double *resNorms, int *flags, int *numConverged) {
   if (nb > 0) {
      const char *Vm = s->V;      /* generalised problem: the residual is W - theta B V */
      if (s->B) {
         if (nb > s->nBT) return PRIMME_FUNCTION_UNAVAILABLE;
         CHK(pa_apply_B(s, s->V, s->ld, s->BT, s->ld, nb));
         Vm = s->BT;
      }
      CHK(hipk_residual_cols(s->ctx, s->dt, s->m, Vm, s->ld, s->W, s->ld, nb, s->hVals, s->d_red));
      CHK(pa_reduce(s, s->d_red, nb, 0, 0));
      for (int i = 0; i < nb; i++) resNorms[i] = sqrt(s->h_red[i]);
      CHK(pa_check_convergence(s, s->V, s->ld, 1, s->W, s->ld, 1, 0, 0, nb, flags, resNorms, s->hVals, NULL, 0));
   }
   int i;
   for (i = 0; i < nb && flags[i] != UNCONV; i++) ;
   *numConverged = i;
   return 0;
}

static int copy_back_candidates(pa_solver *s, int basisSize, double *evals, double *resNorms,
      int numConverged, int *numRet) {
   primme_params *p = s->p;
   if (numConverged >= p->numEvals || basisSize <= 0) return 0;
   int i = 0;
   CHK(pa_push_coefficients(s, basisSize, basisSize));
   while (i < basisSize && numConverged < p->numEvals) {
      int blockSize = PA_MAX(0, PA_MIN(p->numEvals - numConverged, basisSize - i));
      blockSize = PA_MIN(blockSize, 8);
      hipk_job jobs[16];
      double norms[8];
      int nj = 0;
      for (int c = 0; c < blockSize; c++) {
         jobs[nj++] = (hipk_job){HIPK_JOB_XV, i + c, ECOL(s, p->numOrthoConst + numConverged + c), -1};
         jobs[nj++] = (hipk_job){HIPK_JOB_RES, i + c, NULL, c};
      }
      CHK(pa_ritz_update(s, basisSize, jobs, nj, norms, blockSize, 2 * blockSize));
      const int nc0 = numConverged;
      const double targetShift = p->targetShifts ? p->targetShifts[s->targetShiftIndex < 0 ? 0 : s->targetShiftIndex] : 0.0;
      for (int b = 0; b < blockSize; b++, i++) {
         if ((p->target == primme_closest_leq && s->hVals[i] - norms[b] > targetShift) ||
               (p->target == primme_closest_geq && s->hVals[i] + norms[b] < targetShift)) continue;
         evals[numConverged] = s->hVals[i];
         resNorms[numConverged] = norms[b];
         if (nc0 + b != numConverged)
            CHK(hipk_copy_cols(s->ctx, s->dt, s->m, ECOL(s, p->numOrthoConst + nc0 + b), s->ldevecs,
                  ECOL(s, p->numOrthoConst + numConverged), s->ldevecs, 1));
         numConverged++;
      }
   }
   for (i = numConverged; i < p->numEvals; i++) resNorms[i] = -1;
   *numRet = numConverged;
   return 0;
}

/* in-place column permutation of a device panel: new column i = old column perm[i] */
static int permute_dev_cols(pa_solver *s, char *base, int64_t ldb, int n, const int *perm) {
   /* new column i = old column perm[i] */
   int moved = 0;
   for (int i = 0; i < n; i++) if (perm[i] != i) moved = 1;
   if (!moved) return 0;
   unsigned char *done = (unsigned char *)calloc((size_t)n, 1);
   if (!done) return PRIMME_MALLOC_FAILURE;
   int is_perm = 1;
   for (int i = 0; i < n && is_perm; i++) {
      if (perm[i] < 0 || perm[i] >= n || done[perm[i]]) is_perm = 0;
      else done[perm[i]] = 1;
   }
   int rc = 0;
   if (!is_perm) {
      /* a gather with repeated sources: through a copy of the columns */
      free(done);
      for (int i = 0; i < n; i++) if (perm[i] < 0 || perm[i] >= n) return PRIMME_UNEXPECTED_FAILURE;
      char *tmp = NULL;
      const size_t colB = (size_t)ldb * s->es;
      if (n <= s->nT && ldb == s->ld) tmp = s->T;
      else if (hipk_malloc(s->ctx, colB * (size_t)n, (void **)&tmp)) return PRIMME_MALLOC_FAILURE;
      rc = hipk_copy_cols(s->ctx, s->dt, s->m, base, ldb, tmp, ldb, n);
      for (int i = 0; i < n && !rc; i++)
         rc = hipk_copy_cols(s->ctx, s->dt, s->m, tmp + colB * (size_t)perm[i], ldb, PCOL(s, base, ldb, i), ldb, 1);
      if (tmp != s->T) { if (!rc) rc = hipk_sync(s->ctx); hipk_free(s->ctx, tmp); }
      return rc ? (rc < 0 ? rc : PRIMME_UNEXPECTED_FAILURE) : 0;
   }
   /* a permutation: in place along its cycles with one scratch column */
   memset(done, 0, (size_t)n);
   for (int i = 0; i < n && !rc; i++) {
      if (done[i] || perm[i] == i) continue;
      rc = hipk_copy_cols(s->ctx, s->dt, s->m, PCOL(s, base, ldb, i), ldb, TCOL(s, 0), s->ld, 1);
      int j = i;
      while (!rc && perm[j] != i) {
         rc = hipk_copy_cols(s->ctx, s->dt, s->m, PCOL(s, base, ldb, perm[j]), ldb, PCOL(s, base, ldb, j), ldb, 1);
         done[j] = 1;
         j = perm[j];
      }
      if (!rc) rc = hipk_copy_cols(s->ctx, s->dt, s->m, TCOL(s, 0), s->ld, PCOL(s, base, ldb, j), ldb, 1);
      done[j] = 1;
   }
   free(done);
   return rc ? (rc < 0 ? rc : PRIMME_UNEXPECTED_FAILURE) : 0;
}

/* ================================ main iteration ================================ */
static int main_iter(pa_solver *s, double *evals, double *resNorms, int *ret, int *numRet) {
   primme_params *p = s->p;
   int i, blockSize = 0, availableBlockSize = 0, basisSize = 0, numLocked = 0, numGuesses = 0,
       nextGuess = 0, numConverged = 0, recentlyConverged = 0, maxRecentlyConverged = 0,
       restartLimitReached, nprevhVecs = 0, reset = 0, restartsSinceReset = 0, wholeSpace = 0, touch = 0,
       numConvergedStored = 0;
   const int maxNumRandoms = 10;
   int idleRestarts = 0, idleOuter = 0, stalled = 0, bsAtLastRestart = -1, ncAtLastRestart = -1, nlAtLastRestart = -1;
   PRIMME_INT mvAtLastRestart = -1, mvAtLastOuter = -1;
   double smallestResNorm = HUGE_VAL;
   int *flags = s->flags, *map = s->map, *iev = s->iev, *perm = s->perm;
   /* re-evaluated at every use: the dynamic method changes maxInnerIterations on the way */
#define gdNoPrecondLocking (p->locking && !p->correctionParams.precondition && \
                            p->correctionParams.maxInnerIterations == 0)
   pa_cost_model cost;
   double tstart = 0.0;

   *ret = PRIMME_MAIN_ITER_FAILURE;
   *numRet = 0;
   memset(&p->stats, 0, sizeof(p->stats));
   p->stats.estimateMinEVal = HUGE_VAL;
   p->stats.estimateMaxEVal = -HUGE_VAL;
   p->stats.estimateLargestSVal = -HUGE_VAL;
   p->stats.estimateBNorm = p->massMatrixMatvec ? -HUGE_VAL : 1.0;        /* (main_iter.c:387-388) */
   p->stats.estimateInvBNorm = p->massMatrixMatvec ? -HUGE_VAL : 1.0;
   for (i = 0; i < p->numEvals; i++) perm[i] = i;
   for (i = 0; i < p->maxBasisSize; i++) { map[i] = i; s->basisNorms[i] = 0.0; }
   s->targetShiftIndex = 0;

   if (p->numEvals == 0) { p->initSize = 0; *ret = 0; goto clean; }

   CHK(init_basis(s, &basisSize, &nextGuess, &numGuesses));
   CHK(pa_evecs_hat_init(s));
   p->initSize = 0;

   if (p->dynamicMethodSwitch > 0) {
      /* start with GD+k; JDQMR_ETol gets its turn once there are timings (eigs_dynamic.c) */
      pa_dyn_init(&cost, p);
      cost.t_mv = p->stats.timeMatvec / (double)p->stats.numMatvecs;
      p->dynamicMethodSwitch = (p->numEvals < 5 ||
            p->maxBasisSize + (p->locking ? p->numEvals : 0) >= p->n) ? 1 : 3;
      p->correctionParams.maxInnerIterations = 0;
   }

#define OUTER_LIMITS_OK() (p->stats.numMatvecs < p->maxMatvecs && \
      (p->maxOuterIterations == 0 || p->stats.numOuterIterations < p->maxOuterIterations))

   while (OUTER_LIMITS_OK()) {
      p->initSize = numConverged = numConvergedStored = numLocked;
      if (s->nBevecs > p->numOrthoConst) s->nBevecs = p->numOrthoConst;     /* B evecs of the locked columns: formed again on demand */
      reset = 0;
      for (i = 0; i < p->maxBasisSize; i++) flags[i] = UNCONV;
      s->targetShiftIndex = 0;
      if (s->Q) {
         int nQ = 0;
         CHK(pa_update_Q(s, p->targetShifts[s->targetShiftIndex], 0, basisSize, &nQ));
         if (nQ != basisSize) return PRIMME_UNEXPECTED_FAILURE;       /* "Not supported deficient QR" */
      }
      CHK(pa_update_projection(s, 0, basisSize));
      CHK(pa_update_QtV(s, 0, basisSize));
      CHK(pa_solve_H(s, basisSize, numLocked, numConverged));
      s->numArbitraryVecs = 0;
      maxRecentlyConverged = availableBlockSize = blockSize = 0;
      smallestResNorm = HUGE_VAL;
      p->stats.estimateResidualError = 0.0;
      if (!p->locking) p->stats.maxConvTol = 0.0;
      restartsSinceReset = 0;

      while (numConverged < p->numEvals && OUTER_LIMITS_OK() && !wholeSpace) {
         nprevhVecs = 0;
         int candidates_prepared = 0;

         /* ------------------ main block Davidson loop ------------------ */
         while (basisSize < p->maxBasisSize && OUTER_LIMITS_OK()) {
            /* same guard inside the loop: the basis already spans the space left (it can not reach a
             * maxBasisSize above it) and outer iterations go by without an operator application */
            if (p->stats.numMatvecs == mvAtLastOuter) {
               if (++idleOuter >= 8 && basisSize + numLocked + p->numOrthoConst >= p->n) { stalled = wholeSpace = 1; break; }
            } else idleOuter = 0;
            mvAtLastOuter = p->stats.numMatvecs;
            p->stats.numOuterIterations++;
            if (p->numTargetShifts > numConverged + 1 && s->Q) {
               /* a QR per target shift: one pair at a time (reference main_iter.c:527-530) */
               availableBlockSize = 1;
               maxRecentlyConverged = numConverged - numLocked + 1;
            } else {
               availableBlockSize = p->maxBlockSize;
               maxRecentlyConverged = PA_MAX(0, p->numEvals - numConverged);
            }
            availableBlockSize = PA_MIN(availableBlockSize, p->maxBasisSize - basisSize);
            availableBlockSize = PA_MIN(availableBlockSize, maxRecentlyConverged + 1);

            if (availableBlockSize > 0) {
               int practConvCheck = 0;
               if (p->n <= basisSize + numLocked + p->numOrthoConst) practConvCheck = 1;
               else if (gdNoPrecondLocking) practConvCheck = -1;
               CHK(pa_prepare_candidates(s, basisSize, VCOL(s, basisSize), WCOL(s, basisSize), 1, flags,
                     maxRecentlyConverged, s->blockNorms, blockSize, availableBlockSize, numLocked, evals,
                     resNorms, iev, &blockSize, &recentlyConverged, &smallestResNorm, numConverged,
                     s->basisNorms, &reset, nprevhVecs, practConvCheck, map));
               candidates_prepared = 1;
            } else {
               blockSize = recentlyConverged = 0;
            }
            numConverged += recentlyConverged;
            if (recentlyConverged > 0) touch = 0;   /* inner stopping criteria restart with every new pair */
            pa_monitor(s, s->hVals, basisSize, flags, iev, blockSize, s->basisNorms, numConverged, evals,
                  numLocked, s->lockedFlags, resNorms, primme_event_outer_iteration);

            if (p->dynamicMethodSwitch > 0) {
               if (cost.res0 == -1.0) cost.res0 = s->blockNorms[0];
               if (recentlyConverged > 0 || p->dynamicMethodSwitch == 2) {
                  cost.t_mv = p->stats.timeMatvec / (double)p->stats.numMatvecs;
                  if (pa_dyn_observe(&cost, p, tstart, recentlyConverged, 0, numConverged, s->blockNorms[0])) {
                     if (p->dynamicMethodSwitch == 3) CHK(pa_dyn_leave_gd(s, &cost));
                     else if (p->dynamicMethodSwitch == 2 || p->dynamicMethodSwitch == 4) CHK(pa_dyn_leave_jdqmr(s, &cost));
                  }
               }
            }

            if (numConverged >= p->numEvals ||
                  (p->locking && numConverged > numLocked && p->target != primme_smallest &&
                        p->target != primme_largest &&
                        (!s->Q || p->target == primme_closest_geq || p->target == primme_closest_leq)) ||
                  s->targetShiftIndex < 0 || (blockSize == 0 && recentlyConverged > 0) ||
                  (s->Q && fabs(p->targetShifts[s->targetShiftIndex] -
                                p->targetShifts[PA_MIN(p->numTargetShifts - 1, numConverged)]) >=
                                PA_MAX(p->aNorm, p->stats.estimateLargestSVal)) ||
                  (numConverged >= nextGuess - p->numOrthoConst && numGuesses > 0))
               break;

            if (blockSize > 0) {
               if (p->dynamicMethodSwitch > 0) { CHK(hipk_sync(s->ctx)); tstart = pa_wtime(); }
               CHK(solve_correction_gd(s, evals, numLocked, flags, basisSize, s->blockNorms, iev, blockSize, numConvergedStored, &touch));
               if (p->dynamicMethodSwitch > 0) { CHK(hipk_sync(s->ctx)); cost.t_inner += pa_wtime() - tstart; }
            }

            /* orthogonalise the corrections; when GD runs with locking and no
             * preconditioner keep Q'r for the practical-convergence test below */
            HS *Rlocked = NULL;
            const int ldRlocked = p->numOrthoConst + numLocked;
            const int blockSize0 = blockSize;
            if (gdNoPrecondLocking) {
               Rlocked = (HS *)calloc((size_t)(ldRlocked > 0 ? ldRlocked : 1) * (blockSize > 0 ? blockSize : 1), sizeof(HS));
               if (!Rlocked) return PRIMME_MALLOC_FAILURE;
            }
            for (i = 0; i < maxNumRandoms; i++) {
               int basisSizeOut = basisSize;
               int rc = ortho_block(s, s->V, s->ld, basisSize, basisSize + blockSize - 1, s->evecs, s->ldevecs,
                     p->numOrthoConst + numLocked, i == 0 ? Rlocked : NULL, ldRlocked, &basisSizeOut);
               if (rc) { free(Rlocked); return rc; }
               blockSize = basisSizeOut - basisSize;
               if (blockSize > 0 || availableBlockSize <= 0) break;
               rc = pa_random_col(s, VCOL(s, basisSize));
               if (rc) { free(Rlocked); return rc; }
               blockSize = 1;
            }
            if (i >= maxNumRandoms) {
               if (availableBlockSize > 0 && blockSize0 <= 0 && reset == 0) wholeSpace = 1;
               else reset = 2;
               blockSize = 0;
               free(Rlocked);
               break;
            }

            if (gdNoPrecondLocking) {
               /* practical convergence: sqrt(|r|^2 - |Q'r|^2) against the criterion
                * (reference main_iter.c:721-797) */
               if (numLocked > 0) {
                  for (i = 0; i < blockSize0 && numConverged < p->numEvals; i++) {
                     double nR = 0.0, normXx = 0.0;
                     for (int j = 0; j < ldRlocked; j++) nR += HS_ABS2(Rlocked[j + (size_t)i * ldRlocked]);
                     if (s->VtBV) {
                        /* |V_locked' x|^2 from the tracked Gram matrix (reference main_iter.c:747-758) */
                        for (int j = 0; j < numLocked; j++) {
                           HS t = 0.0;
                           for (int q = 0; q < basisSize; q++)
                              t += s->VtBV[j + (size_t)(numLocked + q) * s->ldVtBV] * s->hVecs[q + (size_t)iev[i] * basisSize];
                           normXx += HS_ABS2(t);
                        }
                     }
                     double newBlockNorm = sqrt(PA_MAX(s->blockNorms[i] * s->blockNorms[i] - nR * (1. + normXx), 0.0));
                     int rc = pa_check_convergence(s, VCOL(s, basisSize + i), s->ld, 1, NULL, 0, 0, numLocked, 0, 1,
                           &flags[iev[i]], &newBlockNorm, &s->hVals[iev[i]], &reset, -1);
                     if (rc) { free(Rlocked); return rc; }
                     s->basisNorms[iev[i]] = newBlockNorm;
                     if (flags[iev[i]] == CONV) {
                        flags[iev[i]] = PRACT_CONV;
                        numConverged++;
                        pa_monitor(s, s->hVals, basisSize, flags, &iev[i], 1, s->basisNorms, numConverged, NULL, 0,
                              NULL, NULL, primme_event_converged);
                     }
                  }
               }
               free(Rlocked);
               Rlocked = NULL;
               /* The reference breaks here only with RR, or with closest_geq / closest_leq
                * (main_iter.c:789-796): with harmonic / refined extraction and closest_abs a practically
                * converged pair beyond the wanted window would otherwise be flagged, not locked and
                * re-flagged without an operator application in between. */
               if (numConverged > numLocked && p->target != primme_smallest && p->target != primme_largest &&
                     (!s->Q || p->target == primme_closest_geq || p->target == primme_closest_leq))
                  break;
            }

#if !PA_IS_COMPLEX
            if (s->spec2_valid && s->spec2_k == basisSize && blockSize == 1) {
               /* W(:,k) = A v and V'W(:,k) were produced by the speculative tail (eigs_conv.c) */
               p->stats.numMatvecs += 1;
               for (i = 0; i <= basisSize; i++) s->H[i + (size_t)basisSize * s->K] = s->spec_hcol[i];
            } else
#endif
            {
               CHK(pa_matvec(s, s->V, s->ld, s->W, s->ld, basisSize, blockSize));
               CHK(pa_update_projection(s, basisSize, blockSize));
            }
            s->spec2_valid = 0;
            if (s->Q) {
               int nQ = basisSize;
               CHK(pa_update_Q(s, p->targetShifts[s->targetShiftIndex], basisSize, blockSize, &nQ));
               if (nQ != basisSize + blockSize) { blockSize = 0; reset = 1; break; }
               CHK(pa_update_QtV(s, basisSize, blockSize));
            }

            /* remember the coefficient vectors of this step (the +k directions) */
            for (int j = 0; j < basisSize; j++) {
               memcpy(s->prevhVecs + (size_t)j * s->K, s->hVecs + (size_t)j * basisSize, (size_t)basisSize * sizeof(HS));
               memset(s->prevhVecs + (size_t)j * s->K + basisSize, 0, (size_t)(s->K - basisSize) * sizeof(HS));
            }
            nprevhVecs = basisSize;
            basisSize += blockSize;
            blockSize = 0;
            CHK(pa_solve_H(s, basisSize, numLocked, numConverged));
            s->numArbitraryVecs = 0;
            candidates_prepared = 0;
            /* the smallest singular value of R bounds |theta_0 - tau| from above; a clear violation
             * means the factorisation has drifted: rebuild it (reference main_iter.c:858-884) */
            if (s->refined && basisSize > 0 && restartsSinceReset > 1 && s->targetShiftIndex >= 0 &&
                  fabs(p->targetShifts[s->targetShiftIndex] - s->hVals[0]) -
                              PA_MAX(p->aNorm, p->stats.estimateLargestSVal) * s->mach_eps > s->hSVals[0]) {
               reset = 2;
               break;
            }
         } /* main block Davidson loop */

         if (basisSize >= p->n - p->numOrthoConst - numLocked) {
            if (p->stats.maxConvTol < p->stats.estimateResidualError) reset = 1;
         }
         if (reset > 0) break;

         /* ---- make sure there are candidates (X, R) for after the restart ---- */
         if (!candidates_prepared) {
            if (blockSize > 0) {
               availableBlockSize = blockSize;
               maxRecentlyConverged = 0;
            } else if (p->numTargetShifts > numConverged + 1) {
               maxRecentlyConverged = p->locking
                     ? PA_MAX(PA_MIN(p->numEvals, numLocked + 1) - numConverged, 0)
                     : PA_MAX(PA_MIN(p->numEvals, numConverged + 1) - numConverged, 0);
               availableBlockSize = maxRecentlyConverged;
            } else {
               maxRecentlyConverged = PA_MAX(0, p->numEvals - numConverged);
               availableBlockSize = PA_MAX(0, PA_MIN(p->maxBlockSize, p->maxBasisSize - (numConverged - numLocked)));
               availableBlockSize = PA_MIN(availableBlockSize, maxRecentlyConverged + 1);
            }
            availableBlockSize = (int)PA_MAX(0, PA_MIN((int64_t)availableBlockSize, p->n - numLocked - p->numOrthoConst));

            if (availableBlockSize <= 0 ||
                  p->minRestartSize + p->restartingParams.maxPrevRetain + availableBlockSize < p->maxBasisSize ||
                  p->numOrthoConst + numLocked + basisSize >= p->n || s->Q) {
               double dummyZero = 0.0;
               double *srn = (p->target == primme_closest_abs || p->target == primme_largest_abs)
                                   ? &dummyZero : &smallestResNorm;
               s->plan_allowed = (numGuesses <= 0 && p->dynamicMethodSwitch <= 0);
               s->pl_launched = 0;
               {
                  int rcp = pa_prepare_candidates(s, basisSize, NULL, NULL, 0, flags, maxRecentlyConverged, s->blockNorms,
                        blockSize, availableBlockSize, numLocked, evals, resNorms, iev, &blockSize,
                        &recentlyConverged, srn, numConverged, s->basisNorms, &reset, nprevhVecs, 0, map);
                  s->plan_allowed = 0;
                  CHK(rcp);
               }

               if (s->Q && numConverged + recentlyConverged > numLocked && p->numTargetShifts > numLocked + 1)
                  blockSize = 0;   /* the next pair may belong to a different target shift */
               for (i = 0, numConverged = numLocked; i < basisSize; i++)
                  if (flags[i] != UNCONV && numConverged < p->numEvals &&
                        (i < p->numEvals - numLocked || p->target == primme_closest_geq ||
                              p->target == primme_closest_leq))
                     numConverged++;

               CHK(pa_block_first_reorder(s, basisSize, flags, iev, blockSize, numConverged, numLocked));
            } else {
               blockSize = availableBlockSize;
               for (i = 0; i < blockSize; i++) iev[i] = i;
               pa_map_vecs(s->prevhVecs, basisSize, nprevhVecs, s->K, s->hVecs, 0, basisSize, basisSize, map);
            }
         }
         if (reset > 0) break;

         pa_permute_cols(s->prevhVecs, basisSize, nprevhVecs, s->K, map);

         /* ------------------------------ restart ------------------------------ */
         CHK(pa_restart(s, basisSize, flags, iev, &blockSize, s->blockNorms, perm, evals, resNorms,
               &numConverged, &numLocked, s->lockedFlags, nprevhVecs, numGuesses, &basisSize,
               &restartsSinceReset));
         restartsSinceReset++;
         CHK(pa_evecs_hat_update(s, &numConvergedStored, numConverged));

         if (numGuesses > 0) {
            s->fov_valid = 0; s->spec2_valid = 0; s->fov_carry = 0;
            /* feed remaining initial guesses into the restarted basis */
            int numNew = PA_MAX(0, PA_MIN(p->minRestartSize + numConverged - (nextGuess - p->numOrthoConst), numGuesses));
            numNew = PA_MAX(0, PA_MIN(basisSize + numNew, p->maxBasisSize) - basisSize);
            numNew = (int)PA_MAX(0, PA_MIN((int64_t)(basisSize + numNew + p->numOrthoConst + numLocked), p->n) -
                                          p->numOrthoConst - numLocked - basisSize);
            CHK(hipk_copy_cols(s->ctx, s->dt, s->m, ECOL(s, nextGuess), s->ldevecs, VCOL(s, basisSize), s->ld, numNew));
            nextGuess += numNew;
            numGuesses -= numNew;
            int basisSizeOut = basisSize;
            CHK(ortho_block(s, s->V, s->ld, basisSize, basisSize + numNew - 1, s->evecs, s->ldevecs,
                  numLocked + p->numOrthoConst, NULL, 0, &basisSizeOut));
            numNew = basisSizeOut - basisSize;
            CHK(pa_matvec(s, s->V, s->ld, s->W, s->ld, basisSize, numNew));
            /* the block's X, R slots were overwritten: no candidates carried over */
            blockSize = 0;
            /* H must be dense again for the new columns: rebuild coefficient layout */
            {
               /* hVecs currently has leading dimension basisSize; H is K x K already */
            }
            if (s->Q) {
               int nQ = basisSize;
               CHK(pa_update_Q(s, p->targetShifts[s->targetShiftIndex], basisSize, numNew, &nQ));
               if (nQ != basisSize + numNew) return PRIMME_UNEXPECTED_FAILURE;
            }
            CHK(pa_update_projection(s, basisSize, numNew));
            CHK(pa_update_QtV(s, basisSize, numNew));
            basisSize += numNew;
            CHK(pa_solve_H(s, basisSize, numLocked, numConverged));
         }

         p->stats.numRestarts++;
         p->initSize = numConverged;
         /* Three restarts in a row without a single operator application: the basis cannot grow (it
          * spans what is left of the space) and the pairs in it do not meet the tolerance -- e.g. single
          * precision, a block that is a sizeable fraction of n.  The reference restarts the same
          * basis forever there (seen with n = 88, block 20, 3 constraints, float); this solver leaves
          * the loops with what it has and returns PRIMME_MAIN_ITER_FAILURE.  "Idle" = five restarts
          * in a row with the same basis size, the same converged / locked counts and no operator
          * application in between. */
         if (p->stats.numMatvecs == mvAtLastRestart && basisSize == bsAtLastRestart && numConverged == ncAtLastRestart &&
               numLocked == nlAtLastRestart) { if (++idleRestarts >= 5) stalled = wholeSpace = 1; }
         else idleRestarts = 0;
         mvAtLastRestart = p->stats.numMatvecs; bsAtLastRestart = basisSize; ncAtLastRestart = numConverged; nlAtLastRestart = numLocked;
#if !PA_IS_COMPLEX
         if (s->wtr_enabled && s->fuse_gd && !s->Q) CHK(pa_refresh_wtq(s, basisSize, p->numOrthoConst + numLocked));
#endif
         if (p->dynamicMethodSwitch == 1) {
            /* few eigenpairs: GD+k is judged after each restart, restart cost included */
            CHK(hipk_sync(s->ctx));
            tstart = pa_wtime();
            cost.t_mv = p->stats.timeMatvec / (double)p->stats.numMatvecs;
            pa_dyn_observe(&cost, p, tstart, 0, 1, numConverged, s->blockNorms[0]);
            CHK(pa_dyn_leave_gd(s, &cost));
         }
         for (i = 0; i < p->maxBasisSize; i++) map[i] = i;
      } /* restarting loop */

      if (reset > 0) {
         /* V and W accumulated too much error: re-orthogonalise and recompute W
          * (reference: `continue` with reset set re-enters the outer loop) */
         continue;
      }

      if (p->locking) {
         CHK(copy_back_candidates(s, basisSize, evals, resNorms, numConverged, numRet));
         if (*numRet < numConverged) *numRet = numConverged;
         pa_dyn_recommend(&cost, p);
         p->stats.lockingIssue = 0;
         *ret = (numConverged == p->numEvals || (wholeSpace && !stalled)) ? 0 : PRIMME_MAIN_ITER_FAILURE;
         goto clean;
      } else {
         restartLimitReached = OUTER_LIMITS_OK() ? 0 : 1;
         CHK(verify_norms(s, restartLimitReached ? PA_MIN(p->numEvals, basisSize) : numConverged, resNorms, flags, &numConverged));
         if (restartLimitReached || numConverged >= p->numEvals || wholeSpace) {
            pa_dyn_recommend(&cost, p);
            for (i = 0; i < p->numEvals; i++) { evals[i] = s->hVals[i]; perm[i] = i; }
            CHK(hipk_copy_cols(s->ctx, s->dt, s->m, s->V, s->ld, ECOL(s, p->numOrthoConst), s->ldevecs, p->numEvals));
            *numRet = p->numEvals;
            p->initSize = numConverged;
            *ret = (numConverged >= p->numEvals) ? 0 : PRIMME_MAIN_ITER_FAILURE;
            goto clean;
         }
         /* some pairs lost convergence in the restart: re-orthogonalise, recompute W */
         CHK(ortho_block(s, s->V, s->ld, 0, basisSize - 1, s->evecs, s->ldevecs, p->numOrthoConst, NULL, 0, &basisSize));
         CHK(pa_matvec(s, s->V, s->ld, s->W, s->ld, 0, basisSize));
         restartsSinceReset = 0;
         reset = 0;
         p->stats.estimateResidualError = 0.0;
         numConverged = 0;
      }
   }
   /* limits reached before entering / while re-entering the outer loop */
   if (p->locking) {
      CHK(copy_back_candidates(s, basisSize, evals, resNorms, numConverged, numRet));
      if (*numRet < numConverged) *numRet = numConverged;
   }

clean:
   if (p->aNorm <= 0.0) p->aNorm = p->massMatrixMatvec ? p->stats.estimateLargestSVal / p->stats.estimateInvBNorm : p->stats.estimateLargestSVal;      /* (main_iter.c:1344-1347) */
   /* locked vectors are stored in convergence order: sort them like evals */
   CHK(permute_dev_cols(s, ECOL(s, p->numOrthoConst), s->ldevecs, p->initSize, perm));
   CHK(hipk_sync(s->ctx));
   return 0;
}

/* ================================ front end ===================================== */
static void free_solver(pa_solver *s) {
   if (!s) return;
   if (s->ctx) {
      hipk_sync(s->ctx);
      hipk_free(s->ctx, s->V); hipk_free(s->ctx, s->W); hipk_free(s->ctx, s->T); hipk_free(s->ctx, s->Jw);
      hipk_free(s->ctx, s->V2); hipk_free(s->ctx, s->W2); hipk_free(s->ctx, s->d_coef2);      /* (d_theta2 / h_theta2 live inside) */
      hipk_host_free(s->ctx, s->h_coef2);
      hipk_free(s->ctx, s->evecsHat); hipk_free(s->ctx, s->Q); hipk_free(s->ctx, s->BT); hipk_free(s->ctx, s->Bevecs);
      hipk_free(s->ctx, s->d_red); hipk_free(s->ctx, s->d_coef); hipk_free(s->ctx, s->d_theta);
      hipk_host_free(s->ctx, s->h_red); hipk_host_free(s->ctx, s->h_coef); hipk_host_free(s->ctx, s->h_theta);
      hipk_ctx_destroy(s->ctx);
   }
   free(s->H); free(s->hVecs); free(s->prevhVecs); free(s->hVals); free(s->prevRitzVals);
   free(s->Mq); free(s->Mlu); free(s->Mpiv); free(s->R); free(s->QtV); free(s->hU); free(s->hSVals); free(s->hVecsRot);
   free(s->VtBV); free(s->fVtBV); free(s->blockNorms); free(s->basisNorms); free(s->spec_hcol); free(s->wtq); free(s->rst_y); free(s->rst_ov); free(s->rst_c); free(s->rst_grow);
   free(s->flags); free(s->map); free(s->iev); free(s->perm); free(s->lockedFlags);
   free(s);
}

extern void primme_amd_global_sum(void *, void *, int *, struct primme_params *, int *);

int pa_svds_conv_test_is_vector_free(const primme_params *p);

static int solve(void *evals_out, void *evecs, void *resNorms_out, primme_params *p, hipk_dtype dt, int out_double) {
   const double t0 = pa_wtime();
   if (!p) return -4;
   const int work_is_float = (dt == HIPK_F32 || dt == HIPK_C32);
   const int real_out_is_float = work_is_float && !out_double;
   const double mach_eps = work_is_float ? 1.1920928955078125e-07 : PA_EPS;

   if (p->numProcs <= 1 && evals_out && evecs && resNorms_out) { p->nLocal = p->n; p->procID = 0; }
   primme_set_defaults(p);
   if (p->orth == primme_orth_default)
      p->orth = (work_is_float || p->maxBlockSize > 1) ? primme_orth_explicit_I : primme_orth_implicit_I;
   /* generalised problems run on the tracked-Gram path whatever was asked for (the reference would take Bortho_gen with
    * orth = implicit_I; same subspaces, another orthonormalisation of them) */
   if (p->massMatrixMatvec) p->orth = primme_orth_explicit_I;
   if (p->ldOPs == -1) p->ldOPs = p->nLocal;
   if (!evals_out && !evecs && !resNorms_out) return 0;
   if (p->iseed[0] < 0 || p->iseed[0] > 4095) p->iseed[0] = p->procID % 4096;
   if (p->iseed[1] < 0 || p->iseed[1] > 4095) p->iseed[1] = (int)(p->procID / 4096 + 1) % 4096;
   if (p->iseed[2] < 0 || p->iseed[2] > 4095) p->iseed[2] = (int)((p->procID / 4096) / 4096 + 2) % 4096;
   if (p->iseed[3] < 0 || p->iseed[3] > 4095) p->iseed[3] = (2 * (int)(((p->procID / 4096) / 4096) / 4096) + 1) % 4096;
   /* callbacks without a declared operand type get the entry point's precision, like the reference
    * (primme_c.c:170-183); the library's own callbacks work on doubles.  eigs_callbacks.c converts. */
   {
      const primme_op_datatype scalar_t = work_is_float ? primme_op_float : primme_op_double;
      if (p->matrixMatvec && p->matrixMatvec_type == primme_op_default) p->matrixMatvec_type = scalar_t;
      if (p->applyPreconditioner && p->applyPreconditioner_type == primme_op_default) p->applyPreconditioner_type = scalar_t;
      if (p->globalSumReal && p->globalSumReal_type == primme_op_default)
         p->globalSumReal_type = (p->globalSumReal == primme_amd_global_sum) ? primme_op_double : scalar_t;
      if (p->broadcastReal && p->broadcastReal_type == primme_op_default) p->broadcastReal_type = scalar_t;
      if (p->convTestFun && p->convTestFun_type == primme_op_default) p->convTestFun_type = scalar_t;
      if (p->monitorFun && p->monitorFun_type == primme_op_default) p->monitorFun_type = scalar_t;
   }
   if (!p->convTestFun) {
      p->convTestFun = pa_conv_test_absolute;
      p->convTestFun_type = (dt == HIPK_F32 || dt == HIPK_C32) ? primme_op_float : primme_op_double;
      if (p->eps == 0.0) p->eps = mach_eps * 1e4;
   }
   int rc = pa_check_input(evals_out, evecs, resNorms_out, p, mach_eps);
   if (rc) return rc;

   /* what this build of the path covers; anything else must fail loudly */
#if PA_IS_COMPLEX
   if (dt != HIPK_C64 && dt != HIPK_C32) return PRIMME_FUNCTION_UNAVAILABLE;
   /* the complex objects carry the three extractions with the Generalized-Davidson family, the JDQMR inner solver and the
    * dynamic switch between the two (eigs_dynamic.c holds timings and ratios only, nothing scalar-typed) */
#else
   if (dt != HIPK_F64 && dt != HIPK_F32) return PRIMME_FUNCTION_UNAVAILABLE;
#endif
   const int refined = (p->projectionParams.projection == primme_proj_refined);
   const int harmonic = (p->projectionParams.projection == primme_proj_harmonic) || refined;
   /* harmonic extraction with an extremal target: the reference's own solve_H_Harm has no case for it (solve_projection.c:469-482,
    * `default: assert(0)`), so there is nothing to be at parity with.  Refined extraction with an extremal target is defined
    * as soon as the caller gives the shift of the factorisation (targetShifts[0]; without one the reference dereferences a
    * NULL pointer, main_iter.c:465): round 6 lets it through */
   const int extremal = (p->target == primme_smallest || p->target == primme_largest || p->target == primme_largest_abs);
   /* mass matrix (round 6): real and complex Hermitian panels, Rayleigh-Ritz (check_input: -39 otherwise, like the reference), the
    * Generalized-Davidson family, the JDQMR inner solver (projectors on B Q and B x, eigs_jd.c) and the dynamic switch between them.
    * PRIMME_AMD_MASS_NO_DYNAMIC=1 keeps a PRIMME_DYNAMIC run in its GD+k mode (what the reference's own switch does when it
    * leaves JDQMR: maxInnerIterations = 0, state -2 on return, main_iter.c:2330-2345, :1221-1228): an A/B knob. */
   if (p->massMatrixMatvec && p->dynamicMethodSwitch > 0 && getenv("PRIMME_AMD_MASS_NO_DYNAMIC") != NULL) {
      p->dynamicMethodSwitch = -2;
      p->correctionParams.maxInnerIterations = 0;
   }
   if (harmonic && extremal && !(refined && p->numTargetShifts > 0 && p->targetShifts)) {
      if (p->printLevel > 0 && p->outputFile)
         fprintf(p->outputFile, "primme_amd: requested configuration (harmonic projection with an extremal target) is not on the device path\n");
      return PRIMME_FUNCTION_UNAVAILABLE;
   }

   /* the basis never grows beyond the space itself, whatever maxBasisSize says */
   /* (real panels: the wide-basis restart kernel takes 1 023 columns since round 6, hipk_panels.hip:ritz_big_kernel; the complex
    * update stages its coefficient block in slices of 64 basis columns and its outputs in groups of 16, hipk_complex.hip:zritz_t,
    * so it takes the same width; the reference has no limit, primme_c.c:470-487) */
   const int max_basis = 1023;
   if (PA_MIN((int64_t)p->maxBasisSize, p->n - p->numOrthoConst) > max_basis) {
      if (p->printLevel > 0 && p->outputFile)
         fprintf(p->outputFile, "primme_amd: maxBasisSize > %d is not on the device path\n", max_basis);
      return PRIMME_FUNCTION_UNAVAILABLE;
   }

   pa_solver *s = (pa_solver *)calloc(1, sizeof(pa_solver));
   if (!s) return PRIMME_MALLOC_FAILURE;
   s->p = p; s->dt = dt; s->es = (dt == HIPK_F64) ? 8 : (dt == HIPK_F32) ? 4 : (dt == HIPK_C64) ? 16 : 8; s->mach_eps = mach_eps;
   s->m = p->nLocal; s->ld = p->ldOPs; s->K = p->maxBasisSize;
   /* Columns of V, W and the scratch panels on 128-byte boundaries when nobody outside the library sees their leading
    * dimension (its own operator and preconditioner, ldOPs left at nLocal): a wave's 1 KB of a column is then 8 cache
    * lines instead of 9 (the fused residual kernel: 132 -> 121 us at k = 15, m = 2 000 250, whose columns are only
    * 16-byte aligned).  PRIMME_AMD_NO_LDPAD=1 keeps ldOPs. */
   if (p->matrixMatvec == primme_amd_matvec && (!p->applyPreconditioner || p->applyPreconditioner == primme_amd_jacobi_precond) &&
         p->ldOPs == p->nLocal && p->nLocal > 256 && getenv("PRIMME_AMD_NO_LDPAD") == NULL) {
      const int64_t q = (int64_t)(128 / s->es);
      s->ld = (p->nLocal + q - 1) / q * q;
   }
   s->evecs = (char *)evecs; s->ldevecs = p->ldevecs;
   s->startTime = t0;
   /* the dynamic method's cost model needs device time, not launch time */
   s->phase_timing = p->profile != NULL || p->dynamicMethodSwitch > 0;
   /* PRIMME_AMD_FORCE_COMM: run the cross-rank reduction path with a one-rank communicator, which is
    * how the RCCL calls are exercised on a single-GPU box (tests/test_comm_gpu.py) */
   s->spec2_enabled = getenv("PRIMME_AMD_NO_SPEC2") == NULL;
   /* projection column from W'r of the fused residual pass: default; PRIMME_AMD_NO_WTR
    * switches back to the separate pass over V (measurement knob, read once per solve) */
   s->wtr_enabled = getenv("PRIMME_AMD_NO_WTR") == NULL;
   s->device_rr = getenv("PRIMME_AMD_DEVICE_RR") != NULL;
   s->parallel = ((p->numProcs > 1 || getenv("PRIMME_AMD_FORCE_COMM")) && p->globalSumReal != NULL);
   s->dev_comm = (s->parallel && p->globalSumReal == primme_amd_global_sum);
   s->coef_valid_k = -1;
   s->fuse_gd = !PA_IS_COMPLEX && (p->correctionParams.maxInnerIterations == 0 && p->dynamicMethodSwitch <= 0 && !p->correctionParams.precondition &&
                 !p->correctionParams.projectors.RightX &&
                 (p->convTestFun == pa_conv_test_absolute || pa_svds_conv_test_is_vector_free(p)));
   s->maxRank = p->numOrthoConst + (p->locking ? p->numEvals : 0) + p->maxBasisSize;
   s->B = p->massMatrixMatvec != NULL;
   if (s->B) s->fuse_gd = 0;
   s->ref_soft_alias = s->B && !p->locking && getenv("PRIMME_AMD_JDQMR_REF_SOFT_LOCKING") != NULL;
   if (hipk_ctx_create(&s->ctx, p->queue)) { free(s); return PRIMME_UNEXPECTED_FAILURE; }
   /* peer-to-peer transport: the second stage of a reduction may exchange with the other ranks itself */
   if (s->dev_comm) (void)pa_comm_attach_ctx(p->commInfo, s->ctx);
   /* callbacks find the solver's stream in primme->queue (reference: the queue/handle
    * field carries the device queue, examples/ex_eigs_dhipblas.c:177-179) */
   void *user_queue = p->queue;
   void *own_stream = hipk_ctx_stream(s->ctx);
   if (!p->queue) p->queue = &own_stream;

   const int K = s->K, nev = p->numEvals, b = p->maxBlockSize;
   s->nT = K + 2 * b + 2;
   /* widest block ever orthonormalised at once: the constraints (init_basis) or the whole basis */
   const int Kc = PA_MAX(K, p->numOrthoConst);
   s->coef_cap = (size_t)Kc * Kc;
   s->red_cap = PA_MAX(PA_MAX((s->maxRank + 16) * (b + 8) + 64, K * K + 64), PA_MAX(s->maxRank * K, Kc * Kc) + 64);
   const size_t colBytes = (size_t)(s->ld > 0 ? s->ld : 1) * s->es;
   /* K^-1-weighted right projector (reference main_iter.c:324-333) */
   const int maxEvecs = p->numOrthoConst + nev;
   const int need_hat = p->correctionParams.precondition && p->correctionParams.maxInnerIterations != 0 &&
                        p->correctionParams.projectors.RightQ && p->correctionParams.projectors.SkewQ;
   if (need_hat) {
      s->ldM = maxEvecs;
      s->Mq = (HS *)calloc((size_t)maxEvecs * maxEvecs + 1, sizeof(HS));
      s->Mlu = (HS *)calloc((size_t)maxEvecs * maxEvecs + 1, sizeof(HS));
      s->Mpiv = (int *)calloc((size_t)maxEvecs + 1, sizeof(int));
      if (!s->Mq || !s->Mlu || !s->Mpiv) { free_solver(s); p->queue = user_queue; return PRIMME_MALLOC_FAILURE; }
   }
   rc = hipk_malloc(s->ctx, colBytes * K, (void **)&s->V) || hipk_malloc(s->ctx, colBytes * K, (void **)&s->W) ||
        hipk_malloc(s->ctx, colBytes * s->nT, (void **)&s->T) ||
        ((p->correctionParams.maxInnerIterations != 0 || p->dynamicMethodSwitch > 0) && hipk_malloc(s->ctx, colBytes * (s->B ? 8 : 6) * b, (void **)&s->Jw)) ||
        (s->B && (p->correctionParams.maxInnerIterations != 0 || p->dynamicMethodSwitch > 0) &&
         hipk_malloc(s->ctx, (size_t)(s->ldevecs > 0 ? s->ldevecs : 1) * s->es * (maxEvecs + 1), (void **)&s->Bevecs)) ||
        hipk_malloc(s->ctx, ((size_t)s->red_cap * 3 + 64) * 8, (void **)&s->d_red) ||
        (harmonic && hipk_malloc(s->ctx, colBytes * K, (void **)&s->Q)) ||
        (s->B && hipk_malloc(s->ctx, colBytes * (s->nBT = PA_MAX(3 * (K + b), PA_MAX(p->numOrthoConst, 3 * (b + nev)))), (void **)&s->BT)) ||
        (need_hat && hipk_malloc(s->ctx, (size_t)(s->ldevecs > 0 ? s->ldevecs : 1) * s->es * maxEvecs, (void **)&s->evecsHat)) ||
        hipk_malloc(s->ctx, s->coef_cap * sizeof(HS), (void **)&s->d_coef) || hipk_malloc(s->ctx, (size_t)K * 8, (void **)&s->d_theta) ||
        hipk_host_alloc(s->ctx, ((size_t)s->red_cap * 3 + 64) * 8, (void **)&s->h_red) ||
        hipk_host_alloc(s->ctx, s->coef_cap * sizeof(HS), (void **)&s->h_coef) || hipk_host_alloc(s->ctx, (size_t)K * 8, (void **)&s->h_theta);
   s->H = (HS *)calloc((size_t)K * K, sizeof(HS)); s->hVecs = (HS *)calloc((size_t)K * K, sizeof(HS));
   s->prevhVecs = (HS *)calloc((size_t)K * K, sizeof(HS)); s->hVals = (double *)calloc((size_t)K, 8);
   s->prevRitzVals = (double *)calloc((size_t)K + nev, 8);
   s->blockNorms = (double *)calloc((size_t)K + b, 8); s->basisNorms = (double *)calloc((size_t)K, 8);
   s->spec_hcol = (double *)calloc((size_t)K + 2, 8);
   s->wtq = (double *)calloc((size_t)K * HIPK_WTR_MAX_K + 1, 8);
   s->rst_y = (double *)calloc((size_t)K + 1, 8);
   s->rst_ov = (double *)calloc(4 * (size_t)HIPK_WTR_MAX_K + 4, 8);
   s->rst_c = (double *)calloc(4 * (size_t)HIPK_WTR_MAX_K + 4, 8);
   s->rst_grow = (double *)calloc((size_t)HIPK_WTR_MAX_K + 1, 8);
   s->fused_restart = getenv("PRIMME_AMD_NO_FUSED_RESTART") == NULL;
   s->wtq_rows = -1;
   if (harmonic) {
      s->R = (HS *)calloc((size_t)K * K + 1, sizeof(HS));
      s->hU = (HS *)calloc((size_t)K * K + 1, sizeof(HS));
      if (refined) {
         s->refined = 1;
         s->hSVals = (double *)calloc((size_t)K + 1, 8); s->hVecsRot = (HS *)calloc((size_t)K * K + 1, sizeof(HS));
      } else s->QtV = (HS *)calloc((size_t)K * K + 1, sizeof(HS));
      if (!s->R || !s->hU || (refined ? (!s->hSVals || !s->hVecsRot) : !s->QtV)) { free_solver(s); p->queue = user_queue; return PRIMME_MALLOC_FAILURE; }
   }
   s->flags = (int *)calloc((size_t)K, sizeof(int)); s->map = (int *)calloc((size_t)K, sizeof(int));
   s->iev = (int *)calloc((size_t)K + b, sizeof(int)); s->perm = (int *)calloc((size_t)nev + 1, sizeof(int));
   s->lockedFlags = (int *)calloc((size_t)nev + 1, sizeof(int));
   if (p->orth == primme_orth_explicit_I) {
      s->ldVtBV = s->maxRank;
      s->VtBV = (HS *)calloc((size_t)s->maxRank * s->maxRank, sizeof(HS));
      s->fVtBV = (HS *)calloc((size_t)s->maxRank * s->maxRank, sizeof(HS));
      if (!s->VtBV || !s->fVtBV) rc = 1;
   }
   if (!rc) rc = hipk_ctx_set_mirror(s->ctx, s->d_red, s->h_red, (size_t)s->red_cap * 3 + 64);
   if (!rc) {
      /* [reductions | overlaps of the fused pass | the other overlap buffer (pre-enqueued iteration) | its Ritz pair] */
      s->d_fov = s->d_red + s->red_cap; s->h_fov = s->h_red + s->red_cap;
      s->d_fov_alt = s->d_red + 2 * (size_t)s->red_cap; s->h_fov_alt = s->h_red + 2 * (size_t)s->red_cap;
      s->d_hnext = s->d_red + 3 * (size_t)s->red_cap; s->h_hnext = s->h_red + 3 * (size_t)s->red_cap;
      rc = hipk_memset0(s->ctx, s->d_red, ((size_t)s->red_cap * 3 + 64) * 8);
   }
   s->pre_enabled = getenv("PRIMME_AMD_NO_PRELAUNCH") == NULL;
   s->pre_quiet_fin = getenv("PRIMME_AMD_LOUD_FIN") == NULL;
   if (!rc && s->fused_restart && s->fuse_gd && b == 1 && !harmonic && K <= 32 && s->nT >= 4 &&
         s->red_cap >= 64 + 2 * HIPK_WTR_MAX_K && getenv("PRIMME_AMD_NO_SPEC_RESTART") == NULL) {
      /* alternate panels of the speculative restart (eigs_solver.h); without them the restart runs in place */
      if (hipk_malloc(s->ctx, colBytes * K, (void **)&s->V2) || hipk_malloc(s->ctx, colBytes * K, (void **)&s->W2) ||
            /* coefficient block and Ritz values of the planned restart in ONE buffer (values behind the K x K block), on the device
             * and in the pinned mirror: they travel in one copy launch instead of two (eigs_conv.c:try_speculative_restart) */
            hipk_malloc(s->ctx, (size_t)(K * K + K) * 8, (void **)&s->d_coef2) ||
            hipk_host_alloc(s->ctx, (size_t)(K * K + K) * 8, (void **)&s->h_coef2)) {
         hipk_free(s->ctx, s->V2); hipk_free(s->ctx, s->W2); hipk_free(s->ctx, s->d_coef2);
         hipk_host_free(s->ctx, s->h_coef2);
         s->V2 = s->W2 = NULL; s->d_coef2 = s->h_coef2 = NULL;
      }
      s->d_theta2 = s->d_coef2 ? s->d_coef2 + (size_t)K * K : NULL;
      s->h_theta2 = s->h_coef2 ? s->h_coef2 + (size_t)K * K : NULL;
   }
   if (rc || !s->H || !s->hVecs || !s->prevhVecs || !s->hVals || !s->prevRitzVals || !s->blockNorms ||
         !s->basisNorms || !s->flags || !s->map || !s->iev || !s->perm || !s->lockedFlags) {
      free_solver(s);
      p->queue = user_queue;
      return PRIMME_MALLOC_FAILURE;
   }

   /* evals / resNorms are kept in double inside and narrowed on exit */
   double *evals = (double *)calloc((size_t)nev + 1, 8), *resNorms = (double *)calloc((size_t)nev + 1, 8);
   int ret = 0, numRet = 0;
   rc = main_iter(s, evals, resNorms, &ret, &numRet);
   if (!rc) {
      for (int i = 0; i < numRet; i++) {
         if (real_out_is_float) { ((float *)evals_out)[i] = (float)evals[i]; ((float *)resNorms_out)[i] = (float)resNorms[i]; }
         else { ((double *)evals_out)[i] = evals[i]; ((double *)resNorms_out)[i] = resNorms[i]; }
      }
      rc = ret;
   }
   free(evals);
   free(resNorms);
   pa_last_pre[0] = s->pre_launched; pa_last_pre[1] = s->pre_adopted;
   if (getenv("PRIMME_AMD_PRELAUNCH_STATS"))
      fprintf(stderr, "primme_amd: iterations enqueued ahead of the host: %ld launched, %ld adopted (of %lld outer iterations)\n", s->pre_launched,
            s->pre_adopted, (long long)p->stats.numOuterIterations);
   if (p->convTestFun == pa_conv_test_absolute) { /* leave the struct as the reference does: default installed */ }
   free_solver(s);
   p->queue = user_queue;
   p->stats.elapsedTime = pa_wtime() - t0;
   return rc;
}

#if !PA_IS_COMPLEX
int hip_dprimme(double *evals, double *evecs, double *resNorms, primme_params *primme) {
   return solve(evals, evecs, resNorms, primme, HIPK_F64, 0);
}
int hip_sprimme(float *evals, float *evecs, float *resNorms, primme_params *primme) {
   return solve(evals, evecs, resNorms, primme, HIPK_F32, 0);
}
/* hip_zprimme / hip_cprimme: eigs_complex.c */
#endif

/* internal entry for the svds front end: eigenvalues and residual norms always in double */
int pa_eigs_solve(void *evals_out, void *evecs, void *resNorms_out, primme_params *p, hipk_dtype dt, int out_double) {
   return solve(evals_out, evecs, resNorms_out, p, dt, out_double);
}
