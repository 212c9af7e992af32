/* eigs_restart.c — thick restart with +k retained directions, soft and hard locking.
 *
 *   pa_restart               <- reference src/eigs/restart.c:200-494  (restart_Sprimme)
 *   restart_soft_locking     <- reference :598-722
 *   restart_locking          <- reference :832-1187
 *   restart_RR               <- reference :1614-1735
 *   ortho_coefficient_vectors<- reference :2347-2408
 *   insertion_sort           <- reference src/eigs/auxiliary_eigs_normal.c:543-638
 *
 * The host part is integer/permutation logic on k-vectors, restated literally.
 * The n-length work of a restart is ONE fused pass over V and W (pa_ritz_update):
 * V <- V*h, W <- W*h in place, the next block's X and R, the vectors to lock
 * into evecs and their residual norms.
 */
#include "eigs_solver.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <assert.h>

double pa_problem_norm(int overrideUser, const primme_params *p);
int pa_ritz_update(pa_solver *s, int basisSize, const hipk_job *jobs, int njobs, double *norms_out,
      int nslots, int64_t flop_cols);
int pa_push_coefficients(pa_solver *s, int basisSize, int ldh);
int pa_check_convergence(pa_solver *s, char *X, int64_t ldX, int givenX, char *R, int64_t ldR,
      int givenR, int numLocked, int left, int right, int *flags, double *blockNorms,
      const double *hVals, int *reset, int practConvCheck);
int pa_ortho_local_vec(HS *x, int n, const HS *Q, int ldQ, int nQ, const HS *G,
      int ldG, double *R, int64_t iseed[4]);
int pa_solve_H_RR(pa_solver *s, const HS *H, int ldH, const HS *VtBV, int ldVtBV,
      HS *hVecs, int ldhVecs, double *hVals, int n, int numConverged);
void pa_monitor(pa_solver *s, double *basisEvals, int basisSize, int *basisFlags, int *iblock,
      int blockSize, double *basisNorms, int numConverged, double *lockedEvals, int numLocked,
      int *lockedFlags, double *lockedNorms, primme_event event);

int pa_reduce(pa_solver *s, double *d_buf, int count, int keep_dev, int defer_sync);
int pa_speculative_tail(pa_solver *s, int basisSize, int nLk, const char *rsrc, char *dstc, int nfov, int wtr,
      int speculate2, int parallel_host, int col, int adopted);
int pa_restart_harmonic(pa_solver *s, int ldh, int restartSize, int basisSize, int numConverged);
int pa_restart_refined(pa_solver *s, int ldh, int restartSize, int basisSize, int numConverged, int numPrevRetained,
      int indexOfPreviousVecs, int indexOfPreviousVecsBeforeRestart, const int *restartPerm, const int *hVecsPerm,
      int *numArbitraryVecs);
int pa_solve_H(pa_solver *s, int basisSize, int numLocked, int numConverged);
int pa_update_cholesky(const HS *G, int ldG, HS *fG, int ldfG, int n0, int n);

/* explicit_I: after V <- V*h, W <- W*h recompute from the data the Gram block G = V'V and
 * H = V'W of the restarted basis (reference auxiliary_eigs_normal.c:296-309 does it inside the
 * fused update) and rotate the locked-vs-basis block of VtBV on the host
 * (reference restart.c:1276-1290). */
static int refresh_gram_after_restart(pa_solver *s, int evecsSize, int nVold, int rs, int ldh) {
   if (!s->VtBV || rs <= 0) return 0;
   const int ldG = s->ldVtBV;
   HS *work = (HS *)malloc((size_t)(evecsSize > 0 ? evecsSize : 1) * rs * sizeof(HS));
   if (!work) return PRIMME_MALLOC_FAILURE;
   for (int c = 0; c < rs; c++)
      for (int i = 0; i < evecsSize; i++) {
         HS t = 0.0;
         for (int q = 0; q < nVold; q++) t += s->VtBV[i + (size_t)(evecsSize + q) * ldG] * s->hVecs[q + (size_t)c * ldh];
         work[i + (size_t)c * evecsSize] = t;
      }
   for (int c = 0; c < rs; c++)
      for (int i = 0; i < evecsSize; i++) s->VtBV[i + (size_t)(evecsSize + c) * ldG] = work[i + (size_t)c * evecsSize];
   free(work);
   hipk_seg seg = {s->V, s->ld, rs};
   const char *Vm = s->V;                          /* the metric side: B V for a generalised problem */
   if (s->B) {
      if (rs > s->nBT) return PRIMME_FUNCTION_UNAVAILABLE;
      CHK(pa_apply_B(s, s->V, s->ld, s->BT, s->ld, rs));
      Vm = s->BT;
   }
   CHK(hipk_panel_dots(s->ctx, s->dt, s->m, &seg, 1, Vm, s->ld, rs, s->d_red, rs));
   CHK(pa_reduce(s, s->d_red, SD * rs * rs, 0, 0));
   for (int c = 0; c < rs; c++)
      for (int i = 0; i < rs; i++) s->VtBV[(evecsSize + i) + (size_t)(evecsSize + c) * ldG] = ((const HS *)s->h_red)[i + (size_t)c * rs];
   CHK(hipk_panel_dots(s->ctx, s->dt, s->m, &seg, 1, s->W, s->ld, rs, s->d_red, rs));
   CHK(pa_reduce(s, s->d_red, SD * rs * rs, 0, 0));
   for (int c = 0; c < rs; c++)
      for (int i = 0; i < rs; i++) s->H[i + (size_t)c * s->K] = ((const HS *)s->h_red)[i + (size_t)c * rs];
   return 0;
}

/* Lock newVal by insertion into the sorted evals (smallest/largest) or in
 * convergence order within equal shifts (interior). perm remembers arrival order. */
static int insertion_sort(double newVal, double *evals, double newNorm, double *resNorms,
      int newFlag, int *flags, int *perm, int n, int initialShift, const primme_params *p) {
   int i;
   if (p->target == primme_smallest) {
      for (i = n; i > 0; i--) if (newVal >= evals[i - 1]) break;
   } else if (p->target == primme_largest) {
      for (i = n; i > 0; i--) if (newVal <= evals[i - 1]) break;
   } else {
      const double cur = p->targetShifts[PA_MIN(p->numTargetShifts - 1, initialShift + n)];
      for (i = n; i > 0; i--) {
         const double ith = p->targetShifts[PA_MIN(p->numTargetShifts - 1, initialShift + i - 1)];
         int stop;
         if (p->target == primme_closest_geq) stop = (newVal - cur >= evals[i - 1] - cur);
         else if (p->target == primme_closest_leq) stop = (cur - newVal >= cur - evals[i - 1]);
         else if (p->target == primme_closest_abs) stop = (fabs(newVal - cur) >= fabs(evals[i - 1] - cur));
         else if (p->target == primme_largest_abs) stop = (fabs(newVal - cur) <= fabs(evals[i - 1] - cur));
         else return PRIMME_FUNCTION_UNAVAILABLE;
         if (ith != cur || stop) break;
      }
   }
   for (int c = n - 1; c >= i; c--) {
      evals[c + 1] = evals[c];
      if (resNorms) resNorms[c + 1] = resNorms[c];
      if (perm) perm[c + 1] = perm[c];
      if (flags) flags[c + 1] = flags[c];
   }
   evals[i] = newVal;
   if (resNorms) resNorms[i] = newNorm;
   if (perm) perm[i] = n;
   if (flags) flags[i] = newFlag;
   return 0;
}

/* Orthogonalise up to *numPrevRetained previous coefficient vectors against the
 * first indexOfPreviousVecs (+ already retained) current ones and append them. */
static int ortho_coefficient_vectors(pa_solver *s, int basisSize, int ldh, int indexOfPreviousVecs,
      const HS *G, int ldG, int nprevhVecs, const int *flags, int *numPrevRetained) {
   primme_params *p = s->p;
   int retained = 0;
   for (int i = 0; i < nprevhVecs && retained < *numPrevRetained &&
                   indexOfPreviousVecs + retained < basisSize; i++) {
      if (p->locking == 0 && flags[i] != UNCONV) continue;
      double R = 0.0;
      HS *x = s->prevhVecs + (size_t)i * s->K;
      pa_ortho_local_vec(x, basisSize, s->hVecs, ldh, indexOfPreviousVecs + retained, G, ldG, &R, p->iseed);
      if (fabs(R) < PA_EPS * sqrt(retained + 1.0)) continue;
      memcpy(s->hVecs + (size_t)(indexOfPreviousVecs + retained) * ldh, x, (size_t)basisSize * sizeof(HS));
      retained++;
   }
   *numPrevRetained = retained;
   return 0;
}

/* H = diag(theta) with the dense +k block, coefficient vectors = unit vectors
 * except inside that block (implicit_I); with explicit_I, H and VtBV were
 * recomputed from the data by the fused update and only the block solve remains. */
static int restart_RR(pa_solver *s, int ldh, int newldh, int restartSize, int basisSize,
      int numConverged, int numPrevRetained, int indexOfPreviousVecs, const int *hVecsPerm) {
   primme_params *p = s->p;
   const int K = s->K;
   HS *H = s->H;
   const double aNorm = PA_MAX(p->aNorm, p->stats.estimateLargestSVal);

   if (p->orth == primme_orth_implicit_I) {
      HS *blk = (HS *)malloc((size_t)(numPrevRetained > 0 ? numPrevRetained * numPrevRetained : 1) * sizeof(HS));
      if (!blk) return PRIMME_MALLOC_FAILURE;
      pa_submatrix(s->hVecs + (size_t)indexOfPreviousVecs * ldh, numPrevRetained, ldh, H, basisSize, K,
            blk, numPrevRetained);
      for (int j = 0; j < restartSize; j++)
         for (int i = 0; i < restartSize; i++) H[i + (size_t)j * K] = 0.0;
      for (int j = 0; j < numPrevRetained; j++)
         for (int i = 0; i < numPrevRetained; i++)
            H[(indexOfPreviousVecs + i) + (size_t)(indexOfPreviousVecs + j) * K] = blk[i + (size_t)j * numPrevRetained];
      for (int j = 0; j < indexOfPreviousVecs; j++) H[j + (size_t)j * K] = s->hVals[j];
      for (int j = indexOfPreviousVecs + numPrevRetained; j < restartSize; j++) H[j + (size_t)j * K] = s->hVals[j];
      free(blk);
   }

   const int nLocked = p->numOrthoConst + (p->locking ? numConverged : 0);
   const HS *G = s->VtBV ? s->VtBV + (size_t)nLocked * s->ldVtBV + nLocked : NULL;
   if (p->targetShifts &&
         (s->targetShiftIndex < 0 ||
               fabs(p->targetShifts[s->targetShiftIndex] -
                     p->targetShifts[PA_MIN(p->numTargetShifts - 1, numConverged)]) > s->mach_eps * aNorm)) {
      /* the target moved: solve the whole restarted problem */
      s->targetShiftIndex = PA_MIN(p->numTargetShifts - 1, numConverged);
      CHK(pa_solve_H_RR(s, H, K, G, s->ldVtBV, s->hVecs, newldh, s->hVals, restartSize, numConverged));
      for (int i = 0; i < restartSize; i++) {
         p->stats.estimateMinEVal = PA_MIN(p->stats.estimateMinEVal, s->hVals[i]);
         p->stats.estimateMaxEVal = PA_MAX(p->stats.estimateMaxEVal, s->hVals[i]);
         p->stats.estimateLargestSVal = PA_MAX(p->stats.estimateLargestSVal, fabs(s->hVals[i]));
      }
      return 0;
   }

   int ordered = restartSize;
   for (int i = 0; i < restartSize; i++)
      if (hVecsPerm[i] == indexOfPreviousVecs) { ordered = i; break; }

   for (int j = 0; j < restartSize; j++) {
      for (int i = 0; i < restartSize; i++) s->hVecs[i + (size_t)j * newldh] = 0.0;
      s->hVecs[hVecsPerm[j] + (size_t)j * newldh] = 1.0;
   }
   pa_permute_reals(s->hVals, 1, restartSize, 1, hVecsPerm);

   if (numPrevRetained > 0) {
      const HS *Gb = G ? G + (size_t)indexOfPreviousVecs * s->ldVtBV + indexOfPreviousVecs : NULL;
      CHK(pa_solve_H_RR(s, H + (size_t)indexOfPreviousVecs * K + indexOfPreviousVecs, K, Gb, s->ldVtBV,
            s->hVecs + (size_t)ordered * newldh + indexOfPreviousVecs, newldh, s->hVals + ordered,
            numPrevRetained, numConverged));
      for (int i = 0; i < numPrevRetained; i++) {
         const double v = s->hVals[ordered + i];
         p->stats.estimateMinEVal = PA_MIN(p->stats.estimateMinEVal, v);
         p->stats.estimateMaxEVal = PA_MAX(p->stats.estimateMaxEVal, v);
         p->stats.estimateLargestSVal = PA_MAX(p->stats.estimateLargestSVal, fabs(v));
      }
   }
   return 0;
}

/* column shuffles on the device: dst_k <- src_k for k < n, executed through the
 * scratch panel so overlapping source/destination ranges are safe */
static int move_cols(pa_solver *s, char *base, const int *src, const int *dst, int n) {
   for (int c0 = 0; c0 < n; c0 += s->nT) {
      int nn = PA_MIN(s->nT, n - c0);
      for (int c = 0; c < nn; c++)
         CHK(hipk_copy_cols(s->ctx, s->dt, s->m, PCOL(s, base, s->ld, src[c0 + c]), s->ld, TCOL(s, c), s->ld, 1));
      for (int c = 0; c < nn; c++)
         CHK(hipk_copy_cols(s->ctx, s->dt, s->m, TCOL(s, c), s->ld, PCOL(s, base, s->ld, dst[c0 + c]), s->ld, 1));
   }
   return 0;
}

/* Before a restart: converged pairs and the block first (coefficient vectors, Ritz values and flags alike;
 * reference main_iter.c:971-998) */
int pa_block_first_reorder(pa_solver *s, int basisSize, int *flags, const int *iev, int blockSize, int numConverged,
      int numLocked) {
   int *iwork = (int *)malloc((size_t)(basisSize > 0 ? basisSize : 1) * sizeof(int));
   if (!iwork) return PRIMME_MALLOC_FAILURE;
   int i, j, k, l, mm;
   for (i = k = l = mm = 0; i < basisSize; i++) {
      int inIev = 0;
      for (j = 0; j < blockSize; j++) if (iev[j] == i) inIev = 1;
      if ((flags[i] != UNCONV && mm++ < numConverged - numLocked) || inIev) iwork[k++] = i;
      else iwork[numConverged - numLocked + blockSize + l++] = i;
   }
   pa_permute_reals(s->hVals, 1, basisSize, 1, iwork);
   pa_permute_cols(s->hVecs, basisSize, basisSize, basisSize, iwork);
   pa_permute_ints(flags, basisSize, iwork);
   if (s->hVecsRot) {
      for (int c = s->numArbitraryVecs; c < basisSize; c++) {
         for (int r = 0; r < s->K; r++) s->hVecsRot[r + (size_t)c * s->K] = 0.0;
         s->hVecsRot[c + (size_t)c * s->K] = 1.0;
      }
      pa_permute_cols(s->hVecsRot, basisSize, basisSize, s->K, iwork);
      int last = 0;
      for (i = 0; i < basisSize; i++) if (iwork[i] != i) last = i + 1;
      s->numArbitraryVecs = PA_MAX(s->numArbitraryVecs, last);
   }
   s->coef_valid_k = -1;
   free(iwork);
   return 0;
}

#define PA_PLANNED 7701     /* dry run of the restart: stopped where the pass would start */
#define PA_PLAN_NONE 7702   /* dry run: this restart is not one the speculative pass covers */

#if !PA_IS_COMPLEX      /* the fused / speculative restart is real-arithmetic code (block size 1, eigs_conv.c) */
/* ---- fused restart ------------------------------------------------------
 * The convergence check at a full basis left the candidate's residual in T(:,2) and its overlaps with
 * the old basis in s->rst_ov (eigs_conv.c).  When the restart keeps that candidate as the next block and
 * locks nothing, the residual job of the restart pass is dropped and the overlaps with the restarted
 * basis V h, W h follow from a k x restartSize host product. */
static int stash_matches(const pa_solver *s, int basisSize, int ldh, int col, int nLk) {
   return s->rst_valid && s->rst_k == basisSize && s->rst_L == nLk && s->hVals[col] == s->rst_theta &&
          memcmp(s->hVecs + (size_t)col * ldh, s->rst_y, (size_t)basisSize * sizeof(double)) == 0;
}
/* G = (W h)'Q = h'(W'Q) for the restarted basis: rows 0..k-2 of W'Q are on the host, row k-1 (`grow`) came out of
 * the pass; h = rs coefficient columns, K-strided */
static int transform_wtq(pa_solver *s, const double *h, int k, int rs, int nLk, const double *grow) {
   const int K = s->K;
   if (nLk > 0) {
      double *g = (double *)malloc((size_t)rs * nLk * sizeof(double));
      if (!g) return PRIMME_MALLOC_FAILURE;
      for (int l = 0; l < nLk; l++)
         for (int cc = 0; cc < rs; cc++) {
            double a = h[(k - 1) + (size_t)cc * K] * grow[l];
            for (int j = 0; j < k - 1; j++) a += h[j + (size_t)cc * K] * s->wtq[j + (size_t)l * K];
            g[cc + (size_t)l * rs] = a;
         }
      for (int l = 0; l < nLk; l++)
         for (int cc = 0; cc < rs; cc++) s->wtq[cc + (size_t)l * K] = g[cc + (size_t)l * rs];
      free(g);
   }
   s->wtq_rows = rs; s->wtq_L = nLk;
   return 0;
}
/* h = the rs leading coefficient columns just pushed (K-strided in h_coef) */
static int stash_transform(pa_solver *s, int basisSize, int rs, int nLk) {
   const int k = basisSize, nov = k + nLk, K = s->K;
   const double *ovV = s->rst_ov, *ovQ = s->rst_ov + k, *ovW = s->rst_ov + nov + 1, *grow = s->rst_ov + nov + 1 + k;
   double *c = s->rst_c;
   for (int cc = 0; cc < rs; cc++) {
      double a = 0.0, b = 0.0;
      for (int j = 0; j < k; j++) { a += s->h_coef[j + (size_t)cc * K] * ovV[j]; b += s->h_coef[j + (size_t)cc * K] * ovW[j]; }
      c[cc] = a; c[rs + nLk + 1 + cc] = b;
   }
   for (int l = 0; l < nLk; l++) c[rs + l] = ovQ[l];
   c[rs + nLk] = s->rst_ov[nov];
   CHK(transform_wtq(s, s->h_coef, k, rs, nLk, grow));
   s->rst_ready = 1; s->rst_rs = rs;
   return 0;
}

#endif

static int restart_soft_locking(pa_solver *s, int *restartSize, int basisSize, int ldh,
      int *restartPerm, int *flags, int *iev, int *ievSize, double *blockNorms, double *evals,
      double *resNorms, int *numConverged, int numPrevRetained, int *indexOfPreviousVecs,
      int *hVecsPerm) {
   primme_params *p = s->p;
   int i, j, k;

   /* a previously converged pair whose Ritz value drifted more than its residual
    * norm is targeted again */
   *numConverged = 0;
   for (i = 0; i < p->numEvals; i++) {
      if (flags[i] != UNCONV && fabs(s->hVals[i] - evals[i]) > resNorms[i]) {
         flags[i] = UNCONV;
      } else if (flags[i] != UNCONV) {
         if (flags[i] == CONV) {
            if (*numConverged == 0) p->stats.maxConvTol = 0.0;
            p->stats.maxConvTol = PA_MAX(p->stats.maxConvTol, resNorms[i]);
         }
         (*numConverged)++;
      }
   }

   *indexOfPreviousVecs = *restartSize;
   *restartSize += numPrevRetained;
   *ievSize = PA_MAX(0, PA_MIN(PA_MIN(PA_MIN(PA_MIN(PA_MIN(*ievSize, p->maxBlockSize),
                                                p->numEvals - *numConverged + 1),
                                         p->maxBasisSize - *restartSize),
                                  basisSize - *numConverged),
                           p->minRestartSize - *numConverged));

   for (i = j = k = 0; i < basisSize; i++) {
      if (k >= *numConverged || flags[i] == UNCONV) restartPerm[*numConverged + j++] = i;
      else restartPerm[k++] = i;
   }
   pa_permute_reals(s->hVals, 1, basisSize, 1, restartPerm);
   pa_permute_cols(s->hVecs, basisSize, basisSize, ldh, restartPerm);
#if !PA_IS_COMPLEX
   if (s->plan_only) {        /* dry run (see restart_locking): the candidate is column numConverged */
      const int rs0 = *restartSize, K = s->K, nc0 = *numConverged;
      if (*ievSize != 1 || rs0 > 16 || rs0 + 1 > K || !s->h_coef2 || nc0 >= rs0) return PA_PLAN_NONE;
      for (int c = 0; c < rs0; c++) {
         memcpy(s->h_coef2 + (size_t)c * K, s->hVecs + (size_t)c * ldh, (size_t)basisSize * sizeof(double));
         for (int r2 = basisSize; r2 < K; r2++) s->h_coef2[r2 + (size_t)c * K] = 0.0;
      }
      for (int r2 = 0; r2 < K; r2++) s->h_coef2[r2 + (size_t)rs0 * K] = (r2 == basisSize - 1) ? 1.0 : 0.0;
      memcpy(s->h_theta2, s->hVals, (size_t)basisSize * sizeof(double));
      s->pl_rs = rs0; s->pl_k = basisSize; s->pl_L = p->numOrthoConst; s->pl_cand = nc0; s->pl_nc = nc0;
      return PA_PLANNED;
   }
   int use_plan = 0;
   if (s->pl_launched && *ievSize == 1 && s->fuse_gd && s->V2 && *restartSize == s->pl_rs && basisSize == s->pl_k &&
         p->numOrthoConst == s->pl_L && *numConverged == s->pl_nc && s->pl_cand == *numConverged &&
         s->hVals[*numConverged] == s->h_theta2[*numConverged]) {
      use_plan = 1;
      for (int c = 0; c < *restartSize && use_plan; c++)
         if (memcmp(s->hVecs + (size_t)c * ldh, s->h_coef2 + (size_t)c * s->K, (size_t)basisSize * sizeof(double))) use_plan = 0;
   }
   s->pl_launched = 0;
   s->coef_valid_k = -1;
   if (use_plan) {
      /* the speculative pass already wrote V h, W h (and the converged Ritz vectors) with exactly this block */
      const int rs = *restartSize;
      char *t = s->V; s->V = s->V2; s->V2 = t;
      t = s->W; s->W = s->W2; s->W2 = t;
      blockNorms[0] = sqrt(s->rst_c[rs + p->numOrthoConst]);
      CHK(transform_wtq(s, s->h_coef2, basisSize, rs, p->numOrthoConst, s->rst_grow));
      s->rst_ready = 1; s->rst_rs = rs;
      CHK(refresh_gram_after_restart(s, p->numOrthoConst, basisSize, rs, ldh));
      for (i = 0; i < basisSize; i++) hVecsPerm[restartPerm[i]] = i;
      for (i = 0; i < *ievSize; i++)
         for (j = 0; j < *restartSize; j++)
            if (hVecsPerm[j] == *numConverged + i) iev[i] = j;
      return 0;
   }
#endif
   s->coef_valid_k = -1;
   CHK(pa_push_coefficients(s, basisSize, ldh));

   /* one pass: V, W <- V h, W h (in place); X, R for the next block; converged
    * Ritz vectors copied out to evecs */
   const int rs = *restartSize, nc = *numConverged, nb = *ievSize;
   hipk_job *jobs = (hipk_job *)malloc((size_t)(2 * rs + 2 * nb + nc + 4) * sizeof(hipk_job));
   if (!jobs) return PRIMME_MALLOC_FAILURE;
   int nj = 0;
   for (int c = 0; c < rs; c++) jobs[nj++] = (hipk_job){HIPK_JOB_XV, c, VCOL(s, c), -1};
   if (!s->fuse_gd)
      for (int c = 0; c < nb; c++) jobs[nj++] = (hipk_job){HIPK_JOB_XV, nc + c, VCOL(s, rs + c), -1};
   for (int c = 0; c < nc; c++) jobs[nj++] = (hipk_job){HIPK_JOB_XV, c, ECOL(s, p->numOrthoConst + c), -1};
   for (int c = 0; c < rs; c++) jobs[nj++] = (hipk_job){HIPK_JOB_XW, c, WCOL(s, c), -1};
#if PA_IS_COMPLEX
   const int use_stash = 0;
#else
   const int use_stash = (nb == 1 && s->fuse_gd && stash_matches(s, basisSize, ldh, nc, p->numOrthoConst));
#endif
   if (!use_stash)
      for (int c = 0; c < nb; c++)
         jobs[nj++] = (hipk_job){HIPK_JOB_RES, nc + c, s->fuse_gd ? VCOL(s, rs + c) : WCOL(s, rs + c), c};
   int rc = pa_ritz_update(s, basisSize, jobs, nj, blockNorms, use_stash ? 0 : nb, (int64_t)2 * rs + 2 * nb + nc);
   free(jobs);
   if (rc) return rc;
#if !PA_IS_COMPLEX
   if (use_stash) {
      blockNorms[0] = sqrt(s->rst_ov[basisSize + p->numOrthoConst]);
      CHK(stash_transform(s, basisSize, rs, p->numOrthoConst));
   }
#endif
   CHK(refresh_gram_after_restart(s, p->numOrthoConst, basisSize, rs, ldh));

   for (i = 0; i < basisSize; i++) hVecsPerm[restartPerm[i]] = i;
   for (i = 0; i < *ievSize; i++)
      for (j = 0; j < *restartSize; j++)
         if (hVecsPerm[j] == *numConverged + i) iev[i] = j;
   return 0;
}

static int restart_locking(pa_solver *s, int *restartSize, int basisSize, int ldh,
      int *restartPerm, int *flags, int *iev, int *ievSize, double *blockNorms, double *evals,
      int *numConverged, int *numLocked, double *resNorms, int *lockedFlags, int *evecsperm,
      int numPrevRetained, int *indexOfPreviousVecs, int *hVecsPerm) {
   primme_params *p = s->p;
   int i, j, k, numPacked, failed;
   const int numLocked0 = *numLocked;
   const int nOC = p->numOrthoConst;

   int maxBlockSize = PA_MAX(0, PA_MIN(PA_MIN(*restartSize, p->maxBlockSize), p->numEvals - *numConverged + 1));
   int sizeBlockNorms = PA_MAX(0, PA_MIN(maxBlockSize,
         p->maxBasisSize - *restartSize - numPrevRetained - *numConverged + *numLocked));
   *indexOfPreviousVecs = *restartSize;
   const int left = *restartSize + numPrevRetained;

   for (i = k = numPacked = 0; i < basisSize; i++) {
      if (flags[i] != UNCONV && numPacked < p->numEvals - *numLocked &&
            (i < p->numEvals - *numLocked || p->target == primme_closest_geq ||
                  p->target == primme_closest_leq)) {
         restartPerm[left + numPacked++] = i;
      } else if (k < left) {
         restartPerm[k++] = i;
      } else {
         restartPerm[PA_MIN(*numConverged, p->numEvals) - *numLocked + k++] = i;
      }
   }
   *restartSize = left + numPacked;

   pa_permute_reals(s->hVals, 1, basisSize, 1, restartPerm);
   pa_permute_cols(s->hVecs, basisSize, basisSize, ldh, restartPerm);
#if !PA_IS_COMPLEX
   if (s->plan_only) {
      /* dry run: this is the coefficient block the pass would get.  One more column, a unit vector, makes the
       * pass copy W(:,k-1) next to the residual (its inner products with Q are needed, eigs_conv.c) */
      const int rs0 = *restartSize, K = s->K;
      if (sizeBlockNorms != 1 || numPacked != 0 || rs0 > 16 || rs0 + 1 > K || !s->h_coef2) return PA_PLAN_NONE;
      for (int c = 0; c < rs0; c++) {
         memcpy(s->h_coef2 + (size_t)c * K, s->hVecs + (size_t)c * ldh, (size_t)basisSize * sizeof(double));
         for (int r2 = basisSize; r2 < K; r2++) s->h_coef2[r2 + (size_t)c * K] = 0.0;
      }
      for (int r2 = 0; r2 < K; r2++) s->h_coef2[r2 + (size_t)rs0 * K] = (r2 == basisSize - 1) ? 1.0 : 0.0;
      memcpy(s->h_theta2, s->hVals, (size_t)basisSize * sizeof(double));
      s->pl_rs = rs0; s->pl_k = basisSize; s->pl_L = nOC + *numLocked; s->pl_cand = 0; s->pl_nc = 0;
      return PA_PLANNED;
   }
   /* the speculative pass already ran with exactly this block (eigs_conv.c): adopt its panels */
   int use_plan = 0;
   if (s->pl_launched && sizeBlockNorms == 1 && numPacked == 0 && s->fuse_gd && s->V2 && *restartSize == s->pl_rs &&
         basisSize == s->pl_k && nOC + *numLocked == s->pl_L && s->pl_cand == 0 && s->pl_nc == 0 && s->hVals[0] == s->h_theta2[0]) {
      use_plan = 1;
      for (int c = 0; c < *restartSize && use_plan; c++)
         if (memcmp(s->hVecs + (size_t)c * ldh, s->h_coef2 + (size_t)c * s->K, (size_t)basisSize * sizeof(double))) use_plan = 0;
   }
   s->pl_launched = 0;
#else
   const int use_plan = 0;
#endif
   s->coef_valid_k = -1;
   if (!use_plan) CHK(pa_push_coefficients(s, basisSize, ldh));

   /* one fused pass over V and W */
   double *lockedResNorms = &resNorms[*numLocked];
   const int rs = *restartSize;
   char *X = VCOL(s, rs), *R = WCOL(s, rs);
   hipk_job *jobs = (hipk_job *)malloc((size_t)(2 * rs + 2 * sizeBlockNorms + 2 * numPacked + 4) * sizeof(hipk_job));
   double *norms = (double *)malloc((size_t)(sizeBlockNorms + numPacked + 1) * sizeof(double));
   if (!jobs || !norms) return PRIMME_MALLOC_FAILURE;
   int nj = 0;
   for (int c = 0; c < rs; c++) jobs[nj++] = (hipk_job){HIPK_JOB_XV, c, VCOL(s, c), -1};
   if (!s->fuse_gd)
      for (int c = 0; c < sizeBlockNorms; c++) jobs[nj++] = (hipk_job){HIPK_JOB_XV, c, PCOL(s, X, s->ld, c), -1};
   for (int c = 0; c < numPacked; c++) jobs[nj++] = (hipk_job){HIPK_JOB_XV, left + c, ECOL(s, *numLocked + nOC + c), -1};
   for (int c = 0; c < rs; c++) jobs[nj++] = (hipk_job){HIPK_JOB_XW, c, WCOL(s, c), -1};
#if !PA_IS_COMPLEX
   if (use_plan) {
      /* nothing to run: V2 / W2 hold V h, W h; the residual is in T(:,2); rst_c has its overlaps */
      char *t = s->V; s->V = s->V2; s->V2 = t;
      t = s->W; s->W = s->W2; s->W2 = t;
      blockNorms[0] = sqrt(s->rst_c[rs + nOC + *numLocked]);
      CHK(transform_wtq(s, s->h_coef2, basisSize, rs, nOC + *numLocked, s->rst_grow));
      s->rst_ready = 1; s->rst_rs = rs;
      free(jobs); free(norms);
      jobs = NULL; norms = NULL;
   }
   const int use_stash = use_plan || (sizeBlockNorms == 1 && numPacked == 0 && s->fuse_gd &&
                          stash_matches(s, basisSize, ldh, 0, nOC + *numLocked));
   if (use_plan) goto pass_done;
#else
   const int use_stash = 0;
#endif
   if (!use_stash)
      for (int c = 0; c < sizeBlockNorms; c++)
         jobs[nj++] = (hipk_job){HIPK_JOB_RES, c, s->fuse_gd ? PCOL(s, X, s->ld, c) : PCOL(s, R, s->ld, c), c};
   for (int c = 0; c < numPacked; c++) jobs[nj++] = (hipk_job){HIPK_JOB_RES, left + c, NULL, sizeBlockNorms + c};
   int rc = pa_ritz_update(s, basisSize, jobs, nj, norms, use_stash ? 0 : sizeBlockNorms + numPacked,
         (int64_t)2 * rs + 2 * sizeBlockNorms + 2 * numPacked);
   free(jobs);
   if (rc) { free(norms); return rc; }
#if !PA_IS_COMPLEX
   if (use_stash) {
      norms[0] = sqrt(s->rst_ov[basisSize + nOC + *numLocked]);
      rc = stash_transform(s, basisSize, rs, nOC + *numLocked);
      if (rc) { free(norms); return rc; }
   }
#endif
   for (int c = 0; c < sizeBlockNorms; c++) blockNorms[c] = norms[c];
   for (int c = 0; c < numPacked; c++)
      lockedResNorms[c] = PA_MAX(norms[sizeBlockNorms + c], p->stats.estimateResidualError);
   free(norms);
#if !PA_IS_COMPLEX
pass_done:
#endif

   CHK(refresh_gram_after_restart(s, *numLocked + nOC, basisSize, rs, ldh));

   /* re-test the pairs about to be locked with their true residual norms */
   pa_permute_ints(flags, basisSize, restartPerm);
   CHK(pa_check_convergence(s, VCOL(s, left), s->ld, 1, NULL, 0, 0, *numLocked, left, left + numPacked,
         flags, lockedResNorms, s->hVals, NULL, 0));

   for (i = left, j = 0; i < left + numPacked; i++) {
      if (flags[i] != UNCONV && *numLocked + j < p->numEvals) evals[*numLocked + j++] = s->hVals[i];
      else flags[i] = UNCONV;
   }

   /* pairs that failed the re-test go back into the basis right after the
    * restarted vectors and take part in the next block */
   int *ifailed = (int *)malloc((size_t)(numPacked > 0 ? numPacked : 1) * sizeof(int));
   if (!ifailed) return PRIMME_MALLOC_FAILURE;
   for (i = left, failed = 0; i < left + numPacked; i++) if (flags[i] == UNCONV) ifailed[failed++] = i - left;
   for (i = left, j = 0; i < left + numPacked; i++) if (flags[i] != UNCONV) ifailed[failed + j++] = i - left;

   maxBlockSize = PA_MAX(0, PA_MIN(maxBlockSize,
         p->maxBasisSize - *restartSize - numPrevRetained - *numConverged + *numLocked));
   {
      double *bn0 = (double *)malloc((size_t)(sizeBlockNorms > 0 ? sizeBlockNorms : 1) * sizeof(double));
      for (i = 0; i < sizeBlockNorms; i++) bn0[i] = blockNorms[i];
      for (i = j = k = 0; i < *indexOfPreviousVecs || j < failed; k++) {
         if (i < *indexOfPreviousVecs && (j >= failed || restartPerm[i] < restartPerm[left + ifailed[j]])) {
            if (k < maxBlockSize && i < sizeBlockNorms) blockNorms[k] = bn0[i];
            hVecsPerm[k] = i++;
         } else {
            if (k < maxBlockSize) blockNorms[k] = resNorms[numLocked0 + ifailed[j]];
            hVecsPerm[k] = left + j++;
         }
      }
      free(bn0);
      for (i = 0; i < numPrevRetained; i++) hVecsPerm[k++] = i + *indexOfPreviousVecs;
      for (; k < basisSize; k++) hVecsPerm[k] = -1;
   }

   /* Build the next block's X and R from the candidate columns (xo, ro: already
    * computed above) merged in original order with the failed ones (their X is
    * the Ritz vector, R = W - theta*V), and compact the failed columns to
    * V(:,left..left+failed) / W(...).  (reference restart.c:1047-1052,
    * compute_residual_columns :2464-2538) */
   if (!(use_stash && failed == 0)) {
      const int nd = maxBlockSize;
      /* stage the merged block in the scratch panel: X in T[0..nd), R in T[nd..2nd) */
      int io = 0, ifl = 0;
      for (int id = 0; id < nd; id++) {
         if (io < sizeBlockNorms && hVecsPerm[id] == io) {
            /* fused GD mode: the candidate's residual sits in the X slot and there is no X */
            if (!s->fuse_gd) CHK(hipk_copy_cols(s->ctx, s->dt, s->m, PCOL(s, X, s->ld, io), s->ld, TCOL(s, id), s->ld, 1));
            CHK(hipk_copy_cols(s->ctx, s->dt, s->m, s->fuse_gd ? PCOL(s, X, s->ld, io) : PCOL(s, R, s->ld, io), s->ld, TCOL(s, nd + id), s->ld, 1));
            io++;
         } else if (ifl < failed) {
            const int src = left + ifailed[ifl];
            if (!s->fuse_gd) CHK(hipk_copy_cols(s->ctx, s->dt, s->m, VCOL(s, src), s->ld, TCOL(s, id), s->ld, 1));
            CHK(hipk_copy_cols(s->ctx, s->dt, s->m, WCOL(s, src), s->ld, TCOL(s, nd + id), s->ld, 1));
            double th = s->hVals[src];
            const char *vm = VCOL(s, src);
            if (s->B) { CHK(pa_apply_B(s, VCOL(s, src), s->ld, s->BT, s->ld, 1)); vm = s->BT; }      /* R = W - theta B V */
            CHK(hipk_residual_cols(s->ctx, s->dt, s->m, vm, s->ld, TCOL(s, nd + id), s->ld, 1, &th, s->d_red));
            ifl++;
         } else {
            break;
         }
      }
      const int nbuilt = io + ifl < nd ? io + ifl : nd;
      /* compact failed columns (ascending, sources are never below destinations) */
      for (i = 0; i < failed; i++) {
         if (ifailed[i] != i) {
            CHK(hipk_copy_cols(s->ctx, s->dt, s->m, VCOL(s, left + ifailed[i]), s->ld, VCOL(s, left + i), s->ld, 1));
            CHK(hipk_copy_cols(s->ctx, s->dt, s->m, WCOL(s, left + ifailed[i]), s->ld, WCOL(s, left + i), s->ld, 1));
         }
      }
      if (nbuilt > 0) {
         if (s->fuse_gd) {
            CHK(hipk_copy_cols(s->ctx, s->dt, s->m, TCOL(s, nd), s->ld, VCOL(s, left + failed), s->ld, nbuilt));
         } else {
            CHK(hipk_copy_cols(s->ctx, s->dt, s->m, TCOL(s, 0), s->ld, VCOL(s, left + failed), s->ld, nbuilt));
            CHK(hipk_copy_cols(s->ctx, s->dt, s->m, TCOL(s, nd), s->ld, WCOL(s, left + failed), s->ld, nbuilt));
         }
      }
   }

   /* the same compaction on the host-side coefficient data */
   {
      HS *tmp = (HS *)malloc((size_t)basisSize * (failed > 0 ? failed : 1) * sizeof(HS));
      double *tv = (double *)malloc((size_t)(failed > 0 ? failed : 1) * sizeof(double));
      for (i = 0; i < failed; i++) {
         memcpy(tmp + (size_t)i * basisSize, s->hVecs + (size_t)(left + ifailed[i]) * ldh, (size_t)basisSize * sizeof(HS));
         tv[i] = s->hVals[left + ifailed[i]];
      }
      for (i = 0; i < failed; i++) {
         memcpy(s->hVecs + (size_t)(left + i) * ldh, tmp + (size_t)i * basisSize, (size_t)basisSize * sizeof(HS));
         s->hVals[left + i] = tv[i];
      }
      free(tmp);
      free(tv);
      pa_permute_ints(&restartPerm[left], numPacked, ifailed);
   }
   if (s->VtBV) {
      /* accepted-to-lock columns move to the locked block, then the restarted ones, then the
       * failed ones (reference restart.c:1092-1117) */
      const int nLk = nOC + *numLocked, tot = left + numPacked, nG = nLk + tot, ldG = s->ldVtBV;
      int *iV = (int *)malloc((size_t)(tot > 0 ? tot : 1) * sizeof(int));
      HS *rw = (HS *)malloc((size_t)nG * (tot > 0 ? tot : 1) * sizeof(HS));
      for (i = 0; i < numPacked - failed; i++) iV[i] = ifailed[failed + i] + left;
      for (i = 0; i < left; i++) iV[i + numPacked - failed] = i;
      for (i = 0; i < failed; i++) iV[i + left + numPacked - failed] = ifailed[i] + left;
      for (int c = 0; c < tot; c++)
         for (int r2 = 0; r2 < nG; r2++) rw[r2 + (size_t)c * nG] = s->VtBV[r2 + (size_t)(nLk + iV[c]) * ldG];
      for (int c = 0; c < tot; c++) {
         for (int r2 = 0; r2 < nLk; r2++) s->VtBV[r2 + (size_t)(nLk + c) * ldG] = rw[r2 + (size_t)c * nG];
         for (int r2 = 0; r2 < tot; r2++) s->VtBV[(nLk + r2) + (size_t)(nLk + c) * ldG] = rw[(nLk + iV[r2]) + (size_t)c * nG];
      }
      free(iV); free(rw);
      /* H: failed rows/columns right after the restarted ones (reference :1119-1124) */
      HS *hc = (HS *)malloc((size_t)(tot > 0 ? tot : 1) * (failed > 0 ? failed : 1) * sizeof(HS));
      for (int c = 0; c < failed; c++)
         for (int r2 = 0; r2 < tot; r2++) hc[r2 + (size_t)c * tot] = s->H[r2 + (size_t)(left + ifailed[c]) * s->K];
      for (int c = 0; c < failed; c++)
         for (int r2 = 0; r2 < tot; r2++) s->H[r2 + (size_t)(left + c) * s->K] = hc[r2 + (size_t)c * tot];
      for (int c = 0; c < left + failed; c++) {
         HS tmpc[64];
         for (int r2 = 0; r2 < failed && r2 < 64; r2++) tmpc[r2] = s->H[(left + ifailed[r2]) + (size_t)c * s->K];
         for (int r2 = 0; r2 < failed && r2 < 64; r2++) s->H[(left + r2) + (size_t)c * s->K] = tmpc[r2];
      }
      free(hc);
   }

   /* lock: pack the accepted vectors in evecs and insertion-sort their values */
   for (i = left; i < left + numPacked; i++) {
      if (flags[i] != UNCONV && *numLocked < p->numEvals) {
         const double resNorm = resNorms[*numLocked] = lockedResNorms[i - left];
         const double eval = evals[*numLocked];
         if (numLocked0 + i - left != *numLocked)
            CHK(hipk_copy_cols(s->ctx, s->dt, s->m, ECOL(s, numLocked0 + i - left + nOC), s->ldevecs,
                  ECOL(s, *numLocked + nOC), s->ldevecs, 1));
         (*numLocked)++;
         lockedFlags[*numLocked - 1] = flags[i];
         pa_monitor(s, NULL, 0, NULL, NULL, 0, NULL, 0, evals, *numLocked, lockedFlags, resNorms, primme_event_locked);
         CHK(insertion_sort(eval, evals, resNorm, resNorms, flags[i], lockedFlags, evecsperm, *numLocked - 1, 0, p));
         if (flags[i] == CONV) p->stats.maxConvTol = PA_MAX(p->stats.maxConvTol, resNorm);
      }
   }
   free(ifailed);

   *restartSize = left + failed;
   *ievSize = PA_MIN(maxBlockSize, sizeBlockNorms + failed);
   *numConverged = *numLocked;
   for (i = 0; i < *ievSize; i++) iev[i] = i;
   for (i = 0; i < basisSize; i++) flags[i] = UNCONV;
   return 0;
}

int pa_restart(pa_solver *s, int basisSize, int *flags, int *iev, int *ievSize, double *blockNorms,
      int *evecsPerm, double *evals, double *resNorms, int *numConverged, int *numLocked,
      int *lockedFlags, int nprevhVecs, int numGuesses, int *restartSizeOutput,
      int *restartsSinceReset) {
   primme_params *p = s->p;
   const int ldh = basisSize;
   int i, restartSize;
   if (!s->plan_only) {
      s->fov_valid = 0;
      s->fov_carry = 0;
      s->rst_ready = 0;
      pa_pre_discard(s);        /* an iteration enqueued ahead of the host belongs to the basis that is being replaced */
   }

   for (i = 0, *numConverged = *numLocked; i < basisSize; i++) {
      if (flags[i] == SKIP_RESTART) flags[i] = UNCONV;
      else if (flags[i] != UNCONV && *numConverged < p->numEvals &&
               (i < p->numEvals - *numLocked || p->target == primme_closest_geq ||
                     p->target == primme_closest_leq))
         (*numConverged)++;
   }

   int numPrevRetained = p->restartingParams.maxPrevRetain;
   if (!p->locking && basisSize + *numLocked + p->numOrthoConst >= p->n) {
      restartSize = basisSize;
      numPrevRetained = 0;
   } else if (basisSize <= p->maxBasisSize - p->maxBlockSize) {
      restartSize = basisSize;
      numPrevRetained = 0;
   } else {
      restartSize = PA_MIN(basisSize, p->minRestartSize);
   }
   restartSize -= PA_MIN(PA_MIN(numGuesses, *numConverged - *numLocked), restartSize);
   if (p->locking) restartSize = PA_MIN(restartSize, basisSize - (*numConverged - *numLocked));
   numPrevRetained = PA_MAX(0, PA_MIN(PA_MIN(numPrevRetained, p->maxBasisSize - restartSize - 1),
                                 (int)(p->n - restartSize - *numConverged - p->numOrthoConst)));

   int indexOfPreviousVecs = p->locking ? restartSize + *numConverged - *numLocked : restartSize;
   const int indexOfPreviousVecsBeforeRestart = indexOfPreviousVecs;
   const int nLocked = p->numOrthoConst + *numLocked;
   const HS *G = s->VtBV ? s->VtBV + (size_t)nLocked * s->ldVtBV + nLocked : NULL;
   CHK(ortho_coefficient_vectors(s, basisSize, ldh, indexOfPreviousVecs, G, s->ldVtBV, nprevhVecs, flags,
         &numPrevRetained));

   int *restartPerm = (int *)malloc((size_t)basisSize * sizeof(int));
   int *hVecsPerm = (int *)malloc((size_t)basisSize * sizeof(int));
   if (!restartPerm || !hVecsPerm) return PRIMME_MALLOC_FAILURE;
   int rc;
   if (!p->locking)
      rc = restart_soft_locking(s, &restartSize, basisSize, ldh, restartPerm, flags, iev, ievSize,
            blockNorms, evals, resNorms, numConverged, numPrevRetained, &indexOfPreviousVecs, hVecsPerm);
   else
      rc = restart_locking(s, &restartSize, basisSize, ldh, restartPerm, flags, iev, ievSize, blockNorms,
            evals, numConverged, numLocked, resNorms, lockedFlags, evecsPerm, numPrevRetained,
            &indexOfPreviousVecs, hVecsPerm);
   if (rc) { free(restartPerm); free(hVecsPerm); return rc; }

   if (s->fVtBV) {
      const int newnLocked = p->numOrthoConst + *numLocked;
      CHK(pa_update_cholesky(s->VtBV, s->ldVtBV, s->fVtBV, s->ldVtBV, nLocked, newnLocked + restartSize));
   }

   /* previous Ritz values follow the basis (only used by interior-target shifts) */
   if (p->target != primme_smallest && p->target != primme_largest) {
      if (s->numPrevRitzVals > 0) {
         for (i = s->numPrevRitzVals; i < basisSize; i++) s->prevRitzVals[i] = s->prevRitzVals[s->numPrevRitzVals - 1];
         pa_permute_reals(s->prevRitzVals, 1, basisSize, 1, restartPerm);
      }
      for (i = 0; i < restartSize; i++)
         if (restartPerm[i] >= s->numPrevRitzVals) s->prevRitzVals[i] = s->hVals[i];
      pa_permute_reals(s->prevRitzVals, 1, restartSize, 1, hVecsPerm);
      s->numPrevRitzVals = restartSize;
   }

   if (s->refined) {
      rc = pa_restart_refined(s, ldh, restartSize, basisSize, *numConverged, numPrevRetained, indexOfPreviousVecs,
            indexOfPreviousVecsBeforeRestart, restartPerm, hVecsPerm, &s->numArbitraryVecs);
      for (i = 0; i < restartSize && !rc; i++) {
         p->stats.estimateMinEVal = PA_MIN(p->stats.estimateMinEVal, s->hVals[i]);
         p->stats.estimateMaxEVal = PA_MAX(p->stats.estimateMaxEVal, s->hVals[i]);
         p->stats.estimateLargestSVal = PA_MAX(p->stats.estimateLargestSVal, fabs(s->hVals[i]));
      }
   } else if (s->Q) {
      /* harmonic extraction: fresh QR of (A - tau I) V for the restarted basis, then the projected
       * problem from scratch (reference restart.c:2255-2326) */
      rc = pa_restart_harmonic(s, ldh, restartSize, basisSize, *numConverged);
      if (!rc) rc = pa_solve_H(s, restartSize, p->locking ? *numConverged : 0, *numConverged);
   } else
      rc = restart_RR(s, ldh, restartSize, restartSize, basisSize, *numConverged, numPrevRetained,
            indexOfPreviousVecs, hVecsPerm);
   free(restartPerm);
   if (rc) { free(hVecsPerm); return rc; }
   s->coef_valid_k = -1;

   if (*numConverged >= p->numEvals && !p->locking) {
      /* all wanted pairs converged: bring them to the front of V and W */
      int *inv = (int *)malloc((size_t)restartSize * sizeof(int));
      int *ident = (int *)malloc((size_t)restartSize * sizeof(int));
      for (i = 0; i < restartSize; i++) { inv[i] = hVecsPerm[i]; ident[i] = i; }
      rc = move_cols(s, s->V, inv, ident, restartSize);
      if (!rc) rc = move_cols(s, s->W, inv, ident, restartSize);
      free(inv);
      free(ident);
      if (rc) { free(hVecsPerm); return rc; }
      s->rst_ready = 0;      /* the basis columns moved */
   }
   free(hVecsPerm);
   *restartSizeOutput = restartSize;
   s->rst_valid = 0;

   /* bound on the error accumulated in V and W (implicit_I branch of reference
    * restart.c:418-451; the explicit_I estimate from VtBV is added with that path) */
   double fn = 0.0;
   if (s->VtBV) {
      /* orthogonality level ||I - V'V||_F of [locked V] from the tracked Gram matrix
       * (reference restart.c:398-446) */
      const int nG = p->numOrthoConst + *numLocked + restartSize, ldG = s->ldVtBV;
      double acc = 0.0;
      for (int c = 0; c < nG; c++)
         for (int r2 = 0; r2 < c; r2++) {
            const double g2 = HS_ABS2(s->VtBV[r2 + (size_t)c * ldG]);
            acc += 2 * g2 / HS_ABS(s->VtBV[c + (size_t)c * ldG]) / HS_ABS(s->VtBV[r2 + (size_t)r2 * ldG]);
         }
      fn = sqrt(acc);
   }
   if (fn > 0.0) {
      if (*restartsSinceReset <= 1)
         p->stats.maxConvTol = PA_MAX(p->stats.maxConvTol, fn * p->stats.estimateLargestSVal);
      p->stats.estimateResidualError = sqrt((double)*restartsSinceReset) * fn * pa_problem_norm(1, p);
   } else {
      p->stats.estimateResidualError =
            2 * sqrt((double)*restartsSinceReset) * s->mach_eps * pa_problem_norm(1, p);
   }
#if !PA_IS_COMPLEX
   if (s->rst_ready && restartSize == s->rst_rs && *ievSize == 1 && restartSize + 1 <= p->maxBasisSize && numGuesses <= 0) {
      /* first iteration after the restart: Gram-Schmidt update, operator and the new column of H from the
       * transformed overlaps; the residual is still in T(:,2), the normalised vector lands in V(:,restartSize) */
      const int rs = restartSize, nLk = p->numOrthoConst + *numLocked, nov = rs + nLk, nfov = 2 * nov + 1;
      memcpy(s->h_fov, s->rst_c, (size_t)(2 * rs + nLk + 1) * sizeof(double));
      for (i = 2 * rs + nLk + 1; i < nfov; i++) s->h_fov[i] = 0.0;
      /* (all of them: the Gram-Schmidt update reads the first nov + 1, the one-wave Rayleigh-Ritz kernel of the iteration
       * that is enqueued behind this tail reads W'r as well) */
      CHK(hipk_h2d(s->ctx, s->d_fov, s->h_fov, (size_t)nfov * sizeof(double)));
      CHK(pa_speculative_tail(s, rs, nLk, TCOL(s, 2), VCOL(s, rs), nfov, 1, 1, 0, iev[0], 0));
      s->fov_valid = 1; s->fov_k = rs; s->fov_L = nLk; s->fov_col = VCOL(s, rs); s->fov_s1_off = nfov;
      s->fov_carry = 1;
   } else if (s->rst_ready) {
      /* the stash was used by the restart pass but the tail cannot follow: put the residual where the
       * residual job would have left it */
      CHK(hipk_copy_cols(s->ctx, s->dt, s->m, TCOL(s, 2), s->ld, VCOL(s, s->rst_rs), s->ld, 1));
   }
#endif
   s->rst_ready = 0;
   return 0;
}

#if !PA_IS_COMPLEX

/* Dry run of what the main loop and pa_restart do between the convergence check at a full basis and the restart
 * pass, under the assumption that the candidate iev_in[nblock-1] is NOT converged: on success (0) h_coef2 /
 * h_theta2 hold the coefficient block and Ritz values the pass would be called with, pl_rs its width.  The real
 * functions run on the live coefficient data, which is saved and put back. */
int pa_restart_plan(pa_solver *s, int basisSize, const int *flags_in, const int *iev_in, int nblock, int numLocked,
      int nprevhVecs, const int *map, double *evals, double *resNorms) {
   primme_params *p = s->p;
   const int K = s->K;
   if (!s->V2 || !s->plan_allowed || basisSize > K || nblock < 1) return 1;
   const size_t kk = (size_t)K * K;
   double *sv = (double *)malloc((2 * kk + K) * sizeof(double));
   int *fl = (int *)malloc((size_t)(2 * K + 2) * sizeof(int));
   if (!sv || !fl) { free(sv); free(fl); return 1; }
   int *iv = fl + K;
   memcpy(sv, s->hVecs, kk * sizeof(double));
   memcpy(sv + kk, s->prevhVecs, kk * sizeof(double));
   memcpy(sv + 2 * kk, s->hVals, (size_t)K * sizeof(double));
   memcpy(fl, flags_in, (size_t)basisSize * sizeof(int));
   memcpy(iv, iev_in, (size_t)nblock * sizeof(int));
   const primme_stats st = p->stats;
   PRIMME_INT seed[4];
   memcpy(seed, p->iseed, sizeof(seed));
   const int cvk = s->coef_valid_k, nav = s->numArbitraryVecs;
   /* state the dry run must not touch (advisor, round 2): checked afterwards, a plan that moved any of it is dropped */
   const int g_tsi = s->targetShiftIndex, g_npr = s->numPrevRitzVals, g_init = p->initSize;
   const double g_e0 = evals ? evals[0] : 0.0, g_r0 = resNorms ? resNorms[0] : 0.0;

   int numConverged = numLocked;
   for (int i = 0; i < basisSize; i++)
      if (fl[i] != UNCONV && numConverged < p->numEvals &&
            (i < p->numEvals - numLocked || p->target == primme_closest_geq || p->target == primme_closest_leq))
         numConverged++;
   int rc = pa_block_first_reorder(s, basisSize, fl, iv, nblock, numConverged, numLocked);
   if (!rc) {
      pa_permute_cols(s->prevhVecs, basisSize, nprevhVecs, K, map);
      int nc = numConverged, nl = numLocked, ievSize = nblock, rsOut = 0, rsr = 1;
      s->plan_only = 1;
      rc = pa_restart(s, basisSize, fl, iv, &ievSize, NULL, NULL, evals, resNorms, &nc, &nl, NULL, nprevhVecs, 0, &rsOut, &rsr);
      s->plan_only = 0;
   }
   memcpy(s->hVecs, sv, kk * sizeof(double));
   memcpy(s->prevhVecs, sv + kk, kk * sizeof(double));
   memcpy(s->hVals, sv + 2 * kk, (size_t)K * sizeof(double));
   p->stats = st;
   memcpy(p->iseed, seed, sizeof(seed));
   s->coef_valid_k = cvk; s->numArbitraryVecs = nav;
   free(sv); free(fl);
   if (s->targetShiftIndex != g_tsi || s->numPrevRitzVals != g_npr || p->initSize != g_init ||
         (evals && evals[0] != g_e0) || (resNorms && resNorms[0] != g_r0)) {
      s->targetShiftIndex = g_tsi; s->numPrevRitzVals = g_npr; p->initSize = g_init;
      if (evals) evals[0] = g_e0;
      if (resNorms) resNorms[0] = g_r0;
      if (getenv("PRIMME_AMD_TRACE_ERRORS")) fprintf(stderr, "primme_amd: restart dry run touched solver state outside its saved set; plan dropped\n");
      return 1;
   }
   return rc == PA_PLANNED ? 0 : 1;
}
#endif
