/* eigs_conv.c — convergence tests and candidate selection (host control flow that
 * decides when iterations stop; restated exactly so convergence histories follow
 * the reference).
 *   pa_check_convergence   <- reference src/eigs/convergence.c:85-204, :238-281
 *   pa_project_once        <- reference src/eigs/ortho.c:825-934 (B = I)
 *   pa_map_vecs            <- reference src/eigs/solve_projection.c:1009-1065
 *   pa_prepare_candidates  <- reference src/eigs/main_iter.c:1470-1709
 */
#include "eigs_solver.h"
#include "primme_amd_comm.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

double pa_problem_norm(int overrideUser, const primme_params *p);
int pa_reduce(pa_solver *s, double *d_buf, int count, int keep_dev, int defer_sync);
int pa_ritz_update(pa_solver *s, int basisSize, const hipk_job *jobs, int njobs, double *norms_out,
      int nslots, int64_t flop_cols);
int pa_push_coefficients(pa_solver *s, int basisSize, int ldh);

#if !PA_IS_COMPLEX
/* default convTestFun (reference src/eigs/primme_c.c:555-570) */
void pa_conv_test_absolute(double *eval, void *evec, double *rNorm, int *isConv,
      primme_params *p, int *ierr) {
   (void)eval; (void)evec;
   /* machine epsilon of the working precision is kept in convtest when we install it */
   double meps = p->convtest ? *(double *)p->convtest : PA_EPS;
   *isConv = *rNorm < PA_MAX(p->eps, meps * (p->massMatrixMatvec ? 1 : 2)) * pa_problem_norm(0, p);
   *ierr = 0;
}

#else
void pa_conv_test_absolute(double *eval, void *evec, double *rNorm, int *isConv, primme_params *p, int *ierr);
#endif

static int call_conv_test(pa_solver *s, double eval, void *evec, double rnorm, int *isconv) {
   primme_params *p = s->p;
   if (p->convTestFun == pa_conv_test_absolute) {
      *isconv = rnorm < PA_MAX(p->eps, s->mach_eps * (p->massMatrixMatvec ? 1 : 2)) * pa_problem_norm(0, p);      /* (primme_c.c:555-570) */
      return 0;
   }
   return pa_call_conv_test(p, eval, evec, rnorm, isconv);
}

/* X(:,inX) <- (I - Q Q') X(:,inX), norms of the result (one projector pass) */
int pa_project_once(pa_solver *s, char *Q, int64_t ldQ, int nQ, char *X, int64_t ldX,
      const int *inX, int nX, double *norms) {
   primme_params *p = s->p;
   double t0 = pa_wtime();
   for (int c = 0; c < nX; c++) {
      char *x = PCOL(s, X, ldX, inX ? inX[c] : c);
      hipk_seg seg = {Q, ldQ, nQ};
      CHK(hipk_panel_dots(s->ctx, s->dt, s->m, &seg, 1, x, ldX, 1, s->d_red, nQ));
      CHK(pa_reduce(s, s->d_red, SD * nQ, 1, (s->parallel && !s->dev_comm) ? 0 : 1));
      CHK(hipk_panel_project(s->ctx, s->dt, s->m, &seg, 1, s->d_red, nQ, x, ldX, 1, s->d_red + SD * nQ));
      CHK(pa_reduce(s, s->d_red + SD * nQ, 1, 0, 0));
      if (norms) norms[c] = sqrt(s->h_red[SD * nQ]);
      p->stats.numOrthoInnerProds += nQ + 1;
   }
   p->stats.timeOrtho += pa_wtime() - t0;
   return 0;
}

/* flags[left..right) from blockNorms (indexed from 0), hVals indexed like flags. */
int pa_check_convergence(pa_solver *s, char *X, int64_t ldX, int givenX, char *R, int64_t ldR,
      int givenR, int numLocked, int left, int right, int *flags, double *blockNorms,
      const double *hVals, int *reset, int practConvCheck) {
   primme_params *p = s->p;
   const int nb = right - left;
   if (nb <= 0) return 0;
   int *toProject = (int *)malloc((size_t)nb * sizeof(int));
   if (!toProject) return PRIMME_MALLOC_FAILURE;
   int numToProject = 0;
   const double tol = PA_MAX(s->mach_eps * pa_problem_norm(1, p), p->stats.maxConvTol);
   const double attainableTol = p->locking ? sqrt((double)(p->numOrthoConst + numLocked)) * tol : 0.0;

   for (int i = left; i < right; i++) {
      const double bn = blockNorms[i - left];
      const double targetShift = p->numTargetShifts > 0
            ? p->targetShifts[PA_MIN(p->initSize, p->numTargetShifts - 1)] : 0.0;
      if ((p->target == primme_closest_leq && hVals[i] - bn > targetShift) ||
            (p->target == primme_closest_geq && hVals[i] + bn < targetShift)) {
         flags[i] = UNCONV;
         continue;
      }
      if (bn <= p->stats.maxConvTol) { flags[i] = CONV; continue; }
      int isConv = 0;
      int rc = call_conv_test(s, hVals[i], (X && givenX) ? PCOL(s, X, ldX, i - left) : NULL, bn, &isConv);
      if (rc) { free(toProject); return rc; }
      if (isConv) {
         flags[i] = CONV;
      } else if (bn <= p->stats.estimateResidualError && reset) {
         flags[i] = SKIP_RESTART;
         *reset = 1;
      } else if (p->locking && numLocked > 0 && practConvCheck >= 0 && !s->B) {      /* (the projector of the practical test would need B Q: not taken with a mass matrix) */
         if (givenR && bn < attainableTol) toProject[numToProject++] = i - left;
         else if (flags[i] != PRACT_CONV) flags[i] = UNCONV;
      } else {
         flags[i] = UNCONV;
      }
   }

   if (numToProject > 0) {
      /* practical convergence: || (I - QQ') r || <= tol with Q the locked vectors
       * (reference convergence.c:238-281; R loses its Q components as a side effect) */
      double *norms = (double *)malloc((size_t)numToProject * sizeof(double));
      int rc = pa_project_once(s, s->evecs, s->ldevecs, p->numOrthoConst + numLocked, R, ldR,
            toProject, numToProject, norms);
      if (rc) { free(norms); free(toProject); return rc; }
      for (int i = 0; i < numToProject; i++) {
         blockNorms[toProject[i]] = norms[i];
         flags[left + toProject[i]] = (norms[i] <= tol) ? PRACT_CONV : UNCONV;
      }
      free(norms);
   }
   free(toProject);
   return 0;
}

/* p[i] = column of Vp (m x nV) closest in angle to column i of Wn (m x n), i in [n0,n) */
void pa_map_vecs(const HS *Vp, int mrows, int nV, int ldV, const HS *Wn, int n0, int n,
      int ldW, int *pm) {
   double *vn = (double *)malloc((size_t)(nV > 0 ? nV : 1) * sizeof(double));
   HS *ip = (HS *)malloc((size_t)(nV > 0 ? nV : 1) * sizeof(HS));
   for (int j = 0; j < nV; j++) {
      double t = 0;
      for (int r = 0; r < mrows; r++) t += HS_ABS2(Vp[r + (size_t)j * ldV]);
      vn[j] = sqrt(t);
   }
   for (int i = n0; i < n; i++) {
      for (int j = 0; j < nV; j++) {
         HS t = 0;
         for (int r = 0; r < mrows; r++) t += HS_CONJ(Vp[r + (size_t)j * ldV]) * Wn[r + (size_t)i * ldW];
         ip[j] = t;
      }
      int jmax = -1;
      double ipmax = -1;
      for (int j = 0; j < nV; j++) {
         double ipij = HS_ABS(ip[j]);
         if (ipij > ipmax * vn[j]) {
            int k;
            for (k = 0; k < i && pm[k] != j; k++) ;
            if (k < i) continue;
            ipmax = fabs(ipij / vn[j]);
            jmax = j;
         }
      }
      if (jmax < 0) jmax = i;
      pm[i] = jmax;
   }
   free(vn);
   free(ip);
}

#if PA_IS_COMPLEX
void pa_monitor(pa_solver *s, double *basisEvals, int basisSize, int *basisFlags, int *iblock,
      int blockSize, double *basisNorms, int numConverged, double *lockedEvals, int numLocked,
      int *lockedFlags, double *lockedNorms, primme_event event);
#else
void pa_monitor(pa_solver *s, double *basisEvals, int basisSize, int *basisFlags, int *iblock,
      int blockSize, double *basisNorms, int numConverged, double *lockedEvals, int numLocked,
      int *lockedFlags, double *lockedNorms, primme_event event) {
   primme_params *p = s->p;
   p->stats.elapsedTime = pa_wtime() - s->startTime;
   if (!p->monitorFun) {
      /* the reference's default report (primme_c.c:602-700), same lines so that logs stay comparable */
      if (!p->outputFile || p->procID != 0 || p->printLevel < 2) return;
      const int found = p->locking ? numLocked : numConverged;
      if (event == primme_event_outer_iteration && p->printLevel >= 3) {
         for (int i = 0; i < blockSize; i++)
            fprintf(p->outputFile, "OUT %lld conv %d blk %d MV %lld Sec %E EV %13E |r| %.3E\n",
                  (long long)p->stats.numOuterIterations, found, i, (long long)p->stats.numMatvecs, p->stats.elapsedTime,
                  basisEvals[iblock[i]], basisNorms[iblock[i]]);
      } else if (event == primme_event_converged && ((!p->locking && p->printLevel >= 2) || (p->locking && p->printLevel >= 5))) {
         fprintf(p->outputFile, "#Converged %d eval[ %d ]= %13E norm %e Mvecs %lld Time %g\n", numConverged, iblock[0],
               basisEvals[iblock[0]], basisNorms[iblock[0]], (long long)p->stats.numMatvecs, p->stats.elapsedTime);
      } else if (event == primme_event_locked) {
         fprintf(p->outputFile, "Lock epair[ %d ]= %13E norm %.4e Mvecs %lld Time %.4e Flag %d\n", numLocked,
               lockedEvals[numLocked - 1], lockedNorms[numLocked - 1], (long long)p->stats.numMatvecs, p->stats.elapsedTime,
               lockedFlags[numLocked - 1]);
      }
      return;
   }
   (void)pa_call_monitor(p, basisEvals, basisSize, basisFlags, iblock, blockSize, basisNorms, numConverged,
         lockedEvals, numLocked, lockedFlags, lockedNorms, event);
}

/* (the fused block-size-1 paths below are real-arithmetic code: not in the complex objects) */
/* the library's own operator can run the one-launch tail (scale + A t + t'At) on this solver's panels */
int pa_svds_can_fuse(const primme_params *primme);
int pa_svds_apply_scaled(primme_params *primme, hipk_ctx *ctx, const void *t, const double *norm2_dev, void *xout, void *y,
      double *dot_dev);
int pa_fuse_tail_eligible(const pa_solver *s) {
   const primme_params *p = s->p;
   if (s->nT < 1) return 0;                      /* (single columns: the leading dimension does not enter) */
   if (p->matrixMatvec == primme_amd_matvec)
      return p->matrix && primme_amd_operator_can_fuse((const primme_amd_operator *)p->matrix) &&
             hipk_csr_dtype(primme_amd_operator_matrix((primme_amd_operator *)p->matrix)) == s->dt;
   return pa_svds_can_fuse(p) && !s->parallel;      /* normal equations of the singular value front end, one rank */
}
/* xout = a t, y = A xout, dot = xout'y through the operator that made the tail eligible */
static int fused_apply(pa_solver *s, const void *t, const double *norm2_dev, void *xout, void *y, double *dot_dev) {
   primme_params *p = s->p;
   if (p->matrixMatvec == primme_amd_matvec)
      return primme_amd_operator_apply_scaled((primme_amd_operator *)p->matrix, s->ctx, t, norm2_dev, xout, y, dot_dev);
   return pa_svds_apply_scaled(p, s->ctx, t, norm2_dev, xout, y, dot_dev);
}

/* The pre-restart convergence check may run through the fused residual kernel and hand its overlaps to the
 * restart: block size 1, GD without preconditioner, Rayleigh-Ritz, in-stream reductions, the library's own
 * operator, and G = W'Q known for all basis vectors but the newest. */
int pa_restart_stash_eligible(const pa_solver *s, int basisSize, int nLk) {
   const primme_params *p = s->p;
   if (!s->fused_restart || !s->fuse_gd || p->maxBlockSize != 1 || !s->rst_y || s->nT < 3) return 0;
   if (!s->wtr_enabled || !s->spec2_enabled || s->Q || s->VtBV || s->phase_timing || (s->parallel && !s->dev_comm)) return 0;
   if (basisSize > HIPK_WTR_MAX_K || nLk > HIPK_WTR_MAX_K || basisSize > 32 || nLk > 32) return 0;
   if (nLk > 0 && !(s->wtq_L == nLk && s->wtq_rows >= basisSize - 1)) return 0;
   if (p->n <= (int64_t)p->maxBasisSize + nLk) return 0;      /* the practical-convergence test would read R */
   if (p->maxMatvecs > 0 && p->stats.numMatvecs + 2 >= p->maxMatvecs) return 0;   /* the tail applies the operator ahead of time */
   return 1;      /* any operator: with an application callback the tail is project / scale / callback / t'At (eigs_conv.c) */
}

/* Speculative restart (eigs_solver.h): at the full-basis check of candidate `col`, predict the restart's
 * coefficient block by a dry run and, if the candidate leads it, run the restart pass NOW, out of place into
 * V2 / W2: it forms the residual (scratch column T(:,2)), its norm and its inner products with the new basis in
 * the same pass over V and W that the restart would make anyway; Q'r and W(:,k-1)'Q take one panel product with
 * two right-hand sides.  Returns 1 (done, *norm2 set), 0 (not applicable: the caller runs the fused residual
 * pass on the old basis) or a negative error. */
int pa_restart_plan(pa_solver *s, int basisSize, const int *flags_in, const int *iev_in, int nblock, int numLocked,
      int nprevhVecs, const int *map, double *evals, double *resNorms);
static int try_speculative_restart(pa_solver *s, int basisSize, int nLk, const int *flags, const int *iev, int nblock,
      int numLocked, int nprevhVecs, const int *map, int col, double *evals, double *resNorms, double *norm2) {
   const int K = s->K, ldh = basisSize;
   s->pl_launched = 0;
   if (!s->V2 || !s->plan_allowed || 64 + 2 * nLk > s->red_cap) return 0;
   if (pa_restart_plan(s, basisSize, flags, iev, nblock, numLocked, nprevhVecs, map, evals, resNorms)) return 0;
   const int rs = s->pl_rs;
   const int cc = s->pl_cand, ncv = s->pl_nc;
   if (s->pl_L != nLk || rs < 1 || 8 + 2 * rs + 1 > 64 || ncv > 30 || s->h_theta2[cc] != s->hVals[col] ||
         memcmp(s->h_coef2 + (size_t)cc * K, s->hVecs + (size_t)col * ldh, (size_t)basisSize * sizeof(double)))
      return 0;                                  /* the candidate is not what the restart would continue with */
   /* one copy launch for the coefficient block and the Ritz values behind it (at most K*K + K doubles: K <= 32 here);
    * PRIMME_AMD_RESTART_TWO_COPIES=1: the two launches of round 5 (A/B knob) */
   static int two_copies = -1;
   if (two_copies < 0) two_copies = getenv("PRIMME_AMD_RESTART_TWO_COPIES") != NULL;
   if (two_copies) {
      CHK(hipk_h2d(s->ctx, s->d_coef2, s->h_coef2, (size_t)K * (rs + 1) * sizeof(double)));
      CHK(hipk_h2d(s->ctx, s->d_theta2, s->h_theta2, (size_t)basisSize * sizeof(double)));
   } else CHK(hipk_h2d(s->ctx, s->d_coef2, s->h_coef2, ((size_t)K * K + basisSize) * sizeof(double)));
   hipk_job jobs[2 * 32 + 32 + 2];                /* rs <= 27 (twice), ncv <= 30, the unit column, the residual */
   int nj = 0;
   for (int c = 0; c < rs; c++) jobs[nj++] = (hipk_job){HIPK_JOB_XV, c, PCOL(s, s->V2, s->ld, c), -1};
   for (int c = 0; c < ncv; c++)                                        /* soft locking: converged Ritz vectors out to evecs */
      jobs[nj++] = (hipk_job){HIPK_JOB_XV, c, ECOL(s, s->p->numOrthoConst + c), -1};
   for (int c = 0; c < rs; c++) jobs[nj++] = (hipk_job){HIPK_JOB_XW, c, PCOL(s, s->W2, s->ld, c), -1};
   jobs[nj++] = (hipk_job){HIPK_JOB_XW, rs, TCOL(s, 3), -1};            /* the unit column: W(:,k-1) next to r */
   jobs[nj++] = (hipk_job){HIPK_JOB_RES, cc, TCOL(s, 2), -1};           /* its squared norm is part of the overlaps */
   CHK(hipk_ritz_update_overlaps(s->ctx, s->dt, s->m, s->V, s->W, s->ld, basisSize, s->d_coef2, K, s->d_theta2, jobs, nj,
         NULL, rs, NULL, 0, 0, s->d_red + 8));
   if (nLk > 0) {
      hipk_seg seg = {s->evecs, s->ldevecs, nLk};
      CHK(pa_reduce(s, s->d_red + 8, 2 * rs + 1, 0, 1));
      CHK(hipk_panel_dots(s->ctx, s->dt, s->m, &seg, 1, TCOL(s, 2), s->ld, 2, s->d_red + 64, nLk));
      CHK(pa_reduce(s, s->d_red + 64, 2 * nLk, 0, 0));
   } else {
      CHK(pa_reduce(s, s->d_red + 8, 2 * rs + 1, 0, 0));
   }
   const double *o = s->h_red + 8, *q = s->h_red + 64;
   double *c = s->rst_c;                          /* [V_new'r (rs) | Q'r (L) | r'r | W_new'r (rs)] */
   for (int j = 0; j < rs; j++) { c[j] = o[j]; c[rs + nLk + 1 + j] = o[rs + 1 + j]; }
   for (int l = 0; l < nLk; l++) { c[rs + l] = q[l]; s->rst_grow[l] = q[nLk + l]; }
   c[rs + nLk] = o[rs];
   *norm2 = o[rs];
   s->pl_launched = 1;
   return 1;
}

/* ---- the NEXT iteration enqueued before the host has seen this one -----------------------------
 * Where t'At of the fused tail goes: a slot at the end of the overlap buffer the pass belongs to (the pre-enqueued
 * iteration must not write where the host is still reading this iteration's). */
#define PA_ALPHA_OFF(s) ((s)->red_cap - 2)

/* diagnostics of the last solve of this process: iterations enqueued ahead of the host / adopted (include/primme_amd.h) */
long pa_last_pre[2];
void primme_amd_prelaunch_stats(long *launched, long *adopted) {
   if (launched) *launched = pa_last_pre[0];
   if (adopted) *adopted = pa_last_pre[1];
}

/* The host-side arithmetic after the one wait of the fused tail: the new column of H from W'r. */
static void tail_finish(pa_solver *s, int basisSize, int nLk, int nfov, const double *h_fov, double alpha) {
   const int nov = basisSize + nLk;
   const double *cV = h_fov, *cQ = h_fov + basisSize, *wr = h_fov + nov + 1;
   const double inv = 1.0 / sqrt(h_fov[nfov]);
   if (nLk > 0 && s->wtq_rows == basisSize - 1) {
      for (int l = 0; l < nLk; l++) s->wtq[(basisSize - 1) + (size_t)l * s->K] = wr[basisSize + l];
      s->wtq_rows = basisSize;
   }
   for (int j = 0; j < basisSize; j++) {
      double hc = 0.0;
      for (int i = 0; i < basisSize; i++)
         hc += s->H[(i < j ? i : j) + (size_t)(i < j ? j : i) * s->K] * cV[i];
      for (int l = 0; l < nLk; l++) hc += s->wtq[j + (size_t)l * s->K] * cQ[l];
      s->spec_hcol[j] = (wr[j] - hc) * inv;
   }
   s->spec_hcol[basisSize] = alpha;
}

/* Iteration basisSize + 1 enqueued behind the launches of iteration basisSize, whose results the host has NOT seen yet: the
 * Ritz pair of its residual comes from the one-wave Rayleigh-Ritz step (hipk_rr_arrow's arithmetic: the arrowhead eigenproblem in
 * the Ritz basis the host holds for the current size), then the same launches as any iteration — residual + overlaps,
 * Gram-Schmidt update, scale + A t + t'At — on the other overlap buffer / scratch column.  Nothing here synchronises.
 *
 * Round 6: two halves.  pa_prelaunch_prepare decides whether the case is covered and fills the step's input; the caller hands
 * it to hipk_tail_finish, the ONE small launch that ends the CURRENT iteration's tail (second stages of |t|^2 and t'At,
 * publication, and the step for the next iteration — until round 5 three second-stage launches and a launch for the step);
 * pa_prelaunch_enqueue then puts the next iteration's streaming launches behind it and leaves ITS tail unfinished
 * (s->pre_tail_deferred): whoever adopts that iteration finishes it, at the point where the decomposition of the iteration
 * after it is known. */
static int pa_prelaunch_prepare(pa_solver *s, int basisSize, int nLk, int col, int nfov, hipk_rr_in *in) {
   primme_params *p = s->p;
   const int k = basisSize, k1 = basisSize + 1;
   (void)nfov;
   /* row-partitioned runs: only where every reduction of the pass is exchanged inside the second stage of the launch that forms
    * it (peer-to-peer transport, hipk_xreduce_arm) — the global sums are then in HBM for the next launch without the host; every
    * rank takes the same decisions from the same bits, so the ranks enqueue (and throw away) the same launches */
   const int xr = s->parallel && s->dev_comm && hipk_xreduce_available(s->ctx);
   if (!s->pre_enabled || (s->parallel && !xr) || s->phase_timing || s->Q || s->VtBV || !s->wtr_enabled) return 0;
   if (p->target != primme_smallest && p->target != primme_largest) return 0;
   if (k1 > 16 || nLk > 10 || k1 + 1 > p->maxBasisSize || col < 0 || col > k) return 0;
   if (p->maxMatvecs > 0 && p->stats.numMatvecs + 3 >= p->maxMatvecs) return 0;
   if (p->maxOuterIterations > 0 && p->stats.numOuterIterations + 2 >= p->maxOuterIterations) return 0;
   if (nLk > 0 && !(s->wtq_L == nLk && s->wtq_rows >= k - 1)) return 0;
   const int nov1 = k1 + nLk, nfov1 = 2 * nov1 + 1;
   if (nfov1 + 8 > PA_ALPHA_OFF(s)) return 0;
   if (1 - s->spec_tcol >= s->nT) return 0;
   memset(in, 0, sizeof(*in));
   in->k = k; in->L = nLk; in->cand = col; in->largest = (p->target == primme_largest);
   in->grow_row = (nLk > 0 && s->wtq_rows == k - 1);
   for (int i = 0; i < k; i++) {
      in->theta[i] = s->hVals[i];
      for (int r = 0; r < k; r++) in->Y[r + i * k] = s->hVecs[r + (size_t)i * k];
   }
   for (int l = 0; l < nLk; l++)
      for (int j = 0; j < k; j++) in->G[j + l * k] = s->wtq[j + (size_t)l * s->K];
   return 1;
}

/* the streaming launches of iteration basisSize + 1; the Rayleigh-Ritz step has been enqueued (its pair will be in d_hnext).
 * Returns 0 when enqueued (s->pre_valid set) or when a row-partitioned pass turned out unusable (s->pre_valid clear). */
static int pa_prelaunch_enqueue(pa_solver *s, int basisSize, int nLk, int col, int rr_flagged) {
   primme_params *p = s->p;
   const int k1 = basisSize + 1;
   s->pre_valid = 0;
   const int xr = s->parallel && s->dev_comm && hipk_xreduce_available(s->ctx);
   const int nov1 = k1 + nLk, nfov1 = 2 * nov1 + 1;
   const int tcol = 1 - s->spec_tcol;
   char *dst1 = VCOL(s, k1);
   if (xr) hipk_xreduce_arm(s->ctx);
   /* (rr_flagged: the host looks at these overlaps only after the flag of the hipk_tail_finish that ends this iteration, so
    * the second stage of the pass need not publish one of its own) */
   if (rr_flagged && s->pre_quiet_fin) hipk_skip_next_flag(s->ctx);
   CHK(hipk_ritz_residual_overlaps_dev(s->ctx, s->dt, s->m, s->V, s->W, s->ld, k1, s->d_hnext, dst1, s->evecs, s->ldevecs, nLk, 1, s->d_fov_alt));
   /* the pair's pinned copy may be looked at once a flagged launch behind the Rayleigh-Ritz step is through: the small launch
    * that carried the step itself (rr_flagged: the caller has its sequence number), else the second stage of this pass */
   if (!rr_flagged) s->pre_seq_rr = hipk_seq_issued(s->ctx);
   /* (a second stage that did not take the arm — sums that are not global — means this pass cannot be used: it is left to be
    * overwritten, on every rank alike) */
   if (xr && !hipk_xreduce_covered(s->ctx, s->d_fov_alt, nfov1)) return 0;
   hipk_seg segs[2] = {{s->V, s->ld, k1}, {s->evecs, s->ldevecs, nLk}};
   /* the tail without second-stage launches of its own (hipk_tail_defer): with the library's own operator, whose one-launch
    * form adds the partial sums of |t|^2 itself (one rank: row-partitioned runs need the global |t|^2 first) and leaves the
    * second stage of t'At to hipk_tail_finish */
   /* (one rank only: on the mailboxes the deferred second stage would also defer its exchange to a launch the host issues late —
    * measured with the rows over 2 / 4 processes: 221 / 233 us per iteration against 198 / 211 with the second stages where
    * they were, profiles/r06_ranks_sharing_one_gpu.txt) */
   const int acc = (p->matrixMatvec == primme_amd_matvec && !xr) ? hipk_tail_defer(s->ctx, HIPK_TAIL_NORM | HIPK_TAIL_DOT) : 0;
   if (xr) hipk_xreduce_arm(s->ctx);
   CHK(hipk_panel_project_to(s->ctx, s->dt, s->m, segs, 2, s->d_fov_alt, nov1, dst1, s->ld, TCOL(s, tcol), s->ld, 1, s->d_fov_alt + nfov1));
   if (xr && !hipk_xreduce_covered(s->ctx, s->d_fov_alt + nfov1, 1)) { hipk_tail_abandon(s->ctx); return 0; }
   {
      if (xr && !(acc & HIPK_TAIL_DOT)) hipk_xreduce_arm(s->ctx);
      int rc = (p->matrixMatvec == primme_amd_matvec)
             ? primme_amd_operator_apply_scaled((primme_amd_operator *)p->matrix, s->ctx, TCOL(s, tcol), s->d_fov_alt + nfov1, dst1, WCOL(s, k1), s->d_fov_alt + PA_ALPHA_OFF(s))
             : pa_svds_apply_scaled(p, s->ctx, TCOL(s, tcol), s->d_fov_alt + nfov1, dst1, WCOL(s, k1), s->d_fov_alt + PA_ALPHA_OFF(s));
      if (rc) { hipk_tail_abandon(s->ctx); return rc < 0 ? rc : PRIMME_USER_FAILURE; }
   }
   s->pre_tail_deferred = (hipk_tail_pending(s->ctx) & HIPK_TAIL_DOT) != 0;
   s->pre_seq_end = hipk_seq_issued(s->ctx);       /* (tail not deferred: the flag of its last second stage) */
   if (!s->pre_tail_deferred && xr && !hipk_xreduce_covered(s->ctx, s->d_fov_alt + PA_ALPHA_OFF(s), 1)) return 0;
   s->pre_valid = 1; s->pre_k = k1; s->pre_L = nLk; s->pre_cand = col; s->pre_nfov = nfov1; s->pre_tcol = tcol;
   s->pre_launched++;
   return 0;
}

/* Is the pair the pre-enqueued pass was formed with the one the host's own Rayleigh-Ritz solve arrived at?  It must be an
 * eigenpair of the projected matrix to rounding (its residual IN the projected problem), with the host's Ritz value.  The
 * vectors themselves may differ by eps |H| / gap (two backward-stable solvers); the residual A V h - theta V h they produce
 * then differs by eps |A| (the difference lies along Ritz vectors whose values are as close as the gap). */
static int pre_pair_matches(const pa_solver *s, int k1, int col) {
   const double *h = s->h_hnext;
   if (h[33] != 0.0) return 0;
   double scale = 0.0, nh = 0.0, res = 0.0;
   for (int i = 0; i < k1; i++) { scale = PA_MAX(scale, fabs(s->hVals[i])); nh += h[i] * h[i]; }
   if (!(scale > 0.0) || !(fabs(nh - 1.0) <= 1e-12) || !(fabs(h[32] - s->hVals[col]) <= 1e-12 * scale)) return 0;
   for (int i = 0; i < k1; i++) {
      double t = -h[32] * h[i];
      for (int j = 0; j < k1; j++) t += s->H[(i < j ? i : j) + (size_t)(i < j ? j : i) * s->K] * h[j];
      res = PA_MAX(res, fabs(t));
   }
   if (!(res <= 1e-13 * scale)) return 0;
   /* ... and it must be the host's VECTOR as well (up to its sign): with (nearly) repeated Ritz values two eigenpairs of H that
    * both pass the test above can differ by O(1) in the vector, and the host goes on to use ITS column — for the restart, for
    * locking, for the Ritz vector it returns — while residual and overlaps would be the device's (advisor, round 5).  Two
    * backward-stable solvers agree to eps |H| / gap; anything beyond 1e-6 is another vector of the cluster: not adopted. */
   const double *y = s->hVecs + (size_t)col * k1;
   double dp = 0.0, dm = 0.0;
   for (int i = 0; i < k1; i++) { dp = PA_MAX(dp, fabs(h[i] - y[i])); dm = PA_MAX(dm, fabs(h[i] + y[i])); }
   return PA_MIN(dp, dm) <= 1e-6;
}

/* The speculative tail of a block-size-1 GD iteration, enqueued right after the fused residual pass: the
 * first Gram-Schmidt update with the device-resident overlaps, then (speculate2) normalisation, operator
 * application and the new column of H, so that the host synchronises once.  `rsrc` holds the residual
 * (the basis slot `dstc` itself, or a scratch column after a restart: the update is out of place either
 * way); d_fov / h_fov hold [V'r | Q'r | r'r | W'r | ..] in `nfov` entries. */
int pa_speculative_tail(pa_solver *s, int basisSize, int nLk, const char *rsrc, char *dstc, int nfov, int wtr,
      int speculate2, int parallel_host, int col, int adopted) {
   primme_params *p = s->p;
   const int nov = basisSize + nLk;
   int rc = 0;
   {
   hipk_seg segs[2] = {{s->V, s->ld, basisSize}, {s->evecs, s->ldevecs, nLk}};
   /* with the fused tail the projected vector goes to the scratch column T(:,0): the operator
    * launch gathers from it while it writes the normalised vector into V(:,k) */
   const int fuse_tail = speculate2 && wtr && pa_fuse_tail_eligible(s);
   s->spec_fused = 0;
   if (adopted) {
      /* this iteration was enqueued ahead of time (pa_prelaunch_enqueue) and its pair is the host's: its streaming launches are
       * in the stream.  What goes in behind them NOW: the one small launch that finishes its tail — with the Rayleigh-Ritz step
       * of the NEXT iteration when that one can be enqueued too —, the next iteration's streaming launches, then the one wait
       * for this tail's completion flag and the usual host arithmetic. */
      const int xr = s->parallel && s->dev_comm && hipk_xreduce_available(s->ctx);
      unsigned long long seq_end = s->pre_seq_end;
      const int deferred = s->pre_tail_deferred;
      s->spec_tcol = s->pre_tcol;
      s->spec_fused = 1;
      s->pre_valid = 0; s->pre_tail_deferred = 0;
      s->pre_adopted++;
      hipk_rr_in in;
      const int go = pa_prelaunch_prepare(s, basisSize, nLk, col, nfov, &in);
      if (deferred || go) {
         if (xr && deferred) hipk_xreduce_arm(s->ctx);
         CHK(hipk_tail_finish(s->ctx, go ? &in : NULL, s->d_fov, nfov, s->d_fov + PA_ALPHA_OFF(s), s->d_hnext));
         if (deferred) {
            seq_end = hipk_seq_issued(s->ctx);
            if (xr && !hipk_xreduce_covered(s->ctx, s->d_fov + PA_ALPHA_OFF(s), 1)) return PRIMME_PARALLEL_FAILURE;
         }
      }
      s->pre_seq_rr = seq_end;
      if (go) CHK(pa_prelaunch_enqueue(s, basisSize, nLk, col, deferred));
      CHK(hipk_wait_seq(s->ctx, seq_end));
      if (s->parallel) {                             /* the three exchanges of the adopted pass (residual overlaps, |t|^2, t'At) */
         if (s->dev_comm && pa_comm_failed(p->commInfo)) return PRIMME_PARALLEL_FAILURE;
         p->stats.numGlobalSum += 3;
         p->stats.volumeGlobalSum += nfov + 2;
      }
      tail_finish(s, basisSize, nLk, nfov, s->h_fov, s->h_fov[PA_ALPHA_OFF(s)]);
      s->spec2_valid = 1; s->spec2_k = basisSize;
      s->fov_projected = 1;
      return 0;
   }
   s->spec_tcol = 0;
   /* peer-to-peer transport: every reduction of the tail is exchanged inside the second stage of the launch that
    * forms it (hipk_xreduce_arm), so the tail keeps its one-rank shape — |t|^2 stays on the device, the operator
    * launch normalises on the fly, no scaling launches — and costs no reduction launch at all */
   const int xr = s->parallel && s->dev_comm && hipk_xreduce_available(s->ctx);
   /* Row-partitioned runs on the library's communicator: |t|^2 and t'At travel in ONE all-reduce (see below) */
   const int merge_red = fuse_tail && s->parallel && s->dev_comm && !xr;
   /* the one-launch tail of the library's own operator without second-stage launches of its own (hipk_tail_defer) */
   const int acc = (fuse_tail && speculate2 && !merge_red && !xr && p->matrixMatvec == primme_amd_matvec)
                 ? hipk_tail_defer(s->ctx, HIPK_TAIL_NORM | HIPK_TAIL_DOT) : 0;
   if (xr) hipk_xreduce_arm(s->ctx);
   CHK(hipk_panel_project_to(s->ctx, s->dt, s->m, segs, 2, s->d_fov, nov > 0 ? nov : 1, rsrc, s->ld,
              fuse_tail ? TCOL(s, 0) : dstc, s->ld, 1, s->d_fov + nfov));
   /* Row-partitioned runs on the library's communicator: |t|^2 and t'At travel in ONE all-reduce.
    * The operator is applied to the un-normalised t (no scaling in the launch), both numbers are
    * reduced together, and V(:,k), W(:,k) are scaled afterwards with the value the host then has:
    * two all-reduces per outer iteration instead of three (each is latency, not bandwidth). */
   if (speculate2 && merge_red) {
      rc = fused_apply(s, TCOL(s, 0), NULL, dstc, WCOL(s, basisSize), s->d_fov + nfov + 1);
      if (rc) return rc < 0 ? rc : PRIMME_USER_FAILURE;
      s->spec_fused = 1;
      CHK(pa_reduce(s, s->d_fov + nfov, 2, 0, 0));             /* the one synchronisation */
      const double inv = 1.0 / sqrt(s->h_fov[nfov]);
      CHK(hipk_scale_cols(s->ctx, s->dt, s->m, dstc, s->ld, 1, &inv));
      CHK(hipk_scale_cols(s->ctx, s->dt, s->m, WCOL(s, basisSize), s->ld, 1, &inv));
      const double *cV = s->h_fov, *cQ = s->h_fov + basisSize, *wr = s->h_fov + nov + 1;
      if (nLk > 0 && s->wtq_rows == basisSize - 1) {
         for (int l = 0; l < nLk; l++) s->wtq[(basisSize - 1) + (size_t)l * s->K] = wr[basisSize + l];
         s->wtq_rows = basisSize;
      }
      for (int j = 0; j < basisSize; j++) {
         double hc = 0.0;
         for (int i = 0; i < basisSize; i++)
            hc += s->H[(i < j ? i : j) + (size_t)(i < j ? j : i) * s->K] * cV[i];
         for (int l = 0; l < nLk; l++) hc += s->wtq[j + (size_t)l * s->K] * cQ[l];
         s->spec_hcol[j] = (wr[j] - hc) * inv;
      }
      s->spec_hcol[basisSize] = s->h_fov[nfov + 1] * inv * inv;
      s->spec2_valid = 1; s->spec2_k = basisSize;
   } else if (speculate2) {
      CHK(pa_reduce(s, s->d_fov + nfov, 1, 1, 1));
      if (fuse_tail) {
         /* the library's own operator (one rank, or the peer-to-peer transport: the second stage of the launch exchanges t'At with
          * the other ranks): normalisation, A t and t'At in one launch, reading the un-normalised vector
          * from the scratch column and rebuilding V(:,k) on the way; t'At goes to the alpha slot of the overlap buffer.
          * Before the one wait the NEXT iteration is enqueued behind this one (pa_prelaunch_next), so the wait is for this
          * tail's own completion flag, not for the last launch of the stream. */
         if (xr && !(acc & HIPK_TAIL_DOT)) hipk_xreduce_arm(s->ctx);
         rc = fused_apply(s, TCOL(s, 0), s->d_fov + nfov, dstc, WCOL(s, basisSize), s->d_fov + PA_ALPHA_OFF(s));
         if (rc) { hipk_tail_abandon(s->ctx); return rc < 0 ? rc : PRIMME_USER_FAILURE; }
         s->spec_fused = 1;
         /* the one small launch that finishes this tail (+ the Rayleigh-Ritz step of the next iteration when that one is
          * enqueued behind it), or — nothing deferred — that step as a launch of its own behind the usual second stages */
         const int deferred = (hipk_tail_pending(s->ctx) & HIPK_TAIL_DOT) != 0;
         hipk_rr_in in;
         const int go = pa_prelaunch_prepare(s, basisSize, nLk, col, nfov, &in);
         if (deferred || go) {
            if (xr && deferred) hipk_xreduce_arm(s->ctx);
            CHK(hipk_tail_finish(s->ctx, go ? &in : NULL, s->d_fov, nfov, s->d_fov + PA_ALPHA_OFF(s), s->d_hnext));
         }
         const unsigned long long seq_end = hipk_seq_issued(s->ctx);
         s->pre_seq_rr = seq_end;
         s->pre_valid = 0; s->pre_tail_deferred = 0;
         if (go) CHK(pa_prelaunch_enqueue(s, basisSize, nLk, col, deferred));
         if (go) {                                   /* launches are queued behind the one whose flag is wanted */
            CHK(hipk_wait_seq(s->ctx, seq_end));
            if (xr) {
               (void)hipk_xreduce_covered(s->ctx, s->d_fov + PA_ALPHA_OFF(s), 1);
               if (pa_comm_failed(p->commInfo)) return PRIMME_PARALLEL_FAILURE;
               p->stats.numGlobalSum++;
               p->stats.volumeGlobalSum++;
            }
         } else CHK(pa_reduce(s, s->d_fov + PA_ALPHA_OFF(s), 1, 0, 0));                 /* the one synchronisation */
         tail_finish(s, basisSize, nLk, nfov, s->h_fov, s->h_fov[PA_ALPHA_OFF(s)]);
         s->spec2_valid = 1; s->spec2_k = basisSize;
         s->fov_projected = 1;
         return 0;
      } else {
         CHK(hipk_scale_cols_rsqrt_dev(s->ctx, s->dt, s->m, dstc, s->ld, 1, s->d_fov + nfov));
         int one = 1, ierr = 0;
         PRIMME_INT ldx = s->ld;
         p->matrixMatvec(dstc, &ldx, WCOL(s, basisSize), &ldx, &one, p, &ierr);
         if (ierr) return PRIMME_USER_FAILURE;
      }
      if (wtr) {
         if (xr) hipk_xreduce_arm(s->ctx);
         CHK(hipk_pair_dots(s->ctx, s->dt, s->m, dstc, s->ld, WCOL(s, basisSize), s->ld, 1, s->d_red));
         CHK(pa_reduce(s, s->d_red, 1, 0, 0));                 /* the one synchronisation */
         const double *cV = s->h_fov, *cQ = s->h_fov + basisSize, *wr = s->h_fov + nov + 1;
         const double inv = 1.0 / sqrt(s->h_fov[nfov]);
         if (nLk > 0 && s->wtq_rows == basisSize - 1) {
            for (int l = 0; l < nLk; l++) s->wtq[(basisSize - 1) + (size_t)l * s->K] = wr[basisSize + l];
            s->wtq_rows = basisSize;
         }
         for (int j = 0; j < basisSize; j++) {
            double hc = 0.0;
            for (int i = 0; i < basisSize; i++)
               hc += s->H[(i < j ? i : j) + (size_t)(i < j ? j : i) * s->K] * cV[i];
            for (int l = 0; l < nLk; l++) hc += s->wtq[j + (size_t)l * s->K] * cQ[l];
            s->spec_hcol[j] = (wr[j] - hc) * inv;
         }
         s->spec_hcol[basisSize] = s->h_red[0];
      } else {
         hipk_seg vseg = {s->V, s->ld, basisSize + 1};
         CHK(hipk_panel_dots(s->ctx, s->dt, s->m, &vseg, 1, WCOL(s, basisSize), s->ld, 1, s->d_red, basisSize + 1));
         CHK(pa_reduce(s, s->d_red, basisSize + 1, 0, 0));      /* the one synchronisation */
         memcpy(s->spec_hcol, s->h_red, (size_t)(basisSize + 1) * sizeof(double));
      }
      s->spec2_valid = 1; s->spec2_k = basisSize;
   } else {
      CHK(pa_reduce(s, s->d_fov + nfov, 1, 0, 0));
   }
   s->fov_projected = 1;
   }
   return 0;
}

#endif   /* !PA_IS_COMPLEX */

/* Put the first unconverged Ritz pairs in the block, computing X, R and the
 * residual norms for them; flag converged pairs on the way. */
int pa_prepare_vecs(pa_solver *s, int basisSize, int i0, int blockSize, int *arbitraryVecs, double smallestResNorm,
      const int *flags, int RRForAll);

int pa_prepare_candidates(pa_solver *s, int basisSize, char *X, char *R, int computeXR,
      int *flags, int remainedEvals, double *blockNorms, int blockNormsSize, int maxBlockSize,
      int numLocked, double *evals, double *resNorms, int *iev, int *blockSize,
      int *recentlyConverged, double *smallestResNorm, int numConverged, double *basisNorms,
      int *reset, int nprevhVecs, int practConvChecking, int *map) {
   primme_params *p = s->p;
   const int ldh = basisSize;
   int lasti = -1;
   *blockSize = 0;
   /* fused GD mode: the residual IS the correction, so it is produced directly in the V
    * slot of the new basis vector and the Ritz vector X is never materialised (the reference
    * writes X and R and then copies R over X, correction.c:378 with no preconditioner) */
   const int fused = !PA_IS_COMPLEX && s->fuse_gd && computeXR;
   if (fused) { R = X; X = NULL; }
   /* overlaps (and the speculative tail) carried over a restart stay valid if the one candidate the restart
    * left is still the block when this call returns */
   const int carried = s->fov_carry && s->fov_valid && fused && blockNormsSize == 1 && maxBlockSize >= 1 &&
                       basisSize == s->fov_k && R == s->fov_col;
   const int carried_iev = carried ? iev[0] : -1;
   int relaunched = 0;
   s->fov_carry = 0;
   if (!carried) {
      s->fov_valid = 0;   /* only overlaps computed in THIS call, for the candidate that stays, may be reused */
      s->spec2_valid = 0;
   }
   int *flagsBlock = (int *)malloc((size_t)(maxBlockSize > 0 ? maxBlockSize : 1) * sizeof(int));
   double *hValsBlock = (double *)malloc((size_t)(maxBlockSize > 0 ? maxBlockSize : 1) * sizeof(double));
   hipk_job *jobs = (hipk_job *)malloc((size_t)(2 * maxBlockSize + 2) * sizeof(hipk_job));
   if (!flagsBlock || !hValsBlock || !jobs) return PRIMME_MALLOC_FAILURE;
   int rc = 0;

   for (int i = 0; i < blockNormsSize; i++) hValsBlock[i] = s->hVals[iev[*blockSize + i]];
   if (blockNormsSize > 0) {
      *smallestResNorm = HUGE_VAL;
      for (int i = 0; i < blockNormsSize; i++) *smallestResNorm = PA_MIN(*smallestResNorm, blockNorms[i]);
   }
   /* pair each Ritz vector with the closest one of the previous iteration and
    * carry the flags over */
   pa_map_vecs(s->prevhVecs, basisSize, nprevhVecs, s->K, s->hVecs, 0, basisSize, ldh, map);
   pa_permute_ints(flags, basisSize, map);

   *recentlyConverged = 0;
   for (;;) {
      for (int i = *blockSize; i < *blockSize + blockNormsSize; i++) flagsBlock[i - *blockSize] = flags[iev[i]];
      rc = pa_check_convergence(s, X ? PCOL(s, X, s->ld, *blockSize) : NULL, s->ld, computeXR && !fused,
            R ? PCOL(s, R, s->ld, *blockSize) : NULL, s->ld, computeXR, numLocked, 0, blockNormsSize,
            flagsBlock, blockNorms ? &blockNorms[*blockSize] : NULL, hValsBlock, reset, practConvChecking);
      if (rc) goto out;

      int blki = *blockSize;
      for (int i = 0; i < blockNormsSize && *blockSize < maxBlockSize; i++, blki++) {
         flags[iev[blki]] = flagsBlock[i];
         basisNorms[iev[blki]] = blockNorms[blki];
         const double targetShift = p->targetShifts ? p->targetShifts[s->targetShiftIndex] : 0.0;
         if ((p->target == primme_closest_leq && s->hVals[iev[blki]] - blockNorms[blki] > targetShift) ||
               (p->target == primme_closest_geq && s->hVals[iev[blki]] + blockNorms[blki] < targetShift)) {
            /* value +- residual entirely outside the wanted side: ignore */
         } else if (flagsBlock[i] != UNCONV && *recentlyConverged < remainedEvals &&
               (iev[blki] < p->numEvals - numLocked || p->target == primme_closest_geq ||
                     p->target == primme_closest_leq)) {
            if (!p->locking) {
               evals[iev[blki]] = s->hVals[iev[blki]];
               resNorms[iev[blki]] = blockNorms[blki];
               if (flagsBlock[i] == CONV)
                  p->stats.maxConvTol = PA_MAX(p->stats.maxConvTol, blockNorms[blki]);
            }
            (*recentlyConverged)++;
            if (*blockSize == 0) *smallestResNorm = HUGE_VAL;
            maxBlockSize = PA_MIN(maxBlockSize, p->numEvals + 1 - (*recentlyConverged) - numConverged);
            pa_monitor(s, s->hVals, basisSize, flags, &iev[blki], 1, basisNorms,
                  numConverged + *recentlyConverged, NULL, 0, NULL, NULL, primme_event_converged);
         } else if (flagsBlock[i] == UNCONV) {
            if (*blockSize == 0) *smallestResNorm = HUGE_VAL;
            *smallestResNorm = PA_MIN(*smallestResNorm, blockNorms[blki]);
            blockNorms[*blockSize] = blockNorms[blki];
            iev[*blockSize] = iev[blki];
            if (computeXR && blki != *blockSize) {
               if (X && (rc = hipk_copy_cols(s->ctx, s->dt, s->m, PCOL(s, X, s->ld, blki), s->ld, PCOL(s, X, s->ld, *blockSize), s->ld, 1))) goto out;
               if ((rc = hipk_copy_cols(s->ctx, s->dt, s->m, PCOL(s, R, s->ld, blki), s->ld, PCOL(s, R, s->ld, *blockSize), s->ld, 1))) goto out;
            }
            (*blockSize)++;
         }
         lasti = iev[blki];
      }

      /* refined extraction: well conditioned coefficient vectors for the next candidates */
      if ((rc = pa_prepare_vecs(s, basisSize, lasti + 1, maxBlockSize - *blockSize, &s->numArbitraryVecs,
                 *smallestResNorm, flags, 1))) goto out;
      /* next candidates after the last visited pair */
      blki = *blockSize;
      for (int i = lasti + 1; i < basisSize && blki < maxBlockSize; i++)
         if (flags[i] == UNCONV) iev[blki++] = i;
      if (blki == *blockSize || *recentlyConverged >= remainedEvals) break;
      blockNormsSize = blki - *blockSize;
      for (int i = 0; i < blockNormsSize; i++) hValsBlock[i] = s->hVals[iev[*blockSize + i]];

      /* X = V h, R = W h - X theta, norms — one fused pass over V and W */
      const int nLk = p->numOrthoConst + numLocked;
      (void)nLk;
      relaunched = 1;
#if !PA_IS_COMPLEX
      if (fused && p->maxBlockSize == 1 && blockNormsSize == 1 && basisSize <= 32 && nLk <= 32) {
         /* block size 1, GD without preconditioner: the residual is the next basis vector, so
          * the same pass also delivers the first Gram-Schmidt pass' overlaps [V'r | Q'r | r'r] */
         const int col = iev[*blockSize];
         char *dstc = PCOL(s, R, s->ld, *blockSize);
         double t0 = pa_wtime();
         /* Speculation: candidates are almost never converged (10 of 3287 iterations in
          * config 2), so the first Gram-Schmidt update v -= [V Q]*overlaps and |v|^2 are
          * enqueued right away and the host synchronises ONCE for |r|, the overlaps and |v|^2.
          * If the pair turns out converged the slot is scratch anyway.  Not done when the
          * convergence test may still project R (practical-convergence path). */
         const int nov = basisSize + nLk;
         const int parallel_host = (s->parallel && !s->dev_comm);
         const int speculate = (practConvChecking < 0 || !p->locking || numLocked == 0);
         s->fov_projected = 0;
         s->spec2_valid = 0;
         /* second stage of the speculation: normalise with the norm still on the device, apply
          * the operator and project, so that the whole outer iteration costs ONE host
          * synchronisation.  Needs in-stream reductions (single rank or the RCCL communicator). */
         /* (not when the application could be the one beyond maxMatvecs: a discarded tail is not counted) */
         const int speculate2 = speculate && !parallel_host && !s->phase_timing && dstc == VCOL(s, basisSize) &&
                                basisSize + 1 <= p->maxBasisSize && s->spec2_enabled &&
                                (p->maxMatvecs <= 0 || p->stats.numMatvecs + 1 < p->maxMatvecs);
         /* Default (PRIMME_AMD_NO_WTR=1 switches it off).  With A symmetric and W = A V, the new
          * column of H = V'AV is W't for the new basis vector t = (r - [V Q] c) / |.|, i.e.
          * (W'r - H c_V - (W'Q) c_Q) / |.|: W'r comes out of the residual pass (W is in registers
          * there), H c_V is host arithmetic, only t'At needs the new W column: no separate pass over V
          * for the projection (update_projection.c:99-122).  G = W'Q (first order in the locked pairs' residual norms: W'Q = V'R_Q)
          * is kept on the host: recomputed after every restart (pa_refresh_wtq, one panel product),
          * its row for the newest basis vector comes out of this same pass (W(:,k-1) and Q are both
          * in registers).  DESIGN.md §4d. */
         /* G = W'Q must be known for every basis vector but the newest, whose row this pass delivers */
         const int wtr = speculate2 && s->wtr_enabled && !s->Q && basisSize <= HIPK_WTR_MAX_K && nLk <= HIPK_WTR_MAX_K &&
                         (nLk == 0 || (s->wtq_L == nLk && s->wtq_rows >= basisSize - 1));   /* (RR extraction only) */
         const int nfov = nov + 1 + (wtr ? basisSize + nLk : 0);      /* [V'r | Q'r | r'r | W'r | W(:,k-1)'Q] */
         /* row-partitioned, peer-to-peer transport: the second stage of this pass exchanges the overlaps with the
          * other ranks itself (ortho.c:249's all-reduce inside the launch that forms the local sums) */
         /* Was this very iteration enqueued ahead of time, behind the previous one (pa_prelaunch_next)?  Its pair came from
          * the one-wave kernel; it is adopted when the host's own solve arrived at the same pair, thrown away otherwise
          * (the launches below then run behind it in the stream and overwrite what it wrote). */
         int adopted = 0;
         if (s->pre_valid && s->pre_k == basisSize && s->pre_L == nLk && s->pre_cand == col && speculate && speculate2 && wtr &&
               s->pre_nfov == nfov && pa_fuse_tail_eligible(s) &&
               (!s->parallel || (s->dev_comm && hipk_xreduce_available(s->ctx)))) {
            if ((rc = hipk_wait_seq(s->ctx, s->pre_seq_rr))) goto out;
            if (pre_pair_matches(s, basisSize, col)) {
               double *td = s->d_fov, *th = s->h_fov;
               s->d_fov = s->d_fov_alt; s->h_fov = s->h_fov_alt; s->d_fov_alt = td; s->h_fov_alt = th;
               adopted = 1;
            }
         }
         if (!adopted) {
            pa_pre_discard(s);
            if (s->parallel && s->dev_comm) hipk_xreduce_arm(s->ctx);
            if ((rc = hipk_ritz_residual_overlaps(s->ctx, s->dt, s->m, s->V, s->W, s->ld, basisSize,
                       s->hVecs + (size_t)col * ldh, s->hVals[col], dstc, s->evecs, s->ldevecs, nLk, wtr, s->d_fov))) goto out;
         }
         if (speculate) {
            if (!adopted && (rc = pa_reduce(s, s->d_fov, nfov, 1, parallel_host ? 0 : 1))) goto out;
            if ((rc = pa_speculative_tail(s, basisSize, nLk, dstc, dstc, nfov, wtr, speculate2, parallel_host, col, adopted))) goto out;
         } else {
            if ((rc = pa_reduce(s, s->d_fov, nfov, 1, 0))) goto out;
         }
         blockNorms[*blockSize] = sqrt(s->h_fov[basisSize + nLk]);
         s->fov_valid = 1; s->fov_k = basisSize; s->fov_L = nLk; s->fov_col = dstc; s->fov_s1_off = nfov;
         p->stats.timeDense += pa_wtime() - t0;
         p->stats.flopsDense += (double)s->m * 2.0 * basisSize;
      } else if (!computeXR && blockNormsSize == 1 && pa_restart_stash_eligible(s, basisSize, nLk)) {
         /* Full basis, the one candidate is tested before the restart (its norm is all the caller wants):
          * the same fused pass as in every other iteration instead of a norm-only pass over V and W.  It
          * costs L more columns and leaves, besides |r|, the residual itself (scratch column T(:,2)) and
          * its overlaps with the OLD basis; if the pair is not converged the restart turns those into the
          * overlaps with the restarted basis by a k x restartSize host product and the first iteration
          * after the restart starts from them (eigs_restart.c). */
         const int col = iev[*blockSize];
         const int nov = basisSize + nLk, nfov = 2 * nov + 1;
         double t0 = pa_wtime();
         s->fov_valid = 0;
         s->rst_valid = 0;
         double n2spec = 0.0;
         const int spec = try_speculative_restart(s, basisSize, nLk, flags, iev, *blockSize + 1, numLocked, nprevhVecs, map, col, evals, resNorms, &n2spec);
         if (spec < 0) { rc = spec; goto out; }
         if (spec == 1) {
            blockNorms[*blockSize] = sqrt(n2spec);
            p->stats.timeDense += pa_wtime() - t0;
            p->stats.flopsDense += (double)s->m * 2.0 * basisSize * (s->pl_rs + 1);
         } else {
         if ((rc = hipk_ritz_residual_overlaps(s->ctx, s->dt, s->m, s->V, s->W, s->ld, basisSize,
                    s->hVecs + (size_t)col * ldh, s->hVals[col], TCOL(s, 2), s->evecs, s->ldevecs, nLk, 1, s->d_fov))) goto out;
         if ((rc = pa_reduce(s, s->d_fov, nfov, 0, 0))) goto out;
         blockNorms[*blockSize] = sqrt(s->h_fov[nov]);
         memcpy(s->rst_ov, s->h_fov, (size_t)nfov * sizeof(double));
         memcpy(s->rst_y, s->hVecs + (size_t)col * ldh, (size_t)basisSize * sizeof(double));
         s->rst_theta = s->hVals[col];
         s->rst_k = basisSize; s->rst_L = nLk; s->rst_valid = 1;
         p->stats.timeDense += pa_wtime() - t0;
         p->stats.flopsDense += (double)s->m * 2.0 * basisSize;
         }
      } else
#endif
      {
         s->fov_valid = 0;
         if ((rc = pa_push_coefficients(s, basisSize, ldh))) goto out;
         int nj = 0;
         for (int c = 0; c < blockNormsSize; c++) {
            const int col = iev[*blockSize + c];
            if (computeXR) {
               if (X) jobs[nj++] = (hipk_job){HIPK_JOB_XV, col, PCOL(s, X, s->ld, *blockSize + c), -1};
               jobs[nj++] = (hipk_job){HIPK_JOB_RES, col, PCOL(s, R, s->ld, *blockSize + c), c};
            } else {
               jobs[nj++] = (hipk_job){HIPK_JOB_RES, col, NULL, c};
            }
         }
         if ((rc = pa_ritz_update(s, basisSize, jobs, nj, &blockNorms[*blockSize], blockNormsSize,
                    (int64_t)2 * blockNormsSize))) goto out;
      }
      /* do not trust residual norms below the error already accumulated in V, W.
       * (the reference's loop bounds, main_iter.c:1686-1688, are kept literally) */
      for (int i = *blockSize; i < blockNormsSize; i++)
         blockNorms[i] = PA_MAX(blockNorms[i], p->stats.estimateResidualError);
   }
out:
   if (carried && !relaunched && !(rc == 0 && *blockSize == 1 && iev[0] == carried_iev && *recentlyConverged == 0)) {
      s->fov_valid = 0;
      s->spec2_valid = 0;
   }
   free(flagsBlock);
   free(hValsBlock);
   free(jobs);
   return rc;
}
