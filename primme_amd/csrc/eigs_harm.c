/* eigs_harm.c — harmonic Rayleigh-Ritz extraction (projectionParams.projection = primme_proj_harmonic,
 * SURVEY §8 row f4) for interior eigenvalues.
 *
 *   pa_update_Q          <- reference src/eigs/update_W.c:77-113     (A - tau I) V = Q R, column by column
 *   pa_update_QtV        <- update_projection on (Q, V), unsymmetric (main_iter.c:475-479, :826-830)
 *   pa_solve_H_harm      <- solve_projection.c:430-516               eig of R^-T (Q'V)^T, back-transform,
 *                                                                     orthonormalise, Rayleigh quotients
 *   pa_restart_harmonic  <- restart.c:2255-2326                      H <- h'H h, QR recomputed from scratch
 *
 * Q lives in HBM next to V and W (one more n x maxBasisSize panel); R, Q'V and the left vectors
 * hU are small host matrices.  Orthogonalisation of the new Q columns is the same classical
 * Gram-Schmidt with Daniel's test as for V, with the coefficients recorded in R.
 * Covered: orth = implicit_I (block size 1, double — the default for these targets).  The refined
 * extraction (primme_proj_refined) is not on the device path yet and returns -44.
 */
#include "eigs_solver.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

int pa_reduce(pa_solver *s, double *d_buf, int count, int keep_dev, int defer_sync);
int pa_solve_H_RR(pa_solver *s, const double *H, int ldH, const double *VtBV, int ldVtBV,
      double *hVecs, int ldhVecs, double *hVals, int n, int numConverged);
int pa_ortho_local_vec(double *x, int n, const double *Q, int ldQ, int nQ, const double *G,
      int ldG, double *R, int64_t iseed[4]);

/* CGS + Daniel reorthogonalisation of Q(:, b1..b2) against Q(:, 0..i-1), coefficients into R
 * (reference ortho.c:123-360 with R != NULL).  Stops at the first column that vanishes. */
static int ortho_Q(pa_solver *s, int b1, int b2, int *nQ_out) {
   primme_params *p = s->p;
   const double tol = sqrt(2.0) / 2.0;
   const int K = s->K;
   double t0 = pa_wtime();
   *nQ_out = b1;
   for (int i = b1; i <= b2; i++) {
      char *q = PCOL(s, s->Q, s->ld, i);
      double s0 = 0.0, s1 = 0.0;
      for (int j = 0; j <= i; j++) s->R[j + (size_t)i * K] = 0.0;
      int pass, ok = 0;
      for (pass = 0; pass < 3; pass++) {
         const int first = (pass == 0), nov = i, ndot = nov + (first ? 1 : 0);
         hipk_seg segs[3] = {{s->Q, s->ld, i}, {NULL, 0, 0}, {q, s->ld, first ? 1 : 0}};
         CHK(hipk_panel_dots(s->ctx, s->dt, s->m, segs, 3, q, s->ld, 1, s->d_red, ndot));
         CHK(pa_reduce(s, s->d_red, ndot, 1, 1));
         CHK(hipk_panel_project(s->ctx, s->dt, s->m, segs, 1, s->d_red, nov > 0 ? nov : 1, q, s->ld, 1, s->d_red + nov + 1));
         CHK(pa_reduce(s, s->d_red + nov + 1, 1, 0, 0));
         p->stats.numOrthoInnerProds += ndot + nov + 1;
         for (int j = 0; j < i; j++) s->R[j + (size_t)i * K] += s->h_red[j];
         if (first) s0 = sqrt(s->h_red[nov]);
         s1 = sqrt(s->h_red[nov + 1]);
         if (!isfinite(s0) || !isfinite(s1) || s1 <= s->mach_eps * s0) break;   /* rank deficient */
         if (s1 > tol * s0) { ok = 1; break; }
         s0 = s1;
      }
      if (!ok) break;
      const double inv = 1.0 / s1;
      CHK(hipk_scale_cols(s->ctx, s->dt, s->m, q, s->ld, 1, &inv));
      s->R[i + (size_t)i * K] = s1;
      *nQ_out = i + 1;
   }
   p->stats.timeOrtho += pa_wtime() - t0;
   return 0;
}

/* Q(:, col0:col0+bs) = W - tau V for the new columns, then orthonormalised against Q(:, 0:*nQ) */
int pa_update_Q(pa_solver *s, double tau, int col0, int bs, int *nQ) {
   if (bs <= 0 || !s->Q) return 0;
   double mt[64];
   for (int c0 = 0; c0 < bs; c0 += 64) {
      const int n = PA_MIN(64, bs - c0);
      for (int c = 0; c < n; c++) mt[c] = -tau;
      CHK(hipk_copy_cols(s->ctx, s->dt, s->m, WCOL(s, col0 + c0), s->ld, PCOL(s, s->Q, s->ld, col0 + c0), s->ld, n));
      CHK(hipk_axpy_cols(s->ctx, s->dt, s->m, mt, VCOL(s, col0 + c0), s->ld, PCOL(s, s->Q, s->ld, col0 + c0), s->ld, n));
   }
   CHK(ortho_Q(s, *nQ, *nQ + bs - 1, nQ));
   for (int j = 0; j < col0; j++)                 /* R stays upper triangular */
      for (int i = col0; i < col0 + bs; i++) s->R[i + (size_t)j * s->K] = 0.0;
   return 0;
}

/* QtV(0:n1, col0:n1) and QtV(col0:n1, 0:col0), n1 = col0 + bs */
int pa_update_QtV(pa_solver *s, int col0, int bs) {
   if (bs <= 0 || !s->QtV) return 0;
   const int n1 = col0 + bs, K = s->K;
   for (int c0 = 0; c0 < bs; c0 += 8) {            /* new columns: Q(:,0:n1)' V(:,new) */
      const int n = PA_MIN(8, bs - c0);
      hipk_seg sq = {s->Q, s->ld, n1};
      CHK(hipk_panel_dots(s->ctx, s->dt, s->m, &sq, 1, VCOL(s, col0 + c0), s->ld, n, s->d_red, n1));
      CHK(pa_reduce(s, s->d_red, n1 * n, 0, 0));
      for (int c = 0; c < n; c++)
         for (int i = 0; i < n1; i++) s->QtV[i + (size_t)(col0 + c0 + c) * K] = s->h_red[i + (size_t)c * n1];
   }
   for (int c0 = 0; c0 < col0; c0 += 8) {          /* new rows: Q(:,new)' V(:,0:col0) */
      const int n = PA_MIN(8, col0 - c0);
      hipk_seg sq = {PCOL(s, s->Q, s->ld, col0), s->ld, bs};
      CHK(hipk_panel_dots(s->ctx, s->dt, s->m, &sq, 1, VCOL(s, c0), s->ld, n, s->d_red, bs));
      CHK(pa_reduce(s, s->d_red, bs * n, 0, 0));
      for (int c = 0; c < n; c++)
         for (int i = 0; i < bs; i++) s->QtV[(col0 + i) + (size_t)(c0 + c) * K] = s->h_red[i + (size_t)c * bs];
   }
   return 0;
}

/* harmonic Ritz pairs of the current basis: hVecs (k x k, ld k), hVals (Rayleigh quotients) */
int pa_solve_H_harm(pa_solver *s, int k, const double *G, int ldG) {
   primme_params *p = s->p;
   if (k == 0) return 0;
   const int K = s->K;
   double *X = (double *)malloc(sizeof(double) * (size_t)k * k * 2);
   if (!X) return PRIMME_MALLOC_FAILURE;
   double *HY = X + (size_t)k * k;
   /* X = R^-T (Q'V)^T: forward substitution with R^T (R upper triangular) */
   for (int c = 0; c < k; c++) {
      for (int i = 0; i < k; i++) {
         double t = s->QtV[c + (size_t)i * K];                 /* ((Q'V)^T)(i, c) */
         for (int j = 0; j < i; j++) t -= s->R[j + (size_t)i * K] * X[j + (size_t)c * k];
         X[i + (size_t)c * k] = t / s->R[i + (size_t)i * K];
      }
   }
   /* eigenpairs of X ordered for the inverse problem around shift 0 */
   double zero = 0.0, *oldShifts = p->targetShifts;
   const primme_target oldTarget = p->target;
   p->targetShifts = &zero;
   p->target = (oldTarget == primme_closest_geq) ? primme_largest
             : (oldTarget == primme_closest_leq) ? primme_smallest : primme_largest_abs;
   int rc = pa_solve_H_RR(s, X, k, NULL, 0, s->hVecs, k, s->hVals, k, 0);
   p->targetShifts = oldShifts;
   p->target = oldTarget;
   if (rc) { free(X); return rc; }
   for (int c = 0; c < k; c++) memcpy(s->hU + (size_t)c * k, s->hVecs + (size_t)c * k, sizeof(double) * (size_t)k);
   /* hVecs = R^-1 hU, then orthonormal columns */
   for (int c = 0; c < k; c++) {
      double *y = s->hVecs + (size_t)c * k;
      for (int i = k - 1; i >= 0; i--) {
         double t = y[i];
         for (int j = i + 1; j < k; j++) t -= s->R[i + (size_t)j * K] * y[j];
         y[i] = t / s->R[i + (size_t)i * K];
      }
   }
   for (int c = 0; c < k; c++) {
      double r;
      rc = pa_ortho_local_vec(s->hVecs + (size_t)c * k, k, s->hVecs, k, c, G, ldG, &r, p->iseed);
      if (rc) { free(X); return rc; }
   }
   /* hVals_i = y_i' H y_i (H upper-stored symmetric) */
   for (int c = 0; c < k; c++) {
      const double *y = s->hVecs + (size_t)c * k;
      for (int i = 0; i < k; i++) {
         double t = 0.0;
         for (int j = 0; j < k; j++) t += (j >= i ? s->H[i + (size_t)j * K] : s->H[j + (size_t)i * K]) * y[j];
         HY[i + (size_t)c * k] = t;
      }
      double v = 0.0;
      for (int i = 0; i < k; i++) v += y[i] * HY[i + (size_t)c * k];
      s->hVals[c] = v;
   }
   free(X);
   return 0;
}

/* after V, W <- V h, W h: projected matrix, fresh QR for the (possibly new) target shift */
int pa_restart_harmonic(pa_solver *s, int ldh, int restartSize, int basisSize, int numConverged) {
   primme_params *p = s->p;
   const int K = s->K;
   if (p->orth == primme_orth_implicit_I) {
      double *blk = (double *)malloc(sizeof(double) * (size_t)(restartSize > 0 ? restartSize * restartSize : 1));
      if (!blk) return PRIMME_MALLOC_FAILURE;
      pa_submatrix(s->hVecs, restartSize, ldh, s->H, basisSize, K, blk, restartSize);
      for (int j = 0; j < restartSize; j++)
         for (int i = 0; i < restartSize; i++) s->H[i + (size_t)j * K] = blk[i + (size_t)j * restartSize];
      free(blk);
   }
   s->targetShiftIndex = PA_MIN(p->numTargetShifts - 1, numConverged);
   int nQ = 0;
   CHK(pa_update_Q(s, p->targetShifts[s->targetShiftIndex], 0, restartSize, &nQ));
   if (nQ != restartSize) return PRIMME_UNEXPECTED_FAILURE;      /* "Not supported deficient QR" */
   CHK(pa_update_QtV(s, 0, restartSize));
   return 0;   /* the caller solves the projected problem (pa_solve_H) */
}
