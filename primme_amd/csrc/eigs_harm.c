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
 * Covered: orth = implicit_I (block size 1, double — the default for these targets), for both the
 * harmonic and the refined extraction (second half of this file).
 *
 * Compiled twice (eigs_scalar.h): HS = double, and HS = double complex through eigs_harm_z.c.  In the complex
 * objects R, Q'V, hU and hVecsRot are complex, inner products come back as (re, im) pairs (Q^H x), the singular
 * values and Ritz values stay real.
 */
#include "eigs_solver.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

int pa_reduce(pa_solver *s, double *d_buf, int count, int keep_dev, int defer_sync);
int pa_solve_H_RR(pa_solver *s, const HS *H, int ldH, const HS *VtBV, int ldVtBV,
      HS *hVecs, int ldhVecs, double *hVals, int n, int numConverged);
int pa_ortho_local_vec(HS *x, int n, const HS *Q, int ldQ, int nQ, const HS *G,
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
         /* buffer: nov (+1) scalars of overlaps, then the REAL squared norm of the update (as in pa_ortho_cgs) */
         const int s1_off = SD * (nov + 1);
         CHK(pa_reduce(s, s->d_red, SD * ndot, 1, 1));
         CHK(hipk_panel_project(s->ctx, s->dt, s->m, segs, 1, s->d_red, nov > 0 ? nov : 1, q, s->ld, 1, s->d_red + s1_off));
         CHK(pa_reduce(s, s->d_red + s1_off, 1, 0, 0));
         p->stats.numOrthoInnerProds += ndot + nov + 1;
         for (int j = 0; j < i; j++) s->R[j + (size_t)i * K] += ((const HS *)s->h_red)[j];
         if (first) s0 = sqrt(HS_RE(((const HS *)s->h_red)[nov]));
         s1 = sqrt(s->h_red[s1_off]);
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
   HS mt[64];
   for (int c0 = 0; c0 < bs; c0 += 64) {
      const int n = PA_MIN(64, bs - c0);
      for (int c = 0; c < n; c++) mt[c] = -tau;
      CHK(hipk_copy_cols(s->ctx, s->dt, s->m, WCOL(s, col0 + c0), s->ld, PCOL(s, s->Q, s->ld, col0 + c0), s->ld, n));
      CHK(hipk_axpy_cols(s->ctx, s->dt, s->m, (const double *)mt, VCOL(s, col0 + c0), s->ld, PCOL(s, s->Q, s->ld, col0 + c0), s->ld, n));
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
      CHK(pa_reduce(s, s->d_red, SD * n1 * n, 0, 0));
      for (int c = 0; c < n; c++)
         for (int i = 0; i < n1; i++) s->QtV[i + (size_t)(col0 + c0 + c) * K] = ((const HS *)s->h_red)[i + (size_t)c * n1];
   }
   for (int c0 = 0; c0 < col0; c0 += 8) {          /* new rows: Q(:,new)' V(:,0:col0) */
      const int n = PA_MIN(8, col0 - c0);
      hipk_seg sq = {PCOL(s, s->Q, s->ld, col0), s->ld, bs};
      CHK(hipk_panel_dots(s->ctx, s->dt, s->m, &sq, 1, VCOL(s, c0), s->ld, n, s->d_red, bs));
      CHK(pa_reduce(s, s->d_red, SD * bs * n, 0, 0));
      for (int c = 0; c < n; c++)
         for (int i = 0; i < bs; i++) s->QtV[(col0 + i) + (size_t)(c0 + c) * K] = ((const HS *)s->h_red)[i + (size_t)c * bs];
   }
   return 0;
}

/* harmonic Ritz pairs of the current basis: hVecs (k x k, ld k), hVals (Rayleigh quotients) */
int pa_solve_H_harm(pa_solver *s, int k, const HS *G, int ldG) {
   primme_params *p = s->p;
   if (k == 0) return 0;
   const int K = s->K;
   HS *X = (HS *)malloc(sizeof(HS) * (size_t)k * k * 2);
   if (!X) return PRIMME_MALLOC_FAILURE;
   HS *HY = X + (size_t)k * k;
   /* X = R^-H (Q'V)^H: forward substitution with R^H (R upper triangular) */
   for (int c = 0; c < k; c++) {
      for (int i = 0; i < k; i++) {
         HS t = HS_CONJ(s->QtV[c + (size_t)i * K]);            /* ((Q'V)^H)(i, c) */
         for (int j = 0; j < i; j++) t -= HS_CONJ(s->R[j + (size_t)i * K]) * X[j + (size_t)c * k];
         X[i + (size_t)c * k] = t / HS_CONJ(s->R[i + (size_t)i * K]);
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
   for (int c = 0; c < k; c++) memcpy(s->hU + (size_t)c * k, s->hVecs + (size_t)c * k, sizeof(HS) * (size_t)k);
   /* hVecs = R^-1 hU, then orthonormal columns */
   for (int c = 0; c < k; c++) {
      HS *y = s->hVecs + (size_t)c * k;
      for (int i = k - 1; i >= 0; i--) {
         HS t = y[i];
         for (int j = i + 1; j < k; j++) t -= s->R[i + (size_t)j * K] * y[j];
         y[i] = t / s->R[i + (size_t)i * K];
      }
   }
   for (int c = 0; c < k; c++) {
      double r;
      rc = pa_ortho_local_vec(s->hVecs + (size_t)c * k, k, s->hVecs, k, c, G, ldG, &r, p->iseed);
      if (rc) { free(X); return rc; }
   }
   /* hVals_i = y_i' H y_i (H upper-stored Hermitian) */
   for (int c = 0; c < k; c++) {
      const HS *y = s->hVecs + (size_t)c * k;
      for (int i = 0; i < k; i++) {
         HS t = 0.0;
         for (int j = 0; j < k; j++) t += (j >= i ? s->H[i + (size_t)j * K] : HS_CONJ(s->H[j + (size_t)i * K])) * y[j];
         HY[i + (size_t)c * k] = t;
      }
      double v = 0.0;
      for (int i = 0; i < k; i++) v += HS_RE(HS_CONJ(y[i]) * HY[i + (size_t)c * k]);
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
      HS *blk = (HS *)malloc(sizeof(HS) * (size_t)(restartSize > 0 ? restartSize * restartSize : 1));
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

/* =============================== refined extraction =========================================
 *   pa_solve_H_ref       <- solve_projection.c:541-626   SVD of R; right vectors are the coefficient
 *                                                          vectors, Ritz values by Rayleigh quotient
 *   pa_prepare_vecs      <- solve_projection.c:842-1008  Rayleigh-Ritz inside clusters of close
 *                                                          singular values ("arbitrary vectors")
 *   pa_restart_refined   <- restart.c:1837-2160          Q, R after V <- V h without re-factorising
 * The QR factorisation (A - tau I) V = Q R is the one of the harmonic path (pa_update_Q). */
int pa_svd(const HS *A, int ldA, int n, HS *U, int ldU, double *S, HS *V, int ldV);

int pa_solve_H_ref(pa_solver *s, int k, double *hVals_out) {
   primme_params *p = s->p;
   if (k == 0) return 0;
   const int K = s->K;
   CHK(pa_svd(s->R, K, k, s->hU, k, s->hSVals, s->hVecs, k));
   if (p->target == primme_closest_abs || p->target == primme_closest_leq || p->target == primme_closest_geq) {
      /* ascending singular values: the pairs closest to the shift first */
      for (int i = 0; i < k / 2; i++) {
         const int j = k - 1 - i;
         double ts = s->hSVals[i]; s->hSVals[i] = s->hSVals[j]; s->hSVals[j] = ts;
         for (int r = 0; r < k; r++) {
            HS t = s->hVecs[r + (size_t)i * k]; s->hVecs[r + (size_t)i * k] = s->hVecs[r + (size_t)j * k]; s->hVecs[r + (size_t)j * k] = t;
            t = s->hU[r + (size_t)i * k]; s->hU[r + (size_t)i * k] = s->hU[r + (size_t)j * k]; s->hU[r + (size_t)j * k] = t;
         }
      }
   }
   for (int c = 0; c < k; c++) {
      const HS *y = s->hVecs + (size_t)c * k;
      double v = 0.0;
      for (int i = 0; i < k; i++) {
         HS t = 0.0;
         for (int j = 0; j < k; j++) t += (j >= i ? s->H[i + (size_t)j * K] : HS_CONJ(s->H[j + (size_t)i * K])) * y[j];
         v += HS_RE(HS_CONJ(y[i]) * t);
      }
      hVals_out[c] = v;
   }
   return 0;
}

/* hVecs(:, j:i-1) <- Rayleigh-Ritz vectors of the span of a cluster of singular vectors */
int pa_prepare_vecs(pa_solver *s, int basisSize, int i0, int blockSize, int *arbitraryVecs, double smallestResNorm,
      const int *flags, int RRForAll) {
   primme_params *p = s->p;
   if (!s->refined || basisSize == 0 || blockSize == 0) return 0;
   const int K = s->K, ldh = basisSize;
   const double aNorm = (p->aNorm <= 0.0) ? p->stats.estimateLargestSVal : p->aNorm;
   double eps = p->stats.maxConvTol > 0.0 ? p->stats.maxConvTol : (smallestResNorm < HUGE_VAL ? smallestResNorm / 10.0 : 0.0);
   eps = PA_MAX(6.28 * s->mach_eps, eps);
   int candidates = 0, i = PA_MIN(*arbitraryVecs, basisSize), j = i0;
   while (j < basisSize && candidates < blockSize) {
      for (; j < i; j++) if (!flags || flags[j] == UNCONV) candidates++;
      if (candidates >= blockSize) break;
      int someCandidate = 0;
      double ip = 0.0;
      for (i = j + 1; i < basisSize; i++) {
         const double minDiff = sqrt(2.0) * s->hSVals[basisSize - 1] * PA_EPS / (aNorm * eps / fabs(s->hVals[i] - s->hVals[i - 1]));
         const double ip0 = HS_ABS(s->hVecs[(size_t)(i - 1) * ldh + basisSize - 1]);
         ip += ip0 * ip0;
         const double ip1 = (ip != 0.0) ? ip : HUGE_VAL;
         someCandidate = 1;
         if (fabs(s->hSVals[i] - s->hSVals[i - 1]) >= minDiff &&
               (smallestResNorm >= HUGE_VAL || sqrt(ip1) >= smallestResNorm / aNorm / 3.16))
            break;
      }
      i = PA_MIN(i, basisSize);
      if (i - j > 1 && (someCandidate || RRForAll)) {
         const int an = i - j;
         HS *aH = (HS *)malloc(sizeof(HS) * (size_t)basisSize * an);
         HS *ah = (HS *)calloc((size_t)an * an, sizeof(HS));
         double *av = (double *)malloc(sizeof(double) * (size_t)an);
         if (!aH || !ah || !av) { free(aH); free(ah); free(av); return PRIMME_MALLOC_FAILURE; }
         for (int c = *arbitraryVecs; c < i; c++) {           /* hVecsRot(:, arbitraryVecs:i-1) = I */
            for (int r = 0; r < K; r++) s->hVecsRot[r + (size_t)c * K] = 0.0;
            s->hVecsRot[c + (size_t)c * K] = 1.0;
         }
         pa_submatrix(s->hVecs + (size_t)j * ldh, an, ldh, s->H, basisSize, K, aH, an);
         /* ordered for the shift of the current factorisation (the reference passes targetShiftIndex here) */
         int rc = pa_solve_H_RR(s, aH, an, NULL, 0, ah, an, av, an, PA_MAX(s->targetShiftIndex, 0));
         if (rc) { free(aH); free(ah); free(av); return rc; }
         for (int c = 0; c < an; c++) {
            s->hVals[j + c] = av[c];
            for (int r = 0; r < an; r++) s->hVecsRot[(j + r) + (size_t)(j + c) * K] = ah[r + (size_t)c * an];
         }
         for (int c = 0; c < an; c++)                          /* hVecs(:, j:i-1) *= ahVecs */
            for (int r = 0; r < basisSize; r++) {
               HS t = 0.0;
               for (int q = 0; q < an; q++) t += s->hVecs[r + (size_t)(j + q) * ldh] * ah[q + (size_t)c * an];
               aH[r + (size_t)c * basisSize] = t;
            }
         for (int c = 0; c < an; c++) memcpy(s->hVecs + (size_t)(j + c) * ldh, aH + (size_t)c * basisSize, sizeof(HS) * (size_t)basisSize);
         free(aH); free(ah); free(av);
         s->coef_valid_k = -1;      /* the copy of hVecs in HBM is stale now */
         *arbitraryVecs = i;
      }
   }
   return 0;
}

/* orthonormalise columns b1..b2 of X (rows n, ld ldx) against the previous ones, coefficients into
 * R (K-strided): the host Bortho_local with R of the reference, column by column */
static int ortho_local_cols_R(pa_solver *s, HS *X, int n, int ldx, int b1, int b2, HS *R, int ldR) {
   for (int c = b1; c <= b2; c++) {
      HS *x = X + (size_t)c * ldx;
      /* coefficients: two classical passes recorded in R */
      for (int r = 0; r <= c; r++) R[r + (size_t)c * ldR] = 0.0;
      for (int pass = 0; pass < 3; pass++) {
         for (int q = 0; q < c; q++) {
            const HS *y = X + (size_t)q * ldx;
            HS t = 0.0;
            for (int i = 0; i < n; i++) t += HS_CONJ(y[i]) * x[i];
            for (int i = 0; i < n; i++) x[i] -= t * y[i];
            R[q + (size_t)c * ldR] += t;
         }
      }
      double nr = 0.0;
      for (int i = 0; i < n; i++) nr += HS_ABS2(x[i]);
      nr = sqrt(nr);
      R[c + (size_t)c * ldR] = nr;
      if (nr > 0.0) for (int i = 0; i < n; i++) x[i] /= nr;
      else {                      /* vanished: any unit vector orthogonal to the rest */
         double rr;
         int rc = pa_ortho_local_vec(x, n, X, ldx, c, NULL, 0, &rr, s->p->iseed);
         if (rc) return rc;
         R[c + (size_t)c * ldR] = 0.0;
      }
   }
   return 0;
}

int pa_restart_refined(pa_solver *s, int ldh, int restartSize, int basisSize, int numConverged, int numPrevRetained,
      int indexOfPreviousVecs, int indexOfPreviousVecsBeforeRestart, const int *restartPerm, const int *hVecsPerm,
      int *numArbitraryVecs) {
   primme_params *p = s->p;
   const int K = s->K;
   const double aNorm = PA_MAX(p->aNorm, p->stats.estimateLargestSVal);
   if (p->orth == primme_orth_implicit_I) {
      HS *blk = (HS *)malloc(sizeof(HS) * (size_t)(restartSize > 0 ? restartSize * restartSize : 1));
      if (!blk) return PRIMME_MALLOC_FAILURE;
      pa_submatrix(s->hVecs, restartSize, ldh, s->H, basisSize, K, blk, restartSize);
      for (int j = 0; j < restartSize; j++)
         for (int i = 0; i < restartSize; i++) s->H[i + (size_t)j * K] = blk[i + (size_t)j * restartSize];
      free(blk);
   }
   /* the target moved: rebuild the factorisation for the new shift */
   if (s->targetShiftIndex < 0 ||
         fabs(p->targetShifts[s->targetShiftIndex] - p->targetShifts[PA_MIN(p->numTargetShifts - 1, numConverged)]) >
               s->mach_eps * aNorm) {
      s->targetShiftIndex = PA_MIN(p->numTargetShifts - 1, numConverged);
      int nQ = 0;
      CHK(pa_update_Q(s, p->targetShifts[s->targetShiftIndex], 0, restartSize, &nQ));
      if (nQ != restartSize) return PRIMME_UNEXPECTED_FAILURE;
      CHK(pa_solve_H_ref(s, restartSize, s->hVals));
      *numArbitraryVecs = 0;
      return 0;
   }

   int *rp0 = (int *)malloc(sizeof(int) * (size_t)(restartSize > 0 ? restartSize : 1));
   if (!rp0) return PRIMME_MALLOC_FAILURE;
   for (int i = 0; i < restartSize; i++) rp0[i] = restartPerm[hVecsPerm[i]];
   int newArb = 0;
   for (int i = 0; i < restartSize - numPrevRetained; i++) if (rp0[i] < *numArbitraryVecs) newArb++;

   /* R * prevhVecs */
   HS *RPrev = (HS *)calloc((size_t)(numPrevRetained > 0 ? numPrevRetained : 1) * basisSize, sizeof(HS));
   const int nRegular = restartSize - numPrevRetained;
   int mRot = *numArbitraryVecs;
   for (int i = 0; i < nRegular; i++) mRot = PA_MAX(mRot, rp0[i] + 1);
   HS *Rot0 = (HS *)calloc((size_t)(mRot > 0 ? mRot : 1) * (nRegular > 0 ? nRegular : 1), sizeof(HS));
   HS *work = (HS *)malloc(sizeof(HS) * (size_t)basisSize * (restartSize > 0 ? restartSize : 1));
   if (!RPrev || !Rot0 || !work) { free(rp0); free(RPrev); free(Rot0); free(work); return PRIMME_MALLOC_FAILURE; }
   for (int c = 0; c < numPrevRetained; c++)
      for (int r = 0; r < basisSize; r++) {
         HS t = 0.0;
         for (int q = 0; q < basisSize; q++) t += s->R[r + (size_t)q * K] * s->hVecs[q + (size_t)(indexOfPreviousVecs + c) * ldh];
         RPrev[r + (size_t)c * basisSize] = t;
      }
   /* Rot0 = diag(hSVals) * hVecsRot(:, rp0(0:newArb-1)), then unit columns scaled by their singular value */
   for (int c = 0; c < newArb; c++)
      for (int r = 0; r < *numArbitraryVecs; r++) Rot0[r + (size_t)c * mRot] = s->hVecsRot[r + (size_t)rp0[c] * K] * s->hSVals[r];
   for (int c = newArb; c < nRegular; c++) Rot0[rp0[c] + (size_t)c * mRot] = s->hSVals[rp0[c]];
   for (int j = 0; j < K; j++) for (int i = 0; i < K; i++) s->R[i + (size_t)j * K] = 0.0;
   int rc = ortho_local_cols_R(s, Rot0, mRot, mRot, 0, nRegular - 1, s->R, K);
   /* hU = [hU * Rot0, R prev] */
   if (!rc) {
      for (int c = 0; c < nRegular; c++)
         for (int r = 0; r < basisSize; r++) {
            HS t = 0.0;
            for (int q = 0; q < mRot; q++) t += s->hU[r + (size_t)q * basisSize] * Rot0[q + (size_t)c * mRot];
            work[r + (size_t)c * basisSize] = t;
         }
      for (int c = 0; c < numPrevRetained; c++) memcpy(work + (size_t)(nRegular + c) * basisSize, RPrev + (size_t)c * basisSize, sizeof(HS) * (size_t)basisSize);
      memcpy(s->hU, work, sizeof(HS) * (size_t)basisSize * restartSize);
      rc = ortho_local_cols_R(s, s->hU, basisSize, basisSize, nRegular, nRegular + numPrevRetained - 1, s->R, K);
   }
   if (!rc) {
      for (int i = newArb; i < nRegular; i++)
         if (rp0[i] >= *numArbitraryVecs) {
            for (int j = 0; j <= i; j++) s->R[j + (size_t)i * K] = 0.0;
            s->R[i + (size_t)i * K] = s->hSVals[rp0[i]];
         }
      if (*numArbitraryVecs <= indexOfPreviousVecsBeforeRestart)
         for (int c = nRegular; c < restartSize; c++) for (int r = 0; r < nRegular; r++) s->R[r + (size_t)c * K] = 0.0;
      /* Q <- Q hU on the device */
      for (int c = 0; c < restartSize; c++) memcpy((HS *)s->h_coef + (size_t)c * K, s->hU + (size_t)c * basisSize, sizeof(HS) * (size_t)basisSize);
      rc = hipk_h2d(s->ctx, s->d_coef, s->h_coef, sizeof(HS) * (size_t)K * restartSize);
      s->coef_valid_k = -1;
      hipk_job *jobs = (hipk_job *)malloc(sizeof(hipk_job) * (size_t)(restartSize > 0 ? restartSize : 1));
      if (!jobs) rc = PRIMME_MALLOC_FAILURE;
      if (!rc) {
         for (int c = 0; c < restartSize; c++) jobs[c] = (hipk_job){HIPK_JOB_XV, c, PCOL(s, s->Q, s->ld, c), -1};
         /* one launch: the update is in place */
         rc = hipk_ritz_update(s->ctx, s->dt, s->m, s->Q, NULL, s->ld, basisSize, s->d_coef, K, NULL, jobs, restartSize, NULL);
         if (!rc) rc = hipk_sync(s->ctx);
      }
      free(jobs);
   }
   free(RPrev); free(Rot0); free(work);
   if (rc) { free(rp0); return rc; }

   /* singular triplets of the new R; the Ritz values only move with the permutation */
   double *dummy = (double *)malloc(sizeof(double) * (size_t)(restartSize > 0 ? restartSize : 1));
   if (!dummy) { free(rp0); return PRIMME_MALLOC_FAILURE; }
   rc = pa_solve_H_ref(s, restartSize, dummy);
   free(dummy);
   if (rc) { free(rp0); return rc; }
   pa_permute_reals(s->hVals, 1, restartSize, 1, hVecsPerm);
   int *inv = (int *)malloc(sizeof(int) * (size_t)(restartSize > 0 ? restartSize : 1));
   if (!inv) { free(rp0); return PRIMME_MALLOC_FAILURE; }
   for (int i = 0; i < restartSize; i++) inv[hVecsPerm[i]] = i;
   pa_permute_cols(s->R, restartSize, restartSize, K, inv);
   free(inv);

   if (*numArbitraryVecs <= indexOfPreviousVecsBeforeRestart) {
      *numArbitraryVecs = newArb;
      for (int i = newArb; i < restartSize; i++) if (hVecsPerm[i] != i) *numArbitraryVecs = i + 1;
   } else *numArbitraryVecs = restartSize;

   /* hVecsRot <- hVecs' for the arbitrary vectors, whose coefficient vectors become unit vectors */
   for (int j = 0; j < K; j++) for (int i = 0; i < K; i++) s->hVecsRot[i + (size_t)j * K] = 0.0;
   for (int j = 0; j < *numArbitraryVecs; j++)
      for (int i = 0; i < restartSize; i++) s->hVecsRot[i + (size_t)j * K] = HS_CONJ(s->hVecs[j + (size_t)i * restartSize]);
   for (int j = 0; j < *numArbitraryVecs; j++) {
      for (int i = 0; i < restartSize; i++) s->hVecs[i + (size_t)j * restartSize] = 0.0;
      s->hVecs[hVecsPerm[j] + (size_t)j * restartSize] = 1.0;
   }
   free(rp0);
   return 0;
}
