/* eigs_block.c — block orthogonalisation with a tracked Gram matrix (orth = explicit_I):
 * iterative Cholesky-QR with an eigendecomposition fall-back (SVQB), used when the block
 * size is > 1 or the working precision is single.
 *
 *   pa_ortho_block_gram   <- reference src/eigs/ortho.c:497-803  (Bortho_block_gen_Sprimme)
 *   device step           <- reference src/eigs/ortho.c:963-1072 (Num_ortho_kernel)
 *   decomposition         <- reference :1097-1137
 *   rank_estimation       <- reference :1165-1181
 *   pa_update_cholesky    <- reference :1199-1220
 *
 * Device work per sweep: X <- (X - [Q V]*A) * Y'  (hipk_panel_project + a right-multiply by a
 * b x b matrix through hipk_ritz_update, in place) and the Gram block [Q V X]'X
 * (hipk_panel_dots, b right-hand sides: the TN tall-skinny GEMM of the north star).  The small
 * factorisations (<= (L+K)^2) stay on the host like the reference's HSCALAR objects.
 */
#include "eigs_solver.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

int pa_reduce(pa_solver *s, double *d_buf, int count, int keep_dev, int defer_sync);

/* number of consecutive linearly independent columns judged from the Gram matrix G (upper) */
static int rank_estimation(const HS *G, int n0, int n1, int n, int ldG) {
   int i, j;
   for (i = n0; i < n1; i++) {
      const double Gii = HS_RE(G[i + (size_t)i * ldG]);
      if (!isfinite(Gii) || Gii <= 0.0) break;
      for (j = 0; j < i; j++)
         if (HS_ABS(G[j + (size_t)i * ldG]) > .8 / n * sqrt(Gii * HS_RE(G[j + (size_t)j * ldG]))) break;
      if (j < i) break;
   }
   return i;
}

/* Y = chol(C) (upper, Yortho = 0, D = 1) or, if C is not numerically SPD, the eigenvectors
 * with eigenvalues D in non-increasing order (Yortho = 1).  C upper, n x n, ld n. */
static int decomposition(const HS *Cm, int n, HS *Y, double *D, int *Yortho) {
   for (int j = 0; j < n; j++)
      for (int i = 0; i < n; i++) Y[i + (size_t)j * n] = (i <= j) ? Cm[i + (size_t)j * n] : 0.0;
   if (pa_potrf_upper(n, Y, n) == 0) {
      *Yortho = 0;
      for (int i = 0; i < n; i++) D[i] = 1.0;
      return 0;
   }
   HS *neg = (HS *)malloc((size_t)n * n * sizeof(HS));
   if (!neg) return PRIMME_MALLOC_FAILURE;
   for (int j = 0; j < n; j++)
      for (int i = 0; i < n; i++) neg[i + (size_t)j * n] = (i <= j) ? -Cm[i + (size_t)j * n] : 0.0;
   int rc = pa_sym_eig(n, neg, n, D, Y, n);   /* ascending of -C = descending of C */
   free(neg);
   if (rc) return rc;
   for (int i = 0; i < n; i++) D[i] = -D[i];
   *Yortho = 1;
   return 0;
}

/* fG(:, n0:n) <- Cholesky update given the new columns G(:, n0:n) */
int pa_update_cholesky(const HS *G, int ldG, HS *fG, int ldfG, int n0, int n) {
   if (n <= n0) return 0;
   const int nc = n - n0;
   HS *A = (HS *)malloc((size_t)n * nc * sizeof(HS));
   if (!A) return PRIMME_MALLOC_FAILURE;
   for (int c = 0; c < nc; c++)
      for (int i = 0; i < n; i++) A[i + (size_t)c * n] = (i <= n0 + c) ? G[i + (size_t)(n0 + c) * ldG] : 0.0;
   pa_trsm_left_upper_trans(n0, nc, fG, ldfG, A, n);
   for (int c = 0; c < nc; c++)
      for (int r = 0; r <= c; r++) {
         HS t = 0.0;
         for (int i = 0; i < n0; i++) t += HS_CONJ(A[i + (size_t)r * n]) * A[i + (size_t)c * n];
         A[n0 + r + (size_t)c * n] -= t;
      }
   /* Cholesky of the trailing block (upper part stored at rows n0.. of A) */
   HS *T = (HS *)malloc((size_t)nc * nc * sizeof(HS));
   for (int c = 0; c < nc; c++)
      for (int r = 0; r < nc; r++) T[r + (size_t)c * nc] = (r <= c) ? A[n0 + r + (size_t)c * n] : 0.0;
   (void)pa_potrf_upper(nc, T, nc);   /* like the reference, a failure shows up later in rank estimation */
   for (int c = 0; c < nc; c++)
      for (int r = 0; r < nc; r++) A[n0 + r + (size_t)c * n] = (r <= c) ? T[r + (size_t)c * nc] : 0.0;
   free(T);
   for (int c = 0; c < nc; c++)
      for (int i = 0; i < n; i++) fG[i + (size_t)(n0 + c) * ldfG] = A[i + (size_t)c * n];
   free(A);
   return 0;
}

/* X(:, 0:nX) <- X * M (M nX x nX on the host), in place, one pass over X */
static int right_multiply(pa_solver *s, char *X, int64_t ldX, int nX, const HS *M) {
   /* M travels tightly packed (leading dimension nX): the block can be wider than maxBasisSize
    * (numOrthoConst > maxBasisSize in init_basis), the coefficient buffers hold max(K, numOrthoConst)^2 */
   if ((size_t)nX * nX > s->coef_cap) return PRIMME_UNEXPECTED_FAILURE;
   memcpy(s->h_coef, M, (size_t)nX * nX * sizeof(HS));
   CHK(hipk_h2d(s->ctx, s->d_coef, s->h_coef, (size_t)nX * nX * sizeof(HS)));
   s->coef_valid_k = -1;
   hipk_job jobs[HIPK_MAX_JOBS];
   /* wider than one launch's job table: column chunks read the whole row first, so they would see
    * already-updated columns -> go through the scratch panel */
   if (nX > HIPK_MAX_JOBS) return PRIMME_FUNCTION_UNAVAILABLE;
   for (int c = 0; c < nX; c++) jobs[c] = (hipk_job){HIPK_JOB_XV, c, PCOL(s, X, ldX, c), -1};
   /* the panel X plays the role of "V" (k = nX columns); W is not touched */
   return hipk_ritz_update(s->ctx, s->dt, s->m, X, X, ldX, nX, s->d_coef, nX, s->d_theta, jobs, nX, NULL);
}

/* Orthonormalise Vp(:, b1..b2) against locked (numLocked columns), Vp(:, 0..b1) and among
 * themselves, maintaining s->VtBV / s->fVtBV (indexing: locked columns first, then Vp's). */
int pa_ortho_block_gram(pa_solver *s, char *Vp, int64_t ldV, int b1, int b2, char *locked,
      int64_t ldLocked, int numLocked, HS *RLocked, int ldRLocked, int maxRank, int *b2_out) {
   primme_params *p = s->p;
   HS *G = s->VtBV, *fG = s->fVtBV;
   const int ldG = s->ldVtBV;
   b2++;                                  /* exclusive upper end */
   if (b2 <= b1) { *b2_out = b2; return 0; }
   const double eps_orth = s->mach_eps;
   const int nX = b2 - b1, nVL = b1 + numLocked, nrowsA = numLocked + b2;
   double t0 = pa_wtime();
   char *X = PCOL(s, Vp, ldV, b1);

   HS *A = G + (size_t)(b1 + numLocked) * ldG;      /* new columns of the Gram matrix */
   HS *r = NULL;
   if (RLocked) {
      for (int c = 0; c < nX; c++) for (int j = 0; j < numLocked; j++) RLocked[j + (size_t)c * ldRLocked] = 0.0;
      r = (HS *)calloc((size_t)nX * nX, sizeof(HS));
      for (int i = 0; i < nX; i++) r[i + (size_t)i * nX] = 1.0;
   }
   double *D = (double *)malloc((size_t)nX * sizeof(double)), *N = (double *)malloc((size_t)nX * sizeof(double));
   HS *GdA = (HS *)malloc((size_t)(nVL > 0 ? nVL : 1) * nX * sizeof(HS));
   HS *Y = (HS *)malloc((size_t)nX * nX * sizeof(HS)), *Cm = (HS *)malloc((size_t)nX * nX * sizeof(HS));
   HS *M = (HS *)malloc((size_t)nX * nX * sizeof(HS));
   if (!D || !N || !GdA || !Y || !Cm || !M) return PRIMME_MALLOC_FAILURE;
   int rc = 0;
   /* (red_cap counts scalars: the buffers hold twice as many doubles as that, enough for complex entries) */
   if ((size_t)nrowsA * nX > (size_t)s->red_cap || (size_t)nVL * nX > (size_t)s->red_cap || nrowsA > ldG) {
      free(r); free(D); free(N); free(GdA); free(Y); free(Cm); free(M);
      return PRIMME_UNEXPECTED_FAILURE;
   }

   *b2_out = b2;
   const int maxits = 5;
   int plus1 = 5, Yortho = 1;
   hipk_seg segs[2] = {{locked, ldLocked, numLocked}, {Vp, ldV, b1}};
   hipk_seg segsA[2] = {{locked, ldLocked, numLocked}, {Vp, ldV, b2}};
   for (int its = 0; its < maxits; its++) {
      if (its > 0) {
         /* X <- (X - [Q V]*GdA) * M with M = Y*diag(1/D) (eigenvector form) or Y^-1 (Cholesky) */
         if (Yortho) {
            for (int c = 0; c < nX; c++) for (int i = 0; i < nX; i++) M[i + (size_t)c * nX] = Y[i + (size_t)c * nX] / D[c];
         } else {
            /* M = Y^-1 for upper triangular Y: solve Y M = I */
            for (int c = 0; c < nX; c++) for (int i = 0; i < nX; i++) M[i + (size_t)c * nX] = (i == c) ? 1.0 : 0.0;
            pa_trsm_left_upper(nX, nX, Y, nX, M, nX);
         }
         /* coefficients and M travel through the pinned mirror (stream-ordered copies, no synchronisation):
          * h_red is only rewritten by launches that come later in the stream */
         const size_t ncoef = (size_t)nVL * nX, nM = (size_t)nX * nX;
         int fused = 0;
         if (nX <= 8 && ncoef + nM <= (size_t)s->red_cap) {
            memcpy(s->h_red, GdA, ncoef * sizeof(HS));
            memcpy((HS *)s->h_red + ncoef, M, nM * sizeof(HS));
            CHK(hipk_h2d(s->ctx, s->d_red, s->h_red, (ncoef + nM) * sizeof(HS)));
            int rcf = hipk_panel_project_mul(s->ctx, s->dt, s->m, segs, 2, s->d_red, nVL > 0 ? nVL : 1, s->d_red + SD * ncoef, X, ldV, nX);
            if (rcf < 0) return rcf;
            fused = (rcf == 0);
         }
         if (!fused) {
            if (nVL > 0) {
               memcpy(s->h_red, GdA, ncoef * sizeof(HS));
               CHK(hipk_h2d(s->ctx, s->d_red, s->h_red, ncoef * sizeof(HS)));
               CHK(hipk_panel_project(s->ctx, s->dt, s->m, segs, 2, s->d_red, nVL, X, ldV, nX, NULL));
            }
            CHK(right_multiply(s, X, ldV, nX, M));
         }
      }
      /* A = [Q V(0:b2)]' B X: with a mass matrix B X is formed through the callback first (the reference does the same in every
       * sweep, ortho.c:621-629), into the scratch panel */
      const char *Xrhs = X;
      int64_t ldrhs = ldV;
      if (s->B) {
         if (nX > s->nBT) { rc = PRIMME_FUNCTION_UNAVAILABLE; break; }
         if ((rc = pa_apply_B(s, X, ldV, s->BT, s->ld, nX))) break;
         Xrhs = s->BT; ldrhs = s->ld;
      }
      CHK(hipk_panel_dots(s->ctx, s->dt, s->m, segsA, 2, Xrhs, ldrhs, nX, s->d_red, nrowsA));
      CHK(pa_reduce(s, s->d_red, SD * nrowsA * nX, 0, 0));
      for (int c = 0; c < nX; c++)
         for (int i = 0; i < nrowsA; i++) A[i + (size_t)c * ldG] = ((const HS *)s->h_red)[i + (size_t)c * nrowsA];
      p->stats.numOrthoInnerProds += (double)(numLocked + b1) * nX + (double)((nX + 1) / 2) * nX;

      if (rank_estimation(G, numLocked + b1, numLocked + b2, maxRank, ldG) == numLocked + b2) {
         if (its >= plus1) {
            int i;
            for (i = b1; i < b2 && HS_ABS(G[(numLocked + i) + (size_t)(numLocked + i) * ldG] - 1.0) < .8; i++) ;
            if (i >= b2) break;
         } else plus1 = PA_MIN(its + 1, plus1);
      }

      /* overflowing norms: keep only the diagonal */
      for (int i = 0; i < nX; i++) {
         if (HS_RE(A[i + (size_t)i * ldG]) < DBL_MAX) continue;   /* index as in the reference (ortho.c:671) */
         for (int j = 0; j < numLocked + i; j++) G[j + (size_t)(numLocked + i) * ldG] = 0.0;
         A[i + (size_t)i * ldG] = DBL_MAX;
      }

      /* C = X'X - (X'Vc)(Vc'Vc)^-1(Vc'X); GdA = (Vc'Vc)^-1 Vc'X */
      for (int c = 0; c < nX; c++) {
         for (int i = 0; i < nX; i++) Cm[i + (size_t)c * nX] = A[(numLocked + b1 + i) + (size_t)c * ldG];
         for (int i = 0; i < nVL; i++) GdA[i + (size_t)c * nVL] = A[i + (size_t)c * ldG];
      }
      pa_trsm_left_upper_trans(nVL, nX, fG, s->ldVtBV, GdA, nVL);
      for (int c = 0; c < nX; c++)
         for (int rr = 0; rr < nX; rr++) {
            HS t = 0.0;
            for (int i = 0; i < nVL; i++) t += HS_CONJ(GdA[i + (size_t)rr * nVL]) * GdA[i + (size_t)c * nVL];
            Cm[rr + (size_t)c * nX] -= t;
         }
      pa_trsm_left_upper(nVL, nX, fG, s->ldVtBV, GdA, nVL);
      for (int i = 0; i < nX; i++) N[i] = sqrt(PA_MAX(HS_ABS(Cm[i + (size_t)i * nX]), eps_orth));
      for (int i = 0; i < nX; i++)
         for (int j = 0; j <= i; j++) Cm[j + (size_t)i * nX] /= N[i] * N[j];
      if ((rc = decomposition(Cm, nX, Y, D, &Yortho))) break;
      for (int i = 0; i < nX; i++) D[i] = sqrt(PA_MAX(D[i], eps_orth * nX));

      if (RLocked) {
         /* RLocked += GdA(0:numLocked,:) * r;  r <- D .* (Y' or Y) * (N .* r) */
         for (int c = 0; c < nX; c++)
            for (int j = 0; j < numLocked; j++) {
               HS t = 0.0;
               for (int q = 0; q < nX; q++) t += GdA[j + (size_t)q * nVL] * r[q + (size_t)c * nX];
               RLocked[j + (size_t)c * ldRLocked] += t;
            }
         for (int c = 0; c < nX; c++) for (int j = 0; j < nX; j++) r[j + (size_t)c * nX] *= N[j];
         for (int c = 0; c < nX; c++)
            for (int i = 0; i < nX; i++) {
               HS t = 0.0;
               if (Yortho) { for (int q = 0; q < nX; q++) t += HS_CONJ(Y[q + (size_t)i * nX]) * r[q + (size_t)c * nX]; }
               else { for (int q = i; q < nX; q++) t += Y[i + (size_t)q * nX] * r[q + (size_t)c * nX]; }
               Cm[i + (size_t)c * nX] = t;
            }
         for (int c = 0; c < nX; c++) for (int j = 0; j < nX; j++) r[j + (size_t)c * nX] = D[j] * Cm[j + (size_t)c * nX];
      }
      /* Y <- N \ Y (eigenvector form) or Y * N (Cholesky form) */
      if (Yortho) { for (int c = 0; c < nX; c++) for (int j = 0; j < nX; j++) Y[j + (size_t)c * nX] /= N[j]; }
      else { for (int c = 0; c < nX; c++) for (int j = 0; j < nX; j++) Y[j + (size_t)c * nX] *= N[c]; }
   }
   if (!rc) {
      b2 = rank_estimation(G, numLocked + b1, numLocked + b2, maxRank, ldG) - numLocked;
      *b2_out = b2;
      rc = pa_update_cholesky(G, ldG, fG, s->ldVtBV, numLocked + b1, numLocked + b2);
   }
   free(r); free(D); free(N); free(GdA); free(Y); free(Cm); free(M);
   if (s->phase_timing) hipk_sync(s->ctx);
   p->stats.timeOrtho += pa_wtime() - t0;
   return rc;
}
