/* ex_svds_dseq_host.c — BASELINE configs[4] in the words it is written in: dprimme_svds() (reference
 * include/primme_svds.h:240, the CPU library's entry point) on a rectangular sparse matrix with 5 nonzeros per
 * row, the 10 largest singular triplets, normal-equations method — written the way a CPU application writes it:
 * HOST matvec callback (A x and A'x on host arrays), HOST svecs.  Nothing here knows about the GPU: the library
 * stages every block through pinned memory (csrc/svds_hostapi.c).  For the device rate hand hip_dprimme_svds the
 * library's CSR operator instead (examples/ex_svds_dhip.c).
 *
 *   make -C examples && examples/ex_svds_dseq_host   (exit code 0 = every triplet satisfies
 *                                                     |A v - s u|, |A'u - s v| <= eps |A| and U, V are orthonormal)
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "primme_amd_svds.h"

#define NNZ_ROW 5
typedef struct { int m, n; int *col; double *val; } sparse_rows;      /* m x n, NNZ_ROW entries per row */

static void build(sparse_rows *A, int m, int n) {
   A->m = m; A->n = n;
   A->col = (int *)malloc(sizeof(int) * (size_t)m * NNZ_ROW);
   A->val = (double *)malloc(sizeof(double) * (size_t)m * NNZ_ROW);
   unsigned long long s = 88172645463325252ULL;
   for (int i = 0; i < m; i++)
      for (int k = 0; k < NNZ_ROW; k++) {
         s ^= s << 13; s ^= s >> 7; s ^= s << 17;                      /* xorshift: columns scattered over the width */
         A->col[i * NNZ_ROW + k] = (k == 0) ? i % n : (int)(s % (unsigned long long)n);
         A->val[i * NNZ_ROW + k] = (k == 0) ? 4.0 + 6.0 * (double)(i % n) / n : ((double)((s >> 20) % 2001) - 1000.0) / 2000.0;
      }
}

/* y = A x (transpose == 0: x has n rows, y has m) or y = A'x */
static void matvec(void *x, PRIMME_INT *ldx, void *y, PRIMME_INT *ldy, int *blockSize, int *transpose,
      primme_svds_params *ps, int *ierr) {
   const sparse_rows *A = (const sparse_rows *)ps->matrix;
   for (int c = 0; c < *blockSize; c++) {
      const double *xv = (const double *)x + *ldx * c;
      double *yv = (double *)y + *ldy * c;
      if (!*transpose) {
         for (int i = 0; i < A->m; i++) {
            double t = 0.0;
            for (int k = 0; k < NNZ_ROW; k++) t += A->val[i * NNZ_ROW + k] * xv[A->col[i * NNZ_ROW + k]];
            yv[i] = t;
         }
      } else {
         for (int j = 0; j < A->n; j++) yv[j] = 0.0;
         for (int i = 0; i < A->m; i++)
            for (int k = 0; k < NNZ_ROW; k++) yv[A->col[i * NNZ_ROW + k]] += A->val[i * NNZ_ROW + k] * xv[i];
      }
   }
   *ierr = 0;
}

int main(void) {
   const int m = 8000, n = 2000, k = 10;
   sparse_rows A;
   build(&A, m, n);
   primme_svds_params ps;
   primme_svds_initialize(&ps);
   ps.m = m; ps.n = n;
   ps.numSvals = k;
   ps.eps = 1e-8;
   ps.target = primme_svds_largest;
   ps.matrixMatvec = matvec;
   ps.matrix = &A;
   ps.printLevel = 0;
   primme_svds_set_method(primme_svds_normalequations, PRIMME_DEFAULT_MIN_MATVECS, PRIMME_DEFAULT_METHOD, &ps);

   double *svals = (double *)malloc(sizeof(double) * k), *rnorms = (double *)malloc(sizeof(double) * k);
   double *svecs = (double *)calloc((size_t)(m + n) * k, sizeof(double));
   int ret = dprimme_svds(svals, svecs, rnorms, &ps);
   printf("dprimme_svds returned %d: %d triplets, %lld outer iterations, %lld matvecs, |A| estimate %.6f\n", ret, ps.initSize,
         (long long)ps.stats.numOuterIterations, (long long)ps.stats.numMatvecs, ps.aNorm);
   int bad = (ret != 0 || ps.initSize != k);
   double *t = (double *)malloc(sizeof(double) * (size_t)(m + n));
   const double *U = svecs, *V = svecs + (size_t)m * k;
   for (int i = 0; i < ps.initSize && !bad; i++) {
      PRIMME_INT ldn = n, ldm = m;
      int one = 1, no = 0, tr = 1, e = 0;
      double r1 = 0.0, r2 = 0.0;
      matvec((void *)(V + (size_t)n * i), &ldn, t, &ldm, &one, &no, &ps, &e);
      for (int j = 0; j < m; j++) { const double d = t[j] - svals[i] * U[(size_t)m * i + j]; r1 += d * d; }
      matvec((void *)(U + (size_t)m * i), &ldm, t, &ldn, &one, &tr, &ps, &e);
      for (int j = 0; j < n; j++) { const double d = t[j] - svals[i] * V[(size_t)n * i + j]; r2 += d * d; }
      printf("  sval[%d] = %.12f  reported |r| %.2e  recomputed %.2e\n", i, svals[i], rnorms[i], sqrt(r1 + r2));
      if (!(sqrt(r1 + r2) <= ps.eps * ps.aNorm * 1.5) || (i > 0 && svals[i] > svals[i - 1])) bad = 1;
      for (int l = 0; l <= i; l++) {
         double uu = 0.0, vv = 0.0;
         for (int j = 0; j < m; j++) uu += U[(size_t)m * i + j] * U[(size_t)m * l + j];
         for (int j = 0; j < n; j++) vv += V[(size_t)n * i + j] * V[(size_t)n * l + j];
         if (fabs(uu - (l == i)) > 1e-7 || fabs(vv - (l == i)) > 1e-7) bad = 1;
      }
   }
   free(t); free(svals); free(rnorms); free(svecs); free(A.col); free(A.val);
   primme_svds_free(&ps);
   if (bad) { printf("FAILED\n"); return 1; }
   return 0;
}
