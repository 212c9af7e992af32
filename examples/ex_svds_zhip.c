/* ex_svds_zhip.c — singular values of a COMPLEX matrix through the C ABI (cf. reference examples/ex_svds_zseq.c):
 * A(i, i mod n) = (1 + (i mod n)/n) e^{i phi_i} for i < m, so A^H A is diagonal with entries 5 (1 + j/n)^2 and the
 * largest singular values are known.  The library's own operator takes the real-equivalent form of A (every entry
 * a + ib becomes the block [[a, -b], [b, a]]) and is told that the vectors it is handed are complex.
 *   make -C examples && examples/ex_svds_zhip */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "primme_amd_svds.h"
#include "primme_amd_kernels.h"

int main(void) {
   const int m = 500, n = 100, k = 4;
   /* real-equivalent CSR: 2m rows, two entries in each */
   int32_t *rp = malloc(sizeof(int32_t) * (2 * m + 1)), *ci = malloc(sizeof(int32_t) * 4 * m);
   double *va = malloc(sizeof(double) * 4 * m);
   for (int i = 0; i < m; i++) {
      const double mod = 1.0 + (double)(i % n) / n, phi = 0.7 * i;
      const double a = mod * cos(phi), b = mod * sin(phi);
      const int j = i % n;
      rp[2 * i] = 4 * i; rp[2 * i + 1] = 4 * i + 2;
      ci[4 * i] = 2 * j; va[4 * i] = a;       ci[4 * i + 1] = 2 * j + 1; va[4 * i + 1] = -b;   /* real part of row i */
      ci[4 * i + 2] = 2 * j; va[4 * i + 2] = b; ci[4 * i + 3] = 2 * j + 1; va[4 * i + 3] = a;  /* imaginary part   */
   }
   rp[2 * m] = 4 * m;

   hipk_ctx *ctx;
   primme_amd_svds_operator *op;
   if (hipk_ctx_create(&ctx, NULL)) { fprintf(stderr, "no HIP device\n"); return 2; }
   if (primme_amd_svds_operator_create(&op, ctx, HIPK_F64, 2 * m, 2 * n, rp, ci, va)) return 2;
   primme_amd_svds_operator_set_complex(op, 1);

   primme_svds_params ps;
   primme_svds_initialize(&ps);
   ps.m = m; ps.n = n; ps.numSvals = k; ps.eps = 1e-10; ps.target = primme_svds_largest; ps.printLevel = 0;   /* complex sizes */
   ps.matrix = op;
   ps.matrixMatvec = primme_amd_svds_matvec;
   primme_svds_set_method(primme_svds_default, PRIMME_DEFAULT_METHOD, PRIMME_DEFAULT_METHOD, &ps);

   double svals[4], rnorms[4];
   void *svecs_dev;                                           /* (m + n) k complex numbers */
   if (hipk_malloc(ctx, 2 * sizeof(double) * (m + n) * k, &svecs_dev)) return 2;
   const int ret = hip_zprimme_svds(svals, svecs_dev, rnorms, &ps);
   int bad = (ret != 0 || ps.initSize != k);
   printf("hip_zprimme_svds returned %d, %d triplets, %lld operator applications\n", ret, ps.initSize, (long long)ps.stats.numMatvecs);
   for (int i = 0; i < ps.initSize; i++) {
      const double exact = sqrt(5.0) * (1.0 + (double)(n - 1 - i) / n);
      printf("Sval[%d] = %-22.15E  rnorm %-9.3E  error %.1E\n", i + 1, svals[i], rnorms[i], fabs(svals[i] - exact));
      if (fabs(svals[i] - exact) > 1e-9 * svals[0] || rnorms[i] > 1e-8 * svals[0]) bad = 1;
   }
   hipk_free(ctx, svecs_dev);
   primme_amd_svds_operator_destroy(op);
   hipk_ctx_destroy(ctx);
   free(rp); free(ci); free(va);
   return bad;
}
