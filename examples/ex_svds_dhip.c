/* ex_svds_dhip.c — singular values through the C ABI: a 500 x 100 bidiagonal-like matrix with known
 * largest singular values (cf. reference examples/ex_svds_dseq.c), default (hybrid) method.
 *   make -C examples && examples/ex_svds_dhip */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "primme_amd_svds.h"
#include "primme_amd_kernels.h"

int main(void) {
   const int m = 500, n = 100, k = 4;
   /* A(i, i mod n) = 1 + (i mod n)/n  for i < m: A'A is diagonal with entries 5 (1 + j/n)^2 */
   int32_t *rp = malloc(sizeof(int32_t) * (m + 1)), *ci = malloc(sizeof(int32_t) * m);
   double *va = malloc(sizeof(double) * m);
   for (int i = 0; i < m; i++) { rp[i] = i; ci[i] = i % n; va[i] = 1.0 + (double)(i % n) / n; }
   rp[m] = m;

   hipk_ctx *ctx;
   primme_amd_svds_operator *op;
   if (hipk_ctx_create(&ctx, NULL)) { fprintf(stderr, "no HIP device\n"); return 2; }
   if (primme_amd_svds_operator_create(&op, ctx, HIPK_F64, m, n, rp, ci, va)) return 2;

   primme_svds_params ps;
   primme_svds_initialize(&ps);
   ps.m = m; ps.n = n; ps.numSvals = k; ps.eps = 1e-10; ps.target = primme_svds_largest; ps.printLevel = 0;
   ps.matrix = op;
   ps.matrixMatvec = primme_amd_svds_matvec;
   primme_svds_set_method(primme_svds_default, PRIMME_DEFAULT_METHOD, PRIMME_DEFAULT_METHOD, &ps);

   double svals[4], rnorms[4], *svecs_dev;
   if (hipk_malloc(ctx, sizeof(double) * (m + n) * k, (void **)&svecs_dev)) return 2;
   const int ret = hip_dprimme_svds(svals, svecs_dev, rnorms, &ps);
   int bad = (ret != 0 || ps.initSize != k);
   printf("hip_dprimme_svds returned %d, %d triplets, %lld operator applications\n", ret, ps.initSize, (long long)ps.stats.numMatvecs);
   for (int i = 0; i < ps.initSize; i++) {
      const double exact = sqrt(5.0) * (1.0 + (double)(n - 1 - i) / n);
      printf("Sval[%d] = %-22.15E  rnorm %-9.3E  error %.1E\n", i + 1, svals[i], rnorms[i], fabs(svals[i] - exact));
      if (fabs(svals[i] - exact) > 1e-9 * svals[0]) bad = 1;
   }
   hipk_free(ctx, svecs_dev);
   primme_amd_svds_operator_destroy(op);
   hipk_ctx_destroy(ctx);
   free(rp); free(ci); free(va);
   return bad;
}
