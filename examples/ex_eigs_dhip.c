/* ex_eigs_dhip.c — the reference's examples/ex_eigs_dseq.c problem (1-D Laplacian n = 100, 10
 * smallest eigenvalues, eps 1e-9, Jacobi preconditioner, PRIMME_DYNAMIC) solved through the C ABI of
 * libprimme_amd.so on an MI355X.  Plain C, no HIP headers: device memory comes from hipk_malloc.
 * What changes against the reference example: the header, the operator set-up (device CSR instead
 * of a host callback) and the solver name (hip_dprimme instead of dprimme).
 *
 *   make -C examples && examples/ex_eigs_dhip        (exit code 0 = eigenvalues match 2 - 2cos(k pi/(n+1)))
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "primme_amd.h"
#include "primme_amd_kernels.h"
#include "primme_amd_comm.h"

int main(void) {
   const int n = 100, nev = 10;
   /* tridiagonal [-1 2 -1] in CSR */
   int32_t *rp = malloc(sizeof(int32_t) * (n + 1)), *ci = malloc(sizeof(int32_t) * 3 * n);
   double *va = malloc(sizeof(double) * 3 * n);
   int nnz = 0;
   for (int i = 0; i < n; i++) {
      rp[i] = nnz;
      if (i > 0) { ci[nnz] = i - 1; va[nnz++] = -1.0; }
      ci[nnz] = i; va[nnz++] = 2.0;
      if (i < n - 1) { ci[nnz] = i + 1; va[nnz++] = -1.0; }
   }
   rp[n] = nnz;

   hipk_ctx *ctx;
   hipk_csr *A;
   primme_amd_operator *op;
   if (hipk_ctx_create(&ctx, NULL)) { fprintf(stderr, "no HIP device\n"); return 2; }
   if (hipk_csr_create(ctx, HIPK_F64, n, n, 0, rp, ci, va, &A)) return 2;
   if (primme_amd_operator_create(&op, A, NULL)) return 2;

   primme_params primme;
   primme_initialize(&primme);
   primme.n = n;
   primme.numEvals = nev;
   primme.eps = 1e-9;
   primme.target = primme_smallest;
   primme.matrix = op;
   primme.matrixMatvec = primme_amd_matvec;                 /* ready-made device CSR matvec */
   primme.preconditioner = op;
   primme.applyPreconditioner = primme_amd_jacobi_precond;  /* (diag(A) - shift)^-1, shifts from the solver */
   primme.correctionParams.precondition = 1;
   primme.printLevel = 0;
   primme_set_method(PRIMME_DYNAMIC, &primme);

   double evals[10], rnorms[10], *evecs_dev;
   if (hipk_malloc(ctx, sizeof(double) * n * nev, (void **)&evecs_dev)) return 2;
   const int ret = hip_dprimme(evals, evecs_dev, rnorms, &primme);

   double *evecs = malloc(sizeof(double) * n * nev);
   hipk_d2h(ctx, evecs, evecs_dev, sizeof(double) * n * nev);
   hipk_sync(ctx);

   int bad = (ret != 0 || primme.initSize != nev);
   printf("hip_dprimme returned %d, %d pairs, %lld outer iterations, %lld matvecs, recommended method %d\n", ret,
         primme.initSize, (long long)primme.stats.numOuterIterations, (long long)primme.stats.numMatvecs,
         primme.dynamicMethodSwitch);
   for (int k = 0; k < primme.initSize; k++) {
      const double exact = 2.0 - 2.0 * cos((k + 1) * M_PI / (n + 1));
      double nrm = 0.0;
      for (int i = 0; i < n; i++) nrm += evecs[i + (size_t)k * n] * evecs[i + (size_t)k * n];
      printf("Eval[%d] = %-22.15E  rnorm %-9.3E  |x| %.15f  error %.1E\n", k + 1, evals[k], rnorms[k], sqrt(nrm), fabs(evals[k] - exact));
      if (fabs(evals[k] - exact) > 1e-10 * 4.0 || fabs(sqrt(nrm) - 1.0) > 1e-10) bad = 1;
   }
   hipk_free(ctx, evecs_dev);
   primme_amd_operator_destroy(op);
   hipk_csr_destroy(A);
   hipk_ctx_destroy(ctx);
   primme_free(&primme);
   free(rp); free(ci); free(va); free(evecs);
   return bad;
}
