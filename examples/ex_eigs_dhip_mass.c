/* ex_eigs_dhip_mass.c — a generalised problem A x = lambda B x through the C ABI of libprimme_amd.so: A the 1-D Laplacian
 * [-1 2 -1] (n = 200), B the consistent mass matrix of linear finite elements [1 4 1] / 6 — both tridiagonal Toeplitz, so the
 * eigenvalues are known in closed form, lambda_k = (2 - 2 c_k) / ((4 + 2 c_k) / 6) with c_k = cos(k pi / (n + 1)).
 * What a caller of the reference changes: nothing but the entry point — primme.massMatrix / primme.massMatrixMatvec are the
 * reference's own fields (primme_eigs.h:182-185); here B is a second device CSR behind the ready-made callback
 * primme_amd_mass_matvec.  Solved twice: GD+k and JDQMR (the inner solver on A - sigma B with the Jacobi preconditioner).
 *
 *   make -C examples && examples/ex_eigs_dhip_mass        (exit code 0 = eigenvalues match, vectors B-orthonormal)
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "primme_amd.h"
#include "primme_amd_kernels.h"
#include "primme_amd_comm.h"

static void tridiag(int n, double lo, double di, int32_t *rp, int32_t *ci, double *va) {
   int nnz = 0;
   for (int i = 0; i < n; i++) {
      rp[i] = nnz;
      if (i > 0) { ci[nnz] = i - 1; va[nnz++] = lo; }
      ci[nnz] = i; va[nnz++] = di;
      if (i < n - 1) { ci[nnz] = i + 1; va[nnz++] = lo; }
   }
   rp[n] = nnz;
}

int main(void) {
   const int n = 200, nev = 6;
   int32_t *rp = malloc(sizeof(int32_t) * (n + 1)), *ci = malloc(sizeof(int32_t) * 3 * n);
   double *va = malloc(sizeof(double) * 3 * n), *vb = malloc(sizeof(double) * 3 * n);
   tridiag(n, -1.0, 2.0, rp, ci, va);
   tridiag(n, 1.0 / 6.0, 4.0 / 6.0, rp, ci, vb);

   hipk_ctx *ctx;
   hipk_csr *A, *B;
   primme_amd_operator *opA, *opB;
   if (hipk_ctx_create(&ctx, NULL)) { fprintf(stderr, "no HIP device\n"); return 2; }
   if (hipk_csr_create(ctx, HIPK_F64, n, n, 0, rp, ci, va, &A) || hipk_csr_create(ctx, HIPK_F64, n, n, 0, rp, ci, vb, &B)) return 2;
   if (primme_amd_operator_create(&opA, A, NULL) || primme_amd_operator_create(&opB, B, NULL)) return 2;

   double *evecs_dev, *evecs = malloc(sizeof(double) * n * nev);
   if (hipk_malloc(ctx, sizeof(double) * n * nev, (void **)&evecs_dev)) return 2;
   int bad = 0;
   const primme_preset_method methods[2] = {PRIMME_GD_plusK, PRIMME_JDQMR};
   for (int run = 0; run < 2; run++) {
      primme_params primme;
      primme_initialize(&primme);
      primme.n = n;
      primme.numEvals = nev;
      primme.eps = 1e-10;
      primme.target = primme_smallest;
      primme.matrix = opA;
      primme.matrixMatvec = primme_amd_matvec;
      primme.massMatrix = opB;
      primme.massMatrixMatvec = primme_amd_mass_matvec;          /* y = B x on device pointers */
      if (run == 1) {
         primme.preconditioner = opA;
         primme.applyPreconditioner = primme_amd_jacobi_precond;
         primme.correctionParams.precondition = 1;
         primme.locking = 1;
      }
      primme_set_method(methods[run], &primme);
      double evals[6], rnorms[6];
      const int ret = hip_dprimme(evals, evecs_dev, rnorms, &primme);
      hipk_d2h(ctx, evecs, evecs_dev, sizeof(double) * n * nev);
      hipk_sync(ctx);
      printf("%s: hip_dprimme returned %d, %d pairs, %lld outer iterations, %lld applications of A and B\n", run ? "JDQMR" : "GD+k", ret,
            primme.initSize, (long long)primme.stats.numOuterIterations, (long long)primme.stats.numMatvecs);
      if (ret != 0 || primme.initSize != nev) bad = 1;
      for (int k = 0; k < primme.initSize; k++) {
         const double c = cos((k + 1) * M_PI / (n + 1)), exact = (2.0 - 2.0 * c) / ((4.0 + 2.0 * c) / 6.0);
         /* x' B x with the tridiagonal B */
         double xbx = 0.0;
         const double *x = evecs + (size_t)k * n;
         for (int i = 0; i < n; i++) xbx += x[i] * ((4.0 / 6.0) * x[i] + (i > 0 ? x[i - 1] / 6.0 : 0.0) + (i < n - 1 ? x[i + 1] / 6.0 : 0.0));
         printf("  Eval[%d] = %-22.15E  rnorm %-9.3E  x'Bx %.15f  error %.1E\n", k + 1, evals[k], rnorms[k], xbx, fabs(evals[k] - exact));
         if (fabs(evals[k] - exact) > 1e-9 * 12.0 || fabs(xbx - 1.0) > 1e-9) bad = 1;
      }
      primme_free(&primme);
   }
   hipk_free(ctx, evecs_dev);
   primme_amd_operator_destroy(opA); primme_amd_operator_destroy(opB);
   hipk_csr_destroy(A); hipk_csr_destroy(B);
   hipk_ctx_destroy(ctx);
   free(rp); free(ci); free(va); free(vb); free(evecs);
   return bad;
}
