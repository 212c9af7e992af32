/* ex_eigs_dseq_host.c — BASELINE configs[0]: the problem of the reference's examples/ex_eigs_dseq.c
 * (1-D Laplacian n = 100, 10 smallest eigenvalues, eps 1e-9, Jacobi preconditioner, PRIMME_DYNAMIC)
 * written the way a CPU application writes it — HOST matvec / preconditioner callbacks, HOST evecs —
 * and linked against libprimme_amd.so through the reference's own entry point name, dprimme()
 * (include/primme_eigs.h:386).  Nothing here knows about the GPU: the library stages the vectors.
 *
 *   make -C examples && examples/ex_eigs_dseq_host   (exit code 0 = eigenvalues match 2 - 2cos(k pi/(n+1)))
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "primme_amd.h"

/* y = tridiag(-1, 2, -1) x for a block of host vectors */
static void laplacian_matvec(void *x, PRIMME_INT *ldx, void *y, PRIMME_INT *ldy, int *blockSize,
      primme_params *primme, int *ierr) {
   const PRIMME_INT n = primme->n;
   for (int c = 0; c < *blockSize; c++) {
      const double *xv = (const double *)x + *ldx * c;
      double *yv = (double *)y + *ldy * c;
      for (PRIMME_INT i = 0; i < n; i++) {
         double t = 2.0 * xv[i];
         if (i > 0) t -= xv[i - 1];
         if (i + 1 < n) t -= xv[i + 1];
         yv[i] = t;
      }
   }
   *ierr = 0;
}

/* y = (diag(A) - shift)^-1 x with the shift the solver publishes for each block vector */
static void jacobi_precond(void *x, PRIMME_INT *ldx, void *y, PRIMME_INT *ldy, int *blockSize,
      primme_params *primme, int *ierr) {
   const PRIMME_INT n = primme->n;
   for (int c = 0; c < *blockSize; c++) {
      const double shift = primme->ShiftsForPreconditioner ? primme->ShiftsForPreconditioner[c] : 0.0;
      double d = 2.0 - shift;
      if (fabs(d) < 1e-14) d = d < 0 ? -1e-14 : 1e-14;
      const double *xv = (const double *)x + *ldx * c;
      double *yv = (double *)y + *ldy * c;
      for (PRIMME_INT i = 0; i < n; i++) yv[i] = xv[i] / d;
   }
   *ierr = 0;
}

int main(void) {
   const int n = 100, nev = 10;
   primme_params primme;
   primme_initialize(&primme);
   primme.n = n;
   primme.numEvals = nev;
   primme.eps = 1e-9;
   primme.target = primme_smallest;
   primme.matrixMatvec = laplacian_matvec;
   primme.applyPreconditioner = jacobi_precond;
   primme.correctionParams.precondition = 1;
   primme.printLevel = 0;
   primme_set_method(PRIMME_DYNAMIC, &primme);

   double evals[10], rnorms[10];
   double *evecs = (double *)calloc((size_t)n * nev, sizeof(double));   /* HOST */
   const int ret = dprimme(evals, evecs, rnorms, &primme);

   int bad = (ret != 0 || primme.initSize != nev);
   printf("dprimme returned %d, %d pairs, %lld outer iterations, %lld matvecs\n", ret, primme.initSize,
         (long long)primme.stats.numOuterIterations, (long long)primme.stats.numMatvecs);
   const double pi = 3.14159265358979323846;
   for (int k = 0; k < nev && ret == 0; k++) {
      const double exact = 2.0 - 2.0 * cos((k + 1) * pi / (n + 1));
      /* true residual on the host from the returned host vectors */
      double r2 = 0.0, nrm = 0.0;
      for (int i = 0; i < n; i++) {
         const double *v = evecs + (size_t)k * n;
         double t = 2.0 * v[i] - (i > 0 ? v[i - 1] : 0.0) - (i + 1 < n ? v[i + 1] : 0.0) - evals[k] * v[i];
         r2 += t * t; nrm += v[i] * v[i];
      }
      printf("Eval[%d] = %.15e  error %.1e  rnorm %.1e  true %.1e\n", k + 1, evals[k], fabs(evals[k] - exact), rnorms[k], sqrt(r2));
      if (fabs(evals[k] - exact) > 1e-10 * 4.0 || rnorms[k] > 1e-9 * primme.aNorm * 1.01 || sqrt(r2) > 2e-9 * primme.aNorm ||
            fabs(nrm - 1.0) > 1e-8) bad = 1;
   }
   free(evecs);
   return bad;
}
