/* ex_eigs_zhip_precond.hip — Hermitian problem, the APPLICATION's own device matvec AND a NON-Hermitian
 * diagonal preconditioner: K = diag(A) (1 + i gamma w_j), w_j = ((j mod 7) - 3)/3.  With RightX + SkewX
 * (JDQMR, GD_Olsen) the solver then needs x'K^-1 x as a COMPLEX number (the reference keeps it as an HSCALAR:
 * src/eigs/correction.c:969-977, used at src/eigs/inner_solve.c:737-741) — what this program pins.
 *
 *   A = tridiag(conj(a), d_j, a),  d_j = 1 + j,  a = -0.5 exp(0.7 i)   (unitarily similar to a real tridiagonal)
 *
 *   ex_eigs_zhip_precond <method>      method = jdqmr | jdqmr_etol | gd_olsen
 * prints one line  RESULT {...}  with eigenvalues, residual norms and counts; exit code 0 = eigenpairs verified.
 * tests/test_c_examples_gpu.py compares the line with tests/golden/reference_zprecond.json = the real
 * reference's zprimme on the same problem, start vector and preconditioner (tests/golden/make_zprecond_golden.py).
 */
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "primme_amd.h"

struct problem { int64_t n; double2 a; double gamma; };

__global__ void herm_tridiag(const double2 *__restrict__ x, int64_t ldx, double2 *__restrict__ y, int64_t ldy,
      int64_t n, int ncols, double2 a) {
   const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= n) return;
   const double d = 1.0 + (double)i;
   for (int c = 0; c < ncols; c++) {
      const double2 *xc = x + (size_t)c * ldx;
      double2 v = make_double2(d * xc[i].x, d * xc[i].y);
      if (i + 1 < n) {          /* A(i, i+1) = a */
         v.x += a.x * xc[i + 1].x - a.y * xc[i + 1].y;
         v.y += a.x * xc[i + 1].y + a.y * xc[i + 1].x;
      }
      if (i > 0) {              /* A(i, i-1) = conj(a) */
         v.x += a.x * xc[i - 1].x + a.y * xc[i - 1].y;
         v.y += a.x * xc[i - 1].y - a.y * xc[i - 1].x;
      }
      y[(size_t)c * ldy + i] = v;
   }
}
/* y = x / (d_j (1 + i gamma w_j)) */
__global__ void rotated_jacobi(const double2 *__restrict__ x, int64_t ldx, double2 *__restrict__ y, int64_t ldy,
      int64_t n, int ncols, double gamma) {
   const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= n) return;
   const double d = 1.0 + (double)i;
   const double w = (double)((int)(i % 7) - 3) / 3.0;
   const double kr = d, ki = d * (gamma * w);            /* K_jj = d (1 + i gamma w) */
   const double den = kr * kr + ki * ki;
   for (int c = 0; c < ncols; c++) {
      const double2 v = x[(size_t)c * ldx + i];
      y[(size_t)c * ldy + i] = make_double2((v.x * kr + v.y * ki) / den, (v.y * kr - v.x * ki) / den);
   }
}

static void matvec(void *x, PRIMME_INT *ldx, void *y, PRIMME_INT *ldy, int *blockSize, primme_params *primme, int *ierr) {
   const problem *pb = (const problem *)primme->matrix;
   hipStream_t stream = primme->queue ? *(hipStream_t *)primme->queue : 0;
   hipLaunchKernelGGL(herm_tridiag, dim3((unsigned)((pb->n + 255) / 256)), dim3(256), 0, stream,
         (const double2 *)x, *ldx, (double2 *)y, *ldy, pb->n, *blockSize, pb->a);
   *ierr = hipGetLastError() != hipSuccess;
}
static void precond(void *x, PRIMME_INT *ldx, void *y, PRIMME_INT *ldy, int *blockSize, primme_params *primme, int *ierr) {
   const problem *pb = (const problem *)primme->preconditioner;
   hipStream_t stream = primme->queue ? *(hipStream_t *)primme->queue : 0;
   hipLaunchKernelGGL(rotated_jacobi, dim3((unsigned)((pb->n + 255) / 256)), dim3(256), 0, stream,
         (const double2 *)x, *ldx, (double2 *)y, *ldy, pb->n, *blockSize, pb->gamma);
   *ierr = hipGetLastError() != hipSuccess;
}

int main(int argc, char **argv) {
   const int64_t n = 2000;
   const int nev = 4;
   const char *method = argc > 1 ? argv[1] : "jdqmr";
   problem pb = {n, make_double2(-0.5 * cos(0.7), -0.5 * sin(0.7)), 0.1};

   primme_params primme;
   primme_initialize(&primme);
   primme.n = n;
   primme.numEvals = nev;
   primme.eps = 1e-10;
   primme.aNorm = 2001.0;
   primme.target = primme_smallest;
   primme.matrix = &pb;
   primme.matrixMatvec = matvec;
   primme.preconditioner = &pb;
   primme.applyPreconditioner = precond;
   primme.correctionParams.precondition = 1;
   primme.maxMatvecs = 20000;
   primme.printLevel = 0;
   primme.initSize = 1;
   primme_set_method(!strcmp(method, "jdqmr") ? PRIMME_JDQMR : !strcmp(method, "jdqmr_etol") ? PRIMME_JDQMR_ETol : PRIMME_GD_Olsen_plusK, &primme);

   double evals[4], rnorms[4];
   double2 *evecs_dev, *evecs = (double2 *)malloc(sizeof(double2) * n * nev);
   if (hipMalloc((void **)&evecs_dev, sizeof(double2) * n * nev) != hipSuccess) { fprintf(stderr, "no HIP device\n"); return 2; }
   /* start vector: exactly representable quotients, the same bits as primme_amd.problems.rational_complex_start_vector */
   for (int64_t j = 0; j < n; j++) evecs[j] = make_double2((double)((j * 7 + 3) % 11 - 5) / 5.0, (double)((j * 5 + 1) % 13 - 6) / 6.0);
   (void)hipMemcpy(evecs_dev, evecs, sizeof(double2) * n, hipMemcpyHostToDevice);
   const int ret = hip_zprimme(evals, evecs_dev, rnorms, &primme);
   (void)hipMemcpy(evecs, evecs_dev, sizeof(double2) * n * nev, hipMemcpyDeviceToHost);

   int bad = (ret != 0 || primme.initSize != nev);
   double worst = 0.0;
   for (int k = 0; k < primme.initSize; k++) {
      const double2 *z = evecs + (size_t)k * n;
      double r2 = 0.0, z2 = 0.0;
      for (int64_t i = 0; i < n; i++) {
         const double d = 1.0 + (double)i;
         double vx = (d - evals[k]) * z[i].x, vy = (d - evals[k]) * z[i].y;
         if (i + 1 < n) { vx += pb.a.x * z[i + 1].x - pb.a.y * z[i + 1].y; vy += pb.a.x * z[i + 1].y + pb.a.y * z[i + 1].x; }
         if (i > 0) { vx += pb.a.x * z[i - 1].x + pb.a.y * z[i - 1].y; vy += pb.a.x * z[i - 1].y - pb.a.y * z[i - 1].x; }
         r2 += vx * vx + vy * vy;
         z2 += z[i].x * z[i].x + z[i].y * z[i].y;
      }
      if (sqrt(r2) > worst) worst = sqrt(r2);
      if (sqrt(r2) > 1.5e-10 * 2001.0 || fabs(sqrt(z2) - 1.0) > 1e-9) bad = 1;
   }
   printf("RESULT {\"method\": \"%s\", \"ret\": %d, \"initSize\": %d, \"evals\": [%.17g, %.17g, %.17g, %.17g], \"resNorms\": [%.6e, %.6e, %.6e, %.6e], "
          "\"numOuterIterations\": %lld, \"numMatvecs\": %lld, \"numRestarts\": %lld, \"numPreconds\": %lld, \"worst_true_residual\": %.3e}\n",
         method, ret, primme.initSize, evals[0], evals[1], evals[2], evals[3], rnorms[0], rnorms[1], rnorms[2], rnorms[3],
         (long long)primme.stats.numOuterIterations, (long long)primme.stats.numMatvecs, (long long)primme.stats.numRestarts,
         (long long)primme.stats.numPreconds, worst);
   free(evecs);
   (void)hipFree(evecs_dev);
   primme_free(&primme);
   return bad;
}
