/* ex_eigs_zhip.hip — a Hermitian problem with the APPLICATION's own device matvec, as in the
 * reference's examples/ex_eigs_zseq.c (complex 1-D Laplacian-like operator) but on the GPU:
 * the callback receives device pointers to complex vectors (re, im pairs, leading dimension in
 * complex elements) and launches on the solver's stream, *(hipStream_t *)primme->queue.
 *
 *   A = tridiag(conj(a), 2, a),  a = -exp(0.7 i):  eigenvalues 2 - 2 cos(k pi / (n+1))  (A is
 *   unitarily similar to the real tridiag(-1, 2, -1)).
 *
 *   make -C examples ex_eigs_zhip && examples/ex_eigs_zhip      (exit code 0 = eigenpairs verified)
 */
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "primme_amd.h"

struct problem { int64_t n; double2 a; };

__global__ void herm_tridiag(const double2 *__restrict__ x, int64_t ldx, double2 *__restrict__ y, int64_t ldy,
      int64_t n, int ncols, double2 a) {
   const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= n) return;
   for (int c = 0; c < ncols; c++) {
      const double2 *xc = x + (size_t)c * ldx;
      double2 v = make_double2(2.0 * xc[i].x, 2.0 * xc[i].y);
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

static void matvec(void *x, PRIMME_INT *ldx, void *y, PRIMME_INT *ldy, int *blockSize, primme_params *primme, int *ierr) {
   const problem *pb = (const problem *)primme->matrix;
   hipStream_t stream = primme->queue ? *(hipStream_t *)primme->queue : 0;
   const int threads = 256;
   hipLaunchKernelGGL(herm_tridiag, dim3((unsigned)((pb->n + threads - 1) / threads)), dim3(threads), 0, stream,
         (const double2 *)x, *ldx, (double2 *)y, *ldy, pb->n, *blockSize, pb->a);
   *ierr = hipGetLastError() != hipSuccess;
}

int main(void) {
   const int64_t n = 20000;
   const int nev = 4;
   problem pb = {n, make_double2(-cos(0.7), -sin(0.7))};

   primme_params primme;
   primme_initialize(&primme);
   primme.n = n;
   primme.numEvals = nev;
   primme.eps = 1e-10;
   primme.aNorm = 4.0;
   primme.target = primme_largest;
   primme.matrix = &pb;
   primme.matrixMatvec = matvec;
   primme.printLevel = 0;
   primme_set_method(PRIMME_DEFAULT_MIN_TIME, &primme);

   double evals[4], rnorms[4];
   double2 *evecs_dev;
   if (hipMalloc((void **)&evecs_dev, sizeof(double2) * n * nev) != hipSuccess) { fprintf(stderr, "no HIP device\n"); return 2; }
   const int ret = hip_zprimme(evals, evecs_dev, rnorms, &primme);

   double2 *evecs = (double2 *)malloc(sizeof(double2) * n * nev);
   (void)hipMemcpy(evecs, evecs_dev, sizeof(double2) * n * nev, hipMemcpyDeviceToHost);
   int bad = (ret != 0 || primme.initSize != nev);
   printf("hip_zprimme returned %d, %d pairs, %lld outer iterations, %lld matvecs\n", ret, primme.initSize,
         (long long)primme.stats.numOuterIterations, (long long)primme.stats.numMatvecs);
   for (int k = 0; k < primme.initSize; k++) {
      const double exact = 2.0 - 2.0 * cos((double)(n - k) * M_PI / (double)(n + 1));
      /* true residual on the host: r = A z - lambda z */
      const double2 *z = evecs + (size_t)k * n;
      double r2 = 0.0, z2 = 0.0;
      for (int64_t i = 0; i < n; i++) {
         double vx = (2.0 - evals[k]) * z[i].x, vy = (2.0 - evals[k]) * z[i].y;
         if (i + 1 < n) { vx += pb.a.x * z[i + 1].x - pb.a.y * z[i + 1].y; vy += pb.a.x * z[i + 1].y + pb.a.y * z[i + 1].x; }
         if (i > 0) { vx += pb.a.x * z[i - 1].x + pb.a.y * z[i - 1].y; vy += pb.a.x * z[i - 1].y - pb.a.y * z[i - 1].x; }
         r2 += vx * vx + vy * vy;
         z2 += z[i].x * z[i].x + z[i].y * z[i].y;
      }
      printf("Eval[%d] = %-22.15E  rnorm %-9.3E  true %-9.3E  |z| %.12f  error %.1E\n", k + 1, evals[k], rnorms[k], sqrt(r2), sqrt(z2),
            fabs(evals[k] - exact));
      if (fabs(evals[k] - exact) > 1e-10 * 4.0 || sqrt(r2) > 1.5e-10 * 4.0 || fabs(sqrt(z2) - 1.0) > 1e-9) bad = 1;
   }
   free(evecs);
   (void)hipFree(evecs_dev);
   primme_free(&primme);
   return bad;
}
