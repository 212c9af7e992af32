/* test_hip_wrapper.c — calls the reference-named backend routines of include/primme_amd_wrapper.h
 * (Num_gemm_ddh / Num_gemm_dhd / Num_gemv_* / Num_dot / Num_axpy / Num_scal / Num_compute_gramm_ddh for
 * the double GPU instantiation) on random tall-skinny panels and checks them against plain loops.
 * Exit code 0 = all within 1e-12 relative. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "primme_amd_wrapper.h"

static double rnd(unsigned *s) { *s = *s * 1664525u + 1013904223u; return ((double)(*s >> 8) / 16777216.0) - 0.5; }

int main(void) {
   const PRIMME_INT m = 20011, ld = 20012;
   const int k = 13, n = 5;
   hipk_ctx *ctx;
   if (hipk_ctx_create(&ctx, NULL)) { fprintf(stderr, "no device\n"); return 2; }
   unsigned seed = 7u;
   double *A = malloc(sizeof(double) * ld * k), *B = malloc(sizeof(double) * ld * n), *S = malloc(sizeof(double) * k * n);
   for (PRIMME_INT i = 0; i < ld * k; i++) A[i] = rnd(&seed);
   for (PRIMME_INT i = 0; i < ld * n; i++) B[i] = rnd(&seed);
   for (int i = 0; i < k * n; i++) S[i] = rnd(&seed);
   double *dA, *dB, *dC;
   int bad = 0;
   bad |= Num_malloc_hip_dprimme(ld * k, &dA, ctx) | Num_malloc_hip_dprimme(ld * n, &dB, ctx) | Num_malloc_hip_dprimme(ld * n, &dC, ctx);
   bad |= Num_set_matrix_hip_dprimme(A, m, k, ld, dA, ld, ctx) | Num_set_matrix_hip_dprimme(B, m, n, ld, dB, ld, ctx);
   bad |= Num_check_pointer_hip_dprimme(dA);
   printf("check_pointer(host array) = %d (nonzero on a GPU build: not device memory)\n", Num_check_pointer_hip_dprimme(A));

   /* C(host) = 2 A' B - C */
   double *C = malloc(sizeof(double) * k * n), *Cref = malloc(sizeof(double) * k * n);
   for (int i = 0; i < k * n; i++) C[i] = Cref[i] = rnd(&seed);
   bad |= Num_gemm_ddh_hip_dprimme("C", "N", k, n, m, 2.0, dA, ld, dB, ld, -1.0, C, k, ctx);
   double err = 0, scale = 0;
   for (int j = 0; j < n; j++) for (int i = 0; i < k; i++) {
      double t = 0; for (PRIMME_INT r = 0; r < m; r++) t += A[r + i * ld] * B[r + j * ld];
      t = 2.0 * t - Cref[i + j * k];
      err = fmax(err, fabs(t - C[i + j * k])); scale = fmax(scale, fabs(t));
   }
   printf("gemm_ddh   %.2e\n", err / scale); bad |= !(err <= 1e-12 * scale * sqrt((double)m));

   /* dC = B; dC += -0.5 A S (Gram-Schmidt form), then dC = A S (Ritz form) */
   bad |= Num_copy_matrix_hip_dprimme(dB, m, n, ld, dC, ld, ctx);
   bad |= Num_gemm_dhd_hip_dprimme("N", "N", m, n, k, -0.5, dA, ld, S, k, 1.0, dC, ld, ctx);
   double *H = malloc(sizeof(double) * ld * n);
   bad |= Num_get_matrix_hip_dprimme(dC, m, n, ld, H, ld, ctx);
   err = 0;
   for (int j = 0; j < n; j++) for (PRIMME_INT r = 0; r < m; r++) {
      double t = B[r + j * ld]; for (int i = 0; i < k; i++) t -= 0.5 * A[r + i * ld] * S[i + j * k];
      err = fmax(err, fabs(t - H[r + j * ld]));
   }
   printf("gemm_dhd+  %.2e\n", err); bad |= !(err <= 1e-13);
   bad |= Num_gemm_dhd_hip_dprimme("N", "N", m, n, k, 1.0, dA, ld, S, k, 0.0, dC, ld, ctx);
   bad |= Num_get_matrix_hip_dprimme(dC, m, n, ld, H, ld, ctx);
   err = 0;
   for (int j = 0; j < n; j++) for (PRIMME_INT r = 0; r < m; r++) {
      double t = 0; for (int i = 0; i < k; i++) t += A[r + i * ld] * S[i + j * k];
      err = fmax(err, fabs(t - H[r + j * ld]));
   }
   printf("gemm_dhd=  %.2e\n", err); bad |= !(err <= 1e-13);

   /* y(host) = A' x ; y(dev) += A s ; dot ; axpy ; scal ; gramm */
   double y[13], yref;
   bad |= Num_gemv_ddh_hip_dprimme("C", m, k, 1.0, dA, ld, dB, 1, 0.0, y, 1, ctx);
   err = 0;
   for (int i = 0; i < k; i++) { yref = 0; for (PRIMME_INT r = 0; r < m; r++) yref += A[r + i * ld] * B[r]; err = fmax(err, fabs(yref - y[i])); }
   printf("gemv_ddh   %.2e\n", err); bad |= !(err <= 1e-10);
   bad |= Num_copy_matrix_hip_dprimme(dB, m, 1, ld, dC, ld, ctx);
   bad |= Num_gemv_dhd_hip_dprimme("N", m, k, 3.0, dA, ld, S, 1, 1.0, dC, 1, ctx);
   bad |= Num_axpy_hip_dprimme(m, -2.0, dB + ld, 1, dC, 1, ctx);
   bad |= Num_scal_hip_dprimme(m, 0.25, dC, 1, ctx);
   bad |= Num_get_matrix_hip_dprimme(dC, m, 1, ld, H, ld, ctx);
   err = 0;
   double dref = 0;
   for (PRIMME_INT r = 0; r < m; r++) {
      double t = B[r]; for (int i = 0; i < k; i++) t += 3.0 * A[r + i * ld] * S[i];
      t = 0.25 * (t - 2.0 * B[r + ld]);
      err = fmax(err, fabs(t - H[r])); dref += t * B[r + 2 * ld];
   }
   printf("gemv_dhd/axpy/scal %.2e\n", err); bad |= !(err <= 1e-13);
   const double d = Num_dot_hip_dprimme(m, dC, 1, dB + 2 * ld, 1, ctx);
   printf("dot        %.2e\n", fabs(d - dref)); bad |= !(fabs(d - dref) <= 1e-10);
   double G[25];
   for (int i = 0; i < 25; i++) G[i] = 0.0;
   bad |= Num_compute_gramm_ddh_hip_dprimme(dB, m, n, ld, dB, ld, 0.0, G, n, 1, ctx);
   err = 0;
   for (int j = 0; j < n; j++) for (int i = 0; i <= j; i++) {
      double t = 0; for (PRIMME_INT r = 0; r < m; r++) t += B[r + i * ld] * B[r + j * ld];
      err = fmax(err, fabs(t - G[i + j * n]));
   }
   printf("gramm_ddh  %.2e\n", err); bad |= !(err <= 1e-9);
   bad |= Num_zero_matrix_hip_dprimme(dC, m, n, ld, ctx) | Num_get_matrix_hip_dprimme(dC, m, 1, ld, H, ld, ctx);
   bad |= (H[0] != 0.0 || H[m - 1] != 0.0);
   Num_free_hip_dprimme(dA, ctx); Num_free_hip_dprimme(dB, ctx); Num_free_hip_dprimme(dC, ctx);
   hipk_ctx_destroy(ctx);
   printf(bad ? "FAILED\n" : "hip_wrapper shim: returned 0\n");
   return bad ? 1 : 0;
}
