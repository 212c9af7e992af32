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

   /* ---- the routines added in round 3: every operand on the device, trsm_hd, larnv, copy, gramm on the device ---- */
   {
      double *dS, *dG, Sg[13 * 5], Gh[25];
      bad |= Num_malloc_hip_dprimme(k * n, &dS, ctx) | Num_malloc_hip_dprimme(n * n, &dG, ctx);
      /* dS(device, k x n) = A' B */
      bad |= Num_gemm_hip_dprimme("C", "N", k, n, (int)m, 1.0, dA, (int)ld, dB, (int)ld, 0.0, dS, k, ctx);
      bad |= Num_get_matrix_hip_dprimme(dS, k, n, k, Sg, k, ctx);
      err = 0;
      for (int j = 0; j < n; j++) for (int i = 0; i < k; i++) {
         double t = 0; for (PRIMME_INT r = 0; r < m; r++) t += A[r + i * ld] * B[r + j * ld];
         err = fmax(err, fabs(t - Sg[i + j * k]));
      }
      printf("gemm (C,N) %.2e\n", err); bad |= !(err <= 1e-10);
      /* dC = B - A dS  (all on the device) */
      bad |= Num_copy_matrix_hip_dprimme(dB, m, n, ld, dC, ld, ctx);
      bad |= Num_gemm_hip_dprimme("N", "N", (int)m, n, k, -1.0, dA, (int)ld, dS, k, 1.0, dC, (int)ld, ctx);
      bad |= Num_get_matrix_hip_dprimme(dC, m, n, ld, H, ld, ctx);
      err = 0;
      for (int j = 0; j < n; j++) for (PRIMME_INT r = 0; r < m; r++) {
         double t = B[r + j * ld]; for (int i = 0; i < k; i++) t -= A[r + i * ld] * Sg[i + j * k];
         err = fmax(err, fabs(t - H[r + j * ld]));
      }
      printf("gemm (N,N) %.2e\n", err); bad |= !(err <= 1e-11);
      /* y(device, k) = A' B(:,0) through gemv, compared with the first column of dS */
      double *dy, yh[13];
      bad |= Num_malloc_hip_dprimme(k, &dy, ctx);
      bad |= Num_gemv_hip_dprimme("C", m, k, 1.0, dA, (int)ld, dB, 1, 0.0, dy, 1, ctx);
      bad |= Num_get_matrix_hip_dprimme(dy, k, 1, k, yh, k, ctx);
      err = 0; for (int i = 0; i < k; i++) err = fmax(err, fabs(yh[i] - Sg[i]));
      printf("gemv (C)   %.2e\n", err); bad |= !(err <= 1e-12);
      /* Gram matrix on the device, its Cholesky factor on the host, B R^-1 on the device: orthonormal columns */
      bad |= Num_compute_gramm_hip_dprimme(dB, m, n, (int)ld, dB, ld, 0.0, dG, n, 1, 1, ctx);
      bad |= Num_get_matrix_hip_dprimme(dG, n, n, n, Gh, n, ctx);
      double R[25];
      for (int i = 0; i < 25; i++) R[i] = 0.0;
      for (int j = 0; j < n; j++) {                 /* upper Cholesky factor, G = R' R */
         for (int i = 0; i <= j; i++) {
            double t = Gh[i + j * n];
            for (int l = 0; l < i; l++) t -= R[l + i * n] * R[l + j * n];
            R[i + j * n] = (i == j) ? sqrt(t) : t / R[i + i * n];
         }
      }
      bad |= Num_copy_hip_dprimme(ld * n, dB, 1, dC, 1, ctx);
      bad |= Num_trsm_hd_hip_dprimme("R", "U", "N", "N", (int)m, n, 1.0, R, n, dC, (int)ld, ctx);
      bad |= Num_compute_gramm_ddh_hip_dprimme(dC, m, n, ld, dC, ld, 0.0, Gh, n, 1, ctx);
      err = 0; for (int j = 0; j < n; j++) for (int i = 0; i <= j; i++) err = fmax(err, fabs(Gh[i + j * n] - (i == j)));
      printf("trsm_hd    %.2e (|Q'Q - I| after B R^-1)\n", err); bad |= !(err <= 1e-12);
      /* xLARNV stream: entries in (-1, 1), the seed advances, the same seed gives the same numbers */
      PRIMME_INT s1[4] = {1, 2, 3, 5}, s2[4] = {1, 2, 3, 5};
      double r1[64], r2[64];
      bad |= Num_larnv_hip_dprimme(2, s1, 64, dC, ctx) | Num_get_matrix_hip_dprimme(dC, 64, 1, 64, r1, 64, ctx);
      bad |= Num_larnv_hip_dprimme(2, s2, 64, dC, ctx) | Num_get_matrix_hip_dprimme(dC, 64, 1, 64, r2, 64, ctx);
      int same = 1, inside = 1;
      for (int i = 0; i < 64; i++) { same &= (r1[i] == r2[i]); inside &= (r1[i] > -1.0 && r1[i] < 1.0); }
      printf("larnv      same stream %d, inside (-1,1) %d, seed advanced %d\n", same, inside, s1[0] != 1 || s1[1] != 2 || s1[2] != 3 || s1[3] != 5);
      bad |= !(same && inside) || (s1[0] == 1 && s1[1] == 2 && s1[2] == 3 && s1[3] == 5);
      Num_free_hip_dprimme(dS, ctx); Num_free_hip_dprimme(dG, ctx); Num_free_hip_dprimme(dy, ctx);
   }
   /* ---- the single-precision stem: one pass through the mixed forms ---- */
   {
      float *fA = malloc(sizeof(float) * ld * k), *fB = malloc(sizeof(float) * ld * n), *dfA, *dfB, fC[13 * 5];
      for (PRIMME_INT i = 0; i < ld * k; i++) fA[i] = (float)A[i];
      for (PRIMME_INT i = 0; i < ld * n; i++) fB[i] = (float)B[i];
      bad |= Num_malloc_hip_sprimme(ld * k, &dfA, ctx) | Num_malloc_hip_sprimme(ld * n, &dfB, ctx);
      bad |= Num_set_matrix_hip_sprimme(fA, m, k, ld, dfA, ld, ctx) | Num_set_matrix_hip_sprimme(fB, m, n, ld, dfB, ld, ctx);
      bad |= Num_gemm_ddh_hip_sprimme("C", "N", k, n, m, 1.0f, dfA, ld, dfB, ld, 0.0f, fC, k, ctx);
      err = 0; scale = 0;
      for (int j = 0; j < n; j++) for (int i = 0; i < k; i++) {
         double t = 0; for (PRIMME_INT r = 0; r < m; r++) t += (double)fA[r + i * ld] * (double)fB[r + j * ld];
         err = fmax(err, fabs(t - fC[i + j * k])); scale = fmax(scale, fabs(t));
      }
      printf("sgemm_ddh  %.2e\n", err / scale); bad |= !(err <= 1e-5 * scale);
      const float nrm = Num_dot_hip_sprimme(m, dfB, 1, dfB, 1, ctx);
      double want = 0; for (PRIMME_INT r = 0; r < m; r++) want += (double)fB[r] * (double)fB[r];
      printf("sdot       %.2e\n", fabs(nrm - want) / want); bad |= !(fabs(nrm - want) <= 1e-5 * want);
      bad |= Num_scal_hip_sprimme(m, 2.0f, dfB, 1, ctx);
      bad |= Num_get_matrix_hip_sprimme(dfB, m, 1, ld, fA, ld, ctx);
      bad |= (fA[7] != 2.0f * fB[7]);
      /* device double -> device float */
      bad |= Num_copy_Tmatrix_hip_sprimme(dB, primme_op_double, m, 1, ld, dfA, ld, ctx) | Num_get_matrix_hip_sprimme(dfA, m, 1, ld, fA, ld, ctx);
      bad |= (fA[11] != (float)B[11]);
      Num_free_hip_sprimme(dfA, ctx); Num_free_hip_sprimme(dfB, ctx);
      free(fA); free(fB);
   }
   bad |= Num_zero_matrix_hip_dprimme(dC, m, n, ld, ctx) | Num_get_matrix_hip_dprimme(dC, m, 1, ld, H, ld, ctx);
   bad |= (H[0] != 0.0 || H[m - 1] != 0.0);
   Num_free_hip_dprimme(dA, ctx); Num_free_hip_dprimme(dB, ctx); Num_free_hip_dprimme(dC, ctx);
   hipk_ctx_destroy(ctx);
   printf(bad ? "FAILED\n" : "hip_wrapper shim: returned 0\n");
   return bad ? 1 : 0;
}
