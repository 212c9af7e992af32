/* hip_wrapper.c — the reference's Num_* backend routines (src/linalg/cublas_wrapper.c:162-987) as
 * forwards over the fused device layer; see include/primme_amd_wrapper.h.  Like the reference's GPU
 * backends each call synchronises when an operand lives on the host; the solver of this library does
 * not go through these (it chains the fused launches itself). */
#include <stdlib.h>
#include <string.h>
#include "primme_amd_wrapper.h"

#define WCHK(call) do { int rc_ = (call); if (rc_) return rc_ < 0 ? rc_ : PRIMME_UNEXPECTED_FAILURE; } while (0)
static int is_n(const char *t) { return *t == 'N' || *t == 'n'; }

int Num_check_pointer_hip_dprimme(void *x) { return hipk_is_device_ptr(x) ? 0 : -1; }

int Num_malloc_hip_dprimme(PRIMME_INT n, double **x, hipk_ctx *ctx) {
   void *p = NULL;
   if (hipk_malloc(ctx, sizeof(double) * (size_t)(n > 0 ? n : 0), &p)) return PRIMME_MALLOC_FAILURE;
   *x = (double *)p;
   return 0;
}
int Num_free_hip_dprimme(double *x, hipk_ctx *ctx) { return hipk_free(ctx, x) ? PRIMME_MALLOC_FAILURE : 0; }

int Num_set_matrix_hip_dprimme(double *x, PRIMME_INT m, PRIMME_INT n, PRIMME_INT ldx, double *y, PRIMME_INT ldy, hipk_ctx *ctx) {
   if (m <= 0 || n <= 0) return 0;
   if (ldx == m && ldy == m) WCHK(hipk_h2d(ctx, y, x, sizeof(double) * (size_t)m * n));
   else for (PRIMME_INT c = 0; c < n; c++) WCHK(hipk_h2d(ctx, y + c * ldy, x + c * ldx, sizeof(double) * (size_t)m));
   WCHK(hipk_sync(ctx));                     /* the host array may be reused by the caller right away */
   return 0;
}
int Num_get_matrix_hip_dprimme(double *x, PRIMME_INT m, PRIMME_INT n, PRIMME_INT ldx, double *y, PRIMME_INT ldy, hipk_ctx *ctx) {
   if (m <= 0 || n <= 0) return 0;
   if (ldx == m && ldy == m) WCHK(hipk_d2h(ctx, y, x, sizeof(double) * (size_t)m * n));
   else for (PRIMME_INT c = 0; c < n; c++) WCHK(hipk_d2h(ctx, y + c * ldy, x + c * ldx, sizeof(double) * (size_t)m));
   WCHK(hipk_sync(ctx));
   return 0;
}
int Num_copy_matrix_hip_dprimme(double *x, PRIMME_INT m, PRIMME_INT n, PRIMME_INT ldx, double *y, PRIMME_INT ldy, hipk_ctx *ctx) {
   if (x == y && ldx == ldy) return 0;
   WCHK(hipk_copy_cols(ctx, HIPK_F64, m, x, ldx, y, ldy, (int)n));
   return 0;
}
int Num_zero_matrix_hip_dprimme(double *x, PRIMME_INT m, PRIMME_INT n, PRIMME_INT ldx, hipk_ctx *ctx) {
   if (m <= 0 || n <= 0) return 0;
   if (ldx == m) WCHK(hipk_memset0(ctx, x, sizeof(double) * (size_t)m * n));
   else for (PRIMME_INT c = 0; c < n; c++) WCHK(hipk_memset0(ctx, x + c * ldx, sizeof(double) * (size_t)m));
   return 0;
}

/* TN panel: one launch of the inner-product kernel (+ its fixed-order second stage), one download */
static int tn_panel(double *a, PRIMME_INT lda, int m, double *b, PRIMME_INT ldb, int n, PRIMME_INT k, double alpha,
      double beta, double *c, int ldc, hipk_ctx *ctx) {
   if (m == 0 || n == 0) return 0;
   double *d = NULL, *h = (double *)malloc(sizeof(double) * (size_t)m * n);
   if (!h) return PRIMME_MALLOC_FAILURE;
   if (Num_malloc_hip_dprimme((PRIMME_INT)m * n, &d, ctx)) { free(h); return PRIMME_MALLOC_FAILURE; }
   hipk_seg seg = {a, lda, m};
   int rc = 0;
   if (k > 0) rc = hipk_panel_dots(ctx, HIPK_F64, k, &seg, 1, b, ldb, n, d, m);
   else rc = hipk_memset0(ctx, d, sizeof(double) * (size_t)m * n);
   if (!rc) rc = hipk_d2h(ctx, h, d, sizeof(double) * (size_t)m * n);
   if (!rc) rc = hipk_sync(ctx);
   if (!rc)
      for (int j = 0; j < n; j++)
         for (int i = 0; i < m; i++)
            c[i + (size_t)j * ldc] = alpha * h[i + (size_t)j * m] + (beta != 0.0 ? beta * c[i + (size_t)j * ldc] : 0.0);
   Num_free_hip_dprimme(d, ctx);
   free(h);
   return rc ? (rc < 0 ? rc : PRIMME_UNEXPECTED_FAILURE) : 0;
}

int Num_gemm_ddh_hip_dprimme(const char *transa, const char *transb, int m, int n, PRIMME_INT k, double alpha, double *a,
      PRIMME_INT lda, double *b, PRIMME_INT ldb, double beta, double *c, int ldc, hipk_ctx *ctx) {
   if (is_n(transa) || !is_n(transb)) return PRIMME_FUNCTION_UNAVAILABLE;   /* the solver only forms A' B this way */
   return tn_panel(a, lda, m, b, ldb, n, k, alpha, beta, c, ldc, ctx);
}

/* NN panel with the small factor on the host */
int Num_gemm_dhd_hip_dprimme(const char *transa, const char *transb, PRIMME_INT m, int n, int k, double alpha, double *a,
      PRIMME_INT lda, double *b, int ldb, double beta, double *c, PRIMME_INT ldc, hipk_ctx *ctx) {
   if (!is_n(transa) || !is_n(transb)) return PRIMME_FUNCTION_UNAVAILABLE;
   if (m == 0 || n == 0) return 0;
   if (k == 0) {
      if (beta == 0.0) return Num_zero_matrix_hip_dprimme(c, m, n, ldc, ctx);
      if (beta == 1.0) return 0;
      for (int j = 0; j < n; j++) WCHK(hipk_scale_cols(ctx, HIPK_F64, m, c + (size_t)j * ldc, ldc, 1, &beta));
      return 0;
   }
   double *hb = (double *)malloc(sizeof(double) * (size_t)k * n), *db = NULL;
   if (!hb) return PRIMME_MALLOC_FAILURE;
   if (Num_malloc_hip_dprimme((PRIMME_INT)k * n, &db, ctx)) { free(hb); return PRIMME_MALLOC_FAILURE; }
   int rc = 0;
   if (beta == 1.0) {
      /* C += alpha A B: the Gram-Schmidt update (hipk_panel_project subtracts, so the factor carries -alpha) */
      for (int j = 0; j < n; j++) for (int i = 0; i < k; i++) hb[i + (size_t)j * k] = -alpha * b[i + (size_t)j * ldb];
      hipk_seg seg = {a, lda, k};
      rc = hipk_h2d(ctx, db, hb, sizeof(double) * (size_t)k * n);
      if (!rc) rc = hipk_panel_project(ctx, HIPK_F64, m, &seg, 1, db, k, c, ldc, n, NULL);
   } else if (beta == 0.0 && k <= 255 && n <= HIPK_MAX_JOBS) {
      /* C = alpha A B: the Ritz-vector product (row-wise: C may alias columns of A) */
      for (int j = 0; j < n; j++) for (int i = 0; i < k; i++) hb[i + (size_t)j * k] = alpha * b[i + (size_t)j * ldb];
      hipk_job jobs[HIPK_MAX_JOBS];
      for (int j = 0; j < n; j++) { jobs[j].kind = HIPK_JOB_XV; jobs[j].col = j; jobs[j].dst = c + (size_t)j * ldc; jobs[j].slot = -1; }
      rc = hipk_h2d(ctx, db, hb, sizeof(double) * (size_t)k * n);
      if (!rc) rc = hipk_ritz_update(ctx, HIPK_F64, m, a, a, lda, k, db, k, NULL, jobs, n, NULL);
   } else rc = PRIMME_FUNCTION_UNAVAILABLE;
   if (!rc) rc = hipk_sync(ctx);             /* hb / db are released below */
   Num_free_hip_dprimme(db, ctx);
   free(hb);
   return rc ? (rc < 0 ? rc : PRIMME_UNEXPECTED_FAILURE) : 0;
}

int Num_gemv_ddh_hip_dprimme(const char *transa, PRIMME_INT m, int n, double alpha, double *a, PRIMME_INT lda, double *x,
      int incx, double beta, double *y, int incy, hipk_ctx *ctx) {
   if (is_n(transa) || incx != 1 || incy != 1) return PRIMME_FUNCTION_UNAVAILABLE;
   return tn_panel(a, lda, n, x, m, 1, m, alpha, beta, y, n > 0 ? n : 1, ctx);
}
int Num_gemv_dhd_hip_dprimme(const char *transa, PRIMME_INT m, int n, double alpha, double *a, PRIMME_INT lda, double *x,
      int incx, double beta, double *y, int incy, hipk_ctx *ctx) {
   if (!is_n(transa) || incx != 1 || incy != 1) return PRIMME_FUNCTION_UNAVAILABLE;
   return Num_gemm_dhd_hip_dprimme("N", "N", m, 1, n, alpha, a, lda, x, n > 0 ? n : 1, beta, y, m, ctx);
}

int Num_axpy_hip_dprimme(PRIMME_INT n, double alpha, double *x, int incx, double *y, int incy, hipk_ctx *ctx) {
   if (incx != 1 || incy != 1) return PRIMME_FUNCTION_UNAVAILABLE;
   WCHK(hipk_axpy_cols(ctx, HIPK_F64, n, &alpha, x, n, y, n, 1));
   return 0;
}
double Num_dot_hip_dprimme(PRIMME_INT n, double *x, int incx, double *y, int incy, hipk_ctx *ctx) {
   double r = 0.0, *d = NULL;
   if (incx != 1 || incy != 1 || n <= 0) return 0.0;
   if (Num_malloc_hip_dprimme(1, &d, ctx)) return 0.0;
   if (!hipk_pair_dots(ctx, HIPK_F64, n, x, n, y, n, 1, d) && !hipk_d2h(ctx, &r, d, sizeof(double))) hipk_sync(ctx);
   Num_free_hip_dprimme(d, ctx);
   return r;
}
int Num_scal_hip_dprimme(PRIMME_INT n, double alpha, double *x, int incx, hipk_ctx *ctx) {
   if (incx != 1) return PRIMME_FUNCTION_UNAVAILABLE;
   WCHK(hipk_scale_cols(ctx, HIPK_F64, n, x, n, 1, &alpha));
   return 0;
}
int Num_compute_gramm_ddh_hip_dprimme(double *X, PRIMME_INT m, int n, PRIMME_INT ldX, double *Y, PRIMME_INT ldY, double alpha,
      double *H, int ldH, int isherm, hipk_ctx *ctx) {
   (void)isherm;                             /* the whole block is formed in one launch either way */
   return tn_panel(X, ldX, n, Y, ldY, n, m, 1.0, alpha, H, ldH, ctx);
}
