/* hip_wrapper_body.h — one precision of the reference's Num_* backend routines; included by hip_wrapper.c once per
 * stem with  WSCALAR (device scalar), WHSCALAR (host scalar), WDT (hipk_dtype) and WN(name) (symbol suffix) defined. */

int WN(Num_check_pointer)(void *x) { return hipk_is_device_ptr(x) ? 0 : -1; }

int WN(Num_malloc)(PRIMME_INT n, WSCALAR **x, hipk_ctx *ctx) {
   void *p = NULL;
   if (hipk_malloc(ctx, sizeof(WSCALAR) * (size_t)(n > 0 ? n : 0), &p)) return PRIMME_MALLOC_FAILURE;
   *x = (WSCALAR *)p;
   return 0;
}
int WN(Num_free)(WSCALAR *x, hipk_ctx *ctx) { return hipk_free(ctx, x) ? PRIMME_MALLOC_FAILURE : 0; }

int WN(Num_set_matrix)(WHSCALAR *x, PRIMME_INT m, PRIMME_INT n, PRIMME_INT ldx, WSCALAR *y, PRIMME_INT ldy, hipk_ctx *ctx) {
   if (m <= 0 || n <= 0) return 0;
   if (ldx == m && ldy == m) WCHK(hipk_h2d(ctx, y, x, sizeof(WSCALAR) * (size_t)m * n));
   else for (PRIMME_INT c = 0; c < n; c++) WCHK(hipk_h2d(ctx, y + c * ldy, x + c * ldx, sizeof(WSCALAR) * (size_t)m));
   WCHK(hipk_sync(ctx));                     /* the host array may be reused by the caller right away */
   return 0;
}
int WN(Num_get_matrix)(WSCALAR *x, PRIMME_INT m, PRIMME_INT n, PRIMME_INT ldx, WHSCALAR *y, PRIMME_INT ldy, hipk_ctx *ctx) {
   if (m <= 0 || n <= 0) return 0;
   if (ldx == m && ldy == m) WCHK(hipk_d2h(ctx, y, x, sizeof(WSCALAR) * (size_t)m * n));
   else for (PRIMME_INT c = 0; c < n; c++) WCHK(hipk_d2h(ctx, y + c * ldy, x + c * ldx, sizeof(WSCALAR) * (size_t)m));
   WCHK(hipk_sync(ctx));
   return 0;
}
int WN(Num_copy_matrix)(WSCALAR *x, PRIMME_INT m, PRIMME_INT n, PRIMME_INT ldx, WSCALAR *y, PRIMME_INT ldy, hipk_ctx *ctx) {
   if (x == y && ldx == ldy) return 0;
   if (m <= 0 || n <= 0) return 0;
   WCHK(hipk_copy_cols(ctx, WDT, m, x, ldx, y, ldy, (int)n));
   return 0;
}
/* a matrix of another precision into this one's: only the identity conversion lives on the device (the reference
 * converts through a temporary too, cublas_wrapper.c:254-309) */
int WN(Num_copy_Tmatrix)(void *x, primme_op_datatype xt, PRIMME_INT m, PRIMME_INT n, PRIMME_INT ldx, WSCALAR *y, PRIMME_INT ldy,
      hipk_ctx *ctx) {
   if (xt == primme_op_default || xt == WOPT) return WN(Num_copy_matrix)((WSCALAR *)x, m, n, ldx, y, ldy, ctx);
   if (m <= 0 || n <= 0) return 0;
   if (xt != primme_op_double && xt != primme_op_float) return PRIMME_FUNCTION_UNAVAILABLE;
   /* the other real precision: through the host, column by column */
   const size_t xs = (xt == primme_op_double) ? 8 : 4;
   char *hx = (char *)malloc(xs * (size_t)m);
   WSCALAR *hy = (WSCALAR *)malloc(sizeof(WSCALAR) * (size_t)m);
   int rc = (hx && hy) ? 0 : PRIMME_MALLOC_FAILURE;
   for (PRIMME_INT c = 0; c < n && !rc; c++) {
      rc = hipk_d2h(ctx, hx, (char *)x + xs * (size_t)c * (size_t)ldx, xs * (size_t)m);
      if (!rc) rc = hipk_sync(ctx);
      for (PRIMME_INT i = 0; i < m && !rc; i++) hy[i] = (xt == primme_op_double) ? (WSCALAR)((double *)hx)[i] : (WSCALAR)((float *)hx)[i];
      if (!rc) rc = hipk_h2d(ctx, y + c * ldy, hy, sizeof(WSCALAR) * (size_t)m);
      if (!rc) rc = hipk_sync(ctx);
   }
   free(hx); free(hy);
   return rc ? (rc < 0 ? rc : PRIMME_UNEXPECTED_FAILURE) : 0;
}
int WN(Num_copy)(PRIMME_INT n, WSCALAR *x, int incx, WSCALAR *y, int incy, hipk_ctx *ctx) {
   if (incx != 1 || incy != 1) return PRIMME_FUNCTION_UNAVAILABLE;
   return WN(Num_copy_matrix)(x, n, 1, n, y, n, ctx);
}
int WN(Num_zero_matrix)(WSCALAR *x, PRIMME_INT m, PRIMME_INT n, PRIMME_INT ldx, hipk_ctx *ctx) {
   if (m <= 0 || n <= 0) return 0;
   if (ldx == m) WCHK(hipk_memset0(ctx, x, sizeof(WSCALAR) * (size_t)m * n));
   else for (PRIMME_INT c = 0; c < n; c++) WCHK(hipk_memset0(ctx, x + c * ldx, sizeof(WSCALAR) * (size_t)m));
   return 0;
}

/* TN panel: one launch of the inner-product kernel (+ its fixed-order second stage), one download; the accumulators
 * of the device layer are doubles in every precision */
static int WN(tn_panel)(WSCALAR *a, PRIMME_INT lda, int m, WSCALAR *b, PRIMME_INT ldb, int n, PRIMME_INT k, double alpha,
      double beta, WHSCALAR *c, int ldc, hipk_ctx *ctx) {
   if (m == 0 || n == 0) return 0;
   double *d = NULL, *h = (double *)malloc(sizeof(double) * (size_t)m * n);
   void *dv = NULL;
   if (!h) return PRIMME_MALLOC_FAILURE;
   if (hipk_malloc(ctx, sizeof(double) * (size_t)m * n, &dv)) { free(h); return PRIMME_MALLOC_FAILURE; }
   d = (double *)dv;
   hipk_seg seg = {a, lda, m};
   int rc = 0;
   if (k > 0) rc = hipk_panel_dots(ctx, WDT, k, &seg, 1, b, ldb, n, d, m);
   else rc = hipk_memset0(ctx, d, sizeof(double) * (size_t)m * n);
   if (!rc) rc = hipk_d2h(ctx, h, d, sizeof(double) * (size_t)m * n);
   if (!rc) rc = hipk_sync(ctx);
   if (!rc)
      for (int j = 0; j < n; j++)
         for (int i = 0; i < m; i++)
            c[i + (size_t)j * ldc] = (WHSCALAR)(alpha * h[i + (size_t)j * m] + (beta != 0.0 ? beta * (double)c[i + (size_t)j * ldc] : 0.0));
   hipk_free(ctx, d);
   free(h);
   return rc ? (rc < 0 ? rc : PRIMME_UNEXPECTED_FAILURE) : 0;
}

int WN(Num_gemm_ddh)(const char *transa, const char *transb, int m, int n, PRIMME_INT k, WHSCALAR alpha, WSCALAR *a,
      PRIMME_INT lda, WSCALAR *b, PRIMME_INT ldb, WHSCALAR beta, WHSCALAR *c, int ldc, hipk_ctx *ctx) {
   if (is_n(transa) || !is_n(transb)) return PRIMME_FUNCTION_UNAVAILABLE;   /* the solver only forms A' B this way */
   return WN(tn_panel)(a, lda, m, b, ldb, n, k, (double)alpha, (double)beta, c, ldc, ctx);
}

/* NN panel with the small factor on the host */
int WN(Num_gemm_dhd)(const char *transa, const char *transb, PRIMME_INT m, int n, int k, WHSCALAR alpha, WSCALAR *a,
      PRIMME_INT lda, WHSCALAR *b, int ldb, WHSCALAR beta, WSCALAR *c, PRIMME_INT ldc, hipk_ctx *ctx) {
   if (!is_n(transa) || !is_n(transb)) return PRIMME_FUNCTION_UNAVAILABLE;
   if (m == 0 || n == 0) return 0;
   if (k == 0) {
      if (beta == 0.0) return WN(Num_zero_matrix)(c, m, n, ldc, ctx);
      if (beta == 1.0) return 0;
      const double bd = (double)beta;
      for (int j = 0; j < n; j++) WCHK(hipk_scale_cols(ctx, WDT, m, c + (size_t)j * ldc, ldc, 1, &bd));
      return 0;
   }
   double *hb = (double *)malloc(sizeof(double) * (size_t)k * n), *db = NULL;
   void *dv = NULL;
   if (!hb) return PRIMME_MALLOC_FAILURE;
   if (hipk_malloc(ctx, sizeof(double) * (size_t)k * n, &dv)) { free(hb); return PRIMME_MALLOC_FAILURE; }
   db = (double *)dv;
   int rc = 0;
   if (beta == 1.0) {
      /* C += alpha A B: the Gram-Schmidt update (hipk_panel_project subtracts, so the factor carries -alpha) */
      for (int j = 0; j < n; j++) for (int i = 0; i < k; i++) hb[i + (size_t)j * k] = -(double)alpha * (double)b[i + (size_t)j * ldb];
      hipk_seg seg = {a, lda, k};
      rc = hipk_h2d(ctx, db, hb, sizeof(double) * (size_t)k * n);
      if (!rc) rc = hipk_panel_project(ctx, WDT, m, &seg, 1, db, k, c, ldc, n, NULL);
   } else if (beta == 0.0 && k <= 255 && n <= HIPK_MAX_JOBS) {
      /* C = alpha A B: the Ritz-vector product (row-wise: C may alias columns of A) */
      for (int j = 0; j < n; j++) for (int i = 0; i < k; i++) hb[i + (size_t)j * k] = (double)alpha * (double)b[i + (size_t)j * ldb];
      hipk_job jobs[HIPK_MAX_JOBS];
      for (int j = 0; j < n; j++) { jobs[j].kind = HIPK_JOB_XV; jobs[j].col = j; jobs[j].dst = c + (size_t)j * ldc; jobs[j].slot = -1; }
      rc = hipk_h2d(ctx, db, hb, sizeof(double) * (size_t)k * n);
      if (!rc) rc = hipk_ritz_update(ctx, WDT, m, a, a, lda, k, db, k, NULL, jobs, n, NULL);
   } else rc = PRIMME_FUNCTION_UNAVAILABLE;
   if (!rc) rc = hipk_sync(ctx);             /* hb / db are released below */
   hipk_free(ctx, db);
   free(hb);
   return rc ? (rc < 0 ? rc : PRIMME_UNEXPECTED_FAILURE) : 0;
}

/* every operand on the device (cublas_wrapper.c:397): the small one takes the round trip of the mixed forms above —
 * 'C','N': C (m x n, small) = alpha A' B + beta C;  'N','N': C (m x n panel) = alpha A B (k x n, small) + beta C */
int WN(Num_gemm)(const char *transa, const char *transb, int m, int n, int k, WHSCALAR alpha, WSCALAR *a, int lda, WSCALAR *b,
      int ldb, WHSCALAR beta, WSCALAR *c, int ldc, hipk_ctx *ctx) {
   if (m == 0 || n == 0) return 0;
   if (!is_n(transb)) return PRIMME_FUNCTION_UNAVAILABLE;
   int rc;
   if (!is_n(transa)) {
      WHSCALAR *hc = (WHSCALAR *)calloc((size_t)m * n, sizeof(WHSCALAR));
      if (!hc) return PRIMME_MALLOC_FAILURE;
      rc = (beta != 0.0) ? WN(Num_get_matrix)(c, m, n, ldc, hc, m, ctx) : 0;
      if (!rc) rc = WN(tn_panel)(a, lda, m, b, ldb, n, k, (double)alpha, (double)beta, hc, m, ctx);
      if (!rc) rc = WN(Num_set_matrix)(hc, m, n, m, c, ldc, ctx);
      free(hc);
   } else {
      WHSCALAR *hb = (WHSCALAR *)calloc((size_t)(k > 0 ? k : 1) * n, sizeof(WHSCALAR));
      if (!hb) return PRIMME_MALLOC_FAILURE;
      rc = k > 0 ? WN(Num_get_matrix)(b, k, n, ldb, hb, k, ctx) : 0;
      if (!rc) rc = WN(Num_gemm_dhd)(transa, transb, m, n, k, alpha, a, lda, hb, k > 0 ? k : 1, beta, c, ldc, ctx);
      free(hb);
   }
   return rc;
}

int WN(Num_gemv_ddh)(const char *transa, PRIMME_INT m, int n, WHSCALAR alpha, WSCALAR *a, PRIMME_INT lda, WSCALAR *x,
      int incx, WHSCALAR beta, WHSCALAR *y, int incy, hipk_ctx *ctx) {
   if (is_n(transa) || incx != 1 || incy != 1) return PRIMME_FUNCTION_UNAVAILABLE;
   return WN(tn_panel)(a, lda, n, x, m, 1, m, (double)alpha, (double)beta, y, n > 0 ? n : 1, ctx);
}
int WN(Num_gemv_dhd)(const char *transa, PRIMME_INT m, int n, WHSCALAR alpha, WSCALAR *a, PRIMME_INT lda, WHSCALAR *x,
      int incx, WHSCALAR beta, WSCALAR *y, int incy, hipk_ctx *ctx) {
   if (!is_n(transa) || incx != 1 || incy != 1) return PRIMME_FUNCTION_UNAVAILABLE;
   return WN(Num_gemm_dhd)("N", "N", m, 1, n, alpha, a, lda, x, n > 0 ? n : 1, beta, y, m, ctx);
}
/* all on the device (cublas_wrapper.c:507): y = alpha op(A) x + beta y */
int WN(Num_gemv)(const char *transa, PRIMME_INT m, int n, WHSCALAR alpha, WSCALAR *a, int lda, WSCALAR *x, int incx, WHSCALAR beta,
      WSCALAR *y, int incy, hipk_ctx *ctx) {
   if (incx != 1 || incy != 1) return PRIMME_FUNCTION_UNAVAILABLE;
   if (is_n(transa)) return WN(Num_gemm)("N", "N", (int)m, 1, n, alpha, a, lda, x, n > 0 ? n : 1, beta, y, (int)m, ctx);
   return WN(Num_gemm)("C", "N", n, 1, (int)m, alpha, a, lda, x, (int)m, beta, y, n > 0 ? n : 1, ctx);
}

int WN(Num_axpy)(PRIMME_INT n, WHSCALAR alpha, WSCALAR *x, int incx, WSCALAR *y, int incy, hipk_ctx *ctx) {
   if (incx != 1 || incy != 1) return PRIMME_FUNCTION_UNAVAILABLE;
   const double ad = (double)alpha;
   WCHK(hipk_axpy_cols(ctx, WDT, n, &ad, x, n, y, n, 1));
   return 0;
}
WHSCALAR WN(Num_dot)(PRIMME_INT n, WSCALAR *x, int incx, WSCALAR *y, int incy, hipk_ctx *ctx) {
   double r = 0.0;
   void *d = NULL;
   if (incx != 1 || incy != 1 || n <= 0) return 0;
   if (hipk_malloc(ctx, sizeof(double), &d)) return 0;
   if (!hipk_pair_dots(ctx, WDT, n, x, n, y, n, 1, (double *)d) && !hipk_d2h(ctx, &r, d, sizeof(double))) hipk_sync(ctx);
   hipk_free(ctx, d);
   return (WHSCALAR)r;
}
int WN(Num_scal)(PRIMME_INT n, WHSCALAR alpha, WSCALAR *x, int incx, hipk_ctx *ctx) {
   if (incx != 1) return PRIMME_FUNCTION_UNAVAILABLE;
   const double ad = (double)alpha;
   WCHK(hipk_scale_cols(ctx, WDT, n, x, n, 1, &ad));
   return 0;
}
/* random numbers: LAPACK's xLARNV stream generated on the host and uploaded, as the reference's GPU backends do
 * (cublas_wrapper.c:707-736); idist 1 = uniform (0,1), 2 = uniform (-1,1) */
int WN(Num_larnv)(int idist, PRIMME_INT *iseed, PRIMME_INT length, WSCALAR *x, hipk_ctx *ctx) {
   if (length <= 0) return 0;
   if (idist != 1 && idist != 2) return PRIMME_FUNCTION_UNAVAILABLE;
   double *h = (double *)malloc(sizeof(double) * (size_t)length);
   WSCALAR *hs = (WSCALAR *)malloc(sizeof(WSCALAR) * (size_t)length);
   if (!h || !hs) { free(h); free(hs); return PRIMME_MALLOC_FAILURE; }
   int64_t seed[4] = {iseed[0], iseed[1], iseed[2], iseed[3]};
   pa_larnv_uniform11(seed, length, h);
   for (PRIMME_INT i = 0; i < length; i++) hs[i] = (WSCALAR)(idist == 2 ? h[i] : 0.5 * (h[i] + 1.0));
   for (int i = 0; i < 4; i++) iseed[i] = seed[i];
   int rc = hipk_h2d(ctx, x, hs, sizeof(WSCALAR) * (size_t)length);
   if (!rc) rc = hipk_sync(ctx);
   free(h); free(hs);
   return rc ? PRIMME_UNEXPECTED_FAILURE : 0;
}
/* B (device, m x n) = alpha B op(A)^-1 with A (host, n x n) triangular: side 'R' only, which is all the solver asks
 * of the mixed form (the Cholesky factor of CholQR applied to the block, ortho.c:1041-1047).  The inverse of the
 * small factor is formed on the host and applied by the one-pass right-multiplication kernel. */
int WN(Num_trsm_hd)(const char *side, const char *uplo, const char *transa, const char *diag, int m, int n, WHSCALAR alpha,
      WHSCALAR *a, int lda, WSCALAR *b, int ldb, hipk_ctx *ctx) {
   if (m == 0 || n == 0) return 0;
   if ((*side != 'R' && *side != 'r') || n > 8) return PRIMME_FUNCTION_UNAVAILABLE;
   const int upper = (*uplo == 'U' || *uplo == 'u'), unit = (*diag == 'U' || *diag == 'u'), tr = !is_n(transa);
   /* T = op(A) as an upper or lower triangle, then X = alpha T^-1 by substitution on the identity */
   double T[64], X[64];
   for (int j = 0; j < n; j++)
      for (int i = 0; i < n; i++) {
         const int stored = upper ? (i <= j) : (i >= j);
         const double v = (i == j && unit) ? 1.0 : (stored ? (double)a[i + (size_t)j * lda] : 0.0);
         if (tr) T[j + i * n] = v; else T[i + j * n] = v;
      }
   const int up = tr ? !upper : upper;
   for (int c = 0; c < n; c++) {
      double e[8];
      for (int i = 0; i < n; i++) e[i] = (i == c) ? (double)alpha : 0.0;
      if (up) for (int i = n - 1; i >= 0; i--) { double s = e[i]; for (int l = i + 1; l < n; l++) s -= T[i + l * n] * X[l + c * n]; X[i + c * n] = s / T[i + i * n]; }
      else    for (int i = 0; i < n; i++)      { double s = e[i]; for (int l = 0; l < i; l++)     s -= T[i + l * n] * X[l + c * n]; X[i + c * n] = s / T[i + i * n]; }
   }
   void *dM = NULL;
   if (hipk_malloc(ctx, sizeof(double) * 64, &dM)) return PRIMME_MALLOC_FAILURE;
   int rc = hipk_h2d(ctx, dM, X, sizeof(double) * (size_t)n * n);
   hipk_seg none = {b, ldb, 0};               /* no basis columns: only the right multiplication */
   if (!rc) rc = hipk_panel_project_mul(ctx, WDT, m, &none, 1, (const double *)dM, 1, (const double *)dM, b, ldb, n);
   if (!rc) rc = hipk_sync(ctx);
   hipk_free(ctx, dM);
   return rc ? (rc < 0 ? rc : PRIMME_UNEXPECTED_FAILURE) : 0;
}
/* H (n x n) = X' Y + alpha H; H on the device (cublas_wrapper.c:898) or on the host (:962) */
int WN(Num_compute_gramm_ddh)(WSCALAR *X, PRIMME_INT m, int n, PRIMME_INT ldX, WSCALAR *Y, PRIMME_INT ldY, WHSCALAR alpha,
      WHSCALAR *H, int ldH, int isherm, hipk_ctx *ctx) {
   (void)isherm;                             /* the whole block is formed in one launch either way */
   return WN(tn_panel)(X, ldX, n, Y, ldY, n, m, 1.0, (double)alpha, H, ldH, ctx);
}
int WN(Num_compute_gramm)(WSCALAR *X, PRIMME_INT m, int n, int ldX, WSCALAR *Y, PRIMME_INT ldY, WHSCALAR alpha, WSCALAR *H,
      int ldH, int isherm, int deep, hipk_ctx *ctx) {
   (void)deep;
   if (n == 0) return 0;
   WHSCALAR *hh = (WHSCALAR *)calloc((size_t)n * n, sizeof(WHSCALAR));
   if (!hh) return PRIMME_MALLOC_FAILURE;
   int rc = (alpha != 0.0) ? WN(Num_get_matrix)(H, n, n, ldH, hh, n, ctx) : 0;
   if (!rc) rc = WN(Num_compute_gramm_ddh)(X, m, n, ldX, Y, ldY, alpha, hh, n, isherm, ctx);
   if (!rc) rc = WN(Num_set_matrix)(hh, n, n, n, H, ldH, ctx);
   free(hh);
   return rc;
}
