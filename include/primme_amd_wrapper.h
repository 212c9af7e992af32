/* primme_amd_wrapper.h — the reference's numerical-backend boundary (B2) spelled with
 * the reference's own routine names: what a `src/linalg/hip_wrapper.c` inside the reference tree would
 * contain next to blaslapack.c / cublas_wrapper.c / magma_wrapper.c.  Each function has the argument
 * list of the reference's Num_<op>_Sprimme (src/linalg/cublas_wrapper.c, line cited per function) for
 * the two real GPU instantiations (21 routines each: everything cublas_wrapper.c defines); the one difference is the last argument: the reference
 * passes its `primme_context` by value and finds the device handle in ctx.queue
 * (src/include/common.h:632-633), this C ABI takes the hipk_ctx* itself.
 *
 * Mixed host/device operands are encoded in the name as in the reference: _ddh = A, B on the device and
 * C on the host; _dhd = A and C on the device, B on the host.  Leading dimensions in elements,
 * column-major, increments must be 1.  Return 0 or a PRIMME error code (PRIMME_FUNCTION_UNAVAILABLE for
 * operand shapes outside the tall-skinny cases the solver uses).
 *
 * These are convenience forwards over the fused device layer (primme_amd_kernels.h); the solver in
 * this library does NOT call them — it launches the fused kernels directly (one synchronisation per
 * Gram-Schmidt pass instead of one per BLAS call).
 */
#ifndef PRIMME_AMD_WRAPPER_H
#define PRIMME_AMD_WRAPPER_H
#include "primme_amd.h"
#include "primme_amd_kernels.h"
#ifdef __cplusplus
extern "C" {
#endif
/* ---- double precision (the reference's _dprimme instantiation) ---- */
int Num_check_pointer_hip_dprimme(void *x);                                                      /* cublas_wrapper.c:162 */
int Num_malloc_hip_dprimme(PRIMME_INT n, double **x, hipk_ctx *ctx);                                   /* :187 */
int Num_free_hip_dprimme(double *x, hipk_ctx *ctx);                                                    /* :218 */
int Num_copy_Tmatrix_hip_dprimme(void *x, primme_op_datatype xt, PRIMME_INT m, PRIMME_INT n,      /* :254  device x of type xt -> device y */
      PRIMME_INT ldx, double *y, PRIMME_INT ldy, hipk_ctx *ctx);
int Num_copy_hip_dprimme(PRIMME_INT n, double *x, int incx, double *y, int incy, hipk_ctx *ctx);            /* :312 */
int Num_set_matrix_hip_dprimme(double *x, PRIMME_INT m, PRIMME_INT n, PRIMME_INT ldx, double *y,            /* :335  host x -> device y */
      PRIMME_INT ldy, hipk_ctx *ctx);
int Num_get_matrix_hip_dprimme(double *x, PRIMME_INT m, PRIMME_INT n, PRIMME_INT ldx, double *y,            /* :370  device x -> host y */
      PRIMME_INT ldy, hipk_ctx *ctx);
int Num_copy_matrix_hip_dprimme(double *x, PRIMME_INT m, PRIMME_INT n, PRIMME_INT ldx, double *y,           /* :739  device -> device */
      PRIMME_INT ldy, hipk_ctx *ctx);
int Num_zero_matrix_hip_dprimme(double *x, PRIMME_INT m, PRIMME_INT n, PRIMME_INT ldx, hipk_ctx *ctx); /* :768 */
/* every operand on the device: 'C','N' (C small) and 'N','N' (B small) */
int Num_gemm_hip_dprimme(const char *transa, const char *transb, int m, int n, int k, double alpha,    /* :397 */
      double *a, int lda, double *b, int ldb, double beta, double *c, int ldc, hipk_ctx *ctx);
/* C(host, m x n) = alpha * A' B + beta * C with A (k x m) and B (k x n) on the device: transa 'C'/'T', transb 'N' */
int Num_gemm_ddh_hip_dprimme(const char *transa, const char *transb, int m, int n, PRIMME_INT k, double alpha,   /* :479 */
      double *a, PRIMME_INT lda, double *b, PRIMME_INT ldb, double beta, double *c, int ldc, hipk_ctx *ctx);
/* C(device, m x n) = alpha * A B + beta * C with A (m x k) on the device and B (k x n) on the host: 'N','N';
 * (alpha, beta) = (x, 1) is the Gram-Schmidt update, (x, 0) the Ritz-vector product */
int Num_gemm_dhd_hip_dprimme(const char *transa, const char *transb, PRIMME_INT m, int n, int k, double alpha,   /* :452 */
      double *a, PRIMME_INT lda, double *b, int ldb, double beta, double *c, PRIMME_INT ldc, hipk_ctx *ctx);
int Num_gemv_hip_dprimme(const char *transa, PRIMME_INT m, int n, double alpha, double *a, int lda, double *x,   /* :507  all on the device */
      int incx, double beta, double *y, int incy, hipk_ctx *ctx);
/* y(host) = alpha * A' x + beta * y, A (m x n) and x on the device: transa 'C'/'T' */
int Num_gemv_ddh_hip_dprimme(const char *transa, PRIMME_INT m, int n, double alpha, double *a, PRIMME_INT lda,        /* :566 */
      double *x, int incx, double beta, double *y, int incy, hipk_ctx *ctx);
/* y(device) = alpha * A x + beta * y, A (m x n) on the device, x on the host: transa 'N' */
int Num_gemv_dhd_hip_dprimme(const char *transa, PRIMME_INT m, int n, double alpha, double *a, PRIMME_INT lda,        /* :594 */
      double *x, int incx, double beta, double *y, int incy, hipk_ctx *ctx);
int Num_axpy_hip_dprimme(PRIMME_INT n, double alpha, double *x, int incx, double *y, int incy, hipk_ctx *ctx);   /* :616 */
double Num_dot_hip_dprimme(PRIMME_INT n, double *x, int incx, double *y, int incy, hipk_ctx *ctx);               /* :647 */
int Num_scal_hip_dprimme(PRIMME_INT n, double alpha, double *x, int incx, hipk_ctx *ctx);                   /* :678 */
int Num_larnv_hip_dprimme(int idist, PRIMME_INT *iseed, PRIMME_INT length, double *x, hipk_ctx *ctx);  /* :707  idist 1, 2 */
/* B (device, m x n, n <= 8) = alpha B op(A)^-1, A (host) triangular: side 'R' */
int Num_trsm_hd_hip_dprimme(const char *side, const char *uplo, const char *transa, const char *diag, int m, int n,   /* :785 */
      double alpha, double *a, int lda, double *b, int ldb, hipk_ctx *ctx);
/* H (n x n) = X' Y + alpha H with X, Y (m x n) on the device; H on the device / on the host */
int Num_compute_gramm_hip_dprimme(double *X, PRIMME_INT m, int n, int ldX, double *Y, PRIMME_INT ldY, double alpha,        /* :898 */
      double *H, int ldH, int isherm, int deep, hipk_ctx *ctx);
int Num_compute_gramm_ddh_hip_dprimme(double *X, PRIMME_INT m, int n, PRIMME_INT ldX, double *Y, PRIMME_INT ldY,      /* :962 */
      double alpha, double *H, int ldH, int isherm, hipk_ctx *ctx);

/* ---- single precision (_sprimme) ---- */
int Num_check_pointer_hip_sprimme(void *x);                                                      /* cublas_wrapper.c:162 */
int Num_malloc_hip_sprimme(PRIMME_INT n, float **x, hipk_ctx *ctx);                                   /* :187 */
int Num_free_hip_sprimme(float *x, hipk_ctx *ctx);                                                    /* :218 */
int Num_copy_Tmatrix_hip_sprimme(void *x, primme_op_datatype xt, PRIMME_INT m, PRIMME_INT n,      /* :254  device x of type xt -> device y */
      PRIMME_INT ldx, float *y, PRIMME_INT ldy, hipk_ctx *ctx);
int Num_copy_hip_sprimme(PRIMME_INT n, float *x, int incx, float *y, int incy, hipk_ctx *ctx);            /* :312 */
int Num_set_matrix_hip_sprimme(float *x, PRIMME_INT m, PRIMME_INT n, PRIMME_INT ldx, float *y,            /* :335  host x -> device y */
      PRIMME_INT ldy, hipk_ctx *ctx);
int Num_get_matrix_hip_sprimme(float *x, PRIMME_INT m, PRIMME_INT n, PRIMME_INT ldx, float *y,            /* :370  device x -> host y */
      PRIMME_INT ldy, hipk_ctx *ctx);
int Num_copy_matrix_hip_sprimme(float *x, PRIMME_INT m, PRIMME_INT n, PRIMME_INT ldx, float *y,           /* :739  device -> device */
      PRIMME_INT ldy, hipk_ctx *ctx);
int Num_zero_matrix_hip_sprimme(float *x, PRIMME_INT m, PRIMME_INT n, PRIMME_INT ldx, hipk_ctx *ctx); /* :768 */
/* every operand on the device: 'C','N' (C small) and 'N','N' (B small) */
int Num_gemm_hip_sprimme(const char *transa, const char *transb, int m, int n, int k, float alpha,    /* :397 */
      float *a, int lda, float *b, int ldb, float beta, float *c, int ldc, hipk_ctx *ctx);
/* C(host, m x n) = alpha * A' B + beta * C with A (k x m) and B (k x n) on the device: transa 'C'/'T', transb 'N' */
int Num_gemm_ddh_hip_sprimme(const char *transa, const char *transb, int m, int n, PRIMME_INT k, float alpha,   /* :479 */
      float *a, PRIMME_INT lda, float *b, PRIMME_INT ldb, float beta, float *c, int ldc, hipk_ctx *ctx);
/* C(device, m x n) = alpha * A B + beta * C with A (m x k) on the device and B (k x n) on the host: 'N','N';
 * (alpha, beta) = (x, 1) is the Gram-Schmidt update, (x, 0) the Ritz-vector product */
int Num_gemm_dhd_hip_sprimme(const char *transa, const char *transb, PRIMME_INT m, int n, int k, float alpha,   /* :452 */
      float *a, PRIMME_INT lda, float *b, int ldb, float beta, float *c, PRIMME_INT ldc, hipk_ctx *ctx);
int Num_gemv_hip_sprimme(const char *transa, PRIMME_INT m, int n, float alpha, float *a, int lda, float *x,   /* :507  all on the device */
      int incx, float beta, float *y, int incy, hipk_ctx *ctx);
/* y(host) = alpha * A' x + beta * y, A (m x n) and x on the device: transa 'C'/'T' */
int Num_gemv_ddh_hip_sprimme(const char *transa, PRIMME_INT m, int n, float alpha, float *a, PRIMME_INT lda,        /* :566 */
      float *x, int incx, float beta, float *y, int incy, hipk_ctx *ctx);
/* y(device) = alpha * A x + beta * y, A (m x n) on the device, x on the host: transa 'N' */
int Num_gemv_dhd_hip_sprimme(const char *transa, PRIMME_INT m, int n, float alpha, float *a, PRIMME_INT lda,        /* :594 */
      float *x, int incx, float beta, float *y, int incy, hipk_ctx *ctx);
int Num_axpy_hip_sprimme(PRIMME_INT n, float alpha, float *x, int incx, float *y, int incy, hipk_ctx *ctx);   /* :616 */
float Num_dot_hip_sprimme(PRIMME_INT n, float *x, int incx, float *y, int incy, hipk_ctx *ctx);               /* :647 */
int Num_scal_hip_sprimme(PRIMME_INT n, float alpha, float *x, int incx, hipk_ctx *ctx);                   /* :678 */
int Num_larnv_hip_sprimme(int idist, PRIMME_INT *iseed, PRIMME_INT length, float *x, hipk_ctx *ctx);  /* :707  idist 1, 2 */
/* B (device, m x n, n <= 8) = alpha B op(A)^-1, A (host) triangular: side 'R' */
int Num_trsm_hd_hip_sprimme(const char *side, const char *uplo, const char *transa, const char *diag, int m, int n,   /* :785 */
      float alpha, float *a, int lda, float *b, int ldb, hipk_ctx *ctx);
/* H (n x n) = X' Y + alpha H with X, Y (m x n) on the device; H on the device / on the host */
int Num_compute_gramm_hip_sprimme(float *X, PRIMME_INT m, int n, int ldX, float *Y, PRIMME_INT ldY, float alpha,        /* :898 */
      float *H, int ldH, int isherm, int deep, hipk_ctx *ctx);
int Num_compute_gramm_ddh_hip_sprimme(float *X, PRIMME_INT m, int n, PRIMME_INT ldX, float *Y, PRIMME_INT ldY,      /* :962 */
      float alpha, float *H, int ldH, int isherm, hipk_ctx *ctx);
#ifdef __cplusplus
}
#endif
#endif
