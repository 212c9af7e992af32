/* primme_amd_wrapper.h — the reference's numerical-backend boundary (B2) spelled with
 * the reference's own routine names: what a `src/linalg/hip_wrapper.c` inside the reference tree would
 * contain next to blaslapack.c / cublas_wrapper.c / magma_wrapper.c.  Each function has the argument
 * list of the reference's Num_<op>_Sprimme (src/linalg/cublas_wrapper.c, line cited per function) for
 * the double-precision GPU instantiation; the one difference is the last argument: the reference
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
int Num_check_pointer_hip_dprimme(void *x);                                                      /* cublas_wrapper.c:162 */
int Num_malloc_hip_dprimme(PRIMME_INT n, double **x, hipk_ctx *ctx);                              /* :187 */
int Num_free_hip_dprimme(double *x, hipk_ctx *ctx);                                               /* :218 */
int Num_set_matrix_hip_dprimme(double *x, PRIMME_INT m, PRIMME_INT n, PRIMME_INT ldx, double *y,  /* :335  host x -> device y */
      PRIMME_INT ldy, hipk_ctx *ctx);
int Num_get_matrix_hip_dprimme(double *x, PRIMME_INT m, PRIMME_INT n, PRIMME_INT ldx, double *y,  /* :370  device x -> host y */
      PRIMME_INT ldy, hipk_ctx *ctx);
int Num_copy_matrix_hip_dprimme(double *x, PRIMME_INT m, PRIMME_INT n, PRIMME_INT ldx, double *y, /* :739  device -> device */
      PRIMME_INT ldy, hipk_ctx *ctx);
int Num_zero_matrix_hip_dprimme(double *x, PRIMME_INT m, PRIMME_INT n, PRIMME_INT ldx, hipk_ctx *ctx);   /* :768 */
/* C(host, m x n) = alpha * A' B + beta * C with A (k x m) and B (k x n) on the device: transa 'C'/'T', transb 'N' */
int Num_gemm_ddh_hip_dprimme(const char *transa, const char *transb, int m, int n, PRIMME_INT k, double alpha,   /* :479 */
      double *a, PRIMME_INT lda, double *b, PRIMME_INT ldb, double beta, double *c, int ldc, hipk_ctx *ctx);
/* C(device, m x n) = alpha * A B + beta * C with A (m x k) on the device and B (k x n) on the host: 'N','N';
 * (alpha, beta) = (x, 1) is the Gram-Schmidt update, (x, 0) the Ritz-vector product */
int Num_gemm_dhd_hip_dprimme(const char *transa, const char *transb, PRIMME_INT m, int n, int k, double alpha,   /* :452 */
      double *a, PRIMME_INT lda, double *b, int ldb, double beta, double *c, PRIMME_INT ldc, hipk_ctx *ctx);
/* y(host) = alpha * A' x + beta * y, A (m x n) and x on the device: transa 'C'/'T' */
int Num_gemv_ddh_hip_dprimme(const char *transa, PRIMME_INT m, int n, double alpha, double *a, PRIMME_INT lda,    /* :566 */
      double *x, int incx, double beta, double *y, int incy, hipk_ctx *ctx);
/* y(device) = alpha * A x + beta * y, A (m x n) on the device, x on the host: transa 'N' */
int Num_gemv_dhd_hip_dprimme(const char *transa, PRIMME_INT m, int n, double alpha, double *a, PRIMME_INT lda,    /* :594 */
      double *x, int incx, double beta, double *y, int incy, hipk_ctx *ctx);
int Num_axpy_hip_dprimme(PRIMME_INT n, double alpha, double *x, int incx, double *y, int incy, hipk_ctx *ctx);   /* :616 */
double Num_dot_hip_dprimme(PRIMME_INT n, double *x, int incx, double *y, int incy, hipk_ctx *ctx);               /* :647 */
int Num_scal_hip_dprimme(PRIMME_INT n, double alpha, double *x, int incx, hipk_ctx *ctx);                        /* :678 */
/* H(host, n x n, upper part) = X' Y with X, Y (m x n) on the device */
int Num_compute_gramm_ddh_hip_dprimme(double *X, PRIMME_INT m, int n, PRIMME_INT ldX, double *Y, PRIMME_INT ldY, /* :962 */
      double alpha, double *H, int ldH, int isherm, hipk_ctx *ctx);
#ifdef __cplusplus
}
#endif
#endif
