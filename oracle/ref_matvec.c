/* oracle/ref_matvec.c — TEST/BASELINE INFRASTRUCTURE.
 * The matrixMatvec callback a CPU user of the reference writes (cf. reference
 * examples/ex_eigs_dseq.c:160-178 and tests/COMMON/mat.c:64-90): y = A x for a host CSR
 * matrix, OpenMP over rows.  Used only to time the real reference (oracle/_ref) as
 * bench.py's cpu_baseline and in parity tests; never linked into the product. */
#include <stdint.h>
#include <stddef.h>

typedef struct { int64_t n; const int32_t *rowptr; const int32_t *colind; const double *values; } ref_csr;

/* signature of primme_params.matrixMatvec (reference include/primme_eigs.h:170-173);
 * primme->matrix (offset 264 in the 640-byte struct) holds a ref_csr* */
void ref_csr_matvec(void *x, int64_t *ldx, void *y, int64_t *ldy, int *blockSize, void *primme, int *ierr) {
   const ref_csr *A = *(const ref_csr **)((const char *)primme + 264);
   for (int c = 0; c < *blockSize; c++) {
      const double *xc = (const double *)x + (size_t)c * (size_t)*ldx;
      double *yc = (double *)y + (size_t)c * (size_t)*ldy;
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < A->n; i++) {
         double s = 0.0;
         for (int32_t p = A->rowptr[i]; p < A->rowptr[i + 1]; p++) s += A->values[p] * xc[A->colind[p]];
         yc[i] = s;
      }
   }
   *ierr = 0;
}
