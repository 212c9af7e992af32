/* primme_amd_io.h — C-ABI of the host-side matrix ingest: the on-disk
 * format and the tiler either side of the hot path.
 *
 *   primme_amd_mm_read   replaces what the reference's test driver does with
 *                        tests/COMMON/mmio.c:27-316 and tests/COMMON/csr.c:98-239 (readfullMTX):
 *                        Matrix-Market coordinate file -> CSR with sorted rows, expanding
 *                        symmetric / Hermitian / skew-symmetric storage
 *   primme_amd_csr_tile_block_diagonal   block-diagonal tiling, tile t scaled by
 *                        scale0 + scale_step * t (BASELINE configs[2] from tests/LUNDA.mtx)
 *   primme_amd_csr_transpose             explicit transpose (singular value operator)
 *   primme_amd_csr_complex_to_real       2n x 2n real-equivalent form of a complex matrix in the
 *                        interleaved (re, im) ordering: the operator hip_zprimme / hip_cprimme
 *                        are given for a Hermitian CSR matrix (primme_amd.h)
 *
 * 0-based int32 indices; values double (complex: re, im interleaved).  Returned arrays are
 * malloc'ed by the library: release them with primme_amd_host_free.  Return 0 on success,
 * -1 cannot open, -2 malformed, -3 unsupported variant, -4 too large for int32, -5 out of memory. */
#ifndef PRIMME_AMD_IO_H
#define PRIMME_AMD_IO_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
int primme_amd_mm_read(const char *path, int64_t *m, int64_t *n, int64_t *nnz, int32_t **rowptr,
      int32_t **colind, double **values, int *is_complex);
int primme_amd_csr_transpose(int64_t m, int64_t n, const int32_t *rowptr, const int32_t *colind,
      const void *values, size_t elem_size, int32_t **rowptrT, int32_t **colindT, void **valuesT);
int primme_amd_csr_tile_block_diagonal(int64_t n0, const int32_t *rowptr, const int32_t *colind,
      const double *values, int64_t ntiles, int64_t first_tile, double scale0, double scale_step,
      int32_t **rowptr_out, int32_t **colind_out, double **values_out);
int primme_amd_csr_complex_to_real(int64_t n, const int32_t *rowptr, const int32_t *colind,
      const double *values_re_im, int32_t **rowptr_out, int32_t **colind_out, double **values_out);
void primme_amd_host_free(void *p);
#ifdef __cplusplus
}
#endif
#endif
