/* csr_tools.c — host-side matrix ingest feeding the device operators.
 *
 *   primme_amd_mm_read            <- reference tests/COMMON/mmio.c:27-316 (banner / size parsing)
 *                                    + tests/COMMON/csr.c:98-239 (readfullMTX: COO -> CSR,
 *                                    symmetric / Hermitian / skew expansion, sorted rows)
 *   primme_amd_csr_tile_block_diagonal  the tiler that builds BASELINE configs[2] from LUNDA.mtx
 *   primme_amd_csr_transpose      explicit A' for the singular value operator
 *   primme_amd_csr_complex_to_real  Hermitian matrix -> symmetric real-equivalent form
 *
 * Indices are 0-based int32 (the device kernels' format); values are double, complex as
 * (re, im) pairs.  Everything returned is malloc'ed; release with primme_amd_host_free. */
#include <ctype.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "primme_amd_io.h"

void primme_amd_host_free(void *p) { free(p); }

typedef struct { int32_t r, c; double re, im; } coo_t;
static int coo_cmp(const void *a, const void *b) {
   const coo_t *x = (const coo_t *)a, *y = (const coo_t *)b;
   if (x->r != y->r) return x->r < y->r ? -1 : 1;
   if (x->c != y->c) return x->c < y->c ? -1 : 1;
   return 0;
}

static void lower(char *s) { for (; *s; s++) *s = (char)tolower((unsigned char)*s); }

int primme_amd_mm_read(const char *path, int64_t *m_out, int64_t *n_out, int64_t *nnz_out,
      int32_t **rowptr_out, int32_t **colind_out, double **values_out, int *is_complex_out) {
   FILE *f = fopen(path, "r");
   if (!f) return -1;
   char line[1100], banner[64], object[64], format[64], field[64], symmetry[64];
   if (!fgets(line, sizeof line, f)) { fclose(f); return -2; }
   if (sscanf(line, "%63s %63s %63s %63s %63s", banner, object, format, field, symmetry) != 5) { fclose(f); return -2; }
   lower(object); lower(format); lower(field); lower(symmetry);
   if (strcmp(banner, "%%MatrixMarket") != 0 || strcmp(object, "matrix") != 0) { fclose(f); return -2; }
   if (strcmp(format, "coordinate") != 0) { fclose(f); return -3; }    /* dense arrays: not an operator file */
   const int pattern = !strcmp(field, "pattern"), cplx = !strcmp(field, "complex");
   if (!pattern && !cplx && strcmp(field, "real") != 0 && strcmp(field, "integer") != 0) { fclose(f); return -3; }
   const int sym = !strcmp(symmetry, "symmetric"), herm = !strcmp(symmetry, "hermitian"),
             skew = !strcmp(symmetry, "skew-symmetric");
   if (!sym && !herm && !skew && strcmp(symmetry, "general") != 0) { fclose(f); return -3; }

   do {
      if (!fgets(line, sizeof line, f)) { fclose(f); return -2; }
   } while (line[0] == '%' || line[0] == '\n' || line[0] == '\r');
   long long m, n, nz;
   if (sscanf(line, "%lld %lld %lld", &m, &n, &nz) != 3 || m < 0 || n < 0 || nz < 0) { fclose(f); return -2; }
   if (m >= 2147483647LL || n >= 2147483647LL) { fclose(f); return -4; }

   const int expand = sym || herm || skew;
   coo_t *e = (coo_t *)malloc(sizeof(coo_t) * (size_t)(expand ? 2 * nz : nz) + sizeof(coo_t));
   if (!e) { fclose(f); return -5; }
   long long cnt = 0;
   for (long long k = 0; k < nz; k++) {
      long long i, j;
      double re = 1.0, im = 0.0;
      int got;
      if (pattern) got = fscanf(f, "%lld %lld", &i, &j) == 2;
      else if (cplx) got = fscanf(f, "%lld %lld %lf %lf", &i, &j, &re, &im) == 4;
      else got = fscanf(f, "%lld %lld %lf", &i, &j, &re) == 3;
      if (!got || i < 1 || j < 1 || i > m || j > n) { free(e); fclose(f); return -2; }
      e[cnt].r = (int32_t)(i - 1); e[cnt].c = (int32_t)(j - 1); e[cnt].re = re; e[cnt].im = im; cnt++;
      if (expand && i != j) {
         e[cnt].r = (int32_t)(j - 1); e[cnt].c = (int32_t)(i - 1);
         e[cnt].re = skew ? -re : re;
         e[cnt].im = herm ? -im : (skew ? -im : im);
         cnt++;
      }
   }
   fclose(f);
   if (cnt >= 2147483647LL) { free(e); return -4; }
   qsort(e, (size_t)cnt, sizeof(coo_t), coo_cmp);

   int32_t *rp = (int32_t *)calloc((size_t)m + 1, sizeof(int32_t));
   int32_t *ci = (int32_t *)malloc(sizeof(int32_t) * (size_t)(cnt > 0 ? cnt : 1));
   double *va = (double *)malloc(sizeof(double) * (size_t)(cnt > 0 ? cnt : 1) * (cplx ? 2 : 1));
   if (!rp || !ci || !va) { free(e); free(rp); free(ci); free(va); return -5; }
   for (long long k = 0; k < cnt; k++) {
      rp[e[k].r + 1]++;
      ci[k] = e[k].c;
      if (cplx) { va[2 * k] = e[k].re; va[2 * k + 1] = e[k].im; } else va[k] = e[k].re;
   }
   for (long long i = 0; i < m; i++) rp[i + 1] += rp[i];
   free(e);
   *m_out = m; *n_out = n; *nnz_out = cnt;
   *rowptr_out = rp; *colind_out = ci; *values_out = va; *is_complex_out = cplx;
   return 0;
}

int primme_amd_csr_transpose(int64_t m, int64_t n, const int32_t *rp, const int32_t *ci, const void *val,
      size_t elem_size, int32_t **rpT_out, int32_t **ciT_out, void **valT_out) {
   const int64_t nnz = rp[m];
   int32_t *rpT = (int32_t *)calloc((size_t)n + 2, sizeof(int32_t));
   int32_t *ciT = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nnz > 0 ? nnz : 1));
   char *vT = (char *)malloc(elem_size * (size_t)(nnz > 0 ? nnz : 1));
   if (!rpT || !ciT || !vT) { free(rpT); free(ciT); free(vT); return -5; }
   for (int64_t k = 0; k < nnz; k++) rpT[ci[k] + 2]++;
   for (int64_t j = 0; j < n; j++) rpT[j + 2] += rpT[j + 1];
   /* rpT[j+1] is now the insertion cursor of column j; rows visited in order keep A' sorted */
   for (int64_t i = 0; i < m; i++)
      for (int32_t k = rp[i]; k < rp[i + 1]; k++) {
         const int32_t dst = rpT[ci[k] + 1]++;
         ciT[dst] = (int32_t)i;
         memcpy(vT + (size_t)dst * elem_size, (const char *)val + (size_t)k * elem_size, elem_size);
      }
   *rpT_out = rpT; *ciT_out = ciT; *valT_out = vT;
   return 0;
}

int primme_amd_csr_tile_block_diagonal(int64_t n0, const int32_t *rp, const int32_t *ci, const double *val,
      int64_t ntiles, int64_t first_tile, double scale0, double scale_step, int32_t **rp_out,
      int32_t **ci_out, double **val_out) {
   const int64_t nnz0 = rp[n0];
   if (n0 * (first_tile + ntiles) >= 2147483647LL || nnz0 * ntiles >= 2147483647LL) return -4;
   int32_t *trp = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n0 * ntiles + 1));
   int32_t *tci = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nnz0 * ntiles > 0 ? nnz0 * ntiles : 1));
   double *tva = (double *)malloc(sizeof(double) * (size_t)(nnz0 * ntiles > 0 ? nnz0 * ntiles : 1));
   if (!trp || !tci || !tva) { free(trp); free(tci); free(tva); return -5; }
   trp[0] = 0;
   for (int64_t t = 0; t < ntiles; t++) {
      const double s = scale0 + scale_step * (double)(first_tile + t);
      const int64_t coff = (first_tile + t) * n0;
      for (int64_t i = 0; i < n0; i++) trp[t * n0 + i + 1] = (int32_t)(rp[i + 1] + t * nnz0);
      for (int64_t k = 0; k < nnz0; k++) {
         tci[t * nnz0 + k] = (int32_t)(ci[k] + coff);
         tva[t * nnz0 + k] = val[k] * s;
      }
   }
   *rp_out = trp; *ci_out = tci; *val_out = tva;
   return 0;
}

/* Real-equivalent form of a complex CSR matrix in the interleaved ordering: entry a + ib at
 * (i, j) becomes the 2x2 block [a -b; b a] at rows 2i, 2i+1 and columns 2j, 2j+1, so that the
 * 2n real vector (re0, im0, re1, im1, ...) -- the memory of the complex n-vector -- is mapped to
 * the memory of A x.  A Hermitian gives a symmetric matrix whose eigenvalues are those of A, each
 * twice (eigs_complex.c).  values: (re, im) pairs of double. */
int primme_amd_csr_complex_to_real(int64_t n, const int32_t *rp, const int32_t *ci, const double *val,
      int32_t **rp_out, int32_t **ci_out, double **val_out) {
   const int64_t nnz = rp[n];
   if (2 * n >= 2147483647LL || 4 * nnz >= 2147483647LL) return -4;
   int32_t *rp2 = (int32_t *)malloc(sizeof(int32_t) * (size_t)(2 * n + 1));
   int32_t *ci2 = (int32_t *)malloc(sizeof(int32_t) * (size_t)(4 * nnz > 0 ? 4 * nnz : 1));
   double *va2 = (double *)malloc(sizeof(double) * (size_t)(4 * nnz > 0 ? 4 * nnz : 1));
   if (!rp2 || !ci2 || !va2) { free(rp2); free(ci2); free(va2); return -5; }
   int64_t o = 0;
   rp2[0] = 0;
   for (int64_t i = 0; i < n; i++) {
      for (int32_t k = rp[i]; k < rp[i + 1]; k++) {   /* row 2i: (a, -b) */
         ci2[o] = 2 * ci[k];     va2[o++] = val[2 * (size_t)k];
         ci2[o] = 2 * ci[k] + 1; va2[o++] = -val[2 * (size_t)k + 1];
      }
      rp2[2 * i + 1] = (int32_t)o;
      for (int32_t k = rp[i]; k < rp[i + 1]; k++) {   /* row 2i+1: (b, a) */
         ci2[o] = 2 * ci[k];     va2[o++] = val[2 * (size_t)k + 1];
         ci2[o] = 2 * ci[k] + 1; va2[o++] = val[2 * (size_t)k];
      }
      rp2[2 * i + 2] = (int32_t)o;
   }
   *rp_out = rp2; *ci_out = ci2; *val_out = va2;
   return 0;
}
