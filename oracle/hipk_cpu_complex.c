/* oracle/hipk_cpu_complex.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * The complex instantiation (HIPK_C64 / HIPK_C32) of the plain-C restatement of the device layer: what the
 * reference's numerical backend computes for SCALAR = complex (src/linalg/blaslapack.c with USE_DOUBLECOMPLEX:
 * Num_gemm_ddh "C","N" conjugates its left operand, Num_dot is xDOTC; src/eigs/auxiliary_eigs_normal.c:155-388).
 * Conventions of the C ABI for complex panels (include/primme_amd_kernels.h): panel elements and every
 * "accumulator scalar" (inner products, projection coefficients, Ritz coefficient vectors, axpy factors) are
 * (re, im) pairs, leading dimensions count complex elements; Ritz values, shifts, squared norms and scale factors
 * stay real.  hipk_cpu.c dispatches here.
 */
#include <complex.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "primme_amd_kernels.h"

typedef double _Complex zc;
void hipk_cpu_mirror(const double *out, size_t cnt);   /* hipk_cpu.c: keeps the zero-copy contract on the host build */

static size_t zesz(hipk_dtype dt) { return dt == HIPK_C64 ? 16 : 8; }
static zc ldz(hipk_dtype dt, const void *p, int64_t i) {
   if (dt == HIPK_C64) return ((const double *)p)[2 * i] + I * ((const double *)p)[2 * i + 1];
   return (double)((const float *)p)[2 * i] + I * (double)((const float *)p)[2 * i + 1];
}
static void stz(hipk_dtype dt, void *p, int64_t i, zc v) {
   if (dt == HIPK_C64) { ((double *)p)[2 * i] = creal(v); ((double *)p)[2 * i + 1] = cimag(v); }
   else { ((float *)p)[2 * i] = (float)creal(v); ((float *)p)[2 * i + 1] = (float)cimag(v); }
}
static const void *zcol(hipk_dtype dt, const void *base, int64_t ld, int j) { return (const char *)base + (size_t)j * (size_t)ld * zesz(dt); }
static const void *zseg_col(hipk_dtype dt, const hipk_seg *segs, int nseg, int j) {
   for (int s = 0; s < nseg; s++) {
      if (j < segs[s].ncols) return zcol(dt, segs[s].base, segs[s].ld, j);
      j -= segs[s].ncols > 0 ? segs[s].ncols : 0;
   }
   return NULL;
}
static int zseg_total(const hipk_seg *segs, int nseg) { int t = 0; for (int s = 0; s < nseg; s++) t += segs[s].ncols > 0 ? segs[s].ncols : 0; return t; }
static double norm2z(zc v) { return creal(v) * creal(v) + cimag(v) * cimag(v); }

/* out[j + c*ldout] = col_j^H X(:,c) */
int hipk_z_panel_dots(hipk_dtype dt, int64_t m, const hipk_seg *segs, int nseg, const void *X, int64_t ldX, int nx,
      double *out_, int ldout) {
   zc *out = (zc *)out_;
   const int tot = zseg_total(segs, nseg);
   for (int c = 0; c < nx; c++) {
      const void *x = zcol(dt, X, ldX, c);
      for (int j = 0; j < tot; j++) {
         const void *a = zseg_col(dt, segs, nseg, j);
         zc s = 0.0;
         for (int64_t i = 0; i < m; i++) s += conj(ldz(dt, a, i)) * ldz(dt, x, i);
         out[j + (size_t)c * ldout] = s;
      }
   }
   if (nx > 0) hipk_cpu_mirror(out_, 2 * ((size_t)ldout * (nx - 1) + tot));
   return 0;
}
int hipk_z_panel_project_to(hipk_dtype dt, int64_t m, const hipk_seg *segs, int nseg, const double *coef_, int ldcoef,
      const void *X, int64_t ldX, void *Xout, int64_t ldXout, int nx, double *nrm2) {
   const zc *coef = (const zc *)coef_;
   const int tot = zseg_total(segs, nseg);
   for (int c = 0; c < nx; c++) {
      const void *x = zcol(dt, X, ldX, c);
      void *o = (void *)zcol(dt, Xout, ldXout, c);
      double n2 = 0.0;
      for (int64_t i = 0; i < m; i++) {
         zc v = ldz(dt, x, i);
         for (int j = 0; j < tot; j++) v -= ldz(dt, zseg_col(dt, segs, nseg, j), i) * coef[j + (size_t)c * ldcoef];
         stz(dt, o, i, v);
         n2 += norm2z(ldz(dt, o, i));
      }
      if (nrm2) nrm2[c] = n2;
   }
   if (nrm2) hipk_cpu_mirror(nrm2, nx);
   return 0;
}
int hipk_z_panel_project_mul(hipk_dtype dt, int64_t m, const hipk_seg *segs, int nseg, const double *coef_, int ldcoef,
      const double *M_, void *X, int64_t ldX, int nx) {
   const zc *coef = (const zc *)coef_, *M = (const zc *)M_;
   if (nx <= 0) return 0;
   if (nx > 8) return 1;
   const int tot = zseg_total(segs, nseg);
   for (int64_t i = 0; i < m; i++) {
      zc xv[8], out[8];
      for (int c = 0; c < nx; c++) {
         zc v = ldz(dt, zcol(dt, X, ldX, c), i);
         for (int j = 0; j < tot; j++) v -= ldz(dt, zseg_col(dt, segs, nseg, j), i) * coef[j + (size_t)c * ldcoef];
         xv[c] = v;
      }
      for (int c = 0; c < nx; c++) { zc t = 0.0; for (int q = 0; q < nx; q++) t += xv[q] * M[q + (size_t)c * nx]; out[c] = t; }
      for (int c = 0; c < nx; c++) stz(dt, (void *)zcol(dt, X, ldX, c), i, out[c]);
   }
   return 0;
}
int hipk_z_ritz_update(hipk_dtype dt, int64_t m, const void *V, const void *W, int64_t ld, int k, const double *h_, int ldh,
      const double *theta, const hipk_job *jobs, int njobs, double *nrm2) {
   const zc *h = (const zc *)h_;
   if (k <= 0 || njobs <= 0) return 0;
   zc *vr = malloc((size_t)k * 16), *wr = malloc((size_t)k * 16), *outv = malloc((size_t)njobs * 16);
   for (int q = 0; q < njobs; q++) if (jobs[q].kind == HIPK_JOB_RES && jobs[q].slot >= 0) nrm2[jobs[q].slot] = 0.0;
   for (int64_t i = 0; i < m; i++) {
      for (int j = 0; j < k; j++) { vr[j] = ldz(dt, zcol(dt, V, ld, j), i); wr[j] = W ? ldz(dt, zcol(dt, W, ld, j), i) : 0.0; }
      for (int q = 0; q < njobs; q++) {
         const zc *hc = h + (size_t)jobs[q].col * ldh;
         zc xv = 0, yv = 0;
         for (int j = 0; j < k; j++) { xv += vr[j] * hc[j]; yv += wr[j] * hc[j]; }
         if (jobs[q].kind == HIPK_JOB_XV) outv[q] = xv;
         else if (jobs[q].kind == HIPK_JOB_XW) outv[q] = yv;
         else outv[q] = yv - theta[jobs[q].col] * xv;
      }
      for (int q = 0; q < njobs; q++) {
         zc val = outv[q];
         if (jobs[q].dst) { stz(dt, jobs[q].dst, i, val); val = ldz(dt, jobs[q].dst, i); }
         else if (dt == HIPK_C32) val = (double)(float)creal(val) + I * (double)(float)cimag(val);
         if (jobs[q].kind == HIPK_JOB_RES && jobs[q].slot >= 0) nrm2[jobs[q].slot] += norm2z(val);
      }
   }
   free(vr); free(wr); free(outv);
   { int ns = 0; for (int q = 0; q < njobs; q++) if (jobs[q].kind == HIPK_JOB_RES && jobs[q].slot + 1 > ns) ns = jobs[q].slot + 1;
     if (ns > 0) hipk_cpu_mirror(nrm2, ns); }
   return 0;
}
int hipk_z_scale_cols(hipk_dtype dt, int64_t m, void *X, int64_t ldX, int nx, const double *a) {
   for (int c = 0; c < nx; c++) { void *x = (void *)zcol(dt, X, ldX, c); for (int64_t i = 0; i < m; i++) stz(dt, x, i, a[c] * ldz(dt, x, i)); }
   return 0;
}
int hipk_z_axpy_cols(hipk_dtype dt, int64_t m, const double *a_, const void *X, int64_t ldX, void *Y, int64_t ldY, int nx) {
   const zc *a = (const zc *)a_;
   for (int c = 0; c < nx; c++) {
      const void *x = zcol(dt, X, ldX, c); void *y = (void *)zcol(dt, Y, ldY, c);
      for (int64_t i = 0; i < m; i++) stz(dt, y, i, a[c] * ldz(dt, x, i) + ldz(dt, y, i));
   }
   return 0;
}
int hipk_z_xpay_cols(hipk_dtype dt, int64_t m, const double *a_, const void *X, int64_t ldX, void *Y, int64_t ldY, int nx) {
   const zc *a = (const zc *)a_;
   for (int c = 0; c < nx; c++) {
      const void *x = zcol(dt, X, ldX, c); void *y = (void *)zcol(dt, Y, ldY, c);
      for (int64_t i = 0; i < m; i++) stz(dt, y, i, a[c] * ldz(dt, y, i) + ldz(dt, x, i));
   }
   return 0;
}
int hipk_z_col_norms2(hipk_dtype dt, int64_t m, const void *X, int64_t ldX, int nx, double *out) {
   for (int c = 0; c < nx; c++) { const void *x = zcol(dt, X, ldX, c); double s = 0; for (int64_t i = 0; i < m; i++) s += norm2z(ldz(dt, x, i)); out[c] = s; }
   hipk_cpu_mirror(out, nx);
   return 0;
}
int hipk_z_residual_cols(hipk_dtype dt, int64_t m, const void *X, int64_t ldX, void *Wr, int64_t ldW, int nx, const double *theta, double *nrm2) {
   for (int c = 0; c < nx; c++) {
      const void *x = zcol(dt, X, ldX, c); void *w = (void *)zcol(dt, Wr, ldW, c);
      double s = 0;
      for (int64_t i = 0; i < m; i++) { stz(dt, w, i, ldz(dt, w, i) - theta[c] * ldz(dt, x, i)); s += norm2z(ldz(dt, w, i)); }
      nrm2[c] = s;
   }
   hipk_cpu_mirror(nrm2, nx);
   return 0;
}
/* out[c] = X(:,c)^H Y(:,c) */
int hipk_z_pair_dots(hipk_dtype dt, int64_t m, const void *X, int64_t ldX, const void *Y, int64_t ldY, int nx, double *out_) {
   zc *out = (zc *)out_;
   for (int c = 0; c < nx; c++) {
      const void *x = zcol(dt, X, ldX, c), *y = zcol(dt, Y, ldY, c);
      zc s = 0; for (int64_t i = 0; i < m; i++) s += conj(ldz(dt, x, i)) * ldz(dt, y, i);
      out[c] = s;
   }
   hipk_cpu_mirror(out_, 2 * (size_t)nx);
   return 0;
}
/* y = A x on complex CSR data (values, x, y complex); fetch through halos like the real form */
int hipk_z_csr_matvec(hipk_dtype dt, int64_t nrows, const int32_t *rp, const int32_t *ci, const void *val, int64_t x0, int64_t xlen,
      int64_t halo_lo, const void *xlo, int64_t ld_lo, const void *xhi, int64_t ld_hi, const void *x, int64_t ldx, void *y,
      int64_t ldy, int ncols, const double *shift) {
   for (int c = 0; c < ncols; c++) {
      const void *xc = zcol(dt, x, ldx, c);
      const void *xl = xlo ? zcol(dt, xlo, ld_lo, c) : NULL, *xh = xhi ? zcol(dt, xhi, ld_hi, c) : NULL;
      void *yc = (void *)zcol(dt, y, ldy, c);
      for (int64_t r = 0; r < nrows; r++) {
         zc s = 0.0;
         for (int32_t p = rp[r]; p < rp[r + 1]; p++) {
            const int64_t l = (int64_t)ci[p] - x0;
            const zc xv = (l >= 0 && l < xlen) ? ldz(dt, xc, l) : (l < 0 ? ldz(dt, xl, l + halo_lo) : ldz(dt, xh, l - xlen));
            s += ldz(dt, val, p) * xv;
         }
         if (shift) s -= shift[c] * ldz(dt, xc, r);
         stz(dt, yc, r, s);
      }
   }
   return 0;
}
/* y = x / (Re diag - shift) */
int hipk_z_jacobi_apply(hipk_dtype dt, int64_t m, const void *diag, const double *shift, double min_den, const void *x, int64_t ldx,
      void *y, int64_t ldy, int ncols) {
   for (int c = 0; c < ncols; c++) {
      const void *xc = zcol(dt, x, ldx, c); void *yc = (void *)zcol(dt, y, ldy, c);
      for (int64_t i = 0; i < m; i++) {
         double d = creal(ldz(dt, diag, i)) - (shift ? shift[c] : 0.0);
         if (!(fabs(d) > min_den)) d = copysign(min_den, d);
         stz(dt, yc, i, ldz(dt, xc, i) / d);
      }
   }
   return 0;
}
