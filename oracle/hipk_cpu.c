/* oracle/hipk_cpu.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C restatement of the device layer's C-ABI (include/primme_amd_kernels.h):
 * every entry point computes, with straightforward loops on host memory, what
 * the reference's numerical backend computes for the same step
 * (src/linalg/blaslapack.c: Num_gemm_ddh :610-661, Num_gemv_ddh / Num_gemv_dhd
 * :730-890, Num_dot :894, Num_axpy/Num_scal, and the fused solver steps
 * src/eigs/auxiliary_eigs_normal.c:155-388 Num_update_VWXR, :70-99
 * Num_compute_residuals, src/eigs/ortho.c:229-291, tests/COMMON/mat.c:64-90 amux).
 *
 * Uses:
 *   1. kernel-level oracle: tests/ compare each HIP kernel's output with the
 *      function of the same name here (`-m gpu`);
 *   2. linked (only by oracle/Makefile, only for tests) under the product's host
 *      solver so that the host control flow can be exercised without a GPU
 *      (`-m "not gpu"`) and compared with the real reference (oracle/_ref).
 * The product library never links this file; it fails when no GPU is present.
 * "Device" pointers here are ordinary host pointers.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "primme_amd_kernels.h"

struct hipk_ctx { int dummy; double t0; };

static double now(void) {
   struct timespec ts;
   clock_gettime(CLOCK_MONOTONIC, &ts);
   return ts.tv_sec + 1e-9 * ts.tv_nsec;
}
static size_t esz(hipk_dtype dt) { return dt == HIPK_F64 ? 8 : dt == HIPK_F32 ? 4 : dt == HIPK_C64 ? 16 : 8; }
#define IS_Z(dt) ((dt) == HIPK_C64 || (dt) == HIPK_C32)
/* the complex instantiation: oracle/hipk_cpu_complex.c */
int hipk_z_panel_dots(hipk_dtype dt, int64_t m, const hipk_seg *segs, int nseg, const void *X, int64_t ldX, int nx, double *out, int ldout);
int hipk_z_panel_project_to(hipk_dtype dt, int64_t m, const hipk_seg *segs, int nseg, const double *coef, int ldcoef, const void *X, int64_t ldX, void *Xout, int64_t ldXout, int nx, double *nrm2);
int hipk_z_panel_project_mul(hipk_dtype dt, int64_t m, const hipk_seg *segs, int nseg, const double *coef, int ldcoef, const double *M, void *X, int64_t ldX, int nx);
int hipk_z_ritz_update(hipk_dtype dt, int64_t m, const void *V, const void *W, int64_t ld, int k, const double *h, int ldh, const double *theta, const hipk_job *jobs, int njobs, double *nrm2);
int hipk_z_scale_cols(hipk_dtype dt, int64_t m, void *X, int64_t ldX, int nx, const double *a);
int hipk_z_axpy_cols(hipk_dtype dt, int64_t m, const double *a, const void *X, int64_t ldX, void *Y, int64_t ldY, int nx);
int hipk_z_xpay_cols(hipk_dtype dt, int64_t m, const double *a, const void *X, int64_t ldX, void *Y, int64_t ldY, int nx);
int hipk_z_col_norms2(hipk_dtype dt, int64_t m, const void *X, int64_t ldX, int nx, double *out);
int hipk_z_residual_cols(hipk_dtype dt, int64_t m, const void *X, int64_t ldX, void *Wr, int64_t ldW, int nx, const double *theta, double *nrm2);
int hipk_z_pair_dots(hipk_dtype dt, int64_t m, const void *X, int64_t ldX, const void *Y, int64_t ldY, int nx, double *out);
int hipk_z_csr_matvec(hipk_dtype dt, int64_t nrows, const int32_t *rp, const int32_t *ci, const void *val, int64_t x0, int64_t xlen, int64_t halo_lo, const void *xlo, int64_t ld_lo, const void *xhi, int64_t ld_hi, const void *x, int64_t ldx, void *y, int64_t ldy, int ncols, const double *shift);
int hipk_z_jacobi_apply(hipk_dtype dt, int64_t m, const void *diag, const double *shift, double min_den, const void *x, int64_t ldx, void *y, int64_t ldy, int ncols);
static double ld_(hipk_dtype dt, const void *p, int64_t i) {
   return dt == HIPK_F64 ? ((const double *)p)[i] : (double)((const float *)p)[i];
}
static void st_(hipk_dtype dt, void *p, int64_t i, double v) {
   if (dt == HIPK_F64) ((double *)p)[i] = v; else ((float *)p)[i] = (float)v;
}
static const void *colp(hipk_dtype dt, const void *base, int64_t ld, int j) {
   return (const char *)base + (size_t)j * (size_t)ld * esz(dt);
}
static const void *seg_col(hipk_dtype dt, const hipk_seg *segs, int nseg, int j) {
   for (int s = 0; s < nseg; s++) {
      if (j < segs[s].ncols) return colp(dt, segs[s].base, segs[s].ld, j);
      j -= segs[s].ncols > 0 ? segs[s].ncols : 0;
   }
   return NULL;
}
static int seg_total(const hipk_seg *segs, int nseg) {
   int t = 0;
   for (int s = 0; s < nseg; s++) t += segs[s].ncols > 0 ? segs[s].ncols : 0;
   return t;
}

int hipk_ctx_create(hipk_ctx **ctx, void *s) { (void)s; *ctx = calloc(1, sizeof(hipk_ctx)); return *ctx ? 0 : -2; }
static long g_cnt[8];
/* launch counters of the plain-C layer (tests assert the solver's launch structure) */
void hipk_cpu_counts(long *out, int reset) { for (int i = 0; i < 8; i++) { out[i] = g_cnt[i]; if (reset) g_cnt[i] = 0; } }
int hipk_ctx_destroy(hipk_ctx *ctx) { if (getenv("HIPK_CPU_COUNTS")) fprintf(stderr, "hipk_cpu calls: dots %ld project %ld ritz %ld ritz_cgs %ld scale %ld\n", g_cnt[0], g_cnt[1], g_cnt[2], g_cnt[3], g_cnt[4]); free(ctx); return 0; }
void *hipk_ctx_stream(hipk_ctx *ctx) { (void)ctx; return NULL; }
int hipk_malloc(hipk_ctx *c, size_t b, void **p) { (void)c; *p = calloc(1, b ? b : 8); return *p ? 0 : -2; }
int hipk_free(hipk_ctx *c, void *p) { (void)c; free(p); return 0; }
int hipk_host_alloc(hipk_ctx *c, size_t b, void **p) { return hipk_malloc(c, b, p); }
int hipk_host_free(hipk_ctx *c, void *p) { return hipk_free(c, p); }
int hipk_h2d(hipk_ctx *c, void *d, const void *s, size_t b) { (void)c; memmove(d, s, b); return 0; }
int hipk_d2h(hipk_ctx *c, void *d, const void *s, size_t b) { (void)c; memmove(d, s, b); return 0; }
int hipk_d2d(hipk_ctx *c, void *d, const void *s, size_t b) { (void)c; memmove(d, s, b); return 0; }
int hipk_memset0(hipk_ctx *c, void *d, size_t b) { (void)c; memset(d, 0, b); return 0; }
int hipk_sync(hipk_ctx *c) { (void)c; return 0; }
static double *g_mirror_dev, *g_mirror_host; static size_t g_mirror_n;
int hipk_ctx_set_mirror(hipk_ctx *c, double *d, double *h, size_t n) { (void)c; g_mirror_dev = d; g_mirror_host = h; g_mirror_n = n; return 0; }
void hipk_cpu_mirror(const double *out, size_t cnt);
/* the checker's stand-in for the cross-rank second stage of the product's reductions (hipk_finalize_kernel<., XR>): when the stand-in
 * communicator of oracle/hostcheck_glue.c installs this hook, the results of a reduction pass through it (it sums them over the ranks
 * if the pass was armed, hipk_xreduce_arm) before they are mirrored */
void (*hipk_cpu_xr_hook)(double *out, size_t cnt);
static void mirror(const double *out, size_t cnt) { if (hipk_cpu_xr_hook) hipk_cpu_xr_hook((double *)out, cnt); hipk_cpu_mirror(out, cnt); }
void hipk_cpu_mirror(const double *out, size_t cnt) {   /* keep the zero-copy contract on the host build */
   if (g_mirror_dev && out >= g_mirror_dev && out < g_mirror_dev + g_mirror_n && g_mirror_host != g_mirror_dev)
      memmove(g_mirror_host + (out - g_mirror_dev), out, cnt * sizeof(double));
}
int hipk_is_device_ptr(const void *p) { return p != NULL; }
void pa_larnv_uniform11(int64_t iseed[4], int64_t n, double *x);      /* the host routine checked against LAPACK (tests/test_dense_host.py) */
int hipk_larnv_uniform11(hipk_ctx *ctx, hipk_dtype dt, int64_t iseed[4], int64_t n, void *x) {
   (void)ctx;
   if (n <= 0) return 0;
   double *t = malloc((size_t)n * sizeof(double));
   pa_larnv_uniform11(iseed, n, t);
   if (dt == HIPK_F64 || dt == HIPK_C64) memcpy(x, t, (size_t)n * sizeof(double));
   else for (int64_t i = 0; i < n; i++) ((float *)x)[i] = (float)t[i];
   free(t);
   return 0;
}
int hipk_wait_results(hipk_ctx *ctx) { (void)ctx; return 0; }
int hipk_publish_results(hipk_ctx *ctx, const double *dev, int count) { (void)ctx; mirror(dev, (size_t)count); return 0; }
int hipk_timer_start(hipk_ctx *c) { c->t0 = now(); return 0; }
int hipk_timer_stop(hipk_ctx *c, float *ms) { *ms = (float)((now() - c->t0) * 1e3); return 0; }

/* out[j + c*ldout] = col_j' X(:,c)   (Num_gemm_ddh "C","N") */
int hipk_panel_dots(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const hipk_seg *segs, int nseg,
      const void *X, int64_t ldX, int nx, double *out, int ldout) {
   (void)ctx; g_cnt[0]++;
   if (IS_Z(dt)) return hipk_z_panel_dots(dt, m, segs, nseg, X, ldX, nx, out, ldout);
   const int tot = seg_total(segs, nseg);
   for (int c = 0; c < nx; c++) {
      const void *x = colp(dt, X, ldX, c);
      for (int j = 0; j < tot; j++) {
         const void *a = seg_col(dt, segs, nseg, j);
         double s = 0.0;
         for (int64_t i = 0; i < m; i++) s += ld_(dt, a, i) * ld_(dt, x, i);
         out[j + (size_t)c * ldout] = s;
      }
   }
   mirror(out, (size_t)ldout * (nx - 1) + tot);
   return 0;
}

/* X(:,c) -= [segs]*coef(:,c); nrm2[c] = |X(:,c)|^2   (Num_gemv_dhd "N" + Num_dot) */
int hipk_panel_project_to(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const hipk_seg *segs, int nseg,
      const double *coef, int ldcoef, const void *X, int64_t ldX, void *Xout, int64_t ldXout, int nx, double *nrm2) {
   (void)ctx; g_cnt[1]++;
   if (IS_Z(dt)) return hipk_z_panel_project_to(dt, m, segs, nseg, coef, ldcoef, X, ldX, Xout, ldXout, nx, nrm2);
   const int tot = seg_total(segs, nseg);
   for (int c = 0; c < nx; c++) {
      const void *x = colp(dt, X, ldX, c);
      void *o = (void *)colp(dt, Xout, ldXout, c);
      double n2 = 0.0;
      for (int64_t i = 0; i < m; i++) {
         double v = ld_(dt, x, i);
         for (int j = 0; j < tot; j++) v -= ld_(dt, seg_col(dt, segs, nseg, j), i) * coef[j + (size_t)c * ldcoef];
         st_(dt, o, i, v);
         double w = ld_(dt, o, i);
         n2 += w * w;
      }
      if (nrm2) nrm2[c] = n2;
   }
   if (nrm2) mirror(nrm2, nx);
   return 0;
}
int hipk_panel_project(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const hipk_seg *segs, int nseg,
      const double *coef, int ldcoef, void *X, int64_t ldX, int nx, double *nrm2) {
   return hipk_panel_project_to(ctx, dt, m, segs, nseg, coef, ldcoef, X, ldX, X, ldX, nx, nrm2);
}

int hipk_panel_project_mul(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const hipk_seg *segs, int nseg,
      const double *coef, int ldcoef, const double *M, void *X, int64_t ldX, int nx) {
   (void)ctx;
   if (IS_Z(dt)) return hipk_z_panel_project_mul(dt, m, segs, nseg, coef, ldcoef, M, X, ldX, nx);
   if (nx <= 0) return 0;
   if (nx > 8) return 1;
   const int tot = seg_total(segs, nseg);
   for (int64_t i = 0; i < m; i++) {
      double xv[8], out[8];
      for (int c = 0; c < nx; c++) {
         double v = ld_(dt, colp(dt, X, ldX, c), i);
         for (int j = 0; j < tot; j++) v -= ld_(dt, seg_col(dt, segs, nseg, j), i) * coef[j + (size_t)c * ldcoef];
         xv[c] = v;
      }
      for (int c = 0; c < nx; c++) { double t = 0.0; for (int q = 0; q < nx; q++) t += xv[q] * M[q + (size_t)c * nx]; out[c] = t; }
      for (int c = 0; c < nx; c++) st_(dt, (void *)colp(dt, X, ldX, c), i, out[c]);
   }
   return 0;
}

/* Num_update_VWXR restated row by row (all reads of a row precede its writes) */
int hipk_ritz_update(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const void *V, const void *W,
      int64_t ld, int k, const double *h, int ldh, const double *theta, const hipk_job *jobs,
      int njobs, double *nrm2) {
   (void)ctx; g_cnt[2]++;
   if (IS_Z(dt)) return hipk_z_ritz_update(dt, m, V, W, ld, k, h, ldh, theta, jobs, njobs, nrm2);
   if (k <= 0 || njobs <= 0) return 0;
   double *vr = malloc((size_t)k * 8), *wr = malloc((size_t)k * 8), *outv = malloc((size_t)njobs * 8);
   for (int q = 0; q < njobs; q++) if (jobs[q].kind == HIPK_JOB_RES && jobs[q].slot >= 0) nrm2[jobs[q].slot] = 0.0;
   for (int64_t i = 0; i < m; i++) {
      for (int j = 0; j < k; j++) { vr[j] = ld_(dt, colp(dt, V, ld, j), i); wr[j] = W ? ld_(dt, colp(dt, W, ld, j), i) : 0.0; }
      for (int q = 0; q < njobs; q++) {
         const double *hc = h + (size_t)jobs[q].col * ldh;
         double xv = 0, yv = 0;
         for (int j = 0; j < k; j++) { xv += vr[j] * hc[j]; yv += wr[j] * hc[j]; }
         if (jobs[q].kind == HIPK_JOB_XV) outv[q] = xv;
         else if (jobs[q].kind == HIPK_JOB_XW) outv[q] = yv;
         else outv[q] = yv - theta[jobs[q].col] * xv;
      }
      for (int q = 0; q < njobs; q++) {
         double val = outv[q];
         if (jobs[q].dst) { st_(dt, jobs[q].dst, i, val); val = ld_(dt, jobs[q].dst, i); }
         else if (dt == HIPK_F32) val = (double)(float)val;
         if (jobs[q].kind == HIPK_JOB_RES && jobs[q].slot >= 0) nrm2[jobs[q].slot] += val * val;
      }
   }
   free(vr); free(wr); free(outv);
   { int ns = 0; for (int q = 0; q < njobs; q++) if (jobs[q].kind == HIPK_JOB_RES && jobs[q].slot + 1 > ns) ns = jobs[q].slot + 1;
     if (ns > 0) mirror(nrm2, ns); }
   return 0;
}

/* hipk_ritz_update with one residual job + the residual's inner products with the basis being written:
 * the update first (row by row, reads before writes), then plain dot products with the STORED columns */
int hipk_ritz_update_overlaps(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const void *V, const void *W,
      int64_t ld, int k, const double *h, int ldh, const double *theta, const hipk_job *jobs, int njobs,
      double *nrm2, int nb, const void *Q, int64_t ldQ, int L, double *ov) {
   int nres = 0, rq = -1, nxv = 0, nxw = 0;
   const void *xv[16], *xw[16];
   g_cnt[6]++; g_cnt[2]--;          /* counted as its own kind, not as the plain update it calls */
   if (k <= 0 || k > 32 || nb <= 0 || nb > 16 || L < 0 || L > 32 || !ov) return -1;
   for (int q = 0; q < njobs; q++) {
      if (jobs[q].kind == HIPK_JOB_RES) { nres++; rq = q; }
      else if (jobs[q].kind == HIPK_JOB_XV) { if (nxv < nb) xv[nxv] = jobs[q].dst; nxv++; }
      else if (jobs[q].kind == HIPK_JOB_XW) { if (nxw < nb) xw[nxw] = jobs[q].dst; nxw++; }
   }
   if (nres != 1 || nxv < nb || nxw < nb) return -1;
   /* W(:,k-1)'Q and the residual (which needs the OLD basis) before anything is overwritten */
   const int nsl = 2 * nb + 2 * L + 1;
   double *r = malloc((size_t)(m > 0 ? m : 1) * 8);
   for (int j = 0; j < nsl; j++) ov[j] = 0.0;
   const double *hc = h + (size_t)jobs[rq].col * ldh;
   for (int64_t i = 0; i < m; i++) {
      double x = 0, y = 0;
      for (int j = 0; j < k; j++) { x += ld_(dt, colp(dt, V, ld, j), i) * hc[j]; y += ld_(dt, colp(dt, W, ld, j), i) * hc[j]; }
      double v = y - theta[jobs[rq].col] * x;
      if (dt == HIPK_F32) v = (double)(float)v;
      r[i] = v;
      for (int l = 0; l < L; l++) ov[2 * nb + L + 1 + l] += ld_(dt, colp(dt, W, ld, k - 1), i) * ld_(dt, colp(dt, Q, ldQ, l), i);
   }
   double dummy[64];
   int rc = hipk_ritz_update(ctx, dt, m, V, W, ld, k, h, ldh, theta, jobs, njobs, nrm2 ? nrm2 : dummy);
   if (rc) { free(r); return rc; }
   for (int64_t i = 0; i < m; i++) {
      for (int o = 0; o < nb; o++) { ov[o] += ld_(dt, xv[o], i) * r[i]; ov[nb + L + 1 + o] += ld_(dt, xw[o], i) * r[i]; }
      for (int l = 0; l < L; l++) ov[nb + l] += ld_(dt, colp(dt, Q, ldQ, l), i) * r[i];
      ov[nb + L] += r[i] * r[i];
   }
   free(r);
   mirror(ov, (size_t)nsl);
   return 0;
}

/* r = W h - theta V h; out = [V'r | Q'r | r'r]  (Num_update_VWXR + first CGS pass dots) */
int hipk_ritz_residual_overlaps(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const void *V, const void *W,
      int64_t ld, int k, const double *hcol, double theta, void *dst, const void *Q, int64_t ldQ, int L,
      int want_wtr, double *out) {
   (void)ctx; g_cnt[3]++;
   if (want_wtr && (k > HIPK_WTR_MAX_K || L > HIPK_WTR_MAX_K)) return -1;
   const int nout = k + L + 1 + (want_wtr ? k + L : 0);
   for (int j = 0; j < nout; j++) out[j] = 0.0;
   for (int64_t i = 0; i < m; i++) {
      double x = 0, y = 0;
      for (int j = 0; j < k; j++) { x += ld_(dt, colp(dt, V, ld, j), i) * hcol[j]; y += ld_(dt, colp(dt, W, ld, j), i) * hcol[j]; }
      st_(dt, dst, i, y - theta * x);
   }
   for (int64_t i = 0; i < m; i++) {
      const double r = ld_(dt, dst, i);
      for (int j = 0; j < k; j++) out[j] += ld_(dt, colp(dt, V, ld, j), i) * r;
      for (int q = 0; q < L; q++) out[k + q] += ld_(dt, colp(dt, Q, ldQ, q), i) * r;
      out[k + L] += r * r;
      if (want_wtr) {
         for (int j = 0; j < k; j++) out[k + L + 1 + j] += ld_(dt, colp(dt, W, ld, j), i) * r;
         for (int q = 0; q < L; q++) out[2 * k + L + 1 + q] += ld_(dt, colp(dt, W, ld, k - 1), i) * ld_(dt, colp(dt, Q, ldQ, q), i);
      }
   }
   mirror(out, (size_t)nout);
   return 0;
}

/* the residual pass with the coefficient vector / Ritz value / status left in "device" memory by hipk_rr_arrow */
int hipk_ritz_residual_overlaps_dev(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const void *V, const void *W, int64_t ld, int k,
      const double *hth, void *dst, const void *Q, int64_t ldQ, int L, int want_wtr, double *out) {
   if (!hth) return -1;
   if (hth[33] != 0.0) return 0;                 /* no valid pair: the launch leaves at once */
   return hipk_ritz_residual_overlaps(ctx, dt, m, V, W, ld, k, hth, hth[32], dst, Q, ldQ, L, want_wtr, out);
}
unsigned long long hipk_seq_issued(hipk_ctx *ctx) { (void)ctx; return 0; }
int hipk_wait_seq(hipk_ctx *ctx, unsigned long long seq) { (void)ctx; (void)seq; return 0; }

/* One eigenpair of the arrowhead matrix [diag(theta) z; z' alpha] through the secular equation: the plain-C restatement of
 * rr_arrow_kernel (csrc/hipk_panels.hip; include/primme_amd_kernels.h: hipk_rr_arrow).  Same formulas, sequential sums. */
int hipk_rr_arrow(hipk_ctx *ctx, const hipk_rr_in *in, const double *fov, int nfov, const double *alpha_dev, double *out) {
   (void)ctx;
   if (!in || in->k < 1 || in->k > 16 || in->L < 0 || in->L > 10) return -1;
   const int k = in->k, L = in->L, c = in->cand;
   const double sgn = in->largest ? -1.0 : 1.0;
   const double n2 = fov[nfov], nt = sqrt(n2), alpha = alpha_dev[0];
   double v1[16], v2[16], th[16], z[16], y[17], h[17];
   int status = (c < 0 || c > k || !(n2 > 0.0)) ? 1 : 0;
   for (int j = 0; j < k; j++) {
      double gq = 0.0;
      for (int l = 0; l < L; l++) gq += ((in->grow_row && j == k - 1) ? fov[2 * k + L + 1 + l] : in->G[j + l * k]) * fov[k + l];
      v1[j] = fov[k + L + 1 + j] - gq;
      v2[j] = fov[j];
   }
   double zn2 = 0.0;
   for (int i = 0; i < k; i++) {
      double a1 = 0.0, a2 = 0.0;
      for (int r = 0; r < k; r++) { a1 += in->Y[r + i * k] * v1[r]; a2 += in->Y[r + i * k] * v2[r]; }
      z[i] = sgn * (a1 - in->theta[i] * a2) / nt;
      th[i] = sgn * in->theta[i];
      zn2 += z[i] * z[i];
      if (!isfinite(th[i]) || !isfinite(z[i]) || (i > 0 && !(th[i - 1] < th[i]))) status = status ? status : 2;
   }
   const double al = sgn * alpha, zn = sqrt(zn2);
   if (!isfinite(al)) status = status ? status : 2;
   double lam = 0.0, ynorm2 = 1.0;
   for (int j = 0; j <= 16; j++) y[j] = 0.0;
   if (status == 0) {
      int o;
      double lo, hi;
      if (c == 0) { o = 0; lo = fmin(0.0, al - th[0]) - zn - 1e-300; lo -= 4e-16 * fabs(lo); hi = 0.0; }
      else if (c == k) { o = k - 1; lo = 0.0; hi = fmax(0.0, al - th[k - 1]) + zn + 1e-300; hi += 4e-16 * fabs(hi); }
      else {
         const double gap = th[c] - th[c - 1], mid = 0.5 * gap;
         double sm = 0.0;
         for (int j = 0; j < k; j++) sm += z[j] * z[j] / ((th[j] - th[c - 1]) - mid);
         const double gm = (al - th[c - 1]) - mid - sm;
         if (gm > 0.0) { o = c; lo = -mid; hi = 0.0; }
         else { o = c - 1; lo = 0.0; hi = mid; }
      }
      const double a0 = al - th[o], B = z[o] * z[o];
      const int neg = hi == 0.0;
      double mu = 0.5 * (lo + hi);
      int it = 0;
      /* the pole at the origin exact, the rest of the sum by its tangent: a quadratic per step (csrc/hipk_panels.hip) */
      for (; it < 100; it++) {
         double S = 0.0, Sp = 0.0;
         for (int j = 0; j < k; j++) if (j != o) { const double r = 1.0 / ((th[j] - th[o]) - mu), t = z[j] * z[j] * r; S += t; Sp += t * r; }
         const double pole = B / mu, g = a0 - mu - S + pole;
         if (!(g == g)) { status = 3; break; }
         /* converged: g is zero to the rounding of its own terms (going on from here only moves mu by an ulp — or throws the
          * next point an ulp outside the bracket, whose far end was never tightened, and the midpoint fall-back then crawls) */
         if (fabs(g) <= 2.3e-16 * (fabs(a0) + fabs(mu) + fabs(S) + fabs(pole))) break;
         if (g > 0.0) lo = mu; else hi = mu;
         const double A = 1.0 + Sp, Cc = a0 - S + Sp * mu, disc = sqrt(Cc * Cc + 4.0 * A * B);
         double mn;
         if (neg) mn = (Cc > 0.0) ? -2.0 * B / (Cc + disc) : (Cc - disc) / (2.0 * A);
         else mn = (Cc < 0.0) ? 2.0 * B / (disc - Cc) : (Cc + disc) / (2.0 * A);
         if (!(mn > lo && mn < hi)) { if (fabs(mn - mu) <= 1e-14 * fabs(mu)) break; mn = 0.5 * (lo + hi); }
         const double step = fabs(mn - mu);
         const int done = step <= 4.4e-16 * fabs(mn) || mn == lo || mn == hi;
         if (getenv("HIPK_RR_DEBUG")) fprintf(stderr, "  rr k=%d c=%d it=%d mu=%.17g mn=%.17g g=%.3e lo=%.17g hi=%.17g\n", k, c, it, mu, mn, g, lo, hi);
         mu = mn;
         if (done) break;
      }
      if (it >= 100) status = 4;
      g_cnt[7] += it + 1;                        /* steps of the secular iteration (hipk_cpu_counts: diagnostics) */
      lam = sgn * (th[o] + mu);
      for (int j = 0; j < k; j++) { y[j] = z[j] / (mu - (th[j] - th[o])); ynorm2 += y[j] * y[j]; }
      if (!isfinite(lam) || !isfinite(ynorm2)) status = 5;
   }
   const double inv = 1.0 / sqrt(ynorm2);
   for (int j = 0; j < k; j++) { double hv = 0.0; for (int i = 0; i < k; i++) hv += in->Y[j + i * k] * y[i]; h[j] = hv * inv; }
   h[k] = inv;
   for (int j = 0; j <= k; j++) out[j] = h[j];
   out[32] = lam; out[33] = (double)status;
   mirror(out, 34);
   return 0;
}

/* the tail without its own second-stage launches (include/primme_amd_kernels.h: hipk_tail_defer): on the CPU every
 * reduction is complete when its call returns, so nothing is ever deferred (0 flags accepted) and hipk_tail_finish is the
 * Rayleigh-Ritz step alone — the host solver takes the same decisions in the same order either way */
int hipk_tail_defer(hipk_ctx *ctx, int want) { (void)ctx; (void)want; return 0; }
void hipk_tail_abandon(hipk_ctx *ctx) { (void)ctx; }
void hipk_skip_next_flag(hipk_ctx *ctx) { (void)ctx; }
int hipk_tail_pending(hipk_ctx *ctx) { (void)ctx; return 0; }
int hipk_tail_finish(hipk_ctx *ctx, const hipk_rr_in *in, const double *fov, int nfov, const double *alpha_dev, double *hnext_out) {
   if (in) return hipk_rr_arrow(ctx, in, fov, nfov, alpha_dev, hnext_out);
   return 0;
}

int hipk_scale_cols(hipk_ctx *ctx, hipk_dtype dt, int64_t m, void *X, int64_t ldX, int nx, const double *a) {
   (void)ctx;
   if (IS_Z(dt)) return hipk_z_scale_cols(dt, m, X, ldX, nx, a);
   for (int c = 0; c < nx; c++) { void *x = (void *)colp(dt, X, ldX, c); for (int64_t i = 0; i < m; i++) st_(dt, x, i, a[c] * ld_(dt, x, i)); }
   return 0;
}
int hipk_scale_cols_rsqrt_dev(hipk_ctx *ctx, hipk_dtype dt, int64_t m, void *X, int64_t ldX, int nx, const double *n2) {
   double a[64];
   if (nx > 64) return -1;
   for (int c = 0; c < nx; c++) a[c] = 1.0 / sqrt(n2[c]);
   return hipk_scale_cols(ctx, dt, m, X, ldX, nx, a);
}
int hipk_axpy_cols(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const double *a, const void *X, int64_t ldX,
      void *Y, int64_t ldY, int nx) {
   (void)ctx;
   if (IS_Z(dt)) return hipk_z_axpy_cols(dt, m, a, X, ldX, Y, ldY, nx);
   for (int c = 0; c < nx; c++) {
      const void *x = colp(dt, X, ldX, c); void *y = (void *)colp(dt, Y, ldY, c);
      for (int64_t i = 0; i < m; i++) st_(dt, y, i, a[c] * ld_(dt, x, i) + ld_(dt, y, i));
   }
   return 0;
}
int hipk_copy_cols(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const void *X, int64_t ldX, void *Y, int64_t ldY, int nx) {
   (void)ctx;
   for (int c = 0; c < nx; c++) memmove((void *)colp(dt, Y, ldY, c), colp(dt, X, ldX, c), (size_t)m * esz(dt));
   return 0;
}
int hipk_pair_rotate(hipk_ctx *ctx, hipk_dtype dt, int64_t npairs, const void *X, int64_t ldX, void *Y, int64_t ldY, int nx) {
   (void)ctx;
   for (int c = 0; c < nx; c++) {
      if (dt == HIPK_F64) {
         const double *x = (const double *)colp(dt, X, ldX, c); double *y = (double *)colp(dt, Y, ldY, c);
         for (int64_t i = 0; i < npairs; i++) { double re = x[2 * i], im = x[2 * i + 1]; y[2 * i] = -im; y[2 * i + 1] = re; }
      } else {
         const float *x = (const float *)colp(dt, X, ldX, c); float *y = (float *)colp(dt, Y, ldY, c);
         for (int64_t i = 0; i < npairs; i++) { float re = x[2 * i], im = x[2 * i + 1]; y[2 * i] = -im; y[2 * i + 1] = re; }
      }
   }
   return 0;
}
int hipk_gather_cols(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const void *X, int64_t ldX, const int *perm, int n,
      void *Y, int64_t ldY) {
   (void)ctx;
   for (int c = 0; c < n; c++) memcpy((void *)colp(dt, Y, ldY, c), colp(dt, X, ldX, perm[c]), (size_t)m * esz(dt));
   return 0;
}
int hipk_col_norms2(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const void *X, int64_t ldX, int nx, double *out) {
   (void)ctx;
   if (IS_Z(dt)) return hipk_z_col_norms2(dt, m, X, ldX, nx, out);
   for (int c = 0; c < nx; c++) { const void *x = colp(dt, X, ldX, c); double s = 0; for (int64_t i = 0; i < m; i++) s += ld_(dt, x, i) * ld_(dt, x, i); out[c] = s; }
   mirror(out, nx);
   return 0;
}
int hipk_residual_cols(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const void *X, int64_t ldX, void *Wr, int64_t ldW,
      int nx, const double *theta, double *nrm2) {
   (void)ctx;
   if (IS_Z(dt)) return hipk_z_residual_cols(dt, m, X, ldX, Wr, ldW, nx, theta, nrm2);
   for (int c = 0; c < nx; c++) {
      const void *x = colp(dt, X, ldX, c); void *w = (void *)colp(dt, Wr, ldW, c);
      double s = 0;
      for (int64_t i = 0; i < m; i++) { st_(dt, w, i, ld_(dt, w, i) - theta[c] * ld_(dt, x, i)); double r = ld_(dt, w, i); s += r * r; }
      nrm2[c] = s;
   }
   mirror(nrm2, nx);
   return 0;
}

int hipk_pair_dots(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const void *X, int64_t ldX, const void *Y, int64_t ldY,
      int nx, double *out) {
   (void)ctx;
   if (IS_Z(dt)) return hipk_z_pair_dots(dt, m, X, ldX, Y, ldY, nx, out);
   for (int c = 0; c < nx; c++) {
      const void *x = colp(dt, X, ldX, c), *y = colp(dt, Y, ldY, c);
      double s = 0; for (int64_t i = 0; i < m; i++) s += ld_(dt, x, i) * ld_(dt, y, i);
      out[c] = s;
   }
   mirror(out, nx);
   return 0;
}
int hipk_xpay_cols(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const double *a, const void *X, int64_t ldX,
      void *Y, int64_t ldY, int nx) {
   (void)ctx;
   if (IS_Z(dt)) return hipk_z_xpay_cols(dt, m, a, X, ldX, Y, ldY, nx);
   for (int c = 0; c < nx; c++) {
      const void *x = colp(dt, X, ldX, c); void *y = (void *)colp(dt, Y, ldY, c);
      for (int64_t i = 0; i < m; i++) st_(dt, y, i, a[c] * ld_(dt, y, i) + ld_(dt, x, i));
   }
   return 0;
}
int hipk_axpy_dot(hipk_ctx *ctx, hipk_dtype dt, int64_t m, int nx, const double *a, const void *X, int64_t ldX,
      void *Y, int64_t ldY, const void *Z, int64_t ldZ, double *out) {
   int rc = hipk_axpy_cols(ctx, dt, m, a, X, ldX, Y, ldY, nx);
   if (rc) return rc;
   return hipk_pair_dots(ctx, dt, m, Z ? Z : Y, Z ? ldZ : ldY, Y, ldY, nx, out);
}
int hipk_qmr_update(hipk_ctx *ctx, hipk_dtype dt, int64_t m, int nx, const double *gam, const double *eta,
      const void *D, int64_t ldD, void *Delta, int64_t ldDelta, void *Sol, int64_t ldSol, double *dotsol) {
   (void)ctx;
   for (int c = 0; c < nx; c++) {
      const void *d = colp(dt, D, ldD, c); void *de = (void *)colp(dt, Delta, ldDelta, c); void *so = (void *)colp(dt, Sol, ldSol, c);
      double s = 0;
      for (int64_t i = 0; i < m; i++) {
         st_(dt, de, i, ld_(dt, de, i) * gam[c] + ld_(dt, d, i) * eta[c]);
         st_(dt, so, i, ld_(dt, de, i) + ld_(dt, so, i));
         s += ld_(dt, so, i) * ld_(dt, so, i);
      }
      dotsol[c] = s;
   }
   mirror(dotsol, nx);
   return 0;
}

/* ---- sparse operator (amux: y = A x, CSR) ----------------------------------- */
struct hipk_csr {
   hipk_dtype dt; int kind; int64_t nrows, ncols, row0, nnz, x0, xlen;
   int32_t *rowptr, *colind; void *values; void *diag;
   int64_t halo_lo, halo_hi; const void *xlo, *xhi; int64_t ld_lo, ld_hi; int sx, sy, sz;
};
static double fetch(const hipk_csr *A, const void *x, const void *xlo, const void *xhi, int64_t g) {
   int64_t l = g - A->x0;
   if (l >= 0 && l < A->xlen) return ld_(A->dt, x, l);
   if (l < 0) return ld_(A->dt, xlo, l + A->halo_lo);
   return ld_(A->dt, xhi, l - A->xlen);
}
static int csr_create_impl(hipk_dtype dt, int64_t nr, int64_t nc, int64_t row0, int64_t x0, int64_t xlen,
      const int32_t *rp, const int32_t *ci, const void *val, hipk_csr **out) {
   hipk_csr *A = calloc(1, sizeof(*A));
   A->dt = dt; A->nrows = nr; A->ncols = nc; A->row0 = row0; A->nnz = rp[nr]; A->x0 = x0; A->xlen = xlen;
   A->rowptr = malloc((size_t)(nr + 1) * 4); memcpy(A->rowptr, rp, (size_t)(nr + 1) * 4);
   A->colind = malloc((size_t)A->nnz * 4 + 4); memcpy(A->colind, ci, (size_t)A->nnz * 4);
   A->values = malloc((size_t)A->nnz * esz(dt) + 8); memcpy(A->values, val, (size_t)A->nnz * esz(dt));
   A->diag = calloc((size_t)nr + 1, esz(dt));
   for (int64_t i = 0; i < nr; i++)
      for (int32_t p = rp[i]; p < rp[i + 1]; p++) {
         int64_t g = ci[p];
         if (g < x0 && x0 - g > A->halo_lo) A->halo_lo = x0 - g;
         if (g >= x0 + xlen && g - (x0 + xlen) + 1 > A->halo_hi) A->halo_hi = g - (x0 + xlen) + 1;
         if (g == row0 + i) { if (IS_Z(dt)) memcpy((char *)A->diag + (size_t)i * esz(dt), (const char *)val + (size_t)p * esz(dt), esz(dt)); else st_(dt, A->diag, i, ld_(dt, val, p)); }
      }
   *out = A;
   return 0;
}
int hipk_csr_create(hipk_ctx *ctx, hipk_dtype dt, int64_t nr, int64_t nc, int64_t row0, const int32_t *rp,
      const int32_t *ci, const void *val, hipk_csr **out) {
   (void)ctx;
   return csr_create_impl(dt, nr, nc, row0, row0, nr, rp, ci, val, out);
}
int hipk_csr_create_rect(hipk_ctx *ctx, hipk_dtype dt, int64_t nr, int64_t nc, const int32_t *rp,
      const int32_t *ci, const void *val, hipk_csr **out) {
   (void)ctx;
   return csr_create_impl(dt, nr, nc, 0, 0, nc, rp, ci, val, out);
}
int hipk_stencil_create(hipk_ctx *ctx, hipk_dtype dt, int nx, int ny, int nz, int64_t row0, int64_t nr, hipk_csr **out) {
   (void)ctx;
   hipk_csr *A = calloc(1, sizeof(*A));
   A->dt = dt; A->kind = 1; A->sx = nx; A->sy = ny > 0 ? ny : 1; A->sz = nz > 0 ? nz : 1;
   int64_t n = (int64_t)A->sx * A->sy * A->sz;
   A->nrows = nr; A->ncols = n; A->row0 = row0; A->x0 = row0; A->xlen = nr;
   int dims = A->sz > 1 ? 3 : (A->sy > 1 ? 2 : 1);
   A->nnz = n * (2 * dims + 1);
   int64_t reach = A->sz > 1 ? (int64_t)A->sx * A->sy : (A->sy > 1 ? A->sx : 1);
   A->halo_lo = row0 > 0 ? (reach < row0 ? reach : row0) : 0;
   int64_t above = n - (row0 + nr);
   A->halo_hi = above > 0 ? (reach < above ? reach : above) : 0;
   A->diag = calloc((size_t)nr + 1, esz(dt));
   for (int64_t i = 0; i < nr; i++) st_(dt, A->diag, i, 2.0 * dims);
   *out = A;
   return 0;
}
int hipk_csr_destroy(hipk_csr *A) { if (A) { free(A->rowptr); free(A->colind); free(A->values); free(A->diag); free(A); } return 0; }
const void *hipk_csr_diag(hipk_csr *A) { return A->diag; }
int64_t hipk_csr_nnz(const hipk_csr *A) { return A->nnz; }
int64_t hipk_csr_halo_lo(const hipk_csr *A) { return A->halo_lo; }
int64_t hipk_csr_halo_hi(const hipk_csr *A) { return A->halo_hi; }
int hipk_csr_set_halo(hipk_csr *A, const void *lo, const void *hi) { A->xlo = lo; A->xhi = hi; A->ld_lo = A->halo_lo; A->ld_hi = A->halo_hi; return 0; }
int hipk_csr_set_halo_ld(hipk_csr *A, const void *lo, int64_t ld_lo, const void *hi, int64_t ld_hi) { A->xlo = lo; A->xhi = hi; A->ld_lo = ld_lo; A->ld_hi = ld_hi; return 0; }
int hipk_csr_kind(const hipk_csr *A) { return A->kind; }
hipk_dtype hipk_csr_dtype(const hipk_csr *A) { return A->dt; }
int64_t hipk_csr_nrows(const hipk_csr *A) { return A->nrows; }

int hipk_csr_matvec(hipk_csr *A, void *stream, const void *x, int64_t ldx, void *y, int64_t ldy, int ncols) {
   (void)stream;
   const hipk_dtype dt = A->dt;
   if (IS_Z(dt)) return A->kind == 0 ? hipk_z_csr_matvec(dt, A->nrows, A->rowptr, A->colind, A->values, A->x0, A->xlen, A->halo_lo, A->xlo, A->ld_lo, A->xhi, A->ld_hi, x, ldx, y, ldy, ncols, NULL) : -44;
   for (int c = 0; c < ncols; c++) {
      const void *xc = colp(dt, x, ldx, c);
      const void *lo = A->xlo ? (const char *)A->xlo + (size_t)c * A->ld_lo * esz(dt) : NULL;
      const void *hi = A->xhi ? (const char *)A->xhi + (size_t)c * A->ld_hi * esz(dt) : NULL;
      void *yc = (void *)colp(dt, y, ldy, c);
      if (A->kind == 0) {
         for (int64_t i = 0; i < A->nrows; i++) {
            double s = 0;
            for (int32_t p = A->rowptr[i]; p < A->rowptr[i + 1]; p++) s += ld_(dt, A->values, p) * fetch(A, xc, lo, hi, A->colind[p]);
            st_(dt, yc, i, s);
         }
      } else {
         const int64_t plane = (int64_t)A->sx * A->sy;
         const double dg = A->sz > 1 ? 6.0 : (A->sy > 1 ? 4.0 : 2.0);
         for (int64_t l = 0; l < A->nrows; l++) {
            int64_t g = A->row0 + l;
            int ix = (int)(g % A->sx), iy = (int)((g / A->sx) % A->sy), iz = (int)(g / plane);
            double s = dg * ld_(dt, xc, l);
            if (ix > 0) s -= fetch(A, xc, lo, hi, g - 1);
            if (ix < A->sx - 1) s -= fetch(A, xc, lo, hi, g + 1);
            if (A->sy > 1) { if (iy > 0) s -= fetch(A, xc, lo, hi, g - A->sx); if (iy < A->sy - 1) s -= fetch(A, xc, lo, hi, g + A->sx); }
            if (A->sz > 1) { if (iz > 0) s -= fetch(A, xc, lo, hi, g - plane); if (iz < A->sz - 1) s -= fetch(A, xc, lo, hi, g + plane); }
            st_(dt, yc, l, s);
         }
      }
   }
   return 0;
}

/* y = A (a x), xout = a x, dot[0] = xout' y, a = 1/sqrt(norm2[0]): the normalisation, the operator and
 * the two-vector inner product of the one-synchronisation GD iteration in one call */
int hipk_csr_matvec_scaled(hipk_csr *A, hipk_ctx *ctx, const void *x, const double *norm2, void *xout, void *y, double *dot) {
   (void)ctx; g_cnt[5]++;
   if (A->kind != 0 || A->x0 != A->row0 || A->xlen != A->nrows || x == xout) return -1;
   const hipk_dtype dt = A->dt;
   if (IS_Z(dt)) return -44;
   const double a = norm2 ? 1.0 / sqrt(norm2[0]) : 1.0;
   double d = 0.0;
   for (int64_t i = 0; i < A->nrows; i++) {
      double s = 0;
      for (int32_t p = A->rowptr[i]; p < A->rowptr[i + 1]; p++) {
         /* the gathered entry rounded to the panel type after scaling, as if read from the scaled vector */
         double xv = a * fetch(A, x, A->xlo, A->xhi, A->colind[p]);
         if (dt == HIPK_F32) xv = (double)(float)xv;
         s += ld_(dt, A->values, p) * xv;
      }
      st_(dt, y, i, s);
   }
   for (int64_t i = 0; i < A->nrows; i++) {
      st_(dt, xout, i, a * ld_(dt, x, i));
      d += ld_(dt, xout, i) * ld_(dt, y, i);
   }
   dot[0] = d;
   mirror(dot, 1);
   return 0;
}

int hipk_triple_dots(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const void *X, int64_t ldX, const void *V, int64_t ldV,
      const void *W, int64_t ldW, int nx, double *out) {
   (void)ctx;
   for (int c = 0; c < nx; c++) {
      const void *x = colp(dt, X, ldX, c), *v = colp(dt, V, ldV, c), *w = colp(dt, W, ldW, c);
      double a = 0, b = 0, d = 0;
      for (int64_t i = 0; i < m; i++) { a += ld_(dt, x, i) * ld_(dt, w, i); b += ld_(dt, v, i) * ld_(dt, w, i); d += ld_(dt, v, i) * ld_(dt, x, i); }
      out[c] = a; out[nx + c] = b; out[2 * nx + c] = d;
   }
   mirror(out, (size_t)3 * nx);
   return 0;
}
/* W <- W - [segs] coef, then [x'w | v'w | v'x]: the pair of launches it fuses on the device, called one after the other */
int hipk_project_triple_dots(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const hipk_seg *segs, int nseg, const double *coef,
      int ldcoef, void *W, int64_t ldW, int nx, const void *X, int64_t ldX, const void *V, int64_t ldV, double *out) {
   if (dt == HIPK_C64 || dt == HIPK_C32 || nx > 8 || nx <= 0) return nx <= 0 ? 0 : 1;
   int tot = 0;
   for (int q = 0; q < nseg; q++) tot += segs[q].ncols > 0 ? segs[q].ncols : 0;
   if (tot <= 0) return 1;
   /* (the coefficients live where the results go: keep a copy, as the device kernel reads them before its second stage writes) */
   double cf[192 * 8];
   if (tot > 192) return 1;
   for (int c = 0; c < nx; c++) for (int j = 0; j < tot; j++) cf[j + c * tot] = coef[j + (size_t)c * ldcoef];
   int rc = hipk_panel_project(ctx, dt, m, segs, nseg, cf, tot, W, ldW, nx, NULL);
   if (rc) return rc;
   return hipk_triple_dots(ctx, dt, m, X, ldX, V, ldV, W, ldW, nx, out);
}
int hipk_axpy_proj_dot(hipk_ctx *ctx, hipk_dtype dt, int64_t m, int nx, const double *alpha, const double *xr, const void *W,
      int64_t ldW, const void *X, int64_t ldX, void *G, int64_t ldG, double *out) {
   (void)ctx;
   for (int c = 0; c < nx; c++) {
      const void *w = colp(dt, W, ldW, c), *x = colp(dt, X, ldX, c);
      void *g = (void *)colp(dt, G, ldG, c);
      double s = 0;
      for (int64_t i = 0; i < m; i++) {
         double wp = ld_(dt, w, i) - xr[c] * ld_(dt, x, i);
         if (dt == HIPK_F32) wp = (double)(float)wp;
         st_(dt, g, i, ld_(dt, g, i) - alpha[c] * wp);
         s += ld_(dt, g, i) * ld_(dt, g, i);
      }
      out[c] = s;
   }
   mirror(out, (size_t)nx);
   return 0;
}
int hipk_csr_matvec_shifted(hipk_csr *A, void *stream, const void *x, int64_t ldx, void *y, int64_t ldy, int ncols,
      const double *shift) {
   if (ncols <= 0 || A->nrows == 0) return 0;
   if (A->kind != 0 || A->halo_lo != 0 || A->halo_hi != 0 || A->x0 != A->row0 || A->xlen != A->nrows || ncols > 64 || !shift) return 1;
   if (IS_Z(A->dt)) return hipk_z_csr_matvec(A->dt, A->nrows, A->rowptr, A->colind, A->values, A->x0, A->xlen, 0, NULL, 0, NULL, 0, x, ldx, y, ldy, ncols, shift);
   int rc = hipk_csr_matvec(A, stream, x, ldx, y, ldy, ncols);
   for (int c = 0; c < ncols && !rc; c++)
      for (int64_t i = 0; i < A->nrows; i++)
         st_(A->dt, (void *)colp(A->dt, y, ldy, c), i, ld_(A->dt, colp(A->dt, y, ldy, c), i) - shift[c] * ld_(A->dt, colp(A->dt, x, ldx, c), i));
   return rc;
}
/* The scalar recurrences of one block-QMR step (csrc/hipk_panels.hip: qmr_alpha_dev / qmr_coeffs_dev; csrc/eigs_jd.c evaluates the
 * same expressions on the host).  ISO C: every operation rounded on its own. */
static void qmr_alpha_cpu(const double *tri, int nx, int col, double rho_prev, double eps, double *alpha, double *xr) {
   *xr = tri[col];
   const double t = *xr * tri[2 * nx + col];
   const double sigma = tri[nx + col] - t;
   int bad = !isfinite(sigma) || sigma == 0.0;
   double a = 0.0;
   if (!bad) {
      a = rho_prev / sigma;
      bad = !isfinite(a) || fabs(a) < eps || fabs(a) > 1.0 / eps;
   }
   *alpha = bad ? 0.0 : a;
}
int hipk_axpy_proj_dot_jacobi_dev(hipk_ctx *ctx, hipk_dtype dt, int64_t m, int nx, const double *tri_dev, const double *rho_prev_host,
      double mach_eps, const void *W, int64_t ldW, const void *X, int64_t ldX, void *G, int64_t ldG, const void *diag, const double *shift_host,
      double min_den, double *out_dev) {
   if (nx <= 0) return 0;
   if (nx > 8 || !tri_dev) return -1;
   double al[8], xr[8];
   for (int c = 0; c < nx; c++) qmr_alpha_cpu(tri_dev, nx, c, rho_prev_host[c], mach_eps, &al[c], &xr[c]);
   return hipk_axpy_proj_dot_jacobi(ctx, dt, m, nx, al, xr, W, ldW, X, ldX, G, ldG, diag, shift_host, min_den, out_dev);
}
int hipk_qmr_update_dir_dev(hipk_ctx *ctx, hipk_dtype dt, int64_t m, int nx, const double *tri_dev, const double *ggr_dev,
      const double *rho_prev_host, const double *tau_prev_host, const double *theta_prev_host, double mach_eps, void *D, int64_t ldD,
      void *Delta, int64_t ldDelta, void *Sol, int64_t ldSol, const void *G, int64_t ldG, const void *diag, const double *shift_host,
      double min_den, double *dotsol_dev) {
   if (nx <= 0) return 0;
   if (nx > 8 || !tri_dev || !ggr_dev) return -1;
   double out[8];
   for (int c = 0; c < nx; c++) {
      double a, xr;
      qmr_alpha_cpu(tri_dev, nx, c, rho_prev_host[c], mach_eps, &a, &xr);
      out[c] = 0.0;
      if (a == 0.0) continue;                   /* the column leaves the block at this step: left alone */
      const double theta = sqrt(ggr_dev[c]) / tau_prev_host[c];
      const double t2 = theta * theta;
      const double cs = 1.0 / sqrt(1 + t2);
      const double cc = cs * cs;
      const double g1 = cc * theta_prev_host[c];
      const double gam = g1 * theta_prev_host[c];
      const double e1 = a * cs;
      const double eta = e1 * cs;
      const double bet = ggr_dev[nx + c] / rho_prev_host[c];
      const double sh = shift_host ? shift_host[c] : 0.0;
      double o1 = 0.0;
      int rc = hipk_qmr_update_dir(ctx, dt, m, 1, &gam, &eta, &bet, (void *)colp(dt, D, ldD, c), ldD, (void *)colp(dt, Delta, ldDelta, c), ldDelta,
            (void *)colp(dt, Sol, ldSol, c), ldSol, colp(dt, G, ldG, c), ldG, diag, &sh, min_den, &o1);
      if (rc) return rc;
      out[c] = o1;
   }
   for (int c = 0; c < nx; c++) dotsol_dev[c] = out[c];
   mirror(dotsol_dev, (size_t)nx);
   return 0;
}

int hipk_qmr_update_jacobi(hipk_ctx *ctx, hipk_dtype dt, int64_t m, int nx, const double *gam, const double *eta,
      const void *D, int64_t ldD, void *Delta, int64_t ldDelta, void *Sol, int64_t ldSol, const void *G, int64_t ldG,
      const void *diag, const double *shift, double min_den, void *W, int64_t ldW, double *out) {
   (void)ctx;
   if (!(min_den > 0.0)) min_den = 1e-300;
   for (int c = 0; c < nx; c++) {
      const void *d = colp(dt, D, ldD, c), *g = colp(dt, G, ldG, c);
      void *de = (void *)colp(dt, Delta, ldDelta, c), *so = (void *)colp(dt, Sol, ldSol, c), *w = (void *)colp(dt, W, ldW, c);
      double s1 = 0.0, s2 = 0.0;
      for (int64_t i = 0; i < m; i++) {
         st_(dt, de, i, ld_(dt, de, i) * gam[c] + ld_(dt, d, i) * eta[c]);
         st_(dt, so, i, ld_(dt, de, i) + ld_(dt, so, i));
         s1 += ld_(dt, so, i) * ld_(dt, so, i);
         double den = ld_(dt, diag, i) - (shift ? shift[c] : 0.0);
         if (!(fabs(den) > min_den)) den = copysign(min_den, den);
         st_(dt, w, i, ld_(dt, g, i) / den);
         s2 += ld_(dt, g, i) * ld_(dt, w, i);
      }
      out[c] = s1; out[nx + c] = s2;
   }
   mirror(out, (size_t)2 * nx);
   return 0;
}

int hipk_axpy_proj_dot_jacobi(hipk_ctx *ctx, hipk_dtype dt, int64_t m, int nx, const double *alpha, const double *xr, const void *W,
      int64_t ldW, const void *X, int64_t ldX, void *G, int64_t ldG, const void *diag, const double *shift, double min_den, double *out) {
   (void)ctx;
   if (!(min_den > 0.0)) min_den = 1e-300;
   for (int c = 0; c < nx; c++) {
      const void *w = colp(dt, W, ldW, c), *x = colp(dt, X, ldX, c);
      void *g = (void *)colp(dt, G, ldG, c);
      double s1 = 0, s2 = 0;
      for (int64_t i = 0; i < m; i++) {
         double wp = ld_(dt, w, i) - xr[c] * ld_(dt, x, i);
         if (dt == HIPK_F32) wp = (double)(float)wp;
         st_(dt, g, i, ld_(dt, g, i) - alpha[c] * wp);
         const double gi = ld_(dt, g, i);
         s1 += gi * gi;
         double den = ld_(dt, diag, i) - (shift ? shift[c] : 0.0);
         if (!(fabs(den) > min_den)) den = copysign(min_den, den);
         double wi = gi / den;
         if (dt == HIPK_F32) wi = (double)(float)wi;
         s2 += gi * wi;
      }
      out[c] = s1; out[nx + c] = s2;
   }
   mirror(out, (size_t)2 * nx);
   return 0;
}
int hipk_qmr_update_dir(hipk_ctx *ctx, hipk_dtype dt, int64_t m, int nx, const double *gam, const double *eta, const double *beta,
      void *D, int64_t ldD, void *Delta, int64_t ldDelta, void *Sol, int64_t ldSol, const void *G, int64_t ldG,
      const void *diag, const double *shift, double min_den, double *out) {
   (void)ctx;
   if (!(min_den > 0.0)) min_den = 1e-300;
   for (int c = 0; c < nx; c++) {
      const void *g = colp(dt, G, ldG, c);
      void *d = (void *)colp(dt, D, ldD, c), *de = (void *)colp(dt, Delta, ldDelta, c), *so = (void *)colp(dt, Sol, ldSol, c);
      double s1 = 0.0;
      for (int64_t i = 0; i < m; i++) {
         const double di = ld_(dt, d, i);
         st_(dt, de, i, ld_(dt, de, i) * gam[c] + di * eta[c]);
         st_(dt, so, i, ld_(dt, de, i) + ld_(dt, so, i));
         s1 += ld_(dt, so, i) * ld_(dt, so, i);
         double den = ld_(dt, diag, i) - (shift ? shift[c] : 0.0);
         if (!(fabs(den) > min_den)) den = copysign(min_den, den);
         double wi = ld_(dt, g, i) / den;
         if (dt == HIPK_F32) wi = (double)(float)wi;
         st_(dt, d, i, wi + beta[c] * di);
      }
      out[c] = s1;
   }
   mirror(out, (size_t)nx);
   return 0;
}

int hipk_jacobi_apply(void *stream, hipk_dtype dt, int64_t m, const void *diag, const double *shift,
      double min_den, const void *x, int64_t ldx, void *y, int64_t ldy, int ncols) {
   (void)stream;
   if (!(min_den > 0.0)) min_den = 1e-300;
   if (IS_Z(dt)) return hipk_z_jacobi_apply(dt, m, diag, shift, min_den, x, ldx, y, ldy, ncols);
   for (int c = 0; c < ncols; c++) {
      const void *xc = colp(dt, x, ldx, c); void *yc = (void *)colp(dt, y, ldy, c);
      for (int64_t i = 0; i < m; i++) {
         double d = ld_(dt, diag, i) - (shift ? shift[c] : 0.0);
         if (!(fabs(d) > min_den)) d = copysign(min_den, d);
         st_(dt, yc, i, ld_(dt, xc, i) / d);
      }
   }
   return 0;
}
int hipk_bandwidth_probe(hipk_ctx *ctx, size_t bytes, int reps, double *gbps) { (void)ctx; (void)bytes; (void)reps; *gbps = 0; return 0; }
int hipk_prof_enable(int on) { (void)on; return 0; }
int hipk_prof_reset(void) { return 0; }
int hipk_prof_get(int cls, double *ms, long *launches, double *alg_bytes) { (void)cls; *ms = 0; *launches = 0; *alg_bytes = 0; return 0; }

/* symmetric eigenproblem by cyclic Jacobi (the device kernel's algorithm, sequential pairing) */
int hipk_sym_eig(hipk_ctx *ctx, int n, const double *A_in, int lda, double *evals, double *Z, int ldz) {
   (void)ctx;
   if (n <= 0) return 0;
   double *A = malloc(sizeof(double) * (size_t)n * n), *V = malloc(sizeof(double) * (size_t)n * n);
   for (int j = 0; j < n; j++) for (int i = 0; i < n; i++) {
      A[i + (size_t)j * n] = (i <= j) ? A_in[i + (size_t)j * lda] : A_in[j + (size_t)i * lda];
      V[i + (size_t)j * n] = (i == j);
   }
   for (int sweep = 0; sweep < 60; sweep++) {
      double off = 0, dia = 0;
      for (int j = 0; j < n; j++) for (int i = 0; i < n; i++) { double v = A[i + (size_t)j * n]; if (i == j) dia += v * v; else off += v * v; }
      if (off <= 1e-34 * dia || off == 0.0) break;
      for (int p = 0; p < n - 1; p++) for (int q = p + 1; q < n; q++) {
         double apq = A[p + (size_t)q * n];
         if (fabs(apq) < 1e-300) continue;
         double tau = (A[q + (size_t)q * n] - A[p + (size_t)p * n]) / (2 * apq);
         double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1 + tau * tau)), c = 1 / sqrt(1 + t * t), s = t * c;
         for (int r = 0; r < n; r++) {
            double ap = A[r + (size_t)p * n], aq = A[r + (size_t)q * n];
            A[r + (size_t)p * n] = c * ap - s * aq; A[r + (size_t)q * n] = s * ap + c * aq;
            double vp = V[r + (size_t)p * n], vq = V[r + (size_t)q * n];
            V[r + (size_t)p * n] = c * vp - s * vq; V[r + (size_t)q * n] = s * vp + c * vq;
         }
         for (int r = 0; r < n; r++) {
            double ap = A[p + (size_t)r * n], aq = A[q + (size_t)r * n];
            A[p + (size_t)r * n] = c * ap - s * aq; A[q + (size_t)r * n] = s * ap + c * aq;
         }
      }
   }
   int *perm = malloc(sizeof(int) * (size_t)n);
   for (int i = 0; i < n; i++) perm[i] = i;
   for (int i = 1; i < n; i++) { int pi = perm[i], j = i - 1; while (j >= 0 && A[perm[j] + (size_t)perm[j] * n] > A[pi + (size_t)pi * n]) { perm[j + 1] = perm[j]; j--; } perm[j + 1] = pi; }
   for (int j = 0; j < n; j++) { evals[j] = A[perm[j] + (size_t)perm[j] * n]; for (int i = 0; i < n; i++) Z[i + (size_t)j * ldz] = V[i + (size_t)perm[j] * n]; }
   free(A); free(V); free(perm);
   return 0;
}
