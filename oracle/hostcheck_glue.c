/* oracle/hostcheck_glue.c — TEST INFRASTRUCTURE.  Host-memory stand-ins for the
 * product's HIP-side callbacks (primme_amd/csrc/amd_operator.hip, comm_rccl.hip)
 * so that libprimme_hostcheck.so (product host solver + oracle/hipk_cpu.c) links
 * and the solver's control flow can run in `-m "not gpu"` tests.  Multi-rank
 * reductions in those tests go through a user globalSumReal callback (gloo). */
#include <stdlib.h>
#include "primme_amd.h"
#include "primme_amd_comm.h"
#include "primme_amd_svds.h"
#include "primme_amd_io.h"
#include <math.h>

typedef void (*hostcheck_allreduce_fn)(double *buf, int count);
struct primme_amd_comm { hostcheck_allreduce_fn cb; int rank, size; long calls; int xr; long xr_calls; };
/* comm / xfull: a row slab that references rows of other ranks (halo) gathers the whole vector through the stand-in
 * communicator's all-reduce (every rank contributes its slab to a zero-padded copy) and hands the kernels the two
 * halo windows inside it, like the all-gather mode of the real operator (amd_operator.hip) */
struct primme_amd_operator { hipk_csr *A; int jacobi_fixed; double jacobi_shift; int ldscale; primme_amd_comm *comm; double *xfull; int64_t n, row0, nrows; };
int primme_amd_operator_set_complex(primme_amd_operator *op, int on) { op->ldscale = on ? 2 : 1; return 0; }
int primme_amd_operator_set_jacobi(primme_amd_operator *op, int fixed, double shift) { op->jacobi_fixed = fixed; op->jacobi_shift = shift; return 0; }
int primme_amd_operator_create(primme_amd_operator **op, hipk_csr *A, primme_amd_comm *c) {
   *op = calloc(1, sizeof(**op)); (*op)->A = A; (*op)->ldscale = 1;
   if (c && c->size > 1 && (hipk_csr_halo_lo(A) > 0 || hipk_csr_halo_hi(A) > 0)) {
      /* global size and my first row: sum / prefix of the slab sizes through the all-reduce */
      double *cnt = calloc((size_t)c->size, sizeof(double));
      cnt[c->rank] = (double)hipk_csr_nrows(A);
      c->cb(cnt, c->size);
      (*op)->comm = c; (*op)->nrows = hipk_csr_nrows(A);
      for (int r = 0; r < c->size; r++) { if (r < c->rank) (*op)->row0 += (int64_t)cnt[r]; (*op)->n += (int64_t)cnt[r]; }
      free(cnt);
      (*op)->xfull = calloc((size_t)(*op)->n * 64, sizeof(double));
   }
   return 0;
}
int primme_amd_operator_destroy(primme_amd_operator *op) { if (op) free(op->xfull); free(op); return 0; }
/* x (nrows x nc, ld ldx, doubles) -> xfull (n x nc) on every rank; halo windows set on the matrix */
static int gather_halo(primme_amd_operator *op, const void *x, int64_t ldx, int nc) {
   if (!op->comm) return 0;
   if (nc > 64 || hipk_csr_dtype(op->A) != HIPK_F64) return -44;
   const int64_t n = op->n;
   for (int c = 0; c < nc; c++) {
      for (int64_t i = 0; i < n; i++) op->xfull[i + (size_t)c * n] = 0.0;
      for (int64_t i = 0; i < op->nrows; i++) op->xfull[op->row0 + i + (size_t)c * n] = ((const double *)x)[i + (size_t)c * ldx];
   }
   op->comm->cb(op->xfull, (int)(n * nc));
   const int64_t lo = hipk_csr_halo_lo(op->A);
   return hipk_csr_set_halo_ld(op->A, op->xfull + (op->row0 - lo), n, op->xfull + op->row0 + op->nrows, n);
}
hipk_csr *primme_amd_operator_matrix(primme_amd_operator *op) { return op->A; }
int primme_amd_operator_apply(primme_amd_operator *op, void *st, const void *x, int64_t ldx, void *y, int64_t ldy, int nc) {
   int rc = gather_halo(op, x, ldx, nc);
   if (rc) return rc;
   return hipk_csr_matvec(op->A, st, x, ldx, y, ldy, nc);
}
int primme_amd_operator_can_fuse(const primme_amd_operator *op) { return op && op->ldscale == 1 && hipk_csr_kind(op->A) == 0; }
int primme_amd_operator_apply_scaled(primme_amd_operator *op, hipk_ctx *ctx, const void *x, const double *norm2, void *xout,
      void *y, double *dot) {
   if (!primme_amd_operator_can_fuse(op)) return -1;
   int rc = gather_halo(op, x, hipk_csr_nrows(op->A), 1);        /* the un-normalised vector: the kernel scales the halo entries too */
   if (rc) return rc;
   return hipk_csr_matvec_scaled(op->A, ctx, x, norm2, xout, y, dot);
}
int primme_amd_operator_apply_shifted(primme_amd_operator *op, void *st, const void *x, int64_t ldx, void *y, int64_t ldy,
      int nc, const double *shifts) {
   if (!op || op->ldscale != 1) return 1;
   return hipk_csr_matvec_shifted(op->A, st, x, ldx, y, ldy, nc, shifts);
}
int primme_amd_operator_jacobi_data(primme_amd_operator *op, const void **diag, int *fixed, double *shift) {
   if (!op || op->ldscale != 1) return 1;
   *diag = hipk_csr_diag(op->A); *fixed = op->jacobi_fixed; *shift = op->jacobi_shift;
   return 0;
}
void primme_amd_matvec(void *x, PRIMME_INT *ldx, void *y, PRIMME_INT *ldy, int *bs, struct primme_params *p, int *ierr) {
   primme_amd_operator *op = (primme_amd_operator *)p->matrix;
   *ierr = primme_amd_operator_apply(op, NULL, x, *ldx * op->ldscale, y, *ldy * op->ldscale, *bs);
}
void primme_amd_mass_matvec(void *x, PRIMME_INT *ldx, void *y, PRIMME_INT *ldy, int *bs, struct primme_params *p, int *ierr) {
   primme_amd_operator *op = (primme_amd_operator *)p->massMatrix;
   *ierr = op ? primme_amd_operator_apply(op, NULL, x, *ldx * op->ldscale, y, *ldy * op->ldscale, *bs) : 1;
}
void primme_amd_jacobi_precond(void *x, PRIMME_INT *ldx, void *y, PRIMME_INT *ldy, int *bs, struct primme_params *p, int *ierr) {
   primme_amd_operator *op = (primme_amd_operator *)p->preconditioner;
   double fixed[64];
   for (int c = 0; c < *bs && c < 64; c++) fixed[c] = op->jacobi_shift;
   *ierr = hipk_jacobi_apply(NULL, hipk_csr_dtype(op->A), hipk_csr_nrows(op->A), hipk_csr_diag(op->A),
         op->jacobi_fixed ? fixed : p->ShiftsForPreconditioner, 1e-14 * (p->aNorm >= 0.0 ? p->aNorm : 1.0),
         x, *ldx * op->ldscale, y, *ldy * op->ldscale, *bs);
}
/* Stand-in for the RCCL communicator (comm_rccl.hip): the all-reduce is a callback of the test (gloo), applied in
 * place to the "device" buffer, which is host memory here.  With it installed the solver runs the SAME code path as
 * on several GPUs -- reductions inside the stream of launches, |t|^2 and t'At in one all-reduce, the fused /
 * speculative restart with reduced overlaps -- in the world_size-2 CPU tests (tests/test_multirank_gloo.py). */
/* (the stand-in of the cross-rank second stage, below) */
extern void (*hipk_cpu_xr_hook)(double *out, size_t cnt);
static primme_amd_comm *g_xr_comm; static int g_xr_armed; static const double *g_xr_lo; static size_t g_xr_count;
int primme_amd_hostcheck_comm_create(primme_amd_comm **out, hostcheck_allreduce_fn cb, int rank, int size) {
   primme_amd_comm *c = calloc(1, sizeof(*c));
   if (!c) return -2;
   c->cb = cb; c->rank = rank; c->size = size;
   *out = c;
   return 0;
}
int primme_amd_comm_destroy(primme_amd_comm *c) { if (c == g_xr_comm) { g_xr_comm = NULL; hipk_cpu_xr_hook = NULL; } free(c); return 0; }
int primme_amd_comm_rank(const primme_amd_comm *c) { return c ? c->rank : 0; }
int primme_amd_comm_size(const primme_amd_comm *c) { return c ? c->size : 1; }
long primme_amd_hostcheck_comm_calls(const primme_amd_comm *c) { return c ? c->calls : 0; }
void primme_amd_global_sum(void *s, void *r, int *c, struct primme_params *p, int *ierr) {
   primme_amd_comm *cm = (primme_amd_comm *)p->commInfo;
   if (!cm || !cm->cb) { *ierr = 1; return; }
   if (s != r) for (int i = 0; i < *c; i++) ((double *)r)[i] = ((const double *)s)[i];
   cm->cb((double *)r, *c); cm->calls++;
   *ierr = 0;
}
int pa_comm_allreduce_device(void *ci, double *d, int n, void *st) {
   primme_amd_comm *cm = (primme_amd_comm *)ci;
   (void)st;
   if (!cm || !cm->cb) return -43;
   cm->cb(d, n); cm->calls++;
   return 0;
}

/* The peer-to-peer transport exists on the device only: the checker takes the separate all-reduce above.  Its FUSED SECOND STAGE
 * (hipk_xreduce_arm: the launch that forms the local sums exchanges them with the other ranks itself) has a stand-in, switched on
 * per communicator with primme_amd_hostcheck_comm_set_xr: an armed reduction of oracle/hipk_cpu.c passes its results through the
 * all-reduce callback before it mirrors them (hipk_cpu_xr_hook), and hipk_xreduce_covered answers as on the device.  With it the
 * world_size-2 CPU tests run the host logic of the row-partitioned run on the mailboxes, including the iteration that is enqueued
 * before the host has seen the current one. */
int pa_comm_allreduce_publish(void *ci, struct hipk_ctx *ctx, double *d, int n) { (void)ci; (void)ctx; (void)d; (void)n; return 1; }
static void xr_hook(double *out, size_t cnt) {
   if (!g_xr_armed || !g_xr_comm) return;
   g_xr_armed = 0;
   g_xr_comm->cb(out, (int)cnt); g_xr_comm->calls++; g_xr_comm->xr_calls++;
   g_xr_lo = out; g_xr_count = cnt;
}
int primme_amd_hostcheck_comm_set_xr(primme_amd_comm *c, int on) { if (!c) return -1; c->xr = on ? 1 : 0; return 0; }
long primme_amd_hostcheck_comm_xr_calls(const primme_amd_comm *c) { return c ? c->xr_calls : 0; }
int pa_comm_attach_ctx(void *ci, struct hipk_ctx *ctx) {
   primme_amd_comm *c = (primme_amd_comm *)ci;
   (void)ctx;
   g_xr_comm = (c && c->xr) ? c : NULL; g_xr_armed = 0; g_xr_lo = NULL; g_xr_count = 0;
   hipk_cpu_xr_hook = g_xr_comm ? xr_hook : NULL;
   return g_xr_comm ? 0 : 1;
}
void hipk_xreduce_arm(struct hipk_ctx *ctx) { (void)ctx; if (g_xr_comm) g_xr_armed = 1; }
int hipk_xreduce_available(struct hipk_ctx *ctx) { (void)ctx; return g_xr_comm ? 1 : 0; }
int hipk_xreduce_covered(struct hipk_ctx *ctx, const double *buf, int count) {
   (void)ctx;
   const int yes = g_xr_lo && buf >= g_xr_lo && buf + count <= g_xr_lo + g_xr_count;
   g_xr_lo = NULL; g_xr_count = 0; g_xr_armed = 0;
   return yes;
}
int pa_comm_failed(void *ci) { (void)ci; return 0; }

/* singular value operator on host memory */
struct primme_amd_svds_operator { hipk_csr *A, *At; void *jac_r, *jac_c; int cplx; };
int primme_amd_svds_operator_create(primme_amd_svds_operator **out, struct hipk_ctx *ctx, int dt, int64_t m, int64_t n,
      const int32_t *rp, const int32_t *ci, const void *val) {
   primme_amd_svds_operator *op = calloc(1, sizeof(*op));
   int32_t *rpT = NULL, *ciT = NULL; void *vT = NULL;
   int rc = hipk_csr_create_rect((hipk_ctx *)ctx, (hipk_dtype)dt, m, n, rp, ci, val, &op->A);
   if (!rc) rc = primme_amd_csr_transpose(m, n, rp, ci, val, dt == HIPK_F64 ? 8 : 4, &rpT, &ciT, &vT);
   if (!rc) rc = hipk_csr_create_rect((hipk_ctx *)ctx, (hipk_dtype)dt, n, m, rpT, ciT, vT, &op->At);
   primme_amd_host_free(rpT); primme_amd_host_free(ciT); primme_amd_host_free(vT);
   *out = op;
   return rc;
}
int primme_amd_svds_operator_destroy(primme_amd_svds_operator *op) {
   if (op) { hipk_csr_destroy(op->A); hipk_csr_destroy(op->At); free(op->jac_r); free(op->jac_c); free(op); }
   return 0;
}
void primme_amd_svds_matvec(void *x, PRIMME_INT *ldx, void *y, PRIMME_INT *ldy, int *bs, int *transpose,
      struct primme_svds_params *ps, int *ierr) {
   primme_amd_svds_operator *op = (primme_amd_svds_operator *)ps->matrix;
   const int64_t f = op->cplx ? 2 : 1;
   *ierr = hipk_csr_matvec(*transpose ? op->At : op->A, NULL, x, f * *ldx, y, f * *ldy, *bs);
}
int primme_amd_svds_operator_is_local(const void *op) { return op != NULL; }
int primme_amd_svds_operator_set_complex(primme_amd_svds_operator *op, int on) { if (!op) return -1; op->cplx = on ? 1 : 0; return 0; }

int primme_amd_svds_operator_set_jacobi(primme_amd_svds_operator *op, const int32_t *rp, const int32_t *ci,
      const void *val, double shift) {
   const int64_t m = hipk_csr_nrows(op->A), n = hipk_csr_nrows(op->At);
   const hipk_dtype dt = hipk_csr_dtype(op->A);
   const size_t es = dt == HIPK_F64 ? 8 : 4;
   double *sum = calloc((size_t)(m + n) + 1, sizeof(double));
   for (int64_t i = 0; i < m; i++)
      for (int32_t k = rp[i]; k < rp[i + 1]; k++) {
         const double v = dt == HIPK_F64 ? ((const double *)val)[k] : (double)((const float *)val)[k];
         sum[i] += v * v; sum[m + ci[k]] += v * v;
      }
   free(op->jac_r); free(op->jac_c);
   op->jac_r = malloc(es * (size_t)(m + 1)); op->jac_c = malloc(es * (size_t)(n + 1));
   for (int64_t i = 0; i < m + n; i++) {
      double d = sum[i] - shift * shift;
      if (fabs(d) < 1e-14) d = copysign(1e-14, d);
      void *dst = i < m ? op->jac_r : op->jac_c;
      const int64_t j = i < m ? i : i - m;
      if (dt == HIPK_F64) ((double *)dst)[j] = d; else ((float *)dst)[j] = (float)d;
   }
   free(sum);
   return 0;
}
void primme_amd_svds_jacobi_precond(void *x, PRIMME_INT *ldx, void *y, PRIMME_INT *ldy, int *bs, int *mode,
      struct primme_svds_params *ps, int *ierr) {
   primme_amd_svds_operator *op = (primme_amd_svds_operator *)ps->preconditioner;
   const hipk_dtype dt = hipk_csr_dtype(op->A);
   const size_t es = dt == HIPK_F64 ? 8 : 4;
   const double md = 1e-14 * (ps->aNorm >= 0.0 ? ps->aNorm : 1.0);
   const int64_t m = ps->mLocal, n = ps->nLocal;
   double *zeros = calloc((size_t)*bs + 1, sizeof(double));
   int rc = 1;
   if (*mode == primme_svds_op_AtA) rc = hipk_jacobi_apply(NULL, dt, n, op->jac_c, zeros, md, x, *ldx, y, *ldy, *bs);
   else if (*mode == primme_svds_op_AAt) rc = hipk_jacobi_apply(NULL, dt, m, op->jac_r, zeros, md, x, *ldx, y, *ldy, *bs);
   else if (*mode == primme_svds_op_augmented) {
      rc = hipk_jacobi_apply(NULL, dt, n, op->jac_c, zeros, md, x, *ldx, y, *ldy, *bs);
      if (!rc) rc = hipk_jacobi_apply(NULL, dt, m, op->jac_r, zeros, md, (const char *)x + (size_t)n * es, *ldx,
                                      (char *)y + (size_t)n * es, *ldy, *bs);
   }
   free(zeros);
   *ierr = rc ? 1 : 0;
}

int primme_amd_svds_operator_create_dist(primme_amd_svds_operator **op, struct hipk_ctx *ctx, int dt, int64_t mLocal,
      int64_t n, int64_t nLocal, const int32_t *rp, const int32_t *ci, const void *val, void *comm) {
   (void)op; (void)ctx; (void)dt; (void)mLocal; (void)n; (void)nLocal; (void)rp; (void)ci; (void)val; (void)comm;
   return -43;   /* RCCL only exists in the product library */
}
void primme_amd_svds_global_sum(void *s, void *r, int *c, struct primme_svds_params *ps, int *ierr) {
   (void)s; (void)r; (void)c; (void)ps; *ierr = 1;
}
int primme_amd_comm_reduce_scatter(primme_amd_comm *c, void *st, const void *s, void *r, size_t n, int d) {
   (void)c; (void)st; (void)s; (void)r; (void)n; (void)d; return -43;
}
