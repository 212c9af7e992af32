/* amd_operator.hip — the ready-made matrixMatvec / applyPreconditioner callbacks
 * (include/primme_amd.h) over a device sparse operator, with the halo exchange of
 * the row-partitioned case.  Replaces the per-application callback code of
 * reference examples/ex_eigs_dhipblas.c:239-264 and examples/ex_eigs_mpi.c:150-207.
 */
#include "hipk_internal.h"
#include "primme_amd.h"
#include "primme_amd_comm.h"
#include "comm_internal.h"

struct primme_amd_operator {
   hipk_csr *A;
   primme_amd_comm *comm;
   int mode;                 /* 0 local, 1 neighbour halo, 2 all-gather */
   int64_t lo, hi;           /* rows needed from below / above */
   int64_t send_lo, send_hi; /* rows the neighbours need from me */
   int64_t max_side;         /* the largest halo of any rank (sizes the peer-to-peer landing zones) */
   void *buf_lo, *buf_hi;    /* device halo buffers (grown on demand) */
   size_t cap_lo, cap_hi;
   void *xfull;              /* all-gather buffer */
   size_t cap_full;
   int64_t row0, nrows, n;
   int jacobi_fixed;         /* 1: K = diag(A) - jacobi_shift, 0: per-vector shifts of the solver */
   double jacobi_shift;
   int ldscale;              /* 2 when A is the real-equivalent form of a Hermitian matrix and the
                                callbacks are handed leading dimensions in complex elements */
};

static size_t op_elem(hipk_dtype dt) { return dt == HIPK_F64 ? 8 : dt == HIPK_F32 ? 4 : dt == HIPK_C64 ? 16 : 8; }

extern "C" int primme_amd_operator_create(primme_amd_operator **out, hipk_csr *A, primme_amd_comm *comm) {
   primme_amd_operator *op = (primme_amd_operator *)calloc(1, sizeof(*op));
   if (!op) return -2;
   op->A = A; op->comm = comm; op->ldscale = 1;
   op->lo = hipk_csr_halo_lo(A); op->hi = hipk_csr_halo_hi(A);
   op->nrows = hipk_csr_nrows(A);
   if ((op->lo > 0 || op->hi > 0) && !comm) {
      fprintf(stderr, "primme_amd: operator references rows outside the local slab but no communicator was given\n");
      free(op);
      return -1;
   }
   if (comm && primme_amd_comm_size(comm) > 1) {
      const int P = primme_amd_comm_size(comm), r = primme_amd_comm_rank(comm);
      int64_t mine[3] = {op->lo, op->hi, op->nrows};
      int64_t *all = (int64_t *)malloc((size_t)P * 3 * sizeof(int64_t));
      if (primme_amd_comm_allgather_i64(comm, mine, 3, all)) { free(all); free(op); return -43; }
      int neighbour_ok = 1, any = 0, equal = 1;
      for (int q = 0; q < P; q++) {
         if (all[3 * q] > 0 || all[3 * q + 1] > 0) any = 1;
         if (all[3 * q] > op->max_side) op->max_side = all[3 * q];
         if (all[3 * q + 1] > op->max_side) op->max_side = all[3 * q + 1];
         if (q > 0 && all[3 * q] > all[3 * (q - 1) + 2]) neighbour_ok = 0;     /* needs more than rank q-1 owns */
         if (q < P - 1 && all[3 * q + 1] > all[3 * (q + 1) + 2]) neighbour_ok = 0;
         if (all[3 * q + 2] != all[2]) equal = 0;
      }
      if (!any) op->mode = 0;
      else if (neighbour_ok) {
         op->mode = 1;
         op->send_hi = (r < P - 1) ? all[3 * (r + 1)] : 0;     /* what rank r+1 needs from below = my last rows */
         op->send_lo = (r > 0) ? all[3 * (r - 1) + 1] : 0;     /* what rank r-1 needs from above = my first rows */
      } else {
         if (!equal) {
            fprintf(stderr, "primme_amd: all-gather matvec needs equal slabs per rank\n");
            free(all); free(op);
            return -1;
         }
         op->mode = 2;
      }
      int64_t r0 = 0, n = 0;
      for (int q = 0; q < P; q++) { if (q < r) r0 += all[3 * q + 2]; n += all[3 * q + 2]; }
      op->row0 = r0; op->n = n;
      free(all);
   }
   *out = op;
   return 0;
}

extern "C" int primme_amd_operator_destroy(primme_amd_operator *op) {
   if (!op) return 0;
   if (op->buf_lo) (void)hipFree(op->buf_lo);
   if (op->buf_hi) (void)hipFree(op->buf_hi);
   if (op->xfull) (void)hipFree(op->xfull);
   free(op);
   return 0;
}

extern "C" hipk_csr *primme_amd_operator_matrix(primme_amd_operator *op) { return op->A; }

static int grow(void **buf, size_t *cap, size_t need) {
   if (need <= *cap) return 0;
   if (*buf) HIPK_CHECK(hipFree(*buf));
   HIPK_CHECK(hipMalloc(buf, need));
   *cap = need;
   return 0;
}

extern "C" int primme_amd_operator_apply(primme_amd_operator *op, void *hip_stream, const void *x,
      int64_t ldx, void *y, int64_t ldy, int ncols) {
   const size_t es = op_elem(hipk_csr_dtype(op->A));
   if (op->mode == 1) {
      if (grow(&op->buf_lo, &op->cap_lo, (size_t)(op->lo > 0 ? op->lo : 1) * ncols * es)) return -2;
      if (grow(&op->buf_hi, &op->cap_hi, (size_t)(op->hi > 0 ? op->hi : 1) * ncols * es)) return -2;
      void *zl = NULL, *zh = NULL;
      int rc = pa_comm_halo_auto(op->comm, hip_stream, x, ldx, op->nrows, ncols, es, op->send_lo,
            op->send_hi, op->buf_lo, op->lo, op->buf_hi, op->hi, op->max_side, &zl, &zh);
      if (rc) return rc;
      hipk_csr_set_halo(op->A, zl, zh);
      return hipk_csr_matvec(op->A, hip_stream, x, ldx, y, ldy, ncols);
   } else if (op->mode == 2) {
      /* unstructured columns: gather the whole block [n x ncols] in ONE grouped exchange (a column per
       * all-gather inside an RCCL group = one launch), then one SpMM; lo = everything below my slab,
       * hi = everything above, both addressed inside the gathered block with column stride n */
      if (grow(&op->xfull, &op->cap_full, (size_t)op->n * ncols * es)) return -2;
      int rc = primme_amd_comm_allgather_cols(op->comm, hip_stream, x, ldx, op->xfull, op->n, (size_t)op->nrows * es, es, ncols);
      if (rc) return rc;
      hipk_csr_set_halo_ld(op->A, (const char *)op->xfull + (size_t)(op->row0 - op->lo) * es, op->n,
            (const char *)op->xfull + (size_t)(op->row0 + op->nrows) * es, op->n);
      return hipk_csr_matvec(op->A, hip_stream, x, ldx, y, ldy, ncols);
   }
   return hipk_csr_matvec(op->A, hip_stream, x, ldx, y, ldy, ncols);
}

/* One-synchronisation GD iteration (eigs_conv.c): y = A (a x), xout = a x, dot_dev[0] = xout' y with
 * a = 1/sqrt(norm2_dev[0]) in ONE launch (hipk_csr_matvec_scaled), halo exchange included.  Only for the
 * solver's own use with this ready-made operator: a user matvec callback is a black box and gets the
 * separate normalisation / operator / inner-product launches instead. */
extern "C" int primme_amd_operator_can_fuse(const primme_amd_operator *op) {
   /* every precondition of hipk_csr_matvec_scaled, so that an eligible tail cannot fail with "not applicable":
    * a CSR matrix whose input entries are its own row slab, and halo data this operator knows how to fetch */
   if (!op || op->ldscale != 1 || !hipk_csr_fusable(op->A)) return 0;
   if (op->mode == 0 && (hipk_csr_halo_lo(op->A) > 0 || hipk_csr_halo_hi(op->A) > 0)) return 0;
   return 1;
}
extern "C" int primme_amd_operator_apply_scaled(primme_amd_operator *op, hipk_ctx *ctx, const void *x,
      const double *norm2_dev, void *xout, void *y, double *dot_dev) {
   if (!primme_amd_operator_can_fuse(op) || !ctx) return -1;
   void *hip_stream = hipk_ctx_stream(ctx);
   const size_t es = op_elem(hipk_csr_dtype(op->A));
   if (op->mode == 1) {
      if (grow(&op->buf_lo, &op->cap_lo, (size_t)(op->lo > 0 ? op->lo : 1) * es)) return -2;
      if (grow(&op->buf_hi, &op->cap_hi, (size_t)(op->hi > 0 ? op->hi : 1) * es)) return -2;
      void *zl = NULL, *zh = NULL;
      int rc = pa_comm_halo_auto(op->comm, hip_stream, x, op->nrows, op->nrows, 1, es, op->send_lo,
            op->send_hi, op->buf_lo, op->lo, op->buf_hi, op->hi, op->max_side, &zl, &zh);
      if (rc) return rc;
      hipk_csr_set_halo(op->A, zl, zh);
   } else if (op->mode == 2) {
      if (grow(&op->xfull, &op->cap_full, (size_t)op->n * es)) return -2;
      int rc = primme_amd_comm_allgather(op->comm, hip_stream, x, op->xfull, (size_t)op->nrows * es);
      if (rc) return rc;
      hipk_csr_set_halo(op->A, (const char *)op->xfull + (size_t)(op->row0 - op->lo) * es,
            (const char *)op->xfull + (size_t)(op->row0 + op->nrows) * es);
   }
   return hipk_csr_matvec_scaled(op->A, ctx, x, norm2_dev, xout, y, dot_dev);
}

extern "C" int primme_amd_operator_apply_shifted(primme_amd_operator *op, void *hip_stream, const void *x, int64_t ldx,
      void *y, int64_t ldy, int ncols, const double *shifts_host) {
   if (!op || op->mode != 0 || op->ldscale != 1) return 1;
   return hipk_csr_matvec_shifted(op->A, hip_stream, x, ldx, y, ldy, ncols, shifts_host);
}
extern "C" int primme_amd_operator_jacobi_data(primme_amd_operator *op, const void **diag, int *fixed, double *shift) {
   if (!op || op->ldscale != 1) return 1;
   *diag = hipk_csr_diag(op->A); *fixed = op->jacobi_fixed; *shift = op->jacobi_shift;
   return 0;
}

/* ---- the callbacks ---------------------------------------------------------- */
extern "C" void primme_amd_mass_matvec(void *x, PRIMME_INT *ldx, void *y, PRIMME_INT *ldy, int *blockSize,
      struct primme_params *primme, int *ierr) {
   primme_amd_operator *op = (primme_amd_operator *)primme->massMatrix;
   void *stream = primme->queue ? (void *)*(hipStream_t *)primme->queue : NULL;
   *ierr = op ? primme_amd_operator_apply(op, stream, x, *ldx * op->ldscale, y, *ldy * op->ldscale, *blockSize) : 1;
}
extern "C" void primme_amd_matvec(void *x, PRIMME_INT *ldx, void *y, PRIMME_INT *ldy, int *blockSize,
      struct primme_params *primme, int *ierr) {
   primme_amd_operator *op = (primme_amd_operator *)primme->matrix;
   void *stream = primme->queue ? (void *)*(hipStream_t *)primme->queue : NULL;
   *ierr = op ? primme_amd_operator_apply(op, stream, x, *ldx * op->ldscale, y, *ldy * op->ldscale, *blockSize) : 1;
}

extern "C" void primme_amd_jacobi_precond(void *x, PRIMME_INT *ldx, void *y, PRIMME_INT *ldy,
      int *blockSize, struct primme_params *primme, int *ierr) {
   primme_amd_operator *op = (primme_amd_operator *)primme->preconditioner;
   void *stream = primme->queue ? (void *)*(hipStream_t *)primme->queue : NULL;
   if (!op) { *ierr = 1; return; }
   const hipk_dtype dt = hipk_csr_dtype(op->A);
   const size_t es = op_elem(dt);
   const int64_t lx = *ldx * op->ldscale, ly = *ldy * op->ldscale;
   *ierr = 0;
   for (int c0 = 0; c0 < *blockSize && !*ierr; c0 += 64) {
      const int n = *blockSize - c0 < 64 ? *blockSize - c0 : 64;
      double fixed[64];
      for (int c = 0; c < n; c++) fixed[c] = op->jacobi_shift;
      *ierr = hipk_jacobi_apply(stream, dt, hipk_csr_nrows(op->A), hipk_csr_diag(op->A),
            op->jacobi_fixed ? fixed : primme->ShiftsForPreconditioner + c0,
            1e-14 * (primme->aNorm >= 0.0 ? primme->aNorm : 1.0), (const char *)x + (size_t)c0 * lx * es, lx,
            (char *)y + (size_t)c0 * ly * es, ly, n);
   }
}

extern "C" int primme_amd_operator_set_complex(primme_amd_operator *op, int on) {
   if (!op) return -1;
   op->ldscale = on ? 2 : 1;
   return 0;
}

extern "C" int primme_amd_operator_set_jacobi(primme_amd_operator *op, int fixed, double shift) {
   if (!op) return -1;
   op->jacobi_fixed = fixed; op->jacobi_shift = shift;
   return 0;
}


/* ---- singular value operator: A and A' both resident in CSR ------------------------------ */
#include "primme_amd_svds.h"
#include "primme_amd_io.h"
struct primme_amd_svds_operator {
   hipk_csr *A, *At;
   primme_amd_comm *comm;        /* NULL: single rank */
   int64_t mLocal, n, nLocal;
   void *full;                   /* n-vector staging (all-gather target / reduce-scatter source) */
   size_t full_cap;
   hipk_dtype dt;
   void *jac_r, *jac_c;          /* Jacobi for the normal equations: row / column sums of squares - shift^2 */
   int cplx;                     /* real-equivalent form of a complex matrix: leading dimensions arrive in complex elements */
};

extern "C" int primme_amd_svds_operator_create(primme_amd_svds_operator **out, hipk_ctx *ctx, int dt,
      int64_t m, int64_t n, const int32_t *rp, const int32_t *ci, const void *val) {
   primme_amd_svds_operator *op = (primme_amd_svds_operator *)calloc(1, sizeof(*op));
   if (!op) return -2;
   const size_t es = (dt == HIPK_F64) ? 8 : 4;
   int32_t *rpT = NULL, *ciT = NULL;
   void *vT = NULL;
   int rc = hipk_csr_create_rect(ctx, (hipk_dtype)dt, m, n, rp, ci, val, &op->A);
   if (!rc) rc = primme_amd_csr_transpose(m, n, rp, ci, val, es, &rpT, &ciT, &vT);
   if (!rc) rc = hipk_csr_create_rect(ctx, (hipk_dtype)dt, n, m, rpT, ciT, vT, &op->At);
   primme_amd_host_free(rpT); primme_amd_host_free(ciT); primme_amd_host_free(vT);
   if (rc) { if (op->A) hipk_csr_destroy(op->A); free(op); return rc; }
   *out = op;
   return 0;
}
extern "C" int primme_amd_svds_operator_create_dist(primme_amd_svds_operator **out, hipk_ctx *ctx, int dt,
      int64_t mLocal, int64_t n, int64_t nLocal, const int32_t *rp, const int32_t *ci, const void *val, void *comm) {
   primme_amd_comm *c = (primme_amd_comm *)comm;
   if (!c || nLocal * primme_amd_comm_size(c) != n) return -1;     /* equal column slabs */
   int rc = primme_amd_svds_operator_create(out, ctx, dt, mLocal, n, rp, ci, val);
   if (rc) return rc;
   (*out)->comm = c; (*out)->mLocal = mLocal; (*out)->n = n; (*out)->nLocal = nLocal; (*out)->dt = (hipk_dtype)dt;
   return 0;
}

extern "C" int primme_amd_svds_operator_destroy(primme_amd_svds_operator *op) {
   if (!op) return 0;
   if (op->full) (void)hipFree(op->full);
   if (op->jac_r) (void)hipFree(op->jac_r);
   if (op->jac_c) (void)hipFree(op->jac_c);
   hipk_csr_destroy(op->A); hipk_csr_destroy(op->At);
   free(op);
   return 0;
}
/* diag(A A') - shift^2 and diag(A'A) - shift^2 (reference tests/COMMON/mat.c:353-393, the driver's
 * "jacobi" choice for singular value problems); single-rank operators */
extern "C" int primme_amd_svds_operator_set_jacobi(primme_amd_svds_operator *op, const int32_t *rp,
      const int32_t *ci, const void *val, double shift) {
   if (!op || op->comm) return -44;
   const int64_t m = hipk_csr_nrows(op->A), n = hipk_csr_nrows(op->At);
   const hipk_dtype dt = hipk_csr_dtype(op->A);
   const size_t es = (dt == HIPK_F64) ? 8 : 4;
   double *sum = (double *)calloc((size_t)(m + n) + 1, sizeof(double));
   char *packed = (char *)malloc(es * (size_t)(m + n) + 1);
   if (!sum || !packed) { free(sum); free(packed); return -2; }
   for (int64_t i = 0; i < m; i++)
      for (int32_t k = rp[i]; k < rp[i + 1]; k++) {
         const double v = (dt == HIPK_F64) ? ((const double *)val)[k] : (double)((const float *)val)[k];
         sum[i] += v * v;
         sum[m + ci[k]] += v * v;
      }
   for (int64_t i = 0; i < m + n; i++) {
      double d = sum[i] - shift * shift;
      if (fabs(d) < 1e-14) d = copysign(1e-14, d);
      if (dt == HIPK_F64) ((double *)packed)[i] = d; else ((float *)packed)[i] = (float)d;
   }
   free(sum);
   if (!op->jac_r && hipMalloc(&op->jac_r, es * (size_t)(m > 0 ? m : 1)) != hipSuccess) { free(packed); return -2; }
   if (!op->jac_c && hipMalloc(&op->jac_c, es * (size_t)(n > 0 ? n : 1)) != hipSuccess) { free(packed); return -2; }
   /* through pinned staging on the stream the operator's kernels run on (hipk_upload: never the NULL stream) */
   hipk_ctx *octx = hipk_csr_ctx(op->A);
   const int e1 = m > 0 ? hipk_upload(octx, op->jac_r, packed, es * (size_t)m) : 0;
   const int e2 = n > 0 ? hipk_upload(octx, op->jac_c, packed + es * (size_t)m, es * (size_t)n) : 0;
   free(packed);
   return (e1 == 0 && e2 == 0) ? 0 : -1;
}

/* applyPreconditioner of primme_svds_params for the operator's Jacobi data: y = x / diag(A'A),
 * x / diag(AA') or both halves for the augmented operator (mat.c:395-426) */
extern "C" void primme_amd_svds_jacobi_precond(void *x, PRIMME_INT *ldx, void *y, PRIMME_INT *ldy, int *blockSize,
      int *mode, struct primme_svds_params *ps, int *ierr) {
   primme_amd_svds_operator *op = (primme_amd_svds_operator *)ps->preconditioner;
   *ierr = 1;
   if (!op || !op->jac_r || !op->jac_c) return;
   void *stream = ps->queue ? (void *)*(hipStream_t *)ps->queue : NULL;
   const hipk_dtype dt = hipk_csr_dtype(op->A);
   const size_t es = (dt == HIPK_F64) ? 8 : 4;
   const double min_den = 1e-14 * (ps->aNorm >= 0.0 ? ps->aNorm : 1.0);
   const int64_t m = ps->mLocal, n = ps->nLocal;
   int rc = 0;
   for (int c0 = 0; c0 < *blockSize && !rc; c0 += 64) {
      const int nb = *blockSize - c0 < 64 ? *blockSize - c0 : 64;
      double zeros[64] = {0};
      const char *xc = (const char *)x + (size_t)c0 * *ldx * es;
      char *yc = (char *)y + (size_t)c0 * *ldy * es;
      if (*mode == primme_svds_op_AtA) rc = hipk_jacobi_apply(stream, dt, n, op->jac_c, zeros, min_den, xc, *ldx, yc, *ldy, nb);
      else if (*mode == primme_svds_op_AAt) rc = hipk_jacobi_apply(stream, dt, m, op->jac_r, zeros, min_den, xc, *ldx, yc, *ldy, nb);
      else if (*mode == primme_svds_op_augmented) {
         rc = hipk_jacobi_apply(stream, dt, n, op->jac_c, zeros, min_den, xc, *ldx, yc, *ldy, nb);
         if (!rc) rc = hipk_jacobi_apply(stream, dt, m, op->jac_r, zeros, min_den, xc + (size_t)n * es, *ldx, yc + (size_t)n * es, *ldy, nb);
      } else rc = 1;
   }
   *ierr = rc ? 1 : 0;
}

extern "C" int primme_amd_svds_operator_is_local(const void *op) { return op && ((const primme_amd_svds_operator *)op)->comm == NULL; }

extern "C" int primme_amd_svds_operator_set_complex(primme_amd_svds_operator *op, int on) {
   if (!op) return -1;
   op->cplx = on ? 1 : 0;
   return 0;
}

extern "C" void primme_amd_svds_matvec(void *x, PRIMME_INT *ldx, void *y, PRIMME_INT *ldy, int *blockSize,
      int *transpose, struct primme_svds_params *ps, int *ierr) {
   primme_amd_svds_operator *op = (primme_amd_svds_operator *)ps->matrix;
   void *stream = ps->queue ? (void *)*(hipStream_t *)ps->queue : NULL;
   *ierr = 1;
   if (!op) return;
   if (!op->comm) {
      const int64_t f = op->cplx ? 2 : 1;
      *ierr = hipk_csr_matvec(*transpose ? op->At : op->A, stream, x, f * *ldx, y, f * *ldy, *blockSize);
      return;
   }
   if (op->cplx) return;
   /* row-partitioned A: the block goes through the [n x blockSize] staging panel with ONE grouped
    * collective (a column per call inside an RCCL group) and ONE SpMM per application */
   const size_t es = (op->dt == HIPK_F64) ? 8 : 4;
   const int nb = *blockSize;
   if (nb <= 0) { *ierr = 0; return; }
   if (grow(&op->full, &op->full_cap, (size_t)op->n * nb * es)) return;
   if (!*transpose) {
      if (primme_amd_comm_allgather_cols(op->comm, stream, x, *ldx, op->full, op->n, (size_t)op->nLocal * es, es, nb)) return;
      if (hipk_csr_matvec(op->A, stream, op->full, op->n, y, *ldy, nb)) return;
   } else {
      if (hipk_csr_matvec(op->At, stream, x, *ldx, op->full, op->n, nb)) return;
      if (primme_amd_comm_reduce_scatter_cols(op->comm, stream, op->full, op->n, y, *ldy, (size_t)op->nLocal, op->dt == HIPK_F64, nb)) return;
   }
   *ierr = 0;
}
