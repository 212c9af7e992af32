/* svds_main.c — singular triplets through the normal equations, on the device.
 *
 *   hip_dprimme_svds / hip_sprimme_svds  <- reference src/svds/primme_svds_c.c:219-224, :271-366,
 *                                           wrapper_svds :388-540
 *   stage_begin   <- copy_last_params_from_svds :547-854 (normal-equation branches)
 *   stage_end     <- copy_last_params_to_svds   :856-1025
 *   pa_svds_matvec_eigs     <- matrixMatvec_eigs :1323-1385   (y = A'(A x) or A(A' x))
 *   pa_svds_conv_test_ata   <- convTestFunATA    :1640-1690
 *   pa_svds_default_conv_test <- default_convTestFun :1594-1622
 *
 * The eigenproblem itself is the same device path as hip_dprimme (eigs_main.c).  What this file
 * adds runs on the solver's stream too: the intermediate vector of the two-step operator stays in
 * HBM (allocated once per solve, not per call as the reference does), U = A V / sigma is one SpMM
 * plus a column scaling, and the [U | V] layout shuffles are device-to-device copies.
 *
 * Every method (normal equations, augmented, hybrid) and target runs on the device path; an error of
 * the eigensolver stage comes back offset by -100 / -200 like in the reference.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "eigs_solver.h"
#include "primme_amd_svds.h"

int pa_eigs_solve(void *evals_out, void *evecs, void *resNorms_out, primme_params *p, hipk_dtype dt, int out_double);

/* per-solve device state, found again from the callbacks through the params address */
typedef struct {
   primme_svds_params *key;
   hipk_ctx *ctx;
   hipk_dtype dt;
   char *aux;          /* intermediate of the two-step operator */
   size_t aux_cols;
} svds_side;
#define MAX_SIDES 8
static svds_side g_sides[MAX_SIDES];

static svds_side *side_of(primme_svds_params *ps) {
   for (int i = 0; i < MAX_SIDES; i++) if (g_sides[i].key == ps) return &g_sides[i];
   return NULL;
}
static svds_side *side_new(primme_svds_params *ps) {
   for (int i = 0; i < MAX_SIDES; i++) if (!g_sides[i].key) { memset(&g_sides[i], 0, sizeof(svds_side)); g_sides[i].key = ps; return &g_sides[i]; }
   return NULL;
}

/* Complex panels (hip_zprimme_svds / hip_cprimme_svds, reference primme_svds_c.c instantiated for SCALAR = complex): every
 * n-length operation of THIS file has real coefficients — copies, scalings by real factors, squared norms, Re(u'A v), an
 * axpy with a real factor — so it runs on the complex vectors viewed as real ones of twice the length (RDT / RF below); the
 * complex arithmetic is the eigensolver's (pa_eigs_solve_z: native complex panels) and the user's operator's.  Leading
 * dimensions handed to the callbacks count complex elements, as in the reference. */
static int is_cplx(hipk_dtype dt) { return dt == HIPK_C64 || dt == HIPK_C32; }
static int is_single(hipk_dtype dt) { return dt == HIPK_F32 || dt == HIPK_C32; }
static hipk_dtype rdt_of(hipk_dtype dt) { return dt == HIPK_C64 ? HIPK_F64 : dt == HIPK_C32 ? HIPK_F32 : dt; }
static size_t es_of(hipk_dtype dt) { return dt == HIPK_C64 ? 16 : (dt == HIPK_F64 || dt == HIPK_C32) ? 8 : 4; }
#define RDT(sd) rdt_of((sd)->dt)
#define RF(sd) ((PRIMME_INT)(is_cplx((sd)->dt) ? 2 : 1))
int pa_eigs_solve_z(void *evals_out, void *evecs, void *resNorms_out, primme_params *p, hipk_dtype dt, int out_double);

/* W(:, 0:nb) = A V or A' V through the user's operator (reference :1116-1172) */
static int svds_matvec(primme_svds_params *ps, void *V, PRIMME_INT ldV, void *W, PRIMME_INT ldW, int nb, int transpose) {
   if (nb <= 0) return 0;
   double t0 = pa_wtime();
   int ierr = 0;
   ps->matrixMatvec(V, &ldV, W, &ldW, &nb, &transpose, ps, &ierr);
   if (ierr) return PRIMME_USER_FAILURE;
   ps->stats.timeMatvec += pa_wtime() - t0;
   ps->stats.numMatvecs += nb;
   return 0;
}

static int true_res_norm(primme_svds_params *ps, svds_side *sd, char *u, char *v, double *rNorm);

/* the eigensolver's operator: A'A x (method AtA), A A' x (method AAt) or [0 A'; A 0] x */
void pa_svds_matvec_eigs(void *x, PRIMME_INT *ldx, void *y, PRIMME_INT *ldy, int *blockSize,
      primme_params *primme, int *ierr) {
   primme_svds_params *ps = (primme_svds_params *)primme->matrix;
   svds_side *sd = side_of(ps);
   const primme_svds_operator op = (&ps->primme == primme) ? ps->method : ps->methodStage2;
   *ierr = 1;
   if (!sd || op == primme_svds_op_none) return;
   const size_t es = es_of(sd->dt);
   if (op == primme_svds_op_augmented) {
      /* x = [v; u] (nLocal + mLocal rows): y = [A'u; A v] */
      int tr = 1, no = 0, e = 0;
      ps->matrixMatvec((char *)x + (size_t)ps->nLocal * es, ldx, y, ldy, blockSize, &tr, ps, &e);
      if (e) return;
      ps->matrixMatvec(x, ldx, (char *)y + (size_t)ps->nLocal * es, ldy, blockSize, &no, ps, &e);
      if (e) return;
      *ierr = 0;
      return;
   }
   PRIMME_INT mid = (op == primme_svds_op_AtA) ? ps->mLocal : ps->nLocal;
   int first = (op == primme_svds_op_AtA) ? 0 : 1, second = 1 - first;
   const int nb = *blockSize;
   if ((size_t)nb > sd->aux_cols) {
      if (sd->aux) hipk_free(sd->ctx, sd->aux);
      sd->aux = NULL; sd->aux_cols = 0;
      if (hipk_malloc(sd->ctx, (size_t)mid * es * nb, (void **)&sd->aux)) return;
      sd->aux_cols = nb;
   }
   int bs = nb, e = 0;
   ps->matrixMatvec(x, ldx, sd->aux, &mid, &bs, &first, ps, &e);
   if (e) return;
   ps->matrixMatvec(sd->aux, &mid, y, ldy, &bs, &second, ps, &e);
   if (e) return;
   *ierr = 0;
}

/* The one-synchronisation tail of the eigensolver's block-size-1 iteration (eigs_conv.c) for the normal equations
 * with the library's own operator on one rank: xout = a t (a = 1/sqrt(norm2_dev[0]), or t itself), u = A xout,
 * dot_dev[0] = xout'(A'A xout) = u'u, y = A'u -- everything enqueued, nothing waited for. */
int primme_amd_svds_operator_is_local(const void *op);
int pa_svds_can_fuse(const primme_params *primme) {
   if (primme->matrixMatvec != pa_svds_matvec_eigs || !primme->matrix) return 0;
   primme_svds_params *ps = (primme_svds_params *)primme->matrix;
   const primme_svds_operator op = (&ps->primme == primme) ? ps->method : ps->methodStage2;
   if (op != primme_svds_op_AtA && op != primme_svds_op_AAt) return 0;
   return ps->matrixMatvec == primme_amd_svds_matvec && ps->matrix && primme_amd_svds_operator_is_local(ps->matrix) &&
          side_of(ps) != NULL && !is_cplx(side_of(ps)->dt);
}
int pa_svds_apply_scaled(primme_params *primme, hipk_ctx *ctx, const void *t, const double *norm2_dev, void *xout, void *y,
      double *dot_dev) {
   primme_svds_params *ps = (primme_svds_params *)primme->matrix;
   svds_side *sd = side_of(ps);
   const primme_svds_operator op = (&ps->primme == primme) ? ps->method : ps->methodStage2;
   if (!sd) return PRIMME_UNEXPECTED_FAILURE;
   const size_t es = es_of(sd->dt);
   PRIMME_INT nin = (op == primme_svds_op_AtA) ? ps->nLocal : ps->mLocal, mid = (op == primme_svds_op_AtA) ? ps->mLocal : ps->nLocal;
   int first = (op == primme_svds_op_AtA) ? 0 : 1, second = 1 - first, one = 1, e = 0;
   if (sd->aux_cols < 1) {
      if (hipk_malloc(sd->ctx, (size_t)mid * es, (void **)&sd->aux)) return PRIMME_MALLOC_FAILURE;
      sd->aux_cols = 1;
   }
   CHK(hipk_copy_cols(ctx, sd->dt, nin, t, nin, xout, nin, 1));
   if (norm2_dev) CHK(hipk_scale_cols_rsqrt_dev(ctx, sd->dt, nin, xout, nin, 1, norm2_dev));
   ps->matrixMatvec(xout, &nin, sd->aux, &mid, &one, &first, ps, &e);
   if (e) return PRIMME_USER_FAILURE;
   ps->matrixMatvec(sd->aux, &mid, y, &nin, &one, &second, ps, &e);
   if (e) return PRIMME_USER_FAILURE;
   CHK(hipk_col_norms2(ctx, sd->dt, mid, sd->aux, mid, 1, dot_dev));     /* last: its second stage carries the completion flag */
   return 0;
}

/* the eigensolver's preconditioner: the user's, told which operator it is for (reference :1405-1416) */
static void pa_svds_precond_eigs(void *x, PRIMME_INT *ldx, void *y, PRIMME_INT *ldy, int *blockSize,
      primme_params *primme, int *ierr) {
   primme_svds_params *ps = (primme_svds_params *)primme->preconditioner;
   int method = (int)((&ps->primme == primme) ? ps->method : ps->methodStage2);
   ps->applyPreconditioner(x, ldx, y, ldy, blockSize, &method, ps, ierr);
}

/* |r| < max(eps, 3.16 machEps) |A|: the default test on a triplet */
void pa_svds_default_conv_test(double *sval, void *leftsvec, void *rightsvec, double *rNorm, int *method,
      int *isConv, primme_svds_params *ps, int *ierr) {
   (void)sval;
   svds_side *sd = side_of(ps);
   const double meps = (sd && is_single(sd->dt)) ? 1.1920928955078125e-07 : PA_EPS;
   *isConv = *rNorm < PA_MAX(ps->eps, meps * 3.16) * ps->aNorm;
   *ierr = 0;
   /* the augmented operator's residual norm is an estimate: confirm with the true one */
   if (*isConv && *method == (int)primme_svds_op_augmented && leftsvec && rightsvec && sd) {
      double rn = 0.0;
      if (true_res_norm(ps, sd, (char *)leftsvec, (char *)rightsvec, &rn)) { *ierr = 1; return; }
      *isConv = rn < PA_MAX(ps->eps, meps * 3.16) * ps->aNorm;
   }
}

/* the eigensolver's convergence test: translate (eval, |r_eig|) to (sigma, |r_eig| / sigma) */
void pa_larnv_uniform11(int64_t iseed[4], int64_t n, double *x);

void pa_svds_conv_test_ata(double *eval, void *evec, double *rNorm, int *isConv, primme_params *primme, int *ierr) {
   primme_svds_params *ps = (primme_svds_params *)primme->matrix;
   svds_side *sd = side_of(ps);
   const primme_svds_operator op = (&ps->primme == primme) ? ps->method : ps->methodStage2;
   const double aNorm = primme->aNorm > 0.0 ? primme->aNorm : primme->stats.estimateLargestSVal;
   const double maxaNorm = PA_MAX(primme->aNorm, primme->stats.estimateLargestSVal);
   const double meps = (sd && is_single(sd->dt)) ? 1.1920928955078125e-07 : PA_EPS;
   *ierr = 0;
   if (rNorm && *rNorm < meps * maxaNorm * 3.16) { *isConv = 1; return; }
   const double old = ps->aNorm;
   if (ps->aNorm <= 0.0) ps->aNorm = sqrt(aNorm);
   double sval = eval ? sqrt(fabs(*eval)) : 0.0;
   double srNorm = (rNorm && eval) ? *rNorm / sval : 0.0;
   int method = (int)op;
   *ierr = pa_svds_call_conv_test(ps, sval, (op == primme_svds_op_AAt) ? evec : NULL, (op == primme_svds_op_AtA) ? evec : NULL,
         srNorm, &method, isConv) ? 1 : 0;
   ps->aNorm = old;
}

/* does the installed convergence test look at the vectors? (lets the eigensolver keep its
 * fused residual path, which never forms the Ritz vector) */
int pa_svds_conv_test_is_vector_free(const primme_params *p) {
   if (p->convTestFun != pa_svds_conv_test_ata) return 0;
   const primme_svds_params *ps = (const primme_svds_params *)p->matrix;
   return ps && ps->convTestFun == pa_svds_default_conv_test;
}

/* eigensolver events forwarded to the user's svds monitor with singular values
 * (reference monitor_single_stage :1897-2020, normal-equation branch) */
static void monitor_single_stage(void *basisEvals_, int *basisSize, int *basisFlags, int *iblock, int *blockSize,
      void *basisNorms_, int *numConverged, void *lockedEvals_, int *numLocked, int *lockedFlags,
      void *lockedNorms_, int *inner_its, void *LSRes, const char *msg, double *time, primme_event *event,
      primme_params *primme, int *err) {
   primme_svds_params *ps = (primme_svds_params *)primme->matrix;
   *err = 0;
   if (!ps->monitorFun) return;
   const int nb = (basisEvals_ && basisSize) ? *basisSize : 0, nl = (lockedEvals_ && numLocked) ? *numLocked : 0;
   double *bs = (double *)malloc(sizeof(double) * (size_t)(2 * nb + 2 * nl + 1));
   if (!bs) { *err = 1; return; }
   double *bn = bs + nb, *ls = bn + nb, *ln = ls + nl;
   for (int i = 0; i < nb; i++) {
      bs[i] = sqrt(PA_MAX(0.0, ((double *)basisEvals_)[i]));
      bn[i] = basisNorms_ ? ((double *)basisNorms_)[i] / PA_MAX(bs[i], 1e-300) : 0.0;
   }
   for (int i = 0; i < nl; i++) {
      ls[i] = sqrt(PA_MAX(0.0, ((double *)lockedEvals_)[i]));
      ln[i] = lockedNorms_ ? ((double *)lockedNorms_)[i] / PA_MAX(ls[i], 1e-300) : 0.0;
   }
   int stage = 0;
   void *pbs = bs, *pbn = bn, *pls = ls, *pln = ln;
   if (ps->monitorFun_type == primme_op_float) {
      /* the user's monitor declared float operands: narrow in place, front to back */
      float *f = (float *)bs;
      for (int i = 0; i < 2 * nb + 2 * nl; i++) f[i] = (float)bs[i];
      pbs = f; pbn = f + nb; pls = f + 2 * nb; pln = f + 2 * nb + nl;
   }
   ps->monitorFun(nb ? pbs : NULL, basisSize, basisFlags, iblock, blockSize, nb ? pbn : NULL, numConverged,
         nl ? pls : NULL, numLocked, lockedFlags, nl ? pln : NULL, inner_its, LSRes, msg, time, event, &stage, ps, err);
   free(bs);
}

static int check_input(void *svals, void *svecs, void *resNorms, primme_svds_params *ps) {
   if (!ps) return -4;
   if (ps->n < 0 || ps->m < 0 || ps->nLocal < 0 || ps->mLocal < 0 || ps->nLocal > ps->n || ps->mLocal > ps->m) return -5;
   if (ps->numProcs < 1) return -6;
   if (!ps->matrixMatvec) return -7;
   if (!ps->applyPreconditioner && ps->precondition == 1) return -8;
   if (ps->numProcs > 1 && !ps->globalSumReal) return -9;
   if (ps->numSvals > PA_MIN(ps->n, ps->m)) return -10;
   if (ps->numSvals < 1) return -11;
   if (ps->target != primme_svds_smallest && ps->target != primme_svds_largest && ps->target != primme_svds_closest_abs) return -13;
   if (ps->method != primme_svds_op_AtA && ps->method != primme_svds_op_AAt && ps->method != primme_svds_op_augmented) return -14;
   if ((ps->method == primme_svds_op_augmented && ps->methodStage2 != primme_svds_op_none) ||
         (ps->method != primme_svds_op_augmented && ps->methodStage2 != primme_svds_op_augmented &&
               ps->methodStage2 != primme_svds_op_none)) return -15;
   if (ps->printLevel < 0 || ps->printLevel > 5) return -16;
   if (!svals) return -17;
   if (!svecs) return -18;
   if (!resNorms) return -19;
   return 0;
}

/* device column-block move that tolerates overlap (through a temporary) */
static int move_cols(svds_side *sd, PRIMME_INT rows, int ncols, char *src, char *dst) {
   if (ncols <= 0 || rows <= 0 || src == dst) return 0;
   const size_t bytes = (size_t)rows * ncols * es_of(sd->dt);
   char *tmp = NULL;
   CHK(hipk_malloc(sd->ctx, bytes, (void **)&tmp));
   CHK(hipk_copy_cols(sd->ctx, RDT(sd), RF(sd) * rows, src, RF(sd) * rows, tmp, RF(sd) * rows, ncols));
   CHK(hipk_copy_cols(sd->ctx, RDT(sd), RF(sd) * rows, tmp, RF(sd) * rows, dst, RF(sd) * rows, ncols));
   CHK(hipk_sync(sd->ctx));
   hipk_free(sd->ctx, tmp);
   return 0;
}

/* x(:,i) /= factors[i], or normalise when the factor is unusable (reference :1418-1438) */
static int scale_inverse(primme_svds_params *ps, svds_side *sd, char *x, PRIMME_INT rows, int ncols, const double *factors) {
   if (ncols <= 0) return 0;
   double *f = (double *)malloc(sizeof(double) * (size_t)ncols);
   double *d_n = NULL;
   if (!f) return PRIMME_MALLOC_FAILURE;
   int need_norms = 0;
   for (int i = 0; i < ncols; i++) if (!(factors[i] > 0.0 && 1.0 / factors[i] < 1.79e308)) need_norms = 1;
   double *norms = NULL;
   if (need_norms) {
      norms = (double *)malloc(sizeof(double) * (size_t)ncols);
      if (!norms || hipk_malloc(sd->ctx, sizeof(double) * (size_t)ncols, (void **)&d_n)) { free(f); free(norms); return PRIMME_MALLOC_FAILURE; }
      int rc = hipk_col_norms2(sd->ctx, RDT(sd), RF(sd) * rows, x, RF(sd) * rows, ncols, d_n);
      if (!rc) rc = hipk_d2h(sd->ctx, norms, d_n, sizeof(double) * (size_t)ncols);
      if (!rc) rc = hipk_sync(sd->ctx);
      hipk_free(sd->ctx, d_n);
      if (rc) { free(f); free(norms); return rc; }
      if (pa_svds_call_global_sum(ps, norms, ncols)) { free(f); free(norms); return PRIMME_USER_FAILURE; }
   }
   for (int i = 0; i < ncols; i++)
      f[i] = 1.0 / ((factors[i] > 0.0 && 1.0 / factors[i] < 1.79e308) ? factors[i] : sqrt(norms[i]));
   int rc = hipk_scale_cols(sd->ctx, RDT(sd), RF(sd) * rows, x, RF(sd) * rows, ncols, f);
   free(f); free(norms);
   return rc;
}

/* true residual norm of a triplet, sqrt(|A v/|v| - s u/|u||^2 + |A'u/|u| - s v/|v||^2) with
 * s = u'Av/(|u||v|)  (reference primme_svds_c.c:1512-1570): two operator applications and a few
 * reductions on the device */
static int true_res_norm(primme_svds_params *ps, svds_side *sd, char *u, char *v, double *rNorm) {
   const PRIMME_INT mL = ps->mLocal, nL = ps->nLocal;
   const size_t es = es_of(sd->dt);
   char *Atu = NULL;
   double *d_ip = NULL, ip[3];
   CHK(hipk_malloc(sd->ctx, (size_t)(mL + nL) * es, (void **)&Atu));
   if (hipk_malloc(sd->ctx, 4 * sizeof(double), (void **)&d_ip)) { hipk_free(sd->ctx, Atu); return PRIMME_MALLOC_FAILURE; }
   char *Av = Atu + (size_t)nL * es;
   int rc = svds_matvec(ps, u, mL, Atu, nL, 1, 1);
   if (!rc) rc = svds_matvec(ps, v, nL, Av, mL, 1, 0);
   const hipk_dtype rd = RDT(sd);
   const PRIMME_INT rf = RF(sd), mR = rf * mL, nR = rf * nL;           /* the vectors as real ones: Re(u'Av) is their real inner product */
   if (!rc) rc = hipk_pair_dots(sd->ctx, rd, nR, v, nR, v, nR, 1, d_ip);
   if (!rc) rc = hipk_pair_dots(sd->ctx, rd, mR, u, mR, u, mR, 1, d_ip + 1);
   if (!rc) rc = hipk_pair_dots(sd->ctx, rd, mR, u, mR, Av, mR, 1, d_ip + 2);
   if (!rc) rc = hipk_d2h(sd->ctx, ip, d_ip, 3 * sizeof(double));
   if (!rc) rc = hipk_sync(sd->ctx);
   if (!rc) rc = pa_svds_call_global_sum(ps, ip, 3);
   if (!rc) {
      ip[0] = sqrt(ip[0]); ip[1] = sqrt(ip[1]);
      const double sval = ip[2] / ip[0] / ip[1];
      if (!(sval >= 0.0) || !isfinite(sval)) *rNorm = 1.79e308;   /* negative, or a null vector of [0 A'; A 0] */
      else {
         double a;
         a = 1.0 / ip[1]; rc = hipk_scale_cols(sd->ctx, rd, nR, Atu, nR, 1, &a);
         a = -sval / ip[0]; if (!rc) rc = hipk_axpy_cols(sd->ctx, rd, nR, &a, v, nR, Atu, nR, 1);
         a = 1.0 / ip[0]; if (!rc) rc = hipk_scale_cols(sd->ctx, rd, mR, Av, mR, 1, &a);
         a = -sval / ip[1]; if (!rc) rc = hipk_axpy_cols(sd->ctx, rd, mR, &a, u, mR, Av, mR, 1);
         if (!rc) rc = hipk_col_norms2(sd->ctx, rd, mR + nR, Atu, mR + nR, 1, d_ip);
         if (!rc) rc = hipk_d2h(sd->ctx, ip, d_ip, sizeof(double));
         if (!rc) rc = hipk_sync(sd->ctx);
         if (!rc) rc = pa_svds_call_global_sum(ps, ip, 1);
         if (!rc) *rNorm = sqrt(ip[0]);
      }
   }
   hipk_free(sd->ctx, Atu); hipk_free(sd->ctx, d_ip);
   return rc;
}

/* the eigensolver's convergence test for the augmented operator (reference :1705-1745) */
void pa_svds_conv_test_aug(double *eval, void *evec, double *rNorm, int *isConv, primme_params *primme, int *ierr) {
   primme_svds_params *ps = (primme_svds_params *)primme->matrix;
   const double aNorm = primme->aNorm > 0.0 ? primme->aNorm : primme->stats.estimateLargestSVal;
   *ierr = 0;
   const double old = ps->aNorm;
   if (ps->aNorm <= 0.0) ps->aNorm = aNorm;
   double sval = eval ? fabs(*eval) : 0.0, srNorm = rNorm ? *rNorm * sqrt(2.0) : 0.0;
   int method = (int)primme_svds_op_augmented;
   svds_side *sd = side_of(ps);
   const size_t es = sd ? es_of(sd->dt) : 8;
   *ierr = pa_svds_call_conv_test(ps, sval, evec ? (char *)evec + (size_t)ps->nLocal * es : NULL, evec, srNorm, &method, isConv) ? 1 : 0;
   ps->aNorm = old;
}

typedef struct { primme_params *p; primme_svds_operator op; char *eig_vecs; int allocatedShifts, own_monitor; } svds_stage;

/* parameters and vectors of one stage from the svds block (reference copy_last_params_from_svds) */
static int stage_begin(primme_svds_params *ps, svds_side *sd, int stage, double *svals, char *svecs, double *rnorms,
      void **stream_slot, svds_stage *st) {
   primme_params *p = stage == 0 ? &ps->primme : &ps->primmeStage2;
   const primme_svds_operator op = stage == 0 ? ps->method : ps->methodStage2;
   const size_t es = es_of(sd->dt);
   const PRIMME_INT mL = ps->mLocal, nL = ps->nLocal, tot = mL + nL;
   st->p = p; st->op = op; st->eig_vecs = svecs; st->allocatedShifts = 0; st->own_monitor = 0;
   if (op == primme_svds_op_none) { p->maxMatvecs = 0; return 0; }
   const int normal = (op == primme_svds_op_AtA || op == primme_svds_op_AAt);

   if (!p->matrixMatvec) { p->matrixMatvec = pa_svds_matvec_eigs; p->matrixMatvec_type = ps->matrixMatvec_type; p->matrix = ps; }
   if (ps->applyPreconditioner && !p->applyPreconditioner) {
      p->applyPreconditioner = pa_svds_precond_eigs; p->applyPreconditioner_type = ps->applyPreconditioner_type; p->preconditioner = ps;
   }
   if (ps->aNorm > 0.0) p->aNorm = normal ? ps->aNorm * ps->aNorm : ps->aNorm;
   p->convTestFun = normal ? pa_svds_conv_test_ata : pa_svds_conv_test_aug;
   p->convTestFun_type = primme_op_double;   /* this glue takes doubles; the user's own type is honoured inside */
   p->initSize = ps->initSize;
   p->numOrthoConst = ps->numOrthoConst;
   const int n0 = ps->initSize + ps->numOrthoConst;
   const int nMax = PA_MAX(ps->initSize, ps->numSvals) + ps->numOrthoConst;
   if (normal) {
      /* [Uc U0 Vc V0]: park Vc (and V0 for A'A) at the far right; A'A iterates there */
      char *right = svecs + (size_t)nMax * mL * es;
      CHK(move_cols(sd, nL, op == primme_svds_op_AtA ? n0 : ps->numOrthoConst, svecs + (size_t)mL * n0 * es, right));
      if (op == primme_svds_op_AtA) st->eig_vecs = right;
   } else if (n0 > 0) {
      /* [Uc U Vc V] -> n0 columns [v; u] of length nLocal + mLocal; unit constraints */
      char *aux = NULL;
      CHK(hipk_malloc(sd->ctx, (size_t)tot * n0 * es, (void **)&aux));
      const hipk_dtype rd = RDT(sd);
      const PRIMME_INT rf = RF(sd);
      int rc = hipk_copy_cols(sd->ctx, rd, rf * tot * n0, svecs, rf * tot * n0, aux, rf * tot * n0, 1);
      if (!rc) rc = hipk_copy_cols(sd->ctx, rd, rf * nL, aux + (size_t)mL * n0 * es, rf * nL, svecs, rf * tot, n0);
      if (!rc) rc = hipk_copy_cols(sd->ctx, rd, rf * mL, aux, rf * mL, svecs + (size_t)nL * es, rf * tot, n0);
      if (!rc && ps->numOrthoConst > 0) {
         double f[64];
         for (int c0 = 0; c0 < ps->numOrthoConst && !rc; c0 += 64) {
            const int nc = PA_MIN(64, ps->numOrthoConst - c0);
            for (int c = 0; c < nc; c++) f[c] = 1.0 / sqrt(2.0);
            rc = hipk_scale_cols(sd->ctx, rd, rf * tot, svecs + (size_t)c0 * tot * es, rf * tot, nc, f);
         }
      }
      if (!rc) rc = hipk_sync(sd->ctx);
      hipk_free(sd->ctx, aux);
      if (rc) return rc;
   }
   for (int i = 0; i < 4; i++) p->iseed[i] = ps->iseed[i];
   p->maxMatvecs = (stage == 0) ? ps->maxMatvecs / 2 : ps->maxMatvecs / 2 - ps->primme.stats.numMatvecs;
   if ((stage == 0 && ps->numTargetShifts > 0) ||
         (stage == 1 && p->targetShifts == NULL && ps->target == primme_svds_closest_abs)) {
      p->numTargetShifts = ps->numTargetShifts;
      if (stage == 0 && normal) {
         p->targetShifts = (double *)malloc(sizeof(double) * (size_t)ps->numSvals);
         if (!p->targetShifts) return PRIMME_MALLOC_FAILURE;
         st->allocatedShifts = 1;
         for (int i = 0; i < p->numTargetShifts; i++) p->targetShifts[i] = ps->targetShifts[i] * ps->targetShifts[i];
      } else p->targetShifts = ps->targetShifts;
   }

   if (stage == 1 && p->targetShifts == NULL && ps->target == primme_svds_smallest) {
      /* closest_geq to lower bounds of the singular values found by the normal equations:
       * sqrt(max(s - |r|, 0) s), never below machEps |A| so that the |m - n| null vectors of the
       * augmented operator are not returned (reference primme_svds_c.c:700-735) */
      p->targetShifts = (double *)malloc(sizeof(double) * (size_t)ps->numSvals);
      if (!p->targetShifts) return PRIMME_MALLOC_FAILURE;
      st->allocatedShifts = 1;
      const double min_val = ps->aNorm * PA_EPS;
      int i = 0;
      for (; i < ps->initSize; i++) p->targetShifts[i] = PA_MAX(sqrt(fabs(PA_MAX(svals[i] - rnorms[i], 0.0) * svals[i])), min_val);
      for (; i < ps->numSvals; i++) p->targetShifts[i] = min_val;
      for (int a = 1; a < ps->numSvals; a++) {          /* ascending */
         const double v = p->targetShifts[a];
         int b = a - 1;
         while (b >= 0 && p->targetShifts[b] > v) { p->targetShifts[b + 1] = p->targetShifts[b]; b--; }
         p->targetShifts[b + 1] = v;
      }
      p->numTargetShifts = ps->numSvals;
   } else if (!normal && ps->target == primme_svds_smallest && p->targetShifts == NULL) {
      p->targetShifts = (double *)malloc(sizeof(double));
      if (!p->targetShifts) return PRIMME_MALLOC_FAILURE;
      st->allocatedShifts = 1;
      p->targetShifts[0] = 0.0;
      p->numTargetShifts = 1;
   }

   /* augmented operator without guesses: start from [A'x; x] or [x; A x]  (reference :760-790) */
   if (!normal && p->initSize <= 0) {
      char *v0 = svecs + (size_t)p->numOrthoConst * tot * es, *u0 = v0 + (size_t)nL * es;
      const PRIMME_INT len = (ps->m >= ps->n) ? mL : nL;
      const PRIMME_INT rlen = RF(sd) * len;         /* a complex entry takes two consecutive numbers of the stream: (re, im), as xLARNV */
      double *h = (double *)malloc(sizeof(double) * (size_t)(rlen > 0 ? rlen : 1));
      if (!h) return PRIMME_MALLOC_FAILURE;
      pa_larnv_uniform11(p->iseed, rlen, h);
      int rc = 0;
      if (is_single(sd->dt)) {
         float *hf = (float *)h;
         for (PRIMME_INT i = 0; i < rlen; i++) hf[i] = (float)h[i];
      }
      rc = hipk_h2d(sd->ctx, (ps->m >= ps->n) ? u0 : v0, h, (size_t)len * es);
      if (!rc) rc = hipk_sync(sd->ctx);
      free(h);
      if (!rc) rc = (ps->m >= ps->n) ? svds_matvec(ps, u0, mL, v0, nL, 1, 1) : svds_matvec(ps, v0, nL, u0, mL, 1, 0);
      double n2[2], *d_n = NULL;
      if (!rc) rc = hipk_malloc(sd->ctx, 2 * sizeof(double), (void **)&d_n) ? PRIMME_MALLOC_FAILURE : 0;
      if (!rc) rc = hipk_col_norms2(sd->ctx, RDT(sd), RF(sd) * nL, v0, RF(sd) * nL, 1, d_n);
      if (!rc) rc = hipk_col_norms2(sd->ctx, RDT(sd), RF(sd) * mL, u0, RF(sd) * mL, 1, d_n + 1);
      if (!rc) rc = hipk_d2h(sd->ctx, n2, d_n, 2 * sizeof(double));
      if (!rc) rc = hipk_sync(sd->ctx);
      hipk_free(sd->ctx, d_n);
      if (!rc) rc = pa_svds_call_global_sum(ps, n2, 2);
      if (rc) return rc;
      double f = 1.0 / sqrt(n2[0]);
      CHK(hipk_scale_cols(sd->ctx, RDT(sd), RF(sd) * nL, v0, RF(sd) * nL, 1, &f));
      f = 1.0 / sqrt(n2[1]);
      CHK(hipk_scale_cols(sd->ctx, RDT(sd), RF(sd) * mL, u0, RF(sd) * mL, 1, &f));
      p->initSize = 1;
      if (rnorms) rnorms[0] = 1.79e308;
      p->initBasisMode = primme_init_user;
   }

   /* second stage: triplets that already pass the criterion become constraints (reference :795-826) */
   if (stage == 1) {
      for (int i = 0; p->initSize > 0; i++) {
         int isConv = 0, method = (int)op;
         char *vi = svecs + (size_t)p->numOrthoConst * tot * es;
         if (pa_svds_call_conv_test(ps, svals[i], vi + (size_t)nL * es, vi, rnorms[i], &method, &isConv)) return PRIMME_USER_FAILURE;
         if (!isConv) break;
         p->numOrthoConst++; p->initSize--; p->numEvals--;
      }
   }
   if (ps->locking >= 0) p->locking = ps->locking;
   if (!p->monitorFun && ps->monitorFun) { p->monitorFun = monitor_single_stage; p->monitorFun_type = primme_op_double; st->own_monitor = 1; }
   p->queue = stream_slot;
   p->profile = ps->profile;
   /* the library's own communicator: let the eigensolver reduce its device partials in stream */
   if (ps->globalSumReal == primme_amd_svds_global_sum) { p->globalSumReal = primme_amd_global_sum; p->commInfo = ps->commInfo; }
   return 0;
}

/* results of one stage back into the svds block (reference copy_last_params_to_svds) */
static int stage_end(primme_svds_params *ps, svds_side *sd, int stage, double *svals, char *svecs, double *rnorms,
      svds_stage *st) {
   primme_params *p = st->p;
   const primme_svds_operator op = st->op;
   const size_t es = es_of(sd->dt);
   const PRIMME_INT mL = ps->mLocal, nL = ps->nLocal, tot = mL + nL;
   if (op == primme_svds_op_none) { p->maxMatvecs = 0; return 0; }
   const int normal = (op == primme_svds_op_AtA || op == primme_svds_op_AAt);
   if (stage == 1) {
      const int nconv = ps->numSvals - p->numEvals;
      p->initSize += nconv; p->numOrthoConst -= nconv; p->numEvals += nconv;
   }
   ps->stats.numOuterIterations += p->stats.numOuterIterations;
   ps->stats.numRestarts += p->stats.numRestarts;
   ps->stats.numMatvecs += p->stats.numMatvecs * 2;
   ps->stats.numPreconds += p->stats.numPreconds;
   ps->stats.numGlobalSum += p->stats.numGlobalSum;
   ps->stats.numBroadcast += p->stats.numBroadcast;
   ps->stats.volumeGlobalSum += p->stats.volumeGlobalSum;
   ps->stats.volumeBroadcast += p->stats.volumeBroadcast;
   ps->stats.numOrthoInnerProds += p->stats.numOrthoInnerProds;
   ps->stats.elapsedTime += p->stats.elapsedTime;
   ps->stats.timeMatvec += p->stats.timeMatvec;
   ps->stats.timePrecond += p->stats.timePrecond;
   ps->stats.timeOrtho += p->stats.timeOrtho;
   ps->stats.timeGlobalSum += p->stats.timeGlobalSum;
   ps->stats.timeBroadcast += p->stats.timeBroadcast;
   ps->stats.lockingIssue += p->stats.lockingIssue;
   if (p->aNorm > 0.0) ps->aNorm = normal ? sqrt(p->aNorm) : p->aNorm;
   if (normal) for (int i = 0; i < p->initSize; i++) svals[i] = sqrt(PA_MAX(0.0, svals[i]));

   const int nMax = PA_MAX(ps->initSize, ps->numSvals) + ps->numOrthoConst;
   ps->initSize = p->initSize;
   const int n1 = ps->initSize + ps->numOrthoConst;
   int rc = 0;
   if (op == primme_svds_op_AtA) {
      /* U = A V / Sigma, then [Vc V] moves next to it */
      char *right = svecs + (size_t)nMax * mL * es;
      char *Vfound = right + (size_t)nL * ps->numOrthoConst * es, *U = svecs + (size_t)mL * ps->numOrthoConst * es;
      rc = svds_matvec(ps, Vfound, nL, U, mL, ps->initSize, 0);
      if (!rc) rc = scale_inverse(ps, sd, U, mL, ps->initSize, svals);
      if (!rc) rc = move_cols(sd, nL, n1, right, svecs + (size_t)mL * n1 * es);
   } else if (op == primme_svds_op_AAt) {
      /* V = A' U / Sigma behind [Uc U Vc] */
      char *right = svecs + (size_t)nMax * mL * es;
      rc = move_cols(sd, nL, ps->numOrthoConst, right, svecs + (size_t)mL * n1 * es);
      char *U = svecs + (size_t)mL * ps->numOrthoConst * es;
      char *V = svecs + (size_t)mL * n1 * es + (size_t)nL * ps->numOrthoConst * es;
      if (!rc) rc = svds_matvec(ps, U, mL, V, nL, ps->initSize, 1);
      if (!rc) rc = scale_inverse(ps, sd, V, nL, ps->initSize, svals);
   } else if (n1 > 0) {
      /* constraints back to their scale, [v; u] columns -> [Uc U Vc V], unit u and v */
      if (ps->numOrthoConst > 0) {
         double f[64];
         for (int c0 = 0; c0 < ps->numOrthoConst && !rc; c0 += 64) {
            const int nc = PA_MIN(64, ps->numOrthoConst - c0);
            for (int c = 0; c < nc; c++) f[c] = sqrt(2.0);
            rc = hipk_scale_cols(sd->ctx, RDT(sd), RF(sd) * tot, svecs + (size_t)c0 * tot * es, RF(sd) * tot, nc, f);
         }
      }
      char *aux = NULL;
      const hipk_dtype rd = RDT(sd);
      const PRIMME_INT rf = RF(sd);
      if (!rc) rc = hipk_malloc(sd->ctx, (size_t)tot * n1 * es, (void **)&aux) ? PRIMME_MALLOC_FAILURE : 0;
      if (!rc) rc = hipk_copy_cols(sd->ctx, rd, rf * tot * n1, svecs, rf * tot * n1, aux, rf * tot * n1, 1);
      if (!rc) rc = hipk_copy_cols(sd->ctx, rd, rf * nL, aux, rf * tot, svecs + (size_t)mL * n1 * es, rf * nL, n1);
      if (!rc) rc = hipk_copy_cols(sd->ctx, rd, rf * mL, aux + (size_t)nL * es, rf * tot, svecs, rf * mL, n1);
      if (!rc) rc = hipk_sync(sd->ctx);
      hipk_free(sd->ctx, aux);
      double *zero = (double *)calloc((size_t)n1, sizeof(double));     /* "factor unusable" -> normalise */
      if (!zero) rc = PRIMME_MALLOC_FAILURE;
      if (!rc) rc = scale_inverse(ps, sd, svecs, mL, n1, zero);
      if (!rc) rc = scale_inverse(ps, sd, svecs + (size_t)mL * n1 * es, nL, n1, zero);
      free(zero);
   }
   if (rc) return rc;
   for (int i = 0; i < 4; i++) ps->iseed[i] = p->iseed[i];
   if (st->allocatedShifts) { free(p->targetShifts); p->targetShifts = NULL; p->numTargetShifts = 0; st->allocatedShifts = 0; }
   if (normal) for (int i = 0; i < ps->initSize; i++) rnorms[i] = PA_MIN(rnorms[i] / svals[i], ps->aNorm);
   else for (int i = 0; i < ps->initSize; i++) rnorms[i] *= sqrt(2.0);
   if (st->own_monitor) p->monitorFun = NULL;
   p->queue = NULL;
   return hipk_sync(sd->ctx);
}

static int solve_svds(void *svals_out, void *svecs_, void *resNorms_out, primme_svds_params *ps, hipk_dtype dt) {
   if (!ps) return -4;
   const int out_float = is_single(dt);
   const double meps = out_float ? 1.1920928955078125e-07 : PA_EPS;
   const primme_op_datatype scalar_t = out_float ? primme_op_float : primme_op_double;
   if (ps->matrixMatvec && ps->matrixMatvec_type == primme_op_default) ps->matrixMatvec_type = scalar_t;
   if (ps->applyPreconditioner && ps->applyPreconditioner_type == primme_op_default) ps->applyPreconditioner_type = scalar_t;
   if (ps->globalSumReal && ps->globalSumReal_type == primme_op_default) ps->globalSumReal_type = scalar_t;
   if (ps->broadcastReal && ps->broadcastReal_type == primme_op_default) ps->broadcastReal_type = scalar_t;
   if (ps->convTestFun && ps->convTestFun_type == primme_op_default) ps->convTestFun_type = scalar_t;
   if (ps->monitorFun && ps->monitorFun_type == primme_op_default) ps->monitorFun_type = scalar_t;

   if (ps->numProcs <= 1 && svals_out && svecs_ && resNorms_out) {
      ps->mLocal = ps->m; ps->nLocal = ps->n; ps->procID = 0; ps->numProcs = 1;
   }
   primme_svds_set_defaults(ps);
   if (!svals_out && !svecs_ && !resNorms_out) return 0;
   int rc = check_input(svals_out, svecs_, resNorms_out, ps);
   if (rc) { ps->initSize = 0; return rc; }

   if (!ps->convTestFun) {
      ps->convTestFun = pa_svds_default_conv_test;
      ps->convTestFun_type = primme_op_double;   /* the library's own test reads doubles */
      if (ps->eps == 0.0) ps->eps = meps * 1e4;
   }
   memset(&ps->stats, 0, sizeof(ps->stats));

   svds_side *sd = side_new(ps);
   if (!sd) { ps->initSize = 0; return PRIMME_MALLOC_FAILURE; }
   sd->dt = dt;
   if (hipk_ctx_create(&sd->ctx, ps->queue)) { sd->key = NULL; ps->initSize = 0; return PRIMME_UNEXPECTED_FAILURE; }
   void *stream = hipk_ctx_stream(sd->ctx);
   void *user_queue = ps->queue;
   if (!ps->queue) ps->queue = &stream;

   char *svecs = (char *)svecs_;
   int ret = 0;
   double *svals = (double *)calloc((size_t)ps->numSvals + 1, 8), *rnorms = (double *)calloc((size_t)ps->numSvals + 1, 8);
   svds_stage st;
   memset(&st, 0, sizeof(st));
   if (!svals || !rnorms) { rc = PRIMME_MALLOC_FAILURE; goto done; }

   /* ---- first stage ---- */
   rc = stage_begin(ps, sd, 0, NULL, svecs, rnorms, &stream, &st);
   if (rc) goto done;
   ret = is_cplx(dt) ? pa_eigs_solve_z(svals, st.eig_vecs, rnorms, st.p, dt, 1) : pa_eigs_solve(svals, st.eig_vecs, rnorms, st.p, dt, 1);
   rc = stage_end(ps, sd, 0, svals, svecs, rnorms, &st);
   if (rc) goto done;
   if (ret != 0) ret -= 100;

   /* ---- second stage (hybrid): the augmented operator refines what the normal equations left ---- */
   if (ps->methodStage2 != primme_svds_op_none && ret == 0) {
      rc = stage_begin(ps, sd, 1, svals, svecs, rnorms, &stream, &st);
      if (rc) goto done;
      const int nconv = ps->numSvals - ps->primmeStage2.numEvals;
      ret = is_cplx(dt) ? pa_eigs_solve_z(svals + nconv, st.eig_vecs, rnorms + nconv, st.p, dt, 1)
                        : pa_eigs_solve(svals + nconv, st.eig_vecs, rnorms + nconv, st.p, dt, 1);
      rc = stage_end(ps, sd, 1, svals, svecs, rnorms, &st);
      if (rc) goto done;
      if (ret != 0) ret -= 200;
   }

   for (int i = 0; i < ps->initSize; i++) {
      if (out_float) { ((float *)svals_out)[i] = (float)svals[i]; ((float *)resNorms_out)[i] = (float)rnorms[i]; }
      else { ((double *)svals_out)[i] = svals[i]; ((double *)resNorms_out)[i] = rnorms[i]; }
   }

done:
   if (st.allocatedShifts && st.p && st.p->targetShifts) { free(st.p->targetShifts); st.p->targetShifts = NULL; }
   free(svals); free(rnorms);
   ps->primme.queue = NULL; ps->primmeStage2.queue = NULL;
   ps->queue = user_queue;
   if (sd->aux) hipk_free(sd->ctx, sd->aux);
   hipk_ctx_destroy(sd->ctx);
   sd->key = NULL;
   if (rc) { ps->initSize = 0; return rc; }
   return ret;
}

int hip_dprimme_svds(double *svals, double *svecs, double *resNorms, primme_svds_params *ps) {
   return solve_svds(svals, svecs, resNorms, ps, HIPK_F64);
}
int hip_sprimme_svds(float *svals, float *svecs, float *resNorms, primme_svds_params *ps) {
   return solve_svds(svals, svecs, resNorms, ps, HIPK_F32);
}
/* the native complex front end (called by svds_complex.c's hip_zprimme_svds / hip_cprimme_svds) */
int pa_svds_solve_native_complex(void *svals, void *svecs, void *resNorms, primme_svds_params *ps, int single) {
   return solve_svds(svals, svecs, resNorms, ps, single ? HIPK_C32 : HIPK_C64);
}
