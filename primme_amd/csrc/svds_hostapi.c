/* svds_hostapi.c — dprimme_svds / sprimme_svds / zprimme_svds / cprimme_svds with the reference's
 * HOST-pointer contract (reference include/primme_svds.h:236-243, src/svds/primme_svds_c.c:113-118):
 * svecs is a host array [Uc | U | Vc | V] (left vectors with leading dimension mLocal, right vectors with
 * nLocal), matrixMatvec / applyPreconditioner / convTestFun receive host pointers.  BASELINE configs[4] is
 * worded with this entry point.  Like eigs_hostapi.c these entry points only stage: a shadow
 * primme_svds_params with bridge callbacks goes to hip_?primme_svds; every operator application copies
 * the block device -> pinned host, calls the application's callback on host memory and copies the result
 * back (PCIe per application: the plumbing configuration; an application that wants the device rate hands
 * a device callback, or the library's CSR operator, to hip_?primme_svds).
 */
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include "primme_amd.h"
#include "primme_amd_svds.h"
#include "primme_amd_kernels.h"
#include "eigs_internal.h"

typedef struct {
   primme_svds_params *user;   /* the caller's struct: what its callbacks expect to receive */
   primme_svds_params q;       /* the shadow handed to the device solver */
   hipk_ctx *ctx;              /* staging copies run on the solver's stream (published in q.queue) */
   size_t es;                  /* bytes per vector element */
   char *hx, *hy;              /* pinned staging panels */
   size_t capx, capy;          /* in bytes */
} svds_host_side;

#define SHSIDE(pp) ((svds_host_side *)((char *)(pp) - offsetof(svds_host_side, q)))

static int sh_reserve(svds_host_side *sd, char **buf, size_t *cap, size_t bytes) {
   if (bytes <= *cap) return 0;
   void *a = NULL;
   if (hipk_host_alloc(sd->ctx, bytes, &a)) return -1;
   if (*buf) hipk_host_free(sd->ctx, *buf);
   *buf = (char *)a; *cap = bytes;
   return 0;
}
static int sh_stream(svds_host_side *sd) {
   if (sd->ctx) return 0;
   return hipk_ctx_create(&sd->ctx, sd->q.queue);
}
/* the caller's struct as its callbacks may read it during the solve (the reference hands them the live struct) */
static void sh_mirror(svds_host_side *sd) {
   sd->user->stats = sd->q.stats;
   sd->user->aNorm = sd->q.aNorm;
   sd->user->initSize = sd->q.initSize;
   sd->user->primme.stats = sd->q.primme.stats;
   sd->user->primmeStage2.stats = sd->q.primmeStage2.stats;
   for (int i = 0; i < 4; i++) sd->user->iseed[i] = sd->q.iseed[i];
}

/* rows of the input / output block of an operator application (primme_svds.h:140-157; mode of the
 * preconditioner: primme_svds_op_AtA -> nLocal, _AAt -> mLocal, _augmented -> mLocal + nLocal) */
static void sh_block_op(primme_svds_block_op fn, PRIMME_INT rows_in, PRIMME_INT rows_out, void *x, PRIMME_INT *ldx, void *y,
      PRIMME_INT *ldy, int *blockSize, int *mode, primme_svds_params *qp, int *ierr) {
   svds_host_side *sd = SHSIDE(qp);
   const int nb = *blockSize;
   *ierr = 1;
   if (nb <= 0) { *ierr = 0; return; }
   const size_t cin = (size_t)rows_in * sd->es, cout = (size_t)rows_out * sd->es;
   if (sh_stream(sd) || sh_reserve(sd, &sd->hx, &sd->capx, (cin ? cin : 8) * nb) || sh_reserve(sd, &sd->hy, &sd->capy, (cout ? cout : 8) * nb)) return;
   for (int c = 0; c < nb && cin; c++)
      if (hipk_d2h(sd->ctx, sd->hx + cin * c, (char *)x + (size_t)c * (size_t)*ldx * sd->es, cin)) return;
   if (hipk_sync(sd->ctx)) return;
   sh_mirror(sd);
   PRIMME_INT lin = rows_in, lout = rows_out;
   int e = 0;
   fn(sd->hx, &lin, sd->hy, &lout, blockSize, mode, sd->user, &e);
   if (e) return;
   for (int c = 0; c < nb && cout; c++)
      if (hipk_h2d(sd->ctx, (char *)y + (size_t)c * (size_t)*ldy * sd->es, sd->hy + cout * c, cout)) return;
   if (hipk_sync(sd->ctx)) return;      /* hy is reused by the next application */
   *ierr = 0;
}
static void sh_matvec(void *x, PRIMME_INT *ldx, void *y, PRIMME_INT *ldy, int *bs, int *transpose, primme_svds_params *qp, int *ierr) {
   const PRIMME_INT mL = qp->mLocal, nL = qp->nLocal;
   sh_block_op(SHSIDE(qp)->user->matrixMatvec, *transpose ? mL : nL, *transpose ? nL : mL, x, ldx, y, ldy, bs, transpose, qp, ierr);
}
static void sh_precond(void *x, PRIMME_INT *ldx, void *y, PRIMME_INT *ldy, int *bs, int *mode, primme_svds_params *qp, int *ierr) {
   const PRIMME_INT rows = (*mode == primme_svds_op_AtA) ? qp->nLocal : (*mode == primme_svds_op_AAt) ? qp->mLocal : qp->mLocal + qp->nLocal;
   sh_block_op(SHSIDE(qp)->user->applyPreconditioner, rows, rows, x, ldx, y, ldy, bs, mode, qp, ierr);
}
/* the application's convergence test may look at the vectors: host copies */
static void sh_conv_test(double *sval, void *lsvec, void *rsvec, double *rNorm, int *method, int *isconv, primme_svds_params *qp, int *ierr) {
   svds_host_side *sd = SHSIDE(qp);
   void *hl = NULL, *hr = NULL;
   *ierr = 1;
   if (lsvec || rsvec) {
      const size_t bl = (size_t)qp->mLocal * sd->es, br = (size_t)qp->nLocal * sd->es;
      if (sh_stream(sd) || sh_reserve(sd, &sd->hx, &sd->capx, bl + 8) || sh_reserve(sd, &sd->hy, &sd->capy, br + 8)) return;
      if (lsvec && bl && hipk_d2h(sd->ctx, sd->hx, lsvec, bl)) return;
      if (rsvec && br && hipk_d2h(sd->ctx, sd->hy, rsvec, br)) return;
      if (hipk_sync(sd->ctx)) return;
      hl = lsvec ? sd->hx : NULL; hr = rsvec ? sd->hy : NULL;
   }
   sh_mirror(sd);
   sd->user->convTestFun(sval, hl, hr, rNorm, method, isconv, sd->user, ierr);
}
static void sh_monitor(void *basisSvals, int *basisSize, int *basisFlags, int *iblock, int *blockSize, void *basisNorms,
      int *numConverged, void *lockedSvals, int *numLocked, int *lockedFlags, void *lockedNorms, int *inner_its, void *LSRes,
      const char *msg, double *time, primme_event *event, int *stage, primme_svds_params *qp, int *ierr) {
   svds_host_side *sd = SHSIDE(qp);
   sh_mirror(sd);
   sd->user->monitorFun(basisSvals, basisSize, basisFlags, iblock, blockSize, basisNorms, numConverged, lockedSvals, numLocked,
         lockedFlags, lockedNorms, inner_its, LSRes, msg, time, event, stage, sd->user, ierr);
}
static void sh_global_sum(void *s, void *r, int *count, primme_svds_params *qp, int *ierr) {
   SHSIDE(qp)->user->globalSumReal(s, r, count, SHSIDE(qp)->user, ierr);     /* host buffers already */
}
static void sh_broadcast(void *b, int *count, primme_svds_params *qp, int *ierr) {
   SHSIDE(qp)->user->broadcastReal(b, count, SHSIDE(qp)->user, ierr);
}

typedef int (*svds_dev_solver)(void *, void *, void *, primme_svds_params *);
static int scall_d(void *a, void *b, void *c, primme_svds_params *p) { return hip_dprimme_svds((double *)a, (double *)b, (double *)c, p); }
static int scall_s(void *a, void *b, void *c, primme_svds_params *p) { return hip_sprimme_svds((float *)a, (float *)b, (float *)c, p); }
static int scall_z(void *a, void *b, void *c, primme_svds_params *p) { return hip_zprimme_svds((double *)a, b, (double *)c, p); }
static int scall_c(void *a, void *b, void *c, primme_svds_params *p) { return hip_cprimme_svds((float *)a, b, (float *)c, p); }

static int svds_solve_host(void *svals, void *svecs, void *resNorms, primme_svds_params *ps, svds_dev_solver solver, size_t es) {
   if (!ps) return -4;
   if (!svals && !svecs && !resNorms) return solver(NULL, NULL, NULL, ps);     /* defaults query (primme_svds_c.c:205-209) */
   if (!svals) return -17;       /* the reference's codes for these three (primme_svds_c.c:1085-1090) */
   if (!svecs) return -18;
   if (!resNorms) return -19;
   if (ps->queue) return -21;    /* a device queue belongs to the device entry points (hip_?primme_svds) */

   svds_host_side *sd = (svds_host_side *)calloc(1, sizeof(*sd));
   if (!sd) return PRIMME_MALLOC_FAILURE;
   sd->user = ps; sd->es = es;
   /* the defaults the solver would fill in, so that the sizes are final */
   if (ps->numProcs <= 1) { ps->mLocal = ps->m; ps->nLocal = ps->n; ps->procID = 0; ps->numProcs = 1; }
   primme_svds_set_defaults(ps);
   sd->q = *ps;
   primme_svds_params *q = &sd->q;
   if (ps->matrixMatvec) q->matrixMatvec = sh_matvec;
   if (ps->applyPreconditioner) q->applyPreconditioner = sh_precond;
   if (ps->convTestFun) q->convTestFun = sh_conv_test;
   if (ps->monitorFun) q->monitorFun = sh_monitor;
   if (ps->globalSumReal) q->globalSumReal = sh_global_sum;
   if (ps->broadcastReal) q->broadcastReal = sh_broadcast;

   int ret;
   hipk_ctx *ctx = NULL;
   char *dsvecs = NULL;
   if (hipk_ctx_create(&ctx, NULL)) { free(sd); return PRIMME_UNEXPECTED_FAILURE; }
   const size_t tot = (size_t)(ps->mLocal + ps->nLocal);
   const int nin = ps->numOrthoConst + (ps->initSize > 0 ? ps->initSize : 0);
   const int nmax = ps->numOrthoConst + (ps->numSvals > ps->initSize ? ps->numSvals : ps->initSize);
   if (hipk_malloc(ctx, tot * es * (size_t)(nmax > 0 ? nmax : 1) + 16, (void **)&dsvecs)) { ret = PRIMME_MALLOC_FAILURE; goto done; }
   /* constraints and initial guesses travel to the device as they lie: [Uc U0 | Vc V0], packed (primme_svds.h:214-222) */
   if (nin > 0 && tot > 0) {
      if (hipk_h2d(ctx, dsvecs, svecs, tot * es * (size_t)nin) || hipk_sync(ctx)) { ret = PRIMME_UNEXPECTED_FAILURE; goto done; }
   }
   ret = solver(svals, dsvecs, resNorms, q);
   /* what the solver REPORTS goes back into the caller's struct, field by field */
   ps->stats = q->stats;
   ps->initSize = q->initSize;
   ps->aNorm = q->aNorm;
   ps->eps = q->eps;                /* the default filled in when eps was 0 */
   for (int i = 0; i < 4; i++) ps->iseed[i] = q->iseed[i];
   ps->primme.stats = q->primme.stats; ps->primmeStage2.stats = q->primmeStage2.stats;
   {
      const int nout = ps->numOrthoConst + (ps->initSize > 0 ? ps->initSize : 0);
      if (nout > 0 && tot > 0)
         if (hipk_d2h(ctx, svecs, dsvecs, tot * es * (size_t)nout) || hipk_sync(ctx)) ret = ret ? ret : PRIMME_UNEXPECTED_FAILURE;
   }
done:
   if (dsvecs) hipk_free(ctx, dsvecs);
   if (sd->ctx) {
      if (sd->hx) hipk_host_free(sd->ctx, sd->hx);
      if (sd->hy) hipk_host_free(sd->ctx, sd->hy);
      hipk_ctx_destroy(sd->ctx);
   }
   hipk_ctx_destroy(ctx);
   free(sd);
   return ret;
}

int dprimme_svds(double *svals, double *svecs, double *resNorms, primme_svds_params *ps) {
   return svds_solve_host(svals, svecs, resNorms, ps, scall_d, 8);
}
int sprimme_svds(float *svals, float *svecs, float *resNorms, primme_svds_params *ps) {
   return svds_solve_host(svals, svecs, resNorms, ps, scall_s, 4);
}
int zprimme_svds(double *svals, void *svecs, double *resNorms, primme_svds_params *ps) {
   return svds_solve_host(svals, svecs, resNorms, ps, scall_z, 16);
}
int cprimme_svds(float *svals, void *svecs, float *resNorms, primme_svds_params *ps) {
   return svds_solve_host(svals, svecs, resNorms, ps, scall_c, 8);
}
