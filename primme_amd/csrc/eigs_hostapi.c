/* eigs_hostapi.c — dprimme / sprimme / zprimme / cprimme with the reference's HOST-pointer contract
 * (reference include/primme_eigs.h:386-393, src/eigs/primme_c.c:103-108): evecs is a host array,
 * matrixMatvec / applyPreconditioner / convTestFun receive host pointers with leading dimension
 * ldOPs, exactly as a program written against the CPU library expects (examples/ex_eigs_dseq.c).
 * The solve itself is the device path: these entry points stage.  A shadow primme_params goes to
 * hip_?primme with bridge callbacks installed; every operator application copies the block of
 * vectors device -> pinned host, calls the application's callback on host memory and copies the
 * result back.  That is PCIe per application — the plumbing configuration of BASELINE configs[0];
 * an application that wants the device rate hands over a device callback through hip_?primme.
 */
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include "primme_amd.h"
#include "primme_amd_kernels.h"
#include "eigs_internal.h"

typedef struct {
   primme_params *user;    /* the caller's struct: what its callbacks expect to receive */
   primme_params q;        /* the shadow handed to the device solver */
   hipk_ctx *ctx;          /* staging copies run on the solver's stream (published in q.queue) */
   size_t es;              /* bytes per vector element */
   char *hx, *hy;          /* pinned staging panels, ldh x cap columns */
   PRIMME_INT ldh;         /* the CALLER's ldOPs (>= nLocal): what its callbacks may index with (primme_c.c:333-337) */
   int cap;
} host_side;

#define HSIDE(pp) ((host_side *)((char *)(pp) - offsetof(host_side, q)))

static int stage_reserve(host_side *sd, int ncols) {
   if (ncols <= sd->cap) return 0;
   const size_t bytes = (size_t)(sd->ldh > 0 ? sd->ldh : 1) * sd->es * (size_t)ncols;
   void *a = NULL, *b = NULL;
   if (hipk_host_alloc(sd->ctx, bytes, &a) || hipk_host_alloc(sd->ctx, bytes, &b)) return -1;
   if (sd->hx) hipk_host_free(sd->ctx, sd->hx);
   if (sd->hy) hipk_host_free(sd->ctx, sd->hy);
   sd->hx = (char *)a; sd->hy = (char *)b; sd->cap = ncols;
   return 0;
}

/* a context on the stream the solver published for its callbacks */
static int side_stream(host_side *sd) {
   if (sd->ctx) return 0;
   return hipk_ctx_create(&sd->ctx, sd->q.queue);
}

static void mirror_user(host_side *sd) {
   sd->user->ShiftsForPreconditioner = sd->q.ShiftsForPreconditioner;
   sd->user->stats = sd->q.stats;
   sd->user->aNorm = sd->q.aNorm;
}

/* x (device, ld *ldx) -> host panel (ld nLocal); callback; host panel -> y (device, ld *ldy) */
static void bridge_block_op(void (*fn)(void *, PRIMME_INT *, void *, PRIMME_INT *, int *, struct primme_params *, int *),
      void *x, PRIMME_INT *ldx, void *y, PRIMME_INT *ldy, int *blockSize, primme_params *qp, int *ierr) {
   host_side *sd = HSIDE(qp);
   const int nb = *blockSize;
   const PRIMME_INT m = qp->nLocal;
   *ierr = 1;
   if (nb <= 0) { *ierr = 0; return; }
   if (side_stream(sd) || stage_reserve(sd, nb)) return;
   const size_t colB = (size_t)m * sd->es, colH = (size_t)sd->ldh * sd->es;
   for (int c = 0; c < nb; c++)
      if (hipk_d2h(sd->ctx, sd->hx + colH * c, (char *)x + (size_t)c * (size_t)*ldx * sd->es, colB)) return;
   if (hipk_sync(sd->ctx)) return;
   mirror_user(sd);
   PRIMME_INT ldh = sd->ldh;
   int e = 0;
   fn(sd->hx, &ldh, sd->hy, &ldh, blockSize, sd->user, &e);
   if (e) return;
   for (int c = 0; c < nb; c++)
      if (hipk_h2d(sd->ctx, (char *)y + (size_t)c * (size_t)*ldy * sd->es, sd->hy + colH * c, colB)) return;
   if (hipk_sync(sd->ctx)) return;      /* hy is reused by the next application */
   *ierr = 0;
}

static void bridge_matvec(void *x, PRIMME_INT *ldx, void *y, PRIMME_INT *ldy, int *bs, primme_params *qp, int *ierr) {
   bridge_block_op(HSIDE(qp)->user->matrixMatvec, x, ldx, y, ldy, bs, qp, ierr);
}
static void bridge_mass(void *x, PRIMME_INT *ldx, void *y, PRIMME_INT *ldy, int *bs, primme_params *qp, int *ierr) {
   bridge_block_op(HSIDE(qp)->user->massMatrixMatvec, x, ldx, y, ldy, bs, qp, ierr);      /* generalised problems (round 6) */
}
static void bridge_precond(void *x, PRIMME_INT *ldx, void *y, PRIMME_INT *ldy, int *bs, primme_params *qp, int *ierr) {
   bridge_block_op(HSIDE(qp)->user->applyPreconditioner, x, ldx, y, ldy, bs, qp, ierr);
}
/* the application's convergence test may look at the vector: hand it a host copy */
static void bridge_conv_test(double *eval, void *evec, double *rNorm, int *isconv, primme_params *qp, int *ierr) {
   host_side *sd = HSIDE(qp);
   void *hv = NULL;
   *ierr = 1;
   if (evec) {
      if (side_stream(sd) || stage_reserve(sd, 1)) return;
      if (hipk_d2h(sd->ctx, sd->hx, evec, (size_t)qp->nLocal * sd->es) || hipk_sync(sd->ctx)) return;
      hv = sd->hx;
   }
   mirror_user(sd);
   /* eval / rNorm arrive in the shadow's convTestFun_type, which is the user's (copied) */
   sd->user->convTestFun(eval, hv, rNorm, isconv, sd->user, ierr);
}
static void bridge_monitor(void *basisEvals, int *basisSize, int *basisFlags, int *iblock, int *blockSize,
      void *basisNorms, int *numConverged, void *lockedEvals, int *numLocked, int *lockedFlags, void *lockedNorms,
      int *inner_its, void *LSRes, const char *msg, double *time, primme_event *event, primme_params *qp, int *ierr) {
   host_side *sd = HSIDE(qp);
   mirror_user(sd);
   sd->user->monitorFun(basisEvals, basisSize, basisFlags, iblock, blockSize, basisNorms, numConverged, lockedEvals,
         numLocked, lockedFlags, lockedNorms, inner_its, LSRes, msg, time, event, sd->user, ierr);
}
static void bridge_global_sum(void *s, void *r, int *count, primme_params *qp, int *ierr) {
   host_side *sd = HSIDE(qp);
   sd->user->globalSumReal(s, r, count, sd->user, ierr);     /* host buffers already: the reference's contract */
}
static void bridge_broadcast(void *b, int *count, primme_params *qp, int *ierr) {
   host_side *sd = HSIDE(qp);
   sd->user->broadcastReal(b, count, sd->user, ierr);
}

typedef int (*dev_solver)(void *, void *, void *, primme_params *);
static int call_d(void *a, void *b, void *c, primme_params *p) { return hip_dprimme((double *)a, (double *)b, (double *)c, p); }
static int call_s(void *a, void *b, void *c, primme_params *p) { return hip_sprimme((float *)a, (float *)b, (float *)c, p); }
static int call_z(void *a, void *b, void *c, primme_params *p) { return hip_zprimme((double *)a, b, (double *)c, p); }
static int call_c(void *a, void *b, void *c, primme_params *p) { return hip_cprimme((float *)a, b, (float *)c, p); }

static int solve_host(void *evals, void *evecs, void *resNorms, primme_params *primme, dev_solver solver, size_t es) {
   if (!primme) return -4;
   if (!evals && !evecs && !resNorms) return solver(NULL, NULL, NULL, primme);   /* defaults query (primme_c.c:301-306) */
   if (!evals) return -5;        /* argument checks the device entry cannot make on a host pointer */
   if (!evecs) return -31;
   if (!resNorms) return -6;
   if (primme->queue) return -32; /* a device queue belongs to the device entry points (hip_?primme) */

   host_side *sd = (host_side *)calloc(1, sizeof(*sd));
   if (!sd) return PRIMME_MALLOC_FAILURE;
   sd->user = primme; sd->es = es;
   /* the defaults the solver would fill in, so that sizes (nLocal, ldevecs, numEvals ...) are final */
   if (primme->numProcs <= 1) { primme->nLocal = primme->n; primme->procID = 0; }
   primme_set_defaults(primme);
   if (primme->ldOPs == -1 || primme->ldOPs == 0) primme->ldOPs = primme->nLocal;
   sd->ldh = primme->ldOPs >= primme->nLocal ? primme->ldOPs : primme->nLocal;
   sd->q = *primme;
   primme_params *q = &sd->q;
   const PRIMME_INT m = primme->nLocal, ldu = primme->ldevecs >= m ? primme->ldevecs : m;
   const int ncols = primme->numOrthoConst + (primme->numEvals > primme->initSize ? primme->numEvals : primme->initSize);
   q->ldevecs = m; q->ldOPs = m;
   if (primme->matrixMatvec) q->matrixMatvec = bridge_matvec;
   if (primme->applyPreconditioner) q->applyPreconditioner = bridge_precond;
   if (primme->massMatrixMatvec) q->massMatrixMatvec = bridge_mass;
   if (primme->convTestFun) q->convTestFun = bridge_conv_test;
   if (primme->monitorFun) q->monitorFun = bridge_monitor;
   if (primme->globalSumReal) q->globalSumReal = bridge_global_sum;
   if (primme->broadcastReal) q->broadcastReal = bridge_broadcast;

   int ret;
   hipk_ctx *ctx = NULL;
   char *devecs = NULL;
   if (hipk_ctx_create(&ctx, NULL)) { free(sd); return PRIMME_UNEXPECTED_FAILURE; }
   const size_t colB = (size_t)(m > 0 ? m : 0) * es;
   if (hipk_malloc(ctx, colB * (size_t)(ncols > 0 ? ncols : 1) + 16, (void **)&devecs)) { ret = PRIMME_MALLOC_FAILURE; goto done; }
   /* constraints and initial guesses travel to the device (primme_c.c:384-389) */
   {
      const int nin = primme->numOrthoConst + primme->initSize;
      for (int c = 0; c < nin && colB; c++)
         if (hipk_h2d(ctx, devecs + colB * c, (char *)evecs + (size_t)c * (size_t)ldu * es, colB)) { ret = PRIMME_UNEXPECTED_FAILURE; goto done; }
      if (hipk_sync(ctx)) { ret = PRIMME_UNEXPECTED_FAILURE; goto done; }
   }
   ret = solver(evals, devecs, resNorms, q);
   /* what the solver REPORTS goes back into the caller's struct, field by field: the struct itself stays the
    * caller's (its callbacks may have changed fields of it during the solve, e.g. from a monitor) */
   primme->stats = q->stats;
   primme->initSize = q->initSize;
   primme->aNorm = q->aNorm; primme->BNorm = q->BNorm; primme->invBNorm = q->invBNorm;
   primme->eps = q->eps;            /* the default the solver fills in when eps was 0 (primme_c.c leaves it in the caller's struct) */
   primme->dynamicMethodSwitch = q->dynamicMethodSwitch;
   primme->correctionParams.maxInnerIterations = q->correctionParams.maxInnerIterations;   /* the dynamic method's choice */
   for (int i = 0; i < 4; i++) primme->iseed[i] = q->iseed[i];
   primme->ShiftsForPreconditioner = q->ShiftsForPreconditioner;
   {
      const int nout = primme->numOrthoConst + (primme->initSize > 0 ? primme->initSize : 0);
      const int nback = nout < ncols ? (ret == 0 || ret == PRIMME_MAIN_ITER_FAILURE ? ncols : nout) : ncols;
      for (int c = primme->numOrthoConst; c < nback && colB; c++)
         if (hipk_d2h(ctx, (char *)evecs + (size_t)c * (size_t)ldu * es, devecs + colB * c, colB)) { ret = ret ? ret : PRIMME_UNEXPECTED_FAILURE; break; }
      if (hipk_sync(ctx) && !ret) ret = PRIMME_UNEXPECTED_FAILURE;
   }
done:
   if (devecs) hipk_free(ctx, devecs);
   if (sd->ctx) {
      if (sd->hx) hipk_host_free(sd->ctx, sd->hx);
      if (sd->hy) hipk_host_free(sd->ctx, sd->hy);
      hipk_ctx_destroy(sd->ctx);
   }
   hipk_ctx_destroy(ctx);
   free(sd);
   return ret;
}

int dprimme(double *evals, double *evecs, double *resNorms, primme_params *primme) {
   return solve_host(evals, evecs, resNorms, primme, call_d, 8);
}
int sprimme(float *evals, float *evecs, float *resNorms, primme_params *primme) {
   return solve_host(evals, evecs, resNorms, primme, call_s, 4);
}
int zprimme(double *evals, void *evecs, double *resNorms, primme_params *primme) {
   return solve_host(evals, evecs, resNorms, primme, call_z, 16);
}
int cprimme(float *evals, void *evecs, float *resNorms, primme_params *primme) {
   return solve_host(evals, evecs, resNorms, primme, call_c, 8);
}
