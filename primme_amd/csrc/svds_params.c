/* svds_params.c — defaults and method presets of primme_svds_params.
 *   primme_svds_initialize   <- reference src/svds/primme_svds_interface.c:108-182
 *   primme_svds_set_method   <- :218-255
 *   primme_svds_set_defaults <- :267-283, stage parameter transfer :296-410
 * The values written are ABI: tests/test_svds_host.py compares the whole block byte for byte
 * with what the reference writes for the same inputs. */
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include "primme_amd_svds.h"

int pa_svds_call_global_sum(primme_svds_params *ps, double *buf, int count);

static void sum_via_svds(void *sendBuf, void *recvBuf, int *count, primme_params *primme, int *ierr) {
   /* installed with globalSumReal_type = double (the eigensolver reduces doubles); the user's own
    * declared type is honoured here */
   primme_svds_params *ps = (primme_svds_params *)primme->matrix;
   if (sendBuf != recvBuf) for (int i = 0; i < *count; i++) ((double *)recvBuf)[i] = ((const double *)sendBuf)[i];
   *ierr = pa_svds_call_global_sum(ps, (double *)recvBuf, *count) ? 1 : 0;
}
static void bcast_via_svds(void *buffer, int *count, primme_params *primme, int *ierr) {
   primme_svds_params *ps = (primme_svds_params *)primme->matrix;
   ps->broadcastReal(buffer, count, ps, ierr);
}

primme_svds_params *primme_svds_params_create(void) {
   primme_svds_params *ps = (primme_svds_params *)malloc(sizeof(*ps));
   if (ps) primme_svds_initialize(ps);
   return ps;
}
int primme_svds_params_destroy(primme_svds_params *ps) { free(ps); return 0; }
void primme_svds_free(primme_svds_params *ps) { (void)ps; }

void primme_svds_initialize(primme_svds_params *ps) {
   ps->m = ps->n = 0;
   ps->numSvals = 1;
   ps->target = primme_svds_largest;
   ps->method = ps->methodStage2 = primme_svds_op_none;
   ps->numTargetShifts = 0;
   ps->targetShifts = NULL;
   ps->numProcs = 1;
   ps->procID = 0;
   ps->mLocal = ps->nLocal = -1;
   ps->commInfo = NULL;
   ps->globalSumReal = NULL;  ps->globalSumReal_type = primme_op_default;
   ps->broadcastReal = NULL;  ps->broadcastReal_type = primme_op_default;
   ps->internalPrecision = primme_op_default;
   ps->matrix = ps->preconditioner = NULL;
   ps->matrixMatvec = NULL;         ps->matrixMatvec_type = primme_op_default;
   ps->applyPreconditioner = NULL;  ps->applyPreconditioner_type = primme_op_default;
   ps->aNorm = 0.0;
   ps->eps = 0.0;
   ps->precondition = -1;
   ps->initSize = ps->maxBasisSize = ps->maxBlockSize = 0;
   ps->maxMatvecs = INT_MAX;
   ps->printLevel = 1;
   ps->outputFile = stdout;
   ps->locking = -1;
   ps->numOrthoConst = 0;
   ps->stats.numOuterIterations = ps->stats.numRestarts = ps->stats.numMatvecs = ps->stats.numPreconds = 0;
   ps->stats.numGlobalSum = ps->stats.volumeGlobalSum = ps->stats.numBroadcast = ps->stats.volumeBroadcast = 0;
   ps->stats.numOrthoInnerProds = ps->stats.elapsedTime = ps->stats.timeMatvec = ps->stats.timePrecond = 0.0;
   ps->stats.timeOrtho = ps->stats.timeGlobalSum = ps->stats.timeBroadcast = 0.0;
   for (int i = 0; i < 4; i++) ps->iseed[i] = -1;
   ps->convTestFun = NULL;  ps->convTestFun_type = primme_op_default;  ps->convtest = NULL;
   ps->monitorFun = NULL;   ps->monitorFun_type = primme_op_default;   ps->monitor = NULL;
   ps->queue = NULL;
   ps->profile = NULL;
   primme_initialize(&ps->primme);
   primme_initialize(&ps->primmeStage2);
}

/* what the eigensolver of one stage inherits from the svds block */
static void stage_from_svds(primme_svds_params *ps, int stage) {
   primme_params *p = stage == 0 ? &ps->primme : &ps->primmeStage2;
   const primme_svds_operator op = stage == 0 ? ps->method : ps->methodStage2;
   if (op == primme_svds_op_none) { p->maxMatvecs = 1; return; }
   const int normal = (op == primme_svds_op_AtA || op == primme_svds_op_AAt);

   p->numEvals = ps->numSvals;
   if (ps->aNorm > 0.0) p->aNorm = normal ? ps->aNorm * ps->aNorm : ps->aNorm * sqrt(2.0);
   p->eps = ps->eps;
   p->initSize = ps->initSize;
   if (ps->maxBasisSize > 0) p->maxBasisSize = ps->maxBasisSize;
   if (ps->maxBlockSize > 0) p->maxBlockSize = ps->maxBlockSize;
   p->maxMatvecs = ps->maxMatvecs;
   p->printLevel = ps->printLevel;
   p->outputFile = ps->outputFile;
   p->numOrthoConst = ps->numOrthoConst;
   if (ps->numProcs > 1) { p->procID = ps->procID; p->numProcs = ps->numProcs; p->commInfo = ps->commInfo; }
   if (ps->globalSumReal) { p->globalSumReal = sum_via_svds; p->globalSumReal_type = primme_op_double; }
   if (ps->broadcastReal) p->broadcastReal = bcast_via_svds;

   if (op == primme_svds_op_AtA) {
      p->n = ps->n;
      if (p->nLocal == -1 && ps->nLocal != -1) p->nLocal = ps->nLocal;
   } else if (op == primme_svds_op_AAt) {
      p->n = ps->m;
      if (p->nLocal == -1 && ps->mLocal != -1) p->nLocal = ps->mLocal;
   } else {
      p->n = ps->m + ps->n;
      if (p->nLocal == -1 && ps->mLocal != -1 && ps->nLocal != -1) p->nLocal = ps->mLocal + ps->nLocal;
   }

   if (ps->target == primme_svds_largest) p->target = primme_largest;
   else if (ps->target == primme_svds_smallest) p->target = normal ? primme_smallest : primme_closest_geq;
   else { p->target = primme_closest_abs; p->numTargetShifts = ps->numTargetShifts; }

   if (stage == 1 && p->initBasisMode == primme_init_default) p->initBasisMode = primme_init_user;
   if (((!normal && ps->target != primme_svds_largest) || ps->target == primme_svds_closest_abs) &&
         p->projectionParams.projection == primme_proj_default)
      p->projectionParams.projection = primme_proj_refined;
   if (ps->locking >= 0) p->locking = ps->locking;
   if (ps->precondition >= 0) p->correctionParams.precondition = ps->precondition;
   else if (p->correctionParams.precondition < 0) p->correctionParams.precondition = ps->applyPreconditioner ? 1 : 0;
}

void primme_svds_set_defaults(primme_svds_params *ps) {
   if (ps->method == primme_svds_op_none)
      primme_svds_set_method(primme_svds_default, PRIMME_DEFAULT_METHOD, PRIMME_DEFAULT_METHOD, ps);
   stage_from_svds(ps, 0);
   if (ps->methodStage2 != primme_svds_op_none) stage_from_svds(ps, 1);
}

int primme_svds_set_method(primme_svds_preset_method method, primme_preset_method methodStage1,
      primme_preset_method methodStage2, primme_svds_params *ps) {
   const primme_svds_operator normal = (ps->n <= ps->m) ? primme_svds_op_AtA : primme_svds_op_AAt;
   switch (method) {
   case primme_svds_default:
   case primme_svds_hybrid: ps->method = normal; ps->methodStage2 = primme_svds_op_augmented; break;
   case primme_svds_normalequations: ps->method = normal; ps->methodStage2 = primme_svds_op_none; break;
   case primme_svds_augmented: ps->method = primme_svds_op_augmented; ps->methodStage2 = primme_svds_op_none; break;
   }
   primme_svds_set_defaults(ps);
   primme_set_method(methodStage1, &ps->primme);
   if (methodStage2 == PRIMME_DEFAULT_METHOD && ps->target != primme_svds_largest) methodStage2 = PRIMME_JDQMR;
   if (ps->methodStage2 != primme_svds_op_none) primme_set_method(methodStage2, &ps->primmeStage2);
   return 0;
}
