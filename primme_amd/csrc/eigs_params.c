/* eigs_params.c — parameter block: sentinels, method presets, derived defaults and
 * argument checks.  Behaviour restated from reference src/eigs/primme_interface.c:101-217
 * (primme_initialize), :293-531 (primme_set_method), :543-617 (primme_set_defaults)
 * and src/eigs/primme_c.c:438-535 (check_input): the values and the order in which
 * defaults depend on each other decide basis/restart sizes and locking, hence
 * iteration paths, so they are kept identical; the code is organised as a preset
 * table instead of the reference's if-chain.
 */
#include "eigs_internal.h"
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

void primme_initialize(primme_params *p) {
   memset(p, 0, sizeof(*p));   /* every pointer NULL, every enum *_default, every counter 0 */
   p->numEvals = 1;
   p->target = primme_smallest;
   p->numProcs = 1;
   p->nLocal = -1;
   p->locking = -1;
   p->dynamicMethodSwitch = -1;
   p->maxMatvecs = INT_MAX;
   p->maxOuterIterations = INT_MAX;
   p->restartingParams.maxPrevRetain = -1;
   p->correctionParams.precondition = -1;
   p->correctionParams.maxInnerIterations = -INT_MAX;
   p->correctionParams.convTest = primme_adaptive_ETolerance;
   p->outputFile = stdout;
   p->printLevel = 1;
   p->stats.estimateMinEVal = -HUGE_VAL;
   p->stats.estimateMaxEVal = HUGE_VAL;
   p->stats.estimateLargestSVal = -HUGE_VAL;
   p->stats.estimateBNorm = -HUGE_VAL;
   p->stats.estimateInvBNorm = -HUGE_VAL;
   for (int i = 0; i < 4; i++) p->iseed[i] = -1;
   p->ldevecs = -1;
   p->ldOPs = -1;
}

void primme_free(primme_params *p) { (void)p; }

primme_params *primme_params_create(void) {
   primme_params *p = (primme_params *)malloc(sizeof(primme_params));
   if (p) primme_initialize(p);
   return p;
}
int primme_params_destroy(primme_params *p) { free(p); return 0; }

/* ---- presets --------------------------------------------------------------------
 * KEEP = leave the user's value.  "prev" encodes the +k rule:
 *   PREV_PLUSK : if maxPrevRetain <= 0 -> 2 when (block size 1 and numEvals > 1) or a
 *                mass matrix is present, else maxBlockSize
 *   PREV_ONE_IF_UNSET : if maxPrevRetain < 0 -> 1
 */
#define KEEP (-99)
enum { PREV_KEEP = -99, PREV_PLUSK = -98, PREV_ONE_IF_UNSET = -97 };
enum { LQ_KEEP = -99, LQ_IF_PRECOND = -98 };
typedef struct {
   int prev, robust, inner, LeftQ, LeftX, RightQ, RightX, SkewQ, SkewX, locking, convTest;
   double relTolBase;
} preset;

static const preset presets[] = {
   /* PRIMME_DEFAULT_METHOD (resolved before lookup) */ {0},
   /* DYNAMIC           */ {PREV_PLUSK, KEEP, -1, LQ_IF_PRECOND, 1, 0, 0, 0, 0, KEEP, -2 /*by target*/, 0},
   /* DEFAULT_MIN_TIME  */ {0},
   /* DEFAULT_MIN_MATVECS */ {0},
   /* Arnoldi           */ {0, KEEP, 0, KEEP, KEEP, KEEP, KEEP, KEEP, KEEP, KEEP, KEEP, 0},
   /* GD                */ {0, 1, 0, KEEP, KEEP, KEEP, 0, KEEP, 0, KEEP, KEEP, 0},
   /* GD_plusK          */ {PREV_PLUSK, KEEP, 0, KEEP, KEEP, KEEP, 0, KEEP, 0, KEEP, KEEP, 0},
   /* GD_Olsen_plusK    */ {PREV_PLUSK, KEEP, 0, KEEP, KEEP, KEEP, 1, KEEP, 0, KEEP, KEEP, 0},
   /* JD_Olsen_plusK    */ {PREV_PLUSK, 1, 0, KEEP, KEEP, KEEP, 1, KEEP, 1, KEEP, KEEP, 0},
   /* RQI               */ {0, 1, -1, 1, 1, 0, 1, 0, 0, 1, primme_full_LTolerance, 0},
   /* JDQR              */ {1, 0, -3 /*10 if unset*/, 0, 1, 1, 1, 1, 1, 1, primme_full_LTolerance, 1.5},
   /* JDQMR             */ {PREV_ONE_IF_UNSET, KEEP, -1, LQ_IF_PRECOND, 1, 0, 0, 0, 1, KEEP, primme_adaptive, 0},
   /* JDQMR_ETol        */ {PREV_ONE_IF_UNSET, KEEP, -1, LQ_IF_PRECOND, 1, 0, 0, 0, 0, KEEP, primme_adaptive_ETolerance, 0},
};

static void set_if(int *dst, int v) { if (v != KEEP) *dst = v; }

int primme_set_method(primme_preset_method method, primme_params *p) {
   if (method == PRIMME_DEFAULT_METHOD) method = PRIMME_DYNAMIC;
   if (method == PRIMME_DEFAULT_MIN_MATVECS) method = PRIMME_GD_Olsen_plusK;
   else if (method == PRIMME_DEFAULT_MIN_TIME)
      method = (p->target == primme_smallest || p->target == primme_largest) ? PRIMME_JDQMR_ETol
                                                                             : PRIMME_JDQMR;
   p->dynamicMethodSwitch = (method == PRIMME_DYNAMIC) ? 1 : 0;
   if (p->maxBlockSize == 0) p->maxBlockSize = 1;
   if (p->correctionParams.precondition == -1)
      p->correctionParams.precondition = p->applyPreconditioner ? 1 : 0;

   correction_params *cp = &p->correctionParams;
   if (method >= PRIMME_DYNAMIC && method <= PRIMME_JDQMR_ETol && method != PRIMME_DEFAULT_MIN_TIME &&
         method != PRIMME_DEFAULT_MIN_MATVECS) {
      const preset *s = &presets[method];
      /* +k retention */
      if (s->prev == PREV_PLUSK) {
         if (p->restartingParams.maxPrevRetain <= 0)
            p->restartingParams.maxPrevRetain =
                  ((p->maxBlockSize == 1 && p->numEvals > 1) || p->massMatrixMatvec) ? 2 : p->maxBlockSize;
      } else if (s->prev == PREV_ONE_IF_UNSET) {
         if (p->restartingParams.maxPrevRetain < 0) p->restartingParams.maxPrevRetain = 1;
      } else if (s->prev != PREV_KEEP) {
         p->restartingParams.maxPrevRetain = s->prev;
      }
      set_if(&cp->robustShifts, s->robust);
      if (s->inner == -3) { if (cp->maxInnerIterations == -INT_MAX) cp->maxInnerIterations = 10; }
      else cp->maxInnerIterations = s->inner;
      if (method == PRIMME_Arnoldi) cp->precondition = 0;
      if (s->LeftQ == LQ_IF_PRECOND) cp->projectors.LeftQ = cp->precondition ? 1 : 0;
      else set_if(&cp->projectors.LeftQ, s->LeftQ);
      set_if(&cp->projectors.LeftX, s->LeftX);
      set_if(&cp->projectors.RightQ, s->RightQ);
      set_if(&cp->projectors.RightX, s->RightX);
      set_if(&cp->projectors.SkewQ, s->SkewQ);
      set_if(&cp->projectors.SkewX, s->SkewX);
      set_if(&p->locking, s->locking);
      if (s->convTest == -2)
         cp->convTest = (p->target == primme_smallest || p->target == primme_largest)
                              ? primme_adaptive_ETolerance : primme_adaptive;
      else if (s->convTest != KEEP) cp->convTest = (primme_convergencetest)s->convTest;
      if (s->relTolBase != 0) cp->relTolBase = s->relTolBase;
   } else if (method == PRIMME_STEEPEST_DESCENT) {
      p->locking = 1;
      p->maxBasisSize = p->numEvals * 2;
      p->minRestartSize = p->numEvals;
      p->maxBlockSize = p->numEvals;
      p->restartingParams.maxPrevRetain = 0;
      cp->robustShifts = 0; cp->maxInnerIterations = 0;
      cp->projectors.RightX = 1; cp->projectors.SkewX = 0;
   } else if (method == PRIMME_LOBPCG_OrthoBasis) {
      p->maxBasisSize = p->numEvals * 3;
      p->minRestartSize = p->numEvals;
      p->maxBlockSize = p->numEvals;
      p->restartingParams.maxPrevRetain = p->numEvals;
      cp->robustShifts = 0; cp->maxInnerIterations = 0;
      cp->projectors.RightX = 1; cp->projectors.SkewX = 0;
      p->initBasisMode = primme_init_random;
   } else if (method == PRIMME_LOBPCG_OrthoBasis_Window) {
      if (p->maxBlockSize == 1 &&
            (p->target == primme_closest_leq || p->target == primme_closest_geq)) {
         p->maxBasisSize = 4; p->minRestartSize = 2; p->restartingParams.maxPrevRetain = 1;
      } else {
         p->maxBasisSize = p->maxBlockSize * 3;
         p->minRestartSize = p->maxBlockSize;
         p->restartingParams.maxPrevRetain = p->maxBlockSize;
      }
      cp->robustShifts = 0; cp->maxInnerIterations = 0;
      cp->projectors.RightX = 1; cp->projectors.SkewX = 0;
      p->initBasisMode = primme_init_random;
   } else {
      return -1;
   }
   primme_set_defaults(p);
   return 0;
}

void primme_set_defaults(primme_params *p) {
   if (p->dynamicMethodSwitch < 0) primme_set_method(PRIMME_DYNAMIC, p);
   if (p->ldevecs == -1 && p->nLocal != -1) p->ldevecs = p->nLocal;
   if (p->projectionParams.projection == primme_proj_default)
      p->projectionParams.projection = primme_proj_RR;
   if (p->initBasisMode == primme_init_default) p->initBasisMode = primme_init_krylov;

   const int extremal = (p->target == primme_smallest || p->target == primme_largest);
   const int prev = p->restartingParams.maxPrevRetain;
   if (p->maxBasisSize == 0) {
      /* the reference writes (int)2.5*minRestartSize / (int)1.7*minRestartSize, i.e. the
       * cast binds first: factors 2 and 1 (primme_interface.c:566, :573) */
      int64_t cap = p->n - p->numOrthoConst;
      int want = extremal ? PA_MAX(PA_MAX(15, 4 * p->maxBlockSize + prev), 2 * p->minRestartSize + prev)
                          : PA_MAX(PA_MAX(35, 5 * p->maxBlockSize + prev), 1 * p->minRestartSize + prev);
      p->maxBasisSize = (int)PA_MIN(cap, (int64_t)want);
   }
   if (p->minRestartSize == 0) {
      if (p->n <= 3) p->minRestartSize = (int)(p->n - p->numOrthoConst);
      else p->minRestartSize = (int)(0.5 + (extremal ? 0.4 : 0.6) * p->maxBasisSize);
      if (p->maxBlockSize > 1) {
         /* so that an integer number of blocks fits between restarts */
         if (prev > 0)
            p->minRestartSize = p->maxBasisSize - p->maxBlockSize *
                  (1 + (int)((p->maxBasisSize - p->minRestartSize - 1 - prev) / (double)p->maxBlockSize)) - prev;
         else
            p->minRestartSize = p->maxBasisSize - p->maxBlockSize *
                  (1 + (int)((p->maxBasisSize - p->minRestartSize - 1) / (double)p->maxBlockSize));
      }
   }
   if (p->locking < 0) {
      if (!extremal) p->locking = 1;
      else p->locking = (p->numEvals > p->minRestartSize) ? 1 : 0;
   }
}

/* Argument checks; codes are the reference's (src/eigs/primme_c.c:438-535). */
int pa_check_input(const void *evals, const void *evecs, const void *resNorms,
      const primme_params *p, double machine_eps) {
   if (!p) return -4;
   const correction_params *cp = &p->correctionParams;
   const int interior = (p->target == primme_largest_abs || p->target == primme_closest_geq ||
                         p->target == primme_closest_leq || p->target == primme_closest_abs);
   if (p->n < 0 || p->nLocal < 0 || p->nLocal > p->n) return -5;
   if (p->numProcs < 1) return -6;
   if (!p->matrixMatvec) return -7;
   if (!p->applyPreconditioner && cp->precondition > 0) return -8;
   if (p->numEvals > p->n) return -10;
   if (p->numEvals < 0) return -11;
   if (p->convTestFun != NULL && fabs(p->eps) != 0.0 && p->eps < machine_eps) return -12;
   if (p->target != primme_smallest && p->target != primme_largest && !interior) return -13;
   if (p->numOrthoConst < 0 || p->numOrthoConst > p->n) return -16;
   if (p->maxBasisSize < 2 && p->n > 2) return -17;
   if (p->minRestartSize < 0 || (p->minRestartSize == 0 && p->n > 2 && p->numEvals > 0)) return -18;
   if (p->maxBlockSize < 0 || (p->maxBlockSize == 0 && p->numEvals > 0)) return -19;
   if (p->restartingParams.maxPrevRetain < 0) return -20;
   if (p->initSize < 0) return -22;
   if (p->locking == 0 && p->initSize > p->maxBasisSize) return -23;
   if (p->locking > 0 && p->initSize > p->numEvals) return -24;
   if (p->minRestartSize + p->restartingParams.maxPrevRetain >= p->maxBasisSize &&
         p->n > p->maxBasisSize) return -25;
   if (p->minRestartSize > p->n && p->n > 2) return -26;
   if (p->printLevel < 0 || p->printLevel > 5) return -27;
   if (cp->convTest != primme_full_LTolerance && cp->convTest != primme_decreasing_LTolerance &&
         cp->convTest != primme_adaptive_ETolerance && cp->convTest != primme_adaptive) return -28;
   if (cp->convTest == primme_decreasing_LTolerance && cp->relTolBase <= 1.0) return -29;
   if (!evals) return -30;
   if (!evecs || !hipk_is_device_ptr(evecs)) return -31;
   if (!resNorms) return -32;
   if (p->locking == 0 && p->minRestartSize < p->numEvals && p->n > 2) return -33;
   if (p->ldevecs < p->nLocal) return -34;
   if (p->ldOPs != 0 && p->ldOPs < p->nLocal) return -35;
   if (p->locking == 0 && (p->target == primme_closest_leq || p->target == primme_closest_geq)) return -38;
   if (p->massMatrixMatvec && p->projectionParams.projection != primme_proj_RR) return -39;
   if (interior) {
      if (p->numTargetShifts <= 0) return -14;
      if (!p->targetShifts) return -15;
   }
   return 0;
}
