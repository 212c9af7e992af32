/* eigs_callbacks.c — user callbacks invoked with the operand type they declare.
 *
 * The solver keeps every host-side scalar in double.  The reference hands callbacks data of
 * `*_type` (globalSumReal_type, convTestFun_type, monitorFun_type), which it defaults to the
 * working precision of the entry point — float for sprimme / cprimme (reference
 * src/eigs/primme_c.c:170-183) — and converts around the call (globalSum_Tprimme,
 * src/eigs/auxiliary_eigs.c:391-427; convTestFun_Sprimme :600-640; monitorFun_Sprimme
 * primme_c.c:702-800).  An application ported from cublas_sprimme whose globalSumReal does
 * MPI_Allreduce(MPI_FLOAT) therefore keeps working: these helpers do the same conversions.
 * Only primme_op_float and primme_op_double are accepted (half / quad: PRIMME_FUNCTION_UNAVAILABLE).
 */
#include <stdio.h>
#include "eigs_internal.h"
#include "primme_amd_svds.h"
#include <stdlib.h>

static int type_ok(primme_op_datatype t) { return t == primme_op_double || t == primme_op_float || t == primme_op_default; }

/* in-place global sum of `count` doubles through primme->globalSumReal */
int pa_call_global_sum(primme_params *p, double *buf, int count) {
   int ierr = 0, cnt = count;
   if (count <= 0 || !p->globalSumReal) return 0;
   if (!type_ok(p->globalSumReal_type)) return PRIMME_FUNCTION_UNAVAILABLE;
   if (p->globalSumReal_type == primme_op_float) {
      float *f = (float *)malloc(sizeof(float) * (size_t)count);
      if (!f) return PRIMME_MALLOC_FAILURE;
      for (int i = 0; i < count; i++) f[i] = (float)buf[i];
      p->globalSumReal(f, f, &cnt, p, &ierr);
      for (int i = 0; i < count; i++) buf[i] = (double)f[i];
      free(f);
   } else {
      p->globalSumReal(buf, buf, &cnt, p, &ierr);
   }
   return ierr ? PRIMME_USER_FAILURE : 0;
}

int pa_svds_call_global_sum(primme_svds_params *ps, double *buf, int count) {
   int ierr = 0, cnt = count;
   if (count <= 0 || !ps->globalSumReal) return 0;
   if (!type_ok(ps->globalSumReal_type)) return PRIMME_FUNCTION_UNAVAILABLE;
   if (ps->globalSumReal_type == primme_op_float) {
      float *f = (float *)malloc(sizeof(float) * (size_t)count);
      if (!f) return PRIMME_MALLOC_FAILURE;
      for (int i = 0; i < count; i++) f[i] = (float)buf[i];
      ps->globalSumReal(f, f, &cnt, ps, &ierr);
      for (int i = 0; i < count; i++) buf[i] = (double)f[i];
      free(f);
   } else {
      ps->globalSumReal(buf, buf, &cnt, ps, &ierr);
   }
   return ierr ? PRIMME_USER_FAILURE : 0;
}

/* primme->convTestFun(eval, evec, rNorm, isconv): eval / rNorm in convTestFun_type; evec stays in
 * the solver's working type (a device pointer), as in the reference's GPU flavour */
int pa_call_conv_test(primme_params *p, double eval, void *evec, double rnorm, int *isconv) {
   int ierr = 0;
   if (!type_ok(p->convTestFun_type)) return PRIMME_FUNCTION_UNAVAILABLE;
   if (p->convTestFun_type == primme_op_float) {
      float e = (float)eval, r = (float)rnorm;
      p->convTestFun((double *)(void *)&e, evec, (double *)(void *)&r, isconv, p, &ierr);
   } else {
      p->convTestFun(&eval, evec, &rnorm, isconv, p, &ierr);
   }
   return ierr ? PRIMME_USER_FAILURE : 0;
}

int pa_svds_call_conv_test(primme_svds_params *ps, double sval, void *leftsvec, void *rightsvec, double rnorm,
      int *method, int *isconv) {
   int ierr = 0;
   if (!type_ok(ps->convTestFun_type)) return PRIMME_FUNCTION_UNAVAILABLE;
   if (ps->convTestFun_type == primme_op_float) {
      float s = (float)sval, r = (float)rnorm;
      ps->convTestFun((double *)(void *)&s, leftsvec, rightsvec, (double *)(void *)&r, method, isconv, ps, &ierr);
   } else {
      ps->convTestFun(&sval, leftsvec, rightsvec, &rnorm, method, isconv, ps, &ierr);
   }
   return ierr ? PRIMME_USER_FAILURE : 0;
}

static float *narrow(const double *x, int n) {
   if (!x || n <= 0) return NULL;
   float *f = (float *)malloc(sizeof(float) * (size_t)n);
   if (f) for (int i = 0; i < n; i++) f[i] = (float)x[i];
   return f;
}

/* primme->monitorFun with the value arrays in monitorFun_type */
int pa_call_monitor(primme_params *p, double *basisEvals, int basisSize, int *basisFlags, int *iblock, int blockSize,
      double *basisNorms, int numConverged, double *lockedEvals, int numLocked, int *lockedFlags, double *lockedNorms,
      primme_event event) {
   int err = 0;
   double time = 0.0;
   if (!p->monitorFun) return 0;
   if (!type_ok(p->monitorFun_type)) return PRIMME_FUNCTION_UNAVAILABLE;
   if (p->monitorFun_type == primme_op_float) {
      float *be = narrow(basisEvals, basisSize), *bn = narrow(basisNorms, basisSize);
      float *le = narrow(lockedEvals, numLocked), *ln = narrow(lockedNorms, numLocked);
      p->monitorFun(be, &basisSize, basisFlags, iblock, &blockSize, bn, &numConverged, le, &numLocked, lockedFlags, ln,
            NULL, NULL, NULL, &time, &event, p, &err);
      free(be); free(bn); free(le); free(ln);
   } else {
      p->monitorFun(basisEvals, &basisSize, basisFlags, iblock, &blockSize, basisNorms, &numConverged, lockedEvals,
            &numLocked, lockedFlags, lockedNorms, NULL, NULL, NULL, &time, &event, p, &err);
   }
   return err ? PRIMME_USER_FAILURE : 0;
}

/* the report of one inner (QMR) step: primme_event_inner_iteration with the arguments the reference passes
 * (inner_solve.c:550-558 adaptive tests, :581-588 otherwise; auxiliary_eigs_normal.c:446-490) */
int pa_call_monitor_inner(primme_params *p, double eval, double resNorm, int counts, int innerIts, double lsRes) {
   int err = 0, one = 1, zero = 0, unconv = 0 /* UNCONVERGED */, nconv = counts, nlock = counts;
   double time = 0.0;
   primme_event event = primme_event_inner_iteration;
   if (!p->monitorFun) {
      /* the reference's default report of an inner step (primme_c.c:653-666), same line so that logs stay comparable */
      if (p->outputFile && p->procID == 0 && p->printLevel >= 4)
         fprintf(p->outputFile, "INN MV %lld Sec %e Eval %13E Lin|r| %.3e EV|r| %.3e\n", (long long)p->stats.numMatvecs,
               p->stats.elapsedTime, eval, lsRes, resNorm);
      return 0;
   }
   if (!type_ok(p->monitorFun_type)) return PRIMME_FUNCTION_UNAVAILABLE;
   if (p->monitorFun_type == primme_op_float) {
      float e = (float)eval, r = (float)resNorm, t = (float)lsRes;
      p->monitorFun(&e, &one, &unconv, &zero, &one, &r, &nconv, NULL, &nlock, NULL, NULL, &innerIts, lsRes >= 0 ? &t : NULL, NULL,
            &time, &event, p, &err);
   } else {
      p->monitorFun(&eval, &one, &unconv, &zero, &one, &resNorm, &nconv, NULL, &nlock, NULL, NULL, &innerIts,
            lsRes >= 0 ? &lsRes : NULL, NULL, &time, &event, p, &err);
   }
   return err ? PRIMME_USER_FAILURE : 0;
}
