/* eigs_dynamic.c — run-time choice between GD+k and JDQMR_ETol (preset PRIMME_DYNAMIC).
 *
 * Restates the cost model of reference src/eigs/main_iter.c:
 *   model state                 <- :66-110 (primme_CostModel), initializeModel :2406-2440
 *   pa_dyn_observe              <- update_statistics :2155-2341
 *   pa_dyn_leave_gd             <- switch_from_GDpk   :2047-2103
 *   pa_dyn_leave_jdqmr          <- switch_from_JDQMR  :1943-2011
 *   expected-time ratio         <- ratio_JDQMR_GDpk   :2352-2363, update_slowdown :2373-2401
 *   pa_dyn_recommend            <- :1221-1228 / :1296-1303
 *
 * The model is fed with wall-clock measurements; on this path the operator and the
 * correction step synchronise the stream while the dynamic method is active (see
 * pa_solver.phase_timing), so the clock sees device time, not launch time.  The sequence of
 * switches therefore depends on the machine, exactly as in the reference; the converged
 * eigenpairs do not.
 *
 * dynamicMethodSwitch states: 1/3 = running GD+k (few / many eigenpairs wanted),
 * 2/4 = running JDQMR_ETol; after the solve -1/-2/-3 recommend GD+k / JDQMR_ETol / dynamic. */
#include "eigs_solver.h"
#include <math.h>

int pa_reduce_host(pa_solver *s, double *buf, int count);

void pa_dyn_init(pa_cost_model *c, const primme_params *p) {
   c->t_mv_pr = c->t_mv = c->t_pr = 0.0;
   c->t_qmr = c->t_qmr_mv_pr = 0.0;
   c->t_gd_mv_pr = c->t_gd_mv = 0.0;
   c->rate_gd = c->rate_jd = 1e-4;
   c->slowdown = 1.5;
   c->mv_per_outer = 0.0;
   c->next_reset = 1;
   c->logred_gd = c->logred_jd = 0.0;
   c->mv_gd = c->mv_jd = 0.0;
   c->found_gd = c->found_jd = 0;
   c->mv0 = (int)p->stats.numMatvecs;
   c->it0 = (int)p->stats.numOuterIterations + 1;
   c->t0 = pa_wtime();
   c->t_inner = 0.0;
   c->res0 = -1.0;
   c->acc_jd = c->acc_gd = 0.0;
   c->acc_ratio = 1.0;
}

/* log(rate_gd)/log(rate_jd), clipped to what an inner-outer method can do */
static void refresh_slowdown(pa_cost_model *c) {
   double sd;
   const double g = c->rate_gd, j = c->rate_jd;
   if (g < 1.0) sd = (j < 1.0) ? log(g) / log(j) : (j == 1.0 ? 2.5 : -log(g) / log(j));
   else if (g == 1.0) sd = 1.1;
   else sd = (j < 1.0) ? log(g) / log(j) : (j == 1.0 ? 1.1 : log(j) / log(g));
   sd = PA_MAX(c->mv_per_outer / (c->mv_per_outer - 1.0), PA_MIN(sd, c->mv_per_outer));
   c->slowdown = PA_MAX(1.1, PA_MIN(sd, 2.5));
}

/* expected time(JDQMR) / time(GD+k) */
static double expected_ratio(const pa_cost_model *c, double slowdown, double mv_per_outer) {
   return slowdown * (c->t_qmr_mv_pr + (c->t_gd_mv - c->t_qmr - c->t_qmr_mv_pr) / mv_per_outer) / c->t_gd_mv_pr;
}

/* returns 1 when the model was updated and the methods may be compared */
int pa_dyn_observe(pa_cost_model *c, primme_params *p, double now, int recentConv, int atRestart,
      int numConverged, double currentResNorm) {
   const double elapsed = now - c->t0, t_outer = elapsed - c->t_inner;
   int kout = (int)p->stats.numOuterIterations - c->it0;
   const int nMV = (int)p->stats.numMatvecs - c->mv0;
   if (atRestart) kout++;
   if (kout == 0) return 0;
   const double kinn = (double)nMV / kout - 2;
   const int inJD = (p->correctionParams.maxInnerIterations == -1);
   if (inJD && kinn < 1.0 && c->t_qmr == 0.0) return 0;

   double low_res;
   if (recentConv > 0) {
      low_res = p->stats.maxConvTol;
      if (inJD) c->found_jd += recentConv; else c->found_gd += recentConv;
   } else {
      low_res = currentResNorm;
   }

   c->t_gd_mv = (c->t_gd_mv == 0.0) ? t_outer / kout : (c->t_gd_mv + t_outer / kout) / 2.0;

   if (numConverged / 10 >= c->next_reset) {
      c->logred_gd /= c->found_gd; c->mv_gd /= c->found_gd;
      c->logred_jd /= c->found_jd; c->mv_jd /= c->found_jd;
      c->next_reset = numConverged / 10 + 1;
      c->found_gd = c->found_jd = 1;
   }

   if (p->dynamicMethodSwitch == 1 || p->dynamicMethodSwitch == 3) {
      c->t_pr = (c->t_pr == 0.0) ? c->t_inner / kout : (c->t_pr + c->t_inner / kout) / 2.0;
      c->t_gd_mv_pr = c->t_gd_mv + c->t_pr;
      c->t_mv_pr = c->t_mv + c->t_pr;
      if (low_res <= c->res0) c->logred_gd += log(low_res / c->res0);
      c->mv_gd += nMV;
      c->rate_gd = exp(c->logred_gd / c->mv_gd);
   } else if (p->dynamicMethodSwitch == 2 || p->dynamicMethodSwitch == 4) {
      const double per_inner = (c->t_inner / kout - c->t_mv_pr) / kinn;
      if (c->t_qmr_mv_pr == 0.0) {
         c->t_qmr_mv_pr = per_inner;
         c->mv_per_outer = (double)nMV / kout;
      } else {
         if (kinn != 0.0) c->t_qmr_mv_pr = (c->t_qmr_mv_pr + per_inner) / 2.0;
         c->mv_per_outer = (c->mv_per_outer + (double)nMV / kout) / 2;
      }
      c->t_qmr = c->t_qmr_mv_pr - c->t_mv_pr;
      c->t_gd_mv_pr = c->t_gd_mv + c->t_pr;
      if (low_res <= c->res0) c->logred_jd += log(low_res / c->res0);
      c->mv_jd += nMV;
      c->rate_jd = exp(c->logred_jd / c->mv_jd);
   }
   refresh_slowdown(c);

   c->it0 = (int)p->stats.numOuterIterations + (atRestart ? 1 : 0);
   c->mv0 = (int)p->stats.numMatvecs;
   c->t0 = now;
   c->t_inner = 0.0;
   c->res0 = currentResNorm;
   return 1;
}

static int averaged_ratio(pa_solver *s, double *ratio) {
   if (s->parallel) {
      int rc = pa_reduce_host(s, ratio, 1);
      if (rc) return rc;
      *ratio /= (double)s->p->numProcs;
   }
   return 0;
}

static void account(pa_cost_model *c, double ratio) {
   c->acc_jd += c->t_gd_mv_pr * ratio;
   c->acc_gd += c->t_gd_mv_pr;
   c->acc_ratio = c->acc_jd / c->acc_gd;
}

static void use_jdqmr(primme_params *p, int state) {
   p->dynamicMethodSwitch = state;
   p->correctionParams.maxInnerIterations = -1;
   p->correctionParams.projectors.RightX = 0;
}
static void use_gd(primme_params *p, int state) {
   p->dynamicMethodSwitch = state;
   p->correctionParams.maxInnerIterations = 0;
   p->correctionParams.projectors.RightX = 1;
}

/* GD+k is running (state 1 or 3): move to JDQMR_ETol if it is expected to be 5 % faster, or
 * unconditionally the first time to obtain its timings */
int pa_dyn_leave_gd(pa_solver *s, pa_cost_model *c) {
   primme_params *p = s->p;
   if (p->stats.numRestarts == 0 || p->maxBasisSize + (p->locking ? p->numEvals : 0) >= p->n) return 0;
   const int to = (p->dynamicMethodSwitch == 1) ? 2 : 4;
   if (c->t_qmr == 0.0) { use_jdqmr(p, to); return 0; }
   double ratio = expected_ratio(c, c->slowdown, c->mv_per_outer);
   CHK(averaged_ratio(s, &ratio));
   if (ratio < 0.95) use_jdqmr(p, to);
   account(c, ratio);
   return 0;
}

/* JDQMR_ETol is running (state 2 or 4) */
int pa_dyn_leave_jdqmr(pa_solver *s, pa_cost_model *c) {
   primme_params *p = s->p;
   double ratio;
   if (p->dynamicMethodSwitch == 2) {
      /* few eigenpairs: if even the best case for JDQMR loses, stay with GD+k for good */
      ratio = expected_ratio(c, 1.1, 1000.0);
      CHK(averaged_ratio(s, &ratio));
      if (ratio > 1.05) { use_gd(p, -1); return 0; }
   }
   const int to = (p->dynamicMethodSwitch == 2) ? 1 : 3;
   ratio = expected_ratio(c, c->slowdown, c->mv_per_outer);
   CHK(averaged_ratio(s, &ratio));
   if (ratio > 1.05) use_gd(p, to);
   account(c, ratio);
   return 0;
}

/* method recommendation for later runs, left in dynamicMethodSwitch */
void pa_dyn_recommend(const pa_cost_model *c, primme_params *p) {
   if (p->dynamicMethodSwitch <= 0) return;
   p->dynamicMethodSwitch = (c->acc_ratio < 0.96) ? -2 : (c->acc_ratio > 1.04 ? -1 : -3);
}
