/* eigs_ops.c — solver steps that touch the HBM-resident basis: the matvec wrapper,
 * reductions (with the optional all-reduce), classical Gram-Schmidt with Daniel's
 * reorthogonalisation test, the projection update and the Rayleigh-Ritz solve.
 *
 * Each function names the reference routine whose behaviour it restates; the
 * structure differs because the basis never leaves the device and only the few
 * scalars the control flow needs come back to the host (one synchronisation per
 * CGS pass instead of the reference GPU backend's one per BLAS call).
 */
#include "eigs_solver.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#if !PA_IS_COMPLEX      /* type-independent pieces exist once (the real objects) */
double pa_wtime(void) {
   struct timespec ts;
   clock_gettime(CLOCK_MONOTONIC, &ts);
   return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ---- problem norm / machine epsilons (reference auxiliary_eigs.c:498-591) ------ */
double pa_problem_norm(int overrideUser, const primme_params *p) {
   if (p->massMatrixMatvec) {      /* an estimate of |B^-1 A| (auxiliary_eigs.c:567-591) */
      const double user = (p->aNorm > 0.0 && p->invBNorm > 0.0) ? p->aNorm * p->invBNorm : 0.0;
      if (!overrideUser) return user > 0.0 ? user : p->stats.estimateLargestSVal;
      return PA_MAX(user, p->stats.estimateLargestSVal);
   }
   if (!overrideUser) return p->aNorm > 0.0 ? p->aNorm : p->stats.estimateLargestSVal;
   return PA_MAX(p->aNorm > 0.0 ? p->aNorm : 0.0, p->stats.estimateLargestSVal);
}

/* ---- reductions -----------------------------------------------------------------
 * d_buf (device) holds `count` local partial sums.  On return s->h_red[0..count)
 * holds the global sums; if keep_dev, d_buf holds them as well (the next kernel
 * uses them as coefficients).  Restates globalSum_Tprimme (reference
 * auxiliary_eigs.c:391-427) for device-resident partials.
 * When defer_sync is set and no host callback is involved, the device->host copy
 * is only enqueued; the caller must hipk_sync before reading h_red. */
int pa_reduce(pa_solver *s, double *d_buf, int count, int keep_dev, int defer_sync) {
   primme_params *p = s->p;
   if (count <= 0) return 0;
   /* d_buf lies in [d_red, d_red + 2*red_cap): the second half holds the fused kernel's overlaps */
   if (d_buf < s->d_red || (size_t)(d_buf - s->d_red) + (size_t)count > 3 * (size_t)s->red_cap + 64) return PRIMME_UNEXPECTED_FAILURE;
   const int parallel = s->parallel;
   double t0 = parallel ? pa_wtime() : 0.0;
   if (parallel && s->dev_comm) {
      if (hipk_xreduce_covered(s->ctx, d_buf, count)) {
         /* the producing launch's second stage exchanged its sums with the other ranks itself (peer-to-peer
          * transport, hipk_xreduce_arm): d_buf and the pinned mirror already hold the global sums */
         if (!defer_sync) CHK(hipk_wait_results(s->ctx));
      } else {
      /* peer-to-peer transport: reduction, mirror and completion flag are one launch */
      int rcp = pa_comm_allreduce_publish(p->commInfo, s->ctx, d_buf, count);
      if (rcp < 0) return PRIMME_PARALLEL_FAILURE;
      if (rcp == 0) {
         if (!defer_sync) CHK(hipk_wait_results(s->ctx));
      } else {
      CHK(pa_comm_allreduce_device(p->commInfo, d_buf, count, hipk_ctx_stream(s->ctx)));
      /* the global sums reach the pinned mirror through a one-block launch that also publishes the completion flag
       * (the wait is then a spin on pinned memory, as on one rank); fallback: copy + stream synchronisation */
      static int no_publish = -1;      /* PRIMME_AMD_NO_PUBLISH=1: measurement knob, read once */
      if (no_publish < 0) no_publish = getenv("PRIMME_AMD_NO_PUBLISH") != NULL;
      rcp = no_publish ? 1 : hipk_publish_results(s->ctx, d_buf, count);
      if (rcp < 0) return PRIMME_UNEXPECTED_FAILURE;
      if (rcp == 0) {
         if (!defer_sync) CHK(hipk_wait_results(s->ctx));
      } else {
         CHK(hipk_d2h(s->ctx, s->h_red + (d_buf - s->d_red), d_buf, (size_t)count * sizeof(double)));
         if (!defer_sync) CHK(hipk_sync(s->ctx));
      }
      }
      }
      if (!defer_sync && pa_comm_failed(p->commInfo)) return PRIMME_PARALLEL_FAILURE;
   } else {
      /* the reduction kernels already stored the local sums in h_red (zero-copy mirror) */
      if (parallel) {
         CHK(hipk_sync(s->ctx));
         double *hb = s->h_red + (d_buf - s->d_red);
         CHK(pa_call_global_sum(p, hb, count));
         if (keep_dev) CHK(hipk_h2d(s->ctx, d_buf, hb, (size_t)count * sizeof(double)));
      } else if (!defer_sync) {
         /* the reduction whose result is wanted was the last launch: wait for its completion flag */
         CHK(hipk_wait_results(s->ctx));
      }
   }
   if (parallel) {
      p->stats.numGlobalSum++;
      p->stats.volumeGlobalSum += count;
      p->stats.timeGlobalSum += pa_wtime() - t0;
   }
   return 0;
}

/* global sum of a few host scalars (cost-model ratios): through the same path as the
 * device partials so that every communicator flavour is covered */
/* G = W(:,0:k)' Q for the locked / constraint vectors Q (projection column from W'r,
 * eigs_conv.c): one TN panel product after a restart. */
int pa_refresh_wtq(pa_solver *s, int basisSize, int nLk) {
   if (s->fov_carry && s->wtq_rows == basisSize && s->wtq_L == nLk) return 0;   /* the restart transformed it */
   s->wtq_rows = -1; s->wtq_L = nLk;
   if (!s->wtr_enabled || !s->wtq || basisSize > HIPK_WTR_MAX_K || nLk > HIPK_WTR_MAX_K) return 0;
   if (nLk > 0 && basisSize > 0) {
      if (basisSize * nLk > s->red_cap) return 0;
      hipk_seg seg = {s->W, s->ld, basisSize};
      CHK(hipk_panel_dots(s->ctx, s->dt, s->m, &seg, 1, s->evecs, s->ldevecs, nLk, s->d_red, basisSize));
      CHK(pa_reduce(s, s->d_red, basisSize * nLk, 0, 0));
      for (int l = 0; l < nLk; l++)
         for (int j = 0; j < basisSize; j++) s->wtq[j + (size_t)l * s->K] = s->h_red[j + (size_t)l * basisSize];
   }
   s->wtq_rows = basisSize;
   return 0;
}

int pa_trace_errors(void) {
   static int on = -1;
   if (on < 0) on = getenv("PRIMME_AMD_TRACE_ERRORS") != NULL;
   return on;
}

int pa_reduce_host(pa_solver *s, double *buf, int count) {
   if (count <= 0 || !s->parallel) return 0;
   CHK(hipk_sync(s->ctx));
   memcpy(s->h_red, buf, (size_t)count * sizeof(double));
   if (s->dev_comm) CHK(hipk_h2d(s->ctx, s->d_red, s->h_red, (size_t)count * sizeof(double)));
   CHK(pa_reduce(s, s->d_red, count, 0, 0));
   memcpy(buf, s->h_red, (size_t)count * sizeof(double));
   return 0;
}

#endif   /* !PA_IS_COMPLEX */
#if PA_IS_COMPLEX
double pa_problem_norm(int overrideUser, const primme_params *p);
int pa_reduce(pa_solver *s, double *d_buf, int count, int keep_dev, int defer_sync);
int pa_matvec(pa_solver *s, char *Vp, int64_t ldV, char *Wp, int64_t ldW, int c0, int nc);
int pa_precond(pa_solver *s, char *X, int64_t ldX, char *Y, int64_t ldY, int nc);
#else
/* ---- W(:,c0:c0+nc) = A * V(:,c0:c0+nc)  (reference auxiliary_eigs.c:183-230) ---- */
int pa_matvec(pa_solver *s, char *Vp, int64_t ldV, char *Wp, int64_t ldW, int c0, int nc) {
   primme_params *p = s->p;
   if (nc <= 0) return 0;
   double t0 = pa_wtime();
   int ierr = 0;
   PRIMME_INT ldx = ldV, ldy = ldW;
   p->matrixMatvec(PCOL(s, Vp, ldV, c0), &ldx, PCOL(s, Wp, ldW, c0), &ldy, &nc, p, &ierr);
   if (ierr) return PRIMME_USER_FAILURE;
   if (s->phase_timing) CHK(hipk_sync(s->ctx));
   p->stats.timeMatvec += pa_wtime() - t0;
   p->stats.numMatvecs += nc;
   return 0;
}

/* ---- Y(:,0:nc) = B * X(:,0:nc)  (reference massMatrixMatvec_Sprimme, auxiliary_eigs.c:250-290) ---- */
int pa_apply_B(pa_solver *s, char *X, int64_t ldX, char *Y, int64_t ldY, int nc) {
   primme_params *p = s->p;
   if (nc <= 0) return 0;
   double t0 = pa_wtime();
   /* in blocks of the width the callback was promised (maxBlockSize; the initial basis hands over more at once, init.c:210) */
   int ierr = 0;
   PRIMME_INT ldx = ldX, ldy = ldY;
   p->massMatrixMatvec(X, &ldx, Y, &ldy, &nc, p, &ierr);
   if (ierr) return PRIMME_USER_FAILURE;
   if (s->phase_timing) CHK(hipk_sync(s->ctx));
   p->stats.timeMatvec += pa_wtime() - t0;
   p->stats.numMatvecs += nc;      /* the reference counts applications of B with those of A (auxiliary_eigs.c:283) */
   return 0;
}

/* ---- y = K^-1 x or copy (reference auxiliary_eigs.c:317-364) -------------------- */
int pa_precond(pa_solver *s, char *X, int64_t ldX, char *Y, int64_t ldY, int nc) {
   primme_params *p = s->p;
   if (nc <= 0) return 0;
   double t0 = pa_wtime();
   if (p->correctionParams.precondition) {
      int ierr = 0;
      PRIMME_INT ldx = ldX, ldy = ldY;
      p->applyPreconditioner(X, &ldx, Y, &ldy, &nc, p, &ierr);
      if (ierr) return PRIMME_USER_FAILURE;
      p->stats.numPreconds += nc;
   } else {
      CHK(hipk_copy_cols(s->ctx, s->dt, s->m, X, ldX, Y, ldY, nc));
   }
   if (s->phase_timing) CHK(hipk_sync(s->ctx));
   p->stats.timePrecond += pa_wtime() - t0;
   return 0;
}

#endif
/* random column (reference blaslapack.c:938-988 Num_larnv: host xLARNV then upload; complex entries take two
 * consecutive numbers of the stream, (re, im), as zlarnv does) */
int pa_random_col(pa_solver *s, char *col) {
   s->fov_valid = 0;
   pa_pre_discard(s);
   /* generated on the device (hipk_larnv_uniform11): the same stream, nothing crosses PCIe */
   int64_t seed[4] = {s->p->iseed[0], s->p->iseed[1], s->p->iseed[2], s->p->iseed[3]};
   CHK(hipk_larnv_uniform11(s->ctx, s->dt, seed, s->m * SD, col));
   for (int i = 0; i < 4; i++) s->p->iseed[i] = seed[i];
   return 0;
}

/* ---- classical Gram-Schmidt with reorthogonalisation ---------------------------
 * Restates Bortho_gen_Sprimme (reference src/eigs/ortho.c:123-360) for B = I:
 * orthonormalise columns b1..b2 of Vp against its columns 0..b1-1, against
 * `locked` (numLocked columns) and among themselves.  Daniel's test with
 * threshold sqrt(2)/2, at most 3 passes before the column is replaced by a random
 * vector, at most 10 replacements.  RLocked (host, optional) accumulates
 * locked' * v of the first passes (used by the practical-convergence test,
 * reference main_iter.c:742-775).
 *
 * Per pass one fused chain runs on the device with no host round trip in between:
 *    dots([Vp(0:i) | locked | v]' v) -> (all-reduce) -> v -= [Vp locked]*overlaps,
 *    |v|^2 -> (all-reduce) -> copy back
 * and the host synchronises once to apply the test.
 */
int pa_ortho_cgs(pa_solver *s, char *Vp, int64_t ldV, int b1, int b2, char *locked,
      int64_t ldLocked, int numLocked, HS *RLocked, int ldRLocked, int *b2_out) {
   primme_params *p = s->p;
   const int maxNumOrthos = 3, maxNumRandoms = 10;
   const double tol = sqrt(2.0) / 2.0;
   const double eps_orth = s->mach_eps;
   const int parallel = s->parallel;
   double t0 = pa_wtime();

   if (RLocked)
      for (int c = 0; c <= b2 - b1; c++)
         for (int j = 0; j < numLocked; j++) RLocked[j + (size_t)c * ldRLocked] = 0.0;
   if (b2_out) *b2_out = b1;
   /* overlaps left by the fused residual pass (or carried over a restart) belong to ONE column */
   if (s->fov_valid && !(b1 == b2 && b1 == s->fov_k && Vp == s->V)) { s->fov_valid = 0; s->spec2_valid = 0; }
   if (!(b1 == b2 && s->fov_valid)) pa_pre_discard(s);      /* an iteration enqueued ahead assumed the tail of THIS column stands */
   s->fov_carry = 0;

   for (int i = b1; i <= b2; i++) {
      int nOrth = 0, randomizations = 0, updateR = RLocked ? 1 : 0;
      double s0 = 0.0, s02 = 0.0, s1 = 0.0, s12 = 0.0;
      char *v = PCOL(s, Vp, ldV, i);
      for (;;) {
         if (nOrth >= maxNumOrthos) {
            updateR = 0;
            if (randomizations >= maxNumRandoms) goto done;
            CHK(pa_random_col(s, v));
            randomizations++;
            nOrth = 0;
         }
         nOrth++;
         const int first = (nOrth == 1);
         const int nov = i + numLocked;           /* overlaps proper                  */
         const int ndot = nov + (first ? 1 : 0);  /* + |v|^2 on the first pass        */
         hipk_seg segs[3] = {{Vp, ldV, i}, {locked, ldLocked, numLocked}, {v, ldV, first ? 1 : 0}};
         /* overlaps of the first pass may already be there: the fused residual kernel computed
          * [Vp' v | locked' v | v'v] while it produced v (eigs_conv.c) */
         const int use_fov = !PA_IS_COMPLEX && first && randomizations == 0 && s->fov_valid && Vp == s->V && v == s->fov_col &&
                             i == s->fov_k && numLocked == s->fov_L && (numLocked == 0 || locked == s->evecs);
         double *dbase = use_fov ? s->d_fov : s->d_red;
         double *hbase = use_fov ? s->h_fov : s->h_red;
         /* layout of the buffer: nov (+1) scalars of overlaps, then the REAL squared norm of the update at
          * double offset s1_off */
         const int s1_off = use_fov ? s->fov_s1_off : SD * (nov + 1);
         double *d_ov = dbase, *d_s1 = dbase + s1_off;
         if (first) s->fov_valid = 0;
         if (!use_fov) {
            CHK(hipk_panel_dots(s->ctx, s->dt, s->m, segs, 3, v, ldV, 1, d_ov, ndot));
            if (parallel && !s->dev_comm) {
               CHK(pa_reduce(s, d_ov, SD * ndot, 1, 0));
            } else if (parallel) {
               CHK(pa_reduce(s, d_ov, SD * ndot, 1, 1));
            }
         }
         p->stats.numOrthoInnerProds += ndot;
         p->stats.numOrthoInnerProds += nov + 1;
         const int speculated = use_fov && s->fov_projected;
         if (!speculated) {
            CHK(hipk_panel_project(s->ctx, s->dt, s->m, segs, 2, d_ov, nov > 0 ? nov : 1, v, ldV, 1, d_s1));
            CHK(pa_reduce(s, d_s1, 1, 0, 0)); /* synchronises: hbase now has overlaps, s02, s12 */
         }
         s->fov_projected = 0;
         /* the speculative tail (normalise on device, operator, projection) stands only if THIS
          * first pass is the last one */
         const int tail_done = speculated && s->spec2_valid && s->spec2_k == i && b1 == b2;
         if (!tail_done) { s->spec2_valid = 0; pa_pre_discard(s); }
         /* fused tail (eigs_conv.c): v holds the NORMALISED vector, the projected un-normalised one is in
          * T(:,0); whenever the tail is not accepted as it stands it goes back into v first */
         const int fused_pending = speculated && s->spec_fused;
         s->spec_fused = 0;

         if (updateR)
            for (int j = 0; j < numLocked; j++) RLocked[j + (size_t)(i - b1) * ldRLocked] += ((const HS *)hbase)[i + j];
         if (first) { s02 = HS_RE(((const HS *)hbase)[nov]); s0 = sqrt(s02); }
         s12 = hbase[s1_off];
         s1 = sqrt(s12);

         if (!isfinite(s0) || !isfinite(s1) || s1 <= eps_orth * s0) {
            nOrth = maxNumOrthos;             /* lost all significant digits: randomise */
         } else if (s1 <= tol * s0) {
            s0 = s1; s02 = s12;               /* another pass */
            if (fused_pending) CHK(hipk_copy_cols(s->ctx, s->dt, s->m, TCOL(s, s->spec_tcol), s->ld, v, ldV, 1));
         } else {
            double inv = 1.0 / s1;
            if (isfinite(inv)) {
               if (!tail_done) {
                  if (fused_pending) CHK(hipk_copy_cols(s->ctx, s->dt, s->m, TCOL(s, s->spec_tcol), s->ld, v, ldV, 1));
                  CHK(hipk_scale_cols(s->ctx, s->dt, s->m, v, ldV, 1, &inv));
               }
               break;
            }
            nOrth = maxNumOrthos;
         }
         s->spec2_valid = 0;   /* another pass or a random restart: the tail used a stale vector */
      }
      if (b2_out) *b2_out = i + 1;
   }
done:
   if (s->phase_timing) CHK(hipk_sync(s->ctx));
   p->stats.timeOrtho += pa_wtime() - t0;
   return 0;
}

/* Host-only variant for the k x k coefficient vectors (reference ortho.c:395-415
 * Bortho_local with primme == NULL: up to 7 passes, always reorthogonalise).
 * Orthonormalises the single vector x (length n) against Q (n x nQ, ldQ) in the
 * G inner product (G upper-stored n x n, or NULL for identity).  *R receives the
 * norm left after projection (0 if the vector had to be randomised). */
int pa_ortho_local_vec(HS *x, int n, const HS *Q, int ldQ, int nQ, const HS *G,
      int ldG, double *R, int64_t iseed[4]) {
   const int maxNumOrthos = 7, maxNumRandoms = 10;
   const double tol = sqrt(2.0) / 2.0;
   HS *ov = (HS *)malloc((size_t)(nQ + 1) * sizeof(HS));
   HS *Bx = (HS *)malloc((size_t)(n > 0 ? n : 1) * sizeof(HS));
   if (!ov || !Bx) { free(ov); free(Bx); return PRIMME_MALLOC_FAILURE; }
   int nOrth = 0, randomizations = 0, updateR = 1, rc = -3;
   double s0 = 0, s02 = 0, s1 = 0, s12 = 0;
   *R = 0.0;
#define APPLY_G(dst, src)                                                           \
   do {                                                                             \
      if (!G) memcpy(dst, src, (size_t)n * sizeof(HS));                             \
      else for (int i_ = 0; i_ < n; i_++) {                                         \
         HS t_ = 0;                                                                 \
         for (int j_ = 0; j_ < n; j_++)                                             \
            t_ += ((i_ <= j_) ? G[i_ + (size_t)j_ * ldG] : HS_CONJ(G[j_ + (size_t)i_ * ldG])) * (src)[j_]; \
         (dst)[i_] = t_;                                                            \
      }                                                                             \
   } while (0)
   for (;;) {
      if (nOrth >= maxNumOrthos) {
         if (updateR) { *R = 0.0; updateR = 0; }
         if (randomizations >= maxNumRandoms) break;
         pa_larnv_uniform11(iseed, (int64_t)n * SD, (double *)x);
         randomizations++;
         nOrth = 0;
      }
      nOrth++;
      APPLY_G(Bx, x);
      if (nOrth == 1) { s02 = 0; for (int i = 0; i < n; i++) s02 += HS_RE(HS_CONJ(x[i]) * Bx[i]); s0 = sqrt(s02); }
      for (int j = 0; j < nQ; j++) {
         HS t = 0;
         for (int i = 0; i < n; i++) t += HS_CONJ(Q[i + (size_t)j * ldQ]) * Bx[i];
         ov[j] = t;
      }
      for (int j = 0; j < nQ; j++)
         for (int i = 0; i < n; i++) x[i] -= Q[i + (size_t)j * ldQ] * ov[j];
      APPLY_G(Bx, x);
      s12 = 0;
      for (int i = 0; i < n; i++) s12 += HS_RE(HS_CONJ(x[i]) * Bx[i]);
      s1 = sqrt(s12);
      if (!isfinite(s0) || !isfinite(s1) || s1 <= PA_EPS * s0) {
         nOrth = maxNumOrthos;
      } else if (s1 <= tol * s0 || nOrth < maxNumOrthos) {
         s0 = s1; s02 = s12;
      } else {
         if (updateR) *R = s1;
         double inv = 1.0 / s1;
         if (isfinite(inv)) { for (int i = 0; i < n; i++) x[i] *= inv; rc = 0; break; }
         nOrth = maxNumOrthos;
      }
   }
#undef APPLY_G
   (void)s02;
   free(ov);
   free(Bx);
   return rc;
}

/* ---- H(:, k:k+b) = V(:,0:k+b)' W(:,k:k+b)  (reference update_projection.c:80-165) - */
int pa_update_projection(pa_solver *s, int numCols, int blockSize) {
   if (blockSize <= 0) return 0;
   const int mrows = numCols + blockSize;
   hipk_seg seg = {s->V, s->ld, mrows};
   CHK(hipk_panel_dots(s->ctx, s->dt, s->m, &seg, 1, WCOL(s, numCols), s->ld, blockSize, s->d_red, mrows));
   CHK(pa_reduce(s, s->d_red, SD * mrows * blockSize, 0, 0));
   for (int c = 0; c < blockSize; c++)
      for (int i = 0; i < mrows; i++)
         s->H[i + (size_t)(numCols + c) * s->K] = ((const HS *)s->h_red)[i + (size_t)c * mrows];
   return 0;
}

/* ---- Rayleigh-Ritz on the projected matrix (reference solve_projection.c:94-154,
 *      :188-331): eigenpairs of H (upper) ordered by primme->target, then the
 *      running estimates of the spectrum edges. --------------------------------- */
int pa_solve_H_RR(pa_solver *s, const HS *H, int ldH, const HS *VtBV, int ldVtBV,
      HS *hVecs, int ldhVecs, double *hVals, int n, int numConverged) {
   primme_params *p = s->p;
   if (n == 0) return 0;
   if (p->target == primme_largest) {
      HS *Hn = (HS *)calloc((size_t)n * n, sizeof(HS));
      if (!Hn) return PRIMME_MALLOC_FAILURE;
      for (int j = 0; j < n; j++)
         for (int i = 0; i <= j; i++) Hn[i + (size_t)j * n] = -H[i + (size_t)j * ldH];
#if PA_IS_COMPLEX
      int rc = pa_sym_eig_gen(n, Hn, n, VtBV, ldVtBV, hVals, hVecs, ldhVecs);
#else
      int rc = (s->device_rr && !VtBV && n <= 64) ? hipk_sym_eig(s->ctx, n, Hn, n, hVals, hVecs, ldhVecs)
                                                   : pa_sym_eig_gen(n, Hn, n, VtBV, ldVtBV, hVals, hVecs, ldhVecs);
#endif
      free(Hn);
      if (rc) return rc;
      for (int i = 0; i < n; i++) hVals[i] = -hVals[i];
      return 0;
   }
#if !PA_IS_COMPLEX
   if (s->device_rr && !VtBV && n <= 64) CHK(hipk_sym_eig(s->ctx, n, H, ldH, hVals, hVecs, ldhVecs));
   else
#endif
   CHK(pa_sym_eig_gen(n, H, ldH, VtBV, ldVtBV, hVals, hVecs, ldhVecs));
   if (p->target == primme_smallest) return 0;

   /* interior targets: permutation by closeness to the first unlocked shift */
   int *permu = (int *)malloc((size_t)n * sizeof(int));
   if (!permu) return PRIMME_MALLOC_FAILURE;
   const double shift = p->targetShifts[PA_MIN(p->numTargetShifts - 1, numConverged)];
   int idx = 0, i, j;
   if (p->target == primme_closest_geq) {
      for (j = 0; j < n; j++) if (hVals[j] >= shift) break;
      for (i = j; i < n; i++) permu[idx++] = i;
      for (i = 0; i < j; i++) permu[idx++] = i;
   } else if (p->target == primme_closest_leq) {
      for (j = n - 1; j >= 0; j--) if (hVals[j] <= shift) break;
      for (i = j; i >= 0; i--) permu[idx++] = i;
      for (i = n - 1; i > j; i--) permu[idx++] = i;
   } else if (p->target == primme_closest_abs) {
      for (j = 0; j < n; j++) if (hVals[j] >= shift) break;
      i = j - 1;
      while (i >= 0 && j < n) {
         if (fabs(hVals[i] - shift) < fabs(hVals[j] - shift)) permu[idx++] = i--;
         else permu[idx++] = j++;
      }
      if (i < 0) { for (i = j; i < n; i++) permu[idx++] = i; }
      else if (j >= n) { for (j = i; j >= 0; j--) permu[idx++] = j; }
   } else { /* primme_largest_abs */
      j = 0; i = n - 1;
      while (i >= j) {
         if (fabs(hVals[i] - shift) > fabs(hVals[j] - shift)) permu[idx++] = i--;
         else permu[idx++] = j++;
      }
   }
   pa_permute_reals(hVals, 1, n, 1, permu);
   pa_permute_cols(hVecs, n, n, ldhVecs, permu);
   free(permu);
   return 0;
}

int pa_solve_H_harm(pa_solver *s, int k, const HS *G, int ldG);
int pa_solve_H_ref(pa_solver *s, int k, double *hVals_out);

int pa_solve_H(pa_solver *s, int basisSize, int numLocked, int numConverged) {
   primme_params *p = s->p;
   const int off = p->numOrthoConst + numLocked;
   const HS *G = s->VtBV ? s->VtBV + (size_t)off * s->ldVtBV + off : NULL;
   if (s->refined) CHK(pa_solve_H_ref(s, basisSize, s->hVals));
   else if (s->Q) CHK(pa_solve_H_harm(s, basisSize, G, s->ldVtBV));
   else CHK(pa_solve_H_RR(s, s->H, s->K, G, s->ldVtBV, s->hVecs, basisSize, s->hVals, basisSize, numConverged));
   for (int i = 0; i < basisSize; i++) {
      p->stats.estimateMinEVal = PA_MIN(p->stats.estimateMinEVal, s->hVals[i]);
      p->stats.estimateMaxEVal = PA_MAX(p->stats.estimateMaxEVal, s->hVals[i]);
      p->stats.estimateLargestSVal = PA_MAX(p->stats.estimateLargestSVal, fabs(s->hVals[i]));
   }
   s->coef_valid_k = -1;
   return 0;
}

/* upload the current coefficient vectors / Ritz values once per solve_H */
int pa_push_coefficients(pa_solver *s, int basisSize, int ldh) {
   if (s->coef_valid_k == basisSize) return 0;
   for (int j = 0; j < basisSize; j++)
      memcpy((HS *)s->h_coef + (size_t)j * s->K, s->hVecs + (size_t)j * ldh, (size_t)basisSize * sizeof(HS));
   memcpy(s->h_theta, s->hVals, (size_t)basisSize * sizeof(double));
   CHK(hipk_h2d(s->ctx, s->d_coef, s->h_coef, (size_t)s->K * basisSize * sizeof(HS)));
   CHK(hipk_h2d(s->ctx, s->d_theta, s->h_theta, (size_t)basisSize * sizeof(double)));
   s->coef_valid_k = basisSize;
   return 0;
}

/* ---- fused Ritz / residual update with the reference's bookkeeping -------------
 * Wraps hipk_ritz_update: uploads coefficients if stale, splits norm-only
 * residual jobs into read-only pre-passes when more than 16 residual columns
 * are requested, reduces the squared norms and returns their square roots in
 * norms_out[slot] (clamped from below by estimateResidualError when asked, as
 * reference restart.c:1266-1270 / main_iter.c:1686-1690 do). */
int pa_ritz_update(pa_solver *s, int basisSize, const hipk_job *jobs, int njobs, double *norms_out,
      int nslots, int64_t flop_cols) {
   primme_params *p = s->p;
   double t0 = pa_wtime();
   if (njobs <= 0) return 0;
   /* count residual jobs */
   int nres = 0;
   for (int q = 0; q < njobs; q++) if (jobs[q].kind == HIPK_JOB_RES) nres++;
   if (s->B && nres > 0) {
      /* Generalised problem: R = W h - theta B (V h).  The fused kernel forms W h - theta V h; here the residual jobs become
       * X = V h and Y = W h into the scratch panel BT = [X | Y | B X] (in the same launch as the caller's other outputs, so the
       * in-place semantics are unchanged), then B X through the callback, R = Y - theta B X, |R|^2 and |X|^2 in one reduction,
       * and the residuals that are wanted are copied to their destinations.  |x|_2 of the B-normalised Ritz vectors feeds the
       * estimates of |B| and |B^-1| (main_iter.c:1690-1700). */
      if (3 * nres > s->nBT) return PRIMME_FUNCTION_UNAVAILABLE;
      hipk_job *wk = (hipk_job *)malloc((size_t)(njobs + nres) * sizeof(hipk_job));
      double *mth = (double *)malloc((size_t)nres * SD * sizeof(double));
      if (!wk || !mth) { free(wk); free(mth); return PRIMME_MALLOC_FAILURE; }
      char *Xb = s->BT, *Yb = PCOL(s, s->BT, s->ld, nres), *BXb = PCOL(s, s->BT, s->ld, 2 * nres);
      int cnt = 0, r = 0;
      for (int q = 0; q < njobs; q++) {
         if (jobs[q].kind != HIPK_JOB_RES) { wk[cnt++] = jobs[q]; continue; }
         wk[cnt++] = (hipk_job){HIPK_JOB_XV, jobs[q].col, PCOL(s, Xb, s->ld, r), -1};
         wk[cnt++] = (hipk_job){HIPK_JOB_XW, jobs[q].col, PCOL(s, Yb, s->ld, r), -1};
         for (int d = 0; d < SD; d++) mth[r * SD + d] = 0.0;
         mth[r * SD] = -s->h_theta[jobs[q].col];
         r++;
      }
      int rc = hipk_ritz_update(s->ctx, s->dt, s->m, s->V, s->W, s->ld, basisSize, s->d_coef, s->K, s->d_theta, wk, cnt, NULL);
      if (!rc) rc = pa_apply_B(s, Xb, s->ld, BXb, s->ld, nres);
      if (!rc) rc = hipk_axpy_cols(s->ctx, s->dt, s->m, mth, BXb, s->ld, Yb, s->ld, nres);
      if (!rc) rc = hipk_col_norms2(s->ctx, s->dt, s->m, Xb, s->ld, 2 * nres, s->d_red);      /* [|X|^2 | |R|^2]: X and Y are adjacent */
      if (!rc) rc = pa_reduce(s, s->d_red, 2 * nres, 0, 0);
      r = 0;
      for (int q = 0; q < njobs && !rc; q++) {
         if (jobs[q].kind != HIPK_JOB_RES) continue;
         if (jobs[q].slot >= 0 && norms_out) norms_out[jobs[q].slot] = sqrt(s->h_red[nres + r]);
         const double xn = sqrt(s->h_red[r]);
         if (xn > 0.0) {
            p->stats.estimateBNorm = PA_MAX(p->stats.estimateBNorm, 1.0 / xn);
            p->stats.estimateInvBNorm = PA_MAX(p->stats.estimateInvBNorm, xn);
         }
         if (jobs[q].dst) rc = hipk_copy_cols(s->ctx, s->dt, s->m, PCOL(s, Yb, s->ld, r), s->ld, (char *)jobs[q].dst, s->ld, 1);
         r++;
      }
      free(wk); free(mth);
      p->stats.timeDense += pa_wtime() - t0;
      p->stats.flopsDense += (double)s->m * (double)flop_cols * basisSize;
      return rc;
   }
   hipk_job *work = (hipk_job *)malloc((size_t)njobs * sizeof(hipk_job));
   if (!work) return PRIMME_MALLOC_FAILURE;
   double *d_n = s->d_red;
   int slot_base = 0;
   /* The kernel takes at most 16 residual jobs per launch.  With more (block sizes > 16), the
    * residual jobs are run first, 16 at a time: they only READ V and W, so this is the same
    * "all inputs before any output" semantics; the ones that also store the residual go through
    * the scratch panel T and are copied to their destinations after the main launch. */
   int *slot_of = (int *)malloc((size_t)(nslots > 0 ? nslots : 1) * sizeof(int)); /* slot -> position in d_n */
   char **final_dst = (char **)calloc((size_t)njobs + 1, sizeof(char *));
   if (!slot_of || !final_dst) { free(work); free(slot_of); free(final_dst); return PRIMME_MALLOC_FAILURE; }
   for (int i = 0; i < nslots; i++) slot_of[i] = -1;
   int nstaged = 0;
   if (nres > 16) {
      int cnt = 0;
      for (int q = 0; q <= njobs; q++) {
         if (q < njobs && jobs[q].kind == HIPK_JOB_RES) {
            work[cnt] = jobs[q];
            if (jobs[q].dst) {
               if (nstaged >= s->nT) { free(work); free(slot_of); free(final_dst); return PRIMME_FUNCTION_UNAVAILABLE; }
               final_dst[nstaged] = (char *)jobs[q].dst;
               work[cnt].dst = TCOL(s, nstaged);
               nstaged++;
            }
            if (jobs[q].slot >= 0) { slot_of[jobs[q].slot] = slot_base + cnt; work[cnt].slot = cnt; }
            else work[cnt].slot = cnt;   /* the norm is computed anyway; nobody reads it */
            cnt++;
         }
         if (cnt == 16 || (q == njobs && cnt > 0)) {
            int rc = hipk_ritz_update(s->ctx, s->dt, s->m, s->V, s->W, s->ld, basisSize, s->d_coef, s->K, s->d_theta, work, cnt, d_n + slot_base);
            if (rc) { free(work); free(slot_of); free(final_dst); return rc; }
            slot_base += cnt; cnt = 0;
         }
      }
   }
   /* main launch: everything else */
   int cnt = 0, nmain_slots = 0;
   for (int q = 0; q < njobs; q++) {
      if (nres > 16 && jobs[q].kind == HIPK_JOB_RES) continue;
      work[cnt] = jobs[q];
      if (jobs[q].kind == HIPK_JOB_RES) {
         if (jobs[q].slot >= 0) slot_of[jobs[q].slot] = slot_base + nmain_slots;
         work[cnt].slot = nmain_slots++;
      }
      cnt++;
   }
   if (cnt) {
      int rc = hipk_ritz_update(s->ctx, s->dt, s->m, s->V, s->W, s->ld, basisSize, s->d_coef, s->K, s->d_theta, work, cnt, d_n + slot_base);
      if (rc) { free(work); free(slot_of); free(final_dst); return rc; }
   }
   for (int t = 0; t < nstaged; t++) {
      int rc = hipk_copy_cols(s->ctx, s->dt, s->m, TCOL(s, t), s->ld, final_dst[t], s->ld, 1);
      if (rc) { free(work); free(slot_of); free(final_dst); return rc; }
   }
   free(final_dst);
   const int total_slots = slot_base + nmain_slots;
   if (total_slots > 0) {
      int rc = pa_reduce(s, d_n, total_slots, 0, 0);
      if (rc) { free(work); free(slot_of); return rc; }
      for (int i = 0; i < nslots; i++)
         if (slot_of[i] >= 0 && norms_out) norms_out[i] = sqrt(s->h_red[slot_of[i]]);
   } else if (s->phase_timing) {
      hipk_sync(s->ctx);
   }
   free(work);
   free(slot_of);
   p->stats.timeDense += pa_wtime() - t0;
   p->stats.flopsDense += (double)s->m * (double)flop_cols * basisSize;
   return 0;
}
