/* eigs_solver.h — state of one hip_?primme solve (host side). */
#ifndef EIGS_SOLVER_H
#define EIGS_SOLVER_H

#include "eigs_internal.h"

enum { UNCONV = PRIMME_AMD_UNCONVERGED, SKIP_RESTART = PRIMME_AMD_SKIP_UNTIL_RESTART,
       CONV = PRIMME_AMD_CONVERGED, PRACT_CONV = PRIMME_AMD_PRACTICALLY_CONVERGED };

/* run-time cost model of the dynamic method (eigs_dynamic.c) */
typedef struct {
   double t_mv_pr, t_mv, t_pr;        /* operator, preconditioner time per application      */
   double t_qmr, t_qmr_mv_pr;         /* one QMR step without / with the operators          */
   double t_gd_mv_pr, t_gd_mv;        /* one GD+k outer step with / without the correction  */
   double rate_gd, rate_jd;           /* residual reduction per matvec seen so far          */
   double slowdown, mv_per_outer;
   int next_reset;
   double logred_gd, logred_jd, mv_gd, mv_jd;
   int found_gd, found_jd;
   int it0, mv0;                      /* counters at the last observation                   */
   double t0, t_inner, res0;
   double acc_jd, acc_gd, acc_ratio;  /* expected accumulated times, for the recommendation */
} pa_cost_model;

typedef struct pa_solver {
   primme_params *p;
   hipk_ctx *ctx;
   hipk_dtype dt;
   size_t es;              /* bytes per panel element */
   double mach_eps;        /* of the panel type */
   int64_t m, ld;          /* nLocal, ldOPs */
   int K;                  /* maxBasisSize */
   int maxRank;

   /* HBM-resident panels */
   char *V, *W;            /* m x K */
   char *T;                /* scratch, m x nT */
   /* generalised problems A x = lambda B x (round 6): massMatrixMatvec given.  B enters through the callback only — no B V
    * panel is kept: B X is formed where it is needed (the new block in the orthonormalisation, the Ritz vectors whose
    * residual W h - theta B V h is wanted) into the scratch panel BT (nBT columns).  Always the explicit_I path (tracked
    * V'BV, ortho.c:497-803), Rayleigh-Ritz, the Generalized-Davidson family. */
   int B;
   char *BT;
   int nBT;
   /* ... and an inner solver (JDQMR): B evecs of the constraints and locked vectors, the left projector I - (B Q) Q' and the
    * right one (correction.c:862-997); columns [0, nBevecs) are current */
   char *Bevecs;
   int nBevecs;
   /* PRIMME_AMD_JDQMR_REF_SOFT_LOCKING=1: without locking the reference allocates no B evecs and hands evecs over in its place
    * (main_iter.c:334-336, :659-661), so its projectors read Q where B Q is meant and, for block size 1, B x where x is
    * (correction.c:911-917 copies x, then B x, into the same column).  A parity instrument like the indexing knob. */
   int ref_soft_alias;
   /* harmonic extraction: (A - tau I) V = Q R, with Q in HBM, R / Q'V / left vectors on the host */
   char *Q;
   HS *R, *QtV, *hU;       /* (harmonic / refined extraction: real objects only) */
   int refined;            /* refined extraction: singular triplets of R instead of Q'V */
   double *hSVals; HS *hVecsRot;
   int numArbitraryVecs;   /* leading columns of hVecs that are Rayleigh-Ritz vectors of a cluster */
   /* K^-1-weighted (skew) right projector of the correction equation: evecsHat = K^-1 evecs for
    * the stored converged / constraint vectors, M = evecs' evecsHat and its LU factors (host) */
   char *evecsHat;
   HS *Mq, *Mlu;
   int *Mpiv, ldM;
   char *Jw;               /* JDQMR work panels g, d, delta, w, sol: m x 5b (only with inner iterations) */
   int nT;
   char *evecs;            /* caller's device array: constraints | locked | guesses */
   int64_t ldevecs;

   /* small device buffers */
   double *d_red;          /* reduction results / projection coefficients */
   double *d_coef;         /* K x K Ritz coefficient vectors */
   double *d_theta;        /* K Ritz values */
   int red_cap;            /* doubles in d_red / h_red (twice that is allocated: second half = d_fov) */
   size_t coef_cap;        /* doubles in d_coef / h_coef: max(K, numOrthoConst)^2 */
   /* pinned host mirrors */
   double *h_red, *h_coef, *h_theta;

   /* projected problem (host); HS = double, or double complex in the complex objects (eigs_scalar.h) */
   HS *H, *hVecs, *prevhVecs;
   double *hVals, *prevRitzVals;
   HS *VtBV, *fVtBV;       /* explicit_I only */
   int ldVtBV;
   double *blockNorms, *basisNorms;
   int *flags, *map, *iev, *perm, *lockedFlags;
   int numPrevRitzVals;
   int targetShiftIndex;

   int dev_comm;           /* reductions run on the device through RCCL */
   int phase_timing;       /* primme->profile != NULL: sync after phases to time them */
   /* first-pass CGS overlaps produced by the fused residual kernel (block size 1) */
   double *d_fov, *h_fov;
   int fov_valid, fov_k, fov_L;
   int fov_s1_off;         /* where |v|^2 after the first update sits in d_fov / h_fov */
   int fov_projected;      /* the first pass' update + norm were already run speculatively */
   char *fov_col;
   /* speculative tail of the block-size-1 GD iteration: the new basis vector was normalised with
    * the device-resident norm, multiplied by the operator and projected before the host looked
    * at anything; accepted by the orthogonaliser if Daniel's test passes on the first pass */
   int device_rr;          /* PRIMME_AMD_DEVICE_RR: small Rayleigh-Ritz solve by the device Jacobi kernel */
   int spec2_enabled;      /* off with PRIMME_AMD_NO_SPEC2 (measurement knob, read once per solve) */
   int wtr_enabled;        /* projection column from W'r (default; PRIMME_AMD_NO_WTR=1 disables) */
   double *wtq;            /* G = W'Q for the first wtq_rows basis vectors (K x HIPK_WTR_MAX_K, host) */
   int wtq_rows, wtq_L;    /* -1: not valid */
   int spec2_valid, spec2_k;
   int spec_fused;         /* the tail ran through the fused operator launch: the projected, un-normalised
                              vector sits in T(:,0), V(:,k) already holds the normalised one */
   double *spec_hcol;      /* K+1 entries: V(:,0:k+1)' W(:,k) */
   /* Fused restart: the convergence check at a full basis runs through the fused residual
    * kernel; its residual (scratch column T(:,2)) and overlaps with the old basis are kept here, the restart
    * turns them into the overlaps with the restarted basis and the first iteration after it starts from them */
   int fused_restart;      /* on by default; PRIMME_AMD_NO_FUSED_RESTART=1 disables (measurement knob) */
   int rst_valid, rst_k, rst_L;
   double rst_theta;
   double *rst_y;          /* K: coefficient vector of the checked candidate */
   double *rst_ov;         /* [V'r | Q'r | r'r | W'r | W(:,k-1)'Q] of that pass */
   int rst_ready, rst_rs;  /* the restart used the stash: rst_c = [V_new'r (rs) | Q'r (L) | r'r | W_new'r (rs)] */
   double *rst_c;
   int fov_carry;          /* the overlaps in d_fov / h_fov come from a restart: survive the next candidate check */
   /* Speculative restart (second half): the check at the full basis IS the
    * restart pass, run out of place into the alternate panels V2 / W2 with the coefficient block a dry run of the
    * restart (plan_only) predicts for "candidate not converged"; the real restart adopts the result (swaps the
    * panels) if it arrives at the same coefficients bit for bit, and otherwise runs its own pass on V, W */
   char *V2, *W2;
   double *d_coef2, *d_theta2, *h_coef2, *h_theta2;   /* the predicted block, device + pinned staging */
   int plan_allowed;       /* set by the main loop around the full-basis check (no guesses pending ...) */
   int plan_only;          /* pa_restart runs as a dry run: stops where the pass would start */
   int pl_k, pl_rs, pl_L;  /* the prediction */
   int pl_cand, pl_nc;     /* coefficient column of the candidate; converged pairs copied out by the pass (soft locking) */
   int pl_launched;        /* the pass ran with it: rst_c holds the overlaps with the new basis, rst_grow W(:,k-1)'Q */
   double *rst_grow;
   /* The NEXT block-size-1 iteration enqueued before the host has seen this one (eigs_conv.c:
    * pa_prelaunch_next).  The pre-enqueued pass writes its overlaps into the OTHER overlap buffer (d_fov_alt / h_fov_alt:
    * the host is still reading this iteration's), its projected vector into the other scratch column, its t'At into the
    * alpha slot of that buffer; adopting it swaps the buffers. */
   int pre_enabled;        /* off with PRIMME_AMD_NO_PRELAUNCH=1 (A/B knob, read once per solve) */
   double *d_fov_alt, *h_fov_alt;
   double *d_hnext, *h_hnext;     /* hipk_rr_arrow's output: coefficient vector [0..k], Ritz value [32], status [33] */
   int pre_valid, pre_k, pre_L, pre_cand, pre_nfov, pre_tcol;
   int pre_quiet_fin;      /* the residual pass of a pre-enqueued iteration publishes no flag of its own (PRIMME_AMD_LOUD_FIN=1: it does) */
   int pre_tail_deferred;  /* the pre-enqueued iteration left its tail to hipk_tail_finish (whoever adopts it launches that) */
   unsigned long long pre_seq_rr, pre_seq_end;   /* flags to wait for: the small kernel's, the pass' last reduction's */
   int spec_tcol;          /* scratch column holding the projected, un-normalised vector of the tail that is pending */
   long pre_launched, pre_adopted;
   int parallel;           /* reductions cross ranks (numProcs > 1 and a globalSumReal installed) */
   int fuse_gd;            /* GD without preconditioner/Olsen: residual written straight into V */
   int coef_valid_k;       /* d_coef/d_theta currently hold hVecs/hVals of this size, or -1 */
   double startTime;
} pa_solver;

#define PCOL(s, base, ldc, j) ((char *)(base) + (size_t)(j) * (size_t)(ldc) * (s)->es)
#define VCOL(s, j) PCOL(s, (s)->V, (s)->ld, j)
#define WCOL(s, j) PCOL(s, (s)->W, (s)->ld, j)
#define TCOL(s, j) PCOL(s, (s)->T, (s)->ld, j)
#define ECOL(s, j) PCOL(s, (s)->evecs, (s)->ldevecs, j)

/* error propagation; with PRIMME_AMD_TRACE_ERRORS set the failing call chain is printed, the
 * counterpart of the reference's CHKERR trace (src/include/common.h:437-470) */
/* an iteration enqueued ahead of the host is not going to be used: forget it and the tail it left unfinished (eigs_conv.c) */
static inline void pa_pre_discard(pa_solver *s) {
   if (s->pre_valid && s->pre_tail_deferred) hipk_tail_abandon(s->ctx);
   s->pre_valid = 0; s->pre_tail_deferred = 0;
}
int pa_apply_B(pa_solver *s, char *X, int64_t ldX, char *Y, int64_t ldY, int nc);
int pa_trace_errors(void);
#define CHK(call) do { int rc_ = (call); if (rc_) { \
      if (pa_trace_errors()) fprintf(stderr, "primme_amd: error %d at %s:%d: %s\n", rc_, __FILE__, __LINE__, #call); \
      return rc_ < 0 ? rc_ : PRIMME_UNEXPECTED_FAILURE; } } while (0)

double pa_wtime(void);

#endif
