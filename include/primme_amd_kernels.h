/* primme_amd_kernels.h — C-ABI of the device layer ("hip_wrapper") under the solver.
 *
 * This is boundary B2 of SURVEY.md §8(b): the numerical backend the reference
 * resolves at compile time to src/linalg/blaslapack.c (CPU), cublas_wrapper.c
 * (cuBLAS/hipBLAS) or magma_wrapper.c.  Each entry point below names the
 * reference backend routine(s) or fused solver step it replaces (file:line in
 * /root/reference).  Plain pointers and sizes only; no torch / C++ types.
 *
 * Conventions
 *   - "panel" = tall-skinny column-major matrix with m local rows (10^5..10^7)
 *     and a few (<= 64+64) columns, resident in HBM for the whole solve.
 *   - all launches go to the stream owned by the hipk_ctx; nothing synchronises
 *     unless the name says so (hipk_sync, *_to_host).
 *   - small operands (coefficients, Ritz values, reduction results) live in
 *     DEVICE memory too, so that chains like dots -> all-reduce -> update run
 *     without a host round trip (the reference's GPU backend syncs on every
 *     Num_set_matrix/Num_get_matrix, cublas_wrapper.c:335-395).
 *   - dtype: element type of the panels.  Reductions accumulate and are returned
 *     in double (real) / double complex (complex), laid out as doubles.
 *   - complex panels (HIPK_C64 / HIPK_C32, csrc/hipk_complex.hip; the reference's SCALAR = complex instantiation,
 *     src/include/template_types.h:91): elements are interleaved (re, im) pairs and leading dimensions count complex
 *     elements.  Every "accumulator scalar" argument — inner products (the left operand is conjugated: col_j^H X,
 *     x^H y), projection coefficients, Ritz coefficient vectors, the M of hipk_panel_project_mul, the factors of
 *     hipk_axpy_cols / hipk_xpay_cols — is a (re, im) pair of doubles with leading dimensions in pairs; Ritz values,
 *     shifts, squared norms and the factors of hipk_scale_cols stay real.  Entry points that exist only for the
 *     fused real block-size-1 iteration (hipk_ritz_residual_overlaps, hipk_ritz_update_overlaps,
 *     hipk_csr_matvec_scaled, the early-rho step of the QMR iteration) and the stencil operator return -44 for
 *     complex dtypes.  The QMR recurrences of a complex JDQMR solve run through the REAL kernels on the 2m-real
 *     view of the m complex rows (their coefficients are real; csrc/eigs_jd.c), the harmonic and refined
 *     extractions natively on complex panels (csrc/eigs_harm.c).
 *   - return 0 on success, PRIMME-style negative code otherwise.
 */
#ifndef PRIMME_AMD_KERNELS_H
#define PRIMME_AMD_KERNELS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hipk_ctx hipk_ctx;

typedef enum { HIPK_F64 = 0, HIPK_F32 = 1, HIPK_C64 = 2 /* double complex */,
               HIPK_C32 = 3 /* float complex */ } hipk_dtype;

/* A column range of one panel; kernels take up to HIPK_MAX_SEGS of them so that
 * [locked evecs | V] is streamed in one launch (reference ortho.c:236-246 issues
 * one gemv per array). */
#define HIPK_MAX_SEGS 3
typedef struct { const void *base; int64_t ld; int ncols; } hipk_seg;

/* ---- context, memory, copies ------------------------------------------------
 * replaces Num_malloc/free_Sprimme (cublas_wrapper.c:187-233), Num_set_matrix /
 * Num_get_matrix (:335-395), Num_copy_matrix (:739), Num_zero_matrix (:768). */
int  hipk_ctx_create(hipk_ctx **ctx, void *hip_stream_or_null);
int  hipk_ctx_destroy(hipk_ctx *ctx);
void *hipk_ctx_stream(hipk_ctx *ctx);                 /* the hipStream_t            */
int  hipk_malloc(hipk_ctx *ctx, size_t bytes, void **dptr);
int  hipk_free(hipk_ctx *ctx, void *dptr);
int  hipk_host_alloc(hipk_ctx *ctx, size_t bytes, void **hptr); /* pinned          */
int  hipk_host_free(hipk_ctx *ctx, void *hptr);
/* Host <-> device copies on the context's stream.  STREAM-ORDERED (asynchronous) when the host side is pinned memory:
 * a hipk_host_alloc buffer (<= 1 MB of it: a copy kernel through the mapped address, no runtime copy engine) or pinned
 * memory of the caller's (hipHostMalloc / hipHostRegister, recognised through hipPointerGetAttributes).  For PAGEABLE host
 * memory (malloc, numpy, stack) the copy goes in chunks through the context's pinned staging buffer (grows on demand up
 * to 32 MB, kept for the life of the context) and is COMPLETE ON RETURN: the stream is drained per chunk. */
int  hipk_h2d(hipk_ctx *ctx, void *dst, const void *src, size_t bytes);
int  hipk_d2h(hipk_ctx *ctx, void *dst, const void *src, size_t bytes);
int  hipk_d2d(hipk_ctx *ctx, void *dst, const void *src, size_t bytes);   /* async  */
int  hipk_memset0(hipk_ctx *ctx, void *dst, size_t bytes);
int  hipk_sync(hipk_ctx *ctx);
/* wait for the results of the LAST mirrored reduction enqueued on the context (see
 * hipk_ctx_set_mirror) without a runtime call: the reduction's second stage publishes a completion
 * flag in pinned memory and the host spins on it.  Only valid when that reduction was the last thing
 * enqueued; falls back to hipk_sync otherwise. */
int  hipk_wait_results(hipk_ctx *ctx);
/* copy `count` results that some other producer (an RCCL all-reduce) left at `dev` (inside the mirrored range) into the
 * pinned mirror and publish the completion flag: hipk_wait_results instead of a stream synchronisation; returns 1 when
 * the buffer is not mirrored (the caller then copies and synchronises) */
int  hipk_publish_results(hipk_ctx *ctx, const double *dev, int count);
/* zero-copy results: every reduction whose output lies in [dev_base, dev_base+count) is also
 * written by the kernel into the pinned host array (same offsets); the host then needs only
 * hipk_sync, no device->host copy (the reference GPU backend does a blocking hipMemcpy per
 * Num_*_ddh call, cublas_wrapper.c:479-499).  Pass NULLs to switch off. */
int  hipk_ctx_set_mirror(hipk_ctx *ctx, double *dev_base, double *pinned_host_base, size_t count);
int  hipk_is_device_ptr(const void *p);  /* Num_check_pointer, cublas_wrapper.c:162 */
/* x(0:n) (DEVICE, n REAL numbers of the panel's precision: a complex element takes two) = the next n numbers of
 * LAPACK's xLARNV(idist = 2) stream in (-1, 1), generated on the device by jumping ahead in the 48-bit congruential
 * sequence; iseed (four base-4096 digits, host) is advanced as xLARNV advances it.  Replaces Num_larnv
 * (cublas_wrapper.c:707-736: host generation + upload), same numbers bit for bit. */
int  hipk_larnv_uniform11(hipk_ctx *ctx, hipk_dtype dt, int64_t iseed[4], int64_t n, void *x);
/* events around a region on the ctx stream; ms returned by hipk_timer_stop (syncs) */
int  hipk_timer_start(hipk_ctx *ctx);
int  hipk_timer_stop(hipk_ctx *ctx, float *ms);

/* ---- TN panel: inner products ------------------------------------------------
 * out[j + c*ldout] = sum_i conj(col_j(i)) * X(i,c),  j over all segment columns.
 * `out` is a DEVICE array of accumulator scalars (double / double complex).
 * If upper_from >= 0 only entries with j <= upper_from + c are guaranteed
 * (Hermitian projection: H(:,k+c), reference update_projection.c:99-122).
 * Replaces Num_gemm_ddh (cublas_wrapper.c:479-499, blaslapack.c:610-661),
 * Num_gemv_ddh (:566), Num_dot (:647), Num_compute_gramm(_ddh) (:898-987). */
int hipk_panel_dots(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const hipk_seg *segs, int nseg,
      const void *X, int64_t ldX, int nx, double *out_dev, int ldout);

/* ---- NN panel: X(:,c) -= [segs] * coef(:,c), then nrm2[c] = ||X(:,c)||^2 -------
 * coef: DEVICE array of accumulator scalars, ldcoef rows stride.  nrm2 (DEVICE,
 * nx doubles) may be NULL.  One fused pass over the basis: the CGS update
 * (reference ortho.c:262-291: two Num_gemv_dhd + Num_dot) and the projector of
 * ortho_single_iteration (ortho.c:894-925). */
int hipk_panel_project(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const hipk_seg *segs, int nseg,
      const double *coef_dev, int ldcoef, void *X, int64_t ldX, int nx, double *nrm2_dev);

/* out-of-place form: Xout(:,c) = X(:,c) - [segs] * coef(:,c) (X is only read; Xout may equal X).
 * Lets the next launch consume the projected vector from a scratch column while it rebuilds the
 * basis column itself (hipk_csr_matvec_scaled below). */
int hipk_panel_project_to(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const hipk_seg *segs, int nseg,
      const double *coef_dev, int ldcoef, const void *X, int64_t ldX, void *Xout, int64_t ldXout, int nx,
      double *nrm2_dev);

/* X <- (X - [segs] * coef) * M with M an nx x nx matrix (DEVICE, leading dimension nx), nx <= 8, in one
 * pass over X and the basis: the whole device step of a CholQR / SVQB sweep (reference
 * Num_ortho_kernel, ortho.c:963-1072).  Returns 1 if the shape is not covered (nx > 8): the caller
 * then uses hipk_panel_project + hipk_ritz_update. */
int hipk_panel_project_mul(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const hipk_seg *segs, int nseg,
      const double *coef_dev, int ldcoef, const double *M_dev, void *X, int64_t ldX, int nx);

/* ---- fused Ritz / residual / restart update ----------------------------------
 * The multi-output panel op of reference auxiliary_eigs_normal.c:155-388
 * (Num_update_VWXR) and restart.c:1233-1294.  V, W: m x k panels (ld ldVW).
 * h: DEVICE k x nh coefficients (accumulator scalars, leading dim ldh);
 * theta: DEVICE nh Ritz values (doubles).
 * jobs[]: what to produce, executed row-wise so destinations may alias V / W
 * columns (in-place restart) — all inputs of a row are read before any output
 * of that row is written.
 *   HIPK_JOB_XV  : dst = V*h(:,col)
 *   HIPK_JOB_XW  : dst = W*h(:,col)
 *   HIPK_JOB_RES : dst = W*h(:,col) - theta[col]*V*h(:,col)   (dst may be NULL)
 *                  and nrm2_dev[slot] = ||that||^2  when slot >= 0
 */
typedef enum { HIPK_JOB_XV = 0, HIPK_JOB_XW = 1, HIPK_JOB_RES = 2 } hipk_job_kind;
typedef struct { int kind; int col; void *dst; int slot; } hipk_job;
#define HIPK_MAX_JOBS 160
int hipk_ritz_update(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const void *V, const void *W,
      int64_t ldVW, int k, const double *h_dev, int ldh, const double *theta_dev,
      const hipk_job *jobs, int njobs, double *nrm2_dev);

/* hipk_ritz_update with exactly one residual job (the next candidate) that also returns, computed in the
 * same pass, ov_dev = [ (V h)' r | Q' r | r' r | (W h)' r | W(:,k-1)' Q ] for the first `nbasis` XV and XW jobs
 * (the restarted basis): the inner products the iteration after a restart starts from.  k <= 32,
 * nbasis <= 16, L <= 32.  Replaces Num_update_VWXR (reference restart.c:1233-1294) followed by the
 * Num_gemv_ddh of ortho.c:236-246 and the pass of update_projection.c:99-122 on the restarted basis. */
int hipk_ritz_update_overlaps(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const void *V, const void *W,
      int64_t ldVW, int k, const double *h_dev, int ldh, const double *theta_dev, const hipk_job *jobs, int njobs,
      double *nrm2_dev, int nbasis, const void *Q, int64_t ldQ, int L, double *ov_dev);

/* ---- fused Ritz residual + first Gram-Schmidt pass (block size 1) -------------------
 * dst = W*h - theta*V*h  and  out_dev[0..k+L] = [ V' dst | Q' dst | dst' dst ]  in ONE pass
 * over V, W and Q (hcol_host: k coefficients on the HOST, passed in the kernel arguments; theta by value).  Replaces
 * Num_update_VWXR (auxiliary_eigs_normal.c:155-388) + the Num_gemv_ddh/Num_dot of the first
 * CGS pass (ortho.c:229-249) when the residual itself is the new basis vector (GD without
 * preconditioner).  k <= 32, L <= 32.
 * want_wtr != 0 (k, L <= HIPK_WTR_MAX_K): out_dev[k+L+1 .. 2k+L+1) = W' dst and
 * out_dev[2k+L+1 .. 2k+2L+1) = W(:,k-1)' Q as well, from the W and Q columns the pass holds in
 * registers anyway.  With A symmetric and W = A V this is V' A dst, from
 * which the host forms the new column of the projected matrix without another pass over V
 * (update_projection.c:99-122 reads V again for it): see eigs_conv.c. */
#define HIPK_WTR_MAX_K 32
int hipk_ritz_residual_overlaps(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const void *V,
      const void *W, int64_t ldVW, int k, const double *hcol_host, double theta, void *dst,
      const void *Q, int64_t ldQ, int L, int want_wtr, double *out_dev);

/* ---- the NEXT block-size-1 iteration enqueued before the host has seen this one ----
 * Between the last reduction of an outer iteration and the first kernel of the next the device used to wait for the host:
 * completion flag over PCIe, the small eigenproblem, a launch — 20-26 us per iteration.  With these two entry points the host
 * enqueues the next iteration's launches right behind the current ones; the ONE thing they need from the host, the Ritz pair
 * the next residual is formed with, is computed by a one-wave kernel from what is already in HBM:
 *   hipk_rr_arrow: the host passes the Rayleigh-Ritz decomposition it holds for the CURRENT basis (k Ritz values theta,
 *     coefficient vectors Y, the kept rows of G = W'Q) by value; the kernel reads this iteration's reductions
 *     fov = [V'r (k) | Q'r (L) | r'r | W'r (k) | W(:,k-1)'Q (L)], |t|^2 at fov[nfov] and t'At at alpha_dev[0], forms the new
 *     column of the projected matrix in the Ritz basis, z = Y'(W'r - H V'r - G Q'r)/|t| (H = Y diag(theta) Y'), and solves the
 *     (k+1) x (k+1) ARROWHEAD eigenproblem [diag(theta) z; z' alpha] for its `cand`-th eigenpair (ascending; descending with
 *     largest != 0) through the secular equation (one pole-shifted root, safeguarded Newton on 16 lanes) — O(k) per step
 *     instead of the O(k^3) dense solve, which the host still runs, off the critical path, for its own bookkeeping.
 *     out_dev[0 .. k] = the coefficient vector in the basis [V t], out_dev[32] = the Ritz value, out_dev[33] = status
 *     (0 ok; anything else: no valid pair, the residual launch that reads it leaves at once).  k <= 16, L <= 10.
 *   hipk_ritz_residual_overlaps_dev: hipk_ritz_residual_overlaps with the coefficient vector and the Ritz value read
 *     from DEVICE memory (hth_dev = out_dev of hipk_rr_arrow).
 * The host adopts the result when its own Rayleigh-Ritz solve arrives at the same pair (to rounding) and throws it away
 * otherwise; hipk_seq_issued / hipk_wait_seq let it wait for one particular flagged reduction while later ones are queued. */
typedef struct hipk_rr_in {
   int k, L, cand, largest, grow_row;      /* grow_row != 0: row k-1 of G is this pass' W(:,k-1)'Q (fov[2k+L+1 ..)), not G[k-1][:] */
   int pad;
   double theta[16];                       /* Ritz values of the current basis, in the host's order */
   double Y[16 * 16];                      /* their coefficient vectors, column i = Y[i*k .. i*k + k) */
   double G[16 * 10];                      /* G[j + l*k] = W(:,j)' Q(:,l) */
} hipk_rr_in;
int hipk_rr_arrow(hipk_ctx *ctx, const hipk_rr_in *in, const double *fov_dev, int nfov, const double *alpha_dev, double *out_dev);
int hipk_ritz_residual_overlaps_dev(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const void *V, const void *W, int64_t ldVW, int k,
      const double *hth_dev, void *dst, const void *Q, int64_t ldQ, int L, int want_wtr, double *out_dev);
/* sequence number of the last flagged reduction enqueued on the context, and a wait for a particular one */
unsigned long long hipk_seq_issued(hipk_ctx *ctx);
int hipk_wait_seq(hipk_ctx *ctx, unsigned long long seq);

/* ---- the tail of a block-size-1 iteration without its own second-stage launches (round 6) ----
 * The tail is: Gram-Schmidt update + |t|^2 (hipk_panel_project_to, ortho.c:262-291), then scale + A t + t'At
 * (hipk_csr_matvec_scaled; matrixMatvec + update_projection.c:99-122).  Each of the two used to end in a second-stage launch of
 * its own (globalSumReal's place in the reference, ortho.c:290 and update_projection.c:136), and the pre-enqueued next
 * iteration started with the one-wave launch hipk_rr_arrow: ~16 us and four kernel boundaries per outer iteration that stream
 * nothing.  hipk_tail_defer(ctx, want) arms, for the NEXT hipk_panel_project_to with one column and the NEXT
 * hipk_csr_matvec_scaled on this context:
 *   HIPK_TAIL_NORM  the partial sums of |t|^2 stay in HBM; every workgroup of the operator launch adds them itself (a few KB
 *                   out of L2, one fixed order) before it scales — only honoured by the row-pattern form of the operator,
 *                   whose grid is the resident set (a tile kernel with tens of thousands of workgroups would re-read them
 *                   more often than the launch it saves is worth); not for row-partitioned runs (|t|^2 must be global);
 *   HIPK_TAIL_DOT   the second stage of t'At is left to hipk_tail_finish.
 * It returns the flags it accepted (0: the sequence of separate launches, e.g. HIPK_NO_TAIL_DEFER=1 or the CPU checker).
 * hipk_tail_finish(ctx, in, fov_dev, nfov, alpha_dev, hnext_out) is then ONE small launch: both sums in the order the operator
 * launch used, into HBM and the pinned mirror (+ the exchange with the other ranks when hipk_xreduce_arm was called before it),
 * and — in != NULL — hipk_rr_arrow's step for the next iteration in the same launch (same arithmetic, same outputs), the
 * completion flag last.  The caller may enqueue it LATER than the tail itself (eigs_conv.c does: at the point where it knows
 * the Rayleigh-Ritz decomposition the next iteration's step needs) as long as nothing else used the context's reduction
 * scratch in between.  With nothing deferred it finishes what is pending the separate way and runs hipk_rr_arrow for in != NULL.
 * (The solver arms a deferral on ONE rank only: with the rows over several ranks a deferred second stage also defers its
 * exchange to a launch that comes late, measured 10 % slower per iteration than the separate launches — DESIGN.md section 4g.)
 * hipk_tail_abandon forgets an armed / pending tail (a pre-enqueued iteration that is thrown away). */
#define HIPK_TAIL_NORM 1
#define HIPK_TAIL_DOT 2
int hipk_tail_defer(hipk_ctx *ctx, int want);
int hipk_tail_finish(hipk_ctx *ctx, const hipk_rr_in *in, const double *fov_dev, int nfov, const double *alpha_dev, double *hnext_out);
void hipk_tail_abandon(hipk_ctx *ctx);
int hipk_tail_pending(hipk_ctx *ctx);
/* the NEXT mirrored reduction on the context publishes no completion flag (one-shot): its second stage then has no
 * system-scope fence and no ticket.  For a reduction whose results the host reads only after a LATER flagged launch
 * (the overlaps of a pre-enqueued residual pass: the host waits for the flag of hipk_tail_finish behind it). */
void hipk_skip_next_flag(hipk_ctx *ctx);

/* ---- column utilities ---------------------------------------------------------
 * Num_scal (cublas_wrapper.c:678), Num_axpy (:616), Num_copy_matrix (:739),
 * permute_vecs / Num_compact_vecs on device columns (auxiliary.c:716, :897). */
int hipk_scale_cols(hipk_ctx *ctx, hipk_dtype dt, int64_t m, void *X, int64_t ldX, int nx,
      const double *alpha_host /* nx real scale factors */);
/* X(:,c) *= 1/sqrt(norm2_dev[c]) with the squared norms still in HBM: normalisation without
 * a host round trip (the speculative tail of the block-size-1 GD iteration) */
int hipk_scale_cols_rsqrt_dev(hipk_ctx *ctx, hipk_dtype dt, int64_t m, void *X, int64_t ldX, int nx,
      const double *norm2_dev);
int hipk_axpy_cols(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const double *alpha_host,
      const void *X, int64_t ldX, void *Y, int64_t ldY, int nx);
/* Y(:,c) = alpha[c] Y(:,c) + X(:,c) */
int hipk_xpay_cols(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const double *alpha_host,
      const void *X, int64_t ldX, void *Y, int64_t ldY, int nx);
int hipk_copy_cols(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const void *X, int64_t ldX,
      void *Y, int64_t ldY, int nx);
/* Y = i * X for columns of npairs (re, im) pairs stored as 2*npairs reals: (re, im) -> (-im, re).
 * The one kernel the real-equivalent treatment of Hermitian problems needs beyond the real
 * panel kernels (eigs_complex.c); dt is the REAL type of the parts.  X and Y distinct. */
int hipk_pair_rotate(hipk_ctx *ctx, hipk_dtype dt, int64_t npairs, const void *X, int64_t ldX,
      void *Y, int64_t ldY, int nx);
/* Y(:,i) = X(:,perm[i]) for i < n, X and Y distinct panels */
int hipk_gather_cols(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const void *X, int64_t ldX,
      const int *perm_host, int n, void *Y, int64_t ldY);
/* out_dev[c] = ||X(:,c)||^2 */
int hipk_col_norms2(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const void *X, int64_t ldX,
      int nx, double *out_dev);
/* residual columns r_c = w_c - theta_host[c]*x_c (in place into W) and their
 * squared norms (verify_norms, reference main_iter.c:1864-1879). */
int hipk_residual_cols(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const void *X, int64_t ldX,
      void *Wr, int64_t ldW, int nx, const double *theta_host, double *nrm2_dev);

/* ---- block QMR recurrences (JDQMR inner solver, reference src/eigs/inner_solve.c) ----------
 * out_dev[c] = X(:,c)' Y(:,c)                      Num_dist_dots_real (auxiliary_eigs.c:695-706) */
int hipk_pair_dots(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const void *X, int64_t ldX,
      const void *Y, int64_t ldY, int nx, double *out_dev);
/* Y(:,c) += alpha[c] X(:,c) (stored), then out_dev[c] = Z(:,c)' Y(:,c)  (Z == NULL: |Y(:,c)|^2), in
 * one pass: the axpy + dot pairs of the QMR step (w -= sigma d, x'w; w -= (x'w) x, d'w;
 * g -= alpha w, g'g  — inner_solve.c:853-880, :317-333, :371-377) with the arithmetic of the
 * separate calls (element-wise fma, same reduction tree as hipk_pair_dots) */
int hipk_axpy_dot(hipk_ctx *ctx, hipk_dtype dt, int64_t m, int nx, const double *alpha_host,
      const void *X, int64_t ldX, void *Y, int64_t ldY, const void *Z, int64_t ldZ, double *out_dev);
/* delta = gamma.*delta + eta.*d; sol += delta; dotsol_dev[c] = |sol(:,c)|^2 in one pass
 * (the host path of the reference fuses the same three steps, inner_solve.c:384-397) */
int hipk_qmr_update(hipk_ctx *ctx, hipk_dtype dt, int64_t m, int nx, const double *gamma_host,
      const double *eta_host, const void *D, int64_t ldD, void *Delta, int64_t ldDelta, void *Sol,
      int64_t ldSol, double *dotsol_dev);

/* the same step together with the next Jacobi application: w = g ./ (diag - shift[c]) (|denominator| kept
 * above min_denominator with its sign) and out_dev[nx + c] = g(:,c)' w(:,c), out_dev[c] = |sol(:,c)|^2:
 * one pass instead of three (inner_solve.c:384-397 and :619-634) */
int hipk_qmr_update_jacobi(hipk_ctx *ctx, hipk_dtype dt, int64_t m, int nx, const double *gamma_host,
      const double *eta_host, const void *D, int64_t ldD, void *Delta, int64_t ldDelta, void *Sol, int64_t ldSol,
      const void *G, int64_t ldG, const void *diag, const double *shift_host, double min_denominator, void *W,
      int64_t ldW, double *out_dev);

/* out_dev[c] = x_c' w_c, out_dev[nx + c] = v_c' w_c, out_dev[2 nx + c] = v_c' x_c in one pass: with them
 * sigma = v'(I - x x')w = v'w - (x'w)(v'x) is known before the projected w is formed */
int hipk_triple_dots(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const void *X, int64_t ldX, const void *V, int64_t ldV,
      const void *W, int64_t ldW, int nx, double *out_dev);
/* W <- W - [segs] coef (coefficients in HBM, column c at coef_dev + c*ldcoef), then out_dev = [x'w | v'w | v'x] (3 nx) for
 * the updated W: hipk_panel_project + hipk_triple_dots in one pass with the arithmetic and summation order of that pair
 * (reference apply_projected_matrix, inner_solve.c:853-880: Num_gemm_dhd + three Num_dist_dots).  Returns 1 when the shape
 * is not covered (complex, nx > 8, no columns): the caller runs the two launches. */
int hipk_project_triple_dots(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const hipk_seg *segs, int nseg, const double *coef_dev,
      int ldcoef, void *W, int64_t ldW, int nx, const void *X, int64_t ldX, const void *V, int64_t ldV, double *out_dev);
/* g_c -= alpha_c (w_c - xr_c x_c), out_dev[c] = |g_c|^2: projection of w against x and the residual update of the
 * QMR step in one pass, the projected w is never stored (inner_solve.c:853-880 followed by :371-377) */
int hipk_axpy_proj_dot(hipk_ctx *ctx, hipk_dtype dt, int64_t m, int nx, const double *alpha_host, const double *xr_host,
      const void *W, int64_t ldW, const void *X, int64_t ldX, void *G, int64_t ldG, double *out_dev);

/* the same with the next inner product of the Jacobi-preconditioned QMR taken on the updated g:
 * out_dev[c] = |g_c|^2, out_dev[nx + c] = g_c' (g_c ./ (diag - shift[c])) — rho of the next step one pass early */
int hipk_axpy_proj_dot_jacobi(hipk_ctx *ctx, hipk_dtype dt, int64_t m, int nx, const double *alpha_host, const double *xr_host,
      const void *W, int64_t ldW, const void *X, int64_t ldX, void *G, int64_t ldG, const void *diag, const double *shift_host,
      double min_denominator, double *out_dev);
/* delta = gamma delta + eta d; sol += delta; dotsol_dev[c] = |sol(:,c)|^2; d = g ./ (diag - shift[c]) + beta d in place:
 * the QMR step of hipk_qmr_update_jacobi and the direction update that followed it (w += beta d, d <-> w) in one pass */
int hipk_qmr_update_dir(hipk_ctx *ctx, hipk_dtype dt, int64_t m, int nx, const double *gamma_host, const double *eta_host,
      const double *beta_host, void *D, int64_t ldD, void *Delta, int64_t ldDelta, void *Sol, int64_t ldSol, const void *G,
      int64_t ldG, const void *diag, const double *shift_host, double min_denominator, double *dotsol_dev);

/* The block-QMR step with ONE host synchronisation (csrc/eigs_jd.c): the two launches above with the step's scalar recurrences
 * (inner_solve.c:376-409: alpha = rho_prev / sigma, Theta, c, gamma, eta, beta) evaluated ON THE DEVICE, in the prologue of the
 * launch that applies them, from reduction results still in HBM — tri_dev = [x'w | v'w | v'x] (hipk_triple_dots), ggr_dev =
 * [g'g | g'K^-1 g] (the first of the two) — and the previous step's rho, tau, Theta passed by value.  The host evaluates the same
 * expressions on the mirrored results after its one wait (same roundings: no contraction on either side).  A column whose alpha
 * is unusable (sigma = 0 or not finite, |alpha| outside [eps, 1/eps]) gets alpha = 0 and is left alone by the second launch,
 * as the host drops it from the block.  nx <= 8. */
int hipk_axpy_proj_dot_jacobi_dev(hipk_ctx *ctx, hipk_dtype dt, int64_t m, int nx, const double *tri_dev, const double *rho_prev_host,
      double mach_eps, const void *W, int64_t ldW, const void *X, int64_t ldX, void *G, int64_t ldG, const void *diag,
      const double *shift_host, double min_denominator, double *out_dev);
int hipk_qmr_update_dir_dev(hipk_ctx *ctx, hipk_dtype dt, int64_t m, int nx, const double *tri_dev, const double *ggr_dev,
      const double *rho_prev_host, const double *tau_prev_host, const double *theta_prev_host, double mach_eps, void *D, int64_t ldD,
      void *Delta, int64_t ldDelta, void *Sol, int64_t ldSol, const void *G, int64_t ldG, const void *diag, const double *shift_host,
      double min_denominator, double *dotsol_dev);

/* ---- sparse operator: the user matvec ------------------------------------------
 * Replaces the hipsparseSpMM-based callback of examples/ex_eigs_dhipblas.c:239-264
 * and the SPARSKIT amux of tests/COMMON/mat.c:64-90.
 * CSR, 0-based int32 indices, row-partitioned: this rank owns rows [row0,row0+m).
 * Column indices are GLOBAL; columns outside the owned range are served from the
 * halo set up by primme_amd_comm (single rank: none). */
typedef struct hipk_csr hipk_csr;
int hipk_csr_create(hipk_ctx *ctx, hipk_dtype dt, int64_t nrows_local, int64_t ncols_global,
      int64_t row0, const int32_t *rowptr_host, const int32_t *colind_host,
      const void *values_host, hipk_csr **A);
int hipk_csr_destroy(hipk_csr *A);
/* launches on `hip_stream` (a hipStream_t passed by value as void*), or on the stream of
 * the creating context when NULL */
int hipk_csr_matvec(hipk_csr *A, void *hip_stream, const void *x, int64_t ldx, void *y, int64_t ldy,
      int ncols);
/* y = A (a x), xout = a x (xout != x), dot_dev[0] = xout' y with a = 1/sqrt(norm2_dev[0]) read from HBM
 * (norm2_dev == NULL: a = 1):
 * normalisation (Num_scal, cublas_wrapper.c:678), operator and the inner product t'At in one launch. */
int hipk_csr_matvec_scaled(hipk_csr *A, hipk_ctx *ctx /* the caller's: stream, scratch, result mirror */,
      const void *x, const double *norm2_dev, void *xout, void *y, double *dot_dev);
int hipk_csr_kind(const hipk_csr *A);    /* 0 CSR, 1 stencil */
/* y = A x - shift_host[c] x(:,c) in one launch; returns 1 when the operator is not covered (stencil, halo) */
int hipk_csr_matvec_shifted(hipk_csr *A, void *hip_stream, const void *x, int64_t ldx, void *y, int64_t ldy,
      int ncols, const double *shift_host);
/* diagonal of A (device array of nrows_local elements of dtype) */
const void *hipk_csr_diag(hipk_csr *A);
int64_t hipk_csr_nnz(const hipk_csr *A);
/* Matrix-free Laplacian stencil on an nx x ny x nz grid (nz = 1: 5-point),
 * diag = 2*dims, off = -1, Dirichlet; rows [row0, row0+m) in lexicographic order
 * (x fastest).  Same handle type so primme_amd_matvec dispatches on it. */
int hipk_stencil_create(hipk_ctx *ctx, hipk_dtype dt, int nx, int ny, int nz, int64_t row0,
      int64_t nrows_local, hipk_csr **A);
/* rows below/above the owned slab that the operator reads (0 for block-diagonal) */
/* single-rank rectangular matrix (nrows x ncols), the whole input vector is local: the two
 * factors of the singular value operator A'A */
int hipk_csr_create_rect(hipk_ctx *ctx, hipk_dtype dt, int64_t nrows, int64_t ncols,
      const int32_t *rowptr_host, const int32_t *colind_host, const void *values_host, hipk_csr **A);
int64_t hipk_csr_halo_lo(const hipk_csr *A);
int64_t hipk_csr_halo_hi(const hipk_csr *A);
/* halo buffers (device, halo_lo / halo_hi elements per column, column stride =
 * halo length) the communicator fills before each matvec */
int hipk_csr_set_halo(hipk_csr *A, const void *lo, const void *hi);
/* the same with explicit column strides (halos that live inside a gathered full-length block) */
int hipk_csr_set_halo_ld(hipk_csr *A, const void *lo, int64_t ld_lo, const void *hi, int64_t ld_hi);

/* y(:,c) = x(:,c) / (d - shift[c])  (Jacobi), shifts on host; |d - shift| is kept above
 * min_denominator with its sign (reference tests/COMMON/mat.c:149-165) */
int hipk_jacobi_apply(void *hip_stream, hipk_dtype dt, int64_t m, const void *diag,
      const double *shift_host, double min_denominator, const void *x, int64_t ldx, void *y,
      int64_t ldy, int ncols);
hipk_dtype hipk_csr_dtype(const hipk_csr *A);
int64_t hipk_csr_nrows(const hipk_csr *A);

/* ---- Rayleigh-Ritz small solve on the device (reference solve_projection.c:188-331 calls xHEEVX,
 * blaslapack.c:1024-1143).  Symmetric n x n (n <= 64), upper triangle of A_host referenced;
 * eigenvalues ascending in evals_host, orthonormal eigenvectors in Z_host.  One workgroup, parallel
 * cyclic Jacobi in LDS (round-robin pairing, n/2 rotations per step).  The solver uses the host QL
 * solver by default and this kernel when PRIMME_AMD_DEVICE_RR is set: the projected matrix is
 * assembled on the host anyway and a k <= 41 problem is latency, not throughput. */
int hipk_sym_eig(hipk_ctx *ctx, int n, const double *A_host, int lda, double *evals_host,
      double *Z_host, int ldz);

/* ---- measurement helpers ------------------------------------------------------- */
/* index bytes per nonzero the single-vector CSR kernel streams (2 with the 16-bit index stream built at
 * hipk_csr_create for banded / stencil / block-diagonal patterns, else 4): for byte accounting */
int hipk_csr_index_bytes(const hipk_csr *A);
/* Matrices created with hipk_csr_create_rect whose column pattern is scattered over an input vector larger than an
 * XCD's L2 also get a PANEL-BLOCKED form (csrc/hipk_sparse_pb.hip: columns cut into panels whose slice of x fits the
 * L2, one wave per row tile walking the panels with its row sums in registers; HIPK_PB=0 / 1 never / always,
 * HIPK_PB_KB bytes of x per panel).  Number of panels (0: plain CSR kernels) and the bytes one product streams. */
int hipk_csr_panels(const hipk_csr *A);
double hipk_csr_streamed_bytes(const hipk_csr *A);
/* Matrices whose rows repeat — a row being its sequence of (column - row, value) pairs; constant-coefficient stencils,
 * lattice operators: at most 256 distinct rows of at most 8 entries in the whole matrix — also get a ROW-PATTERN form
 * (csrc/hipk_sparse_pat.hip: one byte per row + a pattern table kept in LDS, one lane per row) that serves the
 * one-column products hipk_csr_matvec / hipk_csr_matvec_scaled with the arithmetic of the CSR tile kernel (y is
 * bit-identical; the fused form's inner product differs in summation order only).  HIPK_SPMV_PAT=0 in the
 * environment, or hipk_set_spmv_format(0) at run time, keeps the CSR tile kernels (A/B measurements, tests); returns
 * the previous setting.  hipk_csr_format: the form that serves one-column products now (0 CSR row tiles, 1
 * panel-blocked, 2 row patterns, 3 stencil); hipk_csr_product_bytes: the bytes one such product moves through HBM in
 * that form (fused: with the second output of hipk_csr_matvec_scaled) — what "streamed bytes" means in bench.py. */
int hipk_set_spmv_format(int use_patterns);
int hipk_csr_format(const hipk_csr *A);
int hipk_csr_npatterns(const hipk_csr *A);
double hipk_csr_product_bytes(const hipk_csr *A, int fused);
/* Second stage of the reductions of the block-size-1 iteration: bit 1 fused residual pass, 2 Gram-Schmidt update, 4 fused
 * SpMV run it inside the producing launch (two-level, write-through partial sums, csrc/hipk_internal.h); 0 = a separate
 * launch each.  Default 0: the in-kernel form is correct but slower on the MI355X (a workgroup waits ~5 us for its
 * write-through partial sums and its ticket, csrc/hipk_internal.h); environment HIPK_INKERNEL_FIN overrides.  Returns the
 * previous mask.  Both forms give bit-reproducible sums; they differ from each other in the last bits (different, fixed
 * summation orders). */
int hipk_set_inkernel_fin(int mask);
/* device copy bandwidth probe: copies `bytes` src->dst `reps` times, returns GB/s (read+write) */
int hipk_bandwidth_probe(hipk_ctx *ctx, size_t bytes, int reps, double *gbps);
/* read-only probe with the access pattern of the panel kernels (16 columns walked together, 16-byte loads): GB/s read */
int hipk_read_probe(hipk_ctx *ctx, size_t bytes, int reps, double *gbps);
/* live per-kernel-class timing with HIP events on the launching stream (process-wide,
 * off by default).  cls: 0 = TN inner products, 1 = NN project, 2 = fused Ritz update,
 * 3 = sparse matvec, 4 = element-wise / few-array passes (QMR recurrences of the JDQMR inner solver, axpy / xpay /
 * scale / copy / gather, column norms and pair products, the Jacobi preconditioner; their second-stage launches are
 * inside the timed span).  alg_bytes = algorithmic HBM bytes of the timed launches. */
int hipk_prof_enable(int on);
int hipk_prof_reset(void);
int hipk_prof_get(int cls, double *ms, long *launches, double *alg_bytes);

#ifdef __cplusplus
}
#endif
#endif /* PRIMME_AMD_KERNELS_H */
