/* hipk_internal.h — shared by the .hip translation units of the device layer. */
#ifndef HIPK_INTERNAL_H
#define HIPK_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "primme_amd_kernels.h"

#define HIPK_BLOCK 256
#define HIPK_WAVE 64

struct hipk_ctx {
   hipStream_t stream;
   int own_stream;
   int device;
   int num_cu;
   double *partials;     /* per-block partial sums of the two-stage reductions */
   size_t partials_cap;  /* in doubles */
   hipEvent_t ev0, ev1;
   /* reduction results written into [mirror_dev, mirror_dev+mirror_count) are also
    * stored by the finalize kernels into pinned host memory (zero-copy), so the host
    * needs only a stream synchronisation, no device->host copy */
   double *mirror_dev, *mirror_host;
   size_t mirror_count;
   void *jobtab;         /* device copy of a large Ritz-update job table (basis sizes > 64) */
   /* completion flag of the mirrored reductions: the finalize kernels, after their results reached the
    * pinned mirror, store the launch's sequence number in pinned host memory; hipk_wait_results spins
    * on it instead of going through hipStreamSynchronize (no runtime call, no interrupt wake-up) */
   volatile unsigned long long *flag_host;   /* pinned, host address */
   unsigned long long *flag_dev;             /* the same word, device address */
   unsigned int *fin_counter;                /* device: blocks of the running finalize launch that are done */
   unsigned int *arrive_counter;             /* device: arrival ticket of the in-kernel second stage (fin_counter + 16) */
   unsigned long long seq_issued;            /* sequence number of the last finalize launch with a mirror */
   unsigned long long seq_waited;            /* the last sequence number hipk_wait_results returned on */
   int need_sync;                            /* results were produced WITHOUT a flagged launch since the last one (early
                                                returns, memset paths): the next wait must drain the stream */
   int spin_wait;                            /* 0: always hipStreamSynchronize (HIPK_NO_SPINWAIT) */
   /* HIPK_HOST_TIMING=1 (measurement knob): time the host spends waiting for results, and from the return of
    * a wait to the next fused-residual launch (the part of an outer iteration the device sits idle for) */
   int host_timing;
   double ht_wait_s, ht_turn_s, ht_t_ret;
   long ht_waits, ht_turns;
   /* cross-rank second stage (row-partitioned runs on the peer-to-peer transport, comm_ipc.hip): when armed, the
    * NEXT finalize launch also exchanges its results with every rank through the mailboxes and stores the global
    * sums — local second stage + all-reduce + publication in one launch.  xr_lo/xr_count remember which results
    * are already global so that the solver's reduce step does not reduce them again. */
   /* pinned staging buffer of hipk_upload / hipk_download (transfers from / to memory the library did not pin itself): grows
    * on demand up to 32 MB, lives as long as the context */
   void *stage;
   size_t stage_cap;
   struct hipk_xreduce *xr;
   int xr_armed;
   const double *xr_lo;
   int xr_count;
   /* the tail of a block-size-1 iteration without its own second-stage launches (round 6, hipk_tail_defer / hipk_tail_finish in
    * include/primme_amd_kernels.h): the partial sums of |t|^2 stay in `tailp` (the operator launch adds them itself), those of
    * t'At in `partials`, and ONE small launch later adds both, publishes them and runs the Rayleigh-Ritz step of the next
    * iteration */
   double *tailp;                            /* device, HIPK_TAIL_MAXPART doubles */
   int tail_want;                            /* flags of the armed deferral (one-shot), 0: none */
   int tail_np2, tail_np3;                   /* partial sums waiting in tailp / partials (0: none) */
   double *tail_norm2_out, *tail_dot_out;    /* where the sums belong */
   int skip_flag_once;                       /* the next mirrored second stage publishes no completion flag (hipk_skip_next_flag) */
};
#define HIPK_TAIL_MAXPART 4096

/* what a kernel needs to take part in a mailbox reduction; filled by pa_ipc_xreduce_args */
#define HIPK_XR_MAXRANKS 16
struct hipk_xreduce {
   unsigned long long **tab;     /* device array [nranks]: every rank's granule area, mapped into this process */
   int nranks, rank;
   unsigned int slot_doubles;    /* capacity of one (generation, source rank) slot */
   unsigned int *seq;            /* HOST counter of the communicator: one tag per reduction, same sequence on every rank */
   int *err_dev;                 /* pinned error word (device address): set when a wait ran into the time limit */
   long long timeout_ticks;      /* of the 100 MHz wall clock */
   /* called (collectively: every rank reaches it at the same reduction) when the 32-bit tag sequence wraps: the owner drains
    * its device, meets the other ranks and clears the granule areas, so that no slot keeps a tag of the previous cycle */
   int (*on_wrap)(void *owner);
   void *owner;
};
/* by-value kernel argument */
struct hipk_xr_dev {
   unsigned long long **tab;
   int nranks, rank;
   unsigned int seq, slot_doubles;
   int *err;
   long long timeout_ticks;
};
static inline unsigned int hipk_xr_next_seq(hipk_xreduce *xr) {
   unsigned int q = ++*xr->seq;
   /* wrap-around after 2^32 - 1 reductions: 0 is the tag of an empty mailbox, and the generation (tag & 1) must keep
    * alternating (0xffffffff was odd), so the sequence continues at 2 — after the granule areas have been cleared on every
    * rank (a slot that was last written in the previous cycle could otherwise carry exactly the tag a reduction of this
    * cycle waits for, and stale data would be taken for the peer's contribution) */
   if (q == 0) {
      if (xr->on_wrap) (void)xr->on_wrap(xr->owner);
      *xr->seq = 2; q = 2;
   }
   return q;
}
static inline hipk_xr_dev hipk_xr_make(hipk_xreduce *xr) {
   hipk_xr_dev d;
   d.tab = xr->tab; d.nranks = xr->nranks; d.rank = xr->rank; d.seq = hipk_xr_next_seq(xr);
   d.slot_doubles = xr->slot_doubles; d.err = xr->err_dev; d.timeout_ticks = xr->timeout_ticks;
   return d;
}
static inline hipk_xr_dev hipk_xr_none(void) {
   hipk_xr_dev d; memset(&d, 0, sizeof(d)); return d;
}
/* the finalize launchers: take the arm (one-shot) and note the range that is global afterwards */
static inline hipk_xr_dev hipk_xr_take(hipk_ctx *ctx, const double *out_dev, int nout) {
   if (!ctx->xr_armed || !ctx->xr || nout > (int)ctx->xr->slot_doubles) { ctx->xr_armed = 0; return hipk_xr_none(); }
   ctx->xr_armed = 0;
   ctx->xr_lo = out_dev; ctx->xr_count = nout;
   return hipk_xr_make(ctx->xr);
}

static inline double *hipk_mirror_of(const hipk_ctx *ctx, const double *out_dev) {
   if (ctx->mirror_dev && out_dev >= ctx->mirror_dev && out_dev < ctx->mirror_dev + ctx->mirror_count)
      return ctx->mirror_host + (out_dev - ctx->mirror_dev);
   return NULL;
}

#define HIPK_CHECK(call)                                                          \
   do {                                                                           \
      hipError_t e_ = (call);                                                     \
      if (e_ != hipSuccess) {                                                     \
         fprintf(stderr, "primme_amd: %s failed at %s:%d: %s\n", #call, __FILE__, \
               __LINE__, hipGetErrorString(e_));                                  \
         return -1;                                                               \
      }                                                                           \
   } while (0)

/* ---- live per-kernel-class timing (HIP events on the launching stream) ----------
 * bench.py's roofline leg: average launch duration and algorithmic bytes of each
 * hot kernel class, measured during the timed solves. Off by default. */
/* VEC: the element-wise / few-array passes (QMR recurrences of the JDQMR inner solver, axpy / xpay / scale / copy / gather,
 * column norms and pair products, the Jacobi preconditioner) */
enum { HIPK_PROF_DOTS = 0, HIPK_PROF_PROJECT = 1, HIPK_PROF_RITZ = 2, HIPK_PROF_SPMV = 3, HIPK_PROF_VEC = 4, HIPK_PROF_NCLASS = 5 };
int hipk_prof_begin(int cls, hipStream_t st, double alg_bytes); /* returns slot or -1 */
int hipk_prof_begin_s(int cls, hipStream_t st, double alg_bytes, double streamed_bytes);   /* the same, with the bytes the format in use really moves */
void hipk_prof_end(int slot, hipStream_t st);
#ifdef __cplusplus
/* times everything enqueued on `st` between construction and the end of the scope as ONE launch of class `cls` */
struct hipk_prof_scope {
   int slot; hipStream_t st;
   hipk_prof_scope(int cls, hipStream_t s, double alg_bytes) : slot(hipk_prof_begin(cls, s, alg_bytes)), st(s) {}
   ~hipk_prof_scope() { hipk_prof_end(slot, st); }
};
#endif

/* arguments every finalize kernel takes for the completion flag (all NULL / 0: no flag) */
struct hipk_fin_flag { unsigned long long *flag; unsigned int *counter; unsigned long long seq; };
/* next sequence number for a finalize launch whose output lies in the mirror (else an empty record) */
static inline hipk_fin_flag hipk_next_flag(hipk_ctx *ctx, const double *out_dev) {
   hipk_fin_flag f = {NULL, NULL, 0};
   if (ctx->skip_flag_once) { ctx->skip_flag_once = 0; ctx->need_sync = 1; return f; }
   if (ctx->flag_dev && hipk_mirror_of(ctx, out_dev)) { f.flag = ctx->flag_dev; f.counter = ctx->fin_counter; f.seq = ++ctx->seq_issued; ctx->need_sync = 0; }
   else ctx->need_sync = 1;      /* a reduction whose results the flag does not cover */
   return f;
}
/* In-kernel second stage, two levels, no fence (round 4): the workgroups of a launch form GROUPS of `gsize` consecutive
 * blocks.  Every workgroup stores its partial sums WRITE-THROUGH (agent-scope relaxed atomic stores = `sc1` stores, o-major:
 * partials[o * nblocks + block]), drains them (s_waitcnt vmcnt(0)) and takes a ticket of its group; the last arriver of
 * a group adds the group's partials in block order (sc1 loads), stores the group sums write-through and takes a ticket of
 * the launch; the last group leader adds the group sums in group order, (peer-to-peer transport: exchanges them with the
 * other ranks,) stores the results in HBM and in the pinned mirror and publishes the completion flag.  Group membership
 * and both summation orders are fixed by the launch geometry, not by arrival: results are bit-reproducible.
 * Round 2's form paid an agent-scope RELEASE per workgroup (`buffer_wbl2`: the XCD's whole dirty L2, i.e. the kernel's own
 * output stream) and lost 2x.  This form needs no release (MI355X_MICROARCH.md, "valid forms": sc1 stores and loads on both
 * sides) and is correct (the whole GPU suite runs green with it, profiles/r04_inkernel_second_stage_ab.txt) — and it still
 * LOSES on the MI355X: a workgroup cannot leave before its write-through stores are acknowledged and its ticket has come
 * back, ~5 us during which its slot streams nothing.  Measured per launch against the separate second-stage launch
 * (~2.5 us + one kernel boundary): fused residual pass (512 workgroups) +6.6 us, Gram-Schmidt update +5.5 us, fused SpMV
 * (6 840 workgroups) +37 us; BASELINE configs[1] 205 -> 241 us per outer iteration with all three, 209 with the residual
 * pass alone.  OFF by default; kept as the tested alternative (HIPK_INKERNEL_FIN=<mask>, hipk_set_inkernel_fin). */
struct hipk_fin_args {
   double *out, *out_host;          /* results (device) and their pinned mirror (device address, may be NULL) */
   unsigned int *arrive;            /* device counters, zero between launches: [0] launch ticket, [1 + g] group tickets */
   double *group;                   /* device: group sums [nout][ngroups] */
   hipk_fin_flag flag;              /* completion flag record (flag == NULL: none) */
   hipk_xr_dev xr;                  /* tab != NULL: the results are exchanged with the other ranks before they are stored */
   int enabled, gsize;
};
/* HIPK_INKERNEL_FIN=<mask> selects the kernels (read once; default 0 = the separate second-stage launches) */
/* bit mask: 1 = fused residual kernel (2 workgroups per CU), 2 = Gram-Schmidt update, 4 = fused SpMV */
enum { HIPK_FIN_RITZ = 1, HIPK_FIN_PROJECT = 2, HIPK_FIN_SPMV = 4 };
#define HIPK_FIN_MAXGROUPS 256
int hipk_inkernel_fin_mask(void);          /* hipk_core.hip; hipk_set_inkernel_fin(mask) changes it at run time (A/B tests) */
int hipk_reserve_partials(hipk_ctx *ctx, size_t n);
/* nblocks workgroups, nout outputs per workgroup; the partials buffer must hold nblocks*nout + nout*HIPK_FIN_MAXGROUPS doubles
 * (hipk_make_fin reserves it: call it BEFORE taking ctx->partials) */
static inline hipk_fin_args hipk_make_fin(hipk_ctx *ctx, double *out_dev, int kind, int nblocks, int nout) {
   hipk_fin_args fa;
   memset(&fa, 0, sizeof(fa));
   fa.out = out_dev; fa.out_host = hipk_mirror_of(ctx, out_dev); fa.arrive = ctx->arrive_counter;
   fa.enabled = (hipk_inkernel_fin_mask() & kind) && ctx->arrive_counter != NULL && nblocks > 0 && nout > 0;
   if (fa.enabled) {
      int g = 8;
      while ((nblocks + g - 1) / g > HIPK_FIN_MAXGROUPS) g *= 2;
      fa.gsize = g;
      if (hipk_reserve_partials(ctx, (size_t)nblocks * nout + (size_t)nout * HIPK_FIN_MAXGROUPS)) { fa.enabled = 0; return fa; }
      fa.group = ctx->partials + (size_t)nblocks * nout;
      fa.xr = hipk_xr_take(ctx, out_dev, nout);
      fa.flag = hipk_next_flag(ctx, out_dev);
   }
   return fa;
}
void hipk_note_turnaround(hipk_ctx *ctx);
/* pageable host array -> device through a pinned staging buffer on the context's stream; complete on return */
int hipk_upload(hipk_ctx *ctx, void *dst, const void *src, size_t bytes);
int hipk_download(hipk_ctx *ctx, void *dst, const void *src, size_t bytes);
/* the context a matrix was created under (its uploads and default launches use that stream) */
hipk_ctx *hipk_csr_ctx(const hipk_csr *A);
/* hipk_csr_matvec_scaled applies: CSR, the input entries are the matrix' own row slab */
int hipk_csr_fusable(const hipk_csr *A);
/* make sure ctx->partials can hold n doubles */
int hipk_reserve_partials(hipk_ctx *ctx, size_t n);
/* out[o] = sum_b partials[b*nout + o], deterministic order */
int hipk_finalize_partials(hipk_ctx *ctx, const double *partials, int nblocks, int nout,
      double *out_dev);
int hipk_finalize_partials_strided(hipk_ctx *ctx, const double *partials, int nblocks, int pstride, int nout,
      double *out_dev);
/* the same for o-major partials (partials[o * nblocks + b]) */
int hipk_finalize_partials_t(hipk_ctx *ctx, const double *partials, int nblocks, int nout, double *out_dev);

static inline int hipk_grid_for_rows(const hipk_ctx *ctx, int64_t m, int rows_per_block,
      int blocks_per_cu) {
   int64_t need = (m + rows_per_block - 1) / rows_per_block;
   int64_t cap = (int64_t)ctx->num_cu * blocks_per_cu;
   if (need < 1) need = 1;
   return (int)(need < cap ? need : cap);
}

/* ---- the complex instantiation (hipk_complex.hip); the entry points of the real files dispatch here ---- */
#define HIPK_IS_Z(dt) ((dt) == HIPK_C64 || (dt) == HIPK_C32)
static inline hipk_dtype hipk_real_of(hipk_dtype dt) { return dt == HIPK_C64 ? HIPK_F64 : dt == HIPK_C32 ? HIPK_F32 : dt; }
int hipk_z_panel_dots(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const hipk_seg *segs, int nseg, const void *X, int64_t ldX, int nx,
      double *out_dev, int ldout);
int hipk_z_panel_project(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const hipk_seg *segs, int nseg, const double *coef, int ldcoef,
      const double *M, const void *X, int64_t ldX, void *Xout, int64_t ldXout, int nx, double *nrm2_dev);
int hipk_z_ritz_update(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const void *V, const void *W, int64_t ld, int k, const double *h, int ldh,
      const double *theta, const hipk_job *jobs, int njobs, double *nrm2_dev);
int hipk_z_axpy(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const double *alpha_host, const void *X, int64_t ldX, void *Y, int64_t ldY, int nx, int xpay);
int hipk_z_pair_dots(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const void *X, int64_t ldX, const void *Y, int64_t ldY, int nx, double *out_dev);
int hipk_z_csr_matvec(hipk_dtype dt, hipStream_t st, const int4 *tileinfo, int ntiles, const int32_t *rowptr, const int32_t *colind, const void *val,
      const void *x, int64_t ldx, void *y, int64_t ldy, int ncols, int64_t x0, int64_t xlen, int64_t halo_lo, int64_t halo_hi, const void *xlo,
      const void *xhi, int64_t ld_lo, int64_t ld_hi, const double *shift_host);
int hipk_z_jacobi(hipStream_t st, int num_cu, hipk_dtype dt, int64_t m, const void *diag, const double *shift_host, double min_den, const void *x,
      int64_t ldx, void *y, int64_t ldy, int ncols);

#ifdef __HIPCC__
/* ---- scalar traits: real types now; complex panels use the same kernels ------ */
template <typename T> struct hipk_num;
template <> struct hipk_num<double> { typedef double acc_t; enum { acc_doubles = 1 }; };
template <> struct hipk_num<float>  { typedef double acc_t; enum { acc_doubles = 1 }; };

/* Mailbox exchange of one value per group of 16 lanes (four values per wave; v and idx identical within a group,
 * `active` false: the group only takes part in the shuffles): lane p of the group writes this rank's value into
 * rank p's mailbox as two 8-byte {tag, half} granules (write-through, system scope: the mailboxes are peer-mapped
 * memory of other processes / other GPUs over xGMI) and polls the granules rank p wrote here; the values are then
 * added in rank order — the same order on every rank, so every rank holds identical bits.  No fence, no separate
 * flag: a granule is valid when its tag is the reduction's sequence number (MI355X_MICROARCH.md, hand-off rows).
 * Generation = seq & 1: a slot is rewritten two reductions later, which its writer can only reach after this rank
 * contributed to the reduction in between, i.e. after it finished this one.  All 64 lanes must call. */
__device__ __forceinline__ double hipk_xr_exchange(const hipk_xr_dev &x, unsigned idx, double v, bool active) {
   const int lane = threadIdx.x & 63, sub = lane & 15;
   const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
   const unsigned long long tag = (unsigned long long)x.seq << 32;
   const size_t gen = (size_t)(x.seq & 1u) * x.nranks;
   double mine = 0.0;
   if (active && sub < x.nranks) {
      unsigned long long *dst = x.tab[sub] + ((gen + x.rank) * x.slot_doubles + idx) * 2;
      __hip_atomic_store(dst, tag | (bits & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(dst + 1, tag | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      const unsigned long long *src = x.tab[x.rank] + ((gen + sub) * x.slot_doubles + idx) * 2;
      const long long t0 = wall_clock64();
      unsigned long long g0, g1;
      for (;;) {
         g0 = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
         g1 = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
         if ((g0 >> 32) == x.seq && (g1 >> 32) == x.seq) break;
         if (wall_clock64() - t0 > x.timeout_ticks) { *(volatile int *)x.err = 1; g0 = g1 = 0; break; }
         __builtin_amdgcn_s_sleep(2);
      }
      mine = __longlong_as_double((long long)((g1 << 32) | (g0 & 0xffffffffull)));
   }
   double acc = 0.0;
   for (int p = 0; p < x.nranks; p++) acc += __shfl(mine, (lane & 48) + p, 64);
   return acc;
}

/* called by thread 0 of every block of a finalize launch after its mirrored store: the block that
 * finishes last publishes the sequence number (system-scope fence first: the pinned results must be
 * visible to the host before the flag) */
__device__ __forceinline__ void hipk_publish_flag(const hipk_fin_flag &f, unsigned nblocks) {
   if (!f.flag) return;
   __threadfence_system();
   const unsigned t = atomicAdd(f.counter, 1u);
   if (t == nblocks - 1) {
      *f.counter = 0;
      __threadfence_system();
      *(volatile unsigned long long *)f.flag = f.seq;
   }
}

__device__ __forceinline__ double hipk_wave_sum_fwd(double v);

/* write-through store of a partial sum when the launch finalises in-kernel, plain store otherwise */
__device__ __forceinline__ void hipk_pstore(const hipk_fin_args &fa, double *p, double v) {
   if (fa.enabled) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
   else *p = v;
}
__device__ __forceinline__ double hipk_pload(const double *p) {
   return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
/* Called by EVERY thread of EVERY workgroup at the end of a kernel whose partials are o-major
 * (partials[o * nblocks + block], stored with hipk_pstore); `s_last` is an int in LDS.  See hipk_fin_args. */
__device__ __forceinline__ void hipk_inkernel_finalize(double *__restrict__ partials, int nout, unsigned nblocks,
      const hipk_fin_args &fa, int *s_last) {
   if (!fa.enabled) return;
   const unsigned G = (unsigned)fa.gsize, g = blockIdx.x / G, ng = (nblocks + G - 1) / G;
   const unsigned gfirst = g * G, gcount = (gfirst + G <= nblocks) ? G : nblocks - gfirst;
   asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      /* this wave's write-through partials have left */
   __syncthreads();
   if (threadIdx.x == 0) {
      const unsigned t = __hip_atomic_fetch_add(fa.arrive + 1 + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *s_last = (t == gcount - 1) ? 1 : 0;
   }
   __syncthreads();
   if (!*s_last) return;
   /* group leader: the group's partials in block order */
   for (int o = threadIdx.x; o < nout; o += blockDim.x) {
      const double *row = partials + (size_t)o * nblocks + gfirst;
      double acc = 0.0;
      for (unsigned i = 0; i < gcount; i++) acc += hipk_pload(row + i);
      __hip_atomic_store(fa.group + (size_t)o * ng + g, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
   }
   asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
   __syncthreads();
   if (threadIdx.x == 0) {
      __hip_atomic_store(fa.arrive + 1 + g, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned t = __hip_atomic_fetch_add(fa.arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *s_last = (t == ng - 1) ? 1 : 0;
   }
   __syncthreads();
   if (!*s_last) return;
   /* the last group leader: the group sums in group order; one 16-lane part of a wave per output */
   const int lane = threadIdx.x & 63, sub = lane & 15, part = (int)(threadIdx.x >> 4), nparts = (int)(blockDim.x >> 4);
   for (int o0 = 0; o0 < nout; o0 += nparts) {
      const int o = o0 + part;
      const bool live = o < nout;
      double acc = 0.0;
      if (live) {
         const double *row = fa.group + (size_t)o * ng;
         for (unsigned i = sub; i < ng; i += 16) acc += hipk_pload(row + i);
      }
      /* 16 lanes -> 1, fixed order */
      acc += __shfl_down(acc, 8, 16);
      acc += __shfl_down(acc, 4, 16);
      acc += __shfl_down(acc, 2, 16);
      acc += __shfl_down(acc, 1, 16);
      acc = __shfl(acc, lane & 48, 64);
      if (fa.xr.tab) acc = hipk_xr_exchange(fa.xr, (unsigned)(live ? o : 0), acc, live);
      if (live && sub == 0) {
         fa.out[o] = acc;
         if (fa.out_host) __hip_atomic_store(fa.out_host + o, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
   }
   asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      /* the mirrored results are out before the flag */
   __syncthreads();
   if (threadIdx.x == 0) {
      __hip_atomic_store(fa.arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (fa.flag.flag) __hip_atomic_store(fa.flag.flag, fa.flag.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
   }
}

__device__ __forceinline__ double hipk_wave_sum(double v) {
#pragma unroll
   for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
   return v;
}
__device__ __forceinline__ double hipk_wave_sum_fwd(double v) { return hipk_wave_sum(v); }

/* Sum of n partial sums another launch left in HBM, by the first 256 threads of a workgroup, in an order that depends on n
 * only: thread t adds p[t], p[t + 256], ... , the four waves add their lanes (hipk_wave_sum) and leave their sums in sm4;
 * after a barrier hipk_block_sum256_get gives every thread the same bits.  Every workgroup of the operator launch that
 * normalises with |t|^2 and the one-workgroup launch that publishes |t|^2 to the host use THIS order, so the number the host
 * sees is the number the vector was scaled with.  Threads >= 256 (a wider finishing launch) pass through. */
__device__ __forceinline__ void hipk_block_sum256_put(const double *__restrict__ p, int n, double *sm4) {
   if (threadIdx.x < 256) {
      double s = 0.0;
      int i = (int)threadIdx.x;
      for (; i + 768 < n; i += 1024) {
         const double a0 = p[i], a1 = p[i + 256], a2 = p[i + 512], a3 = p[i + 768];
         s += a0; s += a1; s += a2; s += a3;
      }
      for (; i < n; i += 256) s += p[i];
      s = hipk_wave_sum(s);
      if ((threadIdx.x & 63) == 0) sm4[threadIdx.x >> 6] = s;
   }
}
__device__ __forceinline__ double hipk_block_sum256_get(const double *sm4) { return (sm4[0] + sm4[1]) + (sm4[2] + sm4[3]); }
#endif

#endif
