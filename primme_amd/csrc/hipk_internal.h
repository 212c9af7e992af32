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
};

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
enum { HIPK_PROF_DOTS = 0, HIPK_PROF_PROJECT = 1, HIPK_PROF_RITZ = 2, HIPK_PROF_SPMV = 3, HIPK_PROF_NCLASS = 4 };
int hipk_prof_begin(int cls, hipStream_t st, double alg_bytes); /* returns slot or -1 */
void hipk_prof_end(int slot, hipStream_t st);

/* make sure ctx->partials can hold n doubles */
int hipk_reserve_partials(hipk_ctx *ctx, size_t n);
/* out[o] = sum_b partials[b*nout + o], deterministic order */
int hipk_finalize_partials(hipk_ctx *ctx, const double *partials, int nblocks, int nout,
      double *out_dev);

static inline int hipk_grid_for_rows(const hipk_ctx *ctx, int64_t m, int rows_per_block,
      int blocks_per_cu) {
   int64_t need = (m + rows_per_block - 1) / rows_per_block;
   int64_t cap = (int64_t)ctx->num_cu * blocks_per_cu;
   if (need < 1) need = 1;
   return (int)(need < cap ? need : cap);
}

#ifdef __HIPCC__
/* ---- scalar traits: real types now; complex panels use the same kernels ------ */
template <typename T> struct hipk_num;
template <> struct hipk_num<double> { typedef double acc_t; enum { acc_doubles = 1 }; };
template <> struct hipk_num<float>  { typedef double acc_t; enum { acc_doubles = 1 }; };

__device__ __forceinline__ double hipk_wave_sum(double v) {
#pragma unroll
   for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
   return v;
}
#endif

#endif
