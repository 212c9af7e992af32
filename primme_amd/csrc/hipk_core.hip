/* hipk_core.hip — context, HBM/pinned memory, stream-ordered copies, the
 * deterministic second stage of every reduction, and the bandwidth probe.
 * See include/primme_amd_kernels.h for the reference routines each entry replaces. */
#include "hipk_internal.h"
#include <time.h>
static double g_alloc_s = 0.0;      /* HIPK_HOST_TIMING: seconds spent in hipMalloc / hipFree */
static long g_alloc_n = 0;

extern "C" int hipk_ctx_create(hipk_ctx **out, void *stream_or_null) {
   int ndev = 0;
   if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
      fprintf(stderr, "primme_amd: no HIP device available (this library has no CPU path)\n");
      return -1;
   }
   hipk_ctx *ctx = (hipk_ctx *)calloc(1, sizeof(hipk_ctx));
   if (!ctx) return -2;
   HIPK_CHECK(hipGetDevice(&ctx->device));
   hipDeviceProp_t prop;
   HIPK_CHECK(hipGetDeviceProperties(&prop, ctx->device));
   ctx->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
   if (stream_or_null) {
      ctx->stream = *(hipStream_t *)stream_or_null;
      ctx->own_stream = 0;
   } else {
      HIPK_CHECK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
      ctx->own_stream = 1;
   }
   HIPK_CHECK(hipEventCreate(&ctx->ev0));
   HIPK_CHECK(hipEventCreate(&ctx->ev1));
   ctx->partials = NULL;
   ctx->partials_cap = 0;
   if (hipk_reserve_partials(ctx, (size_t)1 << 20)) return -2;
   {
      void *fh = NULL, *fd = NULL;
      HIPK_CHECK(hipHostMalloc(&fh, 64, hipHostMallocDefault));
      memset(fh, 0, 64);
      HIPK_CHECK(hipHostGetDevicePointer(&fd, fh, 0));
      ctx->flag_host = (volatile unsigned long long *)fh;
      ctx->flag_dev = (unsigned long long *)fd;
      /* [0, 8): ticket of the separate second-stage launches; [8, 8 + 1 + HIPK_FIN_MAXGROUPS): tickets of the in-kernel form */
      const size_t cbytes = sizeof(unsigned int) * (8 + 1 + HIPK_FIN_MAXGROUPS + 7);
      HIPK_CHECK(hipMalloc((void **)&ctx->fin_counter, cbytes));
      HIPK_CHECK(hipMemsetAsync(ctx->fin_counter, 0, cbytes, ctx->stream));   /* ordered before every kernel of this context */
      HIPK_CHECK(hipStreamSynchronize(ctx->stream));
      ctx->arrive_counter = ctx->fin_counter + 8;
      HIPK_CHECK(hipMalloc((void **)&ctx->tailp, sizeof(double) * HIPK_TAIL_MAXPART));
      ctx->seq_issued = 0;
      ctx->spin_wait = getenv("HIPK_NO_SPINWAIT") == NULL;
      ctx->host_timing = getenv("HIPK_HOST_TIMING") != NULL;
   }
   *out = ctx;
   return 0;
}

extern "C" int hipk_ctx_destroy(hipk_ctx *ctx) {
   if (!ctx) return 0;
   hipStreamSynchronize(ctx->stream);
   if (ctx->host_timing && g_alloc_n) { fprintf(stderr, "hipk host timing: %ld hipMalloc/hipFree calls so far, %.2f ms\n", g_alloc_n, 1e3 * g_alloc_s); }
   if (ctx->host_timing && ctx->ht_waits)
      fprintf(stderr, "hipk host timing: %ld waits, %.2f us each; %ld turnarounds (wait return -> fused residual launch), %.2f us each\n",
            ctx->ht_waits, 1e6 * ctx->ht_wait_s / ctx->ht_waits, ctx->ht_turns, ctx->ht_turns ? 1e6 * ctx->ht_turn_s / ctx->ht_turns : 0.0);
   if (ctx->partials) (void)hipFree(ctx->partials);
   if (ctx->jobtab) (void)hipFree(ctx->jobtab);
   if (ctx->fin_counter) (void)hipFree(ctx->fin_counter);
   if (ctx->tailp) (void)hipFree(ctx->tailp);
   if (ctx->stage) (void)hipHostFree(ctx->stage);
   if (ctx->flag_host) (void)hipHostFree((void *)ctx->flag_host);
   (void)hipEventDestroy(ctx->ev0);
   (void)hipEventDestroy(ctx->ev1);
   if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
   free(ctx);
   return 0;
}

extern "C" void *hipk_ctx_stream(hipk_ctx *ctx) { return (void *)ctx->stream; }

int hipk_reserve_partials(hipk_ctx *ctx, size_t n) {
   if (n <= ctx->partials_cap) return 0;
   /* growing while work is in flight: drain first (rare; sizes settle immediately) */
   HIPK_CHECK(hipStreamSynchronize(ctx->stream));
   if (ctx->partials) HIPK_CHECK(hipFree(ctx->partials));
   HIPK_CHECK(hipMalloc((void **)&ctx->partials, n * sizeof(double)));
   ctx->partials_cap = n;
   return 0;
}

static double alloc_now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
extern "C" int hipk_malloc(hipk_ctx *ctx, size_t bytes, void **dptr) {
   if (bytes == 0) bytes = 8;
   const double t0 = (ctx && ctx->host_timing) ? alloc_now() : 0.0;
   hipError_t e = hipMalloc(dptr, bytes);
   if (ctx && ctx->host_timing) { g_alloc_s += alloc_now() - t0; g_alloc_n++; }
   if (e != hipSuccess) { *dptr = NULL; return -2; }
   return 0;
}
extern "C" int hipk_free(hipk_ctx *ctx, void *dptr) {
   const double t0 = (ctx && ctx->host_timing) ? alloc_now() : 0.0;
   if (dptr) HIPK_CHECK(hipFree(dptr));
   if (ctx && ctx->host_timing && dptr) { g_alloc_s += alloc_now() - t0; g_alloc_n++; }
   return 0;
}
/* The pinned buffers this library handed out: hipk_h2d / hipk_d2h copy asynchronously only to / from these; any
 * other host pointer (the caller's numpy arrays, malloc'd or stack memory) is staged through a pinned buffer and the
 * copy is complete on return, so the runtime is never asked to DMA asynchronously out of or into pageable memory. */
#include <atomic>
#define HIPK_PINNED_MAX 256
static struct { const char *lo, *hi, *dev; } g_pinned[HIPK_PINNED_MAX];      /* dev: the device address of lo (mapped memory) */
static std::atomic_flag g_pinned_lock = ATOMIC_FLAG_INIT;
static void pinned_note(const void *p, size_t bytes, bool add) {
   void *dp = NULL;
   if (add && hipHostGetDevicePointer(&dp, (void *)p, 0) != hipSuccess) { dp = NULL; (void)hipGetLastError(); }
   while (g_pinned_lock.test_and_set(std::memory_order_acquire)) { }
   bool placed = !add;
   for (int i = 0; i < HIPK_PINNED_MAX; i++) {
      if (add ? g_pinned[i].lo == NULL : g_pinned[i].lo == (const char *)p) {
         g_pinned[i].lo = add ? (const char *)p : NULL;
         g_pinned[i].hi = add ? (const char *)p + bytes : NULL;
         g_pinned[i].dev = add ? (const char *)dp : NULL;
         placed = true;
         break;
      }
   }
   g_pinned_lock.clear(std::memory_order_release);
   static bool said = false;
   if (!placed && !said) {
      said = true;       /* still correct: such a buffer is recognised through hipPointerGetAttributes, without the copy kernel */
      fprintf(stderr, "primme_amd: more than %d pinned buffers alive; further ones take the runtime's copy path\n", HIPK_PINNED_MAX);
   }
}
/* is [p, p + bytes) inside one of the library's pinned buffers?  *dev (optional) gets the address a kernel reaches it by */
static bool pinned_has(const void *p, size_t bytes, const char **dev = NULL) {
   bool found = false;
   while (g_pinned_lock.test_and_set(std::memory_order_acquire)) { }
   for (int i = 0; i < HIPK_PINNED_MAX && !found; i++) {
      found = g_pinned[i].lo && (const char *)p >= g_pinned[i].lo && (const char *)p + bytes <= g_pinned[i].hi;
      if (found && dev) *dev = g_pinned[i].dev ? g_pinned[i].dev + ((const char *)p - g_pinned[i].lo) : NULL;
   }
   g_pinned_lock.clear(std::memory_order_release);
   return found;
}
extern "C" int hipk_host_alloc(hipk_ctx *ctx, size_t bytes, void **hptr) {
   (void)ctx;
   if (bytes == 0) bytes = 8;
   hipError_t e = hipHostMalloc(hptr, bytes, hipHostMallocDefault);
   if (e != hipSuccess) { *hptr = NULL; return -2; }
   pinned_note(*hptr, bytes, true);
   return 0;
}
extern "C" int hipk_host_free(hipk_ctx *ctx, void *hptr) {
   (void)ctx;
   if (hptr) { pinned_note(hptr, 0, false); HIPK_CHECK(hipHostFree(hptr)); }
   return 0;
}
/* Host array of ANY kind (pageable: numpy arrays, std::vector data, stack variables) -> device, finished on return.
 * Goes through a pinned staging buffer in chunks, each chunk an asynchronous copy on the context's stream followed
 * by a drain: the only transfers the device layer issues from memory it did not allocate itself.  (Round 2 used
 * blocking hipMemcpy on the NULL stream, which the context's non-blocking stream is not ordered against; an
 * asynchronous copy straight from pageable memory leaves it to the runtime to pin and unpin the caller's pages on the
 * fly — with that variant the GPU suite aborted once inside hipStreamSynchronize and hung once at exit,
 * profiles/r03_gpu_suite_run3_aborted.txt.) */
static int ctx_stage(hipk_ctx *ctx, size_t bytes, void **stage, size_t *cap) {
   const size_t want = bytes < ((size_t)32 << 20) ? bytes : ((size_t)32 << 20);
   if (want > ctx->stage_cap) {
      if (ctx->stage) { HIPK_CHECK(hipStreamSynchronize(ctx->stream)); (void)hipHostFree(ctx->stage); ctx->stage = NULL; ctx->stage_cap = 0; }
      size_t c = (size_t)1 << 16;
      while (c < want) c *= 2;
      HIPK_CHECK(hipHostMalloc(&ctx->stage, c, hipHostMallocDefault));
      ctx->stage_cap = c;
   }
   *stage = ctx->stage; *cap = ctx->stage_cap;
   return 0;
}
int hipk_upload(hipk_ctx *ctx, void *dst, const void *src, size_t bytes) {
   if (bytes == 0) return 0;
   void *stage = NULL;
   size_t cap = 0;
   if (ctx_stage(ctx, bytes, &stage, &cap)) return -1;
   for (size_t off = 0; off < bytes; off += cap) {
      const size_t n = bytes - off < cap ? bytes - off : cap;
      memcpy(stage, (const char *)src + off, n);
      if (hipMemcpyAsync((char *)dst + off, stage, n, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess) return -1;      /* the staging buffer is reused by the next chunk */
   }
   return 0;
}

/* device -> host array of any kind, complete on return (the counterpart of hipk_upload) */
int hipk_download(hipk_ctx *ctx, void *dst, const void *src, size_t bytes) {
   if (bytes == 0) return 0;
   void *stage = NULL;
   size_t cap = 0;
   if (ctx_stage(ctx, bytes, &stage, &cap)) return -1;
   for (size_t off = 0; off < bytes; off += cap) {
      const size_t n = bytes - off < cap ? bytes - off : cap;
      if (hipMemcpyAsync(stage, (const char *)src + off, n, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess) return -1;
      memcpy((char *)dst + off, stage, n);
   }
   return 0;
}
/* pinned host memory the library did not allocate itself (the caller's hipHostMalloc / hipHostRegister): asynchronous copies
 * may use it directly */
static bool foreign_pinned(const void *p) {
   hipPointerAttribute_t a;
   if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
   return a.type == hipMemoryTypeHost;
}
/* Small transfers between the library's pinned (mapped) buffers and HBM go through a copy KERNEL on the context's
 * stream instead of hipMemcpyAsync: the coefficient blocks and Ritz values a solver step uploads are a few KB, and a
 * runtime copy of that size costs 40-240 us of idle device before it starts (profiles/r03_config4_native_gaps.md:
 * 800 copies, 106 ms idle in two solves of configs[3]) where a kernel launch costs 5-8 us.  The kernel reads / writes
 * the host buffer through its device address; ordering and lifetime rules are those of the asynchronous copy it
 * replaces.  HIPK_NO_COPY_KERNEL=1 restores the runtime copies (A/B knob). */
#define HIPK_COPY_KERNEL_MAX ((size_t)1 << 20)
__global__ void __launch_bounds__(HIPK_BLOCK) hipk_small_copy_kernel(uint32_t *__restrict__ dst, const uint32_t *__restrict__ src, size_t nwords) {
   for (size_t i = (size_t)blockIdx.x * HIPK_BLOCK + threadIdx.x; i < nwords; i += (size_t)gridDim.x * HIPK_BLOCK) dst[i] = src[i];
}
static bool small_copy(hipk_ctx *ctx, void *dst_dev, const void *src_dev, size_t bytes) {
   static int off = -1;
   if (off < 0) off = getenv("HIPK_NO_COPY_KERNEL") != NULL;
   if (off || bytes > HIPK_COPY_KERNEL_MAX || (bytes & 3) || ((uintptr_t)dst_dev & 3) || ((uintptr_t)src_dev & 3) || !dst_dev || !src_dev) return false;
   const size_t nw = bytes / 4;
   const int gx = (int)((nw + HIPK_BLOCK - 1) / HIPK_BLOCK < 64 ? (nw + HIPK_BLOCK - 1) / HIPK_BLOCK : 64);
   hipLaunchKernelGGL(hipk_small_copy_kernel, dim3(gx), dim3(HIPK_BLOCK), 0, ctx->stream, (uint32_t *)dst_dev, (const uint32_t *)src_dev, nw);
   return hipGetLastError() == hipSuccess;
}
/* stream-ordered (asynchronous) for the library's pinned buffers, staged and complete on return for anything else */
extern "C" int hipk_h2d(hipk_ctx *ctx, void *dst, const void *src, size_t bytes) {
   if (!bytes) return 0;
   const char *sdev = NULL;
   if (!pinned_has(src, bytes, &sdev)) {
      if (!foreign_pinned(src)) return hipk_upload(ctx, dst, src, bytes);
      HIPK_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
      return 0;
   }
   if (small_copy(ctx, dst, sdev, bytes)) return 0;
   HIPK_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
   return 0;
}
extern "C" int hipk_d2h(hipk_ctx *ctx, void *dst, const void *src, size_t bytes) {
   if (!bytes) return 0;
   const char *ddev = NULL;
   if (!pinned_has(dst, bytes, &ddev)) {
      if (!foreign_pinned(dst)) return hipk_download(ctx, dst, src, bytes);
      HIPK_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
      return 0;
   }
   if (small_copy(ctx, (void *)ddev, src, bytes)) return 0;
   HIPK_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
   return 0;
}
extern "C" int hipk_d2d(hipk_ctx *ctx, void *dst, const void *src, size_t bytes) {
   if (bytes) HIPK_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
   return 0;
}
extern "C" int hipk_memset0(hipk_ctx *ctx, void *dst, size_t bytes) {
   if (bytes) HIPK_CHECK(hipMemsetAsync(dst, 0, bytes, ctx->stream));
   return 0;
}
extern "C" int hipk_sync(hipk_ctx *ctx) {
   HIPK_CHECK(hipStreamSynchronize(ctx->stream));
   ctx->seq_waited = ctx->seq_issued;
   ctx->need_sync = 0;
   return 0;
}
/* Wait until the results of the LAST mirrored reduction enqueued on the context are in the pinned
 * mirror.  The caller guarantees that reduction was the last thing it enqueued and that it only needs
 * those results (the stream is in order: everything before it has completed as well).  Spins on the
 * completion flag the finalize kernel publishes; falls back to a stream synchronisation when no
 * flagged launch is pending or the flag does not show up in time. */
static double ht_now(void) {
   struct timespec ts;
   clock_gettime(CLOCK_MONOTONIC, &ts);
   return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
static int wait_results_impl(hipk_ctx *ctx);
extern "C" int hipk_wait_results(hipk_ctx *ctx) {
   if (!ctx->host_timing) return wait_results_impl(ctx);
   const double t0 = ht_now();
   const int rc = wait_results_impl(ctx);
   ctx->ht_t_ret = ht_now();
   ctx->ht_wait_s += ctx->ht_t_ret - t0;
   ctx->ht_waits++;
   return rc;
}
/* called at the entry of the fused residual launch */
void hipk_note_turnaround(hipk_ctx *ctx) {
   if (!ctx->host_timing || ctx->ht_t_ret == 0.0) return;
   ctx->ht_turn_s += ht_now() - ctx->ht_t_ret;
   ctx->ht_turns++;
   ctx->ht_t_ret = 0.0;
}
static int wait_results_impl(hipk_ctx *ctx) {
   /* NOT a stream synchronisation: it returns when the last FLAGGED reduction has published its results.  Only
    * valid when such a launch was enqueued since the previous wait and nothing un-flagged produced results after
    * it; otherwise (stale sequence number, early-return / memset paths) drain the stream. */
   if (ctx->spin_wait && ctx->flag_host && ctx->seq_issued > ctx->seq_waited && !ctx->need_sync) {
      const unsigned long long want = ctx->seq_issued;
      for (long spins = 0; spins < 200000000L; spins++) {
         if (*ctx->flag_host >= want) { ctx->seq_waited = want; return 0; }
         __builtin_ia32_pause();
      }
   }
   HIPK_CHECK(hipStreamSynchronize(ctx->stream));
   ctx->seq_waited = ctx->seq_issued;
   ctx->need_sync = 0;
   return 0;
}
/* sequence number of the last flagged reduction enqueued on the context; wait for a PARTICULAR one while later ones are
 * already queued behind it (the iteration enqueued ahead of the host, eigs_conv.c).  Falls back to draining the stream. */
extern "C" unsigned long long hipk_seq_issued(hipk_ctx *ctx) { return ctx->seq_issued; }
extern "C" int hipk_wait_seq(hipk_ctx *ctx, unsigned long long want) {
   if (ctx->spin_wait && ctx->flag_host && want > 0 && want <= ctx->seq_issued) {
      for (long spins = 0; spins < 200000000L; spins++) {
         if (*ctx->flag_host >= want) { if (want > ctx->seq_waited) ctx->seq_waited = want; return 0; }
         __builtin_ia32_pause();
      }
   }
   HIPK_CHECK(hipStreamSynchronize(ctx->stream));
   ctx->seq_waited = ctx->seq_issued;
   ctx->need_sync = 0;
   return 0;
}
extern "C" int hipk_is_device_ptr(const void *p) {
   hipPointerAttribute_t attr;
   if (!p) return 0;
   if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
      (void)hipGetLastError(); /* plain host memory: clear the sticky error */
      return 0;
   }
   return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}
extern "C" int hipk_timer_start(hipk_ctx *ctx) {
   HIPK_CHECK(hipEventRecord(ctx->ev0, ctx->stream));
   return 0;
}
extern "C" int hipk_timer_stop(hipk_ctx *ctx, float *ms) {
   HIPK_CHECK(hipEventRecord(ctx->ev1, ctx->stream));
   HIPK_CHECK(hipEventSynchronize(ctx->ev1));
   HIPK_CHECK(hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
   return 0;
}

/* ---- stage 2 of every reduction: one block per output, fixed summation order -- */
#define FIN_MAXBLOCK 1024
/* STRIDE_O = true: partial-major (partials[b * nout + o]); false: o-major (partials[o * nblocks + b]).
 * Every lane issues four loads before the first add (the launch is pure latency: a few KB out of L2),
 * and the block grows with the number of partials so that no lane makes more than a few rounds. */
template <bool PARTIAL_MAJOR, bool XR>
__global__ void __launch_bounds__(FIN_MAXBLOCK)
hipk_finalize_kernel(const double *__restrict__ partials, int nblocks, int nout, int pstride,
      double *__restrict__ out, double *__restrict__ out_host, hipk_fin_flag fin, hipk_xr_dev xr) {
   __shared__ double sm[FIN_MAXBLOCK / HIPK_WAVE];
   const int o = blockIdx.x;
   const int nt = blockDim.x;
   const size_t so = PARTIAL_MAJOR ? (size_t)o : (size_t)o * nblocks;
   const size_t sb = PARTIAL_MAJOR ? (size_t)pstride : 1;
   double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
   int b = threadIdx.x;
   for (; b + 3 * nt < nblocks; b += 4 * nt) {
      const double a0 = partials[so + (size_t)b * sb], a1 = partials[so + (size_t)(b + nt) * sb];
      const double a2 = partials[so + (size_t)(b + 2 * nt) * sb], a3 = partials[so + (size_t)(b + 3 * nt) * sb];
      s0 += a0; s1 += a1; s2 += a2; s3 += a3;
   }
   {  /* up to three more, clamped instead of predicated so that they go out together */
      const int last = nblocks - 1;
      const int b0 = b, b1 = b + nt, b2 = b + 2 * nt;
      const double a0 = partials[so + (size_t)(b0 < last ? b0 : last) * sb];
      const double a1 = partials[so + (size_t)(b1 < last ? b1 : last) * sb];
      const double a2 = partials[so + (size_t)(b2 < last ? b2 : last) * sb];
      s0 += b0 < nblocks ? a0 : 0.0; s1 += b1 < nblocks ? a1 : 0.0; s2 += b2 < nblocks ? a2 : 0.0;
   }
   double s = hipk_wave_sum((s0 + s1) + (s2 + s3));
   if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
   __syncthreads();
   if (XR) {
      /* row-partitioned run on the peer-to-peer transport: this rank's sum goes to every rank's mailbox and the
       * global sum (rank order, identical bits everywhere) is what gets stored: second stage + all-reduce +
       * publication in this one launch (comm_ipc.hip) */
      if (threadIdx.x < 64) {
         double v = 0.0;
         const int nw = nt >> 6;
         for (int w = 0; w < nw; w++) v += sm[w];
         v = hipk_xr_exchange(xr, (unsigned)o, v, threadIdx.x < 16);
         if (threadIdx.x == 0) {
            out[o] = v;
            if (out_host) out_host[o] = v;
            hipk_publish_flag(fin, gridDim.x);
         }
      }
      return;
   }
   if (threadIdx.x == 0) {
      double v = 0.0;
      const int nw = nt >> 6;
      for (int w = 0; w < nw; w++) v += sm[w];
      out[o] = v;
      if (out_host) out_host[o] = v;
      hipk_publish_flag(fin, gridDim.x);
   }
}
static inline int fin_block_for(int nblocks) {
   return nblocks <= 1024 ? HIPK_BLOCK : (nblocks <= 2048 ? 512 : FIN_MAXBLOCK);
}
template <bool PM>
static int launch_finalize(hipk_ctx *ctx, const double *partials, int nblocks, int pstride, int nout, double *out_dev) {
   if (nout <= 0) return 0;
   const hipk_xr_dev xr = hipk_xr_take(ctx, out_dev, nout);
   double *mh = hipk_mirror_of(ctx, out_dev);
   const hipk_fin_flag ff = hipk_next_flag(ctx, out_dev);
   if (xr.tab)
      hipLaunchKernelGGL((hipk_finalize_kernel<PM, true>), dim3(nout), dim3(fin_block_for(nblocks)), 0, ctx->stream,
            partials, nblocks, nout, pstride, out_dev, mh, ff, xr);
   else
      hipLaunchKernelGGL((hipk_finalize_kernel<PM, false>), dim3(nout), dim3(fin_block_for(nblocks)), 0, ctx->stream,
            partials, nblocks, nout, pstride, out_dev, mh, ff, xr);
   HIPK_CHECK(hipGetLastError());
   return 0;
}
int hipk_finalize_partials_t(hipk_ctx *ctx, const double *partials, int nblocks, int nout, double *out_dev) {
   return launch_finalize<false>(ctx, partials, nblocks, nout, nout, out_dev);
}

int hipk_finalize_partials(hipk_ctx *ctx, const double *partials, int nblocks, int nout,
      double *out_dev) {
   return hipk_finalize_partials_strided(ctx, partials, nblocks, nout, nout, out_dev);
}
/* nout results out of rows of `pstride` partials (partials[b * pstride + o], o < nout) */
int hipk_finalize_partials_strided(hipk_ctx *ctx, const double *partials, int nblocks, int pstride, int nout,
      double *out_dev) {
   return launch_finalize<true>(ctx, partials, nblocks, pstride, nout, out_dev);
}

/* which kernels run their second stage in-kernel (hipk_internal.h: hipk_fin_args) */
static int g_fin_mask = -1;
int hipk_inkernel_fin_mask(void) {
   if (g_fin_mask < 0) { const char *e = getenv("HIPK_INKERNEL_FIN"); g_fin_mask = e ? atoi(e) : 0; }
   return g_fin_mask;
}
extern "C" int hipk_set_inkernel_fin(int mask) { const int old = hipk_inkernel_fin_mask(); g_fin_mask = mask & 7; return old; }

/* ---- the iteration tail without its own second-stage launches (primme_amd_kernels.h: hipk_tail_defer) ---- */
static int g_tail_off = -1;
extern "C" int hipk_tail_defer(hipk_ctx *ctx, int want) {
   if (g_tail_off < 0) g_tail_off = getenv("HIPK_NO_TAIL_DEFER") != NULL;      /* A/B knob: the round-5 sequence of launches */
   ctx->tail_want = 0; ctx->tail_np2 = 0; ctx->tail_np3 = 0; ctx->tail_norm2_out = NULL; ctx->tail_dot_out = NULL;
   if (g_tail_off || !ctx->tailp || hipk_inkernel_fin_mask()) return 0;
   ctx->tail_want = want & (HIPK_TAIL_NORM | HIPK_TAIL_DOT);
   return ctx->tail_want;
}
/* the NEXT mirrored second stage on this context stores its results (HBM + pinned mirror) but publishes no completion flag:
 * no system-scope fences, no ticket — for a reduction the host only looks at after a LATER flagged launch of the same stream
 * (the kernel boundary in between has made its mirrored stores visible).  One-shot. */
extern "C" void hipk_skip_next_flag(hipk_ctx *ctx) { ctx->skip_flag_once = 1; }
extern "C" void hipk_tail_abandon(hipk_ctx *ctx) {
   ctx->tail_want = 0; ctx->tail_np2 = 0; ctx->tail_np3 = 0; ctx->tail_norm2_out = NULL; ctx->tail_dot_out = NULL;
}
extern "C" int hipk_tail_pending(hipk_ctx *ctx) { return (ctx->tail_np2 > 0 ? HIPK_TAIL_NORM : 0) | (ctx->tail_np3 > 0 ? HIPK_TAIL_DOT : 0); }

/* ---- cross-rank second stage (see hipk_internal.h: hipk_ctx.xr) ---- */
extern "C" void hipk_xreduce_arm(hipk_ctx *ctx) { if (ctx->xr) ctx->xr_armed = 1; }
extern "C" int hipk_xreduce_available(hipk_ctx *ctx) { return ctx && ctx->xr ? 1 : 0; }
extern "C" int hipk_xreduce_covered(hipk_ctx *ctx, const double *buf, int count) {
   const int yes = ctx->xr_lo && buf >= ctx->xr_lo && buf + count <= ctx->xr_lo + ctx->xr_count;
   ctx->xr_lo = NULL; ctx->xr_count = 0;
   ctx->xr_armed = 0;      /* an arm nobody took (a producer without a second stage) must not leak into a later launch */
   return yes;
}

/* results produced by something else than our reductions (an all-reduce): into the mirror, then the flag */
__global__ void __launch_bounds__(HIPK_BLOCK)
hipk_publish_kernel(const double *__restrict__ src, double *__restrict__ dst_host, int n, hipk_fin_flag fin) {
   for (int i = threadIdx.x; i < n; i += HIPK_BLOCK) dst_host[i] = src[i];
   __syncthreads();
   if (threadIdx.x == 0) hipk_publish_flag(fin, 1);
}
extern "C" int hipk_publish_results(hipk_ctx *ctx, const double *dev, int count) {
   if (count <= 0) return 0;
   double *mh = hipk_mirror_of(ctx, dev);
   if (!mh || !ctx->flag_dev || !ctx->spin_wait || !hipk_mirror_of(ctx, dev + count - 1)) return 1;
   hipLaunchKernelGGL(hipk_publish_kernel, dim3(1), dim3(HIPK_BLOCK), 0, ctx->stream, dev, mh, count, hipk_next_flag(ctx, dev));
   HIPK_CHECK(hipGetLastError());
   return 0;
}

extern "C" int hipk_ctx_set_mirror(hipk_ctx *ctx, double *dev_base, double *pinned_host_base, size_t count) {
   ctx->mirror_dev = dev_base;
   ctx->mirror_count = count;
   ctx->mirror_host = NULL;
   if (dev_base && pinned_host_base) {
      void *dp = NULL;
      HIPK_CHECK(hipHostGetDevicePointer(&dp, pinned_host_base, 0));
      ctx->mirror_host = (double *)dp;
   }
   return 0;
}

/* ---- LAPACK's xLARNV(idist = 2) stream generated on the device ------------------------------------------
 * The reference fills random vectors with Num_larnv (host xLARNV + upload, blaslapack.c:938-988 /
 * cublas_wrapper.c:707-736).  The generator is the 48-bit multiplicative congruential x <- a x mod 2^48,
 * a = 33952834046453; element i of the stream that starts at state s is (a^(i+1) s) mod 2^48, so every lane can
 * jump to its own element with a table of a^(2^b): the SAME numbers, bit for bit, as the host routine
 * (pa_larnv_uniform11, verified against LAPACK), without 8 bytes per element crossing PCIe.  n counts REAL numbers
 * (a complex element takes two consecutive ones: re, im). */
struct LarnvPow { unsigned long long p[48]; };
template <typename T>
__global__ void __launch_bounds__(HIPK_BLOCK)
hipk_larnv_kernel(T *__restrict__ x, int64_t n, unsigned long long s0, LarnvPow pw) {
   const unsigned long long MASK = (1ULL << 48) - 1;
   const int64_t stride = (int64_t)gridDim.x * HIPK_BLOCK;
   for (int64_t i = (int64_t)blockIdx.x * HIPK_BLOCK + threadIdx.x; i < n; i += stride) {
      unsigned long long e = (unsigned long long)i + 1ULL, v = s0;
      for (int b = 0; e; b++, e >>= 1) if (e & 1ULL) v = (v * pw.p[b]) & MASK;
      x[i] = (T)(2.0 * ((double)v / 281474976710656.0) - 1.0);
   }
}
extern "C" int hipk_larnv_uniform11(hipk_ctx *ctx, hipk_dtype dt, int64_t iseed[4], int64_t n, void *x) {
   if (n <= 0) return 0;
   const unsigned long long A = 33952834046453ULL, MASK = (1ULL << 48) - 1;
   const unsigned long long s0 = ((((unsigned long long)iseed[0] * 4096 + (unsigned long long)iseed[1]) * 4096 +
                                   (unsigned long long)iseed[2]) * 4096 + (unsigned long long)iseed[3]) & MASK;
   LarnvPow pw;
   pw.p[0] = A;
   for (int b = 1; b < 48; b++) pw.p[b] = (pw.p[b - 1] * pw.p[b - 1]) & MASK;
   const int gx = hipk_grid_for_rows(ctx, n, HIPK_BLOCK * 2, 8);
   const bool dbl = (dt == HIPK_F64 || dt == HIPK_C64);
   if (dbl) hipLaunchKernelGGL(hipk_larnv_kernel<double>, dim3(gx), dim3(HIPK_BLOCK), 0, ctx->stream, (double *)x, n, s0, pw);
   else hipLaunchKernelGGL(hipk_larnv_kernel<float>, dim3(gx), dim3(HIPK_BLOCK), 0, ctx->stream, (float *)x, n, s0, pw);
   HIPK_CHECK(hipGetLastError());
   /* the state after n numbers: a^n s0 */
   unsigned long long e = (unsigned long long)n, v = s0;
   for (int b = 0; e; b++, e >>= 1) if (e & 1ULL) v = (v * pw.p[b]) & MASK;
   iseed[0] = (int64_t)((v >> 36) & 4095); iseed[1] = (int64_t)((v >> 24) & 4095);
   iseed[2] = (int64_t)((v >> 12) & 4095); iseed[3] = (int64_t)(v & 4095);
   return 0;
}

/* ---- attainable-HBM probe (device copy, 16 B per lane) ------------------------ */
__global__ void __launch_bounds__(HIPK_BLOCK)
hipk_copy16_kernel(const double2 *__restrict__ src, double2 *__restrict__ dst, size_t n16) {
   size_t i = (size_t)blockIdx.x * HIPK_BLOCK + threadIdx.x;
   const size_t stride = (size_t)gridDim.x * HIPK_BLOCK;
   for (; i + 3 * stride < n16; i += 4 * stride) {
      double2 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
      dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
   }
   for (; i < n16; i += stride) dst[i] = src[i];
}

extern "C" int hipk_bandwidth_probe(hipk_ctx *ctx, size_t bytes, int reps, double *gbps) {
   void *a = NULL, *b = NULL;
   if (hipk_malloc(ctx, bytes, &a) || hipk_malloc(ctx, bytes, &b)) return -2;
   HIPK_CHECK(hipMemsetAsync(a, 1, bytes, ctx->stream));
   size_t n16 = bytes / 16;
   int grid = ctx->num_cu * 8;
   for (int w = 0; w < 2; w++)
      hipLaunchKernelGGL(hipk_copy16_kernel, dim3(grid), dim3(HIPK_BLOCK), 0, ctx->stream,
            (const double2 *)a, (double2 *)b, n16);
   float ms = 0;
   hipk_timer_start(ctx);
   for (int r = 0; r < reps; r++)
      hipLaunchKernelGGL(hipk_copy16_kernel, dim3(grid), dim3(HIPK_BLOCK), 0, ctx->stream,
            (const double2 *)a, (double2 *)b, n16);
   if (hipk_timer_stop(ctx, &ms)) return -1;
   *gbps = 2.0 * (double)bytes * reps / (ms * 1e-3) / 1e9;
   hipk_free(ctx, a);
   hipk_free(ctx, b);
   return 0;
}

/* read-only stream probe with the access pattern of the panel kernels: NC columns walked together, 16-byte
 * non-temporal loads, one partial sum per block (what "attainable read bandwidth" means for ritz_cgs_kernel) */
typedef double probe_v2 __attribute__((ext_vector_type(2)));
template <int NC>
__global__ void __launch_bounds__(HIPK_BLOCK) hipk_read_probe_kernel(const probe_v2 *__restrict__ a, size_t rows2, double *__restrict__ part) {
   double acc = 0.0;
   for (size_t i = (size_t)blockIdx.x * HIPK_BLOCK + threadIdx.x; i < rows2; i += (size_t)gridDim.x * HIPK_BLOCK) {
      probe_v2 v[NC];
#pragma unroll
      for (int c = 0; c < NC; c++) v[c] = __builtin_nontemporal_load(a + (size_t)c * rows2 + i);
#pragma unroll
      for (int c = 0; c < NC; c++) acc += v[c].x + v[c].y;
   }
   acc = hipk_wave_sum(acc);
   if ((threadIdx.x & 63) == 0) part[blockIdx.x * 4 + (threadIdx.x >> 6)] = acc;
}
extern "C" int hipk_read_probe(hipk_ctx *ctx, size_t bytes, int reps, double *gbps) {
   const int NC = 16;
   void *a = NULL;
   const size_t rows2 = bytes / (16 * NC);
   if (rows2 == 0) return -1;
   const int grid = ctx->num_cu * 4;
   if (hipk_malloc(ctx, rows2 * 16 * NC, &a) || hipk_reserve_partials(ctx, (size_t)grid * 4)) return -2;
   HIPK_CHECK(hipMemsetAsync(a, 0, rows2 * 16 * NC, ctx->stream));
   for (int w = 0; w < 2; w++)
      hipLaunchKernelGGL(hipk_read_probe_kernel<16>, dim3(grid), dim3(HIPK_BLOCK), 0, ctx->stream, (const probe_v2 *)a, rows2, ctx->partials);
   float ms = 0;
   hipk_timer_start(ctx);
   for (int r = 0; r < reps; r++)
      hipLaunchKernelGGL(hipk_read_probe_kernel<16>, dim3(grid), dim3(HIPK_BLOCK), 0, ctx->stream, (const probe_v2 *)a, rows2, ctx->partials);
   if (hipk_timer_stop(ctx, &ms)) return -1;
   *gbps = (double)(rows2 * 16 * NC) * reps / (ms * 1e-3) / 1e9;
   hipk_free(ctx, a);
   return 0;
}

/* ---- live kernel-class profiler ---------------------------------------------- */
#define PROF_RING 4096
static struct {
   int enabled, inited;
   hipEvent_t e0[PROF_RING], e1[PROF_RING];
   int cls[PROF_RING];
   int head, tail;                  /* slots [tail, head) are in flight */
   double ms[HIPK_PROF_NCLASS], bytes[HIPK_PROF_NCLASS], streamed[HIPK_PROF_NCLASS];   /* streamed: what the format in use moves (<= bytes for a compressed operator) */
   long launches[HIPK_PROF_NCLASS];
} g_prof;

static void prof_drain(int all) {
   while (g_prof.tail != g_prof.head) {
      const int s = g_prof.tail % PROF_RING;
      if (!all && hipEventQuery(g_prof.e1[s]) != hipSuccess) break;
      if (all) (void)hipEventSynchronize(g_prof.e1[s]);
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, g_prof.e0[s], g_prof.e1[s]) == hipSuccess) {
         g_prof.ms[g_prof.cls[s]] += ms;
         g_prof.launches[g_prof.cls[s]]++;
      }
      g_prof.tail++;
   }
}

int hipk_prof_begin(int cls, hipStream_t st, double alg_bytes) { return hipk_prof_begin_s(cls, st, alg_bytes, alg_bytes); }
int hipk_prof_begin_s(int cls, hipStream_t st, double alg_bytes, double streamed_bytes) {
   if (!g_prof.enabled) return -1;
   if (g_prof.head - g_prof.tail >= PROF_RING - 1) prof_drain(0);
   if (g_prof.head - g_prof.tail >= PROF_RING - 1) prof_drain(1);
   const int s = g_prof.head % PROF_RING;
   g_prof.cls[s] = cls;
   g_prof.bytes[cls] += alg_bytes;
   g_prof.streamed[cls] += streamed_bytes;
   (void)hipEventRecord(g_prof.e0[s], st);
   return s;
}
void hipk_prof_end(int slot, hipStream_t st) {
   if (slot < 0) return;
   (void)hipEventRecord(g_prof.e1[slot], st);
   g_prof.head++;
}

extern "C" int hipk_prof_enable(int on) {
   if (on && !g_prof.inited) {
      for (int i = 0; i < PROF_RING; i++) {
         if (hipEventCreate(&g_prof.e0[i]) != hipSuccess || hipEventCreate(&g_prof.e1[i]) != hipSuccess) return -1;
      }
      g_prof.inited = 1;
   }
   if (!on && g_prof.inited) prof_drain(1);
   g_prof.enabled = on;
   return 0;
}
extern "C" int hipk_prof_reset(void) {
   if (g_prof.inited) prof_drain(1);
   for (int c = 0; c < HIPK_PROF_NCLASS; c++) { g_prof.ms[c] = 0; g_prof.bytes[c] = 0; g_prof.streamed[c] = 0; g_prof.launches[c] = 0; }
   return 0;
}
/* cls: 0 dots (TN panel), 1 project (NN accumulate), 2 ritz (fused update), 3 spmv */
extern "C" int hipk_prof_get(int cls, double *ms, long *launches, double *alg_bytes) {
   if (cls < 0 || cls >= HIPK_PROF_NCLASS) return -1;
   if (g_prof.inited) prof_drain(1);
   *ms = g_prof.ms[cls]; *launches = g_prof.launches[cls]; *alg_bytes = g_prof.bytes[cls];
   return 0;
}
/* the bytes the launches of the class really moved through HBM in the form they ran in (the sparse operator in a compressed
 * form moves fewer than the CSR-algorithmic count above; equal for every other class) */
extern "C" double hipk_prof_streamed(int cls) {
   if (cls < 0 || cls >= HIPK_PROF_NCLASS) return 0.0;
   if (g_prof.inited) prof_drain(1);
   return g_prof.streamed[cls];
}


/* ======================= Rayleigh-Ritz small solve: parallel cyclic Jacobi ===================== */
#define EIG_MAXN 64
__global__ void __launch_bounds__(HIPK_BLOCK)
sym_eig_jacobi_kernel(const double *__restrict__ Ain, int n, int np, double *__restrict__ evals,
      double *__restrict__ Zout, int *__restrict__ sweeps_out) {
   extern __shared__ double sh[];
   double *A = sh, *V = sh + (size_t)np * np;
   __shared__ int top[EIG_MAXN / 2], bot[EIG_MAXN / 2];
   __shared__ double cs[EIG_MAXN / 2], sn[EIG_MAXN / 2];
   __shared__ double offd[HIPK_BLOCK / HIPK_WAVE], dia[HIPK_BLOCK / HIPK_WAVE];
   __shared__ int done;
   const int tid = threadIdx.x, half = np / 2;
   for (int idx = tid; idx < np * np; idx += HIPK_BLOCK) {
      const int i = idx % np, j = idx / np;
      double v = 0.0;
      if (i < n && j < n) v = (i <= j) ? Ain[i + (size_t)j * n] : Ain[j + (size_t)i * n];
      A[idx] = v;
      V[idx] = (i == j) ? 1.0 : 0.0;
   }
   if (tid < half) { top[tid] = 2 * tid; bot[tid] = 2 * tid + 1; }
   if (tid == 0) done = 0;
   __syncthreads();
   int sweep = 0;
   for (; sweep < 40; sweep++) {
      /* off-diagonal weight against the diagonal: stop when it is rounding noise */
      double o = 0.0, d = 0.0;
      for (int idx = tid; idx < np * np; idx += HIPK_BLOCK) {
         const int i = idx % np, j = idx / np;
         const double v = A[idx];
         if (i == j) d += v * v; else o += v * v;
      }
      o = hipk_wave_sum(o); d = hipk_wave_sum(d);
      if ((tid & 63) == 0) { offd[tid >> 6] = o; dia[tid >> 6] = d; }
      __syncthreads();
      if (tid == 0) {
         const double ot = (offd[0] + offd[1]) + (offd[2] + offd[3]), dt = (dia[0] + dia[1]) + (dia[2] + dia[3]);
         done = (ot <= 1e-34 * dt) || (ot == 0.0);
      }
      __syncthreads();
      if (done) break;
      for (int step = 0; step < np - 1; step++) {
         if (tid < half) {
            const int p = min(top[tid], bot[tid]), q = max(top[tid], bot[tid]);
            const double apq = A[p + (size_t)q * np], app = A[p + (size_t)p * np], aqq = A[q + (size_t)q * np];
            double c = 1.0, s = 0.0;
            if (fabs(apq) > 1e-300 && fabs(apq) > 1e-20 * sqrt(fabs(app * aqq)) ) {
               const double tau = (aqq - app) / (2.0 * apq);
               const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
               c = 1.0 / sqrt(1.0 + t * t);
               s = t * c;
            }
            cs[tid] = c; sn[tid] = s;
         }
         __syncthreads();
         /* A <- A J, V <- V J (columns p, q of every pair) */
         for (int idx = tid; idx < half * np; idx += HIPK_BLOCK) {
            const int k = idx / np, r = idx % np;
            const int p = min(top[k], bot[k]), q = max(top[k], bot[k]);
            const double c = cs[k], s = sn[k];
            const double ap = A[r + (size_t)p * np], aq = A[r + (size_t)q * np];
            A[r + (size_t)p * np] = c * ap - s * aq;
            A[r + (size_t)q * np] = s * ap + c * aq;
            const double vp = V[r + (size_t)p * np], vq = V[r + (size_t)q * np];
            V[r + (size_t)p * np] = c * vp - s * vq;
            V[r + (size_t)q * np] = s * vp + c * vq;
         }
         __syncthreads();
         /* A <- J' A (rows p, q) */
         for (int idx = tid; idx < half * np; idx += HIPK_BLOCK) {
            const int k = idx / np, r = idx % np;
            const int p = min(top[k], bot[k]), q = max(top[k], bot[k]);
            const double c = cs[k], s = sn[k];
            const double ap = A[p + (size_t)r * np], aq = A[q + (size_t)r * np];
            A[p + (size_t)r * np] = c * ap - s * aq;
            A[q + (size_t)r * np] = s * ap + c * aq;
         }
         __syncthreads();
         if (tid == 0) {            /* round robin: top[0] stays, the others move one seat */
            const int tlast = top[half - 1], b0 = bot[0];
            for (int i = half - 1; i > 1; i--) top[i] = top[i - 1];
            for (int i = 0; i < half - 1; i++) bot[i] = bot[i + 1];
            if (half > 1) { top[1] = b0; bot[half - 1] = tlast; }
         }
         __syncthreads();
      }
   }
   for (int i = tid; i < n; i += HIPK_BLOCK) evals[i] = A[i + (size_t)i * np];
   for (int idx = tid; idx < n * n; idx += HIPK_BLOCK) { const int i = idx % n, j = idx / n; Zout[idx] = V[i + (size_t)j * np]; }
   if (tid == 0) *sweeps_out = sweep;
}

extern "C" int hipk_sym_eig(hipk_ctx *ctx, int n, const double *A_host, int lda, double *evals_host,
      double *Z_host, int ldz) {
   if (n <= 0) return 0;
   if (n > EIG_MAXN) return -1;
   const int np = (n + 1) & ~1;
   double *buf = NULL;
   const size_t nn = (size_t)n * n;
   HIPK_CHECK(hipMalloc((void **)&buf, (2 * nn + n + 2) * sizeof(double)));
   double *dA = buf, *dZ = buf + nn, *dE = dZ + nn;
   int *dS = (int *)(dE + n);
   double *tight = (double *)malloc(nn * sizeof(double));
   if (!tight) { (void)hipFree(buf); return -2; }
   for (int j = 0; j < n; j++) for (int i = 0; i < n; i++) tight[i + (size_t)j * n] = (i <= j) ? A_host[i + (size_t)j * lda] : 0.0;
   hipError_t e = hipk_upload(ctx, dA, tight, nn * sizeof(double)) ? hipErrorUnknown : hipSuccess;
   if (e == hipSuccess) {
      hipLaunchKernelGGL(sym_eig_jacobi_kernel, dim3(1), dim3(HIPK_BLOCK), 2 * (size_t)np * np * sizeof(double), ctx->stream,
            dA, n, np, dE, dZ, dS);
      e = hipGetLastError();
   }
   double *ev = (double *)malloc(n * sizeof(double)), *Z = (double *)malloc(nn * sizeof(double));
   if (e == hipSuccess && hipk_download(ctx, ev, dE, n * sizeof(double))) e = hipErrorUnknown;
   if (e == hipSuccess && hipk_download(ctx, Z, dZ, nn * sizeof(double))) e = hipErrorUnknown;
   (void)hipFree(buf);
   free(tight);
   if (e != hipSuccess) { free(ev); free(Z); fprintf(stderr, "primme_amd: hipk_sym_eig: %s\n", hipGetErrorString(e)); return -1; }
   /* ascending order, like the host solver */
   int perm[EIG_MAXN];
   for (int i = 0; i < n; i++) perm[i] = i;
   for (int i = 1; i < n; i++) { const int pi = perm[i]; int j = i - 1; while (j >= 0 && ev[perm[j]] > ev[pi]) { perm[j + 1] = perm[j]; j--; } perm[j + 1] = pi; }
   for (int j = 0; j < n; j++) {
      evals_host[j] = ev[perm[j]];
      for (int i = 0; i < n; i++) Z_host[i + (size_t)j * ldz] = Z[i + (size_t)perm[j] * n];
   }
   free(ev); free(Z);
   return 0;
}
