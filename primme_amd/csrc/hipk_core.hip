/* hipk_core.hip — context, HBM/pinned memory, stream-ordered copies, the
 * deterministic second stage of every reduction, and the bandwidth probe.
 * See include/primme_amd_kernels.h for the reference routines each entry replaces. */
#include "hipk_internal.h"

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
   *out = ctx;
   return 0;
}

extern "C" int hipk_ctx_destroy(hipk_ctx *ctx) {
   if (!ctx) return 0;
   hipStreamSynchronize(ctx->stream);
   if (ctx->partials) hipFree(ctx->partials);
   if (ctx->jobtab) hipFree(ctx->jobtab);
   hipEventDestroy(ctx->ev0);
   hipEventDestroy(ctx->ev1);
   if (ctx->own_stream) hipStreamDestroy(ctx->stream);
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

extern "C" int hipk_malloc(hipk_ctx *ctx, size_t bytes, void **dptr) {
   (void)ctx;
   if (bytes == 0) bytes = 8;
   hipError_t e = hipMalloc(dptr, bytes);
   if (e != hipSuccess) { *dptr = NULL; return -2; }
   return 0;
}
extern "C" int hipk_free(hipk_ctx *ctx, void *dptr) {
   (void)ctx;
   if (dptr) HIPK_CHECK(hipFree(dptr));
   return 0;
}
extern "C" int hipk_host_alloc(hipk_ctx *ctx, size_t bytes, void **hptr) {
   (void)ctx;
   if (bytes == 0) bytes = 8;
   hipError_t e = hipHostMalloc(hptr, bytes, hipHostMallocDefault);
   if (e != hipSuccess) { *hptr = NULL; return -2; }
   return 0;
}
extern "C" int hipk_host_free(hipk_ctx *ctx, void *hptr) {
   (void)ctx;
   if (hptr) HIPK_CHECK(hipHostFree(hptr));
   return 0;
}
extern "C" int hipk_h2d(hipk_ctx *ctx, void *dst, const void *src, size_t bytes) {
   if (bytes) HIPK_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
   return 0;
}
extern "C" int hipk_d2h(hipk_ctx *ctx, void *dst, const void *src, size_t bytes) {
   if (bytes) HIPK_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
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
__global__ void __launch_bounds__(HIPK_BLOCK)
hipk_finalize_kernel(const double *__restrict__ partials, int nblocks, int nout,
      double *__restrict__ out, double *__restrict__ out_host) {
   __shared__ double sm[HIPK_BLOCK / HIPK_WAVE];
   const int o = blockIdx.x;
   double s = 0.0;
   for (int b = threadIdx.x; b < nblocks; b += HIPK_BLOCK) s += partials[(size_t)b * nout + o];
   s = hipk_wave_sum(s);
   if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
   __syncthreads();
   if (threadIdx.x == 0) {
      const double v = (sm[0] + sm[1]) + (sm[2] + sm[3]);
      out[o] = v;
      if (out_host) out_host[o] = v;
   }
}

int hipk_finalize_partials(hipk_ctx *ctx, const double *partials, int nblocks, int nout,
      double *out_dev) {
   if (nout <= 0) return 0;
   hipLaunchKernelGGL(hipk_finalize_kernel, dim3(nout), dim3(HIPK_BLOCK), 0, ctx->stream,
         partials, nblocks, nout, out_dev, hipk_mirror_of(ctx, out_dev));
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

/* ---- live kernel-class profiler ---------------------------------------------- */
#define PROF_RING 4096
static struct {
   int enabled, inited;
   hipEvent_t e0[PROF_RING], e1[PROF_RING];
   int cls[PROF_RING];
   int head, tail;                  /* slots [tail, head) are in flight */
   double ms[HIPK_PROF_NCLASS], bytes[HIPK_PROF_NCLASS];
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

int hipk_prof_begin(int cls, hipStream_t st, double alg_bytes) {
   if (!g_prof.enabled) return -1;
   if (g_prof.head - g_prof.tail >= PROF_RING - 1) prof_drain(0);
   if (g_prof.head - g_prof.tail >= PROF_RING - 1) prof_drain(1);
   const int s = g_prof.head % PROF_RING;
   g_prof.cls[s] = cls;
   g_prof.bytes[cls] += alg_bytes;
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
   for (int c = 0; c < HIPK_PROF_NCLASS; c++) { g_prof.ms[c] = 0; g_prof.bytes[c] = 0; g_prof.launches[c] = 0; }
   return 0;
}
/* cls: 0 dots (TN panel), 1 project (NN accumulate), 2 ritz (fused update), 3 spmv */
extern "C" int hipk_prof_get(int cls, double *ms, long *launches, double *alg_bytes) {
   if (cls < 0 || cls >= HIPK_PROF_NCLASS) return -1;
   if (g_prof.inited) prof_drain(1);
   *ms = g_prof.ms[cls]; *launches = g_prof.launches[cls]; *alg_bytes = g_prof.bytes[cls];
   return 0;
}
