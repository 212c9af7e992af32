/* hipk_sparse_pb.hip — panel-blocked sparse matrix-vector product for matrices whose column pattern has no
 * locality (BASELINE configs[4]: A 8M x 2M with 5 nonzeros per row at (i p_q + q) mod n, and its transpose).
 *
 * What it replaces: the matrixMatvec callback of the singular value problem (reference
 * src/svds/primme_svds_c.c:1323-1379 calls it for A x and A'u; tests/COMMON/mat.c:64-90 is the CSR loop).
 *
 * Why: with plain CSR every nonzero gathers one 8-byte entry of x out of a vector that is larger than an XCD's
 * 4 MB L2 (x = 16 MB, u = 64 MB): each gather pulls a whole cache line over the fabric, 2.0 GB fetched for 0.59 GB
 * of algorithmic bytes (profiles/r03_pmc_config5_spmv_fetch.md), 0.11-0.15 of the HBM roofline.
 *
 * How: the columns are cut into PANELS whose slice of x fits the L2 (2 MB by default).  One WAVE owns a tile of RT
 * consecutive rows for the whole launch and walks the panels in order — all waves of the chip are started together
 * and do about the same work per panel, so the chip as a whole sweeps x panel by panel and the gathers hit the L2;
 * the tile's row sums live in the wave's slice of LDS across the panels, so y is written exactly once and never
 * read (a launch per panel would read and write y P times).  Storage: for every (tile, panel) the entries of the
 * tile's rows that fall into the panel, row by row, as {value, packed (row in tile << CB | column in panel)}: 8 + 4
 * bytes per entry, streamed once, perfectly coalesced, with non-temporal loads — no row pointers, no counts.  A lane
 * takes entry e, gathers x, and adds the product to its row's LDS slot with ds_add_f64; only this wave touches the
 * slice, its instructions execute in order, and entries of one row sit next to each other.
 * Bytes per product: nnz*(s+4) + m*s (+ x once per XCD out of HBM, the rest out of the L2).
 */
#include "hipk_internal.h"
#include <algorithm>
#include <vector>

struct hipk_pb {
   hipk_ctx *ctx;
   hipk_dtype dt;
   int64_t nrows, ncols, nnz;
   int P, RT, ntiles, cb;        /* panels, rows per tile (one wave), tiles, column bits of the packed word */
   int64_t W;                    /* columns per panel = 1 << cb */
   uint32_t *ptr;                /* device [ntiles*P + 1] */
   void *val;                    /* device [nnz + pad] */
   uint32_t *pk;                 /* device [nnz + pad]: row in tile << cb | column in panel */
};

template <typename T> __device__ __forceinline__ T pb_ldnt(const T *p) { return __builtin_nontemporal_load(p); }

/* PB_U: entries a lane has in flight per step (HIPK_PB_U = 2 | 4 | 8, measurement knob).  The ds_add_f64 is not what
 * bounds the kernel (a plain read-modify-write in its place: 308 vs 321 us), nor is the depth (8: 300 / 338 us for A x / A'u,
 * 4: 321 / 325, 2: 361 / 364; profiles/r04_config5_panel_blocked.txt): 40 M gathers that hit the L2 in ~300 us are ~130 G
 * requests/s, about half of what the chip's 128 L2 channels accept — the request rate of the L2, not HBM, is the bound. */
template <typename T, int PB_U>
__global__ void __launch_bounds__(HIPK_BLOCK)
pb_matvec_kernel(const uint32_t *__restrict__ ptr, const T *__restrict__ val, const uint32_t *__restrict__ pk,
      const T *__restrict__ x, T *__restrict__ y, int64_t m, int64_t n, int P, int ntiles, int RT, int cb) {
   extern __shared__ double pb_sy[];
   const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
   const int t = (int)blockIdx.x * (HIPK_BLOCK / 64) + wv;
   if (t >= ntiles) return;
   double *sy = pb_sy + (size_t)wv * RT;
   for (int r = lane; r < RT; r += 64) sy[r] = 0.0;
   const uint32_t cmask = (1u << cb) - 1u;
   const uint32_t *tp = ptr + (size_t)t * P;
   uint32_t e0 = tp[0];
   for (int p = 0; p < P; p++) {
      const uint32_t e1 = tp[p + 1];
      const T *xb = x + ((size_t)p << cb);
      for (uint32_t e = e0 + lane; e < e1; e += 64 * PB_U) {
         T v[PB_U];
         uint32_t w[PB_U];
#pragma unroll
         for (int u = 0; u < PB_U; u++) {
            const uint32_t q = e + 64u * u < e1 ? e + 64u * u : e1 - 1u;      /* clamped, not predicated: the loads go out together */
            v[u] = pb_ldnt(val + q);
            w[u] = pb_ldnt(pk + q);
         }
         T xv[PB_U];
#pragma unroll
         for (int u = 0; u < PB_U; u++) xv[u] = xb[w[u] & cmask];
#pragma unroll
         for (int u = 0; u < PB_U; u++)
            if (e + 64u * u < e1) unsafeAtomicAdd(sy + (w[u] >> cb), (double)v[u] * (double)xv[u]);
      }
      e0 = e1;
   }
   const int64_t r0 = (int64_t)t * RT;
   for (int r = lane; r < RT; r += 64)
      if (r0 + r < m) y[r0 + r] = (T)sy[r];
}

/* ---- host side ---------------------------------------------------------------------------------------- */
static size_t pb_es(hipk_dtype dt) { return dt == HIPK_F64 ? 8 : 4; }

/* is the column pattern scattered?  distinct 128-byte lines of x per nonzero over sampled groups of 64 rows */
static double pb_scatter(int64_t m, const int32_t *rp, const int32_t *ci, size_t es) {
   const int64_t groups = (m + 63) / 64;
   const int64_t step = groups > 4096 ? groups / 4096 : 1;
   std::vector<int64_t> lines;
   double nnz = 0, distinct = 0;
   for (int64_t g = 0; g < groups; g += step) {
      const int64_t r0 = g * 64, r1 = r0 + 64 < m ? r0 + 64 : m;
      lines.clear();
      for (int32_t q = rp[r0]; q < rp[r1]; q++) lines.push_back((int64_t)ci[q] * (int64_t)es / 128);
      nnz += (double)lines.size();
      std::sort(lines.begin(), lines.end());
      distinct += (double)(std::unique(lines.begin(), lines.end()) - lines.begin());
   }
   return nnz > 0 ? distinct / nnz : 0.0;
}

extern "C" void hipk_pb_destroy(hipk_pb *B) {
   if (!B) return;
   if (B->ptr) (void)hipFree(B->ptr);
   if (B->val) (void)hipFree(B->val);
   if (B->pk) (void)hipFree(B->pk);
   free(B);
}

/* returns 0 and *out = the panel-blocked form, or 1 (not worthwhile / not representable: *out = NULL), or < 0 */
extern "C" int hipk_pb_build(hipk_ctx *ctx, hipk_dtype dt, int64_t m, int64_t n, const int32_t *rp, const int32_t *ci,
      const void *val, hipk_pb **out) {
   *out = NULL;
   if (dt != HIPK_F64 && dt != HIPK_F32) return 1;
   static int mode = -2;           /* HIPK_PB = 0 never, 1 whenever representable, unset: by the pattern */
   static long panel_kb = 0;       /* HIPK_PB_KB: bytes of x per panel, in KB, a power of two (default 2048) */
   static long tile_entries = 0;   /* HIPK_PB_TILE: entries per (tile, panel) step to aim for (default 640) */
   if (mode == -2) {
      const char *e = getenv("HIPK_PB"); mode = e ? atoi(e) : -1;
      const char *k = getenv("HIPK_PB_KB"); panel_kb = k && atol(k) > 0 ? atol(k) : 2048;
      const char *te = getenv("HIPK_PB_TILE"); tile_entries = te && atol(te) > 0 ? atol(te) : 640;
   }
   if (mode == 0 || m <= 0 || n <= 0) return 1;
   const size_t es = pb_es(dt);
   const int64_t nnz = rp[m];
   if (nnz <= 0) return 1;
   if (mode < 0) {
      /* worthwhile when x does not fit an XCD's L2 and consecutive rows do not share its cache lines */
      if ((double)n * es < 6.0 * 1024 * 1024 || nnz < (int64_t)1 << 20) return 1;
      if (pb_scatter(m, rp, ci, es) < 0.4) return 1;
   }
   int cb = 10;
   while (((int64_t)1 << cb) * (int64_t)es < (int64_t)panel_kb * 1024) cb++;
   int64_t W = (int64_t)1 << cb;
   int P = (int)((n + W - 1) / W);
   while (P > 256 && cb < 24) { cb++; W = (int64_t)1 << cb; P = (int)((n + W - 1) / W); }
   /* rows per tile: about tile_entries entries per (tile, panel) step, a power of two that the packed word and the LDS hold */
   const double per_row_panel = (double)nnz / (double)m / (double)P;
   int RT = 64;
   /* (2048 rows: the four waves of a workgroup then hold 64 KB of row sums in LDS, the most a launch gets without asking) */
   while (RT < 2048 && RT < (1 << (32 - cb)) / 2 && RT * per_row_panel < (double)tile_entries) RT *= 2;
   if (RT > (1 << (32 - cb))) return 1;
   const int64_t ntiles = (m + RT - 1) / RT;
   if (ntiles * P + 1 > ((int64_t)1 << 31)) return 1;

   std::vector<uint32_t> ptr((size_t)ntiles * P + 1, 0);
   for (int64_t i = 0; i < m; i++) {
      const int64_t t = i / RT;
      for (int32_t q = rp[i]; q < rp[i + 1]; q++) {
         if (ci[q] < 0 || ci[q] >= n) return 1;
         ptr[(size_t)t * P + (size_t)(ci[q] >> cb) + 1]++;
      }
   }
   for (size_t tp = 0; tp < (size_t)ntiles * P; tp++) ptr[tp + 1] += ptr[tp];
   if ((int64_t)ptr[(size_t)ntiles * P] != nnz) return -1;
   std::vector<char> pv((size_t)(nnz + 1) * es, 0);
   std::vector<uint32_t> pc((size_t)nnz + 1, 0);
   std::vector<uint32_t> cursor((size_t)P);
   for (int64_t t = 0; t < ntiles; t++) {
      for (int p = 0; p < P; p++) cursor[p] = ptr[(size_t)t * P + p];
      const int64_t r1 = (t + 1) * RT < m ? (t + 1) * RT : m;
      for (int64_t i = t * RT; i < r1; i++)
         for (int32_t q = rp[i]; q < rp[i + 1]; q++) {
            const int p = (int)(ci[q] >> cb);
            const uint32_t pos = cursor[p]++;
            memcpy(&pv[(size_t)pos * es], (const char *)val + (size_t)q * es, es);
            pc[pos] = ((uint32_t)(i - t * RT) << cb) | ((uint32_t)ci[q] & (uint32_t)(W - 1));
         }
   }
   hipk_pb *B = (hipk_pb *)calloc(1, sizeof(hipk_pb));
   if (!B) return -2;
   B->ctx = ctx; B->dt = dt; B->nrows = m; B->ncols = n; B->nnz = nnz; B->P = P; B->RT = RT; B->ntiles = (int)ntiles; B->W = W; B->cb = cb;
   if (hipk_malloc(ctx, ptr.size() * 4, (void **)&B->ptr) || hipk_malloc(ctx, pv.size(), &B->val) ||
         hipk_malloc(ctx, pc.size() * 4, (void **)&B->pk)) { hipk_pb_destroy(B); return -2; }
   if (hipk_upload(ctx, B->ptr, ptr.data(), ptr.size() * 4) || hipk_upload(ctx, B->val, pv.data(), pv.size()) ||
         hipk_upload(ctx, B->pk, pc.data(), pc.size() * 4)) { hipk_pb_destroy(B); return -1; }
   *out = B;
   return 0;
}

template <typename T>
static int pb_launch(const hipk_pb *B, hipStream_t st, const T *x, T *y) {
   const dim3 g((unsigned)((B->ntiles + 3) / 4)), b(HIPK_BLOCK);
   static int uu = -1;
   if (uu < 0) { const char *e = getenv("HIPK_PB_U"); uu = e ? atoi(e) : 4; }
#define PBL(U) hipLaunchKernelGGL((pb_matvec_kernel<T, U>), g, b, (size_t)4 * B->RT * sizeof(double), st, B->ptr, (const T *)B->val, B->pk, x, y, B->nrows, \
         B->ncols, B->P, B->ntiles, B->RT, B->cb)
   if (uu == 8) PBL(8); else if (uu == 2) PBL(2); else PBL(4);
#undef PBL
   HIPK_CHECK(hipGetLastError());
   return 0;
}

/* y(:,c) = A x(:,c), one pass over the matrix per column */
extern "C" int hipk_pb_matvec(const hipk_pb *B, void *hip_stream, const void *x, int64_t ldx, void *y, int64_t ldy, int ncols) {
   hipStream_t st = (hipStream_t)hip_stream;
   for (int c = 0; c < ncols; c++) {
      const int rc = B->dt == HIPK_F64 ? pb_launch<double>(B, st, (const double *)x + (size_t)c * ldx, (double *)y + (size_t)c * ldy)
                                       : pb_launch<float>(B, st, (const float *)x + (size_t)c * ldx, (float *)y + (size_t)c * ldy);
      if (rc) return rc;
   }
   return 0;
}
/* bytes one product streams: entries + counts + the output (x comes out of the L2 / Infinity Cache) */
extern "C" double hipk_pb_bytes(const hipk_pb *B) {
   const double es = (double)pb_es(B->dt);
   return (double)B->nnz * (es + 4) + (double)B->ntiles * B->P * 4.0 + (double)B->nrows * es + (double)B->ncols * es;
}
extern "C" int hipk_pb_panels(const hipk_pb *B) { return B ? B->P : 0; }
