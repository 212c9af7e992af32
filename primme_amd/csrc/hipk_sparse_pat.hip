/* hipk_sparse_pat.hip — row-pattern dictionary form of a CSR matrix and its one-column products (gfx950).
 *
 * What it replaces: the same matrixMatvec callback as hipk_sparse.hip (reference examples/ex_eigs_dhipblas.c:239-264,
 * tests/COMMON/mat.c:64-90), for matrices whose rows repeat: constant-coefficient stencils and lattice operators
 * (the 5-/7-point Laplacians of BASELINE.json's north star and configs[1], tight-binding Hamiltonians, ...).
 *
 * Why: the tile kernel streams 10-12 bytes per nonzero and is bound by a chain of dependent round trips per tile
 * (tile record -> entries -> gathers -> LDS -> segmented sum): 0.56-0.59 of the HBM peak at 10 M rows and three tuning
 * experiments without effect.  The rows of such a matrix are a handful of distinct PATTERNS — a
 * row is its sequence of (column - row, value) pairs — so the matrix IS one byte per row plus a table:
 *   pid[r]            pattern of local row r                                   (1 byte per row, streamed once)
 *   tab[p] = {len, off[ML], val[ML]}   off = column - global row, val as double (<= 256 patterns, kept in LDS)
 * One LANE owns a row: no tile record, no staging, no segmented sum, one dependent step (pid -> gathers), and lanes of
 * a wave that share a pattern gather consecutive entries of x (a coalesced 512-byte access per table entry).  HBM
 * traffic per product: m*(1 + 2s) (+ m*s for the fused form's second output) against nnz*(s+2..4) + m*(4 + 2s).
 *
 * Arithmetic: exactly the tile kernel's — products rounded separately and added in CSR order, s = (s + v_j x_j) —
 * so y is BIT-IDENTICAL to csr_stream_kernel's; the fused form's t'At differs in the last bits (another fixed
 * summation order of the per-row terms).  Built by hipk_csr_create when the whole matrix has at most 256 patterns of at
 * most 8 entries (the scan gives up at the 257th pattern: a matrix that does not qualify costs a few hundred rows);
 * the CSR arrays stay for the block kernels.  HIPK_SPMV_PAT=0 / hipk_set_spmv_format(0) switch it off (A/B, tests).
 */
#include "hipk_internal.h"
#include <string>
#include <unordered_map>
#include <vector>

#define PAT_MAXLEN 8
#define PAT_MAXPAT 256
/* rows per lane and trip / resident waves per SIMD the kernel is compiled for, by table width (HIPK_PAT_RPL, a build-time
 * knob for A/B builds: scripts/build_variant.sh) */
#ifndef HIPK_PAT_RPL
#define HIPK_PAT_RPL 4
#endif
#define PAT_RPL_FOR(ML) ((ML) <= 5 ? HIPK_PAT_RPL : ((HIPK_PAT_RPL) > 2 ? (HIPK_PAT_RPL) / 2 : (HIPK_PAT_RPL)))
#ifndef HIPK_PAT_WPS
#define HIPK_PAT_WPS 6
#endif

struct hipk_pat {
   hipk_ctx *ctx;
   hipk_dtype dt;
   int64_t nrows;
   int npat, ml;                   /* patterns, table stride = longest row rounded up to a compiled width */
   uint8_t *pid;                   /* device [nrows + 1] */
   int32_t *toff;                  /* device [npat * ml] */
   double *tval;                   /* device [npat * ml] */
   int32_t *tlen;                  /* device [npat] */
};

/* s + v*x with the product and the sum rounded SEPARATELY (what the tile kernel does through its LDS staging): the default
 * contraction would make it one fma and y would differ from csr_stream_kernel's in the last bit */
__device__ __forceinline__ double pat_mul_add(double s, double v, double x) {
#pragma clang fp contract(off)
   const double p = v * x;
   return s + p;
}

/* the outputs: plain stores by default; HIPK_PAT_NT_STORES=1 (build-time, A/B builds) marks them non-temporal */
#ifndef HIPK_PAT_NT_STORES
#define HIPK_PAT_NT_STORES 0
#endif
template <typename T> __device__ __forceinline__ void pat_store(T *p, T v) {
   if (HIPK_PAT_NT_STORES) __builtin_nontemporal_store(v, p);
   else *p = v;
}

/* one trip of a lane: RPL rows, 256 apart.  Every gather of the trip is issued before the first product; GUARD: the chunk may
 * reach past the end of the slab — a row past the end works on the last row (not stored); an entry past a row's length gathers
 * the row's own x (table: offset 0, value 0) and is not added */
template <typename T, int ML, int RPL, bool FUSED, bool HALO, bool GUARD>
__device__ __forceinline__ void pat_trip(const int (&p)[RPL], const int64_t (&r)[RPL], const double *s_val, const int32_t *s_off,
      const int32_t *s_len, int64_t nrows, const T *__restrict__ x, T *__restrict__ y, int64_t halo_lo, const T *__restrict__ xlo,
      const T *__restrict__ xhi, double a, T *__restrict__ xout, double &dotp) {
   const int64_t last = nrows - 1;
   double xg[RPL][ML], xo[RPL];
#pragma unroll
   for (int u = 0; u < RPL; u++) {
      const int64_t rc = (GUARD && r[u] > last) ? last : r[u];
#pragma unroll
      for (int e = 0; e < ML; e++) {
         const int64_t l = rc + (int64_t)s_off[p[u] * ML + e];
         if (HALO) {
            const T *src = x + l;
            if (l < 0) src = xlo + (l + halo_lo);
            if (l >= nrows) src = xhi + (l - nrows);
            xg[u][e] = (double)*src;
         } else {
            xg[u][e] = (double)x[l];
         }
      }
      if (FUSED) xo[u] = (double)x[rc];
   }
#pragma unroll
   for (int u = 0; u < RPL; u++) {
      const int len = s_len[p[u]];
      double s = 0.0;
#pragma unroll
      for (int e = 0; e < ML; e++) {
         const double xv = FUSED ? (double)(T)(a * xg[u][e]) : xg[u][e];
         const double t = pat_mul_add(s, s_val[p[u] * ML + e], xv);
         s = e < len ? t : s;
      }
      if (!GUARD || r[u] < nrows) {
         const T yt = (T)s;
         pat_store(y + r[u], yt);
         if (FUSED) {
            const double xown = (double)(T)(a * xo[u]);
            pat_store(xout + r[u], (T)xown);
            dotp = fma(xown, (double)yt, dotp);
         }
      }
   }
}

/* XCD-aware PERSISTENT schedule: exactly as many workgroups as the chip holds at once (WPS per SIMD = WPS workgroups of four
 * waves per CU, enforced through __launch_bounds__; a grid larger than the resident set would run its tail after the
 * first workgroups have walked ALL their chunks), dealt round-robin to the 8 XCDs; XCD q owns a contiguous eighth of the
 * row chunks and its workgroups walk it with stride (workgroups per XCD), so the rows an XCD has in flight are one
 * contiguous window and the +-nx / +-plane neighbours of a stencil row are found in ITS L2.
 * RPL rows per lane and trip (256 apart: every access of a wave stays unit-stride): all their gathers are issued before
 * the first product — the bytes a wave keeps in flight are what bounds a kernel whose rows need 9 bytes from HBM. */
template <typename T, int ML, int RPL, int WPS, bool FUSED, bool HALO>
__global__ void __launch_bounds__(HIPK_BLOCK, WPS)
pat_kernel(const uint8_t *__restrict__ pid, const int32_t *__restrict__ toff, const double *__restrict__ tval,
      const int32_t *__restrict__ tlen, int npat, int64_t nrows, const T *__restrict__ x, T *__restrict__ y,
      int64_t halo_lo, const T *__restrict__ xlo, const T *__restrict__ xhi, const double *__restrict__ norm2,
      T *__restrict__ xout, double *__restrict__ partials, hipk_fin_args fa) {
   extern __shared__ double pat_sh[];
   __shared__ int s_last;
   double *s_val = pat_sh;                                     /* [npat * ML] */
   int32_t *s_off = (int32_t *)(pat_sh + (size_t)npat * ML);   /* [npat * ML] */
   int32_t *s_len = s_off + (size_t)npat * ML;                 /* [npat] */
   for (int i = threadIdx.x; i < npat * ML; i += HIPK_BLOCK) { s_val[i] = tval[i]; s_off[i] = toff[i]; }
   for (int i = threadIdx.x; i < npat; i += HIPK_BLOCK) s_len[i] = tlen[i];
   __syncthreads();
   const double a = (FUSED && norm2) ? 1.0 / sqrt(norm2[0]) : 1.0;
   const int64_t CH = (int64_t)HIPK_BLOCK * RPL;
   const int64_t nch = (nrows + CH - 1) / CH;
   const int64_t per = (nch + 7) >> 3;
   const int q = blockIdx.x & 7, j = blockIdx.x >> 3, J = gridDim.x >> 3;
   const int64_t c_lo = (int64_t)q * per, c_hi = c_lo + per < nch ? c_lo + per : nch;
   const int64_t last = nrows - 1;
   double dotp = 0.0;
   int pn[RPL];
   {
      const int64_t c = c_lo + j;
#pragma unroll
      for (int u = 0; u < RPL; u++) {
         const int64_t r = c * CH + threadIdx.x + (int64_t)u * HIPK_BLOCK;
         pn[u] = (c < c_hi) ? (int)__builtin_nontemporal_load(pid + (r < last ? r : last)) : 0;
      }
   }
   for (int64_t c = c_lo + j; c < c_hi; c += J) {
      int p[RPL];
      int64_t r[RPL];
#pragma unroll
      for (int u = 0; u < RPL; u++) { p[u] = pn[u]; r[u] = c * CH + threadIdx.x + (int64_t)u * HIPK_BLOCK; }
      /* the next trip's patterns are on their way before this trip's gathers go out */
      if (c + J < c_hi) {
#pragma unroll
         for (int u = 0; u < RPL; u++) {
            const int64_t rn = (c + J) * CH + threadIdx.x + (int64_t)u * HIPK_BLOCK;
            pn[u] = (int)__builtin_nontemporal_load(pid + (rn < last ? rn : last));
         }
      }
      /* a chunk that lies entirely inside the slab (all but the last one) takes the branch-free form: with the end-of-slab
       * tests in it the compiler sinks the gathers of the last row of every lane into the guarded store and waits for
       * each of them in turn (five full memory latencies per trip, 80 instead of 50 us at 10 M rows) */
      if ((c + 1) * CH <= nrows) pat_trip<T, ML, RPL, FUSED, HALO, false>(p, r, s_val, s_off, s_len, nrows, x, y, halo_lo, xlo, xhi, a, xout, dotp);
      else pat_trip<T, ML, RPL, FUSED, HALO, true>(p, r, s_val, s_off, s_len, nrows, x, y, halo_lo, xlo, xhi, a, xout, dotp);
   }
   if (FUSED) {
      __shared__ double red[HIPK_BLOCK / HIPK_WAVE];
      const double t = hipk_wave_sum(dotp);
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
      __syncthreads();
      if (threadIdx.x == 0) hipk_pstore(fa, partials + blockIdx.x, (red[0] + red[1]) + (red[2] + red[3]));
      hipk_inkernel_finalize(partials, 1, gridDim.x, fa, &s_last);
   }
}

/* ---- host side ---------------------------------------------------------------------------------------- */
extern "C" void hipk_pat_destroy(hipk_pat *B) {
   if (!B) return;
   if (B->pid) (void)hipFree(B->pid);
   if (B->toff) (void)hipFree(B->toff);
   if (B->tval) (void)hipFree(B->tval);
   if (B->tlen) (void)hipFree(B->tlen);
   free(B);
}

static int g_pat_mode = -1;        /* 1 = use the pattern form where it exists (default), 0 = never */
static int pat_mode(void) {
   if (g_pat_mode < 0) { const char *e = getenv("HIPK_SPMV_PAT"); g_pat_mode = e ? (atoi(e) != 0) : 1; }
   return g_pat_mode;
}
/* run-time switch for A/B tests (the environment variable is read once): returns the previous setting */
extern "C" int hipk_set_spmv_format(int use_patterns) { const int old = pat_mode(); g_pat_mode = use_patterns != 0; return old; }
extern "C" int hipk_pat_enabled(void) { return pat_mode(); }

static int pat_width(int maxlen) { return maxlen <= 3 ? 3 : maxlen <= 5 ? 5 : maxlen <= 7 ? 7 : 8; }

/* returns 0 and *out = the pattern form, 1 when the matrix does not qualify (*out = NULL), < 0 on errors.
 * rows [0, m) of a slab whose first global row is row0; columns are global */
extern "C" int hipk_pat_build(hipk_ctx *ctx, hipk_dtype dt, int64_t m, int64_t row0, const int32_t *rp, const int32_t *ci,
      const void *val, hipk_pat **out) {
   *out = NULL;
   if ((dt != HIPK_F64 && dt != HIPK_F32) || m <= 0) return 1;
   const size_t es = dt == HIPK_F64 ? 8 : 4;
   struct Pat { int len; int32_t off[PAT_MAXLEN]; double val[PAT_MAXLEN]; };
   std::vector<Pat> pats;
   std::unordered_map<std::string, int> index;
   std::vector<uint8_t> pid((size_t)m + 1, 0);
   int maxlen = 0, prev = -1;
   Pat cur;
   for (int64_t i = 0; i < m; i++) {
      const int len = rp[i + 1] - rp[i];
      if (len > PAT_MAXLEN) return 1;
      memset(&cur, 0, sizeof(cur));
      cur.len = len;
      for (int e = 0; e < len; e++) {
         const int64_t off = (int64_t)ci[rp[i] + e] - (row0 + i);
         if (off > INT32_MAX || off < INT32_MIN) return 1;
         cur.off[e] = (int32_t)off;
         cur.val[e] = dt == HIPK_F64 ? ((const double *)val)[rp[i] + e] : (double)((const float *)val)[rp[i] + e];
      }
      int id = -1;
      if (prev >= 0 && !memcmp(&pats[prev], &cur, sizeof(Pat))) id = prev;         /* the common case: same as the row above */
      else {
         const std::string key((const char *)&cur, sizeof(Pat));
         auto it = index.find(key);
         if (it != index.end()) id = it->second;
         else {
            if ((int)pats.size() >= PAT_MAXPAT) return 1;
            id = (int)pats.size();
            pats.push_back(cur);
            index.emplace(key, id);
         }
      }
      pid[i] = (uint8_t)id;
      prev = id;
      if (len > maxlen) maxlen = len;
   }
   (void)es;
   const int npat = (int)pats.size(), ml = pat_width(maxlen);
   std::vector<int32_t> toff((size_t)npat * ml, 0), tlen((size_t)npat, 0);
   std::vector<double> tval((size_t)npat * ml, 0.0);
   for (int p = 0; p < npat; p++) {
      tlen[p] = pats[p].len;
      for (int e = 0; e < pats[p].len; e++) { toff[(size_t)p * ml + e] = pats[p].off[e]; tval[(size_t)p * ml + e] = pats[p].val[e]; }
   }
   hipk_pat *B = (hipk_pat *)calloc(1, sizeof(hipk_pat));
   if (!B) return -2;
   B->ctx = ctx; B->dt = dt; B->nrows = m; B->npat = npat; B->ml = ml;
   if (hipk_malloc(ctx, pid.size(), (void **)&B->pid) || hipk_malloc(ctx, toff.size() * 4, (void **)&B->toff) ||
         hipk_malloc(ctx, tval.size() * 8, (void **)&B->tval) || hipk_malloc(ctx, tlen.size() * 4, (void **)&B->tlen)) { hipk_pat_destroy(B); return -2; }
   if (hipk_upload(ctx, B->pid, pid.data(), pid.size()) || hipk_upload(ctx, B->toff, toff.data(), toff.size() * 4) ||
         hipk_upload(ctx, B->tval, tval.data(), tval.size() * 8) || hipk_upload(ctx, B->tlen, tlen.data(), tlen.size() * 4)) { hipk_pat_destroy(B); return -1; }
   *out = B;
   return 0;
}

extern "C" int hipk_pat_npatterns(const hipk_pat *B) { return B ? B->npat : 0; }
/* workgroups of a launch (a multiple of 8: the XCD schedule); the fused form writes one partial sum per workgroup */
extern "C" int hipk_pat_grid(const hipk_pat *B, int num_cu) {
   const int rpl = PAT_RPL_FOR(B->ml);
   const int64_t nch = (B->nrows + (int64_t)HIPK_BLOCK * rpl - 1) / ((int64_t)HIPK_BLOCK * rpl);
   int64_t g = (int64_t)num_cu * HIPK_PAT_WPS;        /* the resident set: HIPK_PAT_WPS workgroups per CU (__launch_bounds__) */
   if (g > nch) g = nch;
   g = g / 8 * 8;                                     /* a multiple of 8 that does not exceed it: the XCD schedule */
   return (int)(g < 8 ? 8 : g);
}
/* bytes one product moves through HBM: the pattern bytes, x once, y (and the second output of the fused form) */
extern "C" double hipk_pat_bytes(const hipk_pat *B, int fused) {
   const double es = B->dt == HIPK_F64 ? 8 : 4;
   return (double)B->nrows * (1.0 + (fused ? 3.0 : 2.0) * es);
}

template <typename T, bool FUSED, bool HALO>
static void pat_launch_ml(const hipk_pat *B, hipStream_t st, int gx, const T *x, T *y, int64_t halo_lo, const T *xlo, const T *xhi,
      const double *norm2, T *xout, double *partials, const hipk_fin_args &fa) {
   const size_t shm = (size_t)B->npat * B->ml * 12 + (size_t)B->npat * 4 + 8;
#define PATL(MLV) hipLaunchKernelGGL((pat_kernel<T, MLV, PAT_RPL_FOR(MLV), HIPK_PAT_WPS, FUSED, HALO>), dim3(gx), dim3(HIPK_BLOCK), shm, st, B->pid, B->toff, B->tval, B->tlen, B->npat, \
         B->nrows, x, y, halo_lo, xlo, xhi, norm2, xout, partials, fa)
   switch (B->ml) {
   case 3: PATL(3); break;
   case 5: PATL(5); break;
   case 7: PATL(7); break;
   default: PATL(8); break;
   }
#undef PATL
}

/* y = A x (xout == NULL) or the fused form y = A (a x), xout = a x, partials[workgroup] = its part of xout'y
 * (a = 1/sqrt(norm2[0]), norm2 == NULL: a = 1).  gx = hipk_pat_grid().  xlo / xhi: halo rows below / above the slab. */
extern "C" int hipk_pat_matvec(const hipk_pat *B, void *hip_stream, int gx, const void *x, void *y, int64_t halo_lo, int64_t halo_hi,
      const void *xlo, const void *xhi, const double *norm2, void *xout, double *partials, const hipk_fin_args *fa_in) {
   hipStream_t st = (hipStream_t)hip_stream;
   const bool halo = halo_lo > 0 || halo_hi > 0;
   const bool fused = xout != NULL;
   hipk_fin_args fa;
   if (fa_in) fa = *fa_in; else memset(&fa, 0, sizeof(fa));
#define PATD(TT) do { \
      if (fused) { if (halo) pat_launch_ml<TT, true, true>(B, st, gx, (const TT *)x, (TT *)y, halo_lo, (const TT *)xlo, (const TT *)xhi, norm2, (TT *)xout, partials, fa); \
                   else pat_launch_ml<TT, true, false>(B, st, gx, (const TT *)x, (TT *)y, halo_lo, (const TT *)xlo, (const TT *)xhi, norm2, (TT *)xout, partials, fa); } \
      else { if (halo) pat_launch_ml<TT, false, true>(B, st, gx, (const TT *)x, (TT *)y, halo_lo, (const TT *)xlo, (const TT *)xhi, norm2, (TT *)xout, partials, fa); \
             else pat_launch_ml<TT, false, false>(B, st, gx, (const TT *)x, (TT *)y, halo_lo, (const TT *)xlo, (const TT *)xhi, norm2, (TT *)xout, partials, fa); } } while (0)
   if (B->dt == HIPK_F64) PATD(double); else PATD(float);
#undef PATD
   HIPK_CHECK(hipGetLastError());
   return 0;
}
