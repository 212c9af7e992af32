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
 * One LANE owns a PAIR of consecutive rows (round 5: 16-byte accesses; a lane per row with 8-byte ones is 20-25 % slower
 * on this part whatever else is tuned, see pat_trip_inner): no tile record, no staging, no segmented sum, one dependent step
 * (pid -> gathers), and lanes of a wave that share a pattern gather consecutive entries of x (a coalesced 1 KB access per
 * table entry).  HBM traffic per product: m*(1 + 2s) (+ m*s for the fused form's second output) against
 * nnz*(s+2..4) + m*(4 + 2s).
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
/* PAIRS of consecutive rows per lane and trip / resident waves per SIMD the kernel is compiled for (build-time knobs for A/B
 * builds: scripts/build_variant.sh with -DHIPK_PAT_RPL=.. -DHIPK_PAT_WPS=..) */
#ifndef HIPK_PAT_RPL
#define HIPK_PAT_RPL 1
#endif
#define PAT_RPL_FOR(ML) (HIPK_PAT_RPL)
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
   int32_t minoff, maxoff;         /* smallest (<= 0) and largest (>= 0) column - row over all patterns */
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

/* buffer addressing: a resource descriptor in scalar registers (base, size) + a scalar byte offset + a 32-bit lane offset —
 * one instruction per access, no 64-bit address arithmetic on the vector unit; two consecutive elements per access */
typedef unsigned int pat_u2 __attribute__((ext_vector_type(2)));
typedef unsigned int pat_u4 __attribute__((ext_vector_type(4)));
template <typename T> __device__ __forceinline__ double pat_bload(__amdgpu_buffer_rsrc_t r, uint32_t vo, uint32_t so) {
   if constexpr (sizeof(T) == 8) return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, vo, so, 0));
   else return (double)__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, vo, so, 0));
}
template <typename T> __device__ __forceinline__ void pat_bload2(__amdgpu_buffer_rsrc_t r, uint32_t vo, uint32_t so, double &v0, double &v1) {
   if constexpr (sizeof(T) == 8) {
      const pat_u4 w = __builtin_amdgcn_raw_buffer_load_b128(r, vo, so, 0);
      const pat_u2 lo = {w.x, w.y}, hi = {w.z, w.w};
      v0 = __builtin_bit_cast(double, lo); v1 = __builtin_bit_cast(double, hi);
   } else {
      const pat_u2 w = __builtin_amdgcn_raw_buffer_load_b64(r, vo, so, 0);
      const unsigned int w0 = w.x, w1 = w.y;      /* (__builtin_bit_cast of a vector ELEMENT reads element 0 with this compiler: scalars first) */
      v0 = (double)__uint_as_float(w0); v1 = (double)__uint_as_float(w1);
   }
}
template <typename T> __device__ __forceinline__ void pat_bstore2(__amdgpu_buffer_rsrc_t r, uint32_t vo, uint32_t so, T v0, T v1) {
   if constexpr (sizeof(T) == 8) {
      const pat_u2 a = __builtin_bit_cast(pat_u2, v0), b = __builtin_bit_cast(pat_u2, v1);
      __builtin_amdgcn_raw_buffer_store_b128(pat_u4{a.x, a.y, b.x, b.y}, r, vo, so, 0);
   } else {
      __builtin_amdgcn_raw_buffer_store_b64(pat_u2{__builtin_bit_cast(unsigned int, v0), __builtin_bit_cast(unsigned int, v1)}, r, vo, so, 0);
   }
}

/* one trip of a lane over a chunk that lies INSIDE the slab with all its neighbours (no end-of-slab test, no halo): what all but
 * a handful of chunks are.  A lane owns PAIRS of consecutive rows: when both rows of a pair have the same pattern (all pairs but
 * the ones that straddle a change of pattern) every gather, the own entry and the stores are ONE 16-byte access per lane for
 * the two rows — the 5-point stencil written with 8-byte lane accesses runs at 0.6-0.7 of the rate of the same stencil with
 * 16-byte ones on this part (scripts/probes/stencil_probe.hip: 46-50 against 36 us at 10 M rows under this kernel's schedule;
 * a 160 MB copy takes 30), and neither the instruction count nor the occupancy nor the order of the loads moves the 8-byte
 * form (profiles/r05_spmv_row_pattern_pairs.md).  A pair of two patterns takes 8-byte accesses row by row.  Same products,
 * same order of additions within a row: y is bit-identical.
 *   rx: x from element (first row of the chunk + minoff) to its end (offsets are biased by -minoff, so every byte offset is
 *   unsigned, and the range check is exact);  ry / ro: y, xout from the chunk's first row;  rowb[u]: byte offset of the lane's
 *   u-th pair within the chunk */
template <typename T, int ML, int RPL, bool FUSED>
__device__ __forceinline__ void pat_trip_inner(const int (&p)[2 * RPL], const double *s_val, const int32_t *s_off, const int32_t *s_len,
      __amdgpu_buffer_rsrc_t rx, __amdgpu_buffer_rsrc_t ry, __amdgpu_buffer_rsrc_t ro, int32_t minoff, bool near,
      const uint32_t (&rowb)[RPL], double a, double &dotp) {
   constexpr int SH = sizeof(T) == 8 ? 3 : 2;
   const uint32_t ownb = (uint32_t)(-minoff) << SH;
   double xa[RPL][ML], xb[RPL][ML], oa[RPL], ob[RPL];
#pragma unroll
   for (int u = 0; u < RPL; u++) {
      const int pa = p[2 * u], pb = p[2 * u + 1];
      /* both rows with the first row's offsets: one access per entry for the pair (an entry past the end of x reads as 0: the
       * descriptor's range check) ... */
#pragma unroll
      for (int e = 0; e < ML; e++)
         pat_bload2<T>(rx, rowb[u] + ((uint32_t)(s_off[pa * ML + e] - minoff) << SH), 0, xa[u][e], xb[u][e]);
      if (FUSED) pat_bload2<T>(rx, rowb[u] + ownb, 0, oa[u], ob[u]);
      /* ... and where the second row has a pattern of its own (the few lanes whose pair straddles a change) its entries again.
       * With the first row's offsets the second element of such a pair's access can lie one past the end of x (the first row
       * references the last column), and what a partially out-of-range access returns for its in-range half is not something
       * to build on: within reach of the end of x (`near`, a handful of chunks) the first row is loaded again as well, entry by
       * entry.  (Two rows with the SAME pattern never overrun: the second row's reference is a column of the matrix.) */
      if (pa != pb) {
#pragma unroll
         for (int e = 0; e < ML; e++)
            xb[u][e] = pat_bload<T>(rx, rowb[u] + (uint32_t)sizeof(T) + ((uint32_t)(s_off[pb * ML + e] - minoff) << SH), 0);
         if (near) {
#pragma unroll
            for (int e = 0; e < ML; e++)
               xa[u][e] = pat_bload<T>(rx, rowb[u] + ((uint32_t)(s_off[pa * ML + e] - minoff) << SH), 0);
         }
      }
   }
#pragma unroll
   for (int u = 0; u < RPL; u++) {
      const int pa = p[2 * u], pb = p[2 * u + 1];
      const int la = s_len[pa], lb = s_len[pb];
      double sa = 0.0, sb = 0.0;
#pragma unroll
      for (int e = 0; e < ML; e++) {
         const double va = FUSED ? (double)(T)(a * xa[u][e]) : xa[u][e], vb = FUSED ? (double)(T)(a * xb[u][e]) : xb[u][e];
         const double ta = pat_mul_add(sa, s_val[pa * ML + e], va), tb = pat_mul_add(sb, s_val[pb * ML + e], vb);
         sa = e < la ? ta : sa;
         sb = e < lb ? tb : sb;
      }
      const T ya = (T)sa, yb = (T)sb;
      pat_bstore2<T>(ry, rowb[u], 0, ya, yb);
      if (FUSED) {
         const double wa = (double)(T)(a * oa[u]), wb = (double)(T)(a * ob[u]);
         pat_bstore2<T>(ro, rowb[u], 0, (T)wa, (T)wb);
         dotp = fma(wa, (double)ya, dotp);
         dotp = fma(wb, (double)yb, dotp);
      }
   }
}

/* the pattern numbers of a lane's RPL pairs in the chunk that starts at row r0: one 2-byte load per pair (pid is even-aligned at
 * even rows and one byte longer than the slab); a pair that reaches past the last row takes the last row's pattern for the
 * rows that do not exist (they are computed on the last row and not stored) */
template <int RPL>
__device__ __forceinline__ void pat_ids(const uint8_t *__restrict__ pid, int64_t r0, int64_t last, int (&pn)[2 * RPL]) {
#pragma unroll
   for (int u = 0; u < RPL; u++) {
      const int64_t r = r0 + 2 * (threadIdx.x + (int64_t)u * HIPK_BLOCK);
      if (r + 1 <= last) {
         const unsigned int w = __builtin_nontemporal_load((const uint16_t *)(pid + r));
         pn[2 * u] = (int)(w & 0xffu); pn[2 * u + 1] = (int)(w >> 8);
      } else {
         pn[2 * u] = (int)pid[r < last ? r : last]; pn[2 * u + 1] = (int)pid[last];
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
      int64_t halo_lo, const T *__restrict__ xlo, const T *__restrict__ xhi, const double *__restrict__ norm2, int np2,
      T *__restrict__ xout, double *__restrict__ partials, hipk_fin_args fa, int32_t minoff, int32_t maxoff) {
   extern __shared__ double pat_sh[];
   __shared__ int s_last;
   __shared__ double s_n2[4];
   double *s_val = pat_sh;                                     /* [npat * ML] */
   int32_t *s_off = (int32_t *)(pat_sh + (size_t)npat * ML);   /* [npat * ML] */
   int32_t *s_len = s_off + (size_t)npat * ML;                 /* [npat] */
   for (int i = threadIdx.x; i < npat * ML; i += HIPK_BLOCK) { s_val[i] = tval[i]; s_off[i] = toff[i]; }
   for (int i = threadIdx.x; i < npat; i += HIPK_BLOCK) s_len[i] = tlen[i];
   /* np2 > 0: norm2 points at the np2 partial sums of |t|^2 the Gram-Schmidt update left (hipk_tail_defer): this workgroup
    * adds them itself, in the order every other workgroup and hipk_tail_finish use — no second-stage launch in between */
   if (FUSED && np2 > 0) hipk_block_sum256_put(norm2, np2, s_n2);
   __syncthreads();
   const double a = (FUSED && norm2) ? 1.0 / sqrt(np2 > 0 ? hipk_block_sum256_get(s_n2) : norm2[0]) : 1.0;
   constexpr int NR = 2 * RPL;                                /* rows of a lane per trip: RPL pairs of consecutive rows */
   const int64_t CH = (int64_t)HIPK_BLOCK * NR;
   const int64_t nch = (nrows + CH - 1) / CH;
   const int64_t per = (nch + 7) >> 3;
   const int q = blockIdx.x & 7, j = blockIdx.x >> 3, J = gridDim.x >> 3;
   const int64_t c_lo = (int64_t)q * per, c_hi = c_lo + per < nch ? c_lo + per : nch;
   const int64_t last = nrows - 1;
   double dotp = 0.0;
   /* lane constants of the buffer-addressed form: byte offset of the lane's u-th pair within a chunk (constant over the trips);
    * pair u of lane t is rows 2 (t + 256 u), 2 (t + 256 u) + 1 of the chunk: a wave's pairs are 128 consecutive rows */
   uint32_t rowb[RPL];
#pragma unroll
   for (int u = 0; u < RPL; u++) rowb[u] = (uint32_t)(2 * (threadIdx.x + u * HIPK_BLOCK)) * (uint32_t)sizeof(T);
   int pn[NR];
   pat_ids<RPL>(pid, (c_lo + j < c_hi ? c_lo + j : 0) * CH, last, pn);
   for (int64_t c = c_lo + j; c < c_hi; c += J) {
      int p[NR];
#pragma unroll
      for (int v = 0; v < NR; v++) p[v] = pn[v];
      /* the next trip's patterns are on their way before this trip's gathers go out */
      if (c + J < c_hi) pat_ids<RPL>(pid, (c + J) * CH, last, pn);
      /* a chunk that lies inside the slab together with every entry its rows can reference (all but the ragged last one and,
       * in a row slab with halos, the few next to the slab's ends) takes the buffer-addressed form in pairs of rows */
      const int64_t r0 = c * CH;
      const bool inner = r0 + CH <= nrows && (!HALO || (r0 + minoff >= 0 && r0 + CH - 1 + maxoff < nrows));
      if (inner) {
         /* descriptors of this chunk (scalar registers): x from element r0 + minoff on — a base below x when the first rows
          * reference earlier ones, never dereferenced there — to the end of the slab, so that the range check is exact */
         const int64_t xbytes = (nrows - r0 - minoff) * (int64_t)sizeof(T), ybytes = (nrows - r0) * (int64_t)sizeof(T);
         const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)(x + r0 + minoff), 0, (int)(xbytes < 0x7fffffff ? xbytes : 0x7fffffff), 0x00020000);
         const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void *)(y + r0), 0, (int)(ybytes < 0x7fffffff ? ybytes : 0x7fffffff), 0x00020000);
         const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)((FUSED ? xout : y) + r0), 0, (int)(ybytes < 0x7fffffff ? ybytes : 0x7fffffff), 0x00020000);
         const bool near = r0 + CH + maxoff >= nrows;               /* some row of the chunk may reference the last column */
         pat_trip_inner<T, ML, RPL, FUSED>(p, s_val, s_off, s_len, rx, ry, ro, minoff, near, rowb, a, dotp);
      } else {
         int64_t r[NR];
#pragma unroll
         for (int v = 0; v < NR; v++) r[v] = r0 + 2 * (threadIdx.x + (int64_t)(v >> 1) * HIPK_BLOCK) + (v & 1);
         pat_trip<T, ML, NR, FUSED, HALO, true>(p, r, s_val, s_off, s_len, nrows, x, y, halo_lo, xlo, xhi, a, xout, dotp);
      }
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
   /* the kernel's 32-bit lane offsets: (row in chunk + column - row - minoff) * element size must stay below 2^31 */
   int64_t minoff = 0, maxoff = 0;
   for (const Pat &q : pats)
      for (int e = 0; e < q.len; e++) { if (q.off[e] < minoff) minoff = q.off[e]; if (q.off[e] > maxoff) maxoff = q.off[e]; }
   if ((maxoff - minoff + (int64_t)HIPK_BLOCK * 8) * (int64_t)es >= ((int64_t)1 << 31)) return 1;     /* (lane byte offsets of the buffer accesses within a chunk) */
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
   B->minoff = (int32_t)minoff; B->maxoff = (int32_t)maxoff;
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
   const int64_t nch = (B->nrows + (int64_t)2 * HIPK_BLOCK * rpl - 1) / ((int64_t)2 * HIPK_BLOCK * rpl);   /* RPL pairs of rows per lane and trip */
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
      const double *norm2, int np2, T *xout, double *partials, const hipk_fin_args &fa) {
   const size_t shm = (size_t)B->npat * B->ml * 12 + (size_t)B->npat * 4 + 8;
#define PATL(MLV) hipLaunchKernelGGL((pat_kernel<T, MLV, PAT_RPL_FOR(MLV), HIPK_PAT_WPS, FUSED, HALO>), dim3(gx), dim3(HIPK_BLOCK), shm, st, B->pid, B->toff, B->tval, B->tlen, B->npat, \
         B->nrows, x, y, halo_lo, xlo, xhi, norm2, np2, xout, partials, fa, B->minoff, B->maxoff)
   switch (B->ml) {
   case 3: PATL(3); break;
   case 5: PATL(5); break;
   case 7: PATL(7); break;
   default: PATL(8); break;
   }
#undef PATL
}

/* y = A x (xout == NULL) or the fused form y = A (a x), xout = a x, partials[workgroup] = its part of xout'y
 * (a = 1/sqrt(norm2[0]), or of the sum of the np2 partial sums norm2[0 .. np2) when np2 > 0; norm2 == NULL: a = 1).
 * gx = hipk_pat_grid().  xlo / xhi: halo rows below / above the slab. */
extern "C" int hipk_pat_matvec(const hipk_pat *B, void *hip_stream, int gx, const void *x, void *y, int64_t halo_lo, int64_t halo_hi,
      const void *xlo, const void *xhi, const double *norm2, int np2, void *xout, double *partials, const hipk_fin_args *fa_in) {
   hipStream_t st = (hipStream_t)hip_stream;
   const bool halo = halo_lo > 0 || halo_hi > 0;
   const bool fused = xout != NULL;
   hipk_fin_args fa;
   if (fa_in) fa = *fa_in; else memset(&fa, 0, sizeof(fa));
#define PATD(TT) do { \
      if (fused) { if (halo) pat_launch_ml<TT, true, true>(B, st, gx, (const TT *)x, (TT *)y, halo_lo, (const TT *)xlo, (const TT *)xhi, norm2, np2, (TT *)xout, partials, fa); \
                   else pat_launch_ml<TT, true, false>(B, st, gx, (const TT *)x, (TT *)y, halo_lo, (const TT *)xlo, (const TT *)xhi, norm2, np2, (TT *)xout, partials, fa); } \
      else { if (halo) pat_launch_ml<TT, false, true>(B, st, gx, (const TT *)x, (TT *)y, halo_lo, (const TT *)xlo, (const TT *)xhi, norm2, np2, (TT *)xout, partials, fa); \
             else pat_launch_ml<TT, false, false>(B, st, gx, (const TT *)x, (TT *)y, halo_lo, (const TT *)xlo, (const TT *)xhi, norm2, np2, (TT *)xout, partials, fa); } } while (0)
   if (B->dt == HIPK_F64) PATD(double); else PATD(float);
#undef PATD
   HIPK_CHECK(hipGetLastError());
   return 0;
}
