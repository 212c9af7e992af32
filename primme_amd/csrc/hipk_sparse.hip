/* hipk_sparse.hip — the user matvec on the device: CSR SpMV/SpMM (LDS-staged
 * row tiles, "CSR-stream"), a wave-per-row path for long rows, a matrix-free
 * Laplacian stencil, and the Jacobi preconditioner.
 *
 * Replaces the hipsparseSpMM callback of reference examples/ex_eigs_dhipblas.c:239-264
 * (which re-queries buffer sizes and rebuilds dense descriptors on every call) and
 * the SPARSKIT amux loop of tests/COMMON/mat.c:64-90.
 *
 * CSR-stream: the host bins consecutive rows into tiles holding <= TILE_NNZ
 * nonzeros and <= 256 rows.  A workgroup streams its tile's (value, column)
 * pairs with fully coalesced loads, multiplies by the gathered x entries, parks
 * the products in LDS, and then one lane per row adds that row's LDS segment.
 * HBM traffic is the algorithmic minimum nnz*(s+4) + rows*(4 + 2s); the x gather
 * is served by L2 (tiles are assigned to XCDs in contiguous row ranges so that
 * neighbouring tiles share their x window in one L2).
 */
#include "hipk_internal.h"
#include <vector>
#include <limits.h>

#define TILE_NNZ 2048
#define TILE_ROWS 256

struct hipk_csr {
   hipk_ctx *ctx;
   hipk_dtype dt;
   int kind;               /* 0 = CSR, 1 = stencil */
   int64_t nrows, ncols_global, row0, nnz;
   int64_t x0, xlen;       /* entries [x0, x0+xlen) of the input vector are owned (= the row slab for
                              square row-partitioned operators, everything for rectangular ones) */
   int32_t *rowptr, *colind;   /* device */
   void *values;               /* device */
   int32_t *tiles;             /* device: ntiles+1 row offsets */
   int4 *tileinfo;             /* device: {first row, end row, first nonzero, end nonzero} per tile */
   int2 *twin;                 /* device: {first column, window width} per tile; width 0 = not windowable */
   uint16_t *col16;            /* device, or NULL: column - (row0 + first row of the tile - c16back) for every nonzero, when that
                                  fits 16 bits for the whole matrix (banded / stencil / block-diagonal patterns): 2 bytes per
                                  nonzero out of HBM instead of 4 in the tile kernels, the base follows from the tile record */
   int64_t c16back;            /* how far below its first row a tile's columns reach, at most */
   int windowed;               /* most tiles have a narrow column window inside the owned slab */
   int cw_max;                 /* widest window among the windowable tiles */
   int ntiles;
   void *diag;                 /* device, nrows elements */
   int64_t halo_lo, halo_hi;   /* extent of off-rank columns below / above */
   const void *xlo, *xhi;      /* device halo buffers for the current matvec */
   int64_t ld_lo, ld_hi;       /* their column strides (default: halo_lo / halo_hi, packed) */
   int sx, sy, sz;             /* stencil grid */
   struct hipk_pb *pb;         /* panel-blocked form (hipk_sparse_pb.hip) for scattered column patterns, or NULL */
   struct hipk_pat *pat;       /* row-pattern dictionary form (hipk_sparse_pat.hip) for matrices whose rows repeat, or NULL */
};
struct hipk_pb;
struct hipk_pat;
extern "C" int hipk_pat_build(hipk_ctx *ctx, hipk_dtype dt, int64_t m, int64_t row0, const int32_t *rp, const int32_t *ci, const void *val, hipk_pat **out);
extern "C" int hipk_pat_matvec(const hipk_pat *B, void *hip_stream, int gx, const void *x, void *y, int64_t halo_lo, int64_t halo_hi, const void *xlo,
      const void *xhi, const double *norm2, int np2, void *xout, double *partials, const hipk_fin_args *fa);
extern "C" void hipk_pat_destroy(hipk_pat *B);
extern "C" int hipk_pat_grid(const hipk_pat *B, int num_cu);
extern "C" double hipk_pat_bytes(const hipk_pat *B, int fused);
extern "C" int hipk_pat_npatterns(const hipk_pat *B);
extern "C" int hipk_pat_enabled(void);
extern "C" int hipk_pb_build(hipk_ctx *ctx, hipk_dtype dt, int64_t m, int64_t n, const int32_t *rp, const int32_t *ci, const void *val, hipk_pb **out);
extern "C" int hipk_pb_matvec(const hipk_pb *B, void *hip_stream, const void *x, int64_t ldx, void *y, int64_t ldy, int ncols);
extern "C" void hipk_pb_destroy(hipk_pb *B);
extern "C" double hipk_pb_bytes(const hipk_pb *B);
extern "C" int hipk_pb_panels(const hipk_pb *B);

/* x element for global column g: owned slab, or the lo / hi halo buffers.  The address is
 * selected, the load itself is unconditional (a branch per gather would serialise the gathers). */
template <typename T>
__device__ __forceinline__ double fetch_x(const T *__restrict__ x, const T *__restrict__ xlo,
      const T *__restrict__ xhi, int64_t row0, int64_t nrows, int64_t halo_lo, int64_t g) {
   const int64_t l = g - row0;
   const T *p = x + l;
   if (l < 0) p = xlo + (l + halo_lo);
   if (l >= nrows) p = xhi + (l - nrows);
   return (double)*p;
}

/* XCD-aware tile order: block b -> tile ((b % 8) * per + b / 8) so that each XCD
 * (blocks are dealt round-robin to the 8 XCDs) owns a contiguous range of tiles. */
__device__ __forceinline__ int xcd_tile(int b, int ntiles) {
   const int per = (ntiles + 7) >> 3;
   return (b & 7) * per + (b >> 3);
}

#define TILE_PER_LANE (TILE_NNZ / HIPK_BLOCK)   /* nonzeros a lane handles per tile */

/* CSR-stream, one column at a time.  The chain of dependent memory round trips per tile is what
 * bounds this kernel, not the bytes: {r0,r1,p0,p1} come in one 16-byte load, a lane then issues
 * ALL of its (value, column) loads of the tile before the first gather (indices clamped instead
 * of predicated, so nothing branches around a load), all gathers before the first product, and
 * the row pointers of the summing phase are fetched before the barrier.
 * FUSED (one column): the input is the un-normalised new basis vector t and |t|^2 sits in HBM
 * (norm2[0]); a = 1/sqrt(|t|^2) is applied to every gathered entry on the fly, the normalised
 * vector is written to xout for the rows this tile owns (xout != x: other tiles still gather from
 * x), and the tile's part of xout' y goes to partials[blockIdx.x].  Replaces the separate
 * normalisation pass and the two-vector inner product t'At of the one-synchronisation GD
 * iteration (eigs_conv.c); same arithmetic per element as scale_rsqrt_kernel + this kernel. */
/* NTM: the (value, index) stream with the non-temporal hint — for matrices too large to stay in the 256 MiB Infinity
 * Cache between two products (the launcher decides by the size of the stream); smaller ones are left to be cached. */
template <typename T, bool FUSED, bool C16, bool NTM>
__global__ void __launch_bounds__(HIPK_BLOCK)
csr_stream_kernel(const int4 *__restrict__ tileinfo, int ntiles, const int32_t *__restrict__ rowptr,
      const int32_t *__restrict__ colind, const uint16_t *__restrict__ col16, int64_t c16off,
      const T *__restrict__ val, const T *__restrict__ x,
      int64_t ldx, T *__restrict__ y, int64_t ldy, int ncols, int64_t row0, int64_t nrows,
      int64_t halo_lo, int64_t halo_hi, const T *__restrict__ xlo, const T *__restrict__ xhi,
      int64_t ld_lo, int64_t ld_hi, const double *__restrict__ norm2, T *__restrict__ xout,
      double *__restrict__ partials, hipk_fin_args fa) {
   __shared__ double prod[TILE_NNZ];
   __shared__ int s_last;
   const int tile = xcd_tile(blockIdx.x, ntiles);
   double dotp = 0.0;
   if (tile < ntiles) {
      const int4 ti = tileinfo[tile];
      const int r0 = ti.x, r1 = ti.y, p0 = ti.z, nz = ti.w - ti.z;
      const bool local = (halo_lo == 0 && halo_hi == 0);
      const double a = (FUSED && norm2) ? 1.0 / sqrt(norm2[0]) : 1.0;   /* norm2 == NULL: no scaling (xout = x) */
      if (nz <= TILE_NNZ) {
         /* row segment of this lane's row: fetched now, used after the barrier */
         const int r = r0 + threadIdx.x;
         const int rc = r < r1 ? r : r1 - 1;
         const int sa = rowptr[rc] - p0, sb = rowptr[rc + 1] - p0;
         /* the tile's (value, column) pairs: all loads in flight at once */
         double v[TILE_PER_LANE];
         int32_t cidx[TILE_PER_LANE];
#pragma unroll
         for (int u = 0; u < TILE_PER_LANE; u++) {
            const int q = threadIdx.x + u * HIPK_BLOCK;
            const int qc = q < nz ? q : (nz > 0 ? nz - 1 : 0);   /* an all-empty tile reads the padded element */
            v[u] = (double)(NTM ? __builtin_nontemporal_load(val + p0 + qc) : val[p0 + qc]);
            /* C16 (a template parameter: exactly one index stream is loaded): global column = row0 + r0 - c16back + entry;
             * an empty tile reads a neighbour's entry against its own base, so its (unused) gather goes to a valid row */
            if (C16) cidx[u] = (int32_t)(NTM ? __builtin_nontemporal_load(col16 + p0 + qc) : col16[p0 + qc]);   /* raw: any arithmetic here would make the batch wait load by load */
            else cidx[u] = NTM ? __builtin_nontemporal_load(colind + p0 + qc) : colind[p0 + qc];
         }
         if (C16) {
            const int32_t cbase = (int32_t)(c16off + r0);
#pragma unroll
            for (int u = 0; u < TILE_PER_LANE; u++) cidx[u] = nz > 0 ? cbase + cidx[u] : (int32_t)row0;
         }
         for (int c = 0; c < ncols; c++) {
            const T *xc = x + (size_t)c * ldx;
            const T *xloc = xlo ? xlo + (size_t)c * ld_lo : xc;
            const T *xhic = xhi ? xhi + (size_t)c * ld_hi : xc;
            double xg[TILE_PER_LANE];
            if (local) {
#pragma unroll
               for (int u = 0; u < TILE_PER_LANE; u++) xg[u] = (double)xc[(int64_t)cidx[u] - row0];
            } else {
#pragma unroll
               for (int u = 0; u < TILE_PER_LANE; u++) xg[u] = fetch_x<T>(xc, xloc, xhic, row0, nrows, halo_lo, (int64_t)cidx[u]);
            }
            double xown = 0.0;
            if (FUSED && r < r1) xown = (double)(T)(a * (double)xc[r]);    /* rows are local entries: r indexes the slab */
#pragma unroll
            for (int u = 0; u < TILE_PER_LANE; u++) {
               const int q = threadIdx.x + u * HIPK_BLOCK;
               const double xv = FUSED ? (double)(T)(a * xg[u]) : xg[u];
               if (q < nz) prod[q] = v[u] * xv;
            }
            __syncthreads();
            if (r < r1) {
               double s = 0.0;
               for (int q = sa; q < sb; q++) s += prod[q];
               const T yt = (T)s;
               y[r + (size_t)c * ldy] = yt;
               if (FUSED) { xout[r] = (T)xown; dotp = fma(xown, (double)yt, dotp); }
            }
            if (c + 1 < ncols) __syncthreads();
         }
      } else {
         /* a tile that is one long row: the whole workgroup reduces it */
         for (int c = 0; c < ncols; c++) {
            const T *xc = x + (size_t)c * ldx;
            const T *xloc = xlo ? xlo + (size_t)c * ld_lo : xc;
            const T *xhic = xhi ? xhi + (size_t)c * ld_hi : xc;
            for (int r = r0; r < r1; r++) {
               const int qa = rowptr[r], qb = rowptr[r + 1];
               double s = 0.0;
               for (int q = qa + threadIdx.x; q < qb; q += HIPK_BLOCK) {
                  const double xg = fetch_x<T>(xc, xloc, xhic, row0, nrows, halo_lo, (int64_t)colind[q]);
                  s = fma((double)val[q], FUSED ? (double)(T)(a * xg) : xg, s);
               }
               s = hipk_wave_sum(s);
               if ((threadIdx.x & 63) == 0) prod[threadIdx.x >> 6] = s;
               __syncthreads();
               if (threadIdx.x == 0) {
                  const T yt = (T)((prod[0] + prod[1]) + (prod[2] + prod[3]));
                  y[r + (size_t)c * ldy] = yt;
                  if (FUSED) { const T xo = (T)(a * (double)xc[r]); xout[r] = xo; dotp = fma((double)xo, (double)yt, dotp); }
               }
               __syncthreads();
            }
         }
      }
   }
   if (FUSED) {
      __shared__ double red[HIPK_BLOCK / HIPK_WAVE];
      __syncthreads();
      const double t = hipk_wave_sum(dotp);
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
      __syncthreads();
      if (threadIdx.x == 0) hipk_pstore(fa, partials + blockIdx.x, (red[0] + red[1]) + (red[2] + red[3]));
      hipk_inkernel_finalize(partials, 1, gridDim.x, fa, &s_last);
   }
}

/* Block of vectors (the JDQMR / block Davidson SpMM).  The tile's (value, column) pairs are
 * staged ONCE in LDS with coalesced loads (one barrier per tile, not two per column); then one
 * lane per (row, group of NC columns) walks its row in LDS and gathers x for its NC columns with
 * independent accumulators, so a lane keeps ~NC x row-length gathers in flight and nothing is
 * re-read from HBM for the second and later columns. */
template <typename T, int NC>
__global__ void __launch_bounds__(HIPK_BLOCK)
csr_rows_block_kernel(const int32_t *__restrict__ tiles, int ntiles, const int32_t *__restrict__ rowptr,
      const int32_t *__restrict__ colind, const T *__restrict__ val, const T *__restrict__ x,
      int64_t ldx, T *__restrict__ y, int64_t ldy, int ncols, int64_t row0, int64_t nrows,
      int64_t halo_lo, int64_t halo_hi, const T *__restrict__ xlo, const T *__restrict__ xhi,
      int64_t ld_lo, int64_t ld_hi) {
   __shared__ T sval[TILE_NNZ];
   __shared__ int32_t scol[TILE_NNZ];
   __shared__ int rp[TILE_ROWS + 1];
   const int tile = xcd_tile(blockIdx.x, ntiles);
   if (tile >= ntiles) return;
   const int r0 = tiles[tile], r1 = tiles[tile + 1];
   const int p0 = rowptr[r0], p1 = rowptr[r1];
   const int nz = p1 - p0, nr = r1 - r0;
   const bool local = (halo_lo == 0 && halo_hi == 0);

   if (nz <= TILE_NNZ) {
      {  /* all of this lane's (value, column) loads in flight at once: indices clamped, not predicated */
         T tv[TILE_PER_LANE];
         int32_t tc[TILE_PER_LANE];
#pragma unroll
         for (int u = 0; u < TILE_PER_LANE; u++) {
            const int q = threadIdx.x + u * HIPK_BLOCK;
            const int qc = q < nz ? q : (nz > 0 ? nz - 1 : 0);
            tv[u] = val[p0 + qc]; tc[u] = colind[p0 + qc];
         }
         const int rr = threadIdx.x <= nr ? threadIdx.x : nr;
         const int rpv = rowptr[r0 + rr] - p0;
#pragma unroll
         for (int u = 0; u < TILE_PER_LANE; u++) {
            const int q = threadIdx.x + u * HIPK_BLOCK;
            if (q < nz) { sval[q] = tv[u]; scol[q] = tc[u]; }
         }
         if (threadIdx.x <= nr) rp[threadIdx.x] = rpv;
         if (threadIdx.x == 0 && nr == TILE_ROWS) rp[TILE_ROWS] = rowptr[r0 + TILE_ROWS] - p0;
      }
      __syncthreads();
      const int ngroups = (ncols + NC - 1) / NC;
      for (int idx = threadIdx.x; idx < nr * ngroups; idx += HIPK_BLOCK) {
         const int g = idx / nr, r = idx - g * nr, c0 = g * NC;
         double acc[NC];
#pragma unroll
         for (int c = 0; c < NC; c++) acc[c] = 0.0;
         const int qa = rp[r], qb = rp[r + 1];
         if (local) {
            /* columns past ncols alias the last valid one (computed, never stored): no branches
             * in the gather loop; 4 row entries per trip = 4*NC independent gathers in flight */
            const T *xg[NC];
#pragma unroll
            for (int c = 0; c < NC; c++) xg[c] = x + (size_t)((c0 + c < ncols) ? c0 + c : ncols - 1) * ldx - row0;
            const int64_t self = row0;   /* any owned entry: multiplied by zero */
            for (int q = qa; q < qb; q += 4) {
               double v[4];
               int64_t gc[4];
#pragma unroll
               for (int u = 0; u < 4; u++) {
                  const bool ok = q + u < qb;
                  v[u] = ok ? (double)sval[ok ? q + u : qa] : 0.0;
                  gc[u] = ok ? (int64_t)scol[ok ? q + u : qa] : self;
               }
#pragma unroll
               for (int u = 0; u < 4; u++)
#pragma unroll
                  for (int c = 0; c < NC; c++) acc[c] = fma(v[u], (double)xg[c][gc[u]], acc[c]);
            }
         } else {
            for (int q = qa; q < qb; q++) {
               const double v = (double)sval[q];
               const int64_t gcol = scol[q];
#pragma unroll
               for (int c = 0; c < NC; c++)
                  if (c0 + c < ncols)
                     acc[c] = fma(v, fetch_x<T>(x + (size_t)(c0 + c) * ldx, xlo ? xlo + (size_t)(c0 + c) * ld_lo : x,
                                                 xhi ? xhi + (size_t)(c0 + c) * ld_hi : x, row0, nrows, halo_lo, gcol), acc[c]);
            }
         }
#pragma unroll
         for (int c = 0; c < NC; c++)
            if (c0 + c < ncols) y[r0 + r + (size_t)(c0 + c) * ldy] = (T)acc[c];
      }
   } else {
      __shared__ double red[HIPK_BLOCK / HIPK_WAVE];
      for (int c = 0; c < ncols; c++) {
         const T *xc = x + (size_t)c * ldx;
         const T *xloc = xlo ? xlo + (size_t)c * ld_lo : xc;
         const T *xhic = xhi ? xhi + (size_t)c * ld_hi : xc;
         for (int r = r0; r < r1; r++) {
            const int a = rowptr[r], b = rowptr[r + 1];
            double sum = 0.0;
            for (int q = a + threadIdx.x; q < b; q += HIPK_BLOCK)
               sum = fma((double)val[q], fetch_x<T>(xc, xloc, xhic, row0, nrows, halo_lo, (int64_t)colind[q]), sum);
            sum = hipk_wave_sum(sum);
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
            __syncthreads();
            if (threadIdx.x == 0) y[r + (size_t)c * ldy] = (T)((red[0] + red[1]) + (red[2] + red[3]));
            __syncthreads();
         }
      }
   }
}

/* The same with the x WINDOW of the tile staged in LDS.  For banded, block-diagonal and 2-D
 * stencil matrices the column indices of a tile's nonzeros span a few hundred rows of x: that slice
 * of the block of vectors ([cmin, cmin + cw) x ncols, at most XS_MAX values) is fetched from HBM
 * with coalesced loads — every x entry the tile needs comes in ONCE per tile instead of once per
 * nonzero through a scattered 8-byte L2 access — and the row walk gathers from LDS.  Tiles whose
 * window is too wide or reaches outside the owned slab (twin.y == 0) use the global gathers.
 * shift != NULL: y = A x - shift[c] x(:,c), the first update of the projected operator in the
 * JDQMR inner iteration (reference inner_solve.c:853-858) fused into the operator. */
#define XS_MAX 3072
#define XS_SMALL 1728      /* 192 window rows of 8 columns on the odd stride of 9: 39 KB of LDS per workgroup, still 4 per CU (1536 until round 6) */
struct SpmmShift { double s[64]; int on; int nosplit; int evenstride; };
template <typename T, int NC, int XS, bool C16>
__global__ void __launch_bounds__(HIPK_BLOCK)
csr_window_block_kernel(const int4 *__restrict__ tileinfo, const int2 *__restrict__ twin, int ntiles,
      const int32_t *__restrict__ rowptr, const int32_t *__restrict__ colind, const uint16_t *__restrict__ col16, int64_t c16off,
      const T *__restrict__ val,
      const T *__restrict__ x, int64_t ldx, T *__restrict__ y, int64_t ldy, int ncols, int64_t row0,
      SpmmShift sh) {
   __shared__ T sval[TILE_NNZ];
   __shared__ int32_t scol[TILE_NNZ];
   __shared__ int rp[TILE_ROWS + 1];
   __shared__ double xs[XS];                 /* xs[w * ncols + c] = x(cmin + w, c) */
   const int tile = xcd_tile(blockIdx.x, ntiles);
   if (tile >= ntiles) return;
   const int4 ti = tileinfo[tile];
   const int2 tw = twin[tile];
   const int r0 = ti.x, r1 = ti.y, p0 = ti.z, nz = ti.w - ti.z, nr = r1 - r0;
   const int cmin = tw.x, cw = tw.y;
   /* LDS rows of the window are `xst` doubles apart, xst ODD: the lanes of a wave walk different rows of the matrix and read
    * the window at unrelated row numbers — with a stride of ncols = 8 doubles (64 bytes = 16 banks) only four bank groups
    * exist and a 32-lane read serialises up to eight-fold; with 9 the 32 lanes hit 32 distinct bank pairs (round 6:
    * profiles/r06_spmm_window_stride.txt) */
   const int xst = sh.evenstride ? ncols : (ncols | 1);
   const bool win = cw > 0 && cw * xst <= XS && nz <= TILE_NNZ;
   if (nz <= TILE_NNZ) {
      {  /* every load of the tile — (value, column) pairs, row pointers, the x window — is issued before
          * the first LDS store: indices clamped, not predicated, so nothing branches around a load */
         T tv[TILE_PER_LANE];
         int32_t tc[TILE_PER_LANE];
         T xw[(XS + HIPK_BLOCK - 1) / HIPK_BLOCK];
#pragma unroll
         for (int u = 0; u < TILE_PER_LANE; u++) {
            const int q = threadIdx.x + u * HIPK_BLOCK;
            const int qc = q < nz ? q : (nz > 0 ? nz - 1 : 0);
            tv[u] = val[p0 + qc];
            if (C16) tc[u] = (int32_t)col16[p0 + qc];            /* raw: the base is added when the entry goes to LDS */
            else tc[u] = colind[p0 + qc];
         }
         const int rr = threadIdx.x <= nr ? threadIdx.x : nr;
         const int rpv = rowptr[r0 + rr] - p0;
         const int nxw = win ? cw * ncols : 0;
         if (win) {
#pragma unroll
            for (int u = 0; u < (XS + HIPK_BLOCK - 1) / HIPK_BLOCK; u++) {
               const int idx = threadIdx.x + u * HIPK_BLOCK;
               const int ic = idx < nxw ? idx : nxw - 1;
               const int c = ic / cw, w = ic - c * cw;          /* coalesced along the window for each column */
               xw[u] = x[(int64_t)cmin - row0 + w + (size_t)c * ldx];
            }
         }
         const int32_t cadj = (C16 ? (int32_t)(c16off + r0) : 0) - (win ? cmin : 0);   /* stored column: global, or window-relative */
#pragma unroll
         for (int u = 0; u < TILE_PER_LANE; u++) {
            const int q = threadIdx.x + u * HIPK_BLOCK;
            if (q < nz) { sval[q] = tv[u]; scol[q] = tc[u] + cadj; }
         }
         if (threadIdx.x <= nr) rp[threadIdx.x] = rpv;
         if (threadIdx.x == 0 && nr == TILE_ROWS) rp[TILE_ROWS] = rowptr[r0 + TILE_ROWS] - p0;
         if (win) {
#pragma unroll
            for (int u = 0; u < (XS + HIPK_BLOCK - 1) / HIPK_BLOCK; u++) {
               const int idx = threadIdx.x + u * HIPK_BLOCK;
               if (idx < nxw) { const int c = idx / cw, w = idx - c * cw; xs[w * xst + c] = (double)xw[u]; }   /* LDS rows are xst apart */
            }
         }
      }
      __syncthreads();
      const int ngroups = (ncols + NC - 1) / NC;
      /* A tile of 2048 nonzeros has ~120 rows of 17: with one lane per (row, column group) half of the workgroup idles
       * through the row walk, a chain of dependent LDS reads.  When the work fits twice, two adjacent lanes share a row:
       * each walks half of its nonzeros and the halves meet in a shuffle (fixed order: low half + high half). */
      const int split = (2 * nr * ngroups <= HIPK_BLOCK && !sh.nosplit) ? 2 : 1;
      for (int idx0 = threadIdx.x; idx0 < nr * ngroups * split; idx0 += HIPK_BLOCK) {
         const int idx = idx0 / split, half = idx0 - idx * split;
         const int g = idx / nr, r = idx - g * nr, c0 = g * NC;
         double acc[NC];
#pragma unroll
         for (int c = 0; c < NC; c++) acc[c] = 0.0;
         int qa = rp[r], qb = rp[r + 1];
         if (split == 2) { const int qm = qa + (qb - qa + 1) / 2; if (half) qa = qm; else qb = qm; }
         if (win) {
            int cofs[NC];
#pragma unroll
            for (int c = 0; c < NC; c++) cofs[c] = (c0 + c < ncols) ? c0 + c : ncols - 1;
            for (int q = qa; q < qb; q++) {
               const double v = (double)sval[q];
               const double *xr = xs + scol[q] * xst;
#pragma unroll
               for (int c = 0; c < NC; c++) acc[c] = fma(v, xr[cofs[c]], acc[c]);
            }
         } else {
            const T *xg[NC];
#pragma unroll
            for (int c = 0; c < NC; c++) xg[c] = x + (size_t)((c0 + c < ncols) ? c0 + c : ncols - 1) * ldx - row0;
            const int64_t self = row0;
            for (int q = qa; q < qb; q += 4) {
               double v[4];
               int64_t gc[4];
#pragma unroll
               for (int u = 0; u < 4; u++) {
                  const bool ok = q + u < qb;
                  v[u] = ok ? (double)sval[ok ? q + u : qa] : 0.0;
                  gc[u] = ok ? (int64_t)scol[ok ? q + u : qa] : self;
               }
#pragma unroll
               for (int u = 0; u < 4; u++)
#pragma unroll
                  for (int c = 0; c < NC; c++) acc[c] = fma(v[u], (double)xg[c][gc[u]], acc[c]);
            }
         }
         if (split == 2) {
            /* (2 nr ngroups <= 256: one trip, both lanes of a pair are in it) */
#pragma unroll
            for (int c = 0; c < NC; c++) { const double o = __shfl_xor(acc[c], 1); acc[c] = half ? o + acc[c] : acc[c] + o; }
            if (half) continue;
         }
#pragma unroll
         for (int c = 0; c < NC; c++)
            if (c0 + c < ncols) {
               double out = acc[c];
               if (sh.on) out = fma(-sh.s[c0 + c], (double)x[(int64_t)(r0 + r) + (size_t)(c0 + c) * ldx], out);
               y[r0 + r + (size_t)(c0 + c) * ldy] = (T)out;
            }
      }
   } else {
      __shared__ double red[HIPK_BLOCK / HIPK_WAVE];
      for (int c = 0; c < ncols; c++) {
         const T *xc = x + (size_t)c * ldx - row0;
         for (int r = r0; r < r1; r++) {
            const int a = rowptr[r], b = rowptr[r + 1];
            double sum = 0.0;
            for (int q = a + threadIdx.x; q < b; q += HIPK_BLOCK) sum = fma((double)val[q], (double)xc[colind[q]], sum);
            sum = hipk_wave_sum(sum);
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
            __syncthreads();
            if (threadIdx.x == 0) {
               double out = (red[0] + red[1]) + (red[2] + red[3]);
               if (sh.on) out = fma(-sh.s[c], (double)xc[row0 + r], out);
               y[r + (size_t)c * ldy] = (T)out;
            }
            __syncthreads();
         }
      }
   }
}

/* Laplacian stencil: diag 2*dims, -1 to each grid neighbour, Dirichlet boundary. */
template <typename T>
__global__ void __launch_bounds__(HIPK_BLOCK)
stencil_kernel(int sx, int sy, int sz, int64_t row0, int64_t nrows, const T *__restrict__ x,
      int64_t ldx, T *__restrict__ y, int64_t ldy, int ncols, int64_t halo_lo, int64_t halo_hi,
      const T *__restrict__ xlo, const T *__restrict__ xhi, int64_t ld_lo, int64_t ld_hi) {
   const int64_t stride = (int64_t)gridDim.x * HIPK_BLOCK;
   const int64_t plane = (int64_t)sx * sy;
   const double dg = (sz > 1) ? 6.0 : (sy > 1 ? 4.0 : 2.0);
   for (int c = 0; c < ncols; c++) {
      const T *xc = x + (size_t)c * ldx;
      const T *xloc = xlo ? xlo + (size_t)c * ld_lo : xc;
      const T *xhic = xhi ? xhi + (size_t)c * ld_hi : xc;
      T *yc = y + (size_t)c * ldy;
      for (int64_t l = (int64_t)blockIdx.x * HIPK_BLOCK + threadIdx.x; l < nrows; l += stride) {
         const int64_t g = row0 + l;
         const int ix = (int)(g % sx);
         const int iy = (int)((g / sx) % sy);
         const int iz = (int)(g / plane);
         double s = dg * (double)xc[l];
#define NB(gn) fetch_x<T>(xc, xloc, xhic, row0, nrows, halo_lo, (gn))
         if (ix > 0) s -= NB(g - 1);
         if (ix < sx - 1) s -= NB(g + 1);
         if (sy > 1) {
            if (iy > 0) s -= NB(g - sx);
            if (iy < sy - 1) s -= NB(g + sx);
         }
         if (sz > 1) {
            if (iz > 0) s -= NB(g - plane);
            if (iz < sz - 1) s -= NB(g + plane);
         }
#undef NB
         yc[l] = (T)s;
      }
   }
}

template <typename T>
__global__ void __launch_bounds__(HIPK_BLOCK)
fill_kernel(T *__restrict__ d, int64_t n, double v) {
   const int64_t stride = (int64_t)gridDim.x * HIPK_BLOCK;
   for (int64_t i = (int64_t)blockIdx.x * HIPK_BLOCK + threadIdx.x; i < n; i += stride) d[i] = (T)v;
}

struct JacShift { double s[64]; };
template <typename T>
__global__ void __launch_bounds__(HIPK_BLOCK)
jacobi_kernel(const T *__restrict__ diag, JacShift sh, double min_den, const T *__restrict__ x, int64_t ldx,
      T *__restrict__ y, int64_t ldy, int ncols, int64_t m) {
   const int64_t stride = (int64_t)gridDim.x * HIPK_BLOCK;
   for (int c = 0; c < ncols; c++) {
      const double shift = sh.s[c];
      for (int64_t i = (int64_t)blockIdx.x * HIPK_BLOCK + threadIdx.x; i < m; i += stride) {
         double d = (double)diag[i] - shift;
         /* same guard as the reference's test preconditioner: avoid dividing by ~0 */
         if (!(fabs(d) > min_den)) d = copysign(min_den, d);
         y[i + (size_t)c * ldy] = (T)((double)x[i + (size_t)c * ldx] / d);
      }
   }
}

static size_t elem_size(hipk_dtype dt) {
   return dt == HIPK_F64 ? 8 : dt == HIPK_F32 ? 4 : dt == HIPK_C64 ? 16 : 8;
}

static int csr_create_impl(hipk_ctx *ctx, hipk_dtype dt, int64_t nrows_local,
      int64_t ncols_global, int64_t row0, int64_t x0, int64_t xlen, const int32_t *rowptr_host,
      const int32_t *colind_host, const void *values_host, hipk_csr **out) {
   if (dt != HIPK_F64 && dt != HIPK_F32 && dt != HIPK_C64 && dt != HIPK_C32) return -44;
   if (nrows_local >= ((int64_t)1 << 31)) return -1;
   hipk_csr *A = (hipk_csr *)calloc(1, sizeof(hipk_csr));
   if (!A) return -2;
   A->ctx = ctx; A->dt = dt; A->kind = 0;
   A->nrows = nrows_local; A->ncols_global = ncols_global; A->row0 = row0;
   A->x0 = x0; A->xlen = xlen;
   const int64_t nnz = rowptr_host[nrows_local];
   A->nnz = nnz;
   const size_t es = elem_size(dt);

   /* tiles + halo extent + diagonal on the host (one pass) */
   std::vector<int32_t> tiles;
   tiles.push_back(0);
   int64_t lo = 0, hi = 0;
   std::vector<char> dg((size_t)nrows_local * es, 0);
   int r = 0;
   while (r < nrows_local) {
      int rs = r;
      int64_t cnt = 0;
      while (r < nrows_local && (r - rs) < TILE_ROWS) {
         int64_t rn = rowptr_host[r + 1] - rowptr_host[r];
         if (cnt + rn > TILE_NNZ && r > rs) break;
         cnt += rn;
         r++;
         if (cnt > TILE_NNZ) break; /* a single long row forms its own tile */
      }
      tiles.push_back(r);
   }
   for (int64_t i = 0; i < nrows_local; i++)
      for (int32_t p = rowptr_host[i]; p < rowptr_host[i + 1]; p++) {
         int64_t g = colind_host[p];
         if (g < x0 && x0 - g > lo) lo = x0 - g;
         if (g >= x0 + xlen && g - (x0 + xlen) + 1 > hi) hi = g - (x0 + xlen) + 1;
         if (g == row0 + i) memcpy(&dg[(size_t)i * es], (const char *)values_host + (size_t)p * es, es);
      }
   A->halo_lo = lo; A->halo_hi = hi; A->ld_lo = lo; A->ld_hi = hi;
   A->ntiles = (int)tiles.size() - 1;
   std::vector<int4> tinfo((size_t)A->ntiles + 1);
   std::vector<int2> twin((size_t)A->ntiles + 1);
   int64_t narrow = 0;
   A->cw_max = 0;
   for (int t = 0; t < A->ntiles; t++) {
      tinfo[t] = make_int4(tiles[t], tiles[t + 1], rowptr_host[tiles[t]], rowptr_host[tiles[t + 1]]);
      int64_t cmin = INT64_MAX, cmax = -1;
      for (int32_t p = rowptr_host[tiles[t]]; p < rowptr_host[tiles[t + 1]]; p++) {
         const int64_t g = colind_host[p];
         if (g < cmin) cmin = g;
         if (g > cmax) cmax = g;
      }
      /* the shifted product also reads x at the tile's own rows: they are local by construction */
      const bool inside = cmax >= cmin && cmin >= x0 && cmax < x0 + xlen && cmax - cmin + 1 <= XS_MAX;
      twin[t] = inside ? make_int2((int)cmin, (int)(cmax - cmin + 1)) : make_int2(0, 0);
      if (inside && (int)(cmax - cmin + 1) > A->cw_max) A->cw_max = (int)(cmax - cmin + 1);
      if (inside && (cmax - cmin + 1) * 8 <= XS_MAX) narrow++;
   }
   A->windowed = (lo == 0 && hi == 0 && A->ntiles > 0 && narrow * 10 >= (int64_t)A->ntiles * 9);
   /* A second, 2-byte index stream when the pattern allows it: entry = column - (row0 + first row of the tile - c16back),
    * c16back = the farthest any tile reaches below its first row.  10 instead of 12 bytes per nonzero out of HBM for a
    * double matrix; the tile kernels rebuild the column from the tile record they load anyway. */
   std::vector<uint16_t> c16;
   A->c16back = 0;
   if ((dt == HIPK_F64 || dt == HIPK_F32) && A->ntiles > 0) {
      int64_t back = 0, reach = 0;
      std::vector<int64_t> tmin((size_t)A->ntiles), tmax((size_t)A->ntiles);
      for (int t = 0; t < A->ntiles; t++) {
         int64_t cmin = INT64_MAX, cmax = -1;
         for (int32_t p = rowptr_host[tiles[t]]; p < rowptr_host[tiles[t + 1]]; p++) {
            const int64_t g = colind_host[p];
            if (g < cmin) cmin = g;
            if (g > cmax) cmax = g;
         }
         tmin[t] = cmin; tmax[t] = cmax;
         if (cmax >= cmin && row0 + tiles[t] - cmin > back) back = row0 + tiles[t] - cmin;
      }
      for (int t = 0; t < A->ntiles; t++)
         if (tmax[t] >= tmin[t] && tmax[t] - (row0 + tiles[t] - back) > reach) reach = tmax[t] - (row0 + tiles[t] - back);
      if (reach <= 65535 && row0 - back > (int64_t)INT32_MIN / 2 && ncols_global < (int64_t)INT32_MAX) {
         A->c16back = back;
         c16.assign((size_t)nnz + 1, 0);
         for (int t = 0; t < A->ntiles; t++)
            for (int32_t p = rowptr_host[tiles[t]]; p < rowptr_host[tiles[t + 1]]; p++)
               c16[p] = (uint16_t)(colind_host[p] - (row0 + tiles[t] - back));
      }
   }

   if (hipk_malloc(ctx, (size_t)(nrows_local + 1) * 4, (void **)&A->rowptr) ||
         hipk_malloc(ctx, (size_t)(nnz + 1) * 4, (void **)&A->colind) ||       /* +1: clamped loads of an empty tile */
         hipk_malloc(ctx, (size_t)(nnz + 1) * es, &A->values) ||
         hipk_malloc(ctx, tiles.size() * 4, (void **)&A->tiles) ||
         hipk_malloc(ctx, tinfo.size() * sizeof(int4), (void **)&A->tileinfo) ||
         hipk_malloc(ctx, twin.size() * sizeof(int2), (void **)&A->twin) ||
         hipk_malloc(ctx, (size_t)nrows_local * es, &A->diag) ||
         (!c16.empty() && hipk_malloc(ctx, c16.size() * sizeof(uint16_t), (void **)&A->col16)))
      return -2;
   /* Every upload goes through pinned staging on the context's stream (hipk_upload) and is complete on return: nothing
    * goes through the NULL stream, which a hipStreamNonBlocking stream is not ordered against, and the runtime is never
    * asked to copy asynchronously out of the caller's pageable arrays. */
   {
      /* the padded element behind the last nonzero (clamped loads of trailing empty tiles): value 0 and a column
       * inside the owned slab, so that the gather stays in range with halos too */
      const int32_t padcol = (int32_t)x0;
      const char zero[16] = {0};
      if (hipk_upload(ctx, A->rowptr, rowptr_host, (size_t)(nrows_local + 1) * 4) ||
            hipk_upload(ctx, A->colind, colind_host, (size_t)nnz * 4) ||
            hipk_upload(ctx, A->values, values_host, (size_t)nnz * es) ||
            hipk_upload(ctx, A->tiles, tiles.data(), tiles.size() * 4) ||
            hipk_upload(ctx, A->tileinfo, tinfo.data(), tinfo.size() * sizeof(int4)) ||
            hipk_upload(ctx, A->twin, twin.data(), twin.size() * sizeof(int2)) ||
            hipk_upload(ctx, A->diag, dg.data(), (size_t)nrows_local * es) ||
            (A->col16 && hipk_upload(ctx, A->col16, c16.data(), c16.size() * sizeof(uint16_t))) ||
            hipk_upload(ctx, A->colind + nnz, &padcol, (size_t)4) ||
            hipk_upload(ctx, (char *)A->values + (size_t)nnz * es, zero, es))
         return -1;
   }
   /* scattered column pattern over an input vector that is all local (the rectangular operators of the singular value
    * problem): the panel-blocked form serves the one-column products */
   if (lo == 0 && hi == 0 && x0 == 0 && xlen == ncols_global && (dt == HIPK_F64 || dt == HIPK_F32)) {
      const int rcp = hipk_pb_build(ctx, dt, nrows_local, ncols_global, rowptr_host, colind_host, values_host, &A->pb);
      if (rcp < 0) return rcp;
   }
   /* rows that repeat (constant-coefficient stencils, lattice operators): one byte per row + a pattern table serves the
    * one-column products (hipk_sparse_pat.hip); the offsets are taken against the row, so the input entries must be
    * numbered like the rows AND be as many as the rows (square operators, row slabs with halos): pat_kernel sizes its x
    * descriptor, its end-of-vector test and the own-row reads of its padded table entries from nrows, so a rectangular
    * operator (hipk_csr_create_rect: x0 == row0 == 0 but xlen != nrows — A and A' of the singular value problem when the
    * panel-blocked builder declines) must stay on the row tiles */
   if (x0 == row0 && xlen == nrows_local && (dt == HIPK_F64 || dt == HIPK_F32) && !A->pb) {
      const int rcp = hipk_pat_build(ctx, dt, nrows_local, row0, rowptr_host, colind_host, values_host, &A->pat);
      if (rcp < 0) return rcp;
   }
   *out = A;
   return 0;
}

extern "C" int hipk_csr_create(hipk_ctx *ctx, hipk_dtype dt, int64_t nrows_local,
      int64_t ncols_global, int64_t row0, const int32_t *rowptr_host,
      const int32_t *colind_host, const void *values_host, hipk_csr **out) {
   return csr_create_impl(ctx, dt, nrows_local, ncols_global, row0, row0, nrows_local, rowptr_host,
         colind_host, values_host, out);
}
extern "C" int hipk_csr_create_rect(hipk_ctx *ctx, hipk_dtype dt, int64_t nrows, int64_t ncols,
      const int32_t *rowptr_host, const int32_t *colind_host, const void *values_host, hipk_csr **out) {
   return csr_create_impl(ctx, dt, nrows, ncols, 0, 0, ncols, rowptr_host, colind_host, values_host, out);
}

extern "C" int hipk_stencil_create(hipk_ctx *ctx, hipk_dtype dt, int nx, int ny, int nz,
      int64_t row0, int64_t nrows_local, hipk_csr **out) {
   if (dt != HIPK_F64 && dt != HIPK_F32) return -44;
   hipk_csr *A = (hipk_csr *)calloc(1, sizeof(hipk_csr));
   if (!A) return -2;
   A->ctx = ctx; A->dt = dt; A->kind = 1;
   A->sx = nx; A->sy = ny > 0 ? ny : 1; A->sz = nz > 0 ? nz : 1;
   const int64_t n = (int64_t)A->sx * A->sy * A->sz;
   A->nrows = nrows_local; A->ncols_global = n; A->row0 = row0;
   A->x0 = row0; A->xlen = nrows_local;
   const int dims = (A->sz > 1) ? 3 : (A->sy > 1 ? 2 : 1);
   A->nnz = n * (2 * dims + 1); /* nominal */
   /* reach of the stencil outside the slab: one x-y plane (3-D), one x line (2-D) */
   const int64_t reach = (A->sz > 1) ? (int64_t)A->sx * A->sy : (A->sy > 1 ? A->sx : 1);
   A->halo_lo = row0 > 0 ? (reach < row0 ? reach : row0) : 0;
   const int64_t above = n - (row0 + nrows_local);
   A->halo_hi = above > 0 ? (reach < above ? reach : above) : 0;
   A->ld_lo = A->halo_lo; A->ld_hi = A->halo_hi;
   const size_t es = elem_size(dt);
   if (hipk_malloc(ctx, (size_t)nrows_local * es, &A->diag)) return -2;
   int gx = hipk_grid_for_rows(ctx, nrows_local, HIPK_BLOCK * 4, 8);
   if (dt == HIPK_F64)
      hipLaunchKernelGGL(fill_kernel<double>, dim3(gx), dim3(HIPK_BLOCK), 0, ctx->stream, (double *)A->diag, nrows_local, 2.0 * dims);
   else
      hipLaunchKernelGGL(fill_kernel<float>, dim3(gx), dim3(HIPK_BLOCK), 0, ctx->stream, (float *)A->diag, nrows_local, 2.0 * dims);
   HIPK_CHECK(hipGetLastError());
   *out = A;
   return 0;
}

extern "C" int hipk_csr_destroy(hipk_csr *A) {
   if (!A) return 0;
   hipStreamSynchronize(A->ctx->stream);
   if (A->rowptr) (void)hipFree(A->rowptr);
   if (A->colind) (void)hipFree(A->colind);
   if (A->values) (void)hipFree(A->values);
   if (A->tiles) (void)hipFree(A->tiles);
   if (A->tileinfo) (void)hipFree(A->tileinfo);
   if (A->twin) (void)hipFree(A->twin);
   if (A->col16) (void)hipFree(A->col16);
   if (A->diag) (void)hipFree(A->diag);
   if (A->pb) hipk_pb_destroy(A->pb);
   if (A->pat) hipk_pat_destroy(A->pat);
   free(A);
   return 0;
}

extern "C" const void *hipk_csr_diag(hipk_csr *A) { return A->diag; }
extern "C" int64_t hipk_csr_nnz(const hipk_csr *A) { return A->nnz; }
extern "C" int64_t hipk_csr_halo_lo(const hipk_csr *A) { return A->halo_lo; }
extern "C" int64_t hipk_csr_halo_hi(const hipk_csr *A) { return A->halo_hi; }
extern "C" int hipk_csr_set_halo(hipk_csr *A, const void *lo, const void *hi) {
   A->xlo = lo; A->xhi = hi; A->ld_lo = A->halo_lo; A->ld_hi = A->halo_hi;
   return 0;
}
extern "C" int hipk_csr_set_halo_ld(hipk_csr *A, const void *lo, int64_t ld_lo, const void *hi, int64_t ld_hi) {
   A->xlo = lo; A->xhi = hi; A->ld_lo = ld_lo; A->ld_hi = ld_hi;
   return 0;
}

/* the 2-byte index stream of a matrix, if it has one (HIPK_SPMM_NO_COL16=1: never, the A/B knob) */
static const uint16_t *csr16(const hipk_csr *A) {
   static int no16 = -1;
   if (no16 < 0) no16 = getenv("HIPK_SPMM_NO_COL16") != NULL;
   return no16 ? (const uint16_t *)NULL : A->col16;
}

/* bytes per nonzero the tile kernels really stream for the index: 2 with the 16-bit index stream, else 4 */
extern "C" int hipk_csr_index_bytes(const hipk_csr *A) { return (A && A->kind == 0 && csr16(A)) ? 2 : 4; }

/* number of column panels of the panel-blocked form (0: plain CSR kernels serve this matrix) */
extern "C" int hipk_csr_panels(const hipk_csr *A) { return A && A->pb ? hipk_pb_panels(A->pb) : 0; }
/* the form that serves the ONE-column products of this matrix right now: 0 CSR row tiles, 1 panel-blocked, 2 row patterns, 3 stencil */
extern "C" int hipk_csr_format(const hipk_csr *A) {
   if (!A) return -1;
   if (A->kind == 1) return 3;
   if (A->pat && hipk_pat_enabled()) return 2;
   return A->pb ? 1 : 0;
}
extern "C" int hipk_csr_npatterns(const hipk_csr *A) { return A && A->pat ? hipk_pat_npatterns(A->pat) : 0; }
/* bytes ONE one-column product (fused = with the second output of hipk_csr_matvec_scaled) moves through HBM in the form in use */
extern "C" double hipk_csr_product_bytes(const hipk_csr *A, int fused) {
   if (!A) return 0.0;
   const double es = (A->dt == HIPK_F64 || A->dt == HIPK_C32) ? 8 : A->dt == HIPK_C64 ? 16 : 4;
   const double vec = (fused ? 3.0 : 2.0) * (double)A->nrows * es;
   switch (hipk_csr_format(A)) {
   case 3: return vec;
   case 2: return hipk_pat_bytes(A->pat, fused);
   case 1: return hipk_pb_bytes(A->pb) + (fused ? (double)A->nrows * es : 0.0);
   default: return (double)A->nnz * (es + hipk_csr_index_bytes(A)) + (A->nrows + 1) * 4.0 + vec;
   }
}
extern "C" double hipk_csr_streamed_bytes(const hipk_csr *A) { return A && A->pb ? hipk_pb_bytes(A->pb) : 0.0; }

/* stream the matrix past the Infinity Cache?  Yes when its (value, index) stream alone is more than about three quarters
 * of the 256 MiB (it cannot stay until the next product anyway); HIPK_SPMV_NT=0 / 1 forces it (A/B knob) */
static bool csr_stream_nt(const hipk_csr *A) {
   static int force = -2;
   if (force == -2) { const char *e = getenv("HIPK_SPMV_NT"); force = e ? atoi(e) : -1; }
   if (force >= 0) return force != 0;
   const double es = (A->dt == HIPK_F64) ? 8 : 4;
   return (double)A->nnz * (es + (A->col16 ? 2 : 4)) > 192.0 * 1024 * 1024;
}

template <typename T>
static int csr_matvec_t(hipk_csr *A, hipStream_t stream, const T *x, int64_t ldx, T *y, int64_t ldy, int ncols,
      const double *shift_host = NULL) {
   hipk_ctx *ctx = A->ctx;
   if ((A->halo_lo > 0 && !A->xlo) || (A->halo_hi > 0 && !A->xhi)) {
      fprintf(stderr, "primme_amd: matvec needs halo data (rows outside the local slab) but none was set\n");
      return -1;
   }
   const double es = sizeof(T);
   const double alg = (A->kind == 1) ? 2.0 * A->nrows * es * ncols
                                     : (double)A->nnz * (es + 4) + (A->nrows + 1) * 4.0 + 2.0 * A->nrows * es * ncols;
   static int pb_maxcols = -1;       /* HIPK_PB_MAXCOLS: widest block served column by column through the panel-blocked form */
   if (pb_maxcols < 0) { const char *e = getenv("HIPK_PB_MAXCOLS"); pb_maxcols = e ? atoi(e) : 2; }
   /* what the form that serves THIS product streams (the class' roofline fraction is taken on these bytes): one pattern byte
    * per row; the panel-blocked entries; the tile kernels' 2- or 4-byte index stream (the windowed and the one-column kernel
    * read the 2-byte one when the matrix has it, the plain row-block kernel the 4-byte one) */
   double streamed = alg;
   if (A->kind == 0) {
      const bool one_pat = A->pat && hipk_pat_enabled() && ncols == 1 && !shift_host;
      const bool one_pb = A->pb && !shift_host && ncols <= pb_maxcols;
      const bool win = A->halo_lo == 0 && A->halo_hi == 0 && ncols <= 64 && ((A->windowed && ncols >= 2) || shift_host);
      if (one_pat) streamed = hipk_pat_bytes(A->pat, 0);
      else if (one_pb) streamed = hipk_pb_bytes(A->pb) * ncols;
      else streamed = (double)A->nnz * (es + ((win || ncols == 1) && csr16(A) ? 2 : 4)) + (A->nrows + 1) * 4.0 + 2.0 * A->nrows * es * ncols;
   }
   const int pslot = hipk_prof_begin_s(HIPK_PROF_SPMV, stream, alg, streamed);
   if (A->pat && hipk_pat_enabled() && ncols == 1 && !shift_host) {
      const int rc = hipk_pat_matvec(A->pat, stream, hipk_pat_grid(A->pat, ctx->num_cu), x, y, A->halo_lo, A->halo_hi, A->xlo, A->xhi, NULL, 0, NULL,
            NULL, NULL);
      hipk_prof_end(pslot, stream);
      return rc;
   }
   if (A->pb && !shift_host && ncols <= pb_maxcols) {
      const int rc = hipk_pb_matvec(A->pb, stream, x, ldx, y, ldy, ncols);
      hipk_prof_end(pslot, stream);
      return rc;
   }
   if (A->kind == 1) {
      int gx = hipk_grid_for_rows(ctx, A->nrows, HIPK_BLOCK, 8);
      hipLaunchKernelGGL(stencil_kernel<T>, dim3(gx), dim3(HIPK_BLOCK), 0, stream, A->sx,
            A->sy, A->sz, A->row0, A->nrows, x, ldx, y, ldy, ncols, A->halo_lo, A->halo_hi,
            (const T *)A->xlo, (const T *)A->xhi, A->ld_lo, A->ld_hi);
   } else {
      int gx = ((A->ntiles + 7) / 8) * 8;
#define LAUNCH_ROWS(NCV) hipLaunchKernelGGL((csr_rows_block_kernel<T, NCV>), dim3(gx), dim3(HIPK_BLOCK), 0, stream, \
               A->tiles, A->ntiles, A->rowptr, A->colind, (const T *)A->values, x, ldx, y, ldy, \
               ncols, A->x0, A->xlen, A->halo_lo, A->halo_hi, (const T *)A->xlo, (const T *)A->xhi, A->ld_lo, A->ld_hi)
      static int force = -1, nowin = -1;         /* HIPK_SPMM_NC, HIPK_NO_WINDOW: measurement knobs, read once */
      if (force < 0) { const char *env = getenv("HIPK_SPMM_NC"); force = env ? atoi(env) : 0; }
      if (nowin < 0) nowin = getenv("HIPK_NO_WINDOW") != NULL;
      const bool use_win = A->kind == 0 && A->halo_lo == 0 && A->halo_hi == 0 && ncols <= 64 &&
                           ((A->windowed && ncols >= 2 && !nowin) || shift_host);
      if (use_win) {
         SpmmShift sh;
         sh.on = shift_host != NULL;
         static int nosplit = -1;                   /* HIPK_SPMM_NO_SPLIT=1: one lane per row always (A/B knob) */
         if (nosplit < 0) nosplit = getenv("HIPK_SPMM_NO_SPLIT") != NULL;
         sh.nosplit = nosplit;
         static int evenstride = -1;                /* HIPK_SPMM_EVEN_STRIDE=1: window rows ncols apart in LDS, as until round 5 (A/B knob) */
         if (evenstride < 0) evenstride = getenv("HIPK_SPMM_EVEN_STRIDE") != NULL;
         sh.evenstride = evenstride;
         for (int c = 0; c < 64; c++) sh.s[c] = (shift_host && c < ncols) ? shift_host[c] : 0.0;
         /* the window buffer in two sizes: 1536 values when every window of this matrix fits (38 KB of LDS per
          * workgroup, 4 per CU) and XS_MAX = 3072 otherwise (50 KB, 3 per CU) */
         static int bigxs = -1;                     /* HIPK_SPMM_BIG_WINDOW=1: always the large buffer (A/B knob) */
         if (bigxs < 0) bigxs = getenv("HIPK_SPMM_BIG_WINDOW") != NULL;
         const bool small = !bigxs && (int64_t)A->cw_max * (evenstride ? ncols : (ncols | 1)) <= XS_SMALL;
#define LAUNCH_WIN(NCV, XSV, C16V) hipLaunchKernelGGL((csr_window_block_kernel<T, NCV, XSV, C16V>), dim3(gx), dim3(HIPK_BLOCK), 0, stream, A->tileinfo, A->twin, \
                  A->ntiles, A->rowptr, A->colind, csr16(A), A->row0 - A->c16back, (const T *)A->values, x, ldx, y, ldy, ncols, A->x0, sh)
#define LAUNCH_WIN2(NCV, XSV) do { if (csr16(A)) LAUNCH_WIN(NCV, XSV, true); else LAUNCH_WIN(NCV, XSV, false); } while (0)
         if (ncols <= 2) { if (small) LAUNCH_WIN2(2, XS_SMALL); else LAUNCH_WIN2(2, XS_MAX); }
         else { if (small) LAUNCH_WIN2(4, XS_SMALL); else LAUNCH_WIN2(4, XS_MAX); }
#undef LAUNCH_WIN2
#undef LAUNCH_WIN
      } else if (ncols == 1 && force == 0) {
#define LAUNCH_STREAM(C16V, NTV) hipLaunchKernelGGL((csr_stream_kernel<T, false, C16V, NTV>), dim3(gx), dim3(HIPK_BLOCK), 0, stream, \
               A->tileinfo, A->ntiles, A->rowptr, A->colind, csr16(A), A->row0 - A->c16back, (const T *)A->values, x, ldx, y, ldy, \
               ncols, A->x0, A->xlen, A->halo_lo, A->halo_hi, (const T *)A->xlo, \
               (const T *)A->xhi, A->ld_lo, A->ld_hi, (const double *)NULL, (T *)NULL, (double *)NULL, hipk_fin_args())
         if (csr_stream_nt(A)) { if (csr16(A)) LAUNCH_STREAM(true, true); else LAUNCH_STREAM(false, true); }
         else { if (csr16(A)) LAUNCH_STREAM(true, false); else LAUNCH_STREAM(false, false); }
#undef LAUNCH_STREAM
      }
      else if (force == 1) LAUNCH_ROWS(1);
      else if (force == 2 || (force == 0 && ncols <= 2)) LAUNCH_ROWS(2);
      else LAUNCH_ROWS(4);
#undef LAUNCH_ROWS
   }
   hipk_prof_end(pslot, stream);
   HIPK_CHECK(hipGetLastError());
   return 0;
}

/* complex matrices: the row-tile kernel of hipk_complex.hip (blocks of up to 64 columns per launch) */
static int csr_matvec_z(hipk_csr *A, hipStream_t st, const void *x, int64_t ldx, void *y, int64_t ldy, int ncols, const double *shift_host) {
   if (A->kind != 0) return -44;
   if ((A->halo_lo > 0 && !A->xlo) || (A->halo_hi > 0 && !A->xhi)) {
      fprintf(stderr, "primme_amd: matvec needs halo data (rows outside the local slab) but none was set\n");
      return -1;
   }
   const size_t es = A->dt == HIPK_C64 ? 16 : 8;
   const int pslot = hipk_prof_begin(HIPK_PROF_SPMV, st, (double)A->nnz * (es + 4) + (A->nrows + 1) * 4.0 + 2.0 * A->nrows * es * ncols);
   int rc = 0;
   for (int c0 = 0; c0 < ncols && !rc; c0 += 64) {
      const int nc = ncols - c0 < 64 ? ncols - c0 : 64;
      rc = hipk_z_csr_matvec(A->dt, st, A->tileinfo, A->ntiles, A->rowptr, A->colind, A->values, (const char *)x + (size_t)c0 * ldx * es, ldx,
            (char *)y + (size_t)c0 * ldy * es, ldy, nc, A->x0, A->xlen, A->halo_lo, A->halo_hi,
            A->xlo ? (const char *)A->xlo + (size_t)c0 * A->ld_lo * es : NULL, A->xhi ? (const char *)A->xhi + (size_t)c0 * A->ld_hi * es : NULL,
            A->ld_lo, A->ld_hi, shift_host ? shift_host + c0 : NULL);
   }
   hipk_prof_end(pslot, st);
   return rc;
}

extern "C" int hipk_csr_matvec(hipk_csr *A, void *hip_stream, const void *x, int64_t ldx, void *y,
      int64_t ldy, int ncols) {
   if (ncols <= 0 || A->nrows == 0) return 0;
   hipStream_t st = hip_stream ? (hipStream_t)hip_stream : A->ctx->stream;
   if (HIPK_IS_Z(A->dt)) return csr_matvec_z(A, st, x, ldx, y, ldy, ncols, NULL);
   if (A->dt == HIPK_F64) return csr_matvec_t<double>(A, st, (const double *)x, ldx, (double *)y, ldy, ncols);
   return csr_matvec_t<float>(A, st, (const float *)x, ldx, (float *)y, ldy, ncols);
}

/* y = A x - shift[c] x(:,c) in one launch (square CSR operators without halo; returns 1 when the
 * operator is not covered and the caller should apply the shift itself) */
extern "C" int hipk_csr_matvec_shifted(hipk_csr *A, void *hip_stream, const void *x, int64_t ldx, void *y, int64_t ldy,
      int ncols, const double *shift_host) {
   if (ncols <= 0 || A->nrows == 0) return 0;
   if (A->kind != 0 || A->halo_lo != 0 || A->halo_hi != 0 || A->x0 != A->row0 || A->xlen != A->nrows || ncols > 64 || !shift_host) return 1;
   hipStream_t st = hip_stream ? (hipStream_t)hip_stream : A->ctx->stream;
   if (HIPK_IS_Z(A->dt)) return csr_matvec_z(A, st, x, ldx, y, ldy, ncols, shift_host);
   if (A->dt == HIPK_F64) return csr_matvec_t<double>(A, st, (const double *)x, ldx, (double *)y, ldy, ncols, shift_host);
   return csr_matvec_t<float>(A, st, (const float *)x, ldx, (float *)y, ldy, ncols, shift_host);
}

/* y = A (a x), xout = a x, dot_dev[0] = xout' y with a = 1/sqrt(norm2_dev[0]) — see csr_stream_kernel<T, true>.
 * One column, CSR operators whose rows and input entries coincide (square, row-partitioned). */
extern "C" int hipk_csr_matvec_scaled(hipk_csr *A, hipk_ctx *ctx, const void *x, const double *norm2_dev,
      void *xout, void *y, double *dot_dev) {
   /* ctx: the CALLER's context (stream, reduction scratch, pinned mirror of the results) — the matrix may
    * have been created under another one */
   if (A->kind != 0 || A->x0 != A->row0 || A->xlen != A->nrows || x == xout || !ctx) return -1;
   if (HIPK_IS_Z(A->dt)) return -44;           /* the fused tail is real-arithmetic code (eigs_scalar.h) */
   hipStream_t st = ctx->stream;
   if ((A->halo_lo > 0 && !A->xlo) || (A->halo_hi > 0 && !A->xhi)) return -1;
   if (A->nrows == 0) {      /* an empty slab: the result is 0; no flagged launch, so the next wait drains the stream */
      ctx->tail_want = 0;
      if (ctx->tail_np2 > 0) { const int np = ctx->tail_np2; ctx->tail_np2 = 0; if (hipk_finalize_partials_t(ctx, ctx->tailp, np, 1, ctx->tail_norm2_out)) return -1; }
      HIPK_CHECK(hipMemsetAsync(dot_dev, 0, sizeof(double), st));
      double *mh = hipk_mirror_of(ctx, dot_dev);
      if (mh) { HIPK_CHECK(hipStreamSynchronize(st)); *mh = 0.0; }
      ctx->need_sync = 1;
      return 0;
   }
   const bool pat = A->pat && hipk_pat_enabled();
   const int gx = pat ? hipk_pat_grid(A->pat, ctx->num_cu) : ((A->ntiles + 7) / 8) * 8;
   if (hipk_reserve_partials(ctx, (size_t)gx)) return -2;
   const double es = A->dt == HIPK_F64 ? 8 : 4;
   /* the tail of a block-size-1 iteration without second-stage launches (hipk_tail_defer): |t|^2 may still be np2 partial sums
    * (only the row-pattern kernel adds them itself; any other form gets them finished the separate way first), and the second
    * stage of t'At may be left to hipk_tail_finish */
   int np2 = 0;
   if (ctx->tail_np2 > 0) {
      if (pat && norm2_dev == ctx->tail_norm2_out) np2 = ctx->tail_np2;
      else {
         const int np = ctx->tail_np2;
         ctx->tail_np2 = 0;
         const int rc = hipk_finalize_partials_t(ctx, ctx->tailp, np, 1, ctx->tail_norm2_out);
         if (rc) return rc;
      }
   }
   const bool defer_dot = (ctx->tail_want & HIPK_TAIL_DOT) && ctx->tail_np3 == 0 && !hipk_inkernel_fin_mask();
   ctx->tail_want = 0;                                           /* one-shot */
   hipk_fin_args fa;
   if (defer_dot) memset(&fa, 0, sizeof(fa)); else fa = hipk_make_fin(ctx, dot_dev, HIPK_FIN_SPMV, gx, 1);
   const int pslot = hipk_prof_begin_s(HIPK_PROF_SPMV, st, (double)A->nnz * (es + 4) + (A->nrows + 1) * 4.0 + 3.0 * A->nrows * es, hipk_csr_product_bytes(A, 1));
   if (pat) {
      const int rc = hipk_pat_matvec(A->pat, st, gx, x, y, A->halo_lo, A->halo_hi, A->xlo, A->xhi, np2 > 0 ? ctx->tailp : norm2_dev, np2, xout, ctx->partials, &fa);
      hipk_prof_end(pslot, st);
      if (rc) return rc;
      if (defer_dot) { ctx->tail_np3 = gx; ctx->tail_dot_out = dot_dev; return 0; }
      if (fa.enabled) return 0;
      return hipk_finalize_partials(ctx, ctx->partials, gx, 1, dot_dev);
   }
#define LAUNCH_FUSED(TT, C16V, NTV) hipLaunchKernelGGL((csr_stream_kernel<TT, true, C16V, NTV>), dim3(gx), dim3(HIPK_BLOCK), 0, st, A->tileinfo, A->ntiles, A->rowptr, \
            A->colind, csr16(A), A->row0 - A->c16back, (const TT *)A->values, (const TT *)x, A->nrows, (TT *)y, A->nrows, 1, A->x0, A->xlen, A->halo_lo, \
            A->halo_hi, (const TT *)A->xlo, (const TT *)A->xhi, A->ld_lo, A->ld_hi, norm2_dev, (TT *)xout, ctx->partials, fa)
   if (csr_stream_nt(A)) {
      if (A->dt == HIPK_F64) { if (csr16(A)) LAUNCH_FUSED(double, true, true); else LAUNCH_FUSED(double, false, true); }
      else { if (csr16(A)) LAUNCH_FUSED(float, true, true); else LAUNCH_FUSED(float, false, true); }
   } else {
      if (A->dt == HIPK_F64) { if (csr16(A)) LAUNCH_FUSED(double, true, false); else LAUNCH_FUSED(double, false, false); }
      else { if (csr16(A)) LAUNCH_FUSED(float, true, false); else LAUNCH_FUSED(float, false, false); }
   }
#undef LAUNCH_FUSED
   hipk_prof_end(pslot, st);
   HIPK_CHECK(hipGetLastError());
   if (defer_dot) { ctx->tail_np3 = gx; ctx->tail_dot_out = dot_dev; return 0; }
   if (fa.enabled) return 0;
   return hipk_finalize_partials(ctx, ctx->partials, gx, 1, dot_dev);
}
extern "C" int hipk_csr_kind(const hipk_csr *A) { return A->kind; }
hipk_ctx *hipk_csr_ctx(const hipk_csr *A) { return A->ctx; }
int hipk_csr_fusable(const hipk_csr *A) { return A && A->kind == 0 && A->x0 == A->row0 && A->xlen == A->nrows && !HIPK_IS_Z(A->dt); }      /* library-internal (hipk_internal.h) */
extern "C" hipk_dtype hipk_csr_dtype(const hipk_csr *A) { return A->dt; }
extern "C" int64_t hipk_csr_nrows(const hipk_csr *A) { return A->nrows; }

extern "C" int hipk_jacobi_apply(void *hip_stream, hipk_dtype dt, int64_t m, const void *diag,
      const double *shift_host, double min_den, const void *x, int64_t ldx, void *y, int64_t ldy, int ncols) {
   if (ncols <= 0) return 0;
   if (!(min_den > 0.0)) min_den = 1e-300;
   if (ncols > 64) return -1;
   hipk_prof_scope ps_(HIPK_PROF_VEC, (hipStream_t)hip_stream, (double)m * (double)elem_size(dt) * (2.0 * ncols) + (double)m * (double)elem_size(hipk_real_of(dt)));
   if (HIPK_IS_Z(dt)) {
      int dev = 0, ncu = 256;
      if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
      return hipk_z_jacobi((hipStream_t)hip_stream, ncu > 0 ? ncu : 256, dt, m, diag, shift_host, min_den, x, ldx, y, ldy, ncols);
   }
   JacShift sh;
   for (int c = 0; c < ncols; c++) sh.s[c] = shift_host ? shift_host[c] : 0.0;
   static int num_cu = 0;                      /* launch geometry only: read the device once */
   if (num_cu == 0) {
      int dev = 0, n = 0;
      if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) num_cu = n;
      else num_cu = 256;
   }
   struct { hipStream_t stream; } ctx_ = {(hipStream_t)hip_stream}, *ctx = &ctx_;
   int64_t need = (m + HIPK_BLOCK * 4 - 1) / (HIPK_BLOCK * 4);
   int gx = (int)(need < 1 ? 1 : (need < (int64_t)num_cu * 8 ? need : (int64_t)num_cu * 8));
   if (dt == HIPK_F64)
      hipLaunchKernelGGL(jacobi_kernel<double>, dim3(gx), dim3(HIPK_BLOCK), 0, ctx->stream, (const double *)diag, sh, min_den, (const double *)x, ldx, (double *)y, ldy, ncols, m);
   else if (dt == HIPK_F32)
      hipLaunchKernelGGL(jacobi_kernel<float>, dim3(gx), dim3(HIPK_BLOCK), 0, ctx->stream, (const float *)diag, sh, min_den, (const float *)x, ldx, (float *)y, ldy, ncols, m);
   else return -44;
   HIPK_CHECK(hipGetLastError());
   return 0;
}
