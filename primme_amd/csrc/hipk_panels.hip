/* hipk_panels.hip — tall-skinny panel kernels of the Davidson inner loop (gfx950).
 *
 * All of these are HBM-bound streaming kernels: every basis column is read once
 * per launch with unit-stride wave accesses (64 lanes x 8 B = 512 B per
 * instruction, NC independent loads in flight per lane), reductions go
 * lane -> wave (DPP shuffles) -> workgroup (LDS) -> per-block partial, and a tiny
 * second launch adds the partials in a fixed order so results are bit-reproducible
 * run to run and identical on every rank.
 *
 * What each kernel replaces in the reference is listed in
 * include/primme_amd_kernels.h.
 */
#include "hipk_internal.h"
#include <vector>
#include <cstddef>

struct SegArgs {
   const void *base[HIPK_MAX_SEGS];
   int64_t ld[HIPK_MAX_SEGS];
   int n[HIPK_MAX_SEGS];
   int total;
};

static int pack_segs(const hipk_seg *segs, int nseg, SegArgs *a) {
   if (nseg < 0 || nseg > HIPK_MAX_SEGS) return -1;
   a->total = 0;
   for (int s = 0; s < HIPK_MAX_SEGS; s++) {
      if (s < nseg && segs[s].ncols > 0) {
         a->base[s] = segs[s].base; a->ld[s] = segs[s].ld; a->n[s] = segs[s].ncols;
      } else {
         a->base[s] = NULL; a->ld[s] = 0; a->n[s] = 0;
      }
      a->total += a->n[s];
   }
   return 0;
}

/* 16-byte lane accesses: VW consecutive rows per lane (2 doubles / 4 floats) when every
 * column involved is 16-byte aligned, VW = 1 otherwise */
template <typename T, int VW> struct lanevec { T e[VW]; };
template <> struct __attribute__((aligned(16))) lanevec<double, 2> { double e[2]; };
template <> struct __attribute__((aligned(16))) lanevec<float, 4> { float e[4]; };
template <> struct __attribute__((aligned(8))) lanevec<float, 2> { float e[2]; };
template <typename T> struct vecwidth { enum { value = 16 / sizeof(T) }; };
/* Streamed panels are loaded (and the restart pass' outputs stored) with the NON-TEMPORAL hint: V and W are read once per
 * kernel and are far larger than the 256 MiB Infinity Cache, so letting them allocate there only evicts what does get
 * re-read every iteration — the CSR matrix and the vectors of the SpMV (215 MB at n = 2 M).  Measured on one box, back to
 * back (profiles/r03_nontemporal_ab.log): configs[1] 13.72 -> 14.82 eigenpairs/s (SpMV 146 -> 124 ms per solve: it now
 * hits the cache; fused residual / restart class 4.96 -> 5.44 TB/s), north-star workload 2.466 -> 2.321 s per 3000
 * iterations.  HIPK_NT_LOADS is a build-time mask for A/B builds (scripts/build_variant.sh): 1 = W in the fused
 * residual kernel, 2 = V, Q there and the panels of the Gram-Schmidt update, 4 = loads of the restart kernels,
 * 8 = stores of the restart pass, 16 = panels of the TN kernel (no gain: left off).  Default 15. */
#ifndef HIPK_NT_LOADS
#define HIPK_NT_LOADS 15
#endif
template <typename T, int NTBIT>
__device__ __forceinline__ T ldstream1(const T *p) {
   if ((HIPK_NT_LOADS & NTBIT) != 0) return __builtin_nontemporal_load(p);
   return *p;
}
template <typename T, int NTBIT>
__device__ __forceinline__ void ststream1(T *p, T v) {
   if ((HIPK_NT_LOADS & NTBIT) != 0) __builtin_nontemporal_store(v, p);
   else *p = v;
}
template <typename T, int VW, int NTBIT>
__device__ __forceinline__ lanevec<T, VW> ldstream(const T *col, int64_t idx) {
   if ((HIPK_NT_LOADS & NTBIT) != 0) {
      typedef T nvec __attribute__((ext_vector_type(VW)));
      const nvec t = __builtin_nontemporal_load((const nvec *)col + idx);
      lanevec<T, VW> r;
#pragma unroll
      for (int i = 0; i < VW; i++) r.e[i] = t[i];
      return r;
   }
   return ((const lanevec<T, VW> *)col)[idx];
}

static inline bool aligned16(const void *p, int64_t ld, size_t es) {
   return (((uintptr_t)p) & 15) == 0 && ((ld * (int64_t)es) & 15) == 0;
}
static bool segs_aligned16(const SegArgs &a, size_t es) {
   for (int s = 0; s < HIPK_MAX_SEGS; s++)
      if (a.n[s] > 0 && !aligned16(a.base[s], a.ld[s], es)) return false;
   return true;
}

template <typename T>
__device__ __forceinline__ const T *seg_col(const SegArgs &s, int j) {
   int q = 0;
   if (j >= s.n[0]) { j -= s.n[0]; q = 1; if (j >= s.n[1]) { j -= s.n[1]; q = 2; } }
   return (const T *)s.base[q] + (size_t)j * (size_t)s.ld[q];
}

/* finalize with an output leading dimension: out[(o % nrows) + (o / nrows)*ldout] */
__global__ void __launch_bounds__(HIPK_BLOCK)
finalize_ld_kernel(const double *__restrict__ partials, int nblocks, int nout, int nrows,
      int ldout, double *__restrict__ out, double *__restrict__ out_host, hipk_fin_flag fin) {
   __shared__ double sm[HIPK_BLOCK / HIPK_WAVE];
   const int o = blockIdx.x;
   double s = 0.0;
   for (int b = threadIdx.x; b < nblocks; b += HIPK_BLOCK) s += partials[(size_t)b * nout + o];
   s = hipk_wave_sum(s);
   if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
   __syncthreads();
   if (threadIdx.x == 0) {
      const double v = (sm[0] + sm[1]) + (sm[2] + sm[3]);
      const size_t at = (o % nrows) + (size_t)(o / nrows) * ldout;
      out[at] = v;
      if (out_host) out_host[at] = v;
      hipk_publish_flag(fin, gridDim.x);
   }
}

/* ============================ TN: inner products ============================== */
template <typename T, int NC, int NX, int VW>
__global__ void __launch_bounds__(HIPK_BLOCK)
dots_kernel(SegArgs segs, const T *__restrict__ X, int64_t ldX, int nx, int64_t m,
      double *__restrict__ partials) {
   typedef lanevec<T, VW> LV;
   const int j0 = blockIdx.y * NC;             /* first basis column of this chunk */
   const int c0 = blockIdx.z * NX;             /* first right-hand column          */
   const int ncv = min(NC, segs.total - j0);
   const int nxv = min(NX, nx - c0);
   const T *cp[NC];
#pragma unroll
   for (int jj = 0; jj < NC; jj++) cp[jj] = seg_col<T>(segs, jj < ncv ? j0 + jj : j0);
   const T *xp = X + (size_t)c0 * (size_t)ldX;

   double acc[NC][NX];
#pragma unroll
   for (int jj = 0; jj < NC; jj++)
#pragma unroll
      for (int c = 0; c < NX; c++) acc[jj][c] = 0.0;

   const int64_t stride = (int64_t)gridDim.x * HIPK_BLOCK;
   const int64_t mg = m / VW;                   /* full lane groups */
   for (int64_t g = (int64_t)blockIdx.x * HIPK_BLOCK + threadIdx.x; g < mg; g += stride) {
      LV xv[NX], a[NC];
#pragma unroll
      for (int c = 0; c < NX; c++)
         if (c < nxv) xv[c] = ((const LV *)(xp + (size_t)c * ldX))[g];
#pragma unroll
      for (int jj = 0; jj < NC; jj++)
         if (jj < ncv) a[jj] = ldstream<T, VW, 16>(cp[jj], g);
#pragma unroll
      for (int jj = 0; jj < NC; jj++)
         if (jj < ncv) {
#pragma unroll
            for (int c = 0; c < NX; c++)
               if (c < nxv) {
#pragma unroll
                  for (int r = 0; r < VW; r++) acc[jj][c] = fma((double)a[jj].e[r], (double)xv[c].e[r], acc[jj][c]);
               }
         }
   }
   if (VW > 1 && blockIdx.x == 0 && threadIdx.x < (unsigned)(m - mg * VW)) {   /* ragged tail rows */
      const int64_t i = mg * VW + threadIdx.x;
#pragma unroll
      for (int jj = 0; jj < NC; jj++)
         if (jj < ncv) {
#pragma unroll
            for (int c = 0; c < NX; c++)
               if (c < nxv) acc[jj][c] = fma((double)cp[jj][i], (double)xp[i + (size_t)c * ldX], acc[jj][c]);
         }
   }

   __shared__ double sm[HIPK_BLOCK / HIPK_WAVE][NC * NX];
   const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
   for (int jj = 0; jj < NC; jj++)
#pragma unroll
      for (int c = 0; c < NX; c++) {
         double v = hipk_wave_sum(acc[jj][c]);
         if (lane == 0) sm[wv][jj * NX + c] = v;
      }
   __syncthreads();
   if (threadIdx.x < NC * NX) {
      const int jj = threadIdx.x / NX, c = threadIdx.x % NX;
      if (jj < ncv && c < nxv) {
         double v = (sm[0][threadIdx.x] + sm[1][threadIdx.x]) + (sm[2][threadIdx.x] + sm[3][threadIdx.x]);
         partials[(size_t)blockIdx.x * ((size_t)segs.total * nx) + (size_t)(j0 + jj) +
                  (size_t)(c0 + c) * segs.total] = v;
      }
   }
}

/* Several right-hand columns (block methods): one WAVE per chunk of NC basis columns, all waves of
 * a workgroup walk the SAME rows, so the NX right-hand columns are fetched from HBM once per
 * workgroup (the other waves hit the CU's L1) instead of once per column chunk.  Each wave reduces
 * its own NC x NX block of the result; no cross-wave reduction is needed. */
template <typename T, int NC, int NX, int VW, int WAVES>
__global__ void __launch_bounds__(64 * WAVES)
dots_wide_kernel(SegArgs segs, const T *__restrict__ X, int64_t ldX, int nx, int64_t m,
      double *__restrict__ partials) {
   typedef lanevec<T, VW> LV;
   const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
   const int j0 = (blockIdx.y * WAVES + wv) * NC;
   const int c0 = blockIdx.z * NX;
   const int ncv = min(NC, segs.total - j0);     /* <= 0: this wave has no columns (still walks along) */
   const int nxv = min(NX, nx - c0);
   const T *cp[NC];
#pragma unroll
   for (int jj = 0; jj < NC; jj++) cp[jj] = seg_col<T>(segs, (jj < ncv) ? j0 + jj : 0);
   const T *xp = X + (size_t)c0 * (size_t)ldX;
   double acc[NC][NX];
#pragma unroll
   for (int jj = 0; jj < NC; jj++)
#pragma unroll
      for (int c = 0; c < NX; c++) acc[jj][c] = 0.0;
   const int64_t stride = (int64_t)gridDim.x * 64;
   const int64_t mg = m / VW;
   if (ncv > 0) {
      for (int64_t g = (int64_t)blockIdx.x * 64 + lane; g < mg; g += stride) {
         LV xv[NX], a[NC];
#pragma unroll
         for (int c = 0; c < NX; c++)
            if (c < nxv) xv[c] = ((const LV *)(xp + (size_t)c * ldX))[g];
#pragma unroll
         for (int jj = 0; jj < NC; jj++)
            if (jj < ncv) a[jj] = ldstream<T, VW, 16>(cp[jj], g);
#pragma unroll
         for (int jj = 0; jj < NC; jj++)
            if (jj < ncv) {
#pragma unroll
               for (int c = 0; c < NX; c++)
                  if (c < nxv) {
#pragma unroll
                     for (int r = 0; r < VW; r++) acc[jj][c] = fma((double)a[jj].e[r], (double)xv[c].e[r], acc[jj][c]);
                  }
            }
      }
      if (VW > 1 && blockIdx.x == 0 && lane < (int)(m - mg * VW)) {
         const int64_t i = mg * VW + lane;
#pragma unroll
         for (int jj = 0; jj < NC; jj++)
            if (jj < ncv) {
#pragma unroll
               for (int c = 0; c < NX; c++)
                  if (c < nxv) acc[jj][c] = fma((double)cp[jj][i], (double)xp[i + (size_t)c * ldX], acc[jj][c]);
            }
      }
#pragma unroll
      for (int jj = 0; jj < NC; jj++)
#pragma unroll
         for (int c = 0; c < NX; c++) {
            const double v = hipk_wave_sum(acc[jj][c]);
            if (lane == 0 && jj < ncv && c < nxv)
               partials[(size_t)blockIdx.x * ((size_t)segs.total * nx) + (size_t)(j0 + jj) + (size_t)(c0 + c) * segs.total] = v;
         }
   }
}

/* Blocks of right-hand columns on the matrix cores: G = [Q V]' X with v_mfma_f64_16x16x4_f64
 * (reference: Num_gemm_ddh / Num_compute_gramm, cublas_wrapper.c:479-499, :898-987; the b >= 4 TN
 * panels of Bortho_block_gen and update_projection).  A workgroup stages MF_ROWS rows of up to
 * 16*NT basis columns and of the (<= 16) right-hand columns in LDS with fully coalesced 16-byte
 * loads — each wave fetches whole columns, 1 KB per instruction — and then feeds the MFMA from
 * LDS: for a 16x16x4 step the k index is a ROW of the panels, lane l supplies
 * A[i = l & 15][k = l >> 4] = V(row, column i) and B[k = l >> 4][j = l & 15] = X(row, column j).
 * The column stride in LDS is MF_ROWS + 2 doubles (= 2 mod 32), which makes the 32 lanes of a
 * ds_read_b64 group hit 32 distinct bank pairs.  Wave w takes the k-steps of rows [32 w, 32 w + 32)
 * of the staged tile for every (basis tile, X) pair; the 4-double accumulators (C/D layout of the
 * f64 form: column = lane & 15, row = (lane >> 4) + 4 reg) stay in registers for the whole row
 * range of the workgroup and the four waves' partial tiles are added in a fixed order at the end.
 * An NT x 16-column tile of accumulators costs 8 NT VGPRs per lane where the FMA form
 * (dots_wide_kernel<8, 8>) holds 64 accumulators in 128. */
#define MF_ROWS 128
#define MF_LDS_STRIDE (MF_ROWS + 2)
typedef double mf_acc __attribute__((ext_vector_type(4)));

template <typename T, int NT>
__global__ void __launch_bounds__(HIPK_BLOCK)
dots_mfma_kernel(SegArgs segs, const T *__restrict__ X, int64_t ldX, int nx, int64_t m,
      double *__restrict__ partials) {
   typedef lanevec<T, 2> LV;
   extern __shared__ double mf_lds[];            /* [(16*NT + 16) columns][MF_LDS_STRIDE] */
   const int lane = threadIdx.x & 63;
   const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
   const int j0 = blockIdx.y * 16 * NT;          /* first basis column of this workgroup */
   const int ncv = min(16 * NT, segs.total - j0);
   const int c0 = blockIdx.z * 16;               /* first right-hand column */
   const int nxv = min(16, nx - c0);
   double *sV = mf_lds, *sX = mf_lds + (size_t)16 * NT * MF_LDS_STRIDE;
   mf_acc acc[NT];
#pragma unroll
   for (int t = 0; t < NT; t++) acc[t] = (mf_acc){0.0, 0.0, 0.0, 0.0};
   const int ci = lane & 15, kg = lane >> 4;

   const int64_t ntile = (m + MF_ROWS - 1) / MF_ROWS;
   for (int64_t tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
      const int64_t r0 = tile * MF_ROWS;
      const bool full = r0 + MF_ROWS <= m;
      /* stage: wave w fetches columns w, w + 4, ... (2 rows per lane, 1 KB per column); every load is
       * issued before the first LDS store, absent columns read a valid one and are zeroed afterwards */
      constexpr int NCW = (16 * NT + 16) / 4;
      LV tv[NCW];
      if (full) {
#pragma unroll
         for (int q = 0; q < NCW; q++) {
            const int c = wv + 4 * q;
            const bool isx = c >= 16 * NT;
            const int cc = isx ? c - 16 * NT : c;
            const T *col = isx ? X + (size_t)(c0 + (cc < nxv ? cc : 0)) * ldX : seg_col<T>(segs, j0 + (cc < ncv ? cc : 0));
            tv[q] = ((const LV *)(col + r0))[lane];
         }
      } else {
#pragma unroll
         for (int q = 0; q < NCW; q++) {
            const int c = wv + 4 * q;
            const bool isx = c >= 16 * NT;
            const int cc = isx ? c - 16 * NT : c;
            const T *col = isx ? X + (size_t)(c0 + (cc < nxv ? cc : 0)) * ldX : seg_col<T>(segs, j0 + (cc < ncv ? cc : 0));
            const int64_t i = r0 + 2 * lane;
            const T e0 = col[i < m ? i : m - 1], e1 = col[i + 1 < m ? i + 1 : m - 1];
            tv[q].e[0] = i < m ? e0 : (T)0;
            tv[q].e[1] = i + 1 < m ? e1 : (T)0;
         }
      }
#pragma unroll
      for (int q = 0; q < NCW; q++) {
         const int c = wv + 4 * q;
         const bool isx = c >= 16 * NT;
         const int cc = isx ? c - 16 * NT : c;
         const bool have = isx ? (cc < nxv) : (cc < ncv);
         double *dstc = mf_lds + (size_t)c * MF_LDS_STRIDE + 2 * lane;
         dstc[0] = have ? (double)tv[q].e[0] : 0.0;
         dstc[1] = have ? (double)tv[q].e[1] : 0.0;
      }
      __syncthreads();
      /* 8 k-steps of 4 rows for this wave */
#pragma unroll
      for (int st = 0; st < MF_ROWS / 16; st++) {
         const int row = wv * (MF_ROWS / 4) + 4 * st + kg;
         const double b = sX[(size_t)ci * MF_LDS_STRIDE + row];
#pragma unroll
         for (int t = 0; t < NT; t++) {
            const double a = sV[(size_t)(16 * t + ci) * MF_LDS_STRIDE + row];
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
         }
      }
      __syncthreads();
   }
   /* add the four waves' tiles in a fixed order: red[wave][tile][reg][lane] in the staging buffer */
   double *red = mf_lds;
#pragma unroll
   for (int t = 0; t < NT; t++)
#pragma unroll
      for (int r = 0; r < 4; r++) red[(((size_t)wv * NT + t) * 4 + r) * 64 + lane] = acc[t][r];
   __syncthreads();
   if (wv == 0) {
      const size_t nout = (size_t)segs.total * nx;
#pragma unroll
      for (int t = 0; t < NT; t++)
#pragma unroll
         for (int r = 0; r < 4; r++) {
            const size_t o = ((size_t)t * 4 + r) * 64 + lane;
            const double v = (red[o] + red[(size_t)NT * 256 + o]) + (red[(size_t)2 * NT * 256 + o] + red[(size_t)3 * NT * 256 + o]);
            const int i = 16 * t + kg + 4 * r, j = ci;     /* C/D layout of v_mfma_f64_16x16x4_f64 */
            if (i < ncv && j < nxv)
               partials[(size_t)blockIdx.x * nout + (size_t)(j0 + i) + (size_t)(c0 + j) * segs.total] = v;
         }
   }
}

static int dots_mfma_enabled(void) {             /* HIPK_NO_MFMA: measurement knob, read once */
   static int v = -1;
   if (v < 0) v = getenv("HIPK_NO_MFMA") == NULL;
   return v;
}

template <typename T, int VW>
static void dots_launch(hipk_ctx *ctx, dim3 grid, int nxt, const SegArgs &sa, const T *X, int64_t ldX,
      int nx, int64_t m) {
   dim3 block(HIPK_BLOCK);
   switch (nxt) {
   case 1: hipLaunchKernelGGL((dots_kernel<T, 8, 1, VW>), grid, block, 0, ctx->stream, sa, X, ldX, nx, m, ctx->partials); break;
   case 2: hipLaunchKernelGGL((dots_kernel<T, 8, 2, VW>), grid, block, 0, ctx->stream, sa, X, ldX, nx, m, ctx->partials); break;
   case 4: hipLaunchKernelGGL((dots_kernel<T, 8, 4, VW>), grid, block, 0, ctx->stream, sa, X, ldX, nx, m, ctx->partials); break;
   default: hipLaunchKernelGGL((dots_kernel<T, 8, 8, VW>), grid, block, 0, ctx->stream, sa, X, ldX, nx, m, ctx->partials); break;
   }
}

template <typename T>
static int panel_dots_t(hipk_ctx *ctx, int64_t m, const SegArgs &sa, const T *X, int64_t ldX,
      int nx, double *out_dev, int ldout) {
   const int NC = 8;
   int nxt = nx <= 1 ? 1 : nx <= 2 ? 2 : nx <= 4 ? 4 : 8;
   const bool vec = segs_aligned16(sa, sizeof(T)) && aligned16(X, ldX, sizeof(T));
   const int VWm = vec ? (int)vecwidth<T>::value : 1;
   int gx = hipk_grid_for_rows(ctx, m / VWm + 1, HIPK_BLOCK * 2, 4);
   int gy = (sa.total + NC - 1) / NC;
   int gz = (nx + nxt - 1) / nxt;
   /* keep the chip full when the chunk grid is already wide */
   while (gx > 1 && (int64_t)gx * gy * gz > (int64_t)ctx->num_cu * 8) gx = (gx + 1) / 2;
   size_t nout = (size_t)sa.total * nx;
   /* blocks of right-hand columns against more than one chunk of basis columns: wave-per-chunk
    * kernel, the right-hand columns are read once per workgroup */
   const bool wide = (nx >= 4 && sa.total > NC && vec);
   if (wide) {
      const int WAVES = 4;
      gx = hipk_grid_for_rows(ctx, m / VWm + 1, 64 * 4, 8);
      gy = (sa.total + NC * WAVES - 1) / (NC * WAVES);
      nxt = (nx <= 4) ? 4 : 8;
      gz = (nx + nxt - 1) / nxt;
   }
   /* matrix cores for blocks of >= 4 right-hand columns (16-byte aligned panels) */
   const bool mfma = (nx >= 4 && vec && dots_mfma_enabled());
   int nt = 1;
   if (mfma) {
      nt = sa.total <= 16 ? 1 : 2;                 /* 2 tiles: 48 staged columns = 50 KB of LDS */
      gy = (sa.total + 16 * nt - 1) / (16 * nt);
      gz = (nx + 15) / 16;
      static int mbpc = -1;                         /* HIPK_MFMA_BPC: workgroups per CU (measurement knob, read once) */
      if (mbpc < 0) { const char *e = getenv("HIPK_MFMA_BPC"); mbpc = e ? atoi(e) : 3; if (mbpc < 1) mbpc = 3; }      /* 3: 0.50 / 0.71 / 0.66 of HBM at 8 / 16 / 24 basis columns, 2: 0.44 / 0.67 / 0.63 (profiles/r06_zpanel_perf.txt) */
      gx = hipk_grid_for_rows(ctx, m, MF_ROWS, mbpc);
      while (gx > 1 && (int64_t)gx * gy * gz > (int64_t)ctx->num_cu * 2 * mbpc) gx = (gx + 1) / 2;
   }
   if (hipk_reserve_partials(ctx, (size_t)gx * nout)) return -2;
   dim3 grid(gx, gy, gz);
   const int pslot = hipk_prof_begin(HIPK_PROF_DOTS, ctx->stream, (double)m * sizeof(T) * (sa.total + nx));
   if (mfma) {
      const size_t shm = (size_t)(16 * nt + 16) * MF_LDS_STRIDE * sizeof(double);
      if (nt == 1) hipLaunchKernelGGL((dots_mfma_kernel<T, 1>), grid, dim3(HIPK_BLOCK), shm, ctx->stream, sa, X, ldX, nx, m, ctx->partials);
      else hipLaunchKernelGGL((dots_mfma_kernel<T, 2>), grid, dim3(HIPK_BLOCK), shm, ctx->stream, sa, X, ldX, nx, m, ctx->partials);
   } else if (wide) {
      if (nxt == 4) hipLaunchKernelGGL((dots_wide_kernel<T, 8, 4, vecwidth<T>::value, 4>), grid, dim3(256), 0, ctx->stream, sa, X, ldX, nx, m, ctx->partials);
      else hipLaunchKernelGGL((dots_wide_kernel<T, 8, 8, vecwidth<T>::value, 4>), grid, dim3(256), 0, ctx->stream, sa, X, ldX, nx, m, ctx->partials);
   } else if (vec) dots_launch<T, vecwidth<T>::value>(ctx, grid, nxt, sa, X, ldX, nx, m);
   else dots_launch<T, 1>(ctx, grid, nxt, sa, X, ldX, nx, m);
   hipk_prof_end(pslot, ctx->stream);
   HIPK_CHECK(hipGetLastError());
   hipLaunchKernelGGL(finalize_ld_kernel, dim3((unsigned)nout), dim3(HIPK_BLOCK), 0, ctx->stream,
         ctx->partials, gx, (int)nout, sa.total, ldout, out_dev, hipk_mirror_of(ctx, out_dev), hipk_next_flag(ctx, out_dev));
   HIPK_CHECK(hipGetLastError());
   return 0;
}

extern "C" int hipk_panel_dots(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const hipk_seg *segs,
      int nseg, const void *X, int64_t ldX, int nx, double *out_dev, int ldout) {
   if (HIPK_IS_Z(dt)) return hipk_z_panel_dots(ctx, dt, m, segs, nseg, X, ldX, nx, out_dev, ldout);
   SegArgs sa;
   if (pack_segs(segs, nseg, &sa)) return -1;
   if (sa.total == 0 || nx <= 0) return 0;
   if (ldout < sa.total) return -1;
   switch (dt) {
   case HIPK_F64: return panel_dots_t<double>(ctx, m, sa, (const double *)X, ldX, nx, out_dev, ldout);
   case HIPK_F32: return panel_dots_t<float>(ctx, m, sa, (const float *)X, ldX, nx, out_dev, ldout);
   default: return -44;
   }
}

/* ===================== NN-accumulate: project + norms ========================= */
#define PROJ_MAXCOLS 192
#define FIN_TAIL_MAXBLOCK 1024
template <typename T, int NX, int VW>
__global__ void __launch_bounds__(HIPK_BLOCK)
project_kernel(SegArgs segs, const double *__restrict__ coef, int ldcoef, T *X,
      int64_t ldX, T *Xout, int64_t ldXout, int nx, int c0, int64_t m, double *__restrict__ partials, hipk_fin_args fa) {
   typedef lanevec<T, VW> LV;
   __shared__ int s_last;
   __shared__ double scoef[PROJ_MAXCOLS * NX];
   __shared__ const T *sptr[PROJ_MAXCOLS];
   const int total = segs.total;
   const int nxv = min(NX, nx - c0);
   for (int t = threadIdx.x; t < total * NX; t += HIPK_BLOCK) {
      int j = t / NX, c = t % NX;
      scoef[t] = (c < nxv) ? coef[j + (size_t)(c0 + c) * ldcoef] : 0.0;
   }
   for (int j = threadIdx.x; j < total; j += HIPK_BLOCK) sptr[j] = seg_col<T>(segs, j);
   __syncthreads();

   T *xp = X + (size_t)c0 * (size_t)ldX;
   T *op = Xout + (size_t)c0 * (size_t)ldXout;      /* == xp for the in-place form */
   double n2[NX];
#pragma unroll
   for (int c = 0; c < NX; c++) n2[c] = 0.0;

   const int64_t stride = (int64_t)gridDim.x * HIPK_BLOCK;
   const int64_t mg = m / VW;
   for (int64_t g = (int64_t)blockIdx.x * HIPK_BLOCK + threadIdx.x; g < mg; g += stride) {
      double xv[NX][VW];
#pragma unroll
      for (int c = 0; c < NX; c++) {
         if (c < nxv) {
            LV t = ((const LV *)(xp + (size_t)c * ldX))[g];
#pragma unroll
            for (int r = 0; r < VW; r++) xv[c][r] = (double)t.e[r];
         } else {
#pragma unroll
            for (int r = 0; r < VW; r++) xv[c][r] = 0.0;
         }
      }
      int j = 0;
      for (; j + 8 <= total; j += 8) {
         LV a[8];
#pragma unroll
         for (int u = 0; u < 8; u++) a[u] = ldstream<T, VW, 2>(sptr[j + u], g);
#pragma unroll
         for (int u = 0; u < 8; u++)
#pragma unroll
            for (int c = 0; c < NX; c++) {
               const double cf = scoef[(j + u) * NX + c];
#pragma unroll
               for (int r = 0; r < VW; r++) xv[c][r] = fma(-(double)a[u].e[r], cf, xv[c][r]);
            }
      }
      for (; j < total; j++) {
         LV a = ((const LV *)sptr[j])[g];
#pragma unroll
         for (int c = 0; c < NX; c++) {
            const double cf = scoef[j * NX + c];
#pragma unroll
            for (int r = 0; r < VW; r++) xv[c][r] = fma(-(double)a.e[r], cf, xv[c][r]);
         }
      }
#pragma unroll
      for (int c = 0; c < NX; c++)
         if (c < nxv) {
            LV o;
#pragma unroll
            for (int r = 0; r < VW; r++) {
               o.e[r] = (T)xv[c][r];
               n2[c] = fma((double)o.e[r], (double)o.e[r], n2[c]);
            }
            ((LV *)(op + (size_t)c * ldXout))[g] = o;
         }
   }
   if (VW > 1 && blockIdx.x == 0 && threadIdx.x < (unsigned)(m - mg * VW)) {   /* ragged tail rows */
      const int64_t i = mg * VW + threadIdx.x;
      for (int c = 0; c < nxv; c++) {
         double v = (double)xp[i + (size_t)c * ldX];
         for (int j = 0; j < total; j++) v = fma(-(double)sptr[j][i], scoef[j * NX + c], v);
         T o = (T)v;
         op[i + (size_t)c * ldXout] = o;
#pragma unroll
         for (int cc = 0; cc < NX; cc++) if (cc == c) n2[cc] = fma((double)o, (double)o, n2[cc]);
      }
   }
   if (partials) {
      __shared__ double sm[HIPK_BLOCK / HIPK_WAVE][NX];
      const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
      for (int c = 0; c < NX; c++) {
         double v = hipk_wave_sum(n2[c]);
         if (lane == 0) sm[wv][c] = v;
      }
      __syncthreads();
      if (threadIdx.x < nxv)
         hipk_pstore(fa, partials + (size_t)(c0 + threadIdx.x) * gridDim.x + blockIdx.x,
               (sm[0][threadIdx.x] + sm[1][threadIdx.x]) + (sm[2][threadIdx.x] + sm[3][threadIdx.x]));
      hipk_inkernel_finalize(partials, nx, gridDim.x, fa, &s_last);
   }
}

/* X <- (X - [segs] coef) * M in ONE pass over X and the basis (M: nx x nx, nx <= NX <= 8): the
 * update and the right-multiplication of a CholQR / SVQB sweep (reference Num_ortho_kernel,
 * ortho.c:963-1072: two GEMMs and a copy through an m x b temporary). */
template <typename T, int NX, int VW>
__global__ void __launch_bounds__(HIPK_BLOCK)
project_mul_kernel(SegArgs segs, const double *__restrict__ coef, int ldcoef, const double *__restrict__ Mr,
      T *X, int64_t ldX, int nx, int64_t m) {
   typedef lanevec<T, VW> LV;
   __shared__ double scoef[PROJ_MAXCOLS * NX];
   __shared__ double sM[NX * NX];
   __shared__ const T *sptr[PROJ_MAXCOLS];
   const int total = segs.total;
   for (int t = threadIdx.x; t < total * NX; t += HIPK_BLOCK) {
      int j = t / NX, c = t % NX;
      scoef[t] = (c < nx) ? coef[j + (size_t)c * ldcoef] : 0.0;
   }
   for (int t = threadIdx.x; t < NX * NX; t += HIPK_BLOCK) {
      int i = t % NX, c = t / NX;                      /* sM[i + c*NX] = M(i, c) */
      sM[t] = (i < nx && c < nx) ? Mr[i + (size_t)c * nx] : 0.0;
   }
   for (int j = threadIdx.x; j < total; j += HIPK_BLOCK) sptr[j] = seg_col<T>(segs, j);
   __syncthreads();
   const int64_t stride = (int64_t)gridDim.x * HIPK_BLOCK;
   const int64_t mg = m / VW;
   for (int64_t g = (int64_t)blockIdx.x * HIPK_BLOCK + threadIdx.x; g < mg; g += stride) {
      double xv[NX][VW];
#pragma unroll
      for (int c = 0; c < NX; c++) {
         const LV t = ((const LV *)(X + (size_t)(c < nx ? c : 0) * ldX))[g];
#pragma unroll
         for (int r = 0; r < VW; r++) xv[c][r] = c < nx ? (double)t.e[r] : 0.0;
      }
      int j = 0;
      for (; j + 8 <= total; j += 8) {
         LV a[8];
#pragma unroll
         for (int u = 0; u < 8; u++) a[u] = ((const LV *)sptr[j + u])[g];
#pragma unroll
         for (int u = 0; u < 8; u++)
#pragma unroll
            for (int c = 0; c < NX; c++) {
               const double cf = scoef[(j + u) * NX + c];
#pragma unroll
               for (int r = 0; r < VW; r++) xv[c][r] = fma(-(double)a[u].e[r], cf, xv[c][r]);
            }
      }
      for (; j < total; j++) {
         LV a = ((const LV *)sptr[j])[g];
#pragma unroll
         for (int c = 0; c < NX; c++) {
            const double cf = scoef[j * NX + c];
#pragma unroll
            for (int r = 0; r < VW; r++) xv[c][r] = fma(-(double)a.e[r], cf, xv[c][r]);
         }
      }
#pragma unroll
      for (int c = 0; c < NX; c++)
         if (c < nx) {
            LV o;
#pragma unroll
            for (int r = 0; r < VW; r++) {
               double sacc = 0.0;
#pragma unroll
               for (int i = 0; i < NX; i++) sacc = fma(xv[i][r], sM[i + c * NX], sacc);
               o.e[r] = (T)sacc;
            }
            ((LV *)(X + (size_t)c * ldX))[g] = o;
         }
   }
   if (VW > 1 && blockIdx.x == 0 && threadIdx.x < (unsigned)(m - mg * VW)) {   /* ragged tail rows */
      const int64_t i = mg * VW + threadIdx.x;
      double xv[NX];
      for (int c = 0; c < NX; c++) {
         double v = c < nx ? (double)X[i + (size_t)c * ldX] : 0.0;
         for (int j = 0; j < total; j++) v = fma(-(double)sptr[j][i], scoef[j * NX + c], v);
         xv[c] = v;
      }
      for (int c = 0; c < nx; c++) {
         double sacc = 0.0;
         for (int q = 0; q < NX; q++) sacc = fma(xv[q], sM[q + c * NX], sacc);
         X[i + (size_t)c * ldX] = (T)sacc;
      }
   }
}

template <typename T, int VW>
static int panel_project_mul_v(hipk_ctx *ctx, int64_t m, const SegArgs &sa, const double *coef, int ldcoef,
      const double *Mr, T *X, int64_t ldX, int nx) {
   int gx = hipk_grid_for_rows(ctx, m / VW + 1, HIPK_BLOCK * 2, 4);
   dim3 block(HIPK_BLOCK);
   const int pslot = hipk_prof_begin(HIPK_PROF_PROJECT, ctx->stream, (double)m * sizeof(T) * ((double)sa.total + 2.0 * nx));
   if (nx <= 2) hipLaunchKernelGGL((project_mul_kernel<T, 2, VW>), dim3(gx), block, 0, ctx->stream, sa, coef, ldcoef, Mr, X, ldX, nx, m);
   else if (nx <= 4) hipLaunchKernelGGL((project_mul_kernel<T, 4, VW>), dim3(gx), block, 0, ctx->stream, sa, coef, ldcoef, Mr, X, ldX, nx, m);
   else hipLaunchKernelGGL((project_mul_kernel<T, 8, VW>), dim3(gx), block, 0, ctx->stream, sa, coef, ldcoef, Mr, X, ldX, nx, m);
   hipk_prof_end(pslot, ctx->stream);
   HIPK_CHECK(hipGetLastError());
   return 0;
}

/* X <- (X - [segs] coef) M; returns 1 when the shape is not covered (caller uses the two-pass form) */
extern "C" int hipk_panel_project_mul(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const hipk_seg *segs, int nseg,
      const double *coef_dev, int ldcoef, const double *M_dev, void *X, int64_t ldX, int nx) {
   if (HIPK_IS_Z(dt)) return hipk_z_panel_project(ctx, dt, m, segs, nseg, coef_dev, ldcoef, M_dev, X, ldX, X, ldX, nx, NULL);
   SegArgs sa;
   if (pack_segs(segs, nseg, &sa)) return -1;
   if (nx <= 0) return 0;
   if (nx > 8 || sa.total > PROJ_MAXCOLS) return 1;
   const size_t es = dt == HIPK_F64 ? 8 : 4;
   const bool vec = segs_aligned16(sa, es) && aligned16(X, ldX, es);
   switch (dt) {
   case HIPK_F64: return vec ? panel_project_mul_v<double, 2>(ctx, m, sa, coef_dev, ldcoef, M_dev, (double *)X, ldX, nx)
                             : panel_project_mul_v<double, 1>(ctx, m, sa, coef_dev, ldcoef, M_dev, (double *)X, ldX, nx);
   case HIPK_F32: return vec ? panel_project_mul_v<float, 4>(ctx, m, sa, coef_dev, ldcoef, M_dev, (float *)X, ldX, nx)
                             : panel_project_mul_v<float, 1>(ctx, m, sa, coef_dev, ldcoef, M_dev, (float *)X, ldX, nx);
   default: return -44;
   }
}

template <typename T, int VW>
static int panel_project_v(hipk_ctx *ctx, int64_t m, const SegArgs &sa, const double *coef,
      int ldcoef, T *X, int64_t ldX, T *Xout, int64_t ldXout, int nx, double *nrm2_dev) {
   int gx = hipk_grid_for_rows(ctx, m / VW + 1, HIPK_BLOCK * 2, 4);
   if (nrm2_dev && hipk_reserve_partials(ctx, (size_t)gx * nx)) return -2;
   dim3 block(HIPK_BLOCK);
   /* one launch covers all columns: its last workgroup finalises the norms itself */
   hipk_fin_args fa;
   memset(&fa, 0, sizeof(fa));
   /* the tail of a block-size-1 iteration (hipk_tail_defer): the partial sums of |t|^2 stay where the operator launch and
    * hipk_tail_finish add them up themselves — no second-stage launch here */
   const bool defer = nrm2_dev && nx == 1 && (ctx->tail_want & HIPK_TAIL_NORM) && ctx->tail_np2 == 0 && gx <= HIPK_TAIL_MAXPART;
   if (!defer && nrm2_dev && (nx == 1 || nx == 2 || nx == 4 || nx == 8)) fa = hipk_make_fin(ctx, nrm2_dev, HIPK_FIN_PROJECT, gx, nx);
   double *part = defer ? ctx->tailp : (nrm2_dev ? ctx->partials : NULL);      /* after hipk_make_fin: it may have grown the buffer */
   const int pslot = hipk_prof_begin(HIPK_PROF_PROJECT, ctx->stream, (double)m * sizeof(T) * ((double)sa.total * ((nx + 7) / 8) + 2.0 * nx));
   for (int c0 = 0; c0 < nx;) {
      int rem = nx - c0;
      int step;
      if (rem >= 8) { step = 8; hipLaunchKernelGGL((project_kernel<T, 8, VW>), dim3(gx), block, 0, ctx->stream, sa, coef, ldcoef, X, ldX, Xout, ldXout, nx, c0, m, part, fa); }
      else if (rem >= 4) { step = 4; hipLaunchKernelGGL((project_kernel<T, 4, VW>), dim3(gx), block, 0, ctx->stream, sa, coef, ldcoef, X, ldX, Xout, ldXout, nx, c0, m, part, fa); }
      else if (rem >= 2) { step = 2; hipLaunchKernelGGL((project_kernel<T, 2, VW>), dim3(gx), block, 0, ctx->stream, sa, coef, ldcoef, X, ldX, Xout, ldXout, nx, c0, m, part, fa); }
      else { step = 1; hipLaunchKernelGGL((project_kernel<T, 1, VW>), dim3(gx), block, 0, ctx->stream, sa, coef, ldcoef, X, ldX, Xout, ldXout, nx, c0, m, part, fa); }
      HIPK_CHECK(hipGetLastError());
      c0 += step;
   }
   hipk_prof_end(pslot, ctx->stream);
   if (defer) { ctx->tail_np2 = gx; ctx->tail_norm2_out = nrm2_dev; return 0; }
   if (nrm2_dev && !fa.enabled) return hipk_finalize_partials_t(ctx, ctx->partials, gx, nx, nrm2_dev);
   return 0;
}

template <typename T>
static int panel_project_t(hipk_ctx *ctx, int64_t m, const SegArgs &sa, const double *coef,
      int ldcoef, T *X, int64_t ldX, T *Xout, int64_t ldXout, int nx, double *nrm2_dev) {
   if (sa.total > PROJ_MAXCOLS) return -1;
   if (segs_aligned16(sa, sizeof(T)) && aligned16(X, ldX, sizeof(T)) && aligned16(Xout, ldXout, sizeof(T)))
      return panel_project_v<T, vecwidth<T>::value>(ctx, m, sa, coef, ldcoef, X, ldX, Xout, ldXout, nx, nrm2_dev);
   return panel_project_v<T, 1>(ctx, m, sa, coef, ldcoef, X, ldX, Xout, ldXout, nx, nrm2_dev);
}

extern "C" int hipk_panel_project_to(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const hipk_seg *segs,
      int nseg, const double *coef_dev, int ldcoef, const void *X, int64_t ldX, void *Xout, int64_t ldXout,
      int nx, double *nrm2_dev) {
   if (HIPK_IS_Z(dt)) return hipk_z_panel_project(ctx, dt, m, segs, nseg, coef_dev, ldcoef, NULL, X, ldX, Xout, ldXout, nx, nrm2_dev);
   SegArgs sa;
   if (pack_segs(segs, nseg, &sa)) return -1;
   if (nx <= 0) return 0;
   if (sa.total > PROJ_MAXCOLS) {
      /* more columns than one launch stages in LDS: project window by window (the operation
       * is a sum over columns); the first window reads X, the later ones update Xout in place;
       * the norms come from the last window */
      const size_t es = (dt == HIPK_F64) ? 8 : 4;
      for (int w0 = 0; w0 < sa.total; w0 += PROJ_MAXCOLS) {
         const int wn = sa.total - w0 < PROJ_MAXCOLS ? sa.total - w0 : PROJ_MAXCOLS;
         hipk_seg sub[HIPK_MAX_SEGS];
         int ns = 0, c = 0;
         for (int q = 0; q < HIPK_MAX_SEGS; q++) {
            const int lo = w0 > c ? w0 : c, hi = (w0 + wn < c + sa.n[q]) ? w0 + wn : c + sa.n[q];
            if (hi > lo) {
               sub[ns].base = (char *)sa.base[q] + (size_t)(lo - c) * (size_t)sa.ld[q] * es;
               sub[ns].ld = sa.ld[q]; sub[ns].ncols = hi - lo; ns++;
            }
            c += sa.n[q];
         }
         int rc = hipk_panel_project_to(ctx, dt, m, sub, ns, coef_dev + w0, ldcoef, w0 == 0 ? X : Xout, w0 == 0 ? ldX : ldXout,
               Xout, ldXout, nx, (w0 + wn >= sa.total) ? nrm2_dev : NULL);
         if (rc) return rc;
      }
      return 0;
   }
   switch (dt) {
   case HIPK_F64: return panel_project_t<double>(ctx, m, sa, coef_dev, ldcoef, (double *)X, ldX, (double *)Xout, ldXout, nx, nrm2_dev);
   case HIPK_F32: return panel_project_t<float>(ctx, m, sa, coef_dev, ldcoef, (float *)X, ldX, (float *)Xout, ldXout, nx, nrm2_dev);
   default: return -44;
   }
}

extern "C" int hipk_panel_project(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const hipk_seg *segs,
      int nseg, const double *coef_dev, int ldcoef, void *X, int64_t ldX, int nx,
      double *nrm2_dev) {
   return hipk_panel_project_to(ctx, dt, m, segs, nseg, coef_dev, ldcoef, X, ldX, X, ldX, nx, nrm2_dev);
}

/* ================= fused Ritz / residual / restart update ===================== */
#define RITZ_MAXOUT 80   /* per list (XV, XW) */
#define RITZ_MAXRES 16
struct RitzArgs {
   int nxv, nxw, nres;
   unsigned char xv_col[RITZ_MAXOUT], xw_col[RITZ_MAXOUT], res_col[RITZ_MAXRES];
   short res_slot[RITZ_MAXRES];
   void *xv_dst[RITZ_MAXOUT], *xw_dst[RITZ_MAXOUT], *res_dst[RITZ_MAXRES];
};

/* PRE: load the V row AND the W row before computing anything (2*NK registers): twice the
 * loads in flight per lane, which is what makes this kernel bandwidth- instead of latency-bound
 * (measured 3.0 -> see profiles/; used for NK <= 32).  !PRE: two phases sharing one row buffer. */
template <typename T, int NK, int NR, bool PRE>
__global__ void __launch_bounds__(HIPK_BLOCK)
ritz_kernel(const T *__restrict__ V, const T *__restrict__ W, int64_t ld, int k,
      const double *__restrict__ h, int ldh, int nh, const double *__restrict__ theta,
      RitzArgs ja, int64_t m, double *__restrict__ partials, int nslots) {
   extern __shared__ double hs[];   /* hs[c*NK + j], zero padded rows j >= k */
   for (int t = threadIdx.x; t < NK * nh; t += HIPK_BLOCK) {
      int c = t / NK, j = t % NK;
      hs[t] = (j < k) ? h[j + (size_t)c * ldh] : 0.0;
   }
   __syncthreads();
   double th[NR];
#pragma unroll
   for (int r = 0; r < NR; r++) th[r] = (r < ja.nres) ? theta[ja.res_col[r]] : 0.0;
   double n2[NR];
#pragma unroll
   for (int r = 0; r < NR; r++) n2[r] = 0.0;
   const bool needW = (ja.nxw > 0 || ja.nres > 0);

   const int64_t stride = (int64_t)gridDim.x * HIPK_BLOCK;
   for (int64_t i = (int64_t)blockIdx.x * HIPK_BLOCK + threadIdx.x; i < m; i += stride) {
      double row[NK];
      double roww[PRE ? NK : 1];
      double xres[NR];
#pragma unroll
      for (int j = 0; j < NK; j++) row[j] = (j < k) ? (double)ldstream1<T, 4>(V + i + (size_t)j * ld) : 0.0;
      if (PRE && needW) {
#pragma unroll
         for (int j = 0; j < NK; j++) roww[PRE ? j : 0] = (j < k) ? (double)ldstream1<T, 4>(W + i + (size_t)j * ld) : 0.0;
      }
#pragma unroll
      for (int r = 0; r < NR; r++) {
         xres[r] = 0.0;
         if (r < ja.nres) {
            const double *hc = hs + (int)ja.res_col[r] * NK;
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < NK; j++) s = fma(row[j], hc[j], s);
            xres[r] = s;
         }
      }
      for (int o = 0; o < ja.nxv; o++) {
         const double *hc = hs + (int)ja.xv_col[o] * NK;
         double s = 0.0;
#pragma unroll
         for (int j = 0; j < NK; j++) s = fma(row[j], hc[j], s);
         ((T *)ja.xv_dst[o])[i] = (T)s;
      }
      if (needW) {
         if (!PRE) {
#pragma unroll
            for (int j = 0; j < NK; j++) row[j] = (j < k) ? (double)ldstream1<T, 4>(W + i + (size_t)j * ld) : 0.0;
         }
         for (int o = 0; o < ja.nxw; o++) {
            const double *hc = hs + (int)ja.xw_col[o] * NK;
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < NK; j++) s = fma(PRE ? roww[PRE ? j : 0] : row[j], hc[j], s);
            ((T *)ja.xw_dst[o])[i] = (T)s;
         }
#pragma unroll
         for (int r = 0; r < NR; r++)
            if (r < ja.nres) {
               const double *hc = hs + (int)ja.res_col[r] * NK;
               double s = 0.0;
#pragma unroll
               for (int j = 0; j < NK; j++) s = fma(PRE ? roww[PRE ? j : 0] : row[j], hc[j], s);
               T res = (T)fma(-th[r], xres[r], s);
               if (ja.res_dst[r]) ((T *)ja.res_dst[r])[i] = res;
               n2[r] = fma((double)res, (double)res, n2[r]);
            }
      }
   }
   if (nslots > 0) {
      __shared__ double sm[HIPK_BLOCK / HIPK_WAVE][NR];
      const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
      for (int r = 0; r < NR; r++) {
         double v = hipk_wave_sum(n2[r]);
         if (lane == 0) sm[wv][r] = v;
      }
      __syncthreads();
      if (threadIdx.x < ja.nres && ja.res_slot[threadIdx.x] >= 0)
         partials[(size_t)blockIdx.x * nslots + ja.res_slot[threadIdx.x]] =
               (sm[0][threadIdx.x] + sm[1][threadIdx.x]) + (sm[2][threadIdx.x] + sm[3][threadIdx.x]);
   }
}


/* General basis size (64 < k <= 255; unrestarted or user-chosen large maxBasisSize, e.g. the
 * reference's tests/tests/test_001 with maxBasisSize 140).  Workgroup = one wave, 32 rows per
 * tile; the tile's V (then W) rows are staged in LDS ([k][32] doubles = k/4 KB), so the update
 * is still in place and every column is read exactly once.  The two half-waves split the
 * outputs.  Not a roofline kernel: with k this large the small dense eigenproblem on the host
 * dominates anyway. */
#define RITZ_BIG_ROWS 32          /* rows per tile for k <= 255; 16 for k <= 511, 8 for k <= 1023 (round 6): k x rows doubles of LDS <= 64 KB */
#define RITZ_BIG_MAXOUT 1024
#define RITZ_BIG_MAXK 1023
struct RitzBigArgs {       /* output lists live in device memory (ctx->jobtab) */
   int nxv, nxw, nres;
   const uint16_t *xv_col, *xw_col;
   void *const *xv_dst, *const *xw_dst;
   uint16_t res_col[RITZ_MAXRES];
   short res_slot[RITZ_MAXRES];
   void *res_dst[RITZ_MAXRES];
};
/* One wave per workgroup; a tile of R rows of all k columns of V (then W) sits in LDS, the 64 lanes are R rows x 64/R PARTS,
 * part p takes the columns p, p + 64/R, ... when the tile is filled and the outputs p, p + 64/R, ... when it is multiplied.
 * R = 32 is the round-1 kernel (k <= 255: restarts of wide bases, maxBasisSize > 64); R = 16 / 8 lift the basis to 511 / 1023
 * columns on the same 64 KB (the reference has no limit, primme_c.c:470-487: VERDICT r05 Missing #3). */
template <typename T, int R>
__global__ void __launch_bounds__(64)
ritz_big_kernel(const T *__restrict__ V, const T *__restrict__ W, int64_t ld, int k,
      const double *__restrict__ h, int ldh, const double *__restrict__ theta, RitzBigArgs ja,
      int64_t m, double *__restrict__ partials, int nslots) {
   extern __shared__ double tile[];                /* tile[j*R + r] */
   __shared__ double xres[RITZ_MAXRES][R];
   constexpr int NP = 64 / R;
   __shared__ double n2s[RITZ_MAXRES][NP];
   const int lane = threadIdx.x, r = lane % R, part = lane / R;
   const bool needW = (ja.nxw > 0 || ja.nres > 0);
   double n2[RITZ_MAXRES];
#pragma unroll
   for (int q = 0; q < RITZ_MAXRES; q++) n2[q] = 0.0;
   const int64_t ntiles = (m + R - 1) / R;
   for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
      const int64_t i = t * R + r;
      const bool live = i < m;
      __syncthreads();
      for (int j = part; j < k; j += NP) tile[j * R + r] = live ? (double)V[i + (size_t)j * ld] : 0.0;
      __syncthreads();
      for (int o = part; o < ja.nxv + ja.nres; o += NP) {
         const bool isres = o >= ja.nxv;
         const int col = isres ? ja.res_col[o - ja.nxv] : ja.xv_col[o];
         const double *hc = h + (size_t)col * ldh;
         double s = 0.0;
         for (int j = 0; j < k; j++) s = fma(tile[j * R + r], hc[j], s);
         if (isres) xres[o - ja.nxv][r] = s;
         else if (live) ((T *)ja.xv_dst[o])[i] = (T)s;
      }
      if (!needW) continue;
      __syncthreads();
      for (int j = part; j < k; j += NP) tile[j * R + r] = live ? (double)W[i + (size_t)j * ld] : 0.0;
      __syncthreads();
      for (int o = part; o < ja.nxw + ja.nres; o += NP) {
         const bool isres = o >= ja.nxw;
         const int q = o - ja.nxw;
         const int col = isres ? ja.res_col[q] : ja.xw_col[o];
         const double *hc = h + (size_t)col * ldh;
         double s = 0.0;
         for (int j = 0; j < k; j++) s = fma(tile[j * R + r], hc[j], s);
         if (!isres) { if (live) ((T *)ja.xw_dst[o])[i] = (T)s; continue; }
         T res = (T)fma(-theta[col], xres[q][r], s);
         if (live) {
            if (ja.res_dst[q]) ((T *)ja.res_dst[q])[i] = res;
#pragma unroll
            for (int qq = 0; qq < RITZ_MAXRES; qq++) if (qq == q) n2[qq] = fma((double)res, (double)res, n2[qq]);
         }
      }
   }
   if (nslots > 0) {
      /* residual q was accumulated by part (nxw + q) % NP only: the other parts add zeros */
#pragma unroll
      for (int q = 0; q < RITZ_MAXRES; q++) {
         double v = n2[q];
         for (int off = R / 2; off > 0; off >>= 1) v += __shfl_down(v, off, R);
         if (r == 0) n2s[q][part] = v;
      }
      __syncthreads();
      if (lane < ja.nres && ja.res_slot[lane] >= 0) {
         double v = 0.0;
         for (int pp = 0; pp < NP; pp++) v += n2s[lane][pp];
         partials[(size_t)blockIdx.x * nslots + ja.res_slot[lane]] = v;
      }
   }
}

template <typename T, int NK, bool PRE>
static int ritz_launch_nk(hipk_ctx *ctx, int64_t m, const T *V, const T *W, int64_t ld, int k,
      const double *h, int ldh, int nh, const double *theta, const RitzArgs &ja, int gx,
      int nslots) {
   size_t shm = (size_t)NK * nh * sizeof(double);
   if (ja.nres <= 4)
      hipLaunchKernelGGL((ritz_kernel<T, NK, 4, PRE>), dim3(gx), dim3(HIPK_BLOCK), shm, ctx->stream, V, W, ld, k, h, ldh, nh, theta, ja, m, ctx->partials, nslots);
   else
      hipLaunchKernelGGL((ritz_kernel<T, NK, RITZ_MAXRES, PRE>), dim3(gx), dim3(HIPK_BLOCK), shm, ctx->stream, V, W, ld, k, h, ldh, nh, theta, ja, m, ctx->partials, nslots);
   HIPK_CHECK(hipGetLastError());
   return 0;
}

template <typename T>
static int ritz_dispatch(hipk_ctx *ctx, int64_t m, const T *V, const T *W, int64_t ld, int k,
      const double *h, int ldh, int nh, const double *theta, const RitzArgs &ja, int gx, int nslots) {
   if (k <= 8) return ritz_launch_nk<T, 8, true>(ctx, m, V, W, ld, k, h, ldh, nh, theta, ja, gx, nslots);
   if (k <= 16) return ritz_launch_nk<T, 16, true>(ctx, m, V, W, ld, k, h, ldh, nh, theta, ja, gx, nslots);
   if (k <= 24) return ritz_launch_nk<T, 24, true>(ctx, m, V, W, ld, k, h, ldh, nh, theta, ja, gx, nslots);
   if (k <= 32) return ritz_launch_nk<T, 32, true>(ctx, m, V, W, ld, k, h, ldh, nh, theta, ja, gx, nslots);
   if (k <= 48) return ritz_launch_nk<T, 48, false>(ctx, m, V, W, ld, k, h, ldh, nh, theta, ja, gx, nslots);
   if (k <= 64) return ritz_launch_nk<T, 64, false>(ctx, m, V, W, ld, k, h, ldh, nh, theta, ja, gx, nslots);
   return -1;
}

template <typename T>
static int ritz_update_t(hipk_ctx *ctx, int64_t m, const T *V, const T *W, int64_t ld, int k,
      const double *h, int ldh, const double *theta, const hipk_job *jobs, int njobs,
      double *nrm2_dev) {
   /* host-side job table; the small kernels take it by value, the general one from HBM */
   struct BigTab { uint16_t xv_col[RITZ_BIG_MAXOUT], xw_col[RITZ_BIG_MAXOUT]; void *xv_dst[RITZ_BIG_MAXOUT], *xw_dst[RITZ_BIG_MAXOUT]; };
   std::vector<char> tab_store(sizeof(BigTab));
   BigTab &tab = *(BigTab *)tab_store.data();
   RitzBigArgs jb_;
   memset(&jb_, 0, sizeof(jb_));
   int nh = 0, nslots = 0;
   for (int q = 0; q < njobs; q++) {
      const hipk_job &jb = jobs[q];
      if (jb.col < 0 || jb.col > RITZ_BIG_MAXK) return -1;
      if (jb.col + 1 > nh) nh = jb.col + 1;
      if (jb.kind == HIPK_JOB_XV) {
         if (jb_.nxv >= RITZ_BIG_MAXOUT) return -1;
         tab.xv_col[jb_.nxv] = (uint16_t)jb.col; tab.xv_dst[jb_.nxv++] = jb.dst;
      } else if (jb.kind == HIPK_JOB_XW) {
         if (jb_.nxw >= RITZ_BIG_MAXOUT) return -1;
         tab.xw_col[jb_.nxw] = (uint16_t)jb.col; tab.xw_dst[jb_.nxw++] = jb.dst;
      } else if (jb.kind == HIPK_JOB_RES) {
         if (jb_.nres >= RITZ_MAXRES) return -1;
         jb_.res_col[jb_.nres] = (uint16_t)jb.col; jb_.res_dst[jb_.nres] = jb.dst;
         jb_.res_slot[jb_.nres++] = (short)jb.slot;
         if (jb.slot + 1 > nslots) nslots = jb.slot + 1;
      } else return -1;
   }
   if (k <= 0 || njobs <= 0) return 0;
   if (nslots > 0 && !nrm2_dev) return -1;
   const bool small = (k <= 64 && jb_.nxv <= RITZ_MAXOUT && jb_.nxw <= RITZ_MAXOUT);
   const int big_rows = k <= 255 ? RITZ_BIG_ROWS : (k <= 511 ? 16 : 8);
   int gx = small ? hipk_grid_for_rows(ctx, m, HIPK_BLOCK * 2, 4) : hipk_grid_for_rows(ctx, m, big_rows, 8);
   if (nslots > 0) {
      /* every block writes every slot; slots must be 0..nslots-1, each used once */
      if (hipk_reserve_partials(ctx, (size_t)gx * nslots)) return -2;
   }
   const int pslot = hipk_prof_begin(HIPK_PROF_RITZ, ctx->stream,
         (double)m * sizeof(T) * ((double)k * (1 + ((jb_.nxw > 0 || jb_.nres > 0) ? 1 : 0)) + jb_.nxv + jb_.nxw + jb_.nres));
   int rc;
   if (small) {
      RitzArgs ja;
      memset(&ja, 0, sizeof(ja));
      ja.nxv = jb_.nxv; ja.nxw = jb_.nxw; ja.nres = jb_.nres;
      for (int q = 0; q < jb_.nxv; q++) { ja.xv_col[q] = (unsigned char)tab.xv_col[q]; ja.xv_dst[q] = tab.xv_dst[q]; }
      for (int q = 0; q < jb_.nxw; q++) { ja.xw_col[q] = (unsigned char)tab.xw_col[q]; ja.xw_dst[q] = tab.xw_dst[q]; }
      for (int q = 0; q < RITZ_MAXRES; q++) ja.res_col[q] = (unsigned char)jb_.res_col[q];
      memcpy(ja.res_slot, jb_.res_slot, sizeof(ja.res_slot));
      memcpy(ja.res_dst, jb_.res_dst, sizeof(ja.res_dst));
      rc = ritz_dispatch<T>(ctx, m, V, W, ld, k, h, ldh, nh, theta, ja, gx, nslots);
   } else {
      const size_t shm = (size_t)k * big_rows * sizeof(double);
      if (k > RITZ_BIG_MAXK || shm > 64 * 1024) return -1;
      if (!ctx->jobtab) HIPK_CHECK(hipMalloc(&ctx->jobtab, sizeof(BigTab)));
      /* the table is small and this is the rare path: staged through the context's pinned buffer and complete on return
       * (since round 3 no runtime copy touches memory the library did not pin itself) */
      if (hipk_upload(ctx, ctx->jobtab, &tab, sizeof(BigTab))) return -1;
      char *dt_ = (char *)ctx->jobtab;
      jb_.xv_col = (const uint16_t *)(dt_ + offsetof(BigTab, xv_col));
      jb_.xw_col = (const uint16_t *)(dt_ + offsetof(BigTab, xw_col));
      jb_.xv_dst = (void *const *)(dt_ + offsetof(BigTab, xv_dst));
      jb_.xw_dst = (void *const *)(dt_ + offsetof(BigTab, xw_dst));
      if (big_rows == 32) hipLaunchKernelGGL((ritz_big_kernel<T, 32>), dim3(gx), dim3(64), shm, ctx->stream, V, W, ld, k, h, ldh, theta, jb_, m, ctx->partials, nslots);
      else if (big_rows == 16) hipLaunchKernelGGL((ritz_big_kernel<T, 16>), dim3(gx), dim3(64), shm, ctx->stream, V, W, ld, k, h, ldh, theta, jb_, m, ctx->partials, nslots);
      else hipLaunchKernelGGL((ritz_big_kernel<T, 8>), dim3(gx), dim3(64), shm, ctx->stream, V, W, ld, k, h, ldh, theta, jb_, m, ctx->partials, nslots);
      HIPK_CHECK(hipGetLastError());
      rc = 0;
   }
   hipk_prof_end(pslot, ctx->stream);
   if (rc) return rc;
   if (nslots > 0) return hipk_finalize_partials(ctx, ctx->partials, gx, nslots, nrm2_dev);
   return 0;
}

extern "C" int hipk_ritz_update(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const void *V,
      const void *W, int64_t ldVW, int k, const double *h_dev, int ldh,
      const double *theta_dev, const hipk_job *jobs, int njobs, double *nrm2_dev) {
   if (HIPK_IS_Z(dt)) return hipk_z_ritz_update(ctx, dt, m, V, W, ldVW, k, h_dev, ldh, theta_dev, jobs, njobs, nrm2_dev);
   switch (dt) {
   case HIPK_F64: return ritz_update_t<double>(ctx, m, (const double *)V, (const double *)W, ldVW, k, h_dev, ldh, theta_dev, jobs, njobs, nrm2_dev);
   case HIPK_F32: return ritz_update_t<float>(ctx, m, (const float *)V, (const float *)W, ldVW, k, h_dev, ldh, theta_dev, jobs, njobs, nrm2_dev);
   default: return -44;
   }
}

/* ============ restart update + residual of the next candidate + its overlaps ============
 * The restart pass (X = V h_c, Y = W h_c, any destinations) that also forms the residual r of ONE candidate
 * and, in registers, its inner products with the basis it is writing: out = [ (V h)' r | Q' r | r' r |
 * (W h)' r | W(:,k-1)' Q ] for the first nb XV / XW outputs -- what hipk_ritz_residual_overlaps delivers for the
 * old basis, here for the restarted one, so that the iteration after a restart needs no pass of its own
 * (reference: Num_update_VWXR in restart.c:1233-1294, then ortho.c:236-246 and update_projection.c:99-122 on the
 * restarted basis).  One lane per row like ritz_kernel (the whole V and W row in registers before the first
 * store, so the update may be in place); NB / QM bound the accumulators a lane carries. */
template <typename T, int NK, int NB, int QM>
__global__ void __launch_bounds__(HIPK_BLOCK)
ritz_ov_kernel(const T *__restrict__ V, const T *__restrict__ W, int64_t ld, int k,
      const double *__restrict__ h, int ldh, int nh, const double *__restrict__ theta,
      RitzArgs ja, int nb, const T *__restrict__ Q, int64_t ldQ, int L, int64_t m, double *__restrict__ partials) {
   extern __shared__ double hs[];   /* hs[c*NK + j], zero padded rows j >= k */
   constexpr int QN = QM > 0 ? QM : 1;
   for (int t = threadIdx.x; t < NK * nh; t += HIPK_BLOCK) {
      int c = t / NK, j = t % NK;
      hs[t] = (j < k) ? h[j + (size_t)c * ldh] : 0.0;
   }
   __syncthreads();
   const int rcol = ja.res_col[0];
   const double th = theta[rcol];
   const double *hr = hs + rcol * NK;
   T *rdst = (T *)ja.res_dst[0];
   const T *wlast = W + (size_t)(k - 1) * ld;
   double ov[NB], ow[NB], oq[QN], og[QN], n2 = 0.0;
#pragma unroll
   for (int o = 0; o < NB; o++) { ov[o] = 0.0; ow[o] = 0.0; }
#pragma unroll
   for (int l = 0; l < QN; l++) { oq[l] = 0.0; og[l] = 0.0; }

   const int64_t stride = (int64_t)gridDim.x * HIPK_BLOCK;
   for (int64_t i = (int64_t)blockIdx.x * HIPK_BLOCK + threadIdx.x; i < m; i += stride) {
      double row[NK], roww[NK], qv[QN];
#pragma unroll
      for (int j = 0; j < NK; j++) row[j] = (j < k) ? (double)ldstream1<T, 4>(V + i + (size_t)j * ld) : 0.0;
#pragma unroll
      for (int j = 0; j < NK; j++) roww[j] = (j < k) ? (double)ldstream1<T, 4>(W + i + (size_t)j * ld) : 0.0;
      double wl = 0.0;
      if (QM > 0) {
#pragma unroll
         for (int l = 0; l < QN; l++) qv[l] = (l < L) ? (double)Q[i + (size_t)l * ldQ] : 0.0;
         wl = (double)wlast[i];
      }
      double xr = 0.0, yr = 0.0;
#pragma unroll
      for (int j = 0; j < NK; j++) xr = fma(row[j], hr[j], xr);
#pragma unroll
      for (int j = 0; j < NK; j++) yr = fma(roww[j], hr[j], yr);
      const T res = (T)fma(-th, xr, yr);
      const double r = (double)res;
      n2 = fma(r, r, n2);
      /* the new basis columns: stored and multiplied with r as stored */
#pragma unroll
      for (int o = 0; o < NB; o++)
         if (o < nb) {
            const double *hv = hs + (int)ja.xv_col[o] * NK;
            const double *hw = hs + (int)ja.xw_col[o] * NK;
            double sv = 0.0, sw = 0.0;
#pragma unroll
            for (int j = 0; j < NK; j++) sv = fma(row[j], hv[j], sv);
#pragma unroll
            for (int j = 0; j < NK; j++) sw = fma(roww[j], hw[j], sw);
            const T tv = (T)sv, tw = (T)sw;
            ststream1<T, 8>((T *)ja.xv_dst[o] + i, tv);
            ststream1<T, 8>((T *)ja.xw_dst[o] + i, tw);
            ov[o] = fma((double)tv, r, ov[o]);
            ow[o] = fma((double)tw, r, ow[o]);
         }
      for (int o = nb; o < ja.nxv; o++) {
         const double *hc = hs + (int)ja.xv_col[o] * NK;
         double sv = 0.0;
#pragma unroll
         for (int j = 0; j < NK; j++) sv = fma(row[j], hc[j], sv);
         ststream1<T, 8>((T *)ja.xv_dst[o] + i, (T)sv);
      }
      for (int o = nb; o < ja.nxw; o++) {
         const double *hc = hs + (int)ja.xw_col[o] * NK;
         double sw = 0.0;
#pragma unroll
         for (int j = 0; j < NK; j++) sw = fma(roww[j], hc[j], sw);
         ststream1<T, 8>((T *)ja.xw_dst[o] + i, (T)sw);
      }
      if (rdst) rdst[i] = res;
      if (QM > 0) {
#pragma unroll
         for (int l = 0; l < QN; l++) { oq[l] = fma(qv[l], r, oq[l]); og[l] = fma(wl, qv[l], og[l]); }
      }
   }
   /* block sums, partial-major: partials[block * nsl + slot], slots [ (Vh)'r | Q'r | r'r | (Wh)'r | W(:,k-1)'Q ] */
   __shared__ double sm[HIPK_BLOCK / HIPK_WAVE][2 * NB + 2 * QN + 1];
   const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
   const int nsl = 2 * nb + 2 * L + 1;
#pragma unroll
   for (int o = 0; o < NB; o++) {
      const double a = hipk_wave_sum(ov[o]), b = hipk_wave_sum(ow[o]);
      if (lane == 0 && o < nb) { sm[wv][o] = a; sm[wv][nb + L + 1 + o] = b; }
   }
#pragma unroll
   for (int l = 0; l < QN; l++) {
      const double a = hipk_wave_sum(oq[l]), b = hipk_wave_sum(og[l]);
      if (lane == 0 && QM > 0 && l < L) { sm[wv][nb + l] = a; sm[wv][2 * nb + L + 1 + l] = b; }
   }
   { const double a = hipk_wave_sum(n2); if (lane == 0) sm[wv][nb + L] = a; }
   __syncthreads();
   if ((int)threadIdx.x < nsl)
      partials[(size_t)blockIdx.x * nsl + threadIdx.x] =
            (sm[0][threadIdx.x] + sm[1][threadIdx.x]) + (sm[2][threadIdx.x] + sm[3][threadIdx.x]);
}

template <typename T, int NK, int NB>
static int ritz_ov_q(hipk_ctx *ctx, int gx, size_t shm, const T *V, const T *W, int64_t ld, int k, const double *h, int ldh, int nh,
      const double *theta, const RitzArgs &ja, int nb, const T *Q, int64_t ldQ, int L, int64_t m) {
#define ROV(QMV) hipLaunchKernelGGL((ritz_ov_kernel<T, NK, NB, QMV>), dim3(gx), dim3(HIPK_BLOCK), shm, ctx->stream, V, W, ld, k, h, ldh, nh, theta, ja, nb, Q, ldQ, L, m, ctx->partials)
   if (L == 0) ROV(0);
   else if (L <= 8) ROV(8);
   else if (L <= 16) ROV(16);
   else if (L <= 32) ROV(32);
   else return -1;
#undef ROV
   HIPK_CHECK(hipGetLastError());
   return 0;
}

template <typename T>
static int ritz_ov_t(hipk_ctx *ctx, int64_t m, const T *V, const T *W, int64_t ld, int k, const double *h, int ldh,
      const double *theta, const hipk_job *jobs, int njobs, double *nrm2_dev, int nb, const T *Q, int64_t ldQ, int L,
      double *ov_dev) {
   if (k <= 0 || k > 32 || nb <= 0 || nb > 16 || L < 0 || L > 32 || !ov_dev) return -1;
   RitzArgs ja;
   memset(&ja, 0, sizeof(ja));
   int nh = 0, res_slot = -1;
   for (int q = 0; q < njobs; q++) {
      const hipk_job &jb = jobs[q];
      if (jb.col < 0 || jb.col > 255) return -1;
      if (jb.col + 1 > nh) nh = jb.col + 1;
      if (jb.kind == HIPK_JOB_XV) {
         if (ja.nxv >= RITZ_MAXOUT) return -1;
         ja.xv_col[ja.nxv] = (unsigned char)jb.col; ja.xv_dst[ja.nxv++] = jb.dst;
      } else if (jb.kind == HIPK_JOB_XW) {
         if (ja.nxw >= RITZ_MAXOUT) return -1;
         ja.xw_col[ja.nxw] = (unsigned char)jb.col; ja.xw_dst[ja.nxw++] = jb.dst;
      } else if (jb.kind == HIPK_JOB_RES) {
         if (ja.nres >= 1) return -1;             /* one candidate */
         ja.res_col[0] = (unsigned char)jb.col; ja.res_dst[0] = jb.dst; ja.res_slot[0] = (short)jb.slot; ja.nres = 1;
         res_slot = jb.slot;
      } else return -1;
   }
   if (ja.nres != 1 || ja.nxv < nb || ja.nxw < nb) return -1;
   const int nsl = 2 * nb + 2 * L + 1;
   const int gx = hipk_grid_for_rows(ctx, m, HIPK_BLOCK * 2, 4);
   if (hipk_reserve_partials(ctx, (size_t)gx * nsl)) return -2;
   const int pslot = hipk_prof_begin(HIPK_PROF_RITZ, ctx->stream, (double)m * sizeof(T) * (2.0 * k + L + ja.nxv + ja.nxw + (ja.res_dst[0] ? 1 : 0)));
   int rc;
   if (k <= 16) {
      const size_t shm = (size_t)16 * nh * sizeof(double);
      rc = nb <= 8 ? ritz_ov_q<T, 16, 8>(ctx, gx, shm, V, W, ld, k, h, ldh, nh, theta, ja, nb, Q, ldQ, L, m)
                   : ritz_ov_q<T, 16, 16>(ctx, gx, shm, V, W, ld, k, h, ldh, nh, theta, ja, nb, Q, ldQ, L, m);
   } else {
      const size_t shm = (size_t)32 * nh * sizeof(double);
      rc = nb <= 8 ? ritz_ov_q<T, 32, 8>(ctx, gx, shm, V, W, ld, k, h, ldh, nh, theta, ja, nb, Q, ldQ, L, m)
                   : ritz_ov_q<T, 32, 16>(ctx, gx, shm, V, W, ld, k, h, ldh, nh, theta, ja, nb, Q, ldQ, L, m);
   }
   hipk_prof_end(pslot, ctx->stream);
   if (rc) return rc;
   if (nrm2_dev && res_slot >= 0) {
      rc = hipk_finalize_partials_strided(ctx, ctx->partials + (nb + L), gx, nsl, 1, nrm2_dev + res_slot);
      if (rc) return rc;
   }
   return hipk_finalize_partials_strided(ctx, ctx->partials, gx, nsl, nsl, ov_dev);
}

extern "C" int hipk_ritz_update_overlaps(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const void *V, const void *W,
      int64_t ldVW, int k, const double *h_dev, int ldh, const double *theta_dev, const hipk_job *jobs, int njobs,
      double *nrm2_dev, int nbasis, const void *Q, int64_t ldQ, int L, double *ov_dev) {
   switch (dt) {
   case HIPK_F64: return ritz_ov_t<double>(ctx, m, (const double *)V, (const double *)W, ldVW, k, h_dev, ldh, theta_dev, jobs, njobs, nrm2_dev, nbasis, (const double *)Q, ldQ, L, ov_dev);
   case HIPK_F32: return ritz_ov_t<float>(ctx, m, (const float *)V, (const float *)W, ldVW, k, h_dev, ldh, theta_dev, jobs, njobs, nrm2_dev, nbasis, (const float *)Q, ldQ, L, ov_dev);
   default: return -44;
   }
}

/* ============ fused Ritz residual + first-pass Gram-Schmidt overlaps (b = 1) ============
 * r = W h - theta V h (written to dst), out = [ V' r | Q' r | r' r ] (+ [ W' r | W(:,k-1)' Q ]).
 * In Generalized Davidson without preconditioner the residual IS the new basis vector, and
 * the V and W rows needed for r are exactly the rows the first classical Gram-Schmidt pass
 * would stream again for V' r (reference: Num_update_VWXR, auxiliary_eigs_normal.c:155-388,
 * followed by Num_gemv_ddh in ortho.c:236-246).  Fusing them removes one full pass over V
 * per outer iteration; the locked vectors Q are streamed here instead of in the dots launch.
 *
 * Column-split layout: the four waves of a workgroup walk the SAME rows (64*VW per step); wave w
 * owns the basis columns [w*CPW, (w+1)*CPW) of V and W and the locked columns [w*QPW, (w+1)*QPW).
 * Each wave forms its part of x = V h and y = W h, the parts meet in LDS (one barrier per step,
 * double-buffered), every wave then knows r for its rows and accumulates the overlaps of ITS
 * columns only.  A lane therefore holds 2*CPW + QPW loaded values and as many accumulators
 * instead of 2k + L of each (206 VGPRs at k = 16, L = 16 in the one-lane-per-row form, i.e. two
 * waves per SIMD): occupancy no longer falls with the basis size, every column is still read
 * exactly once, and the W' r accumulators that make the projection column free
 * cost CPW registers instead of k.
 */
struct HCol { double h[32]; };   /* the coefficient vector travels in the kernel arguments */

template <typename T, int CPW, int QPW, int VW, bool WT>
__global__ void __launch_bounds__(HIPK_BLOCK)
ritz_cgs_kernel(const T *__restrict__ V, const T *__restrict__ W, int64_t ld, int k,
      HCol hcol, double theta, const double *__restrict__ hdev, T *__restrict__ dst, const T *__restrict__ Q,
      int64_t ldQ, int L, int64_t m, double *__restrict__ partials, int blocked, hipk_fin_args fa) {
   typedef lanevec<T, VW> LV;
   constexpr int QN = QPW > 0 ? QPW : 1;
   __shared__ double sxy[2][2][VW][4][64];       /* [buffer][x|y][row in lane][wave][lane] */
   __shared__ int s_last;
   const int lane = threadIdx.x & 63;
   const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
   const int j0 = wv * CPW, q0 = wv * QPW;
   /* hdev != NULL: the coefficient vector and the Ritz value were left in HBM by hipk_rr_arrow (the iteration that was
    * enqueued before the host had seen the previous one); a status other than 0 means there is no valid pair: nothing to do */
   /* (no valid pair: nothing is streamed and nothing is written to dst, but the partial sums ARE stored — zeros — so that the
    * second stage behind this launch, and whatever was enqueued behind that, work on defined numbers) */
   bool dead = false;
   if (hdev) {
      dead = hdev[33] != 0.0;
      theta = hdev[32];
   }
   double hj[CPW];
   const T *vp[CPW], *wp[CPW], *qp[QN];
#pragma unroll
   for (int jj = 0; jj < CPW; jj++) {
      const int j = j0 + jj;
      hj[jj] = (j < k) ? (hdev ? hdev[j < 32 ? j : 0] : hcol.h[j < 32 ? j : 0]) : 0.0;
      vp[jj] = V + (size_t)(j < k ? j : 0) * ld;
      wp[jj] = W + (size_t)(j < k ? j : 0) * ld;
   }
#pragma unroll
   for (int qq = 0; qq < QN; qq++) qp[qq] = (QPW > 0 && L > 0) ? Q + (size_t)(q0 + qq < L ? q0 + qq : 0) * ldQ : (const T *)V;
   const T *wlast = W + (size_t)(k - 1) * ld;    /* newest W column, for W(:,k-1)' Q */
   double ov[CPW], ow[WT ? CPW : 1], oq[QN], og[WT ? QN : 1], n2 = 0.0;
#pragma unroll
   for (int jj = 0; jj < CPW; jj++) ov[jj] = 0.0;
#pragma unroll
   for (int jj = 0; jj < (WT ? CPW : 1); jj++) ow[jj] = 0.0;
#pragma unroll
   for (int qq = 0; qq < QN; qq++) oq[qq] = 0.0;
#pragma unroll
   for (int qq = 0; qq < (WT ? QN : 1); qq++) og[qq] = 0.0;

   const int64_t ngroups = m / (64 * VW);         /* full steps of 64*VW rows */
   int buf = 0;
   /* blocked: each workgroup walks one contiguous range of rows (sequential DRAM pages per column
    * stream) instead of striding through the panel with the whole grid */
   const int64_t gpb = (ngroups + gridDim.x - 1) / gridDim.x;
   const int64_t gbeg = blocked ? (int64_t)blockIdx.x * gpb : blockIdx.x;
   const int64_t gend = dead ? gbeg : (blocked ? (gbeg + gpb < ngroups ? gbeg + gpb : ngroups) : ngroups);
   const int64_t gstep = blocked ? 1 : gridDim.x;
   for (int64_t g = gbeg; g < gend; g += gstep, buf ^= 1) {
      const int64_t e = g * 64 + lane;            /* index in LV units */
      LV v[CPW], w[CPW], q[QN], wl;
#pragma unroll
      for (int jj = 0; jj < CPW; jj++)
         if (j0 + jj < k) { v[jj] = ldstream<T, VW, 2>(vp[jj], e); w[jj] = ldstream<T, VW, 1>(wp[jj], e); }
      if (QPW > 0) {
#pragma unroll
         for (int qq = 0; qq < QN; qq++)
            if (q0 + qq < L) q[qq] = ldstream<T, VW, 2>(qp[qq], e);
         if (WT && q0 < L) wl = ldstream<T, VW, 1>(wlast, e);
      }
      double px[VW], py[VW];
#pragma unroll
      for (int r = 0; r < VW; r++) { px[r] = 0.0; py[r] = 0.0; }
#pragma unroll
      for (int jj = 0; jj < CPW; jj++)
         if (j0 + jj < k) {
#pragma unroll
            for (int r = 0; r < VW; r++) {
               px[r] = fma((double)v[jj].e[r], hj[jj], px[r]);
               py[r] = fma((double)w[jj].e[r], hj[jj], py[r]);
            }
         }
#pragma unroll
      for (int r = 0; r < VW; r++) { sxy[buf][0][r][wv][lane] = px[r]; sxy[buf][1][r][wv][lane] = py[r]; }
      __syncthreads();
      double rr[VW];
      LV rt;
#pragma unroll
      for (int r = 0; r < VW; r++) {
         const double x = (sxy[buf][0][r][0][lane] + sxy[buf][0][r][1][lane]) + (sxy[buf][0][r][2][lane] + sxy[buf][0][r][3][lane]);
         const double y = (sxy[buf][1][r][0][lane] + sxy[buf][1][r][1][lane]) + (sxy[buf][1][r][2][lane] + sxy[buf][1][r][3][lane]);
         rt.e[r] = (T)fma(-theta, x, y);
         rr[r] = (double)rt.e[r];
      }
      if (wv == 0) {
         ((LV *)dst)[e] = rt;
#pragma unroll
         for (int r = 0; r < VW; r++) n2 = fma(rr[r], rr[r], n2);
      }
#pragma unroll
      for (int jj = 0; jj < CPW; jj++)
         if (j0 + jj < k) {
#pragma unroll
            for (int r = 0; r < VW; r++) {
               ov[jj] = fma((double)v[jj].e[r], rr[r], ov[jj]);
               if (WT) ow[WT ? jj : 0] = fma((double)w[jj].e[r], rr[r], ow[WT ? jj : 0]);
            }
         }
      if (QPW > 0) {
#pragma unroll
         for (int qq = 0; qq < QN; qq++)
            if (q0 + qq < L) {
#pragma unroll
               for (int r = 0; r < VW; r++) {
                  oq[qq] = fma((double)q[qq].e[r], rr[r], oq[qq]);
                  if (WT) og[WT ? qq : 0] = fma((double)wl.e[r], (double)q[qq].e[r], og[WT ? qq : 0]);
               }
            }
      }
   }
   /* Ragged tail (fewer than 64*VW rows): ONE more step of the same shape with element-wise guarded loads, done by
    * the workgroup that has the fewest full steps (with the strided assignment the first one behind the remainder):
    * as two one-row-per-lane steps on the last workgroup it was the tail of the whole launch -- 132.6 us against
    * 122.6 us for 31 k more rows that divide evenly (k = 15, L = 10, m = 2 000 250). */
   const int tailwg = blocked ? (int)gridDim.x - 1 : (int)(ngroups % gridDim.x);
   if ((int)blockIdx.x == tailwg && ngroups * 64 * VW < m && !dead) {
      const int64_t i0 = ngroups * 64 * VW + (int64_t)lane * VW;
      double tv[CPW][VW], tw[CPW][VW], tq[QN][VW], twl[VW], px[VW], py[VW];
      bool live[VW];
#pragma unroll
      for (int r = 0; r < VW; r++) { live[r] = i0 + r < m; px[r] = 0.0; py[r] = 0.0; twl[r] = 0.0; }
#pragma unroll
      for (int jj = 0; jj < CPW; jj++) {
#pragma unroll
         for (int r = 0; r < VW; r++) {
            const bool on = live[r] && (j0 + jj < k);
            tv[jj][r] = on ? (double)vp[jj][i0 + r] : 0.0;
            tw[jj][r] = on ? (double)wp[jj][i0 + r] : 0.0;
         }
      }
#pragma unroll
      for (int qq = 0; qq < QN; qq++)
#pragma unroll
         for (int r = 0; r < VW; r++) tq[qq][r] = (QPW > 0 && live[r] && q0 + qq < L) ? (double)qp[qq][i0 + r] : 0.0;
      if (WT && QPW > 0 && q0 < L) {
#pragma unroll
         for (int r = 0; r < VW; r++) twl[r] = live[r] ? (double)wlast[i0 + r] : 0.0;
      }
#pragma unroll
      for (int jj = 0; jj < CPW; jj++)
#pragma unroll
         for (int r = 0; r < VW; r++) { px[r] = fma(tv[jj][r], hj[jj], px[r]); py[r] = fma(tw[jj][r], hj[jj], py[r]); }
#pragma unroll
      for (int r = 0; r < VW; r++) { sxy[buf][0][r][wv][lane] = px[r]; sxy[buf][1][r][wv][lane] = py[r]; }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < VW; r++) {
         const double x = (sxy[buf][0][r][0][lane] + sxy[buf][0][r][1][lane]) + (sxy[buf][0][r][2][lane] + sxy[buf][0][r][3][lane]);
         const double y = (sxy[buf][1][r][0][lane] + sxy[buf][1][r][1][lane]) + (sxy[buf][1][r][2][lane] + sxy[buf][1][r][3][lane]);
         const T rt = (T)fma(-theta, x, y);
         const double rr = live[r] ? (double)rt : 0.0;
         if (wv == 0 && live[r]) { dst[i0 + r] = rt; n2 = fma(rr, rr, n2); }
#pragma unroll
         for (int jj = 0; jj < CPW; jj++) {
            ov[jj] = fma(tv[jj][r], rr, ov[jj]);
            if (WT) ow[WT ? jj : 0] = fma(tw[jj][r], rr, ow[WT ? jj : 0]);
         }
#pragma unroll
         for (int qq = 0; qq < QN; qq++) {
            oq[qq] = fma(tq[qq][r], rr, oq[qq]);
            if (WT) og[WT ? qq : 0] = fma(twl[r], tq[qq][r], og[WT ? qq : 0]);
         }
      }
   }
   /* every output belongs to exactly one wave: wave sums go straight to the block's partials, o-major
    * (partials[o * nblocks + block]), o over [ V'r (k) | Q'r (L) | r'r | W'r (k) | W(:,k-1)'Q (L) ] */
   const int nout = k + L + 1 + (WT ? k + L : 0);
   const size_t nb = gridDim.x;
   double *pcol = partials + blockIdx.x;
#pragma unroll
   for (int jj = 0; jj < CPW; jj++) {
      const double t = hipk_wave_sum(ov[jj]);
      if (lane == 0 && j0 + jj < k) hipk_pstore(fa, pcol + (size_t)(j0 + jj) * nb, t);
      if (WT) {
         const double u = hipk_wave_sum(ow[WT ? jj : 0]);
         if (lane == 0 && j0 + jj < k) hipk_pstore(fa, pcol + (size_t)(k + L + 1 + j0 + jj) * nb, u);
      }
   }
   if (QPW > 0) {
#pragma unroll
      for (int qq = 0; qq < QN; qq++) {
         const double t = hipk_wave_sum(oq[qq]);
         if (lane == 0 && q0 + qq < L) hipk_pstore(fa, pcol + (size_t)(k + q0 + qq) * nb, t);
         if (WT) {
            const double u = hipk_wave_sum(og[WT ? qq : 0]);
            if (lane == 0 && q0 + qq < L) hipk_pstore(fa, pcol + (size_t)(2 * k + L + 1 + q0 + qq) * nb, u);
         }
      }
   }
   { const double t = hipk_wave_sum(n2); if (lane == 0 && wv == 0) hipk_pstore(fa, pcol + (size_t)(k + L) * nb, t); }
   hipk_inkernel_finalize(partials, nout, gridDim.x, fa, &s_last);
}

static int rcgs_blocked(void) {               /* HIPK_RCGS_BLOCKED: measurement knob, read once */
   static int v = -1;
   if (v < 0) { const char *env = getenv("HIPK_RCGS_BLOCKED"); v = env ? atoi(env) : 0; }
   return v;
}

template <typename T, int CPW, int VW, bool WT>
static int ritz_cgs_q(hipk_ctx *ctx, int gx, const T *V, const T *W, int64_t ld, int k, const HCol &hcol,
      double theta, const double *hdev, T *dst, const T *Q, int64_t ldQ, int L, int64_t m, const hipk_fin_args &fa) {
   dim3 g(gx), b(HIPK_BLOCK);
#define RCGS(QPWV) hipLaunchKernelGGL((ritz_cgs_kernel<T, CPW, QPWV, VW, WT>), g, b, 0, ctx->stream, V, W, ld, k, hcol, theta, hdev, dst, Q, ldQ, L, m, ctx->partials, rcgs_blocked(), fa)
   if (L == 0) RCGS(0);
   else if (L <= 8) RCGS(2);
   else if (L <= 16) RCGS(4);
   else if (L <= 32) RCGS(8);
   else return -1;
#undef RCGS
   HIPK_CHECK(hipGetLastError());
   return 0;
}

template <typename T, int VW, bool WT>
static int ritz_cgs_k(hipk_ctx *ctx, int gx, const T *V, const T *W, int64_t ld, int k, const HCol &hcol,
      double theta, const double *hdev, T *dst, const T *Q, int64_t ldQ, int L, int64_t m, const hipk_fin_args &fa) {
   if (k <= 8) return ritz_cgs_q<T, 2, VW, WT>(ctx, gx, V, W, ld, k, hcol, theta, hdev, dst, Q, ldQ, L, m, fa);
   if (k <= 16) return ritz_cgs_q<T, 4, VW, WT>(ctx, gx, V, W, ld, k, hcol, theta, hdev, dst, Q, ldQ, L, m, fa);
   if (k <= 24) return ritz_cgs_q<T, 6, VW, WT>(ctx, gx, V, W, ld, k, hcol, theta, hdev, dst, Q, ldQ, L, m, fa);
   return ritz_cgs_q<T, 8, VW, WT>(ctx, gx, V, W, ld, k, hcol, theta, hdev, dst, Q, ldQ, L, m, fa);
}

template <typename T>
static int ritz_cgs_t(hipk_ctx *ctx, int64_t m, const T *V, const T *W, int64_t ld, int k,
      const double *hcol_host, double theta, const double *hdev, T *dst, const T *Q, int64_t ldQ, int L, int want_wtr, double *out_dev) {
   if (k <= 0 || k > 32 || L < 0 || L > 32) return -1;
   HCol hcol;
   for (int j = 0; j < 32; j++) hcol.h[j] = (j < k && hcol_host) ? hcol_host[j] : 0.0;
   constexpr int VWT = 2;   /* two rows per lane: 16-byte loads in double, 8-byte in float (register budget) */
   const bool vec = aligned16(V, ld, sizeof(T)) && aligned16(W, ld, sizeof(T)) && aligned16(dst, ld, sizeof(T)) &&
                    (L == 0 || aligned16(Q, ldQ, sizeof(T)));
   const int rows_per_step = 64 * (vec ? VWT : 1);
   static int bpc = -1;                        /* HIPK_RCGS_BPC: workgroups per CU (measurement knob, read once) */
   if (bpc < 0) { const char *env = getenv("HIPK_RCGS_BPC"); bpc = env ? atoi(env) : 2; if (bpc < 1) bpc = 2; }
   int gx = hipk_grid_for_rows(ctx, m, rows_per_step, bpc);
   const int nout = k + L + 1 + (want_wtr ? k + L : 0);
   if (hipk_reserve_partials(ctx, (size_t)gx * nout)) return -2;
   const int pslot = hipk_prof_begin(HIPK_PROF_RITZ, ctx->stream, (double)m * sizeof(T) * (2.0 * k + L + 1));
   int rc;
   const hipk_fin_args fa = hipk_make_fin(ctx, out_dev, HIPK_FIN_RITZ, gx, nout);
   if (vec) rc = want_wtr ? ritz_cgs_k<T, VWT, true>(ctx, gx, V, W, ld, k, hcol, theta, hdev, dst, Q, ldQ, L, m, fa)
                          : ritz_cgs_k<T, VWT, false>(ctx, gx, V, W, ld, k, hcol, theta, hdev, dst, Q, ldQ, L, m, fa);
   else rc = want_wtr ? ritz_cgs_k<T, 1, true>(ctx, gx, V, W, ld, k, hcol, theta, hdev, dst, Q, ldQ, L, m, fa)
                      : ritz_cgs_k<T, 1, false>(ctx, gx, V, W, ld, k, hcol, theta, hdev, dst, Q, ldQ, L, m, fa);
   hipk_prof_end(pslot, ctx->stream);
   if (rc) return rc;
   if (fa.enabled) return 0;                    /* the last workgroup finalised */
   return hipk_finalize_partials_t(ctx, ctx->partials, gx, nout, out_dev);
}

extern "C" int hipk_ritz_residual_overlaps(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const void *V,
      const void *W, int64_t ldVW, int k, const double *hcol_host, double theta, void *dst,
      const void *Q, int64_t ldQ, int L, int want_wtr, double *out_dev) {
   hipk_note_turnaround(ctx);
   switch (dt) {
   case HIPK_F64: return ritz_cgs_t<double>(ctx, m, (const double *)V, (const double *)W, ldVW, k, hcol_host, theta, NULL, (double *)dst, (const double *)Q, ldQ, L, want_wtr, out_dev);
   case HIPK_F32: return ritz_cgs_t<float>(ctx, m, (const float *)V, (const float *)W, ldVW, k, hcol_host, theta, NULL, (float *)dst, (const float *)Q, ldQ, L, want_wtr, out_dev);
   default: return -44;
   }
}
/* the same with the coefficient vector (hth_dev[0 .. k)), the Ritz value (hth_dev[32]) and a status word (hth_dev[33]) in HBM:
 * the residual pass of an iteration enqueued before the host has seen the previous one (hipk_rr_arrow) */
extern "C" int hipk_ritz_residual_overlaps_dev(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const void *V, const void *W, int64_t ldVW, int k,
      const double *hth_dev, void *dst, const void *Q, int64_t ldQ, int L, int want_wtr, double *out_dev) {
   if (!hth_dev) return -1;
   switch (dt) {
   case HIPK_F64: return ritz_cgs_t<double>(ctx, m, (const double *)V, (const double *)W, ldVW, k, NULL, 0.0, hth_dev, (double *)dst, (const double *)Q, ldQ, L, want_wtr, out_dev);
   case HIPK_F32: return ritz_cgs_t<float>(ctx, m, (const float *)V, (const float *)W, ldVW, k, NULL, 0.0, hth_dev, (float *)dst, (const float *)Q, ldQ, L, want_wtr, out_dev);
   default: return -44;
   }
}

/* ---- Rayleigh-Ritz step of the pre-enqueued iteration: one eigenpair of the arrowhead matrix (primme_amd_kernels.h) ----
 * One wave; lane j (< k) of every group of 16 lanes owns Ritz value j (all four groups compute the same, so every lane
 * holds the sums).  Secular equation in the coordinate mu = lambda - theta_o of the nearer pole o:
 *    g(mu) = (alpha - theta_o) - mu - sum_j z_j^2 / ((theta_j - theta_o) - mu),   strictly decreasing between two poles,
 * safeguarded Newton (a step that leaves the bracket is replaced by its midpoint), until the step is below two ulps of mu. */
/* sum over the 16 values a wave left in LDS (every lane gets the same bits: a fixed order) — no cross-lane traffic beyond
 * two LDS round trips per Newton step */
__device__ __forceinline__ double rr_sum16(const double *s) {
   double a = 0.0;
#pragma unroll
   for (int i = 0; i < 16; i++) a += s[i];
   return a;
}
/* what the Rayleigh-Ritz step keeps in LDS */
struct RrShared { double sY[256], sG[160], sTh[16], sF[128], s_v[16], s_w[16], s_y[16]; };
/* the step is run by ONE wave: as a launch of its own (a 64-thread workgroup: __syncthreads) or as the last part of the
 * one-workgroup launch that finishes an iteration's tail (wave 0 of a wider workgroup: the LDS traffic of a single wave is
 * ordered by itself, the fences keep the compiler from moving an access across the point) */
struct RrSyncBlock { __device__ static __forceinline__ void sync() { __syncthreads(); } };
struct RrSyncWave {
   __device__ static __forceinline__ void sync() {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
   }
};
/* lane = 0..63 of the one wave.  rr_arrow_stage: the step's inputs into LDS — independent of |t|^2 and t'At, so the launch that
 * first has to add those up issues these loads before it does; rr_arrow_solve: the step itself, n2 = |t|^2 and alpha = t'At of
 * the iteration whose reductions are in fov[0 .. nfov) */
__device__ __forceinline__ void rr_arrow_stage(const hipk_rr_in &in, const double *__restrict__ fov, int nfov, RrShared &S, const int lane) {
   /* The kernel arguments live in host-visible memory on this stack: indexed, per-lane reads of `in` would each be a trip over
    * PCIe (the first version of this kernel did that in its loops and took longer than the host round trip it replaces).
    * Everything is brought into LDS with ONE batch of loads — the arguments and this iteration's reductions together. */
   const int j = lane & 15;
   double ty[4], tg[3], tf[2];
#pragma unroll
   for (int u = 0; u < 4; u++) ty[u] = in.Y[lane + 64 * u];
#pragma unroll
   for (int u = 0; u < 3; u++) tg[u] = (lane + 64 * u < 160) ? in.G[lane + 64 * u] : 0.0;
   const double tt = in.theta[j];
#pragma unroll
   for (int u = 0; u < 2; u++) tf[u] = (lane + 64 * u < nfov) ? fov[lane + 64 * u] : 0.0;
#pragma unroll
   for (int u = 0; u < 4; u++) S.sY[lane + 64 * u] = ty[u];
#pragma unroll
   for (int u = 0; u < 3; u++) if (lane + 64 * u < 160) S.sG[lane + 64 * u] = tg[u];
   if (lane < 16) S.sTh[lane] = tt;
#pragma unroll
   for (int u = 0; u < 2; u++) S.sF[lane + 64 * u] = tf[u];
}
template <class SY>
__device__ __forceinline__ void rr_arrow_solve(const hipk_rr_in &in, int nfov, double n2, double alpha,
      RrShared &S, double *__restrict__ out, double *__restrict__ out_host, const int lane) {
   double *sY = S.sY, *sG = S.sG, *sTh = S.sTh, *sF = S.sF, *s_v = S.s_v, *s_w = S.s_w, *s_y = S.s_y;
   const int j = lane & 15, k = in.k, L = in.L;
   const bool on = lane < 16 && j < k;
   SY::sync();
   const double sgn = in.largest ? -1.0 : 1.0;
   const double nt = sqrt(n2);
   const double cv = on ? sF[j] : 0.0;                        /* V'r */
   const double wr = on ? sF[k + L + 1 + j] : 0.0;            /* W'r */
   double gq = 0.0;                                           /* (G Q'r)_j */
   for (int l = 0; l < L; l++) {
      const double g = (in.grow_row && j == k - 1) ? sF[2 * k + L + 1 + l] : sG[j + l * k];
      gq = on ? fma(g, sF[k + l], gq) : 0.0;
   }
   /* z_i = (Y(:,i)'(W'r - G Q'r) - theta_i Y(:,i)'(V'r)) / |t| */
   SY::sync();
   if (lane < 16) { s_v[lane] = on ? wr - gq : 0.0; s_w[lane] = cv; }
   SY::sync();
   double a1 = 0.0, a2 = 0.0;
   for (int r = 0; r < k; r++) { const double yv = sY[r + j * k]; a1 = fma(yv, s_v[r], a1); a2 = fma(yv, s_w[r], a2); }
   const double thj = on ? sTh[j] : 0.0;
   const double z = on ? sgn * (a1 - thj * a2) / nt : 0.0;     /* the negated problem for `largest`: -M = [-Theta -z; -z' -alpha] */
   const double th = sgn * thj, al = sgn * alpha;               /* ascending in j for both targets */
   const double z2 = z * z;
   const int c = in.cand;
   int status = (k < 1 || k > 16 || L < 0 || L > 10 || c < 0 || c > k || nfov > 126 || !(n2 > 0.0)) ? 1 : 0;
   SY::sync();
   if (lane < 16) { s_v[lane] = on ? th : 0.0; s_w[lane] = z2; }
   SY::sync();
   /* poles strictly increasing, everything finite (every lane looks at all of them: uniform control flow) */
   double zn2 = 0.0;
   for (int i = 0; i < k; i++) {
      const double ti = s_v[i], zi = s_w[i];
      if (!(isfinite(ti) && isfinite(zi)) || (i + 1 < k && !(ti < s_v[i + 1]))) status = status ? status : 2;
      zn2 += zi;
   }
   if (!isfinite(al)) status = status ? status : 2;
   const double zn = sqrt(zn2);
   double lam = 0.0, yj = 0.0, ynorm2 = 1.0;
   if (status == 0) {
      const double th0 = s_v[0], thl = s_v[k - 1];
      const double thc = s_v[c < k ? c : k - 1], thcm = s_v[c > 0 ? c - 1 : 0];
      int o;
      double lo, hi;
      if (c == 0) { o = 0; lo = fmin(0.0, al - th0) - zn - 1e-300; lo -= 4e-16 * fabs(lo); hi = 0.0; }
      else if (c == k) { o = k - 1; lo = 0.0; hi = fmax(0.0, al - thl) + zn + 1e-300; hi += 4e-16 * fabs(hi); }
      else {
         /* the sign of g at the middle of the interval says which pole the root is closer to */
         const double mid = 0.5 * (thc - thcm);
         double sm = 0.0;
         for (int i = 0; i < k; i++) sm += s_w[i] / ((s_v[i] - thcm) - mid);
         const double gm = (al - thcm) - mid - sm;
         if (gm > 0.0) { o = c; lo = -mid; hi = 0.0; }
         else { o = c - 1; lo = 0.0; hi = mid; }
      }
      const double tho = s_v[o];
      const double dj = th - tho, a0 = al - tho;
      const bool neg = hi == 0.0;                               /* the root lies below its pole (mu < 0) or above it (mu > 0) */
      const double B = s_w[o];                                  /* z_o^2: the pole's weight */
      const bool mine = on && j != o;
      double mu = 0.5 * (lo + hi);
      int it = 0;
      /* The pole at the origin is kept EXACT and the rest of the sum is replaced by its tangent at the current point
       * (what LAPACK's dlaed4 does with two poles): g(mu) ~ (a0 - S + S' mu_i) - (1 + S') mu + z_o^2 / mu, a quadratic in mu
       * whose root on the bracket's side of the pole is the next point.  Converges in 3-5 steps where Newton on g itself
       * crawls (the root of the wanted pair sits within 1e-8 of its pole: the first version of this loop averaged 55 steps,
       * 19 us per launch).  Safeguard: a point outside the bracket is replaced by its midpoint. */
      for (; it < 100; it++) {
         const double r = 1.0 / (dj - mu);
         const double t = mine ? z2 * r : 0.0;
         SY::sync();
         if (lane < 16) { s_y[lane] = t; sF[lane] = mine ? t * r : 0.0; }
         SY::sync();
         const double S = rr_sum16(s_y), Sp = rr_sum16(sF);
         const double pole = B / mu, g = a0 - mu - S + pole;
         if (!(g == g)) { status = 3; break; }
         /* converged: g is zero to the rounding of its own terms (going on from here only moves mu by an ulp — or throws the
          * next point an ulp outside the bracket, whose far end was never tightened, and the midpoint fall-back then crawls) */
         if (fabs(g) <= 2.3e-16 * (fabs(a0) + fabs(mu) + fabs(S) + fabs(pole))) break;
         if (g > 0.0) lo = mu; else hi = mu;
         const double A = 1.0 + Sp, Cc = a0 - S + Sp * mu, disc = sqrt(Cc * Cc + 4.0 * A * B);
         double mn;
         if (neg) mn = (Cc > 0.0) ? -2.0 * B / (Cc + disc) : (Cc - disc) / (2.0 * A);
         else mn = (Cc < 0.0) ? 2.0 * B / (disc - Cc) : (Cc + disc) / (2.0 * A);
         if (!(mn > lo && mn < hi)) { if (fabs(mn - mu) <= 1e-14 * fabs(mu)) break; mn = 0.5 * (lo + hi); }
         const double step = fabs(mn - mu);
         const bool done = step <= 4.4e-16 * fabs(mn) || mn == lo || mn == hi;
         mu = mn;
         if (done) break;
      }
      if (it >= 100) status = 4;
      lam = sgn * (tho + mu);
      yj = on ? z / (mu - dj) : 0.0;                              /* eigenvector [y; 1] of the (possibly negated) arrowhead */
      SY::sync();
      if (lane < 16) s_y[lane] = yj * yj;
      SY::sync();
      ynorm2 = 1.0 + rr_sum16(s_y);
      if (!isfinite(lam) || !isfinite(ynorm2)) status = 5;
   }
   /* back to the basis [V t]: h = [Y y; 1] / |[y; 1]| */
   const double inv = 1.0 / sqrt(ynorm2);
   SY::sync();
   if (lane < 16) s_y[lane] = yj;
   SY::sync();
   double hv = 0.0;
   for (int i = 0; i < k; i++) hv = fma(sY[j + i * k], s_y[i], hv);
   hv *= inv;
   if (lane < 16) {
      if (on) { out[j] = hv; if (out_host) out_host[j] = hv; }
      if (lane == 0) {
         out[k] = inv; out[32] = lam; out[33] = (double)status;
         if (out_host) { out_host[k] = inv; out_host[32] = lam; out_host[33] = (double)status; }
      }
   }
}

__global__ void __launch_bounds__(64)
rr_arrow_kernel(hipk_rr_in in, const double *__restrict__ fov, int nfov, const double *__restrict__ alpha_dev,
      double *__restrict__ out, double *__restrict__ out_host) {
   __shared__ RrShared S;
   const double n2 = fov[nfov], alpha = alpha_dev[0];
   rr_arrow_stage(in, fov, nfov, S, (int)threadIdx.x);
   rr_arrow_solve<RrSyncBlock>(in, nfov, n2, alpha, S, out, out_host, (int)threadIdx.x);
   /* no completion flag of its own: the host looks at the pinned copy after the flagged second stage of the residual pass that
    * follows in the stream (a kernel boundary on the queue lies in between) */
}
extern "C" int hipk_rr_arrow(hipk_ctx *ctx, const hipk_rr_in *in, const double *fov_dev, int nfov, const double *alpha_dev, double *out_dev) {
   if (!in || in->k < 1 || in->k > 16 || in->L < 0 || in->L > 10) return -1;
   if (nfov > 126) return -1;
   hipLaunchKernelGGL(rr_arrow_kernel, dim3(1), dim3(64), 0, ctx->stream, *in, fov_dev, nfov, alpha_dev, out_dev, hipk_mirror_of(ctx, out_dev));
   HIPK_CHECK(hipGetLastError());
   return 0;
}

/* ---- the ONE small launch that finishes the tail of a block-size-1 iteration (hipk_tail_defer / hipk_tail_finish) ----
 * np2 partial sums of |t|^2 (left by the Gram-Schmidt update; already added up, in the same order, by every workgroup of the
 * operator launch that normalised with them) and np3 partial sums of t'At (left by that operator launch) -> their sums in HBM
 * and in the pinned mirror; row-partitioned runs on the peer-to-peer transport exchange t'At with the other ranks here; then —
 * when the next iteration is enqueued behind this one — the Rayleigh-Ritz step of that iteration (rr_arrow_body: what
 * rr_arrow_kernel does as a launch of its own), and the completion flag last.  Replaces two second-stage launches and the
 * one-wave launch: three kernel boundaries and ~10 us less per outer iteration. */
template <bool XR>
__global__ void __launch_bounds__(FIN_TAIL_MAXBLOCK)
tail_finish_kernel(const double *__restrict__ p2, int np2, const double *__restrict__ p3, int np3, double *__restrict__ norm2_out,
      double *__restrict__ norm2_host, double *__restrict__ dot_out, double *__restrict__ dot_host, hipk_fin_flag fin, hipk_xr_dev xr,
      int do_rr, hipk_rr_in in, const double *__restrict__ fov, int nfov, double *__restrict__ rr_out, double *__restrict__ rr_host) {
   __shared__ double sm2[4], sm3[FIN_TAIL_MAXBLOCK / HIPK_WAVE];
   __shared__ RrShared S;
   /* the Rayleigh-Ritz step's inputs (3.4 KB of kernel arguments + this iteration's overlaps) do not depend on the two sums:
    * wave 0 has them on their way before anybody adds anything */
   if (do_rr && threadIdx.x < 64) rr_arrow_stage(in, fov, nfov, S, (int)threadIdx.x);
   if (np2 > 0) hipk_block_sum256_put(p2, np2, sm2);
   {  /* t'At: the order of hipk_finalize_kernel (one output) at the same workgroup size — the bits of the separate launch */
      const int nt = blockDim.x;
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
      int b = threadIdx.x;
      for (; b + 3 * nt < np3; b += 4 * nt) {
         const double a0 = p3[b], a1 = p3[b + nt], a2 = p3[b + 2 * nt], a3 = p3[b + 3 * nt];
         s0 += a0; s1 += a1; s2 += a2; s3 += a3;
      }
      {
         const int last = np3 - 1;
         const int b0 = b, b1 = b + nt, b2 = b + 2 * nt;
         const double a0 = p3[b0 < last ? b0 : last], a1 = p3[b1 < last ? b1 : last], a2 = p3[b2 < last ? b2 : last];
         s0 += b0 < np3 ? a0 : 0.0; s1 += b1 < np3 ? a1 : 0.0; s2 += b2 < np3 ? a2 : 0.0;
      }
      const double s = hipk_wave_sum((s0 + s1) + (s2 + s3));
      if ((threadIdx.x & 63) == 0) sm3[threadIdx.x >> 6] = s;
   }
   __syncthreads();
   if (threadIdx.x >= 64) return;
   double v3 = 0.0;
   {
      const int nw = blockDim.x >> 6;
      for (int w = 0; w < nw; w++) v3 += sm3[w];
   }
   const double v2 = np2 > 0 ? hipk_block_sum256_get(sm2) : 0.0;
   if (XR) v3 = hipk_xr_exchange(xr, 0u, v3, threadIdx.x < 16);
   if (threadIdx.x == 0) {
      dot_out[0] = v3;
      if (dot_host) dot_host[0] = v3;
      if (np2 > 0) { norm2_out[0] = v2; if (norm2_host) norm2_host[0] = v2; }
   }
   if (do_rr) rr_arrow_solve<RrSyncWave>(in, nfov, np2 > 0 ? v2 : fov[nfov], v3, S, rr_out, rr_host, (int)threadIdx.x);
   /* one workgroup: no ticket — the mirrored results are out (system scope) before the flag */
   if (threadIdx.x == 0 && fin.flag) {
      __threadfence_system();
      *(volatile unsigned long long *)fin.flag = fin.seq;
   }
}

extern "C" int hipk_tail_finish(hipk_ctx *ctx, const hipk_rr_in *in, const double *fov_dev, int nfov, const double *alpha_dev, double *hnext_out) {
   const int np2 = ctx->tail_np2, np3 = ctx->tail_np3;
   double *n2o = ctx->tail_norm2_out, *doto = ctx->tail_dot_out;
   hipk_tail_abandon(ctx);
   if (in && (in->k < 1 || in->k > 16 || in->L < 0 || in->L > 10 || nfov > 126 || !hnext_out)) return -1;
   if (np3 <= 0) {
      /* t'At was not deferred (another operator path, the CPU checker): finish |t|^2 if it is still waiting, then the
       * Rayleigh-Ritz step as a launch of its own */
      if (np2 > 0) { const int rc = hipk_finalize_partials_t(ctx, ctx->tailp, np2, 1, n2o); if (rc) return rc; }
      if (in) return hipk_rr_arrow(ctx, in, fov_dev, nfov, alpha_dev, hnext_out);
      return 0;
   }
   if (in && (alpha_dev != doto || (np2 > 0 && fov_dev + nfov != n2o))) return -1;
   const hipk_xr_dev xr = hipk_xr_take(ctx, doto, 1);
   if (xr.tab && np2 > 0) return -1;            /* |t|^2 of a row-partitioned run must be global before the operator launch */
   const hipk_fin_flag ff = hipk_next_flag(ctx, doto);
   const int nt = np3 <= 1024 ? HIPK_BLOCK : (np3 <= 2048 ? 512 : FIN_TAIL_MAXBLOCK);      /* = fin_block_for (hipk_core.hip) */
   hipk_rr_in none;
   if (!in) memset(&none, 0, sizeof(none));
   if (xr.tab)
      hipLaunchKernelGGL((tail_finish_kernel<true>), dim3(1), dim3(nt), 0, ctx->stream, ctx->tailp, np2, ctx->partials, np3, n2o, np2 > 0 ? hipk_mirror_of(ctx, n2o) : NULL,
            doto, hipk_mirror_of(ctx, doto), ff, xr, in ? 1 : 0, in ? *in : none, fov_dev, nfov, hnext_out, in ? hipk_mirror_of(ctx, hnext_out) : NULL);
   else
      hipLaunchKernelGGL((tail_finish_kernel<false>), dim3(1), dim3(nt), 0, ctx->stream, ctx->tailp, np2, ctx->partials, np3, n2o, np2 > 0 ? hipk_mirror_of(ctx, n2o) : NULL,
            doto, hipk_mirror_of(ctx, doto), ff, xr, in ? 1 : 0, in ? *in : none, fov_dev, nfov, hnext_out, in ? hipk_mirror_of(ctx, hnext_out) : NULL);
   HIPK_CHECK(hipGetLastError());
   return 0;
}

/* ============================ column utilities ================================ */
#define UTIL_MAXCOLS 64
struct ColScal { double a[UTIL_MAXCOLS]; };
struct ColPerm { int p[UTIL_MAXCOLS]; };

template <typename T>
__global__ void __launch_bounds__(HIPK_BLOCK)
scale_kernel(T *__restrict__ X, int64_t ldX, int nx, ColScal sc, int64_t m) {
   const int64_t stride = (int64_t)gridDim.x * HIPK_BLOCK;
   for (int c = 0; c < nx; c++) {
      T *x = X + (size_t)c * ldX;
      const double a = sc.a[c];
      for (int64_t i = (int64_t)blockIdx.x * HIPK_BLOCK + threadIdx.x; i < m; i += stride)
         x[i] = (T)(a * (double)x[i]);
   }
}

template <typename T>
__global__ void __launch_bounds__(HIPK_BLOCK)
scale_rsqrt_kernel(T *__restrict__ X, int64_t ldX, int nx, const double *__restrict__ norm2, int64_t m) {
   const int64_t stride = (int64_t)gridDim.x * HIPK_BLOCK;
   for (int c = 0; c < nx; c++) {
      T *x = X + (size_t)c * ldX;
      const double a = 1.0 / sqrt(norm2[c]);     /* same two IEEE operations as the host path */
      for (int64_t i = (int64_t)blockIdx.x * HIPK_BLOCK + threadIdx.x; i < m; i += stride)
         x[i] = (T)(a * (double)x[i]);
   }
}

/* Y = i * X for columns holding (re, im) pairs: (re, im) -> (-im, re).  One pair per lane
 * visit, 16- or 8-byte accesses. */
template <typename T>
__global__ void __launch_bounds__(HIPK_BLOCK)
pair_rotate_kernel(const T *__restrict__ X, int64_t ldX, T *__restrict__ Y, int64_t ldY, int nx, int64_t npairs) {
   const int64_t stride = (int64_t)gridDim.x * HIPK_BLOCK;
   for (int c = 0; c < nx; c++) {
      const T *x = X + (size_t)c * ldX;
      T *y = Y + (size_t)c * ldY;
      for (int64_t i = (int64_t)blockIdx.x * HIPK_BLOCK + threadIdx.x; i < npairs; i += stride) {
         const T re = x[2 * i], im = x[2 * i + 1];
         y[2 * i] = -im;
         y[2 * i + 1] = re;
      }
   }
}

template <typename T>
__global__ void __launch_bounds__(HIPK_BLOCK)
axpy_kernel(ColScal sc, const T *__restrict__ X, int64_t ldX, T *__restrict__ Y, int64_t ldY,
      int nx, int64_t m) {
   const int64_t stride = (int64_t)gridDim.x * HIPK_BLOCK;
   for (int c = 0; c < nx; c++) {
      const T *x = X + (size_t)c * ldX;
      T *y = Y + (size_t)c * ldY;
      const double a = sc.a[c];
      for (int64_t i = (int64_t)blockIdx.x * HIPK_BLOCK + threadIdx.x; i < m; i += stride)
         y[i] = (T)fma(a, (double)x[i], (double)y[i]);
   }
}

template <typename T>
__global__ void __launch_bounds__(HIPK_BLOCK)
xpay_kernel(ColScal sc, const T *__restrict__ X, int64_t ldX, T *__restrict__ Y, int64_t ldY,
      int nx, int64_t m) {
   const int64_t stride = (int64_t)gridDim.x * HIPK_BLOCK;
   for (int c = 0; c < nx; c++) {
      const T *x = X + (size_t)c * ldX;
      T *y = Y + (size_t)c * ldY;
      const double a = sc.a[c];
      for (int64_t i = (int64_t)blockIdx.x * HIPK_BLOCK + threadIdx.x; i < m; i += stride)
         y[i] = (T)fma(a, (double)y[i], (double)x[i]);
   }
}

template <typename T>
__global__ void __launch_bounds__(HIPK_BLOCK)
gather_kernel(const T *__restrict__ X, int64_t ldX, ColPerm pm, int n, T *__restrict__ Y,
      int64_t ldY, int64_t m) {
   const int64_t stride = (int64_t)gridDim.x * HIPK_BLOCK;
   for (int c = 0; c < n; c++) {
      const T *x = X + (size_t)pm.p[c] * ldX;
      T *y = Y + (size_t)c * ldY;
      for (int64_t i = (int64_t)blockIdx.x * HIPK_BLOCK + threadIdx.x; i < m; i += stride)
         y[i] = x[i];
   }
}

template <typename T>
__global__ void __launch_bounds__(HIPK_BLOCK)
norms2_kernel(const T *__restrict__ X, int64_t ldX, int nx, int64_t m,
      double *__restrict__ partials) {
   __shared__ double sm[HIPK_BLOCK / HIPK_WAVE];
   const int64_t stride = (int64_t)gridDim.x * HIPK_BLOCK;
   for (int c = 0; c < nx; c++) {
      const T *x = X + (size_t)c * ldX;
      double s = 0.0;
      for (int64_t i = (int64_t)blockIdx.x * HIPK_BLOCK + threadIdx.x; i < m; i += stride) {
         double v = (double)x[i];
         s = fma(v, v, s);
      }
      s = hipk_wave_sum(s);
      if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
      __syncthreads();
      if (threadIdx.x == 0) partials[(size_t)blockIdx.x * nx + c] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
      __syncthreads();
   }
}

template <typename T>
__global__ void __launch_bounds__(HIPK_BLOCK)
residual_kernel(const T *__restrict__ X, int64_t ldX, T *__restrict__ Wr, int64_t ldW, int nx,
      ColScal th, int64_t m, double *__restrict__ partials) {
   __shared__ double sm[HIPK_BLOCK / HIPK_WAVE];
   const int64_t stride = (int64_t)gridDim.x * HIPK_BLOCK;
   for (int c = 0; c < nx; c++) {
      const T *x = X + (size_t)c * ldX;
      T *w = Wr + (size_t)c * ldW;
      const double t = th.a[c];
      double s = 0.0;
      for (int64_t i = (int64_t)blockIdx.x * HIPK_BLOCK + threadIdx.x; i < m; i += stride) {
         T r = (T)fma(-t, (double)x[i], (double)w[i]);
         w[i] = r;
         s = fma((double)r, (double)r, s);
      }
      s = hipk_wave_sum(s);
      if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
      __syncthreads();
      if (threadIdx.x == 0) partials[(size_t)blockIdx.x * nx + c] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
      __syncthreads();
   }
}

/* out[c] = X(:,c)' Y(:,c): b independent dot products (block QMR recurrences) */
template <typename T>
__global__ void __launch_bounds__(HIPK_BLOCK)
pair_dots_kernel(const T *__restrict__ X, int64_t ldX, const T *__restrict__ Y, int64_t ldY, int nx,
      int64_t m, double *__restrict__ partials) {
   __shared__ double sm[HIPK_BLOCK / HIPK_WAVE];
   const int64_t stride = (int64_t)gridDim.x * HIPK_BLOCK;
   for (int c = 0; c < nx; c++) {
      const T *x = X + (size_t)c * ldX;
      const T *y = Y + (size_t)c * ldY;
      double s = 0.0;
      for (int64_t i = (int64_t)blockIdx.x * HIPK_BLOCK + threadIdx.x; i < m; i += stride)
         s = fma((double)x[i], (double)y[i], s);
      s = hipk_wave_sum(s);
      if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
      __syncthreads();
      if (threadIdx.x == 0) partials[(size_t)blockIdx.x * nx + c] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
      __syncthreads();
   }
}

/* y += a x (stored); out[c] = z'y or y'y: the axpy and the dot that follows it in one pass */
template <typename T>
__global__ void __launch_bounds__(HIPK_BLOCK)
axpy_dot_kernel(ColScal sc, const T *__restrict__ X, int64_t ldX, T *__restrict__ Y, int64_t ldY,
      const T *__restrict__ Z, int64_t ldZ, int nx, int64_t m, double *__restrict__ partials) {
   __shared__ double sm[HIPK_BLOCK / HIPK_WAVE];
   const int64_t stride = (int64_t)gridDim.x * HIPK_BLOCK;
   for (int c = 0; c < nx; c++) {
      const T *x = X + (size_t)c * ldX;
      T *y = Y + (size_t)c * ldY;
      const T *z = Z ? Z + (size_t)c * ldZ : NULL;
      const double a = sc.a[c];
      double s = 0.0;
      for (int64_t i = (int64_t)blockIdx.x * HIPK_BLOCK + threadIdx.x; i < m; i += stride) {
         const T ny = (T)fma(a, (double)x[i], (double)y[i]);
         y[i] = ny;
         s = fma(z ? (double)z[i] : (double)ny, (double)ny, s);
      }
      s = hipk_wave_sum(s);
      if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
      __syncthreads();
      if (threadIdx.x == 0) partials[(size_t)blockIdx.x * nx + c] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
      __syncthreads();
   }
}

/* delta = gamma*delta + eta*d; sol += delta; out[c] = |sol(:,c)|^2  (one pass, block QMR) */
template <typename T>
__global__ void __launch_bounds__(HIPK_BLOCK)
qmr_update_kernel(ColScal gam, ColScal eta, const T *__restrict__ D, int64_t ldD, T *__restrict__ Delta,
      int64_t ldDelta, T *__restrict__ Sol, int64_t ldSol, int nx, int64_t m, double *__restrict__ partials) {
   __shared__ double sm[HIPK_BLOCK / HIPK_WAVE];
   const int64_t stride = (int64_t)gridDim.x * HIPK_BLOCK;
   for (int c = 0; c < nx; c++) {
      const T *d = D + (size_t)c * ldD;
      T *de = Delta + (size_t)c * ldDelta;
      T *so = Sol + (size_t)c * ldSol;
      const double g = gam.a[c], e = eta.a[c];
      double s = 0.0;
      for (int64_t i = (int64_t)blockIdx.x * HIPK_BLOCK + threadIdx.x; i < m; i += stride) {
         T nd = (T)fma((double)de[i], g, (double)d[i] * e);
         de[i] = nd;
         T ns = (T)((double)nd + (double)so[i]);
         so[i] = ns;
         s = fma((double)ns, (double)ns, s);
      }
      s = hipk_wave_sum(s);
      if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
      __syncthreads();
      if (threadIdx.x == 0) partials[(size_t)blockIdx.x * nx + c] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
      __syncthreads();
   }
}

/* The QMR step and the next application of the Jacobi preconditioner in one pass (block QMR):
 *    delta = gamma delta + eta d;  sol += delta;  out[c] = |sol(:,c)|^2
 *    w = g ./ (diag - shift[c]);                  out[nx + c] = g(:,c)' w(:,c)
 * Three launches of the unfused sequence (qmr_update, jacobi, pair_dots) read g twice and w once more
 * than this does (reference inner_solve.c:384-397 fuses the first line on the CPU; :619-634 is the second). */
template <typename T>
__global__ void __launch_bounds__(HIPK_BLOCK)
qmr_update_jacobi_kernel(ColScal gam, ColScal eta, ColScal shf, double min_den, const T *__restrict__ D, int64_t ldD,
      T *__restrict__ Delta, int64_t ldDelta, T *__restrict__ Sol, int64_t ldSol, const T *__restrict__ G, int64_t ldG,
      const T *__restrict__ diag, T *__restrict__ Wp, int64_t ldW, int nx, int c0, int64_t m, double *__restrict__ partials) {
   /* rows outside, (up to 8) columns inside: the diagonal is read once per row, not once per column */
   constexpr int NXC = 8;
   __shared__ double sm[HIPK_BLOCK / HIPK_WAVE][2 * NXC];
   const int nc = min(NXC, nx - c0);
   double s1[NXC], s2[NXC];
#pragma unroll
   for (int c = 0; c < NXC; c++) { s1[c] = 0.0; s2[c] = 0.0; }
   const int64_t stride = (int64_t)gridDim.x * HIPK_BLOCK;
   for (int64_t i = (int64_t)blockIdx.x * HIPK_BLOCK + threadIdx.x; i < m; i += stride) {
      const double dg = (double)diag[i];
#pragma unroll
      for (int c = 0; c < NXC; c++)
         if (c < nc) {
            const size_t cc = (size_t)(c0 + c);
            const T nd = (T)fma((double)Delta[i + cc * ldDelta], gam.a[c0 + c], (double)D[i + cc * ldD] * eta.a[c0 + c]);
            Delta[i + cc * ldDelta] = nd;
            const T ns = (T)((double)nd + (double)Sol[i + cc * ldSol]);
            Sol[i + cc * ldSol] = ns;
            s1[c] = fma((double)ns, (double)ns, s1[c]);
            double den = dg - shf.a[c0 + c];
            if (!(fabs(den) > min_den)) den = copysign(min_den, den);
            const double gi = (double)G[i + cc * ldG];
            const T wi = (T)(gi / den);
            Wp[i + cc * ldW] = wi;
            s2[c] = fma(gi, (double)wi, s2[c]);
         }
   }
   const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
   for (int c = 0; c < NXC; c++) {
      const double a = hipk_wave_sum(s1[c]), b = hipk_wave_sum(s2[c]);
      if (lane == 0) { sm[wv][c] = a; sm[wv][NXC + c] = b; }
   }
   __syncthreads();
   if (threadIdx.x < 2 * NXC) {
      const int which = threadIdx.x / NXC, c = threadIdx.x % NXC;
      if (c < nc)
         partials[(size_t)blockIdx.x * 2 * nx + which * nx + c0 + c] =
               (sm[0][threadIdx.x] + sm[1][threadIdx.x]) + (sm[2][threadIdx.x] + sm[3][threadIdx.x]);
   }
}

/* out[c] = x_c' w_c, out[nx + c] = v_c' w_c, out[2 nx + c] = v_c' x_c in one pass over the three panels: what
 * the block QMR step needs to form sigma = v'(I - x x')w = v'w - (x'w)(v'x) without first storing the projected w */
template <typename T>
__global__ void __launch_bounds__(HIPK_BLOCK)
triple_dots_kernel(const T *__restrict__ X, int64_t ldX, const T *__restrict__ Vv, int64_t ldV, const T *__restrict__ Wv,
      int64_t ldW, int nx, int64_t m, double *__restrict__ partials) {
   __shared__ double sm[HIPK_BLOCK / HIPK_WAVE][3];
   const int64_t stride = (int64_t)gridDim.x * HIPK_BLOCK;
   for (int c = 0; c < nx; c++) {
      const T *x = X + (size_t)c * ldX, *v = Vv + (size_t)c * ldV, *w = Wv + (size_t)c * ldW;
      double a = 0.0, b = 0.0, d = 0.0;
      for (int64_t i = (int64_t)blockIdx.x * HIPK_BLOCK + threadIdx.x; i < m; i += stride) {
         const double xi = (double)x[i], vi = (double)v[i], wi = (double)w[i];
         a = fma(xi, wi, a); b = fma(vi, wi, b); d = fma(vi, xi, d);
      }
      a = hipk_wave_sum(a); b = hipk_wave_sum(b); d = hipk_wave_sum(d);
      if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6][0] = a; sm[threadIdx.x >> 6][1] = b; sm[threadIdx.x >> 6][2] = d; }
      __syncthreads();
      if (threadIdx.x < 3)
         partials[(size_t)blockIdx.x * 3 * nx + threadIdx.x * nx + c] =
               (sm[0][threadIdx.x] + sm[1][threadIdx.x]) + (sm[2][threadIdx.x] + sm[3][threadIdx.x]);
      __syncthreads();
   }
}

/* g_c -= alpha_c (w_c - xr_c x_c), out[c] = g_c' g_c: the projection of w against x and the residual update of
 * the QMR step in one pass; the projected w itself is never stored (inner_solve.c:853-880, :371-377) */
template <typename T>
__global__ void __launch_bounds__(HIPK_BLOCK)
axpy_proj_dot_kernel(ColScal alpha, ColScal xr, const T *__restrict__ Wv, int64_t ldW, const T *__restrict__ X, int64_t ldX,
      T *__restrict__ G, int64_t ldG, int nx, int64_t m, double *__restrict__ partials) {
   __shared__ double sm[HIPK_BLOCK / HIPK_WAVE];
   const int64_t stride = (int64_t)gridDim.x * HIPK_BLOCK;
   for (int c = 0; c < nx; c++) {
      const T *w = Wv + (size_t)c * ldW, *x = X + (size_t)c * ldX;
      T *g = G + (size_t)c * ldG;
      const double a = alpha.a[c], r = xr.a[c];
      double s = 0.0;
      for (int64_t i = (int64_t)blockIdx.x * HIPK_BLOCK + threadIdx.x; i < m; i += stride) {
         const T wp = (T)fma(-r, (double)x[i], (double)w[i]);          /* rounded like the stored projected w */
         const T ng = (T)fma(-a, (double)wp, (double)g[i]);
         g[i] = ng;
         s = fma((double)ng, (double)ng, s);
      }
      s = hipk_wave_sum(s);
      if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
      __syncthreads();
      if (threadIdx.x == 0) partials[(size_t)blockIdx.x * nx + c] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
      __syncthreads();
   }
}

/* The same with the NEXT inner product of the preconditioned QMR already taken: out[nx + c] = g_c' K^-1 g_c for the
 * updated g and the Jacobi preconditioner K = diag - shift[c].  With rho known at this synchronisation the step's
 * beta = rho / rho_prev is known before the QMR update runs, and that pass can write the new direction
 * d = K^-1 g + beta d in place (qmr_update_dir_kernel) instead of storing w = K^-1 g and adding beta d in a further pass.
 * Rows outside, (up to 8) columns inside: the diagonal is read once per row. */
/* ---- the scalar recurrences of one block-QMR step, evaluated ON THE DEVICE (eigs_jd.c: the step with one host synchronisation).
 * The launches that apply a step's coefficients compute them in their prologue, every lane for itself, from the reduction
 * results of the launches before them — still in HBM — and from the previous step's state, passed by value.  The host evaluates
 * the same expressions on the mirrored results after its one wait; both sides round every operation separately (no
 * contraction here, ISO C on the host), division and square root are correctly rounded on both: the same bits.
 *   tri = [x'w | v'w | v'x] (hipk_triple_dots), ggr = [g'g | g'K^-1 g] (hipk_axpy_proj_dot_jacobi_dev) */
struct QmrPrev { double rho_prev[8], tau_prev[8], theta_prev[8]; double eps; };
__device__ __forceinline__ void qmr_alpha_dev(const double *__restrict__ tri, int nx, int col, double rho_prev, double eps, double &alpha, double &xr) {
#pragma clang fp contract(off)
   xr = tri[col];
   const double t = xr * tri[2 * nx + col];
   const double sigma = tri[nx + col] - t;
   bool bad = !isfinite(sigma) || sigma == 0.0;
   double a = 0.0;
   if (!bad) {
      a = rho_prev / sigma;
      bad = !isfinite(a) || fabs(a) < eps || fabs(a) > 1.0 / eps;
   }
   alpha = bad ? 0.0 : a;                        /* 0: the column leaves the block at this step (the host sees the same) */
}
__device__ __forceinline__ void qmr_coeffs_dev(const double *__restrict__ ggr, int nx, int col, double alpha, double rho_prev, double tau_prev,
      double theta_prev, double &gam, double &eta, double &bet) {
#pragma clang fp contract(off)
   const double theta = sqrt(ggr[col]) / tau_prev;
   const double t2 = theta * theta;
   const double c = 1.0 / sqrt(1 + t2);
   const double cc = c * c;
   const double g1 = cc * theta_prev;
   gam = g1 * theta_prev;
   const double e1 = alpha * c;
   eta = e1 * c;
   bet = ggr[nx + col] / rho_prev;
}

template <typename T, bool DEV>
__global__ void __launch_bounds__(HIPK_BLOCK)
axpy_proj_dot_jacobi_kernel(ColScal alpha, ColScal xr, ColScal shf, double min_den, const T *__restrict__ Wv, int64_t ldW,
      const T *__restrict__ X, int64_t ldX, T *__restrict__ G, int64_t ldG, const T *__restrict__ diag, int nx, int c0, int64_t m,
      double *__restrict__ partials, const double *__restrict__ tri, QmrPrev pv) {
   constexpr int NXC = 8;
   __shared__ double sm[HIPK_BLOCK / HIPK_WAVE][2 * NXC];
   const int nc = min(NXC, nx - c0);
   double s1[NXC], s2[NXC], al[NXC], xq[NXC];
#pragma unroll
   for (int c = 0; c < NXC; c++) {
      s1[c] = 0.0; s2[c] = 0.0; al[c] = 0.0; xq[c] = 0.0;
      if (c < nc) {
         if (DEV) qmr_alpha_dev(tri, nx, c0 + c, pv.rho_prev[c], pv.eps, al[c], xq[c]);
         else { al[c] = alpha.a[c0 + c]; xq[c] = xr.a[c0 + c]; }
      }
   }
   const int64_t stride = (int64_t)gridDim.x * HIPK_BLOCK;
   for (int64_t i = (int64_t)blockIdx.x * HIPK_BLOCK + threadIdx.x; i < m; i += stride) {
      const double dg = (double)diag[i];
#pragma unroll
      for (int c = 0; c < NXC; c++)
         if (c < nc) {
            const size_t cc = (size_t)(c0 + c);
            const T wp = (T)fma(-xq[c], (double)X[i + cc * ldX], (double)Wv[i + cc * ldW]);   /* rounded like the stored projected w */
            const T ng = (T)fma(-al[c], (double)wp, (double)G[i + cc * ldG]);
            G[i + cc * ldG] = ng;
            s1[c] = fma((double)ng, (double)ng, s1[c]);
            double den = dg - shf.a[c0 + c];
            if (!(fabs(den) > min_den)) den = copysign(min_den, den);
            const T wi = (T)((double)ng / den);                      /* rounded like the stored K^-1 g */
            s2[c] = fma((double)ng, (double)wi, s2[c]);
         }
   }
   const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
   for (int c = 0; c < NXC; c++) {
      const double a = hipk_wave_sum(s1[c]), b = hipk_wave_sum(s2[c]);
      if (lane == 0) { sm[wv][c] = a; sm[wv][NXC + c] = b; }
   }
   __syncthreads();
   if (threadIdx.x < 2 * NXC) {
      const int which = threadIdx.x / NXC, c = threadIdx.x % NXC;
      if (c < nc)
         partials[(size_t)blockIdx.x * 2 * nx + which * nx + c0 + c] =
               (sm[0][threadIdx.x] + sm[1][threadIdx.x]) + (sm[2][threadIdx.x] + sm[3][threadIdx.x]);
   }
}

/* delta = gamma delta + eta d;  sol += delta;  out[c] = |sol(:,c)|^2;  d = g ./ (diag - shift[c]) + beta d (in place):
 * the QMR step and the next search direction in one pass over d, delta, sol, g (seven array passes per column; the
 * sequence qmr_update_jacobi + axpy it replaces makes eleven) */
template <typename T, bool DEV>
__global__ void __launch_bounds__(HIPK_BLOCK)
qmr_update_dir_kernel(ColScal gam, ColScal eta, ColScal bet, ColScal shf, double min_den, T *__restrict__ D, int64_t ldD,
      T *__restrict__ Delta, int64_t ldDelta, T *__restrict__ Sol, int64_t ldSol, const T *__restrict__ G, int64_t ldG,
      const T *__restrict__ diag, int nx, int c0, int64_t m, double *__restrict__ partials, const double *__restrict__ tri,
      const double *__restrict__ ggr, QmrPrev pv) {
   constexpr int NXC = 8;
   __shared__ double sm[HIPK_BLOCK / HIPK_WAVE][NXC];
   const int nc = min(NXC, nx - c0);
   double s1[NXC], ga[NXC], et[NXC], be[NXC];
   bool live[NXC];                  /* DEV: a column whose alpha was unusable leaves the block before this update (its sol stays) */
#pragma unroll
   for (int c = 0; c < NXC; c++) {
      s1[c] = 0.0; ga[c] = 0.0; et[c] = 0.0; be[c] = 0.0; live[c] = c < nc;
      if (c < nc) {
         if (DEV) {
            double a, x_;
            qmr_alpha_dev(tri, nx, c0 + c, pv.rho_prev[c], pv.eps, a, x_);
            live[c] = a != 0.0;
            if (live[c]) qmr_coeffs_dev(ggr, nx, c0 + c, a, pv.rho_prev[c], pv.tau_prev[c], pv.theta_prev[c], ga[c], et[c], be[c]);
         } else { ga[c] = gam.a[c0 + c]; et[c] = eta.a[c0 + c]; be[c] = bet.a[c0 + c]; }
      }
   }
   const int64_t stride = (int64_t)gridDim.x * HIPK_BLOCK;
   for (int64_t i = (int64_t)blockIdx.x * HIPK_BLOCK + threadIdx.x; i < m; i += stride) {
      const double dg = (double)diag[i];
#pragma unroll
      for (int c = 0; c < NXC; c++)
         if (live[c]) {
            const size_t cc = (size_t)(c0 + c);
            const double di = (double)D[i + cc * ldD];
            const T nd = (T)fma((double)Delta[i + cc * ldDelta], ga[c], di * et[c]);
            Delta[i + cc * ldDelta] = nd;
            const T ns = (T)((double)nd + (double)Sol[i + cc * ldSol]);
            Sol[i + cc * ldSol] = ns;
            s1[c] = fma((double)ns, (double)ns, s1[c]);
            double den = dg - shf.a[c0 + c];
            if (!(fabs(den) > min_den)) den = copysign(min_den, den);
            const T wi = (T)((double)G[i + cc * ldG] / den);
            D[i + cc * ldD] = (T)fma(be[c], di, (double)wi);    /* w += beta d, as the axpy pass rounds it */
         }
   }
   const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
   for (int c = 0; c < NXC; c++) {
      const double a = hipk_wave_sum(s1[c]);
      if (lane == 0) sm[wv][c] = a;
   }
   __syncthreads();
   if (threadIdx.x < NXC && (int)threadIdx.x < nc)
      partials[(size_t)blockIdx.x * nx + c0 + threadIdx.x] = (sm[0][threadIdx.x] + sm[1][threadIdx.x]) + (sm[2][threadIdx.x] + sm[3][threadIdx.x]);
}

#define DISPATCH_RT(dt, CALL_D, CALL_F)         \
   switch (dt) {                                \
   case HIPK_F64: { typedef double T; CALL_D; } break; \
   case HIPK_F32: { typedef float T; CALL_F; } break;  \
   default: return -44;                         \
   }

extern "C" int hipk_scale_cols(hipk_ctx *ctx, hipk_dtype dt, int64_t m, void *X, int64_t ldX,
      int nx, const double *alpha_host) {
   /* real factors on complex columns: the real kernel on the panel seen as 2m reals */
   if (HIPK_IS_Z(dt)) return hipk_scale_cols(ctx, hipk_real_of(dt), 2 * m, X, 2 * ldX, nx, alpha_host);
   hipk_prof_scope ps_(HIPK_PROF_VEC, ctx->stream, (double)m * (double)(HIPK_IS_Z(dt) ? 2 : 1) * (dt == HIPK_F64 || dt == HIPK_C64 ? 8.0 : 4.0) * (double)(2 * nx));
   for (int c0 = 0; c0 < nx; c0 += UTIL_MAXCOLS) {
      int n = nx - c0 < UTIL_MAXCOLS ? nx - c0 : UTIL_MAXCOLS;
      ColScal sc;
      for (int c = 0; c < n; c++) sc.a[c] = alpha_host[c0 + c];
      int gx = hipk_grid_for_rows(ctx, m, HIPK_BLOCK * 4, 8);
      DISPATCH_RT(dt,
            hipLaunchKernelGGL(scale_kernel<T>, dim3(gx), dim3(HIPK_BLOCK), 0, ctx->stream, (T *)X + (size_t)c0 * ldX, ldX, n, sc, m),
            hipLaunchKernelGGL(scale_kernel<T>, dim3(gx), dim3(HIPK_BLOCK), 0, ctx->stream, (T *)X + (size_t)c0 * ldX, ldX, n, sc, m));
      HIPK_CHECK(hipGetLastError());
   }
   return 0;
}

extern "C" int hipk_scale_cols_rsqrt_dev(hipk_ctx *ctx, hipk_dtype dt, int64_t m, void *X, int64_t ldX,
      int nx, const double *norm2_dev) {
   if (nx <= 0) return 0;
   if (HIPK_IS_Z(dt)) return hipk_scale_cols_rsqrt_dev(ctx, hipk_real_of(dt), 2 * m, X, 2 * ldX, nx, norm2_dev);
   hipk_prof_scope ps_(HIPK_PROF_VEC, ctx->stream, (double)m * (double)(HIPK_IS_Z(dt) ? 2 : 1) * (dt == HIPK_F64 || dt == HIPK_C64 ? 8.0 : 4.0) * (double)(2 * nx));
   int gx = hipk_grid_for_rows(ctx, m, HIPK_BLOCK * 4, 8);
   DISPATCH_RT(dt,
         hipLaunchKernelGGL(scale_rsqrt_kernel<T>, dim3(gx), dim3(HIPK_BLOCK), 0, ctx->stream, (T *)X, ldX, nx, norm2_dev, m),
         hipLaunchKernelGGL(scale_rsqrt_kernel<T>, dim3(gx), dim3(HIPK_BLOCK), 0, ctx->stream, (T *)X, ldX, nx, norm2_dev, m));
   HIPK_CHECK(hipGetLastError());
   return 0;
}

extern "C" int hipk_axpy_cols(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const double *alpha_host,
      const void *X, int64_t ldX, void *Y, int64_t ldY, int nx) {
   if (HIPK_IS_Z(dt)) return hipk_z_axpy(ctx, dt, m, alpha_host, X, ldX, Y, ldY, nx, 0);      /* (re, im) factors */
   hipk_prof_scope ps_(HIPK_PROF_VEC, ctx->stream, (double)m * (double)(HIPK_IS_Z(dt) ? 2 : 1) * (dt == HIPK_F64 || dt == HIPK_C64 ? 8.0 : 4.0) * (double)(3 * nx));
   for (int c0 = 0; c0 < nx; c0 += UTIL_MAXCOLS) {
      int n = nx - c0 < UTIL_MAXCOLS ? nx - c0 : UTIL_MAXCOLS;
      ColScal sc;
      for (int c = 0; c < n; c++) sc.a[c] = alpha_host[c0 + c];
      int gx = hipk_grid_for_rows(ctx, m, HIPK_BLOCK * 4, 8);
      DISPATCH_RT(dt,
            hipLaunchKernelGGL(axpy_kernel<T>, dim3(gx), dim3(HIPK_BLOCK), 0, ctx->stream, sc, (const T *)X + (size_t)c0 * ldX, ldX, (T *)Y + (size_t)c0 * ldY, ldY, n, m),
            hipLaunchKernelGGL(axpy_kernel<T>, dim3(gx), dim3(HIPK_BLOCK), 0, ctx->stream, sc, (const T *)X + (size_t)c0 * ldX, ldX, (T *)Y + (size_t)c0 * ldY, ldY, n, m));
      HIPK_CHECK(hipGetLastError());
   }
   return 0;
}

extern "C" int hipk_pair_rotate(hipk_ctx *ctx, hipk_dtype dt, int64_t npairs, const void *X, int64_t ldX,
      void *Y, int64_t ldY, int nx) {
   if (nx <= 0 || npairs <= 0) return 0;
   int gx = hipk_grid_for_rows(ctx, npairs, HIPK_BLOCK * 4, 8);
   DISPATCH_RT(dt,
         hipLaunchKernelGGL(pair_rotate_kernel<T>, dim3(gx), dim3(HIPK_BLOCK), 0, ctx->stream, (const T *)X, ldX, (T *)Y, ldY, nx, npairs),
         hipLaunchKernelGGL(pair_rotate_kernel<T>, dim3(gx), dim3(HIPK_BLOCK), 0, ctx->stream, (const T *)X, ldX, (T *)Y, ldY, nx, npairs));
   HIPK_CHECK(hipGetLastError());
   return 0;
}

/* column copy: 16 bytes per lane when the columns allow it (the runtime's 2-D copy reaches 2.4 TB/s) */
template <typename U>
__global__ void __launch_bounds__(HIPK_BLOCK)
copy_cols_kernel(const char *__restrict__ X, size_t ldx_bytes, char *__restrict__ Y, size_t ldy_bytes, size_t n) {
   const U *x = (const U *)(X + (size_t)blockIdx.y * ldx_bytes);
   U *y = (U *)(Y + (size_t)blockIdx.y * ldy_bytes);
   const size_t stride = (size_t)gridDim.x * HIPK_BLOCK;
   size_t i = (size_t)blockIdx.x * HIPK_BLOCK + threadIdx.x;
   for (; i + 3 * stride < n; i += 4 * stride) {
      const U a = x[i], b = x[i + stride], c = x[i + 2 * stride], d = x[i + 3 * stride];
      y[i] = a; y[i + stride] = b; y[i + 2 * stride] = c; y[i + 3 * stride] = d;
   }
   for (; i < n; i += stride) y[i] = x[i];
}

extern "C" int hipk_copy_cols(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const void *X,
      int64_t ldX, void *Y, int64_t ldY, int nx) {
   size_t es = (dt == HIPK_F64) ? 8 : (dt == HIPK_F32) ? 4 : (dt == HIPK_C64) ? 16 : 8;
   if (nx <= 0 || m <= 0) return 0;
   hipk_prof_scope ps_(HIPK_PROF_VEC, ctx->stream, (double)m * (double)(HIPK_IS_Z(dt) ? 2 : 1) * (dt == HIPK_F64 || dt == HIPK_C64 ? 8.0 : 4.0) * (double)(2 * nx));
   const size_t bytes = (size_t)m * es, lx = (size_t)ldX * es, ly = (size_t)ldY * es;
   if (nx > 65535 || bytes < 4096) {
      HIPK_CHECK(hipMemcpy2DAsync(Y, ly, X, lx, bytes, (size_t)nx, hipMemcpyDeviceToDevice, ctx->stream));
      return 0;
   }
   const bool v16 = ((uintptr_t)X % 16 == 0) && ((uintptr_t)Y % 16 == 0) && (nx == 1 || (lx % 16 == 0 && ly % 16 == 0));
   const size_t us = v16 ? 16 : (es == 4 ? 4 : 8);             /* bytes per lane visit */
   const size_t n = bytes / us, head = n * us;
   int gx = hipk_grid_for_rows(ctx, (int64_t)n, HIPK_BLOCK * 4, 8);
   if (nx > 1) { gx = (gx + nx - 1) / nx; if (gx < 1) gx = 1; }
   dim3 grid(gx, nx);
   if (v16) hipLaunchKernelGGL(copy_cols_kernel<uint4>, grid, dim3(HIPK_BLOCK), 0, ctx->stream, (const char *)X, lx, (char *)Y, ly, n);
   else if (es == 4) hipLaunchKernelGGL(copy_cols_kernel<unsigned int>, grid, dim3(HIPK_BLOCK), 0, ctx->stream, (const char *)X, lx, (char *)Y, ly, n);
   else hipLaunchKernelGGL(copy_cols_kernel<unsigned long long>, grid, dim3(HIPK_BLOCK), 0, ctx->stream, (const char *)X, lx, (char *)Y, ly, n);
   HIPK_CHECK(hipGetLastError());
   if (head < bytes)    /* fewer than 16 bytes per column left over */
      HIPK_CHECK(hipMemcpy2DAsync((char *)Y + head, ly, (const char *)X + head, lx, bytes - head, (size_t)nx, hipMemcpyDeviceToDevice, ctx->stream));
   return 0;
}

extern "C" int hipk_gather_cols(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const void *X,
      int64_t ldX, const int *perm_host, int n, void *Y, int64_t ldY) {
   if (HIPK_IS_Z(dt)) return hipk_gather_cols(ctx, hipk_real_of(dt), 2 * m, X, 2 * ldX, perm_host, n, Y, 2 * ldY);
   hipk_prof_scope ps_(HIPK_PROF_VEC, ctx->stream, (double)m * (double)(HIPK_IS_Z(dt) ? 2 : 1) * (dt == HIPK_F64 || dt == HIPK_C64 ? 8.0 : 4.0) * (double)(2 * n));
   for (int c0 = 0; c0 < n; c0 += UTIL_MAXCOLS) {
      int nn = n - c0 < UTIL_MAXCOLS ? n - c0 : UTIL_MAXCOLS;
      ColPerm pm;
      for (int c = 0; c < nn; c++) pm.p[c] = perm_host[c0 + c];
      int gx = hipk_grid_for_rows(ctx, m, HIPK_BLOCK * 4, 8);
      DISPATCH_RT(dt,
            hipLaunchKernelGGL(gather_kernel<T>, dim3(gx), dim3(HIPK_BLOCK), 0, ctx->stream, (const T *)X, ldX, pm, nn, (T *)Y + (size_t)c0 * ldY, ldY, m),
            hipLaunchKernelGGL(gather_kernel<T>, dim3(gx), dim3(HIPK_BLOCK), 0, ctx->stream, (const T *)X, ldX, pm, nn, (T *)Y + (size_t)c0 * ldY, ldY, m));
      HIPK_CHECK(hipGetLastError());
   }
   return 0;
}

extern "C" int hipk_col_norms2(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const void *X,
      int64_t ldX, int nx, double *out_dev) {
   if (nx <= 0) return 0;
   if (HIPK_IS_Z(dt)) return hipk_col_norms2(ctx, hipk_real_of(dt), 2 * m, X, 2 * ldX, nx, out_dev);   /* |z|^2 = re^2 + im^2 */
   hipk_prof_scope ps_(HIPK_PROF_VEC, ctx->stream, (double)m * (double)(HIPK_IS_Z(dt) ? 2 : 1) * (dt == HIPK_F64 || dt == HIPK_C64 ? 8.0 : 4.0) * (double)(nx));
   int gx = hipk_grid_for_rows(ctx, m, HIPK_BLOCK * 4, 4);
   if (hipk_reserve_partials(ctx, (size_t)gx * nx)) return -2;
   DISPATCH_RT(dt,
         hipLaunchKernelGGL(norms2_kernel<T>, dim3(gx), dim3(HIPK_BLOCK), 0, ctx->stream, (const T *)X, ldX, nx, m, ctx->partials),
         hipLaunchKernelGGL(norms2_kernel<T>, dim3(gx), dim3(HIPK_BLOCK), 0, ctx->stream, (const T *)X, ldX, nx, m, ctx->partials));
   HIPK_CHECK(hipGetLastError());
   return hipk_finalize_partials(ctx, ctx->partials, gx, nx, out_dev);
}

extern "C" int hipk_residual_cols(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const void *X,
      int64_t ldX, void *Wr, int64_t ldW, int nx, const double *theta_host, double *nrm2_dev) {
   if (HIPK_IS_Z(dt)) return hipk_residual_cols(ctx, hipk_real_of(dt), 2 * m, X, 2 * ldX, Wr, 2 * ldW, nx, theta_host, nrm2_dev);   /* theta is real */
   hipk_prof_scope ps_(HIPK_PROF_VEC, ctx->stream, (double)m * (double)(HIPK_IS_Z(dt) ? 2 : 1) * (dt == HIPK_F64 || dt == HIPK_C64 ? 8.0 : 4.0) * (double)(3 * nx));
   const size_t es = (dt == HIPK_F64) ? 8 : 4;
   for (int c0 = 0; c0 < nx; c0 += UTIL_MAXCOLS) {
      const int n = nx - c0 < UTIL_MAXCOLS ? nx - c0 : UTIL_MAXCOLS;
      ColScal th;
      for (int c = 0; c < n; c++) th.a[c] = theta_host[c0 + c];
      const char *Xc = (const char *)X + (size_t)c0 * ldX * es;
      char *Wc = (char *)Wr + (size_t)c0 * ldW * es;
      int gx = hipk_grid_for_rows(ctx, m, HIPK_BLOCK * 4, 4);
      if (hipk_reserve_partials(ctx, (size_t)gx * n)) return -2;
      DISPATCH_RT(dt,
            hipLaunchKernelGGL(residual_kernel<T>, dim3(gx), dim3(HIPK_BLOCK), 0, ctx->stream, (const T *)Xc, ldX, (T *)Wc, ldW, n, th, m, ctx->partials),
            hipLaunchKernelGGL(residual_kernel<T>, dim3(gx), dim3(HIPK_BLOCK), 0, ctx->stream, (const T *)Xc, ldX, (T *)Wc, ldW, n, th, m, ctx->partials));
      HIPK_CHECK(hipGetLastError());
      int rc = hipk_finalize_partials(ctx, ctx->partials, gx, n, nrm2_dev + c0);
      if (rc) return rc;
   }
   return 0;
}

extern "C" int hipk_pair_dots(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const void *X, int64_t ldX,
      const void *Y, int64_t ldY, int nx, double *out_dev) {
   if (nx <= 0) return 0;
   if (HIPK_IS_Z(dt)) return hipk_z_pair_dots(ctx, dt, m, X, ldX, Y, ldY, nx, out_dev);
   hipk_prof_scope ps_(HIPK_PROF_VEC, ctx->stream, (double)m * (double)(HIPK_IS_Z(dt) ? 2 : 1) * (dt == HIPK_F64 || dt == HIPK_C64 ? 8.0 : 4.0) * (double)(2 * nx));
   int gx = hipk_grid_for_rows(ctx, m, HIPK_BLOCK * 4, 4);
   if (hipk_reserve_partials(ctx, (size_t)gx * nx)) return -2;
   DISPATCH_RT(dt,
         hipLaunchKernelGGL(pair_dots_kernel<T>, dim3(gx), dim3(HIPK_BLOCK), 0, ctx->stream, (const T *)X, ldX, (const T *)Y, ldY, nx, m, ctx->partials),
         hipLaunchKernelGGL(pair_dots_kernel<T>, dim3(gx), dim3(HIPK_BLOCK), 0, ctx->stream, (const T *)X, ldX, (const T *)Y, ldY, nx, m, ctx->partials));
   HIPK_CHECK(hipGetLastError());
   return hipk_finalize_partials(ctx, ctx->partials, gx, nx, out_dev);
}

extern "C" int hipk_xpay_cols(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const double *alpha_host,
      const void *X, int64_t ldX, void *Y, int64_t ldY, int nx) {
   if (HIPK_IS_Z(dt)) return hipk_z_axpy(ctx, dt, m, alpha_host, X, ldX, Y, ldY, nx, 1);
   hipk_prof_scope ps_(HIPK_PROF_VEC, ctx->stream, (double)m * (double)(HIPK_IS_Z(dt) ? 2 : 1) * (dt == HIPK_F64 || dt == HIPK_C64 ? 8.0 : 4.0) * (double)(3 * nx));
   for (int c0 = 0; c0 < nx; c0 += UTIL_MAXCOLS) {
      int n = nx - c0 < UTIL_MAXCOLS ? nx - c0 : UTIL_MAXCOLS;
      ColScal sc;
      for (int c = 0; c < n; c++) sc.a[c] = alpha_host[c0 + c];
      int gx = hipk_grid_for_rows(ctx, m, HIPK_BLOCK * 4, 8);
      DISPATCH_RT(dt,
            hipLaunchKernelGGL(xpay_kernel<T>, dim3(gx), dim3(HIPK_BLOCK), 0, ctx->stream, sc, (const T *)X + (size_t)c0 * ldX, ldX, (T *)Y + (size_t)c0 * ldY, ldY, n, m),
            hipLaunchKernelGGL(xpay_kernel<T>, dim3(gx), dim3(HIPK_BLOCK), 0, ctx->stream, sc, (const T *)X + (size_t)c0 * ldX, ldX, (T *)Y + (size_t)c0 * ldY, ldY, n, m));
      HIPK_CHECK(hipGetLastError());
   }
   return 0;
}

extern "C" int hipk_axpy_dot(hipk_ctx *ctx, hipk_dtype dt, int64_t m, int nx, const double *alpha_host,
      const void *X, int64_t ldX, void *Y, int64_t ldY, const void *Z, int64_t ldZ, double *out_dev) {
   if (nx <= 0) return 0;
   if (nx > UTIL_MAXCOLS) return -1;
   hipk_prof_scope ps_(HIPK_PROF_VEC, ctx->stream, (double)m * (double)(HIPK_IS_Z(dt) ? 2 : 1) * (dt == HIPK_F64 || dt == HIPK_C64 ? 8.0 : 4.0) * (double)(4 * nx));
   ColScal sc;
   for (int c = 0; c < nx; c++) sc.a[c] = alpha_host[c];
   int gx = hipk_grid_for_rows(ctx, m, HIPK_BLOCK * 4, 4);   /* same grid as hipk_pair_dots: same sums */
   if (hipk_reserve_partials(ctx, (size_t)gx * nx)) return -2;
   DISPATCH_RT(dt,
         hipLaunchKernelGGL(axpy_dot_kernel<T>, dim3(gx), dim3(HIPK_BLOCK), 0, ctx->stream, sc, (const T *)X, ldX, (T *)Y, ldY, (const T *)Z, ldZ, nx, m, ctx->partials),
         hipLaunchKernelGGL(axpy_dot_kernel<T>, dim3(gx), dim3(HIPK_BLOCK), 0, ctx->stream, sc, (const T *)X, ldX, (T *)Y, ldY, (const T *)Z, ldZ, nx, m, ctx->partials));
   HIPK_CHECK(hipGetLastError());
   return hipk_finalize_partials(ctx, ctx->partials, gx, nx, out_dev);
}

extern "C" int hipk_qmr_update(hipk_ctx *ctx, hipk_dtype dt, int64_t m, int nx, const double *gamma_host,
      const double *eta_host, const void *D, int64_t ldD, void *Delta, int64_t ldDelta, void *Sol,
      int64_t ldSol, double *dotsol_dev) {
   if (nx <= 0) return 0;
   if (nx > UTIL_MAXCOLS) return -1;
   hipk_prof_scope ps_(HIPK_PROF_VEC, ctx->stream, (double)m * (double)(HIPK_IS_Z(dt) ? 2 : 1) * (dt == HIPK_F64 || dt == HIPK_C64 ? 8.0 : 4.0) * (double)(5 * nx));
   ColScal g, e;
   for (int c = 0; c < nx; c++) { g.a[c] = gamma_host[c]; e.a[c] = eta_host[c]; }
   int gx = hipk_grid_for_rows(ctx, m, HIPK_BLOCK * 4, 4);
   if (hipk_reserve_partials(ctx, (size_t)gx * nx)) return -2;
   DISPATCH_RT(dt,
         hipLaunchKernelGGL(qmr_update_kernel<T>, dim3(gx), dim3(HIPK_BLOCK), 0, ctx->stream, g, e, (const T *)D, ldD, (T *)Delta, ldDelta, (T *)Sol, ldSol, nx, m, ctx->partials),
         hipLaunchKernelGGL(qmr_update_kernel<T>, dim3(gx), dim3(HIPK_BLOCK), 0, ctx->stream, g, e, (const T *)D, ldD, (T *)Delta, ldDelta, (T *)Sol, ldSol, nx, m, ctx->partials));
   HIPK_CHECK(hipGetLastError());
   return hipk_finalize_partials(ctx, ctx->partials, gx, nx, dotsol_dev);
}

extern "C" int hipk_qmr_update_jacobi(hipk_ctx *ctx, hipk_dtype dt, int64_t m, int nx, const double *gamma_host,
      const double *eta_host, const void *D, int64_t ldD, void *Delta, int64_t ldDelta, void *Sol, int64_t ldSol,
      const void *G, int64_t ldG, const void *diag, const double *shift_host, double min_den, void *W, int64_t ldW,
      double *out_dev) {
   if (nx <= 0) return 0;
   if (nx > UTIL_MAXCOLS) return -1;
   hipk_prof_scope ps_(HIPK_PROF_VEC, ctx->stream, (double)m * (double)(HIPK_IS_Z(dt) ? 2 : 1) * (dt == HIPK_F64 || dt == HIPK_C64 ? 8.0 : 4.0) * (double)(8 * nx + 1));
   if (!(min_den > 0.0)) min_den = 1e-300;
   ColScal g, e, sh;
   for (int c = 0; c < nx; c++) { g.a[c] = gamma_host[c]; e.a[c] = eta_host[c]; sh.a[c] = shift_host ? shift_host[c] : 0.0; }
   int gx = hipk_grid_for_rows(ctx, m, HIPK_BLOCK * 4, 4);
   if (hipk_reserve_partials(ctx, (size_t)gx * 2 * nx)) return -2;
   for (int c0 = 0; c0 < nx; c0 += 8) {
      DISPATCH_RT(dt,
            hipLaunchKernelGGL(qmr_update_jacobi_kernel<T>, dim3(gx), dim3(HIPK_BLOCK), 0, ctx->stream, g, e, sh, min_den, (const T *)D, ldD, (T *)Delta, ldDelta, (T *)Sol, ldSol, (const T *)G, ldG, (const T *)diag, (T *)W, ldW, nx, c0, m, ctx->partials),
            hipLaunchKernelGGL(qmr_update_jacobi_kernel<T>, dim3(gx), dim3(HIPK_BLOCK), 0, ctx->stream, g, e, sh, min_den, (const T *)D, ldD, (T *)Delta, ldDelta, (T *)Sol, ldSol, (const T *)G, ldG, (const T *)diag, (T *)W, ldW, nx, c0, m, ctx->partials));
      HIPK_CHECK(hipGetLastError());
   }
   return hipk_finalize_partials(ctx, ctx->partials, gx, 2 * nx, out_dev);
}

extern "C" int hipk_axpy_proj_dot_jacobi(hipk_ctx *ctx, hipk_dtype dt, int64_t m, int nx, const double *alpha_host, const double *xr_host,
      const void *W, int64_t ldW, const void *X, int64_t ldX, void *G, int64_t ldG, const void *diag, const double *shift_host,
      double min_den, double *out_dev) {
   if (nx <= 0) return 0;
   if (nx > UTIL_MAXCOLS) return -1;
   hipk_prof_scope ps_(HIPK_PROF_VEC, ctx->stream, (double)m * (double)(HIPK_IS_Z(dt) ? 2 : 1) * (dt == HIPK_F64 || dt == HIPK_C64 ? 8.0 : 4.0) * (double)(4 * nx + 1));
   if (!(min_den > 0.0)) min_den = 1e-300;
   ColScal a, r, sh;
   for (int c = 0; c < nx; c++) { a.a[c] = alpha_host[c]; r.a[c] = xr_host[c]; sh.a[c] = shift_host ? shift_host[c] : 0.0; }
   int gx = hipk_grid_for_rows(ctx, m, HIPK_BLOCK * 4, 4);
   if (hipk_reserve_partials(ctx, (size_t)gx * 2 * nx)) return -2;
   QmrPrev pv;
   memset(&pv, 0, sizeof(pv));
   for (int c0 = 0; c0 < nx; c0 += 8) {
      DISPATCH_RT(dt,
            hipLaunchKernelGGL((axpy_proj_dot_jacobi_kernel<T, false>), dim3(gx), dim3(HIPK_BLOCK), 0, ctx->stream, a, r, sh, min_den, (const T *)W, ldW, (const T *)X, ldX, (T *)G, ldG, (const T *)diag, nx, c0, m, ctx->partials, (const double *)NULL, pv),
            hipLaunchKernelGGL((axpy_proj_dot_jacobi_kernel<T, false>), dim3(gx), dim3(HIPK_BLOCK), 0, ctx->stream, a, r, sh, min_den, (const T *)W, ldW, (const T *)X, ldX, (T *)G, ldG, (const T *)diag, nx, c0, m, ctx->partials, (const double *)NULL, pv));
      HIPK_CHECK(hipGetLastError());
   }
   return hipk_finalize_partials(ctx, ctx->partials, gx, 2 * nx, out_dev);
}
/* the same with alpha_c = rho_prev_c / (v'w - (x'w)(v'x)) and xr_c = x'w taken from tri_dev = [x'w | v'w | v'x] in HBM (the
 * results of hipk_triple_dots, which the host has NOT seen yet); nx <= 8 */
extern "C" int hipk_axpy_proj_dot_jacobi_dev(hipk_ctx *ctx, hipk_dtype dt, int64_t m, int nx, const double *tri_dev, const double *rho_prev_host,
      double mach_eps, const void *W, int64_t ldW, const void *X, int64_t ldX, void *G, int64_t ldG, const void *diag, const double *shift_host,
      double min_den, double *out_dev) {
   if (nx <= 0) return 0;
   if (nx > 8 || !tri_dev) return -1;
   hipk_prof_scope ps_(HIPK_PROF_VEC, ctx->stream, (double)m * (dt == HIPK_F64 ? 8.0 : 4.0) * (double)(4 * nx + 1));
   if (!(min_den > 0.0)) min_den = 1e-300;
   ColScal z, sh;
   QmrPrev pv;
   memset(&pv, 0, sizeof(pv)); memset(&z, 0, sizeof(z));
   pv.eps = mach_eps;
   for (int c = 0; c < nx; c++) { pv.rho_prev[c] = rho_prev_host[c]; sh.a[c] = shift_host ? shift_host[c] : 0.0; }
   int gx = hipk_grid_for_rows(ctx, m, HIPK_BLOCK * 4, 4);
   if (hipk_reserve_partials(ctx, (size_t)gx * 2 * nx)) return -2;
   DISPATCH_RT(dt,
         hipLaunchKernelGGL((axpy_proj_dot_jacobi_kernel<T, true>), dim3(gx), dim3(HIPK_BLOCK), 0, ctx->stream, z, z, sh, min_den, (const T *)W, ldW, (const T *)X, ldX, (T *)G, ldG, (const T *)diag, nx, 0, m, ctx->partials, tri_dev, pv),
         hipLaunchKernelGGL((axpy_proj_dot_jacobi_kernel<T, true>), dim3(gx), dim3(HIPK_BLOCK), 0, ctx->stream, z, z, sh, min_den, (const T *)W, ldW, (const T *)X, ldX, (T *)G, ldG, (const T *)diag, nx, 0, m, ctx->partials, tri_dev, pv));
   HIPK_CHECK(hipGetLastError());
   return hipk_finalize_partials(ctx, ctx->partials, gx, 2 * nx, out_dev);
}

extern "C" int hipk_qmr_update_dir(hipk_ctx *ctx, hipk_dtype dt, int64_t m, int nx, const double *gamma_host, const double *eta_host,
      const double *beta_host, void *D, int64_t ldD, void *Delta, int64_t ldDelta, void *Sol, int64_t ldSol, const void *G, int64_t ldG,
      const void *diag, const double *shift_host, double min_den, double *dotsol_dev) {
   if (nx <= 0) return 0;
   if (nx > UTIL_MAXCOLS) return -1;
   hipk_prof_scope ps_(HIPK_PROF_VEC, ctx->stream, (double)m * (double)(HIPK_IS_Z(dt) ? 2 : 1) * (dt == HIPK_F64 || dt == HIPK_C64 ? 8.0 : 4.0) * (double)(7 * nx + 1));
   if (!(min_den > 0.0)) min_den = 1e-300;
   ColScal g, e, b, sh;
   for (int c = 0; c < nx; c++) { g.a[c] = gamma_host[c]; e.a[c] = eta_host[c]; b.a[c] = beta_host[c]; sh.a[c] = shift_host ? shift_host[c] : 0.0; }
   int gx = hipk_grid_for_rows(ctx, m, HIPK_BLOCK * 4, 4);
   if (hipk_reserve_partials(ctx, (size_t)gx * nx)) return -2;
   QmrPrev pv;
   memset(&pv, 0, sizeof(pv));
   for (int c0 = 0; c0 < nx; c0 += 8) {
      DISPATCH_RT(dt,
            hipLaunchKernelGGL((qmr_update_dir_kernel<T, false>), dim3(gx), dim3(HIPK_BLOCK), 0, ctx->stream, g, e, b, sh, min_den, (T *)D, ldD, (T *)Delta, ldDelta, (T *)Sol, ldSol, (const T *)G, ldG, (const T *)diag, nx, c0, m, ctx->partials, (const double *)NULL, (const double *)NULL, pv),
            hipLaunchKernelGGL((qmr_update_dir_kernel<T, false>), dim3(gx), dim3(HIPK_BLOCK), 0, ctx->stream, g, e, b, sh, min_den, (T *)D, ldD, (T *)Delta, ldDelta, (T *)Sol, ldSol, (const T *)G, ldG, (const T *)diag, nx, c0, m, ctx->partials, (const double *)NULL, (const double *)NULL, pv));
      HIPK_CHECK(hipGetLastError());
   }
   return hipk_finalize_partials(ctx, ctx->partials, gx, nx, dotsol_dev);
}
/* the same with gamma, eta, beta of the step formed in the launch from tri_dev (as above), ggr_dev = [g'g | g'K^-1 g] (the results of
 * hipk_axpy_proj_dot_jacobi_dev) and the previous step's rho, tau, Theta; a column whose alpha was unusable is left alone; nx <= 8.
 * dotsol_dev[c] = |sol(:,c)|^2 of the columns that were updated (0 for the others) */
extern "C" int hipk_qmr_update_dir_dev(hipk_ctx *ctx, hipk_dtype dt, int64_t m, int nx, const double *tri_dev, const double *ggr_dev,
      const double *rho_prev_host, const double *tau_prev_host, const double *theta_prev_host, double mach_eps, void *D, int64_t ldD,
      void *Delta, int64_t ldDelta, void *Sol, int64_t ldSol, const void *G, int64_t ldG, const void *diag, const double *shift_host,
      double min_den, double *dotsol_dev) {
   if (nx <= 0) return 0;
   if (nx > 8 || !tri_dev || !ggr_dev) return -1;
   hipk_prof_scope ps_(HIPK_PROF_VEC, ctx->stream, (double)m * (dt == HIPK_F64 ? 8.0 : 4.0) * (double)(7 * nx + 1));
   if (!(min_den > 0.0)) min_den = 1e-300;
   ColScal z, sh;
   QmrPrev pv;
   memset(&pv, 0, sizeof(pv)); memset(&z, 0, sizeof(z));
   pv.eps = mach_eps;
   for (int c = 0; c < nx; c++) {
      pv.rho_prev[c] = rho_prev_host[c]; pv.tau_prev[c] = tau_prev_host[c]; pv.theta_prev[c] = theta_prev_host[c];
      sh.a[c] = shift_host ? shift_host[c] : 0.0;
   }
   int gx = hipk_grid_for_rows(ctx, m, HIPK_BLOCK * 4, 4);
   if (hipk_reserve_partials(ctx, (size_t)gx * nx)) return -2;
   DISPATCH_RT(dt,
         hipLaunchKernelGGL((qmr_update_dir_kernel<T, true>), dim3(gx), dim3(HIPK_BLOCK), 0, ctx->stream, z, z, z, sh, min_den, (T *)D, ldD, (T *)Delta, ldDelta, (T *)Sol, ldSol, (const T *)G, ldG, (const T *)diag, nx, 0, m, ctx->partials, tri_dev, ggr_dev, pv),
         hipLaunchKernelGGL((qmr_update_dir_kernel<T, true>), dim3(gx), dim3(HIPK_BLOCK), 0, ctx->stream, z, z, z, sh, min_den, (T *)D, ldD, (T *)Delta, ldDelta, (T *)Sol, ldSol, (const T *)G, ldG, (const T *)diag, nx, 0, m, ctx->partials, tri_dev, ggr_dev, pv));
   HIPK_CHECK(hipGetLastError());
   return hipk_finalize_partials(ctx, ctx->partials, gx, nx, dotsol_dev);
}

extern "C" int hipk_triple_dots(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const void *X, int64_t ldX, const void *V, int64_t ldV,
      const void *W, int64_t ldW, int nx, double *out_dev) {
   if (nx <= 0) return 0;
   hipk_prof_scope ps_(HIPK_PROF_VEC, ctx->stream, (double)m * (double)(HIPK_IS_Z(dt) ? 2 : 1) * (dt == HIPK_F64 || dt == HIPK_C64 ? 8.0 : 4.0) * (double)(3 * nx));
   int gx = hipk_grid_for_rows(ctx, m, HIPK_BLOCK * 4, 4);
   if (hipk_reserve_partials(ctx, (size_t)gx * 3 * nx)) return -2;
   DISPATCH_RT(dt,
         hipLaunchKernelGGL(triple_dots_kernel<T>, dim3(gx), dim3(HIPK_BLOCK), 0, ctx->stream, (const T *)X, ldX, (const T *)V, ldV, (const T *)W, ldW, nx, m, ctx->partials),
         hipLaunchKernelGGL(triple_dots_kernel<T>, dim3(gx), dim3(HIPK_BLOCK), 0, ctx->stream, (const T *)X, ldX, (const T *)V, ldV, (const T *)W, ldW, nx, m, ctx->partials));
   HIPK_CHECK(hipGetLastError());
   return hipk_finalize_partials(ctx, ctx->partials, gx, 3 * nx, out_dev);
}

/* W <- W - [segs] coef, then out = [x'w | v'w | v'x] for the updated W: the projection of (A - shift) d against the locked
 * vectors (the reference's apply_projected_matrix, inner_solve.c:853-880, Num_gemm + Num_dist_dots) and the three inner
 * products of the block QMR step in ONE pass — the projected panel used to be written by project_kernel and read back by
 * triple_dots_kernel.  Same arithmetic as that pair of launches in the same order (coefficients applied with one fma each in
 * column order; a thread walks the rows triple_dots_kernel gives it, so every partial sum is that kernel's): identical bits,
 * one read of nx columns and one launch less per inner step (round 6). */
template <typename T, int NX>
__global__ void __launch_bounds__(HIPK_BLOCK)
project_triple_kernel(SegArgs segs, const double *__restrict__ coef, int ldcoef, T *__restrict__ Wv, int64_t ldW,
      const T *__restrict__ X, int64_t ldX, const T *__restrict__ Vv, int64_t ldV, int nx, int64_t m, double *__restrict__ partials) {
   __shared__ double scoef[PROJ_MAXCOLS * NX];
   __shared__ const T *sptr[PROJ_MAXCOLS];
   __shared__ double sm[HIPK_BLOCK / HIPK_WAVE][3][NX];
   const int total = segs.total;
   for (int t = threadIdx.x; t < total * NX; t += HIPK_BLOCK) {
      const int j = t / NX, c = t % NX;
      scoef[t] = (c < nx) ? coef[j + (size_t)c * ldcoef] : 0.0;
   }
   for (int j = threadIdx.x; j < total; j += HIPK_BLOCK) sptr[j] = seg_col<T>(segs, j);
   __syncthreads();
   double a[NX], b[NX], d[NX];
#pragma unroll
   for (int c = 0; c < NX; c++) { a[c] = 0.0; b[c] = 0.0; d[c] = 0.0; }
   const int64_t stride = (int64_t)gridDim.x * HIPK_BLOCK;
   for (int64_t i = (int64_t)blockIdx.x * HIPK_BLOCK + threadIdx.x; i < m; i += stride) {
      double wv[NX], xv[NX], vv[NX];
#pragma unroll
      for (int c = 0; c < NX; c++) {
         const int cc = c < nx ? c : 0;
         wv[c] = (double)Wv[i + (size_t)cc * ldW];
         xv[c] = (double)X[i + (size_t)cc * ldX];
         vv[c] = (double)Vv[i + (size_t)cc * ldV];
      }
      int j = 0;
      for (; j + 8 <= total; j += 8) {
         double q[8];
#pragma unroll
         for (int u = 0; u < 8; u++) q[u] = (double)__builtin_nontemporal_load(sptr[j + u] + i);
#pragma unroll
         for (int u = 0; u < 8; u++)
#pragma unroll
            for (int c = 0; c < NX; c++) wv[c] = fma(-q[u], scoef[(j + u) * NX + c], wv[c]);
      }
      for (; j < total; j++) {
         const double q = (double)sptr[j][i];
#pragma unroll
         for (int c = 0; c < NX; c++) wv[c] = fma(-q, scoef[j * NX + c], wv[c]);
      }
#pragma unroll
      for (int c = 0; c < NX; c++)
         if (c < nx) {
            const T o = (T)wv[c];
            Wv[i + (size_t)c * ldW] = o;
            const double wi = (double)o;
            a[c] = fma(xv[c], wi, a[c]); b[c] = fma(vv[c], wi, b[c]); d[c] = fma(vv[c], xv[c], d[c]);
         }
   }
   const int lane = threadIdx.x & 63, wvn = threadIdx.x >> 6;
#pragma unroll
   for (int c = 0; c < NX; c++) {
      const double ta = hipk_wave_sum(a[c]), tb = hipk_wave_sum(b[c]), td = hipk_wave_sum(d[c]);
      if (lane == 0) { sm[wvn][0][c] = ta; sm[wvn][1][c] = tb; sm[wvn][2][c] = td; }
   }
   __syncthreads();
   for (int t = threadIdx.x; t < 3 * nx; t += HIPK_BLOCK) {
      const int w = t / nx, c = t % nx;
      partials[(size_t)blockIdx.x * 3 * nx + w * nx + c] = (sm[0][w][c] + sm[1][w][c]) + (sm[2][w][c] + sm[3][w][c]);
   }
}

extern "C" int hipk_project_triple_dots(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const hipk_seg *segs, int nseg, const double *coef_dev,
      int ldcoef, void *W, int64_t ldW, int nx, const void *X, int64_t ldX, const void *V, int64_t ldV, double *out_dev) {
   if (nx <= 0) return 0;
   SegArgs sa;
   if (HIPK_IS_Z(dt) || pack_segs(segs, nseg, &sa) || nx > 8 || sa.total > PROJ_MAXCOLS || sa.total <= 0) return 1;   /* not covered: the caller runs the two launches */
   const double es = dt == HIPK_F64 ? 8.0 : 4.0;
   int gx = hipk_grid_for_rows(ctx, m, HIPK_BLOCK * 4, 4);        /* = hipk_triple_dots: the same partial sums */
   if (hipk_reserve_partials(ctx, (size_t)gx * 3 * nx)) return -2;
   {
      hipk_prof_scope ps_(HIPK_PROF_VEC, ctx->stream, (double)m * es * ((double)sa.total + 4.0 * nx));
#define PTK(NXV) DISPATCH_RT(dt, \
         hipLaunchKernelGGL((project_triple_kernel<T, NXV>), dim3(gx), dim3(HIPK_BLOCK), 0, ctx->stream, sa, coef_dev, ldcoef, (T *)W, ldW, (const T *)X, ldX, (const T *)V, ldV, nx, m, ctx->partials), \
         hipLaunchKernelGGL((project_triple_kernel<T, NXV>), dim3(gx), dim3(HIPK_BLOCK), 0, ctx->stream, sa, coef_dev, ldcoef, (T *)W, ldW, (const T *)X, ldX, (const T *)V, ldV, nx, m, ctx->partials))
      if (nx <= 2) { PTK(2); } else if (nx <= 4) { PTK(4); } else { PTK(8); }
#undef PTK
      HIPK_CHECK(hipGetLastError());
   }
   return hipk_finalize_partials(ctx, ctx->partials, gx, 3 * nx, out_dev);
}

extern "C" int hipk_axpy_proj_dot(hipk_ctx *ctx, hipk_dtype dt, int64_t m, int nx, const double *alpha_host, const double *xr_host,
      const void *W, int64_t ldW, const void *X, int64_t ldX, void *G, int64_t ldG, double *out_dev) {
   if (nx <= 0) return 0;
   if (nx > UTIL_MAXCOLS) return -1;
   hipk_prof_scope ps_(HIPK_PROF_VEC, ctx->stream, (double)m * (double)(HIPK_IS_Z(dt) ? 2 : 1) * (dt == HIPK_F64 || dt == HIPK_C64 ? 8.0 : 4.0) * (double)(4 * nx));
   ColScal a, r;
   for (int c = 0; c < nx; c++) { a.a[c] = alpha_host[c]; r.a[c] = xr_host[c]; }
   int gx = hipk_grid_for_rows(ctx, m, HIPK_BLOCK * 4, 4);
   if (hipk_reserve_partials(ctx, (size_t)gx * nx)) return -2;
   DISPATCH_RT(dt,
         hipLaunchKernelGGL(axpy_proj_dot_kernel<T>, dim3(gx), dim3(HIPK_BLOCK), 0, ctx->stream, a, r, (const T *)W, ldW, (const T *)X, ldX, (T *)G, ldG, nx, m, ctx->partials),
         hipLaunchKernelGGL(axpy_proj_dot_kernel<T>, dim3(gx), dim3(HIPK_BLOCK), 0, ctx->stream, a, r, (const T *)W, ldW, (const T *)X, ldX, (T *)G, ldG, nx, m, ctx->partials));
   HIPK_CHECK(hipGetLastError());
   return hipk_finalize_partials(ctx, ctx->partials, gx, nx, out_dev);
}
