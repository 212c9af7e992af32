/* hipk_complex.hip — the complex instantiation (HIPK_C64 / HIPK_C32) of the device layer: tall-skinny panel
 * kernels and the CSR SpMV / SpMM on complex data, what the reference instantiates with SCALAR = complex
 * (src/include/template_types.h:91; Num_gemm_ddh "C","N" = conjugate-transposed left operand, Num_dot = xDOTC,
 * src/linalg/cublas_wrapper.c:479-499, :647).  hip_zprimme / hip_cprimme run on these natively
 * (csrc/eigs_complex.c, eigs_scalar.h).
 *
 * Conventions (include/primme_amd_kernels.h): a panel element is an interleaved (re, im) pair; leading dimensions
 * count complex elements; inner products, projection coefficients, Ritz coefficient vectors and axpy factors are
 * (re, im) pairs of doubles; Ritz values, shifts, squared norms and scale factors are real.
 *
 * Shape of every kernel: HBM-bound streaming, one lane per row (a complex double is one 16-byte load, so a wave
 * reads 1 KB of a column per instruction, fully coalesced), double accumulators, two-stage reductions with o-major
 * partials and the library's fixed-order second stage (bit-reproducible, identical on every rank; the second
 * stage also publishes the completion flag the host spins on).  Operations whose arithmetic does not mix real and
 * imaginary parts (copy, gather, real scale, squared norms, w -= theta x) are the REAL kernels on the panel seen
 * as 2m reals: hipk_*_cols dispatch them that way and nothing here repeats them.
 */
#include "hipk_internal.h"
#include <vector>

template <typename R> struct __attribute__((aligned(2 * sizeof(R)))) cpx { R re, im; };
struct zacc { double re, im; };
template <typename R> __device__ __forceinline__ zacc zload(const cpx<R> *p) { const cpx<R> v = *p; return {(double)v.re, (double)v.im}; }
/* a streamed panel element: with the non-temporal hint (as in hipk_panels.hip: the panels never stay in the Infinity
 * Cache, the hint keeps them from evicting what does).  HIPK_Z_NT=0 at build time = plain loads (A/B builds). */
#ifndef HIPK_Z_NT
#define HIPK_Z_NT 1
#endif
template <typename R> __device__ __forceinline__ zacc zload_s(const cpx<R> *p) {
   if (HIPK_Z_NT) {
      typedef R nvec __attribute__((ext_vector_type(2)));
      const nvec t = __builtin_nontemporal_load((const nvec *)p);
      return {(double)t[0], (double)t[1]};
   }
   return zload(p);
}
template <typename R> __device__ __forceinline__ void zstore(cpx<R> *p, zacc v) { cpx<R> o; o.re = (R)v.re; o.im = (R)v.im; *p = o; }
/* a += conj(x) * y */
__device__ __forceinline__ void zfma_conj(zacc &a, const zacc &x, const zacc &y) {
   a.re = fma(x.re, y.re, fma(x.im, y.im, a.re));
   a.im = fma(x.re, y.im, fma(-x.im, y.re, a.im));
}
/* a += x * y */
__device__ __forceinline__ void zfma(zacc &a, const zacc &x, const zacc &y) {
   a.re = fma(x.re, y.re, fma(-x.im, y.im, a.re));
   a.im = fma(x.re, y.im, fma(x.im, y.re, a.im));
}
/* a -= x * y */
__device__ __forceinline__ void zfms(zacc &a, const zacc &x, const zacc &y) {
   a.re = fma(-x.re, y.re, fma(x.im, y.im, a.re));
   a.im = fma(-x.re, y.im, fma(-x.im, y.re, a.im));
}

#define ZSEG_MAX 4
struct ZSegs { const void *base[ZSEG_MAX]; int64_t ld[ZSEG_MAX]; int n[ZSEG_MAX]; int nseg; int total; };
static int zpack(const hipk_seg *segs, int nseg, ZSegs *sa) {
   if (nseg > ZSEG_MAX) return -1;
   sa->nseg = 0; sa->total = 0;
   for (int s = 0; s < nseg; s++) {
      if (segs[s].ncols <= 0) continue;
      sa->base[sa->nseg] = segs[s].base; sa->ld[sa->nseg] = segs[s].ld; sa->n[sa->nseg] = segs[s].ncols;
      sa->nseg++; sa->total += segs[s].ncols;
   }
   for (int s = sa->nseg; s < ZSEG_MAX; s++) { sa->base[s] = NULL; sa->ld[s] = 0; sa->n[s] = 0; }
   return 0;
}
template <typename R> __device__ __forceinline__ const cpx<R> *zseg_col(const ZSegs &sa, int j) {
#pragma unroll
   for (int s = 0; s < ZSEG_MAX; s++) {
      if (j < sa.n[s]) return (const cpx<R> *)sa.base[s] + (size_t)j * (size_t)sa.ld[s];
      j -= sa.n[s];
   }
   return NULL;
}

/* block-wide sum of one double per lane, result valid in thread 0 */
__device__ __forceinline__ double zblock_sum(double v, double *sm /* [HIPK_BLOCK / 64] */) {
   v = hipk_wave_sum(v);
   __syncthreads();
   if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
   __syncthreads();
   return (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

static int zgrid(const hipk_ctx *ctx, int64_t m, int per_cu) {
   int64_t need = (m + HIPK_BLOCK - 1) / HIPK_BLOCK;
   if (need < 1) need = 1;
   const int64_t cap = (int64_t)ctx->num_cu * per_cu;
   return (int)(need < cap ? need : cap);
}

/* ============================ TN panel: out = [segs]^H X =============================================
 * Column-split (like the real ritz_cgs_kernel): the four waves of a workgroup walk the SAME rows, wave w owns basis
 * columns [j0 + w CPW, j0 + (w + 1) CPW) and keeps CPW x NX complex accumulators per lane.  Every basis column is read
 * once; the NX right-hand columns are requested by all four waves but come out of HBM once per workgroup (the other
 * three requests hit the CU's L1 / the XCD's L2).  blockIdx.y = group of 4 CPW basis columns.  No cross-wave
 * reduction: a wave's lanes are summed with shuffles and lane 0 stores partials[o * nblocks + block],
 * o = 2 (j + c * tot) + {re, im}. */
template <typename R, int NX, int CPW>
__global__ void __launch_bounds__(HIPK_BLOCK)
zdots_kernel(ZSegs sa, const cpx<R> *__restrict__ X, int64_t ldX, int nx, int c0, int64_t m, int tot,
      double *__restrict__ partials) {
   const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
   const int j0 = blockIdx.y * (4 * CPW) + wave * CPW;
   const cpx<R> *col[CPW];
#pragma unroll
   for (int t = 0; t < CPW; t++) col[t] = zseg_col<R>(sa, j0 + t < tot ? j0 + t : tot - 1);
   const int nxv = min(NX, nx - c0);
   zacc acc[CPW][NX];
#pragma unroll
   for (int t = 0; t < CPW; t++)
#pragma unroll
      for (int c = 0; c < NX; c++) acc[t][c] = {0.0, 0.0};
   const int64_t stride = (int64_t)gridDim.x * 64;
   if (j0 < tot) {
      for (int64_t i = (int64_t)blockIdx.x * 64 + lane; i < m; i += stride) {
         zacc xv[NX];
#pragma unroll
         for (int c = 0; c < NX; c++) xv[c] = zload(X + (size_t)(c0 + (c < nxv ? c : 0)) * ldX + i);
         zacc av[CPW];
#pragma unroll
         for (int t = 0; t < CPW; t++) av[t] = zload_s(col[t] + i);
#pragma unroll
         for (int t = 0; t < CPW; t++)
#pragma unroll
            for (int c = 0; c < NX; c++) zfma_conj(acc[t][c], av[t], xv[c]);
      }
   }
   const unsigned nb = gridDim.x;
#pragma unroll
   for (int t = 0; t < CPW; t++)
#pragma unroll
      for (int c = 0; c < NX; c++) {
         const double re = hipk_wave_sum(acc[t][c].re), im = hipk_wave_sum(acc[t][c].im);
         if (lane == 0 && j0 + t < tot && c < nxv) {
            const size_t o = 2 * ((size_t)(j0 + t) + (size_t)(c0 + c) * tot);
            partials[o * nb + blockIdx.x] = re;
            partials[(o + 1) * nb + blockIdx.x] = im;
         }
      }
}

/* ---- the same product on the MATRIX CORES (round 6; the real panels' dots_mfma_kernel, hipk_panels.hip) ----------------
 * A complex m x k panel IS a real 2m x k panel V_r (rows re, im interleaved), and with X = p + i q
 *    Re([V]^H X) = V_r' X_r,        Im([V]^H X) = V_r' (J X_r),   J (p, q) = (q, -p)  pair by pair,
 * so the conjugated TN product is ONE real TN product with twice the right-hand columns: [X_r | J X_r].  Tiles of 64
 * complex rows (128 real rows of v_mfma_f64_16x16x4_f64's k dimension) of up to 16 NT basis columns and 8 complex
 * right-hand columns are staged in LDS with one 16-byte load per lane and column (1 KB per wave instruction, every load
 * of a tile issued before the first LDS store; a right-hand element is stored twice, as (p, q) and as (q, -p)); operands
 * are read back with a column stride of 130 doubles (bank-conflict free for the 32-lane halves of ds_read_b64); each wave
 * multiplies a quarter of the tile's rows, accumulators in the C/D layout (column = lane & 15, row = (lane >> 4) + 4 reg),
 * the four waves' tiles are added in a fixed order.  16 real accumulators per 16 basis columns and lane instead of
 * 2 CPW NX doubles: the register budget no longer sets the occupancy.  blockIdx.y: group of 16 NT basis columns,
 * blockIdx.z: group of 8 right-hand columns.  Partials as zdots_kernel's: o = 2 (j + c tot) + {re, im}, o-major. */
#define ZMF_ROWS 128                 /* real rows per tile = 64 complex rows */
#define ZMF_STRIDE (ZMF_ROWS + 2)
typedef double zmf_acc __attribute__((ext_vector_type(4)));
template <typename R, int NT>
__global__ void __launch_bounds__(HIPK_BLOCK)
zdots_mfma_kernel(ZSegs sa, const cpx<R> *__restrict__ X, int64_t ldX, int nx, int64_t m, int tot, double *__restrict__ partials) {
   extern __shared__ double zmf_lds[];           /* [(16 NT + 16) columns][ZMF_STRIDE] */
   const int lane = threadIdx.x & 63;
   const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
   const int j0 = blockIdx.y * 16 * NT;
   const int ncv = min(16 * NT, tot - j0);
   const int c0 = blockIdx.z * 8;
   const int nxv = min(8, nx - c0);
   double *sV = zmf_lds, *sX = zmf_lds + (size_t)16 * NT * ZMF_STRIDE;
   zmf_acc acc[NT];
#pragma unroll
   for (int t = 0; t < NT; t++) acc[t] = (zmf_acc){0.0, 0.0, 0.0, 0.0};
   const int ci = lane & 15, kg = lane >> 4;
   constexpr int NVW = 4 * NT;                    /* basis columns a wave stages per tile */
   const cpx<R> *vcol[NVW], *xcol[2];
#pragma unroll
   for (int q = 0; q < NVW; q++) { const int c = wv + 4 * q; vcol[q] = zseg_col<R>(sa, j0 + (c < ncv ? c : 0)); }
#pragma unroll
   for (int q = 0; q < 2; q++) { const int c = wv + 4 * q; xcol[q] = X + (size_t)(c0 + (c < nxv ? c : 0)) * ldX; }
   const int64_t ntile = (m + 63) / 64;
   for (int64_t tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
      const int64_t i = tile * 64 + lane;
      const bool live = i < m;
      const int64_t ic = live ? i : m - 1;
      zacc tv[NVW], tx[2];
#pragma unroll
      for (int q = 0; q < NVW; q++) tv[q] = zload_s(vcol[q] + ic);
#pragma unroll
      for (int q = 0; q < 2; q++) tx[q] = zload(xcol[q] + ic);
#pragma unroll
      for (int q = 0; q < NVW; q++) {
         const int c = wv + 4 * q;
         const bool have = live && c < ncv;
         double *dst = sV + (size_t)c * ZMF_STRIDE + 2 * lane;
         dst[0] = have ? tv[q].re : 0.0;
         dst[1] = have ? tv[q].im : 0.0;
      }
#pragma unroll
      for (int q = 0; q < 2; q++) {
         const int c = wv + 4 * q;
         const bool have = live && c < nxv;
         double *d0 = sX + (size_t)c * ZMF_STRIDE + 2 * lane, *d1 = sX + (size_t)(8 + c) * ZMF_STRIDE + 2 * lane;
         d0[0] = have ? tx[q].re : 0.0;  d0[1] = have ? tx[q].im : 0.0;
         d1[0] = have ? tx[q].im : 0.0;  d1[1] = have ? -tx[q].re : 0.0;
      }
      __syncthreads();
#pragma unroll
      for (int st = 0; st < ZMF_ROWS / 16; st++) {
         const int row = wv * (ZMF_ROWS / 4) + 4 * st + kg;
         const double b = sX[(size_t)ci * ZMF_STRIDE + row];
#pragma unroll
         for (int t = 0; t < NT; t++) {
            const double a = sV[(size_t)(16 * t + ci) * ZMF_STRIDE + row];
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
         }
      }
      __syncthreads();
   }
   double *red = zmf_lds;
#pragma unroll
   for (int t = 0; t < NT; t++)
#pragma unroll
      for (int r = 0; r < 4; r++) red[(((size_t)wv * NT + t) * 4 + r) * 64 + lane] = acc[t][r];
   __syncthreads();
   if (wv == 0) {
      const unsigned nb = gridDim.x;
#pragma unroll
      for (int t = 0; t < NT; t++)
#pragma unroll
         for (int r = 0; r < 4; r++) {
            const size_t o = ((size_t)t * 4 + r) * 64 + lane;
            const double v = (red[o] + red[(size_t)NT * 256 + o]) + (red[(size_t)2 * NT * 256 + o] + red[(size_t)3 * NT * 256 + o]);
            const int iv = 16 * t + kg + 4 * r, jc = ci & 7, part = ci >> 3;     /* C/D layout: row = basis column, column = right-hand column */
            if (iv < ncv && jc < nxv) {
               const size_t oo = 2 * ((size_t)(j0 + iv) + (size_t)(c0 + jc) * tot) + (size_t)part;
               partials[oo * nb + blockIdx.x] = v;
            }
         }
   }
}
static int g_zmfma = -1;                         /* HIPK_ZMFMA=1 / hipk_set_zdots_mfma(1): the matrix-core form (measured: no faster) */
static int zdots_mfma_enabled(void) {
   if (g_zmfma < 0) { const char *e = getenv("HIPK_ZMFMA"); g_zmfma = (e && atoi(e) != 0 && getenv("HIPK_NO_MFMA") == NULL) ? 1 : 0; }
   return g_zmfma;
}
extern "C" int hipk_set_zdots_mfma(int on) { const int old = zdots_mfma_enabled(); g_zmfma = on != 0; return old; }

template <typename R>
static int zpanel_dots_t(hipk_ctx *ctx, int64_t m, const ZSegs &sa, const void *X, int64_t ldX, int nx, double *out_dev, int ldout) {
   const int tot = sa.total;
   int64_t need = (m + 63) / 64;
   if (need < 1) need = 1;
   static int bpc = -1, mbpc = -1;                 /* HIPK_ZDOTS_BPC / HIPK_ZMFMA_BPC: workgroups per CU (measurement knobs, read once) */
   if (bpc < 0) { const char *e = getenv("HIPK_ZDOTS_BPC"); bpc = e ? atoi(e) : 4; if (bpc < 1) bpc = 4; }      /* 4: in isolation 2 per CU is faster (0.80 against 0.68-0.78 of HBM), inside the configs[3] solve it is SLOWER (0.395-0.405 s against 0.386-0.388 s per solve, same box): profiles/r06_zpanel_perf.txt */
   if (mbpc < 0) { const char *e = getenv("HIPK_ZMFMA_BPC"); mbpc = e ? atoi(e) : 2; if (mbpc < 1) mbpc = 2; }
   const int gx = (int)(need < (int64_t)ctx->num_cu * bpc ? need : (int64_t)ctx->num_cu * bpc);     /* a workgroup walks 64 rows per step */
   const size_t nout = 2 * (size_t)tot * nx;
   if (hipk_reserve_partials(ctx, nout * gx)) return -2;
   const int pslot = hipk_prof_begin(HIPK_PROF_DOTS, ctx->stream, (double)(tot + nx) * (double)m * 2.0 * sizeof(R));
   /* blocks of >= 4 right-hand columns: the matrix cores (zdots_mfma_kernel) */
   if (nx >= 4 && zdots_mfma_enabled()) {
      const int nt = tot <= 16 ? 1 : 2;
      const int gy = (tot + 16 * nt - 1) / (16 * nt), gz = (nx + 7) / 8;
      int gm = (int)(need < (int64_t)ctx->num_cu * mbpc ? need : (int64_t)ctx->num_cu * mbpc);
      while (gm > 1 && (int64_t)gm * gy * gz > (int64_t)ctx->num_cu * 2 * mbpc) gm = (gm + 1) / 2;
      const size_t shm = (size_t)(16 * nt + 16) * ZMF_STRIDE * sizeof(double);
      if (nt == 1) hipLaunchKernelGGL((zdots_mfma_kernel<R, 1>), dim3(gm, gy, gz), dim3(HIPK_BLOCK), shm, ctx->stream, sa, (const cpx<R> *)X, ldX, nx, m, tot, ctx->partials);
      else hipLaunchKernelGGL((zdots_mfma_kernel<R, 2>), dim3(gm, gy, gz), dim3(HIPK_BLOCK), shm, ctx->stream, sa, (const cpx<R> *)X, ldX, nx, m, tot, ctx->partials);
      hipk_prof_end(pslot, ctx->stream);
      HIPK_CHECK(hipGetLastError());
      if (ldout == tot) return hipk_finalize_partials_t(ctx, ctx->partials, gm, (int)nout, out_dev);
      for (int c = 0; c < nx; c++) {
         int rc = hipk_finalize_partials_t(ctx, ctx->partials + (size_t)2 * tot * c * gm, gm, 2 * tot, out_dev + (size_t)2 * ldout * c);
         if (rc) return rc;
      }
      return 0;
   }
   for (int c0 = 0; c0 < nx; c0 += 8) {
      const int nc = nx - c0;
      /* columns per wave: as many as the accumulators allow (CPW * NX <= 32 complex = 64 doubles), no more than needed */
      const int want = (tot + 3) / 4;
#define ZD(NXV, CPWV) hipLaunchKernelGGL((zdots_kernel<R, NXV, CPWV>), dim3(gx, (tot + 4 * CPWV - 1) / (4 * CPWV)), dim3(HIPK_BLOCK), 0, ctx->stream, \
            sa, (const cpx<R> *)X, ldX, nx, c0, m, tot, ctx->partials)
      if (nc == 1) { if (want <= 2) ZD(1, 2); else if (want <= 4) ZD(1, 4); else if (want <= 8) ZD(1, 8); else ZD(1, 16); }
      else if (nc == 2) { if (want <= 2) ZD(2, 2); else if (want <= 4) ZD(2, 4); else ZD(2, 8); }
      else if (nc <= 4) { if (want <= 2) ZD(4, 2); else if (want <= 4) ZD(4, 4); else ZD(4, 8); }
      else { if (want <= 2) ZD(8, 2); else ZD(8, 4); }
#undef ZD
   }
   hipk_prof_end(pslot, ctx->stream);
   HIPK_CHECK(hipGetLastError());
   if (ldout == tot) return hipk_finalize_partials_t(ctx, ctx->partials, gx, (int)nout, out_dev);
   for (int c = 0; c < nx; c++) {      /* padded output columns: one second stage per column */
      int rc = hipk_finalize_partials_t(ctx, ctx->partials + (size_t)2 * tot * c * gx, gx, 2 * tot, out_dev + (size_t)2 * ldout * c);
      if (rc) return rc;
   }
   return 0;
}

/* ============================ NN panel: X -= [segs] coef (x M), |.|^2 =================================
 * coefficients (and M) staged in LDS; a lane streams the basis columns of its row once and updates NX values. */
#define ZPROJ_LDS 2048      /* complex coefficients per launch: tot * NX <= this (32 KB + the column pointers) */
template <typename R, int NX, bool MUL>
__global__ void __launch_bounds__(HIPK_BLOCK)
zproject_kernel(ZSegs sa, const zacc *__restrict__ coef, int ldcoef, const zacc *__restrict__ M, const cpx<R> *X, int64_t ldX,
      cpx<R> *Xout, int64_t ldXout, int nx, int c0, int64_t m, int tot, double *__restrict__ partials) {
   extern __shared__ __attribute__((aligned(16))) char zsm_raw[];
   zacc *scoef = (zacc *)zsm_raw;                 /* tot * NX */
   zacc *sM = scoef + (size_t)tot * NX;           /* NX * NX (MUL) */
   const cpx<R> **sptr = (const cpx<R> **)(sM + (MUL ? NX * NX : 0));    /* tot column pointers */
   __shared__ double sm[HIPK_BLOCK / HIPK_WAVE];
   for (int j = threadIdx.x; j < tot; j += HIPK_BLOCK) sptr[j] = zseg_col<R>(sa, j);
   const int nxv = min(NX, nx - c0);
   for (int t = threadIdx.x; t < tot * NX; t += HIPK_BLOCK) {
      const int j = t / NX, c = t % NX;
      scoef[t] = (c < nxv) ? coef[j + (size_t)(c0 + c) * ldcoef] : zacc{0.0, 0.0};
   }
   if (MUL) for (int t = threadIdx.x; t < NX * NX; t += HIPK_BLOCK) {
      const int q = t / NX, c = t % NX;           /* sM[q * NX + c] = M(q, c) */
      sM[t] = (q < nx && c < nx) ? M[q + (size_t)c * nx] : zacc{0.0, 0.0};
   }
   __syncthreads();
   double n2[NX];
#pragma unroll
   for (int c = 0; c < NX; c++) n2[c] = 0.0;
   const int64_t stride = (int64_t)gridDim.x * HIPK_BLOCK;
   for (int64_t i = (int64_t)blockIdx.x * HIPK_BLOCK + threadIdx.x; i < m; i += stride) {
      zacc xv[NX];
#pragma unroll
      for (int c = 0; c < NX; c++) xv[c] = (c < nxv) ? zload(X + (size_t)(c0 + c) * ldX + i) : zacc{0.0, 0.0};
      for (int j = 0; j < tot; j++) {
         const zacc a = zload_s(sptr[j] + i);
#pragma unroll
         for (int c = 0; c < NX; c++) zfms(xv[c], a, scoef[j * NX + c]);
      }
      if (MUL) {
         zacc o[NX];
#pragma unroll
         for (int c = 0; c < NX; c++) {
            o[c] = {0.0, 0.0};
#pragma unroll
            for (int q = 0; q < NX; q++) zfma(o[c], xv[q], sM[q * NX + c]);
         }
#pragma unroll
         for (int c = 0; c < NX; c++) if (c < nxv) zstore(Xout + (size_t)(c0 + c) * ldXout + i, o[c]);
      } else {
#pragma unroll
         for (int c = 0; c < NX; c++) if (c < nxv) {
            cpx<R> st; st.re = (R)xv[c].re; st.im = (R)xv[c].im;       /* the norm of what is stored */
            Xout[(size_t)(c0 + c) * ldXout + i] = st;
            n2[c] = fma((double)st.re, (double)st.re, fma((double)st.im, (double)st.im, n2[c]));
         }
      }
   }
   if (partials) {
      const unsigned nb = gridDim.x;
#pragma unroll
      for (int c = 0; c < NX; c++) {
         const double v = zblock_sum(n2[c], sm);
         if (threadIdx.x == 0 && c < nxv) partials[(size_t)(c0 + c) * nb + blockIdx.x] = v;
      }
   }
}

template <typename R>
static int zproject_t(hipk_ctx *ctx, int64_t m, const ZSegs &sa, const double *coef, int ldcoef, const double *M, const void *X,
      int64_t ldX, void *Xout, int64_t ldXout, int nx, double *nrm2_dev) {
   const int tot = sa.total;
   const int gx = zgrid(ctx, m, 4);
   const bool mul = M != NULL;
   if (mul && nx > 8) return 1;
   if (nrm2_dev && hipk_reserve_partials(ctx, (size_t)nx * gx)) return -2;
   const int pslot = hipk_prof_begin(HIPK_PROF_PROJECT, ctx->stream, (double)(tot + 2 * nx) * (double)m * 2.0 * sizeof(R));
   for (int c0 = 0; c0 < nx;) {
      const int nc = nx - c0;
      int NXV = mul ? (nx <= 1 ? 1 : nx <= 2 ? 2 : nx <= 4 ? 4 : 8) : (nc >= 4 ? 4 : nc >= 2 ? 2 : 1);
      while (!mul && NXV > 1 && (size_t)tot * NXV > ZPROJ_LDS) NXV >>= 1;
      if ((size_t)tot * NXV > ZPROJ_LDS) return -1;
      const size_t lds = ((size_t)tot * NXV + (mul ? NXV * NXV : 0)) * sizeof(zacc) + (size_t)tot * sizeof(void *);
#define ZP(NV, MV) hipLaunchKernelGGL((zproject_kernel<R, NV, MV>), dim3(gx), dim3(HIPK_BLOCK), lds, ctx->stream, sa, (const zacc *)coef, ldcoef, \
            (const zacc *)M, (const cpx<R> *)X, ldX, (cpx<R> *)Xout, ldXout, nx, c0, m, tot, nrm2_dev ? ctx->partials : (double *)NULL)
      if (mul) { if (NXV == 1) ZP(1, true); else if (NXV == 2) ZP(2, true); else if (NXV == 4) ZP(4, true); else ZP(8, true); c0 = nx; }
      else { if (NXV == 1) ZP(1, false); else if (NXV == 2) ZP(2, false); else ZP(4, false); c0 += NXV; }
#undef ZP
   }
   hipk_prof_end(pslot, ctx->stream);
   HIPK_CHECK(hipGetLastError());
   if (nrm2_dev) return hipk_finalize_partials_t(ctx, ctx->partials, gx, nx, nrm2_dev);
   return 0;
}

/* ============================ fused Ritz / residual / restart update ==================================
 * out_q(i) = sum_j {V | W | W - theta_q V}(i, j) h(j, col_q): a lane streams its row of V and W ONCE, keeps one
 * complex accumulator per job and stores after the last column — destinations may alias columns of V and W.
 * The launcher orders the jobs: plain products first (NJ - NR slots), residual jobs last (NR slots), so that the
 * kind of a slot is known at compile time and only the residual slots carry a squared-norm accumulator. */
#define ZRITZ_JOBS 32
#define ZRITZ_JT 64
struct ZJobs { void *dst[ZRITZ_JOBS]; int col[ZRITZ_JOBS]; signed char isw[ZRITZ_JOBS]; short slot[ZRITZ_JOBS]; };
template <typename R, int NJ, int NR>
__global__ void __launch_bounds__(HIPK_BLOCK)
zritz_kernel(const cpx<R> *V, const cpx<R> *W, int64_t ld, int k, const zacc *__restrict__ h, int ldh,
      const double *__restrict__ theta, ZJobs jb, int64_t m, double *__restrict__ partials) {
   constexpr int NP = NJ - NR;                  /* plain products: slots [0, NP); residuals: [NP, NJ) */
   __shared__ zacc sh[ZRITZ_JT * NJ];           /* sh[jj * NJ + q] = h(j0 + jj, col_q) */
   __shared__ double sth[NR];
   __shared__ double sm[HIPK_BLOCK / HIPK_WAVE];
   for (int q = threadIdx.x; q < NR; q += HIPK_BLOCK) sth[q] = jb.col[NP + q] >= 0 ? theta[jb.col[NP + q]] : 0.0;
   const int64_t stride = (int64_t)gridDim.x * HIPK_BLOCK;
   const int64_t mpad = (m + stride - 1) / stride * stride;      /* every lane runs the same number of trips (barriers inside) */
   double n2[NR];
#pragma unroll
   for (int q = 0; q < NR; q++) n2[q] = 0.0;
   const bool once = k <= ZRITZ_JT;                /* the usual case: the coefficient block is staged once */
   if (once) {
      for (int t = threadIdx.x; t < k * NJ; t += HIPK_BLOCK) {
         const int jj = t / NJ, q = t % NJ;
         sh[t] = (jb.col[q] >= 0) ? h[jj + (size_t)jb.col[q] * ldh] : zacc{0.0, 0.0};
      }
   }
   __syncthreads();
   for (int64_t i = (int64_t)blockIdx.x * HIPK_BLOCK + threadIdx.x; i < mpad; i += stride) {
      const bool live = i < m;
      const int64_t ic = live ? i : m - 1;
      zacc acc[NJ];
#pragma unroll
      for (int q = 0; q < NJ; q++) acc[q] = {0.0, 0.0};
      for (int j0 = 0; j0 < k; j0 += ZRITZ_JT) {
         const int jn = min(ZRITZ_JT, k - j0);
         if (!once) {
            __syncthreads();
            for (int t = threadIdx.x; t < jn * NJ; t += HIPK_BLOCK) {
               const int jj = t / NJ, q = t % NJ;
               sh[t] = (jb.col[q] >= 0) ? h[(j0 + jj) + (size_t)jb.col[q] * ldh] : zacc{0.0, 0.0};
            }
            __syncthreads();
         }
#pragma unroll 2
         for (int jj = 0; jj < jn; jj++) {
            const zacc v = zload_s(V + (size_t)(j0 + jj) * ld + ic);
            const zacc w = W ? zload_s(W + (size_t)(j0 + jj) * ld + ic) : zacc{0.0, 0.0};
#pragma unroll
            for (int q = 0; q < NP; q++) zfma(acc[q], jb.isw[q] ? w : v, sh[jj * NJ + q]);
#pragma unroll
            for (int q = 0; q < NR; q++) {
               const zacc s = {fma(-sth[q], v.re, w.re), fma(-sth[q], v.im, w.im)};
               zfma(acc[NP + q], s, sh[jj * NJ + NP + q]);
            }
         }
      }
      if (live) {
#pragma unroll
         for (int q = 0; q < NJ; q++) {
            cpx<R> st; st.re = (R)acc[q].re; st.im = (R)acc[q].im;
            if (jb.dst[q]) ((cpx<R> *)jb.dst[q])[i] = st;
            if (q >= NP) n2[q - NP] = fma((double)st.re, (double)st.re, fma((double)st.im, (double)st.im, n2[q - NP]));
         }
      }
   }
   if (partials) {
      const unsigned nb = gridDim.x;
#pragma unroll
      for (int q = 0; q < NR; q++) {
         const bool want = jb.slot[NP + q] >= 0;
         const double v = zblock_sum(n2[q], sm);
         if (threadIdx.x == 0 && want) partials[(size_t)jb.slot[NP + q] * nb + blockIdx.x] = v;
      }
   }
}

/* The same with L = 2 or 4 lanes per row for more than 16 outputs: the L lanes of a row read the same elements of V and W
 * (one 16-byte request, merged by the coalescer) and each keeps NS = NVH + NWH + NRH <= 24 of the accumulators — up to 64
 * outputs in ONE pass over V and W, where the one-lane form holds 256 VGPRs at 32 outputs (one wave per SIMD, 3.1 TB/s
 * on the restart pass of configs[3]) and more than 32 used to take two passes through a temporary.  The lanes of a row
 * sit in the same wave, so every load of a row precedes every store to it (destinations may alias V and W).
 * A part's slots are [V-products: NVH | W-products: NWH | residuals: NRH], fixed at compile time: selecting v or w per
 * slot at run time costs four v_cndmask per complex FMA, as many VALU cycles as the FMA itself (SIMDs are 16 lanes wide,
 * every wave64 instruction takes four cycles), and this kernel is VALU / LDS bound, not HBM bound. */
#define ZRITZ2_SLOTS 24
template <int L> struct ZJobsL { void *dst[L][ZRITZ2_SLOTS]; int col[L][ZRITZ2_SLOTS]; short slot[L][ZRITZ2_SLOTS]; };
template <typename R, int L, int NVH, int NWH, int NRH>
__global__ void __launch_bounds__(HIPK_BLOCK)
zritz2_kernel(const cpx<R> *V, const cpx<R> *W, int64_t ld, int k, const zacc *__restrict__ h, int ldh,
      const double *__restrict__ theta, ZJobsL<L> jb, int64_t m, double *__restrict__ partials) {
   constexpr int NS = NVH + NWH + NRH, NPH = NVH + NWH, JT = 128 / L, ROWS = HIPK_BLOCK / L;
   /* the L parts of a wave read L different rows of sh in one ds_read_b128: a row stride of 16 complex (256 bytes = all
    * 64 banks) would put them on the same banks (the L = 4 form ran LDS-bound at 2.0 TB/s); NJS = NS + 1 staggers them */
   constexpr int NJS = NS + 1;
   __shared__ zacc sh[JT * L * NJS];              /* sh[(jj * L + part) * NJS + q] */
   __shared__ double sth[L][NRH];
   __shared__ void *sdst[L][NS];
   __shared__ double sm[HIPK_BLOCK / HIPK_WAVE];
   for (int t = threadIdx.x; t < L * NS; t += HIPK_BLOCK) {
      const int hf = t / NS, q = t % NS;
      sdst[hf][q] = jb.dst[hf][q];
      if (q >= NPH) sth[hf][q - NPH] = jb.col[hf][q] >= 0 ? theta[jb.col[hf][q]] : 0.0;
   }
   const int part = threadIdx.x & (L - 1);
   const int64_t stride = (int64_t)gridDim.x * ROWS;
   const int64_t mpad = (m + stride - 1) / stride * stride;
   double n2[NRH];
#pragma unroll
   for (int q = 0; q < NRH; q++) n2[q] = 0.0;
   const bool once = k <= JT;
   if (once) {
      for (int t = threadIdx.x; t < k * L * NS; t += HIPK_BLOCK) {
         const int jj = t / (L * NS), r = t % (L * NS), hf = r / NS, q = r % NS;
         sh[(jj * L + hf) * NJS + q] = (jb.col[hf][q] >= 0) ? h[jj + (size_t)jb.col[hf][q] * ldh] : zacc{0.0, 0.0};
      }
   }
   __syncthreads();
   for (int64_t i = (int64_t)blockIdx.x * ROWS + (threadIdx.x / L); i < mpad; i += stride) {
      const bool live = i < m;
      const int64_t ic = live ? i : m - 1;
      zacc acc[NS];
#pragma unroll
      for (int q = 0; q < NS; q++) acc[q] = {0.0, 0.0};
      for (int j0 = 0; j0 < k; j0 += JT) {
         const int jn = min(JT, k - j0);
         if (!once) {
            __syncthreads();
            for (int t = threadIdx.x; t < jn * L * NS; t += HIPK_BLOCK) {
               const int jj = t / (L * NS), r = t % (L * NS), hf = r / NS, q = r % NS;
               sh[(jj * L + hf) * NJS + q] = (jb.col[hf][q] >= 0) ? h[(j0 + jj) + (size_t)jb.col[hf][q] * ldh] : zacc{0.0, 0.0};
            }
            __syncthreads();
         }
#pragma unroll 1
         for (int jj = 0; jj < jn; jj++) {
            const zacc v = zload_s(V + (size_t)(j0 + jj) * ld + ic);
            const zacc w = W ? zload_s(W + (size_t)(j0 + jj) * ld + ic) : zacc{0.0, 0.0};
            const zacc *hrow = sh + (size_t)(jj * L + part) * NJS;
#pragma unroll
            for (int q = 0; q < NVH; q++) zfma(acc[q], v, hrow[q]);
#pragma unroll
            for (int q = 0; q < NWH; q++) zfma(acc[NVH + q], w, hrow[NVH + q]);
#pragma unroll
            for (int q = 0; q < NRH; q++) {
               const double th = sth[part][q];
               const zacc sres = {fma(-th, v.re, w.re), fma(-th, v.im, w.im)};
               zfma(acc[NPH + q], sres, hrow[NPH + q]);
            }
         }
      }
      if (live) {
#pragma unroll
         for (int q = 0; q < NS; q++) {
            cpx<R> st; st.re = (R)acc[q].re; st.im = (R)acc[q].im;
            cpx<R> *dq = (cpx<R> *)sdst[part][q];
            if (dq) dq[i] = st;
            if (q >= NPH) n2[q - NPH] = fma((double)st.re, (double)st.re, fma((double)st.im, (double)st.im, n2[q - NPH]));
         }
      }
   }
   if (partials) {
      const unsigned nb = gridDim.x;
      for (int hf = 0; hf < L; hf++)
#pragma unroll
         for (int q = 0; q < NRH; q++) {
            const int slot = jb.slot[hf][NPH + q];
            const double v = zblock_sum(part == hf ? n2[q] : 0.0, sm);
            if (threadIdx.x == 0 && slot >= 0) partials[(size_t)slot * nb + blockIdx.x] = v;
         }
   }
}

/* fill the slot tables of the L-lane form: part p takes the p-th share of the V-products, of the W-products and of the
 * residual jobs; false when a share does not fit its slots */
template <int L>
static bool zjobs_split(const hipk_job *jobs, int nj, int NVH, int NWH, int NRH, ZJobsL<L> &jl) {
   for (int hf = 0; hf < L; hf++) for (int q = 0; q < ZRITZ2_SLOTS; q++) { jl.dst[hf][q] = NULL; jl.col[hf][q] = -1; jl.slot[hf][q] = -1; }
   int n[3] = {0, 0, 0};
   for (int q = 0; q < nj; q++) n[jobs[q].kind == HIPK_JOB_RES ? 2 : jobs[q].kind == HIPK_JOB_XW ? 1 : 0]++;
   const int cap[3] = {NVH, NWH, NRH}, base[3] = {0, NVH, NVH + NWH};
   int share[3], seen[3] = {0, 0, 0}, at[3][L];
   for (int c = 0; c < 3; c++) {
      share[c] = (n[c] + L - 1) / L;
      if (share[c] > cap[c]) return false;
      for (int hf = 0; hf < L; hf++) at[c][hf] = base[c];
   }
   for (int q = 0; q < nj; q++) {
      const int c = jobs[q].kind == HIPK_JOB_RES ? 2 : jobs[q].kind == HIPK_JOB_XW ? 1 : 0;
      const int hf = seen[c]++ / share[c];
      const int a = at[c][hf]++;
      jl.dst[hf][a] = jobs[q].dst; jl.col[hf][a] = jobs[q].col; jl.slot[hf][a] = (short)jobs[q].slot;
   }
   return true;
}

/* one launch: np plain jobs + nr residual jobs; up to 16 + 16 in the one-lane form, up to 60 + 4 / 48 + 16 with four lanes
 * per row.  Returns 1 when the jobs do not fit one pass, -(1000 + grid) when the multi-lane form ran (its own grid). */
template <typename R>
static int zritz_launch(hipk_ctx *ctx, int64_t m, const void *V, const void *W, int64_t ld, int k, const double *h, int ldh,
      const double *theta, const hipk_job *jobs, int nj, int gx, bool norms, bool allow_split) {
   int np = 0, nr = 0;
   for (int q = 0; q < nj; q++) { if (jobs[q].kind == HIPK_JOB_RES) nr++; else np++; }
   static int nosplit = -1;                       /* HIPK_Z_NO_SPLIT=1: the one-lane-per-row form only (A/B knob) */
   if (nosplit < 0) nosplit = getenv("HIPK_Z_NO_SPLIT") != NULL;
   if ((np > 12 || nr > 4) && !(np == 0 && nr <= 16) && nr <= 16 && !nosplit && allow_split) {
      /* more than 16 outputs: the cheapest multi-lane form the jobs fit (cost ~ lanes x slots) */
      int64_t need2 = (m + HIPK_BLOCK / 2 - 1) / (HIPK_BLOCK / 2), need4 = (m + HIPK_BLOCK / 4 - 1) / (HIPK_BLOCK / 4);
      const int64_t gmax = (int64_t)ctx->num_cu * 8;
      const int g2 = (int)(need2 < 1 ? 1 : need2 < gmax ? need2 : gmax), g4 = (int)(need4 < 1 ? 1 : need4 < gmax ? need4 : gmax);
      int launched = 0;
      /* the partial sums of the norms are indexed by THIS grid: the caller's second stage is told through the return value */
#define ZR2(LV, A, B, C, G) do { ZJobsL<LV> jl; if (!launched && zjobs_split<LV>(jobs, nj, A, B, C, jl)) { \
            hipLaunchKernelGGL((zritz2_kernel<R, LV, A, B, C>), dim3(G), dim3(HIPK_BLOCK), 0, ctx->stream, (const cpx<R> *)V, (const cpx<R> *)W, ld, k, \
                  (const zacc *)h, ldh, theta, jl, m, norms ? ctx->partials : (double *)NULL); launched = G; } } while (0)
      if (nr <= 4) { ZR2(2, 7, 7, 2, g2); ZR2(2, 9, 7, 2, g2); ZR2(2, 12, 10, 2, g2); ZR2(4, 6, 5, 1, g4); ZR2(4, 8, 7, 1, g4); }
      ZR2(2, 4, 4, 8, g2); ZR2(4, 8, 4, 4, g4);
#undef ZR2
      if (launched) {
         HIPK_CHECK(hipGetLastError());
         return -(1000 + launched);                /* launched with its own grid: the caller finalises over that many blocks */
      }
   }
   int NJ, NR;
   if (nr <= 4 && np <= 4) { NJ = 8; NR = 4; }
   else if (nr <= 4 && np <= 12) { NJ = 16; NR = 4; }
   else if (nr <= 4 && np <= 28) { NJ = 32; NR = 4; }
   else if (np == 0 && nr <= 16) { NJ = 16; NR = 16; }
   else if (nr <= 16 && np <= 16) { NJ = 32; NR = 16; }
   else return 1;                                 /* does not fit one pass */
   ZJobs jb;
   for (int q = 0; q < ZRITZ_JOBS; q++) { jb.dst[q] = NULL; jb.col[q] = -1; jb.isw[q] = 0; jb.slot[q] = -1; }
   int ip = 0, ir = NJ - NR;
   for (int q = 0; q < nj; q++) {
      const int at = (jobs[q].kind == HIPK_JOB_RES) ? ir++ : ip++;
      jb.dst[at] = jobs[q].dst; jb.col[at] = jobs[q].col; jb.isw[at] = (jobs[q].kind == HIPK_JOB_XW); jb.slot[at] = (short)jobs[q].slot;
   }
#define ZR(NJV, NRV) hipLaunchKernelGGL((zritz_kernel<R, NJV, NRV>), dim3(gx), dim3(HIPK_BLOCK), 0, ctx->stream, (const cpx<R> *)V, (const cpx<R> *)W, ld, k, \
         (const zacc *)h, ldh, theta, jb, m, norms ? ctx->partials : (double *)NULL)
   if (NJ == 8) ZR(8, 4); else if (NJ == 16 && NR == 4) ZR(16, 4); else if (NJ == 32 && NR == 4) ZR(32, 4);
   else if (NJ == 16) ZR(16, 16); else ZR(32, 16);
#undef ZR
   HIPK_CHECK(hipGetLastError());
   return 0;
}

template <typename R>
static int zritz_t(hipk_ctx *ctx, int64_t m, const void *V, const void *W, int64_t ld, int k, const double *h, int ldh,
      const double *theta, const hipk_job *jobs, int njobs, double *nrm2_dev) {
   if (k <= 0 || njobs <= 0) return 0;
   int nslots = 0, nout = 0;
   for (int q = 0; q < njobs; q++) {
      if (jobs[q].kind == HIPK_JOB_RES && jobs[q].slot + 1 > nslots) nslots = jobs[q].slot + 1;
      if (jobs[q].dst) nout++;
   }
   if (nslots > 0 && !nrm2_dev) return -1;
   const int gx = zgrid(ctx, m, 4);
   if (nslots > 0 && hipk_reserve_partials(ctx, (size_t)nslots * (size_t)ctx->num_cu * 8)) return -2;   /* (the two-lane form runs up to 8 workgroups per CU) */
   const int pslot = hipk_prof_begin(HIPK_PROF_RITZ, ctx->stream, (double)(2 * k + nout) * (double)m * 2.0 * sizeof(R));
   int gfin = gx;                                  /* blocks the partial norms were written by */
   int rc = zritz_launch<R>(ctx, m, V, W, ld, k, h, ldh, theta, jobs, njobs, gx, nslots > 0, true);
   if (rc <= -1000) { gfin = -rc - 1000; rc = 0; }
   if (rc == 1) {
      /* more outputs than one pass holds: every chunk reads the OLD panels, so the chunks write a temporary and the
       * columns move to their (possibly aliasing) destinations afterwards; slots never written are zeroed first */
      rc = 0;
      const size_t colB = (size_t)m * sizeof(cpx<R>);
      char *tmp = NULL;
      if (hipMalloc((void **)&tmp, colB * (size_t)(nout > 0 ? nout : 1)) != hipSuccess) return -2;
      if (nslots > 0) HIPK_CHECK(hipMemsetAsync(ctx->partials, 0, sizeof(double) * (size_t)nslots * gx, ctx->stream));
      std::vector<hipk_job> work(jobs, jobs + njobs);
      int t = 0;
      for (int q = 0; q < njobs; q++) if (work[q].dst) work[q].dst = tmp + colB * (size_t)t++;
      std::vector<hipk_job> chunk;
      int cp = 0, cr = 0;
      for (int q = 0; q <= njobs && !rc; q++) {
         const bool res = q < njobs && work[q].kind == HIPK_JOB_RES;
         if (q == njobs || (res ? cr == 16 : cp == 16)) {
            if (!chunk.empty()) rc = zritz_launch<R>(ctx, m, V, W, ld, k, h, ldh, theta, chunk.data(), (int)chunk.size(), gx, nslots > 0, false);
            chunk.clear(); cp = cr = 0;
         }
         if (q < njobs) { chunk.push_back(work[q]); if (res) cr++; else cp++; }
      }
      for (int q = 0; q < njobs && !rc; q++)
         if (jobs[q].dst && hipMemcpyAsync(jobs[q].dst, work[q].dst, colB, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) rc = -1;
      if (hipStreamSynchronize(ctx->stream) != hipSuccess) rc = -1;
      (void)hipFree(tmp);
   }
   hipk_prof_end(pslot, ctx->stream);
   if (rc) return rc;
   if (nslots > 0) return hipk_finalize_partials_t(ctx, ctx->partials, gfin, nslots, nrm2_dev);
   return 0;
}

/* ============================ column kernels with complex factors =====================================*/
struct ZFac { double re[64], im[64]; };
template <typename R, bool XPAY>
__global__ void __launch_bounds__(HIPK_BLOCK)
zaxpy_kernel(ZFac a, const cpx<R> *__restrict__ X, int64_t ldX, cpx<R> *__restrict__ Y, int64_t ldY, int nx, int64_t m) {
   const int64_t stride = (int64_t)gridDim.x * HIPK_BLOCK;
   for (int c = 0; c < nx; c++) {
      const zacc f = {a.re[c], a.im[c]};
      for (int64_t i = (int64_t)blockIdx.x * HIPK_BLOCK + threadIdx.x; i < m; i += stride) {
         const zacc x = zload(X + (size_t)c * ldX + i);
         zacc y = zload(Y + (size_t)c * ldY + i);
         if (XPAY) { zacc t = x; zfma(t, f, y); y = t; }      /* y = a y + x */
         else zfma(y, f, x);                                  /* y = a x + y */
         zstore(Y + (size_t)c * ldY + i, y);
      }
   }
}
template <typename R>
__global__ void __launch_bounds__(HIPK_BLOCK)
zpair_dots_kernel(const cpx<R> *__restrict__ X, int64_t ldX, const cpx<R> *__restrict__ Y, int64_t ldY, int nx, int64_t m,
      double *__restrict__ partials) {
   __shared__ double sm[HIPK_BLOCK / HIPK_WAVE];
   const int64_t stride = (int64_t)gridDim.x * HIPK_BLOCK;
   const unsigned nb = gridDim.x;
   for (int c = 0; c < nx; c++) {
      zacc acc = {0.0, 0.0};
      for (int64_t i = (int64_t)blockIdx.x * HIPK_BLOCK + threadIdx.x; i < m; i += stride)
         zfma_conj(acc, zload(X + (size_t)c * ldX + i), zload(Y + (size_t)c * ldY + i));
      const double re = zblock_sum(acc.re, sm), im = zblock_sum(acc.im, sm);
      if (threadIdx.x == 0) { partials[(size_t)(2 * c) * nb + blockIdx.x] = re; partials[(size_t)(2 * c + 1) * nb + blockIdx.x] = im; }
   }
}

/* ============================ CSR SpMV / SpMM on complex data ==========================================
 * The row tiles of the real kernels (hipk_sparse.hip): a tile's (value, column) pairs are staged once in LDS with
 * coalesced loads, then one lane per row walks its row in LDS and gathers x for NC columns with independent
 * accumulators; y = A x - shift_c x(:,c) when shifts are given.  A tile that is one long row is reduced by the whole
 * workgroup. */
#define ZTILE_NNZ 2048
struct ZShift { double s[64]; int on; };
template <typename R>
__device__ __forceinline__ zacc zfetch_x(const cpx<R> *__restrict__ x, const cpx<R> *__restrict__ xlo, const cpx<R> *__restrict__ xhi,
      int64_t x0, int64_t xlen, int64_t halo_lo, int64_t g) {
   const int64_t l = g - x0;
   const cpx<R> *p = x + l;
   if (l < 0) p = xlo + (l + halo_lo);
   if (l >= xlen) p = xhi + (l - xlen);
   return zload(p);
}
template <typename R, int NC>
__global__ void __launch_bounds__(HIPK_BLOCK)
zcsr_kernel(const int4 *__restrict__ tileinfo, int ntiles, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ colind,
      const cpx<R> *__restrict__ val, const cpx<R> *__restrict__ x, int64_t ldx, cpx<R> *__restrict__ y, int64_t ldy, int ncols,
      int64_t x0, int64_t xlen, int64_t halo_lo, int64_t halo_hi, const cpx<R> *__restrict__ xlo, const cpx<R> *__restrict__ xhi,
      int64_t ld_lo, int64_t ld_hi, ZShift sh) {
   __shared__ cpx<R> sval[ZTILE_NNZ];
   __shared__ int32_t scol[ZTILE_NNZ];
   __shared__ double red[2 * (HIPK_BLOCK / HIPK_WAVE)];
   const int per = (ntiles + 7) >> 3;
   const int tile = (blockIdx.x & 7) * per + (blockIdx.x >> 3);     /* contiguous tile ranges per XCD */
   if (tile >= ntiles) return;
   const int4 ti = tileinfo[tile];
   const int r0 = ti.x, r1 = ti.y, p0 = ti.z, nz = ti.w - ti.z;
   if (nz <= ZTILE_NNZ) {
      for (int q = threadIdx.x; q < nz; q += HIPK_BLOCK) { sval[q] = val[p0 + q]; scol[q] = colind[p0 + q]; }
      __syncthreads();
      const int r = r0 + threadIdx.x;
      if (r >= r1) return;
      const int sa = rowptr[r] - p0, sb = rowptr[r + 1] - p0;
      for (int c0 = 0; c0 < ncols; c0 += NC) {
         zacc acc[NC];
#pragma unroll
         for (int c = 0; c < NC; c++) acc[c] = {0.0, 0.0};
         for (int q = sa; q < sb; q++) {
            const zacc a = {(double)sval[q].re, (double)sval[q].im};
            const int64_t g = scol[q];
#pragma unroll
            for (int c = 0; c < NC; c++) if (c0 + c < ncols)
               zfma(acc[c], a, zfetch_x<R>(x + (size_t)(c0 + c) * ldx, xlo ? xlo + (size_t)(c0 + c) * ld_lo : x, xhi ? xhi + (size_t)(c0 + c) * ld_hi : x,
                     x0, xlen, halo_lo, g));
         }
#pragma unroll
         for (int c = 0; c < NC; c++) if (c0 + c < ncols) {
            if (sh.on) { const zacc xo = zload(x + (size_t)(c0 + c) * ldx + r); acc[c].re = fma(-sh.s[c0 + c], xo.re, acc[c].re); acc[c].im = fma(-sh.s[c0 + c], xo.im, acc[c].im); }
            zstore(y + (size_t)(c0 + c) * ldy + r, acc[c]);
         }
      }
   } else {
      for (int c = 0; c < ncols; c++)
         for (int r = r0; r < r1; r++) {
            zacc acc = {0.0, 0.0};
            for (int q = rowptr[r] + threadIdx.x; q < rowptr[r + 1]; q += HIPK_BLOCK)
               zfma(acc, zload(val + q), zfetch_x<R>(x + (size_t)c * ldx, xlo ? xlo + (size_t)c * ld_lo : x, xhi ? xhi + (size_t)c * ld_hi : x,
                     x0, xlen, halo_lo, (int64_t)colind[q]));
            const double re = hipk_wave_sum(acc.re), im = hipk_wave_sum(acc.im);
            __syncthreads();
            if ((threadIdx.x & 63) == 0) { red[2 * (threadIdx.x >> 6)] = re; red[2 * (threadIdx.x >> 6) + 1] = im; }
            __syncthreads();
            if (threadIdx.x == 0) {
               zacc t = {(red[0] + red[2]) + (red[4] + red[6]), (red[1] + red[3]) + (red[5] + red[7])};
               if (sh.on) { const zacc xo = zload(x + (size_t)c * ldx + r); t.re = fma(-sh.s[c], xo.re, t.re); t.im = fma(-sh.s[c], xo.im, t.im); }
               zstore(y + (size_t)c * ldy + r, t);
            }
         }
   }
}

template <typename R>
__global__ void __launch_bounds__(HIPK_BLOCK)
zjacobi_kernel(const cpx<R> *__restrict__ diag, ZShift sh, double min_den, const cpx<R> *__restrict__ x, int64_t ldx,
      cpx<R> *__restrict__ y, int64_t ldy, int ncols, int64_t m) {
   const int64_t stride = (int64_t)gridDim.x * HIPK_BLOCK;
   for (int c = 0; c < ncols; c++)
      for (int64_t i = (int64_t)blockIdx.x * HIPK_BLOCK + threadIdx.x; i < m; i += stride) {
         double d = (double)diag[i].re - sh.s[c];
         if (!(fabs(d) > min_den)) d = copysign(min_den, d);
         const zacc xv = zload(x + (size_t)c * ldx + i);
         zstore(y + (size_t)c * ldy + i, zacc{xv.re / d, xv.im / d});
      }
}

/* ============================ entry points used by the dispatchers of the real files ================== */
int hipk_z_panel_dots(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const hipk_seg *segs, int nseg, const void *X, int64_t ldX, int nx,
      double *out_dev, int ldout) {
   ZSegs sa;
   if (zpack(segs, nseg, &sa)) return -1;
   if (sa.total == 0 || nx <= 0) return 0;
   if (ldout < sa.total) return -1;
   return dt == HIPK_C64 ? zpanel_dots_t<double>(ctx, m, sa, X, ldX, nx, out_dev, ldout) : zpanel_dots_t<float>(ctx, m, sa, X, ldX, nx, out_dev, ldout);
}
int hipk_z_panel_project(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const hipk_seg *segs, int nseg, const double *coef, int ldcoef,
      const double *M, const void *X, int64_t ldX, void *Xout, int64_t ldXout, int nx, double *nrm2_dev) {
   ZSegs sa;
   if (zpack(segs, nseg, &sa)) return -1;
   if (nx <= 0) return 0;
   return dt == HIPK_C64 ? zproject_t<double>(ctx, m, sa, coef, ldcoef, M, X, ldX, Xout, ldXout, nx, nrm2_dev)
                         : zproject_t<float>(ctx, m, sa, coef, ldcoef, M, X, ldX, Xout, ldXout, nx, nrm2_dev);
}
int hipk_z_ritz_update(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const void *V, const void *W, int64_t ld, int k, const double *h, int ldh,
      const double *theta, const hipk_job *jobs, int njobs, double *nrm2_dev) {
   return dt == HIPK_C64 ? zritz_t<double>(ctx, m, V, W, ld, k, h, ldh, theta, jobs, njobs, nrm2_dev)
                         : zritz_t<float>(ctx, m, V, W, ld, k, h, ldh, theta, jobs, njobs, nrm2_dev);
}
int hipk_z_axpy(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const double *alpha_host, const void *X, int64_t ldX, void *Y, int64_t ldY, int nx, int xpay) {
   hipk_prof_scope ps_(HIPK_PROF_VEC, ctx->stream, (double)m * (dt == HIPK_C64 ? 16.0 : 8.0) * 3.0 * nx);
   for (int c0 = 0; c0 < nx; c0 += 64) {
      const int n = nx - c0 < 64 ? nx - c0 : 64;
      ZFac a;
      for (int c = 0; c < 64; c++) { a.re[c] = c < n ? alpha_host[2 * (c0 + c)] : 0.0; a.im[c] = c < n ? alpha_host[2 * (c0 + c) + 1] : 0.0; }
      const int gx = zgrid(ctx, m, 8);
      const size_t es = dt == HIPK_C64 ? 16 : 8;
      const char *xp = (const char *)X + (size_t)c0 * ldX * es; char *yp = (char *)Y + (size_t)c0 * ldY * es;
      if (dt == HIPK_C64) { if (xpay) hipLaunchKernelGGL((zaxpy_kernel<double, true>), dim3(gx), dim3(HIPK_BLOCK), 0, ctx->stream, a, (const cpx<double> *)xp, ldX, (cpx<double> *)yp, ldY, n, m);
                            else hipLaunchKernelGGL((zaxpy_kernel<double, false>), dim3(gx), dim3(HIPK_BLOCK), 0, ctx->stream, a, (const cpx<double> *)xp, ldX, (cpx<double> *)yp, ldY, n, m); }
      else { if (xpay) hipLaunchKernelGGL((zaxpy_kernel<float, true>), dim3(gx), dim3(HIPK_BLOCK), 0, ctx->stream, a, (const cpx<float> *)xp, ldX, (cpx<float> *)yp, ldY, n, m);
             else hipLaunchKernelGGL((zaxpy_kernel<float, false>), dim3(gx), dim3(HIPK_BLOCK), 0, ctx->stream, a, (const cpx<float> *)xp, ldX, (cpx<float> *)yp, ldY, n, m); }
   }
   HIPK_CHECK(hipGetLastError());
   return 0;
}
int hipk_z_pair_dots(hipk_ctx *ctx, hipk_dtype dt, int64_t m, const void *X, int64_t ldX, const void *Y, int64_t ldY, int nx, double *out_dev) {
   if (nx <= 0) return 0;
   hipk_prof_scope ps_(HIPK_PROF_VEC, ctx->stream, (double)m * (dt == HIPK_C64 ? 16.0 : 8.0) * 2.0 * nx);
   const int gx = zgrid(ctx, m, 4);
   if (hipk_reserve_partials(ctx, (size_t)2 * nx * gx)) return -2;
   if (dt == HIPK_C64) hipLaunchKernelGGL(zpair_dots_kernel<double>, dim3(gx), dim3(HIPK_BLOCK), 0, ctx->stream, (const cpx<double> *)X, ldX, (const cpx<double> *)Y, ldY, nx, m, ctx->partials);
   else hipLaunchKernelGGL(zpair_dots_kernel<float>, dim3(gx), dim3(HIPK_BLOCK), 0, ctx->stream, (const cpx<float> *)X, ldX, (const cpx<float> *)Y, ldY, nx, m, ctx->partials);
   HIPK_CHECK(hipGetLastError());
   return hipk_finalize_partials_t(ctx, ctx->partials, gx, 2 * nx, out_dev);
}
int hipk_z_csr_matvec(hipk_dtype dt, hipStream_t st, const int4 *tileinfo, int ntiles, const int32_t *rowptr, const int32_t *colind, const void *val,
      const void *x, int64_t ldx, void *y, int64_t ldy, int ncols, int64_t x0, int64_t xlen, int64_t halo_lo, int64_t halo_hi, const void *xlo,
      const void *xhi, int64_t ld_lo, int64_t ld_hi, const double *shift_host) {
   if (ncols > 64 && shift_host) return 1;
   ZShift sh;
   sh.on = shift_host != NULL;
   for (int c = 0; c < 64; c++) sh.s[c] = (shift_host && c < ncols) ? shift_host[c] : 0.0;
   const int gx = ((ntiles + 7) / 8) * 8;
#define ZC(RT, NCV) hipLaunchKernelGGL((zcsr_kernel<RT, NCV>), dim3(gx), dim3(HIPK_BLOCK), 0, st, tileinfo, ntiles, rowptr, colind, (const cpx<RT> *)val, \
         (const cpx<RT> *)x, ldx, (cpx<RT> *)y, ldy, ncols, x0, xlen, halo_lo, halo_hi, (const cpx<RT> *)xlo, (const cpx<RT> *)xhi, ld_lo, ld_hi, sh)
   if (dt == HIPK_C64) { if (ncols == 1) ZC(double, 1); else if (ncols == 2) ZC(double, 2); else ZC(double, 4); }
   else { if (ncols == 1) ZC(float, 1); else if (ncols == 2) ZC(float, 2); else ZC(float, 4); }
#undef ZC
   HIPK_CHECK(hipGetLastError());
   return 0;
}
int hipk_z_jacobi(hipStream_t st, int num_cu, hipk_dtype dt, int64_t m, const void *diag, const double *shift_host, double min_den, const void *x,
      int64_t ldx, void *y, int64_t ldy, int ncols) {
   ZShift sh;
   sh.on = 1;
   for (int c = 0; c < 64; c++) sh.s[c] = (shift_host && c < ncols) ? shift_host[c] : 0.0;
   int64_t need = (m + HIPK_BLOCK - 1) / HIPK_BLOCK;
   const int gx = (int)(need < 1 ? 1 : (need < (int64_t)num_cu * 8 ? need : (int64_t)num_cu * 8));
   if (dt == HIPK_C64) hipLaunchKernelGGL(zjacobi_kernel<double>, dim3(gx), dim3(HIPK_BLOCK), 0, st, (const cpx<double> *)diag, sh, min_den, (const cpx<double> *)x, ldx, (cpx<double> *)y, ldy, ncols, m);
   else hipLaunchKernelGGL(zjacobi_kernel<float>, dim3(gx), dim3(HIPK_BLOCK), 0, st, (const cpx<float> *)diag, sh, min_den, (const cpx<float> *)x, ldx, (cpx<float> *)y, ldy, ncols, m);
   HIPK_CHECK(hipGetLastError());
   return 0;
}
