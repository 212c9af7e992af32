/* eigs_dense.c — the small dense kernels of the projected problem, on the host.
 *
 * The reference hands these to LAPACK/BLAS (src/linalg/blaslapack.c:1024-1289:
 * xHEEVX / xHEGVX / xPOTRF / xTRSM), which is an external dependency that does
 * not travel.  Everything here works on matrices of order <= maxBasisSize
 * (15..41) + numEvals, so plain loops are the right tool; what matters is
 * latency (one call per outer iteration sits between two GPU phases).
 *
 * Symmetric eigensolver: Householder tridiagonalisation with accumulated
 * transformations followed by implicit-shift QL sweeps (the classic
 * EISPACK tred2/tql2 pair, restated).  Columns are sorted ascending.
 */
#include "eigs_internal.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#if !PA_IS_COMPLEX
/* A: n x n, column-major, leading dim lda, symmetric, UPPER triangle referenced.
 * On exit evals[0..n) ascending and Z (ldz) the orthonormal eigenvectors. */
int pa_sym_eig(int n, const double *A, int lda, double *evals, double *Z, int ldz) {
   if (n <= 0) return 0;
   double *z = (double *)malloc((size_t)n * n * sizeof(double)); /* row-major work z[i*n+j] */
   double *e = (double *)malloc((size_t)n * sizeof(double));
   double *d = evals;
   if (!z || !e) { free(z); free(e); return PRIMME_MALLOC_FAILURE; }
   for (int i = 0; i < n; i++)
      for (int j = 0; j < n; j++) z[i * n + j] = (i <= j) ? A[i + (size_t)j * lda] : A[j + (size_t)i * lda];

   /* ---- Householder reduction, last row first ---- */
   for (int i = n - 1; i >= 1; i--) {
      const int l = i - 1;
      double h = 0.0, scale = 0.0;
      if (l > 0) {
         for (int k = 0; k <= l; k++) scale += fabs(z[i * n + k]);
         if (scale == 0.0) {
            e[i] = z[i * n + l];
         } else {
            for (int k = 0; k <= l; k++) { z[i * n + k] /= scale; h += z[i * n + k] * z[i * n + k]; }
            double f = z[i * n + l];
            double g = (f >= 0.0) ? -sqrt(h) : sqrt(h);
            e[i] = scale * g;
            h -= f * g;
            z[i * n + l] = f - g;
            f = 0.0;
            for (int j = 0; j <= l; j++) {
               z[j * n + i] = z[i * n + j] / h;
               g = 0.0;
               for (int k = 0; k <= j; k++) g += z[j * n + k] * z[i * n + k];
               for (int k = j + 1; k <= l; k++) g += z[k * n + j] * z[i * n + k];
               e[j] = g / h;
               f += e[j] * z[i * n + j];
            }
            const double hh = f / (h + h);
            for (int j = 0; j <= l; j++) {
               f = z[i * n + j];
               e[j] = g = e[j] - hh * f;
               for (int k = 0; k <= j; k++) z[j * n + k] -= f * e[k] + g * z[i * n + k];
            }
         }
      } else {
         e[i] = z[i * n + l];
      }
      d[i] = h;
   }
   d[0] = 0.0;
   e[0] = 0.0;
   for (int i = 0; i < n; i++) {
      const int l = i - 1;
      if (d[i] != 0.0) {
         for (int j = 0; j <= l; j++) {
            double g = 0.0;
            for (int k = 0; k <= l; k++) g += z[i * n + k] * z[k * n + j];
            for (int k = 0; k <= l; k++) z[k * n + j] -= g * z[k * n + i];
         }
      }
      d[i] = z[i * n + i];
      z[i * n + i] = 1.0;
      for (int j = 0; j <= l; j++) z[j * n + i] = z[i * n + j] = 0.0;
   }

   /* ---- implicit QL on the tridiagonal (d, e) ---- */
   for (int i = 1; i < n; i++) e[i - 1] = e[i];
   e[n - 1] = 0.0;
   for (int l = 0; l < n; l++) {
      int iter = 0, m;
      do {
         for (m = l; m < n - 1; m++) {
            const double dd = fabs(d[m]) + fabs(d[m + 1]);
            if (fabs(e[m]) <= DBL_EPSILON * dd) break;
         }
         if (m != l) {
            if (iter++ == 200) { free(z); free(e); return PRIMME_LAPACK_FAILURE; }
            double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
            double r = hypot(g, 1.0);
            g = d[m] - d[l] + e[l] / (g + (g >= 0.0 ? fabs(r) : -fabs(r)));
            double s = 1.0, c = 1.0, p = 0.0;
            int i;
            for (i = m - 1; i >= l; i--) {
               double f = s * e[i];
               const double b = c * e[i];
               e[i + 1] = (r = hypot(f, g));
               if (r == 0.0) {
                  d[i + 1] -= p;
                  e[m] = 0.0;
                  break;
               }
               s = f / r;
               c = g / r;
               g = d[i + 1] - p;
               r = (d[i] - g) * s + 2.0 * c * b;
               d[i + 1] = g + (p = s * r);
               g = c * r - b;
               for (int k = 0; k < n; k++) {
                  f = z[k * n + i + 1];
                  z[k * n + i + 1] = s * z[k * n + i] + c * f;
                  z[k * n + i] = c * z[k * n + i] - s * f;
               }
            }
            if (r == 0.0 && i >= l) continue;
            d[l] -= p;
            e[l] = g;
            e[m] = 0.0;
         }
      } while (m != l);
   }

   /* ---- ascending order (selection sort on columns) and copy out ---- */
   int *ord = (int *)malloc((size_t)n * sizeof(int));
   if (!ord) { free(z); free(e); return PRIMME_MALLOC_FAILURE; }
   for (int i = 0; i < n; i++) ord[i] = i;
   for (int i = 0; i < n - 1; i++) {
      int k = i;
      for (int j = i + 1; j < n; j++)
         if (d[ord[j]] < d[ord[k]]) k = j;
      int t = ord[i]; ord[i] = ord[k]; ord[k] = t;
   }
   for (int j = 0; j < n; j++) {
      e[j] = d[ord[j]];
      for (int i = 0; i < n; i++) Z[i + (size_t)j * ldz] = z[i * n + ord[j]];
   }
   memcpy(d, e, (size_t)n * sizeof(double));
   free(ord);
   free(z);
   free(e);
   return 0;
}

#else  /* PA_IS_COMPLEX */
/* The Hermitian counterpart (the reference calls zheev / zheevx, blaslapack.c:1024-1145): unblocked Householder
 * reduction to a REAL symmetric tridiagonal matrix (reflectors H = I - tau w w^H with w = (1; v), chosen so that the
 * subdiagonal comes out real), the unitary factor formed explicitly, then the same implicit-shift QL sweeps as the
 * real solver with the (real) rotations applied to its complex columns.  A: UPPER triangle referenced. */
int pa_sym_eig(int n, const HS *A, int lda, double *evals, HS *Z, int ldz) {
   if (n <= 0) return 0;
   HS *a = (HS *)malloc((size_t)n * n * sizeof(HS));       /* full Hermitian copy, column-major */
   HS *q = (HS *)malloc((size_t)n * n * sizeof(HS));
   HS *tau = (HS *)malloc((size_t)n * sizeof(HS)), *w = (HS *)malloc((size_t)n * sizeof(HS));
   double *e = (double *)malloc((size_t)n * sizeof(double)), *d = evals;
   if (!a || !q || !tau || !w || !e) { free(a); free(q); free(tau); free(w); free(e); return PRIMME_MALLOC_FAILURE; }
   for (int j = 0; j < n; j++)
      for (int i = 0; i < n; i++) a[i + (size_t)j * n] = (i < j) ? A[i + (size_t)j * lda] : (i == j ? creal(A[i + (size_t)j * lda]) : conj(A[j + (size_t)i * lda]));

   /* ---- reduction, column k annihilated below the subdiagonal ---- */
   for (int k = 0; k < n - 1; k++) {
      const int nn = n - k - 1;                 /* order of the trailing block, length of the reflector */
      HS *x = a + (k + 1) + (size_t)k * n;      /* (alpha; x) = a[k+1:n, k] */
      const HS alpha = x[0];
      double xnorm2 = 0.0;
      for (int i = 1; i < nn; i++) xnorm2 += creal(x[i]) * creal(x[i]) + cimag(x[i]) * cimag(x[i]);
      if (xnorm2 == 0.0 && cimag(alpha) == 0.0) {
         tau[k] = 0.0;
         e[k] = creal(alpha);
      } else {
         double beta = sqrt(creal(alpha) * creal(alpha) + cimag(alpha) * cimag(alpha) + xnorm2);
         if (creal(alpha) > 0.0) beta = -beta;
         tau[k] = (beta - alpha) / beta;
         const HS sc = 1.0 / (alpha - beta);
         for (int i = 1; i < nn; i++) x[i] *= sc;
         x[0] = 1.0;
         e[k] = beta;
         /* trailing block A22 <- H^H A22 H:  p = tau A22 v;  w = p - (tau/2)(p^H v) v;  A22 -= v w^H + w v^H */
         HS *A22 = a + (k + 1) + (size_t)(k + 1) * n;
         for (int i = 0; i < nn; i++) {
            HS t = 0.0;
            for (int j = 0; j < nn; j++) t += A22[i + (size_t)j * n] * x[j];
            w[i] = tau[k] * t;
         }
         HS pv = 0.0;
         for (int i = 0; i < nn; i++) pv += conj(w[i]) * x[i];
         const HS al = -0.5 * tau[k] * pv;
         for (int i = 0; i < nn; i++) w[i] += al * x[i];
         for (int j = 0; j < nn; j++)
            for (int i = 0; i < nn; i++) A22[i + (size_t)j * n] -= x[i] * conj(w[j]) + w[i] * conj(x[j]);
         for (int i = 0; i < nn; i++) A22[i + (size_t)i * n] = creal(A22[i + (size_t)i * n]);
      }
      d[k] = creal(a[k + (size_t)k * n]);
   }
   d[n - 1] = creal(a[(n - 1) + (size_t)(n - 1) * n]);
   e[n - 1] = 0.0;

   /* ---- Q = H(0) H(1) ... H(n-2), accumulated from the last reflector ---- */
   for (int j = 0; j < n; j++) for (int i = 0; i < n; i++) q[i + (size_t)j * n] = (i == j) ? 1.0 : 0.0;
   for (int k = n - 2; k >= 0; k--) {
      if (tau[k] == 0.0) continue;
      const int nn = n - k - 1;
      const HS *v = a + (k + 1) + (size_t)k * n;
      for (int j = k + 1; j < n; j++) {
         HS *qc = q + (k + 1) + (size_t)j * n;
         HS t = 0.0;
         for (int i = 0; i < nn; i++) t += conj(v[i]) * qc[i];
         t *= tau[k];
         for (int i = 0; i < nn; i++) qc[i] -= t * v[i];
      }
   }

   /* ---- implicit QL on (d, e); the rotations act on the columns of q ---- */
   for (int l = 0; l < n; l++) {
      int iter = 0, m;
      do {
         for (m = l; m < n - 1; m++) {
            const double dd = fabs(d[m]) + fabs(d[m + 1]);
            if (fabs(e[m]) <= DBL_EPSILON * dd) break;
         }
         if (m != l) {
            if (iter++ == 200) { free(a); free(q); free(tau); free(w); free(e); return PRIMME_LAPACK_FAILURE; }
            double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
            double r = hypot(g, 1.0);
            g = d[m] - d[l] + e[l] / (g + (g >= 0.0 ? fabs(r) : -fabs(r)));
            double sn = 1.0, c = 1.0, p = 0.0;
            int i;
            for (i = m - 1; i >= l; i--) {
               double f = sn * e[i];
               const double b = c * e[i];
               e[i + 1] = (r = hypot(f, g));
               if (r == 0.0) { d[i + 1] -= p; e[m] = 0.0; break; }
               sn = f / r;
               c = g / r;
               g = d[i + 1] - p;
               r = (d[i] - g) * sn + 2.0 * c * b;
               d[i + 1] = g + (p = sn * r);
               g = c * r - b;
               HS *qi = q + (size_t)i * n, *qi1 = q + (size_t)(i + 1) * n;
               for (int kk = 0; kk < n; kk++) {
                  const HS fz = qi1[kk];
                  qi1[kk] = sn * qi[kk] + c * fz;
                  qi[kk] = c * qi[kk] - sn * fz;
               }
            }
            if (r == 0.0 && i >= l) continue;
            d[l] -= p;
            e[l] = g;
            e[m] = 0.0;
         }
      } while (m != l);
   }

   /* ---- ascending order and copy out ---- */
   int *ord = (int *)malloc((size_t)n * sizeof(int));
   if (!ord) { free(a); free(q); free(tau); free(w); free(e); return PRIMME_MALLOC_FAILURE; }
   for (int i = 0; i < n; i++) ord[i] = i;
   for (int i = 0; i < n - 1; i++) {
      int kk = i;
      for (int j = i + 1; j < n; j++) if (d[ord[j]] < d[ord[kk]]) kk = j;
      int t = ord[i]; ord[i] = ord[kk]; ord[kk] = t;
   }
   for (int j = 0; j < n; j++) {
      e[j] = d[ord[j]];
      memcpy(Z + (size_t)j * ldz, q + (size_t)ord[j] * n, (size_t)n * sizeof(HS));
   }
   memcpy(d, e, (size_t)n * sizeof(double));
   free(ord); free(a); free(q); free(tau); free(w); free(e);
   return 0;
}
#endif

/* Upper Cholesky A = U'U in place (upper triangle of A referenced/overwritten).
 * Returns 0, or j+1 if the leading minor of order j+1 is not positive definite. */
int pa_potrf_upper(int n, HS *A, int lda) {
   for (int j = 0; j < n; j++) {
      double s = HS_RE(A[j + (size_t)j * lda]);
      for (int k = 0; k < j; k++) s -= HS_ABS2(A[k + (size_t)j * lda]);
      if (!(s > 0.0) || !isfinite(s)) return j + 1;
      const double ujj = sqrt(s);
      A[j + (size_t)j * lda] = ujj;
      for (int c = j + 1; c < n; c++) {
         HS t = A[j + (size_t)c * lda];
         for (int k = 0; k < j; k++) t -= HS_CONJ(A[k + (size_t)j * lda]) * A[k + (size_t)c * lda];
         A[j + (size_t)c * lda] = t / ujj;
      }
   }
   return 0;
}

/* B <- U^-T B  (U upper n x n, B n x nb) */
void pa_trsm_left_upper_trans(int n, int nb, const HS *U, int ldu, HS *B, int ldb) {
   for (int c = 0; c < nb; c++) {
      HS *b = B + (size_t)c * ldb;
      for (int i = 0; i < n; i++) {
         HS t = b[i];
         for (int k = 0; k < i; k++) t -= HS_CONJ(U[k + (size_t)i * ldu]) * b[k];
         b[i] = t / HS_CONJ(U[i + (size_t)i * ldu]);
      }
   }
}

/* B <- U^-1 B */
void pa_trsm_left_upper(int n, int nb, const HS *U, int ldu, HS *B, int ldb) {
   for (int c = 0; c < nb; c++) {
      HS *b = B + (size_t)c * ldb;
      for (int i = n - 1; i >= 0; i--) {
         HS t = b[i];
         for (int k = i + 1; k < n; k++) t -= U[i + (size_t)k * ldu] * b[k];
         b[i] = t / U[i + (size_t)i * ldu];
      }
   }
}

/* B <- B U^-1  (B mb x n) */
void pa_trsm_right_upper(int mb, int n, const HS *U, int ldu, HS *B, int ldb) {
   for (int j = 0; j < n; j++) {
      for (int k = 0; k < j; k++) {
         const HS u = U[k + (size_t)j * ldu];
         for (int i = 0; i < mb; i++) B[i + (size_t)j * ldb] -= B[i + (size_t)k * ldb] * u;
      }
      const HS inv = 1.0 / U[j + (size_t)j * ldu];
      for (int i = 0; i < mb; i++) B[i + (size_t)j * ldb] *= inv;
   }
}

/* Generalised symmetric-definite problem H x = lambda G x, upper triangles of
 * H and G referenced; G == NULL means identity.  Eigenvectors G-orthonormal. */
int pa_sym_eig_gen(int n, const HS *H, int ldh, const HS *G, int ldg, double *evals,
      HS *Z, int ldz) {
   if (!G) return pa_sym_eig(n, H, ldh, evals, Z, ldz);
   if (n <= 0) return 0;
   HS *U = (HS *)malloc((size_t)n * n * sizeof(HS));
   HS *C = (HS *)malloc((size_t)n * n * sizeof(HS));
   if (!U || !C) { free(U); free(C); return PRIMME_MALLOC_FAILURE; }
   for (int j = 0; j < n; j++)
      for (int i = 0; i < n; i++) {
         U[i + (size_t)j * n] = (i <= j) ? G[i + (size_t)j * ldg] : 0.0;
         C[i + (size_t)j * n] = (i <= j) ? H[i + (size_t)j * ldh] : HS_CONJ(H[j + (size_t)i * ldh]);
      }
   if (pa_potrf_upper(n, U, n)) { free(U); free(C); return PRIMME_LAPACK_FAILURE; }
   /* C <- U^-T C U^-1 */
   pa_trsm_left_upper_trans(n, n, U, n, C, n);
   pa_trsm_right_upper(n, n, U, n, C, n);
   /* symmetrise the round-off */
   for (int j = 0; j < n; j++)
      for (int i = 0; i < j; i++) {
         HS s = 0.5 * (C[i + (size_t)j * n] + HS_CONJ(C[j + (size_t)i * n]));
         C[i + (size_t)j * n] = s; C[j + (size_t)i * n] = HS_CONJ(s);
      }
   int rc = pa_sym_eig(n, C, n, evals, Z, ldz);
   if (!rc) pa_trsm_left_upper(n, n, U, n, Z, ldz);
   free(U);
   free(C);
   return rc;
}

/* new column i = old column perm[i]  (reference src/linalg/auxiliary.c:716 semantics) */
void pa_permute_cols(HS *A, int mrows, int n, int lda, const int *perm) {
   if (n <= 0 || mrows <= 0) return;
   HS *tmp = (HS *)malloc((size_t)mrows * n * sizeof(HS));
   for (int j = 0; j < n; j++) memcpy(tmp + (size_t)j * mrows, A + (size_t)perm[j] * lda, (size_t)mrows * sizeof(HS));
   for (int j = 0; j < n; j++) memcpy(A + (size_t)j * lda, tmp + (size_t)j * mrows, (size_t)mrows * sizeof(HS));
   free(tmp);
}
#if !PA_IS_COMPLEX      /* type-independent helpers exist once (the real object) */
void pa_permute_reals(double *A, int mrows, int n, int lda, const int *perm) { pa_permute_cols(A, mrows, n, lda, perm); }
void pa_permute_ints(int *a, int n, const int *perm) {
   if (n <= 0) return;
   int *tmp = (int *)malloc((size_t)n * sizeof(int));
   for (int j = 0; j < n; j++) tmp[j] = a[perm[j]];
   memcpy(a, tmp, (size_t)n * sizeof(int));
   free(tmp);
}
#endif

/* R = X' * Hsym * X with Hsym given by its upper triangle (order nh), X nh x nx.
 * (reference src/linalg/auxiliary.c:598-625 compute_submatrix) */
void pa_submatrix(const HS *X, int nx, int ldx, const HS *H, int nh, int ldh, HS *R,
      int ldr) {
   if (nx <= 0 || nh <= 0) return;
   HS *t = (HS *)calloc((size_t)nh * nx, sizeof(HS));
   for (int c = 0; c < nx; c++)
      for (int j = 0; j < nh; j++) {
         const HS xj = X[j + (size_t)c * ldx];
         for (int i = 0; i < nh; i++) {
            const HS hij = (i <= j) ? H[i + (size_t)j * ldh] : HS_CONJ(H[j + (size_t)i * ldh]);
            t[i + (size_t)c * nh] += hij * xj;
         }
      }
   for (int c = 0; c < nx; c++)
      for (int r = 0; r < nx; r++) {
         HS s = 0.0;
         for (int i = 0; i < nh; i++) s += HS_CONJ(X[i + (size_t)r * ldx]) * t[i + (size_t)c * nh];
         R[r + (size_t)c * ldr] = s;
      }
   free(t);
}

#if !PA_IS_COMPLEX
/* LAPACK xLARNV(idist = 2) stream, restated: 48-bit multiplicative congruential
 * generator x <- a*x mod 2^48, a = 33952834046453, uniform(-1,1) = 2*x/2^48 - 1.
 * iseed holds the state as four base-4096 digits (updated on exit).  Verified
 * bit-for-bit against the LAPACK in this image (tests/test_dense_host.py).
 * The reference calls this through Num_larnv_Sprimme (blaslapack.c:938-988). */
void pa_larnv_uniform11(int64_t iseed[4], int64_t n, double *x) {
   const uint64_t A = 33952834046453ULL, MASK = (1ULL << 48) - 1;
   uint64_t s = ((((uint64_t)iseed[0] * 4096 + (uint64_t)iseed[1]) * 4096 + (uint64_t)iseed[2]) * 4096 +
                 (uint64_t)iseed[3]) & MASK;
   for (int64_t i = 0; i < n; i++) {
      s = (s * A) & MASK;
      x[i] = 2.0 * ((double)s / 281474976710656.0) - 1.0;
   }
   iseed[0] = (int64_t)((s >> 36) & 4095);
   iseed[1] = (int64_t)((s >> 24) & 4095);
   iseed[2] = (int64_t)((s >> 12) & 4095);
   iseed[3] = (int64_t)(s & 4095);
}

#endif   /* !PA_IS_COMPLEX */

/* ---- singular value decomposition A = U diag(S) V' of a small square matrix ------------------
 * One-sided Jacobi (Hestenes): columns of W = A V are rotated until mutually orthogonal; S are
 * their norms, sorted descending like xGESVD (which the reference calls, blaslapack.c Num_gesvd).
 * High relative accuracy also for the small singular values the refined extraction looks at.
 * Complex: the pair (p, q) is first brought to a real inner product by the phase of w_p^H w_q, then
 * rotated as in the real case; V' means V^H. */
int pa_svd(const HS *A, int ldA, int n, HS *U, int ldU, double *S, HS *V, int ldV) {
   if (n <= 0) return 0;
   HS *W = (HS *)malloc(sizeof(HS) * (size_t)n * n);
   if (!W) return PRIMME_MALLOC_FAILURE;
   for (int j = 0; j < n; j++)
      for (int i = 0; i < n; i++) { W[i + (size_t)j * n] = A[i + (size_t)j * ldA]; V[i + (size_t)j * ldV] = (i == j) ? 1.0 : 0.0; }
   for (int sweep = 0; sweep < 60; sweep++) {
      int rotated = 0;
      for (int p = 0; p < n - 1; p++)
         for (int q = p + 1; q < n; q++) {
            double a = 0.0, b = 0.0;
            HS gz = 0.0;
            const HS *wp = W + (size_t)p * n, *wq = W + (size_t)q * n;
            for (int i = 0; i < n; i++) { a += HS_ABS2(wp[i]); b += HS_ABS2(wq[i]); gz += HS_CONJ(wp[i]) * wq[i]; }
            const double g = HS_ABS(gz);
            if (g <= PA_EPS * sqrt(a * b) || g == 0.0) continue;
            rotated = 1;
#if PA_IS_COMPLEX
            const HS ph = HS_CONJ(gz) / g;          /* w_q ph has a real, positive inner product with w_p */
            const double gs = g;
#else
            const HS ph = 1.0;
            const double gs = gz;
#endif
            const double zeta = (b - a) / (2.0 * gs);
            const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
            const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
            HS *xp = W + (size_t)p * n, *xq = W + (size_t)q * n;
            for (int i = 0; i < n; i++) { const HS u = xp[i], v = xq[i] * ph; xp[i] = c * u - sn * v; xq[i] = sn * u + c * v; }
            HS *vp = V + (size_t)p * ldV, *vq = V + (size_t)q * ldV;
            for (int i = 0; i < n; i++) { const HS u = vp[i], v = vq[i] * ph; vp[i] = c * u - sn * v; vq[i] = sn * u + c * v; }
         }
      if (!rotated) break;
   }
   int *perm = (int *)malloc(sizeof(int) * (size_t)n);
   double *nrm = (double *)malloc(sizeof(double) * (size_t)n);
   if (!perm || !nrm) { free(W); free(perm); free(nrm); return PRIMME_MALLOC_FAILURE; }
   for (int j = 0; j < n; j++) {
      double t = 0.0;
      for (int i = 0; i < n; i++) t += HS_ABS2(W[i + (size_t)j * n]);
      nrm[j] = sqrt(t); perm[j] = j;
   }
   for (int i = 1; i < n; i++) {             /* insertion sort, descending */
      const int pi = perm[i];
      int j = i - 1;
      while (j >= 0 && nrm[perm[j]] < nrm[pi]) { perm[j + 1] = perm[j]; j--; }
      perm[j + 1] = pi;
   }
   HS *Vc = (HS *)malloc(sizeof(HS) * (size_t)n * n);
   if (!Vc) { free(W); free(perm); free(nrm); return PRIMME_MALLOC_FAILURE; }
   for (int j = 0; j < n; j++) memcpy(Vc + (size_t)j * n, V + (size_t)perm[j] * ldV, sizeof(HS) * (size_t)n);
   for (int j = 0; j < n; j++) {
      const int pj = perm[j];
      S[j] = nrm[pj];
      memcpy(V + (size_t)j * ldV, Vc + (size_t)j * n, sizeof(HS) * (size_t)n);
      for (int i = 0; i < n; i++) U[i + (size_t)j * ldU] = (nrm[pj] > 0.0) ? W[i + (size_t)pj * n] / nrm[pj] : (i == j ? 1.0 : 0.0);
   }
   /* V is a product of thousands of plane rotations: its columns are orthonormal to about 1e-14 only, and the
    * restarted basis V h inherits that with every restart (xGESVD's Householder vectors keep 1e-16).  Two
    * Gram-Schmidt passes bring it back, from the smallest singular value (left as it is, but normalised) upwards;
    * U and S keep describing A V to a backward error of that size. */
   for (int j = n - 1; j >= 0; j--) {
      HS *vj = V + (size_t)j * ldV;
      for (int pass = 0; pass < 2; pass++)
         for (int k = n - 1; k > j; k--) {
            const HS *vk = V + (size_t)k * ldV;
            HS t = 0.0;
            for (int i = 0; i < n; i++) t += HS_CONJ(vk[i]) * vj[i];
            for (int i = 0; i < n; i++) vj[i] -= t * vk[i];
         }
      double t = 0.0;
      for (int i = 0; i < n; i++) t += HS_ABS2(vj[i]);
      t = sqrt(t);
      if (t > 0.0) for (int i = 0; i < n; i++) vj[i] /= t;
   }
   free(W); free(perm); free(nrm); free(Vc);
   return 0;
}

