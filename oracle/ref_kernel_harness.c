/* oracle/ref_kernel_harness.c — TEST INFRASTRUCTURE (fixture F6 of SURVEY.md §8(c)).
 * Calls the REFERENCE's own panel routines on seeded inputs and prints inputs and outputs as JSON:
 *    update_projection_dprimme        (src/eigs/update_projection.c:80-165)
 *    Num_update_VWXR_dprimme          (src/eigs/auxiliary_eigs_normal.c:155-388)
 *    Bortho_gen_dprimme               (src/eigs/ortho.c:123-360, CGS + Daniel's test)
 *    Bortho_block_dprimme             (src/eigs/ortho.c:429-439 -> :497-803, CholQR with tracked Gram)
 * Compiled against the reference's headers WHERE THEY LIE (-I/root/reference/src/include ...) and
 * linked with oracle/_ref/libprimme_ref.so (oracle/Makefile target kernel-fixture); the JSON it
 * prints is committed as tests/golden/reference_kernels.json by tests/golden/make_kernel_golden.py.
 * Include order as in the reference's own test helpers (tests/COMMON/num.h:32-34).
 */
#define USE_DOUBLE
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "common.h"
#include "template.h"
#include "blaslapack.h"
#include "../eigs/auxiliary_eigs.h"
#include "../eigs/auxiliary_eigs_normal.h"
#include "../eigs/update_projection.h"
#include "../eigs/ortho.h"

static unsigned g_seed = 12345u;
static double rnd(void) { g_seed = g_seed * 1664525u + 1013904223u; return ((double)(g_seed >> 8) / 16777216.0) - 0.5; }
static double *panel(PRIMME_INT ld, int n) { double *p = (double *)malloc(sizeof(double) * ld * n); for (PRIMME_INT i = 0; i < ld * n; i++) p[i] = rnd(); return p; }
static void put(const char *name, const double *a, PRIMME_INT m, int n, PRIMME_INT ld, int last) {
   printf("  \"%s\": {\"rows\": %lld, \"cols\": %d, \"data\": [", name, (long long)m, n);
   for (int j = 0; j < n; j++) for (PRIMME_INT i = 0; i < m; i++) printf("%s%.17g", (i || j) ? "," : "", a[i + j * ld]);
   printf("]}%s\n", last ? "" : ",");
}

/* ---- the same four routines at LARGE shapes (round 5): many workgroups, ragged tails, second-stage reductions over thousands of
 * partial sums, two-tile matrix-core panels.  Nothing m-sized is printed: the inputs are closed forms that the test regenerates
 * (hash-uniform numbers, bit-exact in C and numpy; sine-basis columns, orthonormal analytically, equal to an ulp), the m-sized
 * outputs are reduced to three sums per column (plain, hash-weighted, squares) and a strided sample of 64 elements.
 *    ref_kernel_harness wide <m> <k> <b> <L>  */
static double hu(unsigned long long i, unsigned long long j, unsigned salt) {
   const unsigned v = (unsigned)(i * 2654435761ull + j * 2246822519ull + (unsigned long long)salt * 3266489917ull);
   return ((double)(v >> 8) / 16777216.0) - 0.5;
}
static double *hpanel(PRIMME_INT m, PRIMME_INT ld, int n, unsigned salt) {
   double *p = (double *)calloc((size_t)ld * n, sizeof(double));
   for (int j = 0; j < n; j++) for (PRIMME_INT i = 0; i < m; i++) p[i + (size_t)j * ld] = hu((unsigned long long)i, (unsigned long long)j, salt);
   return p;
}
/* columns j0 .. j0+n-1 of the sine basis: orthonormal */
static double *spanel(PRIMME_INT m, PRIMME_INT ld, int j0, int n) {
   double *p = (double *)calloc((size_t)ld * n, sizeof(double));
   const double sc = sqrt(2.0 / (double)(m + 1));
   for (int j = 0; j < n; j++) for (PRIMME_INT i = 0; i < m; i++)
      p[i + (size_t)j * ld] = sc * sin(M_PI * (double)(i + 1) * (double)(j0 + j + 1) / (double)(m + 1));
   return p;
}
static void digest(const char *name, const double *a, PRIMME_INT m, int n, PRIMME_INT ld, int last) {
   const PRIMME_INT step = (m + 63) / 64;
   printf("  \"%s\": {\"rows\": %lld, \"cols\": %d, \"step\": %lld, \"sums\": [", name, (long long)m, n, (long long)step);
   for (int j = 0; j < n; j++) {
      double s0 = 0, s1 = 0, s2 = 0;
      for (PRIMME_INT i = 0; i < m; i++) { const double x = a[i + (size_t)j * ld]; s0 += x; s1 += x * hu((unsigned long long)i, (unsigned long long)j, 77u); s2 += x * x; }
      printf("%s[%.17g,%.17g,%.17g]", j ? "," : "", s0, s1, s2);
   }
   printf("], \"sample\": [");
   for (int j = 0; j < n; j++) { printf("%s[", j ? "," : ""); for (PRIMME_INT i = 0, c = 0; i < m; i += step, c++) printf("%s%.17g", c ? "," : "", a[i + (size_t)j * ld]); printf("]"); }
   printf("]}%s\n", last ? "" : ",");
}
static int wide(PRIMME_INT m, int k, int b, int L) {
   const PRIMME_INT ld = m;
   const int K = k + b + 7, nh = b + 5;
   primme_params primme;
   primme_initialize(&primme);
   primme.n = primme.nLocal = m; primme.numProcs = 1; primme.maxBasisSize = K; primme.maxBlockSize = b;
   primme.orth = primme_orth_implicit_I;
   primme_context ctx = primme_get_context(&primme);
   printf("{\n \"m\": %lld, \"ld\": %lld, \"k\": %d, \"b\": %d, \"L\": %d, \"K\": %d, \"nh\": %d,\n", (long long)m, (long long)ld, k, b, L, K, nh);
   /* update_projection: V, W hash panels (salts 1, 2) */
   double *V = hpanel(m, ld, k + b, 1u), *W = hpanel(m, ld, k + b, 2u);
   double *H = (double *)calloc((size_t)K * K, sizeof(double));
   if (update_projection_dprimme(V, ld, W, ld, H, K, m, k, b, 1, ctx)) return 1;
   printf(" \"update_projection\": {\n"); put("H_new_columns", H + (size_t)k * K, k + b, b, K, 1); printf(" },\n");
   /* Num_update_VWXR: h hash (salt 3) scaled by 1/sqrt(k), theta as in the small fixture */
   double *h = (double *)calloc((size_t)k * nh, 8), *theta = (double *)calloc(nh, 8), *X0 = (double *)calloc((size_t)ld * b, 8), *R = (double *)calloc((size_t)ld * b, 8),
          *X1 = (double *)calloc((size_t)ld * (nh - b), 8), *Wo = (double *)calloc((size_t)ld * (nh - b), 8), *rn = (double *)calloc(nh, 8), *xn = (double *)calloc(nh, 8);
   for (int j = 0; j < nh; j++) { theta[j] = 0.3 + 0.11 * j; for (int i = 0; i < k; i++) h[i + (size_t)j * k] = hu((unsigned long long)i, (unsigned long long)j, 3u) / sqrt((double)k); }
   if (Num_update_VWXR_dprimme(V, W, NULL, m, k, ld, h, nh, k, theta,
            X0, 0, b, ld, X1, b, nh, ld, NULL, 0, 0, 0, Wo, b, nh, ld, R, 0, b, ld, rn,
            NULL, 0, 0, 0, NULL, 0, 0, 0, NULL, 0, 0, 0, NULL, 0, 0, NULL, 0, 0, NULL, 0, 0, xn, 0, b, ctx)) return 2;
   printf(" \"update_VWXR\": {\n");
   digest("X0", X0, m, b, ld, 0); digest("R", R, m, b, ld, 0); put("Rnorms", rn, b, 1, b, 0); put("xnorms", xn, b, 1, b, 0);
   digest("X1", X1, m, nh - b, ld, 0); digest("Wo", Wo, m, nh - b, ld, 1);
   printf(" },\n");
   /* Bortho_gen: V = sine columns 0..k-1, locked = sine columns k..k+L-1, the new column = hash (salt 4) + 0.5 * V(:,0) + 0.25 * locked(:,0) */
   double *B = (double *)calloc((size_t)ld * (k + 1), 8), *Q = spanel(m, ld, k, L > 0 ? L : 1);
   { double *S = spanel(m, ld, 0, k); memcpy(B, S, sizeof(double) * (size_t)ld * k); free(S); }
   for (PRIMME_INT i = 0; i < m; i++) B[i + (size_t)k * ld] = hu((unsigned long long)i, 0ull, 4u) + 0.5 * B[i] + (L > 0 ? 0.25 * Q[i] : 0.0);
   double *RL = (double *)calloc((size_t)(L > 0 ? L : 1) * 2, 8);
   PRIMME_INT iseed[4] = {1, 2, 3, 5};
   int b2out = 0;
   if (Bortho_gen_dprimme(B, ld, NULL, 0, k, k, L > 0 ? Q : NULL, ld, L, L > 0 ? RL : NULL, L, m, NULL, NULL, iseed, &b2out, ctx) || b2out != k + 1) return 5;
   printf(" \"Bortho_gen\": {\n"); digest("new_column_out", B + (size_t)k * ld, m, 1, ld, 0); put("RLocked", RL, L, 1, L > 0 ? L : 1, 1); printf(" },\n");
   /* Bortho_block: block = hash (salt 5) + 0.5 * V(:,c) */
   const int maxRank = L + K;
   double *VB = (double *)calloc((size_t)ld * (k + b), 8);
   memcpy(VB, B, sizeof(double) * (size_t)ld * k);
   for (int c = 0; c < b; c++) for (PRIMME_INT i = 0; i < m; i++) VB[i + (size_t)(k + c) * ld] = hu((unsigned long long)i, (unsigned long long)c, 5u) + 0.5 * B[i + (size_t)(c % k) * ld];
   double *G = (double *)calloc((size_t)maxRank * maxRank, 8), *fG = (double *)calloc((size_t)maxRank * maxRank, 8);
   for (int i = 0; i < L + k; i++) G[i + (size_t)i * maxRank] = fG[i + (size_t)i * maxRank] = 1.0;
   if (Bortho_block_dprimme(VB, ld, G, maxRank, fG, maxRank, NULL, 0, k, k + b - 1, L > 0 ? Q : NULL, ld, L, NULL, 0, NULL, 0, m, maxRank, &b2out, ctx) ||
         b2out != k + b) return 6;
   printf(" \"Bortho_block\": {\n"); digest("block_out", VB + (size_t)k * ld, m, b, ld, 0);
   put("gram_new_columns", G + (size_t)(L + k) * maxRank, L + k + b, b, maxRank, 1);
   printf(" }\n}\n");
   primme_free_context(ctx);
   return 0;
}

int main(int argc, char **argv) {
   if (argc >= 6 && !strcmp(argv[1], "wide")) return wide((PRIMME_INT)atoll(argv[2]), atoi(argv[3]), atoi(argv[4]), atoi(argv[5]));
   /* ld == m: with ldV > nLocal the reference's Bortho_gen leaves the block visibly non-orthogonal (|V'V - I| ~ 0.1
    * at m = 97, ld = 99; the solver always calls it with ldV = ldOPs = nLocal), so the fixture uses the layout the
    * solver uses */
   const PRIMME_INT m = 97, ld = 97;
   const int k = 11, b = 4, L = 3, K = 20;
   primme_params primme;
   primme_initialize(&primme);
   primme.n = primme.nLocal = m; primme.numProcs = 1; primme.maxBasisSize = K; primme.maxBlockSize = b;
   primme.orth = primme_orth_implicit_I;
   primme_context ctx = primme_get_context(&primme);
   printf("{\n \"m\": %lld, \"ld\": %lld, \"k\": %d, \"b\": %d, \"L\": %d, \"K\": %d,\n", (long long)m, (long long)ld, k, b, L, K);

   /* ---- update_projection: H(0:k+b, k:k+b) = V(:,0:k+b)' W(:,k:k+b) ---- */
   double *V = panel(ld, k + b), *W = panel(ld, k + b);
   double *H = (double *)calloc((size_t)K * K, sizeof(double));
   printf(" \"update_projection\": {\n");
   put("V", V, m, k + b, ld, 0); put("W", W, m, k + b, ld, 0);
   if (update_projection_dprimme(V, ld, W, ld, H, K, m, k, b, 1, ctx)) return 1;
   put("H_new_columns", H + (size_t)k * K, k + b, b, K, 1);
   printf(" },\n");

   /* ---- Num_update_VWXR: X = V h(:,0:b), R = W h(:,0:b) - X diag(theta), |R|; restart X1 = V h(:,b:nh), Wo = W h(:,b:nh) ---- */
   const int nh = 9;
   double *h = panel(k, nh), theta[16], *X0 = (double *)calloc(ld * b, 8), *R = (double *)calloc(ld * b, 8),
          *X1 = (double *)calloc(ld * (nh - b), 8), *Wo = (double *)calloc(ld * (nh - b), 8), rn[16], xn[16];
   for (int i = 0; i < nh; i++) theta[i] = 0.3 + 0.11 * i;
   printf(" \"update_VWXR\": {\n");
   put("V", V, m, k, ld, 0); put("W", W, m, k, ld, 0); put("h", h, k, nh, k, 0); put("theta", theta, nh, 1, nh, 0);
   if (Num_update_VWXR_dprimme(V, W, NULL, m, k, ld, h, nh, k, theta,
            X0, 0, b, ld, X1, b, nh, ld, NULL, 0, 0, 0, Wo, b, nh, ld, R, 0, b, ld, rn,
            NULL, 0, 0, 0, NULL, 0, 0, 0, NULL, 0, 0, 0, NULL, 0, 0, NULL, 0, 0, NULL, 0, 0, xn, 0, b, ctx)) return 2;
   put("X0", X0, m, b, ld, 0); put("R", R, m, b, ld, 0); put("Rnorms", rn, b, 1, b, 0); put("xnorms", xn, b, 1, b, 0);
   put("X1", X1, m, nh - b, ld, 0); put("Wo", Wo, m, nh - b, ld, 1);
   printf(" },\n");

   /* ---- Bortho_gen: orthonormalise `locked` (L), then V(:,0:k) against it, then one new column ---- */
   double *Q = panel(ld, L), *B = panel(ld, k + 1), *RL = (double *)calloc((size_t)L * (k + 1), 8);
   PRIMME_INT iseed[4] = {1, 2, 3, 5};
   int b2out = 0;
   if (Bortho_gen_dprimme(Q, ld, NULL, 0, 0, L - 1, NULL, 0, 0, NULL, 0, m, NULL, NULL, iseed, &b2out, ctx) || b2out != L) return 3;
   if (Bortho_gen_dprimme(B, ld, NULL, 0, 0, k - 1, Q, ld, L, NULL, 0, m, NULL, NULL, iseed, &b2out, ctx) || b2out != k) return 4;
   printf(" \"Bortho_gen\": {\n");
   put("locked", Q, m, L, ld, 0); put("V_orthonormal", B, m, k, ld, 0); put("new_column_in", B + (size_t)k * ld, m, 1, ld, 0);
   if (Bortho_gen_dprimme(B, ld, NULL, 0, k, k, Q, ld, L, RL, L, m, NULL, NULL, iseed, &b2out, ctx) || b2out != k + 1) return 5;
   put("new_column_out", B + (size_t)k * ld, m, 1, ld, 0); put("RLocked", RL, L, 1, L, 1);
   printf(" },\n");

   /* ---- Bortho_block: tracked Gram matrix [locked V]'[locked V] + its Cholesky factor, block of b new columns ---- */
   const int maxRank = L + K;
   double *VB = (double *)malloc(sizeof(double) * ld * (k + b)), *Xin = panel(ld, b);
   memcpy(VB, B, sizeof(double) * ld * k); memcpy(VB + (size_t)k * ld, Xin, sizeof(double) * ld * b);
   double *G = (double *)calloc((size_t)maxRank * maxRank, 8), *fG = (double *)calloc((size_t)maxRank * maxRank, 8);
   for (int i = 0; i < L + k; i++) G[i + (size_t)i * maxRank] = fG[i + (size_t)i * maxRank] = 1.0;   /* [locked V] orthonormal */
   printf(" \"Bortho_block\": {\n");
   put("locked", Q, m, L, ld, 0); put("V_orthonormal", B, m, k, ld, 0); put("block_in", Xin, m, b, ld, 0);
   if (Bortho_block_dprimme(VB, ld, G, maxRank, fG, maxRank, NULL, 0, k, k + b - 1, Q, ld, L, NULL, 0, NULL, 0, m, maxRank, &b2out, ctx) ||
         b2out != k + b) return 6;
   put("block_out", VB + (size_t)k * ld, m, b, ld, 0);
   put("gram_new_columns", G + (size_t)(L + k) * maxRank, L + k + b, b, maxRank, 1);
   printf(" }\n}\n");
   primme_free_context(ctx);
   return 0;
}
