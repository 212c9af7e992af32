/* mailbox_protocol_sim.c — TEST INFRASTRUCTURE: the one-shot reduction protocol of primme_amd/csrc/comm_ipc.hip /
 * hipk_internal.h (hipk_xr_exchange, hipk_xr_next_seq) restated with C11 atomics, one THREAD per rank, so that its
 * hand-off logic — 8-byte {tag, half} granules, generation = tag & 1 with two slots per (source, element), a shared
 * sequence counter that skips 0 and keeps the generations alternating at the 2^32 wrap-around, rank-ordered sums — can
 * be exercised on a box without a GPU, with ranks that run far ahead of or behind each other.
 *
 *   mailbox_protocol_sim <ranks> <reductions> <count> <first_seq> <seed>
 * exit code 0: every rank obtained, for every reduction and element, exactly the rank-ordered sum of all ranks' inputs
 * (identical bits on all ranks) and nobody waited longer than the time limit.
 */
#include <pthread.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define MAXR 16
static int P, NRED, COUNT;
static uint32_t FIRST;
static _Atomic uint64_t *mbox[MAXR];          /* [2 generations][P sources][COUNT][2 granules] per rank */
static double *results[MAXR];                 /* [NRED][COUNT] per rank */
static _Atomic int failed;

static double input(int rank, int red, int e) {           /* something whose sum depends on the order of addition */
   uint64_t h = (uint64_t)(rank + 1) * 0x9E3779B97F4A7C15ull ^ (uint64_t)(red + 1) * 0xC2B2AE3D27D4EB4Full ^ (uint64_t)(e + 1) * 0x165667B19E3779F9ull;
   h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
   return ((double)(h >> 11) / 9007199254740992.0 - 0.5) * ((h & 7) ? 1.0 : 1.0e8);
}
static uint32_t next_seq(uint32_t *seq) {                 /* hipk_xr_next_seq */
   uint32_t q = ++*seq;
   if (q == 0) { *seq = 2; q = 2; }
   return q;
}
static void jitter(unsigned *state, int heavy) {
   *state = *state * 1103515245u + 12345u;
   const unsigned r = (*state >> 16) & 1023;
   if (heavy && r < 8) { struct timespec ts = {0, (long)(r + 1) * 200000L}; nanosleep(&ts, NULL); }
   else if (r < 64) sched_yield();
}
extern int sched_yield(void);

static void *rank_main(void *arg) {
   const int me = (int)(intptr_t)arg;
   uint32_t seq = FIRST;                                   /* every rank advances its own copy identically */
   unsigned st = 12345u + 977u * (unsigned)me;
   for (int red = 0; red < NRED && !atomic_load(&failed); red++) {
      const uint32_t q = next_seq(&seq);
      const size_t gen = (size_t)(q & 1u) * P;
      jitter(&st, 1);
      /* push: my value into everybody's mailbox, slot [gen][me] */
      for (int e = 0; e < COUNT; e++) {
         uint64_t bits; const double v = input(me, red, e); memcpy(&bits, &v, 8);
         const uint64_t tag = (uint64_t)q << 32;
         for (int p = 0; p < P; p++) {
            _Atomic uint64_t *dst = mbox[p] + ((gen + me) * COUNT + e) * 2;
            atomic_store_explicit(dst, tag | (bits & 0xffffffffull), memory_order_relaxed);
            atomic_store_explicit(dst + 1, tag | (bits >> 32), memory_order_relaxed);
         }
         if ((e & 7) == 0) jitter(&st, 0);
      }
      /* gather: poll my own mailbox, add in rank order */
      for (int e = 0; e < COUNT; e++) {
         double acc = 0.0;
         for (int p = 0; p < P; p++) {
            const _Atomic uint64_t *src = mbox[me] + ((gen + p) * COUNT + e) * 2;
            uint64_t g0, g1; long spins = 0;
            for (;;) {
               g0 = atomic_load_explicit(src, memory_order_relaxed);
               g1 = atomic_load_explicit(src + 1, memory_order_relaxed);
               if ((uint32_t)(g0 >> 32) == q && (uint32_t)(g1 >> 32) == q) break;
               if (++spins > 400000000L) { fprintf(stderr, "rank %d: reduction %d (tag %u) element %d: no granule from rank %d\n", me, red, q, e, p); atomic_store(&failed, 1); return NULL; }
               if ((spins & 255) == 0) sched_yield();
            }
            const uint64_t bits = (g1 << 32) | (g0 & 0xffffffffull);
            double v; memcpy(&v, &bits, 8);
            acc += v;
         }
         results[me][(size_t)red * COUNT + e] = acc;
      }
   }
   return NULL;
}

int main(int argc, char **argv) {
   if (argc < 6) { fprintf(stderr, "usage: %s ranks reductions count first_seq seed\n", argv[0]); return 2; }
   P = atoi(argv[1]); NRED = atoi(argv[2]); COUNT = atoi(argv[3]); FIRST = (uint32_t)strtoul(argv[4], NULL, 0);
   if (P < 1 || P > MAXR) return 2;
   pthread_t th[MAXR];
   for (int p = 0; p < P; p++) {
      mbox[p] = (_Atomic uint64_t *)calloc((size_t)2 * P * COUNT * 2, sizeof(uint64_t));
      results[p] = (double *)calloc((size_t)NRED * COUNT, sizeof(double));
   }
   for (int p = 0; p < P; p++) pthread_create(&th[p], NULL, rank_main, (void *)(intptr_t)p);
   for (int p = 0; p < P; p++) pthread_join(th[p], NULL);
   if (atomic_load(&failed)) return 1;
   long bad = 0;
   for (int red = 0; red < NRED; red++)
      for (int e = 0; e < COUNT; e++) {
         double want = 0.0;
         for (int p = 0; p < P; p++) want += input(p, red, e);
         for (int p = 0; p < P; p++)
            if (memcmp(&results[p][(size_t)red * COUNT + e], &want, 8)) bad++;
      }
   printf("ranks %d reductions %d count %d first tag %u: %ld mismatches\n", P, NRED, COUNT, FIRST + 1, bad);
   return bad ? 1 : 0;
}
