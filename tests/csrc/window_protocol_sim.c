/* window_protocol_sim.c — TEST INFRASTRUCTURE: the bulk all-gather of primme_amd/csrc/comm_ipc.hip (pa_ipc_allgather_cols:
 * xr_push_kernel, xr_barrier_kernel, xr_unpack_kernel) restated with C11 atomics, one THREAD per rank: every rank writes its
 * slab into slot [generation = op & 1][me] of EVERY rank's window, passes ONE barrier (a word per rank, monotone), and then
 * reads its own window — while faster ranks already push the next operation into the other generation.
 * exit code 0: every word read belonged to the operation it was read for.
 *   window_protocol_sim <ranks> <operations> <words> */
#include <pthread.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#define MAXR 16
static int P, NOP, WORDS;
static uint64_t *win[MAXR];                    /* [2 generations][P sources][WORDS] per rank */
static _Atomic uint64_t bar[MAXR][MAXR];       /* bar[rank][src]: src's barrier counter as seen by rank */
static _Atomic long bad;
extern int sched_yield(void);
static uint64_t word(int src, uint64_t op, int i) { return ((uint64_t)(src + 1) << 56) ^ (op << 24) ^ (uint64_t)i; }
static void nap(unsigned *st, unsigned every) {
   *st = *st * 1103515245u + 12345u;
   if (((*st >> 16) % every) == 0) { struct timespec ts = {0, 200000L}; nanosleep(&ts, NULL); } else sched_yield();
}
static void *rank_main(void *arg) {
   const int me = (int)(intptr_t)arg;
   unsigned st = 7u + 131u * (unsigned)me;
   for (uint64_t op = 1; op <= (uint64_t)NOP; op++) {
      const size_t gen = (size_t)(op & 1) * P;
      nap(&st, 4 + me);
      for (int p = 0; p < P; p++) {
         uint64_t *dst = win[p] + (gen + me) * WORDS;
         for (int i = 0; i < WORDS; i++) dst[i] = word(me, op, i);
      }
      for (int p = 0; p < P; p++) atomic_store_explicit(&bar[p][me], op, memory_order_release);
      long spins = 0;
      for (int p = 0; p < P; p++)
         while (atomic_load_explicit(&bar[me][p], memory_order_acquire) < op) {
            if (++spins > 2000000000L) { atomic_fetch_add(&bad, 1); return NULL; }
            if ((spins & 63) == 0) sched_yield();
         }
      nap(&st, 3);
      for (int p = 0; p < P; p++) {
         const uint64_t *src = win[me] + (gen + p) * WORDS;
         for (int i = 0; i < WORDS; i++) if (src[i] != word(p, op, i)) atomic_fetch_add(&bad, 1);
      }
   }
   return NULL;
}
int main(int argc, char **argv) {
   if (argc < 4) return 2;
   P = atoi(argv[1]); NOP = atoi(argv[2]); WORDS = atoi(argv[3]);
   if (P < 1 || P > MAXR) return 2;
   pthread_t th[MAXR];
   for (int p = 0; p < P; p++) win[p] = (uint64_t *)calloc((size_t)2 * P * WORDS, sizeof(uint64_t));
   for (int p = 0; p < P; p++) pthread_create(&th[p], NULL, rank_main, (void *)(intptr_t)p);
   for (int p = 0; p < P; p++) pthread_join(th[p], NULL);
   printf("ranks %d operations %d words %d: %ld bad words\n", P, NOP, WORDS, atomic_load(&bad));
   return atomic_load(&bad) ? 1 : 0;
}
