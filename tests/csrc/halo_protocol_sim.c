/* halo_protocol_sim.c — TEST INFRASTRUCTURE: the neighbour exchange of primme_amd/csrc/comm_ipc.hip (xr_halo_kernel,
 * pa_ipc_halo) restated with C11 atomics, one THREAD per rank of a chain: a rank writes its edge rows into the landing
 * zone [generation = seq & 1][side] of each neighbour, raises that neighbour's flag to seq (release), waits until its own
 * two flags are >= seq, and then READS what landed (the product's SpMV) while faster neighbours are already pushing the
 * next exchange.  exit code 0: every word a rank consumed was the one its neighbour wrote for THAT exchange.
 *   halo_protocol_sim <ranks> <exchanges> <rows> */
#include <pthread.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#define MAXR 16
static int P, NEX, ROWS;
static uint64_t *zone[MAXR];                   /* [2 generations][2 sides][ROWS] per rank, written by the neighbours */
static _Atomic uint64_t flag[MAXR][2];         /* [rank][side]: side 0 written by rank-1, side 1 by rank+1 */
static _Atomic long bad;
extern int sched_yield(void);

static uint64_t word(int src, uint64_t seq, int row) { return ((uint64_t)(src + 1) << 56) ^ (seq << 20) ^ (uint64_t)row; }
static void nap(unsigned *st, unsigned every) {
   *st = *st * 1103515245u + 12345u;
   if (((*st >> 16) % every) == 0) { struct timespec ts = {0, 300000L}; nanosleep(&ts, NULL); } else sched_yield();
}
static void *rank_main(void *arg) {
   const int me = (int)(intptr_t)arg;
   unsigned st = 99u + 31u * (unsigned)me;
   for (uint64_t seq = 1; seq <= (uint64_t)NEX; seq++) {
      const int gen = (int)(seq & 1);
      nap(&st, 5 + me);
      if (me > 0) {                             /* my first rows: rank-1 receives them from above = its side 1 */
         uint64_t *dst = zone[me - 1] + ((size_t)gen * 2 + 1) * ROWS;
         for (int r = 0; r < ROWS; r++) dst[r] = word(me, seq, r);
         atomic_store_explicit(&flag[me - 1][1], seq, memory_order_release);
      }
      if (me < P - 1) {
         uint64_t *dst = zone[me + 1] + ((size_t)gen * 2 + 0) * ROWS;
         for (int r = 0; r < ROWS; r++) dst[r] = word(me, seq, r);
         atomic_store_explicit(&flag[me + 1][0], seq, memory_order_release);
      }
      long spins = 0;
      for (;;) {
         const int lo_ok = me == 0 || atomic_load_explicit(&flag[me][0], memory_order_acquire) >= seq;
         const int hi_ok = me == P - 1 || atomic_load_explicit(&flag[me][1], memory_order_acquire) >= seq;
         if (lo_ok && hi_ok) break;
         if (++spins > 2000000000L) { fprintf(stderr, "rank %d stuck at exchange %llu\n", me, (unsigned long long)seq); atomic_fetch_add(&bad, 1); return NULL; }
         if ((spins & 63) == 0) sched_yield();
      }
      nap(&st, 3);                              /* the consumer runs a while after the exchange */
      if (me > 0) { const uint64_t *z = zone[me] + ((size_t)gen * 2 + 0) * ROWS; for (int r = 0; r < ROWS; r++) if (z[r] != word(me - 1, seq, r)) atomic_fetch_add(&bad, 1); }
      if (me < P - 1) { const uint64_t *z = zone[me] + ((size_t)gen * 2 + 1) * ROWS; for (int r = 0; r < ROWS; r++) if (z[r] != word(me + 1, seq, r)) atomic_fetch_add(&bad, 1); }
   }
   return NULL;
}
int main(int argc, char **argv) {
   if (argc < 4) return 2;
   P = atoi(argv[1]); NEX = atoi(argv[2]); ROWS = atoi(argv[3]);
   if (P < 1 || P > MAXR) return 2;
   pthread_t th[MAXR];
   for (int p = 0; p < P; p++) zone[p] = (uint64_t *)calloc((size_t)4 * ROWS, sizeof(uint64_t));
   for (int p = 0; p < P; p++) pthread_create(&th[p], NULL, rank_main, (void *)(intptr_t)p);
   for (int p = 0; p < P; p++) pthread_join(th[p], NULL);
   printf("ranks %d exchanges %d rows %d: %ld bad words\n", P, NEX, ROWS, atomic_load(&bad));
   return atomic_load(&bad) ? 1 : 0;
}
