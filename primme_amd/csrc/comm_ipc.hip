/* comm_ipc.hip — the one-shot peer-to-peer transport of the row-partitioned solve.
 *
 * What it replaces: the MPI_Allreduce behind globalSumReal (reference examples/ex_eigs_mpi.c:209-218) at its
 * call sites in the inner loop — the overlaps and the norm of a Gram-Schmidt pass (src/eigs/ortho.c:249, :290)
 * and the new column of the projected matrix (src/eigs/update_projection.c:136) — and the neighbour exchange
 * of the partitioned matvec (examples/ex_eigs_mpi.c:150-207).  These messages are <= 4 KB: a ring collective
 * costs a launch plus several hops of latency for them, and at 8 ranks that latency, not HBM bandwidth, is
 * what an outer iteration would spend its time on.
 *
 * Here every rank owns a MAILBOX in device memory, exported with hipIpcGetMemHandle and mapped by all peers
 * (other GPUs over xGMI, or other processes on the same GPU).  A reduction is one-shot: every rank stores its
 * partial sums straight into every peer's mailbox as 8-byte {tag, half} granules, polls its own mailbox until
 * the granules of all ranks carry the tag of this reduction, and adds them in rank order — the same order on
 * every rank, so all ranks hold bit-identical sums and solve the identical small projected problem (which is
 * what lets the reference's broadcasts go).  One launch: it can be the second stage of the local two-stage
 * reduction itself (hipk_core.hip: hipk_finalize_kernel with a hipk_xr_dev), which then also publishes the
 * results to the host.  Neighbour halos are pushed the same way (rows written into the neighbour's landing
 * zone, one flag per generation), bulk all-gather / reduce-scatter land in a second exported window.
 *
 * Rendez-vous is a POSIX shared-memory segment named by the communicator id: one node, which is the scope of
 * the partition (BASELINE.json: 8 MI355X on xGMI).
 */
#include "comm_internal.h"
#include <errno.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#define IPC_MAGIC "PAIPC1"
#define IPC_MAXR HIPK_XR_MAXRANKS
#define IPC_SLOT 4096                 /* doubles per (generation, source) slot of the granule area */

struct ipc_shm {
   char magic[8];
   int arrived, generation;          /* sense-reversing barrier of the rendez-vous */
   int failed;                       /* a rank reported a set-up failure: everybody backs out */
   int pad;
   char payload[IPC_MAXR][PA_IPC_PAYLOAD];
};

struct ipc_region {                  /* one exported allocation per rank, mapped by all peers */
   void *mine;
   size_t cap;
   void *peer[IPC_MAXR];             /* peer[rank] == mine */
};

/* layout of the mailbox allocation (bytes):  [granules 2*P*SLOT*16][barrier words P*8][halo flags 2*8][pad] */
struct pa_ipc {
   int rank, nranks;
   ipc_shm *shm;
   char shm_name[64];
   int distinct;                     /* every rank on its own device */
   ipc_region mbox, hz, win;         /* mailbox, halo landing zones, bulk window */
   unsigned long long **tab_dev;     /* device array of the peers' granule areas */
   unsigned long long **bar_dev;     /* device array of the peers' barrier word arrays */
   unsigned int seq;                 /* tag of the last reduction */
   unsigned long long bseq, hseq, wseq;   /* barriers, halo exchanges, window operations issued */
   int *err_host, *err_dev;          /* pinned: the error word (first 64 bytes) and 768 bytes of scratch behind it for the set-up copies */
   unsigned int *ticket;             /* device counter of the multi-block halo launch */
   long long timeout_ticks;
   hipk_xreduce xr;
   int alloc_kind;
   int gpu_ok;                       /* the mailboxes are up; 0: only the host rendez-vous works (RCCL is then bootstrapped through it) */
};

static size_t mbox_gran_bytes(int P) { return (size_t)2 * P * IPC_SLOT * 16; }
static size_t mbox_bytes(int P) { return mbox_gran_bytes(P) + (size_t)IPC_MAXR * 8 + 64; }

/* ---- rendez-vous through the shared segment -------------------------------------------------------- */
static double now_s(void) {
   struct timespec ts;
   clock_gettime(CLOCK_MONOTONIC, &ts);
   return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
static double host_timeout_s(void) {
   const char *e = getenv("PRIMME_AMD_IPC_TIMEOUT_S");
   const double v = e ? atof(e) : 0.0;
   return v > 0.0 ? v : 300.0;
}
/* the FIRST rendez-vous of a communicator: a rank on another node (or in another IPC namespace) cannot see the segment and
 * has left with an error already; the ranks that can see it should not sit here for five minutes */
static double attach_timeout_s(void) {
   const char *e = getenv("PRIMME_AMD_IPC_ATTACH_TIMEOUT_S");
   const double v = e ? atof(e) : 0.0;
   return v > 0.0 ? v : 60.0;
}
static int shm_barrier_lim(pa_ipc *x, double lim, const char *advice);
static int shm_barrier(pa_ipc *x) { return shm_barrier_lim(x, host_timeout_s(), ""); }
static int shm_barrier_lim(pa_ipc *x, double lim, const char *advice) {
   ipc_shm *s = x->shm;
   const int gen = __atomic_load_n(&s->generation, __ATOMIC_ACQUIRE);
   if (__atomic_add_fetch(&s->arrived, 1, __ATOMIC_ACQ_REL) == x->nranks) {
      __atomic_store_n(&s->arrived, 0, __ATOMIC_RELAXED);
      __atomic_store_n(&s->generation, gen + 1, __ATOMIC_RELEASE);
      return 0;
   }
   const double t0 = now_s();
   for (long spins = 0;; spins++) {
      if (__atomic_load_n(&s->generation, __ATOMIC_ACQUIRE) != gen) return 0;
      if ((spins & 1023) == 1023) {
         if (now_s() - t0 > lim) {
            fprintf(stderr, "primme_amd: rank %d waited %.0f s for the other %d rank(s) at a communicator rendez-vous%s\n", x->rank, lim, x->nranks - 1, advice);
            return -43;
         }
         sched_yield();
      }
   }
}
int pa_ipc_host_allgather(pa_ipc *x, const void *mine, size_t bytes, void *all) {
   if (bytes > PA_IPC_PAYLOAD) return -1;
   memcpy(x->shm->payload[x->rank], mine, bytes);
   if (shm_barrier(x)) return -43;
   for (int p = 0; p < x->nranks; p++) memcpy((char *)all + (size_t)p * bytes, x->shm->payload[p], bytes);
   return shm_barrier(x);             /* nobody overwrites a payload before everybody has read it */
}
/* every rank reports ok / not ok; returns non-zero on every rank when any rank failed */
static int shm_agree(pa_ipc *x, int my_failure) {
   int all[IPC_MAXR], bad = 0;
   if (pa_ipc_host_allgather(x, &my_failure, sizeof(int), all)) return -43;
   for (int p = 0; p < x->nranks; p++) bad |= all[p];
   return bad;
}

int pa_ipc_is_ipc_id(const void *id128) { return memcmp(id128, IPC_MAGIC, 6) == 0; }

int pa_ipc_unique_id(void *id128) {
   char name[64];
   for (int attempt = 0; attempt < 16; attempt++) {
      snprintf(name, sizeof(name), "/primme_amd_%d_%lx", (int)getpid(), (unsigned long)(now_s() * 1e6) + attempt);
      const int fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
      if (fd < 0) { if (errno == EEXIST) continue; perror("primme_amd: shm_open"); return -43; }
      if (ftruncate(fd, sizeof(ipc_shm)) != 0) { perror("primme_amd: ftruncate"); close(fd); shm_unlink(name); return -43; }
      ipc_shm *s = (ipc_shm *)mmap(NULL, sizeof(ipc_shm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
      close(fd);
      if (s == MAP_FAILED) { shm_unlink(name); return -43; }
      memset(s, 0, sizeof(*s));
      memcpy(s->magic, IPC_MAGIC, 6);
      munmap(s, sizeof(*s));
      memset(id128, 0, 128);
      memcpy(id128, IPC_MAGIC, 6);
      strncpy((char *)id128 + 8, name, 64);
      return 0;
   }
   return -43;
}

/* ---- exported allocations --------------------------------------------------------------------------- */
static int ipc_alloc(pa_ipc *x, void **p, size_t bytes) {
   /* Memory that peers write and this device polls / reads must not sit in this device's L2: uncached, or fine-grained.
    * Ordinary (coarse-grained) device memory is accepted only on request (PRIMME_AMD_IPC_ALLOC=plain: a bring-up knob for ranks
    * that share one device) — a remote write over xGMI does not invalidate the owner's L2. */
   static int kind = -1;             /* PRIMME_AMD_IPC_ALLOC = uncached (default) | fine | plain */
   if (kind < 0) {
      const char *e = getenv("PRIMME_AMD_IPC_ALLOC");
      kind = !e ? 0 : !strcmp(e, "fine") ? 1 : !strcmp(e, "plain") ? 2 : 0;
   }
   *p = NULL;
   for (int k = kind; k < (kind == 2 ? 3 : 2); k++) {
      hipError_t e = k == 0 ? hipExtMallocWithFlags(p, bytes, hipDeviceMallocUncached)
                   : k == 1 ? hipExtMallocWithFlags(p, bytes, hipDeviceMallocFinegrained) : hipMalloc(p, bytes);
      if (e == hipSuccess && *p) { x->alloc_kind = k; break; }
      (void)hipGetLastError();
      *p = NULL;
   }
   if (!*p) return -2;
   HIPK_CHECK(hipMemset(*p, 0, bytes));
   HIPK_CHECK(hipDeviceSynchronize());
   return 0;
}
static void region_close(pa_ipc *x, ipc_region *r) {
   for (int p = 0; p < x->nranks; p++)
      if (p != x->rank && r->peer[p]) (void)hipIpcCloseMemHandle(r->peer[p]);
   if (r->mine) (void)hipFree(r->mine);
   memset(r, 0, sizeof(*r));
}
/* collective: (re)allocate this rank's piece with `bytes` and map everybody's.  The caller has drained its
 * stream; the rendez-vous inside makes sure every rank has before anything is unmapped. */
static int region_open(pa_ipc *x, ipc_region *r, size_t bytes) {
   if (shm_barrier(x)) return -43;
   region_close(x, r);
   int fail = ipc_alloc(x, &r->mine, bytes) != 0;
   hipIpcMemHandle_t h, all[IPC_MAXR];
   memset(&h, 0, sizeof(h));
   if (!fail && x->nranks > 1 && hipIpcGetMemHandle(&h, r->mine) != hipSuccess) {
      fprintf(stderr, "primme_amd: hipIpcGetMemHandle failed (%s); HSA_ENABLE_IPC_MODE_LEGACY=0 is required on this driver\n",
            hipGetErrorString(hipGetLastError()));
      fail = 1;
   }
   if (pa_ipc_host_allgather(x, &h, sizeof(h), all)) return -43;
   if (shm_agree(x, fail)) { region_close(x, r); return -43; }
   r->cap = bytes;
   for (int p = 0; p < x->nranks && !fail; p++) {
      if (p == x->rank) { r->peer[p] = r->mine; continue; }
      if (hipIpcOpenMemHandle(&r->peer[p], all[p], hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
         fprintf(stderr, "primme_amd: hipIpcOpenMemHandle (rank %d <- %d) failed: %s\n", x->rank, p, hipGetErrorString(hipGetLastError()));
         r->peer[p] = NULL;
         fail = 1;
      }
   }
   if (shm_agree(x, fail)) { region_close(x, r); return -43; }
   return 0;
}
static int region_reserve(pa_ipc *x, hipStream_t st, ipc_region *r, size_t need) {
   if (need <= r->cap) return 0;
   HIPK_CHECK(hipStreamSynchronize(st));
   size_t cap = r->cap ? r->cap : ((size_t)1 << 16);
   while (cap < need) cap *= 2;
   return region_open(x, r, cap);
}

/* ---- kernels ---------------------------------------------------------------------------------------- */
/* dbuf[i] <- sum over ranks, 16 lanes per element (lane p of a group talks to rank p), 16 elements per block */
__global__ void __launch_bounds__(HIPK_BLOCK)
xr_allreduce_kernel(double *__restrict__ buf, double *__restrict__ mirror, int count, hipk_xr_dev x, hipk_fin_flag fin) {
   const int e = (int)blockIdx.x * (HIPK_BLOCK / 16) + (int)(threadIdx.x >> 4);
   const bool active = e < count;
   const double v = active ? buf[e] : 0.0;
   const double s = hipk_xr_exchange(x, active ? (unsigned)e : 0u, v, active);
   if (active && (threadIdx.x & 15) == 0) {
      buf[e] = s;
      if (mirror) mirror[e] = s;
   }
   __syncthreads();
   if (threadIdx.x == 0) hipk_publish_flag(fin, gridDim.x);
}

struct xr_bar_args { unsigned long long **bar; int nranks, rank; unsigned long long seq; int *err; long long timeout_ticks; };
/* device-side barrier over the ranks: everything this rank enqueued before it is visible to the peers that pass it */
__global__ void __launch_bounds__(64) xr_barrier_kernel(xr_bar_args a) {
   const int lane = threadIdx.x;
   __threadfence_system();
   if (lane < a.nranks) {
      __hip_atomic_store(a.bar[lane] + a.rank, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      const unsigned long long *mine = a.bar[a.rank] + lane;
      const long long t0 = wall_clock64();
      while (__hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < a.seq) {
         if (wall_clock64() - t0 > a.timeout_ticks) { *(volatile int *)a.err = 2; break; }
         __builtin_amdgcn_s_sleep(2);
      }
   }
   __threadfence_system();
}

/* words of W bytes: column c of src (ld_src words apart, first word `off`) -> dst + c*cnt, cnt words per column */
template <typename W>
__device__ __forceinline__ void copy_cols(W *__restrict__ dst, const W *__restrict__ src, int64_t ld_src, int64_t off, int64_t cnt,
      int ncols, int64_t start, int64_t stride) {
   const int64_t total = cnt * ncols;
   for (int64_t i = start; i < total; i += stride) {
      const int64_t c = i / cnt, r = i - c * cnt;
      dst[i] = src[c * ld_src + off + r];
   }
}
struct xr_halo_args {
   const void *x; int64_t ldx, nrows; int ncols;
   int64_t send_lo, send_hi;         /* words per column to rank-1 / rank+1 */
   void *dst_lo, *dst_hi;            /* where they land in the neighbours' zones */
   unsigned long long *flag_lo_peer, *flag_hi_peer;   /* the neighbours' flag words for me */
   const unsigned long long *my_flag_lo, *my_flag_hi; /* my flag words: written by rank-1 / rank+1 */
   int wait_lo, wait_hi;
   unsigned long long seq;
   unsigned int *ticket;
   int *err; long long timeout_ticks;
};
template <typename W>
__global__ void __launch_bounds__(HIPK_BLOCK) xr_halo_kernel(xr_halo_args a) {
   __shared__ int s_last;
   const int64_t start = (int64_t)blockIdx.x * HIPK_BLOCK + threadIdx.x, stride = (int64_t)gridDim.x * HIPK_BLOCK;
   if (a.send_lo > 0) copy_cols<W>((W *)a.dst_lo, (const W *)a.x, a.ldx, 0, a.send_lo, a.ncols, start, stride);
   if (a.send_hi > 0) copy_cols<W>((W *)a.dst_hi, (const W *)a.x, a.ldx, a.nrows - a.send_hi, a.send_hi, a.ncols, start, stride);
   __syncthreads();
   if (threadIdx.x == 0) {
      __threadfence_system();
      const unsigned t = atomicAdd(a.ticket, 1u);
      s_last = (t == gridDim.x - 1);
   }
   __syncthreads();
   if (!s_last) return;
   if (threadIdx.x == 0) {
      *a.ticket = 0;
      /* the rows of every block are out: tell the neighbours, then wait for theirs */
      if (a.flag_lo_peer) __hip_atomic_store(a.flag_lo_peer, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (a.flag_hi_peer) __hip_atomic_store(a.flag_hi_peer, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      const long long t0 = wall_clock64();
      for (;;) {
         const bool lo_ok = !a.wait_lo || __hip_atomic_load(a.my_flag_lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= a.seq;
         const bool hi_ok = !a.wait_hi || __hip_atomic_load(a.my_flag_hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= a.seq;
         if (lo_ok && hi_ok) break;
         if (wall_clock64() - t0 > a.timeout_ticks) { *(volatile int *)a.err = 3; break; }
         __builtin_amdgcn_s_sleep(2);
      }
      __threadfence_system();
   }
}

/* bulk pushes into the peers' windows: for every peer p and column c, `cnt` words starting at word
 * off0 + p*off_per_peer of column c of `send` go to peer p's window at ((c*P + me)*cnt) */
struct xr_push_args {
   const void *send; int64_t ld_send; int ncols; int64_t cnt, off0, off_per_peer;
   void *win[IPC_MAXR];               /* the peers' window bases (this generation) */
   int nranks, rank;
};
template <typename W>
__global__ void __launch_bounds__(HIPK_BLOCK) xr_push_kernel(xr_push_args a) {
   const int p = blockIdx.y;
   W *dst = (W *)a.win[p];
   const W *src = (const W *)a.send;
   const int64_t total = a.cnt * a.ncols;
   for (int64_t i = (int64_t)blockIdx.x * HIPK_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * HIPK_BLOCK) {
      const int64_t c = i / a.cnt, r = i - c * a.cnt;
      dst[(c * a.nranks + a.rank) * a.cnt + r] = src[c * a.ld_send + a.off0 + (int64_t)p * a.off_per_peer + r];
   }
}
/* window -> recv: column c = P*cnt contiguous words */
template <typename W>
__global__ void __launch_bounds__(HIPK_BLOCK) xr_unpack_kernel(const W *__restrict__ win, W *__restrict__ recv, int64_t ld_recv, int64_t colwords, int ncols) {
   const int64_t total = colwords * ncols;
   for (int64_t i = (int64_t)blockIdx.x * HIPK_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * HIPK_BLOCK) {
      const int64_t c = i / colwords, r = i - c * colwords;
      recv[c * ld_recv + r] = win[i];
   }
}
/* recv(:,c) = sum over p (rank order) of win[(c*P + p)*cnt + :] */
template <typename T>
__global__ void __launch_bounds__(HIPK_BLOCK) xr_sum_kernel(const T *__restrict__ win, T *__restrict__ recv, int64_t ld_recv, int64_t cnt, int ncols, int P) {
   const int64_t total = cnt * ncols;
   for (int64_t i = (int64_t)blockIdx.x * HIPK_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * HIPK_BLOCK) {
      const int64_t c = i / cnt, r = i - c * cnt;
      T acc = (T)0;
      for (int p = 0; p < P; p++) acc += win[(c * P + p) * cnt + r];
      recv[c * ld_recv + r] = acc;
   }
}

/* ---- set-up ----------------------------------------------------------------------------------------- */
static int upload_tables(pa_ipc *x) {
   unsigned long long *tab[IPC_MAXR] = {0}, *bar[IPC_MAXR] = {0};
   for (int p = 0; p < x->nranks; p++) {
      tab[p] = (unsigned long long *)x->mbox.peer[p];
      bar[p] = (unsigned long long *)((char *)x->mbox.peer[p] + mbox_gran_bytes(x->nranks));
   }
   /* through pinned scratch (no runtime copy out of pageable memory anywhere in the library) */
   char *scratch = (char *)x->err_host + 256;
   memcpy(scratch, tab, sizeof(void *) * IPC_MAXR);
   memcpy(scratch + sizeof(void *) * IPC_MAXR, bar, sizeof(void *) * IPC_MAXR);
   HIPK_CHECK(hipMemcpyAsync(x->tab_dev, scratch, sizeof(void *) * IPC_MAXR * 2, hipMemcpyHostToDevice, NULL));
   HIPK_CHECK(hipStreamSynchronize(NULL));
   return 0;
}

/* the tag sequence of the reductions wraps (hipk_xr_next_seq): every rank gets here at the same reduction.  Drain, meet,
 * clear this rank's granule area (what the peers wrote in the cycle that ends), meet again. */
static int ipc_seq_wrap_impl(pa_ipc *x) {
   if (hipDeviceSynchronize() != hipSuccess) return -1;
   if (shm_barrier(x)) return -43;              /* nobody is inside a reduction of the old cycle any more */
   if (x->mbox.mine && (hipMemset(x->mbox.mine, 0, mbox_gran_bytes(x->nranks)) != hipSuccess || hipDeviceSynchronize() != hipSuccess)) return -1;
   return shm_barrier(x);                       /* nobody writes a tag of the new cycle into an area that is still being cleared */
}
static int ipc_seq_wrap(void *owner) {
   pa_ipc *x = (pa_ipc *)owner;
   const int rc = ipc_seq_wrap_impl(x);
   /* a wrap that did not complete (barrier time limit, HIP error) leaves granule areas that may still carry tags of the old
    * cycle: the transport is unusable from here on — say so through the error word the solver looks at after every reduction
    * (pa_comm_failed -> PRIMME_PARALLEL_FAILURE) instead of restarting the sequence over stale tags */
   if (rc && x->err_host) *(volatile int *)x->err_host = 2;
   return rc;
}

int pa_ipc_attach(pa_ipc **out, const void *id128, int rank, int nranks) {
   *out = NULL;
   if (!pa_ipc_is_ipc_id(id128) || nranks < 1 || nranks > IPC_MAXR || rank < 0 || rank >= nranks) {
      if (nranks > IPC_MAXR) fprintf(stderr, "primme_amd: the peer-to-peer transport serves at most %d ranks\n", IPC_MAXR);
      return -1;
   }
   pa_ipc *x = (pa_ipc *)calloc(1, sizeof(*x));
   if (!x) return -2;
   x->rank = rank; x->nranks = nranks;
   strncpy(x->shm_name, (const char *)id128 + 8, sizeof(x->shm_name) - 1);
   const int fd = shm_open(x->shm_name, O_RDWR, 0600);
   if (fd < 0) {
      fprintf(stderr, "primme_amd: rank %d cannot open the rendez-vous segment %s of the peer-to-peer transport (%s): the mailboxes serve the ranks of ONE "
            "node that share /dev/shm; for a job that spans nodes or containers create the id with PRIMME_AMD_COMM=rccl\n", rank, x->shm_name, strerror(errno));
      free(x);
      return -43;
   }
   x->shm = (ipc_shm *)mmap(NULL, sizeof(ipc_shm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
   close(fd);
   if (x->shm == MAP_FAILED) { free(x); return -43; }
   if (shm_barrier_lim(x, attach_timeout_s(), ": a rank on another node or in another IPC namespace cannot see the segment (it has left with an error); "
         "the mailboxes serve one node — create the id with PRIMME_AMD_COMM=rccl for such a job")) {
      if (rank == 0) shm_unlink(x->shm_name);
      munmap(x->shm, sizeof(ipc_shm)); free(x); return -43;
   }
   if (rank == 0) shm_unlink(x->shm_name);       /* everybody has it mapped: nothing is left behind if a rank dies later */

   const char *te = getenv("PRIMME_AMD_IPC_DEVICE_TIMEOUT_S");
   const double tsec = te && atof(te) > 0.0 ? atof(te) : 60.0;
   x->timeout_ticks = (long long)(tsec * 1.0e8);      /* wall_clock64 runs at 100 MHz */

   /* which devices the ranks sit on */
   char bus[IPC_MAXR][64], mine[64];
   int dev = 0;
   memset(mine, 0, sizeof(mine));
   int fail = hipGetDevice(&dev) != hipSuccess || hipDeviceGetPCIBusId(mine, sizeof(mine) - 1, dev) != hipSuccess;
   if (pa_ipc_host_allgather(x, mine, sizeof(mine), bus)) fail = 1;
   x->distinct = 1;
   for (int p = 0; p < nranks; p++)
      for (int q = 0; q < p; q++)
         if (!strcmp(bus[p], bus[q])) x->distinct = 0;

   /* from here on a failure is not fatal: every rank learns of it through the rendez-vous, the device side is torn down
    * everywhere, and the communicator is told to take its collectives elsewhere (pa_ipc_gpu_ok) */
   if (!fail && (hipHostMalloc((void **)&x->err_host, 1024, hipHostMallocMapped) != hipSuccess ||
                 hipHostGetDevicePointer((void **)&x->err_dev, x->err_host, 0) != hipSuccess)) fail = 1;
   if (!fail) *x->err_host = 0;
   if (!fail && (hipMalloc((void **)&x->tab_dev, sizeof(void *) * IPC_MAXR * 2) != hipSuccess ||
                 hipMalloc((void **)&x->ticket, 64) != hipSuccess || hipMemset(x->ticket, 0, 64) != hipSuccess)) fail = 1;
   x->bar_dev = x->tab_dev ? x->tab_dev + IPC_MAXR : NULL;
   int bad = shm_agree(x, fail);
   if (bad < 0) { pa_ipc_detach(x); return -43; }           /* the rendez-vous itself broke */
   if (!bad) bad = region_open(x, &x->mbox, mbox_bytes(nranks)) != 0 || upload_tables(x) != 0;
   bad = shm_agree(x, bad);
   if (bad < 0) { pa_ipc_detach(x); return -43; }

   x->xr.tab = x->tab_dev; x->xr.nranks = nranks; x->xr.rank = rank; x->xr.slot_doubles = IPC_SLOT;
   x->xr.seq = &x->seq; x->xr.err_dev = x->err_dev; x->xr.timeout_ticks = x->timeout_ticks;
   x->xr.on_wrap = ipc_seq_wrap; x->xr.owner = x;
   {  /* PRIMME_AMD_IPC_SEQ0: where the tag sequence starts (test knob: a value just below 2^32 exercises the wrap-around) */
      const char *s0 = getenv("PRIMME_AMD_IPC_SEQ0");
      if (s0 && *s0) x->seq = (unsigned int)strtoul(s0, NULL, 10) & ~1u;      /* even: the next reduction keeps the generation alternating */
   }

   /* self-test: a reduction with known sums and a short time limit, so that a transport that maps but does not
    * deliver (no peer access, an incoherent mapping) is found here and not in the solver */
   if (!bad) {
      const long long keep = x->xr.timeout_ticks;
      x->xr.timeout_ticks = (long long)(5.0e8);
      double *d = NULL, *h = (double *)((char *)x->err_host + 512);     /* pinned scratch */
      h[0] = 1.0 + rank; h[1] = 0.5 * (rank + 1); h[2] = -2.0;
      int tb = hipMalloc((void **)&d, 3 * sizeof(double)) != hipSuccess ||
               hipMemcpyAsync(d, h, 3 * sizeof(double), hipMemcpyHostToDevice, NULL) != hipSuccess || hipStreamSynchronize(NULL) != hipSuccess;
      hipk_fin_flag nof = {NULL, NULL, 0};
      if (!tb) tb = pa_ipc_allreduce(x, NULL, d, 3, NULL, nof) != 0 || hipDeviceSynchronize() != hipSuccess ||
                    hipMemcpyAsync(h, d, 3 * sizeof(double), hipMemcpyDeviceToHost, NULL) != hipSuccess || hipStreamSynchronize(NULL) != hipSuccess;
      const double P = nranks;
      if (!tb) tb = *x->err_host != 0 || h[0] != P + P * (P - 1) / 2 || h[1] != 0.5 * (P * (P + 1) / 2) || h[2] != -2.0 * P;
      if (d) (void)hipFree(d);
      x->xr.timeout_ticks = keep;
      if (getenv("PRIMME_AMD_IPC_FAIL_SELFTEST")) tb = 1;       /* test knob: exercise the fall-back */
      if (tb) fprintf(stderr, "primme_amd: rank %d: the peer-to-peer mailbox self-test failed (err word %d)\n", rank, x->err_host ? *x->err_host : -1);
      bad = shm_agree(x, tb);
      if (bad < 0) { pa_ipc_detach(x); return -43; }
   }
   x->gpu_ok = !bad;
   if (bad) {
      if (rank == 0) fprintf(stderr, "primme_amd: the peer-to-peer mailboxes did not come up on every rank; the communicator falls back to RCCL\n");
      (void)hipDeviceSynchronize();
      region_close(x, &x->mbox);
      if (x->err_host) *x->err_host = 0;
   }
   if (getenv("PRIMME_AMD_COMM_VERBOSE") && rank == 0 && x->gpu_ok)
      fprintf(stderr, "primme_amd: peer-to-peer transport up: %d ranks, %s devices, mailbox memory kind %d\n", nranks,
            x->distinct ? "distinct" : "shared", x->alloc_kind);
   *out = x;
   return 0;
}

void pa_ipc_detach(pa_ipc *x) {
   if (!x) return;
   (void)hipDeviceSynchronize();
   /* nobody unmaps while a peer may still be writing: rendez-vous first (best effort: a dead peer only costs the time limit) */
   if (x->shm && x->shm != MAP_FAILED) (void)shm_barrier(x);
   region_close(x, &x->win);
   region_close(x, &x->hz);
   region_close(x, &x->mbox);
   if (x->tab_dev) (void)hipFree(x->tab_dev);
   if (x->ticket) (void)hipFree(x->ticket);
   if (x->err_host) (void)hipHostFree(x->err_host);
   if (x->shm && x->shm != MAP_FAILED) munmap(x->shm, sizeof(ipc_shm));
   free(x);
}

int pa_ipc_distinct_devices(pa_ipc *x) { return x->distinct; }
int pa_ipc_gpu_ok(pa_ipc *x) { return x && x->gpu_ok; }
int pa_ipc_error(pa_ipc *x) { return x && x->err_host ? *(volatile int *)x->err_host : 0; }
hipk_xreduce *pa_ipc_xreduce(pa_ipc *x) { return x ? &x->xr : NULL; }

/* ---- operations ------------------------------------------------------------------------------------- */
int pa_ipc_allreduce(pa_ipc *x, hipStream_t st, double *dbuf, int count, double *mirror, hipk_fin_flag fin) {
   for (int c0 = 0; c0 < count; c0 += IPC_SLOT) {
      const int n = count - c0 < IPC_SLOT ? count - c0 : IPC_SLOT;
      const int last = c0 + n >= count;
      hipk_fin_flag nof = {NULL, NULL, 0};
      hipLaunchKernelGGL(xr_allreduce_kernel, dim3((n + 15) / 16), dim3(HIPK_BLOCK), 0, st, dbuf + c0, mirror ? mirror + c0 : NULL, n,
            hipk_xr_make(&x->xr), last ? fin : nof);
      HIPK_CHECK(hipGetLastError());
   }
   return 0;
}

static int launch_barrier(pa_ipc *x, hipStream_t st) {
   xr_bar_args a = {x->bar_dev, x->nranks, x->rank, ++x->bseq, x->err_dev, x->timeout_ticks};
   hipLaunchKernelGGL(xr_barrier_kernel, dim3(1), dim3(64), 0, st, a);
   HIPK_CHECK(hipGetLastError());
   return 0;
}

int pa_ipc_halo(pa_ipc *x, hipStream_t st, const void *xv, int64_t ldx, int64_t nrows, int ncols, size_t elem,
      int64_t send_lo_cnt, int64_t send_hi_cnt, int64_t recv_lo_cnt, int64_t recv_hi_cnt, size_t max_side_bytes,
      void **lo_out, void **hi_out) {
   /* zones: [generation][side: 0 = from rank-1, 1 = from rank+1][side_cap]; all ranks hold the same side_cap */
   size_t side = (max_side_bytes + 255) & ~(size_t)255;
   if (side < 256) side = 256;
   if (4 * side > x->hz.cap) { if (region_reserve(x, st, &x->hz, 4 * side)) return -43; }
   side = x->hz.cap / 4;
   const unsigned long long seq = ++x->hseq;
   const int gen = (int)(seq & 1);
   const int r = x->rank, P = x->nranks;
   char *flags_me = (char *)x->mbox.mine + mbox_gran_bytes(P) + (size_t)IPC_MAXR * 8;
   xr_halo_args a;
   memset(&a, 0, sizeof(a));
   const size_t w = (elem % 8 == 0 && ((uintptr_t)xv % 8) == 0) ? 8 : 4;
   const int64_t f = (int64_t)(elem / w);
   a.x = xv; a.ldx = ldx * f; a.nrows = nrows * f; a.ncols = ncols;
   a.seq = seq; a.ticket = x->ticket; a.err = x->err_dev; a.timeout_ticks = x->timeout_ticks;
   if (r > 0) {
      a.send_lo = send_lo_cnt * f;
      /* my first rows are what rank-1 receives from ABOVE: its side 1 */
      a.dst_lo = (char *)x->hz.peer[r - 1] + ((size_t)gen * 2 + 1) * side;
      a.flag_lo_peer = (unsigned long long *)((char *)x->mbox.peer[r - 1] + mbox_gran_bytes(P) + (size_t)IPC_MAXR * 8) + 1;
      a.my_flag_lo = (const unsigned long long *)flags_me + 0;
      a.wait_lo = 1;
   }
   if (r < P - 1) {
      a.send_hi = send_hi_cnt * f;
      a.dst_hi = (char *)x->hz.peer[r + 1] + ((size_t)gen * 2 + 0) * side;
      a.flag_hi_peer = (unsigned long long *)((char *)x->mbox.peer[r + 1] + mbox_gran_bytes(P) + (size_t)IPC_MAXR * 8) + 0;
      a.my_flag_hi = (const unsigned long long *)flags_me + 1;
      a.wait_hi = 1;
   }
   if ((size_t)(recv_lo_cnt > recv_hi_cnt ? recv_lo_cnt : recv_hi_cnt) * ncols * elem > side ||
       (size_t)(send_lo_cnt > send_hi_cnt ? send_lo_cnt : send_hi_cnt) * ncols * elem > side) {
      fprintf(stderr, "primme_amd: halo of %lld rows x %d columns exceeds the agreed landing zone (%zu bytes)\n",
            (long long)(recv_lo_cnt > recv_hi_cnt ? recv_lo_cnt : recv_hi_cnt), ncols, side);
      return -1;
   }
   const int64_t words = (a.send_lo > a.send_hi ? a.send_lo : a.send_hi) * ncols;
   int gx = (int)((words + HIPK_BLOCK * 4 - 1) / (HIPK_BLOCK * 4));
   gx = gx < 1 ? 1 : gx > 64 ? 64 : gx;
   if (w == 8) hipLaunchKernelGGL(xr_halo_kernel<unsigned long long>, dim3(gx), dim3(HIPK_BLOCK), 0, st, a);
   else hipLaunchKernelGGL(xr_halo_kernel<unsigned int>, dim3(gx), dim3(HIPK_BLOCK), 0, st, a);
   HIPK_CHECK(hipGetLastError());
   *lo_out = (char *)x->hz.mine + ((size_t)gen * 2 + 0) * side;
   *hi_out = (char *)x->hz.mine + ((size_t)gen * 2 + 1) * side;
   return 0;
}

/* one operation on the bulk windows: make room (collective), pick the generation */
static int window_op(pa_ipc *x, hipStream_t st, size_t bytes_per_gen, void **peers, char **mine_out) {
   size_t half = (bytes_per_gen + 255) & ~(size_t)255;
   if (2 * half > x->win.cap) { if (region_reserve(x, st, &x->win, 2 * half)) return -43; }
   half = x->win.cap / 2;
   const int gen = (int)(++x->wseq & 1);
   for (int p = 0; p < x->nranks; p++) peers[p] = (char *)x->win.peer[p] + (size_t)gen * half;
   *mine_out = (char *)x->win.mine + (size_t)gen * half;
   return 0;
}

int pa_ipc_allgather_cols(pa_ipc *x, hipStream_t st, const void *send, int64_t ld_send, void *recv, int64_t ld_recv,
      size_t bytes_per_rank, size_t elem, int ncols) {
   if (ncols <= 0 || bytes_per_rank == 0) return 0;
   const size_t w = (bytes_per_rank % 8 == 0 && elem % 8 == 0 && ((uintptr_t)send % 8) == 0 && ((uintptr_t)recv % 8) == 0) ? 8 : 4;
   if (bytes_per_rank % w || elem % w) return -1;
   char *mine;
   xr_push_args a;
   memset(&a, 0, sizeof(a));
   int rc = window_op(x, st, bytes_per_rank * x->nranks * ncols, a.win, &mine);
   if (rc) return rc;
   a.send = send; a.ld_send = (int64_t)(ld_send * (elem / w)); a.ncols = ncols; a.cnt = (int64_t)(bytes_per_rank / w);
   a.nranks = x->nranks; a.rank = x->rank;
   const int64_t words = a.cnt * ncols;
   int gx = (int)((words + HIPK_BLOCK * 4 - 1) / (HIPK_BLOCK * 4));
   gx = gx < 1 ? 1 : gx > 256 ? 256 : gx;
   if (w == 8) hipLaunchKernelGGL(xr_push_kernel<unsigned long long>, dim3(gx, x->nranks), dim3(HIPK_BLOCK), 0, st, a);
   else hipLaunchKernelGGL(xr_push_kernel<unsigned int>, dim3(gx, x->nranks), dim3(HIPK_BLOCK), 0, st, a);
   HIPK_CHECK(hipGetLastError());
   if (launch_barrier(x, st)) return -1;
   const int64_t colwords = a.cnt * x->nranks;
   int gu = (int)((colwords * ncols + HIPK_BLOCK * 4 - 1) / (HIPK_BLOCK * 4));
   gu = gu < 1 ? 1 : gu > 2048 ? 2048 : gu;
   if (w == 8) hipLaunchKernelGGL(xr_unpack_kernel<unsigned long long>, dim3(gu), dim3(HIPK_BLOCK), 0, st, (const unsigned long long *)mine,
         (unsigned long long *)recv, (int64_t)(ld_recv * (elem / w)), colwords, ncols);
   else hipLaunchKernelGGL(xr_unpack_kernel<unsigned int>, dim3(gu), dim3(HIPK_BLOCK), 0, st, (const unsigned int *)mine,
         (unsigned int *)recv, (int64_t)(ld_recv * (elem / w)), colwords, ncols);
   HIPK_CHECK(hipGetLastError());
   return 0;
}

int pa_ipc_reduce_scatter_cols(pa_ipc *x, hipStream_t st, const void *send, int64_t ld_send, void *recv, int64_t ld_recv,
      size_t count_per_rank, int is_double, int ncols) {
   if (ncols <= 0 || count_per_rank == 0) return 0;
   const size_t es = is_double ? 8 : 4;
   char *mine;
   xr_push_args a;
   memset(&a, 0, sizeof(a));
   int rc = window_op(x, st, count_per_rank * es * x->nranks * ncols, a.win, &mine);
   if (rc) return rc;
   a.send = send; a.ld_send = ld_send; a.ncols = ncols; a.cnt = (int64_t)count_per_rank; a.off_per_peer = (int64_t)count_per_rank;
   a.nranks = x->nranks; a.rank = x->rank;
   const int64_t words = a.cnt * ncols;
   int gx = (int)((words + HIPK_BLOCK * 4 - 1) / (HIPK_BLOCK * 4));
   gx = gx < 1 ? 1 : gx > 256 ? 256 : gx;
   if (is_double) hipLaunchKernelGGL(xr_push_kernel<unsigned long long>, dim3(gx, x->nranks), dim3(HIPK_BLOCK), 0, st, a);
   else hipLaunchKernelGGL(xr_push_kernel<unsigned int>, dim3(gx, x->nranks), dim3(HIPK_BLOCK), 0, st, a);
   HIPK_CHECK(hipGetLastError());
   if (launch_barrier(x, st)) return -1;
   int gu = (int)((words + HIPK_BLOCK * 2 - 1) / (HIPK_BLOCK * 2));
   gu = gu < 1 ? 1 : gu > 2048 ? 2048 : gu;
   if (is_double) hipLaunchKernelGGL(xr_sum_kernel<double>, dim3(gu), dim3(HIPK_BLOCK), 0, st, (const double *)mine, (double *)recv, ld_recv,
         (int64_t)count_per_rank, ncols, x->nranks);
   else hipLaunchKernelGGL(xr_sum_kernel<float>, dim3(gu), dim3(HIPK_BLOCK), 0, st, (const float *)mine, (float *)recv, ld_recv,
         (int64_t)count_per_rank, ncols, x->nranks);
   HIPK_CHECK(hipGetLastError());
   return 0;
}
