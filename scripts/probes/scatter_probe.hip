// Probe for the two alternatives to the panel-blocked products of BASELINE configs[4] that round 5 argued about without
// measuring them (VERDICT r05 Next #5; profiles/r06_gather_calibration.md):
//   atomic_l2 : the scatter half of a fused y = A'(A x): 40 M FP64 atomic adds (no return value: L2 atomics) into a
//               vector of NY doubles, entries streamed as {value 8 B, word 4 B}, the destination confined to a 2 MB panel
//               that moves with the entry number (the schedule pb_matvec_kernel arranges for its gathers) -> adds per ns
//   atomic_any: the same without the panel confinement (destinations all over the vector)
//   pb2       : a PROPAGATION-BLOCKING product y = A x in two passes with a static plan (the matrix is fixed, so every entry's
//               slot in its bin is known at set-up): pass 1 streams the entries in COLUMN order ({value 8 B, slot 4 B}; x is
//               read sequentially, 20 entries per column) and writes the 8-byte contribution value * x[col] to its slot in
//               one of NB row bins (the bins' write frontiers are NB x 64 B: they live in the L2, the scattered 8-byte stores
//               combine there); pass 2, one workgroup per bin, streams the bin's contributions (8 B) and their row-in-bin
//               numbers (2 B, static), accumulates them in LDS (ds_add_f64) and writes the bin's rows of y once.
//               Bytes: pass 1 12 N + 8 NX read, 8 N written; pass 2 10 N read, 8 NY written  =  30 N + 8 (NX + NY).
// to be held against pb_matvec_kernel: 320-327 us per product of 40 M entries (480 MB of entries + 64 MB of y).
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/scatter_probe.hip -o scripts/probes/scatter_probe && scripts/probes/scatter_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <numeric>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
#define U 4

template <int PANEL>
__global__ void __launch_bounds__(256) atomic_add(const double *__restrict__ val, const uint32_t *__restrict__ w, size_t n, uint32_t ny, double *__restrict__ y) {
   const size_t stride = (size_t)gridDim.x * 256;
   for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += stride * U) {
      double v[U]; uint32_t c[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
         const size_t q = e + u * stride < n ? e + u * stride : n - 1;
         v[u] = __builtin_nontemporal_load(val + q);
         const uint32_t k = __builtin_nontemporal_load(w + q);
         if (PANEL) { const uint32_t np = ny / 262144u ? ny / 262144u : 1u; const uint32_t panel = (uint32_t)((q * (uint64_t)np) / n); c[u] = panel * 262144u + (k & 262143u); if (c[u] >= ny) c[u] = k % ny; }
         else c[u] = k % ny;
      }
#pragma unroll
      for (int u = 0; u < U; u++) if (e + u * stride < n) __hip_atomic_fetch_add(y + c[u], v[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
   }
}

// pass 1: entries in column order; column of entry e = colstart search is avoided: the probe stores the column per entry group
// implicitly — entry e belongs to column e / EPC (EPC entries per column, the synthetic pattern's 20): x is read in order
__global__ void __launch_bounds__(256) pb2_scatter(const double *__restrict__ val, const uint32_t *__restrict__ slot, const double *__restrict__ x, size_t n,
      uint32_t epc, double *__restrict__ contrib) {
   const size_t stride = (size_t)gridDim.x * 256;
   for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += stride * U) {
      double v[U], xv[U]; uint32_t s[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
         const size_t q = e + u * stride < n ? e + u * stride : n - 1;
         v[u] = __builtin_nontemporal_load(val + q); s[u] = __builtin_nontemporal_load(slot + q); xv[u] = x[q / epc];
      }
#pragma unroll
      for (int u = 0; u < U; u++) if (e + u * stride < n) contrib[s[u]] = v[u] * xv[u];
   }
}
// pass 2: one workgroup per bin of RB rows
template <int RB>
__global__ void __launch_bounds__(256) pb2_gather(const double *__restrict__ contrib, const uint16_t *__restrict__ rowin, const uint32_t *__restrict__ binptr,
      double *__restrict__ y, uint32_t ny) {
   __shared__ double acc[RB];
   for (int i = threadIdx.x; i < RB; i += 256) acc[i] = 0.0;
   __syncthreads();
   const uint32_t b = blockIdx.x, lo = binptr[b], hi = binptr[b + 1];
   for (uint32_t e = lo + threadIdx.x; e < hi; e += 256 * U) {
      double v[U]; uint16_t r[U];
#pragma unroll
      for (int u = 0; u < U; u++) { const uint32_t q = e + u * 256 < hi ? e + u * 256 : hi - 1; v[u] = __builtin_nontemporal_load(contrib + q); r[u] = __builtin_nontemporal_load(rowin + q); }
#pragma unroll
      for (int u = 0; u < U; u++) if (e + u * 256 < hi) atomicAdd(&acc[r[u]], v[u]);
   }
   __syncthreads();
   const size_t r0 = (size_t)b * RB;
   for (int i = threadIdx.x; i < RB; i += 256) if (r0 + i < ny) y[r0 + i] = acc[i];
}

template <typename F> static double time_us(F launch, int reps) {
   hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
   for (int i = 0; i < 3; i++) launch();
   CHECK(hipEventRecord(a));
   for (int i = 0; i < reps; i++) launch();
   CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
   float ms; CHECK(hipEventElapsedTime(&ms, a, b));
   return 1e3 * ms / reps;
}

int main(int argc, char **argv) {
   const size_t n = 40000000;                                             // entries of one configs[4] product
   const uint32_t nx = argc > 1 ? (uint32_t)atol(argv[1]) : 2000000;      // input length  (2 M: y = A x;  8 M: y = A'u)
   const uint32_t ny = argc > 2 ? (uint32_t)atol(argv[2]) : 8000000;      // output length
   const uint32_t epc = (uint32_t)(n / nx);                               // entries per input column (20 or 5)
   constexpr int RB = 4096;                                               // rows per bin: 32 KB of LDS accumulators
   const uint32_t nb = (ny + RB - 1) / RB;
   double *val, *x, *y, *contrib; uint32_t *w, *slot, *binptr; uint16_t *rowin;
   CHECK(hipMalloc(&val, n * 8)); CHECK(hipMalloc(&w, n * 4)); CHECK(hipMalloc(&slot, n * 4)); CHECK(hipMalloc(&x, (size_t)nx * 8)); CHECK(hipMalloc(&y, (size_t)ny * 8));
   CHECK(hipMalloc(&contrib, n * 8)); CHECK(hipMalloc(&rowin, n * 2)); CHECK(hipMalloc(&binptr, (size_t)(nb + 1) * 4));
   // the pattern: entry e (column order) goes to a pseudo-random row; the static plan sorts entries by bin (stable: column order
   // within a bin, which is also the order pass 1 writes them in -> each bin's frontier advances monotonically)
   std::vector<uint32_t> row(n), hslot(n), hbin(nb + 1, 0), cursor(nb);
   std::vector<uint16_t> hrow(n);
   uint64_t s = 88172645463325252ull;
   for (size_t i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; row[i] = (uint32_t)(s % ny); }
   for (size_t i = 0; i < n; i++) hbin[row[i] / RB + 1]++;
   for (uint32_t b = 0; b < nb; b++) hbin[b + 1] += hbin[b];
   for (uint32_t b = 0; b < nb; b++) cursor[b] = hbin[b];
   for (size_t i = 0; i < n; i++) { const uint32_t b = row[i] / RB; const uint32_t p = cursor[b]++; hslot[i] = p; hrow[p] = (uint16_t)(row[i] % RB); }
   CHECK(hipMemcpy(w, row.data(), n * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(slot, hslot.data(), n * 4, hipMemcpyHostToDevice));
   CHECK(hipMemcpy(rowin, hrow.data(), n * 2, hipMemcpyHostToDevice)); CHECK(hipMemcpy(binptr, hbin.data(), (size_t)(nb + 1) * 4, hipMemcpyHostToDevice));
   std::vector<double> hv(n), hx(nx);
   for (size_t i = 0; i < n; i++) hv[i] = 1.0 + (double)(i % 13) / 13.0;
   for (uint32_t i = 0; i < nx; i++) hx[i] = 1.0 + (double)(i % 7) / 7.0;
   CHECK(hipMemcpy(val, hv.data(), n * 8, hipMemcpyHostToDevice)); CHECK(hipMemcpy(x, hx.data(), (size_t)nx * 8, hipMemcpyHostToDevice));
   int cus = 256; hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0)); cus = p.multiProcessorCount;
   const int grid = cus * 8, reps = 20;
   printf("entries %zu, input %u doubles, output %u doubles (%.0f MB), %u bins of %d rows, grid %d x 256\n", n, nx, ny, ny * 8.0 / 1e6, nb, RB, grid);
   double us;
   CHECK(hipMemset(y, 0, (size_t)ny * 8));
   us = time_us([&] { hipLaunchKernelGGL(atomic_add<1>, dim3(grid), dim3(256), 0, 0, val, w, n, ny, y); }, reps);
   printf("atomic_l2   %8.1f us   %6.1f adds/ns   (12 N bytes streamed, destinations inside a moving 2 MB panel)\n", us, n / us / 1e3);
   us = time_us([&] { hipLaunchKernelGGL(atomic_add<0>, dim3(grid), dim3(256), 0, 0, val, w, n, ny, y); }, reps);
   printf("atomic_any  %8.1f us   %6.1f adds/ns   (destinations all over the %0.f MB vector)\n", us, n / us / 1e3, ny * 8.0 / 1e6);
   const double us1 = time_us([&] { hipLaunchKernelGGL(pb2_scatter, dim3(grid), dim3(256), 0, 0, val, slot, x, n, epc, contrib); }, reps);
   const double us2 = time_us([&] { hipLaunchKernelGGL(pb2_gather<RB>, dim3(nb), dim3(256), 0, 0, contrib, rowin, binptr, y, ny); }, reps);
   const double b1 = 12.0 * n + 8.0 * nx + 8.0 * n, b2 = 10.0 * n + 8.0 * ny;
   printf("pb2 pass 1  %8.1f us   %7.0f GB/s of %.0f MB (entries + x read, contributions written)\n", us1, b1 / us1 / 1e3, b1 / 1e6);
   printf("pb2 pass 2  %8.1f us   %7.0f GB/s of %.0f MB (contributions + row numbers read, y written)\n", us2, b2 / us2 / 1e3, b2 / 1e6);
   printf("pb2 total   %8.1f us   for %.0f MB   (pb_matvec_kernel: 320-327 us for the same product)\n", us1 + us2, (b1 + b2) / 1e6);
   // check pass 1 + 2 against the host
   std::vector<double> hy(ny), ref(ny, 0.0);
   CHECK(hipMemcpy(hy.data(), y, (size_t)ny * 8, hipMemcpyDeviceToHost));
   for (size_t i = 0; i < n; i++) ref[row[i]] += hv[i] * hx[i / epc];
   double err = 0.0; for (uint32_t i = 0; i < ny; i++) err = std::max(err, std::abs(hy[i] - ref[i]));
   printf("pb2 max |y - reference| = %.3e\n", err);
   return 0;
}
