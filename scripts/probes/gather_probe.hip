// Calibration probe for the scattered-gather products of BASELINE configs[4] (profiles/r05_gather_calibration.md):
// known-byte micro-kernels whose FETCH_SIZE (rocprofv3 --pmc FETCH_SIZE --kernel-trace) tells how the counter treats
//   stream12  : a pure 12-byte entry stream (8-byte value + 4-byte word per entry, non-temporal, coalesced)      -> 12 N bytes
//   gather_idx: the 4-byte word stream + one 8-byte gather per entry out of a vector of NX doubles, random column -> 4 N + gathers
//   gather_ari: gathers only, the column computed from the entry number ((e * 7919) mod NX): no index stream      -> gathers only
//   gather_l2 : the same gathers confined to a 2 MB panel of the vector (what the panel-blocked kernel arranges)   -> L2 hits
// and whose run times are the ceiling a gather-bound kernel can be held against: entries per microsecond.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/gather_probe.hip -o scripts/probes/gather_probe && scripts/probes/gather_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
#define U 4

__global__ void __launch_bounds__(256) stream12(const double *__restrict__ val, const uint32_t *__restrict__ w, size_t n, double *out) {
   double acc = 0.0;
   const size_t stride = (size_t)gridDim.x * 256;
   for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += stride * U) {
      double v[U]; uint32_t k[U];
#pragma unroll
      for (int u = 0; u < U; u++) { const size_t q = e + u * stride < n ? e + u * stride : n - 1; v[u] = __builtin_nontemporal_load(val + q); k[u] = __builtin_nontemporal_load(w + q); }
#pragma unroll
      for (int u = 0; u < U; u++) acc += v[u] * (double)(k[u] & 1u);
   }
   if (acc == 123.456) out[0] = acc;
}
template <int MODE>   // 0: index stream, whole vector; 1: arithmetic column, whole vector; 2: arithmetic column inside a 2 MB panel that moves with the entry number
__global__ void __launch_bounds__(256) gather(const uint32_t *__restrict__ w, const double *__restrict__ x, size_t n, uint32_t nx, double *out) {
   double acc = 0.0;
   const size_t stride = (size_t)gridDim.x * 256;
   for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += stride * U) {
      uint32_t c[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
         const size_t q = e + u * stride < n ? e + u * stride : n - 1;
         if (MODE == 0) c[u] = __builtin_nontemporal_load(w + q);
         else if (MODE == 1) c[u] = (uint32_t)((q * 7919ull) % nx);
         else { const uint32_t panel = (uint32_t)((q * 8ull) / n) % 8u; c[u] = panel * (nx / 8) + (uint32_t)((q * 7919ull) % (nx / 8)); }
      }
      double g[U];
#pragma unroll
      for (int u = 0; u < U; u++) g[u] = x[c[u]];
#pragma unroll
      for (int u = 0; u < U; u++) acc += g[u];
   }
   if (acc == 123.456) out[0] = acc;
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
   const size_t n = 40000000;          // entries of one configs[4] product
   const uint32_t nx = argc > 1 ? (uint32_t)atol(argv[1]) : 2000000;      // 2 M (x of A x) or 8 M (u of A'u)
   double *val, *x, *out; uint32_t *w;
   CHECK(hipMalloc(&val, n * 8)); CHECK(hipMalloc(&w, n * 4)); CHECK(hipMalloc(&x, (size_t)nx * 8)); CHECK(hipMalloc(&out, 64));
   std::vector<uint32_t> hw(n);
   uint64_t s = 88172645463325252ull;
   for (size_t i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; hw[i] = (uint32_t)(s % nx); }
   CHECK(hipMemcpy(w, hw.data(), n * 4, hipMemcpyHostToDevice));
   CHECK(hipMemset(val, 0, n * 8)); CHECK(hipMemset(x, 0, (size_t)nx * 8));
   int dev = 0, cus = 256; hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, dev)); cus = p.multiProcessorCount;
   const int grid = cus * 8, reps = 20;
   printf("entries %zu, gathered vector %u doubles (%.0f MB), grid %d x 256, %d entries in flight per lane\n", n, nx, nx * 8.0 / 1e6, grid, U);
   double us;
   us = time_us([&] { hipLaunchKernelGGL(stream12, dim3(grid), dim3(256), 0, 0, val, w, n, out); }, reps);
   printf("stream12    %8.1f us   %7.0f GB/s of 12 N bytes                      %6.1f entries/ns\n", us, 12.0 * n / us / 1e3, n / us / 1e3);
   us = time_us([&] { hipLaunchKernelGGL(gather<0>, dim3(grid), dim3(256), 0, 0, w, x, n, nx, out); }, reps);
   printf("gather_idx  %8.1f us   %7.0f GB/s of 4 N + 8 N useful bytes             %6.1f gathers/ns\n", us, 12.0 * n / us / 1e3, n / us / 1e3);
   us = time_us([&] { hipLaunchKernelGGL(gather<1>, dim3(grid), dim3(256), 0, 0, w, x, n, nx, out); }, reps);
   printf("gather_ari  %8.1f us   %7.0f GB/s of 8 N useful bytes                   %6.1f gathers/ns\n", us, 8.0 * n / us / 1e3, n / us / 1e3);
   us = time_us([&] { hipLaunchKernelGGL(gather<2>, dim3(grid), dim3(256), 0, 0, w, x, n, nx, out); }, reps);
   printf("gather_l2   %8.1f us   %7.0f GB/s of 8 N useful bytes (2 MB panels)     %6.1f gathers/ns\n", us, 8.0 * n / us / 1e3, n / us / 1e3);
   return 0;
}
