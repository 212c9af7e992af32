// Where do the 55 us of the row-pattern product at 10 M rows go?  (profiles/r05_spmv_row_pattern_form.txt: the kernel's time does
// not move with occupancy, rows per lane, non-temporal stores, instruction count or the order of its loads.)  The 5-point stencil
// y[r] = 4 x[r] - x[r-1] - x[r+1] - x[r-nx] - x[r+nx] on nx*ny rows (interior rows only: the probe clamps indices) in forms that
// add the kernel's ingredients one at a time:
//   copy      : y[r] = x[r]                                   one workgroup per 256 rows (what the memory system gives a 1-read 1-write pass)
//   plain     : the stencil, one row per thread, one workgroup per 256 rows, rows in launch order
//   far       : only x[r], x[r-nx], x[r+nx] (three different cache lines);  shfl: +-1 from the neighbouring lanes;  lds: +-1 from LDS
//   pid       : plain + one pattern byte per row read from HBM (not used for addressing)
//   xcd       : plain, but persistent: 256*WPS workgroups, XCD q walks the q-th eighth of the rows (pat_kernel's schedule)
//   xcd4      : xcd with 4 rows per thread and trip
//   blocked   : plain with the launch-order -> rows mapping of xcd but NOT persistent (one workgroup per 256 rows, workgroup b of XCD q takes chunk q*per + b/8)
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/stencil_probe.hip -o scripts/probes/stencil_probe && scripts/probes/stencil_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ double st5(const double *__restrict__ x, int64_t r, int64_t n, int64_t nx) {
   const int64_t a = r - 1 < 0 ? 0 : r - 1, b = r + 1 >= n ? n - 1 : r + 1, c = r - nx < 0 ? 0 : r - nx, d = r + nx >= n ? n - 1 : r + nx;
   const double x0 = x[r], xa = x[a], xb = x[b], xc = x[c], xd = x[d];
   return 4.0 * x0 - xa - xb - xc - xd;
}
__global__ void __launch_bounds__(256) k_copy(const double *__restrict__ x, double *__restrict__ y, int64_t n) {
   const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
   if (r < n) y[r] = x[r];
}
__global__ void __launch_bounds__(256) k_plain(const double *__restrict__ x, double *__restrict__ y, int64_t n, int64_t nx) {
   const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
   if (r < n) y[r] = st5(x, r, n, nx);
}
__global__ void __launch_bounds__(256) k_pid(const double *__restrict__ x, double *__restrict__ y, int64_t n, int64_t nx, const uint8_t *__restrict__ pid) {
   const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
   if (r < n) { const double s = st5(x, r, n, nx); y[r] = pid[r] == 255 ? 0.0 : s; }
}
// no +-1: only the three loads that fall into different cache lines
__global__ void __launch_bounds__(256) k_far(const double *__restrict__ x, double *__restrict__ y, int64_t n, int64_t nx) {
   const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
   if (r < n) { const int64_t c = r - nx < 0 ? 0 : r - nx, d = r + nx >= n ? n - 1 : r + nx; y[r] = 4.0 * x[r] - x[c] - x[d]; }
}
// +-1 out of the neighbouring lanes' registers (ds_bpermute), the two edge lanes of a wave load theirs: one load per cache line and wave
__global__ void __launch_bounds__(256) k_shfl(const double *__restrict__ x, double *__restrict__ y, int64_t n, int64_t nx) {
   const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
   const int lane = threadIdx.x & 63;
   const int64_t rr = r < n ? r : n - 1;
   const int64_t c = rr - nx < 0 ? 0 : rr - nx, d = rr + nx >= n ? n - 1 : rr + nx;
   const double x0 = x[rr], xc = x[c], xd = x[d];
   double xa = __shfl_up(x0, 1, 64), xb = __shfl_down(x0, 1, 64);
   if (lane == 0) xa = x[rr - 1 < 0 ? 0 : rr - 1];
   if (lane == 63) xb = x[rr + 1 >= n ? n - 1 : rr + 1];
   if (r < n) y[r] = 4.0 * x0 - xa - xb - xc - xd;
}
// the whole workgroup's rows (+-1) staged in LDS: every cache line of the own rows requested once per workgroup
__global__ void __launch_bounds__(256) k_lds(const double *__restrict__ x, double *__restrict__ y, int64_t n, int64_t nx) {
   __shared__ double t[258];
   const int64_t r0 = (int64_t)blockIdx.x * 256, r = r0 + threadIdx.x;
   const int64_t rr = r < n ? r : n - 1;
   const int64_t c = rr - nx < 0 ? 0 : rr - nx, d = rr + nx >= n ? n - 1 : rr + nx;
   const double x0 = x[rr], xc = x[c], xd = x[d];
   t[threadIdx.x + 1] = x0;
   if (threadIdx.x == 0) t[0] = x[r0 - 1 < 0 ? 0 : r0 - 1];
   if (threadIdx.x == 255) t[257] = x[r0 + 256 >= n ? n - 1 : r0 + 256];
   __syncthreads();
   if (r < n) y[r] = 4.0 * x0 - t[threadIdx.x] - t[threadIdx.x + 2] - xc - xd;
}
// two consecutive rows per thread: every access is 16 bytes per lane (the +-1 and +-nx ones only 8-byte aligned)
struct __attribute__((packed, aligned(8))) d2u { double a, b; };
__device__ __forceinline__ double2 ld2(const double *p) { const d2u v = *(const d2u *)p; return make_double2(v.a, v.b); }
__global__ void __launch_bounds__(256) k_copy2(const double *__restrict__ x, double *__restrict__ y, int64_t n) {
   const int64_t r = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2;
   if (r + 1 < n) *(double2 *)(y + r) = *(const double2 *)(x + r);
}
__global__ void __launch_bounds__(256) k_plain2(const double *__restrict__ x, double *__restrict__ y, int64_t n, int64_t nx) {
   const int64_t r = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2;
   if (r - nx >= 0 && r + nx + 1 < n) {
      const double2 x0 = *(const double2 *)(x + r), xa = ld2(x + r - 1), xb = ld2(x + r + 1), xc = ld2(x + r - nx), xd = ld2(x + r + nx);
      *(double2 *)(y + r) = make_double2(4.0 * x0.x - xa.x - xb.x - xc.x - xd.x, 4.0 * x0.y - xa.y - xb.y - xc.y - xd.y);
   }
}
__global__ void __launch_bounds__(256) k_far2(const double *__restrict__ x, double *__restrict__ y, int64_t n, int64_t nx) {
   const int64_t r = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2;
   if (r - nx >= 0 && r + nx + 1 < n) {
      const double2 x0 = *(const double2 *)(x + r), xc = ld2(x + r - nx), xd = ld2(x + r + nx);
      *(double2 *)(y + r) = make_double2(4.0 * x0.x - xc.x - xd.x, 4.0 * x0.y - xc.y - xd.y);
   }
}
template <int WPS>
__global__ void __launch_bounds__(256, WPS) k_xcd2(const double *__restrict__ x, double *__restrict__ y, int64_t n, int64_t nx) {
   const int64_t CH = 512, nch = (n + CH - 1) / CH, per = (nch + 7) >> 3;
   const int q = blockIdx.x & 7, j = blockIdx.x >> 3, J = gridDim.x >> 3;
   const int64_t c_lo = q * per, c_hi = c_lo + per < nch ? c_lo + per : nch;
   for (int64_t c = c_lo + j; c < c_hi; c += J) {
      const int64_t r = c * CH + threadIdx.x * 2;
      if (r - nx >= 0 && r + nx + 1 < n) {
         const double2 x0 = *(const double2 *)(x + r), xa = ld2(x + r - 1), xb = ld2(x + r + 1), xc = ld2(x + r - nx), xd = ld2(x + r + nx);
         *(double2 *)(y + r) = make_double2(4.0 * x0.x - xa.x - xb.x - xc.x - xd.x, 4.0 * x0.y - xa.y - xb.y - xc.y - xd.y);
      }
   }
}
template <int RPL, int WPS>
__global__ void __launch_bounds__(256, WPS) k_xcd(const double *__restrict__ x, double *__restrict__ y, int64_t n, int64_t nx) {
   const int64_t CH = 256 * RPL, nch = (n + CH - 1) / CH, per = (nch + 7) >> 3;
   const int q = blockIdx.x & 7, j = blockIdx.x >> 3, J = gridDim.x >> 3;
   const int64_t c_lo = q * per, c_hi = c_lo + per < nch ? c_lo + per : nch;
   for (int64_t c = c_lo + j; c < c_hi; c += J) {
      double s[RPL];
#pragma unroll
      for (int u = 0; u < RPL; u++) { const int64_t r = c * CH + threadIdx.x + u * 256; s[u] = st5(x, r < n ? r : n - 1, n, nx); }
#pragma unroll
      for (int u = 0; u < RPL; u++) { const int64_t r = c * CH + threadIdx.x + u * 256; if (r < n) y[r] = s[u]; }
   }
}
__global__ void __launch_bounds__(256) k_blocked(const double *__restrict__ x, double *__restrict__ y, int64_t n, int64_t nx) {
   const int64_t nch = (n + 255) / 256, per = (nch + 7) >> 3;
   const int q = blockIdx.x & 7;
   const int64_t c = q * per + (blockIdx.x >> 3);
   const int64_t r = c * 256 + threadIdx.x;
   if ((blockIdx.x >> 3) < per && r < n) y[r] = st5(x, r, n, nx);
}

int main() {
   const int64_t nx = 3162, ny = 3163, n = nx * ny;
   const int ncol = 12;                               // rotate over 12 vectors (960 MB each way): nothing stays in the 256 MiB cache
   double *x, *y; uint8_t *pid;
   CHECK(hipMalloc(&x, ncol * n * 8)); CHECK(hipMalloc(&y, ncol * n * 8)); CHECK(hipMalloc(&pid, n));
   CHECK(hipMemset(x, 0, ncol * n * 8)); CHECK(hipMemset(y, 0, ncol * n * 8)); CHECK(hipMemset(pid, 1, n));
   hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
   const int gb = (int)((n + 255) / 256);
   const int reps = 60;
   auto run = [&](const char *name, auto launch, double bytes) {
      for (int i = 0; i < 6; i++) launch(i % ncol);
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0));
      for (int i = 0; i < reps; i++) launch(i % ncol);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      const double us = 1e3 * ms / reps;
      printf("%-10s %8.1f us   %7.0f GB/s of %.0f MB\n", name, us, bytes / us / 1e3, bytes / 1e6);
   };
   run("copy", [&](int c) { hipLaunchKernelGGL(k_copy, dim3(gb), dim3(256), 0, 0, x + c * n, y + c * n, n); }, 16.0 * n);
   run("plain", [&](int c) { hipLaunchKernelGGL(k_plain, dim3(gb), dim3(256), 0, 0, x + c * n, y + c * n, n, nx); }, 16.0 * n);
   run("far", [&](int c) { hipLaunchKernelGGL(k_far, dim3(gb), dim3(256), 0, 0, x + c * n, y + c * n, n, nx); }, 16.0 * n);
   run("shfl", [&](int c) { hipLaunchKernelGGL(k_shfl, dim3(gb), dim3(256), 0, 0, x + c * n, y + c * n, n, nx); }, 16.0 * n);
   run("lds", [&](int c) { hipLaunchKernelGGL(k_lds, dim3(gb), dim3(256), 0, 0, x + c * n, y + c * n, n, nx); }, 16.0 * n);
   run("copy2", [&](int c) { hipLaunchKernelGGL(k_copy2, dim3((gb + 1) / 2), dim3(256), 0, 0, x + c * n, y + c * n, n); }, 16.0 * n);
   run("plain2", [&](int c) { hipLaunchKernelGGL(k_plain2, dim3((gb + 1) / 2), dim3(256), 0, 0, x + c * n, y + c * n, n, nx); }, 16.0 * n);
   run("far2", [&](int c) { hipLaunchKernelGGL(k_far2, dim3((gb + 1) / 2), dim3(256), 0, 0, x + c * n, y + c * n, n, nx); }, 16.0 * n);
   run("xcd2 w8", [&](int c) { hipLaunchKernelGGL((k_xcd2<8>), dim3(256 * 8), dim3(256), 0, 0, x + c * n, y + c * n, n, nx); }, 16.0 * n);
   run("xcd2 w4", [&](int c) { hipLaunchKernelGGL((k_xcd2<4>), dim3(256 * 4), dim3(256), 0, 0, x + c * n, y + c * n, n, nx); }, 16.0 * n);
   run("pid", [&](int c) { hipLaunchKernelGGL(k_pid, dim3(gb), dim3(256), 0, 0, x + c * n, y + c * n, n, nx, pid); }, 17.0 * n);
   run("blocked", [&](int c) { hipLaunchKernelGGL(k_blocked, dim3(((gb + 7) / 8 + 1) * 8), dim3(256), 0, 0, x + c * n, y + c * n, n, nx); }, 16.0 * n);
   run("xcd1 w8", [&](int c) { hipLaunchKernelGGL((k_xcd<1, 8>), dim3(256 * 8), dim3(256), 0, 0, x + c * n, y + c * n, n, nx); }, 16.0 * n);
   run("xcd1 w4", [&](int c) { hipLaunchKernelGGL((k_xcd<1, 4>), dim3(256 * 4), dim3(256), 0, 0, x + c * n, y + c * n, n, nx); }, 16.0 * n);
   run("xcd4 w6", [&](int c) { hipLaunchKernelGGL((k_xcd<4, 6>), dim3(256 * 6), dim3(256), 0, 0, x + c * n, y + c * n, n, nx); }, 16.0 * n);
   run("xcd4 w2", [&](int c) { hipLaunchKernelGGL((k_xcd<4, 2>), dim3(256 * 2), dim3(256), 0, 0, x + c * n, y + c * n, n, nx); }, 16.0 * n);
   return 0;
}
