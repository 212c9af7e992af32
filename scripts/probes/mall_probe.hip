// Does the 256 MB memory-side cache of the MI355X keep part of a panel between two consecutive streaming passes?
// Pass A reads the columns of V and W of every row (the fused residual pass), pass C reads V only (the Gram-Schmidt update).
// Both sweep the rows with a grid-stride loop.  If C sweeps the rows in the OPPOSITE direction, the rows it touches first are the
// ones A touched last: whatever the cache kept of V is read from it instead of from HBM.  Measured: the time of pass C after a
// pass A, forward against backward, plain against non-temporal loads, for configs[1]'s and the headline's row counts.
//   hipcc --offload-arch=gfx950 -O3 -o mall_probe mall_probe.hip && ./mall_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <bool NT> __device__ inline double ld1(const double *p) { return NT ? __builtin_nontemporal_load(p) : *p; }

// reads nv columns of V (and nw of W) for every row; REV: rows swept from the end
template <bool NT, bool REV>
__global__ void __launch_bounds__(256) sweep(const double *V, int nv, const double *W, int nw, long ld, long m, double *out) {
   double acc = 0.0;
   const long stride = (long)gridDim.x * 256;
   for (long i0 = (long)blockIdx.x * 256 + threadIdx.x; i0 < m; i0 += stride) {
      const long i = REV ? m - 1 - i0 : i0;
      for (int j = 0; j < nv; j++) acc += ld1<NT>(V + j * ld + i);
      for (int j = 0; j < nw; j++) acc += ld1<NT>(W + j * ld + i);
   }
   if (acc == 123.456) out[0] = acc;
}

template <bool NT>
void run(const double *V, const double *W, long m, int k, double *out, int grid) {
   hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
   float tf = 0, tb = 0, ta = 0;
   const int reps = 10;
   for (int mode = 0; mode < 2; mode++) {
      for (int r = 0; r < reps + 2; r++) {
         // pass A: V and W, forward
         hipEventRecord(a);
         hipLaunchKernelGGL((sweep<NT, false>), dim3(grid), dim3(256), 0, 0, V, k, W, k, m, m, out);
         hipEventRecord(b); hipEventSynchronize(b);
         float ms; hipEventElapsedTime(&ms, a, b);
         if (r >= 2 && mode == 0) ta += ms;
         // pass C: V only, forward (mode 0) or backward (mode 1)
         hipEventRecord(a);
         if (mode == 0) hipLaunchKernelGGL((sweep<NT, false>), dim3(grid), dim3(256), 0, 0, V, k, W, 0, m, m, out);
         else hipLaunchKernelGGL((sweep<NT, true>), dim3(grid), dim3(256), 0, 0, V, k, W, 0, m, m, out);
         hipEventRecord(b); hipEventSynchronize(b);
         hipEventElapsedTime(&ms, a, b);
         if (r >= 2) (mode == 0 ? tf : tb) += ms;
      }
   }
   const double gbA = 2.0 * k * m * 8 / 1e9, gbC = 1.0 * k * m * 8 / 1e9;
   printf("  %s loads: pass A %.1f us (%.0f GB/s) | pass C forward %.1f us (%.0f GB/s) | pass C backward %.1f us (%.0f GB/s)\n", NT ? "non-temporal" : "plain       ",
         1e3 * ta / reps, gbA / (1e-3 * ta / reps), 1e3 * tf / reps, gbC / (1e-3 * tf / reps), 1e3 * tb / reps, gbC / (1e-3 * tb / reps));
}

int main() {
   double *out;
   CHECK(hipMalloc(&out, 64));
   const long ms[3] = {2000250, 10001406, 1000000};
   for (int t = 0; t < 3; t++) {
      const long m = ms[t];
      for (int k : {6, 10, 15}) {
         double *V, *W;
         CHECK(hipMalloc(&V, sizeof(double) * m * k)); CHECK(hipMalloc(&W, sizeof(double) * m * k));
         CHECK(hipMemset(V, 0, sizeof(double) * m * k)); CHECK(hipMemset(W, 0, sizeof(double) * m * k));
         printf("m = %ld rows, k = %d columns: V = W = %.0f MB\n", m, k, m * k * 8 / 1e6);
         run<false>(V, W, m, k, out, 2048);
         run<true>(V, W, m, k, out, 2048);
         CHECK(hipFree(V)); CHECK(hipFree(W));
      }
   }
   return 0;
}
