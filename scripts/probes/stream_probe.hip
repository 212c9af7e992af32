// Multi-column streaming-read probe: how fast can NC columns of a tall panel be read (and
// reduced) on MI355X as a function of access shape.  Used to set expectations for the panel
// kernels (profiles/r01_stream_probe.txt).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int NC, int VW, int UNROLL>
__global__ void __launch_bounds__(256) read_cols(const double *base, long ld, long m, double *out) {
   double acc = 0.0;
   const long mg = m / VW;
   const long stride = (long)gridDim.x * 256;
   for (long g = (long)blockIdx.x * 256 + threadIdx.x; g < mg; g += stride * UNROLL) {
#pragma unroll
      for (int u = 0; u < UNROLL; u++) {
         long gg = g + u * stride;
         if (gg < mg) {
#pragma unroll
            for (int j = 0; j < NC; j++) {
               if (VW == 2) { double2 v = ((const double2 *)(base + j * ld))[gg]; acc += v.x + v.y; }
               else acc += base[j * ld + gg];
            }
         }
      }
   }
   if (acc == 123.456) out[0] = acc;
}

template <int NC, int VW, int UNROLL>
double run(const double *base, long ld, long m, double *out, int grid) {
   hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
   for (int w = 0; w < 3; w++) hipLaunchKernelGGL((read_cols<NC, VW, UNROLL>), dim3(grid), dim3(256), 0, 0, base, ld, m, out);
   hipEventRecord(a);
   const int reps = 20;
   for (int r = 0; r < reps; r++) hipLaunchKernelGGL((read_cols<NC, VW, UNROLL>), dim3(grid), dim3(256), 0, 0, base, ld, m, out);
   hipEventRecord(b); hipEventSynchronize(b);
   float ms; hipEventElapsedTime(&ms, a, b);
   return (double)NC * m * 8 * reps / (ms * 1e-3) / 1e9;
}

int main() {
   const long m = 2000250, ld = m;
   const int ncols = 64;
   double *base, *out;
   CHECK(hipMalloc(&base, sizeof(double) * ld * ncols));
   CHECK(hipMalloc(&out, 64));
   CHECK(hipMemset(base, 0, sizeof(double) * ld * ncols));
   for (int grid : {512, 1024, 2048, 4096}) {
      printf("grid %5d | NC=8 vw1 u1 %7.0f | NC=16 vw1 u1 %7.0f | NC=32 vw1 u1 %7.0f | NC=16 vw2 u1 %7.0f | NC=32 vw2 u1 %7.0f | NC=16 vw1 u2 %7.0f | NC=32 vw1 u2 %7.0f | NC=64 vw1 u1 %7.0f GB/s\n", grid,
            run<8, 1, 1>(base, ld, m, out, grid), run<16, 1, 1>(base, ld, m, out, grid), run<32, 1, 1>(base, ld, m, out, grid),
            run<16, 2, 1>(base, ld, m, out, grid), run<32, 2, 1>(base, ld, m, out, grid),
            run<16, 1, 2>(base, ld, m, out, grid), run<32, 1, 2>(base, ld, m, out, grid), run<64, 1, 1>(base, ld, m, out, grid));
   }
   return 0;
}
