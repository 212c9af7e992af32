// Single-vector CSR SpMV probe (NOT part of the product).  Result on one MI355X (profiles/r04_spmv_pipeline_probe.txt): A 149 us,
// B 324-326 us, bitwise equal — the in-workgroup pipeline loses 2.2x.
//
// Round 4 found the product's csr_stream_kernel unmoved by 70 % fewer matrix bytes (profiles/r04_spmv_value_table_experiment.txt):
// it is bound by the chain of dependent round trips of a tile (tile record -> value / index batch -> gather -> LDS products ->
// row sums), one tile per workgroup.  This probe compares, on the headline matrix (2-D 5-pt Laplacian, CSR, 2-byte index stream
// relative to the tile's first row like the product's col16):
//   A  one tile per workgroup (the product's structure, restated minimally);
//   B  persistent workgroups that walk tiles t, t + G, t + 2G, ... and keep the NEXT tile's value / index batch in flight
//      while the current one is gathered, multiplied and summed (records two tiles ahead, batches one tile ahead).
// Both must give bit-identical y.  Build: hipcc --offload-arch=gfx950 -O3 scripts/probes/spmv_pipeline_probe.hip -o spmv_probe
// Run:   ./spmv_probe [nx ny]      (default 3162 3163)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define TILE_NNZ 2048
#define TILE_ROWS 256
#define PER_LANE (TILE_NNZ / 256)

struct Tile { int r0, r1, p0, p1; };

__device__ __forceinline__ void tile_body(const Tile ti, const double (&v)[PER_LANE], const int32_t (&ci)[PER_LANE], const int32_t *__restrict__ rowptr,
      const double *__restrict__ x, double *__restrict__ y, double *prod) {
   const int nz = ti.p1 - ti.p0;
   const int r = ti.r0 + (int)threadIdx.x;
   const int rc = r < ti.r1 ? r : ti.r1 - 1;
   const int sa = rowptr[rc] - ti.p0, sb = rowptr[rc + 1] - ti.p0;
   double xg[PER_LANE];
#pragma unroll
   for (int u = 0; u < PER_LANE; u++) xg[u] = x[ci[u]];
#pragma unroll
   for (int u = 0; u < PER_LANE; u++) { const int q = (int)threadIdx.x + u * 256; if (q < nz) prod[q] = v[u] * xg[u]; }
   __syncthreads();
   if (r < ti.r1) { double s = 0.0; for (int q = sa; q < sb; q++) s += prod[q]; y[r] = s; }
   __syncthreads();
}
__device__ __forceinline__ void tile_batch(const Tile ti, const double *__restrict__ val, const uint16_t *__restrict__ c16, int64_t c16off,
      double (&v)[PER_LANE], int32_t (&ci)[PER_LANE]) {
   const int nz = ti.p1 - ti.p0;
#pragma unroll
   for (int u = 0; u < PER_LANE; u++) {
      const int q = (int)threadIdx.x + u * 256;
      const int qc = q < nz ? q : (nz > 0 ? nz - 1 : 0);
      v[u] = __builtin_nontemporal_load(val + ti.p0 + qc);
      ci[u] = (int32_t)__builtin_nontemporal_load(c16 + ti.p0 + qc);
   }
   const int32_t base = (int32_t)(c16off + ti.r0);
#pragma unroll
   for (int u = 0; u < PER_LANE; u++) ci[u] += base;
}

// A: one tile per workgroup
__global__ void __launch_bounds__(256) spmv_tile(const Tile *__restrict__ tiles, int ntiles, const int32_t *__restrict__ rowptr, const double *__restrict__ val,
      const uint16_t *__restrict__ c16, int64_t c16off, const double *__restrict__ x, double *__restrict__ y) {
   __shared__ double prod[TILE_NNZ];
   const int t = blockIdx.x;
   if (t >= ntiles) return;
   const Tile ti = tiles[t];
   double v[PER_LANE]; int32_t ci[PER_LANE];
   tile_batch(ti, val, c16, c16off, v, ci);
   tile_body(ti, v, ci, rowptr, x, y, prod);
}

// B: persistent, next tile's batch in flight
__global__ void __launch_bounds__(256) spmv_pipe(const Tile *__restrict__ tiles, int ntiles, const int32_t *__restrict__ rowptr, const double *__restrict__ val,
      const uint16_t *__restrict__ c16, int64_t c16off, const double *__restrict__ x, double *__restrict__ y) {
   __shared__ double prod[TILE_NNZ];
   const int G = gridDim.x;
   int t = blockIdx.x;
   if (t >= ntiles) return;
   Tile cur = tiles[t];
   Tile nxt = (t + G < ntiles) ? tiles[t + G] : cur;
   double v[PER_LANE]; int32_t ci[PER_LANE];
   tile_batch(cur, val, c16, c16off, v, ci);
   for (; t < ntiles; t += G) {
      const bool more = t + G < ntiles;
      double vn[PER_LANE]; int32_t cn[PER_LANE];
      Tile nn = nxt;
      if (more) {
         tile_batch(nxt, val, c16, c16off, vn, cn);                    // in flight while the current tile is processed
         nn = (t + 2 * G < ntiles) ? tiles[t + 2 * G] : nxt;           // record two tiles ahead
      }
      tile_body(cur, v, ci, rowptr, x, y, prod);
      if (more) {
#pragma unroll
         for (int u = 0; u < PER_LANE; u++) { v[u] = vn[u]; ci[u] = cn[u]; }
         cur = nxt; nxt = nn;
      }
   }
}

int main(int argc, char **argv) {
   const int nx = argc > 2 ? atoi(argv[1]) : 3162, ny = argc > 2 ? atoi(argv[2]) : 3163;
   const int64_t n = (int64_t)nx * ny;
   std::vector<int32_t> rp(n + 1), ci; std::vector<double> va;
   ci.reserve(5 * n); va.reserve(5 * n);
   for (int64_t i = 0; i < n; i++) {
      const int ix = (int)(i % nx), iy = (int)(i / nx);
      rp[i] = (int32_t)ci.size();
      if (iy > 0) { ci.push_back((int32_t)(i - nx)); va.push_back(-1.0); }
      if (ix > 0) { ci.push_back((int32_t)(i - 1)); va.push_back(-1.0); }
      ci.push_back((int32_t)i); va.push_back(4.0);
      if (ix < nx - 1) { ci.push_back((int32_t)(i + 1)); va.push_back(-1.0); }
      if (iy < ny - 1) { ci.push_back((int32_t)(i + nx)); va.push_back(-1.0); }
   }
   rp[n] = (int32_t)ci.size();
   const int64_t nnz = ci.size();
   std::vector<Tile> tiles;
   for (int64_t r = 0; r < n;) {
      int64_t e = r; int cnt = 0;
      while (e < n && e - r < TILE_ROWS && cnt + (rp[e + 1] - rp[e]) <= TILE_NNZ) { cnt += rp[e + 1] - rp[e]; e++; }
      tiles.push_back({(int)r, (int)e, rp[r], rp[e]});
      r = e;
   }
   int64_t back = 0;
   for (auto &t : tiles) for (int q = t.p0; q < t.p1; q++) if (t.r0 - ci[q] > back) back = t.r0 - ci[q];
   std::vector<uint16_t> c16(nnz + 1, 0);
   for (auto &t : tiles) for (int q = t.p0; q < t.p1; q++) { const int64_t d = ci[q] - (t.r0 - back); if (d < 0 || d > 65535) { printf("pattern does not fit 16 bits\n"); return 1; } c16[q] = (uint16_t)d; }
   std::vector<double> x(n); for (int64_t i = 0; i < n; i++) x[i] = 1.0 + 1e-3 * (double)(i % 997);
   printf("n %lld nnz %lld tiles %zu back %lld\n", (long long)n, (long long)nnz, tiles.size(), (long long)back);

   Tile *dt; int32_t *drp; double *dv, *dx, *dy1, *dy2; uint16_t *dc;
   CHECK(hipMalloc(&dt, tiles.size() * sizeof(Tile))); CHECK(hipMalloc(&drp, (n + 1) * 4)); CHECK(hipMalloc(&dv, (nnz + 1) * 8));
   CHECK(hipMalloc(&dc, (nnz + 1) * 2)); CHECK(hipMalloc(&dx, n * 8)); CHECK(hipMalloc(&dy1, n * 8)); CHECK(hipMalloc(&dy2, n * 8));
   va.push_back(0.0);
   CHECK(hipMemcpy(dt, tiles.data(), tiles.size() * sizeof(Tile), hipMemcpyHostToDevice)); CHECK(hipMemcpy(drp, rp.data(), (n + 1) * 4, hipMemcpyHostToDevice));
   CHECK(hipMemcpy(dv, va.data(), (nnz + 1) * 8, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dc, c16.data(), (nnz + 1) * 2, hipMemcpyHostToDevice));
   CHECK(hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice));
   const int nt = (int)tiles.size();
   hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
   hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
   auto time_it = [&](const char *name, auto launch) {
      for (int w = 0; w < 3; w++) launch();
      CHECK(hipEventRecord(a)); const int reps = 20; for (int r = 0; r < reps; r++) launch();
      CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b)); float ms; CHECK(hipEventElapsedTime(&ms, a, b));
      const double us = 1e3 * ms / reps, bytes = (double)nnz * 10 + (double)(n + 1) * 4 + 2.0 * n * 8;
      printf("%-40s %8.1f us  %7.0f GB/s (10 B per nonzero + vectors)\n", name, us, bytes / us / 1e3);
   };
   time_it("A one tile per workgroup", [&] { hipLaunchKernelGGL(spmv_tile, dim3(nt), dim3(256), 0, 0, dt, nt, drp, dv, dc, -back, dx, dy1); });
   for (int per_cu : {4, 6, 8}) {
      char nm[64]; snprintf(nm, sizeof nm, "B persistent, %d workgroups per CU", per_cu);
      const int G = prop.multiProcessorCount * per_cu < nt ? prop.multiProcessorCount * per_cu : nt;
      time_it(nm, [&] { hipLaunchKernelGGL(spmv_pipe, dim3(G), dim3(256), 0, 0, dt, nt, drp, dv, dc, -back, dx, dy2); });
   }
   std::vector<double> y1(n), y2(n);
   CHECK(hipMemcpy(y1.data(), dy1, n * 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(y2.data(), dy2, n * 8, hipMemcpyDeviceToHost));
   printf("A == B bitwise: %s\n", memcmp(y1.data(), y2.data(), n * 8) == 0 ? "yes" : "NO");
   return 0;
}
