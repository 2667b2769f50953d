// Dev probe (GPU box): would a TILE-major state help the 64-parameter float64 kernel?  Memory-only read-modify-write of
// x[64] + energy + width for 2^19 chains (554 MB), in the geometry of k_step_dense64_f64 (persistent 512-thread
// workgroups, a wavefront owns 32 chains, lanes 0-31 rows 0-31, lanes 32-63 rows 32-63, next tile prefetched):
//   component-major  x[row][chain]              -- the engine's layout: 64 rows, 4 MiB apart
//   tile-major       x[tile of 32][row][32]     -- a tile is one contiguous 16 KiB block
// and, for reference, the plain one-lane-per-chain pattern with 66 rows (64-thread blocks, no persistence).
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int D = 64, H = 32, TC = 32;

template <bool TILE>
__global__ void __launch_bounds__(512, 2) k_dense_geometry(double *x, double *e, double *w, long long n) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, cl = lane & 31, half = lane >> 5;
  const long long n_tiles = n / TC, stride = (long long)gridDim.x * 8;
  long long tile = (long long)blockIdx.x * 8 + wave;
  auto addr = [&](long long t, int row) -> double * {
    return TILE ? x + (t * D + row) * TC + cl : x + (long long)row * n + t * TC + cl;
  };
  double v[H], ev = 0, wv = 0;
  if (tile < n_tiles) {
#pragma unroll
    for (int i = 0; i < H; ++i) v[i] = *addr(tile, H * half + i);
    ev = e[tile * TC + cl]; wv = w[tile * TC + cl];
  }
  while (tile < n_tiles) {
    const long long next = tile + stride;
    double vn[H], en = 0, wn = 0;
    if (next < n_tiles) {
#pragma unroll
      for (int i = 0; i < H; ++i) vn[i] = *addr(next, H * half + i);
      en = e[next * TC + cl]; wn = w[next * TC + cl];
    }
#pragma unroll
    for (int i = 0; i < H; ++i) *addr(tile, H * half + i) = v[i] * 1.0000001 + 1e-30;
    if (half == 0) { e[tile * TC + cl] = ev + 1e-30; w[tile * TC + cl] = wv + 1e-30; }
#pragma unroll
    for (int i = 0; i < H; ++i) v[i] = vn[i];
    ev = en; wv = wn; tile = next;
  }
}

__global__ void __launch_bounds__(64) k_plain(double *rw, long long n) {
  const long long c = (long long)blockIdx.x * 64 + threadIdx.x;
  if (c >= n) return;
  double x[D + 2];
#pragma unroll
  for (int r = 0; r < D + 2; ++r) x[r] = rw[r * n + c];
#pragma unroll
  for (int r = 0; r < D + 2; ++r) rw[r * n + c] = x[r] * 1.0000001 + 1e-30;
}

template <class F>
float time_it(F &&launch) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) launch();
  (void)hipEventRecord(e0);
  for (int i = 0; i < 30; ++i) launch();
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms / 30 * 1e3f;
}

int main() {
  const long long n = 1 << 19;
  double *x, *e, *w;
  (void)hipMalloc(&x, 8 * n * (D + 2)); (void)hipMalloc(&e, 8 * n); (void)hipMalloc(&w, 8 * n);
  (void)hipMemset(x, 0, 8 * n * (D + 2)); (void)hipMemset(e, 0, 8 * n); (void)hipMemset(w, 0, 8 * n);
  const double bytes = 16.0 * n * (D + 2);
  for (int blocks : {256, 512}) {
    float us = time_it([&] { hipLaunchKernelGGL(k_dense_geometry<false>, dim3(blocks), dim3(512), 0, 0, x, e, w, n); });
    printf("dense-64 geometry, component-major, %d workgroups: %7.1f us  %5.0f GB/s\n", blocks, us, bytes / us / 1e3);
    us = time_it([&] { hipLaunchKernelGGL(k_dense_geometry<true>, dim3(blocks), dim3(512), 0, 0, x, e, w, n); });
    printf("dense-64 geometry, tile-major,      %d workgroups: %7.1f us  %5.0f GB/s\n", blocks, us, bytes / us / 1e3);
  }
  const float us = time_it([&] { hipLaunchKernelGGL(k_plain, dim3((unsigned)(n / 64)), dim3(64), 0, 0, x, n); });
  printf("one lane per chain, 66 rows component-major, 64-thread blocks: %7.1f us  %5.0f GB/s\n", us, bytes / us / 1e3);
  return 0;
}
