// Dev probe (GPU box): why does "136 read-only rows + 18 RMW rows per chain" (k_step with per-chain factors) take 150 us
// when a plain read of the same 570 MB takes 89 us and the 18 RMW rows 20 us?  Vary block size, grid (persistent
// grid-stride loop against one block per tile), layout and the presence of the RMW rows.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int RO, int RW, int THREADS, bool TILE, bool NT = false, bool NTW = false>
__global__ void __launch_bounds__(THREADS) k(const float *__restrict__ ro, float *rw, long long n) {
  for (long long c = (long long)blockIdx.x * THREADS + threadIdx.x; c < n; c += (long long)gridDim.x * THREADS) {
    const float *p = TILE ? ro + (c >> 6) * (long long)RO * 64 + (c & 63) : ro + c;
    const long long step = TILE ? 64 : n;
    float s = 0.f;
    float v[RO > 0 ? RO : 1];
#pragma unroll
    for (int u = 0; u < RO; ++u) v[u] = NT ? __builtin_nontemporal_load(p + u * step) : p[u * step];
#pragma unroll
    for (int u = 0; u < RO; ++u) s += v[u];
    if constexpr (RW > 0) {
      float x[RW];
#pragma unroll
      for (int r = 0; r < RW; ++r) x[r] = rw[r * n + c];
#pragma unroll
      for (int r = 0; r < RW; ++r) rw[r * n + c] = x[r] + s * 1e-30f;
    } else {
      if (s == 123.456f) rw[c] = s;
    }
  }
}

template <class F>
float time_it(F &&launch) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) launch();
  (void)hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) launch();
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms / 20 * 1e3f;
}

int main() {
  const long long n = 1 << 20;
  constexpr int RO = 136, RW = 18;
  float *ro, *rw;
  (void)hipMalloc(&ro, sizeof(float) * n * RO); (void)hipMalloc(&rw, sizeof(float) * n * RW);
  (void)hipMemset(ro, 0, sizeof(float) * n * RO); (void)hipMemset(rw, 0, sizeof(float) * n * RW);
#define RUN(name, RO_, RW_, T_, TILE_, blocks)                                                                       \
  {                                                                                                                  \
    const float us = time_it([&] { hipLaunchKernelGGL((k<RO_, RW_, T_, TILE_>), dim3(blocks), dim3(T_), 0, 0, ro, rw, n); }); \
    printf("%-58s %7.1f us  %5.0f GB/s\n", name, us, (double)n * (RO_ * 4 + RW_ * 8) / us / 1e3);                    \
  }
  RUN("RO only, comp-major, 64-thr blocks, 1 block/tile", 136, 0, 64, false, n / 64)
  RUN("RO only, tile-major, 64-thr blocks, 1 block/tile", 136, 0, 64, true, n / 64)
  RUN("RO only, comp-major, 64-thr, 4096 persistent blocks", 136, 0, 64, false, 4096)
  RUN("RO only, tile-major, 64-thr, 4096 persistent blocks", 136, 0, 64, true, 4096)
  RUN("RO only, comp-major, 256-thr, 1024 persistent blocks", 136, 0, 256, false, 1024)
  RUN("RO only, tile-major, 256-thr, 1024 persistent blocks", 136, 0, 256, true, 1024)
  RUN("RO only, 34 rows comp-major (x4 chains), 64-thr", 34, 0, 64, false, n / 64)
  RUN("RO+RW, comp-major, 64-thr, 1 block/tile", 136, 18, 64, false, n / 64)
  RUN("RO+RW, comp-major, 64-thr, 4096 persistent", 136, 18, 64, false, 4096)
  RUN("RO+RW, tile-major, 64-thr, 4096 persistent", 136, 18, 64, true, 4096)
  RUN("RO+RW, comp-major, 256-thr, 1024 persistent", 136, 18, 256, false, 1024)
  RUN("RW only, 64-thr, 1 block/tile", 0, 18, 64, false, n / 64)
#define RUNNT(name, TILE_)                                                                                           \
  {                                                                                                                  \
    const float us = time_it([&] { hipLaunchKernelGGL((k<136, 18, 64, TILE_, true>), dim3(n / 64), dim3(64), 0, 0, ro, rw, n); }); \
    printf("%-58s %7.1f us  %5.0f GB/s\n", name, us, (double)n * (136 * 4 + 18 * 8) / us / 1e3);                    \
  }
  RUNNT("RO(nontemporal)+RW, comp-major, 64-thr, 1 block/tile", false)
  RUNNT("RO(nontemporal)+RW, tile-major, 64-thr, 1 block/tile", true)
  {
    const float us = time_it([&] { hipLaunchKernelGGL((k<136, 0, 64, false, true>), dim3(n / 64), dim3(64), 0, 0, ro, rw, n); });
    printf("%-58s %7.1f us  %5.0f GB/s\n", "RO(nontemporal) only, comp-major", us, (double)n * 136 * 4 / us / 1e3);
  }
  return 0;
}
