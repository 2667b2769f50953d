// Dev probe (GPU box): the memory pattern of k_step with per-chain factors -- RO read-only rows (the packed factor)
// plus RW read-modify-write rows (state) per chain, one lane per chain, 64-thread blocks, 2^20 chains, float32 --
// without any arithmetic.  Which of {rows in flight per wave, row layout, access width} bounds it?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/dev/rows_probe.hip -o tools/variants/rows_probe
#include <hip/hip_runtime.h>
#include <cstdio>

template <int RO, int RW, int BATCH, bool TILE>
__global__ void __launch_bounds__(64) k(const float *__restrict__ ro, float *rw, long long n) {
  const long long c = (long long)blockIdx.x * 64 + threadIdx.x;
  if (c >= n) return;
  const float *p = TILE ? ro + (c >> 6) * (long long)RO * 64 + (c & 63) : ro + c;
  const long long step = TILE ? 64 : n;
  float s = 0.f;
#pragma unroll
  for (int k0 = 0; k0 < RO; k0 += BATCH) {
    float v[BATCH];
#pragma unroll
    for (int u = 0; u < BATCH; ++u) if (k0 + u < RO) v[u] = p[(k0 + u) * step];
#pragma unroll
    for (int u = 0; u < BATCH; ++u) if (k0 + u < RO) s += v[u];
    asm volatile("" : "+v"(s) :: "memory");
  }
  float x[RW];
#pragma unroll
  for (int r = 0; r < RW; ++r) x[r] = rw[r * n + c];
#pragma unroll
  for (int r = 0; r < RW; ++r) rw[r * n + c] = x[r] + s * 1e-30f;
}

// the same bytes with 16-byte accesses on a packet layout: [tile][RO/4][lane][4]
template <int RO4, int RW>
__global__ void __launch_bounds__(64) k4(const float4 *__restrict__ ro, float *rw, long long n) {
  const long long c = (long long)blockIdx.x * 64 + threadIdx.x;
  if (c >= n) return;
  const float4 *p = ro + (c >> 6) * (long long)RO4 * 64 + (c & 63);
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < RO4; ++k) { const float4 v = p[k * 64]; s += v.x + v.y + v.z + v.w; }
  float x[RW];
#pragma unroll
  for (int r = 0; r < RW; ++r) x[r] = rw[r * n + c];
#pragma unroll
  for (int r = 0; r < RW; ++r) rw[r * n + c] = x[r] + s * 1e-30f;
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
  const long long n = 1 << 20;
  constexpr int RO = 136, RW = 18;
  float *ro, *rw;
  (void)hipMalloc(&ro, sizeof(float) * n * RO); (void)hipMalloc(&rw, sizeof(float) * n * RW);
  (void)hipMemset(ro, 0, sizeof(float) * n * RO); (void)hipMemset(rw, 0, sizeof(float) * n * RW);
  const dim3 grid((unsigned)(n / 64)), block(64);
  const double bytes = (double)n * (RO * 4 + RW * 8);
  auto report = [&](const char *name, float us) { printf("%-46s %7.1f us  %5.0f GB/s\n", name, us, bytes / us / 1e3); };
  report("component-major, all 136 loads in flight", time_it([&] { hipLaunchKernelGGL((k<RO, RW, 136, false>), grid, block, 0, 0, ro, rw, n); }));
  report("component-major, batches of 32", time_it([&] { hipLaunchKernelGGL((k<RO, RW, 32, false>), grid, block, 0, 0, ro, rw, n); }));
  report("component-major, batches of 8", time_it([&] { hipLaunchKernelGGL((k<RO, RW, 8, false>), grid, block, 0, 0, ro, rw, n); }));
  report("tile-major factor, all in flight", time_it([&] { hipLaunchKernelGGL((k<RO, RW, 136, true>), grid, block, 0, 0, ro, rw, n); }));
  report("tile-major factor, batches of 32", time_it([&] { hipLaunchKernelGGL((k<RO, RW, 32, true>), grid, block, 0, 0, ro, rw, n); }));
  report("tile-major packets, 16-byte loads", time_it([&] { hipLaunchKernelGGL((k4<RO / 4, RW>), grid, block, 0, 0, (const float4 *)ro, rw, n); }));
  // the identity kernel's pattern for scale: 18 RMW rows only
  {
    const double b18 = (double)n * RW * 8;
    const float us = time_it([&] { hipLaunchKernelGGL((k<0, RW, 1, false>), grid, block, 0, 0, ro, rw, n); });
    printf("%-46s %7.1f us  %5.0f GB/s\n", "18 read-modify-write rows only", us, b18 / us / 1e3);
  }
  return 0;
}
