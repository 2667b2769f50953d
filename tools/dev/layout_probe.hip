// Dev probe (GPU box): in-place read-modify-write of R rows per chain, component-major (row stride = n) against
// tile-major ([tile of 64 chains][row][lane]) addressing.  Does the number of concurrent row streams cost bandwidth?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/dev/layout_probe.hip -o /tmp/layout_probe && /tmp/layout_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

template <int R, bool TILE>
__global__ void __launch_bounds__(64) k_rmw(float *f, long long n, float a) {
  const long long stride = (long long)gridDim.x * 64;
  for (long long c = (long long)blockIdx.x * 64 + threadIdx.x; c < n; c += stride) {
    float *p = TILE ? f + (c >> 6) * (long long)R * 64 + (c & 63) : f + c;
    const long long step = TILE ? 64 : n;
#pragma unroll 16
    for (int r = 0; r < R; ++r) {
      p[0] = p[0] * a + 1.0f;
      p += step;
    }
  }
}

// the k_step pattern: all rows of a chain loaded first, then all stored
template <int R, bool TILE>
__global__ void __launch_bounds__(64) k_load_then_store(float *f, long long n, float a) {
  const long long stride = (long long)gridDim.x * 64;
  for (long long c = (long long)blockIdx.x * 64 + threadIdx.x; c < n; c += stride) {
    float *p = TILE ? f + (c >> 6) * (long long)R * 64 + (c & 63) : f + c;
    const long long step = TILE ? 64 : n;
    float v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = p[r * step];
#pragma unroll
    for (int r = 0; r < R; ++r) p[r * step] = v[r] * a + 1.0f;
  }
}

// 16-byte accesses: [tile][row/4][lane][4] ("packets" of four rows of one chain adjacent in memory)
template <int R4>
__global__ void __launch_bounds__(64) k_load_then_store_x4(float4 *f, long long n, float a) {
  const long long stride = (long long)gridDim.x * 64;
  for (long long c = (long long)blockIdx.x * 64 + threadIdx.x; c < n; c += stride) {
    float4 *p = f + (c >> 6) * (long long)R4 * 64 + (c & 63);
    float4 v[R4];
#pragma unroll
    for (int r = 0; r < R4; ++r) v[r] = p[r * 64];
#pragma unroll
    for (int r = 0; r < R4; ++r) {
      v[r].x = v[r].x * a + 1.0f; v[r].y = v[r].y * a + 1.0f; v[r].z = v[r].z * a + 1.0f; v[r].w = v[r].w * a + 1.0f;
      p[r * 64] = v[r];
    }
  }
}

template <int R4>
void run_x4(long long n) {
  float4 *f;
  if (hipMalloc(&f, sizeof(float4) * n * R4) != hipSuccess) { printf("alloc failed\n"); return; }
  (void)hipMemset(f, 0, sizeof(float4) * n * R4);
  const int grid = (int)((n + 63) / 64);
  hipEvent_t t0, t1;
  (void)hipEventCreate(&t0);
  (void)hipEventCreate(&t1);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k_load_then_store_x4<R4>), dim3(grid), dim3(64), 0, 0, f, n, 0.5f);
  (void)hipEventRecord(t0);
  const int reps = 200;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_load_then_store_x4<R4>), dim3(grid), dim3(64), 0, 0, f, n, 0.5f);
  (void)hipEventRecord(t1);
  (void)hipEventSynchronize(t1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, t0, t1);
  ms /= reps;
  printf("load-then-store R=%3d n=2^%d %-15s %8.4f ms  %6.0f GB/s\n", 4 * R4, (int)__builtin_ctzll(n), "tile-major x4", ms,
         32.0 * n * R4 / ms / 1e6);
  (void)hipFree(f);
}

template <int R, bool TILE>
void run_ls(long long n) {
  float *f;
  if (hipMalloc(&f, sizeof(float) * n * R) != hipSuccess) { printf("alloc failed\n"); return; }
  (void)hipMemset(f, 0, sizeof(float) * n * R);
  const int grid = (int)((n + 63) / 64);
  hipEvent_t t0, t1;
  (void)hipEventCreate(&t0);
  (void)hipEventCreate(&t1);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k_load_then_store<R, TILE>), dim3(grid), dim3(64), 0, 0, f, n, 0.5f);
  (void)hipEventRecord(t0);
  const int reps = 200;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_load_then_store<R, TILE>), dim3(grid), dim3(64), 0, 0, f, n, 0.5f);
  (void)hipEventRecord(t1);
  (void)hipEventSynchronize(t1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, t0, t1);
  ms /= reps;
  printf("load-then-store R=%3d n=2^%d %-15s %8.4f ms  %6.0f GB/s\n", R, (int)__builtin_ctzll(n),
         TILE ? "tile-major" : "component-major", ms, 8.0 * n * R / ms / 1e6);
  (void)hipFree(f);
}

template <int R, bool TILE>
void run(long long n) {
  float *f;
  if (hipMalloc(&f, sizeof(float) * n * R) != hipSuccess) { printf("alloc failed\n"); return; }
  (void)hipMemset(f, 0, sizeof(float) * n * R);
  const int grid = (int)((n + 63) / 64);
  hipEvent_t t0, t1;
  (void)hipEventCreate(&t0);
  (void)hipEventCreate(&t1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_rmw<R, TILE>), dim3(grid), dim3(64), 0, 0, f, n, 0.5f);
  (void)hipEventRecord(t0);
  const int reps = 10;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_rmw<R, TILE>), dim3(grid), dim3(64), 0, 0, f, n, 0.5f);
  (void)hipEventRecord(t1);
  (void)hipEventSynchronize(t1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, t0, t1);
  ms /= reps;
  printf("R=%4d n=2^%d %-15s %8.3f ms  %6.0f GB/s\n", R, (int)__builtin_ctzll(n), TILE ? "tile-major" : "component-major", ms,
         8.0 * n * R / ms / 1e6);
  (void)hipFree(f);
}

int main() {
  run_ls<20, false>(1 << 20);
  run_ls<20, true>(1 << 20);
  run_x4<5>(1 << 20);
  run_x4<5>(1 << 22);
  run_ls<20, true>(1 << 22);
  run_ls<18, false>(1 << 20);
  run_ls<18, true>(1 << 20);
  run_ls<18, false>(1 << 21);
  run_ls<18, true>(1 << 21);
  run_ls<18, false>(1 << 22);
  run_ls<18, true>(1 << 22);
  if (getenv("LAYOUT_PROBE_ALL")) {
  run<18, false>(1 << 22);
  run<18, true>(1 << 22);
  run<200, false>(1 << 20);
  run<200, true>(1 << 20);
  run<2080, false>(1 << 17);
  run<2080, true>(1 << 17);
  run<2080, false>(1 << 19);
  run<2080, true>(1 << 19);
  }
  return 0;
}
