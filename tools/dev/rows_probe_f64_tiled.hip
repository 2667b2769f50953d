// Dev probe (GPU box): would a tile-major layout help the HEADLINE kernel's state too?  18 float64 rows read-modify-write,
// one lane per chain, 64-thread blocks, 2^20 (cache-resident) and 2^22 (HBM) chains:
//   component-major  rw[row][chain]                     (the engine's layout below 32 degrees of freedom)
//   tile-major       rw[tile of 64][row][64]            (18 x 512 B = one contiguous 9 KiB block per wavefront)
//   x tile-major (16 rows) + energy / width component-major   (what making only x tile-major would give)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int RW, int MODE>
__global__ void __launch_bounds__(64) k(double *rw, double *ew, long long n) {
  const long long c = (long long)blockIdx.x * 64 + threadIdx.x;
  if (c >= n) return;
  double x[RW];
  auto at = [&](int r) -> double * {
    if (MODE == 0) return rw + (long long)r * n + c;
    if (MODE == 1) return rw + ((c >> 6) * RW + r) * 64 + (c & 63);
    return r < RW - 2 ? rw + ((c >> 6) * (RW - 2) + r) * 64 + (c & 63) : ew + (long long)(r - (RW - 2)) * n + c;
  };
#pragma unroll
  for (int r = 0; r < RW; ++r) x[r] = *at(r);
#pragma unroll
  for (int r = 0; r < RW; ++r) *at(r) = x[r] * 1.0000001 + 1e-30;
}
// 16-byte accesses: [tile of 64][row pair][64 lanes][2]: two rows of one chain are neighbours, 9 loads + 9 stores per lane
__global__ void __launch_bounds__(64) k_pairs(double2 *rw, long long n) {
  const long long c = (long long)blockIdx.x * 64 + threadIdx.x;
  if (c >= n) return;
  double2 *p = rw + (c >> 6) * 9 * 64 + (c & 63);
  double2 x[9];
#pragma unroll
  for (int r = 0; r < 9; ++r) x[r] = p[r * 64];
#pragma unroll
  for (int r = 0; r < 9; ++r) p[r * 64] = double2{x[r].x * 1.0000001 + 1e-30, x[r].y * 1.0000001 + 1e-30};
}
// ... and four rows of one chain as two 16-byte halves 1 KiB apart is what a b128 access of [tile][row quad][64][4] would be;
// the same bytes per instruction as k_pairs, half as many 512-byte row segments: not tried (18 rows do not divide by 4)
void run_pairs(int lg, double *rw) {
  const long long n = 1ll << lg;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_pairs, dim3((unsigned)(n / 64)), dim3(64), 0, 0, (double2 *)rw, n);
  (void)hipEventRecord(e0);
  for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k_pairs, dim3((unsigned)(n / 64)), dim3(64), 0, 0, (double2 *)rw, n);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s 2^%d chains: %7.1f us  %5.0f GB/s\n", "tile-major row PAIRS, 16-byte accesses", lg, ms * 10, 288.0 * n / (ms * 10) / 1e3);
}
template <int MODE>
void run(const char *name, int lg, double *rw, double *ew) {
  const long long n = 1ll << lg;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<18, MODE>), dim3((unsigned)(n / 64)), dim3(64), 0, 0, rw, ew, n);
  (void)hipEventRecord(e0);
  for (int i = 0; i < 100; ++i) hipLaunchKernelGGL((k<18, MODE>), dim3((unsigned)(n / 64)), dim3(64), 0, 0, rw, ew, n);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s 2^%d chains: %7.1f us  %5.0f GB/s\n", name, lg, ms * 10, 288.0 * n / (ms * 10) / 1e3);
}
int main() {
  for (int lg : {20, 22}) {
    const long long n = 1ll << lg;
    double *rw, *ew; (void)hipMalloc(&rw, 8 * n * 18); (void)hipMalloc(&ew, 8 * n * 2);
    (void)hipMemset(rw, 0, 8 * n * 18); (void)hipMemset(ew, 0, 8 * n * 2);
    for (int rep = 0; rep < 2; ++rep) {
      run<0>("component-major", lg, rw, ew);
      run<1>("tile-major (x, energy, width in one block)", lg, rw, ew);
      run<2>("x tile-major, energy / width separate", lg, rw, ew);
      run_pairs(lg, rw);
    }
    (void)hipFree(rw); (void)hipFree(ew);
  }
  return 0;
}
