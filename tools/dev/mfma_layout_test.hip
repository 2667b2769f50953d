// Dev check (GPU box): wave_matmul_64 (me_dense_mfma.h) against a host reference, with an asymmetric matrix.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/dev/mfma_layout_test.hip -o /tmp/mfma_test && /tmp/mfma_test
#include <cmath>
#include <cstdio>
#include <vector>

#include "../../metropolisengine_amd/csrc/me_dense_mfma.h"

using namespace me;

__global__ void k_test(const float *m, const float *x, float *y, int lower) {
  __shared__ float frag[4096];
  stage_a_fragments(frag, m);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  if (threadIdx.x >= 64) return;
  float v[64], out[64];
  for (int d = 0; d < 64; ++d) v[d] = x[d * 64 + lane];
  wave_matmul_64(frag, [&](int b, int t, float (&q)[4]) { if (t < 4) q[t] = v[4 * b + t]; }, out, lane, lower != 0);
  for (int d = 0; d < 64; ++d) y[d * 64 + lane] = out[d];
}

int main() {
  std::vector<float> m(4096), x(4096), y(4096);
  for (int lower = 0; lower < 2; ++lower) {
    for (int i = 0; i < 64; ++i)
      for (int j = 0; j < 64; ++j) {
        m[i * 64 + j] = (lower && j > i) ? 0.0f : (float)((i * 7 + j * 3) % 11) - 5.0f + 0.01f * i;
        x[i * 64 + j] = (float)((i * 5 + j * 13) % 17) - 8.0f + 0.001f * j;
      }
    float *dm, *dx, *dy;
    hipMalloc(&dm, 16384); hipMalloc(&dx, 16384); hipMalloc(&dy, 16384);
    hipMemcpy(dm, m.data(), 16384, hipMemcpyHostToDevice);
    hipMemcpy(dx, x.data(), 16384, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_test, dim3(1), dim3(256), 0, 0, dm, dx, dy, lower);
    hipMemcpy(y.data(), dy, 16384, hipMemcpyDeviceToHost);
    double worst = 0; int bad = 0;
    for (int i = 0; i < 64; ++i)
      for (int c = 0; c < 64; ++c) {
        double ref = 0;
        for (int k = 0; k < 64; ++k) ref += (double)m[i * 64 + k] * x[k * 64 + c];
        const double err = std::fabs(ref - y[i * 64 + c]);
        if (err > 1e-2) { if (bad < 5) printf("  mismatch row %d chain %d: got %g want %g\n", i, c, y[i * 64 + c], ref); ++bad; }
        if (err > worst) worst = err;
      }
    printf("lower=%d  max abs err %.3g  mismatches %d\n", lower, worst, bad);
    for (int rb = 0; rb < 16; ++rb) {
      printf("  rows %2d-%2d:", rb * 4, rb * 4 + 3);
      for (int cb = 0; cb < 2; ++cb) {
        int ok = 0;
        for (int i = rb * 4; i < rb * 4 + 4; ++i)
          for (int c = cb * 32; c < cb * 32 + 32; ++c) {
            double ref = 0;
            for (int k = 0; k < 64; ++k) ref += (double)m[i * 64 + k] * x[k * 64 + c];
            ok += std::fabs(ref - y[i * 64 + c]) <= 1e-2;
          }
        printf(" chains %2d-%2d ok %3d/128;", cb * 32, cb * 32 + 31, ok);
      }
      printf("\n");
    }
  }
  return 0;
}
