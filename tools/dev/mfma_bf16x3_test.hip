// Dev check (GPU box): wave_matmul_64_bf16x3 (me_dense_bf16x3.h) against a float64 host reference, beside the fp32
// MFMA version, on a random asymmetric matrix.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/dev/mfma_bf16x3_test.hip -o /tmp/bf16x3_test && /tmp/bf16x3_test
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../metropolisengine_amd/csrc/me_dense_bf16x3.h"

using namespace me;

__global__ void k_test(const unsigned int *frag_src, const float *m, const float *x, float *y, float *y32) {
  __shared__ unsigned int frag[kBf16FragWords];
  __shared__ float frag32[4096];
  for (int i = threadIdx.x; i < kBf16FragWords; i += blockDim.x) frag[i] = frag_src[i];
  stage_a_fragments(frag32, m);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  if (threadIdx.x >= 64) return;
  float v[64], out[64];
  for (int d = 0; d < 64; ++d) v[d] = x[d * 64 + lane];
  wave_matmul_64_bf16x3(frag, frag, [&](int k) { return v[k]; }, [&](int row, float val) { out[row] = val; }, lane);
  for (int d = 0; d < 64; ++d) y[d * 64 + lane] = out[d];
  wave_matmul_64(frag32, [&](int b, int t, float (&q)[4]) { if (t < 4) q[t] = v[4 * b + t]; }, out, lane, false);
  for (int d = 0; d < 64; ++d) y32[d * 64 + lane] = out[d];
}

static void host_fragments(const std::vector<float> &m, std::vector<unsigned short> &frag) {
  frag.assign(3 * 2 * 4 * 64 * 8, 0);
  for (int i = 0; i < 64; ++i)
    for (int k = 0; k < 64; ++k) {
      float rest = m[i * 64 + k];
      for (int q = 0; q < 3; ++q) {
        unsigned int bits;
        std::memcpy(&bits, &rest, 4);
        bits &= 0xFFFF0000u;
        float head;
        std::memcpy(&head, &bits, 4);
        rest -= head;
        const int mb = i >> 5, r = i & 31, s = k >> 4, h = (k >> 3) & 1, j = k & 7, lane = 32 * h + r;
        frag[((((q * 2 + mb) * 4 + s) * 64 + lane) << 3) + j] = (unsigned short)(bits >> 16);
      }
    }
}

int main() {
  std::vector<float> m(4096), x(4096), y(4096), y32(4096);
  srand(7);
  for (int trial = 0; trial < 3; ++trial) {
    const float scale = trial == 0 ? 1.0f : (trial == 1 ? 1e-3f : 300.0f);
    for (int i = 0; i < 4096; ++i) {
      m[i] = scale * (2.0f * rand() / RAND_MAX - 1.0f);
      x[i] = (2.0f * rand() / RAND_MAX - 1.0f) / scale * (1.0f + (i % 7));
    }
    std::vector<unsigned short> frag;
    host_fragments(m, frag);
    unsigned int *dfrag;
    float *dm, *dx, *dy, *dy32;
    hipMalloc(&dfrag, frag.size() * 2); hipMalloc(&dm, 16384); hipMalloc(&dx, 16384); hipMalloc(&dy, 16384); hipMalloc(&dy32, 16384);
    hipMemcpy(dfrag, frag.data(), frag.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dm, m.data(), 16384, hipMemcpyHostToDevice);
    hipMemcpy(dx, x.data(), 16384, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_test, dim3(1), dim3(256), 0, 0, dfrag, dm, dx, dy, dy32);
    hipMemcpy(y.data(), dy, 16384, hipMemcpyDeviceToHost);
    hipMemcpy(y32.data(), dy32, 16384, hipMemcpyDeviceToHost);
    double worst = 0, worst32 = 0, worst_seq = 0;
    int bad = 0;
    for (int i = 0; i < 64; ++i)
      for (int c = 0; c < 64; ++c) {
        double ref = 0, mag = 0;
        float seq = 0.0f;
        for (int k = 0; k < 64; ++k) {
          ref += (double)m[i * 64 + k] * x[k * 64 + c];
          mag += std::fabs((double)m[i * 64 + k] * x[k * 64 + c]);
          seq = std::fmaf(m[i * 64 + k], x[k * 64 + c], seq);
        }
        const double err = std::fabs(ref - y[i * 64 + c]) / mag, err32 = std::fabs(ref - y32[i * 64 + c]) / mag;
        if (err > 1e-4) { if (bad < 5) printf("  mismatch row %d chain %d: got %g want %g\n", i, c, y[i * 64 + c], ref); ++bad; }
        worst = std::fmax(worst, err);
        worst32 = std::fmax(worst32, err32);
        worst_seq = std::fmax(worst_seq, std::fabs(ref - seq) / mag);
      }
    printf("trial %d: max |err| / sum|terms|: bf16x3 %.3g   fp32 mfma %.3g   fp32 sequential fma %.3g   mismatches %d\n",
           trial, worst, worst32, worst_seq, bad);
  }
  return 0;
}
