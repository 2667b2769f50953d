// Dev microbenchmark (GPU box): v_mfma_f64_16x16x4_f64 on gfx950 -- issue rate, overlap with float64 / integer VALU
// work on the same SIMD, and the operand / result lane maps checked against a host product with asymmetric data.
//   hipcc --offload-arch=gfx950 -O3 tools/dev/mfma_f64_probe.hip -o tools/variants/mfma_f64_probe
#include <cstdio>
#include <vector>
#include <hip/hip_runtime.h>
using f64x4 = __attribute__((ext_vector_type(4))) double;

template <int MODE>   // 1 = MFMA only, 2 = f64 VALU only, 4 = integer VALU only; sums = interleaved in one wave
__global__ void __launch_bounds__(256) k(double *out, int iters) {
  f64x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0}, acc3 = {0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  unsigned v0 = threadIdx.x, v1 = blockIdx.x, v2 = 3, v3 = 5;
  double f0 = a, f1 = b;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (MODE & 1) {
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, b, acc3, 0, 0, 0);
      }
      if (MODE & 2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {   // 32 dependent-ish f64 fma
          f0 = f0 * 1.0001 + f1; f1 = f1 * 0.9999 + f0;
        }
      }
      if (MODE & 4) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {   // 32 int VALU incl. the Philox multiply
          const unsigned long long p = (unsigned long long)0xD2511F53u * v0;
          v0 = (unsigned)(p >> 32) ^ v1 ^ v2; v1 = (unsigned)p + v3;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  double s = f0 + f1 + (double)(v0 + v1);
  for (int r = 0; r < 4; ++r) s += acc0[r] + acc1[r] + acc2[r] + acc3[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
float run(double *d, int blocks) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 100);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 2000);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

// D = A(16x4) B(4x16): lane l supplies A[l&15][l>>4] and B[l>>4][l&15]; which (row, col) does result reg r of lane l hold?
__global__ void layout(const double *A, const double *B, double *D) {
  const int l = threadIdx.x;
  f64x4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)], acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[l * 4 + r] = acc[r];
}

int main() {
  double *d; hipMalloc(&d, 2048 * 256 * 8);
  for (int blocks : {256, 512}) {   // 1 or 2 waves per SIMD
    const float m = run<1>(d, blocks), f = run<2>(d, blocks), i = run<4>(d, blocks);
    printf("blocks %d: mfma-only %.3f ms (%.1f cycles/mfma at 2.4 GHz), f64-valu-only %.3f, int-valu-only %.3f, mfma+f64valu %.3f, "
           "mfma+int %.3f, mfma+both %.3f, valu both %.3f\n", blocks, m, m * 1e-3 * 2.4e9 / (2000.0 * 16 * (blocks / 256)), f, i,
           run<3>(d, blocks), run<5>(d, blocks), run<7>(d, blocks), run<6>(d, blocks));
  }
  std::vector<double> A(64), B(64), D(256);
  for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) A[i * 4 + k] = 1 + i + 100 * k;
  for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = (k == 0 ? 1.0 : 0.0) * (1 + 1000 * j) + (k == 1 ? 0.001 * j : 0.0);
  double *dA, *dB, *dD; hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 2048);
  hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
    const int col = l & 15, row = (l >> 4) + 4 * r;      // guide: f64 C/D map
    double ref = 0; for (int k = 0; k < 4; ++k) ref += A[row * 4 + k] * B[k * 16 + col];
    if (ref != D[l * 4 + r]) ++bad;
  }
  printf("layout check (col = l&15, row = (l>>4) + 4 r): %s (%d mismatches)\n", bad ? "MISMATCH" : "ok", bad);
  return 0;
}
