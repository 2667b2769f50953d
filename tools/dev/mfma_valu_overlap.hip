// Dev microbenchmark (GPU box): do v_mfma_f32_32x32x2_f32 and fp32 / integer VALU work overlap on one SIMD?
#include <cstdio>
#include <hip/hip_runtime.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int MODE>   // 1 = MFMA only, 2 = VALU only, 3 = both interleaved (same wave)
__global__ void __launch_bounds__(256) k(float *out, int iters) {
  f32x16 acc0 = {0}, acc1 = {0};
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  unsigned v0 = threadIdx.x, v1 = blockIdx.x, v2 = 3, v3 = 5;
  float f0 = a, f1 = b;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (MODE & 1) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
      }
      if (MODE & 2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {   // 32 VALU: xor / add / fma mix
          v0 = (v0 ^ v1) + v2; v1 = (v1 ^ v3) + v0; f0 = f0 * 1.0001f + f1; f1 = f1 * 0.9999f + f0;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = f0 + f1 + (float)(v0 + v1);
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
float run(float *d, int blocks) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 100);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 4000);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  float *d; hipMalloc(&d, 2048 * 256 * 4);
  for (int blocks : {256, 512}) {   // 1 or 2 waves per SIMD
    printf("blocks %d: mfma-only %.3f ms, valu-only %.3f ms, both %.3f ms\n", blocks, run<1>(d, blocks), run<2>(d, blocks), run<3>(d, blocks));
  }
  return 0;
}
