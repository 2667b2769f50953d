#include <cstdio>
#include <hip/hip_runtime.h>
__global__ void k(float *out) {
  const int lane = threadIdx.x;
  float a = (float)lane, b = 100.0f + lane;
  const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
  out[lane] = __builtin_bit_cast(float, r[0]);
  out[64 + lane] = __builtin_bit_cast(float, r[1]);
}
int main() {
  float *d, h[128];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("r[0]: lane0 %g lane31 %g lane32 %g lane63 %g\n", h[0], h[31], h[32], h[63]);
  printf("r[1]: lane0 %g lane31 %g lane32 %g lane63 %g\n", h[64], h[95], h[96], h[127]);
  return 0;
}
