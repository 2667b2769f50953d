// Dev microbenchmark (GPU box): issue cost of the 32x32->64 multiply forms available for Philox on gfx950.
#include <cstdio>
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int MODE>
__global__ void __launch_bounds__(256) k(uint32_t *out, int iters) {
  uint32_t a0 = threadIdx.x + 1, a1 = threadIdx.x * 3 + 7, a2 = blockIdx.x + 11, a3 = threadIdx.x ^ 0x5bd1e995u;
  uint32_t s = 0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (MODE == 0) {          // v_mad_u64_u32 (what hipcc emits for (uint64_t)a * b)
        uint64_t p0 = (uint64_t)0xD2511F53u * a0, p1 = (uint64_t)0xCD9E8D57u * a1, p2 = (uint64_t)0xD2511F53u * a2,
                 p3 = (uint64_t)0xCD9E8D57u * a3;
        a0 = (uint32_t)(p0 >> 32) ^ (uint32_t)p1; a1 = (uint32_t)(p1 >> 32) ^ (uint32_t)p2;
        a2 = (uint32_t)(p2 >> 32) ^ (uint32_t)p3; a3 = (uint32_t)(p3 >> 32) ^ (uint32_t)p0;
      } else if (MODE == 1) {   // separate mul_hi + mul_lo
        uint32_t h0 = __umulhi(0xD2511F53u, a0), l0 = 0xD2511F53u * a0, h1 = __umulhi(0xCD9E8D57u, a1), l1 = 0xCD9E8D57u * a1;
        uint32_t h2 = __umulhi(0xD2511F53u, a2), l2 = 0xD2511F53u * a2, h3 = __umulhi(0xCD9E8D57u, a3), l3 = 0xCD9E8D57u * a3;
        asm volatile("" : "+v"(h0), "+v"(l0), "+v"(h1), "+v"(l1));
        a0 = h0 ^ l1; a1 = h1 ^ l2; a2 = h2 ^ l3; a3 = h3 ^ l0;
      } else if (MODE == 2) {   // plain xor/add (reference for a full-rate VALU op count: 8 ops)
        a0 = (a0 ^ a1) + a2; a1 = (a1 ^ a2) + a3; a2 = (a2 ^ a3) + a0; a3 = (a3 ^ a0) + a1;
      } else {                  // fp32 fma (4 ops)
        float f0 = __uint_as_float(a0), f1 = __uint_as_float(a1), f2 = __uint_as_float(a2), f3 = __uint_as_float(a3);
        f0 = f0 * 1.0001f + f1; f1 = f1 * 0.9999f + f2; f2 = f2 * 1.0001f + f3; f3 = f3 * 0.9999f + f0;
        a0 = __float_as_uint(f0); a1 = __float_as_uint(f1); a2 = __float_as_uint(f2); a3 = __float_as_uint(f3);
      }
    }
    s ^= a0;
  }
  out[blockIdx.x * 256 + threadIdx.x] = s ^ a1 ^ a2 ^ a3;
}

template <int MODE>
float run(uint32_t *d, int blocks, int iters) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 10);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  uint32_t *d; (void)hipMalloc(&d, 4096 * 256 * 4);
  const int iters = 2000;
  for (int blocks : {256, 1024, 2048}) {   // 1, 4, 8 waves per SIMD
    const double waves_per_simd = blocks * 4 / 1024.0;
    const double groups = (double)iters * 8 * waves_per_simd;   // per SIMD: number of 4-op groups issued
    float t0 = run<0>(d, blocks, iters), t1 = run<1>(d, blocks, iters), t2 = run<2>(d, blocks, iters), t3 = run<3>(d, blocks, iters);
    printf("waves/SIMD %.0f: mad_u64 %.3f ms (%.1f cyc per mul @2.4GHz), mulhi+mullo %.3f ms (%.1f cyc per pair), xor/add %.3f ms (%.1f cyc/op), fma %.3f ms (%.1f cyc/op)\n",
           waves_per_simd, t0, t0 * 1e-3 * 2.4e9 / (groups * 4), t1, t1 * 1e-3 * 2.4e9 / (groups * 4), t2,
           t2 * 1e-3 * 2.4e9 / (groups * 8), t3, t3 * 1e-3 * 2.4e9 / (groups * 4));
  }
  return 0;
}
