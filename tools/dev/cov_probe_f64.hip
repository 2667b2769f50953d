// Dev probe (GPU box): memory-only floors of the FLOAT64 covariance-carrying kernels at 16 real parameters, 2^20 chains --
// the access patterns of k_measure<double,16,0,fused> and k_step<double,16,0,per-chain factor>, no arithmetic:
//   measure: read x (16 rows) + width; read-modify-write mean (16) and observables (32), component-major, default policy;
//            read-modify-write the packed covariance (136 entries, tile-major) and write the factor (136), non-temporal
//   step:    read-modify-write 18 state rows (default policy), read the packed factor (136, tile-major, non-temporal)
// BATCH = how many packed entries a lane has in flight at a time (136: everything, ~290 registers, one wavefront per
// SIMD as the real kernels; 32 / 16: register-light, several wavefronts per SIMD) -- does occupancy matter to the floor?
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int P = 136, D = 16, NOBS = 32;
using d2 = __attribute__((ext_vector_type(2))) double;

template <int BATCH>
__global__ void __launch_bounds__(64) k_measure_pattern(const double *x, const double *w, double *mean, double *obs, double *cov, double *fac, long long n) {
  const long long c = (long long)blockIdx.x * 64 + threadIdx.x;
  if (c >= n) return;
  double xs[D], mu[D], ob[NOBS];
#pragma unroll
  for (int d = 0; d < D; ++d) xs[d] = x[d * n + c];
#pragma unroll
  for (int d = 0; d < D; ++d) mu[d] = mean[d * n + c];
  double acc = w[c];
#pragma unroll
  for (int d = 0; d < D; ++d) mean[d * n + c] = mu[d] * 0.99 + xs[d] * 0.01;
#pragma unroll
  for (int k = 0; k < NOBS; ++k) ob[k] = obs[k * n + c];
#pragma unroll
  for (int k = 0; k < NOBS; ++k) obs[k * n + c] = ob[k] * 0.99 + xs[k & 15] * 0.01;
  double *pc = cov + (c >> 6) * (long long)P * 64 + (c & 63), *pf = fac + (c >> 6) * (long long)P * 64 + (c & 63);
#pragma unroll
  for (int k0 = 0; k0 < P; k0 += BATCH) {
    double m[BATCH];
#pragma unroll
    for (int u = 0; u < BATCH; ++u)
      if (k0 + u < P) m[u] = __builtin_nontemporal_load(pc + (k0 + u) * 64);
#pragma unroll
    for (int u = 0; u < BATCH; ++u)
      if (k0 + u < P) {
        const double v = m[u] * 0.98 + acc * 1e-30;
        __builtin_nontemporal_store(v, pc + (k0 + u) * 64);
        __builtin_nontemporal_store(v + 1.0, pf + (k0 + u) * 64);
      }
  }
}

template <int BATCH>
__global__ void __launch_bounds__(64) k_step_pattern(double *state, const double *fac, long long n) {
  const long long c = (long long)blockIdx.x * 64 + threadIdx.x;
  if (c >= n) return;
  double xs[D + 2];
#pragma unroll
  for (int d = 0; d < D + 2; ++d) xs[d] = state[d * n + c];
  const double *pf = fac + (c >> 6) * (long long)P * 64 + (c & 63);
  double s = 0.0;
#pragma unroll
  for (int k0 = 0; k0 < P; k0 += BATCH) {
    double m[BATCH];
#pragma unroll
    for (int u = 0; u < BATCH; ++u)
      if (k0 + u < P) m[u] = __builtin_nontemporal_load(pf + (k0 + u) * 64);
#pragma unroll
    for (int u = 0; u < BATCH; ++u)
      if (k0 + u < P) s += m[u];
  }
#pragma unroll
  for (int d = 0; d < D + 2; ++d) state[d * n + c] = xs[d] + s * 1e-30;
}

// the same measure pattern with the packed fields as ENTRY PAIRS, [tile of 64][P / 2][64 lanes][2]: 16-byte accesses,
// half as many memory instructions (would the floor move?  the dense 64-parameter state gained 15 % that way, the
// 16-parameter state nothing: tools/dev/state_layout_probe64.hip, rows_probe_f64_tiled.hip)
template <int BATCH>
__global__ void __launch_bounds__(64) k_measure_pattern_pairs(const double *x, const double *w, double *mean, double *obs, d2 *cov, d2 *fac, long long n) {
  const long long c = (long long)blockIdx.x * 64 + threadIdx.x;
  if (c >= n) return;
  double xs[D], mu[D], ob[NOBS];
#pragma unroll
  for (int d = 0; d < D; ++d) xs[d] = x[d * n + c];
#pragma unroll
  for (int d = 0; d < D; ++d) mu[d] = mean[d * n + c];
  double acc = w[c];
#pragma unroll
  for (int d = 0; d < D; ++d) mean[d * n + c] = mu[d] * 0.99 + xs[d] * 0.01;
#pragma unroll
  for (int k = 0; k < NOBS; ++k) ob[k] = obs[k * n + c];
#pragma unroll
  for (int k = 0; k < NOBS; ++k) obs[k * n + c] = ob[k] * 0.99 + xs[k & 15] * 0.01;
  constexpr int P2 = P / 2;
  d2 *pc = cov + (c >> 6) * (long long)P2 * 64 + (c & 63), *pf = fac + (c >> 6) * (long long)P2 * 64 + (c & 63);
#pragma unroll
  for (int k0 = 0; k0 < P2; k0 += BATCH) {
    d2 m[BATCH];
#pragma unroll
    for (int u = 0; u < BATCH; ++u)
      if (k0 + u < P2) m[u] = __builtin_nontemporal_load(pc + (k0 + u) * 64);
#pragma unroll
    for (int u = 0; u < BATCH; ++u)
      if (k0 + u < P2) {
        const d2 v = m[u] * 0.98 + acc * 1e-30;
        __builtin_nontemporal_store(v, pc + (k0 + u) * 64);
        __builtin_nontemporal_store(v + 1.0, pf + (k0 + u) * 64);
      }
  }
}

template <class F>
float time_it(F &&launch) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) launch();
  (void)hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) launch();
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms / 10 * 1e3f;
}

int main() {
  const long long n = 1 << 20;
  double *x, *w, *mean, *obs, *cov, *fac;
  (void)hipMalloc(&x, 8 * n * (D + 2)); (void)hipMalloc(&w, 8 * n); (void)hipMalloc(&mean, 8 * n * D); (void)hipMalloc(&obs, 8 * n * NOBS);
  (void)hipMalloc(&cov, 8 * n * P); (void)hipMalloc(&fac, 8 * n * P);
  (void)hipMemset(x, 0, 8 * n * (D + 2)); (void)hipMemset(w, 0, 8 * n); (void)hipMemset(mean, 0, 8 * n * D); (void)hipMemset(obs, 0, 8 * n * NOBS);
  (void)hipMemset(cov, 0, 8 * n * P); (void)hipMemset(fac, 0, 8 * n * P);
  const double measure_bytes = 8.0 * n * (D + 1 + 2 * D + 2 * NOBS + 2 * P + P), step_bytes = 8.0 * n * (2 * (D + 2) + P);
#define RUN_M(B) { const float us = time_it([&] { hipLaunchKernelGGL(k_measure_pattern<B>, dim3((unsigned)(n / 64)), dim3(64), 0, 0, x, w, mean, obs, cov, fac, n); }); \
    printf("k_measure<double,16,0> pattern, %3d packed entries in flight: %7.1f us  %5.0f GB/s  (%.3f of 8 TB/s)\n", B, us, measure_bytes / us / 1e3, measure_bytes / us / 8e6); }
#define RUN_S(B) { const float us = time_it([&] { hipLaunchKernelGGL(k_step_pattern<B>, dim3((unsigned)(n / 64)), dim3(64), 0, 0, x, fac, n); }); \
    printf("k_step<double,16,0,per-chain> pattern, %3d entries in flight: %7.1f us  %5.0f GB/s  (%.3f of 8 TB/s)\n", B, us, step_bytes / us / 1e3, step_bytes / us / 8e6); }
  RUN_M(136) RUN_M(68) RUN_M(32) RUN_M(16)
#define RUN_P(B) { const float us = time_it([&] { hipLaunchKernelGGL(k_measure_pattern_pairs<B>, dim3((unsigned)(n / 64)), dim3(64), 0, 0, x, w, mean, obs, (d2 *)cov, (d2 *)fac, n); }); \
    printf("k_measure<double,16,0> pattern, 16-byte entry pairs, %3d pairs in flight: %7.1f us  %5.0f GB/s  (%.3f of 8 TB/s)\n", B, us, measure_bytes / us / 1e3, measure_bytes / us / 8e6); }
  RUN_P(68) RUN_P(34) RUN_P(16)
  RUN_S(136) RUN_S(68) RUN_S(32) RUN_S(16)
  printf("algorithmic bytes per launch: measure %.0f MB, step %.0f MB\n", measure_bytes / 1e6, step_bytes / 1e6);
  return 0;
}
