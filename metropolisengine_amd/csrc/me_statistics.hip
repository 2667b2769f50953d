// Equilibration detection for many recorded series at once: the reference asks pymbar for
// (t0, g, Neff_max) of each DataFrame column of its ONE chain (statistics.py:25-48, metropolis_engine.py:481-504); an
// ensemble run has thousands of traced chains x columns, each an O(T^2) scan on the host.  PARITY UNPINNED as for the
// host version (metropolisengine_amd/statistics.py restates the published algorithm; pymbar is absent): these kernels
// are tested against that host restatement.
//
// k_inefficiency: one wavefront per (series, t0).  For the tail a[t0:] it forms the mean and variance, then walks the
// lags t = 1, 2, 4, 7, 11, ... ("fast": the increment grows by one; otherwise 1, 2, 3, ...) until the normalised
// autocorrelation first turns non-positive after `mintime` lags, accumulating g = 1 + 2 sum C_t (1 - t/n) dt, and
// writes Neff(t0) = (T - t0 + 1) / g.  k_best_start: one workgroup per series takes the first maximum of Neff.
#include "me_internal.h"

namespace me {
namespace {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int w = 32; w > 0; w >>= 1) v += __shfl_xor(v, w, 64);
  return v;
}

__global__ void __launch_bounds__(kBlockThreads) k_inefficiency(const double *series, long long n_series, long long length,
                                                                int fast, int nskip, int mintime, double *g_out,
                                                                double *neff_out) {
  const int lane = threadIdx.x & 63;
  const long long n_starts = (length - 1 + nskip - 1) / nskip;            // t0 = 0, nskip, 2 nskip, ... < length - 1
  const long long task = (long long)blockIdx.x * (kBlockThreads / 64) + (threadIdx.x >> 6);
  if (task >= n_series * n_starts) return;
  const long long s = task / n_starts, t0 = (task % n_starts) * nskip;
  const double *a = series + s * length + t0;
  const long long n = length - t0;
  double sum = 0.0;
  for (long long i = lane; i < n; i += 64) sum += a[i];
  const double mean = wave_sum(sum) / (double)n;
  double ss = 0.0;
  for (long long i = lane; i < n; i += 64) {
    const double d = a[i] - mean;
    ss += d * d;
  }
  const double sigma2 = wave_sum(ss) / (double)n;
  double g;
  if (sigma2 == 0.0) {
    g = (double)(length - t0 + 1);                                        // the host's ValueError branch
  } else {
    g = 1.0;
    long long t = 1, increment = 1;
    while (t < n - 1) {
      double c = 0.0;
      for (long long i = lane; i < n - t; i += 64) c += (a[i] - mean) * (a[i + t] - mean);
      c = wave_sum(c) / ((double)(n - t) * sigma2);
      if (c <= 0.0 && t > mintime) break;
      g += 2.0 * c * (1.0 - (double)t / (double)n) * (double)increment;
      t += increment;
      if (fast) increment += 1;
    }
    if (g < 1.0) g = 1.0;
  }
  if (lane == 0) {
    g_out[s * (length - 1) + t0] = g;
    neff_out[s * (length - 1) + t0] = (double)(length - t0 + 1) / g;
  }
}

__global__ void __launch_bounds__(kBlockThreads) k_best_start(const double *g, const double *neff, long long length,
                                                              long long *t0_out, double *g_out, double *neff_out) {
  __shared__ double best_v[kBlockThreads];
  __shared__ long long best_t[kBlockThreads];
  const long long s = blockIdx.x, m = length - 1;
  double v = -1.0;
  long long at = 0;
  for (long long t = threadIdx.x; t < m; t += kBlockThreads) {
    const double x = neff[s * m + t];
    if (x > v) { v = x; at = t; }                                        // ascending t per thread: keeps the first maximum
  }
  best_v[threadIdx.x] = v;
  best_t[threadIdx.x] = at;
  __syncthreads();
  for (int w = kBlockThreads / 2; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) {
      const double o = best_v[threadIdx.x + w];
      const long long ot = best_t[threadIdx.x + w];
      if (o > best_v[threadIdx.x] || (o == best_v[threadIdx.x] && ot < best_t[threadIdx.x])) {
        best_v[threadIdx.x] = o;
        best_t[threadIdx.x] = ot;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    t0_out[s] = best_t[0];
    g_out[s] = g[s * m + best_t[0]];
    neff_out[s] = best_v[0];
  }
}

__global__ void __launch_bounds__(kBlockThreads) k_fill(double *p, long long count, double value) {
  const long long stride = (long long)gridDim.x * kBlockThreads;
  for (long long i = (long long)blockIdx.x * kBlockThreads + threadIdx.x; i < count; i += stride) p[i] = value;
}

}  // namespace

// series: device [n_series][length]; scratch: device 2 * n_series * (length - 1) doubles; results: device arrays.
hipError_t launch_detect_equilibration(const double *series, long long n_series, long long length, int fast, int nskip,
                                       double *scratch, long long *t0_out, double *g_out, double *neff_out,
                                       hipStream_t stream) {
  const long long m = length - 1;
  double *g = scratch, *neff = scratch + n_series * m;
  // starts skipped by nskip keep the host's initial value 1 (detect_equilibration: g_t = neff_t = ones)
  hipLaunchKernelGGL(k_fill, dim3(1024), dim3(kBlockThreads), 0, stream, scratch, 2 * n_series * m, 1.0);
  const long long n_starts = (m + nskip - 1) / nskip;
  const long long tasks = n_series * n_starts;
  const long long blocks = (tasks + kBlockThreads / 64 - 1) / (kBlockThreads / 64);
  if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_inefficiency, dim3((unsigned)blocks), dim3(kBlockThreads), 0, stream, series, n_series, length, fast,
                     nskip, 3, g, neff);
  hipLaunchKernelGGL(k_best_start, dim3((unsigned)n_series), dim3(kBlockThreads), 0, stream, (const double *)g,
                     (const double *)neff, length, t0_out, g_out, neff_out);
  return hipGetLastError();
}

}  // namespace me
