// Host build of metropolisengine_amd/csrc/me_math64.h for tests/test_math64_cpu.py (g++, no HIP): the float64
// Box-Muller / exp arithmetic of the kernels evaluated on arrays.  -DME_MATH64_TEST_RSQ_NOISE replaces the exact host
// 1/sqrt by an estimate with 2^-21 relative error, as bad as (worse than) the hardware v_rsq_f64 estimate.
#include <cstdint>
#include <cmath>

#include "../../metropolisengine_amd/csrc/me_math64.h"

extern "C" {

void me_math64_normals(const uint32_t *wa, const uint32_t *wb, long n, double *g0, double *g1) {
  for (long i = 0; i < n; ++i) me::math64::normal_pair(wa[i], wb[i], me::math64::kLogTable, g0[i], g1[i]);
}
void me_math64_radius(const uint32_t *wa, long n, double *y, double *r) {
  for (long i = 0; i < n; ++i) {
    y[i] = me::math64::minus_two_log_unit(wa[i], me::math64::kLogTable);
    r[i] = me::math64::sqrt_bounded(y[i]);
  }
}
void me_math64_cos_sin(const uint32_t *wb, long n, double *cs, double *sn) {
  for (long i = 0; i < n; ++i) me::math64::cos_sin_unit(wb[i], cs[i], sn[i]);
}
void me_math64_exp(const double *x, long n, double *out) {
  for (long i = 0; i < n; ++i) out[i] = me::math64::exp_nonpos(x[i]);
}
}
