// float64 arithmetic of one Metropolis step on gfx950, written for the KNOWN argument ranges of the step instead
// of calling the general-purpose device libm:
//
//   normal_pair   Box-Muller on two Philox words:  r = sqrt(-2 ln u1), (r cos 2 pi u2, r sin 2 pi u2),
//                 u = (w + 0.5) 2^-32 in (0, 1)                       (draw_*_group, metropolis_engine.py:261-302)
//   exp_nonpos    e^x for x <= 0, the accept probability exp(-dE/T)   (metropolis_decision, :333-335)
//
// Why: the general log / sincos / sqrt spend most of their instructions on cases that cannot occur here (huge
// arguments and Payne-Hanek reduction, subnormals, negative or infinite inputs, double-double bookkeeping for
// 0.5-ulp results) and their inlined bodies pushed k_step<double> to 180 VGPRs.  Here
//   * ln:   u = z 2^e with z in [0.75, 1.5) taken from the exponent/mantissa bits; c = round(128 z)/128 picks
//           {1/c, 2 ln(1/c)} from a 97-row table in LDS; r = z/c - 1 (one fma, |r| <= 1/192) and a degree-7
//           log1p.  The row c = 1 has 1/c = 1 and ln = 0 EXACTLY, so ln u keeps its relative accuracy as u -> 1.
//   * sqrt: v_rsq_f64 (23 good bits) + one coupled Newton step + one residual correction; the argument is in
//           [2^-32, 45], so no scaling.
//   * sin/cos: the nearest quarter revolution is removed in INTEGER arithmetic on the word itself (exact, no
//           reduction error at all), leaving |p| <= 1/8 revolution; sin(2 pi p) = p S(p^2), cos = C(p^2).
//   * exp:  n = rint(x log2 e), f = x - n ln2 (two-constant Cody-Waite, exact product for |n| < 2^11),
//           e^f = 1 + f + f^2 E(f), v_ldexp_f64.
// Accuracy (tests/test_math64_cpu.py, against long-double libm over 4e6 random words and the edge words):
// -2 ln u <= 2 ulp, radius <= 1.5 ulp, cos/sin <= 2 ulp (relative, also beside their zeros), normals <= 4 ulp, exp <= 1.5 ulp.  The constants come from tools/gen_math64.py.
//
// The header is plain C++ apart from two device builtins, so the same code is compiled for the host by the tests.
#pragma once

#include <stdint.h>

#include "me_math64_coef.h"

#if defined(__HIPCC__)
#define ME_MATH_FN __host__ __device__ __forceinline__
#else
#include <cmath>
#define ME_MATH_FN inline
#endif

namespace me {
namespace math64 {

ME_MATH_FN uint64_t bits_of(double v) { return __builtin_bit_cast(uint64_t, v); }
ME_MATH_FN double double_of(uint32_t hi, uint32_t lo) { return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo); }

// 1/sqrt(y) to >= 20 bits: the hardware estimate on the device
ME_MATH_FN double rsq_estimate(double y) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_rsq(y);
#elif defined(ME_MATH64_TEST_RSQ_NOISE)   // host test hook: an estimate four times worse than v_rsq_f64's
  return (1.0 / std::sqrt(y)) * ((bits_of(y) & 1u) ? 1.0 + 0x1p-21 : 1.0 - 0x1p-21);
#else
  return 1.0 / std::sqrt(y);
#endif
}
ME_MATH_FN double scale_by_pow2(double v, int n) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_ldexp(v, n);
#else
  return std::ldexp(v, n);
#endif
}
ME_MATH_FN double round_even(double v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_rint(v);
#else
  return std::nearbyint(v);
#endif
}

template <int N>
ME_MATH_FN double horner(const double (&c)[N], double v) {
  double s = c[N - 1];
#pragma unroll
  for (int k = N - 2; k >= 0; --k) s = __builtin_fma(s, v, c[k]);
  return s;
}

// -2 ln((w + 0.5) 2^-32) for a 32-bit word w; `table` = kLogTable (in LDS on the device)
ME_MATH_FN double minus_two_log_unit(uint32_t w, const double (*table)[2]) {
  const uint64_t b = bits_of((double)w + 0.5);                   // exact: w + 0.5 has at most 33 significant bits
  const uint32_t hi = (uint32_t)(b >> 32);
  const uint32_t t = hi + 0x00080000u;                           // carries into the exponent iff the mantissa >= 1.5
  const int e = (int)(t >> 20) - (1023 + 32);                    // (w + 0.5) 2^-32 = z 2^e
  const double z = double_of(hi - ((t & 0xfff00000u) - 0x3ff00000u), (uint32_t)b);   // z in [0.75, 1.5)
  const uint32_t j = (uint32_t)bits_of(z + 0x1.8p45);            // ulp(2^45) = 2^-7: the low word is round(128 z)
  const double *row = table[j - (uint32_t)kLogJ0];
  const double r = __builtin_fma(z, row[0], -1.0);
  const double head = __builtin_fma((double)e, kMinusTwoLn2, row[1]);
  return head + __builtin_fma(r * r, horner(kLogTail, r), -2.0 * r);
}

// sqrt(y) for y in [2^-33, 2^6]
ME_MATH_FN double sqrt_bounded(double y) {
  const double rs = rsq_estimate(y);
  double g = y * rs, h = 0.5 * rs;
  const double c = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, c, g);
  h = __builtin_fma(h, c, h);
  return __builtin_fma(__builtin_fma(-g, g, y), h, g);
}

// cos and sin of 2 pi (w + 0.5) 2^-32
ME_MATH_FN void cos_sin_unit(uint32_t w, double &cs, double &sn) {
  const uint32_t q = (w + 0x20000000u) >> 30;                    // nearest quarter revolution, 0 .. 4 (4 == 0)
  const int32_t rem = (int32_t)(w - (q << 30));                  // w - q 2^30 in [-2^29, 2^29)
  const double p = ((double)rem + 0.5) * 0x1p-32;                // exact; |p| < 1/8 revolution
  const double z = p * p;
  const double s = p * horner(kSinRev, z);
  const double c = horner(kCosRev, z);
  // cos(t + q pi/2), sin(t + q pi/2): swap on odd q, signs from q
  const bool odd = (q & 1u) != 0u;
  const uint64_t cb = bits_of(odd ? s : c) ^ ((uint64_t)((q + 1u) & 2u) << 62);
  const uint64_t sb = bits_of(odd ? c : s) ^ ((uint64_t)(q & 2u) << 62);
  cs = __builtin_bit_cast(double, cb);
  sn = __builtin_bit_cast(double, sb);
}

ME_MATH_FN void normal_pair(uint32_t wa, uint32_t wb, const double (*table)[2], double &g0, double &g1) {
  const double radius = sqrt_bounded(minus_two_log_unit(wa, table));
  double cs, sn;
  cos_sin_unit(wb, cs, sn);
  g0 = radius * cs;
  g1 = radius * sn;
}

// e^x for x <= 0 (NaN stays NaN; anything below the subnormal range gives 0)
ME_MATH_FN double exp_nonpos(double x) {
  x = x < -746.0 ? -746.0 : x;
  const double n = round_even(x * kLog2e);
  double f = __builtin_fma(-n, kLn2Hi, x);
  f = __builtin_fma(-n, kLn2Lo, f);
  const double p = __builtin_fma(f * f, horner(kExpTail, f), f) + 1.0;
  return scale_by_pow2(p, (int)n);
}

}  // namespace math64
}  // namespace me
