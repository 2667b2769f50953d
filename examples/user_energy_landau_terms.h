// Example term-wise user energy: the Landau toy of the reference demo as the energy DICTIONARY the demo passes
// (demo/toymodel_complex_and_real.py:17-33):
//     {"complex": {"field"}, "real": {"field", "area"}, "all": {"field", "area"}}
// Same formulas as the built-in ME_ENERGY_LANDAU_TERMS, so the two can be checked against each other.
//   x = [x, y | Re c | Im c]   (2 real + 1 complex),   coef = {k, alpha, beta}
#include "metropolis_user_energy.h"

#define ME_USER_N_TERMS 2
// bit 0: the real group's moves change the term, bit 1: the complex group's moves do
constexpr unsigned me_user_term_groups(int term) { return term == 0 ? 3u : 1u; }

template <typename R>
__device__ R me_user_energy_term(int term, const R *x, const R *coef) {
  const R k = coef[0], alpha = coef[1], beta = coef[2];
  if (term == 0) {                                   // "field": x y (alpha |c|^2 + beta |c|^4)
    const R a2 = x[2] * x[2] + x[3] * x[3];
    return x[0] * x[1] * (alpha * a2 + beta * a2 * a2);
  }
  const R ox = R(1) - x[0], oy = R(1) - x[1];        // "area": k (1-x)^2 + k (1-y)^2
  return k * ox * ox + k * oy * oy;
}
