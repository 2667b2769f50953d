// Example user energy: the cylinder-style surrogate of BASELINE config 5 (2 real + 7 complex Fourier
// coefficients), written against include/metropolis_user_energy.h.  Same formula as the built-in
// ME_ENERGY_CYLINDER and oracle/energies.py:cylinder_surrogate, so the three can be checked against each other.
//   x[0]            surface amplitude (hard wall |x0| >= 1 via ME_REJECT_ABS_REAL0_GE, /metropolis_engine.py:139-141)
//   x[1..ME_NR)     further real shape parameters
//   z_j             field Fourier coefficients, mode number q_j = j - (ME_NC-1)/2
//   coef            {kappa, gamma, wavenumber}
#include "metropolis_user_energy.h"

#define ME_USER_HAS_REJECT
// the cylinder's hard wall: the surface amplitude must stay inside (-1, 1)   (/metropolis_engine.py:139-141)
template <typename R>
__device__ bool me_user_reject(const R *x, const R *coef) {
  return !(x[0] > R(-1) && x[0] < R(1));
}

template <typename R>
__device__ R me_user_energy(const R *x, const R *coef) {
  const R kappa = coef[0], gamma = coef[1], wavenumber = coef[2];
  R s = 0;
#pragma unroll
  for (int i = 0; i < ME_NR; ++i) s += x[i] * x[i];
  const R x0 = x[0];
  const R surface = kappa * s / (R(1) - x0 * x0);
  const R amp = R(1) + R(0.5) * x0 * x0;
  R field = 0, tot = 0;
#pragma unroll
  for (int j = 0; j < ME_NC; ++j) {
    const R q = wavenumber * (R(j) - R(ME_NC - 1) * R(0.5));
    const R mod2 = me_fma(x[ME_NR + j], x[ME_NR + j], x[ME_NR + ME_NC + j] * x[ME_NR + ME_NC + j]);
    field += (gamma + q * q * amp) * mod2;
    tot += mod2;
  }
  return surface + field + R(0.5) * tot * tot;
}
