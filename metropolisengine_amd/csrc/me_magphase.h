// k_step_magphase: the reference's alternative complex sampler, step_complex_group under
// complex_sample_method="magnitude-phase" (/root/reference/metropolisengine/metropolis_engine.py:168-207, :304-317).
//
// One step = two Metropolis decisions on the complex group (the real parameters rest):
//   magnitude stage (:178-192, :304-310)  |z_j| -> |z_j| + (sigma_c^2 Re K_jj) g_j at fixed phase -- the reference
//       hands the variance expression sigma_c^2 K_jj to random.gauss as the *standard deviation* and casts the
//       complex-typed result to real; K is the chain's running covariance_matrix_complex.  Hard wall / energy /
//       accept as usual; the complex width adapts (update_complex_group_sigma, :449-456).
//   phase stage (:194-207, :312-317)      every phase redrawn uniformly in (-pi, pi) at fixed magnitude; accept test;
//       no width update.
// Stream words of one step (oracle/manychain.py:step_magnitude_phase): Box-Muller pairs 0..W1-1 (W1 = 2 ceil(nc/2))
// -> the nc magnitude normals; word W1 -> accept draw of the magnitude stage; words W1+1..W1+nc -> phases
// (-pi + 2 pi u); word W1+nc+1 -> accept draw of the phase stage.
#pragma once

#include "me_device.h"

namespace me {

template <typename R>
struct Trig;
template <>
struct Trig<float> {   // theta = -pi + 2 pi u  ==  (u - 1/2) revolutions: v_cos/v_sin take revolutions
  static __device__ __forceinline__ void unit_phase(float u, float &c, float &s) {
    c = __builtin_amdgcn_cosf(u - 0.5f);
    s = __builtin_amdgcn_sinf(u - 0.5f);
  }
};
template <>
struct Trig<double> {
  static __device__ __forceinline__ void unit_phase(double u, double &c, double &s) {
    const double theta = -3.141592653589793 + 6.283185307179586 * u;
    c = cos(theta);
    s = sin(theta);
  }
};

// a.factor carries the per-chain covariance field (ME_FIELD_COV), not the proposal factors.
template <typename R, int NR, int NC, class Energy, bool INJECT = false>
__global__ void __launch_bounds__(kStepThreads) k_step_magphase(StepArgs<R> a, Energy en) {
  static_assert(NC > 0, "the magnitude-phase sampler acts on complex parameters");
  constexpr int D = NR + 2 * NC;
  constexpr int PR = NR * (NR + 1) / 2;
  constexpr int P = PR + NC * NC;
  constexpr int W1 = 2 * ((NC + 1) / 2);
  constexpr int NWORDS = W1 + NC + 2;
  constexpr int NBLK = (NWORDS + 3) / 4;
  constexpr bool MIXED = NR > 0;
  constexpr int WROW = MIXED ? GROUP_COMPLEX : 0;
  using N_ = Num<R>;

  if constexpr (!INJECT) N_::prepare();
  energy_prepare(en, 0);
  unsigned int wave_accepted = 0;
  bool bad_energy = false, bad_width = false;
  const long long stride = (long long)gridDim.x * kStepThreads;
  using Ledger = EnergyLedger<R, Energy, MIXED ? GROUP_COMPLEX : GROUP_ALL>;   // the complex group's terms (:183-189)
  const XField<R, D> fx(a.x, a.n);
  const Field<R> fe(a.energy, a.n, Ledger::T), fw(a.width, a.n, MIXED ? 3 : 1);
  const TiledField<R> fcov(a.factor, a.n, P);      // the covariance field (tile-major)
  for (long long c = (long long)blockIdx.x * kStepThreads + threadIdx.x; c < a.n; c += stride) {
    const unsigned int coff = (unsigned int)c * (unsigned int)sizeof(R);
    const unsigned int xoff = fx.offset(c);
    R x[D], kdiag[NC];
#pragma unroll
    for (int d = 0; d < D; ++d) x[d] = fx.load(d, xoff);
#pragma unroll
    for (int j = 0; j < NC; ++j) kdiag[j] = fcov.load(cdiag(PR, j), tiled_offset<R>(c, P));
    Ledger ledger;
    ledger.load(fe, coff);
    R w = fw.load(WROW, coff);
    const unsigned long long gid = a.chain_offset + (unsigned long long)c;

    // wall, energy, accept rule (:247-252, :319-338) and commit of a proposed complex block
    auto decide = [&](const R (&xp)[D], R u) -> bool {
      bool rejected = false;
      if (a.reject_kind == ME_REJECT_ABS_REAL0_GE) rejected = !(N_::abs_(xp[0]) < a.reject_bound);
      else if (a.reject_kind == ME_REJECT_USER) rejected = energy_reject(en, xp, 0);
      R terms_new[Ledger::T];
      const R e_new = ledger.propose(en, xp, terms_new);
      const R diff = e_new - ledger.partial();
      bool accept = diff <= R(0);
      if (a.temp > R(0)) accept = accept || N_::uphill(u, diff, a.inv_temp, a.inv_temp_log2e);
      accept = accept && !rejected;
      bad_energy |= (!rejected && !N_::finite(e_new));
#pragma unroll
      for (int d = NR; d < D; ++d) x[d] = accept ? xp[d] : x[d];
      ledger.commit(accept, terms_new);
      wave_accepted += (unsigned int)__popcll(__ballot(accept));
      return accept;
    };

    for (int s = 0; s < a.n_sweeps; ++s) {
      const unsigned long long step = a.step_index + (unsigned long long)s;
      R g[W1], u_mag, u_phase[NC], u_ph;
      if constexpr (INJECT) {
#pragma unroll
        for (int j = 0; j < NC; ++j) g[j] = a.inj_normals[((long long)s * NC + j) * a.n + c];
        u_mag = a.inj_uniforms[((long long)s * (NC + 2) + 0) * a.n + c];
#pragma unroll
        for (int j = 0; j < NC; ++j) u_phase[j] = a.inj_uniforms[((long long)s * (NC + 2) + 1 + j) * a.n + c];
        u_ph = a.inj_uniforms[((long long)s * (NC + 2) + NC + 1) * a.n + c];
      } else {
        uint32_t words[4 * NBLK];
#pragma unroll
        for (int b = 0; b < NBLK; ++b) {
          U4 ctr;
          ctr.x = (uint32_t)gid;
          ctr.y = (uint32_t)(gid >> 32);
          ctr.z = (uint32_t)step;
          ctr.w = ((uint32_t)(step >> 32) << 16) | (uint32_t)b;
          const U4 o = philox4x32_10(ctr, a.seed_lo, a.seed_hi);
          words[4 * b + 0] = o.x;
          words[4 * b + 1] = o.y;
          words[4 * b + 2] = o.z;
          words[4 * b + 3] = o.w;
        }
#pragma unroll
        for (int q = 0; q < W1 / 2; ++q) N_::normal_pair(words[2 * q], words[2 * q + 1], g[2 * q], g[2 * q + 1]);
        u_mag = N_::unit(words[W1]);
#pragma unroll
        for (int j = 0; j < NC; ++j) u_phase[j] = N_::unit(words[W1 + 1 + j]);
        u_ph = N_::unit(words[W1 + NC + 1]);
      }

      R xp[D];
#pragma unroll
      for (int d = 0; d < D; ++d) xp[d] = x[d];
      // ---- magnitude stage: cmath.polar / cmath.rect (:308-310) without the trigonometry
#pragma unroll
      for (int j = 0; j < NC; ++j) {
        const R re = x[NR + j], im = x[NR + NC + j];
        const R mag = N_::sqrt_(re * re + im * im);
        const R inv = mag > R(0) ? R(1) / mag : R(0);
        const R dre = mag > R(0) ? re * inv : R(1);   // polar(0) = (0, phase 0)
        const R dim = mag > R(0) ? im * inv : R(0);
        const R mnew = mag + (w * w * kdiag[j]) * g[j];
        xp[NR + j] = mnew * dre;
        xp[NR + NC + j] = mnew * dim;
      }
      const bool acc_mag = decide(xp, u_mag);
      w = N_::adapt(w, acc_mag, a.ratio, a.p, a.damping, a.up, a.down);   // update_complex_group_sigma (:449-456)
      // ---- phase stage
#pragma unroll
      for (int j = 0; j < NC; ++j) {
        const R re = x[NR + j], im = x[NR + NC + j];
        const R mag = N_::sqrt_(re * re + im * im);
        R cs, sn;
        Trig<R>::unit_phase(u_phase[j], cs, sn);
        xp[NR + j] = mag * cs;
        xp[NR + NC + j] = mag * sn;
      }
      decide(xp, u_ph);
    }
    bad_width |= !(w > R(0));
#pragma unroll
    for (int d = NR; d < D; ++d) fx.store(d, xoff, x[d]);
    ledger.store(fe, coff);
    fw.store(WROW, coff, w);
  }
  if ((threadIdx.x & 63) == 0 && wave_accepted) {
    unsigned long long *slot = a.accept_slots + (size_t)blockIdx.x * (kStepThreads / 64) + (threadIdx.x >> 6);
    *slot += (unsigned long long)wave_accepted;
  }
  const unsigned int bits = (bad_energy ? ST_NONFINITE_ENERGY : 0u) | (bad_width ? ST_BAD_WIDTH : 0u);
  if (bits) atomicOr(a.status, bits);
}

}  // namespace me
