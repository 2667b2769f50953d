// One kernel set per (n_real, n_complex): compiled once per pair with -DME_NR=.. -DME_NC=.. (see build.py), for
// both device dtypes, and self-registered with the C-ABI layer at load time.
//   -DME_DENSE=1     also instantiate the dense quadratic-form energy (x^T A x)
//   -DME_PER_CHAIN=0 omit the per-chain covariance/factor kernels
//   -DME_PER_CHAIN=1 per-chain kernels with the packed matrix in registers (at most 160 entries)
//   -DME_PER_CHAIN=2 per-chain kernels that STREAM the packed matrix (more than 160 entries; pure real spaces)
#include <type_traits>

#include "me_device.h"
#include "me_dense_mfma.h"
#include "me_dense_bf16x3.h"
#include "me_dense_f64.h"
#include "me_factor_tile.h"
#include "me_per_device.h"
#include "me_pool_gram.h"
#include "me_magphase.h"

// User-energy plugin mode: -DME_USER_SOURCE='"file"' -DME_USER_NAME='"name"' compiles the SAME kernels around a
// user-written device function (include/metropolis_user_energy.h) into a plugin library that registers itself.
#ifdef ME_USER_SOURCE
#include ME_USER_SOURCE
#endif

#ifndef ME_NR
#error "compile with -DME_NR=<n_real> -DME_NC=<n_complex>"
#endif
#ifndef ME_DENSE
#define ME_DENSE 0
#endif
#ifndef ME_PER_CHAIN
#define ME_PER_CHAIN 1
#endif
#ifndef ME_TRACK_COV
#define ME_TRACK_COV 1   // k_measure can maintain the per-chain running covariance (streaming form when ME_PER_CHAIN=0)
#endif
#ifndef ME_MEASURE_FUSED_MAX_P
#define ME_MEASURE_FUSED_MAX_P 160   // largest packed size whose Cholesky is fused into k_measure (all per-chain kernel sets; k_factor stays as the split form)
#endif

namespace me {
namespace {

constexpr int NR = ME_NR;
constexpr int NC = ME_NC;
[[maybe_unused]] constexpr int D = NR + 2 * NC;
constexpr bool kLandau = (NR == 2 && NC == 1);
constexpr bool kCylinder = (NR >= 1 && NC >= 1);

#ifdef ME_USER_SOURCE
#ifdef ME_USER_N_TERMS
}  // namespace
}  // namespace me
// a term-wise plugin's total energy is the sum of its terms (calc_energy_total, metropolis_engine.py:158-162)
template <typename R>
__device__ __forceinline__ R me_user_energy(const R *x, const R *coef) {
  R s = me_user_energy_term<R>(0, x, coef);
#pragma unroll
  for (int t = 1; t < ME_USER_N_TERMS; ++t) s += me_user_energy_term<R>(t, x, coef);
  return s;
}
namespace me {
namespace {
#endif
// Direct form: the user function is inlined into k_step (state stays in registers).
template <typename R, int NR_, int NC_>
struct EnergyUser {
  static constexpr int D = NR_ + 2 * NC_;
  const R *coef;
  __device__ __forceinline__ R operator()(const R (&x)[D]) const { return me_user_energy<R>(x, coef); }
#ifdef ME_USER_N_TERMS
  static constexpr int kTerms = ME_USER_N_TERMS;
  static constexpr unsigned term_groups(int t) { return me_user_term_groups(t); }
  __device__ __forceinline__ R term(int t, const R (&x)[D]) const { return me_user_energy_term<R>(t, x, coef); }
#endif
#ifdef ME_USER_HAS_REJECT
  __device__ __forceinline__ bool reject(const R (&x)[D]) const { return me_user_reject<R>(x, coef); }
#endif
};
// Indirect form: the call goes through a __device__ function pointer read from the plugin's code object
// (BASELINE config 5: "user-callback energy via device fn pointer"); costs a real call and a private-memory x.
template <typename R>
using user_fn_t = R (*)(const R *, const R *);
template <typename R>
__device__ R me_user_energy_entry(const R *x, const R *coef) { return me_user_energy<R>(x, coef); }
__device__ user_fn_t<float> g_user_fn_f32 = &me_user_energy_entry<float>;
__device__ user_fn_t<double> g_user_fn_f64 = &me_user_energy_entry<double>;
template <typename R, int NR_, int NC_>
struct EnergyUserIndirect {
  static constexpr int D = NR_ + 2 * NC_;
  user_fn_t<R> fn;
  const R *coef;
  __device__ __forceinline__ R operator()(const R (&x)[D]) const { return fn(x, coef); }
#ifdef ME_USER_HAS_REJECT
  __device__ __forceinline__ bool reject(const R (&x)[D]) const { return me_user_reject<R>(x, coef); }
#endif
};
template <typename R>
user_fn_t<R> load_user_fn() {
  user_fn_t<R> fn = nullptr;
  if constexpr (std::is_same<R, float>::value) (void)hipMemcpyFromSymbol(&fn, HIP_SYMBOL(g_user_fn_f32), sizeof(fn));
  else (void)hipMemcpyFromSymbol(&fn, HIP_SYMBOL(g_user_fn_f64), sizeof(fn));
  return fn;
}
#endif

// rows of the energy ledger for this kind (0: the kind is not in this kernel set)
int energy_terms(int kind) {
#ifdef ME_USER_SOURCE
#ifdef ME_USER_N_TERMS
  if (kind == ME_ENERGY_USER) return ME_USER_N_TERMS;
#else
  if (kind == ME_ENERGY_USER) return 1;
#endif
  return kind == ME_ENERGY_USER_INDIRECT ? 1 : 0;
#endif
  switch (kind) {
    case ME_ENERGY_ISO_QUAD:
    case ME_ENERGY_DIAG_QUAD: return 1;
    case ME_ENERGY_DENSE_QUAD: return ME_DENSE != 0 ? 1 : 0;
    case ME_ENERGY_LANDAU_TOY: return kLandau ? 1 : 0;
    case ME_ENERGY_LANDAU_TERMS: return kLandau ? 2 : 0;
    case ME_ENERGY_CYLINDER: return kCylinder ? 1 : 0;
    default: return 0;
  }
}
bool has_energy(int kind) { return energy_terms(kind) > 0; }

template <typename R>
StepArgs<R> typed(const StepLaunch &l) {
  StepArgs<R> a;
  a.x = (R *)l.x;
  a.energy = (R *)l.energy;
  a.width = (R *)l.width;
  a.factor = (const R *)l.factor;
  a.inj_normals = (const R *)l.inj_normals;
  a.inj_uniforms = (const R *)l.inj_uniforms;
  a.accept_slots = l.accept_slots;
  a.status = l.status;
  a.n = l.n;
  a.chain_offset = l.chain_offset;
  a.step_index = l.step_index;
  a.seed_lo = (uint32_t)l.seed;
  a.seed_hi = (uint32_t)(l.seed >> 32);
  a.n_sweeps = l.n_sweeps;
  a.reject_kind = l.reject_kind;
  a.split_widths = l.split_widths;
  a.stale_total = l.stale_total;
  a.reject_bound = (R)l.reject_bound;
  a.temp = (R)l.temp;
  a.inv_temp = l.temp > 0 ? (R)(1.0 / l.temp) : (R)0;
  a.inv_temp_log2e = l.temp > 0 ? (R)(1.4426950408889634 / l.temp) : (R)0;
  a.ratio = (R)l.ratio;
  a.p = (R)l.target_acceptance;
  a.damping = (R)l.damping;
  a.up = (R)(l.ratio * (1.0 - l.target_acceptance) / l.damping);
  a.down = (R)(-l.ratio * l.target_acceptance / l.damping);
  return a;
}

template <typename R, class Energy, bool INJECT, int GROUP>
hipError_t launch_step_cov(const StepLaunch &l, const StepArgs<R> &a, const Energy &en, dim3 grid, dim3 block,
                           hipStream_t stream) {
  switch (l.cov_kind) {
    case CK_IDENTITY: {
      // A state far beyond the Infinity Cache is streamed non-temporally (StateField).  The threshold is TWICE the cache
      // budget: 16 parameters at 2^22 chains, float32 (302 MB: a good part still hits) 107 us default / 114 us nt;
      // float64 (604 MB) 236 us default / 225 us nt.
      constexpr long long state_bytes = (long long)sizeof(R) * (D + 2);
      if constexpr (!INJECT && GROUP == GROUP_ALL) {
        if (state_bytes * l.n > 2 * cache_budget_bytes()) {
          hipLaunchKernelGGL((k_step<R, NR, NC, Energy, CK_IDENTITY, INJECT, GROUP, true>), grid, block, 0, stream, a, en);
          break;
        }
      }
      hipLaunchKernelGGL((k_step<R, NR, NC, Energy, CK_IDENTITY, INJECT, GROUP>), grid, block, 0, stream, a, en);
      break;
    }
    case CK_SHARED: hipLaunchKernelGGL((k_step<R, NR, NC, Energy, CK_SHARED, INJECT, GROUP>), grid, block, 0, stream, a, en); break;
#if ME_PER_CHAIN == 2
    case CK_PER_CHAIN: {
      // streamed factors (8-37 KB per chain and step) never stay in any cache: one variant, read non-temporally.  (The
      // stream is thousands of lines of straight-line code per instantiation; the replay hook does not get one.)
      if constexpr (INJECT) return hipErrorInvalidValue;
      else hipLaunchKernelGGL((k_step<R, NR, NC, Energy, CK_PER_CHAIN_NT, INJECT, GROUP>), grid, block, 0, stream, a, en);
      break;
    }
#elif ME_PER_CHAIN
    case CK_PER_CHAIN: {
      // Launches whose working set (state + packed factor) cannot stay in the Infinity Cache read the factor
      // non-temporally (CK_PER_CHAIN_NT): streamed with the default policy it evicts the chain state, which then goes to
      // HBM as well (16 real parameters, 2^20 chains: 146 -> 114 us).  Smaller working sets keep the default policy --
      // there the factor IS resident from launch to launch and nt would send it to HBM (config 3: 33 -> 38 us).
      constexpr long long per_chain_bytes = (long long)sizeof(R) * (D + 2 + NR * (NR + 1) / 2 + NC * NC);
      if constexpr (!INJECT) {
        if (per_chain_bytes * l.n > cache_budget_bytes()) {
          hipLaunchKernelGGL((k_step<R, NR, NC, Energy, CK_PER_CHAIN_NT, INJECT, GROUP>), grid, block, 0, stream, a, en);
          break;
        }
      }
      hipLaunchKernelGGL((k_step<R, NR, NC, Energy, CK_PER_CHAIN, INJECT, GROUP>), grid, block, 0, stream, a, en);
      break;
    }
#endif
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

template <typename R, class Energy, bool INJECT>
hipError_t launch_step_group(const StepLaunch &l, const StepArgs<R> &a, const Energy &en, dim3 grid, dim3 block,
                             hipStream_t stream) {
  if (l.group == GROUP_ALL) return launch_step_cov<R, Energy, INJECT, GROUP_ALL>(l, a, en, grid, block, stream);
  if constexpr (NR > 0 && NC > 0) {   // group-wise stepping exists for mixed engines (pure engines alias step_all)
    if (l.group == GROUP_REAL) return launch_step_cov<R, Energy, INJECT, GROUP_REAL>(l, a, en, grid, block, stream);
    if (l.group == GROUP_COMPLEX) return launch_step_cov<R, Energy, INJECT, GROUP_COMPLEX>(l, a, en, grid, block, stream);
  }
  return hipErrorInvalidValue;
}

template <typename R, class Energy>
hipError_t step_with(const StepLaunch &l, const Energy &en, hipStream_t stream) {
  const StepArgs<R> a = typed<R>(l);
  // fused sweeps run in 256-thread blocks, and so does the streamed per-chain factor path at any sweep count: its
  // wavefronts run one per SIMD anyway, and four of them share one copy of the energy's LDS tables
  const bool per_chain = l.cov_kind == CK_PER_CHAIN;
  const int threads = (l.n_sweeps >= kFusedSweepsThreshold || (ME_PER_CHAIN == 2 && per_chain)) ? kFusedStepThreads : kStepThreads;
  const dim3 grid(grid_for(l.n, l.grid_blocks, threads)), block(threads);
  if (l.inj_normals) {
    // injected-stream replay: float64 only (it exists to check trajectories against the float64 reference)
    if constexpr (std::is_same<R, double>::value) return launch_step_group<R, Energy, true>(l, a, en, grid, block, stream);
    else return hipErrorNotSupported;
  }
#if ME_DENSE && !defined(ME_USER_SOURCE)
  // BASELINE config 4: 64 real parameters, dense quadratic form, fp32 -> matrix-core kernels: split-bf16 on the matrix
  // pipe (me_dense_bf16x3.h) or, with METROPOLIS_DENSE64_FP32_MFMA=1, the fp32 MFMA form (me_dense_mfma.h)
  if constexpr (std::is_same<R, float>::value && NR == 64 && NC == 0 &&
                std::is_same<Energy, EnergyDense<float, 64, 0>>::value) {
    // per-chain proposal shapes (cov_mode = reference at 64 parameters) take the generic kernel with streamed factors
    if (per_chain) return launch_step_group<R, Energy, false>(l, a, en, grid, block, stream);
    if (dense64_exact_fp32_mfma()) {
      if (l.cov_kind == CK_IDENTITY)
        return launch_step_dense64_mfma<CK_IDENTITY>(a, en.a, nullptr, l.grid_blocks, stream);
      if (l.cov_kind == CK_SHARED && l.factor_full)
        return launch_step_dense64_mfma<CK_SHARED>(a, en.a, (const float *)l.factor_full, l.grid_blocks, stream);
      return hipErrorInvalidValue;
    }
    if (!l.energy_image) return hipErrorInvalidValue;
    if (l.cov_kind == CK_IDENTITY)
      return launch_step_dense64_bf16x3<CK_IDENTITY>(a, (const unsigned int *)l.energy_image, nullptr, l.grid_blocks, stream);
    if (l.cov_kind == CK_SHARED && l.factor_image)
      return launch_step_dense64_bf16x3<CK_SHARED>(a, (const unsigned int *)l.energy_image,
                                                   (const unsigned int *)l.factor_image, l.grid_blocks, stream);
    return hipErrorInvalidValue;
  } else if constexpr (std::is_same<R, double>::value && NR == 64 && NC == 0 &&
                       std::is_same<Energy, EnergyDense<double, 64, 0>>::value) {
    // ... and at the reference's precision: v_mfma_f64_16x16x4_f64 on the folded lower triangle (me_dense_f64.h)
    if (per_chain) return launch_step_group<R, Energy, false>(l, a, en, grid, block, stream);
    if (!l.energy_image) return hipErrorInvalidValue;
    if (l.cov_kind == CK_IDENTITY)
      return launch_step_dense64_f64<CK_IDENTITY>(a, (const double *)l.energy_image, nullptr, l.grid_blocks, stream);
    if (l.cov_kind == CK_SHARED && l.factor_image)
      return launch_step_dense64_f64<CK_SHARED>(a, (const double *)l.energy_image, (const double *)l.factor_image,
                                                l.grid_blocks, stream);
    return hipErrorInvalidValue;
  } else
#endif
    return launch_step_group<R, Energy, false>(l, a, en, grid, block, stream);
}

// the magnitude-phase complex sampler (me_magphase.h); a.factor carries the covariance field
template <typename R, class Energy>
hipError_t magphase_with(const StepLaunch &l, const Energy &en, hipStream_t stream) {
#if ME_PER_CHAIN
  if constexpr (NC > 0) {
    StepArgs<R> a = typed<R>(l);
    a.factor = (const R *)l.cov;
    const dim3 grid(grid_for(l.n, l.grid_blocks)), block(kStepThreads);
    if (l.inj_normals) {
      if constexpr (std::is_same<R, double>::value)
        hipLaunchKernelGGL((k_step_magphase<R, NR, NC, Energy, true>), grid, block, 0, stream, a, en);
      else
        return hipErrorNotSupported;
    } else {
      hipLaunchKernelGGL((k_step_magphase<R, NR, NC, Energy, false>), grid, block, 0, stream, a, en);
    }
    return hipGetLastError();
  }
#endif
  return hipErrorNotSupported;
}

// Build the by-value energy functor from the coefficient list and hand it to `f`.
template <typename R, class F>
hipError_t with_energy(int kind, const double *coef, int n_coef, const void *coef_device, F &&f) {
#ifdef ME_USER_SOURCE
  if (kind == ME_ENERGY_USER) {
    EnergyUser<R, NR, NC> en{(const R *)coef_device};
    return f(en);
  }
  if (kind == ME_ENERGY_USER_INDIRECT) {
    // the pointer is an address inside THIS device's copy of the plugin's code object: resolve it per device
    static PerDevice<user_fn_t<R>> fn_cache;
    int device = 0;
    if (hipError_t rc = hipGetDevice(&device); rc != hipSuccess) return rc;
    const user_fn_t<R> fn = fn_cache.get(device, [] { return load_user_fn<R>(); });
    if (!fn) return hipErrorInvalidDeviceFunction;
    EnergyUserIndirect<R, NR, NC> en{fn, (const R *)coef_device};
    return f(en);
  }
  return hipErrorInvalidValue;
#else
  switch (kind) {
    case ME_ENERGY_ISO_QUAD: {
      if (n_coef != 1) return hipErrorInvalidValue;
      EnergyIso<R, NR, NC> en{(R)coef[0]};
      return f(en);
    }
    case ME_ENERGY_DIAG_QUAD: {
      if (n_coef != NR + NC) return hipErrorInvalidValue;
      EnergyDiag<R, NR, NC> en;
      for (int i = 0; i < NR; ++i) en.w[i] = (R)coef[i];
      for (int j = 0; j < NC; ++j) en.w[NR + j] = en.w[NR + NC + j] = (R)coef[NR + j];
      return f(en);
    }
#if ME_DENSE
    case ME_ENERGY_DENSE_QUAD: {
      if (n_coef != D * D || !coef_device) return hipErrorInvalidValue;
      EnergyDense<R, NR, NC> en{(const R *)coef_device};
      return f(en);
    }
#endif
    case ME_ENERGY_LANDAU_TOY: {
      if constexpr (kLandau) {
        if (n_coef != 3) return hipErrorInvalidValue;
        EnergyLandau<R, NR, NC> en{(R)coef[0], (R)coef[1], (R)coef[2]};
        return f(en);
      }
      return hipErrorInvalidValue;
    }
    case ME_ENERGY_LANDAU_TERMS: {
      if constexpr (kLandau) {
        if (n_coef != 3) return hipErrorInvalidValue;
        EnergyLandauTerms<R, NR, NC> en{(R)coef[0], (R)coef[1], (R)coef[2]};
        return f(en);
      }
      return hipErrorInvalidValue;
    }
    case ME_ENERGY_CYLINDER: {
      if constexpr (kCylinder) {
        if (n_coef != 3) return hipErrorInvalidValue;
        EnergyCylinder<R, NR, NC> en{(R)coef[0], (R)coef[1], (R)coef[2]};
        return f(en);
      }
      return hipErrorInvalidValue;
    }
    default: return hipErrorInvalidValue;
  }
#endif
}

template <typename R>
hipError_t step(const StepLaunch &l, hipStream_t stream) {
  return with_energy<R>(l.energy_kind, l.coef_host, l.n_coef, l.coef_device,
                        [&](const auto &en) { return step_with<R>(l, en, stream); });
}

template <typename R>
hipError_t magphase(const StepLaunch &l, hipStream_t stream) {
  return with_energy<R>(l.energy_kind, l.coef_host, l.n_coef, l.coef_device,
                        [&](const auto &en) { return magphase_with<R>(l, en, stream); });
}

template <typename R>
hipError_t init_energy(const EnergyLaunch &l, hipStream_t stream) {
  return with_energy<R>(l.energy_kind, l.coef_host, l.n_coef, l.coef_device, [&](const auto &en) {
    using Energy = std::decay_t<decltype(en)>;
    hipLaunchKernelGGL((k_init_energy<R, NR, NC, Energy>), dim3(grid_for(l.n, l.grid_blocks)), dim3(kStepThreads), 0,
                       stream, (const R *)l.x, (R *)l.energy, l.n, l.status, en, l.total_row);
    return hipGetLastError();
  });
}

template <typename R>
MeasureArgs<R> typed_measure(const MeasureLaunch &l) {
  MeasureArgs<R> a;
  a.x = (const R *)l.x;
  a.width = (const R *)l.width;
  a.mean = (R *)l.mean;
  a.cov = (R *)l.cov;
  a.obs_mean = (R *)l.obs_mean;
  a.factor = (R *)l.factor;
  a.status = l.status;
  a.n = l.n;
  const double i = (double)l.measure_count;
  a.keep = (R)((i - 1.0) / i);
  a.inv_i = (R)(1.0 / i);
  a.cov_keep = (R)((i - 2.0) / (i - 1.0));
  a.update_cov = l.update_cov;
  a.split_widths = l.split_widths;
  a.write_factor = l.write_factor;
  return a;
}

// n_sweeps x step_all + measure in one launch (k_cycle, me_device.h) for the kernel sets whose packed matrix lives in
// registers; anything else -- group steps, injected streams, a shared factor, engines that keep no per-chain covariance,
// the matrix-core kernels of the dense 64-parameter form -- answers hipErrorNotSupported and me_cycle issues two launches.
#ifndef ME_FACTOR_TILE
#define ME_FACTOR_TILE 1      // 0 (dev): the one-lane-per-chain k_factor_stream also for 64 parameters and fewer
#endif
#ifndef ME_CYCLE
#define ME_CYCLE (ME_PER_CHAIN == 1)
#endif
template <typename R>
hipError_t cycle(const StepLaunch &l, const MeasureLaunch &ml, hipStream_t stream) {
#if ME_CYCLE
  constexpr int P = NR * (NR + 1) / 2 + NC * NC;
  if constexpr (P <= kMaxPackedInRegisters) {
    if (l.group != GROUP_ALL || l.inj_normals || !ml.cov || !ml.factor) return hipErrorNotSupported;
    if (l.cov_kind != CK_IDENTITY && l.cov_kind != CK_PER_CHAIN) return hipErrorNotSupported;
    return with_energy<R>(l.energy_kind, l.coef_host, l.n_coef, l.coef_device, [&](const auto &en) -> hipError_t {
      using Energy = std::decay_t<decltype(en)>;
#if ME_DENSE && !defined(ME_USER_SOURCE)
      if constexpr (NR == 64 && NC == 0 && std::is_same<Energy, EnergyDense<R, 64, 0>>::value) return hipErrorNotSupported;
#endif
      const StepArgs<R> a = typed<R>(l);
      const MeasureArgs<R> ma = typed_measure<R>(ml);
      // cache policy by size, as in k_step / k_measure: everything the launch touches per chain
      constexpr long long mean_bytes = (long long)sizeof(R) * (2 * D + (2 * NR + NC));
      constexpr long long all_bytes = mean_bytes + (long long)sizeof(R) * (2 + 2 * P);
      const bool nt = all_bytes * l.n > cache_budget_bytes(), ntm = mean_bytes * l.n > cache_budget_bytes();
      const int threads = l.n_sweeps >= kFusedSweepsThreshold ? kFusedStepThreads : kStepThreads;
      const dim3 grid(grid_for(l.n, l.grid_blocks, threads)), block(threads);
      auto launch = [&](auto ck) {
        constexpr int CK = decltype(ck)::value;
        if (ntm) hipLaunchKernelGGL((k_cycle<R, NR, NC, Energy, CK, true, true>), grid, block, 0, stream, a, ma, en);
        else if (nt) hipLaunchKernelGGL((k_cycle<R, NR, NC, Energy, CK, true, false>), grid, block, 0, stream, a, ma, en);
        else hipLaunchKernelGGL((k_cycle<R, NR, NC, Energy, CK, false, false>), grid, block, 0, stream, a, ma, en);
      };
      if (l.cov_kind == CK_PER_CHAIN) launch(std::integral_constant<int, CK_PER_CHAIN>{});
      else launch(std::integral_constant<int, CK_IDENTITY>{});
      return hipGetLastError();
    });
  }
#endif
  return hipErrorNotSupported;
}

template <typename R>
hipError_t measure(const MeasureLaunch &l, hipStream_t stream) {
  const MeasureArgs<R> a = typed_measure<R>(l);
  // small packed matrices: one fused launch; large ones: streaming update, then the factor kernel (see k_measure)
  constexpr int P = NR * (NR + 1) / 2 + NC * NC;
  constexpr bool kFused = ME_PER_CHAIN == 1 && P <= ME_MEASURE_FUSED_MAX_P;
  const dim3 grid(grid_for(l.n, l.grid_blocks)), block(kStepThreads);
  // Cache policy by size (k_measure's comment in me_device.h): the packed fields go non-temporal when the whole working
  // set (x, means, observables, covariance, factor) exceeds the Infinity Cache; the means and observables too when even
  // x + means + observables do.
  constexpr long long mean_bytes = (long long)sizeof(R) * (2 * D + (2 * NR + NC));
  constexpr long long all_bytes = mean_bytes + (long long)sizeof(R) * 2 * P;
  const bool nt = all_bytes * l.n > cache_budget_bytes(), ntm = mean_bytes * l.n > cache_budget_bytes();
  constexpr bool kCov = (ME_PER_CHAIN != 0 || ME_TRACK_COV != 0);
  if (l.cov) {
    if (ntm) hipLaunchKernelGGL((k_measure<R, NR, NC, kCov, kFused, true, true>), grid, block, 0, stream, a);
    else if (nt) hipLaunchKernelGGL((k_measure<R, NR, NC, kCov, kFused, true, false>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((k_measure<R, NR, NC, kCov, kFused, false, false>), grid, block, 0, stream, a);
  } else {
    if (ntm) hipLaunchKernelGGL((k_measure<R, NR, NC, false, false, false, true>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((k_measure<R, NR, NC, false, false, false, false>), grid, block, 0, stream, a);
  }
#if ME_PER_CHAIN == 1
  if constexpr (!kFused) {
    if (l.update_cov && l.write_factor) {
      if (nt) hipLaunchKernelGGL((k_factor<R, NR, NC, true>), grid, block, 0, stream, (const R *)l.cov, (R *)l.factor, l.status, l.n);
      else hipLaunchKernelGGL((k_factor<R, NR, NC, false>), grid, block, 0, stream, (const R *)l.cov, (R *)l.factor, l.status, l.n);
    }
  }
#elif ME_PER_CHAIN == 2
  static_assert(P > kMaxPackedInRegisters, "ME_PER_CHAIN=2 is for spaces with more than 160 packed entries");
  if (l.cov && l.update_cov && l.write_factor) {
    if constexpr (NC > 0) {       // a complex block: the plain one-lane-per-chain form (k_factor_mixed, me_device.h)
      if (nt) hipLaunchKernelGGL((k_factor_mixed<R, NR, NC, true>), grid, block, 0, stream, (const R *)l.cov, (R *)l.factor, l.status, l.n);
      else hipLaunchKernelGGL((k_factor_mixed<R, NR, NC, false>), grid, block, 0, stream, (const R *)l.cov, (R *)l.factor, l.status, l.n);
    } else if constexpr (NR <= 64 && ME_FACTOR_TILE) {     // a lane group per chain, a lane per row: every matrix read once (me_factor_tile.h)
      const hipError_t err = nt ? launch_factor_tile<R, NR, true>((const R *)l.cov, (R *)l.factor, l.status, l.n, stream)
                                : launch_factor_tile<R, NR, false>((const R *)l.cov, (R *)l.factor, l.status, l.n, stream);
      if (err != hipSuccess) return err;
    } else {                      // one lane per chain, finished rows re-read from global memory
      if (nt) hipLaunchKernelGGL((k_factor_stream<R, NR, true>), grid, block, 0, stream, (const R *)l.cov, (R *)l.factor, l.status, l.n);
      else hipLaunchKernelGGL((k_factor_stream<R, NR, false>), grid, block, 0, stream, (const R *)l.cov, (R *)l.factor, l.status, l.n);
    }
  }
#endif
  return hipGetLastError();
}

#ifdef ME_USER_SOURCE
#define ME_SET_NAME ME_USER_NAME
#else
#define ME_SET_NAME nullptr
#endif
// dimension-specific first stage of the pooled-moment reduction (float32): a Gram product on the matrix cores
constexpr auto pool_stage1_f32() -> hipError_t (*)(const void *, long long, double *, int, hipStream_t) {
  if constexpr (NR == 64 && NC == 0) return launch_pool_gram64;
  else if constexpr (D + NR + NC <= 32) return launch_pool_gram32<NR, NC>;
  else return nullptr;
}
constexpr auto pool_stage1_f64() -> hipError_t (*)(const void *, long long, double *, int, hipStream_t) {
  if constexpr (NR == 64 && NC == 0) return launch_pool_gram64_f64;
  else if constexpr (D + NR + NC <= 32) return launch_pool_gram32_f64<NR, NC>;
  else return nullptr;
}
#if ME_DENSE && !defined(ME_USER_SOURCE)
hipError_t prepare_matrix_f32(const void *factor_full, void *image, hipStream_t stream) {
  if constexpr (NR == 64 && NC == 0) {
    hipLaunchKernelGGL(k_dense64_bf16_fragments<0>, dim3(1), dim3(256), 0, stream, (const float *)factor_full,
                       (unsigned int *)image);
    return hipGetLastError();
  }
  return hipSuccess;
}
[[maybe_unused]] constexpr size_t kMatrixImageBytes = (NR == 64 && NC == 0) ? sizeof(unsigned int) * kBf16FragWords : 0;
#define ME_PREPARE_MATRIX_F32 kMatrixImageBytes, (kMatrixImageBytes ? prepare_matrix_f32 : nullptr), pool_stage1_f32()
hipError_t prepare_matrix_f64(const void *matrix, void *image, hipStream_t stream) {
  if constexpr (NR == 64 && NC == 0) {
    hipLaunchKernelGGL(k_dense64_f64_fragments<0>, dim3(1), dim3(256), 0, stream, (const double *)matrix, (double *)image);
    return hipGetLastError();
  }
  return hipSuccess;
}
[[maybe_unused]] constexpr size_t kMatrixImageBytesF64 = (NR == 64 && NC == 0) ? sizeof(double) * kDense64F64ImageDoubles : 0;
#define ME_PREPARE_MATRIX_F64 kMatrixImageBytesF64, (kMatrixImageBytesF64 ? prepare_matrix_f64 : nullptr), pool_stage1_f64()
#else
#define ME_PREPARE_MATRIX_F32 0, nullptr, pool_stage1_f32()
#define ME_PREPARE_MATRIX_F64 0, nullptr, pool_stage1_f64()
#endif
#ifdef ME_USER_HAS_REJECT
constexpr bool kHasUserReject = true;
#else
constexpr bool kHasUserReject = false;
#endif
// ME_ONLY_DTYPE=32 / 64 compiles one precision of the set (build.py splits the main library's sets in two objects so
// that the two halves build in parallel; plugins are one object)
#ifndef ME_ONLY_DTYPE
#define ME_ONLY_DTYPE 0
#endif
#if ME_ONLY_DTYPE != 64
const KernelSet kSetF32 = {ME_SET_NAME, ME_F32, NR, NC, ME_PER_CHAIN != 0, (ME_PER_CHAIN != 0 || ME_TRACK_COV != 0), has_energy, energy_terms, step<float>,
                           (NC > 0 && ME_PER_CHAIN == 1) ? magphase<float> : nullptr, measure<float>, init_energy<float>,
                           ME_PREPARE_MATRIX_F32, kHasUserReject, ME_PER_CHAIN == 2, cycle<float>, D >= kTiledStateDof};
#endif
#if ME_ONLY_DTYPE != 32
const KernelSet kSetF64 = {ME_SET_NAME, ME_F64, NR, NC, ME_PER_CHAIN != 0, (ME_PER_CHAIN != 0 || ME_TRACK_COV != 0), has_energy, energy_terms, step<double>,
                           (NC > 0 && ME_PER_CHAIN == 1) ? magphase<double> : nullptr, measure<double>, init_energy<double>,
                           ME_PREPARE_MATRIX_F64, kHasUserReject, ME_PER_CHAIN == 2, cycle<double>, D >= kTiledStateDof};
#endif

struct Registrar {
  Registrar() {
#if ME_ONLY_DTYPE != 64
    register_kernel_set(&kSetF32);
#endif
#if ME_ONLY_DTYPE != 32
    register_kernel_set(&kSetF64);
#endif
  }
} registrar;

}  // namespace
}  // namespace me
