// Internal interface between the C-ABI layer (me_api.hip) and the per-dimension kernel sets
// (me_kernels.hip compiled once per (n_real, n_complex)).  Not installed; the public ABI is
// include/metropolis_engine.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/metropolis_engine.h"

namespace me {

// How the step kernel obtains the proposal shape.
// CK_PER_CHAIN_NT is a kernel variant, not an engine state: the per-chain factor is read non-temporally (working sets
// beyond the Infinity Cache, me_kernels.hip: launch_step_cov).
enum CovKind { CK_IDENTITY = 0, CK_SHARED = 1, CK_PER_CHAIN = 2, CK_PER_CHAIN_NT = 3 };

// which coordinates a step launch moves (row index of the adapting width in mixed engines)
enum StepGroup { GROUP_ALL = 0, GROUP_REAL = 1, GROUP_COMPLEX = 2 };

enum StatusBits : uint32_t {
  ST_NONFINITE_ENERGY = 1u,  // a proposed state that was not wall-rejected had a non-finite energy
  ST_BAD_PIVOT = 2u,         // Cholesky pivot <= 0 while refreshing a proposal factor
  ST_BAD_WIDTH = 4u          // sampling width <= 0 or non-finite (metropolis_engine.py:438)
};

// A kernel whose per-launch working set stays below this is served from the 256 MiB Infinity Cache from launch to
// launch (measured: 151 MB and 226 MB sets stay resident, MI355X_MICROARCH.md); larger ones stream from HBM and their
// read-once / write-once fields are accessed non-temporally.  A process-wide tuning knob (me_set_cache_budget).
long long cache_budget_bytes();
void set_cache_budget_bytes(long long bytes);

constexpr int kBlockThreads = 256;  // block size of the dimension-independent kernels (me_generic.hip)
// Block size of the per-chain kernels (k_step, k_measure, ...): one lane owns one chain.  Measured on MI355X at
// 2^20 chains x 16 parameters, one sweep per launch: 64 threads 23.8 us, 128: 24.1, 256: 25.1, 512: 27.4 -- these
// kernels share nothing inside a block, so the finest dispatch granularity gives the shortest ramp-up and tail.
#ifndef ME_STEP_THREADS
#define ME_STEP_THREADS 64
#endif
constexpr int kStepThreads = ME_STEP_THREADS;
// k_step launches with several fused sweeps are VALU-bound and ran ~5 % faster with 256-thread blocks
constexpr int kFusedStepThreads = 256;
constexpr int kFusedSweepsThreshold = 4;

// Type-erased launch descriptors; scalars are doubles and are narrowed by the typed launcher.
struct StepLaunch {
  int n_real, n_complex;     // for the runtime-dimension kernel set (compile-time constants everywhere else)
  void *x, *energy, *width;
  const void *factor;        // CK_PER_CHAIN: [P][N]; CK_SHARED: [P]
  const void *factor_full;   // CK_SHARED on pure-real engines: the same factor as a dense [nr][nr] row-major matrix
  const void *factor_image;  // ... and as whatever KernelSet::prepare_matrix made of it (nullptr without that hook)
  const void *energy_image;  // ME_ENERGY_DENSE_QUAD: the coefficient matrix through the same hook (nullptr without it)
  const void *coef_device;   // energy coefficients in device memory (device dtype)
  const double *coef_host;   // same, host doubles
  int n_coef;
  const void *inj_normals;   // non-null: injected streams (f64 kernels only), [sweep][D][n] and [sweep][n]
  const void *inj_uniforms;
  unsigned long long *accept_slots;  // per-wavefront accepted counts, [grid_blocks * 4]
  unsigned int *status;
  long long n;
  unsigned long long chain_offset, step_index, seed;
  int n_sweeps, energy_kind, cov_kind, reject_kind, grid_blocks;
  int group;          // StepGroup: 0 = all (step_all), 1 = real group, 2 = complex group (mixed engines)
  int split_widths;   // mixed engines: rows 1, 2 of the width field differ from row 0
  int stale_total;    // ME_FLAG_REFERENCE_ENERGY_LEDGERS: step_all of a mixed engine uses the extra ledger row only
  const void *cov;    // per-chain covariance field (magnitude-phase sampler reads its diagonal)
  double reject_bound, temp, ratio, target_acceptance, damping;
};

struct MeasureLaunch {
  int n_real, n_complex;
  void *x, *width, *mean, *cov, *obs_mean, *factor;
  unsigned int *status;
  long long n;
  unsigned long long measure_count;  // value AFTER the increment (metropolis_engine.py:343)
  int update_cov;                    // measure_count > 50 (:389, :396)
  int split_widths;                  // see StepLaunch
  int write_factor;                  // refresh the per-chain Cholesky factors
  int grid_blocks;
};

struct EnergyLaunch {
  int n_real, n_complex;
  int total_row;      // >= 0: also write the sum of the terms into this ledger row (the reference's energy_total)
  void *x, *energy;
  const void *coef_device;
  const double *coef_host;
  int n_coef;
  unsigned int *status;
  long long n;
  int energy_kind, grid_blocks;
};

// largest n_real + 2 n_complex of the register-resident kernel sets (build.py: MAX_REGISTER_DOF); beyond it
// find_kernel_set falls back to the runtime-dimension set (me_runtime_dims.hip), registered with n_real = n_complex = -1
constexpr int kMaxRegisterDof = 96;

struct KernelSet {
  const char *user_name;  // nullptr for the built-in sets; the plugin's name for user-energy sets
  int dtype, n_real, n_complex;
  bool per_chain_cov;  // measure can refresh per-chain factors / step can read them
  bool tracks_cov;     // measure can maintain the per-chain running covariance (always true with per_chain_cov;
                       // alone: statistics only, for matrices too large for the factor kernels)
  bool (*has_energy)(int energy_kind);
  int (*energy_terms)(int energy_kind);   // rows of the energy ledger (1 unless the energy is a term dictionary)
  hipError_t (*step)(const StepLaunch &, hipStream_t);
  hipError_t (*magphase)(const StepLaunch &, hipStream_t);   // nullptr: no complex group / no per-chain covariance
  hipError_t (*measure)(const MeasureLaunch &, hipStream_t);
  hipError_t (*init_energy)(const EnergyLaunch &, hipStream_t);
  // optional: turn a dense [D][D] device matrix (the shared proposal factor, the dense energy's coefficients) into a
  // kernel-specific operand image (matrix_image_bytes of device memory owned by the engine); run whenever the
  // matrix is uploaded
  size_t matrix_image_bytes;
  hipError_t (*prepare_matrix)(const void *matrix, void *image, hipStream_t);
  // optional: a dimension-specific first stage of the pooled-moment reduction writing partials[row][entry] for
  // n_rows rows in k_pool_finish's entry order (nullptr: the generic k_pool_reduce)
  hipError_t (*pool_stage1)(const void *x, long long n, double *partials, int n_rows, hipStream_t);
  // user-energy sets only: the plugin source defines me_user_reject (ME_USER_HAS_REJECT), so ME_REJECT_USER means something
  bool has_user_reject;
  // per_chain_cov with the packed matrices STREAMED (more than 160 entries, pure real spaces): the covariance and factor
  // fields are only kept for ME_COV_REFERENCE (or the tracking flag), are walked with 64-bit pointers and may pass 4 GiB
  bool streams_packed;
  // optional: n_sweeps x step_all + measure in ONE launch (k_cycle); nullptr or hipErrorNotSupported: me_cycle issues the
  // step launch and the measure launch instead (same results)
  hipError_t (*cycle)(const StepLaunch &, const MeasureLaunch &, hipStream_t);
  // the kernels of this set address the chain state x TILE-major ([tile of 64 chains][D][64 lanes], me_device.h: XField)
  // instead of component-major: the register-resident sets from kTiledStateDof real degrees of freedom on
  bool tiled_state;
};

void register_kernel_set(const KernelSet *set);
// first registered set for these dimensions that implements `energy_kind` (user kinds: with this plugin name)
const KernelSet *find_kernel_set(int dtype, int n_real, int n_complex, int energy_kind, const char *user_name);
bool has_dims(int dtype, int n_real, int n_complex);

// ---- dimension-independent kernels (me_generic.hip) -------------------------------------------------------
// dst[r*n + c] = row_values[r] for r < rows, c < n   (broadcast one chain's vector to all chains)
hipError_t launch_broadcast_rows(void *dst, const void *row_values, int rows, long long n, int dtype,
                                 hipStream_t stream);
// dst[((c >> 6) * entries + k) * 64 + (c & 63)] = entry_values[k]: the tile-major packed covariance / factor fields
hipError_t launch_broadcast_tiled(void *dst, const void *entry_values, int entries, long long n, int dtype,
                                  hipStream_t stream);
// Ensemble sums (see me_pooled_moments in the public header).  out must hold moments_size doubles; `slots` are the
// per-wavefront acceptance counters (summed by the finishing kernel), proposed is host-known.
hipError_t launch_pool_reduce(const void *x, long long n, int n_real, int n_complex, int dtype,
                              const unsigned long long *slots, long long n_slots, double proposed, double *partials,
                              double *out_device, hipStream_t stream,
                              hipError_t (*stage1)(const void *, long long, double *, int, hipStream_t) = nullptr,
                              bool tiled_state = false);
// blocks of the first reduction stage; partials must hold pool_reduce_blocks(...) * (1 + D + nr + nc + D(D+1)/2) doubles
int pool_reduce_blocks(long long n, int n_real, int n_complex);
// false: the dimensions are beyond the pooled-moment kernels (me_pooled_moments then returns ME_ERR_UNSUPPORTED)
bool pool_reduce_supported(int n_real, int n_complex, int dtype);
// Time-series row of the traced chains (chain t*stride, t < n_traced): out[col][t] as doubles with columns
// [params (d) | energy terms (n_terms) | widths (width_rows)]  -- what measure() appends in the reference (:350-356).
hipError_t launch_trace(const void *x, const void *energy, const void *width, long long n, int d, int n_terms,
                        int width_rows, int dtype, long long n_traced, long long stride, double *out, hipStream_t stream,
                        bool tiled_state = false);
// total[0] = sum of slots[0 .. n_slots)
// Batched equilibration detection (me_statistics.hip; me_detect_equilibration in the public header).
hipError_t launch_detect_equilibration(const double *series, long long n_series, long long length, int fast, int nskip,
                                       double *scratch, long long *t0_out, double *g_out, double *neff_out,
                                       hipStream_t stream);
hipError_t launch_sum_slots(const unsigned long long *slots, long long n_slots, unsigned long long *total,
                            hipStream_t stream);

// blocks launched for n chains (one lane per chain, grid-stride beyond `requested` blocks when requested > 0)
inline int grid_for(long long n, int requested, int threads = kStepThreads) {
  long long blocks = (n + threads - 1) / threads;
  if (requested > 0 && blocks > requested) blocks = requested;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

__host__ __device__ inline int packed_real(int nr) { return nr * (nr + 1) / 2; }
__host__ __device__ inline int packed_total(int nr, int nc) { return nr * (nr + 1) / 2 + nc * nc; }
__host__ __device__ inline int n_observables(int nr, int nc) { return 2 * nr + nc; }
__host__ __device__ inline long long moments_size(int nr, int nc) {
  const int d = nr + 2 * nc;
  return 1 + d + (long long)d * (d + 1) / 2 + n_observables(nr, nc) + 2;
}

}  // namespace me
