// C-ABI of libmetropolis_hip.so (declared in include/metropolis_engine.h): engine lifetime, launches, field
// transfer.  Host logic here mirrors the constructor and counters of the reference class
// (/root/reference/metropolisengine/metropolis_engine.py:17-133); the arithmetic lives in the kernels.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <mutex>
#include <string>
#include <atomic>
#include <chrono>
#include <vector>

#include "me_comm.h"
#include "me_internal.h"

namespace me {

// ------------------------------------------------------------------------------------------------ registry
// Kernel sets register themselves from static constructors (the library's own and those of plugins loaded later, possibly
// while other threads create engines): one mutex guards the list.
static std::vector<const KernelSet *> &registry() {
  static std::vector<const KernelSet *> sets;
  return sets;
}
static std::mutex &registry_mutex() {
  static std::mutex m;
  return m;
}
void register_kernel_set(const KernelSet *set) {
  std::lock_guard<std::mutex> lock(registry_mutex());
  registry().push_back(set);
}
static bool is_user_kind(int kind) { return kind == ME_ENERGY_USER || kind == ME_ENERGY_USER_INDIRECT; }
const KernelSet *find_kernel_set(int dtype, int n_real, int n_complex, int energy_kind, const char *user_name) {
  std::lock_guard<std::mutex> lock(registry_mutex());
  for (const KernelSet *s : registry()) {
    if (s->dtype != dtype || s->n_real != n_real || s->n_complex != n_complex || !s->has_energy(energy_kind)) continue;
    if (is_user_kind(energy_kind)) {
      if (!s->user_name || !user_name || std::strcmp(s->user_name, user_name) != 0) continue;
    }
    return s;
  }
  // beyond the register-resident kernels: the runtime-dimension set (any dimensions, separable energies)
  if (n_real + 2 * n_complex > kMaxRegisterDof && !is_user_kind(energy_kind))
    for (const KernelSet *s : registry())
      if (s->dtype == dtype && s->n_real < 0 && s->has_energy(energy_kind)) return s;
  return nullptr;
}
bool has_dims(int dtype, int n_real, int n_complex) {
  if (n_real + 2 * n_complex > kMaxRegisterDof) return true;
  std::lock_guard<std::mutex> lock(registry_mutex());
  for (const KernelSet *s : registry())
    if (s->dtype == dtype && s->n_real == n_real && s->n_complex == n_complex && !s->user_name) return true;
  return false;
}

// ------------------------------------------------------------------------------------------------ helpers
static thread_local std::string g_create_error;

// Inverse standard normal CDF by Halley iteration on erfc (no tabulated coefficients).
static double norm_ppf(double q) {
  double x = 0.0;
  // crude start: logistic approximation
  x = std::log(q / (1.0 - q)) / 1.702;
  for (int it = 0; it < 50; ++it) {
    const double cdf = 0.5 * std::erfc(-x / std::sqrt(2.0));
    const double pdf = std::exp(-0.5 * x * x) / std::sqrt(2.0 * M_PI);
    const double f = cdf - q;
    const double step = f / (pdf * (1.0 + 0.5 * x * f / pdf));  // Halley: f / (f' - f f''/(2 f')), f'' = -x pdf
    x -= step;
    if (std::fabs(step) < 1e-16 * std::max(1.0, std::fabs(x))) break;
  }
  return x;
}

// In-place packed Cholesky of the initial proposal matrices (host, double).  Returns false on a pivot <= 0
// (numpy raises ValueError for a non-PSD real matrix, metropolis_engine.py:270).
static bool host_factor(std::vector<double> &m, int nr, int nc) {
  auto tri = [](int i, int j) { return i * (i + 1) / 2 + j; };
  const int pr = nr * (nr + 1) / 2;
  auto cre = [pr](int i, int j) { return pr + i * i + 2 * j; };
  auto cim = [pr](int i, int j) { return pr + i * i + 2 * j + 1; };
  auto cdg = [pr](int i) { return pr + i * i + 2 * i; };
  for (int j = 0; j < nr; ++j) {
    double s = m[tri(j, j)];
    for (int k = 0; k < j; ++k) s -= m[tri(j, k)] * m[tri(j, k)];
    if (!(s > 0)) return false;
    const double dg = std::sqrt(s);
    m[tri(j, j)] = dg;
    for (int i = j + 1; i < nr; ++i) {
      double t = m[tri(i, j)];
      for (int k = 0; k < j; ++k) t -= m[tri(i, k)] * m[tri(j, k)];
      m[tri(i, j)] = t / dg;
    }
  }
  for (int j = 0; j < nc; ++j) {
    double s = m[cdg(j)];
    for (int k = 0; k < j; ++k) s -= m[cre(j, k)] * m[cre(j, k)] + m[cim(j, k)] * m[cim(j, k)];
    if (!(s > 0)) return false;
    const double dg = std::sqrt(s);
    m[cdg(j)] = dg;
    for (int i = j + 1; i < nc; ++i) {
      double tr = m[cre(i, j)], ti = m[cim(i, j)];
      for (int k = 0; k < j; ++k) {
        const double ar = m[cre(i, k)], ai = m[cim(i, k)], br = m[cre(j, k)], bi = m[cim(j, k)];
        tr -= ar * br + ai * bi;
        ti -= ai * br - ar * bi;
      }
      m[cre(i, j)] = tr / dg;
      m[cim(i, j)] = ti / dg;
    }
  }
  return true;
}

// whether the caller's initial proposal matrices (me_config.covariance_real / _complex) are the identity (:63-70)
static bool initial_shape_is_identity(const me_config *c) {
  if (c->covariance_real)
    for (int i = 0; i < c->n_real; ++i)
      for (int j = 0; j <= i; ++j)
        if (c->covariance_real[i * c->n_real + j] != (i == j ? 1.0 : 0.0)) return false;
  if (c->covariance_complex)
    for (int i = 0; i < c->n_complex; ++i)
      for (int j = 0; j <= i; ++j) {
        const double re = c->covariance_complex[2 * (i * c->n_complex + j)], im = c->covariance_complex[2 * (i * c->n_complex + j) + 1];
        if (re != (i == j ? 1.0 : 0.0) || (i != j && im != 0.0)) return false;
      }
  return true;
}

}  // namespace me

using namespace me;

struct me_engine {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  const KernelSet *ks = nullptr;
  int dtype = ME_F32;
  size_t esize = 4;
  long long n = 0;
  int nr = 0, nc = 0, d = 0, p = 0, nobs = 0;
  unsigned long long chain_offset = 0, seed = 0;
  double temp = 0, target_acceptance = 0.3, alpha = 0, ratio = 0, reject_bound = 0;
  int m = 0, energy_kind = 0, reject_kind = 0, cov_mode = 0;
  bool stale_total = false;          // ME_FLAG_REFERENCE_ENERGY_LEDGERS on a mixed engine: ledger row n_terms = energy_total
  hipEvent_t time_start = nullptr, time_stop = nullptr;   // me_time_steps
  std::vector<double> shared_host;   // the packed factor last given to me_set_shared_factor (empty: none); checkpoints
  int cov_kind = CK_IDENTITY;
  int grid_blocks = 0;
  int n_terms = 1;   // rows of the energy ledger (KernelSet::energy_terms)
  std::vector<double> coef;
  unsigned long long step_index = 0, measure_count = 1;   // counters start at 1 (metropolis_engine.py:72-75)
  unsigned long long fused_cycles = 0;                    // me_cycle calls that ran as ONE launch (k_cycle)
  // device buffers (SoA: component-major, chain-minor)
  void *x = nullptr, *energy = nullptr, *width = nullptr, *mean = nullptr, *cov = nullptr, *obs_mean = nullptr;
  void *factor = nullptr, *shared_factor = nullptr, *shared_full = nullptr, *shared_image = nullptr, *energy_image = nullptr, *coef_dev = nullptr,
       *row_dev = nullptr;
  unsigned long long *accept_slots = nullptr, *accept_total = nullptr;
  long long n_slots = 0;
  unsigned long long proposed = 0;
  bool x_tiled = false;         // the state field is tile-major (KernelSet::tiled_state)
  int width_rows = 1;           // 3 for mixed engines: [sampling_width, real group, complex group]
  bool widths_synced = true;    // mixed engines: rows 1, 2 are implied equal to row 0 (state after a step_all)
  unsigned int *status = nullptr;
  double *pool_dev = nullptr, *pool_partials = nullptr, *pool_host = nullptr;   // pool_host: pinned
  unsigned long long *host_scratch = nullptr;   // pinned: [0] status bits, [1] accepted total
  // split pooled-moment reduction (me_pooled_moments_begin/_end): second stream for the copy, two events
  hipStream_t copy_stream = nullptr;
  hipEvent_t pool_reduced = nullptr, pool_copied = nullptr;
  bool pool_pending = false;
  // RCCL communicator of this engine's rank (me_comm_init_rank); null = single-GPU engine
  const RcclApi *rccl = nullptr;
  ncclComm_t comm = nullptr;
  int comm_rank = 0, comm_world = 1;
  // time-series trace of a few chains (the reference's per-measure appends, :350-356)
  double *trace_dev = nullptr;
  long long trace_chains = 0, trace_stride = 1, trace_rows = 0, trace_capacity = 0;
  std::string err;
};

namespace {

int fail(me_engine *e, int code, const std::string &msg) {
  if (e) e->err = msg;
  else g_create_error = msg;
  return code;
}

#define ME_HIP(e, call)                                                                                      \
  do {                                                                                                       \
    hipError_t err__ = (call);                                                                               \
    if (err__ != hipSuccess)                                                                                 \
      return fail((e), ME_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(err__));                    \
  } while (0)

void to_device_type(const double *src, size_t count, int dtype, std::vector<unsigned char> &out) {
  out.resize(count * (dtype == ME_F32 ? 4 : 8));
  if (dtype == ME_F32) {
    float *o = reinterpret_cast<float *>(out.data());
    for (size_t i = 0; i < count; ++i) o[i] = (float)src[i];
  } else {
    std::memcpy(out.data(), src, count * 8);
  }
}

// broadcast one host row (doubles) to every chain of a component-major device field
// `tiled`: dst is one of the packed covariance / factor fields (tile-major, me_device.h: TiledField)
int broadcast(me_engine *e, void *dst, const std::vector<double> &row, bool tiled = false) {
  std::vector<unsigned char> bytes;
  to_device_type(row.data(), row.size(), e->dtype, bytes);
  ME_HIP(e, hipMemcpyAsync(e->row_dev, bytes.data(), bytes.size(), hipMemcpyHostToDevice, e->stream));
  if (tiled) ME_HIP(e, launch_broadcast_tiled(dst, e->row_dev, (int)row.size(), e->n, e->dtype, e->stream));
  else ME_HIP(e, launch_broadcast_rows(dst, e->row_dev, (int)row.size(), e->n, e->dtype, e->stream));
  ME_HIP(e, hipStreamSynchronize(e->stream));   // row_dev / bytes are reused by the next call
  return ME_OK;
}

int check_status(me_engine *e);
// Host <-> device copy of chains [chain_begin, chain_begin + n_chains) of a TILE-major packed field (covariance, factor:
// entry k of chain c at ((c >> 6) * entries + k) * 64 + (c & 63), me_device.h).  The chain range covers a contiguous run of
// whole tiles on the device: one copy each way, the gather / scatter into the caller's [chain][entry] doubles on the host.
// Exactly one of dst / src is non-null; a partial first / last tile is read back before it is rewritten.
int copy_tiled(me_engine *e, void *field_ptr, int entries, int64_t chain_begin, int64_t n_chains, double *dst, const double *src) {
  const int64_t tile0 = chain_begin >> 6, tile1 = (chain_begin + n_chains - 1) >> 6;
  const size_t tile_values = (size_t)entries * 64, n_values = (size_t)(tile1 - tile0 + 1) * tile_values;
  unsigned char *dev = (unsigned char *)field_ptr + (size_t)tile0 * tile_values * e->esize;
  std::vector<unsigned char> tmp(n_values * e->esize);
  const bool whole = (chain_begin & 63) == 0 && (((chain_begin + n_chains) & 63) == 0 || chain_begin + n_chains == e->n);
  if (dst || !whole) {
    ME_HIP(e, hipMemcpyAsync(tmp.data(), dev, tmp.size(), hipMemcpyDeviceToHost, e->stream));
    ME_HIP(e, hipStreamSynchronize(e->stream));
  }
  auto at = [&](int64_t c, int k) { return (size_t)((c >> 6) - tile0) * tile_values + (size_t)k * 64 + (size_t)(c & 63); };
  if (e->dtype == ME_F32) {
    float *t = reinterpret_cast<float *>(tmp.data());
    for (int64_t i = 0; i < n_chains; ++i)
      for (int k = 0; k < entries; ++k) {
        if (dst) dst[i * entries + k] = (double)t[at(chain_begin + i, k)];
        else t[at(chain_begin + i, k)] = (float)src[i * entries + k];
      }
  } else {
    double *t = reinterpret_cast<double *>(tmp.data());
    for (int64_t i = 0; i < n_chains; ++i)
      for (int k = 0; k < entries; ++k) {
        if (dst) dst[i * entries + k] = t[at(chain_begin + i, k)];
        else t[at(chain_begin + i, k)] = src[i * entries + k];
      }
  }
  if (src) {
    ME_HIP(e, hipMemcpyAsync(dev, tmp.data(), tmp.size(), hipMemcpyHostToDevice, e->stream));
    ME_HIP(e, hipStreamSynchronize(e->stream));
  }
  return dst ? check_status(e) : ME_OK;
}

int field_info(me_engine *e, int field, void **ptr, int *comps) {
  switch (field) {
    case ME_FIELD_PARAMS: *ptr = e->x; *comps = e->d; return ME_OK;
    case ME_FIELD_ENERGY: *ptr = e->energy; *comps = e->n_terms; return ME_OK;
    case ME_FIELD_WIDTH: *ptr = e->width; *comps = e->width_rows; return ME_OK;
    case ME_FIELD_MEAN: *ptr = e->mean; *comps = e->d; return ME_OK;
    case ME_FIELD_OBS_MEAN: *ptr = e->obs_mean; *comps = e->nobs; return ME_OK;
    case ME_FIELD_COV:
      if (!e->cov)
        return fail(e, ME_ERR_UNSUPPORTED, "this engine keeps no per-chain covariance: for large parameter spaces it is "
                                           "kept on request only (me_config.flags: ME_FLAG_TRACK_COVARIANCE), and not "
                                           "at all where the field would pass 4 GiB");
      *ptr = e->cov; *comps = e->p; return ME_OK;
    case ME_FIELD_ENERGY_TOTAL:
      if (!e->stale_total)
        return fail(e, ME_ERR_UNSUPPORTED, "this engine keeps one coherent energy ledger (sum ME_FIELD_ENERGY); a separate "
                                           "energy_total exists with ME_FLAG_REFERENCE_ENERGY_LEDGERS only");
      *ptr = (unsigned char *)e->energy + (size_t)e->n_terms * (size_t)e->n * e->esize; *comps = 1; return ME_OK;
    case ME_FIELD_FACTOR:
      if (!e->factor) return fail(e, ME_ERR_UNSUPPORTED, "this engine keeps no per-chain proposal factors (dimensions "
                                                         "without per-chain kernels, or a field beyond 4 GiB)");
      *ptr = e->factor; *comps = e->p; return ME_OK;
    default: return fail(e, ME_ERR_INVALID, "unknown field id");
  }
}

// Surface per-chain failure flags (the analogue of the reference's exceptions) and clear them.
// (reads go through a pinned scratch word: a pageable 4-byte copy costs ~10 us more per call)
int report_status(me_engine *e, unsigned int bits) {
  if (!bits) return ME_OK;
  ME_HIP(e, hipMemsetAsync(e->status, 0, sizeof(unsigned int), e->stream));
  std::string msg = "numeric failure in at least one chain:";
  if (bits & ST_NONFINITE_ENERGY) msg += " non-finite energy;";
  if (bits & ST_BAD_PIVOT) msg += " non-positive Cholesky pivot (proposal covariance not positive definite);";
  if (bits & ST_BAD_WIDTH) msg += " sampling width <= 0;";
  return fail(e, ME_ERR_NUMERIC, msg);
}
int check_status(me_engine *e) {
  ME_HIP(e, hipMemcpyAsync(&e->host_scratch[0], e->status, sizeof(unsigned int), hipMemcpyDeviceToHost, e->stream));
  ME_HIP(e, hipStreamSynchronize(e->stream));
  return report_status(e, (unsigned int)e->host_scratch[0]);
}

// Upload the shared proposal factor: packed (ME_FIELD_FACTOR layout) and, for pure-real engines, also as a dense
// row-major [nr][nr] lower-triangular matrix (operand of the matrix-core proposal kernel).
int upload_shared_factor(me_engine *e, const double *packed) {
  std::vector<unsigned char> bytes;
  to_device_type(packed, (size_t)e->p, e->dtype, bytes);
  ME_HIP(e, hipMemcpy(e->shared_factor, bytes.data(), bytes.size(), hipMemcpyHostToDevice));
  if (e->shared_full) {
    std::vector<double> full((size_t)e->nr * e->nr, 0.0);
    for (int i = 0; i < e->nr; ++i)
      for (int j = 0; j <= i; ++j) full[(size_t)i * e->nr + j] = packed[i * (i + 1) / 2 + j];
    to_device_type(full.data(), full.size(), e->dtype, bytes);
    ME_HIP(e, hipMemcpy(e->shared_full, bytes.data(), bytes.size(), hipMemcpyHostToDevice));
    if (e->shared_image) ME_HIP(e, e->ks->prepare_matrix(e->shared_full, e->shared_image, e->stream));
  }
  return ME_OK;
}

void fill_step_launch(me_engine *e, StepLaunch &l, int n_sweeps) {
  l.n_real = e->nr;
  l.n_complex = e->nc;
  l.x = e->x;
  l.energy = e->energy;
  l.width = e->width;
  l.factor = e->cov_kind == CK_PER_CHAIN ? e->factor : e->shared_factor;
  l.factor_full = e->shared_full;
  l.factor_image = e->shared_image;
  l.energy_image = e->energy_image;
  l.coef_device = e->coef_dev;
  l.coef_host = e->coef.data();
  l.n_coef = (int)e->coef.size();
  l.inj_normals = nullptr;
  l.inj_uniforms = nullptr;
  l.group = GROUP_ALL;
  l.split_widths = (e->width_rows == 3 && !e->widths_synced) ? 1 : 0;
  l.stale_total = e->stale_total ? 1 : 0;
  l.cov = e->cov;
  l.accept_slots = e->accept_slots;
  l.status = e->status;
  l.n = e->n;
  l.chain_offset = e->chain_offset;
  l.step_index = e->step_index;
  l.seed = e->seed;
  l.n_sweeps = n_sweeps;
  l.energy_kind = e->energy_kind;
  l.cov_kind = e->cov_kind;
  l.reject_kind = e->reject_kind;
  l.grid_blocks = e->grid_blocks;
  l.reject_bound = e->reject_bound;
  l.temp = e->temp;
  l.ratio = e->ratio;
  l.target_acceptance = e->target_acceptance;
  // step_number_factor = max(measure_step_counter / m, 200)   (metropolis_engine.py:430)
  l.damping = std::max((double)e->measure_count / (double)e->m, 200.0);
}

void release(me_engine *e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
  void *bufs[] = {e->x, e->energy, e->width, e->mean, e->cov, e->obs_mean, e->factor, e->shared_factor, e->shared_full, e->shared_image, e->energy_image,
                  e->coef_dev, e->row_dev, e->accept_slots, e->accept_total, e->status, e->pool_dev, e->pool_partials, e->trace_dev};
  for (void *b : bufs)
    if (b) (void)hipFree(b);
  if (e->comm && e->rccl) (void)e->rccl->comm_destroy(e->comm);
  if (e->pool_copied) (void)hipEventDestroy(e->pool_copied);
  if (e->pool_reduced) (void)hipEventDestroy(e->pool_reduced);
  if (e->copy_stream) (void)hipStreamDestroy(e->copy_stream);
  if (e->pool_host) (void)hipHostFree(e->pool_host);
  if (e->host_scratch) (void)hipHostFree(e->host_scratch);
  if (e->time_start) (void)hipEventDestroy(e->time_start);
  if (e->time_stop) (void)hipEventDestroy(e->time_stop);
  if (e->own_stream && e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
}

}  // namespace

namespace me {
namespace {
std::atomic<long long> g_cache_budget{224ll << 20};
}
long long cache_budget_bytes() { return g_cache_budget.load(std::memory_order_relaxed); }
void set_cache_budget_bytes(long long bytes) { g_cache_budget.store(bytes, std::memory_order_relaxed); }
}  // namespace me

extern "C" {

int me_set_cache_budget(int64_t bytes) {
  if (bytes < 0) return fail(nullptr, ME_ERR_INVALID, "cache budget must be >= 0");
  me::set_cache_budget_bytes(bytes);
  return ME_OK;
}

int me_abi_version(void) { return ME_ABI_VERSION; }

int me_supported(int32_t dtype, int32_t n_real, int32_t n_complex, int32_t energy_kind) {
  return find_kernel_set(dtype, n_real, n_complex, energy_kind, nullptr) ? 1 : 0;
}

int me_create(const me_config *c, me_engine **out) {
  if (!c || !out) return fail(nullptr, ME_ERR_INVALID, "null config or output pointer");
  *out = nullptr;
  if (c->abi_version != ME_ABI_VERSION) return fail(nullptr, ME_ERR_INVALID, "me_config.abi_version mismatch");
  if (c->n_real < 0 || c->n_complex < 0 || c->n_real + c->n_complex == 0)
    return fail(nullptr, ME_ERR_INVALID,
                "must give at least one real or complex parameter (metropolis_engine.py:37-39)");
  if (c->n_chains <= 0) return fail(nullptr, ME_ERR_INVALID, "n_chains must be positive");
  if (!(c->temp >= 0)) return fail(nullptr, ME_ERR_INVALID, "temp must be >= 0 (metropolis_engine.py:92)");
  if (!(c->target_acceptance > 0 && c->target_acceptance < 1))
    return fail(nullptr, ME_ERR_INVALID, "target_acceptance must be in (0, 1)");
  if (!(c->sampling_width > 0)) return fail(nullptr, ME_ERR_INVALID, "sampling_width must be > 0");
  if (c->dtype != ME_F32 && c->dtype != ME_F64) return fail(nullptr, ME_ERR_INVALID, "unknown dtype");
  if (!c->initial_params) return fail(nullptr, ME_ERR_INVALID, "initial_params is required");
  if (c->n_energy_coeffs < 0 || (c->n_energy_coeffs > 0 && !c->energy_coeffs))
    return fail(nullptr, ME_ERR_INVALID, "energy coefficients missing");
  if (c->reject_kind < ME_REJECT_NONE || c->reject_kind > ME_REJECT_USER)
    return fail(nullptr, ME_ERR_INVALID, "unknown reject_kind");
  if (c->reject_kind == ME_REJECT_USER && c->energy_kind != ME_ENERGY_USER && c->energy_kind != ME_ENERGY_USER_INDIRECT)
    return fail(nullptr, ME_ERR_INVALID, "ME_REJECT_USER needs a user-energy plugin (it supplies me_user_reject)");
  if (c->reject_kind == ME_REJECT_ABS_REAL0_GE && c->n_real == 0)
    return fail(nullptr, ME_ERR_INVALID, "ME_REJECT_ABS_REAL0_GE needs a real parameter");
  if (c->cov_mode < ME_COV_REFERENCE || c->cov_mode > ME_COV_POOLED)
    return fail(nullptr, ME_ERR_INVALID, "unknown cov_mode");
  const KernelSet *ks = find_kernel_set(c->dtype, c->n_real, c->n_complex, c->energy_kind, c->user_energy_name);
  if (!ks) {
    if (is_user_kind(c->energy_kind))
      return fail(nullptr, ME_ERR_UNSUPPORTED,
                  std::string("no user-energy plugin named '") + (c->user_energy_name ? c->user_energy_name : "") +
                      "' is loaded for these dimensions (build it with metropolisengine_amd.build.build_user_energy "
                      "and load it with me_load_plugin)");
    if (!has_dims(c->dtype, c->n_real, c->n_complex))
      return fail(nullptr, ME_ERR_UNSUPPORTED,
                  "no kernel set compiled for (dtype, n_real, n_complex) = (" + std::to_string(c->dtype) + ", " +
                      std::to_string(c->n_real) + ", " + std::to_string(c->n_complex) + ")");
    return fail(nullptr, ME_ERR_UNSUPPORTED, "energy kind " + std::to_string(c->energy_kind) +
                                                  " is not compiled for these dimensions");
  }
  // the reference evaluates reject_condition BEFORE the energy (metropolis_engine.py:247-249); a plugin without
  // me_user_reject would silently never reject and let chains walk into the forbidden region
  if (c->reject_kind == ME_REJECT_USER && !ks->has_user_reject)
    return fail(nullptr, ME_ERR_UNSUPPORTED,
                "ME_REJECT_USER: this user-energy plugin defines no me_user_reject (#define ME_USER_HAS_REJECT in its source)");
  if (ks->n_real < 0) {   // the runtime-dimension kernel set (me_runtime_dims.hip)
    if (c->cov_mode == ME_COV_POOLED && c->n_complex > 0)
      return fail(nullptr, ME_ERR_UNSUPPORTED,
                  "parameter spaces beyond " + std::to_string(kMaxRegisterDof) + " real degrees of freedom with complex "
                  "parameters run with the identity proposal shape (ME_COV_FIXED, cov_mode=\"fixed\") or per-chain shapes "
                  "(ME_COV_REFERENCE); one shared factor (ME_COV_POOLED) is there for pure real spaces");
    // an initial covariance that is not the identity makes the engine start with a SHARED factor (cov_kind below), which the
    // runtime set has for pure real spaces only and which doubles the LDS a block needs
    // (with ME_COV_REFERENCE the per-chain factor field exists and is filled with the initial factor: such an engine starts
    // with per-chain shapes right away, which every space has and which park x' only)
    const bool starts_shared = !initial_shape_is_identity(c) && c->cov_mode != ME_COV_REFERENCE;
    if (starts_shared && c->n_complex > 0)
      return fail(nullptr, ME_ERR_UNSUPPORTED,
                  "parameter spaces beyond " + std::to_string(kMaxRegisterDof) + " real degrees of freedom with complex "
                  "parameters run with the identity proposal shape only: leave covariance_matrix_real / _complex unset");
    {
      // a dense energy needs all of x' at once, a shared factor all of g: both are parked in LDS, 64 lanes x D values each
      const long long per_block = (long long)(c->n_real + 2 * c->n_complex) * 64 * (c->dtype == ME_F32 ? 4 : 8);
      const bool dense = c->energy_kind == ME_ENERGY_DENSE_QUAD, pooled = c->cov_mode == ME_COV_POOLED || starts_shared;
      // per-chain shapes (and the tracking flag): x' / the deviations from the running mean / a row of the factor, D values per lane
      const bool per_chain = c->cov_mode == ME_COV_REFERENCE || (c->flags & ME_FLAG_TRACK_COVARIANCE) != 0;
      if ((dense || pooled || per_chain) && per_block * (pooled ? 2 : 1) > 150 * 1024)
        return fail(nullptr, ME_ERR_UNSUPPORTED,
                    "a dense quadratic form, a shared proposal factor or per-chain shapes beyond " + std::to_string(kMaxRegisterDof) +
                        " degrees of freedom are staged in LDS: this many parameters do not fit (float64: about 290, 145 with "
                        "a shared factor or an initial covariance that ME_COV_FIXED keeps; float32 twice that)");
    }
    if (c->flags & ME_FLAG_REFERENCE_ENERGY_LEDGERS)
      return fail(nullptr, ME_ERR_UNSUPPORTED, "ME_FLAG_REFERENCE_ENERGY_LEDGERS is not available beyond " +
                                                   std::to_string(kMaxRegisterDof) + " real degrees of freedom");
    if (c->reject_kind == ME_REJECT_USER) return fail(nullptr, ME_ERR_UNSUPPORTED, "no user plugins at these dimensions");
  }
  if ((c->flags & ME_FLAG_REFERENCE_ENERGY_LEDGERS) && (c->n_real == 0 || c->n_complex == 0))
    return fail(nullptr, ME_ERR_INVALID,
                "ME_FLAG_REFERENCE_ENERGY_LEDGERS applies to mixed engines (real and complex parameters): only their "
                "step_all keeps a separate energy_total (metropolis_engine.py:241-259)");
  if (c->cov_mode == ME_COV_REFERENCE && !ks->per_chain_cov)
    return fail(nullptr, ME_ERR_UNSUPPORTED,
                "per-chain adaptive covariance is not compiled for these dimensions; use ME_COV_FIXED or ME_COV_POOLED");

  // The kernels address each field through a buffer descriptor with 32-bit offsets (me_device.h: Field).  The packed
  // per-chain covariance / factor fields are the widest: where THEY would pass 4 GiB and the proposal shape does not
  // need them (ME_COV_FIXED / ME_COV_POOLED) the engine simply keeps no per-chain covariance -- as for parameter
  // spaces without per-chain kernels, the ensemble covariance then comes from me_pooled_moments.
  // streamed sets (more than 160 packed entries) keep the per-chain matrices only when the proposals need them
  bool keep_per_chain = ks->per_chain_cov && (!ks->streams_packed || c->cov_mode == ME_COV_REFERENCE);
  {
    const long long esz = c->dtype == ME_F32 ? 4 : 8;
    const long long limit = 1ll << 32;
    if (keep_per_chain && !ks->streams_packed && (long long)packed_total(c->n_real, c->n_complex) * ((c->n_chains + 63) / 64 * 64) * esz >= limit) {
      if (c->cov_mode == ME_COV_REFERENCE)
        return fail(nullptr, ME_ERR_UNSUPPORTED,
                    "the per-chain covariance field would exceed 4 GiB on this engine; shard the chains over more "
                    "engines or use ME_COV_FIXED / ME_COV_POOLED");
      keep_per_chain = false;
    }
    const long long rows = std::max<long long>(c->n_real + 2 * c->n_complex, 2 * c->n_real + c->n_complex);
    if (rows * ((c->n_chains + 63) / 64 * 64) * esz >= limit)      // (the tile-major state is padded to whole 64-chain tiles)
      return fail(nullptr, ME_ERR_UNSUPPORTED,
                  "a per-chain field would exceed 4 GiB on this engine; shard the chains over more engines");
  }
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
    return fail(nullptr, ME_ERR_HIP, "no HIP device available (this library has no CPU fallback)");
  if (c->device_id < 0 || c->device_id >= n_dev) return fail(nullptr, ME_ERR_INVALID, "device_id out of range");

  me_engine *e = new me_engine;
  e->device = c->device_id;
  e->ks = ks;
  e->n_terms = ks->energy_terms(c->energy_kind);
  e->dtype = c->dtype;
  e->esize = c->dtype == ME_F32 ? 4 : 8;
  e->n = c->n_chains;
  e->nr = c->n_real;
  e->nc = c->n_complex;
  e->d = e->nr + 2 * e->nc;
  e->p = packed_total(e->nr, e->nc);
  e->nobs = n_observables(e->nr, e->nc);
  e->chain_offset = c->chain_offset;
  e->seed = c->seed;
  e->temp = c->temp;
  e->target_acceptance = c->target_acceptance;
  e->energy_kind = c->energy_kind;
  e->reject_kind = c->reject_kind;
  e->reject_bound = c->reject_bound;
  e->cov_mode = c->cov_mode;
  e->coef.assign(c->energy_coeffs, c->energy_coeffs + c->n_energy_coeffs);
  // adaptation constants (metropolis_engine.py:101-107; quirks Q2 and Q4 reproduced)
  e->alpha = -norm_ppf(c->target_acceptance / 2.0);
  e->m = e->nr + e->nc;
  e->ratio = (1.0 - 1.0 / e->m) * std::sqrt(2.0 * M_PI) * std::exp(e->alpha * e->alpha / 2.0) / 2.0 * e->alpha +
             1.0 / (e->m * c->target_acceptance * (1.0 - c->target_acceptance));
  if (const char *g = std::getenv("ME_GRID_BLOCKS")) e->grid_blocks = std::atoi(g);

#define ME_CREATE_HIP(call)                                                                  \
  do {                                                                                       \
    hipError_t err__ = (call);                                                               \
    if (err__ != hipSuccess) {                                                               \
      g_create_error = std::string(#call) + ": " + hipGetErrorString(err__);                 \
      release(e);                                                                            \
      return ME_ERR_HIP;                                                                     \
    }                                                                                        \
  } while (0)

  ME_CREATE_HIP(hipSetDevice(e->device));
  ME_CREATE_HIP(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
  e->own_stream = true;
  const size_t n = (size_t)e->n, es = e->esize;
  e->x_tiled = ks->tiled_state;
  ME_CREATE_HIP(hipMalloc(&e->x, (e->x_tiled ? (n + 63) / 64 * 64 : n) * e->d * es));   // tile-major: whole 64-chain tiles
  e->stale_total = (c->flags & ME_FLAG_REFERENCE_ENERGY_LEDGERS) != 0;
  ME_CREATE_HIP(hipMalloc(&e->energy, n * (e->n_terms + (e->stale_total ? 1 : 0)) * es));
  e->width_rows = (e->nr > 0 && e->nc > 0) ? 3 : 1;
  ME_CREATE_HIP(hipMalloc(&e->width, n * e->width_rows * es));
  ME_CREATE_HIP(hipMalloc(&e->mean, n * e->d * es));
  ME_CREATE_HIP(hipMalloc(&e->obs_mean, n * e->nobs * es));
  if (keep_per_chain || ((!ks->per_chain_cov || ks->streams_packed) && ks->tracks_cov && (c->flags & ME_FLAG_TRACK_COVARIANCE)))
    ME_CREATE_HIP(hipMalloc(&e->cov, (n + 63) / 64 * 64 * e->p * es));      // tile-major: whole 64-chain tiles
  if (keep_per_chain) ME_CREATE_HIP(hipMalloc(&e->factor, (n + 63) / 64 * 64 * e->p * es));
  ME_CREATE_HIP(hipMalloc(&e->shared_factor, (size_t)e->p * es));
  if (e->nc == 0) ME_CREATE_HIP(hipMalloc(&e->shared_full, (size_t)e->nr * e->nr * es));
  if (e->shared_full && ks->prepare_matrix) ME_CREATE_HIP(hipMalloc(&e->shared_image, ks->matrix_image_bytes));
  ME_CREATE_HIP(hipMalloc(&e->row_dev, (size_t)std::max(std::max(e->d, e->p), e->nobs) * es));
  // one slot per wavefront of the largest grid any step kernel uses: the float64 dense-64 kernel gives a wavefront a tile
  // of 32 chains (me_dense_f64.h), every other kernel 64
  e->n_slots = (e->n + 31) / 32 + 8;
  ME_CREATE_HIP(hipMalloc((void **)&e->accept_slots, (size_t)e->n_slots * sizeof(unsigned long long)));
  ME_CREATE_HIP(hipMalloc((void **)&e->accept_total, sizeof(unsigned long long)));
  ME_CREATE_HIP(hipMalloc((void **)&e->status, sizeof(unsigned int)));
  ME_CREATE_HIP(hipMalloc((void **)&e->pool_dev, sizeof(double) * (size_t)moments_size(e->nr, e->nc)));
  ME_CREATE_HIP(hipHostMalloc((void **)&e->host_scratch, 64, hipHostMallocDefault));
  e->host_scratch[0] = e->host_scratch[1] = 0;
  ME_CREATE_HIP(hipHostMalloc((void **)&e->pool_host, sizeof(double) * (size_t)moments_size(e->nr, e->nc), hipHostMallocDefault));
  if (pool_reduce_supported(e->nr, e->nc, e->dtype))     // (very large parameter spaces have no pooled-moment kernel)
    ME_CREATE_HIP(hipMalloc((void **)&e->pool_partials,
                            sizeof(double) * (size_t)pool_reduce_blocks(e->n, e->nr, e->nc) *
                                (size_t)(1 + e->d + e->nr + e->nc + e->d * (e->d + 1) / 2)));
  ME_CREATE_HIP(hipMemsetAsync(e->accept_slots, 0, (size_t)e->n_slots * sizeof(unsigned long long), e->stream));
  ME_CREATE_HIP(hipMemsetAsync(e->status, 0, sizeof(unsigned int), e->stream));
  if (ks->n_real < 0 && e->energy_kind == ME_ENERGY_DIAG_QUAD) {
    // the runtime-dimension kernels read one weight per real degree of freedom: a_i, then b_j for Re z_j and again for Im z_j
    if ((int)e->coef.size() != e->nr + e->nc) {
      g_create_error = "wrong number of energy coefficients for this energy kind";
      release(e);
      return ME_ERR_INVALID;
    }
    std::vector<double> expanded(e->coef.begin(), e->coef.begin() + e->nr);
    expanded.insert(expanded.end(), e->coef.begin() + e->nr, e->coef.end());
    expanded.insert(expanded.end(), e->coef.begin() + e->nr, e->coef.end());
    std::vector<unsigned char> bytes;
    to_device_type(expanded.data(), expanded.size(), e->dtype, bytes);
    ME_CREATE_HIP(hipMalloc(&e->coef_dev, bytes.size()));
    ME_CREATE_HIP(hipMemcpy(e->coef_dev, bytes.data(), bytes.size(), hipMemcpyHostToDevice));
  }
  if (ks->n_real < 0 && e->energy_kind == ME_ENERGY_DENSE_QUAD) {
    // the runtime-dimension kernels read the FOLDED lower triangle T_ij = A_ij + A_ji (i > j), T_ii = A_ii, row-major
    if ((int)e->coef.size() != e->d * e->d) {
      g_create_error = "wrong number of energy coefficients for this energy kind";
      release(e);
      return ME_ERR_INVALID;
    }
    std::vector<double> folded;
    folded.reserve((size_t)e->d * (e->d + 1) / 2);
    for (int i = 0; i < e->d; ++i)
      for (int j = 0; j <= i; ++j)
        folded.push_back(i == j ? e->coef[(size_t)i * e->d + i] : e->coef[(size_t)i * e->d + j] + e->coef[(size_t)j * e->d + i]);
    std::vector<unsigned char> bytes;
    to_device_type(folded.data(), folded.size(), e->dtype, bytes);
    ME_CREATE_HIP(hipMalloc(&e->coef_dev, bytes.size()));
    ME_CREATE_HIP(hipMemcpy(e->coef_dev, bytes.data(), bytes.size(), hipMemcpyHostToDevice));
  } else
  if (e->energy_kind == ME_ENERGY_DENSE_QUAD || (is_user_kind(e->energy_kind) && !e->coef.empty())) {
    std::vector<unsigned char> bytes;
    to_device_type(e->coef.data(), e->coef.size(), e->dtype, bytes);
    ME_CREATE_HIP(hipMalloc(&e->coef_dev, bytes.size()));
    ME_CREATE_HIP(hipMemcpy(e->coef_dev, bytes.data(), bytes.size(), hipMemcpyHostToDevice));
    if (e->energy_kind == ME_ENERGY_DENSE_QUAD && ks->prepare_matrix) {
      ME_CREATE_HIP(hipMalloc(&e->energy_image, ks->matrix_image_bytes));
      ME_CREATE_HIP(ks->prepare_matrix(e->coef_dev, e->energy_image, e->stream));
    }
  }
#undef ME_CREATE_HIP

  // initial state: every chain starts at the caller's point (:41, :51); running means start there too (:77-78)
  std::vector<double> init(c->initial_params, c->initial_params + e->d);
  std::vector<double> obs(e->nobs);
  for (int i = 0; i < e->nr; ++i) {
    obs[i] = std::fabs(init[i]);
    obs[e->nr + e->nc + i] = init[i] * init[i];
  }
  for (int j = 0; j < e->nc; ++j) obs[e->nr + j] = std::hypot(init[e->nr + j], init[e->nr + e->nc + j]);
  // initial proposal matrices, packed (identity by default, :63-70)
  const int pr = packed_real(e->nr);
  std::vector<double> c0(e->p, 0.0);
  bool identity = true;
  for (int i = 0; i < e->nr; ++i)
    for (int j = 0; j <= i; ++j) {
      const double v = c->covariance_real ? c->covariance_real[i * e->nr + j] : (i == j ? 1.0 : 0.0);
      c0[i * (i + 1) / 2 + j] = v;
      identity = identity && v == (i == j ? 1.0 : 0.0);
    }
  for (int i = 0; i < e->nc; ++i) {
    for (int j = 0; j < i; ++j) {
      const double re = c->covariance_complex ? c->covariance_complex[2 * (i * e->nc + j)] : 0.0;
      const double im = c->covariance_complex ? c->covariance_complex[2 * (i * e->nc + j) + 1] : 0.0;
      c0[pr + i * i + 2 * j] = re;
      c0[pr + i * i + 2 * j + 1] = im;
      identity = identity && re == 0.0 && im == 0.0;
    }
    const double dg = c->covariance_complex ? c->covariance_complex[2 * (i * e->nc + i)] : 1.0;
    c0[pr + i * i + 2 * i] = dg;
    identity = identity && dg == 1.0;
  }
  std::vector<double> f0 = c0;
  for (int i = 0; i < e->nc; ++i)
    for (int j = 0; j < i; ++j) f0[pr + i * i + 2 * j + 1] = -f0[pr + i * i + 2 * j + 1];   // conj(K), quirk Q3
  if (!host_factor(f0, e->nr, e->nc)) {
    release(e);
    return fail(nullptr, ME_ERR_INVALID, "initial covariance matrix is not positive definite (metropolis_engine.py:270)");
  }
  int rc = ME_OK;
  std::vector<double> width_row((size_t)e->width_rows, c->sampling_width);
  if ((rc = broadcast(e, e->x, init, e->x_tiled)) || (rc = broadcast(e, e->mean, init)) || (rc = broadcast(e, e->obs_mean, obs)) ||
      (rc = broadcast(e, e->width, width_row)) || (e->cov && (rc = broadcast(e, e->cov, c0, true))) ||
      (e->factor && (rc = broadcast(e, e->factor, f0, true)))) {
    g_create_error = e->err;
    release(e);
    return rc;
  }
  if (upload_shared_factor(e, f0.data()) != ME_OK) {
    g_create_error = e->err;
    release(e);
    return ME_ERR_HIP;
  }
  // a non-identity initial shape is ONE factor for all chains; the runtime-dimension set reads it from the per-chain field
  // (every chain's copy was written above) when the engine keeps one -- its shared-factor form is for pure real spaces
  // only and needs twice the LDS
  e->cov_kind = identity ? CK_IDENTITY : (ks->n_real < 0 && e->factor) ? CK_PER_CHAIN : CK_SHARED;

  rc = me_recompute_energy(e);
  if (rc == ME_OK) {
    hipError_t herr = hipStreamSynchronize(e->stream);
    if (herr != hipSuccess) rc = fail(e, ME_ERR_HIP, std::string("initial energy evaluation: ") + hipGetErrorString(herr));
  }
  if (rc != ME_OK) {
    g_create_error = e->err;
    release(e);
    return rc;
  }
  *out = e;
  return ME_OK;
}

int me_set_reject_condition(me_engine *e, int32_t reject_kind, double reject_bound) {
  if (!e) return ME_ERR_INVALID;
  if (reject_kind < ME_REJECT_NONE || reject_kind > ME_REJECT_USER) return fail(e, ME_ERR_INVALID, "unknown reject_kind");
  if (reject_kind == ME_REJECT_USER && e->energy_kind != ME_ENERGY_USER && e->energy_kind != ME_ENERGY_USER_INDIRECT)
    return fail(e, ME_ERR_INVALID, "ME_REJECT_USER needs a user-energy plugin (it supplies me_user_reject)");
  if (reject_kind == ME_REJECT_USER && !e->ks->has_user_reject)
    return fail(e, ME_ERR_UNSUPPORTED,
                "ME_REJECT_USER: this user-energy plugin defines no me_user_reject (#define ME_USER_HAS_REJECT in its source)");
  if (reject_kind == ME_REJECT_ABS_REAL0_GE && e->nr == 0)
    return fail(e, ME_ERR_INVALID, "ME_REJECT_ABS_REAL0_GE needs a real parameter");
  e->reject_kind = reject_kind;     // a launch parameter: takes effect from the next step
  e->reject_bound = reject_bound;
  return ME_OK;
}

int me_set_energy(me_engine *e, int32_t energy_kind, const double *coeffs, int32_t n_coeffs, const char *user_name) {
  if (!e) return ME_ERR_INVALID;
  if (n_coeffs < 0 || (n_coeffs > 0 && !coeffs)) return fail(e, ME_ERR_INVALID, "energy coefficients missing");
  if (e->ks->n_real < 0) return fail(e, ME_ERR_UNSUPPORTED, "me_set_energy: not available on the runtime-dimension kernel set");
  if (energy_kind == ME_ENERGY_DENSE_QUAD)
    return fail(e, ME_ERR_UNSUPPORTED, "me_set_energy: the dense quadratic form keeps device images of its matrix; create a new engine");
  const KernelSet *ks = find_kernel_set(e->dtype, e->nr, e->nc, energy_kind, user_name);
  if (!ks || ks->n_real < 0)
    return fail(e, ME_ERR_UNSUPPORTED, is_user_kind(energy_kind)
                    ? std::string("no user-energy plugin named '") + (user_name ? user_name : "") + "' is loaded for these dimensions"
                    : "energy kind " + std::to_string(energy_kind) + " is not compiled for these dimensions");
  // the engine's fields were laid out for the kernel set it was created with: the new set must use them the same way
  if (ks->tiled_state != e->ks->tiled_state || ks->per_chain_cov != e->ks->per_chain_cov || ks->streams_packed != e->ks->streams_packed ||
      ks->tracks_cov != e->ks->tracks_cov)
    return fail(e, ME_ERR_UNSUPPORTED, "me_set_energy: the plugin was built with other per-chain covariance options than this engine's kernel set");
  if (e->reject_kind == ME_REJECT_USER && !ks->has_user_reject)
    return fail(e, ME_ERR_UNSUPPORTED, "me_set_energy: the engine's wall is the plugin's me_user_reject, which the new energy does not define");
  const int n_terms = ks->energy_terms(energy_kind);
  ME_HIP(e, hipSetDevice(e->device));
  ME_HIP(e, hipStreamSynchronize(e->stream));
  if (n_terms != e->n_terms) {      // another number of ledger rows (metropolis_engine.py:134-138: the term names are collected anew)
    void *ledger = nullptr;
    ME_HIP(e, hipMalloc(&ledger, (size_t)e->n * (size_t)(n_terms + (e->stale_total ? 1 : 0)) * e->esize));
    (void)hipFree(e->energy);
    e->energy = ledger;
    e->n_terms = n_terms;
    if (e->trace_chains > 0) {       // the recorded series has one column per term: it starts over
      if (e->trace_dev) (void)hipFree(e->trace_dev);
      e->trace_dev = nullptr;
      e->trace_rows = e->trace_capacity = 0;
    }
  }
  e->coef.assign(coeffs, coeffs + n_coeffs);
  if (e->coef_dev) {
    (void)hipFree(e->coef_dev);
    e->coef_dev = nullptr;
  }
  if (is_user_kind(energy_kind) && !e->coef.empty()) {
    std::vector<unsigned char> bytes;
    to_device_type(e->coef.data(), e->coef.size(), e->dtype, bytes);
    ME_HIP(e, hipMalloc(&e->coef_dev, bytes.size()));
    ME_HIP(e, hipMemcpy(e->coef_dev, bytes.data(), bytes.size(), hipMemcpyHostToDevice));
  }
  e->ks = ks;
  e->energy_kind = energy_kind;
  // every term of the ledger at the current state (initialize_energy_dict, :152-155)
  const int rc = me_recompute_energy(e);
  if (rc != ME_OK) return rc;
  ME_HIP(e, hipStreamSynchronize(e->stream));
  return check_status(e);
}

int me_load_plugin(const char *path) {
  if (!path) return fail(nullptr, ME_ERR_INVALID, "null plugin path");
  // the plugin's static initialiser registers its kernel sets (me::register_kernel_set)
  void *handle = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
  if (!handle) return fail(nullptr, ME_ERR_INVALID, std::string("dlopen failed: ") + dlerror());
  return ME_OK;
}

int me_destroy(me_engine *e) {
  if (!e) return ME_ERR_INVALID;
  (void)hipSetDevice(e->device);
  if (e->stream) (void)hipStreamSynchronize(e->stream);
  if (e->copy_stream) (void)hipStreamSynchronize(e->copy_stream);   // an all-reduce / copy of a reduction nobody collected
  release(e);
  return ME_OK;
}

int me_recompute_energy(me_engine *e) {
  if (!e) return ME_ERR_INVALID;
  ME_HIP(e, hipSetDevice(e->device));
  EnergyLaunch l;
  l.n_real = e->nr;
  l.n_complex = e->nc;
  l.total_row = e->stale_total ? e->n_terms : -1;
  l.x = e->x;
  l.energy = e->energy;
  l.coef_device = e->coef_dev;
  l.coef_host = e->coef.data();
  l.n_coef = (int)e->coef.size();
  l.status = e->status;
  l.n = e->n;
  l.energy_kind = e->energy_kind;
  l.grid_blocks = e->grid_blocks;
  hipError_t err = e->ks->init_energy(l, e->stream);
  if (err == hipErrorInvalidValue) return fail(e, ME_ERR_INVALID, "wrong number of energy coefficients for this energy kind");
  ME_HIP(e, err);
  return ME_OK;
}

// Mixed engines keep three widths per chain [sampling_width, real group, complex group].  After a step_all the
// group rows are implied equal to row 0 (:436-437) and are not written; before the first group-wise launch they
// are materialised here.
static int split_widths(me_engine *e) {
  if (e->width_rows != 3 || !e->widths_synced) return ME_OK;
  const size_t row = (size_t)e->n * e->esize;
  for (int r = 1; r <= 2; ++r)
    ME_HIP(e, hipMemcpyAsync((unsigned char *)e->width + r * row, e->width, row, hipMemcpyDeviceToDevice, e->stream));
  e->widths_synced = false;
  return ME_OK;
}

// kind: 0 = step_all, 1 = real group, 2 = complex group (Gaussian), 3 = magnitude-phase pair on the complex group
static int resolve_step_kind(me_engine *e, int kind, int *group, bool *magphase) {
  *magphase = false;
  *group = GROUP_ALL;
  const bool mixed = e->nr > 0 && e->nc > 0;
  switch (kind) {
    case ME_STEP_ALL: return ME_OK;
    case ME_STEP_REAL_GROUP:
      if (e->nr == 0) return fail(e, ME_ERR_INVALID, "this engine has no real parameters");
      *group = mixed ? GROUP_REAL : GROUP_ALL;     // pure-real engines: step_all IS step_real_group (:56)
      return ME_OK;
    case ME_STEP_COMPLEX_GROUP:
      if (e->nc == 0) return fail(e, ME_ERR_INVALID, "this engine has no complex parameters");
      *group = mixed ? GROUP_COMPLEX : GROUP_ALL;  // :46
      return ME_OK;
    case ME_STEP_COMPLEX_MAGNITUDE_PHASE:
      if (e->nc == 0) return fail(e, ME_ERR_INVALID, "this engine has no complex parameters");
      if (!e->ks->magphase || !e->cov)
        return fail(e, ME_ERR_UNSUPPORTED, "the magnitude-phase sampler needs the per-chain covariance kernels");
      *magphase = true;
      return ME_OK;
    default: return fail(e, ME_ERR_INVALID, "unknown step kind");
  }
}

static int launch_step_kind(me_engine *e, int kind, int n_sweeps, const void *inj_normals, const void *inj_uniforms) {
  int group;
  bool magphase;
  int rc = resolve_step_kind(e, kind, &group, &magphase);
  if (rc != ME_OK) return rc;
  if ((group != GROUP_ALL || magphase) && (rc = split_widths(e)) != ME_OK) return rc;
  StepLaunch l;
  fill_step_launch(e, l, n_sweeps);
  l.group = group;
  l.inj_normals = inj_normals;
  l.inj_uniforms = inj_uniforms;
  hipError_t err = magphase ? e->ks->magphase(l, e->stream) : e->ks->step(l, e->stream);
  if (err == hipErrorNotSupported) return fail(e, ME_ERR_UNSUPPORTED, "this step kind is not compiled for this engine");
  ME_HIP(e, err);
  if (group == GROUP_ALL && !magphase) e->widths_synced = true;   // step_all mirrors row 0 into the group widths
  e->step_index += (unsigned long long)n_sweeps;
  e->proposed += (unsigned long long)e->n * (unsigned long long)n_sweeps * (magphase ? 2ull : 1ull);
  return ME_OK;
}

int me_step(me_engine *e, int32_t n_sweeps) { return me_step_kind(e, ME_STEP_ALL, n_sweeps); }

int me_step_kind(me_engine *e, int32_t kind, int32_t n_sweeps) {
  if (!e) return ME_ERR_INVALID;
  if (n_sweeps <= 0) return fail(e, ME_ERR_INVALID, "n_sweeps must be positive");
  ME_HIP(e, hipSetDevice(e->device));
  return launch_step_kind(e, kind, n_sweeps, nullptr, nullptr);
}

int me_step_injected(me_engine *e, int32_t kind, int32_t n_sweeps, const double *normals, const double *uniforms) {
  if (!e || !normals || !uniforms) return ME_ERR_INVALID;
  if (n_sweeps <= 0) return fail(e, ME_ERR_INVALID, "n_sweeps must be positive");
  if (e->dtype != ME_F64) return fail(e, ME_ERR_UNSUPPORTED, "injected-stream replay is compiled for float64 engines only");
  ME_HIP(e, hipSetDevice(e->device));
  // per chain and sweep: Gaussian kinds take D normals + 1 uniform; the magnitude-phase pair nc normals + nc+2 uniforms
  const bool magphase = kind == ME_STEP_COMPLEX_MAGNITUDE_PHASE;
  const size_t n = (size_t)e->n, k = (size_t)n_sweeps;
  const size_t nz = magphase ? (size_t)e->nc : (size_t)e->d, nu = magphase ? (size_t)e->nc + 2 : 1;
  std::vector<double> zn(k * nz * n), un(k * nu * n);
  for (size_t s = 0; s < k; ++s)
    for (size_t c = 0; c < n; ++c) {
      for (size_t j = 0; j < nz; ++j) zn[(s * nz + j) * n + c] = normals[(s * n + c) * nz + j];
      for (size_t j = 0; j < nu; ++j) un[(s * nu + j) * n + c] = uniforms[(s * n + c) * nu + j];
    }
  double *zd = nullptr, *ud = nullptr;
  ME_HIP(e, hipMalloc((void **)&zd, zn.size() * sizeof(double)));
  hipError_t herr = hipMalloc((void **)&ud, un.size() * sizeof(double));
  if (herr != hipSuccess) {
    (void)hipFree(zd);
    return fail(e, ME_ERR_HIP, "hipMalloc of the injected uniforms failed");
  }
  int rc = ME_OK;
  if (hipMemcpy(zd, zn.data(), zn.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(ud, un.data(), un.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
    rc = fail(e, ME_ERR_HIP, "upload of the injected streams failed");
  } else {
    rc = launch_step_kind(e, kind, n_sweeps, zd, ud);
    if (rc == ME_OK && hipStreamSynchronize(e->stream) != hipSuccess) rc = fail(e, ME_ERR_HIP, "injected step failed");
  }
  (void)hipFree(zd);
  (void)hipFree(ud);
  return rc;
}

namespace {
// measure() as a launch descriptor: `count` is the counter AFTER the increment (metropolis_engine.py:343)
void fill_measure_launch(me_engine *e, MeasureLaunch &l, unsigned long long count) {
  l.n_real = e->nr;
  l.n_complex = e->nc;
  l.x = e->x;
  l.width = e->width;
  l.mean = e->mean;
  l.cov = e->cov;
  l.obs_mean = e->obs_mean;
  l.factor = e->factor;
  l.status = e->status;
  l.n = e->n;
  l.measure_count = count;
  l.update_cov = (count > 50 && e->cov) ? 1 : 0;   // :389, :396
  l.split_widths = (e->width_rows == 3 && !e->widths_synced) ? 1 : 0;
  l.write_factor = (l.update_cov && e->cov_mode == ME_COV_REFERENCE) ? 1 : 0;
  l.grid_blocks = e->grid_blocks;
}

// host bookkeeping once the measure (or cycle) launch succeeded: the counter, the proposal shape the NEXT steps use, and the
// time-series row of the traced chains (:350-356)
int commit_measure(me_engine *e, const MeasureLaunch &l) {
  e->measure_count = l.measure_count;
  if (l.write_factor) e->cov_kind = CK_PER_CHAIN;
  if (e->trace_chains > 0) {
    const long long cols = e->d + e->n_terms + e->width_rows;
    if (e->trace_rows == e->trace_capacity) {   // grow the device-side series (doubling)
      const long long cap = e->trace_capacity ? 2 * e->trace_capacity : 1024;
      double *bigger = nullptr;
      ME_HIP(e, hipMalloc((void **)&bigger, sizeof(double) * (size_t)(cap * cols * e->trace_chains)));
      if (e->trace_dev) {
        ME_HIP(e, hipMemcpyAsync(bigger, e->trace_dev, sizeof(double) * (size_t)(e->trace_rows * cols * e->trace_chains),
                                 hipMemcpyDeviceToDevice, e->stream));
        ME_HIP(e, hipStreamSynchronize(e->stream));
        (void)hipFree(e->trace_dev);
      }
      e->trace_dev = bigger;
      e->trace_capacity = cap;
    }
    // widths: a synced mixed engine keeps only row 0 current; mirror it so that the series reads like the reference's
    const int rows_valid = (e->width_rows == 3 && e->widths_synced) ? 1 : e->width_rows;
    ME_HIP(e, launch_trace(e->x, e->energy, e->width, e->n, e->d, e->n_terms, rows_valid, e->dtype, e->trace_chains, e->trace_stride,
                           e->trace_dev + e->trace_rows * cols * e->trace_chains, e->stream, e->x_tiled));
    if (rows_valid != e->width_rows) {
      double *row = e->trace_dev + e->trace_rows * cols * e->trace_chains + (size_t)(e->d + e->n_terms) * e->trace_chains;
      for (int r = 1; r < 3; ++r)
        ME_HIP(e, hipMemcpyAsync(row + (size_t)r * e->trace_chains, row, sizeof(double) * (size_t)e->trace_chains,
                                 hipMemcpyDeviceToDevice, e->stream));
    }
    e->trace_rows += 1;
  }
  return ME_OK;
}
}  // namespace

int me_measure(me_engine *e) {
  if (!e) return ME_ERR_INVALID;
  ME_HIP(e, hipSetDevice(e->device));
  MeasureLaunch l;
  fill_measure_launch(e, l, e->measure_count + 1);   // committed once the launch succeeded
  ME_HIP(e, e->ks->measure(l, e->stream));
  return commit_measure(e, l);
}

int me_cycle(me_engine *e, int32_t n_sweeps) {
  if (!e) return ME_ERR_INVALID;
  if (n_sweeps <= 0) return fail(e, ME_ERR_INVALID, "n_sweeps must be positive");
  ME_HIP(e, hipSetDevice(e->device));
  if (e->ks->cycle) {
    StepLaunch sl;
    fill_step_launch(e, sl, n_sweeps);
    MeasureLaunch ml;
    fill_measure_launch(e, ml, e->measure_count + 1);
    ml.split_widths = 0;                 // the sweeps are step_all: afterwards the group widths equal the shared one
    const hipError_t err = e->ks->cycle(sl, ml, e->stream);
    if (err == hipSuccess) {
      e->widths_synced = true;
      e->step_index += (unsigned long long)n_sweeps;
      e->proposed += (unsigned long long)e->n * (unsigned long long)n_sweeps;
      e->fused_cycles += 1;
      return commit_measure(e, ml);
    }
    if (err != hipErrorNotSupported) ME_HIP(e, err);
    (void)hipGetLastError();
  }
  // no fused kernel for this engine / state: the same two launches the caller would have issued
  const int rc = me_step(e, n_sweeps);
  return rc != ME_OK ? rc : me_measure(e);
}

int me_cycle_stats(me_engine *e, uint64_t *fused_cycles) {
  if (!e || !fused_cycles) return ME_ERR_INVALID;
  *fused_cycles = e->fused_cycles;
  return ME_OK;
}

int me_trace_enable(me_engine *e, int64_t n_traced, int64_t stride) {
  if (!e) return ME_ERR_INVALID;
  if (n_traced < 0 || stride < 1 || (n_traced > 0 && (n_traced - 1) * stride >= e->n))
    return fail(e, ME_ERR_INVALID, "traced chains out of range");
  ME_HIP(e, hipSetDevice(e->device));
  ME_HIP(e, hipStreamSynchronize(e->stream));
  if (e->trace_dev) (void)hipFree(e->trace_dev);
  e->trace_dev = nullptr;
  e->trace_chains = n_traced;
  e->trace_stride = stride;
  e->trace_rows = e->trace_capacity = 0;
  return ME_OK;
}

int me_trace_shape(me_engine *e, int64_t *rows, int64_t *cols, int64_t *n_traced) {
  if (!e) return ME_ERR_INVALID;
  if (rows) *rows = e->trace_rows;
  if (cols) *cols = e->d + e->n_terms + e->width_rows;
  if (n_traced) *n_traced = e->trace_chains;
  return ME_OK;
}

int me_trace_get(me_engine *e, double *dst, int64_t n_doubles) {
  if (!e || !dst) return ME_ERR_INVALID;
  const long long want = e->trace_rows * (e->d + e->n_terms + e->width_rows) * e->trace_chains;
  if (n_doubles != want) return fail(e, ME_ERR_INVALID, "wrong trace buffer length");
  if (want == 0) return ME_OK;
  ME_HIP(e, hipSetDevice(e->device));
  ME_HIP(e, hipStreamSynchronize(e->stream));
  ME_HIP(e, hipMemcpy(dst, e->trace_dev, sizeof(double) * (size_t)want, hipMemcpyDeviceToHost));
  return ME_OK;
}

int me_field_components(me_engine *e, int32_t field, int32_t *n_components) {
  if (!e || !n_components) return ME_ERR_INVALID;
  void *ptr;
  int comps;
  int rc = field_info(e, field, &ptr, &comps);
  if (rc == ME_OK) *n_components = comps;
  return rc;
}

int me_energy_terms(me_engine *e, int32_t *n_terms) {
  if (!e || !n_terms) return ME_ERR_INVALID;
  *n_terms = e->n_terms;
  return ME_OK;
}

int me_get(me_engine *e, int32_t field, int64_t chain_begin, int64_t n_chains, double *dst) {
  if (!e || !dst) return ME_ERR_INVALID;
  if (chain_begin < 0 || n_chains < 0 || chain_begin + n_chains > e->n) return fail(e, ME_ERR_INVALID, "chain range out of bounds");
  ME_HIP(e, hipSetDevice(e->device));
  void *ptr;
  int comps;
  int rc = field_info(e, field, &ptr, &comps);
  if (rc != ME_OK) return rc;
  if (n_chains == 0) return ME_OK;
  if (field == ME_FIELD_COV || field == ME_FIELD_FACTOR || (field == ME_FIELD_PARAMS && e->x_tiled))
    return copy_tiled(e, ptr, comps, chain_begin, n_chains, dst, nullptr);
  std::vector<unsigned char> tmp((size_t)comps * n_chains * e->esize);
  for (int r = 0; r < comps; ++r)
    ME_HIP(e, hipMemcpyAsync(tmp.data() + (size_t)r * n_chains * e->esize,
                             (const unsigned char *)ptr + ((size_t)r * e->n + chain_begin) * e->esize,
                             (size_t)n_chains * e->esize, hipMemcpyDeviceToHost, e->stream));
  ME_HIP(e, hipStreamSynchronize(e->stream));
  if (e->dtype == ME_F32) {
    const float *t = reinterpret_cast<const float *>(tmp.data());
    for (int r = 0; r < comps; ++r)
      for (int64_t c = 0; c < n_chains; ++c) dst[c * comps + r] = (double)t[(size_t)r * n_chains + c];
  } else {
    const double *t = reinterpret_cast<const double *>(tmp.data());
    for (int r = 0; r < comps; ++r)
      for (int64_t c = 0; c < n_chains; ++c) dst[c * comps + r] = t[(size_t)r * n_chains + c];
  }
  if (field == ME_FIELD_WIDTH && e->width_rows == 3 && e->widths_synced)
    for (int64_t c = 0; c < n_chains; ++c) dst[c * 3 + 1] = dst[c * 3 + 2] = dst[c * 3];
  return check_status(e);
}

int me_set(me_engine *e, int32_t field, int64_t chain_begin, int64_t n_chains, const double *src) {
  if (!e || !src) return ME_ERR_INVALID;
  if (chain_begin < 0 || n_chains < 0 || chain_begin + n_chains > e->n) return fail(e, ME_ERR_INVALID, "chain range out of bounds");
  ME_HIP(e, hipSetDevice(e->device));
  void *ptr;
  int comps;
  int rc = field_info(e, field, &ptr, &comps);
  if (rc != ME_OK) return rc;
  // every argument check comes before the first device write
  if (field == ME_FIELD_WIDTH && e->width_rows == 3 && (chain_begin != 0 || n_chains != e->n))
    return fail(e, ME_ERR_INVALID, "widths of a mixed engine must be set for all chains at once");
  if (n_chains == 0) return ME_OK;
  if (field == ME_FIELD_COV || field == ME_FIELD_FACTOR || (field == ME_FIELD_PARAMS && e->x_tiled)) {
    rc = copy_tiled(e, ptr, comps, chain_begin, n_chains, nullptr, src);
    if (rc != ME_OK) return rc;
    if (field == ME_FIELD_FACTOR) e->cov_kind = CK_PER_CHAIN;
    return ME_OK;
  }
  std::vector<unsigned char> tmp((size_t)comps * n_chains * e->esize);
  if (e->dtype == ME_F32) {
    float *t = reinterpret_cast<float *>(tmp.data());
    for (int r = 0; r < comps; ++r)
      for (int64_t c = 0; c < n_chains; ++c) t[(size_t)r * n_chains + c] = (float)src[c * comps + r];
  } else {
    double *t = reinterpret_cast<double *>(tmp.data());
    for (int r = 0; r < comps; ++r)
      for (int64_t c = 0; c < n_chains; ++c) t[(size_t)r * n_chains + c] = src[c * comps + r];
  }
  for (int r = 0; r < comps; ++r)
    ME_HIP(e, hipMemcpyAsync((unsigned char *)ptr + ((size_t)r * e->n + chain_begin) * e->esize,
                             tmp.data() + (size_t)r * n_chains * e->esize, (size_t)n_chains * e->esize,
                             hipMemcpyHostToDevice, e->stream));
  ME_HIP(e, hipStreamSynchronize(e->stream));
  if (field == ME_FIELD_FACTOR) e->cov_kind = CK_PER_CHAIN;
  if (field == ME_FIELD_WIDTH && e->width_rows == 3) e->widths_synced = false;
  return ME_OK;
}

int me_constants(me_engine *e, double *alpha, int32_t *m, double *ratio) {
  if (!e) return ME_ERR_INVALID;
  if (alpha) *alpha = e->alpha;
  if (m) *m = e->m;
  if (ratio) *ratio = e->ratio;
  return ME_OK;
}

int me_counters(me_engine *e, uint64_t *step_index, uint64_t *measure_step_counter) {
  if (!e) return ME_ERR_INVALID;
  if (step_index) *step_index = e->step_index;
  if (measure_step_counter) *measure_step_counter = e->measure_count;
  return ME_OK;
}

int me_set_counters(me_engine *e, uint64_t step_index, uint64_t measure_step_counter) {
  if (!e) return ME_ERR_INVALID;
  if (measure_step_counter < 1) return fail(e, ME_ERR_INVALID, "measure_step_counter starts at 1");
  e->step_index = step_index;
  e->measure_count = measure_step_counter;
  return ME_OK;
}

int me_accept_stats(me_engine *e, uint64_t *accepted, uint64_t *proposed) {
  if (!e) return ME_ERR_INVALID;
  ME_HIP(e, hipSetDevice(e->device));
  ME_HIP(e, launch_sum_slots(e->accept_slots, e->n_slots, e->accept_total, e->stream));
  ME_HIP(e, hipMemcpyAsync(&e->host_scratch[1], e->accept_total, sizeof(unsigned long long), hipMemcpyDeviceToHost, e->stream));
  ME_HIP(e, hipMemcpyAsync(&e->host_scratch[0], e->status, sizeof(unsigned int), hipMemcpyDeviceToHost, e->stream));
  ME_HIP(e, hipStreamSynchronize(e->stream));      // one wait for both words
  if (accepted) *accepted = e->host_scratch[1];
  if (proposed) *proposed = e->proposed;
  return report_status(e, (unsigned int)e->host_scratch[0]);
}

int me_set_accept_stats(me_engine *e, uint64_t accepted, uint64_t proposed) {
  if (!e) return ME_ERR_INVALID;
  if (accepted > proposed) return fail(e, ME_ERR_INVALID, "accepted exceeds proposed");
  ME_HIP(e, hipSetDevice(e->device));
  // the device keeps one counter per wavefront, summed on demand: all of them to zero, the total into the first
  const unsigned long long total = accepted;
  ME_HIP(e, hipMemsetAsync(e->accept_slots, 0, (size_t)e->n_slots * sizeof(unsigned long long), e->stream));
  ME_HIP(e, hipMemcpyAsync(e->accept_slots, &total, sizeof(total), hipMemcpyHostToDevice, e->stream));
  ME_HIP(e, hipStreamSynchronize(e->stream));
  e->proposed = proposed;
  return ME_OK;
}

int me_pooled_moments_size(me_engine *e, int64_t *n_doubles) {
  if (!e || !n_doubles) return ME_ERR_INVALID;
  *n_doubles = moments_size(e->nr, e->nc);
  return ME_OK;
}

namespace {
// k_pool_reduce + k_pool_finish into `device_out`, enqueued on the engine's stream (no synchronisation)
int enqueue_pooled_moments(me_engine *e, void *device_out, int64_t n_doubles) {
  if (n_doubles != moments_size(e->nr, e->nc)) return fail(e, ME_ERR_INVALID, "wrong pooled-moment buffer length");
  if (e->pool_pending) return fail(e, ME_ERR_STATE, "a pooled-moment reduction is in flight (me_pooled_moments_end first)");
  if (!e->pool_partials) return fail(e, ME_ERR_UNSUPPORTED, "pooled moments: dimension too large for the reduction kernel");
  ME_HIP(e, hipSetDevice(e->device));
  hipError_t err = launch_pool_reduce(e->x, e->n, e->nr, e->nc, e->dtype, e->accept_slots, e->n_slots,
                                      (double)e->proposed, e->pool_partials, (double *)device_out, e->stream,
                                      e->ks->pool_stage1, e->x_tiled);
  if (err == hipErrorInvalidValue) return fail(e, ME_ERR_UNSUPPORTED, "pooled moments: dimension too large for the reduction kernel");
  ME_HIP(e, err);
  return ME_OK;
}
}  // namespace

// Waits for `ev`, polling for the first two milliseconds.  hipEventSynchronize gives up its own active wait after a few
// microseconds and blocks; the thread is then woken by an interrupt, and on a host that is otherwise idle (one OpenMP
// thread, as torch.distributed.run sets it) the core has gone to sleep by then: the overlapped config 5 loop, whose host side
// is 85 us per cycle, ran 3-5 x slower there.
static hipError_t wait_polling(hipEvent_t ev) {
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    const hipError_t rc = hipEventQuery(ev);
    if (rc != hipErrorNotReady) return rc;
    if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) return hipEventSynchronize(ev);
  }
}
// the same for everything queued on a stream
static hipError_t wait_polling(hipStream_t stream) {
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    const hipError_t rc = hipStreamQuery(stream);
    if (rc != hipErrorNotReady) return rc;
    if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) return hipStreamSynchronize(stream);
  }
}

int me_pooled_moments_device(me_engine *e, void *device_out, int64_t n_doubles) {
  if (!e || !device_out) return ME_ERR_INVALID;
  int rc = enqueue_pooled_moments(e, device_out, n_doubles);
  if (rc != ME_OK) return rc;
  ME_HIP(e, wait_polling(e->stream));
  return ME_OK;
}

int me_pooled_moments(me_engine *e, double *host_out, int64_t n_doubles) {
  if (!e || !host_out) return ME_ERR_INVALID;
  int rc = enqueue_pooled_moments(e, e->pool_dev, n_doubles);
  if (rc != ME_OK) return rc;
  // through a pinned staging buffer: one asynchronous copy behind the kernels, one wait
  ME_HIP(e, hipMemcpyAsync(e->pool_host, e->pool_dev, sizeof(double) * (size_t)n_doubles, hipMemcpyDeviceToHost, e->stream));
  ME_HIP(e, wait_polling(e->stream));
  std::memcpy(host_out, e->pool_host, sizeof(double) * (size_t)n_doubles);
  return ME_OK;
}

namespace {
// The split reduction: the two reduction kernels behind the work already queued on the engine's stream; then, on the
// engine's SECOND stream behind an event, (all-reduce over the ranks of the engine's communicator,) the copy into the
// pinned staging buffer and the event me_pooled_moments_end waits for.  Nothing here waits on the host, and the engine's
// main stream never waits for another rank: steps enqueued afterwards run beside the collective.
int begin_pooled(me_engine *e, bool allreduce) {
  if (e->pool_pending) return fail(e, ME_ERR_STATE, "a pooled-moment reduction is already in flight: call me_pooled_moments_end first");
  if (allreduce && !e->comm)
    return fail(e, ME_ERR_STATE, "this engine has no communicator: call me_comm_init_rank first (single-GPU engines use me_pooled_moments)");
  ME_HIP(e, hipSetDevice(e->device));
  if (!e->copy_stream) {
    ME_HIP(e, hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking));
    ME_HIP(e, hipEventCreateWithFlags(&e->pool_reduced, hipEventDisableTiming));
    ME_HIP(e, hipEventCreateWithFlags(&e->pool_copied, hipEventDisableTiming));
  }
  const int64_t n_doubles = moments_size(e->nr, e->nc);
  int rc = enqueue_pooled_moments(e, e->pool_dev, n_doubles);
  if (rc != ME_OK) return rc;
  ME_HIP(e, hipEventRecord(e->pool_reduced, e->stream));
  ME_HIP(e, hipStreamWaitEvent(e->copy_stream, e->pool_reduced, 0));
  if (allreduce) {
    const ncclResult_t nrc = e->rccl->all_reduce(e->pool_dev, e->pool_dev, (size_t)n_doubles, ncclDouble, ncclSum, e->comm, e->copy_stream);
    if (nrc != ncclSuccess) return fail(e, ME_ERR_HIP, std::string("ncclAllReduce: ") + e->rccl->error_string(nrc));
  }
  ME_HIP(e, hipMemcpyAsync(e->pool_host, e->pool_dev, sizeof(double) * (size_t)n_doubles, hipMemcpyDeviceToHost, e->copy_stream));
  ME_HIP(e, hipEventRecord(e->pool_copied, e->copy_stream));
  e->pool_pending = true;
  return ME_OK;
}
}  // namespace

int me_pooled_moments_begin(me_engine *e) {
  if (!e) return ME_ERR_INVALID;
  return begin_pooled(e, false);
}

int me_pooled_moments_allreduce_begin(me_engine *e) {
  if (!e) return ME_ERR_INVALID;
  return begin_pooled(e, true);
}

int me_pooled_moments_allreduce(me_engine *e, double *host_out, int64_t n_doubles) {
  if (!e || !host_out) return ME_ERR_INVALID;
  if (n_doubles != moments_size(e->nr, e->nc)) return fail(e, ME_ERR_INVALID, "wrong pooled-moment buffer length");
  const int rc = begin_pooled(e, true);
  return rc != ME_OK ? rc : me_pooled_moments_end(e, host_out, n_doubles);
}

int me_comm_unique_id(void *id_out, size_t bytes) {
  if (!id_out || bytes != ME_COMM_ID_BYTES) return fail(nullptr, ME_ERR_INVALID, "the unique id is ME_COMM_ID_BYTES bytes");
  static_assert(sizeof(ncclUniqueId) == ME_COMM_ID_BYTES, "ME_COMM_ID_BYTES must be sizeof(ncclUniqueId)");
  std::string err;
  const RcclApi *api = rccl_api(&err);
  if (!api) return fail(nullptr, ME_ERR_UNSUPPORTED, err);
  ncclUniqueId id;
  const ncclResult_t nrc = api->get_unique_id(&id);
  if (nrc != ncclSuccess) return fail(nullptr, ME_ERR_HIP, std::string("ncclGetUniqueId: ") + api->error_string(nrc));
  std::memcpy(id_out, &id, sizeof(id));
  return ME_OK;
}

int me_comm_init_rank(me_engine *e, const void *unique_id, size_t bytes, int32_t rank, int32_t world) {
  if (!e || !unique_id) return ME_ERR_INVALID;
  if (bytes != ME_COMM_ID_BYTES) return fail(e, ME_ERR_INVALID, "the unique id is ME_COMM_ID_BYTES bytes");
  if (world < 1 || rank < 0 || rank >= world) return fail(e, ME_ERR_INVALID, "rank out of range");
  if (e->comm) return fail(e, ME_ERR_STATE, "this engine already has a communicator (me_comm_destroy first)");
  if (e->pool_pending) return fail(e, ME_ERR_STATE, "a pooled-moment reduction is in flight");
  std::string err;
  const RcclApi *api = rccl_api(&err);
  if (!api) return fail(e, ME_ERR_UNSUPPORTED, err);
  ME_HIP(e, hipSetDevice(e->device));       // the communicator binds to the current device: one engine, one GPU, one rank
  ncclUniqueId id;
  std::memcpy(&id, unique_id, sizeof(id));
  ncclComm_t comm = nullptr;
  const ncclResult_t nrc = api->comm_init_rank(&comm, world, id, rank);   // collective: returns once every rank has joined
  if (nrc != ncclSuccess) return fail(e, ME_ERR_HIP, std::string("ncclCommInitRank: ") + api->error_string(nrc));
  e->rccl = api;
  e->comm = comm;
  e->comm_rank = rank;
  e->comm_world = world;
  return ME_OK;
}

int me_comm_destroy(me_engine *e) {
  if (!e) return ME_ERR_INVALID;
  if (!e->comm) return ME_OK;
  if (e->pool_pending) return fail(e, ME_ERR_STATE, "a pooled-moment reduction is in flight (me_pooled_moments_end first)");
  ME_HIP(e, hipSetDevice(e->device));
  if (e->copy_stream) ME_HIP(e, hipStreamSynchronize(e->copy_stream));
  const ncclResult_t nrc = e->rccl->comm_destroy(e->comm);
  e->comm = nullptr;
  e->comm_rank = 0;
  e->comm_world = 1;
  if (nrc != ncclSuccess) return fail(e, ME_ERR_HIP, std::string("ncclCommDestroy: ") + e->rccl->error_string(nrc));
  return ME_OK;
}

int me_comm_info(me_engine *e, int32_t *rank, int32_t *world, int32_t *rccl_version) {
  if (!e) return ME_ERR_INVALID;
  if (rank) *rank = e->comm ? e->comm_rank : -1;
  if (world) *world = e->comm ? e->comm_world : 0;
  if (rccl_version) {
    int v = 0;
    if (e->rccl) (void)e->rccl->get_version(&v);
    *rccl_version = v;
  }
  return ME_OK;
}

int me_pooled_moments_end(me_engine *e, double *host_out, int64_t n_doubles) {
  if (!e || !host_out) return ME_ERR_INVALID;
  if (!e->pool_pending) return fail(e, ME_ERR_STATE, "no pooled-moment reduction in flight: call me_pooled_moments_begin first");
  if (n_doubles != moments_size(e->nr, e->nc)) return fail(e, ME_ERR_INVALID, "wrong pooled-moment buffer length");
  ME_HIP(e, hipSetDevice(e->device));
  ME_HIP(e, wait_polling(e->pool_copied));
  e->pool_pending = false;
  std::memcpy(host_out, e->pool_host, sizeof(double) * (size_t)n_doubles);
  return ME_OK;
}

int me_set_shared_factor(me_engine *e, const double *packed_factor, int64_t n_doubles) {
  if (!e || !packed_factor) return ME_ERR_INVALID;
  if (e->cov_mode != ME_COV_POOLED) return fail(e, ME_ERR_STATE, "me_set_shared_factor needs cov_mode = ME_COV_POOLED");
  if (n_doubles != e->p) return fail(e, ME_ERR_INVALID, "wrong packed factor length");
  ME_HIP(e, hipSetDevice(e->device));
  ME_HIP(e, hipStreamSynchronize(e->stream));
  int rc = upload_shared_factor(e, packed_factor);
  if (rc != ME_OK) return rc;
  e->cov_kind = CK_SHARED;
  e->shared_host.assign(packed_factor, packed_factor + n_doubles);
  return ME_OK;
}

int me_get_shared_factor(me_engine *e, double *packed_factor, int64_t n_doubles, int32_t *is_set) {
  if (!e || !is_set) return ME_ERR_INVALID;
  *is_set = e->shared_host.empty() ? 0 : 1;
  if (!*is_set) return ME_OK;
  if (!packed_factor || n_doubles != e->p) return fail(e, ME_ERR_INVALID, "wrong packed factor length");
  std::copy(e->shared_host.begin(), e->shared_host.end(), packed_factor);
  return ME_OK;
}

int me_sync(me_engine *e) {
  if (!e) return ME_ERR_INVALID;
  ME_HIP(e, hipSetDevice(e->device));
  ME_HIP(e, hipStreamSynchronize(e->stream));
  return check_status(e);
}

int me_set_stream(me_engine *e, void *hip_stream) {
  if (!e) return ME_ERR_INVALID;
  ME_HIP(e, hipSetDevice(e->device));
  ME_HIP(e, hipStreamSynchronize(e->stream));
  if (e->own_stream) (void)hipStreamDestroy(e->stream);
  e->stream = (hipStream_t)hip_stream;
  e->own_stream = false;
  return ME_OK;
}

int me_time_steps(me_engine *e, int32_t n_launches, int32_t n_sweeps, float *elapsed_ms) {
  if (!e || !elapsed_ms) return ME_ERR_INVALID;
  if (n_launches <= 0 || n_sweeps <= 0) return fail(e, ME_ERR_INVALID, "n_launches and n_sweeps must be positive");
  ME_HIP(e, hipSetDevice(e->device));
  // the two events live with the engine: creating and destroying them per call cost ~20 us of a short timed region
  if (!e->time_start) {
    ME_HIP(e, hipEventCreate(&e->time_start));
    ME_HIP(e, hipEventCreate(&e->time_stop));
  }
  hipEvent_t start = e->time_start, stop = e->time_stop;
  ME_HIP(e, hipEventRecord(start, e->stream));
  for (int i = 0; i < n_launches; ++i) {
    StepLaunch l;
    fill_step_launch(e, l, n_sweeps);
    hipError_t err = e->ks->step(l, e->stream);
    if (err != hipSuccess) return fail(e, ME_ERR_HIP, std::string("step launch: ") + hipGetErrorString(err));
    e->step_index += (unsigned long long)n_sweeps;
    e->proposed += (unsigned long long)e->n * (unsigned long long)n_sweeps;
    e->widths_synced = true;
  }
  ME_HIP(e, hipEventRecord(stop, e->stream));
  ME_HIP(e, hipEventSynchronize(stop));
  ME_HIP(e, hipEventElapsedTime(elapsed_ms, start, stop));
  return ME_OK;
}

int me_detect_equilibration(int32_t device_id, const double *series, int64_t n_series, int64_t length, int32_t fast,
                            int32_t nskip, int64_t *t0, double *g, double *neff_max) {
  if (!series || !t0 || !g || !neff_max) return fail(nullptr, ME_ERR_INVALID, "null pointer");
  if (n_series <= 0 || length < 3 || nskip < 1) return fail(nullptr, ME_ERR_INVALID, "need n_series > 0, length >= 3, nskip >= 1");
  ME_HIP(nullptr, hipSetDevice(device_id));
  const size_t m = (size_t)(length - 1), ns = (size_t)n_series;
  double *d_series = nullptr, *d_scratch = nullptr, *d_g = nullptr, *d_neff = nullptr;
  long long *d_t0 = nullptr;
  hipStream_t stream = nullptr;
  auto cleanup = [&]() {
    for (void *p : {(void *)d_series, (void *)d_scratch, (void *)d_g, (void *)d_neff, (void *)d_t0})
      if (p) (void)hipFree(p);
  };
#define ME_EQ_HIP(call)                                                                    \
  do {                                                                                     \
    hipError_t err__ = (call);                                                             \
    if (err__ != hipSuccess) {                                                             \
      cleanup();                                                                           \
      return fail(nullptr, ME_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(err__)); \
    }                                                                                      \
  } while (0)
  ME_EQ_HIP(hipMalloc((void **)&d_series, sizeof(double) * ns * (size_t)length));
  ME_EQ_HIP(hipMalloc((void **)&d_scratch, sizeof(double) * 2 * ns * m));
  ME_EQ_HIP(hipMalloc((void **)&d_g, sizeof(double) * ns));
  ME_EQ_HIP(hipMalloc((void **)&d_neff, sizeof(double) * ns));
  ME_EQ_HIP(hipMalloc((void **)&d_t0, sizeof(long long) * ns));
  ME_EQ_HIP(hipMemcpy(d_series, series, sizeof(double) * ns * (size_t)length, hipMemcpyHostToDevice));
  ME_EQ_HIP(launch_detect_equilibration(d_series, n_series, length, fast, nskip, d_scratch, d_t0, d_g, d_neff, stream));
  ME_EQ_HIP(hipDeviceSynchronize());
  static_assert(sizeof(long long) == sizeof(int64_t), "t0 is copied out as int64");
  ME_EQ_HIP(hipMemcpy(t0, d_t0, sizeof(int64_t) * ns, hipMemcpyDeviceToHost));
  ME_EQ_HIP(hipMemcpy(g, d_g, sizeof(double) * ns, hipMemcpyDeviceToHost));
  ME_EQ_HIP(hipMemcpy(neff_max, d_neff, sizeof(double) * ns, hipMemcpyDeviceToHost));
#undef ME_EQ_HIP
  cleanup();
  // a constant series has no autocorrelation: the host convention is (0, 1, 1)
  for (size_t s = 0; s < ns; ++s) {
    const double *row = series + s * (size_t)length;
    bool constant = true;
    for (int64_t i = 1; i < length && constant; ++i) constant = row[i] == row[0];
    if (constant) { t0[s] = 0; g[s] = 1.0; neff_max[s] = 1.0; }
  }
  return ME_OK;
}

int me_last_error(me_engine *e, char *buf, size_t buf_bytes) {
  if (!buf || buf_bytes == 0) return ME_ERR_INVALID;
  const std::string &s = e ? e->err : g_create_error;
  const size_t k = std::min(buf_bytes - 1, s.size());
  std::memcpy(buf, s.data(), k);
  buf[k] = 0;
  return ME_OK;
}

}  // extern "C"
