/*
 * metropolis_engine.h -- C ABI of libmetropolis_hip.so, the MI355X (gfx950) many-chain Metropolis engine.
 *
 * This is the drop-in boundary for the reference's hot path.  The reference has no FFI: its "plugin API" is
 * the Python class MetropolisEngine (/root/reference/metropolisengine/metropolis_engine.py:10-463).  Each
 * entry point below replaces one piece of that class for N independent chains on one GPU; the Python class
 * metropolisengine_amd.MetropolisEngine binds them with ctypes (see INTEGRATION.md for the stub).
 *
 *   me_create            <- MetropolisEngine.__init__               metropolis_engine.py:17-133
 *   me_step              <- step_all / step_real_group / step_complex_group
 *                           (+ draw_*_group, metropolis_decision, update_*_sigma)   :209-338, :429-456
 *   me_measure           <- measure / measure_real_system / measure_complex_system
 *                           (+ update_*_mean, update_covariance_matrix_*, observables) :342-427, :458-463
 *   me_get / me_set      <- attribute reads/writes (real_params, real_mean, covariance_matrix_real, ...)
 *                           README.md:49-51; also the only "checkpoint" the reference has (ctor warm start, :17)
 *   me_accept_stats      <- the bool returned by step_all (:259), accumulated
 *   me_pooled_moments*   <- no reference equivalent (ensemble estimate across chains)
 *   me_comm_*, me_pooled_moments_allreduce*  <- no reference equivalent (the one collective: RCCL all-reduce of the moments)
 *   me_last_error        <- Python exceptions (:39 ValueError, :92/:438 AssertionError, numpy ValueError :270)
 *
 * Conventions: plain pointers and sizes only; every function returns an me_status (0 = ok); the library owns
 * all device memory; host buffers are caller-owned.  Host-side layouts are row-major [chain][component] doubles
 * whatever the device dtype.  Calls on one engine must not be concurrent (the reference is not reentrant
 * either); different engines may be driven from different threads.  me_step/me_measure are asynchronous on the
 * engine's HIP stream; me_get/me_accept_stats/me_pooled_moments synchronise that stream.
 *
 * State vector of one chain: D = n_real + 2*n_complex reals, ordered
 *   [ real params | real parts of complex params | imaginary parts of complex params ].
 */
#ifndef METROPOLIS_ENGINE_H
#define METROPOLIS_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ME_ABI_VERSION 1

typedef struct me_engine me_engine; /* opaque handle */

typedef enum me_status {
  ME_OK = 0,
  ME_ERR_INVALID = 1,     /* bad argument (the reference raises ValueError / AssertionError) */
  ME_ERR_UNSUPPORTED = 2, /* no kernel compiled for this (dtype, n_real, n_complex, energy, ...) */
  ME_ERR_HIP = 3,         /* HIP runtime error, no device, out of memory */
  ME_ERR_NUMERIC = 4,     /* a chain produced a non-finite energy / non-positive Cholesky pivot / width */
  ME_ERR_STATE = 5        /* call not valid in the engine's current state */
} me_status;

typedef enum me_dtype { ME_F32 = 0, ME_F64 = 1 } me_dtype;

/* Energy kinds: the reference takes a Python callable (metropolis_engine.py:20, :111-120); a GPU build needs
 * a device-side specification instead.  Coefficients are doubles, converted to the device dtype. */
typedef enum me_energy_kind {
  ME_ENERGY_ISO_QUAD = 0,   /* coeffs {a}: a (sum x_i^2 + sum |z_j|^2)                  README.md:26-27 */
  ME_ENERGY_DIAG_QUAD = 1,  /* coeffs {a_0..a_{nr-1}, b_0..b_{nc-1}}: sum a x^2 + sum b |z|^2 */
  ME_ENERGY_DENSE_QUAD = 2, /* coeffs A[D*D] row-major: x^T A x over the real D-vector */
  ME_ENERGY_LANDAU_TOY = 3, /* coeffs {k, alpha, beta}, nr=2, nc=1   demo/toymodel_complex_and_real.py:17-26 */
  ME_ENERGY_CYLINDER = 4,   /* coeffs {kappa, gamma, wavenumber}: cylinder-style surrogate */
  ME_ENERGY_USER = 5,       /* user-written device function from a plugin (include/metropolis_user_energy.h),
                               inlined into the kernels; coeffs are handed to it in device memory */
  ME_ENERGY_USER_INDIRECT = 6, /* the same function called through a __device__ function pointer */
  ME_ENERGY_LANDAU_TERMS = 7  /* the Landau toy as the reference demo passes it: a term dictionary {"field": both
                                 groups, "area": real group} (demo/toymodel_complex_and_real.py:17-33); two ledger rows */
} me_energy_kind;

/* Hard-wall predicate evaluated before the energy (metropolis_engine.py:142-146, :247-249). */
typedef enum me_reject_kind {
  ME_REJECT_NONE = 0,
  ME_REJECT_ABS_REAL0_GE = 1, /* reject when |x_0| >= reject_bound  (legacy /metropolis_engine.py:139-141) */
  ME_REJECT_USER = 2          /* the plugin's me_user_reject (include/metropolis_user_energy.h); user energies only */
} me_reject_kind;

/* Which matrix shapes the proposals. */
typedef enum me_cov_mode {
  ME_COV_REFERENCE = 0, /* initial matrix until measure_step_counter > 50, then each chain's own running
                           covariance (metropolis_engine.py:389, :416-427) */
  ME_COV_FIXED = 1,     /* keep the initial matrix for ever (measure() still updates the statistics) */
  ME_COV_POOLED = 2     /* one factor shared by all chains, installed with me_set_shared_factor */
} me_cov_mode;
/* Availability: every mode for every parameter space up to 96 real degrees of freedom (n_real + 2 n_complex; plugins and
   kernel sets compiled for one size: any size).  Beyond that -- the runtime-dimension kernel set -- ME_COV_FIXED and
   ME_COV_REFERENCE for every space whose D x 64 values fit the LDS (float64: D <= 290), ME_COV_POOLED for pure real spaces
   (two such blocks: D <= 145 in float64); me_create answers ME_ERR_UNSUPPORTED otherwise. */

/* Per-chain fields for me_get / me_set; components per chain in brackets (P = nr(nr+1)/2 + nc^2). */
typedef enum me_field {
  ME_FIELD_PARAMS = 0,   /* [D]  current state */
  ME_FIELD_ENERGY = 1,   /* [T]  the energy ledger, one row per energy term (metropolis_engine.py:111-116, :152-155);
                                 T = 1 ("total") unless the energy is a term dictionary (me_energy_terms) */
  ME_FIELD_WIDTH = 2,    /* [1] the group's sampling width; mixed engines [3]: sampling_width, real group, complex group */
  ME_FIELD_MEAN = 3,     /* [D]  running mean */
  ME_FIELD_COV = 4,      /* [P]  running covariance, packed: real block row-major lower triangle, then for
                                 each complex row i: (Re,Im) of K_ij for j<i, then K_ii */
  ME_FIELD_OBS_MEAN = 5, /* [2nr+nc] running mean of |x_r|, |z_c|, x_r^2 */
  ME_FIELD_FACTOR = 6,   /* [P]  Cholesky factors used by the proposals (same packing; conj(K) for complex) */
  ME_FIELD_ENERGY_TOTAL = 7 /* [1] ME_FLAG_REFERENCE_ENERGY_LEDGERS only: the separate `energy_total` that step_all of a
                               mixed engine compares against and updates (metropolis_engine.py:252-255) */
} me_field;

/* me_config.flags */
typedef enum me_flags {
  ME_FLAG_TRACK_COVARIANCE = 1, /* parameter spaces whose per-chain matrix is too large for registers (more than 160
                                  packed entries, e.g. 64 real parameters) keep it only where the proposals need it
                                  (ME_COV_REFERENCE, pure real spaces: streamed kernels); with ME_COV_FIXED / ME_COV_POOLED
                                  they keep means and observables only.  With this flag measure() also maintains each
                                  chain's running covariance (metropolis_engine.py:416-427) there, as statistics -- P values
                                  per chain, read and written once per measure().  Smaller spaces always track it. */
  ME_FLAG_REFERENCE_ENERGY_LEDGERS = 2 /* reproduce the reference's TWO energy ledgers: step_all of a
                                  mixed engine compares against and updates `energy_total` only (metropolis_engine.py:
                                  252-255), group steps compare against and update `energy[term]` only (:214-221,
                                  :230-237), so a driver that mixes both call styles decides on stale energies exactly
                                  as the reference does.  Default: one coherent ledger per chain.  Mixed engines only. */
} me_flags;

typedef struct me_config {
  uint32_t abi_version; /* ME_ABI_VERSION */
  int32_t device_id;    /* HIP device ordinal */
  int64_t n_chains;     /* chains held by THIS engine (local shard) */
  uint64_t chain_offset; /* global id of local chain 0: random streams are addressed by global id */
  uint64_t seed;
  int32_t n_real;
  int32_t n_complex;
  int32_t dtype;        /* me_dtype of the device state and arithmetic */
  int32_t cov_mode;     /* me_cov_mode */
  double temp;              /* >= 0 (metropolis_engine.py:91-92) */
  double target_acceptance; /* reference default 0.3 */
  double sampling_width;    /* reference default 0.05 */
  int32_t energy_kind;      /* me_energy_kind */
  int32_t n_energy_coeffs;
  const double *energy_coeffs;
  int32_t reject_kind; /* me_reject_kind */
  int32_t flags;       /* me_flags */
  double reject_bound;
  const double *initial_params;     /* [D], broadcast to every chain */
  const double *covariance_real;    /* [nr*nr] row-major or NULL = identity (metropolis_engine.py:63-66) */
  const double *covariance_complex; /* [nc*nc*2] row-major (Re,Im) or NULL = identity (:67-70) */
  const char *user_energy_name;     /* ME_ENERGY_USER*: name the plugin was built with (NULL otherwise) */
} me_config;

int me_abi_version(void);

/* Load a user-energy plugin library (built from the user's device function around the same kernels; it registers
 * itself).  The reference's counterpart is simply passing a Python callable (metropolis_engine.py:20). */
int me_load_plugin(const char *path);

int me_create(const me_config *config, me_engine **out);
int me_destroy(me_engine *engine);

/* n_sweeps fused propose->energy->accept->adapt sweeps over all chains in one launch; n_sweeps = 1 is one
 * reference step_all(). */
int me_step(me_engine *engine, int32_t n_sweeps);
/* Step kinds beyond step_all: a mixed engine's step_real_group / step_complex_group called directly
 * (metropolis_engine.py:225-239, :209-223: only that group moves and only its own width adapts) and the
 * magnitude-phase pair that replaces step_complex_group under complex_sample_method="magnitude-phase"
 * (:168-207; two accept decisions per step).  On pure-real / pure-complex engines the group kinds are step_all
 * (:46, :56).  me_step(e, k) == me_step_kind(e, ME_STEP_ALL, k). */
typedef enum me_step_kind_t {
  ME_STEP_ALL = 0,
  ME_STEP_REAL_GROUP = 1,
  ME_STEP_COMPLEX_GROUP = 2,
  ME_STEP_COMPLEX_MAGNITUDE_PHASE = 3
} me_step_kind_t;
int me_step_kind(me_engine *engine, int32_t kind, int32_t n_sweeps);
int me_measure(me_engine *engine);
/* One cycle of the reference's driver loop (README.md:41-44: `for j in range(k): step_all()` then `measure()`):
 * me_cycle(e, k) has exactly the results of me_step(e, k); me_measure(e), but where a fused kernel exists for the engine
 * (per-chain covariance kept in registers: up to 160 packed entries, step_all sweeps, identity or per-chain shape) it
 * is ONE launch -- state, energy and width are read once and written once, the sweeps run in registers, mean /
 * observables / covariance / factors are updated from the registers.  Otherwise the two launches are issued.
 * me_cycle_stats: how many me_cycle calls of this engine ran as one launch. */
int me_cycle(me_engine *engine, int32_t n_sweeps);
int me_cycle_stats(me_engine *engine, uint64_t *fused_cycles);
/* set_reject_condition (metropolis_engine.py:142-146): the wall predicate is a launch parameter and may be changed
 * between steps (in the reference this setter is the only working way to install one, quirk Q6). */
int me_set_reject_condition(me_engine *engine, int32_t reject_kind, double reject_bound);
/* set_energy_function (metropolis_engine.py:134-138): bind the engine to another energy of the SAME parameter space -- a
 * built-in kind (except ME_ENERGY_DENSE_QUAD) or a loaded user plugin -- re-collect the energy terms (the ledger is resized
 * when their number changes) and re-evaluate every term at the current state (initialize_energy_dict, :152-155).  The
 * chain state, widths, running statistics and counters stay.  Synchronous. */
int me_set_energy(me_engine *engine, int32_t energy_kind, const double *energy_coeffs, int32_t n_energy_coeffs,
                  const char *user_energy_name);
/* Test hook (float64 engines): the same step with the random draws supplied by the caller instead of Philox --
 * for the Gaussian kinds normals [n_sweeps][n_chains][D] standard normals and uniforms [n_sweeps][n_chains][1] accept
 * draws; for ME_STEP_COMPLEX_MAGNITUDE_PHASE normals [n_sweeps][n_chains][nc] and uniforms [n_sweeps][n_chains][nc+2]
 * = {accept (magnitude stage), nc phases in (0,1), accept (phase stage)}.  This is how the reference's golden
 * trajectories (tests/golden/, injected-stream runs of metropolis_engine.py) are replayed through the HIP kernels.
 * Synchronous. */
int me_step_injected(me_engine *engine, int32_t kind, int32_t n_sweeps, const double *normals, const double *uniforms);

int me_field_components(me_engine *engine, int32_t field, int32_t *n_components);
/* Rows of the energy ledger = number of energy terms (self.energy_term_names, metropolis_engine.py:112-118).  Group
 * steps re-evaluate and compare only the terms registered for their group (:214-221, :230-237). */
int me_energy_terms(me_engine *engine, int32_t *n_terms);
/* Copy chains [chain_begin, chain_begin + n_chains) of a field to/from host doubles [chain][component]. */
int me_get(me_engine *engine, int32_t field, int64_t chain_begin, int64_t n_chains, double *dst);
int me_set(me_engine *engine, int32_t field, int64_t chain_begin, int64_t n_chains, const double *src);
/* Re-evaluate the stored energies after me_set(ME_FIELD_PARAMS). */
int me_recompute_energy(me_engine *engine);

/* Robbins-Monro constants the engine derived (metropolis_engine.py:101-107): alpha, m, ratio. */
int me_constants(me_engine *engine, double *alpha, int32_t *m, double *ratio);
int me_counters(me_engine *engine, uint64_t *step_index, uint64_t *measure_step_counter);
int me_set_counters(me_engine *engine, uint64_t step_index, uint64_t measure_step_counter);
int me_accept_stats(me_engine *engine, uint64_t *accepted, uint64_t *proposed);
/* Restore the acceptance counters of a checkpoint (the reference has no counterpart: its only resume is the constructor
 * warm start, metropolis_engine.py:17). */
int me_set_accept_stats(me_engine *engine, uint64_t accepted, uint64_t proposed);

/* Ensemble sums over the local chains, fp64:
 *   [ n, sum x (D), sum x x^T (D(D+1)/2, row-major lower triangle), sum obs (2nr+nc), accepted, proposed ]
 * me_pooled_moments_size gives the length.  The _device form writes into caller-provided DEVICE memory (e.g. a
 * torch tensor) so that the caller can all-reduce it with RCCL in place; it is enqueued on the engine's stream
 * and followed by a stream synchronise. */
int me_pooled_moments_size(me_engine *engine, int64_t *n_doubles);
int me_pooled_moments(me_engine *engine, double *host_out, int64_t n_doubles);
int me_pooled_moments_device(me_engine *engine, void *device_out, int64_t n_doubles);
/* Split form that does not stall the engine's stream: _begin enqueues the reduction behind the work already queued and
 * the copy to the host on a second stream, and returns at once; steps and measures enqueued afterwards run while the
 * result travels.  _end waits for that copy only and hands out the moments of the state at the time of _begin.  One
 * reduction may be in flight per engine (ME_ERR_STATE otherwise). */
int me_pooled_moments_begin(me_engine *engine);
int me_pooled_moments_end(me_engine *engine, double *host_out, int64_t n_doubles);
/* RCCL behind the C ABI: the ONE collective of the engine, a sum all-reduce (fp64) of the pooled-moment vector over the
 * ranks of a multi-GPU job (one process and one engine per GPU; chains shard with no other exchange).  No reference
 * counterpart (the reference runs one chain in one process; BASELINE.json: "RCCL all-reduce only for the pooled
 * covariance/observables").  librccl is loaded on first use; ME_ERR_UNSUPPORTED when it cannot be.
 *   me_comm_unique_id   rank 0: ncclGetUniqueId into a caller buffer of ME_COMM_ID_BYTES, to be handed to every rank by
 *                       whatever channel the host program has (MPI, a file, torch.distributed, a socket)
 *   me_comm_init_rank   every rank: ncclCommInitRank on the engine's device (collective: returns when all have joined)
 *   me_comm_destroy     ncclCommDestroy (me_destroy does it too)
 *   me_comm_info        rank / world of the communicator (-1 / 0 without one) and the RCCL version code
 *   me_pooled_moments_allreduce        k_pool_reduce -> k_pool_finish on the engine's stream, then on the engine's second
 *                       stream behind an event: ncclAllReduce(sum, fp64, in place) -> copy to pinned host memory; waits for
 *                       that copy and hands out the moments of ALL ranks' chains (same layout as me_pooled_moments: n,
 *                       accepted and proposed are global too)
 *   me_pooled_moments_allreduce_begin  the same without waiting: collect with me_pooled_moments_end.  The engine's main
 *                       stream never waits for another rank and the host never synchronises inside a cycle
 * Every rank must call the all-reduce forms the same number of times, in the same order. */
#define ME_COMM_ID_BYTES 128
int me_comm_unique_id(void *id_out, size_t bytes);
int me_comm_init_rank(me_engine *engine, const void *unique_id, size_t bytes, int32_t rank, int32_t world);
int me_comm_destroy(me_engine *engine);
int me_comm_info(me_engine *engine, int32_t *rank, int32_t *world, int32_t *rccl_version);
int me_pooled_moments_allreduce(me_engine *engine, double *host_out, int64_t n_doubles);
int me_pooled_moments_allreduce_begin(me_engine *engine);
/* Install the shared proposal factor of ME_COV_POOLED: packed like ME_FIELD_FACTOR, [P] doubles. */
int me_set_shared_factor(me_engine *engine, const double *packed_factor, int64_t n_doubles);
/* Tuning knob (process-wide, default 224 MiB): launches whose working set exceeds this many bytes access their read-once
 * / write-once fields (packed covariance and factors, large running-mean fields) with the non-temporal cache policy, so
 * that the chain state stays resident in the 256 MiB Infinity Cache; 0 = always, a huge value = never. */
int me_set_cache_budget(int64_t bytes);
/* The factor last installed with me_set_shared_factor ([P] doubles, exactly as given); *is_set = 0 and the buffer is left
 * alone when none has been installed.  Part of a checkpoint of a ME_COV_POOLED engine. */
int me_get_shared_factor(me_engine *engine, double *packed_factor, int64_t n_doubles, int32_t *is_set);

/* Time series (metropolis_engine.py:350-356, :466-479): every me_measure appends one row per traced chain
 * (chains t*stride, t < n_traced) to a device-side series; me_trace_get returns it as doubles
 * [row][column][traced chain] with columns {params (D), energy, widths (1, or 3 for mixed engines)}.
 * n_traced = 0 switches recording off (the default). */
int me_trace_enable(me_engine *engine, int64_t n_traced, int64_t stride);
int me_trace_shape(me_engine *engine, int64_t *rows, int64_t *cols, int64_t *n_traced);
int me_trace_get(me_engine *engine, double *dst, int64_t n_doubles);

int me_sync(me_engine *engine);
/* Use an existing HIP stream (hipStream_t passed as void*) instead of the engine's own. */
int me_set_stream(me_engine *engine, void *hip_stream);
/* Enqueue n_launches launches of n_sweeps sweeps bracketed by HIP events on the engine's stream and return the
 * elapsed milliseconds (device time of the whole span). */
int me_time_steps(me_engine *engine, int32_t n_launches, int32_t n_sweeps, float *elapsed_ms);

/* Text of the last error on this engine (or of the last failed me_create when engine is NULL). */
int me_last_error(me_engine *engine, char *buf, size_t buf_bytes);

/* Capability query: 1 if a kernel set for this combination is compiled in. */
int me_supported(int32_t dtype, int32_t n_real, int32_t n_complex, int32_t energy_kind);

/* Equilibration detection for a batch of recorded series (host arrays in, host arrays out; engine-independent):
 * for each of the n_series rows of series[n_series][length] the production start t0 that maximises the number of
 * effectively uncorrelated samples (length - t0 + 1) / g(t0), with g the statistical inefficiency of the tail.
 * Replaces the per-column pymbar.timeseries.detectEquilibration calls of statistics.py:25-48 /
 * metropolis_engine.py:481-504 (PARITY UNPINNED: pymbar is absent; this follows the published algorithm exactly as
 * metropolisengine_amd/statistics.py restates it).  fast != 0: lag increments grow by one (pymbar's fast=True);
 * nskip: stride of the candidate starts.  Constant series report (0, 1, 1). */
int me_detect_equilibration(int32_t device_id, const double *series, int64_t n_series, int64_t length, int32_t fast,
                            int32_t nskip, int64_t *t0, double *g, double *neff_max);

#ifdef __cplusplus
}
#endif
#endif /* METROPOLIS_ENGINE_H */
