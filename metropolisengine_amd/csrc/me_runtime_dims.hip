// Runtime-dimension kernel set: parameter spaces beyond what the register-resident kernels are built for
// (n_real + 2 n_complex > 96).  The reference has no limit on the number of parameters (metropolis_engine.py:41-60);
// the compile-time-dimension kernels of me_device.h keep a chain's whole state in registers and stop at 96 real degrees of
// freedom.  Here the dimensions are launch arguments and the state STREAMS:
//
//   k_step_runtime   pass 1: for each coordinate d -- load x_d, draw its normal, form x'_d = x_d + sigma g_d, add its term to
//                    the proposed energy.  Nothing is stored.  Then wall, accept rule, width adaptation (:247-259,
//                    :319-338, :429-456).  Pass 2, only if some lane of the wavefront accepted: the normals are drawn AGAIN
//                    (Philox is counter-based, so "again" is a handful of integer instructions, not a stored array) and
//                    the accepted lanes write x_d + sigma g_d.  A sweep therefore reads the state once (twice for
//                    accepted chains, from cache) and writes only what changed -- less traffic than a register-resident
//                    sweep, paid for with drawing the normals twice.
//   k_measure_runtime, k_init_energy_runtime: the same loops over d for the running means / observables / energy.
//
//   k_step_runtime_lds  what cannot stream: a DENSE quadratic form needs all of x' at once, a SHARED proposal factor all of
//                    g.  Both are parked in LDS ([D][64 lanes], one wavefront per block) and the two O(D^2) triangle
//                    products run on the matrix cores (tri_rows_mfma: Y = T V for the wavefront's 64 chains with
//                    v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32, T = folded triangle of A, T_ij = A_ij + A_ji, or
//                    the packed factor L, gathered from global memory as the A operand, V from LDS as the B operand).
//                    Limits: the LDS a block may use (D <= ~290 in float64 with the identity shape, ~145 with a shared
//                    factor; twice that in float32).
//
//   per-chain proposal shapes (ME_COV_REFERENCE: the reference's semantics, :416-427 feeding :261-302):
//                    k_measure_runtime_cov keeps each chain's running covariance (packed, tile-major, streamed; real block
//                    and Hermitian block), k_factor_runtime / k_factor_runtime_complex refresh its Cholesky factors row by
//                    row, and k_step_runtime_lds<CK_PER_CHAIN> forms x' = x + sigma L_chain g by columns (four real columns at
//                    a time, complex columns one by one) with only x' parked in LDS.  Written for
//                    completeness at any size the LDS admits (float64: 290 parameters), not for speed: a step reads the
//                    chain's whole factor (D(D+1)/2 values), a refresh is D^3/6 dependent multiply-adds per lane.
//
// Supported: ME_ENERGY_ISO_QUAD, ME_ENERGY_DIAG_QUAD (streamed), ME_ENERGY_DENSE_QUAD (LDS); the identity proposal
// shape (ME_COV_FIXED), per-chain shapes (ME_COV_REFERENCE) and, for pure real spaces, one shared factor
// (ME_COV_POOLED); running per-chain covariances for statistics (ME_FLAG_TRACK_COVARIANCE); the built-in wall, step_all and
// group-wise steps of mixed engines.  The word layout of a step is the one of every
// other kernel (oracle/philox.py): normal i belongs to coordinate i, word 2 ceil(D/2) is the accept uniform.
#include <type_traits>

#include "me_device.h"
#include "me_per_device.h"

namespace me {
namespace {

// triangle products of the LDS-resident kernel: 1 = matrix cores (tri_rows_mfma), 0 = VALU with v_readlane broadcasts
#ifndef ME_RUNTIME_MFMA
#define ME_RUNTIME_MFMA 1
#endif
constexpr size_t kRuntimeLdsLimit = 150 * 1024;      // dynamic LDS a block of the LDS-resident step kernel may ask for

template <typename R>
struct RuntimeStep {
  int nr, nc, group, energy_kind;
  R iso;                 // ME_ENERGY_ISO_QUAD: a
  const R *weights;      // ME_ENERGY_DIAG_QUAD: D weights (real, then the complex weights twice), device memory
};

// (fma_, me_device.h: one explicit fma in both passes -- the state written by pass 2 is bit for bit the x' whose energy
// pass 1 summed)

// Safe by construction: `weights` is non-null for ME_ENERGY_DIAG_QUAD only (fill() below), every other kind gets `iso`
// (0 for the dense form, whose energy never comes through here).  Round 2's last GPU call core-dumped in an experiment
// build that compiled the dense branch of k_step_runtime_lds out: the kernel then fell through to this function with
// energy_kind = ME_ENERGY_DENSE_QUAD and dereferenced the null `weights` (tools/README.md, "the rt_noenergy fault").
template <typename R>
__device__ __forceinline__ R weight_of(const RuntimeStep<R> &p, int d) {
  return p.weights ? p.weights[d] : p.iso;
}

template <typename R>
__global__ void __launch_bounds__(kBlockThreads) k_step_runtime(StepArgs<R> a, RuntimeStep<R> p) {
  using N_ = Num<R>;
  N_::prepare();
  const int nr = p.nr, nc = p.nc, D = nr + 2 * nc;
  const int NW = 2 * ((D + 1) / 2);                 // words consumed by the Box-Muller pairs
  const bool mixed = nr > 0 && nc > 0;
  const int d0 = p.group == GROUP_COMPLEX ? nr : 0, d1 = p.group == GROUP_REAL ? nr : D;   // coordinates that move
  unsigned int wave_accepted = 0;
  bool bad_energy = false, bad_width = false;
  const long long stride = (long long)gridDim.x * kBlockThreads, n = a.n;
  for (long long c = (long long)blockIdx.x * kBlockThreads + threadIdx.x; c < n; c += stride) {
    R *xc = a.x + c;
    R e = a.energy[c];
    const int wrow = mixed ? p.group : 0;
    R w = a.width[(long long)wrow * n + c];
    R w_r = w, w_c = w;
    if (mixed && p.group == GROUP_ALL && a.split_widths) {
      w_r = a.width[(long long)GROUP_REAL * n + c];
      w_c = a.width[(long long)GROUP_COMPLEX * n + c];
    }
    const unsigned long long gid = a.chain_offset + (unsigned long long)c;
    for (int s = 0; s < a.n_sweeps; ++s) {
      const unsigned long long step = a.step_index + (unsigned long long)s;
      auto block_of = [&](int b) {
        U4 ctr;
        ctr.x = (uint32_t)gid;
        ctr.y = (uint32_t)(gid >> 32);
        ctr.z = (uint32_t)step;
        ctr.w = ((uint32_t)(step >> 32) << 16) | (uint32_t)b;
        return philox4x32_10(ctr, a.seed_lo, a.seed_hi);
      };
      const R s_r = w_r, s_c = w_c * R(0.70710678118654752440);
      // ---- pass 1: proposed energy (and the proposed x_0 for the wall)
      R e_new = R(0), xp0 = R(0);
      for (int b = 0; 4 * b < NW; ++b) {
        const U4 o = block_of(b);
        R g[4];
        N_::normal_pair(o.x, o.y, g[0], g[1]);
        N_::normal_pair(o.z, o.w, g[2], g[3]);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int d = 4 * b + t;
          if (d < D) {
            const R xd = xc[(long long)d * n];
            const bool moves = d >= d0 && d < d1;
            const R xpd = moves ? fma_(d < nr ? s_r : s_c, g[t], xd) : xd;
            e_new += weight_of(p, d) * xpd * xpd;
            if (d == 0) xp0 = xpd;
          }
        }
      }
      const U4 ow = block_of(NW >> 2);
      const uint32_t uword = (NW & 3) == 0 ? ow.x : ow.z;      // word NW is output NW % 4 (0 or 2) of block NW / 4
      const R u = N_::unit(uword);
      bool rejected = false;
      if (a.reject_kind == ME_REJECT_ABS_REAL0_GE) rejected = !(N_::abs_(xp0) < a.reject_bound);
      const R diff = e_new - e;
      bool accept = diff <= R(0);
      if (a.temp > R(0)) accept = accept || N_::uphill(u, diff, a.inv_temp, a.inv_temp_log2e);
      accept = accept && !rejected;
      bad_energy |= (!rejected && !N_::finite(e_new));
      // ---- pass 2: the same normals again, accepted lanes commit
      if (__ballot(accept)) {
        for (int b = 0; 4 * b < NW; ++b) {
          if (4 * b + 3 < d0 || 4 * b >= d1) continue;         // no moving coordinate in this block
          const U4 o = block_of(b);
          R g[4];
          N_::normal_pair(o.x, o.y, g[0], g[1]);
          N_::normal_pair(o.z, o.w, g[2], g[3]);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int d = 4 * b + t;
            if (accept && d >= d0 && d < d1) xc[(long long)d * n] = fma_(d < nr ? s_r : s_c, g[t], xc[(long long)d * n]);
          }
        }
      }
      e = accept ? e_new : e;
      w = N_::adapt(w, accept, a.ratio, a.p, a.damping, a.up, a.down);
      if (p.group == GROUP_ALL) w_r = w_c = w;
      else if (p.group == GROUP_REAL) w_r = w;
      else w_c = w;
      wave_accepted += (unsigned int)__popcll(__ballot(accept));
    }
    bad_width |= !(w > R(0));
    a.energy[c] = e;
    a.width[(long long)wrow * n + c] = w;
  }
  if ((threadIdx.x & 63) == 0 && wave_accepted) {
    unsigned long long *slot = a.accept_slots + (size_t)blockIdx.x * (kBlockThreads >> 6) + (threadIdx.x >> 6);
    *slot += (unsigned long long)wave_accepted;
  }
  const unsigned int bits = (bad_energy ? ST_NONFINITE_ENERGY : 0u) | (bad_width ? ST_BAD_WIDTH : 0u);
  if (bits) atomicOr(a.status, bits);
}

template <typename R>
__global__ void __launch_bounds__(kBlockThreads) k_measure_runtime(MeasureArgs<R> a, int nr, int nc) {
  using N_ = Num<R>;
  const int D = nr + 2 * nc, nobs = 2 * nr + nc;
  const long long stride = (long long)gridDim.x * kBlockThreads, n = a.n;
  for (long long c = (long long)blockIdx.x * kBlockThreads + threadIdx.x; c < n; c += stride) {
    for (int d = 0; d < D; ++d) {                                                          // :404-410
      const long long i = (long long)d * n + c;
      a.mean[i] = a.mean[i] * a.keep + a.x[i] * a.inv_i;
    }
    for (int k = 0; k < nobs; ++k) {                                                       // :458-463, :412-414
      R o;
      if (k < nr) o = N_::abs_(a.x[(long long)k * n + c]);
      else if (k < nr + nc) {
        const R re = a.x[(long long)k * n + c], im = a.x[(long long)(k + nc) * n + c];
        o = N_::sqrt_(re * re + im * im);
      } else {
        const R v = a.x[(long long)(k - nr - nc) * n + c];
        o = v * v;
      }
      const long long i = (long long)k * n + c;
      a.obs_mean[i] = a.obs_mean[i] * a.keep + o * a.inv_i;
    }
  }
}

// measure() of an engine that keeps per-chain covariance matrices at runtime dimensions (cov_mode="reference" on pure real
// spaces, or the tracking flag): means and observables as above, then the covariance recursion of k_measure's streaming
// path (me_device.h: delta = x - mu_old parked in LDS, the packed entries of the tile-major field walked in their own
// order, batches of loads ahead of their updates; :416-427 in the one-pass form).  One wavefront per block, one lane per
// chain, dynamic LDS delta[D][64].
template <typename R>
__global__ void __launch_bounds__(kStepThreads) k_measure_runtime_cov(MeasureArgs<R> a, int nr, int nc) {
  using N_ = Num<R>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_rt[];
  R(*delta)[kStepThreads] = reinterpret_cast<R(*)[kStepThreads]>(smem_rt);
  const int D = nr + 2 * nc, nobs = 2 * nr + nc, lane = threadIdx.x;
  const long long P = (long long)nr * (nr + 1) / 2 + (long long)nc * nc;
  const bool mixed = nr > 0 && nc > 0;
  const long long stride = (long long)gridDim.x * kStepThreads, n = a.n;
  for (long long c = (long long)blockIdx.x * kStepThreads + threadIdx.x; c < n; c += stride) {
    for (int d = 0; d < D; ++d) {                                                          // :404-410
      const long long i = (long long)d * n + c;
      const R mu = a.mean[i], xd = a.x[i];
      delta[d][lane] = xd - mu;
      a.mean[i] = mu * a.keep + xd * a.inv_i;
    }
    for (int k = 0; k < nobs; ++k) {                                                       // :458-463, :412-414
      R o;
      if (k < nr) o = N_::abs_(a.x[(long long)k * n + c]);
      else if (k < nr + nc) {
        const R re = a.x[(long long)k * n + c], im = a.x[(long long)(k + nc) * n + c];
        o = N_::sqrt_(re * re + im * im);
      } else {
        const R v = a.x[(long long)(k - nr - nc) * n + c];
        o = v * v;
      }
      const long long i = (long long)k * n + c;
      a.obs_mean[i] = a.obs_mean[i] * a.keep + o * a.inv_i;
    }
    if (!a.update_cov) continue;
    // :418, :425 -- each block uses its own group's width; they coincide unless group steps made them differ
    R w_real = a.width[c], w_cplx = w_real;
    if (mixed && a.split_widths) {
      w_real = a.width[(long long)GROUP_REAL * n + c];
      w_cplx = a.width[(long long)GROUP_COMPLEX * n + c];
    }
    const R w2_real = w_real * w_real, w2_cplx = w_cplx * w_cplx;
    R *q = a.cov + (c >> 6) * P * 64 + (c & 63);
    constexpr long long ts = 64;
    for (int i = 0; i < nr; ++i) {
      const R di = delta[i][lane];
      int j = 0;
      for (; j + 16 <= i; j += 16) {       // all loads first: a store to q[.] would fence the next load
        R v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = q[u * ts];
#pragma unroll
        for (int u = 0; u < 16; ++u) q[u * ts] = fma_(di * delta[j + u][lane], a.inv_i, v[u] * a.cov_keep);
        q += 16 * ts;
      }
      for (; j < i; ++j) {
        *q = fma_(di * delta[j][lane], a.inv_i, *q * a.cov_keep);
        q += ts;
      }
      *q = fma_(w2_real, a.inv_i, fma_(di * di, a.inv_i, *q * a.cov_keep));
      q += ts;
    }
    for (int i = 0; i < nc; ++i) {         // the Hermitian block: (Re, Im) of the columns j < i, then the real diagonal
      const R ai = delta[nr + i][lane], bi = delta[nr + nc + i][lane];
      for (int j = 0; j < i; ++j) {
        const R aj = delta[nr + j][lane], bj = delta[nr + nc + j][lane];
        const R re = q[0], im = q[ts];
        q[0] = fma_(fma_(ai, aj, bi * bj), a.inv_i, re * a.cov_keep);
        q[ts] = fma_(fma_(bi, aj, -(ai * bj)), a.inv_i, im * a.cov_keep);
        q += 2 * ts;
      }
      *q = fma_(w2_cplx, a.inv_i, fma_(fma_(ai, ai, bi * bi), a.inv_i, *q * a.cov_keep));
      q += ts;
    }
  }
}

// factor = chol(C) per chain at runtime dimensions (pure real spaces): k_factor_stream of me_device.h with the row block in
// dynamic LDS (rows[ROWS][nr][64]) -- Cholesky-Banachiewicz, each lane its own chain, finished rows re-read from the
// factor field itself, ROWS rows built together so that every finished L_jk that is loaded serves ROWS dot products.
template <typename R, int ROWS>
__global__ void __launch_bounds__(kStepThreads) k_factor_runtime(const R *cov, R *factor, unsigned int *status, long long n, int nr,
                                                                 long long P) {      // P: packed entries per chain (real + Hermitian block)
  using N_ = Num<R>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_rt[];
  R(*lds)[kStepThreads] = reinterpret_cast<R(*)[kStepThreads]>(smem_rt);
  auto row_of = [&](int r) { return lds + (size_t)r * nr; };          // rows[r][k][lane] = row_of(r)[k][lane]
  bool bad_pivot = false;
  const long long stride = (long long)gridDim.x * kStepThreads;
  const int lane = threadIdx.x;
  for (long long c = (long long)blockIdx.x * kStepThreads + threadIdx.x; c < n; c += stride) {
    const long long base = (c >> 6) * P * 64 + (c & 63);
    const R *cv = cov + base;
    R *fc = factor + base;
    for (int i0 = 0; i0 < nr; i0 += ROWS) {
      const int nrows = nr - i0 < ROWS ? nr - i0 : ROWS;
      for (int r = 0; r < nrows; ++r) {                    // the covariance rows of the block into LDS
        const R *src = cv + (long long)tri(i0 + r, 0) * 64;
        int j = 0;
        for (; j + 16 <= i0 + r + 1; j += 16) {
          R v[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) v[u] = src[(j + u) * 64];
#pragma unroll
          for (int u = 0; u < 16; ++u) row_of(r)[j + u][lane] = v[u];
        }
        for (; j <= i0 + r; ++j) row_of(r)[j][lane] = src[j * 64];
      }
      // Columns left of the block: finished row j serves all rows of the block.  The walk is bound by memory LATENCY (one
      // wavefront per SIMD at most, every batch of loads a round trip), and column j needs column j - 1 of the block's rows:
      // FOUR finished rows are therefore fetched together -- their first j entries in batches of 4 x 16 loads, then the ten
      // entries of the little triangle between them -- and the four columns are finished one after the other from registers.
      int j = 0;
      for (; j + 4 <= i0; j += 4) {
        const R *lj[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) lj[q] = fc + (long long)tri(j + q, 0) * 64;
        R sum[4][ROWS];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int r = 0; r < ROWS; ++r) sum[q][r] = R(0);
        int k = 0;
        for (; k + 16 <= j; k += 16) {
          R f[4][16];
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int u = 0; u < 16; ++u) f[q][u] = lj[q][(k + u) * 64];
#pragma unroll
          for (int u = 0; u < 16; ++u) {
            R v[ROWS];
#pragma unroll
            for (int r = 0; r < ROWS; ++r) v[r] = r < nrows ? row_of(r)[k + u][lane] : R(0);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
              for (int r = 0; r < ROWS; ++r) sum[q][r] += v[r] * f[q][u];
          }
        }
        for (; k < j; k += 4) {                             // (j is a multiple of 4)
          R f[4][4];
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int u = 0; u < 4; ++u) f[q][u] = lj[q][(k + u) * 64];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            R v[ROWS];
#pragma unroll
            for (int r = 0; r < ROWS; ++r) v[r] = r < nrows ? row_of(r)[k + u][lane] : R(0);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
              for (int r = 0; r < ROWS; ++r) sum[q][r] += v[r] * f[q][u];
          }
        }
        R t[4][4];                                          // L[j + q][j + p], p <= q
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int p = 0; p <= q; ++p) t[q][p] = lj[q][(j + p) * 64];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const R inv = R(1) / t[q][q];
#pragma unroll
          for (int r = 0; r < ROWS; ++r)
            if (r < nrows) {
              R acc = sum[q][r];
#pragma unroll
              for (int p = 0; p < q; ++p) acc += row_of(r)[j + p][lane] * t[q][p];
              row_of(r)[j + q][lane] = (row_of(r)[j + q][lane] - acc) * inv;
            }
        }
      }
      for (; j < i0; ++j) {                                 // (at most three rows left)
        const R *lj = fc + (long long)tri(j, 0) * 64;
        R sum[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) sum[r] = R(0);
        int k = 0;
        for (; k + 16 <= j; k += 16) {
          R f[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) f[u] = lj[(k + u) * 64];
#pragma unroll
          for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int r = 0; r < ROWS; ++r)
              if (r < nrows) sum[r] += row_of(r)[k + u][lane] * f[u];
        }
        for (; k < j; ++k) {
          const R f = lj[k * 64];
#pragma unroll
          for (int r = 0; r < ROWS; ++r)
            if (r < nrows) sum[r] += row_of(r)[k][lane] * f;
        }
        const R inv = R(1) / lj[j * 64];
#pragma unroll
        for (int r = 0; r < ROWS; ++r)
          if (r < nrows) row_of(r)[j][lane] = (row_of(r)[j][lane] - sum[r]) * inv;
      }
      for (int r = 0; r < nrows; ++r) {                    // the triangle inside the block: everything is in LDS
        const int i = i0 + r;
        for (int j = i0; j < i; ++j) {
          const int rj = j - i0;
          R t = R(0);
          for (int k = 0; k < j; ++k) t += row_of(r)[k][lane] * row_of(rj)[k][lane];
          row_of(r)[j][lane] = (row_of(r)[j][lane] - t) / row_of(rj)[j][lane];
        }
        R t = row_of(r)[i][lane];
        for (int k = 0; k < i; ++k) t -= row_of(r)[k][lane] * row_of(r)[k][lane];
        if (!(t > R(0))) { bad_pivot = true; t = R(1e-30); }
        row_of(r)[i][lane] = N_::sqrt_(t);
      }
      for (int r = 0; r < nrows; ++r) {                    // finished rows out (re-read by this same lane: program order suffices)
        R *dst = fc + (long long)tri(i0 + r, 0) * 64;
        for (int j = 0; j <= i0 + r; ++j) dst[j * 64] = row_of(r)[j][lane];
      }
    }
  }
  if (bad_pivot) atomicOr(status, (unsigned int)ST_BAD_PIVOT);
}

// The Hermitian block of the factor at runtime dimensions: L = chol(conj K) (quirk Q3, metropolis_engine.py:292-298), the
// complex half of k_factor_mixed (me_device.h) with runtime sizes -- row by row, each lane its own chain, every operand
// through global memory (a finished L_ik is re-read from the factor field the lane itself wrote).  Written for
// correctness, not speed.
template <typename R>
__global__ void __launch_bounds__(kStepThreads) k_factor_runtime_complex(const R *cov, R *factor, unsigned int *status, long long n, int nr,
                                                                         int nc) {
  using N_ = Num<R>;
  const long long PR = (long long)nr * (nr + 1) / 2, P = PR + (long long)nc * nc;
  bool bad_pivot = false;
  const long long stride = (long long)gridDim.x * kStepThreads;
  for (long long c = (long long)blockIdx.x * kStepThreads + threadIdx.x; c < n; c += stride) {
    const long long base = (c >> 6) * P * 64 + (c & 63);
    const R *cv = cov + base;
    R *fc = factor + base;
    auto re_at = [&](int i, int j) { return (PR + (long long)i * i + 2 * j) * 64; };      // cre; cim = + 64; cdiag = re_at(i, i)
    for (int i = 0; i < nc; ++i)
      for (int j = 0; j <= i; ++j) {
        R sr = cv[re_at(i, j)];
        R si = j < i ? -cv[re_at(i, j) + 64] : R(0);                // conj(K)
        for (int k = 0; k < j; ++k) {                                // s -= L_ik conj(L_jk)
          const R ar = fc[re_at(i, k)], ai = fc[re_at(i, k) + 64];
          const R br = fc[re_at(j, k)], bi = fc[re_at(j, k) + 64];
          sr = fma_(-ai, bi, fma_(-ar, br, sr));
          si = fma_(ar, bi, fma_(-ai, br, si));
        }
        if (j < i) {
          const R d = fc[re_at(j, j)];
          fc[re_at(i, j)] = sr / d;
          fc[re_at(i, j) + 64] = si / d;
        } else {
          if (!(sr > R(0))) { bad_pivot = true; sr = R(1e-30); }
          fc[re_at(i, i)] = N_::sqrt_(sr);
        }
      }
  }
  if (bad_pivot) atomicOr(status, (unsigned int)ST_BAD_PIVOT);
}

template <typename R>
__global__ void __launch_bounds__(kBlockThreads) k_init_energy_runtime(const R *x, R *energy, long long n, unsigned int *status,
                                                                        RuntimeStep<R> p) {
  const int D = p.nr + 2 * p.nc;
  const long long stride = (long long)gridDim.x * kBlockThreads;
  for (long long c = (long long)blockIdx.x * kBlockThreads + threadIdx.x; c < n; c += stride) {
    R e = R(0);
    for (int d = 0; d < D; ++d) {
      const R v = x[(long long)d * n + c];
      e += weight_of(p, d) * v * v;
    }
    energy[c] = e;
    if (!Num<R>::finite(e)) atomicOr(status, (unsigned int)ST_NONFINITE_ENERGY);
  }
}

// Everything that needs the whole proposal (dense energy) or all normals (shared factor): see the header comment.
// `folded`: D(D+1)/2 values T_ij (i >= j) for ME_ENERGY_DENSE_QUAD; `factor`: the packed shared factor (pure real
// spaces: the lower triangle of L, row-major) for CK_SHARED.  Dynamic LDS: xp[D][64] (+ g[D][64] with a factor).
// value of lane `src` (a compile-time constant after unrolling) as a wave-uniform scalar
template <typename R>
__device__ __forceinline__ R from_lane(R v, int src) {
  if constexpr (sizeof(R) == 4) {
    return __builtin_bit_cast(R, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
  } else {
    const unsigned long long w = __builtin_bit_cast(unsigned long long, v);
    const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)w, src);
    const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(w >> 32), src);
    return __builtin_bit_cast(R, ((unsigned long long)hi << 32) | lo);
  }
}

// y_i = sum_{j <= i} T_ij v_j for a packed row-major lower triangle T in global memory (the same for every chain) and a
// vector v parked in LDS ([D][64], one column per lane), handed row by row to done(i, y_i, v_i).  Register-blocked
// 16 x 16: sixteen v_j are read from LDS once per block and serve sixteen rows.  A 16 x 16 tile of T is ONE coalesced
// vector load -- lane l holds T[i0 + l/4][j0 + 4 (l%4) .. +3] -- issued a block ahead of its use, and T_ij reaches the
// multiply-add as a scalar operand through v_readlane.  History (128 parameters x 2^17 chains, dense energy, float32 /
// float64 per sweep): one LDS read + one scalar load per multiply-add 816 / 2 049 us; 16 x 16 blocks with a row's T_ij
// as one contiguous scalar load 355 / 770 us (scalar-load latency exposed: ~100 SGPRs hold five rows in flight, one
// wavefront per SIMD has nothing else to run); tiles through vector loads + v_readlane: see DESIGN.md section 4.
template <typename R, class RowDone>
__device__ __forceinline__ void tri_rows_blocked(const R *__restrict__ tri_packed, R (*v)[kStepThreads], int D, int lane, RowDone &&done) {
  constexpr int B = 16;
  const int tr = lane >> 2, tc = (lane & 3) * 4;      // this lane's row and first column inside a tile
  // tile (i0, j0): full (every column exists in every row) left of the diagonal, triangular on it
  auto load_tile = [&](int i0, int j0, bool diagonal, R (&t)[4]) {
    int i = i0 + tr;
    if (i >= D) i = D - 1;                             // ragged last block: a valid address, values unused
    const R *src = tri_packed + (size_t)i * (size_t)(i + 1) / 2 + j0 + tc;
#pragma unroll
    for (int q = 0; q < 4; ++q) t[q] = (!diagonal || j0 + tc + q <= i) ? src[q] : R(0);
  };
  for (int i0 = 0; i0 < D; i0 += B) {
    R acc[B];
#pragma unroll
    for (int r = 0; r < B; ++r) acc[r] = R(0);
    R t[4];
    load_tile(i0, 0, i0 == 0, t);
    for (int j0 = 0; j0 < i0; j0 += B) {               // full blocks left of the diagonal
      R tn[4];
      load_tile(i0, j0 + B, j0 + B == i0, tn);         // the next tile of this block row (the last one: the diagonal tile)
      R vj[B];
#pragma unroll
      for (int u = 0; u < B; ++u) vj[u] = v[j0 + u][lane];
#pragma unroll
      for (int r = 0; r < B; ++r)
#pragma unroll
        for (int u = 0; u < B; ++u) acc[r] += from_lane(t[u & 3], r * 4 + (u >> 2)) * vj[u];
#pragma unroll
      for (int q = 0; q < 4; ++q) t[q] = tn[q];
    }
    R vd[B];                                           // the diagonal tile: row i0 + r has r + 1 entries (the rest loaded as 0)
#pragma unroll
    for (int u = 0; u < B; ++u) vd[u] = i0 + u < D ? v[i0 + u][lane] : R(0);
#pragma unroll
    for (int r = 0; r < B; ++r) {
#pragma unroll
      for (int u = 0; u <= r; ++u) acc[r] += from_lane(t[u & 3], r * 4 + (u >> 2)) * vd[u];
      if (i0 + r < D) done(i0 + r, acc[r], vd[r]);
    }
  }
}

// The same triangle product on the matrix cores: Y = T V for the 64 chains of a wavefront is a (D x D) x (D x 64) product.
// v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32: lane l holds A[row l & 15][k = l >> 4] -- T[i0 + (l & 15)][k0 + (l >> 4)],
// one gathered load from the packed triangle (zero above the diagonal and past row D) -- and B[k = l >> 4][column l & 15] --
// V[k0 + (l >> 4)][chain 16 cb + (l & 15)], one conflict-free LDS read per 16-chain block cb.  A row block of 16 rows
// costs 4 (ib + 1) k steps of 4 MFMAs; T reaches the arithmetic without any broadcast.  The results arrive with the chain
// on the lane's low four bits and four rows in the registers (float64: row (l >> 4) + 4 r, float32: row 4 (l >> 4) + r);
// done(row, cb, value) is called for each of them with (row, chain 16 cb + (l & 15)).
template <typename R>
struct MfmaTile;
template <>
struct MfmaTile<double> {
  using Acc = __attribute__((ext_vector_type(4))) double;
  static __device__ __forceinline__ Acc mma(double a, double b, Acc c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) + 4 * r; }
};
template <>
struct MfmaTile<float> {
  using Acc = __attribute__((ext_vector_type(4))) float;
  static __device__ __forceinline__ Acc mma(float a, float b, Acc c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ int row(int lane, int r) { return 4 * (lane >> 4) + r; }
};

template <typename R, class Done>
__device__ __forceinline__ void tri_rows_mfma(const R *__restrict__ tri_packed, R (*v)[kStepThreads], int D, int lane, Done &&done) {
  using M = MfmaTile<R>;
  const int j = lane & 15, h = lane >> 4;
  for (int i0 = 0; i0 < D; i0 += 16) {
    typename M::Acc acc[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) acc[cb] = typename M::Acc{R(0), R(0), R(0), R(0)};
    const int i = i0 + j;                                   // this lane's row of T
    const R *trow = tri_packed + (size_t)(i < D ? i : D - 1) * (size_t)((i < D ? i : D - 1) + 1) / 2;
    // Sixteen columns (four k steps) at a time.  Their operands are requested as a batch BEFORE the sixteen MFMAs: the A
    // gathers one block ahead, the sixteen LDS reads of B in one go (left to the compiler, every pair of MFMAs waited for its
    // own ds_read: the matrix cores idled through an LDS latency every 64 cycles; PMC: 56 % of the wavefront in s_waitcnt)
    auto load_a = [&](int k0, R (&a4)[4]) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int k = k0 + 4 * q + h;
        a4[q] = (k <= i && i < D) ? trow[k] : R(0);
      }
    };
    R a_now[4], a_next[4];
    load_a(0, a_now);
    for (int k0 = 0; k0 < i0 + 16; k0 += 16) {
      load_a(k0 + 16, a_next);                              // (columns past the row are zero without a load)
      R b[4][4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int k = k0 + 4 * q + h;
        if (k >= D) k = D - 1;                              // past the triangle A is zero: any finite V will do
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) b[q][cb] = v[k][16 * cb + j];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) acc[cb] = M::mma(a_now[q], b[q][cb], acc[cb]);
#pragma unroll
      for (int q = 0; q < 4; ++q) a_now[q] = a_next[q];
    }
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = i0 + M::row(lane, r);
        if (row < D) done(row, cb, acc[cb][r]);
      }
  }
}

// LDS is the exchange medium between the lanes of ONE wavefront here (lane l parks column l, the matrix-core operands
// are read across columns, tri_rows_mfma's callback writes other lanes' columns): every such hand-over is fenced with
// __builtin_amdgcn_wave_barrier() + an LDS fence (lds_handover) so that it does not rest on the compiler being unable to
// reorder the accesses.
static_assert(kStepThreads == 64, "k_step_runtime_lds: one wavefront per block (cross-lane LDS traffic is fenced per wavefront)");
__device__ __forceinline__ void lds_handover() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <typename R, int CK>
__global__ void __launch_bounds__(kStepThreads) k_step_runtime_lds(StepArgs<R> a, RuntimeStep<R> p, const R *__restrict__ folded) {
  constexpr bool SHARED = CK == CK_SHARED;
  using N_ = Num<R>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_rt[];
  const int nr = p.nr, nc = p.nc, D = nr + 2 * nc;
  R(*xp)[kStepThreads] = reinterpret_cast<R(*)[kStepThreads]>(smem_rt);
  R(*gs)[kStepThreads] = xp + D;                                   // SHARED only
  N_::prepare();
  const int NW = 2 * ((D + 1) / 2);
  const bool mixed = nr > 0 && nc > 0;
  const int d0 = p.group == GROUP_COMPLEX ? nr : 0, d1 = p.group == GROUP_REAL ? nr : D;
  unsigned int wave_accepted = 0;
  bool bad_energy = false, bad_width = false;
  const long long stride = (long long)gridDim.x * kStepThreads, n = a.n;
  const int lane = threadIdx.x;
  // every lane of a wavefront stays in the loop (the triangle products hand T around with v_readlane: a tile entry may
  // sit in any lane): lanes past the last chain shadow chain n - 1 and write nothing
  for (long long c0 = (long long)blockIdx.x * kStepThreads; c0 < n; c0 += stride) {
    const bool valid = c0 + lane < n;
    const long long c = valid ? c0 + lane : n - 1;
    R *xc = a.x + c;
    R e = a.energy[c];
    const int wrow = mixed ? p.group : 0;
    R w = a.width[(long long)wrow * n + c];
    R w_r = w, w_c = w;
    if (mixed && p.group == GROUP_ALL && a.split_widths) {
      w_r = a.width[(long long)GROUP_REAL * n + c];
      w_c = a.width[(long long)GROUP_COMPLEX * n + c];
    }
    const unsigned long long gid = a.chain_offset + (unsigned long long)c;
    for (int s = 0; s < a.n_sweeps; ++s) {
      const unsigned long long step = a.step_index + (unsigned long long)s;
      auto block_of = [&](int b) {
        U4 ctr;
        ctr.x = (uint32_t)gid;
        ctr.y = (uint32_t)(gid >> 32);
        ctr.z = (uint32_t)step;
        ctr.w = ((uint32_t)(step >> 32) << 16) | (uint32_t)b;
        return philox4x32_10(ctr, a.seed_lo, a.seed_hi);
      };
      const R s_r = w_r, s_c = w_c * R(0.70710678118654752440);
      // ---- the state into x' first, sixteen loads in flight at a time (issued one by one where they are used, every
      // load would expose the whole memory latency: one wavefront per SIMD has nothing else to run)
      constexpr int kStateBatch = 16;      // (64 at a time changed nothing)
      for (int d0b = 0; d0b < D; d0b += kStateBatch) {
        R xd[kStateBatch];
#pragma unroll
        for (int t = 0; t < kStateBatch; ++t) xd[t] = d0b + t < D ? xc[(long long)(d0b + t) * n] : R(0);
#pragma unroll
        for (int t = 0; t < kStateBatch; ++t)
          if (d0b + t < D) xp[d0b + t][lane] = xd[t];
      }
      // ---- the normals: straight into x' (identity shape), parked (shared factor), or spread over x' through the
      // chain's own factor
      for (int b = 0; 4 * b < NW; ++b) {
        const U4 o = block_of(b);
        R g[4];
        N_::normal_pair(o.x, o.y, g[0], g[1]);
        N_::normal_pair(o.z, o.w, g[2], g[3]);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int d = 4 * b + t;
          if (d < D) {
            if constexpr (SHARED) gs[d][lane] = g[t];
            else if constexpr (CK == CK_PER_CHAIN) {
              // real columns: below, a whole Philox block at a time.  A complex normal -- Re or Im part of zeta_j -- adds
              // (w_c / sqrt 2) L_ij zeta_j to z'_i for the rows i >= j of the Hermitian block's factor L = chol(conj K)
              // (:274-302; packed as (Re, Im) pairs of a row, then its real diagonal): column by column as well
              if (d >= nr && d >= d0 && d < d1) {
                const bool im_part = d >= nr + nc;
                const int j = d - nr - (im_part ? nc : 0);
                const long long pr = (long long)nr * (nr + 1) / 2;
                const R *fac = a.factor + (c >> 6) * (pr + (long long)nc * nc) * 64 + (c & 63);
                const R sc = s_c * g[t];
                R(*zre)[kStepThreads] = xp + nr, (*zim)[kStepThreads] = xp + nr + nc;
                // the diagonal entry is real: zeta's part goes to the same part of z'_j
                (im_part ? zim : zre)[j][lane] = fma_(sc, fac[(pr + (long long)j * j + 2 * j) * 64], (im_part ? zim : zre)[j][lane]);
                int i = j + 1;
                for (; i + 8 <= nc; i += 8) {
                  R lr[8], li[8];
#pragma unroll
                  for (int u = 0; u < 8; ++u) {
                    const R *e2 = fac + (pr + (long long)(i + u) * (i + u) + 2 * j) * 64;
                    lr[u] = e2[0];
                    li[u] = e2[64];
                  }
#pragma unroll
                  for (int u = 0; u < 8; ++u) {
                    // (lr + i li) (sc) for a Re part, (lr + i li) (i sc) = -li sc + i lr sc for an Im part
                    zre[i + u][lane] = fma_(im_part ? -sc : sc, im_part ? li[u] : lr[u], zre[i + u][lane]);
                    zim[i + u][lane] = fma_(sc, im_part ? lr[u] : li[u], zim[i + u][lane]);
                  }
                }
                for (; i < nc; ++i) {
                  const R *e2 = fac + (pr + (long long)i * i + 2 * j) * 64;
                  const R lr = e2[0], li = e2[64];
                  zre[i][lane] = fma_(im_part ? -sc : sc, im_part ? li : lr, zre[i][lane]);
                  zim[i][lane] = fma_(sc, im_part ? lr : li, zim[i][lane]);
                }
              }
            } else if (d >= d0 && d < d1) xp[d][lane] = fma_(d < nr ? s_r : s_c, g[t], xp[d][lane]);
          }
        }
        if constexpr (CK == CK_PER_CHAIN) {
          // x' = x + w L g by COLUMNS for the real block, four at a time: the normals of this Philox block add
          // w (g_0 L_i,4b + ... + g_3 L_i,4b+3) to every row i >= 4 b.  Only x' has to be parked that way (row by row, all of g
          // would be needed as well); the four entries of a row are neighbours in the packed order, i.e. 4 x 64 values of the
          // wavefront's chains in one 2 KiB run of the tile-major factor field; sixteen loads (four rows) are issued together.
          // (Rows i < nr only have columns <= i: a block that straddles the end of the real parameters needs no special case.)
          const int db = 4 * b;
          if (db < nr && d0 == 0) {
          R sg[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) sg[t] = db + t < nr ? s_r * g[t] : R(0);
          const R *fac = a.factor + (c >> 6) * ((long long)nr * (nr + 1) / 2 + (long long)nc * nc) * 64 + (c & 63);
          for (int i = db; i < db + 4 && i < nr; ++i) {       // the triangle on the diagonal: row i has columns 4 b .. i
            const R *row = fac + (long long)tri(i, db) * 64;
            R acc = xp[i][lane];
            for (int t = 0; t <= i - db; ++t) acc = fma_(sg[t], row[t * 64], acc);
            xp[i][lane] = acc;
          }
          int i = db + 4;
          for (; i + 4 <= nr; i += 4) {
            R f[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const R *row = fac + (long long)tri(i + u, db) * 64;
#pragma unroll
              for (int t = 0; t < 4; ++t) f[u][t] = row[t * 64];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              R acc = xp[i + u][lane];
#pragma unroll
              for (int t = 0; t < 4; ++t) acc = fma_(sg[t], f[u][t], acc);
              xp[i + u][lane] = acc;
            }
          }
          for (; i < nr; ++i) {
            const R *row = fac + (long long)tri(i, db) * 64;
            R acc = xp[i][lane];
#pragma unroll
            for (int t = 0; t < 4; ++t) acc = fma_(sg[t], row[t * 64], acc);
            xp[i][lane] = acc;
          }
          }
        }
      }
      lds_handover();                                       // x (and g) are parked: other lanes read them as MFMA operands
      if constexpr (SHARED) {
        // x' = x + w L g, row by row; L packed lower triangle (pure real spaces: nc == 0)
        if (ME_RUNTIME_MFMA) {
          R s_col[4];                                       // the widths of the chains this lane holds results for
#pragma unroll
          for (int cb = 0; cb < 4; ++cb) s_col[cb] = __shfl(s_r, 16 * cb + (lane & 15));
          tri_rows_mfma(a.factor, gs, D, lane, [&](int row, int cb, R y) {
            const int col = 16 * cb + (lane & 15);
            xp[row][col] = fma_(s_col[cb], y, xp[row][col]);      // every (row, chain) is touched by exactly one lane
          });
        } else {
          tri_rows_blocked(a.factor, gs, D, lane, [&](int i, R acc, R) { xp[i][lane] = fma_(s_r, acc, xp[i][lane]); });
        }
        lds_handover();                                     // x' was written across lanes: visible before anyone reads it
      }
      // ---- energy of the proposal
      R e_new = R(0);
      if (p.energy_kind == ME_ENERGY_DENSE_QUAD) {
        if (ME_RUNTIME_MFMA) {
          // E = sum_i x'_i y_i: a lane adds up the rows it holds for its chain of each 16-chain block, the four lane
          // groups are added, and lane l keeps the block of its own chain, cb = l >> 4
          R part[4] = {R(0), R(0), R(0), R(0)};
          tri_rows_mfma(folded, xp, D, lane, [&](int row, int cb, R y) { part[cb] = fma_(xp[row][16 * cb + (lane & 15)], y, part[cb]); });
#pragma unroll
          for (int cb = 0; cb < 4; ++cb) {
            part[cb] += __shfl_xor(part[cb], 16);
            part[cb] += __shfl_xor(part[cb], 32);
          }
          const int own = lane >> 4;
          e_new = own == 0 ? part[0] : own == 1 ? part[1] : own == 2 ? part[2] : part[3];
        } else {
          tri_rows_blocked(folded, xp, D, lane, [&](int, R y, R xi) { e_new += xi * y; });
        }
      } else {
        for (int d = 0; d < D; ++d) e_new += weight_of(p, d) * xp[d][lane] * xp[d][lane];
      }
      const U4 ow = block_of(NW >> 2);
      const R u = N_::unit((NW & 3) == 0 ? ow.x : ow.z);
      bool rejected = false;
      if (a.reject_kind == ME_REJECT_ABS_REAL0_GE) rejected = !(N_::abs_(xp[0][lane]) < a.reject_bound);
      const R diff = e_new - e;
      bool accept = diff <= R(0);
      if (a.temp > R(0)) accept = accept || N_::uphill(u, diff, a.inv_temp, a.inv_temp_log2e);
      accept = accept && !rejected && valid;
      bad_energy |= (valid && !rejected && !N_::finite(e_new));
      if (accept)
        for (int d = d0; d < d1; ++d) xc[(long long)d * n] = xp[d][lane];
      e = accept ? e_new : e;
      w = N_::adapt(w, accept, a.ratio, a.p, a.damping, a.up, a.down);
      if (p.group == GROUP_ALL) w_r = w_c = w;
      else if (p.group == GROUP_REAL) w_r = w;
      else w_c = w;
      wave_accepted += (unsigned int)__popcll(__ballot(accept));
      lds_handover();                                       // the next sweep re-parks x over what the energy product read
    }
    bad_width |= valid && !(w > R(0));
    if (valid) {
      a.energy[c] = e;
      a.width[(long long)wrow * n + c] = w;
    }
  }
  if ((threadIdx.x & 63) == 0 && wave_accepted) {
    unsigned long long *slot = a.accept_slots + (size_t)blockIdx.x * (kStepThreads >> 6) + (threadIdx.x >> 6);
    *slot += (unsigned long long)wave_accepted;
  }
  const unsigned int bits = (bad_energy ? ST_NONFINITE_ENERGY : 0u) | (bad_width ? ST_BAD_WIDTH : 0u);
  if (bits) atomicOr(a.status, bits);
}

template <typename R>
__global__ void __launch_bounds__(kBlockThreads) k_init_energy_runtime_dense(const R *x, R *energy, long long n, unsigned int *status,
                                                                              int D, const R *__restrict__ folded) {
  const long long stride = (long long)gridDim.x * kBlockThreads;
  for (long long c = (long long)blockIdx.x * kBlockThreads + threadIdx.x; c < n; c += stride) {
    R e = R(0);
    const R *trow = folded;
    for (int i = 0; i < D; ++i) {
      R y = R(0);
      for (int j = 0; j <= i; ++j) y += trow[j] * x[(long long)j * n + c];
      e += x[(long long)i * n + c] * y;
      trow += i + 1;
    }
    energy[c] = e;
    if (!Num<R>::finite(e)) atomicOr(status, (unsigned int)ST_NONFINITE_ENERGY);
  }
}

bool has_energy(int kind) { return kind == ME_ENERGY_ISO_QUAD || kind == ME_ENERGY_DIAG_QUAD || kind == ME_ENERGY_DENSE_QUAD; }
int energy_terms(int kind) { return has_energy(kind) ? 1 : 0; }

template <typename R>
bool fill(RuntimeStep<R> &p, int nr, int nc, int group, int kind, const double *coef, int n_coef, const void *coef_device) {
  p.nr = nr;
  p.nc = nc;
  p.group = group;
  p.energy_kind = kind;
  p.iso = R(0);
  p.weights = nullptr;
  if (kind == ME_ENERGY_ISO_QUAD) {
    if (n_coef != 1) return false;
    p.iso = (R)coef[0];
    return true;
  }
  if (kind == ME_ENERGY_DIAG_QUAD) {
    if (n_coef != nr + nc || !coef_device) return false;
    p.weights = (const R *)coef_device;        // expanded to D entries by me_create (me_api.hip)
    return true;
  }
  if (kind == ME_ENERGY_DENSE_QUAD) {
    const int d = nr + 2 * nc;
    return n_coef == d * d && coef_device != nullptr;    // coef_device = the FOLDED triangle, made by me_create
  }
  return false;
}

int grid_of(long long n, int requested) {
  long long blocks = (n + kBlockThreads - 1) / kBlockThreads;
  if (requested > 0 && blocks > requested) blocks = requested;
  return (int)(blocks < 1 ? 1 : blocks);
}

template <typename R>
hipError_t step(const StepLaunch &l, hipStream_t stream) {
  if (l.inj_normals) return hipErrorNotSupported;
  if (l.cov_kind != CK_IDENTITY && l.cov_kind != CK_PER_CHAIN && !(l.cov_kind == CK_SHARED && l.n_complex == 0)) return hipErrorNotSupported;
  const bool mixed = l.n_real > 0 && l.n_complex > 0;
  if (l.group != GROUP_ALL && !mixed) return hipErrorInvalidValue;
  RuntimeStep<R> p;
  if (!fill(p, l.n_real, l.n_complex, l.group, l.energy_kind, l.coef_host, l.n_coef, l.coef_device)) return hipErrorInvalidValue;
  StepArgs<R> a{};
  a.x = (R *)l.x;
  a.energy = (R *)l.energy;
  a.width = (R *)l.width;
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
  a.reject_bound = (R)l.reject_bound;
  a.temp = (R)l.temp;
  a.inv_temp = l.temp > 0 ? (R)(1.0 / l.temp) : (R)0;
  a.inv_temp_log2e = l.temp > 0 ? (R)(1.4426950408889634 / l.temp) : (R)0;
  a.ratio = (R)l.ratio;
  a.p = (R)l.target_acceptance;
  a.damping = (R)l.damping;
  a.up = (R)(l.ratio * (1.0 - l.target_acceptance) / l.damping);
  a.down = (R)(-l.ratio * l.target_acceptance / l.damping);
  if (l.energy_kind == ME_ENERGY_DENSE_QUAD || l.cov_kind != CK_IDENTITY) {
    // the LDS form: one wavefront per block, x' (and, with a shared factor, g) parked per lane
    const bool shared = l.cov_kind == CK_SHARED;
    const size_t lds = (size_t)(l.n_real + 2 * l.n_complex) * kStepThreads * sizeof(R) * (shared ? 2 : 1);
    if (lds > kRuntimeLdsLimit) return hipErrorNotSupported;
    a.factor = (const R *)l.factor;
    long long blocks = (l.n + kStepThreads - 1) / kStepThreads;
    if (l.grid_blocks > 0 && blocks > l.grid_blocks) blocks = l.grid_blocks;
    const R *folded = l.energy_kind == ME_ENERGY_DENSE_QUAD ? (const R *)l.coef_device : nullptr;
    int device = 0;
    if (hipError_t rc = hipGetDevice(&device); rc != hipSuccess) return rc;
    auto launch = [&](auto ck) -> hipError_t {
      constexpr int CK = decltype(ck)::value;
      static PerDevice<hipError_t> attr_cache;
      const hipError_t rc = attr_cache.get(device, [] {
        return hipFuncSetAttribute((const void *)k_step_runtime_lds<R, CK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRuntimeLdsLimit);
      });
      if (rc != hipSuccess) return rc;
      hipLaunchKernelGGL((k_step_runtime_lds<R, CK>), dim3((unsigned)blocks), dim3(kStepThreads), lds, stream, a, p, folded);
      return hipGetLastError();
    };
    if (shared) return launch(std::integral_constant<int, CK_SHARED>{});
    if (l.cov_kind == CK_PER_CHAIN) return launch(std::integral_constant<int, CK_PER_CHAIN>{});
    return launch(std::integral_constant<int, CK_IDENTITY>{});
  }
  // the acceptance slots are sized for 64-thread blocks over all chains; a 256-thread grid uses a quarter of them
  hipLaunchKernelGGL(k_step_runtime<R>, dim3(grid_of(l.n, l.grid_blocks)), dim3(kBlockThreads), 0, stream, a, p);
  return hipGetLastError();
}

template <typename R>
hipError_t measure(const MeasureLaunch &l, hipStream_t stream) {
  MeasureArgs<R> a{};
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
  if (!l.cov) {
    hipLaunchKernelGGL(k_measure_runtime<R>, dim3(grid_of(l.n, l.grid_blocks)), dim3(kBlockThreads), 0, stream, a, l.n_real,
                       l.n_complex);
    return hipGetLastError();
  }
  // engines that keep per-chain covariance matrices: one wavefront per block, delta (and the factor kernel's row block) in LDS
  const size_t row_bytes = (size_t)(l.n_real + 2 * l.n_complex) * kStepThreads * sizeof(R);
  if (row_bytes > kRuntimeLdsLimit) return hipErrorNotSupported;
  int device = 0;
  if (hipError_t rc = hipGetDevice(&device); rc != hipSuccess) return rc;
  static PerDevice<hipError_t> attr_measure;
  hipError_t rc = attr_measure.get(device, [] {
    return hipFuncSetAttribute((const void *)k_measure_runtime_cov<R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRuntimeLdsLimit);
  });
  if (rc != hipSuccess) return rc;
  const dim3 grid(grid_for(l.n, l.grid_blocks)), block(kStepThreads);
  hipLaunchKernelGGL(k_measure_runtime_cov<R>, grid, block, row_bytes, stream, a, l.n_real, l.n_complex);
  if (l.update_cov && l.write_factor) {
    const long long p_total = (long long)l.n_real * (l.n_real + 1) / 2 + (long long)l.n_complex * l.n_complex;
    const size_t real_row_bytes = (size_t)l.n_real * kStepThreads * sizeof(R);
    auto launch = [&](auto rows_tag) -> hipError_t {
      constexpr int ROWS = decltype(rows_tag)::value;
      static PerDevice<hipError_t> attr_factor;
      const hipError_t err = attr_factor.get(device, [] {
        return hipFuncSetAttribute((const void *)k_factor_runtime<R, ROWS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRuntimeLdsLimit);
      });
      if (err != hipSuccess) return err;
      hipLaunchKernelGGL((k_factor_runtime<R, ROWS>), grid, block, ROWS * real_row_bytes, stream, (const R *)l.cov, (R *)l.factor, l.status,
                         l.n, l.n_real, p_total);
      return hipGetLastError();
    };
    if (l.n_real > 0) {
      // as many rows together as the LDS admits: every finished row that is re-read serves all of them
      const hipError_t err = 4 * real_row_bytes <= kRuntimeLdsLimit   ? launch(std::integral_constant<int, 4>{})
                             : 2 * real_row_bytes <= kRuntimeLdsLimit ? launch(std::integral_constant<int, 2>{})
                                                                      : launch(std::integral_constant<int, 1>{});
      if (err != hipSuccess) return err;
    }
    if (l.n_complex > 0)
      hipLaunchKernelGGL(k_factor_runtime_complex<R>, grid, block, 0, stream, (const R *)l.cov, (R *)l.factor, l.status, l.n, l.n_real,
                         l.n_complex);
  }
  return hipGetLastError();
}

template <typename R>
hipError_t init_energy(const EnergyLaunch &l, hipStream_t stream) {
  RuntimeStep<R> p;
  if (!fill(p, l.n_real, l.n_complex, GROUP_ALL, l.energy_kind, l.coef_host, l.n_coef, l.coef_device)) return hipErrorInvalidValue;
  if (l.energy_kind == ME_ENERGY_DENSE_QUAD) {
    hipLaunchKernelGGL(k_init_energy_runtime_dense<R>, dim3(grid_of(l.n, l.grid_blocks)), dim3(kBlockThreads), 0, stream,
                       (const R *)l.x, (R *)l.energy, l.n, l.status, l.n_real + 2 * l.n_complex, (const R *)l.coef_device);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(k_init_energy_runtime<R>, dim3(grid_of(l.n, l.grid_blocks)), dim3(kBlockThreads), 0, stream,
                     (const R *)l.x, (R *)l.energy, l.n, l.status, p);
  return hipGetLastError();
}

// n_real = n_complex = -1: the wildcard set find_kernel_set falls back to above kMaxRegisterDof (me_api.hip)
// per_chain_cov / tracks_cov / streams_packed: the packed per-chain matrices are kept for cov_mode="reference" (pure real
// spaces, me_create) or the tracking flag only, walked with 64-bit pointers, and may pass 4 GiB
const KernelSet kRuntimeF32 = {nullptr, ME_F32, -1, -1, true, true, has_energy, energy_terms, step<float>, nullptr,
                               measure<float>, init_energy<float>, 0, nullptr, nullptr, false, true, nullptr, false};
const KernelSet kRuntimeF64 = {nullptr, ME_F64, -1, -1, true, true, has_energy, energy_terms, step<double>, nullptr,
                               measure<double>, init_energy<double>, 0, nullptr, nullptr, false, true, nullptr, false};

struct Registrar {
  Registrar() {
    register_kernel_set(&kRuntimeF32);
    register_kernel_set(&kRuntimeF64);
  }
} registrar;

}  // namespace
}  // namespace me
