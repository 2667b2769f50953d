// Device code of the many-chain Metropolis engine for gfx950 (MI355X): one lane owns one chain, one wavefront a
// tile of 64 chains; the chain's state lives in registers for the whole launch.
//
// Reference semantics (file = /root/reference/metropolisengine/metropolis_engine.py):
//   k_step     step_all / step_real_group / step_complex_group :209-259, draw_*_group :261-302,
//              metropolis_decision :319-338, update_*_sigma :429-456
//   k_measure  measure* :342-383, update_*_mean :404-410, update_covariance_matrix_* :416-427,
//              construct_observables / update_observables_mean :458-463, :412-414
// The random streams are the counter-based Philox4x32-10 streams specified in oracle/philox.py.
#pragma once

#include <type_traits>
#include <utility>

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "me_internal.h"
#include "me_math64.h"

namespace me {

// ------------------------------------------------------------------------------------------------ Philox
struct U4 {
  uint32_t x, y, z, w;
};

// Philox4x32-10 (Salmon et al. 2011; the generator of hiprand's PHILOX4_32_10).  The key schedule is
// wave-uniform and lives in scalar registers; each round is two 32x32->64 multiplies (v_mad_u64_u32) and two
// three-input xors (gfx950's v_bitop3_b32 with truth table 0x96; hipcc emits two v_xor_b32 for `a ^ b ^ c`).
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
  return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
}
__device__ __forceinline__ U4 philox4x32_10(U4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c.x;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c.z;
    U4 n;
    n.x = xor3((uint32_t)(p1 >> 32), c.y, k0);
    n.y = (uint32_t)p1;
    n.z = xor3((uint32_t)(p0 >> 32), c.w, k1);
    n.w = (uint32_t)p0;
    c = n;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c;
}

// ------------------------------------------------------------------------------------------------ numerics
// One explicit fused multiply-add.  hipcc contracts a * b + c by itself (-ffp-contract=fast), but for a * b + c * d it may
// fuse either product, and which one depends on the code around the expression -- so the same source line can round
// differently in two kernels.  Wherever a result must be BITWISE the same in k_measure and k_cycle (both inline
// measure_chain) or in pass 1 and pass 2 of the runtime-dimension kernels, sums of two products are spelled with fma_.
__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }

template <typename R>
struct Num;

template <>
struct Num<float> {
  // u = (w + 0.5) 2^-32 in (0, 1]; Box-Muller with the hardware transcendentals: v_log_f32 is log2,
  // v_sin_f32 / v_cos_f32 take their argument in revolutions, which is exactly 2*pi*u2.
  static __device__ __forceinline__ void prepare() {}
  static __device__ __forceinline__ float unit(uint32_t w) {
    return __builtin_fmaf((float)w, 2.3283064365386963e-10f, 1.1641532182693481e-10f);
  }
  static __device__ __forceinline__ void normal_pair(uint32_t wa, uint32_t wb, float &g0, float &g1) {
    const float u1 = unit(wa), u2 = unit(wb);
    const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));  // sqrt(-2 ln u1); raw v_sqrt_f32 (1 ulp) -- the IEEE expansion costs ~10 VALU per draw
    g0 = r * __builtin_amdgcn_cosf(u2);
    g1 = r * __builtin_amdgcn_sinf(u2);
  }
  // accept an uphill move of size d > 0:  u <= exp(-d/T)
  static __device__ __forceinline__ bool uphill(float u, float d, float, float inv_temp_log2e) {
    return u <= __builtin_amdgcn_exp2f(-d * inv_temp_log2e);
  }
  static __device__ __forceinline__ float adapt(float w, bool acc, float ratio, float p, float damping,
                                                float up, float down) {
    return __builtin_fmaf(w, acc ? up : down, w);
  }
  static __device__ __forceinline__ float sqrt_(float v) { return __builtin_sqrtf(v); }
  static __device__ __forceinline__ float abs_(float v) { return __builtin_fabsf(v); }
  static __device__ __forceinline__ bool finite(float v) { return __builtin_isfinite(v); }
};

template <>
struct Num<double> {
  // The float64 Box-Muller / accept arithmetic is me_math64.h (range-specific log, sqrt, sin/cos, exp; <= 4 ulp): the
  // general-purpose device libm versions cost 2.5x the instructions and 180 VGPRs in k_step.  Its 97-row log table
  // (1.5 KiB) lives in LDS: the lookups are per-lane gathers, LDS serves them without touching the vector-memory
  // queue the state rows are arriving on.
  static __device__ __forceinline__ double (*log_table())[2] {
    __shared__ __attribute__((aligned(16))) double table[math64::kLogEntries][2];
    return table;
  }
  // once per block, by every thread, before the first normal_pair
  static __device__ __forceinline__ void prepare() {
    double(*table)[2] = log_table();
    // both rounds of a 64-thread workgroup's share in one batch of loads (a rolled copy loop waits for each load in turn)
    for (int k0 = threadIdx.x; k0 < math64::kLogEntries; k0 += 2 * (int)blockDim.x) {
      const int k1 = k0 + (int)blockDim.x;
      const bool second = k1 < math64::kLogEntries;
      const double a0 = math64::kLogTable[k0][0], a1 = math64::kLogTable[k0][1];
      const double b0 = second ? math64::kLogTable[k1][0] : 0.0, b1 = second ? math64::kLogTable[k1][1] : 0.0;
      table[k0][0] = a0;
      table[k0][1] = a1;
      if (second) {
        table[k1][0] = b0;
        table[k1][1] = b1;
      }
    }
    __syncthreads();
  }
  static __device__ __forceinline__ double unit(uint32_t w) { return ((double)w + 0.5) * (1.0 / 4294967296.0); }
  static __device__ __forceinline__ void normal_pair(uint32_t wa, uint32_t wb, double &g0, double &g1) {
    math64::normal_pair(wa, wb, log_table(), g0, g1);
  }
  static __device__ __forceinline__ bool uphill(double u, double d, double inv_temp, double) {
    return u <= math64::exp_nonpos(-d * inv_temp);
  }
  // literal order of metropolis_engine.py:431-435 so that trajectories track the float64 oracle
  static __device__ __forceinline__ double adapt(double w, bool acc, double ratio, double p, double damping,
                                                 double, double) {
    const double scale = w * ratio;
    return acc ? w + scale * (1.0 - p) / damping : w - scale * p / damping;
  }
  static __device__ __forceinline__ double sqrt_(double v) { return sqrt(v); }
  static __device__ __forceinline__ double abs_(double v) { return fabs(v); }
  static __device__ __forceinline__ bool finite(double v) { return isfinite(v); }
};

// ------------------------------------------------------------------------------------------------ field access
// A component-major field (rows x n chains) addressed through a buffer descriptor: the row offset travels in a
// scalar register (soffset = row * n * sizeof(R)) and the chain offset in ONE vector register, so a kernel that
// touches dozens of rows holds no per-row 64-bit addresses (they cost 2 VGPRs each and were being spilled).
// Out-of-range accesses are dropped by the hardware range check.  me_create guarantees rows*n*sizeof(R) < 4 GiB.
#ifndef ME_NT_AUX
#define ME_NT_AUX 2      // experiments: 0 = default cache policy for the packed fields too
#endif
template <typename R>
struct Field {
  __amdgpu_buffer_rsrc_t rsrc;
  unsigned int row_bytes;
  __device__ __forceinline__ Field(const R *base, long long n, int rows)
      : rsrc(__builtin_amdgcn_make_buffer_rsrc(const_cast<R *>(base), 0, (unsigned int)(rows * n * (long long)sizeof(R)),
                                               0x00020000)),
        row_bytes((unsigned int)(n * (long long)sizeof(R))) {}
  // chain_off = chain index * sizeof(R)
  __device__ __forceinline__ R load(int row, unsigned int chain_off) const {
    if constexpr (sizeof(R) == 4) {
      return __builtin_bit_cast(R, __builtin_amdgcn_raw_buffer_load_b32(rsrc, chain_off, (unsigned int)row * row_bytes, 0));
    } else {
      const auto v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, chain_off, (unsigned int)row * row_bytes, 0);
      return __builtin_bit_cast(R, v);
    }
  }
  // Non-temporal forms (cache-policy bit 1 of the buffer instruction, "nt") for the packed per-chain covariance /
  // factor fields: 0.5-2.3 GB that a launch touches exactly once.  Streamed with the default policy they push the
  // chain state (72-150 MB, otherwise resident in the 256 MiB Infinity Cache from launch to launch) out to HBM:
  // tools/dev/rows_probe3.hip, the pattern of k_step with per-chain factors at 16 parameters: 151.7 us default,
  // 118.4 us with nt on the factor rows (memory only, no arithmetic).
  __device__ __forceinline__ R load_nt(int row, unsigned int chain_off) const {
    if constexpr (sizeof(R) == 4) {
      return __builtin_bit_cast(R, __builtin_amdgcn_raw_buffer_load_b32(rsrc, chain_off, (unsigned int)row * row_bytes, ME_NT_AUX));
    } else {
      const auto v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, chain_off, (unsigned int)row * row_bytes, ME_NT_AUX);
      return __builtin_bit_cast(R, v);
    }
  }
  __device__ __forceinline__ void store_nt(int row, unsigned int chain_off, R value) const {
    if constexpr (sizeof(R) == 4) {
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, value), rsrc, chain_off,
                                            (unsigned int)row * row_bytes, ME_NT_AUX);
    } else {
      using v2 = decltype(__builtin_amdgcn_raw_buffer_load_b64(rsrc, 0u, 0u, 0));
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2, value), rsrc, chain_off,
                                            (unsigned int)row * row_bytes, ME_NT_AUX);
    }
  }
  __device__ __forceinline__ void store(int row, unsigned int chain_off, R value) const {
    if constexpr (sizeof(R) == 4) {
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, value), rsrc, chain_off,
                                            (unsigned int)row * row_bytes, 0);
    } else {
      using v2 = decltype(__builtin_amdgcn_raw_buffer_load_b64(rsrc, 0u, 0u, 0));
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2, value), rsrc, chain_off,
                                            (unsigned int)row * row_bytes, 0);
    }
  }
};

// The packed per-chain covariance / factor fields are TILE-major: [tile of 64 chains][P entries][64 lanes], i.e. entry k of
// chain c sits at ((c >> 6) * P + k) * 64 + (c & 63).  A wavefront's whole matrix is then ONE contiguous run of P x 256
// bytes (34 KiB at 16 real parameters in float32) instead of P rows that lie a whole field row (4 MiB at 2^20 chains)
// apart: memory-only probes of the access patterns (tools/dev/rows_probe*.hip): k_step with per-chain factors 118 -> 105 us,
// k_measure (16,0) 412 -> 388 us, and the spread between runs that comes from where 336 separate row streams land in
// physical memory goes away.  (For the STATE the component-major layout costs nothing: 18-66 rows)
// Same interface as Field; `tile_off` = tiled_offset<R>(c, P), entry offsets are compile-time constants after unrolling.
template <typename R>
__device__ __forceinline__ unsigned int tiled_offset(long long c, int entries) {
  return (unsigned int)(((c >> 6) * (long long)entries * 64 + (c & 63)) * (long long)sizeof(R));
}
template <typename R>
struct TiledField {
  __amdgpu_buffer_rsrc_t rsrc;
  __device__ __forceinline__ TiledField(const R *base, long long n, int entries)
      : rsrc(__builtin_amdgcn_make_buffer_rsrc(const_cast<R *>(base), 0,
                                               (unsigned int)(((n + 63) >> 6) * (long long)entries * 64 * (long long)sizeof(R)),
                                               0x00020000)) {}
  static constexpr unsigned int kEntryBytes = 64u * (unsigned int)sizeof(R);
  template <int AUX>
  __device__ __forceinline__ R load_aux(int k, unsigned int tile_off) const {
    if constexpr (sizeof(R) == 4) {
      return __builtin_bit_cast(R, __builtin_amdgcn_raw_buffer_load_b32(rsrc, tile_off, (unsigned int)k * kEntryBytes, AUX));
    } else {
      const auto v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, tile_off, (unsigned int)k * kEntryBytes, AUX);
      return __builtin_bit_cast(R, v);
    }
  }
  template <int AUX>
  __device__ __forceinline__ void store_aux(int k, unsigned int tile_off, R value) const {
    if constexpr (sizeof(R) == 4) {
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, value), rsrc, tile_off, (unsigned int)k * kEntryBytes, AUX);
    } else {
      using v2 = decltype(__builtin_amdgcn_raw_buffer_load_b64(rsrc, 0u, 0u, 0));
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2, value), rsrc, tile_off, (unsigned int)k * kEntryBytes, AUX);
    }
  }
  __device__ __forceinline__ R load(int k, unsigned int tile_off) const { return load_aux<0>(k, tile_off); }
  __device__ __forceinline__ R load_nt(int k, unsigned int tile_off) const { return load_aux<ME_NT_AUX>(k, tile_off); }
  __device__ __forceinline__ void store(int k, unsigned int tile_off, R value) const { store_aux<0>(k, tile_off, value); }
  __device__ __forceinline__ void store_nt(int k, unsigned int tile_off, R value) const { store_aux<ME_NT_AUX>(k, tile_off, value); }
};

// accesses of k_measure / k_factor to the packed covariance and factor fields; NT: streamed once per launch from / to
// HBM (the launcher decides by working-set size, me_kernels.hip)
template <bool NT, class F>
__device__ __forceinline__ auto packed_load(const F &f, int row, unsigned int chain_off) {
  if constexpr (NT) return f.load_nt(row, chain_off);
  else return f.load(row, chain_off);
}
template <bool NT, class F, typename R>
__device__ __forceinline__ void packed_store(const F &f, int row, unsigned int chain_off, R value) {
  if constexpr (NT) f.store_nt(row, chain_off, value);
  else f.store(row, chain_off, value);
}

// The chain state (x, energy ledger, width) as k_step sees it.  NTS: read and written non-temporally -- chosen by the
// launcher when the state alone is more than twice the Infinity Cache, so that next to nothing of it could stay resident
// from launch to launch anyway (float64, 16 parameters: 236 -> 227 us at 2^22 chains; at 2^20, where the state DOES stay resident, the
// same policy costs 50.6 -> 59.1 us).
template <typename R, bool NTS>
struct StateField : Field<R> {
  using Field<R>::Field;
  __device__ __forceinline__ R load(int row, unsigned int chain_off) const {
    if constexpr (NTS) return Field<R>::load_nt(row, chain_off);
    else return Field<R>::load(row, chain_off);
  }
  __device__ __forceinline__ void store(int row, unsigned int chain_off, R value) const {
    if constexpr (NTS) Field<R>::store_nt(row, chain_off, value);
    else Field<R>::store(row, chain_off, value);
  }
};

// The chain state x[D] as the per-chain kernels address it.  Parameter spaces with fewer than kTiledStateDof real degrees of
// freedom keep it component-major (x[row * n + chain]: a handful of rows, where the layout costs nothing);
// larger ones keep it TILE-major like the packed fields ([tile of 64 chains][D rows][64 lanes]).  Measured for 64
// parameters in float64 at 2^19 chains (tools/dev/state_layout_probe64.hip, the geometry of k_step_dense64_f64, no
// arithmetic): 64 rows 4 MiB apart 101-116 us, one contiguous 32 KiB block per tile 82-88 us -- a quarter of the memory
// phase of BASELINE config 4.  The layout is a compile-time property of the kernel set (KernelSet::tiled_state); the host
// gathers / scatters tiles in me_get / me_set; the runtime-dimension set stays component-major.
#ifndef ME_TILED_STATE_DOF
#define ME_TILED_STATE_DOF 16     // (the 18-row state of the headline kernel gains 2-3 %: tools/dev/rows_probe_f64_tiled.hip)
#endif
constexpr int kTiledStateDof = ME_TILED_STATE_DOF;
template <typename R, int D, bool NTS = false>
struct XField {
  static constexpr bool kTiled = D >= kTiledStateDof;
  std::conditional_t<kTiled, TiledField<R>, StateField<R, NTS>> f;
  __device__ __forceinline__ XField(const R *base, long long n) : f(base, n, D) {}
  // per-lane offset of chain c (the row index is the other argument of load / store)
  static __device__ __forceinline__ unsigned int offset(long long c) {
    if constexpr (kTiled) return tiled_offset<R>(c, D);
    else return (unsigned int)c * (unsigned int)sizeof(R);
  }
  __device__ __forceinline__ R load(int row, unsigned int off) const {
    if constexpr (kTiled && NTS) return f.load_nt(row, off);
    else return f.load(row, off);
  }
  __device__ __forceinline__ void store(int row, unsigned int off, R value) const {
    if constexpr (kTiled && NTS) f.store_nt(row, off, value);
    else f.store(row, off, value);
  }
};
// the same address computation for kernels that walk the state with plain pointers
template <typename R, int D>
__device__ __forceinline__ long long state_index(long long c, int row, long long n) {
  if constexpr (D >= kTiledStateDof) return ((c >> 6) * (long long)D + row) * 64 + (c & 63);
  else return (long long)row * n + c;
}

// packed sizes up to this are kept in registers by the per-chain factor kernels (build.py: MAX_PACKED_IN_REGISTERS)
constexpr int kMaxPackedInRegisters = 160;
// float64, pure real spaces: k_step streams per-chain factors with more entries than this instead of holding them
#ifndef ME_STREAM_F64_ENTRIES
#define ME_STREAM_F64_ENTRIES 96          // the 136-entry factor of 16 real parameters streams; a huge value: never
#endif
constexpr int kStreamF64Entries = ME_STREAM_F64_ENTRIES;

// ------------------------------------------------------------------------------------------------ energies
// An energy is a small by-value functor evaluated on the chain's register-resident state
// x[D] = [real | Re z | Im z]; it stands in for the reference's Python callback (metropolis_engine.py:250).
template <typename R, int NR, int NC>
struct EnergyIso {  // a (sum x^2 + sum |z|^2)                                        README.md:26-27
  static constexpr int D = NR + 2 * NC;
  R a;
  __device__ __forceinline__ R operator()(const R (&x)[D]) const {
    // two interleaved partial sums: independent fma chains that pair up as v_pk_fma_f32 (one chain compiled to
    // 8 v_pk_mul + 15 dependent v_add at D = 16)
    R s0 = 0, s1 = 0;
#pragma unroll
    for (int d = 0; d + 1 < D; d += 2) {
      s0 += x[d] * x[d];
      s1 += x[d + 1] * x[d + 1];
    }
    if constexpr (D % 2 == 1) s0 += x[D - 1] * x[D - 1];
    return a * (s0 + s1);
  }
};

template <typename R, int NR, int NC>
struct EnergyDiag {  // sum a_i x_i^2 + sum b_j |z_j|^2; weights expanded to D entries on the host
  static constexpr int D = NR + 2 * NC;
  // The weights arrive as kernel arguments, i.e. in scalar registers: 2 D of them in float64.  At 16 parameters they no
  // longer fit beside the Philox keys and the field descriptors; the spilled ones come back through v_readlane in the middle
  // of the Box-Muller blocks, the scheduler gives up the occupancy it reaches for EnergyIso, and k_step<double,16,0,identity>
  // takes 171 vector registers (two wavefronts per SIMD) against 120 (four) -- forced to four it spills 44 bytes per lane.
  // Staging the weights in LDS (read back as broadcasts inside the sweep loop, behind a compiler barrier so that they are
  // not hoisted into registers again) was tried in round 3: 133 registers and 55 -> 54 us for that kernel, but the barrier
  // sits in every sweep of k_cycle too, whose packed matrix lives in registers across the sweeps: cycle(10) at (16,0) in
  // float64 1.4 -> 9.6 ms, config 3's cycle 3.1 -> 1.9 x 10^10 chain-steps/s.  Not kept.
  R w[D];
  __device__ __forceinline__ R operator()(const R (&x)[D]) const {
    R s0 = 0, s1 = 0;   // two interleaved chains (see EnergyIso)
#pragma unroll
    for (int d = 0; d + 1 < D; d += 2) {
      s0 += w[d] * x[d] * x[d];
      s1 += w[d + 1] * x[d + 1] * x[d + 1];
    }
    if constexpr (D % 2 == 1) s0 += w[D - 1] * x[D - 1] * x[D - 1];
    return s0 + s1;
  }
};

template <typename R, int NR, int NC>
struct EnergyDense {  // x^T A x, A[D][D] row-major in device memory
  static constexpr int D = NR + 2 * NC;
  static constexpr int PT = D * (D + 1) / 2;
  const R *a;
  // Only the symmetric part of A matters: E = sum_i x_i (T_ii x_i + sum_{j<i} T_ij x_j) with T_ij = A_ij + A_ji, half the
  // multiply-adds.  T lives in LDS (prepare(), once per block): every lane reads the SAME address, which LDS serves as a
  // broadcast.  The first version read A through scalar loads; hipcc hoisted all D*D of them, ran out of SGPRs and the
  // kernels spilled (D = 16: 120-238 AGPR copies in float32, 0.5-1.8 KB of scratch per lane in float64).
  static __device__ __forceinline__ R *folded() {
    __shared__ R t[PT];
    return t;
  }
  __device__ __forceinline__ void prepare() const {
    // Eight entries per thread at a time, their loads issued together: at D = 64 a one-sweep launch of 64-thread
    // workgroups spends 33 rounds per wavefront here, and one entry per round (row found by counting up, two dependent
    // loads) cost ~60 us per wavefront -- a third of k_step's time with streamed per-chain factors.
    R *t = folded();
    constexpr int kBatch = 8;
    for (int k0 = threadIdx.x; k0 < PT; k0 += kBatch * (int)blockDim.x) {
      R lo[kBatch], hi[kBatch];
#pragma unroll
      for (int u = 0; u < kBatch; ++u) {
        const int k = k0 + u * (int)blockDim.x;
        lo[u] = hi[u] = R(0);
        if (k < PT) {
          int i = (int)((__builtin_sqrtf((float)(8 * k + 1)) - 1.0f) * 0.5f);     // row of packed entry k
          if ((i + 1) * (i + 2) / 2 <= k) ++i;
          if (i * (i + 1) / 2 > k) --i;
          const int j = k - i * (i + 1) / 2;
          lo[u] = a[i * D + j];
          if (i != j) hi[u] = a[j * D + i];
        }
      }
#pragma unroll
      for (int u = 0; u < kBatch; ++u) {
        const int k = k0 + u * (int)blockDim.x;
        if (k < PT) t[k] = lo[u] + hi[u];
      }
    }
    __syncthreads();
  }
  __device__ __forceinline__ R operator()(const R (&x)[D]) const {
    const R *t = folded();
    // the reads of T are loop-invariant: hoisted out of the sweep loop they would sit in D(D+1)/2 registers for the whole
    // launch (136 at D = 16).  The clobber keeps them here, where each lives for one multiply-add.
    asm volatile("" ::: "memory");
    R e = 0;
#pragma unroll
    for (int i = 0; i < D; ++i) {
      R y = 0;
#pragma unroll
      for (int j = 0; j <= i; ++j) y += t[i * (i + 1) / 2 + j] * x[j];
      e += x[i] * y;
    }
    return e;
  }
};

template <typename R, int NR, int NC>
struct EnergyLandau {  // k(1-x)^2 + k(1-y)^2 + x y (alpha |c|^2 + beta |c|^4)   demo/toymodel_complex_and_real.py:17-26
  static constexpr int D = NR + 2 * NC;
  R k, alpha, beta;
  __device__ __forceinline__ R operator()(const R (&x)[D]) const {
    static_assert(NR == 2 && NC == 1, "Landau toy is 2 real + 1 complex");
    const R a2 = fma_(x[2], x[2], x[3] * x[3]);
    const R ox = R(1) - x[0], oy = R(1) - x[1];
    return fma_(x[0] * x[1], fma_(beta * a2, a2, alpha * a2), fma_(k * ox, ox, k * oy * oy));
  }
};

template <typename R, int NR, int NC>
struct EnergyCylinder {  // cylinder-style surrogate, see oracle/energies.py:cylinder_surrogate and DESIGN.md
  static constexpr int D = NR + 2 * NC;
  R kappa, gamma, wavenumber;
  __device__ __forceinline__ R operator()(const R (&x)[D]) const {
    static_assert(NR >= 1 && NC >= 1, "cylinder surrogate needs an amplitude and a field");
    R s = 0;
#pragma unroll
    for (int i = 0; i < NR; ++i) s += x[i] * x[i];
    const R x0 = x[0];
    const R surface = kappa * s / (R(1) - x0 * x0);
    const R amp = R(1) + R(0.5) * x0 * x0;
    R field = 0, tot = 0;
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      const R q = wavenumber * (R(j) - R(NC - 1) * R(0.5));
      const R mod2 = fma_(x[NR + j], x[NR + j], x[NR + NC + j] * x[NR + NC + j]);
      field += (gamma + q * q * amp) * mod2;
      tot += mod2;
    }
    return surface + field + R(0.5) * tot * tot;
  }
};

// The two-term dictionary form of the Landau toy (demo/toymodel_complex_and_real.py:17-33): "field" depends on both
// groups, "area" on the real group only.  Term-wise energies expose kTerms, term_groups(t) (bit 0: the real group
// moves it, bit 1: the complex group) and term(t, x); see EnergyLedger below.
template <typename R, int NR, int NC>
struct EnergyLandauTerms {
  static constexpr int D = NR + 2 * NC;
  static constexpr int kTerms = 2;
  R k, alpha, beta;
  static constexpr unsigned term_groups(int t) { return t == 0 ? 3u : 1u; }
  __device__ __forceinline__ R term(int t, const R (&x)[D]) const {
    static_assert(NR == 2 && NC == 1, "Landau toy is 2 real + 1 complex");
    if (t == 0) {
      const R a2 = fma_(x[2], x[2], x[3] * x[3]);
      return x[0] * x[1] * fma_(beta * a2, a2, alpha * a2);
    }
    const R ox = R(1) - x[0], oy = R(1) - x[1];
    return fma_(k * ox, ox, k * oy * oy);
  }
};

// A user-energy plugin may also supply the hard-wall predicate (ME_REJECT_USER): Energy::reject(x).  Built-in
// energies have none; the second overload answers "never reject" for them.
template <class E, typename R, int D>
__device__ __forceinline__ auto energy_reject(const E &en, const R (&x)[D], int) -> decltype(en.reject(x)) {
  return en.reject(x);
}
template <class E, typename R, int D>
__device__ __forceinline__ bool energy_reject(const E &, const R (&)[D], long) {
  return false;
}

// An energy may stage read-only data in LDS once per block (EnergyDense); the others have nothing to prepare.
template <class E>
__device__ __forceinline__ auto energy_prepare(const E &en, int) -> decltype(en.prepare()) {
  en.prepare();
}
template <class E>
__device__ __forceinline__ void energy_prepare(const E &, long) {}

// ------------------------------------------------------------------------------------------------ energy ledger
// The reference caches one energy per term and lets a group move re-evaluate only the terms registered for that group
// (energy dictionaries, metropolis_engine.py:111-116, :214-221, :230-237); a single callable is the one term "total".
// The ledger is the energy field: one row per term.  A GROUP kernel loads, re-evaluates, compares and stores only the
// rows its group can change; step_all touches every row.  With one term this is exactly `e = E(x)`.
template <class E, class = void>
struct term_count { static constexpr int value = 1; };
template <class E>
struct term_count<E, std::void_t<decltype(E::kTerms)>> { static constexpr int value = E::kTerms; };

template <typename R, class Energy, int GROUP>
struct EnergyLedger {
  static constexpr int T = term_count<Energy>::value;
  R term[T];
  static constexpr bool moves(int t) {
    if constexpr (T == 1 || GROUP == GROUP_ALL) return true;
    else return (Energy::term_groups(t) & (GROUP == GROUP_REAL ? 1u : 2u)) != 0u;
  }
  template <class F>
  __device__ __forceinline__ void load(const F &fe, unsigned int coff) {
#pragma unroll
    for (int t = 0; t < T; ++t)
      if (moves(t)) term[t] = fe.load(t, coff);
  }
  template <class F>
  __device__ __forceinline__ void store(const F &fe, unsigned int coff) const {
#pragma unroll
    for (int t = 0; t < T; ++t)
      if (moves(t)) fe.store(t, coff, term[t]);
  }
  __device__ __forceinline__ R partial() const {   // sum of the cached terms the group can change
    R s = 0;
    bool first = true;
#pragma unroll
    for (int t = 0; t < T; ++t)
      if (moves(t)) {
        s = first ? term[t] : s + term[t];
        first = false;
      }
    return s;
  }
  // the same terms at the proposed state -> out[], returns their sum
  template <int D>
  __device__ __forceinline__ R propose(const Energy &en, const R (&xp)[D], R (&out)[T]) const {
    if constexpr (T == 1) {
      out[0] = en(xp);
      return out[0];
    } else {
      R s = 0;
      bool first = true;
#pragma unroll
      for (int t = 0; t < T; ++t)
        if (moves(t)) {
          out[t] = en.term(t, xp);
          s = first ? out[t] : s + out[t];
          first = false;
        }
      return s;
    }
  }
  __device__ __forceinline__ void commit(bool accept, const R (&out)[T]) {
#pragma unroll
    for (int t = 0; t < T; ++t)
      if (moves(t)) term[t] = accept ? out[t] : term[t];
  }
};

// ------------------------------------------------------------------------------------------------ k_step
template <typename R>
struct StepArgs {
  R *x, *energy, *width;
  const R *factor;
  const R *inj_normals;   // INJECT only: [sweep][D][n] standard normals replacing the Philox draws
  const R *inj_uniforms;  // INJECT only: [sweep][n] accept uniforms
  unsigned long long *accept_slots;   // one slot per wavefront of the grid: [gridDim.x * 4]
  unsigned int *status;
  long long n;
  unsigned long long chain_offset, step_index;
  uint32_t seed_lo, seed_hi;
  int n_sweeps, reject_kind;
  int split_widths;   // mixed engines: group widths (rows 1, 2) differ from the shared width (row 0)
  int stale_total;    // quirk Q5 mode: a mixed engine's step_all compares against / updates ledger row T only
  R reject_bound, temp, inv_temp, inv_temp_log2e, ratio, p, damping, up, down;
};

// packed index of element (i, j), j <= i, of a row-major lower triangle
__host__ __device__ constexpr int tri(int i, int j) { return i * (i + 1) / 2 + j; }
// complex block (after the PR real entries): row i holds (Re,Im) of columns j < i, then the real diagonal
__host__ __device__ constexpr int cre(int pr, int i, int j) { return pr + i * i + 2 * j; }
__host__ __device__ constexpr int cim(int pr, int i, int j) { return pr + i * i + 2 * j + 1; }
__host__ __device__ constexpr int cdiag(int pr, int i) { return pr + i * i + 2 * i; }

// fn(integral_constant<int, 0>) ... fn(integral_constant<int, N - 1>): a loop whose index is a constant expression
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&fn, std::integer_sequence<int, I...>) {
  (fn(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&fn) {
  static_for_impl(fn, std::make_integer_sequence<int, N>{});
}
// A 64-bit address put back together from two 32-bit halves that went through v_readfirstlane (which returns int).  Both
// halves are ZERO-extended: the first version or-ed the low half in as a (signed) int, so a tile base whose low half had
// bit 31 set sign-extended into the high half -- a wild pointer in the streamed-factor descriptor, the asynchronous GPU
// fault that aborted test_64_real_per_chain_shapes_follow_the_oracle on Oct 4 (gpurun_out/streamed.log; fixed in dad04c0).
__host__ __device__ constexpr unsigned long long join_halves(int hi, int lo) {
  return ((unsigned long long)(unsigned int)hi << 32) | (unsigned long long)(unsigned int)lo;
}
static_assert(join_halves(0x00007f12, (int)0x80000000u) == 0x00007f1280000000ull, "low half with bit 31 set must not sign-extend");
static_assert(join_halves((int)0xffff8000u, (int)0xfffffff0u) == 0xffff8000fffffff0ull, "both halves are zero-extended");
// bytes of one chunk of the streamed factor per lane (two chunks live in registers)
#ifndef ME_STREAM_CHUNK_BYTES
#define ME_STREAM_CHUNK_BYTES 128
#endif

// The pieces of one sweep of ONE chain, state in registers (metropolis_engine.py:209-338, :429-456): draw_step (the Philox
// block -> normals + accept uniform), propose_registers (x' from the identity shape, a shared factor or the chain's own
// register-resident factor), decide_step (wall -> energy -> accept rule -> commit -> width adaptation).  run_sweeps strings
// them together for k_step and k_cycle; k_step's STREAMED-factor form keeps its proposal in the kernel body (that
// straight-line code over thousands of entries only stays in registers when it is unrolled inside the kernel itself: moved
// into an inlined function it went to 12 KB of scratch per lane and 13 x the compile time).  All forced inline.
template <typename R, int NR, int NC, bool INJECT>
__device__ __forceinline__ void draw_step(const StepArgs<R> &a, long long c, unsigned long long gid, int s,
                                          R (&g)[2 * ((NR + 2 * NC + 1) / 2)], R &u) {
  constexpr int D = NR + 2 * NC;
  constexpr int NW = 2 * ((D + 1) / 2);     // words consumed by the Box-Muller pairs
  constexpr int NBLK = (NW + 1 + 3) / 4;    // Philox blocks per step (word NW is the accept uniform)
  using N_ = Num<R>;
  [[maybe_unused]] const unsigned long long step = a.step_index + (unsigned long long)s;
  if constexpr (INJECT) {
#pragma unroll
    for (int j = 0; j < D; ++j) g[j] = a.inj_normals[((long long)s * D + j) * a.n + c];
    u = a.inj_uniforms[(long long)s * a.n + c];
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
    for (int q = 0; q < NW / 2; ++q) N_::normal_pair(words[2 * q], words[2 * q + 1], g[2 * q], g[2 * q + 1]);
    u = N_::unit(words[NW]);
  }

}

//   fac(k)   entry k of the proposal factor (per-chain: a coalesced load; shared: a scalar load)
template <typename R, int NR, int NC, int CK, int GROUP, class Fac>
__device__ __forceinline__ void propose_registers(const R (&x)[NR + 2 * NC], const R (&g)[2 * ((NR + 2 * NC + 1) / 2)], R w_r, R w_c,
                                                  Fac &&fac, R (&xp)[NR + 2 * NC]) {
  constexpr int D = NR + 2 * NC;
  [[maybe_unused]] constexpr int PR = NR * (NR + 1) / 2;
  constexpr bool MOVE_REAL = NR > 0 && GROUP != GROUP_COMPLEX;
  constexpr bool MOVE_COMPLEX = NC > 0 && GROUP != GROUP_REAL;
  // ---- proposal: x' = x + w_r L_r g_r ; z' = z + w_c L_c (g_re + i g_im)/sqrt2   (:261-302)
#pragma unroll
  for (int d = 0; d < D; ++d) xp[d] = x[d];
  if constexpr (CK == CK_IDENTITY) {
    if constexpr (MOVE_REAL) {
#pragma unroll
      for (int i = 0; i < NR; ++i) xp[i] = x[i] + w_r * g[i];
    }
    if constexpr (MOVE_COMPLEX) {
#pragma unroll
      for (int j = 0; j < NC; ++j) {
        xp[NR + j] = x[NR + j] + w_c * (g[NR + j] * R(0.70710678118654752440));
        xp[NR + NC + j] = x[NR + NC + j] + w_c * (g[NR + NC + j] * R(0.70710678118654752440));
      }
    }
  } else {
    if constexpr (MOVE_REAL) {
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        R acc = 0;
#pragma unroll
        for (int j = 0; j <= i; ++j) acc += fac(tri(i, j)) * g[j];
        xp[i] = x[i] + w_r * acc;
      }
    }
    if constexpr (MOVE_COMPLEX) {
#pragma unroll
      for (int i = 0; i < NC; ++i) {
        R are = 0, aim = 0;
#pragma unroll
        for (int j = 0; j < i; ++j) {
          const R lre = fac(cre(PR, i, j));
          const R lim = fac(cim(PR, i, j));
          const R wre = g[NR + j] * R(0.70710678118654752440), wim = g[NR + NC + j] * R(0.70710678118654752440);
          are += lre * wre - lim * wim;
          aim += lre * wim + lim * wre;
        }
        const R ld = fac(cdiag(PR, i));
        are += ld * (g[NR + i] * R(0.70710678118654752440));
        aim += ld * (g[NR + NC + i] * R(0.70710678118654752440));
        xp[NR + i] = x[NR + i] + w_c * are;
        xp[NR + NC + i] = x[NR + NC + i] + w_c * aim;
      }
    }
  }

}

template <typename R, int NR, int NC, class Energy, int CK, int GROUP, class Ledger>
__device__ __forceinline__ void decide_step(const StepArgs<R> &a, const Energy &en, bool stale_total, R u,
                                            const R (&g)[2 * ((NR + 2 * NC + 1) / 2)], R (&x)[NR + 2 * NC], const R (&xp)[NR + 2 * NC],
                                            Ledger &ledger, R &total_q5, R &w, R &w_r, R &w_c, unsigned int &wave_accepted,
                                            bool &bad_energy) {
  [[maybe_unused]] constexpr int D = NR + 2 * NC;
  constexpr bool MOVE_REAL = NR > 0 && GROUP != GROUP_COMPLEX;
  constexpr bool MOVE_COMPLEX = NC > 0 && GROUP != GROUP_REAL;
  using N_ = Num<R>;
  // ---- hard wall before the energy (:247-249), energy, accept rule (:319-338)
  bool rejected = false;
  if (a.reject_kind == ME_REJECT_ABS_REAL0_GE) rejected = !(N_::abs_(xp[0]) < a.reject_bound);
  else if (a.reject_kind == ME_REJECT_USER) rejected = energy_reject(en, xp, 0);
  R terms_new[Ledger::T];
  const R e_new = ledger.propose(en, xp, terms_new);
  const R diff = e_new - (stale_total ? total_q5 : ledger.partial());
  bool accept = diff <= R(0);
  if (a.temp > R(0)) accept = accept || N_::uphill(u, diff, a.inv_temp, a.inv_temp_log2e);
  accept = accept && !rejected;
  bad_energy |= (!rejected && !N_::finite(e_new));
  if constexpr (CK == CK_IDENTITY) {
    // commit as x += (accept ? w : 0) g: the same fma that formed x' (bit-identical on acceptance, exact identity on
    // rejection since the draws are finite), and it pairs up as v_pk_fma_f32 where 16 v_cndmask_b32 would not
    const R wa_r = accept ? w_r : R(0), wa_c = accept ? w_c : R(0);
    if constexpr (MOVE_REAL) {
#pragma unroll
      for (int i = 0; i < NR; ++i) x[i] = x[i] + wa_r * g[i];
    }
    if constexpr (MOVE_COMPLEX) {
#pragma unroll
      for (int j = 0; j < NC; ++j) {
        x[NR + j] = x[NR + j] + wa_c * (g[NR + j] * R(0.70710678118654752440));
        x[NR + NC + j] = x[NR + NC + j] + wa_c * (g[NR + NC + j] * R(0.70710678118654752440));
      }
    }
  } else {
#pragma unroll
    for (int d = 0; d < D; ++d) x[d] = accept ? xp[d] : x[d];
  }
  if (stale_total) total_q5 = accept ? e_new : total_q5;
  else ledger.commit(accept, terms_new);
  // ---- Robbins-Monro width update (:429-456); step_all of a mixed engine mirrors it into both groups (:436-437)
  w = N_::adapt(w, accept, a.ratio, a.p, a.damping, a.up, a.down);
  if constexpr (GROUP == GROUP_ALL) w_r = w_c = w;
  else if constexpr (GROUP == GROUP_REAL) w_r = w;
  else w_c = w;
  wave_accepted += (unsigned int)__popcll(__ballot(accept));
}

// n_sweeps sweeps with a register-resident proposal shape:  x, ledger / total_q5, w, w_r, w_c are updated in place
template <typename R, int NR, int NC, class Energy, int CK, bool INJECT, int GROUP, class Ledger, class Fac>
__device__ __forceinline__ void run_sweeps(const StepArgs<R> &a, const Energy &en, long long c, unsigned long long gid,
                                           bool stale_total, R (&x)[NR + 2 * NC], Ledger &ledger, R &total_q5, R &w, R &w_r,
                                           R &w_c, Fac &&fac, unsigned int &wave_accepted, bool &bad_energy) {
  constexpr int D = NR + 2 * NC;
  constexpr int NW = 2 * ((D + 1) / 2);
  for (int s = 0; s < a.n_sweeps; ++s) {
    R g[NW], u, xp[D];
    draw_step<R, NR, NC, INJECT>(a, c, gid, s, g, u);
    propose_registers<R, NR, NC, CK, GROUP>(x, g, w_r, w_c, fac, xp);
    decide_step<R, NR, NC, Energy, CK, GROUP>(a, en, stale_total, u, g, x, xp, ledger, total_q5, w, w_r, w_c, wave_accepted, bad_energy);
  }
}

// INJECT = true replaces the Philox draws by caller-provided streams (test hook: replays the reference's golden
// trajectories, tests/golden/, through the very same proposal / accept / adapt code).
// Tuning knobs (experiments; defaults are the shipped configuration):
//   ME_STEP_WAVES_PER_EU  second __launch_bounds__ argument (min waves per SIMD -> VGPR budget); 0 = unset
#ifndef ME_STEP_WAVES_PER_EU
#define ME_STEP_WAVES_PER_EU 0
#endif
#if ME_STEP_WAVES_PER_EU > 0
#define ME_STEP_BOUNDS __launch_bounds__(kFusedStepThreads, ME_STEP_WAVES_PER_EU)
#else
#define ME_STEP_BOUNDS __launch_bounds__(kFusedStepThreads)
#endif
// GROUP selects which coordinates move: GROUP_ALL = step_all (:241-259); GROUP_REAL / GROUP_COMPLEX = a mixed
// engine's step_real_group / step_complex_group called directly (:209-239).  The word layout of a step is the same
// in all three (normal i belongs to coordinate i); draws of the resting group are dead code and are eliminated.
// Widths of mixed engines are three rows [sampling_width, real group, complex group] (:93-99, :429-456):
// step_all adapts row 0 and mirrors it into the group widths (:436-437), a group step adapts only its own row.

// Wavefronts that share a SIMD start together, load together and then compete for the vector pipe together: all of them
// finish their first sweep at the same late moment, and until then nothing is stored (the ramp: a fixed ~4.6 us on top of
// the memory-only kernel's at 2^20 chains, profiles/r03_probes_and_variants.txt).  Different issue priorities for the
// workgroups of a launch let one wavefront of each SIMD run ahead instead.  ME_STEP_PRIO_SHIFT < 0: off.
#ifndef ME_STEP_PRIO_SHIFT
#define ME_STEP_PRIO_SHIFT 8     // measured 0 .. 12 on the float64 headline kernel: 4 .. 10 all give 51.5 -> 49.0 us, 0 / 2 / 12 half of that
#endif
__device__ __forceinline__ void stagger_priority() {
#if ME_STEP_PRIO_SHIFT >= 0
  switch ((blockIdx.x >> ME_STEP_PRIO_SHIFT) & 3u) {
    case 0: __builtin_amdgcn_s_setprio(3); break;
    case 1: __builtin_amdgcn_s_setprio(2); break;
    case 2: __builtin_amdgcn_s_setprio(1); break;
    default: break;
  }
#endif
}

template <typename R, int NR, int NC, class Energy, int CK, bool INJECT = false, int GROUP = GROUP_ALL, bool NTS = false>
__global__ void ME_STEP_BOUNDS k_step(StepArgs<R> a, Energy en) {
  constexpr int D = NR + 2 * NC;
  constexpr bool MIXED = NR > 0 && NC > 0;
  static_assert(GROUP == GROUP_ALL || MIXED, "group-wise kernels exist for mixed engines only");
  using N_ = Num<R>;

  stagger_priority();
  if constexpr (!INJECT) N_::prepare();
  energy_prepare(en, 0);
  unsigned int wave_accepted = 0;
  bool bad_energy = false, bad_width = false;
  const long long stride = (long long)gridDim.x * blockDim.x;   // 64 threads per block, or 256 for fused sweeps
  using Ledger = EnergyLedger<R, Energy, GROUP>;
  const XField<R, D, NTS> fx(a.x, a.n);
  const StateField<R, NTS> fe(a.energy, a.n, Ledger::T + (MIXED ? 1 : 0)), fw(a.width, a.n, MIXED ? 3 : 1);
  // The reference keeps TWO ledgers: step_all of a mixed engine decides against `energy_total` and
  // updates only that (metropolis_engine.py:252-255); group steps use `energy[term]` (:214-221, :230-237).  With
  // ME_FLAG_REFERENCE_ENERGY_LEDGERS the total lives in ledger row T and this kernel touches only the rows the
  // reference's function touches; by default there is one coherent ledger.
  const bool stale_total = MIXED && GROUP == GROUP_ALL && a.stale_total != 0;
  constexpr bool PER_CHAIN = CK == CK_PER_CHAIN || CK == CK_PER_CHAIN_NT;
  constexpr int PF = PER_CHAIN ? NR * (NR + 1) / 2 + NC * NC : 0;
  // per-chain factors beyond the register-resident size are streamed (pure real spaces only) -- and so are float64 factors
  // of more than kStreamF64Entries entries: 136 doubles (16 real parameters) are 272 registers, ONE wavefront per SIMD,
  // and with nobody to run beside it a wavefront's arithmetic (RNG, L g) adds to its load time -- 246 us against a
  // memory-only floor of 214 us at 2^20 chains (tools/dev/cov_probe_f64.hip).  Streamed in two 128-byte chunks per lane
  // the kernel needs ~100 registers and four wavefronts share a SIMD.
  constexpr bool STREAM_FACTOR = PER_CHAIN && (PF > kMaxPackedInRegisters ||
                                               (!INJECT && sizeof(R) == 8 && NC == 0 && PF > kStreamF64Entries));
  // (round 3: mixed and complex spaces stream too -- the complex rows follow the real triangle in the packed order)
  const TiledField<R> ffac(a.factor, a.n, STREAM_FACTOR ? 0 : PF);
  for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < a.n; c += stride) {
    const unsigned int coff = (unsigned int)c * (unsigned int)sizeof(R);
    const unsigned int xoff = fx.offset(c);
    R x[D];
#pragma unroll
    for (int d = 0; d < D; ++d) x[d] = fx.load(d, xoff);
    Ledger ledger;
    R total_q5 = R(0);
    if (stale_total) total_q5 = fe.load(Ledger::T, coff);
    else ledger.load(fe, coff);
    // w: the width this launch adapts; w_r / w_c: the widths the real / complex proposals use
    R w = fw.load(MIXED ? GROUP : 0, coff);
    R w_r = w, w_c = w;
    if constexpr (MIXED && GROUP == GROUP_ALL) {
      if (a.split_widths) {   // a group step ran since the last step_all: the group widths differ from row 0
        w_r = fw.load(GROUP_REAL, coff);
        w_c = fw.load(GROUP_COMPLEX, coff);
      }
    }
    const unsigned long long gid = a.chain_offset + (unsigned long long)c;
    [[maybe_unused]] const unsigned int toff = tiled_offset<R>(c, PF);
    // CK_PER_CHAIN(_NT) reads entry k of the chain's tile (coalesced, tile-major; hipcc hoists all of them out of the sweep loop, so
    // the factor is read once per launch); CK_SHARED reads factor[k] (wave-uniform scalar loads).  The _NT variant reads
    // non-temporally: chosen by the launcher when the working set cannot stay in the Infinity Cache (me_kernels.hip).
    // Tried and dropped: handing the entries out of two alternating register batches (106 instead of 221 VGPRs, four
    // wavefronts per SIMD instead of two) changed nothing -- this kernel sits on the memory-only floor of its access
    // pattern at either occupancy (tools/dev/rows_probe3.hip).
    auto fac = [&](int k) -> R {
      if constexpr (CK == CK_PER_CHAIN_NT) return ffac.load_nt(k, toff);
      else if constexpr (CK == CK_PER_CHAIN) return ffac.load(k, toff);
      else return a.factor[k];
    };

    if constexpr (STREAM_FACTOR) {
      constexpr int NW = 2 * ((D + 1) / 2);
      for (int s = 0; s < a.n_sweeps; ++s) {
        // (draw and decide are spelled out here instead of calling draw_step / decide_step: with the arrays handed to inlined
        // functions by reference this kernel -- thousands of unrolled entries -- kept 1.4 KB of them in scratch)
        static_assert(!INJECT, "the streamed-factor form has no replay hook");
        constexpr int NBLK = (NW + 1 + 3) / 4;
        const unsigned long long step = a.step_index + (unsigned long long)s;
        R g[NW];
        R u_accept;
        {
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
          for (int q = 0; q < NW / 2; ++q) N_::normal_pair(words[2 * q], words[2 * q + 1], g[2 * q], g[2 * q + 1]);
          u_accept = N_::unit(words[NW]);
        }
        R xp[D];
#pragma unroll
        for (int d = 0; d < D; ++d) xp[d] = x[d];
        // Packed factors too large for registers (more than 160 entries; pure real spaces, e.g. 64 parameters = 2 080
        // entries = 8-16 KB per chain and step).  The chain's entries are one run of its tile (64 values apart), read
        // exactly once per step -- the kernel is a STREAM and lives on bytes in flight: with one wavefront per SIMD a
        // batch of loads that is waited for before the next is issued exposes the full memory latency 130 times per
        // step (the first version: 2.9 ms at 2^19 chains against a traffic floor of 0.8).  So: straight-line code over
        // the whole triangle in chunks of ME_STREAM_CHUNK entries, chunk c+1 is in flight while chunk c is multiplied,
        // normals and proposals in registers with compile-time indices.
        // The field may pass the 4 GiB a buffer descriptor spans: the descriptor covers the wavefront's own tile.
        // (the clobber keeps the loop-invariant loads inside the sweep loop: hoisted, they went to 16 KB of scratch)
        asm volatile("" ::: "memory");
        constexpr int CH = (int)(ME_STREAM_CHUNK_BYTES / sizeof(R));
        const R *tile_base = a.factor + (c >> 6) * (long long)PF * 64;
        const unsigned long long tb = (unsigned long long)tile_base;
        const R *tile_uniform = (const R *)join_halves(__builtin_amdgcn_readfirstlane((int)(tb >> 32)),
                                                       __builtin_amdgcn_readfirstlane((int)(unsigned int)tb));
        const __amdgpu_buffer_rsrc_t tile = __builtin_amdgcn_make_buffer_rsrc(const_cast<R *>(tile_uniform), 0,
                                                                               (unsigned int)(PF * 64 * sizeof(R)), 0x00020000);
        const unsigned int lane_off = (unsigned int)(c & 63) * (unsigned int)sizeof(R);
        auto entry = [&](int k) -> R {
          constexpr int AUX = CK == CK_PER_CHAIN_NT ? ME_NT_AUX : 0;
          if constexpr (sizeof(R) == 4)
            return __builtin_bit_cast(R, __builtin_amdgcn_raw_buffer_load_b32(tile, lane_off, (unsigned int)k * 64u * 4u, AUX));
          else
            return __builtin_bit_cast(R, __builtin_amdgcn_raw_buffer_load_b64(tile, lane_off, (unsigned int)k * 64u * 8u, AUX));
        };
        R f[2][CH];
#pragma unroll
        for (int u = 0; u < CH; ++u)
          if (u < PF) f[0][u] = entry(u);
        // entry k of the packed factor, consumed in ascending order: entering a chunk puts the one after it in flight first
        auto use = [&](auto entry_index) -> R {
          constexpr int k = decltype(entry_index)::value;
          if constexpr (k % CH == 0) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < CH; ++u)
              if (k + CH + u < PF) f[(k / CH + 1) & 1][u] = entry(k + CH + u);
            __builtin_amdgcn_sched_barrier(0);
          }
          return f[(k / CH) & 1][k % CH];
        };
        // (row and column are template constants: as `#pragma unroll` loop variables a row of 64 columns with its chunk of
        // loads passes LLVM's size limit for forced unrolling, is unrolled too late, and the arrays stay in scratch)
        static_for<NR>([&](auto row_index) {
          constexpr int i = decltype(row_index)::value;
          R acc = 0;
          static_for<i + 1>([&](auto col_index) {
            constexpr int j = decltype(col_index)::value, k = tri(i, j);
            const R l = use(std::integral_constant<int, k>{});
            acc = j == 0 ? l * g[0] : acc + l * g[j];
          });
          if constexpr (GROUP != GROUP_COMPLEX) xp[i] = x[i] + w_r * acc;      // (a resting group's rows are walked, not applied)
        });
        // complex rows (:274-302): z'_i = z_i + w_c sum_{j <= i} L_ij (g_re_j + i g_im_j) / sqrt 2, L = chol(conj K) packed as
        // (Re, Im) of the columns j < i, then the real diagonal -- the same arithmetic, term by term, as propose_registers
        static_for<NC>([&](auto row_index) {
          constexpr int i = decltype(row_index)::value;
          R are = 0, aim = 0;
          static_for<i>([&](auto col_index) {
            constexpr int j = decltype(col_index)::value;
            const R lre = use(std::integral_constant<int, cre(NR * (NR + 1) / 2, i, j)>{});
            const R lim = use(std::integral_constant<int, cim(NR * (NR + 1) / 2, i, j)>{});
            const R wre = g[NR + j] * R(0.70710678118654752440), wim = g[NR + NC + j] * R(0.70710678118654752440);
            are += lre * wre - lim * wim;
            aim += lre * wim + lim * wre;
          });
          const R ld = use(std::integral_constant<int, cdiag(NR * (NR + 1) / 2, i)>{});
          are += ld * (g[NR + i] * R(0.70710678118654752440));
          aim += ld * (g[NR + NC + i] * R(0.70710678118654752440));
          if constexpr (GROUP != GROUP_REAL) {
            xp[NR + i] = x[NR + i] + w_c * are;
            xp[NR + NC + i] = x[NR + NC + i] + w_c * aim;
          }
        });
        // ---- hard wall before the energy (:247-249), energy, accept rule (:319-338), commit, width (decide_step's code)
        bool rejected = false;
        if (a.reject_kind == ME_REJECT_ABS_REAL0_GE) rejected = !(N_::abs_(xp[0]) < a.reject_bound);
        else if (a.reject_kind == ME_REJECT_USER) rejected = energy_reject(en, xp, 0);
        R terms_new[Ledger::T];
        const R e_new = ledger.propose(en, xp, terms_new);
        const R diff = e_new - (stale_total ? total_q5 : ledger.partial());
        bool accept = diff <= R(0);
        if (a.temp > R(0)) accept = accept || N_::uphill(u_accept, diff, a.inv_temp, a.inv_temp_log2e);
        accept = accept && !rejected;
        bad_energy |= (!rejected && !N_::finite(e_new));
#pragma unroll
        for (int d = 0; d < D; ++d) x[d] = accept ? xp[d] : x[d];
        if (stale_total) total_q5 = accept ? e_new : total_q5;
        else ledger.commit(accept, terms_new);
        w = N_::adapt(w, accept, a.ratio, a.p, a.damping, a.up, a.down);
        if constexpr (GROUP == GROUP_ALL) w_r = w_c = w;
        else if constexpr (GROUP == GROUP_REAL) w_r = w;
        else w_c = w;
        wave_accepted += (unsigned int)__popcll(__ballot(accept));
      }
    } else {
      run_sweeps<R, NR, NC, Energy, CK, INJECT, GROUP>(a, en, c, gid, stale_total, x, ledger, total_q5, w, w_r, w_c, fac, wave_accepted, bad_energy);
    }
    bad_width |= !(w > R(0));
#pragma unroll
    for (int d = 0; d < D; ++d) fx.store(d, xoff, x[d]);
    if (stale_total) fe.store(Ledger::T, coff, total_q5);
    else ledger.store(fe, coff);
    fw.store(MIXED ? GROUP : 0, coff, w);   // after a mixed step_all rows 1, 2 are implied equal to row 0 (host flag)
  }
  // acceptance tracking: ballot + popcount per sweep, then ONE plain read-modify-write of the wavefront's own
  // slot per launch.  (Same-address atomics serialise at ~12 ns each at the memory side: 2^14 wavefronts adding
  // to one counter cost 0.4 ms per launch, 15x the whole state sweep.)  Slots are summed on demand by k_sum_slots.
  if ((threadIdx.x & 63) == 0 && wave_accepted) {
    unsigned long long *slot = a.accept_slots + (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    *slot += (unsigned long long)wave_accepted;
  }
  const unsigned int bits = (bad_energy ? ST_NONFINITE_ENERGY : 0u) | (bad_width ? ST_BAD_WIDTH : 0u);
  if (bits) atomicOr(a.status, bits);
}

template <typename R, int NR, int NC, class Energy>
__global__ void __launch_bounds__(kStepThreads) k_init_energy(const R *xs, R *energy, long long n, unsigned int *status,
                                                                Energy en, int total_row) {
  constexpr int D = NR + 2 * NC;
  energy_prepare(en, 0);
  const long long stride = (long long)gridDim.x * kStepThreads;
  for (long long c = (long long)blockIdx.x * kStepThreads + threadIdx.x; c < n; c += stride) {
    R x[D];
#pragma unroll
    for (int d = 0; d < D; ++d) x[d] = xs[state_index<R, D>(c, d, n)];
    constexpr int T = term_count<Energy>::value;   // every term of the ledger (initialize_energy_dict, :152-155)
    bool finite = true;
    R total = R(0);
#pragma unroll
    for (int t = 0; t < T; ++t) {
      R e;
      if constexpr (T == 1) e = en(x);
      else e = en.term(t, x);
      energy[(long long)t * n + c] = e;
      total = t == 0 ? e : total + e;
      finite = finite && Num<R>::finite(e);
    }
    if (total_row >= 0) energy[(long long)total_row * n + c] = total;   // the reference's energy_total (:125)
    if (!finite) atomicOr(status, (unsigned int)ST_NONFINITE_ENERGY);
  }
}

// ------------------------------------------------------------------------------------------------ k_measure
template <typename R>
struct MeasureArgs {
  const R *x, *width;
  R *mean, *cov, *obs_mean, *factor;
  unsigned int *status;
  long long n;
  R keep;      // (i-1)/i
  R inv_i;     // 1/i
  R cov_keep;  // (i-2)/(i-1)
  int update_cov, write_factor, split_widths;
};

// Running mean, Haario-type covariance recursion with the reference's undivided epsilon term (quirk Q1),
// observables, and the refresh of the packed Cholesky factors the next proposals use.
// The covariance update is the algebraically identical one-pass form
//     C <- C (i-2)/(i-1) + (x - mu_old)(x - mu_old)^H / i + (sigma^2 / i) I
// of metropolis_engine.py:416-427 (mu_old mu_old^H - i/(i-1) mu mu^H + x x^H/(i-1) == delta delta^H / i),
// which has no cancellation when |mean| >> std and is therefore safe in fp32.
// In-register Cholesky of the packed proposal matrix (real block, then the Hermitian block which must already hold
// conj(K), quirk Q3): m is overwritten by the factor.  A non-positive pivot is flagged and clamped.
template <typename R, int NR, int NC>
__device__ __forceinline__ void cholesky_packed(R (&m)[NR * (NR + 1) / 2 + NC * NC], bool &bad_pivot) {
  constexpr int PR = NR * (NR + 1) / 2;
  using N_ = Num<R>;
    // in-register Cholesky, real block
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      R s = m[tri(j, j)];
#pragma unroll
      for (int k = 0; k < j; ++k) s = fma_(-m[tri(j, k)], m[tri(j, k)], s);
      if (!(s > R(0))) { bad_pivot = true; s = R(1e-30); }
      const R dg = N_::sqrt_(s);
      const R inv = R(1) / dg;
      m[tri(j, j)] = dg;
#pragma unroll
      for (int i = j + 1; i < NR; ++i) {
        R t = m[tri(i, j)];
#pragma unroll
        for (int k = 0; k < j; ++k) t = fma_(-m[tri(i, k)], m[tri(j, k)], t);
        m[tri(i, j)] = t * inv;
      }
    }
    // complex Hermitian block: L L^H = conj(K)
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      R s = m[cdiag(PR, j)];
#pragma unroll
      for (int k = 0; k < j; ++k) s = fma_(-m[cim(PR, j, k)], m[cim(PR, j, k)], fma_(-m[cre(PR, j, k)], m[cre(PR, j, k)], s));
      if (!(s > R(0))) { bad_pivot = true; s = R(1e-30); }
      const R dg = N_::sqrt_(s);
      const R inv = R(1) / dg;
      m[cdiag(PR, j)] = dg;
#pragma unroll
      for (int i = j + 1; i < NC; ++i) {
        R tr = m[cre(PR, i, j)], ti = m[cim(PR, i, j)];
#pragma unroll
        for (int k = 0; k < j; ++k) {
          // t -= L_ik conj(L_jk)
          const R ar = m[cre(PR, i, k)], ai2 = m[cim(PR, i, k)];
          const R br = m[cre(PR, j, k)], bi2 = m[cim(PR, j, k)];
          tr = fma_(-ai2, bi2, fma_(-ar, br, tr));
          ti = fma_(ar, bi2, fma_(-ai2, br, ti));
        }
        m[cre(PR, i, j)] = tr * inv;
        m[cim(PR, i, j)] = ti * inv;
      }
    }
}

// FUSED = true refreshes the proposal factors in the same kernel (the updated covariance stays in registers).  With the
// loads of every phase batched ahead of its stores this is the faster form for every per-chain kernel set, 16 real
// parameters (136 entries, 253 VGPRs, two wavefronts per SIMD) included: 460 us against 570 us for the split form
// (FUSED = false followed by k_factor), which is kept for experiments (-DME_MEASURE_FUSED_MAX_P).
#ifndef ME_MEASURE_WAVES
#define ME_MEASURE_WAVES 1
#endif
// NT: the packed covariance / factor fields are accessed non-temporally; NTM: the running means and observables too.
// Both are decided per launch from the sizes (me_kernels.hip: measure): a field goes nt when the fields that COULD stay
// resident in the Infinity Cache between launches no longer fit together with it.  Measured at 2^20 x (16,0): packed
// fields nt 468 -> 375 us, means nt on top of that 375 -> 410 us (x + means + observables = 256 MB still profit from the
// cache); at 2^19 x (64,0), means only: 225 -> 190 us with nt (537 MB: keeping x resident for the next k_step wins).
// Where measure_chain takes the running mean, the observable means and the packed covariance from: memory, when it asks
// (k_measure), or registers that were loaded BEFORE the sweeps of a cycle (k_cycle: the loads travel while the sweeps run).
template <typename R, bool NT, bool NTM>
struct MeasureFromMemory {
  const Field<R> &fmean, &fobs;
  const TiledField<R> &fcov;
  unsigned int coff, toff;
  __device__ __forceinline__ R mean(int d) const { return packed_load<NTM>(fmean, d, coff); }
  __device__ __forceinline__ R obs(int k) const { return packed_load<NTM>(fobs, k, coff); }
  __device__ __forceinline__ R cov(int k) const { return packed_load<NT>(fcov, k, toff); }
};
// LEVEL 1: mean and observables preloaded, covariance from memory; LEVEL 2: everything preloaded
template <typename R, int D, int NOBS, int P, int LEVEL, bool NT>
struct MeasurePreloaded {
  R mu[D], ob[NOBS], cv[LEVEL >= 2 ? P : 1];
  const TiledField<R> &fcov;
  unsigned int toff;
  __device__ __forceinline__ R mean(int d) const { return mu[d]; }
  __device__ __forceinline__ R obs(int k) const { return ob[k]; }
  __device__ __forceinline__ R cov(int k) const {
    if constexpr (LEVEL >= 2) return cv[k];
    else return packed_load<NT>(fcov, k, toff);
  }
};

// measure() of ONE chain whose state x is in registers: running mean (:404-410), observables (:458-463, :412-414),
// covariance recursion (:416-427, one-pass form above) and the refresh of the chain's proposal factors.  Shared by
// k_measure (loads x) and k_cycle (x comes straight out of the sweeps); forced inline.  widths(w_real, w_cplx) supplies
// the group widths of the epsilon term (:418, :425) -- a field load in k_measure, registers in k_cycle.
template <typename R, int NR, int NC, bool PER_CHAIN_COV, bool FUSED, bool NT, bool NTM, class Source, class WidthFn>
__device__ __forceinline__ void measure_chain(const MeasureArgs<R> &a, long long c, const R (&x)[NR + 2 * NC], const Source &src,
                                              const Field<R> &fmean, const Field<R> &fobs, const TiledField<R> &fcov,
                                              const TiledField<R> &ffac, [[maybe_unused]] R (*s_delta)[kStepThreads],
                                              WidthFn &&widths, bool &bad_pivot) {
  constexpr int D = NR + 2 * NC;
  constexpr int PR = NR * (NR + 1) / 2;
  constexpr int P = PR + NC * NC;
  constexpr int NOBS = 2 * NR + NC;
  constexpr bool STREAM = PER_CHAIN_COV && P > kMaxPackedInRegisters;
  using N_ = Num<R>;
  const unsigned int coff = (unsigned int)c * (unsigned int)sizeof(R);
  [[maybe_unused]] const unsigned int toff = tiled_offset<R>(c, P);      // the chain's place in the packed fields
  // Every phase issues ALL its loads before its first store: the compiler cannot prove that a store to one field does
  // not alias the next load of another, so interleaved load-update-store sequences were emitted strictly in order
  // with two loads in flight per wavefront (k_measure<64,0> ran at 0.36 of the HBM peak that way).
  R delta[D];
  {
    R mu[D];
#pragma unroll
    for (int d = 0; d < D; ++d) mu[d] = src.mean(d);
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if constexpr (STREAM) s_delta[d][threadIdx.x] = x[d] - mu[d];   // parked in LDS, see the streaming path below
      else delta[d] = x[d] - mu[d];
      packed_store<NTM>(fmean, d, coff, fma_(x[d], a.inv_i, mu[d] * a.keep));   // :404-410
    }
  }
  // observables [|x_r|, |z_c|, x_r^2] and their running mean (:458-463, :412-414), in batches of kBatch
  constexpr int kBatch = 32;
#pragma unroll
  for (int k0 = 0; k0 < NOBS; k0 += kBatch) {
    R m[kBatch];
#pragma unroll
    for (int u = 0; u < kBatch; ++u)
      if (k0 + u < NOBS) m[u] = src.obs(k0 + u);
#pragma unroll
    for (int u = 0; u < kBatch; ++u) {
      const int k = k0 + u;
      if (k < NOBS) {
        R o;
        if (k < NR) o = N_::abs_(x[k]);
        else if (k < NR + NC) o = N_::sqrt_(fma_(x[k], x[k], x[k + NC] * x[k + NC]));
        else o = x[k - NR - NC] * x[k - NR - NC];
        packed_store<NTM>(fobs, k, coff, fma_(o, a.inv_i, m[u] * a.keep));
      }
    }
  }
  if constexpr (PER_CHAIN_COV) {
    if (a.update_cov) {
      // :418, :425 -- each block uses its own group's width; they coincide unless group steps made them differ
      R w_real, w_cplx;
      widths(w_real, w_cplx);
      // the epsilon term sigma^2 / i (:418, :425) goes into the diagonal through an explicit fma: left as `v += w * w * inv_i`
      // the compiler may or may not fuse the product into the add, kernel by kernel
      const R w2_real = w_real * w_real, w2_cplx = w_cplx * w_cplx;
      if constexpr (STREAM) {
        // delta is parked in LDS (lane-linear, conflict-free) so that the walk over the packed entries can be a
        // ROLLED loop: unrolled, 2 080 entries are ~100 KB of code and the kernel becomes instruction-fetch bound.
        // One 64-bit pointer per lane steps by whole rows (the packed order is exactly the loop order).
        // tile-major: the chain's entries lie 64 values apart from the start of its tile
        R *p = a.cov + (c >> 6) * (long long)P * 64 + (c & 63);
        constexpr long long ts = 64;
        for (int i = 0; i < NR; ++i) {
          const R di = s_delta[i][threadIdx.x];
          int j = 0;
          // batches of 16 (then 4) entries: all loads first (a store to p[.] would otherwise fence the next load, the compiler
          // cannot prove the rows distinct), then the updates
          for (; j + 16 <= i; j += 16) {
            R v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = p[u * ts];
#pragma unroll
            for (int u = 0; u < 16; ++u) p[u * ts] = fma_(di * s_delta[j + u][threadIdx.x], a.inv_i, v[u] * a.cov_keep);
            p += 16 * ts;
          }
          for (; j + 4 <= i; j += 4) {
            R v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = p[u * ts];
#pragma unroll
            for (int u = 0; u < 4; ++u) p[u * ts] = fma_(di * s_delta[j + u][threadIdx.x], a.inv_i, v[u] * a.cov_keep);
            p += 4 * ts;
          }
          for (; j < i; ++j) {
            *p = fma_(di * s_delta[j][threadIdx.x], a.inv_i, *p * a.cov_keep);
            p += ts;
          }
          *p = fma_(w2_real, a.inv_i, fma_(di * di, a.inv_i, *p * a.cov_keep));
          p += ts;
        }
        for (int i = 0; i < NC; ++i) {
          const R ai = s_delta[NR + i][threadIdx.x], bi = s_delta[NR + NC + i][threadIdx.x];
          int j = 0;
          for (; j + 8 <= i; j += 8) {          // eight (Re, Im) pairs: sixteen loads, then the updates
            R v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = p[u * ts];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const R aj = s_delta[NR + j + u][threadIdx.x], bj = s_delta[NR + NC + j + u][threadIdx.x];
              p[(2 * u) * ts] = fma_(fma_(ai, aj, bi * bj), a.inv_i, v[2 * u] * a.cov_keep);
              p[(2 * u + 1) * ts] = fma_(fma_(bi, aj, -(ai * bj)), a.inv_i, v[2 * u + 1] * a.cov_keep);
            }
            p += 16 * ts;
          }
          for (; j < i; ++j) {
            const R aj = s_delta[NR + j][threadIdx.x], bj = s_delta[NR + NC + j][threadIdx.x];
            const R re = p[0], im = p[ts];
            p[0] = fma_(fma_(ai, aj, bi * bj), a.inv_i, re * a.cov_keep);
            p[ts] = fma_(fma_(bi, aj, -(ai * bj)), a.inv_i, im * a.cov_keep);
            p += 2 * ts;
          }
          *p = fma_(w2_cplx, a.inv_i, fma_(fma_(ai, ai, bi * bi), a.inv_i, *p * a.cov_keep));
          p += ts;
        }
      } else {
      // the packed matrix is read whole, then updated and written back (FUSED keeps it for the Cholesky anyway)
      R m[P];
#pragma unroll
      for (int k = 0; k < P; ++k) m[k] = src.cov(k);
#pragma unroll
      for (int i = 0; i < NR; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
          const int k = tri(i, j);
          R v = fma_(delta[i] * delta[j], a.inv_i, m[k] * a.cov_keep);
          if (i == j) v = fma_(w2_real, a.inv_i, v);
          packed_store<NT>(fcov, k, toff, v);
          m[k] = v;
        }
#pragma unroll
      for (int i = 0; i < NC; ++i) {
        const R ai = delta[NR + i], bi = delta[NR + NC + i];
#pragma unroll
        for (int j = 0; j < i; ++j) {
          const R aj = delta[NR + j], bj = delta[NR + NC + j];
          const int kr = cre(PR, i, j), ki = cim(PR, i, j);
          const R vr = fma_(fma_(ai, aj, bi * bj), a.inv_i, m[kr] * a.cov_keep);
          const R vi = fma_(fma_(bi, aj, -(ai * bj)), a.inv_i, m[ki] * a.cov_keep);
          packed_store<NT>(fcov, kr, toff, vr);
          packed_store<NT>(fcov, ki, toff, vi);
          m[kr] = vr;
          m[ki] = -vi;   // the proposals use conj(K) (quirk Q3, :292-298)
        }
        const int kd = cdiag(PR, i);
        const R vd = fma_(w2_cplx, a.inv_i, fma_(fma_(ai, ai, bi * bi), a.inv_i, m[kd] * a.cov_keep));
        packed_store<NT>(fcov, kd, toff, vd);
        m[kd] = vd;
      }
      if constexpr (FUSED) {
        if (a.write_factor) {
          cholesky_packed<R, NR, NC>(m, bad_pivot);
#pragma unroll
          for (int k = 0; k < P; ++k) packed_store<NT>(ffac, k, toff, m[k]);
        }
      }
      }   // !STREAM
    }
  }
}

template <typename R, int NR, int NC, bool PER_CHAIN_COV, bool FUSED, bool NT, bool NTM>
__global__ void __launch_bounds__(kStepThreads, ME_MEASURE_WAVES) k_measure(MeasureArgs<R> a) {
  constexpr int D = NR + 2 * NC;
  constexpr int P = NR * (NR + 1) / 2 + NC * NC;
  constexpr int NOBS = 2 * NR + NC;
  bool bad_pivot = false;
  const long long stride = (long long)gridDim.x * kStepThreads;
  const XField<R, D> fx(a.x, a.n);
  const Field<R> fw(a.width, a.n, (NR > 0 && NC > 0) ? 3 : 1), fmean(a.mean, a.n, D), fobs(a.obs_mean, a.n, NOBS);
  // Large matrices (P > kMaxPackedInRegisters) are statistics only (no factors, FUSED = false) and take the streaming
  // path below; their field may pass the 4 GiB one descriptor spans and is walked with a 64-bit pointer instead.
  constexpr bool STREAM = PER_CHAIN_COV && P > kMaxPackedInRegisters;
  const TiledField<R> fcov(a.cov, a.n, (PER_CHAIN_COV && !STREAM) ? P : 0);
  const TiledField<R> ffac(a.factor, a.n, (PER_CHAIN_COV && FUSED) ? P : 0);
  __shared__ R s_delta[STREAM ? D : 1][kStepThreads];
  for (long long c = (long long)blockIdx.x * kStepThreads + threadIdx.x; c < a.n; c += stride) {
    const unsigned int coff = (unsigned int)c * (unsigned int)sizeof(R);
    const unsigned int xoff = fx.offset(c);
    R x[D];
#pragma unroll
    for (int d = 0; d < D; ++d) x[d] = fx.load(d, xoff);
    const MeasureFromMemory<R, NT, NTM> src{fmean, fobs, fcov, coff, tiled_offset<R>(c, P)};
    measure_chain<R, NR, NC, PER_CHAIN_COV, FUSED, NT, NTM>(a, c, x, src, fmean, fobs, fcov, ffac, s_delta, [&](R &w_real, R &w_cplx) {
      constexpr bool MIXED = NR > 0 && NC > 0;
      w_real = fw.load((MIXED && a.split_widths) ? 1 : 0, coff);
      w_cplx = fw.load((MIXED && a.split_widths) ? 2 : 0, coff);
    }, bad_pivot);
  }
  if (bad_pivot) atomicOr(a.status, (unsigned int)ST_BAD_PIVOT);
}

// ------------------------------------------------------------------------------------------------ k_cycle
// One cycle of the reference's driver loop (README.md:41-44: k x step_all(), then measure()) in ONE launch: the chain's
// state, energy and width are loaded once, the n_sweeps sweeps run in registers (run_sweeps, exactly k_step's), the
// running mean / observables / covariance are updated from the registers and the proposal factors refreshed
// (measure_chain, exactly k_measure's), and everything is stored once.  Against n_sweeps one-sweep launches plus a measure
// launch this removes n_sweeps - 1 round trips of the state and the measure's re-read of x and the width; the results are
// those of `me_step(e, n_sweeps); me_measure(e)` (same code, same order of operations).
// CK: CK_IDENTITY / CK_SHARED (before the 50th measure, or cov_mode fixed / pooled) or CK_PER_CHAIN (each chain's own
// factor, read once per launch).  NT / NTM: cache policy of the packed fields / the running means, as in k_measure; NT
// also covers the factor read of the sweeps.  Packed matrices up to kMaxPackedInRegisters entries.
// Measured (profiles/r03_cycle_prefetch_variants.txt, 10 sweeps per cycle, per-chain factors live): level 0 / 1 / 2 at (4,4) x
// 2^20 chains 150 / 149 / 155 us (float32), 353 / 354 / 499 us (float64); at (2,7) x 2^18 chains 70 / 70 / 70 us, 185 / 192 /
// 187 us -- with ten sweeps between the state loads and the measure half the launch is bound by the sweeps' instructions,
// and the registers the preloaded values occupy cost more than their latency.  Default: off.
#ifndef ME_CYCLE_PREFETCH
#define ME_CYCLE_PREFETCH 0       // 0 / 1 / 2 (experiments)
#endif
constexpr int cycle_prefetch_level(int, int, int, int) { return ME_CYCLE_PREFETCH; }

template <typename R, int NR, int NC, class Energy, int CK, bool NT, bool NTM>
__global__ void ME_STEP_BOUNDS k_cycle(StepArgs<R> a, MeasureArgs<R> ma, Energy en) {
  constexpr int D = NR + 2 * NC;
  constexpr int P = NR * (NR + 1) / 2 + NC * NC;
  constexpr int NOBS = 2 * NR + NC;
  constexpr bool MIXED = NR > 0 && NC > 0;
  static_assert(P <= kMaxPackedInRegisters, "k_cycle keeps the packed matrix in registers");
  static_assert(CK == CK_IDENTITY || CK == CK_SHARED || CK == CK_PER_CHAIN, "NT selects the non-temporal factor read");
  constexpr int CKX = CK == CK_PER_CHAIN ? (NT ? CK_PER_CHAIN_NT : CK_PER_CHAIN) : CK;
  using N_ = Num<R>;
  N_::prepare();
  energy_prepare(en, 0);
  unsigned int wave_accepted = 0;
  bool bad_energy = false, bad_width = false, bad_pivot = false;
  const long long stride = (long long)gridDim.x * blockDim.x;
  using Ledger = EnergyLedger<R, Energy, GROUP_ALL>;
  const XField<R, D> fx(a.x, a.n);
  const Field<R> fe(a.energy, a.n, Ledger::T + (MIXED ? 1 : 0)), fw(a.width, a.n, MIXED ? 3 : 1);
  const Field<R> fmean(ma.mean, a.n, D), fobs(ma.obs_mean, a.n, NOBS);
  const TiledField<R> fcov(ma.cov, a.n, P), ffac(ma.factor, a.n, P);
  const bool stale_total = MIXED && a.stale_total != 0;
  for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < a.n; c += stride) {
    const unsigned int coff = (unsigned int)c * (unsigned int)sizeof(R);
    const unsigned int xoff = fx.offset(c);
    R x[D];
#pragma unroll
    for (int d = 0; d < D; ++d) x[d] = fx.load(d, xoff);
    Ledger ledger;
    R total_q5 = R(0);
    if (stale_total) total_q5 = fe.load(Ledger::T, coff);
    else ledger.load(fe, coff);
    R w = fw.load(0, coff);
    R w_r = w, w_c = w;
    if constexpr (MIXED) {
      if (a.split_widths) {
        w_r = fw.load(GROUP_REAL, coff);
        w_c = fw.load(GROUP_COMPLEX, coff);
      }
    }
    const unsigned long long gid = a.chain_offset + (unsigned long long)c;
    [[maybe_unused]] const unsigned int toff = tiled_offset<R>(c, P);
    auto fac = [&](int k) -> R {
      if constexpr (CK == CK_PER_CHAIN) return packed_load<NT>(ffac, k, toff);
      else return a.factor[k];
    };
    // what the measure half needs from memory is requested BEFORE the sweeps (cycle_prefetch_level: 1 = mean and observables,
    // 2 = the packed covariance too): the loads travel while the sweeps run instead of after them
    constexpr int kPrefetch = cycle_prefetch_level(D, NOBS, P, (int)sizeof(R));
    MeasurePreloaded<R, D, NOBS, P, kPrefetch, NT> pre{{}, {}, {}, fcov, toff};
    if constexpr (kPrefetch >= 1) {
#pragma unroll
      for (int d = 0; d < D; ++d) pre.mu[d] = packed_load<NTM>(fmean, d, coff);
#pragma unroll
      for (int k = 0; k < NOBS; ++k) pre.ob[k] = packed_load<NTM>(fobs, k, coff);
    }
    if constexpr (kPrefetch >= 2) {
      if (ma.update_cov) {
#pragma unroll
        for (int k = 0; k < P; ++k) pre.cv[k] = packed_load<NT>(fcov, k, toff);
      }
    }
    run_sweeps<R, NR, NC, Energy, CKX, false, GROUP_ALL>(a, en, c, gid, stale_total, x, ledger, total_q5, w, w_r, w_c, fac,
                                                         wave_accepted, bad_energy);
    bad_width |= !(w > R(0));
    // the state goes out first: a block of stores in front of measure_chain's loads costs nothing, and x dies as soon as the
    // means and observables are formed
#pragma unroll
    for (int d = 0; d < D; ++d) fx.store(d, xoff, x[d]);
    if (stale_total) fe.store(Ledger::T, coff, total_q5);
    else ledger.store(fe, coff);
    fw.store(0, coff, w);
    // after a step_all both group widths equal the shared width (:436-437): that is what the epsilon terms use
    if constexpr (kPrefetch >= 1) {
      measure_chain<R, NR, NC, true, true, NT, NTM>(ma, c, x, pre, fmean, fobs, fcov, ffac, nullptr,
                                                    [&](R &w_real, R &w_cplx) { w_real = w_cplx = w; }, bad_pivot);
    } else {
      const MeasureFromMemory<R, NT, NTM> src{fmean, fobs, fcov, coff, toff};
      measure_chain<R, NR, NC, true, true, NT, NTM>(ma, c, x, src, fmean, fobs, fcov, ffac, nullptr,
                                                    [&](R &w_real, R &w_cplx) { w_real = w_cplx = w; }, bad_pivot);
    }
  }
  if ((threadIdx.x & 63) == 0 && wave_accepted) {
    unsigned long long *slot = a.accept_slots + (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    *slot += (unsigned long long)wave_accepted;
  }
  const unsigned int bits = (bad_energy ? ST_NONFINITE_ENERGY : 0u) | (bad_width ? ST_BAD_WIDTH : 0u) | (bad_pivot ? ST_BAD_PIVOT : 0u);
  if (bits) atomicOr(a.status, bits);
}

// Refresh the per-chain proposal factors from the per-chain covariance: factor = chol(C_r), chol(conj(K)).
template <typename R, int NR, int NC, bool NT>
__global__ void __launch_bounds__(kStepThreads) k_factor(const R *cov, R *factor, unsigned int *status, long long n) {
  constexpr int PR = NR * (NR + 1) / 2;
  constexpr int P = PR + NC * NC;
  bool bad_pivot = false;
  const long long stride = (long long)gridDim.x * kStepThreads;
  const TiledField<R> fcov(cov, n, P), ffac(factor, n, P);
  for (long long c = (long long)blockIdx.x * kStepThreads + threadIdx.x; c < n; c += stride) {
    const unsigned int toff = tiled_offset<R>(c, P);
    R m[P];
#pragma unroll
    for (int k = 0; k < P; ++k) m[k] = packed_load<NT>(fcov, k, toff);
#pragma unroll
    for (int i = 0; i < NC; ++i)
#pragma unroll
      for (int j = 0; j < i; ++j) m[cim(PR, i, j)] = -m[cim(PR, i, j)];   // conj(K) (quirk Q3, :292-298)
    cholesky_packed<R, NR, NC>(m, bad_pivot);
#pragma unroll
    for (int k = 0; k < P; ++k) packed_store<NT>(ffac, k, toff, m[k]);
  }
  if (bad_pivot) atomicOr(status, (unsigned int)ST_BAD_PIVOT);
}

// Per-chain Cholesky factor of a covariance matrix too large for registers (more than 160 packed entries; pure real
// spaces): factor = chol(C), row by row (Cholesky-Banachiewicz), each lane its own chain, everything through global
// memory in the tile-major layout -- L_ij = (C_ij - sum_{k<j} L_ik L_jk) / L_jj.  The row being built lives in LDS
// ([NR][64 lanes], lane-linear); the finished rows are re-read from the factor field itself, four rows at a time in
// batches of 4 x 16 loads issued ahead of their multiply-adds.  ROWS rows are built together so that every finished L_jk that is loaded serves
// ROWS dot products: the traffic is NR^3 / (6 ROWS) loads per chain (64 parameters, ROWS = 4: 11 k loads = 44 KB in
// float32).  Slow by construction -- an order of magnitude above a measure() with the pooled shape -- and there only
// so that cov_mode="reference" keeps the reference's semantics (metropolis_engine.py:416-421 feeding :268-270) at any
// size.  The fields may pass 4 GiB: 64-bit pointers.
template <typename R, int NR, bool NT>
__global__ void __launch_bounds__(kStepThreads) k_factor_stream(const R *cov, R *factor, unsigned int *status, long long n) {
  constexpr int P = NR * (NR + 1) / 2;
  constexpr int ROWS = sizeof(R) == 4 ? 4 : 2;                  // LDS: ROWS x NR x 64 values (64 KiB at NR = 64)
  using N_ = Num<R>;
  __shared__ R rows[ROWS][NR][kStepThreads];
  bool bad_pivot = false;
  const long long stride = (long long)gridDim.x * kStepThreads;
  const int lane = threadIdx.x;
  for (long long c = (long long)blockIdx.x * kStepThreads + threadIdx.x; c < n; c += stride) {
    const long long base = (c >> 6) * (long long)P * 64 + (c & 63);
    const R *cv = cov + base;
    R *fc = factor + base;
    for (int i0 = 0; i0 < NR; i0 += ROWS) {
      const int nrows = NR - i0 < ROWS ? NR - i0 : ROWS;
      // the covariance rows of the block into LDS
      for (int r = 0; r < nrows; ++r) {
        const R *src = cv + (long long)tri(i0 + r, 0) * 64;
        int j = 0;
        for (; j + 16 <= i0 + r + 1; j += 16) {
          R v[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) v[u] = NT ? __builtin_nontemporal_load(src + (j + u) * 64) : src[(j + u) * 64];
#pragma unroll
          for (int u = 0; u < 16; ++u) rows[r][j + u][lane] = v[u];
        }
        for (; j <= i0 + r; ++j) rows[r][j][lane] = NT ? __builtin_nontemporal_load(src + j * 64) : src[j * 64];
      }
      // Columns left of the block: every finished row j < i0 serves all rows of the block.  The walk is bound by memory
      // LATENCY (every batch of loads a round trip, and column j needs column j - 1 of the block's rows): FOUR finished rows
      // are fetched together -- their first j entries in batches of 4 x 16 loads, then the ten entries of the little triangle
      // between them -- and the four columns are finished one after the other from registers (round 3; one row per visit
      // with two or three dependent round trips each took twice as long: 8.0 -> 4.5 ms per measure at 100 parameters x 2^14
      // chains in float64, tools/dev/time_compiled_vs_runtime.py).
      int j = 0;
      for (; j + 4 <= i0; j += 4) {
        const R *lj[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) lj[q] = fc + (long long)tri(j + q, 0) * 64;
        R s[4][ROWS];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int r = 0; r < ROWS; ++r) s[q][r] = R(0);
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
            for (int r = 0; r < ROWS; ++r) v[r] = rows[r][k + u][lane];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
              for (int r = 0; r < ROWS; ++r) s[q][r] += v[r] * f[q][u];
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
            for (int r = 0; r < ROWS; ++r) v[r] = rows[r][k + u][lane];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
              for (int r = 0; r < ROWS; ++r) s[q][r] += v[r] * f[q][u];
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
          for (int r = 0; r < ROWS; ++r) {
            R acc = s[q][r];
#pragma unroll
            for (int p = 0; p < q; ++p) acc += rows[r][j + p][lane] * t[q][p];
            rows[r][j + q][lane] = (rows[r][j + q][lane] - acc) * inv;
          }
        }
      }
      for (; j < i0; ++j) {                                 // (at most three rows left)
        const R *lj = fc + (long long)tri(j, 0) * 64;
        R s[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) s[r] = R(0);
        int k = 0;
        for (; k + 8 <= j; k += 8) {
          R f[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) f[u] = lj[(k + u) * 64];
#pragma unroll
          for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int r = 0; r < ROWS; ++r) s[r] += rows[r][k + u][lane] * f[u];
        }
        for (; k < j; ++k) {
          const R f = lj[k * 64];
#pragma unroll
          for (int r = 0; r < ROWS; ++r) s[r] += rows[r][k][lane] * f;
        }
        const R inv = R(1) / lj[j * 64];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) rows[r][j][lane] = (rows[r][j][lane] - s[r]) * inv;
      }
      // the triangle inside the block: rows depend on each other here, everything is in LDS
      for (int r = 0; r < nrows; ++r) {
        const int i = i0 + r;
        for (int j = i0; j < i; ++j) {
          const int rj = j - i0;
          R s = R(0);
          for (int k = 0; k < j; ++k) s += rows[r][k][lane] * rows[rj][k][lane];
          rows[r][j][lane] = (rows[r][j][lane] - s) / rows[rj][j][lane];
        }
        R s = rows[r][i][lane];
        for (int k = 0; k < i; ++k) s -= rows[r][k][lane] * rows[r][k][lane];
        if (!(s > R(0))) { bad_pivot = true; s = R(1e-30); }
        rows[r][i][lane] = N_::sqrt_(s);
      }
      // finished rows out
      for (int r = 0; r < nrows; ++r) {
        R *dst = fc + (long long)tri(i0 + r, 0) * 64;
        for (int j = 0; j <= i0 + r; ++j) {
          if constexpr (NT) __builtin_nontemporal_store(rows[r][j][lane], dst + j * 64);
          else dst[j * 64] = rows[r][j][lane];
        }
      }
      // the next block reads these rows back through global memory from this same lane: program order suffices
    }
  }
  if (bad_pivot) atomicOr(status, (unsigned int)ST_BAD_PIVOT);
}

// Per-chain Cholesky factors of a packed matrix too large for registers that has a COMPLEX block (mixed and pure complex
// spaces beyond 160 packed entries, e.g. 1 real + 13 complex = 170): factor = [chol(C_r) | chol(conj K)] (quirk Q3,
// metropolis_engine.py:292-298), row by row (Cholesky-Banachiewicz), each lane its own chain, every operand through global
// memory in the tile-major layout -- a finished L_ik is re-read from the factor field the lane itself wrote (program order
// suffices).  Written for correctness, not speed (a dependent load per multiply-add): it exists so that
// cov_mode="reference" keeps the reference's semantics (:416-427 feeding :274-302) for such spaces at all; the pure real
// sizes have k_factor_tile / k_factor_stream.
template <typename R, int NR, int NC, bool NT>
__global__ void __launch_bounds__(kStepThreads) k_factor_mixed(const R *cov, R *factor, unsigned int *status, long long n) {
  constexpr int PR = NR * (NR + 1) / 2, P = PR + NC * NC;
  using N_ = Num<R>;
  bool bad_pivot = false;
  const long long stride = (long long)gridDim.x * kStepThreads;
  for (long long c = (long long)blockIdx.x * kStepThreads + threadIdx.x; c < n; c += stride) {
    const long long base = (c >> 6) * (long long)P * 64 + (c & 63);
    const R *cv = cov + base;
    R *fc = factor + base;
    auto in = [&](int k) -> R { return NT ? __builtin_nontemporal_load(cv + (long long)k * 64) : cv[(long long)k * 64]; };
    for (int i = 0; i < NR; ++i)
      for (int j = 0; j <= i; ++j) {
        R s = in(tri(i, j));
        for (int k = 0; k < j; ++k) s = fma_(-fc[(long long)tri(i, k) * 64], fc[(long long)tri(j, k) * 64], s);
        if (j < i) {
          fc[(long long)tri(i, j) * 64] = s / fc[(long long)tri(j, j) * 64];
        } else {
          if (!(s > R(0))) { bad_pivot = true; s = R(1e-30); }
          fc[(long long)tri(i, i) * 64] = N_::sqrt_(s);
        }
      }
    for (int i = 0; i < NC; ++i)
      for (int j = 0; j <= i; ++j) {
        R sr = j < i ? in(cre(PR, i, j)) : in(cdiag(PR, i));
        R si = j < i ? -in(cim(PR, i, j)) : R(0);                 // conj(K)
        for (int k = 0; k < j; ++k) {                              // s -= L_ik conj(L_jk)
          const R ar = fc[(long long)cre(PR, i, k) * 64], ai = fc[(long long)cim(PR, i, k) * 64];
          const R br = fc[(long long)cre(PR, j, k) * 64], bi = fc[(long long)cim(PR, j, k) * 64];
          sr = fma_(-ai, bi, fma_(-ar, br, sr));
          si = fma_(ar, bi, fma_(-ai, br, si));
        }
        if (j < i) {
          const R d = fc[(long long)cdiag(PR, j) * 64];
          fc[(long long)cre(PR, i, j) * 64] = sr / d;
          fc[(long long)cim(PR, i, j) * 64] = si / d;
        } else {
          if (!(sr > R(0))) { bad_pivot = true; sr = R(1e-30); }
          fc[(long long)cdiag(PR, i) * 64] = N_::sqrt_(sr);
        }
      }
  }
  if (bad_pivot) atomicOr(status, (unsigned int)ST_BAD_PIVOT);
}

}  // namespace me
