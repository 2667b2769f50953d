// Per-chain Cholesky factors of covariance matrices too large for one lane's registers (more than 160 packed entries; pure
// real spaces, e.g. BASELINE config 4's 64 parameters = 2 080 entries per chain): factor = chol(C), the refresh that
// cov_mode="reference" needs after every measure (metropolis_engine.py:416-421 feeding :268-270).
//
// k_factor_stream (me_device.h) keeps the one-lane-per-chain mapping of every other kernel: a lane cannot hold its matrix, so
// finished rows are re-read from global memory NR^3/(6 ROWS) times -- five to ten times the matrix itself -- and 64 KiB of
// LDS per wavefront leaves two wavefronts per CU to hide that latency (2^19 chains x 64 parameters: 20 / 40 ms).
//
// Here a LANE GROUP owns a chain and a lane owns a ROW, so a matrix is read once and written once:
//
//   load     a workgroup takes kChains neighbouring chains of one 64-chain tile.  In the tile-major layout
//            ([tile][entry][64 chains], me_device.h) that is a run of kChains values per entry -- 64 bytes, one memory
//            sector -- which the whole workgroup copies into LDS chain-major (the transpose happens in this copy), each
//            matrix packed COLUMN by column there: element (i, j) at col(j) + i - j, so that the lanes of a group -- one
//            row each -- touch consecutive words whenever they read or write one column (no bank conflicts);
//   factor   each lane group reads its chain's rows out of LDS into registers (lane i: row i, NR values) and runs the
//            column-by-column recurrence  L_ij = (C_ij - sum_{k<j} L_ik L_jk) / L_jj:  L_ik is the lane's own register k
//            (compile-time index, the loops are straight-line code), L_jk is row j in LDS -- every lane of the group reads
//            the SAME address, a broadcast -- and column j goes back to LDS as soon as it is final, which is what makes
//            row j complete by the time it is broadcast (LDS instructions of a wavefront execute in order: no waiting
//            between the write and the reads).  Two accumulators per row break the dependent chain of multiply-adds;
//            the pivot travels by v_readlane / ds_bpermute; sqrt and reciprocal come from ONE v_rsq and a handful of
//            fmas (sqrt_and_inverse) -- the IEEE sqrt + division they replace cost as much as the multiply-adds;
//   store    the finished factors leave LDS the way the matrices came in.
//
// Pivots and column entries follow cholesky_packed (me_device.h): s = C_jj - sum L_jk^2, L_jj = sqrt(s), L_ij = t * (1 / L_jj);
// a non-positive pivot is flagged and clamped.  The k-sum runs even and odd k apart and sqrt / reciprocal are rounded
// to about an ulp rather than correctly, so results agree with k_factor_stream to rounding.
//
// Measured (MI355X, 2^19 chains x 64 parameters, tools/dev/time_factor_tile.py; -DME_FACTOR_TILE_SKIP_FACTOR / _SKIP_COPY
// time the phases apart): float64 13 ms = 4.8 ms copy (17 GB, 3.6 TB/s) + 8 ms factor; float32 8.4 = 3.0 + 5.4.  The
// factor phase is LDS-bandwidth bound: a broadcast read still moves 64 lanes x 8 bytes, 2 016 of them per matrix are
// 2 MB of LDS traffic per chain, 6.8 ms at 128 bytes per clock and CU.  Two rows per lane and two chains per wavefront
// halve the broadcasts per matrix: float32 measure() 10.4 -> 7.8 ms; float64 would need 2 x 128 registers for the rows
// alone and loses more to occupancy than it gains (16.3 -> 19.8 ms), so it keeps one row per lane.
// Tried and dropped: requesting the NEXT unit's entries into registers before the factorization (33 doubles per thread;
// 256 registers, no scratch): float64 measure() 14.8 -> 16.9 ms; folding the covariance recursion into the load phase
// (one pass over C fewer): 14.8 -> 15.1 ms.
// History: LDS row-major with the pivot through LDS and IEEE sqrt + division 17 ms (float64); column-major, readlane
// pivot, sqrt_and_inverse 10 ms; XCD-aware order: -15 % where a run is half a sector (32 bytes).
#pragma once
#include "me_device.h"
#include "me_per_device.h"

namespace me {

// bytes of one entry's run of neighbouring chains that a workgroup copies (= chains per workgroup x sizeof(R))
#ifndef ME_FACTOR_TILE_RUN_BYTES_F32
#define ME_FACTOR_TILE_RUN_BYTES_F32 64
#endif
#ifndef ME_FACTOR_TILE_RUN_BYTES_F64
#define ME_FACTOR_TILE_RUN_BYTES_F64 64
#endif
// rows per lane for more than 32 parameters.  Measured at 64 parameters x 2^19 chains, whole measure(): float32 10.4 ms with
// one row, 7.8 ms with two; float64 16.3 ms with one, 19.8 ms with two (2 x 64 doubles = 256 registers per lane leave one
// wavefront per SIMD, and the latencies of the column recurrence are no longer hidden)
// every N-th pair of broadcasts through v_readlane instead of LDS (0: none); one matrix per wavefront, one row per lane only.
// float64, whole measure() at 64 parameters x 2^19 chains: none 16.35 ms, every pair 15.33, every 2nd 14.96, every 3rd 14.78
#ifndef ME_FACTOR_TILE_READLANE_F32
#define ME_FACTOR_TILE_READLANE_F32 0
#endif
#ifndef ME_FACTOR_TILE_READLANE_F64
#define ME_FACTOR_TILE_READLANE_F64 3
#endif
#ifndef ME_FACTOR_TILE_ROWS_PER_LANE_F32
#define ME_FACTOR_TILE_ROWS_PER_LANE_F32 2
#endif
#ifndef ME_FACTOR_TILE_ROWS_PER_LANE_F64
#define ME_FACTOR_TILE_ROWS_PER_LANE_F64 1
#endif

template <typename R, int NR>
struct FactorTile {
  static constexpr int P = NR * (NR + 1) / 2;
  static constexpr int PS = P | 1;                                    // odd chain stride in LDS: the transposing copy spreads over the banks
  // rows per lane: with two, a broadcast of L_jk serves two rows of each of two chains -- half the LDS traffic per matrix --
  // and the upper row block drops out of the multiply-adds once j passes it (2 512 instead of 4 032 per pair of matrices)
  static constexpr int RPL = (NR > 32 && (sizeof(R) == 4 ? ME_FACTOR_TILE_ROWS_PER_LANE_F32 : ME_FACTOR_TILE_ROWS_PER_LANE_F64) == 2) ? 2 : 1;
  static constexpr int kRowsPerGroup = (NR + RPL - 1) / RPL;
  static constexpr int LPC = kRowsPerGroup <= 16 ? 16 : kRowsPerGroup <= 32 ? 32 : 64;   // lanes per chain (a power of two)
  static constexpr int CPW = 64 / LPC;                                // chains per wavefront
  // chains per workgroup: a run of ME_FACTOR_TILE_RUN_BYTES per entry, as long as the matrices fit in LDS
  static constexpr int kWanted = (sizeof(R) == 4 ? ME_FACTOR_TILE_RUN_BYTES_F32 : ME_FACTOR_TILE_RUN_BYTES_F64) / (int)sizeof(R);
  static constexpr int kFit = (int)(150 * 1024 / (PS * sizeof(R)));
  static constexpr int kChains = kFit >= kWanted ? kWanted : kFit >= 8 ? 8 : kFit >= 4 ? 4 : kFit >= 2 ? 2 : 1;
  // at most 8 wavefronts per workgroup (256 registers per lane: a row is NR of them); more chains than that take turns
  static constexpr int kPasses = kChains / CPW > 8 ? kChains / CPW / 8 : 1;
  static constexpr int kWaves = kChains / CPW / kPasses > 0 ? kChains / CPW / kPasses : 1;
  static constexpr int kThreads = 64 * kWaves;
  static constexpr size_t kLdsBytes = (size_t)kChains * PS * sizeof(R);
  static_assert(NR <= 64, "one row per lane: up to 64 parameters");
  static_assert(kChains >= CPW && kThreads <= 1024, "workgroup shape");
};

// dg = sqrt(d), inv = 1 / sqrt(d) for a normal positive d: hardware estimate, one coupled Newton step for both (g -> sqrt,
// h -> 1/(2 sqrt)), a residual correction each
// (fma_: me_device.h)
template <typename R>
__device__ __forceinline__ void sqrt_and_inverse(R d, R &dg, R &inv) {
  R rs;
  if constexpr (sizeof(R) == 4) rs = __builtin_amdgcn_rsqf(d);
  else rs = __builtin_amdgcn_rsq(d);
  R g = d * rs, h = R(0.5) * rs;
  const R c = fma_(-h, g, R(0.5));
  g = fma_(g, c, g);
  h = fma_(h, c, h);
  dg = fma_(fma_(-g, g, d), h, g);
  const R h2 = h + h;
  inv = fma_(fma_(-dg, h2, R(1)), h2, h2);
}

// value of lane `src` (of the 64) for every lane
template <typename R>
__device__ __forceinline__ R lane_value(R v, int src, bool uniform_src) {
  if constexpr (sizeof(R) == 4) {
    const int w = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(R, uniform_src ? __builtin_amdgcn_readlane(w, src) : __builtin_amdgcn_ds_bpermute(src << 2, w));
  } else {
    const unsigned long long w = __builtin_bit_cast(unsigned long long, v);
    const int lo = (int)(unsigned int)w, hi = (int)(unsigned int)(w >> 32);
    const unsigned int rlo = (unsigned int)(uniform_src ? __builtin_amdgcn_readlane(lo, src) : __builtin_amdgcn_ds_bpermute(src << 2, lo));
    const unsigned int rhi = (unsigned int)(uniform_src ? __builtin_amdgcn_readlane(hi, src) : __builtin_amdgcn_ds_bpermute(src << 2, hi));
    return __builtin_bit_cast(R, ((unsigned long long)rhi << 32) | rlo);
  }
}

template <typename R, int NR, bool NT>
__global__ void __launch_bounds__((FactorTile<R, NR>::kThreads))
k_factor_tile(const R *cov, R *factor, unsigned int *status, long long n) {
  using T = FactorTile<R, NR>;
  constexpr int P = T::P, PS = T::PS, G = T::kChains, LPC = T::LPC, RPL = T::RPL;
  constexpr int kGroupsPerTile = 64 / G;
  constexpr int kRowsPerPass = T::kThreads / G;      // entries a copy pass of the workgroup covers
  extern __shared__ unsigned char lds_raw[];
  R *lds = reinterpret_cast<R *>(lds_raw);
  const int t = threadIdx.x;
  const int copy_chain = t % G, copy_row = t / G;
  const int lane = t & 63, wave = t >> 6;
  const int my_row = lane % LPC;                     // the rows of its chain this lane works on: my_row + r LPC, r < RPL
  bool bad_pivot = false;
  auto col = [](int j) { return j * NR - j * (j - 1) / 2; };       // start of column j of a column-packed lower triangle
  // Row-major packed entry k = tri(i, j) (the order of the global field) sits at col(j) + i - j in LDS.  A thread walks
  // k = copy_row, copy_row + kRowsPerPass, ...: (i, j) of the first from a square root, the others by stepping.
  struct Walk {
    int i, j;
    __device__ __forceinline__ void start(int k) {
      i = (int)((__builtin_sqrtf((float)(8 * k + 1)) - 1.0f) * 0.5f);
      if ((i + 1) * (i + 2) / 2 <= k) ++i;
      if (i * (i + 1) / 2 > k) --i;
      j = k - i * (i + 1) / 2;
    }
    __device__ __forceinline__ void advance(int step) {
      j += step;
      while (j > i) {
        j -= i + 1;
        ++i;
      }
    }
    __device__ __forceinline__ int lds_index() const { return j * NR - j * (j - 1) / 2 + i - j; }
  };
  const long long tiles = (n + 63) >> 6;
  const long long units = tiles * kGroupsPerTile;
  // XCD-aware order: workgroups go round-robin over the 8 XCDs, each with its own L2.  The groups of ONE tile share
  // every 128-byte line of it (a group's run is 32-64 bytes), so they are handed to workgroups of the same XCD that
  // start together: virtual id v -> XCD v % 8, position v / 8 on it; consecutive positions walk the groups of a tile.
  constexpr int kXcds = 8;
  const long long rounds = (units + kXcds * kGroupsPerTile - 1) / (kXcds * kGroupsPerTile);
  for (long long v = blockIdx.x; v < rounds * kXcds * kGroupsPerTile; v += gridDim.x) {
    const long long pos = v / kXcds;
    const long long unit = ((pos / kGroupsPerTile) * kXcds + v % kXcds) * kGroupsPerTile + pos % kGroupsPerTile;
    if (unit >= units) continue;      // workgroup-uniform
    const long long tile = unit / kGroupsPerTile;
    const int first = (int)(unit % kGroupsPerTile) * G;     // first chain of the group within its tile
    // the descriptor spans this tile only: the fields may pass 4 GiB (unit is workgroup-uniform, so is the pointer)
    const __amdgpu_buffer_rsrc_t src = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<R *>(cov + tile * (long long)P * 64), 0, (unsigned int)(P * 64 * sizeof(R)), 0x00020000);
    const __amdgpu_buffer_rsrc_t dst = __builtin_amdgcn_make_buffer_rsrc(
        factor + tile * (long long)P * 64, 0, (unsigned int)(P * 64 * sizeof(R)), 0x00020000);
    const unsigned int lane_off = (unsigned int)((first + copy_chain) * sizeof(R));
    constexpr int AUX = NT ? ME_NT_AUX : 0;
    auto load = [&](int k) -> R {
      if constexpr (sizeof(R) == 4)
        return __builtin_bit_cast(R, __builtin_amdgcn_raw_buffer_load_b32(src, lane_off + (unsigned int)k * 256u, 0, AUX));
      else
        return __builtin_bit_cast(R, __builtin_amdgcn_raw_buffer_load_b64(src, lane_off + (unsigned int)k * 512u, 0, AUX));
    };
    auto store = [&](int k, R v) {
      if constexpr (sizeof(R) == 4) {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v), dst, lane_off + (unsigned int)k * 256u, 0, AUX);
      } else {
        using v2 = decltype(__builtin_amdgcn_raw_buffer_load_b64(dst, 0u, 0u, 0));
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2, v), dst, lane_off + (unsigned int)k * 512u, 0, AUX);
      }
    };
    // ---- load: tile-major global -> chain-major LDS, kBatch entries per thread in flight
    constexpr int kBatch = 8;
#ifndef ME_FACTOR_TILE_SKIP_COPY
    {
      Walk w;
      w.start(copy_row);
      for (int k0 = copy_row; k0 < P; k0 += kBatch * kRowsPerPass) {
        R v[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
          const int k = k0 + u * kRowsPerPass;
          v[u] = k < P ? load(k) : R(0);
        }
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
          const int k = k0 + u * kRowsPerPass;
          if (k < P) lds[copy_chain * PS + w.lds_index()] = v[u];
          w.advance(kRowsPerPass);
        }
      }
    }
#endif
    __syncthreads();
    // ---- factor: lane = row, in place in LDS
    for (int pass = 0; pass < T::kPasses; ++pass) {
    const int my_chain = (pass * T::kWaves + wave) * T::CPW + lane / LPC;   // chain of the workgroup this lane works on
    R *mine = lds + my_chain * PS;
#ifdef ME_FACTOR_TILE_SKIP_COPY
    const bool valid = false;
#else
    const bool valid = tile * 64 + first + my_chain < n;     // the last tile may be ragged: nothing to flag there
#endif
#ifndef ME_FACTOR_TILE_SKIP_FACTOR
    R a[RPL][NR];
#pragma unroll
    for (int r = 0; r < RPL; ++r)
#pragma unroll
      for (int k = 0; k < NR; ++k) {
        const int row = my_row + r * LPC;
        a[r][k] = (k <= row && row < NR) ? mine[col(k) + row - k] : R(0);
      }
    constexpr int kReadlane = (T::CPW == 1 && RPL == 1) ? (sizeof(R) == 4 ? ME_FACTOR_TILE_READLANE_F32 : ME_FACTOR_TILE_READLANE_F64) : 0;
    static_for<NR>([&](auto column) {
      constexpr int j = decltype(column)::value;
      constexpr int r0 = j / LPC;            // row blocks below r0 are finished: rows r LPC .. r LPC + LPC - 1 < j
      R s0[RPL], s1[RPL];
#pragma unroll
      for (int r = r0; r < RPL; ++r) {
        s0[r] = a[r][j];
        s1[r] = R(0);
      }
#pragma unroll
      for (int k = 0; k + 1 < j; k += 2) {
        // L_jk for the whole group: a broadcast read of LDS -- or, with one matrix per wavefront and one row per lane, lane
        // j's own register k through v_readlane.  The LDS read costs LDS bandwidth (64 lanes x 8 bytes whatever the
        // address pattern), the readlane two vector instructions in float64: taking every ME_FACTOR_TILE_READLANE-th pair
        // from the registers balances the two pipes.
        R l0, l1;
        if (kReadlane > 0 && (k / 2) % (kReadlane > 0 ? kReadlane : 1) == kReadlane - 1) {
          l0 = lane_value(a[0][k], j, true);
          l1 = lane_value(a[0][k + 1], j, true);
        } else {
          l0 = mine[col(k) + j - k];
          l1 = mine[col(k + 1) + j - k - 1];
        }
#pragma unroll
        for (int r = r0; r < RPL; ++r) {
          s0[r] -= a[r][k] * l0;
          s1[r] -= a[r][k + 1] * l1;
        }
      }
      if constexpr (j % 2 == 1) {
        const R l0 = mine[col(j - 1) + 1];
#pragma unroll
        for (int r = r0; r < RPL; ++r) s0[r] -= a[r][j - 1] * l0;
      }
      R s[RPL];
#pragma unroll
      for (int r = r0; r < RPL; ++r) s[r] = s0[r] + s1[r];
      R d = lane_value(s[r0], (lane / LPC) * LPC + j % LPC, T::CPW == 1);      // the pivot: row j's own sum
      if (!(d > R(0))) {
        bad_pivot = bad_pivot || valid;
        d = R(1e-30);
      }
      R dg, inv;
      sqrt_and_inverse(d, dg, inv);
#pragma unroll
      for (int r = r0; r < RPL; ++r) {
        const int row = my_row + r * LPC;
        a[r][j] = row == j ? dg : s[r] * inv;
        if (row >= j && row < NR) mine[col(j) + row - j] = a[r][j];
      }
      asm volatile("" ::: "memory");     // the column is in LDS (in program order) before row j + 1 is broadcast from there
    });
#endif
    }
    __syncthreads();
    // ---- store: chain-major LDS -> tile-major global
#ifndef ME_FACTOR_TILE_SKIP_COPY
    if (tile * 64 + first + copy_chain < n) {
      Walk w;
      w.start(copy_row);
      for (int k0 = copy_row; k0 < P; k0 += kBatch * kRowsPerPass) {
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
          const int k = k0 + u * kRowsPerPass;
          if (k < P) store(k, lds[copy_chain * PS + w.lds_index()]);
          w.advance(kRowsPerPass);
        }
      }
    }
#endif
    __syncthreads();      // the next unit's copy overwrites what this one's stores read
  }
  if (bad_pivot) atomicOr(status, (unsigned int)ST_BAD_PIVOT);
}

template <typename R, int NR, bool NT>
inline hipError_t launch_factor_tile(const R *cov, R *factor, unsigned int *status, long long n, hipStream_t stream) {
  using T = FactorTile<R, NR>;
  static PerDevice<int> blocks_per_device;
  int device = 0;
  hipError_t err = hipGetDevice(&device);
  if (err != hipSuccess) return err;
  const int resident = blocks_per_device.get(device, [&]() -> int {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_factor_tile<R, NR, NT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)T::kLdsBytes) != hipSuccess)
      return -1;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return -1;
    const int per_cu = (int)(160 * 1024 / T::kLdsBytes) > 0 ? (int)(160 * 1024 / T::kLdsBytes) : 1;
    return (prop.multiProcessorCount * per_cu + 7) / 8 * 8;
  });
  if (resident <= 0) return hipErrorInvalidValue;
  const long long units = ((n + 63) >> 6) * (64 / T::kChains);
  const long long cap = (long long)resident * 4;          // a few units per resident workgroup: tails stay short
  const long long padded = (units + 8 * (64 / T::kChains) - 1) / (8 * (64 / T::kChains)) * (8 * (64 / T::kChains));
  // a multiple of 8, so that a workgroup's stride keeps it on the units of its own XCD (see the kernel)
  const unsigned int grid = (unsigned int)((padded < cap ? padded : cap) / 8 * 8);
  hipLaunchKernelGGL((k_factor_tile<R, NR, NT>), dim3(grid), dim3(T::kThreads), T::kLdsBytes, stream, cov, factor, status, n);
  return hipGetLastError();
}

}  // namespace me
