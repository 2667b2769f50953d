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
//            sector -- which the whole workgroup copies into LDS chain-major (the transpose happens in this copy);
//   factor   each lane group reads its chain's rows out of LDS into registers (lane i: row i, NR values) and runs the
//            column-by-column recurrence  L_ij = (C_ij - sum_{k<j} L_ik L_jk) / L_jj:  L_ik is the lane's own register k
//            (compile-time index, the loops are straight-line code), L_jk is row j in LDS -- every lane of the group reads
//            the SAME address, a broadcast -- and column j goes back to LDS as soon as it is final, which is what makes
//            row j complete by the time it is broadcast.  Two accumulators per row break the dependent chain of
//            multiply-adds;
//   store    the finished factors leave LDS the way the matrices came in.
//
// Arithmetic of a pivot and of a column entry is that of cholesky_packed (me_device.h): s = C_jj - sum L_jk^2, L_jj = sqrt(s),
// L_ij = t * (1 / L_jj); a non-positive pivot is flagged and clamped.  Only the order of the k-sum differs (even and odd k
// apart), so results agree with k_factor_stream to rounding.
#pragma once
#include "me_device.h"
#include "me_per_device.h"

namespace me {

template <typename R, int NR>
struct FactorTile {
  static constexpr int P = NR * (NR + 1) / 2;
  static constexpr int PS = P | 1;                                    // odd chain stride in LDS: the transposing copy spreads over the banks
  static constexpr int LPC = NR <= 16 ? 16 : NR <= 32 ? 32 : 64;      // lanes per chain (a power of two >= NR)
  static constexpr int CPW = 64 / LPC;                                // chains per wavefront
  // chains per workgroup: a run of 64 bytes per entry (one sector), as long as the matrices fit in LDS
  static constexpr int kWanted = 64 / (int)sizeof(R);
  static constexpr int kFit = (int)(150 * 1024 / (PS * sizeof(R)));
  static constexpr int kChains = kFit >= kWanted ? kWanted : kFit >= 8 ? 8 : kFit >= 4 ? 4 : kFit >= 2 ? 2 : 1;
  static constexpr int kWaves = kChains / CPW > 0 ? kChains / CPW : 1;
  static constexpr int kThreads = 64 * kWaves;
  static constexpr size_t kLdsBytes = (size_t)kChains * PS * sizeof(R);
  static_assert(NR <= 64, "one row per lane: up to 64 parameters");
  static_assert(kChains >= CPW && kThreads <= 1024, "workgroup shape");
};

template <typename R, int NR, bool NT>
__global__ void __launch_bounds__((FactorTile<R, NR>::kThreads))
k_factor_tile(const R *cov, R *factor, unsigned int *status, long long n) {
  using T = FactorTile<R, NR>;
  using N_ = Num<R>;
  constexpr int P = T::P, PS = T::PS, G = T::kChains, LPC = T::LPC;
  constexpr int kGroupsPerTile = 64 / G;
  constexpr int kRowsPerPass = T::kThreads / G;      // entries a copy pass of the workgroup covers
  extern __shared__ unsigned char lds_raw[];
  R *lds = reinterpret_cast<R *>(lds_raw);
  const int t = threadIdx.x;
  const int copy_chain = t % G, copy_row = t / G;
  const int lane = t & 63, wave = t >> 6;
  const int my_chain = wave * T::CPW + lane / LPC;   // chain of the workgroup this lane works on
  const int my_row = lane % LPC;                     // and its row
  R *mine = lds + my_chain * PS;
  bool bad_pivot = false;
  const long long tiles = (n + 63) >> 6;
  const long long units = tiles * kGroupsPerTile;
  for (long long unit = blockIdx.x; unit < units; unit += gridDim.x) {
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
        if (k < P) lds[copy_chain * PS + k] = v[u];
      }
    }
    __syncthreads();
    // ---- factor: lane = row, in place in LDS
    const bool valid = tile * 64 + first + my_chain < n;     // the last tile may be ragged: nothing to flag there
    R a[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) a[k] = (k <= my_row && my_row < NR) ? mine[my_row * (my_row + 1) / 2 + k] : R(0);
    static_for<NR>([&](auto column) {
      constexpr int j = decltype(column)::value;
      const R *row_j = mine + tri(j, 0);
      R s0 = a[j], s1 = R(0);
#pragma unroll
      for (int k = 0; k + 1 < j; k += 2) {
        s0 -= a[k] * row_j[k];
        s1 -= a[k + 1] * row_j[k + 1];
      }
      if constexpr (j % 2 == 1) s0 -= a[j - 1] * row_j[j - 1];
      const R s = s0 + s1;
      if (my_row == j) mine[tri(j, j)] = s;                 // the pivot, seen by every lane of the group
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      R d = mine[tri(j, j)];
      if (!(d > R(0))) {
        bad_pivot = bad_pivot || valid;
        d = R(1e-30);
      }
      const R dg = N_::sqrt_(d);
      const R inv = R(1) / dg;
      a[j] = my_row == j ? dg : s * inv;
      if (my_row >= j && my_row < NR) mine[my_row * (my_row + 1) / 2 + j] = a[j];
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    });
    __syncthreads();
    // ---- store: chain-major LDS -> tile-major global
    if (tile * 64 + first + copy_chain < n) {
      for (int k0 = copy_row; k0 < P; k0 += kBatch * kRowsPerPass) {
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
          const int k = k0 + u * kRowsPerPass;
          if (k < P) store(k, lds[copy_chain * PS + k]);
        }
      }
    }
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
    return prop.multiProcessorCount * per_cu;
  });
  if (resident <= 0) return hipErrorInvalidValue;
  const long long units = ((n + 63) >> 6) * (64 / T::kChains);
  const long long cap = (long long)resident * 4;          // a few units per resident workgroup: tails stay short
  const unsigned int grid = (unsigned int)(units < cap ? units : cap);
  hipLaunchKernelGGL((k_factor_tile<R, NR, NT>), dim3(grid), dim3(T::kThreads), T::kLdsBytes, stream, cov, factor, status, n);
  return hipGetLastError();
}

}  // namespace me
