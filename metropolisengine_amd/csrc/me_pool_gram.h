// Stage 1 of the pooled-moment reduction for 64 real parameters (float32) on the matrix cores: the 64 x 64 second-moment
// matrix of a 64-chain tile is the Gram product X X^T (X = [parameter][chain]), 2 080 + 129 entries that the generic
// k_pool_reduce (me_generic.hip) walks entry by entry through LDS (237 us at 2^19 chains).  Here one wavefront owns a
// tile: lanes load their chains' rows (coalesced), park the tile in a wavefront-private LDS slab with an odd pitch, and
// read it back TRANSPOSED as MFMA operands -- lane l of v_mfma_f32_32x32x2_f32 holds A[row l&31][k = l>>5], i.e.
// parameter l&31 of chain 2s + (l>>5); B's map is the mirror image, so A's fragment of block nb IS B's operand and only
// the three lower blocks (0,0), (1,0), (1,1) are formed (96 MFMAs per tile).  fp32 products and sums inside a tile, fp64
// across tiles -- the same numerics class as the generic kernel -- and the per-wavefront partial sums go to
// partials[wavefront][entry] in the order k_pool_finish expects.  2^19 chains: 56 us (MFMA loop ~23, loads ~11, row sums
// ~4, the rest LDS staging and the fp64 flush), against 237 us for the generic kernel.
#pragma once

#include "me_dense_mfma.h"
#include "me_dense_f64.h"
#include "me_per_device.h"

namespace me {

constexpr int kGramWaves = 2;            // wavefronts per workgroup, each with its own LDS slab (2 x 16.6 KB)
constexpr int kGramPitch = 65;           // floats per parameter row (odd: column reads of 32 rows hit 32 banks)

template <int UNUSED = 0>
__global__ void __launch_bounds__(64 * kGramWaves) k_pool_gram64(const float *__restrict__ x, long long n,
                                                                 double *__restrict__ partials, int n_rows) {
  constexpr int D = 64;
  constexpr int n_aug = 2 * D;                   // sum x_i, sum |x_i|
  constexpr int n_entries = 1 + n_aug + D * (D + 1) / 2;
  __shared__ float slab[kGramWaves][D * kGramPitch];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row_id = blockIdx.x * kGramWaves + wave;            // this wavefront's row of partials
  if (row_id >= n_rows) return;
  float *tile = slab[wave];
  const int r31 = lane & 31, h = lane >> 5;

  double g[3][16];                               // blocks (0,0), (1,0), (1,1) in MFMA accumulator layout
  double sum_x = 0.0, sum_abs = 0.0, count = 0.0;   // lane l: parameter l
#pragma unroll
  for (int b = 0; b < 3; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) g[b][r] = 0.0;

  const long long n_tiles = (n + 63) / 64;
  // One wavefront per SIMD is resident (1 024 partial rows), so nothing else hides the load latency: the next tile's 64
  // rows are requested before the current tile is reduced (first version without the prefetch: 184 us at 2^19 chains).
  float next[D];
  auto request = [&](long long t) {
    const long long c = t * 64 + lane;
    const bool live = t < n_tiles && c < n;
    const long long cc = live ? c : 0;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const float v = x[state_index<float, D>(cc, d, n)];       // tile-major at 64 parameters (me_device.h: XField)
      next[d] = live ? v : 0.0f;
    }
  };
  request(row_id);
  for (long long t = row_id; t < n_tiles; t += n_rows) {
    // chain `lane`'s 64 parameters -> LDS column `lane` (lane-linear writes)
#pragma unroll
    for (int d = 0; d < D; ++d) tile[d * kGramPitch + lane] = next[d];
    request(t + n_rows);
    count += (double)(n - t * 64 < 64 ? n - t * 64 : 64);
    // row sums: lane l walks parameter l's row (bank = (l + col) mod 64: conflict-free)
    float sx = 0.0f, sa = 0.0f;
#pragma unroll 16
    for (int col = 0; col < 64; ++col) {
      const float v = tile[lane * kGramPitch + col];
      sx += v;
      sa += __builtin_fabsf(v);
    }
    sum_x += (double)sx;
    sum_abs += (double)sa;
    // Gram blocks: k step s covers chains 2s, 2s+1
    f32x16 acc[3];
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[b][r] = 0.0f;
#pragma unroll 8
    for (int s = 0; s < 32; ++s) {
      const float lo = tile[r31 * kGramPitch + 2 * s + h];            // parameters 0..31 of chain 2s + h
      const float hi = tile[(32 + r31) * kGramPitch + 2 * s + h];     // parameters 32..63
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(lo, lo, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(hi, lo, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(hi, hi, acc[2], 0, 0, 0);
    }
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) g[b][r] += (double)acc[b][r];
  }

  double *out = partials + (long long)row_id * n_entries;
  if (lane == 0) out[0] = count;
  out[1 + lane] = sum_x;
  out[1 + D + lane] = sum_abs;
  // accumulator register r of lane l is element (row acc_row(r) + 4h, column l & 31) of its block
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    const int i0 = b == 0 ? 0 : 32, j0 = b == 2 ? 32 : 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = i0 + acc_row(r) + 4 * h, j = j0 + r31;
      if (i >= j) out[1 + n_aug + i * (i + 1) / 2 + j] = g[b][r];
    }
  }
}

inline hipError_t launch_pool_gram64(const void *x, long long n, double *partials, int n_rows, hipStream_t stream) {
  hipLaunchKernelGGL(k_pool_gram64<0>, dim3((unsigned)((n_rows + kGramWaves - 1) / kGramWaves)), dim3(64 * kGramWaves), 0, stream,
                     (const float *)x, n, partials, n_rows);
  return hipGetLastError();
}

// ... and in float64 (the reference's dtype; the generic kernel: 500 us at 2^19 chains): v_mfma_f64_16x16x4_f64, lane l of
// which holds A[row l & 15][k = l >> 4] and B[k = l >> 4][column l & 15] -- for a Gram product both are "parameter
// 16 b + (l & 15) of chain 4 s + (l >> 4)", so ONE fragment per 16-parameter block and k step serves as A of its block
// row and B of its block column: 4 LDS reads and the 10 lower blocks' MFMAs per step of four chains, 160 MFMAs per tile.
// Everything is float64, so the accumulators simply run across the tiles of a wavefront.  The row sums (sum x_i,
// sum |x_i|) come from the same fragments: a lane adds up what it reads -- the chains 4 s + (l >> 4) of its parameter --
// and the four lane groups are added once at the end (two ds_swizzle-free __shfl_xor steps).
constexpr int kGramPitchF64 = 68;        // doubles per parameter row: 16 rows x 4 columns of a fragment read hit 64 different 8-byte slots

template <int UNUSED = 0>
__global__ void __launch_bounds__(64 * kGramWaves) k_pool_gram64_f64(const double *__restrict__ x, long long n,
                                                                     double *__restrict__ partials, int n_rows) {
  constexpr int D = 64;
  constexpr int n_aug = 2 * D;
  constexpr int n_entries = 1 + n_aug + D * (D + 1) / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char gram_lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row_id = blockIdx.x * kGramWaves + wave;
  if (row_id >= n_rows) return;
  double *tile = reinterpret_cast<double *>(gram_lds) + (size_t)wave * D * kGramPitchF64;
  const int j = lane & 15, h = lane >> 4;

  f64x4 acc[10];                                 // blocks (bi, bj), bj <= bi, at index bi (bi + 1) / 2 + bj
#pragma unroll
  for (int b = 0; b < 10; ++b) acc[b] = f64x4{0.0, 0.0, 0.0, 0.0};
  double sum_x[4] = {0.0, 0.0, 0.0, 0.0}, sum_abs[4] = {0.0, 0.0, 0.0, 0.0}, count = 0.0;

  const long long n_tiles = (n + 63) / 64;
  double next[D];                                // the next tile's rows, requested before the current tile is reduced
  auto request = [&](long long t) {
    const long long c = t * 64 + lane;
    const bool live = t < n_tiles && c < n;
    const long long cc = live ? c : 0;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const double v = x[state_index<double, D>(cc, d, n)];
      next[d] = live ? v : 0.0;
    }
  };
  request(row_id);
  for (long long t = row_id; t < n_tiles; t += n_rows) {
#pragma unroll
    for (int d = 0; d < D; ++d) tile[d * kGramPitchF64 + lane] = next[d];
    request(t + n_rows);
    count += (double)(n - t * 64 < 64 ? n - t * 64 : 64);
#pragma unroll 4
    for (int s = 0; s < 16; ++s) {
      double f[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        f[b] = tile[(16 * b + j) * kGramPitchF64 + 4 * s + h];
        sum_x[b] += f[b];
        sum_abs[b] += __builtin_fabs(f[b]);
      }
#pragma unroll
      for (int bi = 0; bi < 4; ++bi)
#pragma unroll
        for (int bj = 0; bj <= bi; ++bj)
          acc[bi * (bi + 1) / 2 + bj] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[bi], f[bj], acc[bi * (bi + 1) / 2 + bj], 0, 0, 0);
    }
    asm volatile("" ::: "memory");               // the tile is consumed before the next one overwrites it
  }

  double *out = partials + (long long)row_id * n_entries;
  if (lane == 0) out[0] = count;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    double sx = sum_x[b], sa = sum_abs[b];
    sx += __shfl_xor(sx, 16);
    sa += __shfl_xor(sa, 16);
    sx += __shfl_xor(sx, 32);
    sa += __shfl_xor(sa, 32);
    if (h == 0) {
      out[1 + 16 * b + j] = sx;
      out[1 + D + 16 * b + j] = sa;
    }
  }
  // accumulator register r of lane l is element (row h + 4 r, column j) of its block
#pragma unroll
  for (int bi = 0; bi < 4; ++bi)
#pragma unroll
    for (int bj = 0; bj <= bi; ++bj)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * bi + h + 4 * r, col = 16 * bj + j;
        if (row >= col) out[1 + n_aug + row * (row + 1) / 2 + col] = acc[bi * (bi + 1) / 2 + bj][r];
      }
}

inline hipError_t launch_pool_gram64_f64(const void *x, long long n, double *partials, int n_rows, hipStream_t stream) {
  constexpr size_t lds = (size_t)kGramWaves * 64 * kGramPitchF64 * sizeof(double);      // 2 x 34 KB
  static PerDevice<int> raised;
  int device = 0;
  hipError_t err = hipGetDevice(&device);
  if (err != hipSuccess) return err;
  if (raised.get(device, [&]() -> int {
        return hipFuncSetAttribute((const void *)k_pool_gram64_f64<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess ? 1 : -1;
      }) < 0)
    return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_pool_gram64_f64<0>, dim3((unsigned)((n_rows + kGramWaves - 1) / kGramWaves)), dim3(64 * kGramWaves), lds, stream,
                     (const double *)x, n, partials, n_rows);
  return hipGetLastError();
}

// The same idea for small parameter spaces: all augmented rows [x (D) | |x_r| (NR) | |z_c| (NC)] fit one 32-row block
// (D + NR + NC <= 32: every prebuilt kernel set except 64 real parameters), so a tile costs 32 MFMAs in float32 and 48 in
// float64; the products the layout does not ask for (anything with an |.| row) are simply not written.
//
// Round 3: this reduction sits inside every cycle of BASELINE config 5's protocol, where it cost 27 us (17 us stage 1 with
// one wavefront per SIMD walking four tiles one after the other, 10 us for k_pool_finish to add up 1 024 partial rows)
// beside a 70 us cycle.  Now a workgroup is 16 (float32) / 8 (float64) wavefronts, each with its own LDS slab and its own
// tiles -- four / two wavefronts per SIMD hide each other's load -> LDS -> MFMA latency chain -- and the workgroup adds its
// wavefronts' sums up in LDS (fixed order: bitwise reproducible) and writes ONE partial row; at most 256 rows
// (pool_reduce_blocks), a quarter of what k_pool_finish had to read.
constexpr int kSmallWavesF32 = 16, kSmallWavesF64 = 8;
constexpr int kSmallSlabBytesF32 = 8832;                       // >= 32 x 65 floats (the tile) and >= 1 089 doubles (the sums)
constexpr int kSmallPitchF64 = 68;
constexpr int kSmallSlabBytesF64 = 32 * kSmallPitchF64 * 8;    // 17 408: the tile; the sums need 897 doubles

template <int NR, int NC>
__global__ void __launch_bounds__(64 * kSmallWavesF32) k_pool_gram32(const float *__restrict__ x, long long n,
                                                                     double *__restrict__ partials, int n_rows) {
  constexpr int D = NR + 2 * NC;
  constexpr int n_aug = D + NR + NC;
  static_assert(n_aug <= 32, "one 32-row MFMA block");
  static_assert(32 * kGramPitch * 4 <= kSmallSlabBytesF32 && 1089 * 8 <= kSmallSlabBytesF32, "slab holds the tile and the sums");
  constexpr int n_entries = 1 + n_aug + D * (D + 1) / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char small_lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float *tile = reinterpret_cast<float *>(small_lds + (size_t)wave * kSmallSlabBytesF32);
  const int r31 = lane & 31, h = lane >> 5;
#pragma unroll
  for (int r = n_aug; r < 32; ++r) tile[r * kGramPitch + lane] = 0.0f;     // unused rows stay zero

  double g[16], row_sum = 0.0, count = 0.0;
#pragma unroll
  for (int r = 0; r < 16; ++r) g[r] = 0.0;

  const long long n_tiles = (n + 63) / 64;
  const long long first = (long long)blockIdx.x * kSmallWavesF32 + wave, step = (long long)n_rows * kSmallWavesF32;
  float next[D];
  auto request = [&](long long t) {
    const long long c = t * 64 + lane;
    const bool live = t < n_tiles && c < n;
    const long long cc = live ? c : 0;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const float v = x[state_index<float, D>(cc, d, n)];
      next[d] = live ? v : 0.0f;
    }
  };
  request(first);
  for (long long t = first; t < n_tiles; t += step) {
#pragma unroll
    for (int d = 0; d < D; ++d) tile[d * kGramPitch + lane] = next[d];
#pragma unroll
    for (int i = 0; i < NR; ++i) tile[(D + i) * kGramPitch + lane] = __builtin_fabsf(next[i]);
#pragma unroll
    for (int j = 0; j < NC; ++j)
      tile[(D + NR + j) * kGramPitch + lane] =
          (float)sqrt((double)next[NR + j] * next[NR + j] + (double)next[NR + NC + j] * next[NR + NC + j]);
    request(t + step);
    count += (double)(n - t * 64 < 64 ? n - t * 64 : 64);
    __builtin_amdgcn_wave_barrier();              // the tile is written lane-linear and read across lanes
    float s = 0.0f;
#pragma unroll 16
    for (int col = 0; col < 32; ++col) s += tile[r31 * kGramPitch + 32 * h + col];
    s += __shfl_xor(s, 32, 64);
    row_sum += (double)s;
    f32x16 acc[2];     // two accumulators: consecutive MFMAs do not wait on each other
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][r] = acc[1][r] = 0.0f;
#pragma unroll 8
    for (int k = 0; k < 32; k += 2) {
      const float a0 = tile[r31 * kGramPitch + 2 * k + h];
      const float a1 = tile[r31 * kGramPitch + 2 * k + 2 + h];
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, a0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, a1, acc[1], 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) g[r] += (double)(acc[0][r] + acc[1][r]);
    __builtin_amdgcn_wave_barrier();              // ... and consumed before the next tile overwrites it
  }

  // the workgroup's sum: every wavefront parks its values in its own slab (v = 64 r + lane: g[r]; 1024 + lane: the row sum;
  // 1088: the chain count), then thread v adds slot v of the wavefronts in a fixed order and writes the partial row
  double *red = reinterpret_cast<double *>(small_lds + (size_t)wave * kSmallSlabBytesF32);
#pragma unroll
  for (int r = 0; r < 16; ++r) red[64 * r + lane] = g[r];
  red[1024 + lane] = row_sum;
  if (lane == 0) red[1088] = count;
  __syncthreads();
  double *out = partials + (long long)blockIdx.x * n_entries;
  for (int v = threadIdx.x; v < 1089; v += 64 * kSmallWavesF32) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < kSmallWavesF32; ++w) s += reinterpret_cast<const double *>(small_lds + (size_t)w * kSmallSlabBytesF32)[v];
    if (v == 1088) {
      out[0] = s;
    } else if (v >= 1024) {
      if (v - 1024 < n_aug) out[1 + (v - 1024)] = s;       // lanes l and l + 32 hold the same row sum: the first 32 are written
    } else {
      const int r = v >> 6, l = v & 63;
      const int i = acc_row(r) + 4 * (l >> 5), j = l & 31;   // accumulator register r of lane l: element (i, j)
      if (i < D && j <= i) out[1 + n_aug + i * (i + 1) / 2 + j] = s;
    }
  }
}

template <int NR, int NC>
inline hipError_t launch_pool_gram32(const void *x, long long n, double *partials, int n_rows, hipStream_t stream) {
  constexpr size_t lds = (size_t)kSmallWavesF32 * kSmallSlabBytesF32;      // 138 KB: one workgroup per CU
  static PerDevice<int> raised;
  int device = 0;
  hipError_t err = hipGetDevice(&device);
  if (err != hipSuccess) return err;
  if (raised.get(device, [&]() -> int {
        return hipFuncSetAttribute((const void *)k_pool_gram32<NR, NC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess ? 1 : -1;
      }) < 0)
    return hipErrorInvalidValue;
  hipLaunchKernelGGL((k_pool_gram32<NR, NC>), dim3((unsigned)n_rows), dim3(64 * kSmallWavesF32), lds, stream, (const float *)x, n, partials, n_rows);
  return hipGetLastError();
}

// ... and at the reference's precision: v_mfma_f64_16x16x4_f64 on the two 16-row blocks of the augmented tile (one fragment
// per block and step of four chains is A of its block row and B of its block column: 2 LDS reads and the 3 lower blocks'
// MFMAs per step, 48 per tile); everything is float64, so the accumulators run across a wavefront's tiles; the row sums come
// from the same fragments.  (The generic k_pool_reduce<double> this replaces: 100 us at 16 parameters x 2^20 chains.)
template <int NR, int NC>
__global__ void __launch_bounds__(64 * kSmallWavesF64) k_pool_gram32_f64(const double *__restrict__ x, long long n,
                                                                         double *__restrict__ partials, int n_rows) {
  constexpr int D = NR + 2 * NC;
  constexpr int n_aug = D + NR + NC;
  static_assert(n_aug <= 32, "two 16-row MFMA blocks");
  static_assert(897 * 8 <= kSmallSlabBytesF64, "slab holds the sums");
  constexpr int n_entries = 1 + n_aug + D * (D + 1) / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char small_lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double *tile = reinterpret_cast<double *>(small_lds + (size_t)wave * kSmallSlabBytesF64);
  const int j = lane & 15, h = lane >> 4;
#pragma unroll
  for (int r = n_aug; r < 32; ++r) tile[r * kSmallPitchF64 + lane] = 0.0;

  f64x4 acc[3];                                  // blocks (0,0), (1,0), (1,1)
#pragma unroll
  for (int b = 0; b < 3; ++b) acc[b] = f64x4{0.0, 0.0, 0.0, 0.0};
  double sum_row[2] = {0.0, 0.0}, count = 0.0;

  const long long n_tiles = (n + 63) / 64;
  const long long first = (long long)blockIdx.x * kSmallWavesF64 + wave, step = (long long)n_rows * kSmallWavesF64;
  double next[D];
  auto request = [&](long long t) {
    const long long c = t * 64 + lane;
    const bool live = t < n_tiles && c < n;
    const long long cc = live ? c : 0;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const double v = x[state_index<double, D>(cc, d, n)];
      next[d] = live ? v : 0.0;
    }
  };
  request(first);
  for (long long t = first; t < n_tiles; t += step) {
#pragma unroll
    for (int d = 0; d < D; ++d) tile[d * kSmallPitchF64 + lane] = next[d];
#pragma unroll
    for (int i = 0; i < NR; ++i) tile[(D + i) * kSmallPitchF64 + lane] = __builtin_fabs(next[i]);
#pragma unroll
    for (int k = 0; k < NC; ++k)
      tile[(D + NR + k) * kSmallPitchF64 + lane] = sqrt(next[NR + k] * next[NR + k] + next[NR + NC + k] * next[NR + NC + k]);
    request(t + step);
    count += (double)(n - t * 64 < 64 ? n - t * 64 : 64);
    __builtin_amdgcn_wave_barrier();
#pragma unroll 4
    for (int s = 0; s < 16; ++s) {
      double f[2];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        f[b] = tile[(16 * b + j) * kSmallPitchF64 + 4 * s + h];
        sum_row[b] += f[b];
      }
      acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[0], f[0], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[1], f[0], acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[1], f[1], acc[2], 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();             // the tile is consumed before the next one overwrites it
  }
#pragma unroll
  for (int b = 0; b < 2; ++b) {                  // the four lane groups hold a quarter of the chains each
    sum_row[b] += __shfl_xor(sum_row[b], 16);
    sum_row[b] += __shfl_xor(sum_row[b], 32);
  }
  // v = 64 (4 b + r) + lane: acc[b][r]; 768 + 64 b + lane: row sums of block b; 896: the chain count
  double *red = reinterpret_cast<double *>(small_lds + (size_t)wave * kSmallSlabBytesF64);
#pragma unroll
  for (int b = 0; b < 3; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[64 * (4 * b + r) + lane] = acc[b][r];
  red[768 + lane] = sum_row[0];
  red[832 + lane] = sum_row[1];
  if (lane == 0) red[896] = count;
  __syncthreads();
  double *out = partials + (long long)blockIdx.x * n_entries;
  for (int v = threadIdx.x; v < 897; v += 64 * kSmallWavesF64) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < kSmallWavesF64; ++w) s += reinterpret_cast<const double *>(small_lds + (size_t)w * kSmallSlabBytesF64)[v];
    if (v == 896) {
      out[0] = s;
    } else if (v >= 768) {
      const int b = (v - 768) >> 6, l = (v - 768) & 63;
      if (l < 16 && 16 * b + l < n_aug) out[1 + 16 * b + l] = s;     // lane group 0 of each block
    } else {
      const int k = v >> 6, l = v & 63, b = k >> 2, r = k & 3;
      const int bi = b == 0 ? 0 : 1, bj = b == 2 ? 1 : 0;
      const int row = 16 * bi + (l >> 4) + 4 * r, col = 16 * bj + (l & 15);   // accumulator register r of lane l
      if (row < D && col <= row) out[1 + n_aug + row * (row + 1) / 2 + col] = s;
    }
  }
}

template <int NR, int NC>
inline hipError_t launch_pool_gram32_f64(const void *x, long long n, double *partials, int n_rows, hipStream_t stream) {
  constexpr size_t lds = (size_t)kSmallWavesF64 * kSmallSlabBytesF64;      // 136 KB: one workgroup per CU
  static PerDevice<int> raised;
  int device = 0;
  hipError_t err = hipGetDevice(&device);
  if (err != hipSuccess) return err;
  if (raised.get(device, [&]() -> int {
        return hipFuncSetAttribute((const void *)k_pool_gram32_f64<NR, NC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess ? 1 : -1;
      }) < 0)
    return hipErrorInvalidValue;
  hipLaunchKernelGGL((k_pool_gram32_f64<NR, NC>), dim3((unsigned)n_rows), dim3(64 * kSmallWavesF64), lds, stream, (const double *)x, n, partials, n_rows);
  return hipGetLastError();
}

}  // namespace me
