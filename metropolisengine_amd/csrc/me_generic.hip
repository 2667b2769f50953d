// Dimension-independent kernels: broadcast fill and the ensemble (pooled) moment reduction.
#include "me_internal.h"
#include "me_per_device.h"

namespace me {
namespace {

template <typename R>
__global__ void __launch_bounds__(kBlockThreads) k_broadcast_rows(R *dst, const R *row_values, int rows, long long n) {
  const long long stride = (long long)gridDim.x * kBlockThreads;
  for (long long c = (long long)blockIdx.x * kBlockThreads + threadIdx.x; c < n; c += stride)
    for (int r = 0; r < rows; ++r) dst[(long long)r * n + c] = row_values[r];
}

// the same for a tile-major packed field [tile of 64 chains][entries][64 lanes] (me_device.h: TiledField)
template <typename R>
__global__ void __launch_bounds__(kBlockThreads) k_broadcast_tiled(R *dst, const R *entry_values, int entries, long long n_padded) {
  const long long total = n_padded * entries, stride = (long long)gridDim.x * kBlockThreads;
  for (long long i = (long long)blockIdx.x * kBlockThreads + threadIdx.x; i < total; i += stride)
    dst[i] = entry_values[(i >> 6) % entries];
}

// (i, j), i >= j, of packed lower-triangle index p
__device__ __forceinline__ void pair_of(int p, int &i, int &j) {
  i = (int)((sqrtf(8.0f * (float)p + 1.0f) - 1.0f) * 0.5f);
  while (i * (i + 1) / 2 > p) --i;
  while ((i + 1) * (i + 2) / 2 <= p) ++i;
  j = p - i * (i + 1) / 2;
}

constexpr int kMaxEntries = 12;   // moment entries per thread and pass -> 3072 entries per pass

// k_pool_reduce: S = sum over chains of [1, x, x x^T (lower), |x_r|, |z_c|] in fp64, in two stages.
// Stage 1: a block stages a tile of TILE chains x (D + nr + nc) augmented rows in LDS (pitch TILE + 1: threads walk
// different rows at the same column); every thread owns a fixed set of entries and walks the tile's columns for each (in
// the device dtype within a tile, fp64 across tiles); per-block partial sums are written to partials[block][entry].
// One pass covers kMaxEntries x 256 = 3072 entries starting at entry_base: parameter spaces with more entries than that
// (beyond ~75 real degrees of freedom, up to the 42 k entries of 290 parameters) take several passes, each staging the
// tile again (x is read once per pass -- this reduction runs once per adaptation, not per step).  TILE is 64 where the
// staged tile fits the LDS, 32 or 16 for the largest spaces.
// Stage 2 (k_pool_finish) sums the partials per entry and writes the result in the public order (me_pooled_moments).
// No atomics: a thousand blocks adding to the same ~200 words with fp64 atomics serialise at the memory side (measured:
// the one-launch atomic version was 2x slower).
template <typename R, int TILE>
__global__ void __launch_bounds__(kBlockThreads) k_pool_reduce(const R *x, long long n, int nr, int nc, int entry_base, int tiled, double *partials) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  R *tile = reinterpret_cast<R *>(smem_raw);
  constexpr int kTilePitch = TILE + 1;
  const int d = nr + 2 * nc;
  const int n_aug = d + nr + nc;
  const int n_pair = d * (d + 1) / 2;
  const int n_entries = 1 + n_aug + n_pair;
  const int tid = threadIdx.x;
  // where row r of chain c sits: component-major, or the tile-major state of the larger register-resident sets
  auto at = [&](int r, long long c) -> long long { return tiled ? ((c >> 6) * (long long)d + r) * 64 + (c & 63) : (long long)r * n + c; };

  // decode this thread's entries once: entry 0 = count, 1..n_aug = row sums, then the pair products (i >= j)
  int row_i[kMaxEntries], row_j[kMaxEntries];
  bool active[kMaxEntries];
  double acc[kMaxEntries];
#pragma unroll
  for (int k = 0; k < kMaxEntries; ++k) {
    const int e = entry_base + tid + k * kBlockThreads;
    acc[k] = 0.0;
    row_i[k] = row_j[k] = -1;
    active[k] = e < n_entries;
    if (!active[k] || e == 0) continue;
    if (e <= n_aug) {
      row_i[k] = e - 1;
    } else {
      pair_of(e - 1 - n_aug, row_i[k], row_j[k]);
    }
  }

  const long long n_tiles = (n + TILE - 1) / TILE;
  for (long long t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const long long base = t * TILE;
    const int valid = (int)((n - base) < TILE ? (n - base) : TILE);
    // stage: thread (r, col) pattern; 256 threads cover 256 / TILE rows x TILE columns per pass
    for (int r = tid / TILE; r < n_aug; r += kBlockThreads / TILE) {
      const int col = tid % TILE;
      R v = 0;
      if (col < valid) {
        const long long c = base + col;
        if (r < d) v = x[at(r, c)];
        else if (r < d + nr) {
          const R xv = x[at(r - d, c)];
          v = xv < 0 ? -xv : xv;
        } else {
          const int j = r - d - nr;
          const R re = x[at(nr + j, c)], im = x[at(nr + nc + j, c)];
          v = (R)sqrt((double)re * re + (double)im * im);
        }
      }
      tile[r * kTilePitch + col] = v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kMaxEntries; ++k) {
      if (!active[k]) continue;
      if (row_i[k] < 0) {
        acc[k] += (double)valid;
      } else if (row_j[k] < 0) {
        const R *ri = tile + row_i[k] * kTilePitch;
        R s = 0;
        for (int col = 0; col < TILE; ++col) s += ri[col];
        acc[k] += (double)s;
      } else {
        const R *ri = tile + row_i[k] * kTilePitch;
        const R *rj = tile + row_j[k] * kTilePitch;
        R s = 0;
        for (int col = 0; col < TILE; ++col) s += ri[col] * rj[col];
        acc[k] += (double)s;
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < kMaxEntries; ++k)
    if (active[k]) partials[(long long)blockIdx.x * n_entries + entry_base + tid + k * kBlockThreads] = acc[k];
}

// Stage 2: 64 entries x 16 block-slices per workgroup; a thread sums every 16th block's partial of its entry (consecutive
// lanes read consecutive entries: coalesced; 8 independent loads in flight), the slices meet in LDS in a fixed order,
// and the total goes where the public layout wants it.  One extra workgroup (the last) sums the per-wavefront
// acceptance slots, so the pooled path needs no separate k_sum_slots launch.
constexpr int kFinishThreads = 1024, kFinishSlices = kFinishThreads / 64;
__global__ void __launch_bounds__(kFinishThreads) k_pool_finish(const double *partials, int n_blocks, int nr, int nc,
                                                                const unsigned long long *slots, long long n_slots,
                                                                double proposed, double *out) {
  const int d = nr + 2 * nc;
  const int n_aug = d + nr + nc;
  const int n_pair = d * (d + 1) / 2;
  const int n_entries = 1 + n_aug + n_pair;
  const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
  if (blockIdx.x == gridDim.x - 1) {   // acceptance slots -> accepted, proposed
    __shared__ unsigned long long wave_sum[kFinishSlices];
    unsigned long long s = 0;
    long long i = threadIdx.x;
    for (; i + 7 * kFinishThreads < n_slots; i += 8 * kFinishThreads) {
      unsigned long long v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = slots[i + u * kFinishThreads];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; i < n_slots; i += kFinishThreads) s += slots[i];
#pragma unroll
    for (int w = 32; w > 0; w >>= 1) s += __shfl_xor(s, w, 64);
    if (lane == 0) wave_sum[slice] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned long long t = 0;
#pragma unroll
      for (int w = 0; w < kFinishSlices; ++w) t += wave_sum[w];
      const long long size = moments_size(nr, nc);
      out[size - 2] = (double)t;
      out[size - 1] = proposed;
    }
    return;
  }
  __shared__ double part[kFinishSlices][64];
  const int e = blockIdx.x * 64 + lane;
  double s = 0.0;
  if (e < n_entries) {
#pragma unroll 8
    for (int b = slice; b < n_blocks; b += kFinishSlices) s += partials[(long long)b * n_entries + e];
  }
  part[slice][lane] = s;
  __syncthreads();
  if (slice != 0) return;
  s = 0.0;
#pragma unroll
  for (int w = 0; w < kFinishSlices; ++w) s += part[w][lane];
  if (e == 0) {
    out[0] = s;                                            // n
  } else if (e <= n_aug) {
    const int r = e - 1;
    out[r < d ? 1 + r : 1 + d + n_pair + (r - d)] = s;     // sum x | sum |x_r|, sum |z_c|
  } else if (e < n_entries) {
    const int p = e - 1 - n_aug;
    int i, j;
    pair_of(p, i, j);
    out[1 + d + p] = s;                                    // sum x_i x_j
    if (i == j && i < nr) out[1 + d + n_pair + nr + nc + i] = s;   // sum x_r^2 (observable)
  }
}

}  // namespace

hipError_t launch_broadcast_rows(void *dst, const void *row_values, int rows, long long n, int dtype,
                                 hipStream_t stream) {
  long long blocks = (n + kBlockThreads - 1) / kBlockThreads;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  if (dtype == ME_F32)
    hipLaunchKernelGGL(k_broadcast_rows<float>, dim3((unsigned)blocks), dim3(kBlockThreads), 0, stream, (float *)dst,
                       (const float *)row_values, rows, n);
  else
    hipLaunchKernelGGL(k_broadcast_rows<double>, dim3((unsigned)blocks), dim3(kBlockThreads), 0, stream, (double *)dst,
                       (const double *)row_values, rows, n);
  return hipGetLastError();
}

hipError_t launch_broadcast_tiled(void *dst, const void *entry_values, int entries, long long n, int dtype,
                                  hipStream_t stream) {
  const long long n_padded = (n + 63) / 64 * 64;
  long long blocks = (n_padded * entries + kBlockThreads - 1) / kBlockThreads;
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  if (dtype == ME_F32)
    hipLaunchKernelGGL(k_broadcast_tiled<float>, dim3((unsigned)blocks), dim3(kBlockThreads), 0, stream, (float *)dst,
                       (const float *)entry_values, entries, n_padded);
  else
    hipLaunchKernelGGL(k_broadcast_tiled<double>, dim3((unsigned)blocks), dim3(kBlockThreads), 0, stream, (double *)dst,
                       (const double *)entry_values, entries, n_padded);
  return hipGetLastError();
}

// One block sums the per-wavefront acceptance slots (at most a few 10^4 of them): 1024 threads, eight independent
// loads in flight per thread (the first version's dependent one-load-per-iteration loop took 12-30 us for 2^14 slots).
constexpr int kSumThreads = 1024;
__global__ void __launch_bounds__(kSumThreads) k_sum_slots(const unsigned long long *slots, long long n_slots,
                                                           unsigned long long *total) {
  __shared__ unsigned long long part[kSumThreads / 64];
  unsigned long long s = 0;
  long long i = threadIdx.x;
  for (; i + 7 * kSumThreads < n_slots; i += 8 * kSumThreads) {
    unsigned long long v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = slots[i + u * kSumThreads];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; i < n_slots; i += kSumThreads) s += slots[i];
  // wavefront sum by butterfly, then one value per wavefront through LDS
#pragma unroll
  for (int w = 32; w > 0; w >>= 1) s += __shfl_xor(s, w, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
#pragma unroll
    for (int w = 0; w < kSumThreads / 64; ++w) t += part[w];
    total[0] = t;
  }
}

template <typename R>
__global__ void __launch_bounds__(kBlockThreads) k_trace(const R *x, const R *energy, const R *width, long long n, int d,
                                                         int n_terms, int width_rows, long long n_traced, long long stride,
                                                         int tiled, double *out) {
  const long long t = (long long)blockIdx.x * kBlockThreads + threadIdx.x;
  if (t >= n_traced) return;
  const long long c = t * stride;
  for (int k = 0; k < d; ++k)
    out[(long long)k * n_traced + t] = (double)x[tiled ? ((c >> 6) * (long long)d + k) * 64 + (c & 63) : (long long)k * n + c];
  for (int k = 0; k < n_terms; ++k) out[(long long)(d + k) * n_traced + t] = (double)energy[(long long)k * n + c];
  for (int r = 0; r < width_rows; ++r)
    out[(long long)(d + n_terms + r) * n_traced + t] = (double)width[(long long)r * n + c];
}

hipError_t launch_trace(const void *x, const void *energy, const void *width, long long n, int d, int n_terms,
                        int width_rows, int dtype, long long n_traced, long long stride, double *out, hipStream_t stream,
                        bool tiled_state) {
  const dim3 grid((unsigned)((n_traced + kBlockThreads - 1) / kBlockThreads)), block(kBlockThreads);
  if (dtype == ME_F32)
    hipLaunchKernelGGL(k_trace<float>, grid, block, 0, stream, (const float *)x, (const float *)energy, (const float *)width,
                       n, d, n_terms, width_rows, n_traced, stride, tiled_state ? 1 : 0, out);
  else
    hipLaunchKernelGGL(k_trace<double>, grid, block, 0, stream, (const double *)x, (const double *)energy,
                       (const double *)width, n, d, n_terms, width_rows, n_traced, stride, tiled_state ? 1 : 0, out);
  return hipGetLastError();
}

hipError_t launch_sum_slots(const unsigned long long *slots, long long n_slots, unsigned long long *total,
                            hipStream_t stream) {
  hipLaunchKernelGGL(k_sum_slots, dim3(1), dim3(kSumThreads), 0, stream, slots, n_slots, total);
  return hipGetLastError();
}

// chains per staged tile: the widest of 64 / 32 / 16 whose (D + nr + nc) rows fit the LDS of a workgroup; 0 = none does
constexpr size_t kPoolLdsLimit = 160 * 1024;
int pool_tile_chains(int nr, int nc, int dtype) {
  const int n_aug = nr + 2 * nc + nr + nc;
  const size_t elem = dtype == ME_F32 ? sizeof(float) : sizeof(double);
  for (int tile : {64, 32, 16})
    if ((size_t)n_aug * (size_t)(tile + 1) * elem <= kPoolLdsLimit) return tile;
  return 0;
}

// whether launch_pool_reduce can handle these dimensions (the LDS tile of the first stage; the entry count no longer
// matters: stage 1 runs in passes of 3072 entries)
bool pool_reduce_supported(int nr, int nc, int dtype) { return pool_tile_chains(nr, nc, dtype) != 0; }

int pool_reduce_blocks(long long n, int nr, int nc) {
  const int d = nr + 2 * nc;
  const long long n_entries = 1 + d + nr + nc + (long long)d * (d + 1) / 2;
  long long cap = (32ll << 20) / (8 * n_entries);   // at most 32 MiB of partials
  if (cap > 1024) cap = 1024;
  if (cap < 64) cap = 64;
  // small parameter spaces (one 32-row block of augmented rows): the stage-1 kernels are whole workgroups of 8 / 16
  // wavefronts that write ONE row each (me_pool_gram.h); one workgroup per CU
  if (d + nr + nc <= 32) cap = 256;
  const long long tiles = (n + 63) / 64;
  return (int)(tiles < cap ? (tiles < 1 ? 1 : tiles) : cap);
}

namespace {
template <typename R, int TILE>
hipError_t launch_pool_stage1(const void *x, long long n, int nr, int nc, int n_entries, int blocks, double *partials, hipStream_t stream, bool tiled) {
  const size_t lds = (size_t)(nr + 2 * nc + nr + nc) * (size_t)(TILE + 1) * sizeof(R);
  if (lds > 64 * 1024) {
    static PerDevice<hipError_t> attr;
    int device = 0;
    if (hipError_t rc = hipGetDevice(&device); rc != hipSuccess) return rc;
    const hipError_t rc = attr.get(device, [] {
      return hipFuncSetAttribute((const void *)k_pool_reduce<R, TILE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPoolLdsLimit);
    });
    if (rc != hipSuccess) return rc;
  }
  for (int base = 0; base < n_entries; base += kMaxEntries * kBlockThreads)
    hipLaunchKernelGGL((k_pool_reduce<R, TILE>), dim3((unsigned)blocks), dim3(kBlockThreads), lds, stream, (const R *)x, n, nr, nc, base,
                       tiled ? 1 : 0, partials);
  return hipGetLastError();
}
template <typename R>
hipError_t launch_pool_stage1_tile(int tile, const void *x, long long n, int nr, int nc, int n_entries, int blocks, double *partials, hipStream_t stream, bool tiled) {
  switch (tile) {
    case 64: return launch_pool_stage1<R, 64>(x, n, nr, nc, n_entries, blocks, partials, stream, tiled);
    case 32: return launch_pool_stage1<R, 32>(x, n, nr, nc, n_entries, blocks, partials, stream, tiled);
    case 16: return launch_pool_stage1<R, 16>(x, n, nr, nc, n_entries, blocks, partials, stream, tiled);
    default: return hipErrorInvalidValue;
  }
}
}  // namespace

hipError_t launch_pool_reduce(const void *x, long long n, int nr, int nc, int dtype, const unsigned long long *slots,
                              long long n_slots, double proposed, double *partials, double *out_device,
                              hipStream_t stream,
                              hipError_t (*stage1)(const void *, long long, double *, int, hipStream_t), bool tiled_state) {
  const int d = nr + 2 * nc;
  const int n_aug = d + nr + nc;
  const int n_entries = 1 + n_aug + d * (d + 1) / 2;
  const int tile = pool_tile_chains(nr, nc, dtype);
  if (!stage1 && tile == 0) return hipErrorInvalidValue;
  const int blocks = pool_reduce_blocks(n, nr, nc);
  hipError_t err;
  if (stage1) err = stage1(x, n, partials, blocks, stream);
  else if (dtype == ME_F32) err = launch_pool_stage1_tile<float>(tile, x, n, nr, nc, n_entries, blocks, partials, stream, tiled_state);
  else err = launch_pool_stage1_tile<double>(tile, x, n, nr, nc, n_entries, blocks, partials, stream, tiled_state);
  if (err != hipSuccess) return err;
  hipLaunchKernelGGL(k_pool_finish, dim3((unsigned)((n_entries + 63) / 64 + 1)), dim3(kFinishThreads), 0, stream,
                     (const double *)partials, blocks, nr, nc, slots, n_slots, proposed, out_device);
  return hipGetLastError();
}

}  // namespace me
