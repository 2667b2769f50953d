// Y = M X on the bf16 matrix cores with fp32-level accuracy: both operands are split into three bf16 pieces
// (truncation splits: v = v1 + v2 + v3 exactly captures >= 22 mantissa bits) and the six products whose piece
// indices sum to <= 4 are accumulated in fp32:
//     M X ~ M1 X1 + (M1 X2 + M2 X1) + (M1 X3 + M3 X1 + M2 X2),   neglected terms <= 2^-24 |M| |X|.
// bf16 x bf16 products are exact in fp32, so the only rounding is the fp32 accumulation -- the same as any fp32
// summation order.  Unlike v_mfma_f32_32x32x2_f32 (which occupies the fp32 VALU lanes, me_dense_mfma.h), the bf16
// MFMAs run on the matrix pipe beside VALU work, and 96 v_mfma_f32_32x32x16_bf16 (32 cycles each) replace
// 128 fp32 MFMAs of 64 cycles.
//
// One wavefront = 64 chains, one chain per lane.  Operand maps (guide section 3): lane l (r = l & 31, h = l >> 5) holds
// A[row r][k = 8h + j] and B[k = 8h + j][col r], j = 0..7, of a 32x32x16 product; C/D as for every 32x32 shape.
//   A fragments: prepared on the host in fragment order, [piece][row block][k step][lane][8 bf16], staged in LDS,
//                read with ds_read_b128 (lane-linear, conflict-free).
//   B fragments: a lane packs its own chain's pieces for k = 16s..16s+7 ("lo") and 16s+8..16s+15 ("hi");
//                v_permlane32_swap(lo, hi) turns the pair into the operands of chain blocks 0 and 1.
#pragma once

#include <cstdlib>

#include "me_dense_mfma.h"

namespace me {

using bf16x8 = __attribute__((ext_vector_type(8))) short;

constexpr int kBf16FragWords = 3 * 2 * 4 * 64 * 4;   // 32-bit words of the A fragments: 24 KiB

// top 16 bits of a float (bf16 by truncation) and the exact remainder
__device__ __forceinline__ float bf16_head(float v, float &rest) {
  const float head = __uint_as_float(__float_as_uint(v) & 0xFFFF0000u);
  rest = v - head;
  return head;
}
// two bf16 heads -> one register {lo = a, hi = b}
__device__ __forceinline__ unsigned int pack_bf16(float a, float b) {
  return (__float_as_uint(a) >> 16) | (__float_as_uint(b) & 0xFFFF0000u);
}

struct Bf16Operands {   // this lane's chain, one piece, one k step: "lo" = params 16s..16s+7, "hi" = 16s+8..16s+15
  unsigned int lo[4], hi[4];
};

__device__ __forceinline__ bf16x8 as_frag(const unsigned int (&w)[4]) {
  using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
  u32x4 v = {w[0], w[1], w[2], w[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// y = M x for the wave's 64 columns; x is read through `get(k)` (this lane's X[k]) and the result handed out row by
// row through `emit(row, y[row])` (this lane's chain), so that no second 64-register array is live beside the
// accumulators; frags = image of M's three bf16 pieces in fragment order (stage_bf16_fragments).
// frags3: where the THIRD piece's fragments are read from (normally frags itself; the half-size workgroup variant keeps
// only pieces 1 and 2 in LDS and reads piece 3, used by one product in six, from the global image).
template <class Get, class Emit>
__device__ __forceinline__ void wave_matmul_64_bf16x3(const unsigned int *frags, const unsigned int *frags3, Get &&get,
                                                      Emit &&emit, int lane) {
  f32x16 acc[2][2];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.0f;

  using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    // this lane's X[16s .. 16s+15] split once into its three pieces, packed two bf16 per register
    Bf16Operands op[3];
#pragma unroll
    for (int w = 0; w < 8; ++w) {
      float piece[3][2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        float rest = get(16 * s + 2 * w + t);
#pragma unroll
        for (int q = 0; q < 3; ++q) piece[q][t] = bf16_head(rest, rest);
      }
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const unsigned int packed = pack_bf16(piece[q][0], piece[q][1]);
        if (w < 4) op[q].lo[w] = packed;
        else op[q].hi[w - 4] = packed;
      }
    }
    // chain block 0 <- [lo of lanes 0-31 | hi of lanes 0-31], chain block 1 <- [lo of lanes 32-63 | hi of lanes 32-63]
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        float a = __uint_as_float(op[q].lo[w]), b = __uint_as_float(op[q].hi[w]);
        swap32(a, b);
        op[q].lo[w] = __float_as_uint(a);
        op[q].hi[w] = __float_as_uint(b);
      }
    // products with piece indices qa + qx <= 2 (0-based), small terms first within the k step
#pragma unroll
    for (int qa = 2; qa >= 0; --qa)
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        const unsigned int *src = qa == 2 ? frags3 : frags;
        const u32x4 raw = *reinterpret_cast<const u32x4 *>(src + ((((qa * 2 + mb) * 4 + s) * 64 + lane) << 2));
        const bf16x8 a = __builtin_bit_cast(bf16x8, raw);
#pragma unroll
        for (int qx = 2 - qa; qx >= 0; --qx) {
          acc[mb][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, as_frag(op[qx].lo), acc[mb][0], 0, 0, 0);
          acc[mb][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, as_frag(op[qx].hi), acc[mb][1], 0, 0, 0);
        }
      }
    __builtin_amdgcn_sched_barrier(0);   // keep the next k step's operand loads from piling up in registers
  }
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float lo = acc[mb][0][r], hi = acc[mb][1][r];
      swap32(lo, hi);                       // lo: own chain, row acc_row(r); hi: own chain, row acc_row(r) + 4
      emit(32 * mb + acc_row(r), lo);
      emit(32 * mb + acc_row(r) + 4, hi);
    }
}

// Split the row-major fp32 matrix m[64][64] into its three bf16 pieces, in fragment order, into LDS.
__device__ __forceinline__ void stage_bf16_fragments(unsigned int *lds_frag, const float *__restrict__ m) {
  unsigned short *dst = reinterpret_cast<unsigned short *>(lds_frag);
  for (int idx = threadIdx.x; idx < 4096; idx += blockDim.x) {
    const int i = idx >> 6, k = idx & 63;
    const int mb = i >> 5, s = k >> 4, lane = 32 * ((k >> 3) & 1) + (i & 31), j = k & 7;
    float rest = m[idx];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const float head = bf16_head(rest, rest);
      dst[((((q * 2 + mb) * 4 + s) * 64 + lane) << 3) + j] = (unsigned short)(__float_as_uint(head) >> 16);
    }
  }
}

// The dense-64 sweep on the bf16 matrix pipe.  Same chain-per-lane geometry, stream contract and epilogue as
// k_step_dense64_mfma; the proposals x' are parked in LDS between their production and the dot product / commit.
// A's fragments live in LDS (24 KiB + 128 KiB of proposals), copied per block from the image that
// k_dense64_bf16_fragments prepares once per engine.  CK_SHARED has no LDS left for the factor's fragments: L's are
// read from its global image (24 KiB, L1/L2-resident).
// THREADS = 512: one workgroup per CU (24 KiB of fragments + 128 KiB of proposals).  THREADS = 256: pieces 1 and 2 of A
// in LDS (16 KiB) + 64 KiB of proposals = 80 KiB, so that TWO independent workgroups share a CU's 160 KiB and one's
// memory phases overlap the other's arithmetic; piece 3 is read from the global image.
template <int THREADS>
constexpr int dense64_lds_frag_words() { return THREADS == 512 ? kBf16FragWords : (kBf16FragWords / 3) * 2; }
template <int CK, int THREADS>
constexpr size_t dense64_bf16_lds_bytes() {
  return sizeof(float) * (dense64_lds_frag_words<THREADS>() + 64 * THREADS);
}

template <int UNUSED = 0>
__global__ void k_dense64_bf16_fragments(const float *__restrict__ m, unsigned int *__restrict__ out) {
  __shared__ unsigned int frag[kBf16FragWords];
  stage_bf16_fragments(frag, m);
  __syncthreads();
  for (int i = threadIdx.x; i < kBf16FragWords; i += blockDim.x) out[i] = frag[i];
}

template <int CK, int THREADS>
__global__ void __launch_bounds__(THREADS, 2) k_step_dense64_bf16x3(StepArgs<float> a,
                                                                             const unsigned int *__restrict__ afrag,
                                                                             const unsigned int *lfrag) {
  constexpr int D = 64;
  using N_ = Num<float>;
  using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
  extern __shared__ __attribute__((aligned(16))) unsigned int smem_u[];
  unsigned int *lds_a = smem_u;
  constexpr int FRAG_WORDS = dense64_lds_frag_words<THREADS>();
  const unsigned int *a_piece3 = THREADS == 512 ? lds_a : afrag;   // see dense64_bf16_lds_bytes
  float *lds_xp = reinterpret_cast<float *>(smem_u + FRAG_WORDS) + threadIdx.x;   // this lane's column, stride THREADS

  const int lane = threadIdx.x & 63;
  unsigned int wave_accepted = 0;
  bool bad_energy = false, bad_width = false;
  const long long stride = (long long)gridDim.x * THREADS;
  const XField<float, D> fx(a.x, a.n);      // tile-major: a wavefront's 64 rows are one contiguous 16 KiB block
  const Field<float> fe(a.energy, a.n, 1), fw(a.width, a.n, 1);

  // every lane stays active (MFMA and permlane need the whole wavefront): tail lanes shadow the last chain
  long long base = (long long)blockIdx.x * THREADS + (threadIdx.x & ~63);
  bool have = base < a.n;
  bool live = false;
  unsigned int coff = 0, xoff = 0;
  unsigned long long gid = 0;
  float x[D], e = 0.0f, w = 0.0f;
  // Loads are issued in the order the first sweep consumes them (width, then rows 0, 1, 2, ...): memory returns in
  // order, so the s_waitcnt before the first use of row 4b can leave the later rows in flight behind the Philox
  // work.  Left to the scheduler the rows were issued scrambled and the first use waited for (almost) all of them.
  auto load_tile = [&]() {
    const long long c_raw = base + lane;
    live = c_raw < a.n;
    const long long c = live ? c_raw : a.n - 1;
    coff = (unsigned int)c * 4u;
    xoff = fx.offset(c);
    gid = a.chain_offset + (unsigned long long)c;
    w = fw.load(0, coff);
    e = fe.load(0, coff);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int d = 0; d < D; ++d) {
      x[d] = fx.load(d, xoff);
      if ((d & 7) == 7) __builtin_amdgcn_sched_barrier(0);
    }
  };
  // A's fragment image (24 KiB, prepared once per engine by k_dense64_bf16_fragments) is requested first and the first
  // tile's state right behind it: the image lands, goes to LDS and the block passes its barrier while the state rows
  // are still arriving
  constexpr int PER_THREAD = FRAG_WORDS / 4 / THREADS;   // 16-byte pieces of the image per thread: 3 (512) or 4 (256)
  static_assert(PER_THREAD * THREADS * 4 == FRAG_WORDS, "the image divides evenly over the workgroup");
  u32x4 image[PER_THREAD];
#pragma unroll
  for (int k = 0; k < PER_THREAD; ++k) image[k] = reinterpret_cast<const u32x4 *>(afrag)[k * THREADS + threadIdx.x];
  __builtin_amdgcn_sched_barrier(0);
  load_tile();   // unconditional (a wave past the end shadows the last chain): a branch here would make the
                 // s_waitcnt in front of the LDS writes below wait for the tile as well
#pragma unroll
  for (int k = 0; k < PER_THREAD; ++k) reinterpret_cast<u32x4 *>(lds_a)[k * THREADS + threadIdx.x] = image[k];
  __syncthreads();

  while (have) {
    for (int s = 0; s < a.n_sweeps; ++s) {
      const unsigned long long step = a.step_index + (unsigned long long)s;
      U4 ctr;
      ctr.x = (uint32_t)gid;
      ctr.y = (uint32_t)(gid >> 32);
      ctr.z = (uint32_t)step;
      const uint32_t step_hi = (uint32_t)(step >> 32) << 16;
      if constexpr (CK == CK_SHARED) {
#pragma unroll
        for (int b = 0; b < 16; ++b) {
          ctr.w = step_hi | (uint32_t)b;
          const U4 r = philox4x32_10(ctr, a.seed_lo, a.seed_hi);
          float g[4];
          N_::normal_pair(r.x, r.y, g[0], g[1]);
          N_::normal_pair(r.z, r.w, g[2], g[3]);
#pragma unroll
          for (int i = 0; i < 4; ++i) lds_xp[(4 * b + i) * THREADS] = g[i];
        }
        // x' = x + w L g; every lane has read all of g before the first row comes out.  The clobber keeps the
        // (loop-invariant) fragment loads of L inside the sweep: hoisted, they would pin 96 registers.
        asm volatile("" ::: "memory");
        wave_matmul_64_bf16x3(lfrag, lfrag, [&](int k) { return lds_xp[k * THREADS]; },
                              [&](int row, float v) { lds_xp[row * THREADS] = x[row] + w * v; }, lane);
      } else {
#pragma unroll
        for (int b = 0; b < 16; ++b) {
          ctr.w = step_hi | (uint32_t)b;
          const U4 r = philox4x32_10(ctr, a.seed_lo, a.seed_hi);
          float g[4];
          N_::normal_pair(r.x, r.y, g[0], g[1]);
          N_::normal_pair(r.z, r.w, g[2], g[3]);
#pragma unroll
          for (int i = 0; i < 4; ++i) lds_xp[(4 * b + i) * THREADS] = x[4 * b + i] + w * g[i];
        }
      }
      float e_new = 0.0f;                                   // E = x'^T (A x')
      wave_matmul_64_bf16x3(lds_a, a_piece3, [&](int k) { return lds_xp[k * THREADS]; },
                            [&](int row, float v) { e_new += lds_xp[row * THREADS] * v; }, lane);
      ctr.w = step_hi | 16u;                                // word 64 = block 16, output 0
      const float u = N_::unit(philox4x32_10(ctr, a.seed_lo, a.seed_hi).x);
      bool rejected = false;
      if (a.reject_kind == ME_REJECT_ABS_REAL0_GE) rejected = !(N_::abs_(lds_xp[0]) < a.reject_bound);
      const float diff = e_new - e;
      bool accept = diff <= 0.0f;
      if (a.temp > 0.0f) accept = accept || N_::uphill(u, diff, a.inv_temp, a.inv_temp_log2e);
      accept = accept && !rejected;
      bad_energy |= (live && !rejected && !N_::finite(e_new));
      if (accept) {
#pragma unroll
        for (int d = 0; d < D; ++d) x[d] = lds_xp[d * THREADS];
      }
      e = accept ? e_new : e;
      w = N_::adapt(w, accept, a.ratio, a.p, a.damping, a.up, a.down);
      wave_accepted += (unsigned int)__popcll(__ballot(accept && live));
    }
    bad_width |= live && !(w > 0.0f);
    if (live) {
#pragma unroll
      for (int d = 0; d < D; ++d) fx.store(d, xoff, x[d]);
      fe.store(0, coff, e);
      fw.store(0, coff, w);
    }
    base += stride;
    have = base < a.n;
    if (have) load_tile();
  }
  if (lane == 0 && wave_accepted) {
    unsigned long long *slot = a.accept_slots + (size_t)blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6);
    *slot += (unsigned long long)wave_accepted;
  }
  const unsigned int bits = (bad_energy ? ST_NONFINITE_ENERGY : 0u) | (bad_width ? ST_BAD_WIDTH : 0u);
  if (bits) atomicOr(a.status, bits);
}

// METROPOLIS_DENSE64_FP32_MFMA=1 selects the fp32 MFMA kernel instead (read once per process).
inline bool dense64_exact_fp32_mfma() {
  static const bool on = [] {
    const char *v = std::getenv("METROPOLIS_DENSE64_FP32_MFMA");
    return v && v[0] == '1';
  }();
  return on;
}

// Host launcher.  afrag / lfrag (CK_SHARED): fragment images of A / the factor made by k_dense64_bf16_fragments.
template <int CK, int THREADS>
inline hipError_t launch_step_dense64_bf16x3_threads(const StepArgs<float> &a, const unsigned int *afrag,
                                                     const unsigned int *lfrag, int grid_blocks, hipStream_t stream) {
  // per DEVICE of the process, not per process (me_per_device.h): the raised dynamic-LDS limit and the CU count
  static PerDevice<hipError_t> attr_cache;
  static PerDevice<int> cu_cache;
  int device = 0;
  hipError_t rc = hipGetDevice(&device);
  if (rc != hipSuccess) return rc;
  rc = attr_cache.get(device, [] {
    return hipFuncSetAttribute((const void *)k_step_dense64_bf16x3<CK, THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)dense64_bf16_lds_bytes<CK, THREADS>());
  });
  if (rc != hipSuccess) return rc;
  // Persistent workgroups, four wavefront-slots' worth per SIMD pair (two 512-thread or four 256-thread groups per CU),
  // each striding over its tiles with the next tile's loads issued before the current one retires: beats one group per
  // tile by 12 % at one sweep per launch (start-up: image copy + barrier with nothing else resident) and ties when
  // sweeps are fused.  tools/dev/time_dense64_grid.py
  const int cus = cu_cache.get(device, [device] {
    int count = 0;
    if (hipDeviceGetAttribute(&count, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || count <= 0) return 256;
    return count;
  });
  long long blocks = (a.n + THREADS - 1) / THREADS;
  const long long cap = grid_blocks > 0 ? grid_blocks : (long long)cus * (1024 / THREADS);
  if (blocks > cap) blocks = cap;
  constexpr size_t lds = dense64_bf16_lds_bytes<CK, THREADS>();
  hipLaunchKernelGGL((k_step_dense64_bf16x3<CK, THREADS>), dim3((unsigned)blocks), dim3(THREADS), lds, stream, a, afrag,
                     lfrag);
  return hipGetLastError();
}

// Default: the half-size workgroups, two per CU (63 -> 60 us per sweep at one sweep per launch, 36.4 -> 35.1 fused, 2^19
// chains); METROPOLIS_DENSE64_THREADS=512 selects one 512-thread workgroup per CU.
template <int CK>
inline hipError_t launch_step_dense64_bf16x3(const StepArgs<float> &a, const unsigned int *afrag, const unsigned int *lfrag,
                                             int grid_blocks, hipStream_t stream) {
  static const bool full = [] {
    const char *v = std::getenv("METROPOLIS_DENSE64_THREADS");
    return v && std::atoi(v) == 512;
  }();
  if (full) return launch_step_dense64_bf16x3_threads<CK, 512>(a, afrag, lfrag, grid_blocks, stream);
  return launch_step_dense64_bf16x3_threads<CK, 256>(a, afrag, lfrag, grid_blocks, stream);
}

}  // namespace me
