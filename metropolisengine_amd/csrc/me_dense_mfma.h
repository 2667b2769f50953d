// k_step for the dense quadratic-form energy E = x^T A x on 64 real parameters (BASELINE config 4), with the
// quadratic form -- and, for a shared (pooled) proposal factor, the proposal L g -- on the matrix cores.
//
// Reference semantics are those of k_step (me_device.h): step_real_group, metropolis_engine.py:225-239 with the
// user energy at :231 being x^T A x.  Only the evaluation strategy differs:
//
//   * one wavefront = 64 chains; Y = A X' for the wave's 64 proposals is a 64x64x64 GEMM done as
//     2 (row blocks) x 2 (chain blocks) x 32 (k pairs) v_mfma_f32_32x32x2_f32.  f32-input MFMA is exact fp32 (a
//     k-ordered fma chain).  MEASURED on MI355X (tools/dev/mfma_valu_overlap.hip): it does NOT run beside VALU
//     work -- an MFMA-only loop, a VALU-only loop and their interleave take 0.98 / 2.38 / 3.44 ms (1 wave per
//     SIMD) and 1.82 / 2.38 / 4.40 ms (2 per SIMD): the times add.  The fp32 matrix instruction occupies the
//     SIMD's fp32 lanes, so its value here is not concurrency but density: 128 instructions and 64 accumulator
//     registers replace 4096 FMAs with 4096 scalar-loaded coefficients (the VALU form of this kernel needed
//     16 KB of scratch per lane and ran 50x slower).
//   * B operand (k pair of proposals for 32 chains): lane l needs X'[k0 + (l>>5)][chain 32 nb + (l&31)].  With one
//     chain per lane, ONE v_permlane32_swap of (x'[k0], x'[k0+1]) yields the operands of both chain blocks.
//   * A operand: fragments A[32 mb + (l&31)][2 kp + (l>>5)] staged once per block in LDS in fragment order
//     (lane-linear, conflict-free ds_read_b32).
//   * the accumulators hold rows {r, r+4} of a chain split over lanes l and l^32; one v_permlane32_swap per
//     accumulator register pair brings both halves home, then E = sum_p x'_p y_p is 64 FMAs per lane.
#pragma once

#include "me_device.h"
#include "me_per_device.h"

namespace me {

using f32x16 = __attribute__((ext_vector_type(16))) float;

__device__ __forceinline__ void swap32(float &a, float &b) {
  // after: a = [a.lanes0-31 | b.lanes0-31], b = [a.lanes32-63 | b.lanes32-63]
  // (extract the two results into scalar locals first: __builtin_bit_cast(float, r[i]) straight on the builtin's
  //  vector result miscompiles on hipcc / ROCm 7.2 -- both outputs come back as r[0])
  const unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
  const auto r = __builtin_amdgcn_permlane32_swap(ua, ub, false, false);
  const unsigned r0 = r[0], r1 = r[1];
  a = __uint_as_float(r0);
  b = __uint_as_float(r1);
}

// row of accumulator register `reg` (0..15) for lane half h: (reg&3) + 8*(reg>>2) + 4*h   (guide section 3)
__host__ __device__ constexpr int acc_row(int reg) { return (reg & 3) + 8 * (reg >> 2); }

// Fill `lds` (4096 floats) with the fragments of a 64x64 row-major matrix M for use as the MFMA A operand:
// lds[(kp*2 + mb)*64 + lane] = M[32 mb + (lane&31)][2 kp + (lane>>5)].
__device__ __forceinline__ void stage_a_fragments(float *lds, const float *__restrict__ m) {
  for (int idx = threadIdx.x; idx < 4096; idx += blockDim.x) {
    const int kp = idx >> 7, mb = (idx >> 6) & 1, lane = idx & 63;
    lds[idx] = m[(32 * mb + (lane & 31)) * 64 + 2 * kp + (lane >> 5)];
  }
}

// Y = M X for the wave's 64 columns (one chain per lane).
//
// `stage(b, t, v)` performs slice t (0..7) of the work that produces this lane's X[4b..4b+3] into v[4]; the eight
// slices of block b+1 are issued one behind each of the 8 MFMAs of block b (k pairs 2b, 2b+1), pinned with
// sched_barrier.  For the first product the slices are: 5 x two Philox rounds, 2 x one Box-Muller pair, 1 x finish.
// Producing the operands just in time keeps at most two blocks of normals live (the kernel fits 256 registers = 2
// wavefronts per SIMD, so the other wavefront's global loads/stores overlap this one's arithmetic); it does not
// make the fp32 MFMAs overlap the VALU work (see the header note).
// Result: y[64], rows in natural order, one chain per lane.
template <class Stage>
__device__ __forceinline__ void wave_matmul_64(const float *lds_frag, Stage &&stage, float (&y)[64], int lane,
                                               bool lower_triangular) {
  f32x16 acc[2][2];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.0f;
  float cur[4];
#pragma unroll
  for (int t = 0; t < 8; ++t) stage(0, t, cur);
  swap32(cur[0], cur[1]);   // -> B operands of k pair 0 for chain blocks 0 / 1
  swap32(cur[2], cur[3]);   // -> k pair 1
#pragma unroll
  for (int b = 0; b < 16; ++b) {
    float nxt[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      // MFMA q of this block: k pair 2b + (q >> 2), row block (q >> 1) & 1, chain block q & 1
      const int kp = 2 * b + (q >> 2), mb = (q >> 1) & 1, nb = q & 1;
      if (!(lower_triangular && mb == 0 && kp >= 16)) {   // rows 0..31 of a lower-triangular M: no columns >= 32
        const float a_frag = lds_frag[(kp * 2 + mb) * 64 + lane];
        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_frag, cur[2 * (q >> 2) + nb], acc[mb][nb], 0, 0, 0);
      }
      if (b < 15) stage(b + 1, q, nxt);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (b < 15) {
      swap32(nxt[0], nxt[1]);
      swap32(nxt[2], nxt[3]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) cur[i] = nxt[i];
  }
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float lo = acc[mb][0][r], hi = acc[mb][1][r];
      swap32(lo, hi);                       // lo: own chain, row acc_row(r); hi: own chain, row acc_row(r) + 4
      y[32 * mb + acc_row(r)] = lo;
      y[32 * mb + acc_row(r) + 4] = hi;
    }
}

// Two Philox4x32 rounds on a running (counter, key) state: the sliced form of philox4x32_10 (me_device.h).
__device__ __forceinline__ void philox_two_rounds(U4 &c, uint32_t &k0, uint32_t &k1) {
#pragma unroll
  for (int r = 0; r < 2; ++r) {
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
}

// CK: CK_IDENTITY or CK_SHARED (lfull = the shared 64x64 Cholesky factor as a dense lower-triangular matrix).
//
// Geometry: 512-thread blocks (8 wavefronts = 2 per SIMD) with ONE copy of the A (and L) fragments per block and the
// proposals x' parked in LDS ([64][512] floats, lane-linear) between their production and the final dot product /
// commit.  That keeps the kernel at <= 256 registers, i.e. two resident wavefronts per SIMD: while one is in its
// MFMA-heavy k loop the other's Philox / Box-Muller VALU work issues, and loads of the next tile overlap both.
constexpr int kDenseBlockThreads = 512;
template <int CK>
constexpr size_t dense64_lds_bytes() {
  return sizeof(float) * (4096 * (CK == CK_SHARED ? 2 : 1) + 64 * kDenseBlockThreads);
}

template <int CK>
__global__ void __launch_bounds__(kDenseBlockThreads, 2) k_step_dense64_mfma(StepArgs<float> a, const float *__restrict__ amat,
                                                                           const float *__restrict__ lfull) {
  constexpr int D = 64;
  using N_ = Num<float>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *lds_a = smem;
  float *lds_l = smem + 4096;
  float *lds_xp = smem + 4096 * (CK == CK_SHARED ? 2 : 1) + threadIdx.x;   // this lane's column, stride 512
  stage_a_fragments(lds_a, amat);
  if constexpr (CK == CK_SHARED) stage_a_fragments(lds_l, lfull);
  __syncthreads();

  const int lane = threadIdx.x & 63;
  unsigned int wave_accepted = 0;
  bool bad_energy = false, bad_width = false;
  const long long stride = (long long)gridDim.x * kDenseBlockThreads;
  const XField<float, D> fx(a.x, a.n);
  const Field<float> fe(a.energy, a.n, 1), fw(a.width, a.n, 1);
  // every lane stays active (MFMA and permlane need the whole wavefront): tail lanes shadow the last chain
  for (long long base = (long long)blockIdx.x * kDenseBlockThreads + (threadIdx.x & ~63); base < a.n; base += stride) {
    const long long c_raw = base + lane;
    const bool live = c_raw < a.n;
    const long long c = live ? c_raw : a.n - 1;
    const unsigned int coff = (unsigned int)c * 4u, xoff = fx.offset(c);
    float x[D];
#pragma unroll
    for (int d = 0; d < D; ++d) x[d] = fx.load(d, xoff);
    float e = fe.load(0, coff);
    float w = fw.load(0, coff);
    const unsigned long long gid = a.chain_offset + (unsigned long long)c;

    for (int s = 0; s < a.n_sweeps; ++s) {
      const unsigned long long step = a.step_index + (unsigned long long)s;
      // Philox block b yields the normals of parameters 4b..4b+3 (k pairs 2b and 2b+1 of the first product);
      // it is produced in eight slices, one behind each MFMA of the previous block.
      U4 pc;
      uint32_t pk0 = 0, pk1 = 0;
      auto draw_slice = [&](int b, int t, float (&g)[4]) {
        if (t == 0) {
          pc.x = (uint32_t)gid;
          pc.y = (uint32_t)(gid >> 32);
          pc.z = (uint32_t)step;
          pc.w = ((uint32_t)(step >> 32) << 16) | (uint32_t)b;
          pk0 = a.seed_lo;
          pk1 = a.seed_hi;
        }
        if (t < 5) philox_two_rounds(pc, pk0, pk1);
        else if (t == 5) N_::normal_pair(pc.x, pc.y, g[0], g[1]);
        else if (t == 6) N_::normal_pair(pc.z, pc.w, g[2], g[3]);
      };
      float y[D];
      if constexpr (CK == CK_SHARED) {
        wave_matmul_64(lds_l, draw_slice, y, lane, true);      // y = L g, normals drawn on the fly
#pragma unroll
        for (int d = 0; d < D; ++d) lds_xp[d * kDenseBlockThreads] = x[d] + w * y[d];
        wave_matmul_64(lds_a, [&](int b, int t, float (&v)[4]) {
          if (t < 4) v[t] = lds_xp[(4 * b + t) * kDenseBlockThreads];
        }, y, lane, false);
      } else {
        // identity shape: x' = x + w g is formed block by block and fed straight into y = A x'
        wave_matmul_64(lds_a, [&](int b, int t, float (&v)[4]) {
          draw_slice(b, t, v);
          if (t == 7) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              v[i] = x[4 * b + i] + w * v[i];
              lds_xp[(4 * b + i) * kDenseBlockThreads] = v[i];
            }
          }
        }, y, lane, false);
      }
      float u;
      {
        U4 ctr;
        ctr.x = (uint32_t)gid;
        ctr.y = (uint32_t)(gid >> 32);
        ctr.z = (uint32_t)step;
        ctr.w = ((uint32_t)(step >> 32) << 16) | 16u;     // word 64 = block 16, output 0
        u = N_::unit(philox4x32_10(ctr, a.seed_lo, a.seed_hi).x);
      }
      float e_new = 0.0f;
#pragma unroll
      for (int d = 0; d < D; ++d) e_new += lds_xp[d * kDenseBlockThreads] * y[d];
      bool rejected = false;
      if (a.reject_kind == ME_REJECT_ABS_REAL0_GE) rejected = !(N_::abs_(lds_xp[0]) < a.reject_bound);
      const float diff = e_new - e;
      bool accept = diff <= 0.0f;
      if (a.temp > 0.0f) accept = accept || N_::uphill(u, diff, a.inv_temp, a.inv_temp_log2e);
      accept = accept && !rejected;
      bad_energy |= (live && !rejected && !N_::finite(e_new));
      if (accept) {
#pragma unroll
        for (int d = 0; d < D; ++d) x[d] = lds_xp[d * kDenseBlockThreads];
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
  }
  if (lane == 0 && wave_accepted) {
    unsigned long long *slot = a.accept_slots + (size_t)blockIdx.x * (kDenseBlockThreads / 64) + (threadIdx.x >> 6);
    *slot += (unsigned long long)wave_accepted;
  }
  const unsigned int bits = (bad_energy ? ST_NONFINITE_ENERGY : 0u) | (bad_width ? ST_BAD_WIDTH : 0u);
  if (bits) atomicOr(a.status, bits);
}

// Host launcher: 512-thread blocks, dynamic LDS above 64 KiB needs the function attribute once.
template <int CK>
inline hipError_t launch_step_dense64_mfma(const StepArgs<float> &a, const float *amat, const float *lfull, int grid_blocks,
                                           hipStream_t stream) {
  static PerDevice<hipError_t> attr_cache;      // per device of the process (me_per_device.h)
  int device = 0;
  hipError_t rc = hipGetDevice(&device);
  if (rc != hipSuccess) return rc;
  rc = attr_cache.get(device, [] {
    return hipFuncSetAttribute((const void *)k_step_dense64_mfma<CK>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)dense64_lds_bytes<CK>());
  });
  if (rc != hipSuccess) return rc;
  long long blocks = (a.n + kDenseBlockThreads - 1) / kDenseBlockThreads;
  if (grid_blocks > 0 && blocks > grid_blocks) blocks = grid_blocks;
  hipLaunchKernelGGL(k_step_dense64_mfma<CK>, dim3((unsigned)blocks), dim3(kDenseBlockThreads), dense64_lds_bytes<CK>(),
                     stream, a, amat, lfull);
  return hipGetLastError();
}

}  // namespace me
