// k_step for the dense quadratic-form energy E = x^T A x on 64 real parameters (BASELINE config 4), with the
// quadratic form -- and, for a shared (pooled) proposal factor, the proposal L g -- on the matrix cores.
//
// Reference semantics are those of k_step (me_device.h): step_real_group, metropolis_engine.py:225-239 with the
// user energy at :231 being x^T A x.  Only the evaluation strategy differs:
//
//   * one wavefront = 64 chains; Y = A X' for the wave's 64 proposals is a 64x64x64 GEMM done as
//     2 (row blocks) x 2 (chain blocks) x 32 (k pairs) v_mfma_f32_32x32x2_f32.  f32-input MFMA is exact fp32 (a
//     k-ordered fma chain) at the fp32 VALU rate, but it runs on the matrix pipe, so the 8k cycles of FMAs per
//     sweep overlap the Philox / Box-Muller VALU work instead of adding to it.
//   * B operand (k pair of proposals for 32 chains): lane l needs X'[k0 + (l>>5)][chain 32 nb + (l&31)].  With one
//     chain per lane, ONE v_permlane32_swap of (x'[k0], x'[k0+1]) yields the operands of both chain blocks.
//   * A operand: fragments A[32 mb + (l&31)][2 kp + (l>>5)] staged once per block in LDS in fragment order
//     (lane-linear, conflict-free ds_read_b32).
//   * the accumulators hold rows {r, r+4} of a chain split over lanes l and l^32; one v_permlane32_swap per
//     accumulator register pair brings both halves home, then E = sum_p x'_p y_p is 64 FMAs per lane.
#pragma once

#include "me_device.h"

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
  for (int idx = threadIdx.x; idx < 4096; idx += kBlockThreads) {
    const int kp = idx >> 7, mb = (idx >> 6) & 1, lane = idx & 63;
    lds[idx] = m[(32 * mb + (lane & 31)) * 64 + 2 * kp + (lane >> 5)];
  }
}

// Y = M X for the wave's 64 columns; X given one column (chain) per lane as v[64].  Result comes back one chain
// per lane as y[64] (rows in natural order).
__device__ __forceinline__ void wave_matmul_64(const float *lds_frag, const float (&v)[64], float (&y)[64], int lane,
                                               bool lower_triangular) {
  f32x16 acc[2][2];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.0f;
#pragma unroll
  for (int kp = 0; kp < 32; ++kp) {
    float b0 = v[2 * kp], b1 = v[2 * kp + 1];
    swap32(b0, b1);
    const float a1 = lds_frag[(kp * 2 + 1) * 64 + lane];
    if (!(lower_triangular && kp >= 16)) {   // rows 0..31 of a lower-triangular matrix have no columns >= 32
      const float a0 = lds_frag[(kp * 2 + 0) * 64 + lane];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
    }
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
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

// CK: CK_IDENTITY or CK_SHARED (factor = packed lower triangle of the shared 64x64 Cholesky factor).
template <int CK>
__global__ void __launch_bounds__(kBlockThreads) k_step_dense64_mfma(StepArgs<float> a, const float *__restrict__ amat,
                                                                     const float *__restrict__ lfull) {
  constexpr int D = 64;
  constexpr int NBLK = 17;   // 16 Philox blocks of normals + the block holding the accept uniform (word 64)
  using N_ = Num<float>;
  __shared__ float lds_a[4096];
  __shared__ float lds_l[CK == CK_SHARED ? 4096 : 1];
  stage_a_fragments(lds_a, amat);
  if constexpr (CK == CK_SHARED) stage_a_fragments(lds_l, lfull);
  __syncthreads();

  const int lane = threadIdx.x & 63;
  unsigned int wave_accepted = 0;
  bool bad_energy = false, bad_width = false;
  const long long stride = (long long)gridDim.x * kBlockThreads;
  // every lane stays active (MFMA and permlane need the whole wavefront): tail lanes shadow the last chain
  for (long long base = (long long)blockIdx.x * kBlockThreads + (threadIdx.x & ~63); base < a.n; base += stride) {
    const long long c_raw = base + lane;
    const bool live = c_raw < a.n;
    const long long c = live ? c_raw : a.n - 1;
    float x[D];
#pragma unroll
    for (int d = 0; d < D; ++d) x[d] = a.x[(long long)d * a.n + c];
    float e = a.energy[c];
    float w = a.width[c];
    const unsigned long long gid = a.chain_offset + (unsigned long long)c;

    for (int s = 0; s < a.n_sweeps; ++s) {
      const unsigned long long step = a.step_index + (unsigned long long)s;
      float g[D];
      float u = 0.0f;
#pragma unroll
      for (int b = 0; b < NBLK; ++b) {
        U4 ctr;
        ctr.x = (uint32_t)gid;
        ctr.y = (uint32_t)(gid >> 32);
        ctr.z = (uint32_t)step;
        ctr.w = ((uint32_t)(step >> 32) << 16) | (uint32_t)b;
        const U4 o = philox4x32_10(ctr, a.seed_lo, a.seed_hi);
        if (b < 16) {
          N_::normal_pair(o.x, o.y, g[4 * b], g[4 * b + 1]);
          N_::normal_pair(o.z, o.w, g[4 * b + 2], g[4 * b + 3]);
        } else {
          u = N_::unit(o.x);
        }
      }
      float xp[D];
      if constexpr (CK == CK_SHARED) {
        float z[D];
        wave_matmul_64(lds_l, g, z, lane, true);     // z = L g
#pragma unroll
        for (int d = 0; d < D; ++d) xp[d] = x[d] + w * z[d];
      } else {
#pragma unroll
        for (int d = 0; d < D; ++d) xp[d] = x[d] + w * g[d];
      }
      bool rejected = false;
      if (a.reject_kind == ME_REJECT_ABS_REAL0_GE) rejected = !(N_::abs_(xp[0]) < a.reject_bound);
      float y[D];
      wave_matmul_64(lds_a, xp, y, lane, false);     // y = A x'
      float e_new = 0.0f;
#pragma unroll
      for (int d = 0; d < D; ++d) e_new += xp[d] * y[d];
      const float diff = e_new - e;
      bool accept = diff <= 0.0f;
      if (a.temp > 0.0f) accept = accept || N_::uphill(u, diff, a.temp, a.inv_temp_log2e);
      accept = accept && !rejected;
      bad_energy |= (live && !rejected && !N_::finite(e_new));
#pragma unroll
      for (int d = 0; d < D; ++d) x[d] = accept ? xp[d] : x[d];
      e = accept ? e_new : e;
      w = N_::adapt(w, accept, a.ratio, a.p, a.damping, a.up, a.down);
      wave_accepted += (unsigned int)__popcll(__ballot(accept && live));
    }
    bad_width |= live && !(w > 0.0f);
    if (live) {
#pragma unroll
      for (int d = 0; d < D; ++d) a.x[(long long)d * a.n + c] = x[d];
      a.energy[c] = e;
      a.width[c] = w;
    }
  }
  if (lane == 0 && wave_accepted) {
    unsigned long long *slot = a.accept_slots + (size_t)blockIdx.x * (kBlockThreads / 64) + (threadIdx.x >> 6);
    *slot += (unsigned long long)wave_accepted;
  }
  const unsigned int bits = (bad_energy ? ST_NONFINITE_ENERGY : 0u) | (bad_width ? ST_BAD_WIDTH : 0u);
  if (bits) atomicOr(a.status, bits);
}

}  // namespace me
