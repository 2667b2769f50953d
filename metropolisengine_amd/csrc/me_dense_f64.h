// k_step for the dense quadratic-form energy E = x^T A x on 64 real parameters in FLOAT64 -- BASELINE config 4 at
// the reference's precision -- with the quadratic form, and for a shared (pooled) proposal factor the proposal L g,
// on the matrix cores (v_mfma_f64_16x16x4_f64).
//
// Reference semantics are those of k_step (me_device.h): step_real_group, metropolis_engine.py:225-239 with the user
// energy at :231 being x^T A x, proposals x' = x + sigma L g (:261-272).  Only the evaluation strategy differs.
//
// What the matrix instruction is for here.  MEASURED on MI355X (tools/dev/mfma_f64_probe.hip): one
// v_mfma_f64_16x16x4_f64 (2 048 flop) issues every ~64-79 cycles per SIMD, i.e. exactly the v_fma_f64 rate
// (64 lanes x 2 flop per 4 cycles), and an MFMA-only loop, a float64-VALU loop and their interleave take
// 1.23 / 0.60 / 1.77 ms -- the times ADD, in one wave and across two waves of a SIMD.  So it buys no flops and no
// overlap with the Philox / Box-Muller work; it buys DENSITY: the generic k_step<double,64,0,EnergyDense> needs 4 096
// v_fma_f64 with 4 096 scalar-loaded coefficients per chain-step and 256 live registers of state beside them, and
// compiled to 32-49 KB of scratch per lane (16 ms per launch).  Here 160 instructions and 32 accumulator registers do
// the product, the coefficients are 40 LDS words per lane, and nothing spills.
//
//   * Only the symmetric part of A matters to x^T A x.  T = tril(A + A^T, -1) + diag(A) is LOWER TRIANGULAR and
//     E = x'^T (T x'), so row block mb (16 rows) of the product needs k < 16 (mb + 1) only: 4+8+12+16 = 40 k-steps of
//     4 instead of 64, times 4 chain blocks = 160 MFMAs instead of 256.  The shared factor L is lower triangular by
//     construction and goes through the same code (fold(L) = L).
//   * One wavefront = one tile of 32 chains; a chain's 64 rows are split over TWO lanes (lane = chain + 32 half).  The
//     proposals are parked in LDS as xp[row][chain] (16 KiB per wave, chain index XOR-swizzled by the row's parity so
//     that both the MFMA B-operand reads -- lane l wants row 4 ks + (l >> 4), chain 16 nb + (l & 15) -- and the owners'
//     column accesses are bank-conflict free).
//   * A-operand fragments of T: image[(mb, ks)][lane] = T[16 mb + (l & 15)][4 ks + (l >> 4)], built once per engine
//     (k_dense64_f64_fragments), 20 KiB, copied to LDS once per block.  The factor image (CK_SHARED) stays in global
//     memory (L1/L2 resident, read through a buffer descriptor): LDS is full.
//   * Results: lane l holds rows (l >> 4) + 4 r (r < 4) of chain 16 nb + (l & 15) of a 16 x 16 block (the float64 C/D
//     map differs from every other MFMA's, checked in the probe).  The energy is reduced where the results are:
//     partial dot products against xp from LDS, then an exchange through 1 KiB of LDS brings each chain's total to both
//     of its lanes.  For L g the row blocks are produced in DESCENDING order, so that block mb of the result can
//     overwrite rows 16 mb .. 16 mb + 15 of g in place (later blocks read only lower rows).
//   * Two wavefronts per SIMD (8 x 17 KiB + 20 KiB of the CU's 160 KiB of LDS; <= 256 registers), and the NEXT tile's rows
//     are loaded into a second register set while the current tile computes.  The first version gave a wavefront 64
//     chains -- one wavefront per SIMD -- and took 162 us per one-sweep launch and 110 us per fused sweep at 2^19 chains;
//     this one 135 and 94.  The kernel is balanced between the float64 pipe (~94 us) and HBM (~110 us for its 554 MB,
//     measured with the arithmetic compiled out: -DME_DENSE64_F64_MEMORY_ONLY); issuing the prefetch earlier or staggering
//     the two halves of the workgroup changed nothing (tools/dev/time_dense64_variants.py).
#pragma once

#include "me_device.h"
#include "me_per_device.h"

namespace me {

using f64x4 = __attribute__((ext_vector_type(4))) double;

#ifndef ME_DENSE64_F64_PREFETCH_EARLY
#define ME_DENSE64_F64_PREFETCH_EARLY 0
#endif
#ifndef ME_DENSE64_F64_STAGGER
#define ME_DENSE64_F64_STAGGER 0
#endif
#ifndef ME_DENSE64_F64_RNG_UNROLL
#define ME_DENSE64_F64_RNG_UNROLL 2     // 2 blocks (4 Box-Muller pairs) per iteration: -2.5 % on the fused sweep against 1; 4: no further gain
#endif
#ifndef ME_DENSE64_F64_PRIO
#define ME_DENSE64_F64_PRIO 0
#endif
#ifndef ME_DENSE64_F64_NT
#define ME_DENSE64_F64_NT 0             // experiment: bit 0 = state loads non-temporal, bit 1 = state stores non-temporal
#endif
constexpr int kDense64F64Threads = 512;                  // 8 waves, two per SIMD
constexpr int kDense64F64Frags = 40;                     // (mb, ks) pairs with 4 ks < 16 (mb + 1)
constexpr int kDense64F64ImageDoubles = kDense64F64Frags * 64;
constexpr int kDense64F64WaveDoubles = 64 * 32 + 128;    // xp[64 rows][32 chains] + the 2 x 4 x 16 energy exchange
constexpr size_t kDense64F64LdsBytes = sizeof(double) * (kDense64F64ImageDoubles + 8 * kDense64F64WaveDoubles);

__host__ __device__ constexpr int dense64_f64_frag_index(int mb, int ks) { return 2 * mb * (mb + 1) + ks; }

// image[(mb, ks)][lane] = T[16 mb + (lane & 15)][4 ks + (lane >> 4)],  T = tril(M + M^T, -1) + diag(M)
template <int UNUSED = 0>   // a template so that every kernel-set object may carry it (one definition after linking)
__global__ void k_dense64_f64_fragments(const double *__restrict__ m, double *__restrict__ image) {
  for (int idx = threadIdx.x; idx < kDense64F64ImageDoubles; idx += blockDim.x) {
    const int frag = idx >> 6, lane = idx & 63;
    int mb = 0;
    while (dense64_f64_frag_index(mb + 1, 0) <= frag) ++mb;
    const int ks = frag - dense64_f64_frag_index(mb, 0);
    const int i = 16 * mb + (lane & 15), j = 4 * ks + (lane >> 4);
    image[idx] = i > j ? m[i * 64 + j] + m[j * 64 + i] : (i == j ? m[i * 64 + i] : 0.0);
  }
}

// ---- geometry: a wavefront owns a tile of 32 chains, each chain's 64 rows split over TWO lanes (lane = chain + 32 half,
// half 0 owns rows 0..31, half 1 rows 32..63).  That halves the LDS a wavefront parks its proposals in (16 KiB) and the
// registers a lane keeps state in (64 + 64 for the prefetched tile), so that TWO wavefronts fit each SIMD: while one is in
// its arithmetic the other's rows travel.  (The first version gave a wavefront 64 chains: one wavefront per SIMD, and at
// one sweep per launch the float64 pipe (~110 us) and HBM (~100 us) only overlapped through an explicit prefetch:
// 160-170 us.)  Nothing else changes: a chain's draws, products and decisions are the same arithmetic in the same order.
constexpr int kTileChains64 = 32;

// swizzled position of (row, chain) in a wave's xp block [64 rows][32 chains]: odd rows swap the two 16-chain halves, so
// that the two rows a 32-lane access group touches (lane >> 4 = 0, 1) fall into different halves of the 64 banks
__device__ __forceinline__ int xp_at(int row, int chain) { return row * kTileChains64 + (chain ^ ((row & 1) << 4)); }

// Y = T X for the wave's 32 columns parked in `xp`; row blocks ascending or descending; after each row block
// `sink(mb, acc)` receives its two 16 x 16 result blocks (acc[nb][r]: row 16 mb + (lane >> 4) + 4 r of chain
// 16 nb + (lane & 15)).  `frags`: the fragment image, in LDS (T) or global memory (the shared factor).
// where the A-operand fragments come from: LDS (T, copied once per block) ...
struct FragsInLds {
  const double *base;
  __device__ __forceinline__ double get(int frag, int lane) const { return base[(frag << 6) + lane]; }
};
// ... or global memory through a buffer descriptor (the shared factor: LDS is full).  With plain pointer arithmetic hipcc
// hoisted the 40 loop-invariant 64-bit fragment addresses out of the tile loop (80 registers) and spilled rows of the state.
struct FragsInGlobal {
  __amdgpu_buffer_rsrc_t rsrc;
  __device__ __forceinline__ explicit FragsInGlobal(const double *image)
      : rsrc(__builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(image), 0, (unsigned int)(kDense64F64ImageDoubles * sizeof(double)),
                                               0x00020000)) {}
  __device__ __forceinline__ double get(int frag, int lane) const {
    const auto v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (unsigned int)lane * 8u, (unsigned int)frag * 512u, 0);
    return __builtin_bit_cast(double, v);
  }
};

template <bool DESCENDING, class Frags, class Sink>
__device__ __forceinline__ void wave_tri_product_64(const Frags &frags, const double *xp, int lane, Sink &&sink) {
  const int j = lane & 15, h = lane >> 4;
#pragma unroll
  for (int step = 0; step < 4; ++step) {
    const int mb = DESCENDING ? 3 - step : step;
    f64x4 acc[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) acc[nb] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < 4 * (mb + 1); ++ks) {
      const double a = frags.get(dense64_f64_frag_index(mb, ks), lane);
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const double b = xp[xp_at(4 * ks + h, 16 * nb + j)];
        acc[nb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[nb], 0, 0, 0);
      }
    }
    sink(mb, acc);
    // one row block at a time: without this fence hipcc merges the row blocks to share their B reads and keeps every
    // accumulator live
    asm volatile("" ::: "memory");
  }
}

template <int CK>
__global__ void __launch_bounds__(kDense64F64Threads, 2)
    k_step_dense64_f64(StepArgs<double> a, const double *__restrict__ t_image, const double *__restrict__ l_image) {
  constexpr int kRngUnroll = CK == CK_SHARED ? 1 : ME_DENSE64_F64_RNG_UNROLL;
  constexpr int D = 64, H = 32;          // H rows per lane
  using N_ = Num<double>;
  extern __shared__ __attribute__((aligned(16))) double smem64[];
  double *lds_t = smem64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double *xp = smem64 + kDense64F64ImageDoubles + wave * kDense64F64WaveDoubles;
  double *exch = xp + D * kTileChains64;
  {   // the image in one batch of loads (a rolled copy loop waits for every load before it issues the next)
    constexpr int kPieces = kDense64F64ImageDoubles / kDense64F64Threads;
    static_assert(kPieces * kDense64F64Threads == kDense64F64ImageDoubles, "the image divides evenly over the workgroup");
    double piece[kPieces];
#pragma unroll
    for (int k = 0; k < kPieces; ++k) piece[k] = t_image[k * kDense64F64Threads + threadIdx.x];
#pragma unroll
    for (int k = 0; k < kPieces; ++k) lds_t[k * kDense64F64Threads + threadIdx.x] = piece[k];
  }
  N_::prepare();    // the log table of the float64 Box-Muller; ends with the block barrier that also covers lds_t

  const int j = lane & 15, h = lane >> 4;                 // position in the MFMA result blocks
  const int cl = lane & 31, half = lane >> 5;             // chain of the tile, and which 32 rows of it this lane owns
  unsigned int wave_accepted = 0;
  bool bad_energy = false, bad_width = false;
  // the state is tile-major (me_device.h: XField): a wavefront's 32 chains x 64 rows are half of one contiguous 32 KiB block
  const TiledField<double> fx(a.x, a.n, D);
  const Field<double> fe(a.energy, a.n, 1), fw(a.width, a.n, 1);
  const long long n_tiles = (a.n + kTileChains64 - 1) / kTileChains64;
  const long long tile_stride = (long long)gridDim.x * (kDense64F64Threads / 64);

  // every lane stays active (the MFMAs need the whole wavefront): tail lanes shadow the last chain
  auto chain_of = [&](long long tile) {
    const long long c_raw = tile * kTileChains64 + cl;
    return c_raw < a.n ? c_raw : a.n - 1;
  };
  // lane offset into the state field: the chain's place in its tile, plus 32 rows for the upper half (the row index of
  // load / store is a wave-uniform immediate)
  auto state_off = [&](long long c) { return tiled_offset<double>(c, D) + (unsigned int)half * (unsigned int)H * TiledField<double>::kEntryBytes; };
  long long tile = (long long)blockIdx.x * (kDense64F64Threads / 64) + wave;
  double x[H], e = 0.0, w = 0.0;
  if (tile < n_tiles) {
    const long long c = chain_of(tile);
    const unsigned int xoff = state_off(c);
#pragma unroll
    for (int i = 0; i < H; ++i) x[i] = (ME_DENSE64_F64_NT & 1) ? fx.load_nt(i, xoff) : fx.load(i, xoff);
    e = fe.load(0, (unsigned int)c * 8u);
    w = fw.load(0, (unsigned int)c * 8u);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);        // enter the loop with nothing pending (see the end of the tile loop)
#if ME_DENSE64_F64_PRIO
  // the two wavefronts of a SIMD (wave, wave + 4) get different issue priorities: one runs ahead of the other instead of
  // both finishing their tiles at the same moment
  if (wave < 4) __builtin_amdgcn_s_setprio(2);
#endif
  if constexpr (ME_DENSE64_F64_STAGGER > 0) {
    // the two wavefronts of a SIMD run the same program and start together: delay the second half of the workgroup so
    // that one wavefront's memory phases fall into the other's arithmetic (guide: two waves per SIMD, item 9)
    if (wave >= 4) {
#pragma unroll 1
      for (int k = 0; k < ME_DENSE64_F64_STAGGER; ++k) __builtin_amdgcn_s_sleep(127);
    }
  }
  while (tile < n_tiles) {
    const long long c = chain_of(tile);
    const bool live = tile * kTileChains64 + cl < a.n;
    const unsigned int coff = (unsigned int)c * 8u;
    const unsigned long long gid = a.chain_offset + (unsigned long long)c;
    // the next tile's rows travel while this one computes (second register set)
    const long long next = tile + tile_stride;
    const bool have_next = next < n_tiles;
    double xn[H], en = 0.0, wn = 0.0;
    // (tile-major rows are compile-time immediates of the buffer instructions: nothing loop-invariant to keep in registers)
    const TiledField<double> &fxt = fx;
    auto prefetch = [&]() {
      if (have_next) {       // wave-uniform
        const long long cn = chain_of(next);
        const unsigned int noff = state_off(cn);
#pragma unroll
        for (int i = 0; i < H; ++i) xn[i] = (ME_DENSE64_F64_NT & 1) ? fxt.load_nt(i, noff) : fxt.load(i, noff);
        en = fe.load(0, (unsigned int)cn * 8u);
        wn = fw.load(0, (unsigned int)cn * 8u);
      }
    };

    if constexpr (ME_DENSE64_F64_PREFETCH_EARLY && CK != CK_SHARED) prefetch();
#ifdef ME_DENSE64_F64_MEMORY_ONLY      // experiment: the kernel's memory traffic without its arithmetic
    prefetch();
    for (int s = 0; s < 0; ++s) {
#else
    for (int s = 0; s < a.n_sweeps; ++s) {
#endif
      const unsigned long long step = a.step_index + (unsigned long long)s;
      // the fragment reads of T are loop-invariant (80 registers if hoisted out of the sweep loop): keep them here
      asm volatile("" ::: "memory");
      // ---- this lane's 32 normals (Philox blocks 8 half .. 8 half + 7, two Box-Muller pairs each) -> LDS; a rolled loop
#pragma unroll kRngUnroll
      for (int k = 0; k < 8; ++k) {
        const int b = 8 * half + k;
        U4 ctr;
        ctr.x = (uint32_t)gid;
        ctr.y = (uint32_t)(gid >> 32);
        ctr.z = (uint32_t)step;
        ctr.w = ((uint32_t)(step >> 32) << 16) | (uint32_t)b;
        const U4 o = philox4x32_10(ctr, a.seed_lo, a.seed_hi);
        double g[4];
        N_::normal_pair(o.x, o.y, g[0], g[1]);
        N_::normal_pair(o.z, o.w, g[2], g[3]);
#pragma unroll
        for (int i = 0; i < 4; ++i) xp[xp_at(4 * b + i, cl)] = g[i];
      }
      __builtin_amdgcn_wave_barrier();
      if constexpr (CK == CK_SHARED) {
        // y = L g, row blocks descending, written over g in place
        wave_tri_product_64<true>(FragsInGlobal(l_image), xp, lane, [&](int mb, const f64x4 (&acc)[2]) {
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) xp[xp_at(16 * mb + h + 4 * r, 16 * nb + j)] = acc[nb][r];
          __builtin_amdgcn_wave_barrier();
        });
      }
      // The prefetch goes out here, behind the first sweep's draws: a wave can have at most 63 vector-memory operations
      // outstanding (vmcnt is 6 bits), so queued directly behind the previous tile's stores it would stall the wave at
      // issue; and with a shared factor it has to stay behind that sweep's L g, whose fragment loads from global memory
      // would otherwise wait for it (loads return in order).
      if (s == 0 && !(ME_DENSE64_F64_PREFETCH_EARLY && CK != CK_SHARED)) prefetch();
      // x' = x + w g (or x + w L g) by the owners, in place
#pragma unroll
      for (int i0 = 0; i0 < H; i0 += 8) {       // eight at a time: 32 reads in flight would be 64 more live registers
#pragma unroll
        for (int i = i0; i < i0 + 8; ++i) xp[xp_at(H * half + i, cl)] = x[i] + w * xp[xp_at(H * half + i, cl)];
        asm volatile("" ::: "memory");
      }
      __builtin_amdgcn_wave_barrier();
      // ---- E' = x'^T (T x'): partial dot products where the results are, then the exchange over the four row groups
      double part[2] = {0.0, 0.0};
      wave_tri_product_64<false>(FragsInLds{lds_t}, xp, lane, [&](int mb, const f64x4 (&acc)[2]) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int r = 0; r < 4; ++r) part[nb] = __builtin_fma(xp[xp_at(16 * mb + h + 4 * r, 16 * nb + j)], acc[nb][r], part[nb]);
      });
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) exch[(nb * 4 + h) * 16 + j] = part[nb];
      __builtin_amdgcn_wave_barrier();
      double e_new = 0.0;      // both lanes of a chain sum the same four values in the same order
#pragma unroll
      for (int hh = 0; hh < 4; ++hh) e_new += exch[((cl >> 4) * 4 + hh) * 16 + (cl & 15)];
      // ---- accept uniform: word 64 = block 16, output 0
      U4 ctr;
      ctr.x = (uint32_t)gid;
      ctr.y = (uint32_t)(gid >> 32);
      ctr.z = (uint32_t)step;
      ctr.w = ((uint32_t)(step >> 32) << 16) | 16u;
      const double u = N_::unit(philox4x32_10(ctr, a.seed_lo, a.seed_hi).x);
      bool rejected = false;
      if (a.reject_kind == ME_REJECT_ABS_REAL0_GE) rejected = !(N_::abs_(xp[xp_at(0, cl)]) < a.reject_bound);
      const double diff = e_new - e;
      bool accept = diff <= 0.0;
      if (a.temp > 0.0) accept = accept || N_::uphill(u, diff, a.inv_temp, a.inv_temp_log2e);
      accept = accept && !rejected;
      bad_energy |= (live && !rejected && !N_::finite(e_new));
      if (accept) {
#pragma unroll
        for (int i = 0; i < H; ++i) x[i] = xp[xp_at(H * half + i, cl)];
      }
      e = accept ? e_new : e;
      w = N_::adapt(w, accept, a.ratio, a.p, a.damping, a.up, a.down);
      wave_accepted += (unsigned int)__popcll(__ballot(accept && live && half == 0));
      __builtin_amdgcn_wave_barrier();      // the next sweep overwrites xp
    }
    bad_width |= live && !(w > 0.0);
    // vmcnt is a 6-bit counter and memory operations retire in order: once this tile's stores and the next prefetch's
    // loads are queued behind them, ANY wait on the rows prefetched above can only be expressed as "drain (almost)
    // everything".  Wait for them here instead -- they were issued most of a tile of arithmetic ago.
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0), expcnt / lgkmcnt untouched
    if (live) {
      const unsigned int xoff = state_off(c);
#pragma unroll
      for (int i = 0; i < H; ++i) {
        if constexpr ((ME_DENSE64_F64_NT & 2) != 0) fxt.store_nt(i, xoff, x[i]);
        else fxt.store(i, xoff, x[i]);
      }
      if (half == 0) {
        fe.store(0, coff, e);
        fw.store(0, coff, w);
      }
    }
#pragma unroll
    for (int i = 0; i < H; ++i) x[i] = xn[i];
    e = en;
    w = wn;
    tile = next;
  }
  if (lane == 0 && wave_accepted) {
    unsigned long long *slot = a.accept_slots + (size_t)blockIdx.x * (kDense64F64Threads / 64) + wave;
    *slot += (unsigned long long)wave_accepted;
  }
  const unsigned int bits = (bad_energy ? ST_NONFINITE_ENERGY : 0u) | (bad_width ? ST_BAD_WIDTH : 0u);
  if (bits) atomicOr(a.status, bits);
}

// Host launcher: one persistent 512-thread workgroup per CU (8 wavefronts, two per SIMD; LDS admits no more).  Both
// per-device properties -- the raised dynamic-LDS limit and the CU count -- are resolved per device of the process.
template <int CK>
inline hipError_t launch_step_dense64_f64(const StepArgs<double> &a, const double *t_image, const double *l_image,
                                          int grid_blocks, hipStream_t stream) {
  static PerDevice<hipError_t> attr_cache;
  static PerDevice<int> cu_cache;
  int device = 0;
  hipError_t rc = hipGetDevice(&device);
  if (rc != hipSuccess) return rc;
  rc = attr_cache.get(device, [] {
    return hipFuncSetAttribute((const void *)k_step_dense64_f64<CK>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)kDense64F64LdsBytes);
  });
  if (rc != hipSuccess) return rc;
  const int cus = cu_cache.get(device, [device] {
    int count = 0;
    if (hipDeviceGetAttribute(&count, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || count <= 0) return 256;
    return count;
  });
  constexpr int chains_per_block = kTileChains64 * (kDense64F64Threads / 64);
  long long blocks = (a.n + chains_per_block - 1) / chains_per_block;
  const long long cap = grid_blocks > 0 ? grid_blocks : (long long)cus;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL((k_step_dense64_f64<CK>), dim3((unsigned)blocks), dim3(kDense64F64Threads), kDense64F64LdsBytes, stream,
                     a, t_image, l_image);
  return hipGetLastError();
}

}  // namespace me
