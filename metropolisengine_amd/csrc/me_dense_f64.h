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
// compiled to 32-49 KB of scratch per lane (16 ms per launch).  Here 80 instructions per 32 chains and 8 accumulator
// registers do the product, the coefficients are 40 LDS words per lane, and nothing spills.
//
//   * Only the symmetric part of A matters to x^T A x.  T = tril(A + A^T, -1) + diag(A) is LOWER TRIANGULAR and
//     E = x'^T (T x'), so row block mb (16 rows) of the product needs k < 16 (mb + 1) only: 4+8+12+16 = 40 k-steps of
//     4 instead of 64, times 4 chain blocks = 160 MFMAs instead of 256.  The shared factor L is lower triangular by
//     construction and goes through the same code (fold(L) = L).
//   * One wavefront = one tile of 32 chains, state AND proposals in registers in the matrix instruction's own operand
//     layout (see "geometry" below): the products read their B operands from the registers the proposals were computed
//     in and deliver their results to the lanes that own those rows.  LDS holds the fragments of T, 1.25 KiB of exchange
//     per wavefront and the slab the NEXT tile's rows are prefetched into.
//   * A-operand fragments of T: image[(mb, ks)][lane] = T[16 mb + (l & 15)][4 ks + (l >> 4)], built once per engine
//     (k_dense64_f64_fragments), 20 KiB, copied to LDS once per block.  The factor image (CK_SHARED) stays in global
//     memory (L1/L2 resident, read through a buffer descriptor).
//   * Results: lane l holds rows (l >> 4) + 4 r (r < 4) of chain column (l & 15) of a 16 x 16 block (the float64 C/D
//     map differs from every other MFMA's, checked in the probe).  For L g the row blocks are produced in DESCENDING order,
//     so that block mb of the result can overwrite slots 4 mb .. 4 mb + 3 of g in place (later blocks read only lower rows).
//   * Two wavefronts per SIMD (<= 256 registers: 64 + 64 of state and proposals).
//
// History and measurements (2^19 chains, one sweep per launch / per fused sweep, tools/dev/time_dense64_ramp.py):
//   64 chains per wavefront, one wavefront per SIMD, proposals in LDS                       162 / 110 us
//   32 chains per wavefront, a chain's rows over two lanes, proposals in LDS (16 KiB per
//   wavefront), next tile prefetched into a second register set                             135 /  94 us
//   ... tile-major state                                                                    128 /  97 us
//   this version (registers only, 16-byte accesses, prefetch by buffer_load ... lds)        125 /  92 us
// What bounds the one-sweep launch (profiles/r03_dense64_f64_register_tiles.txt): with loads and stores compiled out it
// takes 103 us (the fused sweep's 91 us + 10 % per-tile work + launch); the state loads alone add 8 us, the stores alone
// 9 us, both 22-25 us -- whether the next tile is prefetched (registers or LDS-DMA) or loaded when the tile is done, the
// stores issued at the tile's end, deferred and spread over the next tile's draws, or paced two per Philox block; whether
// the wavefronts of a workgroup start together or staggered; with any cache policy of the stores; from HBM or from a
// 32 KiB region that never leaves the L2.  The SQ counters show the difference as issue stalls (SQ_WAIT_INST_ANY), not as
// waits on memory; the shader clock stays at 2.33-2.40 GHz (1 360 W with the traffic, 1 000 W without).  The memory traffic
// of the kernel on its own (-DME_DENSE64_F64_MEMORY_ONLY) takes 72 us.
#pragma once

#include "me_device.h"
#include "me_per_device.h"

#include <type_traits>

namespace me {

using f64x4 = __attribute__((ext_vector_type(4))) double;

#ifndef ME_DENSE64_F64_PREFETCH
#define ME_DENSE64_F64_PREFETCH 2       // the next tile's rows: 0 = loaded when this tile is done, 1 = into a second register set, 2 = into LDS by the load unit
#endif
#ifndef ME_DENSE64_F64_STORE_AUX
#define ME_DENSE64_F64_STORE_AUX 0      // cache policy of the state stores: bit 0 = sc0, bit 1 = nt, bit 4 = sc1
#endif
#ifndef ME_DENSE64_F64_STAGGER
#define ME_DENSE64_F64_STAGGER 0        // dev: wavefront w of a workgroup starts w x this many x 64 cycles late
#endif
#ifndef ME_DENSE64_F64_EXPERIMENT
#define ME_DENSE64_F64_EXPERIMENT 0     // dev: bit 0 = no state stores, bit 1 = no state loads, bit 3 = every tile uses the first 32 KiB of the state
#endif
using f64x2 = __attribute__((ext_vector_type(2))) double;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
constexpr int kDense64F64Threads = 512;                  // 8 waves, two per SIMD
constexpr int kDense64F64Frags = 40;                     // (mb, ks) pairs with 4 ks < 16 (mb + 1)
constexpr int kDense64F64ImageDoubles = kDense64F64Frags * 64;
// per wavefront: the 2 x 4 x 16 energy exchange, 2 x 16 wall values, and the slab the next tile's rows are prefetched into
constexpr int kDense64F64WaveDoubles = 160 + (ME_DENSE64_F64_PREFETCH == 2 ? 64 * 32 : 0);
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

// ---- geometry.  A wavefront owns a tile of 32 chains and keeps state AND proposals in registers, in the matrix
// instruction's own operand layout: lane (h, j) = (lane >> 4, lane & 15) owns rows {4 ks + h : ks < 16} of the two chains
// 2 j and 2 j + 1 of the tile ("slots" ks, chain blocks nb = 0, 1).  That is at once
//   * the B-operand layout of v_mfma_f64_16x16x4_f64 (k-step ks wants row 4 ks + h of column j from lane (h, j)): the
//     proposals feed the matrix cores straight from the registers they were computed in;
//   * its C/D layout (lane (h, j) receives rows 16 mb + h + 4 r of column j): row block mb of T x' (or of L g) arrives in the
//     lanes that own those rows, as slots 4 mb + r -- the energy's partial dot products and the in-place L g need no
//     data movement at all;
//   * a memory access of 16 bytes per lane (both chains of a row are neighbours in the tile-major state): 16 loads and 16
//     stores per tile, each covering four 256-byte row segments.
// What does need to move are the normals: Philox block b of a chain yields the normals of rows 4 b .. 4 b + 3, one for
// each of the chain's four lanes.  Lane h draws blocks 4 q + h (q < 4) of both chains and a 4 x 4 transposition across the
// wavefront's four 16-lane rows (v_permlane32_swap + v_permlane16_swap, 8 instructions per four doubles) hands every
// normal to its owner -- no lane draws anything twice.  The accept decision of chain nb is taken by the lanes with
// h >> 1 == nb (twice each, as before); its four partial energies meet in 1 KiB of LDS.
// The version before this one parked the proposals in LDS ([row][chain], 16 KiB per wavefront) between the draws, the
// products and the decision: 240 LDS instructions per tile and sweep that are gone now, and the LDS they occupied takes
// the NEXT tile's rows instead, written by the load unit itself (buffer_load ... lds) while this tile computes: no second
// register set.
constexpr int kTileChains64 = 32;
constexpr int kDense64F64Slots = 16;                     // rows per lane and chain

// a[lanes 32-63] <-> b[lanes 0-31]
__device__ __forceinline__ void swap_rows_by_32(double &a, double &b) {
  const unsigned long long ua = __builtin_bit_cast(unsigned long long, a), ub = __builtin_bit_cast(unsigned long long, b);
  // (the builtin's two results go through scalar locals: me_dense_mfma.h swap32)
  const auto lo = __builtin_amdgcn_permlane32_swap((unsigned int)ua, (unsigned int)ub, false, false);
  const unsigned int lo0 = lo[0], lo1 = lo[1];
  const auto hi = __builtin_amdgcn_permlane32_swap((unsigned int)(ua >> 32), (unsigned int)(ub >> 32), false, false);
  const unsigned int hi0 = hi[0], hi1 = hi[1];
  a = __builtin_bit_cast(double, ((unsigned long long)hi0 << 32) | lo0);
  b = __builtin_bit_cast(double, ((unsigned long long)hi1 << 32) | lo1);
}
// a[odd 16-lane rows] <-> b[even 16-lane rows]
__device__ __forceinline__ void swap_rows_by_16(double &a, double &b) {
  const unsigned long long ua = __builtin_bit_cast(unsigned long long, a), ub = __builtin_bit_cast(unsigned long long, b);
  const auto lo = __builtin_amdgcn_permlane16_swap((unsigned int)ua, (unsigned int)ub, false, false);
  const unsigned int lo0 = lo[0], lo1 = lo[1];
  const auto hi = __builtin_amdgcn_permlane16_swap((unsigned int)(ua >> 32), (unsigned int)(ub >> 32), false, false);
  const unsigned int hi0 = hi[0], hi1 = hi[1];
  a = __builtin_bit_cast(double, ((unsigned long long)hi0 << 32) | lo0);
  b = __builtin_bit_cast(double, ((unsigned long long)hi1 << 32) | lo1);
}
// d[w] of 16-lane row h  <->  d[h] of row w
__device__ __forceinline__ void transpose_rows_4x4(double (&d)[4]) {
  swap_rows_by_32(d[0], d[2]);
  swap_rows_by_32(d[1], d[3]);
  swap_rows_by_16(d[0], d[1]);
  swap_rows_by_16(d[2], d[3]);
}
// v as held by lanes 0-31 (`low`) and by lanes 32-63 (`high`), in every lane
__device__ __forceinline__ void both_halves(double v, double &low, double &high) {
  low = v;
  high = v;
  swap_rows_by_32(low, high);
}

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

// Y = T X for the wave's 32 columns held in `v` (slot ks, chain block nb = v[ks][nb]); row blocks ascending or descending;
// after each row block `sink(mb, acc)` receives its two 16 x 16 result blocks, acc[nb][r] = row 16 mb + h + 4 r = slot
// 4 mb + r of this lane's chain of block nb.
template <bool DESCENDING, class Frags, class Sink>
__device__ __forceinline__ void wave_tri_product_64(const Frags &frags, const f64x2 (&v)[kDense64F64Slots], int lane, Sink &&sink) {
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
      for (int nb = 0; nb < 2; ++nb) acc[nb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, v[ks][nb], acc[nb], 0, 0, 0);
    }
    sink(mb, acc);
    // one row block at a time: without this fence hipcc merges the row blocks and keeps every accumulator live
    asm volatile("" ::: "memory");
  }
}

template <int CK>
__global__ void __launch_bounds__(kDense64F64Threads, 2)
    k_step_dense64_f64(StepArgs<double> a, const double *__restrict__ t_image, const double *__restrict__ l_image) {
  constexpr int D = 64, S = kDense64F64Slots;
  constexpr unsigned int kSlotBytes = 4u * TiledField<double>::kEntryBytes;      // slot ks -> row 4 ks + h: 2 KiB apart
  using N_ = Num<double>;
  extern __shared__ __attribute__((aligned(16))) double smem64[];
  double *lds_t = smem64;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  double *mine = smem64 + kDense64F64ImageDoubles + wave * kDense64F64WaveDoubles;
  double *exch = mine;                       // [chain block][h][j]: the four partial energies of a chain
  double *wall = mine + 128;                 // [chain block][j]: row 0 of the proposals, for the hard wall
  [[maybe_unused]] double *slab = mine + 160;   // [slot][lane][chain block]: the next tile's rows
  const int j = lane & 15, h = lane >> 4;
  const int own = h >> 1;                    // the chain block whose accept decision, energy and width this lane carries
  unsigned int wave_accepted = 0;
  bool bad_energy = false, bad_width = false;
  // the state is tile-major (me_device.h: XField), padded to whole 64-chain tiles and initialised there: a ragged last tile
  // computes on the padding (energy and width of chains beyond n read as 0 through the range-checked descriptors, their
  // stores are dropped)
  const TiledField<double> fx(a.x, a.n, D);
  const Field<double> fe(a.energy, a.n, 1), fw(a.width, a.n, 1);
  const long long n_tiles = (a.n + kTileChains64 - 1) / kTileChains64;
  const long long tile_stride = (long long)gridDim.x * (kDense64F64Threads / 64);
  // byte offset of (row h, chain 2 j) of 32-chain tile t: t >> 1 is the 64-chain tile of the layout, t & 1 its half
  auto state_off = [&](long long t) {
    if constexpr ((ME_DENSE64_F64_EXPERIMENT & 8) != 0) t &= 1;      // dev: every tile reads and writes the first 32 KiB
    return (unsigned int)(((t >> 1) * (long long)(D * 64) + (t & 1) * 32 + 2 * j) * 8) + (unsigned int)h * TiledField<double>::kEntryBytes;
  };
  auto load_rows = [&](long long t, f64x2 (&dst)[S]) {
    const unsigned int off = state_off(t);
#pragma unroll
    for (int ks = 0; ks < S; ++ks) {
      if constexpr ((ME_DENSE64_F64_EXPERIMENT & 2) != 0) dst[ks] = f64x2{(double)off, 0.0};
      else dst[ks] = __builtin_bit_cast(f64x2, __builtin_amdgcn_raw_buffer_load_b128(fx.rsrc, off, (unsigned int)ks * kSlotBytes, 0));
    }
  };

  long long tile = (long long)blockIdx.x * (kDense64F64Threads / 64) + wave;
  f64x2 x[S];
  double e = 0.0, w = 0.0;
  if (tile < n_tiles) {
    load_rows(tile, x);
    const unsigned int coff = (unsigned int)(tile * kTileChains64 + 2 * j + own) * 8u;
    e = fe.load(0, coff);
    w = fw.load(0, coff);
  }
  // the workgroup's prologue runs BEHIND the first tile's loads (the loads of ~2 000 wavefronts go out at the same moment
  // and take a few microseconds to come back)
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
  __builtin_amdgcn_s_waitcnt(0x0F70);        // enter the loop with nothing pending
#if ME_DENSE64_F64_STAGGER > 0
  // every wavefront of the workgroup starts a little later than the one before it
#pragma unroll 1
  for (int k = 0; k < wave; ++k) __builtin_amdgcn_s_sleep(ME_DENSE64_F64_STAGGER);
#endif
  while (tile < n_tiles) {
    const long long c_own = tile * kTileChains64 + 2 * j + own;
    const bool live = c_own < a.n;
    const unsigned int coff = (unsigned int)c_own * 8u;
    const unsigned long long gid0 = a.chain_offset + (unsigned long long)(tile * kTileChains64 + 2 * j);
    const long long next = tile + tile_stride;
    const bool have_next = next < n_tiles;        // wave-uniform
    double en = 0.0, wn = 0.0;
#if ME_DENSE64_F64_PREFETCH == 1
    f64x2 xn[S];
#endif
    // the next tile's rows travel while this one computes
    auto prefetch = [&]() {
      if (have_next) {
        const unsigned int ncoff = (unsigned int)(next * kTileChains64 + 2 * j + own) * 8u;
        en = fe.load(0, ncoff);
        wn = fw.load(0, ncoff);
#if ME_DENSE64_F64_PREFETCH == 1
        load_rows(next, xn);
#elif ME_DENSE64_F64_PREFETCH == 2
        const unsigned int off = state_off(next);
#pragma unroll
        for (int ks = 0; ks < S; ++ks)     // 64 lanes x 16 bytes land at slab[ks][lane]
          __builtin_amdgcn_raw_ptr_buffer_load_lds(fx.rsrc, (__attribute__((address_space(3))) void *)(slab + ks * 128), 16, off,
                                                   (unsigned int)ks * kSlotBytes, 0, 0);
#endif
      }
    };
    if constexpr (CK != CK_SHARED) prefetch();
    // one sweep; FIRST: the first of the tile, which carries the previous tile's stores (and, with a shared factor, the prefetch)
    auto sweep = [&](auto first_tag, int s) {
      constexpr bool FIRST = decltype(first_tag)::value;
      const unsigned long long step = a.step_index + (unsigned long long)s;
      // the fragment reads of T are loop-invariant (80 registers if hoisted out of the sweep loop): keep them here
      asm volatile("" ::: "memory");
      // ---- the normals: this lane draws Philox blocks 4 q + h of both chains (two Box-Muller pairs each); the
      // transposition hands word w of block 4 q + h' to lane row w as slot 4 q + h'
      f64x2 xp[S];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          const unsigned long long gid = gid0 + (unsigned long long)nb;
          U4 ctr;
          ctr.x = (uint32_t)gid;
          ctr.y = (uint32_t)(gid >> 32);
          ctr.z = (uint32_t)step;
          ctr.w = ((uint32_t)(step >> 32) << 16) | (uint32_t)(4 * q + h);
          const U4 o = philox4x32_10(ctr, a.seed_lo, a.seed_hi);
          double g[4];
          N_::normal_pair(o.x, o.y, g[0], g[1]);
          N_::normal_pair(o.z, o.w, g[2], g[3]);
          transpose_rows_4x4(g);
#pragma unroll
          for (int k = 0; k < 4; ++k) xp[4 * q + k][nb] = g[k];
          __builtin_amdgcn_sched_barrier(0);      // one block at a time: interleaved, the blocks' temporaries spill the state
        }
      }
      if constexpr (CK == CK_SHARED) {
        // y = L g, row blocks descending: block mb overwrites slots 4 mb .. 4 mb + 3, which the later blocks do not read
        wave_tri_product_64<true>(FragsInGlobal(l_image), xp, lane, [&](int mb, const f64x4 (&acc)[2]) {
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) xp[4 * mb + r][nb] = acc[nb][r];
        });
        // with a shared factor the prefetch stays behind the first sweep's L g, whose fragment loads from global memory
        // would otherwise queue up behind it (loads return in order)
        if constexpr (FIRST) prefetch();
      }
      // x' = x + w g (or x + w L g)
      double wv[2];
      both_halves(w, wv[0], wv[1]);
#pragma unroll
      for (int ks = 0; ks < S; ++ks)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) xp[ks][nb] = __builtin_fma(wv[nb], xp[ks][nb], x[ks][nb]);
      // ---- E' = x'^T (T x'): partial dot products where the results are, then the exchange over the four lane rows
      double part[2] = {0.0, 0.0};
      wave_tri_product_64<false>(FragsInLds{lds_t}, xp, lane, [&](int mb, const f64x4 (&acc)[2]) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int r = 0; r < 4; ++r) part[nb] = __builtin_fma(xp[4 * mb + r][nb], acc[nb][r], part[nb]);
      });
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) exch[(nb * 4 + h) * 16 + j] = part[nb];
      const bool walled = a.reject_kind == ME_REJECT_ABS_REAL0_GE;      // wave-uniform
      if (walled && h == 0) {
        wall[j] = xp[0][0];
        wall[16 + j] = xp[0][1];
      }
      __builtin_amdgcn_wave_barrier();
      double e_new = 0.0;      // the four lanes of a chain block sum the same four values in the same order
#pragma unroll
      for (int hh = 0; hh < 4; ++hh) e_new += exch[(own * 4 + hh) * 16 + j];
      bool rejected = false;
      if (walled) rejected = !(N_::abs_(wall[own * 16 + j]) < a.reject_bound);
      // ---- accept uniform: word 64 = block 16, output 0
      const unsigned long long gid = gid0 + (unsigned long long)own;
      U4 ctr;
      ctr.x = (uint32_t)gid;
      ctr.y = (uint32_t)(gid >> 32);
      ctr.z = (uint32_t)step;
      ctr.w = ((uint32_t)(step >> 32) << 16) | 16u;
      const double u = N_::unit(philox4x32_10(ctr, a.seed_lo, a.seed_hi).x);
      const double diff = e_new - e;
      bool accept = diff <= 0.0;
      if (a.temp > 0.0) accept = accept || N_::uphill(u, diff, a.inv_temp, a.inv_temp_log2e);
      accept = accept && !rejected;
      bad_energy |= (live && !rejected && !N_::finite(e_new));
      // lanes 0-15 speak for chain block 0, lanes 32-47 for chain block 1
      const unsigned long long votes = __ballot(accept);
      const bool take0 = (((unsigned int)votes >> j) & 1u) != 0u, take1 = (((unsigned int)(votes >> 32) >> j) & 1u) != 0u;
#pragma unroll
      for (int ks = 0; ks < S; ++ks) {
        x[ks][0] = take0 ? xp[ks][0] : x[ks][0];
        x[ks][1] = take1 ? xp[ks][1] : x[ks][1];
      }
      e = accept ? e_new : e;
      w = N_::adapt(w, accept, a.ratio, a.p, a.damping, a.up, a.down);
      wave_accepted += (unsigned int)__popcll(__ballot(accept && live && (h & 1) == 0));
      __builtin_amdgcn_wave_barrier();      // the next sweep overwrites the exchange
    };
#ifndef ME_DENSE64_F64_MEMORY_ONLY     // (defined: the kernel's memory traffic without its arithmetic)
    sweep(std::true_type{}, 0);
    for (int s = 1; s < a.n_sweeps; ++s) sweep(std::false_type{}, s);
#endif
    bad_width |= live && !(w > 0.0);
    // The prefetch was issued a tile of arithmetic ago: wait for it BEFORE this tile's stores are queued behind it (vmcnt
    // counts in order; afterwards any wait on the prefetch would drain the stores as well).
    // (an asm with a memory clobber, not the builtin: the compiler must not move the reads of the slab above this wait -- it
    // does not know that the load unit writes LDS behind its back)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    {
      const unsigned int off = state_off(tile);
#pragma unroll
      for (int ks = 0; ks < S; ++ks)
        if ((ME_DENSE64_F64_EXPERIMENT & 1) == 0 || x[ks][0] == 1.2345)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, x[ks]), fx.rsrc, off, (unsigned int)ks * kSlotBytes, ME_DENSE64_F64_STORE_AUX);
      if (live && (h & 1) == 0) {
        fe.store(0, coff, e);
        fw.store(0, coff, w);
      }
    }
    if (have_next) {
#if ME_DENSE64_F64_PREFETCH == 1
#pragma unroll
      for (int ks = 0; ks < S; ++ks) x[ks] = xn[ks];
#elif ME_DENSE64_F64_PREFETCH == 2
#pragma unroll
      for (int ks = 0; ks < S; ++ks) x[ks] = *reinterpret_cast<const f64x2 *>(slab + ks * 128 + lane * 2);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // the slab has been read: free for the next prefetch
#else
      load_rows(next, x);
      __builtin_amdgcn_s_waitcnt(0x0F70);
#endif
    }
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
