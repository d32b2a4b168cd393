// gemm_ws.h -- "weights-stationary" fp32 MFMA GEMM engine for gfx950.
//
// Every dense contraction on the DPTN path is   out[M, NOUT] = f(A[M, KIN]) * W[NOUT, KIN]^T   with a
// HUGE M (B*S*K ~ 3.4e5 tokens) and TINY weights (<= 1024 x 256 fp32).  So the tiling is turned
// inside out relative to a square GEMM: a workgroup loads its slice of W ONCE into registers as
// ready-made MFMA B-fragments (the 512 KiB register file of a CU is the biggest on-chip memory)
// and then streams 32-row token tiles through LDS, persistent-style.
//
//   block = 256 threads = WR x WC waves; wave (wr, wc) owns rows [32wr, 32wr+32) of the BM = 32*WR row
//   tile and columns col0 + (wc*NT + nt)*32 + c, nt < NT.
//
// Template hooks
//   ALoad::load4(tile, row, k4)      -> float4 of A for tile row `row` (0..BM-1), floats [4*k4, 4*k4+4)
//                                       (gather / prologue ops such as PReLU live here)
//   Epi::DIRECT == true : Epi::store_acc(tile, wr, cb, acc, c, hh)   register epilogue (LSTM pre-acts)
//   Epi::DIRECT == false: Epi::row(tile, row, c4, float4 v)          row-space epilogue: the C tile goes
//                          through LDS, a row of the tile is handled by GROUP = WGCOLS/4 adjacent lanes
//                          (so LayerNorm statistics are a sub-wave shuffle reduction).
#pragma once
#include <type_traits>
#include <utility>

#include "common.h"
#include "lstm.h"
#include "lstm16.h"

// phase stamps for tools/microbench/gemm_phases.hip (a diagnostic build defines these; the product build does not)
#ifndef GEMM_STAMP
#define GEMM_STAMP_ENTRY
#define GEMM_STAMP_DECL
#define GEMM_STAMP(i)
#define GEMM_STAMP_ACC(i, x)
#define GEMM_STAMP_END
#endif

// Epilogues may declare per-column constants (bias, LayerNorm gamma / beta ...) that the engine loads ONCE per
// workgroup, before the tile loop:   struct Cols {...};  Cols cols(colgroup, c4) const;   row(..., const Cols&).
// (Loaded inside row() they cost two dependent L1 round trips per pass and tile -- stamps: 3.5 k of the 13.3 k cycles a
// wave spent per tile of the out-projection GEMM.)
template <class E, class = void>
struct epi_has_cols : std::false_type {};
template <class E>
struct epi_has_cols<E, std::void_t<typename E::Cols>> : std::true_type {};
struct EpiNoCols {};
// Row-space epilogues may request a SECOND per-row operand ahead of the MFMA block (`prefetch2`, clamped / branch-free
// like prefetchc): a row operand loaded inside row() costs one exposed global round trip per pass.
template <class E, class = void>
struct has_prefetch2 : std::false_type {};
template <class E>
struct has_prefetch2<E, std::void_t<decltype(std::declval<const E&>().prefetch2(0, 0, 0))>> : std::true_type {};
// DIRECT epilogues may offer `float direct_const(colgroup, cb, c)` (e.g. the scaled bias of the lane's column): loaded
// once per workgroup and handed to store_acc instead of two dependent global loads per column block and tile.
// Epi::FOLD (DIRECT epilogues of the form acc * col_scale + direct_const): the engine folds the scale into the weight
// fragments when it loads them and the constant into the accumulators' start (one extra MFMA: ones x constant), so the
// epilogue is 16-byte stores straight from the accumulator registers -- no per-element fma, no AGPR -> VGPR moves
// (64 + 64 instructions per K4 tile beside fp32 MFMAs, where every vector instruction costs its issue time).
template <class E, class = void>
struct epi_folds : std::false_type {};
template <class E>
struct epi_folds<E, std::enable_if_t<E::FOLD>> : std::true_type {};
template <class E, class = void>
struct has_direct_const : std::false_type {};
template <class E>
struct has_direct_const<E, std::void_t<decltype(std::declval<const E&>().direct_const(0, 0, 0))>> : std::true_type {};
// Epilogues may opt in (static constexpr bool PIN_SCHEDULE = true) to the hand-pinned schedule of the N -> N kernels
// (fragment batch in front of the MFMAs, two accumulation chains, prefetched operands held behind the MFMA block):
// measured on the out-projection + LayerNorm kernel; other shapes were faster with the compiler's own schedule.
template <class E, class = void>
struct epi_pins : std::false_type {};
template <class E>
struct epi_pins<E, std::enable_if_t<E::PIN_SCHEDULE>> : std::true_type {};
// Loaders / epilogues may offer BRANCH-FREE variants `load4c` / `prefetchc` that clamp an out-of-range row to the last
// valid one instead of returning zeros (the engine never stores such rows, so their content is irrelevant to it).
// Straight-line code lets the compiler count outstanding memory operations exactly: with the bounds branches it fell
// back to s_waitcnt vmcnt(0) in front of the MFMA block (exposing the prefetch it had just issued) and at the loop top
// (waiting for the previous tile's stores to retire).
template <class T, class = void>
struct has_load4c : std::false_type {};
template <class T>
struct has_load4c<T, std::void_t<decltype(std::declval<const T&>().load4c(0, 0, 0))>> : std::true_type {};
template <class T, class = void>
struct has_prefetchc : std::false_type {};
template <class T>
struct has_prefetchc<T, std::void_t<decltype(std::declval<const T&>().prefetchc(0, 0, 0))>> : std::true_type {};
struct EpiColsFallback { using Cols = EpiNoCols; };

// operand fetch of the weight-gradient code: the loader's branch-free zero-filling form where it has one
template <class L, class = void>
struct has_load4z : std::false_type {};
template <class L>
struct has_load4z<L, std::void_t<decltype(std::declval<const L&>().load4z(0, 0, 0))>> : std::true_type {};
template <class L>
DEV float4 wg_load(const L& l, int tile, int row, int k4) {
  if constexpr (has_load4z<L>::value) return l.load4z(tile, row, k4);
  else return l.load4(tile, row, k4);
}

// RIDER: a data-gradient GEMM  dX = dY W  stages the very tile of dY that the weight gradient of the same layer,
// dW[KIN][KK] = sum_tokens dY^T X  (+ db = column sums of dY), needs as its A operand.  With a rider the kernel also
// stages the X tile, runs the 16 x (KIN/128) x (KK/32) MFMAs of the weight gradient per tile on accumulators that live
// through the whole launch, and leaves one partial dW per workgroup (fragment order, slab_reduce_frag_kernel) -- instead
// of a second kernel (wgrad_kernel) that reads dY again and pays its own staging, barriers, tickets and store tail.
// Rows >= M: the X loader zero-fills (the engine's A loader clamps), the column sums skip them.
struct NoRider {
  static constexpr bool ON = false;
};
template <class R, class = void>
struct rider_kk : std::integral_constant<int, 32> {};
template <class R>
struct rider_kk<R, std::enable_if_t<R::ON>> : std::integral_constant<int, R::KK> {};
template <class R, class = void>
struct rider_colsum : std::false_type {};
template <class R>
struct rider_colsum<R, std::enable_if_t<R::ON>> : std::integral_constant<bool, R::COLSUM> {};
template <int KK_, class XLoad, bool COLSUM_>
struct WgradRider {
  static constexpr bool ON = true;
  static constexpr int KK = KK_;
  static constexpr bool COLSUM = COLSUM_;
  XLoad xl;
  float* slab;      // [gridDim.x][KIN * KK] partial tiles, fragment order
  float* colslab;   // [gridDim.x][KIN] partial column sums of dY (COLSUM)
  int64_t M;
};

template <class R>
DEV float4 rider_load(const R& r, int tile, int row, int k4) {
  if constexpr (R::ON) return wg_load(r.xl, tile, row, k4);
  else return make_float4(0.f, 0.f, 0.f, 0.f);
}
template <class R>
DEV int64_t rider_rows(const R& r) {
  if constexpr (R::ON) return r.M;
  else return 0;
}

// partial dW of this workgroup in fragment order (wgrad_kernel's layout: slab_reduce_frag_kernel<RB, CB>), and the
// partial column sums: a thread's accumulator belongs to column block tid % (KIN/4); owner thread cb < KIN/4 sums its
// contributors in tid order (fixed association order)
template <int KIN, int RB, int CB, bool CS, class R, class Acc>
DEV void rider_finish(const R& r, Acc& racc, float4 rcsum, float* smem, int tid, int wave, int lane) {
  if constexpr (R::ON) {
    float* out = r.slab + (size_t)blockIdx.x * (KIN * CB * 32);
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
      for (int j = 0; j < CB; ++j)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
          *reinterpret_cast<float4*>(out + ((((size_t)(wave * RB + i) * CB + j) * 4 + g4) * 64 + lane) * 4) =
              make_float4(racc[i][j][4 * g4], racc[i][j][4 * g4 + 1], racc[i][j][4 * g4 + 2], racc[i][j][4 * g4 + 3]);
    if constexpr (CS) {
      constexpr int Y4 = KIN / 4;
      __syncthreads();
      float4* red = reinterpret_cast<float4*>(smem);
      red[tid] = rcsum;
      __syncthreads();
      if (tid < Y4) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int t2 = tid; t2 < 256; t2 += Y4) {
          const float4 u = red[t2];
          a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
        }
        *reinterpret_cast<float4*>(r.colslab + (size_t)blockIdx.x * KIN + 4 * tid) = a;
      }
    }
  }
}

template <int KIN, int NT, int WR, int WC>
struct GemmShape {
  static constexpr int BM = 32 * WR;
  static constexpr int WGCOLS = 32 * NT * WC;
  static constexpr int LDA = KIN + 4;      // +4 floats: conflict-free ds_read_b128 fragments
  static constexpr int LDC = WGCOLS + 4;
  static constexpr int KS = KIN / 2;       // MFMA k-steps
  static constexpr int LDAB = KIN + 8;     // split mode: bf16 rows (hi and lo images), conflict-free ds_read_b128
  static constexpr size_t lds_bytes(bool direct, bool split = false, int rider_kk = 0) {
    // split mode: the double-buffered A tile is two bf16 images (hi, lo) instead of one fp32 image
    return sizeof(float) * (4 + (split ? 2 * (size_t)BM * LDAB : 2 * (size_t)BM * LDA) + (direct ? 0 : (size_t)BM * LDC) +
                            (rider_kk ? 2 * 32 * (size_t)(rider_kk + 4) : 0));
  }
};

// WT = true: the B fragments are fetched from W TRANSPOSED, i.e. out[m][j] = sum_k A[m][k] * W[k][j] with W stored
// [KIN][ldw] -- the data-gradient form of a layer whose forward weight is W[NOUT_fwd = KIN][KIN_fwd = out cols].
// SPLIT = true (OPT-IN mode "split_bf16", never the default): the product runs on v_mfma_f32_32x32x16_bf16 with both
// operands split into bf16 hi + lo -- A while it is staged to LDS, W while it is loaded into registers (same register
// bytes) -- as hi*hi + hi*lo + lo*hi with fp32 accumulation: 24 bf16 MFMAs of 32 cycles replace 64 fp32 ones of 64 per
// 32x32x128 block.  Accumulator layout, loaders and epilogues are shared with the fp32 form.
template <int KIN, int NT, int WR, int WC, class ALoad, class Epi, bool WT = false, bool SPLIT = false, class Rider = NoRider>
__global__ __launch_bounds__(256) void gemm_ws_kernel(const float* __restrict__ W,
                                                       const float* __restrict__ Walt, int ldw, int ntiles,
                                                       unsigned* __restrict__ tile_queue, ALoad aload, Epi epi,
                                                       Rider rider = Rider{}) {
  using Sh = GemmShape<KIN, NT, WR, WC>;
  static_assert(WR * WC == 4, "4 waves");
  GEMM_STAMP_ENTRY
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int* s_next = reinterpret_cast<int*>(smem);       // [2] tile tickets (double buffered), 16-byte slot
  float* As = smem + 4;
  float* Cs = As + (SPLIT ? 2 * Sh::BM * Sh::LDAB : 2 * Sh::BM * Sh::LDA);
  // RIDER: X tiles [2][32][KK + 4] behind the C tile
  constexpr int RKK = rider_kk<Rider>::value, LDXR = RKK + 4;
  float* Xr = Cs + (Epi::DIRECT ? 0 : Sh::BM * Sh::LDC);
  static_assert(!Rider::ON || (!SPLIT && WR == 1 && WC == 4 && KIN % 128 == 0 && RKK % 32 == 0), "rider shape");
  __bf16* Ahi = reinterpret_cast<__bf16*>(As);                 // SPLIT: [2][BM][LDAB] hi image, then the lo image
  __bf16* Alo = Ahi + 2 * Sh::BM * Sh::LDAB;
  static_assert(!(SPLIT && WT), "split mode: forward weight layout only");

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, c = lane & 31, hh = lane >> 5;
  const int wr = wave / WC, wc = wave % WC;   // wave-uniform (scalar registers)
  const int colgroup = blockIdx.y;  // which WGCOLS-wide slice of NOUT

  // ---- weights -> B fragments, once ----------------------------------------------------------
  // (Walt != null: column group 1 reads a second weight tensor, e.g. the reverse-direction W_ih)
  const float* Wsel = (Walt != nullptr && colgroup == 1) ? Walt : W;
  const int jbase = Walt != nullptr ? 0 : colgroup * Sh::WGCOLS;
  constexpr int KM = KIN / 16;                       // SPLIT: bf16 MFMAs per product term
  float wf[SPLIT ? 1 : NT][SPLIT ? 1 : Sh::KS];
  bf16x8 whi[SPLIT ? NT : 1][SPLIT ? KM : 1], wlo[SPLIT ? NT : 1][SPLIT ? KM : 1];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int j = jbase + (wc * NT + nt) * 32 + c;
    if constexpr (SPLIT) {
      const float* wrow = Wsel + (int64_t)j * ldw + 8 * hh;
#pragma unroll
      for (int m = 0; m < KM; ++m) {
        const float4 v0 = *reinterpret_cast<const float4*>(wrow + 16 * m), v1 = *reinterpret_cast<const float4*>(wrow + 16 * m + 4);
        const float xv[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          __bf16 hi, lo;
          split_bf16(xv[e], hi, lo);
          whi[nt][m][e] = hi;
          wlo[nt][m][e] = lo;
        }
      }
    } else if constexpr (!WT) {
      // ldw == 0: W is a FRAGMENT-ORDER copy (gemm_pack_rows_kernel: [column block 32][k chunk 8][lane 64][4]) -- a wave's
      // fragment load is 1 KiB contiguous.  Straight from a row-major tensor it is row-per-lane: 32 cache lines per wave
      // instruction, and the prologue of a workgroup that loads a 512-column slice costs 24 k cycles (1.2 tiles' worth,
      // tools/microbench/gemm_phases) against ~6 k from the copy.  Same values either way: bit-identical results.
      const float* wrow = ldw == 0 ? Wsel + ((int64_t)(j >> 5) * (KIN / 8)) * 256 + 4 * (hh * 32 + c)
                                   : Wsel + (int64_t)j * ldw + 4 * hh;
      const int mstep = ldw == 0 ? 256 : 8;
#pragma unroll
      for (int m = 0; m < KIN / 8; ++m) {
        const float4 v = *reinterpret_cast<const float4*>(wrow + mstep * m);
        wf[nt][4 * m + 0] = v.x;
        wf[nt][4 * m + 1] = v.y;
        wf[nt][4 * m + 2] = v.z;
        wf[nt][4 * m + 3] = v.w;
      }
    } else {
      const float* wcol = Wsel + j + (int64_t)(4 * hh) * ldw;   // element (k, j) at W[k*ldw + j]
#pragma unroll
      for (int m = 0; m < KIN / 8; ++m)
#pragma unroll
        for (int t = 0; t < 4; ++t) wf[nt][4 * m + t] = wcol[(int64_t)(8 * m + t) * ldw];
    }
  }

  constexpr bool FOLD = !SPLIT && !WT && epi_folds<Epi>::value;
  if constexpr (FOLD) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const float cs = epi.col_scale(wc * NT + nt);
#pragma unroll
      for (int k = 0; k < Sh::KS; ++k) wf[nt][k] *= cs;
    }
  }

  constexpr int K4 = KIN / 4;                       // float4 per A row
  constexpr int NLD = (Sh::BM * K4) / 256;          // float4 loads per thread per tile
  static_assert((Sh::BM * K4) % 256 == 0, "tile/threads");
  constexpr int C4 = Sh::WGCOLS / 4;                // float4 per C row
  constexpr int NPASS = Epi::DIRECT ? 1 : (Sh::BM * C4) / 256;
  static_assert(Epi::DIRECT || (Sh::BM * C4) % 256 == 0, "epilogue mapping");

  // Software pipeline: the A tile of iteration i+1 is fetched into registers while iteration i computes;
  // As is double buffered, so one barrier orders "tile written" -> "fragments read" and nothing else is needed
  // for As (a buffer is rewritten two iterations later, behind the next iteration's barrier).
  // Tiles are handed out by a device-wide ticket counter (one per column group, zeroed by the host before the
  // launch) instead of a static grid-stride: when this kernel shares the chip with another stream's kernel
  // (dptnav_forward overlaps two half-batches), workgroups that start late simply find fewer tickets left.
  // (thread 0 always holds the ticket AFTER the next one in a register, so the atomic's round trip -- a microsecond
  // when exposed in front of a barrier -- overlaps a whole tile of MFMAs)
  // per-column epilogue constants of this thread's NPASS row-space slots (identical across tiles)
  using EpiCols = typename std::conditional<epi_has_cols<Epi>::value, Epi, EpiColsFallback>::type::Cols;
  // (hoisted out of the tile loop unless the kernel is already at its register limit: with KIN = 256 the twelve extra
  //  registers turned into AGPR shuffles inside the loop, +3.6 %, with or without smaller fragment batches)
  constexpr bool PIN = !SPLIT && epi_pins<Epi>::value && NT == 1 && KIN < 256;
  constexpr bool HOIST_COLS = epi_has_cols<Epi>::value && !Epi::DIRECT && KIN < 256;
  EpiCols ecols[NPASS];
  if constexpr (HOIST_COLS) {
#pragma unroll
    for (int p = 0; p < NPASS; ++p) ecols[p] = epi.cols(colgroup, (p * 256 + tid) % C4);
  }

  float dconst[NT];
  if constexpr (Epi::DIRECT && has_direct_const<Epi>::value) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) dconst[nt] = epi.direct_const(colgroup, wc * NT + nt, c);
  }
  // FOLD: the constant enters as D = ones x const: A fragment (row c, k-slot hh) = [1, 0], B fragment (k-slot hh, column c) =
  // [const, 0] -> every row of the tile starts from its column's constant
  const float fold_one = hh == 0 ? 1.0f : 0.0f;
  if constexpr (FOLD) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) dconst[nt] = hh == 0 ? dconst[nt] : 0.0f;
  }

  TileTickets tickets{tile_queue ? tile_queue + colgroup : nullptr, (int)blockIdx.x, (int)gridDim.x};
  int ticket_ahead = 0;
  if (tickets.dynamic()) {
    if (tid == 0) {
      s_next[0] = tickets.take();
      ticket_ahead = tickets.take();
    }
  } else if (tid == 0) {
    s_next[0] = tickets.first;
  }
  __syncthreads();
  float4 pf[NLD];
  // RIDER state: the weight-gradient accumulators of this workgroup (whole launch), the staged X rows, column sums of dY
  constexpr int RRB = KIN / 128, RCB = RKK / 32, RX4 = RKK / 4, RNX = Rider::ON ? (32 * RX4) / 256 : 1;
  constexpr bool RCS = rider_colsum<Rider>::value;
  static_assert(!RCS || 256 % K4 == 0, "rider column sums: a thread's staging slots share their column block");
  f32x16 racc[Rider::ON ? RRB : 1][Rider::ON ? RCB : 1];
  float4 px[RNX];
  float4 rcsum = make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (Rider::ON) {
#pragma unroll
    for (int a = 0; a < RRB; ++a)
#pragma unroll
      for (int b = 0; b < RCB; ++b) racc[a][b] = zero16();
  }
  int tile = __builtin_amdgcn_readfirstlane(s_next[0]);
  auto load_tile = [&](int t) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = i * 256 + tid;
      if constexpr (has_load4c<ALoad>::value) pf[i] = aload.load4c(t, idx / K4, idx % K4);
      else pf[i] = aload.load4(t, idx / K4, idx % K4);
    }
    if constexpr (Rider::ON) {
#pragma unroll
      for (int i = 0; i < RNX; ++i) {
        const int idx = i * 256 + tid;
        px[i] = rider_load(rider, t, idx / RX4, idx % RX4);
      }
    }
  };
  if (tile < ntiles) load_tile(tile);
  int buf = 0;
  GEMM_STAMP_DECL
  while (tile < ntiles) {
    float* Ab = As + buf * (Sh::BM * Sh::LDA);
    if (tickets.dynamic() && tid == 0) s_next[buf ^ 1] = ticket_ahead;   // publish the next ticket (requested one tile ago)
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = i * 256 + tid;
      if constexpr (SPLIT) {
        const float xv[4] = {pf[i].x, pf[i].y, pf[i].z, pf[i].w};
        bf16x4 h4, l4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          __bf16 hi, lo;
          split_bf16(xv[e], hi, lo);
          h4[e] = hi;
          l4[e] = lo;
        }
        const int off = buf * (Sh::BM * Sh::LDAB) + (idx / K4) * Sh::LDAB + 4 * (idx % K4);
        *reinterpret_cast<bf16x4*>(&Ahi[off]) = h4;
        *reinterpret_cast<bf16x4*>(&Alo[off]) = l4;
      } else {
        *reinterpret_cast<float4*>(&Ab[(idx / K4) * Sh::LDA + 4 * (idx % K4)]) = pf[i];
      }
      if constexpr (RCS) {   // db: rows beyond M are clamped copies here and must not count
        if ((int64_t)tile * Sh::BM + idx / K4 < rider_rows(rider)) {
          rcsum.x += pf[i].x; rcsum.y += pf[i].y; rcsum.z += pf[i].z; rcsum.w += pf[i].w;
        }
      }
    }
    if constexpr (Rider::ON) {
      float* Xb = Xr + buf * (32 * LDXR);
#pragma unroll
      for (int i = 0; i < RNX; ++i) {
        const int idx = i * 256 + tid;
        *reinterpret_cast<float4*>(&Xb[(idx / RX4) * LDXR + 4 * (idx % RX4)]) = px[i];
      }
    }
    GEMM_STAMP(0);     // (diagnostic builds: phase 0 = ticket + A tile -> LDS, phase 1 = the barrier)
    __syncthreads();   // also orders the previous iteration's Cs reads before this iteration's Cs writes
    GEMM_STAMP(1);

    // wave-uniform: scalar address arithmetic
    const int next = tickets.dynamic() ? __builtin_amdgcn_readfirstlane(s_next[buf ^ 1]) : tile + tickets.stride;
    if (next < ntiles) load_tile(next);
    // epilogue operands that do not depend on the product (residual rows ...) are requested now as well
    float4 epf[NPASS];
    float4 epf2[has_prefetch2<Epi>::value ? NPASS : 1];
    if constexpr (!Epi::DIRECT) {
#pragma unroll
      for (int p = 0; p < NPASS; ++p) {
        const int idx = p * 256 + tid;
        if constexpr (has_prefetchc<Epi>::value) epf[p] = epi.prefetchc(tile, idx / C4, idx % C4);
        else epf[p] = epi.prefetch(tile, idx / C4, idx % C4);
        if constexpr (has_prefetch2<Epi>::value) epf2[p] = epi.prefetch2(tile, idx / C4, idx % C4);
      }
    }

    // request the ticket after the next one HERE, in front of the MFMA block: issued at the loop top it was the youngest
    // memory operation when the A-tile registers are waited for (s_waitcnt vmcnt(0)), i.e. wave 0 sat out the atomic's
    // round trip every tile and the other waves waited for it at the barrier (~1 k cycles per tile)
    if (tickets.dynamic() && tid == 0) ticket_ahead = tickets.take();
    // The ticket must stay in a VGPR until it is published at the top of the next iteration.  These kernels fill the
    // register file, and where the allocator parked `ticket_ahead` in an AGPR (a register spill) it had to wait for the
    // atomic's result RIGHT HERE to copy it over -- s_waitcnt vmcnt(0) behind the A-tile prefetch, i.e. the prefetch's HBM
    // latency plus the atomic's round trip exposed on every tile (the K4 pre-activation GEMM: 0.79 -> 0.72 ms per launch
    // alone on the chip once it was gone).  TICKET_KEEP marks a use inside the MFMA block, which makes spilling it
    // expensive for the allocator; tools/ticket_waits.py checks the generated code of every instantiation.
#define TICKET_KEEP asm volatile("" ::"v"(ticket_ahead))
    GEMM_STAMP(2);
    // ---- A fragments + MFMA --------------------------------------------------------------------
    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      if constexpr (FOLD) acc[nt] = mfma32(fold_one, dconst[nt], zero16());
      else acc[nt] = zero16();
    }
    f32x16 acc_odd = zero16();   // (PIN only)
    if constexpr (SPLIT) {
      const int rowoff = buf * (Sh::BM * Sh::LDAB) + (wr * 32 + c) * Sh::LDAB + 8 * hh;
      bf16x8 ah[KM], al[KM];
#pragma unroll
      for (int m = 0; m < KM; ++m) {
        ah[m] = *reinterpret_cast<const bf16x8*>(&Ahi[rowoff + 16 * m]);
        al[m] = *reinterpret_cast<const bf16x8*>(&Alo[rowoff + 16 * m]);
      }
#pragma unroll
      for (int m = 0; m < KM; ++m)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          acc[nt] = mfma32_bf16(ah[m], whi[nt][m], acc[nt]);
          acc[nt] = mfma32_bf16(ah[m], wlo[nt][m], acc[nt]);
          acc[nt] = mfma32_bf16(al[m], whi[nt][m], acc[nt]);
          TICKET_KEEP;
        }
    } else {
      // fragments are fetched in batches of 16 x ds_read_b128 BEFORE the MFMAs that use them: a read placed
      // between MFMAs is followed by s_waitcnt lgkmcnt(0) and exposes a full LDS round trip per k-chunk
      const float* arow = &Ab[(wr * 32 + c) * Sh::LDA + 4 * hh];
      // (KIN = 192, the in-projection data gradient of the 64-feature model: 24 chunks = 2 x 12)
      constexpr int CH = KIN / 8, BATCH = CH < 16 ? CH : (CH % 16 == 0 ? 16 : (CH % 12 == 0 ? 12 : 8));
      static_assert(CH % BATCH == 0, "fragment batches must tile the k chunks");
#pragma unroll
      for (int m0 = 0; m0 < CH; m0 += BATCH) {
        float4 afr[BATCH];
#pragma unroll
        for (int m = 0; m < BATCH; ++m) afr[m] = *reinterpret_cast<const float4*>(arow + 8 * (m0 + m));
        // (without the wait AND the scheduling barrier the compiler pairs every read with its MFMAs and waits for it
        //  there -- 16 exposed LDS round trips per tile; MFMAs are no memory operations, a "memory" clobber alone does
        //  not keep them behind the reads)
        // (only for NT == 1: with several column blocks per wave the compiler's pairing hides the LDS latency behind
        //  the other blocks' MFMAs and was measured faster than the up-front wait)
        if constexpr (PIN) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int m = 0; m < BATCH; ++m) {
          TICKET_KEEP;
          const float av[4] = {afr[m].x, afr[m].y, afr[m].z, afr[m].w};
#pragma unroll
          for (int tt = 0; tt < 4; ++tt) {
            if constexpr (PIN) {
              // a single accumulator makes every MFMA wait for the previous one (64-cycle issue, ~80-cycle dependent
              // latency: +1 k cycles per 64 MFMAs): two interleaved chains, summed once at the end
              if (tt & 1) acc_odd = mfma32(av[tt], wf[0][4 * (m0 + m) + tt], acc_odd);
              else acc[0] = mfma32(av[tt], wf[0][4 * (m0 + m) + tt], acc[0]);
            } else {
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma32(av[tt], wf[nt][4 * (m0 + m) + tt], acc[nt]);
            }
          }
        }
      }
      if constexpr (PIN) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][r] += acc_odd[r];
      }
    }

    if constexpr (Rider::ON) {
      // dW += dY_tile^T X_tile: MFMA step s takes tokens (2s, 2s+1) as its two k-slots; wave w owns rows (w + 4 i) * 32 of dW
      const float* Xb = Xr + buf * (32 * LDXR);
#pragma unroll
      for (int s2 = 0; s2 < 16; ++s2) {
        const float* yrow = Ab + (2 * s2 + hh) * Sh::LDA + c;
        const float* xrow = Xb + (2 * s2 + hh) * LDXR + c;
        float ra[RRB], rb[RCB];
#pragma unroll
        for (int i = 0; i < RRB; ++i) ra[i] = yrow[(wave + 4 * i) * 32];
#pragma unroll
        for (int j = 0; j < RCB; ++j) rb[j] = xrow[j * 32];
#pragma unroll
        for (int i = 0; i < RRB; ++i)
#pragma unroll
          for (int j = 0; j < RCB; ++j) racc[i][j] = mfma32(ra[i], rb[j], racc[i][j]);
      }
    }
    // Keep epilogue arithmetic on the prefetched operands (bias + residual ...) behind the MFMA block: scheduled in
    // front of it, it waited (s_waitcnt vmcnt(0)) for loads issued a moment earlier.  The empty asm makes the operands
    // "produced" here; plain arithmetic is not ordered by sched_barrier at instruction selection.
    if constexpr (PIN && !Epi::DIRECT) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int p = 0; p < NPASS; ++p) asm volatile("" : "+v"(epf[p].x), "+v"(epf[p].y), "+v"(epf[p].z), "+v"(epf[p].w));
    }
    GEMM_STAMP_ACC(3, acc[NT - 1][15]);
    if constexpr (Epi::DIRECT) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        if constexpr (FOLD) epi.store_raw(tile, wr, colgroup, wc * NT + nt, acc[nt], c, hh);
        else if constexpr (has_direct_const<Epi>::value) epi.store_acc(tile, wr, colgroup, wc * NT + nt, acc[nt], c, hh, dconst[nt]);
        else epi.store_acc(tile, wr, colgroup, wc * NT + nt, acc[nt], c, hh);
      }
    } else {
      // ---- C tile -> LDS (each half-wave writes 128 B contiguous), then row-space epilogue ------
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int col = (wc * NT + nt) * 32 + c;
#pragma unroll
        for (int r = 0; r < 16; ++r) Cs[(wr * 32 + ROW32(r, hh)) * Sh::LDC + col] = acc[nt][r];
      }
      __syncthreads();
      GEMM_STAMP(4);
#pragma unroll
      for (int p = 0; p < NPASS; ++p) {
        const int idx = p * 256 + tid;
        const int row = idx / C4, c4 = idx % C4;
        const float4 v = *reinterpret_cast<const float4*>(&Cs[row * Sh::LDC + 4 * c4]);
        if constexpr (has_prefetch2<Epi>::value) {
          if constexpr (HOIST_COLS) epi.row(tile, row, colgroup, c4, v, epf[p], epf2[p], ecols[p]);
          else if constexpr (epi_has_cols<Epi>::value) epi.row(tile, row, colgroup, c4, v, epf[p], epf2[p], epi.cols(colgroup, c4));
          else epi.row(tile, row, colgroup, c4, v, epf[p], epf2[p]);
        } else {
          if constexpr (HOIST_COLS) epi.row(tile, row, colgroup, c4, v, epf[p], ecols[p]);
          else if constexpr (epi_has_cols<Epi>::value) epi.row(tile, row, colgroup, c4, v, epf[p], epi.cols(colgroup, c4));
          else epi.row(tile, row, colgroup, c4, v, epf[p]);
        }
      }
    }
    GEMM_STAMP(5);
    TICKET_KEEP;
#undef TICKET_KEEP
    tile = next;
    buf ^= 1;
  }
  GEMM_STAMP_END
  if constexpr (Epi::HAS_FINISH) epi.finish(smem, tid);   // e.g. per-workgroup partial sums of parameter gradients
  if constexpr (Rider::ON) rider_finish<KIN, RRB, RCB, RCS>(rider, racc, rcsum, smem, tid, wave, lane);
}

// ------------------------------------------------------------------------------------------------
// A loaders
// ------------------------------------------------------------------------------------------------
// rows are consecutive tokens of a dense [M][lda] matrix
struct ALoadDense {
  const float* A;
  int64_t M;
  int lda;
  int bm;
  DEV float4 load4(int tile, int row, int k4) const {
    // (wave-uniform tile base + an offset that does not change from tile to tile: no 64-bit multiply per load)
    const int64_t r0 = (int64_t)tile * bm;
    if (r0 + row >= M) return make_float4(0.f, 0.f, 0.f, 0.f);
    return *reinterpret_cast<const float4*>(A + r0 * lda + (unsigned)(row * lda + 4 * k4));
  }
  DEV float4 load4c(int tile, int row, int k4) const {
    const int64_t r0 = (int64_t)tile * bm;
    const int last = (int)(M - 1 - r0 < bm - 1 ? M - 1 - r0 : bm - 1);   // wave-uniform
    return *reinterpret_cast<const float4*>(A + r0 * lda + (unsigned)((row < last ? row : last) * lda + 4 * k4));
  }
  // zero-filling like load4, branch-free like load4c (weight-gradient kernels: loads issued between MFMAs)
  DEV float4 load4z(int tile, int row, int k4) const {
    const int64_t r0 = (int64_t)tile * bm;
    const int last = (int)(M - 1 - r0 < bm - 1 ? M - 1 - r0 : bm - 1);
    const bool ok = row <= last;
    const float4 v = *reinterpret_cast<const float4*>(A + r0 * lda + (unsigned)((ok ? row : last) * lda + 4 * k4));
    return mask4(v, ok);
  }
};

// dense rows with PReLU applied on the fly (dptn_wav.py:27 -- single shared slope)
struct ALoadDensePReLU {
  const float* A;
  const float* slope;
  int64_t M;
  int lda;
  int bm;
  DEV float4 load4(int tile, int row, int k4) const {
    const int64_t r = (int64_t)tile * bm + row;
    if (r >= M) return make_float4(0.f, 0.f, 0.f, 0.f);
    float4 v = *reinterpret_cast<const float4*>(A + r * lda + 4 * k4);
    const float a = *slope;
    v.x = v.x >= 0.f ? v.x : a * v.x;
    v.y = v.y >= 0.f ? v.y : a * v.y;
    v.z = v.z >= 0.f ? v.z : a * v.z;
    v.w = v.w >= 0.f ? v.w : a * v.w;
    return v;
  }
};

// tile = (sequence tile st, position t): the 32 rows are the 32 sequences of tile st at position t
struct ALoadSeqTile {
  const float* A;  // token-major [M][lda]
  int lda;
  SeqGeom g;
  DEV float4 load4(int tile, int row, int k4) const {
    const int st = tile / g.len, t = tile - st * g.len;
    const int q = st * 32 + row;
    if (q >= g.nseq) return make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t tok = seq_token_base(g, q) + (int64_t)t * seq_token_stride(g);
    return *reinterpret_cast<const float4*>(A + tok * lda + 4 * k4);
  }
  DEV float4 load4c(int tile, int row, int k4) const {   // padded sequences read the last real one
    const int st = tile / g.len, t = tile - st * g.len;
    const int q = st * 32 + row;
    const int64_t tok = seq_token_base(g, q < g.nseq ? q : g.nseq - 1) + (int64_t)t * seq_token_stride(g);
    return *reinterpret_cast<const float4*>(A + tok * lda + 4 * k4);
  }
};

// ------------------------------------------------------------------------------------------------
// epilogues
// ------------------------------------------------------------------------------------------------
// out[row][colgroup*WGCOLS + 4*c4 ..] = v + bias
struct EpiBiasStore {
  static constexpr bool DIRECT = false;
  static constexpr bool HAS_FINISH = false;
  float* out;
  const float* bias;
  int64_t M;
  int ldo;
  int bm;
  int wgcols;
  struct Cols { float4 b; };
  DEV Cols cols(int colgroup, int c4) const { return Cols{*reinterpret_cast<const float4*>(bias + colgroup * wgcols + 4 * c4)}; }
  DEV float4 prefetch(int, int, int) const { return make_float4(0.f, 0.f, 0.f, 0.f); }
  DEV void row(int tile, int row, int colgroup, int c4, float4 v, float4 /*pre*/, const Cols& k) const {
    const int64_t r0 = (int64_t)tile * bm;
    if (r0 + row >= M) return;
    v.x += k.b.x; v.y += k.b.y; v.z += k.b.z; v.w += k.b.w;
    *reinterpret_cast<float4*>(out + r0 * ldo + (unsigned)(row * ldo + colgroup * wgcols + 4 * c4)) = v;
  }
};

// y = LayerNorm(v + bias + residual) over the NCOL = 4*GROUP columns of the row (dptn.py:46-47, 50-51)
template <int GROUP>
struct EpiBiasResLN {
  static constexpr bool DIRECT = false;
  static constexpr bool HAS_FINISH = false;
  static constexpr bool PIN_SCHEDULE = true;
  float* out;
  const float* bias;
  const float* res;    // [M][ld]
  const float* gamma;
  const float* beta;
  int64_t M;
  int ld;
  int bm;
  struct Cols { float4 b, ga, be; };
  DEV Cols cols(int /*colgroup*/, int c4) const {
    return Cols{*reinterpret_cast<const float4*>(bias + 4 * c4), *reinterpret_cast<const float4*>(gamma + 4 * c4),
                *reinterpret_cast<const float4*>(beta + 4 * c4)};
  }
  DEV float4 prefetch(int tile, int row, int c4) const {   // the residual row: independent of the product
    const int64_t r0 = (int64_t)tile * bm;
    return r0 + row < M ? *reinterpret_cast<const float4*>(res + r0 * ld + (unsigned)(row * ld + 4 * c4))
                        : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  DEV float4 prefetchc(int tile, int row, int c4) const {
    const int64_t r0 = (int64_t)tile * bm;
    const int last = (int)(M - 1 - r0 < bm - 1 ? M - 1 - r0 : bm - 1);   // wave-uniform
    return *reinterpret_cast<const float4*>(res + r0 * ld + (unsigned)((row < last ? row : last) * ld + 4 * c4));
  }
  DEV void row(int tile, int row, int /*colgroup*/, int c4, float4 v, float4 x, const Cols& k) const {
    const int64_t r0 = (int64_t)tile * bm;
    const bool ok = r0 + row < M;
    v.x += k.b.x + x.x; v.y += k.b.y + x.y; v.z += k.b.z + x.z; v.w += k.b.w + x.w;
    const float s = group_sum<GROUP>((v.x + v.y) + (v.z + v.w));
    const float mu = s * (1.0f / (4 * GROUP));
    const float dx = v.x - mu, dy = v.y - mu, dz = v.z - mu, dw = v.w - mu;
    const float q = group_sum<GROUP>((dx * dx + dy * dy) + (dz * dz + dw * dw));
    const float rstd = rsqrtf(q * (1.0f / (4 * GROUP)) + 1e-5f);
    if (!ok) return;
    float4 y;
    y.x = dx * rstd * k.ga.x + k.be.x;
    y.y = dy * rstd * k.ga.y + k.be.y;
    y.z = dz * rstd * k.ga.z + k.be.z;
    y.w = dw * rstd * k.ga.w + k.be.w;
    *reinterpret_cast<float4*>(out + r0 * ld + (unsigned)(row * ld + 4 * c4)) = y;
  }
};

// EpiBiasResLN that also leaves the normalised rows zn = (z - mu) rstd and rstd on the training tape: the backward then
// applies the LayerNorm derivative in a bandwidth-bound pass (ln_backward_kernel, backward.h) instead of recomputing the
// GEMM to rebuild z (the recompute + epilogue launches ran at 0.38-0.42 MFMA-busy: 8 % of a training step's kernel time).
template <int GROUP>
struct EpiBiasResLNSave {
  static constexpr bool DIRECT = false;
  static constexpr bool HAS_FINISH = false;
  static constexpr bool PIN_SCHEDULE = true;
  float* out;
  const float* bias;
  const float* res;    // [M][ld]
  const float* gamma;
  const float* beta;
  int64_t M;
  int ld;
  int bm;
  float* zn_out;       // [M][ld]
  float* rstd_out;     // [M]
  struct Cols { float4 b, ga, be; };
  DEV Cols cols(int /*colgroup*/, int c4) const {
    return Cols{*reinterpret_cast<const float4*>(bias + 4 * c4), *reinterpret_cast<const float4*>(gamma + 4 * c4),
                *reinterpret_cast<const float4*>(beta + 4 * c4)};
  }
  DEV float4 prefetch(int tile, int row, int c4) const {
    const int64_t r0 = (int64_t)tile * bm;
    return r0 + row < M ? *reinterpret_cast<const float4*>(res + r0 * ld + (unsigned)(row * ld + 4 * c4))
                        : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  DEV float4 prefetchc(int tile, int row, int c4) const {
    const int64_t r0 = (int64_t)tile * bm;
    const int last = (int)(M - 1 - r0 < bm - 1 ? M - 1 - r0 : bm - 1);   // wave-uniform
    return *reinterpret_cast<const float4*>(res + r0 * ld + (unsigned)((row < last ? row : last) * ld + 4 * c4));
  }
  DEV void row(int tile, int row, int /*colgroup*/, int c4, float4 v, float4 x, const Cols& k) const {
    const int64_t r0 = (int64_t)tile * bm;
    const bool ok = r0 + row < M;
    v.x += k.b.x + x.x; v.y += k.b.y + x.y; v.z += k.b.z + x.z; v.w += k.b.w + x.w;
    const float s = group_sum<GROUP>((v.x + v.y) + (v.z + v.w));
    const float mu = s * (1.0f / (4 * GROUP));
    const float dx = v.x - mu, dy = v.y - mu, dz = v.z - mu, dw = v.w - mu;
    const float q = group_sum<GROUP>((dx * dx + dy * dy) + (dz * dz + dw * dw));
    const float rstd = rsqrtf(q * (1.0f / (4 * GROUP)) + 1e-5f);
    if (!ok) return;
    const float4 zn = make_float4(dx * rstd, dy * rstd, dz * rstd, dw * rstd);
    float4 y;
    y.x = zn.x * k.ga.x + k.be.x;
    y.y = zn.y * k.ga.y + k.be.y;
    y.z = zn.z * k.ga.z + k.be.z;
    y.w = zn.w * k.ga.w + k.be.w;
    *reinterpret_cast<float4*>(out + r0 * ld + (unsigned)(row * ld + 4 * c4)) = y;
    *reinterpret_cast<float4*>(zn_out + r0 * ld + (unsigned)(row * ld + 4 * c4)) = zn;
    if (c4 == 0) rstd_out[r0 + row] = rstd;
  }
};

// y = LayerNorm(v + bias) + residual  (DPRNN: norm BEFORE the residual add, dprnn.py:41-45, 83-87)
template <int GROUP>
struct EpiBiasLNRes {
  static constexpr bool DIRECT = false;
  static constexpr bool HAS_FINISH = false;
  float* out;
  const float* bias;
  const float* res;    // [M][ld]
  const float* gamma;
  const float* beta;
  int64_t M;
  int ld;
  int bm;
  struct Cols { float4 b, ga, be; };
  DEV Cols cols(int /*colgroup*/, int c4) const {
    return Cols{*reinterpret_cast<const float4*>(bias + 4 * c4), *reinterpret_cast<const float4*>(gamma + 4 * c4),
                *reinterpret_cast<const float4*>(beta + 4 * c4)};
  }
  DEV float4 prefetch(int tile, int row, int c4) const {
    const int64_t r = (int64_t)tile * bm + row;
    return r < M ? *reinterpret_cast<const float4*>(res + r * ld + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  DEV float4 prefetchc(int tile, int row, int c4) const {   // clamped, branch-free (rows beyond M are never stored)
    const int64_t r0 = (int64_t)tile * bm;
    const int last = (int)(M - 1 - r0 < bm - 1 ? M - 1 - r0 : bm - 1);   // wave-uniform
    return *reinterpret_cast<const float4*>(res + r0 * ld + (unsigned)((row < last ? row : last) * ld + 4 * c4));
  }
  DEV void row(int tile, int row, int /*colgroup*/, int c4, float4 v, float4 x, const Cols& k) const {
    const int64_t r0 = (int64_t)tile * bm;
    v.x += k.b.x; v.y += k.b.y; v.z += k.b.z; v.w += k.b.w;
    const float s = group_sum<GROUP>((v.x + v.y) + (v.z + v.w));
    const float mu = s * (1.0f / (4 * GROUP));
    const float dx = v.x - mu, dy = v.y - mu, dz = v.z - mu, dw = v.w - mu;
    const float q = group_sum<GROUP>((dx * dx + dy * dy) + (dz * dz + dw * dw));
    const float rstd = rsqrtf(q * (1.0f / (4 * GROUP)) + 1e-5f);
    if (r0 + row >= M) return;
    float4 y;
    y.x = dx * rstd * k.ga.x + k.be.x + x.x;
    y.y = dy * rstd * k.ga.y + k.be.y + x.y;
    y.z = dz * rstd * k.ga.z + k.be.z + x.z;
    y.w = dw * rstd * k.ga.w + k.be.w + x.w;
    *reinterpret_cast<float4*>(out + r0 * ld + (unsigned)(row * ld + 4 * c4)) = y;
  }
};

// EpiBiasLNRes that also leaves zn / rstd on the training tape (DPRNN blocks; see EpiBiasResLNSave)
template <int GROUP>
struct EpiBiasLNResSave {
  static constexpr bool DIRECT = false;
  static constexpr bool HAS_FINISH = false;
  float* out;
  const float* bias;
  const float* res;    // [M][ld]
  const float* gamma;
  const float* beta;
  int64_t M;
  int ld;
  int bm;
  float* zn_out;       // [M][ld]
  float* rstd_out;     // [M]
  struct Cols { float4 b, ga, be; };
  DEV Cols cols(int /*colgroup*/, int c4) const {
    return Cols{*reinterpret_cast<const float4*>(bias + 4 * c4), *reinterpret_cast<const float4*>(gamma + 4 * c4),
                *reinterpret_cast<const float4*>(beta + 4 * c4)};
  }
  DEV float4 prefetch(int tile, int row, int c4) const {
    const int64_t r = (int64_t)tile * bm + row;
    return r < M ? *reinterpret_cast<const float4*>(res + r * ld + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  DEV float4 prefetchc(int tile, int row, int c4) const {   // clamped, branch-free (rows beyond M are never stored)
    const int64_t r0 = (int64_t)tile * bm;
    const int last = (int)(M - 1 - r0 < bm - 1 ? M - 1 - r0 : bm - 1);   // wave-uniform
    return *reinterpret_cast<const float4*>(res + r0 * ld + (unsigned)((row < last ? row : last) * ld + 4 * c4));
  }
  DEV void row(int tile, int row, int /*colgroup*/, int c4, float4 v, float4 x, const Cols& k) const {
    const int64_t r0 = (int64_t)tile * bm;
    v.x += k.b.x; v.y += k.b.y; v.z += k.b.z; v.w += k.b.w;
    const float s = group_sum<GROUP>((v.x + v.y) + (v.z + v.w));
    const float mu = s * (1.0f / (4 * GROUP));
    const float dx = v.x - mu, dy = v.y - mu, dz = v.z - mu, dw = v.w - mu;
    const float q = group_sum<GROUP>((dx * dx + dy * dy) + (dz * dz + dw * dw));
    const float rstd = rsqrtf(q * (1.0f / (4 * GROUP)) + 1e-5f);
    if (r0 + row >= M) return;
    const float4 zn = make_float4(dx * rstd, dy * rstd, dz * rstd, dw * rstd);
    float4 y;
    y.x = zn.x * k.ga.x + k.be.x + x.x;
    y.y = zn.y * k.ga.y + k.be.y + x.y;
    y.z = zn.z * k.ga.z + k.be.z + x.z;
    y.w = zn.w * k.ga.w + k.be.w + x.w;
    *reinterpret_cast<float4*>(out + r0 * ld + (unsigned)(row * ld + 4 * c4)) = y;
    *reinterpret_cast<float4*>(zn_out + r0 * ld + (unsigned)(row * ld + 4 * c4)) = zn;
    if (c4 == 0) rstd_out[r0 + row] = rstd;
  }
};

// LSTM pre-activations straight from the accumulators into the fragment layout (common.h)
struct EpiLstmPre {
  static constexpr bool DIRECT = true;
  static constexpr bool HAS_FINISH = false;
  float* pre;
  const float* b_ih[2];
  const float* b_hh[2];
  SeqGeom g;
  // gate rows leave pre-scaled by -log2(e) (i, f, o) or -2 log2(e) (g): lstm.hip evaluates the activations without a
  // multiply (cb >> 2 is the gate; the bias add became an fma)
  DEV float direct_const(int d, int cb, int c) const {
    const int j = cb * 32 + c;
    return (b_ih[d][j] + b_hh[d][j]) * lstm_gate_scale(cb >> 2);
  }
  static constexpr bool FOLD = true;   // fp32 engine: scale folded into W_ih's fragments, bias into the accumulators' start
  DEV float col_scale(int cb) const { return lstm_gate_scale(cb >> 2); }
  DEV void store_raw(int tile, int /*wr*/, int d, int cb, const f32x16& acc, int c, int hh) const {
    const int st = tile / g.len, t = tile - st * g.len;
    float* base = pre + pre_tile_offset(d, st, t, g.nst, g.len) + (int64_t)cb * 1024 + hh * 128 + c * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      *reinterpret_cast<float4*>(base + q * 256) = make_float4(acc[4 * q + 0], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
  }
  DEV void store_acc(int tile, int /*wr*/, int d, int cb, const f32x16& acc, int c, int hh, float bias) const {
    const int st = tile / g.len, t = tile - st * g.len;
    const float gs = lstm_gate_scale(cb >> 2);
    float* base = pre + pre_tile_offset(d, st, t, g.nst, g.len) + (int64_t)cb * 1024 + hh * 128 + c * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float4 v;
      v.x = fmaf(acc[4 * q + 0], gs, bias);
      v.y = fmaf(acc[4 * q + 1], gs, bias);
      v.z = fmaf(acc[4 * q + 2], gs, bias);
      v.w = fmaf(acc[4 * q + 3], gs, bias);
      *reinterpret_cast<float4*>(base + q * 256) = v;
    }
  }
};


// Same, in the 16-sequence-tile layout of lstm16.h: PRE16[d][st16][t][w][b][lane64][4], pre-scaled per gate.
// Accumulator registers 4q..4q+3 of lane (c,hh) are rows 8q+4hh+i of the 32-sequence GEMM tile -> 16-sequence tile
// 2*st + (q>>1), ks = 2(q&1)+hh.  Column block cb (32 of the direction's 512 gate columns): gate = cb >> 2,
// wave w = cb & 3 (hidden units [32w, 32w+32)), half = c >> 4, i16 = c & 15.
struct EpiLstmPre16 {
  static constexpr bool DIRECT = true;
  static constexpr bool HAS_FINISH = false;
  float* pre;
  const float* b_ih[2];
  const float* b_hh[2];
  SeqGeom g;
  int nst16;
  // gate rows leave pre-scaled by -log2(e) (i, f, o) or -2 log2(e) (g): lstm16.hip evaluates the activations without a
  // multiply
  DEV float direct_const(int d, int cb, int c) const {
    const int j = cb * 32 + c;
    return (b_ih[d][j] + b_hh[d][j]) * l16_gate_scale(cb >> 2);
  }
  static constexpr bool FOLD = true;   // fp32 engine: scale folded into W_ih's fragments, bias into the accumulators' start
  DEV float col_scale(int cb) const { return l16_gate_scale(cb >> 2); }
  DEV void store_raw(int tile, int /*wr*/, int d, int cb, const f32x16& acc, int c, int hh) const {
    const int st = tile / g.len, t = tile - st * g.len;
    const int gate = cb >> 2;
    const int lane_off = (c >> 4) * 256 + (hh * 16 + (c & 15)) * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int st16 = 2 * st + (q >> 1);
      if (st16 >= nst16) continue;   // a 16-sequence tile made of padding only
      float* ubase = pre + pre16_tile_offset(d, st16, t, nst16, g.len) + (cb & 3) * 2048 + 2 * gate * 256 + (q & 1) * 128;
      *reinterpret_cast<float4*>(ubase + lane_off) = make_float4(acc[4 * q + 0], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
    }
  }
  DEV void store_acc(int tile, int /*wr*/, int d, int cb, const f32x16& acc, int c, int hh, float bias) const {
    const int st = tile / g.len, t = tile - st * g.len;
    const int gate = cb >> 2;
    const float gs = l16_gate_scale(gate);
    // wave-uniform part of the address (tile, column block) + the lane's part
    const int lane_off = (c >> 4) * 256 + (hh * 16 + (c & 15)) * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int st16 = 2 * st + (q >> 1);
      if (st16 >= nst16) continue;   // a 16-sequence tile made of padding only
      float* ubase = pre + pre16_tile_offset(d, st16, t, nst16, g.len) + (cb & 3) * 2048 + 2 * gate * 256 + (q & 1) * 128;
      *reinterpret_cast<float4*>(ubase + lane_off) = make_float4(fmaf(acc[4 * q + 0], gs, bias), fmaf(acc[4 * q + 1], gs, bias),
                                                                 fmaf(acc[4 * q + 2], gs, bias), fmaf(acc[4 * q + 3], gs, bias));
    }
  }
};

// ------------------------------------------------------------------------------------------------
// training-step hooks (backward pass)
// ------------------------------------------------------------------------------------------------
// dense rows with ReLU applied on the fly (training keeps the raw LSTM output h; ffn = ReLU -> Linear, dptn.py:30-33)
struct ALoadDenseReLU {
  const float* A;
  int64_t M;
  int lda;
  int bm;
  DEV float4 load4(int tile, int row, int k4) const {
    const int64_t r = (int64_t)tile * bm + row;
    if (r >= M) return make_float4(0.f, 0.f, 0.f, 0.f);
    float4 v = *reinterpret_cast<const float4*>(A + r * lda + 4 * k4);
    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    return v;
  }
};

// out[row][col] = (v + addend[row][col]) * (gate[row][col] > 0 ? 1 : 0).  Which of the two operands exist is a
// COMPILE-TIME property (ADD, GATE): as null checks at run time they put branches into the prefetches and the engine's
// operand loads could not be issued as one batch (0.32 MFMA-busy on the d h launch).
template <bool ADD, bool GATE>
struct EpiAddMaskStoreT {
  static constexpr bool DIRECT = false;
  static constexpr bool HAS_FINISH = false;
  float* out;
  const float* addend;   // [M][ldo] (ADD)
  const float* gate;     // [M][ldo] (GATE): ReLU mask source (forward activation)
  int64_t M;
  int ldo;
  int bm;
  int wgcols;
  DEV int last_row(int64_t r0) const { return (int)(M - 1 - r0 < bm - 1 ? M - 1 - r0 : bm - 1); }   // wave-uniform
  DEV float4 prefetch(int tile, int row, int c4) const {   // (single column group only when an addend / gate is given)
    if constexpr (!ADD) return make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t r0 = (int64_t)tile * bm;
    const int last = last_row(r0);
    return *reinterpret_cast<const float4*>(addend + r0 * ldo + (unsigned)((row < last ? row : last) * ldo + 4 * c4));
  }
  DEV float4 prefetchc(int tile, int row, int c4) const { return prefetch(tile, row, c4); }
  DEV float4 prefetch2(int tile, int row, int c4) const {   // the ReLU-mask source row
    if constexpr (!GATE) return make_float4(1.f, 1.f, 1.f, 1.f);
    const int64_t r0 = (int64_t)tile * bm;
    const int last = last_row(r0);
    return *reinterpret_cast<const float4*>(gate + r0 * ldo + (unsigned)((row < last ? row : last) * ldo + 4 * c4));
  }
  DEV void row(int tile, int row, int colgroup, int c4, float4 v, float4 a, float4 g) const {
    const int64_t r0 = (int64_t)tile * bm;
    if (r0 + row >= M) return;
    if constexpr (ADD) { v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w; }
    if constexpr (GATE) {
      v.x = g.x > 0.f ? v.x : 0.f; v.y = g.y > 0.f ? v.y : 0.f; v.z = g.z > 0.f ? v.z : 0.f; v.w = g.w > 0.f ? v.w : 0.f;
    }
    *reinterpret_cast<float4*>(out + r0 * ldo + (unsigned)(row * ldo + colgroup * wgcols + 4 * c4)) = v;
  }
};

// Recompute z = v + bias + res (the pre-LayerNorm activation) and apply the LayerNorm backward to the incoming
// gradient:  dz = rstd * (g - mean(g) - zn * mean(g * zn)),  g = dout * gamma,  zn = (z - mu) * rstd;
// d gamma += dout * zn, d beta += dout (per-thread column sums, reduced per workgroup into `partials`).
// MODE 0: LayerNorm(v + bias + res)  (DPTN: dptn.py:46-47,50-51);  MODE 1: LayerNorm(v + bias) + res  (DPRNN).
template <int GROUP, int MODE>
struct EpiLNBackward {
  static constexpr bool DIRECT = false;
  static constexpr bool HAS_FINISH = true;
  float* dz;             // [M][ld] gradient w.r.t. the pre-norm activation
  const float* bias;
  const float* res;      // [M][ld]
  const float* gamma;
  const float* dout;     // [M][ld]
  float* partials;       // [gridDim.x][2 * 4*GROUP]  (d gamma | d beta)
  int64_t M;
  int ld;
  int bm;
  float4 sg = {0.f, 0.f, 0.f, 0.f}, sb = {0.f, 0.f, 0.f, 0.f};   // this thread's column sums (its c4 is fixed)
  struct Cols { float4 b, ga; };
  DEV Cols cols(int /*colgroup*/, int c4) const {
    return Cols{*reinterpret_cast<const float4*>(bias + 4 * c4), *reinterpret_cast<const float4*>(gamma + 4 * c4)};
  }
  DEV int last_row(int64_t r0) const { return (int)(M - 1 - r0 < bm - 1 ? M - 1 - r0 : bm - 1); }   // wave-uniform
  DEV float4 prefetch(int tile, int row, int c4) const {   // residual row (clamped: rows beyond M are never stored)
    const int64_t r0 = (int64_t)tile * bm;
    const int last = last_row(r0);
    return *reinterpret_cast<const float4*>(res + r0 * ld + (unsigned)((row < last ? row : last) * ld + 4 * c4));
  }
  DEV float4 prefetch2(int tile, int row, int c4) const {   // incoming gradient row
    const int64_t r0 = (int64_t)tile * bm;
    const int last = last_row(r0);
    return *reinterpret_cast<const float4*>(dout + r0 * ld + (unsigned)((row < last ? row : last) * ld + 4 * c4));
  }
  DEV void row(int tile, int row, int /*colgroup*/, int c4, float4 v, float4 x, float4 d, const Cols& k) {
    const int64_t r0 = (int64_t)tile * bm;
    const bool ok = r0 + row < M;
    v.x += k.b.x; v.y += k.b.y; v.z += k.b.z; v.w += k.b.w;
    if (MODE == 0) { v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w; }
    const float mu = group_sum<GROUP>((v.x + v.y) + (v.z + v.w)) * (1.0f / (4 * GROUP));
    const float dx = v.x - mu, dy = v.y - mu, dzz = v.z - mu, dw = v.w - mu;
    const float var = group_sum<GROUP>((dx * dx + dy * dy) + (dzz * dzz + dw * dw)) * (1.0f / (4 * GROUP));
    const float rstd = rsqrtf(var + 1e-5f);
    const float4 zn = make_float4(dx * rstd, dy * rstd, dzz * rstd, dw * rstd);
    if (!ok) d = make_float4(0.f, 0.f, 0.f, 0.f);   // padded rows must not reach the column sums
    sg.x += d.x * zn.x; sg.y += d.y * zn.y; sg.z += d.z * zn.z; sg.w += d.w * zn.w;
    sb.x += d.x; sb.y += d.y; sb.z += d.z; sb.w += d.w;
    const float4 g = make_float4(d.x * k.ga.x, d.y * k.ga.y, d.z * k.ga.z, d.w * k.ga.w);
    const float m1 = group_sum<GROUP>((g.x + g.y) + (g.z + g.w)) * (1.0f / (4 * GROUP));
    const float m2 = group_sum<GROUP>((g.x * zn.x + g.y * zn.y) + (g.z * zn.z + g.w * zn.w)) * (1.0f / (4 * GROUP));
    if (!ok) return;
    float4 o;
    o.x = rstd * (g.x - m1 - zn.x * m2);
    o.y = rstd * (g.y - m1 - zn.y * m2);
    o.z = rstd * (g.z - m1 - zn.z * m2);
    o.w = rstd * (g.w - m1 - zn.w * m2);
    *reinterpret_cast<float4*>(dz + r0 * ld + (unsigned)(row * ld + 4 * c4)) = o;
  }
  // 256 threads = (256/GROUP) row lanes x GROUP column lanes: reduce the row lanes through LDS
  DEV void finish(float* smem, int tid) {
    __syncthreads();
    float4* red = reinterpret_cast<float4*>(smem);          // [2][256]
    red[tid] = sg;
    red[256 + tid] = sb;
    __syncthreads();
    if (tid < GROUP) {
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
      for (int k = tid; k < 256; k += GROUP) {
        const float4 u = red[k], w = red[256 + k];
        a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
        b.x += w.x; b.y += w.y; b.z += w.z; b.w += w.w;
      }
      float* p = partials + (size_t)blockIdx.x * (8 * GROUP);
      *reinterpret_cast<float4*>(p + 4 * tid) = a;
      *reinterpret_cast<float4*>(p + 4 * GROUP + 4 * tid) = b;
    }
  }
};

// ------------------------------------------------------------------------------------------------
// fragment-order copies of row-major weight matrices W[rows][K] (rows % 32 == 0, K % 8 == 0) for the engine's ldw == 0 form:
//   dst[matrix][column block cb = row / 32][k chunk m = k / 8][lane = 32 hh + c][4] = W[32 cb + c][8 m + 4 hh .. + 3]
// ------------------------------------------------------------------------------------------------
constexpr int GEMM_PACK_MAX = 24;
struct GemmPackArgs {
  const float* src[GEMM_PACK_MAX];   // nullptr: that slot stays untouched (a path without a reverse direction)
  int rows, K;
};
__global__ __launch_bounds__(256) void gemm_pack_rows_kernel(GemmPackArgs a, float* __restrict__ dst) {
  const float* W = a.src[blockIdx.y];
  if (W == nullptr) return;
  const int km = a.K / 8, n4 = a.rows * a.K / 4;
  float4* out = reinterpret_cast<float4*>(dst + (size_t)blockIdx.y * a.rows * a.K);
  for (int f = blockIdx.x * 256 + threadIdx.x; f < n4; f += gridDim.x * 256) {
    const int lane = f & 63, m = (f >> 6) % km, cb = (f >> 6) / km;
    out[f] = *reinterpret_cast<const float4*>(W + (size_t)(32 * cb + (lane & 31)) * a.K + 8 * m + 4 * (lane >> 5));
  }
}
