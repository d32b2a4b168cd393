// backward.h -- parameter-gradient building blocks of the training step (BASELINE config 4), fp32, gfx950.
//
//   wgrad_kernel     dW[NN][KK] = sum over tokens  dY[m][0:NN]^T  X[m][0:KK]     (a GEMM whose K dimension is the
//                    3.4e5 tokens and whose output is tiny): every workgroup accumulates its share of 32-token tiles
//                    in MFMA accumulators and writes ONE partial tile to a slab; slab_reduce_kernel sums the slabs in
//                    a fixed order; no float atomics.  (Tiles are handed out by dynamic tickets: WHICH workgroup sums
//                    which tiles varies between runs, so results agree to fp32 summation order, not bit for bit.)
//   colsum_kernel    bias gradients: column sums of dY over the tokens, same slab scheme.
#pragma once
#include "lstm_bptt.h"
#include "common.h"
#include "gemm_ws.h"
#include "lstm.h"

// rows of a token-major matrix shifted by `shift` positions along the sequence (h_{t-1} for the W_hh gradient):
// row r of the logical matrix is token r; the loader returns the row of the token `shift` positions earlier in the
// same sequence, zeros at the sequence boundary.
struct ALoadSeqShift {
  const float* A;   // [M][lda]
  int64_t M;
  int lda, col0, bm;
  int shift;        // +1: previous position (forward direction), -1: next position (reverse direction)
  SeqGeom g;
  unsigned magK;    // floor(2^32 / K) + 1: x / K == umulhi(x, magK) for x < K + bm  (make_seq_shift)
  int sstride;      // rows between a token and its predecessor along the sequence, times shift
  DEV float4 load4(int tile, int row, int k4) const {
    const int64_t r = (int64_t)tile * bm + row;
    if (r >= M) return make_float4(0.f, 0.f, 0.f, 0.f);
    // token -> (sequence, position)
    int pos;
    int64_t src;
    if (g.mode == 0) {               // intra: token = q*K + k
      pos = (int)(r % g.K);
      src = r - shift;
    } else {                         // inter: token = (b*S + s)*K + k, position s
      pos = (int)((r / g.K) % g.S);
      src = r - (int64_t)shift * g.K;
    }
    const int p2 = pos - shift;
    if (p2 < 0 || p2 >= g.len) return make_float4(0.f, 0.f, 0.f, 0.f);
    return *reinterpret_cast<const float4*>(A + src * lda + col0 + 4 * k4);
  }
  // The same value without 64-bit divisions and without branches (rows are < 2^31: make_plan): the tile base is
  // divided once per tile (wave-uniform), the row inside the tile by a multiply-high; outside the matrix / the sequence
  // the load goes to row 0 and the result is replaced by zeros.  The weight-gradient kernels issue these between the
  // MFMAs of a tile: with the branchy form above every load sat in its own basic block IN FRONT of the MFMA block and
  // its ~130 instructions of address arithmetic were not overlapped with anything (one wave per SIMD).
  DEV float4 load4z(int tile, int row, int k4) const {
    const unsigned K = (unsigned)g.K, S = (unsigned)g.S;
    const unsigned r0 = (unsigned)tile * (unsigned)bm;
    const unsigned q0 = r0 / K, base = r0 - q0 * K;
    const unsigned x = base + (unsigned)row;
    const unsigned q1 = __umulhi(x, magK);
    const unsigned k = x - q1 * K, rk = q0 + q1;
    const unsigned pos = g.mode == 0 ? k : rk - (rk / S) * S;
    const int p2 = (int)pos - shift;
    const unsigned r = r0 + (unsigned)row;
    const bool ok = r < (unsigned)M && p2 >= 0 && p2 < g.len;
    const unsigned src = ok ? (unsigned)((int)r - sstride) : 0u;
    const float4 v = *reinterpret_cast<const float4*>(A + (size_t)src * (unsigned)lda + (unsigned)(col0 + 4 * k4));
    return mask4(v, ok);
  }
};
static inline ALoadSeqShift make_seq_shift(const float* A, int64_t M, int lda, int col0, int bm, int shift, const SeqGeom& g) {
  ALoadSeqShift l{A, M, lda, col0, bm, shift, g, 0u, 0};
  l.magK = g.K > 1 ? (unsigned)((1ull << 32) / (unsigned)g.K) + 1u : 0u;   // K == 1: x / 1 is not a multiply-high; rejected by the caller
  l.sstride = shift * (g.mode == 0 ? 1 : g.K);
  return l;
}

// dense rows of a column slice [col0, col0 + width) of a token-major matrix.  RELU (ffn[0] applied while loading the raw
// LSTM output the training tape keeps) is a COMPILE-TIME flag: as a runtime member it put a branch behind every load, the
// loads of a tile could no longer be issued as one batch, and the K = 256 GEMMs of the training step ran 1.6x longer
// than the same instantiation in inference (rocprofv3 PMC pass: 552 k vs 341 k active cycles per launch).
template <bool RELU>
struct ALoadColsT {
  const float* A;
  int64_t M;
  int lda, col0, bm;
  DEV float4 load4(int tile, int row, int k4) const {
    const int64_t r0 = (int64_t)tile * bm;
    if (r0 + row >= M) return make_float4(0.f, 0.f, 0.f, 0.f);
    float4 v = *reinterpret_cast<const float4*>(A + r0 * lda + (unsigned)(row * lda + col0 + 4 * k4));
    if constexpr (RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    return v;
  }
  // zero-filling like load4, branch-free like load4c (the load of a row beyond M goes to the last row and is discarded)
  DEV float4 load4z(int tile, int row, int k4) const {
    const int64_t r0 = (int64_t)tile * bm;
    const int last = (int)(M - 1 - r0 < bm - 1 ? M - 1 - r0 : bm - 1);   // wave-uniform
    const bool ok = row <= last;
    float4 v = *reinterpret_cast<const float4*>(A + r0 * lda + (unsigned)((ok ? row : last) * lda + col0 + 4 * k4));
    if constexpr (RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    return mask4(v, ok);
  }
  // GEMM-engine form (rows beyond M are never stored there): clamped, branch-free.  The weight-gradient kernels need
  // the zero-filling load4 / load4z (padded rows must not contribute to the sums).
  DEV float4 load4c(int tile, int row, int k4) const {
    const int64_t r0 = (int64_t)tile * bm;
    const int last = (int)(M - 1 - r0 < bm - 1 ? M - 1 - r0 : bm - 1);
    float4 v = *reinterpret_cast<const float4*>(A + r0 * lda + (unsigned)((row < last ? row : last) * lda + col0 + 4 * k4));
    if constexpr (RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    return v;
  }
};
using ALoadCols = ALoadColsT<false>;
using ALoadColsReLU = ALoadColsT<true>;

template <int NN, int KK>
struct WgradShape {
  static constexpr int LDY = NN + 4, LDX = KK + 4;
  static constexpr int RB = NN / 128;          // 32-row blocks of dW owned by one wave
  static constexpr int CB = KK / 32;           // 32-column blocks (every wave owns all of them)
  static constexpr size_t lds_bytes() { return sizeof(float) * (4 + 32 * (size_t)(LDY + LDX)); }
};

// grid.x workgroups; slab[blockIdx.x][NN][KK] receives the partial sum of this workgroup's tiles.
constexpr int wgrad_gcd(int a, int b) { return b == 0 ? a : wgrad_gcd(b, a % b); }

// COLSUM: the column sums of Y (= the bias gradient of the same layer) ride along: every thread adds up the float4s it
// stages (their column is the same for every tile), the workgroup combines them in a fixed order at the end into
// colslab[blockIdx.x][NN] -- instead of a second pass over Y by colsum_kernel.
template <int NN, int KK, class YLoad, class XLoad, bool COLSUM = false>
__global__ __launch_bounds__(256) void wgrad_kernel(int ntiles, unsigned* __restrict__ queue, YLoad yl, XLoad xl,
                                                     float* __restrict__ slab, float* __restrict__ colslab = nullptr) {
  using Sh = WgradShape<NN, KK>;
  static_assert(NN % 128 == 0 && KK % 32 == 0, "wgrad tile");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int* s_next = reinterpret_cast<int*>(smem);
  float* Ys = smem + 4;
  float* Xs = Ys + 32 * Sh::LDY;
  const int tid = threadIdx.x;
  const int w = tid >> 6, lane = tid & 63, c = lane & 31, hh = lane >> 5;

  f32x16 acc[Sh::RB][Sh::CB];
#pragma unroll
  for (int a = 0; a < Sh::RB; ++a)
#pragma unroll
    for (int b = 0; b < Sh::CB; ++b) acc[a][b] = zero16();

  // software pipeline as in the GEMM engine: the next tile's rows are fetched into registers while the current
  // tile's MFMAs run; tickets double buffered in LDS
  constexpr int Y4 = NN / 4, X4 = KK / 4;
  constexpr int NY = (32 * Y4) / 256, NX = (32 * X4) / 256;
  static_assert((32 * Y4) % 256 == 0 && (32 * X4) % 256 == 0, "staging map");
  float4 py[NY], px[NX];
  // a thread's staging slot i holds column block (i*256 + tid) % Y4: NCS = Y4 / gcd(Y4, 256) distinct ones, visited
  // round robin in i
  constexpr int NCS = COLSUM ? Y4 / wgrad_gcd(Y4, 256) : 1;
  float4 csum[NCS];
#pragma unroll
  for (int k = 0; k < NCS; ++k) csum[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  TileTickets tickets{queue, (int)blockIdx.x, (int)gridDim.x};
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
  int tile = s_next[0];
  if (tile < ntiles) {
#pragma unroll
    for (int i = 0; i < NY; ++i) py[i] = wg_load(yl, tile, (i * 256 + tid) / Y4, (i * 256 + tid) % Y4);
#pragma unroll
    for (int i = 0; i < NX; ++i) px[i] = wg_load(xl, tile, (i * 256 + tid) / X4, (i * 256 + tid) % X4);
  }
  // The 16 MFMA steps of a tile are ONE straight-line block: the fragments of step s+1 are read from LDS into the other
  // register set before the MFMAs of step s, and the next tile's operand rows are fetched a few loads per step
  // BETWEEN the MFMAs (a wave issues in order and is alone on its SIMD: address arithmetic in front of the block is
  // time the matrix pipe idles, behind an MFMA it is free).  On the last tile the fetch re-reads that tile (clamped)
  // instead of branching around the loads.
  constexpr int NL = NY + NX, LPS = (NL + 7) / 8;   // all fetched in the first half of the block: their latency ends inside it
  int par = 0;
  while (tile < ntiles) {
    __syncthreads();                                   // previous tile's fragments fully consumed
#pragma unroll
    for (int i = 0; i < NY; ++i) {
      const int idx = i * 256 + tid;
      *reinterpret_cast<float4*>(&Ys[(idx / Y4) * Sh::LDY + 4 * (idx % Y4)]) = py[i];
      if constexpr (COLSUM) {
        float4& a = csum[i % NCS];
        a.x += py[i].x; a.y += py[i].y; a.z += py[i].z; a.w += py[i].w;
      }
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int idx = i * 256 + tid;
      *reinterpret_cast<float4*>(&Xs[(idx / X4) * Sh::LDX + 4 * (idx % X4)]) = px[i];
    }
    if (tickets.dynamic() && tid == 0) s_next[par ^ 1] = ticket_ahead;   // publish the next ticket (requested one tile ago)
    __syncthreads();
    const int next = tickets.dynamic() ? __builtin_amdgcn_readfirstlane(s_next[par ^ 1]) : tile + tickets.stride;
    const int nf = next < ntiles ? next : ntiles - 1;
    if (tickets.dynamic() && tid == 0) ticket_ahead = tickets.take();
    // D[i = row of dW][j = col of dW] += sum over the tile's tokens; MFMA step s covers tokens 2s (slot 0), 2s+1 (slot 1)
    float a[2][Sh::RB], b[2][Sh::CB];
    auto frag = [&](int s, float* fa, float* fb) {
      const float* yrow = Ys + (2 * s + hh) * Sh::LDY + c;
      const float* xrow = Xs + (2 * s + hh) * Sh::LDX + c;
#pragma unroll
      for (int i = 0; i < Sh::RB; ++i) fa[i] = yrow[(w + 4 * i) * 32];
#pragma unroll
      for (int j = 0; j < Sh::CB; ++j) fb[j] = xrow[j * 32];
    };
    frag(0, a[0], b[0]);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      if (s + 1 < 16) frag(s + 1, a[(s + 1) & 1], b[(s + 1) & 1]);
#pragma unroll
      for (int q = 0; q < LPS; ++q) {
        const int i = s * LPS + q;
        if (i < NY) py[i] = wg_load(yl, nf, (i * 256 + tid) / Y4, (i * 256 + tid) % Y4);
        else if (i < NL) px[i - NY] = wg_load(xl, nf, ((i - NY) * 256 + tid) / X4, ((i - NY) * 256 + tid) % X4);
      }
#pragma unroll
      for (int i = 0; i < Sh::RB; ++i)
#pragma unroll
        for (int j = 0; j < Sh::CB; ++j) acc[i][j] = mfma32(a[s & 1][i], b[s & 1][j], acc[i][j]);
    }
    tile = next;
    par ^= 1;
  }
  // partial tile in FRAGMENT order (wgrad_frag_index): four accumulator registers = one 16-byte store, a wave's store
  // = 1 KiB contiguous.  Row-major order needed 16 x RB x CB dword stores per lane and the store queue, not the MFMA
  // block, ended the kernel; slab_reduce_frag_kernel puts the sums back into row-major order.
  float* out = slab + (size_t)blockIdx.x * (NN * KK);
#pragma unroll
  for (int i = 0; i < Sh::RB; ++i)
#pragma unroll
    for (int j = 0; j < Sh::CB; ++j)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4)
        *reinterpret_cast<float4*>(out + ((((size_t)(w * Sh::RB + i) * Sh::CB + j) * 4 + g4) * 64 + lane) * 4) =
            make_float4(acc[i][j][4 * g4], acc[i][j][4 * g4 + 1], acc[i][j][4 * g4 + 2], acc[i][j][4 * g4 + 3]);
  if constexpr (COLSUM) {
    // thread tid's accumulator k belongs to column block (k*256 + tid) % Y4; owner thread cb sums its contributors in
    // (k, tid) order -- a fixed association order
    __syncthreads();
    float4* red = reinterpret_cast<float4*>(smem);   // [NCS][256] float4 (the tile buffers are free now)
    static_assert(sizeof(float) * (4 + 32 * (size_t)(Sh::LDY + Sh::LDX)) >= sizeof(float4) * NCS * 256, "reduction scratch");
#pragma unroll
    for (int k = 0; k < NCS; ++k) red[k * 256 + tid] = csum[k];
    __syncthreads();
    if (tid < Y4) {
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int k = 0; k < NCS; ++k)
        for (int t2 = ((tid - k * 256) % Y4 + Y4) % Y4; t2 < 256; t2 += Y4) {   // (k*256 + t2) % Y4 == tid
          const float4 u = red[k * 256 + t2];
          a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
        }
      *reinterpret_cast<float4*>(colslab + (size_t)blockIdx.x * NN + 4 * tid) = a;
    }
  }
}

// The same contraction for shapes wgrad_kernel is not built for (NN a multiple of 32 but not of 128: the 64-feature models'
// out-projection 64 x 64, ffn 64 x 256, in-projection 192 x 64): dW[NN][KK] = sum_tokens Y^T X with the (NN/32) x (KK/32)
// output blocks dealt round robin to the four waves, 32-token tiles through LDS, a grid-stride loop instead of tickets and
// the partial tile written ROW-MAJOR (slab_reduce_kernel sums it).  Plain rather than tuned: these shapes are not on a
// BASELINE configuration's training path.
template <int NN, int KK, class YLoad, class XLoad>
__global__ __launch_bounds__(256) void wgrad_generic_kernel(int ntiles, YLoad yl, XLoad xl, float* __restrict__ slab) {
  static_assert(NN % 32 == 0 && KK % 32 == 0, "wgrad tile");
  constexpr int LDY = NN + 4, LDX = KK + 4, RBT = NN / 32, CBT = KK / 32, NBLK = RBT * CBT, NBW = (NBLK + 3) / 4;
  constexpr int Y4 = NN / 4, X4 = KK / 4, NY = (32 * Y4) / 256, NX = (32 * X4) / 256;
  static_assert((32 * Y4) % 256 == 0 && (32 * X4) % 256 == 0, "staging map");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ys = smem;
  float* Xs = Ys + 32 * LDY;
  const int tid = threadIdx.x;
  const int w = tid >> 6, lane = tid & 63, c = lane & 31, hh = lane >> 5;
  f32x16 acc[NBW];
#pragma unroll
  for (int k = 0; k < NBW; ++k) acc[k] = zero16();
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    float4 py[NY], px[NX];
#pragma unroll
    for (int i = 0; i < NY; ++i) py[i] = wg_load(yl, tile, (i * 256 + tid) / Y4, (i * 256 + tid) % Y4);
#pragma unroll
    for (int i = 0; i < NX; ++i) px[i] = wg_load(xl, tile, (i * 256 + tid) / X4, (i * 256 + tid) % X4);
    __syncthreads();                                   // previous tile's fragments fully consumed
#pragma unroll
    for (int i = 0; i < NY; ++i) {
      const int idx = i * 256 + tid;
      *reinterpret_cast<float4*>(&Ys[(idx / Y4) * LDY + 4 * (idx % Y4)]) = py[i];
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int idx = i * 256 + tid;
      *reinterpret_cast<float4*>(&Xs[(idx / X4) * LDX + 4 * (idx % X4)]) = px[i];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NBW; ++k) {
      const int blk = w + 4 * k;                       // wave-uniform
      if (blk < NBLK) {
        const int rb = blk / CBT, cb = blk % CBT;
#pragma unroll
        for (int s = 0; s < 16; ++s)                   // MFMA step s covers tokens 2 s (slot 0) and 2 s + 1 (slot 1)
          acc[k] = mfma32(Ys[(2 * s + hh) * LDY + rb * 32 + c], Xs[(2 * s + hh) * LDX + cb * 32 + c], acc[k]);
      }
    }
  }
  float* out = slab + (size_t)blockIdx.x * (NN * KK);
#pragma unroll
  for (int k = 0; k < NBW; ++k) {
    const int blk = w + 4 * k;
    if (blk < NBLK) {
      const int rb = blk / CBT, cb = blk % CBT;
#pragma unroll
      for (int r = 0; r < 16; ++r) out[(size_t)(rb * 32 + ROW32(r, hh)) * KK + cb * 32 + c] = acc[k][r];
    }
  }
}

// Two weight gradients that share their Y operand in ONE pass over Y:  dWa = sum Y^T Xa,  dWb = sum Y^T Xb  (the LSTM's
// W_ih and W_hh gradients: Y = dP, Xa = the layer input, Xb = h_{t-1}).  The Y tile is staged and its fragments are
// read once for 2 x CB MFMAs each; 2 x RB x CB accumulator tiles per wave (256 registers for <256,128>: they live in
// the AGPR half, one workgroup per CU as before).
// NSL independent problems ("slices": the 2 directions x 2 halves of the 512 gate rows) share ONE launch: workgroup b
// works on slice b % NSL with its own ticket counter, so a slice is summed by gridDim.x / NSL workgroups that each
// take NSL times more tiles than with one launch per slice.  What that buys is the END of the kernel: a workgroup's
// partial tiles are 256 KB, a CU stores ~7-11 B/cycle, and in-kernel stamps put that store tail at 64 k cycles against
// 20 tiles x 16.4 k cycles of MFMAs per launch (14 %); now it is paid once per 83 tiles, and the slabs to reduce are
// NSL times fewer.   slab[slice][gridDim.x / NSL][2][NN x KK in fragment order].
template <class YLoad, class XLoadA, class XLoadB, int NSL>
struct Wgrad2Args {
  YLoad yl[NSL];
  XLoadA xa[NSL];
  XLoadB xb[NSL];
};
template <int NN, int KK, class YLoad, class XLoadA, class XLoadB, int NSL>
__global__ __launch_bounds__(256) void wgrad2_kernel(int ntiles, unsigned* __restrict__ queue_base,
                                                      const Wgrad2Args<YLoad, XLoadA, XLoadB, NSL> args,
                                                      float* __restrict__ slab_base) {
  const int sl = blockIdx.x % NSL;
  const YLoad yl = args.yl[sl];
  const XLoadA xa = args.xa[sl];
  const XLoadB xb = args.xb[sl];
  unsigned* queue = queue_base ? queue_base + sl : nullptr;
  float* slab = slab_base + (size_t)sl * (gridDim.x / NSL) * (2 * NN * KK);
  const int wg = blockIdx.x / NSL;
  using Sh = WgradShape<NN, KK>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int* s_next = reinterpret_cast<int*>(smem);
  float* Ys = smem + 4;
  float* Xas = Ys + 32 * Sh::LDY;
  float* Xbs = Xas + 32 * Sh::LDX;
  const int tid = threadIdx.x;
  const int w = tid >> 6, lane = tid & 63, c = lane & 31, hh = lane >> 5;
  f32x16 acca[Sh::RB][Sh::CB], accb[Sh::RB][Sh::CB];
#pragma unroll
  for (int a = 0; a < Sh::RB; ++a)
#pragma unroll
    for (int b = 0; b < Sh::CB; ++b) {
      acca[a][b] = zero16();
      accb[a][b] = zero16();
    }
  constexpr int Y4 = NN / 4, X4 = KK / 4;
  constexpr int NY = (32 * Y4) / 256, NX = (32 * X4) / 256;
  float4 py[NY], pxa[NX], pxb[NX];
  TileTickets tickets{queue, wg, (int)gridDim.x / NSL};
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
  int tile = s_next[0];
  if (tile < ntiles) {
#pragma unroll
    for (int i = 0; i < NY; ++i) py[i] = wg_load(yl, tile, (i * 256 + tid) / Y4, (i * 256 + tid) % Y4);
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      pxa[i] = wg_load(xa, tile, (i * 256 + tid) / X4, (i * 256 + tid) % X4);
      pxb[i] = wg_load(xb, tile, (i * 256 + tid) / X4, (i * 256 + tid) % X4);
    }
  }
  // straight-line tile block as in wgrad_kernel: fragments double buffered, next tile's rows fetched between the MFMAs
  constexpr int NL = NY + 2 * NX, LPS = (NL + 7) / 8;
  int par = 0;
  while (tile < ntiles) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NY; ++i) {
      const int idx = i * 256 + tid;
      *reinterpret_cast<float4*>(&Ys[(idx / Y4) * Sh::LDY + 4 * (idx % Y4)]) = py[i];
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int idx = i * 256 + tid;
      *reinterpret_cast<float4*>(&Xas[(idx / X4) * Sh::LDX + 4 * (idx % X4)]) = pxa[i];
      *reinterpret_cast<float4*>(&Xbs[(idx / X4) * Sh::LDX + 4 * (idx % X4)]) = pxb[i];
    }
    if (tickets.dynamic() && tid == 0) s_next[par ^ 1] = ticket_ahead;
    __syncthreads();
    const int next = tickets.dynamic() ? __builtin_amdgcn_readfirstlane(s_next[par ^ 1]) : tile + tickets.stride;
    const int nf = next < ntiles ? next : ntiles - 1;
    if (tickets.dynamic() && tid == 0) ticket_ahead = tickets.take();
    float a[2][Sh::RB], ba[2][Sh::CB], bb[2][Sh::CB];
    auto frag = [&](int s, float* fa, float* fba, float* fbb) {
      const float* yrow = Ys + (2 * s + hh) * Sh::LDY + c;
      const float* xarow = Xas + (2 * s + hh) * Sh::LDX + c;
      const float* xbrow = Xbs + (2 * s + hh) * Sh::LDX + c;
#pragma unroll
      for (int i = 0; i < Sh::RB; ++i) fa[i] = yrow[(w + 4 * i) * 32];
#pragma unroll
      for (int j = 0; j < Sh::CB; ++j) {
        fba[j] = xarow[j * 32];
        fbb[j] = xbrow[j * 32];
      }
    };
    frag(0, a[0], ba[0], bb[0]);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      if (s + 1 < 16) frag(s + 1, a[(s + 1) & 1], ba[(s + 1) & 1], bb[(s + 1) & 1]);
#pragma unroll
      for (int q = 0; q < LPS; ++q) {
        const int i = s * LPS + q;
        if (i < NY) py[i] = wg_load(yl, nf, (i * 256 + tid) / Y4, (i * 256 + tid) % Y4);
        else if (i < NY + NX) pxa[i - NY] = wg_load(xa, nf, ((i - NY) * 256 + tid) / X4, ((i - NY) * 256 + tid) % X4);
        else if (i < NL) pxb[i - NY - NX] = wg_load(xb, nf, ((i - NY - NX) * 256 + tid) / X4, ((i - NY - NX) * 256 + tid) % X4);
      }
#pragma unroll
      for (int i = 0; i < Sh::RB; ++i)
#pragma unroll
        for (int j = 0; j < Sh::CB; ++j) {
          acca[i][j] = mfma32(a[s & 1][i], ba[s & 1][j], acca[i][j]);
          accb[i][j] = mfma32(a[s & 1][i], bb[s & 1][j], accb[i][j]);
        }
    }
    tile = next;
    par ^= 1;
  }
  float* outa = slab + (size_t)wg * (2 * NN * KK);
  float* outb = outa + NN * KK;
#pragma unroll
  for (int i = 0; i < Sh::RB; ++i)
#pragma unroll
    for (int j = 0; j < Sh::CB; ++j)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {   // fragment order, as wgrad_kernel
        const size_t o = ((((size_t)(w * Sh::RB + i) * Sh::CB + j) * 4 + g4) * 64 + lane) * 4;
        *reinterpret_cast<float4*>(outa + o) =
            make_float4(acca[i][j][4 * g4], acca[i][j][4 * g4 + 1], acca[i][j][4 * g4 + 2], acca[i][j][4 * g4 + 3]);
        *reinterpret_cast<float4*>(outb + o) =
            make_float4(accb[i][j][4 * g4], accb[i][j][4 * g4 + 1], accb[i][j][4 * g4 + 2], accb[i][j][4 * g4 + 3]);
      }
}

// column sums over the rows of Y[M][ld] (columns [col0, col0+C)): slab[blockIdx.x][C]
template <int C>
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ Y, int64_t M, int ld, int col0,
                                                      float* __restrict__ slab) {
  constexpr int C4 = C / 4, RPB = (256 / C4) > 0 ? (256 / C4) : 1, ACTIVE = C4 * RPB;
  static_assert(C4 <= 256, "colsum width");
  __shared__ float4 red[256];
  const int tid = threadIdx.x;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (tid < ACTIVE) {
    const int c4 = tid % C4;
    for (int64_t r = (int64_t)blockIdx.x * RPB + tid / C4; r < M; r += (int64_t)gridDim.x * RPB) {
      const float4 v = *reinterpret_cast<const float4*>(Y + r * ld + col0 + 4 * c4);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  red[tid] = s;
  __syncthreads();
  if (tid < C4) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = tid; k < ACTIVE; k += C4) { const float4 u = red[k]; a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w; }
    *reinterpret_cast<float4*>(slab + (size_t)blockIdx.x * C + 4 * tid) = a;
  }
}

// out[i] (+)= sum_s slab[s][i] in a FIXED association order: a block owns 32 consecutive elements,
// its 8 slab-lanes each sum the slabs s = lane, lane+8, ... with four loads in flight, then the lanes are combined in
// lane order.  (A single thread walking all slabs serially was latency-bound: ~10 % of the training step.)
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slab, int nslabs, int64_t count,
                                                           float* __restrict__ out, int accumulate) {
  __shared__ float red[8][32];
  const int e = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int64_t i = (int64_t)blockIdx.x * 32 + e;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < count) {
    int k = sl;
    for (; k + 24 < nslabs; k += 32) {
      s0 += slab[(size_t)k * count + i];
      s1 += slab[(size_t)(k + 8) * count + i];
      s2 += slab[(size_t)(k + 16) * count + i];
      s3 += slab[(size_t)(k + 24) * count + i];
    }
    for (; k < nslabs; k += 8) s0 += slab[(size_t)k * count + i];
  }
  red[sl][e] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (sl == 0 && i < count) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += red[j][e];
    out[i] = accumulate ? out[i] + s : s;
  }
}

// LayerNorm backward from the rows the training forward left on the tape (EpiBiasResLNSave: zn = (z - mu) rstd, rstd):
//   dz = rstd * (g - mean(g) - zn * mean(g * zn)),  g = dout * gamma;   d gamma += dout * zn,  d beta += dout
// (dptn.py:47,51).  Bandwidth-bound: 1.5 kB per token for N = 128; a row = N/4 adjacent lanes, four rows in flight per
// thread.  Column sums per thread -> per workgroup through LDS -> partials[gridDim.x][2 N] (d gamma | d beta), reduced in
// a fixed order by slab_reduce_to2_kernel, like the partials of EpiLNBackward.
template <int N>
__global__ __launch_bounds__(256) void ln_backward_kernel(const float* __restrict__ dout, const float* __restrict__ zn,
                                                           const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                           float* __restrict__ dz, float* __restrict__ partials, int64_t M) {
  constexpr int GROUP = N / 4, RPB = 256 / GROUP, U = 4;
  __shared__ float4 red[2][256];
  const int tid = threadIdx.x, rl = tid / GROUP, c4 = tid % GROUP;
  const float4 ga = *reinterpret_cast<const float4*>(gamma + 4 * c4);
  float4 sg = make_float4(0.f, 0.f, 0.f, 0.f), sb = sg;
  const int64_t npass = (M + RPB - 1) / RPB;
  for (int64_t p0 = blockIdx.x; p0 < npass; p0 += (int64_t)U * gridDim.x) {
    float4 d[U], z[U];
    float rs[U];
    bool ok[U];
    int64_t row[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t r = (p0 + (int64_t)u * gridDim.x) * RPB + rl;
      ok[u] = r < M;
      row[u] = ok[u] ? r : M - 1;
      d[u] = mask4(*reinterpret_cast<const float4*>(dout + row[u] * N + 4 * c4), ok[u]);
      z[u] = *reinterpret_cast<const float4*>(zn + row[u] * N + 4 * c4);
      rs[u] = rstd[row[u]];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      sg.x += d[u].x * z[u].x; sg.y += d[u].y * z[u].y; sg.z += d[u].z * z[u].z; sg.w += d[u].w * z[u].w;
      sb.x += d[u].x; sb.y += d[u].y; sb.z += d[u].z; sb.w += d[u].w;
      const float4 g = make_float4(d[u].x * ga.x, d[u].y * ga.y, d[u].z * ga.z, d[u].w * ga.w);
      const float m1 = group_sum<GROUP>((g.x + g.y) + (g.z + g.w)) * (1.0f / N);
      const float m2 = group_sum<GROUP>((g.x * z[u].x + g.y * z[u].y) + (g.z * z[u].z + g.w * z[u].w)) * (1.0f / N);
      if (ok[u]) {
        float4 o;
        o.x = rs[u] * (g.x - m1 - z[u].x * m2);
        o.y = rs[u] * (g.y - m1 - z[u].y * m2);
        o.z = rs[u] * (g.z - m1 - z[u].z * m2);
        o.w = rs[u] * (g.w - m1 - z[u].w * m2);
        *reinterpret_cast<float4*>(dz + row[u] * N + 4 * c4) = o;
      }
    }
  }
  red[0][tid] = sg;
  red[1][tid] = sb;
  __syncthreads();
  if (tid < GROUP) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    for (int k = tid; k < 256; k += GROUP) {
      const float4 u = red[0][k], w = red[1][k];
      a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
      b.x += w.x; b.y += w.y; b.z += w.z; b.w += w.w;
    }
    float* pp = partials + (size_t)blockIdx.x * (2 * N);
    *reinterpret_cast<float4*>(pp + 4 * tid) = a;
    *reinterpret_cast<float4*>(pp + N + 4 * tid) = b;
  }
}

// slab_reduce_kernel with two destinations (overwrite): split >= 0: elements [0, split) go to out_a, the rest to out_b
// (LayerNorm: d gamma | d beta from one slab row); split < 0: every element goes to both (b_ih and b_hh have the same
// gradient).  Same association order as slab_reduce_kernel.
__global__ __launch_bounds__(256) void slab_reduce_to2_kernel(const float* __restrict__ slab, int nslabs, int64_t count,
                                                               float* __restrict__ out_a, float* __restrict__ out_b, int split) {
  __shared__ float red[8][32];
  const int e = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int64_t i = (int64_t)blockIdx.x * 32 + e;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < count) {
    int k = sl;
    for (; k + 24 < nslabs; k += 32) {
      s0 += slab[(size_t)k * count + i];
      s1 += slab[(size_t)(k + 8) * count + i];
      s2 += slab[(size_t)(k + 16) * count + i];
      s3 += slab[(size_t)(k + 24) * count + i];
    }
    for (; k < nslabs; k += 8) s0 += slab[(size_t)k * count + i];
  }
  red[sl][e] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (sl == 0 && i < count) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += red[j][e];
    if (split < 0) {
      out_a[i] = s;
      out_b[i] = s;
    } else if (i < split) {
      out_a[i] = s;
    } else {
      out_b[i - split] = s;
    }
  }
}

// Sum of the weight-gradient kernels' partial tiles, which are stored in fragment order: float4 number
//   f = (((w * RB + i) * CB + j) * 4 + g4) * 64 + lane      holds rows (w + 4 i) * 32 + 8 g4 + 4 (lane / 32) + 0..3
//                                                            of column j * 32 + lane % 32.
// A workgroup owns 32 consecutive float4s; its 8 slab-lanes sum the slabs s = lane, lane + 8, ... with four 16-byte
// loads in flight (fixed association order), then thread e < 32 combines the lanes in lane order and writes its four
// rows of dW[NN][KK].  slabs are `stride` floats apart.  Several sums per launch (blockIdx.y, at most 8).
struct FragOuts {
  float* out[8];
};
// blockIdx.y = which sum: its slabs start at slab + (y / per_group) * group_stride + (y % per_group) * RB*CB*4096
// Workgroups beyond the RB*CB*32 of the tiles (blockIdx.y == 0 only) sum the column-sum rows colslab[nslabs][ncol] of the
// same launch (the bias gradient) the slab_reduce_kernel way.
template <int RB, int CB>
__global__ __launch_bounds__(256) void slab_reduce_frag_kernel(const float* __restrict__ slab, int nslabs, int64_t stride,
                                                                FragOuts outs, int64_t group_stride, int per_group,
                                                                const float* __restrict__ colslab, int ncol,
                                                                float* __restrict__ bias_out) {
  constexpr int KK = CB * 32;
  __shared__ float4 red[8][32];
  const int e = threadIdx.x & 31, sl = threadIdx.x >> 5;
  if (blockIdx.x >= RB * CB * 32) {                  // wave-uniform: bias part
    if (blockIdx.y != 0) return;
    float* redf = reinterpret_cast<float*>(red);
    const int i = (blockIdx.x - RB * CB * 32) * 32 + e;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < ncol) {
      int k = sl;
      for (; k + 24 < nslabs; k += 32) {
        s0 += colslab[(size_t)k * ncol + i];
        s1 += colslab[(size_t)(k + 8) * ncol + i];
        s2 += colslab[(size_t)(k + 16) * ncol + i];
        s3 += colslab[(size_t)(k + 24) * ncol + i];
      }
      for (; k < nslabs; k += 8) s0 += colslab[(size_t)k * ncol + i];
    }
    redf[sl * 32 + e] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sl == 0 && i < ncol) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) s += redf[j * 32 + e];
      bias_out[i] = s;
    }
    return;
  }
  const int f = blockIdx.x * 32 + e;
  const int y = blockIdx.y;
  float* __restrict__ out = outs.out[y];
  const float4* src = reinterpret_cast<const float4*>(slab + (y / per_group) * group_stride + (int64_t)(y % per_group) * (RB * CB * 4096)) + f;
  const int64_t st4 = stride / 4;
  auto add = [](float4& a, const float4 b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; };
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
  int k = sl;
  for (; k + 24 < nslabs; k += 32) {
    add(s0, src[(int64_t)k * st4]);
    add(s1, src[(int64_t)(k + 8) * st4]);
    add(s2, src[(int64_t)(k + 16) * st4]);
    add(s3, src[(int64_t)(k + 24) * st4]);
  }
  for (; k < nslabs; k += 8) add(s0, src[(int64_t)k * st4]);
  add(s0, s1);
  add(s2, s3);
  add(s0, s2);
  red[sl][e] = s0;
  __syncthreads();
  if (sl == 0) {
    float4 s = red[0][e];
#pragma unroll
    for (int j = 1; j < 8; ++j) add(s, red[j][e]);
    const int lane = f & 63, g4 = (f >> 6) & 3, t = f >> 8;          // t = (w * RB + i) * CB + j
    const int j = t % CB, wi = t / CB, i = wi % RB, w = wi / RB;
    const int row = (w + 4 * i) * 32 + 8 * g4 + 4 * (lane >> 5), col = j * 32 + (lane & 31);
    out[(size_t)row * KK + col] = s.x;
    out[(size_t)(row + 1) * KK + col] = s.y;
    out[(size_t)(row + 2) * KK + col] = s.z;
    out[(size_t)(row + 3) * KK + col] = s.w;
  }
}

// (LSTM backward through time: lstm_bptt.hip)

// ------------------------------------------------------------------------------------------------
// attention backward (per sequence and head, everything on chip: len <= 256)
// ------------------------------------------------------------------------------------------------
//   P = softmax(scale * Q K^T);  O = P V  (forward, attention.h).  Given dO:
//     delta[q] = <dO[q], O[q]>;  dP = dO V^T;  dS = P * (dP - delta);  dQ = scale dS K;  dK = scale dS^T Q;  dV = P^T dO
// Phase A (wave = query block, score tiles transposed exactly as in the forward): softmax statistics, dQ.
// Phase B (wave = key block): recompute the score tiles non-transposed (keys on lanes) so that the reductions over
// queries (dK, dV) run over MFMA k-slots; per-query m, 1/l and delta come from phase A through LDS.
template <int DH>
struct AttnBwdShape {
  static constexpr int LD = DH + 4;
  // two row arrays (K,V in phase A; Q,dO in phase B) + one float4 of per-query statistics
  static constexpr size_t lds_bytes(int nkb) { return sizeof(float) * ((size_t)nkb * 32 * (2 * LD + 4)); }
};

// PHASE 0 and PHASE 1 are separate launches (each gets its own register allocation: together they needed 170 VGPRs,
// two short of three waves per SIMD); the per-query statistics travel through `stats` [token][head][4].
// v where the lane's bit of the 64-bit lane mask m (a wave-uniform value: SGPR pair) is set, else 0 -- one instruction
DEV float keep_by_lane_mask(unsigned long long m, float v) {
  float o;
  asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(o) : "v"(v), "s"(m));
  return o;
}

template <int DH, int NKB, int PHASE>
__global__ __launch_bounds__(64 * NKB, (NKB >= 4 && NKB <= 6) ? 3 : 2) void attention_bwd_kernel(const float* __restrict__ qkv,
                                                                  const float* __restrict__ att,
                                                                  const float* __restrict__ datt,
                                                                  float* __restrict__ dqkv, float* __restrict__ stats,
                                                                  int heads, int N, SeqGeom g, float scale, DropCfg drop,
                                                                  const float2* __restrict__ fstats,
                                                                  const unsigned long long* __restrict__ amask) {
  using Sh = AttnBwdShape<DH>;
  constexpr int LD = Sh::LD, ROWS = NKB * 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // the two row arrays hold K,V during phase A and are re-staged with Q,dO for phase B (47 KiB instead of 94 KiB
  // per workgroup at 160 rows -> three workgroups per CU instead of one)
  float* Ks = smem;
  float* Vs = Ks + ROWS * LD;
  float* Qs = Ks;                      // phase B alias
  float* Ds = Vs;                      // phase B alias (dO)
  float4* St = reinterpret_cast<float4*>(Vs + ROWS * LD);   // per query: row max (log2 domain), 1 / row sum, delta, dropout seed
  const int tid = threadIdx.x;
  const int wv = tid >> 6, lane = tid & 63, c = lane & 31, hh = lane >> 5;
  const int seq = blockIdx.x, head = blockIdx.y;
  const int len = g.len;
  const int64_t tok0 = seq_token_base(g, seq);
  const int tstride = seq_token_stride(g);
  const int ld3 = 3 * N;
  const float sl2e = scale * 1.4426950408889634f;

  constexpr int R4 = DH / 4;
  if constexpr (PHASE == 0) {
  // ---- every global load of the workgroup is requested up front: this lane's q / dO / O row fragments and the K, V
  //      rows to stage (rows >= len are zero).  (First version: a staging loop that waited for each pair of loads
  //      before its LDS store, then the barrier, then the row fragments -- five exposed round trips per workgroup.)
  const int qb = wv;
  const int p = qb * 32 + c;
  float4 q4r[DH / 8], d4r[DH / 8], o4r[DH / 8];
  {
    const int64_t tok = tok0 + (int64_t)(p < len ? p : 0) * tstride;
    const float* qrow = qkv + tok * ld3 + head * DH + 4 * hh;
    const float* drow = datt + tok * N + head * DH + 4 * hh;
    const float* orow = att + tok * N + head * DH + 4 * hh;
#pragma unroll
    for (int m = 0; m < DH / 8; ++m) {
      q4r[m] = *reinterpret_cast<const float4*>(qrow + 8 * m);
      d4r[m] = *reinterpret_cast<const float4*>(drow + 8 * m);
      o4r[m] = *reinterpret_cast<const float4*>(orow + 8 * m);
    }
  }
  constexpr int NST = (ROWS * R4) / (64 * NKB);
  static_assert((ROWS * R4) % (64 * NKB) == 0, "staging map");
  float4 kst[NST], vst[NST];
#pragma unroll
  for (int i = 0; i < NST; ++i) {
    const int idx = i * (64 * NKB) + tid;
    const int pr = idx / R4, f = idx % R4;
    const float* row = qkv + (tok0 + (int64_t)(pr < len ? pr : len - 1) * tstride) * ld3 + head * DH + 4 * f;
    kst[i] = mask4(*reinterpret_cast<const float4*>(row + N), pr < len);
    vst[i] = mask4(*reinterpret_cast<const float4*>(row + 2 * N), pr < len);
  }
  // softmax statistics of this lane's query, from the training forward's tape (attention.h): one key block at a time
  // is enough then -- S^T tile, dP^T tile, dS, dQ -- instead of the whole score row (80 registers less)
  float2 ms = fstats[(tok0 + (int64_t)(p < len ? p : 0) * tstride) * heads + head];
  if (p >= len) ms = make_float2(0.f, 0.f);
#pragma unroll
  for (int i = 0; i < NST; ++i) {
    const int idx = i * (64 * NKB) + tid;
    const int pr = idx / R4, f = idx % R4;
    *reinterpret_cast<float4*>(&Ks[pr * LD + 4 * f]) = kst[i];
    *reinterpret_cast<float4*>(&Vs[pr * LD + 4 * f]) = vst[i];
  }
  __syncthreads();

  // =================================== phase A: wave = query block ===================================
  {
    float qf[DH / 2], df[DH / 2];
    float dsum = 0.f;
#pragma unroll
    for (int m = 0; m < DH / 8; ++m) {
      float4 q4 = q4r[m], d4 = d4r[m], o4 = o4r[m];
      if (p >= len) q4 = d4 = o4 = make_float4(0.f, 0.f, 0.f, 0.f);
      qf[4 * m + 0] = q4.x * sl2e; qf[4 * m + 1] = q4.y * sl2e; qf[4 * m + 2] = q4.z * sl2e; qf[4 * m + 3] = q4.w * sl2e;
      df[4 * m + 0] = d4.x; df[4 * m + 1] = d4.y; df[4 * m + 2] = d4.z; df[4 * m + 3] = d4.w;
      dsum += d4.x * o4.x + d4.y * o4.y + d4.z * o4.z + d4.w * o4.w;
    }
    const float delta = dsum + __shfl_xor(dsum, 32);
    const float mx = ms.x, inv = ms.y;
    if (hh == 0 && p < len)
      *reinterpret_cast<float4*>(stats + ((tok0 + (int64_t)p * tstride) * heads + head) * 4) = make_float4(mx, inv, delta, 0.f);
    // the forward's keep decisions of this query block (attention.h): word (rb, r) is the lane mask of register (rb, r) in exactly
    // this layout (lane = query, register = key), read through the scalar unit -- the address is wave-uniform
    const unsigned long long* mw = amask + ((((int64_t)seq * heads + head) * NKB + __builtin_amdgcn_readfirstlane(qb)) * NKB) * 16;
    f32x16 dq = zero16();
#pragma unroll
    for (int rb = 0; rb < NKB; ++rb) {
      // this key block's 16 mask words: requested here (two s_load_dwordx16), they arrive behind the 32 MFMAs below
      unsigned long long mk[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) mk[r] = mw[rb * 16 + r];
      f32x16 sc = zero16(), dp = zero16();
      const float* krow = Ks + (rb * 32 + c) * LD + 4 * hh;
      const float* vrow = Vs + (rb * 32 + c) * LD + 4 * hh;
#pragma unroll
      for (int m = 0; m < DH / 8; ++m) {
        const float4 k = *reinterpret_cast<const float4*>(krow + 8 * m);
        const float4 v = *reinterpret_cast<const float4*>(vrow + 8 * m);
        sc = mfma32(k.x, qf[4 * m + 0], sc);
        dp = mfma32(v.x, df[4 * m + 0], dp);
        sc = mfma32(k.y, qf[4 * m + 1], sc);
        dp = mfma32(v.y, df[4 * m + 1], dp);
        sc = mfma32(k.z, qf[4 * m + 2], sc);
        dp = mfma32(v.z, df[4 * m + 2], dp);
        sc = mfma32(k.w, qf[4 * m + 3], sc);
        dp = mfma32(v.w, df[4 * m + 3], dp);
      }
#pragma unroll
      for (int r0 = 0; r0 < 16; r0 += 8) {      // K rows fetched 8 at a time
        float kk[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const float kval = Ks[(rb * 32 + ROW32(r0 + r, hh)) * LD + (c < DH ? c : 0)];
          kk[r] = c < DH ? kval : 0.f;
        }
        // ONE wait for the batch (and the mask words), the dQ MFMAs kept behind it: left alone the compiler sinks every K-row read to
        // the MFMA that uses it and waits for it there -- 64 `s_waitcnt lgkmcnt(0)` per wave, each an exposed LDS round trip in
        // front of a dependent MFMA (attention.h's forward has had the same fence since round 1).  Round 5: this, not the hash's
        // instruction count, is what the kernel's time hangs on (halving its vector instructions moved nothing).
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const int key = rb * 32 + ROW32(r0 + r, hh);
          // (straight-line: padded keys by select; thresh = 0 / inv_keep = 1 without dropout)
          float pr = fast_exp2(sc[r0 + r] - mx) * inv;
          if (rb == NKB - 1) pr = key < len ? pr : 0.f;
          const float dpv = keep_by_lane_mask(mk[r0 + r], dp[r0 + r] * drop.inv_keep);
          dq = mfma32(pr * (dpv - delta), kk[r], dq);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int pq = qb * 32 + ROW32(r, hh);
      if (pq < len && c < DH) dqkv[(tok0 + (int64_t)pq * tstride) * ld3 + head * DH + c] = dq[r] * scale;
    }
  }
  } else {
  // =================================== phase B: wave = key block =====================================
  {
    const int kb = wv;
    const int key = kb * 32 + c;
    const bool key_ok = key < len;
    float kf[DH / 2], vf[DH / 2];
    {
      const float* krow = qkv + (tok0 + (int64_t)(key_ok ? key : 0) * tstride) * ld3 + N + head * DH + 4 * hh;
#pragma unroll
      for (int m = 0; m < DH / 8; ++m) {
        float4 k4 = *reinterpret_cast<const float4*>(krow + 8 * m);
        float4 v4 = *reinterpret_cast<const float4*>(krow + N + 8 * m);
        if (!key_ok) k4 = v4 = make_float4(0.f, 0.f, 0.f, 0.f);
        kf[4 * m + 0] = k4.x; kf[4 * m + 1] = k4.y; kf[4 * m + 2] = k4.z; kf[4 * m + 3] = k4.w;
        vf[4 * m + 0] = v4.x; vf[4 * m + 1] = v4.y; vf[4 * m + 2] = v4.z; vf[4 * m + 3] = v4.w;
      }
    }
    // stage Q and dO rows and the per-query statistics of phase A: all loads requested before the first LDS store
    constexpr int NST = (ROWS * R4) / (64 * NKB);
    static_assert((ROWS * R4) % (64 * NKB) == 0 && ROWS <= 64 * NKB, "staging map");
    float4 qst[NST], dst_[NST];
#pragma unroll
    for (int i = 0; i < NST; ++i) {
      const int idx = i * (64 * NKB) + tid;
      const int p = idx / R4, f = idx % R4;
      const int64_t tok = tok0 + (int64_t)(p < len ? p : len - 1) * tstride;
      qst[i] = mask4(*reinterpret_cast<const float4*>(qkv + tok * ld3 + head * DH + 4 * f), p < len);
      dst_[i] = mask4(*reinterpret_cast<const float4*>(datt + tok * N + head * DH + 4 * f), p < len);
    }
    {
      const int p = tid < ROWS ? tid : ROWS - 1;
      const float4 st4 = mask4(*reinterpret_cast<const float4*>(stats + ((tok0 + (int64_t)(p < len ? p : len - 1) * tstride) * heads + head) * 4),
                               p < len);
      if (tid < ROWS) St[p] = st4;
    }
#pragma unroll
    for (int i = 0; i < NST; ++i) {
      const int idx = i * (64 * NKB) + tid;
      const int p = idx / R4, f = idx % R4;
      *reinterpret_cast<float4*>(&Qs[p * LD + 4 * f]) = qst[i];
      *reinterpret_cast<float4*>(&Ds[p * LD + 4 * f]) = dst_[i];
    }
    __syncthreads();
    f32x16 dk = zero16(), dv = zero16();
#pragma unroll 1
    for (int qb = 0; qb < NKB; ++qb) {
      f32x16 s2 = zero16(), dp2 = zero16();
      const float* qrow = Qs + (qb * 32 + c) * LD + 4 * hh;
      const float* drow = Ds + (qb * 32 + c) * LD + 4 * hh;
#pragma unroll
      for (int m = 0; m < DH / 8; ++m) {
        const float4 q4 = *reinterpret_cast<const float4*>(qrow + 8 * m);
        const float4 d4 = *reinterpret_cast<const float4*>(drow + 8 * m);
        s2 = mfma32(q4.x, kf[4 * m + 0], s2);
        s2 = mfma32(q4.y, kf[4 * m + 1], s2);
        s2 = mfma32(q4.z, kf[4 * m + 2], s2);
        s2 = mfma32(q4.w, kf[4 * m + 3], s2);
        dp2 = mfma32(d4.x, vf[4 * m + 0], dp2);
        dp2 = mfma32(d4.y, vf[4 * m + 1], dp2);
        dp2 = mfma32(d4.z, vf[4 * m + 2], dp2);
        dp2 = mfma32(d4.w, vf[4 * m + 3], dp2);
      }
      // keep decisions of (query block qb, key block kb), transposed use: this lane's key c sits in register
      // r' = (c & 3) + 4 (c >> 3) of the lanes with hh' = (c >> 2) & 1 there, its query ROW32(r, hh) is lane ROW32(r, hh) + 32 hh'
      // -> one 8-byte word per lane and block pair, then bit ROW32(r, 0) of the half selected by hh', shifted by 4 hh
      uint32_t kw;
      {
        const uint2 w2 = *reinterpret_cast<const uint2*>(amask + ((((int64_t)seq * heads + head) * NKB + qb) * NKB + kb) * 16 +
                                                         ((c & 3) + 4 * (c >> 3)));
        kw = (((c >> 2) & 1) ? w2.y : w2.x) >> (4 * hh);
      }
      const uint32_t ikb = __float_as_uint(drop.inv_keep);
      float qq[16], dd[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qrow_i = qb * 32 + ROW32(r, hh);
        const float4 st4 = St[qrow_i];
        const float pm = st4.x, pl = st4.y, pe = st4.z;
        // straight-line on purpose (selects, no branches): with a branch per element the compiler serialised the sixteen
        // LDS round trips of a tile.  Without dropout thresh = 0 and inv_keep = 1: every element is "kept".
        const float pexp = fast_exp2(s2[r] * sl2e - pm) * pl;
        const float p2 = key_ok ? pexp : 0.f;
        // (bit -> all ones / zero -> inv_keep / 0.0f: v_bfe_i32 + v_and_b32)
        const float keep = __uint_as_float((uint32_t)(((int)(kw << (31 - ROW32(r, 0)))) >> 31) & ikb);
        s2[r] = p2 * keep;                          // dropped P (feeds dV)
        dp2[r] = p2 * (dp2[r] * keep - pe);         // dS
        const float qv = Qs[qrow_i * LD + (c < DH ? c : 0)], dv_ = Ds[qrow_i * LD + (c < DH ? c : 0)];
        qq[r] = c < DH ? qv : 0.f;
        dd[r] = c < DH ? dv_ : 0.f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        dv = mfma32(s2[r], dd[r], dv);
        dk = mfma32(dp2[r], qq[r], dk);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int pk = kb * 32 + ROW32(r, hh);
      if (pk < len && c < DH) {
        float* row = dqkv + (tok0 + (int64_t)pk * tstride) * ld3 + head * DH + c;
        row[N] = dk[r] * scale;
        row[2 * N] = dv[r];
      }
    }
  }
  }
}

// test helper: materialise the dropout keep-mask exactly as the attention kernels regenerate it
//   mask[seq][head][query][key] in {0,1}
__global__ void dropout_mask_kernel(float* __restrict__ mask, SeqGeom g, int heads, DropCfg drop) {
  const int seq = blockIdx.x, head = blockIdx.y;
  const int64_t tok0 = seq_token_base(g, seq);
  const int tstride = seq_token_stride(g);
  for (int idx = threadIdx.x; idx < g.len * g.len; idx += blockDim.x) {
    const int q = idx / g.len, k = idx - q * g.len;
    const uint32_t qh = (uint32_t)(tok0 + (int64_t)q * tstride) * (uint32_t)heads + (uint32_t)head;
    mask[(((int64_t)seq * heads + head) * g.len + q) * g.len + k] =
        drop.thresh == 0u || drop_rand(drop.seed, qh, (uint32_t)k) >= drop.thresh ? 1.f : 0.f;
  }
}
