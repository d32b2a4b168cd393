// lstm4.hip -- LSTM recurrence on 4-sequence tiles for gfx950: the LOW-LATENCY form of K5, for launches so small that
// 16-sequence tiles leave most of the chip idle (bs = 1 of the reference's profiler.py protocol: 141 / 150 sequences =
// 18 / 20 workgroups of lstm16.hip on 256 CUs, and the 1 746 serial steps of a forward ARE its latency).
//
// Reference semantics: torch.nn.LSTM(batch_first=True) cell, gate order i, f, g, o (src/model/dptn.py:22-30,
// src/model/dprnn.py:24-47): one workgroup = one direction x 4 sequences x all positions.
//
// v_mfma_f32_4x4x1_16B_f32 computes 16 independent 4 x 4 outer products (64 FLOP-equivalent lanes at the same
// MAC rate as the 16x16x4 form: 256 MACs in 2 passes), so a step of 4 sequences costs a quarter of the MFMA
// cycles of a step of 16 (256 MFMAs x 8 cycles per wave instead of 256 x 32) -- the step shrinks from ~10 k to
// ~3 k cycles and four times as many workgroups spread over the idle CUs.
//   operand map, lane l: blk = l >> 2, x = l & 3
//     A (h_{t-1}):  lane (i = x, blk) supplies A_blk[i];  with CBSZ = 4 / ABID = b the 4 lanes of block b supply the
//                   A rows of ALL 16 blocks -- so ONE register holds h[seq i][k = 16m + blk] and the 16 MFMAs of a
//                   k-chunk m walk ABID = 0..15: the whole 4 x 128 operand is 8 registers (two ds_read_b128);
//     B (W_hh):     lane (j = x, blk) supplies B_blk[k][j] = W[k][column 4 blk + j = l];
//     D:            reg r of lane l = D[seq r][column l].
//   A wave owns hidden units [32w, 32w+32) as two column groups of 64: group 0 = gates (i | f), group 1 = gates
//   (g | o), lanes 0..31 the first gate of the pair, lanes 32..63 the second, unit 32w + (l & 31).  After the products
//   four v_permlane32_swap bring i, f, g, o of a unit into one lane: lanes 0..31 update sequences 0, 1 and lanes
//   32..63 sequences 2, 3 of their unit (one packed lstm_cell2 per lane and step).
// Everything else follows lstm16.hip: W_hh resident in the AGPRs (256 fragments, pre-scaled by the gate's exp2
// scale), h_t exchanged through a double-buffered LDS tile with one barrier per step, ReLU-or-not as a template flag.
// Input is the PRE16 layout of lstm16.h as it stands (a lane's float4 there = the 4 sequences of one 4-tile for one
// gate column), fetched straight into registers PF steps ahead -- no second producer epilogue.  Inference only.
#include <hip/hip_runtime.h>

#include "lstm16.h"

typedef float f32x4v __attribute__((ext_vector_type(4)));

DEV float relu1_l4(float x) { return __int_as_float(max(__float_as_int(x), 0)); }

template <int ABID>
DEV f32x4v mfma4(float a, float b, f32x4v c) {
  return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, ABID, 0);
}

constexpr int L4_PF = 4;                         // steps of pre-activations in flight (registers)
constexpr int L4_HS_FLOATS = 2 * 4 * L16_H;      // h double buffer, [buf][seq][blk 16][m 8]
constexpr size_t L4_LDS_BYTES = sizeof(float) * L4_HS_FLOATS;

// NCH accumulator chains per column group (a chain's MFMAs depend on each other; 2 x NCH independent streams per wave).
// Measured: 1 and 2 chains take the same time (the compiler's one s_nop between the two groups' dependent pairs is not
// what bounds the step), so the product uses 1 -- the k order of the sum is then simply 0..127.
template <bool RELU, int NCH = 1>
__global__ __launch_bounds__(256) void lstm4_kernel(const float* __restrict__ pre, const float* __restrict__ whh_f,
                                                     const float* __restrict__ whh_b, float* __restrict__ hc, int ldh,
                                                     int dump_row, SeqGeom g, int nst16, int packed) {
  __shared__ __attribute__((aligned(16))) float Hs[L4_HS_FLOATS];
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, c32 = lane & 31, up = lane >> 5;
  const int st4 = blockIdx.x, d = blockIdx.y;
  const float* whh = d ? whh_b : whh_f;
  const int unit = 32 * w + c32;

  // W_hh rows of the lane's two gate columns, scaled for the exp2 forms of sigmoid / tanh.  From the nn.Module tensor a
  // wave instruction touches 64 rows = 64 cache lines (16 k cycles of address processing per workgroup, ~7 us of a 170 us
  // launch); `packed`: from the copy lstm4_pack_launch makes once per pass, [wave][group][q][lane][4] -- 1 KiB contiguous.
  float wf[2][128];
#pragma unroll
  for (int grp = 0; grp < 2; ++grp) {
    const int gate = 2 * grp + up;
    const float* wrow = packed ? whh + ((w * 2 + grp) * 32) * 256 + lane * 4 : whh + (int64_t)(gate * L16_H + unit) * L16_H;
    const int qstep = packed ? 256 : 4;
    const float gs = l16_gate_scale(gate);
#pragma unroll
    for (int piece = 0; piece < 4; ++piece) {   // handed to the AGPR half in pieces (all 256 at once would spill)
#pragma unroll
      for (int q = 8 * piece; q < 8 * piece + 8; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(wrow + qstep * q);
        wf[grp][4 * q + 0] = v.x * gs;
        wf[grp][4 * q + 1] = v.y * gs;
        wf[grp][4 * q + 2] = v.z * gs;
        wf[grp][4 * q + 3] = v.w * gs;
      }
#pragma unroll
      for (int k = 32 * piece; k < 32 * piece + 32; ++k) asm volatile("" : "+a"(wf[grp][k]));
    }
  }

  const int t0 = d ? g.len - 1 : 0, tdir = d ? -1 : 1;
  const int tstride = seq_token_stride(g);
  // the lane's two sequences: s = 2 up + j of the tile; their h goes to HBM as dwords (a wave's 32 lanes = 128 bytes of a row)
  unsigned soff[2];
  const unsigned sstep = (unsigned)(tdir * tstride * ldh * 4);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int q = st4 * 4 + 2 * up + j;
    const unsigned tokb = q < g.nseq ? (unsigned)seq_token_base(g, q) : (unsigned)dump_row;
    soff[j] = ((tokb + (unsigned)(t0 * tstride)) * (unsigned)ldh + (unsigned)(d * L16_H + unit)) * 4u;   // < 2^32: host
  }
  char* const hcb = reinterpret_cast<char*>(hc);

  // pre-activations: PRE16[d][st16][t][w][b = 2 gate + half][lane16 = 16 (st4 & 3) + i16][4 sequences]
  const float* pre_lane[2];
#pragma unroll
  for (int grp = 0; grp < 2; ++grp) {
    const int gate = 2 * grp + up;
    pre_lane[grp] = pre + pre16_tile_offset(d, st4 >> 2, 0, nst16, g.len) + (int64_t)w * 2048 + (2 * gate + (c32 >> 4)) * 256 +
                    (16 * (st4 & 3) + (c32 & 15)) * 4;
  }
  float4 pf[L4_PF][2];
  auto fetch = [&](int slot, int step) {
    const int sc = step < g.len ? step : g.len - 1;   // past the end: re-request the last tile (in bounds, unused)
    const int64_t o = (int64_t)(t0 + tdir * sc) * L16_TILE_FLOATS;
    pf[slot][0] = *reinterpret_cast<const float4*>(pre_lane[0] + o);
    pf[slot][1] = *reinterpret_cast<const float4*>(pre_lane[1] + o);
  };
#pragma unroll
  for (int u = 0; u < L4_PF; ++u) fetch(u, u);

  for (int i = tid; i < 4 * L16_H; i += 256) Hs[i] = 0.f;   // h_{-1} = 0 (buffer 0)
  f32x2 cst = (f32x2){0.f, 0.f};
  // A operand: lane (i = lane & 3, blk = lane >> 2) reads h[seq i][16 m + blk], m = 0..7, stored as [seq][blk][m]
  const int a_off = ((lane & 3) * 16 + (lane >> 2)) * 8;
  // h_t of (sequence 2 up + j, unit): k = unit -> [seq][blk = unit & 15][m = unit >> 4]
  const int h_off = (2 * up * 16 + (unit & 15)) * 8 + (unit >> 4);
  __syncthreads();

  for (int s0 = 0; s0 < g.len; s0 += L4_PF) {
#pragma unroll
    for (int u = 0; u < L4_PF; ++u) {
      const int step = s0 + u;
      if (step >= g.len) break;   // wave-uniform
      const float* hcur = Hs + (step & 1) * 4 * L16_H;
      float* hnext = Hs + ((step + 1) & 1) * 4 * L16_H;
      const float4 a0 = *reinterpret_cast<const float4*>(hcur + a_off);
      const float4 a1 = *reinterpret_cast<const float4*>(hcur + a_off + 4);
      const float areg[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};

      f32x4v acc[2][NCH];
#pragma unroll
      for (int grp = 0; grp < 2; ++grp) {
        acc[grp][0] = (f32x4v){pf[u][grp].x, pf[u][grp].y, pf[u][grp].z, pf[u][grp].w};
#pragma unroll
        for (int ch = 1; ch < NCH; ++ch) acc[grp][ch] = (f32x4v){0.f, 0.f, 0.f, 0.f};
      }
      fetch(u, step + L4_PF);

#define L4_STEP(B)                                                                    \
  acc[0][m % NCH] = mfma4<B>(areg[m], wf[0][16 * m + B], acc[0][m % NCH]);            \
  acc[1][m % NCH] = mfma4<B>(areg[m], wf[1][16 * m + B], acc[1][m % NCH]);
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        L4_STEP(0) L4_STEP(1) L4_STEP(2) L4_STEP(3) L4_STEP(4) L4_STEP(5) L4_STEP(6) L4_STEP(7)
        L4_STEP(8) L4_STEP(9) L4_STEP(10) L4_STEP(11) L4_STEP(12) L4_STEP(13) L4_STEP(14) L4_STEP(15)
      }
#undef L4_STEP
#pragma unroll
      for (int grp = 0; grp < 2; ++grp)
#pragma unroll
        for (int ch = 1; ch < NCH; ++ch) acc[grp][0] += acc[grp][ch];

      // gather i, f, g, o of (sequence 2 up + j, unit) into this lane: swap the upper half of reg j with the lower half
      // of reg j + 2.  Before: lanes 0..31 hold the pair's first gate, lanes 32..63 its second, regs = sequences 0..3.
      f32x2 gv[4];   // i, f, g, o
#pragma unroll
      for (int grp = 0; grp < 2; ++grp)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[grp][0][j]), __float_as_uint(acc[grp][0][j + 2]),
                                                          false, false);
          gv[2 * grp + 0][j] = __uint_as_float(r[0]);   // first gate of the pair: seq j (lower lanes) / seq j + 2 (upper)
          gv[2 * grp + 1][j] = __uint_as_float(r[1]);   // second gate
        }
      const LstmCell2 cu = lstm_cell2(gv[0], gv[1], gv[2], gv[3], cst);
      cst = cu.c;
      hnext[h_off] = cu.h.x;
      hnext[h_off + 16 * 8] = cu.h.y;
      // h_t leaves for HBM from the registers (position t of the lane's two sequences)
      const float o0 = RELU ? relu1_l4(cu.h.x) : cu.h.x, o1 = RELU ? relu1_l4(cu.h.y) : cu.h.y;
      *reinterpret_cast<float*>(hcb + soff[0]) = o0;
      *reinterpret_cast<float*>(hcb + soff[1]) = o1;
      soff[0] += sstep;
      soff[1] += sstep;
      __syncthreads();
    }
  }
}

int lstm4_launch(bool relu, int nst4, int nst16, int ndir, void* stream, const float* pre, const float* whh_f,
                 const float* whh_b, float* hc, int ldh, int dump_row, const SeqGeom& g, bool packed) {
  using Kern = void (*)(const float*, const float*, const float*, float*, int, int, SeqGeom, int, int);
  const Kern kern = relu ? lstm4_kernel<true> : lstm4_kernel<false>;
  if (nst4 < 1 || nst4 > 4 * nst16) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(kern, dim3(nst4, ndir), dim3(256), 0, static_cast<hipStream_t>(stream), pre, whh_f, whh_b, hc, ldh, dump_row,
                     g, nst16, packed ? 1 : 0);
  return (int)hipGetLastError();
}

// Fragment-order copies of up to LSTM4_PACK_MAX W_hh matrices ([512][128], gate-major rows) for lstm4_kernel's `packed` form:
//   dst[matrix][wave 4][group 2][q 32][lane 64][4] = W_hh[(2 group + (lane >> 5)) * 128 + 32 wave + (lane & 31)][4 q .. 4 q + 3]
struct Lstm4PackArgs {
  const float* src[LSTM4_PACK_MAX];   // nullptr: slot left untouched (a path without a reverse direction)
};
__global__ __launch_bounds__(256) void lstm4_pack_kernel(Lstm4PackArgs a, float* __restrict__ dst) {
  const float* W = a.src[blockIdx.y];
  if (W == nullptr) return;
  float4* out = reinterpret_cast<float4*>(dst + (size_t)blockIdx.y * (4 * L16_H * L16_H));
  for (int f = blockIdx.x * 256 + threadIdx.x; f < L16_H * L16_H; f += gridDim.x * 256) {
    const int lane = f & 63, q = (f >> 6) & 31, grp = (f >> 11) & 1, wv = f >> 12;
    const int row = (2 * grp + (lane >> 5)) * L16_H + 32 * wv + (lane & 31);
    out[f] = *reinterpret_cast<const float4*>(W + (size_t)row * L16_H + 4 * q);
  }
}
int lstm4_pack_launch(void* stream, const float* const* src, int n, float* dst) {
  if (n < 1 || n > LSTM4_PACK_MAX) return (int)hipErrorInvalidValue;
  Lstm4PackArgs a;
  for (int i = 0; i < LSTM4_PACK_MAX; ++i) a.src[i] = i < n ? src[i] : nullptr;
  hipLaunchKernelGGL(lstm4_pack_kernel, dim3(16, n), dim3(256), 0, static_cast<hipStream_t>(stream), a, dst);
  return (int)hipGetLastError();
}
