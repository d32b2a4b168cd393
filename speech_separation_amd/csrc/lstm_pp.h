// lstm_pp.h -- "ping-pong" LSTM recurrence for gfx950 (H = 128): the cell update of one half of the sequence
// tile runs in the shadow of the other half's MFMAs.
//
// Same mapping as lstm.h (one workgroup = one direction x 32 sequences x all steps, W_hh resident in registers as
// MFMA B-fragments) but on v_mfma_f32_16x16x4_f32, which lets the 32-sequence tile be two INDEPENDENT 16-sequence
// sub-tiles A and B.  A wave issues in order, so VALU work only overlaps matrix work if it sits BETWEEN that
// wave's own MFMAs; with two sub-tiles there is always independent VALU work available:
//
//     phase 1:  MFMA(A, t)  ||  cell update + publish h of (B, t-1)      barrier
//     phase 2:  MFMA(B, t)  ||  cell update + publish h of (A, t)        barrier
//
// Fragment maps (lane l: i16 = l & 15, ks = l >> 4), 16x16x4:  A[i16][ks], B[ks][i16], D reg r = (row 4ks+r, col i16).
// k-permutation: MFMA step 4m+t uses true k = 16m + 4ks + t (16-byte fragment fetches, as in common.h).
// Wave w owns hidden units [32w, 32w+32): for gate g two 16-column blocks (half = 0,1) -> 8 blocks, 256 W registers.
// Pre-activations arrive by LDS-DMA one step ahead in the PRE16 layout
//     PRE16[d][st][t][sub(2)][cb(32)][lane(64)][i(4)]          (cb = 8g + 2w + half; reg i of lane = row 4ks+i)
#pragma once
#include "common.h"

typedef float f32x4v __attribute__((ext_vector_type(4)));

DEV f32x4v mfma16(float a, float b, f32x4v c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

constexpr int PP_H = 128;
constexpr int PP_LDH = PP_H + 8;                      // 136: conflict-free ds_read_b128 for the 16x16x4 A map
constexpr int PP_HS_FLOATS = 2 * 16 * PP_LDH;         // h of sub-tile A and B (single buffered, two barriers/step)
constexpr int PP_PRE_FLOATS = 2 * 4 * 8 * 256;        // [sub][wave][piece][lane*4]
constexpr size_t PP_LDS_BYTES = sizeof(float) * (PP_HS_FLOATS + PP_PRE_FLOATS);

DEV int64_t pre16_tile_offset(int d, int st, int t, int nst, int len) {
  return (((int64_t)d * nst + st) * len + t) * (int64_t)(512 * 32);
}

// hc must have room for the rows of padded sequences: they are written to rows [dump_row, dump_row + S*K).
__global__ __launch_bounds__(256) void lstm_pp_kernel(const float* __restrict__ pre, const float* __restrict__ whh_f,
                                                       const float* __restrict__ whh_b, float* __restrict__ hc,
                                                       int ldh, int dump_row, SeqGeom g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Hs = smem;                    // [2][16][PP_LDH]
  float* Ps = smem + PP_HS_FLOATS;     // [2][4][8][256]

  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, i16 = lane & 15, ks = lane >> 4;
  const int st = blockIdx.x, d = blockIdx.y;
  const float* whh = d ? whh_b : whh_f;

  // ---- W_hh slice -> registers ---------------------------------------------------------------------
  float wf[8][32];
#pragma unroll
  for (int b = 0; b < 8; ++b) {           // b = 2*gate + half
    const float* wrow = whh + (int64_t)((b >> 1) * PP_H + 32 * w + 16 * (b & 1) + i16) * PP_H + 4 * ks;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const float4 v = *reinterpret_cast<const float4*>(wrow + 16 * m);
      wf[b][4 * m + 0] = v.x;
      wf[b][4 * m + 1] = v.y;
      wf[b][4 * m + 2] = v.z;
      wf[b][4 * m + 3] = v.w;
    }
  }

  // ---- output indices: element (sub, half, r) of this lane is row 16 sub + 4 ks + r, unit 32w + 16 half + i16 ----
  const int t0 = d ? g.len - 1 : 0;
  const int tdir = d ? -1 : 1;
  const int tstride = seq_token_stride(g);
  const unsigned ostep = (unsigned)(tdir * tstride * ldh);
  unsigned oidx[2][4];                       // [sub][r]; + 16*half added at the store
#pragma unroll
  for (int sub = 0; sub < 2; ++sub)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int q = st * 32 + 16 * sub + 4 * ks + r;
      const int tokb = q < g.nseq ? (int)seq_token_base(g, q) : dump_row;
      oidx[sub][r] = (unsigned)((tokb + t0 * tstride) * ldh + d * PP_H + 32 * w + i16);
    }

  for (int i = tid; i < PP_HS_FLOATS; i += 256) Hs[i] = 0.f;   // h_{-1} = 0 for both sub-tiles
  f32x4v cst[2][2];                                             // [sub][half]
#pragma unroll
  for (int sub = 0; sub < 2; ++sub)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) cst[sub][hf] = (f32x4v){0.f, 0.f, 0.f, 0.f};

  // ---- pre-activation stream -------------------------------------------------------------------------
  const float* pre_lane = pre + pre16_tile_offset(d, st, 0, g.nst, g.len) + (int64_t)(2 * w) * 256 + lane * 4;
  auto issue_pre = [&](int sub, int t) {
    const float* p = pre_lane + (int64_t)t * (512 * 32) + sub * 8192;
    float* dst = Ps + (sub * 4 + w) * (8 * 256);
#pragma unroll
    for (int b = 0; b < 8; ++b)    // column block cb = 8*gate + 2w + half
      glds16(p + ((b >> 1) * 8 + (b & 1)) * 256, dst + b * 256);
  };
  issue_pre(0, t0);
  issue_pre(1, t0);
  __syncthreads();

  f32x4v acc[2][8];       // [sub][2*gate + half]
  float hout[2][2][4];    // ReLU(h) waiting to be stored: [sub][half][r]

  // load accumulators of sub-tile SUB from the LDS copy of its pre-activations, refill the copy for step+1
  auto begin_sub = [&](int sub, int t, bool more) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const float* src = Ps + (sub * 4 + w) * (8 * 256) + lane * 4;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const float4 v = *reinterpret_cast<const float4*>(src + b * 256);
      acc[sub][b] = (f32x4v){v.x, v.y, v.z, v.w};
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (more) issue_pre(sub, t + tdir);
  };
  auto store_sub = [&](int sub) {
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
      for (int r = 0; r < 4; ++r) hc[oidx[sub][r] + 16 * hf] = hout[sub][hf][r];
#pragma unroll
    for (int r = 0; r < 4; ++r) oidx[sub][r] += ostep;
  };
  // MFMA of sub-tile SUB interleaved (by the scheduler directives below) with the cell update of sub-tile OTH
  auto phase = [&](int sub, int oth, bool do_cell) {
    const float* arow = Hs + sub * (16 * PP_LDH) + i16 * PP_LDH + 4 * ks;
    float* hw = Hs + oth * (16 * PP_LDH) + (4 * ks) * PP_LDH + 32 * w + i16;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const float4 a = *reinterpret_cast<const float4*>(arow + 16 * m);
      // 16x16x4 has a 40-cycle dependent latency vs a 32-cycle issue interval: walk the 8 independent
      // accumulators in the inner loop so that no MFMA waits for its predecessor
      const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[sub][b] = mfma16(av[t], wf[b][4 * m + t], acc[sub][b]);
      if (do_cell) {   // element m of the other sub-tile: half = m >> 2, r = m & 3
        const int hf = m >> 2, r = m & 3;
        const float ig = fast_sigmoid(acc[oth][0 + hf][r]);
        const float fg = fast_sigmoid(acc[oth][2 + hf][r]);
        const float gg = fast_tanh(acc[oth][4 + hf][r]);
        const float og = fast_sigmoid(acc[oth][6 + hf][r]);
        const float cn = fmaf(fg, cst[oth][hf][r], ig * gg);
        cst[oth][hf][r] = cn;
        const float hn = og * fast_tanh(cn);
        hw[r * PP_LDH + 16 * hf] = hn;
        hout[oth][hf][r] = fmaxf(hn, 0.f);
      }
    }
    // scheduling: one VALU/LDS-write slot behind every MFMA so that the cell update hides in the MFMA gaps
#pragma unroll
    for (int i = 0; i < 256; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);   // up to 2 VALU
    }
  };

  // step 0, phase 1: MFMA(A, t0) alone (there is no cell(B, -1))
  begin_sub(0, t0, g.len > 1);
  phase(0, 1, false);
  __syncthreads();
  for (int step = 0; step < g.len; ++step) {
    const int t = t0 + tdir * step;
    // phase 2 of this step: MFMA(B, t) || cell(A, t)
    begin_sub(1, t, step + 1 < g.len);
    if (step > 0) store_sub(1);              // ReLU(h_B(t-1)), produced in the phase before this one
    phase(1, 0, true);
    __syncthreads();
    if (step + 1 < g.len) {
      // phase 1 of the next step: MFMA(A, t+1) || cell(B, t)
      begin_sub(0, t + tdir, step + 2 < g.len);
      store_sub(0);                          // ReLU(h_A(t)), produced in the phase before this one
      phase(0, 1, true);
      __syncthreads();
    }
  }
  // drain: cell(B, len-1), then the last stores
  {
    float* hw = Hs + 1 * (16 * PP_LDH) + (4 * ks) * PP_LDH + 32 * w + i16;
    (void)hw;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int hf = m >> 2, r = m & 3;
      const float ig = fast_sigmoid(acc[1][0 + hf][r]);
      const float fg = fast_sigmoid(acc[1][2 + hf][r]);
      const float gg = fast_tanh(acc[1][4 + hf][r]);
      const float og = fast_sigmoid(acc[1][6 + hf][r]);
      const float cn = fmaf(fg, cst[1][hf][r], ig * gg);
      hout[1][hf][r] = fmaxf(og * fast_tanh(cn), 0.f);
    }
  }
  store_sub(0);
  store_sub(1);
}
