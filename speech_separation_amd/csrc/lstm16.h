// lstm16.h -- the LSTM recurrence on 16-sequence tiles (v_mfma_f32_16x16x4_f32), for launches whose sequence count
// leaves CUs idle with 32-sequence tiles.
//
// dptnav_forward runs the batch as two chained half-batches (dptnav.hip): a half has ~1130 sequences per direction,
// i.e. 72 workgroups of 32 sequences -- 28 % of the chip for 1.4 ms, and the chain of 24 such launches IS the step
// time.  With 16-sequence tiles the same launch occupies 142 CUs for half as long (8192 MFMA cycles per step instead
// of 16384): equal CU-time, half the serial chain, and twice the achieved FLOP/s per launch.
//
// Same structure as lstm.h (W_hh resident in registers as MFMA B fragments, pre-activations by LDS-DMA one step
// ahead, side work pinned into the MFMA shadow, lane-local cell update, one barrier per step); only the fragment
// maps differ.  Lane l: i16 = l & 15, ks = l >> 4;  A[i16][ks], B[ks][i16], D reg r = (row 4ks + r, col i16);
// MFMA step 4m+t uses true k = 16m + 4ks + t.  Wave w owns hidden units [32w, 32w+32) = blocks b = 2*gate + half.
// Pre-activation layout  PRE16[d][st16][t][cb32][lane64][4]  (cb32 = 8*gate + 2w + half; reg i of a lane = row 4ks+i).
#pragma once
#include "common.h"

typedef float f32x4v __attribute__((ext_vector_type(4)));
DEV f32x4v mfma16(float a, float b, f32x4v c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

constexpr int L16_H = 128;
constexpr int L16_LDH = L16_H + 8;                    // 136: conflict-free ds_read_b128 for the 16x16x4 A map
constexpr int L16_HS_FLOATS = 2 * 16 * L16_LDH;       // h double buffer
constexpr int L16_PRE_FLOATS = 4 * 8 * 256;           // [wave][piece][lane*4]
constexpr size_t L16_LDS_BYTES = sizeof(float) * (L16_HS_FLOATS + L16_PRE_FLOATS);

DEV int64_t pre16_tile_offset(int d, int st16, int t, int nst16, int len) {
  return (((int64_t)d * nst16 + st16) * len + t) * (int64_t)(512 * 16);
}

__global__ __launch_bounds__(256) void lstm16_kernel(const float* __restrict__ pre, const float* __restrict__ whh_f,
                                                      const float* __restrict__ whh_b, float* __restrict__ hc, int ldh,
                                                      int dump_row, SeqGeom g, int nst16, int relu_out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Hs = smem;                     // [2][16][L16_LDH]
  float* Ps = smem + L16_HS_FLOATS;     // [4][8][256]
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, i16 = lane & 15, ks = lane >> 4;
  const int st = blockIdx.x, d = blockIdx.y;
  const float* whh = d ? whh_b : whh_f;

  float wf[8][32];
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const float* wrow = whh + (int64_t)((b >> 1) * L16_H + 32 * w + 16 * (b & 1) + i16) * L16_H + 4 * ks;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const float4 v = *reinterpret_cast<const float4*>(wrow + 16 * m);
      wf[b][4 * m + 0] = v.x;
      wf[b][4 * m + 1] = v.y;
      wf[b][4 * m + 2] = v.z;
      wf[b][4 * m + 3] = v.w;
    }
  }

  const int t0 = d ? g.len - 1 : 0, tdir = d ? -1 : 1;
  const int tstride = seq_token_stride(g);
  const unsigned ostep = (unsigned)(tdir * tstride * ldh);
  unsigned oidx[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int q = st * 16 + 4 * ks + r;
    const unsigned tokb = q < g.nseq ? (unsigned)seq_token_base(g, q) : (unsigned)dump_row;
    oidx[r] = (tokb + (unsigned)(t0 * tstride)) * (unsigned)ldh + (unsigned)(d * L16_H + 32 * w + i16);
  }
  for (int i = tid; i < 16 * L16_LDH; i += 256) Hs[i] = 0.f;      // h_{-1} = 0 (buffer 0)
  f32x4v cst[2] = {(f32x4v){0.f, 0.f, 0.f, 0.f}, (f32x4v){0.f, 0.f, 0.f, 0.f}};
  float hout[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) hout[e] = 0.f;

  const float* pre_lane = pre + pre16_tile_offset(d, st, 0, nst16, g.len) + (int64_t)(2 * w) * 256 + lane * 4;
  float* ps_wave = Ps + w * (8 * 256);
#pragma unroll
  for (int b = 0; b < 8; ++b)
    glds16(pre_lane + (int64_t)t0 * (512 * 16) + ((b >> 1) * 8 + (b & 1)) * 256, ps_wave + b * 256);
  __syncthreads();

  for (int step = 0; step < g.len; ++step) {
    const int t = t0 + tdir * step;
    const float* hcur = Hs + (step & 1) * 16 * L16_LDH;
    float* hnext = Hs + ((step + 1) & 1) * 16 * L16_LDH;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    f32x4v acc[8];
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const float4 v = *reinterpret_cast<const float4*>(ps_wave + b * 256 + lane * 4);
      acc[b] = (f32x4v){v.x, v.y, v.z, v.w};
    }
    const float* arow = hcur + i16 * L16_LDH + 4 * ks;
    float4 afr[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) afr[m] = *reinterpret_cast<const float4*>(arow + 16 * m);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    const float* pnext = pre_lane + (int64_t)(step + 1 < g.len ? t + tdir : t) * (512 * 16);
    const unsigned adv = step > 0 ? ostep : 0u;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const float av[4] = {afr[m].x, afr[m].y, afr[m].z, afr[m].w};
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[b] = mfma16(av[tt], wf[b][4 * m + tt], acc[b]);
        const int slot = 4 * m + tt;
        if (slot < 8) {
          glds16(pnext + ((slot >> 1) * 8 + (slot & 1)) * 256, ps_wave + slot * 256);
        } else if (slot < 16) {
          const int e = slot - 8, hf = e >> 2, r = e & 3;
          hc[oidx[r] + 16 * hf] = hout[e];
          if (hf == 1) oidx[r] += adv;
        }
        if (slot < 16) __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int hf = e >> 2, r = e & 3;
      const float ig = fast_sigmoid(acc[0 + hf][r]);
      const float fg = fast_sigmoid(acc[2 + hf][r]);
      const float gg = fast_tanh(acc[4 + hf][r]);
      const float og = fast_sigmoid(acc[6 + hf][r]);
      const float cn = fmaf(fg, cst[hf][r], ig * gg);
      cst[hf][r] = cn;
      const float hn = og * fast_tanh(cn);
      hnext[(4 * ks + r) * L16_LDH + 32 * w + 16 * hf + i16] = hn;
      hout[e] = relu_out ? fmaxf(hn, 0.f) : hn;
    }
    __syncthreads();
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) hc[oidx[e & 3] + 16 * (e >> 2)] = hout[e];
}
