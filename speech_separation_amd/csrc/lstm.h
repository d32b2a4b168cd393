// lstm.h -- persistent bidirectional LSTM recurrence for gfx950 (H = 128).
//
// Reference semantics: nn.LSTM(N, 128, bidirectional, batch_first) inside TransformerDPRNN
// (src/model/dptn.py:23-29,49): gates i|f|g|o, zero initial state, reverse direction runs t = T-1..0.
// The input projection x W_ih^T + b_ih + b_hh was produced by the GEMM engine (EpiLstmPre); this
// kernel only carries the serial part  gates_t = PRE_t + h_{t-1} W_hh^T.
//
// Mapping: one workgroup (4 waves, one per SIMD) = one direction x one tile of 32 sequences, for ALL
// time steps.  W_hh (512x128 fp32 = 256 KiB) does not fit the 160 KiB LDS, but it fits the register
// file: wave w keeps the rows of W_hh belonging to hidden units [32w, 32w+32) for all four gates as
// 4 x 64 ready-made MFMA B-fragments (256 VGPR/AGPR per lane).  Per step:
//     acc[g] <- PRE tile (one 16-byte load per 4 accumulator registers, fragment layout)
//     acc[g] += h_{t-1}[32 x 128] * W_hh[g-slice]^T      256 x v_mfma_f32_32x32x2_f32 per wave
//     i,f,g,o are the SAME accumulator slot in the four tiles -> the cell update is lane-local
//     h_t -> LDS (double buffered, one barrier per step) and -> HBM as ReLU(h_t) (the only consumer is
//     ffn = ReLU -> Linear, dptn.py:30-33,50)
#pragma once
#include "common.h"

constexpr int LSTM_H = 128;
constexpr int LSTM_LDH = LSTM_H + 4;
constexpr int LSTM_HS_FLOATS = 2 * 32 * LSTM_LDH;          // h double buffer
constexpr int LSTM_PRE_FLOATS = 4 * 16 * 256;               // one step of pre-activations: [wave][gate*4+q][lane][4]
constexpr size_t LSTM_LDS_BYTES = sizeof(float) * (LSTM_HS_FLOATS + LSTM_PRE_FLOATS);

// hc has one extra "dump" row at index M (rows of padded sequences are written there, branch-free).
// STAMP = true is a diagnostic build: per-wave s_memtime sums of the step's segments are written to `stamps`
// ([workgroup][wave][4] cycles: acc-init, MFMA, cell, barrier); its run time is not representative.
// SAVE = true (training forward): the post-activation gates i,f,g,o and the cell state c_t of every step are kept for
// the BPTT kernel, in the same accumulator-fragment order as the pre-activations:
//   tape_gates[d][st][t][cb16][q4][hh2][c32][i4]   (64 KiB per tile and step)
//   tape_c    [d][st][t][w4 ][q4][hh2][c32][i4]    (16 KiB per tile and step)
template <bool STAMP, bool SAVE = false>
__global__ __launch_bounds__(256) void lstm_recurrence_kernel(const float* __restrict__ pre,
                                                               const float* __restrict__ whh_f,
                                                               const float* __restrict__ whh_b,
                                                               float* __restrict__ hc, int ldh, int dump_row,
                                                               SeqGeom g, unsigned long long* __restrict__ stamps,
                                                               int relu_out, float* __restrict__ tape_gates = nullptr,
                                                               float* __restrict__ tape_c = nullptr) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Hs = smem;                      // [2][32][LSTM_LDH]
  float* Ps = smem + LSTM_HS_FLOATS;     // [4 waves][16 pieces][64 lanes][4]

  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, c = lane & 31, hh = lane >> 5;
  const int st = blockIdx.x, d = blockIdx.y;
  const float* whh = d ? whh_b : whh_f;

  // ---- W_hh slice -> registers (B fragments) ---------------------------------------------------
  float wf[4][64];
#pragma unroll
  for (int gi = 0; gi < 4; ++gi) {
    const float* wrow = whh + (int64_t)(gi * LSTM_H + 32 * w + c) * LSTM_H + 4 * hh;
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const float4 v = *reinterpret_cast<const float4*>(wrow + 8 * m);
      wf[gi][4 * m + 0] = v.x;
      wf[gi][4 * m + 1] = v.y;
      wf[gi][4 * m + 2] = v.z;
      wf[gi][4 * m + 3] = v.w;
    }
  }

  // ---- per-lane output element index of (row rho, this step) in hc, advanced by a uniform stride ----
  const int outcol = d * LSTM_H + 32 * w + c;
  const int t0 = d ? g.len - 1 : 0;
  const int tdir = d ? -1 : 1;
  const int tstride = seq_token_stride(g);
  // (rows of padded sequences walk through the dump rows [dump_row, dump_row + S*K) with the same stride)
  unsigned oidx[16];
  const unsigned ostep = (unsigned)(tdir * tstride * ldh);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int q = st * 32 + ROW32(r, hh);
    const unsigned tokb = q < g.nseq ? (unsigned)seq_token_base(g, q) : (unsigned)dump_row;
    oidx[r] = (tokb + (unsigned)(t0 * tstride)) * (unsigned)ldh + (unsigned)outcol;   // < 2^32, checked by the host
  }

  for (int i = tid; i < 32 * LSTM_LDH; i += 256) Hs[i] = 0.f;  // h_{-1} = 0 (buffer 0)
  f32x16 cst = zero16();

  // pre-activation stream: wave w owns pieces (gate gi, quarter q) = column block gi*4+w, 1 KiB each
  const float* pre_lane = pre + pre_tile_offset(d, st, 0, g.nst, g.len) + (int64_t)w * 1024 + lane * 4;
  float* ps_wave = Ps + w * (16 * 256);
  auto issue_pre = [&](int t) {
    const float* p = pre_lane + (int64_t)t * (512 * 32);
#pragma unroll
    for (int gi = 0; gi < 4; ++gi)
#pragma unroll
      for (int q = 0; q < 4; ++q) glds16(p + gi * 4096 + q * 256, ps_wave + (gi * 4 + q) * 256);
  };
  issue_pre(t0);
  __syncthreads();

  // ReLU(h) of the PREVIOUS step is stored during the next step's MFMA block: vmcnt counts stores too, so storing
  // just before the end-of-step barrier would make every step wait for store retirement.
  float hout[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) hout[r] = 0.f;   // step 0 stores these zeros at position t0; step 1 overwrites them
  auto store_prev = [&]() {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      hc[oidx[r]] = hout[r];
      oidx[r] += ostep;
    }
  };

  unsigned long long seg[4] = {0, 0, 0, 0};
  for (int step = 0; step < g.len; ++step) {
    unsigned long long c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    if (STAMP) c0 = __builtin_amdgcn_s_memtime();
    const int t = t0 + tdir * step;
    const float* hcur = Hs + (step & 1) * 32 * LSTM_LDH;
    float* hnext = Hs + ((step + 1) & 1) * 32 * LSTM_LDH;

    // accumulators start from the pre-activations that the LDS-DMA delivered during the previous step
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    f32x16 acc[4];
#pragma unroll
    for (int gi = 0; gi < 4; ++gi)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(ps_wave + (gi * 4 + q) * 256 + lane * 4);
        acc[gi][4 * q + 0] = v.x;
        acc[gi][4 * q + 1] = v.y;
        acc[gi][4 * q + 2] = v.z;
        acc[gi][4 * q + 3] = v.w;
      }
    // the whole A operand (h_{t-1}, 32 x 128) is fetched in one batch as well: a prefetch placed inside the MFMA
    // block gets an s_waitcnt lgkmcnt(0) right behind it and exposes one LDS round trip per k-chunk
    const float* arow = hcur + c * LSTM_LDH + 4 * hh;
    float4 afr[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) afr[m] = *reinterpret_cast<const float4*>(arow + 8 * m);
    // the wave's own LDS region may only be refilled once these reads have returned
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (STAMP) c1 = __builtin_amdgcn_s_memtime();

    // h_{t-1} W_hh^T: 64 pinned groups of 4 MFMAs.  The step's side work rides in the MFMA shadow, one item per
    // group: slots 0..15 issue the LDS-DMA of the NEXT step's pre-activations (the LDS copy was consumed above),
    // slots 16..31 store ReLU(h) of the PREVIOUS step.
    // branch-free: the last step re-requests its own tile, step 0 stores zeros without advancing
    const float* pnext = pre_lane + (int64_t)(step + 1 < g.len ? t + tdir : t) * (512 * 32);
    const unsigned adv = step > 0 ? ostep : 0u;
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const float av[4] = {afr[m].x, afr[m].y, afr[m].z, afr[m].w};
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) acc[gi] = mfma32(av[tt], wf[gi][4 * m + tt], acc[gi]);
        const int slot = 4 * m + tt;
        if (slot < 16) {
          glds16(pnext + (slot >> 2) * 4096 + (slot & 3) * 256, ps_wave + slot * 256);
        } else if (slot < 32) {
          hc[oidx[slot - 16]] = hout[slot - 16];
          oidx[slot - 16] += adv;
        }
        if (slot < 32) __builtin_amdgcn_sched_barrier(0);   // pin only the groups that carry side work
      }
    }
    if (STAMP) {
      asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::"v"(acc[0][0]), "v"(acc[3][15]));
      c2 = __builtin_amdgcn_s_memtime();
    }
    // cell update (lane-local) + publish h_t
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float ig = fast_sigmoid(acc[0][r]);
      const float fg = fast_sigmoid(acc[1][r]);
      const float gg = fast_tanh(acc[2][r]);
      const float og = fast_sigmoid(acc[3][r]);
      const float cn = fmaf(fg, cst[r], ig * gg);
      cst[r] = cn;
      if (SAVE) { acc[0][r] = ig; acc[1][r] = fg; acc[2][r] = gg; acc[3][r] = og; }
      const float hn = og * fast_tanh(cn);
      hnext[ROW32(r, hh) * LSTM_LDH + 32 * w + c] = hn;
      hout[r] = relu_out ? fmaxf(hn, 0.f) : hn;   // DPTN feeds ffn = ReLU -> Linear; DPRNN feeds fc directly
    }
    if (SAVE) {
      float* tg = tape_gates + pre_tile_offset(d, st, t, g.nst, g.len) + (int64_t)w * 1024 + lane * 4;
      float* tc = tape_c + pre_tile_offset(d, st, t, g.nst, g.len) / 4 + (int64_t)w * 1024 + lane * 4;
#pragma unroll
      for (int gi = 0; gi < 4; ++gi)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<float4*>(tg + gi * 4096 + q * 256) =
              make_float4(acc[gi][4 * q], acc[gi][4 * q + 1], acc[gi][4 * q + 2], acc[gi][4 * q + 3]);
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4*>(tc + q * 256) = make_float4(cst[4 * q], cst[4 * q + 1], cst[4 * q + 2], cst[4 * q + 3]);
    }
    if (STAMP) c3 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    if (STAMP) {
      const unsigned long long c4 = __builtin_amdgcn_s_memtime();
      seg[0] += c1 - c0;
      seg[1] += c2 - c1;
      seg[2] += c3 - c2;
      seg[3] += c4 - c3;
    }
  }
  store_prev();
  if (STAMP && lane == 0) {
    unsigned long long* o = stamps + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + w) * 4;
    o[0] = seg[0]; o[1] = seg[1]; o[2] = seg[2]; o[3] = seg[3];
  }
}
