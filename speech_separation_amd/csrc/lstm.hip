// lstm.hip -- persistent bidirectional LSTM recurrence for gfx950 on 32-sequence tiles (H = 128).
//
// Reference semantics: nn.LSTM(N, 128, bidirectional, batch_first) inside TransformerDPRNN
// (src/model/dptn.py:23-29,49) and the DPRNN blocks (src/model/dprnn.py:24-47): gates i|f|g|o, zero initial state,
// reverse direction runs t = T-1..0.  The input projection x W_ih^T + b_ih + b_hh was produced by the GEMM engine
// (EpiLstmPre); this kernel only carries the serial part  gates_t = PRE_t + h_{t-1} W_hh^T.
//
// Mapping: one workgroup (4 waves, one per SIMD) = one direction x one tile of 32 sequences, for ALL time steps.
// W_hh (512x128 fp32 = 256 KiB) does not fit the 160 KiB LDS, but it fits the register file: wave w keeps the rows of
// W_hh belonging to hidden units [32w, 32w+32) for all four gates as 4 x 64 ready-made MFMA B-fragments (the 256
// AGPRs of a lane).  Per step:
//     acc[g] <- PRE tile (LDS-DMA'd one step ahead; one 16-byte LDS read per 4 accumulator registers)
//     acc[g] += h_{t-1}[32 x 128] * W_hh[g-slice]^T      256 x v_mfma_f32_32x32x2_f32 per wave
//     i,f,g,o are the SAME accumulator slot in the four tiles -> the cell update is lane-local
//     h_t -> LDS (double buffered, one barrier per step); h_{t-1} -> HBM as whole 512-byte rows read back from LDS
// Like lstm16.hip the step is written for a minimal vector-instruction count (beside fp32 MFMAs every VALU
// instruction costs its full issue time): accumulators in VGPRs / W_hh in AGPRs, pre-scaled gates,
// immediate-offset LDS-DMA, 16-byte row stores, ReLU / tape as template flags.
#include <hip/hip_runtime.h>

#include "lstm.h"

template <int OFF>
DEV void glds16_off(const float* gsrc, float* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, OFF, 0);
}
// ReLU in ONE instruction on a value that comes from memory (fmaxf would canonicalise first)
DEV float relu1(float x) { return __int_as_float(max(__float_as_int(x), 0)); }

template <bool STAMP, bool SAVE, bool RELU>
__global__ __launch_bounds__(256) void lstm_recurrence_kernel(const float* __restrict__ pre,
                                                               const float* __restrict__ whh_f,
                                                               const float* __restrict__ whh_b,
                                                               float* __restrict__ hc, int ldh, int dump_row,
                                                               SeqGeom g, unsigned long long* __restrict__ stamps,
                                                               float* __restrict__ tape_gates,
                                                               float* __restrict__ tape_c) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Hs = smem;                      // [2][32][LSTM_LDH]
  float* Ps = smem + LSTM_HS_FLOATS;     // [4 waves][16 pieces][64 lanes][4]

  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, c = lane & 31, hh = lane >> 5;
  const int st = blockIdx.x, d = blockIdx.y;
  const float* whh = d ? whh_b : whh_f;

  // ---- W_hh slice -> registers (B fragments), gate rows pre-scaled; handed to the AGPR half of the file ----------
  float wf[4][64];
#pragma unroll
  for (int gi = 0; gi < 4; ++gi) {
    const float* wrow = whh + (int64_t)(gi * LSTM_H + 32 * w + c) * LSTM_H + 4 * hh;
    const float gs = lstm_gate_scale(gi);
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const float4 v = *reinterpret_cast<const float4*>(wrow + 8 * m);
      wf[gi][4 * m + 0] = v.x * gs;
      wf[gi][4 * m + 1] = v.y * gs;
      wf[gi][4 * m + 2] = v.z * gs;
      wf[gi][4 * m + 3] = v.w * gs;
    }
  }
#pragma unroll
  for (int gi = 0; gi < 4; ++gi)
#pragma unroll
    for (int i = 0; i < 64; ++i) asm volatile("" : "+a"(wf[gi][i]));

  // ---- h rows leave through the LDS tile: wave w stores tile rows 8w + 2j + (lane >> 5), j = 0..3, as 16 bytes at
  //      column 4 * (lane & 31) of the direction's 128 outputs; one pointer per row, advanced by a uniform stride ----
  const int t0 = d ? g.len - 1 : 0;
  const int tdir = d ? -1 : 1;
  const int tstride = seq_token_stride(g);
  const int srow = 8 * w + hh, scol = 4 * c;
  const int64_t sstep = (int64_t)tdir * tstride * ldh;
  float* sp[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int q = st * 32 + srow + 2 * j;
    // (rows of padded sequences walk through the dump rows [dump_row, dump_row + S*K) with the same stride)
    const int64_t tokb = q < g.nseq ? seq_token_base(g, q) : (int64_t)dump_row;
    sp[j] = hc + (tokb + (int64_t)t0 * tstride) * ldh + d * LSTM_H + scol;
  }

  for (int i = tid; i < 32 * LSTM_LDH; i += 256) Hs[i] = 0.f;  // h_{-1} = 0 (buffer 0)
  f32x16 cst = zero16();

  // pre-activation stream: wave w owns pieces (gate gi, quarter q) = column block gi*4+w, 1 KiB each
  const float* pre_lane = pre + pre_tile_offset(d, st, 0, g.nst, g.len) + (int64_t)w * 1024 + lane * 4;
  float* ps_wave = Ps + w * (16 * 256);
  auto issue_pre = [&](const float* p) {   // one address per gate + immediate offsets for its four quarters
#pragma unroll
    for (int gi = 0; gi < 4; ++gi) {
      glds16_off<0>(p + gi * 4096, ps_wave + gi * 1024);
      glds16_off<1024>(p + gi * 4096, ps_wave + gi * 1024);
      glds16_off<2048>(p + gi * 4096, ps_wave + gi * 1024);
      glds16_off<3072>(p + gi * 4096, ps_wave + gi * 1024);
    }
  };
  auto issue_pre_piece = [&](const float* p, int piece) {   // piece = gi*4 + q (compile-time after unrolling)
    const int gi = piece >> 2;
    switch (piece & 3) {
      case 0: glds16_off<0>(p + gi * 4096, ps_wave + gi * 1024); break;
      case 1: glds16_off<1024>(p + gi * 4096, ps_wave + gi * 1024); break;
      case 2: glds16_off<2048>(p + gi * 4096, ps_wave + gi * 1024); break;
      default: glds16_off<3072>(p + gi * 4096, ps_wave + gi * 1024); break;
    }
  };
  issue_pre(pre_lane + (int64_t)t0 * (512 * 32));
  __syncthreads();

  // accumulators start from the pre-activations that the LDS-DMA delivers one step ahead; they are fetched from LDS
  // BEFORE the barrier that closes a step (they do not depend on h)
  f32x16 acc[4];
  auto preload_acc = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
  };
  preload_acc();

  unsigned long long seg[4] = {0, 0, 0, 0};
  for (int step = 0; step < g.len; ++step) {
    unsigned long long c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    if (STAMP) c0 = __builtin_amdgcn_s_memtime();
    const int t = t0 + tdir * step;
    const float* hcur = Hs + (step & 1) * 32 * LSTM_LDH;
    float* hnext = Hs + ((step + 1) & 1) * 32 * LSTM_LDH;

    // the whole A operand (h_{t-1}, 32 x 128) is fetched in one batch: a fragment read placed inside the MFMA block
    // gets an s_waitcnt lgkmcnt(0) right behind it and exposes one LDS round trip per k-chunk.  Plus the four
    // h_{t-1} rows this lane sends to HBM.
    const float* arow = hcur + c * LSTM_LDH + 4 * hh;
    float4 afr[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) afr[m] = *reinterpret_cast<const float4*>(arow + 8 * m);
    float4 hs[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) hs[j] = *reinterpret_cast<const float4*>(hcur + (srow + 2 * j) * LSTM_LDH + scol);
    // the wave's own LDS region may only be refilled once these reads (and the accumulator preload) have returned
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int gi = 0; gi < 4; ++gi) asm volatile("" : "+v"(acc[gi]));   // accumulators in architectural VGPRs
    // (without this pin the allocator tries the full AGPR half for these load results and spills them)
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(hs[j].x), "+v"(hs[j].y), "+v"(hs[j].z), "+v"(hs[j].w));
    if (STAMP) c1 = __builtin_amdgcn_s_memtime();

    // branch-free: the last step re-requests its own tile; step 0 stores the zeros of h_{-1} at position t0 without
    // advancing and step 1 overwrites them (same lane, same address, program order)
    const float* pnext = pre_lane + (int64_t)(step + 1 < g.len ? t + tdir : t) * (512 * 32);
    const int64_t adv = step > 0 ? sstep : 0;
    if (RELU) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        hs[j] = make_float4(relu1(hs[j].x), relu1(hs[j].y), relu1(hs[j].z), relu1(hs[j].w));
    }

    // h_{t-1} W_hh^T: 64 groups of 4 MFMAs; the next step's LDS-DMA requests and the stores of h_{t-1} sit behind the
    // first groups (vmcnt counts stores too: storing just before the end-of-step barrier would make every step wait
    // for store retirement)
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const float av[4] = {afr[m].x, afr[m].y, afr[m].z, afr[m].w};
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) acc[gi] = mfma32(av[tt], wf[gi][4 * m + tt], acc[gi]);
        const int slot = 4 * m + tt;
        // one memory instruction per MFMA group: slots 0..3 the stores of h_{t-1}, slots 4..19 the LDS-DMA requests
        // (issued back to back they stall the wave on the memory pipeline's queue)
        if (slot < 4) {
          *reinterpret_cast<float4*>(sp[slot]) = hs[slot];
          sp[slot] += adv;
          __builtin_amdgcn_sched_barrier(0);
        } else if (slot < 20) {
          issue_pre_piece(pnext, slot - 4);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if (STAMP) {
      asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::"v"(acc[0][0]), "v"(acc[3][15]));
      c2 = __builtin_amdgcn_s_memtime();
    }
    // cell update (lane-local, two accumulator slots per call) + publish h_t
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const LstmCell2 u = lstm_cell2((f32x2){acc[0][r], acc[0][r + 1]}, (f32x2){acc[1][r], acc[1][r + 1]},
                                     (f32x2){acc[2][r], acc[2][r + 1]}, (f32x2){acc[3][r], acc[3][r + 1]},
                                     (f32x2){cst[r], cst[r + 1]});
      cst[r] = u.c.x;
      cst[r + 1] = u.c.y;
      if (SAVE) {
        acc[0][r] = u.i.x; acc[0][r + 1] = u.i.y;
        acc[1][r] = u.f.x; acc[1][r + 1] = u.f.y;
        acc[2][r] = u.g.x; acc[2][r + 1] = u.g.y;
        acc[3][r] = u.o.x; acc[3][r + 1] = u.o.y;
      }
      hnext[ROW32(r, hh) * LSTM_LDH + 32 * w + c] = u.h.x;
      hnext[ROW32(r + 1, hh) * LSTM_LDH + 32 * w + c] = u.h.y;
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
    if (step + 1 < g.len) preload_acc();
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
  // h of the last step: the barrier above published it in buffer (len & 1)
  {
    const float* hfin = Hs + (g.len & 1) * 32 * LSTM_LDH;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float4 v = *reinterpret_cast<const float4*>(hfin + (srow + 2 * j) * LSTM_LDH + scol);
      if (RELU) v = make_float4(relu1(v.x), relu1(v.y), relu1(v.z), relu1(v.w));
      *reinterpret_cast<float4*>(sp[j]) = v;
    }
  }
  if (STAMP && lane == 0) {
    unsigned long long* o = stamps + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + w) * 4;
    o[0] = seg[0]; o[1] = seg[1]; o[2] = seg[2]; o[3] = seg[3];
  }
}

int lstm32_launch(bool stamp, bool save, bool relu, int nst, int ndir, void* stream, const float* pre, const float* whh_f,
                  const float* whh_b, float* hc, int ldh, int dump_row, const SeqGeom& g, unsigned long long* stamps,
                  float* tape_gates, float* tape_c) {
  using Kern = void (*)(const float*, const float*, const float*, float*, int, int, SeqGeom, unsigned long long*, float*,
                        float*);
  Kern kern;
  int id;
  if (save) { kern = lstm_recurrence_kernel<false, true, false>; id = 0; }
  else if (stamp) { kern = relu ? lstm_recurrence_kernel<true, false, true> : lstm_recurrence_kernel<true, false, false>; id = 1 + relu; }
  else { kern = relu ? lstm_recurrence_kernel<false, false, true> : lstm_recurrence_kernel<false, false, false>; id = 3 + relu; }
  static PerDeviceOnce ready[5];
  const int dev = current_hip_device();
  if (!ready[id].done(dev)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)LSTM_LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    ready[id].set(dev);
  }
  hipLaunchKernelGGL(kern, dim3(nst, ndir), dim3(256), LSTM_LDS_BYTES, static_cast<hipStream_t>(stream), pre, whh_f, whh_b, hc,
                     ldh, dump_row, g, stamps, tape_gates, tape_c);
  return (int)hipGetLastError();
}
