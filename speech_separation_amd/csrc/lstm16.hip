// lstm16.hip -- LSTM recurrence kernel on 16-sequence tiles for gfx950 (see lstm16.h for the why and the data layout).
//
// Reference semantics: torch.nn.LSTM(batch_first=True) cell, gate order i, f, g, o (src/model/dptn.py:22-30,
// src/model/dprnn.py:24-47): one workgroup = one direction x 16 sequences x all positions.
//
// Structure (as lstm.h): W_hh resident in registers as MFMA B fragments, pre-activations delivered by LDS-DMA one
// step ahead, lane-local cell update, h_t exchanged through a double-buffered LDS tile, one barrier per step.
// Fragment map of v_mfma_f32_16x16x4_f32, lane l: i16 = l & 15, ks = l >> 4;  A[i16][ks], B[ks][i16],
// D reg r = (row 4ks + r, col i16);  MFMA step 4m+t uses true k = 16m + 4ks + t (so A and W fragments are 16-byte
// loads).  Wave w owns hidden units [32w, 32w+32) as 8 blocks b = 2*gate + half of 16 units.
//
// What shapes this file: beside fp32 MFMAs every other vector instruction of the wave costs its full issue time --
// nothing hides in an "MFMA shadow" (tools/microbench/mfma16_rate.hip: one independent v_fma per MFMA gap = +8
// cycles, one v_exp = +10) -- and with W_hh filling the register file there is exactly one wave per SIMD.  So the
// step is written for a minimal instruction count:
//   * compiled with -mllvm -amdgpu-mfma-vgpr-form: accumulators live in architectural VGPRs, the 256 W_hh
//     fragments in AGPRs (MFMA reads its B operand from there) -> no v_accvgpr moves in the loop;
//   * gate pre-activations arrive pre-scaled by -log2(e) / -2 log2(e) (W_hh is scaled at load, PRE16 by its
//     producer), so sigmoid = rcp(1 + exp2(a)), tanh = 2 rcp(1 + exp2(a)) - 1: no multiply;
//   * the wave's 8 LDS-DMA pieces are contiguous in PRE16 and in LDS: one address per step + immediate offsets;
//   * h_{t-1} leaves as two 16-byte row stores per lane, read back from the LDS tile the recurrence uses anyway
//     (512-byte contiguous rows), instead of eight scattered dword stores;
//   * ReLU-or-not is a template flag.
#include <hip/hip_runtime.h>

#include "lstm16.h"

typedef float f32x4v __attribute__((ext_vector_type(4)));
DEV f32x4v mfma16(float a, float b, f32x4v c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// LDS-DMA with an immediate offset (applied to the global AND the LDS address)
template <int OFF>
DEV void glds16_off(const float* gsrc, float* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, OFF, 0);
}

// ReLU in ONE instruction (v_max_i32 on the bit pattern: negative floats are negative integers): fmaxf on a value
// that comes from memory costs a canonicalising v_max_f32 first
DEV float relu1(float x) { return __int_as_float(max(__float_as_int(x), 0)); }


// STAMP: diagnostic build, per-wave s_memtime sums of [init, MFMA block (with the interleaved cell update), exposed
// cell update, barrier] go to stamps[dir][tile][wave][4].  DIAG (timing-only ablations, results are wrong):
// bit 0 no LDS-DMA, bit 1 no h stores, bit 3 identity activations.
// SAVE (training forward): the post-activation gates and the cell state of every step are kept for lstm_bptt16.hip, in
// the fragment order of this kernel:
//     tape_gates[d][st16][t][w][b = 2*gate + half][lane64][4]   (32 KiB per tile and step, = the PRE16 layout)
//     tape_c    [d][st16][t][w][half][lane64][4]                ( 8 KiB per tile and step)
// The ten 16-byte stores of a step ride between the NEXT step's first MFMA groups (one memory instruction per group).
template <bool STAMP, bool RELU, int DIAG, bool SAVE = false>
__global__ __launch_bounds__(256) void lstm16_kernel(const float* __restrict__ pre, const float* __restrict__ whh_f,
                                                      const float* __restrict__ whh_b, float* __restrict__ hc, int ldh,
                                                      int dump_row, SeqGeom g, int nst16,
                                                      unsigned long long* __restrict__ stamps,
                                                      float* __restrict__ tape_gates = nullptr,
                                                      float* __restrict__ tape_c = nullptr) {
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
    const float gs = l16_gate_scale(b >> 1);
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const float4 v = *reinterpret_cast<const float4*>(wrow + 16 * m);
      wf[b][4 * m + 0] = v.x * gs;
      wf[b][4 * m + 1] = v.y * gs;
      wf[b][4 * m + 2] = v.z * gs;
      wf[b][4 * m + 3] = v.w * gs;
    }
  }

  // hand the fragments to the AGPR half of the register file once; the MFMAs then read their B operand there
#pragma unroll
  for (int b = 0; b < 8; ++b)
#pragma unroll
    for (int i = 0; i < 32; ++i) asm volatile("" : "+a"(wf[b][i]));

  const int t0 = d ? g.len - 1 : 0, tdir = d ? -1 : 1;
  const int tstride = seq_token_stride(g);
  // h rows leave through the LDS tile: wave w stores tile rows 4w + 2j + (lane >> 5), j = 0, 1, as 16 bytes at column
  // 4 * (lane & 31) of the direction's 128 outputs.  Rows of padded sequences walk through the dump rows.
  const int srow = 4 * w + (lane >> 5), scol = 4 * (lane & 31);
  unsigned soff[2];
  const unsigned sstep = (unsigned)(tdir * tstride * ldh * 4);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int q = st * 16 + srow + 2 * j;
    const unsigned tokb = q < g.nseq ? (unsigned)seq_token_base(g, q) : (unsigned)dump_row;
    soff[j] = ((tokb + (unsigned)(t0 * tstride)) * (unsigned)ldh + (unsigned)(d * L16_H + scol)) * 4u;   // < 2^32: host
  }
  char* const hcb = reinterpret_cast<char*>(hc);

  for (int i = tid; i < 16 * L16_LDH; i += 256) Hs[i] = 0.f;      // h_{-1} = 0 (buffer 0)
  f32x4v cst[2] = {(f32x4v){0.f, 0.f, 0.f, 0.f}, (f32x4v){0.f, 0.f, 0.f, 0.f}};

  const float* pre_lane = pre + pre16_tile_offset(d, st, 0, nst16, g.len) + (int64_t)w * 2048 + lane * 4;
  float* ps_wave = Ps + w * 2048;
  auto issue_pre = [&](const float* p) {   // the wave's 8 KiB of one step: two addresses, four immediates each
    glds16_off<0>(p, ps_wave);
    glds16_off<1024>(p, ps_wave);
    glds16_off<2048>(p, ps_wave);
    glds16_off<3072>(p, ps_wave);
    glds16_off<0>(p + 1024, ps_wave + 1024);
    glds16_off<1024>(p + 1024, ps_wave + 1024);
    glds16_off<2048>(p + 1024, ps_wave + 1024);
    glds16_off<3072>(p + 1024, ps_wave + 1024);
  };
  auto issue_pre_piece = [&](const float* p, int piece) {   // compile-time after unrolling
    const int hi = piece >> 2;
    switch (piece & 3) {
      case 0: glds16_off<0>(p + hi * 1024, ps_wave + hi * 1024); break;
      case 1: glds16_off<1024>(p + hi * 1024, ps_wave + hi * 1024); break;
      case 2: glds16_off<2048>(p + hi * 1024, ps_wave + hi * 1024); break;
      default: glds16_off<3072>(p + hi * 1024, ps_wave + hi * 1024); break;
    }
  };
  issue_pre(pre_lane + (int64_t)t0 * L16_TILE_FLOATS);
  __syncthreads();

  // accumulators start from the pre-activations that the LDS-DMA delivers one step ahead; they are fetched from LDS
  // BEFORE the barrier that closes a step (they do not depend on h), so that only the h reads follow it
  f32x4v acc[8];
  auto preload_acc = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const float4 v = *reinterpret_cast<const float4*>(ps_wave + b * 256 + lane * 4);
      acc[b] = (f32x4v){v.x, v.y, v.z, v.w};
    }
  };
  preload_acc();

  f32x4v sv[SAVE ? 8 : 1];   // gates of the step just finished, block order b = 2*gate + half (SAVE only)
  auto save_piece = [&](int piece, int tpos) {   // piece 0..7: gate block, 8..9: cell state of half piece - 8
    const int64_t tile = pre16_tile_offset(d, st, tpos, nst16, g.len);
    if (piece < 8) {
      *reinterpret_cast<float4*>(tape_gates + tile + (int64_t)w * 2048 + piece * 256 + lane * 4) =
          make_float4(sv[piece][0], sv[piece][1], sv[piece][2], sv[piece][3]);
    } else {
      const int hf = piece - 8;
      *reinterpret_cast<float4*>(tape_c + tile / 4 + (int64_t)w * 512 + hf * 256 + lane * 4) =
          make_float4(cst[hf][0], cst[hf][1], cst[hf][2], cst[hf][3]);
    }
  };

  unsigned long long seg[4] = {0, 0, 0, 0};
  for (int step = 0; step < g.len; ++step) {
    unsigned long long c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    if (STAMP) c0 = __builtin_amdgcn_s_memtime();
    const int t = t0 + tdir * step;
    const float* hcur = Hs + (step & 1) * 16 * L16_LDH;
    float* hnext = Hs + ((step + 1) & 1) * 16 * L16_LDH;
    // (the accumulators were preloaded with this step's pre-activations before the barrier that opened the step)
    // the whole A operand (h_{t-1}, 16 x 128) in one batch, plus the two h_{t-1} rows this lane sends to HBM
    const float* arow = hcur + i16 * L16_LDH + 4 * ks;
    float4 afr[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) afr[m] = *reinterpret_cast<const float4*>(arow + 16 * m);
    float4 hs[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) hs[j] = *reinterpret_cast<const float4*>(hcur + (srow + 2 * j) * L16_LDH + scol);
#ifdef L16_EXPLICIT_WAIT
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    // accumulators in architectural VGPRs (the cell update reads them with plain VALU instructions).  Their LDS reads
    // were issued before the barrier: the wait the compiler puts here leaves the ten reads above in flight, and the
    // first MFMAs start as soon as THEIR fragment has arrived (waiting for all of them up front cost ~100 cycles/step)
#pragma unroll
    for (int b = 0; b < 8; ++b) asm volatile("" : "+v"(acc[b]));
    if (STAMP) c1 = __builtin_amdgcn_s_memtime();

    // branch-free: the last step re-requests its own tile; step 0 stores the zeros of h_{-1} at position t0 without
    // advancing and step 1 overwrites them (same lane, same address, program order)
    const float* pnext = pre_lane + (int64_t)(step + 1 < g.len ? t + tdir : t) * L16_TILE_FLOATS;
    const unsigned adv = step > 0 ? sstep : 0u;

    // cell update of unit half hf (lane-local, two accumulator slots per call) + publish h_t
    auto cell_half = [&](int hf) {
#pragma unroll
      for (int r = 0; r < 4; r += 2) {
        float h0, h1;
        if (DIAG & 8) {   // timing-only ablation: no transcendentals
          h0 = (acc[hf][r] + acc[2 + hf][r] + acc[4 + hf][r] + acc[6 + hf][r]) * 1e-3f;
          h1 = (acc[hf][r + 1] + acc[2 + hf][r + 1] + acc[4 + hf][r + 1] + acc[6 + hf][r + 1]) * 1e-3f;
        } else {
          const LstmCell2 u = lstm_cell2((f32x2){acc[hf][r], acc[hf][r + 1]}, (f32x2){acc[2 + hf][r], acc[2 + hf][r + 1]},
                                         (f32x2){acc[4 + hf][r], acc[4 + hf][r + 1]},
                                         (f32x2){acc[6 + hf][r], acc[6 + hf][r + 1]}, (f32x2){cst[hf][r], cst[hf][r + 1]});
          cst[hf][r] = u.c.x;
          cst[hf][r + 1] = u.c.y;
          if constexpr (SAVE) {
            sv[0 + hf][r] = u.i.x; sv[0 + hf][r + 1] = u.i.y;
            sv[2 + hf][r] = u.f.x; sv[2 + hf][r + 1] = u.f.y;
            sv[4 + hf][r] = u.g.x; sv[4 + hf][r + 1] = u.g.y;
            sv[6 + hf][r] = u.o.x; sv[6 + hf][r + 1] = u.o.y;
          }
          h0 = u.h.x;
          h1 = u.h.y;
        }
        hnext[(4 * ks + r) * L16_LDH + 32 * w + 16 * hf + i16] = h0;
        hnext[(4 * ks + r + 1) * L16_LDH + 32 * w + 16 * hf + i16] = h1;
      }
    };

    // h_{t-1} W_hh^T.  Unit half hf = the wave's units [32w + 16hf, +16): its four gate blocks finish before the other
    // half starts, so half 0's cell update can be spread between half 1's MFMAs (which keeps its dependent VALU
    // chains from stalling the MFMA stream; it is not free there, see the header).
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const float av[4] = {afr[m].x, afr[m].y, afr[m].z, afr[m].w};
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
#pragma unroll
          for (int gt = 0; gt < 4; ++gt) acc[2 * gt + hf] = mfma16(av[tt], wf[2 * gt + hf][4 * m + tt], acc[2 * gt + hf]);
          const int slot = 4 * m + tt;
          // one memory instruction per MFMA group (issued back to back they stall the wave on the memory pipeline's
          // queue): slots 0, 1 the stores of h_{t-1}, slots 2..9 the LDS-DMA requests
          if (hf == 0 && slot < 2) {
            // (the pin keeps the allocator from trying the full AGPR half for these load results and spilling them)
            asm volatile("" : "+v"(hs[slot].x), "+v"(hs[slot].y), "+v"(hs[slot].z), "+v"(hs[slot].w));
            if (RELU) hs[slot] = make_float4(relu1(hs[slot].x), relu1(hs[slot].y), relu1(hs[slot].z), relu1(hs[slot].w));
            if (!(DIAG & 2)) *reinterpret_cast<float4*>(hcb + soff[slot]) = hs[slot];
            soff[slot] += adv;
            __builtin_amdgcn_sched_barrier(0);
          } else if (hf == 0 && slot < 10) {
            if (!(DIAG & 1)) issue_pre_piece(pnext, slot - 2);
            __builtin_amdgcn_sched_barrier(0);
          } else if (SAVE && hf == 0 && slot < 20) {
            if (step > 0) save_piece(slot - 10, t - tdir);   // the previous step's gates / cell state
            __builtin_amdgcn_sched_barrier(0);
          } else if (hf == 1) {
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x006, 5, 0);
          }
        }
      }
      if (hf == 0) cell_half(0);
    }
    if (STAMP) {
      asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::"v"(acc[1][0]), "v"(acc[7][3]));
      c2 = __builtin_amdgcn_s_memtime();
    }
    cell_half(1);
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
  if constexpr (SAVE) {
#pragma unroll
    for (int piece = 0; piece < 10; ++piece) save_piece(piece, t0 + tdir * (g.len - 1));
  }
  // h of the last step: the barrier above published it in buffer (len & 1)
  {
    const float* hfin = Hs + (g.len & 1) * 16 * L16_LDH;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float4 v = *reinterpret_cast<const float4*>(hfin + (srow + 2 * j) * L16_LDH + scol);
      if (RELU) v = make_float4(relu1(v.x), relu1(v.y), relu1(v.z), relu1(v.w));
      *reinterpret_cast<float4*>(hcb + soff[j]) = v;
    }
  }
  if (STAMP && lane == 0) {
    unsigned long long* o = stamps + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + w) * 4;
    o[0] = seg[0]; o[1] = seg[1]; o[2] = seg[2]; o[3] = seg[3];
  }
}

int lstm16_launch(int variant, bool relu, int nst16, int ndir, void* stream, const float* pre, const float* whh_f,
                  const float* whh_b, float* hc, int ldh, int dump_row, const SeqGeom& g, unsigned long long* stamps,
                  float* tape_gates, float* tape_c) {
  using Kern = void (*)(const float*, const float*, const float*, float*, int, int, SeqGeom, int, unsigned long long*, float*,
                        float*);
  Kern kern;
  switch (variant) {
    case 0: kern = relu ? lstm16_kernel<false, true, 0> : lstm16_kernel<false, false, 0>; break;
    case 1: kern = relu ? lstm16_kernel<true, true, 0> : lstm16_kernel<true, false, 0>; break;
    case 2: kern = lstm16_kernel<true, true, 1>; break;    // no LDS-DMA
    case 3: kern = lstm16_kernel<true, true, 2>; break;    // no stores
    case 4: kern = lstm16_kernel<true, true, 3>; break;    // neither
    case 5: kern = lstm16_kernel<true, true, 8>; break;    // identity activations
    case 6: kern = lstm16_kernel<true, true, 11>; break;   // bare MFMA stream
    case 7: kern = lstm16_kernel<false, false, 0, true>; break;   // training forward: raw h + tape
    default: return (int)hipErrorInvalidValue;
  }
  if (variant == 7 && (!tape_gates || !tape_c)) return (int)hipErrorInvalidValue;
  static PerDeviceOnce ready[8][2];
  const int dev = current_hip_device();
  if (!ready[variant][relu].done(dev)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)L16_LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    ready[variant][relu].set(dev);
  }
  hipLaunchKernelGGL(kern, dim3(nst16, ndir), dim3(256), L16_LDS_BYTES, static_cast<hipStream_t>(stream), pre, whh_f, whh_b,
                     hc, ldh, dump_row, g, nst16, stamps, tape_gates, tape_c);
  return (int)hipGetLastError();
}
