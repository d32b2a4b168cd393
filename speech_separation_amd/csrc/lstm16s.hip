// lstm16s.hip -- OPT-IN split-precision variant of the 16-sequence-tile LSTM recurrence (option "split_bf16"; the
// default and every parity claim stay on the fp32 kernel of lstm16.hip).
//
// fp32 MFMA runs at the vector rate (v_mfma_f32_16x16x4_f32: 32 cycles for 2 kFLOP); v_mfma_f32_16x16x32_bf16 does
// 16 kFLOP in 16 cycles.  Each fp32 operand is split into two bf16 halves, x = hi + lo with hi = bf16(x),
// lo = bf16(x - hi) (16 significant bits together), and the product h W^T is formed as hi*hi + hi*lo + lo*hi with fp32
// accumulation: 3 bf16 MFMAs replace 8 fp32 ones, ~5x less matrix time at ~2^-17 relative error per product (fp32:
// 2^-24).  W_hh costs the same 256 registers per lane as in fp32 (two bf16 per register).  Everything else is
// lstm16.hip's design: one workgroup = one direction x 16 sequences, W_hh resident in AGPRs as B fragments,
// pre-activations (fp32, PRE16 layout, pre-scaled), lane-local cell update in fp32, one barrier per step.  The split
// step takes ~2 us -- about one HBM round trip -- so the pre-activations are fetched TWO steps ahead, straight into the
// registers that will be that step's accumulators (the PRE16 layout hands every lane its own 16 bytes; three accumulator
// sets rotate, the step loop is unrolled by three): the compiler then counts the outstanding loads exactly, whereas any
// LDS access behind an LDS-DMA makes it wait for ALL of them (s_waitcnt vmcnt(0)), which pins the prefetch distance of
// lstm16.hip's scheme to one step.  h_t is exchanged through LDS three times over: fp32 rows (they leave for HBM as 16-byte
// row pieces, full precision) and the bf16 hi / lo images the next step's A fragments are read from (ds_read_b128 =
// 8 consecutive k of one sequence row = one fragment of v_mfma_f32_16x16x32_bf16: lane l holds A[row l&15][8(l>>4)+j]).
#include <hip/hip_runtime.h>

#include <type_traits>

#include "lstm.h"
#include "lstm16.h"

typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int LDB = L16_H + 8;                         // bf16 row stride (272 bytes): conflict-free ds_read_b128 fragments
constexpr int HB_ELEMS = 2 * 16 * LDB;                 // one bf16 image, double buffered


DEV f32x4v mfma_bf16(bf16x8 a, bf16x8 b, f32x4v c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

template <int OFF>
DEV void glds16_off(const float* gsrc, float* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, OFF, 0);
}
DEV float relu1(float x) { return __int_as_float(max(__float_as_int(x), 0)); }

template <bool RELU>
__global__ __launch_bounds__(256) void lstm16s_kernel(const float* __restrict__ pre, const float* __restrict__ whh_f,
                                                      const float* __restrict__ whh_b, float* __restrict__ hc, int ldh,
                                                      int dump_row, SeqGeom g, int nst16) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Hs = smem;                                              // fp32 [2][16][L16_LDH]
  __bf16* Hhi = reinterpret_cast<__bf16*>(smem + L16_HS_FLOATS);   // [2][16][LDB]
  __bf16* Hlo = Hhi + HB_ELEMS;
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, i16 = lane & 15, ks = lane >> 4;
  const int st = blockIdx.x, d = blockIdx.y;
  const float* whh = d ? whh_b : whh_f;

  // W_hh slice -> bf16 hi / lo B fragments (block b = 2*gate + half, k-chunk m: k = 32m + 8ks + j), pre-scaled per gate
  bf16x8 whi[8][4], wlo[8][4];
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const float* wrow = whh + (int64_t)((b >> 1) * L16_H + 32 * w + 16 * (b & 1) + i16) * L16_H + 8 * ks;
    const float gs = l16_gate_scale(b >> 1);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const float4 v0 = *reinterpret_cast<const float4*>(wrow + 32 * m), v1 = *reinterpret_cast<const float4*>(wrow + 32 * m + 4);
      const float x[8] = {v0.x * gs, v0.y * gs, v0.z * gs, v0.w * gs, v1.x * gs, v1.y * gs, v1.z * gs, v1.w * gs};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const __bf16 hi = (__bf16)x[j];
        whi[b][m][j] = hi;
        wlo[b][m][j] = (__bf16)(x[j] - (float)hi);
      }
    }
  }
  // parked in the AGPR half of the register file; the MFMAs read their B operand there
#pragma unroll
  for (int b = 0; b < 8; ++b)
#pragma unroll
    for (int m = 0; m < 4; ++m) asm volatile("" : "+a"(whi[b][m]), "+a"(wlo[b][m]));

  const int t0 = d ? g.len - 1 : 0, tdir = d ? -1 : 1;
  const int tstride = seq_token_stride(g);
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

  for (int i = tid; i < 16 * L16_LDH; i += 256) Hs[i] = 0.f;                          // h_{-1} = 0 (buffer 0)
  for (int i = tid; i < 16 * LDB; i += 256) { Hhi[i] = (__bf16)0.f; Hlo[i] = (__bf16)0.f; }
  f32x4v cst[2] = {(f32x4v){0.f, 0.f, 0.f, 0.f}, (f32x4v){0.f, 0.f, 0.f, 0.f}};

  const float* pre_lane = pre + pre16_tile_offset(d, st, 0, nst16, g.len) + (int64_t)w * 2048 + lane * 4;
  auto tile_of = [&](int step) { return pre_lane + (int64_t)(t0 + tdir * (step < g.len ? step : g.len - 1)) * L16_TILE_FLOATS; };
  // three accumulator sets: step s accumulates in X[s % 3]; the loads of step s + 2 land in X[(s + 2) % 3]
  f32x4v X[3][8];
  auto fetch_piece = [&](f32x4v (&dst)[8], const float* tile, int b) {
    const float4 v = *reinterpret_cast<const float4*>(tile + b * 256);
    dst[b] = (f32x4v){v.x, v.y, v.z, v.w};
  };
#pragma unroll
  for (int b = 0; b < 8; ++b) fetch_piece(X[0], tile_of(0), b);
#pragma unroll
  for (int b = 0; b < 8; ++b) fetch_piece(X[1], tile_of(1), b);
  __syncthreads();

  auto step_body = [&](auto ring, int step) {
    constexpr int R = decltype(ring)::value;
    f32x4v (&acc)[8] = X[R];
    f32x4v (&nxt2)[8] = X[(R + 2) % 3];
    const int cur = step & 1, nxt = cur ^ 1;
    const float* hcur = Hs + cur * 16 * L16_LDH;
    float* hnext = Hs + nxt * 16 * L16_LDH;
    __bf16* hinext = Hhi + nxt * 16 * LDB;
    __bf16* lonext = Hlo + nxt * 16 * LDB;
    // the whole A operand (h_{t-1}: 16 x 128 as hi and lo fragments) in one batch + the two fp32 row pieces of h_{t-1}
    // this lane sends to HBM
    bf16x8 ahi[4], alo[4];
    {
      const __bf16* ar = Hhi + cur * 16 * LDB + i16 * LDB + 8 * ks;
      const __bf16* br = Hlo + cur * 16 * LDB + i16 * LDB + 8 * ks;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        ahi[m] = *reinterpret_cast<const bf16x8*>(ar + 32 * m);
        alo[m] = *reinterpret_cast<const bf16x8*>(br + 32 * m);
      }
    }
    float4 hs[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) hs[j] = *reinterpret_cast<const float4*>(hcur + (srow + 2 * j) * L16_LDH + scol);
    // (without this pin the allocator tries the full AGPR half for these load results and spills them)
#pragma unroll
    for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(hs[j].x), "+v"(hs[j].y), "+v"(hs[j].z), "+v"(hs[j].w));
    // branch-free: beyond the last step the requests re-read the last tile into a set nobody uses any more
    const float* pnext = tile_of(step + 2);
    const unsigned adv = step > 0 ? sstep : 0u;

    auto cell_half = [&](int hf) {
#pragma unroll
      for (int r = 0; r < 4; r += 2) {
        const LstmCell2 u = lstm_cell2((f32x2){acc[hf][r], acc[hf][r + 1]}, (f32x2){acc[2 + hf][r], acc[2 + hf][r + 1]},
                                       (f32x2){acc[4 + hf][r], acc[4 + hf][r + 1]},
                                       (f32x2){acc[6 + hf][r], acc[6 + hf][r + 1]}, (f32x2){cst[hf][r], cst[hf][r + 1]});
        cst[hf][r] = u.c.x;
        cst[hf][r + 1] = u.c.y;
        const float hv[2] = {u.h.x, u.h.y};
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int row = 4 * ks + r + e, col = 32 * w + 16 * hf + i16;
          const __bf16 hi = (__bf16)hv[e];
          hnext[row * L16_LDH + col] = hv[e];
          hinext[row * LDB + col] = hi;
          lonext[row * LDB + col] = (__bf16)(hv[e] - (float)hi);
        }
      }
    };

#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
#pragma unroll
        for (int gt = 0; gt < 4; ++gt) {
          const int b = 2 * gt + hf;
          acc[b] = mfma_bf16(ahi[m], whi[b][m], acc[b]);
          acc[b] = mfma_bf16(ahi[m], wlo[b][m], acc[b]);
          acc[b] = mfma_bf16(alo[m], whi[b][m], acc[b]);
          // one memory instruction per MFMA group: the two row stores of h_{t-1}, then the eight pre-activation loads of
          // step + 2
          const int slot = 4 * m + gt;
          if (hf == 0 && slot < 2) {
            if (RELU) hs[slot] = make_float4(relu1(hs[slot].x), relu1(hs[slot].y), relu1(hs[slot].z), relu1(hs[slot].w));
            *reinterpret_cast<float4*>(hcb + soff[slot]) = hs[slot];
            soff[slot] += adv;
          } else if (hf == 0 && slot < 10) {
            fetch_piece(nxt2, pnext, slot - 2);
          }
        }
      }
      if (hf == 0) cell_half(0);
    }
    cell_half(1);
    __syncthreads();
  };
  for (int step = 0; step < g.len;) {
    step_body(std::integral_constant<int, 0>{}, step);
    if (++step >= g.len) break;
    step_body(std::integral_constant<int, 1>{}, step);
    if (++step >= g.len) break;
    step_body(std::integral_constant<int, 2>{}, step);
    ++step;
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
}

// ---------------------------------------------------------------------------------------------------------------------
// 32-sequence tiles (whole-batch launches, DPRNN / long utterances: config 5), same split-precision scheme on
// v_mfma_f32_32x32x16_bf16.  Structure of lstm.hip: wave w = hidden units [32w, 32w+32) of all four gates (4 accumulator
// tiles, rows = sequences), W_hh as 4 x 8 hi / lo B fragments in the 256 AGPRs (lane (c,hh): W[gate*128 + 32w + c]
// [16m + 8hh + j]), pre-activations (fp32, PRE layout of lstm.h) by LDS-DMA one step ahead, h_t through LDS as fp32 rows
// (for HBM) + bf16 hi / lo images (A fragments: ds_read_b128 = 8 consecutive k of one sequence row).
constexpr int LDB32 = LSTM_H + 8;
constexpr int HB32_ELEMS = 2 * 32 * LDB32;
constexpr size_t LDS32_BYTES = LSTM_LDS_BYTES + sizeof(__bf16) * 2 * HB32_ELEMS;

DEV f32x16 mfma3(const bf16x8& ah, const bf16x8& al, const bf16x8& bh, const bf16x8& bl, f32x16 c) {
  c = mfma32_bf16(ah, bh, c);
  c = mfma32_bf16(ah, bl, c);
  return mfma32_bf16(al, bh, c);
}

template <bool RELU>
__global__ __launch_bounds__(256) void lstm32s_kernel(const float* __restrict__ pre, const float* __restrict__ whh_f,
                                                      const float* __restrict__ whh_b, float* __restrict__ hc, int ldh,
                                                      int dump_row, SeqGeom g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Hs = smem;                      // fp32 [2][32][LSTM_LDH]
  float* Ps = smem + LSTM_HS_FLOATS;     // [4 waves][16 pieces][64 lanes][4]
  __bf16* Hhi = reinterpret_cast<__bf16*>(smem + LSTM_HS_FLOATS + LSTM_PRE_FLOATS);   // [2][32][LDB32]
  __bf16* Hlo = Hhi + HB32_ELEMS;
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, c = lane & 31, hh = lane >> 5;
  const int st = blockIdx.x, d = blockIdx.y;
  const float* whh = d ? whh_b : whh_f;

  bf16x8 whi[4][8], wlo[4][8];
#pragma unroll
  for (int gi = 0; gi < 4; ++gi) {
    const float* wrow = whh + (int64_t)(gi * LSTM_H + 32 * w + c) * LSTM_H + 8 * hh;
    const float gs = lstm_gate_scale(gi);
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const float4 v0 = *reinterpret_cast<const float4*>(wrow + 16 * m), v1 = *reinterpret_cast<const float4*>(wrow + 16 * m + 4);
      const float x[8] = {v0.x * gs, v0.y * gs, v0.z * gs, v0.w * gs, v1.x * gs, v1.y * gs, v1.z * gs, v1.w * gs};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const __bf16 hi = (__bf16)x[j];
        whi[gi][m][j] = hi;
        wlo[gi][m][j] = (__bf16)(x[j] - (float)hi);
      }
    }
  }
#pragma unroll
  for (int gi = 0; gi < 4; ++gi)
#pragma unroll
    for (int m = 0; m < 8; ++m) asm volatile("" : "+a"(whi[gi][m]), "+a"(wlo[gi][m]));

  const int t0 = d ? g.len - 1 : 0;
  const int tdir = d ? -1 : 1;
  const int tstride = seq_token_stride(g);
  const int srow = 8 * w + hh, scol = 4 * c;
  const int64_t sstep = (int64_t)tdir * tstride * ldh;
  float* sp[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int q = st * 32 + srow + 2 * j;
    const int64_t tokb = q < g.nseq ? seq_token_base(g, q) : (int64_t)dump_row;
    sp[j] = hc + (tokb + (int64_t)t0 * tstride) * ldh + d * LSTM_H + scol;
  }
  for (int i = tid; i < 32 * LSTM_LDH; i += 256) Hs[i] = 0.f;
  for (int i = tid; i < 32 * LDB32; i += 256) { Hhi[i] = (__bf16)0.f; Hlo[i] = (__bf16)0.f; }
  f32x16 cst = zero16();

  const float* pre_lane = pre + pre_tile_offset(d, st, 0, g.nst, g.len) + (int64_t)w * 1024 + lane * 4;
  float* ps_wave = Ps + w * (16 * 256);
  auto issue_pre_piece = [&](const float* p, int piece) {   // piece = gi*4 + q (compile-time after unrolling)
    const int gi = piece >> 2;
    switch (piece & 3) {
      case 0: glds16_off<0>(p + gi * 4096, ps_wave + gi * 1024); break;
      case 1: glds16_off<1024>(p + gi * 4096, ps_wave + gi * 1024); break;
      case 2: glds16_off<2048>(p + gi * 4096, ps_wave + gi * 1024); break;
      default: glds16_off<3072>(p + gi * 4096, ps_wave + gi * 1024); break;
    }
  };
#pragma unroll
  for (int piece = 0; piece < 16; ++piece) issue_pre_piece(pre_lane + (int64_t)t0 * (512 * 32), piece);
  __syncthreads();

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

  for (int step = 0; step < g.len; ++step) {
    const int t = t0 + tdir * step;
    const int cur = step & 1, nxt = cur ^ 1;
    const float* hcur = Hs + cur * 32 * LSTM_LDH;
    float* hnext = Hs + nxt * 32 * LSTM_LDH;
    __bf16* hinext = Hhi + nxt * 32 * LDB32;
    __bf16* lonext = Hlo + nxt * 32 * LDB32;
    bf16x8 ahi[8], alo[8];
    {
      const __bf16* ar = Hhi + cur * 32 * LDB32 + c * LDB32 + 8 * hh;
      const __bf16* br = Hlo + cur * 32 * LDB32 + c * LDB32 + 8 * hh;
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        ahi[m] = *reinterpret_cast<const bf16x8*>(ar + 16 * m);
        alo[m] = *reinterpret_cast<const bf16x8*>(br + 16 * m);
      }
    }
    float4 hs[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) hs[j] = *reinterpret_cast<const float4*>(hcur + (srow + 2 * j) * LSTM_LDH + scol);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int gi = 0; gi < 4; ++gi) asm volatile("" : "+v"(acc[gi]));
    // (without this pin the allocator tries the full AGPR half for these load results and spills them)
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(hs[j].x), "+v"(hs[j].y), "+v"(hs[j].z), "+v"(hs[j].w));
    const float* pnext = pre_lane + (int64_t)(step + 1 < g.len ? t + tdir : t) * (512 * 32);
    const int64_t adv = step > 0 ? sstep : 0;
    if (RELU) {
#pragma unroll
      for (int j = 0; j < 4; ++j) hs[j] = make_float4(relu1(hs[j].x), relu1(hs[j].y), relu1(hs[j].z), relu1(hs[j].w));
    }
#pragma unroll
    for (int m = 0; m < 8; ++m) {
#pragma unroll
      for (int gi = 0; gi < 4; ++gi) {
        acc[gi] = mfma3(ahi[m], alo[m], whi[gi][m], wlo[gi][m], acc[gi]);
        // one memory instruction per MFMA group: the four row stores of h_{t-1}, then the 16 LDS-DMA requests
        const int slot = 4 * m + gi;
        if (slot < 4) {
          *reinterpret_cast<float4*>(sp[slot]) = hs[slot];
          sp[slot] += adv;
        } else if (slot < 20) {
          issue_pre_piece(pnext, slot - 4);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const LstmCell2 u = lstm_cell2((f32x2){acc[0][r], acc[0][r + 1]}, (f32x2){acc[1][r], acc[1][r + 1]},
                                     (f32x2){acc[2][r], acc[2][r + 1]}, (f32x2){acc[3][r], acc[3][r + 1]},
                                     (f32x2){cst[r], cst[r + 1]});
      cst[r] = u.c.x;
      cst[r + 1] = u.c.y;
      const float hv[2] = {u.h.x, u.h.y};
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int row = ROW32(r + e, hh), col = 32 * w + c;
        const __bf16 hi = (__bf16)hv[e];
        hnext[row * LSTM_LDH + col] = hv[e];
        hinext[row * LDB32 + col] = hi;
        lonext[row * LDB32 + col] = (__bf16)(hv[e] - (float)hi);
      }
    }
    if (step + 1 < g.len) preload_acc();
    __syncthreads();
  }
  {
    const float* hfin = Hs + (g.len & 1) * 32 * LSTM_LDH;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float4 v = *reinterpret_cast<const float4*>(hfin + (srow + 2 * j) * LSTM_LDH + scol);
      if (RELU) v = make_float4(relu1(v.x), relu1(v.y), relu1(v.z), relu1(v.w));
      *reinterpret_cast<float4*>(sp[j]) = v;
    }
  }
}

constexpr size_t LDS_BYTES = sizeof(float) * L16_HS_FLOATS + sizeof(__bf16) * 2 * HB_ELEMS;

}  // namespace

int lstm16s_launch(bool relu, int nst16, int ndir, void* stream, const float* pre, const float* whh_f, const float* whh_b,
                   float* hc, int ldh, int dump_row, const SeqGeom& g) {
  auto kern = relu ? lstm16s_kernel<true> : lstm16s_kernel<false>;
  static PerDeviceOnce ready[2];
  const int dev = current_hip_device();
  if (!ready[relu].done(dev)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    ready[relu].set(dev);
  }
  hipLaunchKernelGGL(kern, dim3(nst16, ndir), dim3(256), LDS_BYTES, static_cast<hipStream_t>(stream), pre, whh_f, whh_b, hc, ldh,
                     dump_row, g, nst16);
  return (int)hipGetLastError();
}

int lstm32s_launch(bool relu, int nst, int ndir, void* stream, const float* pre, const float* whh_f, const float* whh_b, float* hc,
                   int ldh, int dump_row, const SeqGeom& g) {
  auto kern = relu ? lstm32s_kernel<true> : lstm32s_kernel<false>;
  static PerDeviceOnce ready[2];
  const int dev = current_hip_device();
  if (!ready[relu].done(dev)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)LDS32_BYTES);
    if (e != hipSuccess) return (int)e;
    ready[relu].set(dev);
  }
  hipLaunchKernelGGL(kern, dim3(nst, ndir), dim3(256), LDS32_BYTES, static_cast<hipStream_t>(stream), pre, whh_f, whh_b, hc, ldh,
                     dump_row, g);
  return (int)hipGetLastError();
}
