// lstm16x.hip -- LSTM recurrence on 16-sequence tiles WITH the input projection inside (num_features = 64), gfx950.
//
// Reference semantics: torch.nn.LSTM(batch_first=True), gate order i, f, g, o (src/model/dptn.py:22-30,
// src/model/dprnn.py:24-47):   gates_t = x_t W_ih^T + b_ih + h_{t-1} W_hh^T + b_hh.
//
// lstm16.hip takes `x_t W_ih^T + b` from HBM (PRE16, written by the K4 GEMM): 2 KiB per token and direction written and
// read again -- for 64 features an 8x expansion of the 256-byte input row, and 80 % of a forward's HBM traffic.  Here the
// pre-activations never exist in memory: one workgroup = one direction x 16 sequences x all positions keeps
//   * W_hh (512 x 128) in the AGPR half of its register file, as lstm16.hip does, and
//   * W_ih (512 x 64 = 128 KiB) in LDS, in MFMA B-fragment order (each wave its own 32 KiB: 8 blocks x 4 k-chunks),
// reads the step's 16 x 64 input rows (4 KiB, one float4 per thread, requested three steps ahead) and accumulates
//   acc = b + x_t W_ih^T  (128 MFMAs per wave, issued at the END of the previous step: they do not depend on h_{t-1})
//   acc += h_{t-1} W_hh^T (256 MFMAs per wave)
// in one chain.  The recurrence gets 50 % more matrix work per step; the K4 launch (as many FLOPs, plus 4 KiB of stores per
// token) disappears.  Fragment maps, cell update and the h exchange are lstm16.hip's (v_mfma_f32_16x16x4_f32:
// lane l: i16 = l & 15, ks = l >> 4; A[i16][ks], B[ks][i16], D reg r = (row 4ks + r, col i16); MFMA step 4m+t uses true
// k = 16m + 4ks + t).  Compiled with -mllvm -amdgpu-mfma-vgpr-form (build.py).
#include <hip/hip_runtime.h>

#include "lstm16.h"

typedef float f32x4v __attribute__((ext_vector_type(4)));
static __device__ __forceinline__ f32x4v mfma16(float a, float b, f32x4v c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
static __device__ __forceinline__ float relu1(float x) { return __int_as_float(max(__float_as_int(x), 0)); }

namespace {

template <int NIN>
struct X16 {
  static constexpr int MK = NIN / 16;                   // k-chunks of the input projection
  static constexpr int LDX = NIN + 8;                   // input rows in LDS: conflict-free ds_read_b128 for the 16x16x4 A map
  static constexpr int WL_FLOATS = 4 * MK * 8 * 256;    // [wave][m][block][lane][4]
  static constexpr int XS_FLOATS = 2 * 16 * LDX;        // two steps of input rows
  static constexpr int NXL = (16 * NIN / 4) / 256;      // float4 of an input tile per thread
  static constexpr size_t LDS_BYTES = sizeof(float) * (WL_FLOATS + L16_HS_FLOATS + XS_FLOATS);
  static_assert((16 * NIN / 4) % 256 == 0 && NIN % 16 == 0, "input tile / threads");
  static_assert(LDS_BYTES <= 160 * 1024, "W_ih must fit the LDS beside the h and x tiles");
};

template <int NIN, bool RELU>
__global__ __launch_bounds__(256) void lstm16x_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ wih_f,
                                                       const float* __restrict__ wih_b, const float* __restrict__ bih_f,
                                                       const float* __restrict__ bih_b, const float* __restrict__ bhh_f,
                                                       const float* __restrict__ bhh_b, const float* __restrict__ whh_f,
                                                       const float* __restrict__ whh_b, float* __restrict__ hc, int ldh,
                                                       int64_t dump_row, SeqGeom g) {
  using Sh = X16<NIN>;
  constexpr int MK = Sh::MK, LDX = Sh::LDX, NXL = Sh::NXL;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Wl = smem;                               // [4][MK][8][256]
  float* Hs = smem + Sh::WL_FLOATS;               // [2][16][L16_LDH]
  float* Xs = Hs + L16_HS_FLOATS;                 // [2][16][LDX]
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, i16 = lane & 15, ks = lane >> 4;
  const int st = blockIdx.x, d = blockIdx.y;
  const float* whh = d ? whh_b : whh_f;
  const float* wih = d ? wih_b : wih_f;
  const float* bih = d ? bih_b : bih_f;
  const float* bhh = d ? bhh_b : bhh_f;

  // ---- W_hh -> AGPRs (B fragments of the recurrent product), pre-scaled for the exp2 forms of sigmoid / tanh ----------
  float wf[8][32];
  float bsc[8];                                   // (b_ih + b_hh) of the lane's column in block b, pre-scaled
  float* wl_wave = Wl + w * (MK * 8 * 256) + lane * 4;
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const int row = (b >> 1) * L16_H + 32 * w + 16 * (b & 1) + i16;
    const float gs = l16_gate_scale(b >> 1);
    const float* wrow = whh + (int64_t)row * L16_H + 4 * ks;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const float4 v = *reinterpret_cast<const float4*>(wrow + 16 * m);
      wf[b][4 * m + 0] = v.x * gs;
      wf[b][4 * m + 1] = v.y * gs;
      wf[b][4 * m + 2] = v.z * gs;
      wf[b][4 * m + 3] = v.w * gs;
    }
    bsc[b] = (bih[row] + bhh[row]) * gs;
    // ---- W_ih -> LDS, the same fragment map (this lane writes exactly what it will read) ----------------------------------
    const float* irow = wih + (int64_t)row * NIN + 4 * ks;
#pragma unroll
    for (int m = 0; m < MK; ++m) {
      const float4 v = *reinterpret_cast<const float4*>(irow + 16 * m);
      *reinterpret_cast<float4*>(wl_wave + (m * 8 + b) * 256) = make_float4(v.x * gs, v.y * gs, v.z * gs, v.w * gs);
    }
  }
#pragma unroll
  for (int b = 0; b < 8; ++b)
#pragma unroll
    for (int i = 0; i < 32; ++i) asm volatile("" : "+a"(wf[b][i]));

  const int t0 = d ? g.len - 1 : 0, tdir = d ? -1 : 1;
  const int tstride = seq_token_stride(g);
  // ---- h rows leave through the LDS tile (lstm16.hip): wave w stores tile rows 4w + 2j + (lane >> 5), 16 bytes at column
  // 4 (lane & 31).  Rows of padded sequences walk through the dump rows.  64-bit addresses: a DPRNN launch of 16 mixtures
  // x 8 s is 4 M tokens x 1 KiB.
  const int srow = 4 * w + (lane >> 5), scol = 4 * (lane & 31);
  char* hp[2];
  const int64_t hstep = (int64_t)tdir * tstride * ldh * 4;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int q = st * 16 + srow + 2 * j;
    const int64_t tokb = q < g.nseq ? seq_token_base(g, q) : dump_row;
    hp[j] = reinterpret_cast<char*>(hc) + ((tokb + (int64_t)t0 * tstride) * ldh + d * L16_H + scol) * 4;
  }
  // ---- input rows: thread -> (row, 16-byte chunk) of the 16 x NIN tile; padded sequences read the last real one ------------
  const char* xp[NXL];
  int xoff[NXL];                                  // float offset inside an Xs buffer
  const int64_t xstep = (int64_t)tdir * tstride * ldx * 4;
#pragma unroll
  for (int i = 0; i < NXL; ++i) {
    const int idx = i * 256 + tid, row = idx / (NIN / 4), ch = idx % (NIN / 4);
    const int q = st * 16 + row;
    const int64_t tokb = seq_token_base(g, q < g.nseq ? q : g.nseq - 1);
    xp[i] = reinterpret_cast<const char*>(x) + ((tokb + (int64_t)t0 * tstride) * ldx + 4 * ch) * 4;
    xoff[i] = row * LDX + 4 * ch;
  }
  float4 xr[NXL];
  // x_0 -> Xs[0], x_1 -> Xs[1], x_2 -> registers (positions past the end re-read the last one: never used)
  auto load_x = [&](int s) {
    const int64_t adv = (int64_t)(s < g.len ? s : g.len - 1) * xstep;
#pragma unroll
    for (int i = 0; i < NXL; ++i) xr[i] = *reinterpret_cast<const float4*>(xp[i] + adv);
  };
  auto stage_x = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NXL; ++i) *reinterpret_cast<float4*>(Xs + buf * 16 * LDX + xoff[i]) = xr[i];
  };
  load_x(0);
  stage_x(0);
  load_x(1);
  stage_x(1);
  load_x(2);
  for (int i = tid; i < 16 * L16_LDH; i += 256) Hs[i] = 0.f;      // h_{-1} = 0 (buffer 0)
  f32x4v cst[2] = {(f32x4v){0.f, 0.f, 0.f, 0.f}, (f32x4v){0.f, 0.f, 0.f, 0.f}};
  __syncthreads();

  // ---- acc = b + x W_ih^T of one step, from the staged rows (all of a wave's B fragments come from its own LDS slice) -----
  f32x4v acc[8];
  auto input_part = [&](int buf) {
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[b] = (f32x4v){bsc[b], bsc[b], bsc[b], bsc[b]};
    const float* xrow = Xs + buf * 16 * LDX + i16 * LDX + 4 * ks;
    float4 xa[MK];
#pragma unroll
    for (int m = 0; m < MK; ++m) xa[m] = *reinterpret_cast<const float4*>(xrow + 16 * m);
    float4 wcur[8];
#pragma unroll
    for (int b = 0; b < 8; ++b) wcur[b] = *reinterpret_cast<const float4*>(wl_wave + b * 256);
#pragma unroll
    for (int m = 0; m < MK; ++m) {
      float4 wnext[8];
      if (m + 1 < MK) {
#pragma unroll
        for (int b = 0; b < 8; ++b) wnext[b] = *reinterpret_cast<const float4*>(wl_wave + ((m + 1) * 8 + b) * 256);
      }
      const float av[4] = {xa[m].x, xa[m].y, xa[m].z, xa[m].w};
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
#pragma unroll
        for (int b = 0; b < 8; ++b) {
          const float bv[4] = {wcur[b].x, wcur[b].y, wcur[b].z, wcur[b].w};
          acc[b] = mfma16(av[tt], bv[tt], acc[b]);
        }
      }
      if (m + 1 < MK) {
#pragma unroll
        for (int b = 0; b < 8; ++b) wcur[b] = wnext[b];
      }
    }
  };
  input_part(0);
  __syncthreads();      // step 0 overwrites the buffer of x_0 with x_2

  for (int step = 0; step < g.len; ++step) {
    const float* hcur = Hs + (step & 1) * 16 * L16_LDH;
    float* hnext = Hs + ((step + 1) & 1) * 16 * L16_LDH;
    const float* arow = hcur + i16 * L16_LDH + 4 * ks;
    float4 afr[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) afr[m] = *reinterpret_cast<const float4*>(arow + 16 * m);
    float4 hs[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) hs[j] = *reinterpret_cast<const float4*>(hcur + (srow + 2 * j) * L16_LDH + scol);
#pragma unroll
    for (int b = 0; b < 8; ++b) asm volatile("" : "+v"(acc[b]));
    // step 0 stores the zeros of h_{-1} at position t0 without advancing and step 1 overwrites them
    const int64_t adv = step > 0 ? hstep : 0;

    auto cell_half = [&](int hf) {
#pragma unroll
      for (int r = 0; r < 4; r += 2) {
        const LstmCell2 u = lstm_cell2((f32x2){acc[hf][r], acc[hf][r + 1]}, (f32x2){acc[2 + hf][r], acc[2 + hf][r + 1]},
                                       (f32x2){acc[4 + hf][r], acc[4 + hf][r + 1]}, (f32x2){acc[6 + hf][r], acc[6 + hf][r + 1]},
                                       (f32x2){cst[hf][r], cst[hf][r + 1]});
        cst[hf][r] = u.c.x;
        cst[hf][r + 1] = u.c.y;
        hnext[(4 * ks + r) * L16_LDH + 32 * w + 16 * hf + i16] = u.h.x;
        hnext[(4 * ks + r + 1) * L16_LDH + 32 * w + 16 * hf + i16] = u.h.y;
      }
    };

    // h_{t-1} W_hh^T, unit half by unit half (lstm16.hip); one memory instruction per MFMA group in the first groups:
    // slots 0, 1 the stores of h_{t-1}; then the input rows of step + 2 go from registers to LDS (they are read at the end of
    // step + 1, behind the barrier that closes this step) and the rows of step + 3 are requested
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
          if (hf == 0 && slot < 2) {
            asm volatile("" : "+v"(hs[slot].x), "+v"(hs[slot].y), "+v"(hs[slot].z), "+v"(hs[slot].w));
            if (RELU) hs[slot] = make_float4(relu1(hs[slot].x), relu1(hs[slot].y), relu1(hs[slot].z), relu1(hs[slot].w));
            *reinterpret_cast<float4*>(hp[slot]) = hs[slot];
            hp[slot] += adv;
            __builtin_amdgcn_sched_barrier(0);
          } else if (hf == 0 && slot == 2) {
            stage_x(step & 1);                    // x_{step+2}: buffer of x_step, last read before the previous barrier
            __builtin_amdgcn_sched_barrier(0);
          } else if (hf == 0 && slot == 3) {
            load_x(step + 3);
            __builtin_amdgcn_sched_barrier(0);
          } else if (hf == 1) {
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x006, 5, 0);
          }
        }
      }
      if (hf == 0) cell_half(0);
    }
    cell_half(1);
    // the next step's input part: independent of h_t, so it runs in front of the barrier the wave would otherwise wait at
    if (step + 1 < g.len) input_part((step + 1) & 1);
    __syncthreads();
  }
  // h of the last step: the barrier above published it in buffer (len & 1)
  {
    const float* hfin = Hs + (g.len & 1) * 16 * L16_LDH;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float4 v = *reinterpret_cast<const float4*>(hfin + (srow + 2 * j) * L16_LDH + scol);
      if (RELU) v = make_float4(relu1(v.x), relu1(v.y), relu1(v.z), relu1(v.w));
      *reinterpret_cast<float4*>(hp[j]) = v;
    }
  }
}

// =====================================================================================================================
// num_features = 128: W_ih (512 x 128) is as large as W_hh and the two do not fit one CU's registers together -- W_hh fills
// the AGPR half (256 registers per lane), the LDS has room for 31 of the wave's 64 W_ih fragment sets (k-chunk m, block b)
// beside the h and x tiles, and the other 33 sets (132 registers) live in the architectural VGPRs.  That leaves ~120
// registers for everything else, so the step is written for short live ranges: the recurrent product walks the k-chunks with
// both unit halves per chunk (one h fragment live at a time; the cell update of both halves follows the last MFMA), and the
// input part takes its LDS fragments in groups of four sets, one group ahead.
constexpr int X128_RES = 33;                                   // fragment sets p = 8 m + b < X128_RES: registers; others: LDS
constexpr int X128_WL_FLOATS = 4 * (64 - X128_RES) * 256;      // [wave][set - RES][lane][4]
constexpr int X128_XS_FLOATS = 16 * 128;                       // one step's input rows, unpadded (LDS-DMA), 16-byte chunks XOR-swizzled
constexpr size_t X128_LDS_BYTES = sizeof(float) * (X128_WL_FLOATS + L16_HS_FLOATS + 2 * X128_XS_FLOATS);
static_assert(X128_LDS_BYTES <= 160 * 1024, "LDS budget");

template <class F, int... I>
static __device__ __forceinline__ void x16_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N_, class F>
static __device__ __forceinline__ void x16_static_for(F&& f) {
  x16_static_for_impl(f, std::make_integer_sequence<int, N_>{});
}

template <bool RELU>
__global__ __launch_bounds__(256) void lstm16x128_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ wih_f,
                                                          const float* __restrict__ wih_b, const float* __restrict__ bih_f,
                                                          const float* __restrict__ bih_b, const float* __restrict__ bhh_f,
                                                          const float* __restrict__ bhh_b, const float* __restrict__ whh_f,
                                                          const float* __restrict__ whh_b, float* __restrict__ hc, int ldh,
                                                          int64_t dump_row, SeqGeom g) {
  constexpr int NIN = 128, RES = X128_RES;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Wl = smem;                               // [4][64 - RES][256]
  float* Hs = smem + X128_WL_FLOATS;              // [2][16][L16_LDH]
  float* Xs = Hs + L16_HS_FLOATS;                 // [2][16][128], chunk c of row r stored at chunk position c ^ (r & 15)
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, i16 = lane & 15, ks = lane >> 4;
  const int st = blockIdx.x, d = blockIdx.y;
  const float* whh = d ? whh_b : whh_f;
  const float* wih = d ? wih_b : wih_f;
  const float* bih = d ? bih_b : bih_f;
  const float* bhh = d ? bhh_b : bhh_f;

  float wf[8][32];                                // W_hh fragments -> AGPRs
  float bsc[8];
  float* wl_wave = Wl + w * ((64 - RES) * 256) + lane * 4;
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const int row = (b >> 1) * L16_H + 32 * w + 16 * (b & 1) + i16;
    const float gs = l16_gate_scale(b >> 1);
    const float* wrow = whh + (int64_t)row * L16_H + 4 * ks;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const float4 v = *reinterpret_cast<const float4*>(wrow + 16 * m);
      wf[b][4 * m + 0] = v.x * gs;
      wf[b][4 * m + 1] = v.y * gs;
      wf[b][4 * m + 2] = v.z * gs;
      wf[b][4 * m + 3] = v.w * gs;
    }
    bsc[b] = (bih[row] + bhh[row]) * gs;
  }
#pragma unroll
  for (int b = 0; b < 8; ++b)
#pragma unroll
    for (int i = 0; i < 32; ++i) asm volatile("" : "+a"(wf[b][i]));
  float wiv[RES][4];                              // W_ih fragment sets 0 .. RES-1 -> VGPRs, the others -> LDS
#pragma unroll
  for (int p = 0; p < 64; ++p) {
    const int m = p >> 3, b = p & 7;
    const int row = (b >> 1) * L16_H + 32 * w + 16 * (b & 1) + i16;
    const float gs = l16_gate_scale(b >> 1);
    const float4 v = *reinterpret_cast<const float4*>(wih + (int64_t)row * NIN + 16 * m + 4 * ks);
    if (p < RES) {
      constexpr int dummy = 0;
      (void)dummy;
      wiv[p % RES][0] = v.x * gs;
      wiv[p % RES][1] = v.y * gs;
      wiv[p % RES][2] = v.z * gs;
      wiv[p % RES][3] = v.w * gs;
    } else {
      *reinterpret_cast<float4*>(wl_wave + (p - RES) * 256) = make_float4(v.x * gs, v.y * gs, v.z * gs, v.w * gs);
    }
  }
#pragma unroll
  for (int p = 0; p < RES; ++p)
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(wiv[p][i]));

  const int t0 = d ? g.len - 1 : 0, tdir = d ? -1 : 1;
  const int tstride = seq_token_stride(g);
  const int srow = 4 * w + (lane >> 5), scol = 4 * (lane & 31);
  // h rows and input rows are addressed as a wave-uniform base + a 32-bit byte offset per lane (the host checks that both tensors
  // are below 4 GiB for this kernel): four registers less than 64-bit pointers, which is what keeps the step loop free of scratch
  unsigned hp[2];
  const unsigned hstep = (unsigned)(tdir * tstride * ldh * 4);      // (may be "negative": offsets are taken modulo 2^32)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int q = st * 16 + srow + 2 * j;
    const int64_t tokb = q < g.nseq ? seq_token_base(g, q) : dump_row;
    hp[j] = (unsigned)(((tokb + (int64_t)t0 * tstride) * ldh + d * L16_H + scol) * 4);
  }
  char* const hcb = reinterpret_cast<char*>(hc);
  // input rows: 16 x 512 bytes per step, by LDS-DMA (no registers): wave w issues two 1-KiB requests, request j = 2w + i covers
  // rows 2j and 2j + 1 -- lane L delivers 16 bytes to LDS position 16 L of the request's KiB, so it FETCHES the chunk that
  // belongs there: row 2j + (L >> 5), chunk (L & 31) ^ (row & 15) (the XOR spreads a fragment read's 16 rows over the banks).
  // Padded sequences read the last real one.
  unsigned xp[2];
  const unsigned xstep = (unsigned)(tdir * tstride * ldx * 4);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = 2 * (2 * w + i) + (lane >> 5);
    const int q = st * 16 + row;
    const int64_t tokb = seq_token_base(g, q < g.nseq ? q : g.nseq - 1);
    xp[i] = (unsigned)(((tokb + (int64_t)t0 * tstride) * ldx + 4 * ((lane & 31) ^ (row & 15))) * 4);
  }
  // Issued as inline assembly ON PURPOSE (fcln.hip has the long version): through __builtin_amdgcn_global_load_lds the compiler knows
  // that memory -> LDS traffic is outstanding and, unable to tell the buffers apart, puts s_waitcnt vmcnt(0) in front of the NEXT LDS
  // access -- here the a_next fragment read eleven instructions behind the request, so every step sat out the round trip of the rows
  // it had just asked for (rounds 4's kernel: line `s_waitcnt vmcnt(0)` right behind the two requests).  The one wait
  // this traffic needs is written by hand at the end of the step.
  const uint32_t xs_lds = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)Xs;
  auto dma_x = [&](int s, int buf, int i) {        // x_s -> Xs[buf], this wave's request i
    const unsigned adv = (unsigned)(s < g.len ? s : g.len - 1) * xstep;
    const uint32_t dst = xs_lds + (uint32_t)((buf * X128_XS_FLOATS + (2 * w + i) * 256) * 4);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(xp[i] + adv), "s"(x) : "memory", "m0");
  };
  dma_x(0, 0, 0); dma_x(0, 0, 1);
  dma_x(1, 1, 0); dma_x(1, 1, 1);
  for (int i = tid; i < 16 * L16_LDH; i += 256) Hs[i] = 0.f;
  f32x4v cst[2] = {(f32x4v){0.f, 0.f, 0.f, 0.f}, (f32x4v){0.f, 0.f, 0.f, 0.f}};
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  f32x4v acc[8];
  // the lane's A fragment of k-chunk m: chunk 4m + ks of row i16, at chunk position (4m + ks) ^ i16 = 4 (m ^ (i16 >> 2)) + (ks ^ (i16 & 3))
  const int xlane = i16 * 128 + 4 * (ks ^ (i16 & 3));
  const int xq = i16 >> 2;
  // acc = b + x W_ih^T: 16 groups of four fragment sets (k-chunk m = group / 2, blocks 4 (group & 1) .. + 3); the sets that live
  // in LDS are fetched ONE PAIR of sets (8 MFMAs = 256 cycles) ahead, by hand: written as ordinary loads "one group ahead" the
  // compiler -- with all 256 + 256 registers in use -- sank every one of them to the MFMA that consumes it and waited for it there:
  // a dozen exposed LDS round trips per step in the generated loop (`ds_read_b128; s_waitcnt lgkmcnt(0); 32 MFMAs`, round 5).  The
  // reads are volatile asm statements now (they stay where they are written, 16 registers for two pairs) and each pair is waited
  // for by one hand-written `s_waitcnt lgkmcnt(0)` that carries the pair's registers as operands.
  const uint32_t wl_lds = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)wl_wave;
  // k-chunk 0 of the NEXT step's input part is held back: its x fragment is read here (x0) and its 32 MFMAs -- registers only, sets
  // 0..7 are resident -- are issued by the next step right behind the barrier, in front of the recurrent product, where they cover
  // the round trip of the first h fragment (which cannot be requested before the barrier).
  // (HALF of it: k = 0, 1 of every lane's four -- 16 MFMAs = 512 cycles cover an LDS round trip, and two registers across the
  //  barrier fit where four brought a scratch reload into the loop)
  f32x2 x0;
  auto input_chunk0 = [&](float t0v, float t1v, int tt0) {
    const float av[2] = {t0v, t1v};
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int p = 4 * half + j;
          acc[p & 7] = mfma16(av[tt], wiv[p][tt0 + tt], acc[p & 7]);
        }
  };
  auto input_part = [&](int buf) {
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[b] = (f32x4v){bsc[b], bsc[b], bsc[b], bsc[b]};
    const float* xrow = Xs + buf * X128_XS_FLOATS + xlane;
    {
      const float4 xc0 = *reinterpret_cast<const float4*>(xrow + 16 * (0 ^ xq));
      x0 = (f32x2){xc0.x, xc0.y};
      input_chunk0(xc0.z, xc0.w, 2);      // k = 2, 3 of chunk 0 now, k = 0, 1 behind the barrier
    }
    float4 xa = *reinterpret_cast<const float4*>(xrow + 16 * (1 ^ xq));
    static_assert(RES >= 33 && RES <= 34 && (RES & 1) == 1, "sets 0..32 resident: the first LDS pair is (32 resident, 33)");
    // ---- k-chunks 1..3: sets 8..31, all resident
#pragma unroll
    for (int m = 1; m < 4; ++m) {
      const float4 xa_next = *reinterpret_cast<const float4*>(xrow + 16 * (((m + 1) & 3) ^ xq) + 64 * ((m + 1) >> 2));
      const float av[4] = {xa.x, xa.y, xa.z, xa.w};
#pragma unroll
      for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int p = 8 * m + 4 * half + j;
            acc[p & 7] = mfma16(av[tt], wiv[p][tt], acc[p & 7]);
          }
      xa = xa_next;
    }
    // ---- k-chunks 4..7: sets 32..63 in pairs (p0, p0 + 1); set 32 is resident, the other 31 come from LDS
    f32x4v fa[2][2];
    auto request = [](auto CH, f32x4v (&dst)[2], uint32_t base) {      // the LDS sets of pair CH
      constexpr int p0 = 32 + 2 * decltype(CH)::value;
      if constexpr (p0 >= RES)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[0]) : "v"(base), "n"((p0 - RES) * 1024));
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[1]) : "v"(base), "n"((p0 + 1 - RES) * 1024));
    };
    request(std::integral_constant<int, 0>{}, fa[0], wl_lds);
    float4 xa_next = xa;
    x16_static_for<16>([&](auto CH) {
      constexpr int ch = decltype(CH)::value, p0 = 32 + 2 * ch, m = p0 >> 3;
      if constexpr ((p0 & 7) == 0 && m + 1 < 8) xa_next = *reinterpret_cast<const float4*>(xrow + 16 * (((m + 1) & 3) ^ xq) + 64 * ((m + 1) >> 2));
      // this pair's sets have landed (nothing else of this wave is outstanding on the LDS counter that is younger)
      if constexpr (p0 >= RES) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[ch & 1][0]), "+v"(fa[ch & 1][1]) : : "memory");
      else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[ch & 1][1]) : : "memory");
      if constexpr (ch + 1 < 16) request(std::integral_constant<int, ch + 1 < 16 ? ch + 1 : 0>{}, fa[(ch + 1) & 1], wl_lds);
      __builtin_amdgcn_sched_barrier(0);      // (the pair's MFMAs behind the request: arithmetic is scheduled across an asm statement otherwise)
      const float av[4] = {xa.x, xa.y, xa.z, xa.w};
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        const float b0 = p0 < RES ? wiv[p0 < RES ? p0 : 0][tt] : fa[ch & 1][0][tt];
        acc[p0 & 7] = mfma16(av[tt], b0, acc[p0 & 7]);
        acc[(p0 + 1) & 7] = mfma16(av[tt], fa[ch & 1][1][tt], acc[(p0 + 1) & 7]);
      }
      if constexpr ((p0 & 7) == 6) xa = xa_next;      // last pair of the k-chunk
    });
  };
  input_part(0);
  __syncthreads();      // step 0 requests x_2 into the buffer of x_0

  for (int step = 0; step < g.len; ++step) {
    const float* hcur = Hs + (step & 1) * 16 * L16_LDH;
    float* hnext = Hs + ((step + 1) & 1) * 16 * L16_LDH;
    const float* arow = hcur + i16 * L16_LDH + 4 * ks;
    // (the two h rows this lane stores: the second is read when the first has left -- one of them live at a time)
    float4 hs = *reinterpret_cast<const float4*>(hcur + srow * L16_LDH + scol);
    float4 a = *reinterpret_cast<const float4*>(arow);
#pragma unroll
    for (int b = 0; b < 8; ++b) asm volatile("" : "+v"(acc[b]));
    input_chunk0(x0.x, x0.y, 0);      // (this step's x_t, fragment read before the barrier)
    __builtin_amdgcn_sched_barrier(0);  // (these 16 MFMAs FIRST: the h row store below was scheduled in front of them, with its wait)
    const unsigned adv = step > 0 ? hstep : 0u;

    // h_{t-1} W_hh^T, k-chunk by k-chunk, all eight blocks per chunk; one memory instruction per MFMA group in the first
    // groups: the stores of h_{t-1}, the input rows of step + 2 to LDS, the requests for the rows of step + 3
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      float4 a_next = a;
      if (m + 1 < 8) a_next = *reinterpret_cast<const float4*>(arow + 16 * (m + 1));
      const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[b] = mfma16(av[tt], wf[b][4 * m + tt], acc[b]);
        const int slot = 4 * m + tt;
        if (slot < 2) {
          asm volatile("" : "+v"(hs.x), "+v"(hs.y), "+v"(hs.z), "+v"(hs.w));
          if (RELU) hs = make_float4(relu1(hs.x), relu1(hs.y), relu1(hs.z), relu1(hs.w));
          *reinterpret_cast<float4*>(hcb + hp[slot]) = hs;
          hp[slot] += adv;
          if (slot == 0) hs = *reinterpret_cast<const float4*>(hcur + (srow + 2) * L16_LDH + scol);
          __builtin_amdgcn_sched_barrier(0);
        } else if (slot < 4) {
          dma_x(step + 2, step & 1, slot - 2);    // x_{step+2} -> the buffer of x_step (last read before the previous barrier)
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      a = a_next;
    }
    // cell update of both unit halves (lane-local), h_t -> the other LDS buffer
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
      for (int r = 0; r < 4; r += 2) {
        const LstmCell2 u = lstm_cell2((f32x2){acc[hf][r], acc[hf][r + 1]}, (f32x2){acc[2 + hf][r], acc[2 + hf][r + 1]},
                                       (f32x2){acc[4 + hf][r], acc[4 + hf][r + 1]}, (f32x2){acc[6 + hf][r], acc[6 + hf][r + 1]},
                                       (f32x2){cst[hf][r], cst[hf][r + 1]});
        cst[hf][r] = u.c.x;
        cst[hf][r + 1] = u.c.y;
        hnext[(4 * ks + r) * L16_LDH + 32 * w + 16 * hf + i16] = u.h.x;
        hnext[(4 * ks + r + 1) * L16_LDH + 32 * w + 16 * hf + i16] = u.h.y;
      }
    }
    if (step + 1 < g.len) input_part((step + 1) & 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's rows of x_{step+2} have landed (read behind the NEXT barrier)
    __syncthreads();
  }
  {
    const float* hfin = Hs + (g.len & 1) * 16 * L16_LDH;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float4 v = *reinterpret_cast<const float4*>(hfin + (srow + 2 * j) * L16_LDH + scol);
      if (RELU) v = make_float4(relu1(v.x), relu1(v.y), relu1(v.z), relu1(v.w));
      *reinterpret_cast<float4*>(hcb + hp[j]) = v;
    }
  }
}

}  // namespace

int lstm16x_launch(int nin, bool relu, int nst16, int ndir, void* stream, const float* x, int ldx, const float* const* wih,
                   const float* const* bih, const float* const* bhh, const float* const* whh, float* hc, int ldh,
                   int64_t dump_row, const SeqGeom& g) {
  if ((nin != 64 && nin != 128) || nst16 < 1 || ndir < 1 || ndir > 2 || g.len < 1) return (int)hipErrorInvalidValue;
  auto kern = nin == 64 ? (relu ? lstm16x_kernel<64, true> : lstm16x_kernel<64, false>)
                        : (relu ? lstm16x128_kernel<true> : lstm16x128_kernel<false>);
  const size_t lds = nin == 64 ? X16<64>::LDS_BYTES : X128_LDS_BYTES;
  static PerDeviceOnce ready[4];
  const int dev = current_hip_device();
  const int ki = (nin == 128 ? 2 : 0) + (relu ? 1 : 0);
  if (!ready[ki].done(dev)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    ready[ki].set(dev);
  }
  const int r = ndir == 2 ? 1 : 0;
  hipLaunchKernelGGL(kern, dim3(nst16, ndir), dim3(256), lds, static_cast<hipStream_t>(stream), x, ldx, wih[0], wih[r], bih[0],
                     bih[r], bhh[0], bhh[r], whh[0], whh[r], hc, ldh, dump_row, g);
  return (int)hipGetLastError();
}
