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
