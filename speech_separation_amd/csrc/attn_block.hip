// attn_block.hip -- the attention half of one TransformerDPRNN in ONE kernel (inference, N = 128, 4 heads of 32):
//
//     y1 = LayerNorm1( MHA(x) W_o^T + b_o + x )        src/model/dptn.py:46-47 (nn.MultiheadAttention :16-21, ln1 :22)
//
// i.e. what dptnav.hip otherwise runs as three launches (K1 in-projection GEMM, K2 attention, K3 out-projection +
// residual + LayerNorm) with QKV[M][384] and ATT[M][128] going through HBM in between: 5.5 kB of traffic per token and
// path become 1 kB (read x, write y1).  Own translation unit: compiled with -mllvm -amdgpu-mfma-vgpr-form (build.py) so
// that the score tiles the softmax works on live in architectural VGPRs.
//
// One workgroup = one sequence (len <= 160 positions), wave w = head w.  LDS holds the sequence's token rows (staged once
// with coalesced loads; every head reads them as MFMA fragments, and the epilogue takes the residual from there) and the
// out-projection partial sums; Q, K, V, the scores and the attention output never leave the registers:
//   * K^T and V of the head stay in REGISTERS for the whole sequence (2 x 16 x NKB accumulator registers per lane),
//     produced directly in the fragment layouts their consumers need:
//       K^T tile = W_k,h X^T   (A = weight rows, B = token rows)  -> reg r of lane (c,hh) = K[token c][d = ROW32(r,hh)]
//                                                                    = the A operand of S^T = K Q^T, MFMA step r
//       V tile   = X W_v,h^T   (A = token rows, B = weight rows)  -> reg r of lane (c,hh) = V[token ROW32(r,hh)][d = c]
//                                                                    = the A operand of O^T = V^T P^T, MFMA step r
//       Q^T tile = W_q,h X^T                                       -> the B operand of S^T, MFMA step r
//     (token rows X[token][k] serve as A or B operand from the same registers: lane (c,hh) holds X[token c][8m+4hh+t]);
//   * softmax is the streaming form over key blocks of 32 (running max / sum per query, everything of a query in its
//     lane pair (c,0),(c,1): one v_permlane32_swap per reduction), the S^T registers are the B operand of O^T as-is;
//   * O^T (reg r = O[query c][d = ROW32(r,hh)]) is the A operand of the head's share of the out-projection,
//     Y_h = O_h W_o[:, 32h:32h+32]^T; the four heads' partial tiles meet in LDS and the row-space epilogue (bias +
//     residual + LayerNorm, a row = 32 adjacent lanes) sums them in a fixed order.  No atomics: bit-reproducible.
// The head's weight fragments (W_q, W_k, W_v rows and its W_o slice: 4 x 16 KiB per wave) are loaded ONCE per sequence
// and stay in registers: row-per-lane fragment loads cost the texture path 32 cache lines per wave instruction, and
// streaming them per token block (first version) made the kernel L1-bound, slower than the three separate launches.
#include <hip/hip_runtime.h>

#include "attn_block.h"

namespace {

constexpr int N = 128, DH = 32;
constexpr int LDX = N + 4;                  // token rows: conflict-free ds_read_b128 fragments
constexpr int LDP = 136;                    // partial-tile row stride: rows 4 apart (the two lane halves) are 32 banks apart

DEV float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }
// group_sum<32> of FOUR independent values, step by step side by side: one reduction is a chain of five dependent
// cross-lane operations (each waits out the previous one's result), and the row-space code was four such passes one
// after the other -- 2.6-2.9 k cycles per 32-row block in the phase stamps (tools/attn_stamps.py).
DEV void group_sum32_x4(float (&v)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] += dpp_move<0xB1>(v[i]);
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] += dpp_move<0x4E>(v[i]);
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] += dpp_move<0x141>(v[i]);
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] += dpp_move<0x140>(v[i]);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned u = __builtin_bit_cast(unsigned, v[i]);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    v[i] = __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
  }
}
// LayerNorm of four rows at once: v[i] = this lane's 4 columns of row i (a row = 32 adjacent lanes).  The element-wise part is
// written on 2-vectors so that it issues as packed fp32 instructions (v_pk_add / v_pk_mul / v_pk_fma: two columns per issue;
// beside fp32 MFMAs every vector instruction costs its issue time, section 3.5 of DESIGN.md)
DEV void layernorm_rows_x4(float4 (&v)[4], const float4 ga, const float4 be) {
  f32x2 lo[4], hi[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    lo[i] = (f32x2){v[i].x, v[i].y};
    hi[i] = (f32x2){v[i].z, v[i].w};
  }
  float s[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f32x2 t = lo[i] + hi[i];
    s[i] = t.x + t.y;
  }
  group_sum32_x4(s);
  float q[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float mu = s[i] * (1.0f / N);
    const f32x2 m2 = (f32x2){mu, mu};
    lo[i] -= m2;
    hi[i] -= m2;
    const f32x2 t = lo[i] * lo[i] + hi[i] * hi[i];
    q[i] = t.x + t.y;
  }
  group_sum32_x4(q);
  const f32x2 galo = (f32x2){ga.x, ga.y}, gahi = (f32x2){ga.z, ga.w}, belo = (f32x2){be.x, be.y}, behi = (f32x2){be.z, be.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float rstd = rsqrtf(q[i] * (1.0f / N) + 1e-5f);
    const f32x2 r2 = (f32x2){rstd, rstd};
    const f32x2 a = lo[i] * r2 * galo + belo, b = hi[i] * r2 * gahi + behi;
    v[i] = make_float4(a.x, a.y, b.x, b.y);
  }
}
// MFMAs issued by iteration kb of the softmax pipeline: PV(kb-1) (kb > 0) + S(kb+1) (kb + 1 < nkb), 16 each
constexpr int kb_mfmas(int kb, int nkb) { return (kb > 0 ? 16 : 0) + (kb + 1 < nkb ? 16 : 0); }
template <class F, int... I>
DEV void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N_, class F>
DEV void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N_>{});
}

// max / sum across the two 32-lane halves (lanes (c,0) and (c,1) hold the two halves of a query's keys)
DEV float half_max(float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
}
DEV float half_sum(float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}

// PRO = true: the block's input rows are not read from memory but PRODUCED here, as the second half of the PREVIOUS
// TransformerDPRNN:  x = LayerNorm2(ReLU(h) W_f^T + b_f + y1_prev)  (dptn.py:50-51; K6 of dptnav.hip).  That GEMM is
// token-wise and every token belongs to exactly one sequence of this launch, so each workgroup computes it for its own
// rows (32-token blocks through LDS, wave w = output columns [32w, 32w+32) with its W_f slice in registers, LayerNorm in
// row space) straight into the staged tile: x never goes to HBM.  y1 is updated IN PLACE (a workgroup reads y1_prev of
// its tokens here and writes their y1 at the end; no other workgroup touches them).
struct FfnPro {
  const float* hc;      // [M][256] ReLU(h_fwd | h_bwd) of the previous path
  const float* wf;      // ffn.1.weight [128][256]
  const float* bf;
  const float* g2;      // ln2 weight / bias of the previous path
  const float* b2;
};
constexpr int LDHC = 256 + 4;

// Phase stamps (diagnostic build only: -DATTN_STAMPS, tools/attn_stamps.py): s_memtime sums per phase and wave,
// added up over all waves of all launches in g_ab_stamps.  The product build compiles them to nothing.
#ifdef ATTN_STAMPS
__device__ unsigned long long g_ab_stamps[16];
DEV unsigned long long ab_now() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  return t;
}
#define AB_DECL unsigned long long ab_t[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long ab_last = ab_now();
#define AB_MARK(i) { const unsigned long long ab_n = ab_now(); ab_t[i] += ab_n - ab_last; ab_last = ab_n; }
#define AB_KEEP(x) asm volatile("" :: "v"(x));
#define AB_END                                                                                      \
  if ((threadIdx.x & 63) == 0) {                                                                    \
    for (int ab_i = 0; ab_i < 12; ++ab_i) atomicAdd(&g_ab_stamps[ab_i], ab_t[ab_i]);                \
    atomicAdd(&g_ab_stamps[12], 1ull);                                                              \
  }
#else
#define AB_DECL
#define AB_MARK(i)
#define AB_KEEP(x)
#define AB_END
#endif

template <int NKB, bool PRO>
__global__ __launch_bounds__(256) void attn_block_kernel(const float* __restrict__ x, const float* __restrict__ wp_in,
                                                         const float* __restrict__ b_in, const float* __restrict__ wp_o,
                                                         const float* __restrict__ b_o, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* __restrict__ y1, SeqGeom g,
                                                         float scale_log2e, FfnPro pro) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Xs = smem;                          // [NKB*32][LDX]  the sequence's token rows (rows >= len repeat the last one)
  float* P = smem + NKB * 32 * LDX;          // [4 heads][32 rows][LDP]  out-projection partial tiles of one query block
  const int tid = threadIdx.x;
  const int h = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave = head
  const int lane = tid & 63, c = lane & 31, hh = lane >> 5;
  const int seq = blockIdx.x, len = g.len;
  const int64_t tok0 = seq_token_base(g, seq);
  const int tstride = seq_token_stride(g);
  // Row addresses as a wave-uniform 64-bit base (the sequence's first token) + a 32-bit byte offset per lane
  // (row x stride: < 160 x 150 x 1 KiB): the loads / stores then take the base from SGPRs and the offset costs a v_min (last
  // block only), a 24-bit multiply and an add -- written with 64-bit token arithmetic per access it was ~600 of the kernel's
  // 3 100 vector instructions (v_mul_lo_u32 at a quarter of the rate among them), beside fp32 MFMAs that pay for each one.
  const unsigned rs_h = (unsigned)tstride * 1024u;       // bytes between consecutive positions in hc [M][256]
  const unsigned rs_y = (unsigned)tstride * 512u;        // ... in x / y1 [M][128]
  char* const ybase = reinterpret_cast<char*>(y1) + tok0 * 512;
  auto row_off = [&](int row, unsigned rstride, bool clamp) -> unsigned {      // row = uniform part + lane part, < 2^24
    if (clamp) row = row < len ? row : len - 1;
    return (unsigned)__umul24((unsigned)row, rstride);
  };

  AB_DECL
  // This head's W_k / W_v fragments, resident through phase 1.  The row-per-lane fragment loads (32 rows x 32 bytes per
  // wave instruction) are expensive for the texture path and their round trip was 8.6 k exposed cycles per sequence in
  // the phase stamps: with the prologue they are requested in front of its LAST 32-token block (peeled from the loop
  // so that the 128 registers are not live through the others) and arrive behind its 128 MFMAs.
  float wkf[64], wvf[64];
  auto fetch_wkv = [&]() {
    const float* wk = wp_in + ((1 * 4 + h) * 16 * 64 + lane) * 4;   // packed: [sel][head][m][lane][4]
    const float* wv = wp_in + ((2 * 4 + h) * 16 * 64 + lane) * 4;
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const float4 a = ldg4(wk + m * 256), b = ldg4(wv + m * 256);
      wkf[4 * m + 0] = a.x; wkf[4 * m + 1] = a.y; wkf[4 * m + 2] = a.z; wkf[4 * m + 3] = a.w;
      wvf[4 * m + 0] = b.x; wvf[4 * m + 1] = b.y; wvf[4 * m + 2] = b.z; wvf[4 * m + 3] = b.w;
    }
  };
  if constexpr (PRO) {
    // ---- prologue: x rows = LN2(ReLU(h) W_f^T + b_f + y1_prev) of the previous path, block of 32 tokens at a time ----
    float* Hs = P;                           // [32][LDHC]  ReLU(h) rows of the block (A operand)
    float* Cs = P + 32 * LDHC;               // [32][LDP]   product tile on its way to row space
    float wff[128];                          // W_f[32h + c][8m + 4hh + t]: this wave's 32 output columns, K = 256
    {
      const float* wr = pro.wf + (h * 32 * 64 + lane) * 4;        // packed: [head][m][lane][4]
#pragma unroll
      for (int m = 0; m < 32; ++m) {
        const float4 t = ldg4(wr + m * 256);
        wff[4 * m + 0] = t.x; wff[4 * m + 1] = t.y; wff[4 * m + 2] = t.z; wff[4 * m + 3] = t.w;
      }
    }
    const int pc4 = tid & 31, prs = tid >> 5;
    const float4 bfc = ldg4(pro.bf + 4 * pc4), g2c = ldg4(pro.g2 + 4 * pc4), b2c = ldg4(pro.b2 + 4 * pc4);
    auto tok_of = [&](int row) { return tok0 + (int64_t)(row < len ? row : len - 1) * tstride; };
    // a block's 32 x 256 floats: 8 x 16 bytes per thread, fetched TWO blocks ahead (one block = ~10 k cycles was not
    // always enough under load: the stamps showed ~1.4 k cycles per block in front of the staging stores)
    float4 hst[2][8];
    auto fetch_h = [&](int rb, float4* dst) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int idx = i * 256 + tid;
        dst[i] = ldg4(pro.hc + tok_of(rb * 32 + (idx >> 6)) * 256 + 4 * (idx & 63));
      }
    };
    fetch_h(0, hst[0]);
    if (NKB > 1) fetch_h(1, hst[1]);
    // (the pin is a use: the W_f loads are waited for HERE, with the first rows already requested behind them)
#pragma unroll
    for (int i = 0; i < 128; ++i) asm volatile("" : "+a"(wff[i]));
    AB_MARK(0)
    auto pro_block = [&](auto RB) {
      constexpr int rb = decltype(RB)::value;
      if constexpr (rb == NKB - 1) fetch_wkv();
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int idx = i * 256 + tid;
        *reinterpret_cast<float4*>(&Hs[(idx >> 6) * LDHC + 4 * (idx & 63)]) = hst[rb & 1][i];
      }
      __syncthreads();
      AB_MARK(1)
      if constexpr (rb + 2 < NKB) fetch_h(rb + 2, hst[rb & 1]);
      float4 res[4];                         // residual rows y1_prev of this thread's four row-space slots
#pragma unroll
      for (int pass = 0; pass < 4; ++pass)
        res[pass] = *reinterpret_cast<const float4*>(ybase + row_off(rb * 32 + pass * 8 + prs, rs_y, rb == NKB - 1) + 16u * pc4);
      f32x16 a0 = zero16(), a1 = zero16();
      const float* ar = &Hs[c * LDHC + 4 * hh];
#pragma unroll
      for (int m0 = 0; m0 < 32; m0 += 8) {
        float4 af[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) af[m] = *reinterpret_cast<const float4*>(ar + 8 * (m0 + m));
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          a0 = mfma32(af[m].x, wff[4 * (m0 + m) + 0], a0);
          a1 = mfma32(af[m].y, wff[4 * (m0 + m) + 1], a1);
          a0 = mfma32(af[m].z, wff[4 * (m0 + m) + 2], a0);
          a1 = mfma32(af[m].w, wff[4 * (m0 + m) + 3], a1);
        }
      }
      AB_KEEP(a0[15]) AB_KEEP(a1[15])
      AB_MARK(2)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2 t = (f32x2){a0[r], a0[r + 1]} + (f32x2){a1[r], a1[r + 1]};
        Cs[ROW32(r, hh) * LDP + 32 * h + c] = t.x;
        Cs[ROW32(r + 1, hh) * LDP + 32 * h + c] = t.y;
      }
      __syncthreads();
      AB_MARK(3)
      {
        float4 v[4];
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
          const float4 cv = *reinterpret_cast<const float4*>(&Cs[(pass * 8 + prs) * LDP + 4 * pc4]);
          const f32x2 lo = (f32x2){cv.x, cv.y} + (f32x2){bfc.x, bfc.y} + (f32x2){res[pass].x, res[pass].y};
          const f32x2 hi = (f32x2){cv.z, cv.w} + (f32x2){bfc.z, bfc.w} + (f32x2){res[pass].z, res[pass].w};
          v[pass] = make_float4(lo.x, lo.y, hi.x, hi.y);
        }
        layernorm_rows_x4(v, g2c, b2c);
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) *reinterpret_cast<float4*>(&Xs[(rb * 32 + pass * 8 + prs) * LDX + 4 * pc4]) = v[pass];
      }
      AB_MARK(4)
    };
    static_for<NKB>(pro_block);
  } else {
    // ---- stage the token rows: coalesced 512-byte rows -> LDS (every wave reads all of them as MFMA fragments) -----
    constexpr int NLD = NKB * 4;             // float4 per thread
    float4 st[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int row = i * 8 + (tid >> 5);
      st[i] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(x) + tok0 * 512 + row_off(row, rs_y, i >= 4 * (NKB - 1)) +
                                               16u * (tid & 31));
    }
    fetch_wkv();
#pragma unroll
    for (int i = 0; i < NLD; ++i) *reinterpret_cast<float4*>(&Xs[(i * 8 + (tid >> 5)) * LDX + 4 * (tid & 31)]) = st[i];
  }
  __syncthreads();
  AB_MARK(5)

  // W_q fragments (A operand of the Q^T tiles, parked in AGPRs): requested here, they arrive behind phase 1
  float wqf[64];
  {
    const float* wq = wp_in + ((0 * 4 + h) * 16 * 64 + lane) * 4;
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const float4 a = ldg4(wq + m * 256);
      wqf[4 * m + 0] = a.x; wqf[4 * m + 1] = a.y; wqf[4 * m + 2] = a.z; wqf[4 * m + 3] = a.w;
    }
  }
  // ---- phase 1: K^T and V of this head for every key block, kept in registers --------------------------------
  f32x16 kt[NKB], vv[NKB];
  {
    const float bv = b_in[2 * N + h * DH + c];                  // V tile: column d = c
    float bk[16];                                               // K^T tile: row d = ROW32(r,hh) = 8j + 4hh + i
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 t = ldg4(b_in + N + h * DH + 8 * j + 4 * hh);
      bk[4 * j + 0] = t.x; bk[4 * j + 1] = t.y; bk[4 * j + 2] = t.z; bk[4 * j + 3] = t.w;
    }
#pragma unroll
    for (int rb = 0; rb < NKB; ++rb) {
      const float* xr = &Xs[(rb * 32 + c) * LDX + 4 * hh];
      f32x16 ka = zero16(), va = zero16();
#pragma unroll
      for (int m0 = 0; m0 < 16; m0 += 8) {                      // token-row fragments: two batches of 8 x ds_read_b128
        float4 xf[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) xf[m] = *reinterpret_cast<const float4*>(xr + 8 * (m0 + m));
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          const float xa[4] = {xf[m].x, xf[m].y, xf[m].z, xf[m].w};
#pragma unroll
          for (int t = 0; t < 4; ++t) {                         // two independent chains
            ka = mfma32(wkf[4 * (m0 + m) + t], xa[t], ka);
            va = mfma32(xa[t], wvf[4 * (m0 + m) + t], va);
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2 tk = (f32x2){ka[r], ka[r + 1]} + (f32x2){bk[r], bk[r + 1]}, tv = (f32x2){va[r], va[r + 1]} + (f32x2){bv, bv};
        kt[rb][r] = tk.x; kt[rb][r + 1] = tk.y;
        vv[rb][r] = tv.x; vv[rb][r + 1] = tv.y;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        // parked in the AGPR half of the register file (one wave per SIMD: 256 + 256 registers per lane); the MFMAs of
        // phase 2 read their A operand there
        asm volatile("" : "+a"(kt[rb][r]), "+a"(vv[rb][r]));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  AB_MARK(6)
  // ---- phase 2 constants: W_q fragments (A operand, parked in AGPRs), W_o slice of this head (B operand), biases ----
#pragma unroll
  for (int i = 0; i < 64; ++i) asm volatile("" : "+a"(wqf[i]));
  float wof[4][16];                                              // W_o[32 jt + c][32 h + ROW32(r,hh)]
#pragma unroll
  for (int jt = 0; jt < 4; ++jt) {
    const float* wr = wp_o + (((h * 4 + jt) * 4) * 64 + lane) * 4;   // packed: [head][jt][j][lane][4]
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 t = ldg4(wr + j * 256);
      wof[jt][4 * j + 0] = t.x; wof[jt][4 * j + 1] = t.y; wof[jt][4 * j + 2] = t.z; wof[jt][4 * j + 3] = t.w;
    }
  }
  // row-space epilogue: thread = (row in pass, 4 columns); its per-column constants
  const int c4 = tid & 31, rsub = tid >> 5;
  const float4 bo = ldg4(b_o + 4 * c4), ga = ldg4(gamma + 4 * c4), be = ldg4(beta + 4 * c4);
  float qbias[16];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float4 t = ldg4(b_in + h * DH + 8 * j + 4 * hh);
    qbias[4 * j + 0] = t.x; qbias[4 * j + 1] = t.y; qbias[4 * j + 2] = t.z; qbias[4 * j + 3] = t.w;
  }
  float* Pw = P + h * 32 * LDP;

  // ---- phase 2: one query block at a time -----------------------------------------------------------------
  // Q^T tile of a block: 64 MFMAs in two chains (bias, scale by log2(e)/sqrt(dh) applied by the caller)
  auto q_tile_mfmas = [&](int qb, f32x16& q0, f32x16& q1) {
    const float* xr = &Xs[(qb * 32 + c) * LDX + 4 * hh];
#pragma unroll
    for (int m0 = 0; m0 < 16; m0 += 8) {
      float4 xf[8];
#pragma unroll
      for (int m = 0; m < 8; ++m) xf[m] = *reinterpret_cast<const float4*>(xr + 8 * (m0 + m));
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        q0 = mfma32(wqf[4 * (m0 + m) + 0], xf[m].x, q0);
        q1 = mfma32(wqf[4 * (m0 + m) + 1], xf[m].y, q1);
        q0 = mfma32(wqf[4 * (m0 + m) + 2], xf[m].z, q0);
        q1 = mfma32(wqf[4 * (m0 + m) + 3], xf[m].w, q1);
      }
    }
  };
  // row-space epilogue of a block: y1 = LayerNorm(sum_h Y_h + b_o + x); a row = 32 adjacent lanes, 16 bytes per lane; the
  // residual row comes from the staged tile
  auto ln_epilogue = [&](int qb) {
    float4 v[4];
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int row = pass * 8 + rsub;
      const float* pr = P + row * LDP + 4 * c4;
      const float4 a0 = *reinterpret_cast<const float4*>(pr);
      const float4 a1 = *reinterpret_cast<const float4*>(pr + 32 * LDP);
      const float4 a2 = *reinterpret_cast<const float4*>(pr + 64 * LDP);
      const float4 a3 = *reinterpret_cast<const float4*>(pr + 96 * LDP);
      const float4 xres = *reinterpret_cast<const float4*>(&Xs[(qb * 32 + row) * LDX + 4 * c4]);
      const f32x2 lo = (((f32x2){a0.x, a0.y} + (f32x2){a1.x, a1.y}) + ((f32x2){a2.x, a2.y} + (f32x2){a3.x, a3.y})) + (f32x2){bo.x, bo.y} +
                       (f32x2){xres.x, xres.y};
      const f32x2 hi = (((f32x2){a0.z, a0.w} + (f32x2){a1.z, a1.w}) + ((f32x2){a2.z, a2.w} + (f32x2){a3.z, a3.w})) + (f32x2){bo.z, bo.w} +
                       (f32x2){xres.z, xres.w};
      v[pass] = make_float4(lo.x, lo.y, hi.x, hi.y);
    }
    layernorm_rows_x4(v, ga, be);
    // Straight-line on purpose (a branch per row would cut this region into basic blocks and nothing could be scheduled
    // between the MFMAs next to it): padded rows p >= len carry the staged copy of row len-1 through the same
    // arithmetic, so their result IS row len-1's, bit for bit, and is stored there once more.
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int p = qb * 32 + pass * 8 + rsub;
      *reinterpret_cast<float4*>(ybase + row_off(p, rs_y, true) + 16u * c4) = v[pass];
    }
  };
  f32x16 qa = zero16(), qb_ = zero16();
  q_tile_mfmas(0, qa, qb_);
  AB_KEEP(qa[15]) AB_KEEP(qb_[15]) AB_KEEP(wof[3][15]) AB_KEEP(qbias[15]) AB_KEEP(bo.x) AB_KEEP(ga.x) AB_KEEP(be.x)
  AB_MARK(7)
  for (int qb = 0; qb < NKB; ++qb) {
    f32x16 q;
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const f32x2 t = ((f32x2){qa[r], qa[r + 1]} + (f32x2){qb_[r], qb_[r + 1]} + (f32x2){qbias[r], qbias[r + 1]}) * (f32x2){scale_log2e, scale_log2e};
      q[r] = t.x;
      q[r + 1] = t.y;
    }
    __builtin_amdgcn_sched_barrier(0);
    // Streaming softmax over the key blocks; O^T accumulated transposed (lane = query).  One wave per SIMD, so nothing
    // runs beside this wave: the loop is a two-stage pipeline in which iteration kb issues the MFMAs of
    // O^T += V^T P^T of block kb-1 and of the score tile S^T of block kb+1 while the vector unit does the softmax of
    // block kb (maximum, exp2, sum: ~100 instructions placed BETWEEN those MFMAs by the sched_group_barrier pattern;
    // behind them they were ~400 cycles per block in which the matrix pipe idled).  The rescale of O^T by
    // alpha(kb) waits until PV(kb-1) is in (end of the iteration; skipped, wave-uniformly, when no maximum moved).
    float mrun = -1e30f, lrun = 0.f;
    f32x16 o = zero16();
    f32x16 s = zero16();
#pragma unroll
    for (int r = 0; r < 16; ++r) s = mfma32(kt[0][r], q[r], s);          // S^T[key ROW32(.,hh)][query c]
    f32x16 pprev = zero16();
    auto softmax_step = [&](auto KB) {
      constexpr int kb = decltype(KB)::value;
      __builtin_amdgcn_sched_barrier(0);
      // ---- matrix stream of this iteration
      f32x16 snext = zero16();
      if constexpr (kb > 0 && kb + 1 < NKB) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          o = mfma32(vv[kb - 1][r], pprev[r], o);                         // O^T[d ROW32(.,hh)][query c]
          snext = mfma32(kt[kb + 1][r], q[r], snext);
        }
      } else if constexpr (kb > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) o = mfma32(vv[kb - 1][r], pprev[r], o);
      } else if constexpr (kb + 1 < NKB) {
#pragma unroll
        for (int r = 0; r < 16; ++r) snext = mfma32(kt[kb + 1][r], q[r], snext);
      }
      // ---- vector stream: softmax of block kb
      if (kb == NKB - 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kb * 32 + ROW32(r, hh) >= len) s[r] = -1e30f;
      }
      // (four partial maxima / sums: a 16-long dependent chain of v_max / v_add between the MFMAs stalls the in-order
      //  issue behind it)
      float m4[4] = {s[0], s[1], s[2], s[3]};
#pragma unroll
      for (int r = 4; r < 16; ++r) m4[r & 3] = fmaxf(m4[r & 3], s[r]);
      const float mx = half_max(fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3])));
      // Reference point of the exponentials: the running reference is kept while the block's maximum exceeds it by at
      // most 2^8 (p <= 256: no overflow, fp32 accumulation) -- mathematically the same softmax, and the rescale of O^T
      // (which has to wait for this iteration's PV MFMAs) becomes a rare event instead of one per block.
      const float mnew = (kb == 0 || mx > mrun + 8.0f) ? mx : mrun;
      const float alpha = fast_exp2(mrun - mnew);
      f32x16 p;
      f32x2 s2[2] = {(f32x2){0.f, 0.f}, (f32x2){0.f, 0.f}};      // four partial sums, two per packed add
      const f32x2 mn2 = (f32x2){mnew, mnew};
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2 dlt = (f32x2){s[r], s[r + 1]} - mn2;
        const f32x2 e = (f32x2){fast_exp2(dlt.x), fast_exp2(dlt.y)};
        p[r] = e.x;
        p[r + 1] = e.y;
        s2[(r >> 1) & 1] += e;
      }
      const float sum = half_sum((s2[0].x + s2[0].y) + (s2[1].x + s2[1].y));
      lrun = lrun * alpha + sum;
      // one MFMA, then a few vector instructions, ... (the groups are taken from this region in program order)
      constexpr int NM = kb_mfmas(kb, NKB);
      if constexpr (NM > 0) {
#pragma unroll
        for (int i = 0; i < NM; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, (120 + NM - 1) / NM, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // the running maximum rarely moves after the first blocks: rescale O^T only if some query's did (wave-uniform)
      if (kb > 0 && __builtin_amdgcn_ballot_w64(mnew != mrun) != 0ull) {
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] *= alpha;
      }
      mrun = mnew;
      pprev = p;
      s = snext;
    };
    static_for<NKB>(softmax_step);
    __builtin_amdgcn_sched_barrier(0);
    // PV of the LAST key block: MFMA steps 4g..4g+3 contract keys 8g..8g+7 of the block (ROW32), and the keys past the
    // sequence's end have p = 0 exactly -- their steps are skipped (wave-uniform; adds of +0, so the result is the same
    // bit for bit): 8 of 16 steps at 141 positions, 4 of 16 at 150
    const int last_groups = (len - 32 * (NKB - 1) + 7) >> 3;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      if (g4 < last_groups) {
#pragma unroll
        for (int r = 4 * g4; r < 4 * g4 + 4; ++r) o = mfma32(vv[NKB - 1][r], pprev[r], o);
      }
    }
    {
      const float inv = fast_rcp(lrun);
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2 t = (f32x2){o[r], o[r + 1]} * (f32x2){inv, inv};
        o[r] = t.x;
        o[r + 1] = t.y;
      }
    }
    AB_KEEP(o[15])
    AB_MARK(8)
    // Every wave has long finished the epilogue of the previous block (it ran beside that block's successor's Q tile,
    // a whole softmax loop ago): this barrier only makes that formal before the partial tiles are overwritten.
    __syncthreads();
    // this head's share of the out-projection, two column tiles at a time (two chains): Y_h[query][32 jt + c]
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
      f32x16 ya = zero16(), yb = zero16();
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        ya = mfma32(o[r], wof[2 * jp][r], ya);
        yb = mfma32(o[r], wof[2 * jp + 1][r], yb);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        Pw[ROW32(r, hh) * LDP + 32 * (2 * jp) + c] = ya[r];
        Pw[ROW32(r, hh) * LDP + 32 * (2 * jp + 1) + c] = yb[r];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    AB_MARK(9)
    // The row-space epilogue of this block (LDS reads, ~300 vector instructions with two 32-lane reductions per row,
    // the store) runs BETWEEN the 64 MFMAs of the next block's Q^T tile: neither depends on the other.
    __builtin_amdgcn_sched_barrier(0);
    if (qb + 1 < NKB) {
      qa = zero16();
      qb_ = zero16();
      q_tile_mfmas(qb + 1, qa, qb_);
      ln_epilogue(qb);
#pragma unroll
      for (int i = 0; i < 64; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x1f6, 6, 0);     // up to six of: VALU, SALU, VMEM, DS
      }
    } else {
      ln_epilogue(qb);
    }
    __builtin_amdgcn_sched_barrier(0);
    AB_KEEP(qa[15]) AB_KEEP(qb_[15])
    AB_MARK(10)
  }
  AB_END
}


// =====================================================================================================================
// OPT-IN split-precision variant (option "split_bf16"; never the default, never a parity claim): the same kernel on
// v_mfma_f32_32x32x16_bf16 with every operand split into bf16 hi + lo (x = hi + lo to 16 significant bits) and each
// product formed as hi*hi + hi*lo + lo*hi with fp32 accumulation -- 3 bf16 MFMAs of 32 cycles replace 8 fp32 MFMAs of
// 64.  The accumulator layout of the bf16 instruction is the fp32 one's (ROW32), and its operand fragment (lane (c,hh)
// holds elements k = 8hh + j, j < 8) is filled with accumulator registers 8mm .. 8mm+7 of the producing tile: element
// (hh, j) then means d (or key) ROW32(8mm + j, hh) in BOTH operands of the consuming product, which is all a
// contraction needs.  So the register-resident chain of the fp32 kernel carries over: K^T / V tiles are parked as bf16
// hi / lo packs (16 registers per tile, as before), Q^T, the probabilities and O^T are split on the fly.  Token rows are
// staged in LDS as two bf16 images; the residual of the epilogue is re-read from global memory in fp32.
DEV void split_tile(const f32x16& t, bf16x8 (&hi)[2], bf16x8 (&lo)[2]) {
#pragma unroll
  for (int mm = 0; mm < 2; ++mm)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      __bf16 h, l;
      split_bf16(t[8 * mm + j], h, l);
      hi[mm][j] = h;
      lo[mm][j] = l;
    }
}
DEV void split_row8(const float* p, bf16x8& hi, bf16x8& lo) {     // 8 consecutive floats (two 16-byte loads)
  const float4 v0 = ldg4(p), v1 = ldg4(p + 4);
  const float xv[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    __bf16 h, l;
    split_bf16(xv[e], h, l);
    hi[e] = h;
    lo[e] = l;
  }
}
// three-term product on split operands
DEV f32x16 mfma3(const bf16x8& ah, const bf16x8& al, const bf16x8& bh, const bf16x8& bl, f32x16 c) {
  c = mfma32_bf16(ah, bh, c);
  c = mfma32_bf16(ah, bl, c);
  return mfma32_bf16(al, bh, c);
}

constexpr int LDXB = N + 8;                 // bf16 token rows (272 bytes): conflict-free ds_read_b128 fragments

template <int NKB>
__global__ __launch_bounds__(256) void attn_block_split_kernel(const float* __restrict__ x, const float* __restrict__ w_in,
                                                               const float* __restrict__ b_in, const float* __restrict__ w_o,
                                                               const float* __restrict__ b_o, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float* __restrict__ y1, SeqGeom g,
                                                               float scale_log2e) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* P = smem;                                                     // [4 heads][32 rows][LDP] fp32 partial tiles
  __bf16* Xhi = reinterpret_cast<__bf16*>(smem + 4 * 32 * LDP);        // [NKB*32][LDXB]
  __bf16* Xlo = Xhi + NKB * 32 * LDXB;
  const int tid = threadIdx.x;
  const int h = __builtin_amdgcn_readfirstlane(tid >> 6);              // wave = head
  const int lane = tid & 63, c = lane & 31, hh = lane >> 5;
  const int seq = blockIdx.x, len = g.len;
  const int64_t tok0 = seq_token_base(g, seq);
  const int tstride = seq_token_stride(g);

  // ---- stage the token rows as bf16 hi / lo images -----------------------------------------------------------
  {
    constexpr int NLD = NKB * 4;
    float4 st[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int row = i * 8 + (tid >> 5);
      st[i] = ldg4(x + (tok0 + (int64_t)(row < len ? row : len - 1) * tstride) * N + 4 * (tid & 31));
    }
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const float xv[4] = {st[i].x, st[i].y, st[i].z, st[i].w};
      bf16x4 h4, l4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        __bf16 hv, lv;
        split_bf16(xv[e], hv, lv);
        h4[e] = hv;
        l4[e] = lv;
      }
      const int off = (i * 8 + (tid >> 5)) * LDXB + 4 * (tid & 31);
      *reinterpret_cast<bf16x4*>(&Xhi[off]) = h4;
      *reinterpret_cast<bf16x4*>(&Xlo[off]) = l4;
    }
  }
  // ---- this head's K / V weight fragments (k = 16m + 8hh + j), resident for phase 1 ------------------------------
  bf16x8 wkh[8], wkl[8], wvh[8], wvl[8];
  {
    const float* wk = w_in + (int64_t)(1 * N + h * DH + c) * N + 8 * hh;
    const float* wv = w_in + (int64_t)(2 * N + h * DH + c) * N + 8 * hh;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      split_row8(wk + 16 * m, wkh[m], wkl[m]);
      split_row8(wv + 16 * m, wvh[m], wvl[m]);
    }
  }
  __syncthreads();

  // ---- phase 1: K^T and V tiles of every key block, parked as bf16 hi / lo packs --------------------------------
  bf16x8 kth[NKB][2], ktl[NKB][2], vvh[NKB][2], vvl[NKB][2];
  {
    const float bv = b_in[2 * N + h * DH + c];
    float bk[16];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 t = ldg4(b_in + N + h * DH + 8 * j + 4 * hh);
      bk[4 * j + 0] = t.x; bk[4 * j + 1] = t.y; bk[4 * j + 2] = t.z; bk[4 * j + 3] = t.w;
    }
#pragma unroll
    for (int rb = 0; rb < NKB; ++rb) {
      const int roff = (rb * 32 + c) * LDXB + 8 * hh;
      f32x16 ka = zero16(), va = zero16();
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const bf16x8 xh = *reinterpret_cast<const bf16x8*>(&Xhi[roff + 16 * m]);
        const bf16x8 xl = *reinterpret_cast<const bf16x8*>(&Xlo[roff + 16 * m]);
        ka = mfma3(wkh[m], wkl[m], xh, xl, ka);         // K^T = W_k X^T: A = weight rows, B = token rows
        va = mfma3(xh, xl, wvh[m], wvl[m], va);         // V   = X W_v^T
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        ka[r] += bk[r];
        va[r] += bv;
      }
      split_tile(ka, kth[rb], ktl[rb]);
      split_tile(va, vvh[rb], vvl[rb]);
#pragma unroll
      for (int mm = 0; mm < 2; ++mm) asm volatile("" : "+a"(kth[rb][mm]), "+a"(ktl[rb][mm]), "+a"(vvh[rb][mm]), "+a"(vvl[rb][mm]));
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- phase 2 constants ------------------------------------------------------------------------------------
  bf16x8 wqh[8], wql[8];
  {
    const float* wq = w_in + (int64_t)(0 * N + h * DH + c) * N + 8 * hh;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      split_row8(wq + 16 * m, wqh[m], wql[m]);
      asm volatile("" : "+a"(wqh[m]), "+a"(wql[m]));
    }
  }
  bf16x8 woh[4][2], wol[4][2];                         // W_o[32 jt + c][32 h + ROW32(8 mm + j, hh)]
#pragma unroll
  for (int jt = 0; jt < 4; ++jt) {
    const float* wr = w_o + (int64_t)(32 * jt + c) * N + h * DH + 4 * hh;
#pragma unroll
    for (int mm = 0; mm < 2; ++mm) {
      const float4 v0 = ldg4(wr + 16 * mm), v1 = ldg4(wr + 16 * mm + 8);
      const float xv[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        __bf16 hv, lv;
        split_bf16(xv[e], hv, lv);
        woh[jt][mm][e] = hv;
        wol[jt][mm][e] = lv;
      }
    }
  }
  const int c4 = tid & 31, rsub = tid >> 5;
  const float4 bo = ldg4(b_o + 4 * c4), ga = ldg4(gamma + 4 * c4), be = ldg4(beta + 4 * c4);
  float qbias[16];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float4 t = ldg4(b_in + h * DH + 8 * j + 4 * hh);
    qbias[4 * j + 0] = t.x; qbias[4 * j + 1] = t.y; qbias[4 * j + 2] = t.z; qbias[4 * j + 3] = t.w;
  }
  float* Pw = P + h * 32 * LDP;

  for (int qb = 0; qb < NKB; ++qb) {
    // the fp32 residual rows of this block's epilogue: requested now, used after the out-projection
    float4 xres[4];
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int p = qb * 32 + pass * 8 + rsub;
      xres[pass] = ldg4(x + (tok0 + (int64_t)(p < len ? p : len - 1) * tstride) * N + 4 * c4);
    }
    // Q^T tile, bias, scale, split
    bf16x8 qh[2], ql[2];
    {
      const int roff = (qb * 32 + c) * LDXB + 8 * hh;
      f32x16 q = zero16();
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const bf16x8 xh = *reinterpret_cast<const bf16x8*>(&Xhi[roff + 16 * m]);
        const bf16x8 xl = *reinterpret_cast<const bf16x8*>(&Xlo[roff + 16 * m]);
        q = mfma3(wqh[m], wql[m], xh, xl, q);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) q[r] = (q[r] + qbias[r]) * scale_log2e;
      split_tile(q, qh, ql);
    }
    float mrun = -1e30f, lrun = 0.f;
    f32x16 o = zero16();
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      f32x16 s = zero16();
#pragma unroll
      for (int mm = 0; mm < 2; ++mm) s = mfma3(kth[kb][mm], ktl[kb][mm], qh[mm], ql[mm], s);   // S^T[key][query]
      if (kb == NKB - 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kb * 32 + ROW32(r, hh) >= len) s[r] = -1e30f;
      }
      float mx = s[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
      mx = half_max(mx);
      const float mnew = fmaxf(mrun, mx);
      const float alpha = fast_exp2(mrun - mnew);
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[r] = fast_exp2(s[r] - mnew);
        sum += s[r];
      }
      sum = half_sum(sum);
      lrun = lrun * alpha + sum;
      if (kb > 0 && __builtin_amdgcn_ballot_w64(mnew != mrun) != 0ull) {
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] *= alpha;
      }
      mrun = mnew;
      bf16x8 ph[2], pl[2];
      split_tile(s, ph, pl);
#pragma unroll
      for (int mm = 0; mm < 2; ++mm) o = mfma3(vvh[kb][mm], vvl[kb][mm], ph[mm], pl[mm], o);   // O^T[d][query]
    }
    {
      const float inv = fast_rcp(lrun);
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2 t = (f32x2){o[r], o[r + 1]} * (f32x2){inv, inv};
        o[r] = t.x;
        o[r + 1] = t.y;
      }
    }
    bf16x8 oh[2], ol[2];
    split_tile(o, oh, ol);
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      f32x16 ya = zero16();
#pragma unroll
      for (int mm = 0; mm < 2; ++mm) ya = mfma3(oh[mm], ol[mm], woh[jt][mm], wol[jt][mm], ya);
#pragma unroll
      for (int r = 0; r < 16; ++r) Pw[ROW32(r, hh) * LDP + 32 * jt + c] = ya[r];
    }
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int row = pass * 8 + rsub;
      const int p = qb * 32 + row;
      const float* pr = P + row * LDP + 4 * c4;
      const float4 a0 = *reinterpret_cast<const float4*>(pr);
      const float4 a1 = *reinterpret_cast<const float4*>(pr + 32 * LDP);
      const float4 a2 = *reinterpret_cast<const float4*>(pr + 64 * LDP);
      const float4 a3 = *reinterpret_cast<const float4*>(pr + 96 * LDP);
      float4 v;
      v.x = ((a0.x + a1.x) + (a2.x + a3.x)) + bo.x + xres[pass].x;
      v.y = ((a0.y + a1.y) + (a2.y + a3.y)) + bo.y + xres[pass].y;
      v.z = ((a0.z + a1.z) + (a2.z + a3.z)) + bo.z + xres[pass].z;
      v.w = ((a0.w + a1.w) + (a2.w + a3.w)) + bo.w + xres[pass].w;
      const float mu = group_sum<32>((v.x + v.y) + (v.z + v.w)) * (1.0f / N);
      const float dx = v.x - mu, dy = v.y - mu, dz = v.z - mu, dw = v.w - mu;
      const float var = group_sum<32>((dx * dx + dy * dy) + (dz * dz + dw * dw)) * (1.0f / N);
      const float rstd = rsqrtf(var + 1e-5f);
      if (p < len) {
        float4 y;
        y.x = dx * rstd * ga.x + be.x;
        y.y = dy * rstd * ga.y + be.y;
        y.z = dz * rstd * ga.z + be.z;
        y.w = dw * rstd * ga.w + be.w;
        *reinterpret_cast<float4*>(y1 + (tok0 + (int64_t)p * tstride) * N + 4 * c4) = y;
      }
    }
    __syncthreads();
  }
}

// ---- weight packing (see attn_block.h) ---------------------------------------------------------------------------
struct AttnPackArgs {
  AttnPackSrc src[ATTN_PACK_MAX_PATHS];
};
__global__ __launch_bounds__(256) void attn_pack_kernel(AttnPackArgs a, float* __restrict__ dst) {
  const int path = blockIdx.y;
  const int f = blockIdx.x * 256 + threadIdx.x;               // float4 index inside the path's pack
  const AttnPackSrc s = a.src[path];
  const int lane = f & 63, c = lane & 31, hh = lane >> 5;
  const float* src;
  if (f < ATTN_PACK_IN / 4) {                                  // [sel][head][m][lane]
    const int m = (f >> 6) & 15, h = (f >> 10) & 3, sel = f >> 12;
    src = s.w_in + (size_t)(sel * N + h * DH + c) * N + 8 * m + 4 * hh;
  } else if (f < (ATTN_PACK_IN + ATTN_PACK_OUT) / 4) {         // [head][jt][j][lane]
    const int g = f - ATTN_PACK_IN / 4;
    const int j = (g >> 6) & 3, jt = (g >> 8) & 3, h = g >> 10;
    src = s.w_o + (size_t)(32 * jt + c) * N + h * DH + 8 * j + 4 * hh;
  } else {                                                     // [head][m][lane]
    if (s.w_f == nullptr) {   // a path whose FFN is not 128 x 256 (unidirectional LSTM: 128 x 128): never used as a prologue
      *reinterpret_cast<float4*>(dst + (size_t)path * ATTN_PACK_FLOATS + 4 * (size_t)f) = make_float4(0.f, 0.f, 0.f, 0.f);
      return;
    }
    const int g = f - (ATTN_PACK_IN + ATTN_PACK_OUT) / 4;
    const int m = (g >> 6) & 31, h = g >> 11;
    src = s.w_f + (size_t)(32 * h + c) * 256 + 8 * m + 4 * hh;
  }
  *reinterpret_cast<float4*>(dst + (size_t)path * ATTN_PACK_FLOATS + 4 * (size_t)f) = *reinterpret_cast<const float4*>(src);
}

}  // namespace

int attn_pack_launch(void* stream, const AttnPackSrc* src, int npaths, float* dst) {
  static_assert(ATTN_PACK_FLOATS % 1024 == 0, "pack grid");
  for (int p0 = 0; p0 < npaths; p0 += ATTN_PACK_MAX_PATHS) {
    const int n = npaths - p0 < ATTN_PACK_MAX_PATHS ? npaths - p0 : ATTN_PACK_MAX_PATHS;
    AttnPackArgs a{};
    for (int i = 0; i < n; ++i) a.src[i] = src[p0 + i];
    hipLaunchKernelGGL(attn_pack_kernel, dim3(ATTN_PACK_FLOATS / 1024, n), dim3(256), 0, static_cast<hipStream_t>(stream), a,
                       dst + (size_t)p0 * ATTN_PACK_FLOATS);
  }
  return (int)hipGetLastError();
}

size_t attn_block_lds_bytes(int nkb) { return sizeof(float) * ((size_t)nkb * 32 * LDX + 4 * 32 * LDP); }

int attn_block_launch(void* stream, const float* x, const float* w_in, const float* b_in, const float* w_o, const float* b_o,
                      const float* gamma, const float* beta, float* y1, const SeqGeom& g, bool split, const AttnFfnPrologue* pro,
                      const float* wpack) {
  using Kern = void (*)(const float*, const float*, const float*, const float*, const float*, const float*, const float*, float*,
                        SeqGeom, float, FfnPro);
  using KernS = void (*)(const float*, const float*, const float*, const float*, const float*, const float*, const float*, float*,
                         SeqGeom, float);
  const int nkb = (g.len + 31) / 32;
  if (nkb < 1 || nkb > 5 || (split && pro) || (!split && !wpack)) return (int)hipErrorInvalidValue;
  const float scale_log2e = 1.4426950408889634f / sqrtf((float)DH);
  const int dev = current_hip_device();
  if (split) {
    KernS kern;
    switch (nkb) {
      case 1: kern = attn_block_split_kernel<1>; break;
      case 2: kern = attn_block_split_kernel<2>; break;
      case 3: kern = attn_block_split_kernel<3>; break;
      case 4: kern = attn_block_split_kernel<4>; break;
      default: kern = attn_block_split_kernel<5>; break;
    }
    // (the split variant keeps two bf16 images of the token rows: 2 x 272 bytes per row instead of 528)
    const size_t lds = sizeof(float) * 4 * 32 * LDP + sizeof(__bf16) * 2 * (size_t)nkb * 32 * LDXB;
    static PerDeviceOnce ready[6];
    if (!ready[nkb].done(dev)) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return (int)e;
      ready[nkb].set(dev);
    }
    hipLaunchKernelGGL(kern, dim3(g.nseq), dim3(256), lds, static_cast<hipStream_t>(stream), x, w_in, b_in, w_o, b_o, gamma, beta,
                       y1, g, scale_log2e);
    return (int)hipGetLastError();
  }
  Kern kern;
  const bool p = pro != nullptr;
  switch (nkb) {
    case 1: kern = p ? attn_block_kernel<1, true> : attn_block_kernel<1, false>; break;
    case 2: kern = p ? attn_block_kernel<2, true> : attn_block_kernel<2, false>; break;
    case 3: kern = p ? attn_block_kernel<3, true> : attn_block_kernel<3, false>; break;
    case 4: kern = p ? attn_block_kernel<4, true> : attn_block_kernel<4, false>; break;
    default: kern = p ? attn_block_kernel<5, true> : attn_block_kernel<5, false>; break;
  }
  const size_t lds = attn_block_lds_bytes(nkb);
  static PerDeviceOnce ready_all[2][6];
  PerDeviceOnce* ready = ready_all[p ? 1 : 0];
  if (!ready[nkb].done(dev)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    ready[nkb].set(dev);
  }
  FfnPro fp{};
  if (p) fp = FfnPro{pro->hc, pro->wf, pro->bf, pro->g2, pro->b2};
  hipLaunchKernelGGL(kern, dim3(g.nseq), dim3(256), lds, static_cast<hipStream_t>(stream), x, wpack, b_in, wpack + ATTN_PACK_IN,
                     b_o, gamma, beta, y1, g, scale_log2e, fp);
  return (int)hipGetLastError();
}

#ifdef ATTN_STAMPS
// diagnostic build only (tools/attn_stamps.py): read / reset the phase sums
extern "C" int dptnav_debug_attn_stamps(unsigned long long* out, int reset) {
  if (reset) {
    unsigned long long z[16] = {0};
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_ab_stamps), z, sizeof(z));
  }
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ab_stamps), 16 * sizeof(unsigned long long));
}
#endif
