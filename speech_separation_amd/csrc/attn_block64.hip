// attn_block64.hip -- the attention half of one TransformerDPRNN in ONE kernel for num_features = 64 (4 heads of 16:
// src/configs/model/dptn_wav.yaml, BASELINE config 2), inference:
//
//     y1 = LayerNorm1( MHA(x) W_o^T + b_o + x )        src/model/dptn.py:46-47 (nn.MultiheadAttention :16-21, ln1 :22)
//
// and, as its optional prologue, the FFN half of the PREVIOUS TransformerDPRNN, which produces the block's input rows
//
//     x = LayerNorm2( ReLU(h) W_f^T + b_f + y1_prev )  dptn.py:50-51
//
// i.e. what dptnav.hip otherwise runs as K6 + K1 + K2 + K3 (four launches with x, QKV[M][192] and ATT[M][64] through HBM).
// Same design as attn_block.hip (N = 128), re-tiled for head dimension 16 on v_mfma_f32_16x16x4_f32 -- a 16 x 16 tile
// IS one head's slice of 16 tokens, so sequences are padded to multiples of 16 positions only (141 -> 144, 150 -> 160):
//
//   lane l: i16 = l & 15, ks = l >> 4.   A operand: lane (i16, ks) supplies A[row i16][k-slot ks];
//   B operand: B[k-slot ks][col i16];    D: register r of lane (i16, ks) = D[row 4 ks + r][col i16].
//   Fragments are fetched as float4: MFMA step 4 m + t uses the true k = 16 m + 4 ks + t in BOTH operands.
//
// One workgroup = one sequence (<= 160 positions), wave w = head w; two workgroups per CU (<= 256 registers per lane,
// <= 78 KB of LDS each), so one sequence's barriers / softmax / LayerNorm phases overlap the other's MFMAs.
//   K^T tile = W_k,h X^T  (A = weight rows, B = token rows)  -> reg r = K[token i16][d = 4 ks + r] = A operand of S^T, step r
//   V tile   = X W_v,h^T  (A = token rows, B = weight rows)  -> reg r = V[token 4 ks + r][d = i16] = A operand of O^T, step r
//   Q^T tile = W_q,h X^T                                      -> reg r = Q[query i16][d = 4 ks + r] = B operand of S^T, step r
//   S^T = K Q^T (4 MFMAs per 16 keys x 16 queries): reg r = score(key 4 ks + r, query i16): a query's scores sit in the
//   four lanes (i16, 0..3) -- ALL key blocks of a query block are formed first (<= 40 registers), so the softmax is the
//   plain two-pass form (max, exp2, sum; two cross-row swaps per reduction) and the S^T registers are the B operand of
//   O^T = V^T P^T as they stand; O^T (reg r = O[query i16][d = 4 ks + r]) is the A operand of the head's share of the
//   out-projection.  The four heads' partial tiles meet in LDS (double buffered: one barrier per query block) and the
//   row-space epilogue (bias + residual + LayerNorm, a row = 16 adjacent lanes) sums them in a fixed order.
// No atomics: bit-reproducible.  Weights are read straight from the nn.Module tensors (48 + 16 + 64 KiB per path: they
// live in L2; a fragment load is 16 rows x 64 contiguous bytes).
#include <hip/hip_runtime.h>

#include "attn_block64.h"

namespace {

constexpr int N = 64, DH = 16;
constexpr int LDX = N + 4;                  // token rows: conflict-free ds_read_b128 fragments (row stride = 4 banks mod 64)
constexpr int LDP = N + 4;                  // partial tiles / product tile
constexpr int KF = 256, LDHC = KF + 4;      // ReLU(h) rows of the prologue

typedef float f32x4v __attribute__((ext_vector_type(4)));
DEV f32x4v mfma16(float a, float b, f32x4v c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
DEV f32x4v zero4() { return (f32x4v){0.f, 0.f, 0.f, 0.f}; }
DEV float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// max / sum over the four lanes (i16, ks = 0..3) that share a query: lanes 16 and 32 apart
DEV float quad_rows_max(float v) {
  unsigned u = __builtin_bit_cast(unsigned, v);
  auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  v = fmaxf(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
  u = __builtin_bit_cast(unsigned, v);
  r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
}
DEV float quad_rows_sum(float v) {
  unsigned u = __builtin_bit_cast(unsigned, v);
  auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  v = __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
  u = __builtin_bit_cast(unsigned, v);
  r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}

// LayerNorm over a 64-wide row held by 16 adjacent lanes (4 columns each)
DEV float4 layernorm_row16(float4 v, const float4 ga, const float4 be) {
  const float mu = group_sum<16>((v.x + v.y) + (v.z + v.w)) * (1.0f / N);
  v.x -= mu; v.y -= mu; v.z -= mu; v.w -= mu;
  const float var = group_sum<16>((v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w)) * (1.0f / N);
  const float rstd = rsqrtf(var + 1e-5f);
  return make_float4(v.x * rstd * ga.x + be.x, v.y * rstd * ga.y + be.y, v.z * rstd * ga.z + be.z, v.w * rstd * ga.w + be.w);
}

template <class F, int... I>
DEV void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N_, class F>
DEV void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N_>{});
}

struct FfnPro64 {
  const float* hc;      // [M][256] ReLU(h_fwd | h_bwd) of the previous path
  const float* wf;      // ffn.1.weight [64][256]
  const float* bf;
  const float* g2;      // ln2 weight / bias of the previous path
  const float* b2;
};

// NB = ceil(len / 16) blocks of 16 positions
template <int NB, bool PRO>
__global__ __launch_bounds__(256, 2) void attn_block64_kernel(const float* __restrict__ x, const float* __restrict__ w_in,
                                                              const float* __restrict__ b_in, const float* __restrict__ w_o,
                                                              const float* __restrict__ b_o, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float* __restrict__ y1, SeqGeom g,
                                                              float scale_log2e, FfnPro64 pro) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Xs = smem;                          // [NB*16][LDX]  the sequence's token rows (rows >= len repeat the last one)
  float* P = smem + NB * 16 * LDX;           // phase 2: [2][4 heads][16 rows][LDP];  prologue: Hs [16][LDHC] + Cs [16][LDP]
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave = head
  const int lane = tid & 63, i16 = lane & 15, ks = lane >> 4;
  const int rrow = tid >> 4, c4 = tid & 15;                    // row space: thread = (row of a 16-row block, 4 columns)
  const int seq = blockIdx.x, len = g.len;
  const int64_t tok0 = seq_token_base(g, seq);
  const int tstride = seq_token_stride(g);
  auto tok_of = [&](int row) { return tok0 + (int64_t)(row < len ? row : len - 1) * tstride; };

  if constexpr (PRO) {
    // ---- prologue: x rows = LN2(ReLU(h) W_f^T + b_f + y1_prev) of the previous path, 16 tokens at a time ------------
    float* Hs = P;                           // [16][LDHC]  ReLU(h) rows of the block (A operand)
    float* Cs = P + 16 * LDHC;               // [16][LDP]   product tile on its way to row space
    float wff[64];                           // W_f[16 w + i16][16 m + 4 ks + t]: this wave's 16 output columns, K = 256
    {
      const float* wr = pro.wf + (int64_t)(16 * w + i16) * KF + 4 * ks;
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        const float4 t = ldg4(wr + 16 * m);
        wff[4 * m + 0] = t.x; wff[4 * m + 1] = t.y; wff[4 * m + 2] = t.z; wff[4 * m + 3] = t.w;
      }
    }
    const float4 bfc = ldg4(pro.bf + 4 * c4), g2c = ldg4(pro.g2 + 4 * c4), b2c = ldg4(pro.b2 + 4 * c4);
    // a block's 16 x 256 floats: 4 x 16 bytes per thread, fetched two blocks ahead
    float4 hst[2][4];
    auto fetch_h = [&](int rb, float4* dst) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int idx = i * 256 + tid;
        dst[i] = ldg4(pro.hc + tok_of(rb * 16 + (idx >> 6)) * KF + 4 * (idx & 63));
      }
    };
    fetch_h(0, hst[0]);
    if (NB > 1) fetch_h(1, hst[1]);
    auto pro_block = [&](auto RB) {
      constexpr int rb = decltype(RB)::value;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int idx = i * 256 + tid;
        *reinterpret_cast<float4*>(&Hs[(idx >> 6) * LDHC + 4 * (idx & 63)]) = hst[rb & 1][i];
      }
      __syncthreads();
      if constexpr (rb + 2 < NB) fetch_h(rb + 2, hst[rb & 1]);
      const float4 res = ldg4(y1 + tok_of(rb * 16 + rrow) * N + 4 * c4);      // residual row y1_prev
      f32x4v a0 = zero4(), a1 = zero4(), a2 = zero4(), a3 = zero4();        // four chains over k
      const float* ar = &Hs[i16 * LDHC + 4 * ks];
#pragma unroll
      for (int m0 = 0; m0 < 16; m0 += 8) {
        float4 af[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) af[m] = *reinterpret_cast<const float4*>(ar + 16 * (m0 + m));
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          a0 = mfma16(af[m].x, wff[4 * (m0 + m) + 0], a0);
          a1 = mfma16(af[m].y, wff[4 * (m0 + m) + 1], a1);
          a2 = mfma16(af[m].z, wff[4 * (m0 + m) + 2], a2);
          a3 = mfma16(af[m].w, wff[4 * (m0 + m) + 3], a3);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) Cs[(4 * ks + r) * LDP + 16 * w + i16] = (a0[r] + a1[r]) + (a2[r] + a3[r]);
      __syncthreads();
      {
        const float4 cv = *reinterpret_cast<const float4*>(&Cs[rrow * LDP + 4 * c4]);
        float4 v = make_float4(cv.x + bfc.x + res.x, cv.y + bfc.y + res.y, cv.z + bfc.z + res.z, cv.w + bfc.w + res.w);
        v = layernorm_row16(v, g2c, b2c);
        *reinterpret_cast<float4*>(&Xs[(rb * 16 + rrow) * LDX + 4 * c4]) = v;
      }
    };
    static_for<NB>(pro_block);
  } else {
    // ---- stage the token rows: coalesced 256-byte rows -> LDS (every wave reads all of them as MFMA fragments) -----
    float4 st[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) st[i] = ldg4(x + tok_of(i * 16 + rrow) * N + 4 * c4);
#pragma unroll
    for (int i = 0; i < NB; ++i) *reinterpret_cast<float4*>(&Xs[(i * 16 + rrow) * LDX + 4 * c4]) = st[i];
  }

  // ---- this head's projection weights: A operand rows (W_q, W_k) / B operand rows (W_v): W[16 w + i16][16 m + 4 ks + t] ----
  float wq[16], wk[16], wv[16];
  {
    const float* wr = w_in + (int64_t)(16 * w + i16) * N + 4 * ks;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const float4 a = ldg4(wr + 16 * m), b = ldg4(wr + (int64_t)N * N + 16 * m), c = ldg4(wr + (int64_t)2 * N * N + 16 * m);
      wq[4 * m + 0] = a.x; wq[4 * m + 1] = a.y; wq[4 * m + 2] = a.z; wq[4 * m + 3] = a.w;
      wk[4 * m + 0] = b.x; wk[4 * m + 1] = b.y; wk[4 * m + 2] = b.z; wk[4 * m + 3] = b.w;
      wv[4 * m + 0] = c.x; wv[4 * m + 1] = c.y; wv[4 * m + 2] = c.z; wv[4 * m + 3] = c.w;
    }
  }
  const float4 bq4 = ldg4(b_in + 16 * w + 4 * ks), bk4 = ldg4(b_in + N + 16 * w + 4 * ks);   // rows d = 4 ks + r
  const float bq[4] = {bq4.x, bq4.y, bq4.z, bq4.w}, bk[4] = {bk4.x, bk4.y, bk4.z, bk4.w};
  const float bv = b_in[2 * N + 16 * w + i16];                                               // column d = i16
  __syncthreads();

  // ---- phase 1: K^T and V of this head for every key block, kept in registers -----------------------------------
  f32x4v kt[NB], vv[NB];
#pragma unroll
  for (int rb = 0; rb < NB; ++rb) {
    const float* xr = &Xs[(rb * 16 + i16) * LDX + 4 * ks];
    float4 xf[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) xf[m] = *reinterpret_cast<const float4*>(xr + 16 * m);
    f32x4v ka = zero4(), va = zero4();
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const float xa[4] = {xf[m].x, xf[m].y, xf[m].z, xf[m].w};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        ka = mfma16(wk[4 * m + t], xa[t], ka);
        va = mfma16(xa[t], wv[4 * m + t], va);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      kt[rb][r] = ka[r] + bk[r];
      vv[rb][r] = va[r] + bv;
    }
  }

  // ---- phase 2 constants: W_o slice of this head (B operand), row-space constants ----------------------------------
  float wo[4][4];                                                // W_o[16 jt + i16][16 w + 4 ks + r]
#pragma unroll
  for (int jt = 0; jt < 4; ++jt) {
    const float4 t = ldg4(w_o + (int64_t)(16 * jt + i16) * N + 16 * w + 4 * ks);
    wo[jt][0] = t.x; wo[jt][1] = t.y; wo[jt][2] = t.z; wo[jt][3] = t.w;
  }
  const float4 bo = ldg4(b_o + 4 * c4), ga = ldg4(gamma + 4 * c4), be = ldg4(beta + 4 * c4);

  // ---- phase 2: one block of 16 queries at a time ---------------------------------------------------------------
#pragma unroll 1
  for (int qb = 0; qb < NB; ++qb) {
    // Q^T tile of the block (bias, scale by log2(e)/sqrt(dh))
    f32x4v q;
    {
      const float* xr = &Xs[(qb * 16 + i16) * LDX + 4 * ks];
      float4 xf[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) xf[m] = *reinterpret_cast<const float4*>(xr + 16 * m);
      f32x4v q0 = zero4(), q1 = zero4();
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        q0 = mfma16(wq[4 * m + 0], xf[m].x, q0);
        q1 = mfma16(wq[4 * m + 1], xf[m].y, q1);
        q0 = mfma16(wq[4 * m + 2], xf[m].z, q0);
        q1 = mfma16(wq[4 * m + 3], xf[m].w, q1);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) q[r] = (q0[r] + q1[r] + bq[r]) * scale_log2e;
    }
    // scores of the block's queries against EVERY key: s[kb][r] = S^T[key 16 kb + 4 ks + r][query i16]
    f32x4v s[NB];
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
      s[kb] = zero4();
#pragma unroll
      for (int r = 0; r < 4; ++r) s[kb] = mfma16(kt[kb][r], q[r], s[kb]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if ((NB - 1) * 16 + 4 * ks + r >= len) s[NB - 1][r] = -1e30f;
    float mx = -1e30f;
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) mx = fmaxf(mx, fmaxf(fmaxf(s[kb][0], s[kb][1]), fmaxf(s[kb][2], s[kb][3])));
    mx = quad_rows_max(mx);
    float sum0 = 0.f, sum1 = 0.f;
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
#pragma unroll
      for (int r = 0; r < 4; ++r) s[kb][r] = fast_exp2(s[kb][r] - mx);
      sum0 += s[kb][0] + s[kb][1];
      sum1 += s[kb][2] + s[kb][3];
    }
    const float inv = fast_rcp(quad_rows_sum(sum0 + sum1));
    // O^T = V^T P^T over all key blocks (two chains), normalised per query (lane = query)
    f32x4v o0 = zero4(), o1 = zero4();
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (kb & 1) o1 = mfma16(vv[kb][r], s[kb][r], o1);
        else o0 = mfma16(vv[kb][r], s[kb][r], o0);
      }
    }
    float o[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = (o0[r] + o1[r]) * inv;                      // O[query i16][d = 4 ks + r]
    // this head's share of the out-projection: Y_h[query][16 jt + i16], four independent column tiles
    float* Pw = P + ((qb & 1) * 4 + w) * 16 * LDP;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      f32x4v y = zero4();
#pragma unroll
      for (int r = 0; r < 4; ++r) y = mfma16(o[r], wo[jt][r], y);
#pragma unroll
      for (int r = 0; r < 4; ++r) Pw[(4 * ks + r) * LDP + 16 * jt + i16] = y[r];
    }
    __syncthreads();
    // row-space epilogue: y1 = LayerNorm(sum_h Y_h + b_o + x); the residual row comes from the staged tile.  The partial
    // tiles are double buffered: the next block's stores go to the other buffer, and the block after that writes this
    // one only behind the next barrier, which every thread reaches after this read.
    {
      const float* pr = P + (qb & 1) * 4 * 16 * LDP + rrow * LDP + 4 * c4;
      const float4 a0 = *reinterpret_cast<const float4*>(pr);
      const float4 a1 = *reinterpret_cast<const float4*>(pr + 16 * LDP);
      const float4 a2 = *reinterpret_cast<const float4*>(pr + 32 * LDP);
      const float4 a3 = *reinterpret_cast<const float4*>(pr + 48 * LDP);
      const float4 xres = *reinterpret_cast<const float4*>(&Xs[(qb * 16 + rrow) * LDX + 4 * c4]);
      float4 v;
      v.x = ((a0.x + a1.x) + (a2.x + a3.x)) + bo.x + xres.x;
      v.y = ((a0.y + a1.y) + (a2.y + a3.y)) + bo.y + xres.y;
      v.z = ((a0.z + a1.z) + (a2.z + a3.z)) + bo.z + xres.z;
      v.w = ((a0.w + a1.w) + (a2.w + a3.w)) + bo.w + xres.w;
      v = layernorm_row16(v, ga, be);
      const int p = qb * 16 + rrow;
      if (p < len) *reinterpret_cast<float4*>(y1 + (tok0 + (int64_t)p * tstride) * N + 4 * c4) = v;
    }
  }
}

template <int NB>
int launch_nb(hipStream_t st, bool p, size_t lds, int dev, const float* x, const float* w_in, const float* b_in, const float* w_o,
              const float* b_o, const float* gamma, const float* beta, float* y1, const SeqGeom& g, float scale, const FfnPro64& fp) {
  auto kern = p ? attn_block64_kernel<NB, true> : attn_block64_kernel<NB, false>;
  static PerDeviceOnce ready[2];
  if (!ready[p ? 1 : 0].done(dev)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    ready[p ? 1 : 0].set(dev);
  }
  hipLaunchKernelGGL(kern, dim3(g.nseq), dim3(256), lds, st, x, w_in, b_in, w_o, b_o, gamma, beta, y1, g, scale, fp);
  return (int)hipGetLastError();
}

}  // namespace

size_t attn_block64_lds_bytes(int nb) {
  const size_t prologue = 16 * LDHC + 16 * LDP, partials = 2 * 4 * 16 * LDP;
  return sizeof(float) * ((size_t)nb * 16 * LDX + (prologue > partials ? prologue : partials));
}

int attn_block64_launch(void* stream, const float* x, const float* w_in, const float* b_in, const float* w_o, const float* b_o,
                        const float* gamma, const float* beta, float* y1, const SeqGeom& g, const AttnFfnPrologue* pro) {
  const int nb = (g.len + 15) / 16;
  if (nb < 1 || nb > ATTN_BLOCK_MAX_LEN / 16) return (int)hipErrorInvalidValue;
  const float scale = 1.4426950408889634f / sqrtf((float)DH);
  const int dev = current_hip_device();
  const size_t lds = attn_block64_lds_bytes(nb);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool p = pro != nullptr;
  FfnPro64 fp{};
  if (p) fp = FfnPro64{pro->hc, pro->wf, pro->bf, pro->g2, pro->b2};
  switch (nb) {
    case 1: return launch_nb<1>(st, p, lds, dev, x, w_in, b_in, w_o, b_o, gamma, beta, y1, g, scale, fp);
    case 2: return launch_nb<2>(st, p, lds, dev, x, w_in, b_in, w_o, b_o, gamma, beta, y1, g, scale, fp);
    case 3: return launch_nb<3>(st, p, lds, dev, x, w_in, b_in, w_o, b_o, gamma, beta, y1, g, scale, fp);
    case 4: return launch_nb<4>(st, p, lds, dev, x, w_in, b_in, w_o, b_o, gamma, beta, y1, g, scale, fp);
    case 5: return launch_nb<5>(st, p, lds, dev, x, w_in, b_in, w_o, b_o, gamma, beta, y1, g, scale, fp);
    case 6: return launch_nb<6>(st, p, lds, dev, x, w_in, b_in, w_o, b_o, gamma, beta, y1, g, scale, fp);
    case 7: return launch_nb<7>(st, p, lds, dev, x, w_in, b_in, w_o, b_o, gamma, beta, y1, g, scale, fp);
    case 8: return launch_nb<8>(st, p, lds, dev, x, w_in, b_in, w_o, b_o, gamma, beta, y1, g, scale, fp);
    case 9: return launch_nb<9>(st, p, lds, dev, x, w_in, b_in, w_o, b_o, gamma, beta, y1, g, scale, fp);
    default: return launch_nb<10>(st, p, lds, dev, x, w_in, b_in, w_o, b_o, gamma, beta, y1, g, scale, fp);
  }
}
