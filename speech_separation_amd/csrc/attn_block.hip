// attn_block.hip -- the attention half of one TransformerDPRNN in ONE kernel (inference, N = 128, 4 heads of 32):
//
//     y1 = LayerNorm1( MHA(x) W_o^T + b_o + x )        src/model/dptn.py:46-47 (nn.MultiheadAttention :16-21, ln1 :22)
//
// i.e. what dptnav.hip otherwise runs as three launches (K1 in-projection GEMM, K2 attention, K3 out-projection +
// residual + LayerNorm) with QKV[M][384] and ATT[M][128] going through HBM in between: 5.5 kB of traffic per token and
// path become 1 kB (read x, write y1).  Own translation unit: compiled with -mllvm -amdgpu-mfma-vgpr-form (build.py) so
// that the score tiles the softmax works on live in architectural VGPRs.
//
// One workgroup = one sequence (len <= 160 positions), wave w = head w.  Nothing but partial sums touches LDS:
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
// Weights (256 KiB for in- and out-projection) are streamed from L2 per 32-token block; x rows are read twice per head.
#include <hip/hip_runtime.h>

#include "attn_block.h"

namespace {

constexpr int N = 128, DH = 32;
constexpr int LDP = 136;                    // partial-tile row stride: rows 4 apart (the two lane halves) are 32 banks apart

DEV float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }

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

template <int NKB>
__global__ __launch_bounds__(256) void attn_block_kernel(const float* __restrict__ x, const float* __restrict__ w_in,
                                                             const float* __restrict__ b_in, const float* __restrict__ w_o,
                                                             const float* __restrict__ b_o, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float* __restrict__ y1, SeqGeom g,
                                                             float scale_log2e) {
  extern __shared__ __attribute__((aligned(16))) float P[];   // [4 heads][32 rows][LDP]
  const int tid = threadIdx.x;
  const int h = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave = head
  const int lane = tid & 63, c = lane & 31, hh = lane >> 5;
  const int seq = blockIdx.x, len = g.len;
  const int64_t tok0 = seq_token_base(g, seq);
  const int tstride = seq_token_stride(g);
  // the lane's token row of block b (positions beyond len read the last real one: finite, masked / never stored)
  auto xrow = [&](int b) {
    const int p = b * 32 + c;
    return x + (tok0 + (int64_t)(p < len ? p : len - 1) * tstride) * N + 4 * hh;
  };
  const float* wq = w_in + (int64_t)(0 * N + h * DH + c) * N + 4 * hh;
  const float* wk = w_in + (int64_t)(1 * N + h * DH + c) * N + 4 * hh;
  const float* wv = w_in + (int64_t)(2 * N + h * DH + c) * N + 4 * hh;

  // ---- phase 1: K^T and V of this head for every key block, kept in registers --------------------------------
  // Operands stream in chunks of two k-groups (x, W_k, W_v: 6 x 16 bytes per lane and chunk), fetched ONE chunk ahead of
  // the 16 MFMAs that consume them; the scheduling barriers pin that order -- left alone the scheduler hoists a whole
  // block's 48 loads (192 registers) to the top and the allocator spills the K^T / V tiles.
  f32x16 kt[NKB], vv[NKB];
  {
    const float bv = b_in[2 * N + h * DH + c];                  // V tile: column d = c
    float bk[16];                                               // K^T tile: row d = ROW32(r,hh) = 8j + 4hh + i
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 t = ldg4(b_in + N + h * DH + 8 * j + 4 * hh);
      bk[4 * j + 0] = t.x; bk[4 * j + 1] = t.y; bk[4 * j + 2] = t.z; bk[4 * j + 3] = t.w;
    }
    constexpr int NCH = 8 * NKB;                                // chunks of the whole phase, across blocks
    float4 xb[2][2], kb_[2][2], vb[2][2];
    auto fetch = [&](int ch, int buf) {
      const int rb = ch >> 3, m0 = 2 * (ch & 7);
      const float* xr = xrow(rb);
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        xb[buf][m] = ldg4(xr + 8 * (m0 + m));
        kb_[buf][m] = ldg4(wk + 8 * (m0 + m));
        vb[buf][m] = ldg4(wv + 8 * (m0 + m));
      }
    };
    fetch(0, 0);
    f32x16 ka = zero16(), va = zero16();
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int buf = ch & 1;
      if (ch + 1 < NCH) fetch(ch + 1, buf ^ 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const float xa[4] = {xb[buf][m].x, xb[buf][m].y, xb[buf][m].z, xb[buf][m].w};
        const float ka4[4] = {kb_[buf][m].x, kb_[buf][m].y, kb_[buf][m].z, kb_[buf][m].w};
        const float va4[4] = {vb[buf][m].x, vb[buf][m].y, vb[buf][m].z, vb[buf][m].w};
#pragma unroll
        for (int t = 0; t < 4; ++t) {                           // two independent chains
          ka = mfma32(ka4[t], xa[t], ka);
          va = mfma32(xa[t], va4[t], va);
        }
      }
      if ((ch & 7) == 7) {                                      // block finished: bias, park the tiles
        const int rb = ch >> 3;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          kt[rb][r] = ka[r] + bk[r];
          vv[rb][r] = va[r] + bv;
          // parked in the AGPR half of the register file (one wave per SIMD: 256 + 256 registers per lane); the MFMAs of
          // phase 2 read their A operand there, the architectural VGPRs stay free for the streaming operands
          asm volatile("" : "+a"(kt[rb][r]), "+a"(vv[rb][r]));
        }
        ka = zero16();
        va = zero16();
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // per-lane constants of the row-space epilogue (thread = (row in pass, 4 columns)); re-read per block from L1
  const int c4 = tid & 31, rsub = tid >> 5;

  // ---- phase 2: one query block at a time -----------------------------------------------------------------
  for (int qb = 0; qb < NKB; ++qb) {
    // Q^T tile (operands one chunk ahead, as in phase 1), bias, scale by log2(e)/sqrt(dh)
    f32x16 q;
    {
      const float* xr = xrow(qb);
      float4 xb[2][2], wb[2][2];
      auto fetch = [&](int ch, int buf) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          xb[buf][m] = ldg4(xr + 8 * (2 * ch + m));
          wb[buf][m] = ldg4(wq + 8 * (2 * ch + m));
        }
      };
      fetch(0, 0);
      f32x16 q0 = zero16(), q1 = zero16();
#pragma unroll
      for (int ch = 0; ch < 8; ++ch) {
        const int buf = ch & 1;
        if (ch + 1 < 8) fetch(ch + 1, buf ^ 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          q0 = mfma32(wb[buf][m].x, xb[buf][m].x, q0);
          q1 = mfma32(wb[buf][m].y, xb[buf][m].y, q1);
          q0 = mfma32(wb[buf][m].z, xb[buf][m].z, q0);
          q1 = mfma32(wb[buf][m].w, xb[buf][m].w, q1);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 t = ldg4(b_in + h * DH + 8 * j + 4 * hh);
        q[4 * j + 0] = (q0[4 * j + 0] + q1[4 * j + 0] + t.x) * scale_log2e;
        q[4 * j + 1] = (q0[4 * j + 1] + q1[4 * j + 1] + t.y) * scale_log2e;
        q[4 * j + 2] = (q0[4 * j + 2] + q1[4 * j + 2] + t.z) * scale_log2e;
        q[4 * j + 3] = (q0[4 * j + 3] + q1[4 * j + 3] + t.w) * scale_log2e;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // streaming softmax over the key blocks; O^T accumulated transposed (lane = query).  One wave per SIMD: the score
    // tile of block kb+1 is computed between the MFMAs of O^T += V^T P^T of block kb (two independent chains).
    float mrun = -1e30f, lrun = 0.f;
    f32x16 o = zero16();
    f32x16 s = zero16();
#pragma unroll
    for (int r = 0; r < 16; ++r) s = mfma32(kt[0][r], q[r], s);          // S^T[key ROW32(.,hh)][query c]
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
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
      f32x16 p;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        p[r] = fast_exp2(s[r] - mnew);
        sum += p[r];
      }
      sum = half_sum(sum);
      lrun = lrun * alpha + sum;
      mrun = mnew;
      if (kb > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] *= alpha;
      }
      __builtin_amdgcn_sched_barrier(0);
      if (kb + 1 < NKB) {
        s = zero16();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          o = mfma32(vv[kb][r], p[r], o);                                 // O^T[d ROW32(.,hh)][query c]
          s = mfma32(kt[kb + 1][r], q[r], s);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) o = mfma32(vv[kb][r], p[r], o);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    {
      const float inv = fast_rcp(lrun);
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] *= inv;
    }
    // this head's share of the out-projection, two column tiles at a time (two chains; the next pair's weights in
    // flight): Y_h[query][32 jt + c]
    float* Pw = P + h * 32 * LDP;
    {
      float4 wo[2][2][4];
      auto fetch = [&](int jp, int buf) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const float* wr = w_o + (int64_t)(32 * (2 * jp + u) + c) * N + h * DH + 4 * hh;
#pragma unroll
          for (int j = 0; j < 4; ++j) wo[buf][u][j] = ldg4(wr + 8 * j);
        }
      };
      fetch(0, 0);
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {
        if (jp == 0) fetch(1, 1);
        __builtin_amdgcn_sched_barrier(0);
        f32x16 ya = zero16(), yb = zero16();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          ya = mfma32(o[4 * j + 0], wo[jp][0][j].x, ya);
          yb = mfma32(o[4 * j + 0], wo[jp][1][j].x, yb);
          ya = mfma32(o[4 * j + 1], wo[jp][0][j].y, ya);
          yb = mfma32(o[4 * j + 1], wo[jp][1][j].y, yb);
          ya = mfma32(o[4 * j + 2], wo[jp][0][j].z, ya);
          yb = mfma32(o[4 * j + 2], wo[jp][1][j].z, yb);
          ya = mfma32(o[4 * j + 3], wo[jp][0][j].w, ya);
          yb = mfma32(o[4 * j + 3], wo[jp][1][j].w, yb);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          Pw[ROW32(r, hh) * LDP + 32 * (2 * jp) + c] = ya[r];
          Pw[ROW32(r, hh) * LDP + 32 * (2 * jp + 1) + c] = yb[r];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
    // row-space epilogue: y1 = LayerNorm(sum_h Y_h + b_o + x); a row = 32 adjacent lanes, 16 bytes per lane
    {
      const float4 bo = ldg4(b_o + 4 * c4), ga = ldg4(gamma + 4 * c4), be = ldg4(beta + 4 * c4);
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const int row = pass * 8 + rsub;
        const int p = qb * 32 + row;
        const bool ok = p < len;
        const int64_t tok = tok0 + (int64_t)(ok ? p : len - 1) * tstride;
        const float4 xres = ldg4(x + tok * N + 4 * c4);
        const float* pr = P + row * LDP + 4 * c4;
        const float4 a0 = *reinterpret_cast<const float4*>(pr);
        const float4 a1 = *reinterpret_cast<const float4*>(pr + 32 * LDP);
        const float4 a2 = *reinterpret_cast<const float4*>(pr + 64 * LDP);
        const float4 a3 = *reinterpret_cast<const float4*>(pr + 96 * LDP);
        float4 v;
        v.x = ((a0.x + a1.x) + (a2.x + a3.x)) + bo.x + xres.x;
        v.y = ((a0.y + a1.y) + (a2.y + a3.y)) + bo.y + xres.y;
        v.z = ((a0.z + a1.z) + (a2.z + a3.z)) + bo.z + xres.z;
        v.w = ((a0.w + a1.w) + (a2.w + a3.w)) + bo.w + xres.w;
        const float mu = group_sum<32>((v.x + v.y) + (v.z + v.w)) * (1.0f / N);
        const float dx = v.x - mu, dy = v.y - mu, dz = v.z - mu, dw = v.w - mu;
        const float var = group_sum<32>((dx * dx + dy * dy) + (dz * dz + dw * dw)) * (1.0f / N);
        const float rstd = rsqrtf(var + 1e-5f);
        if (ok) {
          float4 y;
          y.x = dx * rstd * ga.x + be.x;
          y.y = dy * rstd * ga.y + be.y;
          y.z = dz * rstd * ga.z + be.z;
          y.w = dw * rstd * ga.w + be.w;
          *reinterpret_cast<float4*>(y1 + tok * N + 4 * c4) = y;
        }
      }
    }
    __syncthreads();   // the partial tiles are rewritten by the next query block
  }
}

}  // namespace

size_t attn_block_lds_bytes() { return sizeof(float) * 4 * 32 * LDP; }

int attn_block_launch(void* stream, const float* x, const float* w_in, const float* b_in, const float* w_o, const float* b_o,
                      const float* gamma, const float* beta, float* y1, const SeqGeom& g) {
  using Kern = void (*)(const float*, const float*, const float*, const float*, const float*, const float*, const float*, float*,
                        SeqGeom, float);
  Kern kern;
  const int nkb = (g.len + 31) / 32;
  switch (nkb) {
    case 1: kern = attn_block_kernel<1>; break;
    case 2: kern = attn_block_kernel<2>; break;
    case 3: kern = attn_block_kernel<3>; break;
    case 4: kern = attn_block_kernel<4>; break;
    case 5: kern = attn_block_kernel<5>; break;
    default: return (int)hipErrorInvalidValue;
  }
  static PerDeviceOnce ready[6];
  const int dev = current_hip_device();
  const size_t lds = attn_block_lds_bytes();
  if (!ready[nkb].done(dev)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    ready[nkb].set(dev);
  }
  const float scale_log2e = 1.4426950408889634f / sqrtf((float)DH);
  hipLaunchKernelGGL(kern, dim3(g.nseq), dim3(256), lds, static_cast<hipStream_t>(stream), x, w_in, b_in, w_o, b_o, gamma, beta,
                     y1, g, scale_log2e);
  return (int)hipGetLastError();
}
