// backward_ends.h -- backward of the bandwidth-bound ends of the path (tail: decoder / post-processing / overlap-add /
// separation conv / PReLU; head: chunking, video gate, encoder) for the training step.  fp32, no atomics: every
// parameter gradient is a per-workgroup partial + fixed-order reduce.
#pragma once
#include "backward.h"
#include "headtail.h"
#include "sisnr.h"

// ------------------------------------------------------------------------------------------------
// tail, GEMM-engine epilogue: recompute q = u W_post^T + b_post + E (what the decoder consumed), then
//   dD[j]  = dy[spk][b][pad_left + stride*t + j]                      (transposed-conv gather, dptn_wav.py:188-192)
//   dq[c]  = sum_j dD[j] wdec[c][j]                                   -> DQ (gradient w.r.t. q, also d E)
//   dwdec[c][j] += q[c] dD[j]                                         -> per-workgroup partials [N][8]
// ------------------------------------------------------------------------------------------------
template <int GROUP>
struct EpiDecoderBwd {
  static constexpr bool DIRECT = false;
  static constexpr bool HAS_FINISH = true;
  float* DQ;            // (2*B*L, N)
  const float* bias;    // postprocessing bias
  const float* E;       // (B*L, N)
  const float* wdec;    // (N,1,k)
  const float* dy1;     // (B,T) gradient of s1_pred
  const float* dy2;
  float* partials;      // [gridDim.x][N*8]
  int64_t BL, T;
  int L, kenc, stride, pad_left, bm;
  float wacc[4][8] = {};
  DEV float4 prefetch(int tile, int row, int c4) const {
    const int64_t r = (int64_t)tile * bm + row;
    const int64_t e = r < BL ? r : (r < 2 * BL ? r - BL : 0);   // r mod BL for the two speakers' rows
    return *reinterpret_cast<const float4*>(E + e * (4 * GROUP) + 4 * c4);
  }
  DEV void row(int tile, int row, int /*colgroup*/, int c4, float4 v, float4 x) {
    const int64_t r = (int64_t)tile * bm + row;
    if (r >= 2 * BL) return;
    const float4 b = *reinterpret_cast<const float4*>(bias + 4 * c4);
    const float q[4] = {v.x + b.x + x.x, v.y + b.y + x.y, v.z + b.z + x.z, v.w + b.w + x.w};
    const int spk = r >= BL ? 1 : 0;
    const int rem = (int)(r - (int64_t)spk * BL);
    const int bb = fast_div(rem, L, 1.0f / (float)L);
    const int t = rem - bb * L;
    const float* dy = (spk ? dy2 : dy1) + (int64_t)bb * T;
    float dq[4] = {0.f, 0.f, 0.f, 0.f};
    // (taps unrolled to the table width with a predicate: a runtime trip count made wacc[i][j] a dynamically indexed
    //  array, i.e. 232 bytes of scratch per lane and 0.07 MFMA-busy for the whole launch)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool tap = j < kenc;
      const int64_t n = (int64_t)pad_left + (int64_t)stride * t + j;
      const float d = (tap && n >= 0 && n < T) ? dy[n] : 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        dq[i] = fmaf(d, tap ? wdec[(4 * c4 + i) * kenc + j] : 0.f, dq[i]);
        wacc[i][j] = fmaf(q[i], d, wacc[i][j]);
      }
    }
    *reinterpret_cast<float4*>(DQ + r * (4 * GROUP) + 4 * c4) = make_float4(dq[0], dq[1], dq[2], dq[3]);
  }
  DEV void finish(float* smem, int tid) {
    // threads with equal c4 = tid % GROUP own the same channels: reduce them through LDS, one tap at a time
    __syncthreads();
    float* red = smem;   // [256][4]
    float* p = partials + (size_t)blockIdx.x * (4 * GROUP * 8);
    for (int j = 0; j < 8; ++j) {
      *reinterpret_cast<float4*>(red + 4 * tid) = make_float4(wacc[0][j], wacc[1][j], wacc[2][j], wacc[3][j]);
      __syncthreads();
      if (tid < GROUP) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = tid; k < 256; k += GROUP) {
          const float4 u = *reinterpret_cast<const float4*>(red + 4 * k);
          a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
        }
        p[(4 * tid + 0) * 8 + j] = a.x;
        p[(4 * tid + 1) * 8 + j] = a.y;
        p[(4 * tid + 2) * 8 + j] = a.z;
        p[(4 * tid + 3) * 8 + j] = a.w;
      }
      __syncthreads();
    }
  }
};

// scatter the [N][8] tap-gradient table into the (N,1,k) layout of decoder.weight
__global__ void decoder_wgrad_finish_kernel(const float* __restrict__ table, float* __restrict__ grad, int N, int kenc) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N * kenc) grad[i] = table[(i / kenc) * 8 + (i % kenc)];
}

// overlap-add backward: dZ[m][spk*N + c] = dU[(spk*B + b)*L + P*s + k + left][c]      (dprnn.py:145-163, dptn_wav.py:52-57)
__global__ __launch_bounds__(256) void ola_grad_gather_kernel(const float* __restrict__ DU, float* __restrict__ DZ, int N,
                                                               int B, int L, int S, int K, int P, int left) {
  const int64_t m = blockIdx.x;     // token
  const int n4 = 2 * N / 4;
  const int b = (int)(m / ((int64_t)S * K));
  const int sk = (int)(m - (int64_t)b * S * K);
  const int s = sk / K, k = sk - s * K;
  const int t = P * s + k + left;
  for (int i = threadIdx.x; i < n4; i += blockDim.x) {
    const int spk = (4 * i) / N, cc = 4 * i - spk * N;
    *reinterpret_cast<float4*>(DZ + m * (2 * N) + 4 * i) =
        *reinterpret_cast<const float4*>(DU + (((int64_t)spk * B + b) * L + t) * N + cc);
  }
}

// PReLU backward as the epilogue of the separation-conv data gradient (single shared slope, dptn_wav.py:27):
//   dx = g * (x > 0 ? 1 : a);  da += g * x * (x <= 0)
struct EpiPReLUBwd {
  static constexpr bool DIRECT = false;
  static constexpr bool HAS_FINISH = true;
  float* dx;            // (M, N)
  const float* x;       // (M, N) forward input of the PReLU
  const float* slope;
  float* partials;      // [gridDim.x]
  int64_t M;
  int ld, bm;
  float sacc = 0.f;
  DEV float4 prefetch(int tile, int row, int c4) const {
    const int64_t r = (int64_t)tile * bm + row;
    return r < M ? *reinterpret_cast<const float4*>(x + r * ld + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  DEV void row(int tile, int row, int /*colgroup*/, int c4, float4 v, float4 xv) {
    const int64_t r = (int64_t)tile * bm + row;
    if (r >= M) return;
    const float a = *slope;
    const float g[4] = {v.x, v.y, v.z, v.w}, xx[4] = {xv.x, xv.y, xv.z, xv.w};
    float o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      o[i] = xx[i] > 0.f ? g[i] : a * g[i];
      sacc += xx[i] > 0.f ? 0.f : g[i] * xx[i];
    }
    *reinterpret_cast<float4*>(dx + r * ld + 4 * c4) = make_float4(o[0], o[1], o[2], o[3]);
  }
  DEV void finish(float* smem, int tid) {
    __syncthreads();
    smem[tid] = sacc;
    __syncthreads();
    if (tid == 0) {
      float s = 0.f;
      for (int k = 0; k < 256; ++k) s += smem[k];
      partials[blockIdx.x] = s;
    }
  }
};

// ------------------------------------------------------------------------------------------------
// head backward, step 1: one thread group per frame (b, l):
//   dE = DQ[spk 0] + DQ[spk 1] (decoder skip) + chunking backward (dX0 of the <= 2 chunks holding the frame)
//   video gate: fused = enc + tanh(gate) * LN(interp(vid))  ->  d tanh(gate), d LN params (partials), d interp(vid)
// ------------------------------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(256) void head_bwd_frames_kernel(const float* __restrict__ DQ, const float* __restrict__ dX0,
                                                               const float* __restrict__ vid, const float* __restrict__ gate,
                                                               const float* __restrict__ ln_w, float* __restrict__ DE,
                                                               float* __restrict__ DVI, float* __restrict__ partials,
                                                               int B, int L, int Tv, int S, int K, int P) {
  constexpr int GROUP = N / 4, FPB = 256 / GROUP;
  __shared__ float4 red[3][256];
  const int b = blockIdx.y;
  const int l = blockIdx.x * FPB + threadIdx.x / GROUP;
  const int c4 = threadIdx.x % GROUP;
  const bool ok = l < L;
  const int lc = ok ? l : L - 1;
  const int64_t BL = (int64_t)B * L;
  float4 de = make_float4(0.f, 0.f, 0.f, 0.f);
  {
    const float4 a = *reinterpret_cast<const float4*>(DQ + ((int64_t)b * L + lc) * N + 4 * c4);
    const float4 c = *reinterpret_cast<const float4*>(DQ + (BL + (int64_t)b * L + lc) * N + 4 * c4);
    de = make_float4(a.x + c.x, a.y + c.y, a.z + c.z, a.w + c.w);
    int s_hi = lc / P;
    if (s_hi > S - 1) s_hi = S - 1;
    for (int s = s_hi; s >= 0 && lc - P * s < K; --s) {
      const float4 g = *reinterpret_cast<const float4*>(dX0 + (((int64_t)b * S + s) * K + (lc - P * s)) * N + 4 * c4);
      de.x += g.x; de.y += g.y; de.z += g.z; de.w += g.w;
    }
  }
  if (!ok) de = make_float4(0.f, 0.f, 0.f, 0.f);
  if (ok) *reinterpret_cast<float4*>(DE + ((int64_t)b * L + l) * N + 4 * c4) = de;
  float4 sgam = make_float4(0.f, 0.f, 0.f, 0.f), sbet = sgam;
  float sgate = 0.f;
  if (vid != nullptr) {
    const float scale = (float)Tv / (float)L;
    float src = ((float)lc + 0.5f) * scale - 0.5f;
    src = src < 0.f ? 0.f : src;
    const int i0 = (int)floorf(src);
    const int i1 = i0 + 1 < Tv ? i0 + 1 : Tv - 1;
    const float lam = src - (float)i0;
    const float4 a = *reinterpret_cast<const float4*>(vid + ((int64_t)b * Tv + i0) * N + 4 * c4);
    const float4 bb = *reinterpret_cast<const float4*>(vid + ((int64_t)b * Tv + i1) * N + 4 * c4);
    float u[4] = {a.x * (1.f - lam) + bb.x * lam, a.y * (1.f - lam) + bb.y * lam, a.z * (1.f - lam) + bb.z * lam,
                  a.w * (1.f - lam) + bb.w * lam};
    const float mu = group_sum<GROUP>((u[0] + u[1]) + (u[2] + u[3])) * (1.0f / N);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { u[i] -= mu; q += u[i] * u[i]; }
    const float rstd = rsqrtf(group_sum<GROUP>(q) * (1.0f / N) + 1e-5f);
    const float tg = tanhf(*gate);
    const float4 ga = *reinterpret_cast<const float4*>(ln_w + 4 * c4);
    const float dv[4] = {de.x, de.y, de.z, de.w}, gam[4] = {ga.x, ga.y, ga.z, ga.w};
    float zn[4], g[4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      zn[i] = u[i] * rstd;
      // d tanh(gate) needs the LN output itself: zn*gamma + beta; beta's part is added on the host side reduce
      g[i] = tg * dv[i] * gam[i];
      s1 += g[i];
      s2 += g[i] * zn[i];
    }
    const float m1 = group_sum<GROUP>(s1) * (1.0f / N), m2 = group_sum<GROUP>(s2) * (1.0f / N);
    float o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = rstd * (g[i] - m1 - zn[i] * m2);
    if (ok) *reinterpret_cast<float4*>(DVI + ((int64_t)b * L + l) * N + 4 * c4) = make_float4(o[0], o[1], o[2], o[3]);
    // per-thread partials: d gamma_v = tg*dE*zn, d beta_v = tg*dE, d tanh(gate) = sum dE*(zn*gamma + beta) -- the beta
    // term equals <d beta_v, beta>/tg and is formed from the reduced d beta_v on the host side kernel
    sgam = make_float4(tg * dv[0] * zn[0], tg * dv[1] * zn[1], tg * dv[2] * zn[2], tg * dv[3] * zn[3]);
    sbet = make_float4(tg * dv[0], tg * dv[1], tg * dv[2], tg * dv[3]);
    sgate = dv[0] * zn[0] * gam[0] + dv[1] * zn[1] * gam[1] + dv[2] * zn[2] * gam[2] + dv[3] * zn[3] * gam[3];
  }
  // block partials: [blk][0:N) d gamma_v | [N:2N) d beta_v | [2N] d tanh(gate) without the beta term
  red[0][threadIdx.x] = sgam;
  red[1][threadIdx.x] = sbet;
  red[2][threadIdx.x] = make_float4(sgate, 0.f, 0.f, 0.f);
  __syncthreads();
  float* p = partials + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (2 * N + 4);
  if (threadIdx.x < GROUP) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), c = a;
    for (int k = threadIdx.x; k < 256; k += GROUP) {
      a.x += red[0][k].x; a.y += red[0][k].y; a.z += red[0][k].z; a.w += red[0][k].w;
      c.x += red[1][k].x; c.y += red[1][k].y; c.z += red[1][k].z; c.w += red[1][k].w;
    }
    *reinterpret_cast<float4*>(p + 4 * threadIdx.x) = a;
    *reinterpret_cast<float4*>(p + N + 4 * threadIdx.x) = c;
  }
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int k = 0; k < 256; ++k) s += red[2][k].x;
    p[2 * N] = s;
    p[2 * N + 1] = p[2 * N + 2] = p[2 * N + 3] = 0.f;
  }
}

// finish the gate gradient: d gate = (1 - tanh(gate)^2) * (S_gate + <d beta_v, beta_v> / tanh(gate)), where the
// reduced vector holds [d gamma_v | d beta_v | S_gate]
__global__ void gate_grad_finish_kernel(const float* __restrict__ reduced, const float* __restrict__ gate,
                                        const float* __restrict__ ln_b, float* __restrict__ g_gate,
                                        float* __restrict__ g_lnw, float* __restrict__ g_lnb, int N) {
  __shared__ float red[256];
  const float tg = tanhf(*gate);
  float s = 0.f;
  for (int i = threadIdx.x; i < N; i += blockDim.x) {
    g_lnw[i] = reduced[i];
    g_lnb[i] = reduced[N + i];
    // d beta_v = tg * sum dE  ->  sum dE * beta = d beta_v * beta / tg
    s += (tg != 0.f ? reduced[N + i] / tg : 0.f) * ln_b[i];
  }
  red[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = reduced[2 * N];
    for (int k = 0; k < (int)blockDim.x; ++k) t += red[k];
    *g_gate = t * (1.f - tg * tg);
  }
}

// encoder weight gradient: dw[c][j] = sum_{b,l} DE[b][l][c] * mix[b][stride*l + j]; partial per block over a slab of
// frames: partials[blk][N*8]
template <int N>
__global__ __launch_bounds__(256) void encoder_wgrad_kernel(const float* __restrict__ DE, const float* __restrict__ mix,
                                                             float* __restrict__ partials, int B, int64_t T, int L,
                                                             int kenc, int stride, int frames_per_block) {
  constexpr int GROUP = N / 4, FPB = 256 / GROUP;
  __shared__ float red[256 * 4];
  const int c4 = threadIdx.x % GROUP, fl = threadIdx.x / GROUP;
  const int b = blockIdx.y;
  float acc[4][8] = {};
  const int l0 = blockIdx.x * frames_per_block;
  for (int l = l0 + fl; l < l0 + frames_per_block && l < L; l += FPB) {
    const float4 d = *reinterpret_cast<const float4*>(DE + ((int64_t)b * L + l) * N + 4 * c4);
    const float* m = mix + (int64_t)b * T + (int64_t)stride * l;
    for (int j = 0; j < kenc; ++j) {
      const float x = m[j];
      acc[0][j] = fmaf(d.x, x, acc[0][j]);
      acc[1][j] = fmaf(d.y, x, acc[1][j]);
      acc[2][j] = fmaf(d.z, x, acc[2][j]);
      acc[3][j] = fmaf(d.w, x, acc[3][j]);
    }
  }
  float* p = partials + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (N * 8);
  for (int j = 0; j < 8; ++j) {
    __syncthreads();
    *reinterpret_cast<float4*>(red + 4 * threadIdx.x) = make_float4(acc[0][j], acc[1][j], acc[2][j], acc[3][j]);
    __syncthreads();
    if (threadIdx.x < GROUP) {
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int k = threadIdx.x; k < 256; k += GROUP) {
        const float4 u = *reinterpret_cast<const float4*>(red + 4 * k);
        a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
      }
      p[(4 * threadIdx.x + 0) * 8 + j] = a.x;
      p[(4 * threadIdx.x + 1) * 8 + j] = a.y;
      p[(4 * threadIdx.x + 2) * 8 + j] = a.z;
      p[(4 * threadIdx.x + 3) * 8 + j] = a.w;
    }
  }
}

// linear-interpolation backward (transpose of F.interpolate align_corners=False): DV[b][tv][c] = sum_l w(l,tv) DVI[b][l][c]
// 512 threads = 4 window quarters x 128 channels (N <= 128); a thread walks its quarter of the
// frames that can touch tv with UNCONDITIONAL loads (weight 0 where they do not: eight in flight), the quarters are
// summed in a fixed order.  (One thread per channel walking the whole window behind a branch: 0.19 ms per launch.)
__global__ __launch_bounds__(512) void interp_bwd_kernel(const float* __restrict__ DVI, float* __restrict__ DV, int N, int L,
                                                          int Tv) {
  __shared__ float part[4][128];
  const int b = blockIdx.y, tv = blockIdx.x;
  const int c = threadIdx.x & 127, qd = threadIdx.x >> 7;
  const float scale = (float)Tv / (float)L;
  // frames whose i0 or i1 can equal tv: src in (tv-1, tv+1)  ->  l in a window
  int lo = (int)floorf(((float)tv - 1.0f + 0.5f) / scale - 0.5f) - 2;
  int hi = (int)ceilf(((float)tv + 1.0f + 0.5f) / scale - 0.5f) + 2;
  lo = lo < 0 ? 0 : lo;
  hi = hi > L - 1 ? L - 1 : hi;
  if (tv == 0) lo = 0;
  float s = 0.f;
  if (c < N) {
#pragma unroll 8
    for (int l = lo + qd; l <= hi; l += 4) {
      float src = ((float)l + 0.5f) * scale - 0.5f;
      src = src < 0.f ? 0.f : src;
      const int i0 = (int)floorf(src);
      const int i1 = i0 + 1 < Tv ? i0 + 1 : Tv - 1;
      const float lam = src - (float)i0;
      float wgt = 0.f;
      if (i0 == tv) wgt += 1.f - lam;
      if (i1 == tv) wgt += lam;
      s = fmaf(wgt, DVI[((int64_t)b * L + l) * N + c], s);
    }
  }
  part[qd][c] = s;
  __syncthreads();
  if (qd == 0 && c < N) DV[((int64_t)b * Tv + tv) * N + c] = (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]);
}

// visual_compression gradients: dW[o][cv] = sum_{b,t,spk} DV[b][t][spk*half + o] e_spk[b][cv][t]; db[o] = sum DV.
// grid (half, B): one workgroup per (output feature, mixture) writes its partial row slab[b][o][0:Cv] and its partial
// bias colslab[b][o]; slab_reduce_kernel sums over the mixtures in a fixed order.  (First version: one workgroup per
// output feature walked all B x 2 x Tv products per thread and thread 0 summed the bias alone: 0.44 ms at the very end
// of each half's chain on 64 of 256 CUs.)
__global__ __launch_bounds__(256) void video_linear_bwd_kernel(const float* __restrict__ DV, const float* __restrict__ e1,
                                                                const float* __restrict__ e2, float* __restrict__ slab,
                                                                float* __restrict__ colslab, int Cv, int Tv, int half) {
  __shared__ float dvs[2][256];          // this (mixture, feature)'s dV column of both speakers (Tv <= 256: host)
  __shared__ double red[4];
  const int o = blockIdx.x, b = blockIdx.y, nhalf = gridDim.x;
  double part = 0.0;
  for (int i = threadIdx.x; i < 2 * Tv; i += 256) {
    const int spk = i / Tv, t = i - spk * Tv;
    const float v = DV[((int64_t)b * Tv + t) * (2 * half) + spk * half + o];
    dvs[spk][t] = v;
    part += v;
  }
  part = block_sum(part, red);           // (ends with a barrier: dvs is visible)
  if (threadIdx.x == 0) colslab[(int64_t)b * nhalf + o] = (float)part;
  for (int cv = threadIdx.x; cv < Cv; cv += 256) {
    float s = 0.f;
#pragma unroll
    for (int spk = 0; spk < 2; ++spk) {
      const float* e = (spk ? e2 : e1) + ((int64_t)b * Cv + cv) * Tv;
      for (int t = 0; t < Tv; ++t) s = fmaf(dvs[spk][t], e[t], s);
    }
    slab[((int64_t)b * nhalf + o) * Cv + cv] = s;
  }
}
