// headtail.h -- bandwidth-bound ends of the path (encoder + video gate + chunking; overlap-add +
// post-processing + skip + transposed-conv decoder).  fp32, coalesced 16-byte accesses, LayerNorm
// statistics by sub-wave shuffles.
#pragma once
#include "common.h"
#include "gemm_ws.h"

// ------------------------------------------------------------------------------------------------
// video branch, step 1: Linear(Cv -> HV/2) for both speakers (shared weights), concatenated
//   reference: dptn_wav.py:173-179.   e* (B,Cv,Tv) -> vid (B,Tv,HV)
// ------------------------------------------------------------------------------------------------
//   grid (B*2, half), block 256 = ONE output feature: the four waves take a quarter of the Cv input channels each (lanes = frame
//   slots, the embedding read is coalesced over frames, the weight is wave-uniform) and meet in LDS, summed in a fixed order.
//   (Four features per workgroup with the whole contraction per wave was 50 us for a 40-MFLOP product: 512 dependent-latency
//   iterations per wave at the very start of a forward, when nothing else runs.)
__global__ __launch_bounds__(256) void video_linear_kernel(const float* __restrict__ e1,
                                                            const float* __restrict__ e2,
                                                            const float* __restrict__ W,
                                                            const float* __restrict__ bias,
                                                            float* __restrict__ vid, int Cv, int Tv, int half) {
  __shared__ float part[4][64];
  const int b = blockIdx.x >> 1, spk = blockIdx.x & 1;
  const int o = blockIdx.y, wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float* e = (spk ? e2 : e1) + (int64_t)b * Cv * Tv;
  const float* w = W + (int64_t)o * Cv;
  const int q = (Cv + 3) / 4, c0 = wv * q, c1 = c0 + q < Cv ? c0 + q : Cv;
  for (int t0 = 0; t0 < Tv; t0 += 64) {
    const int t = t0 + lane;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (t < Tv) {
      int cc = c0;
#pragma unroll 8
      for (; cc + 4 <= c1; cc += 4) {
        a0 = fmaf(w[cc + 0], e[(int64_t)(cc + 0) * Tv + t], a0);
        a1 = fmaf(w[cc + 1], e[(int64_t)(cc + 1) * Tv + t], a1);
        a2 = fmaf(w[cc + 2], e[(int64_t)(cc + 2) * Tv + t], a2);
        a3 = fmaf(w[cc + 3], e[(int64_t)(cc + 3) * Tv + t], a3);
      }
      for (; cc < c1; ++cc) a0 = fmaf(w[cc], e[(int64_t)cc * Tv + t], a0);
    }
    part[wv][lane] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (wv == 0 && t < Tv)
      vid[((int64_t)b * Tv + t) * (2 * half) + spk * half + o] = ((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane])) + bias[o];
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// head: encoder Conv1d(1->N,k,stride) + [linear interpolation Tv->L of vid, LayerNorm(N), * tanh(gate)]
//       + write the frame-major latent E[b][l][n] and the chunked tokens X[b][s][k][n]
//   reference: dptn_wav.py:180-184, dprnn.py:122-136.   GROUP = N/4 lanes per frame.
// ------------------------------------------------------------------------------------------------
constexpr int ENC_PASSES = 8;   // passes of 256 / (N/4) frames per workgroup
constexpr int ENC_KMAX = 8;     // encoder taps held in registers (kernel_size_enc is in [2, 8]: dptnav_create)
template <int N>
__global__ __launch_bounds__(256) void encoder_fuse_kernel(const float* __restrict__ mix,
                                                            const float* __restrict__ wenc,  // (N,1,k)
                                                            const float* __restrict__ vid,   // (B,Tv,N) or null
                                                            const float* __restrict__ gate,
                                                            const float* __restrict__ ln_w,
                                                            const float* __restrict__ ln_b,
                                                            float* __restrict__ E, float* __restrict__ X,
                                                            int64_t T, int L, int kenc, int stride, int Tv, int S,
                                                            int K, int P) {
  constexpr int GROUP = N / 4;
  constexpr int FPB = 256 / GROUP;  // frames per pass
  const int b = blockIdx.y;
  const int c4 = threadIdx.x % GROUP;
  // ENC_PASSES passes of FPB frames per workgroup, the encoder taps of this thread's 4 channels held in registers (one
  // pass per workgroup meant 10.7 k workgroups of a few hundred cycles each for a half batch: the launch ran at the
  // dispatch rate, 0.9 TB/s)
  float we[4][ENC_KMAX];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < ENC_KMAX; ++j) we[i][j] = j < kenc ? wenc[(4 * c4 + i) * kenc + j] : 0.f;
  float4 ga = make_float4(0.f, 0.f, 0.f, 0.f), be = ga;
  float tg = 0.f;
  if (vid != nullptr) {
    ga = *reinterpret_cast<const float4*>(ln_w + 4 * c4);
    be = *reinterpret_cast<const float4*>(ln_b + 4 * c4);
    tg = tanhf(*gate);
  }
#pragma unroll 2
  for (int pass = 0; pass < ENC_PASSES; ++pass) {
  const int l = (blockIdx.x * ENC_PASSES + pass) * FPB + threadIdx.x / GROUP;
  const bool ok = l < L;
  const int lc = ok ? l : L - 1;

  float v[4] = {0.f, 0.f, 0.f, 0.f};
  const float* m = mix + (int64_t)b * T + (int64_t)stride * lc;
#pragma unroll
  for (int j = 0; j < ENC_KMAX; ++j) {
    const float x = j < kenc ? m[j] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = fmaf(we[i][j], x, v[i]);
  }
  if (vid != nullptr) {
    // F.interpolate(mode="linear", align_corners=False)
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
    const float s = group_sum<GROUP>((u[0] + u[1]) + (u[2] + u[3]));
    const float mu = s * (1.0f / N);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      u[i] -= mu;
      q += u[i] * u[i];
    }
    q = group_sum<GROUP>(q);
    const float rstd = rsqrtf(q * (1.0f / N) + 1e-5f);
    v[0] += tg * (u[0] * rstd * ga.x + be.x);
    v[1] += tg * (u[1] * rstd * ga.y + be.y);
    v[2] += tg * (u[2] * rstd * ga.z + be.z);
    v[3] += tg * (u[3] * rstd * ga.w + be.w);
  }
  if (ok) {
    const float4 o4 = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(E + ((int64_t)b * L + l) * N + 4 * c4) = o4;
    // chunks s with P*s <= l < P*s + K   (F.unfold semantics: trailing frames belong to no chunk)
    int s_hi = l / P;
    if (s_hi > S - 1) s_hi = S - 1;
    for (int s = s_hi; s >= 0 && l - P * s < K; --s)
      *reinterpret_cast<float4*>(X + (((int64_t)b * S + s) * K + (l - P * s)) * N + 4 * c4) = o4;
  }
  }
}

// ------------------------------------------------------------------------------------------------
// tail stage 2 (GEMM-engine hooks): rows are (spk, b, frame t).  A row = overlap-add of the separated
// tokens shifted by `left` frames (dptn_wav.py:49-57, dprnn.py:145-163); the GEMM is the shared
// post-processing 1x1 conv (dptn_wav.py:31-33,59); the epilogue adds bias + fused latent E (skip,
// dptn_wav.py:188) and projects on the k decoder taps (ConvTranspose1d weights, dptn_wav.py:167-169).
// ------------------------------------------------------------------------------------------------
// Exact floor(n / d) for n >= 0, d >= 1 without an integer division where it matters (a 64-bit division costs ~100
// instructions, and the loaders / epilogues below ran three per 16-byte load: 0.13 MFMA-busy): below 2^23 the float
// conversion of n is exact and the estimate n * (1/d) is within +-1 of the quotient, fixed by one correction step each
// way (checked exhaustively against // on the CPU for the divisors in use); larger n take the hardware path.
DEV int fast_div(int n, int d, float inv_d) {
  if (n >= (1 << 23)) return n / d;
  int q = (int)((float)n * inv_d);
  const int r = n - q * d;
  q += r >= d ? 1 : 0;
  q -= r < 0 ? 1 : 0;
  return q;
}

struct ALoadOla {
  const float* Z;  // (M, 2N) separated tokens
  int N, B, L, S, K, P, left, ola, bm;
  DEV float4 load4(int tile, int row, int k4) const {
    // rows r = (spk*B + b)*L + frame, all below 2^31 (host-checked plan limit): 32-bit arithmetic, no divisions
    const int r = tile * bm + row;
    const int BL = B * L;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r >= 2 * BL) return acc;
    const int spk = r >= BL ? 1 : 0;
    const int rem = r - spk * BL;
    const int b = fast_div(rem, L, 1.0f / (float)L);
    const int t = rem - b * L - left;
    if (t < 0 || t >= ola) return acc;
    int s_hi = fast_div(t, P, 1.0f / (float)P);
    if (s_hi > S - 1) s_hi = S - 1;
    for (int s = s_hi; s >= 0 && t - P * s < K; --s) {
      const float4 z = *reinterpret_cast<const float4*>(Z + (((int64_t)b * S + s) * K + (t - P * s)) * (2 * N) +
                                                        spk * N + 4 * k4);
      acc.x += z.x; acc.y += z.y; acc.z += z.z; acc.w += z.w;
    }
    return acc;
  }
};

template <int GROUP>
struct EpiSkipDecoderTaps {
  static constexpr bool DIRECT = false;
  static constexpr bool HAS_FINISH = false;
  float* D;            // (2*B*L, 8) decoder tap products
  const float* bias;   // postprocessing bias
  const float* E;      // (B*L, N)
  const float* wdec;   // (N,1,k)
  int64_t BL;          // B*L
  int kenc, bm;
  DEV float4 prefetch(int tile, int row, int c4) const {   // fused latent row of this frame (skip connection)
    const int64_t r = (int64_t)tile * bm + row;
    const int64_t e = r < BL ? r : (r < 2 * BL ? r - BL : 0);   // r mod BL for the two speakers' rows
    return *reinterpret_cast<const float4*>(E + e * (4 * GROUP) + 4 * c4);
  }
  DEV void row(int tile, int row, int /*colgroup*/, int c4, float4 v, float4 x) const {
    const int64_t r = (int64_t)tile * bm + row;
    const bool ok = r < 2 * BL;
    const float4 b = *reinterpret_cast<const float4*>(bias + 4 * c4);
    v.x += b.x + x.x; v.y += b.y + x.y; v.z += b.z + x.z; v.w += b.w + x.w;
    float mine = 0.f;
    for (int j = 0; j < kenc; ++j) {
      float s = v.x * wdec[(4 * c4 + 0) * kenc + j] + v.y * wdec[(4 * c4 + 1) * kenc + j] +
                v.z * wdec[(4 * c4 + 2) * kenc + j] + v.w * wdec[(4 * c4 + 3) * kenc + j];
      s = group_sum<GROUP>(s);
      if (c4 == j) mine = s;
    }
    if (ok && c4 < 8) D[r * 8 + c4] = c4 < kenc ? mine : 0.f;
  }
};

// ------------------------------------------------------------------------------------------------
// tail stage 2, FOLDED form (inference): the decoder only ever sees q W_dec with q = u W_post^T + b_post + E, so the
// 1x1 post-processing conv, the skip and the k tap projections collapse into ONE contraction of length 2N per frame:
//     D[r][j] = [u_r | E_r] . [G_j | W_dec[:, j]] + bd_j,   G = W_dec^T W_post (k x N),  bd = W_dec^T b_post
// (dptn_wav.py:59,188-190: ConvTranspose1d(postprocessing(mask) + encoded)).  The GEMM-engine form above spends its time
// in an epilogue of k 32-lane reductions per row (0.13 MFMA-busy); this form is bandwidth-bound: 1.5 kB in, 32 B out
// per frame.  fold_decoder_kernel builds [G | W_dec^T | bd] from the CURRENT weights on every call (they may have
// been updated), 8 x 2N + 8 floats.
// ------------------------------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(N) void fold_decoder_kernel(const float* __restrict__ wpost, const float* __restrict__ bpost,
                                                          const float* __restrict__ wdec, int kenc, float* __restrict__ Wf) {
  // grid = 8 taps, thread = input channel k.  Wf[j][0:N] = G_j, Wf[j][N:2N] = W_dec[:, j], Wf[8][j] = bd_j  (rows j >= kenc: 0)
  __shared__ float red[N];
  const int j = blockIdx.x, k = threadIdx.x;
  float g = 0.f;
  if (j < kenc)
    for (int c = 0; c < N; ++c) g = fmaf(wdec[c * kenc + j], wpost[c * N + k], g);   // wpost row c: coalesced over k
  Wf[j * 2 * N + k] = g;
  Wf[j * 2 * N + N + k] = j < kenc ? wdec[k * kenc + j] : 0.f;
  red[k] = j < kenc ? bpost[k] * wdec[k * kenc + j] : 0.f;
  __syncthreads();
  if (k == 0) {
    float b = 0.f;
    for (int c = 0; c < N; ++c) b += red[c];
    Wf[8 * 2 * N + j] = b;
  }
}

// One workgroup = 32 frames (rows r of the (2, B, L) frame list); thread = (frame f = tid / 8, tap j = tid % 8).
template <int N>
__global__ __launch_bounds__(256) void taps_fold_kernel(const float* __restrict__ Z, const float* __restrict__ E,
                                                         const float* __restrict__ Wf, float* __restrict__ D, int B, int L,
                                                         int S, int K, int P, int left, int ola) {
  constexpr int LDU = 2 * N + 4;            // rows 4 banks apart: the wave's 8 frame rows / 8 tap rows read conflict-free
  __shared__ __attribute__((aligned(16))) float Us[32 * LDU];
  __shared__ __attribute__((aligned(16))) float Ws[8 * LDU];
  const int tid = threadIdx.x;
  const int BL = B * L, rows = 2 * BL;
  const int r0 = blockIdx.x * 32;
  for (int i = tid; i < 8 * (2 * N / 4); i += 256) {
    const int j = i / (2 * N / 4), k4 = i - j * (2 * N / 4);
    *reinterpret_cast<float4*>(&Ws[j * LDU + 4 * k4]) = *reinterpret_cast<const float4*>(Wf + j * 2 * N + 4 * k4);
  }
  const float invL = 1.0f / (float)L, invP = 1.0f / (float)P;
  constexpr int K4 = 2 * N / 4;             // float4 per staged row: [u (N) | E (N)]
  constexpr int NI = (32 * K4) / 256;       // staging slots per thread
  if (K <= 2 * P) {
    // At most two chunks cover a frame (the usual 50 % overlap).  Branch-free: every slot requests exactly two rows
    // (clamped addresses; the latent-row slots request their row twice) and masks what does not count, so that all
    // 2 x NI loads of a thread are in flight together.  With a branch per slot and a data-dependent chunk loop every
    // load was waited for before the next was issued and the kernel ran at 1.0 TB/s.
    float4 z0[NI], z1[NI];
    bool ok0[NI], ok1[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int idx = i * 256 + tid;
      const int row = idx / K4, k4 = idx - row * K4;
      const int r = r0 + row;
      const bool inr = r < rows;
      const int rc = inr ? r : rows - 1;
      const int spk = rc >= BL ? 1 : 0;
      const int rem = rc - spk * BL;
      const bool lat = k4 >= N / 4;         // fused latent row (skip connection)
      const int b = fast_div(rem, L, invL);
      const int t = rem - b * L - left;
      const bool tin = t >= 0 && t < ola;
      const int tc = tin ? t : 0;
      int s_hi = fast_div(tc, P, invP);
      if (s_hi > S - 1) s_hi = S - 1;
      const int s1 = s_hi - 1;
      const int k0 = tc - P * s_hi, k1 = tc - P * s1;
      const bool v0 = tin && k0 < K, v1 = tin && s1 >= 0 && k1 < K;
      const int64_t zo0 = (((int64_t)b * S + s_hi) * K + (v0 ? k0 : 0)) * (2 * N) + spk * N + 4 * (lat ? 0 : k4);
      const int64_t zo1 = (((int64_t)b * S + (v1 ? s1 : s_hi)) * K + (v1 ? k1 : 0)) * (2 * N) + spk * N + 4 * (lat ? 0 : k4);
      const float* p0 = lat ? E + (int64_t)rem * N + 4 * (k4 - N / 4) : Z + zo0;
      const float* p1 = lat ? p0 : Z + zo1;
      z0[i] = *reinterpret_cast<const float4*>(p0);
      z1[i] = *reinterpret_cast<const float4*>(p1);
      ok0[i] = inr && (lat || v0);
      ok1[i] = inr && !lat && v1;
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int idx = i * 256 + tid;
      const int row = idx / K4, k4 = idx - row * K4;
      const float4 m0 = mask4(z0[i], ok0[i]), m1 = mask4(z1[i], ok1[i]);
      *reinterpret_cast<float4*>(&Us[row * LDU + 4 * k4]) = make_float4(m0.x + m1.x, m0.y + m1.y, m0.z + m1.z, m0.w + m1.w);
    }
  } else {
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int idx = i * 256 + tid;
    const int row = idx / K4, k4 = idx - row * K4;
    const int r = r0 + row;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < rows) {
      const int spk = r >= BL ? 1 : 0;
      const int rem = r - spk * BL;
      if (k4 >= N / 4) {                    // fused latent row (skip connection)
        v = *reinterpret_cast<const float4*>(E + (int64_t)rem * N + 4 * (k4 - N / 4));
      } else {                              // overlap-add gather of the speaker's mask rows (ALoadOla)
        const int b = fast_div(rem, L, invL);
        const int t = rem - b * L - left;
        if (t >= 0 && t < ola) {
          int s_hi = fast_div(t, P, invP);
          if (s_hi > S - 1) s_hi = S - 1;
          for (int s = s_hi; s >= 0 && t - P * s < K; --s) {
            const float4 z = *reinterpret_cast<const float4*>(Z + (((int64_t)b * S + s) * K + (t - P * s)) * (2 * N) + spk * N + 4 * k4);
            v.x += z.x; v.y += z.y; v.z += z.z; v.w += z.w;
          }
        }
      }
    }
    *reinterpret_cast<float4*>(&Us[row * LDU + 4 * k4]) = v;
  }
  }
  __syncthreads();
  const int f = tid >> 3, j = tid & 7;
  const float* u = &Us[f * LDU];
  const float* w = &Ws[j * LDU];
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 8
  for (int k4 = 0; k4 < K4; ++k4) {
    const float4 x = *reinterpret_cast<const float4*>(u + 4 * k4), y = *reinterpret_cast<const float4*>(w + 4 * k4);
    a0 = fmaf(x.x, y.x, a0); a1 = fmaf(x.y, y.y, a1); a2 = fmaf(x.z, y.z, a2); a3 = fmaf(x.w, y.w, a3);
  }
  if (r0 + f < rows) D[(int64_t)(r0 + f) * 8 + j] = ((a0 + a1) + (a2 + a3)) + Wf[8 * 2 * N + j];
}

// tail stage 3: y[n] = sum over taps j == n (mod stride) of D[(n-j)/stride][j]; zero pad to T
//   (ConvTranspose1d scatter turned into a gather; dptn_wav.py:188-192)
__global__ __launch_bounds__(256) void decoder_gather_kernel(const float* __restrict__ D, float* __restrict__ s1,
                                                              float* __restrict__ s2, int B, int64_t T, int L,
                                                              int kenc, int stride, int pad_left) {
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y, spk = blockIdx.z;
  if (n >= T) return;
  const int64_t m = n - pad_left;
  const int64_t ndec = (int64_t)(L - 1) * stride + kenc;
  float acc = 0.f;
  if (m >= 0 && m < ndec) {
    const float* Db = D + ((int64_t)spk * B + b) * L * 8;
    for (int j = (int)(m % stride); j < kenc; j += stride) {
      const int64_t i = (m - j) / stride;
      if (i >= 0 && i < L) acc += Db[i * 8 + j];
    }
  }
  (spk ? s2 : s1)[(int64_t)b * T + n] = acc;
}
