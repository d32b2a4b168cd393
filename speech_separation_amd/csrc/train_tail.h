// train_tail.h -- the tail of the training step on the device: no host synchronisation anywhere.
//
//   PIT SI-SNR loss, forward AND backward     src/loss/ss_losses.py:21-26 (batch-level PIT), :100-114 (SiSNRLoss)
//   global-norm gradient clip                 src/trainer/base_trainer.py:383-391 (clip_grad_norm_(parameters, max_grad_norm))
//   AdamW                                     src/configs/dptn_wav_av.yaml:9-11 (torch.optim.AdamW, lr 1e-3, torch defaults)
//
// The reference runs these as ~40 small PyTorch kernels with a tensor -> bool conversion inside the loss
// (`if loss_perm_2 < loss_perm_1`) and .item() calls around them.  Here the permutation is chosen on the device by the
// kernel that writes d loss / d prediction, and clip + AdamW read the flat gradient layout of dptnav_flat_offset().
// Everything is summed in a fixed order (no float atomics): repeated steps are bit-identical.
#pragma once
#include "common.h"
#include "sisnr.h"

constexpr int PIT_STAT = 8;          // doubles per (item, pair): mx, my, D, G, Nn, loss term, pad, pad
constexpr int CLIP_PARTS = 256;      // partial sums of squares (one per workgroup of the first clip launch)

// One workgroup per (item, pair); pairs = (p1,s1) (p1,s2) (p2,s1) (p2,s2).  Two passes over the 2 x T samples (the second
// one hits L2): the noise energy |p~ - a g~|^2 is summed directly, not as a difference of energies.
__global__ __launch_bounds__(256) void pit_stats_kernel(const float* __restrict__ p1, const float* __restrict__ p2,
                                                         const float* __restrict__ s1, const float* __restrict__ s2,
                                                         int64_t T, double* __restrict__ stats) {
  __shared__ double red[4];
  const int b = blockIdx.x, pair = blockIdx.y;
  const float* x = (pair < 2 ? p1 : p2) + (int64_t)b * T;
  const float* y = ((pair & 1) ? s2 : s1) + (int64_t)b * T;
  double sx = 0, sy = 0, sxy = 0, syy = 0;
  for (int64_t i = threadIdx.x; i < T; i += 256) {
    const double xv = x[i], yv = y[i];
    sx += xv; sy += yv; sxy += xv * yv; syy += yv * yv;
  }
  sx = block_sum(sx, red); sy = block_sum(sy, red); sxy = block_sum(sxy, red); syy = block_sum(syy, red);
  const double mx = sx / (double)T, my = sy / (double)T;
  const double D = sxy - (double)T * mx * my;        // <p~, g~>
  const double G = syy - (double)T * my * my;        // |g~|^2
  const double a = D / G;
  double nn = 0;
  for (int64_t i = threadIdx.x; i < T; i += 256) {
    const double e = ((double)x[i] - mx) - a * ((double)y[i] - my);
    nn += e * e;
  }
  nn = block_sum(nn, red);
  if (threadIdx.x == 0) {
    double* o = stats + ((int64_t)b * 4 + pair) * PIT_STAT;
    o[0] = mx; o[1] = my; o[2] = D; o[3] = G; o[4] = nn;
    o[5] = -20.0 * log10((a * a * G) / nn);          // SiSNRLoss of this (prediction, target) pair, ss_losses.py:114
  }
}

// grid (ceil(T / 1024), B, 2 predictions).  Every workgroup resolves the batch-level permutation itself from the 4 B loss
// terms (same fixed-order sum everywhere), then writes its 1024 samples of d loss / d prediction:
//   loss = w sum_items -20 log10(|a g~|^2 / |e|^2),  w = grad_scale / (2 B)
//   d / d p = w (-20 / ln 10) (2 g~ / D - 2 e / |e|^2)          (g~ and e are zero-mean: the mean subtraction drops out)
__global__ __launch_bounds__(256) void pit_grad_kernel(const float* __restrict__ p1, const float* __restrict__ p2,
                                                        const float* __restrict__ s1, const float* __restrict__ s2, int B,
                                                        int64_t T, const double* __restrict__ stats, float grad_scale,
                                                        float* __restrict__ d1, float* __restrict__ d2,
                                                        float* __restrict__ loss_out) {
  __shared__ double red[4];
  double a1 = 0, a2 = 0;
  for (int b = threadIdx.x; b < B; b += 256) {
    const double* s = stats + (int64_t)b * 4 * PIT_STAT;
    a1 += s[0 * PIT_STAT + 5] + s[3 * PIT_STAT + 5];   // perm 1: (p1,s1) + (p2,s2)
    a2 += s[1 * PIT_STAT + 5] + s[2 * PIT_STAT + 5];   // perm 2: (p1,s2) + (p2,s1)
  }
  a1 = block_sum(a1, red);
  a2 = block_sum(a2, red);
  const double l1 = a1 / (2.0 * B), l2 = a2 / (2.0 * B);
  const bool swap = l2 < l1;                            // ss_losses.py:23-25
  if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) {
    loss_out[0] = (float)(swap ? l2 : l1);
    loss_out[1] = swap ? 1.f : 0.f;
    loss_out[2] = (float)l1;
    loss_out[3] = (float)l2;
  }
  const int b = blockIdx.y, z = blockIdx.z;
  const int pair = swap ? (z == 0 ? 1 : 2) : (z == 0 ? 0 : 3);
  const float* x = (z == 0 ? p1 : p2) + (int64_t)b * T;
  const float* y = ((pair & 1) ? s2 : s1) + (int64_t)b * T;
  float* d = (z == 0 ? d1 : d2) + (int64_t)b * T;
  const double* s = stats + ((int64_t)b * 4 + pair) * PIT_STAT;
  const double mx = s[0], my = s[1], D = s[2], G = s[3], nn = s[4];
  const double c = (double)grad_scale / (2.0 * B) * (-20.0 / 2.302585092994046);
  const float cg = (float)(c * 2.0 / D), ce = (float)(c * 2.0 / nn), al = (float)(D / G), fmx = (float)mx, fmy = (float)my;
  const int64_t i0 = (int64_t)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int64_t i = i0 + j * 256;
    if (i < T) {
      const float g = y[i] - fmy, e = (x[i] - fmx) - al * g;
      d[i] = cg * g - ce * e;
    }
  }
}

// ---- global-norm clip over the flat gradient buffer (padding between slots is zero) --------------------------------
__global__ __launch_bounds__(256) void sumsq_partials_kernel(const float* __restrict__ g, int64_t n4,
                                                              double* __restrict__ partials) {
  __shared__ double red[4];
  double acc = 0;
  const f32x4* g4 = reinterpret_cast<const f32x4*>(g);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const f32x4 v = g4[i];
    acc += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}
// clip_grad_norm_: coef = max_norm / (norm + 1e-6) clamped to 1, gradients scaled in place (torch multiplies even when
// coef == 1).  max_norm <= 0: only the norm is reported.
__global__ __launch_bounds__(256) void clip_scale_kernel(float* __restrict__ g, int64_t n4, const double* __restrict__ partials,
                                                          int nparts, float max_norm, float* __restrict__ norm_out) {
  __shared__ double red[4];
  double acc = 0;
  for (int i = threadIdx.x; i < nparts; i += 256) acc += partials[i];
  const double norm = sqrt(block_sum(acc, red));
  if (blockIdx.x == 0 && threadIdx.x == 0) norm_out[0] = (float)norm;
  if (max_norm <= 0.f) return;
  const float coef = fminf((float)((double)max_norm / (norm + 1e-6)), 1.0f);
  f32x4* g4 = reinterpret_cast<f32x4*>(g);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) g4[i] = g4[i] * coef;
}

// ---- AdamW (torch.optim.AdamW single-tensor semantics, amsgrad off, maximize off) ---------------------------------------
// grid (tensors of this launch, ADAMW_YBLOCKS); the parameters live in their own nn.Parameter storages (pointer per slot),
// gradient / exp_avg / exp_avg_sq share the flat layout.
constexpr int ADAMW_MAX = 64;
constexpr int ADAMW_YBLOCKS = 16;
struct AdamwArgs {
  float* param[ADAMW_MAX];
  int64_t off[ADAMW_MAX];
  int n[ADAMW_MAX];
};
__global__ __launch_bounds__(256) void adamw_kernel(AdamwArgs a, const float* __restrict__ grad, float* __restrict__ m,
                                                     float* __restrict__ v, float beta1, float one_minus_beta1, float beta2,
                                                     float one_minus_beta2 /* both 1 - beta in double on the host, as torch */,
                                                     float eps, float decay /* 1 - lr * weight_decay */,
                                                     float step_size /* lr / bc1 */, float inv_sqrt_bc2) {
  const int e = blockIdx.x;
  float* p = a.param[e];
  const int64_t o = a.off[e];
  for (int i = blockIdx.y * 256 + threadIdx.x; i < a.n[e]; i += ADAMW_YBLOCKS * 256) {
    const float g = grad[o + i];
    const float m0 = m[o + i];
    const float mi = m0 + one_minus_beta1 * (g - m0);                // exp_avg.lerp_(grad, 1 - beta1)
    const float vi = beta2 * v[o + i] + one_minus_beta2 * (g * g);   // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    m[o + i] = mi;
    v[o + i] = vi;
    const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;              // (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
    p[i] = p[i] * decay - step_size * (mi / denom);                  // param.mul_(1 - lr wd); param.addcdiv_(exp_avg, denom, -step_size)
  }
}
