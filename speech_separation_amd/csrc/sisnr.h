// sisnr.h -- scale-invariant SNR statistics for the loss (A10) and the SI-SNRi metric (A11), one launch, one sync.
//
// Reference semantics
//   SiSNRLoss.forward            src/loss/ss_losses.py:100-114   -20 log10(|a t|^2 / |p - a t|^2), zero-mean, no eps
//   torchmetrics SI-SNR (A11)    src/metrics/si_snri.py:10,25-26 +10 log10((|a t|^2 + eps) / (|p - a t|^2 + eps)),
//                                zero-mean, a = (<p,t> + eps) / (<t,t> + eps), eps = FLT_EPSILON  (third-party,
//                                restated from its published definition; DESIGN.md section 0: parity unpinned)
// One workgroup per (item, pair); pairs = (p1,s1) (p1,s2) (p2,s1) (p2,s2) (mix,s1) (mix,s2).  Two passes over the
// 2 x T samples (L2 resident): the noise energy is summed directly, not as a difference of energies (which cancels
// catastrophically at high SI-SNR).  Partial sums are double precision.
#pragma once
#include "common.h"

DEV double block_sum(double v, double* red) {
  // wave reduce (64 lanes) then across the 4 waves
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void sisnr_pairs_kernel(const float* __restrict__ p1, const float* __restrict__ p2,
                                                           const float* __restrict__ s1, const float* __restrict__ s2,
                                                           const float* __restrict__ mix, int64_t T,
                                                           float* __restrict__ out /* [B][6][2] = (metric dB, loss) */) {
  __shared__ double red[4];
  const int b = blockIdx.x, pair = blockIdx.y;
  const float* x = (pair < 2 ? p1 : pair < 4 ? p2 : mix) + (int64_t)b * T;
  const float* y = ((pair & 1) ? s2 : s1) + (int64_t)b * T;
  double sx = 0, sy = 0, sxy = 0, syy = 0;
  for (int64_t i = threadIdx.x; i < T; i += 256) {
    const double xv = x[i], yv = y[i];
    sx += xv; sy += yv; sxy += xv * yv; syy += yv * yv;
  }
  sx = block_sum(sx, red); sy = block_sum(sy, red); sxy = block_sum(sxy, red); syy = block_sum(syy, red);
  const double mx = sx / (double)T, my = sy / (double)T;
  const double dot = sxy - (double)T * mx * my;        // <x - mx, y - my>
  const double eyy = syy - (double)T * my * my;        // |y - my|^2
  const double eps = 1.1920928955078125e-07;           // torch.finfo(float32).eps
  const double a_loss = dot / eyy, a_met = (dot + eps) / (eyy + eps);
  double n_loss = 0, n_met = 0;
  for (int64_t i = threadIdx.x; i < T; i += 256) {
    const double xc = (double)x[i] - mx, yc = (double)y[i] - my;
    const double e1 = xc - a_loss * yc, e2 = xc - a_met * yc;
    n_loss += e1 * e1; n_met += e2 * e2;
  }
  n_loss = block_sum(n_loss, red); n_met = block_sum(n_met, red);
  if (threadIdx.x == 0) {
    out[((int64_t)b * 6 + pair) * 2 + 0] = (float)(10.0 * log10((a_met * a_met * eyy + eps) / (n_met + eps)));
    out[((int64_t)b * 6 + pair) * 2 + 1] = (float)(-20.0 * log10((a_loss * a_loss * eyy) / n_loss));
  }
}
