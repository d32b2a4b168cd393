// lstm_bptt.hip -- LSTM backward through time for gfx950 (mirror image of lstm.hip), its own translation unit
// (compiled with -mllvm -amdgpu-mfma-vgpr-form, see build.py).
//
// What torch.autograd derives from nn.LSTM's cell (src/model/dptn.py:22-30) given the tape of lstm.hip (SAVE):
//   dh = upstream[t] + recurrent;  do = dh*tanh(c);  dc += dh*o*(1-tanh(c)^2);  di = dc*g;  dg = dc*i;  df = dc*c_prev;
//   dP = (di*i(1-i), df*f(1-f), dg*(1-g^2), do*o(1-o))   [lane-local: same fragment slots as the forward]
//   recurrent dh_prev = dP[32 x 512] * W_hh[512 x 128]   (256 MFMA per wave, W_hh^T resident in AGPRs)
// One workgroup = one direction x 32 sequences, all steps, walked against the forward order.
//
// Step structure (beside fp32 MFMAs VALU work is paid in full but memory instructions ride along for free when they
// are spread one per MFMA group, DESIGN.md 3.5 -- so everything that touches memory lives in the MFMA phase):
//   A. dP of the step from registers (tape tiles + upstream dh were loaded during the PREVIOUS step's MFMA phase),
//      packed fp32 math, dP -> LDS tile (double buffered); bias-gradient column sums ride along.
//   B. one barrier.
//   C. 256 MFMAs, A fragments (dP rows) software-pipelined from LDS one batch ahead; between the MFMA groups:
//      the 40 tape loads + 16 upstream-dh loads of the NEXT step (straight into the registers phase A just freed) and
//      the 16 row stores of this step's dP (token-major, 2 KiB contiguous per row, read back from the LDS tile).
#include <hip/hip_runtime.h>

#include "lstm_bptt.h"

namespace {

// d tanh-free pieces of the cell backward for TWO accumulator slots (packed fp32)
struct Bptt2 {
  f32x2 dpi, dpf, dpg, dpo, dc_next;
};
DEV Bptt2 bptt_cell2(f32x2 dh, f32x2 dc_in, f32x2 i, f32x2 f, f32x2 g, f32x2 o, f32x2 c, f32x2 cprev) {
  // tanh(c) = 2 / (1 + 2^(-2 log2e c)) - 1
  const f32x2 e = exp2_2(c * -2.8853900817779268f);
  const f32x2 tc = 2.0f * rcp_2(1.0f + e) - 1.0f;
  const f32x2 dho = dh * o;
  const f32x2 dc = dc_in + dho * (1.0f - tc * tc);
  Bptt2 r;
  r.dpo = dh * tc * (o - o * o);
  const f32x2 dci = dc * i;
  r.dpi = dc * g * (i - i * i);
  r.dpf = dc * cprev * (f - f * f);
  r.dpg = dci - dci * g * g;
  r.dc_next = dc * f;
  return r;
}

__global__ __launch_bounds__(256) void lstm_bptt_kernel(const float* __restrict__ tape_gates,
                                                         const float* __restrict__ tape_c,
                                                         const float* __restrict__ whh_f, const float* __restrict__ whh_b,
                                                         const float* __restrict__ dh_up, int ldh,
                                                         float* __restrict__ dg_out, int ldg, int dump_row, SeqGeom g,
                                                         float* __restrict__ bias_partials /* [ndir][nst][512] */) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* DPs = smem;   // [2][32][BPTT_LDP]
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, c = lane & 31, hh = lane >> 5;
  const int st = blockIdx.x, d = blockIdx.y;
  const float* whh = d ? whh_b : whh_f;

  // B operand of dh_prev = dP W_hh:  B[k = gate column][j = hidden unit 32w + c] = W_hh[k][32w + c]
  float wf[256];
#pragma unroll
  for (int m = 0; m < 64; ++m)
#pragma unroll
    for (int t = 0; t < 4; ++t) wf[4 * m + t] = whh[(int64_t)(8 * m + 4 * hh + t) * LSTM_H + 32 * w + c];
#pragma unroll
  for (int i = 0; i < 256; ++i) asm volatile("" : "+a"(wf[i]));

  const int tstride = seq_token_stride(g);
  // processing order: against the forward order of this direction; the forward-order predecessor of t is the NEXT
  // position processed
  const int t_first = d ? 0 : g.len - 1, tdir = d ? 1 : -1;
  unsigned hidx[16];
  unsigned vmask = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int q = st * 32 + ROW32(r, hh);
    const unsigned tokb = (q < g.nseq ? (unsigned)seq_token_base(g, q) : (unsigned)dump_row) + (unsigned)(t_first * tstride);
    hidx[r] = tokb * (unsigned)ldh + (unsigned)(d * LSTM_H + 32 * w + c);
    vmask |= (q < g.nseq ? 1u : 0u) << r;
  }
  // dP rows leave through the LDS tile: wave w stores tile rows 8w + j (wave-uniform bases), each row 512 floats of
  // this direction = two 16-byte pieces per lane
  float* grow[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int q = st * 32 + 8 * w + j;
    const int64_t tokb = (q < g.nseq ? seq_token_base(g, q) : (int64_t)dump_row) + (int64_t)t_first * tstride;
    grow[j] = dg_out + tokb * ldg + d * 512;
  }
  const int64_t gstep = (int64_t)tdir * tstride * ldg;

  // tape tiles of the step being processed: gates [gate][quarter] (elements 4q..4q+3), c, c of the forward predecessor
  float4 G[4][4], Cc[4], Cp[4];
  float dhu[16];
  auto tape_g = [&](int t) { return tape_gates + pre_tile_offset(d, st, t, g.nst, g.len) + (int64_t)w * 1024 + lane * 4; };
  auto tape_cc = [&](int t) { return tape_c + pre_tile_offset(d, st, t, g.nst, g.len) / 4 + (int64_t)w * 1024 + lane * 4; };
  // memory operation k of the prefetch for position t (its predecessor tp): 16 gate pieces, 4 + 4 cell pieces, 16 dh
  const float* dhp = dh_up;   // wave-uniform base of the position being fetched; hidx stays the row's invariant offset
  auto prefetch_op = [&](int k, const float* tg, const float* tcc, const float* tcp) {
    if (k < 16) G[k >> 2][k & 3] = *reinterpret_cast<const float4*>(tg + (k >> 2) * 4096 + (k & 3) * 256);
    else if (k < 20) Cc[k - 16] = *reinterpret_cast<const float4*>(tcc + (k - 16) * 256);
    else if (k < 24) Cp[k - 20] = *reinterpret_cast<const float4*>(tcp + (k - 20) * 256);
    else if (k < 40) dhu[k - 24] = dhp[hidx[k - 24]];
  };
  {
    const int tp = t_first - (d ? -1 : 1);
    const bool hp = tp >= 0 && tp < g.len;
    const float* tg = tape_g(t_first);
    const float* tcc = tape_cc(t_first);
    const float* tcp = tape_cc(hp ? tp : t_first);
#pragma unroll
    for (int k = 0; k < 40; ++k) prefetch_op(k, tg, tcc, tcp);
  }

  // bias gradients (b_ih and b_hh share them) = column sums of dP over rows and steps, one partial row per workgroup
  f32x2 bsum[4] = {(f32x2){0.f, 0.f}, (f32x2){0.f, 0.f}, (f32x2){0.f, 0.f}, (f32x2){0.f, 0.f}};
  const bool full_tile = (st + 1) * 32 <= g.nseq;
  f32x16 dh_rec = zero16(), dc_rec = zero16();
  for (int step = 0; step < g.len; ++step) {
    const int t = t_first + tdir * step;
    const int t_prev = t + tdir;                      // forward-order predecessor = next position processed
    const bool has_prev = t_prev >= 0 && t_prev < g.len;
    float* dp = DPs + (step & 1) * 32 * BPTT_LDP;

    // ---- A. dP of this step (registers only) ---------------------------------------------------------------
    if (!has_prev) {
#pragma unroll
      for (int q = 0; q < 4; ++q) Cp[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (!full_tile) {   // rows of padded sequences carry zeros (their upstream dh comes from the dump rows): then their
                        // dP, recurrent dh and bias contributions are zero as well
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (!((vmask >> r) & 1u)) dhu[r] = 0.f;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float iv[4] = {G[0][q].x, G[0][q].y, G[0][q].z, G[0][q].w}, fv[4] = {G[1][q].x, G[1][q].y, G[1][q].z, G[1][q].w};
      const float gv[4] = {G[2][q].x, G[2][q].y, G[2][q].z, G[2][q].w}, ov[4] = {G[3][q].x, G[3][q].y, G[3][q].z, G[3][q].w};
      const float cv[4] = {Cc[q].x, Cc[q].y, Cc[q].z, Cc[q].w}, pv[4] = {Cp[q].x, Cp[q].y, Cp[q].z, Cp[q].w};
#pragma unroll
      for (int e = 0; e < 4; e += 2) {
        const int r = 4 * q + e;
        const f32x2 dh = (f32x2){dhu[r], dhu[r + 1]} + (f32x2){dh_rec[r], dh_rec[r + 1]};
        const Bptt2 u = bptt_cell2(dh, (f32x2){dc_rec[r], dc_rec[r + 1]}, (f32x2){iv[e], iv[e + 1]}, (f32x2){fv[e], fv[e + 1]},
                                   (f32x2){gv[e], gv[e + 1]}, (f32x2){ov[e], ov[e + 1]}, (f32x2){cv[e], cv[e + 1]},
                                   (f32x2){pv[e], pv[e + 1]});
        dc_rec[r] = u.dc_next.x;
        dc_rec[r + 1] = u.dc_next.y;
        bsum[0] += u.dpi;
        bsum[1] += u.dpf;
        bsum[2] += u.dpg;
        bsum[3] += u.dpo;
        float* lp0 = dp + ROW32(r, hh) * BPTT_LDP + 32 * w + c;
        float* lp1 = dp + ROW32(r + 1, hh) * BPTT_LDP + 32 * w + c;
        lp0[0] = u.dpi.x; lp0[128] = u.dpf.x; lp0[256] = u.dpg.x; lp0[384] = u.dpo.x;
        lp1[0] = u.dpi.y; lp1[128] = u.dpf.y; lp1[256] = u.dpg.y; lp1[384] = u.dpo.y;
      }
    }
    if (step + 1 < g.len) dhp += (int64_t)tdir * tstride * ldh;   // (no position beyond the last one)
    // ---- B. ------------------------------------------------------------------------------------------------
    __syncthreads();

    // ---- C. dh_rec = dP W_hh (rows = sequences, K = 512 gate columns, this wave's 32 hidden units) ----------
    // next position's tape (clamped on the last step: a harmless reload)
    const int tn = step + 1 < g.len ? t_prev : t;
    const int tnp = tn + tdir;
    const float* ntg = tape_g(tn);
    const float* ntcc = tape_cc(tn);
    const float* ntcp = tape_cc(tnp >= 0 && tnp < g.len ? tnp : tn);
    dh_rec = zero16();
    const float* arow = dp + c * BPTT_LDP + 4 * hh;
    const float* srow = dp + (8 * w) * BPTT_LDP + lane * 4;
    // 32 batches of 2 k-chunks (8 MFMAs), fetched one batch ahead (a deeper pipeline does not fit the registers);
    // every other batch also carries one of the wave's 8 x 2 dP row pieces
    float4 afr[2][2], rowv;
    auto fetch_batch = [&](int b, int buf) {
#pragma unroll
      for (int m = 0; m < 2; ++m) afr[buf][m] = *reinterpret_cast<const float4*>(arow + 8 * (2 * b + m));
    };
    fetch_batch(0, 0);
#pragma unroll
    for (int b = 0; b < 32; ++b) {
      const int buf = b & 1;
      if (b + 1 < 32) fetch_batch(b + 1, buf ^ 1);
      const int piece = b >> 1;   // row piece stored in the second batch of each pair, fetched in the first
      if (!(b & 1)) rowv = *reinterpret_cast<const float4*>(srow + (piece >> 1) * BPTT_LDP + (piece & 1) * 256);
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const float av[4] = {afr[buf][m].x, afr[buf][m].y, afr[buf][m].z, afr[buf][m].w};
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) dh_rec = mfma32(av[tt], wf[4 * (2 * b + m) + tt], dh_rec);
        // one memory instruction per group of 4 MFMAs: 16 row stores, 40 loads (64 slots)
        const int slot = 2 * b + m;
        if ((slot & 3) == 3) {
          *reinterpret_cast<float4*>(grow[piece >> 1] + (piece & 1) * 256 + lane * 4) = rowv;
        } else {
          const int k = 3 * (slot >> 2) + (slot & 3);   // 0..47
          if (k < 40) prefetch_op(k, ntg, ntcc, ntcp);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) grow[j] += gstep;
  }
#pragma unroll
  for (int gi = 0; gi < 4; ++gi) {
    const float own = bsum[gi].x + bsum[gi].y;
    const float v = own + __shfl_xor(own, 32);
    if (hh == 0) bias_partials[((size_t)d * g.nst + st) * 512 + gi * LSTM_H + 32 * w + c] = v;
  }
}

}  // namespace

int lstm_bptt_launch(int nst, int ndir, void* stream, const float* tape_gates, const float* tape_c, const float* whh_f,
                     const float* whh_b, const float* dh_up, int ldh, float* dg_out, int ldg, int dump_row,
                     const SeqGeom& g, float* bias_partials) {
  static PerDeviceOnce ready;
  const int dev = current_hip_device();
  if (!ready.done(dev)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_bptt_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)BPTT_LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    ready.set(dev);
  }
  hipLaunchKernelGGL(lstm_bptt_kernel, dim3(nst, ndir == 1 ? 1 : 2), dim3(256), BPTT_LDS_BYTES, static_cast<hipStream_t>(stream), tape_gates,
                     tape_c, whh_f, whh_b, dh_up, ldh, dg_out, ldg, dump_row, g, bias_partials);
  return (int)hipGetLastError();
}
