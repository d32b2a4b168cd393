// lstm_bptt16.hip -- LSTM backward through time on 16-sequence tiles (v_mfma_f32_16x16x4_f32): the training
// counterpart of lstm16.hip, for launches whose sequence count leaves CUs idle with 32-sequence tiles (the two halves
// of a split training batch).  Same mathematics and step structure as lstm_bptt.hip; own translation unit, compiled
// with -mllvm -amdgpu-mfma-vgpr-form (build.py).
//
//   dh = upstream[t] + recurrent;  do = dh*tanh(c);  dc += dh*o*(1-tanh(c)^2);  di = dc*g;  dg = dc*i;  df = dc*c_prev;
//   dP = (di*i(1-i), df*f(1-f), dg*(1-g^2), do*o(1-o))   [lane-local: the fragment slots of lstm16.hip]
//   recurrent dh_prev = dP[16 x 512] * W_hh[512 x 128]   (256 MFMA per wave, W_hh resident in AGPRs)
//
// Fragment map (lane l: i16 = l & 15, ks = l >> 4; wave w owns hidden units 32w + 16*half + i16):
//   cell slot (half, r) = sequence 4ks + r of the tile, unit 32w + 16*half + i16  (accumulator register r of block half)
//   A operand = dP rows from LDS: lane (i16 = sequence, ks) reads 16 bytes at gate column 16m + 4ks (MFMA step 4m + t
//   uses true k = 16m + 4ks + t);  B operand = W_hh[k][unit] resident;  D register r = (sequence 4ks + r, unit i16).
// Tape (written by lstm16.hip, SAVE):  gates [d][st16][t][w][b = 2*gate + half][lane][4],  c [d][st16][t][w][half][lane][4].
#include <hip/hip_runtime.h>

#include "lstm_bptt.h"

namespace {

typedef float f32x4v __attribute__((ext_vector_type(4)));
DEV f32x4v mfma16(float a, float b, f32x4v c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// the cell backward for TWO slots (packed fp32) -- identical to lstm_bptt.hip
struct Bptt2 {
  f32x2 dpi, dpf, dpg, dpo, dc_next;
};
DEV Bptt2 bptt_cell2(f32x2 dh, f32x2 dc_in, f32x2 i, f32x2 f, f32x2 g, f32x2 o, f32x2 c, f32x2 cprev) {
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

__global__ __launch_bounds__(256) void lstm_bptt16_kernel(const float* __restrict__ tape_gates,
                                                           const float* __restrict__ tape_c,
                                                           const float* __restrict__ whh_f,
                                                           const float* __restrict__ whh_b,
                                                           const float* __restrict__ dh_up, int ldh,
                                                           float* __restrict__ dg_out, int ldg, int dump_row, SeqGeom g,
                                                           int nst16, float* __restrict__ bias_partials /* [ndir][nst16][512] */) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* DPs = smem;   // [2][16][BPTT16_LDP]
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, i16 = lane & 15, ks = lane >> 4;
  const int st = blockIdx.x, d = blockIdx.y;
  const float* whh = d ? whh_b : whh_f;

  // B operand of dh_prev = dP W_hh:  B[k = gate column][j = hidden unit] = W_hh[k][32w + 16*half + i16]
  // (in eight pieces: a piece's 32 loads land in VGPRs and are handed to the AGPRs before the next piece is requested --
  //  with all 256 in flight at once the prologue needed 269 VGPRs and spilled 13 of them, 56 B/lane of scratch)
  float wf[2][128];
#pragma unroll
  for (int hf = 0; hf < 2; ++hf)
#pragma unroll
    for (int q = 0; q < 4; ++q) {        // a quarter of a half: 32 loads in flight, then their hand-over
#pragma unroll
      for (int m = 8 * q; m < 8 * q + 8; ++m)
#pragma unroll
        for (int t = 0; t < 4; ++t) wf[hf][4 * m + t] = whh[(int64_t)(16 * m + 4 * ks + t) * LSTM_H + 32 * w + 16 * hf + i16];
#pragma unroll
      for (int i = 32 * q; i < 32 * q + 32; ++i) asm volatile("" : "+a"(wf[hf][i]));
      __builtin_amdgcn_sched_barrier(0);
    }

  const int tstride = seq_token_stride(g);
  // processing order: against the forward order of this direction; the forward-order predecessor of t is the NEXT
  // position processed
  const int t_first = d ? 0 : g.len - 1, tdir = d ? 1 : -1;
  unsigned hidx[4];
  unsigned vmask = 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int q = st * 16 + 4 * ks + r;
    const unsigned tokb = (q < g.nseq ? (unsigned)seq_token_base(g, q) : (unsigned)dump_row) + (unsigned)(t_first * tstride);
    hidx[r] = tokb * (unsigned)ldh + (unsigned)(d * LSTM_H + 32 * w + i16);   // half 1: + 16
    vmask |= (q < g.nseq ? 1u : 0u) << r;
  }
  // dP rows leave through the LDS tile: wave w stores tile rows 4w + j (wave-uniform bases), each row 512 floats of this
  // direction = two 16-byte pieces per lane
  float* grow[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int q = st * 16 + 4 * w + j;
    const int64_t tokb = (q < g.nseq ? seq_token_base(g, q) : (int64_t)dump_row) + (int64_t)t_first * tstride;
    grow[j] = dg_out + tokb * ldg + d * 512;
  }
  const int64_t gstep = (int64_t)tdir * tstride * ldg;

  // tape tiles of the step being processed: gates [gate][half], c and c of the forward predecessor [half]
  float4 G[4][2], Cc[2], Cp[2];
  float dhu[2][4];
  auto tape_g = [&](int t) { return tape_gates + pre16_tile_offset(d, st, t, nst16, g.len) + (int64_t)w * 2048 + lane * 4; };
  auto tape_cc = [&](int t) { return tape_c + pre16_tile_offset(d, st, t, nst16, g.len) / 4 + (int64_t)w * 512 + lane * 4; };
  const float* dhp = dh_up;   // wave-uniform base of the position being fetched; hidx stays the row's invariant offset
  // memory operation k of the prefetch: 8 gate pieces, 2 + 2 cell pieces, 8 upstream-dh values
  auto prefetch_op = [&](int k, const float* tg, const float* tcc, const float* tcp) {
    if (k < 8) G[k >> 1][k & 1] = *reinterpret_cast<const float4*>(tg + k * 256);       // b = 2*gate + half = k
    else if (k < 10) Cc[k - 8] = *reinterpret_cast<const float4*>(tcc + (k - 8) * 256);
    else if (k < 12) Cp[k - 10] = *reinterpret_cast<const float4*>(tcp + (k - 10) * 256);
    else if (k < 20) dhu[(k - 12) >> 2][(k - 12) & 3] = dhp[hidx[(k - 12) & 3] + 16 * ((k - 12) >> 2)];
  };
  {
    const int tp = t_first + tdir;
    const bool hp = tp >= 0 && tp < g.len;
    const float* tg = tape_g(t_first);
    const float* tcc = tape_cc(t_first);
    const float* tcp = tape_cc(hp ? tp : t_first);
#pragma unroll
    for (int k = 0; k < 20; ++k) prefetch_op(k, tg, tcc, tcp);
  }

  // bias gradients (b_ih and b_hh share them) = column sums of dP over rows and steps, one partial row per workgroup
  f32x2 bsum[4][2];
#pragma unroll
  for (int gi = 0; gi < 4; ++gi)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) bsum[gi][hf] = (f32x2){0.f, 0.f};
  const bool full_tile = (st + 1) * 16 <= g.nseq;
  f32x4v dh_rec[2] = {(f32x4v){0.f, 0.f, 0.f, 0.f}, (f32x4v){0.f, 0.f, 0.f, 0.f}};
  f32x4v dc_rec[2] = {(f32x4v){0.f, 0.f, 0.f, 0.f}, (f32x4v){0.f, 0.f, 0.f, 0.f}};
  for (int step = 0; step < g.len; ++step) {
    const int t = t_first + tdir * step;
    const int t_prev = t + tdir;                      // forward-order predecessor = next position processed
    const bool has_prev = t_prev >= 0 && t_prev < g.len;
    float* dp = DPs + (step & 1) * 16 * BPTT16_LDP;

    // ---- A. dP of this step (registers only) ---------------------------------------------------------------
    if (!has_prev) {
      Cp[0] = make_float4(0.f, 0.f, 0.f, 0.f);
      Cp[1] = Cp[0];
    }
    if (!full_tile) {   // rows of padded sequences carry zeros (their upstream dh comes from the dump rows)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (!((vmask >> r) & 1u)) dhu[hf][r] = 0.f;
    }
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const float iv[4] = {G[0][hf].x, G[0][hf].y, G[0][hf].z, G[0][hf].w}, fv[4] = {G[1][hf].x, G[1][hf].y, G[1][hf].z, G[1][hf].w};
      const float gv[4] = {G[2][hf].x, G[2][hf].y, G[2][hf].z, G[2][hf].w}, ov[4] = {G[3][hf].x, G[3][hf].y, G[3][hf].z, G[3][hf].w};
      const float cv[4] = {Cc[hf].x, Cc[hf].y, Cc[hf].z, Cc[hf].w}, pv[4] = {Cp[hf].x, Cp[hf].y, Cp[hf].z, Cp[hf].w};
#pragma unroll
      for (int r = 0; r < 4; r += 2) {
        const f32x2 dh = (f32x2){dhu[hf][r], dhu[hf][r + 1]} + (f32x2){dh_rec[hf][r], dh_rec[hf][r + 1]};
        const Bptt2 u = bptt_cell2(dh, (f32x2){dc_rec[hf][r], dc_rec[hf][r + 1]}, (f32x2){iv[r], iv[r + 1]},
                                   (f32x2){fv[r], fv[r + 1]}, (f32x2){gv[r], gv[r + 1]}, (f32x2){ov[r], ov[r + 1]},
                                   (f32x2){cv[r], cv[r + 1]}, (f32x2){pv[r], pv[r + 1]});
        dc_rec[hf][r] = u.dc_next.x;
        dc_rec[hf][r + 1] = u.dc_next.y;
        bsum[0][hf] += u.dpi;
        bsum[1][hf] += u.dpf;
        bsum[2][hf] += u.dpg;
        bsum[3][hf] += u.dpo;
        float* lp0 = dp + (4 * ks + r) * BPTT16_LDP + 32 * w + 16 * hf + i16;
        float* lp1 = lp0 + BPTT16_LDP;
        lp0[0] = u.dpi.x; lp0[128] = u.dpf.x; lp0[256] = u.dpg.x; lp0[384] = u.dpo.x;
        lp1[0] = u.dpi.y; lp1[128] = u.dpf.y; lp1[256] = u.dpg.y; lp1[384] = u.dpo.y;
      }
    }
    if (step + 1 < g.len) dhp += (int64_t)tdir * tstride * ldh;   // (no position beyond the last one)
    // ---- B. ------------------------------------------------------------------------------------------------
    __syncthreads();

    // ---- C. dh_rec = dP W_hh (rows = sequences, K = 512 gate columns, this wave's 2 x 16 hidden units) ----------
    // next position's tape (clamped on the last step: a harmless reload)
    const int tn = step + 1 < g.len ? t_prev : t;
    const int tnp = tn + tdir;
    const float* ntg = tape_g(tn);
    const float* ntcc = tape_cc(tn);
    const float* ntcp = tape_cc(tnp >= 0 && tnp < g.len ? tnp : tn);
    dh_rec[0] = (f32x4v){0.f, 0.f, 0.f, 0.f};
    dh_rec[1] = dh_rec[0];
    const float* arow = dp + i16 * BPTT16_LDP + 4 * ks;
    const float* srow = dp + (4 * w) * BPTT16_LDP + lane * 4;
    // 16 batches of 2 k-chunks (16 MFMAs = 4 groups of 4), fetched one batch ahead; every other batch also carries one
    // of the wave's 4 x 2 dP row pieces; one memory instruction per group: 8 row stores, 20 loads (64 slots)
    float4 afr[2][2], rowv;
    auto fetch_batch = [&](int b, int buf) {
#pragma unroll
      for (int m = 0; m < 2; ++m) afr[buf][m] = *reinterpret_cast<const float4*>(arow + 16 * (2 * b + m));
    };
    fetch_batch(0, 0);
#pragma unroll
    for (int b = 0; b < 16; ++b) {
      const int buf = b & 1;
      if (b + 1 < 16) fetch_batch(b + 1, buf ^ 1);
      const int piece = b >> 1;   // row piece stored in the second batch of each pair, fetched in the first
      if (!(b & 1)) rowv = *reinterpret_cast<const float4*>(srow + (piece >> 1) * BPTT16_LDP + (piece & 1) * 256);
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const float av[4] = {afr[buf][m].x, afr[buf][m].y, afr[buf][m].z, afr[buf][m].w};
#pragma unroll
        for (int half = 0; half < 2; ++half) {   // two groups of 4 MFMAs per k-chunk
#pragma unroll
          for (int tt = 2 * half; tt < 2 * half + 2; ++tt)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) dh_rec[hf] = mfma16(av[tt], wf[hf][4 * (2 * b + m) + tt], dh_rec[hf]);
          const int slot = 4 * b + 2 * m + half;   // 0..63
          if ((b & 1) && m == 1 && half == 1) {
            *reinterpret_cast<float4*>(grow[piece >> 1] + (piece & 1) * 256 + lane * 4) = rowv;
          } else {
            const int k = slot - (slot >> 3);      // skips the store slots (slot % 8 == 7): 0..55
            if (k < 20) prefetch_op(k, ntg, ntcc, ntcp);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) grow[j] += gstep;
  }
  // column sums: the four lane groups ks hold the same columns (different sequences)
#pragma unroll
  for (int gi = 0; gi < 4; ++gi)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      float v = bsum[gi][hf].x + bsum[gi][hf].y;
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      if (ks == 0) bias_partials[((size_t)d * nst16 + st) * 512 + gi * LSTM_H + 32 * w + 16 * hf + i16] = v;
    }
}

}  // namespace

int lstm_bptt16_launch(int nst16, int ndir, void* stream, const float* tape_gates, const float* tape_c, const float* whh_f,
                       const float* whh_b, const float* dh_up, int ldh, float* dg_out, int ldg, int dump_row,
                       const SeqGeom& g, float* bias_partials) {
  static PerDeviceOnce ready;
  const int dev = current_hip_device();
  if (!ready.done(dev)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_bptt16_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)BPTT16_LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    ready.set(dev);
  }
  hipLaunchKernelGGL(lstm_bptt16_kernel, dim3(nst16, ndir == 1 ? 1 : 2), dim3(256), BPTT16_LDS_BYTES, static_cast<hipStream_t>(stream),
                     tape_gates, tape_c, whh_f, whh_b, dh_up, ldh, dg_out, ldg, dump_row, g, nst16, bias_partials);
  return (int)hipGetLastError();
}
