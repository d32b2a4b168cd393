// attention.h -- chunk-local multi-head self-attention for gfx950, exact fp32.
//
// Reference semantics: nn.MultiheadAttention(N, heads, batch_first=True) in eval mode, self-attention,
// no mask (src/model/dptn.py:16-21,46): softmax(Q K^T / sqrt(dh)) V per head.  The in/out projections
// are GEMM-engine launches; this kernel is the part between them.
//
// Sequences are short (K = 150 intra, S = 141 inter), so one workgroup = one (sequence, head) and the
// whole score matrix lives in registers -- no online softmax.
//   * NKB = ceil(len/32) waves; wave qb owns queries [32qb, 32qb+32).
//   * K, V of the (sequence, head) are staged once in LDS (rows >= len are zero).
//   * scores are computed TRANSPOSED: S^T[key][query] = K Q^T, so lane (c,hh) holds 16*NKB scores that
//     all belong to query c -> the row max / row sum are register-local plus ONE cross-half shuffle.
//   * P V is accumulated transposed, O^T = V^T P^T: the accumulator register (rb, r) of S^T is used AS-IS as the
//     MFMA B operand (k-slot hh means key rb*32 + ROW32(r,hh)); the A operand is the matching V row, read from
//     LDS.  No transposes, no LDS round trip for P, and every output value of a query stays in the query's lane.
//   * token addressing is strided (SeqGeom), so intra and inter views read the same QKV buffer.
#pragma once
#include "common.h"

// phase stamps for tools/microbench/attn_phases.hip (a diagnostic build defines ATTN_STAMP; the product build does not)
#ifndef ATTN_STAMP
#define ATTN_STAMP(i)
#endif

template <int DH>
struct AttnShape {
  static constexpr int LDK = DH + 4;  // K rows: conflict-free ds_read_b128
  static constexpr int LDV = 32;      // V rows: 32 columns (zero padded when DH = 16)
  static constexpr size_t lds_bytes(int nkb) { return sizeof(float) * (size_t)nkb * 32 * (LDK + LDV); }
};

template <int DH, int NKB>
__global__ __launch_bounds__(64 * NKB, (NKB >= 4 && NKB <= 6) ? 3 : 2) void attention_kernel(const float* __restrict__ qkv,
                                                              float* __restrict__ out, int N, int heads, SeqGeom g,
                                                              float scale_log2e, DropCfg drop,
                                                              float2* __restrict__ stats_out = nullptr,
                                                              unsigned long long* __restrict__ mask_out = nullptr) {
  using Sh = AttnShape<DH>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ks = smem;                        // [NKB*32][LDK]
  float* Vs = smem + NKB * 32 * Sh::LDK;   // [NKB*32][32]

  ATTN_STAMP(0);
  const int tid = threadIdx.x;
  const int qb = tid >> 6, lane = tid & 63, c = lane & 31, hh = lane >> 5;
  // sequence-fastest grid order (head-adjacent workgroups measured 1-4 % slower)
  const int seq = blockIdx.x, head = blockIdx.y;
  const int len = g.len;
  const int64_t tok0 = seq_token_base(g, seq);
  const int tstride = seq_token_stride(g);
  const int ld = 3 * N;
  const float* base = qkv + head * DH;

  // ---- issue EVERY global load first (K, V staging rows and the Q fragment), then store to LDS ----
  constexpr int R4 = DH / 4;                        // float4 per row
  constexpr int NST = (32 * R4) / 64;               // staging float4 per thread for K (and for V)
  static_assert((32 * R4) % 64 == 0, "staging map");
  float4 kreg[NST], vreg[NST];
#pragma unroll
  for (int i = 0; i < NST; ++i) {
    const int idx = i * (64 * NKB) + tid;
    const int p = idx / R4, f = idx % R4;
    kreg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    vreg[i] = kreg[i];
    if (p < len) {
      const float* row = base + (tok0 + (int64_t)p * tstride) * ld + 4 * f;
      kreg[i] = *reinterpret_cast<const float4*>(row + N);
      vreg[i] = *reinterpret_cast<const float4*>(row + 2 * N);
    }
  }
  // Q fragments (B operand), pre-scaled by log2(e)/sqrt(dh)
  float qf[DH / 2];
  {
    const int p = qb * 32 + c;
    const float* row = base + (tok0 + (int64_t)(p < len ? p : 0) * tstride) * ld + 4 * hh;
#pragma unroll
    for (int m = 0; m < DH / 8; ++m) {
      float4 v = *reinterpret_cast<const float4*>(row + 8 * m);
      if (p >= len) v = make_float4(0.f, 0.f, 0.f, 0.f);
      qf[4 * m + 0] = v.x * scale_log2e;
      qf[4 * m + 1] = v.y * scale_log2e;
      qf[4 * m + 2] = v.z * scale_log2e;
      qf[4 * m + 3] = v.w * scale_log2e;
    }
  }
#pragma unroll
  for (int i = 0; i < NST; ++i) {
    const int idx = i * (64 * NKB) + tid;
    const int p = idx / R4, f = idx % R4;
    *reinterpret_cast<float4*>(&Ks[p * Sh::LDK + 4 * f]) = kreg[i];
    *reinterpret_cast<float4*>(&Vs[p * Sh::LDV + 4 * f]) = vreg[i];
    if (DH == 16) *reinterpret_cast<float4*>(&Vs[p * Sh::LDV + 16 + 4 * f]) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();
  ATTN_STAMP(1);

  // ---- S^T = K Q^T ------------------------------------------------------------------------------
  f32x16 s[NKB];
#pragma unroll
  for (int rb = 0; rb < NKB; ++rb) {
    s[rb] = zero16();
    const float* krow = &Ks[(rb * 32 + c) * Sh::LDK + 4 * hh];
#pragma unroll
    for (int m = 0; m < DH / 8; ++m) {
      const float4 k = *reinterpret_cast<const float4*>(krow + 8 * m);
      s[rb] = mfma32(k.x, qf[4 * m + 0], s[rb]);
      s[rb] = mfma32(k.y, qf[4 * m + 1], s[rb]);
      s[rb] = mfma32(k.z, qf[4 * m + 2], s[rb]);
      s[rb] = mfma32(k.w, qf[4 * m + 3], s[rb]);
    }
  }

  ATTN_STAMP(2);
  // ---- softmax over keys (per query = per lane column); only the last key block can hold padding ----
#pragma unroll
  for (int r = 0; r < 16; ++r)
    if ((NKB - 1) * 32 + ROW32(r, hh) >= len) s[NKB - 1][r] = -1e30f;
  float mx = -1e30f;
#pragma unroll
  for (int rb = 0; rb < NKB; ++rb)
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[rb][r]);
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  float sum = 0.f;
#pragma unroll
  for (int rb = 0; rb < NKB; ++rb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e = fast_exp2(s[rb][r] - mx);
      s[rb][r] = e;
      sum += e;
    }
  sum += __shfl_xor(sum, 32);
  const float inv = fast_rcp(sum);
  // training forward: the softmax statistics of every query (row max in the log2 domain, 1 / row sum) go to the tape;
  // with them the backward handles one key block at a time instead of keeping the whole score row in registers
  if (stats_out != nullptr && hh == 0 && qb * 32 + c < len)
    stats_out[(tok0 + (int64_t)(qb * 32 + c) * tstride) * heads + head] = make_float2(mx, inv);
  if (drop.thresh != 0u || mask_out != nullptr) {   // train-mode dropout on the (normalised) probabilities; the normaliser keeps all keys
    const uint32_t qseed = drop_qseed(drop.seed, (uint32_t)(tok0 + (int64_t)(qb * 32 + c) * tstride) * (uint32_t)heads + (uint32_t)head);
    // The keep decisions go to the tape as BIT MASKS (training forward): the compare of register (rb, r) is a 64-bit lane mask
    // already -- bit L = keep(query c(L), key 32 rb + ROW32(r, hh(L))) -- so word (rb, r) of this wave costs two v_writelane to
    // park (lane 16 rb + r of a register pair) and the wave's 16 NKB words leave as one or two coalesced 8-byte stores per lane.
    // Both backward kernels then READ the decision (one select on the mask, or a bit extract in the transposed phase) instead
    // of re-hashing it: the hash was ~850 of their ~1 700 vector instructions beside 160-240 MFMAs.
    //   mask[seq][head][qb][rb][r] (uint64)
    // (fence: the hash does not depend on the softmax, and hoisted above it every compare's mask has to survive until its
    //  probability exists -- the compiler parked 80 of them in a 129th VGPR by v_writelane / v_readlane pairs)
    __builtin_amdgcn_sched_barrier(0);
    int wlo = 0, whi = 0;
    unsigned long long* mw = mask_out == nullptr ? nullptr : mask_out + ((((int64_t)seq * heads + head) * NKB + qb) * NKB) * 16;
#pragma unroll
    for (int rb = 0; rb < NKB; ++rb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const bool keep = drop_rand_q(qseed, (uint32_t)(rb * 32 + ROW32(r, hh))) >= drop.thresh;
        s[rb][r] = keep ? s[rb][r] * drop.inv_keep : 0.f;
        asm volatile("" : "+v"(s[rb][r]));      // the select HERE: sunk to the P V loop it kept every mask alive (78 SGPR spills)
        // (parked unconditionally -- a branch per element otherwise; only the stores below depend on mask_out.  The s_nop is
        //  REQUIRED: the mask comes out of a v_cmp, and a v_writelane that reads an SGPR in the instruction slot behind the VALU
        //  instruction that wrote it gets the OLD value -- the compiler's hazard pass does not look inside an asm statement.  Found
        //  on the hardware: every low half written directly behind its compare was 0, every high half, one slot later, correct.)
        const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
        asm("s_nop 3\n\tv_writelane_b32 %0, %2, %4\n\tv_writelane_b32 %1, %3, %4"
            : "+v"(wlo), "+v"(whi)
            : "s"((unsigned)m), "s"((unsigned)(m >> 32)), "n"((rb * 16 + r) & 63));
        // four elements at a time: left alone the scheduler forms all 16 NKB compares first (they do not depend on the softmax)
        // and keeps their masks alive in SGPRs -- 82 of them spilled, and the kernel went from 124 to 129 VGPRs = from three
        // workgroups per CU to two (training step 152.3 -> 150.5 mixtures/s, same-box A/B of two builds)
        if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
      // 64 words are a register pair's worth: key blocks 0..3 leave here, the rest (NKB > 4) behind the last block
      if ((rb & 3) == 3 || rb == NKB - 1) {
        const int nw = 16 * ((rb & 3) + 1);
        if (mw != nullptr && lane < nw) mw[64 * (rb >> 2) + lane] = (unsigned long long)(unsigned)wlo | ((unsigned long long)(unsigned)whi << 32);
      }
    }
  }

  ATTN_STAMP(3);
  // ---- O^T = V^T P^T: V rows are fetched one key block (16 ds_read_b32) ahead of the MFMAs that use them.  The
  //      output is accumulated TRANSPOSED (A = V^T: lane = d, k-slot hh <-> key ROW32(r,hh); B = the S^T registers:
  //      lane = query), so a lane ends up with 16 output values of ITS query: the normaliser is lane-local and the
  //      row leaves as 16-byte pieces (no cross-lane traffic in the epilogue) ----
  f32x16 o = zero16();
  const float* vcol = Vs + (4 * hh) * Sh::LDV + c;
#pragma unroll
  for (int rb = 0; rb < NKB; ++rb) {
    float vv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) vv[r] = vcol[(rb * 32 + (r & 3) + 8 * (r >> 2)) * Sh::LDV];
    // one wait for the whole batch, and MFMAs kept behind it: left alone the compiler pairs every two MFMAs with their
    // own LDS read and waits for it there (40 exposed round trips per wave)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 16; ++r) o = mfma32(vv[r], s[rb][r], o);
  }

  ATTN_STAMP(4);
  // ---- normalise and store: reg 4j+i of lane (c,hh) is O[query c][d = 8j + 4hh + i] ---------------------------
  {
    const int p = qb * 32 + c;
    if (p < len) {
      float* orow = out + (tok0 + (int64_t)p * tstride) * N + head * DH;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int d0 = 8 * j + 4 * hh;
        if (d0 < DH)
          *reinterpret_cast<float4*>(orow + d0) =
              make_float4(o[4 * j + 0] * inv, o[4 * j + 1] * inv, o[4 * j + 2] * inv, o[4 * j + 3] * inv);
      }
    }
  }
  ATTN_STAMP(5);
}

// ------------------------------------------------------------------------------------------------
// Long sequences (len > 256: more than ~7 s of audio on the inter-chunk path): streaming softmax.
// ------------------------------------------------------------------------------------------------
// Same transposed score tiles, but the keys are walked in blocks of 32 with a running (max, sum) per query, and the
// output is accumulated TRANSPOSED as well:  O^T[d][query] = V^T P^T, with the S^T accumulator registers used as-is
// as the MFMA *B* operand (k-slot hh <-> key ROW32(r,hh)) and V^T rows as the A operand.  Then every quantity of a
// query -- its running max, its sum, its 16 output values per lane -- lives in the lanes c = query, so the rescale
// o *= 2^(m_old - m_new) is lane-local and the only cross-lane traffic per key block is the max / sum exchange
// between the two lane halves.  One workgroup = one (sequence, head); wave w takes query blocks w, w+4, ...;
// K / V rows are read straight from global memory (the four waves and the heads' workgroups share them through L2),
// one key block ahead of the MFMAs.  No LDS, no length limit.  Inference only (no dropout, no backward).
template <int DH>
__global__ __launch_bounds__(256, 2) void attention_long_kernel(const float* __restrict__ qkv, float* __restrict__ out, int N,
                                                                 SeqGeom g, float scale_log2e) {
  static_assert(DH == 32 || DH == 16, "head width");
  const int tid = threadIdx.x;
  const int w = tid >> 6, lane = tid & 63, c = lane & 31, hh = lane >> 5;
  const int seq = blockIdx.x, head = blockIdx.y;
  const int len = g.len;
  const int64_t tok0 = seq_token_base(g, seq);
  const int tstride = seq_token_stride(g);
  const int ld = 3 * N;
  const float* base = qkv + head * DH;
  const int nkb = (len + 31) / 32;
  const bool dcol = c < DH;                       // DH = 16: only half of the 32 V^T rows exist
  const int cd = dcol ? c : 0;

  for (int qb = w; qb < nkb; qb += 4) {
    // Q fragments (B operand of S^T = K Q^T), pre-scaled by log2(e)/sqrt(dh)
    const int pq = qb * 32 + c;
    float qf[DH / 2];
    {
      const float* row = base + (tok0 + (int64_t)(pq < len ? pq : 0) * tstride) * ld + 4 * hh;
#pragma unroll
      for (int m = 0; m < DH / 8; ++m) {
        float4 v = *reinterpret_cast<const float4*>(row + 8 * m);
        if (pq >= len) v = make_float4(0.f, 0.f, 0.f, 0.f);
        qf[4 * m + 0] = v.x * scale_log2e;
        qf[4 * m + 1] = v.y * scale_log2e;
        qf[4 * m + 2] = v.z * scale_log2e;
        qf[4 * m + 3] = v.w * scale_log2e;
      }
    }
    // key-block operands: K rows (A of S^T: lane = key c, k-slot hh) and V^T (A of O^T: lane = d c, k-slot hh <-> key
    // ROW32(r,hh)); rows beyond len read row 0 and are masked / zeroed.  Element offsets are 32-bit (host-checked);
    // two statically named buffers (a runtime-indexed register array would be demoted to scratch).
    const unsigned koff = (unsigned)(N + 4 * hh), voff = (unsigned)(2 * N + cd);
    const unsigned tbase = (unsigned)tok0, ts = (unsigned)tstride, ldu = (unsigned)ld;
    auto fetch = [&](int kb, float4 (&kf)[DH / 8], float (&vf)[16]) {
      const int key = kb * 32 + c;
      const float* krow = base + (tbase + (unsigned)(key < len ? key : 0) * ts) * ldu + koff;
#pragma unroll
      for (int m = 0; m < DH / 8; ++m) kf[m] = *reinterpret_cast<const float4*>(krow + 8 * m);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kr = kb * 32 + ROW32(r, hh);
        const float v = base[(tbase + (unsigned)(kr < len ? kr : 0) * ts) * ldu + voff];
        vf[r] = (kr < len && dcol) ? v : 0.f;
      }
    };
    float mrun = -1e30f, lrun = 0.f;
    f32x16 o = zero16();
    auto block = [&](int kb, const float4 (&kf)[DH / 8], const float (&vf)[16]) {
      f32x16 s = zero16();
#pragma unroll
      for (int m = 0; m < DH / 8; ++m) {
        s = mfma32(kf[m].x, qf[4 * m + 0], s);
        s = mfma32(kf[m].y, qf[4 * m + 1], s);
        s = mfma32(kf[m].z, qf[4 * m + 2], s);
        s = mfma32(kf[m].w, qf[4 * m + 3], s);
      }
      if ((kb + 1) * 32 > len) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kb * 32 + ROW32(r, hh) >= len) s[r] = -1e30f;
      }
      float mx = s[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float mnew = fmaxf(mrun, mx);
      const float alpha = fast_exp2(mrun - mnew);
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[r] = fast_exp2(s[r] - mnew);
        sum += s[r];
      }
      sum += __shfl_xor(sum, 32);
      lrun = lrun * alpha + sum;
      mrun = mnew;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] *= alpha;
      // O^T += V^T P^T : A = V^T (lane = d, slot <-> key ROW32(r,hh)), B = the S^T registers (lane = query)
#pragma unroll
      for (int r = 0; r < 16; ++r) o = mfma32(vf[r], s[r], o);
    };
    float4 kfa[DH / 8], kfb[DH / 8];
    float vfa[16], vfb[16];
    fetch(0, kfa, vfa);
    for (int kb = 0; kb < nkb; kb += 2) {
      if (kb + 1 < nkb) fetch(kb + 1, kfb, vfb);
      block(kb, kfa, vfa);
      if (kb + 1 < nkb) {
        if (kb + 2 < nkb) fetch(kb + 2, kfa, vfa);
        block(kb + 1, kfb, vfb);
      }
    }
    // o[r] = O^T[d = ROW32(r,hh)][query c]: four 16-byte pieces of the query's output row
    if (pq < len) {
      const float inv = fast_rcp(lrun);
      float* orow = out + (tok0 + (int64_t)pq * tstride) * N + head * DH;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int d0 = 8 * j + 4 * hh;
        if (d0 < DH)
          *reinterpret_cast<float4*>(orow + d0) =
              make_float4(o[4 * j + 0] * inv, o[4 * j + 1] * inv, o[4 * j + 2] * inv, o[4 * j + 3] * inv);
      }
    }
  }
}
