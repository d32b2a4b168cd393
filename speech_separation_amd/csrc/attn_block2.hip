// attn_block2.hip -- the fused attention half of a TransformerDPRNN (attn_block.hip) with its two LayerNorms in FRAGMENT space
// (round 5; inference, N = 128, 4 heads of 32, sequences of <= 160 positions):
//
//     x  = LayerNorm2( ReLU(h) W_f^T + b_f + y1_prev )      (prologue: the FFN half of the previous path, dptn.py:50-51)
//     y1 = LayerNorm1( MHA(x) W_o^T + b_o + x )              src/model/dptn.py:46-47 (nn.MultiheadAttention :16-21, ln1 :22)
//
// What attn_block.hip does between its MFMA blocks, and what the phase stamps charge for it (profiles/r04_attn_block_phase_stamps.txt,
// cycles per sequence and wave of 235 k): both LayerNorms run in ROW space -- the product tile of every 32-token block goes
// through LDS (64 ds_write_b32 per lane for the four heads' out-projection partial tiles, 16 for the FFN product), a thread owns
// (row, 4 columns) and reduces each row over 32 lanes with two five-step DPP chains: 9.5 k + 7.9 k cycles of vector work, 5.2 k +
// 2.3 k of LDS traffic and barriers, 5.6 k of staging h rows through registers.  With fp32 MFMAs on the vector pipe every one of
// those instructions is paid in full, so the only lever is to have fewer of them.  This kernel keeps the
// MFMA side of attn_block.hip (K^T / V tiles resident in AGPRs, streaming softmax, same instruction stream) and changes the rest:
//   * both token-wise products come out TRANSPOSED, wave w = output columns [32 w, 32 w + 32): reg r of lane (c, hh) =
//     Y[token c][32 w + ROW32(r, hh)].  A token's 32 columns then sit in ONE lane pair, a row reduction is 15 in-lane adds + one
//     v_permlane32_swap, bias / residual / gamma / beta are 16-byte fragment reads, and the four waves only exchange (mean, M2)
//     of their 32 columns through 1 KiB of LDS -- Chan's exact merge of the four groups, so the statistics are those of a
//     two-pass LayerNorm.  The out-projection is formed per COLUMN tile, not per head: the heads' O^T tiles (16 registers, already
//     the B-operand layout) meet in LDS (4 x ds_write_b128 + 16 x ds_read_b128 per lane instead of 64 + 16 partial-tile accesses)
//     and the MFMA accumulates over the heads, so no partial tiles are summed by hand;
//   * the h rows of the prologue arrive by LDS-DMA (global_load_lds_dwordx4: 1 KiB = one row per wave instruction, two blocks in
//     flight, no registers, no ds_write), the residual rows as fragment loads in inline assembly, all of it counted by hand
//     (s_waitcnt vmcnt(12)): the compiler does not know those requests exist and therefore does not wait for them anywhere else.
// LDS: token rows 84.5 KB + two h blocks 66.6 KB (the O^T exchange reuses the first) + statistics 1 KB + constants 3 KB.
// Rounding differs from attn_block.hip in the LayerNorm statistics only (grouping of the sums); both are exact fp32 FMA chains.
#include <hip/hip_runtime.h>

#include "attn_block.h"

namespace {

constexpr int N = 128, DH = 32;
constexpr int LDX = N + 4;                  // token rows: conflict-free ds_read_b128 fragments
constexpr int LDHC = 256 + 4;               // h rows (1 040 bytes: a DMA request per row, 16-byte aligned)
constexpr int OB_FLOATS = 4 * 4 * 64 * 4;   // O^T exchange: [head][j][lane][4]

typedef float f32x4 __attribute__((ext_vector_type(4)));

DEV float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }
DEV uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p; }
// One LDS-DMA request: lane L's 16 bytes at (wave-uniform base + voff) land at LDS byte address lds_base + 16 L.  Inline assembly on
// purpose (fcln.hip has the long version): issued through the builtin, the compiler waits with vmcnt(0) in front of every later LDS
// access.  Every wait on this traffic is written by hand below.  (s_nop: the wait state the ISA asks for between a scalar write of
// M0 and an LDS-DMA that reads it; the hazard recogniser does not look into inline assembly.)
DEV void dma_row(const void* sbase, uint32_t voff, uint32_t lds_base) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_base), "v"(voff), "s"(sbase) : "memory", "m0");
}
// a fragment load the compiler does not count (same reason): 16 bytes at base + voff + OFF
template <int OFF>
DEV void ldg4_uncounted(f32x4& dst, const void* sbase, uint32_t voff) {
  asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}
// ... into the AGPR half of the register file (the W_f fragments: A operands of the prologue's MFMAs, which read them there)
template <int OFF>
DEV void ldg4_uncounted_a(f32x4& dst, const void* sbase, uint32_t voff) {
  asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:%3" : "=a"(dst) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}
// wait until at most KEEP of this wave's vector-memory requests are outstanding; sixteen AGPR quads are operands so that no use of
// them can be scheduled in front of the wait (an asm statement takes at most 30 operands: call it once per half of W_f)
template <int KEEP>
DEV void wait_vm_a16(f32x4* r) {
  asm volatile("s_waitcnt vmcnt(%[n])"
               : "+a"(r[0]), "+a"(r[1]), "+a"(r[2]), "+a"(r[3]), "+a"(r[4]), "+a"(r[5]), "+a"(r[6]), "+a"(r[7]), "+a"(r[8]), "+a"(r[9]),
                 "+a"(r[10]), "+a"(r[11]), "+a"(r[12]), "+a"(r[13]), "+a"(r[14]), "+a"(r[15])
               : [n] "n"(KEEP)
               : "memory");
}
// wait until at most KEEP of this wave's vector-memory requests are outstanding; the four registers are operands so that no use
// of them can be scheduled in front of the wait
template <int KEEP>
DEV void wait_vm(f32x4 (&r)[4]) {
  asm volatile("s_waitcnt vmcnt(%[n])" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : [n] "n"(KEEP) : "memory");
}

constexpr int kb_mfmas(int kb, int nkb) { return (kb > 0 ? 16 : 0) + (kb + 1 < nkb ? 16 : 0); }
template <class F, int... I>
DEV void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N_, class F>
DEV void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N_>{});
}

// max / sum across the two 32-lane halves (lanes (c,0) and (c,1) hold the two halves of a query's keys / a token's columns)
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

// ---- LayerNorm over 128 columns held as 4 waves x (lane pair x 16 registers) -------------------------------------------------
// Step 1 (per wave): mean and centred sum of squares of the wave's 32 columns of token c; v becomes v - mean_w.
DEV void ln_group_stats(f32x16& v, float& mu_w, float& m2_w) {
  f32x2 s0 = (f32x2){v[0], v[1]} + (f32x2){v[2], v[3]}, s1 = (f32x2){v[4], v[5]} + (f32x2){v[6], v[7]};
  s0 += (f32x2){v[8], v[9]};
  s1 += (f32x2){v[10], v[11]};
  s0 += (f32x2){v[12], v[13]};
  s1 += (f32x2){v[14], v[15]};
  const f32x2 s = s0 + s1;
  mu_w = half_sum(s.x + s.y) * (1.0f / 32);
  const f32x2 m2 = (f32x2){mu_w, mu_w};
  f32x2 q0 = (f32x2){0.f, 0.f}, q1 = (f32x2){0.f, 0.f};
#pragma unroll
  for (int r = 0; r < 16; r += 4) {
    const f32x2 d0 = (f32x2){v[r], v[r + 1]} - m2, d1 = (f32x2){v[r + 2], v[r + 3]} - m2;
    v[r] = d0.x; v[r + 1] = d0.y; v[r + 2] = d1.x; v[r + 3] = d1.y;
    q0 += d0 * d0;
    q1 += d1 * d1;
  }
  const f32x2 q = q0 + q1;
  m2_w = half_sum(q.x + q.y);
}
// Step 2 (after the exchange): the row's mean and 1/sigma from the four groups' (mean, M2) -- Chan et al.'s pairwise merge written
// for four groups of equal size: M2 = sum M2_w + 32 sum (mean_w - mean)^2.  Returns mean_w - mean of THIS wave in `delta`.
DEV float ln_merge(const float4 a, const float4 b, float mu_w, float& delta) {
  const float mu = ((a.x + a.z) + (b.x + b.z)) * 0.25f;
  const float d0 = a.x - mu, d1 = a.z - mu, d2 = b.x - mu, d3 = b.z - mu;
  const float m2 = ((a.y + a.w) + (b.y + b.w)) + 32.0f * ((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
  delta = mu_w - mu;
  return rsqrtf(m2 * (1.0f / N) + 1e-5f);
}

struct FfnPro2 {
  const float* hc;      // [M][256] ReLU(h_fwd | h_bwd) of the previous path
  const float* wf;      // packed ffn.1.weight (attn_pack_launch)
  const float* bf;
  const float* g2;      // ln2 weight / bias of the previous path
  const float* b2;
};

#ifdef ATTN_STAMPS
__device__ unsigned long long g_ab2_stamps[16];
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
    for (int ab_i = 0; ab_i < 12; ++ab_i) atomicAdd(&g_ab2_stamps[ab_i], ab_t[ab_i]);               \
    atomicAdd(&g_ab2_stamps[12], 1ull);                                                             \
  }
#else
#define AB_DECL
#define AB_MARK(i)
#define AB_KEEP(x)
#define AB_END
#endif

template <int NKB, bool PRO>
__global__ __launch_bounds__(256) void attn_block2_kernel(const float* __restrict__ x, const float* __restrict__ wp_in,
                                                          const float* __restrict__ b_in, const float* __restrict__ wp_o,
                                                          const float* __restrict__ b_o, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ y1, SeqGeom g,
                                                          float scale_log2e, FfnPro2 pro) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Xs = smem;                                   // [NKB*32][LDX]  the sequence's token rows (rows >= len repeat the last one)
  float* Hs = smem + NKB * 32 * LDX;                  // PRO: [2][32][LDHC] h blocks (DMA targets)
  float* Ob = Hs;                                     // phase 2: [4 heads][4][64 lanes][4]  O^T tiles of one query block
  float* Red = Hs + (PRO ? 2 * 32 * LDHC : OB_FLOATS);  // [2][32 tokens][4 waves][2]  (mean_w, M2_w)
  float* Cst = Red + 512;                             // b_o | gamma | beta | (PRO) b_f | g2 | b2, 128 floats each
  const int tid = threadIdx.x;
  const int h = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave = head (attention) = column tile (token-wise products)
  const int lane = tid & 63, c = lane & 31, hh = lane >> 5;
  const int seq = blockIdx.x, len = g.len;
  const int64_t tok0 = seq_token_base(g, seq);
  const int tstride = seq_token_stride(g);
  const unsigned rs_y = (unsigned)tstride * 512u;        // bytes between consecutive positions in x / y1 [M][128]
  char* const ybase = reinterpret_cast<char*>(y1) + tok0 * 512;
  auto row_off = [&](int row, unsigned rstride, bool clamp) -> unsigned {      // row = uniform part + lane part, < 2^24
    if (clamp) row = row < len ? row : len - 1;
    return (unsigned)__umul24((unsigned)row, rstride);
  };
  const unsigned frag_col = (unsigned)(32 * h + 4 * hh);      // first of this lane's columns 32 h + 8 j + 4 hh + t

  AB_DECL
  // per-column constants of the two LayerNorms -> LDS (read back as 16-byte fragments where they are used)
  if (tid < (PRO ? 192 : 96)) {
    const int which = tid >> 5, q4 = tid & 31;
    const float* src = which == 0 ? b_o : which == 1 ? gamma : which == 2 ? beta : which == 3 ? pro.bf : which == 4 ? pro.g2 : pro.b2;
    *reinterpret_cast<float4*>(&Cst[which * 128 + 4 * q4]) = ldg4(src + 4 * q4);
  }
  auto cst16 = [&](int which, f32x16& d) {      // d[4 j + t] = constant[32 h + 8 j + 4 hh + t]
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 t = *reinterpret_cast<const float4*>(&Cst[which * 128 + frag_col + 8 * j]);
      d[4 * j + 0] = t.x; d[4 * j + 1] = t.y; d[4 * j + 2] = t.z; d[4 * j + 3] = t.w;
    }
  };
  // the exchange of the LayerNorm statistics: both halves of a lane pair hold the same two values and write the same 8 bytes
  // (two sets, by the parity of the block: with one barrier per block a wave may publish block rb while another still merges rb - 1)
  auto red_put = [&](int par, float mu_w, float m2_w) {
    *reinterpret_cast<float2*>(&Red[par * 256 + (c * 4 + h) * 2]) = make_float2(mu_w, m2_w);
  };
  auto red_get = [&](int par, float mu_w, float& delta) {
    const float4 a = *reinterpret_cast<const float4*>(&Red[par * 256 + c * 8]), b = *reinterpret_cast<const float4*>(&Red[par * 256 + c * 8 + 4]);
    return ln_merge(a, b, mu_w, delta);
  };

  float wkf[64], wvf[64];
  auto fetch_wkv1 = [&](int m) {      // k-chunk m of this head's W_k / W_v fragments
    const float* wk = wp_in + ((1 * 4 + h) * 16 * 64 + lane) * 4;   // packed: [sel][head][m][lane][4]
    const float* wv = wp_in + ((2 * 4 + h) * 16 * 64 + lane) * 4;
    const float4 a = ldg4(wk + m * 256), b = ldg4(wv + m * 256);
    wkf[4 * m + 0] = a.x; wkf[4 * m + 1] = a.y; wkf[4 * m + 2] = a.z; wkf[4 * m + 3] = a.w;
    wvf[4 * m + 0] = b.x; wvf[4 * m + 1] = b.y; wvf[4 * m + 2] = b.z; wvf[4 * m + 3] = b.w;
  };
  auto fetch_wkv = [&]() {
#pragma unroll
    for (int m = 0; m < 16; ++m) fetch_wkv1(m);
  };
  if constexpr (PRO) {
    // ---- prologue: x rows = LN2(ReLU(h) W_f^T + b_f + y1_prev) of the previous path, block of 32 tokens at a time ----
    // W_f[32h + c][8m + 4hh + t]: this wave's 32 output columns, K = 256 (A operand).  Loaded by uncounted requests like the rows:
    // counted by the compiler, their wait (vmcnt(0) in its book) would also sit out the first two blocks requested behind them.
    f32x4 wf4[32];
    const uint32_t lane16 = (uint32_t)lane * 16u;
    {
      const char* wb = reinterpret_cast<const char*>(pro.wf + h * 32 * 64 * 4);        // packed: [head][m][lane][4]
      static_for<8>([&](auto MQ) {
        constexpr int mq = decltype(MQ)::value;
        ldg4_uncounted_a<0>(wf4[4 * mq + 0], wb + mq * 4096, lane16);
        ldg4_uncounted_a<1024>(wf4[4 * mq + 1], wb + mq * 4096, lane16);
        ldg4_uncounted_a<2048>(wf4[4 * mq + 2], wb + mq * 4096, lane16);
        ldg4_uncounted_a<3072>(wf4[4 * mq + 3], wb + mq * 4096, lane16);
      });
    }
    const uint32_t hs_lds = lds_addr(Hs);
    const char* const hbase = reinterpret_cast<const char*>(pro.hc) + tok0 * 1024;      // wave-uniform: row pointers stay in SGPRs
    const uint32_t rs_h = (uint32_t)tstride * 1024u;                                     // bytes between consecutive positions in hc
    // Requests of the prologue, all counted by hand: D(b) = 8 DMA requests (wave w: rows 8 w .. 8 w + 7 of block b -> h buffer
    // hbuf(b)), R(b) = the 4 residual fragments of this lane's token (y1_prev[token c][32 h + 8 j + 4 hh ..]).
    // Three h buffers from NKB >= 4: the third is the part of the token-row tile that is written last (x rows of the last two
    // blocks: 33 792 bytes for 33 280), free until block NKB - 2 is normalised -- so D(b) goes out TWO blocks ahead, behind the
    // barrier that opens block b - 2 (its buffer's last reader, block b - 3, is through by then).  Measured with two buffers and
    // one barrier per block the request had one block (~9 k cycles) and the wait for it was 2 k cycles per block, as run.
    constexpr bool TRI = NKB >= 4;
    auto hbuf_lds = [&](int b) -> uint32_t {      // LDS byte address of the buffer of block b
      if (TRI) return (b % 3) == 2 ? lds_addr(Xs) + (uint32_t)((NKB - 2) * 32 * LDX * 4) : hs_lds + (uint32_t)((b % 3) * 32 * LDHC * 4);
      return hs_lds + (uint32_t)((b & 1) * 32 * LDHC * 4);
    };
    auto hbuf = [&](int b) -> const float* { return TRI ? ((b % 3) == 2 ? Xs + (NKB - 2) * 32 * LDX : Hs + (b % 3) * 32 * LDHC) : Hs + (b & 1) * 32 * LDHC; };
    f32x4 res[2][4];
    // one request of block rb: k = 0..7 DMA row 8 h + k, k = 8..11 residual fragment k - 8 (k is a constant after unrolling)
    auto issue_d1 = [&](auto RB, int i) {
      constexpr int rb = decltype(RB)::value;
      int row = rb * 32 + 8 * h + i;
      if (rb == NKB - 1) row = row < len ? row : len - 1;
      dma_row(hbase + (uint32_t)row * rs_h, lane16, hbuf_lds(rb) + (uint32_t)((8 * h + i) * LDHC * 4));
    };
    auto issue_r1 = [&](auto RB, f32x4 (&rr)[4], int j) {
      constexpr int rb = decltype(RB)::value;
      const unsigned ro = row_off(rb * 32 + c, rs_y, rb == NKB - 1) + 4u * frag_col;
      switch (j) {
        case 0: ldg4_uncounted<0>(rr[0], ybase, ro); break;
        case 1: ldg4_uncounted<32>(rr[1], ybase, ro); break;
        case 2: ldg4_uncounted<64>(rr[2], ybase, ro); break;
        default: ldg4_uncounted<96>(rr[3], ybase, ro); break;
      }
    };
    auto issue_d = [&](auto RB) {
#pragma unroll
      for (int i = 0; i < 8; ++i) issue_d1(RB, i);
    };
    auto issue_r = [&](auto RB, f32x4 (&rr)[4]) {
#pragma unroll
      for (int j = 0; j < 4; ++j) issue_r1(RB, rr, j);
    };
    // issue order: W_f | D(0) R(0) | inside block 0: R(1) D(1) [D(2)] | inside block rb >= 1: R(rb + 1), then D(rb + AHEAD)
    // (AHEAD = 2 with three buffers, else 1; the first fetch of a sequence is 44 requests back to back -- D(1) is not one of them)
    constexpr int AHEAD = TRI ? 2 : 1;
    issue_d(std::integral_constant<int, 0>{});
    issue_r(std::integral_constant<int, 0>{}, res[0]);
    constexpr int BEHIND_WF = 12;                      // requests behind the W_f loads
    wait_vm_a16<BEHIND_WF>(wf4);                       // W_f is in; the first two blocks stay in flight
    wait_vm_a16<BEHIND_WF>(wf4 + 16);
    AB_MARK(0)
    // One barrier per block: B(rb) publishes the block's h rows AND the LayerNorm statistics of block rb - 1, whose merge /
    // normalise / x -> LDS then runs beside the 128 MFMAs of block rb (it reads LDS and has two dependent chains; alone it waits
    // out every one of them).
    f32x16 vprev = zero16();
    float mu_prev = 0.f;
    auto normalise_prev = [&](int pb) {        // x rows of block pb from vprev / mu_prev and the exchanged statistics
      float delta;
      const float rstd = red_get(pb & 1, mu_prev, delta);
      f32x16 g2c, b2c;
      cst16(4, g2c);
      cst16(5, b2c);
      const f32x2 d2 = (f32x2){delta, delta}, r2 = (f32x2){rstd, rstd};
      float* xw = &Xs[(pb * 32 + c) * LDX + frag_col];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x2 lo = ((f32x2){vprev[4 * j], vprev[4 * j + 1]} + d2) * r2 * (f32x2){g2c[4 * j], g2c[4 * j + 1]} + (f32x2){b2c[4 * j], b2c[4 * j + 1]};
        const f32x2 hi = ((f32x2){vprev[4 * j + 2], vprev[4 * j + 3]} + d2) * r2 * (f32x2){g2c[4 * j + 2], g2c[4 * j + 3]} +
                         (f32x2){b2c[4 * j + 2], b2c[4 * j + 3]};
        *reinterpret_cast<float4*>(xw + 8 * j) = make_float4(lo.x, lo.y, hi.x, hi.y);
      }
    };
    auto pro_block = [&](auto RB) {
      constexpr int rb = decltype(RB)::value;
      // D(rb) and R(rb) are in; behind them only the rows of block rb + 1 may be on their way: with AHEAD = 2, for 0 < rb < NKB - 1
      // (requested inside block rb - 1, behind R(rb)).  rb = 0: nothing else has been requested yet; last block: nothing is left.
      constexpr int KEEP = (AHEAD == 2 && rb >= 1 && rb + 1 < NKB) ? 8 : 0;
      wait_vm<KEEP>(res[rb & 1]);
      __syncthreads();
      AB_MARK(1)
      // The next requests -- R(rb + 1), then D(rb + AHEAD): the order the waits above count on -- go out ONE AT A TIME between the
      // MFMAs of the first three quarters of the block (behind every second k-chunk): issued back to back, twelve requests cost
      // 1.5-2.4 k cycles of a wave that has nothing else to issue (measured: profiles/r05_attn_block2_experiments.txt), beside
      // MFMAs nothing measurable.  The last quarter shares its region with the normalisation of block rb - 1.
      // block 0 also requests D(1) (between R(1) and D(2)): slots 0..3 R(rb + 1), 4..11 D(1) [rb = 0] or D(rb + AHEAD), 12..19 D(2)
      // [rb = 0, three buffers].  The LAST block has nothing left to request: its slots carry this head's W_k / W_v fragments
      // (ordinary loads; behind the wait above the compiler's own count is exact), which arrive behind its MFMAs.
      constexpr bool HAS_R = rb + 1 < NKB, HAS_D = rb + AHEAD < NKB && rb + AHEAD >= 2, HAS_D1 = rb == 0 && NKB > 1;
      constexpr int NREQ = rb == NKB - 1 ? 16 : 4 + (HAS_D1 ? 8 : 0) + (HAS_D ? 8 : 0);
      auto request = [&](int k) {
        if constexpr (rb == NKB - 1) {
          fetch_wkv1(k);
        } else if (k < 4) {
          if constexpr (HAS_R) issue_r1(std::integral_constant<int, HAS_R ? rb + 1 : 0>{}, res[(rb + 1) & 1], k);
        } else if (HAS_D1 && k < 12) {
          if constexpr (HAS_D1) issue_d1(std::integral_constant<int, 1>{}, k - 4);
        } else {
          if constexpr (HAS_D) issue_d1(std::integral_constant<int, HAS_D ? rb + AHEAD : 0>{}, k - (HAS_D1 ? 12 : 4));
        }
      };
      f32x16 a0 = zero16(), a1 = zero16();
      const float* ar = hbuf(rb) + c * LDHC + 4 * hh;
#pragma unroll
      for (int m0 = 0; m0 < 32; m0 += 8) {
        float4 af[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) af[m] = *reinterpret_cast<const float4*>(ar + 8 * (m0 + m));
#pragma unroll
        for (int m = 0; m < 8; ++m) {                          // C^T = W_f H^T: reg r of lane (c,hh) = C[token c][32 h + ROW32(r,hh)]
          a0 = mfma32(wf4[m0 + m][0], af[m].x, a0);
          a1 = mfma32(wf4[m0 + m][1], af[m].y, a1);
          a0 = mfma32(wf4[m0 + m][2], af[m].z, a0);
          a1 = mfma32(wf4[m0 + m][3], af[m].w, a1);
          // one request behind every k-chunk (20 requests: block 0 with three buffers) or every second one, in the first
          // three quarters of the block
          constexpr int EVERY = NREQ > 12 ? 1 : 2;
          if ((m0 + m) % EVERY == EVERY - 1 && (m0 + m) / EVERY < NREQ && (m0 + m) < 24) {
            __builtin_amdgcn_sched_barrier(0);        // (MFMAs are scheduled across a volatile asm statement otherwise: pin the place)
            request((m0 + m) / EVERY);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      if constexpr (rb >= 1) {
        normalise_prev(rb - 1);
#pragma unroll
        for (int i = 0; i < 32; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // one MFMA of the last quarter
          __builtin_amdgcn_sched_group_barrier(0x1f6, 4, 0);     // four of: VALU, SALU, VMEM, DS
        }
      }
      AB_KEEP(a0[15]) AB_KEEP(a1[15])
      AB_MARK(2)
      f32x16 bfc;
      cst16(3, bfc);
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2 t = ((f32x2){a0[r], a0[r + 1]} + (f32x2){a1[r], a1[r + 1]}) + (f32x2){bfc[r], bfc[r + 1]} +
                        (f32x2){res[rb & 1][r >> 2][r & 3], res[rb & 1][r >> 2][(r & 3) + 1]};
        vprev[r] = t.x;
        vprev[r + 1] = t.y;
      }
      float m2_w;
      ln_group_stats(vprev, mu_prev, m2_w);
      red_put(rb & 1, mu_prev, m2_w);
      AB_MARK(3)
    };
    static_for<NKB>(pro_block);
    __syncthreads();
    normalise_prev(NKB - 1);
    AB_MARK(4)
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
  // ---- phase 1: K^T and V of this head for every key block, kept in registers (attn_block.hip) ----------------
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
      for (int r = 0; r < 16; ++r) asm volatile("" : "+a"(kt[rb][r]), "+a"(vv[rb][r]));
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  AB_MARK(6)
  // ---- phase 2 constants: W_q fragments (parked in AGPRs), rows [32 h, 32 h + 32) of W_o (A operand of Y^T), the Q bias ----
#pragma unroll
  for (int i = 0; i < 64; ++i) asm volatile("" : "+a"(wqf[i]));
  float wo2[4][16];                                              // W_o[32 h + c][32 hp + ROW32(r,hh)]
#pragma unroll
  for (int hp = 0; hp < 4; ++hp) {
    const float* wr = wp_o + (((hp * 4 + h) * 4) * 64 + lane) * 4;   // packed: [head][jt][j][lane][4], jt = this wave
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 t = ldg4(wr + j * 256);
      wo2[hp][4 * j + 0] = t.x; wo2[hp][4 * j + 1] = t.y; wo2[hp][4 * j + 2] = t.z; wo2[hp][4 * j + 3] = t.w;
    }
  }
  float qbias[16];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float4 t = ldg4(b_in + h * DH + 8 * j + 4 * hh);
    qbias[4 * j + 0] = t.x; qbias[4 * j + 1] = t.y; qbias[4 * j + 2] = t.z; qbias[4 * j + 3] = t.w;
  }

  // ---- phase 2: one query block at a time -----------------------------------------------------------------
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
  f32x16 qa = zero16(), qb_ = zero16();
  q_tile_mfmas(0, qa, qb_);
  AB_KEEP(qa[15]) AB_KEEP(qb_[15]) AB_KEEP(wo2[3][15]) AB_KEEP(qbias[15])
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
    // Streaming softmax over the key blocks (attn_block.hip: two-stage pipeline, lazy rescale)
    float mrun = -1e30f, lrun = 0.f;
    f32x16 o = zero16();
    f32x16 s = zero16();
#pragma unroll
    for (int r = 0; r < 16; ++r) s = mfma32(kt[0][r], q[r], s);          // S^T[key ROW32(.,hh)][query c]
    f32x16 pprev = zero16();
    auto softmax_step = [&](auto KB) {
      constexpr int kb = decltype(KB)::value;
      __builtin_amdgcn_sched_barrier(0);
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
      if (kb == NKB - 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kb * 32 + ROW32(r, hh) >= len) s[r] = -1e30f;
      }
      float m4[4] = {s[0], s[1], s[2], s[3]};
#pragma unroll
      for (int r = 4; r < 16; ++r) m4[r & 3] = fmaxf(m4[r & 3], s[r]);
      const float mx = half_max(fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3])));
      const float mnew = (kb == 0 || mx > mrun + 8.0f) ? mx : mrun;
      const float alpha = fast_exp2(mrun - mnew);
      f32x16 p;
      f32x2 s2[2] = {(f32x2){0.f, 0.f}, (f32x2){0.f, 0.f}};
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
      constexpr int NM = kb_mfmas(kb, NKB);
      if constexpr (NM > 0) {
#pragma unroll
        for (int i = 0; i < NM; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, (120 + NM - 1) / NM, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
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
    // ---- the heads' O^T tiles meet in LDS: reg r of lane (c,hh) = O[query c][32 h + ROW32(r,hh)] is the B operand of Y^T as it is
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *reinterpret_cast<float4*>(&Ob[((h * 4 + j) * 64 + lane) * 4]) = make_float4(o[4 * j], o[4 * j + 1], o[4 * j + 2], o[4 * j + 3]);
    __syncthreads();
    // Y^T tile of this wave's 32 output columns: W_o[32 h + c][:] (A) x O^T of all heads (B), 64 MFMAs in two chains
    f32x16 ya = zero16(), yb = zero16();
#pragma unroll
    for (int hp = 0; hp < 4; ++hp) {
      float4 of[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) of[j] = *reinterpret_cast<const float4*>(&Ob[((hp * 4 + j) * 64 + lane) * 4]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        ya = mfma32(wo2[hp][4 * j + 0], of[j].x, ya);
        yb = mfma32(wo2[hp][4 * j + 1], of[j].y, yb);
        ya = mfma32(wo2[hp][4 * j + 2], of[j].z, ya);
        yb = mfma32(wo2[hp][4 * j + 3], of[j].w, yb);
      }
    }
    // + b_o + x (the residual rows come from the staged tile, as fragments), LayerNorm statistics of this wave's columns
    f32x16 v;
    {
      f32x16 boc;
      cst16(0, boc);
      const float* xr = &Xs[(qb * 32 + c) * LDX + frag_col];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 xres = *reinterpret_cast<const float4*>(xr + 8 * j);
        const f32x2 lo = ((f32x2){ya[4 * j], ya[4 * j + 1]} + (f32x2){yb[4 * j], yb[4 * j + 1]}) + (f32x2){boc[4 * j], boc[4 * j + 1]} +
                         (f32x2){xres.x, xres.y};
        const f32x2 hi = ((f32x2){ya[4 * j + 2], ya[4 * j + 3]} + (f32x2){yb[4 * j + 2], yb[4 * j + 3]}) +
                         (f32x2){boc[4 * j + 2], boc[4 * j + 3]} + (f32x2){xres.z, xres.w};
        v[4 * j] = lo.x; v[4 * j + 1] = lo.y; v[4 * j + 2] = hi.x; v[4 * j + 3] = hi.y;
      }
    }
    float mu_w, m2_w, delta;
    ln_group_stats(v, mu_w, m2_w);
    red_put(0, mu_w, m2_w);
    __syncthreads();
    AB_MARK(9)
    // normalise + store, beside the 64 MFMAs of the next block's Q^T tile (neither depends on the other).  Padded rows p >= len
    // carry the staged copy of row len-1 through the same arithmetic and are stored there once more.
    __builtin_amdgcn_sched_barrier(0);
    if (qb + 1 < NKB) {
      qa = zero16();
      qb_ = zero16();
      q_tile_mfmas(qb + 1, qa, qb_);
    }
    {
      const float rstd = red_get(0, mu_w, delta);
      f32x16 gac, bec;
      cst16(1, gac);
      cst16(2, bec);
      const f32x2 d2 = (f32x2){delta, delta}, r2 = (f32x2){rstd, rstd};
      char* yw = ybase + row_off(qb * 32 + c, rs_y, true) + 4u * frag_col;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x2 lo = ((f32x2){v[4 * j], v[4 * j + 1]} + d2) * r2 * (f32x2){gac[4 * j], gac[4 * j + 1]} + (f32x2){bec[4 * j], bec[4 * j + 1]};
        const f32x2 hi = ((f32x2){v[4 * j + 2], v[4 * j + 3]} + d2) * r2 * (f32x2){gac[4 * j + 2], gac[4 * j + 3]} +
                         (f32x2){bec[4 * j + 2], bec[4 * j + 3]};
        *reinterpret_cast<float4*>(yw + 32 * j) = make_float4(lo.x, lo.y, hi.x, hi.y);
      }
    }
    if (qb + 1 < NKB) {
#pragma unroll
      for (int i = 0; i < 64; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x1f6, 2, 0);     // up to two of: VALU, SALU, VMEM, DS
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    AB_KEEP(qa[15]) AB_KEEP(qb_[15])
    AB_MARK(10)
  }
  AB_END
}

}  // namespace

size_t attn_block2_lds_bytes(int nkb, bool pro) {
  return sizeof(float) * ((size_t)nkb * 32 * LDX + (pro ? 2 * 32 * LDHC : OB_FLOATS) + 512 + 6 * 128);
}

int attn_block2_launch(void* stream, const float* x, const float* b_in, const float* b_o, const float* gamma, const float* beta,
                       float* y1, const SeqGeom& g, const AttnFfnPrologue* pro, const float* wpack) {
  using Kern = void (*)(const float*, const float*, const float*, const float*, const float*, const float*, const float*, float*,
                        SeqGeom, float, FfnPro2);
  const int nkb = (g.len + 31) / 32;
  if (nkb < 1 || nkb > 5 || !wpack) return (int)hipErrorInvalidValue;
  const float scale_log2e = 1.4426950408889634f / sqrtf((float)DH);
  const int dev = current_hip_device();
  Kern kern;
  const bool p = pro != nullptr;
  switch (nkb) {
    case 1: kern = p ? attn_block2_kernel<1, true> : attn_block2_kernel<1, false>; break;
    case 2: kern = p ? attn_block2_kernel<2, true> : attn_block2_kernel<2, false>; break;
    case 3: kern = p ? attn_block2_kernel<3, true> : attn_block2_kernel<3, false>; break;
    case 4: kern = p ? attn_block2_kernel<4, true> : attn_block2_kernel<4, false>; break;
    default: kern = p ? attn_block2_kernel<5, true> : attn_block2_kernel<5, false>; break;
  }
  const size_t lds = attn_block2_lds_bytes(nkb, p);
  static PerDeviceOnce ready_all[2][6];
  PerDeviceOnce* ready = ready_all[p ? 1 : 0];
  if (!ready[nkb].done(dev)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    ready[nkb].set(dev);
  }
  FfnPro2 fp{};
  if (p) fp = FfnPro2{pro->hc, pro->wf, pro->bf, pro->g2, pro->b2};
  hipLaunchKernelGGL(kern, dim3(g.nseq), dim3(256), lds, static_cast<hipStream_t>(stream), x, wpack, b_in, wpack + ATTN_PACK_IN,
                     b_o, gamma, beta, y1, g, scale_log2e, fp);
  return (int)hipGetLastError();
}

#ifdef ATTN_STAMPS
// diagnostic build only (tools/attn_stamps.py): read / reset the phase sums
extern "C" int dptnav_debug_attn2_stamps(unsigned long long* out, int reset) {
  if (reset) {
    unsigned long long z[16] = {0};
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_ab2_stamps), z, sizeof(z));
  }
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ab2_stamps), 16 * sizeof(unsigned long long));
}
#endif
