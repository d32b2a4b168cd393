// common.h -- shared device helpers for the gfx950 DPTN kernels.
//
// Conventions used by every kernel in this directory
//   * wave = 64 lanes; c = lane & 31 ("column" lane), hh = lane >> 5 (half).
//   * fp32 MFMA v_mfma_f32_32x32x2_f32:  D[i][j] += sum_{slot<2} A[i][slot] * B[slot][j]
//       A operand: lane (i=c, slot=hh) supplies one float; B operand: lane (j=c, slot=hh).
//       C/D: reg r of lane (c,hh) is element (row = ROW32(r,hh), col = c).
//   * k-permutation: operands are fetched as float4 (16 B) so MFMA step s = 4m+t (t<4) uses the
//     true k index 8m + 4hh + t for BOTH operands -- any bijection of k is a valid summation
//     order as long as A and B agree.  This makes every fragment fetch a ds_read_b128 /
//     global_load_dwordx4.
#pragma once
#include <atomic>
#include <cstdint>
#include <type_traits>
#include <utility>
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define DEV __device__ __forceinline__

// ---- split-precision helpers (OPT-IN mode "split_bf16"): x = hi + lo with hi = bf16(x), lo = bf16(x - hi) ------------
// v_mfma_f32_32x32x16_bf16: lane (c = l & 31, hh = l >> 5) holds A[row c][k = 8 hh + j] / B[k = 8 hh + j][col c], j < 8;
// the accumulator layout is the one of v_mfma_f32_32x32x2_f32 (ROW32), so epilogues do not care which produced it.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
DEV void split_bf16(float x, __bf16& hi, __bf16& lo) {
  hi = (__bf16)x;
  lo = (__bf16)(x - (float)hi);
}

// row index of accumulator register r (0..15) for half hh in a 32x32 tile
#define ROW32(r, hh) (((r) & 3) + 8 * ((r) >> 2) + 4 * (hh))

DEV f32x16 mfma32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
DEV f32x16 mfma32_bf16(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }

DEV f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}

// v_exp_f32 / v_rcp_f32 directly (1 ulp each): sigma(x) = 1/(1+2^(-x log2 e)).  __frcp_rn / expf would expand
// to ~10-instruction IEEE sequences, which showed up as 40 % of the LSTM step's VALU work.
DEV float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
DEV float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
DEV float fast_sigmoid(float x) { return fast_rcp(1.0f + fast_exp2(-1.4426950408889634f * x)); }
// 2/(1+e^-2x) - 1 : saturates correctly to +-1 (no inf/inf)
DEV float fast_tanh(float x) { return fmaf(2.0f, fast_rcp(1.0f + fast_exp2(-2.8853900817779268f * x)), -1.0f); }

// LSTM cell update for TWO accumulator slots at once, from PRE-SCALED gate pre-activations (scale -log2e for i, f, o and
// -2 log2e for g: sigmoid = rcp(1 + exp2(a)), tanh = 2 rcp(1 + exp2(a)) - 1).  Written on 2-vectors so that the
// non-transcendental half of the work issues as packed fp32 instructions (v_pk_add/mul/fma_f32: two slots per issue).
typedef float f32x2 __attribute__((ext_vector_type(2)));
DEV f32x2 exp2_2(f32x2 a) { return (f32x2){fast_exp2(a.x), fast_exp2(a.y)}; }
DEV f32x2 rcp_2(f32x2 a) { return (f32x2){fast_rcp(a.x), fast_rcp(a.y)}; }
struct LstmCell2 {
  f32x2 i, f, g, o, c, h;
};
DEV LstmCell2 lstm_cell2(f32x2 ai, f32x2 af, f32x2 ag, f32x2 ao, f32x2 c_prev) {
  LstmCell2 r;
  r.i = rcp_2(1.0f + exp2_2(ai));
  r.f = rcp_2(1.0f + exp2_2(af));
  r.g = 2.0f * rcp_2(1.0f + exp2_2(ag)) - 1.0f;
  r.o = rcp_2(1.0f + exp2_2(ao));
  r.c = r.f * c_prev + r.i * r.g;
  const f32x2 th = 2.0f * rcp_2(1.0f + exp2_2(r.c * -2.8853900817779268f)) - 1.0f;
  r.h = r.o * th;
  return r;
}

// Sum over groups of 16 or 32 adjacent lanes, result in every lane, entirely on the VALU (DPP + one
// v_permlane16_swap): __shfl_xor lowers to ds_bpermute_b32, whose LDS-crossbar round trip (~100 cycles, waited
// for with lgkmcnt(0)) dominated the LayerNorm epilogues.
template <int CTRL>
DEV float dpp_move(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int GROUP>
DEV float group_sum(float v) {
  static_assert(GROUP == 16 || GROUP == 32, "group size");
  v += dpp_move<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_move<0x4E>(v);    // quad_perm [2,3,0,1]  -> quad sums
  v += dpp_move<0x141>(v);   // row_half_mirror       -> 8-lane sums
  v += dpp_move<0x140>(v);   // row_mirror            -> 16-lane sums
  if (GROUP == 32) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);   // rows {0,1} and {2,3} exchange
    v = __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
  }
  return v;
}

// Counter-based dropout randomness for the attention probabilities (train mode, dptn.py:16-21): a pure function of
// (seed, query token x head, key position), so the forward and both backward phases regenerate identical masks.
// (lowbias32 mixer)
DEV uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
// Two levels: a full-strength seed per (query token, head) -- computed once per query and hoisted or staged through
// LDS by the kernels -- and a cheap per-key step on top of it: two multiply-xorshift rounds on 24-bit multiplies
// (v_mul_u32_u24 / v_mad_u32_u24 issue at full rate, v_mul_lo_u32 at a quarter) returning 24 random bits.
DEV uint32_t drop_qseed(uint32_t seed, uint32_t qtok_head) { return mix32(seed ^ (qtok_head * 0x9E3779B9u)); }
DEV uint32_t drop_rand_q(uint32_t qseed, uint32_t key) {
  // (HIP declares __umul24 as returning int: cast before shifting, or the shifts are arithmetic)
  uint32_t x = qseed + (uint32_t)__umul24(key, 0x9E3779u);
  x ^= x >> 15;
  x = (uint32_t)__umul24(x, 0xB5297Bu) >> 6;     // bits 6..31 of the product's low word: the well-mixed ones
  x ^= x >> 11;
  x = (uint32_t)__umul24(x, 0x68E31Du) >> 8;
  return x & 0xffffffu;
}
// keep  <=>  drop_rand(...) >= thresh24,  thresh24 = p * 2^24
DEV uint32_t drop_rand(uint32_t seed, uint32_t qtok_head, uint32_t key) {
  return drop_rand_q(drop_qseed(seed, qtok_head), key);
}
struct DropCfg {
  uint32_t seed;     // already mixed with the call index (block, path, step)
  uint32_t thresh;   // p * 2^24 (0: no dropout)
  float inv_keep;    // 1 / (1 - p)
};

// Sequence geometry of one TransformerDPRNN call over the token tensor x[b][s][k][n].
//   mode 0 (intra-chunk): sequence q = b*S + s, position t = k   -> token q*K + t
//   mode 1 (inter-chunk): sequence q = b*K + k, position t = s   -> token (b*S + t)*K + k
// v where keep, zeros elsewhere, as a bit mask: written as a select on the loaded value the compiler turns it back into
// a branch around the load (a basic block per load; see ALoadSeqShift::load4z)
DEV float4 mask4(float4 v, bool keep) {
  const unsigned m = keep ? 0xffffffffu : 0u;
  return make_float4(__uint_as_float(__float_as_uint(v.x) & m), __uint_as_float(__float_as_uint(v.y) & m),
                     __uint_as_float(__float_as_uint(v.z) & m), __uint_as_float(__float_as_uint(v.w) & m));
}

struct SeqGeom {
  int mode;
  int B, S, K;
  int nseq;  // B*S or B*K
  int len;   // K or S
  int nst;   // ceil(nseq/32) sequence tiles
};

DEV int64_t seq_token_base(const SeqGeom& g, int q) {
  if (g.mode == 0) return (int64_t)q * g.K;
  int b = q / g.K, k = q - b * g.K;
  return (int64_t)b * g.S * g.K + k;
}
DEV int seq_token_stride(const SeqGeom& g) { return g.mode == 0 ? 1 : g.K; }

static inline SeqGeom make_geom(int mode, int B, int S, int K) {
  SeqGeom g;
  g.mode = mode;
  g.B = B;
  g.S = S;
  g.K = K;
  g.nseq = mode == 0 ? B * S : B * K;
  g.len = mode == 0 ? K : S;
  g.nst = (g.nseq + 31) / 32;
  return g;
}

// Pre-activation (x W_ih^T + b) layout consumed by the LSTM kernel, in floats:
//   PRE[d][st][t][cb(16)][q(4)][hh(2)][c(32)][i(4)]
// where for direction d, sequence tile st, position t: gate column j = cb*32 + c (j = g*H + u),
// tile row rho = 8q + 4hh + i.  A lane's four consecutive accumulator registers 4q..4q+3 are one
// float4, and one wave instruction moves 1 KiB contiguously.
DEV int64_t pre_tile_offset(int d, int st, int t, int nst, int len) {
  return (((int64_t)d * nst + st) * len + t) * (int64_t)(512 * 32);
}


// Tile tickets of the token-tile kernels (GEMM engine, weight-gradient kernels).  Default: a device-wide counter (zeroed by
// the host), so a workgroup that starts late -- the kernel shares the chip with another stream's -- simply finds fewer
// tiles left.  With queue == nullptr (option "deterministic") the static sequence first, first + stride, ...: WHICH
// workgroup sums WHICH tiles is then fixed, and so is the rounding of every token reduction (bias / weight gradients).
// `queue` is a kernel argument, so `dynamic()` is wave-uniform.  The static sequence never goes through the ticket register
// or LDS (every wave computes tile + stride itself): sharing the register with the atomic's result made the compiler wait
// for "a possibly outstanding atomic" before every static update, i.e. for the A-tile prefetch just issued.
struct TileTickets {
  unsigned* queue;
  int first, stride;
  DEV bool dynamic() const { return queue != nullptr; }
  DEV int take() const { return (int)atomicAdd(queue, 1u); }   // dynamic() only, one lane
};

// Host helper: "done once per HIP device" flags for function attributes (hipFuncSetAttribute applies to the device
// that is current; a process may drive several GPUs through several engines).  Up to 64 devices.
// Atomic: two engines may be driven from two host threads (each handle is single-threaded, the statics are shared);
// the guarded work (hipFuncSetAttribute, an occupancy query) is idempotent, so a lost race only repeats it.
struct PerDeviceOnce {
  std::atomic<uint64_t> mask{0};
  bool done(int dev) const { return dev >= 0 && dev < 64 && ((mask.load(std::memory_order_acquire) >> dev) & 1u); }
  void set(int dev) { if (dev >= 0 && dev < 64) mask.fetch_or((uint64_t)1 << dev, std::memory_order_release); }
};
inline int current_hip_device() {
  int d = 0;
  return hipGetDevice(&d) == hipSuccess ? d : 0;
}
