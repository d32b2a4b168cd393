// dgrad_t.hip -- data gradient of a wide layer into the 128-feature stream, TRANSPOSED and weights-stationary (round 5):
//
//     out[M][128] = addend[M][128] + A[M][0:KIN] W[KIN][128]        KIN = 512: d x = d_out + dP_dir W_ih_dir (dptn.py:48, the
//                                                                     backward of nn.LSTM's input product), twice per path
//
// Why not the GEMM engine (gemm_ws.h, WT = true), which ran this shape at 0.65 MFMA-busy and 17 % of a training step's kernel
// time (profiles/r05_train_mfma_utilisation.txt)?  Its generated tile loop, KIN = 512, one column block per wave:
//   * ONE accumulator: 256 MFMAs in one dependent chain (each waits ~16 cycles for its predecessor's last pass: +4 k of 16.4 k);
//   * one fragment in flight: ds_read_b128 -> s_waitcnt lgkmcnt(0) -> 4 MFMAs, 64 times (the 256 weight registers + 64 staging
//     registers of the next tile leave no room for a batch);
//   * the next tile's 16 loads back to back in front of the block (2.7 k cycles: DESIGN.md section 3.0.3), 16 ds_write_b128 and a
//     barrier behind them, the C tile through LDS (16 ds_write_b32, a second barrier, a row-space pass);
//   * a weight prologue of 256 scalar loads.
// Here (the prologue of attn_block2.hip, which is the same product with K = 256, grown into a kernel of its own):
//   * the product is formed transposed -- wave w = output columns [32 w, 32 w + 32), register r of lane (c, hh) =
//     out[token c][32 w + ROW32(r, hh)] -- so addend and result are four 16-byte fragments per lane straight from / to memory: no C
//     tile, no second barrier, no row pass;
//   * the A rows arrive by LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave instruction, no staging registers, no ds_write), the
//     next tile's 16 requests per wave one at a time between the MFMAs of the first half of a tile, every wait counted by hand (the comments at the waits say
//     what may be outstanding);
//   * the 64 registers this frees hold two accumulation chains and two batches of eight fragments (the next batch is read while
//     the current one multiplies).
// One barrier per 32-token tile.  LDS: two row buffers of 32 x (KIN + 4) floats (132 KB at KIN = 512): one workgroup per CU, which
// the 256 weight registers per lane dictate anyway.
#include <hip/hip_runtime.h>

#include <type_traits>
#include <utility>

#include "common.h"
#include "dgrad_t.h"

namespace {

DEV uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p; }
// One LDS-DMA request (global_load_lds_dwordx4): lane L's 16 bytes at (wave-uniform base + voff) land at LDS byte address
// lds_base + 16 L (attn_block2.hip).  HALF: only lanes 0..31 take part (512 bytes: the tail of a 1 536-byte row) -- the upper half
// of EXEC is cleared around the request inside the one asm statement (EXEC is full wherever this is called).  s_nop: the wait
// state the ISA asks for between a scalar write of M0 and an LDS-DMA that reads it (the hazard recogniser does not look into
// inline assembly).
template <bool HALF>
DEV void dma_part(const void* sbase, uint32_t voff, uint32_t lds_base) {
  if constexpr (!HALF) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_base), "v"(voff), "s"(sbase) : "memory", "m0");
  } else {
    uint32_t saved;
    asm volatile("s_mov_b32 m0, %1\n\ts_mov_b32 %0, exec_hi\n\ts_mov_b32 exec_hi, 0\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 exec_hi, %0"
                 : "=&s"(saved)
                 : "s"(lds_base), "v"(voff), "s"(sbase)
                 : "memory", "m0");
  }
}
// fragment load / store the compiler does not count: 16 bytes at base + voff + OFF
template <int OFF>
DEV void ldg4_uncounted(f32x4& dst, const void* sbase, uint32_t voff) {
  asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}
// ... into the AGPR half of the register file
template <int OFF>
DEV void ldg4_uncounted_a(f32x4& dst, const void* sbase, uint32_t voff) {
  asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:%3" : "=a"(dst) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}
template <int OFF>
DEV void stg4_uncounted(void* sbase, uint32_t voff, f32x4 v) {
  // (s_nop: a store of more than 8 bytes reads its data registers for a few cycles after issue, and the hazard recogniser does
  //  not look into inline assembly -- without it the next VALU write to `v` changed what lanes 8-15 / 24-31 of each half stored)
  asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 offset:%3\n\ts_nop 1" ::"v"(voff), "v"(v), "s"(sbase), "n"(OFF) : "memory");
}
// wait until at most KEEP of this wave's vector-memory operations are outstanding; the registers are operands so that no use of
// them is scheduled in front of the wait
template <int KEEP>
DEV void wait_vm(f32x4 (&r)[4]) {
  asm volatile("s_waitcnt vmcnt(%[n])" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : [n] "n"(KEEP) : "memory");
}
template <int KEEP>
DEV void wait_vm_a16(f32x4* r) {
  asm volatile("s_waitcnt vmcnt(%[n])"
               : "+a"(r[0]), "+a"(r[1]), "+a"(r[2]), "+a"(r[3]), "+a"(r[4]), "+a"(r[5]), "+a"(r[6]), "+a"(r[7]), "+a"(r[8]), "+a"(r[9]),
                 "+a"(r[10]), "+a"(r[11]), "+a"(r[12]), "+a"(r[13]), "+a"(r[14]), "+a"(r[15])
               : [n] "n"(KEEP)
               : "memory");
}
template <int KEEP>
DEV void wait_vm1(int& r) {
  asm volatile("s_waitcnt vmcnt(%[n])" : "+v"(r) : [n] "n"(KEEP) : "memory");
}

template <class F, int... I>
DEV void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N_, class F>
DEV void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N_>{});
}

constexpr int NOUT = 128;

#ifdef DGRAD_T_STAMPS      // diagnostic build (tools/microbench/dgrad_t_check.hip): per-wave cycle sums of the phases of a tile
__device__ unsigned long long g_dgt_stamps[8];
DEV unsigned long long dgt_now() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
#define DGT_DECL unsigned long long dgt_t[6] = {0, 0, 0, 0, 0, 0}; unsigned long long dgt_last = dgt_now();
#define DGT_MARK(i) { const unsigned long long n_ = dgt_now(); dgt_t[i] += n_ - dgt_last; dgt_last = n_; }
#define DGT_KEEP(x) asm volatile("" ::"v"(x));
#define DGT_END                                                                            \
  if ((threadIdx.x & 63) == 0) {                                                           \
    for (int i_ = 0; i_ < 6; ++i_) atomicAdd(&g_dgt_stamps[i_], dgt_t[i_]);               \
    atomicAdd(&g_dgt_stamps[6], 1ull);                                                     \
  }
#else
#define DGT_DECL
#define DGT_MARK(i)
#define DGT_KEEP(x)
#define DGT_END
#endif

template <int KIN>
struct DgradTShape {
  static constexpr int LDA = KIN + 4;            // floats per staged row (+16 bytes: conflict-free ds_read_b128 fragments)
  static constexpr int BUF = 32 * LDA;           // floats per row buffer
  static constexpr int RPR = (KIN * 4 + 1023) / 1024;   // requests per row: 1 KiB each, the last one 512 bytes when KIN = 384
  static constexpr int NREQ = 8 * RPR;           // requests per wave and tile (rows 8 w .. 8 w + 7)
  static constexpr int CH = KIN / 8;             // k-chunks (one ds_read_b128 fragment + 4 MFMAs each)
  static constexpr int NB = CH / 8;              // fragment batches
  static constexpr size_t lds_bytes() { return sizeof(float) * (4 + 2 * (size_t)BUF); }
};

// One tile = 32 tokens.  Vector-memory operations of a wave inside tile i, in issue order (every wait below is derived from it):
//     [ticket atomic, one lane]  |  addend(i) x 4  |  rows(i + 1) x NREQ, from the start of the MFMA block   [a next tile exists: HN]
//                                |  stores(i - 1) x 4, behind them                                         [a previous tile: HP]
//     -- end of the block: addend(i) must be in      -> at most (HN ? NREQ : 0) + (HP ? 4 : 0) younger operations outstanding
//     -- end of the tile:  rows(i + 1) must be in    -> at most (HP ? 4 : 0)
// The results of a tile stay in 16 registers and leave one store at a time between the MFMAs of the NEXT tile (four stores back
// to back cost 600 of a tile's 18.8 k cycles: phase stamps, tools/microbench/dgrad_t_check.hip); the last tile stores at once.
// Tokens beyond M: their lanes carry copies of token M - 1 through the same arithmetic (rows and addend are clamped, read before
// any lane of the tile stores) and store the same values to the same place once more -- no masked stores, exact counts.
template <int KIN>
__global__ __launch_bounds__(256) void dgrad_t_kernel(const float* __restrict__ A, int lda, const float* __restrict__ W,
                                                      const float* addend, float* out, int64_t M, int ntiles, unsigned* queue) {
  using Sh = DgradTShape<KIN>;
  constexpr int LDA = Sh::LDA, BUF = Sh::BUF, RPR = Sh::RPR, NREQ = Sh::NREQ, CH = Sh::CH, NB = Sh::NB;
  static_assert(2 * (NREQ + 4) <= CH, "one request, then one store, behind every second k-chunk");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int* s_next = reinterpret_cast<int*>(smem);      // [2] tile tickets
  float* As = smem + 4;                            // [2][32][LDA]
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, c = lane & 31, hh = lane >> 5;
  const bool dyn = queue != nullptr;
  int ticket_ahead = 0;
  if (dyn) {
    if (tid == 0) {
      s_next[0] = (int)atomicAdd(queue, 1u);
      ticket_ahead = (int)atomicAdd(queue, 1u);
    }
  } else if (tid == 0) {
    s_next[0] = (int)blockIdx.x;
  }
  __syncthreads();
  int tile = __builtin_amdgcn_readfirstlane(s_next[0]);
  if (tile >= ntiles) return;

  const uint32_t lane16 = (uint32_t)lane * 16u;
  const uint32_t as_lds = lds_addr(As);
  const char* const Abytes = reinterpret_cast<const char*>(A);
  const int64_t row_bytes = (int64_t)lda * 4;
  // request r of this wave for tile t into buffer b: row 8 w + r / RPR, 1-KiB part r % RPR (rows beyond M repeat the last one)
  auto issue_req = [&](int t, int b, int r) {
    const int row = 8 * w + r / RPR, part = r % RPR;
    int64_t grow = (int64_t)t * 32 + row;
    grow = grow < M ? grow : M - 1;
    const char* src = Abytes + grow * row_bytes + part * 1024;
    const uint32_t dst = as_lds + (uint32_t)((b * BUF + row * LDA) * 4 + part * 1024);
    if ((part + 1) * 1024 <= KIN * 4) dma_part<false>(src, lane16, dst);
    else dma_part<true>(src, lane16, dst);
  };
#pragma unroll
  for (int r = 0; r < NREQ; ++r) issue_req(tile, 0, r);

  // W^T fragments (A operand): lane (c, hh) holds W[8 m + 4 hh + t][32 w + c], t = 0..3, once per workgroup, from the fragment-order
  // copy (dgrad_t_pack_launch: one 16-byte load per k-chunk, a wave instruction = 1 KiB contiguous; straight from the row-major
  // weight it is four scalar loads per k-chunk: 45 k instead of ~25 k cycles of prologue).  Loaded straight into the AGPR half of
  // the register file, where the MFMAs read them as they are (through ordinary loads all of W would sit in VGPRs first).
  f32x4 wf4[CH];
  {
    const char* wb = reinterpret_cast<const char*>(W + (size_t)w * CH * 256);
    static_for<CH / 4>([&](auto MQ) {
      constexpr int mq = decltype(MQ)::value;
      ldg4_uncounted_a<0>(wf4[4 * mq + 0], wb + mq * 4096, lane16);
      ldg4_uncounted_a<1024>(wf4[4 * mq + 1], wb + mq * 4096, lane16);
      ldg4_uncounted_a<2048>(wf4[4 * mq + 2], wb + mq * 4096, lane16);
      ldg4_uncounted_a<3072>(wf4[4 * mq + 3], wb + mq * 4096, lane16);
    });
  }
  DGT_DECL
  // the first tile's rows (this wave's requests) and the weights are in; every AGPR quad is an operand so that no MFMA is
  // scheduled in front of the wait (an asm statement takes at most 30 operands)
  static_for<CH / 16>([&](auto Q) { wait_vm_a16<0>(wf4 + 16 * decltype(Q)::value); });
  DGT_MARK(0)

  int buf = 0;
  f32x4 res[4];              // results of the previous tile, on their way out
  char* pbase = nullptr;     // ... their tile's base in `out` and this lane's offset in it
  uint32_t poff = 0;
  auto store_piece = [&](char* base, uint32_t off, int j) {
    switch (j) {
      case 0: stg4_uncounted<0>(base, off, res[0]); break;
      case 1: stg4_uncounted<32>(base, off, res[1]); break;
      case 2: stg4_uncounted<64>(base, off, res[2]); break;
      default: stg4_uncounted<96>(base, off, res[3]); break;
    }
  };
  auto body = [&](auto HAS_NEXT, auto HAS_PREV, int next) {
    constexpr bool HN = decltype(HAS_NEXT)::value, HP = decltype(HAS_PREV)::value;
    const int64_t tok0 = (int64_t)tile * 32;
    const int last = (int)(M - 1 - tok0 < 31 ? M - 1 - tok0 : 31);      // wave-uniform
    const uint32_t eoff = (uint32_t)(((c < last ? c : last) * NOUT + 32 * w + 4 * hh) * 4);
    const char* const abase = reinterpret_cast<const char*>(addend) + tok0 * (NOUT * 4);
    char* const obase = reinterpret_cast<char*>(out) + tok0 * (NOUT * 4);
    // addend fragments of this tile: addend[token c][32 w + 8 j + 4 hh ..], consumed behind the MFMA block
    f32x4 ad[4];
    ldg4_uncounted<0>(ad[0], abase, eoff);
    ldg4_uncounted<32>(ad[1], abase, eoff);
    ldg4_uncounted<64>(ad[2], abase, eoff);
    ldg4_uncounted<96>(ad[3], abase, eoff);

    f32x16 a0 = zero16(), a1 = zero16();
    const float* arow = As + buf * BUF + c * LDA + 4 * hh;
    float4 af[2][8];
#pragma unroll
    for (int m = 0; m < 8; ++m) af[0][m] = *reinterpret_cast<const float4*>(arow + 8 * m);
    static_for<NB>([&](auto B_) {
      constexpr int b = decltype(B_)::value;
      if constexpr (b + 1 < NB) {
#pragma unroll
        for (int m = 0; m < 8; ++m) af[(b + 1) & 1][m] = *reinterpret_cast<const float4*>(arow + 8 * (8 * (b + 1) + m));
      }
#pragma unroll
      for (int m = 0; m < 8; ++m) {                   // out^T = W^T A^T: two independent chains
        const int ch = 8 * b + m;
        a0 = mfma32(wf4[ch][0], af[b & 1][m].x, a0);
        a1 = mfma32(wf4[ch][1], af[b & 1][m].y, a1);
        a0 = mfma32(wf4[ch][2], af[b & 1][m].z, a0);
        a1 = mfma32(wf4[ch][3], af[b & 1][m].w, a1);
        // the next tile's requests behind every second k-chunk from the start of the block (the last one then has a third of a
        // block or more, >= 4 k cycles, to land), behind them the previous tile's stores, same spacing.  (MFMAs are scheduled across a volatile
        // asm statement otherwise: the barriers pin each place.)
        if (ch % 2 == 1 && ch / 2 < NREQ) {
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (HN) {
            issue_req(next, buf ^ 1, ch / 2);
            __builtin_amdgcn_sched_barrier(0);
          }
        } else if (ch % 2 == 1 && ch / 2 < NREQ + 4) {
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (HP) {
            store_piece(pbase, poff, ch / 2 - NREQ);
            __builtin_amdgcn_sched_barrier(0);
          }
        } else if (ch % 4 == 3) {
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    });
    DGT_KEEP(a0[15]) DGT_KEEP(a1[15])
    DGT_MARK(2)
    wait_vm<(HN ? NREQ : 0) + (HP ? 4 : 0)>(ad);      // the addend is in (the header of this kernel lists what is younger)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      res[j][0] = (a0[4 * j + 0] + a1[4 * j + 0]) + ad[j][0];
      res[j][1] = (a0[4 * j + 1] + a1[4 * j + 1]) + ad[j][1];
      res[j][2] = (a0[4 * j + 2] + a1[4 * j + 2]) + ad[j][2];
      res[j][3] = (a0[4 * j + 3] + a1[4 * j + 3]) + ad[j][3];
    }
    pbase = obase;
    poff = eoff;
    if constexpr (!HN) {
#pragma unroll
      for (int j = 0; j < 4; ++j) store_piece(pbase, poff, j);
    }
    DGT_MARK(3)
    // the next tile's rows (this wave's requests) are in; the ticket requested in front of the addend is older still
    if constexpr (HN) wait_vm1<HP ? 4 : 0>(ticket_ahead);
    DGT_MARK(4)
  };

  bool first = true;
  while (true) {
    if (dyn && tid == 0) s_next[buf ^ 1] = ticket_ahead;      // the ticket AFTER the next one, requested a tile ago
    __syncthreads();      // every wave's rows of `tile` are in LDS; everyone is through with the other buffer
    DGT_MARK(1)
    const int next = dyn ? __builtin_amdgcn_readfirstlane(s_next[buf ^ 1]) : tile + (int)gridDim.x;
    if (dyn && tid == 0) {
      // uncounted like the rest (a counted atomic would make the compiler wait for everything, stores included, where the
      // ticket is published); the wait at the end of the tile covers it
      asm volatile("s_nop 4\n\tglobal_atomic_add %0, %1, %2, %3 sc0" : "=v"(ticket_ahead) : "v"(0u), "v"(1u), "s"(queue) : "memory");
    }
    if (next < ntiles) {
      if (first) body(std::true_type{}, std::false_type{}, next);
      else body(std::true_type{}, std::true_type{}, next);
      first = false;
      tile = next;
      buf ^= 1;
    } else {
      if (first) body(std::false_type{}, std::false_type{}, next);
      else body(std::false_type{}, std::true_type{}, next);
      break;
    }
  }
  DGT_END
}

// [KIN][128] row-major -> the kernel's fragment order [wave 4][k-chunk KIN / 8][lane 64][4]; blockIdx.y = which matrix
// (null sources -- a path without a reverse direction -- are skipped)
struct DgradTPackArgs {
  const float* src[DGRAD_PACK_MAX];
  long long dst_off[DGRAD_PACK_MAX];      // floats, relative to dst
  int kin;
};
__global__ __launch_bounds__(256) void dgrad_t_pack_kernel(DgradTPackArgs a, float* __restrict__ dst) {
  const float* W = a.src[blockIdx.y];
  if (W == nullptr) return;
  const int ch_n = a.kin / 8, n4 = 4 * ch_n * 64;
  float4* out = reinterpret_cast<float4*>(dst + a.dst_off[blockIdx.y]);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) {
    const int lane = i & 63, m = (i >> 6) % ch_n, w = (i >> 6) / ch_n;
    const float* wp = W + (size_t)(8 * m + 4 * (lane >> 5)) * NOUT + 32 * w + (lane & 31);
    out[i] = make_float4(wp[0], wp[NOUT], wp[2 * NOUT], wp[3 * NOUT]);
  }
}

}  // namespace

#ifdef DGRAD_T_STAMPS
int dgrad_t_debug_stamps(unsigned long long* out, int reset) {
  if (reset) {
    unsigned long long z[8] = {0};
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_dgt_stamps), z, sizeof(z));
  }
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dgt_stamps), 8 * sizeof(unsigned long long));
}
#endif

template <int KIN>
static int dgrad_t_launch_k(void* stream, const DgradTArgs& a, int num_cus) {
  using Sh = DgradTShape<KIN>;
  auto kern = dgrad_t_kernel<KIN>;
  const int dev = current_hip_device();
  static PerDeviceOnce ready;
  if (!ready.done(dev)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)Sh::lds_bytes());
    if (e != hipSuccess) return (int)e;
    ready.set(dev);
  }
  const int ntiles = (int)((a.M + 31) / 32);
  const int grid = ntiles < num_cus ? ntiles : num_cus;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), Sh::lds_bytes(), static_cast<hipStream_t>(stream), a.A, a.lda, a.W,
                     a.addend, a.out, a.M, ntiles, a.queue);
  return (int)hipGetLastError();
}

int dgrad_t_launch(void* stream, const DgradTArgs& a, int num_cus) {
  if ((a.kin != 512 && a.kin != 384) || a.M < 1 || a.lda < a.kin || (a.lda & 3) || !a.A || !a.W || !a.addend || !a.out)
    return (int)hipErrorInvalidValue;
  if ((a.M + 31) / 32 > 0x7fffffff / 2 || ((uintptr_t)a.A & 15) || ((uintptr_t)a.addend & 15) || ((uintptr_t)a.out & 15))
    return (int)hipErrorInvalidValue;
  return a.kin == 512 ? dgrad_t_launch_k<512>(stream, a, num_cus) : dgrad_t_launch_k<384>(stream, a, num_cus);
}

int dgrad_t_pack_launch(void* stream, const float* const* srcs, const long long* dst_off, int n, int kin, float* dst) {
  if ((kin != 512 && kin != 384) || n < 1 || n > DGRAD_PACK_MAX || !srcs || !dst_off || !dst) return (int)hipErrorInvalidValue;
  DgradTPackArgs pa;
  for (int i = 0; i < DGRAD_PACK_MAX; ++i) {
    pa.src[i] = i < n ? srcs[i] : nullptr;
    pa.dst_off[i] = i < n ? dst_off[i] : 0;
  }
  pa.kin = kin;
  hipLaunchKernelGGL(dgrad_t_pack_kernel, dim3(16, n), dim3(256), 0, static_cast<hipStream_t>(stream), pa, dst);
  return (int)hipGetLastError();
}
