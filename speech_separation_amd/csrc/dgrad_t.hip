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
// One LDS-DMA request: lane L's 16 bytes at (wave-uniform base + voff) land at LDS byte address lds_base + 16 L (attn_block2.hip)
DEV void dma_1k(const void* sbase, uint32_t voff, uint32_t lds_base) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_base), "v"(voff), "s"(sbase) : "memory", "m0");
}
// fragment load / store the compiler does not count: 16 bytes at base + voff + OFF
template <int OFF>
DEV void ldg4_uncounted(f32x4& dst, const void* sbase, uint32_t voff) {
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}
template <int OFF>
DEV void stg4_uncounted(void* sbase, uint32_t voff, f32x4 v) {
  // (s_nop: a store of more than 8 bytes reads its data registers for a few cycles after issue, and the hazard recogniser does
  //  not look into inline assembly -- without it the next VALU write to `v` changed what lanes 8-15 / 24-31 of each half stored)
  asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3\n\ts_nop 1" ::"v"(voff), "v"(v), "s"(sbase), "n"(OFF) : "memory");
}
// wait until at most KEEP of this wave's vector-memory operations are outstanding; the registers are operands so that no use of
// them is scheduled in front of the wait
template <int KEEP>
DEV void wait_vm(f32x4 (&r)[4]) {
  asm volatile("s_waitcnt vmcnt(%[n])" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : [n] "n"(KEEP) : "memory");
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

template <int KIN>
struct DgradTShape {
  static constexpr int LDA = KIN + 4;            // floats per staged row (+16 bytes: conflict-free ds_read_b128 fragments)
  static constexpr int BUF = 32 * LDA;           // floats per row buffer
  static constexpr int RPR = KIN / 256;          // 1-KiB requests per row
  static constexpr int NREQ = 8 * RPR;           // requests per wave and tile (rows 8 w .. 8 w + 7)
  static constexpr int CH = KIN / 8;             // k-chunks (one ds_read_b128 fragment + 4 MFMAs each)
  static constexpr int NB = CH / 8;              // fragment batches
  static constexpr size_t lds_bytes() { return sizeof(float) * (4 + 2 * (size_t)BUF); }
};

template <int KIN>
__global__ __launch_bounds__(256) void dgrad_t_kernel(const float* __restrict__ A, int lda, const float* __restrict__ W,
                                                      const float* addend, float* out, int64_t M, int ntiles, unsigned* queue) {
  using Sh = DgradTShape<KIN>;
  constexpr int LDA = Sh::LDA, BUF = Sh::BUF, RPR = Sh::RPR, NREQ = Sh::NREQ, CH = Sh::CH, NB = Sh::NB;
  static_assert(CH / 2 >= NREQ, "one request behind every second k-chunk of the first half");
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
    dma_1k(Abytes + grow * row_bytes + part * 1024, lane16, as_lds + (uint32_t)((b * BUF + row * LDA) * 4 + part * 1024));
  };
#pragma unroll
  for (int r = 0; r < NREQ; ++r) issue_req(tile, 0, r);

  // W^T fragments (A operand): lane (c, hh) holds W[8 m + 4 hh + t][32 w + c], once per workgroup.  A wave instruction reads two
  // 128-byte row segments (coalesced as it is: no fragment-order copy needed, unlike the forward form of gemm_ws.h); scalar row
  // base + one 32-bit lane offset, so the 256 loads carry no vector address arithmetic.
  float wf[KIN / 2];
  {
    const unsigned lane_off = (unsigned)(4 * hh * NOUT + 32 * w + c);
#pragma unroll
    for (int m = 0; m < CH; ++m)
#pragma unroll
      for (int t = 0; t < 4; ++t) wf[4 * m + t] = (W + (8 * m + t) * NOUT)[lane_off];
  }
  // ... parked in the AGPR half of the register file, where the MFMAs read them as they are (left to itself the allocator fills
  // the 256 architectural registers first and copies every fragment that did not fit back through a VGPR before its MFMA)
#pragma unroll
  for (int i = 0; i < KIN / 2; ++i) asm volatile("" : "+a"(wf[i]));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the first tile's rows (this wave's requests)

  int buf = 0;
  auto body = [&](auto HAS_NEXT, int next) {
    constexpr bool HN = decltype(HAS_NEXT)::value;
    const int64_t tok0 = (int64_t)tile * 32;
    const int last = (int)(M - 1 - tok0 < 31 ? M - 1 - tok0 : 31);      // wave-uniform
    const bool ok = c <= last;
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
        a0 = mfma32(wf[4 * ch + 0], af[b & 1][m].x, a0);
        a1 = mfma32(wf[4 * ch + 1], af[b & 1][m].y, a1);
        a0 = mfma32(wf[4 * ch + 2], af[b & 1][m].z, a0);
        a1 = mfma32(wf[4 * ch + 3], af[b & 1][m].w, a1);
        // the next tile's requests behind every second k-chunk of the FIRST HALF of the block: the last one then has half a block
        // (8 k cycles) to land before the wait at the end of the tile
        if (ch % 2 == 1 && ch / 2 < NREQ) {           // (MFMAs are scheduled across a volatile asm statement otherwise: pin the place)
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (HN) {
            issue_req(next, buf ^ 1, ch / 2);
            __builtin_amdgcn_sched_barrier(0);
          }
        } else if (ch % 4 == 3) {
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    });
    // the addend is in: behind its four loads only this tile's NREQ row requests went out (none without a next tile)
    wait_vm<HN ? NREQ : 0>(ad);
    if (ok) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4 v;
        v[0] = (a0[4 * j + 0] + a1[4 * j + 0]) + ad[j][0];
        v[1] = (a0[4 * j + 1] + a1[4 * j + 1]) + ad[j][1];
        v[2] = (a0[4 * j + 2] + a1[4 * j + 2]) + ad[j][2];
        v[3] = (a0[4 * j + 3] + a1[4 * j + 3]) + ad[j][3];
        switch (j) {
          case 0: stg4_uncounted<0>(obase, eoff, v); break;
          case 1: stg4_uncounted<32>(obase, eoff, v); break;
          case 2: stg4_uncounted<64>(obase, eoff, v); break;
          default: stg4_uncounted<96>(obase, eoff, v); break;
        }
      }
    }
    // the next tile's rows (this wave's requests) are in: behind them only the four stores above went out (every wave has at
    // least one token of the tile, so the stores are always issued); the ticket requested in front of the addend is older still
    if constexpr (HN) wait_vm1<4>(ticket_ahead);
  };

  while (true) {
    if (dyn && tid == 0) s_next[buf ^ 1] = ticket_ahead;      // the ticket AFTER the next one, requested a tile ago
    __syncthreads();      // every wave's rows of `tile` are in LDS; everyone is through with the other buffer
    const int next = dyn ? __builtin_amdgcn_readfirstlane(s_next[buf ^ 1]) : tile + (int)gridDim.x;
    if (dyn && tid == 0) {
      // uncounted like the rest (a counted atomic would make the compiler wait for everything, stores included, where the
      // ticket is published); the wait at the end of the tile covers it
      asm volatile("global_atomic_add %0, %1, %2, %3 sc0" : "=v"(ticket_ahead) : "v"(0u), "v"(1u), "s"(queue) : "memory");
    }
    if (next < ntiles) {
      body(std::true_type{}, next);
      tile = next;
      buf ^= 1;
    } else {
      body(std::false_type{}, next);
      break;
    }
  }
}

}  // namespace

int dgrad_t_launch(void* stream, const DgradTArgs& a, int num_cus) {
  if (a.kin != 512 || a.M < 1 || a.lda < a.kin || (a.lda & 3) || !a.A || !a.W || !a.addend || !a.out) return (int)hipErrorInvalidValue;
  if ((a.M + 31) / 32 > 0x7fffffff / 2) return (int)hipErrorInvalidValue;
  using Sh = DgradTShape<512>;
  auto kern = dgrad_t_kernel<512>;
  const int dev = current_hip_device();
  static PerDeviceOnce ready;
  if (!ready.done(dev)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)Sh::lds_bytes());
    if (e != hipSuccess) return (int)e;
    ready.set(dev);
  }
  const int ntiles = (int)((a.M + 31) / 32);
  const int grid = ntiles < num_cus ? ntiles : num_cus;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), Sh::lds_bytes(), static_cast<hipStream_t>(stream), a.A, a.lda, a.W, a.addend, a.out,
                     a.M, ntiles, a.queue);
  return (int)hipGetLastError();
}
