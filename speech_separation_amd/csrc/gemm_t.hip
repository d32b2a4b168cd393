// gemm_t.hip -- a K = 128 forward projection with a wide output, transposed and weights-stationary like dgrad_t.hip (round 5):
//
//     out[M][NOUT] = A[M][128] W[NOUT][128]^T + bias           NOUT = 384: qkv = x W_in^T + b_in, the in-projection of
//                                                                nn.MultiheadAttention (dptn.py:16-21, 46) in the TRAINING forward
//
// (inference forms it inside the fused attention block).  It was a gemm_ws.h launch at 0.65 MFMA-busy (388 k cycles per launch,
// profiles/r05_train_mfma_utilisation.txt): the A tile staged through registers, the C tile through LDS with a second barrier and a
// 12-pass row epilogue.  Here: A tile by hand-counted LDS-DMA (512-byte rows: half-EXEC requests), wave w = output columns
// [NOUT/4 w, NOUT/4 (w + 1)) in NT = NOUT / 128 blocks of 32 (three accumulation chains alternate), the accumulators START from the
// bias (a per-lane constant vector read from LDS), so a tile's result is 4 NT 16-byte fragment stores and nothing else; the
// previous tile's stores and the next tile's requests sit between the MFMAs; one barrier per 32-token tile.
// Weights: the fragment-order copy gemm_pack_rows_kernel makes (gemm_ws.h: [column block 32][k-chunk 16][lane 64][4]).
#include <hip/hip_runtime.h>

#include <type_traits>
#include <utility>

#include "common.h"
#include "gemm_t.h"

namespace {

DEV uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p; }
// (s_nop 4 / s_nop 1: the wait states of dgrad_r.hip -- a VALU-written scalar operand in front of the statement, a 16-byte store's data)
DEV void dma_half(const void* sbase, uint32_t voff, uint32_t lds_base) {      // lanes 0..31: 512 bytes -> LDS lds_base + 16 L
  uint32_t saved;
  asm volatile("s_mov_b32 m0, %1\n\ts_mov_b32 %0, exec_hi\n\ts_mov_b32 exec_hi, 0\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 exec_hi, %0"
               : "=&s"(saved)
               : "s"(lds_base), "v"(voff), "s"(sbase)
               : "memory", "m0");
}
template <int OFF>
DEV void ldg4_uncounted_a(f32x4& dst, const void* sbase, uint32_t voff) {
  asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:%3" : "=a"(dst) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}
template <int OFF>
DEV void stg4_uncounted(void* sbase, uint32_t voff, f32x4 v) {
  asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 offset:%3\n\ts_nop 1" ::"v"(voff), "v"(v), "s"(sbase), "n"(OFF) : "memory");
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

constexpr int KIN = 128, CH = KIN / 8, LDA = KIN + 4;

// Vector-memory operations of a wave inside tile i, in issue order: [ticket atomic, one lane] | rows(i + 1) x 8 [HN] | stores(i - 1)
// x NS [HP] (two slots per k-chunk: the requests first, the stores behind them) -- end of the tile: rows(i + 1) must be in -> at most
// (HP ? NS : 0) younger operations outstanding.
template <int NOUT>
__global__ __launch_bounds__(256) void gemm_t_kernel(const float* __restrict__ A, const float* __restrict__ Wp, const float* __restrict__ bias,
                                                     float* __restrict__ out, int64_t M, int ntiles, unsigned* queue) {
  constexpr int NT = NOUT / 128, NS = 4 * NT, WCOLS = NOUT / 4;
  static_assert(8 + NS <= 2 * CH, "two request slots per k-chunk");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int* s_next = reinterpret_cast<int*>(smem);      // [2] tile tickets
  float* Bs = smem + 4;                            // [NOUT] bias
  float* As = Bs + NOUT;                           // [2][32][LDA]
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
  for (int i = tid; i < NOUT; i += 256) Bs[i] = bias[i];
  __syncthreads();
  int tile = __builtin_amdgcn_readfirstlane(s_next[0]);
  if (tile >= ntiles) return;

  const uint32_t lane16 = (uint32_t)lane * 16u;
  const uint32_t as_lds = lds_addr(As);
  const char* const Ab = reinterpret_cast<const char*>(A);
  auto issue_req = [&](int t, int b, int r) {      // row 8 w + r of tile t -> buffer b (rows beyond M repeat the last one)
    const int row = 8 * w + r;
    int64_t grow = (int64_t)t * 32 + row;
    grow = grow < M ? grow : M - 1;
    dma_half(Ab + grow * (KIN * 4), lane16, as_lds + (uint32_t)((b * 32 + row) * LDA * 4));
  };
#pragma unroll
  for (int r = 0; r < 8; ++r) issue_req(tile, 0, r);

  // W fragments (A operand): wf4[nt * 16 + m] of lane (c, hh) = W[WCOLS w + 32 nt + c][8 m + 4 hh .. + 3], straight into AGPRs
  f32x4 wf4[NT * CH];
  {
    const char* wb = reinterpret_cast<const char*>(Wp + (size_t)w * NT * CH * 256);
    static_for<NT * CH / 4>([&](auto MQ) {
      constexpr int mq = decltype(MQ)::value;
      ldg4_uncounted_a<0>(wf4[4 * mq + 0], wb + mq * 4096, lane16);
      ldg4_uncounted_a<1024>(wf4[4 * mq + 1], wb + mq * 4096, lane16);
      ldg4_uncounted_a<2048>(wf4[4 * mq + 2], wb + mq * 4096, lane16);
      ldg4_uncounted_a<3072>(wf4[4 * mq + 3], wb + mq * 4096, lane16);
    });
  }
  static_for<NT * CH / 16>([&](auto Q) { wait_vm_a16<0>(wf4 + 16 * decltype(Q)::value); });   // weights and the first tile's rows are in

  int buf = 0;
  f32x4 res[NS];             // results of the previous tile, on their way out
  char* pbase = nullptr;
  uint32_t poff = 0;
  auto store_piece = [&](char* base, uint32_t off, auto J) {      // piece j = 4 nt + jj: columns WCOLS w + 32 nt + 8 jj + 4 hh ..
    constexpr int j = decltype(J)::value;
    stg4_uncounted<(j / 4) * 128 + (j % 4) * 32>(base, off, res[j]);
  };
  auto body = [&](auto HAS_NEXT, auto HAS_PREV, int next) {
    constexpr bool HN = decltype(HAS_NEXT)::value, HP = decltype(HAS_PREV)::value;
    const int64_t tok0 = (int64_t)tile * 32;
    const int last = (int)(M - 1 - tok0 < 31 ? M - 1 - tok0 : 31);      // wave-uniform
    const uint32_t eoff = (uint32_t)(((c < last ? c : last) * NOUT + WCOLS * w + 4 * hh) * 4);
    char* const obase = reinterpret_cast<char*>(out) + tok0 * (NOUT * 4);
    // accumulators start from the bias: register 4 jj + t of block nt belongs to column WCOLS w + 32 nt + 8 jj + 4 hh + t
    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const float4 bv = *reinterpret_cast<const float4*>(Bs + WCOLS * w + 32 * nt + 8 * jj + 4 * hh);
        acc[nt][4 * jj] = bv.x; acc[nt][4 * jj + 1] = bv.y; acc[nt][4 * jj + 2] = bv.z; acc[nt][4 * jj + 3] = bv.w;
      }
    const float* arow = As + (buf * 32 + c) * LDA + 4 * hh;
    static_for<2>([&](auto B_) {
      constexpr int b = decltype(B_)::value;
      float4 af[8];
#pragma unroll
      for (int m = 0; m < 8; ++m) af[m] = *reinterpret_cast<const float4*>(arow + 8 * (8 * b + m));
      static_for<8>([&](auto M_) {
        constexpr int m = decltype(M_)::value, ch = 8 * b + m;
        static_for<2>([&](auto H_) {      // half a k-chunk: 2 NT MFMAs, the NT chains alternate
          constexpr int h2 = decltype(H_)::value, slot = 2 * ch + h2;
          const float a0 = h2 == 0 ? af[m].x : af[m].z, a1 = h2 == 0 ? af[m].y : af[m].w;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma32(wf4[nt * CH + ch][2 * h2], a0, acc[nt]);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma32(wf4[nt * CH + ch][2 * h2 + 1], a1, acc[nt]);
          __builtin_amdgcn_sched_barrier(0);      // (MFMAs are scheduled across a volatile asm statement otherwise: pin the place)
          if constexpr (slot < 8) {
            if constexpr (HN) {
              issue_req(next, buf ^ 1, slot);
              __builtin_amdgcn_sched_barrier(0);
            }
          } else if constexpr (slot < 8 + NS) {
            if constexpr (HP) {
              store_piece(pbase, poff, std::integral_constant<int, slot - 8>{});
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        });
      });
    });
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
        res[4 * nt + jj] = (f32x4){acc[nt][4 * jj], acc[nt][4 * jj + 1], acc[nt][4 * jj + 2], acc[nt][4 * jj + 3]};
    pbase = obase;
    poff = eoff;
    if constexpr (!HN) static_for<NS>([&](auto J) { store_piece(pbase, poff, J); });
    if constexpr (HN) wait_vm1<HP ? NS : 0>(ticket_ahead);      // the next tile's rows (this wave's requests) are in
  };

  bool first = true;
  while (true) {
    if (dyn && tid == 0) s_next[buf ^ 1] = ticket_ahead;      // the ticket AFTER the next one, requested a tile ago
    __syncthreads();      // every wave's rows of `tile` are in LDS; everyone is through with the other buffer
    const int next = dyn ? __builtin_amdgcn_readfirstlane(s_next[buf ^ 1]) : tile + (int)gridDim.x;
    if (dyn && tid == 0)
      asm volatile("s_nop 4\n\tglobal_atomic_add %0, %1, %2, %3 sc0" : "=v"(ticket_ahead) : "v"(0u), "v"(1u), "s"(queue) : "memory");
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
}

}  // namespace

int gemm_t_launch(void* stream, const GemmTArgs& a, int num_cus) {
  if (a.nout != 384 || a.M < 1 || !a.A || !a.Wpacked || !a.bias || !a.out) return (int)hipErrorInvalidValue;
  if ((a.M + 31) / 32 > 0x7fffffff / 2 || ((uintptr_t)a.A & 15) || ((uintptr_t)a.out & 15) || ((uintptr_t)a.Wpacked & 15)) return (int)hipErrorInvalidValue;
  auto kern = gemm_t_kernel<384>;
  const size_t lds = sizeof(float) * (4 + 384 + 2 * 32 * (size_t)LDA);
  const int dev = current_hip_device();
  static PerDeviceOnce ready;
  if (!ready.done(dev)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    ready.set(dev);
  }
  const int ntiles = (int)((a.M + 31) / 32);
  const int grid = ntiles < num_cus ? ntiles : num_cus;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, static_cast<hipStream_t>(stream), a.A, a.Wpacked, a.bias, a.out, a.M, ntiles, a.queue);
  return (int)hipGetLastError();
}
