// dgrad_r.hip -- the K = 128 data gradients of a training step WITH their layer's weight gradient riding on the staged tiles,
// transposed and weights-stationary like dgrad_t.hip (round 5):
//
//     out[M][NOUT] = (A[M][128] W[128][NOUT]) (.) gate        dW[128][NOUT] = sum_tokens A^T X        db[128] = column sums of A
//
//   NOUT = 128, no gate:   d att = dz1 W_o,  dW_o = dz1^T att,  db_o          (dptn.py:46-47: out_proj of nn.MultiheadAttention)
//   NOUT = 256, ReLU gate: d h = (dz2 W_f) masked by h > 0,  dW_f = dz2^T relu(h),  db_f      (dptn.py:50: ffn = ReLU -> Linear)
//
// These were gemm_ws.h launches with a WgradRider at 0.52 / 0.56 MFMA-busy (15.9 k / 29 k cycles per 32-token tile of 8.2 k /
// 16.4 k MFMA issue; profiles/r05_train_mfma_utilisation.txt): both tiles staged through registers (12 loads back to back, 12
// ds_write_b128), the C tile through LDS with a second barrier and an 8-pass row epilogue that fetches the gate rows again from
// memory.  Here: A and X tiles by hand-counted LDS-DMA (rows of 512 bytes by half-EXEC requests), the data gradient formed
// transposed (wave w = output columns [NOUT/4 w, NOUT/4 (w + 1)): result fragments are 16-byte stores, the ReLU gate is the
// X tile already in LDS), the previous tile's stores between the rider's MFMAs, one barrier per tile.  The rider keeps the
// engine's accumulator layout (wave w = rows [32 w, 32 w + 32) of dW, fragment-order slab per workgroup), so
// slab_reduce_frag_kernel and the column-sum rows are shared with it.
// Tokens beyond M (last tile): rows are clamped copies of token M - 1 for the data gradient (stored once more to the same place),
// and masked out of the rider's A operand (so neither dW nor db sees them).
#include <hip/hip_runtime.h>

#include <type_traits>
#include <utility>

#include "common.h"
#include "dgrad_r.h"

namespace {

DEV uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p; }
// LDS-DMA request (dgrad_t.hip): lane L's 16 bytes at (base + voff) -> LDS lds_base + 16 L; HALF: lanes 0..31 only (512 bytes).
// s_nop 4 in EVERY asm statement here that issues a vector-memory instruction with a scalar operand: the operand may have been
// written by a VALU instruction just in front of the statement (v_readlane of a spilled SGPR, v_readfirstlane), the ISA asks for
// five wait states between such a write and a VMEM instruction that reads the register, and the compiler's recogniser does not
// look into inline assembly.  Found on the hardware: this kernel has SGPR spills, and a request went out with a stale base
// (memory access fault at M = 4 097; none at M = 1 or 33 -- the hazard depends on what the allocator put in front of the statement).
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
template <int OFF>
DEV void ldg4_uncounted(f32x4& dst, const void* sbase, uint32_t voff) {
  asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}
// (s_nop: a store of more than 8 bytes reads its data registers for a few cycles after issue -- dgrad_t.hip)
template <int OFF>
DEV void stg4_uncounted(void* sbase, uint32_t voff, f32x4 v) {
  asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 offset:%3\n\ts_nop 1" ::"v"(voff), "v"(v), "s"(sbase), "n"(OFF) : "memory");
}
template <int KEEP>
DEV void wait_vm_v16(f32x4* r) {
  asm volatile("s_waitcnt vmcnt(%[n])"
               : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]), "+v"(r[9]),
                 "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
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
DEV float half_sum(float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}

constexpr int KIN = 128, CH = KIN / 8;      // 16 k-chunks

template <int NOUT>
struct DgradRShape {
  static constexpr int NT = NOUT / 128;          // 32-column blocks of the data gradient per wave
  static constexpr int CB = NOUT / 32;           // 32-column blocks of dW per wave (its 32 rows x all NOUT columns)
  static constexpr int LDA = KIN + 4;            // floats per staged A row
  static constexpr int LDX = NOUT + 4;           // floats per staged X row
  static constexpr int STAGE = 32 * (LDA + LDX); // floats per buffer: A tile, then X tile
  static constexpr int NREQ = 16;                // requests per wave and tile: rows 8 w .. 8 w + 7 of A and of X
  static constexpr int NS = 4 * NT;              // 16-byte result stores per lane and tile
  static constexpr size_t lds_bytes() { return sizeof(float) * (4 + 2 * (size_t)STAGE); }
};

// Vector-memory operations of a wave inside tile i, in issue order (the waits are derived from it, as in dgrad_t.hip):
//     [ticket atomic, one lane] | rows(i + 1) x 16 behind the 16 k-chunks of the data gradient [HN] | stores(i - 1) x NS between the
//     rider's MFMAs [HP]        -- end of the tile: rows(i + 1) must be in -> at most (HP ? NS : 0) younger operations outstanding
template <int NOUT, bool GATE>
__global__ __launch_bounds__(256) void dgrad_r_kernel(const float* __restrict__ A, const float* __restrict__ Wp, const float* __restrict__ X,
                                                      float* __restrict__ out, int64_t M, int ntiles, unsigned* queue,
                                                      float* __restrict__ slab, float* __restrict__ colslab) {
  using Sh = DgradRShape<NOUT>;
  constexpr int NT = Sh::NT, CB = Sh::CB, LDA = Sh::LDA, LDX = Sh::LDX, STAGE = Sh::STAGE, NS = Sh::NS;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int* s_next = reinterpret_cast<int*>(smem);      // [2] tile tickets
  float* St = smem + 4;                            // [2][ A [32][LDA] | X [32][LDX] ]
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

  const uint32_t lane16 = (uint32_t)lane * 16u;
  const uint32_t st_lds = lds_addr(St);
  const char* const Ab = reinterpret_cast<const char*>(A);
  const char* const Xb = reinterpret_cast<const char*>(X);
  // request r of this wave for tile t into buffer b: r < 8: A row 8 w + r (512 bytes), r >= 8: X row 8 w + r - 8 (NOUT * 4 bytes);
  // rows beyond M repeat the last one
  auto issue_req = [&](int t, int b, int r) {
    const int row = 8 * w + (r & 7);
    int64_t grow = (int64_t)t * 32 + row;
    grow = grow < M ? grow : M - 1;
    if (r < 8) dma_part<true>(Ab + grow * (KIN * 4), lane16, st_lds + (uint32_t)((b * STAGE + row * LDA) * 4));
    else if (NOUT == 256) dma_part<false>(Xb + grow * (NOUT * 4), lane16, st_lds + (uint32_t)((b * STAGE + 32 * LDA + row * LDX) * 4));
    else dma_part<true>(Xb + grow * (NOUT * 4), lane16, st_lds + (uint32_t)((b * STAGE + 32 * LDA + row * LDX) * 4));
  };
  if (tile < ntiles) {
#pragma unroll
    for (int r = 0; r < 16; ++r) issue_req(tile, 0, r);
  }

  // W^T fragments (A operand of the data gradient) from the fragment-order copy (dgrad_r_pack_launch), by loads the compiler does
  // not count.  This file is compiled WITHOUT -amdgpu-mfma-vgpr-form: the 16 NT + 16 CB accumulator registers (160 at NOUT = 256)
  // live in AGPRs, the weights (64 NT), the staged fragments and the results on their way out in VGPRs:
  // wf4[nt][m] of lane (c, hh) = W[8 m + 4 hh + t][NOUT/4 w + 32 nt + c], t = 0..3
  f32x4 wf4[NT * CH];
  {
    const char* wb = reinterpret_cast<const char*>(Wp + (size_t)w * NT * CH * 256);
    static_for<NT * CH / 4>([&](auto MQ) {
      constexpr int mq = decltype(MQ)::value;
      ldg4_uncounted<0>(wf4[4 * mq + 0], wb + mq * 4096, lane16);
      ldg4_uncounted<1024>(wf4[4 * mq + 1], wb + mq * 4096, lane16);
      ldg4_uncounted<2048>(wf4[4 * mq + 2], wb + mq * 4096, lane16);
      ldg4_uncounted<3072>(wf4[4 * mq + 3], wb + mq * 4096, lane16);
    });
  }
  static_for<NT * CH / 16>([&](auto Q) { wait_vm_v16<0>(wf4 + 16 * decltype(Q)::value); });   // weights and the first tile's rows are in

  // rider state: dW rows [32 w, 32 w + 32) x all NOUT columns, and this lane's share of the column sums of A
  f32x16 racc[CB];
#pragma unroll
  for (int j = 0; j < CB; ++j) racc[j] = zero16();
  float csum = 0.f;
  // where this lane's pieces of the partial tile and its column sum go: formed here and kept in VGPRs across the tile loop
  float* so_lane = slab + (size_t)blockIdx.x * (KIN * NOUT) + ((size_t)w * CB * 4 * 64 + lane) * 4;
  float* cs_lane = colslab + (size_t)blockIdx.x * KIN + 32 * w + c;
  asm volatile("" : "+v"(so_lane), "+v"(cs_lane));

  int buf = 0;
  f32x4 res[NS];             // results of the previous tile, on their way out
  char* pbase = nullptr;
  uint32_t poff = 0;
  auto store_piece = [&](char* base, uint32_t off, auto J) {      // piece j = 4 nt + jj: columns NOUT/4 w + 32 nt + 8 jj + 4 hh ..
    constexpr int j = decltype(J)::value;
    stg4_uncounted<(j / 4) * 128 + (j % 4) * 32>(base, off, res[j]);
  };
  auto body = [&](auto HAS_NEXT, auto HAS_PREV, int next) {
    constexpr bool HN = decltype(HAS_NEXT)::value, HP = decltype(HAS_PREV)::value;
    const int64_t tok0 = (int64_t)tile * 32;
    const int last = (int)(M - 1 - tok0 < 31 ? M - 1 - tok0 : 31);      // wave-uniform
    const uint32_t eoff = (uint32_t)(((c < last ? c : last) * NOUT + (NOUT / 4) * w + 4 * hh) * 4);
    char* const obase = reinterpret_cast<char*>(out) + tok0 * (NOUT * 4);
    const float* As = St + buf * STAGE;
    const float* Xs = As + 32 * LDA;

    // ---- data gradient, transposed: out^T = W^T A^T, NT column blocks of this wave; one request of the next tile per k-chunk ----
    // (two accumulation chains: a single one makes every MFMA wait for its predecessor's last pass -- NT = 1: even / odd k-slots,
    //  summed in the epilogue; NT = 2: the two column blocks alternate)
    f32x16 acc[2];
    acc[0] = zero16();
    acc[1] = zero16();
    const float* arow = As + c * LDA + 4 * hh;
    static_for<2>([&](auto B_) {
      constexpr int b = decltype(B_)::value;
      float4 af[8];
#pragma unroll
      for (int m = 0; m < 8; ++m) af[m] = *reinterpret_cast<const float4*>(arow + 8 * (8 * b + m));
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const int ch = 8 * b + m;
        if constexpr (NT == 1) {
          acc[0] = mfma32(wf4[ch][0], af[m].x, acc[0]);
          acc[1] = mfma32(wf4[ch][1], af[m].y, acc[1]);
          acc[0] = mfma32(wf4[ch][2], af[m].z, acc[0]);
          acc[1] = mfma32(wf4[ch][3], af[m].w, acc[1]);
        } else {
          acc[0] = mfma32(wf4[ch][0], af[m].x, acc[0]);
          acc[1] = mfma32(wf4[CH + ch][0], af[m].x, acc[1]);
          acc[0] = mfma32(wf4[ch][1], af[m].y, acc[0]);
          acc[1] = mfma32(wf4[CH + ch][1], af[m].y, acc[1]);
          acc[0] = mfma32(wf4[ch][2], af[m].z, acc[0]);
          acc[1] = mfma32(wf4[CH + ch][2], af[m].z, acc[1]);
          acc[0] = mfma32(wf4[ch][3], af[m].w, acc[0]);
          acc[1] = mfma32(wf4[CH + ch][3], af[m].w, acc[1]);
        }
        __builtin_amdgcn_sched_barrier(0);      // (MFMAs are scheduled across a volatile asm statement otherwise: pin the place)
        if constexpr (HN) {
          issue_req(next, buf ^ 1, ch);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    });

    // ---- rider: dW[32 w + k][n] += sum over the tile's tokens A[tok][32 w + k] X[tok][n]; MFMA step s takes tokens (2 s, 2 s + 1) ----
    const float* yrow = As + hh * LDA + 32 * w + c;
    const float* xrow = Xs + hh * LDX + c;
    float ra[2], rb[2][CB];
    auto fetch = [&](int s2, int p) {
      ra[p] = yrow[2 * s2 * LDA];
#pragma unroll
      for (int j = 0; j < CB; ++j) rb[p][j] = xrow[2 * s2 * LDX + 32 * j];
    };
    fetch(0, 0);
    static_for<16>([&](auto S2) {
      constexpr int s2 = decltype(S2)::value, p = s2 & 1;
      if constexpr (s2 + 1 < 16) fetch(s2 + 1, p ^ 1);
      // tokens beyond M stay out of dW and db: only a workgroup's LAST tile can be the batch's last (tile indices grow), so the tiles
      // with a successor skip the select
      const float a = (HN || 2 * s2 + hh <= last) ? ra[p] : 0.f;
      csum += a;
#pragma unroll
      for (int j = 0; j < CB; ++j) {
        float xv = rb[p][j];
        if (GATE) xv = __int_as_float(max(__float_as_int(xv), 0));      // ReLU in one instruction (negative floats are negative integers)
        racc[j] = mfma32(a, xv, racc[j]);
      }
      // (a fence per token pair: left alone the scheduler issues all 16 (1 + CB) fragment reads of the tile up front and the kernel spills)
      __builtin_amdgcn_sched_barrier(0);
      // the previous tile's stores, one behind every second token pair
      if constexpr ((s2 & 1) == 1 && s2 / 2 < NS && HP) {
        store_piece(pbase, poff, std::integral_constant<int, (s2 / 2 < NS) ? s2 / 2 : 0>{});
        __builtin_amdgcn_sched_barrier(0);
      }
    });

    // ---- this tile's results: register 4 jj + t of block nt = out[token c][NOUT/4 w + 32 nt + 8 jj + 4 hh + t], gated by the X tile ----
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        f32x4 v;
        if constexpr (NT == 1) {
          v = (f32x4){acc[0][4 * jj] + acc[1][4 * jj], acc[0][4 * jj + 1] + acc[1][4 * jj + 1], acc[0][4 * jj + 2] + acc[1][4 * jj + 2],
                      acc[0][4 * jj + 3] + acc[1][4 * jj + 3]};
        } else {
          v = (f32x4){acc[nt][4 * jj], acc[nt][4 * jj + 1], acc[nt][4 * jj + 2], acc[nt][4 * jj + 3]};
        }
        if (GATE) {
          const float4 gt = *reinterpret_cast<const float4*>(Xs + c * LDX + (NOUT / 4) * w + 32 * nt + 8 * jj + 4 * hh);
          v[0] = gt.x > 0.f ? v[0] : 0.f;
          v[1] = gt.y > 0.f ? v[1] : 0.f;
          v[2] = gt.z > 0.f ? v[2] : 0.f;
          v[3] = gt.w > 0.f ? v[3] : 0.f;
        }
        res[4 * nt + jj] = v;
      }
    pbase = obase;
    poff = eoff;
    if constexpr (!HN) static_for<NS>([&](auto J) { store_piece(pbase, poff, J); });
    // the next tile's rows (this wave's requests) are in; the ticket requested in front of them is older still
    if constexpr (HN) wait_vm1<HP ? NS : 0>(ticket_ahead);
  };

  bool first = true;
  while (tile < ntiles) {
    if (dyn && tid == 0) s_next[buf ^ 1] = ticket_ahead;      // the ticket AFTER the next one, requested a tile ago
    __syncthreads();      // every wave's rows of `tile` are in LDS; everyone is through with the other buffer
    const int next = dyn ? __builtin_amdgcn_readfirstlane(s_next[buf ^ 1]) : tile + (int)gridDim.x;
    if (dyn && tid == 0) asm volatile("s_nop 4\n\tglobal_atomic_add %0, %1, %2, %3 sc0" : "=v"(ticket_ahead) : "v"(0u), "v"(1u), "s"(queue) : "memory");
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

  // ---- this workgroup's partial dW in fragment order (gemm_ws.h rider_finish / wgrad_kernel: slab_reduce_frag_kernel<1, CB>) and its
  //      partial column sums: lane (c, hh) summed A[tok][32 w + c] over the tokens of parity hh ----
#pragma unroll
  for (int j = 0; j < CB; ++j)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4)
      *reinterpret_cast<float4*>(so_lane + (j * 4 + g4) * 256) =
          make_float4(racc[j][4 * g4], racc[j][4 * g4 + 1], racc[j][4 * g4 + 2], racc[j][4 * g4 + 3]);
  const float cs = half_sum(csum);
  if (hh == 0) *cs_lane = cs;
}

// W [128][nout] row-major -> [wave 4][nt][k-chunk 16][lane 64][4]; blockIdx.y = which matrix (null sources are skipped)
struct DgradRPackArgs {
  const float* src[DGRAD_R_PACK_MAX];
  long long dst_off[DGRAD_R_PACK_MAX];      // floats, relative to dst
  int nout;
};
__global__ __launch_bounds__(256) void dgrad_r_pack_kernel(DgradRPackArgs a, float* __restrict__ dst) {
  const float* W = a.src[blockIdx.y];
  if (W == nullptr) return;
  const int nout = a.nout, nt_n = nout / 128, n4 = 4 * nt_n * CH * 64;
  float4* out = reinterpret_cast<float4*>(dst + a.dst_off[blockIdx.y]);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) {
    const int lane = i & 63, m = (i >> 6) % CH, nt = ((i >> 6) / CH) % nt_n, w = (i >> 6) / (CH * nt_n);
    const float* wp = W + (size_t)(8 * m + 4 * (lane >> 5)) * nout + (nout / 4) * w + 32 * nt + (lane & 31);
    out[i] = make_float4(wp[0], wp[nout], wp[2 * nout], wp[3 * nout]);
  }
}

template <int NOUT, bool GATE>
int launch_k(void* stream, const DgradRArgs& a, int num_cus, int* grid_used) {
  using Sh = DgradRShape<NOUT>;
  auto kern = dgrad_r_kernel<NOUT, GATE>;
  const int dev = current_hip_device();
  static PerDeviceOnce ready;
  if (!ready.done(dev)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)Sh::lds_bytes());
    if (e != hipSuccess) return (int)e;
    ready.set(dev);
  }
  const int ntiles = (int)((a.M + 31) / 32);
  int grid = ntiles < num_cus ? ntiles : num_cus;
  if (grid > a.max_slabs) grid = a.max_slabs;
  if (grid_used) *grid_used = grid;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), Sh::lds_bytes(), static_cast<hipStream_t>(stream), a.A, a.Wpacked, a.X, a.out, a.M, ntiles,
                     a.queue, a.slab, a.colslab);
  return (int)hipGetLastError();
}

}  // namespace

int dgrad_r_launch(void* stream, const DgradRArgs& a, int num_cus, int* grid_used) {
  if ((a.nout != 128 && a.nout != 256) || a.M < 1 || a.max_slabs < 1 || !a.A || !a.Wpacked || !a.X || !a.out || !a.slab || !a.colslab)
    return (int)hipErrorInvalidValue;
  if ((a.M + 31) / 32 > 0x7fffffff / 2 || ((uintptr_t)a.A & 15) || ((uintptr_t)a.X & 15) || ((uintptr_t)a.out & 15)) return (int)hipErrorInvalidValue;
  if (a.nout == 256 && !a.relu_gate) return (int)hipErrorInvalidValue;
  if (a.nout == 128 && a.relu_gate) return (int)hipErrorInvalidValue;
  return a.nout == 128 ? launch_k<128, false>(stream, a, num_cus, grid_used) : launch_k<256, true>(stream, a, num_cus, grid_used);
}

int dgrad_r_pack_launch(void* stream, const float* const* srcs, const long long* dst_off, int n, int nout, float* dst) {
  if ((nout != 128 && nout != 256) || n < 1 || n > DGRAD_R_PACK_MAX || !srcs || !dst_off || !dst) return (int)hipErrorInvalidValue;
  DgradRPackArgs pa;
  for (int i = 0; i < DGRAD_R_PACK_MAX; ++i) {
    pa.src[i] = i < n ? srcs[i] : nullptr;
    pa.dst_off[i] = i < n ? dst_off[i] : 0;
  }
  pa.nout = nout;
  hipLaunchKernelGGL(dgrad_r_pack_kernel, dim3(16, n), dim3(256), 0, static_cast<hipStream_t>(stream), pa, dst);
  return (int)hipGetLastError();
}
