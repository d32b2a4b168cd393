// fcln.hip -- a Linear layer with its LayerNorm and residual on 16-token tiles (gfx950); the forms are listed in fcln.h:
//     DPRNN blocks         out = LayerNorm(h W_fc^T + b) + x                      src/model/dprnn.py:41-45, 83-87   (256 -> 64)
//     DPTN                 y1  = LayerNorm1(att W_o^T + b_o + x)                  src/model/dptn.py:46-47            (128 -> 128, tape)
//                          out = LayerNorm2(ReLU(h) W_f^T + b_f + y1)             src/model/dptn.py:50-51            (256 -> N)
//     separation conv      Z   = PReLU(x) W_sep^T + b_sep   (no LayerNorm)        src/model/dptn_wav.py:26-29, 47    (N -> 2 N)
//
// Why a kernel of its own: in the weights-stationary engine (gemm_ws.h) these launches run ONE workgroup per CU
// (64- or 32-token tiles double-buffered in LDS), one wave per SIMD, and nothing covers the staging, the barriers and the row-space
// epilogue: 0.36-0.52 of the fp32 peak for kernels whose 1.5-2.5 kB of traffic per token make them as much a bandwidth problem as
// a matrix one (21-26 FLOP / B; the chip's balance point is 25).  Here:
//   * tile = 16 tokens (v_mfma_f32_16x16x4_f32): wave w owns NOUT / 4 output columns with its W rows resident in registers (64 per
//     lane for 16 columns at K = 256), every wave reads the whole 16 x K token tile as A fragments;
//   * the token tile AND the residual rows come by LDS-DMA, no registers: 16-byte chunks XOR-swizzled by the row so that the
//     fragment reads of 16 rows spread over the banks without padding; one or two tiles in flight per workgroup;
//   * 40-66 KB of LDS and <= 256 registers: two or three workgroups per CU, so one's barriers and epilogue overlap the others'
//     MFMAs (memory waits overlap; vector instructions do not, section 3.5 of DESIGN.md);
//   * row-space epilogue as in the engine: the 16 x NOUT product tile goes through LDS, a thread owns (row, 4 columns per 64):
//     bias, LayerNorm statistics over the row's 16 lanes by DPP, residual, 16-byte stores (+ the normalised row and 1/sigma for
//     the LayerNorm backward when the training tape asks for them).
// Persistent workgroups, tiles by static stride.  Compiled with -amdgpu-mfma-vgpr-form.
#include <hip/hip_runtime.h>

#include "common.h"
#include "fcln.h"

typedef float f32x4v __attribute__((ext_vector_type(4)));
static __device__ __forceinline__ f32x4v mfma16(float a, float b, f32x4v c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

namespace {

// One LDS-DMA request of the wave: lane L's 16 bytes at `g` land at LDS byte address lds_base + 16 L.  Issued as inline assembly ON
// PURPOSE: the compiler then does not know that memory -> LDS traffic is outstanding.  If it knows (the builtin), it puts
// s_waitcnt vmcnt(0) in front of EVERY LDS access that follows -- it cannot tell the buffers apart -- and a wave sits out the whole
// latency of the tiles it has just requested, each iteration.  The price: every wait on this traffic is written by hand below.
static __device__ __forceinline__ void dma16(const void* g, uint32_t lds_base) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_base), "v"(g) : "memory", "m0");
}
static __device__ __forceinline__ uint32_t lds_addr(const void* p) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}

// ACT: what A is read through (0 nothing, 1 ReLU, 2 PReLU with the shared slope *act_w); LN: LayerNorm + residual epilogue (else
// bias + store: no residual tile is fetched); PRE: the residual goes in before the LayerNorm; SAVE: normalised rows + 1/sigma out
template <int KIN, int NOUT, int NBUF, int ACT, bool LN, bool PRE, bool SAVE>
__global__ __launch_bounds__(256) void fcln_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                    const float* __restrict__ bias, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, const float* __restrict__ res,
                                                    const float* __restrict__ act_w, float* __restrict__ out,
                                                    float* __restrict__ zn_out, float* __restrict__ rstd_out, int64_t M, int ntiles) {
  static_assert((KIN == 256 || KIN == 128 || KIN == 64) && (NOUT == 64 || NOUT == 128 || NOUT == 256), "shapes of the DPRNN / DPTN blocks");
  static_assert(LN || (!PRE && !SAVE), "the plain epilogue has no residual and no tape");
  constexpr int MK = KIN / 16;                      // k-chunks of 16
  constexpr int NB = NOUT / 64;                     // 16-column blocks per wave = 64-column segments per row
  constexpr int LDC = NOUT + 4;
  constexpr int TILE = 16 * KIN;                    // floats per staged token tile (unpadded, swizzled)
  constexpr int RT = LN ? 16 * NOUT : 0;            // floats per residual tile
  constexpr int CPR = KIN / 4;                      // 16-byte chunks per token row
  constexpr int RPR = 256 / KIN;                    // token rows per 1-KiB request
  constexpr int AREQ = KIN / 64, RREQ = LN ? NOUT / 64 : 0;  // requests per lane and tile
  constexpr int NREQ = AREQ + RREQ;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                                 // [NBUF][16][KIN]
  float* Rs = smem + NBUF * TILE;                   // [NBUF][16][NOUT]
  float* Cs = Rs + NBUF * RT;                       // [16][LDC]
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, i16 = lane & 15, ks = lane >> 4;
  const int rrow = tid >> 4, c4 = tid & 15;          // row space: thread = (row of the tile, 4 columns of every 64)

  // ---- W rows of this wave's columns -> registers (B fragments: lane (col i16, k-slot ks), true k = 16 m + 4 ks + t) --------
  float wf[NB][4 * MK];
#pragma unroll
  for (int bb = 0; bb < NB; ++bb) {
    const float* wr = W + (int64_t)((NOUT / 4) * w + 16 * bb + i16) * KIN + 4 * ks;
#pragma unroll
    for (int m = 0; m < MK; ++m) {
      const float4 t = *reinterpret_cast<const float4*>(wr + 16 * m);
      wf[bb][4 * m + 0] = t.x; wf[bb][4 * m + 1] = t.y; wf[bb][4 * m + 2] = t.z; wf[bb][4 * m + 3] = t.w;
    }
  }
  float4 bc[NB], ga[LN ? NB : 1], be[LN ? NB : 1];
#pragma unroll
  for (int s = 0; s < NB; ++s) {
    bc[s] = *reinterpret_cast<const float4*>(bias + 64 * s + 4 * c4);
    if (LN) {
      ga[s] = *reinterpret_cast<const float4*>(gamma + 64 * s + 4 * c4);
      be[s] = *reinterpret_cast<const float4*>(beta + 64 * s + 4 * c4);
    }
  }
  const float slope = ACT == 2 ? *act_w : 0.f;      // nn.PReLU(): one slope for all channels
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // from here on the only loads in flight are the hand-counted requests below

  // ---- staging: tile t -> buffer b.  Wave w fetches token rows 4 w .. 4 w + 3 (1 KiB per request: one row at K = 256, two at
  // K = 128) and its 64 lanes' share of the residual tile.  Row r's 16-byte chunk c sits at chunk position c ^ (r & 15) (low four
  // bits): a fragment read -- 16 rows, same chunk -- then covers 16 different 16-byte bank groups without padding.  The lane FETCHES
  // the chunk that belongs at its position.  Rows beyond M read the last row.
  const uint32_t as0 = lds_addr(As) + (uint32_t)(4 * w * KIN * 4), rs0 = lds_addr(Rs) + (uint32_t)(w * 1024);
  auto stage = [&](int tile, int buf) {
#pragma unroll
    for (int j = 0; j < AREQ; ++j) {
      const int row = 4 * w + j * RPR + lane / CPR;
      const int pos = lane % CPR;
      int64_t tok = (int64_t)tile * 16 + row;
      tok = tok < M ? tok : M - 1;
      const int ch = (pos & ~15) | ((pos ^ row) & 15);
      dma16(A + tok * KIN + 4 * ch, as0 + (uint32_t)((buf * TILE + j * RPR * KIN) * 4));
    }
#pragma unroll
    for (int q = 0; q < RREQ; ++q) {
      const int idx = tid + 256 * q;                // chunk of the residual tile, row-major
      int64_t tok = (int64_t)tile * 16 + idx / (NOUT / 4);
      tok = tok < M ? tok : M - 1;
      dma16(res + tok * NOUT + 4 * (idx % (NOUT / 4)), rs0 + (uint32_t)((buf * RT + 1024 * q) * 4));
    }
  };
  // Tiles by static stride (workgroup g: tiles g, g + G, ...; a ticket counter's returning atomic would join the hand-counted
  // requests).  Past the end a valid tile is requested and never consumed: the request count per iteration stays constant.
  const int G = (int)gridDim.x;
  const int last = ntiles - 1;
#pragma unroll
  for (int b = 0; b < NBUF; ++b) {
    const int64_t t = (int64_t)blockIdx.x + (int64_t)b * G;
    stage(t < ntiles ? (int)t : last, b);
  }

  // A-fragment offsets of this lane: chunk 4 m + ks of row i16 at position (4 m + ks) ^ i16 = 4 (m ^ (i16 >> 2)) | (ks ^ (i16 & 3))
  const int alane = i16 * KIN + 4 * (ks ^ (i16 & 3));
  const int aq = i16 >> 2;

  int b = 0;
  for (int tile = (int)blockIdx.x; tile < ntiles; tile += G) {
    // Tile `tile` was requested NBUF iterations ago; NREQ (NBUF - 1) requests of this wave are younger.  Loads return in order, so
    // "at most that many operations outstanding" completes this tile's requests whatever the (unordered) stores among them do.
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NREQ * (NBUF - 1)) : "memory");   // all four waves' rows are in
    // (raw barriers: __syncthreads() carries a release fence, i.e. s_waitcnt vmcnt(0))

    // ---- product: 16 x (NOUT / 4) tile of this wave, four chains over k per column block -----------------------------------
    f32x4v acc[NB][4];
#pragma unroll
    for (int bb = 0; bb < NB; ++bb)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[bb][t] = (f32x4v){0.f, 0.f, 0.f, 0.f};
    const float* ab = As + b * TILE + alane;
    constexpr int MB = MK < 8 ? MK : 8;            // fragment reads in flight
#pragma unroll
    for (int m0 = 0; m0 < MK; m0 += MB) {
      float4 af[MB];
#pragma unroll
      for (int m = 0; m < MB; ++m) {
        af[m] = *reinterpret_cast<const float4*>(ab + 16 * (((m0 + m) & 3) ^ aq) + 64 * ((m0 + m) >> 2));
        if (ACT == 1) af[m] = make_float4(fmaxf(af[m].x, 0.f), fmaxf(af[m].y, 0.f), fmaxf(af[m].z, 0.f), fmaxf(af[m].w, 0.f));
        if (ACT == 2)     // max(x, 0) + slope * min(x, 0)
          af[m] = make_float4(fmaf(slope, fminf(af[m].x, 0.f), fmaxf(af[m].x, 0.f)), fmaf(slope, fminf(af[m].y, 0.f), fmaxf(af[m].y, 0.f)),
                              fmaf(slope, fminf(af[m].z, 0.f), fmaxf(af[m].z, 0.f)), fmaf(slope, fminf(af[m].w, 0.f), fmaxf(af[m].w, 0.f)));
      }
#pragma unroll
      for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int bb = 0; bb < NB; ++bb) {
          acc[bb][0] = mfma16(af[m].x, wf[bb][4 * (m0 + m) + 0], acc[bb][0]);
          acc[bb][1] = mfma16(af[m].y, wf[bb][4 * (m0 + m) + 1], acc[bb][1]);
          acc[bb][2] = mfma16(af[m].z, wf[bb][4 * (m0 + m) + 2], acc[bb][2]);
          acc[bb][3] = mfma16(af[m].w, wf[bb][4 * (m0 + m) + 3], acc[bb][3]);
        }
    }
    // D reg r of lane (i16, ks) = (row 4 ks + r, column (NOUT / 4) w + 16 bb + i16)
#pragma unroll
    for (int bb = 0; bb < NB; ++bb) {
      const f32x4v s = (acc[bb][0] + acc[bb][1]) + (acc[bb][2] + acc[bb][3]);
#pragma unroll
      for (int r = 0; r < 4; ++r) Cs[(4 * ks + r) * LDC + (NOUT / 4) * w + 16 * bb + i16] = s[r];
    }
    float4 rs[LN ? NB : 1];
    if (LN) {
#pragma unroll
      for (int s = 0; s < NB; ++s) rs[s] = *reinterpret_cast<const float4*>(Rs + b * RT + rrow * NOUT + 64 * s + 4 * c4);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // Cs complete; every wave is done with buffer b
    {
      const int64_t t = (int64_t)tile + (int64_t)NBUF * G;
      stage(t < ntiles ? (int)t : last, b);
    }

    // ---- row space: bias (+ residual), LayerNorm over the row's NOUT columns (16 adjacent lanes), (+ residual), store ---------
    // (the next write of Cs is behind the next iteration's first barrier)
    if (!LN) {
      const int64_t tok = (int64_t)tile * 16 + rrow;
      if (tok < M) {
#pragma unroll
        for (int s = 0; s < NB; ++s) {
          const float4 cv = *reinterpret_cast<const float4*>(&Cs[rrow * LDC + 64 * s + 4 * c4]);
          const f32x2 lo = (f32x2){cv.x, cv.y} + (f32x2){bc[s].x, bc[s].y}, hi = (f32x2){cv.z, cv.w} + (f32x2){bc[s].z, bc[s].w};
          *reinterpret_cast<float4*>(out + tok * NOUT + 64 * s + 4 * c4) = make_float4(lo.x, lo.y, hi.x, hi.y);
        }
      }
    } else {
      const int64_t tok = (int64_t)tile * 16 + rrow;
      f32x2 lo[NB], hi[NB];
      f32x2 t = (f32x2){0.f, 0.f};
#pragma unroll
      for (int s = 0; s < NB; ++s) {
        const float4 cv = *reinterpret_cast<const float4*>(&Cs[rrow * LDC + 64 * s + 4 * c4]);
        lo[s] = (f32x2){cv.x, cv.y} + (f32x2){bc[s].x, bc[s].y};
        hi[s] = (f32x2){cv.z, cv.w} + (f32x2){bc[s].z, bc[s].w};
        if (PRE) {
          lo[s] += (f32x2){rs[s].x, rs[s].y};
          hi[s] += (f32x2){rs[s].z, rs[s].w};
        }
        t += lo[s] + hi[s];
      }
      const float mu = group_sum<16>(t.x + t.y) * (1.0f / NOUT);
      const f32x2 m2 = (f32x2){mu, mu};
      f32x2 q = (f32x2){0.f, 0.f};
#pragma unroll
      for (int s = 0; s < NB; ++s) {
        lo[s] -= m2;
        hi[s] -= m2;
        q += lo[s] * lo[s] + hi[s] * hi[s];
      }
      const float rstd = rsqrtf(group_sum<16>(q.x + q.y) * (1.0f / NOUT) + 1e-5f);
      const f32x2 r2 = (f32x2){rstd, rstd};
      if (tok < M) {
#pragma unroll
        for (int s = 0; s < NB; ++s) {
          const f32x2 za = lo[s] * r2, zb = hi[s] * r2;
          f32x2 ya = za * (f32x2){ga[s].x, ga[s].y} + (f32x2){be[s].x, be[s].y};
          f32x2 yb = zb * (f32x2){ga[s].z, ga[s].w} + (f32x2){be[s].z, be[s].w};
          if (!PRE) {
            ya += (f32x2){rs[s].x, rs[s].y};
            yb += (f32x2){rs[s].z, rs[s].w};
          }
          *reinterpret_cast<float4*>(out + tok * NOUT + 64 * s + 4 * c4) = make_float4(ya.x, ya.y, yb.x, yb.y);
          if (SAVE) *reinterpret_cast<float4*>(zn_out + tok * NOUT + 64 * s + 4 * c4) = make_float4(za.x, za.y, zb.x, zb.y);
        }
        if (SAVE && c4 == 0) rstd_out[tok] = rstd;
      }
    }
    b = b + 1 == NBUF ? 0 : b + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may be in flight when the workgroup retires
}

template <int KIN, int NOUT, int NBUF, int ACT, bool LN, bool PRE, bool SAVE>
int launch(hipStream_t st, const FclnArgs& a, int num_cus) {
  auto kern = fcln_kernel<KIN, NOUT, NBUF, ACT, LN, PRE, SAVE>;
  const size_t lds = sizeof(float) * ((size_t)NBUF * 16 * (KIN + (LN ? NOUT : 0)) + 16 * (NOUT + 4));
  static PerDeviceOnce ready;          // (per instantiation)
  static std::atomic<int> per_cu{1};
  const int dev = current_hip_device();
  if (!ready.done(dev)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    int nb = 0;     // workgroups a CU holds (LDS and registers): the persistent grid is exactly what is resident
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(kern), 256, lds);
    if (e != hipSuccess) return (int)e;
    per_cu.store(nb < 1 ? 1 : nb);
    ready.set(dev);
  }
  const int ntiles = (int)((a.M + 15) / 16);
  const int wgs = per_cu.load() * (num_cus > 0 ? num_cus : 256);
  const int grid = ntiles < wgs ? ntiles : wgs;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a.A, a.W, a.bias, a.gamma, a.beta, a.res, a.act_w, a.out, a.zn, a.rstd, a.M,
                     ntiles);
  return (int)hipGetLastError();
}

}  // namespace

int fcln_launch(void* stream, const FclnArgs& a, int num_cus) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (a.M < 1 || (a.M + 15) / 16 > (1 << 27) || (a.zn == nullptr) != (a.rstd == nullptr)) return (int)hipErrorInvalidValue;
  if (!a.A || !a.W || !a.bias || !a.out || (a.layernorm && (!a.gamma || !a.beta || !a.res))) return (int)hipErrorInvalidValue;
  const bool save = a.zn != nullptr;
  if (!a.layernorm) {                                            // separation conv: PReLU -> Linear(N -> 2 N)
    if (a.act != 2 || a.act_w == nullptr || a.pre_res || save) return (int)hipErrorInvalidValue;
    if (a.kin == 128 && a.nout == 256) return launch<128, 256, 2, 2, false, false, false>(st, a, num_cus);
    if (a.kin == 64 && a.nout == 128) return launch<64, 128, 2, 2, false, false, false>(st, a, num_cus);
    return (int)hipErrorInvalidValue;
  }
  if (a.kin == 256 && a.nout == 64 && !a.pre_res && a.act == 0 && !save)
    return a.nbuf == 3 ? launch<256, 64, 3, 0, true, false, false>(st, a, num_cus) : launch<256, 64, 2, 0, true, false, false>(st, a, num_cus);
  if (a.kin == 256 && a.nout == 128 && a.pre_res && a.act == 1 && save) return launch<256, 128, 2, 1, true, true, true>(st, a, num_cus);
  if (a.kin == 128 && a.nout == 128 && a.pre_res && a.act == 0 && save) return launch<128, 128, 2, 0, true, true, true>(st, a, num_cus);
  if (a.kin == 256 && a.nout == 128 && a.pre_res && a.act == 0 && !save) return launch<256, 128, 2, 0, true, true, false>(st, a, num_cus);
  if (a.kin == 256 && a.nout == 64 && a.pre_res && a.act == 0 && !save) return launch<256, 64, 2, 0, true, true, false>(st, a, num_cus);
  return (int)hipErrorInvalidValue;      // (other shapes: the GEMM engine)
}
