// fcln64.hip -- out = LayerNorm(A W^T + b) + res  for 64 output columns (DPRNN blocks: IntraChunkRNN / InterChunkRNN,
// src/model/dprnn.py:40-46, 83-87: bi-LSTM output [M][256] -> Linear(256 -> 64) -> LayerNorm(64) -> + x), gfx950.
//
// Why a kernel of its own: in the weights-stationary engine (gemm_ws.h) a 64-column output means a 64-row
// token tile for its four waves, i.e. 133 KB of double-buffered A tile + 17 KB of C tile in LDS -- ONE workgroup per CU, one wave
// per SIMD, and nothing to cover the staging, the barriers and the row-space epilogue: 0.52 of the fp32 peak and 3.7 TB/s for
// a kernel whose 1.5 kB of traffic per token makes it as much a bandwidth problem as a matrix one (21 FLOP / B).  Here:
//   * tile = 16 tokens (v_mfma_f32_16x16x4_f32): wave w owns output columns [16 w, 16 w + 16) with its W rows resident in
//     registers (64 per lane for K = 256), every wave reads the whole 16 x K token tile as A fragments;
//   * the tile comes by LDS-DMA, no registers: one 1-KiB request per token row, 16-byte chunks XOR-swizzled by the row so that
//     the fragment reads of 16 rows spread over the banks without padding; three tiles in flight per workgroup;
//   * 52 KB of LDS and < 128 registers: three workgroups per CU, so one's barriers and epilogue overlap the others' MFMAs
//     (memory waits overlap; vector instructions do not, section 3.5 of DESIGN.md);
//   * row-space epilogue as in the engine: the 16 x 64 product tile goes through LDS, a thread owns (row, 4 columns): bias,
//     LayerNorm statistics over the row's 16 lanes by DPP, residual, one 16-byte store.
// Persistent workgroups, tiles by static stride.  Compiled with -amdgpu-mfma-vgpr-form.
#include <hip/hip_runtime.h>

#include "common.h"
#include "fcln64.h"

typedef float f32x4v __attribute__((ext_vector_type(4)));
static __device__ __forceinline__ f32x4v mfma16(float a, float b, f32x4v c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

namespace {

constexpr int NOUT = 64, LDC = NOUT + 4;

// One LDS-DMA request of the wave: lane L's 16 bytes at `g` land at LDS byte address lds_base + 16 L.  Issued as inline assembly ON
// PURPOSE: the compiler then does not know that memory -> LDS traffic is outstanding.  If it knows (the builtin), it puts
// s_waitcnt vmcnt(0) in front of EVERY LDS access that follows -- it cannot tell the buffers apart -- and a wave sits out the whole
// latency of the tiles it has just requested, each iteration.  The price: every wait on this traffic is written by hand below.
static __device__ __forceinline__ void dma16(const void* g, uint32_t lds_base) {
  asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_base), "v"(g) : "memory", "m0");
}
static __device__ __forceinline__ uint32_t lds_addr(const void* p) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}

template <int KIN, int NBUF>
__global__ __launch_bounds__(256) void fcln64_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                      const float* __restrict__ bias, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, const float* __restrict__ res,
                                                      float* __restrict__ out, int64_t M, int ntiles) {
  static_assert(KIN == 256, "one 1-KiB request per token row");
  constexpr int MK = KIN / 16;                      // k-chunks of 16
  constexpr int TILE = 16 * KIN;                    // floats per staged tile (unpadded, swizzled)
  constexpr int RT = 16 * NOUT;                     // floats per residual tile
  constexpr int NREQ = 5;                           // requests per lane and tile: 4 token rows + 1 residual chunk
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                                 // [NBUF][16][KIN]
  float* Rs = smem + NBUF * TILE;                   // [NBUF][16][NOUT]
  float* Cs = Rs + NBUF * RT;                       // [16][LDC]
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, i16 = lane & 15, ks = lane >> 4;
  const int rrow = tid >> 4, c4 = tid & 15;          // row space: thread = (row of the tile, 4 columns)

  // ---- W rows of this wave's 16 columns -> registers (B fragments: lane (col i16, k-slot ks), true k = 16 m + 4 ks + t) ------
  float wf[4 * MK];
  {
    const float* wr = W + (int64_t)(16 * w + i16) * KIN + 4 * ks;
#pragma unroll
    for (int m = 0; m < MK; ++m) {
      const float4 t = *reinterpret_cast<const float4*>(wr + 16 * m);
      wf[4 * m + 0] = t.x; wf[4 * m + 1] = t.y; wf[4 * m + 2] = t.z; wf[4 * m + 3] = t.w;
    }
  }
  const float4 bc = *reinterpret_cast<const float4*>(bias + 4 * c4), ga = *reinterpret_cast<const float4*>(gamma + 4 * c4),
               be = *reinterpret_cast<const float4*>(beta + 4 * c4);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // from here on the only loads in flight are the hand-counted requests below

  // ---- staging: tile t -> buffer b.  Wave w fetches token rows 4 w .. 4 w + 3 (1 KiB each) and the residual rows 4 w .. 4 w + 3
  // (4 x 256 B = one request).  Row r's 16-byte chunk c sits at chunk position c ^ (r & 15) (low four bits): a fragment read -- 16
  // rows, same chunk -- then covers 16 different 16-byte bank groups without padding.  The lane FETCHES the chunk that belongs at
  // its position.  Rows beyond M read the last row.  The residual chunk of lane L of wave w is the one thread (w, L) consumes.
  const uint32_t as0 = lds_addr(As) + (uint32_t)(4 * w * KIN * 4), rs0 = lds_addr(Rs) + (uint32_t)(w * 1024);
  auto stage = [&](int tile, int buf) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = 4 * w + j;
      int64_t tok = (int64_t)tile * 16 + row;
      tok = tok < M ? tok : M - 1;
      const int ch = (lane & ~15) | ((lane ^ row) & 15);
      dma16(A + tok * KIN + 4 * ch, as0 + (uint32_t)((buf * TILE + j * KIN) * 4));
    }
    int64_t tok = (int64_t)tile * 16 + rrow;
    tok = tok < M ? tok : M - 1;
    dma16(res + tok * NOUT + 4 * c4, rs0 + (uint32_t)(buf * RT * 4));
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

    // ---- product: 16 x 16 tile of this wave, four chains over k --------------------------------------------------------
    f32x4v acc[4] = {(f32x4v){0.f, 0.f, 0.f, 0.f}, (f32x4v){0.f, 0.f, 0.f, 0.f}, (f32x4v){0.f, 0.f, 0.f, 0.f}, (f32x4v){0.f, 0.f, 0.f, 0.f}};
    const float* ab = As + b * TILE + alane;
#pragma unroll
    for (int m0 = 0; m0 < MK; m0 += 8) {
      float4 af[8];
#pragma unroll
      for (int m = 0; m < 8; ++m) af[m] = *reinterpret_cast<const float4*>(ab + 16 * (((m0 + m) & 3) ^ aq) + 64 * ((m0 + m) >> 2));
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        acc[0] = mfma16(af[m].x, wf[4 * (m0 + m) + 0], acc[0]);
        acc[1] = mfma16(af[m].y, wf[4 * (m0 + m) + 1], acc[1]);
        acc[2] = mfma16(af[m].z, wf[4 * (m0 + m) + 2], acc[2]);
        acc[3] = mfma16(af[m].w, wf[4 * (m0 + m) + 3], acc[3]);
      }
    }
    const f32x4v s = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    // D reg r of lane (i16, ks) = (row 4 ks + r, column 16 w + i16)
#pragma unroll
    for (int r = 0; r < 4; ++r) Cs[(4 * ks + r) * LDC + 16 * w + i16] = s[r];
    const float4 rs = *reinterpret_cast<const float4*>(Rs + b * RT + 4 * tid);     // (this thread's own request)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // Cs complete; every wave is done with buffer b
    {
      const int64_t t = (int64_t)tile + (int64_t)NBUF * G;
      stage(t < ntiles ? (int)t : last, b);
    }

    // ---- row space: bias, LayerNorm over the row's 64 columns (16 adjacent lanes), + residual, store ------------------------
    // (the next write of Cs is behind the next iteration's first barrier)
    {
      const int64_t tok = (int64_t)tile * 16 + rrow;
      const float4 cv = *reinterpret_cast<const float4*>(&Cs[rrow * LDC + 4 * c4]);
      f32x2 lo = (f32x2){cv.x, cv.y} + (f32x2){bc.x, bc.y}, hi = (f32x2){cv.z, cv.w} + (f32x2){bc.z, bc.w};
      const f32x2 t = lo + hi;
      const float mu = group_sum<16>(t.x + t.y) * (1.0f / NOUT);
      const f32x2 m2 = (f32x2){mu, mu};
      lo -= m2;
      hi -= m2;
      const f32x2 q = lo * lo + hi * hi;
      const float rstd = rsqrtf(group_sum<16>(q.x + q.y) * (1.0f / NOUT) + 1e-5f);
      const f32x2 r2 = (f32x2){rstd, rstd};
      const f32x2 ya = lo * r2 * (f32x2){ga.x, ga.y} + (f32x2){be.x, be.y} + (f32x2){rs.x, rs.y};
      const f32x2 yb = hi * r2 * (f32x2){ga.z, ga.w} + (f32x2){be.z, be.w} + (f32x2){rs.z, rs.w};
      if (tok < M) *reinterpret_cast<float4*>(out + tok * NOUT + 4 * c4) = make_float4(ya.x, ya.y, yb.x, yb.y);
    }
    b = b + 1 == NBUF ? 0 : b + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may be in flight when the workgroup retires
}

}  // namespace

static size_t lds_bytes(int nbuf) { return sizeof(float) * ((size_t)nbuf * 16 * (256 + NOUT) + 16 * LDC); }

int fcln64_launch(void* stream, int kin, const float* A, const float* W, const float* bias, const float* gamma, const float* beta,
                  const float* res, float* out, int64_t M, int num_cus, int nbuf) {
  if (kin != 256 || M < 1 || (M + 15) / 16 > (1 << 27)) return (int)hipErrorInvalidValue;   // (one direction, K = 128: the GEMM engine)
  if (nbuf != 2 && nbuf != 3) return (int)hipErrorInvalidValue;
  auto kern = nbuf == 3 ? fcln64_kernel<256, 3> : fcln64_kernel<256, 2>;
  const size_t lds = lds_bytes(nbuf);
  static PerDeviceOnce ready[2];
  const int dev = current_hip_device();
  const int ki = nbuf - 2;
  if (!ready[ki].done(dev)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    ready[ki].set(dev);
  }
  const int ntiles = (int)((M + 15) / 16);
  const int per_cu = nbuf == 3 ? 2 : 3;              // what the LDS admits (65.8 KB / 45.3 KB per workgroup)
  const int wgs = per_cu * (num_cus > 0 ? num_cus : 256);
  const int grid = ntiles < wgs ? ntiles : wgs;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, static_cast<hipStream_t>(stream), A, W, bias, gamma, beta, res, out, M, ntiles);
  return (int)hipGetLastError();
}
