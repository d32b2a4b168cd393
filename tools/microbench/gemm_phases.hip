// Diagnostic: where a workgroup of the weights-stationary GEMM engine spends its cycles per 32-row tile, for the
// out-projection + LayerNorm kernel (K3: 128 -> 128) and the QKV kernel (K1: 128 -> 384) at the half-batch shape
// (169 200 tokens).  Per-wave s_memtime differences between the phase boundaries, summed over tiles.
// build: hipcc -O3 --offload-arch=gfx950 -I../../speech_separation_amd/csrc -I../../include -o gemm_phases gemm_phases.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ unsigned long long* g_seg;
#define GEMM_STAMP 1
// per-wave accumulators in registers, written once at the end (a stamp costs one s_memtime + a 64-bit add)
#define GEMM_STAMP_ENTRY const unsigned long long t_entry_ = __builtin_amdgcn_s_memtime();
#define GEMM_STAMP_DECL                                   \
  unsigned long long seg_[7] = {0, 0, 0, 0, 0, 0, 0};    \
  unsigned long long t_last_ = __builtin_amdgcn_s_memtime(); \
  seg_[6] = t_last_ - t_entry_;   /* prologue: weight fragments, first ticket, first A tile requested */
#define GEMM_STAMP(i)                                                 \
  do {                                                                \
    const unsigned long long n_ = __builtin_amdgcn_s_memtime();       \
    seg_[i] += n_ - t_last_;                                          \
    t_last_ = n_;                                                     \
  } while (0)
#define GEMM_STAMP_ACC(i, x)                             \
  do {                                                   \
    asm volatile("s_nop 15\n\ts_nop 15" ::"v"(x));        \
    GEMM_STAMP(i);                                       \
  } while (0)
#define GEMM_STAMP_END                                                                                       \
  if ((threadIdx.x & 63) == 0) {                                                                             \
    for (int i_ = 0; i_ < 7; ++i_) g_seg[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + i_] = seg_[i_];  \
  }
#include "gemm_ws.h"
#include "backward.h"

// timing-only ablation: the LSTM pre-activation epilogue WITHOUT its global stores (is the A->LDS phase waiting for the
// previous tile's stores to retire?)
struct EpiLstmPre16NoStore : EpiLstmPre16 {
  DEV void store_acc(int tile, int wr, int d, int cb, const f32x16& acc, int c, int hh, float bias) const {
    const float gs = l16_gate_scale(cb >> 2);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float4 v = make_float4(fmaf(acc[4 * q + 0], gs, bias), fmaf(acc[4 * q + 1], gs, bias), fmaf(acc[4 * q + 2], gs, bias),
                             fmaf(acc[4 * q + 3], gs, bias));
      asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
    }
  }
};

template <class Kern, class AL, class EP>
void run(const char* name, Kern kern, size_t lds, int wgs, int ntiles, const float* W, int ldw, unsigned* queue, AL al, EP ep,
         unsigned long long* seg) {
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  float ms = 0;
  for (int it = 0; it < 3; ++it) {
    hipMemset(queue, 0, 64);
    hipMemset(seg, 0, (size_t)wgs * 4 * 8 * 8);
    hipEventRecord(a);
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), lds, 0, W, (const float*)nullptr, ldw, ntiles, queue, al, ep, NoRider{});
    hipEventRecord(b);
    hipDeviceSynchronize();
    hipEventElapsedTime(&ms, a, b);
  }
  std::vector<unsigned long long> h((size_t)wgs * 32);
  hipMemcpy(h.data(), seg, h.size() * 8, hipMemcpyDeviceToHost);
  double s[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int w = 0; w < wgs * 4; ++w)
    for (int i = 0; i < 7; ++i) s[i] += (double)h[(size_t)w * 8 + i];
  const double per = (double)ntiles * 4;   // wave-tiles
  printf("%-26s %4d WGs %.3f ms | cycles per tile and wave: ticket + A->LDS %.0f  barrier %.0f  prefetch issue %.0f  frag+MFMA %.0f  "
         "C->LDS+barrier %.0f  epilogue %.0f | sum %.0f | prologue per wave %.0f cycles, %.1f tiles per workgroup\n",
         name, wgs, ms, s[0] / per, s[1] / per, s[2] / per, s[3] / per, s[4] / per, s[5] / per,
         (s[0] + s[1] + s[2] + s[3] + s[4] + s[5]) / per, s[6] / (wgs * 4.0), (double)ntiles / wgs);
}

int main() {
  const int64_t M = 169200;
  const int N = 128, ntiles = (int)((M + 31) / 32);
  float *A, *X, *Y, *Q, *W, *bias, *gam, *bet;
  unsigned* queue;
  unsigned long long* seg;
  hipMalloc(&A, M * N * 4); hipMalloc(&X, M * N * 4); hipMalloc(&Y, M * N * 4); hipMalloc(&Q, M * 3 * N * 4);
  hipMalloc(&W, 3 * N * N * 4); hipMalloc(&bias, 3 * N * 4); hipMalloc(&gam, N * 4); hipMalloc(&bet, N * 4);
  hipMalloc(&queue, 64); hipMalloc(&seg, 2048 * 4 * 8 * 8);
  hipMemset(A, 0, M * N * 4); hipMemset(X, 0, M * N * 4); hipMemset(W, 0, 3 * N * N * 4); hipMemset(bias, 0, 3 * N * 4);
  hipMemset(gam, 0, N * 4); hipMemset(bet, 0, N * 4);
  hipMemcpyToSymbol(HIP_SYMBOL(g_seg), &seg, sizeof(seg));
  {
    ALoadDense al{A, M, N, 32};
    EpiBiasResLN<32> ep{Y, bias, X, gam, bet, M, N, 32};
    auto kern = gemm_ws_kernel<128, 1, 1, 4, ALoadDense, EpiBiasResLN<32>, false>;
    const size_t lds = GemmShape<128, 1, 1, 4>::lds_bytes(false);
    for (int wgs : {256, 512, 768}) run("K3 out-proj + LN", kern, lds, wgs, ntiles, W, N, queue, al, ep, seg);
  }
  {
    ALoadDense al{A, M, N, 32};
    EpiBiasStore ep{Q, bias, M, 3 * N, 32, 3 * N};
    auto kern = gemm_ws_kernel<128, 3, 1, 4, ALoadDense, EpiBiasStore, false>;
    const size_t lds = GemmShape<128, 3, 1, 4>::lds_bytes(false);
    for (int wgs : {256}) run("K1 qkv", kern, lds, wgs, ntiles, W, N, queue, al, ep, seg);
  }
  {   // K6: ffn 256 -> 128 + LN, A = ReLU(h) columns
    float* HC;
    hipMalloc(&HC, (M + 32) * 256 * 4);
    hipMemset(HC, 0, (M + 32) * 256 * 4);
    float* W6;
    hipMalloc(&W6, 128 * 256 * 4);
    hipMemset(W6, 0, 128 * 256 * 4);
    ALoadCols al{HC, M, 256, 0, 32};
    EpiBiasResLN<32> ep{Y, bias, X, gam, bet, M, N, 32};
    auto kern = gemm_ws_kernel<256, 1, 1, 4, ALoadCols, EpiBiasResLN<32>, false>;
    const size_t lds = GemmShape<256, 1, 1, 4>::lds_bytes(false);
    for (int wgs : {256}) run("K6 ffn + LN", kern, lds, wgs, ntiles, W6, 256, queue, al, ep, seg);
  }
  {   // K4: LSTM pre-activations 128 -> 2 x 512, sequence-tile rows, direct epilogue (one direction = one column group)
    const int B = 8, S = 141, K = 150;
    SeqGeom g = make_geom(0, B, S, K);
    float *PRE, *W4, *b4;
    const int nst16 = (g.nseq + 15) / 16;
    hipMalloc(&PRE, (size_t)2 * g.nst * g.len * 16384 * 4);
    hipMalloc(&W4, 1024 * 128 * 4); hipMemset(W4, 0, 1024 * 128 * 4);
    hipMalloc(&b4, 1024 * 4); hipMemset(b4, 0, 1024 * 4);
    ALoadSeqTile al{A, N, g};
    EpiLstmPre16 ep{PRE, {b4, b4 + 512}, {b4, b4 + 512}, g, nst16};
    auto kern = gemm_ws_kernel<128, 4, 1, 4, ALoadSeqTile, EpiLstmPre16, false>;
    const size_t lds = GemmShape<128, 4, 1, 4>::lds_bytes(true);
    // one column group only (the harness launches a 1-D grid): 256 workgroups over nst*len tiles
    run("K4 lstm-pre (1 direction)", kern, lds, 256, g.nst * g.len, W4, N, queue, al, ep, seg);
    EpiLstmPre16NoStore ep2{{PRE, {b4, b4 + 512}, {b4, b4 + 512}, g, nst16}};
    auto kern2 = gemm_ws_kernel<128, 4, 1, 4, ALoadSeqTile, EpiLstmPre16NoStore, false>;
    run("K4 without its stores", kern2, lds, 256, g.nst * g.len, W4, N, queue, al, ep2, seg);
  }
  {   // training: d x = d_out + dP W_ih (K = 512 gate columns of one direction -> 128), the largest kernel of a training step
    float *DG, *DX, *W5;
    hipMalloc(&DG, (M + 32) * 1024 * 4); hipMemset(DG, 0, (M + 32) * 1024 * 4);
    hipMalloc(&DX, (M + 32) * 128 * 4); hipMemset(DX, 0, (M + 32) * 128 * 4);
    hipMalloc(&W5, 512 * 128 * 4); hipMemset(W5, 0, 512 * 128 * 4);
    ALoadCols al{DG, M, 1024, 0, 32};
    EpiAddMaskStoreT<true, false> ep{DX, X, nullptr, M, N, 32, N};
    auto kern = gemm_ws_kernel<512, 1, 1, 4, ALoadCols, EpiAddMaskStoreT<true, false>, true>;
    const size_t lds = GemmShape<512, 1, 1, 4>::lds_bytes(false);
    run("dgrad dP W_ih (K = 512)", kern, lds, 256, ntiles, W5, N, queue, al, ep, seg);
  }
  return 0;
}
