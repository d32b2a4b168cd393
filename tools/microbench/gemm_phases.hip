// Diagnostic: where a workgroup of the weights-stationary GEMM engine spends its cycles per 32-row tile, for the
// out-projection + LayerNorm kernel (K3: 128 -> 128) and the QKV kernel (K1: 128 -> 384) at the half-batch shape
// (169 200 tokens).  Per-wave s_memtime differences between the phase boundaries, summed over tiles.
// build: hipcc -O3 --offload-arch=gfx950 -I../../speech_separation_amd/csrc -I../../include -o gemm_phases gemm_phases.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ unsigned long long* g_seg;
#define GEMM_STAMP 1
#define GEMM_STAMP_DECL unsigned long long t_last_ = __builtin_amdgcn_s_memtime();
#define GEMM_STAMP(i)                                                                                              \
  do {                                                                                                             \
    const unsigned long long n_ = __builtin_amdgcn_s_memtime();                                                    \
    if ((threadIdx.x & 63) == 0) g_seg[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + (i)] += n_ - t_last_;  \
    t_last_ = n_;                                                                                                  \
  } while (0)
#define GEMM_STAMP_ACC(i, x)                             \
  do {                                                   \
    asm volatile("s_nop 15\n\ts_nop 15" ::"v"(x));        \
    GEMM_STAMP(i);                                       \
  } while (0)
#include "gemm_ws.h"

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
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), lds, 0, W, (const float*)nullptr, ldw, ntiles, queue, al, ep);
    hipEventRecord(b);
    hipDeviceSynchronize();
    hipEventElapsedTime(&ms, a, b);
  }
  std::vector<unsigned long long> h((size_t)wgs * 32);
  hipMemcpy(h.data(), seg, h.size() * 8, hipMemcpyDeviceToHost);
  double s[6] = {0, 0, 0, 0, 0, 0};
  for (int w = 0; w < wgs * 4; ++w)
    for (int i = 0; i < 6; ++i) s[i] += (double)h[(size_t)w * 8 + i];
  const double per = (double)ntiles * 4;   // wave-tiles
  printf("%-26s %4d WGs %.3f ms | cycles per tile and wave: loop-top %.0f  A->LDS+barrier %.0f  prefetch issue %.0f  frag+MFMA %.0f  "
         "C->LDS+barrier %.0f  epilogue %.0f | sum %.0f\n",
         name, wgs, ms, s[0] / per, s[1] / per, s[2] / per, s[3] / per, s[4] / per, s[5] / per,
         (s[0] + s[1] + s[2] + s[3] + s[4] + s[5]) / per);
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
  return 0;
}
