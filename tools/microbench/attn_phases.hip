// Diagnostic: where one (sequence, head) workgroup of attention_kernel<32,5> spends its cycles, at the bench shape
// (B=16: 2256 sequences x 150 positions, 4 heads).  Stamps are s_memtime differences averaged over waves.
// build: hipcc -O3 --offload-arch=gfx950 -I../../speech_separation_amd/csrc -I../../include -o attn_phases attn_phases.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ unsigned long long* g_stamps;
#define ATTN_STAMP(i)                                                                                         \
  do {                                                                                                        \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");                       \
    if ((threadIdx.x & 63) == 0)                                                                              \
      g_stamps[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + (threadIdx.x >> 6)) * 8 + (i)] = __builtin_amdgcn_s_memtime();      \
  } while (0)
#include "attention.h"

int main() {
  const int B = 16, S = 141, K = 150, N = 128, heads = 4;
  const size_t M = (size_t)B * S * K;
  float *qkv, *out;
  unsigned long long* st;
  hipMalloc(&qkv, M * 3 * N * 4);
  hipMalloc(&out, M * N * 4);
  std::vector<float> h(M * 3 * N);
  unsigned s = 12345u;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.f - 0.5f; }
  hipMemcpy(qkv, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  for (int mode = 0; mode < 2; ++mode) {
    SeqGeom g = make_geom(mode, B, S, K);
    const int nwg = g.nseq * heads;
    hipMalloc(&st, (size_t)nwg * 64 * 8);
    hipMemset(st, 0, (size_t)nwg * 64 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &st, sizeof(st));
    auto kern = attention_kernel<32, 5>;
    const size_t lds = AttnShape<32>::lds_bytes(5);
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    DropCfg drop{0u, 0u, 1.f};
    const float scale = 1.4426950408889634f / sqrtf(32.f);
    for (int it = 0; it < 3; ++it) {
      hipEventRecord(a);
      hipLaunchKernelGGL(kern, dim3(g.nseq, heads), dim3(320), lds, 0, qkv, out, N, heads, g, scale, drop);
      hipEventRecord(b);
      hipDeviceSynchronize();
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> hs((size_t)nwg * 64);
    hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost);
    double seg[5] = {0, 0, 0, 0, 0}, life = 0;
    for (int w = 0; w < nwg; ++w) {
      unsigned long long t0 = ~0ull, t1 = 0;
      for (int q = 0; q < 5; ++q) {
        const unsigned long long* p = &hs[((size_t)w * 8 + q) * 8];
        for (int i = 0; i < 5; ++i) seg[i] += (double)(p[i + 1] - p[i]);
        if (p[0] < t0) t0 = p[0];
        if (p[5] > t1) t1 = p[5];
      }
      life += (double)(t1 - t0);
    }
    const double nw = (double)nwg * 5;
    printf("%s: %.3f ms (with stamps)  per wave cycles: load+stage %.0f  QK^T %.0f  softmax %.0f  PV %.0f  store %.0f   WG lifetime %.0f\n",
           mode ? "inter" : "intra", ms, seg[0] / nw, seg[1] / nw, seg[2] / nw, seg[3] / nw, seg[4] / nw, life / nwg);
    hipFree(st);
  }
  return 0;
}
