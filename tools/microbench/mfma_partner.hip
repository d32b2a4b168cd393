// Micro-benchmark: does VALU work of a PARTNER wave (same SIMD) hide behind a wave's fp32 MFMA stream?
// 512 threads: waves 0-3 issue back-to-back v_mfma_f32_16x16x4_f32 (one per SIMD), waves 4-7 (their SIMD partners)
// issue FILL x v_fma / v_exp per iteration or idle.  Reports cycles per MFMA seen by the MFMA waves.
// build: hipcc -O3 --offload-arch=gfx950 -o mfma_partner mfma_partner.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4v __attribute__((ext_vector_type(4)));

template <int MODE>   // 0 partner idle (exits), 1 partner v_fma stream, 2 partner v_exp stream
__global__ __launch_bounds__(512) void k(const float* __restrict__ w, float* __restrict__ out,
                                          unsigned long long* __restrict__ cyc, int iters) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (wave < 4) {
    float wf[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) wf[i] = w[(size_t)i * 64 + lane];
    f32x4v acc[8];
#pragma unroll
    for (int a = 0; a < 8; ++a) acc[a] = (f32x4v){0.f, 0.f, 0.f, 0.f};
    const float av = 0.5f + lane;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int m = 0; m < 32; ++m)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, wf[(b * 8 + m) & 63], acc[b], 0, 0, 0);
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::"v"(acc[0][0]), "v"(acc[7][0]));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < 8; ++a) s += acc[a][0] + acc[a][1] + acc[a][2] + acc[a][3];
    out[(size_t)blockIdx.x * 512 + tid] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
  } else {
    if (MODE == 0) return;
    float fl[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) fl[i] = 0.01f * (lane + i);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int m = 0; m < 256; ++m) fl[m & 7] = MODE == 2 ? __builtin_amdgcn_exp2f(fl[m & 7]) : __builtin_fmaf(fl[m & 7], 0.999f, 0.001f);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += fl[i];
    out[(size_t)blockIdx.x * 512 + tid] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
  }
}

template <int MODE>
void run(const char* name, const float* w, float* out, unsigned long long* cyc) {
  const int iters = 400, grid = 142;
  hipMemset(cyc, 0, grid * 8 * 8);
  hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(512), 0, 0, w, out, cyc, 10);
  hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(512), 0, 0, w, out, cyc, iters);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(grid * 8);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double mm = 0, vv = 0;
  for (int g = 0; g < grid; ++g)
    for (int q = 0; q < 8; ++q) (q < 4 ? mm : vv) += (double)h[g * 8 + q];
  printf("%-28s MFMA waves: %.2f cycles/MFMA   partner waves: %.2f cycles/VALU instr\n", name, mm / (grid * 4) / (iters * 256.0),
         vv / (grid * 4) / (iters * 256.0));
}

int main() {
  float *w, *out;
  unsigned long long* cyc;
  hipMalloc(&w, 64 * 64 * 4);
  hipMalloc(&out, 256 * 512 * 4);
  hipMalloc(&cyc, 256 * 8 * 8);
  std::vector<float> hw(64 * 64);
  for (size_t i = 0; i < hw.size(); ++i) hw[i] = 0.001f * (float)((i * 7919) % 1000) - 0.5f;
  hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
  run<0>("partner idle", w, out, cyc);
  run<1>("partner v_fma stream", w, out, cyc);
  run<2>("partner v_exp stream", w, out, cyc);
  return 0;
}
