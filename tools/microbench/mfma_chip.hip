// Micro-benchmark: CHIP-WIDE sustained throughput of dense MFMA streams (one wave per SIMD on every CU, operands in
// registers, ~20 ms per launch): what clock does the chip hold, and does it depend on the instruction shape?
//   v_mfma_f32_32x32x2_f32 (GEMM engine / attention)   vs   v_mfma_f32_16x16x4_f32 (lstm16)   vs the bf16 forms
// build: hipcc -O3 --offload-arch=gfx950 -o mfma_chip mfma_chip.hip ; run: ./mfma_chip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ __launch_bounds__(256) void k(float* __restrict__ out, int iters) {
  const int lane = threadIdx.x & 63;
  float a = 0.001f * lane, b = 0.002f * lane;
  bf16x8 ab, bb;
  for (int i = 0; i < 8; ++i) { ab[i] = (__bf16)(a + i); bb[i] = (__bf16)(b - i); }
  f32x16 c32[4];
  f32x4 c16[8];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) c32[i][r] = 0.f;
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) c16[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (KIND == 0) {
#pragma unroll
        for (int g = 0; g < 4; ++g) c32[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c32[g], 0, 0, 0);
      } else if (KIND == 1) {
#pragma unroll
        for (int g = 0; g < 8; ++g) c16[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c16[g], 0, 0, 0);
      } else if (KIND == 2) {
#pragma unroll
        for (int g = 0; g < 4; ++g) c32[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c32[g], 0, 0, 0);
      } else {
#pragma unroll
        for (int g = 0; g < 8; ++g) c16[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, c16[g], 0, 0, 0);
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += c32[i][r];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) s += c16[i][r];
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND>
void run(const char* name, double flop_per_mfma, int per_iter, int grid, float* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = KIND < 2 ? 40000 : 160000;
  hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, out, 100);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)grid * 4 * iters * per_iter;
    const double cyc_per = KIND == 0 ? 64 : KIND == 1 ? 32 : KIND == 2 ? 32 : 16;
    printf("%-28s grid %4d  %8.2f ms  %8.1f TFLOP/s  implied clock %.3f GHz\n", name, grid, ms,
           mfmas * flop_per_mfma / (ms * 1e-3) / 1e12, mfmas * cyc_per / (grid * 4.0) / (ms * 1e-3) / 1e9);
  }
}

int main() {
  float* out;
  hipMalloc(&out, 1024 * 256 * sizeof(float));
  for (int grid : {256, 146}) {
    run<0>("f32 32x32x2", 4096.0, 64, grid, out);
    run<1>("f32 16x16x4", 2048.0, 128, grid, out);
    run<2>("bf16 32x32x16", 32768.0, 64, grid, out);
    run<3>("bf16 16x16x32", 16384.0, 128, grid, out);
  }
  return 0;
}
