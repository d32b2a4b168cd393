// Micro-benchmark: sustained rate of v_mfma_f32_16x16x4_f32 streams shaped like lstm16.h's step (256 MFMAs, 8
// accumulators, W resident in NW registers, A fragments batched from LDS).
// build: hipcc -O3 --offload-arch=gfx950 -o mfma16_rate mfma16_rate.hip ; run: ./mfma16_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4v __attribute__((ext_vector_type(4)));

// ORDER 0: k-major (all 8 accumulators per k step)   ORDER 1: half-major (4 accumulators, all k; then the other 4)
template <int NW, bool LDSA, int ORDER, int FILL = 0>
__global__ __launch_bounds__(256) void k(const float* __restrict__ w, float* __restrict__ out,
                                          unsigned long long* __restrict__ cyc, int iters) {
  __shared__ float As[16 * 136];
  const int tid = threadIdx.x, lane = tid & 63;
  float wf[NW];
#pragma unroll
  for (int i = 0; i < NW; ++i) wf[i] = w[(size_t)i * 64 + lane];
  for (int i = tid; i < 16 * 136; i += 256) As[i] = 0.001f * (i % 17);
  __syncthreads();
  f32x4v acc[8];
#pragma unroll
  for (int a = 0; a < 8; ++a) acc[a] = (f32x4v){0.f, 0.f, 0.f, 0.f};
  const float* arow = As + (lane & 15) * 136 + 4 * (lane >> 4);
  float fl[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) fl[i] = 0.01f * (lane + i);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    float4 afr[8];
#pragma unroll
    for (int m = 0; m < 8; ++m)
      afr[m] = LDSA ? *reinterpret_cast<const float4*>(arow + 16 * m) : make_float4(0.5f + lane, 0.25f, 0.125f, 1.f + m);
    if (LDSA) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int hf = 0; hf < (ORDER ? 2 : 1); ++hf)
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const float av[4] = {afr[m].x, afr[m].y, afr[m].z, afr[m].w};
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
          for (int b = 0; b < (ORDER ? 4 : 8); ++b) {
            const int blk = ORDER ? 2 * b + hf : b;
            acc[blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[tt], wf[(blk * 32 + 4 * m + tt) % NW], acc[blk], 0, 0, 0);
            // FILL independent VALU instructions per MFMA gap (8 rotating chains): FILL > 0 plain fma, FILL < 0 v_exp_f32
#pragma unroll
            for (int f = 0; f < (FILL < 0 ? -FILL : FILL); ++f) {
              const int j = (b * 3 + f) & 7;
              fl[j] = FILL < 0 ? __builtin_amdgcn_exp2f(fl[j]) : __builtin_fmaf(fl[j], 0.999f, 0.001f);
            }
            if (FILL != 0) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x002, FILL < 0 ? -FILL : FILL, 0);
            }
          }
      }
  }
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::"v"(acc[0][0]), "v"(acc[7][0]));
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int a = 0; a < 8; ++a) s += acc[a][0] + acc[a][1] + acc[a][2] + acc[a][3] + fl[a];
  out[(size_t)blockIdx.x * 256 + tid] = s;
  if (lane == 0) cyc[blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
}

template <int NW, bool LDSA, int ORDER, int FILL = 0>
void run(const char* name, int grid, const float* w, float* out, unsigned long long* cyc) {
  const int iters = 400;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipLaunchKernelGGL((k<NW, LDSA, ORDER, FILL>), dim3(grid), dim3(256), 0, 0, w, out, cyc, 10);
  hipEventRecord(a);
  hipLaunchKernelGGL((k<NW, LDSA, ORDER, FILL>), dim3(grid), dim3(256), 0, 0, w, out, cyc, iters);
  hipEventRecord(b);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  std::vector<unsigned long long> h(grid * 4);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double mean = 0;
  for (auto v : h) mean += (double)v;
  mean /= h.size();
  printf("%-44s grid %3d: %.2f cycles/MFMA  %.3f ms  %.1f TFLOP/s  clock~%.2f GHz\n", name, grid, mean / (iters * 256.0), ms,
         (double)grid * 4 * iters * 256.0 * 2048.0 / (ms * 1e-3) / 1e12, mean / (ms * 1e-3) / 1e9);
}

int main() {
  float *w, *out;
  unsigned long long* cyc;
  hipMalloc(&w, 256 * 64 * 4);
  hipMalloc(&out, 256 * 256 * 4);
  hipMalloc(&cyc, 256 * 4 * 8);
  std::vector<float> hw(256 * 64);
  for (size_t i = 0; i < hw.size(); ++i) hw[i] = 0.001f * (float)((i * 7919) % 1000) - 0.5f;
  hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
  for (int grid : {142}) {
    run<256, false, 0>("W=256 regs, A in regs, k-major", grid, w, out, cyc);
    run<64, false, 0>("W=64 regs, A in regs, k-major", grid, w, out, cyc);
    run<256, true, 0>("W=256 regs, A from LDS, k-major", grid, w, out, cyc);
    run<256, true, 1>("W=256 regs, A from LDS, half-major", grid, w, out, cyc);
    run<64, true, 1>("W=64 regs, A from LDS, half-major", grid, w, out, cyc);
    run<256, true, 0, 1>("W=256, LDS A, k-major + 1 fma/gap", grid, w, out, cyc);
    run<256, true, 0, 3>("W=256, LDS A, k-major + 3 fma/gap", grid, w, out, cyc);
    run<256, true, 0, 6>("W=256, LDS A, k-major + 6 fma/gap", grid, w, out, cyc);
    run<256, true, 0, -1>("W=256, LDS A, k-major + 1 exp/gap", grid, w, out, cyc);
    run<256, true, 0, -2>("W=256, LDS A, k-major + 2 exp/gap", grid, w, out, cyc);
    run<256, true, 0, -3>("W=256, LDS A, k-major + 3 exp/gap", grid, w, out, cyc);
  }
  return 0;
}
