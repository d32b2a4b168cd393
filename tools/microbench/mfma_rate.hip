// Micro-benchmark: sustained rate of v_mfma_f32_32x32x2_f32 streams shaped like the LSTM / GEMM-engine inner loops.
// build: hipcc -O3 --offload-arch=gfx950 -o mfma_rate mfma_rate.hip ; run: ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NW, bool LDSA, int NACC>
__global__ __launch_bounds__(256) void k(const float* __restrict__ w, float* __restrict__ out,
                                          unsigned long long* __restrict__ cyc, int iters) {
  __shared__ float As[32 * 132];
  const int tid = threadIdx.x, lane = tid & 63;
  float wf[NW];
#pragma unroll
  for (int i = 0; i < NW; ++i) wf[i] = w[(size_t)i * 64 + lane];
  for (int i = tid; i < 32 * 132; i += 256) As[i] = 0.001f * (i % 17);
  __syncthreads();
  f32x16 acc[NACC];
#pragma unroll
  for (int a = 0; a < NACC; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  float areg[4] = {0.5f + lane, 0.25f, 0.125f, 1.f};
  const float* arow = As + (lane & 31) * 132 + 4 * (lane >> 5);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      float av[4] = {areg[0], areg[1], areg[2], areg[3]};
      if (LDSA) {
        const float4 a = *reinterpret_cast<const float4*>(arow + 8 * m);
        av[0] = a.x; av[1] = a.y; av[2] = a.z; av[3] = a.w;
      }
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int widx = (g * 64 + 4 * m + tt) % NW;
          acc[g % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tt], wf[widx], acc[g % NACC], 0, 0, 0);
        }
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::"v"(acc[0][0]));
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int a = 0; a < NACC; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[(size_t)blockIdx.x * 256 + tid] = s;
  if (lane == 0) cyc[blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
}

template <int NW, bool LDSA, int NACC>
void run(const char* name, int grid, const float* w, float* out, unsigned long long* cyc) {
  const int iters = 200;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipLaunchKernelGGL((k<NW, LDSA, NACC>), dim3(grid), dim3(256), 0, 0, w, out, cyc, 10);
  hipEventRecord(a);
  hipLaunchKernelGGL((k<NW, LDSA, NACC>), dim3(grid), dim3(256), 0, 0, w, out, cyc, iters);
  hipEventRecord(b);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  std::vector<unsigned long long> h(grid * 4);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double mean = 0;
  for (auto v : h) mean += (double)v;
  mean /= h.size();
  const double per = mean / (iters * 256.0);
  const double tf = (double)grid * 4 * iters * 256.0 * 4096.0 / (ms * 1e-3) / 1e12;
  printf("%-34s grid %3d: %.1f cycles/MFMA (s_memtime)  %.3f ms  %.1f TFLOP/s  clock~%.2f GHz\n", name, grid, per, ms, tf,
         mean / (ms * 1e-3) / 1e9);
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
  for (int grid : {256, 142}) {
    run<256, false, 4>("W=256 regs, A in regs, 4 acc", grid, w, out, cyc);
    run<128, false, 4>("W=128 regs, A in regs, 4 acc", grid, w, out, cyc);
    run<64, false, 4>("W=64 regs, A in regs, 4 acc", grid, w, out, cyc);
    run<256, true, 4>("W=256 regs, A from LDS, 4 acc", grid, w, out, cyc);
    run<64, true, 4>("W=64 regs, A from LDS, 4 acc", grid, w, out, cyc);
    run<64, false, 1>("W=64 regs, A in regs, 1 acc", grid, w, out, cyc);
    run<64, false, 2>("W=64 regs, A in regs, 2 acc", grid, w, out, cyc);
  }
  return 0;
}
