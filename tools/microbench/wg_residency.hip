// wg_residency.hip -- how many workgroups of the training attention kernels' footprint (320 threads, 48.6 KB of LDS, ~114 VGPRs) does a
// CU really hold over a launch of many SHORT workgroups?  (round 5: the counters of attention_bwd_kernel say 1.8 waves per SIMD on
// average where registers and LDS allow 3.75.)  Every workgroup spins for `spin` iterations of dependent FMAs (~58 k cycles, the
// lifetime of a phase-A workgroup) and records the s_memrealtime at its start and end; from the intervals: the average number of
// workgroups alive per CU and the launch's duration against the ideal (items x lifetime / (CUs x 3)).
//   hipcc --offload-arch=gfx950 -O2 -o wg_residency wg_residency.hip && ./wg_residency
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(320) void probe(unsigned long long* t, int spin, int items_per_wg, float* sink) {
  extern __shared__ float lds[];
  float x = (float)threadIdx.x, y = 1.0001f;
  for (int it = 0; it < items_per_wg; ++it) {
    const int item = blockIdx.x + it * gridDim.x;
    unsigned long long t0 = 0;
    if (threadIdx.x == 0) asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
    lds[threadIdx.x] = x;
    __syncthreads();
    for (int i = 0; i < spin; ++i) x = x * y + 0.5f;      // dependent chain: ~4-8 cycles per iteration per wave
    x += lds[(threadIdx.x + 1) % 320];
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned long long t1;
      asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
      t[2 * item] = t0;
      t[2 * item + 1] = t1;
    }
  }
  if (x == 1234.5f) *sink = x;
}
int main() {
  const int items = 4512 * 2, lds = 49664;
  unsigned long long* d;
  float* sink;
  hipMalloc(&d, 2 * items * sizeof(unsigned long long));
  hipMalloc(&sink, 4);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  std::vector<unsigned long long> h(2 * items);
  for (int spin : {1000, 300})
  for (int per_wg : {1, 12}) {
    const int grid = items / per_wg;
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(probe, dim3(grid), dim3(320), lds, 0, d, spin, per_wg, sink);
      hipDeviceSynchronize();
    }
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long lo = ~0ull, hi = 0;
    double alive = 0;
    for (int i = 0; i < items; ++i) {
      lo = std::min(lo, h[2 * i]);
      hi = std::max(hi, h[2 * i + 1]);
      alive += (double)(h[2 * i + 1] - h[2 * i]);
    }
    const double dur = (double)(hi - lo), life = alive / items;      // 100 MHz ticks
    printf("%5d workgroups x %2d items: launch %.1f us, item lifetime %.2f us, items alive on average %.1f = %.2f per CU; ideal launch at 3 per CU %.1f us\n",
           grid, per_wg, dur / 100.0, life / 100.0, alive / dur, alive / dur / 256.0, items * life / 100.0 / (256 * 3));
  }
  return 0;
}
