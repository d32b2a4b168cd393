// wave_simd.hip -- which SIMD of its CU does wave i of a workgroup land on?  (round 5: the training attention kernels run one
// workgroup = 5 waves = one wave per 32-query block; if wave i always goes to SIMD i % 4, one SIMD of four carries two waves of every
// workgroup and the kernel cannot be more than 62 % busy.)   hipcc --offload-arch=gfx950 -O2 -o wave_simd wave_simd.hip && ./wave_simd
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned* hist, int spin) {
  const int wave = threadIdx.x >> 6;
  unsigned hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  const unsigned simd = (hw >> 4) & 3u;
  // keep the wave resident for a while so that later workgroups are placed beside running ones, as in a real launch
  float x = (float)threadIdx.x;
  for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f;
  if ((threadIdx.x & 63) == 0) atomicAdd(&hist[wave * 4 + simd], 1u);
  if (x == 12345.678f) hist[63] = 1;
}
int main() {
  unsigned* d;
  hipMalloc(&d, 64 * sizeof(unsigned));
  for (int waves : {4, 5, 8, 10}) {
    hipMemset(d, 0, 64 * sizeof(unsigned));
    hipLaunchKernelGGL(probe, dim3(9024), dim3(64 * waves), 0, 0, d, 20000);
    hipDeviceSynchronize();
    unsigned h[64];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("workgroups of %d waves: share of the workgroup's wave i on SIMD 0..3\n", waves);
    unsigned per_simd[4] = {0, 0, 0, 0};
    for (int w = 0; w < waves; ++w) {
      const unsigned tot = h[w * 4] + h[w * 4 + 1] + h[w * 4 + 2] + h[w * 4 + 3];
      printf("  wave %2d: %5.2f %5.2f %5.2f %5.2f\n", w, h[w * 4] / (double)tot, h[w * 4 + 1] / (double)tot, h[w * 4 + 2] / (double)tot, h[w * 4 + 3] / (double)tot);
      for (int s = 0; s < 4; ++s) per_simd[s] += h[w * 4 + s];
    }
    const double all = per_simd[0] + per_simd[1] + per_simd[2] + per_simd[3];
    printf("  all waves: %5.3f %5.3f %5.3f %5.3f  (balanced = 0.250 each)\n", per_simd[0] / all, per_simd[1] / all, per_simd[2] / all, per_simd[3] / all);
  }
  return 0;
}
