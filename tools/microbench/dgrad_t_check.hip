// dgrad_t_check.hip -- csrc/dgrad_t.hip alone: results against an fp64 host product on sampled rows, run-to-run bit equality
// (static and ticket order, in place and out of place), and the launch time with HIP events.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -I ../../speech_separation_amd/csrc \
//         -o dgrad_t_check dgrad_t_check.hip ../../speech_separation_amd/csrc/dgrad_t.hip && ./dgrad_t_check
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "dgrad_t.h"
#ifdef DGRAD_T_STAMPS
int dgrad_t_debug_stamps(unsigned long long* out, int reset);
#endif

#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      printf("%s: %s\n", #x, hipGetErrorString(e_));                                \
      return 1;                                                                     \
    }                                                                               \
  } while (0)

static float frand(unsigned& s) {
  s = s * 1664525u + 1013904223u;
  return ((s >> 8) & 0xffff) / 32768.0f - 1.0f;
}

int main(int argc, char** argv) {
  const int N = 128, lda = 1024;
  int cus = 256;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  cus = prop.multiProcessorCount;
  if (argc > 2 && !strcmp(argv[1], "loop")) {      // keep the big launch running for argv[2] seconds (clock / power read-out beside it)
    const int K = 512;
    const long M = 169200;
    float *dA, *dW, *dWp, *dadd, *dout;
    CK(hipMalloc(&dA, (size_t)M * lda * 4));
    CK(hipMalloc(&dW, (size_t)K * N * 4));
    CK(hipMalloc(&dWp, (size_t)K * N * 4));
    CK(hipMalloc(&dadd, (size_t)M * N * 4));
    CK(hipMalloc(&dout, (size_t)M * N * 4));
    std::vector<float> h((size_t)M * lda);
    unsigned s = 7u;
    for (auto& v : h) v = frand(s);
    CK(hipMemcpy(dA, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dadd, h.data(), (size_t)M * N * 4, hipMemcpyHostToDevice));
    for (size_t i = 0; i < (size_t)K * N; ++i) h[i] *= 0.05f;
    CK(hipMemcpy(dW, h.data(), (size_t)K * N * 4, hipMemcpyHostToDevice));
    { const float* srcs[1] = {dW}; const long long offs[1] = {0}; dgrad_t_pack_launch(nullptr, srcs, offs, 1, K, dWp); }
    DgradTArgs a;
    a.A = dA; a.lda = lda; a.W = dWp; a.addend = dadd; a.out = dout; a.M = M; a.kin = K; a.queue = nullptr;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const double secs = atof(argv[2]);
    double total = 0;
    while (total < secs * 1e3) {
      CK(hipEventRecord(e0));
      for (int i = 0; i < 200; ++i) dgrad_t_launch(nullptr, a, cus);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      total += ms;
      printf("200 launches: %.1f us each\n", ms * 5);
      fflush(stdout);
    }
    return 0;
  }
  const long Ms[] = {1, 31, 32, 33, 4097, 169200};
  int bad = 0;
  for (int K : {512, 384})
  for (long M : Ms) {
    std::vector<float> hA((size_t)M * lda), hW((size_t)K * N), hadd((size_t)M * N);
    unsigned s = 12345u + (unsigned)M;
    for (auto& v : hA) v = frand(s);
    for (auto& v : hW) v = frand(s) * 0.05f;
    for (auto& v : hadd) v = frand(s);
    float *dA, *dW, *dWp, *dadd, *dout, *dout2;
    unsigned* dq;
    CK(hipMalloc(&dA, hA.size() * 4));
    CK(hipMalloc(&dW, hW.size() * 4));
    CK(hipMalloc(&dadd, hadd.size() * 4));
    CK(hipMalloc(&dout, hadd.size() * 4));
    CK(hipMalloc(&dout2, hadd.size() * 4));
    CK(hipMalloc(&dq, 64));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dWp, hW.size() * 4));
    { const float* srcs[1] = {dW}; const long long offs[1] = {0};
      if (int rc = dgrad_t_pack_launch(nullptr, srcs, offs, 1, K, dWp)) { printf("pack rc %d\n", rc); return 1; } }
    CK(hipMemcpy(dadd, hadd.data(), hadd.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> ref_out(hadd.size()), got(hadd.size());
    for (int col0 : {0, 512 - (K == 384 ? 128 : 0) * 0}) {
      DgradTArgs a;
      a.A = dA + col0; a.lda = lda; a.W = dWp; a.addend = dadd; a.out = dout; a.M = M; a.kin = K; a.queue = nullptr;
      CK(hipMemset(dout, 0xff, hadd.size() * 4));
      if (int rc = dgrad_t_launch(nullptr, a, cus)) { printf("launch rc %d\n", rc); return 1; }
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(ref_out.data(), dout, hadd.size() * 4, hipMemcpyDeviceToHost));
      // fp64 host product on sampled rows
      double worst = 0;
      for (long r = 0; r < M; r += (M > 4096 ? 997 : 1)) {
        for (int n = 0; n < N; ++n) {
          double acc = hadd[r * N + n];
          for (int k = 0; k < K; ++k) acc += (double)hA[r * lda + col0 + k] * hW[k * N + n];
          worst = std::fmax(worst, std::fabs(acc - ref_out[r * N + n]));
        }
      }
      if (M == 33 && col0 == 0 && getenv("DGT_ROWS")) {
        for (long r = 0; r < M; ++r) {
          double w2 = 0;
          int wn = -1;
          for (int n = 0; n < N; ++n) {
            double acc = hadd[r * N + n];
            for (int k = 0; k < K; ++k) acc += (double)hA[r * lda + col0 + k] * hW[k * N + n];
            if (std::fabs(acc - ref_out[r * N + n]) > w2) { w2 = std::fabs(acc - ref_out[r * N + n]); wn = n; }
          }
          printf("   row %ld: max err %.3e at col %d", r, w2, wn);
          if (w2 > 1e-3) {      // whose A row (and whose addend row) does the stored value belong to?
            for (long ra = 0; ra < M; ++ra)
              for (long rd = 0; rd < M; ++rd) {
                double e2 = 0;
                for (int n = 0; n < N; n += 17) {
                  double acc = hadd[rd * N + n];
                  for (int k = 0; k < K; ++k) acc += (double)hA[ra * lda + col0 + k] * hW[k * N + n];
                  e2 = std::fmax(e2, std::fabs(acc - ref_out[r * N + n]));
                }
                if (e2 < 1e-4) printf("  <- A row %ld, addend row %ld", ra, rd);
              }
          }
          printf("\n");
          if (r == 8 || r == 9 || r == 24) {
            for (int n = 0; n < N; ++n) {
              double acc = hadd[r * N + n], p0 = 0, p1 = 0;
              for (int k = 0; k < K; ++k) {
                acc += (double)hA[r * lda + col0 + k] * hW[k * N + n];
                (k < 256 ? p0 : p1) += (double)hA[r * lda + col0 + k] * hW[k * N + n];
              }
              printf("%s%+.2e", n % 16 == 0 ? "\n      " : " ", ref_out[r * N + n] - acc);
            }
            printf("\n");
          }
        }
      }
      // last row explicitly
      {
        const long r = M - 1;
        for (int n = 0; n < N; ++n) {
          double acc = hadd[r * N + n];
          for (int k = 0; k < K; ++k) acc += (double)hA[r * lda + col0 + k] * hW[k * N + n];
          worst = std::fmax(worst, std::fabs(acc - ref_out[r * N + n]));
        }
      }
      int diff_runs = 0;
      for (int rep = 0; rep < 6; ++rep) {
        const bool dyn = rep & 1, inplace = rep >= 4;
        if (dyn) CK(hipMemset(dq, 0, 64));
        a.queue = dyn ? dq : nullptr;
        if (inplace) {
          CK(hipMemcpy(dout2, dadd, hadd.size() * 4, hipMemcpyDeviceToDevice));
          a.addend = dout2; a.out = dout2;
        } else {
          CK(hipMemset(dout2, 0xff, hadd.size() * 4));
          a.addend = dadd; a.out = dout2;
        }
        if (int rc = dgrad_t_launch(nullptr, a, cus)) { printf("launch rc %d\n", rc); return 1; }
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(got.data(), dout2, hadd.size() * 4, hipMemcpyDeviceToHost));
        long nd = 0, first = -1;
        for (size_t i = 0; i < got.size(); ++i)
          if (std::memcmp(&got[i], &ref_out[i], 4) != 0) { if (first < 0) first = (long)i; ++nd; }
        if (nd) { ++diff_runs; printf("  M=%ld col0=%d rep %d (dyn %d inplace %d): %ld elements differ, first at row %ld col %ld\n", M, col0, rep, (int)dyn, (int)inplace, nd, first / N, first % N); }
      }
      printf("K=%d M=%ld col0=%d: max |err| vs fp64 %.3e, %d of 6 repeats differ\n", K, M, col0, worst, diff_runs);
      if (worst > 2e-4 || diff_runs) ++bad;
    }
    if (M == 169200) {
      DgradTArgs a;
      a.A = dA; a.lda = lda; a.W = dWp; a.addend = dadd; a.out = dout; a.M = M; a.kin = K; a.queue = nullptr;
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0));
      CK(hipEventCreate(&e1));
      for (int mode = 0; mode < 2; ++mode) {
        for (int i = 0; i < 3; ++i) dgrad_t_launch(nullptr, a, cus);
        CK(hipEventRecord(e0));
        const int reps = 20;
        for (int i = 0; i < reps; ++i) {
          if (mode) { hipMemsetAsync(dq, 0, 64); a.queue = dq; }
          dgrad_t_launch(nullptr, a, cus);
        }
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1000 / reps, flops = 2.0 * M * K * N;
        printf("K=%d M=%ld %s: %.1f us per launch, %.1f TFLOP/s (fp32 MFMA peak 157.3), A read at %.2f TB/s\n", K, M, mode ? "tickets" : "static", us,
               flops / us * 1e-6, (double)M * K * 4 / us * 1e-6);
      }
    }
#ifdef DGRAD_T_STAMPS
    if (M == 169200) {
      DgradTArgs a;
      a.A = dA; a.lda = lda; a.W = dWp; a.addend = dadd; a.out = dout; a.M = M; a.kin = K; a.queue = nullptr;
      dgrad_t_debug_stamps(nullptr, 1);
      dgrad_t_launch(nullptr, a, cus);
      unsigned long long st[8];
      dgrad_t_debug_stamps(st, 0);
      const char* nm[6] = {"prologue (weights, first rows)", "ticket + barrier", "MFMA block (+ requests)", "addend wait + epilogue", "wait for the next tile's rows", "-"};
      const double tiles = (double)((M + 31) / 32) * 4;      // wave-tiles
      printf("stamps: %llu waves, %.0f wave-tiles; s_memtime cycles\n", st[6], tiles);
      for (int i = 0; i < 5; ++i) printf("  %-34s %10.1f per wave-tile   (%10.1f per wave)\n", nm[i], st[i] / tiles, (double)st[i] / st[6]);
    }
#endif
    hipFree(dA); hipFree(dW); hipFree(dWp); hipFree(dadd); hipFree(dout); hipFree(dout2); hipFree(dq);
  }
  printf(bad ? "FAILED\n" : "ok\n");
  return bad ? 1 : 0;
}
