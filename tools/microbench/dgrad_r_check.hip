// dgrad_r_check.hip -- csrc/dgrad_r.hip alone: data gradient, weight gradient (partial tiles summed and put back into row-major
// order here) and bias gradient against fp64 host sums, run-to-run bit equality in the static order, and the launch time.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ../../speech_separation_amd/csrc -o dgrad_r_check dgrad_r_check.hip \
//         ../../speech_separation_amd/csrc/dgrad_r.hip && ./dgrad_r_check [loop seconds]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "dgrad_r.h"

#define CK(x)                                                        \
  do {                                                               \
    hipError_t e_ = (x);                                             \
    if (e_ != hipSuccess) {                                          \
      printf("%s: %s\n", #x, hipGetErrorString(e_));                 \
      return 1;                                                      \
    }                                                                \
  } while (0)

static float frand(unsigned& s) {
  s = s * 1664525u + 1013904223u;
  return ((s >> 8) & 0xffff) / 32768.0f - 1.0f;
}
static int row32(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

int main(int argc, char** argv) {
  const int K = 128;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount, max_slabs = 256;
  const bool loop = argc > 2 && !strcmp(argv[1], "loop");
  const bool one = argc > 3 && !strcmp(argv[1], "case");      // case <nout> <M>: one shape only
  setvbuf(stdout, nullptr, _IONBF, 0);
  int bad = 0;
  for (int nout : {128, 256})
    for (long M : {1L, 33L, 4097L, 169200L}) {
      if (loop && M != 169200) continue;
      if (one && (nout != atoi(argv[2]) || M != atol(argv[3]))) continue;
      printf("nout=%d M=%ld ...\n", nout, M);
      const bool gate = nout == 256;
      std::vector<float> hA((size_t)M * K), hW((size_t)K * nout), hX((size_t)M * nout);
      unsigned s = 99u + (unsigned)M + nout;
      for (auto& v : hA) v = frand(s);
      for (auto& v : hW) v = frand(s) * 0.1f;
      for (auto& v : hX) v = frand(s);
      float *dA, *dW, *dWp, *dX, *dout, *dslab, *dcol;
      unsigned* dq;
      CK(hipMalloc(&dA, hA.size() * 4));
      CK(hipMalloc(&dW, hW.size() * 4));
      CK(hipMalloc(&dWp, hW.size() * 4));
      CK(hipMalloc(&dX, hX.size() * 4));
      CK(hipMalloc(&dout, hX.size() * 4));
      CK(hipMalloc(&dslab, (size_t)max_slabs * K * nout * 4));
      CK(hipMalloc(&dcol, (size_t)max_slabs * K * 4));
      CK(hipMalloc(&dq, 64));
      CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
      CK(hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
      CK(hipMemcpy(dX, hX.data(), hX.size() * 4, hipMemcpyHostToDevice));
      { const float* srcs[1] = {dW}; const long long offs[1] = {0};
        if (int rc = dgrad_r_pack_launch(nullptr, srcs, offs, 1, nout, dWp)) { printf("pack rc %d\n", rc); return 1; } }
      DgradRArgs a;
      a.A = dA; a.Wpacked = dWp; a.X = dX; a.out = dout; a.M = M; a.nout = nout; a.relu_gate = gate;
      a.slab = dslab; a.colslab = dcol; a.max_slabs = max_slabs; a.queue = nullptr;
      int grid = 0;
      if (loop) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        double total = 0;
        while (total < atof(argv[2]) * 500) {
          CK(hipEventRecord(e0));
          for (int i = 0; i < 200; ++i) dgrad_r_launch(nullptr, a, cus, &grid);
          CK(hipEventRecord(e1));
          CK(hipEventSynchronize(e1));
          float ms;
          CK(hipEventElapsedTime(&ms, e0, e1));
          total += ms;
          const double flops = 2.0 * M * K * nout * 2;
          printf("nout=%d: 200 launches, %.1f us each, %.1f TFLOP/s\n", nout, ms * 5, flops / (ms * 5) * 1e-6);
          fflush(stdout);
        }
        continue;
      }
      std::vector<float> o1(hX.size()), o2(hX.size()), sl((size_t)max_slabs * K * nout), cl((size_t)max_slabs * K);
      std::vector<double> dWsum((size_t)K * nout), dbsum(K);
      std::vector<float> first_dw, first_db;
      int diff = 0;
      double e_out = 0, e_dw = 0, e_db = 0;
      for (int rep = 0; rep < 4; ++rep) {
        const bool dyn = rep == 3;
        if (dyn) CK(hipMemset(dq, 0, 64));
        a.queue = dyn ? dq : nullptr;
        CK(hipMemset(dout, 0xff, hX.size() * 4));
        CK(hipMemset(dslab, 0xff, sl.size() * 4));
        CK(hipMemset(dcol, 0xff, cl.size() * 4));
        if (int rc = dgrad_r_launch(nullptr, a, cus, &grid)) { printf("launch rc %d\n", rc); return 1; }
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(o2.data(), dout, hX.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(sl.data(), dslab, sl.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(cl.data(), dcol, cl.size() * 4, hipMemcpyDeviceToHost));
        // partial tiles -> row-major sums (fp32 in slab order, as slab_reduce_frag_kernel would, but here in double for the check)
        const int CB = nout / 32;
        std::vector<float> dw((size_t)K * nout), db(K);
        for (int w = 0; w < 4; ++w)
          for (int j = 0; j < CB; ++j)
            for (int g4 = 0; g4 < 4; ++g4)
              for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 4; ++e) {
                  double acc = 0;
                  for (int g = 0; g < grid; ++g) acc += sl[(size_t)g * K * nout + ((((size_t)w * CB + j) * 4 + g4) * 64 + lane) * 4 + e];
                  dw[(size_t)(32 * w + row32(4 * g4 + e, lane >> 5)) * nout + 32 * j + (lane & 31)] = (float)acc;
                }
        for (int k = 0; k < K; ++k) {
          double acc = 0;
          for (int g = 0; g < grid; ++g) acc += cl[(size_t)g * K + k];
          db[k] = (float)acc;
        }
        if (rep == 0) {
          o1 = o2;
          first_dw = dw;
          first_db = db;
          // fp64 references
          for (long r = 0; r < M; r += (M > 4096 ? 991 : 1))
            for (int n = 0; n < nout; ++n) {
              double acc = 0;
              for (int k = 0; k < K; ++k) acc += (double)hA[r * K + k] * hW[(size_t)k * nout + n];
              if (gate && !(hX[r * nout + n] > 0.f)) acc = 0;
              if (!(std::fabs(acc - o2[r * nout + n]) < 1e30)) e_out = 1e30;
              e_out = std::fmax(e_out, std::fabs(acc - o2[r * nout + n]));
            }
          for (int k = 0; k < K; k += (M > 4096 ? 13 : 1)) {
            double sb = 0;
            for (long r = 0; r < M; ++r) sb += hA[r * K + k];
            if (!(std::fabs(sb - db[k]) < 1e30)) e_db = 1e30;      // NaN (a column sum that was never written) is a failure, not a skip
            e_db = std::fmax(e_db, std::fabs(sb - db[k]) / (1.0 + std::sqrt((double)M)));
            for (int n = 0; n < nout; n += (M > 4096 ? 7 : 1)) {
              double acc = 0;
              for (long r = 0; r < M; ++r) {
                const float xv = gate ? std::fmax(hX[r * nout + n], 0.f) : hX[r * nout + n];
                acc += (double)hA[r * K + k] * xv;
              }
              if (!(std::fabs(acc - dw[(size_t)k * nout + n]) < 1e30)) e_dw = 1e30;
              e_dw = std::fmax(e_dw, std::fabs(acc - dw[(size_t)k * nout + n]) / (1.0 + std::sqrt((double)M)));
            }
          }
        } else {
          long nd = 0;
          for (size_t i = 0; i < o2.size(); ++i) nd += std::memcmp(&o2[i], &o1[i], 4) != 0;
          long nw = 0;
          if (!dyn) {      // the static order fixes which workgroup sums which tiles
            for (size_t i = 0; i < dw.size(); ++i) nw += std::memcmp(&dw[i], &first_dw[i], 4) != 0;
            for (int k = 0; k < K; ++k) nw += std::memcmp(&db[k], &first_db[k], 4) != 0;
          } else {
            for (size_t i = 0; i < dw.size(); ++i) nw += std::fabs(dw[i] - first_dw[i]) > 1e-3 * (1 + std::sqrt((double)M));
          }
          if (nd || nw) { ++diff; printf("  nout=%d M=%ld rep %d (dyn %d): %ld outputs, %ld gradient elements differ\n", nout, M, rep, (int)dyn, nd, nw); }
        }
      }
      printf("nout=%d M=%ld grid %d: max |err| out %.3e, dW %.3e, db %.3e (the latter two per 1 + sqrt(M)); %d of 3 repeats differ\n", nout, M, grid,
             e_out, e_dw, e_db, diff);
      if (e_out > 2e-4 || e_dw > 2e-5 || e_db > 2e-5 || diff) ++bad;
      hipFree(dA); hipFree(dW); hipFree(dWp); hipFree(dX); hipFree(dout); hipFree(dslab); hipFree(dcol); hipFree(dq);
    }
  if (!loop) printf(bad ? "FAILED\n" : "ok\n");
  return bad ? 1 : 0;
}
