// fcln.h -- launcher of fcln.hip: a Linear layer with its LayerNorm and residual on 16-token tiles, several workgroups per CU.
//
//   pre_res = 0:  out = LayerNorm(A W^T + bias) * gamma + beta + res      DPRNN blocks: fc (256 -> 64), dprnn.py:41-45, 83-87
//   pre_res = 1:  out = LayerNorm(A W^T + bias + res) * gamma + beta      DPTN: out-projection + LN1 (dptn.py:46-47), FFN + LN2 (:50-51)
//   relu_a      :  A is read through ReLU (the DPTN ffn = ReLU -> Linear on the raw h rows the training tape keeps)
//   zn / rstd   :  (both or neither) the normalised rows [M][nout] and 1/sigma [M] for the LayerNorm backward (training tape)
//
// A [M][kin], W [nout][kin] (nn.Linear layout), res / out / zn [M][nout].  Shapes taken: (kin, nout) = (256, 64) inference form,
// (256, 128) and (128, 128) with the tape.  Returns a hipError_t as int; hipErrorInvalidValue for anything else (the caller then
// uses the GEMM engine).
#pragma once
#include <cstddef>
#include <cstdint>

struct FclnArgs {
  const float* A;
  const float* W;
  const float* bias;
  const float* gamma;
  const float* beta;
  const float* res;
  float* out;
  float* zn = nullptr;
  float* rstd = nullptr;
  int64_t M = 0;
  int kin = 0, nout = 0;
  bool pre_res = false, relu_a = false;
  int nbuf = 2;          // token tiles in flight per workgroup + 1 (nout = 64 only: 2 = three workgroups per CU, 3 = two)
};
int fcln_launch(void* stream, const FclnArgs& a, int num_cus);
