// fcln.h -- launcher of fcln.hip: a Linear layer with its LayerNorm and residual on 16-token tiles, several workgroups per CU.
//
//   pre_res = 0:  out = LayerNorm(A W^T + bias) * gamma + beta + res      DPRNN blocks: fc (256 -> 64), dprnn.py:41-45, 83-87
//   pre_res = 1:  out = LayerNorm(A W^T + bias + res) * gamma + beta      DPTN: out-projection + LN1 (dptn.py:46-47), FFN + LN2 (:50-51)
//   layernorm=0:  out = act(A) W^T + bias                                   separation conv: PReLU -> Conv2d(N, 2N, 1), dptn_wav.py:26-29, 47
//   act         :  A is read through nothing (0), ReLU (1: the DPTN ffn = ReLU -> Linear on the raw h rows the training tape keeps) or
//                  PReLU with the shared slope *act_w (2)
//   zn / rstd   :  (both or neither) the normalised rows [M][nout] and 1/sigma [M] for the LayerNorm backward (training tape)
//
// A [M][kin], W [nout][kin] (nn.Linear layout), res / out / zn [M][nout].  Shapes taken (kin -> nout): 256 -> 64 (both residual
// orders), 256 -> 128 (with / without ReLU + tape), 128 -> 128 with the tape, and the plain forms 128 -> 256, 64 -> 128 behind
// PReLU.  Returns a hipError_t as int; hipErrorInvalidValue for anything else (the caller then uses the GEMM engine).
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
  bool pre_res = false;
  int act = 0;
  int nbuf = 2;          // token tiles in flight per workgroup + 1 (256 -> 64 without pre_res only: 2 = three workgroups per CU, 3 = two)
  bool layernorm = true;
  const float* act_w = nullptr;
};
int fcln_launch(void* stream, const FclnArgs& a, int num_cus);
