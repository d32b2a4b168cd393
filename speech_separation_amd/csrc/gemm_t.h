// gemm_t.h -- launcher of gemm_t.hip: out[M][nout] = A[M][128] W[nout][128]^T + bias for a wide output (nout = 384: the
// in-projection of nn.MultiheadAttention in the training forward, dptn.py:16-21, 46).  A and out dense; Wpacked = the fragment-order
// copy gemm_pack_rows_kernel (gemm_ws.h) makes of the row-major weight; queue: zeroed ticket counter or null (static tile order).
// Every output element is one fixed MFMA chain started from its bias, whichever workgroup takes the tile.
// Returns a hipError_t as int; hipErrorInvalidValue for shapes it does not take (the caller then uses the GEMM engine).
#pragma once
#include <cstdint>

struct GemmTArgs {
  const float* A;
  const float* Wpacked;
  const float* bias;
  float* out;
  int64_t M = 0;
  int nout = 0;
  unsigned* queue = nullptr;
};
int gemm_t_launch(void* stream, const GemmTArgs& a, int num_cus);
