// fcln64.h -- launcher of fcln64.hip: out[M][64] = LayerNorm(A[M][kin] W[64][kin]^T + bias) * gamma + beta + res[M][64]
// (DPRNN blocks, dprnn.py:40-46, 83-87), kin = 256.  Returns a hipError_t as int; hipErrorInvalidValue for shapes it does not
// take (the caller then uses the GEMM engine).
#pragma once
#include <cstddef>
#include <cstdint>
int fcln64_launch(void* stream, int kin, const float* A, const float* W, const float* bias, const float* gamma, const float* beta,
                  const float* res, float* out, int64_t M, int num_cus, int nbuf = 2);
// nbuf: token tiles in flight per workgroup + 1 (2: three workgroups per CU; 3: two)
