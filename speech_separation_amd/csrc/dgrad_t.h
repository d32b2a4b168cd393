// dgrad_t.h -- launcher of dgrad_t.hip: the data gradient of a wide layer into a 128-feature stream,
//
//     out[M][128] = addend[M][128] + A[M][0:kin] W[kin][128]          kin = 512: d x = d_out + dP W_ih, the LSTM's input side
//                                                                       (src/model/dptn.py:48, backward of nn.LSTM's W_ih product);
//                                                                       kin = 384: d x = dz + d qkv W_in (dptn.py:46, the
//                                                                       in-projection of nn.MultiheadAttention)
//
// A rows are `lda` floats apart (the column slice is folded into the pointer); W is the FRAGMENT-ORDER copy that
// dgrad_t_pack_launch makes of the forward weight as nn.LSTM / nn.Linear keep it ([out features = kin here][in features = 128],
// row-major); `out` may alias `addend` (each element is read and written by the same lane).  `queue`: a zeroed device counter for dynamic tile tickets, or null for the static grid-stride
// order (bit-reproducible association of nothing here -- every output element is one fixed MFMA chain either way).
// Returns a hipError_t as int; hipErrorInvalidValue for shapes it does not take (the caller then uses the GEMM engine).
#pragma once
#include <cstdint>

struct DgradTArgs {
  const float* A;
  int lda = 0;
  const float* W;
  const float* addend;
  float* out;
  int64_t M = 0;
  int kin = 0;
  unsigned* queue = nullptr;
};
int dgrad_t_launch(void* stream, const DgradTArgs& a, int num_cus);
// fragment-order copies of n (<= DGRAD_PACK_MAX) row-major [kin][128] weights in ONE launch: srcs[i] (null: skipped) -> dst + dst_off[i]
// (kin x 128 floats each)
constexpr int DGRAD_PACK_MAX = 24;
int dgrad_t_pack_launch(void* stream, const float* const* srcs, const long long* dst_off, int n, int kin, float* dst);
