// attn_block.h -- launcher of the fused attention half of a TransformerDPRNN (attn_block.hip: in-projection, 4-head
// attention, out-projection, residual and LayerNorm 1 in one kernel; inference, num_features = 128, sequences of at
// most ATTN_BLOCK_MAX_LEN positions).  Returns a hipError_t as int.
#pragma once
#include "common.h"

constexpr int ATTN_BLOCK_MAX_LEN = 160;
size_t attn_block_lds_bytes(int nkb);
// Optional prologue: the block's input rows are computed in the kernel as the FFN half of the PREVIOUS TransformerDPRNN,
// x = LayerNorm2(hc W_f^T + b_f + y1) with hc = ReLU(h) [M][256] of that path and y1 its LayerNorm-1 output -- read from
// the SAME buffer the block writes its own y1 to (in place); `x` is then unused.  fp32 variant only.
struct AttnFfnPrologue {
  const float* hc;
  const float* wf;
  const float* bf;
  const float* g2;
  const float* b2;
};
int attn_block_launch(void* stream, const float* x, const float* w_in, const float* b_in, const float* w_o, const float* b_o,
                      const float* gamma, const float* beta, float* y1, const SeqGeom& g, bool split = false,
                      const AttnFfnPrologue* pro = nullptr);
