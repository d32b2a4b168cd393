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
  const float* wf;     // fp32 kernel: PACKED ffn.1 segment of that path (attn_pack_launch)
  const float* bf;
  const float* g2;
  const float* b2;
};
// The fp32 kernel reads its weights from a PACKED copy in MFMA-fragment order (one wave instruction = 1 KiB contiguous;
// straight from the nn.Module tensors a fragment load touches 32 cache lines): per path
//   [ in_proj: sel(q,k,v) x head x m(16) x lane(64) x 4 | out_proj: head x jt(4) x j(4) x lane x 4 | ffn.1: head x m(32) x lane x 4 ]
// written by attn_pack_launch from the CURRENT weights (call it once per forward: an optimizer may have stepped).
constexpr int ATTN_PACK_IN = 3 * 128 * 128, ATTN_PACK_OUT = 128 * 128, ATTN_PACK_FFN = 128 * 256;
constexpr int ATTN_PACK_FLOATS = ATTN_PACK_IN + ATTN_PACK_OUT + ATTN_PACK_FFN;
struct AttnPackSrc {
  const float* w_in;   // mha.in_proj_weight [384][128]
  const float* w_o;    // mha.out_proj.weight [128][128]
  const float* w_f;    // ffn.1.weight [128][256]; null for a path with one LSTM direction ([128][128]: no prologue there)
};
constexpr int ATTN_PACK_MAX_PATHS = 32;        // per launch
int attn_pack_launch(void* stream, const AttnPackSrc* src, int npaths, float* dst /* [npaths][ATTN_PACK_FLOATS] */);
// wpack: this path's packed weights (fp32 variant; the split variant reads w_in / w_o);  pro->wf: the PREVIOUS path's
// packed ffn.1 segment (its wpack + ATTN_PACK_IN + ATTN_PACK_OUT)
int attn_block_launch(void* stream, const float* x, const float* w_in, const float* b_in, const float* w_o, const float* b_o,
                      const float* gamma, const float* beta, float* y1, const SeqGeom& g, bool split = false,
                      const AttnFfnPrologue* pro = nullptr, const float* wpack = nullptr);

// attn_block2.hip (round 5): the same block with both LayerNorms in fragment space and the prologue's h rows by LDS-DMA; fp32,
// packed weights only.  Same arguments, same results up to the grouping of the LayerNorm sums.
size_t attn_block2_lds_bytes(int nkb, bool pro);
int attn_block2_launch(void* stream, const float* x, const float* b_in, const float* b_o, const float* gamma, const float* beta,
                       float* y1, const SeqGeom& g, const AttnFfnPrologue* pro, const float* wpack);
