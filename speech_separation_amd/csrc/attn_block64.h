// attn_block64.h -- launcher of the fused attention half of a TransformerDPRNN for num_features = 64 (attn_block64.hip:
// optional FFN/LayerNorm-2 prologue of the previous path, in-projection, 4-head attention with head dimension 16,
// out-projection, residual and LayerNorm 1 in one kernel on v_mfma_f32_16x16x4_f32; inference, sequences of at most
// ATTN_BLOCK_MAX_LEN positions).  Weights are read in place from the nn.Module tensors (no packed copy).  Returns a
// hipError_t as int.
#pragma once
#include "attn_block.h"

size_t attn_block64_lds_bytes(int nb);
// pro (optional): hc = ReLU(h) [M][256] of the PREVIOUS path (both directions), wf = its ffn.1.weight [64][256] as stored,
// bf / g2 / b2 its ffn.1.bias and ln2 weight / bias; y1 then holds that path's LayerNorm-1 output and is updated in place.
int attn_block64_launch(void* stream, const float* x, const float* w_in, const float* b_in, const float* w_o, const float* b_o,
                        const float* gamma, const float* beta, float* y1, const SeqGeom& g, const AttnFfnPrologue* pro = nullptr);
