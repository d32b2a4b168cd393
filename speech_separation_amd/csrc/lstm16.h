// lstm16.h -- the LSTM recurrence on 16-sequence tiles (v_mfma_f32_16x16x4_f32), for launches whose sequence count
// leaves CUs idle with 32-sequence tiles.  Shared declarations; the kernel is in lstm16.hip (its own translation
// unit: it is compiled with -mllvm -amdgpu-mfma-vgpr-form, see build.py).
//
// dptnav_forward runs the batch as two chained half-batches (dptnav.hip): a half has ~1130 sequences per direction,
// i.e. 72 workgroups of 32 sequences -- 28 % of the chip for 1.4 ms, and the chain of 24 such launches IS the step
// time.  With 16-sequence tiles the same launch occupies 142 CUs for half as long (8192 MFMA cycles per step instead
// of 16384): equal CU-time, half the serial chain, and twice the achieved FLOP/s per launch.
//
// Pre-activation layout (written by EpiLstmPre16, gemm_ws.h; read by LDS-DMA in lstm16.hip):
//     PRE16[d][st16][t][w][b][lane64][4]      one (tile, step) = 4 waves x 8 KiB
//   w = wave = hidden units [32w, 32w+32);  b = 2*gate + half = the wave's MFMA block (16 units);  lane = 16*ks + i16;
//   the 4 floats of a lane are sequences 4ks..4ks+3 of the tile, unit 32w + 16*half + i16 of gate `gate`.
//   Values are PRE-SCALED by l16_gate_scale(gate), so the kernel evaluates sigmoid / tanh without a multiply.
#pragma once
#include "common.h"

constexpr int L16_H = 128;
constexpr int L16_LDH = L16_H + 8;                    // 136: conflict-free ds_read_b128 for the 16x16x4 A map
constexpr int L16_HS_FLOATS = 2 * 16 * L16_LDH;       // h double buffer
constexpr int L16_PRE_FLOATS = 4 * 8 * 256;           // [wave][block][lane*4]
constexpr size_t L16_LDS_BYTES = sizeof(float) * (L16_HS_FLOATS + L16_PRE_FLOATS);
constexpr int L16_TILE_FLOATS = 4 * L16_H * 16;       // one (tile, step) of PRE16

// scale of gate `gate`'s pre-activation rows (gate order i, f, g, o: torch.nn.LSTM):
// sigmoid(x) = 1 / (1 + 2^(-log2e x)),  tanh(x) = 2 / (1 + 2^(-2 log2e x)) - 1
DEV constexpr float l16_gate_scale(int gate) { return gate == 2 ? -2.8853900817779268f : -1.4426950408889634f; }

DEV int64_t pre16_tile_offset(int d, int st16, int t, int nst16, int len) {
  return (((int64_t)d * nst16 + st16) * len + t) * (int64_t)L16_TILE_FLOATS;
}

// Host-side launcher (lstm16.hip).  `variant`: 0 = product kernel; 1..6 = diagnostic builds with per-wave s_memtime
// stamps (1 = exact; 2.. = timing-only ablations with wrong results, see lstm16.hip); 7 = training forward (raw h,
// gates and cell states kept on the tape for lstm_bptt16.hip).  hc byte offsets are 32-bit: the caller guarantees
// (dump_row + max sequence extent) * ldh * 4 < 2^32.  Returns a hipError_t as int.
constexpr int L16_VARIANT_TRAIN = 7;
int lstm16_launch(int variant, bool relu, int nst16, int ndir, void* stream, const float* pre, const float* whh_f,
                  const float* whh_b, float* hc, int ldh, int dump_row, const SeqGeom& g, unsigned long long* stamps,
                  float* tape_gates = nullptr, float* tape_c = nullptr);

// OPT-IN split-precision variant (lstm16s.hip, option "split_bf16"): the recurrent product on bf16 MFMAs with every
// operand split into bf16 hi + lo (3 products, fp32 accumulation); same PRE16 input, same fp32 output.  Inference only.
int lstm16s_launch(bool relu, int nst16, int ndir, void* stream, const float* pre, const float* whh_f, const float* whh_b,
                   float* hc, int ldh, int dump_row, const SeqGeom& g);
// ... and of the 32-sequence-tile recurrence (same file; PRE layout of lstm.h).
int lstm32s_launch(bool relu, int nst, int ndir, void* stream, const float* pre, const float* whh_f, const float* whh_b, float* hc,
                   int ldh, int dump_row, const SeqGeom& g);

// LOW-LATENCY variant on 4-sequence tiles (lstm4.hip, v_mfma_f32_4x4x1_16B_f32): reads the PRE16 layout above as it
// stands (nst16 = its tile count), one workgroup per direction and 4 sequences, nst4 = ceil(sequences / 4).  Inference
// only.
int lstm4_launch(bool relu, int nst4, int nst16, int ndir, void* stream, const float* pre, const float* whh_f,
                 const float* whh_b, float* hc, int ldh, int dump_row, const SeqGeom& g, bool packed = false);
// `packed`: whh_f / whh_b are fragment-order copies made by this launch (n <= LSTM4_PACK_MAX matrices of 512 x 128, nullptr
// entries skipped) -> dst[n][512 * 128]
constexpr int LSTM4_PACK_MAX = 24;
int lstm4_pack_launch(void* stream, const float* const* src, int n, float* dst);

// FUSED input projection (lstm16x.hip, num_features = 64): x_t W_ih^T + b is formed inside the recurrence (W_ih resident in
// LDS), no pre-activation tensor in memory.  x: token-major [M][ldx]; wih / bih / bhh / whh: [direction] -> the nn.LSTM
// tensors in their PyTorch layouts; hc addressed with 64-bit offsets (any launch size).  Inference only.
int lstm16x_launch(int nin, bool relu, int nst16, int ndir, void* stream, const float* x, int ldx, const float* const* wih,
                   const float* const* bih, const float* const* bhh, const float* const* whh, float* hc, int ldh,
                   int64_t dump_row, const SeqGeom& g);
