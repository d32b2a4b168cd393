// lstm.h -- LSTM recurrence on 32-sequence tiles (H = 128): shared declarations.  The kernel is in lstm.hip (its own
// translation unit, compiled with -mllvm -amdgpu-mfma-vgpr-form like lstm16.hip, see build.py).
//
// Pre-activation layout (written by EpiLstmPre, gemm_ws.h; read by LDS-DMA):
//     PRE[d][st][t][cb16][q4][hh2][c32][i4]      one (tile, step) = 32 sequences x 512 gate columns = 64 KiB
//   = the 32x512 gate tile in MFMA accumulator-fragment order (cb = gate*4 + wave; registers 4q..4q+3 of lane (c,hh)).
//   Values are PRE-SCALED by lstm_gate_scale(gate) so that the kernel evaluates sigmoid / tanh without a multiply.
// Training tape (SAVE): post-activation gates i,f,g,o and the cell state c_t of every step, same fragment order:
//     tape_gates[d][st][t][cb16][q4][hh2][c32][i4]   (64 KiB per tile and step)
//     tape_c    [d][st][t][w4 ][q4][hh2][c32][i4]    (16 KiB per tile and step)
#pragma once
#include "common.h"

constexpr int LSTM_H = 128;
constexpr int LSTM_LDH = LSTM_H + 4;
constexpr int LSTM_HS_FLOATS = 2 * 32 * LSTM_LDH;          // h double buffer
constexpr int LSTM_PRE_FLOATS = 4 * 16 * 256;               // one step of pre-activations: [wave][gate*4+q][lane][4]
constexpr size_t LSTM_LDS_BYTES = sizeof(float) * (LSTM_HS_FLOATS + LSTM_PRE_FLOATS);

// scale of gate `gate`'s pre-activation rows (gate order i, f, g, o: torch.nn.LSTM):
// sigmoid(x) = 1 / (1 + 2^(-log2e x)),  tanh(x) = 2 / (1 + 2^(-2 log2e x)) - 1
DEV constexpr float lstm_gate_scale(int gate) { return gate == 2 ? -2.8853900817779268f : -1.4426950408889634f; }

// Host-side launcher (lstm.hip).  hc has dump rows from index `dump_row` on (rows of padded sequences are written
// there, branch-free).  relu: store ReLU(h) (DPTN inference: the only consumer is ffn = ReLU -> Linear, dptn.py:30-33).
// save: training forward, keeps the tape (tape_gates / tape_c).  stamp: diagnostic build, per-wave s_memtime sums of
// [acc-init, MFMA block, cell update, barrier] -> stamps[dir][tile][wave][4].  Returns a hipError_t as int.
int lstm32_launch(bool stamp, bool save, bool relu, int nst, int ndir, void* stream, const float* pre, const float* whh_f,
                  const float* whh_b, float* hc, int ldh, int dump_row, const SeqGeom& g, unsigned long long* stamps,
                  float* tape_gates, float* tape_c);
