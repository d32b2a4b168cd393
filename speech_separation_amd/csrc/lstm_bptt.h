// lstm_bptt.h -- LSTM backward through time: shared declarations (kernel in lstm_bptt.hip).
#pragma once
#include "common.h"
#include "lstm.h"
#include "lstm16.h"

constexpr int BPTT_LDP = 512 + 4;
constexpr size_t BPTT_LDS_BYTES = sizeof(float) * 2 * 32 * BPTT_LDP;

// One launch = both directions of one TransformerDPRNN's LSTM over `nst` tiles of 32 sequences.
//   tape_gates / tape_c : the forward's tape (lstm.h);  dh_up[tok][ldh] : upstream gradient of h (dump rows included);
//   dg_out[tok][ldg]    : dP = gradient of the gate pre-activations, token-major (feeds the W_ih / W_hh gradient GEMMs
//                         and the data gradient);  bias_partials[ndir][nst][512] : per-workgroup column sums of dP.
// Returns a hipError_t as int.
int lstm_bptt_launch(int nst, int ndir, void* stream, const float* tape_gates, const float* tape_c, const float* whh_f,
                     const float* whh_b, const float* dh_up, int ldh, float* dg_out, int ldg, int dump_row,
                     const SeqGeom& g, float* bias_partials);

// 16-sequence-tile variant (lstm_bptt16.hip): consumes the tape of lstm16.hip's training forward; bias_partials is
// [ndir][nst16][512].
constexpr int BPTT16_LDP = 512 + 8;
constexpr size_t BPTT16_LDS_BYTES = sizeof(float) * 2 * 16 * BPTT16_LDP;
int lstm_bptt16_launch(int nst16, int ndir, void* stream, const float* tape_gates, const float* tape_c, const float* whh_f,
                       const float* whh_b, const float* dh_up, int ldh, float* dg_out, int ldg, int dump_row,
                       const SeqGeom& g, float* bias_partials);
