// dgrad_r.h -- launcher of dgrad_r.hip: a K = 128 data gradient with its layer's weight / bias gradient formed on the same
// staged tiles ("rider"):
//
//     out[M][nout] = (A[M][128] W[128][nout]) masked by X > 0 when relu_gate        nout = 128: d att = dz1 W_o  (dptn.py:46-47)
//     slab[workgroup]    = partial  A^T relu?(X)   [128][nout], fragment order       nout = 256: d h = (dz2 W_f) (.) (h > 0), X = h
//     colslab[workgroup] = partial column sums of A  [128]                                        (dptn.py:50: ReLU -> Linear)
//
// A [M][128], X [M][nout], out [M][nout] dense; Wpacked = dgrad_r_pack_launch's fragment-order copy of the row-major forward
// weight [128][nout].  The partial tiles have the layout of gemm_ws.h's WgradRider (slab_reduce_frag_kernel<1, nout / 32> sums
// them; *grid_used says how many were written, at most max_slabs).  queue: zeroed ticket counter or null (static tile order:
// bit-reproducible gradients).  Returns a hipError_t as int; hipErrorInvalidValue for shapes it does not take.
#pragma once
#include <cstdint>

struct DgradRArgs {
  const float* A;
  const float* Wpacked;
  const float* X;
  float* out;
  int64_t M = 0;
  int nout = 0;
  bool relu_gate = false;
  unsigned* queue = nullptr;
  float* slab = nullptr;
  float* colslab = nullptr;
  int max_slabs = 0;
};
int dgrad_r_launch(void* stream, const DgradRArgs& a, int num_cus, int* grid_used);
// fragment-order copies of n (<= DGRAD_R_PACK_MAX) row-major [128][nout] weights in ONE launch: srcs[i] (null: skipped) -> dst + dst_off[i]
constexpr int DGRAD_R_PACK_MAX = 24;
int dgrad_r_pack_launch(void* stream, const float* const* srcs, const long long* dst_off, int n, int nout, float* dst);
