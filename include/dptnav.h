/*
 * dptnav.h -- C ABI of libdptnav.so: the MI355X (gfx950) implementation of the
 * DPTN / DPTN-AV raw-waveform separation FORWARD path of teasgen/speech_separation.
 *
 * What this boundary replaces in the reference (paths relative to /root/reference):
 *   The reference has no FFI/plugin layer; its boundary for this path is the nn.Module
 *   duck type consumed by the trainer/inferencer:
 *     construction  hydra.utils.instantiate(config.model)            train.py:43, inference.py:40
 *                   kwargs = src/configs/model/dptn_wav_av.yaml:1-12 (dptn_wav.yaml:1-10)
 *     call          outputs = self.model(**batch)                    src/trainer/trainer.py:40,
 *                                                                    src/trainer/inferencer.py:117
 *     body          DPTNAVWavEncDec.forward                          src/model/dptn_wav.py:171-194
 *                   DPTNWavEncDec.forward (audio only)               src/model/dptn_wav.py:105-117
 *     checkpoint    state_dict()/load_state_dict()                   src/trainer/base_trainer.py:476,519,557-560
 *   The Python shim speech_separation_amd/model.py keeps that duck type (same ctor kwargs,
 *   same forward signature, same state_dict keys) and calls the entry points below through
 *   ctypes with raw device pointers -- see INTEGRATION.md for the reference-side binding.
 *
 * Conventions
 *   - plain C types only; every pointer named *dev* / every tensor argument is a DEVICE pointer
 *     to contiguous fp32 owned by the caller (PyTorch-ROCm caching allocator in practice);
 *   - the library allocates nothing on the hot path: the caller passes a workspace of at
 *     least dptnav_workspace_bytes() bytes (256-byte aligned);
 *   - all work is enqueued on the hipStream_t passed as `stream` (void* here so that C callers
 *     do not need hip headers); nothing synchronises the device;
 *   - every function returns 0 on success, non-zero on error; the message is retrievable with
 *     dptnav_last_error().  The library never aborts and never falls back to a CPU path;
 *   - a handle is bound to the device current at dptnav_create() time and is NOT thread-safe
 *     (the reference's caller is single-threaded: trainer.py runs under the GIL).
 */
#ifndef DPTNAV_H_
#define DPTNAV_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DPTNAV_ABI_VERSION 3

/* error codes */
#define DPTNAV_OK 0
#define DPTNAV_ERR_INVALID 1     /* bad argument / unsupported shape */
#define DPTNAV_ERR_WORKSPACE 2   /* workspace too small or misaligned */
#define DPTNAV_ERR_WEIGHTS 3     /* weights not bound / wrong count */
#define DPTNAV_ERR_HIP 4         /* a HIP call or launch failed */

/* Constructor arguments of DPTNAVWavEncDec / DPTNWavEncDec (dptn_wav.py:137-150, 72-83). */
typedef struct dptnav_config {
  int32_t num_features;    /* N: 128 (dptn_wav_av.yaml:2) or 64 (dptn_wav.yaml:2)              */
  int32_t video_emb_size;  /* 512; ignored when audio_only                                      */
  int32_t hidden_video;    /* must equal num_features (it is added to the latent, :184)         */
  int32_t kernel_size_enc; /* 7 -> stride 3                                                     */
  int32_t hidden_dim;      /* H: LSTM hidden size, must be 128                                  */
  int32_t num_blocks;      /* 6                                                                 */
  int32_t chunk_size;      /* K = 150 (<= 256)                                                  */
  int32_t step_size;       /* P = 75                                                            */
  int32_t num_heads;       /* 4; head dim N/heads must be 32 or 16                              */
  int32_t bidir;           /* inter-chunk LSTM bidirectional (intra always is, dptn.py:59)      */
  int32_t audio_only;      /* 1 -> DPTNWavEncDec / DPRNNEncDec (no video branch, no gate)       */
  int32_t arch;            /* 0: DPTN blocks = TransformerDPRNN (dptn.py:9-52)
                              1: DPRNN blocks = IntraChunkRNN/InterChunkRNN (dprnn.py:7-89): LSTM -> fc -> LayerNorm -> +res,
                                 no attention; head and tail are identical (dprnn.py:200-227,260-274)           */
} dptnav_config;

typedef struct dptnav_ctx* dptnav_handle;

/* ---- lifetime --------------------------------------------------------------------------- */
int dptnav_abi_version(void);
int dptnav_create(const dptnav_config* cfg, dptnav_handle* out);
void dptnav_destroy(dptnav_handle h);
/* last error of this handle (h may be NULL: error of the last failed dptnav_create). */
const char* dptnav_last_error(dptnav_handle h);

/* ---- weights: the caller keeps ownership (nn.Parameter storage) ------------------------- */
/* The weight table has one slot per state_dict() tensor, in state_dict() order
 * (228 slots for dptn_wav_av.yaml).  Names are the checkpoint keys. */
int dptnav_num_weights(dptnav_handle h);
const char* dptnav_weight_name(dptnav_handle h, int slot);
int64_t dptnav_weight_numel(dptnav_handle h, int slot);
/* Borrow n device pointers (fp32, contiguous, original PyTorch layouts).  May be called again
 * whenever the parameters move (e.g. after .to(device) or load_state_dict).  No copies are made. */
int dptnav_bind_weights(dptnav_handle h, const float* const* dev_ptrs, int n);

/* ---- sizes ------------------------------------------------------------------------------ */
int64_t dptnav_frames(dptnav_handle h, int64_t T);  /* L = (T-k)/stride + 1                    */
int64_t dptnav_chunks(dptnav_handle h, int64_t T);  /* S = (L-K)/P + 1                         */
size_t dptnav_workspace_bytes(dptnav_handle h, int B, int64_t T, int Tv);

/* ---- the hot path ----------------------------------------------------------------------- */
/* replaces: DPTNAVWavEncDec.forward(mix, s1_embedding, s2_embedding, **batch) dptn_wav.py:171-194
 *   mix      (B,T)            fp32     <- batch["mix"]
 *   e1, e2   (B,Cv,Tv)        fp32     <- batch["s1_embedding"], batch["s2_embedding"] (NULL if audio_only)
 *   s1_pred, s2_pred (B,T)    fp32     -> {"s1_pred","s2_pred"}
 */
int dptnav_forward(dptnav_handle h, const float* mix, const float* e1, const float* e2, int B, int64_t T,
                   int Tv, float* s1_pred, float* s2_pred, void* workspace, size_t workspace_bytes,
                   void* stream);

/* ---- stage entry points (the forward is exactly head -> 2*num_blocks paths -> tail) ------ */
/* Token layout used between stages: x[b][s][k][n] (channel-last), i.e. the reference's
 * (B,N,S,K) tensor permuted to (B,S,K,N) -- intra sequences are rows (b,s), inter sequences
 * are strided views (b,k); no rearrange copies (dptn.py:71,74,77 do three per block).
 *
 * head: encoder conv + video fusion + chunking                 dptn_wav.py:172-184, dprnn.py:122-136
 *   encoded (B,L,N) = fused latent (kept for the decoder skip, dptn_wav.py:188)
 *   chunked (B,S,K,N) */
int dptnav_stage_head(dptnav_handle h, const float* mix, const float* e1, const float* e2, int B, int64_t T,
                      int Tv, float* encoded, float* chunked, void* workspace, size_t workspace_bytes,
                      void* stream);
/* path: one TransformerDPRNN (dptn.py:36-52) of block `block`; path 0 = intra-chunk, 1 = inter-chunk.
 *   x_in, x_out (B,S,K,N), must not alias. */
int dptnav_stage_path(dptnav_handle h, int block, int path, const float* x_in, float* x_out, int B, int S,
                      void* workspace, size_t workspace_bytes, void* stream);
/* tail: PReLU + 1x1 conv + overlap-add + pad + postprocessing + skip + transposed conv + pad
 *   dptn_wav.py:47-61, 186-193 */
int dptnav_stage_tail(dptnav_handle h, const float* x, const float* encoded, int B, int64_t T, float* s1_pred,
                      float* s2_pred, void* workspace, size_t workspace_bytes, void* stream);

/* ---- loss / metric statistics --------------------------------------------------------------------
 * replaces the arithmetic of SiSNRLoss.forward (src/loss/ss_losses.py:100-114) and of the four + two
 * ScaleInvariantSignalNoiseRatio calls of SISNRiMetric / SS2BaseMetric (src/metrics/si_snri.py:12-30,
 * src/metrics/base_metric.py:41-60), which cost the reference >= 6 device->host syncs per batch.
 *   all inputs (B,T) fp32 on the device; out (B,6,2) fp32 on the device:
 *   pair order (s1_pred,s1) (s1_pred,s2) (s2_pred,s1) (s2_pred,s2) (mix,s1) (mix,s2);
 *   [..][0] = SI-SNR in dB with torchmetrics' eps convention, [..][1] = the reference loss term -20 log10(.) (no eps).
 * Batch means and the batch-level PIT are a few flops on 12*B numbers: speech_separation_amd/metrics.py. */
int dptnav_sisnr_pairs(dptnav_handle h, const float* s1_pred, const float* s2_pred, const float* s1, const float* s2,
                       const float* mix, int B, int64_t T, float* out, void* stream);

/* ---- introspection for tests / profiling ------------------------------------------------ */
/* Offsets (in bytes, from the workspace base) of intermediates left behind by the LAST
 * dptnav_stage_path call: "qkv" (M,3N), "att" (M,N), "y1" (M,N) [post-LN1], "hc" (M,2H)
 * [ReLU(h_fwd|h_bwd)].  Returns non-zero for an unknown name. */
int dptnav_workspace_tap(dptnav_handle h, int B, int64_t T, int Tv, const char* name, size_t* offset_bytes,
                         size_t* numel);
/* ---- training step, path level (BASELINE config 4; DPTN architecture, num_features = 128, dropout 0) ------------
 * The backward of one TransformerDPRNN (what torch.autograd derives from dptn.py:36-52 in the reference's
 * loss.backward(), src/trainer/trainer.py:47).  The forward variant keeps on a caller-provided tape what the backward
 * needs (qkv, attention output, LayerNorm-1 output, raw LSTM output, post-activation gates and cell states);
 * attention probabilities and the pre-LayerNorm activations are recomputed.  Parameter gradients are WRITTEN (not
 * accumulated) to the buffers bound with dptnav_bind_grads (same slot order as dptnav_bind_weights).
 *   x_in, x_out, d_out, d_in: (B,S,K,N) token layout; d_in may not alias d_out. */
int dptnav_bind_grads(dptnav_handle h, float* const* dev_ptrs, int n);
size_t dptnav_train_path_tape_bytes(dptnav_handle h, int B, int S);
size_t dptnav_train_bwd_workspace_bytes(dptnav_handle h, int B, int S);
int dptnav_train_path_forward(dptnav_handle h, int block, int path, const float* x_in, float* x_out, int B, int S,
                              void* tape, size_t tape_bytes, void* workspace, size_t workspace_bytes, void* stream);
int dptnav_train_path_backward(dptnav_handle h, int block, int path, const float* x_in, const float* d_out, float* d_in,
                               int B, int S, void* tape, size_t tape_bytes, void* bwd_workspace,
                               size_t bwd_workspace_bytes, void* stream);

/* ---- training step, whole model ------------------------------------------------------------------------------
 * dptnav_train_forward  = dptnav_forward that records the tape (a batch >= 2 runs as two halves on the two internal
 *   streams like dptnav_forward: option "train_overlap", default 1);
 * dptnav_train_backward = everything torch.autograd would do for the model part of loss.backward()
 *   (src/trainer/trainer.py:47): given d loss / d s1_pred and d loss / d s2_pred it WRITES the gradient of every
 *   parameter into the buffers bound with dptnav_bind_grads.  The loss itself (src/loss/ss_losses.py), gradient
 *   clipping and the optimizer stay the reference's own PyTorch code (speech_separation_amd/model.py wraps these two
 *   calls in a torch.autograd.Function).  Train-mode attention dropout: options dropout_ppm / dropout_seed below.
 *   Limit of the training step (not of inference): Tv <= 256 video frames per mixture (> 10 s of 25 fps lip embeddings);
 *   the size queries return 0 and dptnav_train_forward fails with DPTNAV_ERR_INVALID BEFORE launching anything. */
size_t dptnav_train_tape_bytes(dptnav_handle h, int B, int64_t T, int Tv);
size_t dptnav_train_workspace_bytes(dptnav_handle h, int B, int64_t T, int Tv);
int dptnav_train_forward(dptnav_handle h, const float* mix, const float* e1, const float* e2, int B, int64_t T, int Tv,
                         float* s1_pred, float* s2_pred, void* tape, size_t tape_bytes, void* workspace,
                         size_t workspace_bytes, void* stream);
int dptnav_train_backward(dptnav_handle h, const float* mix, const float* e1, const float* e2, const float* d_s1_pred,
                          const float* d_s2_pred, int B, int64_t T, int Tv, void* tape, size_t tape_bytes,
                          void* workspace, size_t workspace_bytes, void* stream);

/* ---- the tail of the training step on the device: no host synchronisation ------------------------
 * Flat layout shared by the gradient / optimizer-state buffers: slot i (state_dict order) starts at
 * dptnav_flat_offset(h, i) floats -- the sum of the previous slots' sizes, each rounded up to 64 floats (256 bytes) --
 * and dptnav_flat_numel(h) floats hold all slots; the padding between slots must be zero.  The gradient pointers handed
 * to dptnav_bind_grads may (and in speech_separation_amd do) point into one such buffer, so that data parallelism needs
 * ONE all-reduce and clip + AdamW need no per-tensor launches. */
int64_t dptnav_flat_offset(dptnav_handle h, int slot);
int64_t dptnav_flat_numel(dptnav_handle h);
size_t dptnav_tail_scratch_bytes(dptnav_handle h, int B);
/* replaces: SiSNRWavLoss.forward + its autograd backward           src/loss/ss_losses.py:21-26 (BaseSSLoss: batch-level PIT),
 *                                                                  :100-114 (SiSNRLoss), src/trainer/trainer.py:43,47
 *   s1_pred, s2_pred, s1, s2 (B,T) fp32 -> loss_out[4] = {loss, chosen permutation (0: (p1,s1)+(p2,s2), 1: swapped),
 *   loss of permutation 0, loss of permutation 1}; d_s1_pred, d_s2_pred (B,T) = grad_scale * d loss / d prediction -- the
 *   inputs of dptnav_train_backward.  The permutation is resolved on the device (the reference converts a tensor to bool).
 *   scratch: >= dptnav_tail_scratch_bytes(h, B) bytes, 8-byte aligned. */
int dptnav_pit_sisnr_loss(dptnav_handle h, const float* s1_pred, const float* s2_pred, const float* s1, const float* s2,
                          int B, int64_t T, float grad_scale, float* d_s1_pred, float* d_s2_pred, float* loss_out,
                          void* scratch, size_t scratch_bytes, void* stream);
/* replaces: clip_grad_norm_(model.parameters(), max_grad_norm)     src/trainer/base_trainer.py:383-391
 *   flat_grad: n_flat floats in the flat layout; scaled in place by min(1, max_norm / (norm + 1e-6)); norm_out[0] = the
 *   global L2 norm BEFORE clipping (what clip_grad_norm_ returns).  max_norm <= 0: norm only. */
int dptnav_grad_clip(dptnav_handle h, float* flat_grad, int64_t n_flat, float max_norm, void* scratch, size_t scratch_bytes,
                     float* norm_out, void* stream);
/* replaces: torch.optim.AdamW.step()                               src/configs/dptn_wav_av.yaml:9-11, src/trainer/trainer.py:49
 *   Updates IN PLACE the parameters bound with dptnav_bind_weights (those pointers must be writable); flat_grad,
 *   exp_avg, exp_avg_sq: dptnav_flat_numel(h) floats each; step = 1 for the first update (bias correction). */
int dptnav_adamw_step(dptnav_handle h, const float* flat_grad, float* exp_avg, float* exp_avg_sq, int64_t n_flat, double lr,
                      double beta1, double beta2, double eps, double weight_decay, int step, void* stream);

/* Test helper: mask (nseq, heads, len, len) fp32 of ones/zeros = the keep-mask of path (block, path). */
int dptnav_dropout_mask(dptnav_handle h, int block, int path, int B, int S, float* mask, void* stream);

/* Tuning / diagnostic knobs (never needed for correct results).  Keys:
 *   "overlap" (0/1, default 1): dptnav_forward runs the batch as two halves on two internal streams (forked from
 *                 and joined to the caller's stream by events) so that one half's GEMM/attention launches fill the
 *                 CUs the other half's LSTM recurrence cannot use; 0 = everything on the caller's stream.
 *   "serialize" (0/1, default 0): a MEASUREMENT knob -- dptnav_forward keeps its sub-batch cut and kernel selection but enqueues
 *                 every launch on the caller's stream, one after the other (no internal stream, no event): each launch is then
 *                 alone on the chip, which is what bench.py's roofline figures time.  Results are bit-identical to 0.
 *   "dropout_ppm" (0..999999), "dropout_seed": train-mode dropout of the attention probabilities (dptn.py:16-21,
 *                 nn.MultiheadAttention(dropout=0.1)) for dptnav_train_forward/backward.  The keep-mask is a counter-based
 *                 hash of (seed, block, path, query token, head, key position): reproducible, identical in forward and
 *                 backward, but NOT PyTorch's Philox stream (parity with the reference is statistical only).  The caller
 *                 changes the seed every step and sets the same seed for the backward of that step.
 *   "train_overlap" (0/1, default 1): run the training step (dptnav_train_forward / _backward) of a batch >= 2 as two
 *                 halves on the two internal streams, like the forward ("overlap" = 0 switches this off too).  Must not
 *                 change between a forward and its backward: the tape layout depends on it.
 *   "fuse_attn" (0/1, default 1): inference with num_features = 128 runs in-projection + attention + out-projection +
 *                 LayerNorm 1 of a TransformerDPRNN as ONE kernel when the sequences are <= 160 positions long (QKV and
 *                 the attention output stay on chip; the "qkv" / "att" workspace taps are then not written); 0 = the three
 *                 separate launches (always used by the training forward, which keeps qkv / att on the tape).
 *   "fuse_ffn" (0/1, default 1): dptnav_forward leaves the FFN half (K6: Linear + residual + LayerNorm 2) of a path to the
 *                 NEXT path's fused attention block, which produces its own input rows (attn_block.hip prologue): the
 *                 tensor between two TransformerDPRNNs never goes through HBM.  Only where that block is the fused
 *                 fp32 kernel; the stage entry points always run whole paths.
 *   "fold_tail" (0/1, default 1): inference computes the decoder tap products as ONE contraction per frame,
 *                 [OLA(mask) | encoded] . [W_dec^T W_post | W_dec^T] + W_dec^T b_post (folded from the current weights on
 *                 every call), instead of the post-processing GEMM with a k-reduction epilogue.
 *   "wgrad2" (0/1, default 1): training computes the W_ih and W_hh gradients of an LSTM in one pass over dP, the four
 *                 (direction, gate-row half) problems of a path in one launch; 0 = eight single-gradient launches.
 *   "wgrad_ride" (0/1, default 1): training forms the out-projection and ffn.1 weight / bias gradients inside the
 *                 data-gradient GEMMs of those layers (they stage the same dY tile); 0 = separate weight-gradient launches.
 *   "ln_tape" (0/1, default 1): the training forward leaves the normalised rows and 1/sigma of both LayerNorms of a path on the
 *                 tape (1 kB per token and path more) and the backward applies the LayerNorm derivative in a bandwidth-bound
 *                 pass; 0 = recompute the pre-LayerNorm rows in a GEMM with a derivative epilogue (round 2).  Set it before
 *                 dptnav_train_tape_bytes: the tape is sized for it.
 *   "wgrad_side" (0/1, default 1): training with a split batch runs the LSTM weight-gradient launches on a side stream per
 *                 half (two more dP buffers in the backward workspace); 0 = in the half's own stream.
 *   "split_policy" (0/1, default 1): 1 = three sub-batches where a half-batch recurrence launch needs more than half of the
 *                 CUs but a third needs less (B = 13..18 at T = 32000); 0 = always as few sub-batches as make every
 *                 recurrence launch fit the chip in one round, at least two.  Set it before dptnav_workspace_bytes.
 *   "lstm_chain" (0/1, default 0): 1 = the recurrence (and BPTT) launches of the sub-batches / halves are chained by events so
 *                 that one is in flight at a time (the schedule of rounds 1-2; measured slower at every batch size in round
 *                 3); "lstm_inflight" (0..32, default 0 = off) sets the depth of that chain for experiments
 *                 (tools/split_sweep.py).
 *   "sub_batches" (0..32, default 0): how many sub-batches dptnav_forward cuts a batch into; 0 = the policy above.  A measurement knob
 *                 (tools/subbatch_sweep.py): results of different cuts agree to fp32 rounding, not bit for bit.  Set it
 *                 before dptnav_workspace_bytes -- the workspace is sized for the cut.
 *   "split_bf16" (0/1, default 0): OPT-IN experiment, never a parity claim -- the 16-sequence-tile recurrence of the
 *                 inference forward runs on bf16 MFMAs with every operand split into bf16 hi + lo (three products, fp32
 *                 accumulation: ~2^-17 relative error per product instead of 2^-24, ~5x less matrix time); lstm16s.hip.
 *   "lstm16" (0/1, default 1): use 16-sequence LSTM tiles whenever a launch then still fits the chip in one round
 *                 (half-batch launches): same CU-time, half the serial time of the recurrence.
 *   "deterministic" (0/1, default 0): the token-tile kernels (GEMM engine, weight-gradient kernels) take their tiles in a
 *                 static order (workgroup b: tiles b, b + grid, ...) instead of from device-wide ticket counters.  Outputs
 *                 of the forward are bit-reproducible either way; with 1 the parameter gradients of the training step are
 *                 too (with tickets WHICH workgroup sums which tiles varies from run to run: ~1e-7 relative).  Costs 3 % of
 *                 a training step: a workgroup that starts late, beside another stream's kernel, keeps its share.
 *   "pack_wih" (0/1, default 1): the LSTM pre-activation GEMM reads W_ih from a fragment-order copy made at the start of
 *                 every pass (coalesced fragment loads in every workgroup's prologue); bit-identical to 0.
 *   "fuse_pre" (0/1, default 1): num_features = 64, inference: the LSTM input projection x_t W_ih^T + b runs INSIDE the recurrence
 *                 (lstm16x.hip: W_ih resident in LDS, 50 % more matrix work per step) instead of as a GEMM launch that writes
 *                 the pre-activations to the workspace (2 KiB per token and direction, read again by the recurrence); launches
 *                 small enough for the low-latency recurrence (lstm4) keep the GEMM.  Same sums in another order: results
 *                 agree with 0 to fp32 rounding (> 110 dB).
 *   "fuse_pre128" (0/1/2, default 1): the same for num_features = 128 (lstm16x128_kernel: W_ih is as large as W_hh, so 33 of a
 *                 wave's 64 W_ih fragment sets live in VGPRs and 31 in LDS; twice the matrix work per step).  1 = for
 *                 batches of at least 12 mixtures (a fused launch takes its 1.17 ms whatever its size: below that the chip
 *                 is not full and the GEMM + recurrence pair is the shorter chain -- B = 8: 17.0 vs 18.0 ms); 2 = always;
 *                 0 = never.  Measured at B = 16 x 4 s: the same step time as the pair (29.75 vs 29.75 ms), 1.7-2 %
 *                 faster from B = 24, and 32.8 GB per step less HBM traffic (52.1 -> 19.3 GB: the pre-activation tensor was
 *                 80 % of a forward's).
 *   "fcln" (0/1/2, default 1): a Linear layer with its LayerNorm and residual by fcln.hip -- 16-token tiles fetched by LDS-DMA,
 *                 two or three workgroups per CU -- instead of the GEMM engine's 32- / 64-token tiles, one workgroup per CU (0):
 *                 (a) DPRNN blocks with num_features = 64 and two directions, inference: Linear(256 -> 64) + LayerNorm +
 *                 residual (dprnn.py:41-45, 83-87); 1 = one tile ahead, three workgroups per CU, 2 = two ahead, two per CU.
 *                 B = 32 x 16 s: 1.25 / 1.21 vs 1.64 ms per launch alone on the chip, 327.1 / 327.8 vs 332.6 ms per forward;
 *                 (b) the training forward with num_features = 128: out-projection + LayerNorm 1 and ReLU -> Linear +
 *                 LayerNorm 2 (dptn.py:46-47, 50-51) with the LayerNorm tape;
 *                 (c) inference, DPTN: Linear + LayerNorm 2 of a path whose FFN does not ride in the next attention block (the last
 *                 one of a forward), and the separation conv behind PReLU (dptn_wav.py:26-29, 47; no LayerNorm) in every
 *                 forward, 64 and 128 features.  Same sums in another order (> 100 dB to 0).
 *   "pack_whh" (0/1, default 1): the low-latency recurrence (lstm4) reads W_hh from a fragment-order copy made at its
 *                 first launch of a pass; bit-identical to 0.
 *   "lstm4" (0/1/2, default 1): the LOW-LATENCY recurrence on 4-sequence tiles (lstm4.hip, v_mfma_f32_4x4x1_16B_f32; a
 *                 step takes ~1.2 us instead of ~4.3 us, four times the workgroups) for inference launches of at most
 *                 1.15 rounds of the chip -- bs = 1..3 whole, sub-batches of up to 4 mixtures at 4 s; 2 = wherever
 *                 the 16-tile layout is in use (experiments, tests); 0 = never.  Same results to fp32 rounding (the k
 *                 order of the recurrent sum differs: ~135 dB between the two kernels' forwards).
 *   "inject_fail" (n > 0): fault injection for the error-path tests -- the n-th GEMM-engine launch from now on returns
 *                 DPTNAV_ERR_INVALID (once); the forked entry points must still join their internal streams.
 *   "lstm_stamps" (0/1): diagnostic LSTM build that writes per-wave s_memtime segment sums (u64 [dir][tile][wave][4]:
 *                 accumulator init, MFMA, cell update, barrier) to the "lstm_stamps" workspace tap.
 *   "attn_v2" (0/1, default 1): the fused attention block of the inference forward (num_features = 128) in the form with both
 *                 LayerNorms in fragment space and the FFN prologue's rows by LDS-DMA (attn_block2.hip); 0 = round 2-4's kernel
 *                 (attn_block.hip, kept for same-process A/B).  Same results up to the grouping of the LayerNorm sums (> 100 dB).
 *   "dgrad_t" (0/1, default 1): training backward, num_features = 128 -- the K = 512 data gradient of the LSTM's input product
 *                 (d y1 = dz + dG W_ih, dptn.py:48) and the K = 384 one of the attention in-projection (dptn.py:46) by dgrad_t.hip
 *                 (transposed product, rows by LDS-DMA) instead of the GEMM engine; 0 = the engine (kept for same-process A/B).
 *                 Same sums in another association: gradients agree to fp32 rounding, the forward is untouched.
 *   "dgrad_r" (0/1, default 1): training backward, num_features = 128 -- the two K = 128 data gradients whose layer's weight / bias
 *                 gradient is formed on the same staged tiles (d att + dW_o + db_o, dptn.py:46-47; d h + dW_f + db_f behind the
 *                 ReLU, dptn.py:50) by dgrad_r.hip; 0 = the GEMM engine with its WgradRider.  Same sums in another association.
 *   "gemm_t" (0/1, default 1): training forward, num_features = 128 -- the attention in-projection qkv = x W_in^T + b_in
 *                 (dptn.py:16-21, 46) by gemm_t.hip (transposed product, accumulators started from the bias); 0 = the GEMM engine.
 *                 Same sums in another association: outputs and gradients agree to fp32 rounding.
 *   "train_fuse_probe" (0/1, default 0): MEASUREMENT ONLY (tools/train_fuse_probe.py) -- the training forward runs the inference
 *                 attention block: no qkv / attention / LayerNorm tape is written and no dropout is applied.  While it is set
 *                 dptnav_train_backward and dptnav_train_path_backward return DPTNAV_ERR_INVALID. */
int dptnav_set_option(dptnav_handle h, const char* key, int value);

/* Opt-in per-kernel timing: while enabled every launch is bracketed by two hipEvents recorded on the
 * launch stream (so the figures are device time of that kernel, not host time).  collect() waits for
 * the recorded events and accumulates them per kernel class; ms()/count() read the accumulators.
 * Kernel classes: video_linear, encoder_fuse, qkv_gemm, attention, outproj_ln_gemm, lstm_pre_gemm,
 * lstm_recurrence, ffn_ln_gemm, sep_gemm, postproc_gemm, decoder_gather.  Not graph-capturable. */
int dptnav_profile_enable(dptnav_handle h, int on);
int dptnav_profile_collect(dptnav_handle h);
int dptnav_profile_reset(dptnav_handle h);
int dptnav_profile_num(void);
const char* dptnav_profile_name(int cls);
double dptnav_profile_ms(dptnav_handle h, int cls);
int64_t dptnav_profile_count(dptnav_handle h, int cls);

/* Algorithmic cost model used for roofline reporting. */
double dptnav_flops_per_mixture(dptnav_handle h, int64_t T);
double dptnav_min_bytes_per_mixture(dptnav_handle h, int64_t T);

#ifdef __cplusplus
}
#endif
#endif /* DPTNAV_H_ */
