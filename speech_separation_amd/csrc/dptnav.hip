// dptnav.hip -- host side of libdptnav.so: handle, weight table, workspace plan, launch sequence and
// the extern "C" boundary declared in include/dptnav.h.  gfx950 only.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/dptnav.h"
#include "attention.h"
#include "attn_block.h"
#include "attn_block64.h"
#include "common.h"
#include "gemm_ws.h"
#include "headtail.h"
#include "lstm.h"
#include "lstm16.h"
#include "fcln.h"
#include "dgrad_t.h"
#include "dgrad_r.h"
#include "gemm_t.h"
#include "sisnr.h"
#include "train_tail.h"
#include "backward.h"
#include "backward_ends.h"

namespace {

thread_local std::string g_create_error;

struct PathWeights {
  const float *in_w, *in_b, *out_w, *out_b, *ln1_w, *ln1_b;
  const float *w_ih[2], *w_hh[2], *b_ih[2], *b_hh[2];
  const float *ffn_w, *ffn_b, *ln2_w, *ln2_b;
  int ndir;
};

constexpr int QUEUE_SLOTS = 1024;

struct Plan {  // workspace offsets in floats
  int64_t L, S, M;
  size_t queue, vid, E, X0, X1, qkv, att, y1, pre, hc, stamps, wfold, wpack, wih, whh4, winp, total;
  size_t qkv_n, att_n, y1_n, hc_n;
};

// One in-flight pass over a (sub-)batch: its workspace slice, stream and ticket-counter cursor.
struct Run {
  float* ws;
  Plan pl;
  hipStream_t st;
  int slot;
  hipEvent_t lstm_wait = nullptr;    // if set: the recurrence launch waits for this event (other half's recurrence)
  hipEvent_t lstm_record = nullptr;  // if set: recorded right after the recurrence launch
  int half = 0;                      // which half of a split batch this run is (salts the dropout seed)
  bool packed_whh4 = false;          // ws + pl.whh4 holds the fragment-order W_hh copies of ALL paths for lstm4.hip (made at its first launch of the pass)
  bool packed_inw = false;           // ws + pl.winp holds the fragment-order in-projection weights of ALL paths (dptnav_train_forward); otherwise slot 0 = this path
  bool packed_wih = false;           // ws + pl.wih holds the fragment-order W_ih copies of ALL paths; otherwise run_path packs its own
  bool fuse128 = false;              // num_features = 128: input projection inside the recurrence for this pass (dptnav_ctx::fuse128_for)
  int batch_total = 0;               // mixtures of the whole call this run is a sub-batch of (0: a stage entry point, unknown)
  bool packed = false;               // ws + pl.wpack holds the packed weights of ALL paths (dptnav_forward); otherwise
                                     // run_path packs the path it is about to run
  unsigned* take_queue(int n) {
    unsigned* q = reinterpret_cast<unsigned*>(ws + pl.queue) + slot;
    slot += n;
    return q;
  }
};

}  // namespace

enum ProfCat { CAT_VIDEO = 0, CAT_ENCODER, CAT_QKV, CAT_ATTN, CAT_OUTPROJ, CAT_LSTM_PRE, CAT_LSTM, CAT_FFN,
               CAT_SEP, CAT_POST, CAT_DECODER, CAT_COUNT };
static const char* const kProfNames[CAT_COUNT] = {
    "video_linear", "encoder_fuse", "qkv_gemm", "attention", "outproj_ln_gemm", "lstm_pre_gemm",
    "lstm_recurrence", "ffn_ln_gemm", "sep_gemm", "postproc_gemm", "decoder_gather"};

struct ProfRec {
  int cat;
  hipEvent_t a, b;
};

struct dptnav_ctx {
  dptnav_config cfg;
  bool prof_on = false;
  bool opt_lstm_stamps = false;
  bool opt_overlap = true;
  bool opt_serialize = false;       // measurement: dptnav_forward keeps its sub-batch cut but enqueues every launch on the caller's stream
  bool opt_lstm16 = true;
  int opt_fuse_pre128 = 1;          // the same for num_features = 128 (lstm16x128_kernel: W_ih split between LDS and VGPRs): 0 never, 2 always,
                                    // 1 (default) from B = 12: a fused launch is 1.17 ms whatever its size -- below that the chip is not full
                                    // and the shorter K4 + recurrence chain wins (B = 8: 17.0 vs 18.0 ms; B = 16: equal; B = 24: 44.8 vs 44.0)
  bool fuse128_for(int B) const { return opt_fuse_pre128 == 2 || (opt_fuse_pre128 == 1 && B >= 12); }
  int opt_fcln = 1;                 // Linear + LayerNorm + residual on 16-token tiles, several workgroups per CU (fcln.hip) instead of the GEMM engine (0):
                                    // DPRNN fc (64 features, two directions, inference; 2 = two tiles ahead, two workgroups per CU) and the
                                    // training forward's out-projection + LN1 / FFN + LN2 with the LayerNorm tape (128 features)
  bool opt_fuse_pre = true;         // num_features = 64, inference: the input projection runs INSIDE the recurrence (lstm16x.hip), no K4 launch, no PRE tensor
  bool opt_pack_whh = true;         // lstm4.hip reads W_hh from a fragment-order copy made at its first launch of a pass (0: row per lane)
  bool opt_pack_wih = true;         // K4 reads W_ih from a fragment-order copy made once per pass (0: from the nn.Module tensor, row per lane)
  bool opt_deterministic = false;   // 1: static tile assignment instead of device-wide tickets (TileTickets): bit-reproducible gradients
  int opt_lstm4 = 1;          // 4-sequence recurrence tiles (lstm4.hip): 0 never, 1 for launches of up to 1.15 rounds of the chip, 2 whenever PRE16 is in use
  bool opt_split_bf16 = false;      // opt-in: LSTM recurrence on bf16 MFMAs with hi/lo-split operands (lstm16s.hip)
  bool opt_fuse_attn = true;        // inference: K1 + K2 + K3 as one kernel (attn_block.hip) where it applies
  bool opt_gemm_t = true;           // training forward: the attention in-projection (128 -> 384) by gemm_t.hip instead of the GEMM engine
  bool opt_dgrad_r = true;          // training: the K = 128 data gradients with weight-gradient riders by dgrad_r.hip instead of the GEMM engine
  bool opt_dgrad_t = true;          // training: the K = 512 data gradient (d P W_ih) by dgrad_t.hip instead of the GEMM engine
  bool opt_attn_v2 = true;          // ... in the form with both LayerNorms in fragment space and h rows by LDS-DMA (attn_block2.hip)
  bool opt_fold_tail = true;        // inference: post-processing conv + skip + decoder taps as one folded contraction
  bool opt_wgrad_ride = true;       // training: out-projection / ffn weight gradients formed inside their data-gradient GEMMs
  bool opt_wgrad2 = true;           // training: LSTM W_ih / W_hh gradients in one pass over dP (wgrad2_kernel)
  bool opt_ln_tape = true;          // training: LayerNorm backward from zn / rstd left on the tape instead of a recomputed GEMM
  bool opt_fuse_ffn = true;         // ... and K6 of a path as the prologue of the next path's block (dptnav_forward only)
  int opt_lstm_diag = 0;
  int opt_inject_fail = 0;          // > 0: the n-th GEMM-engine / fcln launch from now on returns an error (tests)
  bool opt_debug_sync = false;      // debugging aid: name every launch class on stderr and synchronise behind it
  bool opt_train_fuse_probe = false;   // MEASUREMENT ONLY (no tape is written: dptnav_train_backward / _path_backward refuse while it is set) (tools/train_fuse_probe.py): the training forward runs the inference attention block
                                       // (no qkv / att / LayerNorm tape: a backward after it is garbage) -- the upper bound of a fused front half
  int opt_dropout_ppm = 0;          // train-mode attention dropout probability x 1e6 (0 = off)
  unsigned opt_dropout_seed = 0;
  DropCfg drop_cfg(int block, int path, bool train, int half = 0) const {
    DropCfg d{0u, 0u, 1.0f};
    if (!train || opt_dropout_ppm <= 0) return d;
    const double p = opt_dropout_ppm * 1e-6;
    // (a split batch numbers the tokens of each half from 0: the half index keeps their masks apart)
    d.seed = opt_dropout_seed ^ (0x9E3779B9u * (unsigned)(2 * block + path + 1)) ^ (0x7F4A7C15u * (unsigned)half);
    d.thresh = (unsigned)(p * 16777216.0);   // 24 random bits per element (common.h)
    d.inv_keep = (float)(1.0 / (1.0 - p));
    return d;
  }
  static constexpr int NSTREAMS = 4;   // internal streams (dptnav_forward uses min(sub-batches, NSTREAMS); training two)
  hipStream_t streams[NSTREAMS] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_fork = nullptr, ev_join[NSTREAMS] = {nullptr, nullptr, nullptr, nullptr}, ev_lstm[2] = {nullptr, nullptr};
  hipEvent_t ev_side[2][4] = {{nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr}};   // side streams of the training backward
  bool opt_wgrad_side = true;       // training, split batches: LSTM weight gradients on a side stream per half
  int opt_sub_batches = 0;          // 0: forward_split decides; n > 0: that many sub-batches (experiments)
  int opt_lstm_inflight = 0;        // > 0: that many recurrence launches of dptnav_forward's sub-batches in flight (experiments)
  int opt_split_policy = 1;         // 1: sub-batch pairs with two recurrences in flight where >= 3 sub-batches result; 0: round-2 rule
  bool opt_lstm_chain = false;      // 1: at most one recurrence / BPTT launch of the sub-batches in flight (event chain, rounds 1-2)
  std::vector<hipEvent_t> ev_sub;   // recurrence-chain events of dptnav_forward's sub-batches (created on demand)
  int ensure_sub_events(int n) {
    while ((int)ev_sub.size() < n) {
      hipEvent_t e = nullptr;
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return fail(DPTNAV_ERR_HIP, "event creation");
      ev_sub.push_back(e);
    }
    return 0;
  }
  bool streams_ready = false;          // set LAST by ensure_streams: internal streams / events exist, all of them
  void release_streams() {             // destroys whatever exists (a partial creation included) and forgets it
    for (int i = 0; i < NSTREAMS; ++i) {
      if (streams[i]) { hipStreamDestroy(streams[i]); streams[i] = nullptr; }
      if (ev_join[i]) { hipEventDestroy(ev_join[i]); ev_join[i] = nullptr; }
      if (i < 2 && ev_lstm[i]) { hipEventDestroy(ev_lstm[i]); ev_lstm[i] = nullptr; }
    }
    if (ev_fork) { hipEventDestroy(ev_fork); ev_fork = nullptr; }
    for (int i = 0; i < 2; ++i)
      for (int k = 0; k < 4; ++k)
        if (ev_side[i][k]) { hipEventDestroy(ev_side[i][k]); ev_side[i][k] = nullptr; }
    streams_ready = false;
  }
  int ensure_streams() {               // all or nothing: a failure leaves no half-made set behind for the next call
    if (streams_ready) return 0;
    bool ok = true;
    for (int i = 0; i < 2 && ok; ++i)
      for (int k = 0; k < 4 && ok; ++k) ok = hipEventCreateWithFlags(&ev_side[i][k], hipEventDisableTiming) == hipSuccess;
    for (int i = 0; i < NSTREAMS && ok; ++i)
      ok = hipStreamCreateWithFlags(&streams[i], hipStreamNonBlocking) == hipSuccess &&
           hipEventCreateWithFlags(&ev_join[i], hipEventDisableTiming) == hipSuccess &&
           (i >= 2 || hipEventCreateWithFlags(&ev_lstm[i], hipEventDisableTiming) == hipSuccess);
    ok = ok && hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming) == hipSuccess;
    if (!ok) {
      (void)hipGetLastError();
      release_streams();
      return fail(DPTNAV_ERR_HIP, "cannot create internal streams/events");
    }
    streams_ready = true;
    return 0;
  }
  std::vector<ProfRec> prof_pending;
  std::vector<hipEvent_t> prof_pool;
  double prof_ms[CAT_COUNT] = {0};
  int64_t prof_n[CAT_COUNT] = {0};
  hipEvent_t prof_event() {
    if (!prof_pool.empty()) {
      hipEvent_t e = prof_pool.back();
      prof_pool.pop_back();
      return e;
    }
    hipEvent_t e;
    hipEventCreate(&e);
    return e;
  }
  std::vector<std::string> names;
  std::vector<int64_t> numel;
  std::vector<const float*> ptr;
  std::vector<float*> gptr;   // parameter-gradient destinations (dptnav_bind_grads), same slots as ptr
  std::vector<float*> gptr_half1;   // second half of a split training batch: scratch destinations in the workspace
  bool opt_train_overlap = true;
  bool bound = false;
  std::string err;
  int stride;
  int dh;
  int num_cus = 256;
  int device_id = 0;   // the HIP device that was current at dptnav_create
  std::vector<PathWeights> pw;   // [2*block + path], rebuilt by dptnav_bind_weights

  int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    err = buf;
    return code;
  }
  int slot(const std::string& n) const {
    for (size_t i = 0; i < names.size(); ++i)
      if (names[i] == n) return (int)i;
    return -1;
  }
  const float* w(const std::string& n) const { return ptr[slot(n)]; }
};

namespace {

inline size_t align64(size_t nfloats) { return (nfloats + 63) & ~(size_t)63; }

// Opt-in per-kernel timing: HIP events recorded on the launch stream around one launch.
// fault injection for the error-path tests (option "inject_fail"): the n-th launch of the GEMM engine / of fcln.hip from now on fails
static int inject_failure(dptnav_ctx* c, const char* what);
struct ProfScope {
  dptnav_ctx* c;
  hipStream_t st;
  ProfRec rec;
  int cat_;
  ProfScope(dptnav_ctx* c_, int cat, hipStream_t st_) : c(c_), st(st_), cat_(cat) {
    if (c->opt_debug_sync) fprintf(stderr, "[dptnav] -> %s\n", kProfNames[cat]);
    if (!c->prof_on) return;
    rec.cat = cat;
    rec.a = c->prof_event();
    rec.b = c->prof_event();
    hipEventRecord(rec.a, st);
  }
  ~ProfScope() {
    if (c->opt_debug_sync) {   // debugging aid (option debug_sync): wait for the launch and say how it ended
      const hipError_t e = hipStreamSynchronize(st);
      fprintf(stderr, "[dptnav] <- %s: %s\n", kProfNames[cat_], hipGetErrorString(e));
    }
    if (!c->prof_on) return;
    hipEventRecord(rec.b, st);
    c->prof_pending.push_back(rec);
  }
};

static int inject_failure(dptnav_ctx* c, const char* what) {
  if (c->opt_inject_fail > 0 && --c->opt_inject_fail == 0) return c->fail(DPTNAV_ERR_INVALID, "%s: injected failure", what);
  return DPTNAV_OK;
}

void build_names(dptnav_ctx* c) {
  const dptnav_config& g = c->cfg;
  const int64_t N = g.num_features, H = g.hidden_dim, k = g.kernel_size_enc;
  auto add = [&](const std::string& n, int64_t e) {
    c->names.push_back(n);
    c->numel.push_back(e);
  };
  if (!g.audio_only) add("gate", 1);
  add("encoder.weight", N * k);
  if (!g.audio_only) {
    add("visual_compression.weight", (int64_t)(g.hidden_video / 2) * g.video_emb_size);
    add("visual_compression.bias", g.hidden_video / 2);
    add("video_ln.weight", g.hidden_video);
    add("video_ln.bias", g.hidden_video);
  }
  for (int b = 0; b < g.num_blocks; ++b)
    for (int p = 0; p < 2; ++p) {
      const std::string pre =
          "dprnn.model." + std::to_string(b) + (p == 0 ? ".intra_chunk_block." : ".inter_chunk_block.");
      const int ndir = (p == 0 || g.bidir) ? 2 : 1;
      const bool dptn = g.arch == 0;
      if (dptn) {
        add(pre + "mha.in_proj_weight", 3 * N * N);
        add(pre + "mha.in_proj_bias", 3 * N);
        add(pre + "mha.out_proj.weight", N * N);
        add(pre + "mha.out_proj.bias", N);
        add(pre + "ln1.weight", N);
        add(pre + "ln1.bias", N);
      }
      add(pre + "rnn.weight_ih_l0", 4 * H * N);
      add(pre + "rnn.weight_hh_l0", 4 * H * H);
      add(pre + "rnn.bias_ih_l0", 4 * H);
      add(pre + "rnn.bias_hh_l0", 4 * H);
      if (ndir == 2) {
        add(pre + "rnn.weight_ih_l0_reverse", 4 * H * N);
        add(pre + "rnn.weight_hh_l0_reverse", 4 * H * H);
        add(pre + "rnn.bias_ih_l0_reverse", 4 * H);
        add(pre + "rnn.bias_hh_l0_reverse", 4 * H);
      }
      add(pre + (dptn ? "ffn.1.weight" : "fc.weight"), N * H * ndir);
      add(pre + (dptn ? "ffn.1.bias" : "fc.bias"), N);
      add(pre + (dptn ? "ln2.weight" : "norm1d.weight"), N);
      add(pre + (dptn ? "ln2.bias" : "norm1d.bias"), N);
    }
  add("dprnn.speakers_separation.0.weight", 1);
  add("dprnn.speakers_separation.1.weight", 2 * N * N);
  add("dprnn.speakers_separation.1.bias", 2 * N);
  add("dprnn.postprocessing.0.weight", N * N);
  add("dprnn.postprocessing.0.bias", N);
  add("decoder.weight", N * k);
  c->ptr.assign(c->names.size(), nullptr);
}

PathWeights path_weights(const dptnav_ctx* c, int block, int path) {
  const std::string pre =
      "dprnn.model." + std::to_string(block) + (path == 0 ? ".intra_chunk_block." : ".inter_chunk_block.");
  PathWeights p{};
  p.ndir = (path == 0 || c->cfg.bidir) ? 2 : 1;
  const bool dptn = c->cfg.arch == 0;
  if (dptn) {
    p.in_w = c->w(pre + "mha.in_proj_weight");
    p.in_b = c->w(pre + "mha.in_proj_bias");
    p.out_w = c->w(pre + "mha.out_proj.weight");
    p.out_b = c->w(pre + "mha.out_proj.bias");
    p.ln1_w = c->w(pre + "ln1.weight");
    p.ln1_b = c->w(pre + "ln1.bias");
  }
  p.w_ih[0] = c->w(pre + "rnn.weight_ih_l0");
  p.w_hh[0] = c->w(pre + "rnn.weight_hh_l0");
  p.b_ih[0] = c->w(pre + "rnn.bias_ih_l0");
  p.b_hh[0] = c->w(pre + "rnn.bias_hh_l0");
  if (p.ndir == 2) {
    p.w_ih[1] = c->w(pre + "rnn.weight_ih_l0_reverse");
    p.w_hh[1] = c->w(pre + "rnn.weight_hh_l0_reverse");
    p.b_ih[1] = c->w(pre + "rnn.bias_ih_l0_reverse");
    p.b_hh[1] = c->w(pre + "rnn.bias_hh_l0_reverse");
  } else {
    p.w_ih[1] = p.w_ih[0];
    p.w_hh[1] = p.w_hh[0];
    p.b_ih[1] = p.b_ih[0];
    p.b_hh[1] = p.b_hh[0];
  }
  p.ffn_w = c->w(pre + (dptn ? "ffn.1.weight" : "fc.weight"));
  p.ffn_b = c->w(pre + (dptn ? "ffn.1.bias" : "fc.bias"));
  p.ln2_w = c->w(pre + (dptn ? "ln2.weight" : "norm1d.weight"));
  p.ln2_b = c->w(pre + (dptn ? "ln2.bias" : "norm1d.bias"));
  return p;
}

int make_plan(dptnav_ctx* c, int B, int64_t T, int Tv, Plan* p) {
  const dptnav_config& g = c->cfg;
  if (B <= 0 || T < g.kernel_size_enc) return c->fail(DPTNAV_ERR_INVALID, "bad shape B=%d T=%lld", B, (long long)T);
  const int64_t N = g.num_features, H = g.hidden_dim;
  p->L = (T - g.kernel_size_enc) / c->stride + 1;
  if (p->L < g.chunk_size)
    return c->fail(DPTNAV_ERR_INVALID, "T=%lld gives L=%lld frames < chunk_size=%d", (long long)T, (long long)p->L,
                   g.chunk_size);
  p->S = (p->L - g.chunk_size) / g.step_size + 1;
  if (g.arch == 0 && g.chunk_size > 256)
    return c->fail(DPTNAV_ERR_INVALID, "chunk_size > 256 unsupported by the attention kernel (S=%lld, K=%d)",
                   (long long)p->S, g.chunk_size);
  {  // token indices are 32-bit inside the kernels: (tokens + dump rows) x widest row must stay below 2^31
    const int64_t widest = std::max<int64_t>(3 * N, 2 * H);
    if (((int64_t)B + 1) * p->S * g.chunk_size * widest >= ((int64_t)1 << 31))
      return c->fail(DPTNAV_ERR_INVALID, "batch too large for 32-bit token indexing (B=%d, %lld tokens per mixture)",
                     B, (long long)(p->S * g.chunk_size));
  }
  p->M = (int64_t)B * p->S * g.chunk_size;
  const int64_t nst_a = ((int64_t)B * p->S + 31) / 32, nst_e = ((int64_t)B * g.chunk_size + 31) / 32;
  const int64_t pre_tiles = std::max(nst_a * g.chunk_size, nst_e * p->S);
  size_t o = 0;
  auto take = [&](size_t n) {
    size_t at = o;
    o += align64(n);
    return at;
  };
  p->queue = take(QUEUE_SLOTS);   // tile-ticket counters of the GEMM-engine launches (zeroed per call)
  p->vid = take((size_t)B * (Tv > 0 ? Tv : 1) * N);
  p->E = take((size_t)B * p->L * N);
  p->X0 = take((size_t)p->M * N);
  p->X1 = take((size_t)p->M * N);
  p->qkv_n = (size_t)p->M * 3 * N;
  p->qkv = take(p->qkv_n);
  p->att_n = (size_t)p->M * N;
  p->att = take(std::max(p->att_n, (size_t)2 * B * p->L * 8));
  p->y1_n = (size_t)p->M * N;
  p->y1 = take(p->y1_n);
  p->pre = take((size_t)2 * pre_tiles * 512 * 32);
  p->hc_n = (size_t)p->M * 2 * H;
  p->hc = take(p->hc_n + (size_t)p->S * g.chunk_size * 2 * H);   // + dump rows for padded sequences
  p->stamps = take((size_t)2 * 2 * (nst_a > nst_e ? nst_a : nst_e) * 4 * 4 * 2);   // u64 [dir][tile][wave][4]
  p->wfold = take((size_t)8 * 2 * N + 8);   // folded decoder weights [G | W_dec^T | bd] (run_tail)
  // fragment-order copies of the attention / FFN weights of every path for the fused attention block (attn_block.h)
  p->wpack = take(g.arch == 0 && N == 128 ? (size_t)2 * g.num_blocks * ATTN_PACK_FLOATS : 0);
  // fragment-order copies of W_ih (both directions) of every path for the K4 launches (gemm_ws.h, ldw == 0)
  p->wih = take((size_t)2 * g.num_blocks * 2 * 4 * H * N);
  // ... and of W_hh in the order of the low-latency recurrence (lstm4.hip, `packed`)
  p->whh4 = take((size_t)2 * g.num_blocks * 2 * 4 * H * H);
  // ... and of the attention in-projection weights of every path, for gemm_t.hip (training forward)
  p->winp = take(g.arch == 0 && N == 128 ? (size_t)2 * g.num_blocks * 3 * N * N : 0);
  p->total = o;
  return DPTNAV_OK;
}

#define LAUNCH_CHECK(c, what)                                                                  \
  do {                                                                                          \
    if ((c)->opt_debug_sync) fprintf(stderr, "[dptnav] launched %s\n", what);                   \
    hipError_t e_ = hipGetLastError();                                                          \
    if (e_ != hipSuccess) return (c)->fail(DPTNAV_ERR_HIP, "%s: %s", what, hipGetErrorString(e_)); \
  } while (0)

template <class Kern>
int set_lds(dptnav_ctx* c, Kern kern, size_t bytes, const char* what) {
  if (bytes > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return c->fail(DPTNAV_ERR_HIP, "%s: set LDS %zu: %s", what, bytes, hipGetErrorString(e));
  }
  return DPTNAV_OK;
}

inline int cap_grid(int64_t ntiles, int cap) { return (int)(ntiles < cap ? ntiles : cap); }

// ---- attention dispatch over the number of 32-key blocks ----------------------------------------
template <int DH, int NKB>
int launch_attn_nkb(dptnav_ctx* c, const float* qkv, float* att, int N, const SeqGeom& g, int heads,
                    hipStream_t st, DropCfg drop, float2* stats, unsigned long long* mask) {
  auto kern = attention_kernel<DH, NKB>;
  const size_t lds = AttnShape<DH>::lds_bytes(NKB);
  static PerDeviceOnce ready;   // per instantiation and device
  if (!ready.done(c->device_id)) {
    if (int rc = set_lds(c, kern, lds, "attention")) return rc;
    ready.set(c->device_id);
  }
  const float scale = 1.4426950408889634f / sqrtf((float)DH);
  ProfScope ps(c, CAT_ATTN, st);
  hipLaunchKernelGGL(kern, dim3(g.nseq, heads), dim3(64 * NKB), lds, st, qkv, att, N, heads, g, scale, drop, stats, mask);
  LAUNCH_CHECK(c, "attention");
  return DPTNAV_OK;
}
template <int DH>
int launch_attn(dptnav_ctx* c, const float* qkv, float* att, int N, const SeqGeom& g, int heads, hipStream_t st,
                DropCfg drop = DropCfg{0u, 0u, 1.0f}, float2* stats = nullptr, unsigned long long* mask = nullptr) {
  switch ((g.len + 31) / 32) {
    case 1: return launch_attn_nkb<DH, 1>(c, qkv, att, N, g, heads, st, drop, stats, mask);
    case 2: return launch_attn_nkb<DH, 2>(c, qkv, att, N, g, heads, st, drop, stats, mask);
    case 3: return launch_attn_nkb<DH, 3>(c, qkv, att, N, g, heads, st, drop, stats, mask);
    case 4: return launch_attn_nkb<DH, 4>(c, qkv, att, N, g, heads, st, drop, stats, mask);
    case 5: return launch_attn_nkb<DH, 5>(c, qkv, att, N, g, heads, st, drop, stats, mask);
    case 6: return launch_attn_nkb<DH, 6>(c, qkv, att, N, g, heads, st, drop, stats, mask);
    case 7: return launch_attn_nkb<DH, 7>(c, qkv, att, N, g, heads, st, drop, stats, mask);
    case 8: return launch_attn_nkb<DH, 8>(c, qkv, att, N, g, heads, st, drop, stats, mask);
  }
  // longer sequences (inter-chunk path of utterances beyond ~7 s): streaming-softmax kernel, inference only
  if (drop.thresh != 0u) return c->fail(DPTNAV_ERR_INVALID, "attention: dropout needs sequence length <= 256 (got %d)", g.len);
  {
    ProfScope ps(c, CAT_ATTN, st);
    hipLaunchKernelGGL(attention_long_kernel<DH>, dim3(g.nseq, heads), dim3(256), 0, st, qkv, att, N, g,
                       1.4426950408889634f / sqrtf((float)DH));
    LAUNCH_CHECK(c, "attention (long)");
  }
  return DPTNAV_OK;
}

// ---- generic GEMM-engine launch -------------------------------------------------------------------
constexpr int BWD_SLAB_WGS = 256;       // workgroups of one wgrad / colsum launch (one partial slab each)

// The engine is persistent (grid-stride over tiles, weights loaded once per workgroup), so the grid is
// sized to what is co-resident: CUs x blocks/CU from the occupancy query, queried once per instantiation.
// RD = WgradRider<...> (gemm_ws.h): the launch also forms the weight gradient of the layer whose data gradient it is;
// one workgroup per CU then (as many partial tiles as the stand-alone weight-gradient kernels leave), grid in *grid_used.
template <int KIN, int NT, int WR, int WC, bool WT = false, bool SPLIT = false, class AL, class EP, class RD = NoRider>
int launch_gemm(dptnav_ctx* c, Run& run, int cat, const char* what, const float* W, int64_t ntiles, int colgroups,
                const AL& al, const EP& ep, const float* Walt = nullptr, int ldw = KIN, int* grid_used = nullptr,
                const RD& rider = RD{}) {
  hipStream_t st = run.st;
  if (int rc = inject_failure(c, what)) return rc;
  if (run.slot + colgroups > QUEUE_SLOTS) return c->fail(DPTNAV_ERR_INVALID, "%s: ticket counters exhausted", what);
  auto kern = gemm_ws_kernel<KIN, NT, WR, WC, AL, EP, WT, SPLIT, RD>;
  const size_t lds = GemmShape<KIN, NT, WR, WC>::lds_bytes(EP::DIRECT, SPLIT, rider_kk<RD>::value * (RD::ON ? 1 : 0));
  static std::atomic<int> resident_dev[64];  // per instantiation and device (zero-initialised; idempotent fill)
  int resident = resident_dev[c->device_id & 63].load(std::memory_order_acquire);
  if (resident == 0) {
    if (int rc = set_lds(c, kern, lds, what)) return rc;
    int per_cu = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, lds);
    if (e != hipSuccess || per_cu < 1) return c->fail(DPTNAV_ERR_HIP, "%s: occupancy query: %s", what, hipGetErrorString(e));
    // two workgroups per CU cover each other's barriers; a third only adds a weight-load prologue per launch
    // (out-projection GEMM: 0.157 ms with 512 workgroups, 0.160 with 768)
    resident = (RD::ON ? 1 : std::min(per_cu, 2)) * c->num_cus;
    if (RD::ON && resident > BWD_SLAB_WGS) resident = BWD_SLAB_WGS;   // one partial tile per workgroup in the slab region
    resident_dev[c->device_id & 63].store(resident, std::memory_order_release);
  }
  int gx = resident / colgroups;
  if (gx < 1) gx = 1;
  ProfScope ps(c, cat, st);
  if (grid_used) *grid_used = cap_grid(ntiles, gx);
  unsigned* const queue = run.take_queue(colgroups);
  hipLaunchKernelGGL(kern, dim3(cap_grid(ntiles, gx), colgroups), dim3(256), lds, st, W, Walt, ldw, (int)ntiles,
                     c->opt_deterministic ? nullptr : queue, al, ep, rider);
  LAUNCH_CHECK(c, what);
  return DPTNAV_OK;
}

// Where one path keeps its intermediates: the shared workspace slice (inference) or a per-path tape (training, where
// the backward needs them: qkv, att, y1, raw h, post-activation gates and cell states).
struct PathBufs {
  float *qkv, *att, *y1, *pre, *hc, *gates, *cst;
  bool train;
  float* astats = nullptr;   // training: where the attention forward leaves its softmax statistics
  float* amask = nullptr;    // training: ... and its dropout keep decisions as bit masks (attention.h)
  // training, option ln_tape: normalised rows and 1/sigma of LayerNorm 1 / LayerNorm 2 for the backward (EpiBiasResLNSave)
  float *zn1 = nullptr, *rs1 = nullptr, *zn2 = nullptr, *rs2 = nullptr;
};
inline PathBufs inference_bufs(const Run& run) {
  return PathBufs{run.ws + run.pl.qkv, run.ws + run.pl.att, run.ws + run.pl.y1, run.ws + run.pl.pre,
                  run.ws + run.pl.hc, nullptr, nullptr, false};
}

// K4/K5 tile height: 32 sequences per workgroup, or 16 when that still fits the chip in one round (sub-batch launches of
// dptnav_forward, the halves of a split training batch): same CU-time, half the serial time of the recurrence.  The
// training backward asks the same question (the tape layout follows the tile height).
// (lstm16 addresses hc with 32-bit byte offsets: rows [0, M + S*K) x ldh floats)
inline bool lstm_use16(const dptnav_ctx* c, const SeqGeom& geom, int ndir, int64_t M) {
  const int nst16 = (geom.nseq + 15) / 16;
  return c->opt_lstm16 && nst16 * ndir <= c->num_cus &&
         (uint64_t)(M + (int64_t)geom.S * geom.K) * (uint64_t)(ndir * LSTM_H) * 4u < (1ull << 32);
}

// ---- one TransformerDPRNN (dptn.py:36-52) ---------------------------------------------------------
// chain: bit 0 -- the attention block PRODUCES its input rows itself, as the FFN half (K6) of the previous path (its hc /
// y1 are still in the workspace; attn_block.hip prologue); bit 1 -- this path's K6 is left to the next path's block.
constexpr int CHAIN_PRO = 1, CHAIN_SKIP_FFN = 2;
template <int N>
bool path_fusable(const dptnav_ctx* c, int path, int B, int S) {
  const SeqGeom geom = make_geom(path, B, S, c->cfg.chunk_size);
  // N = 128: attn_block.hip (fp32 and split variants); N = 64: attn_block64.hip (fp32 only)
  return c->cfg.arch == 0 && (N == 128 || (N == 64 && !c->opt_split_bf16)) && c->opt_fuse_attn && geom.len <= ATTN_BLOCK_MAX_LEN &&
         c->cfg.num_heads == 4;
}

// fragment-order copies of the attention / FFN weights of paths [first, first + n) for the fused attention block, from
// the weights as they are NOW (an optimizer may have stepped since the last call)
static int pack_attn_weights(dptnav_ctx* c, hipStream_t st, int first, int n, float* dst) {
  std::vector<AttnPackSrc> src((size_t)n);
  for (int i = 0; i < n; ++i) {
    const PathWeights& w = c->pw[first + i];
    src[(size_t)i] = AttnPackSrc{w.in_w, w.out_w, w.ndir == 2 ? w.ffn_w : nullptr};   // [128][256] only with both directions
  }
  const int rc = attn_pack_launch(st, src.data(), n, dst);
  if (rc != 0) return c->fail(DPTNAV_ERR_HIP, "attention weight pack: %s", hipGetErrorString((hipError_t)rc));
  return DPTNAV_OK;
}

// fragment-order copies of W_ih (forward, reverse) of paths [first, first + n) -> dst[(path - first)][2][512 x N]
template <int N>
static int pack_wih(dptnav_ctx* c, hipStream_t st, int first, int n, float* dst) {
  GemmPackArgs a;
  for (int i = 0; i < GEMM_PACK_MAX; ++i) a.src[i] = nullptr;
  if (2 * n > GEMM_PACK_MAX) return c->fail(DPTNAV_ERR_INVALID, "W_ih pack: %d paths in one launch", n);
  for (int i = 0; i < n; ++i) {
    const PathWeights& w = c->pw[first + i];
    a.src[2 * i] = w.w_ih[0];
    a.src[2 * i + 1] = w.ndir == 2 ? w.w_ih[1] : nullptr;
  }
  a.rows = 4 * LSTM_H;
  a.K = N;
  hipLaunchKernelGGL(gemm_pack_rows_kernel, dim3(16, 2 * n), dim3(256), 0, st, a, dst);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return c->fail(DPTNAV_ERR_HIP, "W_ih pack: %s", hipGetErrorString(e));
  return DPTNAV_OK;
}

// all paths of the model into run.ws + pl.wih (dptnav_forward / dptnav_train_forward: once per pass and sub-batch)
template <int N>
static int pack_wih_all(dptnav_ctx* c, Run& run) {
  if (!c->opt_pack_wih) return DPTNAV_OK;
  const int npaths = 2 * c->cfg.num_blocks;
  for (int first = 0; first < npaths; first += GEMM_PACK_MAX / 2) {
    const int n = std::min(GEMM_PACK_MAX / 2, npaths - first);
    if (int rc = pack_wih<N>(c, run.st, first, n, run.ws + run.pl.wih + (size_t)first * 2 * (4 * LSTM_H * N))) return rc;
  }
  run.packed_wih = true;
  return DPTNAV_OK;
}

// fragment-order copies of the attention in-projection weights of paths [first, first + n) -> dst[(path - first)][384 x 128]
// (gemm_t.hip, training forward)
static int pack_inw(dptnav_ctx* c, hipStream_t st, int first, int n, float* dst) {
  GemmPackArgs a;
  for (int i = 0; i < GEMM_PACK_MAX; ++i) a.src[i] = nullptr;
  if (n > GEMM_PACK_MAX) return c->fail(DPTNAV_ERR_INVALID, "W_in pack: %d paths in one launch", n);
  for (int i = 0; i < n; ++i) a.src[i] = c->pw[first + i].in_w;
  a.rows = 3 * 128;
  a.K = 128;
  hipLaunchKernelGGL(gemm_pack_rows_kernel, dim3(16, n), dim3(256), 0, st, a, dst);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return c->fail(DPTNAV_ERR_HIP, "W_in pack: %s", hipGetErrorString(e));
  return DPTNAV_OK;
}
static int pack_inw_all(dptnav_ctx* c, Run& run) {
  if (!c->opt_gemm_t || c->cfg.num_features != 128 || c->cfg.arch != 0) return DPTNAV_OK;
  const int npaths = 2 * c->cfg.num_blocks;
  for (int first = 0; first < npaths; first += GEMM_PACK_MAX) {
    const int n = std::min(GEMM_PACK_MAX, npaths - first);
    if (int rc = pack_inw(c, run.st, first, n, run.ws + run.pl.winp + (size_t)first * 3 * 128 * 128)) return rc;
  }
  run.packed_inw = true;
  return DPTNAV_OK;
}

// fcln.hip for one Linear (+ LayerNorm) launch.  Returns 1 when the launch went out, 0 when fcln_launch does not take the shape
// (hipErrorInvalidValue, fcln.h: the caller then uses the GEMM engine; the sticky error is cleared), a negative DPTNAV error otherwise.
static int try_fcln(dptnav_ctx* c, hipStream_t st, const FclnArgs& fa, int cat, const char* what) {
  if (int rc = inject_failure(c, what)) return rc < 0 ? rc : -rc;
  ProfScope ps(c, cat, st);
  const int rc = fcln_launch(st, fa, c->num_cus);
  if (rc == 0) return 1;
  if (rc == (int)hipErrorInvalidValue) {
    (void)hipGetLastError();
    return 0;
  }
  const int e = c->fail(DPTNAV_ERR_HIP, "%s: %s", what, hipGetErrorString((hipError_t)rc));
  return e < 0 ? e : -e;
}

// dgrad_t.hip for one wide data-gradient launch (out = addend + A W, kin = 512 or 384; option dgrad_t).  `Wpacked`: the
// fragment-order copy pack_bwd_weights made on the same stream.  Returns like try_fcln: 1 launched, 0 not taken (the caller uses the
// GEMM engine), negative DPTNAV error.
static int try_dgrad_t(dptnav_ctx* c, Run& run, int cat, const char* what, const float* A, int lda, int kin, const float* Wpacked,
                       const float* addend, float* out, int64_t M) {
  if (!c->opt_dgrad_t) return 0;
  if (int rc = inject_failure(c, what)) return rc < 0 ? rc : -rc;
  if (run.slot + 1 > QUEUE_SLOTS) {
    const int e = c->fail(DPTNAV_ERR_INVALID, "%s: ticket counters exhausted", what);
    return e < 0 ? e : -e;
  }
  DgradTArgs a;
  a.A = A; a.lda = lda; a.W = Wpacked; a.addend = addend; a.out = out; a.M = M; a.kin = kin;
  unsigned* const queue = run.take_queue(1);
  a.queue = c->opt_deterministic ? nullptr : queue;
  ProfScope ps(c, cat, run.st);
  const int rc = dgrad_t_launch(run.st, a, c->num_cus);
  if (rc == 0) return 1;
  if (rc == (int)hipErrorInvalidValue) {
    (void)hipGetLastError();
    run.slot -= 1;
    return 0;
  }
  const int e = c->fail(DPTNAV_ERR_HIP, "%s: %s", what, hipGetErrorString((hipError_t)rc));
  return e < 0 ? e : -e;
}
// Fragment-order weight copies the backward's dedicated kernels read (dgrad_t.hip, dgrad_r.hip): one slot of BWD_PACK_FLOATS per
// path in the backward workspace -- [W_ih fwd | W_ih rev | in_proj_weight | out_proj.weight | ffn.1.weight].  dptnav_train_backward
// fills the slots of ALL paths with four launches at its start (weights change every step, so once per step); the path-level entry
// point packs the one path it runs into slot 0.  (Packed where they are used -- two to four 10-us launches in front of every
// data-gradient launch -- they were 120 launches and 0.6 ms of each stream's chain per step.)
constexpr size_t BWD_PACK_WIH = 0, BWD_PACK_INW = (size_t)2 * 512 * 128, BWD_PACK_OUTW = BWD_PACK_INW + (size_t)384 * 128,
                 BWD_PACK_FFNW = BWD_PACK_OUTW + (size_t)128 * 128, BWD_PACK_FLOATS = BWD_PACK_FFNW + (size_t)128 * 256;
static int pack_bwd_weights(dptnav_ctx* c, hipStream_t st, int first, int n, float* dst) {      // paths [first, first + n) -> slots 0 .. n - 1
  if (c->cfg.num_features != 128 || c->cfg.arch != 0 || (!c->opt_dgrad_t && !c->opt_dgrad_r)) return DPTNAV_OK;
  if (2 * n > DGRAD_PACK_MAX || n > DGRAD_R_PACK_MAX) return c->fail(DPTNAV_ERR_INVALID, "backward weight pack: %d paths in one launch", n);
  const float* src[DGRAD_PACK_MAX];
  long long off[DGRAD_PACK_MAX];
  int rc = 0;
  if (c->opt_dgrad_t) {
    for (int i = 0; i < n; ++i) {
      const PathWeights& w = c->pw[first + i];
      src[2 * i] = w.w_ih[0];
      src[2 * i + 1] = w.ndir == 2 ? w.w_ih[1] : nullptr;
      off[2 * i] = (long long)(i * BWD_PACK_FLOATS + BWD_PACK_WIH);
      off[2 * i + 1] = off[2 * i] + 512 * 128;
    }
    rc = dgrad_t_pack_launch(st, src, off, 2 * n, 512, dst);
    for (int i = 0; i < n && rc == 0; ++i) {
      src[i] = c->pw[first + i].in_w;
      off[i] = (long long)(i * BWD_PACK_FLOATS + BWD_PACK_INW);
    }
    if (rc == 0) rc = dgrad_t_pack_launch(st, src, off, n, 384, dst);
  }
  if (rc == 0 && c->opt_dgrad_r) {
    for (int i = 0; i < n; ++i) {
      src[i] = c->pw[first + i].out_w;
      off[i] = (long long)(i * BWD_PACK_FLOATS + BWD_PACK_OUTW);
    }
    rc = dgrad_r_pack_launch(st, src, off, n, 128, dst);
    for (int i = 0; i < n; ++i) {
      const PathWeights& w = c->pw[first + i];
      src[i] = w.ndir == 2 ? w.ffn_w : nullptr;      // [128][256] only with both directions
      off[i] = (long long)(i * BWD_PACK_FLOATS + BWD_PACK_FFNW);
    }
    if (rc == 0) rc = dgrad_r_pack_launch(st, src, off, n, 256, dst);
  }
  if (rc != 0) return c->fail(DPTNAV_ERR_HIP, "backward weight pack: %s", hipGetErrorString((hipError_t)rc));
  return DPTNAV_OK;
}

// dgrad_r.hip for one K = 128 data gradient with its weight / bias gradient riding (option dgrad_r): `wpacked` from pack_bwd_weights;
// leaves the partial tiles in `slab` / `colslab` and their number in *grid.  Returns like try_fcln.
static int try_dgrad_r(dptnav_ctx* c, Run& run, int cat, const char* what, const float* A, const float* wpacked, int nout, bool gate,
                       const float* X, float* out, int64_t M, float* slab, float* colslab, int max_slabs, int* grid) {
  if (!c->opt_dgrad_r) return 0;
  if (int rc = inject_failure(c, what)) return rc < 0 ? rc : -rc;
  if (run.slot + 1 > QUEUE_SLOTS) {
    const int e = c->fail(DPTNAV_ERR_INVALID, "%s: ticket counters exhausted", what);
    return e < 0 ? e : -e;
  }
  DgradRArgs a;
  a.A = A; a.Wpacked = wpacked; a.X = X; a.out = out; a.M = M; a.nout = nout; a.relu_gate = gate;
  a.slab = slab; a.colslab = colslab; a.max_slabs = max_slabs;
  unsigned* const queue = run.take_queue(1);
  a.queue = c->opt_deterministic ? nullptr : queue;
  ProfScope ps(c, cat, run.st);
  const int rc = dgrad_r_launch(run.st, a, c->num_cus, grid);
  if (rc == 0) return 1;
  run.slot -= 1;
  if (rc == (int)hipErrorInvalidValue) {
    (void)hipGetLastError();
    return 0;
  }
  const int e = c->fail(DPTNAV_ERR_HIP, "%s: %s", what, hipGetErrorString((hipError_t)rc));
  return e < 0 ? e : -e;
}

template <int N>
int run_path(dptnav_ctx* c, Run& run, int block, int path, const float* x_in, float* x_out, int B, int S,
             const PathBufs* bufs = nullptr, int chain = 0) {
  float* ws = run.ws;
  const Plan& pl = run.pl;
  hipStream_t st = run.st;
  const PathBufs pb = bufs ? *bufs : inference_bufs(run);
  constexpr int WR = N == 128 ? 1 : 2, WC = N == 128 ? 4 : 2, GROUP = N / 4, DH = N / 4;
  constexpr int BM = 32 * WR;
  const dptnav_config& g = c->cfg;
  const PathWeights& w = c->pw[2 * block + path];
  const int K = g.chunk_size;
  const int64_t M = (int64_t)B * S * K;
  const SeqGeom geom = make_geom(path, B, S, K);
  float *qkv = pb.qkv, *att = pb.att, *y1 = pb.y1, *pre = pb.pre, *hc = pb.hc;
  const int64_t ntiles = (M + BM - 1) / BM;
  const bool dptn = g.arch == 0;
  const float* lstm_in = dptn ? y1 : x_in;   // DPRNN feeds the chunk tokens straight into the LSTM (dprnn.py:37-40)

  // K1 + K2 + K3 in one launch (inference, N = 128, sequences of <= 160 positions): QKV and the attention output never
  // leave the chip -- 1 kB of HBM traffic per token instead of 5.5 kB
  const bool fused = dptn && (!pb.train || c->opt_train_fuse_probe) && path_fusable<N>(c, path, B, S);
  if (fused && (chain & CHAIN_PRO) && c->pw[2 * block + path - 1].ndir != 2)
    return c->fail(DPTNAV_ERR_INVALID, "internal: FFN prologue needs both LSTM directions of the previous path");
  if (fused && N == 64) {
    ProfScope ps(c, CAT_ATTN, st);
    AttnFfnPrologue pro{};
    if (chain & CHAIN_PRO) {
      const PathWeights& pw = c->pw[2 * block + path - 1];
      pro = AttnFfnPrologue{hc, pw.ffn_w, pw.ffn_b, pw.ln2_w, pw.ln2_b};
    }
    const int rc = attn_block64_launch(st, x_in, w.in_w, w.in_b, w.out_w, w.out_b, w.ln1_w, w.ln1_b, y1, geom,
                                       (chain & CHAIN_PRO) ? &pro : nullptr);
    if (rc != 0) return c->fail(DPTNAV_ERR_HIP, "attention block (N = 64): %s", hipGetErrorString((hipError_t)rc));
  } else if (fused) {
    ProfScope ps(c, CAT_ATTN, st);
    AttnFfnPrologue pro{};
    if (chain & CHAIN_PRO) {
      const PathWeights& pw = c->pw[2 * block + path - 1];
      pro = AttnFfnPrologue{hc, pw.ffn_w, pw.ffn_b, pw.ln2_w, pw.ln2_b};
    }
    const float* wpack = nullptr;
    if (!c->opt_split_bf16) {
      float* pk = ws + pl.wpack + (size_t)(2 * block + path) * ATTN_PACK_FLOATS;
      if (!run.packed) {   // stage entry points: this path only (dptnav_forward packs all paths in one launch)
        if (int rc = pack_attn_weights(c, st, 2 * block + path, 1, pk)) return rc;
      }
      wpack = pk;
      if (chain & CHAIN_PRO) pro.wf = pk - ATTN_PACK_FLOATS + ATTN_PACK_IN + ATTN_PACK_OUT;   // previous path's ffn.1 segment
    }
    const int rc = (c->opt_attn_v2 && !c->opt_split_bf16)
                       ? attn_block2_launch(st, x_in, w.in_b, w.out_b, w.ln1_w, w.ln1_b, y1, geom, (chain & CHAIN_PRO) ? &pro : nullptr, wpack)
                       : attn_block_launch(st, x_in, w.in_w, w.in_b, w.out_w, w.out_b, w.ln1_w, w.ln1_b, y1, geom, c->opt_split_bf16,
                                           (chain & CHAIN_PRO) ? &pro : nullptr, wpack);
    if (rc != 0) return c->fail(DPTNAV_ERR_HIP, "attention block: %s", hipGetErrorString((hipError_t)rc));
  } else if (chain & CHAIN_PRO) {
    return c->fail(DPTNAV_ERR_INVALID, "internal: FFN prologue requested for an unfused attention block");
  }
  // K1: qkv = x W_in^T + b_in                                  (nn.MultiheadAttention in-projection)
  if (dptn && !fused) {
    int done = 0;
    if constexpr (N == 128) {
      if (pb.train && c->opt_gemm_t) {      // training forward: gemm_t.hip on the fragment-order copy of W_in (packed here, on the stream)
        if (int rc = inject_failure(c, "qkv gemm")) return rc;
        if (run.slot + 1 > QUEUE_SLOTS) return c->fail(DPTNAV_ERR_INVALID, "qkv gemm: ticket counters exhausted");
        float* wp = ws + pl.winp + (run.packed_inw ? (size_t)(2 * block + path) * 3 * N * N : 0);
        if (!run.packed_inw)
          if (int rc = pack_inw(c, st, 2 * block + path, 1, wp)) return rc;
        GemmTArgs ga;
        ga.A = x_in; ga.Wpacked = wp; ga.bias = w.in_b; ga.out = qkv; ga.M = M; ga.nout = 3 * N;
        unsigned* const queue = run.take_queue(1);
        ga.queue = c->opt_deterministic ? nullptr : queue;
        ProfScope ps(c, CAT_QKV, st);
        const int rc = gemm_t_launch(st, ga, c->num_cus);
        if (rc == 0) done = 1;
        else if (rc == (int)hipErrorInvalidValue) { (void)hipGetLastError(); run.slot -= 1; }
        else return c->fail(DPTNAV_ERR_HIP, "qkv gemm: %s", hipGetErrorString((hipError_t)rc));
      }
    }
    if (!done) {
      ALoadDense al{x_in, M, N, BM};
      EpiBiasStore ep{qkv, w.in_b, M, 3 * N, BM, 3 * N};
      if (int rc = launch_gemm<N, 3, WR, WC>(c, run, CAT_QKV, "qkv gemm", w.in_w, ntiles, 1, al, ep)) return rc;
    }
  }
  // K2: softmax(Q K^T / sqrt(dh)) V per (sequence, head)
  if (dptn && !fused)
    if (int rc = launch_attn<DH>(c, qkv, att, N, geom, g.num_heads, st, c->drop_cfg(block, path, pb.train, run.half),
                                 reinterpret_cast<float2*>(pb.astats), reinterpret_cast<unsigned long long*>(pb.amask)))
      return rc;
  // K3: y1 = LN1(att W_o^T + b_o + x)                           (dptn.py:46-47)
  if (dptn && !fused) {
    ALoadDense al{att, M, N, BM};
    bool done = false;
    {
      if (pb.train && pb.zn1 && N == 128 && c->opt_fcln) {   // 16-token tiles, three workgroups per CU (fcln.hip)
        FclnArgs fa{att, w.out_w, w.out_b, w.ln1_w, w.ln1_b, x_in, y1, pb.zn1, pb.rs1, M, N, N, /*pre_res*/ true, /*act*/ 0};
        const int r = try_fcln(c, st, fa, CAT_OUTPROJ, "fcln (out-projection)");
        if (r < 0) return -r;
        done = r > 0;
      }
      if (!done && pb.train && pb.zn1) {   // + zn / rstd on the tape for the LayerNorm backward
        EpiBiasResLNSave<GROUP> ep{y1, w.out_b, x_in, w.ln1_w, w.ln1_b, M, N, BM, pb.zn1, pb.rs1};
        if (int rc = launch_gemm<N, 1, WR, WC>(c, run, CAT_OUTPROJ, "out-proj gemm (tape)", w.out_w, ntiles, 1, al, ep)) return rc;
        done = true;
      }
    }
    if (!done) {
      EpiBiasResLN<GROUP> ep{y1, w.out_b, x_in, w.ln1_w, w.ln1_b, M, N, BM};
      if (int rc = launch_gemm<N, 1, WR, WC>(c, run, CAT_OUTPROJ, "out-proj gemm", w.out_w, ntiles, 1, al, ep)) return rc;
    }
  }
  const int nst16 = (geom.nseq + 15) / 16;
  const bool use16 = lstm_use16(c, geom, w.ndir, M);
  const bool split = c->opt_split_bf16 && !pb.train;
  // low-latency recurrence on 4-sequence tiles (reads PRE16): small launches, where 16-sequence tiles leave the chip idle.
  // A step of 4 sequences takes ~1.25 us against ~4.3 us for 16 (alone on the chip, bs = 1: 0.19 ms per launch instead of
  // 0.64 ms), so up to a little over one round of the chip it is the shorter launch AND the smaller CU-time.  Measured
  // (profiles/r03_lstm4_sweep.txt): faster for sub-batches of up to 4 mixtures at 4 s (1.1 rounds; B = 8: 18.0 -> 16.3 ms),
  // slower inside the B = 16 forward when its 5-mixture sub-batches (1.38 rounds) take it (30.5 -> 31.1 ms).
  const int nst4 = (geom.nseq + 3) / 4;
  // Round 5 (tools/small_batch_sweep.py, profiles/r05_small_batch_sweep.txt): while the whole call is at most 10 mixtures -- two
  // sub-batches of up to 5, which do not fill the chip between them -- launches of up to 1.5 rounds are still faster on 4-sequence
  // tiles (B = 8: 16.47 -> 15.68 ms, B = 10: 19.64 -> 19.12 ms); from B = 12 on the chip is full and the 16-sequence tiles' lower
  // CU-time wins (23.07 vs 22.2 ms).
  const int rounds20 = (run.batch_total > 0 && run.batch_total <= 10) ? 30 : 23;      // launch size limit in twentieths of a round
  const bool use4 = use16 && !split && !pb.train && !c->opt_lstm_stamps &&
                    (c->opt_lstm4 == 2 || (c->opt_lstm4 == 1 && 20 * nst4 * w.ndir <= rounds20 * c->num_cus));
  // num_features = 64: input projection inside the recurrence (lstm16x.hip) -- no K4 launch, no pre-activation tensor
  // (the 128-feature kernel addresses hc and its input rows with 32-bit byte offsets: both tensors below 4 GiB)
  const bool fits32 = (uint64_t)(M + (int64_t)geom.S * geom.K) * (uint64_t)(w.ndir * LSTM_H) * 4u < (1ull << 32);
  const bool usex = (N == 64 ? c->opt_fuse_pre : (run.fuse128 && fits32)) && c->opt_lstm16 && !pb.train && !split && !use4 && !c->opt_lstm_stamps;
  // K4: LSTM pre-activations for every (direction, sequence tile, position), in accumulator-fragment order
  if (!usex) {
    ALoadSeqTile al{lstm_in, N, geom};
    const int64_t nt4 = (int64_t)geom.nst * geom.len;
    int rc;
    // fp32 form: W_ih from its fragment-order copy (made for all paths at the start of the pass, or here for this path)
    const float *wih0 = w.w_ih[0], *wih1 = w.w_ih[1];
    int ldw_ih = N;
    if (c->opt_pack_wih && !split) {
      float* pk = ws + pl.wih + (size_t)(2 * block + path) * 2 * (4 * LSTM_H * N);
      if (!run.packed_wih)
        if (int rcp = pack_wih<N>(c, st, 2 * block + path, 1, pk)) return rcp;
      wih0 = pk;
      wih1 = pk + 4 * LSTM_H * N;
      ldw_ih = 0;
    }
    if (use16 && split) {   // opt-in split-precision mode (bf16 hi/lo operands, fp32 accumulation)
      EpiLstmPre16 ep{pre, {w.b_ih[0], w.b_ih[1]}, {w.b_hh[0], w.b_hh[1]}, geom, nst16};
      rc = launch_gemm<N, 4, 1, 4, false, true>(c, run, CAT_LSTM_PRE, "lstm-pre gemm (split)", w.w_ih[0], nt4, w.ndir, al, ep, w.w_ih[1]);
    } else if (use16) {
      EpiLstmPre16 ep{pre, {w.b_ih[0], w.b_ih[1]}, {w.b_hh[0], w.b_hh[1]}, geom, nst16};
      rc = launch_gemm<N, 4, 1, 4>(c, run, CAT_LSTM_PRE, "lstm-pre gemm", wih0, nt4, w.ndir, al, ep, wih1, ldw_ih);
    } else if (split) {
      EpiLstmPre ep{pre, {w.b_ih[0], w.b_ih[1]}, {w.b_hh[0], w.b_hh[1]}, geom};
      rc = launch_gemm<N, 4, 1, 4, false, true>(c, run, CAT_LSTM_PRE, "lstm-pre gemm (split)", w.w_ih[0], nt4, w.ndir, al, ep, w.w_ih[1]);
    } else {
      EpiLstmPre ep{pre, {w.b_ih[0], w.b_ih[1]}, {w.b_hh[0], w.b_hh[1]}, geom};
      rc = launch_gemm<N, 4, 1, 4>(c, run, CAT_LSTM_PRE, "lstm-pre gemm", wih0, nt4, w.ndir, al, ep, wih1, ldw_ih);
    }
    if (rc) return rc;
  }
  // K5: recurrence, both directions concurrently; writes ReLU(h) (ffn[0], dptn.py:31)
  if (run.lstm_wait && hipStreamWaitEvent(st, run.lstm_wait, 0) != hipSuccess)
    return c->fail(DPTNAV_ERR_HIP, "lstm stagger wait");
  if (usex) {
    ProfScope ps(c, CAT_LSTM, st);
    const int rc = lstm16x_launch(N, dptn, nst16, w.ndir, st, lstm_in, N, w.w_ih, w.b_ih, w.b_hh, w.w_hh, hc, w.ndir * LSTM_H,
                                  M, geom);
    if (rc != 0) return c->fail(DPTNAV_ERR_HIP, "lstm16x: %s", hipGetErrorString((hipError_t)rc));
  } else if (use16 && split) {
    ProfScope ps(c, CAT_LSTM, st);
    const int rc = lstm16s_launch(c->cfg.arch == 0, nst16, w.ndir, st, pre, w.w_hh[0], w.w_hh[1], hc, w.ndir * LSTM_H, (int)M, geom);
    if (rc != 0) return c->fail(DPTNAV_ERR_HIP, "lstm16s: %s", hipGetErrorString((hipError_t)rc));
  } else if (use4) {
    const float *whh0 = w.w_hh[0], *whh1 = w.w_hh[1];
    if (c->opt_pack_whh && 2 * (int)c->pw.size() <= LSTM4_PACK_MAX) {
      float* base = ws + pl.whh4;
      if (!run.packed_whh4) {   // first 4-sequence recurrence of this pass: copy every path's W_hh, one launch
        const float* src[LSTM4_PACK_MAX];
        for (size_t i = 0; i < c->pw.size(); ++i) {
          src[2 * i] = c->pw[i].w_hh[0];
          src[2 * i + 1] = c->pw[i].ndir == 2 ? c->pw[i].w_hh[1] : nullptr;
        }
        const int rcp = lstm4_pack_launch(st, src, 2 * (int)c->pw.size(), base);
        if (rcp != 0) return c->fail(DPTNAV_ERR_HIP, "lstm4 W_hh pack: %s", hipGetErrorString((hipError_t)rcp));
        run.packed_whh4 = true;
      }
      whh0 = base + (size_t)(2 * (2 * block + path)) * (4 * LSTM_H * LSTM_H);
      whh1 = whh0 + 4 * LSTM_H * LSTM_H;
    }
    ProfScope ps(c, CAT_LSTM, st);
    const int rc = lstm4_launch(c->cfg.arch == 0, nst4, nst16, w.ndir, st, pre, whh0, whh1, hc,
                                w.ndir * LSTM_H, (int)M, geom, whh0 != w.w_hh[0]);
    if (rc != 0) return c->fail(DPTNAV_ERR_HIP, "lstm4: %s", hipGetErrorString((hipError_t)rc));
  } else if (use16) {
    // lstm_stamps: diagnostic builds; lstm_diag > 0 are timing-only ablations (wrong results), see lstm16.hip
    const int variant = pb.train ? L16_VARIANT_TRAIN : (c->opt_lstm_stamps ? 1 + c->opt_lstm_diag : 0);
    unsigned long long* stamps = reinterpret_cast<unsigned long long*>(ws + pl.stamps);   // room: nst16 <= 2 nst
    ProfScope ps(c, CAT_LSTM, st);
    // DPTN inference stores ReLU(h) (ffn = ReLU -> Linear, dptn.py:31); training keeps raw h for the tape, DPRNN
    // feeds fc directly
    const int rc = lstm16_launch(variant, c->cfg.arch == 0 && !pb.train, nst16, w.ndir, st, pre, w.w_hh[0], w.w_hh[1], hc,
                                 w.ndir * LSTM_H, (int)M, geom, stamps, pb.gates, pb.cst);
    if (rc != 0) return c->fail(DPTNAV_ERR_HIP, "lstm16: %s", hipGetErrorString((hipError_t)rc));
  } else if (split) {
    ProfScope ps(c, CAT_LSTM, st);
    const int rc = lstm32s_launch(c->cfg.arch == 0, geom.nst, w.ndir, st, pre, w.w_hh[0], w.w_hh[1], hc, w.ndir * LSTM_H, (int)M, geom);
    if (rc != 0) return c->fail(DPTNAV_ERR_HIP, "lstm32s: %s", hipGetErrorString((hipError_t)rc));
  } else {
    // diagnostic stamps land behind the dump rows of hc (see make_plan)
    unsigned long long* stamps = reinterpret_cast<unsigned long long*>(ws + pl.stamps);
    ProfScope ps(c, CAT_LSTM, st);
    // DPTN inference stores ReLU(h) (ffn = ReLU -> Linear, dptn.py:31); training keeps raw h for the tape, DPRNN
    // feeds fc directly
    const int rc = lstm32_launch(c->opt_lstm_stamps && !pb.train, pb.train, c->cfg.arch == 0 && !pb.train, geom.nst, w.ndir,
                                 st, pre, w.w_hh[0], w.w_hh[1], hc, w.ndir * LSTM_H, (int)M, geom, stamps, pb.gates, pb.cst);
    if (rc != 0) return c->fail(DPTNAV_ERR_HIP, "lstm: %s", hipGetErrorString((hipError_t)rc));
  }
  if (run.lstm_record && hipEventRecord(run.lstm_record, st) != hipSuccess)
    return c->fail(DPTNAV_ERR_HIP, "lstm stagger record");
  // K6 (DPRNN): x_out = LayerNorm(h W_fc^T + b_fc) + x_in       (dprnn.py:41-45, 83-87)
  if (!dptn) {
    EpiBiasLNRes<GROUP> ep{x_out, w.ffn_b, x_in, w.ln2_w, w.ln2_b, M, N, BM};
    if (w.ndir == 2 && split) {
      ALoadDense al{hc, M, 2 * LSTM_H, BM};
      if (int rc = launch_gemm<2 * LSTM_H, 1, WR, WC, false, true>(c, run, CAT_FFN, "fc gemm (split)", w.ffn_w, ntiles, 1, al, ep)) return rc;
    } else if (w.ndir == 2 && pb.train && pb.zn2) {   // + zn / rstd on the tape for the LayerNorm backward
      ALoadDense al{hc, M, 2 * LSTM_H, BM};
      EpiBiasLNResSave<GROUP> eps{x_out, w.ffn_b, x_in, w.ln2_w, w.ln2_b, M, N, BM, pb.zn2, pb.rs2};
      if (int rc = launch_gemm<2 * LSTM_H, 1, WR, WC>(c, run, CAT_FFN, "fc gemm (tape)", w.ffn_w, ntiles, 1, al, eps)) return rc;
    } else if (w.ndir == 2) {
      int r = 0;
      if (N == 64 && c->opt_fcln) {     // 16-token tiles, several workgroups per CU (fcln.hip)
        FclnArgs fa{hc, w.ffn_w, w.ffn_b, w.ln2_w, w.ln2_b, x_in, x_out, nullptr, nullptr, M, 2 * LSTM_H, N, /*pre_res*/ false, /*act*/ 0, c->opt_fcln == 2 ? 3 : 2};
        r = try_fcln(c, st, fa, CAT_FFN, "fcln (fc)");
        if (r < 0) return -r;
      }
      if (r == 0) {
        ALoadDense al{hc, M, 2 * LSTM_H, BM};
        if (int rc = launch_gemm<2 * LSTM_H, 1, WR, WC>(c, run, CAT_FFN, "fc gemm", w.ffn_w, ntiles, 1, al, ep)) return rc;
      }
    } else if (pb.train && pb.zn2) {
      ALoadDense al{hc, M, LSTM_H, BM};
      EpiBiasLNResSave<GROUP> eps{x_out, w.ffn_b, x_in, w.ln2_w, w.ln2_b, M, N, BM, pb.zn2, pb.rs2};
      if (int rc = launch_gemm<LSTM_H, 1, WR, WC>(c, run, CAT_FFN, "fc gemm (tape)", w.ffn_w, ntiles, 1, al, eps)) return rc;
    } else {
      ALoadDense al{hc, M, LSTM_H, BM};
      if (int rc = launch_gemm<LSTM_H, 1, WR, WC>(c, run, CAT_FFN, "fc gemm", w.ffn_w, ntiles, 1, al, ep)) return rc;
    }
    return DPTNAV_OK;
  }
  // K6: x_out = LN2(relu(h) W_f^T + b_f + y1)                    (dptn.py:50-51)
  if (chain & CHAIN_SKIP_FFN) return DPTNAV_OK;   // done by the next path's attention block (its prologue)
  {
    EpiBiasResLN<GROUP> ep{x_out, w.ffn_b, y1, w.ln2_w, w.ln2_b, M, N, BM};
    // (training keeps the raw h on the tape and applies ffn[0] = ReLU while loading)
    if (w.ndir == 2 && split) {
      ALoadCols al{hc, M, 2 * LSTM_H, 0, BM};
      if (int rc = launch_gemm<2 * LSTM_H, 1, WR, WC, false, true>(c, run, CAT_FFN, "ffn gemm (split)", w.ffn_w, ntiles, 1, al, ep)) return rc;
    } else if (w.ndir == 2 && pb.train) {
      ALoadColsReLU al{hc, M, 2 * LSTM_H, 0, BM};
      bool done = false;
      {
        if (pb.zn2 && N == 128 && c->opt_fcln) {     // ReLU while loading, 16-token tiles, two workgroups per CU (fcln.hip)
          FclnArgs fa{hc, w.ffn_w, w.ffn_b, w.ln2_w, w.ln2_b, y1, x_out, pb.zn2, pb.rs2, M, 2 * LSTM_H, N, /*pre_res*/ true, /*act: ReLU*/ 1};
          const int r = try_fcln(c, st, fa, CAT_FFN, "fcln (ffn)");
          if (r < 0) return -r;
          done = r > 0;
        }
        if (!done && pb.zn2) {
          EpiBiasResLNSave<GROUP> eps{x_out, w.ffn_b, y1, w.ln2_w, w.ln2_b, M, N, BM, pb.zn2, pb.rs2};
          if (int rc = launch_gemm<2 * LSTM_H, 1, WR, WC>(c, run, CAT_FFN, "ffn gemm (tape)", w.ffn_w, ntiles, 1, al, eps)) return rc;
          done = true;
        }
      }
      if (!done)
        if (int rc = launch_gemm<2 * LSTM_H, 1, WR, WC>(c, run, CAT_FFN, "ffn gemm", w.ffn_w, ntiles, 1, al, ep)) return rc;
    } else if (w.ndir == 2) {
      int r = 0;
      if (c->opt_fcln) {       // (the paths whose FFN does not ride in the next attention block; hc = ReLU(h) already)
        FclnArgs fa{hc, w.ffn_w, w.ffn_b, w.ln2_w, w.ln2_b, y1, x_out, nullptr, nullptr, M, 2 * LSTM_H, N, /*pre_res*/ true, /*act*/ 0};
        r = try_fcln(c, st, fa, CAT_FFN, "fcln (ffn)");
        if (r < 0) return -r;
      }
      if (r == 0) {
        ALoadCols al{hc, M, 2 * LSTM_H, 0, BM};
        if (int rc = launch_gemm<2 * LSTM_H, 1, WR, WC>(c, run, CAT_FFN, "ffn gemm", w.ffn_w, ntiles, 1, al, ep)) return rc;
      }
    } else if (pb.train && pb.zn2) {
      ALoadColsReLU al{hc, M, LSTM_H, 0, BM};
      EpiBiasResLNSave<GROUP> eps{x_out, w.ffn_b, y1, w.ln2_w, w.ln2_b, M, N, BM, pb.zn2, pb.rs2};
      if (int rc = launch_gemm<LSTM_H, 1, WR, WC>(c, run, CAT_FFN, "ffn gemm (tape)", w.ffn_w, ntiles, 1, al, eps)) return rc;
    } else if (pb.train) {
      ALoadColsReLU al{hc, M, LSTM_H, 0, BM};
      if (int rc = launch_gemm<LSTM_H, 1, WR, WC>(c, run, CAT_FFN, "ffn gemm", w.ffn_w, ntiles, 1, al, ep)) return rc;
    } else {
      ALoadCols al{hc, M, LSTM_H, 0, BM};
      if (int rc = launch_gemm<LSTM_H, 1, WR, WC>(c, run, CAT_FFN, "ffn gemm", w.ffn_w, ntiles, 1, al, ep)) return rc;
    }
  }
  return DPTNAV_OK;
}

template <int N>
int run_head(dptnav_ctx* c, Run& run, const float* mix, const float* e1, const float* e2, int B, int64_t T, int Tv,
             float* E, float* X, float* vidbuf = nullptr) {
  float* ws = run.ws;
  const Plan& pl = run.pl;
  hipStream_t st = run.st;
  const dptnav_config& g = c->cfg;
  const float* vid = nullptr;
  if (!g.audio_only) {
    float* v = vidbuf ? vidbuf : ws + pl.vid;
    ProfScope ps(c, CAT_VIDEO, st);
    hipLaunchKernelGGL(video_linear_kernel, dim3(2 * B, g.hidden_video / 2), dim3(256), 0, st, e1, e2, c->w("visual_compression.weight"),
                       c->w("visual_compression.bias"), v, g.video_emb_size, Tv, g.hidden_video / 2);
    LAUNCH_CHECK(c, "video linear");
    vid = v;
  }
  constexpr int FPB = 256 / (N / 4);
  ProfScope ps(c, CAT_ENCODER, st);
  hipLaunchKernelGGL(encoder_fuse_kernel<N>, dim3((unsigned)((pl.L + FPB * ENC_PASSES - 1) / (FPB * ENC_PASSES)), B), dim3(256), 0, st, mix,
                     c->w("encoder.weight"), vid, g.audio_only ? nullptr : c->w("gate"),
                     g.audio_only ? nullptr : c->w("video_ln.weight"), g.audio_only ? nullptr : c->w("video_ln.bias"),
                     E, X, T, (int)pl.L, g.kernel_size_enc, c->stride, Tv, (int)pl.S, g.chunk_size, g.step_size);
  LAUNCH_CHECK(c, "encoder+fusion");
  return DPTNAV_OK;
}

template <int N>
int run_tail(dptnav_ctx* c, Run& run, const float* x, const float* E, int B, int64_t T, float* s1, float* s2,
             float* Zbuf = nullptr) {
  float* ws = run.ws;
  const Plan& pl = run.pl;
  hipStream_t st = run.st;
  constexpr int WR = N == 128 ? 1 : 2, WC = N == 128 ? 4 : 2, GROUP = N / 4;
  constexpr int BM = 32 * WR;
  const dptnav_config& g = c->cfg;
  float *Z = Zbuf ? Zbuf : ws + pl.qkv, *D = ws + pl.att;
  const int64_t M = pl.M;
  // T1: Z = PReLU(x) W_sep^T + b_sep                            (dptn_wav.py:26-29,47)
  int sep_done = 0;
  if (c->opt_fcln) {       // 16-token tiles, W_sep in registers, two or more workgroups per CU (fcln.hip, its plain form)
    FclnArgs fa{x, c->w("dprnn.speakers_separation.1.weight"), c->w("dprnn.speakers_separation.1.bias"), nullptr, nullptr, nullptr, Z, nullptr,
                nullptr, M, N, 2 * N, /*pre_res*/ false, /*act: PReLU*/ 2, 2, /*layernorm*/ false, c->w("dprnn.speakers_separation.0.weight")};
    sep_done = try_fcln(c, st, fa, CAT_SEP, "fcln (separation conv)");
    if (sep_done < 0) return -sep_done;
  }
  if (!sep_done) {
    ALoadDensePReLU al{x, c->w("dprnn.speakers_separation.0.weight"), M, N, 32};
    EpiBiasStore ep{Z, c->w("dprnn.speakers_separation.1.bias"), M, 2 * N, 32, 2 * N};
    if (int rc = launch_gemm<N, N / 64, 1, 4>(c, run, CAT_SEP, "separation gemm",
                                              c->w("dprnn.speakers_separation.1.weight"), (M + 31) / 32, 1, al, ep))
      return rc;
  }
  // T2: overlap-add gather -> post-processing conv -> + E -> decoder tap products
  const int ola = (int)((pl.S - 1) * g.step_size + g.chunk_size);
  const int left = (int)((pl.L - ola) / 2);
  if (c->opt_fold_tail) {
    // the three linear steps folded into one contraction of length 2N per frame (headtail.h); the training forward too:
    // its backward recomputes q from Z and E itself (run_tail_backward), nothing of the GEMM form is kept
    float* Wf = ws + pl.wfold;
    ProfScope ps(c, CAT_POST, st);
    hipLaunchKernelGGL(fold_decoder_kernel<N>, dim3(8), dim3(N), 0, st, c->w("dprnn.postprocessing.0.weight"),
                       c->w("dprnn.postprocessing.0.bias"), c->w("decoder.weight"), g.kernel_size_enc, Wf);
    const int64_t rows = (int64_t)2 * B * pl.L;
    hipLaunchKernelGGL(taps_fold_kernel<N>, dim3((unsigned)((rows + 31) / 32)), dim3(256), 0, st, Z, E, Wf, D, B, (int)pl.L,
                       (int)pl.S, g.chunk_size, g.step_size, left, ola);
    LAUNCH_CHECK(c, "folded post-processing / decoder taps");
  } else {
    const int64_t rows = (int64_t)2 * B * pl.L;
    ALoadOla al{Z, N, B, (int)pl.L, (int)pl.S, g.chunk_size, g.step_size, left, ola, BM};
    EpiSkipDecoderTaps<GROUP> ep{D, c->w("dprnn.postprocessing.0.bias"), E, c->w("decoder.weight"),
                                 (int64_t)B * pl.L, g.kernel_size_enc, BM};
    if (int rc = launch_gemm<N, 1, WR, WC>(c, run, CAT_POST, "postproc gemm", c->w("dprnn.postprocessing.0.weight"),
                                           (rows + BM - 1) / BM, 1, al, ep))
      return rc;
  }
  // T3: transposed-conv gather + zero padding
  {
    const int64_t ndec = (pl.L - 1) * c->stride + g.kernel_size_enc;
    const int pad_left = (int)((T - ndec) / 2);
    ProfScope ps(c, CAT_DECODER, st);
    hipLaunchKernelGGL(decoder_gather_kernel, dim3((unsigned)((T + 255) / 256), B, 2), dim3(256), 0, st, D, s1, s2, B,
                       T, (int)pl.L, g.kernel_size_enc, c->stride, pad_left);
    LAUNCH_CHECK(c, "decoder gather");
  }
  return DPTNAV_OK;
}

// =================================================================================================
// training step, path level (BASELINE config 4): forward with a tape, backward from the tape
// =================================================================================================
struct PathTape {  // offsets in floats inside one path's tape
  size_t qkv, att, y1, hc, gates, cst, astats, total;   // astats: softmax (max, 1/sum) per (token, head)
  size_t amask;   // dropout keep decisions of the attention forward: [seq][head][qb][kb][16] 64-bit lane masks (attention.h)
  size_t zn1, rs1, zn2, rs2;   // option ln_tape: LayerNorm 1 / 2 normalised rows [M][N] and 1/sigma [M]; 0 when off
};
struct BwdPlan {   // offsets in floats inside the backward workspace
  size_t queue, dz, dh, dg, dy1, datt, dqkv, slab, lnp, dxa, dxb, dq, du, de, dvi, dv, total;
  size_t wiht;                      // fragment-order copies of the weights dgrad_t.hip multiplies by (2 x W_ih, in_proj_weight), per path
  size_t dg2, dg3, slab2, queue2;   // more dP buffers / second slab region / ticket counters: the LSTM weight gradients on a side stream (option wgrad_side)
  int slab_wgs;
};
constexpr int BWD_LNP_WGS = 2048;       // upper bound of GEMM-engine workgroups writing LayerNorm partials

int make_path_tape(dptnav_ctx* c, int B, int S, PathTape* t) {
  const dptnav_config& g = c->cfg;
  const int64_t N = g.num_features, H = g.hidden_dim, K = g.chunk_size;
  const int64_t M = (int64_t)B * S * K;
  const int64_t nst = std::max(((int64_t)B * S + 31) / 32 * K, ((int64_t)B * K + 31) / 32 * S);
  size_t o = 0;
  auto take = [&](size_t n) { size_t at = o; o += align64(n); return at; };
  const bool dptn = g.arch == 0;       // DPRNN blocks have no attention half: hc, gates, cell states and the LayerNorm rows only
  t->qkv = take(dptn ? (size_t)M * 3 * N : 0);
  t->att = take(dptn ? (size_t)M * N : 0);
  t->y1 = take(dptn ? (size_t)M * N : 0);
  t->hc = take((size_t)(M + S * K) * 2 * H);
  t->gates = take((size_t)2 * nst * 512 * 32);
  t->cst = take((size_t)2 * nst * 128 * 32);
  t->astats = take(dptn ? (size_t)M * g.num_heads * 2 : 0);
  {   // 2 floats per word; the larger of the two paths' (sequences x blocks^2)
    const int64_t nk = (K + 31) / 32, ns = (S + 31) / 32;
    const int64_t words = std::max((int64_t)B * S * nk * nk, (int64_t)B * K * ns * ns) * g.num_heads * 16;
    t->amask = take(dptn ? (size_t)words * 2 : 0);
  }
  t->zn1 = t->rs1 = t->zn2 = t->rs2 = 0;
  if (c->opt_ln_tape) {
    t->zn1 = take(dptn ? (size_t)M * N : 64);      // (never 0: "zn1 != 0" is how the callers see that the LayerNorm tape is on)
    t->rs1 = take(dptn ? (size_t)M : 0);
    t->zn2 = take((size_t)M * N);
    t->rs2 = take((size_t)M);
  }
  t->total = o;
  return DPTNAV_OK;
}

int make_bwd_plan(dptnav_ctx* c, int B, int S, BwdPlan* p, int64_t L = 0, int Tv = 1) {
  const dptnav_config& g = c->cfg;
  const int64_t N = g.num_features, H = g.hidden_dim, K = g.chunk_size;
  const int64_t M = (int64_t)B * S * K, MD = M + (int64_t)S * K;
  if (L == 0) L = (int64_t)(S - 1) * g.step_size + K;
  size_t o = 0;
  auto take = [&](size_t n) { size_t at = o; o += align64(n); return at; };
  p->queue = take(QUEUE_SLOTS);
  p->dz = take((size_t)M * N);
  p->dh = take((size_t)MD * 2 * H);
  p->dg = take((size_t)MD * 2 * 4 * H);
  p->dy1 = take((size_t)M * N);
  p->datt = take((size_t)M * N);
  p->dqkv = take((size_t)M * 3 * N);
  p->slab = take((size_t)BWD_SLAB_WGS * 512 * 128);
  p->dg2 = take((size_t)MD * 2 * 4 * H);
  p->dg3 = take((size_t)MD * 2 * 4 * H);
  p->slab2 = take((size_t)BWD_SLAB_WGS * 512 * 128);
  p->queue2 = take(QUEUE_SLOTS);
  p->wiht = take(g.arch == 0 && N == 128 ? (size_t)2 * g.num_blocks * BWD_PACK_FLOATS : 0);      // pack_bwd_weights
  p->lnp = take((size_t)BWD_LNP_WGS * 8 * N);      // LayerNorm (2N) or decoder-tap (8N) partials per workgroup
  p->dxa = take((size_t)M * N);                     // gradient ping-pong between paths
  p->dxb = take((size_t)M * N);
  p->dq = take((size_t)2 * B * L * N);
  p->du = take((size_t)2 * B * L * N);
  p->de = take((size_t)B * L * N);
  p->dvi = take((size_t)B * L * N);
  p->dv = take((size_t)B * (Tv > 0 ? Tv : 1) * N);
  p->total = o;
  p->slab_wgs = BWD_SLAB_WGS;
  return DPTNAV_OK;
}

struct BwdRun {
  float* ws;
  BwdPlan pl;
  hipStream_t st;
  int slot;
  float* const* gptr;                // where this run WRITES the parameter gradients (slot order)
  int half = 0;
  hipEvent_t lstm_wait = nullptr;    // BPTT launches of the two halves of a split batch are chained like the forward's
  hipEvent_t lstm_record = nullptr;
  // Side stream (option wgrad_side, split batches): nothing needs dW before the step ends, so the LSTM weight-gradient
  // launches leave the half's chain and run whenever CUs are free -- in particular while this half waits for the other
  // half's BPTT (8.9 ms per step with 114 CUs idle in the kernel timeline).  dP rotates through three buffers: the BPTT
  // of path p-3 waits for the weight gradients of path p to have read theirs.
  hipStream_t side = nullptr;
  hipEvent_t ev_bptt = nullptr, ev_wg[3] = {nullptr, nullptr, nullptr};
  bool wg_pending[3] = {false, false, false};
  int dg_sel = 0, side_slot = 0;
  bool packed_all = false;           // ws + pl.wiht holds the fragment-order weight copies of ALL paths (dptnav_train_backward); else slot 0 = this path
  unsigned* take_queue_side(int n) {   // own counters: the main stream re-zeroes its region while side launches may be in flight
    if (side_slot + n > QUEUE_SLOTS) return nullptr;        // the caller fails the launch (as launch_gemm does for `slot`)
    unsigned* q = reinterpret_cast<unsigned*>(ws + pl.queue2) + side_slot;
    side_slot += n;
    return q;
  }
  unsigned* take_queue(int n) {          // nullptr when the zeroed region is used up (the caller fails the launch): the
    if (slot + n > QUEUE_SLOTS) return nullptr;   // backward recycles its counters per path while slot > QUEUE_SLOTS - 64, and a path takes
    unsigned* q = reinterpret_cast<unsigned*>(ws + pl.queue) + slot;   // at most 40 (N = 128: 18 GEMMs + 6 wgrad; N = 64 / DPRNN: <= 40)
    slot += n;
    return q;
  }
};

// dW[NN][KK] = sum_tokens Y^T X  ->  grad (overwrite);  bias_grad (optional): column sums of Y, fused into the same pass
template <int NN, int KK, class YL, class XL>
int launch_wgrad(dptnav_ctx* c, BwdRun& br, const char* what, int64_t ntiles, const YL& yl, const XL& xl, float* grad,
                 float* bias_grad = nullptr) {
  const size_t lds = WgradShape<NN, KK>::lds_bytes();
  const int grid = cap_grid(ntiles, br.pl.slab_wgs);
  float* slab = br.ws + br.pl.slab;
  float* colslab = slab + (size_t)grid * NN * KK;   // behind the weight slabs (BWD_SLAB_WGS x 512 x 128 floats in all)
  if (bias_grad) {
    auto kern = wgrad_kernel<NN, KK, YL, XL, true>;
    static PerDeviceOnce ready;
    if (!ready.done(c->device_id)) {
      if (int rc = set_lds(c, kern, lds, what)) return rc;
      ready.set(c->device_id);
    }
    unsigned* const queue = br.take_queue(1);
    if (!queue) return c->fail(DPTNAV_ERR_INVALID, "%s: ticket counters exhausted", what);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, br.st, (int)ntiles, c->opt_deterministic ? nullptr : queue, yl, xl, slab, colslab);
  } else {
    auto kern = wgrad_kernel<NN, KK, YL, XL, false>;
    static PerDeviceOnce ready;
    if (!ready.done(c->device_id)) {
      if (int rc = set_lds(c, kern, lds, what)) return rc;
      ready.set(c->device_id);
    }
    unsigned* const queue = br.take_queue(1);
    if (!queue) return c->fail(DPTNAV_ERR_INVALID, "%s: ticket counters exhausted", what);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, br.st, (int)ntiles, c->opt_deterministic ? nullptr : queue, yl, xl, slab, (float*)nullptr);
  }
  LAUNCH_CHECK(c, what);
  constexpr int64_t count = (int64_t)NN * KK;
  FragOuts outs{};
  outs.out[0] = grad;
  // the bias gradient's column sums are reduced by extra workgroups of the same launch
  hipLaunchKernelGGL((slab_reduce_frag_kernel<NN / 128, KK / 32>), dim3((unsigned)(count / 128) + (bias_grad ? (NN + 31) / 32 : 0)),
                     dim3(256), 0, br.st, slab, grid, count, outs, (int64_t)0, 1, colslab, bias_grad ? NN : 0, bias_grad);
  LAUNCH_CHECK(c, what);
  return DPTNAV_OK;
}

// the same for the shapes wgrad_kernel is not built for (NN not a multiple of 128): wgrad_generic_kernel, row-major partial
// tiles summed by slab_reduce_kernel; the bias gradient (column sums of Y) is a separate pass over Y (launch_colsum below)
template <int NN, int KK, class YL, class XL>
int launch_wgrad_generic(dptnav_ctx* c, BwdRun& br, const char* what, int64_t ntiles, const YL& yl, const XL& xl, float* grad) {
  const size_t lds = sizeof(float) * 32 * (size_t)((NN + 4) + (KK + 4));
  const int grid = cap_grid(ntiles, br.pl.slab_wgs);
  float* slab = br.ws + br.pl.slab;
  static_assert((size_t)NN * KK <= 512 * 128, "slab size");
  auto kern = wgrad_generic_kernel<NN, KK, YL, XL>;
  static PerDeviceOnce ready;
  if (!ready.done(c->device_id)) {
    if (int rc = set_lds(c, kern, lds, what)) return rc;
    ready.set(c->device_id);
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, br.st, (int)ntiles, yl, xl, slab);
  LAUNCH_CHECK(c, what);
  constexpr int64_t count = (int64_t)NN * KK;
  hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((count + 31) / 32)), dim3(256), 0, br.st, slab, grid, count, grad, 0);
  LAUNCH_CHECK(c, what);
  return DPTNAV_OK;
}

// sums of the partial tiles / column sums a WgradRider launch (launch_gemm) left in the slab region
template <int NN, int KK>
int reduce_rider(dptnav_ctx* c, BwdRun& br, const char* what, int grid, int col_after, float* grad, float* bias_grad) {
  constexpr int64_t count = (int64_t)NN * KK;
  float* slab = br.ws + br.pl.slab;
  float* colslab = slab + (size_t)col_after * count;   // where the launch put its column-sum rows
  FragOuts outs{};
  outs.out[0] = grad;
  hipLaunchKernelGGL((slab_reduce_frag_kernel<NN / 128, KK / 32>), dim3((unsigned)(count / 128) + (bias_grad ? (NN + 31) / 32 : 0)),
                     dim3(256), 0, br.st, slab, grid, count, outs, (int64_t)0, 1, colslab, bias_grad ? NN : 0, bias_grad);
  LAUNCH_CHECK(c, what);
  return DPTNAV_OK;
}

// NSL pairs of gradients, each pair sharing its Y operand, in one pass and ONE launch (wgrad2_kernel):
// gradA[s] = sum Y_s^T Xa_s, gradB[s] = sum Y_s^T Xb_s
template <int NN, int KK, int NSL, class YL, class XA, class XB>
int launch_wgrad2(dptnav_ctx* c, BwdRun& br, const char* what, int64_t ntiles, const Wgrad2Args<YL, XA, XB, NSL>& args,
                  float* const* gradA, float* const* gradB, hipStream_t wst, float* slab, unsigned* queue) {
  const size_t lds = sizeof(float) * (4 + 32 * (size_t)(WgradShape<NN, KK>::LDY + 2 * WgradShape<NN, KK>::LDX));
  static_assert(2 * NSL <= 8, "FragOuts");
  int grid = cap_grid(ntiles * NSL, br.pl.slab_wgs);
  grid -= grid % NSL;                          // the same number of workgroups for every slice
  if (grid < NSL) grid = NSL;
  // slab: [NSL][grid / NSL][2][NN][KK]: BWD_SLAB_WGS x 512 x 128 floats hold <256,128> exactly
  static_assert(2 * NN * KK <= 512 * 128, "slab size");
  auto kern = wgrad2_kernel<NN, KK, YL, XA, XB, NSL>;
  static PerDeviceOnce ready;
  if (!ready.done(c->device_id)) {
    if (int rc = set_lds(c, kern, lds, what)) return rc;
    ready.set(c->device_id);
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, wst, (int)ntiles, c->opt_deterministic ? nullptr : queue, args, slab);
  LAUNCH_CHECK(c, what);
  constexpr int64_t count = (int64_t)NN * KK;
  FragOuts outs{};
  for (int s = 0; s < NSL; ++s) {
    outs.out[2 * s] = gradA[s];
    outs.out[2 * s + 1] = gradB[s];
  }
  // sum y = (slice, a / b): the slabs of a slice are 2 * count apart, a slice's region is (grid / NSL) of them
  hipLaunchKernelGGL((slab_reduce_frag_kernel<NN / 128, KK / 32>), dim3((unsigned)(count / 128), 2 * NSL), dim3(256), 0, wst, slab,
                     grid / NSL, 2 * count, outs, (int64_t)(grid / NSL) * 2 * count, 2, (const float*)nullptr, 0, (float*)nullptr);
  LAUNCH_CHECK(c, what);
  return DPTNAV_OK;
}

// grad[C] = column sums of Y[:, col0:col0+C]   (second destination optional: b_ih and b_hh share their gradient)
template <int C>
int launch_colsum(dptnav_ctx* c, BwdRun& br, const char* what, const float* Y, int64_t M, int ld, int col0, float* grad,
                  float* grad2 = nullptr) {
  const int grid = br.pl.slab_wgs;
  float* slab = br.ws + br.pl.slab;
  hipLaunchKernelGGL(colsum_kernel<C>, dim3(grid), dim3(256), 0, br.st, Y, M, ld, col0, slab);
  LAUNCH_CHECK(c, what);
  hipLaunchKernelGGL(slab_reduce_kernel, dim3((C + 31) / 32), dim3(256), 0, br.st, slab, grid, (int64_t)C, grad, 0);
  if (grad2) hipLaunchKernelGGL(slab_reduce_kernel, dim3((C + 31) / 32), dim3(256), 0, br.st, slab, grid, (int64_t)C, grad2, 0);
  LAUNCH_CHECK(c, what);
  return DPTNAV_OK;
}

// One DPRNN block half backward (IntraChunkRNN / InterChunkRNN, dprnn.py:24-47,65-89):
//   forward   h = biLSTM(x);  v = h W_fc^T + b_fc;  out = LayerNorm(v) + x
//   backward  dz = LayerNorm'(d_out)  [tape: zn, rstd];  dW_fc = dz^T h, db_fc = colsum dz;  dh = dz W_fc;  BPTT -> dP;
//             dW_ih = dP^T x, dW_hh = dP^T h_{t-1}, db = colsum dP;  d_in = d_out (residual) + dP_f W_ih_f + dP_b W_ih_b
// The kernels are the DPTN chain's (ln_backward_kernel, BPTT, weight gradients, data-gradient GEMMs); no attention half.
template <int N>
int run_path_backward_dprnn(dptnav_ctx* c, BwdRun& br, int block, int path, const float* x_in, const float* d_out, float* d_in,
                            int B, int S, float* tape, const PathTape& tp) {
  constexpr int WRn = N == 128 ? 1 : 2, WCn = N == 128 ? 4 : 2, BMn = 32 * WRn;
  const dptnav_config& g = c->cfg;
  const PathWeights& w = c->pw[2 * block + path];
  const std::string pre = "dprnn.model." + std::to_string(block) + (path == 0 ? ".intra_chunk_block." : ".inter_chunk_block.");
  auto G = [&](const char* leaf) { return br.gptr[c->slot(pre + leaf)]; };
  const int nd = w.ndir;                 // 2, or 1 for the inter-chunk path of a bidir = False model (dprnn.py:56-63)
  if (!tp.zn2) return c->fail(DPTNAV_ERR_INVALID, "training step of DPRNN blocks needs option ln_tape = 1");
  const int K = g.chunk_size;
  const int64_t M = (int64_t)B * S * K;
  const SeqGeom geom = make_geom(path, B, S, K);
  hipStream_t st = br.st;
  float *hc = tape + tp.hc, *gates = tape + tp.gates, *cst = tape + tp.cst;
  float *DZ = br.ws + br.pl.dz, *DHb = br.ws + br.pl.dh, *DG = br.ws + br.pl.dg, *LNP = br.ws + br.pl.lnp;
  const int64_t ntiles = (M + 31) / 32, ntiles_n = (M + BMn - 1) / BMn;
  Run run;
  run.ws = br.ws;
  run.pl = Plan{};
  run.pl.queue = br.pl.queue;
  run.st = st;
  run.slot = br.slot;
  // 1. LayerNorm backward from the tape
  {
    const int64_t npass = (M + (256 / (N / 4)) - 1) / (256 / (N / 4));
    const int lgrid = (int)std::min<int64_t>(std::min<int64_t>(BWD_LNP_WGS, 4 * (int64_t)c->num_cus), (npass + 3) / 4);
    ProfScope ps(c, CAT_FFN, st);
    hipLaunchKernelGGL(ln_backward_kernel<N>, dim3(lgrid), dim3(256), 0, st, d_out, tape + tp.zn2, tape + tp.rs2, w.ln2_w, DZ, LNP, M);
    hipLaunchKernelGGL(slab_reduce_to2_kernel, dim3((2 * N + 31) / 32), dim3(256), 0, st, LNP, lgrid, (int64_t)2 * N, G("norm1d.weight"),
                       G("norm1d.bias"), N);
    LAUNCH_CHECK(c, "norm1d backward");
  }
  // 2. fc gradients, d h = dz W_fc
  if (nd == 1) {
    {
      ALoadCols yl{DZ, M, N, 0, 32};
      ALoadCols xl{hc, M, LSTM_H, 0, 32};
      if constexpr (N == 128) {
        if (int rc = launch_wgrad<N, LSTM_H>(c, br, "d fc weight + bias", ntiles, yl, xl, G("fc.weight"), G("fc.bias"))) return rc;
      } else {
        if (int rc = launch_wgrad_generic<N, LSTM_H>(c, br, "d fc weight", ntiles, yl, xl, G("fc.weight"))) return rc;
        if (int rc = launch_colsum<N>(c, br, "d fc bias", DZ, M, N, 0, G("fc.bias"))) return rc;
      }
    }
    run.slot = br.slot;
    {
      ALoadDense al{DZ, M, N, 32};
      EpiAddMaskStoreT<false, false> ep{DHb, nullptr, nullptr, M, LSTM_H, 32, LSTM_H};
      if (int rc = launch_gemm<N, 1, 1, 4, true>(c, run, CAT_FFN, "d h", w.ffn_w, ntiles, 1, al, ep, nullptr, LSTM_H)) return rc;
    }
    br.slot = run.slot;
  } else {
    {
      ALoadCols yl{DZ, M, N, 0, 32};
      ALoadCols xl{hc, M, 2 * LSTM_H, 0, 32};
      if constexpr (N == 128) {
        if (int rc = launch_wgrad<N, 2 * LSTM_H>(c, br, "d fc weight + bias", ntiles, yl, xl, G("fc.weight"), G("fc.bias"))) return rc;
      } else {
        if (int rc = launch_wgrad_generic<N, 2 * LSTM_H>(c, br, "d fc weight", ntiles, yl, xl, G("fc.weight"))) return rc;
        if (int rc = launch_colsum<N>(c, br, "d fc bias", DZ, M, N, 0, G("fc.bias"))) return rc;
      }
    }
    run.slot = br.slot;
    {
      ALoadDense al{DZ, M, N, 32};
      EpiAddMaskStoreT<false, false> ep{DHb, nullptr, nullptr, M, 2 * LSTM_H, 32, 2 * LSTM_H};
      if (int rc = launch_gemm<N, 2, 1, 4, true>(c, run, CAT_FFN, "d h", w.ffn_w, ntiles, 1, al, ep, nullptr, 2 * LSTM_H)) return rc;
    }
    br.slot = run.slot;
  }
  // 3. LSTM backward through time (tile height as in the forward that wrote the tape)
  const bool use16 = lstm_use16(c, geom, nd, M);
  const int ntl = use16 ? (geom.nseq + 15) / 16 : geom.nst;
  if (br.lstm_wait && hipStreamWaitEvent(st, br.lstm_wait, 0) != hipSuccess) return c->fail(DPTNAV_ERR_HIP, "bptt stagger wait");
  {
    ProfScope ps(c, CAT_LSTM, st);
    const int rc = use16 ? lstm_bptt16_launch(ntl, nd, st, gates, cst, w.w_hh[0], w.w_hh[1], DHb, nd * LSTM_H, DG, nd * 512, (int)M, geom, LNP)
                         : lstm_bptt_launch(geom.nst, nd, st, gates, cst, w.w_hh[0], w.w_hh[1], DHb, nd * LSTM_H, DG, nd * 512, (int)M, geom,
                                            LNP);
    if (rc != 0) return c->fail(DPTNAV_ERR_HIP, "lstm bptt: %s", hipGetErrorString((hipError_t)rc));
  }
  if (br.lstm_record && hipEventRecord(br.lstm_record, st) != hipSuccess) return c->fail(DPTNAV_ERR_HIP, "bptt stagger record");
  // 4. LSTM parameter gradients (the LSTM's input is the block input x)
  for (int d = 0; d < nd; ++d) {
    const char* sfx = d ? "_reverse" : "";
    const std::string wih = std::string("rnn.weight_ih_l0") + sfx, whh = std::string("rnn.weight_hh_l0") + sfx,
                      bih = std::string("rnn.bias_ih_l0") + sfx, bhh = std::string("rnn.bias_hh_l0") + sfx;
    hipLaunchKernelGGL(slab_reduce_to2_kernel, dim3(512 / 32), dim3(256), 0, st, LNP + (size_t)d * ntl * 512, ntl, (int64_t)512,
                       G(bih.c_str()), G(bhh.c_str()), -1);
    LAUNCH_CHECK(c, "d lstm bias");
    const ALoadDense xl{x_in, M, N, 32};
    const ALoadSeqShift hl = make_seq_shift(hc, M, nd * LSTM_H, d * LSTM_H, 32, d ? -1 : 1, geom);
    for (int half = 0; half < 2; ++half) {
      const ALoadCols yl{DG, M, nd * 512, d * 512 + half * 256, 32};
      if (int rc = launch_wgrad<256, N>(c, br, "d w_ih", ntiles, yl, xl, G(wih.c_str()) + half * 256 * N)) return rc;
      if (int rc = launch_wgrad<256, LSTM_H>(c, br, "d w_hh", ntiles, yl, hl, G(whh.c_str()) + half * 256 * LSTM_H)) return rc;
    }
  }
  // 5. d_in = d_out (residual) + dP_f W_ih_f + dP_b W_ih_b
  run.slot = br.slot;
  for (int d = 0; d < nd; ++d) {
    if constexpr (N == 128) {
      ALoadCols al{DG, M, nd * 512, d * 512, 32};
      EpiAddMaskStoreT<true, false> ep{d_in, d == 0 ? d_out : d_in, nullptr, M, N, 32, N};
      if (int rc = launch_gemm<512, 1, 1, 4, true>(c, run, CAT_LSTM_PRE, "d x", w.w_ih[d], ntiles, 1, al, ep, nullptr, N)) return rc;
    } else {
      for (int half = 0; half < 2; ++half) {
        ALoadCols al{DG, M, nd * 512, d * 512 + half * 256, BMn};
        EpiAddMaskStoreT<true, false> ep{d_in, d == 0 && half == 0 ? d_out : d_in, nullptr, M, N, BMn, N};
        if (int rc = launch_gemm<256, 1, WRn, WCn, true>(c, run, CAT_LSTM_PRE, "d x", w.w_ih[d] + (size_t)half * 256 * N, ntiles_n, 1, al,
                                                         ep, nullptr, N))
          return rc;
      }
    }
  }
  br.slot = run.slot;
  return DPTNAV_OK;
}

template <int N>
int run_path_backward(dptnav_ctx* c, BwdRun& br, int block, int path, const float* x_in, const float* d_out,
                      float* d_in, int B, int S, float* tape, const PathTape& tp) {
  constexpr int GROUP = N / 4, DH = N / 4;
  // N = 128 (BASELINE config 4): the tuned kernels (weight-gradient riders, wgrad2, fragment-order partial tiles);  N = 64
  // (DPTNWavEncDec): the same chain with the generic weight-gradient kernel where a tuned shape does not exist and
  // 64-row x 64-column tiles for the data-gradient GEMMs whose output is N wide
  constexpr int WRn = N == 128 ? 1 : 2, WCn = N == 128 ? 4 : 2, BMn = 32 * WRn;
  if (c->cfg.arch == 1) return run_path_backward_dprnn<N>(c, br, block, path, x_in, d_out, d_in, B, S, tape, tp);
  const dptnav_config& g = c->cfg;
  const PathWeights& w = c->pw[2 * block + path];
  const std::string pre = "dprnn.model." + std::to_string(block) + (path == 0 ? ".intra_chunk_block." : ".inter_chunk_block.");
  auto G = [&](const char* leaf) { return br.gptr[c->slot(pre + leaf)]; };
  const int nd = w.ndir;                 // 2, or 1 for the inter-chunk path of a bidir = False model (dptn.py:60)
  const int K = g.chunk_size;
  const int64_t M = (int64_t)B * S * K;
  const SeqGeom geom = make_geom(path, B, S, K);
  hipStream_t st = br.st;
  float *qkv = tape + tp.qkv, *att = tape + tp.att, *y1 = tape + tp.y1, *hc = tape + tp.hc, *gates = tape + tp.gates,
        *cst = tape + tp.cst;
  float *DZ = br.ws + br.pl.dz, *DHb = br.ws + br.pl.dh, *DG = br.ws + br.pl.dg, *DY1 = br.ws + br.pl.dy1,
        *DATT = br.ws + br.pl.datt, *DQKV = br.ws + br.pl.dqkv, *LNP = br.ws + br.pl.lnp;
  const int64_t ntiles = (M + 31) / 32, ntiles_n = (M + BMn - 1) / BMn;
  // fragment-order weight copies of this path (pack_bwd_weights): made for all paths by dptnav_train_backward, or here into slot 0
  float* const wpk = br.ws + br.pl.wiht + (br.packed_all ? (size_t)(2 * block + path) * BWD_PACK_FLOATS : 0);
  if constexpr (N == 128) {
    if (!br.packed_all)
      if (int rc = pack_bwd_weights(c, st, 2 * block + path, 1, wpk)) return rc;
  }
  Run run;   // the GEMM engine takes its ticket counters from a Run: alias it onto the backward workspace
  run.ws = br.ws;
  run.pl = Plan{};
  run.pl.queue = br.pl.queue;
  run.st = st;
  run.slot = br.slot;
  int grid = 0;

  // 1. push d_out through LayerNorm 2: from the tape's zn / rstd (option ln_tape), or by recomputing
  //    z2 = relu(h) W_f^T + b_f + y1 in a GEMM whose epilogue applies the derivative
  auto ln_from_tape = [&](const float* dout, const float* zn, const float* rs, const float* gamma, const char* gw, const char* gb) {
    const int64_t npass = (M + 7) / 8;
    const int lgrid = (int)std::min<int64_t>(std::min<int64_t>(BWD_LNP_WGS, 4 * (int64_t)c->num_cus), (npass + 3) / 4);
    ProfScope ps(c, CAT_FFN, st);
    hipLaunchKernelGGL(ln_backward_kernel<N>, dim3(lgrid), dim3(256), 0, st, dout, zn, rs, gamma, DZ, LNP, M);
    hipLaunchKernelGGL(slab_reduce_to2_kernel, dim3((2 * N + 31) / 32), dim3(256), 0, st, LNP, lgrid, (int64_t)2 * N, G(gw), G(gb), N);
    return hipGetLastError() == hipSuccess;
  };
  if (tp.zn2) {
    if (!ln_from_tape(d_out, tape + tp.zn2, tape + tp.rs2, w.ln2_w, "ln2.weight", "ln2.bias")) return c->fail(DPTNAV_ERR_HIP, "ln2 backward");
  } else if constexpr (N != 128) {
    return c->fail(DPTNAV_ERR_INVALID, "training with num_features = %d needs option ln_tape = 1", N);
  } else {
    ALoadColsReLU al{hc, M, 2 * LSTM_H, 0, 32};
    EpiLNBackward<GROUP, 0> ep{DZ, w.ffn_b, y1, w.ln2_w, d_out, LNP, M, N, 32};
    if (int rc = launch_gemm<2 * LSTM_H, 1, 1, 4>(c, run, CAT_FFN, "ffn recompute + ln2 bwd", w.ffn_w, ntiles, 1, al, ep,
                                                 nullptr, 2 * LSTM_H, &grid))
      return rc;
    hipLaunchKernelGGL(slab_reduce_to2_kernel, dim3((2 * N + 31) / 32), dim3(256), 0, st, LNP, grid, (int64_t)2 * N, G("ln2.weight"),
                       G("ln2.bias"), N);
    LAUNCH_CHECK(c, "ln2 grads");
  }
  br.slot = run.slot;
  // 2 + 3. ffn parameter gradients and d h = (dz2 W_f) masked by the ReLU.  Option wgrad_ride (default): ONE launch -- the
  //        data-gradient GEMM stages the dz2 tile anyway and forms dW_f / db_f on the side (WgradRider, gemm_ws.h)
  if (nd == 1) {   // one direction: W_f is [N][128]; the plain schedule (no rider)
    {
      ALoadCols yl{DZ, M, N, 0, 32};
      ALoadColsReLU xl{hc, M, LSTM_H, 0, 32};
      if constexpr (N == 128) {
        if (int rc = launch_wgrad<N, LSTM_H>(c, br, "d ffn weight + bias", ntiles, yl, xl, G("ffn.1.weight"), G("ffn.1.bias"))) return rc;
      } else {
        if (int rc = launch_wgrad_generic<N, LSTM_H>(c, br, "d ffn weight", ntiles, yl, xl, G("ffn.1.weight"))) return rc;
        if (int rc = launch_colsum<N>(c, br, "d ffn bias", DZ, M, N, 0, G("ffn.1.bias"))) return rc;
      }
    }
    run.slot = br.slot;
    {
      ALoadDense al{DZ, M, N, 32};
      EpiAddMaskStoreT<false, true> ep{DHb, nullptr, hc, M, LSTM_H, 32, LSTM_H};
      if (int rc = launch_gemm<N, 1, 1, 4, true>(c, run, CAT_FFN, "d h", w.ffn_w, ntiles, 1, al, ep, nullptr, LSTM_H)) return rc;
    }
    br.slot = run.slot;
  } else {
    bool rode = false;
    if constexpr (N == 128) {
      if (c->opt_wgrad_ride) {
        rode = true;
        run.slot = br.slot;
        ALoadDense al{DZ, M, N, 32};
        EpiAddMaskStoreT<false, true> ep{DHb, nullptr, hc, M, 2 * LSTM_H, 32, 2 * LSTM_H};
        float* slab = br.ws + br.pl.slab;
        int rgrid = 0;
        const int t = try_dgrad_r(c, run, CAT_FFN, "d h + d ffn weight", DZ, wpk + BWD_PACK_FFNW, 2 * LSTM_H, true, hc, DHb, M, slab,
                                  slab + (size_t)BWD_SLAB_WGS * N * 2 * LSTM_H, BWD_SLAB_WGS, &rgrid);
        if (t < 0) return -t;
        // (grid known only after the occupancy query inside: the column-sum rows go behind BWD_SLAB_WGS partial tiles)
        WgradRider<2 * LSTM_H, ALoadColsReLU, true> rd{ALoadColsReLU{hc, M, 2 * LSTM_H, 0, 32}, slab,
                                                       slab + (size_t)BWD_SLAB_WGS * N * 2 * LSTM_H, M};
        if (!t)
          if (int rc = launch_gemm<N, 2, 1, 4, true, false>(c, run, CAT_FFN, "d h + d ffn weight", w.ffn_w, ntiles, 1, al, ep, nullptr,
                                                            2 * LSTM_H, &rgrid, rd))
            return rc;
        br.slot = run.slot;
        if (int rc = reduce_rider<N, 2 * LSTM_H>(c, br, "d ffn weight + bias", rgrid, BWD_SLAB_WGS, G("ffn.1.weight"), G("ffn.1.bias")))
          return rc;
      }
    }
    if (!rode) {
      {
        ALoadCols yl{DZ, M, N, 0, 32};
        ALoadColsReLU xl{hc, M, 2 * LSTM_H, 0, 32};
        if constexpr (N == 128) {
          if (int rc = launch_wgrad<N, 2 * LSTM_H>(c, br, "d ffn weight + bias", ntiles, yl, xl, G("ffn.1.weight"), G("ffn.1.bias")))
            return rc;
        } else {
          if (int rc = launch_wgrad_generic<N, 2 * LSTM_H>(c, br, "d ffn weight", ntiles, yl, xl, G("ffn.1.weight"))) return rc;
          if (int rc = launch_colsum<N>(c, br, "d ffn bias", DZ, M, N, 0, G("ffn.1.bias"))) return rc;
        }
      }
      run.slot = br.slot;
      {
        ALoadDense al{DZ, M, N, 32};
        EpiAddMaskStoreT<false, true> ep{DHb, nullptr, hc, M, 2 * LSTM_H, 32, 2 * LSTM_H};
        if (int rc = launch_gemm<N, 2, 1, 4, true>(c, run, CAT_FFN, "d h", w.ffn_w, ntiles, 1, al, ep, nullptr, 2 * LSTM_H))
          return rc;
      }
      br.slot = run.slot;
    }
  }
  // 4. LSTM backward through time (tile height as in the forward that wrote the tape)
  const bool use16 = lstm_use16(c, geom, nd, M);
  const int ntl = use16 ? (geom.nseq + 15) / 16 : geom.nst;   // workgroups per direction = partial bias rows
  if (br.lstm_wait && hipStreamWaitEvent(st, br.lstm_wait, 0) != hipSuccess) return c->fail(DPTNAV_ERR_HIP, "bptt stagger wait");
  const bool wg2 = c->opt_wgrad2 && N == LSTM_H && nd == 2;   // wgrad2 pairs two X operands of equal width, four slices
  const bool side = br.side != nullptr && wg2;
  if (side) {   // this path's dP goes to the buffer the weight gradients of two paths ago have (or will have) read
    if (br.dg_sel) DG = br.ws + (br.dg_sel == 1 ? br.pl.dg2 : br.pl.dg3);
    if (br.wg_pending[br.dg_sel] && hipStreamWaitEvent(st, br.ev_wg[br.dg_sel], 0) != hipSuccess)
      return c->fail(DPTNAV_ERR_HIP, "side stream wait");
  } else if (br.side != nullptr && br.wg_pending[0]) {
    // A path whose weight gradients stay in this stream (one LSTM direction: bidir = False) writes its dP to buffer 0 --
    // which a side-stream launch of an EARLIER path may still be reading.  Found by tests/test_gpu_memsafety.py (round 4):
    // block 1's intra-chunk LSTM weight gradients of a bidir = False model came out wrong whenever the next (inter-chunk)
    // BPTT overtook them.
    if (hipStreamWaitEvent(st, br.ev_wg[0], 0) != hipSuccess) return c->fail(DPTNAV_ERR_HIP, "side stream wait");
    br.wg_pending[0] = false;
  }
  {
    ProfScope ps(c, CAT_LSTM, st);
    const int rc = use16 ? lstm_bptt16_launch(ntl, nd, st, gates, cst, w.w_hh[0], w.w_hh[1], DHb, nd * LSTM_H, DG, nd * 512, (int)M,
                                              geom, LNP)
                         : lstm_bptt_launch(geom.nst, nd, st, gates, cst, w.w_hh[0], w.w_hh[1], DHb, nd * LSTM_H, DG, nd * 512, (int)M,
                                            geom, LNP);
    if (rc != 0) return c->fail(DPTNAV_ERR_HIP, "lstm bptt: %s", hipGetErrorString((hipError_t)rc));
  }
  if (br.lstm_record && hipEventRecord(br.lstm_record, st) != hipSuccess) return c->fail(DPTNAV_ERR_HIP, "bptt stagger record");
  // 5. LSTM parameter gradients
  Wgrad2Args<ALoadCols, ALoadDense, ALoadSeqShift, 4> wa{};
  float *gA[4], *gB[4];
  for (int d = 0; d < nd; ++d) {
    const char* sfx = d ? "_reverse" : "";
    const std::string wih = std::string("rnn.weight_ih_l0") + sfx, whh = std::string("rnn.weight_hh_l0") + sfx,
                      bih = std::string("rnn.bias_ih_l0") + sfx, bhh = std::string("rnn.bias_hh_l0") + sfx;
    // bias gradients: per-workgroup partial rows written by the BPTT kernel
    hipLaunchKernelGGL(slab_reduce_to2_kernel, dim3(512 / 32), dim3(256), 0, st, LNP + (size_t)d * ntl * 512, ntl,
                       (int64_t)512, G(bih.c_str()), G(bhh.c_str()), -1);
    LAUNCH_CHECK(c, "d lstm bias");
    const ALoadDense xl{y1, M, N, 32};
    const ALoadSeqShift hl = make_seq_shift(hc, M, nd * LSTM_H, d * LSTM_H, 32, d ? -1 : 1, geom);
    for (int half = 0; half < 2; ++half) {   // 512 gate rows as 2 x 256: 16 accumulator tiles per wave would spill
      const ALoadCols yl{DG, M, nd * 512, d * 512 + half * 256, 32};
      if (wg2) {   // W_ih and W_hh gradients in ONE pass over dP; the four (direction, half) slices in one launch
        if constexpr (N == LSTM_H) {
          const int s4 = 2 * d + half;
          wa.yl[s4] = yl;
          wa.xa[s4] = xl;
          wa.xb[s4] = hl;
          gA[s4] = G(wih.c_str()) + half * 256 * N;
          gB[s4] = G(whh.c_str()) + half * 256 * LSTM_H;
        }
      } else {
        if (int rc = launch_wgrad<256, N>(c, br, "d w_ih", ntiles, yl, xl, G(wih.c_str()) + half * 256 * N)) return rc;
        if (int rc = launch_wgrad<256, LSTM_H>(c, br, "d w_hh", ntiles, yl, hl, G(whh.c_str()) + half * 256 * LSTM_H)) return rc;
      }
    }
  }
  if constexpr (N == LSTM_H) {
    if (wg2) {
      if (side) {
        if (hipEventRecord(br.ev_bptt, st) != hipSuccess || hipStreamWaitEvent(br.side, br.ev_bptt, 0) != hipSuccess)
          return c->fail(DPTNAV_ERR_HIP, "side stream fork");
        unsigned* side_q = br.take_queue_side(4);
        if (!side_q) return c->fail(DPTNAV_ERR_INVALID, "d w_ih + d w_hh: side-stream ticket counters exhausted");
        if (int rc = launch_wgrad2<256, N, 4>(c, br, "d w_ih + d w_hh", ntiles, wa, gA, gB, br.side, br.ws + br.pl.slab2, side_q))
          return rc;
        if (hipEventRecord(br.ev_wg[br.dg_sel], br.side) != hipSuccess) return c->fail(DPTNAV_ERR_HIP, "side stream record");
        br.wg_pending[br.dg_sel] = true;
        br.dg_sel = (br.dg_sel + 1) % 3;
      } else {
        unsigned* q4 = br.take_queue(4);
        if (!q4) return c->fail(DPTNAV_ERR_INVALID, "d w_ih + d w_hh: ticket counters exhausted");
        if (int rc = launch_wgrad2<256, N, 4>(c, br, "d w_ih + d w_hh", ntiles, wa, gA, gB, st, br.ws + br.pl.slab, q4))
          return rc;
      }
    }
  }
  // 6. d y1 = dz2 (residual) + dG_f W_ih_f + dG_b W_ih_b
  run.slot = br.slot;
  for (int d = 0; d < nd; ++d) {
    if constexpr (N == 128) {
      const int t = try_dgrad_t(c, run, CAT_LSTM_PRE, "d y1", DG + d * 512, nd * 512, 512, wpk + BWD_PACK_WIH + (size_t)d * 512 * 128,
                                d == 0 ? DZ : DY1, DY1, M);
      if (t < 0) return -t;
      if (t) continue;
      ALoadCols al{DG, M, nd * 512, d * 512, 32};
      EpiAddMaskStoreT<true, false> ep{DY1, d == 0 ? DZ : DY1, nullptr, M, N, 32, N};
      if (int rc = launch_gemm<512, 1, 1, 4, true>(c, run, CAT_LSTM_PRE, "d y1", w.w_ih[d], ntiles, 1, al, ep, nullptr, N))
        return rc;
    } else {
      // 64-row tiles: a 64 x 512 A tile (double buffered) does not fit the LDS, so the gate columns go in two halves
      for (int half = 0; half < 2; ++half) {
        ALoadCols al{DG, M, nd * 512, d * 512 + half * 256, BMn};
        EpiAddMaskStoreT<true, false> ep{DY1, d == 0 && half == 0 ? DZ : DY1, nullptr, M, N, BMn, N};
        if (int rc = launch_gemm<256, 1, WRn, WCn, true>(c, run, CAT_LSTM_PRE, "d y1", w.w_ih[d] + (size_t)half * 256 * N, ntiles_n, 1,
                                                         al, ep, nullptr, N))
          return rc;
      }
    }
  }
  // 7. push d y1 through LayerNorm 1 (tape, or recompute z1 = att W_o^T + b_o + x)
  if (tp.zn1) {
    if (!ln_from_tape(DY1, tape + tp.zn1, tape + tp.rs1, w.ln1_w, "ln1.weight", "ln1.bias")) return c->fail(DPTNAV_ERR_HIP, "ln1 backward");
  } else if constexpr (N != 128) {
    return c->fail(DPTNAV_ERR_INVALID, "training with num_features = %d needs option ln_tape = 1", N);
  } else {
    ALoadDense al{att, M, N, 32};
    EpiLNBackward<GROUP, 0> ep{DZ, w.out_b, x_in, w.ln1_w, DY1, LNP, M, N, 32};
    if (int rc = launch_gemm<N, 1, 1, 4>(c, run, CAT_OUTPROJ, "out-proj recompute + ln1 bwd", w.out_w, ntiles, 1, al, ep,
                                        nullptr, N, &grid))
      return rc;
    hipLaunchKernelGGL(slab_reduce_to2_kernel, dim3((2 * N + 31) / 32), dim3(256), 0, st, LNP, grid, (int64_t)2 * N, G("ln1.weight"),
                       G("ln1.bias"), N);
    LAUNCH_CHECK(c, "ln1 grads");
  }
  br.slot = run.slot;
  // 8. out-projection gradients and d att (one launch with option wgrad_ride, as in 2 + 3)
  bool rode8 = false;
  if constexpr (N == 128) {
    if (c->opt_wgrad_ride) {
      rode8 = true;
      run.slot = br.slot;
      ALoadDense al{DZ, M, N, 32};
      EpiAddMaskStoreT<false, false> ep{DATT, nullptr, nullptr, M, N, 32, N};
      float* slab = br.ws + br.pl.slab;
      int rgrid = 0;
      const int t = try_dgrad_r(c, run, CAT_OUTPROJ, "d att + d out weight", DZ, wpk + BWD_PACK_OUTW, N, false, att, DATT, M, slab,
                                slab + (size_t)BWD_SLAB_WGS * N * N, BWD_SLAB_WGS, &rgrid);
      if (t < 0) return -t;
      WgradRider<N, ALoadDense, true> rd{ALoadDense{att, M, N, 32}, slab, slab + (size_t)BWD_SLAB_WGS * N * N, M};
      if (!t)
        if (int rc = launch_gemm<N, 1, 1, 4, true, false>(c, run, CAT_OUTPROJ, "d att + d out weight", w.out_w, ntiles, 1, al, ep,
                                                          nullptr, N, &rgrid, rd))
          return rc;
      br.slot = run.slot;
      if (int rc = reduce_rider<N, N>(c, br, "d out weight + bias", rgrid, BWD_SLAB_WGS, G("mha.out_proj.weight"), G("mha.out_proj.bias")))
        return rc;
    }
  }
  if (!rode8) {
    {
      ALoadCols yl{DZ, M, N, 0, 32};
      ALoadDense xl{att, M, N, 32};
      if constexpr (N == 128) {
        if (int rc = launch_wgrad<N, N>(c, br, "d out weight + bias", ntiles, yl, xl, G("mha.out_proj.weight"),
                                        G("mha.out_proj.bias")))
          return rc;
      } else {
        if (int rc = launch_wgrad_generic<N, N>(c, br, "d out weight", ntiles, yl, xl, G("mha.out_proj.weight"))) return rc;
        if (int rc = launch_colsum<N>(c, br, "d out bias", DZ, M, N, 0, G("mha.out_proj.bias"))) return rc;
      }
    }
    run.slot = br.slot;
    {
      ALoadDense al{DZ, M, N, BMn};
      EpiAddMaskStoreT<false, false> ep{DATT, nullptr, nullptr, M, N, BMn, N};
      if (int rc = launch_gemm<N, 1, WRn, WCn, true>(c, run, CAT_OUTPROJ, "d att", w.out_w, ntiles_n, 1, al, ep, nullptr, N)) return rc;
    }
    br.slot = run.slot;
  }
  // 9. attention backward
  {
    const int nkb = (geom.len + 31) / 32;
    const float scale = 1.0f / sqrtf((float)DH);
    float* stats = DATT == nullptr ? nullptr : br.ws + br.pl.dy1;   // dy1 is dead by now: reuse it for [token][head][4]
    auto launch = [&](auto kern0, auto kern1, int threads) -> int {
      const size_t lds = AttnBwdShape<DH>::lds_bytes(nkb);
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern0), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return c->fail(DPTNAV_ERR_HIP, "attention bwd lds: %s", hipGetErrorString(e));
      ProfScope ps(c, CAT_ATTN, st);
      const DropCfg drop = c->drop_cfg(block, path, true, br.half);
      const unsigned long long* amask = reinterpret_cast<const unsigned long long*>(tape + tp.amask);
      hipLaunchKernelGGL(kern0, dim3(geom.nseq, g.num_heads), dim3(threads), lds, st, qkv, att, DATT, DQKV, stats, g.num_heads, N,
                         geom, scale, drop, reinterpret_cast<const float2*>(tape + tp.astats), amask);
      hipLaunchKernelGGL(kern1, dim3(geom.nseq, g.num_heads), dim3(threads), lds, st, qkv, att, DATT, DQKV, stats, g.num_heads, N,
                         geom, scale, drop, (const float2*)nullptr, amask);
      return DPTNAV_OK;
    };
    int rc = DPTNAV_OK;
    switch (nkb) {
      case 1: rc = launch(attention_bwd_kernel<DH, 1, 0>, attention_bwd_kernel<DH, 1, 1>, 64); break;
      case 2: rc = launch(attention_bwd_kernel<DH, 2, 0>, attention_bwd_kernel<DH, 2, 1>, 128); break;
      case 3: rc = launch(attention_bwd_kernel<DH, 3, 0>, attention_bwd_kernel<DH, 3, 1>, 192); break;
      case 4: rc = launch(attention_bwd_kernel<DH, 4, 0>, attention_bwd_kernel<DH, 4, 1>, 256); break;
      case 5: rc = launch(attention_bwd_kernel<DH, 5, 0>, attention_bwd_kernel<DH, 5, 1>, 320); break;
      case 6: rc = launch(attention_bwd_kernel<DH, 6, 0>, attention_bwd_kernel<DH, 6, 1>, 384); break;
      case 7: rc = launch(attention_bwd_kernel<DH, 7, 0>, attention_bwd_kernel<DH, 7, 1>, 448); break;
      case 8: rc = launch(attention_bwd_kernel<DH, 8, 0>, attention_bwd_kernel<DH, 8, 1>, 512); break;
      default: return c->fail(DPTNAV_ERR_INVALID, "attention bwd: sequence length %d > 256", geom.len);
    }
    if (rc) return rc;
    LAUNCH_CHECK(c, "attention bwd");
  }
  // 10. in-projection gradients and d x = dz1 (residual) + dqkv W_in
  {
    ALoadCols yl{DQKV, M, 3 * N, 0, 32};
    ALoadDense xl{x_in, M, N, 32};
    if constexpr (N == 128) {
      if (int rc = launch_wgrad<3 * N, N>(c, br, "d in weight + bias", ntiles, yl, xl, G("mha.in_proj_weight"),
                                          G("mha.in_proj_bias")))
        return rc;
    } else {
      if (int rc = launch_wgrad_generic<3 * N, N>(c, br, "d in weight", ntiles, yl, xl, G("mha.in_proj_weight"))) return rc;
      if (int rc = launch_colsum<3 * N>(c, br, "d in bias", DQKV, M, 3 * N, 0, G("mha.in_proj_bias"))) return rc;
    }
  }
  run.slot = br.slot;
  {
    int t = 0;
    if constexpr (N == 128) {      // d x = dz + d qkv W_in (K = 384) by dgrad_t.hip
      t = try_dgrad_t(c, run, CAT_QKV, "d x", DQKV, 3 * N, 384, wpk + BWD_PACK_INW, DZ, d_in, M);
      if (t < 0) return -t;
    }
    if (!t) {
      ALoadDense al{DQKV, M, 3 * N, BMn};
      EpiAddMaskStoreT<true, false> ep{d_in, DZ, nullptr, M, N, BMn, N};
      if (int rc = launch_gemm<3 * N, 1, WRn, WCn, true>(c, run, CAT_QKV, "d x", w.in_w, ntiles_n, 1, al, ep, nullptr, N)) return rc;
    }
  }
  br.slot = run.slot;
  return DPTNAV_OK;
}

// =================================================================================================
// training step, whole model
// =================================================================================================
struct ModelTape {   // offsets in floats
  size_t E, vid, Z, X0;            // X0: (2*num_blocks + 1) token buffers, M*N each
  size_t paths;                    // 2*num_blocks path tapes
  size_t path_stride, x_stride, total;
  PathTape pt;
};

int make_model_tape(dptnav_ctx* c, int B, int64_t L, int S, int Tv, ModelTape* t) {
  const dptnav_config& g = c->cfg;
  const int64_t N = g.num_features, K = g.chunk_size, M = (int64_t)B * S * K;
  make_path_tape(c, B, S, &t->pt);
  size_t o = 0;
  auto take = [&](size_t n) { size_t at = o; o += align64(n); return at; };
  t->E = take((size_t)B * L * N);
  t->vid = take((size_t)B * (Tv > 0 ? Tv : 1) * N);
  t->Z = take((size_t)M * 2 * N);
  t->x_stride = align64((size_t)M * N);
  t->X0 = take(t->x_stride * (2 * g.num_blocks + 1));
  t->path_stride = align64(t->pt.total);
  t->paths = take(t->path_stride * 2 * g.num_blocks);
  t->total = o;
  return DPTNAV_OK;
}

template <int N>
int run_tail_backward(dptnav_ctx* c, BwdRun& br, Run& run, const float* x, const float* E, const float* Z,
                      const float* d_s1, const float* d_s2, float* d_x, int B, int64_t T, int64_t L, int S) {
  constexpr int GROUP = N / 4;
  constexpr int WRn = N == 128 ? 1 : 2, WCn = N == 128 ? 4 : 2, BMn = 32 * WRn;   // tiles of the GEMMs whose output is N wide
  const dptnav_config& g = c->cfg;
  hipStream_t st = br.st;
  const int K = g.chunk_size, P = g.step_size;
  const int64_t M = (int64_t)B * S * K, rows = (int64_t)2 * B * L;
  const int ola = (int)((S - 1) * P + K), left = (int)((L - ola) / 2);
  const int64_t ndec = (L - 1) * c->stride + g.kernel_size_enc;
  const int pad_left = (int)((T - ndec) / 2);
  float *DQ = br.ws + br.pl.dq, *DU = br.ws + br.pl.du, *DZs = br.ws + br.pl.dqkv, *LNP = br.ws + br.pl.lnp,
        *slab = br.ws + br.pl.slab;
  auto G = [&](const char* name) { return br.gptr[c->slot(name)]; };
  int grid = 0;
  // T2 recompute: q = OLA(Z) W_post^T + b_post + E ; d q, d decoder.weight
  {
    ALoadOla al{Z, N, B, (int)L, S, K, P, left, ola, BMn};
    EpiDecoderBwd<GROUP> ep{DQ, c->w("dprnn.postprocessing.0.bias"), E, c->w("decoder.weight"), d_s1, d_s2, LNP,
                            (int64_t)B * L, T, (int)L, g.kernel_size_enc, c->stride, pad_left, BMn};
    if (int rc = launch_gemm<N, 1, WRn, WCn>(c, run, CAT_POST, "postproc recompute + decoder bwd",
                                            c->w("dprnn.postprocessing.0.weight"), (rows + BMn - 1) / BMn, 1, al, ep, nullptr, N, &grid))
      return rc;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((N * 8 + 31) / 32), dim3(256), 0, st, LNP, grid, (int64_t)N * 8, slab, 0);
    hipLaunchKernelGGL(decoder_wgrad_finish_kernel, dim3((N * g.kernel_size_enc + 255) / 256), dim3(256), 0, st, slab,
                       G("decoder.weight"), N, g.kernel_size_enc);
    LAUNCH_CHECK(c, "decoder weight grad");
  }
  br.slot = run.slot;
  // post-processing conv gradients; d u = d q W_post
  if (int rc = launch_colsum<N>(c, br, "d postproc bias", DQ, rows, N, 0, G("dprnn.postprocessing.0.bias"))) return rc;
  {
    ALoadCols yl{DQ, rows, N, 0, 32};
    ALoadOla xl{Z, N, B, (int)L, S, K, P, left, ola, 32};
    if constexpr (N == 128) {
      if (int rc = launch_wgrad<N, N>(c, br, "d postproc weight", (rows + 31) / 32, yl, xl, G("dprnn.postprocessing.0.weight")))
        return rc;
    } else {
      if (int rc = launch_wgrad_generic<N, N>(c, br, "d postproc weight", (rows + 31) / 32, yl, xl, G("dprnn.postprocessing.0.weight")))
        return rc;
    }
  }
  run.slot = br.slot;
  {
    ALoadDense al{DQ, rows, N, BMn};
    EpiAddMaskStoreT<false, false> ep{DU, nullptr, nullptr, rows, N, BMn, N};
    if (int rc = launch_gemm<N, 1, WRn, WCn, true>(c, run, CAT_POST, "d u", c->w("dprnn.postprocessing.0.weight"),
                                                  (rows + BMn - 1) / BMn, 1, al, ep, nullptr, N))
      return rc;
  }
  br.slot = run.slot;
  // overlap-add backward -> d Z ; separation conv gradients ; PReLU backward -> d x
  hipLaunchKernelGGL(ola_grad_gather_kernel, dim3((unsigned)M), dim3(64), 0, st, DU, DZs, N, B, (int)L, S, K, P, left);
  LAUNCH_CHECK(c, "ola backward");
  if (int rc = launch_colsum<2 * N>(c, br, "d sep bias", DZs, M, 2 * N, 0, G("dprnn.speakers_separation.1.bias"))) return rc;
  {
    ALoadCols yl{DZs, M, 2 * N, 0, 32};
    ALoadDensePReLU xl{x, c->w("dprnn.speakers_separation.0.weight"), M, N, 32};
    if (int rc = launch_wgrad<2 * N, N>(c, br, "d sep weight", (M + 31) / 32, yl, xl, G("dprnn.speakers_separation.1.weight")))
      return rc;
  }
  run.slot = br.slot;
  {
    ALoadDense al{DZs, M, 2 * N, BMn};
    EpiPReLUBwd ep{d_x, x, c->w("dprnn.speakers_separation.0.weight"), LNP, M, N, BMn};
    if (int rc = launch_gemm<2 * N, 1, WRn, WCn, true>(c, run, CAT_SEP, "d prelu", c->w("dprnn.speakers_separation.1.weight"),
                                                      (M + BMn - 1) / BMn, 1, al, ep, nullptr, N, &grid))
      return rc;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(1), dim3(256), 0, st, LNP, grid, (int64_t)1,
                       G("dprnn.speakers_separation.0.weight"), 0);
    LAUNCH_CHECK(c, "prelu slope grad");
  }
  br.slot = run.slot;
  return DPTNAV_OK;
}

template <int N>
int run_head_backward(dptnav_ctx* c, BwdRun& br, const float* mix, const float* e1, const float* e2, const float* vid,
                      const float* dX0, int B, int64_t T, int64_t L, int S, int Tv) {
  const dptnav_config& g = c->cfg;
  hipStream_t st = br.st;
  constexpr int FPB = 256 / (N / 4);
  float *DQ = br.ws + br.pl.dq, *DE = br.ws + br.pl.de, *DVI = br.ws + br.pl.dvi, *DV = br.ws + br.pl.dv,
        *slab = br.ws + br.pl.slab, *red = br.ws + br.pl.lnp;
  auto G = [&](const char* name) { return br.gptr[c->slot(name)]; };
  const bool av = !g.audio_only;
  const unsigned gx = (unsigned)((L + FPB - 1) / FPB);
  hipLaunchKernelGGL(head_bwd_frames_kernel<N>, dim3(gx, B), dim3(256), 0, st, DQ, dX0, av ? vid : nullptr,
                     av ? c->w("gate") : nullptr, av ? c->w("video_ln.weight") : nullptr, DE, DVI, slab, B, (int)L, Tv, S,
                     g.chunk_size, g.step_size);
  LAUNCH_CHECK(c, "head backward frames");
  if (av) {
    const int64_t cnt = 2 * N + 4;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((cnt + 31) / 32)), dim3(256), 0, st, slab, (int)(gx * B), cnt, red, 0);
    hipLaunchKernelGGL(gate_grad_finish_kernel, dim3(1), dim3(128), 0, st, red, c->w("gate"), c->w("video_ln.bias"), G("gate"),
                       G("video_ln.weight"), G("video_ln.bias"), N);
    hipLaunchKernelGGL(interp_bwd_kernel, dim3(Tv, B), dim3(512), 0, st, DVI, DV, N, (int)L, Tv);
    {
      if (Tv > 256) return c->fail(DPTNAV_ERR_INVALID, "training step: more than 256 video frames (Tv=%d)", Tv);
      const int half = g.hidden_video / 2;
      const int64_t wcnt = (int64_t)half * g.video_emb_size;
      float* colslab = slab + (size_t)B * wcnt;
      hipLaunchKernelGGL(video_linear_bwd_kernel, dim3(half, B), dim3(256), 0, st, DV, e1, e2, slab, colslab, g.video_emb_size, Tv, half);
      hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((wcnt + 31) / 32)), dim3(256), 0, st, slab, B, wcnt,
                         G("visual_compression.weight"), 0);
      hipLaunchKernelGGL(slab_reduce_kernel, dim3((half + 31) / 32), dim3(256), 0, st, colslab, B, (int64_t)half,
                         G("visual_compression.bias"), 0);
    }
    LAUNCH_CHECK(c, "video branch backward");
  }
  {
    const int fpb = 512;
    const unsigned ex = (unsigned)((L + fpb - 1) / fpb);
    hipLaunchKernelGGL(encoder_wgrad_kernel<N>, dim3(ex, B), dim3(256), 0, st, DE, mix, slab, B, T, (int)L, g.kernel_size_enc,
                       c->stride, fpb);
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((N * 8 + 31) / 32), dim3(256), 0, st, slab, (int)(ex * B), (int64_t)N * 8, red, 0);
    hipLaunchKernelGGL(decoder_wgrad_finish_kernel, dim3((N * g.kernel_size_enc + 255) / 256), dim3(256), 0, st, red,
                       G("encoder.weight"), N, g.kernel_size_enc);
    LAUNCH_CHECK(c, "encoder weight grad");
  }
  return DPTNAV_OK;
}

int check_common(dptnav_ctx* c, int B, int64_t T, int Tv, void* ws, size_t ws_bytes, Plan* pl) {
  if (!c->bound) return c->fail(DPTNAV_ERR_WEIGHTS, "weights not bound: call dptnav_bind_weights first");
  if (int rc = make_plan(c, B, T, Tv, pl)) return rc;
  if (ws == nullptr || ((uintptr_t)ws & 255) != 0)
    return c->fail(DPTNAV_ERR_WORKSPACE, "workspace must be non-null and 256-byte aligned");
  if (ws_bytes < pl->total * sizeof(float))
    return c->fail(DPTNAV_ERR_WORKSPACE, "workspace too small: %zu < %zu bytes", ws_bytes, pl->total * sizeof(float));
  return DPTNAV_OK;
}

// Start a pass on `st` over the workspace slice `ws`: zero its ticket counters (stream ordered).
int begin_run(dptnav_ctx* c, Run* run, float* ws, const Plan& pl, hipStream_t st) {
  run->ws = ws;
  run->pl = pl;
  run->st = st;
  run->slot = 0;
  run->packed = false;
  run->packed_wih = false;
  run->packed_whh4 = false;
  hipError_t e = hipMemsetAsync(ws + pl.queue, 0, QUEUE_SLOTS * sizeof(unsigned), st);
  if (e != hipSuccess) return c->fail(DPTNAV_ERR_HIP, "ticket counter reset: %s", hipGetErrorString(e));
  return DPTNAV_OK;
}

// Join the caller's stream to BOTH internal streams (always, also after a failed enqueue: kernels already queued there
// must be ordered before whatever the caller does next) and return the body's error, or the join's if the body was fine.
int join_after(dptnav_ctx* c, hipStream_t st, bool forked, int rc_body, int nstreams = 2) {
  if (!forked) return rc_body;
  const std::string body_err = c->err;
  bool ok = true;
  for (int s = 0; s < nstreams; ++s)
    ok &= hipEventRecord(c->ev_join[s], c->streams[s]) == hipSuccess && hipStreamWaitEvent(st, c->ev_join[s], 0) == hipSuccess;
  if (rc_body) {
    c->err = body_err;
    return rc_body;
  }
  return ok ? DPTNAV_OK : c->fail(DPTNAV_ERR_HIP, "join event");
}

}  // namespace

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

int dptnav_abi_version(void) { return DPTNAV_ABI_VERSION; }

const char* dptnav_last_error(dptnav_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int dptnav_create(const dptnav_config* cfg, dptnav_handle* out) {
  auto bad = [&](const char* m) {
    g_create_error = m;
    return DPTNAV_ERR_INVALID;
  };
  if (!cfg || !out) return bad("null argument");
  *out = nullptr;
  if (cfg->num_features != 128 && cfg->num_features != 64) return bad("num_features must be 128 or 64");
  if (cfg->hidden_dim != 128) return bad("hidden_dim must be 128");
  if (cfg->arch != 0 && cfg->arch != 1) return bad("arch must be 0 (DPTN) or 1 (DPRNN)");
  if (cfg->arch == 0 && cfg->num_heads != 4) return bad("num_heads must be 4 (head dim 32 or 16)");
  if (cfg->kernel_size_enc < 2 || cfg->kernel_size_enc > 8) return bad("kernel_size_enc must be in [2,8]");
  if (cfg->num_blocks < 1) return bad("num_blocks must be >= 1");
  if (cfg->chunk_size < 1 || (cfg->arch == 0 && cfg->chunk_size > 256))
    return bad("chunk_size must be in [1,256] (attention kernel limit; DPRNN has none)");
  if (cfg->step_size < 1 || cfg->step_size > cfg->chunk_size) return bad("step_size must be in [1,chunk_size]");
  if (!cfg->audio_only) {
    if (cfg->hidden_video != cfg->num_features) return bad("hidden_video must equal num_features");
    if (cfg->video_emb_size < 1) return bad("video_emb_size must be >= 1");
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return bad("no HIP device visible: libdptnav has no CPU path");
  dptnav_ctx* c = new dptnav_ctx();
  c->cfg = *cfg;
  c->stride = cfg->kernel_size_enc / 2;
  c->dh = cfg->num_features / cfg->num_heads;
  {
    int devid = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&devid) == hipSuccess) c->device_id = devid;
    if (hipGetDeviceProperties(&prop, devid) == hipSuccess &&
        prop.multiProcessorCount > 0)
      c->num_cus = prop.multiProcessorCount;
  }
  build_names(c);
  *out = c;
  return DPTNAV_OK;
}

void dptnav_destroy(dptnav_handle h) {
  if (!h) return;
  for (ProfRec& r : h->prof_pending) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
  for (hipEvent_t e : h->prof_pool) hipEventDestroy(e);
  h->release_streams();
  for (hipEvent_t e : h->ev_sub) hipEventDestroy(e);
  delete h;
}

int dptnav_num_weights(dptnav_handle h) { return h ? (int)h->names.size() : 0; }
const char* dptnav_weight_name(dptnav_handle h, int s) {
  return (h && s >= 0 && s < (int)h->names.size()) ? h->names[s].c_str() : nullptr;
}
int64_t dptnav_weight_numel(dptnav_handle h, int s) {
  return (h && s >= 0 && s < (int)h->numel.size()) ? h->numel[s] : -1;
}

int dptnav_bind_weights(dptnav_handle h, const float* const* dev_ptrs, int n) {
  if (!h) return DPTNAV_ERR_INVALID;
  if (!dev_ptrs || n != (int)h->names.size())
    return h->fail(DPTNAV_ERR_WEIGHTS, "expected %zu weight pointers, got %d", h->names.size(), n);
  for (int i = 0; i < n; ++i) {
    if (dev_ptrs[i] == nullptr) return h->fail(DPTNAV_ERR_WEIGHTS, "null pointer for %s", h->names[i].c_str());
    if (((uintptr_t)dev_ptrs[i] & 15) != 0)
      return h->fail(DPTNAV_ERR_WEIGHTS, "%s is not 16-byte aligned", h->names[i].c_str());
  }
  h->ptr.assign(dev_ptrs, dev_ptrs + n);
  h->pw.clear();
  for (int b = 0; b < h->cfg.num_blocks; ++b)
    for (int p = 0; p < 2; ++p) h->pw.push_back(path_weights(h, b, p));
  h->bound = true;
  return DPTNAV_OK;
}

int64_t dptnav_frames(dptnav_handle h, int64_t T) {
  return h ? (T - h->cfg.kernel_size_enc) / h->stride + 1 : -1;
}
int64_t dptnav_chunks(dptnav_handle h, int64_t T) {
  return h ? (dptnav_frames(h, T) - h->cfg.chunk_size) / h->cfg.step_size + 1 : -1;
}

// How dptnav_forward cuts a batch: sub-batches small enough that every recurrence launch fits the 16-sequence-tile kernel
// in one round (2 * ceil(sequences / 16) <= CUs on both paths), as few of them as possible, at least two (the halves of
// round 1) so that one sub-batch's recurrence always has another one's GEMM / attention kernels beside it.  Sequences
// too long for that (fewer than 4 mixtures would fit) keep the plain two halves.
constexpr int MAX_SUB = 32;
// -> number of sub-batches; *inflight (optional) = how many of their recurrence launches may be in flight together.
// Round 3 measurements (tools/split_sweep.py -> profiles/r03_split_sweep.txt; B x sub-batches x launches in flight):
//  * the event chain of rounds 1-2 (ONE recurrence in flight: it had paid with the static-grid GEMMs of round 1) now
//    costs time at every batch size -- nothing at B >= 20, 1-5 % at B = 12..16, 29 % (N = 128) / 39 % (N = 64) at B = 4,
//    where both sub-batches' recurrences fit the chip side by side and the chain serialised them.  Default: no chain
//    (option lstm_chain = 1 restores it; lstm_inflight = n sets the depth for experiments);
//  * sub-batches: as few as make every recurrence launch fit the 16-sequence-tile kernel in one round, at least two
//    (8 + 8 at B = 16, 12 + 12 at B = 24, 11 + 11 + 10 at B = 32).  One exception, option split_policy = 1 (default): when
//    a half-batch launch needs more than half of the CUs but a third needs less (B = 13..18: 6 + 5 + 5 mixtures = 106 +
//    94 + 94 workgroups) three sub-batches are 1 % faster for N = 128 (30.6 vs 31.0 ms at B = 16) and equal for N = 64.
static int forward_split(dptnav_handle h, int B, int64_t T, int Tv, int* sizes, int* inflight = nullptr) {
  if (inflight) *inflight = 1;
  if (!h->opt_overlap || B < 2) { sizes[0] = B; return 1; }
  Plan pl;
  int nsub = 2, depth = 1;
  if (h->opt_lstm16 && make_plan(h, 1, T, Tv, &pl) == DPTNAV_OK) {
    const int64_t S = pl.S, K = h->cfg.chunk_size;
    const int ndir_inter = h->cfg.bidir ? 2 : 1;
    auto wgs = [&](int b) { return std::max(((b * S + 15) / 16) * 2, ((b * K + 15) / 16) * ndir_inter); };
    int bfit = 0;
    while (bfit < B && wgs(bfit + 1) <= h->num_cus) ++bfit;
    if (bfit >= 4) nsub = std::max(2, (B + bfit - 1) / bfit);
    if (h->opt_split_policy == 1 && h->cfg.arch == 0) {
      int bpair = 0;
      while (bpair < B && 2 * wgs(bpair + 1) <= h->num_cus) ++bpair;
      const int n2 = bpair >= 2 ? (B + bpair - 1) / bpair : 0;
      // exactly three: four or more small sub-batches lose to two large ones (B = 24: 12 + 12 mixtures, 226 workgroups per launch)
      if (n2 == 3) { nsub = n2; depth = 2; }       // depth: only with the chain (lstm_chain = 1)
    }
  }
  if (h->opt_sub_batches > 0) { nsub = h->opt_sub_batches; depth = 1; }
  if (!h->opt_lstm_chain) depth = MAX_SUB;       // no chain: every sub-batch's recurrence may be in flight
  if (h->opt_lstm_inflight > 0) depth = h->opt_lstm_inflight;
  if (nsub > MAX_SUB) nsub = MAX_SUB;
  if (nsub > B) nsub = B;
  for (int i = 0; i < nsub; ++i) sizes[i] = B / nsub + (i < B % nsub ? 1 : 0);
  if (inflight) *inflight = std::max(1, std::min(depth, nsub));
  return nsub;
}

size_t dptnav_workspace_bytes(dptnav_handle h, int B, int64_t T, int Tv) {
  if (!h) return 0;
  // dptnav_forward (option overlap=1) runs B >= 2 as independently planned sub-batches; the stage entry points and
  // overlap=0 use one plan for the whole batch.  The workspace covers whichever of the two is valid/larger.
  Plan pl;
  size_t need = 0;
  if (make_plan(h, B, T, Tv, &pl) == DPTNAV_OK) need = pl.total;
  if (B >= 2) {
    const bool keep = h->opt_overlap;
    h->opt_overlap = true;                       // size for the split even if the option is switched on later
    int sizes[MAX_SUB];
    const int nsub = forward_split(h, B, T, Tv, sizes);
    h->opt_overlap = keep;
    size_t sum = 0;
    bool ok = true;
    for (int i = 0; i < nsub && ok; ++i) {
      Plan a;
      ok = make_plan(h, sizes[i], T, Tv, &a) == DPTNAV_OK;
      sum += (a.total + 63) & ~(size_t)63;
    }
    if (ok && sum > need) need = sum;
  }
  return need * sizeof(float);
}

int dptnav_stage_head(dptnav_handle h, const float* mix, const float* e1, const float* e2, int B, int64_t T, int Tv,
                      float* encoded, float* chunked, void* ws, size_t ws_bytes, void* stream) {
  if (!h) return DPTNAV_ERR_INVALID;
  Plan pl;
  if (int rc = check_common(h, B, T, Tv, ws, ws_bytes, &pl)) return rc;
  if (!mix || !encoded || !chunked || (!h->cfg.audio_only && (!e1 || !e2 || Tv < 1)))
    return h->fail(DPTNAV_ERR_INVALID, "null tensor argument");
  Run run;
  if (int rc = begin_run(h, &run, (float*)ws, pl, (hipStream_t)stream)) return rc;
  return h->cfg.num_features == 128 ? run_head<128>(h, run, mix, e1, e2, B, T, Tv, encoded, chunked)
                                    : run_head<64>(h, run, mix, e1, e2, B, T, Tv, encoded, chunked);
}

int dptnav_stage_path(dptnav_handle h, int block, int path, const float* x_in, float* x_out, int B, int S, void* ws,
                      size_t ws_bytes, void* stream) {
  if (!h) return DPTNAV_ERR_INVALID;
  if (block < 0 || block >= h->cfg.num_blocks || (path != 0 && path != 1))
    return h->fail(DPTNAV_ERR_INVALID, "bad block/path %d/%d", block, path);
  if (!x_in || !x_out || x_in == x_out) return h->fail(DPTNAV_ERR_INVALID, "x_in/x_out null or aliased");
  if (S < 1 || S > 65535) return h->fail(DPTNAV_ERR_INVALID, "S=%d out of range", S);
  // a T that yields exactly S chunks
  const int64_t L = (int64_t)(S - 1) * h->cfg.step_size + h->cfg.chunk_size;
  const int64_t T = (L - 1) * h->stride + h->cfg.kernel_size_enc;
  Plan pl;
  if (int rc = check_common(h, B, T, 1, ws, ws_bytes, &pl)) return rc;
  Run run;
  if (int rc = begin_run(h, &run, (float*)ws, pl, (hipStream_t)stream)) return rc;
  run.fuse128 = h->fuse128_for(B);
  run.batch_total = B;
  return h->cfg.num_features == 128 ? run_path<128>(h, run, block, path, x_in, x_out, B, S)
                                    : run_path<64>(h, run, block, path, x_in, x_out, B, S);
}

int dptnav_stage_tail(dptnav_handle h, const float* x, const float* encoded, int B, int64_t T, float* s1, float* s2,
                      void* ws, size_t ws_bytes, void* stream) {
  if (!h) return DPTNAV_ERR_INVALID;
  Plan pl;
  if (int rc = check_common(h, B, T, 1, ws, ws_bytes, &pl)) return rc;
  if (!x || !encoded || !s1 || !s2) return h->fail(DPTNAV_ERR_INVALID, "null tensor argument");
  Run run;
  if (int rc = begin_run(h, &run, (float*)ws, pl, (hipStream_t)stream)) return rc;
  return h->cfg.num_features == 128 ? run_tail<128>(h, run, x, encoded, B, T, s1, s2)
                                    : run_tail<64>(h, run, x, encoded, B, T, s1, s2);
}

int dptnav_forward(dptnav_handle h, const float* mix, const float* e1, const float* e2, int B, int64_t T, int Tv,
                   float* s1, float* s2, void* ws, size_t ws_bytes, void* stream) {
  if (!h) return DPTNAV_ERR_INVALID;
  if (!h->bound) return h->fail(DPTNAV_ERR_WEIGHTS, "weights not bound: call dptnav_bind_weights first");
  if (!mix || !s1 || !s2 || (!h->cfg.audio_only && (!e1 || !e2 || Tv < 1)))
    return h->fail(DPTNAV_ERR_INVALID, "null tensor argument");
  if (ws == nullptr || ((uintptr_t)ws & 255) != 0)
    return h->fail(DPTNAV_ERR_WORKSPACE, "workspace must be non-null and 256-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const bool big = h->cfg.num_features == 128;
  const dptnav_config& g = h->cfg;

  // Mixtures are independent, so the batch is processed as sub-batches on two internal streams (forward_split): a
  // recurrence launch occupies one CU per (direction, 16-sequence tile), and the other stream's GEMM / attention
  // launches (dynamic tile tickets) fill the rest of the chip meanwhile.  Fork/join by events on the caller's stream.
  int Bs[MAX_SUB];
  int depth = 1;
  const int nsub = forward_split(h, B, T, Tv, Bs, &depth);
  Plan pl[MAX_SUB];
  size_t need = 0, base[MAX_SUB];
  for (int i = 0; i < nsub; ++i) {
    if (int rc = make_plan(h, Bs[i], T, Tv, &pl[i])) return rc;
    base[i] = need;
    need += (pl[i].total + 63) & ~(size_t)63;
  }
  if (ws_bytes < need * sizeof(float))
    return h->fail(DPTNAV_ERR_WORKSPACE, "workspace too small: %zu < %zu bytes", ws_bytes, need * sizeof(float));
  // sub-batches go round robin over min(nsub, NSTREAMS) internal streams (two for the default two halves)
  const int nstr = std::min(nsub, (int)dptnav_ctx::NSTREAMS);
  // option serialize: the same sub-batches and kernels, one after the other on the caller's stream (every launch alone on
  // the chip: what bench.py's roofline times); no internal stream, no event
  const bool forked = nsub > 1 && !h->opt_serialize;
  if (forked) {
    if (int rc = h->ensure_streams()) return rc;
    if (int rc = h->ensure_sub_events(nsub)) return rc;
    if (hipEventRecord(h->ev_fork, st) != hipSuccess) return h->fail(DPTNAV_ERR_HIP, "fork event");
    for (int s = 0; s < nstr; ++s)
      if (hipStreamWaitEvent(h->streams[s], h->ev_fork, 0) != hipSuccess) return h->fail(DPTNAV_ERR_HIP, "fork wait");
  }
  const int64_t Cv = g.audio_only ? 0 : g.video_emb_size;
  Run run[MAX_SUB];
  const float *mixi[MAX_SUB], *e1i[MAX_SUB], *e2i[MAX_SUB];
  float *s1i[MAX_SUB], *s2i[MAX_SUB];
  // Everything between the fork and the join runs inside `enqueue`: whatever it returns, the caller's stream is joined
  // to the two internal streams afterwards, so work already enqueued there is ordered before anything the caller does
  // next with the workspace / output tensors (it may free them when it sees the error).
  auto enqueue = [&]() -> int {
  int64_t b0 = 0;
  for (int i = 0; i < nsub; ++i) {
    hipStream_t si = forked ? h->streams[i % nstr] : st;   // sub-batches go round robin over the internal streams
    if (int rc = begin_run(h, &run[i], (float*)ws + base[i], pl[i], si)) return rc;
    run[i].fuse128 = h->fuse128_for(B);      // by the forward's whole batch, not the sub-batch
    run[i].batch_total = B;
    mixi[i] = mix + b0 * T;
    e1i[i] = e1 ? e1 + b0 * Cv * Tv : nullptr;
    e2i[i] = e2 ? e2 + b0 * Cv * Tv : nullptr;
    s1i[i] = s1 + b0 * T;
    s2i[i] = s2 + b0 * T;
    b0 += Bs[i];
  }
  // The sub-batches advance in lock step on the host, but their recurrences are chained by events
  // (L(s0,p) -> L(s1,p) -> ... -> L(s0,p+1) ...): at any time at most ONE sub-batch sits in the LSTM while the
  // other stream's GEMM / attention kernels use the remaining CUs -- without the chain the streams reach the
  // recurrence together and nothing is gained.
  auto E = [&](int i) { return run[i].ws + run[i].pl.E; };
  auto X0 = [&](int i) { return run[i].ws + run[i].pl.X0; };
  auto X1 = [&](int i) { return run[i].ws + run[i].pl.X1; };
  for (int i = 0; i < nsub; ++i) {
    int rc = big ? run_head<128>(h, run[i], mixi[i], e1i[i], e2i[i], Bs[i], T, Tv, E(i), X0(i))
                 : run_head<64>(h, run[i], mixi[i], e1i[i], e2i[i], Bs[i], T, Tv, E(i), X0(i));
    if (rc) return rc;
    if (big && g.arch == 0 && h->opt_fuse_attn && !h->opt_split_bf16) {
      if (int rc2 = pack_attn_weights(h, run[i].st, 0, 2 * g.num_blocks, run[i].ws + run[i].pl.wpack)) return rc2;
      run[i].packed = true;
    }
    if (!h->opt_split_bf16)
      if (int rc2 = big ? pack_wih_all<128>(h, run[i]) : pack_wih_all<64>(h, run[i])) return rc2;
  }
  // recurrence launch n (path-major, sub-batch minor) waits for launch n - depth: `depth` recurrences may be in flight
  // (forward_split; > 1 pays when two launches fit the chip together: three or more sub-batches).  ev_sub[j] is re-recorded by sub-batch j once per path, so its latest record IS launch n - depth.
  int nlaunch = 0;
  for (int b = 0; b < g.num_blocks; ++b)
    for (int path = 0; path < 2; ++path)
      for (int i = 0; i < nsub; ++i, ++nlaunch) {
        if (forked) {
          run[i].lstm_wait = nlaunch >= depth && depth < nsub ? h->ev_sub[(nlaunch - depth) % nsub] : nullptr;
          run[i].lstm_record = h->ev_sub[i];
        }
        const float* xin = path == 0 ? X0(i) : X1(i);
        float* xout = path == 0 ? X1(i) : X0(i);
        // FFN (K6) of a path runs as the prologue of the NEXT path's attention block when that block is the fused kernel
        // (fp32, N = 128, sequences <= 160): the block input x then never goes through HBM
        int chain = 0;
        if (h->opt_fuse_ffn && !h->opt_split_bf16) {
          const int S_i = (int)pl[i].S;
          const bool first = b == 0 && path == 0, last = b == g.num_blocks - 1 && path == 1;
          auto fusable = [&](int p) { return big ? path_fusable<128>(h, p, Bs[i], S_i) : path_fusable<64>(h, p, Bs[i], S_i); };
          // the prologue contracts over ReLU(h) of BOTH directions (K = 256): a unidirectional inter-chunk LSTM
          // (bidir = false) keeps its own K6 launch
          const bool this_two = h->pw[2 * b + path].ndir == 2;
          const bool prev_two = first || h->pw[2 * b + path - 1].ndir == 2;
          if (!first && prev_two && fusable(path)) chain |= CHAIN_PRO;
          if (!last && this_two && fusable(1 - path)) chain |= CHAIN_SKIP_FFN;
        }
        int rc = big ? run_path<128>(h, run[i], b, path, xin, xout, Bs[i], (int)pl[i].S, nullptr, chain)
                     : run_path<64>(h, run[i], b, path, xin, xout, Bs[i], (int)pl[i].S, nullptr, chain);
        if (rc) return rc;
      }
  for (int i = 0; i < nsub; ++i) {
    int rc = big ? run_tail<128>(h, run[i], X0(i), E(i), Bs[i], T, s1i[i], s2i[i])
                 : run_tail<64>(h, run[i], X0(i), E(i), Bs[i], T, s1i[i], s2i[i]);
    if (rc) return rc;
  }
  return DPTNAV_OK;
  };
  const int rc_body = enqueue();
  return join_after(h, st, forked, rc_body, nstr);
}

int dptnav_workspace_tap(dptnav_handle h, int B, int64_t T, int Tv, const char* name, size_t* off, size_t* numel) {
  if (!h || !name || !off || !numel) return DPTNAV_ERR_INVALID;
  Plan pl;
  if (int rc = make_plan(h, B, T, Tv, &pl)) return rc;
  const std::string n(name);
  if (n == "qkv") { *off = pl.qkv * 4; *numel = pl.qkv_n; }
  else if (n == "att") { *off = pl.att * 4; *numel = pl.att_n; }
  else if (n == "y1") { *off = pl.y1 * 4; *numel = pl.y1_n; }
  else if (n == "hc") { *off = pl.hc * 4; *numel = pl.hc_n; }
  else if (n == "encoded") { *off = pl.E * 4; *numel = (size_t)B * pl.L * h->cfg.num_features; }
  else if (n == "lstm_stamps") { *off = pl.stamps * 4; *numel = (pl.total - pl.stamps); }
  else return h->fail(DPTNAV_ERR_INVALID, "unknown tap '%s'", name);
  return DPTNAV_OK;
}

// ---- loss / metric statistics (A10, A11): one launch, the caller does ONE device->host copy of 12*B floats ----
int dptnav_sisnr_pairs(dptnav_handle h, const float* s1_pred, const float* s2_pred, const float* s1, const float* s2,
                       const float* mix, int B, int64_t T, float* out, void* stream) {
  if (!h) return DPTNAV_ERR_INVALID;
  if (!s1_pred || !s2_pred || !s1 || !s2 || !mix || !out || B < 1 || T < 2)
    return h->fail(DPTNAV_ERR_INVALID, "sisnr_pairs: bad argument");
  hipLaunchKernelGGL(sisnr_pairs_kernel, dim3(B, 6), dim3(256), 0, (hipStream_t)stream, s1_pred, s2_pred, s1, s2, mix, T,
                     out);
  LAUNCH_CHECK(h, "sisnr_pairs");
  return DPTNAV_OK;
}

// ---- the training step's tail on the device (N1): loss forward + backward, clip, AdamW; no host synchronisation ----
int64_t dptnav_flat_offset(dptnav_handle h, int slot) {
  if (!h || slot < 0 || slot > (int)h->numel.size()) return -1;
  int64_t o = 0;
  for (int i = 0; i < slot; ++i) o += (int64_t)align64((size_t)h->numel[i]);
  return o;
}
int64_t dptnav_flat_numel(dptnav_handle h) { return h ? dptnav_flat_offset(h, (int)h->numel.size()) : -1; }

size_t dptnav_tail_scratch_bytes(dptnav_handle h, int B) {
  if (!h || B < 1) return 0;
  return ((size_t)B * 4 * PIT_STAT + CLIP_PARTS) * sizeof(double);
}

int dptnav_pit_sisnr_loss(dptnav_handle h, const float* s1_pred, const float* s2_pred, const float* s1, const float* s2, int B,
                          int64_t T, float grad_scale, float* d_s1_pred, float* d_s2_pred, float* loss_out, void* scratch,
                          size_t scratch_bytes, void* stream) {
  if (!h) return DPTNAV_ERR_INVALID;
  if (!s1_pred || !s2_pred || !s1 || !s2 || !d_s1_pred || !d_s2_pred || !loss_out || B < 1 || T < 2)
    return h->fail(DPTNAV_ERR_INVALID, "pit_sisnr_loss: bad argument");
  if (!scratch || ((uintptr_t)scratch & 7) || scratch_bytes < dptnav_tail_scratch_bytes(h, B))
    return h->fail(DPTNAV_ERR_WORKSPACE, "pit_sisnr_loss: scratch too small / misaligned (need %zu bytes)",
                   dptnav_tail_scratch_bytes(h, B));
  hipStream_t st = (hipStream_t)stream;
  double* stats = (double*)scratch;
  hipLaunchKernelGGL(pit_stats_kernel, dim3(B, 4), dim3(256), 0, st, s1_pred, s2_pred, s1, s2, T, stats);
  hipLaunchKernelGGL(pit_grad_kernel, dim3((unsigned)((T + 1023) / 1024), B, 2), dim3(256), 0, st, s1_pred, s2_pred, s1, s2, B, T,
                     stats, grad_scale, d_s1_pred, d_s2_pred, loss_out);
  LAUNCH_CHECK(h, "pit_sisnr_loss");
  return DPTNAV_OK;
}

int dptnav_grad_clip(dptnav_handle h, float* flat_grad, int64_t n_flat, float max_norm, void* scratch, size_t scratch_bytes,
                     float* norm_out, void* stream) {
  if (!h) return DPTNAV_ERR_INVALID;
  if (!flat_grad || !norm_out || n_flat < 4 || (n_flat & 3) || ((uintptr_t)flat_grad & 15))
    return h->fail(DPTNAV_ERR_INVALID, "grad_clip: flat gradient must be 16-byte aligned with a multiple of 4 floats");
  if (!scratch || ((uintptr_t)scratch & 7) || scratch_bytes < CLIP_PARTS * sizeof(double))
    return h->fail(DPTNAV_ERR_WORKSPACE, "grad_clip: scratch too small / misaligned");
  hipStream_t st = (hipStream_t)stream;
  double* partials = (double*)scratch;
  hipLaunchKernelGGL(sumsq_partials_kernel, dim3(CLIP_PARTS), dim3(256), 0, st, flat_grad, n_flat / 4, partials);
  hipLaunchKernelGGL(clip_scale_kernel, dim3(h->num_cus * 2), dim3(256), 0, st, flat_grad, n_flat / 4, partials, CLIP_PARTS, max_norm,
                     norm_out);
  LAUNCH_CHECK(h, "grad_clip");
  return DPTNAV_OK;
}

int dptnav_adamw_step(dptnav_handle h, const float* flat_grad, float* exp_avg, float* exp_avg_sq, int64_t n_flat, double lr,
                      double beta1, double beta2, double eps, double weight_decay, int step, void* stream) {
  if (!h) return DPTNAV_ERR_INVALID;
  if (!h->bound) return h->fail(DPTNAV_ERR_WEIGHTS, "adamw_step: weights not bound (the step updates the bound parameters in place)");
  if (!flat_grad || !exp_avg || !exp_avg_sq || n_flat != dptnav_flat_numel(h) || step < 1)
    return h->fail(DPTNAV_ERR_INVALID, "adamw_step: bad argument (flat buffers must hold %lld floats, step >= 1)",
                   (long long)dptnav_flat_numel(h));
  // hyper-parameters arrive as doubles (Python floats) and every derived constant is formed in double before it is
  // rounded to fp32 once, as torch does: 1 - 0.999f would already be off by 1.3e-5 relative
  const double bc1 = 1.0 - std::pow(beta1, step), bc2 = 1.0 - std::pow(beta2, step);
  const float step_size = (float)(lr / bc1), inv_sqrt_bc2 = (float)(1.0 / std::sqrt(bc2));
  const float decay = (float)(1.0 - lr * weight_decay);
  const int n = (int)h->names.size();
  int64_t off = 0;
  for (int lo = 0; lo < n; lo += ADAMW_MAX) {
    AdamwArgs a{};
    const int cnt = std::min(ADAMW_MAX, n - lo);
    for (int e = 0; e < cnt; ++e) {
      a.param[e] = const_cast<float*>(h->ptr[lo + e]);
      a.off[e] = off;
      a.n[e] = (int)h->numel[lo + e];
      off += (int64_t)align64((size_t)h->numel[lo + e]);
    }
    hipLaunchKernelGGL(adamw_kernel, dim3(cnt, ADAMW_YBLOCKS), dim3(256), 0, (hipStream_t)stream, a, flat_grad, exp_avg, exp_avg_sq,
                       (float)beta1, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, decay, step_size,
                       inv_sqrt_bc2);
    LAUNCH_CHECK(h, "adamw_step");
  }
  return DPTNAV_OK;
}

// ---- training step, path level ------------------------------------------------------------------------
int dptnav_bind_grads(dptnav_handle h, float* const* dev_ptrs, int n) {
  if (!h) return DPTNAV_ERR_INVALID;
  if (!dev_ptrs || n != (int)h->names.size())
    return h->fail(DPTNAV_ERR_WEIGHTS, "expected %zu gradient pointers, got %d", h->names.size(), n);
  for (int i = 0; i < n; ++i)
    if (dev_ptrs[i] == nullptr || ((uintptr_t)dev_ptrs[i] & 15) != 0)
      return h->fail(DPTNAV_ERR_WEIGHTS, "gradient pointer for %s is null or not 16-byte aligned", h->names[i].c_str());
  h->gptr.assign(dev_ptrs, dev_ptrs + n);
  return DPTNAV_OK;
}
size_t dptnav_train_path_tape_bytes(dptnav_handle h, int B, int S) {
  if (!h) return 0;
  PathTape t;
  make_path_tape(h, B, S, &t);
  return t.total * sizeof(float);
}
size_t dptnav_train_bwd_workspace_bytes(dptnav_handle h, int B, int S) {
  if (!h) return 0;
  BwdPlan p;
  make_bwd_plan(h, B, S, &p);
  return p.total * sizeof(float);
}
int dptnav_train_path_forward(dptnav_handle h, int block, int path, const float* x_in, float* x_out, int B, int S,
                              void* tape, size_t tape_bytes, void* ws, size_t ws_bytes, void* stream) {
  if (!h) return DPTNAV_ERR_INVALID;
  if (block < 0 || block >= h->cfg.num_blocks || (path != 0 && path != 1) || !x_in || !x_out || !tape)
    return h->fail(DPTNAV_ERR_INVALID, "train_path_forward: bad argument");
  PathTape tp;
  make_path_tape(h, B, S, &tp);
  if (tape_bytes < tp.total * sizeof(float)) return h->fail(DPTNAV_ERR_WORKSPACE, "tape too small");
  const int64_t L = (int64_t)(S - 1) * h->cfg.step_size + h->cfg.chunk_size;
  const int64_t T = (L - 1) * h->stride + h->cfg.kernel_size_enc;
  Plan pl;
  if (int rc = check_common(h, B, T, 1, ws, ws_bytes, &pl)) return rc;
  Run run;
  if (int rc = begin_run(h, &run, (float*)ws, pl, (hipStream_t)stream)) return rc;
  float* tb = (float*)tape;
  PathBufs pb{tb + tp.qkv, tb + tp.att, tb + tp.y1, run.ws + pl.pre, tb + tp.hc, tb + tp.gates, tb + tp.cst, true, tb + tp.astats, tb + tp.amask};
  if (tp.zn1) { pb.zn1 = tb + tp.zn1; pb.rs1 = tb + tp.rs1; pb.zn2 = tb + tp.zn2; pb.rs2 = tb + tp.rs2; }
  return h->cfg.num_features == 128 ? run_path<128>(h, run, block, path, x_in, x_out, B, S, &pb)
                                    : run_path<64>(h, run, block, path, x_in, x_out, B, S, &pb);
}
int dptnav_train_path_backward(dptnav_handle h, int block, int path, const float* x_in, const float* d_out, float* d_in,
                               int B, int S, void* tape, size_t tape_bytes, void* bws, size_t bws_bytes, void* stream) {
  if (!h) return DPTNAV_ERR_INVALID;
  if (h->gptr.size() != h->names.size()) return h->fail(DPTNAV_ERR_WEIGHTS, "gradients not bound: call dptnav_bind_grads");
  if (block < 0 || block >= h->cfg.num_blocks || (path != 0 && path != 1) || !x_in || !d_out || !d_in || !tape || !bws)
    return h->fail(DPTNAV_ERR_INVALID, "train_path_backward: bad argument");
  if (h->opt_train_fuse_probe)
    return h->fail(DPTNAV_ERR_INVALID, "train_path_backward: option train_fuse_probe is set (measurement only: the forward wrote no tape)");
  PathTape tp;
  make_path_tape(h, B, S, &tp);
  BwdPlan bp;
  make_bwd_plan(h, B, S, &bp);
  if (tape_bytes < tp.total * sizeof(float) || bws_bytes < bp.total * sizeof(float) || ((uintptr_t)bws & 255) != 0)
    return h->fail(DPTNAV_ERR_WORKSPACE, "tape or backward workspace too small / misaligned");
  BwdRun br{(float*)bws, bp, (hipStream_t)stream, 0, h->gptr.data()};
  if (hipMemsetAsync(br.ws + bp.queue, 0, QUEUE_SLOTS * sizeof(unsigned), br.st) != hipSuccess)
    return h->fail(DPTNAV_ERR_HIP, "ticket counter reset");
  return h->cfg.num_features == 128 ? run_path_backward<128>(h, br, block, path, x_in, d_out, d_in, B, S, (float*)tape, tp)
                                    : run_path_backward<64>(h, br, block, path, x_in, d_out, d_in, B, S, (float*)tape, tp);
}

// ---- training step, whole model -------------------------------------------------------------------------
static int train_shapes(dptnav_handle h, int B, int64_t T, int Tv, Plan* pl, ModelTape* mt, BwdPlan* bp) {
  if ((h->cfg.num_features != 128 || h->cfg.arch != 0) && !h->opt_ln_tape)
    return h->fail(DPTNAV_ERR_INVALID, "training step of this configuration needs option ln_tape = 1");
  if (int rc = make_plan(h, B, T, Tv, pl)) return rc;
  make_model_tape(h, B, pl->L, (int)pl->S, Tv, mt);
  make_bwd_plan(h, B, (int)pl->S, bp, pl->L, Tv);
  return DPTNAV_OK;
}

// The training step runs a batch of B >= 2 as TWO halves on the two internal streams, like dptnav_forward (option
// "train_overlap"): the LSTM forward and BPTT launches fill 57 % of the CUs at B = 16, and the other half's GEMM /
// attention / weight-gradient kernels take the rest.  Each half has its own tape slice and workspace slice; the second
// half writes its parameter gradients to scratch buffers in its workspace slice and ONE launch adds them to the bound
// gradients at the end (fixed order).
struct TrainSplit {
  int nhalf;
  int Bh[2];
  Plan pl[2];
  ModelTape mt[2];
  BwdPlan bp[2];
  size_t tape_off[2], ws_off[2];   // floats
  size_t scratch_off;              // floats, inside the workspace (half 1's gradient scratch); 0 if nhalf == 1
  size_t tape_total, ws_total;     // floats
};
static size_t grad_scratch_floats(dptnav_handle h) {
  size_t o = 0;
  for (size_t i = 0; i < h->names.size(); ++i) o += align64((size_t)dptnav_weight_numel(h, (int)i));
  return o;
}
static int train_split(dptnav_handle h, int B, int64_t T, int Tv, TrainSplit* sp) {
  // checked HERE (sizes, forward, backward all come through), not in the last stage of the backward: video_linear_bwd_kernel
  // keeps one (mixture, feature) column of both speakers' frames in LDS
  if (!h->cfg.audio_only && Tv > 256)
    return h->fail(DPTNAV_ERR_INVALID, "training step: at most 256 video frames per mixture (Tv=%d); inference has no such limit", Tv);
  sp->nhalf = (h->opt_overlap && h->opt_train_overlap && B >= 2) ? 2 : 1;
  sp->Bh[0] = sp->nhalf == 2 ? (B + 1) / 2 : B;
  sp->Bh[1] = sp->nhalf == 2 ? B / 2 : 0;
  sp->tape_total = sp->ws_total = 0;
  for (int i = 0; i < sp->nhalf; ++i) {
    if (int rc = train_shapes(h, sp->Bh[i], T, Tv, &sp->pl[i], &sp->mt[i], &sp->bp[i])) return rc;
    sp->tape_off[i] = sp->tape_total;
    sp->tape_total += align64(sp->mt[i].total);
    sp->ws_off[i] = sp->ws_total;
    sp->ws_total += align64(std::max(sp->pl[i].total, sp->bp[i].total));
  }
  sp->scratch_off = 0;
  if (sp->nhalf == 2) {
    sp->scratch_off = sp->ws_total;
    sp->ws_total += grad_scratch_floats(h);
  }
  return DPTNAV_OK;
}
size_t dptnav_train_tape_bytes(dptnav_handle h, int B, int64_t T, int Tv) {
  if (!h) return 0;
  TrainSplit sp;
  if (train_split(h, B, T, Tv, &sp)) return 0;
  return sp.tape_total * sizeof(float);
}
size_t dptnav_train_workspace_bytes(dptnav_handle h, int B, int64_t T, int Tv) {
  if (!h) return 0;
  TrainSplit sp;
  if (train_split(h, B, T, Tv, &sp)) return 0;
  return sp.ws_total * sizeof(float);
}
static int train_fork(dptnav_handle h, const TrainSplit& sp, hipStream_t st, hipStream_t* si) {
  si[0] = si[1] = st;
  if (sp.nhalf == 1) return DPTNAV_OK;
  if (int rc = h->ensure_streams()) return rc;
  if (hipEventRecord(h->ev_fork, st) != hipSuccess) return h->fail(DPTNAV_ERR_HIP, "fork event");
  for (int i = 0; i < 2; ++i) {
    si[i] = h->streams[i];
    if (hipStreamWaitEvent(si[i], h->ev_fork, 0) != hipSuccess) return h->fail(DPTNAV_ERR_HIP, "fork wait");
  }
  return DPTNAV_OK;
}

int dptnav_train_forward(dptnav_handle h, const float* mix, const float* e1, const float* e2, int B, int64_t T, int Tv,
                         float* s1, float* s2, void* tape, size_t tape_bytes, void* ws, size_t ws_bytes, void* stream) {
  if (!h) return DPTNAV_ERR_INVALID;
  if (!h->bound) return h->fail(DPTNAV_ERR_WEIGHTS, "weights not bound: call dptnav_bind_weights first");
  TrainSplit sp;
  if (int rc = train_split(h, B, T, Tv, &sp)) return rc;
  if (!ws || ((uintptr_t)ws & 255) || ws_bytes < sp.ws_total * sizeof(float))
    return h->fail(DPTNAV_ERR_WORKSPACE, "workspace too small / misaligned: %zu < %zu bytes", ws_bytes, sp.ws_total * sizeof(float));
  if (!tape || tape_bytes < sp.tape_total * sizeof(float) || ((uintptr_t)tape & 255)) return h->fail(DPTNAV_ERR_WORKSPACE, "tape too small / misaligned");
  if (!mix || !s1 || !s2 || (!h->cfg.audio_only && (!e1 || !e2 || Tv < 1))) return h->fail(DPTNAV_ERR_INVALID, "null tensor argument");
  hipStream_t st = (hipStream_t)stream, si[2];
  if (int rc = train_fork(h, sp, st, si)) return rc;
  auto enqueue = [&]() -> int {   // between fork and join: see join_after
  const int nb = h->cfg.num_blocks;
  const int64_t Cv = h->cfg.audio_only ? 0 : h->cfg.video_emb_size;
  Run run[2];
  float* tb[2];
  int64_t b0[2] = {0, sp.Bh[0]};
  for (int i = 0; i < sp.nhalf; ++i) {
    if (int rc = begin_run(h, &run[i], (float*)ws + sp.ws_off[i], sp.pl[i], si[i])) return rc;
    run[i].half = i;
    tb[i] = (float*)tape + sp.tape_off[i];
    const ModelTape& mt = sp.mt[i];
    if (int rc = (h->cfg.num_features == 128 ? run_head<128>(h, run[i], mix + b0[i] * T, e1 ? e1 + b0[i] * Cv * Tv : nullptr, e2 ? e2 + b0[i] * Cv * Tv : nullptr,
                               sp.Bh[i], T, Tv, tb[i] + mt.E, tb[i] + mt.X0, tb[i] + mt.vid) : run_head<64>(h, run[i], mix + b0[i] * T, e1 ? e1 + b0[i] * Cv * Tv : nullptr, e2 ? e2 + b0[i] * Cv * Tv : nullptr,
                               sp.Bh[i], T, Tv, tb[i] + mt.E, tb[i] + mt.X0, tb[i] + mt.vid)))
      return rc;
    if (int rc = h->cfg.num_features == 128 ? pack_wih_all<128>(h, run[i]) : pack_wih_all<64>(h, run[i])) return rc;
    if (int rc = pack_inw_all(h, run[i])) return rc;
  }
  // the halves advance in lock step on the host; their recurrences are chained by events as in dptnav_forward
  bool have_prev = false;
  for (int p = 0; p < 2 * nb; ++p)
    for (int i = 0; i < sp.nhalf; ++i) {
      const ModelTape& mt = sp.mt[i];
      if (sp.nhalf == 2) {
        run[i].lstm_wait = have_prev && h->opt_lstm_chain ? h->ev_lstm[1 - i] : nullptr;
        run[i].lstm_record = h->ev_lstm[i];
      }
      float* pt = tb[i] + mt.paths + (size_t)p * mt.path_stride;
      PathBufs pb{pt + mt.pt.qkv, pt + mt.pt.att, pt + mt.pt.y1, run[i].ws + sp.pl[i].pre, pt + mt.pt.hc, pt + mt.pt.gates,
                  pt + mt.pt.cst, true, pt + mt.pt.astats, pt + mt.pt.amask};
      if (mt.pt.zn1) { pb.zn1 = pt + mt.pt.zn1; pb.rs1 = pt + mt.pt.rs1; pb.zn2 = pt + mt.pt.zn2; pb.rs2 = pt + mt.pt.rs2; }
      if (int rc = (h->cfg.num_features == 128 ? run_path<128>(h, run[i], p / 2, p % 2, tb[i] + mt.X0 + (size_t)p * mt.x_stride,
                                 tb[i] + mt.X0 + (size_t)(p + 1) * mt.x_stride, sp.Bh[i], (int)sp.pl[i].S, &pb) : run_path<64>(h, run[i], p / 2, p % 2, tb[i] + mt.X0 + (size_t)p * mt.x_stride,
                                 tb[i] + mt.X0 + (size_t)(p + 1) * mt.x_stride, sp.Bh[i], (int)sp.pl[i].S, &pb)))
        return rc;
      have_prev = true;
    }
  // tail: Z (separation-conv output) is needed again by the backward -> kept on the tape
  for (int i = 0; i < sp.nhalf; ++i) {
    const ModelTape& mt = sp.mt[i];
    if (int rc = (h->cfg.num_features == 128 ? run_tail<128>(h, run[i], tb[i] + mt.X0 + (size_t)(2 * nb) * mt.x_stride, tb[i] + mt.E, sp.Bh[i], T,
                               s1 + b0[i] * T, s2 + b0[i] * T, tb[i] + mt.Z) : run_tail<64>(h, run[i], tb[i] + mt.X0 + (size_t)(2 * nb) * mt.x_stride, tb[i] + mt.E, sp.Bh[i], T,
                               s1 + b0[i] * T, s2 + b0[i] * T, tb[i] + mt.Z)))
      return rc;
  }
  return DPTNAV_OK;
  };
  const int rc_body = enqueue();
  return join_after(h, st, sp.nhalf == 2, rc_body);
}

// grads[i] += scratch[i] for up to GRAD_ADD_MAX parameters per launch (pointers travel as kernel arguments)
constexpr int GRAD_ADD_MAX = 96;
struct GradAddArgs {
  float* dst[GRAD_ADD_MAX];
  const float* src[GRAD_ADD_MAX];
  int n[GRAD_ADD_MAX];
};
__global__ void grad_add_kernel(GradAddArgs a) {
  const int e = blockIdx.x;
  float* d = a.dst[e];
  const float* s = a.src[e];
  for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < a.n[e]; i += gridDim.y * blockDim.x) d[i] += s[i];
}

int dptnav_train_backward(dptnav_handle h, const float* mix, const float* e1, const float* e2, const float* d_s1,
                          const float* d_s2, int B, int64_t T, int Tv, void* tape, size_t tape_bytes, void* ws,
                          size_t ws_bytes, void* stream) {
  if (!h) return DPTNAV_ERR_INVALID;
  TrainSplit sp;
  if (int rc = train_split(h, B, T, Tv, &sp)) return rc;
  if (!h->bound) return h->fail(DPTNAV_ERR_WEIGHTS, "weights not bound");
  if (h->gptr.size() != h->names.size()) return h->fail(DPTNAV_ERR_WEIGHTS, "gradients not bound: call dptnav_bind_grads");
  if (!tape || tape_bytes < sp.tape_total * sizeof(float) || !ws || ws_bytes < sp.ws_total * sizeof(float) || ((uintptr_t)ws & 255))
    return h->fail(DPTNAV_ERR_WORKSPACE, "tape or workspace too small / misaligned");
  if (!mix || !d_s1 || !d_s2) return h->fail(DPTNAV_ERR_INVALID, "null tensor argument");
  if (h->opt_train_fuse_probe)
    return h->fail(DPTNAV_ERR_INVALID, "train_backward: option train_fuse_probe is set (measurement only: the forward wrote no tape)");
  hipStream_t st = (hipStream_t)stream, si[2];
  if (int rc = train_fork(h, sp, st, si)) return rc;
  auto enqueue = [&]() -> int {   // between fork and join: see join_after
  const int nb = h->cfg.num_blocks;
  const int64_t Cv = h->cfg.audio_only ? 0 : h->cfg.video_emb_size;
  const int64_t b0[2] = {0, sp.Bh[0]};
  if (sp.nhalf == 2) {   // half 1 writes its parameter gradients to scratch
    h->gptr_half1.resize(h->names.size());
    size_t o = sp.scratch_off;
    for (size_t i = 0; i < h->names.size(); ++i) {
      h->gptr_half1[i] = (float*)ws + o;
      o += align64((size_t)dptnav_weight_numel(h, (int)i));
    }
  }
  BwdRun br[2];
  Run run[2];
  float *tb[2], *dcur[2], *dnext[2];
  for (int i = 0; i < sp.nhalf; ++i) {
    const BwdPlan& bp = sp.bp[i];
    br[i] = BwdRun{(float*)ws + sp.ws_off[i], bp, si[i], 0, i == 0 ? h->gptr.data() : h->gptr_half1.data()};
    br[i].half = i;
    if (hipMemsetAsync(br[i].ws + bp.queue, 0, QUEUE_SLOTS * sizeof(unsigned), si[i]) != hipSuccess)
      return h->fail(DPTNAV_ERR_HIP, "ticket counter reset");
    if (sp.nhalf == 2 && h->opt_wgrad_side) {
      if (hipMemsetAsync(br[i].ws + bp.queue2, 0, QUEUE_SLOTS * sizeof(unsigned), si[i]) != hipSuccess)
        return h->fail(DPTNAV_ERR_HIP, "ticket counter reset");
      br[i].side = h->streams[2 + i];
      br[i].ev_bptt = h->ev_side[i][0];
      br[i].ev_wg[0] = h->ev_side[i][1];
      br[i].ev_wg[1] = h->ev_side[i][2];
      br[i].ev_wg[2] = h->ev_side[i][3];
    }
    for (int first = 0; first < 2 * nb; first += DGRAD_R_PACK_MAX / 2) {      // weight copies of every path, once per step and half
      const int n = std::min(DGRAD_R_PACK_MAX / 2, 2 * nb - first);
      if (int rc = pack_bwd_weights(h, si[i], first, n, br[i].ws + bp.wiht + (size_t)first * BWD_PACK_FLOATS)) return rc;
    }
    br[i].packed_all = true;
    run[i].ws = br[i].ws;
    run[i].pl = Plan{};
    run[i].pl.queue = bp.queue;
    run[i].st = si[i];
    run[i].slot = 0;
    run[i].half = i;
    tb[i] = (float*)tape + sp.tape_off[i];
    dcur[i] = br[i].ws + bp.dxa;
    dnext[i] = br[i].ws + bp.dxb;
    const ModelTape& mt = sp.mt[i];
    if (int rc = (h->cfg.num_features == 128 ? run_tail_backward<128>(h, br[i], run[i], tb[i] + mt.X0 + (size_t)(2 * nb) * mt.x_stride, tb[i] + mt.E, tb[i] + mt.Z,
                                        d_s1 + b0[i] * T, d_s2 + b0[i] * T, dcur[i], sp.Bh[i], T, sp.pl[i].L, (int)sp.pl[i].S) : run_tail_backward<64>(h, br[i], run[i], tb[i] + mt.X0 + (size_t)(2 * nb) * mt.x_stride, tb[i] + mt.E, tb[i] + mt.Z,
                                        d_s1 + b0[i] * T, d_s2 + b0[i] * T, dcur[i], sp.Bh[i], T, sp.pl[i].L, (int)sp.pl[i].S)))
      return rc;
  }
  bool have_prev = false;
  for (int p = 2 * nb - 1; p >= 0; --p)
    for (int i = 0; i < sp.nhalf; ++i) {
      const ModelTape& mt = sp.mt[i];
      if (br[i].slot > QUEUE_SLOTS - 64) {   // plenty of launches per path: recycle the ticket counters
        if (hipMemsetAsync(br[i].ws + sp.bp[i].queue, 0, QUEUE_SLOTS * sizeof(unsigned), si[i]) != hipSuccess)
          return h->fail(DPTNAV_ERR_HIP, "ticket counter reset");
        br[i].slot = 0;
      }
      if (sp.nhalf == 2) {
        br[i].lstm_wait = have_prev && h->opt_lstm_chain ? h->ev_lstm[1 - i] : nullptr;
        br[i].lstm_record = h->ev_lstm[i];
      }
      float* pt = tb[i] + mt.paths + (size_t)p * mt.path_stride;
      if (int rc = (h->cfg.num_features == 128 ? run_path_backward<128>(h, br[i], p / 2, p % 2, tb[i] + mt.X0 + (size_t)p * mt.x_stride, dcur[i], dnext[i],
                                          sp.Bh[i], (int)sp.pl[i].S, pt, mt.pt) : run_path_backward<64>(h, br[i], p / 2, p % 2, tb[i] + mt.X0 + (size_t)p * mt.x_stride, dcur[i], dnext[i],
                                          sp.Bh[i], (int)sp.pl[i].S, pt, mt.pt)))
        return rc;
      std::swap(dcur[i], dnext[i]);
      have_prev = true;
    }
  for (int i = 0; i < sp.nhalf; ++i)
    if (int rc = (h->cfg.num_features == 128 ? run_head_backward<128>(h, br[i], mix + b0[i] * T, e1 ? e1 + b0[i] * Cv * Tv : nullptr,
                                        e2 ? e2 + b0[i] * Cv * Tv : nullptr, tb[i] + sp.mt[i].vid, dcur[i], sp.Bh[i], T,
                                        sp.pl[i].L, (int)sp.pl[i].S, Tv) : run_head_backward<64>(h, br[i], mix + b0[i] * T, e1 ? e1 + b0[i] * Cv * Tv : nullptr,
                                        e2 ? e2 + b0[i] * Cv * Tv : nullptr, tb[i] + sp.mt[i].vid, dcur[i], sp.Bh[i], T,
                                        sp.pl[i].L, (int)sp.pl[i].S, Tv)))
      return rc;
  return DPTNAV_OK;
  };
  // (all four internal streams are joined: the side streams carry the LSTM weight gradients, also after an error)
  if (int rc = join_after(h, st, sp.nhalf == 2, enqueue(), sp.nhalf == 2 && h->opt_wgrad_side ? 4 : 2)) return rc;
  if (sp.nhalf == 2) {
    const int n = (int)h->names.size();
    for (int lo = 0; lo < n; lo += GRAD_ADD_MAX) {
      GradAddArgs a{};
      const int cnt = std::min(GRAD_ADD_MAX, n - lo);
      for (int e = 0; e < cnt; ++e) {
        a.dst[e] = h->gptr[lo + e];
        a.src[e] = h->gptr_half1[lo + e];
        a.n[e] = (int)dptnav_weight_numel(h, lo + e);
      }
      hipLaunchKernelGGL(grad_add_kernel, dim3(cnt, 8), dim3(256), 0, st, a);
      LAUNCH_CHECK(h, "gradient add");
    }
  }
  return DPTNAV_OK;
}

// test helper: the keep-mask (1/0) the attention kernels of (block, path) use under the current dropout options
int dptnav_dropout_mask(dptnav_handle h, int block, int path, int B, int S, float* mask, void* stream) {
  if (!h || !mask) return DPTNAV_ERR_INVALID;
  const SeqGeom geom = make_geom(path, B, S, h->cfg.chunk_size);
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(geom.nseq, h->cfg.num_heads), dim3(256), 0, (hipStream_t)stream, mask, geom,
                     h->cfg.num_heads, h->drop_cfg(block, path, true));
  LAUNCH_CHECK(h, "dropout mask");
  return DPTNAV_OK;
}

// ---- tuning / diagnostic knobs ----------------------------------------------------------------------
int dptnav_set_option(dptnav_handle h, const char* key, int value) {
  if (!h || !key) return DPTNAV_ERR_INVALID;
  const std::string k(key);
  if (k == "lstm_stamps") h->opt_lstm_stamps = value != 0;
  else if (k == "dropout_ppm" && value >= 0 && value < 1000000) h->opt_dropout_ppm = value;
  else if (k == "dropout_seed") h->opt_dropout_seed = (unsigned)value;
  else if (k == "overlap") h->opt_overlap = value != 0;
  else if (k == "serialize") h->opt_serialize = value != 0;
  else if (k == "deterministic") h->opt_deterministic = value != 0;
  else if (k == "pack_wih") h->opt_pack_wih = value != 0;
  else if (k == "fuse_pre") h->opt_fuse_pre = value != 0;
  else if (k == "fcln" && value >= 0 && value <= 2) h->opt_fcln = value;
  else if (k == "fuse_pre128" && value >= 0 && value <= 2) h->opt_fuse_pre128 = value;
  else if (k == "pack_whh") h->opt_pack_whh = value != 0;
  else if (k == "lstm16") h->opt_lstm16 = value != 0;
  else if (k == "lstm4") {
    if (value < 0 || value > 2) return h->fail(DPTNAV_ERR_INVALID, "lstm4: 0, 1 or 2");
    h->opt_lstm4 = (int)value;
  }
  else if (k == "fuse_attn") h->opt_fuse_attn = value != 0;
  else if (k == "attn_v2") h->opt_attn_v2 = value != 0;
  else if (k == "dgrad_t") h->opt_dgrad_t = value != 0;
  else if (k == "dgrad_r") h->opt_dgrad_r = value != 0;
  else if (k == "gemm_t") h->opt_gemm_t = value != 0;
  else if (k == "fuse_ffn") h->opt_fuse_ffn = value != 0;
  else if (k == "wgrad2") h->opt_wgrad2 = value != 0;
  else if (k == "ln_tape") h->opt_ln_tape = value != 0;
  else if (k == "fold_tail") h->opt_fold_tail = value != 0;
  else if (k == "wgrad_ride") h->opt_wgrad_ride = value != 0;
  else if (k == "wgrad_side") h->opt_wgrad_side = value != 0;
  else if (k == "sub_batches" && value >= 0 && value <= MAX_SUB) h->opt_sub_batches = value;
  else if (k == "lstm_inflight" && value >= 0 && value <= MAX_SUB) h->opt_lstm_inflight = value;
  else if (k == "split_policy" && (value == 0 || value == 1)) h->opt_split_policy = value;
  else if (k == "lstm_chain") h->opt_lstm_chain = value != 0;
  else if (k == "split_bf16") h->opt_split_bf16 = value != 0;
  else if (k == "train_overlap") h->opt_train_overlap = value != 0;
  else if (k == "lstm_diag") h->opt_lstm_diag = (int)value;
  else if (k == "inject_fail") h->opt_inject_fail = value;
  else if (k == "debug_sync") h->opt_debug_sync = value != 0;
  else if (k == "train_fuse_probe") h->opt_train_fuse_probe = value != 0;
  else return h->fail(DPTNAV_ERR_INVALID, "unknown option '%s'", key);
  return DPTNAV_OK;
}

// ---- opt-in per-kernel timing (HIP events on the launch stream) --------------------------------
int dptnav_profile_enable(dptnav_handle h, int on) {
  if (!h) return DPTNAV_ERR_INVALID;
  h->prof_on = on != 0;
  return DPTNAV_OK;
}
int dptnav_profile_collect(dptnav_handle h) {
  if (!h) return DPTNAV_ERR_INVALID;
  for (ProfRec& r : h->prof_pending) {
    float ms = 0.f;
    hipError_t e = hipEventSynchronize(r.b);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, r.a, r.b);
    if (e != hipSuccess) return h->fail(DPTNAV_ERR_HIP, "profile: %s", hipGetErrorString(e));
    h->prof_ms[r.cat] += ms;
    h->prof_n[r.cat] += 1;
    h->prof_pool.push_back(r.a);
    h->prof_pool.push_back(r.b);
  }
  h->prof_pending.clear();
  return DPTNAV_OK;
}
int dptnav_profile_reset(dptnav_handle h) {
  if (!h) return DPTNAV_ERR_INVALID;
  if (int rc = dptnav_profile_collect(h)) return rc;
  for (int i = 0; i < CAT_COUNT; ++i) { h->prof_ms[i] = 0; h->prof_n[i] = 0; }
  return DPTNAV_OK;
}
int dptnav_profile_num(void) { return CAT_COUNT; }
const char* dptnav_profile_name(int i) { return (i >= 0 && i < CAT_COUNT) ? kProfNames[i] : nullptr; }
double dptnav_profile_ms(dptnav_handle h, int i) { return (h && i >= 0 && i < CAT_COUNT) ? h->prof_ms[i] : -1.0; }
int64_t dptnav_profile_count(dptnav_handle h, int i) { return (h && i >= 0 && i < CAT_COUNT) ? h->prof_n[i] : -1; }

// Algorithmic cost model (2 FLOP per MAC, GEMM-like terms only) -- SURVEY.md section 8d restated.
double dptnav_flops_per_mixture(dptnav_handle h, int64_t T) {
  if (!h) return 0;
  const dptnav_config& g = h->cfg;
  const double N = g.num_features, H = g.hidden_dim, K = g.chunk_size;
  const double L = (double)dptnav_frames(h, T), S = (double)dptnav_chunks(h, T), M = S * K;
  auto path = [&](double len, double ndir) {
    const double rnn = ndir * (2 * N * 4 * H + 2 * H * 4 * H) + 2 * ndir * H * N;
    return g.arch == 0 ? 2 * N * 3 * N + 2 * N * N + 4 * len * N + rnn : rnn;
  };
  double f = g.num_blocks * M * (path(K, 2) + path(S, g.bidir ? 2 : 1));
  f += 2 * N * 2 * N * M;                 // separation conv
  f += 2 * 2 * N * N * L;                 // post-processing conv, both speakers
  f += 2 * N * g.kernel_size_enc * L;     // encoder
  f += 2 * 2 * N * g.kernel_size_enc * L; // decoder
  if (!g.audio_only) f += 2.0 * 2 * 50 * g.video_emb_size * (g.hidden_video / 2);
  return f;
}

// Ideal-fusion HBM traffic: every TransformerDPRNN reads x once and writes y once.
double dptnav_min_bytes_per_mixture(dptnav_handle h, int64_t T) {
  if (!h) return 0;
  const dptnav_config& g = h->cfg;
  const double N = g.num_features, K = g.chunk_size;
  const double L = (double)dptnav_frames(h, T), S = (double)dptnav_chunks(h, T), M = S * K;
  const double A = M * N * 4, F = L * N * 4;
  return g.num_blocks * 2 * 2 * A + (A + 2 * 2 * A) + 2 * 2 * F + 4 * F + F + 3.0 * T * 4;
}

}  // extern "C"
