"""CPU ORACLE (test infrastructure, NOT product code) -- numpy restatement of the
reference's DPTN(-AV) raw-waveform forward, its loss and its SI-SNRi metric.

  * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
    import this file.  The product path (speech_separation_amd/) never does and
    raises if the HIP library is missing.
  * Parity status: PINNED.  tests/test_oracle_golden.py checks every function
    below against tensors captured from the reference itself (imported and run
    on CPU by tools/gen_golden.py; fixtures under tests/golden/).
  * One exception, stated where it occurs: the SI-SNRi metric wraps the
    third-party ``torchmetrics.audio.ScaleInvariantSignalNoiseRatio``
    (requirements.txt:3, unpinned, not installed here) -> that single function
    is "parity unpinned"; it is restated from its published definition and
    cross-checked against the reference's own SiSNRLoss (ss_losses.py:96-114).

Every function cites the reference lines (relative to /root/reference) it
follows.  Written as explicit formulas (no torch, no fused library ops) so it is
independent of the ATen kernels the reference dispatches to.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy as np

LN_EPS = 1e-5  # torch.nn.LayerNorm default (dptn.py:22,34; dptn_wav.py:156)


# --------------------------------------------------------------------------
# elementary pieces
# --------------------------------------------------------------------------
def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def layer_norm(x: np.ndarray, weight: np.ndarray, bias: np.ndarray) -> np.ndarray:
    """nn.LayerNorm over the last dim, biased variance, eps inside the sqrt."""
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return (x - mu) / np.sqrt(var + LN_EPS) * weight + bias


def encoder_conv(mix: np.ndarray, w: np.ndarray, stride: int) -> np.ndarray:
    """A2: nn.Conv1d(1, N, k, stride=k//2, bias=False)  (dptn_wav.py:153,180).

    mix (B,T), w (N,1,k) -> (B,N,L):  enc[b,c,l] = sum_j w[c,0,j] * mix[b, stride*l + j]
    """
    B, T = mix.shape
    N, _, k = w.shape
    L = (T - k) // stride + 1
    idx = stride * np.arange(L)[:, None] + np.arange(k)[None, :]          # (L,k)
    frames = mix[:, idx]                                                   # (B,L,k)
    return np.einsum("blj,cj->bcl", frames, w[:, 0, :]).astype(mix.dtype)


def interp_linear(v: np.ndarray, L: int) -> np.ndarray:
    """F.interpolate(mode='linear', align_corners=False) along the last dim
    (dptn_wav.py:181-183).  v (B,C,Tv) -> (B,C,L).

    src = max((l+0.5)*Tv/L - 0.5, 0); i0 = floor(src); i1 = min(i0+1, Tv-1); lam = src - i0
    """
    Tv = v.shape[-1]
    scale = np.asarray(Tv / L, dtype=v.dtype)
    src = np.maximum((np.arange(L, dtype=v.dtype) + np.asarray(0.5, v.dtype)) * scale - np.asarray(0.5, v.dtype), 0)
    i0 = np.floor(src).astype(np.int64)
    i1 = np.minimum(i0 + 1, Tv - 1)
    lam = (src - i0).astype(v.dtype)
    return v[..., i0] * (1 - lam) + v[..., i1] * lam


def video_fusion(enc: np.ndarray, e1: np.ndarray, e2: np.ndarray, p: Dict[str, np.ndarray]) -> np.ndarray:
    """A3: dptn_wav.py:173-184.  Linear(512->64) per speaker (shared weights), concat,
    linear interpolation Tv->L, LayerNorm over the 128 features, * tanh(gate), added to enc.
    enc (B,N,L), e* (B,Cv,Tv) -> fused (B,N,L)
    """
    W, b = p["visual_compression.weight"], p["visual_compression.bias"]
    v1 = np.einsum("bct,oc->bto", e1, W) + b                               # (B,Tv,64)
    v2 = np.einsum("bct,oc->bto", e2, W) + b
    video = np.concatenate([v1, v2], -1)                                   # (B,Tv,128)
    video = interp_linear(video.transpose(0, 2, 1), enc.shape[-1]).transpose(0, 2, 1)   # (B,L,128)
    vn = layer_norm(video, p["video_ln.weight"], p["video_ln.bias"])
    return enc + np.tanh(p["gate"]) * vn.transpose(0, 2, 1)


def split_to_folds(x: np.ndarray, K: int, P: int) -> np.ndarray:
    """A4: SplitToFolds (dprnn.py:122-136).  (B,N,L) -> (B,N,S,K), x[b,n,s,k] = in[b,n,P*s+k];
    trailing frames that do not fill a chunk are dropped (F.unfold semantics)."""
    L = x.shape[-1]
    S = (L - K) // P + 1
    idx = P * np.arange(S)[:, None] + np.arange(K)[None, :]
    return x[:, :, idx]


def overlap_add(x: np.ndarray, P: int) -> np.ndarray:
    """A8: OverlapAdd (dprnn.py:145-163, F.fold): plain sum, no window/normalisation.
    (B,C,S,K) -> (B,C,(S-1)*P+K)"""
    B, C, S, K = x.shape
    out = np.zeros((B, C, (S - 1) * P + K), dtype=x.dtype)
    for s in range(S):
        out[:, :, s * P:s * P + K] += x[:, :, s, :]
    return out


def multi_head_attention(x: np.ndarray, p: Dict[str, np.ndarray], pre: str, heads: int) -> np.ndarray:
    """A6: nn.MultiheadAttention(N, heads, batch_first=True), self-attention, eval mode
    (dptn.py:16-21,46).  x (R,T,N) -> (R,T,N) (before the residual).
    in_proj rows [0:N]=Q, [N:2N]=K, [2N:3N]=V; head h = features [h*dh,(h+1)*dh);
    softmax(Q K^T / sqrt(dh)) V; out_proj.  No mask, no positional encoding."""
    R, T, N = x.shape
    dh = N // heads
    qkv = x @ p[pre + "mha.in_proj_weight"].T + p[pre + "mha.in_proj_bias"]
    q, k, v = (qkv[..., i * N:(i + 1) * N].reshape(R, T, heads, dh).transpose(0, 2, 1, 3) for i in range(3))
    s = (q @ k.transpose(0, 1, 3, 2)) / np.sqrt(np.asarray(dh, dtype=x.dtype))
    s = s - s.max(-1, keepdims=True)
    e = np.exp(s)
    a = e / e.sum(-1, keepdims=True)
    o = (a @ v).transpose(0, 2, 1, 3).reshape(R, T, N)
    return o @ p[pre + "mha.out_proj.weight"].T + p[pre + "mha.out_proj.bias"]


def lstm_direction(x: np.ndarray, w_ih, w_hh, b_ih, b_hh, reverse: bool) -> np.ndarray:
    """A7: one direction of nn.LSTM(batch_first=True), zero initial state (dptn.py:23-29,49).
    gates = x_t W_ih^T + b_ih + h W_hh^T + b_hh, row blocks in order i | f | g | o;
    c = sig(f) c + sig(i) tanh(g); h = sig(o) tanh(c).  reverse runs t = T-1..0 and the
    output at position t is the state after consuming x_t."""
    R, T, _ = x.shape
    H = w_hh.shape[1]
    h = np.zeros((R, H), dtype=x.dtype)
    c = np.zeros((R, H), dtype=x.dtype)
    out = np.zeros((R, T, H), dtype=x.dtype)
    pre = x @ w_ih.T + (b_ih + b_hh)
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        g = pre[:, t] + h @ w_hh.T
        i, f, gg, o = g[:, :H], g[:, H:2 * H], g[:, 2 * H:3 * H], g[:, 3 * H:]
        c = _sigmoid(f) * c + _sigmoid(i) * np.tanh(gg)
        h = _sigmoid(o) * np.tanh(c)
        out[:, t] = h
    return out


def transformer_dprnn(x: np.ndarray, p: Dict[str, np.ndarray], pre: str, heads: int) -> np.ndarray:
    """A6+A7: TransformerDPRNN.forward (dptn.py:36-52): MHA + res -> LN1 -> (bi)LSTM -> ReLU ->
    Linear -> + res -> LN2.  nn.LSTM(dropout=1) with one layer is a no-op."""
    y = multi_head_attention(x, p, pre, heads) + x
    y = layer_norm(y, p[pre + "ln1.weight"], p[pre + "ln1.bias"])
    hs = [lstm_direction(y, p[pre + "rnn.weight_ih_l0"], p[pre + "rnn.weight_hh_l0"],
                         p[pre + "rnn.bias_ih_l0"], p[pre + "rnn.bias_hh_l0"], False)]
    if pre + "rnn.weight_ih_l0_reverse" in p:
        hs.append(lstm_direction(y, p[pre + "rnn.weight_ih_l0_reverse"], p[pre + "rnn.weight_hh_l0_reverse"],
                                 p[pre + "rnn.bias_ih_l0_reverse"], p[pre + "rnn.bias_hh_l0_reverse"], True))
    r = np.concatenate(hs, -1)
    z = np.maximum(r, 0) @ p[pre + "ffn.1.weight"].T + p[pre + "ffn.1.bias"] + y
    return layer_norm(z, p[pre + "ln2.weight"], p[pre + "ln2.bias"])


def dptn_block(x: np.ndarray, p: Dict[str, np.ndarray], b: int, heads: int,
               taps: Optional[dict] = None) -> np.ndarray:
    """A5: DPTNBlock.forward (dptn.py:62-79).  x (B,N,S,K) -> (B,N,S,K)."""
    B, N, S, K = x.shape
    pre = f"dprnn.model.{b}."
    intra_in = x.transpose(0, 2, 3, 1).reshape(B * S, K, N)               # b n s k -> (b s) k n
    intra = transformer_dprnn(intra_in, p, pre + "intra_chunk_block.", heads)
    inter_in = intra.reshape(B, S, K, N).transpose(0, 2, 1, 3).reshape(B * K, S, N)   # -> (b k) s n
    inter = transformer_dprnn(inter_in, p, pre + "inter_chunk_block.", heads)
    out = inter.reshape(B, K, S, N).transpose(0, 3, 2, 1)                  # -> b n s k
    if taps is not None:
        taps[f"blk{b}_intra"], taps[f"blk{b}_inter"], taps[f"blk{b}_out"] = intra, inter, out
    return out


def dprnn_path(x: np.ndarray, p: Dict[str, np.ndarray], pre: str) -> np.ndarray:
    """IntraChunkRNN / InterChunkRNN on sequence-major input (dprnn.py:24-47, 65-89): (bi)LSTM -> Linear ->
    LayerNorm over the features -> + residual.  NB the norm comes BEFORE the residual add and there is no ReLU."""
    hs = [lstm_direction(x, p[pre + "rnn.weight_ih_l0"], p[pre + "rnn.weight_hh_l0"],
                         p[pre + "rnn.bias_ih_l0"], p[pre + "rnn.bias_hh_l0"], False)]
    if pre + "rnn.weight_ih_l0_reverse" in p:
        hs.append(lstm_direction(x, p[pre + "rnn.weight_ih_l0_reverse"], p[pre + "rnn.weight_hh_l0_reverse"],
                                 p[pre + "rnn.bias_ih_l0_reverse"], p[pre + "rnn.bias_hh_l0_reverse"], True))
    y = np.concatenate(hs, -1) @ p[pre + "fc.weight"].T + p[pre + "fc.bias"]
    return layer_norm(y, p[pre + "norm1d.weight"], p[pre + "norm1d.bias"]) + x


def dprnn_block(x: np.ndarray, p: Dict[str, np.ndarray], b: int, taps: Optional[dict] = None) -> np.ndarray:
    """DPRNNBlock.forward (dprnn.py:103-113).  x (B,N,S,K) -> (B,N,S,K); taps are in that layout too."""
    B, N, S, K = x.shape
    pre = f"dprnn.model.{b}."
    intra = dprnn_path(x.transpose(0, 2, 3, 1).reshape(B * S, K, N), p, pre + "intra_chunk_block.")
    intra = intra.reshape(B, S, K, N).transpose(0, 3, 1, 2)                                 # b n s k
    inter = dprnn_path(intra.transpose(0, 3, 2, 1).reshape(B * K, S, N), p, pre + "inter_chunk_block.")
    out = inter.reshape(B, K, S, N).transpose(0, 3, 2, 1)                                   # b n s k
    if taps is not None:
        taps[f"blk{b}_intra"], taps[f"blk{b}_inter"], taps[f"blk{b}_out"] = intra, out, out
    return out


def separation_tail(x: np.ndarray, L: int, p: Dict[str, np.ndarray], P: int,
                    taps: Optional[dict] = None) -> np.ndarray:
    """A8: DPTNWav.forward tail (dptn_wav.py:47-61).  PReLU (one slope) -> Conv2d 1x1 (N->2N) ->
    OverlapAdd -> zero-pad (left=(L-ola)//2, right=rest) -> view(B,2,N,L).transpose(0,1) ->
    shared Conv1d 1x1 (N->N) per speaker.  Returns (2,B,N,L)."""
    B, N, S, K = x.shape
    a = p["dprnn.speakers_separation.0.weight"]
    y = np.where(x >= 0, x, a * x)
    W = p["dprnn.speakers_separation.1.weight"][:, :, 0, 0]
    sep = np.einsum("bnsk,on->bosk", y, W) + p["dprnn.speakers_separation.1.bias"][None, :, None, None]
    ola = overlap_add(sep, P)
    pad = L - ola.shape[-1]
    left = pad // 2
    padded = np.zeros((B, 2 * N, L), dtype=x.dtype)
    padded[:, :, left:left + ola.shape[-1]] = ola
    spk = padded.reshape(B, 2, N, L).transpose(1, 0, 2, 3)
    Wp = p["dprnn.postprocessing.0.weight"][:, :, 0]
    masks = np.einsum("jbnl,on->jbol", spk, Wp) + p["dprnn.postprocessing.0.bias"][None, None, :, None]
    if taps is not None:
        taps["sep"], taps["ola"], taps["masks"] = sep, ola, masks
    return masks


def decoder_deconv(x: np.ndarray, w: np.ndarray, stride: int, T: int) -> np.ndarray:
    """A9: nn.ConvTranspose1d(N,1,k,stride=k//2,bias=False) then right/left zero pad to T
    (dptn_wav.py:167-169,186-193).  x (B,N,L), w (N,1,k) -> (B,T):
    y[b, stride*i + j] += sum_c x[b,c,i] w[c,0,j]; pad_left = (T-len)//2."""
    B, N, L = x.shape
    k = w.shape[-1]
    n = (L - 1) * stride + k
    taps_ = np.einsum("bcl,cj->blj", x, w[:, 0, :])                        # (B,L,k)
    y = np.zeros((B, n), dtype=x.dtype)
    for j in range(k):
        y[:, j:j + stride * L:stride] += taps_[:, :, j]
    pad = T - n
    out = np.zeros((B, T), dtype=x.dtype)
    out[:, pad // 2:pad // 2 + n] = y
    return out


# --------------------------------------------------------------------------
# whole path
# --------------------------------------------------------------------------
def forward(cfg, params: Dict[str, np.ndarray], mix: np.ndarray, s1_embedding: Optional[np.ndarray] = None,
            s2_embedding: Optional[np.ndarray] = None, dtype=np.float32, taps: Optional[dict] = None,
            **_ignored) -> Dict[str, np.ndarray]:
    """DPTNAVWavEncDec.forward (dptn_wav.py:171-194) / DPTNWavEncDec.forward (:105-117) / DPRNNEncDec.forward
    (dprnn.py:260-274; arch == "dprnn").  The head and the tail are the same code in all three (DPRNN.forward
    dprnn.py:200-227 == DPTNWav.forward dptn_wav.py:35-61).  arch == "dprnn" with audio_only == False is this
    repo's "DPRNN-AV" (BASELINE config 5): the reference has no such class, so only its parts are pinned.
    Extra batch keys are swallowed like the reference's **batch."""
    p = {k: np.asarray(v, dtype=dtype) for k, v in params.items()}
    mix = np.asarray(mix, dtype=dtype)
    B, T = mix.shape
    enc = encoder_conv(mix, p["encoder.weight"], cfg.stride_enc)
    if taps is not None:
        taps["enc_conv"] = enc
    if not cfg.audio_only:
        enc = video_fusion(enc, np.asarray(s1_embedding, dtype), np.asarray(s2_embedding, dtype), p)
    L = enc.shape[-1]
    x = split_to_folds(enc, cfg.chunk_size, cfg.step_size)
    if taps is not None:
        taps["encoded"], taps["chunked"] = enc, x
    for b in range(cfg.num_blocks):
        x = dptn_block(x, p, b, cfg.num_heads, taps) if cfg.arch == "dptn" else dprnn_block(x, p, b, taps)
    masks = separation_tail(x, L, p, cfg.step_size, taps)
    preds = [decoder_deconv(masks[j] + enc, p["decoder.weight"], cfg.stride_enc, T) for j in range(2)]
    return {"s1_pred": preds[0], "s2_pred": preds[1]}


# --------------------------------------------------------------------------
# loss (A10) and metric (A11)
# --------------------------------------------------------------------------
def si_snr_loss(pred: np.ndarray, gt: np.ndarray) -> float:
    """SiSNRLoss.forward (ss_losses.py:100-114): zero-mean both, project, -20*log10(ratio) mean
    over the batch (note 20, not 10, and no eps)."""
    pred = pred - pred.mean(-1, keepdims=True)
    gt = gt - gt.mean(-1, keepdims=True)
    scale = (gt * pred).sum(-1, keepdims=True) / (gt ** 2).sum(-1, keepdims=True)
    sg = scale * gt
    return float((-20.0 * np.log10((sg ** 2).sum(-1) / ((pred - sg) ** 2).sum(-1))).mean())


def pit_loss(s1_pred, s2_pred, s1, s2) -> float:
    """BaseSSLoss.forward (ss_losses.py:21-26): BATCH-level PIT (compares batch means)."""
    p1 = (si_snr_loss(s1_pred, s1) + si_snr_loss(s2_pred, s2)) / 2
    p2 = (si_snr_loss(s1_pred, s2) + si_snr_loss(s2_pred, s1)) / 2
    return p2 if p2 < p1 else p1


def si_snr_db(pred: np.ndarray, target: np.ndarray) -> float:
    """torchmetrics ScaleInvariantSignalNoiseRatio restated (PARITY UNPINNED, see header):
    zero-mean, alpha = (<p,t>+eps)/(<t,t>+eps), 10*log10((|alpha t|^2+eps)/(|p-alpha t|^2+eps)),
    mean over the batch.  eps = finfo(dtype).eps."""
    eps = np.finfo(pred.dtype).eps
    pred = pred - pred.mean(-1, keepdims=True)
    target = target - target.mean(-1, keepdims=True)
    alpha = ((pred * target).sum(-1, keepdims=True) + eps) / ((target ** 2).sum(-1, keepdims=True) + eps)
    ts = alpha * target
    val = ((ts ** 2).sum(-1) + eps) / (((pred - ts) ** 2).sum(-1) + eps)
    return float((10.0 * np.log10(val)).mean())


def si_snri_metric(s1_pred, s2_pred, s1, s2, mix) -> float:
    """SISNRiMetric.__call__ (si_snri.py:12-30) + SS2BaseMetric.forward (base_metric.py:41-60):
    batch-level PIT by max of the two permutation means, minus the mixture's mean SI-SNR."""
    perm1 = (si_snr_db(s1_pred, s1) + si_snr_db(s2_pred, s2)) / 2
    perm2 = (si_snr_db(s1_pred, s2) + si_snr_db(s2_pred, s1)) / 2
    base = (si_snr_db(mix, s1) + si_snr_db(mix, s2)) / 2
    return max(perm1, perm2) - base


def agreement_db(a: np.ndarray, b: np.ndarray) -> float:
    """10*log10(|b|^2/|a-b|^2): how many dB below the signal the difference sits."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    d = ((a - b) ** 2).sum()
    return float("inf") if d == 0 else float(10 * np.log10((b ** 2).sum() / d))
