"""numpy restatement of the library's counter-based attention-dropout keep-mask (csrc/common.h: mix32, drop_qseed,
drop_rand_q; csrc/dptnav.hip: drop_cfg; csrc/backward.h: dropout_mask_kernel).  Test infrastructure: the GPU mask must
equal this bit for bit, and its statistics are checked on the CPU.  (The reference uses PyTorch's Philox stream --
nn.MultiheadAttention(dropout=0.1), src/model/dptn.py:16-21 -- which a fused kernel cannot reproduce; the contract is
the same distribution, regenerated identically in forward and backward.)"""
import numpy as np

M32 = np.uint64(0xFFFFFFFF)
M24 = np.uint64(0xFFFFFF)


def mix32(x):
    x = x.astype(np.uint64) & M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & M32
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & M32
    x ^= x >> np.uint64(16)
    return x


def umul24(a, b):
    return ((a & M24) * (np.uint64(b) & M24)) & M32


def drop_qseed(seed, qtok_head):
    return mix32(np.uint64(seed) ^ ((qtok_head.astype(np.uint64) * np.uint64(0x9E3779B9)) & M32))


def drop_rand_q(qseed, key):
    x = (qseed + umul24(key.astype(np.uint64), 0x9E3779)) & M32
    x ^= x >> np.uint64(15)
    x = umul24(x, 0xB5297B) >> np.uint64(6)
    x ^= x >> np.uint64(11)
    x = umul24(x, 0x68E31D) >> np.uint64(8)
    return x & M24


def keep_mask(block, path, B, S, K, heads, ppm, seed):
    """(nseq, heads, len, len) float mask of path (block, path): 1 = keep."""
    nseq, ln = (B * S, K) if path == 0 else (B * K, S)
    if ppm <= 0:
        return np.ones((nseq, heads, ln, ln), np.float32)
    p = ppm * 1e-6
    cseed = (int(seed) ^ ((0x9E3779B9 * (2 * block + path + 1)) & 0xFFFFFFFF)) & 0xFFFFFFFF
    thresh = int(p * 16777216.0)
    q = np.arange(nseq, dtype=np.int64)
    if path == 0:      # token of (sequence, position): q*K + t
        tok0, tstride = q * K, 1
    else:              # sequence q = b*K + k over chunks: (b*S + t)*K + k
        tok0, tstride = (q // K) * S * K + (q % K), K
    tok = tok0[:, None] + np.arange(ln, dtype=np.int64)[None, :] * tstride                 # (nseq, len_q)
    qh = ((tok[:, None, :] & 0xFFFFFFFF) * heads + np.arange(heads)[None, :, None]) & 0xFFFFFFFF   # (nseq, heads, len_q)
    qs = drop_qseed(cseed, qh.astype(np.uint64))
    r = drop_rand_q(qs[..., None], np.arange(ln, dtype=np.uint64)[None, None, None, :])
    return (r >= np.uint64(thresh)).astype(np.float32)
