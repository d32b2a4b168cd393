"""CPU-side checks: C-ABI surface, checkpoint/shape spec, nn.Module drop-in contract (no compute calls)."""
import os
import re

import numpy as np
import pytest
import torch

from speech_separation_amd import _lib
from speech_separation_amd.spec import (DPTN_AUDIO, DPTN_AV, DPTN_TINY, DPTNConfig, num_parameters, state_dict_spec,
                                        synthetic_inputs, synthetic_state_dict)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "dptnav.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(dptnav_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    lib = _lib.load()                       # raises if the .so is missing or a symbol is absent
    assert lib.dptnav_abi_version() == _lib.ABI_VERSION
    assert lib.dptnav_profile_num() == 11


@pytest.mark.skipif(torch.cuda.is_available(), reason="only meaningful on a box without a GPU")
def test_no_cpu_path_without_gpu():
    import ctypes as C
    lib = _lib.load()
    h = C.c_void_p()
    cfg = _lib.DptnavConfig(128, 512, 128, 7, 128, 6, 150, 75, 4, 1, 0, 0)
    assert lib.dptnav_create(C.byref(cfg), C.byref(h)) != 0
    assert b"no CPU path" in lib.dptnav_last_error(None)
    from speech_separation_amd.engine import DptnEngine
    with pytest.raises(RuntimeError):
        DptnEngine(DPTN_AV, "cpu")


def test_spec_matches_reference_checkpoint_surface():
    spec = state_dict_spec(DPTN_AV)
    assert len(spec) == 228 and num_parameters(DPTN_AV) == 4_448_194
    assert num_parameters(DPTN_AUDIO) == 2_797_377                              # paper.tex "2.8 M"
    keys = [k for k, _ in spec]
    assert keys[0] == "gate" and keys[1] == "encoder.weight" and keys[-1] == "decoder.weight"
    assert dict(spec)["dprnn.model.3.inter_chunk_block.mha.in_proj_weight"] == (384, 128)
    assert dict(spec)["dprnn.model.5.intra_chunk_block.rnn.weight_hh_l0_reverse"] == (512, 128)
    uni = DPTNConfig(**{**DPTN_AV.to_dict(), "bidir": False})
    d = dict(state_dict_spec(uni))
    assert "dprnn.model.0.inter_chunk_block.rnn.weight_ih_l0_reverse" not in d
    assert "dprnn.model.0.intra_chunk_block.rnn.weight_ih_l0_reverse" in d        # intra is always bidirectional
    assert d["dprnn.model.0.inter_chunk_block.ffn.1.weight"] == (128, 128)
    assert d["dprnn.model.0.intra_chunk_block.ffn.1.weight"] == (128, 256)


def test_derived_sizes():
    assert DPTN_AV.frames(32000) == 10665 and DPTN_AV.chunks(10665) == 141
    assert DPTN_AV.ola_len(141) == 10650 and DPTN_AV.tokens(16, 32000) == 338400
    assert DPTN_TINY.frames(209) == 68 and DPTN_TINY.chunks(68) == 12


def test_synthetic_data_is_deterministic():
    a, b = synthetic_state_dict(DPTN_TINY, 3), synthetic_state_dict(DPTN_TINY, 3)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    i = synthetic_inputs(DPTN_TINY, B=2, T=209, Tv=9)
    assert np.allclose(i["mix"], i["s1"] + i["s2"]) and i["s1_embedding"].shape == (2, 24, 9)


def test_module_is_a_dropin_for_the_reference_class():
    from speech_separation_amd import DPTNAVWavEncDec, DPTNWavEncDec
    kw = dict(num_features=128, video_emb_size=512, hidden_video=128, kernel_size_enc=7, hidden_dim=128, num_blocks=6,
              chunk_size=150, step_size=75, dropout=0.1, num_heads=4, bidir=True)   # src/configs/model/dptn_wav_av.yaml
    m = DPTNAVWavEncDec(**kw)
    sd = m.state_dict()
    assert [(k, tuple(v.shape)) for k, v in sd.items()] == state_dict_spec(DPTN_AV)
    assert str(m).endswith("All parameters: 4448194\nTrainable parameters: 4448194")   # dptn_wav.py:196-207
    # a reference-format checkpoint loads strictly (base_trainer.py:557-560 accepts raw or wrapped dicts)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic_state_dict(DPTN_AV, 1).items()}, strict=True)
    assert all(torch.isfinite(p).all() for p in m.parameters())
    # compute is GPU-only: a CPU call must raise, never fall back
    with torch.no_grad(), pytest.raises(RuntimeError, match="no CPU"):
        m(mix=torch.zeros(1, 32000), s1_embedding=torch.zeros(1, 512, 50), s2_embedding=torch.zeros(1, 512, 50))
    a = DPTNWavEncDec(num_features=64, kernel_size_enc=7, hidden_dim=128, num_blocks=6, chunk_size=150, step_size=75,
                      num_heads=4, dropout=0.1, bidir=True)                            # dptn_wav.yaml
    assert [(k, tuple(v.shape)) for k, v in a.state_dict().items()] == state_dict_spec(DPTN_AUDIO)


def test_dprnn_dropin_state_dict():
    from speech_separation_amd import DPRNNAVEncDec, DPRNNEncDec
    from speech_separation_amd.spec import DPRNN_AUDIO, DPRNN_AV
    m = DPRNNEncDec(num_features=64, kernel_size_enc=2, hidden_dim=128, num_blocks=6, chunk_size=250, step_size=125,
                    bidir=True)                                                    # src/configs/model/dprnn.yaml
    assert [(k, tuple(v.shape)) for k, v in m.state_dict().items()] == state_dict_spec(DPRNN_AUDIO)
    assert sum(p.numel() for p in m.parameters()) == 2_595_521                      # paper.tex "2.6 M"
    av = DPRNNAVEncDec(num_features=64, hidden_video=64, kernel_size_enc=2, hidden_dim=128, num_blocks=6,
                       chunk_size=250, step_size=125)
    assert [(k, tuple(v.shape)) for k, v in av.state_dict().items()] == state_dict_spec(DPRNN_AV)


def test_product_code_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "speech_separation_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b|oracle[./](dptn_oracle|torch_stock)|import_module\(.oracle",
                                     src, flags=re.M), f"{f} reaches into oracle/"


def test_dropout_generator_statistics():
    """The counter-based keep-mask generator (tests/dropout_ref.py = csrc/common.h drop_rand): keep rate, uniformity
    and independence between neighbouring keys / queries / heads."""
    from dropout_ref import drop_qseed, drop_rand_q, keep_mask
    m = keep_mask(1, 0, 2, 3, 150, 4, 100000, 777).astype(np.float64)       # (6, 4, 150, 150)
    assert abs(m.mean() - 0.9) < 3e-3
    c = m - m.mean()
    var = c.var()
    tol = 6.0 / np.sqrt(c.size)
    assert abs((c[..., 1:] * c[..., :-1]).mean() / var) < tol               # adjacent keys
    assert abs((c[..., 1:, :] * c[..., :-1, :]).mean() / var) < tol         # adjacent queries
    assert abs((c[:, 1:] * c[:, :-1]).mean() / var) < tol                   # adjacent heads
    assert not np.array_equal(m, keep_mask(1, 0, 2, 3, 150, 4, 100000, 778))   # the seed matters
    assert not np.array_equal(m, keep_mask(2, 0, 2, 3, 150, 4, 100000, 777))   # and so does the call site
    q = np.arange(1 << 14, dtype=np.uint64)
    u = drop_rand_q(drop_qseed(123, q)[:, None], np.arange(160, dtype=np.uint64)[None, :]) / 2.0 ** 24
    assert abs(u.mean() - 0.5) < 2e-3 and abs(u.var() - 1 / 12) < 2e-3
    h = np.histogram(u, bins=64)[0] / (u.size / 64)
    assert h.min() > 0.97 and h.max() < 1.03


def _build_c_consumer(tmp_path):
    """gcc -std=c99 -pedantic on tests/cabi/consumer.c against include/dptnav.h + libdptnav.so: the header is plain C and the
    library links from C (no C++ runtime symbols, no HIP headers, no torch on the consumer's side)."""
    import subprocess
    from speech_separation_amd.build import OUT, build_lib
    build_lib()
    exe = str(tmp_path / "consumer")
    libdir = os.path.dirname(OUT)
    cmd = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cabi", "consumer.c"),
           "-o", exe, "-L", libdir, "-l:libdptnav.so", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


def test_c_consumer_compiles_links_and_gets_clean_errors(tmp_path):
    import subprocess
    exe = _build_c_consumer(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "ok host" in r.stdout, r.stdout + r.stderr
