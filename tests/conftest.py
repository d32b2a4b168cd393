"""pytest configuration: registers the ``gpu`` marker and exposes the repo root on sys.path."""
import ast
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:   # tests/dropout_ref.py
    sys.path.insert(0, HERE)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """-> (cfg: DPTNConfig, arrays: dict) for tests/golden/<name>.npz"""
    from speech_separation_amd.spec import DPTNConfig
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    cfg = DPTNConfig(**ast.literal_eval(str(z["cfg"])))
    return cfg, {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden():
    return load_golden
