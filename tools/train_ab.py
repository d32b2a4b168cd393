"""Same-box A/B of the training step (config 4), alternating two settings in separate bench.py processes:
    python3 tools/train_ab.py <option> <value_a> <value_b> [rounds]      e.g.  fcln 0 1
    python3 tools/train_ab.py lib <path to another libdptnav.so>          (two BUILDS: the other library against the tree's)"""
import json
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
if sys.argv[1] == "lib":
    legs = [("other", [], {"DPTNAV_LIB": os.path.abspath(sys.argv[2])}), ("tree ", [], {})]
    rounds = 3
else:
    opt, va, vb = sys.argv[1], sys.argv[2], sys.argv[3]
    legs = [(f"{opt}={v}", ["--opt", f"{opt}={v}"], {}) for v in (va, vb)]
    rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 3
for r in range(rounds):
    for name, extra, envx in legs:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "dptn_av_train", "--steps", "8", "--warmup", "3"] + extra,
                             capture_output=True, text=True, env=dict(os.environ, **envx)).stdout
        d = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])
        print(name, d["value"], d["ms_per_step"], flush=True)
