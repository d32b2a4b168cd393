import os, sys, time, json, subprocess
# alternate two builds of the library in separate processes (same box): training step of config 4
for r in range(3):
    for lib in ("ab_build/base.so", ""):
        env = dict(os.environ)
        if lib: env["DPTNAV_LIB"] = os.path.abspath(lib)
        out = subprocess.run([sys.executable, "bench.py", "--config", "dptn_av_train", "--steps", "8", "--warmup", "3"], capture_output=True, text=True, env=env).stdout
        d = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
        print("base" if lib else "new ", d["value"], d["ms_per_step"], flush=True)
