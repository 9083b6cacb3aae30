#!/usr/bin/env python3
"""A/B harness for sweep experiments: the bench's C3 tick (lists + missiles) under different ZRK_EXP / env settings,
each in its own process.  usage: sweep_exp.py "ENV1=a ENV2=b" "ENV1=c" ...   (an empty string = defaults)"""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CHILD = r'''
import sys, time
import numpy as np, torch
sys.path.insert(0, %r)
from zrk_modulation_amd import scenario as S
from zrk_modulation_amd.engine import HotPathEngine
n, R, m = 1_000_000, 16, 10_000
ids, sp, vel, t0 = S.synthetic_targets(n, 1237)
eng = HotPathEngine(device="cuda:0", dt_ms=10, seed=1237, noise="philox")
eng.load(ids, sp, vel, t0, S.synthetic_radars(R), missile_capacity=m).enable_lists()
eng.launch_missiles(S.missile_targets(n, m))
eng.run(100)
torch.cuda.synchronize()
res = []
for rep in range(3):
    ms = np.zeros(200, np.float32)
    t = time.perf_counter(); eng.run(200, sweep_ms=ms, prof_stride=1); torch.cuda.synchronize()
    wall = (time.perf_counter() - t) / 200 * 1e6
    t = time.perf_counter(); eng.run(400); torch.cuda.synchronize()
    wall2 = (time.perf_counter() - t) / 400 * 1e6
    res.append((float(ms.mean() * 1e3), float(np.median(ms) * 1e3), wall2))
print("sweep mean/median us, tick wall us:", " | ".join("%%.2f %%.2f %%.2f" %% r for r in res), flush=True)
''' % str(ROOT)

for cfg in (sys.argv[1:] or [""]):
    env = dict(os.environ)
    for kv in cfg.split():
        k, v = kv.split("=", 1)
        env[k] = v
    out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    print(f"[{cfg or 'default'}]", out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-2000:], flush=True)
