#!/usr/bin/env python3
"""Time the fused sweep kernel under ablations (noise on/off, radar count) with HIP events.
Development aid; not part of the product or the bench contract."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from zrk_modulation_amd import scenario as S
from zrk_modulation_amd.engine import HotPathEngine


def run(n, R, noise, lists, ticks=200, m=0):
    ids, sp, vel, t0 = S.synthetic_targets(n, 1237)
    eng = HotPathEngine(device="cuda:0", dt_ms=10, seed=1, noise=noise)
    eng.load(ids, sp, vel, t0, S.synthetic_radars(R), missile_capacity=m)
    if lists:
        eng.enable_lists()
    if m:
        eng.launch_missiles(S.missile_targets(n, m))
    eng.run(20)
    ms = np.zeros(ticks, np.float32)
    import time, torch
    torch.cuda.synchronize()
    t = time.perf_counter()
    eng.run(ticks, sweep_ms=ms, prof_stride=1)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t) / ticks * 1e6
    return float(ms.mean() * 1e3), wall


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    print(f"n={n}")
    for noise in ("off", "philox"):
        for R in (0, 1, 2, 4, 8, 16, 32):
            k, w = run(n, R, noise, lists=False)
            print(f"noise={noise:6s} R={R:2d}  sweep={k:7.1f} us   tick wall={w:7.1f} us", flush=True)
    k, w = run(n, 16, "philox", lists=True, m=10000)
    print(f"full tick (lists + missiles): sweep={k:.1f} us, tick wall={w:.1f} us")
