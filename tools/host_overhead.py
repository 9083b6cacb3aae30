#!/usr/bin/env python3
"""Pure host cost of one tick of zrk_run_ticks (a table so small that the device never is the limit)."""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from zrk_modulation_amd import scenario as S
from zrk_modulation_amd.engine import HotPathEngine
for n, R, m in [(2048, 16, 64), (2048, 4, 0), (2048, 16, 0)]:
    ids, sp, vel, t0 = S.synthetic_targets(n, 1)
    eng = HotPathEngine(device="cuda:0", dt_ms=10, seed=1, noise="philox")
    eng.load(ids, sp, vel, t0, S.synthetic_radars(R), missile_capacity=m).enable_lists()
    if m:
        eng.launch_missiles(S.missile_targets(n, m))
    eng.run(200); torch.cuda.synchronize()
    t = time.perf_counter(); eng.run(100); host = (time.perf_counter() - t) * 20
    torch.cuda.synchronize(); wall = (time.perf_counter() - t) * 20
    print(f"n={n} R={R} m={m}: host {host / 2000 * 1e6:.2f} us per tick, wall {wall / 2000 * 1e6:.2f}")
