"""What the missile phase costs the small-table loop: configs[1]'s scene (1e5 targets, 4 radars) with 1000, 100 and no missiles in
flight, tick time of a 2000-tick call and the pair sweep's own duration (stamps).   usage: c2_missile_share.py"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from zrk_modulation_amd import scenario as S  # noqa: E402
from zrk_modulation_amd.engine import HotPathEngine  # noqa: E402

n, R, _ = S.WORKLOADS["C2"]
for m in (1000, 100, 0):
    ids, sp, vel, t0 = S.synthetic_targets(n, S.SEEDS["C2"])
    eng = HotPathEngine(device="cuda:0", dt_ms=10, seed=S.SEEDS["C2"], noise="philox")
    eng.load(ids, sp, vel, t0, S.synthetic_radars(R), missile_capacity=max(m, 1)).enable_lists()
    if m:
        eng.launch_missiles(S.missile_targets(n, m))
    eng.sweep_stamps(True)
    eng.run(100)
    torch.cuda.synchronize()
    t_a = time.perf_counter()
    eng.run(2000)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t_a
    us, ticks = eng.read_sweep_stamps()
    print(f"missiles in flight {eng.store.m:5d}: {dt / 2000 * 1e6:6.2f} us per tick; pair sweep first wave in to last wave out {float(np.mean(us[ticks == 2])):6.2f} us "
          f"(min {float(us[ticks == 2].min()):.2f})")
