#!/usr/bin/env python3
"""How often a 20-tick overlapped call is held up by the library's own threads waking late (they sleep after a millisecond
without work and are woken at the call's entry):  python tools/stall_probe.py [calls] [pause_ms]
Calls of 20 ticks at C3 size with a pause in between that lets the helper threads fall asleep; prints the distribution of
call + synchronisation times and every call that took more than twice the median."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from zrk_modulation_amd import scenario as S  # noqa: E402
from zrk_modulation_amd.engine import HotPathEngine  # noqa: E402

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
pause_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
n, R, m = 1_000_000, 16, 10_000
ids, sp, vel, t0 = S.synthetic_targets(n, 1237)
eng = HotPathEngine(device="cuda:0", dt_ms=10, seed=1, noise="philox")
eng.load(ids, sp, vel, t0, S.synthetic_radars(R), missile_capacity=m).enable_lists()
eng.launch_missiles(S.missile_targets(n, m))
eng.run(40)
torch.cuda.synchronize()
ts = np.zeros(calls)
for k in range(calls):
    time.sleep(pause_ms * 1e-3)
    a = time.perf_counter()
    eng.run(20)
    torch.cuda.synchronize()
    ts[k] = (time.perf_counter() - a) * 1e6
med = float(np.median(ts))
late = np.nonzero(ts > 2 * med)[0]
print(f"{calls} calls of 20 ticks, {pause_ms} ms apart: median {med:.0f} us, p90 {np.percentile(ts, 90):.0f}, p99 {np.percentile(ts, 99):.0f}, "
      f"max {ts.max():.0f}; {len(late)} calls took more than twice the median: {[int(ts[i]) for i in late[:20]]}")
