#!/usr/bin/env python3
"""Long run of the overlapped loop against the two-launch loop (development aid):
    python tools/soak_overlap.py [n] [R] [m] [ticks] [seed]
Calls of random length; after every call the two engines must hold the same position bits (both buffers), flags,
last-tick masks, lists, ordered events and missile table (the comparison of tests/test_gpu_overlap.py)."""
import os
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from tests.test_gpu_overlap import _same, _state  # noqa: E402
from tests.test_gpu_engine import _engine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
R = int(sys.argv[2]) if len(sys.argv) > 2 else 16
m = int(sys.argv[3]) if len(sys.argv) > 3 else 10_000
ticks = int(sys.argv[4]) if len(sys.argv) > 4 else 1500
seed = int(sys.argv[5]) if len(sys.argv) > 5 else 7
os.environ["ZRK_OVERLAP_MIN_ROWS"] = "0"
os.environ["ZRK_OVERLAP"] = "0"
ref, _, launched = _engine(n, R, m, seed=seed, noise="philox")
os.environ["ZRK_OVERLAP"] = "1"
ovl, _, _ = _engine(n, R, m, seed=seed, noise="philox")
g = np.random.Generator(np.random.PCG64(seed))
done = calls = events = 0
t0 = time.time()
while done < ticks:
    K = int(g.integers(1, 41))
    ref.run(K)                      # (one call of K ticks on the two-launch loop equals K calls of one: tested elsewhere)
    ovl.run(K)
    a, b = _state(ref), _state(ovl)
    _same(a, b, f"after {done + K} ticks (call of {K})")
    done += K; calls += 1; events += int(a["ne"][0])
    if calls % 20 == 0:
        print(f"{done} ticks, {calls} calls, alive {int(a['alive'].sum())}, {time.time() - t0:.0f} s", flush=True)
print(f"ok: {done} ticks in {calls} calls, {launched} missiles launched, alive at the end {int(a['alive'].sum())} of {len(a['alive'])}")
